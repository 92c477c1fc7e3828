// fiber-pair (rank-split) instantiations for the other models (one kernel per varying dimension)
#include "launch_fpw.hpp"
#include "launch_fpp.hpp"
#include "models.hpp"
namespace c3sc {
#define REG3P(RP)                                        \
    C3SC_REG_FPP1(C3SC_MODEL_DUBINS3D, RP, 0, Dubins3D)  \
    C3SC_REG_FPP1(C3SC_MODEL_DUBINS3D, RP, 1, Dubins3D)  \
    C3SC_REG_FPP1(C3SC_MODEL_DUBINS3D, RP, 2, Dubins3D)
REG3P(4)
REG3P(6)
REG3P(8)
#define REG3R(RP)                                          \
    C3SC_REG_FPP1(C3SC_MODEL_ROSSLER3D, RP, 0, Rossler3D)  \
    C3SC_REG_FPP1(C3SC_MODEL_ROSSLER3D, RP, 1, Rossler3D)  \
    C3SC_REG_FPP1(C3SC_MODEL_ROSSLER3D, RP, 2, Rossler3D)
REG3R(4)
REG3R(8)
#define REG4P(RP)                                    \
    C3SC_REG_FPP1(C3SC_MODEL_SCAR4D, RP, 0, Scar4D)  \
    C3SC_REG_FPP1(C3SC_MODEL_SCAR4D, RP, 1, Scar4D)  \
    C3SC_REG_FPP1(C3SC_MODEL_SCAR4D, RP, 2, Scar4D)  \
    C3SC_REG_FPP1(C3SC_MODEL_SCAR4D, RP, 3, Scar4D)
REG4P(4)
REG4P(8)
C3SC_REG_FPP1(C3SC_MODEL_LQGND, 4, 0, LqgNd<2>)
C3SC_REG_FPP1(C3SC_MODEL_LQGND, 4, 1, LqgNd<2>)
} // namespace c3sc
