// fiber-pair (rank-split) instantiations: synthetic 7-D car, one kernel per varying dimension
#include "launch_fpw.hpp"
#include "launch_fpp.hpp"
#include "models.hpp"
namespace c3sc {
#define REG7P(RP)                                  \
    C3SC_REG_FPP1(C3SC_MODEL_CAR7D, RP, 0, Car7D)  \
    C3SC_REG_FPP1(C3SC_MODEL_CAR7D, RP, 1, Car7D)  \
    C3SC_REG_FPP1(C3SC_MODEL_CAR7D, RP, 2, Car7D)  \
    C3SC_REG_FPP1(C3SC_MODEL_CAR7D, RP, 3, Car7D)  \
    C3SC_REG_FPP1(C3SC_MODEL_CAR7D, RP, 4, Car7D)  \
    C3SC_REG_FPP1(C3SC_MODEL_CAR7D, RP, 5, Car7D)  \
    C3SC_REG_FPP1(C3SC_MODEL_CAR7D, RP, 6, Car7D)
REG7P(4)
REG7P(10)
} // namespace c3sc
