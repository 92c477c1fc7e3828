// fiber-pair (rank-split) instantiations: 6-D LQG (examples/lqgnd)
#include "launch_fpw.hpp"
#include "launch_fpp.hpp"
#include "models.hpp"
namespace c3sc {
#define REG6P(RP)                                     \
    C3SC_REG_FPP1(C3SC_MODEL_LQGND, RP, 0, LqgNd<6>)  \
    C3SC_REG_FPP1(C3SC_MODEL_LQGND, RP, 1, LqgNd<6>)  \
    C3SC_REG_FPP1(C3SC_MODEL_LQGND, RP, 2, LqgNd<6>)  \
    C3SC_REG_FPP1(C3SC_MODEL_LQGND, RP, 3, LqgNd<6>)  \
    C3SC_REG_FPP1(C3SC_MODEL_LQGND, RP, 4, LqgNd<6>)  \
    C3SC_REG_FPP1(C3SC_MODEL_LQGND, RP, 5, LqgNd<6>)
REG6P(4)
REG6P(8)
} // namespace c3sc
