// kernel instantiations: n-D LQG (examples/lqgnd, examples/lqg2d_new) + 2-D / 6-D stencil
#include "launch_fpw.hpp"
#include "models.hpp"
namespace c3sc {
C3SC_REG_FPW(C3SC_MODEL_LQGND, 4, 1, LqgNd<2>)
C3SC_REG_FPW(C3SC_MODEL_LQGND, 4, 1, LqgNd<6>)
C3SC_REG_FPW(C3SC_MODEL_LQGND, 8, 1, LqgNd<6>)
C3SC_REG_FPW(C3SC_MODEL_CHAIN, 4, 1, Chain<2>)
C3SC_REG_STENCIL(2, 4, 1)
C3SC_REG_STENCIL(6, 4, 1)
C3SC_REG_STENCIL(6, 8, 1)
} // namespace c3sc
