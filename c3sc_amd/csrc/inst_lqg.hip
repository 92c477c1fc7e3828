// kernel instantiations: n-D LQG (examples/lqgnd, examples/lqg2d_new) + 2-D / 6-D stencil
#include "launch_fpw.hpp"
#include "models.hpp"
namespace c3sc {
C3SC_REG_FPW_BOX(C3SC_MODEL_LQGND, 4, 1, LqgNd<2>)
// ranks an adaptive cross approximation of the 2-D problems reaches (c3control_vi_solve / pi_solve, N up to 128)
C3SC_REG_FPW_BOX(C3SC_MODEL_LQGND, 4, 2, LqgNd<2>)
C3SC_REG_FPW_BOX(C3SC_MODEL_LQGND, 8, 1, LqgNd<2>)
C3SC_REG_FPW_BOX(C3SC_MODEL_LQGND, 8, 2, LqgNd<2>)
C3SC_REG_FPW_BOX(C3SC_MODEL_LQGND, 12, 1, LqgNd<2>)
C3SC_REG_FPW_BOX(C3SC_MODEL_LQGND, 12, 2, LqgNd<2>)
C3SC_REG_FPW_BOX(C3SC_MODEL_LQGND, 20, 1, LqgNd<2>)
C3SC_REG_FPW_BOX(C3SC_MODEL_LQGND, 20, 2, LqgNd<2>)
C3SC_REG_FPW_BOX(C3SC_MODEL_LQGND, 4, 1, LqgNd<6>)
C3SC_REG_FPW_BOX(C3SC_MODEL_LQGND, 8, 1, LqgNd<6>)
C3SC_REG_FPW(C3SC_MODEL_CHAIN, 4, 1, Chain<2>)
C3SC_REG_STENCIL(2, 4, 1)
C3SC_REG_STENCIL(2, 4, 2)
C3SC_REG_STENCIL(2, 8, 1)
C3SC_REG_STENCIL(2, 8, 2)
C3SC_REG_STENCIL(2, 12, 1)
C3SC_REG_STENCIL(2, 12, 2)
C3SC_REG_STENCIL(2, 20, 1)
C3SC_REG_STENCIL(2, 20, 2)
C3SC_REG_STENCIL(6, 4, 1)
C3SC_REG_STENCIL(6, 8, 1)
} // namespace c3sc
