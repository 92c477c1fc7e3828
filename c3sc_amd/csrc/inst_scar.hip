// kernel instantiations: 4-D skidding car (examples/skidding_car) + 4-D chain + 4-D stencil
#include "launch_fpw.hpp"
#include "models.hpp"
namespace c3sc {
C3SC_REG_FPW(C3SC_MODEL_SCAR4D, 4, 1, Scar4D)
C3SC_REG_FPW(C3SC_MODEL_SCAR4D, 8, 1, Scar4D)
C3SC_REG_FPW(C3SC_MODEL_SCAR4D, 12, 1, Scar4D)
C3SC_REG_FPW(C3SC_MODEL_SCAR4D, 16, 1, Scar4D)
C3SC_REG_FPW(C3SC_MODEL_SCAR4D, 20, 1, Scar4D)
C3SC_REG_FPW(C3SC_MODEL_CHAIN, 4, 1, Chain<4>)
C3SC_REG_FPW_BOX(C3SC_MODEL_LQGND, 4, 1, LqgNd<4>)
C3SC_REG_STENCIL(4, 4, 1)
C3SC_REG_STENCIL(4, 8, 1)
C3SC_REG_STENCIL(4, 12, 1)
C3SC_REG_STENCIL(4, 16, 1)
C3SC_REG_STENCIL(4, 20, 1)
} // namespace c3sc
