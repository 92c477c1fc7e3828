// kernel instantiations: 10-D integrator chain (examples/double_int, synthetic quad10d) + 10-D stencil
#include "launch_fpw.hpp"
#include "models.hpp"
namespace c3sc {
C3SC_REG_FPW(C3SC_MODEL_CHAIN, 4, 1, Chain<10>)
C3SC_REG_FPW(C3SC_MODEL_CHAIN, 8, 1, Chain<10>)
C3SC_REG_FPW(C3SC_MODEL_CHAIN, 12, 1, Chain<10>)
C3SC_REG_FPW(C3SC_MODEL_CHAIN, 16, 1, Chain<10>)
C3SC_REG_FPW(C3SC_MODEL_CHAIN, 20, 1, Chain<10>)
C3SC_REG_STENCIL(10, 4, 1)
C3SC_REG_STENCIL(10, 8, 1)
C3SC_REG_STENCIL(10, 12, 1)
C3SC_REG_STENCIL(10, 16, 1)
C3SC_REG_STENCIL(10, 20, 1)
} // namespace c3sc
