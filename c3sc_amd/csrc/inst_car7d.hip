// kernel instantiations: synthetic 7-D car + 7-D stencil
#include "launch_fpw.hpp"
#include "models.hpp"
namespace c3sc {
C3SC_REG_FPW(C3SC_MODEL_CAR7D, 4, 1, Car7D)
C3SC_REG_FPW(C3SC_MODEL_CAR7D, 10, 1, Car7D)
C3SC_REG_STENCIL(7, 4, 1)
C3SC_REG_STENCIL(7, 10, 1)
} // namespace c3sc
