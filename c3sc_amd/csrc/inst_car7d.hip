// kernel instantiations: synthetic 7-D car + 7-D stencil
#include "launch_fpw.hpp"
#include "models.hpp"
namespace c3sc {
C3SC_REG_FPW(C3SC_MODEL_CAR7D, 4, 1, Car7D)
C3SC_REG_FPW(C3SC_MODEL_CAR7D, 10, 1, Car7D)
// the ranks an adaptive cross approximation may reach (valuefunc.c:636-649 kicks the rank up to maxrank)
C3SC_REG_FPW(C3SC_MODEL_CAR7D, 12, 1, Car7D)
C3SC_REG_FPW(C3SC_MODEL_CAR7D, 16, 1, Car7D)
C3SC_REG_FPW(C3SC_MODEL_CAR7D, 20, 1, Car7D)
C3SC_REG_STENCIL(7, 4, 1)
C3SC_REG_STENCIL(7, 10, 1)
C3SC_REG_STENCIL(7, 12, 1)
C3SC_REG_STENCIL(7, 16, 1)
C3SC_REG_STENCIL(7, 20, 1)
} // namespace c3sc
