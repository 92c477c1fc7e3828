// kernel instantiations: 3-D Dubins car (examples/dubinscar_new) + 3-D stencil
#include "launch_fpw.hpp"
#include "models.hpp"
namespace c3sc {
C3SC_REG_FPW(C3SC_MODEL_DUBINS3D, 4, 1, Dubins3D)
C3SC_REG_FPW(C3SC_MODEL_DUBINS3D, 4, 2, Dubins3D)
C3SC_REG_FPW(C3SC_MODEL_DUBINS3D, 6, 1, Dubins3D)
C3SC_REG_FPW(C3SC_MODEL_DUBINS3D, 6, 2, Dubins3D)
C3SC_REG_FPW(C3SC_MODEL_DUBINS3D, 8, 1, Dubins3D)
C3SC_REG_FPW(C3SC_MODEL_DUBINS3D, 8, 2, Dubins3D)
C3SC_REG_STENCIL(3, 4, 1)
C3SC_REG_STENCIL(3, 4, 2)
C3SC_REG_STENCIL(3, 6, 1)
C3SC_REG_STENCIL(3, 6, 2)
C3SC_REG_STENCIL(3, 8, 1)
C3SC_REG_STENCIL(3, 8, 2)
} // namespace c3sc
