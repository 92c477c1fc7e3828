// kernel instantiations: examples/skidding5d (5-D skidding car, candidate lists) and examples/cothrust2 (6-D quadcopter, three
// continuous controls: candidate lists and the box minimiser)
#include "launch_fpw.hpp"
#include "models.hpp"
namespace c3sc {
C3SC_REG_FPW(C3SC_MODEL_SKID5D, 4, 1, Skid5D)
C3SC_REG_FPW(C3SC_MODEL_SKID5D, 8, 1, Skid5D)
C3SC_REG_FPW(C3SC_MODEL_SKID5D, 12, 1, Skid5D)
C3SC_REG_FPW(C3SC_MODEL_SKID5D, 16, 1, Skid5D) // the example's maxrank is 15
C3SC_REG_FPW(C3SC_MODEL_SKID5D, 20, 1, Skid5D)
C3SC_REG_FPW_BOX(C3SC_MODEL_COTHRUST6D, 4, 1, Cothrust6D)
C3SC_REG_FPW_BOX(C3SC_MODEL_COTHRUST6D, 8, 1, Cothrust6D)
C3SC_REG_FPW_BOX(C3SC_MODEL_COTHRUST6D, 12, 1, Cothrust6D) // the example's maxrank is 10
C3SC_REG_FPW_BOX(C3SC_MODEL_COTHRUST6D, 16, 1, Cothrust6D)
C3SC_REG_FPW_BOX(C3SC_MODEL_COTHRUST6D, 20, 1, Cothrust6D)
C3SC_REG_STENCIL(5, 4, 1)
C3SC_REG_STENCIL(5, 8, 1)
C3SC_REG_STENCIL(5, 12, 1)
C3SC_REG_STENCIL(5, 16, 1)
C3SC_REG_STENCIL(5, 20, 1)
} // namespace c3sc
