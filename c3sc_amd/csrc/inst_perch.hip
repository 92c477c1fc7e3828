// kernel instantiations: examples/perching (7-D glider, one continuous control): candidate lists and the box minimiser
#include "launch_fpw.hpp"
#include "models.hpp"
namespace c3sc {
C3SC_REG_FPW_BOX(C3SC_MODEL_PERCH7D, 4, 1, Perch7D)
C3SC_REG_FPW_BOX(C3SC_MODEL_PERCH7D, 8, 1, Perch7D)
C3SC_REG_FPW_BOX(C3SC_MODEL_PERCH7D, 12, 1, Perch7D)
C3SC_REG_FPW_BOX(C3SC_MODEL_PERCH7D, 16, 1, Perch7D) // the example's maxrank is 15
C3SC_REG_FPW_BOX(C3SC_MODEL_PERCH7D, 20, 1, Perch7D)
} // namespace c3sc
