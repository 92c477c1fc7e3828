// fiber-per-lane instantiations: synthetic 7-D car, one kernel per varying dimension
#include "launch_fpw.hpp"
#include "launch_fpl.hpp"
#include "models.hpp"
namespace c3sc {
#define REG7(RP)                                   \
    C3SC_REG_FPL1(C3SC_MODEL_CAR7D, RP, 0, Car7D)  \
    C3SC_REG_FPL1(C3SC_MODEL_CAR7D, RP, 1, Car7D)  \
    C3SC_REG_FPL1(C3SC_MODEL_CAR7D, RP, 2, Car7D)  \
    C3SC_REG_FPL1(C3SC_MODEL_CAR7D, RP, 3, Car7D)  \
    C3SC_REG_FPL1(C3SC_MODEL_CAR7D, RP, 4, Car7D)  \
    C3SC_REG_FPL1(C3SC_MODEL_CAR7D, RP, 5, Car7D)  \
    C3SC_REG_FPL1(C3SC_MODEL_CAR7D, RP, 6, Car7D)
REG7(4)
REG7(10)
} // namespace c3sc
