// fiber-quad (MFMA) instantiations: synthetic 7-D car, one kernel per varying dimension
#include "launch_fpw.hpp"
#include "launch_fq.hpp"
#include "models.hpp"
namespace c3sc {
#define REG7Q(RP, NWV)                                 \
    C3SC_REG_FQ1(C3SC_MODEL_CAR7D, RP, 0, NWV, Car7D)  \
    C3SC_REG_FQ1(C3SC_MODEL_CAR7D, RP, 1, NWV, Car7D)  \
    C3SC_REG_FQ1(C3SC_MODEL_CAR7D, RP, 2, NWV, Car7D)  \
    C3SC_REG_FQ1(C3SC_MODEL_CAR7D, RP, 3, NWV, Car7D)  \
    C3SC_REG_FQ1(C3SC_MODEL_CAR7D, RP, 4, NWV, Car7D)  \
    C3SC_REG_FQ1(C3SC_MODEL_CAR7D, RP, 5, NWV, Car7D)  \
    C3SC_REG_FQ1(C3SC_MODEL_CAR7D, RP, 6, NWV, Car7D)
REG7Q(4, 8)
REG7Q(12, 8)
} // namespace c3sc
