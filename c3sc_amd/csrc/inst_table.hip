// kernel instantiations: universal table-driven model (host-evaluated callbacks), fiber-per-wave
#include "launch_fpw.hpp"
#include "models.hpp"
namespace c3sc {
C3SC_REG_FPW(C3SC_MODEL_TABLE, 4, 1, TableModel<2>)
C3SC_REG_FPW(C3SC_MODEL_TABLE, 4, 1, TableModel<3>)
C3SC_REG_FPW(C3SC_MODEL_TABLE, 4, 2, TableModel<3>)
C3SC_REG_FPW(C3SC_MODEL_TABLE, 6, 1, TableModel<3>)
C3SC_REG_FPW(C3SC_MODEL_TABLE, 6, 2, TableModel<3>)
C3SC_REG_FPW(C3SC_MODEL_TABLE, 8, 1, TableModel<3>)
C3SC_REG_FPW(C3SC_MODEL_TABLE, 8, 2, TableModel<3>)
C3SC_REG_FPW(C3SC_MODEL_TABLE, 4, 1, TableModel<4>)
C3SC_REG_FPW(C3SC_MODEL_TABLE, 8, 1, TableModel<4>)
C3SC_REG_FPW(C3SC_MODEL_TABLE, 20, 1, TableModel<4>)
C3SC_REG_FPW(C3SC_MODEL_TABLE, 4, 1, TableModel<6>)
C3SC_REG_FPW(C3SC_MODEL_TABLE, 8, 1, TableModel<6>)
C3SC_REG_FPW(C3SC_MODEL_TABLE, 4, 1, TableModel<7>)
C3SC_REG_FPW(C3SC_MODEL_TABLE, 10, 1, TableModel<7>)
C3SC_REG_FPW(C3SC_MODEL_TABLE, 4, 1, TableModel<10>)
C3SC_REG_FPW(C3SC_MODEL_TABLE, 16, 1, TableModel<10>)
} // namespace c3sc
