// fiber-per-lane (one wavefront per 64 fibers, no partner) instantiations: the low-rank, low-dimensional models
#include "launch_fl.hpp"
#include "models.hpp"
// registers: four wavefronts per SIMD (<= 128 VGPRs) at rank 4, three (<= 168) above
#define FL_WPS(RP) ((RP) <= 4 ? 4 : 3)
namespace c3sc {
#define REG3L(MID, RP, ...)                      \
    C3SC_REG_FL1(MID, RP, 0, FL_WPS(RP), __VA_ARGS__) \
    C3SC_REG_FL1(MID, RP, 1, FL_WPS(RP), __VA_ARGS__) \
    C3SC_REG_FL1(MID, RP, 2, FL_WPS(RP), __VA_ARGS__)
REG3L(C3SC_MODEL_DUBINS3D, 4, Dubins3D)
REG3L(C3SC_MODEL_DUBINS3D, 6, Dubins3D)
REG3L(C3SC_MODEL_DUBINS3D, 8, Dubins3D)
REG3L(C3SC_MODEL_ROSSLER3D, 4, Rossler3D)
REG3L(C3SC_MODEL_ROSSLER3D, 8, Rossler3D)
C3SC_REG_FL1(C3SC_MODEL_LQGND, 4, 0, FL_WPS(4), LqgNd<2>)
C3SC_REG_FL1(C3SC_MODEL_LQGND, 4, 1, FL_WPS(4), LqgNd<2>)
} // namespace c3sc
