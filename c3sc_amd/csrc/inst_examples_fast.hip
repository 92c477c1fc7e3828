// pair / quad kernels for the three examples that so far only had the fiber-per-wave kernel (0.10-0.14 of the FP64 roofline at
// their own sizes, 2^17 / 2^20 fibers): examples/cothrust2 (6-D, rank 10) on the fiber-pair kernel like car7d: 0.135 -> 0.42 / 0.49;
// examples/skidding5d (5-D, rank 15 -> 16) and examples/perching (7-D, rank 15 -> 16, 41 candidates, a square root per candidate) on the
// fiber-quad kernels (f64 MFMA for the shared-core products).  Measured, in this order: one wavefront per 16 fibers at two
// wavefronts per SIMD (114-156 / ~290 spilled VGPRs): skid5d 0.28, perch7d 0.15; the same with one wavefront per SIMD and the whole
// register file: 0.28 / 0.23; TWO wavefronts per 16 fibers (k_fiber_quad_duo, half of the neighbour vectors each: no spills, two
// wavefronts per SIMD): **skid5d 0.107 -> 0.33 / 0.34, perch7d 0.10 -> 0.26 / 0.26** -- registered first, the single-wavefront forms
// stay behind them for grids whose cores do not fit the duo kernel's staging (skid5d at N = 40: one staging buffer instead of two)
#include "launch_fpw.hpp"
#include "launch_fpp.hpp"
#include "launch_fq.hpp"
#include "models.hpp"
namespace c3sc {
C3SC_REG_FPW_BOX(C3SC_MODEL_COTHRUST6D, 10, 1, Cothrust6D) // small batches at the pair kernel's padded rank
C3SC_REG_FPP1(C3SC_MODEL_COTHRUST6D, 10, 0, Cothrust6D)
C3SC_REG_FPP1(C3SC_MODEL_COTHRUST6D, 10, 1, Cothrust6D)
C3SC_REG_FPP1(C3SC_MODEL_COTHRUST6D, 10, 2, Cothrust6D)
C3SC_REG_FPP1(C3SC_MODEL_COTHRUST6D, 10, 3, Cothrust6D)
C3SC_REG_FPP1(C3SC_MODEL_COTHRUST6D, 10, 4, Cothrust6D)
C3SC_REG_FPP1(C3SC_MODEL_COTHRUST6D, 10, 5, Cothrust6D)
#define REG_FQD_SB(MODEL_ID, RP, K, NWV, ...)                                                                               \
    static Registrar C3SC_CAT(reg_fqds_, __COUNTER__)(KernelEntry{                                                          \
        MODEL_ID, __VA_ARGS__::D, RP, 0, C3SC_VARIANT_FIBER_QUAD, 128, K, &launch_fq_duo<__VA_ARGS__, RP, K, NWV, false>,  \
        "k_fiber_quad_duo<" #__VA_ARGS__ "," #RP ",K=" #K ">"});
REG_FQD_SB(C3SC_MODEL_SKID5D, 16, 0, 8, Skid5D)
REG_FQD_SB(C3SC_MODEL_SKID5D, 16, 1, 8, Skid5D)
REG_FQD_SB(C3SC_MODEL_SKID5D, 16, 3, 8, Skid5D)
REG_FQD_SB(C3SC_MODEL_SKID5D, 16, 4, 8, Skid5D)
C3SC_REG_FQD(C3SC_MODEL_PERCH7D, 16, 0, 8, Perch7D)
C3SC_REG_FQD(C3SC_MODEL_PERCH7D, 16, 1, 8, Perch7D)
C3SC_REG_FQD(C3SC_MODEL_PERCH7D, 16, 2, 8, Perch7D)
C3SC_REG_FQD(C3SC_MODEL_PERCH7D, 16, 3, 8, Perch7D)
C3SC_REG_FQD(C3SC_MODEL_PERCH7D, 16, 4, 8, Perch7D)
C3SC_REG_FQD(C3SC_MODEL_PERCH7D, 16, 5, 8, Perch7D)
C3SC_REG_FQD(C3SC_MODEL_PERCH7D, 16, 6, 8, Perch7D)
C3SC_REG_FQ1(C3SC_MODEL_SKID5D, 16, 0, 8, Skid5D)
C3SC_REG_FQ1(C3SC_MODEL_SKID5D, 16, 1, 8, Skid5D)
C3SC_REG_FQ1(C3SC_MODEL_SKID5D, 16, 2, 8, Skid5D)
C3SC_REG_FQ1(C3SC_MODEL_SKID5D, 16, 3, 8, Skid5D)
C3SC_REG_FQ1(C3SC_MODEL_SKID5D, 16, 4, 8, Skid5D)
C3SC_REG_FQ1(C3SC_MODEL_PERCH7D, 16, 0, 4, Perch7D)
C3SC_REG_FQ1(C3SC_MODEL_PERCH7D, 16, 1, 4, Perch7D)
C3SC_REG_FQ1(C3SC_MODEL_PERCH7D, 16, 2, 4, Perch7D)
C3SC_REG_FQ1(C3SC_MODEL_PERCH7D, 16, 3, 4, Perch7D)
C3SC_REG_FQ1(C3SC_MODEL_PERCH7D, 16, 4, 4, Perch7D)
C3SC_REG_FQ1(C3SC_MODEL_PERCH7D, 16, 5, 4, Perch7D)
C3SC_REG_FQ1(C3SC_MODEL_PERCH7D, 16, 6, 4, Perch7D)
} // namespace c3sc
