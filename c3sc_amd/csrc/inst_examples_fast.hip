// pair / quad kernels for the three examples that so far only had the fiber-per-wave kernel (0.10-0.14 of the FP64 roofline at
// their own sizes, 2^17 / 2^20 fibers): examples/cothrust2 (6-D, rank 10) on the fiber-pair kernel like car7d: 0.135 -> 0.42 / 0.49;
// examples/skidding5d (5-D, rank 15 -> 16) on the fiber-quad kernel (f64 MFMA for the shared-core products), two wavefronts per
// SIMD: 0.107 -> 0.27 / 0.28 (one wavefront per SIMD for the spilling dimensions measured equal); examples/perching (7-D, rank
// 15 -> 16, 41 candidates, a square root per candidate) on the quad kernel with ONE wavefront per SIMD and the whole register file:
// 0.10 -> 0.23 (two per SIMD spill ~290 VGPRs: 0.15)
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
