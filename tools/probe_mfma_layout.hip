// probe_mfma_layout.hip -- checks, on the device, the operand layout of v_mfma_f64_16x16x4_f64 and the lane exchange of
// v_permlane32_swap / v_permlane16_swap that kernel_fiber_quad.hpp relies on.  Build + run on the GPU box:
//   hipcc --offload-arch=gfx950 -O2 tools/probe_mfma_layout.hip -o /tmp/probe && /tmp/probe
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

typedef double v4d __attribute__((ext_vector_type(4)));

__global__ void k_probe(const double *A /*16x4 row-major*/, const double *B /*4x16 row-major*/, double *D /*16x16 row-major*/, double *Draw /*[64][4] as held*/,
                        unsigned *sw32 /*[2][64]*/, unsigned *sw16 /*[2][64]*/)
{
    const int l = threadIdx.x;
    // layout: A[i = l%16][k = l/16], B[k = l/16][n = l%16], D[i = 4*r + l/16][n = l%16]
    const double a = A[(l % 16) * 4 + l / 16], b = B[(l / 16) * 16 + l % 16];
    v4d c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; r++) D[(4 * r + l / 16) * 16 + l % 16] = c[r];
    for (int r = 0; r < 4; r++) Draw[l * 4 + r] = c[r];
    unsigned x = 1000 + l, y = 2000 + l;
    auto r32 = __builtin_amdgcn_permlane32_swap(x, y, false, false);
    sw32[l] = r32[0];
    sw32[64 + l] = r32[1];
    auto r16 = __builtin_amdgcn_permlane16_swap(x, y, false, false);
    sw16[l] = r16[0];
    sw16[64 + l] = r16[1];
}

int main()
{
    std::vector<double> A(64), B(64), D(256, 0.0), ref(256, 0.0);
    for (int i = 0; i < 64; i++) { A[i] = 1.0 + 0.01 * i; B[i] = 2.0 - 0.03 * i; }
    for (int i = 0; i < 16; i++)
        for (int n = 0; n < 16; n++)
            for (int k = 0; k < 4; k++) ref[i * 16 + n] += A[i * 4 + k] * B[k * 16 + n];
    double *dA, *dB, *dD, *dR;
    unsigned *d32, *d16;
    hipMalloc(&dA, 64 * 8); hipMalloc(&dB, 64 * 8); hipMalloc(&dD, 256 * 8); hipMalloc(&dR, 256 * 8); hipMalloc(&d32, 128 * 4); hipMalloc(&d16, 128 * 4);
    hipMemcpy(dA, A.data(), 64 * 8, hipMemcpyHostToDevice);
    hipMemcpy(dB, B.data(), 64 * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_probe, dim3(1), dim3(64), 0, 0, dA, dB, dD, dR, d32, d16);
    hipMemcpy(D.data(), dD, 256 * 8, hipMemcpyDeviceToHost);
    unsigned s32[128], s16[128];
    hipMemcpy(s32, d32, 128 * 4, hipMemcpyDeviceToHost);
    hipMemcpy(s16, d16, 128 * 4, hipMemcpyDeviceToHost);
    std::vector<double> Rw(256);
    hipMemcpy(Rw.data(), dR, 256 * 8, hipMemcpyDeviceToHost);
    // where does each held value belong?  (A / B operand layout as assumed; the products are pairwise distinct)
    printf("D layout found by value: lane reg -> (row i, col n)\n");
    for (int l = 0; l < 64; l += 1) {
        if (!(l < 4 || l % 16 == 0 || l == 17 || l == 63)) continue;
        printf("  lane %2d:", l);
        for (int r = 0; r < 4; r++) {
            int fi = -1, fn = -1;
            for (int i = 0; i < 256; i++) if (std::abs(Rw[l * 4 + r] - ref[i]) < 1e-12) { fi = i / 16; fn = i % 16; }
            printf("  r%d->(%2d,%2d)", r, fi, fn);
        }
        printf("\n");
    }
    double err = 0;
    for (int i = 0; i < 256; i++) err = std::max(err, std::abs(D[i] - ref[i]));
    printf("mfma_f64_16x16x4 layout: max err %.3e -> %s\n", err, err < 1e-12 ? "AS ASSUMED" : "DIFFERENT");
    // expected: swap32(x, y): x.hi <-> y.lo : x' = [x.lo | y.lo], y' = [x.hi | y.hi]
    int ok32 = 1, ok16 = 1;
    for (int l = 0; l < 64; l++) {
        const unsigned ex = l < 32 ? 1000 + l : 2000 + (l - 32), ey = l < 32 ? 1000 + (l + 32) : 2000 + l;
        ok32 &= (s32[l] == ex) && (s32[64 + l] == ey);
        // swap16(x, y): odd rows of x <-> even rows of y: x' row1 = y row0, y' row0 = x row1, x' row3 = y row2, y' row2 = x row3
        const int row = l / 16, t = l % 16;
        const unsigned fx = (row & 1) ? 2000 + (16 * (row - 1) + t) : 1000 + l, fy = (row & 1) ? 2000 + l : 1000 + (16 * (row + 1) + t);
        ok16 &= (s16[l] == fx) && (s16[64 + l] == fy);
    }
    printf("permlane32_swap: %s\npermlane16_swap: %s\n", ok32 ? "AS ASSUMED" : "DIFFERENT", ok16 ? "AS ASSUMED" : "DIFFERENT");
    if (!ok32 || !ok16) {
        for (int l = 0; l < 64; l += 8) printf("lane %2d: sw32 x'=%u y'=%u   sw16 x'=%u y'=%u\n", l, s32[l], s32[64 + l], s16[l], s16[64 + l]);
    }
    return (err < 1e-12 && ok32 && ok16) ? 0 : 1;
}
