// probe_fold_mfma.hip -- sizes the "fold as a GEMM on the matrix cores" re-blocking of the quad kernels' folding phase (VERDICT r3,
// next 3) before it is built.  The verdict's form -- bucket a tile's fibers by i_m so that a level is (F_g x RP)(RP x RP) -- has
// 2.5 fibers per bucket at N = 25 and 64-fiber tiles; the form probed here needs no sort: the GEMM's N dimension is ONE fiber's own
// folded vectors (up to 2(d-1)+1 = 19 of them at d = 10), which all meet the same matrix G_m[i_m] at level m:
//     [vectors of fiber f at level m+1] (RP x nvec) = G_m[i_m(f)] (RP x RP) . [vectors at level m] (RP x nvec)
// i.e. per fiber and level C = RP/4 v_mfma_f64_16x16x4_f64 with the pre-permuted A operands k_quad_aux already keeps in HBM/L2
// ([N][c|a][MB][C][64 lanes]: one coalesced 512-byte load per K step, no LDS, no bank conflicts, no level-synchronous staging, no
// barrier), D feeding the next level's B directly (D register r of lane (q, t) is row 4r + q = the B row of K step r).
//
// What this probe measures: the MAIN chain only -- d-1 dependent mfma groups per fiber, A operands of level l+1 requested while
// level l multiplies, 16 fibers per wavefront one after the other, 2 wavefronts per SIMD, F = 2^17 fibers as in the bench.  Not
// included (so the number is a LOWER bound of the re-blocked fold): the two new neighbour vectors every level creates (each is a
// matrix-vector product with its OWN matrix G_m[i_m +- 1]: 1/16 of an MFMA's columns, or 2 x RP^2 FMAs on the vector ALU with the
// running prefix broadcast across the wavefront), and the 16 x 16 lane transposition (fiber <-> vector index) the node loop's
// lane = (rank quarter, fiber) layout needs afterwards.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/probe_fold_mfma.hip -o /tmp/probe_fold && /tmp/probe_fold
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

typedef double v4d __attribute__((ext_vector_type(4)));
constexpr int RP = 16, C = RP / 4, D = 10, N = 25, LEVELS = D - 1;
constexpr int NODE = 2 * C * 64; // doubles of one node's A operands: [c | a][C steps][64 lanes]

template <int SEEDS>
__global__ void __launch_bounds__(256, 2) k_fold(const double *__restrict__ aop, const int *__restrict__ idx, double *__restrict__ out, long F)
{
    const int lane = threadIdx.x & 63;
    const long wave = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = ((long)gridDim.x * blockDim.x) >> 6;
    double keep = 0.0;
    for (long f0 = wave * 16; f0 < F; f0 += nwaves * 16) {
        for (int ff = 0; ff < 16; ff++) { // the wavefront's 16 fibers, one after the other
            const long f = f0 + ff < F ? f0 + ff : F - 1;
            double X[C];
#pragma unroll
            for (int s = 0; s < C; s++) X[s] = 1.0 + 1e-3 * (lane + s); // the level-0 vectors (core rows in the real kernel)
            double a[2][C], alo[2][C], ahi[2][C];
            {
                const int i = idx[f * D + 0];
                const double *p = aop + ((size_t)(0 * N + i)) * NODE;
#pragma unroll
                for (int s = 0; s < C; s++) a[0][s] = p[s * 64 + lane];
                if (SEEDS) {
                    const double *pl = aop + ((size_t)(0 * N + (i > 0 ? i - 1 : i))) * NODE, *ph = aop + ((size_t)(0 * N + (i + 1 < N ? i + 1 : i))) * NODE;
#pragma unroll
                    for (int s = 0; s < C; s++) { alo[0][s] = pl[s * 64 + lane]; ahi[0][s] = ph[s * 64 + lane]; }
                }
            }
#pragma unroll
            for (int l = 0; l < LEVELS; l++) {
                if (l + 1 < LEVELS) { // next level's operands in flight during this level's products
                    const int i = idx[f * D + l + 1];
                    const double *p = aop + ((size_t)((l + 1) * N + i)) * NODE + ((l & 1) ? C * 64 : 0);
#pragma unroll
                    for (int s = 0; s < C; s++) a[(l + 1) & 1][s] = p[s * 64 + lane];
                    if (SEEDS) {
                        const double *pl = aop + ((size_t)((l + 1) * N + (i > 0 ? i - 1 : i))) * NODE, *ph = aop + ((size_t)((l + 1) * N + (i + 1 < N ? i + 1 : i))) * NODE;
#pragma unroll
                        for (int s = 0; s < C; s++) { alo[(l + 1) & 1][s] = pl[s * 64 + lane]; ahi[(l + 1) & 1][s] = ph[s * 64 + lane]; }
                    }
                }
                v4d acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int s = 0; s < C; s++) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[l & 1][s], X[s], acc, 0, 0, 0);
                if (SEEDS) { // the simplest form of the two new neighbour vectors: two more MFMA groups whose other 15 columns are wasted
                    v4d lo = {0.0, 0.0, 0.0, 0.0}, hi = {0.0, 0.0, 0.0, 0.0};
                    double B0[C];
#pragma unroll
                    for (int s = 0; s < C; s++) B0[s] = __shfl(X[s], lane & 48); // the running prefix (column 0) into every column
#pragma unroll
                    for (int s = 0; s < C; s++) { lo = __builtin_amdgcn_mfma_f64_16x16x4f64(alo[l & 1][s], B0[s], lo, 0, 0, 0); hi = __builtin_amdgcn_mfma_f64_16x16x4f64(ahi[l & 1][s], B0[s], hi, 0, 0, 0); }
                    const int t = lane & 15;
#pragma unroll
                    for (int s = 0; s < C; s++) acc[s] = (t == 2 * l + 1) ? lo[s] : ((t == 2 * l + 2) ? hi[s] : acc[s]);
                }
#pragma unroll
                for (int s = 0; s < C; s++) X[s] = acc[s] * 0.0625; // keeps the values bounded; one multiply per level
            }
#pragma unroll
            for (int s = 0; s < C; s++) keep += X[s];
        }
    }
    if (keep == 12345.678) out[0] = keep;
}

int main()
{
    const long F = 1 << 17;
    std::vector<double> aop((size_t)D * N * NODE);
    for (size_t i = 0; i < aop.size(); i++) aop[i] = 0.2 + 0.1 * ((i * 2654435761u) % 1000) / 1000.0;
    std::vector<int> idx((size_t)F * D);
    unsigned long long s = 0xF1BE;
    for (auto &v : idx) { s = s * 6364136223846793005ull + 1442695040888963407ull; v = (int)((s >> 33) % N); }
    double *dA, *dO;
    int *dI;
    hipMalloc(&dA, aop.size() * 8); hipMalloc(&dI, idx.size() * 4); hipMalloc(&dO, 64);
    hipMemcpy(dA, aop.data(), aop.size() * 8, hipMemcpyHostToDevice);
    hipMemcpy(dI, idx.data(), idx.size() * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int seeds = 0; seeds < 2; seeds++) {
        for (int rep = 0; rep < 3; rep++) {
            hipEventRecord(e0);
            if (seeds) hipLaunchKernelGGL(k_fold<1>, dim3(512), dim3(256), 0, 0, dA, dI, dO, F);
            else hipLaunchKernelGGL(k_fold<0>, dim3(512), dim3(256), 0, 0, dA, dI, dO, F);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms = 0;
            hipEventElapsedTime(&ms, e0, e1);
            if (rep == 2) {
                const double cyc = ms * 1e-3 * 2.4e9 / (F / 1024.0); // cycles per fiber and SIMD (1024 SIMDs)
                printf("%s: %.4f ms per launch of %ld fibers (%d levels, rank %d): %.0f cycles per fiber and SIMD; MFMA groups per fiber %d\n",
                       seeds ? "main chain + 2 seed products per level on the matrix cores" : "main chain only (lower bound)", ms, F, LEVELS, RP, cyc,
                       LEVELS * (seeds ? 3 : 1));
            }
        }
    }
    printf("for comparison (profiles/r02_quad10d_duo_ablation.txt): the duo kernel's folding products take 0.25 ms of its 0.59 ms launch = ~4500 cycles per fiber and SIMD\n");
    return 0;
}
