// launch_fq.hpp -- host launcher + registration macro for the fiber-quad kernels (kernel_fiber_quad.hpp).
#pragma once
#include "kernel_fiber_quad.hpp"
#include "registry.hpp"

namespace c3sc {

template <class Model, int RP, int K, int NWV, bool ONEPASS>
hipError_t launch_fq_impl(const KArgs &A, const LaunchIO &io)
{
    constexpr int D = Model::D;
    if (A.cmode == 1) return hipErrorNotSupported; // candidate lists only
    if (K > 0 && K < D - 1 && A.quad_aop_off[K] == 0) return hipErrorNotSupported;
    // LDS: the largest staged fixed core, then the per-wave node values, then the candidate table
    if (A.ncand > 64) return hipErrorNotSupported; // see launch_fpp.hpp
    size_t doubles = 0;
    for (int m = 0; m < D; m++) {
        if (m == K) continue;
        const size_t need = (size_t)A.ngrid[m] * quad_stride((m == 0 || m == D - 1) ? RP : RP * RP);
        if (need > doubles) doubles = need;
    }
    doubles = (doubles + 1) & ~(size_t)1;
    KArgs B = A;
    B.quad_sv_off = (int)doubles;
    doubles += (size_t)NWV * (ONEPASS ? 18 : A.N) * 16; // per-wave node values: a ring with one pass, all N with two
    B.tbl_off = (int)doubles;
    doubles += (size_t)CandLds<Model>::doubles(A.ncand);
    const size_t shmem = doubles * sizeof(double);
    if (shmem > 160u * 1024u) return hipErrorOutOfMemory;
    auto kern = k_fiber_quad<Model, RP, K, NWV, ONEPASS>;
    static LaunchCache cache;
    int blocks_per_cu = 1, num_cu = 256;
    hipError_t e = cache.prepare((const void *)kern, 64 * NWV, shmem, blocks_per_cu, num_cu);
    if (e != hipSuccess) return e;
    const long per_tile = 16L * NWV, ntiles = (A.F + per_tile - 1) / per_tile;
    const long cap = (long)num_cu * blocks_per_cu;
    int grid = (int)(ntiles < cap ? ntiles : cap);
    if (grid < 1) grid = 1;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * NWV), shmem, io.stream, B, io.ro, io.idx, io.out, io.uidx, io.absorbed);
    return hipGetLastError();
}

#ifndef FQ_ONEPASS
#define FQ_ONEPASS 1 // node loop in one pass with a one-round delay (0: node values in a pass of their own)
#endif
template <class Model, int RP, int K, int NWV>
hipError_t launch_fq(const KArgs &A, const LaunchIO &io)
{
    // one pass keeps a round's stencil (2d+1 doubles) in registers for one more round: at rank 16 x d = 10 the kernel is already
    // over its register budget and the second set of products is cheaper than the spills (0.82 vs 0.87 ms on quad10d)
    constexpr bool ONEPASS = (FQ_ONEPASS != 0) && !(RP >= 16 && Model::D >= 8);
    return launch_fq_impl<Model, RP, K, NWV, ONEPASS>(A, io);
}

// two wavefronts per 16 fibers (k_fiber_quad_duo): NWV / 2 pairs per workgroup
template <class Model, int RP, int K, int NWV, bool DBUF>
hipError_t launch_fq_duo(const KArgs &A, const LaunchIO &io)
{
    constexpr int D = Model::D;
    if (A.cmode == 1) return hipErrorNotSupported; // candidate lists only
    if (K > 0 && K < D - 1 && A.quad_aop_off[K] == 0) return hipErrorNotSupported;
    if (A.ncand > 64) return hipErrorNotSupported; // see launch_fpp.hpp
    size_t doubles = 0;
    for (int m = 0; m < D; m++) {
        if (m == K) continue;
        const size_t need = (size_t)A.ngrid[m] * quad_stride((m == 0 || m == D - 1) ? RP : RP * RP);
        if (need > doubles) doubles = need;
    }
    doubles = (doubles + 1) & ~(size_t)1;
    KArgs B = A;
    B.quad_m1_off = (int)doubles; // two staging buffers: level l+1 is copied while level l is applied
    if (DBUF) doubles *= 2;
    B.quad_sv_off = (int)doubles;
    doubles += (size_t)(NWV / 2) * A.N * 16; // node values of each pair's 16 fibers
    B.quad_ix_off = (int)doubles;
    doubles += (size_t)(NWV / 2) * 2 * D * 8; // fiber indices [16][D] ints per pair, this tile's and the next one's
    B.quad_x_off = (int)doubles;
    doubles += (size_t)(NWV / 2) * 2 * D * 64; // half stencils between the wavefronts of a pair
    B.tbl_off = (int)doubles;
    doubles += (size_t)CandLds<Model>::doubles(A.ncand);
    const size_t shmem = doubles * sizeof(double);
    if (shmem > 160u * 1024u) return hipErrorOutOfMemory;
    auto kern = k_fiber_quad_duo<Model, RP, K, NWV, DBUF>;
    static LaunchCache cache;
    int blocks_per_cu = 1, num_cu = 256;
    hipError_t e = cache.prepare((const void *)kern, 64 * NWV, shmem, blocks_per_cu, num_cu);
    if (e != hipSuccess) return e;
    const long per_tile = 16L * (NWV / 2), ntiles = (A.F + per_tile - 1) / per_tile;
    const long cap = (long)num_cu * blocks_per_cu;
    int grid = (int)(ntiles < cap ? ntiles : cap);
    if (grid < 1) grid = 1;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * NWV), shmem, io.stream, B, io.ro, io.idx, io.out, io.uidx, io.absorbed);
    return hipGetLastError();
}

#ifndef FQD_DBUF
#define FQD_DBUF true
#endif
#define C3SC_REG_FQD(MODEL_ID, RP, K, NWV, ...)                                                                                \
    static Registrar C3SC_CAT(reg_fqd_, __COUNTER__)(KernelEntry{                                                              \
        MODEL_ID, __VA_ARGS__::D, RP, 0, C3SC_VARIANT_FIBER_QUAD, 128, K, &launch_fq_duo<__VA_ARGS__, RP, K, NWV, FQD_DBUF>,  \
        "k_fiber_quad_duo<" #__VA_ARGS__ "," #RP ",K=" #K ">"});

#define C3SC_REG_FQ1(MODEL_ID, RP, K, NWV, ...)                                                                  \
    static Registrar C3SC_CAT(reg_fq_, __COUNTER__)(KernelEntry{                                                 \
        MODEL_ID, __VA_ARGS__::D, RP, 0, C3SC_VARIANT_FIBER_QUAD, 128, K, &launch_fq<__VA_ARGS__, RP, K, NWV>,  \
        "k_fiber_quad<" #__VA_ARGS__ "," #RP ",K=" #K ">"});

} // namespace c3sc
