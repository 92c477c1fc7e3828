// launch_fpp.hpp -- host launcher + registration macro for the fiber-pair (rank-split) kernels.
#pragma once
#include "kernel_fiber_pair.hpp"
#include "registry.hpp"

namespace c3sc {

template <class Model, int RP, int K, bool FORCED>
hipError_t launch_fpp_impl(const KArgs &A, const LaunchIO &io)
{
    constexpr int D = Model::D;
    constexpr int NV = 2 * (D - 1), NP = NV + 1, RH = RP / 2;
    if (A.ncand > 64) return hipErrorNotSupported; // one lane per candidate fills the table: the per-wave kernel walks longer lists
    // LDS: max(largest staged fixed core, half-swap buffer, per-node exchange buffers)
    size_t doubles = (size_t)NV * RH * 64;
    const size_t exch = (size_t)(2 * RP + (NP + 1) * 2 + 2 * NP) * 64; // L, R rows + exchange rows
    if (exch > doubles) doubles = exch;
    for (int m = 0; m < D; m++) {
        if (m == K || fpp_direct<Model, RP>()) continue; // no staged core when the fold reads the cores directly
        const int elems = (m == 0 || m == D - 1) ? RP : RP * RP;
        const size_t need = ((size_t)A.ngrid[m] * fpl_lds_stride(elems) + 1) & ~(size_t)1; // whole 16-byte LDS-DMA pieces
        if (need > doubles) doubles = need;
    }
    // wave-uniform tables (candidates, nodes of dim K) behind everything else: they persist across tiles
    KArgs B = A;
    B.tbl_off = (int)doubles;
    doubles += (size_t)CandLds<Model>::doubles(A.ncand) + (size_t)NodeLds<Model, K>::doubles(A.N);
    const size_t shmem = doubles * sizeof(double);
    auto kern = k_fiber_pair<Model, RP, K, FORCED>;
    static LaunchCache cache;
    int blocks_per_cu = 1, num_cu = 256;
    hipError_t e = cache.prepare((const void *)kern, FPP_THREADS, shmem, blocks_per_cu, num_cu);
    if (e != hipSuccess) return e;
    const long ntiles = (A.F + 63) / 64;
    const long cap = (long)num_cu * blocks_per_cu;
    // Persistent workgroups (one per resident slot, striding over the tiles) amortise the per-workgroup set-up, but a
    // static split leaves the slots that drew the slow tiles running at the end.  Once there are many tiles per slot the
    // hardware dispatcher balances better with one workgroup per tile (car7d, 2^20 fibers = 16 tiles per slot: 1.84 vs
    // 1.92 ms; at 2 tiles per slot it is the other way round, 0.290 vs 0.270 ms, and at 4 they are equal).  In between is worse
    // still: 2 / 4 / 8 tiles per workgroup at 2^20 fibers take 1.85 / 1.92 / 2.05 ms against 1.82 (the last round of a coarser
    // grid leaves slots idle).
    int grid = (int)(ntiles < cap ? ntiles : cap);
    if (ntiles >= 8 * cap && ntiles < 0x7fffffffL) grid = (int)ntiles;
    if (grid < 1) grid = 1;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(FPP_THREADS), shmem, io.stream, B, io.ro, io.idx, io.out, io.uidx, io.absorbed);
    return hipGetLastError();
}

template <class Model, int RP, int K>
hipError_t launch_fpp(const KArgs &A, const LaunchIO &io)
{
    return A.forced ? launch_fpp_impl<Model, RP, K, true>(A, io) : launch_fpp_impl<Model, RP, K, false>(A, io);
}

#define C3SC_REG_FPP1(MODEL_ID, RP, K, ...)                                                                   \
    static Registrar C3SC_CAT(reg_fpp_, __COUNTER__)(KernelEntry{                                             \
        MODEL_ID, __VA_ARGS__::D, RP, 0, C3SC_VARIANT_FIBER_PAIR, 128, K, &launch_fpp<__VA_ARGS__, RP, K>, \
        "k_fiber_pair<" #__VA_ARGS__ "," #RP ",K=" #K ">"});

} // namespace c3sc
