// launch_fpl.hpp -- host launcher + registration macro for the fiber-per-lane kernels.
#pragma once
#include <utility>

#include "kernel_fiber_per_lane.hpp"
#include "registry.hpp"

namespace c3sc {

template <class Model, int RP, int K>
hipError_t launch_fpl(const KArgs &A, const LaunchIO &io)
{
    constexpr int D = Model::D;
    // LDS: the largest staged fixed core (dims != K) with the padded node stride
    size_t doubles = 16;
    for (int m = 0; m < D; m++) {
        if (m == K) continue;
        const int elems = (m == 0 || m == D - 1) ? RP : RP * RP;
        const size_t need = (size_t)A.ngrid[m] * fpl_lds_stride(elems);
        if (need > doubles) doubles = need;
    }
    constexpr int NLDS = fpl_nlds<D, RP>();
    constexpr int RPe = RP + (RP & 1);
    const size_t shmem = (doubles + (size_t)NLDS * RPe * FPL_THREADS) * sizeof(double);
    auto kern = k_fiber_per_lane<Model, RP, K>;
    static int blocks_per_cu = 0;
    static size_t attr_shmem = 0, occ_shmem = (size_t)-1;
    hipError_t e;
    if (shmem > attr_shmem) {
        e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
        if (e != hipSuccess) return e;
        attr_shmem = shmem;
    }
    if (shmem != occ_shmem) {
        int nb = 0;
        e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kern, FPL_THREADS, shmem);
        if (e != hipSuccess) return e;
        blocks_per_cu = nb > 0 ? nb : 1;
        occ_shmem = shmem;
    }
    const long ntiles = (A.F + FPL_THREADS - 1) / FPL_THREADS;
    const long cap = 256L * blocks_per_cu;
    int grid = (int)(ntiles < cap ? ntiles : cap);
    if (grid < 1) grid = 1;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(FPL_THREADS), shmem, io.stream, A, io.ro, io.idx, io.out, io.uidx, io.absorbed);
    return hipGetLastError();
}

#define C3SC_REG_FPL1(MODEL_ID, RP, K, ...)                                                                   \
    static Registrar C3SC_CAT(reg_fpl_, __COUNTER__)(KernelEntry{                                             \
        MODEL_ID, __VA_ARGS__::D, RP, 0, C3SC_VARIANT_FIBER_PER_LANE, 128, K, &launch_fpl<__VA_ARGS__, RP, K>, \
        "k_fiber_per_lane<" #__VA_ARGS__ "," #RP ",K=" #K ">"});

} // namespace c3sc
