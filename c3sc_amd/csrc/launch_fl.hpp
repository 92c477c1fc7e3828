// launch_fl.hpp -- host launcher + registration macro for the fiber-per-lane (low-rank) kernels.
#pragma once
#include <cstdlib>

#include "kernel_fiber_lane.hpp"
#include "registry.hpp"

namespace c3sc {

template <class Model, int RP, int K, int WPS, bool FORCED>
hipError_t launch_fl_impl(const KArgs &A, const LaunchIO &io)
{
    if (A.ncand > 64 || A.N > 128) return hipErrorNotSupported; // one lane per candidate / two nodes per lane fill the tables
    KArgs B = A;
    B.tbl_off = 0;
    const size_t shmem = ((size_t)CandLds<Model>::doubles(A.ncand) + (size_t)NodeLds<Model, K>::doubles(A.N)) * sizeof(double);
    auto kern = k_fiber_lane<Model, RP, K, WPS, FORCED>;
    static LaunchCache cache;
    int blocks_per_cu = 1, num_cu = 256;
    hipError_t e = cache.prepare((const void *)kern, 64, shmem, blocks_per_cu, num_cu);
    if (e != hipSuccess) return e;
    const long ntiles = (A.F + 63) / 64;
    const long cap = (long)num_cu * blocks_per_cu;
    // one wavefront per tile once the tiles outnumber the resident slots: the per-workgroup set-up is two small tables, and
    // the dispatcher balances what a static stride cannot (C3SC_FL_PERSIST=1 keeps the persistent grid, for measurements)
    static const bool persist = getenv("C3SC_FL_PERSIST") != nullptr;
    int grid = (int)(ntiles < cap ? ntiles : cap);
    if (!persist && ntiles > cap && ntiles < 0x7fffffffL) grid = (int)ntiles;
    if (grid < 1) grid = 1;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64), shmem, io.stream, B, io.ro, io.idx, io.out, io.uidx, io.absorbed);
    return hipGetLastError();
}

template <class Model, int RP, int K, int WPS>
hipError_t launch_fl(const KArgs &A, const LaunchIO &io)
{
    return A.forced ? launch_fl_impl<Model, RP, K, WPS, true>(A, io) : launch_fl_impl<Model, RP, K, WPS, false>(A, io);
}

#ifndef C3SC_CAT
#define C3SC_CAT2(a, b) a##b
#define C3SC_CAT(a, b) C3SC_CAT2(a, b)
#endif
#define C3SC_REG_FL1(MODEL_ID, RP, K, WPS, ...)                                                                   \
    static Registrar C3SC_CAT(reg_fl_, __COUNTER__)(KernelEntry{                                                  \
        MODEL_ID, __VA_ARGS__::D, RP, 0, C3SC_VARIANT_FIBER_PER_LANE, 128, K, &launch_fl<__VA_ARGS__, RP, K, WPS>, \
        "k_fiber_lane<" #__VA_ARGS__ "," #RP ",K=" #K ">"});

} // namespace c3sc
