// launch_fpw.hpp -- host launcher + registration macro for the fiber-per-wave kernels.
#pragma once
#include "kernel_fiber_per_wave.hpp"
#include "registry.hpp"

namespace c3sc {

template <class Model, int RP, int NPL, bool STENCIL, bool BOX, bool STAGED>
hipError_t launch_fpw_impl(const KArgs &A, const LaunchIO &io)
{
    constexpr int D = Model::D;
    constexpr int WS = 4 * RP + 2 * D * RP + 64 * NPL;
    const bool kedge = (A.k == 0) || (A.k == D - 1);
    size_t doubles = (size_t)(4 * WS + (STAGED ? A.N * kcore_stride(RP, kedge) : 0));
    KArgs B = A;
    B.tbl_off = (int)doubles; // candidate table behind everything else
    if constexpr (!STENCIL && !Model::IS_TABLE) doubles += (size_t)CandLds<Model>::doubles(A.ncand);
    const size_t shmem = doubles * sizeof(double);
    auto kern = k_fiber_per_wave<Model, RP, NPL, STENCIL, BOX, STAGED>;
    static LaunchCache cache;
    int blocks_per_cu = 1, num_cu = 256;
    hipError_t e = cache.prepare((const void *)kern, 256, shmem, blocks_per_cu, num_cu);
    if (e != hipSuccess) return e;
    long want = (A.F + 3) / 4;
    long cap = (long)num_cu * blocks_per_cu;
    int grid = (int)(want < cap ? want : cap);
    if (grid < 1) grid = 1;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), shmem, io.stream, B, io.ro, io.idx, io.out, io.uidx, io.absorbed,
                       io.nbf, io.nbv, io.tbl, io.tcost);
    return hipGetLastError();
}

template <class Model, int RP, int NPL, bool STENCIL, bool BOX = false>
hipError_t launch_fpw(const KArgs &A, const LaunchIO &io)
{
    if (A.cmode == 1 && !BOX) return hipErrorNotSupported; // this entry has no box-minimiser instantiation
    constexpr int D = Model::D;
    constexpr int WS = 4 * RP + 2 * D * RP + 64 * NPL;
    const bool kedge = (A.k == 0) || (A.k == D - 1);
    size_t staged = (size_t)(4 * WS + A.N * kcore_stride(RP, kedge)) * sizeof(double);
    if constexpr (!STENCIL && !Model::IS_TABLE) staged += (size_t)CandLds<Model>::doubles(A.ncand) * sizeof(double);
    // the varying core (N x RP^2 doubles) is staged in LDS when it fits the CU, otherwise read from L2
    if (staged <= 160u * 1024u) return launch_fpw_impl<Model, RP, NPL, STENCIL, BOX, true>(A, io);
    if constexpr (RP >= 12) return launch_fpw_impl<Model, RP, NPL, STENCIL, BOX, false>(A, io);
    else return hipErrorOutOfMemory; // small ranks always fit for N <= 128
}

#define C3SC_CAT2(a, b) a##b
#define C3SC_CAT(a, b) C3SC_CAT2(a, b)

// MODEL may contain commas/angle brackets, hence the variadic tail
#define C3SC_REG_FPW(MODEL_ID, RP, NPL, ...)                                                                 \
    static Registrar C3SC_CAT(reg_fpw_, __COUNTER__)(KernelEntry{                                            \
        MODEL_ID, __VA_ARGS__::D, RP, NPL, C3SC_VARIANT_FIBER_PER_WAVE, 64 * NPL, -1,                          \
        &launch_fpw<__VA_ARGS__, RP, NPL, false>, "k_fiber_per_wave<" #__VA_ARGS__ "," #RP "," #NPL ">"});

// the same entry serving both the candidate-list kernel and the box minimiser (continuous controls)
template <class Model, int RP, int NPL>
hipError_t launch_fpw_both(const KArgs &A, const LaunchIO &io)
{
    return A.cmode == 1 ? launch_fpw<Model, RP, NPL, false, true>(A, io) : launch_fpw<Model, RP, NPL, false, false>(A, io);
}
#define C3SC_REG_FPW_BOX(MODEL_ID, RP, NPL, ...)                                                             \
    static Registrar C3SC_CAT(reg_fpwb_, __COUNTER__)(KernelEntry{                                           \
        MODEL_ID, __VA_ARGS__::D, RP, NPL, C3SC_VARIANT_FIBER_PER_WAVE, 64 * NPL, -1,                          \
        &launch_fpw_both<__VA_ARGS__, RP, NPL>, "k_fiber_per_wave<" #__VA_ARGS__ "," #RP "," #NPL ">"});

#define C3SC_REG_STENCIL(DIM, RP, NPL)                                                                       \
    static Registrar C3SC_CAT(reg_st_, __COUNTER__)(KernelEntry{                                             \
        0, DIM, RP, NPL, C3SC_VARIANT_FIBER_PER_WAVE, 64 * NPL, -1, &launch_fpw<NoModel<DIM>, RP, NPL, true>,    \
        "k_fiber_per_wave<stencil," #DIM "," #RP "," #NPL ">"});

} // namespace c3sc
