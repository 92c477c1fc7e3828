// registry.hpp -- table of compiled kernel instantiations (model, dim, padded rank, nodes per lane).
#pragma once
#include <hip/hip_runtime.h>

#include <mutex>
#include <vector>

#include "kernel_common.hpp"

namespace c3sc {

struct LaunchIO {
    const double *ro;
    const int32_t *idx;
    double *out; // bellman: [F][N]; stencil: [F][N][2d+1]
    int32_t *uidx;
    int32_t *absorbed;
    const int32_t *nbf; // stencil only: explicit fixed-dim neighbours [F][2(d-1)] or null
    const int32_t *nbv; // stencil only: explicit varying-dim neighbours [F][N][2] or null
    const double *tbl;   // TableModel only: [F][N][U][2d+1] host-evaluated callbacks
    const double *tcost; // TableModel only: [F][N][2] (boundcost, obscost)
    hipStream_t stream;
};

typedef hipError_t (*launch_fn)(const KArgs &A, const LaunchIO &io);

struct KernelEntry {
    int model;   // C3SC_MODEL_* or 0 for the stencil-only kernels
    int d;       // state dimension
    int rp;      // padded rank
    int npl;     // nodes per lane (fiber-per-wave) ; 0 = any
    int variant; // C3SC_VARIANT_*
    int max_n;   // largest N_k this instantiation handles
    int k;       // dim_vary this instantiation is compiled for, -1 = any
    launch_fn fn;
    const char *name;
};

std::vector<KernelEntry> &kernel_registry();

// What a launcher has to find out once per (kernel, device, dynamic-LDS size): the opt-in to more than 64 KiB of dynamic
// LDS (hipFuncSetAttribute applies to the CURRENT device only), the resident workgroups per CU and the CU count.  One
// instance per kernel instantiation (a function-local static of the launcher), slots per device, guarded by a mutex: the
// C API allows contexts on several devices and calls from several threads.
struct LaunchCache {
    static constexpr int MAXDEV = 16;
    std::mutex mu;
    size_t attr_shmem[MAXDEV] = {0}, occ_shmem[MAXDEV] = {0};
    int blocks_per_cu[MAXDEV] = {0}, num_cu[MAXDEV] = {0};
    hipError_t prepare(const void *kern, int threads, size_t shmem, int &blocks, int &cus)
    {
        int dev = 0;
        hipError_t e = hipGetDevice(&dev);
        if (e != hipSuccess) return e;
        if (dev < 0 || dev >= MAXDEV) return hipErrorInvalidDevice;
        std::lock_guard<std::mutex> lk(mu);
        if (shmem > attr_shmem[dev]) {
            e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
            if (e != hipSuccess) return e;
            attr_shmem[dev] = shmem;
        }
        if (blocks_per_cu[dev] == 0 || shmem != occ_shmem[dev]) {
            int nb = 0;
            e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kern, threads, shmem);
            if (e != hipSuccess) return e;
            blocks_per_cu[dev] = nb > 0 ? nb : 1;
            occ_shmem[dev] = shmem;
        }
        if (num_cu[dev] == 0) {
            hipDeviceProp_t prop;
            e = hipGetDeviceProperties(&prop, dev);
            if (e != hipSuccess) return e;
            num_cu[dev] = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
        }
        blocks = blocks_per_cu[dev];
        cus = num_cu[dev];
        return hipSuccess;
    }
};

struct Registrar {
    explicit Registrar(const KernelEntry &e) { kernel_registry().push_back(e); }
};

} // namespace c3sc
