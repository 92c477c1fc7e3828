// registry.hpp -- table of compiled kernel instantiations (model, dim, padded rank, nodes per lane).
#pragma once
#include <hip/hip_runtime.h>

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

struct Registrar {
    explicit Registrar(const KernelEntry &e) { kernel_registry().push_back(e); }
};

} // namespace c3sc
