// fiber-quad (MFMA) instantiations for the other models: 10-D chain (quad10d, rank 15 -> 16), skidding car (rank 20), 6-D LQG
#include "launch_fpw.hpp"
#include "launch_fq.hpp"
#include "models.hpp"
namespace c3sc {
#define REG10Q(RP, NWV)                                    \
    C3SC_REG_FQ1(C3SC_MODEL_CHAIN, RP, 0, NWV, Chain<10>)  \
    C3SC_REG_FQ1(C3SC_MODEL_CHAIN, RP, 1, NWV, Chain<10>)  \
    C3SC_REG_FQ1(C3SC_MODEL_CHAIN, RP, 2, NWV, Chain<10>)  \
    C3SC_REG_FQ1(C3SC_MODEL_CHAIN, RP, 3, NWV, Chain<10>)  \
    C3SC_REG_FQ1(C3SC_MODEL_CHAIN, RP, 4, NWV, Chain<10>)  \
    C3SC_REG_FQ1(C3SC_MODEL_CHAIN, RP, 5, NWV, Chain<10>)  \
    C3SC_REG_FQ1(C3SC_MODEL_CHAIN, RP, 6, NWV, Chain<10>)  \
    C3SC_REG_FQ1(C3SC_MODEL_CHAIN, RP, 7, NWV, Chain<10>)  \
    C3SC_REG_FQ1(C3SC_MODEL_CHAIN, RP, 8, NWV, Chain<10>)  \
    C3SC_REG_FQ1(C3SC_MODEL_CHAIN, RP, 9, NWV, Chain<10>)
REG10Q(4, 8)
#define REG10QD(RP, NWV)                                   \
    C3SC_REG_FQD(C3SC_MODEL_CHAIN, RP, 0, NWV, Chain<10>)  \
    C3SC_REG_FQD(C3SC_MODEL_CHAIN, RP, 1, NWV, Chain<10>)  \
    C3SC_REG_FQD(C3SC_MODEL_CHAIN, RP, 2, NWV, Chain<10>)  \
    C3SC_REG_FQD(C3SC_MODEL_CHAIN, RP, 3, NWV, Chain<10>)  \
    C3SC_REG_FQD(C3SC_MODEL_CHAIN, RP, 4, NWV, Chain<10>)  \
    C3SC_REG_FQD(C3SC_MODEL_CHAIN, RP, 5, NWV, Chain<10>)  \
    C3SC_REG_FQD(C3SC_MODEL_CHAIN, RP, 6, NWV, Chain<10>)  \
    C3SC_REG_FQD(C3SC_MODEL_CHAIN, RP, 7, NWV, Chain<10>)  \
    C3SC_REG_FQD(C3SC_MODEL_CHAIN, RP, 8, NWV, Chain<10>)  \
    C3SC_REG_FQD(C3SC_MODEL_CHAIN, RP, 9, NWV, Chain<10>)
#ifndef FQ_DUO10
#define FQ_DUO10 1
#endif
#if FQ_DUO10
REG10QD(16, 8) // 20 vectors of 4 doubles per lane do not fit one wavefront: two per 16 fibers, each with half of the neighbour vectors
#endif
REG10Q(16, 4) // one wave per SIMD with the whole 512-entry register file: behind the duo kernel, for grids its two staging buffers do not hold (N > 25)
#define REG4Q(RP, NWV)                                   \
    C3SC_REG_FQ1(C3SC_MODEL_SCAR4D, RP, 0, NWV, Scar4D)  \
    C3SC_REG_FQ1(C3SC_MODEL_SCAR4D, RP, 1, NWV, Scar4D)  \
    C3SC_REG_FQ1(C3SC_MODEL_SCAR4D, RP, 2, NWV, Scar4D)  \
    C3SC_REG_FQ1(C3SC_MODEL_SCAR4D, RP, 3, NWV, Scar4D)
REG4Q(8, 8)
// rank 20: the end-point dimensions fold three levels with five components per lane (one wave per SIMD, 512 registers);
// the middle ones fit two waves per SIMD
// (the two-wavefront form was tried for the end-point dimensions as well -- six wavefronts per workgroup, one staging buffer,
// 1.5 waves per SIMD instead of 1: 3.66 against 3.56 ms per launch of 2^20 fibers, removed)
C3SC_REG_FQ1(C3SC_MODEL_SCAR4D, 20, 0, 4, Scar4D)
C3SC_REG_FQ1(C3SC_MODEL_SCAR4D, 20, 1, 8, Scar4D)
C3SC_REG_FQ1(C3SC_MODEL_SCAR4D, 20, 2, 8, Scar4D)
C3SC_REG_FQ1(C3SC_MODEL_SCAR4D, 20, 3, 4, Scar4D)
#define REG6Q(RP, NWV)                                     \
    C3SC_REG_FQ1(C3SC_MODEL_LQGND, RP, 0, NWV, LqgNd<6>)   \
    C3SC_REG_FQ1(C3SC_MODEL_LQGND, RP, 1, NWV, LqgNd<6>)   \
    C3SC_REG_FQ1(C3SC_MODEL_LQGND, RP, 2, NWV, LqgNd<6>)   \
    C3SC_REG_FQ1(C3SC_MODEL_LQGND, RP, 3, NWV, LqgNd<6>)   \
    C3SC_REG_FQ1(C3SC_MODEL_LQGND, RP, 4, NWV, LqgNd<6>)   \
    C3SC_REG_FQ1(C3SC_MODEL_LQGND, RP, 5, NWV, LqgNd<6>)
REG6Q(8, 8)
} // namespace c3sc
