// MFMA (v_mfma_f64_16x16x4_f64 + v_mfma_f64_4x4x4_4b_f64) kernels -- placeholder until the tiled kernels land.
#pragma once
#include <hip/hip_runtime.h>
#include "kernels_valu.hpp"

namespace rsrec {

struct MfmaOperator {
    const char* build(int, int, int, int, int, const double*, const double*, const double*, const double*, const double*, const double*) { return nullptr; }
    void release() {}
};

inline void launch_hop_mfma(const MfmaOperator&, const DevProblem& P, const ChainView& CV, const ApplyArgs& G, dim3 grid, hipStream_t stream) {
    k_apply<AM_LANCZOS><<<grid, NTHREADS, 0, stream>>>(P, CV, G);
}

}  // namespace rsrec
