// Kernels of the stochastic Kubo double-moment path (compute_moments_stochastic, recursion.f90:979-1234) beside the SpMM:
// the Chebyshev three-term combine on whole vectors, and the transposing copies between the engine's vector layout and the
// column-major matrices the moment GEMM (rocBLAS zgemm) reads.
#pragma once
#include <hip/hip_runtime.h>
#include "kernels_mfma.hpp"

namespace rsrec {

// ham_vec_matmul's epilogue + the caller's recurrence (recursion.f90:968-970, :1132-1136), element-wise on nd doubles:
//   FIRST: out = (t - b cur) / a                    (T_1 = x)
//   else : out = 2 ((t - b cur) / a) - old          (T_{m} = 2 x T_{m-1} - T_{m-2})
template <bool FIRST>
__global__ void k_cheb_combine(size_t nd, const double* __restrict__ t, const double* __restrict__ cur, const double* __restrict__ old,
                               double* __restrict__ out, double a, double b) {
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < nd; e += (size_t)gridDim.x * blockDim.x) {
        double v = t[e] - b * cur[e];
        v = v / a;
        if (!FIRST) v = 2.0 * v - old[e];
        out[e] = v;
    }
}

// CI vector -> columns of a column-major complex matrix with rows (atom k, orbital row r):  M[k 18 + r, col0 + c] = vec(k)[r][c].
// One workgroup per atom; the 18x18 block is transposed through LDS so that both sides are accessed in runs.
__global__ __launch_bounds__(384) void k_vec_to_cols(int kk, const double2* __restrict__ vec /*[kk][18 r][18 c]*/, double2* __restrict__ M, size_t ld, int col0) {
    __shared__ double2 blk[BLK];
    for (int k = blockIdx.x; k < kk; k += gridDim.x) {
        if (threadIdx.x < BLK) blk[threadIdx.x] = vec[(size_t)k * BLK + threadIdx.x];      // [r][c], c fastest
        __syncthreads();
        if (threadIdx.x < BLK) {
            const int r = threadIdx.x % NB, c = threadIdx.x / NB;                          // r fastest on the way out
            M[(size_t)(col0 + c) * ld + (size_t)k * NB + r] = blk[r * NB + c];
        }
        __syncthreads();
    }
}

// reference layout (column-major interleaved 18x18 blocks, as the Fortran arrays psi(18,18,kk)) <-> CI (row-major): a block transpose
template <bool TO_CI>
__global__ __launch_bounds__(384) void k_block_transpose(int kk, const double2* __restrict__ src, double2* __restrict__ dst) {
    __shared__ double2 blk[BLK];
    for (int k = blockIdx.x; k < kk; k += gridDim.x) {
        if (threadIdx.x < BLK) blk[threadIdx.x] = src[(size_t)k * BLK + threadIdx.x];
        __syncthreads();
        if (threadIdx.x < BLK) { const int i = threadIdx.x % NB, j = threadIdx.x / NB; dst[(size_t)k * BLK + threadIdx.x] = blk[i * NB + j]; }
        __syncthreads();
    }
}

}  // namespace rsrec
