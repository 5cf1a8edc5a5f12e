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

// ---- chebyshev_orbital_mod (recursion.f90:2834-3049) ------------------------------------------------------------------------------
// left_k = i (Y_k alat (x_s alat t_k) - X_k alat (y_s alat t_k)),  t = H~ psiref of the chain's seed atom s  (:2944-2971: psiref lives on
// atom s alone, so X|r> = alat x_s |r> and the two whole-lattice products of the reference are the same vector t times a number).
// One workgroup column per chain (grid.y); element-wise on the 324 complex entries of every atom block: layout-blind.
__global__ void k_orb_left(int kk, size_t vstride, const int* __restrict__ seed /*[chain]*/, const double* __restrict__ cr /*(3,kk)*/, double alat,
                           double2* __restrict__ vec /*in: t, out: left*/) {
    const int chain = blockIdx.y;
    const int s = seed[chain];
    const double xs = cr[3 * (size_t)s] * alat, ys = cr[3 * (size_t)s + 1] * alat;
    double2* v = vec + (size_t)chain * (vstride / 2);
    const size_t n = (size_t)kk * BLK;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x) {
        const size_t k = e / BLK;
        const double xk = cr[3 * k] * alat, yk = cr[3 * k + 1] * alat;
        const double2 t = v[e];
        const double l1r = yk * (xs * t.x), l1i = yk * (xs * t.y), l2r = xk * (ys * t.x), l2i = xk * (ys * t.y);
        v[e] = make_double2(-(l1i - l2i), l1r - l2r);                  // i (l1 - l2)
    }
}

// 36x36 Gram partials of k_mfma_adot -> the 18x18 complex matrix sum_rows X^H Y of every chain, as it is (no coefficient sandwich)
__global__ __launch_bounds__(1024) void k_reduce_gram_out(const double* __restrict__ partial, int nblk, double2* out, size_t stride, int ci) {
    __shared__ double lds[1296];
    const int chain = blockIdx.x;
    const double2 c = reduce_gram(partial + (size_t)chain * nblk * 1296, nblk, lds, ci);
    if (threadIdx.x < BLK) out[chain * stride + threadIdx.x] = c;
}

}  // namespace rsrec
