// Kernels of the stochastic Kubo double-moment path (compute_moments_stochastic, recursion.f90:979-1234) beside the SpMM:
// the Chebyshev three-term combine on whole vectors, and the moment contraction L^H R on the FP64 matrix cores (k_kubo_gram: the
// vectors are read where they lie; rounds 2-3 packed them into column-major copies for rocBLAS zgemm).
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

// ---- the moment contraction of compute_moments_stochastic (recursion.f90:1177-1234) -----------------------------------------------
//   mu(c, c', n, m) = sum_k [L_m(k)]^H [R_n(k)]   (18x18 blocks; L_m = T_{m-1}(H~) r, R_n = v_a T_{n-1}(H~) v_b r)
// In the CI layout a whole vector IS a dense row-major complex matrix: row (k, r) = 18 k + r at 36 (18 k + r) doubles, 18 complex columns.
// With the vectors of a chunk side by side (vector stride `ls` / `rs`) the contraction is the complex GEMM
//   C[i][j] = sum_rho conj(L[rho][i]) R[rho][j],   i = 18 m + c,  j = 18 n + c',  rho = 18 k + r  (K = 18 kk: 144 000 for 8 000 atoms)
// read IN PLACE: no column-major copies (round 3 packed both operands for rocBLAS zgemm: 21 GB of copies per vector at cond_ll = 500).
// One wave = a 48 x 48 block of C (3 x 3 tiles of v_mfma_f64_16x16x4) over one slice of rho; a complex MAC is four real MFMAs
// (Cre += Lr Rr + Li Ri, Cim += Lr Ri - Li Rr), the operands of a k-step are 16-byte (re, im) loads: lane (l15, l4) reads element
// (rho0 + l4, i0 + l15) -- sixteen consecutive complex numbers per row segment -- and they are requested one k-step ahead.
// Work split: the C blocks x KS slices of rho; XCD x owns the slices x, x + 8, ... so the waves of an XCD sweep the C blocks of ONE slice
// together and every vector element crosses the fabric once per slice owner (L2-resident panels: 57 KB per k-step at cond_ll = 50).
// The slices' partial blocks are summed in slice order by k_kubo_gram_reduce (fixed order: run-to-run reproducible), which also writes
// the reference's layout mu_nm(c, c', n, m).
constexpr int KG_T = 3;                       // tiles of 16 per wave, both ways
constexpr int KG_BLK = 16 * KG_T;             // 48

__global__ __launch_bounds__(256, 2) void k_kubo_gram(const double* __restrict__ L, size_t ls, int mrows, const double* __restrict__ R, size_t rs, int ncols,
                                                      int ksteps_total, int ksplit, double2* __restrict__ part, int nbm, int nbn) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int l15 = lane & 15, l4 = lane >> 4;
    const int ntask_mn = nbm * nbn;
    const int xcd = blockIdx.x & 7, t = (int)(blockIdx.x >> 3) * 4 + wave;
    const int per_xcd = ntask_mn * (ksplit >> 3);
    if (t >= per_xcd) return;
    const int ks = xcd + 8 * (t / ntask_mn), mn = t % ntask_mn;
    const int bm = mn / nbn, bn = mn - bm * nbn;
    const int per = (ksteps_total + ksplit - 1) / ksplit;
    const int s0 = ks * per, s1 = min(ksteps_total, s0 + per);
    // per-lane row / column of the three A and three B tiles (clamped: the padding rows of the last block repeat a valid one, never stored)
    const double* pa[KG_T];
    const double* pb[KG_T];
#pragma unroll
    for (int q = 0; q < KG_T; ++q) {
        const int i = min(bm * KG_BLK + 16 * q + l15, mrows - 1), j = min(bn * KG_BLK + 16 * q + l15, ncols - 1);
        pa[q] = L + (size_t)(i / 18) * ls + 2 * (i % 18) + (size_t)36 * (4 * (size_t)s0 + l4);
        pb[q] = R + (size_t)(j / 18) * rs + 2 * (j % 18) + (size_t)36 * (4 * (size_t)s0 + l4);
    }
    double4_t cre[KG_T][KG_T], cim[KG_T][KG_T];
#pragma unroll
    for (int q = 0; q < KG_T; ++q)
#pragma unroll
        for (int u = 0; u < KG_T; ++u) { cre[q][u] = (double4_t){0, 0, 0, 0}; cim[q][u] = (double4_t){0, 0, 0, 0}; }
    typedef double kg_d2 __attribute__((ext_vector_type(2)));
    kg_d2 a0[KG_T], b0[KG_T], a1[KG_T], b1[KG_T];          // two operand sets, used alternately (no register copies in the loop)
    auto fetch = [&](kg_d2 (&a)[KG_T], kg_d2 (&b)[KG_T]) {
#pragma unroll
        for (int q = 0; q < KG_T; ++q) { a[q] = *reinterpret_cast<const kg_d2*>(pa[q]); b[q] = *reinterpret_cast<const kg_d2*>(pb[q]); pa[q] += 144; pb[q] += 144; }
    };
    auto mac = [&](const kg_d2 (&a)[KG_T], const kg_d2 (&b)[KG_T]) {
#pragma unroll
        for (int q = 0; q < KG_T; ++q) {
            const double ar = a[q][0], ai = a[q][1], nai = -ai;
#pragma unroll
            for (int u = 0; u < KG_T; ++u) {
                cre[q][u] = __builtin_amdgcn_mfma_f64_16x16x4f64(ar, b[u][0], cre[q][u], 0, 0, 0);
                cim[q][u] = __builtin_amdgcn_mfma_f64_16x16x4f64(ar, b[u][1], cim[q][u], 0, 0, 0);
                cre[q][u] = __builtin_amdgcn_mfma_f64_16x16x4f64(ai, b[u][1], cre[q][u], 0, 0, 0);
                cim[q][u] = __builtin_amdgcn_mfma_f64_16x16x4f64(nai, b[u][0], cim[q][u], 0, 0, 0);
            }
        }
    };
    int s = s0;
    if (s < s1) fetch(a0, b0);
#pragma unroll 1
    for (; s + 2 <= s1; s += 2) {
        fetch(a1, b1);
        mac(a0, b0);
        if (s + 2 < s1) fetch(a0, b0);
        mac(a1, b1);
    }
    if (s < s1) mac(a0, b0);
    // D register rr of lane (l15, l4): row l4 + 4 rr, column l15 of the tile.  Partial blocks: part[ks][i][j], row-major over the PADDED
    // block grid (nbm x 48 rows, nbn x 48 columns): sixteen consecutive complex numbers per lane row
    const size_t ldp = (size_t)nbn * KG_BLK;
    double2* P = part + (size_t)ks * ((size_t)nbm * KG_BLK) * ldp;
#pragma unroll
    for (int q = 0; q < KG_T; ++q)
#pragma unroll
        for (int u = 0; u < KG_T; ++u)
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const int i = bm * KG_BLK + 16 * q + l4 + 4 * rr, j = bn * KG_BLK + 16 * u + l15;
                P[(size_t)i * ldp + j] = make_double2(cre[q][u][rr], cim[q][u][rr]);
            }
}

// (Measured and not kept, round 4: the same contraction on v_mfma_f64_4x4x4_4b -- A operand = 4 rows replicated over the instruction's
// four blocks, 32 x 48 of C per wave, 8 A + 3 B loads per 96 instructions.  The shape sustains 70-75 TFLOP/s in isolation
// (profiles/ubench_f64_r01.txt) but the kernel reached 50.0 / 55.7 TFLOP/s at cond_ll = 50 / 500 against 59.1 / 64.1 for this one.)
// sum of the slices' partial blocks in slice order -> mu_nm(c, c', n0 + n, m0 + m) in the reference's index order
// (mu_nm_stochastic(18,18,cond_ll,cond_ll,vec), recursion.f90:1204-1228); one thread per (i, j)
__global__ __launch_bounds__(256) void k_kubo_gram_reduce(const double2* __restrict__ part, int ksplit, int prow /*padded rows of a slice*/, int pcol, int mrows, int ncols,
                                                         double2* __restrict__ mu /*this vector's (18,18,cond_ll,cond_ll)*/, int cond_ll, int m0, int n0) {
    const size_t ldp = (size_t)pcol, slice = (size_t)prow * ldp;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < (size_t)mrows * ncols; e += (size_t)gridDim.x * blockDim.x) {
        const int i = (int)(e / ncols), j = (int)(e - (size_t)i * ncols);
        double sr = 0.0, si = 0.0;
        for (int ks = 0; ks < ksplit; ++ks) { const double2 v = part[(size_t)ks * slice + (size_t)i * ldp + j]; sr += v.x; si += v.y; }
        const int m = i / 18, c = i - 18 * m, n = j / 18, cp = j - 18 * n;
        mu[(size_t)c + 18 * ((size_t)cp + 18 * ((size_t)(n0 + n) + (size_t)cond_ll * (m0 + m)))] = make_double2(sr, si);
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
