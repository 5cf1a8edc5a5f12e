// FP64 matrix-core kernels for gfx950: v_mfma_f64_16x16x4_f64 (full 16x16 tiles) + v_mfma_f64_4x4x4_4b_f64 (the 4-row
// remainder of the 36-row real form), measured in profiles/ubench_f64_r01.txt.
//
// Block sparse H|psi> (hop_b, reference recursion.f90:1560-1625) as real GEMMs.  A complex 18x18 block product
// out = H * x is the real 36x36 * 36xN product
//        [ out_re ]   [ Hr  -Hi ] [ x_re ]
//        [ out_im ] = [ Hi   Hr ] [ x_im ]            (row index kappa = 18*part + r)
// For one neighbour slot the SAME 36x36 operator multiplies the psi blocks of all atoms of a type, so 8 atoms of one
// type form one wave's GEMM:  M = 36 (= 16 + 16 + 4, no padding waste thanks to the 4x4x4 instruction),
// K = 36 (nine k-steps of 4), N = 8 atoms x 18 columns = 144 = 9 tiles of 16:
//   tiles 0..7 : columns 0..15 of atom 0..7 (atom-aligned), tile 8 : columns 16,17 of the eight atoms.
// Work vectors use LayoutRM (kernels_valu.hpp): a B fragment (4 k-rows x 16 columns) is four 128-byte row segments.
// Operator blocks are pre-swizzled on the host into A-fragment order (one coalesced 512-byte load per fragment).
#pragma once
#include <hip/hip_runtime.h>
#include <string>
#include <vector>
#include "kernels_valu.hpp"

namespace rsrec {

typedef double double4_t __attribute__((ext_vector_type(4)));

constexpr int GROUP = 8;            // atoms per wave (one type)
constexpr int MF_WAVES = 4;         // waves per workgroup

// CI layout ("complex interleaved", the large-launch path's ONE vector layout): element (r, c) of an 18x18 block = one complex
// number at doubles 36 r + 2 c (re), + 1 (im): row-major, the transpose of the reference's column-major block.  A row is 36
// reals in memory order [re0 im0 re1 im1 ...]; the Gram / right-multiply kernels below work on such rows generically (they are
// layout-blind: only the coefficient tables and the Gram -> complex conversion know whether real column j means
// (part j / 18, c = j % 18) -- LayoutRM, the small-launch path -- or (part j & 1, c = j >> 1) -- CI).  k_spmm5 reads AND
// writes CI with 16-byte accesses, so a vector exists once (round 1 kept a second "k-pair" copy for the SpMM).
struct LayoutCI {
    static __device__ __forceinline__ double2 ld(const double* b, int r, int c) { return make_double2(b[36 * r + 2 * c], b[36 * r + 2 * c + 1]); }
    static __device__ __forceinline__ void st(double* b, int r, int c, double2 v) { b[36 * r + 2 * c] = v.x; b[36 * r + 2 * c + 1] = v.y; }
};
// real column of (part, c) inside a 36-real row
__host__ __device__ constexpr int gram_col(int part, int c, int ci) { return ci ? 2 * c + part : 18 * part + c; }

// XCD-aware work split: workgroups are dealt round-robin over the 8 XCDs (blockIdx % 8 shares an XCD, guide T1), and each
// XCD has its own 4 MiB L2.  Every psi block is read by ~15 neighbouring atoms, so groups that are close in the region order
// must run on the SAME XCD for those re-reads to hit L2: XCD x owns the contiguous chunk x of the group list and its resident
// workgroups sweep that chunk as a sliding window.  Placement only affects speed, never results.
// Workgroups that take part in a (chain, level) pass: a function of that chain's group count only, so that the work
// assignment and the order of the partial sums do not depend on which chains share a launch (batch-size reproducibility);
// the launch itself is sized by the largest chain of the batch and surplus workgroups leave at once.
__device__ __forceinline__ int active_workgroups(int ngroups) { return max(1, min((int)gridDim.x, (ngroups + MF_WAVES - 1) / MF_WAVES)); }

struct GroupWalk {
    int g, end, step;
    __device__ __forceinline__ GroupWalk(int ngroups, int wave, int nbx_eff = 0) {
        const int nbx = nbx_eff > 0 ? nbx_eff : gridDim.x, bx = blockIdx.x;
        const int xcd = bx & 7, j = bx >> 3;
        const int per_xcd = (nbx >> 3) + ((xcd < (nbx & 7)) ? 1 : 0);       // workgroups that share this XCD label
        const int chunk = (ngroups + 7) >> 3;
        const int lo = xcd * chunk;
        end = min(ngroups, lo + chunk);
        g = lo + j * MF_WAVES + wave;
        step = per_xcd * MF_WAVES;
        if (nbx < 8) { g = bx * MF_WAVES + wave; end = ngroups; step = nbx * MF_WAVES; }
    }
};

// ======================================================================================================================
// Post-hop kernels on the matrix cores.  In LayoutRM an atom block IS the real 18x36 matrix [X_re | X_im] (row-major), and
//   * right-multiplication by a global complex 18x18 matrix G (psi*A_n, pmn*B^-1, psi*B; crecal_b :1927,:1966-1967) is
//       [Y_re | Y_im] = [X_re | X_im] * Ghat,   Ghat = [[Gr, Gi], [-Gi, Gr]]   (36x36 real),
//     i.e. a (18*natoms) x 36 x 36 GEMM: M = stacked rows (8 atoms = 144 rows = 9 tiles), K = 36, N = 36 = 16 + 16 + 4;
//   * the 18x18 complex reductions  sum_i X_i^H Y_i  (A_n :1642, B^2 :1931) are the real 36x36 Gram matrices
//       Gm[k'][k] = sum_rows Xhat[row][k'] * Yhat[row][k],   C_re = Gm[re,re] + Gm[im,im],  C_im = Gm[re,im] - Gm[im,re],
//     with the stacked rows as the MFMA K dimension.  A 16x16x4 D register (rows l4+4j, column l15) is directly the operand
//     fragment of k-step j, so freshly computed rows feed the Gram MFMAs without leaving registers.
// ======================================================================================================================

struct RowRef { unsigned off; bool valid; };
__device__ __forceinline__ RowRef group_row(const int* __restrict__ grp, int rho, int zero_block) {
    const int slot = rho / 18, r = rho - 18 * slot;
    const int a = grp[slot];
    RowRef R;
    R.valid = a >= 0;
    R.off = (unsigned)BLD * (unsigned)(R.valid ? a : zero_block) + 36u * (unsigned)r;
    return R;
}

// canonical 36x36 (row-major) image of a wave's Gram accumulators, written to LDS and summed over the waves of a workgroup
struct GramAcc {
    double4_t t00, t01, t10, t11;   // 16x16 tiles (row tile, column tile)
    double tr0, tr1;                // rows 32..35 x columns of tile 0 / 1   (4x4x4, blocks over N)
    double t0r, t1r;                // rows of tile 0 / 1 x columns 32..35   (4x4x4, blocks over M)
    double trr;                     // rows 32..35 x columns 32..35
    __device__ __forceinline__ void zero() { t00 = t01 = t10 = t11 = (double4_t){0, 0, 0, 0}; tr0 = tr1 = t0r = t1r = trr = 0.0; }
    __device__ __forceinline__ void to_lds(double* G /*[1296]*/, int lane, bool sym) const {
        const int l15 = lane & 15, l4 = lane >> 4, l3 = lane & 3, lg = (lane >> 2) & 3;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int r = l4 + 4 * j;
            G[36 * r + l15] = t00[j];
            G[36 * r + 16 + l15] = t01[j];
            G[36 * (16 + r) + 16 + l15] = t11[j];
            if (!sym) G[36 * (16 + r) + l15] = t10[j];
        }
        G[36 * (32 + l4) + l15] = tr0;
        G[36 * (32 + l4) + 16 + l15] = tr1;
        if (!sym) { G[36 * (4 * lg + l4) + 32 + l3] = t0r; G[36 * (16 + 4 * lg + l4) + 32 + l3] = t1r; }
        if (lg == 0) G[36 * (32 + l4) + 32 + l3] = trr;
    }
};

__device__ __forceinline__ void gram_block_out(const GramAcc& A, double* lds /*[MF_WAVES][1296]*/, double* gout /*[1296]*/, bool sym) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    A.to_lds(lds + wave * 1296, lane, sym);
    __syncthreads();
    for (int e = threadIdx.x; e < 1296; e += blockDim.x) {
        double s = 0.0;
#pragma unroll
        for (int wv = 0; wv < MF_WAVES; ++wv) s += lds[wv * 1296 + e];
        if (sym) {                                   // fill the tiles that were not computed from their transposes
            const int r = e / 36, c = e % 36;
            const bool have = (r >= 32) || (r < 16) || (c >= 16 && c < 32);
            if (!have || (r < 32 && c >= 32)) {
                double t = 0.0;
#pragma unroll
                for (int wv = 0; wv < MF_WAVES; ++wv) t += lds[wv * 1296 + 36 * c + r];
                s = t;
            }
        }
        gout[e] = s;
    }
}

// Streamed once per kernel (the post-hop passes read and write whole vectors of several GB: nothing is reused from a cache):
// RSREC_NT_STREAMS = 1 marks those accesses non-temporal.
#ifndef RSREC_NT_STREAMS
#define RSREC_NT_STREAMS 0
#endif
template <class T> __device__ __forceinline__ T ld_stream(const T* p) {
#if RSREC_NT_STREAMS
    return __builtin_nontemporal_load(p);
#else
    return *p;
#endif
}
__device__ __forceinline__ double2 ld_stream(const double2* p) {
#if RSREC_NT_STREAMS
    typedef double nt_d2 __attribute__((ext_vector_type(2)));
    const nt_d2 v = __builtin_nontemporal_load(reinterpret_cast<const nt_d2*>(p));
    return make_double2(v[0], v[1]);
#else
    return *p;
#endif
}
template <class T> __device__ __forceinline__ void st_stream(T* p, T v) {
#if RSREC_NT_STREAMS
    __builtin_nontemporal_store(v, p);
#else
    *p = v;
#endif
}

#ifndef ADOT_UNROLL
#define ADOT_UNROLL 4
#endif
// ---- A_n partial: Gm = sum_rows psihat^T * that   (hop_b :1642) -------------------------------------------------------
__global__ __launch_bounds__(MF_WAVES * 64, 2) void k_mfma_adot(ChainView CV, int level, int zero_block, const double* __restrict__ psi,
                                                               const double* __restrict__ tvec, double* partial /*[chain][nblk][1296]*/) {
    __shared__ double lds[MF_WAVES * 1296];
    const int chain = blockIdx.y;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int ngroups = CV.count_of(chain, level) / GROUP;
    const int* order = CV.order_of(chain, level);
    const size_t vo = (size_t)chain * CV.vstride;
    const double* ps = psi + vo;
    const double* tv = tvec + vo;
    const int l15 = lane & 15, l4 = lane >> 4, l3 = lane & 3;
    GramAcc A;
    A.zero();
    const int nbx = active_workgroups(ngroups);
    for (GroupWalk w((int)blockIdx.x < nbx ? ngroups : 0, wave, nbx); w.g < w.end; w.g += w.step) {
        const int* grp = order + (size_t)w.g * GROUP;
#pragma unroll ADOT_UNROLL
        for (int kq = 0; kq < 36; ++kq) {
            const RowRef rk = group_row(grp, 4 * kq + l4, zero_block);               // k-row of this lane
            const double p0 = ld_stream(ps + rk.off + l15), p1 = ld_stream(ps + rk.off + 16 + l15), pr = ld_stream(ps + rk.off + 32 + l3);
            const double h0 = ld_stream(tv + rk.off + l15), h1 = ld_stream(tv + rk.off + 16 + l15), hr = ld_stream(tv + rk.off + 32 + l3);
            A.t00 = __builtin_amdgcn_mfma_f64_16x16x4f64(p0, h0, A.t00, 0, 0, 0);
            A.t01 = __builtin_amdgcn_mfma_f64_16x16x4f64(p0, h1, A.t01, 0, 0, 0);
            A.t10 = __builtin_amdgcn_mfma_f64_16x16x4f64(p1, h0, A.t10, 0, 0, 0);
            A.t11 = __builtin_amdgcn_mfma_f64_16x16x4f64(p1, h1, A.t11, 0, 0, 0);
            A.tr0 = __builtin_amdgcn_mfma_f64_4x4x4f64(pr, h0, A.tr0, 0, 0, 0);
            A.tr1 = __builtin_amdgcn_mfma_f64_4x4x4f64(pr, h1, A.tr1, 0, 0, 0);
            A.t0r = __builtin_amdgcn_mfma_f64_4x4x4f64(p0, hr, A.t0r, 0, 0, 0);
            A.t1r = __builtin_amdgcn_mfma_f64_4x4x4f64(p1, hr, A.t1r, 0, 0, 0);
            A.trr = __builtin_amdgcn_mfma_f64_4x4x4f64(pr, hr, A.trr, 0, 0, 0);
        }
    }
    gram_block_out(A, lds, partial + ((size_t)chain * gridDim.x + blockIdx.x) * 1296, false);
}

struct SpmmDims {
    int kk, nslots, nmax, nlev, cpo, ostride, level; size_t vstride; const int* obase; int nchains = 0;
    // operators with several classes of atoms (k_spmm5): the list of ALL atoms (the one a chain uses once its region covers the lattice,
    // obase == sat_base) is sorted by class, so every class is one run of groups, the same for every chain.
    //   run_hi > 0 : this launch serves the groups [run_lo, run_hi) of that list only, and only chains that use it (the launches with the
    //                run's operator stream in LDS);
    //   nskip > 0  : this launch leaves the runs skip_lo/hi[] of that list to such launches (chains on their level-major list: everything).
    //   nruns > 0  : ONE persistent launch serves up to four runs: the workgroup rows run_row0[r] .. run_row0[r + 1] - 1 (rows of 8 or 16
    //                workgroups, the kernel's XCD mapping) hold the stream of class run_tau[r] and take the groups [run_glo[r], run_ghi[r]).
    int sat_base = 0, run_lo = 0, run_hi = 0, nskip = 0, skip_lo[4] = {0, 0, 0, 0}, skip_hi[4] = {0, 0, 0, 0};
    int nruns = 0, run_tau[4] = {0, 0, 0, 0}, run_glo[4] = {0, 0, 0, 0}, run_ghi[4] = {0, 0, 0, 0}, run_row0[5] = {0, 0, 0, 0, 0};
    // atoms with their own operator blocks (classes 0 .. nmax - 1) served by the chain-octet launch (k_spmm5<., ., true>: group g = atom g for
    // 8 chains, whatever list a chain is on): skip_pa = 1 makes every other launch pass over their groups
    int skip_pa = 0;
};

// ---- reductions of the 36x36 real partials ----------------------------------------------------------------------------
// returns C[cp + 18 c] (complex) for tid < 324:  C_re = G[re cp][re c] + G[im cp][im c],  C_im = G[re cp][im c] - G[im cp][re c]
__device__ __forceinline__ double2 reduce_gram(const double* __restrict__ partial /*[nblk][1296]*/, int nblk, double* lds /*1296*/, int ci = 0) {
    for (int e = threadIdx.x; e < 1296; e += blockDim.x) {
        double s = 0.0;
        for (int p = 0; p < nblk; ++p) s += partial[(size_t)p * 1296 + e];
        lds[e] = s;
    }
    __syncthreads();
    double2 c = make_double2(0, 0);
    if (threadIdx.x < BLK) {
        const int cp = threadIdx.x % NB, cc = threadIdx.x / NB;
        const int rp = gram_col(0, cp, ci), ip = gram_col(1, cp, ci), rc = gram_col(0, cc, ci), ic = gram_col(1, cc, ci);
        c.x = lds[36 * rp + rc] + lds[36 * ip + ic];
        c.y = lds[36 * rp + ic] - lds[36 * ip + rc];
    }
    __syncthreads();
    return c;
}

}  // namespace rsrec
