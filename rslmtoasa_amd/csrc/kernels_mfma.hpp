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

constexpr int FRAG_PER_SLOT = 9 * 3 * 64;   // doubles: [q][f][lane]

// Host-side builder + device storage of the fragment tables.
struct MfmaOperator {
    double* d_frag = nullptr;       // [2 (h | h*o)][ntau][nslots][9][3][64]
    size_t bytes = 0;
    int ntau = 0, nslots = 0, have_o = 0;

    void release() { if (d_frag) (void)hipFree(d_frag); d_frag = nullptr; bytes = 0; }

    // one complex block (column-major interleaved) -> 9*3*64 doubles in fragment order
    static void swizzle(const double* blk, double* out) {
        auto R = [&](int ko, int ki) -> double {   // real 36x36 form, kappa = 18*part + index
            const int po = ko / 18, ro = ko % 18, pi = ki / 18, ri = ki % 18;
            const double hr = blk[2 * (ro + 18 * ri)], hi = blk[2 * (ro + 18 * ri) + 1];
            if (po == pi) return hr;
            return po == 0 ? -hi : hi;
        };
        for (int q = 0; q < 9; ++q)
            for (int f = 0; f < 3; ++f)
                for (int l = 0; l < 64; ++l) {
                    const int k = 4 * q + (l >> 4);
                    const int m = (f < 2) ? 16 * f + (l & 15) : 32 + (l & 3);   // f = 2: 4x4x4 A operand, replicated over the 4 blocks
                    out[(q * 3 + f) * 64 + l] = R(m, k);
                }
    }

    // st/loc/... are the SAME host arrays the VALU path uploads (slot 0 already carries +lsham when !hoh)
    const char* build(int nslots_lat, int hstride, int ntype, int nmax, int hoh, const double* st, const double* loc, const double* eeo,
                      const double* hallo, const double* /*enim*/, const double* /*lsham*/) {
        ntau = nmax + ntype; nslots = nslots_lat; have_o = hoh ? 1 : 0;
        const size_t per_set = (size_t)ntau * nslots * FRAG_PER_SLOT;
        std::vector<double> host(per_set * (have_o ? 2 : 1), 0.0);
        for (int set = 0; set < (have_o ? 2 : 1); ++set)
            for (int tau = 0; tau < ntau; ++tau)
                for (int s = 0; s < nslots; ++s) {
                    const double* src;
                    if (tau < nmax) src = (set ? hallo : loc) + 2 * (size_t)BLK * (s + (size_t)hstride * tau);
                    else src = (set ? eeo : st) + 2 * (size_t)BLK * (s + (size_t)hstride * (tau - nmax));
                    swizzle(src, host.data() + set * per_set + ((size_t)tau * nslots + s) * FRAG_PER_SLOT);
                }
        const size_t need = host.size() * sizeof(double);
        if (need > bytes) {
            release();
            if (hipMalloc(reinterpret_cast<void**>(&d_frag), need) != hipSuccess) return "hipMalloc of MFMA operator fragments failed";
            bytes = need;
        }
        if (hipMemcpy(d_frag, host.data(), need, hipMemcpyHostToDevice) != hipSuccess) return "upload of MFMA operator fragments failed";
        return nullptr;
    }
    const double* set_ptr(int set) const { return d_frag + (size_t)set * ntau * nslots * FRAG_PER_SLOT; }
};

struct SpmmArgs {
    const double* frag;     // fragment table of the operator set in use
    const double* in;       // vector the neighbour sum runs over
    double* out;            // result vector (same layout)
    int level;
};

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

// out_i = sum_slots H_slot * in_{nbr(i,slot)} for every atom of the (padded, type-homogeneous) order prefix.
// One wave = one group of 8 atoms.  Group entries < 0 are padding.  Operand fragments are software-pipelined one k-step
// ahead in registers (hipcc otherwise waits for every load right before its three MFMAs).
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

// Fragment table of a global 18x18 complex matrix G for right-multiplication: [9 q][3 f][64 lanes] doubles,
//   f = 0,1 : B operand of the 16x16x4 MFMA, columns 16f + l15 ;  f = 2 : B operand of the 4x4x4 MFMA, columns 32 + (l & 3).
__device__ __forceinline__ void emit_rhs_frags(const double2* M /*18x18 column-major, LDS or global*/, double sign, double* out) {
    for (int e = threadIdx.x; e < 27 * 64; e += blockDim.x) {
        const int l = e & 63, qf = e >> 6, q = qf / 3, f = qf % 3;
        const int ki = 4 * q + (l >> 4);
        const int ko = (f < 2) ? 16 * f + (l & 15) : 32 + (l & 3);
        const int pi = ki / 18, ci = ki % 18, po = ko / 18, co = ko % 18;
        const double2 g = M[ci + 18 * co];
        double v = (pi == po) ? g.x : (pi == 0 ? g.y : -g.y);
        out[e] = sign * v;
    }
}

struct RowRef { unsigned off; bool valid; };
__device__ __forceinline__ RowRef group_row(const int* __restrict__ grp, int rho, int zero_block) {
    const int slot = rho / 18, r = rho - 18 * slot;
    const int a = grp[slot];
    RowRef R;
    R.valid = a >= 0;
    R.off = (unsigned)BLD * (unsigned)(R.valid ? a : zero_block) + 36u * (unsigned)r;
    return R;
}

// ---- K3: psi_i <- pmn_i * Binv ; pmn_i <- psi_i * B  (crecal_b :1963-1969) --------------------------------------------
// HALF = true ("three-term" scheme): only psi_next = pmn * Binv is formed and written to `psi` (which then holds the buffer of
// psi_{n-1}, dead by now); pmn_next = psi_n * B is never materialised -- the next level's orthogonalisation subtracts
// psi_{n-1} * B_n on the fly (k_mfma_orth<2>).  Saves one block read and one block write per atom-step.
template <bool HALF>
__global__ __launch_bounds__(MF_WAVES * 64, 2) void k_mfma_update(ChainView CV, int level, int zero_block, double* psi, double* pmn,
                                                                 const double* __restrict__ bfrags /*[chain][3][27*64]: B, Binv, -B*/) {
    const int chain = blockIdx.y;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int ngroups = CV.count_of(chain, level) / GROUP;
    const int* order = CV.order_of(chain, level);
    const size_t vo = (size_t)chain * CV.vstride;
    double* ps = psi + vo;
    double* pm = pmn + vo;
    const int l15 = lane & 15, l4 = lane >> 4, l3 = lane & 3, lg = (lane >> 2) & 3;
    double gB[HALF ? 1 : 27], gBi[27];
    {
        const double* fb = bfrags + (size_t)chain * 3 * 27 * 64 + lane;
#pragma unroll
        for (int e = 0; e < 27; ++e) { if (!HALF) gB[e] = fb[e * 64]; gBi[e] = fb[(27 + e) * 64]; }
    }
    for (GroupWalk w(ngroups, wave); w.g < w.end; w.g += w.step) {
        const int* grp = order + (size_t)w.g * GROUP;
#pragma unroll 1
        for (int mt = 0; mt < 9; ++mt) {
            const RowRef ra = group_row(grp, 16 * mt + l15, zero_block);           // A-operand row of this lane
            double a1[9], a2[HALF ? 1 : 9];
#pragma unroll
            for (int q = 0; q < 9; ++q) { a1[q] = pm[ra.off + 4 * q + l4]; if (!HALF) a2[q] = ps[ra.off + 4 * q + l4]; }
            double4_t y1a = {0, 0, 0, 0}, y1b = {0, 0, 0, 0}, y2a = {0, 0, 0, 0}, y2b = {0, 0, 0, 0};
            double y1r = 0.0, y2r = 0.0;
#pragma unroll
            for (int q = 0; q < 9; ++q) {
                y1a = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[q], gBi[3 * q + 0], y1a, 0, 0, 0);
                y1b = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[q], gBi[3 * q + 1], y1b, 0, 0, 0);
                y1r = __builtin_amdgcn_mfma_f64_4x4x4f64(a1[q], gBi[3 * q + 2], y1r, 0, 0, 0);
                if (!HALF) {
                    y2a = __builtin_amdgcn_mfma_f64_16x16x4f64(a2[q], gB[3 * q + 0], y2a, 0, 0, 0);
                    y2b = __builtin_amdgcn_mfma_f64_16x16x4f64(a2[q], gB[3 * q + 1], y2b, 0, 0, 0);
                    y2r = __builtin_amdgcn_mfma_f64_4x4x4f64(a2[q], gB[3 * q + 2], y2r, 0, 0, 0);
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const RowRef rs = group_row(grp, 16 * mt + l4 + 4 * j, zero_block);
                if (rs.valid) {
                    ps[rs.off + l15] = y1a[j]; ps[rs.off + 16 + l15] = y1b[j];
                    if (!HALF) { pm[rs.off + l15] = y2a[j]; pm[rs.off + 16 + l15] = y2b[j]; }
                }
            }
            const RowRef rr = group_row(grp, 16 * mt + 4 * lg + l4, zero_block);    // 4x4x4 D: row 4g + i, column 32 + j
            if (rr.valid) { ps[rr.off + 32 + l3] = y1r; if (!HALF) pm[rr.off + 32 + l3] = y2r; }
        }
    }
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
#pragma unroll 4
        for (int kq = 0; kq < 36; ++kq) {
            const RowRef rk = group_row(grp, 4 * kq + l4, zero_block);               // k-row of this lane
            const double p0 = ps[rk.off + l15], p1 = ps[rk.off + 16 + l15], pr = ps[rk.off + 32 + l3];
            const double h0 = tv[rk.off + l15], h1 = tv[rk.off + 16 + l15], hr = tv[rk.off + 32 + l3];
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

// ---- K2: pmn_i <- (t_i - pmn_i) - psi_i * A ; Gm += pmnhat^T pmnhat   (hop_b :1641, crecal_b :1922-1934) ----------------
// MODE 0: pmn <- pmn - psi A                      (pmn already holds H psi - pmn_old: VALU/fused epilogues)
// MODE 1: pmn <- (t - pmn) - psi A                (t = H psi from the SpMM kernel)
// MODE 2: pmn <- t - psi_prev B_n - psi A         (three-term scheme: pmn_old = psi_prev B_n is formed here, never stored)
template <int MODE>
__global__ __launch_bounds__(MF_WAVES * 64, 2) void k_mfma_orth(ChainView CV, int level, int zero_block, const double* __restrict__ psi, double* pmn,
                                                               const double* __restrict__ tvec, const double* __restrict__ afrags /*[chain][27*64] = -A*/,
                                                               double* partial /*[chain][nblk][1296]*/, const double* __restrict__ psi_prev = nullptr,
                                                               const double* __restrict__ bfrags = nullptr /*[chain][3][27*64], third = -B_n*/) {
    constexpr bool HAS_T = MODE != 0;
    __shared__ double lds[MF_WAVES * 1296];
    const int chain = blockIdx.y;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int ngroups = CV.count_of(chain, level) / GROUP;
    const int* order = CV.order_of(chain, level);
    const size_t vo = (size_t)chain * CV.vstride;
    const double* ps = psi + vo;
    const double* tv = HAS_T ? tvec + vo : nullptr;
    double* pm = pmn + vo;
    const int l15 = lane & 15, l4 = lane >> 4, l3 = lane & 3, lg = (lane >> 2) & 3;
    const double* pp = (MODE == 2) ? psi_prev + vo : nullptr;
    double nA[27], nB[MODE == 2 ? 27 : 1];
    {
        const double* fa = afrags + (size_t)chain * 27 * 64 + lane;
#pragma unroll
        for (int e = 0; e < 27; ++e) nA[e] = fa[e * 64];
        if (MODE == 2) {
            const double* fb = bfrags + ((size_t)chain * 3 + 2) * 27 * 64 + lane;
#pragma unroll
            for (int e = 0; e < 27; ++e) nB[e] = fb[e * 64];
        }
    }
    GramAcc Gm;
    Gm.zero();
    for (GroupWalk w(ngroups, wave); w.g < w.end; w.g += w.step) {
        const int* grp = order + (size_t)w.g * GROUP;
#pragma unroll 1
        for (int mt = 0; mt < 9; ++mt) {
            const RowRef ra = group_row(grp, 16 * mt + l15, zero_block);
            double a[9], ap[MODE == 2 ? 9 : 1];
#pragma unroll
            for (int q = 0; q < 9; ++q) { a[q] = ps[ra.off + 4 * q + l4]; if (MODE == 2) ap[q] = pp[ra.off + 4 * q + l4]; }
            // C in D layout
            RowRef rs[4];
            double4_t ca, cb;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                rs[j] = group_row(grp, 16 * mt + l4 + 4 * j, zero_block);
                if (MODE == 2) {
                    ca[j] = tv[rs[j].off + l15];
                    cb[j] = tv[rs[j].off + 16 + l15];
                } else if (MODE == 1) {
                    ca[j] = tv[rs[j].off + l15] - pm[rs[j].off + l15];
                    cb[j] = tv[rs[j].off + 16 + l15] - pm[rs[j].off + 16 + l15];
                } else {
                    ca[j] = pm[rs[j].off + l15];
                    cb[j] = pm[rs[j].off + 16 + l15];
                }
            }
            const RowRef rr = group_row(grp, 16 * mt + 4 * lg + l4, zero_block);
            double cr = (MODE == 2) ? tv[rr.off + 32 + l3] : (MODE == 1 ? tv[rr.off + 32 + l3] - pm[rr.off + 32 + l3] : pm[rr.off + 32 + l3]);
            if (MODE == 2) {
#pragma unroll
                for (int q = 0; q < 9; ++q) {       // - psi_prev * B_n first (the reference subtracts pmn_old before psi A)
                    ca = __builtin_amdgcn_mfma_f64_16x16x4f64(ap[q], nB[3 * q + 0], ca, 0, 0, 0);
                    cb = __builtin_amdgcn_mfma_f64_16x16x4f64(ap[q], nB[3 * q + 1], cb, 0, 0, 0);
                    cr = __builtin_amdgcn_mfma_f64_4x4x4f64(ap[q], nB[3 * q + 2], cr, 0, 0, 0);
                }
            }
#pragma unroll
            for (int q = 0; q < 9; ++q) {
                ca = __builtin_amdgcn_mfma_f64_16x16x4f64(a[q], nA[3 * q + 0], ca, 0, 0, 0);
                cb = __builtin_amdgcn_mfma_f64_16x16x4f64(a[q], nA[3 * q + 1], cb, 0, 0, 0);
                cr = __builtin_amdgcn_mfma_f64_4x4x4f64(a[q], nA[3 * q + 2], cr, 0, 0, 0);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (rs[j].valid) { pm[rs[j].off + l15] = ca[j]; pm[rs[j].off + 16 + l15] = cb[j]; }
            if (rr.valid) pm[rr.off + 32 + l3] = cr;
            // Gram update: D register j = rows 4j..4j+3 of this tile = operand fragment of k-step j
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const double f0 = ca[j], f1 = cb[j];
                const double fr = __shfl(cr, l3 + 4 * j + 16 * l4, 64);   // Y[row 4j + l4][32 + l3], replicated over the 4 blocks
                Gm.t00 = __builtin_amdgcn_mfma_f64_16x16x4f64(f0, f0, Gm.t00, 0, 0, 0);
                Gm.t01 = __builtin_amdgcn_mfma_f64_16x16x4f64(f0, f1, Gm.t01, 0, 0, 0);
                Gm.t11 = __builtin_amdgcn_mfma_f64_16x16x4f64(f1, f1, Gm.t11, 0, 0, 0);
                Gm.tr0 = __builtin_amdgcn_mfma_f64_4x4x4f64(fr, f0, Gm.tr0, 0, 0, 0);
                Gm.tr1 = __builtin_amdgcn_mfma_f64_4x4x4f64(fr, f1, Gm.tr1, 0, 0, 0);
                Gm.trr = __builtin_amdgcn_mfma_f64_4x4x4f64(fr, fr, Gm.trr, 0, 0, 0);
            }
        }
    }
    gram_block_out(Gm, lds, partial + ((size_t)chain * gridDim.x + blockIdx.x) * 1296, true);
}

struct SpmmDims { int kk, nslots, nmax, nlev, cpo, ostride, level; size_t vstride; const int* obase; int nchains = 0; };

// All pointers are separate __restrict__ kernel parameters: only `out` is written, so hipcc can prove the index tables
// read-only and fetch them with scalar loads (they then never enter the vmcnt queue the operand prefetch relies on).
// FUSE = false: out_i = sum_slots H_slot in_nbr (store mode).
// FUSE = true : the whole hop_b (recursion.f90:1560-1648): out holds pmn and becomes  H psi - pmn ; the A_n partial
//               sum_i psi_i^H (H psi)_i is formed from the accumulators through a wave-private LDS transpose, so H psi never
//               goes to HBM.  `partial` receives one 36x36 real Gram image per workgroup.
template <int WPS, bool FUSE>
__global__ __launch_bounds__(MF_WAVES * 64, WPS) void k_mfma_spmm(SpmmDims D, const int* __restrict__ order_all, const int* __restrict__ cum,
                                                                 const int* __restrict__ nbr, const int* __restrict__ izp,
                                                                 const double* __restrict__ frag, const double* __restrict__ in_all,
                                                                 double* __restrict__ out_all, double* __restrict__ partial) {
    __shared__ double lds[FUSE ? MF_WAVES * 1296 : 1];
    const int chain = blockIdx.y;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave-uniform: group bookkeeping lives in SGPRs
    const int count = cum[(chain / D.cpo) * D.nlev + D.level];           // multiple of GROUP
    const int ngroups = count / GROUP;
    const int* __restrict__ order = order_all + (size_t)(chain / D.cpo) * D.ostride + D.obase[(chain / D.cpo) * D.nlev + D.level];
    const size_t vo = (size_t)chain * D.vstride;
    const double* __restrict__ in = in_all + vo;
    double* __restrict__ out = out_all + vo;
    struct { int kk, nslots, nmax; } P = {D.kk, D.nslots, D.nmax};
    const int zero_block = P.kk;                              // index of the all-zero block (absent neighbours, padding)
    const int l15 = lane & 15, l4 = lane >> 4, l3 = lane & 3;
    GramAcc Gm;
    Gm.zero();
    // per-lane element offset of k-row (4q + l4) inside a block: kappa = 18*part + r -> 36 r + 18 part
    int koff[9];
#pragma unroll
    for (int q = 0; q < 9; ++q) { const int kap = 4 * q + l4; koff[q] = (kap < 18) ? 36 * kap : 36 * (kap - 18) + 18; }

    for (GroupWalk w(ngroups, wave); w.g < w.end; w.g += w.step) {
        const int* __restrict__ grp = order + (size_t)w.g * GROUP;
        int atom[GROUP];
#pragma unroll
        for (int t = 0; t < GROUP; ++t) atom[t] = grp[t];
        const int first = atom[0];                             // groups are never empty: entry 0 is a real atom
        const int tau = first < P.nmax ? first : P.nmax + izp[first];
        const int my_rem_atom = grp[l15 >> 1];                 // remainder tile: lane -> (atom (l15>>1), column 16 + (l15&1))
        const double* __restrict__ fr = frag + (size_t)tau * P.nslots * FRAG_PER_SLOT + lane;

        auto load_src = [&](int s, unsigned (&src)[9]) {
#pragma unroll
            for (int t = 0; t < GROUP; ++t) {
                int n = atom[t] >= 0 ? nbr[(size_t)P.nslots * atom[t] + s] : -1;
                if (n < 0) n = zero_block;
                src[t] = (unsigned)BLD * n + l15;              // column l15 of the atom-aligned tile
            }
            int n = my_rem_atom >= 0 ? nbr[(size_t)P.nslots * my_rem_atom + s] : -1;
            if (n < 0) n = zero_block;
            src[8] = (unsigned)BLD * n + 16 + (l15 & 1);
        };

        double4_t acc0[9], acc1[9];
        double acc2[9];
#pragma unroll
        for (int t = 0; t < 9; ++t) { acc0[t] = (double4_t){0, 0, 0, 0}; acc1[t] = (double4_t){0, 0, 0, 0}; acc2[t] = 0.0; }

        // Operand ring: three buffers, loads issued TWO k-steps ahead of their MFMAs.  9 k-steps per slot = 3 x 3, so the
        // buffer a k-step uses is the same in every slot and no register copies are needed.
        unsigned src[9], srcn[9];
        double bq[3][9], aq[3][3];
        load_src(0, src);
#pragma unroll
        for (int p = 0; p < 2; ++p) {
#pragma unroll
            for (int t = 0; t < 9; ++t) bq[p][t] = in[src[t] + koff[p]];
#pragma unroll
            for (int f = 0; f < 3; ++f) aq[p][f] = fr[(p * 3 + f) * 64];
        }

        for (int s = 0; s < P.nslots; ++s) {
            const int sn = (s + 1 < P.nslots) ? s + 1 : 0;    // last slot prefetches slot 0 again (discarded): no tail branch
            load_src(sn, srcn);
            const double* fs = fr + (size_t)s * FRAG_PER_SLOT;
            const double* fsn = fr + (size_t)sn * FRAG_PER_SLOT;
#pragma unroll
            for (int q = 0; q < 9; ++q) {
                const int cur = q % 3, nxt = (q + 2) % 3;
                if (q + 2 < 9) {
#pragma unroll
                    for (int t = 0; t < 9; ++t) bq[nxt][t] = in[src[t] + koff[q + 2]];
#pragma unroll
                    for (int f = 0; f < 3; ++f) aq[nxt][f] = fs[((q + 2) * 3 + f) * 64];
                } else {
#pragma unroll
                    for (int t = 0; t < 9; ++t) bq[nxt][t] = in[srcn[t] + koff[q + 2 - 9]];
#pragma unroll
                    for (int f = 0; f < 3; ++f) aq[nxt][f] = fsn[((q + 2 - 9) * 3 + f) * 64];
                }
                __builtin_amdgcn_sched_barrier(0);             // keep the prefetch ABOVE this k-step's MFMAs (hipcc sinks it otherwise)
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    acc0[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(aq[cur][0], bq[cur][t], acc0[t], 0, 0, 0);
                    acc1[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(aq[cur][1], bq[cur][t], acc1[t], 0, 0, 0);
                    acc2[t] = __builtin_amdgcn_mfma_f64_4x4x4f64(aq[cur][2], bq[cur][t], acc2[t], 0, 0, 0);
                }
            }
#pragma unroll
            for (int t = 0; t < 9; ++t) src[t] = srcn[t];
        }
        // epilogue.  D layout: acc0/acc1 row kappa = l4 + 4j (+16), column = l15 ;  acc2: row 32 + l4.
        // kappa = 18*part + r  ->  element offset 36 r + 18 part inside the block.
        int ro0[4], ro1[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            ro0[j] = 36 * (l4 + 4 * j);
            const int k1 = 16 + l4 + 4 * j;
            ro1[j] = (k1 < 18) ? 36 * k1 : 36 * (k1 - 18) + 18;
        }
        const int ro2 = 36 * (14 + l4) + 18;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int a = (t < 8) ? atom[t] : my_rem_atom;
            if (a < 0) continue;
            double* ob = out + (size_t)BLD * a + ((t < 8) ? l15 : 16 + (l15 & 1));
            if (FUSE) {                                        // pmn <- H psi - pmn   (hop_b :1641)
#pragma unroll
                for (int j = 0; j < 4; ++j) { ob[ro0[j]] = acc0[t][j] - ob[ro0[j]]; ob[ro1[j]] = acc1[t][j] - ob[ro1[j]]; }
                ob[ro2] = acc2[t] - ob[ro2];
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) { ob[ro0[j]] = acc0[t][j]; ob[ro1[j]] = acc1[t][j]; }
                ob[ro2] = acc2[t];
            }
        }
        if (FUSE) {
            // A_n partial: two atoms at a time, H psi goes through this wave's LDS slab (2 blocks in LayoutRM) so that the
            // stacked rows (atom, r) become the MFMA K dimension.
            double* hw = lds + wave * 1296;
#pragma unroll
            for (int pr = 0; pr < 4; ++pr) {
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int t = 2 * pr + u;
                    double* hb = hw + u * BLD + l15;
#pragma unroll
                    for (int j = 0; j < 4; ++j) { hb[ro0[j]] = acc0[t][j]; hb[ro1[j]] = acc1[t][j]; }
                    hb[ro2] = acc2[t];
                }
                if ((l15 >> 2) == pr) {                        // remainder tile: lanes of atoms 2pr, 2pr+1
                    double* hb = hw + ((l15 >> 1) & 1) * BLD + 16 + (l15 & 1);
#pragma unroll
                    for (int j = 0; j < 4; ++j) { hb[ro0[j]] = acc0[8][j]; hb[ro1[j]] = acc1[8][j]; }
                    hb[ro2] = acc2[8];
                }
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int kq = 0; kq < 9; ++kq) {
                    const int rho = 4 * kq + l4;               // stacked row: atom (2pr + rho/18), row rho%18
                    const int which = rho >= 18 ? 1 : 0, r = rho - 18 * which;
                    int a = which ? atom[2 * pr + 1] : atom[2 * pr];
                    if (a < 0) a = zero_block;
                    const double* pb = in + (size_t)BLD * a + 36 * r;
                    const double* hb = hw + which * BLD + 36 * r;
                    const double p0 = pb[l15], p1 = pb[16 + l15], pq = pb[32 + l3];
                    const double h0 = hb[l15], h1 = hb[16 + l15], hq = hb[32 + l3];
                    Gm.t00 = __builtin_amdgcn_mfma_f64_16x16x4f64(p0, h0, Gm.t00, 0, 0, 0);
                    Gm.t01 = __builtin_amdgcn_mfma_f64_16x16x4f64(p0, h1, Gm.t01, 0, 0, 0);
                    Gm.t10 = __builtin_amdgcn_mfma_f64_16x16x4f64(p1, h0, Gm.t10, 0, 0, 0);
                    Gm.t11 = __builtin_amdgcn_mfma_f64_16x16x4f64(p1, h1, Gm.t11, 0, 0, 0);
                    Gm.tr0 = __builtin_amdgcn_mfma_f64_4x4x4f64(pq, h0, Gm.tr0, 0, 0, 0);
                    Gm.tr1 = __builtin_amdgcn_mfma_f64_4x4x4f64(pq, h1, Gm.tr1, 0, 0, 0);
                    Gm.t0r = __builtin_amdgcn_mfma_f64_4x4x4f64(p0, hq, Gm.t0r, 0, 0, 0);
                    Gm.t1r = __builtin_amdgcn_mfma_f64_4x4x4f64(p1, hq, Gm.t1r, 0, 0, 0);
                    Gm.trr = __builtin_amdgcn_mfma_f64_4x4x4f64(pq, hq, Gm.trr, 0, 0, 0);
                }
                __builtin_amdgcn_wave_barrier();
            }
        }
    }
    if (FUSE) {
        __syncthreads();
        gram_block_out(Gm, lds, partial + ((size_t)chain * gridDim.x + blockIdx.x) * 1296, false);
    }
}

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

__global__ __launch_bounds__(1024) void k_reduce_a_mf(const double* __restrict__ partial, int nblk, double2* a_out, size_t astride, double* afrags) {
    __shared__ double lds[1296];
    __shared__ double2 Am[BLK];
    const int chain = blockIdx.x;
    const double2 c = reduce_gram(partial + (size_t)chain * nblk * 1296, nblk, lds);
    if (threadIdx.x < BLK) { a_out[chain * astride + threadIdx.x] = c; Am[threadIdx.x] = c; }
    __syncthreads();
    emit_rhs_frags(Am, -1.0, afrags + (size_t)chain * 27 * 64);
}

// complex (VALU-layout) partials -> A_n and its fragment table (used when a VALU epilogue produced the partials)
__global__ __launch_bounds__(1024) void k_reduce_a_c2f(const double2* __restrict__ partial, int nblk, double2* a_out, size_t astride, double* afrags) {
    __shared__ double2 lds[3 * BLK];
    __shared__ double2 Am[BLK];
    const int chain = blockIdx.x;
    const double2 c = reduce_partials(partial + (size_t)chain * nblk * BLK, nblk, 1, 0, lds);
    if (threadIdx.x < BLK) { a_out[chain * astride + threadIdx.x] = c; Am[threadIdx.x] = c; }
    __syncthreads();
    emit_rhs_frags(Am, -1.0, afrags + (size_t)chain * 27 * 64);
}

__global__ __launch_bounds__(1024) void k_reduce_b_eig_mf(const double* __restrict__ partial, int nblk, double2* b2_out, size_t bstride, double2* Bmats,
                                                         double* bfrags, int* status) {
    __shared__ double lds[1296];
    __shared__ Eig18Shared sh;
    __shared__ double2 Bm[BLK];
    const int chain = blockIdx.x;
    const double2 c = reduce_gram(partial + (size_t)chain * nblk * 1296, nblk, lds);
    if (threadIdx.x < BLK) { b2_out[chain * bstride + threadIdx.x] = c; sh.A[threadIdx.x] = c; }
    __syncthreads();
    const int sw = jacobi18(sh);
    if (sw < 0 && threadIdx.x == 0) atomicOr(status, 1);
    if (threadIdx.x < NB) { const double l = sqrt(sh.ev[threadIdx.x]); sh.f1[threadIdx.x] = l; sh.f2[threadIdx.x] = 1.0 / l; }
    __syncthreads();
    double2* Bout = Bmats + (size_t)chain * 2 * BLK;
    for (int which = 0; which < 2; ++which) {
        matfun18(sh, which ? sh.f2 : sh.f1, Bm);                                   // into LDS
        __syncthreads();
        for (int e = threadIdx.x; e < BLK; e += blockDim.x) Bout[which * BLK + e] = Bm[e];
        emit_rhs_frags(Bm, 1.0, bfrags + ((size_t)chain * 3 + which) * 27 * 64);
        if (which == 0) emit_rhs_frags(Bm, -1.0, bfrags + ((size_t)chain * 3 + 2) * 27 * 64);   // -B_{n+1} for the next level (three-term scheme)
        __syncthreads();
    }
}

}  // namespace rsrec
