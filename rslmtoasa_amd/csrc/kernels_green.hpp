// Green function from block recursion coefficients: green%bgreen (green.f90:1191-1339), driven per site by block_green (:588-621).
//
//   g(E) = B_1^H [ E - A_1 - B_2^H [ E - A_2 - ... ]^-1 B_2 ]^-1 B_1 ,   closed by the square-root terminator of (a_inf, b_inf)
//
// Every (site, energy) pair is an independent chain of lld-1 levels: Q <- (E + eta) 1 - A_l - Q ; Q <- Q^-1 ; Q <- B_l^H Q B_l.
// One WAVE owns one pair: its three 18x18 complex work matrices live in the wave's slice of LDS, the 64 lanes stride over the
// 324 elements, and every phase is wave-synchronous (no workgroup barriers in the level loop).  The inverse is an in-place
// Gauss-Jordan elimination with partial pivoting; the pivot rule is LAPACK's (first row of maximal |re| + |im|, izamax), so the
// pivots are those of the reference's zgetrf unless two candidates differ by rounding only.  A_l, B_l of a site (lld x 10 KB)
// are shared by all its energies and stay L2-resident.
// Quirks of the reference kept: eta is added to the diagonal only where E /= 0 (:1296-1300); the clean-up test
// `abs(..) .lt. 10**(-12)` is integer arithmetic (= 0) and never fires (:1301-1303); Dfac = 1, Cshi = 0.
#pragma once
#include <hip/hip_runtime.h>
#include "kernels_valu.hpp"

namespace rsrec {

#ifndef GREEN_WAVES_N
#define GREEN_WAVES_N 4
#endif
constexpr int GREEN_WAVES = GREEN_WAVES_N;                       // (site, energy) pairs per workgroup

// LDS slice of one wave.  The running matrix itself lives in REGISTERS: lane (ig, jg) = (lane / 6, lane % 6), lanes 0..53, owns
// the 2 x 3 block rows {2 ig, 2 ig + 1} x columns {3 jg .. 3 jg + 2}.  LDS carries what has to cross lanes: the pivot column and
// one or two rows per elimination step, and the operands of the two products (each lane reads 2 rows / 3 columns per k: 5
// 16-byte loads for 6 complex FMAs).  The all-in-LDS formulation this replaces kept the LDS pipe 72 % busy (tools/pmc_green.sh).
struct GreenLds {
    double2 M[BLK];        // staging of a whole matrix for the products (Q^-1, then X), column-major
    double2 B[BLK];        // B_l
    double2 col[NB], rowk[NB], rowp[NB];
    int piv[NB + 2];
};

__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ __forceinline__ double2 gmul(double2 a, double2 b) { return make_double2(fma(a.x, b.x, -(a.y * b.y)), fma(a.x, b.y, a.y * b.x)); }
// s + a b,  s - a b,  s + conj(a) b  as explicit FMA chains (4 instructions each)
__device__ __forceinline__ double2 cfma(double2 a, double2 b, double2 s) { return make_double2(fma(-a.y, b.y, fma(a.x, b.x, s.x)), fma(a.y, b.x, fma(a.x, b.y, s.y))); }
__device__ __forceinline__ double2 cfms(double2 a, double2 b, double2 s) { return make_double2(fma(a.y, b.y, fma(-a.x, b.x, s.x)), fma(-a.y, b.x, fma(-a.x, b.y, s.y))); }
__device__ __forceinline__ double2 cfmah(double2 a, double2 b, double2 s) { return make_double2(fma(a.y, b.y, fma(a.x, b.x, s.x)), fma(-a.y, b.x, fma(a.x, b.y, s.y))); }
__device__ __forceinline__ double readlane_d(double v, int src) {
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffll), src), hi = __builtin_amdgcn_readlane((int)(b >> 32), src);
    return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}
__device__ __forceinline__ double2 sel(bool c, double2 a, double2 b) { return make_double2(c ? a.x : b.x, c ? a.y : b.y); }

// row-of-16 shift of a double on the DPP crossbar (no LDS traffic): lane i takes lane i - n of its row, lanes 0..n-1 keep their own
template <int CTRL>
__device__ __forceinline__ double dpp_d(double v) {
    const long long b = __double_as_longlong(v);
    int lo = (int)(b & 0xffffffffll), hi = (int)(b >> 32);
    lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xf, 0xf, false);
    return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}

template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
    const int b = __float_as_int(v);
    return __int_as_float(__builtin_amdgcn_update_dpp(b, b, CTRL, 0xf, 0xf, false));
}

// One Gauss-Jordan step (column K, a compile-time constant: every register index below is static) of the in-place inverse with
// partial pivoting.  Row interchange k <-> p, then  M'[i][j] = x[i][j] - f_i r_j  with  r = (row k with M[k][k] := 1) / pivot,
// f = column k with f_k := -1, x = M with row k and column k zeroed.  Row K sits in register row K & 1 of the lanes ig == K >> 1,
// column K in register column K % 3 of the lanes jg == K / 3: only those registers carry a select.
template <int K>
__device__ __forceinline__ void gauss_jordan_step(double2 (&q)[2][3], GreenLds& L, int lane, int ig, int jg, bool act) {
    constexpr int KR = K & 1, KIG = K >> 1, KC = K % 3, KJG = K / 3;
    // the lane predicates of a step are two compares; keep the compiler from hoisting 18 steps' worth of masks out of the level loop
    int igp = ig, jgp = jg;
    asm volatile("" : "+v"(igp), "+v"(jgp));
    const bool in_row = igp == KIG, in_col = jgp == KJG;
    if (act && in_col) { L.col[2 * ig] = q[0][KC]; L.col[2 * ig + 1] = q[1][KC]; }
    if (act && in_row) { L.rowk[3 * jg] = q[KR][0]; L.rowk[3 * jg + 1] = q[KR][1]; L.rowk[3 * jg + 2] = q[KR][2]; }
    wave_sync();
    // pivot: first row i >= K of maximal |re| + |im| (izamax): maximum by a DPP scan of the two 16-lane rows that hold the 18
    // candidates, then the first lane that attains it
    double2 cx = make_double2(0.0, 0.0);
    double v = -1.0;
    if (lane < NB) { cx = L.col[lane]; if (lane >= K) v = fabs(cx.x) + fabs(cx.y); }
    // single-precision keys decide unless two candidates round to the same float; then the exact comparison does.  The keys are
    // non-negative floats (or -1 for the lanes that hold no candidate), so their order is the order of their bit patterns as signed
    // integers: the scan is four v_max_i32 with a DPP row shift each (the float version cost a canonicalising v_max plus two moves per
    // stage, 20 vector instructions instead of 4)
    const float kf = (float)v;
    int mi = __float_as_int(kf);
    asm volatile("s_nop 1\n\tv_max_i32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
                 "v_max_i32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
                 "v_max_i32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
                 "v_max_i32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\ts_nop 1" : "+v"(mi));
    const int vmaxi = max(__builtin_amdgcn_readlane(mi, 15), __builtin_amdgcn_readlane(mi, 31));
    unsigned long long hit = __ballot(__float_as_int(kf) == vmaxi);
    if (__builtin_popcountll(hit) != 1) {                        // wave-uniform, rare
        double m = fmax(v, dpp_d<0x111>(v));
        m = fmax(m, dpp_d<0x112>(m));
        m = fmax(m, dpp_d<0x114>(m));
        m = fmax(m, dpp_d<0x118>(m));
        const double vmax = fmax(readlane_d(m, 15), readlane_d(m, 31));
        hit = __ballot(v == vmax);
    }
    const int p = hit ? (int)__builtin_ctzll(hit) : K;
    if (lane == 0) L.piv[K] = p;
    const double2 pv = make_double2(readlane_d(cx.x, p), readlane_d(cx.y, p));      // pivot = M[p][K]
    const double den = pv.x * pv.x + pv.y * pv.y;
    double rden = __builtin_amdgcn_rcp(den);                     // two Newton steps on the hardware estimate: full double precision
    rden = fma(rden, fma(-den, rden, 1.0), rden);
    rden = fma(rden, fma(-den, rden, 1.0), rden);
    const double2 ip = make_double2(pv.x * rden, -pv.y * rden);
    double2* prow = L.rowk;
    if (p != K) {                                                // wave-uniform: rows K and p change places
        const int pig = p >> 1;
        const bool odd = p & 1, mine = act && igp == pig;
        if (mine) {
#pragma unroll
            for (int c = 0; c < 3; ++c) L.rowp[3 * jg + c] = sel(odd, q[1][c], q[0][c]);
        }
        if (lane == 0) L.col[p] = make_double2(readlane_d(cx.x, K), readlane_d(cx.y, K));   // interchanged column K: row p carries the old M[K][K]
        wave_sync();
        if (mine) {                                              // row p takes the old row K (row K itself is rebuilt below)
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const double2 x = L.rowk[3 * jg + c];
                q[0][c] = sel(!odd, x, q[0][c]);
                q[1][c] = sel(odd, x, q[1][c]);
            }
        }
        prow = L.rowp;
    }
    // M'[i][j] = x[i][j] - f_i r_j with r = (pivot row, its entry K := 1) / pivot, f = column K with f_K := -1, x = M with row K and column K
    // zeroed.  The two substitutions are made ONCE, in LDS, by one lane (a wave's LDS operations complete in order), and the zeroing is a
    // multiplication by a 0 / 1 lane mask: one instruction per double where a select is two (round 4; the step was 114 vector instructions
    // for 24 useful multiply-adds, tools/pmc_green.sh)
    if (lane == 0) { prow[K] = make_double2(1.0, 0.0); L.col[K] = make_double2(-1.0, 0.0); }
    wave_sync();
    const double mrow = in_row ? 0.0 : 1.0, mcol = in_col ? 0.0 : 1.0, mboth = (in_row || in_col) ? 0.0 : 1.0;
    double2 r[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) r[c] = gmul(prow[3 * jg + c], ip);
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
        const double2 f = L.col[2 * ig + rr];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            double2 x = q[rr][c];
            if (rr == KR && c == KC) x = make_double2(x.x * mboth, x.y * mboth);
            else if (rr == KR) x = make_double2(x.x * mrow, x.y * mrow);
            else if (c == KC) x = make_double2(x.x * mcol, x.y * mcol);
            q[rr][c] = cfms(f, r[c], x);
        }
    }
}
template <int K>
__device__ __forceinline__ void gauss_jordan_from(double2 (&q)[2][3], GreenLds& L, int lane, int ig, int jg, bool act) {
    if constexpr (K < NB) {
        gauss_jordan_step<K>(q, L, lane, ig, jg, act);
        gauss_jordan_from<K + 1>(q, L, lane, ig, jg, act);
    }
}
// undo the row interchanges of the elimination as column interchanges, last first
template <int K>
__device__ __forceinline__ void unpermute_from(double2 (&q)[2][3], GreenLds& L, int ig, int jg, bool act) {
    if constexpr (K >= 0) {
        constexpr int KC = K % 3, KJG = K / 3;
        const int p = __builtin_amdgcn_readfirstlane(L.piv[K]);
        if (p != K) {                                            // wave-uniform
            const int pjg = p / 3, pc = p - 3 * pjg;
            asm volatile("" : "+v"(jg));
            if (act && jg == KJG) { L.rowk[2 * ig] = q[0][KC]; L.rowk[2 * ig + 1] = q[1][KC]; }
            if (act && jg == pjg) {
#pragma unroll
                for (int rr = 0; rr < 2; ++rr) L.rowp[2 * ig + rr] = sel(pc == 0, q[rr][0], sel(pc == 1, q[rr][1], q[rr][2]));
            }
            wave_sync();
            if (act && jg == KJG) { q[0][KC] = L.rowp[2 * ig]; q[1][KC] = L.rowp[2 * ig + 1]; }
            if (act && jg == pjg) {
#pragma unroll
                for (int rr = 0; rr < 2; ++rr) {
                    const double2 x = L.rowk[2 * ig + rr];
                    q[rr][0] = sel(pc == 0, x, q[rr][0]); q[rr][1] = sel(pc == 1, x, q[rr][1]); q[rr][2] = sel(pc == 2, x, q[rr][2]);
                }
            }
            wave_sync();
        }
        unpermute_from<K - 1>(q, L, ig, jg, act);
    }
}

// A_l block of this lane (fetched one level ahead: shared by all energies of the site, L2-resident)
__device__ __forceinline__ void green_fetch(double2 (&an)[2][3], const double2* __restrict__ A, int ig, int jg) {
#pragma unroll
    for (int rr = 0; rr < 2; ++rr)
#pragma unroll
        for (int c = 0; c < 3; ++c) an[rr][c] = A[(2 * ig + rr) + NB * (3 * jg + c)];
}
// B_l straight into the wave's LDS slice (global_load_lds_dwordx4: lane i of load m writes element 64 m + i; no registers are held while
// the elimination of the level runs -- the register copy this replaces was the kernel's 112 bytes of scratch per lane).  The products that
// read L.B wait for it with vmcnt(0).
__device__ __forceinline__ void green_stage_b(GreenLds& L, const double2* __restrict__ Bl, int lane) {
#pragma unroll
    for (int m = 0; m < 6; ++m) {
        const int el = lane + 64 * m;
        if (el < BLK)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(Bl + el), (__attribute__((address_space(3))) void*)(L.B + 64 * m), 16, 0, 0);
    }
}

// grid = (ceil(nen / GREEN_WAVES), nsites).  a_b, b_sqrt: [site][lld][324] complex; a_inf, b_inf: [site][324] real; g0: [site][nen][324]
#ifndef GREEN_WAVES_PER_SIMD
#define GREEN_WAVES_PER_SIMD 3
#endif
// LDOS = true: only Im g0(j,j) leaves the kernel, gim[site][nen][18] (the LDOS stage needs nothing else: bands.f90:258-268), 144 B
// per (site, energy) instead of 5184 B.
template <bool LDOS>
__global__ __launch_bounds__(GREEN_WAVES * 64, GREEN_WAVES_PER_SIMD) void k_block_green(int lld, int nen, const double* __restrict__ ene, double eta_re, double eta_im, int sym_term,
                                                                 const double* __restrict__ a_inf, const double* __restrict__ b_inf,
                                                                 const double2* __restrict__ a_b, const double2* __restrict__ b_sqrt, double2* __restrict__ g0,
                                                                 double* __restrict__ gim = nullptr) {
    __shared__ GreenLds lds[GREEN_WAVES];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ei = blockIdx.x * GREEN_WAVES + wave, site = blockIdx.y;
    if (ei >= nen) return;                                       // wave-uniform; no workgroup barriers below
    GreenLds& L = lds[wave];
    const bool act = lane < 54;                                  // lanes 54..63 shadow lane 53 and never store
    const int ig = act ? lane / 6 : 8, jg = act ? lane % 6 : 5;
    const double e = ene[ei];
    const double* ai = a_inf + (size_t)site * BLK;
    const double* bi = b_inf + (size_t)site * BLK;
    const double a_diag = 0.5 * (ai[0] + ai[9 + NB * 9]), b_diag = 0.5 * (bi[0] + bi[9 + NB * 9]);
    double2 q[2][3];
    // terminator (:1263-1289)
#pragma unroll
    for (int rr = 0; rr < 2; ++rr)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const int i = 2 * ig + rr, j = 3 * jg + c;
            double2 x = make_double2(0.0, 0.0);
            if (i == j) {
                double etop, ebot, aref;
                if (sym_term) { etop = a_diag + 2.0 * b_diag; ebot = a_diag - 2.0 * b_diag; aref = a_diag; }
                else {
                    const double av = ai[i + NB * i], bv = bi[i + NB * i];
                    const double w = (i == 0 || i == 9) ? 2.0 * bv * 1.025 : 2.0 * bv;
                    etop = av + w; ebot = av - w; aref = av;
                }
                const double det = (e - etop) * (e - ebot);
                const double zr = det >= 0.0 ? sqrt(det) : 0.0, zi = det >= 0.0 ? 0.0 : sqrt(-det);
                x = make_double2((e + eta_re - aref - zr) * 0.5, (eta_im - zi) * 0.5);
            }
            q[rr][c] = x;
        }
    const double pr = e + (e != 0.0 ? eta_re : 0.0), pim = (e != 0.0 ? eta_im : 0.0);
    double2 an[2][3];
    if (lld > 1) green_fetch(an, a_b + ((size_t)site * lld + (lld - 2)) * BLK, ig, jg);
#pragma unroll 1
    for (int l = lld - 1; l >= 1; --l) {
#pragma unroll
        for (int rr = 0; rr < 2; ++rr)
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const bool dg = (2 * ig + rr) == (3 * jg + c);
                q[rr][c] = make_double2((dg ? pr : 0.0) - an[rr][c].x - q[rr][c].x, (dg ? pim : 0.0) - an[rr][c].y - q[rr][c].y);
            }
        green_stage_b(L, b_sqrt + ((size_t)site * lld + (l - 1)) * BLK, lane);      // (the previous level's products have read L.B: wave_sync below)
        if (l > 1) green_fetch(an, a_b + ((size_t)site * lld + (l - 2)) * BLK, ig, jg);
        gauss_jordan_from<0>(q, L, lane, ig, jg, act);
        unpermute_from<NB - 1>(q, L, ig, jg, act);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                            // B_l is in LDS
        // X = Q^-1 B
        if (act) {
#pragma unroll
            for (int rr = 0; rr < 2; ++rr)
#pragma unroll
                for (int c = 0; c < 3; ++c) L.M[(2 * ig + rr) + NB * (3 * jg + c)] = q[rr][c];
        }
        wave_sync();
        double2 x[2][3];
#pragma unroll
        for (int rr = 0; rr < 2; ++rr)
#pragma unroll
            for (int c = 0; c < 3; ++c) x[rr][c] = make_double2(0.0, 0.0);
#pragma unroll 3
        for (int k = 0; k < NB; ++k) {
            const double2 m0 = L.M[2 * ig + NB * k], m1 = L.M[2 * ig + 1 + NB * k];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const double2 b = L.B[k + NB * (3 * jg + c)];
                x[0][c] = cfma(m0, b, x[0][c]);
                x[1][c] = cfma(m1, b, x[1][c]);
            }
        }
        wave_sync();
        if (act) {
#pragma unroll
            for (int rr = 0; rr < 2; ++rr)
#pragma unroll
                for (int c = 0; c < 3; ++c) L.M[(2 * ig + rr) + NB * (3 * jg + c)] = x[rr][c];
        }
        wave_sync();
        // Q = B^H X
#pragma unroll
        for (int rr = 0; rr < 2; ++rr)
#pragma unroll
            for (int c = 0; c < 3; ++c) q[rr][c] = make_double2(0.0, 0.0);
#pragma unroll 3
        for (int k = 0; k < NB; ++k) {
            const double2 b0 = L.B[k + NB * (2 * ig)], b1 = L.B[k + NB * (2 * ig + 1)];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const double2 xv = L.M[k + NB * (3 * jg + c)];
                q[0][c] = cfmah(b0, xv, q[0][c]);
                q[1][c] = cfmah(b1, xv, q[1][c]);
            }
        }
        wave_sync();
    }
    // coalesced store through the staging matrix
    if (act) {
#pragma unroll
        for (int rr = 0; rr < 2; ++rr)
#pragma unroll
            for (int c = 0; c < 3; ++c) L.M[(2 * ig + rr) + NB * (3 * jg + c)] = q[rr][c];
    }
    wave_sync();
    if (LDOS) {
        if (lane < NB) gim[((size_t)site * nen + ei) * NB + lane] = L.M[lane * (NB + 1)].y;
        return;
    }
    double2* out = g0 + ((size_t)site * nen + ei) * BLK;
#pragma unroll
    for (int m = 0; m < 6; ++m) { const int el = lane + 64 * m; if (el < BLK) out[el] = L.M[el]; }
}

// green%chebyshev_green (green.f90:1030-1108): g0(:,:,ie) = sum_i mu_ng(:,:,i) (-i exp(-i (i-1) acos w_ie)) / sqrt(a^2 - (e_ie - b)^2),
// mu_ng = mu_n * jackson kernel * (2 for i > 1).  grid = (nen, nsites); the nm phase factors of an energy are formed once in LDS,
// then every thread sums the moments of its matrix elements in the reference's order (i = 1 .. nm).
__global__ __launch_bounds__(256) void k_chebyshev_green(int nm, int nen, const double* __restrict__ ene, double a, double b,
                                                        const double* __restrict__ kern /*[nm] jackson * {1,2,2,...}*/,
                                                        const double2* __restrict__ mu /*[site][nm][324]*/, double2* __restrict__ g0 /*[site][nen][324]*/) {
    extern __shared__ double2 ef[];
    const int ie = blockIdx.x, site = blockIdx.y;
    const double e = ene[ie];
    const double th = acos((e - b) / a);
    for (int i = threadIdx.x; i < nm; i += blockDim.x) {
        const double x = (double)i * th;                    // (i - 1) with the reference's 1-based i
        ef[i] = make_double2(-sin(x) * kern[i], -cos(x) * kern[i]);
    }
    __syncthreads();
    const double den = sqrt(a * a - (e - b) * (e - b));
    const double2* m = mu + (size_t)site * nm * BLK;
    for (int el = threadIdx.x; el < BLK; el += blockDim.x) {
        double sr = 0.0, si = 0.0;
        for (int i = 0; i < nm; ++i) {
            const double2 v = m[(size_t)i * BLK + el], f = ef[i];
            sr += v.x * f.x - v.y * f.y;
            si += v.x * f.y + v.y * f.x;
        }
        g0[((size_t)site * nen + ie) * BLK + el] = make_double2(sr / den, si / den);
    }
}

}  // namespace rsrec
