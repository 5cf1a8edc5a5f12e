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
#define GREEN_WAVES_N 2     // measured: 1 -> 123 ms, 2 -> 118 ms, 3 -> 128 ms, 4 -> 130 ms, 5 -> 187 ms (64 sites x 2510 energies x LL=50): LDS slice = occupancy
#endif
constexpr int GREEN_WAVES = GREEN_WAVES_N;                       // (site, energy) pairs per workgroup
constexpr int GREEN_LDS_DOUBLES = GREEN_WAVES * (3 * 2 * BLK + 32);   // Q, X, B (complex) + pivot rows, per wave

__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ __forceinline__ double2 gmul(double2 a, double2 b) { return make_double2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }

// in-place inverse of the 18x18 complex matrix M (column-major, in this wave's LDS slice); piv: 18 ints of LDS scratch
__device__ __forceinline__ void wave_inverse18(double2* M, int* piv, int lane) {
#pragma unroll 1
    for (int k = 0; k < NB; ++k) {
        // pivot: first row i >= k of maximal |re| + |im| in column k
        double v = -1.0;
        int idx = lane;
        if (lane >= k && lane < NB) { const double2 x = M[lane + NB * k]; v = fabs(x.x) + fabs(x.y); }
#pragma unroll
        for (int off = 16; off >= 1; off >>= 1) {
            const double v2 = __shfl_xor(v, off, 64);
            const int i2 = __shfl_xor(idx, off, 64);
            if (v2 > v || (v2 == v && i2 < idx)) { v = v2; idx = i2; }
        }
        const int p = __builtin_amdgcn_readfirstlane(idx);
        if (lane == 0) piv[k] = p;
        if (p != k && lane < NB) {                               // swap rows k and p
            const double2 a = M[k + NB * lane], b = M[p + NB * lane];
            M[k + NB * lane] = b; M[p + NB * lane] = a;
        }
        wave_sync();
        const double2 pv = M[k + NB * k];
        const double den = pv.x * pv.x + pv.y * pv.y;
        const double2 ip = make_double2(pv.x / den, -pv.y / den);
        double2 f[6];                                            // multiplier M[i][k] of the row each of my elements sits in
#pragma unroll
        for (int m = 0; m < 6; ++m) { const int e = lane + 64 * m; f[m] = (e < BLK) ? M[(e % NB) + NB * k] : make_double2(0.0, 0.0); }
        wave_sync();
        if (lane < NB) {                                         // pivot row: (k,k) -> 1, then scale
            const double2 x = (lane == k) ? make_double2(1.0, 0.0) : M[k + NB * lane];
            M[k + NB * lane] = gmul(x, ip);
        }
        wave_sync();
#pragma unroll
        for (int m = 0; m < 6; ++m) {
            const int e = lane + 64 * m;
            if (e < BLK) {
                const int i = e % NB, j = e / NB;
                if (i != k) {
                    const double2 r = M[k + NB * j];
                    const double2 x = (j == k) ? make_double2(0.0, 0.0) : M[e];
                    const double2 t = gmul(f[m], r);
                    M[e] = make_double2(x.x - t.x, x.y - t.y);
                }
            }
        }
        wave_sync();
    }
#pragma unroll 1
    for (int k = NB - 1; k >= 0; --k) {                          // undo the row interchanges as column interchanges
        const int p = piv[k];
        if (p != k && lane < NB) {
            const double2 a = M[lane + NB * k], b = M[lane + NB * p];
            M[lane + NB * k] = b; M[lane + NB * p] = a;
        }
        wave_sync();
    }
}

// grid = (ceil(nen / GREEN_WAVES), nsites).  a_b, b_sqrt: [site][lld][324] complex; a_inf, b_inf: [site][324] real; g0: [site][nen][324]
__global__ __launch_bounds__(GREEN_WAVES * 64) void k_block_green(int lld, int nen, const double* __restrict__ ene, double eta_re, double eta_im, int sym_term,
                                                                 const double* __restrict__ a_inf, const double* __restrict__ b_inf,
                                                                 const double2* __restrict__ a_b, const double2* __restrict__ b_sqrt, double2* __restrict__ g0) {
    __shared__ double lds[GREEN_LDS_DOUBLES];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ei = blockIdx.x * GREEN_WAVES + wave, site = blockIdx.y;
    if (ei >= nen) return;                                       // wave-uniform; no workgroup barriers below
    double* mine = lds + (size_t)wave * (3 * 2 * BLK + 32);
    double2* Q = reinterpret_cast<double2*>(mine);
    double2* X = Q + BLK;
    double2* B = X + BLK;
    int* piv = reinterpret_cast<int*>(B + BLK);
    const double e = ene[ei];
    const double* ai = a_inf + (size_t)site * BLK;
    const double* bi = b_inf + (size_t)site * BLK;
    const double a_diag = 0.5 * (ai[0] + ai[9 + NB * 9]), b_diag = 0.5 * (bi[0] + bi[9 + NB * 9]);
    // terminator (:1263-1289)
#pragma unroll
    for (int m = 0; m < 6; ++m) {
        const int el = lane + 64 * m;
        if (el < BLK) {
            const int i = el % NB, j = el / NB;
            double2 q = make_double2(0.0, 0.0);
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
                q = make_double2((e + eta_re - aref - zr) * 0.5, (eta_im - zi) * 0.5);
            }
            Q[el] = q;
        }
    }
    wave_sync();
    const double pr = e + (e != 0.0 ? eta_re : 0.0), pim = (e != 0.0 ? eta_im : 0.0);
#pragma unroll 1
    for (int l = lld - 1; l >= 1; --l) {
        const double2* A = a_b + ((size_t)site * lld + (l - 1)) * BLK;
        const double2* Bl = b_sqrt + ((size_t)site * lld + (l - 1)) * BLK;
#pragma unroll
        for (int m = 0; m < 6; ++m) {
            const int el = lane + 64 * m;
            if (el < BLK) {
                const int i = el % NB, j = el / NB;
                const double2 a = A[el], q = Q[el];
                Q[el] = make_double2((i == j ? pr : 0.0) - a.x - q.x, (i == j ? pim : 0.0) - a.y - q.y);
                B[el] = Bl[el];
            }
        }
        wave_sync();
        wave_inverse18(Q, piv, lane);
        // X = Q B
#pragma unroll
        for (int m = 0; m < 6; ++m) {
            const int el = lane + 64 * m;
            if (el < BLK) {
                const int i = el % NB, j = el / NB;
                double2 s = make_double2(0.0, 0.0);
#pragma unroll 6
                for (int k = 0; k < NB; ++k) {
                    const double2 t = gmul(Q[i + NB * k], B[k + NB * j]);
                    s.x += t.x; s.y += t.y;
                }
                X[el] = s;
            }
        }
        wave_sync();
        // Q = B^H X
#pragma unroll
        for (int m = 0; m < 6; ++m) {
            const int el = lane + 64 * m;
            if (el < BLK) {
                const int i = el % NB, j = el / NB;
                double2 s = make_double2(0.0, 0.0);
#pragma unroll 6
                for (int k = 0; k < NB; ++k) {
                    const double2 b = B[k + NB * i], x = X[k + NB * j];
                    s.x += b.x * x.x + b.y * x.y;
                    s.y += b.x * x.y - b.y * x.x;
                }
                Q[el] = s;
            }
        }
        wave_sync();
    }
    double2* out = g0 + ((size_t)site * nen + ei) * BLK;
#pragma unroll
    for (int m = 0; m < 6; ++m) { const int el = lane + 64 * m; if (el < BLK) out[el] = Q[el]; }
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
