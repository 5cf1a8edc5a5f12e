// Device-side Hermitian 18x18 eigen-decomposition and matrix functions (B = S^{1/2}, B^{-1} = S^{-1/2}).
//
// Replaces the LAPACK/BLAS sequence of crecal_b (reference recursion.f90:1938-1959: zheev('v','u',18) followed by
// four 18x18x18 zgemm) and of zsqr (:2011-2020).  One workgroup per matrix, parallel cyclic Jacobi with the
// round-robin pair ordering (9 disjoint rotations per round, 17 rounds per sweep); all data stay in LDS.
// Only the matrix FUNCTIONS are used downstream, never the eigenvectors themselves (they are not unique for the
// degenerate spectra cubic symmetry produces; the functions are).
#pragma once
#include <hip/hip_runtime.h>

namespace rsrec {

struct Eig18Shared {
    double2 A[18 * 18];   // working copy, becomes diagonal
    double2 V[18 * 18];   // eigenvectors (columns)
    double rot[9][4];     // per pair: c, s, ph.re, ph.im
    int pq[9][2];
    double red[32];
    double ev[18];
    double f1[18], f2[18];
    int flag;
};

__device__ inline double2 cmul(double2 a, double2 b) { return make_double2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
__device__ inline double2 cmulc(double2 a, double2 b) { /* a * conj(b) */ return make_double2(a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y); }

// In: sh.A = Hermitian matrix (column-major).  Out: sh.ev eigenvalues, sh.V eigenvectors. Returns sweeps used (-1: not converged, -2: non-finite input).
__device__ inline int jacobi18(Eig18Shared& sh) {
    const int tid = threadIdx.x, nt = blockDim.x;
    for (int e = tid; e < 324; e += nt) sh.V[e] = make_double2((e % 18) == (e / 18) ? 1.0 : 0.0, 0.0);
    __syncthreads();
    // Frobenius norm (fixed-order sum by thread 0 over 18 column sums -> deterministic)
    if (tid < 18) {
        double s = 0.0;
        for (int r = 0; r < 18; ++r) { double2 v = sh.A[r + 18 * tid]; s += v.x * v.x + v.y * v.y; }
        sh.red[tid] = s;
    }
    __syncthreads();
    double nrm = 0.0;
    for (int c = 0; c < 18; ++c) nrm += sh.red[c];
    __syncthreads();
    if (!(nrm == nrm) || nrm > 1.0e300) {
        // NaN / Inf in -> NaN out, reported as -2.  This is what a Krylov breakdown looks like one level later: sqrt of a
        // rounding-negative (or 1 / an exactly zero) eigenvalue of B^2 puts NaN into B, B^-1 and every vector (recursion.f90:1950-1951),
        // and the NEXT level's zheev gets a NaN matrix.  The compiled reference (MKL zheev) returns info /= 0 for it and crecal_b calls
        // g_logger%fatal('Diagonalization error') (:1942) -- measured: oracle/make_fixtures.py fuzz_seed_case, the 8-atom cell at
        // LL = 14; round 1-3 assumed a silent NaN here.  zsqr only prints the info (:2013) and goes on: its caller ignores -2.
        if (tid < 18) sh.ev[tid] = __builtin_nan("");
        __syncthreads();
        return -2;
    }
    int sweep = 0;
    for (; sweep < 40; ++sweep) {
        if (tid < 18) {
            double s = 0.0;
            for (int r = 0; r < tid; ++r) { double2 v = sh.A[r + 18 * tid]; s += v.x * v.x + v.y * v.y; }
            sh.red[tid] = s;
        }
        __syncthreads();
        double off = 0.0;
        for (int c = 0; c < 18; ++c) off += sh.red[c];
        __syncthreads();
        if (off <= 1.0e-34 * nrm) break;
        for (int round = 0; round < 17; ++round) {
            if (tid < 9) {
                int p, q;
                if (tid == 0) { p = 17; q = round; }
                else { p = (round + tid) % 17; q = (round - tid + 17) % 17; }
                if (p > q) { int t = p; p = q; q = t; }
                const double2 x = sh.A[p + 18 * q];
                const double beta = hypot(x.x, x.y);
                double c = 1.0, s = 0.0, phr = 1.0, phi = 0.0;
                if (beta != 0.0) {
                    const double app = sh.A[p + 18 * p].x, aqq = sh.A[q + 18 * q].x;
                    const double tau = (aqq - app) / (2.0 * beta);
                    const double t = (tau >= 0.0 ? 1.0 : -1.0) / (fabs(tau) + sqrt(1.0 + tau * tau));
                    c = 1.0 / sqrt(1.0 + t * t);
                    s = t * c;
                    phr = x.x / beta; phi = x.y / beta;
                }
                sh.rot[tid][0] = c; sh.rot[tid][1] = s; sh.rot[tid][2] = phr; sh.rot[tid][3] = phi;
                sh.pq[tid][0] = p; sh.pq[tid][1] = q;
            }
            __syncthreads();
            // columns: M <- M*G for M in {A, V};  G_pp=c, G_pq=s, G_qp=-s*conj(ph), G_qq=c*conj(ph)
            for (int task = tid; task < 9 * 18 * 2; task += nt) {
                const int pair = task / 36, rem = task % 36, k = rem % 18, which = rem / 18;
                const double c = sh.rot[pair][0], s = sh.rot[pair][1];
                const double2 ph = make_double2(sh.rot[pair][2], sh.rot[pair][3]);
                const int p = sh.pq[pair][0], q = sh.pq[pair][1];
                double2* M = which ? sh.V : sh.A;
                const double2 mp = M[k + 18 * p], mq = M[k + 18 * q];
                const double2 gqp = make_double2(-s * ph.x, s * ph.y), gqq = make_double2(c * ph.x, -c * ph.y);
                const double2 t1 = cmul(mq, gqp), t2 = cmul(mq, gqq);
                M[k + 18 * p] = make_double2(mp.x * c + t1.x, mp.y * c + t1.y);
                M[k + 18 * q] = make_double2(mp.x * s + t2.x, mp.y * s + t2.y);
            }
            __syncthreads();
            // rows: A <- G^H * A
            for (int task = tid; task < 9 * 18; task += nt) {
                const int pair = task / 18, k = task % 18;
                const double c = sh.rot[pair][0], s = sh.rot[pair][1];
                const double2 ph = make_double2(sh.rot[pair][2], sh.rot[pair][3]);
                const int p = sh.pq[pair][0], q = sh.pq[pair][1];
                const double2 ap = sh.A[p + 18 * k], aq = sh.A[q + 18 * k];
                // conj(G_qp) = -s*ph ; conj(G_qq) = c*ph
                const double2 cgqp = make_double2(-s * ph.x, -s * ph.y), cgqq = make_double2(c * ph.x, c * ph.y);
                const double2 t1 = cmul(cgqp, aq), t2 = cmul(cgqq, aq);
                sh.A[p + 18 * k] = make_double2(c * ap.x + t1.x, c * ap.y + t1.y);
                sh.A[q + 18 * k] = make_double2(s * ap.x + t2.x, s * ap.y + t2.y);
            }
            __syncthreads();
            if (tid < 9) {
                const int p = sh.pq[tid][0], q = sh.pq[tid][1];
                sh.A[p + 18 * q] = make_double2(0.0, 0.0);
                sh.A[q + 18 * p] = make_double2(0.0, 0.0);
                sh.A[p + 18 * p].y = 0.0;
                sh.A[q + 18 * q].y = 0.0;
            }
            __syncthreads();
        }
    }
    if (tid < 18) sh.ev[tid] = sh.A[tid + 18 * tid].x;
    __syncthreads();
    return sweep < 40 ? sweep : -1;
}

// F = V * diag(f) * V^H, written to global `out` (column-major, interleaved complex)
__device__ inline void matfun18(const Eig18Shared& sh, const double* f, double2* out) {
    for (int e = threadIdx.x; e < 324; e += blockDim.x) {
        const int i = e % 18, j = e / 18;
        double2 acc = make_double2(0.0, 0.0);
        for (int k = 0; k < 18; ++k) {
            const double2 t = cmulc(sh.V[i + 18 * k], sh.V[j + 18 * k]);
            acc.x += t.x * f[k];
            acc.y += t.y * f[k];
        }
        out[e] = acc;
    }
}

}  // namespace rsrec
