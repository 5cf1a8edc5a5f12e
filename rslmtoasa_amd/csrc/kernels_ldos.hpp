// The LDOS stage behind the block recursion, on the device: terminator, and the reduction of g0 to densities of states.
//
//  * k_terminator: recursion%get_terminf (recursion.f90:2092-2135) with get_cinf (:2030-2086), bpopt (:3540-3580) and emami
//    (:3589-3700).  For each of the 18 x 18 matrix elements of a site the real parts of A_l(i,j) and of sqrt(B_l^2)(i,j) are
//    treated as a scalar chain; bpopt iterates the Beer-Pettifor band-edge condition, calling emami (largest / smallest
//    eigenvalue of the tridiagonal matrix by bisection on the Sturm count) once per iteration.  One thread per matrix element:
//    64 sites x 324 elements run concurrently; every arithmetic operation is the reference's, in the reference's order (IEEE
//    double, no reassociation), including its quirks: B(1) = B(N+1) = 0 inside emami, the shared 50-step cap of each bisection
//    whose early return leaves the CURRENT bracket in (emax, emin), NaN from all-zero chains (turned into 0 by get_terminf),
//    diagonal zeros -> 0.5, b_inf(1,1) and b_inf(10,10) scaled by 1.01.
//  * k_ldos_finish: bands%calculate_fermi's reduction (bands.f90:258-268): dosial = -Im g0_jj / pi, dosia and dtot summed in the
//    reference's loop order (site, then j = 1..9 pairing orbital j with j + 9), written into zero-padded images over all sites --
//    the arrays the reference all-reduces (bands.f90:271-274).
#pragma once
#include <hip/hip_runtime.h>
#include "kernels_valu.hpp"

namespace rsrec {

// chain of one matrix element, staged in LDS: a[l], rb[l] for l = 0..ll-1 (reference index l + 1), one column per thread
struct TermChain {
    const double* a;
    const double* rb;
    int stride;       // threads per workgroup (LDS column stride)
    __device__ __forceinline__ double A(int i) const { return a[(i - 1) * stride]; }     // 1-based like the reference
    __device__ __forceinline__ double RB(int i) const { return rb[(i - 1) * stride]; }
};

// emami (recursion.f90:3589-3700) on the shifted/scaled chain bpopt builds: AZ(i), RBZ(i), i = 1..n, formed on the fly with the
// reference's expressions (AZ(i) = 0.5 (A(i) - ainf) for i < n, AZ(n) = A(n) - ainf; RBZ(i) = 0.5 RB(i) for 2 <= i < n,
// RBZ(n) = 1/sqrt(2) RB(n); emami then zeroes B(1) and B(n+1)).
struct TermShifted {
    TermChain c;
    double ainf;
    int n;
    __device__ __forceinline__ double az(int i) const {
#pragma clang fp contract(off)
        return i < n ? 0.5 * (c.A(i) - ainf) : c.A(n) - ainf;
    }
    __device__ __forceinline__ double b(int i) const {                     // emami's local B(1..n+1)
#pragma clang fp contract(off)
        if (i <= 1 || i > n) return 0.0;
        return i < n ? 0.5 * c.RB(i) : (1.0 / sqrt(2.0)) * c.RB(n);
    }
};

__device__ __forceinline__ int term_sturm(const TermShifted& S, double e) {
#pragma clang fp contract(off)
    const double relfeh = 1.8189894035458565e-12;        // 2.d0**(-39)
    int num = 0;
    double p = S.az(1) - e;
    if (p < 0.0) ++num;
    for (int i = 2; i <= S.n; ++i) {
        const double bi = S.b(i);
        if (p == 0.0) p = (S.az(i) - e) - fabs(bi) / relfeh;
        else p = (S.az(i) - e) - bi * bi / p;
        if (p < 0.0) ++num;
    }
    return num;
}

__device__ __forceinline__ void term_emami(const TermShifted& S, double& emax, double& emin) {
#pragma clang fp contract(off)
    const int n = S.n;
    double emax0 = -1.0e6, emin0 = 1.0e6;
    for (int i = 1; i <= n; ++i) {
        const double a = S.az(i), b0 = fabs(S.b(i)), b1 = fabs(S.b(i + 1));
        const double x1 = a + b0 + b1, x2 = a - b0 - b1;
        if (emax0 <= x1) emax0 = x1;
        if (emin0 > x2) emin0 = x2;
    }
    const double eps = 1.0e-6;
    int istop = 0;
    emax = emax0; emin = emin0;
    double e;
    for (;;) {                                            // largest eigenvalue
        e = (emax + emin) / 2.0;
        if (++istop > 50) return;                         // `goto 1000`: leaves the current bracket in (emax, emin)
        const int num = term_sturm(S, e);
        if (num == n) emax = e;
        if (num < n) emin = e;
        const double dele = fabs((emax - emin) / ((emax + emin) / 2.0));
        if (dele <= eps) break;
    }
    const double e1 = e;
    istop = 0;
    emax = e1; emin = emin0;
    for (;;) {                                            // smallest eigenvalue
        e = (emax + emin) / 2.0;
        if (++istop > 50) return;
        const int num = term_sturm(S, e);
        if (num == 0) emin = e;
        if (num > 0) emax = e;
        const double dele = fabs((emax - emin) / ((emax + emin) / 2.0));
        if (dele <= eps) break;
    }
    emax = e1; emin = e;
}

// grid = (ceil(324 / T), nsites), block = T threads, dynamic LDS = 2 * lld * T doubles.
// a_b, b_sqrt: [site][lld][324] complex (b_sqrt = b2_b after zsqr, self.f90:829); a_inf, b_inf: [site][324] real (column-major 18x18),
// a_inf0, b_inf0: [site] (may be null).
__global__ void k_terminator(int lld, const double2* __restrict__ a_b, const double2* __restrict__ b_sqrt, double* __restrict__ a_inf,
                             double* __restrict__ b_inf) {
#pragma clang fp contract(off)
    extern __shared__ double chain_lds[];
    const int T = blockDim.x, tid = threadIdx.x, site = blockIdx.y;
    const int el = blockIdx.x * T + tid;
    double* la = chain_lds + tid;
    double* lb = chain_lds + (size_t)lld * T + tid;
    if (el < BLK) {
        for (int l = 0; l < lld; ++l) {
            la[(size_t)l * T] = a_b[((size_t)site * lld + l) * BLK + el].x;         // real(Acoef_b), real(B2coef_b) (:2108-2111)
            lb[(size_t)l * T] = b_sqrt[((size_t)site * lld + l) * BLK + el].x;
        }
    }
    if (el >= BLK) return;                                  // (no barrier below: every thread reads only its own LDS column)
    TermShifted S;
    S.c.a = la; S.c.rb = lb; S.c.stride = T;
    const int n = lld - 1;                                  // bpopt(ll, AA, BB, LL - 1, ...) (:2081)
    S.n = n;
    double ainf = S.c.A(n), bmax = 0.0, bmin = 0.0;
    int jiter = 0;
    for (;;) {                                              // bpopt :3557-3576
        ++jiter;
        S.ainf = ainf;
        term_emami(S, bmax, bmin);
        const double bm = fabs(bmax + bmin);
        ainf = ainf + (bmax + bmin);
        if (bm <= 1.0e-5) break;
        else if (jiter > 300) break;
    }
    double rbinf = (bmax - bmin) / 2.0;
    // get_terminf :2114-2131
    if (isnan(ainf)) ainf = 0.0;
    if (isnan(rbinf)) rbinf = 0.0;
    const int i = el % NB, j = el / NB;
    if (i == j) {
        if (ainf == 0.0) ainf = 0.5;
        if (rbinf == 0.0) rbinf = 0.5;
        if (i == 0 || i == 9) rbinf = rbinf * 1.01;
    }
    a_inf[(size_t)site * BLK + el] = ainf;
    b_inf[(size_t)site * BLK + el] = rbinf;
}

// ---- the stage behind the SCALAR recursion: dos%density (density_of_states.f90:248-363) with bprldos (:370-404), which green%sgreen
// (green.f90:628-705) turns into g0.  Chain c = (orbital nl, site ia, direction md) = nl + 18 (ia + nsites md); a, b2: [chain][llmax].
//
// k_scalar_edges: the Beer-Pettifor band of every chain -- bpOPT(lld, AA, sqrt(BB), lld - 1, ...) (:282), bm * 1.01 on orbitals 1 and 10
// (:283), edge = am - 2 bm, width = 4 bm (:287-288) -> edges[2 c] = edge, edges[2 c + 1] = edge + width.  One thread per chain, the chain
// staged in the thread's LDS column (as k_terminator); block = T threads, dynamic LDS = 2 * lld * T doubles.
__global__ void k_scalar_edges(int lld, int llmax, int nchain, const double* __restrict__ a, const double* __restrict__ b2, double* __restrict__ edges) {
#pragma clang fp contract(off)
    extern __shared__ double chain_lds[];
    const int T = blockDim.x, tid = threadIdx.x, c = blockIdx.x * T + tid;
    if (c >= nchain) return;                                // (no barrier below: every thread reads only its own LDS column)
    double* la = chain_lds + tid;
    double* lb = chain_lds + (size_t)lld * T + tid;
    for (int l = 0; l < lld; ++l) {
        la[(size_t)l * T] = a[(size_t)c * llmax + l];
        lb[(size_t)l * T] = sqrt(b2[(size_t)c * llmax + l]);
    }
    TermShifted S;
    S.c.a = la; S.c.rb = lb; S.c.stride = T;
    const int n = lld - 1;
    S.n = n;
    double ainf = S.c.A(n), bmax = 0.0, bmin = 0.0;
    int jiter = 0;
    for (;;) {                                              // bpopt :3557-3576
        ++jiter;
        S.ainf = ainf;
        term_emami(S, bmax, bmin);
        const double bm = fabs(bmax + bmin);
        ainf = ainf + (bmax + bmin);
        if (bm <= 1.0e-5) break;
        else if (jiter > 300) break;
    }
    double bm1 = (bmax - bmin) / 2.0;
    const int nl = c % NB;
    if (nl == 0 || nl == 9) bm1 = 1.01 * bm1;
    const double edge = ainf - 2.0 * bm1, width = 4.0 * bm1;
    edges[2 * (size_t)c] = edge;
    edges[2 * (size_t)c + 1] = edge + width;
}

// k_scalar_density: tdens(nl, ie, ia, md) = bprldos(ene(ie) / dw_l(nl, ia) - cshi(nl, ia), a, b2, lld, band) / dw_l(nl, ia)   (:331-341).
// bprldos: Qt = (e - emid -+ sqrt((e - etop)(e - ebot))) / 2 on the branch with Im Qt <= 0, then Qt = B2(l) / (e - A(l) - Qt) for
// l = lld - 1 .. 1, result -Im Qt / pi -- complex arithmetic spelled out the way the Fortran evaluates it (real operands promoted to (x, +0)).
// grid = (ceil(npts / blockDim.x), nchain): one thread per (chain, energy); the chain is read through the cache by all threads of a block.
__global__ void k_scalar_density(int lld, int llmax, int npts, int nsites, const double* __restrict__ a, const double* __restrict__ b2, const double* __restrict__ ene,
                                 const double* __restrict__ dw_l, const double* __restrict__ cshi, const double* __restrict__ edges, double* __restrict__ tdens) {
#pragma clang fp contract(off)
    const int c = blockIdx.y, ie = blockIdx.x * blockDim.x + threadIdx.x;
    if (ie >= npts) return;
    const int nl = c % NB, ia = (c / NB) % nsites, md = c / (NB * nsites);
    const double dw = dw_l[nl + NB * (size_t)ia], cs = cshi[nl + NB * (size_t)ia];
    const double e = ene[ie] / dw - 1.00 * cs;
    const double ebot = edges[2 * (size_t)c], etop = edges[2 * (size_t)c + 1];
    const double emid_r = 0.5 * (etop + ebot), emid_i = 0.5 * (0.0 + 0.0);
    const double ea_r = e - etop, ea_i = 0.0 - 0.0, eb_r = e - ebot, eb_i = 0.0 - 0.0;
    const double det_r = ea_r * eb_r - ea_i * eb_i, det_i = ea_r * eb_i + ea_i * eb_r;
    double z_r, z_i;                                         // principal square root (signed zero of the imaginary part decides the branch on the cut)
    if (det_i == 0.0) {
        if (det_r >= 0.0) { z_r = sqrt(det_r); z_i = det_i; }
        else { z_r = 0.0; z_i = copysign(sqrt(-det_r), det_i); }
    } else {
        const double m = hypot(det_r, det_i);
        z_r = sqrt(0.5 * (m + det_r));
        z_i = copysign(sqrt(0.5 * (m - det_r)), det_i);
    }
    double q_r = ((e - emid_r) - z_r) * 0.5, q_i = ((0.0 - emid_i) - z_i) * 0.5;
    if (q_i > 0.0) { q_r = ((e - emid_r) + z_r) * 0.5; q_i = ((0.0 - emid_i) + z_i) * 0.5; }
    const double* aa = a + (size_t)c * llmax;
    const double* bb = b2 + (size_t)c * llmax;
    for (int l = lld - 1; l >= 1; --l) {
        const double cr = (e - aa[l - 1]) - q_r, ci = 0.0 - q_i, den = cr * cr + ci * ci;
        q_r = bb[l - 1] * cr / den;
        q_i = -(bb[l - 1] * ci) / den;
    }
    const double dens = -q_i / 3.14159265358979323846;
    tdens[nl + NB * ((size_t)ie + (size_t)npts * ((size_t)ia + (size_t)nsites * md))] = 0.0 + 1.0 * dens / dw;
}

// a_inf0(n) = mean diagonal of a_inf, b_inf0 likewise (after the 1.01 scaling), summed in index order (:2121-2131)
__global__ void k_terminator_means(const double* __restrict__ a_inf, const double* __restrict__ b_inf, double* __restrict__ a_inf0, double* __restrict__ b_inf0, int nsites) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= nsites) return;
    double a = 0.0, b = 0.0;
    for (int i = 0; i < NB; ++i) { a = a + a_inf[(size_t)s * BLK + i * (NB + 1)]; b = b + b_inf[(size_t)s * BLK + i * (NB + 1)]; }
    a_inf0[s] = a / NB; b_inf0[s] = b / NB;
}

// gim: [site][nen][18] = Im g0(j,j,ie,site).  One thread per energy; the sums run in the reference's order (bands.f90:258-268:
// do ia; do i; do j = 1, 9: dtot(i) -= aimag(g(j,j) + g(j+9,j+9))/pi, dosia likewise, dosial(ia,j,i) = -aimag(g(j,j))/pi).
// Images: dosial(ntot, 18, nen), dosia(ntot, nen), dtot(nen) in Fortran order; sites off+1 .. off+n are this rank's, the rest zero.
__global__ void k_ldos_finish(const double* __restrict__ gim, int n, int nen, int off, int ntot, double* __restrict__ dosial, double* __restrict__ dosia,
                              double* __restrict__ dtot) {
#pragma clang fp contract(off)
    const int ie = blockIdx.x * blockDim.x + threadIdx.x;
    if (ie >= nen) return;
    const double pi = 3.14159265358979323846;              // math.f90: pi = 4 atan(1) rounds to the same double
    double t = 0.0;
    for (int ia = 0; ia < ntot; ++ia) {
        const int s = ia - off;
        const bool mine = s >= 0 && s < n;
        double d = 0.0;
        for (int j = 0; j < 9; ++j) {
            const double g1 = mine ? gim[((size_t)s * nen + ie) * NB + j] : 0.0, g2 = mine ? gim[((size_t)s * nen + ie) * NB + j + 9] : 0.0;
            if (mine) { t = t - (g1 + g2) / pi; d = d - (g1 + g2) / pi; }
            dosial[(size_t)ia + (size_t)ntot * (j + (size_t)NB * ie)] = mine ? -g1 / pi : 0.0;
            dosial[(size_t)ia + (size_t)ntot * (j + 9 + (size_t)NB * ie)] = mine ? -g2 / pi : 0.0;
        }
        dosia[(size_t)ia + (size_t)ntot * ie] = d;
    }
    dtot[ie] = t;
}

}  // namespace rsrec
