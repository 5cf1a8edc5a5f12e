// Straightforward FP64 VALU kernels for every step of the recursion ("reference kernels" of the engine).
//
// They work directly on the reference's native layout (18x18 complex(8) column-major blocks, interleaved re/im)
// and are the on-device cross-check for the MFMA kernels (kernels_mfma.hpp).  Region handling: instead of the
// reference's izero/idum flags (recursion.f90:1604-1636) the host computes the breadth-first order of the atoms
// around the seed once; the active region after `level` applications of H is the prefix order[0..cum[level]).
// Blocks outside the region are exactly zero, so multiplying them is arithmetically identical to skipping them.
#pragma once
#include <hip/hip_runtime.h>
#include "eig18.hpp"

namespace rsrec {

constexpr int NB = 18;
constexpr int BLK = 324;            // complex elements per block
constexpr int BLD = 648;            // doubles per block (both layouts)

// Work-vector layouts.  A block is 648 doubles either way.
//  LayoutCM: the reference's own storage, complex(8) column-major interleaved: (r,c) -> re at 2*(r+18c), im next to it.
//  LayoutRM: "row-major planar", the MFMA kernels' layout: (r,c) -> re at 36r+c, im at 36r+18+c.  Rows of 36 doubles
//            [re(0..17) | im(0..17)] are what v_mfma_f64_16x16x4 wants as B operand (k = row, n = column on the lanes).
struct LayoutCM {
    static __device__ __forceinline__ double2 ld(const double* b, int r, int c) { return reinterpret_cast<const double2*>(b)[r + NB * c]; }
    static __device__ __forceinline__ void st(double* b, int r, int c, double2 v) { reinterpret_cast<double2*>(b)[r + NB * c] = v; }
};
struct LayoutRM {
    static __device__ __forceinline__ double2 ld(const double* b, int r, int c) { return make_double2(b[36 * r + c], b[36 * r + 18 + c]); }
    static __device__ __forceinline__ void st(double* b, int r, int c, double2 v) { b[36 * r + c] = v.x; b[36 * r + 18 + c] = v.y; }
};
constexpr int TILE_ATOMS = 14;      // 14 atoms x 18 = 252 of 256 threads
constexpr int NTHREADS = 256;

struct DevProblem {
    int kk, nslots, hstride, nmax, hoh;   // nslots = slots per atom in nbr; hstride = slots per type/atom in the operator tables
    const int* nbr;          // [kk][nslots]: 0-based atom of each slot (slot 0 = the atom itself), -1 = absent
    const int* iz;           // [kk] 0-based type
    const double2* h_st;     // [ntype][nslots][324]  stencil blocks, slot 0 = on-site (+lsham folded in unless hoh)
    const double2* h_loc;    // [nmax][nslots][324]   per-atom blocks of the impurity region
    const double2* ho_st;    // h*obar counterparts (hoh)
    const double2* ho_loc;
    const double2* enim;     // [ntype][324]
    const double2* lsham;    // [ntype][324]
};

struct ChainView {           // per launch: a batch of chains with identical strides
    const int* order;        // [nchain][ostride], entries < 0 are padding
    const int* cum;          // [nchain][nlev]
    const int* obase;        // [nchain][nlev] start of the list to use at a level inside the order row (0 = level-major list;
                             //   levels at which the region is (nearly) the whole lattice point at a spatially sorted list of ALL atoms)
    int nlev;
    size_t vstride;          // doubles between chains in a work vector = (kk+1)*648 (last block is an all-zero block)
    int cpo;                 // chains sharing one order/cum row (1; 18 for the scalar recursion's orbital chains)
    int ostride;             // entries per order row (kk, or more when the list is padded into type-homogeneous groups of 8)
    __device__ __forceinline__ const int* order_of(int chain, int level) const { return order + (size_t)(chain / cpo) * ostride + obase[(chain / cpo) * nlev + level]; }
    __device__ __forceinline__ int count_of(int chain, int level) const { return cum[(chain / cpo) * nlev + level]; }
};

__device__ __forceinline__ void cfma(double2& c, const double2 a, const double2 b) {
    c.x = fma(a.x, b.x, c.x); c.x = fma(-a.y, b.y, c.x);
    c.y = fma(a.x, b.y, c.y); c.y = fma(a.y, b.x, c.y);
}
__device__ __forceinline__ void cfma_conj(double2& c, const double2 a, const double2 b) {  // c += conj(a)*b
    c.x = fma(a.x, b.x, c.x); c.x = fma(a.y, b.y, c.x);
    c.y = fma(a.x, b.y, c.y); c.y = fma(-a.y, b.x, c.y);
}

__device__ __forceinline__ const double2* op_block(const DevProblem& P, bool o, int i, int slot) {
    if (i < P.nmax) return (o ? P.ho_loc : P.h_loc) + (size_t)BLK * (slot + (size_t)P.hstride * i);
    return (o ? P.ho_st : P.h_st) + (size_t)BLK * (slot + (size_t)P.hstride * P.iz[i]);
}

// acc[r] += sum_slots H(slot)[r,:] * in_{nbr(i,slot)}[:,c]
template <class L>
__device__ __forceinline__ void apply_column(const DevProblem& P, bool o, int i, const double* __restrict__ in, int c, double2 acc[NB]) {
    const int* nb = P.nbr + (size_t)P.nslots * i;
    for (int slot = 0; slot < P.nslots; ++slot) {
        const int n = nb[slot];
        if (n < 0) continue;
        const double* x = in + (size_t)BLD * n;
        const double2* H = op_block(P, o, i, slot);
#pragma unroll 2
        for (int k = 0; k < NB; ++k) {
            const double2 xk = L::ld(x, k, c);
#pragma unroll
            for (int r = 0; r < NB; ++r) cfma(acc[r], H[r + NB * k], xk);
        }
    }
}
// acc[r] += M[r,:] * x[:,c] for a single per-type block
template <class L>
__device__ __forceinline__ void apply_block_col(const double2* __restrict__ M, const double* __restrict__ xblk, int c, double2 acc[NB]) {
    for (int k = 0; k < NB; ++k) {
        const double2 xk = L::ld(xblk, k, c);
#pragma unroll
        for (int r = 0; r < NB; ++r) cfma(acc[r], M[r + NB * k], xk);
    }
}

// Block-level sum of per-thread 18-vectors part[c'] (thread owns column c of atom slot a) into red[324], fixed order.
__device__ __forceinline__ void block_reduce_cols(double2* red /*LDS 324*/, const double2 part[NB], int a, int c, bool valid) {
    for (int e = threadIdx.x; e < BLK; e += blockDim.x) red[e] = make_double2(0.0, 0.0);
    __syncthreads();
    for (int s = 0; s < TILE_ATOMS; ++s) {
        if (valid && a == s) {
#pragma unroll
            for (int cp = 0; cp < NB; ++cp) { red[cp + NB * c].x += part[cp].x; red[cp + NB * c].y += part[cp].y; }
        }
        __syncthreads();
    }
}

enum ApplyMode {
    AM_LANCZOS = 0,   // pmn <- H psi - pmn ; A += psi^H (H psi)                       hop_b :1641-1642
    AM_STORE = 1,     // out <- h in                                                    hop_b_hoh pass 1 :1430-1477
    AM_HOH_LANCZOS = 2,  // t = hpsi - ho*hpsi + enim psi + lsham psi ; then as AM_LANCZOS  :1482-1548
    AM_CHEB1 = 3,     // psi1 <- (H psi0 - b psi0)/a ; mom += psi0^H psi1               cheb_1st_mom :2169-2238
    AM_CHEBN = 4,     // psi2 <- 2 (H psi1 - b psi1)/a - psi0 ; d1 += psi1^H psi1 ; d2 += psi2^H psi1   :2495-2597
    AM_HOH_CHEB1 = 5, // the two above with the hoh operator (second pass)              :2245-2369, :2605-2763
    AM_HOH_CHEBN = 6
};

struct ApplyArgs {
    const double* in;      // vector the neighbour sum runs over (psi | hpsi | psi1)
    const double* v0;      // psi (Lanczos) | psi0 (Chebyshev)
    const double* v1;      // hoh second pass: hpsi (first-pass result) ; Chebyshev hoh: same
    double* out;           // pmn | hpsi | psi1 | psi2
    const double* cur;     // Chebyshev: psi1 (the vector H was applied to in pass 1); Lanczos hoh: psi
    double2* partial;      // [nchain][nblk][nk][324]
    int level;             // region = order[0 .. cum[level])
    double a, b;
};

// PRE = true: the neighbour sum was already formed by the matrix-core SpMM (kernels_mfma.hpp) and sits in G.in; this kernel
// then only runs the per-atom epilogue of the mode (combine / scale-shift / moments).
template <int MODE, class L, bool PRE = false>
__global__ __launch_bounds__(NTHREADS) void k_apply(DevProblem P, ChainView CV, ApplyArgs G) {
    __shared__ double2 red[BLK];
    const int chain = blockIdx.y;
    const int a = threadIdx.x / NB, c = threadIdx.x % NB;
    const int count = CV.count_of(chain, G.level);
    const int* order = CV.order_of(chain, G.level);
    const size_t vo = (size_t)chain * CV.vstride;
    const double* in = G.in + vo;
    constexpr bool second_pass = (MODE == AM_HOH_LANCZOS || MODE == AM_HOH_CHEB1 || MODE == AM_HOH_CHEBN);
    constexpr int nk = (MODE == AM_CHEBN || MODE == AM_HOH_CHEBN) ? 2 : 1;
    double2 part0[NB], part1[NB];
#pragma unroll
    for (int r = 0; r < NB; ++r) { part0[r] = make_double2(0, 0); part1[r] = make_double2(0, 0); }

    for (int tile = blockIdx.x; tile * TILE_ATOMS < count; tile += gridDim.x) {
        const int t = tile * TILE_ATOMS + a;
        if (a >= TILE_ATOMS || t >= count) continue;
        const int i = order[t];
        if (i < 0) continue;                       // padding entry of a type-homogeneous group list
        const size_t bo = vo + (size_t)BLD * i;
        double2 acc[NB];
#pragma unroll
        for (int r = 0; r < NB; ++r) acc[r] = make_double2(0, 0);
        if (PRE) {
#pragma unroll
            for (int r = 0; r < NB; ++r) acc[r] = L::ld(in + (size_t)BLD * i, r, c);
        } else {
            apply_column<L>(P, second_pass, i, in, c, acc);
        }
        if (MODE == AM_STORE) {
#pragma unroll
            for (int r = 0; r < NB; ++r) L::st(G.out + bo, r, c, acc[r]);
            continue;
        }
        if (second_pass) {
            // acc = (h*o) applied to hpsi ; total = hpsi_i - acc + enim*x_i + lsham*x_i   (x = psi | psi0 | psi1)
            const double* x = G.cur + bo;
            double2 t2[NB];
#pragma unroll
            for (int r = 0; r < NB; ++r) { const double2 h = L::ld(G.v1 + bo, r, c); t2[r] = make_double2(h.x - acc[r].x, h.y - acc[r].y); acc[r] = make_double2(0, 0); }
            apply_block_col<L>(P.enim + (size_t)BLK * P.iz[i], x, c, acc);
#pragma unroll
            for (int r = 0; r < NB; ++r) { t2[r].x += acc[r].x; t2[r].y += acc[r].y; acc[r] = make_double2(0, 0); }
            apply_block_col<L>(P.lsham + (size_t)BLK * P.iz[i], x, c, acc);
#pragma unroll
            for (int r = 0; r < NB; ++r) { acc[r].x += t2[r].x; acc[r].y += t2[r].y; }
        }
        if (MODE == AM_LANCZOS || MODE == AM_HOH_LANCZOS) {
            const double* psi = G.v0 + bo;
#pragma unroll
            for (int r = 0; r < NB; ++r) {
                const double2 p = L::ld(G.out + bo, r, c);
                L::st(G.out + bo, r, c, make_double2(acc[r].x - p.x, acc[r].y - p.y));
            }
#pragma unroll
            for (int cp = 0; cp < NB; ++cp) {
#pragma unroll
                for (int r = 0; r < NB; ++r) cfma_conj(part0[cp], L::ld(psi, r, cp), acc[r]);
            }
        } else {
            // Chebyshev: x = vector H was applied to (psi0 for the first moment, psi1 afterwards)
            const double* x = G.cur + bo;
            const bool first = (MODE == AM_CHEB1 || MODE == AM_HOH_CHEB1);
            double2 nv[NB];
#pragma unroll
            for (int r = 0; r < NB; ++r) {
                const double2 xr = L::ld(x, r, c);
                double vx = acc[r].x - G.b * xr.x, vy = acc[r].y - G.b * xr.y;
                vx = vx / G.a; vy = vy / G.a;
                if (!first) {
                    vx = 2.0 * vx; vy = 2.0 * vy;
                    const double2 p0 = L::ld(G.v0 + bo, r, c);
                    vx -= p0.x; vy -= p0.y;
                }
                nv[r] = make_double2(vx, vy);
                L::st(G.out + bo, r, c, nv[r]);
            }
            if (first) {
                const double* p0 = G.v0 + bo;   // psi0^H psi1
#pragma unroll
                for (int cp = 0; cp < NB; ++cp) {
#pragma unroll
                    for (int r = 0; r < NB; ++r) cfma_conj(part0[cp], L::ld(p0, r, cp), nv[r]);
                }
            } else {
                // d1[cp][c] += psi1[:,cp]^H psi1[:,c] ; d2[cp][c] += psi2[:,cp]^H psi1[:,c]: this thread owns column c of
                // psi2 but needs column c of psi1 on the right -> d2 is accumulated transposed-conjugated:
                // d2[c][cp] = conj( psi1[:,cp]^H psi2[:,c] ), fixed up in the reduction kernel.
                double2 x1c[NB];
#pragma unroll
                for (int r = 0; r < NB; ++r) x1c[r] = L::ld(x, r, c);
#pragma unroll
                for (int cp = 0; cp < NB; ++cp) {
#pragma unroll
                    for (int r = 0; r < NB; ++r) {
                        const double2 x1 = L::ld(x, r, cp);
                        cfma_conj(part0[cp], x1, x1c[r]);
                        cfma_conj(part1[cp], x1, nv[r]);
                    }
                }
            }
        }
    }
    if (MODE == AM_STORE) return;
    const bool valid = a < TILE_ATOMS;
    double2* pout = G.partial + ((size_t)chain * gridDim.x + blockIdx.x) * nk * BLK;
    block_reduce_cols(red, part0, a, c, valid);
    for (int e = threadIdx.x; e < BLK; e += blockDim.x) pout[e] = red[e];
    if (nk == 2) {
        __syncthreads();
        block_reduce_cols(red, part1, a, c, valid);
        for (int e = threadIdx.x; e < BLK; e += blockDim.x) pout[BLK + e] = red[e];
    }
}

// A partial only: A += psi_i^H t_i  (t = H psi already stored by an SpMM kernel)            hop_b :1642
template <class L>
__global__ __launch_bounds__(NTHREADS) void k_adot(ChainView CV, int level, const double* __restrict__ psi, const double* __restrict__ tvec, double2* partial) {
    __shared__ double2 red[BLK];
    const int chain = blockIdx.y;
    const int a = threadIdx.x / NB, c = threadIdx.x % NB;
    const int count = CV.count_of(chain, level);
    const int* order = CV.order_of(chain, level);
    const size_t vo = (size_t)chain * CV.vstride;
    double2 part[NB];
#pragma unroll
    for (int r = 0; r < NB; ++r) part[r] = make_double2(0, 0);
    for (int tile = blockIdx.x; tile * TILE_ATOMS < count; tile += gridDim.x) {
        const int t = tile * TILE_ATOMS + a;
        if (a >= TILE_ATOMS || t >= count) continue;
        const int i = order[t];
        if (i < 0) continue;
        const size_t bo = vo + (size_t)BLD * i;
        double2 tc[NB];
#pragma unroll
        for (int r = 0; r < NB; ++r) tc[r] = L::ld(tvec + bo, r, c);
#pragma unroll
        for (int cp = 0; cp < NB; ++cp) {
#pragma unroll
            for (int r = 0; r < NB; ++r) cfma_conj(part[cp], L::ld(psi + bo, r, cp), tc[r]);
        }
    }
    block_reduce_cols(red, part, a, c, a < TILE_ATOMS);
    double2* pout = partial + ((size_t)chain * gridDim.x + blockIdx.x) * BLK;
    for (int e = threadIdx.x; e < BLK; e += blockDim.x) pout[e] = red[e];
}

// K2: pmn_i <- [t_i -] pmn_i - psi_i * A ; B2 += pmn_i^H pmn_i      crecal_b :1922-1934 (+ hop_b :1641 when tvec != null)
template <class L>
__global__ __launch_bounds__(NTHREADS) void k_orth(ChainView CV, int level, const double* __restrict__ psi, double* pmn, const double* __restrict__ tvec,
                                                   const double2* __restrict__ Amat /*[nchain] stride astride*/, size_t astride, double2* partial) {
    __shared__ double2 As[BLK];
    __shared__ double2 red[BLK];
    extern __shared__ __attribute__((aligned(16))) unsigned char dyn[];
    double2* tilebuf = reinterpret_cast<double2*>(dyn);   // [TILE_ATOMS][324]
    const int chain = blockIdx.y;
    const int a = threadIdx.x / NB, r = threadIdx.x % NB;
    const int count = CV.count_of(chain, level);
    const int* order = CV.order_of(chain, level);
    const size_t vo = (size_t)chain * CV.vstride;
    for (int e = threadIdx.x; e < BLK; e += blockDim.x) As[e] = Amat[chain * astride + e];
    double2 part[NB];
#pragma unroll
    for (int q = 0; q < NB; ++q) part[q] = make_double2(0, 0);
    __syncthreads();
    for (int tile = blockIdx.x; tile * TILE_ATOMS < count; tile += gridDim.x) {
        const int t = tile * TILE_ATOMS + a;
        const int i = (a < TILE_ATOMS && t < count) ? order[t] : -1;
        const bool valid = i >= 0;
        if (valid) {
            const size_t bo = vo + (size_t)BLD * i;
            double2 prow[NB], srow[NB];
#pragma unroll
            for (int k = 0; k < NB; ++k) { srow[k] = L::ld(psi + bo, r, k); prow[k] = L::ld(pmn + bo, r, k); }
            if (tvec) {
#pragma unroll
                for (int k = 0; k < NB; ++k) { const double2 h = L::ld(tvec + bo, r, k); prow[k] = make_double2(h.x - prow[k].x, h.y - prow[k].y); }
            }
#pragma unroll
            for (int c = 0; c < NB; ++c) {
                double2 acc = make_double2(0, 0);
#pragma unroll
                for (int k = 0; k < NB; ++k) cfma(acc, srow[k], As[k + NB * c]);
                prow[c].x -= acc.x; prow[c].y -= acc.y;
            }
#pragma unroll
            for (int c = 0; c < NB; ++c) { L::st(pmn + bo, r, c, prow[c]); tilebuf[a * BLK + r + NB * c] = prow[c]; }
        }
        __syncthreads();
        if (valid) {
            const int c = r;  // now the thread owns COLUMN c of the updated block
            const double2* T = tilebuf + a * BLK;
#pragma unroll
            for (int cp = 0; cp < NB; ++cp) {
#pragma unroll 6
                for (int q = 0; q < NB; ++q) cfma_conj(part[cp], T[q + NB * cp], T[q + NB * c]);
            }
        }
        __syncthreads();
    }
    block_reduce_cols(red, part, a, r, a < TILE_ATOMS);
    double2* pout = partial + ((size_t)chain * gridDim.x + blockIdx.x) * BLK;
    for (int e = threadIdx.x; e < BLK; e += blockDim.x) pout[e] = red[e];
}

// K3: psi_i <- pmn_i * Binv ; pmn_i <- psi_old_i * B       crecal_b :1963-1969
template <class L>
__global__ __launch_bounds__(NTHREADS) void k_update(ChainView CV, int level, double* psi, double* pmn, const double2* __restrict__ Bmats /*[nchain][2][324]: B, Binv*/) {
    __shared__ double2 Bs[BLK], Bis[BLK];
    const int chain = blockIdx.y;
    const int a = threadIdx.x / NB, r = threadIdx.x % NB;
    const int count = CV.count_of(chain, level);
    const int* order = CV.order_of(chain, level);
    const size_t vo = (size_t)chain * CV.vstride;
    for (int e = threadIdx.x; e < BLK; e += blockDim.x) { Bs[e] = Bmats[(size_t)chain * 2 * BLK + e]; Bis[e] = Bmats[(size_t)chain * 2 * BLK + BLK + e]; }
    __syncthreads();
    for (int tile = blockIdx.x; tile * TILE_ATOMS < count; tile += gridDim.x) {
        const int t = tile * TILE_ATOMS + a;
        if (a >= TILE_ATOMS || t >= count) continue;
        const int i = order[t];
        if (i < 0) continue;
        const size_t bo = vo + (size_t)BLD * i;
        double2 prow[NB], srow[NB];
#pragma unroll
        for (int k = 0; k < NB; ++k) { srow[k] = L::ld(psi + bo, r, k); prow[k] = L::ld(pmn + bo, r, k); }
        for (int c = 0; c < NB; ++c) {
            double2 x = make_double2(0, 0), y = make_double2(0, 0);
#pragma unroll
            for (int k = 0; k < NB; ++k) { cfma(x, prow[k], Bis[k + NB * c]); cfma(y, srow[k], Bs[k + NB * c]); }
            L::st(psi + bo, r, c, x);
            L::st(pmn + bo, r, c, y);
        }
    }
}

// Deterministic sum of the per-block partials: out = sum_p partial[chain][p][which]  (fixed order, 3 interleaved lanes)
__device__ __forceinline__ double2 reduce_partials(const double2* __restrict__ partial, int nblk, int nk, int which, double2* lds /*3*324*/) {
    const int tid = threadIdx.x, c = tid % BLK, g = tid / BLK;
    if (g < 3) {
        double2 s = make_double2(0, 0);
        for (int p = g; p < nblk; p += 3) { const double2 v = partial[((size_t)p * nk + which) * BLK + c]; s.x += v.x; s.y += v.y; }
        lds[g * BLK + c] = s;
    }
    __syncthreads();
    double2 s = make_double2(0, 0);
    if (tid < BLK) { for (int q = 0; q < 3; ++q) { s.x += lds[q * BLK + tid].x; s.y += lds[q * BLK + tid].y; } }
    __syncthreads();
    return s;   // valid for tid < 324
}

// R1: A_n  (atemp_b(:,:,ll) = summ, hop_b :1647)
__global__ __launch_bounds__(1024) void k_reduce_a(const double2* __restrict__ partial, int nblk, double2* a_out, size_t astride) {
    __shared__ double2 lds[3 * BLK];
    const int chain = blockIdx.x;
    const double2 s = reduce_partials(partial + (size_t)chain * nblk * BLK, nblk, 1, 0, lds);
    if (threadIdx.x < BLK) a_out[chain * astride + threadIdx.x] = s;
}

// R2: B2_{n+1} = sum ; B = sqrt(B2), Binv   (crecal_b :1936-1960).  b2_out receives the summed B^2.
__global__ __launch_bounds__(1024) void k_reduce_b_eig(const double2* __restrict__ partial, int nblk, double2* b2_out, size_t bstride, double2* Bmats, int* status) {
    __shared__ double2 lds[3 * BLK];
    __shared__ Eig18Shared sh;
    const int chain = blockIdx.x;
    const double2 s = reduce_partials(partial + (size_t)chain * nblk * BLK, nblk, 1, 0, lds);
    if (threadIdx.x < BLK) { b2_out[chain * bstride + threadIdx.x] = s; sh.A[threadIdx.x] = s; }
    __syncthreads();
    const int sw = jacobi18(sh);
    if (sw < 0 && threadIdx.x == 0) atomicOr(status, 1);
    if (threadIdx.x < NB) { const double l = sqrt(sh.ev[threadIdx.x]); sh.f1[threadIdx.x] = l; sh.f2[threadIdx.x] = 1.0 / l; }
    __syncthreads();
    matfun18(sh, sh.f1, Bmats + (size_t)chain * 2 * BLK);
    matfun18(sh, sh.f2, Bmats + (size_t)chain * 2 * BLK + BLK);
}

// zsqr: in-place sqrt of nmat Hermitian matrices (recursion.f90:1980-2023)
__global__ __launch_bounds__(256) void k_zsqr(double2* mats, int* status) {
    __shared__ Eig18Shared sh;
    double2* M = mats + (size_t)blockIdx.x * BLK;
    for (int e = threadIdx.x; e < BLK; e += blockDim.x) sh.A[e] = M[e];
    __syncthreads();
    const int sw = jacobi18(sh);
    if (sw == -1 && threadIdx.x == 0) atomicOr(status, 1);    // (non-finite input, -2: the reference's zsqr prints zheev's info and goes on, :2013)
    if (threadIdx.x < NB) sh.f1[threadIdx.x] = sqrt(sh.ev[threadIdx.x]);
    __syncthreads();
    matfun18(sh, sh.f1, M);
}

// Chebyshev moment reductions.  first: mu[1] = sum(psi0^H psi1).  else: mu[2ll] = 2 d1 - mu[0], mu[2ll+1] = 2 d2 - mu[1]
// (0-based moment index; recursion.f90:2591-2592) and the divergence test of :2594.
__global__ __launch_bounds__(1024) void k_reduce_cheb(const double2* __restrict__ partial, int nblk, int first, int ll, double2* mu, size_t mustride, int* status, int check_both) {
    __shared__ double2 lds[3 * BLK];
    __shared__ double tr[BLK];
    const int chain = blockIdx.x, tid = threadIdx.x;
    double2* m = mu + chain * mustride;
    if (first) {
        const double2 s = reduce_partials(partial + (size_t)chain * nblk * BLK, nblk, 1, 0, lds);
        if (tid < BLK) m[BLK + tid] = s;
        return;
    }
    const double2 d1 = reduce_partials(partial + (size_t)chain * nblk * 2 * BLK, nblk, 2, 0, lds);
    const double2 d2t = reduce_partials(partial + (size_t)chain * nblk * 2 * BLK, nblk, 2, 1, lds);
    // d2t[cp + 18 c] = psi1[:,cp]^H psi2[:,c]  ->  d2[c][cp] = conj(d2t[cp][c])
    if (tid < BLK) lds[tid] = d2t;
    __syncthreads();
    if (tid < BLK) {
        const int i = tid % NB, j = tid / NB;
        const double2 tt = lds[j + NB * i];
        const double2 d2 = make_double2(tt.x, -tt.y);
        const double2 m0 = m[tid], m1 = m[BLK + tid];
        const double2 o1 = make_double2(2.0 * d1.x - m0.x, 2.0 * d1.y - m0.y);
        const double2 o2 = make_double2(2.0 * d2.x - m1.x, 2.0 * d2.y - m1.y);
        m[(size_t)(2 * ll) * BLK + tid] = o1;
        m[(size_t)(2 * ll + 1) * BLK + tid] = o2;
        tr[tid] = o2.x;
        if (check_both) lds[tid] = make_double2(o1.x, 0.0);
    }
    __syncthreads();
    if (tid == 0) {
        double s = 0.0, s1 = 0.0;
        for (int e = 0; e < BLK; ++e) { s += tr[e]; if (check_both) s1 += lds[e].x; }
        if (s > 1000.0 || (check_both && s1 > 1000.0)) atomicOr(status, 2);   // :2594 (and :2484 for the pair variant)
    }
}

// psi(l,l,seed_k) = coef_k for k = 1..nseed IN ORDER (a later seed on the same atom overwrites, recursion.f90:1709-1714)
template <class L>
__global__ void k_seed(double* psi, size_t vstride, const int* seed_atoms, const double2* seed_coef, int nseed) {
    const int chain = blockIdx.x;
    if (threadIdx.x < NB)
        for (int s = 0; s < nseed; ++s) {
            const int atom = seed_atoms[chain * nseed + s];
            L::st(psi + chain * vstride + (size_t)BLD * atom, threadIdx.x, threadIdx.x, seed_coef[chain * nseed + s]);
        }
}
__global__ void k_set_identity(double2* m, size_t stride, const double* scale = nullptr) {
    double2* p = m + blockIdx.x * stride;
    const double sc = scale ? scale[blockIdx.x] : 1.0;
    for (int e = threadIdx.x; e < BLK; e += blockDim.x) p[e] = make_double2((e % NB) == (e / NB) ? sc : 0.0, 0.0);
}

// ------------------------------------------------------------------------------------------------
// Scalar Haydock recursion (nsp = 1): 18 orbital chains per site, vectors psi/pmn are (18, kk) per chain.
// hop :3310-3416 uses only the two spin-diagonal 9x9 sub-blocks of every H block.
// ------------------------------------------------------------------------------------------------
// v_i = sum_slots Hdiag(slot) psi_n ; pmn_i += v_i ; a += Re sum conj(psi_i) v_i        (one thread per (atom,row))
__global__ __launch_bounds__(NTHREADS) void k_scalar_hop(DevProblem P, ChainView CV, int level, const double2* __restrict__ psi, double2* pmn, double* partial) {
    __shared__ double red[NTHREADS];
    const int chain = blockIdx.y;
    const int a = threadIdx.x / NB, r = threadIdx.x % NB;
    const int count = CV.count_of(chain, level);
    const int* order = CV.order_of(chain, level);
    const size_t vo = (size_t)chain * CV.vstride;
    double asum = 0.0;
    for (int tile = blockIdx.x; tile * TILE_ATOMS < count; tile += gridDim.x) {
        const int t = tile * TILE_ATOMS + a;
        if (a >= TILE_ATOMS || t >= count) continue;
        const int i = order[t];
        const int* nb = P.nbr + (size_t)P.nslots * i;
        const int off = (r < 9) ? 0 : 9;
        double2 d = make_double2(0, 0);
        for (int slot = 0; slot < P.nslots; ++slot) {
            const int n = nb[slot];
            if (n < 0) continue;
            const double2* H = op_block(P, false, i, slot);
            const double2* x = psi + vo + (size_t)NB * n + off;
#pragma unroll
            for (int m = 0; m < 9; ++m) cfma(d, H[r + NB * (m + off)], x[m]);
            if (slot == 0 && !P.hoh) {
                // the scalar hop reads ee(:,:,1,ih) / hall(:,:,1,i) alone (:3336, :3372), without the spin-orbit block that the block
                // recursion's on-site term carries (locham = ee + lsham, :1608) and that the on-site table holds folded in: take it out
                // again (a no-op in the reference's own nsp = 1 runs, where lsham is never built and stays zero)
                const double2* S = P.lsham + (size_t)BLK * P.iz[i];
#pragma unroll
                for (int m = 0; m < 9; ++m) { const double2 sv = S[r + NB * (m + off)]; cfma(d, make_double2(-sv.x, -sv.y), x[m]); }
            }
        }
        const double2 p = psi[vo + (size_t)NB * i + r];
        asum += d.x * p.x + d.y * p.y;
        double2 q = pmn[vo + (size_t)NB * i + r];
        q.x += d.x; q.y += d.y;
        pmn[vo + (size_t)NB * i + r] = q;
    }
    red[threadIdx.x] = asum;
    __syncthreads();
    if (threadIdx.x == 0) {
        double s = 0.0;
        for (int e = 0; e < NTHREADS; ++e) s += red[e];
        partial[(size_t)chain * gridDim.x + blockIdx.x] = s;
    }
}
// pmn -= a psi ; b2 = |pmn|^2 partial
__global__ __launch_bounds__(NTHREADS) void k_scalar_orth(int kk, ChainView CV, int level, const double2* __restrict__ psi, double2* pmn, const double* __restrict__ acoef, size_t astride, int ll, double* partial) {
    __shared__ double red[NTHREADS];
    const int chain = blockIdx.y;
    const int count = CV.count_of(chain, level);
    const int* order = CV.order_of(chain, level);
    const size_t vo = (size_t)chain * CV.vstride;
    const double an = acoef[chain * astride + ll];
    double s2 = 0.0;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < count * NB; e += gridDim.x * blockDim.x) {
        const size_t idx = vo + (size_t)NB * order[e / NB] + (e % NB);
        double2 p = pmn[idx];
        const double2 x = psi[idx];
        p.x -= an * x.x; p.y -= an * x.y;
        pmn[idx] = p;
        s2 += p.x * p.x + p.y * p.y;
    }
    red[threadIdx.x] = s2;
    __syncthreads();
    if (threadIdx.x == 0) {
        double s = 0.0;
        for (int e = 0; e < NTHREADS; ++e) s += red[e];
        partial[(size_t)chain * gridDim.x + blockIdx.x] = s;
    }
}
// psi <- pmn/sqrt(b2) ; pmn <- -psi_old*sqrt(b2)
__global__ __launch_bounds__(NTHREADS) void k_scalar_update(int kk, ChainView CV, int level, double2* psi, double2* pmn, const double* __restrict__ b2coef, size_t bstride, int ll) {
    const int chain = blockIdx.y;
    const int count = CV.count_of(chain, level);
    const int* order = CV.order_of(chain, level);
    const size_t vo = (size_t)chain * CV.vstride;
    const double summ = b2coef[chain * bstride + ll + 1];
    const double sinv = 1.0 / sqrt(summ), sq = sqrt(summ);
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < count * NB; e += gridDim.x * blockDim.x) {
        const size_t idx = vo + (size_t)NB * order[e / NB] + (e % NB);
        const double2 p = pmn[idx], x = psi[idx];
        psi[idx] = make_double2(p.x * sinv, p.y * sinv);
        pmn[idx] = make_double2(-x.x * sq, -x.y * sq);
    }
}
// coef[chain][ll_out] = sum of per-block partial scalars (fixed order)
__global__ void k_scalar_reduce(const double* __restrict__ partial, int nblk, double* coef, size_t stride, int ll_out) {
    const int chain = blockIdx.x;
    if (threadIdx.x == 0) {
        double s = 0.0;
        for (int p = 0; p < nblk; ++p) s += partial[(size_t)chain * nblk + p];
        coef[chain * stride + ll_out] = s;
    }
}
__global__ void k_scalar_seed(double2* psi, size_t vstride, const int* seed_atom_orb /*[nchain][2]*/, double* b2coef, size_t bstride) {
    const int chain = blockIdx.x;
    if (threadIdx.x == 0) {
        psi[chain * vstride + (size_t)NB * seed_atom_orb[2 * chain] + seed_atom_orb[2 * chain + 1]] = make_double2(1.0, 0.0);
        b2coef[chain * bstride] = 1.0;
    }
}

}  // namespace rsrec
