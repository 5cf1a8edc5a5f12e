// Block Lanczos on UN-NORMALISED vectors ("u-scheme"): the post-hop pipeline with one kernel fewer.
//
// The reference (crecal_b, recursion.f90:1873-1973) keeps psi_n orthonormal: psi_{n+1} = pmn B_{n+1}^-1 is formed and
// stored at every level (:1963-1969).  Every operator on the path acts from the LEFT on the 18x18 blocks and every
// coefficient matrix from the RIGHT, so the normalisation commutes with H and can be carried as a small matrix instead
// of a pass over the vector:
//     u_n := pmn of level n-1  (psi_n = u_n Binv_n,  B_n^2 = sum u_n^H u_n,  u_1 = seed, Binv_1 = I)
//     t'   = H u_n                                                  (block SpMM, unchanged)
//     A_n  = psi_n^H H psi_n = Binv_n (sum u_n^H t') Binv_n           (k_mfma_adot + k_reduce_a_u)
//     u_{n+1} = H psi_n - psi_n A_n - psi_{n-1} B_n
//             = t' Binv_n  -  u_n (Binv_n A_n)  -  u_{n-1} (Binv_{n-1} B_n)     (k_mfma_orth3, three 36x36 right-multiplies)
//     B_{n+1}^2 = sum u_{n+1}^H u_{n+1}  -> eigen-decomposition -> B_{n+1}, Binv_{n+1}   (k_reduce_b_u)
// Per atom-step this reads t', u_n, u_{n-1} and writes u_{n+1} in place of u_{n-1}: 3 block reads + 1 write after the
// SpMM (+2 reads for A_n), against 4 reads + 2 writes (+2) for the normalised three-term form.  The coefficients A_n,
// B_n^2 are the same matrices up to rounding (checked against the reference's golden coefficients, tests/test_gpu_parity.py).
//
// Operand loads are 16 bytes per lane: the A-operand lane (row l15, k = l4) of MFMA k-step q < 8 takes column
// 8 (q >> 1) + 2 l4 + (q & 1), so two consecutive k-steps come from one double2; the fragment tables use the same k order.
#pragma once
#include "kernels_mfma.hpp"
#include "kernels_spmm5.hpp"

namespace rsrec {

__host__ __device__ constexpr int pk_col(int q, int l4) { return q < 8 ? 8 * (q >> 1) + 2 * l4 + (q & 1) : 32 + l4; }

// fragment table of G (18x18 complex, column-major) for [X_re | X_im] * Ghat with the paired k order
// ci: the vectors are in the CI layout (real column j of a row = (part j & 1, column j >> 1)); else LayoutRM (part j / 18, column j % 18)
__device__ __forceinline__ void emit_rhs_frags_pk(const double2* M, double sign, double* out, int ci_layout) {
    for (int e = threadIdx.x; e < 27 * 64; e += blockDim.x) {
        const int l = e & 63, qf = e >> 6, q = qf / 3, f = qf % 3;
        const int ki = pk_col(q, l >> 4);
        const int ko = (f < 2) ? 16 * f + (l & 15) : 32 + (l & 3);
        const int pi = ci_layout ? (ki & 1) : ki / 18, ci = ci_layout ? (ki >> 1) : ki % 18;
        const int po = ci_layout ? (ko & 1) : ko / 18, co = ci_layout ? (ko >> 1) : ko % 18;
        const double2 g = M[ci + 18 * co];
        const double v = (pi == po) ? g.x : (pi == 0 ? g.y : -g.y);
        out[e] = sign * v;
    }
}

// C = A * B for 18x18 complex column-major matrices in LDS (all threads of the block; caller syncs)
__device__ __forceinline__ void matmul18(const double2* A, const double2* B, double2* C) {
    for (int e = threadIdx.x; e < BLK; e += blockDim.x) {
        const int i = e % NB, j = e / NB;
        double2 acc = make_double2(0.0, 0.0);
        for (int k = 0; k < NB; ++k) {
            const double2 a = A[i + NB * k], b = B[k + NB * j];
            acc.x += a.x * b.x - a.y * b.y;
            acc.y += a.x * b.y + a.y * b.x;
        }
        C[e] = acc;
    }
}

struct U3Operands { double2 x[4]; double2 y; };           // y: columns 32 + 2 (l4 >> 1), + 1 (the lane uses member l4 & 1)

__device__ __forceinline__ void u3_load(U3Operands& o, const double* __restrict__ base, unsigned off, int l4) {
    const double2* p = reinterpret_cast<const double2*>(base + off + 2 * l4);
#pragma unroll
    for (int qq = 0; qq < 4; ++qq) o.x[qq] = ld_stream(p + 4 * qq);        // columns 8 qq + 2 l4, +1
    // the ninth k-step's column 32 + l4 as one more 16-byte load (all five loads of an operand set have the same shape: an 8-byte
    // load here made the register allocator recycle its destination pair as an address temporary -> a vmcnt(0) in the tile loop)
    o.y = ld_stream(reinterpret_cast<const double2*>(base + off + 32 + 2 * (l4 >> 1)));
}

template <int Q>
__device__ __forceinline__ double u3_k(const U3Operands& o, bool odd) { return Q < 8 ? ((Q & 1) ? o.x[Q >> 1].y : o.x[Q >> 1].x) : (odd ? o.y.y : o.y.x); }

__device__ __forceinline__ void u3_mac(double4_t& ca, double4_t& cb, double& cr, const U3Operands& o, const double (&T)[27], bool odd) {
#define U3_STEP(Q)                                                                              \
    {                                                                                           \
        const double a = u3_k<Q>(o, odd);                                                          \
        ca = __builtin_amdgcn_mfma_f64_16x16x4f64(a, T[3 * (Q) + 0], ca, 0, 0, 0);              \
        cb = __builtin_amdgcn_mfma_f64_16x16x4f64(a, T[3 * (Q) + 1], cb, 0, 0, 0);              \
        cr = __builtin_amdgcn_mfma_f64_4x4x4f64(a, T[3 * (Q) + 2], cr, 0, 0, 0);                \
    }
    U3_STEP(0) U3_STEP(1) U3_STEP(2) U3_STEP(3) U3_STEP(4) U3_STEP(5) U3_STEP(6) U3_STEP(7) U3_STEP(8)
#undef U3_STEP
}

// u_next = t' T1 + u_prev T2 + u_cur T3, written over u_prev; Gram partial of u_next.
// tabs[chain][3][27*64]: T1 = Binv_n, T2 = -Binv_{n-1} B_n, T3 = -Binv_n A_n (paired-k fragment tables).
// One wave per SIMD (the three tables live in registers); the operands of the next two row tiles are in flight while the
// current one is multiplied (~27 KB of reads per wave).  The kernel treats a block row as 36 reals in memory order: it serves
// LayoutRM and CI vectors alike (the tables carry the column meaning); 3 block reads + 1 block write per atom-step.
__global__ __launch_bounds__(MF_WAVES * 64, 1) void k_mfma_orth3(ChainView CV, int level, int zero_block, const double* __restrict__ tvec,
                                                                const double* __restrict__ ucur, double* uprev,
                                                                const double* __restrict__ tabs, double* partial /*[chain][nblk][1296]*/,
                                                                double* unext = nullptr /*where u_{n+1} goes (nullptr: over u_{n-1})*/) {
    __shared__ double lds[MF_WAVES * 1296];
    const int chain = blockIdx.y;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int ngroups = CV.count_of(chain, level) / GROUP;
    const int* order = CV.order_of(chain, level);
    const size_t vo = (size_t)chain * CV.vstride;
    const double* tv = tvec + vo;
    const double* uc = ucur + vo;
    double* up = uprev + vo;
    double* uo = (unext ? unext : uprev) + vo;
    const int l15 = lane & 15, l4 = lane >> 4, l3 = lane & 3, lg = (lane >> 2) & 3;
    double T1[27], T2[27], T3[27];
    {
        const double* f = tabs + (size_t)chain * 3 * 27 * 64 + lane;
#pragma unroll
        for (int e = 0; e < 27; ++e) { T1[e] = f[e * 64]; T2[e] = f[(27 + e) * 64]; T3[e] = f[(54 + e) * 64]; }
    }
    GramAcc Gm;
    Gm.zero();
    const int nbx = active_workgroups(ngroups);
    GroupWalk w((int)blockIdx.x < nbx ? ngroups : 0, wave, nbx);
    // row tiles of this wave, flattened: tile it = (group w.g + (it / 9) * step, rows 16 (it % 9) .. +15)
    const int ntile = (w.g < w.end) ? ((w.end - w.g + w.step - 1) / w.step) * 9 : 0;
    // The atoms of a row tile come from the order list through SCALAR loads: a tile of 16 stacked rows touches at most two
    // consecutive atoms of its group, (16 mt) / 18 and the next one, and the tile index is wave-uniform.  (A per-lane vector load of
    // the atom index put a dependent load + s_waitcnt vmcnt(0) in front of every operand prefetch and every store group: the
    // prefetched tiles were drained before the next ones were issued -- 44 % of the wave cycles in s_waitcnt.)
    struct TileAtoms { int s0, a_lo, a_hi; };
    auto tile_atoms = [&](int it) {
        it = min(it, ntile - 1);
        const int* __restrict__ grp = order + (size_t)(w.g + (it / 9) * w.step) * GROUP;
        TileAtoms t;
        t.s0 = (16 * (it % 9)) / 18;
        t.a_lo = grp[t.s0];
        t.a_hi = grp[min(t.s0 + 1, GROUP - 1)];
        return t;
    };
    auto tile_row = [&](const TileAtoms& t, int rho) {          // rho: stacked row inside the group, 16 mt <= rho < 16 mt + 16
        const bool hi = rho >= 18 * (t.s0 + 1);
        const int a = hi ? t.a_hi : t.a_lo;
        RowRef R;
        R.valid = a >= 0;
        R.off = (unsigned)BLD * (unsigned)(R.valid ? a : zero_block) + 36u * (unsigned)(rho - 18 * (t.s0 + (hi ? 1 : 0)));
        return R;
    };
    auto load_tile = [&](int it, U3Operands& a, U3Operands& c, U3Operands& p) {
        const TileAtoms t = tile_atoms(it);
        const RowRef ra = tile_row(t, 16 * (min(it, ntile - 1) % 9) + l15);
        u3_load(a, tv, ra.off, l4);
        u3_load(c, uc, ra.off, l4);
        u3_load(p, up, ra.off, l4);
    };
    // Three operand sets rotate through the roles (current, next, next-but-one) by NAME: the loop is unrolled three times, so a
    // set is never copied while its loads are in flight (a register copy of a load target waits for the load -- the rotating
    // copies of round 1 cut the effective prefetch distance to one tile: 38 % of the wave cycles sat in s_waitcnt,
    // profiles/r02b_c2_sq_pmc.txt).  Loads of tile it + 2 are issued before the MFMAs of tile it.
    // Software pipeline across row tiles: the stores and the Gram update of tile it - 1 are issued in the middle of tile it's
    // multiplications.  With one wave per SIMD nothing else covers the tail of a tile (the stores and the Gram MFMAs need the last
    // MFMA results, the Gram's third operand a cross-lane shuffle of them): run back to back that tail idled the matrix pipe for
    // about a third of each tile (profiles/r02i_c2_sq_pmc.txt: pipe 56 % busy, 68 % of the wave cycles in issue stalls).
    struct Pending { double4_t ca, cb; double cr; TileAtoms ta; int mt; };
    Pending P;
    bool pending = false;
    auto retire = [&](const double (&fr)[4]) {                 // stores + Gram update of the pending tile
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const RowRef rs = tile_row(P.ta, 16 * P.mt + l4 + 4 * j);
            if (rs.valid) { st_stream(uo + rs.off + l15, (double)P.ca[j]); st_stream(uo + rs.off + 16 + l15, (double)P.cb[j]); }
        }
        const RowRef rr = tile_row(P.ta, 16 * P.mt + 4 * lg + l4);
        if (rr.valid) st_stream(uo + rr.off + 32 + l3, P.cr);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const double f0 = P.ca[j], f1 = P.cb[j];
            Gm.t00 = __builtin_amdgcn_mfma_f64_16x16x4f64(f0, f0, Gm.t00, 0, 0, 0);
            Gm.t01 = __builtin_amdgcn_mfma_f64_16x16x4f64(f0, f1, Gm.t01, 0, 0, 0);
            Gm.t11 = __builtin_amdgcn_mfma_f64_16x16x4f64(f1, f1, Gm.t11, 0, 0, 0);
            Gm.tr0 = __builtin_amdgcn_mfma_f64_4x4x4f64(fr[j], f0, Gm.tr0, 0, 0, 0);
            Gm.tr1 = __builtin_amdgcn_mfma_f64_4x4x4f64(fr[j], f1, Gm.tr1, 0, 0, 0);
            Gm.trr = __builtin_amdgcn_mfma_f64_4x4x4f64(fr[j], fr[j], Gm.trr, 0, 0, 0);
        }
    };
    auto compute_tile = [&](int it, const U3Operands& ot, const U3Operands& oc, const U3Operands& op) {
        double fr[4] = {0.0, 0.0, 0.0, 0.0};
        if (pending) {
#pragma unroll
            for (int j = 0; j < 4; ++j) fr[j] = __shfl(P.cr, l3 + 4 * j + 16 * l4, 64);     // Y[row 4j + l4][32 + l3] of the pending tile
        }
        Pending N;
        N.ta = tile_atoms(it);
        N.mt = it % 9;
        N.ca = (double4_t){0, 0, 0, 0}; N.cb = (double4_t){0, 0, 0, 0}; N.cr = 0.0;
        u3_mac(N.ca, N.cb, N.cr, ot, T1, l4 & 1);
        if (pending) retire(fr);
        u3_mac(N.ca, N.cb, N.cr, op, T2, l4 & 1);
        u3_mac(N.ca, N.cb, N.cr, oc, T3, l4 & 1);
        P = N;
        pending = true;
    };
    U3Operands at, ac, ap, bt, bc, bp, ct, cc, cp;
    if (ntile > 0) { load_tile(0, at, ac, ap); load_tile(1, bt, bc, bp); }
#pragma unroll 1
    for (int it = 0; it < ntile; it += 3) {
        // (the scheduling fences keep every use of a freshly issued operand set -- even the lane select of its ninth k-step --
        //  below the fence of the tile that consumes it: hipcc otherwise hoists such uses to the load and drains the queue there)
        load_tile(it + 2, ct, cc, cp);
        __builtin_amdgcn_sched_barrier(0);
        compute_tile(it, at, ac, ap);
        __builtin_amdgcn_sched_barrier(0);
        if (it + 1 < ntile) {
            load_tile(it + 3, at, ac, ap);
            __builtin_amdgcn_sched_barrier(0);
            compute_tile(it + 1, bt, bc, bp);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (it + 2 < ntile) {
            load_tile(it + 4, bt, bc, bp);
            __builtin_amdgcn_sched_barrier(0);
            compute_tile(it + 2, ct, cc, cp);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    if (pending) {                                             // drain the pipeline: the last tile of this wave
        double fr[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) fr[j] = __shfl(P.cr, l3 + 4 * j + 16 * l4, 64);
        retire(fr);
    }
    gram_block_out(Gm, lds, partial + ((size_t)chain * gridDim.x + blockIdx.x) * 1296, true);
}

// ---- two waves per SIMD ------------------------------------------------------------------------------------------------
// The kernel above is bound by neither pipe: 11.6 GB per launch at 4.6 TB/s, matrix pipe 52 % busy, 73 % of the wave cycles in waits
// (profiles/r02t_c2_sq_pmc.txt) -- per 16-row tile the matrix work is 4848 cycles for 18.4 KB of traffic, i.e. the kernel needs
// 9.3 TB/s to keep the pipe full: it is HBM-bound, and ONE 512-register wave per SIMD has nothing to cover its own waits with.
// Here the three coefficient tables (81 doubles per lane = 162 registers) live in LDS instead (41 KB per workgroup, one ds_read_b64
// per MFMA: 23 B/clk per CU of the LDS's 128) and a wave fits 256 registers: two workgroups per CU, two waves per SIMD, each with the
// operands of the next row tile in flight.
template <int F>
__device__ __forceinline__ void u3_mac_lds(double4_t& ca, double4_t& cb, double& cr, const U3Operands& o, const double* __restrict__ T /*LDS: [27][64] + lane*/, bool odd) {
#define U3L_STEP(Q)                                                                             \
    {                                                                                           \
        const double a = u3_k<Q>(o, odd);                                                          \
        ca = __builtin_amdgcn_mfma_f64_16x16x4f64(a, T[(3 * (Q) + 0) * 64], ca, 0, 0, 0);       \
        cb = __builtin_amdgcn_mfma_f64_16x16x4f64(a, T[(3 * (Q) + 1) * 64], cb, 0, 0, 0);       \
        cr = __builtin_amdgcn_mfma_f64_4x4x4f64(a, T[(3 * (Q) + 2) * 64], cr, 0, 0, 0);         \
    }
    U3L_STEP(0) U3L_STEP(1) U3L_STEP(2) U3L_STEP(3) U3L_STEP(4) U3L_STEP(5) U3L_STEP(6) U3L_STEP(7) U3L_STEP(8)
#undef U3L_STEP
}

__global__ __launch_bounds__(MF_WAVES * 64, 2) void k_mfma_orth3w(ChainView CV, int level, int zero_block, const double* __restrict__ tvec,
                                                                 const double* __restrict__ ucur, double* uprev,
                                                                 const double* __restrict__ tabs, double* partial /*[chain][nblk][1296]*/) {
    __shared__ double lds[MF_WAVES * 1296];          // = 3 * 27 * 64: the three tables during the pass, the waves' Gram images at its end
    const int chain = blockIdx.y;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int ngroups = CV.count_of(chain, level) / GROUP;
    const int* order = CV.order_of(chain, level);
    const size_t vo = (size_t)chain * CV.vstride;
    const double* tv = tvec + vo;
    const double* uc = ucur + vo;
    double* up = uprev + vo;
    const int l15 = lane & 15, l4 = lane >> 4, l3 = lane & 3, lg = (lane >> 2) & 3;
    {
        const double* f = tabs + (size_t)chain * 3 * 27 * 64;
        for (int e = threadIdx.x; e < 3 * 27 * 64; e += MF_WAVES * 64) lds[e] = f[e];
        __syncthreads();
    }
    const double* T1 = lds + lane;
    const double* T2 = lds + 27 * 64 + lane;
    const double* T3 = lds + 54 * 64 + lane;
    GramAcc Gm;
    Gm.zero();
    const int nbx = active_workgroups(ngroups);
    GroupWalk w((int)blockIdx.x < nbx ? ngroups : 0, wave, nbx);
    const int ntile = (w.g < w.end) ? ((w.end - w.g + w.step - 1) / w.step) * 9 : 0;
    struct TileAtoms { int s0, a_lo, a_hi; };
    auto tile_atoms = [&](int it) {
        it = min(it, ntile - 1);
        const int* __restrict__ grp = order + (size_t)(w.g + (it / 9) * w.step) * GROUP;
        TileAtoms t;
        t.s0 = (16 * (it % 9)) / 18;
        t.a_lo = grp[t.s0];
        t.a_hi = grp[min(t.s0 + 1, GROUP - 1)];
        return t;
    };
    auto tile_row = [&](const TileAtoms& t, int rho) {
        const bool hi = rho >= 18 * (t.s0 + 1);
        const int a = hi ? t.a_hi : t.a_lo;
        RowRef R;
        R.valid = a >= 0;
        R.off = (unsigned)BLD * (unsigned)(R.valid ? a : zero_block) + 36u * (unsigned)(rho - 18 * (t.s0 + (hi ? 1 : 0)));
        return R;
    };
    auto load_tile = [&](int it, U3Operands& a, U3Operands& c, U3Operands& p) {
        const TileAtoms t = tile_atoms(it);
        const RowRef ra = tile_row(t, 16 * (min(it, ntile - 1) % 9) + l15);
        u3_load(a, tv, ra.off, l4);
        u3_load(c, uc, ra.off, l4);
        u3_load(p, up, ra.off, l4);
    };
    auto compute_tile = [&](int it, const U3Operands& ot, const U3Operands& oc, const U3Operands& op) {
        const TileAtoms ta = tile_atoms(it);
        const int mt = it % 9;
        double4_t ca = (double4_t){0, 0, 0, 0}, cb = (double4_t){0, 0, 0, 0};
        double cr = 0.0;
        u3_mac_lds<0>(ca, cb, cr, ot, T1, l4 & 1);
        u3_mac_lds<1>(ca, cb, cr, op, T2, l4 & 1);
        u3_mac_lds<2>(ca, cb, cr, oc, T3, l4 & 1);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const RowRef rs = tile_row(ta, 16 * mt + l4 + 4 * j);
            if (rs.valid) { up[rs.off + l15] = ca[j]; up[rs.off + 16 + l15] = cb[j]; }
        }
        const RowRef rr = tile_row(ta, 16 * mt + 4 * lg + l4);
        if (rr.valid) up[rr.off + 32 + l3] = cr;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const double fr = __shfl(cr, l3 + 4 * j + 16 * l4, 64);      // Y[row 4 j + l4][32 + l3]
            const double f0 = ca[j], f1 = cb[j];
            Gm.t00 = __builtin_amdgcn_mfma_f64_16x16x4f64(f0, f0, Gm.t00, 0, 0, 0);
            Gm.t01 = __builtin_amdgcn_mfma_f64_16x16x4f64(f0, f1, Gm.t01, 0, 0, 0);
            Gm.t11 = __builtin_amdgcn_mfma_f64_16x16x4f64(f1, f1, Gm.t11, 0, 0, 0);
            Gm.tr0 = __builtin_amdgcn_mfma_f64_4x4x4f64(fr, f0, Gm.tr0, 0, 0, 0);
            Gm.tr1 = __builtin_amdgcn_mfma_f64_4x4x4f64(fr, f1, Gm.tr1, 0, 0, 0);
            Gm.trr = __builtin_amdgcn_mfma_f64_4x4x4f64(fr, fr, Gm.trr, 0, 0, 0);
        }
    };
    U3Operands at, ac, ap, bt, bc, bp;
    if (ntile > 0) load_tile(0, at, ac, ap);
#pragma unroll 1
    for (int it = 0; it < ntile; it += 2) {
        load_tile(it + 1, bt, bc, bp);
        __builtin_amdgcn_sched_barrier(0);
        compute_tile(it, at, ac, ap);
        __builtin_amdgcn_sched_barrier(0);
        if (it + 1 < ntile) {
            load_tile(it + 2, at, ac, ap);
            __builtin_amdgcn_sched_barrier(0);
            compute_tile(it + 1, bt, bc, bp);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    __syncthreads();                                   // every wave is done with the tables: the buffer becomes the Gram staging area
    gram_block_out(Gm, lds, partial + ((size_t)chain * gridDim.x + blockIdx.x) * 1296, true);
}

// G = sum u_n^H t'  ->  A_n = Binv_n G Binv_n (the coefficient, recursion.f90:1642), T3 = -Binv_n A_n
__global__ __launch_bounds__(1024) void k_reduce_a_u(const double* __restrict__ partial, int nblk, double2* a_out, size_t astride,
                                                    const double2* __restrict__ Bmats /*[chain][2][324]: B_n, Binv_n*/, double* tabs, int ci) {
    __shared__ double lds[1296];
    __shared__ double2 Gm[BLK], Bi[BLK], M1[BLK], M2[BLK];
    const int chain = blockIdx.x;
    const double2 c = reduce_gram(partial + (size_t)chain * nblk * 1296, nblk, lds, ci);
    if (threadIdx.x < BLK) { Gm[threadIdx.x] = c; Bi[threadIdx.x] = Bmats[(size_t)chain * 2 * BLK + BLK + threadIdx.x]; }
    __syncthreads();
    matmul18(Gm, Bi, M1);            // G Binv
    __syncthreads();
    matmul18(Bi, M1, M2);            // A = Binv G Binv
    __syncthreads();
    if (threadIdx.x < BLK) a_out[chain * astride + threadIdx.x] = M2[threadIdx.x];
    matmul18(Bi, M2, M1);            // Binv A
    __syncthreads();
    emit_rhs_frags_pk(M1, -1.0, tabs + ((size_t)chain * 3 + 2) * 27 * 64, ci);
}

// B_{n+1}^2 = sum u_{n+1}^H u_{n+1} (b2_b, recursion.f90:1931) -> B_{n+1}, Binv_{n+1} (:1937-1960);
// tables for the next level: T1 = Binv_{n+1}, T2 = -Binv_n B_{n+1}
__global__ __launch_bounds__(1024) void k_reduce_b_u(const double* __restrict__ partial, int nblk, double2* b2_out, size_t bstride, double2* Bmats,
                                                    double* tabs, int* status, int ci) {
    __shared__ double lds[1296];
    __shared__ Eig18Shared sh;
    __shared__ double2 Bn[BLK], Bin[BLK], Bold[BLK], M1[BLK];
    const int chain = blockIdx.x;
    const double2 c = reduce_gram(partial + (size_t)chain * nblk * 1296, nblk, lds, ci);
    double2* Bout = Bmats + (size_t)chain * 2 * BLK;
    if (threadIdx.x < BLK) { b2_out[chain * bstride + threadIdx.x] = c; sh.A[threadIdx.x] = c; Bold[threadIdx.x] = Bout[BLK + threadIdx.x]; }
    __syncthreads();
    const int sw = jacobi18(sh);
    if (sw < 0 && threadIdx.x == 0) atomicOr(status, 1);
    if (threadIdx.x < NB) { const double l = sqrt(sh.ev[threadIdx.x]); sh.f1[threadIdx.x] = l; sh.f2[threadIdx.x] = 1.0 / l; }
    __syncthreads();
    matfun18(sh, sh.f1, Bn);
    matfun18(sh, sh.f2, Bin);
    __syncthreads();
    matmul18(Bold, Bn, M1);          // Binv_n B_{n+1}
    for (int e = threadIdx.x; e < BLK; e += blockDim.x) { Bout[e] = Bn[e]; Bout[BLK + e] = Bin[e]; }
    __syncthreads();
    emit_rhs_frags_pk(Bin, 1.0, tabs + ((size_t)chain * 3 + 0) * 27 * 64, ci);
    emit_rhs_frags_pk(M1, -1.0, tabs + ((size_t)chain * 3 + 1) * 27 * 64, ci);
}

// initial state of a chain: B_1 = Binv_1 = I, T1 = I, T2 = T3 = 0
__global__ void k_uscheme_init(double2* Bmats, double* tabs, int ci) {
    __shared__ double2 I[BLK];
    const int chain = blockIdx.x;
    for (int e = threadIdx.x; e < BLK; e += blockDim.x) I[e] = make_double2((e % NB) == (e / NB) ? 1.0 : 0.0, 0.0);
    __syncthreads();
    for (int e = threadIdx.x; e < 2 * BLK; e += blockDim.x) Bmats[(size_t)chain * 2 * BLK + e] = I[e % BLK];
    emit_rhs_frags_pk(I, 1.0, tabs + ((size_t)chain * 3 + 0) * 27 * 64, ci);
    for (int e = threadIdx.x; e < 2 * 27 * 64; e += blockDim.x) tabs[((size_t)chain * 3 + 1) * 27 * 64 + e] = 0.0;
}


// ======================================================================================================================
// Chebyshev step on the matrix cores (chebyshev_recur_ll, recursion.f90:2495-2597; cheb_1st_mom :2169-2238).
// The SpMM kernel leaves t = H psi1 in a vector; this kernel does everything after it in ONE pass over the active blocks:
//   FIRST:  psi1 = (t - b psi0) / a                      ; G2 = sum psi0^H psi1                       (:2228-2236)
//   else :  psi2 = ((t - b psi1) / a) * 2 - psi0         ; G1 = sum psi1^H psi1, G2 = sum psi2^H psi1 (:2548-2587)
// in the reference's operation order, element-wise on rows of 36 reals: LayoutRM and CI vectors alike.  The 18x18 complex
// reductions are real 36x36 Gram matrices with the stacked rows as MFMA K dimension (same construction as k_mfma_adot);
// partial[chain][workgroup][2][1296].
// ======================================================================================================================
// FUSED: the element-wise step was done by the SpMM's epilogue (S5Epilogue): `out_all` already holds the new vector, only the Gram
// matrices are formed here, in one pass over cur and out.
#ifndef CHEB_UNROLL
#define CHEB_UNROLL 4      // four k-steps of 4 stacked rows in flight (2: 1.31 ms per launch at 64 x 22^3, 4: 1.07 ms; 6: no further gain)
#endif
template <bool FIRST, bool FUSED = false>
__global__ __launch_bounds__(MF_WAVES * 64, 2) void k_mfma_cheb(ChainView CV, int level, int zero_block, const double* __restrict__ tvec,
                                                               const double* __restrict__ cur_all, const double* __restrict__ old_all,
                                                               double* __restrict__ out_all, double a, double b, double* partial) {
    __shared__ double lds[MF_WAVES * 1296];
    const int chain = blockIdx.y;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int ngroups = CV.count_of(chain, level) / GROUP;
    const int* order = CV.order_of(chain, level);
    const size_t vo = (size_t)chain * CV.vstride;
    const double* tv = FUSED ? nullptr : tvec + vo;
    const double* cu = cur_all + vo;
    const double* ol = (FIRST || FUSED) ? nullptr : old_all + vo;
    double* out = out_all + vo;
    const int l15 = lane & 15, l4 = lane >> 4, l3 = lane & 3, lg = (lane >> 2) & 3;
    GramAcc G1, G2;
    G1.zero(); G2.zero();
    const int nbx = active_workgroups(ngroups);
    for (GroupWalk w((int)blockIdx.x < nbx ? ngroups : 0, wave, nbx); w.g < w.end; w.g += w.step) {
        const int* grp = order + (size_t)w.g * GROUP;
#pragma unroll CHEB_UNROLL
        for (int kq = 0; kq < 36; ++kq) {
            const RowRef rk = group_row(grp, 4 * kq + l4, zero_block);
            const double c0 = cu[rk.off + l15], c1 = cu[rk.off + 16 + l15], cr = cu[rk.off + 32 + l3];
            double n0, n1, nr;
            if (FUSED) {
                n0 = out[rk.off + l15]; n1 = out[rk.off + 16 + l15]; nr = out[rk.off + 32 + l3];
            } else {
                const double t0 = tv[rk.off + l15], t1 = tv[rk.off + 16 + l15], tr = tv[rk.off + 32 + l3];
                n0 = (t0 - b * c0) / a; n1 = (t1 - b * c1) / a; nr = (tr - b * cr) / a;
                if (!FIRST) {
                    const double z0 = ol[rk.off + l15], z1 = ol[rk.off + 16 + l15], zr = ol[rk.off + 32 + l3];
                    n0 = n0 * 2.0 - z0; n1 = n1 * 2.0 - z1; nr = nr * 2.0 - zr;
                }
                if (rk.valid) {
                    out[rk.off + l15] = n0; out[rk.off + 16 + l15] = n1;
                    if (lg == 0) out[rk.off + 32 + l3] = nr;
                }
            }
            if (FIRST) {          // G2 = psi0hat^T psi1hat
                G2.t00 = __builtin_amdgcn_mfma_f64_16x16x4f64(c0, n0, G2.t00, 0, 0, 0);
                G2.t01 = __builtin_amdgcn_mfma_f64_16x16x4f64(c0, n1, G2.t01, 0, 0, 0);
                G2.t10 = __builtin_amdgcn_mfma_f64_16x16x4f64(c1, n0, G2.t10, 0, 0, 0);
                G2.t11 = __builtin_amdgcn_mfma_f64_16x16x4f64(c1, n1, G2.t11, 0, 0, 0);
                G2.tr0 = __builtin_amdgcn_mfma_f64_4x4x4f64(cr, n0, G2.tr0, 0, 0, 0);
                G2.tr1 = __builtin_amdgcn_mfma_f64_4x4x4f64(cr, n1, G2.tr1, 0, 0, 0);
                G2.t0r = __builtin_amdgcn_mfma_f64_4x4x4f64(c0, nr, G2.t0r, 0, 0, 0);
                G2.t1r = __builtin_amdgcn_mfma_f64_4x4x4f64(c1, nr, G2.t1r, 0, 0, 0);
                G2.trr = __builtin_amdgcn_mfma_f64_4x4x4f64(cr, nr, G2.trr, 0, 0, 0);
            } else {              // G1 = psi1hat^T psi1hat (symmetric half), G2 = psi2hat^T psi1hat
                G1.t00 = __builtin_amdgcn_mfma_f64_16x16x4f64(c0, c0, G1.t00, 0, 0, 0);
                G1.t01 = __builtin_amdgcn_mfma_f64_16x16x4f64(c0, c1, G1.t01, 0, 0, 0);
                G1.t11 = __builtin_amdgcn_mfma_f64_16x16x4f64(c1, c1, G1.t11, 0, 0, 0);
                G1.tr0 = __builtin_amdgcn_mfma_f64_4x4x4f64(cr, c0, G1.tr0, 0, 0, 0);
                G1.tr1 = __builtin_amdgcn_mfma_f64_4x4x4f64(cr, c1, G1.tr1, 0, 0, 0);
                G1.trr = __builtin_amdgcn_mfma_f64_4x4x4f64(cr, cr, G1.trr, 0, 0, 0);
                G2.t00 = __builtin_amdgcn_mfma_f64_16x16x4f64(n0, c0, G2.t00, 0, 0, 0);
                G2.t01 = __builtin_amdgcn_mfma_f64_16x16x4f64(n0, c1, G2.t01, 0, 0, 0);
                G2.t10 = __builtin_amdgcn_mfma_f64_16x16x4f64(n1, c0, G2.t10, 0, 0, 0);
                G2.t11 = __builtin_amdgcn_mfma_f64_16x16x4f64(n1, c1, G2.t11, 0, 0, 0);
                G2.tr0 = __builtin_amdgcn_mfma_f64_4x4x4f64(nr, c0, G2.tr0, 0, 0, 0);
                G2.tr1 = __builtin_amdgcn_mfma_f64_4x4x4f64(nr, c1, G2.tr1, 0, 0, 0);
                G2.t0r = __builtin_amdgcn_mfma_f64_4x4x4f64(n0, cr, G2.t0r, 0, 0, 0);
                G2.t1r = __builtin_amdgcn_mfma_f64_4x4x4f64(n1, cr, G2.t1r, 0, 0, 0);
                G2.trr = __builtin_amdgcn_mfma_f64_4x4x4f64(nr, cr, G2.trr, 0, 0, 0);
            }
        }
    }
    double* pout = partial + ((size_t)chain * gridDim.x + blockIdx.x) * 2 * 1296;
    gram_block_out(G1, lds, pout, true);
    __syncthreads();
    gram_block_out(G2, lds, pout + 1296, false);
}

// 36x36 Gram partials of k_mfma_cheb -> moments.  first: mu[1] = sum psi0^H psi1.  else: mu[2ll] = 2 d1 - mu[0],
// mu[2ll+1] = 2 d2 - mu[1] (0-based moment index; recursion.f90:2591-2592) and the divergence test of :2594 / :2484.
__global__ __launch_bounds__(1024) void k_reduce_cheb_mf(const double* __restrict__ partial, int nblk, int first, int ll, double2* mu, size_t mustride,
                                                        int* status, int check_both, int ci) {
    __shared__ double img[2][1296];
    __shared__ double tr[2][BLK];
    const int chain = blockIdx.x, tid = threadIdx.x;
    const double* P = partial + (size_t)chain * nblk * 2 * 1296;
    for (int e = tid; e < 2 * 1296; e += blockDim.x) {
        const int which = e / 1296, k = e % 1296;
        double s = 0.0;
        for (int p = 0; p < nblk; ++p) s += P[((size_t)p * 2 + which) * 1296 + k];
        img[which][k] = s;
    }
    __syncthreads();
    double2* m = mu + chain * mustride;
    if (tid < BLK) {
        const int cp = tid % NB, cc = tid / NB;
        const int rp = gram_col(0, cp, ci), ip = gram_col(1, cp, ci), rc = gram_col(0, cc, ci), ic = gram_col(1, cc, ci);
        const double2 d1 = make_double2(img[0][36 * rp + rc] + img[0][36 * ip + ic], img[0][36 * rp + ic] - img[0][36 * ip + rc]);
        const double2 d2 = make_double2(img[1][36 * rp + rc] + img[1][36 * ip + ic], img[1][36 * rp + ic] - img[1][36 * ip + rc]);
        if (first) {
            m[BLK + tid] = d2;
            tr[0][tid] = 0.0; tr[1][tid] = 0.0;
        } else {
            const double2 m0 = m[tid], m1 = m[BLK + tid];
            const double2 o1 = make_double2(2.0 * d1.x - m0.x, 2.0 * d1.y - m0.y);
            const double2 o2 = make_double2(2.0 * d2.x - m1.x, 2.0 * d2.y - m1.y);
            m[(size_t)(2 * ll) * BLK + tid] = o1;
            m[(size_t)(2 * ll + 1) * BLK + tid] = o2;
            tr[0][tid] = o2.x; tr[1][tid] = o1.x;
        }
    }
    __syncthreads();
    if (tid == 0 && !first) {
        double s = 0.0, s1 = 0.0;
        for (int e = 0; e < BLK; ++e) { s += tr[0][e]; s1 += tr[1][e]; }
        if (s > 1000.0 || (check_both && s1 > 1000.0)) atomicOr(status, 2);
    }
}

// First stage of the partial-sum reductions: out[chain][b][:] = sum of in[chain][16 b .. 16 b + 15][:] in index order.  The
// 36x36 Gram partials of up to 256 workgroups per chain (2.6 MB) were summed by ONE workgroup per chain in the reduce kernels:
// at a single chain that is one CU's L2 bandwidth, 40-100 us per level and half of the single-site recursion.  Fixed grouping
// (16) and fixed order keep the result independent of launch width and batch composition (trailing all-zero partials add +0).
// grid = (nblk2 * ceil(width / 256), chains): one output element per thread, its 16 addends in flight together
__global__ __launch_bounds__(256) void k_presum16(const double* __restrict__ in, int nblk, int nblk2, int width, double* __restrict__ out) {
    const int nchunk = (width + 255) >> 8;
    const int b = blockIdx.x / nchunk, e = (blockIdx.x % nchunk) * 256 + threadIdx.x, chain = blockIdx.y;
    if (e >= width) return;
    const int p0 = 16 * b, p1 = min(nblk, p0 + 16);
    const double* src = in + (size_t)chain * nblk * width + e;
    double v[16];
#pragma unroll
    for (int p = 0; p < 16; ++p) v[p] = (p0 + p < p1) ? src[(size_t)(p0 + p) * width] : 0.0;
    double s = 0.0;
#pragma unroll
    for (int p = 0; p < 16; ++p) s += v[p];                  // fixed order; missing trailing partials add +0
    out[((size_t)chain * nblk2 + b) * width + e] = s;
}

}  // namespace rsrec
