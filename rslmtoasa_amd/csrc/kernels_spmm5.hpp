// Block-sparse H|psi> for large launches ("spmm5"): spin-split waves, 16-byte operand loads, two waves per SIMD.
//
// out_i = sum_slots H_slot in_{nbr(i,slot)}   (hop_b, recursion.f90:1576-1625; hop_b_hoh :1411-1552 as two such passes)
//
// How it got here (profiles/ of both rounds; DESIGN.md has the numbers):
//   * k_spmm4 (one 500-register wave per SIMD, 8-byte loads): matrix pipe 51 % busy, 65 % of the wave cycles stalled at issue;
//     with the operand LOADS removed (timing probe) it ran 1.8x faster wherever the data came from -- load instructions, not
//     bytes, were the cost.
//   * spin-split waves: rows of the real form are ordered spin-major and each spin padded from 18 to 20 rows, so the collinear
//     operators (no spin-flip hopping; hamiltonian.f90:1553-1617) are exactly block diagonal at MFMA granularity and ONE WAVE
//     OWNS ONE OUTPUT SPIN of a group of 8 atoms: 90 accumulator registers, two waves per SIMD cover each other's stalls.
//     Spin-mixing blocks (spin-orbit on-site term, non-collinear operators) take the same path with both input spins.
//   * CI vector layout (kernels_mfma.hpp) with the real form's k order chosen so that the real and imaginary part of one vector
//     element are the B operands of two consecutive k-steps: one 16-byte load feeds two MFMA k-steps; the operator fragments
//     are stored the same way.  The same layout is written back (16-byte stores) and read by the Gram / orthogonalisation
//     kernels: a vector exists once.
//   * "wide" rows: the first 16 real-form rows of a spin go through ONE v_mfma_f64_16x16x4 per tile and k-step instead of four
//     4x4x4 row blocks: the operator arrives in 2 operand registers per k-step instead of 5 (rows 16..19 keep a 4x4x4), 33
//     instead of 42 operand loads and 90 instead of 225 MFMA instructions per neighbour slot (-4 % although the 16x16x4
//     instruction sustains 66 TF against the 4x4x4's 75, profiles/ubench_f64_r01.txt).
//   * operand prefetch TWO steps ahead with the same three register sets (-4 %): a set is reloaded right after the step that
//     consumed it; one step (1440 matrix cycles) did not cover gathers that miss L2.
//   * neighbour blocks are wave-uniform: their addresses are SGPR bases, the lane part of every address is one of three
//     loop-invariant registers, everything else an instruction immediate: no address arithmetic on the vector ALU.
// Tried on top of this and not adopted: a wave walking several groups with the next group's first operands requested during the
// last entry of the current one (the group prologue -- a chain of dependent loads -- then overlaps matrix work): 1.5 % / 4 % / 7 %
// SLOWER at 2 / 3 / 4 groups per wave; many short one-group workgroups that the hardware dispatcher balances win.
#pragma once
#include <hip/hip_runtime.h>
#include <algorithm>
#include <vector>
#include "kernels_valu.hpp"
#include "kernels_mfma.hpp"
#include "kernels_spmm4.hpp"

namespace rsrec {

#ifndef S5_WG_GROUPS
#define S5_WG_GROUPS 4   // groups of 8 atoms per workgroup (x 2 spin waves each): 4 -> 512 threads
#endif
constexpr int S5_FRAG_PER_PART = 320;                          // doubles: k-pair 0 (128), k-pair 1 (128), single k-step (64)
constexpr int S5_FRAG_PER_SLOT = 2 * 2 * 2 * S5_FRAG_PER_PART;   // [sigma_out][sigma_in][rows 0..15 | rows 16..19][320]

struct Spmm5Operator {
    double* d_frag = nullptr;    // [set][tau][slot 0..nslots][sigma_out][sigma_in][2][320]
    int* d_meta = nullptr;       // [set][tau][1 + 2 S4_MAXSLOTS]: count, then entries slot | flip << 8 (flip: input spin = the other spin)
    size_t frag_bytes = 0, meta_bytes = 0;
    int ntau = 0, nslots = 0, have_o = 0;
    static constexpr int META = 1 + 2 * S4_MAXSLOTS;

    void release() {
        if (d_frag) (void)hipFree(d_frag);
        if (d_meta) (void)hipFree(d_meta);
        d_frag = nullptr; d_meta = nullptr; frag_bytes = meta_bytes = 0;
    }
    // entry (row ko, column ki) of the padded 40x40 real form: index = 20 sigma + rho.  rho = 4 s + l (k-step s, lane row l):
    // s = 2 P + e < 4 -> (part e, m = 4 P + l); s = 4 -> l = 0: (re, m = 8), l = 1: (im, m = 8), l = 2, 3: padding -- the two
    // members of a k-pair are the real and imaginary part of ONE complex element of the vector (CI layout)
    static bool decode20(int rho, int& part, int& m) {
        const int st = rho >> 2, l = rho & 3;
        if (st < 4) { part = st & 1; m = 4 * (st >> 1) + l; return true; }
        if (l < 2) { part = l; m = 8; return true; }
        return false;
    }
    static double real40(const double* blk, int ko, int ki) {
        const int so = ko / 20, wo = ko % 20, si = ki / 20, wi = ki % 20;
        int po, mo, pi, mi;
        if (!decode20(wo, po, mo) || !decode20(wi, pi, mi)) return 0.0;
        const int ro = 9 * so + mo, ri = 9 * si + mi;
        const double hr = blk[2 * (ro + 18 * ri)], hi = blk[2 * (ro + 18 * ri) + 1];
        if (po == pi) return hr;
        return po == 0 ? -hi : hi;
    }
    // one complex block (column-major interleaved) -> fragments: A operand of the 16x16x4 MFMA for rows 0..15 (lane (l15 = row,
    // l4 = k)) and of the 4x4x4 MFMA for rows 16..19 (lane (i + 4 g + 16 k) = A[16 + i][k], same for the 4 blocks g)
    static void swizzle(const double* blk, double* out) {
        for (int so = 0; so < 2; ++so)
            for (int si = 0; si < 2; ++si) {
                double* o = out + (so * 2 + si) * 2 * S5_FRAG_PER_PART;
                for (int l = 0; l < 64; ++l) {
                    const int k = l >> 4;
                    for (int part = 0; part < 2; ++part) {
                        const int ko = 20 * so + (part == 0 ? (l & 15) : 16 + (l & 3));
                        double* q = o + part * S5_FRAG_PER_PART;
                        for (int p = 0; p < 2; ++p)
                            for (int e = 0; e < 2; ++e) q[128 * p + 2 * l + e] = real40(blk, ko, 20 * si + 8 * p + 4 * e + k);
                        q[256 + l] = real40(blk, ko, 20 * si + 16 + k);
                    }
                }
            }
    }
    // General table: blk[(set * ntau + tau) * (nslots + 1) + s] = column-major interleaved 18x18 complex block of operator class tau,
    // fragment slot s (s = nslots: the extra on-site slot that reads the second input vector), or nullptr = absent (contributes
    // nothing and is not scheduled).  Schedule: the spin-diagonal part of every block, plus the spin-flip part of blocks that have one.
    const char* build_custom(int nslots_lat, int ntau_, int nset, const std::vector<const double*>& blk) {
        if (nslots_lat + 1 > S4_MAXSLOTS) return "too many neighbour slots for the spmm5 kernel";
        ntau = ntau_; nslots = nslots_lat; have_o = nset > 1 ? 1 : 0;
        const int nfs = nslots + 1;
        const size_t per_set = (size_t)ntau * nfs * S5_FRAG_PER_SLOT;
        std::vector<double> host(per_set * nset, 0.0);
        std::vector<int> meta((size_t)nset * ntau * META, 0);
        for (int set = 0; set < nset; ++set)
            for (int tau = 0; tau < ntau; ++tau) {
                int* M = meta.data() + ((size_t)set * ntau + tau) * META;
                for (int s = 0; s < nfs; ++s) {
                    const double* src = blk[((size_t)set * ntau + tau) * nfs + s];
                    if (!src) continue;
                    swizzle(src, host.data() + set * per_set + ((size_t)tau * nfs + s) * S5_FRAG_PER_SLOT);
                    M[1 + M[0]] = s; M[0]++;
                    if (Spmm4Operator::pattern_of(src) == 0) { M[1 + M[0]] = s | (1 << 8); M[0]++; }
                }
            }
        const size_t need = host.size() * sizeof(double), mneed = meta.size() * sizeof(int);
        if (need > frag_bytes) {
            if (d_frag) (void)hipFree(d_frag);
            d_frag = nullptr; frag_bytes = 0;
            if (hipMalloc(reinterpret_cast<void**>(&d_frag), need) != hipSuccess) return "hipMalloc of spmm5 operator fragments failed";
            frag_bytes = need;
        }
        if (mneed > meta_bytes) {
            if (d_meta) (void)hipFree(d_meta);
            d_meta = nullptr; meta_bytes = 0;
            if (hipMalloc(reinterpret_cast<void**>(&d_meta), mneed) != hipSuccess) return "hipMalloc of spmm5 schedule failed";
            meta_bytes = mneed;
        }
        if (hipMemcpy(d_frag, host.data(), need, hipMemcpyHostToDevice) != hipSuccess) return "upload of spmm5 fragments failed";
        if (hipMemcpy(d_meta, meta.data(), mneed, hipMemcpyHostToDevice) != hipSuccess) return "upload of spmm5 schedule failed";
        return nullptr;
    }
    // The Hamiltonian itself.  Set 0: the blocks of h (slot 0 carries + l.s when !hoh).  Set 1 (hoh second pass) is built so that
    // ONE SpMM pass over hpsi = h psi plus one extra on-site slot reading psi gives the whole
    //   H psi = hpsi - (h o) hpsi + (e_nu + l.s) psi   (recursion.f90:1543):
    // slot 0 -> 1 - (h o)_0,  slot s -> -(h o)_s,  slot `nslots` (extra) -> enim + lsham of the atom's type.
    const char* build(int nslots_lat, int hstride, int ntype, int nmax, int hoh, const double* st, const double* loc, const double* eeo, const double* hallo,
                      const double* enim, const double* lsham, const int* iz0) {
        const int nt = nmax + ntype, nset = hoh ? 2 : 1, nfs = nslots_lat + 1;
        const size_t B = 2 * (size_t)BLK;
        std::vector<const double*> blk((size_t)nset * nt * nfs, nullptr);
        std::vector<double> tmp((size_t)(hoh ? nt * nfs : 0) * B, 0.0);
        for (int tau = 0; tau < nt; ++tau)
            for (int s = 0; s < nslots_lat; ++s) {
                const double* h0 = tau < nmax ? loc + B * (s + (size_t)hstride * tau) : st + B * (s + (size_t)hstride * (tau - nmax));
                blk[(size_t)tau * nfs + s] = h0;
                if (hoh) {
                    const double* ho = tau < nmax ? hallo + B * (s + (size_t)hstride * tau) : eeo + B * (s + (size_t)hstride * (tau - nmax));
                    double* d = tmp.data() + B * ((size_t)tau * nfs + s);
                    for (size_t e = 0; e < B; ++e) d[e] = -ho[e];
                    if (s == 0) for (int q = 0; q < NB; ++q) d[2 * (q + NB * q)] += 1.0;
                    blk[((size_t)nt + tau) * nfs + s] = d;
                }
            }
        if (hoh)
            for (int tau = 0; tau < nt; ++tau) {
                const int ty = tau < nmax ? iz0[tau] : tau - nmax;
                double* d = tmp.data() + B * ((size_t)tau * nfs + nslots_lat);
                for (size_t e = 0; e < B; ++e) d[e] = enim[B * ty + e] + lsham[B * ty + e];
                blk[((size_t)nt + tau) * nfs + nslots_lat] = d;
            }
        return build_custom(nslots_lat, nt, nset, blk);
    }
    const double* frag_set(int set) const { return d_frag + (size_t)set * ntau * (nslots + 1) * S5_FRAG_PER_SLOT; }
    const int* meta_set(int set) const { return d_meta + (size_t)set * ntau * META; }
};

typedef double s5_d2 __attribute__((ext_vector_type(2)));
struct S5Pair { s5_d2 b[9]; s5_d2 a[2]; };      // operands of two k-steps: nine psi tiles; operator rows 0..15 and 16..19
struct S5Single { double b[9]; double a[2]; };  // the spin's fifth k-step (row m = 8: re, im + padding)
struct S5Acc { double4_t m[9]; double r[9]; };  // per tile: 16x16x4 result (rows 0..15) and 4x4x4 result (rows 16..19)

// wave-uniform addressing state of one neighbour slot
struct S5Slot {
    const char* tile[GROUP];   // block of the neighbour of atom t (spin 0 half)
    const char* base;          // vector the slot reads (the second input for the extra on-site slot)
    unsigned rem;              // remainder tile: byte offset of this lane's neighbour block (per lane: atom l15 >> 1)
};

template <int P>
__device__ __forceinline__ void s5_load_pair(S5Pair& o, const S5Slot& S, unsigned spin_off, const char* __restrict__ fb,
                                             unsigned lane_main, unsigned lane_rem, unsigned lane16) {
#pragma unroll
    for (int t = 0; t < GROUP; ++t) o.b[t] = *reinterpret_cast<const s5_d2*>(S.tile[t] + spin_off + lane_main + 1152 * P);
    o.b[8] = *reinterpret_cast<const s5_d2*>(S.base + spin_off + (S.rem + lane_rem) + 1152 * P);
#pragma unroll
    for (int q = 0; q < 2; ++q) o.a[q] = *reinterpret_cast<const s5_d2*>(fb + lane16 + (q * S5_FRAG_PER_PART * 8 + 1024 * P));
}
__device__ __forceinline__ void s5_load_single(S5Single& o, const S5Slot& S, unsigned spin_off, const char* __restrict__ fb,
                                               unsigned lane_single, unsigned lane_rem_single, unsigned lane8) {
#pragma unroll
    for (int t = 0; t < GROUP; ++t) o.b[t] = *reinterpret_cast<const double*>(S.tile[t] + spin_off + lane_single);
    o.b[8] = *reinterpret_cast<const double*>(S.base + spin_off + (S.rem + lane_rem_single));
#pragma unroll
    for (int q = 0; q < 2; ++q) o.a[q] = *reinterpret_cast<const double*>(fb + lane8 + (q * S5_FRAG_PER_PART * 8 + 2048));
}

// issue order: one operand load, then PER MFMAs
template <int NL, int NM>
__device__ __forceinline__ void s5_interleave() {
    constexpr int PER = NM / NL;
#pragma unroll
    for (int i = 0; i < NL; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x20, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x8, PER, 0);
    }
    __builtin_amdgcn_sched_group_barrier(0x8, NM - PER * NL, 0);
}

__device__ __forceinline__ void s5_mfma_pair(S5Acc& acc, const S5Pair& o) {
#pragma unroll
    for (int e = 0; e < 2; ++e)
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            acc.m[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(o.a[0][e], o.b[t][e], acc.m[t], 0, 0, 0);
            acc.r[t] = __builtin_amdgcn_mfma_f64_4x4x4f64(o.a[1][e], o.b[t][e], acc.r[t], 0, 0, 0);
        }
}
__device__ __forceinline__ void s5_mfma_single(S5Acc& acc, const S5Single& o) {
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        acc.m[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(o.a[0], o.b[t], acc.m[t], 0, 0, 0);
        acc.r[t] = __builtin_amdgcn_mfma_f64_4x4x4f64(o.a[1], o.b[t], acc.r[t], 0, 0, 0);
    }
}

// All schedule entries of a group for one wave (output spin `sig`).  An entry = (neighbour slot, flip): the wave multiplies
// the slot's [sig][si] quadrant with input spin si = sig (spin-diagonal part; the only entry of a collinear hopping block)
// or si = 1 - sig (spin-flip part of the spin-orbit / non-collinear blocks).  Three steps per entry -- k-pair 0 (X), k-pair 1 (Y),
// single k-step (Z) -- with the operands requested TWO steps ahead: while X runs, Y (issued one step ago) and Z (issued now) are
// in flight; a register set is reloaded right after the step that consumed it.
// TWO: slot id `nslots` (one past the lattice's slots) is the extra on-site slot; it reads the second input vector in2b (hoh second
// pass, recursion.f90:1543: the (e_nu + l.s) term acts on psi itself; local-axis runs: the per-chain on-site term) and, if
// fr_extra is given, takes its fragments from that per-chain table.
template <bool TWO>
__device__ __forceinline__ void s5_run_slots(S5Acc& acc, const int* __restrict__ share, const double* __restrict__ fr, const double* __restrict__ fr_extra,
                                             const char* __restrict__ inb, const char* __restrict__ in2b,
                                             const int* __restrict__ nbr5 /*(kk+1) x (nslots+1): absent -> zero block, last column = self*/,
                                             const int (&atom)[GROUP] /*padding -> zero block*/, unsigned rem_row /*per lane: (nslots+1) * atom of the remainder column*/,
                                             int nslots, int sig,
                                             unsigned lane_main, unsigned lane_single, unsigned lane_rem, unsigned lane_rem_single, unsigned lane16, unsigned lane8) {
    const int nmine = share[0];
    if (nmine <= 0) return;
    const int nstride = nslots + 1;
    // neighbour indices: wave-uniform scalar loads for the 8 atom tiles, one per-lane load for the remainder tile
    auto load_idx = [&](int s, int (&n)[GROUP], int& nr) {
#pragma unroll
        for (int t = 0; t < GROUP; ++t) n[t] = nbr5[(size_t)nstride * atom[t] + s];
        nr = nbr5[rem_row + (unsigned)s];
    };
    auto make_slot = [&](const int (&n)[GROUP], int nr, S5Slot& S, int s) {
        const char* base = (TWO && s == nslots) ? in2b : inb;
#pragma unroll
        for (int t = 0; t < GROUP; ++t) S.tile[t] = base + (size_t)n[t] * (BLD * 8);
        S.base = base;
        S.rem = (unsigned)nr * (BLD * 8u);
    };
    const double* __restrict__ fr_sig = fr + (size_t)sig * (2 * 2 * S5_FRAG_PER_PART);
    const double* __restrict__ fx_sig = (TWO && fr_extra) ? fr_extra + (size_t)sig * (2 * 2 * S5_FRAG_PER_PART) - (size_t)nslots * S5_FRAG_PER_SLOT : fr_sig;
    auto frag_of = [&](int s, int si) {
        const double* __restrict__ base = (TWO && s == nslots) ? fx_sig : fr_sig;
        return reinterpret_cast<const char*>(base + (size_t)s * S5_FRAG_PER_SLOT + si * (2 * S5_FRAG_PER_PART));
    };
    auto spin_of = [&](int e) { return (e >> 8) ? 1 - sig : sig; };
    constexpr int NL = 11, NM_PAIR = 36, NM_SINGLE = 18;     // operand loads / MFMA instructions per step, for the issue interleave
    S5Slot cur;
    int nraw[GROUP], nrem;
    int e_cur = share[1];
    int e_nxt = share[1 + ((1 < nmine) ? 1 : 0)];
    load_idx(e_cur & 255, nraw, nrem);
    make_slot(nraw, nrem, cur, e_cur & 255);
    S5Pair X, Y;
    S5Single Z;
    s5_load_pair<0>(X, cur, 2592u * spin_of(e_cur), frag_of(e_cur & 255, spin_of(e_cur)), lane_main, lane_rem, lane16);
    s5_load_pair<1>(Y, cur, 2592u * spin_of(e_cur), frag_of(e_cur & 255, spin_of(e_cur)), lane_main, lane_rem, lane16);
    for (int j = 0; j < nmine; ++j) {
        const int e_nxt2 = share[1 + ((j + 2 < nmine) ? j + 2 : 0)];   // the last entry prefetches the first again (discarded): no tail branch
        load_idx(e_nxt & 255, nraw, nrem);
        const int si = spin_of(e_cur);
        s5_load_single(Z, cur, 2592u * si, frag_of(e_cur & 255, si), lane_single, lane_rem_single, lane8);     // Z(e): the last load of this entry
        s5_mfma_pair(acc, X);
        s5_interleave<NL, NM_PAIR>();
        __builtin_amdgcn_sched_barrier(0);
        make_slot(nraw, nrem, cur, e_nxt & 255);        // every load of the current entry has been issued: reuse its addressing state
        s5_load_pair<0>(X, cur, 2592u * spin_of(e_nxt), frag_of(e_nxt & 255, spin_of(e_nxt)), lane_main, lane_rem, lane16);
        s5_mfma_pair(acc, Y);
        s5_interleave<NL, NM_PAIR>();
        __builtin_amdgcn_sched_barrier(0);
        s5_load_pair<1>(Y, cur, 2592u * spin_of(e_nxt), frag_of(e_nxt & 255, spin_of(e_nxt)), lane_main, lane_rem, lane16);
        s5_mfma_single(acc, Z);
        s5_interleave<NL, NM_SINGLE>();
        __builtin_amdgcn_sched_barrier(0);
        e_cur = e_nxt;
        e_nxt = e_nxt2;
    }
}

// One wave = (group of 8 atoms, output spin).  Workgroup = 8 waves = 4 groups x 2 spins; waves w and w + 4 (same group,
// different spin) land on the same SIMD.  Input and output vectors in the CI layout.
template <bool TWO>
__global__ __launch_bounds__(S5_WG_GROUPS * 128) void k_spmm5(SpmmDims D, const int* __restrict__ order_all, const int* __restrict__ cum,
                                               const int* __restrict__ nbr /*nbr5: (kk+1) x (nslots+1), absent -> kk, last column = self*/,
                                               const int* __restrict__ izp, const double* __restrict__ frag, const int* __restrict__ meta,
                                               const double* __restrict__ in_all, double* __restrict__ out_all,
                                               const double* __restrict__ in2_all = nullptr,
                                               const double* __restrict__ frag_extra = nullptr /*[chain][tau][S5_FRAG_PER_SLOT]: per-chain extra-slot operator*/,
                                               int ntau = 0) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int sig = wave / S5_WG_GROUPS, gslot = wave % S5_WG_GROUPS;
    const int l15 = lane & 15, l4 = lane >> 4;
    // CI layout (kernels_mfma.hpp): element (r, c) of a block = complex at doubles 36 r + 2 c; row r = 9 sigma + m.  k-pair P of
    // input spin sigma = rows m = 4 P + l4 (re, im = the pair's two k-steps); single k-step = row m = 8 (lane row 0: re, 1: im)
    const unsigned lane_main = 8u * (36 * l4 + 2 * l15), lane_single = 8u * (288 + 2 * l15 + (l4 & 1));
    const unsigned lane_rem = 8u * (36 * l4 + 32 + 2 * (l15 & 1)), lane_rem_single = 8u * (288 + 32 + 2 * (l15 & 1) + (l4 & 1));
    const unsigned lane16 = 16u * lane, lane8 = 8u * lane;
    // grid.y may be smaller than the number of chains ("chain_fold"): a workgroup then serves chains blockIdx.y, + gridDim.y, ...
#pragma unroll 1
    for (int chain = blockIdx.y; chain < D.nchains; chain += gridDim.y) {
    const int count = cum[(chain / D.cpo) * D.nlev + D.level];
    const int ngroups = count / GROUP;
    const int* __restrict__ order = order_all + (size_t)(chain / D.cpo) * D.ostride + D.obase[(chain / D.cpo) * D.nlev + D.level];
    const size_t vo = (size_t)chain * D.vstride;
    const char* __restrict__ inb = reinterpret_cast<const char*>(in_all + vo);
    const char* __restrict__ in2b = TWO ? reinterpret_cast<const char*>(in2_all + vo) : nullptr;
    double* __restrict__ out = out_all + vo;
    const int zero_block = D.kk;

    // XCD x sweeps chunk x of the group list (workgroups are dealt round-robin over the 8 XCDs, each with its own L2)
    int g, gend, gstep;
    {
        const int nbx = max(1, min((int)gridDim.x, (ngroups + S5_WG_GROUPS - 1) / S5_WG_GROUPS)), bx = blockIdx.x;
        if (bx >= nbx) continue;                             // launch sized for the largest chain of the batch
        if (nbx < 8) { g = bx * S5_WG_GROUPS + gslot; gend = ngroups; gstep = nbx * S5_WG_GROUPS; }
        else {
            const int xcd = bx & 7, j = bx >> 3;
            const int per_xcd = (nbx >> 3) + ((xcd < (nbx & 7)) ? 1 : 0);
            const int chunk = (ngroups + 7) >> 3;
            const int lo = xcd * chunk;
            gend = min(ngroups, lo + chunk);
            g = lo + j * S5_WG_GROUPS + gslot;
            gstep = per_xcd * S5_WG_GROUPS;
        }
    }
    for (; g < gend; g += gstep) {
        const int* __restrict__ grp = order + (size_t)g * GROUP;
        int atom[GROUP];                                    // padding entries (-1) become the zero block: no predicates in the slot loop
#pragma unroll
        for (int t = 0; t < GROUP; ++t) { const int a = grp[t]; atom[t] = a >= 0 ? a : zero_block; }
        const int first = atom[0];
        const int tau = first < D.nmax ? first : D.nmax + izp[first];
        int my_rem_atom = grp[l15 >> 1];
        my_rem_atom = my_rem_atom >= 0 ? my_rem_atom : zero_block;
        const unsigned rem_row = (unsigned)(D.nslots + 1) * (unsigned)my_rem_atom;
        const int* __restrict__ M = meta + (size_t)tau * Spmm5Operator::META;
        const double* __restrict__ fr = frag + (size_t)tau * (D.nslots + 1) * S5_FRAG_PER_SLOT;
        const double* __restrict__ fx = (TWO && frag_extra) ? frag_extra + ((size_t)chain * ntau + tau) * S5_FRAG_PER_SLOT : nullptr;

        S5Acc acc;
#pragma unroll
        for (int t = 0; t < 9; ++t) { acc.m[t] = (double4_t){0, 0, 0, 0}; acc.r[t] = 0.0; }

        s5_run_slots<TWO>(acc, M, fr, fx, inb, in2b, nbr, atom, rem_row, D.nslots, sig, lane_main, lane_single, lane_rem, lane_rem_single, lane16, lane8);

        // 16x16x4 result register j, lane (l15, l4): real-form row l4 + 4 j of spin sig = (part j & 1, m = l4 + 4 (j >> 1)), column l15:
        // registers (2 p, 2 p + 1) are the real and imaginary part of element (m = 4 p + l4, c) -> one 16-byte store in the CI
        // layout; the 4x4x4 result: row 16 + l4 -> l4 = 0: re, 1: im of m = 8
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int a = (t < 8) ? atom[t] : my_rem_atom;
            if (a == zero_block) continue;
            double* ob = out + (size_t)BLD * a + 324 * sig + ((t < 8) ? 2 * l15 : 32 + 2 * (l15 & 1));
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                s5_d2 v; v[0] = acc.m[t][2 * p]; v[1] = acc.m[t][2 * p + 1];
                *reinterpret_cast<s5_d2*>(ob + 36 * (4 * p + l4)) = v;
            }
            if (l4 < 2) ob[288 + l4] = acc.r[t];
        }
    }
    }   // chains of this workgroup
}

}  // namespace rsrec
