// Block-sparse H|psi>, third generation ("spmm5"): spin-split waves, 16-byte operand loads, two waves per SIMD.
//
// What the profiles of the previous kernel (k_spmm4, profiles/r01_spmm4_rocprof_summary.txt + tools/pmc_sq.sh) showed:
// the matrix pipe was busy 51 % of the time, the waves spent 65 % of their cycles stalled at instruction issue, and removing
// the operand LOADS from the k-loop (results wrong, timing only: tools/probe_spmm_bound.sh) made the kernel 1.8x faster no
// matter where the data came from.  With one 500-register wave per SIMD every 8-byte-per-lane global load costs the wave
// ~45 cycles of issue that nothing else covers, and 130 of them per neighbour slot also keep the CU's address unit busy
// ~2100 of the slot's 7056 matrix cycles.  This kernel attacks the load count and the exposure:
//   * rows of the real form are ordered spin-major and each spin padded from 18 to 20 rows (5 four-row blocks): the
//     collinear operators (no spin-flip hopping; hamiltonian.f90:1553-1617) are then exactly block diagonal at MFMA
//     granularity and ONE WAVE OWNS ONE SPIN of a group of 8 atoms: 45 accumulators instead of 162, so two waves fit
//     on a SIMD and cover each other's issue stalls.  Spin-mixing blocks (spin-orbit on-site term, non-collinear
//     operators) take the same path with both input spins.
//   * the work vectors are complex-interleaved row-major ("CI", kernels_mfma.hpp) and the real form's k order is chosen so that
//     the real and imaginary part of one element are the B operands of two consecutive k-steps: one 16-byte load feeds two
//     MFMA k-steps; the operator fragments are stored the same way.  Per slot and wave: 42 loads for 225 MFMAs (before: 130
//     for 441).  (Round 1 used a separate "k-pair" copy of the vector for this; CI is the ONE layout of the large-launch path:
//     the Gram / orthogonalisation kernels read and write it with the same 16-byte accesses.)
//   * neighbour blocks are wave-uniform: their addresses are SGPR bases, the lane part of every address is one of three
//     loop-invariant registers, everything else an instruction immediate: no address arithmetic on the vector ALU.
#pragma once
#include <hip/hip_runtime.h>
#include <algorithm>
#include <vector>
#include "kernels_valu.hpp"
#include "kernels_mfma.hpp"
#include "kernels_spmm4.hpp"

namespace rsrec {

// (kp_offset, the element map of the KP layout, lives in kernels_mfma.hpp: the Gram / orthogonalisation kernels read KP vectors too)

#ifndef S5_WG_GROUPS
#define S5_WG_GROUPS 4   // groups of 8 atoms per workgroup (x 2 spin waves each): 4 -> 512 threads, 2 -> 256 threads
#endif
#ifndef S5_VARIANT
#define S5_VARIANT 0     // development A/B switches, measured on one box against 0 (5.43-5.47 ms per level):
                         //   1 = operand loads not interleaved with the MFMAs (+2.5 %), 2 = no scheduling fence between steps (+0.6 %),
                         //   3 = plain round-robin group walk instead of XCD chunks (+13 %), 4 = loads front-loaded in the step (+0.6 %),
                         //   5 = different s_setprio for the two spin waves (+0.7 %);  S5_WG_GROUPS = 2 (256-thread workgroups): +15 %
#endif
#ifndef S5_PROBE
#define S5_PROBE 0   // timing probes only (results wrong): bit 0 = no operator-fragment loads in the slot loop, bit 1 = no psi loads, bit 2 = schedule walked twice
#endif
constexpr int S5_FRAG_PER_RB = 320;                         // doubles: pair 0 (128), pair 1 (128), single (64)
constexpr int S5_FRAG_PER_SLOT = 2 * 2 * 5 * S5_FRAG_PER_RB;   // [sigma_out][sigma_in][rb]
// "wide" variant (k_spmm5<.., true>): the first 16 real-form rows of a spin go through ONE v_mfma_f64_16x16x4 per tile and k-step
// instead of four 4x4x4 row blocks -- the same matrix cycles (64 vs 4 x 16), but the operator is delivered in 2 operand registers
// per k-step instead of 5 (rows 16..19 keep a 4x4x4): 33 instead of 42 operand loads per neighbour slot and 90 instead of 225 MFMA
// instructions.  Fragments: [sigma_out][sigma_in][2: rows 0..15 | rows 16..19][320].
constexpr int S5W_FRAG_PER_SLOT = 2 * 2 * 2 * S5_FRAG_PER_RB;
template <bool WIDE> struct S5Cfg { static constexpr int NA = WIDE ? 2 : 5; static constexpr int FRAG_PER_SLOT = WIDE ? S5W_FRAG_PER_SLOT : S5_FRAG_PER_SLOT; };

struct Spmm5Operator {
    double* d_frag = nullptr;    // [set][tau][slot 0..nslots][sigma_out][sigma_in][rb][320]
    double* d_fragw = nullptr;   // wide variant: [set][tau][slot 0..nslots][sigma_out][sigma_in][2][320]
    size_t fragw_bytes = 0;
    int* d_meta = nullptr;       // [set][tau][1 + 2 S4_MAXSLOTS]: count, then entries slot | flip << 8 (flip: input spin = the other spin)
    size_t frag_bytes = 0, meta_bytes = 0;
    int ntau = 0, nslots = 0, have_o = 0;
    static constexpr int META = 1 + 2 * S4_MAXSLOTS;

    void release() {
        if (d_frag) (void)hipFree(d_frag);
        if (d_fragw) (void)hipFree(d_fragw);
        if (d_meta) (void)hipFree(d_meta);
        d_frag = nullptr; d_fragw = nullptr; d_meta = nullptr; frag_bytes = fragw_bytes = meta_bytes = 0;
    }
    // entry (row ko, column ki) of the padded 40x40 real form: index = 20 sigma + rho.  rho = 4 s + l (k-step s, lane row l):
    // s = 2 P + e < 4 -> (part e, m = 4 P + l); s = 4 -> l = 0: (re, m = 8), l = 1: (im, m = 8), l = 2, 3: padding -- the two
    // members of a k-pair are the real and imaginary part of ONE complex element of the vector (layout CI below)
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
    static void swizzle(const double* blk, double* out) {
        for (int so = 0; so < 2; ++so)
            for (int si = 0; si < 2; ++si)
                for (int rb = 0; rb < 5; ++rb) {
                    double* o = out + ((so * 2 + si) * 5 + rb) * S5_FRAG_PER_RB;
                    for (int l = 0; l < 64; ++l) {       // A operand of the 4x4x4 MFMA: lane (i + 4 g + 16 k) = A[i][k], same for the 4 blocks g
                        const int ko = 20 * so + 4 * rb + (l & 3), k = l >> 4;
                        for (int p = 0; p < 2; ++p)
                            for (int e = 0; e < 2; ++e) o[128 * p + 2 * l + e] = real40(blk, ko, 20 * si + 8 * p + 4 * e + k);
                        o[256 + l] = real40(blk, ko, 20 * si + 16 + k);
                    }
                }
    }
    // wide variant: A operand of the 16x16x4 MFMA for rows 0..15 (lane (l15 = row, l4 = k)), of the 4x4x4 MFMA for rows 16..19
    static void swizzle_wide(const double* blk, double* out) {
        for (int so = 0; so < 2; ++so)
            for (int si = 0; si < 2; ++si) {
                double* o = out + (so * 2 + si) * 2 * S5_FRAG_PER_RB;
                for (int l = 0; l < 64; ++l) {
                    const int k = l >> 4;
                    for (int part = 0; part < 2; ++part) {
                        const int ko = 20 * so + (part == 0 ? (l & 15) : 16 + (l & 3));
                        double* q = o + part * S5_FRAG_PER_RB;
                        for (int p = 0; p < 2; ++p)
                            for (int e = 0; e < 2; ++e) q[128 * p + 2 * l + e] = real40(blk, ko, 20 * si + 8 * p + 4 * e + k);
                        q[256 + l] = real40(blk, ko, 20 * si + 16 + k);
                    }
                }
            }
    }
    // Set 0: the blocks of h (slot 0 carries + l.s when !hoh).  Set 1 (hoh second pass) is built so that ONE SpMM pass over
    // hpsi = h psi plus one extra on-site slot reading psi gives the whole  H psi = hpsi - (h o) hpsi + (e_nu + l.s) psi
    // (recursion.f90:1543):  slot 0 -> 1 - (h o)_0,  slot s -> -(h o)_s,  slot `nslots` (extra) -> enim + lsham of the atom's type.
    // Every tau has nslots + 1 fragment slots in both sets (the extra one stays zero and unlisted in set 0).
    const char* build(int nslots_lat, int hstride, int ntype, int nmax, int hoh, const double* st, const double* loc, const double* eeo, const double* hallo,
                      const double* enim, const double* lsham, const int* iz0) {
        if (nslots_lat + 1 > S4_MAXSLOTS) return "too many neighbour slots for the spmm5 kernel";
        ntau = nmax + ntype; nslots = nslots_lat; have_o = hoh ? 1 : 0;
        const int nset = have_o ? 2 : 1;
        const int nfs = nslots + 1;
        const size_t per_set = (size_t)ntau * nfs * S5_FRAG_PER_SLOT, per_setw = (size_t)ntau * nfs * S5W_FRAG_PER_SLOT;
        std::vector<double> host(per_set * nset, 0.0), hostw(per_setw * nset, 0.0);
        std::vector<int> meta((size_t)nset * ntau * META, 0);
        std::vector<double> tmp(2 * BLK);
        for (int set = 0; set < nset; ++set)
            for (int tau = 0; tau < ntau; ++tau) {
                int* M = meta.data() + ((size_t)set * ntau + tau) * META;
                for (int s = 0; s < nslots + (set ? 1 : 0); ++s) {
                    const double* src;
                    if (s == nslots) {
                        const int ty = tau < nmax ? iz0[tau] : tau - nmax;
                        for (int e = 0; e < 2 * BLK; ++e) tmp[e] = enim[2 * (size_t)BLK * ty + e] + lsham[2 * (size_t)BLK * ty + e];
                        src = tmp.data();
                    } else {
                        if (tau < nmax) src = (set ? hallo : loc) + 2 * (size_t)BLK * (s + (size_t)hstride * tau);
                        else src = (set ? eeo : st) + 2 * (size_t)BLK * (s + (size_t)hstride * (tau - nmax));
                        if (set) {
                            for (int e = 0; e < 2 * BLK; ++e) tmp[e] = -src[e];
                            if (s == 0) for (int d = 0; d < NB; ++d) tmp[2 * (d + NB * d)] += 1.0;
                            src = tmp.data();
                        }
                    }
                    swizzle(src, host.data() + set * per_set + ((size_t)tau * nfs + s) * S5_FRAG_PER_SLOT);
                    swizzle_wide(src, hostw.data() + set * per_setw + ((size_t)tau * nfs + s) * S5W_FRAG_PER_SLOT);
                    // schedule: the spin-diagonal part of every block, plus the spin-flip part of blocks that have one
                    M[1 + M[0]] = s; M[0]++;
                    if (Spmm4Operator::pattern_of(src) == 0) { M[1 + M[0]] = s | (1 << 8); M[0]++; }
                }
            }
        return upload(host, hostw, meta);
    }
    // General table: blk[(set * ntau + tau) * (nslots + 1) + s] = column-major interleaved 18x18 complex block of operator class tau,
    // fragment slot s (s = nslots: the extra on-site slot that reads the second input vector), or nullptr = absent (contributes
    // nothing and is not scheduled).  Used for operators other than H itself: the Kubo velocity operators (velo_vec_matmul,
    // recursion.f90:587) and their hoh combinations.
    const char* build_custom(int nslots_lat, int ntau_, int nset, const std::vector<const double*>& blk) {
        if (nslots_lat + 1 > S4_MAXSLOTS) return "too many neighbour slots for the spmm5 kernel";
        ntau = ntau_; nslots = nslots_lat; have_o = nset > 1 ? 1 : 0;
        const int nfs = nslots + 1;
        const size_t per_set = (size_t)ntau * nfs * S5_FRAG_PER_SLOT, per_setw = (size_t)ntau * nfs * S5W_FRAG_PER_SLOT;
        std::vector<double> host(per_set * nset, 0.0), hostw(per_setw * nset, 0.0);
        std::vector<int> meta((size_t)nset * ntau * META, 0);
        for (int set = 0; set < nset; ++set)
            for (int tau = 0; tau < ntau; ++tau) {
                int* M = meta.data() + ((size_t)set * ntau + tau) * META;
                for (int s = 0; s < nfs; ++s) {
                    const double* src = blk[((size_t)set * ntau + tau) * nfs + s];
                    if (!src) continue;
                    swizzle(src, host.data() + set * per_set + ((size_t)tau * nfs + s) * S5_FRAG_PER_SLOT);
                    swizzle_wide(src, hostw.data() + set * per_setw + ((size_t)tau * nfs + s) * S5W_FRAG_PER_SLOT);
                    M[1 + M[0]] = s; M[0]++;
                    if (Spmm4Operator::pattern_of(src) == 0) { M[1 + M[0]] = s | (1 << 8); M[0]++; }
                }
            }
        return upload(host, hostw, meta);
    }
    const char* upload(const std::vector<double>& host, const std::vector<double>& hostw, const std::vector<int>& meta) {
        const size_t need = host.size() * sizeof(double), mneed = meta.size() * sizeof(int), needw = hostw.size() * sizeof(double);
        if (needw > fragw_bytes) {
            if (d_fragw) (void)hipFree(d_fragw);
            d_fragw = nullptr; fragw_bytes = 0;
            if (hipMalloc(reinterpret_cast<void**>(&d_fragw), needw) != hipSuccess) return "hipMalloc of spmm5 operator fragments failed";
            fragw_bytes = needw;
        }
        if (hipMemcpy(d_fragw, hostw.data(), needw, hipMemcpyHostToDevice) != hipSuccess) return "upload of spmm5 fragments failed";
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
    const double* frag_set(int set, bool wide = false) const {
        return wide ? d_fragw + (size_t)set * ntau * (nslots + 1) * S5W_FRAG_PER_SLOT : d_frag + (size_t)set * ntau * (nslots + 1) * S5_FRAG_PER_SLOT;
    }
    const int* meta_set(int set) const { return d_meta + (size_t)set * ntau * META; }
};

typedef double s5_d2 __attribute__((ext_vector_type(2)));
template <bool WIDE> struct S5Pair { s5_d2 b[9]; s5_d2 a[S5Cfg<WIDE>::NA]; };      // operands of two k-steps: psi tiles, operator row blocks
template <bool WIDE> struct S5Single { double b[9]; double a[S5Cfg<WIDE>::NA]; };  // the spin's fifth k-step (row m = 8: re, im + padding)
// accumulators of a wave: narrow = 5 row blocks x 9 tiles of 4x4x4 results; wide = 9 tiles x (16x16x4 result + 4x4x4 result)
template <bool WIDE> struct S5Acc;
template <> struct S5Acc<false> { double v[5][9]; };
template <> struct S5Acc<true> { double4_t m[9]; double r[9]; };

// wave-uniform addressing state of one neighbour slot
struct S5Slot {
    const char* tile[GROUP];   // KP block of the neighbour of atom t (spin 0 half)
    const char* base;          // vector the slot reads (the second input for the extra on-site slot of the hoh second pass)
    unsigned rem;              // remainder tile: byte offset of this lane's neighbour block (per lane: atom l15 >> 1)
};

template <int P, bool WIDE, bool LOOP = true>
__device__ __forceinline__ void s5_load_pair(S5Pair<WIDE>& o, const S5Slot& S, unsigned spin_off, const char* __restrict__ fb,
                                             unsigned lane_main, unsigned lane_rem, unsigned lane16) {
    if (!(LOOP && (S5_PROBE & 2))) {
#pragma unroll
        for (int t = 0; t < GROUP; ++t) o.b[t] = *reinterpret_cast<const s5_d2*>(S.tile[t] + spin_off + lane_main + 1152 * P);
        o.b[8] = *reinterpret_cast<const s5_d2*>(S.base + spin_off + (S.rem + lane_rem) + 1152 * P);
    }
    if (!(LOOP && (S5_PROBE & 1))) {
#pragma unroll
        for (int rb = 0; rb < S5Cfg<WIDE>::NA; ++rb) o.a[rb] = *reinterpret_cast<const s5_d2*>(fb + lane16 + (rb * S5_FRAG_PER_RB * 8 + 1024 * P));
    }
}
template <bool WIDE>
__device__ __forceinline__ void s5_load_single(S5Single<WIDE>& o, const S5Slot& S, unsigned spin_off, const char* __restrict__ fb,
                                               unsigned lane_single, unsigned lane_rem_single, unsigned lane8) {
    if (!(S5_PROBE & 2)) {
#pragma unroll
        for (int t = 0; t < GROUP; ++t) o.b[t] = *reinterpret_cast<const double*>(S.tile[t] + spin_off + lane_single);
        o.b[8] = *reinterpret_cast<const double*>(S.base + spin_off + (S.rem + lane_rem_single));
    } else {
#pragma unroll
        for (int t = 0; t < 9; ++t) o.b[t] = 1e-3 * (t + 1);
    }
    if (!(S5_PROBE & 1)) {
#pragma unroll
        for (int rb = 0; rb < S5Cfg<WIDE>::NA; ++rb) o.a[rb] = *reinterpret_cast<const double*>(fb + lane8 + (rb * S5_FRAG_PER_RB * 8 + 2048));
    } else {
#pragma unroll
        for (int rb = 0; rb < S5Cfg<WIDE>::NA; ++rb) o.a[rb] = 1e-3 * (rb + 1);
    }
}

// issue order: one operand load, then PER MFMAs (see k_spmm4's s4_interleave)
template <int NL, int NM>
__device__ __forceinline__ void s5_interleave() {
    if (S5_VARIANT == 1) return;
    constexpr int PER = (S5_VARIANT == 4) ? NM / (2 * NL) : NM / NL;
#pragma unroll
    for (int i = 0; i < NL; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x20, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x8, PER, 0);
    }
    __builtin_amdgcn_sched_group_barrier(0x8, NM - PER * NL, 0);
}

__device__ __forceinline__ void s5_mfma_pair(S5Acc<false>& acc, const S5Pair<false>& o) {
#pragma unroll
    for (int e = 0; e < 2; ++e)
#pragma unroll
        for (int rb = 0; rb < 5; ++rb)
#pragma unroll
            for (int t = 0; t < 9; ++t) acc.v[rb][t] = __builtin_amdgcn_mfma_f64_4x4x4f64(o.a[rb][e], o.b[t][e], acc.v[rb][t], 0, 0, 0);
}
__device__ __forceinline__ void s5_mfma_single(S5Acc<false>& acc, const S5Single<false>& o) {
#pragma unroll
    for (int rb = 0; rb < 5; ++rb)
#pragma unroll
        for (int t = 0; t < 9; ++t) acc.v[rb][t] = __builtin_amdgcn_mfma_f64_4x4x4f64(o.a[rb], o.b[t], acc.v[rb][t], 0, 0, 0);
}
__device__ __forceinline__ void s5_mfma_pair(S5Acc<true>& acc, const S5Pair<true>& o) {
#pragma unroll
    for (int e = 0; e < 2; ++e)
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            acc.m[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(o.a[0][e], o.b[t][e], acc.m[t], 0, 0, 0);
            acc.r[t] = __builtin_amdgcn_mfma_f64_4x4x4f64(o.a[1][e], o.b[t][e], acc.r[t], 0, 0, 0);
        }
}
__device__ __forceinline__ void s5_mfma_single(S5Acc<true>& acc, const S5Single<true>& o) {
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        acc.m[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(o.a[0], o.b[t], acc.m[t], 0, 0, 0);
        acc.r[t] = __builtin_amdgcn_mfma_f64_4x4x4f64(o.a[1], o.b[t], acc.r[t], 0, 0, 0);
    }
}

// All schedule entries of a group for one wave (output spin `sig`).  An entry = (neighbour slot, flip): the wave multiplies
// the slot's [sig][si] quadrant with input spin si = sig (spin-diagonal part; the only entry of a collinear hopping block)
// or si = 1 - sig (spin-flip part of the spin-orbit / non-collinear blocks).  Three steps per entry: k-pair 0, k-pair 1,
// single k-step; the operands of the next step are loaded while the MFMAs of the current one run.
// TWO: slot id `nslots` (one past the lattice's slots) is the extra on-site slot of the hoh second pass; it reads the second
// input vector in2b (recursion.f90:1543: H psi = h psi - (h o)(h psi) + (e_nu + l.s) psi, the last term acts on psi itself).
template <bool TWO, bool WIDE, int PF = 1>
__device__ __forceinline__ void s5_run_slots(S5Acc<WIDE>& acc, const int* __restrict__ share, const double* __restrict__ fr, const double* __restrict__ fr_extra, const char* __restrict__ inb,
                                             const char* __restrict__ in2b, const int* __restrict__ nbr5 /*(kk+1) x (nslots+1): absent -> zero block, last column = self*/,
                                             const int (&atom)[GROUP] /*padding -> zero block*/, unsigned rem_row /*per lane: (nslots+1) * atom of the remainder column*/,
                                             int nslots, int sig,
                                             unsigned lane_main, unsigned lane_single, unsigned lane_rem, unsigned lane_rem_single, unsigned lane16, unsigned lane8) {
    const int n0 = share[0];
    const int nmine = (S5_PROBE & 4) ? 2 * n0 : n0;           // probe 4: the schedule is walked twice (fixed per-group cost = 2 T(1x) - T(2x))
    if (nmine <= 0) return;
    const int nstride = nslots + 1;
    // neighbour indices: wave-uniform scalar loads for the 8 atom tiles, one per-lane load for the remainder tile; both are
    // issued a whole entry before they are turned into addresses
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
    constexpr int NA = S5Cfg<WIDE>::NA, FPS = S5Cfg<WIDE>::FRAG_PER_SLOT;
    const double* __restrict__ fr_sig = fr + (size_t)sig * (2 * NA * S5_FRAG_PER_RB);
    // fr_extra (TWO only, may be null): fragments of the extra on-site slot taken from a per-chain table instead of the shared one
    const double* __restrict__ fx_sig = (TWO && fr_extra) ? fr_extra + (size_t)sig * (2 * NA * S5_FRAG_PER_RB) - (size_t)nslots * FPS : fr_sig;
    auto frag_of = [&](int s, int si) {
        const double* __restrict__ base = (TWO && s == nslots) ? fx_sig : fr_sig;
        return reinterpret_cast<const char*>(base + (size_t)s * FPS + si * (NA * S5_FRAG_PER_RB));
    };
    auto spin_of = [&](int e) { return (e >> 8) ? 1 - sig : sig; };
    S5Slot cur;
    int nraw[GROUP], nrem;
    int e_cur = share[1];
    int e_nxt = share[1 + ((1 < nmine) ? 1 % n0 : 0)];
    load_idx(e_cur & 255, nraw, nrem);
    make_slot(nraw, nrem, cur, e_cur & 255);
    S5Pair<WIDE> X, Y;
    S5Single<WIDE> Z;
    // operand loads per step / MFMA instructions per step, for the issue interleave
    constexpr int NL = 9 + NA, NM_PAIR = WIDE ? 36 : 90, NM_SINGLE = WIDE ? 18 : 45;
    s5_load_pair<0, WIDE, false>(X, cur, 2592u * spin_of(e_cur), frag_of(e_cur & 255, spin_of(e_cur)), lane_main, lane_rem, lane16);
    if (S5_PROBE || PF == 2) s5_load_pair<1, WIDE, false>(Y, cur, 2592u * spin_of(e_cur), frag_of(e_cur & 255, spin_of(e_cur)), lane_main, lane_rem, lane16);
    if (S5_VARIANT == 5) { if (sig) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(2); }
    if (PF == 2) {
        // Prefetch distance TWO steps with the same three register sets: while step X runs, Y (issued one step ago) and Z (issued
        // now) are in flight; a set is reloaded right after the step that consumed it.  (Distance one left every load a single
        // step -- 1440 matrix cycles -- to arrive; gathers that miss L2 take longer.)
        for (int j = 0; j < nmine; ++j) {
            const int e_nxt2 = share[1 + ((j + 2 < nmine) ? ((S5_PROBE & 4) ? (j + 2) % n0 : j + 2) : 0)];
            load_idx(e_nxt & 255, nraw, nrem);
            const int si = spin_of(e_cur);
            const unsigned so = 2592u * si;
            const char* __restrict__ fb = frag_of(e_cur & 255, si);
            s5_load_single<WIDE>(Z, cur, so, fb, lane_single, lane_rem_single, lane8);     // Z(e): the last load of this entry
            s5_mfma_pair(acc, X);
            s5_interleave<NL, NM_PAIR>();
            __builtin_amdgcn_sched_barrier(0);
            make_slot(nraw, nrem, cur, e_nxt & 255);
            s5_load_pair<0, WIDE>(X, cur, 2592u * spin_of(e_nxt), frag_of(e_nxt & 255, spin_of(e_nxt)), lane_main, lane_rem, lane16);
            s5_mfma_pair(acc, Y);
            s5_interleave<NL, NM_PAIR>();
            __builtin_amdgcn_sched_barrier(0);
            s5_load_pair<1, WIDE>(Y, cur, 2592u * spin_of(e_nxt), frag_of(e_nxt & 255, spin_of(e_nxt)), lane_main, lane_rem, lane16);
            s5_mfma_single(acc, Z);
            s5_interleave<NL, NM_SINGLE>();
            __builtin_amdgcn_sched_barrier(0);
            e_cur = e_nxt;
            e_nxt = e_nxt2;
        }
        return;
    }
    for (int j = 0; j < nmine; ++j) {
        const int e_nxt2 = share[1 + ((j + 2 < nmine) ? ((S5_PROBE & 4) ? (j + 2) % n0 : j + 2) : 0)];   // the last entry prefetches the first again (discarded)
        load_idx(e_nxt & 255, nraw, nrem);
        // (pinning the 32-bit lane offsets inside the loop makes hipcc emit SGPR-base + VGPR-offset loads, but costs 10 more VGPRs
        //  -> scratch spills in the group prologue and a 3.5 % slower kernel; measured, not used)
        const int si = spin_of(e_cur);
        const unsigned so = 2592u * si;
        const char* __restrict__ fb = frag_of(e_cur & 255, si);
        s5_load_pair<1, WIDE>(Y, cur, so, fb, lane_main, lane_rem, lane16);
        s5_mfma_pair(acc, X);
        s5_interleave<NL, NM_PAIR>();
        if (S5_VARIANT != 2) __builtin_amdgcn_sched_barrier(0);
        s5_load_single<WIDE>(Z, cur, so, fb, lane_single, lane_rem_single, lane8);
        s5_mfma_pair(acc, Y);
        s5_interleave<NL, NM_PAIR>();
        if (S5_VARIANT != 2) __builtin_amdgcn_sched_barrier(0);
        make_slot(nraw, nrem, cur, e_nxt & 255);        // the current entry's operands are all in flight or consumed: reuse its state
        s5_load_pair<0, WIDE>(X, cur, 2592u * spin_of(e_nxt), frag_of(e_nxt & 255, spin_of(e_nxt)), lane_main, lane_rem, lane16);
        s5_mfma_single(acc, Z);
        s5_interleave<NL, NM_SINGLE>();
        if (S5_VARIANT != 2) __builtin_amdgcn_sched_barrier(0);
        e_cur = e_nxt;
        e_nxt = e_nxt2;
    }
}

// One wave = (group of 8 atoms, output spin).  Workgroup = 8 waves = 4 groups x 2 spins; waves w and w + 4 (same group,
// different spin) land on the same SIMD.  Input and output vectors in the CI layout.
// TWO: second input vector for the extra on-site slot (second pass of hoh; the per-chain on-site term of local-axis runs).
template <bool TWO, bool WIDE = false, int PF = 1>
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
    // A workgroup serves the chains blockIdx.y, blockIdx.y + gridDim.y, ...: a 512-thread, 256-register workgroup lives only
    // 25-50 us per chain, so its launch cost is amortised over several chains (grid.y < number of chains)
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

    int g, gend, gstep;
    {
        const int nbx = max(1, min((int)gridDim.x, (ngroups + S5_WG_GROUPS - 1) / S5_WG_GROUPS)), bx = blockIdx.x;
        if (bx >= nbx) continue;                             // launch sized for the largest chain of the batch
        if (nbx < 8 || S5_VARIANT == 3) { g = bx * S5_WG_GROUPS + gslot; gend = ngroups; gstep = nbx * S5_WG_GROUPS; }
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
        const double* __restrict__ fr = frag + (size_t)tau * (D.nslots + 1) * S5Cfg<WIDE>::FRAG_PER_SLOT;

        S5Acc<WIDE> acc;
        if constexpr (WIDE) {
#pragma unroll
            for (int t = 0; t < 9; ++t) { acc.m[t] = (double4_t){0, 0, 0, 0}; acc.r[t] = 0.0; }
        } else {
#pragma unroll
            for (int rb = 0; rb < 5; ++rb)
#pragma unroll
                for (int t = 0; t < 9; ++t) acc.v[rb][t] = 0.0;
        }

        const double* __restrict__ fx = (TWO && frag_extra) ? frag_extra + ((size_t)chain * ntau + tau) * S5Cfg<WIDE>::FRAG_PER_SLOT : nullptr;
        s5_run_slots<TWO, WIDE, PF>(acc, M, fr, fx, inb, in2b, nbr, atom, rem_row, D.nslots, sig, lane_main, lane_single, lane_rem, lane_rem_single, lane16, lane8);

        // D layout: real-form row rho = 4 rb + l4 of spin sig, column l15.  rb = 2 P + e is (part e, m = 4 P + l4): the accumulators
        // (2P, 2P+1) are the real and imaginary part of element (m, c) -> one 16-byte store in the CI layout; rb = 4: m = 8
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int a = (t < 8) ? atom[t] : my_rem_atom;
            if (a == zero_block) continue;
            double* ob = out + (size_t)BLD * a + 324 * sig + ((t < 8) ? 2 * l15 : 32 + 2 * (l15 & 1));
            // (wide: the 16x16x4 result register j holds row l4 + 4 j of the spin = (part j & 1, m = l4 + 4 (j >> 1)): the same pairing)
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                s5_d2 v;
                if constexpr (WIDE) { v[0] = acc.m[t][2 * p]; v[1] = acc.m[t][2 * p + 1]; }
                else { v[0] = acc.v[2 * p][t]; v[1] = acc.v[2 * p + 1][t]; }
                *reinterpret_cast<s5_d2*>(ob + 36 * (4 * p + l4)) = v;
            }
            if (l4 < 2) { if constexpr (WIDE) ob[288 + l4] = acc.r[t]; else ob[288 + l4] = acc.v[4][t]; }
        }
    }
    }   // chains of this workgroup
}

// "Streamed" variant of the wide kernel: a wave walks SEVERAL groups and treats their schedule entries as one stream -- the first
// operands of the next group are requested during the last entry of the current one (prefetch distance two steps, as inside a
// group), so the group prologue (order list -> atoms -> operator class -> schedule -> neighbour table -> first operand loads, a
// chain of dependent loads of 3-5 us against 25 us of work) and the result stores of the finished group overlap with matrix work
// instead of idling the CU: with one 8-wave workgroup per CU nothing else covers them (matrix pipe 70 % busy,
// profiles/r02l_c2_sq_pmc.txt).  Launched with fewer, longer-lived workgroups (option "s5_items" groups per wave).
struct S5Group {
    int atom[GROUP];           // wave-uniform: the eight atoms (padding -> zero block)
    int rem_atom;              // per lane: atom of the lane's remainder-tile column
    unsigned rem_row;          // per lane: (nslots + 1) * rem_atom
    const int* M;              // schedule of the group's operator class: count, entries
    const double* fr;          // fragments of the class, this wave's output spin
    const double* fx;          // per-chain extra-slot fragments (same offset convention as fr) or fr
};

template <bool TWO>
__global__ __launch_bounds__(S5_WG_GROUPS * 128) void k_spmm5s(SpmmDims D, const int* __restrict__ order_all, const int* __restrict__ cum,
                                               const int* __restrict__ nbr, const int* __restrict__ izp, const double* __restrict__ frag,
                                               const int* __restrict__ meta, const double* __restrict__ in_all, double* __restrict__ out_all,
                                               const double* __restrict__ in2_all = nullptr, const double* __restrict__ frag_extra = nullptr, int ntau = 0) {
    constexpr bool WIDE = true;
    constexpr int NA = S5Cfg<WIDE>::NA, FPS = S5Cfg<WIDE>::FRAG_PER_SLOT, NL = 9 + NA, NM_PAIR = 36, NM_SINGLE = 18;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int sig = wave / S5_WG_GROUPS, gslot = wave % S5_WG_GROUPS;
    const int l15 = lane & 15, l4 = lane >> 4;
    const unsigned lane_main = 8u * (36 * l4 + 2 * l15), lane_single = 8u * (288 + 2 * l15 + (l4 & 1));
    const unsigned lane_rem = 8u * (36 * l4 + 32 + 2 * (l15 & 1)), lane_rem_single = 8u * (288 + 32 + 2 * (l15 & 1) + (l4 & 1));
    const unsigned lane16 = 16u * lane, lane8 = 8u * lane;
    const int nslots = D.nslots, nstride = D.nslots + 1;
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
        int g, gend, gstep;
        {
            const int nbx = max(1, min((int)gridDim.x, (ngroups + S5_WG_GROUPS - 1) / S5_WG_GROUPS)), bx = blockIdx.x;
            if (bx >= nbx) continue;
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
        if (g >= gend) continue;
        auto load_group = [&](int gg, S5Group& G) {
            const int* __restrict__ grp = order + (size_t)gg * GROUP;
#pragma unroll
            for (int t = 0; t < GROUP; ++t) { const int a = grp[t]; G.atom[t] = a >= 0 ? a : zero_block; }
            const int first = G.atom[0];
            const int tau = first < D.nmax ? first : D.nmax + izp[first];
            int ra = grp[l15 >> 1];
            G.rem_atom = ra >= 0 ? ra : zero_block;
            G.rem_row = (unsigned)nstride * (unsigned)G.rem_atom;
            G.M = meta + (size_t)tau * Spmm5Operator::META;
            G.fr = frag + (size_t)tau * nstride * FPS + (size_t)sig * (2 * NA * S5_FRAG_PER_RB);
            G.fx = (TWO && frag_extra) ? frag_extra + ((size_t)chain * ntau + tau) * FPS + (size_t)sig * (2 * NA * S5_FRAG_PER_RB) - (size_t)nslots * FPS : G.fr;
        };
        auto load_idx = [&](const S5Group& G, int s, int (&n)[GROUP], int& nr) {
#pragma unroll
            for (int t = 0; t < GROUP; ++t) n[t] = nbr[(size_t)nstride * G.atom[t] + s];
            nr = nbr[G.rem_row + (unsigned)s];
        };
        auto make_slot = [&](const int (&n)[GROUP], int nr, S5Slot& S, int s) {
            const char* base = (TWO && s == nslots) ? in2b : inb;
#pragma unroll
            for (int t = 0; t < GROUP; ++t) S.tile[t] = base + (size_t)n[t] * (BLD * 8);
            S.base = base;
            S.rem = (unsigned)nr * (BLD * 8u);
        };
        auto frag_of = [&](const S5Group& G, int s, int si) {
            const double* __restrict__ base = (TWO && s == nslots) ? G.fx : G.fr;
            return reinterpret_cast<const char*>(base + (size_t)s * FPS + si * (NA * S5_FRAG_PER_RB));
        };
        auto spin_of = [&](int e) { return (e >> 8) ? 1 - sig : sig; };

        S5Group G, Gn;
        load_group(g, G);
        S5Slot cur;
        int nraw[GROUP], nrem;
        int e_cur = G.M[1];
        load_idx(G, e_cur & 255, nraw, nrem);
        make_slot(nraw, nrem, cur, e_cur & 255);
        S5Pair<WIDE> X, Y;
        S5Single<WIDE> Z;
        s5_load_pair<0, WIDE, false>(X, cur, 2592u * spin_of(e_cur), frag_of(G, e_cur & 255, spin_of(e_cur)), lane_main, lane_rem, lane16);
        s5_load_pair<1, WIDE, false>(Y, cur, 2592u * spin_of(e_cur), frag_of(G, e_cur & 255, spin_of(e_cur)), lane_main, lane_rem, lane16);
#pragma unroll 1
        for (;;) {
            const int gn = g + gstep;
            const bool has_next = gn < gend;
            if (has_next) load_group(gn, Gn); else Gn = G;         // (no next group: the final prefetch re-reads this group's first entry, discarded)
            S5Acc<WIDE> acc;
#pragma unroll
            for (int t = 0; t < 9; ++t) { acc.m[t] = (double4_t){0, 0, 0, 0}; acc.r[t] = 0.0; }
            const int n = G.M[0];
#pragma unroll 1
            for (int j = 0; j < n; ++j) {
                const bool last = (j + 1 == n);
                const int e_nxt = last ? Gn.M[1] : G.M[2 + j];
                if (last) load_idx(Gn, e_nxt & 255, nraw, nrem); else load_idx(G, e_nxt & 255, nraw, nrem);
                const int si = spin_of(e_cur);
                const unsigned so = 2592u * si;
                const char* __restrict__ fb = frag_of(G, e_cur & 255, si);
                s5_load_single<WIDE>(Z, cur, so, fb, lane_single, lane_rem_single, lane8);
                s5_mfma_pair(acc, X);
                s5_interleave<NL, NM_PAIR>();
                __builtin_amdgcn_sched_barrier(0);
                make_slot(nraw, nrem, cur, e_nxt & 255);
                const char* __restrict__ fbn = last ? frag_of(Gn, e_nxt & 255, spin_of(e_nxt)) : frag_of(G, e_nxt & 255, spin_of(e_nxt));
                s5_load_pair<0, WIDE>(X, cur, 2592u * spin_of(e_nxt), fbn, lane_main, lane_rem, lane16);
                s5_mfma_pair(acc, Y);
                s5_interleave<NL, NM_PAIR>();
                __builtin_amdgcn_sched_barrier(0);
                s5_load_pair<1, WIDE>(Y, cur, 2592u * spin_of(e_nxt), fbn, lane_main, lane_rem, lane16);
                s5_mfma_single(acc, Z);
                s5_interleave<NL, NM_SINGLE>();
                __builtin_amdgcn_sched_barrier(0);
                e_cur = e_nxt;
            }
            // results of group G (the next group's first operands are in flight meanwhile)
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int a = (t < 8) ? G.atom[t] : G.rem_atom;
                if (a == zero_block) continue;
                double* ob = out + (size_t)BLD * a + 324 * sig + ((t < 8) ? 2 * l15 : 32 + 2 * (l15 & 1));
#pragma unroll
                for (int p = 0; p < 2; ++p) {
                    s5_d2 v; v[0] = acc.m[t][2 * p]; v[1] = acc.m[t][2 * p + 1];
                    *reinterpret_cast<s5_d2*>(ob + 36 * (4 * p + l4)) = v;
                }
                if (l4 < 2) ob[288 + l4] = acc.r[t];
            }
            if (!has_next) break;
            G = Gn;
            g = gn;
        }
    }
}

}  // namespace rsrec
