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
//   * K packed across neighbour slots: a spin has 9 complex = 18 real inputs per block, 4.5 MFMA k-steps; with one slot per
//     three steps (4 + 4 + 1 orbitals, the fifth k-step half empty) 10 % of the matrix work multiplied padding.  The schedule of a
//     group is now ONE STREAM OF ORBITALS -- entry after entry, 9 orbitals each -- cut into triples of steps that take 4, 4 and 2
//     orbitals (two k-steps, two k-steps, one k-step): a step may end one entry and begin the next (its lanes then gather from two
//     different neighbour blocks), the operator fragments are stored in stream order, and nothing but the tail is padded.
//     (-10 % matrix instructions; measured 4.90 -> 4.65 ms per level on 64 x 22^3, spin-mixing stencil 9.0 -> 8.1 ms.)  The step
//     pattern repeats after 10 entries = 27 steps and is unrolled statically; the steps of the tail are skipped one by one.
//     Control-flow forms of that unrolled period that were tried: one exit per step / per triple (also with a private copy of the
//     result stores per exit), the whole period as one block with the tail in a second copy, a wave-uniform switch over nine static
//     triples -- all made the register allocator spill accumulator tuples; a run-time triple (entry boundaries as scalar state,
//     per-lane block select in every step) kept its registers but paid 3 vector-ALU instructions per tile and step: 4.99 ms.
//     The skippable steps cost exactness of the compiler's vmcnt waits at the joins (it has to assume the shortest path).  An
//     all-pair stream (steps of 4 orbitals only, period 4 entries = 9 steps that can run without branches) needs a third two-k-step
//     operand set: 18 registers over the 256 of a wave at two waves per SIMD.
//   * operator stream in LDS + persistent workgroups (LDSA): the CU's texture addresser was 83 % busy with the 11 loads per step, two of
//     which fetch fragments that are the same for every group of an operator class.  256 workgroups (one per CU, 8 waves of one spin)
//     copy that stream to LDS once and then take groups one at a time from per-(chain, XCD) counters: -2.8 % (plain), -7.6 % (hoh),
//     -7.5 % (spin-mixing stencil, both spins on every XCD).  With one-round workgroups the copy cost what the reads saved.
// Tried on top of this and not adopted: a wave walking several groups with the next group's first operands requested during the
// last entry of the current one (the group prologue -- a chain of dependent loads -- then overlaps matrix work): 1.5 % / 4 % / 7 %
// SLOWER at 2 / 3 / 4 groups per wave; many short one-group workgroups that the hardware dispatcher balances win.
#pragma once
#include <hip/hip_runtime.h>
#include <algorithm>
#include <type_traits>
#include <vector>
#include "kernels_valu.hpp"
#include "kernels_mfma.hpp"
#include "kernels_spmm4.hpp"

namespace rsrec {

#ifndef RSREC_S5_KSPLIT_PROBE
#define RSREC_S5_KSPLIT_PROBE 0
#endif
#ifndef RSREC_S5_PRIO
#define RSREC_S5_PRIO 0
#endif
#ifndef S5_WG_GROUPS
#define S5_WG_GROUPS 4   // groups of 8 atoms per workgroup (x 2 spin waves each): 4 -> 512 threads
#endif
// Operator fragments in STREAM ORDER.  The schedule of an operator class is a list of entries (neighbour column, spin part); entry j
// contributes the orbitals 9 j .. 9 j + 8 of its input spin to the stream.  Triple T of steps covers the stream orbitals
//   X: 10 T .. +3 (two k-steps: real, imaginary parts)   Y: 10 T + 4 .. +7 (two k-steps)   Z: 10 T + 8, + 9 (ONE k-step: re, im, re, im)
constexpr int S5_TRIPLE = 640;                                    // doubles per triple: X [rows 0..15: 128 | rows 16..19: 128], Y [128 | 128], Z [64 | 64]
constexpr int S5_TRIPLE_BYTES = S5_TRIPLE * 8;
constexpr int S5_HEAD_TRIPLES = 2;                                 // the triples that hold the two extra entries (orbitals 0..17)
constexpr int S5_HEAD_DOUBLES = 2 * S5_HEAD_TRIPLES * S5_TRIPLE;   // per-chain head of a stream: [sigma_out][2 triples][640]
constexpr int S5_MAXENT = 2 * S4_MAXSLOTS + 2;                     // diagonal and spin-flip part of every slot + the two extra entries
constexpr int S5_ENTPAD = 8;                                       // null entries behind the list (operands are requested two steps ahead)

// Device-side assembly of the operator streams (our side of the boundary): block (set, tau, slot) = sign * a [+ b] [+ diag * 1],
// read from the RAW blocks as the caller handed them over (already on the device for the VALU kernel set): h itself, -(h o) with the
// identity added on-site, e_nu + l.s.  a == nullptr: the class has no such block.
struct S5Desc { const double* a; const double* b; double sign; double diag; };

__device__ __forceinline__ double s5_desc_value(const S5Desc& D, int ro, int ri, int c /*0 re, 1 im*/) {
    double v = D.sign * D.a[2 * (ro + 18 * ri) + c];
    if (D.b) v = v + D.b[2 * (ro + 18 * ri) + c];
    if (D.diag != 0.0 && ro == ri && c == 0) v = v + D.diag;
    return v;
}

// One thread per fragment double: grid (ntr, 2 sigma_out, nset * ntau), 640 threads.  Same index arithmetic as Spmm5Operator::emit_stream
// (the host version, kept for the general tables of the Kubo / local-axis paths): bitwise the same streams.
__global__ __launch_bounds__(640) void k_s5_emit(const S5Desc* __restrict__ desc, const int* __restrict__ meta, int meta_stride, int nfs, int null_col, int ntr,
                                                 double* __restrict__ frag) {
    const int t = blockIdx.x, sig = blockIdx.y, st = blockIdx.z;
    const int* codes = meta + (size_t)st * meta_stride + 2;
    const S5Desc* D = desc + (size_t)st * nfs;
    const int e0 = threadIdx.x;                               // 0..639 inside the triple
    const int K = e0 < 256 ? 0 : e0 < 512 ? 1 : 2;
    const int r = e0 - (K == 0 ? 0 : K == 1 ? 256 : 512);
    int q, l, e;
    if (K < 2) { q = r >> 7; l = (r & 127) >> 1; e = r & 1; } else { q = r >> 6; l = r & 63; e = 0; }
    const int k = l >> 4, rho = q == 0 ? (l & 15) : 16 + (l & 3);
    // decode20
    const int stq = rho >> 2, ll = rho & 3;
    int po = 0, mo = 0;
    bool row = true;
    if (stq < 4) { po = stq & 1; mo = 4 * (stq >> 1) + ll; } else if (ll < 2) { po = ll; mo = 8; } else row = false;
    const int o0 = 10 * t + (K == 0 ? 0 : K == 1 ? 4 : 8);
    const int o = K < 2 ? o0 + k : o0 + (k >> 1);
    const int pi = K < 2 ? e : (k & 1);
    double v = 0.0;
    if (row) {
        const int j = o / 9, code = codes[j], col = code & 255, flip = code >> 8;
        if (col != null_col && D[col].a) {
            const int so = sig, si = flip ? 1 - sig : sig, mi = o % 9;
            const int ro = 9 * so + mo, ri = 9 * si + mi;
            const double hr = s5_desc_value(D[col], ro, ri, 0), hi = s5_desc_value(D[col], ro, ri, 1);
            v = (po == pi) ? hr : (po == 0 ? -hi : hi);
        }
    }
    frag[(((size_t)st * 2 + sig) * ntr + t) * S5_TRIPLE + e0] = v;
}

struct Spmm5Operator {
    double* d_frag = nullptr;    // [set][tau][sigma_out][ntr][640]
    int* d_meta = nullptr;       // [set][tau][META]: number of steps, number of extra entries (0 / 2), entry codes column | flip << 8
    size_t frag_bytes = 0, meta_bytes = 0;
    S5Desc* d_desc = nullptr;    // device-side assembly: [set][tau][nslots + 1]
    size_t desc_bytes = 0;
    std::vector<signed char> sched_sig;   // block structure (absent / spin-diagonal / spin-mixing per block) the uploaded schedule was built for
    int sched_dims[3] = {0, 0, 0};
    int sched_epoch = 0;         // counts rebuilt schedules: whatever holds the schedule's shape by value (a captured graph) is keyed by it
    int ntau = 0, nslots = 0, have_o = 0, ntr = 0, spin_mixing = 0;   // spin_mixing: some regular (hopping) block has a spin-flip part
    static constexpr int META = 2 + S5_MAXENT + S5_ENTPAD;
    struct Entry { const double* blk; int col; int flip; };        // blk == nullptr: null entry (zero fragments, reads the zero block)
    struct Head { std::vector<double> blk; int flip = 0; bool valid = false; };
    std::vector<Head> head_main;   // [set][tau]: first regular entry of the schedule (its first two orbitals share triple 1 with the extra entries)
    std::vector<int> ksteps;       // [set][tau]: MFMA k-steps a wave runs per group (X, Y steps: 2, Z steps: 1)
    std::vector<signed char> mixing;   // [set][tau][nslots + 1]: -1 absent, 0 spin-diagonal block (one quadrant pair: 23 328 flop), 1 spin-mixing (46 656 flop)

    void release() {
        if (d_frag) (void)hipFree(d_frag);
        if (d_meta) (void)hipFree(d_meta);
        if (d_desc) (void)hipFree(d_desc);
        d_frag = nullptr; d_meta = nullptr; d_desc = nullptr; frag_bytes = meta_bytes = desc_bytes = 0;
        sched_sig.clear();
    }
    // row rho (0..19) of an output spin's padded real form: rho = 4 s + l; s < 4 -> (part s & 1, m = 4 (s >> 1) + l); s = 4 -> l = 0: (re, m = 8),
    // l = 1: (im, m = 8), l = 2, 3: padding.  (The 16x16x4 result register j of lane row l4 is row l4 + 4 j: registers (2 p, 2 p + 1) are the
    // real and imaginary part of element m = 4 p + l4 -> one 16-byte store in the CI layout.)
    static bool decode20(int rho, int& part, int& m) {
        const int st = rho >> 2, l = rho & 3;
        if (st < 4) { part = st & 1; m = 4 * (st >> 1) + l; return true; }
        if (l < 2) { part = l; m = 8; return true; }
        return false;
    }
    // real form of a complex block (column-major interleaved): row (spin so, part po, m mo), column (spin si, part pi, m mi)
    static double hreal(const double* blk, int so, int po, int mo, int si, int pi, int mi) {
        const int ro = 9 * so + mo, ri = 9 * si + mi;
        const double hr = blk[2 * (ro + 18 * ri)], hi = blk[2 * (ro + 18 * ri) + 1];
        if (po == pi) return hr;
        return po == 0 ? -hi : hi;
    }
    static int steps_of(int nent) {
        const int n = 9 * nent, r = n % 10;
        return 3 * (n / 10) + (r == 0 ? 0 : r <= 4 ? 1 : r <= 8 ? 2 : 3);
    }
    // fragments of triples t0 .. t0 + nt - 1 of the stream of `E` for output spin `sig`: A operand of the 16x16x4 MFMA for rows 0..15
    // (lane (l15 = row, l4 = k)) and of the 4x4x4 MFMA for rows 16..19 (lane (i + 4 g + 16 k) = A[16 + i][k], same for the 4 blocks g);
    // k-step lane k of X / Y = stream orbital o0 + k (two k-steps: real / imaginary part of the input), of Z = orbital o0 + (k >> 1), part k & 1
    static void emit_stream(const std::vector<Entry>& E, int sig, int t0, int nt, double* out) {
        auto value = [&](int o, int pi, int po, int mo) {
            const int j = o / 9;
            if (j >= (int)E.size() || !E[j].blk) return 0.0;
            return hreal(E[j].blk, sig, po, mo, E[j].flip ? 1 - sig : sig, pi, o % 9);
        };
        for (int t = t0; t < t0 + nt; ++t) {
            double* T = out + (size_t)(t - t0) * S5_TRIPLE;
            for (int K = 0; K < 3; ++K) {
                const int base = K == 0 ? 0 : K == 1 ? 256 : 512, o0 = 10 * t + (K == 0 ? 0 : K == 1 ? 4 : 8);
                for (int q = 0; q < 2; ++q)
                    for (int l = 0; l < 64; ++l) {
                        const int k = l >> 4, rho = q == 0 ? (l & 15) : 16 + (l & 3);
                        int po, mo;
                        const bool row = decode20(rho, po, mo);
                        if (K < 2) for (int e = 0; e < 2; ++e) T[base + 128 * q + 2 * l + e] = row ? value(o0 + k, e, po, mo) : 0.0;
                        else T[base + 64 * q + l] = row ? value(o0 + (k >> 1), k & 1, po, mo) : 0.0;
                    }
            }
        }
    }
    // Schedule of every (set, tau) from the block STRUCTURE alone: kind[(set * ntau + tau) * (nslots + 1) + s] = -1 absent, 0 spin-diagonal,
    // 1 spin-mixing (s = nslots: the extra on-site slot that reads the second input vector).  The two extra entries first (spin-diagonal
    // and spin-flip part; a null entry if the block has no such part -- the kernel switches from the second input to the first after
    // orbital 18, a step boundary), then the spin-diagonal part of every block, plus the spin-flip part of the blocks that have one.
    // Fills meta / ksteps / mixing / ntr and uploads the meta table; `sched` (optional) receives the entry lists as (column, flip) pairs,
    // column -1 = null entry.  Unchanged structure (every SCF iteration after the first): nothing is rebuilt or uploaded.
    const char* make_schedule(int nslots_lat, int ntau_, int nset, const std::vector<signed char>& kind, std::vector<std::vector<std::pair<int, int>>>* sched_out) {
        if (nslots_lat + 1 > S4_MAXSLOTS) return "too many neighbour slots for the spmm5 kernel";
        const bool same = !sched_out && d_meta && sched_dims[0] == nslots_lat && sched_dims[1] == ntau_ && sched_dims[2] == nset && sched_sig == kind;
        if (same) return nullptr;
        ++sched_epoch;
        ntau = ntau_; nslots = nslots_lat; have_o = nset > 1 ? 1 : 0; spin_mixing = 0;
        const int nfs = nslots + 1, null_col = nslots + 1;
        std::vector<std::vector<std::pair<int, int>>> sched((size_t)nset * ntau);
        std::vector<int> meta((size_t)nset * ntau * META, 0);
        ksteps.assign((size_t)nset * ntau, 0);
        mixing = kind;
        int maxent = 0;
        for (int set = 0; set < nset; ++set)
            for (int tau = 0; tau < ntau; ++tau) {
                std::vector<std::pair<int, int>>& E = sched[(size_t)set * ntau + tau];
                const signed char* Kd = kind.data() + ((size_t)set * ntau + tau) * nfs;
                int* M = meta.data() + ((size_t)set * ntau + tau) * META;
                if (Kd[nslots] >= 0) {
                    E.push_back({nslots, 0});
                    E.push_back({Kd[nslots] == 1 ? nslots : -1, 1});
                    M[1] = 2;
                }
                for (int s = 0; s < nslots; ++s) {
                    if (Kd[s] < 0) continue;
                    E.push_back({s, 0});
                    if (Kd[s] == 1) { E.push_back({s, 1}); if (s > 0) spin_mixing = 1; }
                }
                M[0] = steps_of((int)E.size());
                ksteps[(size_t)set * ntau + tau] = 5 * (M[0] / 3) + 2 * (M[0] % 3);
                for (int j = 0; j < S5_MAXENT + S5_ENTPAD; ++j)
                    M[2 + j] = j < (int)E.size() ? ((E[j].first >= 0 ? E[j].first : null_col) | (E[j].second << 8)) : null_col;
                maxent = std::max(maxent, (int)E.size());
            }
        ntr = (9 * maxent + 9) / 10 + 1;                       // + one zero triple: the last steps request operands beyond the end
        const size_t need = (size_t)nset * ntau * 2 * ntr * S5_TRIPLE * sizeof(double), mneed = meta.size() * sizeof(int);
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
        if (hipMemcpy(d_meta, meta.data(), mneed, hipMemcpyHostToDevice) != hipSuccess) return "upload of spmm5 schedule failed";
        sched_sig = kind; sched_dims[0] = nslots_lat; sched_dims[1] = ntau_; sched_dims[2] = nset;
        if (sched_out) sched_out->swap(sched);
        return nullptr;
    }
    static std::vector<signed char> kinds_of(const std::vector<const double*>& blk) {
        std::vector<signed char> kind(blk.size(), -1);
        for (size_t q = 0; q < blk.size(); ++q)
            if (blk[q]) kind[q] = Spmm4Operator::pattern_of(blk[q]) == 0 ? 1 : 0;
        return kind;
    }
    // General table: blk[(set * ntau + tau) * (nslots + 1) + s] = column-major interleaved 18x18 complex block of operator class tau,
    // slot s (s = nslots: the extra on-site slot that reads the second input vector), or nullptr = absent (contributes nothing and is
    // not scheduled).  Fragments swizzled on the HOST (the general tables of the Kubo velocity operators, the local-axis operator, the
    // plain operator under hoh; the Hamiltonian itself takes build() -> device assembly).
    const char* build_custom(int nslots_lat, int ntau_, int nset, const std::vector<const double*>& blk) {
        std::vector<std::vector<std::pair<int, int>>> sched;
        if (const char* msg = make_schedule(nslots_lat, ntau_, nset, kinds_of(blk), &sched)) return msg;
        const int nfs = nslots + 1;
        head_main.assign((size_t)nset * ntau, Head());
        const size_t per_sig = (size_t)ntr * S5_TRIPLE, per_set = (size_t)ntau * 2 * per_sig;
        std::vector<double> host(per_set * nset, 0.0);
        for (int set = 0; set < nset; ++set)
            for (int tau = 0; tau < ntau; ++tau) {
                const double* const* B = blk.data() + ((size_t)set * ntau + tau) * nfs;
                std::vector<Entry> E;
                for (const auto& en : sched[(size_t)set * ntau + tau]) E.push_back({en.first >= 0 ? B[en.first] : nullptr, en.first >= 0 ? en.first : nslots, en.second});
                const int nextra = (B[nslots] != nullptr) ? 2 : 0;
                if ((int)E.size() > nextra) {
                    Head& H = head_main[(size_t)set * ntau + tau];
                    H.blk.assign(E[nextra].blk, E[nextra].blk + 2 * BLK); H.flip = 0; H.valid = true;
                }
                for (int sig = 0; sig < 2; ++sig)
                    emit_stream(E, sig, 0, (9 * (int)E.size() + 9) / 10, host.data() + set * per_set + ((size_t)tau * 2 + sig) * per_sig);
            }
        if (hipMemcpy(d_frag, host.data(), host.size() * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) return "upload of spmm5 fragments failed";
        return nullptr;
    }
    // Head of the stream of (set, tau) with `extra` as the block of the extra on-site slot (both spin parts scheduled): the per-chain
    // fragments of local-axis runs.  out: [sigma_out][2 triples][640]
    void emit_head(int set, int tau, const double* extra, double* out) const {
        const Head& H = head_main[(size_t)set * ntau + tau];
        std::vector<Entry> E = {{extra, nslots, 0}, {extra, nslots, 1}};
        if (H.valid) E.push_back({H.blk.data(), 0, H.flip});
        for (int sig = 0; sig < 2; ++sig) emit_stream(E, sig, 0, S5_HEAD_TRIPLES, out + (size_t)sig * S5_HEAD_TRIPLES * S5_TRIPLE);
    }
    // The Hamiltonian itself.  Set 0: the blocks of h (slot 0 carries + l.s when !hoh).  Set 1 (hoh second pass) is built so that
    // ONE SpMM pass over hpsi = h psi plus one extra on-site slot reading psi gives the whole
    //   H psi = hpsi - (h o) hpsi + (e_nu + l.s) psi   (recursion.f90:1543):
    // slot 0 -> 1 - (h o)_0,  slot s -> -(h o)_s,  slot `nslots` (extra) -> enim + lsham of the atom's type.
    // st / loc / eeo / hallo / enim / lsham: host arrays (st, loc with l.s folded into slot 0 when !hoh).  dev != nullptr: the same arrays
    // on the device, [0] st, [1] loc, [2] eeo, [3] hallo, [4] enim, [5] lsham -- the streams are then assembled THERE (k_s5_emit) from
    // descriptors, and the host only looks at the block structure; with an unchanged structure (every SCF iteration after the first)
    // a call is one small descriptor upload + one kernel.
    const char* build(int nslots_lat, int hstride, int ntype, int nmax, int hoh, const double* st, const double* loc, const double* eeo, const double* hallo,
                      const double* enim, const double* lsham, const int* iz0, const double* const* dev = nullptr, hipStream_t stream = nullptr) {
        const int nt = nmax + ntype, nset = hoh ? 2 : 1, nfs = nslots_lat + 1;
        const size_t B = 2 * (size_t)BLK;
        std::vector<const double*> blk((size_t)nset * nt * nfs, nullptr);
        // set 1 on the host: -(h o)_s (+ 1 on-site) per slot when the streams are swizzled here; with device assembly only the extra on-site
        // slot e_nu + l.s of every class is formed here (for the block structure), the rest stays a descriptor
        const size_t tmp_per = dev ? 1 : (size_t)nfs;
        std::vector<double> tmp((size_t)(hoh ? nt : 0) * tmp_per * B, 0.0);
        std::vector<S5Desc> desc(dev ? (size_t)nset * nt * nfs : 0, S5Desc{nullptr, nullptr, 1.0, 0.0});
        for (int tau = 0; tau < nt; ++tau)
            for (int s = 0; s < nslots_lat; ++s) {
                const size_t off = tau < nmax ? B * (s + (size_t)hstride * tau) : B * (s + (size_t)hstride * (tau - nmax));
                const double* h0 = (tau < nmax ? loc : st) + off;
                blk[(size_t)tau * nfs + s] = h0;
                if (dev) desc[(size_t)tau * nfs + s] = S5Desc{(tau < nmax ? dev[1] : dev[0]) + off, nullptr, 1.0, 0.0};
                if (hoh) {
                    const double* ho = (tau < nmax ? hallo : eeo) + off;
                    if (dev) { blk[((size_t)nt + tau) * nfs + s] = ho; desc[((size_t)nt + tau) * nfs + s] = S5Desc{(tau < nmax ? dev[3] : dev[2]) + off, nullptr, -1.0, s == 0 ? 1.0 : 0.0}; }
                    else {
                        double* d = tmp.data() + B * ((size_t)tau * tmp_per + s);
                        for (size_t e = 0; e < B; ++e) d[e] = -ho[e];
                        if (s == 0) for (int q = 0; q < NB; ++q) d[2 * (q + NB * q)] += 1.0;
                        blk[((size_t)nt + tau) * nfs + s] = d;
                    }
                }
            }
        if (hoh)
            for (int tau = 0; tau < nt; ++tau) {
                const int ty = tau < nmax ? iz0[tau] : tau - nmax;
                double* d = tmp.data() + B * ((size_t)tau * tmp_per + (dev ? 0 : nslots_lat));
                for (size_t e = 0; e < B; ++e) d[e] = enim[B * ty + e] + lsham[B * ty + e];
                blk[((size_t)nt + tau) * nfs + nslots_lat] = d;
                if (dev) desc[((size_t)nt + tau) * nfs + nslots_lat] = S5Desc{dev[4] + B * ty, dev[5] + B * ty, 1.0, 0.0};
            }
        if (!dev) return build_custom(nslots_lat, nt, nset, blk);
        // device assembly: the structure from the host copies (the negation and the added identity of set 1 change no spin-flip entry)
        if (const char* msg = make_schedule(nslots_lat, nt, nset, kinds_of(blk), nullptr)) return msg;
        head_main.assign((size_t)nset * nt, Head());
        const size_t dneed = desc.size() * sizeof(S5Desc);
        if (dneed > desc_bytes) {
            if (d_desc) (void)hipFree(d_desc);
            d_desc = nullptr; desc_bytes = 0;
            if (hipMalloc(reinterpret_cast<void**>(&d_desc), dneed) != hipSuccess) return "hipMalloc of spmm5 block descriptors failed";
            desc_bytes = dneed;
        }
        if (hipMemcpyAsync(d_desc, desc.data(), dneed, hipMemcpyHostToDevice, stream) != hipSuccess) return "upload of spmm5 block descriptors failed";
        if (hipStreamSynchronize(stream) != hipSuccess) return "upload of spmm5 block descriptors failed";      // desc is a local
        k_s5_emit<<<dim3(ntr, 2, nset * nt), S5_TRIPLE, 0, stream>>>(d_desc, d_meta, META, nfs, nslots_lat + 1, ntr, d_frag);
        if (hipGetLastError() != hipSuccess) return "launch of the operator-stream assembly failed";
        return nullptr;
    }
    // the one operator class of set `set` that has a schedule, or -1 if several have (then groups of different classes meet in a launch)
    int single_class(int set) const {
        int one = -1;
        for (int tau = 0; tau < ntau; ++tau)
            if (ksteps[(size_t)set * ntau + tau] > 0) { if (one >= 0) return -1; one = tau; }
        return one;
    }
    // flops the block structure REQUIRES for one multiplication by block (set, tau, slot): a spin-diagonal block (hopping block of a
    // collinear magnet, hamiltonian.f90:1553-1617) is two 9x9 complex quadrants = half of the reference's 18x18x18 zgemm (recursion.f90:1618)
    double required_flops(int set, int tau, int slot) const {
        const size_t q = ((size_t)set * ntau + tau) * (nslots + 1) + slot;
        if (q >= mixing.size() || mixing[q] < 0) return 0.0;
        return mixing[q] ? 46656.0 : 23328.0;
    }
    // matrix flops one group of 8 atoms of operator class tau costs in set `set` (both spin waves): per k-step nine tiles of one 16x16x4
    // and one 4x4x4 (4 blocks) MFMA
    double flops_per_group(int set, int tau) const { return ksteps.empty() ? 0.0 : 2.0 * ksteps[(size_t)set * ntau + tau] * 9.0 * (2.0 * 16 * 16 * 4 + 2.0 * 4 * 4 * 4 * 4); }
    const double* frag_set(int set) const { return d_frag + (size_t)set * ntau * 2 * ntr * S5_TRIPLE; }
    const int* meta_set(int set) const { return d_meta + (size_t)set * ntau * META; }
};

typedef double s5_d2 __attribute__((ext_vector_type(2)));
struct S5Pair { s5_d2 b[9]; s5_d2 a[2]; };      // operands of an X / Y step (two k-steps): nine psi tiles; operator rows 0..15 and 16..19
struct S5Single { double b[9]; double a[2]; };  // operands of a Z step (one k-step)
struct S5Acc { double4_t m[9]; double r[9]; };  // per tile: 16x16x4 result (rows 0..15) and 4x4x4 result (rows 16..19)

// addressing state of one schedule entry: byte offsets (from the input vector, spin half included) of the neighbour blocks of the 8 atom
// tiles (wave-uniform) and of this lane's atom of the remainder tile
struct S5Ent { unsigned off[GROUP]; unsigned rem; };

// step S (0..26) of a period of 10 entries = 90 stream orbitals = 9 triples
template <int S> struct S5St {
    static constexpr int T = S / 3, K = S % 3;
    static constexpr int W = K == 2 ? 2 : 4;                                  // orbitals of the step
    static constexpr int o0 = 10 * T + (K == 0 ? 0 : K == 1 ? 4 : 8);
    static constexpr int eA = o0 / 9, mA0 = o0 % 9, sp = 9 - mA0;             // entry of the first orbital; orbitals >= sp belong to entry eA + 1
    static constexpr bool straddle = sp < W;
    static constexpr bool opens = straddle || mA0 == 0;                       // an entry begins in this step:
    static constexpr int eOpen = mA0 == 0 ? eA : eA + 1;                      //   this one
};

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

template <int NL = 11>
__device__ __forceinline__ void s5_mfma(S5Acc& acc, const S5Pair& o) {
#pragma unroll
    for (int e = 0; e < 2; ++e)
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            acc.m[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(o.a[0][e], o.b[t][e], acc.m[t], 0, 0, 0);
            acc.r[t] = __builtin_amdgcn_mfma_f64_4x4x4f64(o.a[1][e], o.b[t][e], acc.r[t], 0, 0, 0);
        }
    s5_interleave<NL, 36>();
}
template <int NL = 11>
__device__ __forceinline__ void s5_mfma(S5Acc& acc, const S5Single& o) {
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        acc.m[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(o.a[0], o.b[t], acc.m[t], 0, 0, 0);
        acc.r[t] = __builtin_amdgcn_mfma_f64_4x4x4f64(o.a[1], o.b[t], acc.r[t], 0, 0, 0);
    }
    s5_interleave<NL, 18>();
}

template <int N> using S5C = std::integral_constant<int, N>;
#define S5_GLOBAL __attribute__((address_space(1)))
#define S5_CONST __attribute__((address_space(4)))
#define S5_LDS __attribute__((address_space(3)))
typedef const S5_GLOBAL char* s5_gp;      // explicit global address space: the loads stay global_load behind the scalar-base barriers below

// The whole stream of a group for one wave (output spin `sig`).  Operands are requested TWO steps ahead into three register sets
// (X, Y: two k-steps; Z: one): while step s runs, the operands of s + 1 are in flight and those of s + 2 are issued into the set that
// step s - 1 consumed.  The step pattern repeats after 10 entries (27 steps), so which steps end one entry and begin the next, and at
// which orbital, is static: such a step selects per lane between the neighbour blocks of two entries, every other step addresses its
// blocks through wave-uniform bases.  An entry's neighbour indices are scalar loads issued when the entry before it is opened.
// TWO: the first two entries (18 orbitals = steps 0..4 of the first period) are the extra on-site slot; they read the second input
// vector in2b (hoh second pass, recursion.f90:1543: the (e_nu + l.s) term acts on psi itself; local-axis runs: the per-chain on-site
// term), and if fr_head is given the first two triples of fragments come from that per-chain table.
// OCT: the 8 tiles are 8 CHAINS on one atom (atom[] holds that atom 8 times, or the zero block for a chain that does not exist): tile t reads
// the neighbour block of chain c0 + t, which lies tile_blk[t] = t (kk + 1) blocks behind chain c0's in the batch's vectors.
template <bool TWO, bool LDSA, bool OCT = false>
__device__ __forceinline__ void s5_run_stream(S5Acc& acc, const int* __restrict__ meta, const char* __restrict__ fr /*LDSA: the staged stream in LDS*/, const char* __restrict__ fr_head,
                                              const char* __restrict__ inb, const char* __restrict__ in2b,
                                              const int* __restrict__ nbr5 /*(kk+1) x ncol: absent -> zero block, column nslots = self, nslots + 1 = zero block*/,
                                              const int (&atom)[GROUP] /*padding -> zero block*/, int rem_sel /*per lane: tile (atom of the group) of its remainder column*/,
                                              int ncol, int sig, int l4,
                                              unsigned lane_main, unsigned lane_z, unsigned lane_rem, unsigned lane_rem_z, unsigned lane16, unsigned lane8,
                                              const int (&tile_blk)[GROUP], int rem_blk /*per lane: tile_blk of its remainder tile*/, int kpart = 0) {
    int left = meta[0];
    int ebase = 0, tbase = 0;
    bool first = true;
#if RSREC_S5_KSPLIT_PROBE
    // TIMING PROBE ONLY (results invalid): a task is one PERIOD-ALIGNED part of a group's stream -- part 0 the first 27 steps (10 entries),
    // part 1 the rest -- so that twice as many waves work on one group and an XCD has half as many atoms in flight
    if (kpart == 0) left = min(left, 27);
    else { left -= 27; ebase = 10; tbase = 9; first = false; }
#endif
    if (left <= 0) return;
    const bool extras = TWO && meta[1] != 0;
    // the schedule and the neighbour table through the constant address space: scalar loads (a vector load + readfirstlane here put a
    // vmcnt wait on an old load into the step loop, which also drained the operand loads issued after it)
    const S5_CONST int* codes = (const S5_CONST int*)(meta + 2);
    const S5_CONST int* nbr = (const S5_CONST int*)nbr5;
    S5Ent E0, E1;
    int raw[GROUP];
    int code_cur = codes[ebase], code_nxt = codes[ebase + 1];
    auto load_idx = [&](int col) {
#pragma unroll
        for (int t = 0; t < GROUP; ++t) raw[t] = nbr[(size_t)ncol * atom[t] + col];
    };
    load_idx(code_cur & 255);
    unsigned vlane_main = lane_main, vlane_z = lane_z, vlane16 = lane16, vlane8 = lane8;
    // entry j (its indices are in raw, its code in code_cur) becomes addressable; the indices of entry j + 1 are requested
    auto open_entry = [&](int j, S5Ent& E) {
        const unsigned so = 2592u * (unsigned)((code_cur >> 8) ? 1 - sig : sig);
#pragma unroll
        for (int t = 0; t < GROUP; ++t) E.off[t] = (unsigned)(OCT ? raw[t] + tile_blk[t] : raw[t]) * (BLD * 8u) + so;
        // remainder tile: this lane's atom is tile rem_sel = l15 >> 1 of the group -- selected from the scalar indices
        const int r01 = (rem_sel & 1) ? raw[1] : raw[0], r23 = (rem_sel & 1) ? raw[3] : raw[2], r45 = (rem_sel & 1) ? raw[5] : raw[4], r67 = (rem_sel & 1) ? raw[7] : raw[6];
        const int r03 = (rem_sel & 2) ? r23 : r01, r47 = (rem_sel & 2) ? r67 : r45;
        E.rem = (unsigned)(((rem_sel & 4) ? r47 : r03) + (OCT ? rem_blk : 0)) * (BLD * 8u) + so;
        code_cur = code_nxt;
        code_nxt = codes[j + 2];
        load_idx(code_cur & 255);
    };
    auto issue = [&](auto s2c, auto& o) {
        constexpr int S2 = decltype(s2c)::value, S = S2 % 27, wrap = S2 / 27;
        using I = S5St<S>;
        S5Ent& EA = (I::eA & 1) ? E1 : E0;
        S5Ent& EB = (I::eA & 1) ? E0 : E1;
        if constexpr (I::opens) open_entry(ebase + 10 * wrap + I::eOpen, (I::eOpen & 1) ? E1 : E0);
        const bool head = wrap == 0 && first && extras;
        const char* __restrict__ base = (TWO && S < 5 && head) ? in2b : inb;
        s5_gp fbg = nullptr;
        if constexpr (!LDSA) {
            const char* fb = ((TWO && I::T < S5_HEAD_TRIPLES && head && fr_head) ? fr_head + I::T * S5_TRIPLE_BYTES
                                                                                 : fr + (size_t)(tbase + 9 * wrap + I::T) * S5_TRIPLE_BYTES)
                             + (I::K == 0 ? 0 : I::K == 1 ? 2048 : 4096);
            fbg = (s5_gp)fb;
            asm("" : "+s"(fbg));
        }
        constexpr unsigned rowA = 288u * I::mA0, rowB = 0u - 288u * I::sp;
        // the lane parts as values (re)defined in this step's block: instruction selection then folds base (SGPR pair) + lane (32-bit VGPR)
        // + immediate into the load (a zero-extension hoisted out of the loop made every address a 64-bit vector add).  They are
        // redefined IN PLACE (one register each for the whole loop): a per-step copy landed in a register of the operand set about to be
        // loaded, and the write-after-write check against that set's older loads became a vmcnt(0) at the top of every step.
        unsigned& lm = I::K < 2 ? vlane_main : vlane_z;
        unsigned& lf = I::K < 2 ? vlane16 : vlane8;
        const unsigned lr = I::K < 2 ? lane_rem : lane_rem_z;
        asm volatile("" : "+v"(lm));
        asm volatile("" : "+v"(lf));
        using V = std::remove_reference_t<decltype(o.b[0])>;
        if constexpr (!I::straddle) {
#pragma unroll
            for (int t = 0; t < GROUP; ++t) {
                s5_gp pt = (s5_gp)base + EA.off[t];
                asm("" : "+s"(pt));            // a wave-uniform base for the load's scalar address operand (not re-associated into a vector add)
                o.b[t] = *(const S5_GLOBAL V*)(pt + lm + rowA);
            }
            o.b[8] = *(const S5_GLOBAL V*)((s5_gp)base + (EA.rem + lr) + rowA);
        } else {
            const bool inB = (I::K < 2 ? l4 : (l4 >> 1)) >= I::sp;
#pragma unroll
            for (int t = 0; t < GROUP; ++t) {
                const unsigned off = (inB ? EB.off[t] + rowB : EA.off[t] + rowA) + lm;
                o.b[t] = *(const S5_GLOBAL V*)((s5_gp)base + off);
            }
            const unsigned off = (inB ? EB.rem + rowB : EA.rem + rowA) + lr;
            o.b[8] = *(const S5_GLOBAL V*)((s5_gp)base + off);
        }
        if constexpr (LDSA) {
            // operator fragments from the workgroup's LDS copy of the stream: LDS reads do not pass the texture addresser
            const S5_LDS char* fl = (const S5_LDS char*)fr + (unsigned)(tbase + 9 * wrap + I::T) * S5_TRIPLE_BYTES + (I::K == 0 ? 0 : I::K == 1 ? 2048 : 4096);
#pragma unroll
            for (int q = 0; q < 2; ++q) o.a[q] = *(const S5_LDS V*)(fl + lf + q * (I::K < 2 ? 1024 : 512));
        } else if constexpr (I::K < 2) {
#pragma unroll
            for (int q = 0; q < 2; ++q) o.a[q] = *(const S5_GLOBAL V*)(fbg + lf + q * 1024);
        } else {
#pragma unroll
            for (int q = 0; q < 2; ++q) o.a[q] = *(const S5_GLOBAL V*)(fbg + lf + q * 512);
        }
    };
    S5Pair X, Y;
    S5Single Z;
    issue(S5C<0>{}, X);
    issue(S5C<1>{}, Y);
#define S5_STEP(S, NEXT, CUR)                 \
    if (left > (S)) {                         \
        issue(S5C<(S) + 2>{}, NEXT);          \
        s5_mfma<LDSA ? 9 : 11>(acc, CUR);     \
        __builtin_amdgcn_sched_barrier(0);    \
    }
    for (;; left -= 27) {
        S5_STEP(0, Z, X) S5_STEP(1, X, Y) S5_STEP(2, Y, Z) S5_STEP(3, Z, X) S5_STEP(4, X, Y) S5_STEP(5, Y, Z)
        S5_STEP(6, Z, X) S5_STEP(7, X, Y) S5_STEP(8, Y, Z) S5_STEP(9, Z, X) S5_STEP(10, X, Y) S5_STEP(11, Y, Z)
        S5_STEP(12, Z, X) S5_STEP(13, X, Y) S5_STEP(14, Y, Z) S5_STEP(15, Z, X) S5_STEP(16, X, Y) S5_STEP(17, Y, Z)
        S5_STEP(18, Z, X) S5_STEP(19, X, Y) S5_STEP(20, Y, Z) S5_STEP(21, Z, X) S5_STEP(22, X, Y) S5_STEP(23, Y, Z)
        S5_STEP(24, Z, X) S5_STEP(25, X, Y) S5_STEP(26, Y, Z)
        if (left <= 27) break;
        ebase += 10; tbase += 9; first = false;
    }
#undef S5_STEP
}


// ---- split tasks (NSP = 3; LDS / persistent form): one wave = (group of 8 atoms, output spin, THIRD of the nine tiles) ----------------------
// A wave of the form above holds the nine tiles of a group for ~75 us; the 256 waves of an XCD are then spread over 1 024 atoms whose
// neighbourhood (9 MB) does not fit the XCD's 4 MB L2, and two 236-register waves per SIMD are all that can cover each other's stalls.
// Here a task is three tiles of a group -- part 0: atoms 0..2, part 1: atoms 3..5, part 2: atoms 6, 7 and the remainder tile (columns 16, 17 of
// all eight atoms) -- with 30 accumulator registers and three small operand sets: the waves of an XCD work on a third as many atoms per wave,
// and four waves fit a SIMD.  Every tile is accumulated exactly as in the nine-tile wave (same k order): results are bitwise the same.
template <int NT> struct S5PairN { s5_d2 b[NT]; s5_d2 a[2]; };
template <int NT> struct S5SingleN { double b[NT]; double a[2]; };
template <int NT> struct S5AccN { double4_t m[NT]; double r[NT]; };

template <int NT>
__device__ __forceinline__ void s5_mfma_n(S5AccN<NT>& acc, const S5PairN<NT>& o) {
#pragma unroll
    for (int e = 0; e < 2; ++e)
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            acc.m[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(o.a[0][e], o.b[t][e], acc.m[t], 0, 0, 0);
            acc.r[t] = __builtin_amdgcn_mfma_f64_4x4x4f64(o.a[1][e], o.b[t][e], acc.r[t], 0, 0, 0);
        }
    s5_interleave<NT, 4 * NT>();
}
template <int NT>
__device__ __forceinline__ void s5_mfma_n(S5AccN<NT>& acc, const S5SingleN<NT>& o) {
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        acc.m[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(o.a[0], o.b[t], acc.m[t], 0, 0, 0);
        acc.r[t] = __builtin_amdgcn_mfma_f64_4x4x4f64(o.a[1], o.b[t], acc.r[t], 0, 0, 0);
    }
    s5_interleave<NT, 2 * NT>();
}

struct S5EntN { unsigned off[3]; unsigned rem; };

// NMAIN atom tiles (wave-uniform neighbour blocks) + (HASREM) the remainder tile of the whole group (ratom: its eight atoms)
template <bool TWO, int NMAIN, bool HASREM>
__device__ __forceinline__ void s5_run_stream_n(S5AccN<NMAIN + (HASREM ? 1 : 0)>& acc, const int* __restrict__ meta, const char* __restrict__ fr /*the staged stream in LDS*/,
                                                const char* __restrict__ inb, const char* __restrict__ in2b, const int* __restrict__ nbr5,
                                                const int (&atom)[NMAIN], const int (&ratom)[GROUP], int rem_sel, int ncol, int sig, int l4,
                                                unsigned lane_main, unsigned lane_z, unsigned lane_rem, unsigned lane_rem_z, unsigned lane16, unsigned lane8) {
    constexpr int NT = NMAIN + (HASREM ? 1 : 0);
    int left = meta[0];
    int ebase = 0, tbase = 0;
    bool first = true;
    if (left <= 0) return;
    const bool extras = TWO && meta[1] != 0;
    const S5_CONST int* codes = (const S5_CONST int*)(meta + 2);
    const S5_CONST int* nbr = (const S5_CONST int*)nbr5;
    S5EntN E0, E1;
    int raw[NMAIN], rraw[GROUP];
    int code_cur = codes[ebase], code_nxt = codes[ebase + 1];
    auto load_idx = [&](int col) {
#pragma unroll
        for (int t = 0; t < NMAIN; ++t) raw[t] = nbr[(size_t)ncol * atom[t] + col];
        if constexpr (HASREM) {
#pragma unroll
            for (int t = 0; t < GROUP; ++t) rraw[t] = nbr[(size_t)ncol * ratom[t] + col];
        }
    };
    load_idx(code_cur & 255);
    unsigned vlane_main = lane_main, vlane_z = lane_z, vlane16 = lane16, vlane8 = lane8;
    auto open_entry = [&](int j, S5EntN& E) {
        const unsigned so = 2592u * (unsigned)((code_cur >> 8) ? 1 - sig : sig);
#pragma unroll
        for (int t = 0; t < NMAIN; ++t) E.off[t] = (unsigned)raw[t] * (BLD * 8u) + so;
        if constexpr (HASREM) {
            const int r01 = (rem_sel & 1) ? rraw[1] : rraw[0], r23 = (rem_sel & 1) ? rraw[3] : rraw[2], r45 = (rem_sel & 1) ? rraw[5] : rraw[4], r67 = (rem_sel & 1) ? rraw[7] : rraw[6];
            const int r03 = (rem_sel & 2) ? r23 : r01, r47 = (rem_sel & 2) ? r67 : r45;
            E.rem = (unsigned)((rem_sel & 4) ? r47 : r03) * (BLD * 8u) + so;
        }
        code_cur = code_nxt;
        code_nxt = codes[j + 2];
        load_idx(code_cur & 255);
    };
    auto issue = [&](auto s2c, auto& o) {
        constexpr int S2 = decltype(s2c)::value, S = S2 % 27, wrap = S2 / 27;
        using I = S5St<S>;
        S5EntN& EA = (I::eA & 1) ? E1 : E0;
        S5EntN& EB = (I::eA & 1) ? E0 : E1;
        if constexpr (I::opens) open_entry(ebase + 10 * wrap + I::eOpen, (I::eOpen & 1) ? E1 : E0);
        const bool head = wrap == 0 && first && extras;
        const char* __restrict__ base = (TWO && S < 5 && head) ? in2b : inb;
        constexpr unsigned rowA = 288u * I::mA0, rowB = 0u - 288u * I::sp;
        unsigned& lm = I::K < 2 ? vlane_main : vlane_z;
        unsigned& lf = I::K < 2 ? vlane16 : vlane8;
        const unsigned lr = I::K < 2 ? lane_rem : lane_rem_z;
        asm volatile("" : "+v"(lm));
        asm volatile("" : "+v"(lf));
        using V = std::remove_reference_t<decltype(o.b[0])>;
        if constexpr (!I::straddle) {
#pragma unroll
            for (int t = 0; t < NMAIN; ++t) {
                s5_gp pt = (s5_gp)base + EA.off[t];
                asm("" : "+s"(pt));
                o.b[t] = *(const S5_GLOBAL V*)(pt + lm + rowA);
            }
            if constexpr (HASREM) o.b[NMAIN] = *(const S5_GLOBAL V*)((s5_gp)base + (EA.rem + lr) + rowA);
        } else {
            const bool inB = (I::K < 2 ? l4 : (l4 >> 1)) >= I::sp;
#pragma unroll
            for (int t = 0; t < NMAIN; ++t) {
                const unsigned off = (inB ? EB.off[t] + rowB : EA.off[t] + rowA) + lm;
                o.b[t] = *(const S5_GLOBAL V*)((s5_gp)base + off);
            }
            if constexpr (HASREM) {
                const unsigned off = (inB ? EB.rem + rowB : EA.rem + rowA) + lr;
                o.b[NMAIN] = *(const S5_GLOBAL V*)((s5_gp)base + off);
            }
        }
        const S5_LDS char* fl = (const S5_LDS char*)fr + (unsigned)(tbase + 9 * wrap + I::T) * S5_TRIPLE_BYTES + (I::K == 0 ? 0 : I::K == 1 ? 2048 : 4096);
#pragma unroll
        for (int q = 0; q < 2; ++q) o.a[q] = *(const S5_LDS V*)(fl + lf + q * (I::K < 2 ? 1024 : 512));
    };
    // operands FIVE steps ahead in six register sets (two banks of X, Y, Z): a step of three tiles is 480 matrix cycles, and two steps of
    // them would not cover a gather that misses L2.  The set of step S + 5 is the one step S - 1 consumed; the 27-step period is odd, so
    // two periods are unrolled for the set names to repeat.
    S5PairN<NT> X0, Y0, X1, Y1;
    S5SingleN<NT> Z0, Z1;
    issue(S5C<0>{}, X0);
    issue(S5C<1>{}, Y0);
    issue(S5C<2>{}, Z0);
    issue(S5C<3>{}, X1);
    issue(S5C<4>{}, Y1);
    // (one exit per step -- so that a step is reachable only through the step before it and the compiler's vmcnt bookkeeping need not
    // assume that the steps between a request and its use were skipped: it emits vmcnt(2..3) where 14 loads may stay in flight -- made the
    // register allocator spill 9 KB per lane, as it did for the nine-tile wave)
#define S5_STEP(S, NEXT, CUR)                 \
    if (left > (S)) {                         \
        issue(S5C<(S) + 5>{}, NEXT);          \
        s5_mfma_n<NT>(acc, CUR);              \
        __builtin_amdgcn_sched_barrier(0);    \
    }
    for (;; left -= 54) {
        S5_STEP(0, Z1, X0) S5_STEP(1, X0, Y0) S5_STEP(2, Y0, Z0) S5_STEP(3, Z0, X1) S5_STEP(4, X1, Y1) S5_STEP(5, Y1, Z1)
        S5_STEP(6, Z1, X0) S5_STEP(7, X0, Y0) S5_STEP(8, Y0, Z0) S5_STEP(9, Z0, X1) S5_STEP(10, X1, Y1) S5_STEP(11, Y1, Z1)
        S5_STEP(12, Z1, X0) S5_STEP(13, X0, Y0) S5_STEP(14, Y0, Z0) S5_STEP(15, Z0, X1) S5_STEP(16, X1, Y1) S5_STEP(17, Y1, Z1)
        S5_STEP(18, Z1, X0) S5_STEP(19, X0, Y0) S5_STEP(20, Y0, Z0) S5_STEP(21, Z0, X1) S5_STEP(22, X1, Y1) S5_STEP(23, Y1, Z1)
        S5_STEP(24, Z1, X0) S5_STEP(25, X0, Y0) S5_STEP(26, Y0, Z0) S5_STEP(27, Z0, X1) S5_STEP(28, X1, Y1) S5_STEP(29, Y1, Z1)
        S5_STEP(30, Z1, X0) S5_STEP(31, X0, Y0) S5_STEP(32, Y0, Z0) S5_STEP(33, Z0, X1) S5_STEP(34, X1, Y1) S5_STEP(35, Y1, Z1)
        S5_STEP(36, Z1, X0) S5_STEP(37, X0, Y0) S5_STEP(38, Y0, Z0) S5_STEP(39, Z0, X1) S5_STEP(40, X1, Y1) S5_STEP(41, Y1, Z1)
        S5_STEP(42, Z1, X0) S5_STEP(43, X0, Y0) S5_STEP(44, Y0, Z0) S5_STEP(45, Z0, X1) S5_STEP(46, X1, Y1) S5_STEP(47, Y1, Z1)
        S5_STEP(48, Z1, X0) S5_STEP(49, X0, Y0) S5_STEP(50, Y0, Z0) S5_STEP(51, Z0, X1) S5_STEP(52, X1, Y1) S5_STEP(53, Y1, Z1)
        if (left <= 54) break;
        ebase += 20; tbase += 18; first = false;
    }
#undef S5_STEP
}

// One wave = (group of 8 atoms, output spin).  Input and output vectors in the CI layout.
// !LDSA: workgroup = 8 waves = 4 groups x 2 spins; waves w and w + 4 (same group, different spin) land on the same SIMD; the operator
//        fragments are global loads (L2-resident tables).
// LDSA:  workgroup = 8 groups of ONE output spin (the workgroup's parity); the fragment stream of that spin -- the same for every group
//        when the operator has one class of atoms -- is copied to LDS once and read from there: the texture addresser of the CU, 83 %
//        busy with the 11 loads per step, loses two of them (profiles/: TA_TA_BUSY).  Even XCDs then work on spin 0 and odd ones on
//        spin 1: an XCD's L2 holds only one spin half of the neighbour blocks of a collinear operator.
// Element-wise epilogue on the result block of every atom, t = H psi still in registers:
//   kind 0: out = t.
//   kind 1 (cheb_1st_mom, recursion.f90:2228-2236):        out = (t - b cur) / a                      (cur = psi0)
//   kind 2 (chebyshev_recur_ll, :2548-2587):               out = ((t - b cur) / a) * 2 - old          (cur = psi1, old = psi0)
// in the reference's operation order -- the Chebyshev step then never writes and re-reads H psi (two of the six block streams of a
// level); its Gram matrices are formed by k_mfma_cheb<., true> in one pass over cur and out.
struct S5Epilogue { int kind = 0; const double* cur = nullptr; const double* old = nullptr; double a = 1.0, b = 0.0; };

// OCT (global-load form only): a group is ONE atom with its own operator blocks (class tau < nmax) and its 8 tiles are 8 chains of the
// batch -- blockIdx.y counts octets of chains, the groups are the run [run_lo, run_hi) of the class-sorted list of all atoms, which
// every chain of the launch must be on (the host launches this form only then).  Chains share an atom's fragments the way 8 atoms of a
// type do; as groups of their own such atoms fill one tile of nine.
// NSP = 3 (LDS / persistent form only): split tasks -- a wave takes a third of a group's tiles (s5_run_stream_n); up to 16 waves per workgroup.
template <bool TWO, bool LDSA, bool OCT = false, int NSP = 1>
__global__ __launch_bounds__(NSP == 3 ? 768 : S5_WG_GROUPS * 128) void k_spmm5(SpmmDims D, const int* __restrict__ order_all, const int* __restrict__ cum,
                                               const int* __restrict__ nbr /*nbr5: (kk+1) x (nslots+2)*/,
                                               const int* __restrict__ izp, const double* __restrict__ frag, const int* __restrict__ meta, int ntr,
                                               const double* __restrict__ in_all, double* __restrict__ out_all,
                                               const double* __restrict__ in2_all = nullptr,
                                               const double* __restrict__ frag_head = nullptr /*[chain][tau][S5_HEAD_DOUBLES]: per-chain head of the stream*/,
                                               int ntau = 0, int lds_tau = 0 /*LDSA: the operator class whose stream is staged (all groups must be of it)*/,
                                               int* __restrict__ queue = nullptr /*LDSA: [chain][16] group counters, zero at launch: persistent workgroups, one group per pull*/,
                                               int spin_by_xcd = 1 /*LDSA: 1: even XCDs spin 0, odd spin 1 (collinear operators); 0: both spins on every XCD*/,
                                               S5Epilogue epi = S5Epilogue()) {
    extern __shared__ double s5_lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int sig = LDSA ? (int)((spin_by_xcd ? blockIdx.x : (blockIdx.x >> 3)) & 1) : wave / S5_WG_GROUPS, gslot = LDSA ? wave : wave % S5_WG_GROUPS;
#if RSREC_S5_PRIO
    // TIMING PROBE: static priority for the second-dispatched half of the workgroup's waves (the arbitration loser of a two-waves-per-SIMD kernel)
    if (wave >= (int)(blockDim.x >> 7)) __builtin_amdgcn_s_setprio(1);
#endif
    const int WGG = LDSA ? (int)(blockDim.x >> 6) : S5_WG_GROUPS;     // groups per workgroup (LDSA: one per wave; 8 waves, or 4 when the launch leaves half of the CU to other kernels)
    // several class runs in one persistent launch: this workgroup's run, and its place among the workgroups of that run
    int bxl = blockIdx.x, gdxl = gridDim.x, run_lo = D.run_lo, run_hi = D.run_hi, qrun = 0;
    if (LDSA && D.nruns > 0) {
        const int row = spin_by_xcd ? 8 : 16, rowi = (int)blockIdx.x / row;
        int r = 0;
#pragma unroll
        for (int q = 1; q < 4; ++q) if (q < D.nruns && rowi >= D.run_row0[q]) r = q;
        lds_tau = D.run_tau[r]; run_lo = D.run_glo[r]; run_hi = D.run_ghi[r];
        bxl = (int)blockIdx.x - row * D.run_row0[r];
        gdxl = row * (D.run_row0[r + 1] - D.run_row0[r]);
        qrun = r;
    }
    if constexpr (LDSA) {
        const s5_d2* __restrict__ src = reinterpret_cast<const s5_d2*>(frag + ((size_t)lds_tau * 2 + sig) * ntr * S5_TRIPLE);
        s5_d2* dst = reinterpret_cast<s5_d2*>(s5_lds);
        // all loads of a thread in flight at once (a rolled loop paid one memory latency per iteration: the copy then cost more than
        // the LDS reads save, because a workgroup lives for one round of groups only)
        const int n = ntr * (S5_TRIPLE / 2);
        constexpr int PER = (int)(160 * 1024 / 16 / (NSP == 3 ? 512 : S5_WG_GROUPS * 64));      // 40: the LDS limit over the threads of a 4-wave workgroup (split tasks: workgroups of 8 waves or more)
        const int nthr = (int)blockDim.x;
        s5_d2 v[PER];
#pragma unroll
        for (int i = 0; i < PER; ++i) { const int e = threadIdx.x + i * nthr; if (e < n) v[i] = src[e]; }
#pragma unroll
        for (int i = 0; i < PER; ++i) { const int e = threadIdx.x + i * nthr; if (e < n) dst[e] = v[i]; }
        __syncthreads();
    }
    const int l15 = lane & 15, l4 = lane >> 4;
    // CI layout (kernels_mfma.hpp): element (r, c) of a block = complex at doubles 36 r + 2 c; row r = 9 sigma + m.  Lane (l15, l4) of an
    // X / Y step reads element (m0 + l4, c = l15) as one 16-byte load (re, im = the step's two k-steps); of a Z step element
    // (m0 + (l4 >> 1), l15), part l4 & 1.  m0 (and the entry) are per-step constants of the stream (S5St).
    const unsigned lane_main = 8u * (36 * l4 + 2 * l15), lane_z = 8u * (36 * (l4 >> 1) + 2 * l15 + (l4 & 1));
    const unsigned lane_rem = 8u * (36 * l4 + 32 + 2 * (l15 & 1)), lane_rem_z = 8u * (36 * (l4 >> 1) + 32 + 2 * (l15 & 1) + (l4 & 1));
    const unsigned lane16 = 16u * lane, lane8 = 8u * lane;
    const int ncol = D.nslots + 2;
    // grid.y may be smaller than the number of chains ("chain_fold"): a workgroup then serves chains blockIdx.y, + gridDim.y, ...
#pragma unroll 1
    for (int chain = OCT ? GROUP * (int)blockIdx.y : (int)blockIdx.y; chain < D.nchains; chain += OCT ? GROUP * (int)gridDim.y : (int)gridDim.y) {
    const int count = cum[(chain / D.cpo) * D.nlev + D.level];
    int ngroups = count / GROUP;
    const int ob = D.obase[(chain / D.cpo) * D.nlev + D.level];
    const int* __restrict__ order = order_all + (size_t)(chain / D.cpo) * D.ostride + ob;
    const bool on_full_list = D.sat_base > 0 && ob == D.sat_base;          // the class-sorted list of all atoms
    if constexpr (OCT) ngroups = D.nmax;                                    // group g = atom g (no list), for the 8 chains from `chain` on
    else if (run_hi > 0) {                                                       // one class run of that list (its stream is what this workgroup holds in LDS)
        if (!on_full_list) continue;
        order += (size_t)run_lo * GROUP;
        ngroups = run_hi - run_lo;
    }
    const bool skipping = D.nskip > 0 && on_full_list;
    const size_t vo = (size_t)chain * D.vstride;
    const char* __restrict__ inb = reinterpret_cast<const char*>(in_all + vo);
    const char* __restrict__ in2b = TWO ? reinterpret_cast<const char*>(in2_all + vo) : nullptr;
    double* __restrict__ out = out_all + vo;
    const int zero_block = D.kk;

    // XCD x sweeps chunk x of the group list (workgroups are dealt round-robin over the 8 XCDs, each with its own L2);
    // LDSA, spin_by_xcd: XCD x sweeps quarter x >> 1 of the list for spin x & 1 -- an XCD's L2 then holds one spin half of the neighbour
    // blocks of a collinear operator; LDSA, !spin_by_xcd (spin-mixing operators read both halves): the workgroups of an XCD alternate
    // between the spins and share the XCD's eighth of the list
    int g, gend, gstep, glo = 0;
    bool dynamic = false;
    {
        const int bx = bxl;
        const int need = (ngroups + WGG - 1) / WGG;               // workgroups one round of the chain takes (per spin for LDSA)
        if constexpr (!LDSA) {
            const int nbx = max(1, min((int)gridDim.x, need));
            if (bx >= nbx) continue;                             // launch sized for the largest chain of the batch
            if (nbx < 8) { g = bx * WGG + gslot; gend = ngroups; gstep = nbx * WGG; }
            else {
                const int xcd = bx & 7, j = bx >> 3;
                const int per_xcd = (nbx >> 3) + ((xcd < (nbx & 7)) ? 1 : 0);
                const int chunk = (ngroups + 7) >> 3;
                const int lo = xcd * chunk;
                gend = min(ngroups, lo + chunk);
                g = lo + j * WGG + gslot;
                gstep = per_xcd * WGG;
            }
        } else {
            // whole rows of 8 (spin_by_xcd) or 16 (both spins per XCD) workgroups; a chain too short for one row per XCD chunk is dealt
            // over the first row as one list
            const int row = spin_by_xcd ? 8 : 16;
            const int nbx = max(row, min(gdxl / row * row, (2 * need + row - 1) / row * row));
            if (bx >= nbx) continue;
            const int xcd = bx & 7, j = bx >> 3;
            if (2 * need <= row) {                               // short chain: the first row, 4 (8) workgroups per spin
                const int per_spin = row / 2, idx = spin_by_xcd ? (xcd >> 1) : xcd;
                if (j >= row / 8) continue;
                g = idx * WGG + gslot; gend = ngroups; gstep = per_spin * WGG;
            } else {
                const int rows = nbx / row;                      // workgroups of this spin on this XCD
                const int nchunk = spin_by_xcd ? 4 : 8, ichunk = spin_by_xcd ? (xcd >> 1) : xcd;
                const int chunk = (ngroups + nchunk - 1) / nchunk;
                const int lo = ichunk * chunk;
                gend = min(ngroups, lo + chunk);
                g = lo + (spin_by_xcd ? j : (j >> 1)) * WGG + gslot;
                gstep = rows * WGG;
                glo = lo;
                dynamic = queue != nullptr;
            }
        }
    }
    // queue mode (LDSA, grid.x a multiple of 8, few long-lived workgroups): the waves of XCD x take the groups of its chunk one at a
    // time from the chain's counter x -- the copy of the operator stream into LDS is paid once per workgroup, the balancing stays
    // dynamic at the granularity of one group; every wave leaves when the counters of all its chains have run past their chunks
    int* __restrict__ ctr = dynamic ? queue + ((size_t)qrun * D.nchains + chain) * 16 + 2 * (blockIdx.x & 7) + (spin_by_xcd ? 0 : sig) : nullptr;
    // (issuing the pull for the NEXT group before the current group's work was tried: the returning atomic is the oldest entry of the
    // in-order vmcnt queue and every operand wait of the first steps then waits for it as well -- 19 % slower)
    // (round 3: the pull issued right after the group's last operands were consumed, its round trip under the result stores: +1.5 %)
    int part = 0;                                           // NSP = 3: the third of the group's tiles this task covers
    auto next_task = [&]() {                                // static assignment: the parts of a group one after the other, then the wave's next group
        if (NSP == 3 && !dynamic && part < 2) { ++part; return; }
        part = 0; g += gstep;
    };
    for (;; next_task()) {
        int kpart = 0;
        if (dynamic) {
            int gi = 0;
            if (lane == 0) gi = atomicAdd(ctr, 1);
            gi = __builtin_amdgcn_readfirstlane(gi);
#if RSREC_S5_KSPLIT_PROBE
            kpart = gi & 1; gi >>= 1;
#endif
            if constexpr (NSP == 3) { const unsigned gu = (unsigned)gi, gq = gu / 3u; part = (int)(gu - 3u * gq); gi = (int)gq; }
            g = glo + gi;
        }
        if (g >= gend) break;
        if (skipping) {                                     // groups of a class run that a launch with that class's stream in LDS serves
            bool skip = false;
#pragma unroll
            for (int r = 0; r < 4; ++r) skip |= r < D.nskip && g >= D.skip_lo[r] && g < D.skip_hi[r];
            if (skip) continue;
        }
        const int* __restrict__ grp = order + (size_t)g * GROUP;
        if constexpr (NSP == 3) {
            static_assert(NSP != 3 || (LDSA && !OCT), "split tasks: LDS / persistent form only");
            const int g0 = grp[0];
            const int first3 = g0 >= 0 ? g0 : zero_block;
            if (D.skip_pa && first3 < D.nmax) continue;
            const int tau3 = first3 < D.nmax ? first3 : D.nmax + izp[first3];
            const int* __restrict__ M3 = meta + (size_t)tau3 * Spmm5Operator::META;
            const char* __restrict__ fr3 = reinterpret_cast<const char*>(s5_lds);
            S5AccN<3> acc3;
#pragma unroll
            for (int t = 0; t < 3; ++t) { acc3.m[t] = (double4_t){0, 0, 0, 0}; acc3.r[t] = 0.0; }
            // result tiles -> memory (as `finish` below): tile i belongs to block av[i]; REM: tile 2 is the remainder tile (av[2] per lane)
            auto finish3 = [&](auto remc, const int (&av)[3]) {
                constexpr bool REM = decltype(remc)::value;
                size_t eo[3];
#pragma unroll
                for (int i = 0; i < 3; ++i) eo[i] = (size_t)BLD * av[i] + 324 * sig + ((REM && i == 2) ? 32 + 2 * (l15 & 1) : 2 * l15);
                s5_d2 ec[3][2], ez[3][2];
                double ecr[3], ezr[3];
                if (epi.kind) {
                    const double* cb = epi.cur + vo;
#pragma unroll
                    for (int i = 0; i < 3; ++i) {
#pragma unroll
                        for (int p = 0; p < 2; ++p) ec[i][p] = *reinterpret_cast<const s5_d2*>(cb + eo[i] + 36 * (4 * p + l4));
                        ecr[i] = cb[eo[i] + 288 + (l4 & 1)];
                    }
                    if (epi.kind == 2) {
                        const double* zb = epi.old + vo;
#pragma unroll
                        for (int i = 0; i < 3; ++i) {
#pragma unroll
                            for (int p = 0; p < 2; ++p) ez[i][p] = *reinterpret_cast<const s5_d2*>(zb + eo[i] + 36 * (4 * p + l4));
                            ezr[i] = zb[eo[i] + 288 + (l4 & 1)];
                        }
                    }
                }
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    const int a = av[i];
                    double* ob = out + eo[i];
#pragma unroll
                    for (int p = 0; p < 2; ++p) {
                        s5_d2 v; v[0] = acc3.m[i][2 * p]; v[1] = acc3.m[i][2 * p + 1];
                        if (epi.kind) {
                            v[0] = (v[0] - epi.b * ec[i][p][0]) / epi.a; v[1] = (v[1] - epi.b * ec[i][p][1]) / epi.a;
                            if (epi.kind == 2) { v[0] = v[0] * 2.0 - ez[i][p][0]; v[1] = v[1] * 2.0 - ez[i][p][1]; }
                        }
                        if (a != zero_block) *reinterpret_cast<s5_d2*>(ob + 36 * (4 * p + l4)) = v;
                    }
                    double r = acc3.r[i];
                    if (epi.kind) {
                        r = (r - epi.b * ecr[i]) / epi.a;
                        if (epi.kind == 2) r = r * 2.0 - ezr[i];
                    }
                    if (a != zero_block && l4 < 2) ob[288 + l4] = r;
                }
            };
            int ratom[GROUP];
            if (part < 2) {
                int a3[3];
#pragma unroll
                for (int j = 0; j < 3; ++j) { const int a = grp[3 * part + j]; a3[j] = a >= 0 ? a : zero_block; }
#pragma unroll
                for (int t = 0; t < GROUP; ++t) ratom[t] = zero_block;
                s5_run_stream_n<TWO, 3, false>(acc3, M3, fr3, inb, in2b, nbr, a3, ratom, 0, ncol, sig, l4, lane_main, lane_z, lane_rem, lane_rem_z, lane16, lane8);
                finish3(std::false_type{}, a3);
            } else {
#pragma unroll
                for (int t = 0; t < GROUP; ++t) { const int a = grp[t]; ratom[t] = a >= 0 ? a : zero_block; }
                const int a2[2] = {ratom[6], ratom[7]};
                int mra = grp[l15 >> 1];
                mra = mra >= 0 ? mra : zero_block;
                s5_run_stream_n<TWO, 2, true>(acc3, M3, fr3, inb, in2b, nbr, a2, ratom, l15 >> 1, ncol, sig, l4, lane_main, lane_z, lane_rem, lane_rem_z, lane16, lane8);
                const int av[3] = {a2[0], a2[1], mra};
                finish3(std::true_type{}, av);
            }
            continue;
        }
        int atom[GROUP];                                    // padding entries (-1) become the zero block: no predicates in the step loop
        int vatom[GROUP], tile_blk[GROUP];                  // OCT: block index of tile t's own atom seen from chain c0's vectors; its neighbour blocks' shift
        int my_rem_atom, rem_blk = 0;
        if constexpr (OCT) {
            const int a0 = g;                               // the group's one atom
#pragma unroll
            for (int t = 0; t < GROUP; ++t) {
                const bool ok = a0 >= 0 && chain + t < D.nchains;
                atom[t] = ok ? a0 : zero_block;
                tile_blk[t] = ok ? t * (D.kk + 1) : 0;
                vatom[t] = ok ? a0 + t * (D.kk + 1) : zero_block;
            }
            const int tr = l15 >> 1;
            const bool okr = a0 >= 0 && chain + tr < D.nchains;
            rem_blk = okr ? tr * (D.kk + 1) : 0;
            my_rem_atom = okr ? a0 + tr * (D.kk + 1) : zero_block;
        } else {
#pragma unroll
            for (int t = 0; t < GROUP; ++t) { const int a = grp[t]; atom[t] = a >= 0 ? a : zero_block; vatom[t] = atom[t]; tile_blk[t] = 0; }
            my_rem_atom = grp[l15 >> 1];
            my_rem_atom = my_rem_atom >= 0 ? my_rem_atom : zero_block;
        }
        const int first = atom[0];
        if (!OCT && D.skip_pa && first < D.nmax) continue;  // an atom with its own operator blocks: the chain-octet launch serves it
        const int tau = first < D.nmax ? first : D.nmax + izp[first];
        const int* __restrict__ M = meta + (size_t)tau * Spmm5Operator::META;
        const char* __restrict__ fr = LDSA ? reinterpret_cast<const char*>(s5_lds) : reinterpret_cast<const char*>(frag + ((size_t)tau * 2 + sig) * ntr * S5_TRIPLE);
        const char* __restrict__ fh = (TWO && frag_head) ? reinterpret_cast<const char*>(frag_head + ((size_t)chain * ntau + tau) * S5_HEAD_DOUBLES + (size_t)sig * S5_HEAD_TRIPLES * S5_TRIPLE) : nullptr;

        S5Acc acc;
#pragma unroll
        for (int t = 0; t < 9; ++t) { acc.m[t] = (double4_t){0, 0, 0, 0}; acc.r[t] = 0.0; }

        if constexpr (OCT) s5_run_stream<TWO, LDSA, true>(acc, M, fr, fh, inb, in2b, nbr, atom, l15 >> 1, ncol, sig, l4, lane_main, lane_z, lane_rem, lane_rem_z, lane16, lane8, tile_blk, rem_blk);
        else s5_run_stream<TWO, LDSA>(acc, M, fr, fh, inb, in2b, nbr, atom, l15 >> 1, ncol, sig, l4, lane_main, lane_z, lane_rem, lane_rem_z, lane16, lane8, tile_blk, rem_blk, kpart);
#if RSREC_S5_KSPLIT_PROBE
        if (kpart) continue;                                // (probe: one part stores)
#endif

        // 16x16x4 result register j, lane (l15, l4): real-form row l4 + 4 j of spin sig = (part j & 1, m = l4 + 4 (j >> 1)), column l15:
        // registers (2 p, 2 p + 1) are the real and imaginary part of element (m = 4 p + l4, c) -> one 16-byte store in the CI
        // layout; the 4x4x4 result: row 16 + l4 -> l4 = 0: re, 1: im of m = 8
        // Epilogue operands first, for several tiles at once (the three operand sets of the stream are dead: ~110 registers are free):
        // two memory round trips per group instead of one per tile.  Padding atoms read the zero block; only their stores are skipped.
        auto finish = [&](auto t0c, auto t1c) {
            constexpr int T0 = decltype(t0c)::value, T1 = decltype(t1c)::value, NT = T1 - T0;
            size_t eo[NT];
#pragma unroll
            for (int i = 0; i < NT; ++i) { const int t = T0 + i; eo[i] = (size_t)BLD * ((t < 8) ? vatom[t] : my_rem_atom) + 324 * sig + ((t < 8) ? 2 * l15 : 32 + 2 * (l15 & 1)); }
            s5_d2 ec[NT][2], ez[NT][2];
            double ecr[NT], ezr[NT];
            if (epi.kind) {
                const double* cb = epi.cur + vo;
#pragma unroll
                for (int i = 0; i < NT; ++i) {
#pragma unroll
                    for (int p = 0; p < 2; ++p) ec[i][p] = *reinterpret_cast<const s5_d2*>(cb + eo[i] + 36 * (4 * p + l4));
                    ecr[i] = cb[eo[i] + 288 + (l4 & 1)];
                }
                if (epi.kind == 2) {
                    const double* zb = epi.old + vo;
#pragma unroll
                    for (int i = 0; i < NT; ++i) {
#pragma unroll
                        for (int p = 0; p < 2; ++p) ez[i][p] = *reinterpret_cast<const s5_d2*>(zb + eo[i] + 36 * (4 * p + l4));
                        ezr[i] = zb[eo[i] + 288 + (l4 & 1)];
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < NT; ++i) {
                const int t = T0 + i;
                const int a = (t < 8) ? vatom[t] : my_rem_atom;
                double* ob = out + eo[i];
#pragma unroll
                for (int p = 0; p < 2; ++p) {
                    s5_d2 v; v[0] = acc.m[t][2 * p]; v[1] = acc.m[t][2 * p + 1];
                    if (epi.kind) {
                        v[0] = (v[0] - epi.b * ec[i][p][0]) / epi.a; v[1] = (v[1] - epi.b * ec[i][p][1]) / epi.a;
                        if (epi.kind == 2) { v[0] = v[0] * 2.0 - ez[i][p][0]; v[1] = v[1] * 2.0 - ez[i][p][1]; }
                    }
                    if (a != zero_block) *reinterpret_cast<s5_d2*>(ob + 36 * (4 * p + l4)) = v;
                }
                double r = acc.r[t];
                if (epi.kind) {
                    r = (r - epi.b * ecr[i]) / epi.a;
                    if (epi.kind == 2) r = r * 2.0 - ezr[i];
                }
                if (a != zero_block && l4 < 2) ob[288 + l4] = r;
            }
        };
        finish(S5C<0>{}, S5C<5>{});
        finish(S5C<5>{}, S5C<9>{});
    }
    }   // chains of this workgroup
}

}  // namespace rsrec
