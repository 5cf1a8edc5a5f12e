// Block-sparse H|psi> on v_mfma_f64_4x4x4_4b_f64 only ("spmm4"), second generation of the SpMM kernel.
//
// Why 4x4x4 everywhere: measured on MI355X (profiles/ubench_f64_r01.txt) the four-block 4x4x4 FP64 MFMA sustains 65-75
// TFLOP/s against 54-66 for the 16x16x4 shape, its B/D lane maps coincide with the 16x16 tile's (B[k = lane>>4][n = lane&15],
// D[row = lane>>4][col = lane&15]), and 36 real rows are exactly nine 4-row blocks -- no remainder shape, and structural
// zeros can be skipped at 4-row x 4-column granularity:
//   rows/columns of the real 36x36 form are ordered SPIN-MAJOR, ks = 18*sigma + 9*part + m  (orbital row r = 9*sigma + m),
//   so for collinear magnets without spin-orbit hopping (every nsp <= 2 case of the reference: ee = [[H0+Hz, 0],[0, H0-Hz]],
//   hamiltonian.f90:1553-1617) the up-down sub-blocks are exactly zero and 32 of the 81 (row block, k-step) products vanish.
//   Skipping them is exact: they would add +0.0.  Blocks with any non-zero entry there use the full pattern.
//
// Cooperative split: the four waves of a workgroup work on the SAME group of 8 atoms, each on its own share of the
// neighbour slots (host-balanced by cost), and their partial sums meet in LDS.  That cuts the atoms in flight per XCD from
// 1024 to 256, which is what lets the 15-fold neighbour re-reads of psi blocks hit the XCD's 4 MiB L2 instead of the fabric
// (measured before: 52 GB fetched per launch for 5.7 GB algorithmic, profiles/r01_mfma_pmc_summary.txt).
#pragma once
#include <cstring>
#include <hip/hip_runtime.h>
#include <algorithm>
#include <vector>
#include "kernels_valu.hpp"
#include "kernels_mfma.hpp"

namespace rsrec {

#ifndef S4_PROBE
#define S4_PROBE 0   // timing probes only (tools/probe_spmm_bound.sh): 1 = no operator-fragment loads, 2 = no psi loads in the k-loop
#endif
constexpr int S4_FRAG_PER_SLOT = 9 * 9 * 64;   // doubles: [q][rb][lane]
constexpr int S4_MAXSLOTS = 32;

// spin-major real index ks (0..35) -> element offset inside a LayoutRM block (column 0)
__host__ __device__ inline int s4_row_offset(int ks) {
    const int sigma = ks / 18, rem = ks % 18, part = rem / 9, m = rem % 9;
    return 36 * (9 * sigma + m) + 18 * part;
}
// is (row block rb, k-step q) structurally non-zero for pattern PAT (0 = full, 1 = spin-diagonal)?
__host__ __device__ constexpr bool s4_nz(int pat, int rb, int q) {
    return pat == 0 || rb == 4 || q == 4 || ((rb < 4) == (q < 4));
}

struct Spmm4Operator {
    double* d_frag = nullptr;    // [set][tau][slot][9][9][64]
    int* d_meta = nullptr;       // [set][tau] records of S4_META ints: per wave share (count, slots...), then per-slot pattern
    size_t frag_bytes = 0, meta_bytes = 0;
    int ntau = 0, nslots = 0, have_o = 0;
    static constexpr int META = 5 * 2 * (1 + S4_MAXSLOTS);   // [share][pattern][count, slots...]; share 0 = all slots (one wave per group), 1..4 = the four cooperating waves

    void release() {
        if (d_frag) (void)hipFree(d_frag);
        if (d_meta) (void)hipFree(d_meta);
        d_frag = nullptr; d_meta = nullptr; frag_bytes = meta_bytes = 0;
    }

    static double real_form(const double* blk, int kso, int ksi) {
        const int so = kso / 18, po = (kso % 18) / 9, mo = kso % 9, si = ksi / 18, pi = (ksi % 18) / 9, mi = ksi % 9;
        const int ro = 9 * so + mo, ri = 9 * si + mi;
        const double hr = blk[2 * (ro + 18 * ri)], hi = blk[2 * (ro + 18 * ri) + 1];
        if (po == pi) return hr;
        return po == 0 ? -hi : hi;
    }
    // 1: the spin-flip quadrants of the block are zero (a spin-diagonal block), 0: not.  Branch-free over the nine contiguous complex numbers
    // of a quadrant column (an impurity region of 1000 atoms hands over 32 000 blocks per call): the bit patterns are OR-ed, the sign
    // bit dropped at the end (-0.0 is zero, NaN is not -- as `!= 0.0` has it)
    static int pattern_of(const double* blk) {
        unsigned long long acc = 0;
        for (int c = 0; c < 18; ++c) {
            unsigned long long w[18];
            memcpy(w, blk + 2 * (18 * c + (c < 9 ? 9 : 0)), sizeof w);
            for (int i = 0; i < 18; ++i) acc |= w[i];
        }
        return (acc & 0x7fffffffffffffffULL) ? 0 : 1;
    }
    static void swizzle(const double* blk, double* out) {
        for (int q = 0; q < 9; ++q)
            for (int rb = 0; rb < 9; ++rb)
                for (int l = 0; l < 64; ++l)      // A operand of the 4x4x4 MFMA: lane (i + 4g + 16k) = A[i][k], same for the 4 blocks g
                    out[(q * 9 + rb) * 64 + l] = real_form(blk, 4 * rb + (l & 3), 4 * q + (l >> 4));
    }

    const char* build(int nslots_lat, int hstride, int ntype, int nmax, int hoh, const double* st, const double* loc, const double* eeo, const double* hallo,
                      int /*unused*/ = 0) {
        if (nslots_lat > S4_MAXSLOTS) return "too many neighbour slots for the spmm4 kernel";
        ntau = nmax + ntype; nslots = nslots_lat; have_o = hoh ? 1 : 0;
        const int nset = have_o ? 2 : 1;
        const size_t per_set = (size_t)ntau * nslots * S4_FRAG_PER_SLOT;
        std::vector<double> host(per_set * nset, 0.0);
        std::vector<int> meta((size_t)nset * ntau * META, 0);
        for (int set = 0; set < nset; ++set)
            for (int tau = 0; tau < ntau; ++tau) {
                int* M = meta.data() + ((size_t)set * ntau + tau) * META;
                int cost[S4_MAXSLOTS], pats[S4_MAXSLOTS];
                for (int s = 0; s < nslots; ++s) {
                    const double* src;
                    if (tau < nmax) src = (set ? hallo : loc) + 2 * (size_t)BLK * (s + (size_t)hstride * tau);
                    else src = (set ? eeo : st) + 2 * (size_t)BLK * (s + (size_t)hstride * (tau - nmax));
                    swizzle(src, host.data() + set * per_set + ((size_t)tau * nslots + s) * S4_FRAG_PER_SLOT);
                    pats[s] = pattern_of(src);
                    cost[s] = pats[s] ? 49 : 81;
                }
                // share 0: every slot; shares 1..4: slots balanced over four cooperating waves, heaviest first onto the lightest wave
                std::vector<int> ord(nslots);
                for (int s = 0; s < nslots; ++s) ord[s] = s;
                std::stable_sort(ord.begin(), ord.end(), [&](int a, int b) { return cost[a] > cost[b]; });
                int load[4] = {0, 0, 0, 0};
                for (int s : ord) {
                    int w = 0;
                    for (int x = 1; x < 4; ++x) if (load[x] < load[w]) w = x;
                    int* W = M + ((1 + w) * 2 + pats[s]) * (1 + S4_MAXSLOTS);
                    W[1 + W[0]] = s; W[0]++;
                    load[w] += cost[s];
                    int* W0 = M + pats[s] * (1 + S4_MAXSLOTS);
                    W0[1 + W0[0]] = s; W0[0]++;
                }
                for (int w = 0; w < 10; ++w) { int* W = M + w * (1 + S4_MAXSLOTS); std::sort(W + 1, W + 1 + W[0]); }
            }
        const size_t need = host.size() * sizeof(double), mneed = meta.size() * sizeof(int);
        if (need > frag_bytes) {
            if (d_frag) (void)hipFree(d_frag);
            d_frag = nullptr; frag_bytes = 0;
            if (hipMalloc(reinterpret_cast<void**>(&d_frag), need) != hipSuccess) return "hipMalloc of spmm4 operator fragments failed";
            frag_bytes = need;
        }
        if (mneed > meta_bytes) {
            if (d_meta) (void)hipFree(d_meta);
            d_meta = nullptr; meta_bytes = 0;
            if (hipMalloc(reinterpret_cast<void**>(&d_meta), mneed) != hipSuccess) return "hipMalloc of spmm4 schedule failed";
            meta_bytes = mneed;
        }
        if (hipMemcpy(d_frag, host.data(), need, hipMemcpyHostToDevice) != hipSuccess) return "upload of spmm4 fragments failed";
        if (hipMemcpy(d_meta, meta.data(), mneed, hipMemcpyHostToDevice) != hipSuccess) return "upload of spmm4 schedule failed";
        return nullptr;
    }
    const double* frag_set(int set) const { return d_frag + (size_t)set * ntau * nslots * S4_FRAG_PER_SLOT; }
    const int* meta_set(int set) const { return d_meta + (size_t)set * ntau * META; }
};

// One k-step: 9 (or fewer) row blocks x 9 column tiles of 4x4x4 MFMAs.  PAT/Q are compile-time so skipped products vanish
// from the instruction stream.
template <int PAT, int Q>
__device__ __forceinline__ void s4_mfma_step(double (&acc)[9][9], const double (&a)[9], const double (&b)[9]) {
#pragma unroll
    for (int rb = 0; rb < 9; ++rb) {
        if (!s4_nz(PAT, rb, Q)) continue;
#pragma unroll
        for (int t = 0; t < 9; ++t) acc[rb][t] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[rb], b[t], acc[rb][t], 0, 0, 0);
    }
}

// number of structurally non-zero row blocks of k-step q
__host__ __device__ constexpr int s4_nrb(int pat, int q) {
    int n = 0;
    for (int rb = 0; rb < 9; ++rb) n += s4_nz(pat, rb, q) ? 1 : 0;
    return n;
}
// Issue order of one k-step: every operand load (with the one VALU add that forms its 32-bit offset) is followed by PER MFMAs.
// At one wave per SIMD nothing else covers a load's issue slot: left to itself hipcc puts the ~18 loads of a k-step in
// one burst, during which the matrix pipe drains and idles (measured: 46 % of the kernel, tools/probe_spmm_bound.sh).
template <int NA, int NM>
__device__ __forceinline__ void s4_interleave() {
    constexpr int PER = NM / (9 + NA);
#pragma unroll
    for (int i = 0; i < 9; ++i) {                            // psi operands: one offset add + one load each
        __builtin_amdgcn_sched_group_barrier(0x2, 1, 0);     // VALU
        __builtin_amdgcn_sched_group_barrier(0x20, 1, 0);    // VMEM read
        __builtin_amdgcn_sched_group_barrier(0x8, PER, 0);   // MFMA
    }
    __builtin_amdgcn_sched_group_barrier(0x2, 2, 0);         // operator fragments: one 64-bit base for the k-step, immediate offsets
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x20, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x8, PER, 0);
    }
    __builtin_amdgcn_sched_group_barrier(0x8, NM - PER * (9 + NA), 0);
}

// All slots of one pattern for one wave: operand ring (3 buffers, loads two k-steps ahead) + MFMAs.
// (A variant that keeps the ring alive across passes and groups was measured: hipcc then spills ~23 doubles per lane and
//  the kernel gets 6-13 % slower, so every pass primes its own ring.)
template <int PAT>
__device__ __forceinline__ void s4_run_slots(double (&acc)[9][9], const int* __restrict__ share, const double* __restrict__ fr, const double* __restrict__ in,
                                             const int* __restrict__ nbr, const int (&atom)[GROUP], int nslots, int zero_block, int l15,
                                             const unsigned (&koff)[9] /*bytes*/, unsigned lane8) {
    const int nmine = share[0];
    if (nmine <= 0) return;
    const char* __restrict__ inb = reinterpret_cast<const char*>(in);
    // Neighbour indices are wave-uniform (scalar loads).  They are fetched one whole slot ahead and only turned into
    // lane addresses right before the k-step that first needs them, so the index latency hides behind ~440 MFMAs
    // (waiting on them at the top of the slot cost a full memory round trip per slot at one wave per SIMD).
    auto load_idx = [&](int s, int (&n)[GROUP]) {
#pragma unroll
        for (int t = 0; t < GROUP; ++t) n[t] = nbr[(size_t)nslots * max(atom[t], 0) + s];
    };
    const int rem_t = l15 >> 1;            // atom of this lane's remainder-tile column
    auto make_src = [&](const int (&n)[GROUP], unsigned (&src)[9]) {
        int mr = zero_block;
#pragma unroll
        for (int t = 0; t < GROUP; ++t) {
            const int m = (atom[t] >= 0 && n[t] >= 0) ? n[t] : zero_block;
            src[t] = ((unsigned)BLD * m + l15) * 8u;          // byte offsets: 32-bit, added to the wave-uniform base by the load itself
            mr = (rem_t == t) ? m : mr;
        }
        src[8] = ((unsigned)BLD * mr + 16 + (l15 & 1)) * 8u;
    };
    unsigned src[9], srcn[9];
    int nraw[GROUP];
    double bq[3][9], aq[3][9];
    int s_cur = share[1];
    int s_nxt = share[1 + ((1 < nmine) ? 1 : 0)];
    load_idx(s_cur, nraw);
    make_src(nraw, src);
#pragma unroll
    for (int p = 0; p < 2; ++p) {
#pragma unroll
        for (int t = 0; t < 9; ++t) bq[p][t] = *reinterpret_cast<const double*>(inb + (src[t] + koff[p]));
#pragma unroll
        for (int rb = 0; rb < 9; ++rb)
            if (s4_nz(PAT, rb, p))
                aq[p][rb] = *reinterpret_cast<const double*>(reinterpret_cast<const char*>(fr + (size_t)s_cur * S4_FRAG_PER_SLOT + (p * 9 + rb) * 64) + lane8);
    }
    for (int j = 0; j < nmine; ++j) {
        // the last slot prefetches the first again (discarded)
        const int s_nxt2 = share[1 + ((j + 2 < nmine) ? j + 2 : 0)];
        load_idx(s_nxt, nraw);
        const double* __restrict__ fs = fr + (size_t)s_cur * S4_FRAG_PER_SLOT;
        const double* __restrict__ fsn = fr + (size_t)s_nxt * S4_FRAG_PER_SLOT;
#define S4_KSTEP(Q)                                                                                                    \
    {                                                                                                                  \
        constexpr int cur = (Q) % 3, nxt = ((Q) + 2) % 3, qn = ((Q) + 2) % 9;                                          \
        /* fragment base of k-step qn (wave-uniform), centred on row block 4: every load = SGPR base + lane offset + immediate */ \
        const char* __restrict__ fa = reinterpret_cast<const char*>((((Q) + 2 < 9) ? fs : fsn) + (qn * 9 + 4) * 64);   \
        const unsigned* sp = ((Q) + 2 < 9) ? src : srcn;                                                               \
        if (!(S4_PROBE & 2)) { _Pragma("unroll") for (int t = 0; t < 9; ++t)                                           \
            bq[nxt][t] = *reinterpret_cast<const double*>(inb + (sp[t] + koff[qn])); }                                 \
        if (!(S4_PROBE & 1)) { _Pragma("unroll") for (int rb = 0; rb < 9; ++rb)                                        \
            if (s4_nz(PAT, rb, qn)) aq[nxt][rb] = reinterpret_cast<const double*>(fa + lane8)[(rb - 4) * 64]; }        \
        s4_mfma_step<PAT, Q>(acc, aq[cur], bq[cur]);                                                                   \
        s4_interleave<s4_nrb(PAT, qn), 9 * s4_nrb(PAT, Q)>();                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                                             \
    }
        S4_KSTEP(0) S4_KSTEP(1) S4_KSTEP(2) S4_KSTEP(3) S4_KSTEP(4) S4_KSTEP(5) S4_KSTEP(6)
        make_src(nraw, srcn);
        S4_KSTEP(7) S4_KSTEP(8)
#undef S4_KSTEP
#pragma unroll
        for (int t = 0; t < 9; ++t) src[t] = srcn[t];
        s_cur = s_nxt;
        s_nxt = s_nxt2;
    }
}

// NSPLIT = 1: one wave per group of 8 atoms (four independent groups per workgroup).
// NSPLIT = 4: the four waves share one group, each taking the slots of its schedule; partial sums meet in LDS.
template <int NSPLIT>
__global__ __launch_bounds__(MF_WAVES * 64, 1) void k_spmm4(SpmmDims D, const int* __restrict__ order_all, const int* __restrict__ cum,
                                                            const int* __restrict__ nbr, const int* __restrict__ izp, const double* __restrict__ frag,
                                                            const int* __restrict__ meta, const double* __restrict__ in_all, double* __restrict__ out_all) {
    extern __shared__ __attribute__((aligned(16))) unsigned char s4_dyn[];
    double* lds = reinterpret_cast<double*>(s4_dyn);
    const int chain = blockIdx.y;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int count = cum[(chain / D.cpo) * D.nlev + D.level];
    const int ngroups = count / GROUP;
    const int* __restrict__ order = order_all + (size_t)(chain / D.cpo) * D.ostride + D.obase[(chain / D.cpo) * D.nlev + D.level];
    const size_t vo = (size_t)chain * D.vstride;
    const double* __restrict__ in = in_all + vo;
    double* __restrict__ out = out_all + vo;
    const int zero_block = D.kk;
    const int l15 = lane & 15, l4 = lane >> 4;
    unsigned koff[9];                  // this lane's k-row (4q + l4) of every k-step, as a byte offset
#pragma unroll
    for (int q = 0; q < 9; ++q) koff[q] = 8u * (unsigned)s4_row_offset(4 * q + l4);
    const unsigned lane8 = 8u * (unsigned)lane;

    // group walk: with NSPLIT = 4 the WORKGROUP is the unit (same XCD-chunked sliding window as GroupWalk)
    int g, gend, gstep;
    {
        const int nbx = gridDim.x, bx = blockIdx.x;
        const int units = (NSPLIT == 4) ? 1 : MF_WAVES;      // groups a workgroup advances per step
        const int mine = (NSPLIT == 4) ? 0 : wave;
        if (nbx < 8) { g = bx * units + mine; gend = ngroups; gstep = nbx * units; }
        else {
            const int xcd = bx & 7, j = bx >> 3;
            const int per_xcd = (nbx >> 3) + ((xcd < (nbx & 7)) ? 1 : 0);
            const int chunk = (ngroups + 7) >> 3;
            const int lo = xcd * chunk;
            gend = min(ngroups, lo + chunk);
            g = lo + j * units + mine;
            gstep = per_xcd * units;
        }
    }

    for (; g < gend; g += gstep) {
        const int* __restrict__ grp = order + (size_t)g * GROUP;
        int atom[GROUP];
#pragma unroll
        for (int t = 0; t < GROUP; ++t) atom[t] = grp[t];
        const int first = atom[0];
        const int tau = first < D.nmax ? first : D.nmax + izp[first];
        const int my_rem_atom = grp[l15 >> 1];
        const int* __restrict__ M = meta + (size_t)tau * Spmm4Operator::META + ((NSPLIT == 4) ? 1 + wave : 0) * 2 * (1 + S4_MAXSLOTS);
        const double* __restrict__ fr = frag + (size_t)tau * D.nslots * S4_FRAG_PER_SLOT;   // wave-uniform

        double acc[9][9];
#pragma unroll
        for (int rb = 0; rb < 9; ++rb)
#pragma unroll
            for (int t = 0; t < 9; ++t) acc[rb][t] = 0.0;

        // one pass per structural pattern (each pass is a single straight-line body: mixing both in one loop made hipcc spill)
        s4_run_slots<0>(acc, M, fr, in, nbr, atom, D.nslots, zero_block, l15, koff, lane8);
        s4_run_slots<1>(acc, M + (1 + S4_MAXSLOTS), fr, in, nbr, atom, D.nslots, zero_block, l15, koff, lane8);

        // output rows of row block rb: ks = 4 rb + l4, column l15 (D layout of the 4x4x4 MFMA with blocks over N)
        int ro[9];
#pragma unroll
        for (int rb = 0; rb < 9; ++rb) ro[rb] = s4_row_offset(4 * rb + l4);
        if (NSPLIT == 4) {
            // wave w owns tiles {w, w+4} (wave 0 also tile 8); the other waves' partial sums for them come through LDS:
            // slab[src wave][k-th tile that wave does NOT own][rb][lane]  (7 tiles per wave at most: 4*7*9*64*8 B = 126 KiB)
            auto owner = [](int t) { return t == 8 ? 0 : (t & 3); };
            auto slot_of = [](int w, int t) { return t - (t > w ? 1 : 0) - (t > w + 4 ? 1 : 0); };   // index of tile t among the tiles wave w does not own
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                if (owner(t) == wave) continue;
                const int k = slot_of(wave, t);
#pragma unroll
                for (int rb = 0; rb < 9; ++rb) lds[((wave * 7 + k) * 9 + rb) * 64 + lane] = acc[rb][t];
            }
            __syncthreads();
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                if (owner(t) != wave) continue;
                for (int w2 = 0; w2 < 4; ++w2) {          // fixed order: run-to-run reproducible
                    if (w2 == wave) continue;
                    const int k = slot_of(w2, t);
#pragma unroll
                    for (int rb = 0; rb < 9; ++rb) acc[rb][t] += lds[((w2 * 7 + k) * 9 + rb) * 64 + lane];
                }
                const int a = (t < 8) ? atom[t] : my_rem_atom;
                if (a >= 0) {
                    double* ob = out + (size_t)BLD * a + ((t < 8) ? l15 : 16 + (l15 & 1));
#pragma unroll
                    for (int rb = 0; rb < 9; ++rb) ob[ro[rb]] = acc[rb][t];
                }
            }
            __syncthreads();
        } else {
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int a = (t < 8) ? atom[t] : my_rem_atom;
                if (a < 0) continue;
                double* ob = out + (size_t)BLD * a + ((t < 8) ? l15 : 16 + (l15 & 1));
#pragma unroll
                for (int rb = 0; rb < 9; ++rb) ob[ro[rb]] = acc[rb][t];
            }
        }
    }
}

constexpr size_t S4_LDS_BYTES = (size_t)4 * 7 * 9 * 64 * sizeof(double);   // 129 024 B

}  // namespace rsrec
