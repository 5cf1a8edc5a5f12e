// FP64 matrix-core kernels for gfx950: v_mfma_f64_16x16x4_f64 (full 16x16 tiles) + v_mfma_f64_4x4x4_4b_f64 (the 4-row
// remainder of the 36-row real form), measured in profiles/ubench_f64_r01.txt.
//
// Block sparse H|psi> (hop_b, reference recursion.f90:1560-1625) as real GEMMs.  A complex 18x18 block product
// out = H * x is the real 36x36 * 36xN product
//        [ out_re ]   [ Hr  -Hi ] [ x_re ]
//        [ out_im ] = [ Hi   Hr ] [ x_im ]            (row index kappa = 18*part + r)
// For one neighbour slot the SAME 36x36 operator multiplies the psi blocks of all atoms of a type, so 8 atoms of one
// type form one wave's GEMM:  M = 36 (= 16 + 16 + 4, no padding waste thanks to the 4x4x4 instruction),
// K = 36 (nine k-steps of 4), N = 8 atoms x 18 columns = 144 = 9 tiles of 16:
//   tiles 0..7 : columns 0..15 of atom 0..7 (atom-aligned), tile 8 : columns 16,17 of the eight atoms.
// Work vectors use LayoutRM (kernels_valu.hpp): a B fragment (4 k-rows x 16 columns) is four 128-byte row segments.
// Operator blocks are pre-swizzled on the host into A-fragment order (one coalesced 512-byte load per fragment).
#pragma once
#include <hip/hip_runtime.h>
#include <string>
#include <vector>
#include "kernels_valu.hpp"

namespace rsrec {

typedef double double4_t __attribute__((ext_vector_type(4)));

constexpr int GROUP = 8;            // atoms per wave (one type)
constexpr int MF_WAVES = 4;         // waves per workgroup
constexpr int FRAG_PER_SLOT = 9 * 3 * 64;   // doubles: [q][f][lane]

// Host-side builder + device storage of the fragment tables.
struct MfmaOperator {
    double* d_frag = nullptr;       // [2 (h | h*o)][ntau][nslots][9][3][64]
    size_t bytes = 0;
    int ntau = 0, nslots = 0, have_o = 0;

    void release() { if (d_frag) (void)hipFree(d_frag); d_frag = nullptr; bytes = 0; }

    // one complex block (column-major interleaved) -> 9*3*64 doubles in fragment order
    static void swizzle(const double* blk, double* out) {
        auto R = [&](int ko, int ki) -> double {   // real 36x36 form, kappa = 18*part + index
            const int po = ko / 18, ro = ko % 18, pi = ki / 18, ri = ki % 18;
            const double hr = blk[2 * (ro + 18 * ri)], hi = blk[2 * (ro + 18 * ri) + 1];
            if (po == pi) return hr;
            return po == 0 ? -hi : hi;
        };
        for (int q = 0; q < 9; ++q)
            for (int f = 0; f < 3; ++f)
                for (int l = 0; l < 64; ++l) {
                    const int k = 4 * q + (l >> 4);
                    const int m = (f < 2) ? 16 * f + (l & 15) : 32 + (l & 3);   // f = 2: 4x4x4 A operand, replicated over the 4 blocks
                    out[(q * 3 + f) * 64 + l] = R(m, k);
                }
    }

    // st/loc/... are the SAME host arrays the VALU path uploads (slot 0 already carries +lsham when !hoh)
    const char* build(int nslots_lat, int hstride, int ntype, int nmax, int hoh, const double* st, const double* loc, const double* eeo,
                      const double* hallo, const double* /*enim*/, const double* /*lsham*/) {
        ntau = nmax + ntype; nslots = nslots_lat; have_o = hoh ? 1 : 0;
        const size_t per_set = (size_t)ntau * nslots * FRAG_PER_SLOT;
        std::vector<double> host(per_set * (have_o ? 2 : 1), 0.0);
        for (int set = 0; set < (have_o ? 2 : 1); ++set)
            for (int tau = 0; tau < ntau; ++tau)
                for (int s = 0; s < nslots; ++s) {
                    const double* src;
                    if (tau < nmax) src = (set ? hallo : loc) + 2 * (size_t)BLK * (s + (size_t)hstride * tau);
                    else src = (set ? eeo : st) + 2 * (size_t)BLK * (s + (size_t)hstride * (tau - nmax));
                    swizzle(src, host.data() + set * per_set + ((size_t)tau * nslots + s) * FRAG_PER_SLOT);
                }
        const size_t need = host.size() * sizeof(double);
        if (need > bytes) {
            release();
            if (hipMalloc(reinterpret_cast<void**>(&d_frag), need) != hipSuccess) return "hipMalloc of MFMA operator fragments failed";
            bytes = need;
        }
        if (hipMemcpy(d_frag, host.data(), need, hipMemcpyHostToDevice) != hipSuccess) return "upload of MFMA operator fragments failed";
        return nullptr;
    }
    const double* set_ptr(int set) const { return d_frag + (size_t)set * ntau * nslots * FRAG_PER_SLOT; }
};

struct SpmmArgs {
    const double* frag;     // fragment table of the operator set in use
    const double* in;       // vector the neighbour sum runs over
    double* out;            // result vector (same layout)
    int level;
};

// out_i = sum_slots H_slot * in_{nbr(i,slot)} for every atom of the (padded, type-homogeneous) order prefix.
// One wave = one group of 8 atoms.  Group entries < 0 are padding.
__global__ __launch_bounds__(MF_WAVES * 64, 2) void k_mfma_spmm(DevProblem P, ChainView CV, SpmmArgs G) {
    const int chain = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int count = CV.count_of(chain, G.level);          // multiple of GROUP
    const int ngroups = count / GROUP;
    const int* order = CV.order_of(chain);
    const size_t vo = (size_t)chain * CV.vstride;
    const double* __restrict__ in = G.in + vo;
    double* out = G.out + vo;
    const int zero_block = P.kk;                              // index of the all-zero block (absent neighbours, padding)
    const int l15 = lane & 15, l4 = lane >> 4;

    for (int g = blockIdx.x * MF_WAVES + wave; g < ngroups; g += gridDim.x * MF_WAVES) {
        const int* grp = order + (size_t)g * GROUP;
        int atom[GROUP];
#pragma unroll
        for (int t = 0; t < GROUP; ++t) atom[t] = grp[t];
        const int first = atom[0];                             // groups are never empty: entry 0 is a real atom
        const int tau = first < P.nmax ? first : P.nmax + P.iz[first];
        const int my_rem_atom = grp[l15 >> 1];                 // remainder tile: lane -> (atom (l15>>1), column 16 + (l15&1))
        const double* fr = G.frag + (size_t)tau * P.nslots * FRAG_PER_SLOT + lane;

        double4_t acc0[9], acc1[9];
        double acc2[9];
#pragma unroll
        for (int t = 0; t < 9; ++t) { acc0[t] = (double4_t){0, 0, 0, 0}; acc1[t] = (double4_t){0, 0, 0, 0}; acc2[t] = 0.0; }

        for (int s = 0; s < P.nslots; ++s) {
            // source block of every tile for this slot
            size_t src[9];
#pragma unroll
            for (int t = 0; t < GROUP; ++t) {
                int n = atom[t] >= 0 ? P.nbr[(size_t)P.nslots * atom[t] + s] : -1;
                if (n < 0) n = zero_block;
                src[t] = (size_t)BLD * n + l15;                // column l15 of the atom-aligned tile
            }
            {
                int n = my_rem_atom >= 0 ? P.nbr[(size_t)P.nslots * my_rem_atom + s] : -1;
                if (n < 0) n = zero_block;
                src[8] = (size_t)BLD * n + 16 + (l15 & 1);
            }
            const double* fs = fr + (size_t)s * FRAG_PER_SLOT;
#pragma unroll
            for (int q = 0; q < 9; ++q) {
                const double a0 = fs[(q * 3 + 0) * 64], a1 = fs[(q * 3 + 1) * 64], a2 = fs[(q * 3 + 2) * 64];
                const int kap = 4 * q + l4;                    // this lane's k-row: kappa = 18*part + r
                const int koff = (kap < 18) ? 36 * kap : 36 * (kap - 18) + 18;
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    const double b = in[src[t] + koff];
                    acc0[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b, acc0[t], 0, 0, 0);
                    acc1[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b, acc1[t], 0, 0, 0);
                    acc2[t] = __builtin_amdgcn_mfma_f64_4x4x4f64(a2, b, acc2[t], 0, 0, 0);
                }
            }
        }
        // store: D layout row = l4 + 4*j (+16 for acc1), column = l15;  acc2: row 32 + l4
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int a = (t < 8) ? atom[t] : my_rem_atom;
            if (a < 0) continue;
            double* ob = out + (size_t)BLD * a + ((t < 8) ? l15 : 16 + (l15 & 1));
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int k0 = l4 + 4 * j;                     // 0..15 -> part 0, r = k0
                ob[36 * k0] = acc0[t][j];
                const int k1 = 16 + l4 + 4 * j;                // 16..31
                ob[(k1 < 18) ? 36 * k1 : 36 * (k1 - 18) + 18] = acc1[t][j];
            }
            ob[36 * (14 + l4) + 18] = acc2[t];                 // kappa = 32 + l4 -> part 1, r = 14 + l4
        }
    }
}

}  // namespace rsrec
