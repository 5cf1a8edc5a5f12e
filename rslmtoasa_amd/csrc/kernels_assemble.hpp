// Raw operator blocks assembled on the device: what build_bulkham / build_locham (hamiltonian.f90:1553-1667) do on the
// host for every class atom (one per atom type, one per impurity atom) and neighbour slot m:
//   ee(:,:,m) = [[H0 + Hz, Hx - i Hy], [Hx + i Hy, H0 - Hz]]      from the four 9x9 parts chbar_nc leaves in hmag(:,:,m,1..4) = Hx, Hy, Hz, H0
//   eeo(:,:,m) = ee(:,:,m) * obarm(:,:,type of the atom behind slot m)   (hoh only; zero where the slot is empty)
// Tiny, launch-bound work (ncls x nslots blocks of 18x18): one workgroup per block, one thread per element, the product from LDS.
#pragma once
#include <hip/hip_runtime.h>

namespace rsrec {

// hmag: complex (9, 9, nslots, 4, ncls) column-major; nbr_type: (nslots, ncls), 1-based type, 0 = no atom; obarm: complex (18, 18, ntype)
// out / out_o: complex (18, 18, nslots, ncls)
__global__ __launch_bounds__(384) void k_assemble_blocks(const double2* __restrict__ hmag, const int* __restrict__ nbr_type, const double2* __restrict__ obarm,
                                                         int nslots, int hoh, double2* __restrict__ out, double2* __restrict__ out_o) {
    __shared__ double2 E[324];
    const int m = blockIdx.x, c = blockIdx.y, e = threadIdx.x;
    const size_t blk = (size_t)c * nslots + m;
    const int r = e % 18, col = e / 18;
    if (e < 324) {
        const int j = r % 9, i = col % 9, so = r / 9, si = col / 9;
        const double2* hm = hmag + ((size_t)c * 4 * nslots + m) * 81 + (j + 9 * i);      // part p at + p * nslots * 81
        const size_t ps = (size_t)nslots * 81;
        double2 v;
        if (so == si) {
            const double2 h0 = hm[3 * ps], hz = hm[2 * ps];
            v = so == 0 ? make_double2(h0.x + hz.x, h0.y + hz.y) : make_double2(h0.x - hz.x, h0.y - hz.y);
        } else {
            const double2 hx = hm[0], hy = hm[ps];
            // Hx -/+ i Hy with i Hy = (-hy.y, hy.x)
            v = so == 0 ? make_double2(hx.x + hy.y, hx.y - hy.x) : make_double2(hx.x - hy.y, hx.y + hy.x);
        }
        E[e] = v;
        out[blk * 324 + e] = v;
    }
    if (!hoh) return;
    __syncthreads();
    if (e < 324) {
        const int ty = nbr_type[(size_t)c * nslots + m];
        double2 s = make_double2(0.0, 0.0);
        if (ty > 0) {
            const double2* O = obarm + (size_t)(ty - 1) * 324 + 18 * col;
            for (int k = 0; k < 18; ++k) {
                const double2 a = E[r + 18 * k], b = O[k];
                s.x += a.x * b.x - a.y * b.y;
                s.y += a.x * b.y + a.y * b.x;
            }
        }
        out_o[blk * 324 + e] = s;
    }
}

// the on-site block as hop_b uses it when hoh is off: slot 0 of class c += lsham(:,:,type of c)  (locham = ee(:,:,1,ih) + lsham(:,:,ih),
// recursion.f90:1582,1608).  type == nullptr: class c is type c.  Grid: ncls workgroups of 324 threads.
__global__ __launch_bounds__(324) void k_fold_onsite(double2* __restrict__ blocks, int nslots, const double2* __restrict__ lsham, const int* __restrict__ type) {
    const int c = blockIdx.x, e = threadIdx.x;
    const double2 a = lsham[(size_t)(type ? type[c] : c) * 324 + e];
    double2& d = blocks[(size_t)c * nslots * 324 + e];
    d.x += a.x; d.y += a.y;
}

}  // namespace rsrec
