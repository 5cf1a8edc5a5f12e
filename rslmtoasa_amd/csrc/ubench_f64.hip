// Micro-benchmarks that decide the kernel design on gfx950 (MI355X):
//   (1) v_fma_f64 VALU rate, (2) v_mfma_f64_16x16x4_f64 rate, (3) v_mfma_f64_4x4x4_4b_f64 rate,
//   (4) VALU + MFMA f64 issued from different waves of the same SIMD (do they add?),
//   (5) lane maps of both f64 MFMA shapes, checked with exact integer data.
// Build: hipcc --offload-arch=gfx950 -O3 ubench_f64.hip -o ubench_f64
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>

typedef double double4_t __attribute__((ext_vector_type(4)));

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

template <int MODE>
__global__ __launch_bounds__(256) void k_rate(double* out, int iters, double seed) {
    // MODE 0: VALU fma only; 1: MFMA16 only; 2: MFMA4 only; 3: even waves VALU, odd waves MFMA16
    const int wave = threadIdx.x >> 6;
    double r = 0.0;
    bool do_valu = (MODE == 0) || (MODE == 3 && (wave & 1) == 0);
    bool do_m16 = (MODE == 1) || (MODE == 3 && (wave & 1) == 1);
    if (do_valu) {
        double a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, a4 = seed + 4, a5 = seed + 5, a6 = seed + 6, a7 = seed + 7;
        const double x = 1.0000001, y = 1e-9 * threadIdx.x;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                a0 = __builtin_fma(a0, x, y); a1 = __builtin_fma(a1, x, y); a2 = __builtin_fma(a2, x, y); a3 = __builtin_fma(a3, x, y);
                a4 = __builtin_fma(a4, x, y); a5 = __builtin_fma(a5, x, y); a6 = __builtin_fma(a6, x, y); a7 = __builtin_fma(a7, x, y);
            }
        }
        r = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    } else if (do_m16) {
        double4_t c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
        double a = seed + threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
                c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0);
                c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
            }
        }
        r = c0[0] + c1[1] + c2[2] + c3[3];
    } else if (MODE == 2) {
        double c0 = 0, c1 = 0, c2 = 0, c3 = 0;
        double a = seed + threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                c0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c1, 0, 0, 0);
                c2 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c2, 0, 0, 0);
                c3 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c3, 0, 0, 0);
            }
        }
        r = c0 + c1 + c2 + c3;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

__global__ void k_map16(const double* A, const double* B, double* D) {
    // A: 16x4 row-major, B: 4x16 row-major -> D 16x16 row-major using the guide's lane map
    int l = threadIdx.x;
    double a = A[(l & 15) * 4 + (l >> 4)];
    double b = B[(l >> 4) * 16 + (l & 15)];
    double4_t c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) D[((l >> 4) + 4 * r) * 16 + (l & 15)] = c[r];
}

__global__ void k_map4(const double* a_in, const double* b_in, double* d_out) {
    // raw dump: lane l gets a_in[l], b_in[l]; d_out[l] = result. Host decodes the map.
    int l = threadIdx.x;
    double c = 0;
    c = __builtin_amdgcn_mfma_f64_4x4x4f64(a_in[l], b_in[l], c, 0, 0, 0);
    d_out[l] = c;
}

template <int MODE>
double run_rate(const char* name, int blocks, int iters, double flop_per_wave_iter_valu, double flop_per_wave_iter_mfma) {
    double* out;
    CK(hipMalloc(&out, sizeof(double) * blocks * 256));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    k_rate<MODE><<<blocks, 256>>>(out, 10, 1.0);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    k_rate<MODE><<<blocks, 256>>>(out, iters, 1.0);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    double waves = (double)blocks * 4;
    double fl;
    if (MODE == 0) fl = waves * iters * flop_per_wave_iter_valu;
    else if (MODE == 1 || MODE == 2) fl = waves * iters * flop_per_wave_iter_mfma;
    else fl = waves / 2 * iters * (flop_per_wave_iter_valu + flop_per_wave_iter_mfma);
    printf("%-28s blocks=%5d iters=%d  %.3f ms  %.2f TFLOP/s\n", name, blocks, iters, ms, fl / ms * 1e-9);
    CK(hipFree(out));
    return ms;
}

int main() {
    hipDeviceProp_t p;
    CK(hipGetDeviceProperties(&p, 0));
    printf("device %s CUs=%d clock=%d kHz\n", p.name, p.multiProcessorCount, p.clockRate);
    const double valu = 64.0 * 8 * 8 * 2;          // 64 fma per unrolled iter per lane * 64 lanes * 2 flop
    const double m16 = 16.0 * 2048;                 // 16 MFMA 16x16x4 per iter
    const double m4 = 16.0 * 512;                   // 16 MFMA 4x4x4(4 blocks)
    for (int bpc : {1, 2, 4}) {
        int blocks = p.multiProcessorCount * bpc;
        run_rate<0>("valu_fma_f64", blocks, 4000, valu, 0);
        run_rate<1>("mfma_f64_16x16x4", blocks, 4000, 0, m16);
        run_rate<2>("mfma_f64_4x4x4_4b", blocks, 4000, 0, m4);
        run_rate<3>("valu(even)+mfma16(odd) waves", blocks, 4000, valu, m16);
    }
    // lane-map checks
    {
        std::vector<double> A(64), B(64), D(256), R(256, 0.0);
        for (int i = 0; i < 64; ++i) { A[i] = (i * 7 + 3) % 11 - 5; B[i] = (i * 5 + 1) % 13 - 6; }
        for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) for (int k = 0; k < 4; ++k) R[i * 16 + j] += A[i * 4 + k] * B[k * 16 + j];
        double *dA, *dB, *dD;
        CK(hipMalloc(&dA, 512)); CK(hipMalloc(&dB, 512)); CK(hipMalloc(&dD, 2048));
        CK(hipMemcpy(dA, A.data(), 512, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), 512, hipMemcpyHostToDevice));
        k_map16<<<1, 64>>>(dA, dB, dD);
        CK(hipMemcpy(D.data(), dD, 2048, hipMemcpyDeviceToHost));
        int bad = 0;
        for (int i = 0; i < 256; ++i) if (D[i] != R[i]) ++bad;
        printf("mfma_f64_16x16x4 lane map (A[l&15][l>>4], B[l>>4][l&15], D[(l>>4)+4r][l&15]): %s (%d mismatches)\n", bad ? "WRONG" : "OK", bad);
        // 4x4x4: probe with unit vectors to find which (lane_a, lane_b) pairs feed which output lane
        std::vector<double> a(64), b(64), d(64);
        printf("mfma_f64_4x4x4 map probe: for output lane L list (la,lb) contributing\n");
        std::vector<std::vector<std::pair<int,int>>> contrib(64);
        for (int la = 0; la < 64; ++la) {
            for (int i = 0; i < 64; ++i) { a[i] = (i == la) ? 1.0 : 0.0; b[i] = (double)(i + 1); }
            CK(hipMemcpy(dA, a.data(), 512, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, b.data(), 512, hipMemcpyHostToDevice));
            k_map4<<<1, 64>>>(dA, dB, dD);
            CK(hipMemcpy(d.data(), dD, 512, hipMemcpyDeviceToHost));
            for (int L = 0; L < 64; ++L) if (d[L] != 0.0) contrib[L].push_back({la, (int)d[L] - 1});
        }
        for (int L = 0; L < 64; ++L) {
            printf("  D lane %2d <-", L);
            for (auto& pr : contrib[L]) printf(" a%d*b%d", pr.first, pr.second);
            printf("\n");
        }
        CK(hipFree(dA)); CK(hipFree(dB)); CK(hipFree(dD));
    }
    return 0;
}
