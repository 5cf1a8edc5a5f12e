// Attainable rates on MI355X for the two shapes the recursion level is made of (round 3; output committed as profiles/ubench_attain_r03.txt):
//   (1) the FP64 matrix pipe fed like k_spmm5's step -- nine independent (16x16x4 + 4x4x4) accumulator pairs per k-step, operands in
//       registers (no loads at all) -- at one, two and four waves per SIMD: the ceiling of the SpMM's instruction mix;
//   (2) HBM with the post-hop kernels' stream mixes over a footprint far beyond the caches: 2 reads (k_mfma_adot), 3 reads + 1 write
//       in place (k_mfma_orth3: t', u, u_prev read, u_next written over u_prev), and the 1 read + 1 write copy the guide quotes.
// Build: hipcc --offload-arch=gfx950 -O3 ubench_attain.hip -o ubench_attain
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double double4_t __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

template <int WPS>
__global__ __launch_bounds__(256, WPS > 2 ? 2 : WPS) void k_mix(double* out, int iters, double seed) {
    double4_t c[9];
    double d[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) { c[i] = (double4_t){0, 0, 0, 0}; d[i] = 0.0; }
    double a = seed + threadIdx.x * 1e-3, a2 = a * 0.5, b[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) b[i] = 1.0 + threadIdx.x * 1e-4 * (i + 1);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            c[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b[i], c[i], 0, 0, 0);
            d[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a2, b[i], d[i], 0, 0, 0);
        }
    }
    double r = 0;
#pragma unroll
    for (int i = 0; i < 9; ++i) r += c[i][0] + c[i][3] + d[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int WPS>
void run_mix(int cus, int iters) {
    const int nb = cus * WPS;                 // 4 waves per block: WPS blocks per CU = WPS waves per SIMD
    double* out;
    CK(hipMalloc(&out, sizeof(double) * nb * 256));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    k_mix<WPS><<<nb, 256>>>(out, 10, 1.0);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    k_mix<WPS><<<nb, 256>>>(out, iters, 1.0);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double fl = (double)nb * 4 * iters * 9 * (2048.0 + 512.0);
    printf("mfma mix 9 x (16x16x4 + 4x4x4), %d wave(s)/SIMD: %8.3f ms  %6.2f TFLOP/s  (%.3f of 78.6)\n", WPS, ms, fl / ms * 1e-9, fl / ms * 1e-9 / 78.6);
    CK(hipFree(out));
}

// MODE 0: sum of two streams (2 R); 1: c = a + b + c in place (3 R + 1 W); 2: copy (1 R + 1 W)
template <int MODE>
__global__ __launch_bounds__(256) void k_stream(const d2* __restrict__ a, const d2* __restrict__ b, d2* c, size_t n, double* sink) {
    d2 acc = {0, 0};
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        if (MODE == 0) { const d2 x = a[i], y = b[i]; acc += x * y; }
        else if (MODE == 1) { const d2 x = a[i], y = b[i], z = c[i]; c[i] = x + y * 0.5 + z * 0.25; }
        else c[i] = a[i];
    }
    if (MODE == 0 && acc[0] + acc[1] == 12345.678) sink[0] = acc[0];
}

template <int MODE>
void run_stream(const char* name, size_t n, double streams, int blocks) {
    d2 *a, *b, *c; double* sink;
    CK(hipMalloc(&a, n * 16)); CK(hipMalloc(&b, n * 16)); CK(hipMalloc(&c, n * 16)); CK(hipMalloc(&sink, 64));
    CK(hipMemset(a, 0, n * 16)); CK(hipMemset(b, 0, n * 16)); CK(hipMemset(c, 0, n * 16));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    k_stream<MODE><<<blocks, 256>>>(a, b, c, n, sink);
    CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0));
        k_stream<MODE><<<blocks, 256>>>(a, b, c, n, sink);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    printf("hbm %-34s %5.1f GB in %7.3f ms  %6.2f TB/s  (%.3f of 8.0)  [%d blocks]\n", name, streams * n * 16 * 1e-9, best, streams * n * 16 / best * 1e-9, streams * n * 16 / best * 1e-9 / 8.0, blocks);
    CK(hipFree(a)); CK(hipFree(b)); CK(hipFree(c)); CK(hipFree(sink));
}

int main() {
    hipDeviceProp_t p;
    CK(hipGetDeviceProperties(&p, 0));
    printf("device %s CUs=%d\n", p.name, p.multiProcessorCount);
    const int cus = p.multiProcessorCount;
    run_mix<1>(cus, 3000);
    run_mix<2>(cus, 3000);
    run_mix<4>(cus, 1500);
    const size_t n = (size_t)3 << 27;      // 3 x 2^27 x 16 B = 6.4 GB per stream
    for (int blocks : {cus * 8, cus * 16}) {
        run_stream<2>("copy (1 R + 1 W)", n, 2.0, blocks);
        run_stream<0>("two read streams (k_mfma_adot)", n, 2.0, blocks);
        run_stream<1>("3 R + 1 W in place (k_mfma_orth3)", n, 4.0, blocks);
    }
    return 0;
}
