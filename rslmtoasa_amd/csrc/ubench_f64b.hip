// Follow-up micro-benchmark: can ONE wave per SIMD keep the FP64 MFMA pipe busy when its MFMAs are independent?
// NACC independent 16x16x4 accumulators per wave (our SpMM has 18 + 9 of the 4x4x4 kind per k-step).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double double4_t __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

template <int NACC, int N4>
__global__ __launch_bounds__(256, 1) void k(double* out, int iters, double seed) {
    double4_t c[NACC > 0 ? NACC : 1];
    double d[N4 > 0 ? N4 : 1];
#pragma unroll
    for (int i = 0; i < NACC; ++i) c[i] = (double4_t){0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < N4; ++i) d[i] = 0.0;
    double a = seed + threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4, a2 = a * 0.5;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) c[i] = __builtin_amdgcn_mfma_f64_16x16x4f64((i & 1) ? a : a2, b, c[i], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < N4; ++i) d[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, d[i], 0, 0, 0);
    }
    double r = 0;
#pragma unroll
    for (int i = 0; i < NACC; ++i) r += c[i][0] + c[i][3];
#pragma unroll
    for (int i = 0; i < N4; ++i) r += d[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int NACC, int N4>
void run(int blocks, int iters) {
    double* out;
    CK(hipMalloc(&out, sizeof(double) * blocks * 256));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    k<NACC, N4><<<blocks, 256>>>(out, 10, 1.0);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    k<NACC, N4><<<blocks, 256>>>(out, iters, 1.0);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    double fl = (double)blocks * 4 * iters * (NACC * 2048.0 + N4 * 512.0);
    printf("1 wave/SIMD  %2d x mfma16 + %2d x mfma4 independent per iteration: %.3f ms  %.2f TFLOP/s\n", NACC, N4, ms, fl / ms * 1e-9);
    CK(hipFree(out));
}

int main() {
    run<4, 0>(256, 8000);
    run<8, 0>(256, 4000);
    run<18, 0>(256, 2000);
    run<18, 9>(256, 2000);
    run<1, 16>(256, 8000);
    run<27, 0>(256, 2000);
    return 0;
}
