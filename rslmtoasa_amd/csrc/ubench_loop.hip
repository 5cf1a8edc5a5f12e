// Inner-loop micro-benchmark for the SpMM (round 3): the operand traffic of one stream step -- nine 16-byte gathers per lane from an
// L2-resident vector (1 KB per wave instruction, rows 288 B apart like the CI layout) and the operator fragments from LDS -- under
//   (A) today's MFMA mix: per step of 4 orbitals two k-steps x nine (16x16x4 + 4x4x4)              [36 MFMA, 1440 matrix cycles]
//   (B) the real-basis mix: per step of 4 orbitals ONE k-step x 18 column tiles x three 4x4x4 row blocks  [54 MFMA,  864 matrix cycles]
// Two waves per SIMD, operands requested one step ahead.  Prints time per step: the ratio B / A is what a real-basis 4x4x4 SpMM
// could gain on spin-diagonal real blocks before any of its other costs.
// Build: hipcc --offload-arch=gfx950 -O3 ubench_loop.hip -o ubench_loop
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double double4_t __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

template <int MODE, int LOADS>
__global__ __launch_bounds__(512, 2) void k_loop(const double* __restrict__ vec, size_t nblocks, int steps, double* out) {
    extern __shared__ double lds[];
    for (int e = threadIdx.x; e < 8192; e += blockDim.x) lds[e] = 1e-3 * (e % 97);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l15 = lane & 15, l4 = lane >> 4;
    const size_t gw = (size_t)blockIdx.x * 8 + wave;
    double4_t c[9];
    double r[54];
#pragma unroll
    for (int i = 0; i < 9; ++i) c[i] = (double4_t){0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < 54; ++i) r[i] = 0.0;
    d2 b[2][9];
    auto load = [&](int s, d2 (&o)[9]) {
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const unsigned blk = ((unsigned)gw * 131u + (unsigned)s * 17u + (unsigned)t * 29u) & (unsigned)(nblocks - 1);        // scattered blocks, L2-resident footprint
            o[t] = *reinterpret_cast<const d2*>(vec + (size_t)blk * 648 + 36 * (l4 + 4 * (s % 2)) + 2 * l15);
        }
    };
    load(0, b[0]); load(1, b[1]);
    for (int s = 0; s < steps; s += 2) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            if (LOADS) load(s + h + 1, b[1 - h]);
            const d2* A = reinterpret_cast<const d2*>(lds + ((s + h) % 16) * 512) + lane;
            if (MODE == 0) {
                const d2 a0 = A[0], a1 = A[64];
#pragma unroll
                for (int e = 0; e < 2; ++e)
#pragma unroll
                    for (int t = 0; t < 9; ++t) {
                        c[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[e], b[h][t][e], c[t], 0, 0, 0);
                        r[t] = __builtin_amdgcn_mfma_f64_4x4x4f64(a1[e], b[h][t][e], r[t], 0, 0, 0);
                    }
            } else if (MODE == 2) {
                // rows 16, 17 of the real form on the vector ALU instead of a 4x4x4 tile: per k-step two v_fma_f64 per tile (partial sums per k lane)
                const d2 a0 = A[0], a1 = A[64], a2 = A[128];
#pragma unroll
                for (int e = 0; e < 2; ++e)
#pragma unroll
                    for (int t = 0; t < 9; ++t) {
                        c[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[e], b[h][t][e], c[t], 0, 0, 0);
                        r[2 * t] = __builtin_fma(a1[e], b[h][t][e], r[2 * t]);
                        r[2 * t + 1] = __builtin_fma(a2[e], b[h][t][e], r[2 * t + 1]);
                    }
            } else {
                const d2 a0 = A[0], a1 = A[64];
                const double ar[3] = {a0[0], a0[1], a1[0]};
#pragma unroll
                for (int t = 0; t < 9; ++t)
#pragma unroll
                    for (int e = 0; e < 2; ++e)
#pragma unroll
                        for (int rb = 0; rb < 3; ++rb)
                            r[(2 * t + e) * 3 + rb] = __builtin_amdgcn_mfma_f64_4x4x4f64(ar[rb], b[h][t][e], r[(2 * t + e) * 3 + rb], 0, 0, 0);
            }
        }
    }
    double acc = 0;
#pragma unroll
    for (int i = 0; i < 9; ++i) acc += c[i][0] + c[i][3];
#pragma unroll
    for (int i = 0; i < 54; ++i) acc += r[i];
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

template <int MODE, int LOADS>
void run(const char* name, const double* vec, size_t nblocks, int blocks, int steps) {
    double* out;
    CK(hipMalloc(&out, sizeof(double) * blocks * 512));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_loop<MODE, LOADS>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    k_loop<MODE, LOADS><<<blocks, 512, 65536>>>(vec, nblocks, 16, out);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    k_loop<MODE, LOADS><<<blocks, 512, 65536>>>(vec, nblocks, steps, out);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double cyc = ms * 1e-3 * 2.4e9 / steps / 2.0;            // two waves share a SIMD: SIMD cycles per step per wave
    printf("%-46s %8.3f ms for %d steps: %7.1f SIMD cycles per step and wave (matrix cycles nominal: %d)\n", name, ms, steps, cyc, MODE == 0 ? 1440 : MODE == 1 ? 864 : 1152);
    CK(hipFree(out));
}

int main(int argc, char** argv) {
    hipDeviceProp_t p;
    CK(hipGetDeviceProperties(&p, 0));
    const size_t nblocks = argc > 1 ? (size_t)atol(argv[1]) : 4096;      // power of two; 4096 = 21 MB (Infinity Cache), 512 = 2.6 MB (every XCD's L2)
    printf("footprint %.1f MB\n", nblocks * 648 * 8 / 1e6);
    double* vec;
    CK(hipMalloc(&vec, nblocks * 648 * 8));
    CK(hipMemset(vec, 0, nblocks * 648 * 8));
    const int blocks = p.multiProcessorCount;          // one 8-wave workgroup per CU: two waves per SIMD
    run<0, 1>("A: today's mix (2 k-steps x 9 x (16x16x4 + 4x4x4))", vec, nblocks, blocks, 4000);
    run<1, 1>("B: real-basis mix (18 tiles x 3 row blocks of 4x4x4)", vec, nblocks, blocks, 4000);
    run<2, 1>("C: 16x16x4 rows + rows 16, 17 on the vector ALU", vec, nblocks, blocks, 4000);
    run<0, 0>("A without the gathers", vec, nblocks, blocks, 4000);
    run<2, 0>("C without the gathers", vec, nblocks, blocks, 4000);
    run<1, 0>("B without the gathers", vec, nblocks, blocks, 4000);
    return 0;
}
