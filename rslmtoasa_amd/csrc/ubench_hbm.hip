// HBM stream-mix micro-benchmark (round 3): how fast can the post-hop passes' access patterns run at all?
//   2R        two read streams                      (k_mfma_adot)
//   3R1W      three reads, one written in place     (k_mfma_orth3)
// swept over: 16-byte loads in flight per thread (U), workgroups per CU, default / non-temporal loads and stores, footprint per stream.
// Build: hipcc --offload-arch=gfx950 -O3 ubench_hbm.hip -o ubench_hbm
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double d2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

template <int NT> __device__ __forceinline__ d2 ld(const d2* p) { return NT ? __builtin_nontemporal_load(p) : *p; }
template <int NT> __device__ __forceinline__ void st(d2* p, d2 v) { if (NT) __builtin_nontemporal_store(v, p); else *p = v; }

// each workgroup walks contiguous chunks of U x 256 x 16 B per stream
template <int MODE, int U, int NT>
__global__ __launch_bounds__(256) void k_stream(const d2* __restrict__ a, const d2* __restrict__ b, d2* c, size_t n, double* sink) {
    d2 acc = {0, 0};
    const size_t chunk = (size_t)U * 256;
    for (size_t base = (size_t)blockIdx.x * chunk; base + chunk <= n; base += (size_t)gridDim.x * chunk) {
        d2 x[U], y[U], z[U];
#pragma unroll
        for (int u = 0; u < U; ++u) x[u] = ld<NT>(a + base + u * 256 + threadIdx.x);
#pragma unroll
        for (int u = 0; u < U; ++u) y[u] = ld<NT>(b + base + u * 256 + threadIdx.x);
        if (MODE == 2) {
            // the same two read streams with 8-byte loads (two per 16 bytes), as k_mfma_adot / k_mfma_cheb issue them
            const double* a8 = reinterpret_cast<const double*>(a + base);
            const double* b8 = reinterpret_cast<const double*>(b + base);
            double xs[2 * U], ys[2 * U];
#pragma unroll
            for (int u = 0; u < 2 * U; ++u) xs[u] = a8[u * 256 + threadIdx.x];
#pragma unroll
            for (int u = 0; u < 2 * U; ++u) ys[u] = b8[u * 256 + threadIdx.x];
#pragma unroll
            for (int u = 0; u < 2 * U; ++u) acc[0] += xs[u] * ys[u];
            continue;
        }
        if (MODE == 1) {
#pragma unroll
            for (int u = 0; u < U; ++u) z[u] = ld<NT>(c + base + u * 256 + threadIdx.x);
#pragma unroll
            for (int u = 0; u < U; ++u) st<NT>(c + base + u * 256 + threadIdx.x, x[u] + y[u] * 0.5 + z[u] * 0.25);
        } else {
#pragma unroll
            for (int u = 0; u < U; ++u) acc += x[u] * y[u];
        }
    }
    if (MODE != 1 && acc[0] + acc[1] == 12345.678) sink[0] = acc[0];
}

template <int MODE, int U, int NT>
void run(size_t n, int cus, int wg_per_cu, d2* a, d2* b, d2* c, double* sink) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int blocks = cus * wg_per_cu;
    k_stream<MODE, U, NT><<<blocks, 256>>>(a, b, c, n, sink);
    CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0));
        k_stream<MODE, U, NT><<<blocks, 256>>>(a, b, c, n, sink);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    const double streams = MODE == 1 ? 4.0 : 2.0;
    printf("%-5s U=%d %s wg/CU=%-2d footprint/stream %.2f GB: %7.3f ms  %5.2f TB/s\n", MODE == 0 ? "2R" : MODE == 1 ? "3R1W" : "2R-8B", U, NT ? "nt " : "def", wg_per_cu, n * 16e-9, best,
           streams * n * 16 / best * 1e-9);
}

template <int MODE, int NT>
void sweep(size_t n, int cus, d2* a, d2* b, d2* c, double* sink) {
    for (int w : {4, 8, 16}) {
        run<MODE, 1, NT>(n, cus, w, a, b, c, sink);
        run<MODE, 2, NT>(n, cus, w, a, b, c, sink);
        run<MODE, 4, NT>(n, cus, w, a, b, c, sink);
        run<MODE, 8, NT>(n, cus, w, a, b, c, sink);
    }
}

int main(int argc, char** argv) {
    hipDeviceProp_t p;
    CK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount;
    const size_t nmax = (size_t)3 << 27;                       // 6.4 GB per stream
    d2 *a, *b, *c; double* sink;
    CK(hipMalloc(&a, nmax * 16)); CK(hipMalloc(&b, nmax * 16)); CK(hipMalloc(&c, nmax * 16)); CK(hipMalloc(&sink, 64));
    CK(hipMemset(a, 0, nmax * 16)); CK(hipMemset(b, 0, nmax * 16)); CK(hipMemset(c, 0, nmax * 16));
    for (size_t n : {nmax, nmax / 8}) {
        sweep<0, 0>(n, cus, a, b, c, sink);
        sweep<2, 0>(n, cus, a, b, c, sink);
        sweep<0, 1>(n, cus, a, b, c, sink);
        sweep<1, 0>(n, cus, a, b, c, sink);
        sweep<1, 1>(n, cus, a, b, c, sink);
    }
    return 0;
}
