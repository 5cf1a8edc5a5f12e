// librsrec: C ABI (include/rsrec.h) + host orchestration of the MI355X recursion engine.
//
// Reference path replaced: source/recursion.f90 of rslmtoasa/rslmtoasa -- recur_b/crecal_b/hop_b/hop_b_hoh
// (:1807/:1873/:1560/:1411), chebyshev_recur & helpers (:3057, :2145-2763), recur/crecal/hop (:3485/:3423/:3310),
// zsqr (:1980).  Everything numerical runs in hand-written HIP kernels for gfx950; there is no CPU fallback.
//
// Engine design:
//  * all chains of a call are independent (one per recursion site); they are advanced TOGETHER in batches so that
//    the small active regions of the first steps still fill the 256 CUs;
//  * the region growth of the reference (izero/idum/irlist) is purely topological: a breadth-first search from the
//    seed on the host gives the atom order; "active after L applications of H" = a prefix of that order;
//  * work vectors (psi, pmn, ...) never leave HBM; only the 18x18 coefficients come back to the host;
//  * reductions are two-stage and fixed-order (no float atomics) so results are run-to-run reproducible.
#include <hip/hip_runtime.h>
#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <tuple>
#include <vector>

#include "../../include/rsrec.h"
#include "kernels_valu.hpp"
#include "kernels_mfma.hpp"
#include "kernels_spmm4.hpp"
#include "kernels_spmm5.hpp"
#include "kernels_uscheme.hpp"
#include "kernels_green.hpp"
#include "kernels_ldos.hpp"
#include "kernels_kubo.hpp"
#include "kernels_assemble.hpp"
#include <dlfcn.h>

using namespace rsrec;

namespace {

struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
    hipError_t reserve(size_t n) {
        if (n <= bytes) return hipSuccess;
        if (p) { (void)hipFree(p); p = nullptr; bytes = 0; }
        hipError_t e = hipMalloc(&p, n);
        if (e == hipSuccess) bytes = n;
        return e;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; bytes = 0; }
    template <class T> T* as() const { return reinterpret_cast<T*>(p); }
};

}  // namespace

struct rsrec_handle {
    int device = 0;
    hipStream_t stream = nullptr;
    hipStream_t copy_stream = nullptr;     // device -> host copies of the Green stage, overlapped with its kernels
    hipEvent_t ev_green[2] = {nullptr, nullptr};
    hipStream_t side_stream = nullptr;     // B_{n+1} reduction + eigen-solve of level n, concurrent with H u_{n+1} of level n + 1 (u-scheme)
    hipEvent_t ev_orth = nullptr, ev_bred = nullptr;
    hipStream_t oct_stream = nullptr;      // the chain-octet launch of a level's H|psi> beside its main launch (disjoint outputs)
    hipEvent_t ev_oct_in = nullptr, ev_oct_out = nullptr;
    size_t p2_slot = 0;                    // doubles per slot of d_partial2 (slot 1: the side stream's presum)
    std::string err;
    // lattice (host copies for the region search + device tables)
    bool have_lattice = false, have_ham = false;
    std::vector<int32_t> lat_nn, lat_iz;   // the caller's tables as last uploaded: an SCF loop passes the same lattice every iteration
    int lat_nncols = 0;
    std::vector<double> lat_cr;
    int kk = 0, nslots = 0, nmax = 0, ntype = 0;
    std::vector<int> nbr;        // [kk][nslots] 0-based, -1 absent, slot 0 = self
    std::vector<int> iz0;        // 0-based types
    std::vector<int> radj_ptr, radj;  // reverse adjacency: atoms whose neighbour list contains n
    DevBuf d_nbr, d_iz, d_nbr5;   // nbr5: (kk+1) x (nslots+2), absent neighbours and the extra row point at the zero block (k_spmm5)
    // hamiltonian
    int hslots = 0, hoh = 0, nsp = 2;
    DevBuf d_hst, d_hloc, d_host, d_holoc, d_enim, d_lsham;
    Spmm4Operator s4_op;
    int s4_built_split = 0;
    Spmm5Operator s5_op;
    DevBuf d_s5queue;            // group counters of the persistent k_spmm5 form
    int s5_built = 0;
    size_t s5_lds_limit = (size_t)-1;   // LDS a k_spmm5 workgroup may ask for on THIS handle's device ((size_t)-1: not asked yet; hipFuncSetAttribute is per device)
    int n_cu = 0;                       // compute units of the device (size of the persistent launches)
    bool s4_attr = false;               // k_spmm4's and k_terminator's LDS opt-ins, per handle for the same reason
    size_t term_attr_lds = 0;
    std::vector<double> host_ee, host_lsham, host_eeo, host_enim, host_hall, host_hallo;   // operator arrays as last set (Kubo operator tables; local-axis runs)
    std::vector<double> host_st, host_loc;   // ee / hall with l.s folded into the on-site block when !hoh (what d_hst / d_hloc hold)
    // raw blocks assembled on the device (rsrec_assemble_blocks): [part: 0 per-type, 1 per-atom][0: blocks, 1: blocks x obar]; asm_host = what the
    // caller got back -- rsrec_set_hamiltonian recognises those arrays bitwise and then takes the device copies instead of uploading
    DevBuf d_asm[2][2], d_asm_in;
    std::vector<double> asm_host[2][2];
    int asm_nslots[2] = {0, 0}, asm_ncls[2] = {0, 0}, asm_hoh[2] = {0, 0};
    double n_kubo_chain_launches = 0;   // rsrec_kubo_moments: whole-lattice products of the last call (launches x vectors in flight)
    int n_octet_launch = 0;      // launches of the last call that formed the groups of per-atom-block atoms over 8 chains (k_spmm5<., false, true>)
    int n_asm_reused = 0;        // block arrays the last rsrec_set_hamiltonian took from those device copies (0..4)
    long n_asm_calls = 0, n_ldos_calls = 0, n_recursion_calls = 0;   // life-time counters of the handle (RSREC_REPORT)
    Spmm5Operator s5_la; int s5_la_ok = 0;   // operator tables of local-axis runs: H without the on-site l.s term, which comes per chain
    DevBuf d_la_extra;
    Spmm5Operator kubo_op[2], kubo_hbulk;   // v_a / v_b tables of the last rsrec_kubo_moments call
    Spmm5Operator orb_plain;                // h as ham_vec_matmul applies it when hoh is set (rsrec_orbital_moments, rsrec_apply_operator vel = 2)
    // work
    DevBuf d_green_in, d_green_out;   // rsrec_block_green
    DevBuf d_kubo[5];                 // rsrec_kubo_moments: work vectors, left / right matrices, slice partials, moments -- kept between calls (tens of GB:
                                      // their hipMalloc / hipFree cost 0.1-1.3 s per call on some boxes of the pool); given back when the recursion plans a batch
    DevBuf d_bsqrt, d_term, d_gim, d_ldos;   // LDOS stage on resident coefficients: sqrt(B^2), terminators, Im g0_jj, output images
    void* pin = nullptr;              // pinned host staging buffer: every per-call transfer goes through it (see xfer_*)
    size_t pin_bytes = 0;
    DevBuf d_frags, d_vec[6], d_order, d_cum, d_partial, d_partial2, d_coefA, d_coefB, d_bmats, d_status, d_seed, d_seedcoef, d_mu, d_scal, d_zsqr;
    // options
    long opt_batch = 0, opt_kernels = 0, opt_nblk = 0, opt_spmm5 = 2, opt_chain_fold = 1, opt_s5_cap = 0, opt_side = 1, opt_s5_lds = 1, opt_s5_queue = 1, opt_cheb_fused = 1;
    long opt_s5_waves = 8;
    long opt_sat_pct = 100;      // a chain whose region holds at least this share (per cent) of the lattice runs on the list of ALL atoms (blocks outside the region are zero); rounds 1-3: 80 -- with a position-sorted list per level the superset no longer buys locality and costs its extra atoms (46^3: -1 %)
    long opt_orth_oop = 1;       // 1 = k_mfma_orth3 writes u_{n+1} into a third u vector instead of over u_{n-1} (faster on the HBM; one more work vector)
    long opt_s5_split = 0;       // persistent k_spmm5: 3 = a wave takes a third of a group's tiles (k_spmm5<., true, false, 3>; s5_waves = 8 / 12 / 16 waves per CU then)
    long opt_s5_run_min = 0;     // operators with several classes: smallest class run (in groups) that gets an LDS launch of its own (0: by launch size)
    long opt_s5_spin_xcd = 0;    // persistent k_spmm5 on collinear operators: 1 = even XCDs serve output spin 0, odd XCDs spin 1; 0 = both spins on every XCD
    long opt_s5_octet = 8;       // atoms with their own operator blocks (nmax) from which their groups are formed over 8 CHAINS instead of one atom + 7 padding tiles (0: never; round 3: 64; B2FeCo, nmax = 15: 2.38 -> 2.34 ms per launch)
    long opt_s5_host_emit = 0;   // 1: swizzle k_spmm5's operator streams on the host (round-2 path) instead of assembling them on the device
    long opt_kubo_lchunk = 0;    // rsrec_kubo_moments: left vectors held at a time (0: as many as fit)
    long opt_kubo_vbatch = 0;    // rsrec_kubo_moments: random vectors advanced together as the chains of one launch (0: up to 8, as many as fit beside a whole left matrix)
    int n_kubo_left_chunks = 0;
    long opt_orth3 = 1;          // k_mfma_orth3: 1 one 512-register wave per SIMD (tables in registers), 2 two waves per SIMD (tables in LDS)
    long opt_graph = 1;          // level loop of small batches as one HIP graph: 0 never, 1 calls of up to 8 chains, 2 every single-batch call
    // The captured level loop of the last small-batch block-Lanczos call (every SCF iteration repeats it with the same lattice, seeds,
    // depth and buffers; the operator's VALUES are read through device pointers and may change).  key = everything the nodes hold by value.
    hipGraphExec_t graph_exec = nullptr;
    std::vector<uintptr_t> graph_key;
    bool capturing = false;      // between hipStreamBeginCapture and hipStreamEndCapture: no timing events, no allocation
    // timing of last call
    double t_total_ms = 0, t_hop_ms = 0, t_rest_ms = 0, t_host_ms = 0;
    double n_hop_launch = 0, n_atom_steps = 0, n_block_mult = 0, n_hop_mfma_flop = 0;   // mfma_flop: matrix flops EXECUTED by the timed k_spmm5 launches
    double n_req_flop = 0;    // flops of H|psi> the operator's block structure requires (spin-diagonal blocks: half a zgemm), see required_hop_flops
    int hop_fuses_a = 1;      // 1: the timed H|psi> kernel also forms pmn and the A_n partial (VALU path); 0: pure SpMM (MFMA path)
    std::vector<hipEvent_t> ev_pool;
    size_t ev_used = 0;
    // regions are purely topological: they are reused while the lattice and the seeds stay the same (every SCF iteration
    // of the reference calls recur_b with the same lattice%nn and lattice%irec)
    struct RegionEntry {
        std::vector<int> seeds;
        int nlev, napply, flags, epoch, ostride;
        double atom_steps, block_mults;
        DevBuf order, cum;
        std::vector<int> level_max;     // per level: largest active-atom count over the chains of this entry
        std::vector<double> level_groups;   // [level][tau]: groups of 8 atoms (padding included) of operator class tau, summed over the chains
        std::vector<double> mult_hist;      // [pass 0/1][tau][nslots + 1]: block multiplications of the call by (pass, operator class, slot), summed over the chains
        // the list of ALL atoms (used once a region covers the lattice) is sorted by operator class: one run of groups per class
        struct ClassRun { int tau, lo, hi; };
        std::vector<ClassRun> sat_runs;
        std::vector<int> level_sat;         // per level: chains of this entry that use that list
        int sat_base = 0;                   // its offset inside an order row
    };
    std::vector<RegionEntry*> region_cache;
    int lattice_epoch = 0;
    const int* cur_order = nullptr;
    const int* cur_cum = nullptr;     // [nrows][nlev] counts, [nrows][nlev] list offsets (the lists of H|psi>), then the same two tables for the streaming passes
    const std::vector<int>* cur_level_max = nullptr;
    const std::vector<double>* cur_level_groups = nullptr;
    const std::vector<double>* cur_mult_hist = nullptr;
    const RegionEntry* cur_entry = nullptr;
    int cur_nrows = 0;
    std::vector<unsigned> spatial_key;   // per atom: position along a space-filling curve (locality hint for the saturated order)
    // coefficients left on the device by the last recursion call: 0 = none, 1 = block Lanczos (d_coefA = a_b, d_coefB = b2_b or its root),
    // 2 = Chebyshev (d_mu = mu_n of all chains)
    int res_kind = 0, res_n = 0, res_lld = 0, res_sqrt = 0;
    // library-level communicator (RCCL, bound with dlopen at rsrec_comm_init): the one exchange of the path without MPI or torch
    void* comm = nullptr;
    int comm_rank = 0, comm_nranks = 1;
    DevBuf d_comm;               // staging buffer of rsrec_allreduce_sum on host arrays
};

namespace {


int fail(rsrec_t* h, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (h) h->err = buf;
    return code;
}

#define HIPCK(h, call)                                                                                         \
    do {                                                                                                       \
        hipError_t e_ = (call);                                                                                \
        if (e_ != hipSuccess) return fail(h, RSREC_ERR_DEVICE, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

hipEvent_t next_event(rsrec_t* h) {
    static const bool off = getenv("RSREC_NO_EVENTS") != nullptr;      // diagnostics only
    if (off || h->capturing) return nullptr;
    if (h->ev_used == h->ev_pool.size()) {
        hipEvent_t e;
        if (hipEventCreate(&e) != hipSuccess) return nullptr;
        h->ev_pool.push_back(e);
    }
    hipEvent_t e = h->ev_pool[h->ev_used++];
    (void)hipEventRecord(e, h->stream);
    return e;
}

// Host <-> device transfers of per-call data go through one pinned staging buffer.  hipMemcpyAsync straight from/to the
// caller's pageable arrays (freshly allocated numpy / Fortran ALLOCATE memory) was measured at 60-90 ms PER CALL for a 259 KB
// coefficient download on the MI355X boxes (rocprofv3 --hip-trace: the time is inside hipMemcpyAsync, the GPU idles) --
// four times the whole single-site recursion.  Staged copies are synchronous by construction (stream order + one wait).
constexpr size_t PIN_BYTES = (size_t)8 << 20;
int pin_ready(rsrec_t* h) {
    if (h->pin) return RSREC_OK;
    if (hipHostMalloc(&h->pin, PIN_BYTES, hipHostMallocDefault) != hipSuccess) { h->pin = nullptr; return fail(h, RSREC_ERR_DEVICE, "hipHostMalloc of the staging buffer failed"); }
    h->pin_bytes = PIN_BYTES;
    return RSREC_OK;
}
// device -> caller memory; waits for everything queued on the stream before it
constexpr size_t PIN_MAX_XFER = (size_t)8 << 20;     // larger transfers are bandwidth-, not latency-bound: direct copy (no extra host memcpy)
int xfer_d2h(rsrec_t* h, void* dst, const void* src_dev, size_t bytes) {
    if (bytes > PIN_MAX_XFER) {
        HIPCK(h, hipMemcpyAsync(dst, src_dev, bytes, hipMemcpyDeviceToHost, h->stream));
        HIPCK(h, hipStreamSynchronize(h->stream));
        return RSREC_OK;
    }
    int rc = pin_ready(h);
    if (rc) return rc;
    for (size_t off = 0; off < bytes; off += h->pin_bytes) {
        const size_t n = std::min(h->pin_bytes, bytes - off);
        HIPCK(h, hipMemcpyAsync(h->pin, static_cast<const char*>(src_dev) + off, n, hipMemcpyDeviceToHost, h->stream));
        HIPCK(h, hipStreamSynchronize(h->stream));
        memcpy(static_cast<char*>(dst) + off, h->pin, n);
    }
    return RSREC_OK;
}
// caller memory -> device; returns when the data is on the device (the staging buffer is free again)
int xfer_h2d(rsrec_t* h, void* dst_dev, const void* src, size_t bytes) {
    if (bytes > PIN_MAX_XFER) {
        HIPCK(h, hipMemcpyAsync(dst_dev, src, bytes, hipMemcpyHostToDevice, h->stream));
        HIPCK(h, hipStreamSynchronize(h->stream));
        return RSREC_OK;
    }
    int rc = pin_ready(h);
    if (rc) return rc;
    for (size_t off = 0; off < bytes; off += h->pin_bytes) {
        const size_t n = std::min(h->pin_bytes, bytes - off);
        memcpy(h->pin, static_cast<const char*>(src) + off, n);
        HIPCK(h, hipMemcpyAsync(static_cast<char*>(dst_dev) + off, h->pin, n, hipMemcpyHostToDevice, h->stream));
        HIPCK(h, hipStreamSynchronize(h->stream));
    }
    return RSREC_OK;
}
#define XFER(call) do { int rc__ = (call); if (rc__) return rc__; } while (0)

struct Region {
    std::vector<int> order;   // atoms sorted by (distance, index)
    std::vector<int> cum;     // cum[L] = #atoms with distance <= L, L = 0..nlev-1
};

// Breadth-first growth of the active region, restating izero/idum of hop_b (recursion.f90:1604-1636):
// atom i joins the region when one of ITS neighbour slots holds an atom already in it.
void grow_region(const rsrec_t* h, const int* seeds, int nseed, int nlev, Region& R) {
    const int kk = h->kk;
    std::vector<int> dist(kk, -1);
    R.order.clear();
    R.order.reserve(kk);
    R.cum.assign(nlev, 0);
    std::vector<int> frontier, next;
    for (int s = 0; s < nseed; ++s)
        if (dist[seeds[s]] < 0) { dist[seeds[s]] = 0; frontier.push_back(seeds[s]); }
    std::sort(frontier.begin(), frontier.end());
    int level = 0;
    while (!frontier.empty() && level < nlev) {
        R.order.insert(R.order.end(), frontier.begin(), frontier.end());
        R.cum[level] = (int)R.order.size();
        next.clear();
        for (int n : frontier)
            for (int q = h->radj_ptr[n]; q < h->radj_ptr[n + 1]; ++q) {
                const int i = h->radj[q];
                if (dist[i] < 0) { dist[i] = level + 1; next.push_back(i); }
            }
        std::sort(next.begin(), next.end());
        frontier.swap(next);
        ++level;
    }
    for (int L = std::max(level, 1); L < nlev; ++L) R.cum[L] = R.cum[L - 1];
    R.order.resize(kk, 0);   // tail is never read (cum bounds every loop)
}

}  // namespace

namespace {

inline unsigned spread3(unsigned v) {           // 10 bits -> every third bit
    v &= 1023u;
    v = (v | (v << 16)) & 0x030000FFu;
    v = (v | (v << 8)) & 0x0300F00Fu;
    v = (v | (v << 4)) & 0x030C30C3u;
    v = (v | (v << 2)) & 0x09249249u;
    return v;
}
inline unsigned morton3(unsigned x, unsigned y, unsigned z) { return spread3(x) | (spread3(y) << 1) | (spread3(z) << 2); }

// breadth-first distances over the (symmetrised) neighbour graph from one atom
void bfs_dist(const rsrec_t* h, int start, std::vector<int>& dist) {
    const int kk = h->kk, ns = h->nslots;
    dist.assign(kk, -1);
    std::vector<int> cur{start}, nxt;
    dist[start] = 0;
    int d = 0;
    while (!cur.empty()) {
        nxt.clear();
        for (int n : cur) {
            for (int j = 1; j < ns; ++j) { const int i = h->nbr[(size_t)n * ns + j]; if (i >= 0 && dist[i] < 0) { dist[i] = d + 1; nxt.push_back(i); } }
            for (int q = h->radj_ptr[n]; q < h->radj_ptr[n + 1]; ++q) { const int i = h->radj[q]; if (dist[i] < 0) { dist[i] = d + 1; nxt.push_back(i); } }
        }
        cur.swap(nxt);
        ++d;
    }
}

// Locality hint when the caller gives no coordinates: graph distances to three far-apart landmark atoms act as
// pseudo-coordinates (adequate for ordering: atoms with similar distance triples are close in the lattice).
void spatial_key_from_graph(rsrec_t* h) {
    const int kk = h->kk;
    std::vector<int> d0, d1, d2;
    bfs_dist(h, 0, d0);
    int l1 = 0;
    for (int i = 0; i < kk; ++i) if (d0[i] >= d0[l1]) l1 = i;
    bfs_dist(h, l1, d1);
    int l2 = 0, best = -1;
    for (int i = 0; i < kk; ++i) { const int m = std::min(d0[i], d1[i]); if (m >= best) { best = m; l2 = i; } }
    bfs_dist(h, l2, d2);
    h->spatial_key.resize(kk);
    for (int i = 0; i < kk; ++i) h->spatial_key[i] = morton3((unsigned)std::max(d0[i], 0), (unsigned)std::max(d1[i], 0), (unsigned)std::max(d2[i], 0));
}

}  // namespace

// ------------------------------------------------------------------------------------------------------------------
extern "C" int rsrec_version(void) { return 100; }

extern "C" int rsrec_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

// RSREC_REPORT set in the environment: one line of the handle's counters when the process exits -- the evidence a host that cannot
// ask (the reference's unmodified main program behind the shadow modules, fortran/build_dropin.sh) leaves in its log that its
// operator was assembled on the device and taken from there, and that the LDOS stage ran there.  Host-side counters only: no HIP call
// is made from the exit handler.
static rsrec_handle* g_report_handle = nullptr;
static void report_at_exit() {
    const rsrec_handle* h = g_report_handle;
    if (!h) return;
    std::printf("rsrec report: library_calls=%ld device_assemblies=%ld operator_arrays_from_device=%d device_ldos_calls=%ld\n", h->n_recursion_calls, h->n_asm_calls,
                h->n_asm_reused, h->n_ldos_calls);
    std::fflush(stdout);
}

extern "C" int rsrec_create(rsrec_t** out, int device) {
    if (!out) return RSREC_ERR_ARG;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return RSREC_ERR_DEVICE;   // no CPU fallback by design
    if (device < 0 || device >= ndev) return RSREC_ERR_ARG;
    if (hipSetDevice(device) != hipSuccess) return RSREC_ERR_DEVICE;
    rsrec_t* h = new rsrec_handle();
    h->device = device;
    if (hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess) { delete h; return RSREC_ERR_DEVICE; }
    if (hipDeviceGetAttribute(&h->n_cu, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || h->n_cu <= 0) h->n_cu = 256;
    *out = h;
    if (getenv("RSREC_REPORT") && !g_report_handle) {
        static bool registered = false;
        if (!registered) { registered = std::atexit(report_at_exit) == 0; }
        g_report_handle = h;
    }
    return RSREC_OK;
}

extern "C" int rsrec_destroy(rsrec_t* h) {
    if (!h) return RSREC_ERR_ARG;
    if (g_report_handle == h) { report_at_exit(); g_report_handle = nullptr; }
    (void)hipSetDevice(h->device);
    (void)hipStreamSynchronize(h->stream);
    (void)rsrec_comm_destroy(h);
    h->d_comm.release();
    for (auto e : h->ev_pool) (void)hipEventDestroy(e);
    DevBuf* all[] = {&h->d_bsqrt, &h->d_term, &h->d_gim, &h->d_ldos, &h->d_green_in, &h->d_green_out, &h->d_kubo[0], &h->d_kubo[1], &h->d_kubo[2], &h->d_kubo[3], &h->d_kubo[4], &h->d_nbr, &h->d_nbr5, &h->d_s5queue, &h->d_iz, &h->d_hst, &h->d_hloc, &h->d_host, &h->d_holoc, &h->d_enim, &h->d_lsham, &h->d_vec[0], &h->d_vec[1],
                     &h->d_vec[2], &h->d_vec[3], &h->d_vec[4], &h->d_vec[5], &h->d_order, &h->d_cum, &h->d_partial, &h->d_partial2, &h->d_coefA, &h->d_coefB, &h->d_bmats,
                     &h->d_status, &h->d_frags, &h->d_seed, &h->d_seedcoef, &h->d_mu, &h->d_scal, &h->d_zsqr};
    for (auto b : all) b->release();
    for (auto* e : h->region_cache) { e->order.release(); e->cum.release(); delete e; }
    h->s4_op.release();
    h->s5_op.release();
    h->kubo_op[0].release(); h->kubo_op[1].release(); h->kubo_hbulk.release(); h->orb_plain.release(); h->s5_la.release(); h->d_la_extra.release();
    if (h->pin) (void)hipHostFree(h->pin);
    (void)hipStreamDestroy(h->stream);
    if (h->copy_stream) (void)hipStreamDestroy(h->copy_stream);
    for (hipEvent_t e : h->ev_green) if (e) (void)hipEventDestroy(e);
    if (h->graph_exec) (void)hipGraphExecDestroy(h->graph_exec);
    if (h->side_stream) (void)hipStreamDestroy(h->side_stream);
    if (h->oct_stream) (void)hipStreamDestroy(h->oct_stream);
    if (h->ev_oct_in) (void)hipEventDestroy(h->ev_oct_in);
    if (h->ev_oct_out) (void)hipEventDestroy(h->ev_oct_out);
    if (h->ev_orth) (void)hipEventDestroy(h->ev_orth);
    if (h->ev_bred) (void)hipEventDestroy(h->ev_bred);
    delete h;
    return RSREC_OK;
}

extern "C" int rsrec_last_error(rsrec_t* h, char* buf, size_t n) {
    if (!h || !buf || n == 0) return RSREC_ERR_ARG;
    snprintf(buf, n, "%s", h->err.c_str());
    return RSREC_OK;
}

extern "C" int rsrec_set_option(rsrec_t* h, const char* key, long value) {
    if (!h || !key) return RSREC_ERR_ARG;
    if (!strcmp(key, "batch")) h->opt_batch = value;
    else if (!strcmp(key, "kernels")) h->opt_kernels = value;
    else if (!strcmp(key, "nblk")) h->opt_nblk = value;
    else if (!strcmp(key, "spmm5")) h->opt_spmm5 = value;
    else if (!strcmp(key, "chain_fold")) h->opt_chain_fold = value;
    else if (!strcmp(key, "s5_cap")) h->opt_s5_cap = value;
    else if (!strcmp(key, "s5_lds")) h->opt_s5_lds = value;
    else if (!strcmp(key, "s5_queue")) h->opt_s5_queue = value;
    else if (!strcmp(key, "cheb_fused")) h->opt_cheb_fused = value;
    else if (!strcmp(key, "side_stream")) h->opt_side = value;
    else if (!strcmp(key, "graph")) h->opt_graph = value;
    else if (!strcmp(key, "orth3")) h->opt_orth3 = value;
    else if (!strcmp(key, "s5_waves")) h->opt_s5_waves = value;
    else if (!strcmp(key, "s5_split")) h->opt_s5_split = value;
    else if (!strcmp(key, "orth_oop")) h->opt_orth_oop = value;
    else if (!strcmp(key, "sat_pct")) h->opt_sat_pct = value;
    else if (!strcmp(key, "kubo_lchunk")) h->opt_kubo_lchunk = value;
    else if (!strcmp(key, "kubo_vbatch")) h->opt_kubo_vbatch = value;
    else if (!strcmp(key, "s5_host_emit")) h->opt_s5_host_emit = value;
    else if (!strcmp(key, "s5_octet")) h->opt_s5_octet = value;
    else if (!strcmp(key, "s5_spin_xcd")) h->opt_s5_spin_xcd = value;
    else if (!strcmp(key, "s5_run_min")) h->opt_s5_run_min = value;
    else return fail(h, RSREC_ERR_ARG, "unknown option '%s'", key);
    return RSREC_OK;
}

extern "C" int rsrec_get_timing(rsrec_t* h, double* out, int n) {
    if (!h || !out) return RSREC_ERR_ARG;
    const double v[12] = {h->t_total_ms, h->t_hop_ms, h->n_hop_launch, h->n_atom_steps, h->n_block_mult, h->t_rest_ms, h->t_host_ms, (double)h->hop_fuses_a, h->n_hop_mfma_flop,
                          h->n_req_flop, (double)h->n_asm_reused, (double)h->n_octet_launch};
    for (int i = 0; i < n && i < 12; ++i) out[i] = v[i];
    return RSREC_OK;
}

extern "C" void rsrec_site_partition(int rank, int nprocs, int nsites, int* start_atom, int* end_atom) {
    // get_mpi_variables, mpi.f90:37-46
    int per = nsites / nprocs;
    const int rem = nsites % nprocs;
    int start;
    if (rank < rem) { per += 1; start = rank * per + 1; }
    else start = rank * per + rem + 1;
    *start_atom = start;
    *end_atom = start + per - 1;
}

// cached regions (device order lists) of lattice epochs that ended; the stream is idle between calls
static void release_graph(rsrec_t* h) {
    if (h->graph_exec) { (void)hipDeviceSynchronize(); (void)hipGraphExecDestroy(h->graph_exec); h->graph_exec = nullptr; }
    h->graph_key.clear();
}

// cached regions (device order lists) of lattice epochs that ended; the stream is idle between calls.  The captured level loop holds
// the region lists, the lattice tables and their dimensions BY VALUE: it goes with them (a new RegionEntry / hipMalloc of the same
// size commonly returns the old address, so pointers alone cannot tell a new lattice from the old one)
static void release_regions(rsrec_t* h) {
    release_graph(h);
    if (!h->region_cache.empty()) (void)hipDeviceSynchronize();
    for (auto* e : h->region_cache) { e->order.release(); e->cum.release(); delete e; }
    h->region_cache.clear();
    h->cur_order = nullptr; h->cur_cum = nullptr; h->cur_entry = nullptr; h->cur_level_max = nullptr; h->cur_level_groups = nullptr; h->cur_mult_hist = nullptr;
}

extern "C" int rsrec_set_lattice(rsrec_t* h, int kk, int nncols, const int32_t* nn, const int32_t* iz, int nmax, int ntype) {
    if (!h || !nn || !iz || kk <= 0 || nncols < 1 || nmax < 0 || nmax > kk || ntype < 1) return fail(h, RSREC_ERR_ARG, "rsrec_set_lattice: bad argument");
    HIPCK(h, hipSetDevice(h->device));
    // unchanged lattice (every SCF iteration of the reference calls the drivers with the same lattice%nn): keep the device tables
    // and, above all, the cached regions -- building them costs as much host time as the recursion costs device time
    if (h->have_lattice && kk == h->kk && nncols == h->lat_nncols && nmax == h->nmax && ntype == h->ntype &&
        h->lat_nn.size() == (size_t)kk * nncols && std::memcmp(h->lat_nn.data(), nn, h->lat_nn.size() * sizeof(int32_t)) == 0 &&
        std::memcmp(h->lat_iz.data(), iz, (size_t)kk * sizeof(int32_t)) == 0)
        return RSREC_OK;
    // validate and convert into local tables first: a refused call leaves the handle as it was
    int nslots = 1;
    for (int i = 0; i < kk; ++i) {
        const int nr = nn[i];
        if (nr < 0 || nr > nncols) return fail(h, RSREC_ERR_ARG, "rsrec_set_lattice: nn(%d,1)=%d outside 0..%d", i + 1, nr, nncols);
        nslots = std::max(nslots, nr);
        if (iz[i] < 1 || iz[i] > ntype) return fail(h, RSREC_ERR_ARG, "rsrec_set_lattice: iz(%d)=%d outside 1..%d", i + 1, iz[i], ntype);
    }
    std::vector<int> nbr_new((size_t)kk * nslots, -1), iz_new(kk), deg(kk + 1, 0);
    for (int i = 0; i < kk; ++i) {
        iz_new[i] = iz[i] - 1;
        nbr_new[(size_t)i * nslots] = i;
        const int nr = nn[i];
        for (int j = 1; j < nr; ++j) {   // slots 2..nn(i,1) of the reference (recursion.f90:1614)
            const int n = nn[(size_t)i + (size_t)kk * j];
            if (n == 0) continue;
            if (n < 0 || n > kk) return fail(h, RSREC_ERR_ARG, "rsrec_set_lattice: nn(%d,%d)=%d outside 0..%d", i + 1, j + 1, n, kk);
            nbr_new[(size_t)i * nslots + j] = n - 1;
            deg[n - 1]++;
        }
    }
    // from here on the handle changes: until the last table is uploaded it holds no lattice (a failed upload must not leave the
    // unchanged-lattice shortcut above pointing at half-replaced tables), and the cached regions of the old lattice are released
    h->have_lattice = false;
    h->have_ham = false;
    h->lat_nn.clear(); h->lat_iz.clear();
    release_regions(h);
    h->kk = kk; h->nslots = nslots; h->nmax = nmax; h->ntype = ntype;
    h->nbr.swap(nbr_new);
    h->iz0.swap(iz_new);
    h->radj_ptr.assign(kk + 1, 0);
    for (int n = 0; n < kk; ++n) h->radj_ptr[n + 1] = h->radj_ptr[n] + deg[n];
    h->radj.resize(h->radj_ptr[kk]);
    std::vector<int> fill(h->radj_ptr.begin(), h->radj_ptr.end() - 1);
    for (int i = 0; i < kk; ++i)
        for (int j = 1; j < nslots; ++j) {
            const int n = h->nbr[(size_t)i * nslots + j];
            if (n >= 0) h->radj[fill[n]++] = i;
        }
    HIPCK(h, h->d_nbr.reserve(h->nbr.size() * sizeof(int)));
    HIPCK(h, h->d_iz.reserve((size_t)kk * sizeof(int)));
    HIPCK(h, hipMemcpy(h->d_nbr.p, h->nbr.data(), h->nbr.size() * sizeof(int), hipMemcpyHostToDevice));
    HIPCK(h, hipMemcpy(h->d_iz.p, h->iz0.data(), (size_t)kk * sizeof(int), hipMemcpyHostToDevice));
    {
        // (kk+1) x (nslots+2): absent neighbours and the extra row -> zero block; column nslots = the atom itself (extra on-site slot),
        // column nslots + 1 = zero block (null entries of the k_spmm5 schedule)
        std::vector<int> n5((size_t)(kk + 1) * (nslots + 2), kk);
        for (int i = 0; i < kk; ++i) {
            for (int j = 0; j < nslots; ++j) { const int n = h->nbr[(size_t)i * nslots + j]; if (n >= 0) n5[(size_t)i * (nslots + 2) + j] = n; }
            n5[(size_t)i * (nslots + 2) + nslots] = i;
        }
        HIPCK(h, h->d_nbr5.reserve(n5.size() * sizeof(int)));
        HIPCK(h, hipMemcpy(h->d_nbr5.p, n5.data(), n5.size() * sizeof(int), hipMemcpyHostToDevice));
    }
    spatial_key_from_graph(h);
    h->lat_nn.assign(nn, nn + (size_t)kk * nncols);
    h->lat_iz.assign(iz, iz + kk);
    h->lat_nncols = nncols;
    h->lat_cr.clear();
    h->have_lattice = true;
    h->lattice_epoch++;
    h->have_ham = false;   // operator tables depend on nmax/ntype: must be set again
    return RSREC_OK;
}

extern "C" int rsrec_set_positions(rsrec_t* h, const double* cr) {
    if (!h || !cr) return fail(h, RSREC_ERR_ARG, "rsrec_set_positions: bad argument");
    if (!h->have_lattice) return fail(h, RSREC_ERR_ARG, "rsrec_set_positions: call rsrec_set_lattice first");
    const int kk = h->kk;
    if (h->lat_cr.size() == 3 * (size_t)kk && std::memcmp(h->lat_cr.data(), cr, h->lat_cr.size() * sizeof(double)) == 0) return RSREC_OK;
    h->lat_cr.assign(cr, cr + 3 * (size_t)kk);
    double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
    for (int i = 0; i < kk; ++i)
        for (int a = 0; a < 3; ++a) { lo[a] = std::min(lo[a], cr[3 * (size_t)i + a]); hi[a] = std::max(hi[a], cr[3 * (size_t)i + a]); }
    double span = 1e-300;
    for (int a = 0; a < 3; ++a) span = std::max(span, hi[a] - lo[a]);
    h->spatial_key.resize(kk);
    for (int i = 0; i < kk; ++i) {
        unsigned q[3];
        for (int a = 0; a < 3; ++a) q[a] = (unsigned)std::min(1023.0, std::max(0.0, (cr[3 * (size_t)i + a] - lo[a]) / span * 1023.0));
        h->spatial_key[i] = morton3(q[0], q[1], q[2]);
    }
    h->lattice_epoch++;        // cached region orders were built with the old keys: they can never match again
    release_regions(h);
    return RSREC_OK;
}

extern "C" int rsrec_set_hamiltonian(rsrec_t* h, int nslots, int hoh, int nsp, const double* ee, const double* lsham, const double* eeo,
                                     const double* enim, const double* hall, const double* hallo) {
    if (!h) return RSREC_ERR_ARG;
    if (!h->have_lattice) return fail(h, RSREC_ERR_ARG, "rsrec_set_hamiltonian: call rsrec_set_lattice first");
    if (!ee || !lsham || nslots < h->nslots) return fail(h, RSREC_ERR_ARG, "rsrec_set_hamiltonian: ee/lsham missing or nslots=%d < lattice slots %d", nslots, h->nslots);
    if (hoh && (!eeo || !enim)) return fail(h, RSREC_ERR_ARG, "rsrec_set_hamiltonian: hoh requires eeo and enim");
    if (h->nmax > 0 && (!hall || (hoh && !hallo))) return fail(h, RSREC_ERR_ARG, "rsrec_set_hamiltonian: nmax>0 requires hall (and hallo with hoh)");
    HIPCK(h, hipSetDevice(h->device));
    const size_t B = 2 * (size_t)BLK;   // doubles per block
    const int ntype = h->ntype, nmax = h->nmax;
    h->hslots = nslots; h->hoh = hoh ? 1 : 0; h->nsp = nsp;
    h->host_ee.assign(ee, ee + 2 * (size_t)BLK * nslots * h->ntype);
    h->host_lsham.assign(lsham, lsham + 2 * (size_t)BLK * h->ntype);
    h->host_eeo.clear(); h->host_enim.clear(); h->host_hall.clear(); h->host_hallo.clear();
    if (hoh) { h->host_eeo.assign(eeo, eeo + 2 * (size_t)BLK * nslots * h->ntype); h->host_enim.assign(enim, enim + 2 * (size_t)BLK * h->ntype); }
    if (h->nmax > 0) { h->host_hall.assign(hall, hall + 2 * (size_t)BLK * nslots * h->nmax); if (hoh) h->host_hallo.assign(hallo, hallo + 2 * (size_t)BLK * nslots * h->nmax); }
    h->s5_la_ok = 0;
    // stencil with the on-site spin-orbit block folded into slot 0 (locham = ee(:,:,1,ih) + lsham(:,:,ih), recursion.f90:1608)
    std::vector<double> st(ee, ee + B * nslots * ntype);
    if (!hoh)
        for (int t = 0; t < ntype; ++t)
            for (size_t e = 0; e < B; ++e) st[B * nslots * t + e] += lsham[B * t + e];
    // Arrays that rsrec_assemble_blocks produced (the caller hands back, bit for bit, what it was given) are already on the device:
    // a device-to-device copy (+ the l.s fold) replaces the upload.  Anything else -- arrays built on the host, or edited since -- is uploaded.
    h->n_asm_reused = 0;
    auto resident = [&](int part, int which, const double* arr, int ncls) -> const double* {
        const size_t n = B * nslots * ncls;
        if (!arr || h->asm_nslots[part] != nslots || h->asm_ncls[part] != ncls || h->asm_host[part][which].size() != n) return nullptr;
        if (std::memcmp(arr, h->asm_host[part][which].data(), n * 8) != 0) return nullptr;
        h->n_asm_reused++;
        return h->d_asm[part][which].as<double>();
    };
    auto place = [&](DevBuf& dst, const double* dev_src, const double* host_src, size_t n) -> hipError_t {
        hipError_t e = dst.reserve(n * 8);
        if (e != hipSuccess) return e;
        if (dev_src) return hipMemcpyAsync(dst.p, dev_src, n * 8, hipMemcpyDeviceToDevice, h->stream);
        return hipMemcpy(dst.p, host_src, n * 8, hipMemcpyHostToDevice);
    };
    HIPCK(h, h->d_lsham.reserve(B * ntype * 8));
    HIPCK(h, hipMemcpy(h->d_lsham.p, lsham, B * ntype * 8, hipMemcpyHostToDevice));
    {
        const double* dev = resident(0, 0, ee, ntype);
        HIPCK(h, place(h->d_hst, dev, st.data(), st.size()));
        if (dev && !hoh) k_fold_onsite<<<ntype, 324, 0, h->stream>>>(h->d_hst.as<double2>(), nslots, h->d_lsham.as<double2>(), nullptr);
    }
    std::vector<double> loc;
    if (nmax > 0) {
        loc.assign(hall, hall + B * nslots * nmax);
        if (!hoh)
            for (int i = 0; i < nmax; ++i)
                for (size_t e = 0; e < B; ++e) loc[B * nslots * i + e] += lsham[B * h->iz0[i] + e];   // :1582
        const double* dev = resident(1, 0, hall, nmax);
        HIPCK(h, place(h->d_hloc, dev, loc.data(), loc.size()));
        if (dev && !hoh) k_fold_onsite<<<nmax, 324, 0, h->stream>>>(h->d_hloc.as<double2>(), nslots, h->d_lsham.as<double2>(), h->d_iz.as<int>());
    }
    if (hoh) {
        HIPCK(h, place(h->d_host, resident(0, 1, eeo, ntype), eeo, B * nslots * ntype));
        HIPCK(h, h->d_enim.reserve(B * ntype * 8));
        HIPCK(h, hipMemcpy(h->d_enim.p, enim, B * ntype * 8, hipMemcpyHostToDevice));
        if (nmax > 0) HIPCK(h, place(h->d_holoc, resident(1, 1, hallo, nmax), hallo, B * nslots * nmax));
    }
    HIPCK(h, hipGetLastError());
    // MFMA-fragment form of the same operator tables.  k_spmm5's streams are assembled on the DEVICE from the raw blocks uploaded above
    // (Spmm5Operator::build -> k_s5_emit; option s5_host_emit = 1: on the host, the round-2 path, kept as the cross-check).  k_spmm4's
    // tables (small launches of the plain operator only) are built when a call first needs them (ensure_s4).
    {
        h->s4_built_split = 0; h->s5_built = 0;
        if (h->nslots + 1 <= S4_MAXSLOTS) {
            h->host_st.swap(st);
            h->host_loc.swap(loc);
            const double* dev[6] = {h->d_hst.as<double>(), h->d_hloc.as<double>(), h->d_host.as<double>(), h->d_holoc.as<double>(), h->d_enim.as<double>(), h->d_lsham.as<double>()};
            const char* msg = h->s5_op.build(h->nslots, nslots, ntype, nmax, h->hoh, h->host_st.data(), nmax > 0 ? h->host_loc.data() : nullptr, hoh ? eeo : nullptr,
                                             (hoh && nmax > 0) ? hallo : nullptr, hoh ? enim : nullptr, lsham, h->iz0.data(), h->opt_s5_host_emit ? nullptr : dev, h->stream);
            if (msg) return fail(h, RSREC_ERR_DEVICE, "rsrec_set_hamiltonian: %s", msg);
            h->s5_built = 1;
        }
    }
    h->have_ham = true;
    return RSREC_OK;
}

extern "C" int rsrec_assemble_blocks(rsrec_t* h, int part, int ncls, int nslots, int hoh, const double* hmag, const int32_t* nbr_type, const double* obarm, int ntype,
                                     double* blocks, double* blocks_o) {
    if (!h) return RSREC_ERR_ARG;
    if (part < 0 || part > 1 || ncls < 1 || nslots < 1 || !hmag || !blocks) return fail(h, RSREC_ERR_ARG, "rsrec_assemble_blocks: part=%d ncls=%d nslots=%d or a missing array", part, ncls, nslots);
    if (hoh && (!nbr_type || !obarm || !blocks_o || ntype < 1)) return fail(h, RSREC_ERR_ARG, "rsrec_assemble_blocks: hoh needs nbr_type, obarm (ntype=%d) and blocks_o", ntype);
    if (hoh)
        for (size_t q = 0; q < (size_t)ncls * nslots; ++q)
            if (nbr_type[q] < 0 || nbr_type[q] > ntype) return fail(h, RSREC_ERR_ARG, "rsrec_assemble_blocks: nbr_type[%zu]=%d outside 0..%d", q, nbr_type[q], ntype);
    HIPCK(h, hipSetDevice(h->device));
    h->n_asm_calls++;
    h->asm_nslots[part] = h->asm_ncls[part] = 0;          // nothing valid while this runs
    const size_t nblk = (size_t)ncls * nslots, hm_bytes = nblk * 4 * 81 * 16, ty_bytes = nblk * 4, ob_bytes = hoh ? (size_t)ntype * 324 * 16 : 0;
    const size_t ty_off = (hm_bytes + 255) / 256 * 256, ob_off = ty_off + (ty_bytes + 255) / 256 * 256;
    HIPCK(h, h->d_asm_in.reserve(ob_off + ob_bytes + 256));
    char* in = h->d_asm_in.as<char>();
    XFER(xfer_h2d(h, in, hmag, hm_bytes));
    if (hoh) { XFER(xfer_h2d(h, in + ty_off, nbr_type, ty_bytes)); XFER(xfer_h2d(h, in + ob_off, obarm, ob_bytes)); }
    HIPCK(h, h->d_asm[part][0].reserve(nblk * 324 * 16));
    if (hoh) HIPCK(h, h->d_asm[part][1].reserve(nblk * 324 * 16));
    k_assemble_blocks<<<dim3(nslots, ncls), 384, 0, h->stream>>>(reinterpret_cast<const double2*>(in), reinterpret_cast<const int*>(in + ty_off),
                                                                 reinterpret_cast<const double2*>(in + ob_off), nslots, hoh ? 1 : 0, h->d_asm[part][0].as<double2>(),
                                                                 hoh ? h->d_asm[part][1].as<double2>() : nullptr);
    HIPCK(h, hipGetLastError());
    XFER(xfer_d2h(h, blocks, h->d_asm[part][0].p, nblk * 324 * 16));
    if (hoh) XFER(xfer_d2h(h, blocks_o, h->d_asm[part][1].p, nblk * 324 * 16));
    h->asm_host[part][0].assign(blocks, blocks + nblk * 648);
    if (hoh) h->asm_host[part][1].assign(blocks_o, blocks_o + nblk * 648); else h->asm_host[part][1].clear();
    h->asm_nslots[part] = nslots; h->asm_ncls[part] = ncls; h->asm_hoh[part] = hoh ? 1 : 0;
    return RSREC_OK;
}

namespace {

DevProblem make_problem(const rsrec_t* h) {
    DevProblem P;
    P.kk = h->kk; P.nslots = h->nslots; P.hstride = h->hslots; P.nmax = h->nmax; P.hoh = h->hoh;
    P.nbr = h->d_nbr.as<int>(); P.iz = h->d_iz.as<int>();
    P.h_st = h->d_hst.as<double2>(); P.h_loc = h->d_hloc.as<double2>();
    P.ho_st = h->d_host.as<double2>(); P.ho_loc = h->d_holoc.as<double2>();
    P.enim = h->d_enim.as<double2>(); P.lsham = h->d_lsham.as<double2>();
    return P;
}

struct BatchPlan {
    int batch = 1, nblk = 1;
};

// how many chains are advanced together, and how many workgroups each gets
int plan_batch(rsrec_t* h, int nchains, int nvec, size_t vec_elems_per_chain, BatchPlan& bp) {
    for (auto& kb : h->d_kubo) kb.release();                     // (a Kubo call keeps its buffers for the next one; the recursion takes the memory back)
    size_t free_b = 0, total_b = 0;
    HIPCK(h, hipMemGetInfo(&free_b, &total_b));
    size_t reusable = 0;
    for (int v = 0; v < 5; ++v) reusable += h->d_vec[v].bytes;
    const double per_chain = (double)nvec * vec_elems_per_chain * sizeof(double2) + (double)h->kk * 4 + 4096;
    long cap = (long)((0.85 * (double)(free_b + reusable)) / per_chain);
    if (cap < 1) return fail(h, RSREC_ERR_DEVICE, "not enough device memory for one chain (%.1f MB needed, %.1f MB free)", per_chain / 1e6, free_b / 1e6);
    long b = h->opt_batch > 0 ? h->opt_batch : 64;
    b = std::min<long>(b, cap);
    b = std::min<long>(b, nchains);
    bp.batch = (int)std::max<long>(b, 1);
    long nblk = h->opt_nblk > 0 ? h->opt_nblk : std::max<long>(2048 / bp.batch, 16);
    nblk = std::min<long>(nblk, 256);
    nblk = std::min<long>(nblk, (h->kk + TILE_ATOMS - 1) / TILE_ATOMS);
    bp.nblk = (int)std::max<long>(nblk, 1);
    return RSREC_OK;
}

// Upload the regions of one batch. seeds: [nb][nseed] 0-based.
// grouped = true: every BFS level is sorted by operator class tau (per-atom blocks first, then types) and each class run is
// padded with -1 to a multiple of GROUP, so that 8 consecutive entries always share their operator blocks (MFMA kernels).
int upload_regions(rsrec_t* h, const int* seeds0, int nb, int nseed, int nlev, int napply, bool two_pass, bool grouped, int& ostride,
                   double& atom_steps, double& block_mults) {
    const int kk = h->kk;
    // order row = [one list per level of the growing region, each sorted by (operator class, position) | the same for ALL atoms]; every
    // list is padded into operator-class groups of 8 when grouped.  (Rounds 1-3 kept ONE level-major list -- shell after shell, a level's
    // atoms a prefix of it -- which costs 4 bytes per atom and chain but walks a level shell by shell: the waves of an XCD then work on a band
    // of one shell while the blocks they gather lie in the bands of the two neighbouring shells, handled elsewhere and at other times.
    // Measured per launch on 64 x 22^3 / 46^3 (tools/per_level_pmc.sh, per_level_trace.sh): L2 hit rate 0.38-0.41 on the growing levels
    // against 0.55 on the list of all atoms, 1.7x the fabric bytes and 1.5x the time per atom -- and on the 46^3 cell the regions grow for 26
    // of the 49 levels.  A list per level, position-sorted as a whole, gives the growing region the locality of the saturated one.)
    // The streaming passes behind H|psi> (Gram sums, orthogonalisation, moment sums) read every active block once and gather nothing: they keep
    // the level-major list (shell after shell, each shell sorted by class and position; a level = a prefix), whose walk is closer to address
    // order -- on the per-level lists k_mfma_orth3 was 2.8 % slower (46^3, same box).  row = [level-major | per level | all atoms].
    const int cap = grouped ? kk + 7 * h->nmax + 7 * h->ntype + 8 : kk;          // capacity of one list of all atoms
    const int cap_pre = ((grouped ? kk + 7 * h->nmax + 7 * h->ntype * nlev + 8 : kk) + GROUP - 1) / GROUP * GROUP;   // capacity of the level-major list (a multiple of 8: every list starts on a group border)
    const int flags = (two_pass ? 1 : 0) | (grouped ? 2 : 0) | ((int)std::min<long>(100, std::max<long>(1, h->opt_sat_pct)) << 2) | (nseed << 9);
    for (auto* e : h->region_cache)
        if (e->epoch == h->lattice_epoch && e->nlev == nlev && e->napply == napply && e->flags == flags && (int)e->seeds.size() == nb * nseed &&
            std::equal(e->seeds.begin(), e->seeds.end(), seeds0)) {
            h->cur_order = e->order.as<int>(); h->cur_cum = e->cum.as<int>(); h->cur_nrows = nb; h->cur_level_max = &e->level_max; h->cur_level_groups = &e->level_groups;
            h->cur_mult_hist = &e->mult_hist;
            h->cur_entry = e;
            ostride = e->ostride;
            atom_steps += e->atom_steps; block_mults += e->block_mults;
            return RSREC_OK;
        }
    const int ntau_h = h->nmax + h->ntype, nfs_h = h->nslots + 1;
    const int sat_pct = (int)std::min<long>(100, std::max<long>(1, h->opt_sat_pct));
    const int sat_from = (int)(((long)sat_pct * kk + 99) / 100);   // regions of at least this many atoms run on the list of all atoms (100: only the whole lattice)
    // the regions first (a few short-lived host threads; see below), then the row size they need
    std::vector<Region> regs(nb);
    auto run_threads = [&](auto&& fn) {
        const int nthr = std::max(1, std::min({nb, 8, (int)std::thread::hardware_concurrency()}));
        if (nthr == 1) { for (int c = 0; c < nb; ++c) fn(c); return; }
        std::atomic<int> next_chain{0};
        std::vector<std::thread> pool;
        for (int t = 0; t < nthr; ++t)
            pool.emplace_back([&]() { for (int c = next_chain++; c < nb; c = next_chain++) fn(c); });
        for (auto& th : pool) th.join();
    };
    run_threads([&](int c) { grow_region(h, seeds0 + (size_t)c * nseed, nseed, nlev, regs[c]); });
    size_t lvneed = 8;
    for (int c = 0; c < nb; ++c) {
        size_t need = 0;
        for (int L = 0; L < nlev; ++L) {
            const int n = regs[c].cum[L];
            if (n >= sat_from || (L > 0 && n == regs[c].cum[L - 1])) continue;       // list of all atoms / the previous level's list again
            need += (size_t)n + (grouped ? (size_t)7 * std::min(ntau_h, n) + 8 : 0);
        }
        lvneed = std::max(lvneed, need);
    }
    lvneed = (lvneed + 7) / 8 * 8;
    if (lvneed + (size_t)cap > (size_t)INT32_MAX) return fail(h, RSREC_ERR_ARG, "region lists of %zu entries per chain exceed the index range", lvneed + (size_t)cap);
    if ((size_t)cap_pre + lvneed + (size_t)cap > (size_t)INT32_MAX) return fail(h, RSREC_ERR_ARG, "region lists of %zu entries per chain exceed the index range", (size_t)cap_pre + lvneed + (size_t)cap);
    const int sat_off = cap_pre + (int)lvneed;
    ostride = sat_off + cap;
    std::vector<int> order((size_t)nb * ostride, -1), cum((size_t)4 * nb * nlev, 0);      // counts, offsets (H|psi>); counts, offsets (streaming passes)
    std::vector<double> as(nb, 0.0), bm(nb, 0.0);
    const size_t hist_n = (size_t)2 * ntau_h * nfs_h;
    std::vector<double> hist((size_t)nb * hist_n, 0.0);
    auto tau = [&](int i) { return i < h->nmax ? i : h->nmax + h->iz0[i]; };
    const std::vector<unsigned>& key = h->spatial_key;
    auto before = [&](int x, int y) {                       // operator class first (groups must be homogeneous), then position
        const int tx = grouped ? tau(x) : 0, ty = grouped ? tau(y) : 0;
        if (tx != ty) return tx < ty;
        if (key[x] != key[y]) return key[x] < key[y];
        return x < y;
    };
    // the list of all atoms is the same for every chain of the lattice
    std::vector<int> all(kk), sat_list;
    for (int i = 0; i < kk; ++i) all[i] = i;
    std::sort(all.begin(), all.end(), before);
    sat_list.reserve(cap);
    for (int q = 0; q < kk; ++q) {
        if (grouped && q > 0 && tau(all[q]) != tau(all[q - 1])) while (sat_list.size() % GROUP) sat_list.push_back(-1);
        sat_list.push_back(all[q]);
    }
    if (grouped) while (sat_list.size() % GROUP) sat_list.push_back(-1);
    const int sat_count = (int)sat_list.size();
    // One region per chain, built by a few short-lived host threads.  (Not OpenMP: its idle workers spin for 200 ms after a
    // parallel region -- one per visible CPU, 256 on the GPU boxes -- and burn the process's CPU quota: the host thread was then
    // descheduled for 60-90 ms at a time during the next two or three calls, inside whatever HIP call it happened to be in.)
    std::vector<double> lgroups((size_t)nb * nlev * ntau_h, 0.0);     // [chain][level][class]: groups of the level's own list
    auto build_chain = [&](int c) {
        const Region& R = regs[c];
        int* orow = order.data() + (size_t)c * ostride;
        int* crow = cum.data() + (size_t)c * nlev;
        int* brow = cum.data() + (size_t)(nb + c) * nlev;
        int* crow2 = cum.data() + (size_t)(2 * nb + c) * nlev;
        int* brow2 = cum.data() + (size_t)(3 * nb + c) * nlev;
        std::copy(sat_list.begin(), sat_list.end(), orow + sat_off);
        int w = cap_pre, wp = 0;
        std::vector<int> lev, region, merged;
        for (int L = 0; L < nlev; ++L) {
            const int lo = L ? R.cum[L - 1] : 0, hi = R.cum[L];
            // once the region covers most of the lattice the sorted list of all atoms is used instead: blocks outside
            // the region are exactly zero, so a superset changes nothing
            if (hi >= sat_from) { crow[L] = crow2[L] = sat_count; brow[L] = brow2[L] = sat_off; continue; }
            lev.assign(R.order.begin() + lo, R.order.begin() + hi);
            std::sort(lev.begin(), lev.end(), before);
            for (size_t q = 0; q < lev.size(); ++q) {            // level-major list: this shell behind the earlier ones
                if (grouped && q > 0 && tau(lev[q]) != tau(lev[q - 1])) while (wp % GROUP) orow[wp++] = -1;
                orow[wp++] = lev[q];
            }
            if (grouped) while (wp % GROUP) orow[wp++] = -1;
            crow2[L] = wp; brow2[L] = 0;
            if (L > 0 && hi == lo) { crow[L] = crow[L - 1]; brow[L] = brow[L - 1]; continue; }     // the region has stopped growing: the previous level's list again
            merged.resize(region.size() + lev.size());
            std::merge(region.begin(), region.end(), lev.begin(), lev.end(), merged.begin(), before);
            region.swap(merged);
            brow[L] = w;
            double* G = lgroups.data() + ((size_t)c * nlev + L) * ntau_h;
            for (size_t q = 0; q < region.size(); ++q) {
                if (grouped && q > 0 && tau(region[q]) != tau(region[q - 1])) while (w % GROUP) orow[w++] = -1;
                if (w % GROUP == 0) G[grouped ? tau(region[q]) : 0] += 1.0;
                orow[w++] = region[q];
            }
            if (grouped) while (w % GROUP) orow[w++] = -1;
            crow[L] = w - brow[L];
        }
        // bookkeeping in the reference's terms: application t (1..napply) multiplies one block per (atom, slot) whose
        // source atom lies in the region before it; post-hop work runs on the region after it.
        double a_s = 0.0, b_m = 0.0;
        for (int t = 1; t <= napply; ++t) {
            const int lv_after = two_pass ? 2 * t : t;
            a_s += R.cum[std::min(lv_after, nlev - 1)];
        }
        std::vector<int> lev_of(kk, -1);
        {
            int lv = 0;
            for (int q = 0; q < R.cum[nlev - 1]; ++q) {
                while (q >= R.cum[lv]) ++lv;
                lev_of[R.order[q]] = lv;
            }
        }
        // uses1[d] / uses2[d]: applications whose first / second (hoh) pass sees a source at distance d inside the region
        std::vector<double> uses1(nlev + 1, 0.0), uses2(nlev + 1, 0.0);
        for (int d = 0; d < nlev; ++d)
            for (int t = 1; t <= napply; ++t) {
                if (!two_pass) { if (d <= t - 1) uses1[d] += 1.0; }
                else { if (d <= 2 * (t - 1)) uses1[d] += 1.0; if (d <= 2 * t - 1) uses2[d] += 1.0; }
            }
        // one multiplication per (target atom i, slot s) whose source nbr(i, s) is active (hop_b :1576-1625), kept by (pass, operator class
        // of the target, slot): the reference's count is their sum, the flops the block structure requires weigh them by block class
        double* H = hist.data() + (size_t)c * hist_n;
        const int ns = h->nslots;
        for (int i = 0; i < kk; ++i) {
            const int ti = tau(i);
            for (int s = 0; s < ns; ++s) {
                const int n = h->nbr[(size_t)i * ns + s];
                if (n < 0 || lev_of[n] < 0) continue;
                H[(size_t)ti * nfs_h + s] += uses1[lev_of[n]];
                if (two_pass) H[((size_t)ntau_h + ti) * nfs_h + s] += uses2[lev_of[n]];
            }
            if (two_pass && lev_of[i] >= 0) H[((size_t)ntau_h + ti) * nfs_h + ns] += uses1[lev_of[i]];   // e_nu psi + l.s psi on-site (:1437-1438), one merged block here
        }
        for (size_t q = 0; q < hist_n; ++q) b_m += H[q];
        if (two_pass) for (int t2 = 0; t2 < ntau_h; ++t2) b_m += H[((size_t)ntau_h + t2) * nfs_h + ns];       // the reference multiplies enim and lsham separately
        as[c] = a_s; bm[c] = b_m;
    };
    run_threads(build_chain);
    double as_sum = 0.0, bm_sum = 0.0;
    for (int c = 0; c < nb; ++c) { as_sum += as[c]; bm_sum += bm[c]; }
    atom_steps += as_sum; block_mults += bm_sum;
    if (h->region_cache.size() >= 256) {          // bounded: drop everything (the engine is idle between calls)
        HIPCK(h, hipStreamSynchronize(h->stream));
        for (auto* e : h->region_cache) { e->order.release(); e->cum.release(); delete e; }
        h->region_cache.clear();
        h->cur_entry = nullptr;
    }
    auto* e = new rsrec_handle::RegionEntry();
    e->seeds.assign(seeds0, seeds0 + (size_t)nb * nseed);
    e->nlev = nlev; e->napply = napply; e->flags = flags; e->epoch = h->lattice_epoch; e->ostride = ostride;
    e->atom_steps = as_sum; e->block_mults = bm_sum;
    e->mult_hist.assign(hist_n, 0.0);
    for (int c = 0; c < nb; ++c)
        for (size_t q = 0; q < hist_n; ++q) e->mult_hist[q] += hist[(size_t)c * hist_n + q];
    h->region_cache.push_back(e);
    HIPCK(h, e->order.reserve(order.size() * 4));
    HIPCK(h, e->cum.reserve(cum.size() * 4));
    XFER(xfer_h2d(h, e->order.p, order.data(), order.size() * 4));
    XFER(xfer_h2d(h, e->cum.p, cum.data(), cum.size() * 4));
    HIPCK(h, hipStreamSynchronize(h->stream));   // order/cum are stack-local vectors
    e->level_max.assign(nlev, 0);
    e->level_groups.assign((size_t)nlev * ntau_h, 0.0);
    {
        // groups by operator class: the saturated list is the same for every chain; a level-major list is walked once per chain
        std::vector<double> sat_hist(ntau_h, 0.0);
        for (int g = 0; g < sat_count / GROUP; ++g) sat_hist[tau(sat_list[(size_t)g * GROUP])] += 1.0;
        e->sat_base = sat_off;
        e->level_sat.assign(nlev, 0);
        if (grouped)
            for (int g = 0; g < sat_count / GROUP; ++g) {
                const int t = tau(sat_list[(size_t)g * GROUP]);
                if (e->sat_runs.empty() || e->sat_runs.back().tau != t) e->sat_runs.push_back({t, g, g + 1});
                else e->sat_runs.back().hi = g + 1;
            }
        for (int c = 0; c < nb; ++c) {
            const double* prev = nullptr;
            for (int l = 0; l < nlev; ++l) {
                const int cnt = cum[(size_t)c * nlev + l];
                e->level_max[l] = std::max(e->level_max[l], cnt);
                const bool sat = cum[(size_t)(nb + c) * nlev + l] == sat_off;   // (level_max: the larger of the two lists' counts sizes the launches)
                e->level_max[l] = std::max(e->level_max[l], cum[(size_t)(2 * nb + c) * nlev + l]);
                if (sat) e->level_sat[l] += 1;
                const double* G = lgroups.data() + ((size_t)c * nlev + l) * ntau_h;
                if (!sat && l > 0 && cum[(size_t)(nb + c) * nlev + l] == cum[(size_t)(nb + c) * nlev + l - 1] && prev) G = prev;     // the previous level's list again
                for (int t2 = 0; t2 < ntau_h; ++t2) e->level_groups[(size_t)l * ntau_h + t2] += sat ? sat_hist[t2] : G[t2];
                prev = sat ? nullptr : G;
            }
        }
    }
    h->cur_order = e->order.as<int>(); h->cur_cum = e->cum.as<int>(); h->cur_nrows = nb; h->cur_level_max = &e->level_max; h->cur_level_groups = &e->level_groups;
    h->cur_mult_hist = &e->mult_hist;
    h->cur_entry = e;
    return RSREC_OK;
}

// Flops of the H|psi> applications of the current region entry that the operator's block structure requires: the reference multiplies
// full 18x18 blocks (zgemm, recursion.f90:1618: 46 656 flop each), but the hopping blocks of a collinear magnet are spin-diagonal
// (hamiltonian.f90:1553-1617) and need half of that.  This -- not the reference's count -- is what a roofline fraction is measured in.
double required_hop_flops(const rsrec_t* h, const Spmm5Operator& op) {
    if (!h->cur_mult_hist) return 0.0;
    const int ntau = h->nmax + h->ntype, nfs = h->nslots + 1;
    const std::vector<double>& H = *h->cur_mult_hist;
    if (H.size() != (size_t)2 * ntau * nfs) return 0.0;
    double f = 0.0;
    for (int pass = 0; pass < 2; ++pass)
        for (int t = 0; t < ntau; ++t)
            for (int s = 0; s < nfs; ++s) {
                const double n = H[((size_t)pass * ntau + t) * nfs + s];
                if (n == 0.0) continue;
                const double w = (op.mixing.empty() || op.ntau != ntau || op.nslots != h->nslots) ? 46656.0 : op.required_flops(pass, t, s);
                f += n * (w > 0.0 ? w : 46656.0);
            }
    return f;
}

void reset_timing(rsrec_t* h) {
    h->t_total_ms = h->t_hop_ms = h->t_rest_ms = h->t_host_ms = 0;
    h->n_hop_launch = h->n_atom_steps = h->n_block_mult = h->n_hop_mfma_flop = h->n_req_flop = 0;
    h->n_octet_launch = 0;
    h->ev_used = 0;
    h->n_recursion_calls++;          // (every timed entry point: recursions, Green / LDOS stages, Kubo moments)
}

double ev_ms(hipEvent_t a, hipEvent_t b) {
    float ms = 0.f;
    if (!a || !b || hipEventElapsedTime(&ms, a, b) != hipSuccess) return 0.0;
    return ms;
}

int check_ready(rsrec_t* h, const char* who) {
    if (!h) return RSREC_ERR_ARG;
    if (!h->have_lattice || !h->have_ham) return fail(h, RSREC_ERR_ARG, "%s: lattice and hamiltonian must be set first", who);
    return RSREC_OK;
}

}  // namespace

// ------------------------------------------------------------------------------------------------------------------
namespace {

// Workgroups per chain of the matrix-core post-hop kernels (Gram, orthogonalisation, Chebyshev step): they hold their 36x36
// coefficient tables in registers and run one wave per SIMD, so long-lived waves win -- about two workgroups per CU over
// the whole batch (measured at 64 chains: 8 per chain 4.3 ms per level, 64 per chain 5.0 ms).  Three fixed bands, so that the
// work assignment (and with it the rounding of the reductions) only changes when the batch size crosses a band.
int mfma_workgroups_per_chain(const rsrec_t* h, int batch) {
    if (h->opt_nblk > 0) return (int)std::min<long>(2 * h->opt_nblk, 256);
    return batch >= 32 ? 8 : (batch >= 16 ? 32 : 256);
}

// launch size of a (level) pass: enough workgroups for the largest chain of the batch at that level, at most `full.x`
// (the kernels' own active_workgroups() keeps each chain's work assignment independent of it)
dim3 level_grid(const rsrec_t* h, dim3 full, int level) {
    if (!h->cur_level_max || level < 0 || level >= (int)h->cur_level_max->size()) return full;
    const int groups = (*h->cur_level_max)[level] / GROUP;
    return dim3(std::max(1, std::min((int)full.x, (groups + MF_WAVES - 1) / MF_WAVES)), full.y);
}

// k_spmm5 launch: level-sized in x; in y one workgroup per `chain_fold` chains (the kernel loops over them)
// x: one workgroup per 4 groups of the largest chain (not capped at `full.x`): every workgroup then does at most one round of
// groups and the hardware dispatcher balances the CUs; with the 256 cap 10 of the 32 workgroups of an XCD did two rounds
// (measured: folding chains into longer-lived workgroups, i.e. LESS dynamic balancing, costs 10-25 %).
dim3 s5_grid(const rsrec_t* h, dim3 full, int level) {
    int gx = full.x;
    if (h->cur_level_max && level >= 0 && level < (int)h->cur_level_max->size()) {
        const int groups = (*h->cur_level_max)[level] / GROUP;
        gx = std::max(1, (groups + S5_WG_GROUPS - 1) / S5_WG_GROUPS);
        if (gx > 8) gx = (gx + 7) / 8 * 8;                 // same number of workgroups on every XCD
        if (h->opt_s5_cap > 0) gx = std::min(gx, (int)h->opt_s5_cap);
    }
    const int fold = (int)std::max<long>(1, h->opt_chain_fold);
    return dim3(gx, (full.y + fold - 1) / fold);
}

// Two-stage reduction of per-workgroup partials (k_presum16): returns the buffer and count the final reduce kernel reads.
const double* presum(rsrec_t* h, const double* partial, int nb, int& nblk, int width, hipStream_t stream = nullptr, int slot = 0) {
    if (nblk <= 32) return partial;
    const int nblk2 = (nblk + 15) / 16;
    const size_t need = (size_t)nb * nblk2 * width;
    if (slot == 0 && h->p2_slot == 0) {
        if (h->d_partial2.reserve(need * sizeof(double)) != hipSuccess) return partial;     // fall back to the single-stage sum
    } else if (need > h->p2_slot) return partial;
    double* out = h->d_partial2.as<double>() + (size_t)slot * h->p2_slot;
    const int nchunk = (width + 255) / 256;
    k_presum16<<<dim3(nblk2 * nchunk, nb), 256, 0, stream ? stream : h->stream>>>(partial, nblk, nblk2, width, out);
    nblk = nblk2;
    return out;
}

// k_spmm5 launch (large launches; CI vectors).  The variant with the operator fragments in LDS serves operators with ONE class of
// atoms (a bulk crystal of one type: every group runs the same stream) whose stream of one spin fits the CU's LDS; per-chain stream
// heads (local-axis runs) keep the global-load variant.
constexpr size_t S5_LDS_LIMIT = 160 * 1024;
// LDS a k_spmm5 workgroup may ask for: what the device grants on request (160 KB on MI355X); asked for once per handle, i.e. per device --
// the attribute is a property of the (function, device) pair, a second handle on another GPU of the process needs its own opt-in
void s5_prepare(rsrec_t* h) {
    if (h->s5_lds_limit != (size_t)-1) return;
    int optin = 0;
    h->s5_lds_limit = 0;
    if (hipDeviceGetAttribute(&optin, hipDeviceAttributeSharedMemPerBlockOptin, h->device) == hipSuccess && optin > 64 * 1024) {
        const int ask = (int)std::min<size_t>((size_t)optin, S5_LDS_LIMIT);
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(k_spmm5<false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, ask) == hipSuccess &&
            hipFuncSetAttribute(reinterpret_cast<const void*>(k_spmm5<true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, ask) == hipSuccess &&
            hipFuncSetAttribute(reinterpret_cast<const void*>(k_spmm5<false, true, false, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, ask) == hipSuccess &&
            hipFuncSetAttribute(reinterpret_cast<const void*>(k_spmm5<true, true, false, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, ask) == hipSuccess)
            h->s5_lds_limit = (size_t)ask;
    }
    (void)hipGetLastError();
}
// one launch: the LDS / persistent form for operator class `one` (>= 0) if it is wanted and fits, else the global-load form
template <bool TWO>
void launch_s5_one(rsrec_t* h, dim3 grid, const SpmmDims& SD, const int* order, const int* cum, const int* iz, const Spmm5Operator& op, int set,
                   const double* in, double* out, const double* in2, const double* extra, int ntau, S5Epilogue epi, int one) {
    const size_t lds_bytes = (size_t)op.ntr * S5_TRIPLE * sizeof(double);
    // s5_lds: 0 never, 1 (default) whenever the groups of the launch are of one class and its stream fits
    const bool want = h->opt_s5_lds >= 1;
    s5_prepare(h);
    const size_t lds_limit = h->s5_lds_limit;
    if (want && one >= 0 && !extra && lds_bytes <= lds_limit) {
        // both output spins on every XCD (workgroup rows alternate between them, an XCD sweeps an eighth of the list) also for collinear
        // operators: the split "even XCDs spin 0, odd XCDs spin 1" of round 2 (an XCD's L2 then holds one spin half of the neighbour blocks)
        // measured 2-4 % slower on every workload at the end of round 3 (tools/ab_spin_xcd.sh); option s5_spin_xcd = 1 brings it back
        const int spin_by_xcd = (op.spin_mixing || !h->opt_s5_spin_xcd) ? 0 : 1;
        const unsigned row = spin_by_xcd ? 8 : 16;
        dim3 g2(std::max(row, (grid.x + row - 1) / row * row), grid.y);
        int* queue = nullptr;
        const unsigned ncu = (unsigned)std::max(16, h->n_cu / 16 * 16);     // whole rows of 8 / 16 workgroups (the kernel's XCD mapping)
        if ((h->opt_s5_queue == 1 && grid.x >= ncu) || h->opt_s5_queue >= 2) {
            // persistent form: one workgroup per CU for the whole launch, groups from per-(chain, XCD[, spin]) counters
            if ((!h->capturing || h->d_s5queue.bytes >= (size_t)SD.nchains * 16 * sizeof(int)) &&     // (no allocation inside a stream capture)
                h->d_s5queue.reserve((size_t)SD.nchains * 16 * sizeof(int)) == hipSuccess &&
                hipMemsetAsync(h->d_s5queue.p, 0, (size_t)SD.nchains * 16 * sizeof(int), h->stream) == hipSuccess) {
                queue = h->d_s5queue.as<int>();
                g2 = dim3(ncu, 1);
            }
        }
        // s5_waves = 4 (persistent form only): half-size workgroups, one wave per SIMD -- the other half of every CU's registers stays free
        // for the kernels of another stream (the HBM-bound post-hop passes of the other half batch)
        const unsigned thr = (queue && h->opt_s5_waves == 4) ? S5_WG_GROUPS * 64 : S5_WG_GROUPS * 128;
        if (queue && h->opt_s5_split == 3) {
            const unsigned thr3 = 64u * (unsigned)std::min<long>(12, std::max<long>(8, h->opt_s5_waves));       // (launch bounds: 768 threads)
            // the split stream requests its operands five steps ahead: up to two triples past the stream's end (never used) -- one more triple of LDS where it fits
            const size_t lds3 = std::min(lds_limit, lds_bytes + (size_t)S5_TRIPLE_BYTES);
            k_spmm5<TWO, true, false, 3><<<g2, thr3, lds3, h->stream>>>(SD, order, cum, h->d_nbr5.as<int>(), iz, op.frag_set(set), op.meta_set(set), op.ntr, in, out, in2, nullptr, ntau, one, queue, spin_by_xcd, epi);
        } else
        k_spmm5<TWO, true><<<g2, thr, lds_bytes, h->stream>>>(SD, order, cum, h->d_nbr5.as<int>(), iz, op.frag_set(set), op.meta_set(set), op.ntr, in, out, in2, nullptr, ntau, one, queue, spin_by_xcd, epi);
    } else
        k_spmm5<TWO, false><<<grid, S5_WG_GROUPS * 128, 0, h->stream>>>(SD, order, cum, h->d_nbr5.as<int>(), iz, op.frag_set(set), op.meta_set(set), op.ntr, in, out, in2, extra, ntau, 0, nullptr, 1, epi);
}

// k_spmm5 launch (CI vectors).  The operator fragments sit in LDS (one copy per persistent workgroup) when all groups a workgroup takes
// run the same stream: operators with ONE class of atoms (a bulk crystal of one type).  Operators with several classes (surfaces,
// compounds, impurity clusters) can take the same form per LARGE CLASS RUN of the class-sorted list of all atoms -- the list every chain
// uses once its region covers the lattice: one persistent launch whose workgroup rows are dealt to the runs, plus one global-load launch
// for what remains (small classes such as the per-atom blocks of an impurity region, chains whose regions are still growing) -- option
// s5_lds = 2; by default they keep the global-load form, which is faster for them.  Per-chain stream heads (local-axis runs): global loads.
template <bool TWO>
void launch_s5(rsrec_t* h, dim3 grid, const SpmmDims& SD0, const int* order, const int* cum, const int* iz, const Spmm5Operator& op, int set,
               const double* in, double* out, const double* in2 = nullptr, const double* extra = nullptr, int ntau = 0, S5Epilogue epi = S5Epilogue()) {
    SpmmDims SD = SD0;
    const int one = op.single_class(set);
    const rsrec_handle::RegionEntry* E = h->cur_entry;
    // s5_lds = 2: the class-run form (measured slower than the global-load form on both multi-class workloads of bench.py -- fccCu001
    // 5.58 vs 5.25 ms per launch, and the launch-per-run variant B2FeCo 3.07 vs 2.36 ms: the runs cannot balance against each other --
    // so it is not the default; the parity tests run it)
    const bool multi = one < 0 && h->opt_s5_lds >= 2 && !extra && E && E->sat_base > 0 && SD.cpo == 1 && SD.level >= 0 && SD.level < (int)E->level_sat.size() &&
                       E->level_sat[SD.level] > 0 && op.ntau == h->nmax + h->ntype && (grid.x >= (unsigned)std::max(16, h->n_cu / 16 * 16) || h->opt_s5_queue >= 2) && (size_t)op.ntr * S5_TRIPLE * sizeof(double) <= (s5_prepare(h), h->s5_lds_limit);
    // Atoms with their own operator blocks (an impurity region: classes 0 .. nmax - 1) are one-tile groups of their own in a chain's lists.
    // From `s5_octet` such atoms on, and once at least half of the (atom, chain) pairs of the batch are inside their chains' regions, a
    // launch of its own forms their groups over 8 CHAINS instead -- the chains share the atom's fragments; an atom outside a chain's
    // region has no active neighbour there and comes out as the zeros it already is -- and the main launch passes over them.
    bool octets = false, oct_aside = false;
    double pa_pairs = 0.0;
    if (h->cur_level_groups && SD.level >= 0 && (size_t)(SD.level + 1) * op.ntau <= h->cur_level_groups->size() && op.ntau == h->nmax + h->ntype)
        for (int t = 0; t < h->nmax; ++t) pa_pairs += (*h->cur_level_groups)[(size_t)SD.level * op.ntau + t];
    if (!multi && one < 0 && h->opt_s5_octet > 0 && h->nmax >= h->opt_s5_octet && !extra && E && SD.cpo == 1 && SD.nchains >= 2 && op.ntau == h->nmax + h->ntype &&
        2.0 * pa_pairs >= (double)h->nmax * SD.nchains && (double)GROUP * (double)(h->kk + 1) * BLD * 8.0 < 4294967296.0) {
        SpmmDims SO = SD;
        const dim3 go((unsigned)((h->nmax + S5_WG_GROUPS - 1) / S5_WG_GROUPS), (unsigned)((SD.nchains + GROUP - 1) / GROUP));
        // beside the main launch, on a stream of its own (the two write disjoint blocks of `out`): alone it is a launch of a few hundred
        // groups with the whole GPU to itself
        hipStream_t so = h->stream;
        if (h->opt_side && !h->oct_stream && !h->capturing &&
            (hipStreamCreateWithFlags(&h->oct_stream, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&h->ev_oct_in, hipEventDisableTiming) != hipSuccess ||
             hipEventCreateWithFlags(&h->ev_oct_out, hipEventDisableTiming) != hipSuccess)) { h->oct_stream = nullptr; (void)hipGetLastError(); }
        if (h->opt_side && h->oct_stream && h->ev_oct_in && h->ev_oct_out && hipEventRecord(h->ev_oct_in, h->stream) == hipSuccess &&
            hipStreamWaitEvent(h->oct_stream, h->ev_oct_in, 0) == hipSuccess) { so = h->oct_stream; oct_aside = true; }
        k_spmm5<TWO, false, true><<<go, S5_WG_GROUPS * 128, 0, so>>>(SO, order, cum, h->d_nbr5.as<int>(), iz, op.frag_set(set), op.meta_set(set), op.ntr, in, out, in2, nullptr, ntau,
                                                                   0, nullptr, 1, epi);
        if (oct_aside) (void)hipEventRecord(h->ev_oct_out, h->oct_stream);
        const bool all_sat = E->sat_base > 0 && SD.level < (int)E->level_sat.size() && E->level_sat[SD.level] == SD.nchains && (int)E->sat_runs.size() >= h->nmax &&
                             E->sat_runs[0].tau == 0 && E->sat_runs[0].lo == 0 && E->sat_runs[h->nmax - 1].tau == h->nmax - 1 && E->sat_runs[h->nmax - 1].hi == h->nmax;
        if (all_sat) {
            // every chain is on the class-sorted list of all atoms, whose first nmax groups are those atoms: the main launch takes the rest
            // of that list as ONE run, so that its XCD chunks are cut from what it serves
            SD.sat_base = E->sat_base; SD.run_lo = h->nmax; SD.run_hi = E->sat_runs.back().hi;
        } else SD.skip_pa = 1;
        octets = true;
        h->n_octet_launch++;
    }
    if (!multi) {
        launch_s5_one<TWO>(h, grid, SD, order, cum, iz, op, set, in, out, in2, extra, ntau, epi, one);
        if (oct_aside) (void)hipStreamWaitEvent(h->stream, h->ev_oct_out, 0);            // the level goes on when both launches are done
    } else {
        // class runs worth workgroups of their own: at least two groups per workgroup of a full persistent launch over the chains on the list
        SD.sat_base = E->sat_base;
        const long min_groups = h->opt_s5_run_min > 0 ? h->opt_s5_run_min : std::max<long>(64, 2L * h->n_cu * 8 / std::max(1, E->level_sat[SD.level]));
        const int spin_by_xcd = op.spin_mixing ? 0 : 1;
        const int row = spin_by_xcd ? 8 : 16;
        const int ncu = std::max(16, h->n_cu / 16 * 16), total_rows = ncu / row;
        SpmmDims SR = SD;
        long sum_groups = 0;
        for (const auto& R : E->sat_runs) {
            if (SR.nruns == 4 || R.hi - R.lo < min_groups || op.ksteps[(size_t)set * op.ntau + R.tau] == 0) continue;
            SR.run_tau[SR.nruns] = R.tau; SR.run_glo[SR.nruns] = R.lo; SR.run_ghi[SR.nruns] = R.hi; ++SR.nruns;
            sum_groups += R.hi - R.lo;
        }
        const size_t qbytes = (size_t)4 * SD.nchains * 16 * sizeof(int);
        const bool can = SR.nruns > 0 && SR.nruns <= total_rows && (h->opt_s5_queue >= 1) && (!h->capturing || h->d_s5queue.bytes >= qbytes) &&
                         h->d_s5queue.reserve(qbytes) == hipSuccess && hipMemsetAsync(h->d_s5queue.p, 0, qbytes, h->stream) == hipSuccess;
        if (can) {
            // ONE persistent launch: the workgroup rows are dealt to the runs in proportion to their groups (at least one row each), every
            // run with its own counters; the launch-per-run form paid one tail per run (measured: B2FeCo 22 %, fccCu001 5 % slower)
            int left = total_rows - SR.nruns, acc = 0;
            for (int r = 0; r < SR.nruns; ++r) {
                const long gr = SR.run_ghi[r] - SR.run_glo[r];
                int extra_rows = r == SR.nruns - 1 ? left : (int)std::min<long>(left, (gr * (total_rows - SR.nruns) + sum_groups / 2) / std::max(1L, sum_groups));
                left -= extra_rows;
                SR.run_row0[r] = acc;
                acc += 1 + extra_rows;
            }
            SR.run_row0[SR.nruns] = acc;
            const size_t lds_bytes = (size_t)op.ntr * S5_TRIPLE * sizeof(double);
            k_spmm5<TWO, true><<<dim3(ncu, 1), S5_WG_GROUPS * 128, lds_bytes, h->stream>>>(SR, order, cum, h->d_nbr5.as<int>(), iz, op.frag_set(set), op.meta_set(set), op.ntr, in, out, in2, nullptr, ntau,
                                                                                          0, h->d_s5queue.as<int>(), spin_by_xcd, epi);
            for (int r = 0; r < SR.nruns; ++r) { SD.skip_lo[r] = SR.run_glo[r]; SD.skip_hi[r] = SR.run_ghi[r]; }
            SD.nskip = SR.nruns;
        }
        launch_s5_one<TWO>(h, grid, SD, order, cum, iz, op, set, in, out, in2, nullptr, ntau, epi, -1);
    }
    if (h->cur_level_groups && SD.level >= 0 && (size_t)(SD.level + 1) * op.ntau <= h->cur_level_groups->size() && SD.cpo == 1 && op.ntau == h->nmax + h->ntype)
        for (int t = 0; t < op.ntau; ++t) {
            double groups = (*h->cur_level_groups)[(size_t)SD.level * op.ntau + t];
            if (octets && t < h->nmax) groups = (double)((SD.nchains + GROUP - 1) / GROUP);          // one group per octet of chains instead of one per chain
            h->n_hop_mfma_flop += groups * op.flops_per_group(set, t);
        }
}

// k_spmm4 addresses a chain's vector with 32-bit byte offsets: only below 4 GiB per chain vector (828 000 atoms)
bool spmm4_usable(const rsrec_t* h) { return h->s5_built && !h->hoh && (size_t)(h->kk + 1) * BLD * sizeof(double) < ((size_t)1 << 32); }

// k_spmm4's fragment tables of the operator as last set (plain operator only: hoh calls always take k_spmm5), built on first use
int ensure_s4(rsrec_t* h) {
    if (h->s4_built_split) return RSREC_OK;
    const char* msg = h->s4_op.build(h->nslots, h->hslots, h->ntype, h->nmax, 0, h->host_st.data(), h->nmax > 0 ? h->host_loc.data() : nullptr, nullptr, nullptr, 1);
    if (msg) return fail(h, RSREC_ERR_DEVICE, "k_spmm4 operator tables: %s", msg);
    h->s4_built_split = 1;
    return RSREC_OK;
}

// small-launch SpMM on LayoutRM vectors: out = sum_slots H_slot in_nbr, four waves share one group of atoms (k_spmm4<4>)
int s4_prepare(rsrec_t* h) {
    if (!h->s4_attr) {
        HIPCK(h, hipFuncSetAttribute(reinterpret_cast<const void*>(k_spmm4<4>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)S4_LDS_BYTES));
        h->s4_attr = true;
    }
    return RSREC_OK;
}
int launch_spmm(rsrec_t* h, const SpmmDims& SD, const ChainView& CV, const DevProblem& P, int set, const double* in, double* out, dim3 grid_mf) {
    { const int rc = s4_prepare(h); if (rc) return rc; }
    // one group per workgroup at a time: 4x as many workgroups keep the same number of groups in flight per launch
    dim3 g4(std::min<unsigned>(grid_mf.x * 4, 1024), grid_mf.y);
    k_spmm4<4><<<g4, MF_WAVES * 64, S4_LDS_BYTES, h->stream>>>(SD, CV.order, CV.cum, P.nbr, P.iz, h->s4_op.frag_set(set), h->s4_op.meta_set(set), in, out);
    return RSREC_OK;
}

// One implementation for both kernel sets: L = LayoutCM with the VALU kernels, LayoutRM with the MFMA SpMM.
template <class L, bool MFMA>
int run_block_lanczos(rsrec_t* h, int nchains, int nseed, const int32_t* seed_atoms, const double* seed_coef, int lld, double* a_b, double* b2_b,
                      const double* rot = nullptr /*local-axis runs: complex (18,18,nchains), the spin-frame rotation of every chain*/) {
    const int kk = h->kk;
    const bool hoh = h->hoh != 0;
    const int nsteps = lld - 1;
    const int nlev = (hoh ? 2 * nsteps : nsteps) + 1;
    const size_t velems = (size_t)(kk + 1) * BLD;           // doubles per chain per vector (+1: the all-zero block)
    int nvec = MFMA ? 4 : (hoh ? 3 : 2);
    // Matrix-core set: k_spmm5 with every vector in the CI layout (option spmm5 = 2, the default since round 3: with the operator
    // streams assembled on the device a call no longer pays k_spmm4's host swizzle -- 3 ms per SCF iteration on the 18 operator classes
    // of B2FeCo, tools/time_set_hamiltonian.py -- and k_spmm5 is as fast on one chain).  spmm5 = 1: small launches of the plain operator
    // take the cooperative k_spmm4<4> on LayoutRM (kept as the cross-check of the parity tests); 0: k_spmm4 whenever it can
    const bool large = (long)std::min(nchains, 64) * (kk / GROUP + 1) >= 4096;
    const int ci = (MFMA && (hoh || rot || h->opt_spmm5 == 2 || (h->opt_spmm5 == 1 && large) || !spmm4_usable(h))) ? 1 : 0;   // vectors of this call are CI (else LayoutRM / LayoutCM)
    if (rot && !MFMA) return fail(h, RSREC_ERR_ARG, "local-axis recursion needs the matrix-core kernel set (option kernels = 0 or 2)");
    if (MFMA && !ci) { const int rc4 = ensure_s4(h); if (rc4) return rc4; }
    const int ntau = h->nmax + h->ntype;
    const Spmm5Operator& OP = rot ? h->s5_la : h->s5_op;
    const int la_fps = S5_HEAD_DOUBLES;
    if (rot) HIPCK(h, h->d_la_extra.reserve((size_t)std::min(nchains, 64) * ntau * la_fps * sizeof(double)));
    // the coefficients of ALL chains of the call stay on the device (resident input of rsrec_pack_diag / rsrec_block_ldos): reserved before
    // the batch is planned from the free memory
    h->res_kind = 0;
    HIPCK(h, h->d_coefA.reserve((size_t)nchains * lld * BLK * sizeof(double2)));
    HIPCK(h, h->d_coefB.reserve((size_t)nchains * lld * BLK * sizeof(double2)));
    // matrix-core set without hoh: vector 1 (pmn of the VALU set, h psi of the hoh passes) is not used by the u-scheme -- it is neither allocated
    // nor cleared (32 GB and a 5 ms memset per call for 64 sites of the 10^5-atom cell).  Of vector 2, H psi, only the zero block kk is
    // cleared: the SpMM writes every atom the passes behind it read, but those passes run padding entries of the lists as the zero block,
    // which has to BE zero in every vector (left uncleared on recycled device memory, dying chains of the fuzz seeds survived:
    // tests/test_gpu_breakdown.py).  46^3 x 64 sites: 2 387 -> 2 361-2 381 ms per step.
    // u_{n+1} goes into a THIRD u vector instead of over u_{n-1} (option orth_oop, round 4): k_mfma_orth3 then reads three vectors and writes a
    // fourth one -- 64 x 46^3: 25.7 -> 24.5 ms per saturated level (the in-place pass alternated 26.3 / 25.0 with the level's parity, the out-of-place
    // one cycles 25.0 / 25.6 / 22.9 with the three arrangements of its buffers), the step 2 387 -> 2 361 ms, 22^3 387 -> 379 ms; bitwise the same results.
    // The vector is number 1 without hoh (unused by the u-scheme otherwise) and number 4 with hoh (vector 1 holds h psi of the first pass).
    const bool oop = MFMA && h->opt_orth_oop != 0 && h->opt_orth3 != 2;
    if (oop && hoh) nvec = 5;
    const bool use_v1 = !MFMA || hoh || oop;
    BatchPlan bp;
    int rc = plan_batch(h, nchains, nvec - (use_v1 ? 0 : 1), velems / 2, bp);
    if (rc) return rc;
    const int B = bp.batch, nblk = bp.nblk;
    for (int v = 0; v < nvec; ++v) if (v != 1 || use_v1) HIPCK(h, h->d_vec[v].reserve((size_t)B * velems * sizeof(double)));
    const size_t gram_elems = (size_t)B * 256 * 1296;                                  // doubles: Gram partials of one kernel (<= 256 workgroups per chain)
    HIPCK(h, h->d_partial.reserve(std::max((size_t)B * std::max(nblk * 2, 256) * 2 * BLK * sizeof(double2), 2 * gram_elems * sizeof(double))));
    h->p2_slot = (size_t)B * 16 * 2 * 1296;
    HIPCK(h, h->d_partial2.reserve(2 * h->p2_slot * sizeof(double)));                  // second stage of the partial sums (presum), two slots: sized once, never grown mid-stream
    if (!h->side_stream) {
        HIPCK(h, hipStreamCreateWithFlags(&h->side_stream, hipStreamNonBlocking));
        HIPCK(h, hipEventCreateWithFlags(&h->ev_orth, hipEventDisableTiming));
        HIPCK(h, hipEventCreateWithFlags(&h->ev_bred, hipEventDisableTiming));
    }
    HIPCK(h, h->d_frags.reserve((size_t)B * 3 * 27 * 64 * sizeof(double)));
    HIPCK(h, h->d_bmats.reserve((size_t)B * 2 * BLK * sizeof(double2)));
    HIPCK(h, h->d_status.reserve(64));
    HIPCK(h, h->d_seed.reserve((size_t)B * nseed * 4));
    HIPCK(h, h->d_seedcoef.reserve((size_t)B * nseed * sizeof(double2)));
    HIPCK(h, hipMemsetAsync(h->d_status.p, 0, 64, h->stream));
    double* psi = h->d_vec[0].as<double>();
    double* pmn = h->d_vec[1].as<double>();
    double* hpsi = h->d_vec[2].as<double>();
    double* t2 = h->d_vec[3].as<double>();
    double2* partial = h->d_partial.as<double2>();
    double* gpartial = h->d_partial.as<double>();
    double* gpartial_b = gpartial + gram_elems;          // Gram partials of k_mfma_orth3 when their reduction runs on the side stream
    double* bfrags = h->d_frags.as<double>();            // [chain][3][27 * 64]: the three right-multiply tables of k_mfma_orth3
    const DevProblem P = make_problem(h);
    HIPCK(h, hipFuncSetAttribute(reinterpret_cast<const void*>(k_orth<L>), hipFuncAttributeMaxDynamicSharedMemorySize, TILE_ATOMS * BLK * (int)sizeof(double2)));
    const size_t cstride = (size_t)lld * BLK;
    const size_t orth_lds = TILE_ATOMS * BLK * sizeof(double2);
    hipEvent_t ev_begin = next_event(h);
    std::vector<std::pair<hipEvent_t, hipEvent_t>> hop_ev;
    h->hop_fuses_a = MFMA ? 0 : 1;

    for (int c0 = 0; c0 < nchains; c0 += B) {
        const int nb = std::min(B, nchains - c0);
        double2* dA = h->d_coefA.as<double2>() + (size_t)c0 * cstride;       // this batch's slice of the resident coefficients
        double2* dB = h->d_coefB.as<double2>() + (size_t)c0 * cstride;
        const auto th0 = std::chrono::steady_clock::now();
        std::vector<int> seeds0((size_t)nb * nseed);
        std::vector<double> coef((size_t)nb * nseed * 2);
        for (int q = 0; q < nb * nseed; ++q) {
            seeds0[q] = seed_atoms[(size_t)c0 * nseed + q] - 1;
            coef[2 * q] = seed_coef ? seed_coef[2 * ((size_t)c0 * nseed + q)] : 1.0;
            coef[2 * q + 1] = seed_coef ? seed_coef[2 * ((size_t)c0 * nseed + q) + 1] : 0.0;
        }
        int ostride = kk;
        rc = upload_regions(h, seeds0.data(), nb, nseed, nlev, nsteps, hoh, MFMA, ostride, h->n_atom_steps, h->n_block_mult);
        if (rc) return rc;
        h->n_req_flop += required_hop_flops(h, OP);
        XFER(xfer_h2d(h, h->d_seed.p, seeds0.data(), seeds0.size() * 4));
        XFER(xfer_h2d(h, h->d_seedcoef.p, coef.data(), coef.size() * 8));
        const double* la_extra = nullptr;
        if (rot) {
            // per-chain on-site term of the local-axis operator in the GLOBAL frame: (e_nu +) R l.s R^H  (see rsrec_block_lanczos_local_axis)
            std::vector<double> fr((size_t)nb * ntau * la_fps), E(2 * BLK), T(2 * BLK);
            for (int c = 0; c < nb; ++c) {
                const double* R = rot + 2 * (size_t)BLK * (c0 + c);
                for (int tau = 0; tau < ntau; ++tau) {
                    const int ty = tau < h->nmax ? h->iz0[tau] : tau - h->nmax;
                    const double* ls = h->host_lsham.data() + 2 * (size_t)BLK * ty;
                    for (int j = 0; j < NB; ++j)                    // T = l.s R^H
                        for (int i = 0; i < NB; ++i) {
                            double sr = 0.0, si = 0.0;
                            for (int k = 0; k < NB; ++k) {
                                const double ar = ls[2 * (i + NB * k)], ai = ls[2 * (i + NB * k) + 1], br = R[2 * (j + NB * k)], bi = -R[2 * (j + NB * k) + 1];
                                sr += ar * br - ai * bi; si += ar * bi + ai * br;
                            }
                            T[2 * (i + NB * j)] = sr; T[2 * (i + NB * j) + 1] = si;
                        }
                    for (int j = 0; j < NB; ++j)                    // E = R T
                        for (int i = 0; i < NB; ++i) {
                            double sr = 0.0, si = 0.0;
                            for (int k = 0; k < NB; ++k) {
                                const double ar = R[2 * (i + NB * k)], ai = R[2 * (i + NB * k) + 1], br = T[2 * (k + NB * j)], bi = T[2 * (k + NB * j) + 1];
                                sr += ar * br - ai * bi; si += ar * bi + ai * br;
                            }
                            E[2 * (i + NB * j)] = sr; E[2 * (i + NB * j) + 1] = si;
                        }
                    if (hoh) for (int e = 0; e < 2 * BLK; ++e) E[e] += h->host_enim[2 * (size_t)BLK * ty + e];
                    OP.emit_head(hoh ? 1 : 0, tau, E.data(), fr.data() + ((size_t)c * ntau + tau) * la_fps);
                }
            }
            XFER(xfer_h2d(h, h->d_la_extra.p, fr.data(), fr.size() * sizeof(double)));
            la_extra = h->d_la_extra.as<double>();
        }
        HIPCK(h, hipStreamSynchronize(h->stream));
        h->t_host_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - th0).count();

        ChainView CV;
        CV.order = h->cur_order; CV.cum = h->cur_cum; CV.obase = h->cur_cum + (size_t)h->cur_nrows * nlev; CV.nlev = nlev; CV.vstride = velems; CV.cpo = 1; CV.ostride = ostride;
        ChainView CVp = CV;                                   // the streaming passes walk the level-major lists (upload_regions)
        CVp.cum = h->cur_cum + (size_t)2 * h->cur_nrows * nlev; CVp.obase = h->cur_cum + (size_t)3 * h->cur_nrows * nlev;
        // Everything from here to the coefficients' download is stream work only (kernels, memsets, cross-stream events): for small
        // batches it is captured ONCE as a HIP graph and replayed by every later call with the same lattice, seeds, depth and buffers
        // (each SCF iteration of the reference: recur_b on the same <= 4 sites) -- 49 levels x 6-8 dependent launches otherwise cost
        // more host time than device time (13 ms for one site of the 22^3 cell, two thirds of it launch latency).
        auto enqueue_levels = [&]() -> int {
            for (int v = 0; v < nvec; ++v) {
                if (v == 1 && !use_v1) continue;
                if (v == 2 && MFMA && !hoh) {
                    // H psi: only its zero block (one 2-D memset over the chains); every other block is written by the SpMM before it is read
                    HIPCK(h, hipMemset2DAsync(static_cast<char*>(h->d_vec[v].p) + (size_t)kk * BLD * sizeof(double), velems * sizeof(double), 0, BLD * sizeof(double), (size_t)nb, h->stream));
                    continue;
                }
                HIPCK(h, hipMemsetAsync(h->d_vec[v].p, 0, (size_t)nb * velems * sizeof(double), h->stream));
            }
            HIPCK(h, hipMemsetAsync(dA, 0, (size_t)nb * cstride * sizeof(double2), h->stream));
            HIPCK(h, hipMemsetAsync(dB, 0, (size_t)nb * cstride * sizeof(double2), h->stream));
            if (MFMA) HIPCK(h, hipMemsetAsync(bfrags, 0, (size_t)nb * 3 * 27 * 64 * sizeof(double), h->stream));
            psi = h->d_vec[0].as<double>(); t2 = h->d_vec[3].as<double>();   // (swapped every level)
            double* t3 = oop ? h->d_vec[hoh ? 4 : 1].as<double>() : nullptr;
            if (ci) k_seed<LayoutCI><<<nb, 64, 0, h->stream>>>(psi, velems, h->d_seed.as<int>(), h->d_seedcoef.as<double2>(), nseed);
            else k_seed<L><<<nb, 64, 0, h->stream>>>(psi, velems, h->d_seed.as<int>(), h->d_seedcoef.as<double2>(), nseed);
            k_set_identity<<<nb, 256, 0, h->stream>>>(dB, cstride);                                  // b2temp_b(:,:,1) = I  (:1837)
            if (MFMA) k_uscheme_init<<<nb, 256, 0, h->stream>>>(h->d_bmats.as<double2>(), bfrags, ci);
            const dim3 grid(nblk, nb);
            const dim3 grid_mf(std::max(1, std::min(mfma_workgroups_per_chain(h, B), (ostride / GROUP + MF_WAVES - 1) / MF_WAVES)), nb);
            // u-scheme: H u_{n+1} does not need B_{n+1}, so the reduction of sum u_{n+1}^H u_{n+1} and its 18x18 eigen-solve (one
            // workgroup per chain, 80 us) leave the critical path: they run on the side stream while the main stream already applies H.
            // The main stream waits for them before k_reduce_a_u of the next level (first consumer of Binv_{n+1}).
            const bool side = h->opt_side && h->side_stream;
            double* gp_b = side ? gpartial_b : gpartial;
            bool b_pending = false;
            auto wait_b_level = [&]() -> int {
                if (b_pending) { HIPCK(h, hipStreamWaitEvent(h->stream, h->ev_bred, 0)); b_pending = false; }
                return RSREC_OK;
            };
            auto reduce_b_level = [&](int nwg, int ll) -> int {
                hipStream_t st = h->stream;
                if (side) {
                    HIPCK(h, hipEventRecord(h->ev_orth, h->stream));
                    HIPCK(h, hipStreamWaitEvent(h->side_stream, h->ev_orth, 0));
                    st = h->side_stream;
                }
                int n2 = nwg;
                const double* p2 = presum(h, gp_b, nb, n2, 1296, st, side ? 1 : 0);
                k_reduce_b_u<<<nb, 1024, 0, st>>>(p2, n2, dB + (size_t)(ll + 1) * BLK, cstride, h->d_bmats.as<double2>(), bfrags, h->d_status.as<int>(), ci);
                if (side) { HIPCK(h, hipEventRecord(h->ev_bred, h->side_stream)); b_pending = true; }
                return RSREC_OK;
            };
            for (int ll = 0; ll < nsteps; ++ll) {
                const int lv_final = hoh ? 2 * ll + 2 : ll + 1;
                const double* tvec = nullptr;                      // H psi when it is held in a vector of its own
                hipEvent_t e0 = next_event(h);
                hipEvent_t e1 = nullptr;
                ApplyArgs G{};
                G.partial = partial;
                if (MFMA) {
                    // matrix-core kernel set, un-normalised vectors (kernels_uscheme.hpp): psi = u_n, t2 = u_{n-1}; u_{n+1} overwrites u_{n-1}
                    SpmmDims SD{kk, P.nslots, P.nmax, nlev, 1, ostride, hoh ? 2 * ll + 1 : lv_final, velems, CV.obase, nb};
                    if (!hoh) {
                        if (ci && rot) launch_s5<true>(h, s5_grid(h, grid_mf, lv_final), SD, CV.order, CV.cum, P.iz, OP, 0, psi, hpsi, psi, la_extra, ntau);
                        else if (ci) launch_s5<false>(h, s5_grid(h, grid_mf, lv_final), SD, CV.order, CV.cum, P.iz, OP, 0, psi, hpsi);
                        else { rc = launch_spmm(h, SD, CV, P, 0, psi, hpsi, grid_mf); if (rc) return rc; }
                    } else {
                        // H = h - (h o) h + e_nu + l.s in two passes of k_spmm5: h psi, then the rest with psi as second input (extra on-site slot)
                        double* hps = pmn;                   // (the pmn buffer is free in the u-scheme)
                        launch_s5<false>(h, s5_grid(h, grid_mf, 2 * ll + 1), SD, CV.order, CV.cum, P.iz, OP, 0, psi, hps);
                        SD.level = lv_final;
                        launch_s5<true>(h, s5_grid(h, grid_mf, lv_final), SD, CV.order, CV.cum, P.iz, OP, 1, hps, hpsi, psi, la_extra, ntau);
                    }
                    e1 = next_event(h);
                    const dim3 gl = level_grid(h, grid_mf, lv_final);
                    k_mfma_adot<<<gl, MF_WAVES * 64, 0, h->stream>>>(CVp, lv_final, kk, psi, hpsi, gpartial);
                    { int n2 = gl.x; const double* p2 = presum(h, gpartial, nb, n2, 1296);
                      rc = wait_b_level(); if (rc) return rc;
                      k_reduce_a_u<<<nb, 1024, 0, h->stream>>>(p2, n2, dA + (size_t)ll * BLK, cstride, h->d_bmats.as<double2>(), bfrags, ci); }
                    if (h->opt_orth3 == 2) k_mfma_orth3w<<<gl, MF_WAVES * 64, 0, h->stream>>>(CVp, lv_final, kk, hpsi, psi, t2, bfrags, gp_b);
                else k_mfma_orth3<<<gl, MF_WAVES * 64, 0, h->stream>>>(CVp, lv_final, kk, hpsi, psi, t2, bfrags, gp_b, t3);
                    rc = reduce_b_level(gl.x, ll); if (rc) return rc;
                    if (oop) { double* f = t2; t2 = psi; psi = t3; t3 = f; }      // u_{n+1} is in t3; the vector of u_{n-1} is free
                    else std::swap(psi, t2);
                    hop_ev.emplace_back(e0, e1);
                    h->n_hop_launch += hoh ? 2 : 1;
                    continue;
                }
                // FP64 VALU kernel set: the reference's literal order on the reference's layout
                if (!hoh) {
                    G.in = psi; G.v0 = psi; G.out = pmn; G.level = lv_final;
                    k_apply<AM_LANCZOS, L><<<grid, NTHREADS, 0, h->stream>>>(P, CV, G);
                    e1 = next_event(h);
                } else {
                    G.in = psi; G.out = hpsi; G.level = 2 * ll + 1;
                    k_apply<AM_STORE, L><<<grid, NTHREADS, 0, h->stream>>>(P, CV, G);
                    G.in = hpsi; G.v1 = hpsi; G.cur = psi; G.v0 = psi; G.out = pmn; G.level = lv_final;
                    k_apply<AM_HOH_LANCZOS, L><<<grid, NTHREADS, 0, h->stream>>>(P, CV, G);
                    e1 = next_event(h);
                }
                hop_ev.emplace_back(e0, e1);                       // [e0,e1] brackets exactly the H|psi> kernel(s) of this step
                h->n_hop_launch += hoh ? 2 : 1;
                k_reduce_a<<<nb, 1024, 0, h->stream>>>(partial, nblk, dA + (size_t)ll * BLK, cstride);
                k_orth<L><<<grid, NTHREADS, orth_lds, h->stream>>>(CV, lv_final, psi, pmn, tvec, dA + (size_t)ll * BLK, cstride, partial);
                k_reduce_b_eig<<<nb, 1024, 0, h->stream>>>(partial, nblk, dB + (size_t)(ll + 1) * BLK, cstride, h->d_bmats.as<double2>(), h->d_status.as<int>());
                k_update<L><<<grid, NTHREADS, 0, h->stream>>>(CV, lv_final, psi, pmn, h->d_bmats.as<double2>());
            }
            HIPCK(h, hipGetLastError());
            rc = wait_b_level(); if (rc) return rc;
            return RSREC_OK;
        };
        const bool use_graph = MFMA && nchains <= B && ((h->opt_graph == 1 && nchains <= 8) || h->opt_graph >= 2) && getenv("RSREC_NO_GRAPH") == nullptr;
        if (!use_graph) { rc = enqueue_levels(); if (rc) return rc; }
        else {
            if (ci) HIPCK(h, h->d_s5queue.reserve((size_t)nb * 16 * sizeof(int)));
            // what the nodes hold BY VALUE: every pointer and dimension a kernel argument is made of
            std::vector<uintptr_t> key = {(uintptr_t)1 /*block Lanczos*/, (uintptr_t)nb, (uintptr_t)lld, (uintptr_t)nseed, (uintptr_t)hoh, (uintptr_t)ci, (uintptr_t)(rot != nullptr), (uintptr_t)kk,
                                          (uintptr_t)h->cur_order, (uintptr_t)h->cur_cum, (uintptr_t)ostride, (uintptr_t)OP.d_frag, (uintptr_t)OP.d_meta, (uintptr_t)OP.ntr,
                                          (uintptr_t)h->s4_op.frag_set(0), (uintptr_t)h->s4_op.meta_set(0), (uintptr_t)h->d_nbr.p, (uintptr_t)h->d_nbr5.p, (uintptr_t)h->d_iz.p,
                                          (uintptr_t)h->d_partial.p, (uintptr_t)h->d_partial2.p, (uintptr_t)h->d_frags.p, (uintptr_t)dA, (uintptr_t)dB, (uintptr_t)h->d_bmats.p,
                                          (uintptr_t)h->d_status.p, (uintptr_t)h->d_seed.p, (uintptr_t)h->d_seedcoef.p, (uintptr_t)h->d_la_extra.p, (uintptr_t)h->d_s5queue.p,
                                          (uintptr_t)h->opt_side, (uintptr_t)h->opt_orth3, (uintptr_t)h->opt_orth_oop, (uintptr_t)h->opt_nblk, (uintptr_t)h->opt_spmm5, (uintptr_t)h->opt_s5_lds, (uintptr_t)h->opt_s5_queue, (uintptr_t)h->opt_s5_run_min, (uintptr_t)h->cur_entry,
                                          (uintptr_t)h->opt_chain_fold, (uintptr_t)h->opt_s5_cap, (uintptr_t)h->p2_slot, (uintptr_t)OP.single_class(0), (uintptr_t)OP.spin_mixing, (uintptr_t)h->opt_s5_octet, (uintptr_t)h->opt_s5_spin_xcd,
                                          (uintptr_t)h->lattice_epoch, (uintptr_t)OP.sched_epoch, (uintptr_t)h->nslots, (uintptr_t)h->nmax, (uintptr_t)h->ntype, (uintptr_t)h->hslots,
                                          (uintptr_t)h->opt_s5_waves, (uintptr_t)h->opt_s5_split, (uintptr_t)h->opt_batch, (uintptr_t)B, (uintptr_t)h->opt_kernels, (uintptr_t)h->n_cu};
            for (int v = 0; v < nvec; ++v) key.push_back((uintptr_t)h->d_vec[v].p);
            if (!h->graph_exec || key != h->graph_key) {
                if (h->graph_exec) { (void)hipGraphExecDestroy(h->graph_exec); h->graph_exec = nullptr; }
                h->graph_key.clear();
                s5_prepare(h);
                rc = s4_prepare(h); if (rc) return rc;
                HIPCK(h, hipStreamSynchronize(h->stream));
                HIPCK(h, hipStreamBeginCapture(h->stream, hipStreamCaptureModeRelaxed));
                h->capturing = true;
                rc = enqueue_levels();
                h->capturing = false;
                hipGraph_t graph = nullptr;
                const hipError_t ec = hipStreamEndCapture(h->stream, &graph);
                if (rc) { if (graph) (void)hipGraphDestroy(graph); return rc; }
                if (ec != hipSuccess || !graph) return fail(h, RSREC_ERR_DEVICE, "hipStreamEndCapture failed: %s", hipGetErrorString(ec));
                const hipError_t ei = hipGraphInstantiate(&h->graph_exec, graph, nullptr, nullptr, 0);
                (void)hipGraphDestroy(graph);
                if (ei != hipSuccess) { h->graph_exec = nullptr; return fail(h, RSREC_ERR_DEVICE, "hipGraphInstantiate failed: %s", hipGetErrorString(ei)); }
                h->graph_key = key;
            }
            HIPCK(h, hipGraphLaunch(h->graph_exec, h->stream));
            h->n_hop_launch += (double)nsteps * (hoh ? 2 : 1);
        }
        XFER(xfer_d2h(h, a_b + (size_t)c0 * cstride * 2, dA, (size_t)nb * cstride * sizeof(double2)));
        XFER(xfer_d2h(h, b2_b + (size_t)c0 * cstride * 2, dB, (size_t)nb * cstride * sizeof(double2)));
        HIPCK(h, hipStreamSynchronize(h->stream));
    }
    hipEvent_t ev_end = next_event(h);
    HIPCK(h, hipStreamSynchronize(h->stream));
    h->t_total_ms = ev_ms(ev_begin, ev_end);
    for (auto& pr : hop_ev) h->t_hop_ms += ev_ms(pr.first, pr.second);
    h->t_rest_ms = h->t_total_ms - h->t_hop_ms;
    int status = 0;
    XFER(xfer_d2h(h, &status, h->d_status.p, 4));
    if (status & 1) return fail(h, RSREC_ERR_EIG, "Diagonalization error (18x18 Jacobi did not converge)");
    h->res_kind = 1; h->res_n = nchains; h->res_lld = lld; h->res_sqrt = 0;
    return RSREC_OK;
}

}  // namespace

extern "C" int rsrec_block_lanczos_seeded(rsrec_t* h, int nchains, int nseed, const int32_t* seed_atoms, const double* seed_coef, int lld,
                                          double* a_b, double* b2_b) {
    int rc = check_ready(h, "rsrec_block_lanczos");
    if (rc) return rc;
    if (nchains < 0 || nseed < 1 || lld < 1 || !a_b || !b2_b || (nchains > 0 && !seed_atoms)) return fail(h, RSREC_ERR_ARG, "rsrec_block_lanczos: bad argument");
    for (int q = 0; q < nchains * nseed; ++q)
        if (seed_atoms[q] < 1 || seed_atoms[q] > h->kk) return fail(h, RSREC_ERR_ARG, "rsrec_block_lanczos: seed atom %d outside 1..%d", seed_atoms[q], h->kk);
    HIPCK(h, hipSetDevice(h->device));
    reset_timing(h);
    if (nchains == 0) return RSREC_OK;
    // kernels: 0 = auto, 1 = FP64 VALU kernel set, 2 = matrix-core set.  The matrix-core SpMM tables exist for lattices with up to
    // S4_MAXSLOTS - 1 neighbour slots; beyond that the VALU set (any stencil) runs.
    const bool use_mfma = (h->opt_kernels != 1) && h->s5_built;
    if (use_mfma) return run_block_lanczos<LayoutRM, true>(h, nchains, nseed, seed_atoms, seed_coef, lld, a_b, b2_b);
    return run_block_lanczos<LayoutCM, false>(h, nchains, nseed, seed_atoms, seed_coef, lld, a_b, b2_b);
}

extern "C" int rsrec_block_lanczos(rsrec_t* h, int nsites, const int32_t* seed_atoms, int lld, double* a_b, double* b2_b) {
    return rsrec_block_lanczos_seeded(h, nsites, 1, seed_atoms, nullptr, lld, a_b, b2_b);
}

namespace {

// Operator tables of a local-axis run: the blocks as set (GLOBAL spin frame), on-site slot WITHOUT l.s; the l.s term (with e_nu for
// hoh) enters through the extra on-site slot whose fragments come per chain.  The placeholder only makes the slot appear in the
// schedule with both spin parts.
int build_local_axis_operator(rsrec_t* h) {
    if (h->s5_la_ok) return RSREC_OK;
    const int ntau = h->nmax + h->ntype, nfs = h->nslots + 1, nset = h->hoh ? 2 : 1;
    const size_t B = 2 * (size_t)BLK;
    std::vector<const double*> blk((size_t)nset * ntau * nfs, nullptr);
    std::vector<double> dense(B, 1.0), neg((size_t)ntau * h->nslots * B, 0.0);
    for (int tau = 0; tau < ntau; ++tau) {
        for (int s = 0; s < h->nslots; ++s) {
            const double* src = tau < h->nmax ? h->host_hall.data() + B * (s + (size_t)h->hslots * tau) : h->host_ee.data() + B * (s + (size_t)h->hslots * (tau - h->nmax));
            blk[((size_t)0 * ntau + tau) * nfs + s] = src;
            if (nset > 1) {
                const double* so = tau < h->nmax ? h->host_hallo.data() + B * (s + (size_t)h->hslots * tau) : h->host_eeo.data() + B * (s + (size_t)h->hslots * (tau - h->nmax));
                double* d = neg.data() + B * (s + (size_t)h->nslots * tau);
                for (size_t e = 0; e < B; ++e) d[e] = -so[e];
                if (s == 0) for (int q = 0; q < NB; ++q) d[2 * (q + NB * q)] += 1.0;
                blk[((size_t)1 * ntau + tau) * nfs + s] = d;
            }
        }
        blk[((size_t)(nset - 1) * ntau + tau) * nfs + h->nslots] = dense.data();
    }
    const char* msg = h->s5_la.build_custom(h->nslots, ntau, nset, blk);
    if (msg) return fail(h, RSREC_ERR_DEVICE, "rsrec_block_lanczos_local_axis: %s", msg);
    h->s5_la_ok = 1;
    return RSREC_OK;
}

}  // namespace

// recur_b with hamiltonian%local_axis = T (recursion.f90:1830-1832), all sites in one batched call.  The reference rotates every
// block into the spin frame of site i's moment before that site's chain, H'_i = R_i^H H R_i blockwise -- except the on-site l.s term,
// which rotate_to_local_axis (hamiltonian.f90:2442-2465) leaves alone.  With phi = R_i psi (blocks multiplied from the left) the
// chain of H'_i from the seed 1 is the chain of  H''_i = H + onsite(R_i l.s R_i^H - l.s)  from the seed R_i, i.e. from the seed 1
// followed by a right-multiplication with the unitary R_i:   A'_n = R_i^H A''_n R_i,  B'^2_n = R_i^H B''^2_n R_i.
// So every chain runs on the SAME global-frame blocks and differs only in its on-site term (per-chain extra slot of k_spmm5); the
// 18x18 outputs are conjugated on the host.  Checked against the compiled reference on four sites with four moment directions.
extern "C" int rsrec_block_lanczos_local_axis(rsrec_t* h, int nsites, const int32_t* seed_atoms, const double* rot, int lld, double* a_b, double* b2_b) {
    int rc = check_ready(h, "rsrec_block_lanczos_local_axis");
    if (rc) return rc;
    if (nsites < 0 || lld < 1 || !a_b || !b2_b || (nsites > 0 && (!seed_atoms || !rot))) return fail(h, RSREC_ERR_ARG, "rsrec_block_lanczos_local_axis: bad argument");
    for (int q = 0; q < nsites; ++q)
        if (seed_atoms[q] < 1 || seed_atoms[q] > h->kk) return fail(h, RSREC_ERR_ARG, "rsrec_block_lanczos_local_axis: seed atom %d outside 1..%d", seed_atoms[q], h->kk);
    if (!h->s5_built) return fail(h, RSREC_ERR_ARG, "rsrec_block_lanczos_local_axis: lattice has too many neighbour slots for the SpMM kernel");
    HIPCK(h, hipSetDevice(h->device));
    reset_timing(h);
    if (nsites == 0) return RSREC_OK;
    rc = build_local_axis_operator(h); if (rc) return rc;
    rc = run_block_lanczos<LayoutRM, true>(h, nsites, 1, seed_atoms, nullptr, lld, a_b, b2_b, rot);
    if (rc) return rc;
    // A' = R^H A R, B'^2 = R^H B^2 R; b2_b(:,:,1) = I and a_b(:,:,lld) = 0 stay exact (recursion.f90:1836-1837)
    std::vector<double> T(2 * BLK);
    auto conj_sim = [&](double* M, const double* R) {
        for (int j = 0; j < NB; ++j)
            for (int i = 0; i < NB; ++i) {
                double sr = 0.0, si = 0.0;
                for (int k = 0; k < NB; ++k) {
                    const double ar = M[2 * (i + NB * k)], ai = M[2 * (i + NB * k) + 1], br = R[2 * (k + NB * j)], bi = R[2 * (k + NB * j) + 1];
                    sr += ar * br - ai * bi; si += ar * bi + ai * br;
                }
                T[2 * (i + NB * j)] = sr; T[2 * (i + NB * j) + 1] = si;
            }
        for (int j = 0; j < NB; ++j)
            for (int i = 0; i < NB; ++i) {
                double sr = 0.0, si = 0.0;
                for (int k = 0; k < NB; ++k) {
                    const double ar = R[2 * (k + NB * i)], ai = -R[2 * (k + NB * i) + 1], br = T[2 * (k + NB * j)], bi = T[2 * (k + NB * j) + 1];
                    sr += ar * br - ai * bi; si += ar * bi + ai * br;
                }
                M[2 * (i + NB * j)] = sr; M[2 * (i + NB * j) + 1] = si;
            }
    };
    for (int s = 0; s < nsites; ++s) {
        const double* R = rot + 2 * (size_t)BLK * s;
        for (int ll = 0; ll < lld; ++ll) {
            if (ll < lld - 1) conj_sim(a_b + 2 * (size_t)BLK * ((size_t)s * lld + ll), R);
            if (ll > 0) conj_sim(b2_b + 2 * (size_t)BLK * ((size_t)s * lld + ll), R);
        }
    }
    h->res_kind = 0;          // the resident coefficients are the un-rotated ones: not valid input for the LDOS stage
    return RSREC_OK;
}

namespace {

// true if p is device memory of this process (torch / hipMalloc allocations): outputs may then stay on the GPU.
// A Python process that imports torch holds TWO HIP runtimes (torch bundles its own libamdhip64; this library links the system one)
// on top of ONE shared ROCr/HSA runtime and one GPU address space: a tensor's address is valid in our kernels, but our HIP runtime
// has never heard of it.  So the question is put to ROCr (hsa_amd_pointer_info), which knows every allocation of the process.
bool is_device_ptr(const void* p) {
    hipPointerAttribute_t at;
    if (hipPointerGetAttributes(&at, p) == hipSuccess) return at.type == hipMemoryTypeDevice;
    (void)hipGetLastError();
    hsa_amd_pointer_info_t info;
    memset(&info, 0, sizeof info);
    info.size = sizeof info;
    if (hsa_amd_pointer_info(const_cast<void*>(p), &info, nullptr, nullptr, nullptr) != HSA_STATUS_SUCCESS) return false;
    if (info.type != HSA_EXT_POINTER_TYPE_HSA) return false;                          // unknown = plain pageable host memory
    hsa_device_type_t dt = HSA_DEVICE_TYPE_CPU;
    if (hsa_agent_get_info(info.agentOwner, HSA_AGENT_INFO_DEVICE, &dt) != HSA_STATUS_SUCCESS) return false;
    return dt == HSA_DEVICE_TYPE_GPU;
}

// a(ll, l, site) = Re a_b(l, l, ll, site), b2 likewise (recursion.f90:1850-1851), written into zero-padded images over all sites
__global__ void k_pack_diag(const double2* __restrict__ A, const double2* __restrict__ B, int lld, int n, int off, int ntot,
                            double* __restrict__ a_img, double* __restrict__ b_img) {
    const size_t total = (size_t)lld * NB * ntot;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int ll = (int)(e % lld), l = (int)((e / lld) % NB), s = (int)(e / ((size_t)lld * NB)) - off;
        double a = 0.0, b = 0.0;
        if (s >= 0 && s < n) {
            const size_t q = ((size_t)s * lld + ll) * BLK + (size_t)l * (NB + 1);
            a = A[q].x; b = B[q].x;
        }
        a_img[e] = a; b_img[e] = b;
    }
}

}  // namespace

// The per-site result the ranks exchange after the recursion, packed on the device: bands.f90:271-274 gathers per-site arrays with
// MPI_ALLREDUCE(MPI_SUM) on zero-padded images; this writes this rank's part of such an image (every other site zero).
extern "C" int rsrec_pack_diag(rsrec_t* h, int site_offset, int nsites_total, double* a_img, double* b2_img) {
    if (!h || !a_img || !b2_img || site_offset < 0) return fail(h, RSREC_ERR_ARG, "rsrec_pack_diag: bad argument");
    if (h->res_kind != 1) return fail(h, RSREC_ERR_ARG, "rsrec_pack_diag: no block-Lanczos coefficients resident (call rsrec_block_lanczos first)");
    if (site_offset + h->res_n > nsites_total) return fail(h, RSREC_ERR_ARG, "rsrec_pack_diag: sites %d..%d outside 1..%d", site_offset + 1, site_offset + h->res_n, nsites_total);
    HIPCK(h, hipSetDevice(h->device));
    const size_t n = (size_t)h->res_lld * NB * nsites_total;
    const bool dev = is_device_ptr(a_img) && is_device_ptr(b2_img);
    double *da = a_img, *db = b2_img;
    if (!dev) { HIPCK(h, h->d_scal.reserve(2 * n * sizeof(double))); da = h->d_scal.as<double>(); db = da + n; }
    const int blocks = (int)std::min<size_t>(1024, (n + 255) / 256);
    k_pack_diag<<<blocks, 256, 0, h->stream>>>(h->d_coefA.as<double2>(), h->d_coefB.as<double2>(), h->res_lld, h->res_n, site_offset, nsites_total, da, db);
    HIPCK(h, hipGetLastError());
    if (!dev) { XFER(xfer_d2h(h, a_img, da, n * sizeof(double))); XFER(xfer_d2h(h, b2_img, db, n * sizeof(double))); }
    HIPCK(h, hipStreamSynchronize(h->stream));
    return RSREC_OK;
}

namespace {

// mu_n(18,18,nmom,site) of this rank's sites inside a zero image over all sites
__global__ void k_pack_moments(const double2* __restrict__ mu, size_t per_site, int n, int off, int ntot, double2* __restrict__ img) {
    const size_t total = per_site * (size_t)ntot;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int s = (int)(e / per_site) - off;
        img[e] = (s >= 0 && s < n) ? mu[(size_t)s * per_site + e % per_site] : make_double2(0.0, 0.0);
    }
}

}  // namespace

// The Chebyshev counterpart of rsrec_pack_diag: the moments mu_n(18,18,2 lld + 2,site) of the last rsrec_chebyshev call, as they lie on
// the device, written into a zero-padded image over all sites (the buffer a sum all-reduce turns into the all-gather the reference's
// commented-out MPI_Allgather of recursion.f90:1790-1793 describes).  mu_img: device or host memory, complex (18,18,2 lld + 2,nsites_total).
extern "C" int rsrec_pack_moments(rsrec_t* h, int site_offset, int nsites_total, double* mu_img) {
    if (!h || !mu_img || site_offset < 0) return fail(h, RSREC_ERR_ARG, "rsrec_pack_moments: bad argument");
    if (h->res_kind != 2) return fail(h, RSREC_ERR_ARG, "rsrec_pack_moments: no Chebyshev moments resident (call rsrec_chebyshev first)");
    if (site_offset + h->res_n > nsites_total) return fail(h, RSREC_ERR_ARG, "rsrec_pack_moments: sites %d..%d outside 1..%d", site_offset + 1, site_offset + h->res_n, nsites_total);
    HIPCK(h, hipSetDevice(h->device));
    const size_t per_site = (size_t)(2 * h->res_lld + 2) * BLK, n = per_site * nsites_total;
    const bool dev = is_device_ptr(mu_img);
    double2* out = reinterpret_cast<double2*>(mu_img);
    if (!dev) { HIPCK(h, h->d_scal.reserve(n * sizeof(double2))); out = h->d_scal.as<double2>(); }
    k_pack_moments<<<(int)std::min<size_t>(2048, (n + 255) / 256), 256, 0, h->stream>>>(h->d_mu.as<double2>(), per_site, h->res_n, site_offset, nsites_total, out);
    HIPCK(h, hipGetLastError());
    if (!dev) XFER(xfer_d2h(h, mu_img, out, n * sizeof(double2)));
    HIPCK(h, hipStreamSynchronize(h->stream));
    return RSREC_OK;
}

// ------------------------------------------------------------------------------------------------------------------
// Library-level communicator.  The reference's only exchange on this path is MPI_ALLREDUCE(MPI_IN_PLACE, ..., MPI_SUM) on zero-padded
// per-site arrays (bands.f90:271-274; mpi.f90:32-58 gives every rank whole sites).  On a node of MI355X this is one RCCL all-reduce
// over xGMI on a device image; RCCL is bound with dlopen when a communicator is asked for -- the recursion itself does not depend on
// it -- so a Fortran (or any) host needs neither MPI nor torch for it: the 128-byte id travels by whatever the host has (MPI_Bcast, a
// file on a shared directory: rsrec_comm_init_file).
namespace {

struct NcclId { char internal[RSREC_COMM_ID_BYTES]; };
struct RcclApi {
    void* lib = nullptr;
    int (*get_unique_id)(NcclId*) = nullptr;
    int (*comm_init_rank)(void**, int, NcclId, int) = nullptr;
    int (*all_reduce)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
    int (*comm_destroy)(void*) = nullptr;
    const char* (*get_error_string)(int) = nullptr;
};
RcclApi g_rccl;

const char* rccl_ready() {
    if (g_rccl.lib) return nullptr;
    void* lib = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
    if (!lib) lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!lib) lib = dlopen("/opt/rocm/lib/librccl.so", RTLD_NOW | RTLD_LOCAL);
    if (!lib) return "RCCL (librccl.so) not found";
    g_rccl.get_unique_id = reinterpret_cast<decltype(g_rccl.get_unique_id)>(dlsym(lib, "ncclGetUniqueId"));
    g_rccl.comm_init_rank = reinterpret_cast<decltype(g_rccl.comm_init_rank)>(dlsym(lib, "ncclCommInitRank"));
    g_rccl.all_reduce = reinterpret_cast<decltype(g_rccl.all_reduce)>(dlsym(lib, "ncclAllReduce"));
    g_rccl.comm_destroy = reinterpret_cast<decltype(g_rccl.comm_destroy)>(dlsym(lib, "ncclCommDestroy"));
    g_rccl.get_error_string = reinterpret_cast<decltype(g_rccl.get_error_string)>(dlsym(lib, "ncclGetErrorString"));
    if (!g_rccl.get_unique_id || !g_rccl.comm_init_rank || !g_rccl.all_reduce || !g_rccl.comm_destroy) return "RCCL symbols missing";
    g_rccl.lib = lib;
    return nullptr;
}
const char* rccl_err(int rc) { return g_rccl.get_error_string ? g_rccl.get_error_string(rc) : "RCCL error"; }

}  // namespace

// id: RSREC_COMM_ID_BYTES bytes; created by ONE rank and given to all (ncclGetUniqueId).
extern "C" int rsrec_comm_unique_id(char* id) {
    if (!id) return RSREC_ERR_ARG;
    if (rccl_ready()) return RSREC_ERR_DEVICE;
    NcclId u;
    if (g_rccl.get_unique_id(&u) != 0) return RSREC_ERR_DEVICE;
    memcpy(id, u.internal, RSREC_COMM_ID_BYTES);
    return RSREC_OK;
}

extern "C" int rsrec_comm_destroy(rsrec_t* h) {
    if (!h) return RSREC_ERR_ARG;
    if (h->comm) {
        (void)hipSetDevice(h->device);
        (void)hipStreamSynchronize(h->stream);
        (void)g_rccl.comm_destroy(h->comm);
        h->comm = nullptr;
    }
    h->comm_rank = 0; h->comm_nranks = 1;
    return RSREC_OK;
}

// Collective over all ranks: every rank calls it with the same id and its own rank; the handle's device is the rank's GPU.
extern "C" int rsrec_comm_init(rsrec_t* h, int rank, int nranks, const char* id) {
    if (!h || !id || nranks < 1 || rank < 0 || rank >= nranks) return fail(h, RSREC_ERR_ARG, "rsrec_comm_init: bad argument");
    if (const char* msg = rccl_ready()) return fail(h, RSREC_ERR_DEVICE, "rsrec_comm_init: %s", msg);
    (void)rsrec_comm_destroy(h);
    HIPCK(h, hipSetDevice(h->device));
    NcclId u;
    memcpy(u.internal, id, RSREC_COMM_ID_BYTES);
    const int rc = g_rccl.comm_init_rank(&h->comm, nranks, u, rank);
    if (rc != 0) { h->comm = nullptr; return fail(h, RSREC_ERR_DEVICE, "ncclCommInitRank failed: %s", rccl_err(rc)); }
    h->comm_rank = rank; h->comm_nranks = nranks;
    return RSREC_OK;
}

// The same with the id exchanged through a file (hosts without MPI): rank 0 creates the id and publishes it at `path` (written to a
// temporary name, then renamed: readers never see a partial file); the other ranks wait for it up to `timeout_s` seconds.  `path` must
// be fresh for every communicator (rank 0 replaces an existing file before the others may read a stale one only if they start later:
// use a per-job name).
extern "C" int rsrec_comm_init_file(rsrec_t* h, int rank, int nranks, const char* path, double timeout_s) {
    if (!h || !path || nranks < 1 || rank < 0 || rank >= nranks) return fail(h, RSREC_ERR_ARG, "rsrec_comm_init_file: bad argument");
    char id[RSREC_COMM_ID_BYTES];
    if (rank == 0) {
        if (rsrec_comm_unique_id(id) != RSREC_OK) return fail(h, RSREC_ERR_DEVICE, "rsrec_comm_init_file: ncclGetUniqueId failed");
        const std::string tmp = std::string(path) + ".tmp";
        FILE* f = fopen(tmp.c_str(), "wb");
        if (!f || fwrite(id, 1, sizeof id, f) != sizeof id) { if (f) fclose(f); return fail(h, RSREC_ERR_ARG, "rsrec_comm_init_file: cannot write %s", tmp.c_str()); }
        fclose(f);
        if (rename(tmp.c_str(), path) != 0) return fail(h, RSREC_ERR_ARG, "rsrec_comm_init_file: cannot publish %s", path);
    } else {
        const auto t0 = std::chrono::steady_clock::now();
        for (;;) {
            FILE* f = fopen(path, "rb");
            if (f) {
                const size_t got = fread(id, 1, sizeof id, f);
                fclose(f);
                if (got == sizeof id) break;
            }
            if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout_s)
                return fail(h, RSREC_ERR_DEVICE, "rsrec_comm_init_file: no id at %s after %.0f s", path, timeout_s);
            std::this_thread::sleep_for(std::chrono::milliseconds(20));
        }
    }
    return rsrec_comm_init(h, rank, nranks, id);
}

// In-place sum over the ranks of n doubles (the reference's MPI_ALLREDUCE(MPI_IN_PLACE, buf, n, MPI_DOUBLE_PRECISION, MPI_SUM, ...),
// bands.f90:271-274).  buf: DEVICE memory (the images rsrec_pack_diag / rsrec_pack_moments / rsrec_block_ldos wrote: reduced where they
// lie, on the engine's stream) or host memory (staged through the device).  Returns when the result is in buf.  Without a communicator
// (or with one rank) it is the identity, like the reference built without MPI.
extern "C" int rsrec_allreduce_sum(rsrec_t* h, double* buf, size_t n) {
    if (!h || (!buf && n > 0)) return fail(h, RSREC_ERR_ARG, "rsrec_allreduce_sum: bad argument");
    if (n == 0) return RSREC_OK;
    HIPCK(h, hipSetDevice(h->device));
    if (!h->comm) { HIPCK(h, hipStreamSynchronize(h->stream)); return RSREC_OK; }
    const bool dev = is_device_ptr(buf);
    double* d = buf;
    if (!dev) {
        HIPCK(h, h->d_comm.reserve(n * sizeof(double)));
        d = h->d_comm.as<double>();
        XFER(xfer_h2d(h, d, buf, n * sizeof(double)));
    }
    const int rc = g_rccl.all_reduce(d, d, n, 8 /*ncclDouble*/, 0 /*ncclSum*/, h->comm, h->stream);
    if (rc != 0) return fail(h, RSREC_ERR_DEVICE, "ncclAllReduce failed: %s", rccl_err(rc));
    if (!dev) XFER(xfer_d2h(h, buf, d, n * sizeof(double)));
    HIPCK(h, hipStreamSynchronize(h->stream));
    return RSREC_OK;
}

extern "C" int rsrec_comm_size(rsrec_t* h, int* rank, int* nranks) {
    if (!h) return RSREC_ERR_ARG;
    if (rank) *rank = h->comm_rank;
    if (nranks) *nranks = h->comm ? h->comm_nranks : 1;
    return RSREC_OK;
}

extern "C" int rsrec_zsqr(rsrec_t* h, int nmat, double* b2_b) {
    if (!h || nmat < 0 || (nmat > 0 && !b2_b)) return fail(h, RSREC_ERR_ARG, "rsrec_zsqr: bad argument");
    if (nmat == 0) return RSREC_OK;
    HIPCK(h, hipSetDevice(h->device));
    const size_t bytes = (size_t)nmat * BLK * sizeof(double2);
    HIPCK(h, h->d_zsqr.reserve(bytes));              // (not d_mu: the Chebyshev moments of the last call stay resident there)
    HIPCK(h, h->d_status.reserve(64));
    HIPCK(h, hipMemsetAsync(h->d_status.p, 0, 64, h->stream));
    XFER(xfer_h2d(h, h->d_zsqr.p, b2_b, bytes));
    k_zsqr<<<nmat, 256, 0, h->stream>>>(h->d_zsqr.as<double2>(), h->d_status.as<int>());
    HIPCK(h, hipGetLastError());
    XFER(xfer_d2h(h, b2_b, h->d_zsqr.p, bytes));
    int status = 0;
    XFER(xfer_d2h(h, &status, h->d_status.p, 4));
    HIPCK(h, hipStreamSynchronize(h->stream));
    if (status & 1) return fail(h, RSREC_ERR_EIG, "Diagonalization error (18x18 Jacobi did not converge)");
    return RSREC_OK;
}

namespace {

// The Green stage produces far more than it consumes (13 MB of g0 per site for 2510 energies): the sites are cut into chunks,
// chunk c + 1 is computed on h->stream while chunk c leaves the device on h->copy_stream (two output buffers).  The caller's
// array is pageable, so the copy blocks the host -- after the next kernel has been queued.
// launch(s0, ns, out): queue the kernel for sites s0 .. s0 + ns - 1 of the call, writing ns * gbytes bytes at `out`.
template <class Launch>
int green_pipeline(rsrec_t* h, int nsites, size_t gbytes, double* g0, Launch launch) {
    if (!h->copy_stream) HIPCK(h, hipStreamCreateWithFlags(&h->copy_stream, hipStreamNonBlocking));
    for (hipEvent_t& e : h->ev_green)
        if (!e) HIPCK(h, hipEventCreateWithFlags(&e, hipEventDisableTiming));
    const size_t cap = ((size_t)1 << 29) / gbytes;                                  // <= 2 x 512 MiB of g0 on the device
    const int chunk = (int)std::max<size_t>(1, std::min<size_t>(cap, ((size_t)nsites + 7) / 8));
    HIPCK(h, h->d_green_out.reserve(2 * (size_t)chunk * gbytes));
    char* out[2] = {static_cast<char*>(h->d_green_out.p), static_cast<char*>(h->d_green_out.p) + (size_t)chunk * gbytes};
    std::vector<std::pair<hipEvent_t, hipEvent_t>> spans;
    auto drain = [&](int s0, int ns, int b) -> int {
        HIPCK(h, hipStreamWaitEvent(h->copy_stream, h->ev_green[b], 0));
        HIPCK(h, hipMemcpyAsync(reinterpret_cast<char*>(g0) + (size_t)s0 * gbytes, out[b], (size_t)ns * gbytes, hipMemcpyDeviceToHost, h->copy_stream));
        HIPCK(h, hipStreamSynchronize(h->copy_stream));
        return RSREC_OK;
    };
    int prev_s0 = -1, prev_ns = 0, c = 0;
    for (int s0 = 0; s0 < nsites; s0 += chunk, ++c) {
        const int ns = std::min(chunk, nsites - s0);
        hipEvent_t k0 = next_event(h);
        launch(s0, ns, out[c & 1]);
        hipEvent_t k1 = next_event(h);
        HIPCK(h, hipGetLastError());
        HIPCK(h, hipEventRecord(h->ev_green[c & 1], h->stream));
        spans.emplace_back(k0, k1);
        if (prev_s0 >= 0) XFER(drain(prev_s0, prev_ns, (c - 1) & 1));               // buffer (c - 1) & 1 is free again before kernel c + 1 is queued
        prev_s0 = s0; prev_ns = ns;
    }
    if (prev_s0 >= 0) XFER(drain(prev_s0, prev_ns, (c - 1) & 1));
    HIPCK(h, hipStreamSynchronize(h->stream));
    for (auto& sp : spans) h->t_hop_ms += ev_ms(sp.first, sp.second);              // "hop_ms": the Green kernels themselves; total_ms includes the transfers
    return RSREC_OK;
}

}  // namespace

// green%block_green / bgreen (green.f90:588-621, :1191-1339): g0(:,:,:,site) from the block coefficients of every site.
extern "C" int rsrec_block_green(rsrec_t* h, int nsites, int lld, int nen, const double* ene, double eta_re, double eta_im, int sym_term,
                                 const double* a_inf, const double* b_inf, const double* a_b, const double* b_sqrt, double* g0) {
    if (!h) return RSREC_ERR_ARG;
    if (nsites < 0 || lld < 1 || nen < 0 || (nsites > 0 && nen > 0 && (!ene || !a_inf || !b_inf || !a_b || !b_sqrt || !g0)))
        return fail(h, RSREC_ERR_ARG, "rsrec_block_green: bad argument");
    if (nsites == 0 || nen == 0) return RSREC_OK;
    HIPCK(h, hipSetDevice(h->device));
    const size_t cbytes = (size_t)lld * BLK * sizeof(double2);       // coefficients of one site (each of a_b, b_sqrt)
    const size_t tbytes = (size_t)BLK * sizeof(double);              // terminator of one site (each of a_inf, b_inf)
    const size_t gbytes = (size_t)nen * BLK * sizeof(double2);       // g0 of one site
    const size_t in_site = 2 * cbytes + 2 * tbytes;
    const int super = (int)std::max<size_t>(1, std::min<size_t>((size_t)nsites, ((size_t)4 << 30) / in_site));   // <= 4 GiB of coefficients resident
    HIPCK(h, h->d_green_in.reserve((size_t)super * in_site + (size_t)nen * sizeof(double)));
    char* base = static_cast<char*>(h->d_green_in.p);
    double* d_ene = reinterpret_cast<double*>(base);
    double2* d_ab = reinterpret_cast<double2*>(base + (size_t)nen * sizeof(double));
    double2* d_bs = reinterpret_cast<double2*>(reinterpret_cast<char*>(d_ab) + (size_t)super * cbytes);
    double* d_ai = reinterpret_cast<double*>(reinterpret_cast<char*>(d_bs) + (size_t)super * cbytes);
    double* d_bi = reinterpret_cast<double*>(reinterpret_cast<char*>(d_ai) + (size_t)super * tbytes);
    XFER(xfer_h2d(h, d_ene, ene, (size_t)nen * sizeof(double)));
    reset_timing(h);
    hipEvent_t ev0 = next_event(h);
    for (int u0 = 0; u0 < nsites; u0 += super) {
        const int nu = std::min(super, nsites - u0);
        XFER(xfer_h2d(h, d_ab, a_b + (size_t)u0 * lld * BLK * 2, (size_t)nu * cbytes));
        XFER(xfer_h2d(h, d_bs, b_sqrt + (size_t)u0 * lld * BLK * 2, (size_t)nu * cbytes));
        XFER(xfer_h2d(h, d_ai, a_inf + (size_t)u0 * BLK, (size_t)nu * tbytes));
        XFER(xfer_h2d(h, d_bi, b_inf + (size_t)u0 * BLK, (size_t)nu * tbytes));
        XFER(green_pipeline(h, nu, gbytes, g0 + (size_t)u0 * nen * BLK * 2, [&](int s0, int ns, char* out) {
            const dim3 grid((nen + GREEN_WAVES - 1) / GREEN_WAVES, ns);
            k_block_green<false><<<grid, GREEN_WAVES * 64, 0, h->stream>>>(lld, nen, d_ene, eta_re, eta_im, sym_term, d_ai + (size_t)s0 * BLK, d_bi + (size_t)s0 * BLK,
                                                                    d_ab + (size_t)s0 * lld * BLK, d_bs + (size_t)s0 * lld * BLK, reinterpret_cast<double2*>(out));
        }));
    }
    hipEvent_t ev1 = next_event(h);
    HIPCK(h, hipStreamSynchronize(h->stream));
    h->t_total_ms = ev_ms(ev0, ev1);
    return RSREC_OK;
}

namespace {

// get_terminf for n sites on the device: a_inf, b_inf [site][324], optional means
int launch_terminator(rsrec_t* h, int n, int lld, const double2* d_ab, const double2* d_bs, double* d_ainf, double* d_binf) {
    if (lld < 2) return fail(h, RSREC_ERR_ARG, "terminator needs lld >= 2");
    // one LDS column of 2 * lld doubles per thread
    int T = 64;
    while (T > 8 && (size_t)2 * lld * T * sizeof(double) > (size_t)128 * 1024) T >>= 1;
    const size_t lds = (size_t)2 * lld * T * sizeof(double);
    if (lds > (size_t)150 * 1024) return fail(h, RSREC_ERR_ARG, "terminator: lld = %d too deep for the LDS staging", lld);
    if (lds > h->term_attr_lds) {
        HIPCK(h, hipFuncSetAttribute(reinterpret_cast<const void*>(k_terminator), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        h->term_attr_lds = lds;
    }
    k_terminator<<<dim3((BLK + T - 1) / T, n), T, lds, h->stream>>>(lld, d_ab, d_bs, d_ainf, d_binf);
    HIPCK(h, hipGetLastError());
    return RSREC_OK;
}

}  // namespace

// recursion%get_terminf (recursion.f90:2092-2135) for `nsites` sites, host arrays in and out.
extern "C" int rsrec_terminator(rsrec_t* h, int nsites, int lld, const double* a_b, const double* b_sqrt, double* a_inf, double* b_inf,
                                double* a_inf0, double* b_inf0) {
    if (!h) return RSREC_ERR_ARG;
    if (nsites < 0 || lld < 2 || (nsites > 0 && (!a_b || !b_sqrt || !a_inf || !b_inf))) return fail(h, RSREC_ERR_ARG, "rsrec_terminator: bad argument");
    if (nsites == 0) return RSREC_OK;
    HIPCK(h, hipSetDevice(h->device));
    const size_t cbytes = (size_t)nsites * lld * BLK * sizeof(double2), tbytes = (size_t)nsites * BLK * sizeof(double);
    HIPCK(h, h->d_green_in.reserve(2 * cbytes));
    HIPCK(h, h->d_term.reserve(2 * tbytes + 2 * (size_t)nsites * sizeof(double)));
    double2* d_ab = h->d_green_in.as<double2>();
    double2* d_bs = d_ab + (size_t)nsites * lld * BLK;
    double* d_ai = h->d_term.as<double>();
    double* d_bi = d_ai + (size_t)nsites * BLK;
    double* d_a0 = d_bi + (size_t)nsites * BLK;
    XFER(xfer_h2d(h, d_ab, a_b, cbytes));
    XFER(xfer_h2d(h, d_bs, b_sqrt, cbytes));
    reset_timing(h);
    hipEvent_t e0 = next_event(h);
    int rc = launch_terminator(h, nsites, lld, d_ab, d_bs, d_ai, d_bi);
    if (rc) return rc;
    k_terminator_means<<<(nsites + 63) / 64, 64, 0, h->stream>>>(d_ai, d_bi, d_a0, d_a0 + nsites, nsites);
    hipEvent_t e1 = next_event(h);
    XFER(xfer_d2h(h, a_inf, d_ai, tbytes));
    XFER(xfer_d2h(h, b_inf, d_bi, tbytes));
    if (a_inf0) XFER(xfer_d2h(h, a_inf0, d_a0, (size_t)nsites * sizeof(double)));
    if (b_inf0) XFER(xfer_d2h(h, b_inf0, d_a0 + nsites, (size_t)nsites * sizeof(double)));
    HIPCK(h, hipStreamSynchronize(h->stream));
    h->t_total_ms = h->t_hop_ms = ev_ms(e0, e1);
    return RSREC_OK;
}

// dos%density for every (site, direction) of a scalar-recursion run (density_of_states.f90:248-363, bprldos :370-404; called by
// green%sgreen, green.f90:661): a, b2 (llmax, 18, nsites, nmdir) as recursion%a / %b2 hold them, dw_l, cshi (18, nsites) = the potential
// parameters of the sites' atoms, ene (npts) -> tdens (18, npts, nsites, nmdir).
extern "C" int rsrec_scalar_density(rsrec_t* h, int nsites, int nmdir, int llmax, int lld, const double* a, const double* b2, int npts, const double* ene,
                                    const double* dw_l, const double* cshi, double* tdens) {
    if (!h) return RSREC_ERR_ARG;
    if (nsites < 0 || nmdir < 1 || lld < 2 || lld > llmax || npts < 0 || ((nsites > 0 && npts > 0) && (!a || !b2 || !ene || !dw_l || !cshi || !tdens)))
        return fail(h, RSREC_ERR_ARG, "rsrec_scalar_density: bad argument");
    if (nsites == 0 || npts == 0) return RSREC_OK;
    HIPCK(h, hipSetDevice(h->device));
    const int nchain = NB * nsites * nmdir;
    const size_t cb = (size_t)nchain * llmax * sizeof(double), pb = (size_t)NB * nsites * sizeof(double), eb = (size_t)npts * sizeof(double),
                 gb = 2 * (size_t)nchain * sizeof(double), tb = (size_t)nchain * npts * sizeof(double);
    HIPCK(h, h->d_green_in.reserve(2 * cb + 2 * pb + eb + gb));
    HIPCK(h, h->d_green_out.reserve(tb));
    double* d_a = h->d_green_in.as<double>();
    double* d_b = d_a + (size_t)nchain * llmax;
    double* d_dw = d_b + (size_t)nchain * llmax;
    double* d_cs = d_dw + (size_t)NB * nsites;
    double* d_en = d_cs + (size_t)NB * nsites;
    double* d_ed = d_en + npts;
    double* d_t = h->d_green_out.as<double>();
    XFER(xfer_h2d(h, d_a, a, cb));
    XFER(xfer_h2d(h, d_b, b2, cb));
    XFER(xfer_h2d(h, d_dw, dw_l, pb));
    XFER(xfer_h2d(h, d_cs, cshi, pb));
    XFER(xfer_h2d(h, d_en, ene, eb));
    reset_timing(h);
    hipEvent_t e0 = next_event(h);
    // one LDS column of 2 lld doubles per thread: as many threads per workgroup as 64 KB hold (64 up to lld = 64, one at lld = 4096)
    const int T = (int)std::max<size_t>(1, std::min<size_t>(64, (64 * 1024) / (2 * (size_t)lld * sizeof(double))));
    const size_t lds = 2 * (size_t)lld * T * sizeof(double);
    if (lds > 64 * 1024) return fail(h, RSREC_ERR_ARG, "rsrec_scalar_density: lld = %d too deep for the band-edge kernel", lld);
    k_scalar_edges<<<(nchain + T - 1) / T, T, lds, h->stream>>>(lld, llmax, nchain, d_a, d_b, d_ed);
    k_scalar_density<<<dim3((npts + 127) / 128, nchain), 128, 0, h->stream>>>(lld, llmax, npts, nsites, d_a, d_b, d_en, d_dw, d_cs, d_ed, d_t);
    HIPCK(h, hipGetLastError());
    hipEvent_t e1 = next_event(h);
    XFER(xfer_d2h(h, tdens, d_t, tb));
    HIPCK(h, hipStreamSynchronize(h->stream));
    h->t_total_ms = h->t_hop_ms = ev_ms(e0, e1);
    return RSREC_OK;
}

// The whole LDOS stage for the sites of the last rsrec_block_lanczos call, from the coefficients it left on the device:
// zsqr (recursion.f90:1980) -> get_terminf (:2092) -> bgreen (green.f90:1191) -> the reduction of calculate_fermi (bands.f90:258-268).
// Only the densities of states leave the GPU (18 doubles per site and energy instead of 648).
extern "C" int rsrec_block_ldos(rsrec_t* h, int nen, const double* ene, double eta_re, double eta_im, int sym_term, int site_offset, int nsites_total,
                                double* dtot, double* dosia, double* dosial, double* a_inf_out, double* b_inf_out) {
    if (!h) return RSREC_ERR_ARG;
    if (nen < 1 || !ene || !dtot || !dosia || !dosial || site_offset < 0) return fail(h, RSREC_ERR_ARG, "rsrec_block_ldos: bad argument");
    if (h->res_kind != 1) return fail(h, RSREC_ERR_ARG, "rsrec_block_ldos: no block-Lanczos coefficients resident (call rsrec_block_lanczos first)");
    const int n = h->res_n, lld = h->res_lld;
    if (site_offset + n > nsites_total) return fail(h, RSREC_ERR_ARG, "rsrec_block_ldos: sites %d..%d outside 1..%d", site_offset + 1, site_offset + n, nsites_total);
    HIPCK(h, hipSetDevice(h->device));
    h->n_ldos_calls++;
    const size_t cel = (size_t)n * lld * BLK;
    HIPCK(h, h->d_bsqrt.reserve(cel * sizeof(double2)));
    HIPCK(h, h->d_term.reserve(2 * (size_t)n * BLK * sizeof(double) + 2 * (size_t)n * sizeof(double)));
    HIPCK(h, h->d_gim.reserve((size_t)n * nen * NB * sizeof(double) + (size_t)nen * sizeof(double)));
    const size_t img = (size_t)nen * ((size_t)nsites_total * (NB + 1) + 1);          // dosial + dosia + dtot
    const bool dev = is_device_ptr(dtot) && is_device_ptr(dosia) && is_device_ptr(dosial);
    if (!dev) HIPCK(h, h->d_ldos.reserve(img * sizeof(double)));
    HIPCK(h, h->d_status.reserve(64));
    HIPCK(h, hipMemsetAsync(h->d_status.p, 0, 64, h->stream));
    double* d_gim = h->d_gim.as<double>();
    double* d_ene = d_gim + (size_t)n * nen * NB;
    XFER(xfer_h2d(h, d_ene, ene, (size_t)nen * sizeof(double)));
    reset_timing(h);
    hipEvent_t e0 = next_event(h);
    const double2* dA = h->d_coefA.as<double2>();
    double2* dBs = h->d_bsqrt.as<double2>();
    // b2_b of the recursion stays B^2 (the caller may still fetch it); the stage works on its own square root
    HIPCK(h, hipMemcpyAsync(dBs, h->d_coefB.p, cel * sizeof(double2), hipMemcpyDeviceToDevice, h->stream));
    k_zsqr<<<n * lld, 256, 0, h->stream>>>(dBs, h->d_status.as<int>());
    double* d_ai = h->d_term.as<double>();
    double* d_bi = d_ai + (size_t)n * BLK;
    int rc = launch_terminator(h, n, lld, dA, dBs, d_ai, d_bi);
    if (rc) return rc;
    hipEvent_t k0 = next_event(h);
    {
        const dim3 grid((nen + GREEN_WAVES - 1) / GREEN_WAVES, n);
        k_block_green<true><<<grid, GREEN_WAVES * 64, 0, h->stream>>>(lld, nen, d_ene, eta_re, eta_im, sym_term, d_ai, d_bi, dA, dBs, nullptr, d_gim);
    }
    hipEvent_t k1 = next_event(h);
    double* o_dosial = dev ? dosial : h->d_ldos.as<double>();
    double* o_dosia = dev ? dosia : o_dosial + (size_t)nsites_total * NB * nen;
    double* o_dtot = dev ? dtot : o_dosia + (size_t)nsites_total * nen;
    k_ldos_finish<<<(nen + 63) / 64, 64, 0, h->stream>>>(d_gim, n, nen, site_offset, nsites_total, o_dosial, o_dosia, o_dtot);
    HIPCK(h, hipGetLastError());
    hipEvent_t e1 = next_event(h);
    if (!dev) {
        XFER(xfer_d2h(h, dosial, o_dosial, (size_t)nsites_total * NB * nen * sizeof(double)));
        XFER(xfer_d2h(h, dosia, o_dosia, (size_t)nsites_total * nen * sizeof(double)));
        XFER(xfer_d2h(h, dtot, o_dtot, (size_t)nen * sizeof(double)));
    }
    if (a_inf_out) XFER(xfer_d2h(h, a_inf_out, d_ai, (size_t)n * BLK * sizeof(double)));
    if (b_inf_out) XFER(xfer_d2h(h, b_inf_out, d_bi, (size_t)n * BLK * sizeof(double)));
    int status = 0;
    XFER(xfer_d2h(h, &status, h->d_status.p, 4));
    HIPCK(h, hipStreamSynchronize(h->stream));
    h->t_total_ms = ev_ms(e0, e1);
    h->t_hop_ms = ev_ms(k0, k1);                      // the Green kernel alone
    h->t_rest_ms = h->t_total_ms - h->t_hop_ms;       // zsqr + terminator + reduction
    if (status & 1) return fail(h, RSREC_ERR_EIG, "Diagonalization error (18x18 Jacobi did not converge)");
    return RSREC_OK;
}

// green%chebyshev_green (green.f90:1030-1108): g0 from the Chebyshev moments of every site.
extern "C" int rsrec_chebyshev_green(rsrec_t* h, int nsites, int lld, int nen, const double* ene, double energy_min, double energy_max,
                                     const double* mu_n, double* g0) {
    if (!h) return RSREC_ERR_ARG;
    if (nsites < 0 || lld < 1 || nen < 0 || (nsites > 0 && nen > 0 && (!ene || !mu_n || !g0))) return fail(h, RSREC_ERR_ARG, "rsrec_chebyshev_green: bad argument");
    if (nsites == 0 || nen == 0) return RSREC_OK;
    HIPCK(h, hipSetDevice(h->device));
    const int nm = 2 * lld + 2;
    // scale/shift as the reference writes them (default-REAL literals 2 and 0.3, green.f90:1046-1047) and the Jackson kernel
    // (math.f90:1641-1655; real(ll) is a default-REAL conversion, exact for these small integers)
    const double a = (energy_max - energy_min) / (double)(2.0f - 0.3f), b = (energy_max + energy_min) / 2;
    std::vector<double> kern(nm);
    {
        const double pi = 3.14159265358979323846;
        for (int ll = 1; ll <= nm; ++ll) {
            const double theta = pi * ((double)ll - 1.0) / ((double)nm + 1.0);
            kern[ll - 1] = (((double)nm - ((double)ll - 1.0) + 1.0) * cos(theta) + sin(theta) / tan(pi / ((double)nm + 1.0))) / ((double)nm + 1.0);
            if (ll > 1) kern[ll - 1] *= 2.0;                 // mu_ng(:,:,2:) *= 2 (:1074)
        }
    }
    const size_t mbytes = (size_t)nm * BLK * sizeof(double2), gbytes = (size_t)nen * BLK * sizeof(double2);
    const int super = (int)std::max<size_t>(1, std::min<size_t>((size_t)nsites, ((size_t)4 << 30) / mbytes));
    HIPCK(h, h->d_green_in.reserve((size_t)super * mbytes + (size_t)(nen + nm) * sizeof(double)));
    double* d_ene = static_cast<double*>(h->d_green_in.p);
    double* d_kern = d_ene + nen;
    double2* d_mu = reinterpret_cast<double2*>(d_kern + nm);
    XFER(xfer_h2d(h, d_ene, ene, (size_t)nen * sizeof(double)));
    XFER(xfer_h2d(h, d_kern, kern.data(), (size_t)nm * sizeof(double)));
    reset_timing(h);
    hipEvent_t ev0 = next_event(h);
    for (int u0 = 0; u0 < nsites; u0 += super) {
        const int nu = std::min(super, nsites - u0);
        XFER(xfer_h2d(h, d_mu, mu_n + (size_t)u0 * nm * BLK * 2, (size_t)nu * mbytes));
        XFER(green_pipeline(h, nu, gbytes, g0 + (size_t)u0 * nen * BLK * 2, [&](int s0, int ns, char* out) {
            k_chebyshev_green<<<dim3(nen, ns), 256, (size_t)nm * sizeof(double2), h->stream>>>(nm, nen, d_ene, a, b, d_kern, d_mu + (size_t)s0 * nm * BLK, reinterpret_cast<double2*>(out));
        }));
    }
    hipEvent_t ev1 = next_event(h);
    HIPCK(h, hipStreamSynchronize(h->stream));
    h->t_total_ms = ev_ms(ev0, ev1);
    return RSREC_OK;
}

namespace {

template <class L, bool MFMA>
int run_chebyshev(rsrec_t* h, int nsites, int nseed, const int32_t* seed_atoms, const double* seed_coef, int lld, double a, double b, double* mu_n) {
    int rc = 0;
    const int kk = h->kk;
    const bool hoh = h->hoh != 0;
    const int napply = lld + 1;                                  // first moment + lld steps
    const int nlev = (hoh ? 2 * napply : napply) + 1;
    const int nmom = 2 * lld + 2;
    const size_t velems = (size_t)(kk + 1) * BLD;
    // matrix-core set: k_mfma_cheb epilogue; large launches and hoh use k_spmm5 on CI vectors, small launches of the plain
    // operator the cooperative k_spmm4<4> on LayoutRM
    const bool mf_cheb = MFMA;
    const bool use_kp = MFMA && (hoh || h->opt_spmm5 == 2 || (h->opt_spmm5 == 1 && (long)std::min(nsites, 64) * (kk / GROUP + 1) >= 4096) || !spmm4_usable(h));
    const int nvec = MFMA ? (hoh ? 5 : 4) : (hoh ? 4 : 3);
    const int ci = use_kp ? 1 : 0;                              // vectors of this call are CI (else LayoutRM / LayoutCM)
    if (MFMA && !use_kp) { rc = ensure_s4(h); if (rc) return rc; }
    // the moments of ALL chains of the call stay on the device (rsrec_pack_moments): reserved BEFORE the batch is planned from the free
    // memory, so a call over many sites (nrec ~ kk) sizes its vectors around them instead of failing behind them
    h->res_kind = 0;
    HIPCK(h, h->d_mu.reserve((size_t)nsites * nmom * BLK * sizeof(double2)));
    // with the Chebyshev step fused into the SpMM's epilogue (the default of the matrix-core set) H psi is never held: vector 3 is neither
    // allocated nor cleared
    const bool use_v3 = !(mf_cheb && (hoh || use_kp) && h->opt_cheb_fused);
    BatchPlan bp;
    rc = plan_batch(h, nsites, nvec - (use_v3 ? 0 : 1), velems / 2, bp);
    if (rc) return rc;
    const int B = bp.batch, nblk = bp.nblk;
    for (int v = 0; v < nvec; ++v) if (v != 3 || use_v3) HIPCK(h, h->d_vec[v].reserve((size_t)B * velems * sizeof(double)));
    HIPCK(h, h->d_partial.reserve(std::max((size_t)B * nblk * 2 * BLK * sizeof(double2), (size_t)B * 256 * 2 * 1296 * sizeof(double))));
    h->p2_slot = (size_t)B * 16 * 2 * 1296;
    HIPCK(h, h->d_partial2.reserve(2 * h->p2_slot * sizeof(double)));
    if (!h->side_stream) {
        HIPCK(h, hipStreamCreateWithFlags(&h->side_stream, hipStreamNonBlocking));
        HIPCK(h, hipEventCreateWithFlags(&h->ev_orth, hipEventDisableTiming));
        HIPCK(h, hipEventCreateWithFlags(&h->ev_bred, hipEventDisableTiming));
    }
    const bool side = h->opt_side && h->side_stream;     // moment reduction of level t under the SpMM of level t + 1 (it feeds nothing on the device)
    bool red_pending = false;
    HIPCK(h, h->d_status.reserve(64));
    HIPCK(h, h->d_seed.reserve((size_t)B * nseed * 4));
    HIPCK(h, h->d_seedcoef.reserve((size_t)B * (nseed + 1) * sizeof(double2)));
    HIPCK(h, hipMemsetAsync(h->d_status.p, 0, 64, h->stream));
    const DevProblem P = make_problem(h);
    const size_t mstride = (size_t)nmom * BLK;
    hipEvent_t ev_begin = next_event(h);
    std::vector<std::pair<hipEvent_t, hipEvent_t>> hop_ev;
    for (int c0 = 0; c0 < nsites; c0 += B) {
        const int nb = std::min(B, nsites - c0);
        double2* mu = h->d_mu.as<double2>() + (size_t)c0 * mstride;       // this batch's slice of the resident moments
        const auto th0 = std::chrono::steady_clock::now();
        std::vector<int> seeds0((size_t)nb * nseed);
        std::vector<double> coef((size_t)nb * nseed * 2 + nb);     // coefficients, then one mu_1 scale per chain
        for (int c = 0; c < nb; ++c) {
            double m0 = 0.0;
            for (int k = 0; k < nseed; ++k) {
                const size_t q = (size_t)c * nseed + k, g = (size_t)(c0 + c) * nseed + k;
                seeds0[q] = seed_atoms[g] - 1;
                coef[2 * q] = seed_coef ? seed_coef[2 * g] : 1.0;
                coef[2 * q + 1] = seed_coef ? seed_coef[2 * g + 1] : 0.0;
            }
            // mu_1 = sum over seed atoms of |final coefficient|^2 (later seeds overwrite earlier ones on the same atom)
            for (int k = 0; k < nseed; ++k) {
                bool overwritten = false;
                for (int k2 = k + 1; k2 < nseed; ++k2) overwritten |= seeds0[(size_t)c * nseed + k2] == seeds0[(size_t)c * nseed + k];
                const size_t q = (size_t)c * nseed + k;
                if (!overwritten) m0 += coef[2 * q] * coef[2 * q] + coef[2 * q + 1] * coef[2 * q + 1];
            }
            coef[(size_t)nb * nseed * 2 + c] = m0;
        }
        int ostride = kk;
        rc = upload_regions(h, seeds0.data(), nb, nseed, nlev, napply, hoh, MFMA, ostride, h->n_atom_steps, h->n_block_mult);
        if (rc) return rc;
        h->n_req_flop += required_hop_flops(h, h->s5_op);
        XFER(xfer_h2d(h, h->d_seed.p, seeds0.data(), seeds0.size() * 4));
        XFER(xfer_h2d(h, h->d_seedcoef.p, coef.data(), coef.size() * 8));
        HIPCK(h, hipStreamSynchronize(h->stream));
        h->t_host_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - th0).count();
        ChainView CV;
        CV.order = h->cur_order; CV.cum = h->cur_cum; CV.obase = h->cur_cum + (size_t)h->cur_nrows * nlev; CV.nlev = nlev; CV.vstride = velems; CV.cpo = 1; CV.ostride = ostride;
        ChainView CVp = CV;                                   // the streaming moment pass walks the level-major lists (upload_regions)
        CVp.cum = h->cur_cum + (size_t)2 * h->cur_nrows * nlev; CVp.obase = h->cur_cum + (size_t)3 * h->cur_nrows * nlev;
        for (int v = 0; v < nvec; ++v) if (v != 3 || use_v3) HIPCK(h, hipMemsetAsync(h->d_vec[v].p, 0, (size_t)nb * velems * sizeof(double), h->stream));
        HIPCK(h, hipMemsetAsync(mu, 0, (size_t)nb * mstride * sizeof(double2), h->stream));
        double* p0 = h->d_vec[0].as<double>();
        double* p1 = h->d_vec[1].as<double>();
        double* p2 = h->d_vec[2].as<double>();
        double* tmp = h->d_vec[3].as<double>();
        double* tmp2 = h->d_vec[4].as<double>();
        const dim3 grid_mf(std::max(1, std::min(mfma_workgroups_per_chain(h, B), (ostride / GROUP + MF_WAVES - 1) / MF_WAVES)), nb);
        if (ci) k_seed<LayoutCI><<<nb, 64, 0, h->stream>>>(p0, velems, h->d_seed.as<int>(), h->d_seedcoef.as<double2>(), nseed);
        else k_seed<L><<<nb, 64, 0, h->stream>>>(p0, velems, h->d_seed.as<int>(), h->d_seedcoef.as<double2>(), nseed);
        k_set_identity<<<nb, 256, 0, h->stream>>>(mu, mstride, h->d_seedcoef.as<double>() + (size_t)nb * nseed * 2);   // mu_1 (cheb_0th_mom :2157)
        double* hps = (use_kp && hoh) ? tmp2 : nullptr;         // hoh: h psi of the first pass
        const dim3 grid(nblk, nb);
        for (int t = 1; t <= napply; ++t) {      // t = 1: first moment; t >= 2: recursion step ll = t-1
            const bool first = (t == 1);
            const int lv_final = hoh ? 2 * t : t;
            hipEvent_t e0 = next_event(h);
            ApplyArgs G{};
            G.partial = h->d_partial.as<double2>();
            G.a = a; G.b = b;
            double* src = first ? p0 : p1;
            double* dst = first ? p1 : p2;
            if (mf_cheb) {
                SpmmDims SD{kk, P.nslots, P.nmax, nlev, 1, ostride, lv_final, velems, CV.obase, nb};
                const dim3 gl = level_grid(h, grid_mf, lv_final);
                // k_spmm5 forms the new vector in its epilogue (dst = (H src - b src)/a [* 2 - p0]); k_mfma_cheb then only sums the Grams
                const bool fused = (hoh || use_kp) && h->opt_cheb_fused;
                S5Epilogue E;
                if (fused) { E.kind = first ? 1 : 2; E.cur = src; E.old = first ? nullptr : p0; E.a = a; E.b = b; }
                if (hoh) {
                    SD.level = 2 * t - 1;
                    launch_s5<false>(h, s5_grid(h, grid_mf, 2 * t - 1), SD, CV.order, CV.cum, P.iz, h->s5_op, 0, src, hps);
                    SD.level = lv_final;
                    launch_s5<true>(h, s5_grid(h, grid_mf, lv_final), SD, CV.order, CV.cum, P.iz, h->s5_op, 1, hps, fused ? dst : tmp, src, nullptr, 0, E);
                } else if (use_kp) launch_s5<false>(h, s5_grid(h, grid_mf, lv_final), SD, CV.order, CV.cum, P.iz, h->s5_op, 0, src, fused ? dst : tmp, nullptr, nullptr, 0, E);
                else { rc = launch_spmm(h, SD, CV, P, 0, src, tmp, grid_mf); if (rc) return rc; }
                hipEvent_t e1 = next_event(h);
                hop_ev.emplace_back(e0, e1);
                h->n_hop_launch += hoh ? 2 : 1;
                double* gp = h->d_partial.as<double>();
                if (red_pending) { HIPCK(h, hipStreamWaitEvent(h->stream, h->ev_bred, 0)); red_pending = false; }   // gp is free again
                if (fused) {
                    if (first) k_mfma_cheb<true, true><<<gl, MF_WAVES * 64, 0, h->stream>>>(CVp, lv_final, kk, nullptr, src, nullptr, dst, a, b, gp);
                    else k_mfma_cheb<false, true><<<gl, MF_WAVES * 64, 0, h->stream>>>(CVp, lv_final, kk, nullptr, src, nullptr, dst, a, b, gp);
                } else if (first) k_mfma_cheb<true><<<gl, MF_WAVES * 64, 0, h->stream>>>(CVp, lv_final, kk, tmp, src, nullptr, dst, a, b, gp);
                else k_mfma_cheb<false><<<gl, MF_WAVES * 64, 0, h->stream>>>(CVp, lv_final, kk, tmp, src, p0, dst, a, b, gp);
                hipStream_t rs = h->stream;
                if (side) {
                    HIPCK(h, hipEventRecord(h->ev_orth, h->stream));
                    HIPCK(h, hipStreamWaitEvent(h->side_stream, h->ev_orth, 0));
                    rs = h->side_stream;
                }
                int n2 = gl.x; const double* gp2 = presum(h, gp, nb, n2, 2 * 1296, rs, side ? 1 : 0);
                k_reduce_cheb_mf<<<nb, 1024, 0, rs>>>(gp2, n2, first ? 1 : 0, t - 1, mu, mstride, h->d_status.as<int>(), nseed > 1 ? 1 : 0, ci);
                if (side) { HIPCK(h, hipEventRecord(h->ev_bred, h->side_stream)); red_pending = true; }
                if (!first) { double* o = p0; p0 = p1; p1 = p2; p2 = o; }
                continue;
            }
            if (!hoh) {
                G.in = src; G.cur = src; G.v0 = p0; G.out = dst; G.level = lv_final;
                if (first) k_apply<AM_CHEB1, L><<<grid, NTHREADS, 0, h->stream>>>(P, CV, G);
                else k_apply<AM_CHEBN, L><<<grid, NTHREADS, 0, h->stream>>>(P, CV, G);
            } else {
                G.in = src; G.out = tmp; G.level = 2 * t - 1;
                k_apply<AM_STORE, L><<<grid, NTHREADS, 0, h->stream>>>(P, CV, G);
                G.in = tmp; G.v1 = tmp; G.cur = src; G.v0 = p0; G.out = dst; G.level = lv_final;
                if (first) k_apply<AM_HOH_CHEB1, L><<<grid, NTHREADS, 0, h->stream>>>(P, CV, G);
                else k_apply<AM_HOH_CHEBN, L><<<grid, NTHREADS, 0, h->stream>>>(P, CV, G);
            }
            hipEvent_t e1 = next_event(h);
            hop_ev.emplace_back(e0, e1);
            h->n_hop_launch += hoh ? 2 : 1;
            k_reduce_cheb<<<nb, 1024, 0, h->stream>>>(h->d_partial.as<double2>(), nblk, first ? 1 : 0, t - 1, mu, mstride, h->d_status.as<int>(), nseed > 1 ? 1 : 0);
            if (!first) { double* o = p0; p0 = p1; p1 = p2; p2 = o; }   // psi0 <- psi1 <- psi2 (:2585-2587) by rotating buffers
        }
        HIPCK(h, hipGetLastError());
        if (red_pending) { HIPCK(h, hipStreamWaitEvent(h->stream, h->ev_bred, 0)); red_pending = false; }
        XFER(xfer_d2h(h, mu_n + (size_t)c0 * mstride * 2, mu, (size_t)nb * mstride * sizeof(double2)));
        HIPCK(h, hipStreamSynchronize(h->stream));
    }
    hipEvent_t ev_end = next_event(h);
    HIPCK(h, hipStreamSynchronize(h->stream));
    h->t_total_ms = ev_ms(ev_begin, ev_end);
    for (auto& pr : hop_ev) h->t_hop_ms += ev_ms(pr.first, pr.second);
    h->t_rest_ms = h->t_total_ms - h->t_hop_ms;
    int status = 0;
    XFER(xfer_d2h(h, &status, h->d_status.p, 4));
    if (status & 2) return fail(h, RSREC_ERR_DIVERGED, "Chebyshev moments did not converge. Check energy limits energy_min and energy_max");
    h->res_kind = 2; h->res_n = nsites; h->res_lld = lld; h->res_sqrt = 0;
    return RSREC_OK;
}

}  // namespace

extern "C" int rsrec_chebyshev_seeded(rsrec_t* h, int nchains, int nseed, const int32_t* seed_atoms, const double* seed_coef, int lld, double a, double b,
                                      double* mu_n) {
    int rc = check_ready(h, "rsrec_chebyshev");
    if (rc) return rc;
    if (nchains < 0 || nseed < 1 || nseed > 8 || lld < 1 || !mu_n || (nchains > 0 && !seed_atoms) || a == 0.0) return fail(h, RSREC_ERR_ARG, "rsrec_chebyshev: bad argument");
    for (int q = 0; q < nchains * nseed; ++q)
        if (seed_atoms[q] < 1 || seed_atoms[q] > h->kk) return fail(h, RSREC_ERR_ARG, "rsrec_chebyshev: seed atom %d outside 1..%d", seed_atoms[q], h->kk);
    HIPCK(h, hipSetDevice(h->device));
    reset_timing(h);
    if (nchains == 0) return RSREC_OK;
    if (h->opt_kernels != 1 && h->s5_built) return run_chebyshev<LayoutRM, true>(h, nchains, nseed, seed_atoms, seed_coef, lld, a, b, mu_n);
    return run_chebyshev<LayoutCM, false>(h, nchains, nseed, seed_atoms, seed_coef, lld, a, b, mu_n);
}

extern "C" int rsrec_chebyshev(rsrec_t* h, int nsites, const int32_t* seed_atoms, int lld, double a, double b, double* mu_n) {
    return rsrec_chebyshev_seeded(h, nsites, 1, seed_atoms, nullptr, lld, a, b, mu_n);
}

// ------------------------------------------------------------------------------------------------------------------
// Stochastic Kubo double moments
namespace {

// Operator tables of a velocity-type operator (recursion.f90:587-784): which = 0 -> kubo_op[0] (v_a), 1 -> kubo_op[1] (v_b).
//   set 0: V itself -- per-type blocks v_op(:,:,slot,type) for the bulk atoms; the reference has no velocity operator for the
//          per-atom (impurity) region yet (":591 NOT YET IMPLEMENTED"): those rows of V psi are zero, as there.
//   set 1 (hoh): -vo_op(:,:,slot,type) for slots >= 2 and the identity in the extra slot, so that one pass over h psi with V psi as
//          second input gives  V psi - sum_{slot >= 2} vo_slot (h psi)_nbr  (velo_hoh_vec_matmul :750-776; its on-site vo term is
//          commented out in the reference, its e_nu / l.s terms are zero).
int build_kubo_operator(rsrec_t* h, int which, const double* v, const double* vo) {
    const int ntau = h->nmax + h->ntype, nfs = h->nslots + 1, nset = h->hoh ? 2 : 1;
    const size_t B = 2 * (size_t)BLK;
    std::vector<const double*> blk((size_t)nset * ntau * nfs, nullptr);
    std::vector<double> neg((size_t)h->ntype * h->nslots * B, 0.0), ident(B, 0.0);
    for (int d = 0; d < NB; ++d) ident[2 * (d + NB * d)] = 1.0;
    for (int t = 0; t < h->ntype; ++t)
        for (int s = 0; s < h->nslots; ++s) {
            blk[((size_t)0 * ntau + h->nmax + t) * nfs + s] = v + B * (s + (size_t)h->hslots * t);
            if (nset > 1 && s >= 1) {
                double* d = neg.data() + B * (s + (size_t)h->nslots * t);
                const double* src = vo + B * (s + (size_t)h->hslots * t);
                for (size_t e = 0; e < B; ++e) d[e] = -src[e];
                blk[((size_t)1 * ntau + h->nmax + t) * nfs + s] = d;
            }
        }
    if (nset > 1)
        for (int t = 0; t < h->ntype; ++t) blk[((size_t)1 * ntau + h->nmax + t) * nfs + h->nslots] = ident.data();
    const char* msg = h->kubo_op[which].build_custom(h->nslots, ntau, nset, blk);
    if (msg) return fail(h, RSREC_ERR_DEVICE, "rsrec_kubo_moments: %s", msg);
    return RSREC_OK;
}

// h restricted to the bulk atoms (psi1 of velo_hoh_vec_matmul :727-741 is only formed for k > nmax)
int build_kubo_hbulk(rsrec_t* h) {
    const int ntau = h->nmax + h->ntype, nfs = h->nslots + 1;
    const size_t B = 2 * (size_t)BLK;
    std::vector<const double*> blk((size_t)ntau * nfs, nullptr);
    for (int t = 0; t < h->ntype; ++t)
        for (int s = 0; s < h->nslots; ++s) blk[((size_t)h->nmax + t) * nfs + s] = h->host_ee.data() + B * (s + (size_t)h->hslots * t);
    const char* msg = h->kubo_hbulk.build_custom(h->nslots, ntau, 1, blk);
    if (msg) return fail(h, RSREC_ERR_DEVICE, "rsrec_kubo_moments: %s", msg);
    return RSREC_OK;
}

struct KuboCtx {
    rsrec_t* h;
    SpmmDims SD;
    ChainView CV;
    dim3 grid;
    const int* iz;
    double *hps, *p1, *p2;      // temporaries of the two-pass products
    std::vector<std::pair<hipEvent_t, hipEvent_t>>* spmm_ev = nullptr;   // timing of every SpMM launch (rsrec_kubo_moments)
    std::vector<std::tuple<const Spmm5Operator*, int, double>>* req = nullptr;
};

// flops one whole-lattice product with operator (op, set) requires by its block structure (see required_hop_flops)
double kubo_required_flops(const rsrec_t* h, const Spmm5Operator& op, int set) {
    double f = 0.0;
    const int ns = h->nslots;
    for (int i = 0; i < h->kk; ++i) {
        const int tau = i < h->nmax ? i : h->nmax + h->iz0[i];
        for (int s2 = 0; s2 < ns; ++s2)
            if (h->nbr[(size_t)i * ns + s2] >= 0) f += op.required_flops(set, tau, s2);
        f += op.required_flops(set, tau, ns);               // the extra on-site slot of two-input passes (0 if the class has none)
    }
    return f;
}

void kubo_spmm(const KuboCtx& K, const Spmm5Operator& op, int set, const double* in, double* out, const double* in2, S5Epilogue epi = S5Epilogue()) {
    rsrec_t* h = K.h;
    hipEvent_t e0 = K.spmm_ev ? next_event(h) : nullptr;
    if (in2) launch_s5<true>(h, K.grid, K.SD, K.CV.order, K.CV.cum, K.iz, op, set, in, out, in2, nullptr, 0, epi);
    else launch_s5<false>(h, K.grid, K.SD, K.CV.order, K.CV.cum, K.iz, op, set, in, out, nullptr, nullptr, 0, epi);
    if (K.spmm_ev) {
        K.spmm_ev->emplace_back(e0, next_event(h));
        double f = -1.0;                                  // required flops of (op, set): computed once per call
        for (auto& e : *K.req) if (std::get<0>(e) == &op && std::get<1>(e) == set) f = std::get<2>(e);
        if (f < 0.0) { f = kubo_required_flops(h, op, set); K.req->emplace_back(&op, set, f); }
        h->n_req_flop += f * K.SD.nchains;
        h->n_kubo_chain_launches += K.SD.nchains;
    }
}
// out = H in   (ham_vec_matmul :913 / ham_hoh_vec_matmul :785 before their scale-and-shift), or with an epilogue the whole Chebyshev
// step  out = (H in - b in)/a  [* 2 - old]  (their scale-and-shift :968-970 and the caller's recurrence :1132-1136) in the same kernel
void kubo_apply_h(const KuboCtx& K, const double* in, double* out, S5Epilogue epi = S5Epilogue()) {
    if (!K.h->hoh) { kubo_spmm(K, K.h->s5_op, 0, in, out, nullptr, epi); return; }
    kubo_spmm(K, K.h->s5_op, 0, in, K.hps, nullptr);
    kubo_spmm(K, K.h->s5_op, 1, K.hps, out, in, epi);
}
S5Epilogue cheb_epilogue(bool first, const double* cur, const double* old, double a, double b) {
    S5Epilogue E; E.kind = first ? 1 : 2; E.cur = cur; E.old = first ? nullptr : old; E.a = a; E.b = b;
    return E;
}
// out = V in   (velo_vec_matmul :587 / velo_hoh_vec_matmul :656)
void kubo_apply_v(const KuboCtx& K, const Spmm5Operator& vop, const double* in, double* out) {
    if (!K.h->hoh) { kubo_spmm(K, vop, 0, in, out, nullptr); return; }
    kubo_spmm(K, vop, 0, in, K.p2, nullptr);
    kubo_spmm(K, K.h->nmax > 0 ? K.h->kubo_hbulk : K.h->s5_op, 0, in, K.p1, nullptr);
    kubo_spmm(K, vop, 1, K.p1, out, K.p2);
}

}  // namespace

// compute_moments_stochastic (recursion.f90:979-1234):  mu(:,:,n,m,i) = sum_k [T_{m-1}(H~) r_i]_k^H [v_a T_{n-1}(H~) v_b r_i]_k,
// H~ = (H - b)/a.  The SpMMs are k_spmm5 over all atoms (blocks outside the reference's growing region are exact zeros); the
// cond_ll x cond_ll moment contraction of a vector is one complex GEMM  L^H R  over the (atom, row) index (rocBLAS zgemm).
extern "C" int rsrec_kubo_moments(rsrec_t* h, int nvec, int nseed, const int32_t* seed_atoms, const double* seed_coef, int cond_ll, double a, double b,
                                  const double* v_a, const double* vo_a, const double* v_b, const double* vo_b, double* mu_nm) {
    int rc = check_ready(h, "rsrec_kubo_moments");
    if (rc) return rc;
    if (nvec < 0 || nseed < 1 || cond_ll < 1 || a == 0.0 || !v_a || !v_b || !mu_nm || (nvec > 0 && (!seed_atoms || !seed_coef)))
        return fail(h, RSREC_ERR_ARG, "rsrec_kubo_moments: bad argument");
    if (h->hoh && (!vo_a || !vo_b)) return fail(h, RSREC_ERR_ARG, "rsrec_kubo_moments: hoh requires vo_a and vo_b");
    if (!h->s5_built) return fail(h, RSREC_ERR_ARG, "rsrec_kubo_moments: lattice has too many neighbour slots for the SpMM kernel");
    for (size_t q = 0; q < (size_t)nvec * nseed; ++q)
        if (seed_atoms[q] < 0 || seed_atoms[q] > h->kk) return fail(h, RSREC_ERR_ARG, "rsrec_kubo_moments: seed atom %d outside 0..%d", seed_atoms[q], h->kk);
    HIPCK(h, hipSetDevice(h->device));
    reset_timing(h);
    if (nvec == 0) return RSREC_OK;
    rc = build_kubo_operator(h, 0, v_a, vo_a); if (rc) return rc;
    rc = build_kubo_operator(h, 1, v_b, vo_b); if (rc) return rc;
    if (h->hoh && h->nmax > 0) { rc = build_kubo_hbulk(h); if (rc) return rc; }
    const int kk = h->kk;
    const size_t velems = (size_t)(kk + 1) * BLD, nd = (size_t)kk * BLD;
    const int nchunk = std::min(cond_ll, 64);                // right vectors per contraction
    // The moment contraction L^H R is k_kubo_gram on the vectors where they lie (CI layout = dense row-major (18 kk) x 18 matrices side by
    // side): the slots of Lm / Rm ARE the left / right vectors, written there by the SpMMs themselves.
    // device memory: 11 work vectors, `lchunk` left vectors, one chunk of right vectors, the slices' partial blocks, one vector's moments.
    // The left matrix is held in chunks of `lchunk` vectors (all of them if they fit: cond_ll x kk x 5184 B is 21 GB for cond_ll = 500
    // on 8 000 atoms, 252 GB on 10^5): each chunk continues the left recurrence where the previous one stopped and is contracted
    // with ALL right vectors, so the right recurrence (2 of the 3 SpMMs per moment order) is repeated once per chunk.
    size_t free_b = 0, total_b = 0;
    HIPCK(h, hipMemGetInfo(&free_b, &total_b));
    size_t reusable = 0;
    for (int v = 0; v < 6; ++v) reusable += h->d_vec[v].bytes;
    for (auto& kb : h->d_kubo) reusable += kb.bytes;             // the buffers of the previous call (reused where they are large enough)
    const double budget = 0.9 * (double)(free_b + reusable);
    const int ksteps_total = (int)((NB * (size_t)kk + 3) / 4);            // k-steps of 4 rows (the last one may end inside the zero block)
    const int nbn_max = (nchunk * NB + KG_BLK - 1) / KG_BLK;
    // slices of the row index per contraction: enough wave tasks for a few rounds of the device, at least 64 k-steps per task
    auto ksplit_for = [&](int lc) {
        const long blocks = (long)((lc * NB + KG_BLK - 1) / KG_BLK) * nbn_max;
        long ksp = (12L * 8 * h->n_cu + blocks - 1) / std::max(1L, blocks);          // up to a dozen rounds of the device's wave slots
        ksp = std::min<long>({ksp, 64, std::max(1, ksteps_total / 64)});
        return (int)std::max<long>(8, (ksp + 7) / 8 * 8);
    };
    auto part_bytes = [&](int lc) { return (double)ksplit_for(lc) * ((lc * NB + KG_BLK - 1) / KG_BLK) * KG_BLK * (double)nbn_max * KG_BLK * 16.0; };
    // Vectors in flight: the vectors of a call are independent (recursion.f90:1104: one pass of the loop each) and one whole-lattice product is
    // kk / 8 groups -- 1 000 on 8 000 atoms, half a round of the device's wave slots per spin.  Up to 8 of them advance together as the CHAINS of
    // every launch (chain c of a buffer slot lies c vectors behind chain 0, exactly like the sites of a recursion batch); each keeps its own
    // left / right matrices and is contracted by itself.  A whole left matrix per vector goes first: vectors are added only while it fits.
    auto need_for = [&](int lc, int nv) { return (11.0 + lc + nchunk) * nv * velems * 8 + part_bytes(lc) + (double)nv * cond_ll * cond_ll * BLK * 16.0; };
    int lchunk = cond_ll;
    if (h->opt_kubo_lchunk > 0) lchunk = (int)std::min<long>(cond_ll, h->opt_kubo_lchunk);
    int nbv = (int)std::min<long>(nvec, h->opt_kubo_vbatch > 0 ? h->opt_kubo_vbatch : 8);
    while (nbv > 1 && need_for(lchunk, nbv) > budget) --nbv;
    while (lchunk > 1 && need_for(lchunk, nbv) > budget) lchunk = (lchunk + 1) / 2;
    if (need_for(lchunk, nbv) > budget)
        return fail(h, RSREC_ERR_DEVICE, "rsrec_kubo_moments: %.1f GB needed for one left vector at a time on %d atoms, %.1f GB free", need_for(1, 1) * 1e-9, kk, free_b * 1e-9);
    for (int v = 0; v < 6; ++v) h->d_vec[v].release();
    const size_t sstride = (size_t)nbv * velems;                                // doubles between two slots of a buffer (nbv chains each)
    DevBuf &work = h->d_kubo[0], &Lm = h->d_kubo[1], &Rm = h->d_kubo[2], &Part = h->d_kubo[3], &Mu = h->d_kubo[4];
    auto cleanup = [&]() {};                                      // the buffers stay with the handle for the next call
    {   // buffers that have to grow are given back first, so that the new sizes are asked of the memory the budget counted on
        const size_t want[5] = {11 * sstride * 8, (size_t)lchunk * sstride * 8, (size_t)nchunk * sstride * 8, (size_t)part_bytes(lchunk), (size_t)nbv * cond_ll * cond_ll * BLK * 16};
        for (int q = 0; q < 5; ++q) if (h->d_kubo[q].bytes < want[q]) h->d_kubo[q].release();
    }
    if (work.reserve(11 * sstride * 8) != hipSuccess || Lm.reserve((size_t)lchunk * sstride * 8) != hipSuccess || Rm.reserve((size_t)nchunk * sstride * 8) != hipSuccess ||
        Part.reserve((size_t)part_bytes(lchunk)) != hipSuccess || Mu.reserve((size_t)nbv * cond_ll * cond_ll * BLK * 16) != hipSuccess) {
        for (auto& kb : h->d_kubo) kb.release();
        return fail(h, RSREC_ERR_DEVICE, "rsrec_kubo_moments: out of device memory");
    }
    HIPCK(h, hipMemsetAsync(Lm.p, 0, (size_t)lchunk * sstride * 8, h->stream));    // (block kk of every slot stays the zero block)
    HIPCK(h, hipMemsetAsync(Rm.p, 0, (size_t)nchunk * sstride * 8, h->stream));
    HIPCK(h, hipMemsetAsync(work.p, 0, 11 * sstride * 8, h->stream));            // block kk of every vector stays the zero block
    double* V[11];
    for (int v = 0; v < 11; ++v) V[v] = work.as<double>() + (size_t)v * sstride;
    double *psiref = V[0], *w0 = V[1], *w1 = V[2], *w2 = V[3];
    double *l0 = V[4], *l1 = V[9];                                              // T_{m0-2} r, T_{m0-1} r: the left recurrence across a chunk border
    HIPCK(h, h->d_seed.reserve((size_t)nseed * 4));
    HIPCK(h, h->d_seedcoef.reserve((size_t)nseed * sizeof(double2)));
    // region list: all atoms (every launch of this path runs over the whole lattice), one row shared by the chains of a launch
    std::vector<int> all(kk);
    for (int i = 0; i < kk; ++i) all[i] = i;
    int ostride = kk;
    double dummy1 = 0, dummy2 = 0;
    rc = upload_regions(h, all.data(), 1, kk, 1, 1, false, true, ostride, dummy1, dummy2);
    if (rc) { cleanup(); return rc; }
    KuboCtx K;
    K.h = h;
    K.CV.order = h->cur_order; K.CV.cum = h->cur_cum; K.CV.obase = h->cur_cum + (size_t)h->cur_nrows * 1; K.CV.nlev = 1; K.CV.vstride = velems; K.CV.cpo = nbv; K.CV.ostride = ostride;
    K.iz = h->d_iz.as<int>();
    K.hps = V[6]; K.p1 = V[7]; K.p2 = V[8];
    std::vector<std::pair<hipEvent_t, hipEvent_t>> spmm_ev, gemm_ev;
    K.spmm_ev = &spmm_ev;
    std::vector<std::tuple<const Spmm5Operator*, int, double>> req_tab;
    K.req = &req_tab;
    h->n_kubo_chain_launches = 0;
    hipEvent_t e_begin = next_event(h);
    int n_left_chunks = 0;
    auto Lslot = [&](int q) { return Lm.as<double>() + (size_t)q * sstride; };
    auto Rslot = [&](int q) { return Rm.as<double>() + (size_t)q * sstride; };
    for (int iv0 = 0; iv0 < nvec; iv0 += nbv) {
        const int nb = std::min(nbv, nvec - iv0);                                 // vectors of this batch = chains of its launches
        K.SD = SpmmDims{kk, h->nslots, h->nmax, 1, nbv, ostride, 0, velems, K.CV.obase, nb};
        K.grid = s5_grid(h, dim3(256, (unsigned)nb), 0);
        // r_i: psiref(l,l,seed(k)) = coef(k); seed atom 0 = unused entry
        HIPCK(h, hipMemsetAsync(psiref, 0, sstride * 8, h->stream));
        for (int c = 0; c < nb; ++c) {
            const int iv = iv0 + c;
            std::vector<int> s0; std::vector<double> c0;
            for (int k = 0; k < nseed; ++k) {
                const int at = seed_atoms[(size_t)iv * nseed + k];
                if (at == 0) continue;
                s0.push_back(at - 1); c0.push_back(seed_coef[2 * ((size_t)iv * nseed + k)]); c0.push_back(seed_coef[2 * ((size_t)iv * nseed + k) + 1]);
            }
            if (s0.empty()) { cleanup(); return fail(h, RSREC_ERR_ARG, "rsrec_kubo_moments: vector %d has no seed", iv + 1); }
            XFER(xfer_h2d(h, h->d_seed.p, s0.data(), s0.size() * 4));
            XFER(xfer_h2d(h, h->d_seedcoef.p, c0.data(), c0.size() * 8));
            k_seed<LayoutCI><<<1, 64, 0, h->stream>>>(psiref + (size_t)c * velems, velems, h->d_seed.as<int>(), h->d_seedcoef.as<double2>(), (int)s0.size());
            HIPCK(h, hipStreamSynchronize(h->stream));                            // the seed tables are reused by the next vector
        }
        const size_t cpy = ((size_t)(nb - 1) * velems + nd) * 8;                  // the chains of a slot, up to the last one's zero block
        for (int m0 = 0; m0 < cond_ll; m0 += lchunk) {
            const int ml = std::min(lchunk, cond_ll - m0), m_rows = ml * NB;
            ++n_left_chunks;
            // left vectors  T_{m-1}(H~) r,  m = m0 .. m0 + ml - 1 (recursion.f90:1120-1142), each written by its SpMM into its slot of Lm;
            // the recurrence reads the two slots before it -- across a chunk border the copies l0 = T_{m0-2} r, l1 = T_{m0-1} r
            for (int m = m0; m < m0 + ml; ++m) {
                double* out = Lslot(m - m0);
                const double* prev1 = m - 1 >= m0 ? Lslot(m - 1 - m0) : l1;
                const double* prev2 = m - 2 >= m0 ? Lslot(m - 2 - m0) : (m - 2 == m0 - 1 ? l1 : l0);
                if (m == 0) HIPCK(h, hipMemcpyAsync(out, psiref, cpy, hipMemcpyDeviceToDevice, h->stream));
                else if (m == 1) kubo_apply_h(K, prev1, out, cheb_epilogue(true, prev1, nullptr, a, b));
                else kubo_apply_h(K, prev1, out, cheb_epilogue(false, prev1, prev2, a, b));
            }
            if (m0 + ml < cond_ll) {                                          // state for the next chunk (its slots are about to be reused)
                if (ml >= 2) HIPCK(h, hipMemcpyAsync(l0, Lslot(ml - 2), cpy, hipMemcpyDeviceToDevice, h->stream));
                else HIPCK(h, hipMemcpyAsync(l0, l1, cpy, hipMemcpyDeviceToDevice, h->stream));
                HIPCK(h, hipMemcpyAsync(l1, Lslot(ml - 1), cpy, hipMemcpyDeviceToDevice, h->stream));
            }
            // right vectors  v_a T_{n-1}(H~) v_b r  (:1154-1187), written into the slots of Rm and contracted with the left vectors of
            // this chunk, 64 at a time
            double *y0 = w0, *y1 = w1, *y2 = w2;
            kubo_apply_v(K, h->kubo_op[1], psiref, y1);                       // v1 = v0 = v_b r
            for (int n = 0; n < cond_ll; ++n) {
                if (n == 1) {
                    std::swap(y0, y1);
                    kubo_apply_h(K, y0, y1, cheb_epilogue(true, y0, nullptr, a, b));
                } else if (n > 1) {
                    kubo_apply_h(K, y1, y2, cheb_epilogue(false, y1, y0, a, b));
                    double* o = y0; y0 = y1; y1 = y2; y2 = o;
                }
                const int nl = n % nchunk;
                kubo_apply_v(K, h->kubo_op[0], y1, Rslot(nl));
                if (nl == nchunk - 1 || n == cond_ll - 1) {
                    const int ncols = (nl + 1) * NB, n0 = n - nl;
                    // C[(m,c)][(n,c')] = sum_{k,r} conj(L_m[(k,r)][c]) R_n[(k,r)][c']: blocks of C x `ksplit` slices of (k,r), one wave each.
                    const int bmr = KG_BLK, bnc = KG_BLK;
                    const int nbm = (m_rows + bmr - 1) / bmr, nbn = (ncols + bnc - 1) / bnc;
                    // slices: the multiple of 8 (<= what the partial buffer was sized for, >= 64 k-steps per task) that fills whole rounds of
                    // the device's wave slots best -- 361 blocks x 24 slices are 4.2 rounds of 2 048 slots (85 % of the last round idle), x 32 are 5.6
                    int ksplit = 8;
                    {
                        const long slots = 8L * h->n_cu, cap = std::min<long>(ksplit_for(lchunk), std::max(8, ksteps_total / 64 / 8 * 8));
                        double best = 0.0;
                        for (long c = 8; c <= cap; c += 8) {
                            const long tasks = (long)nbm * nbn * c, rounds = (tasks + slots - 1) / slots;
                            const double eff = (double)tasks / (double)(rounds * slots) * (rounds >= 3 ? 1.0 : 0.9);   // (few rounds: the tail of the slowest wave shows)
                            if (eff > best + 1e-9) { best = eff; ksplit = (int)c; }
                        }
                    }
                    HIPCK(h, hipGetLastError());
                    hipEvent_t g0 = next_event(h);
                    const unsigned wgs = 8u * (unsigned)(((long)nbm * nbn * (ksplit / 8) + 3) / 4);
                    // one contraction per vector of the batch: its matrices are the chain-c columns of the slots (leading dimension = a whole slot)
                    for (int c = 0; c < nb; ++c) {
                        k_kubo_gram<<<wgs, 256, 0, h->stream>>>(Lm.as<double>() + (size_t)c * velems, sstride, m_rows, Rm.as<double>() + (size_t)c * velems, sstride, ncols, ksteps_total, ksplit, Part.as<double2>(), nbm, nbn);
                        k_kubo_gram_reduce<<<std::min(4096, (m_rows * ncols + 255) / 256), 256, 0, h->stream>>>(Part.as<double2>(), ksplit, nbm * bmr, nbn * bnc, m_rows, ncols, Mu.as<double2>() + (size_t)c * cond_ll * cond_ll * BLK, cond_ll, m0, n0);
                    }
                    gemm_ev.emplace_back(g0, next_event(h));
                }
            }
        }
        HIPCK(h, hipGetLastError());
        XFER(xfer_d2h(h, mu_nm + 2 * (size_t)BLK * cond_ll * cond_ll * iv0, Mu.p, (size_t)nb * cond_ll * cond_ll * BLK * 16));
    }
    hipEvent_t e_end = next_event(h);
    HIPCK(h, hipStreamSynchronize(h->stream));
    HIPCK(h, hipGetLastError());
    h->t_total_ms = ev_ms(e_begin, e_end);
    for (auto& pr : spmm_ev) h->t_hop_ms += ev_ms(pr.first, pr.second);          // the SpMM kernels (H and velocity products)
    for (auto& pr : gemm_ev) h->t_rest_ms += ev_ms(pr.first, pr.second);         // "rest_ms" here: the moment contractions (k_kubo_gram + its reduction)
    h->n_hop_launch = (double)spmm_ev.size();
    h->n_kubo_left_chunks = n_left_chunks;
    // work in the reference's terms: every product is over the whole lattice -- one block multiply per (atom, present slot)
    {
        double fan = 0.0;
        for (int i = 0; i < kk; ++i)
            for (int s2 = 0; s2 < h->nslots; ++s2) if (h->nbr[(size_t)i * h->nslots + s2] >= 0) fan += 1.0;
        h->n_block_mult = fan * h->n_kubo_chain_launches;
        h->n_atom_steps = (double)kk * h->n_kubo_chain_launches;
    }
    cleanup();
    return RSREC_OK;
}

namespace {

// h as ham_vec_matmul applies it (recursion.f90:913-977): per-type blocks ee (per-atom hall for the impurity region) with l.s added to
// the on-site block -- whatever hamiltonian%hoh says.  Without hoh that is set 0 of s5_op; with hoh a table of its own.
int build_plain_operator(rsrec_t* h) {
    const int ntau = h->nmax + h->ntype, nfs = h->nslots + 1;
    const size_t B = 2 * (size_t)BLK;
    std::vector<const double*> blk((size_t)ntau * nfs, nullptr);
    std::vector<double> onsite((size_t)ntau * B);
    for (int tau = 0; tau < ntau; ++tau) {
        const int ty = tau < h->nmax ? h->iz0[tau] : tau - h->nmax;
        const double* base = tau < h->nmax ? h->host_hall.data() + B * (size_t)h->hslots * tau : h->host_ee.data() + B * (size_t)h->hslots * (tau - h->nmax);
        for (size_t e = 0; e < B; ++e) onsite[(size_t)tau * B + e] = base[e] + h->host_lsham[B * ty + e];
        blk[(size_t)tau * nfs] = onsite.data() + (size_t)tau * B;
        for (int s = 1; s < h->nslots; ++s) blk[(size_t)tau * nfs + s] = base + B * s;
    }
    const char* msg = h->orb_plain.build_custom(h->nslots, ntau, 1, blk);
    if (msg) return fail(h, RSREC_ERR_DEVICE, "plain operator table: %s", msg);
    return RSREC_OK;
}

}  // namespace

// chebyshev_orbital_mod (recursion.f90:2834-3049), the moment part (:2893-3013), device-resident: the seeds are chains advanced together.
// For seed atom s:  psiref = 1 on s;  left = i (Y H~ X - X H~ Y) psiref  (X, Y = alat cr(1,:), alat cr(2,:); H~ = ham_vec_matmul, the
// plain operator also when hoh is set);  v_1 = psiref, v_2 = H~' v_1, v_n = 2 H~' v_{n-1} - v_{n-2}  (H~' = ham_hoh_vec_matmul with hoh);
// mu(:,:,n) = sum_k left_k^H v_n,k.  Every product runs over the whole lattice (the reference sets izero = 1, :2920).
//   mu_orb  complex (18,18,lld): the SUM over the seeds of the call, added in seed order (the reference loops over all kk atoms and
//           divides by kk afterwards, :3006);  mu_seed (optional) complex (18,18,lld,nseeds): every seed's contribution.
extern "C" int rsrec_orbital_moments(rsrec_t* h, int nseeds, const int32_t* seed_atoms, int lld, double a, double b, const double* cr, double alat,
                                     double* mu_orb, double* mu_seed) {
    int rc = check_ready(h, "rsrec_orbital_moments");
    if (rc) return rc;
    if (nseeds < 0 || lld < 1 || a == 0.0 || !cr || !mu_orb || (nseeds > 0 && !seed_atoms)) return fail(h, RSREC_ERR_ARG, "rsrec_orbital_moments: bad argument");
    for (int q = 0; q < nseeds; ++q)
        if (seed_atoms[q] < 1 || seed_atoms[q] > h->kk) return fail(h, RSREC_ERR_ARG, "rsrec_orbital_moments: seed atom %d outside 1..%d", seed_atoms[q], h->kk);
    if (!h->s5_built) return fail(h, RSREC_ERR_ARG, "rsrec_orbital_moments: lattice has too many neighbour slots for the SpMM kernel");
    HIPCK(h, hipSetDevice(h->device));
    reset_timing(h);
    std::fill(mu_orb, mu_orb + 2 * (size_t)BLK * lld, 0.0);
    if (nseeds == 0) return RSREC_OK;
    const int kk = h->kk;
    const bool hoh = h->hoh != 0;
    if (hoh) { rc = build_plain_operator(h); if (rc) return rc; }
    const Spmm5Operator& plain = hoh ? h->orb_plain : h->s5_op;
    const size_t velems = (size_t)(kk + 1) * BLD, nd = (size_t)kk * BLD;
    const int nvec = hoh ? 5 : 4;                                  // left, v0, v1, v2 (+ h v of the two-pass product)
    BatchPlan bp;
    rc = plan_batch(h, nseeds, nvec, velems / 2, bp);
    if (rc) return rc;
    const int B = bp.batch;
    for (int v = 0; v < nvec; ++v) HIPCK(h, h->d_vec[v].reserve((size_t)B * velems * sizeof(double)));
    const size_t gram_elems = (size_t)B * 256 * 1296;
    HIPCK(h, h->d_partial.reserve(2 * gram_elems * sizeof(double)));
    h->p2_slot = (size_t)B * 16 * 2 * 1296;
    HIPCK(h, h->d_partial2.reserve(2 * h->p2_slot * sizeof(double)));
    HIPCK(h, h->d_seed.reserve((size_t)B * 4));
    HIPCK(h, h->d_seedcoef.reserve((size_t)B * sizeof(double2)));
    HIPCK(h, h->d_scal.reserve((size_t)3 * kk * sizeof(double)));
    HIPCK(h, h->d_zsqr.reserve((size_t)B * lld * BLK * sizeof(double2)));           // the chains' moments
    XFER(xfer_h2d(h, h->d_scal.p, cr, (size_t)3 * kk * sizeof(double)));
    std::vector<int> all(kk);
    for (int i = 0; i < kk; ++i) all[i] = i;
    int ostride = kk;
    double dummy1 = 0, dummy2 = 0;
    rc = upload_regions(h, all.data(), 1, kk, 1, 1, false, true, ostride, dummy1, dummy2);
    if (rc) return rc;
    double2* d_out = h->d_zsqr.as<double2>();
    const size_t ostr = (size_t)lld * BLK;
    std::vector<double> host_out;
    hipEvent_t e_begin = next_event(h);
    std::vector<std::pair<hipEvent_t, hipEvent_t>> spmm_ev;
    std::vector<std::tuple<const Spmm5Operator*, int, double>> req_tab;
    for (int c0 = 0; c0 < nseeds; c0 += B) {
        const int nb = std::min(B, nseeds - c0);
        std::vector<int> s0(nb);
        std::vector<double> one(2 * (size_t)nb, 0.0);
        for (int q = 0; q < nb; ++q) { s0[q] = seed_atoms[c0 + q] - 1; one[2 * (size_t)q] = 1.0; }
        XFER(xfer_h2d(h, h->d_seed.p, s0.data(), s0.size() * 4));
        XFER(xfer_h2d(h, h->d_seedcoef.p, one.data(), one.size() * 8));
        KuboCtx K;
        K.h = h;
        K.CV.order = h->cur_order; K.CV.cum = h->cur_cum; K.CV.obase = h->cur_cum + (size_t)h->cur_nrows * 1; K.CV.nlev = 1; K.CV.vstride = velems; K.CV.cpo = nb; K.CV.ostride = ostride;
        K.SD = SpmmDims{kk, h->nslots, h->nmax, 1, nb, ostride, 0, velems, K.CV.obase, nb};
        K.grid = s5_grid(h, dim3(256, nb), 0);
        K.iz = h->d_iz.as<int>();
        K.spmm_ev = &spmm_ev; K.req = &req_tab;
        double* left = h->d_vec[0].as<double>();
        double *v0 = h->d_vec[1].as<double>(), *v1 = h->d_vec[2].as<double>(), *v2 = h->d_vec[3].as<double>();
        K.hps = hoh ? h->d_vec[4].as<double>() : nullptr; K.p1 = nullptr; K.p2 = nullptr;
        for (int v = 0; v < nvec; ++v) HIPCK(h, hipMemsetAsync(h->d_vec[v].p, 0, (size_t)nb * velems * sizeof(double), h->stream));
        k_seed<LayoutCI><<<nb, 64, 0, h->stream>>>(v1, velems, h->d_seed.as<int>(), h->d_seedcoef.as<double2>(), 1);     // v_1 = psiref
        // t = H~ psiref with the plain operator, then the position factors
        kubo_spmm(K, plain, 0, v1, left, nullptr, cheb_epilogue(true, v1, nullptr, a, b));
        k_orb_left<<<dim3(std::min(kk, 1024), nb), 256, 0, h->stream>>>(kk, velems, h->d_seed.as<int>(), h->d_scal.as<double>(), alat, reinterpret_cast<double2*>(left));
        const dim3 grid_mf(std::max(1, std::min(mfma_workgroups_per_chain(h, nb), (ostride / GROUP + MF_WAVES - 1) / MF_WAVES)), nb);
        const dim3 gl = level_grid(h, grid_mf, 0);
        for (int n = 0; n < lld; ++n) {
            if (n == 1) {
                std::swap(v0, v1);
                kubo_apply_h(K, v0, v1, cheb_epilogue(true, v0, nullptr, a, b));
            } else if (n > 1) {
                kubo_apply_h(K, v1, v2, cheb_epilogue(false, v1, v0, a, b));
                double* o = v0; v0 = v1; v1 = v2; v2 = o;
            }
            k_mfma_adot<<<gl, MF_WAVES * 64, 0, h->stream>>>(K.CV, 0, kk, left, v1, h->d_partial.as<double>());
            int n2 = gl.x;
            const double* p2 = presum(h, h->d_partial.as<double>(), nb, n2, 1296);
            k_reduce_gram_out<<<nb, 1024, 0, h->stream>>>(p2, n2, d_out + (size_t)n * BLK, ostr, 1);
        }
        HIPCK(h, hipGetLastError());
        host_out.resize((size_t)nb * ostr * 2);
        XFER(xfer_d2h(h, host_out.data(), d_out, host_out.size() * sizeof(double)));
        for (int q = 0; q < nb; ++q) {                                // seed order, like the reference's loop (:2893)
            const double* src = host_out.data() + (size_t)q * ostr * 2;
            for (size_t e = 0; e < ostr * 2; ++e) mu_orb[e] += src[e];
            if (mu_seed) memcpy(mu_seed + (size_t)(c0 + q) * ostr * 2, src, ostr * 2 * sizeof(double));
        }
    }
    hipEvent_t e_end = next_event(h);
    HIPCK(h, hipStreamSynchronize(h->stream));
    h->t_total_ms = ev_ms(e_begin, e_end);
    for (auto& pr : spmm_ev) h->t_hop_ms += ev_ms(pr.first, pr.second);
    h->t_rest_ms = h->t_total_ms - h->t_hop_ms;
    h->n_hop_launch = (double)spmm_ev.size();
    h->res_kind = 0;
    (void)nd;
    return RSREC_OK;
}

// ham_vec_matmul / ham_hoh_vec_matmul (recursion.f90:913 / :785): psi_out = (H psi_in - b psi_in) / a on whole vectors
// psi(18,18,kk) in the reference's layout (host arrays); with vel = 1: velo_vec_matmul / velo_hoh_vec_matmul (:587 / :656) with
// the operator blocks v_op (and vo_op with hoh), no scaling.
extern "C" int rsrec_apply_operator(rsrec_t* h, int vel, const double* v_op, const double* vo_op, const double* psi_in, double* psi_out, double a, double b) {
    int rc = check_ready(h, "rsrec_apply_operator");
    if (rc) return rc;
    if (!psi_in || !psi_out || (vel != 1 && a == 0.0) || (vel == 1 && (!v_op || (h->hoh && !vo_op))) || vel < 0 || vel > 2) return fail(h, RSREC_ERR_ARG, "rsrec_apply_operator: bad argument");
    if (!h->s5_built) return fail(h, RSREC_ERR_ARG, "rsrec_apply_operator: lattice has too many neighbour slots for the SpMM kernel");
    HIPCK(h, hipSetDevice(h->device));
    reset_timing(h);
    const int kk = h->kk;
    const size_t velems = (size_t)(kk + 1) * BLD, nd = (size_t)kk * BLD;
    if (vel == 1) { rc = build_kubo_operator(h, 0, v_op, vo_op); if (rc) return rc; if (h->hoh && h->nmax > 0) { rc = build_kubo_hbulk(h); if (rc) return rc; } }
    if (vel == 2 && h->hoh) { rc = build_plain_operator(h); if (rc) return rc; }        // ham_vec_matmul under hoh: the plain operator (recursion.f90:913)
    for (int v = 0; v < 6; ++v) HIPCK(h, h->d_vec[v].reserve(velems * 8));
    for (int v = 0; v < 6; ++v) HIPCK(h, hipMemsetAsync(h->d_vec[v].p, 0, velems * 8, h->stream));
    std::vector<int> all(kk);
    for (int i = 0; i < kk; ++i) all[i] = i;
    int ostride = kk;
    double dummy1 = 0, dummy2 = 0;
    rc = upload_regions(h, all.data(), 1, kk, 1, 1, false, true, ostride, dummy1, dummy2);
    if (rc) return rc;
    KuboCtx K;
    K.h = h;
    K.CV.order = h->cur_order; K.CV.cum = h->cur_cum; K.CV.obase = h->cur_cum + (size_t)h->cur_nrows * 1; K.CV.nlev = 1; K.CV.vstride = velems; K.CV.cpo = 1; K.CV.ostride = ostride;
    K.SD = SpmmDims{kk, h->nslots, h->nmax, 1, 1, ostride, 0, velems, K.CV.obase, 1};
    K.grid = s5_grid(h, dim3(256, 1), 0);
    K.iz = h->d_iz.as<int>();
    double* in = h->d_vec[0].as<double>(); double* out = h->d_vec[1].as<double>(); double* tmp = h->d_vec[2].as<double>();
    K.hps = h->d_vec[3].as<double>(); K.p1 = h->d_vec[4].as<double>(); K.p2 = h->d_vec[5].as<double>();
    XFER(xfer_h2d(h, tmp, psi_in, nd * 8));
    hipEvent_t e0 = next_event(h);
    k_block_transpose<true><<<std::min(kk, 2048), 384, 0, h->stream>>>(kk, reinterpret_cast<const double2*>(tmp), reinterpret_cast<double2*>(in));
    if (vel == 1) kubo_apply_v(K, h->kubo_op[0], in, out);
    else if (vel == 2 && h->hoh) {
        kubo_spmm(K, h->orb_plain, 0, in, tmp, nullptr);
        k_cheb_combine<true><<<(int)std::min<size_t>(4096, (nd + 255) / 256), 256, 0, h->stream>>>(nd, tmp, in, nullptr, out, a, b);
    } else {
        kubo_apply_h(K, in, tmp);
        k_cheb_combine<true><<<(int)std::min<size_t>(4096, (nd + 255) / 256), 256, 0, h->stream>>>(nd, tmp, in, nullptr, out, a, b);
    }
    k_block_transpose<false><<<std::min(kk, 2048), 384, 0, h->stream>>>(kk, reinterpret_cast<const double2*>(out), reinterpret_cast<double2*>(tmp));
    hipEvent_t e1 = next_event(h);
    HIPCK(h, hipGetLastError());
    XFER(xfer_d2h(h, psi_out, tmp, nd * 8));
    HIPCK(h, hipStreamSynchronize(h->stream));
    h->t_total_ms = ev_ms(e0, e1);
    return RSREC_OK;
}

extern "C" int rsrec_scalar_lanczos(rsrec_t* h, int nsites, const int32_t* seed_atoms, int lld, int llmax, double* a, double* b2) {
    int rc = check_ready(h, "rsrec_scalar_lanczos");
    if (rc) return rc;
    if (nsites < 0 || lld < 1 || llmax < lld || !a || !b2 || (nsites > 0 && !seed_atoms)) return fail(h, RSREC_ERR_ARG, "rsrec_scalar_lanczos: bad argument");
    for (int q = 0; q < nsites; ++q)
        if (seed_atoms[q] < 1 || seed_atoms[q] > h->kk) return fail(h, RSREC_ERR_ARG, "rsrec_scalar_lanczos: seed atom %d outside 1..%d", seed_atoms[q], h->kk);
    HIPCK(h, hipSetDevice(h->device));
    reset_timing(h);
    std::fill(a, a + (size_t)llmax * NB * nsites, 0.0);
    std::fill(b2, b2 + (size_t)llmax * NB * nsites, 0.0);
    if (nsites == 0) return RSREC_OK;
    if (h->nsp != 1) {
        // hop() is a no-op unless nsp = 1 (recursion.f90:3326-3415): a stays 0 and crecal (:3466) divides by sqrt(0),
        // so the reference returns b2 = (1, 0, 0, NaN, NaN, ...).  Mirrored, not "fixed".
        for (size_t q = 0; q < (size_t)NB * nsites; ++q) {
            b2[q * llmax] = 1.0;
            for (int ll = 3; ll < lld; ++ll) b2[q * llmax + ll] = std::nan("");
        }
        return RSREC_OK;
    }
    const int kk = h->kk;
    const int nsteps = lld - 1;
    const int nlev = nsteps + 1;
    const size_t velems = (size_t)kk * NB;
    const int B = (int)std::min<long>(nsites, h->opt_batch > 0 ? h->opt_batch : 16);
    const int nblk = (int)std::min<long>(std::max<long>(1, (kk + TILE_ATOMS - 1) / TILE_ATOMS), 64);
    const int nch = B * NB;
    for (int v = 0; v < 2; ++v) HIPCK(h, h->d_vec[v].reserve((size_t)nch * velems * sizeof(double2)));
    HIPCK(h, h->d_scal.reserve((size_t)nch * (nblk + 2 * (size_t)lld) * sizeof(double) + 64));
    HIPCK(h, h->d_seed.reserve((size_t)nch * 2 * 4));
    double2* psi = h->d_vec[0].as<double2>();
    double2* pmn = h->d_vec[1].as<double2>();
    double* part = h->d_scal.as<double>();
    double* ca = part + (size_t)nch * nblk;
    double* cb = ca + (size_t)nch * lld;
    const DevProblem P = make_problem(h);
    hipEvent_t ev_begin = next_event(h);
    std::vector<double> ha((size_t)nch * lld), hb((size_t)nch * lld);
    for (int c0 = 0; c0 < nsites; c0 += B) {
        const int nb = std::min(B, nsites - c0);
        const int nc = nb * NB;
        std::vector<int> seeds0(nb), so((size_t)nc * 2);
        for (int q = 0; q < nb; ++q) {
            seeds0[q] = seed_atoms[c0 + q] - 1;
            for (int l = 0; l < NB; ++l) { so[2 * (q * NB + l)] = seeds0[q]; so[2 * (q * NB + l) + 1] = l; }
        }
        double dummy1 = 0, dummy2 = 0;
        int ostride = kk;
        rc = upload_regions(h, seeds0.data(), nb, 1, nlev, nsteps, false, false, ostride, dummy1, dummy2);
        if (rc) return rc;
        h->n_atom_steps += dummy1 * NB;
        XFER(xfer_h2d(h, h->d_seed.p, so.data(), so.size() * 4));
        HIPCK(h, hipStreamSynchronize(h->stream));
        ChainView CV;
        CV.order = h->cur_order; CV.cum = h->cur_cum; CV.obase = h->cur_cum + (size_t)h->cur_nrows * nlev; CV.nlev = nlev; CV.vstride = velems; CV.cpo = NB; CV.ostride = ostride;
        for (int v = 0; v < 2; ++v) HIPCK(h, hipMemsetAsync(h->d_vec[v].p, 0, (size_t)nc * velems * sizeof(double2), h->stream));
        HIPCK(h, hipMemsetAsync(ca, 0, (size_t)nch * 2 * lld * sizeof(double), h->stream));
        k_scalar_seed<<<nc, 64, 0, h->stream>>>(psi, velems, h->d_seed.as<int>(), cb, lld);
        const dim3 grid(nblk, nc);
        for (int ll = 0; ll < nsteps; ++ll) {
            k_scalar_hop<<<grid, NTHREADS, 0, h->stream>>>(P, CV, ll + 1, psi, pmn, part);
            k_scalar_reduce<<<nc, 64, 0, h->stream>>>(part, nblk, ca, lld, ll);
            k_scalar_orth<<<grid, NTHREADS, 0, h->stream>>>(kk, CV, ll + 1, psi, pmn, ca, lld, ll, part);
            k_scalar_reduce<<<nc, 64, 0, h->stream>>>(part, nblk, cb, lld, ll + 1);
            k_scalar_update<<<grid, NTHREADS, 0, h->stream>>>(kk, CV, ll + 1, psi, pmn, cb, lld, ll);
            h->n_hop_launch += 1;
        }
        HIPCK(h, hipGetLastError());
        XFER(xfer_d2h(h, ha.data(), ca, (size_t)nc * lld * sizeof(double)));
        XFER(xfer_d2h(h, hb.data(), cb, (size_t)nc * lld * sizeof(double)));
        HIPCK(h, hipStreamSynchronize(h->stream));
        for (int q = 0; q < nc; ++q)
            for (int ll = 0; ll < lld; ++ll) {
                a[(size_t)llmax * ((size_t)c0 * NB + q) + ll] = ha[(size_t)q * lld + ll];
                b2[(size_t)llmax * ((size_t)c0 * NB + q) + ll] = hb[(size_t)q * lld + ll];
            }
    }
    hipEvent_t ev_end = next_event(h);
    HIPCK(h, hipStreamSynchronize(h->stream));
    h->t_total_ms = ev_ms(ev_begin, ev_end);
    return RSREC_OK;
}
