// api.hip — C ABI of libwlsqm_hip.so (see include/wlsqm_hip.h for the contract and the
// reference file:line each entry point replaces).  Host logic only; kernels live in
// fit_lane.hip / fit_tile.hip / fit_wave.hip.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <ctime>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "wlsqm_internal.hpp"
#include "hostio.hpp"

namespace wlsqm {

static thread_local std::string g_err;
static thread_local const char* g_kernel = "";
void set_error(const std::string& msg) { g_err = msg; }
int hip_fail(hipError_t e, const char* what) {
    g_err = std::string("HIP error: ") + hipGetErrorString(e) + " in " + what;
    if (e == hipErrorOutOfMemory) return WLSQM_EMEMORY;
    if (e == hipErrorNoDevice || e == hipErrorInvalidDevice) return WLSQM_ENODEVICE;
    return WLSQM_ERUNTIME;
}
const char* last_kernel_name() { return g_kernel; }
void note_kernel(const char* name) { g_kernel = name; }

int launch_fit_lane(int dimension, int order, const KParams& p, hipStream_t stream);
int launch_fit_tile(int dimension, int order, const KParams& p, long long max_nk, hipStream_t stream, bool* handled);
int launch_fit_wave(int dimension, int order, const KParams& p, hipStream_t stream);
int launch_fit_moment(int dimension, int order, const KParams& p, long long max_nk, hipStream_t stream, bool* handled);
int launch_fit_ring(int dimension, int order, const KParams& p, long long max_nk, hipStream_t stream, bool* handled);
int launch_fit_ring_gather(int dimension, int order, const KParams& p, long long max_nk, hipStream_t stream, bool* handled);
int launch_fit_tilek(int dimension, int order, const KParams& p, long long max_nk, hipStream_t stream, bool* handled);
long long preferred_slots(int dimension, int order, long long max_nk);
int launch_fit_rows(int dimension, int order, const KParams& p, hipStream_t stream, bool* handled);
int launch_fit_chunk(int dimension, int order, const KParams& p, long long K, hipStream_t stream, bool* handled);
int launch_fit_chunk_refine(int dimension, int order, const KParams& p, long long K, hipStream_t stream, bool* handled);
int launch_fit_sens(int dimension, int order, const KParams& p, long long K, hipStream_t stream, bool* handled);
int launch_fit_strict(int dimension, int order, const KParams& p, const StrictDebug* dbg, hipStream_t stream);
int launch_fit_stage(int dimension, int order, const KParams& p, long long K, hipStream_t stream, bool* handled);
int launch_fit_stage_refine(int dimension, int order, const KParams& p, long long K, hipStream_t stream, bool* handled);

// Numerics mode of the calling thread: 0 = the fast kernels, 1 = reference-order arithmetic (fit_strict.hip), 2 = accurate
// (fit_accurate.hip: reference-order arithmetic with the normal matrix assembled from its upper triangle).  The first use on a
// thread takes WLSQM_HIP_STRICT from the environment (unset / 0, 1, 2 or "accurate"); wlsqm_hip_set_strict() overrides it.
static thread_local int g_strict = -1;
// what the calling thread says about the neighbour counts of the dense DEVICE-resident batches it hands over (wlsqm_hip_set_row_hint; the host
// entry points look at the counts themselves): 1 every case fills its row (default), 2 ragged, 0 unknown (the kernels find out)
static thread_local int g_row_hint = 1;
static thread_local int g_order_hint = 1;                            // the neighbours of a row: 1 sorted by distance (default: a k-nearest-neighbour search's rows), 0 in no order
int row_hint_value() { return g_row_hint; }
int order_hint_value() { return g_order_hint; }
static int strict_mode_value();
bool accurate_mode() { return strict_mode_value() == 2; }
bool strict_mode() { return strict_mode_value() >= 1; }
static int strict_mode_value() {
    if (g_strict < 0) {
        const char* e = getenv("WLSQM_HIP_STRICT");
        g_strict = (!e || !e[0] || e[0] == '0') ? 0 : ((e[0] == '2' || e[0] == 'a' || e[0] == 'A') ? 2 : 1);
    }
    return g_strict;
}

// Dense rows the tiled kernels cannot take as they are — a strided neighbour or case axis, rows that are not multiples of 16
// bytes (odd K), misaligned bases — are repacked on the device into contiguous [ncases, K', dim] / [ncases, K'] scratch
// (K' = preferred_slots: even, and a moment size for 2D order 4) in front of the same kernels: two extra passes over xk and fk
// instead of the lane-per-case kernel (8.5 % of the HBM peak on C2).  Only slots k < K are read; the pad slot replays slot 0 and
// is masked by nk like every unused slot.
__global__ void repack_rows_kernel(const KParams p, int dim, long long K, long long Kp, double* __restrict__ xk, double* __restrict__ fk,
                                   int* __restrict__ nkc) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= p.ncases * Kp) return;
    const long long j = t / Kp; const long long k = t - j * Kp;
    const long long ks = k < K ? k : 0;
    const double* src = p.xk + j * p.sxk_j + ks * p.sxk_k;
    for (int m = 0; m < dim; ++m) xk[t * dim + m] = src[m];
    fk[t] = p.fk[j * p.sfk_j + ks * p.sfk_k];
    // the LOGICAL neighbour count stays clamped to the caller's K: the kernels behind the repack clamp to the slot count Kp,
    // which may be K + 1 (a bad nk[j] > K must not count the pad slot, nor write a sens row for it)
    if (k == 0) nkc[j] = (int)min((long long)p.nk[j * p.snk], K);
}

// Index-based rows (S / F / hoods) of a shape without an index-based tile kernel: gathered once into dense scratch rows
// (slots k >= nk[j] are never dereferenced: they replay the case's own point and stay masked).
__global__ void gather_rows_kernel(const KParams p, int dim, long long K, long long Kp, double* __restrict__ xk, double* __restrict__ fk,
                                   double* __restrict__ xi, int* __restrict__ nkc) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= p.ncases * Kp) return;
    const long long j = t / Kp; const long long k = t - j * Kp;
    const long long pj = own_point(p, j);                                 // without point_index, case j of the batch sits at point pbase + j
    const long long idx = (k < K && k < p.nk[j * p.snk]) ? (long long)p.hoods[j * p.shoods_j + k] : pj;
    for (int m = 0; m < dim; ++m) xk[t * dim + m] = p.S[idx * dim + m];
    fk[t] = p.F[idx];
    if (k == 0) {
        for (int m = 0; m < dim; ++m) xi[j * dim + m] = p.S[pj * dim + m];
        nkc[j] = (int)min((long long)p.nk[j * p.snk], K);
    }
}

// Scratch of the repack / gather passes is bounded: the batch goes through in slices of at most this many bytes of dense rows
// (WLSQM_HIP_REPACK_MB, default 512), so a 16M-case odd-K or index-based call needs O(slice) extra memory, not a second copy
// of the batch.  The slices run back to back on the stream and reuse the same block.
static size_t repack_slice_bytes() {
    const char* e = getenv("WLSQM_HIP_REPACK_MB");
    const long long mb = e ? atoll(e) : 512;
    return (size_t)(mb > 0 ? mb : 512) << 20;
}

// Runs the batch through `stage` (repack or gather into dense scratch rows) + the tiled kernels, slice by slice.
// Returns WLSQM_OK with *done = false when the scratch could not be allocated: the caller falls through to the kernels that
// need none (chunk / lane), as before this path existed.
static int fit_through_dense_scratch(int dimension, int order, const KParams& p, long long max_nk, hipStream_t stream, bool gather,
                                     bool* done) {
    *done = false;
    const long long Kp = preferred_slots(dimension, order, max_nk);
    const size_t per_case = ((size_t)Kp * (dimension + 1) + (gather ? dimension : 0)) * sizeof(double) + sizeof(int) + 8;
    long long per_slice = (long long)(repack_slice_bytes() / per_case);
    per_slice = std::max(4096LL, per_slice - per_slice % 64);
    const long long ns = std::min(per_slice, p.ncases);
    const size_t nx = (size_t)ns * Kp * dimension, nf = (size_t)ns * Kp, ni = gather ? (size_t)ns * dimension : 0;
    double* ws = nullptr;
    int rc = scratch_alloc_async(reinterpret_cast<void**>(&ws), (nx + nf + ni) * sizeof(double) + (size_t)ns * sizeof(int) + 16, stream);
    if (rc != WLSQM_OK) {
        if (rc == WLSQM_EMEMORY) { (void)hipGetLastError(); set_error(""); return WLSQM_OK; }      // no scratch: not an error
        return rc;
    }
    int* nkc = reinterpret_cast<int*>(ws + nx + nf + ni);
    // (slices of a batch must all take the same kernel family, or fast-mode bits would depend on WLSQM_HIP_REPACK_MB and on
    // ncases modulo the slice size: a tail shorter than 256 cases — the threshold below which the dispatcher prefers the lane
    // kernel to a repack — is cut from the last two slices' sum in two halves instead)
    for (long long j0 = 0; j0 < p.ncases && rc == WLSQM_OK;) {
        long long n = std::min(ns, p.ncases - j0);
        const long long rest = p.ncases - j0 - n;
        if (rest > 0 && rest < 256 && n > 512) n -= 256;           // leave the next (last) slice at least 256 cases
        const KParams src = slice_cases(p, j0, n);
        const long long threads = n * Kp;
        if (gather)
            hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, stream, src, dimension, max_nk, Kp,
                               ws, ws + nx, ws + nx + nf, nkc);
        else
            hipLaunchKernelGGL(repack_rows_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, stream, src, dimension, max_nk, Kp,
                               ws, ws + nx, nkc);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) { rc = hip_fail(e, gather ? "gather_rows_kernel" : "repack_rows_kernel"); break; }
        KParams q = src;
        q.xk = ws; q.sxk_j = Kp * dimension; q.sxk_k = dimension;
        q.fk = ws + nx; q.sfk_j = Kp; q.sfk_k = 1;
        q.nk = nkc; q.snk = 1;
        if (gather) {
            q.hoods = nullptr; q.S = nullptr; q.F = nullptr; q.pidx = nullptr; q.shoods_j = 0;
            q.xi = ws + nx + nf; q.sxi_j = dimension;
        }
        rc = launch_fit(dimension, order, q, Kp, stream);          // contiguous now: takes the tiled kernels
        j0 += n;
    }
    const int rc2 = scratch_free_async(ws, stream);
    *done = true;
    return rc != WLSQM_OK ? rc : rc2;
}

static bool dense_layout_ok(int dim, const KParams& p, long long K) {
    if (p.sxk_k != dim || p.sxk_j != K * dim || p.sfk_k != 1 || p.sfk_j != K || (K % 2) != 0) return false;
    return ((reinterpret_cast<uintptr_t>(p.xk) | reinterpret_cast<uintptr_t>(p.fk)) & 15u) == 0;
}

int launch_fit(int dimension, int order, const KParams& p_in, long long max_nk, hipStream_t stream) {
    KParams p = p_in;
    p.max_nk = max_nk;
    const int no = wlsqm_hip_number_of_dofs(dimension, order);
    if (no < 0) { set_error("bad dimension/order"); return WLSQM_EVALUE; }
    if (p.ncases <= 0) return WLSQM_OK;
    if (p.hoods && no > 15 && (p.do_sens || p.iterative)) {
        if (!strict_mode()) {
            set_error("index-based input with sensitivities or refinement supports systems with at most 15 DOFs");
            return WLSQM_EVALUE;
        }
    }
    if (strict_mode()) return launch_fit_strict(dimension, order, p, nullptr, stream);   // any layout, any shape, all extras
    {
        const char* off = getenv("WLSQM_HIP_DISABLE_TILE");
        const char* norp = getenv("WLSQM_HIP_DISABLE_REPACK");
        const bool tiles_on = !(off && off[0] == '1') && !(norp && norp[0] == '1');
        if (tiles_on && !p.hoods && !p.case_index && p.xk && p.fk && no <= 15 && max_nk >= 2 && p.ncases >= 256 &&
            !dense_layout_ok(dimension, p, max_nk)) {
            bool done = false;
            const int rc = fit_through_dense_scratch(dimension, order, p, max_nk, stream, /*gather=*/false, &done);
            if (rc != WLSQM_OK || done) return rc;
        }
    }
    bool handled = false;
    int rc = launch_fit_stage(dimension, order, p, max_nk, stream, &handled);     // one lane per case, rows staged through LDS (round 4)
    if (rc != WLSQM_OK || handled) return rc;
    rc = launch_fit_stage_refine(dimension, order, p, max_nk, stream, &handled);  // the same mapping with the refinement sweeps (fit_stage_iter.hip)
    if (rc != WLSQM_OK || handled) return rc;
    rc = launch_fit_ring(dimension, order, p, max_nk, stream, &handled);          // one-kernel fit of the 15-unknown systems
    if (rc != WLSQM_OK || handled) return rc;
    rc = launch_fit_ring_gather(dimension, order, p, max_nk, stream, &handled);   // the same on index-based input (2D order 4)
    if (rc != WLSQM_OK || handled) return rc;
    rc = launch_fit_moment(dimension, order, p, max_nk, stream, &handled);
    if (rc != WLSQM_OK || handled) return rc;
    rc = launch_fit_tile(dimension, order, p, max_nk, stream, &handled);
    if (rc != WLSQM_OK || handled) return rc;
    rc = launch_fit_tilek(dimension, order, p, max_nk, stream, &handled);
    if (rc != WLSQM_OK || handled) return rc;
    rc = launch_fit_sens(dimension, order, p, max_nk, stream, &handled);     // sensitivities of the shapes without a tile kernel
    if (rc != WLSQM_OK || handled) return rc;
    if (no <= 15) {
        rc = launch_fit_chunk_refine(dimension, order, p, max_nk, stream, &handled);   // refinement of the 10- / 15-unknown systems
        if (rc != WLSQM_OK || handled) return rc;
    }
    {
        // index-based input no tiled kernel took: gather it into dense rows and dispatch again (the dense tables are complete)
        const char* off = getenv("WLSQM_HIP_DISABLE_TILE");
        const char* norp = getenv("WLSQM_HIP_DISABLE_REPACK");
        const bool tiles_on = !(off && off[0] == '1') && !(norp && norp[0] == '1');
        if (tiles_on && p.hoods && !p.case_index && no <= 15 && max_nk >= 2 && p.ncases >= 256) {
            bool done = false;
            rc = fit_through_dense_scratch(dimension, order, p, max_nk, stream, /*gather=*/true, &done);
            if (rc != WLSQM_OK || done) return rc;
        }
    }
    if (no <= 15) {
        rc = launch_fit_chunk(dimension, order, p, max_nk, stream, &handled);   // any K: neighbours through LDS in chunks, two passes
        if (rc != WLSQM_OK || handled) return rc;
        return launch_fit_lane(dimension, order, p, stream);
    }
    rc = launch_fit_rows(dimension, order, p, stream, &handled);        // basic fit of the 3D order-3/4 systems
    if (rc != WLSQM_OK || handled) return rc;
    return launch_fit_wave(dimension, order, p, stream);
}

int check_device(int device) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        (void)hipGetLastError();
        set_error("no HIP device available: libwlsqm_hip has no CPU fallback");
        return WLSQM_ENODEVICE;
    }
    if (device < 0 || device >= n) { set_error("invalid device ordinal"); return WLSQM_ENODEVICE; }
    WLSQM_HIP_CHECK(hipSetDevice(device));
    return WLSQM_OK;
}

int scratch_alloc_async(void** out, size_t bytes, hipStream_t stream) {
    static hipMemPool_t pools[16] = {};
    static std::mutex pools_mutex;                               // the first calls of two threads must not both create the pool
    int dev = 0;
    WLSQM_HIP_CHECK(hipGetDevice(&dev));
    if (dev < 0 || dev >= 16) { set_error("device ordinal out of range"); return WLSQM_EVALUE; }
    std::lock_guard<std::mutex> lock(pools_mutex);
    if (!pools[dev]) {
        hipMemPoolProps props{};
        props.allocType = hipMemAllocationTypePinned;
        props.handleTypes = hipMemHandleTypeNone;
        props.location.type = hipMemLocationTypeDevice;
        props.location.id = dev;
        hipMemPool_t pool;
        WLSQM_HIP_CHECK(hipMemPoolCreate(&pool, &props));
        // freed blocks up to this much stay in the pool across synchronisations (the next call gets them back without a device
        // synchronisation); anything above goes back to the device at the next synchronisation, so that one large call does
        // not pin its scratch until the process exits (torch's allocator cannot see this pool)
        uint64_t keep = 1ull << 30;
        WLSQM_HIP_CHECK(hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &keep));
        pools[dev] = pool;
    }
    *out = nullptr;
    hipError_t e = hipMallocFromPoolAsync(out, bytes, pools[dev], stream);
    if (e != hipSuccess) return hip_fail(e, "hipMallocFromPoolAsync");
    return WLSQM_OK;
}
namespace {
struct StreamBuffer { int dev; hipStream_t stream; void* p; size_t bytes; unsigned long long used; };
std::vector<StreamBuffer>& stream_buffer_table() { static std::vector<StreamBuffer> t; return t; }
std::mutex& stream_buffer_mutex() { static std::mutex m; return m; }
unsigned long long g_stream_buffer_clock = 0;
}  // namespace
int call_scratch_acquire(CallScratch* cs, size_t bytes, hipStream_t stream) {
    cs->p = nullptr; cs->pooled = false;
    int dev = 0;
    WLSQM_HIP_CHECK(hipGetDevice(&dev));
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(stream, &cap) != hipSuccess) { (void)hipGetLastError(); cap = hipStreamCaptureStatusNone; }
    if (cap == hipStreamCaptureStatusNone) {
        // (one lock for the table and for the launches around the buffer: a call holds it for the few microseconds its enqueues take)
        std::unique_lock<std::mutex> lock(stream_buffer_mutex());
        auto& tab = stream_buffer_table();
        StreamBuffer* hit = nullptr;
        for (auto& e : tab) if (e.dev == dev && e.stream == stream) hit = &e;
        if (!hit && tab.size() >= 64) {
            // a process that keeps creating streams: the entry used longest ago goes.  Its stream may be gone: hipFree (which waits for the
            // device) rather than a stream-ordered free
            size_t old = 0;
            for (size_t i = 1; i < tab.size(); ++i) if (tab[i].used < tab[old].used) old = i;
            (void)hipFree(tab[old].p);
            tab.erase(tab.begin() + (long)old);
        }
        if (!hit) { tab.push_back(StreamBuffer{dev, stream, nullptr, 0, 0}); hit = &tab.back(); }
        if (hit->bytes < bytes) {
            const size_t want = bytes < 65536 ? 65536 : bytes + bytes / 2;
            void* p = nullptr;
            hipError_t e = hipMalloc(&p, want);
            if (e != hipSuccess) return hip_fail(e, "hipMalloc (call scratch)");
            if (hit->p) (void)hipFreeAsync(hit->p, stream);          // behind whatever still reads the old one on this stream
            hit->p = p; hit->bytes = want;
        }
        hit->used = ++g_stream_buffer_clock;
        cs->p = hit->p;
        cs->lock = std::move(lock);
        return WLSQM_OK;
    }
    cs->pooled = true;
    return scratch_alloc_async(&cs->p, bytes, stream);
}
int call_scratch_release(CallScratch* cs, hipStream_t stream) {
    int rc = WLSQM_OK;
    if (cs->pooled) rc = scratch_free_async(cs->p, stream);
    cs->p = nullptr; cs->pooled = false;
    if (cs->lock.owns_lock()) cs->lock.unlock();
    return rc;
}
int scratch_free_async(void* p, hipStream_t stream) {
    if (!p) return WLSQM_OK;
    WLSQM_HIP_CHECK(hipFreeAsync(p, stream));
    return WLSQM_OK;
}

static int validate_batch(const wlsqm_batch* b) {
    if (!b) { set_error("null batch"); return WLSQM_EVALUE; }
    if (b->dimension < 1 || b->dimension > 3) { set_error("dimension must be 1, 2 or 3"); return WLSQM_EVALUE; }
    if (b->ncases < 1) { set_error("max_cases must be >= 1"); return WLSQM_EVALUE; }   // infra.pyx:311-313
    if (!b->xk || !b->fk || !b->nk || !b->xi || !b->fi || !b->order || !b->knowns || !b->weighting_method) {
        set_error("null array in batch"); return WLSQM_EVALUE;
    }
    if (b->max_nk < 0) { set_error("max_nk must be >= 0"); return WLSQM_EVALUE; }
    return WLSQM_OK;
}

static KParams params_from(const wlsqm_batch* b) {
    KParams p{};
    p.xk = b->xk; p.sxk_j = b->xk_stride_case; p.sxk_k = b->xk_stride_k;
    p.fk = b->fk; p.sfk_j = b->fk_stride_case; p.sfk_k = b->fk_stride_k;
    p.nk = b->nk; p.snk = b->nk_stride;
    p.xi = b->xi; p.sxi_j = b->xi_stride_case;
    p.fi = b->fi; p.sfi_j = b->fi_stride_case;
    p.sens = (b->do_sens ? b->sens : nullptr); p.ss_j = b->sens_stride_case; p.ss_k = b->sens_stride_k;
    p.knowns = (const long long*)b->knowns; p.sknowns = b->knowns_stride;
    p.wm = b->weighting_method; p.swm = b->wm_stride;
    p.case_index = nullptr; p.ncases = b->ncases;
    p.do_sens = (b->do_sens && b->sens) ? 1 : 0;
    p.iterative = b->iterative ? 1 : 0;
    p.max_iter = b->max_iter;
    p.iters_out = nullptr;
    p.ragged = row_hint_value();                                    // (device-resident counts: the caller's word; the host entry points overwrite it with what they see)
    p.rows_sorted = order_hint_value();
    return p;
}

}  // namespace wlsqm

using namespace wlsqm;

extern "C" {

const char* wlsqm_hip_last_error(void) { return g_err.c_str(); }
const char* wlsqm_hip_last_kernel(void) { return last_kernel_name(); }

int wlsqm_hip_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return n;
}

int wlsqm_hip_set_strict(int mode) {
    const int prev = strict_mode_value();
    g_strict = mode == 2 ? 2 : (mode ? 1 : 0);
    return prev;
}
int wlsqm_hip_get_strict(void) { return strict_mode_value(); }

int wlsqm_hip_set_row_hint(int hint) {
    const int prev = g_row_hint;
    g_row_hint = (hint == 0 || hint == 2) ? hint : 1;
    return prev;
}
int wlsqm_hip_set_order_hint(int sorted) {
    const int prev = g_order_hint;
    g_order_hint = sorted ? 1 : 0;
    return prev;
}

int wlsqm_hip_number_of_dofs(int dimension, int order) {
    if (dimension < 1 || dimension > 3) return -1;
    if (order < 0 || order > 4) return -2;
    static const int tab[3][5] = {{1, 2, 3, 4, 5}, {1, 3, 6, 10, 15}, {1, 4, 10, 20, 35}};
    return tab[dimension - 1][order];
}

int wlsqm_hip_number_of_reduced_dofs(int n, int64_t mask) {
    return n - __builtin_popcountll((unsigned long long)mask);
}

int wlsqm_hip_remap(int32_t* o2r, int32_t* r2o, int n, int64_t mask) {
    int k = 0;
    for (int j = 0; j < n; ++j) {
        if (mask & (1LL << j)) o2r[j] = -1;
        else o2r[j] = k++;
    }
    for (int j = 0; j < n; ++j)
        if (o2r[j] != -1) r2o[o2r[j]] = j;
    for (int j = k; j < n; ++j) r2o[j] = -1;
    return k;
}

int wlsqm_hip_fit_many_device(const wlsqm_batch* b, int device, void* stream, int order_uniform,
                              const int64_t* case_index, int64_t ncases_sel, int32_t* iterations_out) {
    int rc = validate_batch(b);
    if (rc != WLSQM_OK) return rc;
    DeviceScope scope;
    rc = scope.enter(device);
    if (rc != WLSQM_OK) return rc;
    if (order_uniform < 0 || order_uniform > 4) { set_error("order_uniform must be 0..4"); return WLSQM_EVALUE; }
    hipStream_t s = (hipStream_t)stream;
    KParams p = params_from(b);
    if (case_index) { p.case_index = (const long long*)case_index; p.ncases = ncases_sel; }
    // The iteration counter is only kept when the caller asks for it: without iterations_out the call enqueues kernels and
    // nothing else (no allocation, no host synchronisation: legal inside a stream capture).
    int* d_it = nullptr;
    if (b->iterative && iterations_out) {
        rc = scratch_alloc_async(reinterpret_cast<void**>(&d_it), sizeof(int), s); if (rc != WLSQM_OK) return rc;
        WLSQM_HIP_CHECK(hipMemsetAsync(d_it, 0, sizeof(int), s));
        p.iters_out = d_it;
    }
    rc = launch_fit(b->dimension, order_uniform, p, b->max_nk, s);
    if (rc != WLSQM_OK) { (void)scratch_free_async(d_it, s); return rc; }
    if (iterations_out) {
        *iterations_out = 0;
        if (d_it) {
            WLSQM_HIP_CHECK(hipMemcpyAsync(iterations_out, d_it, sizeof(int), hipMemcpyDeviceToHost, s));
            WLSQM_HIP_CHECK(hipStreamSynchronize(s));
            // a case without unknowns is a no-op for the kernels, but the reference's loop still runs for it and stops at its
            // second pass (impl.pyx:1026-1081): the maximum over the cases is never below 1
            if (*iterations_out < 1) *iterations_out = 1;
            rc = scratch_free_async(d_it, s);
        }
    }
    return rc;
}

}  // extern "C"

namespace wlsqm {
// Order buckets on the device: idx[o * n + pos] = case number, counts[o] = cases of order o.  STABLE (round 4): a bucket lists its
// cases in ascending case number, so the tiles a case shares with others — and with them every wave-uniform code path choice of the
// kernels — are a function of the batch alone: the same call gives the same bits run after run (round 3 handed the positions out
// by atomicAdd: bucket order, tile mates and so, for the 14- / 15-unknown systems, a case's last bits varied between runs).
// Three passes, no host synchronisation: (1) per block of OB_BLOCK cases the five counts; (2) one workgroup per order scans the
// block counts; (3) rank inside the block by wave ballots.  Cases whose order is not 0..max_order are left out of every bucket.
constexpr int OB_BLOCK = 256;
__device__ __forceinline__ int ob_order_of(const int* __restrict__ order, long long sorder, long long n, long long t, int max_order) {
    if (t >= n) return -1;
    const int o = order[t * sorder];
    return (o < 0 || o > max_order) ? -1 : o;         // (the host entry points raise ValueError; here the case is left untouched)
}
__global__ __launch_bounds__(OB_BLOCK) void order_count_kernel(const int* __restrict__ order, long long sorder, long long n, int max_order,
                                                                long long nb, long long* __restrict__ blk) {   // blk[o * nb + block]
    __shared__ int cnt[5];
    if (threadIdx.x < 5) cnt[threadIdx.x] = 0;
    __syncthreads();
    const long long t = (long long)blockIdx.x * OB_BLOCK + threadIdx.x;
    const int o = ob_order_of(order, sorder, n, t, max_order);
#pragma unroll
    for (int q = 0; q < 5; ++q) {
        const unsigned long long m = __ballot(o == q);
        if ((threadIdx.x & 63) == 0 && m) atomicAdd(&cnt[q], __popcll(m));       // integer adds: order-free
    }
    __syncthreads();
    if (threadIdx.x < 5) blk[(long long)threadIdx.x * nb + blockIdx.x] = cnt[threadIdx.x];
}
// one workgroup per order: exclusive scan of blk[o][0..nb) in place, total -> counts[o]
__global__ __launch_bounds__(1024) void order_scan_kernel(long long nb, long long* __restrict__ blk, long long* __restrict__ counts) {
    __shared__ long long part[1024];
    long long* row = blk + (long long)blockIdx.x * nb;
    const long long per = (nb + 1023) / 1024, b0 = (long long)threadIdx.x * per, b1 = (b0 + per < nb) ? b0 + per : nb;
    long long sum = 0;
    for (long long b = b0; b < b1; ++b) sum += row[b];
    part[threadIdx.x] = sum;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {        // inclusive Hillis-Steele over the 1 024 partial sums
        const long long v = (threadIdx.x >= (unsigned)off) ? part[threadIdx.x - off] : 0;
        __syncthreads();
        part[threadIdx.x] += v;
        __syncthreads();
    }
    long long run = part[threadIdx.x] - sum;
    for (long long b = b0; b < b1; ++b) { const long long c = row[b]; row[b] = run; run += c; }
    if (threadIdx.x == 1023) counts[blockIdx.x] = part[1023];
}
__global__ __launch_bounds__(OB_BLOCK) void order_scatter_kernel(const int* __restrict__ order, long long sorder, long long n, int max_order,
                                                                  long long nb, const long long* __restrict__ blk, long long* __restrict__ idx) {
    __shared__ int wcnt[OB_BLOCK / 64][5];
    const long long t = (long long)blockIdx.x * OB_BLOCK + threadIdx.x;
    const int o = ob_order_of(order, sorder, n, t, max_order);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    int below = 0;                                     // cases of my order in the lower lanes of my wave
#pragma unroll
    for (int q = 0; q < 5; ++q) {
        const unsigned long long m = __ballot(o == q);
        if (lane == 0) wcnt[wave][q] = __popcll(m);
        if (o == q) below = __popcll(m & ((1ull << lane) - 1ull));
    }
    __syncthreads();
    if (o < 0) return;
    for (int w2 = 0; w2 < wave; ++w2) below += wcnt[w2][o];
    idx[(long long)o * n + blk[(long long)o * nb + blockIdx.x] + below] = t;
}
}  // namespace wlsqm

extern "C" {

int wlsqm_hip_fit_many_device_orders(const wlsqm_batch* b, int device, void* stream, const int32_t* order_dev, int64_t order_stride,
                                     int max_order, int32_t* iterations_out) {
    int rc = validate_batch(b);
    if (rc != WLSQM_OK) return rc;
    if (!order_dev) { set_error("null order array"); return WLSQM_EVALUE; }
    if (max_order < 0 || max_order > 4) { set_error("max_order must be 0..4"); return WLSQM_EVALUE; }
    DeviceScope scope;
    rc = scope.enter(device);
    if (rc != WLSQM_OK) return rc;
    hipStream_t s = (hipStream_t)stream;
    const long long n = b->ncases;
    const long long nb = (n + OB_BLOCK - 1) / OB_BLOCK;
    long long* ws = nullptr;                          // [8] counts, then [5][n] case numbers, then [5][nb] block counts / offsets
    rc = scratch_alloc_async(reinterpret_cast<void**>(&ws), (size_t)(5 * n + 8 + 5 * nb) * sizeof(long long), s);
    if (rc != WLSQM_OK) return rc;
    long long* blk = ws + 8 + 5 * n;
    int* d_it = nullptr;
    auto cleanup = [&](int code) { (void)scratch_free_async(ws, s); (void)scratch_free_async(d_it, s); return code; };
    if (nb > 0x7fffffffll) { set_error("fit_many_device_orders: batch too large for one launch"); return cleanup(WLSQM_EVALUE); }
    // (kernels only, no hipMemsetAsync: a memset node on memory allocated inside a stream capture aborted the replay on ROCm 7.2;
    // the scan writes all five totals, also for an empty batch)
    hipError_t e = hipSuccess;
    if (nb > 0) {
        hipLaunchKernelGGL(order_count_kernel, dim3((unsigned)nb), dim3(OB_BLOCK), 0, s, order_dev, (long long)order_stride, n, max_order, nb, blk);
        e = hipGetLastError();
        if (e != hipSuccess) return cleanup(hip_fail(e, "order_count_kernel"));
    }
    hipLaunchKernelGGL(order_scan_kernel, dim3(5), dim3(1024), 0, s, nb, blk, ws);
    e = hipGetLastError();
    if (e != hipSuccess) return cleanup(hip_fail(e, "order_scan_kernel"));
    if (nb > 0) {
        hipLaunchKernelGGL(order_scatter_kernel, dim3((unsigned)nb), dim3(OB_BLOCK), 0, s, order_dev, (long long)order_stride, n, max_order, nb,
                           blk, ws + 8);
        e = hipGetLastError();
        if (e != hipSuccess) return cleanup(hip_fail(e, "order_scatter_kernel"));
    }
    KParams p = params_from(b);
    if (b->iterative && iterations_out) {
        rc = scratch_alloc_async(reinterpret_cast<void**>(&d_it), sizeof(int), s);
        if (rc != WLSQM_OK) return cleanup(rc);
        e = hipMemsetAsync(d_it, 0, sizeof(int), s);
        if (e != hipSuccess) return cleanup(hip_fail(e, "hipMemsetAsync"));
        p.iters_out = d_it;
    }
    for (int o = 0; o <= max_order; ++o) {            // buckets above max_order are empty by construction and never launched: a
        // kernel of order o writes no(o) doubles per fi / sens row, and the caller's rows are only promised wide enough for max_order
        if (wlsqm_hip_number_of_dofs(b->dimension, o) < 0) continue;
        p.case_index = ws + 8 + (long long)o * n;
        p.ncases = n;                                 // the launch is sized for the whole batch; the bucket's real size stays on the device
        p.ncases_dev = ws + o;
        rc = launch_fit(b->dimension, o, p, b->max_nk, s);
        if (rc != WLSQM_OK) return cleanup(rc);
    }
    if (iterations_out) {
        *iterations_out = 0;
        if (d_it) {
            e = hipMemcpyAsync(iterations_out, d_it, sizeof(int), hipMemcpyDeviceToHost, s);
            if (e == hipSuccess) e = hipStreamSynchronize(s);
            if (e != hipSuccess) return cleanup(hip_fail(e, "iterations_out"));
            if (*iterations_out < 1) *iterations_out = 1;
        }
    }
    return cleanup(WLSQM_OK);
}

int wlsqm_hip_strict_intermediates_device(const wlsqm_batch* b, int device, void* stream, int order_uniform,
                                          double* w, int64_t w_stride, double* A, double* LU, int64_t mat_stride,
                                          double* row_scale, double* col_scale, int32_t* ipiv, int64_t vec_stride) {
    int rc = validate_batch(b);
    if (rc != WLSQM_OK) return rc;
    DeviceScope scope;
    rc = scope.enter(device);
    if (rc != WLSQM_OK) return rc;
    const int no = wlsqm_hip_number_of_dofs(b->dimension, order_uniform);
    if (no < 0) { set_error("order_uniform must be 0..4"); return WLSQM_EVALUE; }
    if (!w || !A || !LU || !row_scale || !col_scale || !ipiv) { set_error("null output array"); return WLSQM_EVALUE; }
    if (w_stride < b->max_nk || mat_stride < (int64_t)no * no || vec_stride < no) { set_error("output strides too small"); return WLSQM_EVALUE; }
    KParams p = params_from(b);
    p.max_nk = b->max_nk;
    StrictDebug dbg{w, w_stride, A, LU, mat_stride, row_scale, col_scale, vec_stride, ipiv};
    return launch_fit_strict(b->dimension, order_uniform, p, &dbg, (hipStream_t)stream);
}

// Per-thread, per-device transfer context of the host-array entry points (never freed: it must not outlive
// the HIP runtime at process exit, and a thread's buffers are reused by its next call).
struct HostCtx {
    Stager st;
    GrowBuf xk, fk, xi, fi, nk, wm, kn, sens, it, idx;
};
static HostCtx* host_ctx(int device) {
    static thread_local HostCtx* ctx[16] = {nullptr};
    if (device < 0 || device >= 16) return nullptr;
    if (!ctx[device]) ctx[device] = new HostCtx();
    return ctx[device];
}

// Host-array entry point: pack -> device -> kernels -> commit.  Rows are packed by all host threads into pinned
// staging buffers while the previous chunk is in flight (hostio.hpp); the arrays must provide max_nk neighbour
// slots per case (entries k >= nk[j] are transferred but never used).
static double now_s() { timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }

int wlsqm_hip_fit_many_host(const wlsqm_batch* b, int device, int32_t* iterations_out) {
    const bool trace = getenv("WLSQM_HIP_TRACE") != nullptr;
    const double t_start = now_s();
    double t_prev = t_start;
    auto mark = [&](const char* what) {
        if (!trace) return;
        (void)hipDeviceSynchronize();
        const double t = now_s();
        fprintf(stderr, "[wlsqm_hip] %-28s %8.3f ms\n", what, (t - t_prev) * 1e3);
        t_prev = t;
    };
    int rc = validate_batch(b);
    if (rc != WLSQM_OK) return rc;
    const int dim = b->dimension;
    const int64_t n = b->ncases;
    // metadata (also validates order)
    std::vector<int32_t> h_nk(n), h_order(n), h_wm(n), h_no(n);
    std::vector<int64_t> h_kn(n);
    int64_t max_nk = 0; int max_no = 0; bool bad_order = false, bad_nk = false;
#pragma omp parallel for schedule(static) num_threads(copy_threads()) reduction(max : max_nk) reduction(max : max_no) reduction(|| : bad_order) reduction(|| : bad_nk)
    for (int64_t j = 0; j < n; ++j) {
        h_nk[j] = b->nk[j * b->nk_stride];
        h_order[j] = b->order[j * b->order_stride];
        h_wm[j] = b->weighting_method[j * b->wm_stride];
        h_kn[j] = b->knowns[j * b->knowns_stride];
        const int no = wlsqm_hip_number_of_dofs(dim, h_order[j]);
        if (no < 0) { bad_order = true; h_no[j] = 0; continue; }
        if (h_nk[j] < 0) bad_nk = true;
        h_no[j] = no;
        if (h_nk[j] > max_nk) max_nk = h_nk[j];
        if (no > max_no) max_no = no;
    }
    if (bad_order) { set_error("order must be 0, 1, 2, 3 or 4"); return WLSQM_EVALUE; }
    if (bad_nk) { set_error("nk must be >= 0"); return WLSQM_EVALUE; }
    if (b->max_nk > 0 && max_nk > b->max_nk) { set_error("max(nk) exceeds the neighbour axis (max_nk)"); return WLSQM_EVALUE; }
    // RAGGED batches (the reference's own harness: a ball query, nk 30..100 per case — examples/wlsqm_example.py:103-133) are packed in
    // NEIGHBOUR-COUNT order: the staged kernels move the chunks a wave's 64 cases need (fit_stage.hip), so waves of equal counts move no
    // padding.  A stable counting sort on the host, applied while the rows are staged for the upload anyway; results go back to the
    // caller's rows.  (Uniform-order batches only: the order buckets have an index of their own.)  WLSQM_HIP_HOST_NK_ORDER=0: off.
    std::vector<int64_t> perm;
    {
        const char* e = getenv("WLSQM_HIP_HOST_NK_ORDER");
        int32_t min_nk = n > 0 ? h_nk[0] : 0;
        bool same_order = true;
        for (int64_t j = 0; j < n; ++j) { if (h_nk[j] < min_nk) min_nk = h_nk[j]; same_order = same_order && h_order[j] == h_order[0]; }
        if (!(e && e[0] == '0') && same_order && n >= 1024 && max_nk - min_nk >= 8) {
            std::vector<int64_t> start((size_t)max_nk + 2, 0);
            for (int64_t j = 0; j < n; ++j) ++start[(size_t)h_nk[j] + 1];
            for (size_t v = 1; v < start.size(); ++v) start[v] += start[v - 1];
            perm.resize((size_t)n);
            for (int64_t j = 0; j < n; ++j) perm[(size_t)start[(size_t)h_nk[j]]++] = j;
            auto apply = [&](auto& v) { auto w = v; for (int64_t r = 0; r < n; ++r) v[(size_t)r] = w[(size_t)perm[(size_t)r]]; };
            apply(h_nk); apply(h_order); apply(h_wm); apply(h_no); apply(h_kn);
        }
    }
    const int64_t* const pm = perm.empty() ? nullptr : perm.data();
    auto user_row = [&](int64_t r) { return pm ? pm[r] : r; };
    mark("metadata");
    DeviceScope scope;
    rc = scope.enter(device);
    if (rc != WLSQM_OK) return rc;
    HostCtx* cx = host_ctx(device);
    if (!cx) { set_error("device ordinal out of range"); return WLSQM_ENODEVICE; }
    if ((rc = cx->st.ensure(device))) return rc;

    // device rows hold an even number of neighbour slots (16-byte rows for the tiled kernels), for some shapes a few more
    // (the next size with a specialised kernel); pad slots are staged but masked by k < nk[j]
    const bool uniform_order = std::all_of(h_order.begin(), h_order.end(), [&](int o) { return o == h_order[0]; });
    const int64_t K = preferred_slots(dim, uniform_order ? h_order[0] : -1, max_nk);
    const bool want_sens = b->do_sens && b->sens;
    if ((rc = cx->xk.need((size_t)n * K * dim * 8)) || (rc = cx->fk.need((size_t)n * K * 8)) ||
        (rc = cx->xi.need((size_t)n * dim * 8)) || (rc = cx->fi.need((size_t)n * max_no * 8)) ||
        (rc = cx->nk.need(n * 4)) || (rc = cx->wm.need(n * 4)) || (rc = cx->kn.need(n * 8)) || (rc = cx->it.need(4)))
        return rc;
    if (want_sens && (rc = cx->sens.need((size_t)n * K * max_no * 8))) return rc;
    hipStream_t s = nullptr;
    mark("buffers");
    if (max_nk > 0) {
        if ((rc = cx->st.upload_rows(cx->xk.b.p, b->xk, n, max_nk * dim, b->xk_stride_case, b->xk_stride_k, dim, 8, s, K * dim, pm))) return rc;
        if ((rc = cx->st.upload_rows(cx->fk.b.p, b->fk, n, max_nk, b->fk_stride_case, b->fk_stride_k, 1, 8, s, K, pm))) return rc;
    }
    if ((rc = cx->st.upload_rows(cx->xi.b.p, b->xi, n, dim, b->xi_stride_case, dim, dim, 8, s, 0, pm))) return rc;
    if ((rc = cx->st.upload_rows(cx->fi.b.p, b->fi, n, max_no, b->fi_stride_case, max_no, max_no, 8, s, 0, pm))) return rc;
    WLSQM_HIP_CHECK(hipMemcpyAsync(cx->nk.b.p, h_nk.data(), n * 4, hipMemcpyHostToDevice, s));
    WLSQM_HIP_CHECK(hipMemcpyAsync(cx->wm.b.p, h_wm.data(), n * 4, hipMemcpyHostToDevice, s));
    WLSQM_HIP_CHECK(hipMemcpyAsync(cx->kn.b.p, h_kn.data(), n * 8, hipMemcpyHostToDevice, s));
    WLSQM_HIP_CHECK(hipMemsetAsync(cx->it.b.p, 0, 4, s));
    if (want_sens) WLSQM_HIP_CHECK(hipMemsetAsync(cx->sens.b.p, 0, (size_t)n * K * max_no * 8, s));

    mark("upload");
    KParams p{};
    p.xk = cx->xk.as<double>(); p.sxk_j = K * dim; p.sxk_k = dim;
    p.fk = cx->fk.as<double>(); p.sfk_j = K; p.sfk_k = 1;
    p.nk = cx->nk.as<int>(); p.snk = 1;
    p.xi = cx->xi.as<double>(); p.sxi_j = dim;
    p.fi = cx->fi.as<double>(); p.sfi_j = max_no;
    p.sens = want_sens ? cx->sens.as<double>() : nullptr; p.ss_j = K * max_no; p.ss_k = max_no;
    p.knowns = cx->kn.as<long long>(); p.sknowns = 1;
    p.wm = cx->wm.as<int>(); p.swm = 1;
    p.do_sens = want_sens ? 1 : 0; p.iterative = b->iterative ? 1 : 0; p.max_iter = b->max_iter;
    p.iters_out = cx->it.as<int>();
    {
        int32_t lo = n > 0 ? h_nk[0] : 0;
        for (int64_t j = 0; j < n; ++j) lo = h_nk[j] < lo ? h_nk[j] : lo;
        p.ragged = (max_nk - lo >= 8) ? 2 : 1;                         // (the staged kernels: their RAGGED copy, or the plain one without the marking)
        // ... and at the rows themselves: are the neighbours sorted by distance?  64 cases spread over the batch (in the caller's memory: the
        // arrays are right here); more than half of them out of order: the form of the staged kernels for unsorted rows (fit_stage.hip)
        p.rows_sorted = sampled_rows_sorted(n, dim, b->xk, b->xk_stride_case, b->xk_stride_k, b->xi, b->xi_stride_case,
                                            [&](int64_t r) { return h_nk[r]; }, user_row);
    }

    // bucket by order (the kernels are specialised per (dimension, order))
    if (uniform_order) {
        p.case_index = nullptr; p.ncases = n;
        rc = launch_fit(dim, h_order[0], p, K, s);
        if (rc != WLSQM_OK) return rc;
    } else {
        std::vector<long long> idx; idx.reserve(n);
        int64_t off[6] = {0};
        for (int o = 0; o <= 4; ++o) {
            off[o] = (int64_t)idx.size();
            for (int64_t j = 0; j < n; ++j) if (h_order[j] == o) idx.push_back(j);
        }
        off[5] = (int64_t)idx.size();
        if ((rc = cx->idx.need(idx.size() * 8))) return rc;
        WLSQM_HIP_CHECK(hipMemcpyAsync(cx->idx.b.p, idx.data(), idx.size() * 8, hipMemcpyHostToDevice, s));
        WLSQM_HIP_CHECK(hipStreamSynchronize(s));      // idx is a local vector
        for (int o = 0; o <= 4; ++o) {
            if (off[o + 1] == off[o]) continue;
            p.case_index = cx->idx.as<long long>() + off[o]; p.ncases = off[o + 1] - off[o];
            rc = launch_fit(dim, o, p, K, s);
            if (rc != WLSQM_OK) return rc;
        }
    }
    mark("kernels");
    // commit: everything was read before anything is written back (simple.pyx:1010-1019)
    rc = cx->st.download_rows(cx->fi.b.p, n, max_no, 8, s, [&](int64_t j, const char* row) {
        if (wlsqm_hip_number_of_reduced_dofs(h_no[j], h_kn[j]) < 1) return;     // nr < 1: the reference leaves the case untouched
        std::memcpy(b->fi + user_row(j) * b->fi_stride_case, row, (size_t)h_no[j] * 8);
    });
    if (rc != WLSQM_OK) return rc;
    if (want_sens) {
        rc = cx->st.download_rows(cx->sens.b.p, n, K * max_no, 8, s, [&](int64_t j, const char* row) {
            if (wlsqm_hip_number_of_reduced_dofs(h_no[j], h_kn[j]) < 1) return;
            unsigned long long known, dropped;
            effective_mask_host(h_no[j], h_kn[j], known, dropped);
            const double* r = reinterpret_cast<const double*>(row);
            double* sr = b->sens + user_row(j) * b->sens_stride_case;
            for (int64_t k = 0; k < h_nk[j]; ++k)
                for (int a = 0; a < h_no[j]; ++a) {
                    if ((dropped >> a) & 1ull) continue;       // never written by the reference
                    sr[k * b->sens_stride_k + a] = r[k * max_no + a];
                }
        });
        if (rc != WLSQM_OK) return rc;
    }
    int h_it = 0;
    WLSQM_HIP_CHECK(hipMemcpyAsync(&h_it, cx->it.b.p, 4, hipMemcpyDeviceToHost, s));
    WLSQM_HIP_CHECK(hipStreamSynchronize(s));
    if (iterations_out) *iterations_out = b->iterative ? (h_it < 1 ? 1 : h_it) : 0;     // (never below 1: see wlsqm_hip_fit_many_device)
    mark("commit");
    return WLSQM_OK;
}

static int cloud_params(KParams& p, int dimension, int order, int64_t ncases, int64_t max_nk, const double* S, const double* F,
                        const int32_t* hoods, int64_t hoods_stride_case, const int32_t* point_index, const int32_t* nk,
                        const int64_t* knowns, const int32_t* wm, double* fi, int64_t fi_stride_case) {
    if (dimension < 1 || dimension > 3) { set_error("dimension must be 1, 2 or 3"); return WLSQM_EVALUE; }
    if (wlsqm_hip_number_of_dofs(dimension, order) < 0) { set_error("order must be 0..4"); return WLSQM_EVALUE; }
    if (ncases < 1) { set_error("max_cases must be >= 1"); return WLSQM_EVALUE; }
    if (!S || !F || !hoods || !nk || !knowns || !wm || !fi) { set_error("null array"); return WLSQM_EVALUE; }
    p = KParams{};
    p.hoods = hoods; p.shoods_j = hoods_stride_case; p.S = S; p.F = F; p.pidx = point_index;
    p.nk = nk; p.snk = 1; p.knowns = (const long long*)knowns; p.sknowns = 1; p.wm = wm; p.swm = 1;
    p.fi = fi; p.sfi_j = fi_stride_case; p.ncases = ncases;
    (void)max_nk;
    return WLSQM_OK;
}

int wlsqm_hip_fit_cloud_device(int dimension, int order, int64_t ncases, int64_t max_nk,
                               const double* S, const double* F, const int32_t* hoods, int64_t hoods_stride_case,
                               const int32_t* point_index, const int32_t* nk, const int64_t* knowns,
                               const int32_t* weighting_method, double* fi, int64_t fi_stride_case,
                               double* sens, int64_t sens_stride_case, int64_t sens_stride_k, int do_sens,
                               int iterative, int max_iter, int device, void* stream, int32_t* iterations_out) {
    KParams p;
    int rc = cloud_params(p, dimension, order, ncases, max_nk, S, F, hoods, hoods_stride_case, point_index, nk, knowns,
                          weighting_method, fi, fi_stride_case);
    if (rc != WLSQM_OK) return rc;
    DeviceScope scope;
    rc = scope.enter(device);
    if (rc != WLSQM_OK) return rc;
    hipStream_t s = (hipStream_t)stream;
    p.sens = (do_sens ? sens : nullptr); p.ss_j = sens_stride_case; p.ss_k = sens_stride_k;
    p.do_sens = (do_sens && sens) ? 1 : 0; p.iterative = iterative ? 1 : 0; p.max_iter = max_iter;
    int* d_it = nullptr;               // as in wlsqm_hip_fit_many_device: only when the caller wants the count
    if (iterative && iterations_out) {
        rc = scratch_alloc_async(reinterpret_cast<void**>(&d_it), sizeof(int), s); if (rc != WLSQM_OK) return rc;
        WLSQM_HIP_CHECK(hipMemsetAsync(d_it, 0, sizeof(int), s));
        p.iters_out = d_it;
    }
    rc = launch_fit(dimension, order, p, max_nk, s);
    if (rc != WLSQM_OK) { (void)scratch_free_async(d_it, s); return rc; }
    if (iterations_out) *iterations_out = 0;
    if (d_it) {
        WLSQM_HIP_CHECK(hipMemcpyAsync(iterations_out, d_it, sizeof(int), hipMemcpyDeviceToHost, s));
        WLSQM_HIP_CHECK(hipStreamSynchronize(s));
        if (*iterations_out < 1) *iterations_out = 1;
        rc = scratch_free_async(d_it, s);
    }
    return rc;
}

static int time_launches(int dimension, int order, const KParams& p, long long max_nk, hipStream_t s, int reps, float* ms_out) {
    hipEvent_t e0, e1;
    WLSQM_HIP_CHECK(hipEventCreate(&e0));
    WLSQM_HIP_CHECK(hipEventCreate(&e1));
    int rc = launch_fit(dimension, order, p, max_nk, s);   // warm-up / code object load
    if (rc == WLSQM_OK) {
        (void)hipEventRecord(e0, s);
        for (int r = 0; r < reps && rc == WLSQM_OK; ++r) rc = launch_fit(dimension, order, p, max_nk, s);
        (void)hipEventRecord(e1, s);
        hipError_t e = hipEventSynchronize(e1);
        if (e != hipSuccess) rc = hip_fail(e, "hipEventSynchronize");
        else { float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1); *ms_out = ms / reps; }
    }
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    return rc;
}

int wlsqm_hip_time_fit_cloud_device(int dimension, int order, int64_t ncases, int64_t max_nk,
                                    const double* S, const double* F, const int32_t* hoods, int64_t hoods_stride_case,
                                    const int32_t* point_index, const int32_t* nk, const int64_t* knowns,
                                    const int32_t* weighting_method, double* fi, int64_t fi_stride_case,
                                    int device, void* stream, int reps, float* ms_out) {
    KParams p;
    int rc = cloud_params(p, dimension, order, ncases, max_nk, S, F, hoods, hoods_stride_case, point_index, nk, knowns,
                          weighting_method, fi, fi_stride_case);
    if (rc != WLSQM_OK) return rc;
    DeviceScope scope;
    rc = scope.enter(device);
    if (rc != WLSQM_OK) return rc;
    if (reps < 1 || !ms_out) { set_error("reps must be >= 1"); return WLSQM_EVALUE; }
    return time_launches(dimension, order, p, max_nk, (hipStream_t)stream, reps, ms_out);
}

int wlsqm_hip_time_fit_device(const wlsqm_batch* b, int device, void* stream, int order_uniform, int reps, float* ms_out) {
    int rc = validate_batch(b);
    if (rc != WLSQM_OK) return rc;
    DeviceScope scope;
    rc = scope.enter(device);
    if (rc != WLSQM_OK) return rc;
    if (reps < 1 || !ms_out) { set_error("reps must be >= 1"); return WLSQM_EVALUE; }
    return time_launches(b->dimension, order_uniform, params_from(b), b->max_nk, (hipStream_t)stream, reps, ms_out);
}

}  // extern "C"
