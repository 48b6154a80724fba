// wlsqm_internal.hpp — host-side declarations shared by the translation units of libwlsqm_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <mutex>
#include <stdint.h>
#include <cstdlib>
#include <string>

#include "wlsqm_hip.h"

#define WLSQM_HIP_CHECK(expr)                                                   \
    do {                                                                        \
        hipError_t _e = (expr);                                                 \
        if (_e != hipSuccess) return ::wlsqm::hip_fail(_e, #expr);              \
    } while (0)

namespace wlsqm {

// Kernel parameter block (by value).  Strides in elements.
struct KParams {
    const double* xk;  long long sxk_j, sxk_k;
    const double* fk;  long long sfk_j, sfk_k;
    const int* nk;     long long snk;
    long long max_nk;              // extent of the neighbour axis: every kernel clamps nk[j] to it (set by launch_fit)
    const double* xi;  long long sxi_j;
    double* fi;        long long sfi_j;
    double* sens;      long long ss_j, ss_k;
    const long long* knowns; long long sknowns;
    const int* wm;     long long swm;
    // Index-based ("cloud") mode, hoods != nullptr: the neighbours of case j are rows hoods[j, k] of the point
    // table S[npoints, dim] with values F[npoints]; xi = S[pidx ? pidx[j] : j]; xk / fk / xi above are unused.
    const int* hoods;  long long shoods_j;
    const double* S;   const double* F;
    const int* pidx;
    // Without pidx, case j of the launch sits at point pbase + j: a launch over the sub-batch [j0, j0 + n) of an index-based call
    // (slice_cases) keeps the cases' own points (the sliced 3D order-4 path fitted slices after the first around S[j - j0]).
    long long pbase = 0;
    const long long* case_index;   // nullable
    long long ncases;              // cases this launch processes
    // Nullable: the number of entries of case_index that are real, in DEVICE memory (the order buckets built on the device by
    // wlsqm_hip_fit_many_device_orders: the host never learns the bucket sizes).  The launch is sized for `ncases`; the kernels
    // that take case_index (lane, tile1, rows, wave, strict) stop at min(ncases, *ncases_dev).
    const long long* ncases_dev;
    int do_sens, iterative, max_iter;
    int* iters_out;                // device int (atomicMax), nullable
    // Two-kernel moment path (fit_moment.hip): the tile kernel parks the reduced moments of case j at
    // ws[e * ws_stride + j] (structure of arrays) and the solve kernel picks them up.
    double* ws;        long long ws_stride;
    // Refinement in ROUNDS (fit_tilek.hip, the one-wave tile kernel with extras; round 4).  The reference's stop test (impl.pyx:1057)
    // fires after 3.1 sweeps on average on BASELINE configs[1] but a 16-case tile holds a case that runs 8 of them: a round does the
    // sweeps [it_first, it_stop) for its cases, writes the cases that are still running to `cont_list` (their number to
    // `cont_count`, their last residual norm to it_state[case]) and the next round — a launch over that list — carries on from the
    // iterate in fi.  it_stop == 0: no rounds (the whole loop in one launch, as before).
    int it_first = 0, it_stop = 0;
    long long* cont_list = nullptr; long long* cont_count = nullptr;
    double* it_state = nullptr;
    // the neighbours of every row come sorted by distance (1: a k-nearest-neighbour search's rows — the default) or in no order (0: a ball
    // query): the caller's word (wlsqm_hip_set_order_hint).  Picks the FORM of the staged kernel of the small dense systems (fit_stage.hip:
    // two waves per SIMD, or one that owns its SIMD and finds the second pass's rows in L2) — the same bits either way.
    int rows_sorted = 1;
    // what the caller knows about the neighbour counts of a dense batch: 0 nothing (device-resident counts), 1 every case fills its row,
    // 2 ragged (the host entry points look: fit_stage.hip runs its RAGGED copy)
    int ragged = 1;
};

// Cases a launch really has (see KParams::ncases_dev).
__device__ __forceinline__ long long live_cases(const KParams& p) {
    if (!p.ncases_dev) return p.ncases;
    const long long d = *p.ncases_dev;
    return d < p.ncases ? d : p.ncases;
}

// The case's own point of an index-based launch (xi = S[own_point]).
__device__ __forceinline__ long long own_point(const KParams& p, long long j) { return p.pidx ? (long long)p.pidx[j] : p.pbase + j; }

// The sub-batch [j0, j0 + n) of a launch (dense or index-based input, no case_index).
inline KParams slice_cases(const KParams& p, long long j0, long long n) {
    KParams q = p;
    if (p.xk) q.xk = p.xk + j0 * p.sxk_j;
    if (p.fk) q.fk = p.fk + j0 * p.sfk_j;
    q.nk = p.nk + j0 * p.snk;
    if (p.xi) q.xi = p.xi + j0 * p.sxi_j;
    q.fi = p.fi + j0 * p.sfi_j;
    if (p.sens) q.sens = p.sens + j0 * p.ss_j;
    q.knowns = p.knowns + j0 * p.sknowns;
    q.wm = p.wm + j0 * p.swm;
    if (p.hoods) q.hoods = p.hoods + j0 * p.shoods_j;
    if (p.pidx) q.pidx = p.pidx + j0; else q.pbase = p.pbase + j0;
    q.ncases = n;
    return q;
}

void set_error(const std::string& msg);
int hip_fail(hipError_t e, const char* what);

// Launch the fit kernels for one (dimension, order) bucket.  Returns WLSQM_* code.
// max_nk: extent of the neighbour axis (upper bound of nk[j]).
int launch_fit(int dimension, int order, const KParams& p, long long max_nk, hipStream_t stream);

// Optional capture of the reference's intermediates by the strict kernel (tests: bit-for-bit against tests/golden/sweep_*.npz).
struct StrictDebug {
    double* w; long long w_stride;                  // [case, k]
    double* A; double* LU; long long mat_stride;    // [case, j + nr * m]: unscaled normal matrix / scaled LU factor
    double* row_scale; double* col_scale; long long vec_stride;
    int* ipiv;                                      // [case, j] at vec_stride, 1-based
};

// true when the calling thread asked for reference-order numerics (WLSQM_HIP_STRICT / wlsqm_hip_set_strict): launch_fit then
// dispatches every shape to fit_strict.hip
bool strict_mode();
// true for mode 2, "accurate" (fit_accurate.hip): strict_mode() is true as well — every call takes the strict dispatch, in which the
// basic fits of the 2D / 3D systems up to 10 unknowns without a known DOF run fit_accurate_kernel (reference-order arithmetic with
// the normal matrix assembled from its upper triangle)
bool accurate_mode();

// name of the kernel family the last launch_fit on this thread dispatched to ("lane", "tile", "wave")
const char* last_kernel_name();
void note_kernel(const char* name);

// Per-device launch facts of one persistent kernel (a process may drive several GPUs): CU count, the one-time opt-in to
// more than 64 KB of dynamic LDS, and (when the LDS size never changes) the workgroups that fit one CU.
struct KernelSetup { int cus[16] = {}; int per_cu[16] = {}; };

// Workgroups launched per resident workgroup slot.  A grid of exactly the resident workgroups leaves the tail of the launch
// unbalanced (1M C2 cases are 62 500 tiles over 3 072 waves: 20 or 21 tiles each, and the waves do not finish their tiles at
// the same pace); launching several workgroups per slot lets the dispatcher hand the leftovers to whichever slot frees up first
// (tools/tune.py g1 / g8 / g16 / g1000, interleaved: C2 0.1737 / 0.1665 / 0.1656 / 0.1655 ms, C5 0.3512 / 0.3426 / 0.3386 /
// 0.3409, C3 0.672 / 0.650 / 0.639).  WLSQM_HIP_GRID_MULT overrides it (A/B).
inline double grid_multiple() {
    const char* e = getenv("WLSQM_HIP_GRID_MULT");
    const double v = e ? atof(e) : 16.0;
    return v > 0.0 ? v : 16.0;
}

// Grid of a persistent launch: resident workgroups per CU x CUs of the current device x grid_multiple()
// (callers clamp it to the number of tiles).
inline int persistent_grid(const void* kern, int threads, size_t lds_bytes, size_t lds_optin, bool fixed_lds, KernelSetup& ks,
                           long long* grid) {
    int dev = 0;
    WLSQM_HIP_CHECK(hipGetDevice(&dev));
    if (dev < 0 || dev >= 16) { set_error("device ordinal out of range"); return WLSQM_EVALUE; }
    if (!ks.cus[dev]) {
        hipDeviceProp_t prop;
        WLSQM_HIP_CHECK(hipGetDeviceProperties(&prop, dev));
        if (lds_optin > 64 * 1024)
            WLSQM_HIP_CHECK(hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_optin));
        ks.cus[dev] = prop.multiProcessorCount;
    }
    int occ = fixed_lds ? ks.per_cu[dev] : 0;
    if (!occ) {
        WLSQM_HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kern, threads, lds_bytes));
        if (occ < 1) occ = 1;
        if (fixed_lds) ks.per_cu[dev] = occ;
    }
    *grid = (long long)((double)occ * ks.cus[dev] * grid_multiple());
    if (*grid < 1) *grid = 1;
    return WLSQM_OK;
}

// RAII device buffer
struct DevBuf {
    void* p = nullptr; size_t n = 0;
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    ~DevBuf() { if (p) (void)hipFree(p); }
    int alloc(size_t bytes) {
        if (p) { (void)hipFree(p); p = nullptr; }
        n = bytes;
        if (bytes == 0) return WLSQM_OK;
        hipError_t e = hipMalloc(&p, bytes);
        if (e != hipSuccess) { p = nullptr; n = 0; return hip_fail(e, "hipMalloc"); }
        return WLSQM_OK;
    }
    template <class T> T* as() const { return static_cast<T*>(p); }
};

int check_device(int device);

// Makes `device` current for the duration of one API call and puts the caller's device back on the way out (the library
// must not leave a different current device behind in a C caller's thread).
struct DeviceScope {
    int prev = -1;
    DeviceScope() = default;
    DeviceScope(const DeviceScope&) = delete;
    DeviceScope& operator=(const DeviceScope&) = delete;
    ~DeviceScope() { if (prev >= 0) (void)hipSetDevice(prev); }
    int enter(int device) {
        int cur = -1;
        if (hipGetDevice(&cur) != hipSuccess) { (void)hipGetLastError(); cur = -1; }
        const int rc = check_device(device);
        if (rc == WLSQM_OK && cur >= 0 && cur != device) prev = cur;
        return rc;
    }
};

// Stream-ordered scratch memory from a PRIVATE pool per device (the process-wide default pool and its release threshold are
// left alone): freed blocks stay in the pool, so the next call on the stream gets the same memory back without a device
// synchronisation.  Legal during stream capture (becomes an allocation node of the graph).
int scratch_alloc_async(void** out, size_t bytes, hipStream_t stream);
int scratch_free_async(void* p, hipStream_t stream);

// Hand-over space between the kernels of ONE call (status bytes, small lists): a device buffer that PERSISTS per (device, stream) — a
// stream-ordered allocation per call costs the stream a marker on either side (~10 us of a 0.27 ms accurate-mode call, measured in rounds
// 5 and 6).  The contents are UNDEFINED when it is handed out (the call's first kernel writes what its later kernels read: no clearing,
// no state between calls).  The caller keeps the CallScratch — it holds a lock — until it has enqueued every kernel that touches the
// buffer: two host threads that share a stream cannot interleave their launches around it.  Inside a graph capture (and when the
// table is full of other streams' buffers) the space is stream-ordered scratch instead: call_scratch_release frees it behind the kernels.
struct CallScratch { void* p = nullptr; bool pooled = false; std::unique_lock<std::mutex> lock; };
int call_scratch_acquire(CallScratch* cs, size_t bytes, hipStream_t stream);
int call_scratch_release(CallScratch* cs, hipStream_t stream);

// the calling thread's word about the rows of the device-resident batches it hands over (wlsqm_hip_set_row_hint / _order_hint, api.hip)
int row_hint_value();
int order_hint_value();

// Host mirror of effective_mask() in wlsqm_kernels.hpp (infra.pyx:119-121 quirk): returns the
// mask of DOFs the reference never writes (true knowns | dropped) and the dropped subset.
inline void effective_mask_host(int no, long long raw, unsigned long long& known, unsigned long long& dropped) {
    const unsigned long long full = (no >= 64) ? ~0ull : ((1ull << no) - 1ull);
    known = (unsigned long long)raw & full; dropped = 0;
    int extra = __builtin_popcountll((unsigned long long)raw & ~full);
    for (int t = no - 1; t >= 0 && extra > 0; --t)
        if (!((known >> t) & 1ull)) { known |= 1ull << t; dropped |= 1ull << t; --extra; }
}

}  // namespace wlsqm

