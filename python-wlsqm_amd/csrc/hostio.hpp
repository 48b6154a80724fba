// hostio.hpp — host <-> device staging for the host-array entry points (the reference's own API hands over
// numpy arrays).  A per-thread context owns two pinned staging buffers and grow-only device buffers, so a call
// pays neither hipMalloc/hipFree nor pageable-memory DMA: rows are packed into pinned memory by all host
// threads (OpenMP memcpy) while the previous chunk is on the PCIe link.
#pragma once
#include <cstdlib>
#include <cstring>
#include <vector>

#include <omp.h>

#include "wlsqm_internal.hpp"

namespace wlsqm {

// Threads used for packing/committing rows: a small team wakes up far faster than one thread per hardware
// thread of a 256-way host, and 8-16 memcpy streams already exceed the PCIe rate.
inline int copy_threads() {
    static int n = 0;
    if (!n) {
        const char* e = getenv("WLSQM_HIP_COPY_THREADS");
        int want = e ? atoi(e) : 16;
        int have = omp_get_max_threads();
        n = want < 1 ? 1 : (want > have ? have : want);
    }
    return n;
}

struct Stager {
    static constexpr size_t CHUNK = size_t(32) << 20;          // bytes per pinned buffer
    void* pin[2] = {nullptr, nullptr};
    hipEvent_t ev[2] = {nullptr, nullptr};
    bool busy[2] = {false, false};
    int device = -1;
    int turn = 0;

    ~Stager() { release(); }
    void release() {
        for (int i = 0; i < 2; ++i) {
            if (pin[i]) (void)hipHostFree(pin[i]);
            if (ev[i]) (void)hipEventDestroy(ev[i]);
            pin[i] = nullptr; ev[i] = nullptr; busy[i] = false;
        }
        device = -1;
    }
    int ensure(int dev) {
        if (device == dev && pin[0]) return WLSQM_OK;
        release();
        for (int i = 0; i < 2; ++i) {
            hipError_t e = hipHostMalloc(&pin[i], CHUNK, hipHostMallocDefault);
            if (e != hipSuccess) { pin[i] = nullptr; return hip_fail(e, "hipHostMalloc"); }
            e = hipEventCreateWithFlags(&ev[i], hipEventDisableTiming);
            if (e != hipSuccess) return hip_fail(e, "hipEventCreate");
        }
        device = dev;
        return WLSQM_OK;
    }
    // next free pinned buffer (waits for its previous transfer)
    int acquire(int& slot) {
        slot = turn; turn ^= 1;
        if (busy[slot]) { WLSQM_HIP_CHECK(hipEventSynchronize(ev[slot])); busy[slot] = false; }
        return WLSQM_OK;
    }
    int drain() {
        for (int i = 0; i < 2; ++i)
            if (busy[i]) { WLSQM_HIP_CHECK(hipEventSynchronize(ev[i])); busy[i] = false; }
        return WLSQM_OK;
    }

    // Upload `nrows` rows of `row_elems` elements of size `esz` to a dense device array.  Source row r starts at
    // src + r*src_row_stride (elements); inside a row the elements have stride `inner_stride` groups of `group`
    // contiguous elements (group = dim for xk rows, 1 for fk).  Unit inner stride -> memcpy per row (or per chunk).
    // `dst_row_elems` (>= row_elems, default = row_elems) is the row length of the device array: the tail of each
    // device row is padding that is transferred but never read.
    // `perm` (optional): device row r is source row perm[r] (the host path packs ragged batches in neighbour-count order).
    int upload_rows(void* dst_dev, const void* src, int64_t nrows, int64_t row_elems, int64_t src_row_stride,
                    int64_t inner_stride, int64_t group, size_t esz, hipStream_t s, int64_t dst_row_elems = 0, const int64_t* perm = nullptr) {
        if (nrows <= 0 || row_elems <= 0) return WLSQM_OK;
        if (dst_row_elems < row_elems) dst_row_elems = row_elems;
        const size_t copy_bytes = (size_t)row_elems * esz;
        const size_t row_bytes = (size_t)dst_row_elems * esz;
        const int64_t rows_per_chunk = std::max<int64_t>(1, (int64_t)(CHUNK / row_bytes));
        const bool inner_contig = (inner_stride == group) || (row_elems == group);
        const bool fully_contig = inner_contig && src_row_stride == row_elems && dst_row_elems == row_elems && !perm;
        for (int64_t r0 = 0; r0 < nrows; r0 += rows_per_chunk) {
            const int64_t nr = std::min<int64_t>(rows_per_chunk, nrows - r0);
            int slot; int rc = acquire(slot); if (rc != WLSQM_OK) return rc;
            char* stage = static_cast<char*>(pin[slot]);
            const char* base = static_cast<const char*>(src);
            if (fully_contig) {
                const size_t total = (size_t)nr * row_bytes, piece = size_t(1) << 20;
                const int64_t npieces = (int64_t)((total + piece - 1) / piece);
#pragma omp parallel for schedule(static) num_threads(copy_threads())
                for (int64_t i = 0; i < npieces; ++i) {
                    const size_t o = (size_t)i * piece;
                    std::memcpy(stage + o, base + (size_t)r0 * row_bytes + o, std::min(piece, total - o));
                }
            } else if (inner_contig) {
#pragma omp parallel for schedule(static) num_threads(copy_threads())
                for (int64_t r = 0; r < nr; ++r)
                    std::memcpy(stage + (size_t)r * row_bytes, base + (size_t)(perm ? perm[r0 + r] : r0 + r) * src_row_stride * esz, copy_bytes);
            } else {
                const int64_t ngroups = row_elems / group;
#pragma omp parallel for schedule(static) num_threads(copy_threads())
                for (int64_t r = 0; r < nr; ++r) {
                    const char* sr = base + (size_t)(perm ? perm[r0 + r] : r0 + r) * src_row_stride * esz;
                    char* dr = stage + (size_t)r * row_bytes;
                    for (int64_t g = 0; g < ngroups; ++g)
                        std::memcpy(dr + (size_t)g * group * esz, sr + (size_t)g * inner_stride * esz, (size_t)group * esz);
                }
            }
            WLSQM_HIP_CHECK(hipMemcpyAsync(static_cast<char*>(dst_dev) + (size_t)r0 * row_bytes, stage, (size_t)nr * row_bytes,
                                           hipMemcpyHostToDevice, s));
            WLSQM_HIP_CHECK(hipEventRecord(ev[slot], s));
            busy[slot] = true;
        }
        return WLSQM_OK;
    }

    // Download a dense device array of `nrows` rows of `row_elems` doubles; `commit(r, rowptr)` is called for every
    // row (in parallel) to write it where it belongs in the user's array.
    template <class Commit>
    int download_rows(const void* src_dev, int64_t nrows, int64_t row_elems, size_t esz, hipStream_t s, Commit commit) {
        if (nrows <= 0 || row_elems <= 0) return WLSQM_OK;
        const size_t row_bytes = (size_t)row_elems * esz;
        const int64_t rows_per_chunk = std::max<int64_t>(1, (int64_t)(CHUNK / row_bytes));
        int rc = drain(); if (rc != WLSQM_OK) return rc;
        // two-deep pipeline: chunk i+1 is on the link while chunk i is committed
        int64_t r0 = 0; int slot_prev = -1; int64_t r0_prev = 0, nr_prev = 0;
        while (r0 < nrows || slot_prev >= 0) {
            int slot = -1; int64_t nr = 0;
            if (r0 < nrows) {
                nr = std::min<int64_t>(rows_per_chunk, nrows - r0);
                slot = turn; turn ^= 1;
                WLSQM_HIP_CHECK(hipMemcpyAsync(pin[slot], static_cast<const char*>(src_dev) + (size_t)r0 * row_bytes,
                                               (size_t)nr * row_bytes, hipMemcpyDeviceToHost, s));
                WLSQM_HIP_CHECK(hipEventRecord(ev[slot], s));
            }
            if (slot_prev >= 0) {
                WLSQM_HIP_CHECK(hipEventSynchronize(ev[slot_prev]));
                const char* stage = static_cast<const char*>(pin[slot_prev]);
#pragma omp parallel for schedule(static) num_threads(copy_threads())
                for (int64_t r = 0; r < nr_prev; ++r) commit(r0_prev + r, stage + (size_t)r * row_bytes);
            }
            slot_prev = slot; r0_prev = r0; nr_prev = nr;
            r0 += nr;
        }
        return WLSQM_OK;
    }
};

// grow-only device buffer
struct GrowBuf {
    DevBuf b; size_t cap = 0;
    int need(size_t bytes) {
        if (bytes <= cap && b.p) return WLSQM_OK;
        int rc = b.alloc(bytes + bytes / 8);
        cap = (rc == WLSQM_OK) ? bytes + bytes / 8 : 0;
        return rc;
    }
    template <class T> T* as() const { return b.as<T>(); }
};

// Are the neighbours of these rows sorted by distance from the case's point?  64 cases spread over the batch, in the CALLER's memory (host
// arrays); more than half of them out of order says "unsorted": the form of the staged kernels for such rows (KParams::rows_sorted,
// fit_stage.hip).  nk_of(r) / row_of(r): the neighbour count and the caller's row of the r-th case.
template <class NkOf, class RowOf>
inline int sampled_rows_sorted(int64_t n, int dim, const double* xk, int64_t sxk_j, int64_t sxk_k, const double* xi, int64_t sxi_j,
                               NkOf nk_of, RowOf row_of) {
    int looked = 0, unsorted = 0;
    const int64_t step = n / 64 > 0 ? n / 64 : 1;
    for (int64_t r = 0; r < n && looked < 64; r += step) {
        const int64_t j = row_of(r);
        const int64_t nkj = nk_of(r);
        if (nkj < 3) continue;
        ++looked;
        double prev = -1.0; bool mono = true;
        for (int64_t k = 0; k < nkj && mono; ++k) {
            double d2 = 0.0;
            for (int m = 0; m < dim; ++m) { const double d = xk[j * sxk_j + k * sxk_k + m] - xi[j * sxi_j + m]; d2 += d * d; }
            mono = d2 >= prev; prev = d2;
        }
        if (!mono) ++unsorted;
    }
    return (looked > 0 && 2 * unsorted > looked) ? 0 : 1;
}

}  // namespace wlsqm
