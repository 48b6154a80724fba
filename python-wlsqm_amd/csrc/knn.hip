// knn.hip — k nearest neighbours of every point of a cloud, on the GPU (the step BEFORE the fit).
//
// The reference's examples build the neighbourhoods on the host with scipy.spatial.cKDTree
// (examples/expertsolver_example.py:48-66: tree.query(x, 1 + nk), self dropped; examples/wlsqm_example.py:103-133) and
// hand `hoods` / `x[hoods]` to the fitter.  For a device-resident cloud that search is the last host step; this file
// replaces it with an exact uniform-grid search:
//   1. bounding box (atomic min/max on order-preserving integer keys), grid of ~4 points per cell;
//   2. points sorted by cell (hipcub radix sort), cell start offsets by binary search, coordinates gathered in cell order;
//   3. one lane per query, in CELL order (the 64 lanes of a wave are spatial neighbours: same cells, coherent loops and
//      cache lines): Chebyshev rings of cells around the query's cell, best-k candidates in LDS ([slot][lane], conflict
//      free), until the k-th best distance is no larger than the distance to the unvisited region;
//   4. the k results sorted by (distance, index) — cKDTree's order — and written to hoods[point, :].
// Exact (no approximation): the stop test is a proof that no unvisited point can be closer.
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstring>

#include <hipcub/hipcub.hpp>

#include "wlsqm_internal.hpp"
#include "wlsqm_interp.hpp"

namespace wlsqm {

struct KnnGrid {
    double lo[3], inv_cell[3], cell[3];
    int g[3];
    int dim;
};

__device__ __forceinline__ unsigned long long key_of(double v) {      // order-preserving map double -> uint64
    const unsigned long long u = (unsigned long long)__double_as_longlong(v);
    return (u >> 63) ? ~u : (u | 0x8000000000000000ull);
}
static inline double value_of(unsigned long long k) {
    const unsigned long long u = (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k;
    double v; memcpy(&v, &u, 8); return v;
}

// Grid-stride partial extrema per lane, met per wave by shuffles and per block through LDS: one pair of atomics per block
// and axis (one per WAVE and axis serialises ~60k same-address atomics in L2 for 1M points: 0.71 ms instead of ~0.03).
__global__ __launch_bounds__(256) void knn_bbox_kernel(const double* __restrict__ S, long long n, int dim,
                                                       unsigned long long* mm /*[2][3]*/) {
    __shared__ double s_lo[4][3], s_hi[4][3];
    const long long stride = (long long)gridDim.x * blockDim.x;
    const long long i0 = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int wave = threadIdx.x >> 6;
    for (int m = 0; m < dim; ++m) {
        double lo = S[(i0 < n ? i0 : n - 1) * dim + m], hi = lo;   // tail lanes replay the last point
        for (long long i = i0 + stride; i < n; i += stride) {
            const double v = S[i * dim + m];
            lo = (v < lo || v != v) ? v : lo;                       // NaN wins, so that it is reported
            hi = (v > hi || v != v) ? v : hi;
        }
        for (int off = 32; off > 0; off >>= 1) {
            const double a = __shfl_xor(lo, off, 64), b = __shfl_xor(hi, off, 64);
            lo = (a < lo || a != a) ? a : lo;
            hi = (b > hi || b != b) ? b : hi;
        }
        if ((threadIdx.x & 63) == 0) { s_lo[wave][m] = lo; s_hi[wave][m] = hi; }
    }
    __syncthreads();
    if (threadIdx.x < dim) {
        const int m = threadIdx.x;
        double lo = s_lo[0][m], hi = s_hi[0][m];
        for (int w = 1; w < (int)(blockDim.x >> 6); ++w) {
            const double a = s_lo[w][m], b = s_hi[w][m];
            lo = (a < lo || a != a) ? a : lo;
            hi = (b > hi || b != b) ? b : hi;
        }
        atomicMin(&mm[m], key_of(lo));
        atomicMax(&mm[3 + m], key_of(hi));
    }
}

__device__ __forceinline__ int cell_coord(double x, const KnnGrid& G, int m) {
    int c = (int)((x - G.lo[m]) * G.inv_cell[m]);
    return c < 0 ? 0 : (c >= G.g[m] ? G.g[m] - 1 : c);
}

__global__ void knn_cell_kernel(const double* __restrict__ S, long long n, KnnGrid G, unsigned* __restrict__ cell,
                                int* __restrict__ idx) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    unsigned c = 0;
    for (int m = G.dim - 1; m >= 0; --m) c = c * (unsigned)G.g[m] + (unsigned)cell_coord(S[i * G.dim + m], G, m);
    cell[i] = c;
    idx[i] = (int)i;
}

// start[c] = first sorted position whose cell id is >= c  (c = 0..ncells)
__global__ void knn_start_kernel(const unsigned* __restrict__ sorted_cell, long long n, long long ncells, int* __restrict__ start) {
    const long long c = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (c > ncells) return;
    long long lo = 0, hi = n;
    while (lo < hi) { const long long mid = (lo + hi) >> 1; if ((long long)sorted_cell[mid] < c) lo = mid + 1; else hi = mid; }
    start[c] = (int)lo;
}

__global__ void knn_gather_kernel(const double* __restrict__ S, const int* __restrict__ perm, long long n, int dim,
                                  double* __restrict__ Ss) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const long long src = perm[i];
    for (int m = 0; m < dim; ++m) Ss[i * dim + m] = S[src * dim + m];
}

template <int DIM>
__global__ __launch_bounds__(64) void knn_query_kernel(const double* __restrict__ Ss, const int* __restrict__ perm,
                                                      const int* __restrict__ start, long long n, int k, KnnGrid G,
                                                      double r2max, int row_stride, int* __restrict__ hoods,
                                                      int* __restrict__ counts, long long nquery) {
    extern __shared__ __attribute__((aligned(8))) unsigned char smem[];
    double* bd = reinterpret_cast<double*>(smem) + threadIdx.x;                       // [k][64] best squared distances
    int* bi = reinterpret_cast<int*>(smem + (size_t)k * 64 * sizeof(double)) + threadIdx.x;   // [k][64] their sorted positions
    const long long q = (long long)blockIdx.x * 64 + threadIdx.x;
    if (q >= n) return;
    // subset search: only the first `nquery` points of the cloud ask (the others are candidates only — the halo band of a
    // partitioned cloud, wlsqm/sharded.py); their lanes retire here
    if (perm[q] >= nquery) return;
    double x[DIM]; int cq[DIM];
#pragma unroll
    for (int m = 0; m < DIM; ++m) { x[m] = Ss[q * DIM + m]; cq[m] = cell_coord(x[m], G, m); }

    int count = 0;
    double worst = -1.0;          // squared distance at the root of the full heap (k kept)
    int rmax = 0;
#pragma unroll
    for (int m = 0; m < DIM; ++m) rmax = max(rmax, max(cq[m], G.g[m] - 1 - cq[m]));

    for (int r = 0; r <= rmax; ++r) {
        // cells on the Chebyshev shell of radius r around the query's cell
        const int z0 = DIM > 2 ? -r : 0, z1 = DIM > 2 ? r : 0, y0 = DIM > 1 ? -r : 0, y1 = DIM > 1 ? r : 0;
        for (int dz = z0; dz <= z1; ++dz) {
            const int cz = DIM > 2 ? cq[2] + dz : 0;
            if (DIM > 2 && (cz < 0 || cz >= G.g[2])) continue;
            for (int dy = y0; dy <= y1; ++dy) {
                const int cy = DIM > 1 ? cq[1] + dy : 0;
                if (DIM > 1 && (cy < 0 || cy >= G.g[1])) continue;
                const bool face = (DIM > 2 && (dz == -r || dz == r)) || (DIM > 1 && (dy == -r || dy == r));
                // on a face of the shell: the whole x-run; otherwise only its two ends
                const int step = (face || r == 0) ? 1 : 2 * r;
                for (int dx = -r; dx <= r; dx += step) {
                    const int cx = cq[0] + dx;
                    if (cx < 0 || cx >= G.g[0]) continue;
                    long long cid = cx;
                    if (DIM > 1) cid += (long long)G.g[0] * cy;
                    if (DIM > 2) cid += (long long)G.g[0] * G.g[1] * cz;
                    const int p0 = start[cid], p1 = start[cid + 1];
                    for (int pos = p0; pos < p1; ++pos) {
                        if (pos == q) continue;                                        // the point itself
                        double d2 = 0.0;
#pragma unroll
                        for (int m = 0; m < DIM; ++m) { const double d = Ss[(long long)pos * DIM + m] - x[m]; d2 += d * d; }
                        if (d2 > r2max) continue;                                      // ball search: outside the radius
                        if (count < k) {
                            // append and sift up (max-heap on the squared distance: the root is the worst kept so far)
                            int i = count++;
                            while (i > 0) {
                                const int up = (i - 1) >> 1;
                                const double v = bd[up * 64];
                                if (v >= d2) break;
                                bd[i * 64] = v; bi[i * 64] = bi[up * 64];
                                i = up;
                            }
                            bd[i * 64] = d2; bi[i * 64] = pos;
                            if (count == k) worst = bd[0];
                        } else if (d2 < worst) {
                            // replace the root and sift down
                            int i = 0;
                            for (;;) {
                                int ch = 2 * i + 1;
                                if (ch >= k) break;
                                double cv = bd[ch * 64];
                                if (ch + 1 < k) { const double rv = bd[(ch + 1) * 64]; if (rv > cv) { cv = rv; ++ch; } }
                                if (cv <= d2) break;
                                bd[i * 64] = cv; bi[i * 64] = bi[ch * 64];
                                i = ch;
                            }
                            bd[i * 64] = d2; bi[i * 64] = pos;
                            worst = bd[0];
                        }
                    }
                }
            }
        }
        {
            // every unvisited point lies outside the block of cells [cq - r, cq + r]: at least `reach` away
            double reach = DBL_MAX;
#pragma unroll
            for (int m = 0; m < DIM; ++m) {
                // (minus a sliver: a point's cell number and this boundary are rounded independently)
                if (cq[m] - r > 0) reach = fmin(reach, x[m] - (G.lo[m] + (cq[m] - r) * G.cell[m]) - 1e-9 * G.cell[m]);
                if (cq[m] + r < G.g[m] - 1) reach = fmin(reach, (G.lo[m] + (cq[m] + r + 1) * G.cell[m]) - x[m] - 1e-9 * G.cell[m]);
            }
            if (reach == DBL_MAX) break;                                   // the whole grid has been visited
            if (reach > 0.0 && ((count == k && worst <= reach * reach) || reach * reach > r2max)) break;
        }
    }
    // map sorted positions to point indices
    const int self = perm[q];
    const long long row = (long long)self * row_stride;
    if (counts) counts[self] = count;
    for (int i = count; i < row_stride; ++i) hoods[row + i] = self;       // unused slots: a valid index, masked by the count
    for (int i = 0; i < count; ++i) bi[i * 64] = perm[bi[i * 64]];
    // ascending (distance, original index): heap sort in place on the full key (the search ordered by distance only)
    auto sift = [&](int i, int size, double v, int vi) {
        for (;;) {
            int ch = 2 * i + 1;
            if (ch >= size) break;
            double cv = bd[ch * 64]; int ci = bi[ch * 64];
            if (ch + 1 < size) {
                const double rv = bd[(ch + 1) * 64]; const int ri = bi[(ch + 1) * 64];
                if (rv > cv || (rv == cv && ri > ci)) { cv = rv; ci = ri; ++ch; }
            }
            if (!(cv > v || (cv == v && ci > vi))) break;
            bd[i * 64] = cv; bi[i * 64] = ci;
            i = ch;
        }
        bd[i * 64] = v; bi[i * 64] = vi;
    };
    for (int i = count / 2 - 1; i >= 0; --i) sift(i, count, bd[i * 64], bi[i * 64]);
    for (int size = count - 1; size > 0; --size) {
        const double v = bd[size * 64]; const int vi = bi[size * 64];
        bd[size * 64] = bd[0]; bi[size * 64] = bi[0];
        sift(0, size, v, vi);
    }
    for (int i = 0; i < count; ++i) hoods[row + i] = bi[i * 64];
}

}  // namespace wlsqm

using namespace wlsqm;

// Uniform grid over a device-resident cloud: bounding box, cells of ~4 points, points sorted by cell.
struct GridIndex {
    KnnGrid G{};
    long long ncells = 1;
    DevBuf d_perm, d_start, d_Ss;
    int build(int dimension, int64_t npoints, const double* S, hipStream_t s);
};

int GridIndex::build(int dimension, int64_t npoints, const double* S, hipStream_t s) {
    int rc;
    const long long n = npoints;
    const unsigned blocks = (unsigned)((n + 255) / 256);
    // 1. bounding box
    DevBuf d_mm;
    if ((rc = d_mm.alloc(6 * sizeof(unsigned long long)))) return rc;
    unsigned long long h_mm[6] = {~0ull, ~0ull, ~0ull, 0ull, 0ull, 0ull};
    WLSQM_HIP_CHECK(hipMemcpyAsync(d_mm.p, h_mm, sizeof(h_mm), hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(knn_bbox_kernel, dim3(blocks < 1024u ? blocks : 1024u), dim3(256), 0, s, S, n, dimension,
                       d_mm.as<unsigned long long>());
    WLSQM_HIP_CHECK(hipMemcpyAsync(h_mm, d_mm.p, sizeof(h_mm), hipMemcpyDeviceToHost, s));
    WLSQM_HIP_CHECK(hipStreamSynchronize(s));
    G = KnnGrid{};
    G.dim = dimension;
    double ext[3] = {0, 0, 0}, vol = 1.0;
    int live = 0;
    for (int m = 0; m < dimension; ++m) {
        const double lo = value_of(h_mm[m]), hi = value_of(h_mm[3 + m]);
        if (!(lo == lo) || !(hi == hi) || hi - lo > DBL_MAX) { set_error("non-finite coordinates"); return WLSQM_EVALUE; }
        G.lo[m] = lo; ext[m] = hi - lo;
        if (ext[m] > 0.0) { vol *= ext[m]; ++live; }
    }
    // ~4 points per cell over the non-degenerate axes, at most 2^27 cells
    const double target_cells = std::min((double)n / 4.0, 134217728.0);
    const double h = live ? std::pow(vol / std::max(target_cells, 1.0), 1.0 / live) : 1.0;
    ncells = 1;
    for (int m = 0; m < dimension; ++m) {
        int g = ext[m] > 0.0 ? (int)std::min(std::max(ext[m] / h, 1.0), 4096.0 * 4096.0) : 1;
        if (dimension == 3) g = std::min(g, 1024); else if (dimension == 2) g = std::min(g, 16384);
        G.g[m] = std::max(g, 1);
        G.cell[m] = ext[m] > 0.0 ? ext[m] / G.g[m] : 1.0;
        G.inv_cell[m] = 1.0 / G.cell[m];
        ncells *= G.g[m];
    }
    for (int m = dimension; m < 3; ++m) { G.g[m] = 1; G.cell[m] = 1.0; G.inv_cell[m] = 1.0; G.lo[m] = 0.0; }

    // 2. sort by cell
    DevBuf d_cell, d_cell2, d_idx, d_tmp;
    if ((rc = d_cell.alloc(n * 4)) || (rc = d_cell2.alloc(n * 4)) || (rc = d_idx.alloc(n * 4)) || (rc = d_perm.alloc(n * 4)) ||
        (rc = d_start.alloc((ncells + 1) * 4)) || (rc = d_Ss.alloc((size_t)n * dimension * 8))) return rc;
    hipLaunchKernelGGL(knn_cell_kernel, dim3(blocks), dim3(256), 0, s, S, n, G, d_cell.as<unsigned>(), d_idx.as<int>());
    int bits = 1;
    while ((1ll << bits) < ncells) ++bits;
    size_t tmp_bytes = 0;
    WLSQM_HIP_CHECK(hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, d_cell.as<unsigned>(), d_cell2.as<unsigned>(),
                                                       d_idx.as<int>(), d_perm.as<int>(), (int)n, 0, bits, s));
    if ((rc = d_tmp.alloc(tmp_bytes ? tmp_bytes : 16))) return rc;
    WLSQM_HIP_CHECK(hipcub::DeviceRadixSort::SortPairs(d_tmp.p, tmp_bytes, d_cell.as<unsigned>(), d_cell2.as<unsigned>(),
                                                       d_idx.as<int>(), d_perm.as<int>(), (int)n, 0, bits, s));
    hipLaunchKernelGGL(knn_start_kernel, dim3((unsigned)((ncells + 1 + 255) / 256)), dim3(256), 0, s, d_cell2.as<unsigned>(), n,
                       ncells, d_start.as<int>());
    hipLaunchKernelGGL(knn_gather_kernel, dim3(blocks), dim3(256), 0, s, S, d_perm.as<int>(), n, dimension, d_Ss.as<double>());
    WLSQM_HIP_CHECK(hipGetLastError());
    WLSQM_HIP_CHECK(hipStreamSynchronize(s));      // the sort's temporaries are freed on return
    return WLSQM_OK;
}

// k nearest other points within squared distance r2max of every point; row_stride slots per row of hoods
static int neighbour_search(int dimension, int64_t npoints, const double* S, int k, double r2max, int row_stride,
                            int32_t* hoods, int32_t* counts, int device, void* stream_, int64_t nquery = -1) {
    if (nquery < 0) nquery = npoints;
    if (nquery > npoints) { set_error("nquery must be <= npoints"); return WLSQM_EVALUE; }
    if (dimension < 1 || dimension > 3) { set_error("dimension must be 1, 2 or 3"); return WLSQM_EVALUE; }
    if (!S || !hoods) { set_error("null array"); return WLSQM_EVALUE; }
    if (k < 1 || npoints < 2 || (int64_t)k > npoints - 1) { set_error("k must be in 1 .. npoints - 1"); return WLSQM_EVALUE; }
    if (npoints > 0x7fffffffLL) { set_error("at most 2^31 - 1 points"); return WLSQM_EVALUE; }
    const size_t lds = (size_t)k * 64 * (sizeof(double) + sizeof(int));
    if (lds > 160 * 1024) { set_error("k too large for the LDS-resident candidate lists (k <= 213)"); return WLSQM_EVALUE; }
    DeviceScope scope; int rc = scope.enter(device);
    if (rc != WLSQM_OK) return rc;
    hipStream_t s = (hipStream_t)stream_;
    const long long n = npoints;
    GridIndex grid;
    if ((rc = grid.build(dimension, npoints, S, s))) return rc;
    const KnnGrid G = grid.G;
    const unsigned qblocks = (unsigned)((n + 63) / 64);
#define KNN_LAUNCH(D)                                                                                                   \
    {                                                                                                                   \
        auto kern = knn_query_kernel<D>;                                                                                \
        if (lds > 64 * 1024)                                                                                            \
            WLSQM_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
        hipLaunchKernelGGL(kern, dim3(qblocks), dim3(64), lds, s, grid.d_Ss.as<double>(), grid.d_perm.as<int>(),        \
                           grid.d_start.as<int>(), n, k, G, r2max, row_stride, hoods, counts, (long long)nquery);     \
    }
    if (dimension == 1) KNN_LAUNCH(1) else if (dimension == 2) KNN_LAUNCH(2) else KNN_LAUNCH(3)
#undef KNN_LAUNCH
    WLSQM_HIP_CHECK(hipGetLastError());
    WLSQM_HIP_CHECK(hipStreamSynchronize(s));      // the grid's buffers are freed on return
    return WLSQM_OK;
}

namespace wlsqm {

// Nearest data point of every query point (queries are NOT part of the cloud: no self-exclusion): ExpertSolver.interpolate's
// model lookup (expert.pyx:830-895 queries a cKDTree of the origins xi).  One lane per query; ties go to the smaller index.
template <int DIM>
__global__ __launch_bounds__(64) void nearest_kernel(const double* __restrict__ Ss, const int* __restrict__ perm,
                                                     const int* __restrict__ start, const double* __restrict__ X, long long sx,
                                                     long long nq, KnnGrid G, long long* __restrict__ out) {
    const long long q = (long long)blockIdx.x * 64 + threadIdx.x;
    if (q >= nq) return;
    double x[DIM]; int cq[DIM];
#pragma unroll
    for (int m = 0; m < DIM; ++m) { x[m] = X[q * sx + m]; cq[m] = cell_coord(x[m], G, m); }
    int rmax = 0;
#pragma unroll
    for (int m = 0; m < DIM; ++m) rmax = max(rmax, max(cq[m], G.g[m] - 1 - cq[m]));
    double best = DBL_MAX; int best_idx = 0x7fffffff;
    for (int r = 0; r <= rmax; ++r) {
        const int z0 = DIM > 2 ? -r : 0, z1 = DIM > 2 ? r : 0, y0 = DIM > 1 ? -r : 0, y1 = DIM > 1 ? r : 0;
        for (int dz = z0; dz <= z1; ++dz) {
            const int cz = DIM > 2 ? cq[2] + dz : 0;
            if (DIM > 2 && (cz < 0 || cz >= G.g[2])) continue;
            for (int dy = y0; dy <= y1; ++dy) {
                const int cy = DIM > 1 ? cq[1] + dy : 0;
                if (DIM > 1 && (cy < 0 || cy >= G.g[1])) continue;
                const bool face = (DIM > 2 && (dz == -r || dz == r)) || (DIM > 1 && (dy == -r || dy == r));
                const int step = (face || r == 0) ? 1 : 2 * r;
                for (int dx = -r; dx <= r; dx += step) {
                    const int cx = cq[0] + dx;
                    if (cx < 0 || cx >= G.g[0]) continue;
                    long long cid = cx;
                    if (DIM > 1) cid += (long long)G.g[0] * cy;
                    if (DIM > 2) cid += (long long)G.g[0] * G.g[1] * cz;
                    for (int pos = start[cid]; pos < start[cid + 1]; ++pos) {
                        double d2 = 0.0;
#pragma unroll
                        for (int m = 0; m < DIM; ++m) { const double d = Ss[(long long)pos * DIM + m] - x[m]; d2 += d * d; }
                        const int idx = perm[pos];
                        if (d2 < best || (d2 == best && idx < best_idx)) { best = d2; best_idx = idx; }
                    }
                }
            }
        }
        if (best < DBL_MAX) {
            // unvisited points lie outside the block [cq - r, cq + r] of cells (the query itself may lie outside the grid)
            double reach = DBL_MAX;
#pragma unroll
            for (int m = 0; m < DIM; ++m) {
                if (cq[m] - r > 0) reach = fmin(reach, x[m] - (G.lo[m] + (cq[m] - r) * G.cell[m]) - 1e-9 * G.cell[m]);
                if (cq[m] + r < G.g[m] - 1) reach = fmin(reach, (G.lo[m] + (cq[m] + r + 1) * G.cell[m]) - x[m] - 1e-9 * G.cell[m]);
            }
            if (reach == DBL_MAX || (reach > 0.0 && best <= reach * reach)) break;
        }
    }
    out[q] = best_idx;
}

// nearest data point (index into S, int64) of each of the nquery device-resident points X[nquery, x_stride]
int nearest_search(int dimension, int64_t ndata, const double* S, int64_t nquery, const double* X, int64_t x_stride,
                   long long* out, hipStream_t s) {
    if (ndata < 1 || ndata > 0x7fffffffLL) { set_error("1 .. 2^31 - 1 data points"); return WLSQM_EVALUE; }
    if (nquery <= 0) return WLSQM_OK;
    GridIndex grid;
    int rc = grid.build(dimension, ndata, S, s);
    if (rc != WLSQM_OK) return rc;
    const unsigned qblocks = (unsigned)((nquery + 63) / 64);
#define NEAREST_LAUNCH(D)                                                                                               \
    hipLaunchKernelGGL(nearest_kernel<D>, dim3(qblocks), dim3(64), 0, s, grid.d_Ss.as<double>(), grid.d_perm.as<int>(),   \
                       grid.d_start.as<int>(), X, (long long)x_stride, (long long)nquery, grid.G, out);
    if (dimension == 1) NEAREST_LAUNCH(1) else if (dimension == 2) NEAREST_LAUNCH(2) else NEAREST_LAUNCH(3)
#undef NEAREST_LAUNCH
    WLSQM_HIP_CHECK(hipGetLastError());
    WLSQM_HIP_CHECK(hipStreamSynchronize(s));
    return WLSQM_OK;
}

// ExpertSolver.interpolate(mode='continuous') (expert.pyx:898-985: query_ball_tree of the origins, then the weighted average
// of the models within r).  One lane per evaluation point; the ball's bounding block of cells is walked directly (cell_coord
// is monotone, so every origin within r lies in a cell of [cell(x - r), cell(x + r)] on each axis), no lists are built.
template <int DIM>
__global__ __launch_bounds__(64) void interp_ball_kernel(const double* __restrict__ Ss, const int* __restrict__ perm,
                                                         const int* __restrict__ start, KnnGrid G, const InterpParams q,
                                                         double r) {
    const long long m = (long long)blockIdx.x * 64 + threadIdx.x;
    if (m >= q.nx) return;
    double xp[DIM]; int c0[3] = {0, 0, 0}, c1[3] = {0, 0, 0};
#pragma unroll
    for (int c = 0; c < DIM; ++c) {
        xp[c] = q.x[m * q.sx + c];
        c0[c] = cell_coord(xp[c] - r, G, c); c1[c] = cell_coord(xp[c] + r, G, c);
    }
    const double r2 = r * r;
    double acc = 0.0, sum_w = 0.0;
    for (int cz = c0[2]; cz <= c1[2]; ++cz)
        for (int cy = c0[1]; cy <= c1[1]; ++cy) {
            // the cells of one x-run are consecutive in the sorted order: one range of positions
            const long long row = (long long)G.g[0] * (cy + (long long)G.g[1] * cz);
            const int p0 = start[row + c0[0]], p1 = start[row + c1[0] + 1];
            for (int pos = p0; pos < p1; ++pos) {
                double d2 = 0.0;
#pragma unroll
                for (int c = 0; c < DIM; ++c) { const double d = Ss[(long long)pos * DIM + c] - xp[c]; d2 += d * d; }
                if (d2 > r2) continue;
                const double v = eval_model<DIM>(q, perm[pos], xp, nullptr);
                const double t = 1.0 - sqrt(d2 / r2);     // expert.pyx:45-46: alpha = 0, beta = 1
                const double w = t * t;
                acc += w * v; sum_w += w;
            }
        }
    q.out[m] = acc / sum_w;                               // no model in range: 0/0 = NaN, as the reference
}

int interp_continuous(int dimension, const InterpParams& q, double r, hipStream_t s) {
    if (q.nmodels < 1 || q.nmodels > 0x7fffffffLL) { set_error("1 .. 2^31 - 1 models"); return WLSQM_EVALUE; }
    if (!(r > 0.0)) { set_error("r must be positive"); return WLSQM_EVALUE; }
    if (q.sxi != dimension) { set_error("origins must be contiguous"); return WLSQM_EVALUE; }
    if (q.nx <= 0) return WLSQM_OK;
    GridIndex grid;
    int rc = grid.build(dimension, q.nmodels, q.xi, s);
    if (rc != WLSQM_OK) return rc;
    const unsigned blocks = (unsigned)((q.nx + 63) / 64);
#define BALL_LAUNCH(D)                                                                                                    \
    hipLaunchKernelGGL(interp_ball_kernel<D>, dim3(blocks), dim3(64), 0, s, grid.d_Ss.as<double>(), grid.d_perm.as<int>(), \
                       grid.d_start.as<int>(), grid.G, q, r);
    if (dimension == 1) BALL_LAUNCH(1) else if (dimension == 2) BALL_LAUNCH(2) else BALL_LAUNCH(3)
#undef BALL_LAUNCH
    WLSQM_HIP_CHECK(hipGetLastError());
    WLSQM_HIP_CHECK(hipStreamSynchronize(s));      // the grid index dies with this scope
    return WLSQM_OK;
}

}  // namespace wlsqm

extern "C" int wlsqm_hip_knn_device(int dimension, int64_t npoints, const double* S, int k, int32_t* hoods, int device,
                                    void* stream) {
    return neighbour_search(dimension, npoints, S, k, DBL_MAX, k, hoods, nullptr, device, stream);
}

extern "C" int wlsqm_hip_knn_subset_device(int dimension, int64_t npoints, const double* S, int k, int64_t nquery, int32_t* hoods,
                                           int device, void* stream) {
    return neighbour_search(dimension, npoints, S, k, DBL_MAX, k, hoods, nullptr, device, stream, nquery);
}
extern "C" int wlsqm_hip_ball_device(int dimension, int64_t npoints, const double* S, double radius, int max_nk,
                                     int32_t* hoods, int32_t* nk, int device, void* stream) {
    if (!(radius > 0.0)) { set_error("radius must be positive"); return WLSQM_EVALUE; }
    if (max_nk < 1 || !nk) { set_error("max_nk must be >= 1 and nk non-null"); return WLSQM_EVALUE; }
    const int k = (int)std::min<int64_t>(max_nk, npoints - 1);
    return neighbour_search(dimension, npoints, S, k, radius * radius, max_nk, hoods, nk, device, stream);
}

extern "C" int wlsqm_hip_nearest_device(int dimension, int64_t ndata, const double* S, int64_t nquery, const double* X,
                                        int64_t x_stride, int64_t* nearest, int device, void* stream) {
    if (dimension < 1 || dimension > 3) { set_error("dimension must be 1, 2 or 3"); return WLSQM_EVALUE; }
    if (!S || !X || !nearest) { set_error("null array"); return WLSQM_EVALUE; }
    DeviceScope scope; int rc = scope.enter(device);
    if (rc != WLSQM_OK) return rc;
    return nearest_search(dimension, ndata, S, nquery, X, x_stride, reinterpret_cast<long long*>(nearest), (hipStream_t)stream);
}
