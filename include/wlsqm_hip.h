/* wlsqm_hip.h — C ABI of libwlsqm_hip.so, the MI355X (gfx950) batched WLSQM fitter.
 *
 * This is the drop-in boundary for ONE hot path of Technologicat/python-wlsqm:
 * the per-case "assemble weighted normal equations -> factor -> solve" loop behind
 * wlsqm.fitter.simple.fit_*D_many[_parallel] / fit_*D_iterative_many[_parallel]
 * and wlsqm.fitter.expert.ExpertSolver.prepare()/solve().  The reference has no
 * FFI layer of its own (it is Cython); each entry point below names the reference
 * interface it replaces (file:line relative to the reference repo).  Plain pointers
 * and sizes only; no torch / numpy types.  All functions return 0 on success or a
 * negative WLSQM_E* code; wlsqm_hip_last_error() gives a thread-local message.
 *
 * Strides are in ELEMENTS of the array's dtype (not bytes).  Arrays follow the
 * reference's layout contract (simple.pyx:131-159): the innermost axis of xk, xi,
 * fi and sens must be contiguous, every other axis may have any stride.
 */
#ifndef WLSQM_HIP_H
#define WLSQM_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- constants: wlsqm/fitter/defs.pyx:69-75 ---- */
#define WLSQM_ALGO_BASIC      1
#define WLSQM_ALGO_ITERATIVE  2
#define WLSQM_WEIGHT_UNIFORM  1
#define WLSQM_WEIGHT_CENTER   2

/* ---- error codes (mapped by the Python shim to the reference's exception classes) ---- */
#define WLSQM_OK            0
#define WLSQM_EVALUE       -1   /* -> ValueError   (bad dimension/order/length; expert.pyx:131-159, infra.pyx:311-313) */
#define WLSQM_ERUNTIME     -2   /* -> RuntimeError (solve before prepare, expert.pyx:493-494; HIP failure) */
#define WLSQM_EMEMORY      -3   /* -> MemoryError  (infra.pyx:237,243,363,819) */
#define WLSQM_ENODEVICE    -4   /* -> RuntimeError: no HIP device; there is NO CPU fallback in this library */

const char* wlsqm_hip_last_error(void);
/* Diagnostics: name of the kernel family the last fit launch of this thread dispatched to ("tile", "tile-gather",
 * "tile1", "tile1-extras", "tilek", "moment", "rows", "lane", "wave", "solve-many"; "" before the first launch).  The tests
 * use it to make sure a fast path is what they exercise. */
const char* wlsqm_hip_last_kernel(void);

/* Number of usable HIP devices (0 if none).  Never initialises a context on failure. */
int wlsqm_hip_device_count(void);

/* ---- host-side index contract (pure integer work, bit-exact with the reference) ---- */

/* infra.pyx:67-112  number_of_dofs: returns no, or -1 (bad dimension) / -2 (bad order). */
int wlsqm_hip_number_of_dofs(int dimension, int order);
/* infra.pyx:119-121 number_of_reduced_dofs = n - popcountll(mask) (bits >= n are NOT masked off). */
int wlsqm_hip_number_of_reduced_dofs(int n, int64_t mask);
/* infra.pyx:145-200 remap: fills o2r[n], r2o[n] with -1 sentinels, returns nr. */
int wlsqm_hip_remap(int32_t* o2r, int32_t* r2o, int n, int64_t mask);

/* ---- batch description shared by the fit entry points ----
 * Mirrors the argument list of simple.pyx:953-957 generic_fit_basic_many_parallel
 * (and :1065-1069 for the iterative variant).  dimension==1: xk is [ncases,max_nk]
 * (xk_stride_k is the neighbour stride), xi is [ncases].
 */
typedef struct wlsqm_batch {
    int32_t dimension;            /* 1, 2 or 3 */
    int32_t do_sens;              /* simple.pyx: do_sens */
    int64_t ncases;
    const double*  xk;   int64_t xk_stride_case, xk_stride_k;   /* [j,k,m], m contiguous */
    const double*  fk;   int64_t fk_stride_case, fk_stride_k;   /* [j,k] */
    const int32_t* nk;   int64_t nk_stride;                     /* [j] */
    const double*  xi;   int64_t xi_stride_case;                /* [j,m], m contiguous */
    double*        fi;   int64_t fi_stride_case;                /* [j,n] in/out: knowns read, unknowns written */
    double*        sens; int64_t sens_stride_case, sens_stride_k; /* [j,k,n] or NULL */
    const int32_t* order;            int64_t order_stride;      /* [j] */
    const int64_t* knowns;           int64_t knowns_stride;     /* [j] */
    const int32_t* weighting_method; int64_t wm_stride;         /* [j] */
    int32_t iterative;            /* 0: ALGO_BASIC body (impl.pyx:731 solve); 1: impl.pyx:986 solve_iterative */
    int32_t max_iter;             /* simple.pyx:112 default 10 */
    int64_t max_nk;               /* extent of the k axis of xk/fk/sens (upper bound on nk[j]) */
} wlsqm_batch;

/* fit_{1,2,3}D_many_parallel / _iterative_many_parallel on HOST arrays
 * (simple.pyx:192-234, 379-421, 562-604; drivers :953-1058, :1065-1170).
 * Copies the batch to the device, runs the HIP kernels, commits fi (and sens) back.
 * Aliasing guarantee of simple.pyx:1010-1019 holds: all inputs are read before any
 * output is committed.  *iterations_out (nullable) receives the return value of the
 * reference function: 0 for basic, max refinement iterations otherwise.
 * device: HIP device ordinal. */
int wlsqm_hip_fit_many_host(const wlsqm_batch* b, int device, int32_t* iterations_out);

/* Same computation on DEVICE-RESIDENT arrays (all pointers in `b` are device pointers of
 * `device`), enqueued on `stream` (a hipStream_t, NULL = default stream).  With iterations_out == NULL the
 * call only enqueues kernels (no allocation, no host synchronisation: it can be captured into a hipGraph, refinement
 * included); with iterations_out != NULL (a host pointer) and b->iterative it allocates a stream-ordered 4-byte counter,
 * copies it back and synchronises `stream`.  The arrays must cover b->ncases rows (or every row `case_index` names),
 * b->max_nk neighbour slots per row and number_of_dofs(dimension, order_uniform) columns of fi (and of sens); nk[j]
 * is clamped to b->max_nk by every kernel.
 * `order_uniform` >= 0 states that every case has this order (required: the kernels are
 * specialised per (dimension, order); heterogeneous batches are bucketed by the caller,
 * one call per order with `case_index`).  `case_index` (device, nullable): the ncases_sel
 * case numbers to process; NULL = all b->ncases cases in order.
 * fi is updated in place (knowns untouched); the caller guarantees fk/xk do not alias fi. */
int wlsqm_hip_fit_many_device(const wlsqm_batch* b, int device, void* stream, int order_uniform,
                              const int64_t* case_index, int64_t ncases_sel, int32_t* iterations_out);

/* Per-case polynomial orders on the device-resident path (the reference takes a per-case `order` array, simple.pyx:379-381):
 * as wlsqm_hip_fit_many_device, with order_dev[j * order_stride] (int32, DEVICE memory) instead of order_uniform.  The cases are
 * bucketed by order ON THE DEVICE (one counting kernel, index lists in stream-ordered scratch) and every bucket is launched
 * with its size left in device memory: no host synchronisation, nothing but kernel launches and stream-ordered allocations
 * unless iterations_out is given.  b->order is not read.  max_order (0..4) is the largest order the caller's fi / sens rows are
 * wide enough for (the reference's "(ncases, >= max no)" rule, simple.pyx:379-381): cases whose order is not 0..max_order are
 * left untouched and their buckets are never launched.  The buckets are STABLE (ascending case number: a stable counting
 * partition, no atomics on positions), so results are bit-identical from run to run. */
int wlsqm_hip_fit_many_device_orders(const wlsqm_batch* b, int device, void* stream, const int32_t* order_dev, int64_t order_stride,
                                     int max_order, int32_t* iterations_out);

/* ---- numerics mode (extension) ----
 * 0 (default): the fast kernels — moment form, neighbour sums split over lanes, FMA contraction, LDL^T: results agree with the
 *    reference to kappa * eps rounding.
 * 1: reference-order arithmetic (csrc/fit_strict.hip): every floating-point operation of make_c_nD / Case_make_weights / make_A
 *    (impl.pyx:70-602, infra.pyx:668-702), rescale_ruiz2001_c (lapackdrivers.pyx:553-623), dgetrf / dgetrs (:1628-1665), solve
 *    and solve_iterative (impl.pyx:731-1083) in the reference's order, one lane per case, IEEE divide and sqrt, no FMA contraction.
 *    Applies to every entry point that fits (wlsqm_hip_fit_many_*, wlsqm_hip_fit_cloud_device, wlsqm_hip_expert_solve*; a stacked
 *    solve runs one fit per field).  Several times slower; for validation against the reference at 1e-10.
 * 2: accurate (csrc/fit_accurate.hip): mode 1 with ONE change — the normal matrix of make_A (impl.pyx:566-602) is assembled from
 *    its upper triangle, entry (j, m) = sum_k (w c_m) c_j for m >= j, and mirrored; the equilibration then sees a symmetric matrix
 *    (row and column scales coincide bit for bit) and runs one pass per sweep.  Quotients and roots are correctly rounded (the
 *    compiler's IEEE sequences without their range scaling where every operand is checked to be in range, the full sequences
 *    otherwise).  Applies to the basic fits of the 2D / 3D systems up to 10 unknowns, per case, for the cases without a known
 *    DOF; every other case of an accurate-mode call runs mode 1.  Within 1e-10 of the reference on every column of BASELINE
 *    configs[1] and configs[4] (as mode 1), at the fast kernels' order of magnitude in time.
 * The mode belongs to the calling thread; its initial value is the environment variable WLSQM_HIP_STRICT (unset / 0: fast; 1;
 * 2 or "accurate").  wlsqm_hip_set_strict returns the previous mode. */
int wlsqm_hip_set_strict(int mode);
int wlsqm_hip_get_strict(void);

/* Round 6: what the calling thread knows about the neighbour counts nk[j] of the DENSE DEVICE-RESIDENT batches it hands over (only
 * entries below nk[j] of a row are touched: simple.pyx:147; the reference's own harness passes ball-query rows of 100 slots with 30..100
 * valid entries, examples/wlsqm_example.py:103-133):
 *   1 (default) every case fills its row, or nearly: the plain kernels;
 *   2 ragged: the staged kernels run the copy whose waves move only the chunks their own 64 cases need (csrc/fit_stage.hip);
 *   0 unknown: the plain kernels mark the waves none of whose cases reaches the row's last chunk and the ragged copy runs those
 *     behind them (an idle launch, ~2 us, for a batch of full rows).
 * A hint, never a condition of correctness: every setting returns the same bits.  The host entry points (wlsqm_hip_fit_many_host)
 * look at the counts themselves and ignore it.  Returns the previous value. */
int wlsqm_hip_set_row_hint(int hint);

/* Round 6: are the neighbours of every row sorted by distance (1, the default: what a k-nearest-neighbour search returns — scipy's cKDTree.query,
 * wlsqm_hip_knn — and every BASELINE config) or in no order (0: a ball query)?  The weights need the largest squared distance of a case before
 * the first term can be summed (infra.pyx:668-702); sorted rows give it away (the last neighbour), unsorted rows cost a pass of their own.
 * The staged kernels of the small dense systems have a form for either (csrc/fit_stage.hip) and the CALLER's word picks it — the time of a
 * call is a function of its arguments, not of what ran before it on the stream.  A hint, never a condition of correctness: the kernels verify
 * the order bit for bit and the results are the same bits either way.  Belongs to the calling thread; returns the previous value. */
int wlsqm_hip_set_order_hint(int sorted);

/* Test hook of the strict mode: runs the reference-order fit of `b` (uniform order) and also stores the reference's intermediates,
 * for bit-for-bit comparison with values captured from the reference (tests/golden/sweep_*.npz): w[j * w_stride + k]
 * (Case_make_weights), the unscaled A and the scaled LU factor as nr x nr Fortran-order blocks at [j * mat_stride]
 * (make_A impl.pyx:566-602; dgetrf), row_scale / col_scale (Ruiz) and the 1-based ipiv at [j * vec_stride + i]. */
int wlsqm_hip_strict_intermediates_device(const wlsqm_batch* b, int device, void* stream, int order_uniform,
                                          double* w, int64_t w_stride, double* A, double* LU, int64_t mat_stride,
                                          double* row_scale, double* col_scale, int32_t* ipiv, int64_t vec_stride);

/* Index-based ("cloud") variant of wlsqm_hip_fit_many_device — an EXTENSION of the reference surface (the step
 * before the path: examples/expertsolver_example.py:91-92 builds xk = S[hoods], fk = F[hoods] on the host).
 * The kernels gather the neighbour rows themselves from the device-resident point tables S[npoints, dim] and
 * F[npoints] through hoods[ncases, max_nk] (int32), which cuts the algorithmic HBM bytes per fit from
 * 8 nk (dim+1) to 4 nk.  xi of case j is S[point_index ? point_index[j] : j].  All cases have polynomial order
 * `order`; nk / knowns / weighting_method are per-case device arrays (unit stride); fi is in/out as usual.
 * Only the slots k < nk[j] of a hoods row are dereferenced: the padding of a ragged row may hold anything (-1, npoints,
 * ...), as with the reference's dense arrays (simple.pyx:147).  Slots k < nk[j] must be valid point numbers.
 * iterations_out as for wlsqm_hip_fit_many_device (NULL: nothing but kernel launches).
 * Every (dimension, order); with sensitivities or refinement only systems with no <= 15 DOFs (not 3D order 3/4). */
int wlsqm_hip_fit_cloud_device(int dimension, int order, int64_t ncases, int64_t max_nk,
                               const double* S, const double* F, const int32_t* hoods, int64_t hoods_stride_case,
                               const int32_t* point_index, const int32_t* nk, const int64_t* knowns,
                               const int32_t* weighting_method, double* fi, int64_t fi_stride_case,
                               double* sens, int64_t sens_stride_case, int64_t sens_stride_k, int do_sens,
                               int iterative, int max_iter, int device, void* stream, int32_t* iterations_out);

/* ---- ExpertSolver (expert.pyx:66-781): handle-based prepare-once / solve-many ---- */
typedef struct wlsqm_expert wlsqm_expert;

/* expert.pyx:92-264 __init__: host arrays nk/order/knowns/weighting_method of length ncases. */
int wlsqm_hip_expert_create(wlsqm_expert** out, int device, int dimension, int64_t ncases,
                            const int32_t* nk, const int32_t* order, const int64_t* knowns,
                            const int32_t* weighting_method, int algorithm, int do_sens, int max_iter);
/* expert.pyx:92-93, 112-126, 163-189, 243-252 ExpertSolver(..., host=other): "guest mode".  The new solver shares the
 * host's device-resident geometry and per-case metadata (nk, order, knowns, weighting_method) instead of holding a copy,
 * and owns only its field buffers (fk, fi, sens).  The host must be prepared (WLSQM_ERUNTIME otherwise, as the
 * reference's RuntimeError).  The shared state is reference-counted: destroying the host first is safe. */
int wlsqm_hip_expert_create_guest(wlsqm_expert** out, wlsqm_expert* host, int algorithm, int do_sens, int max_iter);
/* expert.pyx:309-426 prepare(xi, xk): host arrays; geometry is uploaded and kept device-resident.  On a guest the
 * arrays are ignored (expert.pyx:350-352: the host's geometry is used) and only the host's ready state is checked. */
int wlsqm_hip_expert_prepare(wlsqm_expert* h, const double* xi, int64_t xi_stride_case,
                             const double* xk, int64_t xk_stride_case, int64_t xk_stride_k, int64_t max_nk);
/* Extension: prepare() from device-resident arrays (xi[ncases, xi_stride_case], xk[ncases, xk_stride_case / dimension,
 * dimension] with xk_stride_k == dimension); the geometry is copied device-to-device on `stream` into the solver's own block.
 * The calling thread's wlsqm_hip_set_order_hint is recorded with the geometry (the host form looks at the rows itself). */
int wlsqm_hip_expert_prepare_device(wlsqm_expert* h, void* stream, const double* xi, int64_t xi_stride_case,
                                    const double* xk, int64_t xk_stride_case, int64_t xk_stride_k);
/* expert.pyx:467-655 solve(fk, fi, sens): host arrays; returns max iterations via *iterations_out. */
int wlsqm_hip_expert_solve(wlsqm_expert* h, const double* fk, int64_t fk_stride_case, int64_t fk_stride_k,
                           double* fi, int64_t fi_stride_case,
                           double* sens, int64_t sens_stride_case, int64_t sens_stride_k,
                           int32_t* iterations_out);
/* Device-resident variant of solve(): fk [ncases,max_nk] and fi [ncases,fi_stride_case] are device
 * pointers (contiguous k axis); enqueued on `stream`; no host synchronisation.  fi must be at least max_no doubles wide
 * (every row is written at fi_stride_case pitch).  As in the reference, interpolate() afterwards evaluates the coefficients
 * of the LATEST solve of any kind: after this call that is the caller's `fi` array itself (no copy is made), so it must
 * stay allocated until the next solve or until interpolation is no longer used. */
int wlsqm_hip_expert_solve_device(wlsqm_expert* h, void* stream, const double* fk, int64_t fk_stride_case,
                                  double* fi, int64_t fi_stride_case);
/* Extension: the neighbour search the reference's examples run on the host before calling the fitter
 * (examples/expertsolver_example.py:48-66: cKDTree(x).query(x, 1 + nk), the point itself dropped;
 * examples/wlsqm_example.py:103-133).  For every point of the device-resident cloud S[npoints, dimension] the k nearest
 * OTHER points, ascending by (distance, index), into the device array hoods[npoints, k] (int32) — the `hoods` argument of
 * wlsqm_hip_fit_cloud_device.  Exact (uniform-grid search with a provable stop test).  1 <= k <= min(npoints - 1, 213).
 * Synchronises `stream` before returning. */
int wlsqm_hip_knn_device(int dimension, int64_t npoints, const double* S, int k, int32_t* hoods, int device, void* stream);
/* Extension: the same search asked only by the FIRST nquery points of the cloud; the remaining npoints - nquery points are
 * candidates only (hoods[nquery, k], indices into S).  The rank of a partitioned cloud searches its own points against its
 * own points plus a halo band of its neighbours' (wlsqm/sharded.py) instead of the global cloud. */
int wlsqm_hip_knn_subset_device(int dimension, int64_t npoints, const double* S, int k, int64_t nquery, int32_t* hoods,
                                int device, void* stream);
/* Extension: the radius form of the same search (examples/wlsqm_example.py:103-133: query_ball_point(x, r), at most
 * max_nk neighbours kept).  For every point the other points within `radius`, nearest first, at most max_nk of them:
 * hoods[npoints, max_nk] (int32; unused slots hold the point's own index) and nk[npoints] (int32, the counts). */
int wlsqm_hip_ball_device(int dimension, int64_t npoints, const double* S, double radius, int max_nk,
                          int32_t* hoods, int32_t* nk, int device, void* stream);
/* Extension: nearest point of the device-resident cloud S[ndata, dimension] for each of the device-resident query points
 * X[nquery, x_stride] (queries are not members of the cloud): nearest[nquery] (int64 device array, indices into S; ties go
 * to the smaller index).  What ExpertSolver.interpolate(mode='nearest') needs (cKDTree.query in expert.pyx:830-895). */
int wlsqm_hip_nearest_device(int dimension, int64_t ndata, const double* S, int64_t nquery, const double* X,
                             int64_t x_stride, int64_t* nearest, int device, void* stream);

/* Extension: build the stored solution operator of the prepared geometry now (8 x 16 x K' bytes per case, K' = the neighbour
 * slots rounded up to a multiple of 8; shared with guests), so that later stacked solves only enqueue work — e.g. before a stream
 * capture.  Without this call the first wlsqm_hip_expert_solve_many[_device] that wants the operator builds it and synchronises
 * `stream`.  *built = 1 when the operator exists afterwards, 0 when the shape has none (more than 15 unknowns, more than 64 or an
 * odd number of slots, more than 4 knowns per case, mixed orders, ALGO_ITERATIVE) or it does not fit the free device memory: the
 * stacked solve then takes its other kernels. */
int wlsqm_hip_expert_prepare_operator(wlsqm_expert* h, void* stream, int* built);
/* Extension (no reference counterpart; BASELINE config 4 "prepare once + 256 RHS solves"): nrhs fields on the prepared
 * geometry in one call.  Equivalent to nrhs calls of expert.pyx:467-655 solve() with ALGO_BASIC and no sensitivities,
 * in one launch: stacks of 64 fields or more, and every stack on a shape with more than 6 unknowns or 32 neighbour slots,
 * apply the stored solution operator (wlsqm_hip_expert_prepare_operator; csrc/solve_op.hip, one batched GEMM on the matrix
 * cores); short stacks on small shapes share the geometry work (weights, monomials, normal matrix, factorisation) inside the
 * launch (csrc/solve_many.hip); anything else runs one fused launch per field.  Device variant: fk[nrhs][ncases][max_nk]
 * and fi[nrhs][ncases][fi_stride_case] are device pointers with the given element strides (k contiguous), enqueued on
 * `stream`.  Host variant: strided host arrays, transferred in chunks of right-hand sides. */
int wlsqm_hip_expert_solve_many_device(wlsqm_expert* h, void* stream, int64_t nrhs,
                                       const double* fk, int64_t fk_stride_rhs, int64_t fk_stride_case,
                                       double* fi, int64_t fi_stride_rhs, int64_t fi_stride_case);
int wlsqm_hip_expert_solve_many(wlsqm_expert* h, int64_t nrhs,
                                const double* fk, int64_t fk_stride_rhs, int64_t fk_stride_case, int64_t fk_stride_k,
                                double* fi, int64_t fi_stride_rhs, int64_t fi_stride_case);
/* expert.pyx:429-464 conds(): 2-norm condition number of the Ruiz-scaled reduced matrix of every case
 * (impl.pyx:662-682), out[ncases] on the host.  Diagnostics path (one-sided Jacobi SVD per case). */
int wlsqm_hip_expert_conds(wlsqm_expert* h, double* out);
/* expert.pyx:687-781 interpolate(): evaluate the models of the last solve() (or their derivative `diff`, a DOF
 * index) at nx host points x[nx, dim].  mode='nearest': I[nx] (host) names the model per point (expert.pyx:830-895;
 * the nearest-origin search itself stays on the host, scipy cKDTree, as in the reference); mode='continuous':
 * CSR lists list_off[nx+1], list_idx[] of the models within radius r of each point (expert.pyx:898-985). */
int wlsqm_hip_expert_interpolate(wlsqm_expert* h, const double* x, int64_t x_stride, int64_t nx, const int64_t* I,
                                 const int64_t* list_off, const int64_t* list_idx, double r, int diff, double* out);
/* mode='nearest' with the nearest-origin search on the device too (the reference queries a cKDTree of the origins,
 * expert.pyx:830-895): out[nx] the values, I_out[nx] (nullable) the model chosen for every point. */
int wlsqm_hip_expert_interpolate_nearest(wlsqm_expert* h, const double* x, int64_t x_stride, int64_t nx, int diff,
                                         double* out, int64_t* I_out);
/* mode='continuous' with the ball search on the device too (the reference builds the lists with
 * cKDTree.query_ball_tree, expert.pyx:898-903): out[nx] = weighted average (weights (1 - d/r)^2, expert.pyx:45-46)
 * of the models whose origin lies within r of the point; NaN where there is none. */
int wlsqm_hip_expert_interpolate_continuous(wlsqm_expert* h, const double* x, int64_t x_stride, int64_t nx, double r,
                                            int diff, double* out);
/* expert.pyx:289-306 memory_used(): (bytes in use, bytes reserved) of the device-side state; a guest counts only
 * what it owns (the shared geometry is the host's). */
int wlsqm_hip_expert_memory_used(const wlsqm_expert* h, int64_t* used, int64_t* total);
/* expert.pyx:267-286 __del__ */
int wlsqm_hip_expert_destroy(wlsqm_expert* h);

/* ---- model evaluation (the step after the path) ---- */
/* interp.pyx:34-143 interpolate_fit: one model (xi[dim], fi[no], order) or its derivative `diff` at nx host points. */
int wlsqm_hip_interpolate_fit_host(int dimension, int order, const double* xi, const double* fi,
                                   const double* x, int64_t x_stride, int64_t nx, int diff, double* out, int device);

/* ---- measurement hooks used by bench.py (not part of the reference surface) ---- */
/* Runs `reps` back-to-back launches of the fit kernel for batch `b` (device-resident, uniform
 * order) on `stream`, bracketed by HIP events on that stream; returns the mean kernel time in
 * milliseconds in *ms_out. */
int wlsqm_hip_time_fit_device(const wlsqm_batch* b, int device, void* stream, int order_uniform,
                              int reps, float* ms_out);
/* Same for the index-based path (basic algorithm, no sensitivities). */
int wlsqm_hip_time_fit_cloud_device(int dimension, int order, int64_t ncases, int64_t max_nk,
                                    const double* S, const double* F, const int32_t* hoods, int64_t hoods_stride_case,
                                    const int32_t* point_index, const int32_t* nk, const int64_t* knowns,
                                    const int32_t* weighting_method, double* fi, int64_t fi_stride_case,
                                    int device, void* stream, int reps, float* ms_out);

#ifdef __cplusplus
}
#endif
#endif /* WLSQM_HIP_H */
