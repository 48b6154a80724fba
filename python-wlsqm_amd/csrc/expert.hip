// expert.hip — handle-based ExpertSolver state (expert.pyx:66-655): configuration is fixed at
// create (expert.pyx:92-264), geometry is uploaded once by prepare (expert.pyx:309-426) and stays
// resident in HBM, solve (expert.pyx:467-655) streams fk in and fi in/out.
//
// On MI355X the "prepared" state is the device-resident, packed geometry (xi, xk, nk, knowns,
// weighting, order buckets): refitting a case from its 8*nk*dim bytes of coordinates moves fewer
// HBM bytes than reading back a stored factorisation (no x nk operator or c/w/LU arrays, what the
// reference's Case holds, infra.pxd:124-182) and the fit kernels are HBM-bound, so solve() re-runs
// the fused assemble+factor+solve kernel on the resident geometry.
#include <algorithm>
#include <memory>
#include <new>
#include <vector>

#include "wlsqm_internal.hpp"
#include "wlsqm_interp.hpp"
#include "hostio.hpp"

// Geometry + per-case metadata: what prepare() makes resident.  Shared (not copied) between a host solver and its
// guests (expert.pyx:112-126 "guest mode": several fields on the exact same geometry); reference-counted, so unlike
// the reference a guest stays valid if its host is destroyed first.
struct wlsqm_expert_geometry {
    int device = 0, dimension = 0;
    int64_t ncases = 0, max_nk = 1;
    int64_t slots = 2;                 // neighbour slots per device row (>= max_nk; preferred_slots in fit_tile.hip)
    int ragged = 1;                    // 2: the neighbour counts differ by a chunk or more (KParams::ragged)
    int rows_sorted = 1;               // 0: prepare() found the neighbours out of distance order (KParams::rows_sorted)
    int max_no = 0;
    bool ready = false, uniform_order = true;
    std::vector<int32_t> nk, order, wm, no;
    std::vector<int64_t> kn;
    std::vector<long long> idx; int64_t off[6] = {0, 0, 0, 0, 0, 0};
    wlsqm::DevBuf d_nk, d_wm, d_kn, d_order, d_idx, d_xk, d_xi;
    // solution operator of every case (solve_op.hip): built on the first stacked solve of a supported shape, shared with guests
    wlsqm::DevBuf d_op, d_T;
    int op_state = 0;                  // 0 not built, 1 ready, -1 shape not supported
    int op_any_known = 0;
    int64_t bytes() const { return (int64_t)(d_nk.n + d_wm.n + d_kn.n + d_order.n + d_idx.n + d_xk.n + d_xi.n + d_op.n + d_T.n); }
};

struct wlsqm_expert {
    std::shared_ptr<wlsqm_expert_geometry> g;
    bool guest = false;
    int algorithm = 1, do_sens = 0, max_iter = 10;
    bool solved = false;
    // the coefficients interpolate() evaluates = the result of the LATEST solve of any kind (case.fi in the reference): the
    // handle's own d_fi after solve(), the caller's device array after solve_device() / solve_many_device() (last field)
    const double* fi_view = nullptr; int64_t fi_view_stride = 0;
    wlsqm::DevBuf d_fk, d_fi, d_sens, d_it;
    wlsqm::GrowBuf d_fkm, d_fim;        // solve_many: a chunk of stacked right-hand sides / solutions
    wlsqm::Stager st;
    int64_t own_bytes() const { return (int64_t)(d_fk.n + d_fi.n + d_sens.n + d_it.n + d_fkm.b.n + d_fim.b.n); }
    int64_t bytes() const { return own_bytes() + (guest ? 0 : g->bytes()); }
    int alloc_fields() {
        int rc;
        if ((rc = d_it.alloc(4)) || (rc = d_fk.alloc((size_t)g->ncases * g->slots * 8)) ||
            (rc = d_fi.alloc((size_t)g->ncases * g->max_no * 8))) return rc;
        if (do_sens && (rc = d_sens.alloc((size_t)g->ncases * g->slots * g->max_no * 8))) return rc;
        return WLSQM_OK;
    }
};

using namespace wlsqm;

namespace wlsqm {
int nearest_search(int dimension, int64_t ndata, const double* S, int64_t nquery, const double* X, int64_t x_stride,
                   long long* out, hipStream_t s);
long long preferred_slots(int dimension, int order, long long max_nk);
int launch_solve_many(int dimension, int order, const KParams& p, long long K, long long nrhs,
                      const double* fk, long long sfk_r, long long sfk_j, double* fi, long long sfi_r, long long sfi_j,
                      hipStream_t stream, bool* handled);
int solve_op_build(int dimension, int order, const KParams& geom, long long K, const long long* h_knowns, long long ncases,
                   DevBuf& d_op, DevBuf& d_T, int* any_known, hipStream_t s, bool* ok);
int launch_solve_op(int dimension, int order, const KParams& geom, long long K, const double* op, const double* T, int any_known,
                    long long nrhs, const double* fk, long long sfk_r, long long sfk_j, double* fi, long long sfi_r, long long sfi_j,
                    hipStream_t stream, bool* handled);
long long cond_workspace_doubles(int no);
int launch_conds(int dimension, int order, const KParams& p, const int* order_arr, double* ws, long long CH,
                 long long case0, double* out, hipStream_t stream);
}

static KParams expert_params(const wlsqm_expert* h, const double* d_fk, int64_t sfk_j, double* d_fi, int64_t sfi_j) {
    KParams p{};
    const int dim = h->g->dimension;
    p.xk = h->g->d_xk.as<double>(); p.sxk_j = h->g->slots * dim; p.sxk_k = dim;
    p.fk = d_fk; p.sfk_j = sfk_j; p.sfk_k = 1;
    p.nk = h->g->d_nk.as<int>(); p.snk = 1; p.max_nk = h->g->slots;
    p.ragged = h->g->ragged;                                        // (the constructor has seen nk: fit_stage.hip's RAGGED copy for ragged geometries)
    p.rows_sorted = h->g->rows_sorted;                              // (prepare() has seen the rows, or was told: the form for unsorted ones)
    p.xi = h->g->d_xi.as<double>(); p.sxi_j = dim;
    p.fi = d_fi; p.sfi_j = sfi_j;
    p.sens = nullptr; p.ss_j = 0; p.ss_k = 0;
    p.knowns = h->g->d_kn.as<long long>(); p.sknowns = 1;
    p.wm = h->g->d_wm.as<int>(); p.swm = 1;
    p.case_index = nullptr; p.ncases = h->g->ncases;
    p.do_sens = 0; p.iterative = (h->algorithm == WLSQM_ALGO_ITERATIVE) ? 1 : 0; p.max_iter = h->max_iter;
    p.iters_out = h->d_it.as<int>();
    return p;
}

static int expert_launch(const wlsqm_expert* h, KParams p, hipStream_t s) {
    if (h->g->uniform_order) return launch_fit(h->g->dimension, h->g->order[0], p, h->g->slots, s);
    for (int o = 0; o <= 4; ++o) {
        if (h->g->off[o + 1] == h->g->off[o]) continue;
        p.case_index = h->g->d_idx.as<long long>() + h->g->off[o];
        p.ncases = h->g->off[o + 1] - h->g->off[o];
        int rc = launch_fit(h->g->dimension, o, p, h->g->slots, s);
        if (rc != WLSQM_OK) return rc;
    }
    return WLSQM_OK;
}

extern "C" {

int wlsqm_hip_expert_create(wlsqm_expert** out, int device, int dimension, int64_t ncases,
                            const int32_t* nk, const int32_t* order, const int64_t* knowns,
                            const int32_t* weighting_method, int algorithm, int do_sens, int max_iter) {
    if (!out) { set_error("null out"); return WLSQM_EVALUE; }
    *out = nullptr;
    if (dimension < 1 || dimension > 3) { set_error("Dimension must be 1, 2 or 3"); return WLSQM_EVALUE; }   // expert.pyx:134-135
    if (algorithm != WLSQM_ALGO_BASIC && algorithm != WLSQM_ALGO_ITERATIVE) { set_error("Unknown algorithm specifier"); return WLSQM_EVALUE; }   // :151-156
    if (ncases < 1) { set_error("max_cases must be >= 1"); return WLSQM_EVALUE; }                              // infra.pyx:311-313
    if (!nk || !order || !knowns || !weighting_method) { set_error("null array"); return WLSQM_EVALUE; }
    wlsqm_expert* h = new (std::nothrow) wlsqm_expert();
    if (!h) { set_error("out of memory"); return WLSQM_EMEMORY; }
    h->g = std::make_shared<wlsqm_expert_geometry>();
    wlsqm_expert_geometry& g = *h->g;
    g.device = device; g.dimension = dimension; g.ncases = ncases;
    h->algorithm = algorithm; h->do_sens = do_sens ? 1 : 0; h->max_iter = max_iter;
    g.nk.assign(nk, nk + ncases); g.order.assign(order, order + ncases);
    g.wm.assign(weighting_method, weighting_method + ncases); g.kn.assign(knowns, knowns + ncases);
    g.no.resize(ncases);
    int64_t mk = 0;
    for (int64_t j = 0; j < ncases; ++j) {
        const int no = wlsqm_hip_number_of_dofs(dimension, order[j]);
        if (no < 0 || nk[j] < 0) { delete h; set_error("order must be 0..4 and nk >= 0"); return WLSQM_EVALUE; }
        g.no[j] = no; g.max_no = std::max(g.max_no, no); mk = std::max<int64_t>(mk, nk[j]);
    }
    g.max_nk = std::max<int64_t>(mk, 1);
    { int64_t lo = mk; for (int64_t j = 0; j < ncases; ++j) lo = std::min<int64_t>(lo, nk[j]); g.ragged = (mk - lo >= 8) ? 2 : 1; }
    g.uniform_order = std::all_of(g.order.begin(), g.order.end(), [&](int o) { return o == g.order[0]; });
    g.slots = preferred_slots(dimension, g.uniform_order ? g.order[0] : -1, g.max_nk);
    DeviceScope scope; int rc = scope.enter(device);
    if (rc != WLSQM_OK) { delete h; return rc; }
    if ((rc = g.d_nk.alloc(ncases * 4)) || (rc = g.d_wm.alloc(ncases * 4)) || (rc = g.d_kn.alloc(ncases * 8)) ||
        (rc = g.d_order.alloc(ncases * 4)) || (rc = g.d_xk.alloc((size_t)ncases * g.slots * dimension * 8)) ||
        (rc = g.d_xi.alloc((size_t)ncases * dimension * 8)) || (rc = h->alloc_fields())) { delete h; return rc; }
    hipError_t e;
    if ((e = hipMemcpy(g.d_nk.p, g.nk.data(), g.d_nk.n, hipMemcpyHostToDevice)) != hipSuccess ||
        (e = hipMemcpy(g.d_wm.p, g.wm.data(), g.d_wm.n, hipMemcpyHostToDevice)) != hipSuccess ||
        (e = hipMemcpy(g.d_kn.p, g.kn.data(), g.d_kn.n, hipMemcpyHostToDevice)) != hipSuccess ||
        (e = hipMemcpy(g.d_order.p, g.order.data(), g.d_order.n, hipMemcpyHostToDevice)) != hipSuccess) {
        delete h; return hip_fail(e, "hipMemcpy(expert metadata)");
    }
    if (!g.uniform_order) {
        for (int o = 0; o <= 4; ++o) {
            g.off[o] = (int64_t)g.idx.size();
            for (int64_t j = 0; j < ncases; ++j) if (g.order[j] == o) g.idx.push_back(j);
        }
        g.off[5] = (int64_t)g.idx.size();
        if ((rc = g.d_idx.alloc(g.idx.size() * 8))) { delete h; return rc; }
        if ((e = hipMemcpy(g.d_idx.p, g.idx.data(), g.d_idx.n, hipMemcpyHostToDevice)) != hipSuccess) {
            delete h; return hip_fail(e, "hipMemcpy(expert idx)");
        }
    }
    *out = h;
    return WLSQM_OK;
}

int wlsqm_hip_expert_create_guest(wlsqm_expert** out, wlsqm_expert* host, int algorithm, int do_sens, int max_iter) {
    if (!out) { set_error("null out"); return WLSQM_EVALUE; }
    *out = nullptr;
    if (!host) { set_error("null host"); return WLSQM_EVALUE; }
    if (algorithm != WLSQM_ALGO_BASIC && algorithm != WLSQM_ALGO_ITERATIVE) { set_error("Unknown algorithm specifier"); return WLSQM_EVALUE; }
    if (!host->g->ready) {                                                                                       // expert.pyx:165-166
        set_error("In guest mode, host must be in the ready state (host.prepare() must have been called first)");
        return WLSQM_ERUNTIME;
    }
    DeviceScope scope; int rc = scope.enter(host->g->device);
    if (rc != WLSQM_OK) return rc;
    wlsqm_expert* h = new (std::nothrow) wlsqm_expert();
    if (!h) { set_error("out of memory"); return WLSQM_EMEMORY; }
    h->g = host->g; h->guest = true;
    h->algorithm = algorithm; h->do_sens = do_sens ? 1 : 0; h->max_iter = max_iter;
    if ((rc = h->alloc_fields())) { delete h; return rc; }
    *out = h;
    return WLSQM_OK;
}

int wlsqm_hip_expert_prepare(wlsqm_expert* h, const double* xi, int64_t xi_stride_case,
                             const double* xk, int64_t xk_stride_case, int64_t xk_stride_k, int64_t max_nk) {
    if (!h) { set_error("null argument"); return WLSQM_EVALUE; }
    if (h->guest) {       // the geometry is the host's (expert.pyx:350-352): nothing to upload
        if (!h->g->ready) { set_error("In guest mode, host must be in the ready state"); return WLSQM_ERUNTIME; }
        return WLSQM_OK;
    }
    if (!xi || !xk) { set_error("null argument"); return WLSQM_EVALUE; }
    h->g->ready = false;
    h->g->op_state = 0;                // a new geometry: the stored solution operator is stale
    if (max_nk < h->g->max_nk && h->g->max_nk > 1) { set_error("xk has fewer neighbour slots than max(nk)"); return WLSQM_EVALUE; }
    DeviceScope scope; int rc = scope.enter(h->g->device);
    if (rc != WLSQM_OK) return rc;
    const int dim = h->g->dimension; const int64_t n = h->g->ncases, K = h->g->slots, mk = h->g->max_nk;
    h->g->rows_sorted = sampled_rows_sorted(n, dim, xk, xk_stride_case, xk_stride_k, xi, xi_stride_case,
                                            [&](int64_t r) { return h->g->nk[r]; }, [](int64_t r) { return r; });
    if ((rc = h->st.ensure(h->g->device))) return rc;
    hipStream_t s = nullptr;
    if ((rc = h->st.upload_rows(h->g->d_xk.p, xk, n, mk * dim, xk_stride_case, xk_stride_k, dim, 8, s, K * dim))) return rc;
    if ((rc = h->st.upload_rows(h->g->d_xi.p, xi, n, dim, xi_stride_case, dim, dim, 8, s))) return rc;
    if ((rc = h->st.drain())) return rc;
    WLSQM_HIP_CHECK(hipStreamSynchronize(s));
    h->g->ready = true;
    return WLSQM_OK;
}

// prepare() from DEVICE-resident arrays (extension): xi[ncases, xi_stride_case], xk[ncases, >= max_nk, dimension] with the
// coordinate axis contiguous and xk_stride_k == dimension; copied (device to device, on `stream`) into the solver's own
// padded geometry block, so the caller's tensors may be freed afterwards.
int wlsqm_hip_expert_prepare_device(wlsqm_expert* h, void* stream, const double* xi, int64_t xi_stride_case,
                                    const double* xk, int64_t xk_stride_case, int64_t xk_stride_k) {
    if (!h) { set_error("null argument"); return WLSQM_EVALUE; }
    if (h->guest) {
        if (!h->g->ready) { set_error("In guest mode, host must be in the ready state"); return WLSQM_ERUNTIME; }
        return WLSQM_OK;
    }
    if (!xi || !xk) { set_error("null argument"); return WLSQM_EVALUE; }
    wlsqm_expert_geometry& g = *h->g;
    const int dim = g.dimension;
    if (xk_stride_k != dim || xk_stride_case < g.max_nk * dim || xi_stride_case < dim) {
        set_error("prepare_device needs xk[ncases, >= max_nk, dimension] with contiguous neighbour rows"); return WLSQM_EVALUE;
    }
    g.ready = false;
    g.op_state = 0;                    // a new geometry: the stored solution operator is stale
    g.rows_sorted = order_hint_value();                                // (device-resident rows: the caller's word, wlsqm_hip_set_order_hint)
    DeviceScope scope; int rc = scope.enter(g.device);
    if (rc != WLSQM_OK) return rc;
    hipStream_t s = (hipStream_t)stream;
    WLSQM_HIP_CHECK(hipMemcpy2DAsync(g.d_xk.p, (size_t)g.slots * dim * 8, xk, (size_t)xk_stride_case * 8, (size_t)g.max_nk * dim * 8,
                                     (size_t)g.ncases, hipMemcpyDeviceToDevice, s));
    WLSQM_HIP_CHECK(hipMemcpy2DAsync(g.d_xi.p, (size_t)dim * 8, xi, (size_t)xi_stride_case * 8, (size_t)dim * 8, (size_t)g.ncases,
                                     hipMemcpyDeviceToDevice, s));
    WLSQM_HIP_CHECK(hipStreamSynchronize(s));
    g.ready = true;
    return WLSQM_OK;
}

int wlsqm_hip_expert_solve(wlsqm_expert* h, const double* fk, int64_t fk_stride_case, int64_t fk_stride_k,
                           double* fi, int64_t fi_stride_case,
                           double* sens, int64_t sens_stride_case, int64_t sens_stride_k, int32_t* iterations_out) {
    if (!h || !fk || !fi) { set_error("null argument"); return WLSQM_EVALUE; }
    if (!h->g->ready) { set_error("Solver is not in the ready state; prepare() must be called before solve()"); return WLSQM_ERUNTIME; }   // expert.pyx:493-494
    DeviceScope scope; int rc = scope.enter(h->g->device);
    if (rc != WLSQM_OK) return rc;
    const int64_t n = h->g->ncases, K = h->g->slots; const int NO = h->g->max_no;
    if ((rc = h->st.ensure(h->g->device))) return rc;
    hipStream_t s = nullptr;
    const bool want_sens = h->do_sens && sens;
    if ((rc = h->st.upload_rows(h->d_fk.p, fk, n, h->g->max_nk, fk_stride_case, fk_stride_k, 1, 8, s, K))) return rc;
    if ((rc = h->st.upload_rows(h->d_fi.p, fi, n, NO, fi_stride_case, NO, NO, 8, s))) return rc;
    WLSQM_HIP_CHECK(hipMemsetAsync(h->d_it.p, 0, 4, s));
    KParams p = expert_params(h, h->d_fk.as<double>(), K, h->d_fi.as<double>(), NO);
    if (want_sens) {
        WLSQM_HIP_CHECK(hipMemsetAsync(h->d_sens.p, 0, h->d_sens.n, s));
        p.sens = h->d_sens.as<double>(); p.ss_j = K * NO; p.ss_k = NO; p.do_sens = 1;
    }
    rc = expert_launch(h, p, s);
    if (rc != WLSQM_OK) return rc;
    rc = h->st.download_rows(h->d_fi.p, n, NO, 8, s, [&](int64_t j, const char* row) {
        if (wlsqm_hip_number_of_reduced_dofs(h->g->no[j], h->g->kn[j]) < 1) return;
        std::memcpy(fi + j * fi_stride_case, row, (size_t)h->g->no[j] * 8);
    });
    if (rc != WLSQM_OK) return rc;
    if (want_sens) {
        rc = h->st.download_rows(h->d_sens.p, n, K * NO, 8, s, [&](int64_t j, const char* row) {
            if (wlsqm_hip_number_of_reduced_dofs(h->g->no[j], h->g->kn[j]) < 1) return;
            unsigned long long known, dropped;
            effective_mask_host(h->g->no[j], h->g->kn[j], known, dropped);
            const double* r = reinterpret_cast<const double*>(row);
            for (int64_t k = 0; k < h->g->nk[j]; ++k)
                for (int a = 0; a < h->g->no[j]; ++a) {
                    if ((dropped >> a) & 1ull) continue;
                    sens[j * sens_stride_case + k * sens_stride_k + a] = r[k * NO + a];
                }
        });
        if (rc != WLSQM_OK) return rc;
    }
    int h_it = 0;
    WLSQM_HIP_CHECK(hipMemcpyAsync(&h_it, h->d_it.p, 4, hipMemcpyDeviceToHost, s));
    WLSQM_HIP_CHECK(hipStreamSynchronize(s));
    if (iterations_out) *iterations_out = (h->algorithm == WLSQM_ALGO_ITERATIVE) ? (h_it < 1 ? 1 : h_it) : 0;
    h->solved = true;                  // d_fi now holds the coefficients interpolate() evaluates (case.fi in the reference)
    h->fi_view = h->d_fi.as<double>(); h->fi_view_stride = NO;
    return WLSQM_OK;
}

int wlsqm_hip_expert_solve_device(wlsqm_expert* h, void* stream, const double* fk, int64_t fk_stride_case,
                                  double* fi, int64_t fi_stride_case) {
    if (!h || !fk || !fi) { set_error("null argument"); return WLSQM_EVALUE; }
    if (!h->g->ready) { set_error("Solver is not in the ready state; prepare() must be called before solve()"); return WLSQM_ERUNTIME; }
    DeviceScope scope; int rc = scope.enter(h->g->device);
    if (rc != WLSQM_OK) return rc;
    hipStream_t s = (hipStream_t)stream;
    KParams p = expert_params(h, fk, fk_stride_case, fi, fi_stride_case);
    p.iters_out = nullptr;
    rc = expert_launch(h, p, s);
    if (rc == WLSQM_OK) { h->solved = true; h->fi_view = fi; h->fi_view_stride = fi_stride_case; }
    return rc;
}

// Many fields on the prepared geometry, device-resident (extension; BASELINE config 4).  Fast path: one launch that
// shares the geometry work between the fields (solve_many.hip); otherwise nrhs launches of the fused kernel.
// Which stacked-solve kernel: WLSQM_HIP_SOLVE_MANY = "fma" (solve_many.hip: geometry work shared inside the launch, FMA loop,
// no <= 6 and K <= 32 only), "op" (solve_op.hip: stored solution operator, batched GEMM on the matrix cores) or unset = the
// default rule below.
static int solve_many_choice() {
    const char* e = getenv("WLSQM_HIP_SOLVE_MANY");
    if (e && e[0] == 'f') return 1;
    if (e && e[0] == 'o') return 2;
    return 0;
}

static int ensure_operator(wlsqm_expert* h, hipStream_t s) {
    wlsqm_expert_geometry& g = *h->g;
    if (g.op_state != 0) return WLSQM_OK;
    bool ok = false;
    KParams p = expert_params(h, nullptr, 0, nullptr, 0);
    // the operator is 8 * no * K bytes per case (plus a temporary sensitivity block of <= 128 MB): only when it comfortably fits
    // what is free now; otherwise (and if an allocation fails all the same) the stacked solve takes the other kernels
    {
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { (void)hipGetLastError(); free_b = 0; }
        const double need = (double)g.ncases * (double)g.max_no * (double)((g.slots + 7) / 8 * 8) * 8.0 + 160.0 * 1048576.0;
        if (need > 0.6 * (double)free_b) { g.op_state = -1; return WLSQM_OK; }
    }
    int rc = solve_op_build(g.dimension, g.order[0], p, g.slots, reinterpret_cast<const long long*>(g.kn.data()), g.ncases,
                            g.d_op, g.d_T, &g.op_any_known, s, &ok);
    if (rc == WLSQM_EMEMORY) { g.d_op.alloc(0); g.d_T.alloc(0); g.op_state = -1; set_error(""); return WLSQM_OK; }
    if (rc != WLSQM_OK) { g.d_op.alloc(0); g.d_T.alloc(0); return rc; }
    g.op_state = ok ? 1 : -1;
    if (!ok) { g.d_op.alloc(0); g.d_T.alloc(0); }
    return WLSQM_OK;
}

// Many fields on the prepared geometry, device-resident (extension; BASELINE config 4).  Fast paths: the stored solution
// operator applied as one batched GEMM (solve_op.hip; built on the first call: that call synchronises the stream), or one launch
// that shares the geometry work between the fields (solve_many.hip); otherwise nrhs launches of the fused kernel.
static int solve_many_on_device(wlsqm_expert* h, hipStream_t s, int64_t nrhs, const double* fk, int64_t sfk_r, int64_t sfk_j,
                                double* fi, int64_t sfi_r, int64_t sfi_j) {
    wlsqm_expert_geometry& g = *h->g;
    bool handled = false;
    // reference-order numerics (fit_strict.hip): one strict fit per field, as the reference's solve() per field
    if (g.uniform_order && h->algorithm == WLSQM_ALGO_BASIC && !strict_mode()) {
        KParams p = expert_params(h, nullptr, 0, nullptr, 0);
        const int choice = solve_many_choice();
        const int no = wlsqm_hip_number_of_dofs(g.dimension, g.order[0]);
        // default: the FMA kernel where it exists and the stack is short (it needs no stored operator: 8 * 16 * K bytes per case);
        // the operator from 64 fields on, and wherever the FMA kernel has no instantiation (no > 6 or K > 32)
        const bool fma_shape = no <= 6 && g.slots <= 32;
        const bool want_op = choice == 2 || (choice == 0 && (!fma_shape || nrhs >= 64));
        if (want_op && nrhs >= 2) {
            int rc = ensure_operator(h, s);
            if (rc != WLSQM_OK) return rc;
            if (g.op_state == 1) {
                rc = launch_solve_op(g.dimension, g.order[0], p, g.slots, g.d_op.as<double>(), g.d_T.as<double>(), g.op_any_known, nrhs,
                                     fk, sfk_r, sfk_j, fi, sfi_r, sfi_j, s, &handled);
                if (rc != WLSQM_OK) return rc;
            }
        }
        if (!handled && choice != 2) {
            int rc = launch_solve_many(g.dimension, g.order[0], p, g.slots, nrhs, fk, sfk_r, sfk_j, fi, sfi_r, sfi_j, s, &handled);
            if (rc != WLSQM_OK) return rc;
        }
    }
    if (handled) return WLSQM_OK;
    for (int64_t r = 0; r < nrhs; ++r) {
        KParams p = expert_params(h, fk + r * sfk_r, sfk_j, fi + r * sfi_r, sfi_j);
        p.iters_out = nullptr;
        int rc = expert_launch(h, p, s);
        if (rc != WLSQM_OK) return rc;
    }
    return WLSQM_OK;
}

int wlsqm_hip_expert_prepare_operator(wlsqm_expert* h, void* stream, int* built) {
    if (built) *built = 0;
    if (!h) { set_error("null argument"); return WLSQM_EVALUE; }
    if (!h->g->ready) { set_error("Solver is not in the ready state; prepare() must be called before solve()"); return WLSQM_ERUNTIME; }
    if (!h->g->uniform_order || h->algorithm != WLSQM_ALGO_BASIC) return WLSQM_OK;
    DeviceScope scope; int rc = scope.enter(h->g->device);
    if (rc != WLSQM_OK) return rc;
    rc = ensure_operator(h, (hipStream_t)stream);
    if (rc == WLSQM_OK && built) *built = h->g->op_state == 1 ? 1 : 0;
    return rc;
}

int wlsqm_hip_expert_solve_many_device(wlsqm_expert* h, void* stream, int64_t nrhs,
                                       const double* fk, int64_t fk_stride_rhs, int64_t fk_stride_case,
                                       double* fi, int64_t fi_stride_rhs, int64_t fi_stride_case) {
    if (!h || !fk || !fi) { set_error("null argument"); return WLSQM_EVALUE; }
    if (nrhs < 1) { set_error("nrhs must be >= 1"); return WLSQM_EVALUE; }
    if (!h->g->ready) { set_error("Solver is not in the ready state; prepare() must be called before solve()"); return WLSQM_ERUNTIME; }
    if (h->do_sens) { set_error("solve_many does not compute sensitivities"); return WLSQM_EVALUE; }
    DeviceScope scope; int rc = scope.enter(h->g->device);
    if (rc != WLSQM_OK) return rc;
    rc = solve_many_on_device(h, (hipStream_t)stream, nrhs, fk, fk_stride_rhs, fk_stride_case, fi, fi_stride_rhs, fi_stride_case);
    if (rc == WLSQM_OK) { h->solved = true; h->fi_view = fi + (nrhs - 1) * fi_stride_rhs; h->fi_view_stride = fi_stride_case; }
    return rc;
}

int wlsqm_hip_expert_solve_many(wlsqm_expert* h, int64_t nrhs,
                                const double* fk, int64_t fk_stride_rhs, int64_t fk_stride_case, int64_t fk_stride_k,
                                double* fi, int64_t fi_stride_rhs, int64_t fi_stride_case) {
    if (!h || !fk || !fi) { set_error("null argument"); return WLSQM_EVALUE; }
    if (nrhs < 1) { set_error("nrhs must be >= 1"); return WLSQM_EVALUE; }
    if (!h->g->ready) { set_error("Solver is not in the ready state; prepare() must be called before solve()"); return WLSQM_ERUNTIME; }
    if (h->do_sens) { set_error("solve_many does not compute sensitivities"); return WLSQM_EVALUE; }
    const wlsqm_expert_geometry& g = *h->g;
    DeviceScope scope; int rc = scope.enter(g.device);
    if (rc != WLSQM_OK) return rc;
    const int64_t n = g.ncases, K = g.slots; const int NO = g.max_no;
    if ((rc = h->st.ensure(g.device))) return rc;
    hipStream_t s = nullptr;
    // right-hand sides travel in chunks of at most ~1 GiB of fk
    int64_t chunk = std::max<int64_t>(1, (int64_t(1) << 30) / std::max<int64_t>(1, n * K * 8));
    chunk = std::min(chunk, nrhs);
    if ((rc = h->d_fkm.need((size_t)chunk * n * K * 8)) || (rc = h->d_fim.need((size_t)chunk * n * NO * 8))) return rc;
    double* d_fk = h->d_fkm.as<double>(); double* d_fi = h->d_fim.as<double>();
    for (int64_t r0 = 0; r0 < nrhs; r0 += chunk) {
        const int64_t nr = std::min(chunk, nrhs - r0);
        for (int64_t r = 0; r < nr; ++r) {
            if ((rc = h->st.upload_rows(d_fk + r * n * K, fk + (r0 + r) * fk_stride_rhs, n, g.max_nk, fk_stride_case, fk_stride_k, 1, 8, s, K))) return rc;
            if ((rc = h->st.upload_rows(d_fi + r * n * NO, fi + (r0 + r) * fi_stride_rhs, n, NO, fi_stride_case, NO, NO, 8, s))) return rc;
        }
        if ((rc = solve_many_on_device(h, s, nr, d_fk, n * K, K, d_fi, n * NO, NO))) return rc;
        for (int64_t r = 0; r < nr; ++r) {
            double* out = fi + (r0 + r) * fi_stride_rhs;
            rc = h->st.download_rows(d_fi + r * n * NO, n, NO, 8, s, [&](int64_t j, const char* row) {
                if (wlsqm_hip_number_of_reduced_dofs(g.no[j], g.kn[j]) < 1) return;
                std::memcpy(out + j * fi_stride_case, row, (size_t)g.no[j] * 8);
            });
            if (rc != WLSQM_OK) return rc;
        }
    }
    // the last field's coefficients are what interpolate() evaluates from now on
    {
        const int64_t last = (nrhs - 1) % chunk;
        WLSQM_HIP_CHECK(hipMemcpyAsync(h->d_fi.p, d_fi + last * n * NO, (size_t)n * NO * 8, hipMemcpyDeviceToDevice, s));
        h->solved = true; h->fi_view = h->d_fi.as<double>(); h->fi_view_stride = NO;
    }
    WLSQM_HIP_CHECK(hipStreamSynchronize(s));
    return WLSQM_OK;
}

// ExpertSolver.interpolate (expert.pyx:687-781): evaluate the models of the last solve() at nx host points.
// I (host, nullable) names the model per point (mode='nearest', expert.pyx:830-895); list_off/list_idx (host CSR,
// nullable) give the models within radius r of each point (mode='continuous', expert.pyx:898-985).
int wlsqm_hip_expert_interpolate(wlsqm_expert* h, const double* x, int64_t x_stride, int64_t nx, const int64_t* I,
                                 const int64_t* list_off, const int64_t* list_idx, double r, int diff, double* out) {
    if (!h || !x || !out) { set_error("null argument"); return WLSQM_EVALUE; }
    if (!h->g->ready || !h->solved) { set_error("interpolate() needs prepare() and solve() first"); return WLSQM_ERUNTIME; }
    if (!I && !list_off) { set_error("either I or the neighbour lists must be given"); return WLSQM_EVALUE; }
    DeviceScope scope; int rc = scope.enter(h->g->device);
    if (rc != WLSQM_OK) return rc;
    if (nx <= 0) return WLSQM_OK;
    // a solve_device() result may still be in flight on the caller's (possibly non-blocking) stream
    if (h->fi_view != h->d_fi.as<double>()) WLSQM_HIP_CHECK(hipDeviceSynchronize());
    const int dim = h->g->dimension;
    std::vector<double> sx((size_t)nx * dim);
    for (int64_t m = 0; m < nx; ++m)
        for (int c = 0; c < dim; ++c) sx[(size_t)m * dim + c] = x[m * x_stride + c];
    DevBuf d_x, d_I, d_off, d_idx, d_out;
    hipStream_t s = nullptr;
    if ((rc = d_x.alloc(sx.size() * 8)) || (rc = d_out.alloc((size_t)nx * 8))) return rc;
    WLSQM_HIP_CHECK(hipMemcpyAsync(d_x.p, sx.data(), d_x.n, hipMemcpyHostToDevice, s));
    InterpParams q{};
    q.xi = h->g->d_xi.as<double>(); q.sxi = dim; q.fi = h->fi_view; q.sfi = h->fi_view_stride;
    q.order = h->g->d_order.as<int>(); q.sorder = 1; q.nmodels = h->g->ncases;
    q.x = d_x.as<double>(); q.sx = dim; q.nx = nx; q.diff = diff; q.out = d_out.as<double>();
    if (list_off) {
        const int64_t nlist = list_off[nx];
        if ((rc = d_off.alloc((size_t)(nx + 1) * 8)) || (rc = d_idx.alloc((size_t)std::max<int64_t>(nlist, 1) * 8))) return rc;
        WLSQM_HIP_CHECK(hipMemcpyAsync(d_off.p, list_off, (size_t)(nx + 1) * 8, hipMemcpyHostToDevice, s));
        if (nlist > 0) WLSQM_HIP_CHECK(hipMemcpyAsync(d_idx.p, list_idx, (size_t)nlist * 8, hipMemcpyHostToDevice, s));
        q.list_off = d_off.as<long long>(); q.list_idx = d_idx.as<long long>(); q.r2 = r * r;
    } else {
        if ((rc = d_I.alloc((size_t)nx * 8))) return rc;
        WLSQM_HIP_CHECK(hipMemcpyAsync(d_I.p, I, (size_t)nx * 8, hipMemcpyHostToDevice, s));
        q.I = d_I.as<long long>();
    }
    rc = launch_interp(dim, q, s);
    if (rc != WLSQM_OK) return rc;
    WLSQM_HIP_CHECK(hipMemcpyAsync(out, d_out.p, (size_t)nx * 8, hipMemcpyDeviceToHost, s));
    WLSQM_HIP_CHECK(hipStreamSynchronize(s));
    return WLSQM_OK;
}

// mode='nearest' without host-side search (expert.pyx:830-895 queries a cKDTree of the origins): the nearest origin of
// every x is found on the device (knn.hip), then the model is evaluated as above.  I_out (host, nullable) receives the
// model numbers, the second return value of the reference's interpolate().
int wlsqm_hip_expert_interpolate_nearest(wlsqm_expert* h, const double* x, int64_t x_stride, int64_t nx, int diff,
                                         double* out, int64_t* I_out) {
    if (!h || !x || !out) { set_error("null argument"); return WLSQM_EVALUE; }
    if (!h->g->ready || !h->solved) { set_error("interpolate() needs prepare() and solve() first"); return WLSQM_ERUNTIME; }
    DeviceScope scope; int rc = scope.enter(h->g->device);
    if (rc != WLSQM_OK) return rc;
    if (nx <= 0) return WLSQM_OK;
    // a solve_device() result may still be in flight on the caller's (possibly non-blocking) stream
    if (h->fi_view != h->d_fi.as<double>()) WLSQM_HIP_CHECK(hipDeviceSynchronize());
    const int dim = h->g->dimension;
    std::vector<double> sx((size_t)nx * dim);
    for (int64_t m = 0; m < nx; ++m)
        for (int c = 0; c < dim; ++c) sx[(size_t)m * dim + c] = x[m * x_stride + c];
    DevBuf d_x, d_I, d_out;
    hipStream_t s = nullptr;
    if ((rc = d_x.alloc(sx.size() * 8)) || (rc = d_out.alloc((size_t)nx * 8)) || (rc = d_I.alloc((size_t)nx * 8))) return rc;
    WLSQM_HIP_CHECK(hipMemcpyAsync(d_x.p, sx.data(), d_x.n, hipMemcpyHostToDevice, s));
    if ((rc = nearest_search(dim, h->g->ncases, h->g->d_xi.as<double>(), nx, d_x.as<double>(), dim, d_I.as<long long>(), s))) return rc;
    InterpParams q{};
    q.xi = h->g->d_xi.as<double>(); q.sxi = dim; q.fi = h->fi_view; q.sfi = h->fi_view_stride;
    q.order = h->g->d_order.as<int>(); q.sorder = 1; q.nmodels = h->g->ncases;
    q.x = d_x.as<double>(); q.sx = dim; q.nx = nx; q.diff = diff; q.out = d_out.as<double>();
    q.I = d_I.as<long long>();
    rc = launch_interp(dim, q, s);
    if (rc != WLSQM_OK) return rc;
    WLSQM_HIP_CHECK(hipMemcpyAsync(out, d_out.p, (size_t)nx * 8, hipMemcpyDeviceToHost, s));
    if (I_out) WLSQM_HIP_CHECK(hipMemcpyAsync(I_out, d_I.p, (size_t)nx * 8, hipMemcpyDeviceToHost, s));
    WLSQM_HIP_CHECK(hipStreamSynchronize(s));
    return WLSQM_OK;
}

// mode='continuous' without host-side lists (expert.pyx:898-985 builds them with cKDTree.query_ball_tree): the models within
// r of every x are found on the device by a grid walk (knn.hip) and averaged in the same kernel.
int wlsqm_hip_expert_interpolate_continuous(wlsqm_expert* h, const double* x, int64_t x_stride, int64_t nx, double r, int diff,
                                            double* out) {
    if (!h || !x || !out) { set_error("null argument"); return WLSQM_EVALUE; }
    if (!h->g->ready || !h->solved) { set_error("interpolate() needs prepare() and solve() first"); return WLSQM_ERUNTIME; }
    if (!(r > 0.0)) { set_error("r must be positive"); return WLSQM_EVALUE; }
    DeviceScope scope; int rc = scope.enter(h->g->device);
    if (rc != WLSQM_OK) return rc;
    if (nx <= 0) return WLSQM_OK;
    // a solve_device() result may still be in flight on the caller's (possibly non-blocking) stream
    if (h->fi_view != h->d_fi.as<double>()) WLSQM_HIP_CHECK(hipDeviceSynchronize());
    const int dim = h->g->dimension;
    std::vector<double> sx((size_t)nx * dim);
    for (int64_t m = 0; m < nx; ++m)
        for (int c = 0; c < dim; ++c) sx[(size_t)m * dim + c] = x[m * x_stride + c];
    DevBuf d_x, d_out;
    hipStream_t s = nullptr;
    if ((rc = d_x.alloc(sx.size() * 8)) || (rc = d_out.alloc((size_t)nx * 8))) return rc;
    WLSQM_HIP_CHECK(hipMemcpyAsync(d_x.p, sx.data(), d_x.n, hipMemcpyHostToDevice, s));
    InterpParams q{};
    q.xi = h->g->d_xi.as<double>(); q.sxi = dim; q.fi = h->fi_view; q.sfi = h->fi_view_stride;
    q.order = h->g->d_order.as<int>(); q.sorder = 1; q.nmodels = h->g->ncases;
    q.x = d_x.as<double>(); q.sx = dim; q.nx = nx; q.diff = diff; q.out = d_out.as<double>();
    if ((rc = interp_continuous(dim, q, r, s))) return rc;
    WLSQM_HIP_CHECK(hipMemcpyAsync(out, d_out.p, (size_t)nx * 8, hipMemcpyDeviceToHost, s));
    WLSQM_HIP_CHECK(hipStreamSynchronize(s));
    return WLSQM_OK;
}

int wlsqm_hip_expert_conds(wlsqm_expert* h, double* out) {
    if (!h || !out) { set_error("null argument"); return WLSQM_EVALUE; }
    if (!h->g->ready) { set_error("Solver is not in the ready state; prepare() must be called before conds()"); return WLSQM_ERUNTIME; }   // expert.pyx:438-439
    DeviceScope scope; int rc = scope.enter(h->g->device);
    if (rc != WLSQM_OK) return rc;
    const int64_t n = h->g->ncases;
    const long long CH = std::min<int64_t>(n, 32768);
    DevBuf ws, d_out;
    if ((rc = ws.alloc((size_t)CH * cond_workspace_doubles(h->g->max_no) * 8)) || (rc = d_out.alloc((size_t)n * 8))) return rc;
    KParams p = expert_params(h, nullptr, 0, nullptr, 0);
    hipStream_t s = nullptr;
    bool present[5] = {false, false, false, false, false};
    for (int64_t j = 0; j < n; ++j) present[h->g->order[j]] = true;
    for (long long c0 = 0; c0 < n; c0 += CH)
        for (int o = 0; o <= 4; ++o) {
            if (!present[o]) continue;
            rc = launch_conds(h->g->dimension, o, p, h->g->uniform_order ? nullptr : h->g->d_order.as<int>(), ws.as<double>(), CH, c0,
                              d_out.as<double>(), s);
            if (rc != WLSQM_OK) return rc;
        }
    WLSQM_HIP_CHECK(hipMemcpyAsync(out, d_out.p, (size_t)n * 8, hipMemcpyDeviceToHost, s));
    WLSQM_HIP_CHECK(hipStreamSynchronize(s));
    return WLSQM_OK;
}

int wlsqm_hip_expert_memory_used(const wlsqm_expert* h, int64_t* used, int64_t* total) {
    if (!h) { set_error("null handle"); return WLSQM_EVALUE; }
    if (used) *used = h->bytes();
    if (total) *total = h->bytes();      // expert.pyx:296-297: the buffer is exactly as large as needed
    return WLSQM_OK;
}

int wlsqm_hip_expert_destroy(wlsqm_expert* h) {
    if (!h) return WLSQM_OK;
    DeviceScope scope;
    (void)scope.enter(h->g->device);
    delete h;
    return WLSQM_OK;
}

}  // extern "C"
