// fit_sens.hip — the fit with sensitivities (do_sens, impl.pyx:776-778, 821-846) for the shapes without a tile kernel of their
// own: 15-unknown systems (2D order 4), any K > 128, 3D orders 3 and 4.  Two kernels per slice of the batch:
//
//   1. the basic fit, which also leaves the INVERSE of every case's knowns-eliminated normal matrix in scratch memory
//      (inv[t][no][no], fit_chunk.hip for up to 15 unknowns, fit_wave.hip above): `no` substitutions per case instead of one
//      per neighbour (impl.pyx:831-834 runs dgetrs over all nk right-hand sides);
//   2. sens_apply_kernel, one wave per case, ONE LANE PER NEIGHBOUR: the lane rebuilds its neighbour's weighted monomial row
//      w c[k, :] (impl.pyx:826-829) and the wave multiplies the rows of 16 neighbours with the inverse on the matrix
//      cores (v_mfma_f64_16x16x4, the inverse as the A operand, loaded once per case with ordinary vector loads).  A
//      per-case matrix has to reach all of the case's neighbour lanes; the MFMA does that broadcast in hardware.  Measured
//      alternative: the inverse as SGPR operands of v_fma_f64 (scalar loads, the case is wave-uniform) — every row of
//      the inverse is a scalar-cache miss the wave waits out (15 serialised ~1 us round trips per case, 100 SGPRs hold two
//      rows): 3.7 us per 1000 C3 cases against the lane kernel's 5.7.  The rows of 64 neighbours are one contiguous run of
//      the output; they pass through LDS once and leave as full 512-byte stores (a lane storing its own row of `no` doubles
//      touches 64 lines per instruction; the lane-per-case kernel this replaces wrote 1.0 TB/s, the wave-per-case kernel
//      0.25-0.5 TB/s).
//
// The sensitivities are a linear map of the right-hand side (sens[k, :] = A^-1 (w c[k, :]), known DOFs masked), so the explicit
// inverse gives the same numbers as the substitutions up to rounding of the order cond(A) eps — the bound every other path has.
// The batch is cut into slices so that the scratch for the inverses stays bounded (200k cases, slices of 32 / 96 / 256 / 1024 MB:
// C3-like 1.27 / 1.07 / 1.05 / 1.04 ms, 3D order 4 at K = 100 22.3 / 20.3 / 19.2 / 18.7 ms — no Infinity Cache effect to be had).
#include "wlsqm_internal.hpp"
#include "wlsqm_kernels.hpp"

namespace wlsqm {

bool chunk_inverse_ok(int dimension, int order, const KParams& p, long long K);
int launch_fit_chunk_inverse(int dimension, int order, const KParams& p, long long K, hipStream_t stream);
int launch_fit_wave_inverse(int dimension, int order, const KParams& p, double* inv, hipStream_t stream);

typedef double sd4_ __attribute__((ext_vector_type(4)));

// LDS traffic between the lanes of ONE wave (one wave per workgroup): LDS operations of a wave complete in order, so only the
// compiler has to be kept from moving a read above the write it depends on — no s_barrier, and above all no vmcnt(0), which
// __syncthreads() implies and which would wait out the next case's prefetch.
__device__ __forceinline__ void lds_wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Persistent waves, one case at a time per wave; everything a case needs from memory (its scalars, the A operand, the first 64
// neighbours) is one batch of independent loads issued while the previous case is being computed.
template <int DIM, int ORDER>
__global__ __launch_bounds__(64) void sens_apply_kernel(const KParams p, const double* __restrict__ inv_all) {
    constexpr int WV = 64, NO = ndofs(DIM, ORDER), NOP = NO | 1;          // odd row stride in LDS
    constexpr int RB = (NO + 15) / 16, KS = (NO + 3) / 4;                  // 16-row blocks of the inverse, 4-column steps
    constexpr unsigned long long FULL = (NO >= 64) ? ~0ull : ((1ull << NO) - 1ull);
    __shared__ double sO[WV * NOP];
    const int lane = threadIdx.x, n = lane & 15, g = lane >> 4;
    const int kmax = (int)p.max_nk;
    const double qnan = __longlong_as_double(0x7ff8000000000000LL);

    struct In { int nk, wm; long long kn; double xi[DIM]; double A[RB][KS]; double x0[DIM]; };
    auto fetch = [&](long long t, In& q) {
        q.nk = p.nk[t * p.snk]; q.wm = p.wm[t * p.swm]; q.kn = p.knowns[t * p.sknowns];
#pragma unroll
        for (int m = 0; m < DIM; ++m) q.xi[m] = p.xi[t * p.sxi_j + m];
        // A operand of v_mfma_f64_16x16x4 (lane l: A[l % 16][l / 16]): rows 16 rb + n, columns 4 s + g of the inverse
        // (rows and columns of known DOFs are zero there)
        const double* inv = inv_all + t * (long long)(NO * NO);
#pragma unroll
        for (int rb = 0; rb < RB; ++rb)
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const int i = 16 * rb + n, b = 4 * s + g;
                q.A[rb][s] = (i < NO && b < NO) ? inv[i * NO + b] : 0.0;
            }
        const int k0 = lane < kmax ? lane : kmax - 1;                      // rows have max_nk slots
#pragma unroll
        for (int m = 0; m < DIM; ++m) q.x0[m] = p.xk[t * p.sxk_j + k0 * p.sxk_k + m];
    };

    // element q = lane + 64 i of a 64-row run sits at LDS index q + (q / no) * (NOP - no)
    int lrow[NO];
#pragma unroll
    for (int i = 0; i < NO; ++i) lrow[i] = ((lane + WV * i) / NO) * (NOP - NO);

    long long t = blockIdx.x;
    if (t >= p.ncases) return;
    In nxt;
    fetch(t, nxt);
    for (; t < p.ncases; t += gridDim.x) {
        const In cur = nxt;
        {
            const long long tn = t + gridDim.x < p.ncases ? t + gridDim.x : p.ncases - 1;
            fetch(tn, nxt);
        }
        const long long j = t;
        const int nkc = min(cur.nk, kmax);
        const bool uniform = (cur.wm == WLSQM_WEIGHT_UNIFORM);
        unsigned long long known, dropped;
        effective_mask<NO>(cur.kn, known, dropped);
        if (known == FULL) continue;                                       // nr < 1: no-op (impl.pyx:574, 636, 742)
        const double* xr = p.xk + j * p.sxk_j;

        // largest squared distance of the case (infra.pyx:668-702)
        double max_d2 = 0.0;
        if (!uniform) {
            if (lane < nkc) {
#pragma unroll
                for (int m = 0; m < DIM; ++m) { const double dd = cur.x0[m] - cur.xi[m]; max_d2 = fma(dd, dd, max_d2); }
            }
            for (int k = lane + WV; k < nkc; k += WV) {
                double d2 = 0.0;
#pragma unroll
                for (int m = 0; m < DIM; ++m) { const double dd = xr[k * p.sxk_k + m] - cur.xi[m]; d2 = fma(dd, dd, d2); }
                max_d2 = d2 > max_d2 ? d2 : max_d2;
            }
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) { const double o = __shfl_xor(max_d2, off, WV); max_d2 = o > max_d2 ? o : max_d2; }
        }
        const double inv_max = inverse_max(max_d2);

        for (int kb = 0; kb < nkc; kb += WV) {
            double xc[DIM];                                                // neighbour kb + lane
            if (kb == 0) {
#pragma unroll
                for (int m = 0; m < DIM; ++m) xc[m] = cur.x0[m];
            } else {
                const int kk = kb + lane < kmax ? kb + lane : kmax - 1;
#pragma unroll
                for (int m = 0; m < DIM; ++m) xc[m] = xr[kk * p.sxk_k + m];
            }
            lds_wave_sync();                                               // the previous 64 rows have left LDS
            // ---- lane = neighbour: its weighted monomial row w c[k, :] (impl.pyx:826-829) into LDS row `lane`
            {
                double d[DIM], c[NO];
#pragma unroll
                for (int m = 0; m < DIM; ++m) d[m] = xc[m] - cur.xi[m];
                const double d2 = monomials<DIM, ORDER>(d, c);
                const double w = weight(d2, inv_max, uniform);
#pragma unroll
                for (int b = 0; b < NO; ++b) sO[lane * NOP + b] = (b == 0) ? w : w * c[b];
            }
            lds_wave_sync();
            // ---- 16 neighbours per MFMA block; the block's rows of LDS are read as the B operand (lane l: B[l / 16][l % 16] =
            // entry 4 s + g of neighbour n) and then overwritten in place with the results (LDS operations of a wave are in order)
#pragma unroll 1
            for (int nb = 0; nb < 4; ++nb) {
                if (kb + nb * 16 >= nkc) break;                            // wave-uniform; rows past nk are computed and not stored
                double* row = sO + (nb * 16 + n) * NOP;
                double B[KS];
#pragma unroll
                for (int s = 0; s < KS; ++s) B[s] = (4 * s + 3 < NO || 4 * s + g < NO) ? row[4 * s + g] : 0.0;
                sd4_ acc[RB];
#pragma unroll
                for (int rb = 0; rb < RB; ++rb) {
                    acc[rb] = sd4_{0.0, 0.0, 0.0, 0.0};
#pragma unroll
                    for (int s = 0; s < KS; ++s) acc[rb] = __builtin_amdgcn_mfma_f64_16x16x4f64(cur.A[rb][s], B[s], acc[rb], 0, 0, 0);
                }
                lds_wave_sync();
                // D[4 v + l / 16][l % 16]: DOF 16 rb + 4 v + g of neighbour n
#pragma unroll
                for (int rb = 0; rb < RB; ++rb)
#pragma unroll
                    for (int v = 0; v < 4; ++v) {
                        const int a = 16 * rb + 4 * v + g;
                        if (16 * rb + 4 * v + 3 < NO || a < NO) row[a] = acc[rb][v];
                    }
            }
            if (known) {                                                   // NaN for knowns (impl.pyx:821-823)
                lds_wave_sync();
#pragma unroll
                for (int a = 0; a < NO; ++a)
                    if ((known >> a) & 1ull) sO[lane * NOP + a] = qnan;
            }
            lds_wave_sync();
            // the rows of these 64 neighbours: one contiguous run of the output when its rows are `no` doubles long
            const int total = min(WV, nkc - kb) * NO;
            if (p.ss_k == NO && dropped == 0) {
                double* out = p.sens + j * p.ss_j + (long long)kb * NO;
#pragma unroll
                for (int i = 0; i < NO; ++i) {
                    const int q = lane + WV * i;
                    if (q < total) out[q] = sO[q + lrow[i]];
                }
            } else {
                double* out = p.sens + j * p.ss_j + (long long)kb * p.ss_k;
#pragma unroll
                for (int i = 0; i < NO; ++i) {
                    const int q = lane + WV * i;
                    if (q < total) {
                        const int r = q / NO, a = q - r * NO;
                        if (!((dropped >> a) & 1ull)) out[(long long)r * p.ss_k + a] = sO[r * NOP + a];
                    }
                }
            }
        }
    }
}

// Waves per resident slot: a wave should see several cases (the prefetch pays from the second one on), and the tail of the
// launch should stay balanced.
template <int DIM, int ORDER>
static int launch_apply(const KParams& p, const double* inv, hipStream_t stream) {
    static KernelSetup setup;
    auto kern = sens_apply_kernel<DIM, ORDER>;
    long long grid = 0;
    int rc = persistent_grid(reinterpret_cast<const void*>(kern), 64, 0, 0, true, setup, &grid);
    if (rc != WLSQM_OK) return rc;
    const char* e = getenv("WLSQM_HIP_SENS_GRID_MULT");
    const double mult = (e && atof(e) > 0.0) ? atof(e) : 2.0;
    grid = (long long)((double)grid / grid_multiple() * mult);
    if (grid < 1) grid = 1;
    if (grid > p.ncases) grid = p.ncases;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(64), 0, stream, p, inv);
    WLSQM_HIP_CHECK(hipGetLastError());
    return WLSQM_OK;
}

static int apply_dispatch(int dimension, int order, const KParams& p, const double* inv, hipStream_t stream) {
#define ACASE(D, O) if (dimension == D && order == O) return launch_apply<D, O>(p, inv, stream);
    ACASE(1, 0) ACASE(1, 1) ACASE(1, 2) ACASE(1, 3) ACASE(1, 4)
    ACASE(2, 0) ACASE(2, 1) ACASE(2, 2) ACASE(2, 3) ACASE(2, 4)
    ACASE(3, 0) ACASE(3, 1) ACASE(3, 2) ACASE(3, 3) ACASE(3, 4)
#undef ACASE
    set_error("sens_apply: unsupported (dimension, order)");
    return WLSQM_EVALUE;
}

// Dense input (any strides for the 3D order-3/4 systems; contiguous even-K rows for the others — api.hip repacks), sensitivities
// without refinement.  WLSQM_HIP_DISABLE_SENS_APPLY=1 leaves these shapes to the generic kernels (A/B);
// WLSQM_HIP_SENS_SLICE_MB sets the size of a slice's inverses (default 256).
int launch_fit_sens(int dimension, int order, const KParams& p, long long K, hipStream_t stream, bool* handled) {
    *handled = false;
    const char* off = getenv("WLSQM_HIP_DISABLE_SENS_APPLY");
    const char* off2 = getenv("WLSQM_HIP_DISABLE_TILE");
    if ((off && off[0] == '1') || (off2 && off2[0] == '1')) return WLSQM_OK;
    if (!p.do_sens || !p.sens || p.iterative || p.hoods || p.case_index || !p.xk || !p.fk || K < 1) return WLSQM_OK;
    const int no = wlsqm_hip_number_of_dofs(dimension, order);
    const bool big = no > 15;
    if (big ? !(dimension == 3 && (order == 3 || order == 4)) : !chunk_inverse_ok(dimension, order, p, K)) return WLSQM_OK;
    if (p.ncases > 0x7fffffffLL) return WLSQM_OK;
    *handled = true;
    const char* mb = getenv("WLSQM_HIP_SENS_SLICE_MB");
    const double slice_mb = (mb && atof(mb) > 0.0) ? atof(mb) : 256.0;
    long long per = (long long)(slice_mb * 1048576.0 / (8.0 * no * no));
    per = per < 1024 ? 1024 : per;
    per = (per / 16) * 16;                                                 // whole tiles
    if (per > p.ncases) per = p.ncases;
    double* inv = nullptr;
    int rc = scratch_alloc_async(reinterpret_cast<void**>(&inv), (size_t)per * no * no * sizeof(double), stream);
    if (rc != WLSQM_OK) return rc;
    for (long long j0 = 0; j0 < p.ncases && rc == WLSQM_OK; j0 += per) {
        const long long n = (p.ncases - j0 < per) ? (p.ncases - j0) : per;
        KParams q = slice_cases(p, j0, n);
        q.max_nk = K;
        if (big) rc = launch_fit_wave_inverse(dimension, order, q, inv, stream);
        else {
            KParams a = q;
            a.ws = inv; a.do_sens = 0; a.sens = nullptr;
            rc = launch_fit_chunk_inverse(dimension, order, a, K, stream);
        }
        if (rc == WLSQM_OK) rc = apply_dispatch(dimension, order, q, inv, stream);
    }
    const int rc2 = scratch_free_async(inv, stream);
    if (rc == WLSQM_OK) note_kernel("sens-apply");
    return rc != WLSQM_OK ? rc : rc2;
}

}  // namespace wlsqm
