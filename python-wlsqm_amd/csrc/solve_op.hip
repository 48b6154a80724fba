// solve_op.hip — many right-hand sides on one prepared geometry as ONE batched GEMM on the matrix cores
// (BASELINE configs[3], north star: "MFMA ... only for the optional many-RHS ExpertSolver.solve path where the factored system x
// stacked RHS really is a batched dense GEMM").
//
// Reference: ExpertSolver.solve (expert.pyx:467-655) keeps the factored matrices of prepare() and runs dgetrs per case and field
// (impl.pyx:731-846).  Here the prepared state of a case is its SOLUTION OPERATOR  S = M_red^{-1} (W C)_red^T  [no x nk]  — exactly
// the sensitivities d fi / d fk of impl.pyx:821-834, computed once by the fit kernels' do_sens path — so that a field is
//     fi[unknown] = S (f - C[:, known] v) = S f - T v,     T = S C[:, known]  [no x (number of knowns)],  v = the known values,
// (impl.pyx:792-818: knowns move to the right-hand side) and R fields of a case are the GEMM  [no x nk] x [nk x R].
//
// Kernel (solve_op_mfma_kernel): one wave per case at a time; v_mfma_f64_16x16x4_f64 with
//     A[i][k] = S (rows padded to 16, DOF a = 4 (i % 4) + i / 4 in row i),  B[k][j] = fk of field r0 + j,  D[i][j] -> fi.
// Register layout measured on gfx950 (tools/ubench/mfma_f64_layout.hip): lane l holds A[l % 16][l / 16], B[l / 16][l % 16] and
// D[4 v + l / 16][l % 16] in accumulator element v.  The k index of MFMA step s in lane group q = l / 16 is chosen as
// 8 (s / 2) + 2 q + s % 2: with every 16-byte load instruction the four lane groups of a row read four CONSECUTIVE pieces — 64
// contiguous bytes of the field's fk row (and of the operator row) per instruction.  (First version: k = q K/4 + s, every lane
// one contiguous K/4-double piece of its own: each load instruction then touched four 16-byte pieces 64 bytes apart in each of its
// 16 rows; BASELINE configs[3]: 17.0 -> 15.4 ms per 256 fields, 0.58 -> 0.64 of the HBM peak.)  With the row permutation above a
// lane's four accumulator elements are four consecutive DOFs of one fi row.  The operator of the case (K/4 doubles per lane) stays
// in registers for all R fields; per case and field only fk (8 nk B) is read and fi (8 no B) written.  The fk pieces of four
// (three) blocks of 16 fields are requested at the top of an iteration and every block waits for its own set only.
//
// Rate measured with the same microbenchmark: 106 cycles per v_mfma_f64_16x16x4_f64 per SIMD at four waves per SIMD (47 TFLOP/s;
// the fp64 VECTOR pipe reaches 74) — the matrix cores are not the faster fp64 engine on MI355X, but the GEMM is bound by the fk
// stream (2-3 flop per byte), and the MFMA does the reduction over the neighbours that an FMA loop pays wave shuffles for and
// leaves the vector pipe to the loads.  A/B against the FMA kernel of solve_many.hip: DESIGN.md section 6.
#include <algorithm>
#include <vector>

#include "wlsqm_internal.hpp"
#include "wlsqm_kernels.hpp"

#ifndef WLSQM_OP_NT
#define WLSQM_OP_NT 2      // 1 = non-temporal loads of the fields (configs[3]: 16.9 against 14.7 ms), 2 = non-temporal stores of the results (1.0-1.6 % faster in three interleaved pairs, profiles/r03i_ab_c4_nt.txt: kept), 3 = both
#endif

namespace wlsqm {

constexpr int OP_ROWS = 16;       // rows of the MFMA's A operand (no <= 15 padded); the STORED operator has `no` rows per case, DOF order
constexpr int OP_NKN = 4;         // known DOFs per case the correction term is stored for

typedef double od2_ __attribute__((ext_vector_type(2)));
typedef double od4_ __attribute__((ext_vector_type(4)));

__host__ __device__ constexpr int op_row_of_dof(int a) { return 4 * (a % 4) + a / 4; }
__host__ __device__ constexpr int op_dof_of_row(int i) { return 4 * (i % 4) + i / 4; }

// sens[ncases, K, no] (NaN for knowns, slots >= nk untouched) -> op[ncases, no, KP] (DOF order; zero rows for knowns).  The rows
// the MFMA pads to 16 are not stored (round 2 stored all 16: 4 KB instead of 1.5 KB per case of 2D order 2).
__global__ void op_transpose_kernel(const double* __restrict__ sens, const int* __restrict__ nk, const long long* __restrict__ knowns,
                                    long long ncases, int K, int KP, int no, double* __restrict__ op) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= ncases * K) return;
    const long long j = t / K; const int k = (int)(t - j * K);
    const bool live = k < nk[j];
    // DOFs the fit does not solve for (knowns, and the unknowns a mask with stray high bits drops: infra.pyx:119-121): zero rows.
    // (The sens block is not cleared beforehand: what the fit kernels leave untouched there is never read.)
    unsigned long long known, dropped;
    {
        const unsigned long long full = (1ull << no) - 1ull, raw = (unsigned long long)knowns[j];
        known = raw & full; dropped = 0;
        int extra = __popcll(raw & ~full);
        for (int b = no - 1; b >= 0 && extra > 0; --b)
            if (!((known >> b) & 1ull)) { known |= 1ull << b; --extra; }
    }
    const double* s = sens + t * no;
    for (int a = 0; a < no; ++a) {
        const double v = (live && !((known >> a) & 1ull)) ? s[a] : 0.0;
        op[(j * no + a) * KP + k] = (v == v) ? v : 0.0;
    }
}

// T[j][row][t] = sum_k op[j][row][k] * c_k[a_t]   (a_t = t-th true known DOF of the case, ascending)
template <int DIM, int ORDER>
__global__ void op_known_kernel(const double* __restrict__ op, const double* __restrict__ xk, const double* __restrict__ xi,
                                const int* __restrict__ nk, const long long* __restrict__ knowns, long long ncases, int K, int KP,
                                double* __restrict__ T) {
    constexpr int NO = ndofs(DIM, ORDER);
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= ncases * OP_ROWS) return;
    const long long j = t / OP_ROWS; const int row = (int)(t - j * OP_ROWS);
    unsigned long long known, dropped;
    effective_mask<NO>(knowns[j], known, dropped);
    const unsigned long long tk = known & ~dropped;
    double acc[OP_NKN] = {0.0, 0.0, 0.0, 0.0};
    if (tk && op_dof_of_row(row) < NO) {
        const int n = min(nk[j], K);
        for (int k = 0; k < n; ++k) {
            double d[DIM], cc[NO];
#pragma unroll
            for (int m = 0; m < DIM; ++m) d[m] = xk[(j * K + k) * DIM + m] - xi[j * DIM + m];
            monomials<DIM, ORDER>(d, cc);
            const double s = op[(j * NO + op_dof_of_row(row)) * KP + k];
            int slot = 0;
#pragma unroll
            for (int a = 0; a < NO; ++a)
                if (((tk >> a) & 1ull) && slot < OP_NKN) { acc[slot] = fma(s, (a == 0) ? 1.0 : cc[a], acc[slot]); ++slot; }
        }
    }
#pragma unroll
    for (int s = 0; s < OP_NKN; ++s) T[t * OP_NKN + s] = acc[s];
}

struct OpParams {
    const double* op; const double* T;
    const int* nk; const long long* knowns;
    long long ncases; int K, no; int any_known; int dbg;       // K: fk slots per row (even); the operator rows hold KP = 4 KQ >= K
    unsigned inv_no;               // ceil(2^32 / no): exact quotients for the store phase's small indices
    long long nrhs;
    const double* fk; long long sfk_r, sfk_j;
    double* fi; long long sfi_r, sfi_j;
};

// KQ = K / 4 (even: 16-byte pieces).  A workgroup of WPG waves takes WPG CONSECUTIVE cases, one per wave; per block of 16
// fields every wave multiplies its case and parks the 16 x no results in LDS, and after one barrier the whole workgroup writes,
// for each of the 16 fields, the WPG * no CONSECUTIVE doubles of fi that its cases own (8 B per lane, consecutive lanes on
// consecutive addresses).  Measured on the 3D order-2 geometry (1M cases, 64 fields): with every wave storing its own 80-byte
// row per field (16 rows x 4 pieces of 8 B per store instruction) the kernel took 0.139 ms per field, 0.075 ms with the stores
// removed and 0.088 ms with the loads removed — the scattered partial-line writes, not the fk stream, set the pace.
// KNOWN: some case has known DOFs (the correction columns cost 32 registers).  The fk pieces of the next TWO blocks of fields
// are in flight while one is multiplied (three register sets in rotation).
template <int KQ, bool KNOWN, int WPG>
__global__ __launch_bounds__(64 * WPG) void solve_op_mfma_kernel(const OpParams P) {
    constexpr int KP = 4 * KQ;                                        // operator row length: the fk row length P.K rounded up to 8
    extern __shared__ __attribute__((aligned(16))) double lds[];      // [2][16][runp] results, then [WPG] known masks
    const int wave = threadIdx.x >> 6, l = threadIdx.x & 63, q = l >> 4, c16 = l & 15;
    const int no = P.no, run = WPG * no;                              // doubles per field and workgroup
    // Field pitch of the result slab: ODD.  A ds_write_b64 is served in groups of 16 consecutive lanes on 32 dword banks, and the
    // 16 lanes of a group are the 16 fields c16 of one (wave, q): with the pitch = run (96 doubles for 2D order 2: a multiple of
    // 16) all 16 hit ONE pair of banks (round 2: SQ_LDS_BANK_CONFLICT = 80 % of the LDS-active cycles; now 0: profiles/
    // r03a_C4_pmc_summary.json); an odd pitch spreads them over all 16 pairs.  The store phase reads a field's run contiguously.
    const int runp = run | 1;
    const unsigned long long full = (1ull << no) - 1ull;
    unsigned long long* smask = reinterpret_cast<unsigned long long*>(lds + 2 * 16 * runp);
    for (long long j0 = (long long)blockIdx.x * WPG; j0 < P.ncases; j0 += (long long)gridDim.x * WPG) {
        const long long j = j0 + wave;
        const bool have = j < P.ncases;                               // wave-uniform
        const long long jc = have ? j : P.ncases - 1;
        // per-case scalars (wave-uniform)
        const int nkc = min(P.nk[jc], P.K);
        const unsigned long long raw = (unsigned long long)P.knowns[jc];
        unsigned long long known = raw & full, dropped = 0;
        {
            int extra = __popcll(raw & ~full);                                  // infra.pyx:119-121 quirk (effective_mask)
            for (int t = no - 1; t >= 0 && extra > 0; --t)
                if (!((known >> t) & 1ull)) { known |= 1ull << t; dropped |= 1ull << t; --extra; }
        }
        const bool work = have && known != full;
        const unsigned long long tk = known & ~dropped;
        __syncthreads();                                              // the previous group's store phase has read smask
        if (l == 0) smask[wave] = have ? known : full;                // DOFs the store phase must leave alone
        // operator piece of this lane: row c16, k in [q KQ, (q + 1) KQ)
        double A[KQ];
        const int adof = op_dof_of_row(c16);                          // DOF whose operator row is row c16 of the A operand
        if (adof >= no) {
#pragma unroll
            for (int s = 0; s < KQ; ++s) A[s] = 0.0;                  // padding rows of the 16-row operand: not stored, not loaded
        } else {
            // k index of element s of lane group q: 8 (s / 2) + 2 q + s % 2 — the four lane groups of a row read four CONSECUTIVE 16-byte
            // pieces with every load instruction (64 contiguous bytes per field / operator row) instead of one piece each from four
            // places 64 bytes apart; the sum over k does not care about the order, operator and fk use the same one
            const od2_* src = reinterpret_cast<const od2_*>(P.op + (jc * no + adof) * (long long)KP + 2 * q);
#pragma unroll
            for (int s = 0; s < KQ / 2; ++s) { const od2_ v = src[4 * s]; A[2 * s] = v.x; A[2 * s + 1] = v.y; }
        }
        // correction rows of this lane's four output DOFs (rows 4 v + q) and the known DOFs' positions
        double Tl[KNOWN ? 4 : 1][OP_NKN];
        int ka[OP_NKN] = {0, 0, 0, 0}; int nkn = 0;
        if (KNOWN && tk) {
            for (int a = 0; a < no && nkn < OP_NKN; ++a) if ((tk >> a) & 1ull) ka[nkn++] = a;
#pragma unroll
            for (int v = 0; v < 4; ++v)
#pragma unroll
                for (int t = 0; t < OP_NKN; ++t) Tl[KNOWN ? v : 0][t] = P.T[((jc * OP_ROWS + 4 * v + q) * OP_NKN) + t];
        }
        const bool ragged = nkc < KP;
        const double* frow = P.fk + jc * P.sfk_j;
        auto load_b = [&](long long r0, double (&B)[KQ]) {
            // (always issued — rows past the last field replay it, a case without unknowns loads and ignores: no branch, so that
            // the compiler can count what is in flight)
            long long r = r0 + c16; r = r < P.nrhs ? r : P.nrhs - 1;
            const double* src = frow + r * P.sfk_r;
#pragma unroll
            for (int s = 0; s < KQ / 2; ++s) {
                // pieces beyond the fk row (K not a multiple of 8) replay the row's first pair: their operator columns are zero and
                // the ragged mask below clears them
                const int e = 8 * s + 2 * q;
#if (WLSQM_OP_NT & 1)
                const od2_ v = __builtin_nontemporal_load(reinterpret_cast<const od2_*>(src + (e < P.K ? e : 0)));
#else
                const od2_ v = *reinterpret_cast<const od2_*>(src + (e < P.K ? e : 0));
#endif
                B[2 * s] = v.x; B[2 * s + 1] = v.y;
            }
        };
        int parity = 0;
        auto stage = [&](long long r0, double (&B)[KQ]) {
            double* out = lds + parity * 16 * runp;
            if (work) {
                if (ragged) {
                    // slots beyond nk[j] may hold anything (padding of the device rows): 0 * NaN would poison the sum
#pragma unroll
                    for (int s = 0; s < KQ; ++s) B[s] = (8 * (s / 2) + 2 * q + (s & 1) < nkc) ? B[s] : 0.0;
                }
                // (two accumulators, to halve the chain of dependent MFMAs: no gain measured on the 64-neighbour geometry at two
                // waves per SIMD, and the extra registers put the 128-register fit of the 16-wave workgroups at risk)
                od4_ acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int s = 0; s < KQ; ++s) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(A[s], B[s], acc, 0, 0, 0);
                if (KNOWN && nkn) {
                    // knowns of this field (impl.pyx:815-818: values from its fi row) times the stored correction columns
                    // (loading the first known value early, with the field's fk piece, measured slower: 0.228 against 0.20 ms per field on
                    // the 64-neighbour geometry — 14 more registers)
                    long long r = r0 + c16; r = r < P.nrhs ? r : P.nrhs - 1;
                    const double* fin = P.fi + r * P.sfi_r + jc * P.sfi_j;
                    for (int t = 0; t < nkn; ++t) {
                        const double vt = fin[ka[t]];
#pragma unroll
                        for (int v = 0; v < 4; ++v) acc[v] = fma(-Tl[KNOWN ? v : 0][t], vt, acc[v]);
                    }
                }
                double* mine = out + c16 * runp + wave * no + 4 * q;           // [field][case][dof], field pitch runp
#pragma unroll
                for (int v = 0; v < 4; ++v)
                    if (4 * q + v < no) mine[v] = acc[v];
            }
            // LDS-only barrier: __syncthreads() is also `s_waitcnt vmcnt(0)`, which would wait out the fk pieces of the next two
            // blocks of fields (in flight on purpose) and the previous block's stores at every block
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            // store phase: element e of the [16][WPG * no] block -> field e / run, position e % run of that field's run
            const long long nfield = (P.nrhs - r0 < 16) ? (P.nrhs - r0) : 16;
            const int total = (int)nfield * run;
            // Up to 8 unknowns the 16 no / 64 <= 2 rounds are written out: a real loop makes the compiler flush vmcnt in its preheader,
            // i.e. wait for the fk pieces in flight (four rounds unrolled for every `no` spill 276 B per lane at the 128 registers
            // of the 16-wave workgroups)
            auto put = [&](int e) {
                const int row = no == 1 ? e : (int)__umulhi((unsigned)e, P.inv_no);   // e / no = f * WPG + cs (e < 2^12: exact; no == 1: 2^32 does not fit inv_no)
                const int a = e - row * no;
                const int f = row / WPG, cs = row - f * WPG;                    // WPG is a power of two
                if (!((smask[cs] >> a) & 1ull) && !(P.dbg & 1))
#if (WLSQM_OP_NT & 2)
                    __builtin_nontemporal_store(out[e + f * (runp - run)], &P.fi[(r0 + f) * P.sfi_r + (j0 + cs) * P.sfi_j + a]);
#else
                    P.fi[(r0 + f) * P.sfi_r + (j0 + cs) * P.sfi_j + a] = out[e + f * (runp - run)];
#endif
            };
            if (no <= 8) {
                const int e0 = (int)threadIdx.x, e1 = e0 + 64 * WPG;
                if (e0 < total) put(e0);
                if (e1 < total) put(e1);
            } else {
                for (int e = threadIdx.x; e < total; e += 64 * WPG) put(e);
            }
            parity ^= 1;
        };
        // Four (three) blocks of 16 fields per iteration: the fk pieces of all of them are requested at the top and every block waits
        // for its own set only (s_waitcnt vmcnt(12 / 8 / 4 / 0) plus the stores in between).  Register sets kept in flight ACROSS the loop
        // header do not work with this compiler: it loses their pending count there and waits with vmcnt(0) at the first use —
        // the three rotating sets of the first version were requested and then waited for in every block (ISA), no prefetch at all.
        // Round 3 tried to keep the sets in flight ACROSS the header anyway: fk pieces requested by inline assembly (invisible to
        // the compiler's wait insertion), hand-written s_waitcnt vmcnt(N) with N counted exactly (later sets x loads + this wave's
        // stores in between, by ballot), one block per step in rotation.  Correct (all stacked-solve tests) and SLOWER: configs[3]
        // 14.5 -> 15.8 ms (conservative counts) / 16.1 ms (exact counts) — with one 16-wave workgroup per CU in lock step at the
        // barrier, a burst of 12 requests per wave followed by three compute + store phases suits this memory system better than
        // requests interleaved with the stores.  Kept as tools/experiments/solve_op_asm_pipeline.hip.
#ifndef WLSQM_OP_NSET16
#define WLSQM_OP_NSET16 3
#endif
        constexpr int NSET = (WPG == 16 && KQ >= 8) ? WLSQM_OP_NSET16 : 4;   // (the 16-wave workgroups have 128 registers per lane)
        double B[NSET][KQ];
        for (long long r0 = 0; r0 < P.nrhs; r0 += 16 * NSET) {
#pragma unroll
            for (int b = 0; b < NSET; ++b) {
                load_b(r0 + 16 * b, B[b]);
                __builtin_amdgcn_sched_barrier(0);                          // in this order: the first block's set must not be the last one requested
            }
#pragma unroll
            for (int b = 0; b < NSET; ++b)
                if (r0 + 16 * b < P.nrhs) stage(r0 + 16 * b, B[b]);
        }
    }
}

// ---- host side

long long preferred_slots(int dimension, int order, long long max_nk);
int launch_fit(int dimension, int order, const KParams& p, long long max_nk, hipStream_t stream);

bool solve_op_shape_ok(int dimension, int order, long long K) {
    const int no = ndofs(dimension, order);
    return order >= 0 && no <= 15 && K >= 10 && K <= 64 && (K % 2) == 0;      // (the operator rows are padded to a multiple of 8)
}

// Build the operator (and the correction columns) of the dense resident geometry in `geom` (its fk / fi / sens are ignored).
// *bytes_needed is filled even on failure paths that return WLSQM_OK with *ok = false (shape not supported).
int solve_op_build(int dimension, int order, const KParams& geom, long long K, const long long* h_knowns, long long ncases,
                   DevBuf& d_op, DevBuf& d_T, int* any_known, hipStream_t s, bool* ok) {
    *ok = false;
    if (!solve_op_shape_ok(dimension, order, K)) return WLSQM_OK;
    const int no = ndofs(dimension, order);
    // knowns: at most OP_NKN true knowns per case (cases with every DOF known have nothing to solve)
    int anyk = 0;
    for (long long j = 0; j < ncases; ++j) {
        unsigned long long known, dropped;
        effective_mask_host(no, h_knowns[j], known, dropped);
        const unsigned long long fullm = (1ull << no) - 1ull;
        if (known == fullm) continue;
        const int c = __builtin_popcountll(known & ~dropped);
        if (c > OP_NKN) return WLSQM_OK;
        if (c) anyk = 1;
    }
    int rc;
    const long long KP = (K + 7) / 8 * 8;
    if ((rc = d_op.alloc((size_t)ncases * no * KP * 8))) return rc;
    if ((rc = d_T.alloc(anyk ? (size_t)ncases * OP_ROWS * OP_NKN * 8 : 16))) return rc;
    // (the transpose writes every column k < K of every row; only the pad columns K .. KP - 1 need zeros)
    if (KP != K) WLSQM_HIP_CHECK(hipMemsetAsync(d_op.p, 0, d_op.n, s));
    // sensitivities of the geometry, a chunk of cases at a time (the dense sens block is 8 K no bytes per case).  Round 3: the
    // temporaries come from the stream-ordered pool (a chunk is <= 128 MB of sens: no hipMalloc / hipFree per build — those, not
    // the kernels, were most of the 9-37 ms a build took), the sens block is not cleared (the transpose reads live slots only)
    long long chunk = (128ll << 20) / ((long long)K * no * 8);
    chunk = std::max(4096ll, chunk - chunk % 64);
    chunk = std::min(chunk, ncases);
    const size_t n_sens = (size_t)chunk * K * no, n_fk = (size_t)chunk * K, n_fi = (size_t)chunk * no;
    double* tmp = nullptr;
    if ((rc = scratch_alloc_async(reinterpret_cast<void**>(&tmp), (n_sens + n_fk + n_fi) * sizeof(double), s))) return rc;
    double* t_sens = tmp; double* t_fk = tmp + n_sens; double* t_fi = t_fk + n_fk;
    hipError_t e = hipMemsetAsync(t_fk, 0, (n_fk + n_fi) * sizeof(double), s);
    if (e != hipSuccess) { (void)scratch_free_async(tmp, s); return hip_fail(e, "hipMemsetAsync"); }
    for (long long j0 = 0; j0 < ncases; j0 += chunk) {
        const long long n = std::min(chunk, ncases - j0);
        KParams p = slice_cases(geom, j0, n);
        p.fk = t_fk; p.sfk_j = K; p.sfk_k = 1;
        p.fi = t_fi; p.sfi_j = no;
        p.sens = t_sens; p.ss_j = K * no; p.ss_k = no;
        p.do_sens = 1; p.iterative = 0; p.iters_out = nullptr; p.case_index = nullptr;
        if ((rc = launch_fit(dimension, order, p, K, s))) { (void)scratch_free_async(tmp, s); return rc; }
        const long long threads = n * K;
        hipLaunchKernelGGL(op_transpose_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s, t_sens,
                           p.nk, p.knowns, n, (int)K, (int)KP, no, d_op.as<double>() + j0 * no * KP);
        e = hipGetLastError();
        if (e != hipSuccess) { (void)scratch_free_async(tmp, s); return hip_fail(e, "op_transpose_kernel"); }
    }
    if ((rc = scratch_free_async(tmp, s))) return rc;
    if (anyk) {
        const long long threads = ncases * OP_ROWS;
        const dim3 grid((unsigned)((threads + 255) / 256)), block(256);
#define KN_CASE(D, O)                                                                                                    \
        if (dimension == D && order == O)                                                                               \
            hipLaunchKernelGGL((op_known_kernel<D, O>), grid, block, 0, s, d_op.as<double>(), geom.xk, geom.xi, geom.nk, \
                               geom.knowns, ncases, (int)K, (int)KP, d_T.as<double>());
        KN_CASE(1, 0) KN_CASE(1, 1) KN_CASE(1, 2) KN_CASE(1, 3) KN_CASE(1, 4)
        KN_CASE(2, 0) KN_CASE(2, 1) KN_CASE(2, 2) KN_CASE(2, 3) KN_CASE(2, 4)
        KN_CASE(3, 0) KN_CASE(3, 1) KN_CASE(3, 2)
#undef KN_CASE
        WLSQM_HIP_CHECK(hipGetLastError());
    }
    WLSQM_HIP_CHECK(hipStreamSynchronize(s));      // the operator is complete when this returns (a later solve may use another stream)
    *any_known = anyk;
    *ok = true;
    return WLSQM_OK;
}

int launch_solve_op(int dimension, int order, const KParams& geom, long long K, const double* op, const double* T, int any_known,
                    long long nrhs, const double* fk, long long sfk_r, long long sfk_j, double* fi, long long sfi_r, long long sfi_j,
                    hipStream_t stream, bool* handled) {
    *handled = false;
    if (!solve_op_shape_ok(dimension, order, K) || sfk_j != K) return WLSQM_OK;
    if ((reinterpret_cast<uintptr_t>(fk) & 15u) || (sfk_r % 2) != 0) return WLSQM_OK;       // 16-byte pieces of every field's rows
    const char* dbg = getenv("WLSQM_HIP_OP_DEBUG");      // experiments: 1 = no stores, 2 = only the first block of fields is loaded
    const int no_ = ndofs(dimension, order);
    OpParams P{op, T, geom.nk, geom.knowns, geom.ncases, (int)K, no_, any_known, dbg ? atoi(dbg) : 0,
               (unsigned)((0x100000000ull + (unsigned)no_ - 1) / (unsigned)no_), nrhs, fk, sfk_r, sfk_j, fi, sfi_r, sfi_j};
    int dev = 0;
    WLSQM_HIP_CHECK(hipGetDevice(&dev));
    static int cus[16] = {};
    if (dev >= 0 && dev < 16 && !cus[dev]) {
        hipDeviceProp_t prop;
        WLSQM_HIP_CHECK(hipGetDeviceProperties(&prop, dev));
        cus[dev] = prop.multiProcessorCount;
    }
    const int ncu = cus[dev < 16 && dev >= 0 ? dev : 0];
    const char* wenv = getenv("WLSQM_HIP_OP_WPG");                   // A/B: waves (= consecutive cases) per workgroup
    // (sfi_j == no is not required: the store phase addresses every row by its stride; contiguous rows make the runs contiguous)
#define OP_LAUNCH(KQ_, KN_, WPG_)                                                                                        \
    {                                                                                                                    \
        const size_t lds_bytes = sizeof(double) * 2 * 16 * ((WPG_ * P.no) | 1) + 8 * WPG_ + 8;                           \
        const long long wgs = (geom.ncases + WPG_ - 1) / WPG_;                                                           \
        long long grid = (long long)ncu * (16 / WPG_) * 4;   /* a few workgroups per resident slot, striding over the cases */ \
        if (grid > wgs) grid = wgs;                                                                                      \
        if (grid < 1) grid = 1;                                                                                          \
        hipLaunchKernelGGL((solve_op_mfma_kernel<KQ_, KN_, WPG_>), dim3((unsigned)grid), dim3(64 * WPG_), lds_bytes, stream, P); \
    }
#define OP_CASE(KQ_, WDEF_)                                                                                   \
    if ((K + 7) / 8 * 8 == 4 * KQ_) {                                                                        \
        const int wpg = wenv ? atoi(wenv) : WDEF_;                                                           \
        if (any_known) { if (wpg >= 8) OP_LAUNCH(KQ_, true, 8) else OP_LAUNCH(KQ_, true, 4) }                \
        else if (wpg >= 16 && KQ_ <= 10) OP_LAUNCH(KQ_, false, 16)                                           \
        else if (wpg >= 8) OP_LAUNCH(KQ_, false, 8)                                                          \
        else OP_LAUNCH(KQ_, false, 4)                                                                        \
        WLSQM_HIP_CHECK(hipGetLastError());                                                                  \
        *handled = true; note_kernel("solve-op-mfma"); return WLSQM_OK;                                      \
    }
    OP_CASE(4, 16) OP_CASE(6, 16) OP_CASE(8, 16) OP_CASE(10, 16) OP_CASE(12, 8) OP_CASE(14, 8) OP_CASE(16, 8)
#undef OP_CASE
#undef OP_LAUNCH
    return WLSQM_OK;
}

}  // namespace wlsqm
