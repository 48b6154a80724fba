// fit_stage_iter.hip — fit + ITERATIVE REFINEMENT with ONE LANE PER CASE (round 4): solve_iterative of the reference
// (impl.pyx:986-1083: fit, then sweeps of  residual at the neighbours -> correction with the same factor -> stop when two successive
// residual norms are equal, impl.pyx:1057) on the mapping of fit_stage.hip.
//
// Reference path (file:line in /root/reference): make_c_nD impl.pyx:286-432 / 70-269, Case_make_weights infra.pyx:668-702, make_A
// impl.pyx:566-602, solve with knowns elimination impl.pyx:731-846, the Taylor evaluation of the model polyeval.pyx:82-951,
// solve_iterative impl.pyx:986-1083 (return value: for / else, :1080-1081).
//
// Why.  The refinement kernels so far gave a case to FOUR lanes of a 16-case tile (fit_tilek.hip, fit_chunk.hip ITER): every lane of
// a case repeats the case's substitution, a sweep ends in a butterfly over the four lanes, the neighbours of a lane come in runs
// of two with a 15-deep dependent chain each for a lone wave per SIMD, and the fit in front of the sweeps is the slow chunked one
// (0.62 ms per 400k configs[2] cases against the staged kernel's 0.16).  With one lane per case there is one substitution per case and
// sweep, nothing to reduce, and GRP independent neighbours in flight per lane.  The factor (up to 120 entries) stays in the lane's
// registers between the sweeps.
//
// Where the rows are during the sweeps (the kernel reads them max_iter + 1 times):
//   RESIDENT   the wave's 64 rows fit LDS three times per CU (64 (K (DIM + 1) + 4) 8 bytes <= 53 KB: configs[1], 32 neighbours): they are
//              staged ONCE, whole, and every pass — largest distance, moments, sweeps — reads LDS; no speculation needed.
//   otherwise  every pass re-stages the rows in 8-neighbour chunks exactly as fit_stage.hip does (the chunks of a wave are 50-100 KB
//              that were read a few microseconds ago: they come back from L2 / the Infinity Cache, not from HBM).  The moment pass is
//              speculative as there (largest squared distance = the last neighbour's, verified bit for bit, repeated otherwise).
//              Where four waves per CU still fit beside the staging rows, 64 x K doubles per wave stay in LDS between the sweeps: the
//              WEIGHTS (2D: a sweep is bound by its instructions, and the weight is 14 of them per neighbour) or the VALUES fk (3D:
//              bound by the re-staged bytes; the sweeps then fetch the coordinates only).  See CACHE at the kernel.
// One sum per moment / per right-hand-side entry over k DESCENDING in one lane (the order of fit_stage.hip).  A case with exactly the
// function value known keeps the factor of its 14 x 14 system (2D order 4), as in the staged fit kernel.  Every form — resident,
// re-staging, cached or not — returns the same bits for a case (tests/test_gpu_round4.py::test_staged_refinement_kernel).
//
// Round 5: the 15-unknown re-staging forms take their chunks by LDS-DMA (DMA at the kernel: 168 -> 40 registers in scratch, C3+iter@400k
// 1.33 -> 1.10 ms; profiles/r05j_refine_dma.txt).
//
// Measured and left off (switches at the top; records under profiles/r04zb_*): two chunks in flight, the next pass's first chunk behind
// the current pass's last, two waves per SIMD for the 10-unknown systems, partial sums of the model evaluation, the caches at three
// waves per CU, and the SENS form (sensitivities on this mapping: correct, 1.2-1.9x slower than the inverse + matrix-core path).
#include <atomic>
#include <type_traits>

#include "wlsqm_internal.hpp"
#include "wlsqm_kernels.hpp"
#include "wlsqm_moments.hpp"

#ifndef WLSQM_SITER_GRP
#define WLSQM_SITER_GRP 4           // neighbours of a lane in flight in the moment pass and in a sweep
#endif
#ifndef WLSQM_SITER_MODEL_CHAINS
#define WLSQM_SITER_MODEL_CHAINS 1  // partial sums of the model evaluation in a sweep (3, also with -amdgpu-sched-strategy=max-ilp: flat, profiles/r04zb_ab_siter_chains.txt)
#endif
#ifndef WLSQM_SITER_REDUCED
#define WLSQM_SITER_REDUCED 1       // 15-unknown systems: a case with exactly F known keeps the factor of its 14 x 14 system
#endif
#ifndef WLSQM_SITER_WARM
#define WLSQM_SITER_WARM 0          // 1: the first chunk of the next pass is requested behind the last chunk of the current one (measured slower, off: see the kernel)
#endif
#ifndef WLSQM_SITER_TWO_WAVES_UPTO
#define WLSQM_SITER_TWO_WAVES_UPTO 6    // systems up to this many unknowns are compiled for two waves per SIMD (re-staging form)
#endif
#ifndef WLSQM_SITER_DEEP
#define WLSQM_SITER_DEEP 0          // 1: two chunks in flight for the systems with 7 .. 10 unknowns (see the kernel: measured slower, off)
#endif

namespace wlsqm {

namespace siter {
typedef double d2_ __attribute__((ext_vector_type(2)));
constexpr int CH = 8;               // neighbours per staged chunk
__host__ __device__ constexpr int sens_group(int no) { return no <= 6 ? 4 : 2; }       // neighbours per sens tile: at most 16 pieces of 16 bytes per case
__host__ __device__ constexpr int pitch2(int doubles) {               // row pitch in 16-byte units: odd, so that 16 lanes' b128 reads hit 16 different slots
    const int h = (doubles + 1) / 2;
    return (h % 2) ? h : h + 1;
}
}

// SENS: the sensitivities d fi[a] / d fk[k] as well (impl.pyx:776-778, 821-846: one substitution per neighbour with the kept factor),
// in one more pass over the rows between the fit and the sweeps.  A lane's G x NO results per group of G neighbours are G NO
// consecutive doubles of ITS case's sens block: they leave through an LDS tile, one case per store instruction (G NO 8 contiguous
// bytes), instead of 64 lanes storing 8 bytes each at a pitch of K NO 8 bytes.
// WCACHE (re-staging form, no sensitivities): the neighbours' WEIGHTS stay in LDS between the sweeps — the moment pass computes them
// anyway, and a sweep's reciprocal root + Newton steps + blend are 14 of its 52 (3D order 2) / 79 (2D order 4) vector instructions per
// neighbour.  64 x K doubles per wave at an odd pitch: chosen by the launcher while four waves per CU still fit beside the staging rows
// (2D: up to 48 neighbours, 3D: up to 40 — configs[4]).
// CACHE = 2: the neighbours' VALUES fk instead (the same 64 x K doubles): the sweeps then re-stage the coordinates only — 24 of 32 bytes
// per 3D neighbour.  For the shape whose sweeps are bound by the rows coming back from the Infinity Cache rather than by their
// instructions: 3D order 2 (configs[4]: the weight cache gave 1.5 % there, against 6-8 % on the 2D shapes).
// DMA (round 5): the re-staging forms of the 15-unknown systems take their chunks by global_load_lds_dwordx4 into a ring of two slots
// (lane-linear image and hand-counted waits: fit_stage.hip) — at the same one wave per SIMD: the point is the ~50 registers the staging
// sets, their addresses and the hoisted LDS reads occupied in a kernel that spilled into scratch next to its 105-entry factor.
#ifndef WLSQM_SITER_DMA
#define WLSQM_SITER_DMA 1
#endif
template <int DIM, int ORDER, bool RESIDENT, bool SENS, int CACHE = 0>
__global__ __launch_bounds__(64, (!RESIDENT && !SENS && CACHE == 0 && ndofs(DIM, ORDER) <= WLSQM_SITER_TWO_WAVES_UPTO) ? 2 : 1) void fit_stage_refine_kernel(const KParams p, const int XP2r, const int FP2r) {
    static_assert(CACHE == 0 || (!RESIDENT && !SENS), "the caches belong to the re-staging refinement form");
    constexpr bool WCACHE = CACHE == 1, FCACHE = CACHE == 2;
    using namespace siter;
    constexpr int NO = ndofs(DIM, ORDER), NE = NO * (NO + 1) / 2, NM = mom_count<DIM>(2 * ORDER);
    constexpr bool DMA = (WLSQM_SITER_DMA != 0) && !RESIDENT && !SENS && NO == 15;
    constexpr int SG = sens_group(NO), SR = SG * NO / 2, TP2 = pitch2(SG * NO);      // neighbours per sens tile; 16-byte pieces of a case's tile; tile pitch
    constexpr int TILE2 = SENS ? 64 * TP2 : 0;
    constexpr int XPC = CH * DIM * 8 / 16, FPC = CH * 8 / 16;        // 16-byte pieces of one case's chunk: coordinates, values
    constexpr int XCPI = 64 / XPC, XNI = (64 + XCPI - 1) / XCPI;      // whole cases per load instruction; instructions per chunk
    constexpr int FCPI = 64 / FPC, FNI = 64 / FCPI;
    constexpr int GRP = WLSQM_SITER_GRP < CH ? WLSQM_SITER_GRP : CH;
    constexpr int XP2s = pitch2(CH * DIM), FP2s = pitch2(CH);         // chunk staging pitches (not RESIDENT)
    // (DMA) 16-byte units from one load instruction's image to the next: a KiB + ONE slot (round 6, as in fit_stage.hip: dense KiB blocks put the
    // 16 lanes of a ds_read_b128 group on two to four slots of the bank row — `SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE` 0.84 on configs[2]
    // with refinement, profiles/r05_pmc_refine.txt; 2D: slot (8 (c mod 8)) mod 16 unskewed)
#ifndef WLSQM_SITER_DMA_SKEW
#define WLSQM_SITER_DMA_SKEW 1
#endif
    constexpr int KIB2 = 64 + WLSQM_SITER_DMA_SKEW;
    constexpr int SLOT2 = (XNI + FNI) * KIB2;                         // 16-byte units of a slot
    constexpr int STAGE2 = DMA ? 2 * SLOT2 : 64 * XP2s + 64 * FP2s, OUT2 = 64 * NO / 2;
    constexpr int ROWS2s = STAGE2 > OUT2 ? STAGE2 : OUT2;
    __shared__ __attribute__((aligned(16))) d2_ lds_s[RESIDENT ? 1 : ROWS2s + TILE2];
    extern __shared__ __attribute__((aligned(16))) d2_ lds_d[];
    d2_* const lds = RESIDENT ? lds_d : lds_s;
    const int XP2 = RESIDENT ? XP2r : XP2s, FP2 = RESIDENT ? FP2r : FP2s;
    d2_* const xs = lds;
    d2_* const fs = lds + 64 * XP2;
    d2_* const tile2 = lds + (RESIDENT ? 64 * (XP2 + FP2) : ROWS2s);      // (SENS) behind the rows

    const int lane = threadIdx.x;
    const long long t0 = (long long)blockIdx.x * 64, t = t0 + lane;
    const int nvalid = (p.ncases - t0 < 64) ? (int)(p.ncases - t0) : 64;      // wave-uniform
    const bool valid = lane < nvalid;
    const long long j = valid ? t : t0 + nvalid - 1;                          // tail lanes replay the last case (never stored)
    const int K = (int)p.max_nk;
    const int nkc = min(p.nk[j * p.snk], K);
    const bool uniform = (p.wm[j * p.swm] == WLSQM_WEIGHT_UNIFORM);
    unsigned long long known, dropped;
    effective_mask<NO>(p.knowns[j * p.sknowns], known, dropped);
    double xi[DIM];
#pragma unroll
    for (int m = 0; m < DIM; ++m) xi[m] = p.xi[j * p.sxi_j + m];
    const int Q = (K + CH - 1) / CH;

    auto sqdist = [&](const double (&d)[DIM]) {
        double d2 = d[0] * d[0];
#pragma unroll
        for (int m = 1; m < DIM; ++m) d2 = fma(d[m], d[m], d2);
        return d2;
    };

    // ---- staging (fit_stage.hip): a load instruction moves the chunks of XCPI (FCPI) whole cases, XPC (FPC) consecutive lanes per case
    const int xsub = lane % XPC, xc0 = lane / XPC, fsub = lane % FPC, fc0 = lane / FPC;
    const unsigned xrowb = (unsigned)K * DIM * 8, frowb = (unsigned)K * 8;
    const bool xlane = lane < XCPI * XPC;
    const char* const xtile = reinterpret_cast<const char*>(p.xk + t0 * (long long)K * DIM);
    const char* const ftile = reinterpret_cast<const char*>(p.fk + t0 * (long long)K);
    // DEEP: TWO chunks in flight (two register sets) — an experiment, OFF.  The idea: a lone wave waits out every exposed load, and with
    // one chunk in flight a sweep of configs[4]'s shape computes ~1.3 us per chunk while 1 024 waves x 18 KB in flight at the measured
    // 7.5 TB/s say ~2.5 us for a chunk to come back from L2 / the Infinity Cache.  Measured (profiles/r04zb_ab_siter_deep.txt, max_iter
    // 10): 3D order 2 at 40 neighbours, 1M cases, 3.51 against 2.03 ms, at 124 neighbours 3.83 against 2.92 — the second set costs 118
    // spilled registers and 13 scratch reloads of load addresses per chunk; 2D order 3 at 80 neighbours (no spill): 1.252 against
    // 1.240 ms, i.e. the latency is NOT what bounds the sweeps there.
    constexpr bool DEEP = (WLSQM_SITER_DEEP != 0) && !RESIDENT && NO > 6 && NO <= 10;      // (the 6-unknown systems run two waves per SIMD)
    d2_ xr[XNI], fr[FNI], xr2[DEEP ? XNI : 1], fr2[DEEP ? FNI : 1];
    auto fetch_into = [&](d2_ (&xr)[XNI], d2_ (&fr)[FNI], int q, auto nof_tag) __attribute__((always_inline)) {
        constexpr bool NOF = decltype(nof_tag)::value;                // (FCACHE, sweeps) the values are in LDS already: coordinates only
        unsigned xo = (unsigned)q * (CH * DIM * 8) + (unsigned)xsub * 16u, fo = (unsigned)q * (CH * 8) + (unsigned)fsub * 16u;
        xo = xo < xrowb ? xo : xrowb - 16u; fo = fo < frowb ? fo : frowb - 16u;      // (rows are multiples of 16 bytes: K even)
        const char* xb = xtile + xo;
        const char* fb = ftile + fo;
#pragma unroll
        for (int i = 0; i < XNI; ++i) {
            int cc = xc0 + i * XCPI;
            cc = cc < nvalid ? cc : nvalid - 1;                       // tail group / idle lanes of the last instruction: replay a valid row
            if (xlane) xr[i] = *reinterpret_cast<const d2_*>(xb + (size_t)(unsigned)cc * xrowb);
        }
        if constexpr (!NOF) {
#pragma unroll
            for (int i = 0; i < FNI; ++i) {
                int cc = fc0 + i * FCPI;
                cc = cc < nvalid ? cc : nvalid - 1;
                fr[i] = *reinterpret_cast<const d2_*>(fb + (size_t)(unsigned)cc * frowb);
            }
        }
    };
    auto park_from = [&](const d2_ (&xr)[XNI], const d2_ (&fr)[FNI], int q, auto nof_tag) __attribute__((always_inline)) {      // RESIDENT: chunk q at its place in the whole row
        constexpr bool NOF = decltype(nof_tag)::value;
        d2_* xl = xs + xc0 * XP2 + xsub + (RESIDENT ? q * (CH * DIM / 2) : 0);
        d2_* fl = fs + fc0 * FP2 + fsub + (RESIDENT ? q * (CH / 2) : 0);
#pragma unroll
        for (int i = 0; i < XNI; ++i)
            if (xlane && (i * XCPI + XCPI <= 64 || xc0 + i * XCPI < 64)) xl[i * XCPI * XP2] = xr[i];
        if constexpr (!NOF) {
#pragma unroll
            for (int i = 0; i < FNI; ++i) fl[i * FCPI * FP2] = fr[i];
        }
    };
    constexpr bool WARM = (WLSQM_SITER_WARM != 0) && !RESIDENT && !DEEP;
    bool inflight = false;                                            // (wave-uniform) chunk Q - 1 of the next pass is on its way in (xr, fr)
    double* const wrow = reinterpret_cast<double*>(lds_d) + lane * FP2r;      // (WCACHE / FCACHE) this lane's weights / values; FP2r: the pitch in doubles (odd)
    const d2_* const xrow = DMA ? lds + (lane / XCPI) * KIB2 + (lane % XCPI) * XPC : xs + lane * XP2;      // (DMA: in slot 0; slot q & 1 at + SLOT2)
    const d2_* const frow = DMA ? lds + XNI * KIB2 + (lane / FCPI) * KIB2 + (lane % FCPI) * FPC : fs + lane * FP2;
    auto dma_fetch = [&](int q, auto nof_tag) __attribute__((always_inline)) {
        constexpr bool NOF = decltype(nof_tag)::value;
        const char* const xt = xtile; const char* const ft = ftile;
        (void)xt; (void)ft;
        if constexpr (DMA) {
            unsigned xo = (unsigned)q * (CH * DIM * 8) + (unsigned)xsub * 16u, fo = (unsigned)q * (CH * 8) + (unsigned)fsub * 16u;
            xo = xo < xrowb ? xo : xrowb - 16u; fo = fo < frowb ? fo : frowb - 16u;
            const unsigned slot = (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)(uintptr_t)lds + (unsigned)(q & 1) * (unsigned)(SLOT2 * 16)));
#pragma unroll
            for (int i = 0; i < XNI; ++i) {
                int cc = xc0 + i * XCPI;
                cc = cc < nvalid ? cc : nvalid - 1;
                unsigned keep;
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                             : "=&s"(keep) : "v"(xo + (unsigned)cc * xrowb), "s"(xt), "s"(slot + (unsigned)i * (unsigned)(KIB2 * 16)) : "memory");
            }
            if constexpr (!NOF) {
#pragma unroll
                for (int i = 0; i < FNI; ++i) {
                    int cc = fc0 + i * FCPI;
                    cc = cc < nvalid ? cc : nvalid - 1;
                    unsigned keep;
                    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                                 : "=&s"(keep) : "v"(fo + (unsigned)cc * frowb), "s"(ft), "s"(slot + (unsigned)(XNI + i) * (unsigned)(KIB2 * 16)) : "memory");
                }
            }
        }
    };
    // wait for a chunk with `younger` (0 / 1) chunks requested behind it (in-order counter: anything else in flight only lengthens the wait)
    auto dma_wait = [&](bool younger, auto nof_tag) __attribute__((always_inline)) {
        constexpr bool NOF = decltype(nof_tag)::value;
        if constexpr (DMA) {
            if (!younger) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NOF ? XNI : XNI + FNI) : "memory");
        }
    };

    // One pass over the neighbours, chunks LAST FIRST (descending k, the order of fit_stage.hip).  body(d, f, live, k) per neighbour,
    // after_group(k0) behind every group of GRP neighbours k0 .. k0 + GRP - 1.
    // Not RESIDENT: the chunks travel through the staging rows.
    // RESIDENT and !fill: everything is in LDS.  RESIDENT and fill: the first pass, which parks the chunks at their places.
    auto for_neighbours = [&](auto masked_tag, auto grp_tag, auto body, auto after_group, const bool fill, auto nof_tag) __attribute__((always_inline)) {
        constexpr bool MASKED = decltype(masked_tag)::value;
        constexpr bool NOF = decltype(nof_tag)::value;                // (FCACHE) the values come from the lane's cache row
        constexpr int GRP = decltype(grp_tag)::value;                 // neighbours of a lane in flight
        const bool staged = !RESIDENT || fill;
        auto compute = [&](const int q) __attribute__((always_inline)) {
            const d2_* xq = xrow + (RESIDENT ? q * (CH * DIM / 2) : DMA ? (q & 1) * SLOT2 : 0);
            const d2_* fq = frow + (RESIDENT ? q * (CH / 2) : DMA ? (q & 1) * SLOT2 : 0);
#pragma unroll
            for (int g = CH / GRP - 1; g >= 0; --g) {
                double xv[GRP * DIM], fv[GRP];
                static_assert(GRP % 2 == 0 || (GRP == 1 && DIM == 2), "whole 16-byte pieces per group");
#pragma unroll
                for (int i = 0; i < GRP * DIM / 2; ++i) { const d2_ v = xq[g * (GRP * DIM / 2) + i]; xv[2 * i] = v.x; xv[2 * i + 1] = v.y; }
                if constexpr (NOF) {
#pragma unroll
                    for (int i = 0; i < GRP; ++i) fv[i] = wrow[q * CH + g * GRP + i];
                } else {
                    if constexpr (GRP == 1) { const d2_ v = fq[g / 2]; fv[0] = (g & 1) ? v.y : v.x; }
#pragma unroll
                    for (int i = 0; i < GRP / 2; ++i) { const d2_ v = fq[g * (GRP / 2) + i]; fv[2 * i] = v.x; fv[2 * i + 1] = v.y; }
                }
#pragma unroll
                for (int kk = GRP - 1; kk >= 0; --kk) {
                    const int ks = g * GRP + kk;
                    const bool live = MASKED ? (q * CH + ks < nkc) : true;
                    double d[DIM];
#pragma unroll
                    for (int m = 0; m < DIM; ++m) { d[m] = xv[kk * DIM + m] - xi[m]; if (MASKED) d[m] = live ? d[m] : 0.0; }
                    body(d, MASKED ? (live ? fv[kk] : 0.0) : fv[kk], live, q * CH + ks);
                }
                __builtin_amdgcn_sched_barrier(0);                    // GRP neighbours in flight at a time
                after_group(q * CH + g * GRP);
            }
        };
        // chunk q arrives in a register set, is parked and consumed; the set is refilled with the chunk `ahead` further on — behind
        // the pass's last chunk with the FIRST chunk of the next pass (WARM; an experiment, OFF).  The idea: a lone wave has nothing to
        // cover the start of a pass with, and there are max_iter + 1 passes.  Measured (profiles/r04zb_ab_siter_warm.txt, max_iter 10):
        // 2D order 4 at 64 neighbours 1.564 against 1.386 ms per 400k, 2D order 3 at 80 1.295 against 1.239, 3D order 2 at 40 2.070
        // against 2.014 per 1M — the set's 48 registers stay live across the substitution between two sweeps (2D order 4: 122 instead
        // of 88 spilled), which costs more than the exposed start
        auto step = [&](d2_ (&xa)[XNI], d2_ (&fa)[FNI], const int q, const int ahead) __attribute__((always_inline)) {
            if (staged) {
                if (!RESIDENT) __syncthreads();                       // the previous chunk has been read by every lane
                park_from(xa, fa, q, nof_tag);
                __syncthreads();
                if (q - ahead >= 0) fetch_into(xa, fa, q - ahead, nof_tag);
                else if (WARM && q == 0) { fetch_into(xa, fa, Q - 1, std::false_type{}); inflight = true; }
            }
            compute(q);
        };
        if constexpr (DMA) {
            dma_fetch(Q - 1, nof_tag);
            for (int q = Q - 1; q >= 0; --q) {
                if (q >= 1) dma_fetch(q - 1, nof_tag);                // into the slot of chunk q + 1: consumed
                dma_wait(q >= 1, nof_tag);
                compute(q);
            }
        } else if constexpr (DEEP) {
            fetch_into(xr, fr, Q - 1, nof_tag);
            if (Q > 1) fetch_into(xr2, fr2, Q - 2, nof_tag);
            for (int q = Q - 1; q >= 0; q -= 2) {
                step(xr, fr, q, 2);
                if (q >= 1) step(xr2, fr2, q - 1, 2);
            }
        } else {
            if (staged && !(WARM && inflight)) fetch_into(xr, fr, Q - 1, nof_tag);
            for (int q = Q - 1; q >= 0; --q) step(xr, fr, q, 1);
        }
    };
    const bool full = (K % CH == 0) && __all(nkc >= K);               // wave-uniform: no ragged case in this group, whole chunks
    auto no_hook = [](int) __attribute__((always_inline)) {};
    auto pass = [&](auto body, const bool fill, auto nof_tag) __attribute__((always_inline)) {
        if (full) for_neighbours(std::false_type{}, std::integral_constant<int, GRP>{}, body, no_hook, fill, nof_tag);
        else for_neighbours(std::true_type{}, std::integral_constant<int, GRP>{}, body, no_hook, fill, nof_tag);
    };

    // ---- the fit: largest squared distance, moments
    double mu[NM], nu[NO];
    double max_d2 = 0.0, inv_max = 0.0;
    auto max_body = [&](const double (&d)[DIM], double, bool, int) __attribute__((always_inline)) {
        const double d2 = sqdist(d);
        max_d2 = d2 > max_d2 ? d2 : max_d2;                           // (a masked slot contributes 0)
    };
    auto mom_body = [&](const double (&d)[DIM], double f, bool live, int k) __attribute__((always_inline)) {
        const double d2 = sqdist(d);
        max_d2 = d2 > max_d2 ? d2 : max_d2;
        double w = weight(d2, inv_max, uniform);
        w = live ? w : 0.0;
        if constexpr (WCACHE) wrow[k] = w;                            // (a repeated pass overwrites them with the final ones)
        if constexpr (FCACHE) wrow[k] = f;                            // (masked: 0 beyond the case's neighbours)
        accumulate_moments_best<DIM, ORDER>(mu, nu, d, w, f);
    };
    auto moments = [&](const double maxv) __attribute__((always_inline)) {
#pragma unroll
        for (int e = 0; e < NM; ++e) mu[e] = 0.0;
#pragma unroll
        for (int a = 0; a < NO; ++a) nu[a] = 0.0;
        max_d2 = 0.0;
        inv_max = inverse_max(maxv);
        pass(mom_body, false, std::false_type{});
    };
    if constexpr (RESIDENT) {
        pass(max_body, true, std::false_type{});
        moments(max_d2);
    } else {
        // speculative (fit_stage.hip): the last neighbour is the farthest for sorted neighbour lists; verified bit for bit
        double guess = 0.0;
        if (nkc > 0) {
            const double* q = p.xk + j * (long long)K * DIM + (long long)(nkc - 1) * DIM;
            double dg[DIM];
#pragma unroll
            for (int m = 0; m < DIM; ++m) dg[m] = q[m] - xi[m];
            guess = sqdist(dg);
        }
        moments(guess);
        if (!__all(uniform || max_d2 == guess)) moments(max_d2);
    }

    // ---- solve: masked full system, unpivoted LDL^T; the factor stays
    constexpr unsigned long long FULL = (1ull << NO) - 1ull;
    double* const fio = p.fi + j * p.sfi_j;
    double M[NE], fi[NO];
    // A case with exactly the function value known (knowns = b?_F: the reference's default mask and BASELINE configs[2]) keeps the factor of
    // its 14 x 14 system (fit_stage.hip: 105 + 14 entries to expand, factor and substitute instead of 120 + 15), chosen by the case's
    // own mask — a mixed wave runs both forms, a case's bits do not depend on its wave-mates.  M then holds the 105 entries of that factor.
    constexpr bool REDUCED = (WLSQM_SITER_REDUCED != 0) && NO == 15 && !SENS;
    constexpr int N1 = NO - 1, NE1 = N1 * (N1 + 1) / 2;
    const bool mine1 = REDUCED && known == 1ull && dropped == 0ull;
    if constexpr (REDUCED) {
        if (mine1) {
            const double v0 = fio[0];
            double M1[NE1], r1[N1];
#pragma unroll
            for (int a = 1; a < NO; ++a) {
                const int pa = Mono<DIM>::P[a], qa = Mono<DIM>::Q[a], ra = Mono<DIM>::R[a];
                r1[a - 1] = nu[mom_index<DIM>(pa, qa, ra)] * (mom_inv_fact(pa) * mom_inv_fact(qa) * mom_inv_fact(ra));
            }
#pragma unroll
            for (int i = 0; i < NM; ++i) {
                const double m = mu[i];
#pragma unroll
                for (int a = 1; a < NO; ++a) {
                    const int pa = Mono<DIM>::P[a], qa = Mono<DIM>::Q[a], ra = Mono<DIM>::R[a];
                    const double fa = mom_inv_fact(pa) * mom_inv_fact(qa) * mom_inv_fact(ra);
                    if (mom_index<DIM>(pa, qa, ra) == i) r1[a - 1] = fma(-(m * (1.0 * fa)), v0, r1[a - 1]);      // M[0, a] * fi[0] (impl.pyx:815-818)
#pragma unroll
                    for (int b = a; b < NO; ++b) {
                        const int pb = Mono<DIM>::P[b], qb = Mono<DIM>::Q[b], rb = Mono<DIM>::R[b];
                        const double fb = mom_inv_fact(pb) * mom_inv_fact(qb) * mom_inv_fact(rb);
                        if (mom_index<DIM>(pa + pb, qa + qb, ra + rb) == i) M1[tri<N1>(a - 1, b - 1)] = m * (fa * fb);
                    }
                }
            }
            ldlt_factor<N1>(M1);
            ldlt_solve<N1>(M1, r1);
            fi[0] = v0;
#pragma unroll
            for (int a = 1; a < NO; ++a) fi[a] = r1[a - 1];
#pragma unroll
            for (int e = 0; e < NE; ++e) M[e] = e < NE1 ? M1[e < NE1 ? e : 0] : 0.0;
        }
    }
    if (!mine1) {
        double g[NO], val[NO];
        expand_moments<DIM, ORDER>(mu, nu, M, g);
#pragma unroll
        for (int a = 0; a < NO; ++a) val[a] = (((known & ~dropped) >> a) & 1ull) ? fio[a] : 0.0;
        eliminate_knowns<NO>(M, g, known, val);
        ldlt_factor<NO>(M);
        ldlt_solve<NO>(M, g);
#pragma unroll
        for (int a = 0; a < NO; ++a) fi[a] = ((known >> a) & 1ull) ? (((dropped >> a) & 1ull) ? fio[a] : val[a]) : g[a];
    }

    // ---- sensitivities (impl.pyx:776-778, 821-823, 831-846): sens[k, a] = d fi[a] / d fk[k] for k < nk; NaN in the columns of the
    // true knowns; a dropped DOF's column, the rows from nk on and the cases with every DOF known are not written (fit_lane.hip)
    if constexpr (SENS) {
        static_assert(SR <= 16 && (SG * NO) % 2 == 0 && CH % SG == 0, "a case's tile is at most 16 pieces of 16 bytes");
        const int lim = (valid && known != FULL) ? nkc : 0;           // rows of this lane's case that are written
        // Cooperative stores (wave-uniform): rows of NO contiguous doubles, 16-byte aligned case blocks, no dropped DOF in the wave.
        // The SG x NO results of a lane go to its row of the LDS tile; then SIXTEEN lanes store one case's SG NO 8 contiguous bytes,
        // four cases per instruction.  Otherwise every lane stores its own 8-byte results (any strides).
        const bool coop = p.ss_k == NO && (p.ss_j % 2) == 0 && ((reinterpret_cast<uintptr_t>(p.sens) & 15u) == 0) && __all(dropped == 0ull);
        const double qnan = __longlong_as_double(0x7ff8000000000000LL);
        double* const trow = reinterpret_cast<double*>(tile2 + lane * TP2);
        double* const srow = p.sens + j * p.ss_j;
        auto sens_body = [&](const double (&d)[DIM], double, bool, int k) __attribute__((always_inline)) {
            double cc[NO], sv[NO];
            monomials<DIM, ORDER>(d, cc);
            const double w = weight(sqdist(d), inv_max, uniform);     // (the moment pass's weight, bit for bit)
#pragma unroll
            for (int a = 0; a < NO; ++a) sv[a] = ((known >> a) & 1ull) ? 0.0 : ((a == 0) ? w : w * cc[a]);
            ldlt_solve<NO>(M, sv);
            if (coop) {
#pragma unroll
                for (int a = 0; a < NO; ++a) trow[(k % SG) * NO + a] = ((known >> a) & 1ull) ? qnan : sv[a];
            } else if (k < lim) {
#pragma unroll
                for (int a = 0; a < NO; ++a) {
                    if (!((known >> a) & 1ull)) srow[k * p.ss_k + a] = sv[a];
                    else if (!((dropped >> a) & 1ull)) srow[k * p.ss_k + a] = qnan;
                }
            }
        };
        const int sc = lane / 16, e2 = lane % 16;                     // this lane's case of an instruction's four, its piece of the case's tile
        const int r0 = (2 * e2) / NO, r1 = (2 * e2 + 1) / NO;         // tile rows of the piece's two doubles
        const unsigned voff = (unsigned)sc * (unsigned)(p.ss_j * 8) + (unsigned)e2 * 16u;
        // (the 15-unknown systems take one neighbour at a time: two of them — 60 registers beside the 240 of the factor — spilled 350)
        constexpr int SGRP = NO >= 15 ? 1 : SG;
        auto flush = [&](const int k0) __attribute__((always_inline)) {
            if (!coop || (k0 % SG) != 0) return;
            __syncthreads();                                          // the tile is written
            const bool allrows = __all(lim >= k0 + SG || !valid);
            char* const base = reinterpret_cast<char*>(p.sens + t0 * p.ss_j + (long long)k0 * NO);
#pragma unroll 4
            for (int i = 0; i < 16; ++i) {
                const int c = 4 * i + sc;
                const d2_ v = tile2[c * TP2 + e2];
                char* const dst = base + (size_t)i * 4u * (size_t)(p.ss_j * 8) + voff;
                bool ok0 = e2 < SR && c < nvalid, ok1 = ok0;
                if (!allrows) {
                    const int n_ok = __shfl(lim, c, 64) - k0;         // rows of this tile that case c writes
                    ok0 = ok0 && r0 < n_ok; ok1 = ok1 && r1 < n_ok;
                }
                // (plain stores instead measured mixed: 2D order 2 / 3 / 4, 3D order 2: 0.355 / 0.571 / 3.13 / 0.694 against 0.305 / 0.654 / 2.72 / 0.842 ms)
                if (ok1) __builtin_nontemporal_store(v, reinterpret_cast<d2_*>(dst));
                else if (ok0) __builtin_nontemporal_store(v.x, reinterpret_cast<double*>(dst));
            }
            __syncthreads();                                          // the tile is read
        };
        if (p.sens) {
            if (full) for_neighbours(std::false_type{}, std::integral_constant<int, SGRP>{}, sens_body, flush, false, std::false_type{});
            else for_neighbours(std::true_type{}, std::integral_constant<int, SGRP>{}, sens_body, flush, false, std::false_type{});
        }
    }

    // ---- sweeps (impl.pyx:1016-1081)
    bool done = !(valid && known != FULL);
    bool broke = false;
    int it_case = 0;
    double prev_norm = -1.0;
    for (int it = 0; it < (p.iterative ? p.max_iter : 0); ++it) {
        if (__ballot(!done) == 0ull) break;                           // every case of the wave has stopped
        double norm = 0.0, r[NO];
#pragma unroll
        for (int a = 0; a < NO; ++a) r[a] = 0.0;
        auto sweep_body = [&](const double (&d)[DIM], double f, bool live, int k) __attribute__((always_inline)) {
            double cc[NO];
            monomials<DIM, ORDER>(d, cc);
            // the weight of the moment pass, bit for bit (the reference computes its weights once, infra.pyx:668-702): cached, or
            // recomputed from the same rounding sequence of the squared distance
            double w;
            if constexpr (WCACHE) w = wrow[k];
            else { w = weight(sqdist(d), inv_max, uniform); w = live ? w : 0.0; }
#if WLSQM_SITER_MODEL_CHAINS > 1
            // taylor_*D (polyeval.pyx): sum_a c[a] fi[a], as WLSQM_SITER_MODEL_CHAINS interleaved partial sums (shorter dependent chains)
            double part[WLSQM_SITER_MODEL_CHAINS];
#pragma unroll
            for (int i = 0; i < WLSQM_SITER_MODEL_CHAINS; ++i) part[i] = i == 0 ? fi[0] : 0.0;
#pragma unroll
            for (int a = 1; a < NO; ++a) part[a % WLSQM_SITER_MODEL_CHAINS] = fma(cc[a], fi[a], part[a % WLSQM_SITER_MODEL_CHAINS]);
            double model = part[0];
#pragma unroll
            for (int i = 1; i < WLSQM_SITER_MODEL_CHAINS; ++i) model += part[i];
#else
            double model = fi[0];                                     // taylor_*D (polyeval.pyx): sum_a c[a] fi[a]
#pragma unroll
            for (int a = 1; a < NO; ++a) model = fma(cc[a], fi[a], model);
#endif
            const double res = live ? f - model : 0.0;
            const double ar = fabs(res);
            norm = ar > norm ? ar : norm;                             // impl.pyx:1037-1041
            const double wr = w * res;
#pragma unroll
            for (int a = 0; a < NO; ++a) r[a] = fma(wr, (a == 0) ? 1.0 : cc[a], r[a]);
        };
        pass(sweep_body, false, std::bool_constant<FCACHE>{});
        if (!done) {
            if (norm == prev_norm) { broke = true; done = true; it_case = it; }      // impl.pyx:1057
            else {
                prev_norm = norm;
                if constexpr (REDUCED) {
                    if (mine1) {                                      // the 14 x 14 factor: the correction of the known function value is 0
                        double M1[NE1], r1[N1];
#pragma unroll
                        for (int e = 0; e < NE1; ++e) M1[e] = M[e];
#pragma unroll
                        for (int a = 1; a < NO; ++a) r1[a - 1] = r[a];
                        ldlt_solve<N1>(M1, r1);
#pragma unroll
                        for (int a = 1; a < NO; ++a) fi[a] += r1[a - 1];
                    }
                }
                if (!mine1) {
#pragma unroll
                    for (int a = 0; a < NO; ++a) if ((known >> a) & 1ull) r[a] = 0.0;    // knowns of the correction are 0
                    ldlt_solve<NO>(M, r);
#pragma unroll
                    for (int a = 0; a < NO; ++a) if (!((known >> a) & 1ull)) fi[a] += r[a];
                }
            }
        }
    }
    if (p.iterative && p.iters_out) {
        int iters = (valid && known != FULL) ? (broke ? it_case : (p.max_iter > 0 ? p.max_iter : 1)) : 0;      // for / else, impl.pyx:1080-1081
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) { const int o = __shfl_xor(iters, off, 64); iters = o > iters ? o : iters; }
        if (lane == 0 && iters > 0) atomicMax(p.iters_out, iters);
    }

    // ---- results (fit_stage.hip): a full group with contiguous fi rows leaves as ONE run of 64 NO doubles through LDS
    const bool whole = nvalid == 64 && p.sfi_j == NO && ((reinterpret_cast<uintptr_t>(p.fi) & 15u) == 0) && (64 * NO) % 2 == 0 &&
                       __all(dropped == 0ull && known != FULL);
    if (whole) {
        __syncthreads();                                              // the last pass has been read
        double* const o = reinterpret_cast<double*>(lds);
#pragma unroll
        for (int a = 0; a < NO; ++a) o[lane * NO + a] = fi[a];        // (known DOFs: their own bits, Case_get_fi infra.pyx:780-795)
        __syncthreads();
        d2_* out = reinterpret_cast<d2_*>(p.fi + t0 * NO);
#pragma unroll
        for (int q = lane; q < 64 * NO / 2; q += 64) __builtin_nontemporal_store(lds[q], &out[q]);
    } else if (valid && known != FULL) {
#pragma unroll
        for (int a = 0; a < NO; ++a)
            if (!((known >> a) & 1ull)) fio[a] = fi[a];
    }
}

template <int DIM, int ORDER>
static int launch_stage_refine(const KParams& p, long long K, hipStream_t stream) {
    using namespace siter;
    const long long groups = (p.ncases + 63) / 64;
    if (groups <= 0) return WLSQM_OK;
    if (groups > 0x7fffffffLL) { set_error("fit_stage_refine: batch too large for one launch"); return WLSQM_EVALUE; }
    if (p.do_sens) {
        // (the rows are re-staged per pass: with the sens tile beside them whole rows would leave two waves per CU)
        hipLaunchKernelGGL((fit_stage_refine_kernel<DIM, ORDER, false, true>), dim3((unsigned)groups), dim3(64), 0, stream, p, 0, 0);
        WLSQM_HIP_CHECK(hipGetLastError());
        note_kernel(p.iterative ? "stage-sens-refine" : "stage-sens");
        return WLSQM_OK;
    }
    // whole rows in LDS when three waves per CU still fit (WLSQM_HIP_REFINE_RESIDENT_KB overrides the bound; 0: never)
    const int Q = (int)((K + CH - 1) / CH);
    const int XP2 = pitch2(Q * CH * DIM), FP2 = pitch2(Q * CH);
    const size_t bytes = (size_t)64 * (XP2 + FP2) * 16;
    size_t bound = 53 * 1024;
    if (const char* e = getenv("WLSQM_HIP_REFINE_RESIDENT_KB")) bound = (size_t)atol(e) * 1024;
    if (bound > 160 * 1024) bound = 160 * 1024;
    // (from three sweeps on: below that the re-staging form's four to five waves per CU win — 2D order 4 at 26 neighbours, max_iter 0 / 1 /
    // 2 / 4 / 10: 0.132 / 0.216 / 0.297 / 0.433 / 0.813 against 0.158 / 0.218 / 0.276 / 0.394 / 0.738 ms resident, 3D order 2 at 20: 0.088 /
    // 0.128 / 0.174 / 0.277 / 0.508 against 0.115 / 0.148 / 0.187 / 0.255 / 0.460, profiles/r04zb_ab_siter_resident.txt; an explicit
    // WLSQM_HIP_REFINE_RESIDENT_KB applies to every max_iter)
    // 2D order 3 never (with the weight cache the re-staging form wins at every sweep count: 24 / 30 / 32 neighbours, max_iter 10: 0.371 /
    // 0.554 / 0.523 against 0.377 / 0.638 / 0.558 ms resident; 2D order 4 and 3D order 2 the other way round: 26 neighbours 0.81 against
    // 0.73, 20 neighbours 0.51 against 0.46)
    // (rows of at most 40 KB leave four waves per CU as well: resident always — 2D order 3 at 16 neighbours 0.234 against 0.258 ms)
    const bool few = !getenv("WLSQM_HIP_REFINE_RESIDENT_KB") && bytes > 40 * 1024 && (p.max_iter < 3 || (DIM == 2 && ORDER == 3));
    if (!few && bytes <= bound && bytes >= (size_t)64 * ndofs(DIM, ORDER) * 8) {
        auto kern = fit_stage_refine_kernel<DIM, ORDER, true, false>;
        static std::atomic<unsigned> optin{0};                        // per device, once: more than 64 KB of dynamic LDS
        int dev = 0;
        WLSQM_HIP_CHECK(hipGetDevice(&dev));
        if (bytes > 64 * 1024 && dev >= 0 && dev < 32 && !((optin.load() >> dev) & 1u)) {
            WLSQM_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            optin.fetch_or(1u << dev);
        }
        hipLaunchKernelGGL(kern, dim3((unsigned)groups), dim3(64), bytes, stream, p, XP2, FP2);
        WLSQM_HIP_CHECK(hipGetLastError());
        note_kernel("stage-refine-resident");
        return WLSQM_OK;
    }
    {
        // the weights beside the staging rows while four waves per CU still fit (WLSQM_HIP_REFINE_WCACHE=0: never)
        const char* e = getenv("WLSQM_HIP_REFINE_WCACHE");
        const int WP = (Q * CH) | 1;
        // (static LDS of the kernel: the padded staging rows, or the DMA form's two slots of a KiB per load instruction — 24 KB for the
        // 15-unknown systems, whose weight cache therefore ends at 30 neighbours instead of 48)
        constexpr bool dma15 = (WLSQM_SITER_DMA != 0) && ndofs(DIM, ORDER) == 15;
        constexpr int xni = 64 / (64 / (CH * DIM * 8 / 16)) + ((64 % (64 / (CH * DIM * 8 / 16))) ? 1 : 0), fni = 64 / (64 / (CH * 8 / 16));
        const size_t wbytes = (size_t)64 * WP * 8, stat = dma15 ? (size_t)2 * (xni + fni) * (1024 + 16 * WLSQM_SITER_DMA_SKEW) : (size_t)64 * (pitch2(CH * DIM) + pitch2(CH)) * 16;
        size_t budget = 40 * 1024;                                   // four waves per CU (WLSQM_HIP_REFINE_CACHE_KB: A/B)
        if (const char* b = getenv("WLSQM_HIP_REFINE_CACHE_KB")) budget = (size_t)atol(b) * 1024;
        if (!(e && e[0] == '0') && p.max_iter >= 1 && ndofs(DIM, ORDER) > 6 && wbytes + stat <= budget && wbytes + stat <= 64 * 1024) {        // (no opt-in to more than 64 KB of LDS for this kernel)
            // 3D: the values (the sweeps are bound by the re-staged bytes); 2D: the weights (by their instructions); =w / =f force one
            constexpr int DEF = DIM == 3 ? 2 : 1;
            const int which = (e && e[0] == 'w') ? 1 : (e && e[0] == 'f') ? 2 : DEF;
            if (which == 2) hipLaunchKernelGGL((fit_stage_refine_kernel<DIM, ORDER, false, false, 2>), dim3((unsigned)groups), dim3(64), wbytes, stream, p, 0, WP);
            else
            hipLaunchKernelGGL((fit_stage_refine_kernel<DIM, ORDER, false, false, 1>), dim3((unsigned)groups), dim3(64), wbytes, stream, p, 0, WP);
            WLSQM_HIP_CHECK(hipGetLastError());
            note_kernel("stage-refine");
            return WLSQM_OK;
        }
    }
    hipLaunchKernelGGL((fit_stage_refine_kernel<DIM, ORDER, false, false>), dim3((unsigned)groups), dim3(64), 0, stream, p, 0, 0);
    WLSQM_HIP_CHECK(hipGetLastError());
    note_kernel("stage-refine");
    return WLSQM_OK;
}

// Fits with refinement of dense contiguous batches with an even neighbour count (2D orders 2-4, 3D order 2); everything else keeps its
// kernels.  WLSQM_HIP_STAGE_REFINE=0 disables it, =all sends every covered shape here (A/B).
//
// The SENS form (sensitivities, with or without refinement) is built, tested (tests/test_gpu_round4.py) and OFF: it only runs with
// WLSQM_HIP_STAGE_SENS=all.  Measured (tools/time_sens.py, 400k cases, profiles/r04zb_time_sens.txt) against tile1-extras / the
// inverse + MFMA path: 2D order 2 at 32 neighbours 0.305 against 0.249 ms, 2D order 3 at 30 0.654 against 0.519, 2D order 4 at 64
// 2.72 against 1.43, 3D order 2 at 40 0.842 against 0.658; with refinement in the same launch (200k cases) 0.43 / 0.63 / 2.23 / 0.86
// against 0.24 / 0.50 / 2.26 (lane) / 0.62.  One substitution per neighbour and lane is NO^2 dependent multiply-adds through a factor
// that lives half in the accumulation registers (2D order 4: 37 of its 120 entries in scratch as well), and a lane's results must
// cross the wave through LDS before they can leave as contiguous runs; the matrix cores do the same NO x NO by NO x K product on
// the inverse at 2 147 GB/s of sens rows.
//
// Refinement alone (tools/time_refine.py, 400k cases, max_iter 10, against tile1-extras / chunk-refine / refine-apply;
// profiles/r04zb_time_refine.txt): 2D order 4 at 26 / 40 / 64 / 100 neighbours 0.78 / 1.01 / 1.40 / 2.17 against 1.56 / 2.11 / 2.15 /
// 4.01 ms, 3D order 2 at 20 / 40 / 124 0.51 / 0.88 / 2.92 against 0.56 / 0.94 / 5.00, 2D order 3 at 80 1.24 against 1.61 — but 2D
// order 3 at 30 0.62 against 0.54 and 2D order 2 at 16 / 32 / 64 0.19 / 0.43 / 1.08 against 0.17 / 0.31 / 0.75: a 64-case wave sweeps
// until its LAST case stops and re-reads its rows per sweep where the 16-case tiles of those kernels keep them in LDS at two waves per
// SIMD.  The fit in front of the sweeps is 2-3x faster here, so the small systems come here for few sweeps only (the crossover:
// max_iter 2 for 2D order 2 — later by neighbour count —, the figures above are from before the LDS caches: with the weight
// cache 2D order 3 wins at every neighbour and sweep count, profiles/r04zb_time_refine_23.txt).
int launch_fit_stage_refine(int dimension, int order, const KParams& p, long long K, hipStream_t stream, bool* handled) {
    *handled = false;
    const char* off = getenv("WLSQM_HIP_DISABLE_TILE");
    if (off && off[0] == '1') return WLSQM_OK;
    if (!(p.iterative || p.do_sens) || p.case_index || p.hoods || p.it_stop != 0) return WLSQM_OK;
    { const char* r = getenv("WLSQM_HIP_REFINE_ROUNDS"); if (r && r[0] == '1') return WLSQM_OK; }      // (the rounds experiment of fit_tilek.hip keeps its shapes)
    if (p.do_sens && !p.sens) return WLSQM_OK;
    const char* e = getenv(p.do_sens ? "WLSQM_HIP_STAGE_SENS" : "WLSQM_HIP_STAGE_REFINE");
    if (e && e[0] == '0') return WLSQM_OK;
    const bool all = e && e[0] == 'a';
    if (p.do_sens && !all) return WLSQM_OK;                           // (measured slower than the kernels these calls have: see above)
    if (!p.xk || !p.fk || !p.xi) return WLSQM_OK;
    if (K < 8 || K % 2 != 0 || K > 65536) return WLSQM_OK;          // (row bytes and the tile's offsets are 32-bit)
    if (p.sxk_k != dimension || p.sxk_j != K * dimension || p.sfk_k != 1 || p.sfk_j != K) return WLSQM_OK;
    if ((reinterpret_cast<uintptr_t>(p.xk) | reinterpret_cast<uintptr_t>(p.fk)) & 15u) return WLSQM_OK;
    if (p.do_sens && (p.ss_j * 8 * 4 > 0x7fffffffLL)) return WLSQM_OK;            // (32-bit offsets inside a store instruction's four cases)
#define RCASE(D, O, COND) if (dimension == D && order == O && (all || (COND))) { *handled = true; return launch_stage_refine<D, O>(p, K, stream); }
    RCASE(2, 2, p.max_iter <= 2 || (K >= 48 && p.max_iter <= 5) || (K > 64 && p.max_iter <= 8))     // (64 neighbours, max_iter 4: 0.504 against 0.596 ms; 160, max_iter 4: 1.31 against 2.08, 10: 2.71 against 2.58)
    RCASE(2, 3, true)                                               // (16 / 20 / 24 / 30 / 32 / 36 neighbours, max_iter 10: 0.234 / 0.386 / 0.385 / 0.539 / 0.523 / 0.639 against 0.336 / 0.451 / 0.480 / 0.542 / 0.543 / 0.876 ms on tile1-extras)
    RCASE(2, 4, true)
    RCASE(3, 2, true)
#undef RCASE
    return WLSQM_OK;
}

}  // namespace wlsqm
