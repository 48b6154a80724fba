// fit_stage.hip — ONE LANE PER CASE fast fit on dense contiguous rows (round 4): the tile / ring kernels' arithmetic (moment form,
// FMA, rsq-seeded weights, unpivoted LDL^T on the masked full system: wlsqm_kernels.hpp / wlsqm_moments.hpp) without their
// cross-lane machinery.
//
// Reference path (file:line in /root/reference): make_c_nD impl.pyx:286-432 / 70-269, Case_make_weights infra.pyx:668-702, make_A
// impl.pyx:566-602, solve with knowns elimination impl.pyx:731-846, dgetrf / dgetrs lapackdrivers.pyx:1628-1665.
//
// Why.  The ring kernels (wlsqm_ring.hpp) give a case to FOUR lanes: 16 (10) neighbours each, then a reduce-scatter of the partial
// moments over the four lanes, the quarters of four tiles parked in (accumulation) registers, a 4 x 4 transposition so that the wave
// can solve 64 different cases at once.  Counted from the ISA (profiles/isa_r03.txt): BASELINE configs[2] executes
// 2 584 vector instructions per 16-case tile = 10 336 per 64 cases, of which the reduce-scatter, the parking and the transposition
// (388 of its 690 instructions are v_accvgpr moves) and the per-tile prologues are about a fifth — work that exists only because a
// case is spread over lanes.  With one lane per case nothing is reduced, parked or transposed: 64 neighbours x (81 moment operations +
// the weight) + one solve per lane.  What kept that mapping slow before (fit_lane_kernel: 8.5 % of the HBM peak) was its memory side —
// every lane reading its own row, 64 cache lines per load instruction — and that is what the accurate mode's staging (fit_accurate.hip)
// solved: the rows of a wave's 64 consecutive cases travel through LDS in chunks of 8 neighbours, global loads are coalesced 16-byte
// pieces of whole 128- / 192-byte runs, the next chunk is in flight in registers while the current one is consumed.
//
// The weights need the largest squared distance before the first moment can be summed (a second pass over the rows, for which there
// is no room in LDS at 64 cases per wave).  As in the accurate mode the pass is SPECULATIVE: neighbour lists out of a k-nearest-
// neighbour search are sorted by distance (scipy's cKDTree.query, wlsqm.hip.knn: the reference's examples and every BASELINE config),
// so it runs with the last neighbour's squared distance as the maximum while it tracks the true one, and the guess is verified bit for
// bit; a wave with a wrong guess (unsorted neighbours) repeats the pass with the true maxima — same bits either way.
//
// Round 5: the chunks of the 7..10- and 35-unknown systems go from memory straight into LDS (global_load_lds_dwordx4, `DMA` in the
// kernel: no vector register holds a chunk in flight).  That took the 10-unknown kernels from 330-408 to 184-186 registers: they run TWO
// waves per SIMD on one 17 KB slot each (configs[4] 0.290 -> 0.275 ms), the transfers with a split cache policy (whole-line pieces
// non-temporal, pieces in a line shared with the next chunk default); 3D order 4's moment halves lost their scratch.  What was measured
// on the way, and what did not pay: DESIGN section 4.1 (round-5 part), profiles/r05g_ab_stage_dma.txt.
//
// One sum per moment over k (descending: the chunks are staged last-first, see the kernel) in ONE lane: of the fast kernels this one is
// the closest to the reference's summation
// (profiles/r03_attribution.txt: the lane-split sums are the largest single contribution to the fast kernels' distance from it).
#include <atomic>
#include <type_traits>

#include "wlsqm_internal.hpp"
#include "wlsqm_kernels.hpp"
#include "wlsqm_moments.hpp"

#ifndef WLSQM_STAGE_NT_LOADS
#define WLSQM_STAGE_NT_LOADS 0      // 1: the once-read rows by non-temporal loads — slower: configs[1] 0.175 against 0.152 ms, configs[2] 0.407 against 0.398, configs[4] 0.317 against 0.286 (profiles/r04zb_ab_stage_nt_loads.txt)
#endif
#ifndef WLSQM_STAGE_GRP
#define WLSQM_STAGE_GRP 4           // neighbours the scheduler may interleave in the moment pass (with WLSQM_STAGE_SCHED_BARRIER)
#endif
#ifndef WLSQM_STAGE_SCHED_BARRIER
#define WLSQM_STAGE_SCHED_BARRIER 0
#endif

namespace wlsqm {

namespace stage {
typedef double d2_ __attribute__((ext_vector_type(2)));
constexpr int CH = 8;               // neighbours per staged chunk
}

#ifndef WLSQM_STAGE_GRP20
#define WLSQM_STAGE_GRP20 8         // ... of the 20-unknown systems (1 / 2 / 4 / 8: 0.608 / 0.594 / 0.594 / 0.569 ms per 1M cases of 32 neighbours, profiles/r04t_ab_stage33.txt)
#endif
#ifndef WLSQM_STAGE_SOLVE_BARRIERS
#define WLSQM_STAGE_SOLVE_BARRIERS 1
#endif
#ifndef WLSQM_STAGE_LDS_ROWS
#define WLSQM_STAGE_LDS_ROWS 4      // rows of a 20 x 20 normal matrix kept in LDS during the solve (74 of its 210 entries: 37 KB per wave)
#endif
#ifndef WLSQM_STAGE_MINW6
#define WLSQM_STAGE_MINW6 2         // waves per SIMD the systems up to 6 unknowns are compiled for
#endif
#ifndef WLSQM_STAGE_MINW6G
#define WLSQM_STAGE_MINW6G 2        // ... their gathering form
#endif
#ifndef WLSQM_STAGE_MINW35
#define WLSQM_STAGE_MINW35 1        // ... the two moment passes of the 35-unknown systems (dense input; two: ~100 sums per lane leave no room, 1 898 / 49 spilled registers)
#endif
#ifndef WLSQM_STAGE_MINW10
#define WLSQM_STAGE_MINW10 2        // waves per SIMD the systems with 7..10 unknowns are compiled for (dense input; the gathering form keeps its SIMD)
#endif
#ifndef WLSQM_STAGE_MINW10G
#define WLSQM_STAGE_MINW10G 1       // ... their gathering form (two: 3D order 2 spills)
#endif
#ifndef WLSQM_STAGE_MINW15
#define WLSQM_STAGE_MINW15 1        // ... the 15-unknown systems
#endif
#ifndef WLSQM_STAGE_GRP15
#define WLSQM_STAGE_GRP15 8         // neighbours the scheduler may interleave in their moment pass
#endif
// PART = 4 (round 5, 2D order 4): the whole fit, and the INVERSE of every case's knowns-eliminated normal matrix at p.ws for the
// sensitivities (fit_sens.hip) — layout and meaning of moment_solve_kernel<.., INV> (fit_moment.hip): inv[group of 64][column][case][row].
// PART = 0: the whole fit.  PART = 1 / 2 (3D order 4, 35 unknowns): the kernel stops after the moment pass and leaves HALF of the case's
// 165 + 35 moments (wlsqm_moments.hpp: stage_part) at p.ws — entry e of case t at ws[((t / 64) 200 + e) 64 + t % 64] — for the
// four-lanes-per-case solve of csrc/fit_quad.hip; all 200 accumulators at once are 400 registers, more than an instruction can name.
// GATHER: index-based input (p.hoods: the neighbours of case j are rows hoods[j, k] of the point table p.S / p.F).  A lane gathers ITS
// case's neighbours, so the data arrives in the lane that consumes it: no LDS staging — the next chunk's eight points are in flight in
// registers while the current ones are consumed.  Everything behind the fetch is the dense kernel's code (same bits as the dense
// kernel on the gathered rows).
// RAGGED (round 6): a copy of the same code whose passes run over the chunks ITS wave's cases need — Q = ceil(max over the lanes of nk / CH)
// instead of the row's QK — for the reference's own harness shape: a ball query, nk 30..100 in rows of 100 slots (examples/wlsqm_example.py:
// 103-133; only entries below nk[j] are touched: simple.pyx:147).  A COPY, not a run-time choice inside one kernel: BASELINE configs[2]'s kernel
// is a lone wave per SIMD walking ~85 KB of code past a 64 KB instruction cache, and every version that carried the ragged path inside it lost
// 10-13 % on full rows (chunk count and warm start as run-time values: 0.457 against 0.405 ms; profiles/r06c_ragged.txt).  status != nullptr
// (RAGGED = false, a batch whose neighbour counts the host has not seen): a wave none of whose cases reaches the row's last chunk marks its
// group and leaves; fit_stage_ragged_kernel behind it runs the marked groups through the RAGGED copy.
template <int DIM, int ORDER, int PART, bool GATHER, int RAG>      // RAG: 0 the plain kernel, 1 it marks its short waves in `status` and leaves them, 2 the RAGGED copy
__device__ __forceinline__ void stage_group(const KParams& p, const unsigned grp, unsigned char* const status) {
    constexpr bool RAGGED = RAG == 2;
    using namespace stage;
    constexpr int NO = ndofs(DIM, ORDER), NE = NO * (NO + 1) / 2, NM = mom_count<DIM>(2 * ORDER);
#ifndef WLSQM_STAGE_CH10
#define WLSQM_STAGE_CH10 8          // neighbours per staged chunk of the systems with 7 .. 10 unknowns (4: 96-byte pieces straddle 64-byte sectors — by LDS-DMA at two waves per SIMD 0.30 against 0.275 ms and 1.83 GB fetched per 1M configs[4] cases, profiles/r05g_ab_stage_dma.txt; register-staged the loops spilled, r04z_ab_c5_two_waves.txt)
#endif
    constexpr int CH = (NO > 6 && NO <= 10) ? WLSQM_STAGE_CH10 : stage::CH;
    static_assert(mom_count<DIM>(ORDER) == NO, "one right-hand-side moment per DOF");
    constexpr int XPC = CH * DIM * 8 / 16, FPC = CH * 8 / 16;        // 16-byte pieces of one case's chunk: coordinates, values
    constexpr int XPITCH = CH * DIM + 2, FPITCH = CH + 2;            // doubles per staged row (+ 16 bytes: conflict-free b128 reads)
    constexpr int XCPI = 64 / XPC, XNI = (64 + XCPI - 1) / XCPI;      // whole cases per load instruction; instructions per chunk
    constexpr int FCPI = 64 / FPC, FNI = 64 / FCPI;
    // neighbours the scheduler may interleave: the 6-unknown systems run two waves per SIMD and fit their 256 registers only with
    // two at a time (no scratch; four: 44 B, eight: 76 B); the larger systems own their SIMD and take the whole chunk
    constexpr bool SCHED_BARRIER = WLSQM_STAGE_SCHED_BARRIER || NO <= 6 || NO >= 20 || (NO > 10 && WLSQM_STAGE_GRP15 < CH);
    constexpr int GRP = WLSQM_STAGE_SCHED_BARRIER ? (WLSQM_STAGE_GRP < CH ? WLSQM_STAGE_GRP : CH) : (NO <= 6 ? 2 : NO >= 20 ? WLSQM_STAGE_GRP20 : NO > 10 ? (WLSQM_STAGE_GRP15 < CH ? WLSQM_STAGE_GRP15 : CH) : CH);
    // the staging rows; behind them (reusing the same bytes after the last chunk) the 64 result rows of the wave
    // (and, for the 20-unknown systems, the top R0 rows of every lane's normal matrix during the solve: see below)
    constexpr int R0 = NO == 20 ? WLSQM_STAGE_LDS_ROWS : 0, TOP_D = 64 * tri<NO>(R0, R0);
    // DMA (round 5): the chunks go from memory STRAIGHT into LDS (global_load_lds_dwordx4: no vector register holds a chunk in flight), into
    // a ring of NSLOT slots with NSLOT - 1 chunks in flight while one is consumed.  A wave instruction's 64 pieces land lane-linear —
    // piece of lane l at 16 l of the instruction's KiB — so the image of a slot is [instruction][case in instruction][piece] and case c reads
    // its CH neighbours at KiB c / XCPI, offset (c % XCPI) XPC 16 (contiguous, as in the padded rows of the register-staged form; the idle
    // lanes of an instruction land in the tail of its KiB, which nobody reads).  The wave waits for its OWN transfers with a counted
    // s_waitcnt vmcnt — all that orders a ds_read behind an LDS-DMA of the same wave (MI355X_MICROARCH.md, co-residence item 7).
#ifndef WLSQM_STAGE_DMA
#define WLSQM_STAGE_DMA 18           // bit 0: the systems up to 6 unknowns, 1: 7..10, 2: 11..15, 3: 20, 4: 35
#endif
#ifndef WLSQM_STAGE_DMA_NT
#define WLSQM_STAGE_DMA_NT 0        // the transfers with the non-temporal policy
#endif
#if WLSQM_STAGE_DMA_NT
#define WLSQM_STAGE_DMA_POLICY " nt"
#else
#define WLSQM_STAGE_DMA_POLICY ""
#endif
#ifndef WLSQM_STAGE_DMA_SLOTS
#define WLSQM_STAGE_DMA_SLOTS 2      // slots of the ring (1: no chunk in flight while one is consumed — the second wave of the SIMD covers the wait)
#endif
#ifndef WLSQM_STAGE_DMA_SLOTS10
#define WLSQM_STAGE_DMA_SLOTS10 1    // ... of the systems with 7..10 unknowns (two waves per SIMD: 17 KB each)
#endif
    constexpr bool DMA = !GATHER && PART != 5 && (((WLSQM_STAGE_DMA) >> (NO <= 6 ? 0 : NO <= 10 ? 1 : NO <= 15 ? 2 : NO <= 20 ? 3 : 4)) & 1);
#ifndef WLSQM_STAGE_DMA_SLOTS35
#define WLSQM_STAGE_DMA_SLOTS35 2
#endif
    constexpr int NSLOT = DMA ? ((NO > 6 && NO <= 10) ? WLSQM_STAGE_DMA_SLOTS10 : NO == 35 ? WLSQM_STAGE_DMA_SLOTS35 : WLSQM_STAGE_DMA_SLOTS) : 1, PF = NSLOT - 1;
    // SKEW (round 6): instruction i lands at i (1 KiB + 16 B), not at i KiB.  A lane reads ITS case's neighbours with
    // ds_read_b128, and the 16 lanes of a read group (MI355X_MICROARCH.md: {0-3, 12-15, 20-27}, ...) must hit 16 different 16-byte slots of the
    // 256-byte bank row; in the dense KiB image the slot of case c was (12 (c mod 5)) mod 16 (3D: 192-byte chunks, 5 cases per KiB) — four
    // slots for sixteen lanes, 6-way conflicts (`SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE` 0.80 on configs[4], profiles/r05_C5_pmc_summary.json)
    // — or (8 (c mod 8)) mod 16 (2D order 3: 8-way).  With the instruction's KiB shifted by one slot the worst group is 2-3-way (brute force
    // over the four lane groups: 25 -> 9 of an ideal 4 cycles per read in 3D, 32 -> 8 in 2D, 16 -> 8 for the values).
#ifndef WLSQM_STAGE_DMA_SKEW
#define WLSQM_STAGE_DMA_SKEW 16
#endif
    constexpr int KIB_B = 1024 + WLSQM_STAGE_DMA_SKEW, KIB_D = KIB_B / 8;      // bytes / doubles from one instruction's image to the next
    constexpr int SLOT_D = (XNI + FNI) * KIB_D;                        // doubles of one slot
    static_assert(!DMA || (NSLOT >= 1 && PF * (XNI + FNI) <= 63), "the chunks in flight must fit the vmcnt counter");
    constexpr int STAGE_D = DMA ? NSLOT * SLOT_D : 64 * XPITCH + 64 * FPITCH, OUT_D = 64 * NO;
    constexpr int LDS_D = (STAGE_D > OUT_D ? STAGE_D : OUT_D) > TOP_D ? (STAGE_D > OUT_D ? STAGE_D : OUT_D) : TOP_D;
    __shared__ __attribute__((aligned(16))) double lds[LDS_D];
    double* const xs = lds;
    double* const fs = lds + 64 * XPITCH;

    const int lane = threadIdx.x;
    const long long t0 = (long long)grp * 64, t = t0 + lane;
    const int nvalid = (p.ncases - t0 < 64) ? (int)(p.ncases - t0) : 64;      // wave-uniform
    const bool valid = lane < nvalid;
    const long long j = valid ? t : t0 + nvalid - 1;                          // tail lanes replay the last case (never stored)
    const int K = (int)p.max_nk;
    const int QK = (K + CH - 1) / CH;                                 // chunks of a whole row (a last partial chunk: its pieces beyond the row replay the row's last one; masked)
    // ---- staging: a load instruction moves the chunks of XCPI (FCPI) whole cases, XPC (FPC) consecutive lanes per case
    const int xsub = lane % XPC, xc0 = lane / XPC, fsub = lane % FPC, fc0 = lane / FPC;
    const unsigned xrowb = (unsigned)K * DIM * 8, frowb = (unsigned)K * 8;
    const bool xlane = lane < XCPI * XPC;
    const char* const xtile = GATHER ? nullptr : reinterpret_cast<const char*>(p.xk + t0 * (long long)K * DIM);
    const char* const ftile = GATHER ? nullptr : reinterpret_cast<const char*>(p.fk + t0 * (long long)K);
    // GATHER: the chunk in flight (nx, nf) and the chunk being consumed (cx, cf)
    double nx[GATHER ? CH : 1][DIM], nf[GATHER ? CH : 1], cx[GATHER ? CH : 1][DIM], cf[GATHER ? CH : 1];
    const bool rows16 = GATHER && ((reinterpret_cast<uintptr_t>(p.hoods) & 15u) == 0) && (p.shoods_j % 4 == 0);      // wave-uniform
    d2_ xr[XNI], fr[FNI];
    bool want_f = true;                                               // wave-uniform: the pass that only looks for the largest squared distance moves no values
    auto fetch_into = [&](d2_ (&xr)[XNI], d2_ (&fr)[FNI], int q) __attribute__((always_inline)) {
        unsigned xo = (unsigned)q * (CH * DIM * 8) + (unsigned)xsub * 16u, fo = (unsigned)q * (CH * 8) + (unsigned)fsub * 16u;
        xo = xo < xrowb ? xo : xrowb - 16u; fo = fo < frowb ? fo : frowb - 16u;      // (rows are multiples of 16 bytes: K even)
        const char* xb = xtile + xo;
        const char* fb = ftile + fo;
#pragma unroll
        for (int i = 0; i < XNI; ++i) {
            int cc = xc0 + i * XCPI;
            cc = cc < nvalid ? cc : nvalid - 1;                       // tail group / idle lanes of the last instruction: replay a valid row
#if WLSQM_STAGE_NT_LOADS
            if (xlane) xr[i] = __builtin_nontemporal_load(reinterpret_cast<const d2_*>(xb + (size_t)(unsigned)cc * xrowb));
#else
            if (xlane) xr[i] = *reinterpret_cast<const d2_*>(xb + (size_t)(unsigned)cc * xrowb);
#endif
        }
        if (want_f) {
#pragma unroll
            for (int i = 0; i < FNI; ++i) {
                int cc = fc0 + i * FCPI;
                cc = cc < nvalid ? cc : nvalid - 1;
#if WLSQM_STAGE_NT_LOADS
                fr[i] = __builtin_nontemporal_load(reinterpret_cast<const d2_*>(fb + (size_t)(unsigned)cc * frowb));
#else
                fr[i] = *reinterpret_cast<const d2_*>(fb + (size_t)(unsigned)cc * frowb);
#endif
            }
        }
    };
    // ---- DMA: chunk q into slot q % NSLOT; wait for a chunk with `younger` chunks requested after it
    auto dma_fetch = [&](int q) __attribute__((always_inline)) {
        const char* const xt = xtile; const char* const ft = ftile;
        (void)xt; (void)ft;
        if constexpr (DMA) {
            unsigned xo = (unsigned)q * (CH * DIM * 8) + (unsigned)xsub * 16u, fo = (unsigned)q * (CH * 8) + (unsigned)fsub * 16u;
            xo = xo < xrowb ? xo : xrowb - 16u; fo = fo < frowb ? fo : frowb - 16u;
            const unsigned slot = (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)(uintptr_t)lds + (unsigned)(q % NSLOT) * (unsigned)(SLOT_D * 8)));
#ifndef WLSQM_STAGE_DMA_SPLIT
#define WLSQM_STAGE_DMA_SPLIT 1      // (configs[4]: same 0.275 ms, FETCH_SIZE 1.80 -> 1.64 GB per launch; its shuffled form 0.427 -> 0.406 ms: profiles/r05g_ab_stage_dma.txt)
#endif
            constexpr bool SPLIT = WLSQM_STAGE_DMA_SPLIT && PF == 0;      // (two masked instructions per transfer: only where every wait is vmcnt(0))
            const unsigned xt_lo = (unsigned)(uintptr_t)xt & 127u;
            (void)xt_lo;
#pragma unroll
            for (int i = 0; i < XNI; ++i) {
                int cc = xc0 + i * XCPI;
                cc = cc < nvalid ? cc : nvalid - 1;                   // (idle lanes and tail groups replay a valid row)
                unsigned keep;
                if constexpr (SPLIT) {
                    // a piece whose 128-byte line lies wholly inside this chunk's run of the row is read once: non-temporal; a line shared
                    // with the neighbouring chunk (or row) keeps the default policy, so that L2 holds the halves that WILL be read again
                    const unsigned a = xt_lo + xo + (unsigned)cc * xrowb, line = a & ~127u;
                    const unsigned r0 = xt_lo + (unsigned)cc * xrowb + (unsigned)q * (CH * DIM * 8);
                    const unsigned rend = xt_lo + (unsigned)(cc + 1) * xrowb, r1 = r0 + CH * DIM * 8 < rend ? r0 + CH * DIM * 8 : rend;
                    const unsigned long long inner = __ballot(line >= r0 && line + 128u <= r1);
                    unsigned long long keepx;
                    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_mov_b64 %1, exec\n\ts_and_b64 exec, %1, %5\n\tglobal_load_lds_dwordx4 %2, %3 nt\n\t"
                                 "s_andn2_b64 exec, %1, %5\n\tglobal_load_lds_dwordx4 %2, %3\n\ts_mov_b64 exec, %1\n\ts_mov_b32 m0, %0"
                                 : "=&s"(keep), "=&s"(keepx) : "v"(xo + (unsigned)cc * xrowb), "s"(xt), "s"(slot + (unsigned)i * (unsigned)KIB_B), "s"(inner) : "memory", "scc");      // (s_and_b64 / s_andn2_b64 write SCC: ADVICE r5)
                } else
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" WLSQM_STAGE_DMA_POLICY "\n\ts_mov_b32 m0, %0"
                             : "=&s"(keep) : "v"(xo + (unsigned)cc * xrowb), "s"(xt), "s"(slot + (unsigned)i * (unsigned)KIB_B) : "memory");
            }
            if (want_f) {
#pragma unroll
                for (int i = 0; i < FNI; ++i) {
                    int cc = fc0 + i * FCPI;
                    cc = cc < nvalid ? cc : nvalid - 1;
                    unsigned keep;
                    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" WLSQM_STAGE_DMA_POLICY "\n\ts_mov_b32 m0, %0"
                                 : "=&s"(keep) : "v"(fo + (unsigned)cc * frowb), "s"(ft), "s"(slot + (unsigned)(XNI + i) * (unsigned)KIB_B) : "memory");
                }
            }
        }
    };
    auto dma_wait = [&](int younger) __attribute__((always_inline)) {
        if constexpr (DMA) {
            // (in-order counter: anything else in flight only makes the wait longer, never too short)
            if (younger <= 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else if (younger == 1) { if (want_f) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(XNI + FNI) : "memory"); else asm volatile("s_waitcnt vmcnt(%0)" :: "n"(XNI) : "memory"); }
            else if (PF >= 2 && younger == 2) { if (want_f) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(PF >= 2 ? 2 * (XNI + FNI) : 0) : "memory"); else asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * XNI) : "memory"); }
            else { if (want_f) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(PF >= 3 ? 3 * (XNI + FNI) : 0) : "memory"); else asm volatile("s_waitcnt vmcnt(%0)" :: "n"(PF >= 3 ? 3 * XNI : 0) : "memory"); }
        }
    };
    auto dma_prime = [&](const int Qn) __attribute__((always_inline)) {      // the first PF chunks of a pass over Qn chunks (processed last chunk first)
#pragma unroll
        for (int i = 1; i <= PF; ++i) if (Qn - i >= 0) dma_fetch(Qn - i);
    };
    // the rows of lane's case in chunk q
    auto xrow_of = [&](int q, bool image) __attribute__((always_inline)) -> const double* {
        return image ? lds + (q % NSLOT) * SLOT_D + (lane / XCPI) * KIB_D + (lane % XCPI) * (XPC * 2) : xs + lane * XPITCH;
    };
    auto frow_of = [&](int q, bool image) __attribute__((always_inline)) -> const double* {
        return image ? lds + (q % NSLOT) * SLOT_D + XNI * KIB_D + (lane / FCPI) * KIB_D + (lane % FCPI) * (FPC * 2) : fs + lane * FPITCH;
    };
    // EARLY FETCH (round 5): the first staged chunk is requested BEFORE the case's scalars are — its addresses depend on the launch
    // parameters alone.  In source order the scalars (nk, weighting, knowns: one round trip), then the centre and a ragged case's last
    // neighbour (a second), then the chunk (a third) were three memory latencies in a row at the head of every 64-case group, ~3.5 us of a
    // lone wave's 18-26 us per group (ISA: s_waitcnt vmcnt(1) / vmcnt(0) in front of the chunk's first load); now they overlap.
    // Measured (tools/ab_unit.sh, one box, profiles/r05f_ab_early_fetch.txt): configs[2] 0.392 -> 0.3876 ms (+1.2 %); configs[1] and configs[4]
    // within +-0.5 % either way (on for the 15-unknown systems and up: 2).  The head of a group is not where a lone wave loses its time.
#ifndef WLSQM_STAGE_EARLY_FETCH
#define WLSQM_STAGE_EARLY_FETCH 2
#endif
    constexpr bool EARLY = (WLSQM_STAGE_EARLY_FETCH != 0) && !GATHER && !RAGGED && (WLSQM_STAGE_EARLY_FETCH == 1 || NO > 10);      // (2: the 15-unknown systems and up only)
    if constexpr (EARLY && DMA) { dma_prime(QK); if constexpr (PF == 0) dma_fetch(QK - 1); }      // (a ring of one slot: its only chunk)
    else if constexpr (EARLY) fetch_into(xr, fr, QK - 1);
    // (every scalar of the case is requested before the first of them is looked at: the mask arithmetic below waits for its own)
    double xi[DIM];
    const long long pj = GATHER ? (own_point(p, j)) : 0;      // the case's own point (also the stand-in for padding slots)
    const int* const hrow = GATHER ? p.hoods + j * p.shoods_j : nullptr;
#pragma unroll
    for (int m = 0; m < DIM; ++m) xi[m] = GATHER ? p.S[pj * DIM + m] : p.xi[j * p.sxi_j + m];
    const int nk_raw = p.nk[j * p.snk];
    const int wm_raw = p.wm[j * p.swm];
    const long long kn_raw = p.knowns[j * p.sknowns];
    const int nkc = min(nk_raw, K);
    const bool uniform = (wm_raw == WLSQM_WEIGHT_UNIFORM);
    // Q: the chunks this wave's passes run over
    int Qw = QK;
    if constexpr (RAGGED) {
        int nkmax = nkc;                                              // (tail lanes replay the group's last case)
#pragma unroll
        for (int sft = 32; sft >= 1; sft >>= 1) nkmax = max(nkmax, __shfl_xor(nkmax, sft, 64));
        Qw = max(1, (__builtin_amdgcn_readfirstlane(nkmax) + CH - 1) / CH);
    } else if constexpr (RAG == 1) {
        const bool short_wave = (QK > 1) && !__any(nkc > (QK - 1) * CH);      // wave-uniform
        if (lane == 0) status[grp] = short_wave ? 1 : 0;
        if (short_wave) return;
    }
    const int Q = Qw;
    unsigned long long known, dropped;
    effective_mask<NO>(kn_raw, known, dropped);

    auto sqdist = [&](const double (&d)[DIM]) {       // one rounding sequence for the guess and for the pass: they are compared for equality
        double d2 = d[0] * d[0];
#pragma unroll
        for (int m = 1; m < DIM; ++m) d2 = fma(d[m], d[m], d2);
        return d2;
    };
    // the guess: the last neighbour is the farthest (see the header).  The chunks are processed LAST CHUNK FIRST (neighbours in
    // descending k: a fixed order per case, so results do not depend on the route), so that the guess comes out of the first staged
    // chunk: a separate load of the last neighbour at the start fetched a whole line per case that the staging fetched AGAIN four
    // to eight chunks later (954 instead of 800 MB read per 1M configs[1] cases, 1 530 instead of 1 321 MB on configs[4]:
    // profiles/r04m_*_pmc_summary.json).  Only a ragged case whose last neighbour lies in an earlier chunk loads it directly.
    double guess = 0.0;
    const bool guess_staged = nkc > (Q - 1) * CH;                     // the last neighbour sits in the chunk that is staged first
    if (nkc > 0 && !guess_staged) {
        const double* q = GATHER ? p.S + (long long)hrow[nkc - 1] * DIM : p.xk + j * (long long)K * DIM + (long long)(nkc - 1) * DIM;
        double dg[DIM];
#pragma unroll
        for (int m = 0; m < DIM; ++m) dg[m] = q[m] - xi[m];
        guess = sqdist(dg);
    }

    auto gather = [&](int q) __attribute__((always_inline)) {
        if constexpr (GATHER) {
            int idx[CH];
            if (CH == 8 && rows16 && (q + 1) * CH <= K) {             // whole chunk, 16-byte aligned index rows: two loads
                typedef int i4_ __attribute__((ext_vector_type(4)));
                const i4_ a = *reinterpret_cast<const i4_*>(hrow + q * CH), b = *reinterpret_cast<const i4_*>(hrow + q * CH + 4);
                const int both[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
                for (int ks = 0; ks < CH; ++ks) idx[ks] = both[ks < 8 ? ks : 7];
            } else {
#pragma unroll
                for (int ks = 0; ks < CH; ++ks) { const int k = q * CH + ks; idx[ks] = hrow[k < K ? k : K - 1]; }
            }
#pragma unroll
            for (int ks = 0; ks < CH; ++ks) {
                // a slot beyond the case's neighbours holds anything (padding): never dereferenced, the case's own point stands in
                const long long pk = (q * CH + ks < nkc) ? (long long)idx[ks] : pj;
#pragma unroll
                for (int m = 0; m < DIM; ++m) nx[ks][m] = p.S[pk * DIM + m];
                nf[ks] = p.F[pk];
            }
        }
    };
    auto fetch = [&](int q) __attribute__((always_inline)) { if constexpr (GATHER) gather(q); else fetch_into(xr, fr, q); };
    auto park_from = [&](const d2_ (&xr)[XNI], const d2_ (&fr)[FNI]) __attribute__((always_inline)) {
        double* xl = xs + xc0 * XPITCH + xsub * 2;
        double* fl = fs + fc0 * FPITCH + fsub * 2;
#pragma unroll
        for (int i = 0; i < XNI; ++i)
            if (xlane && (i * XCPI + XCPI <= 64 || xc0 + i * XCPI < 64)) *reinterpret_cast<d2_*>(xl + i * XCPI * XPITCH) = xr[i];
        if (want_f) {
#pragma unroll
            for (int i = 0; i < FNI; ++i) *reinterpret_cast<d2_*>(fl + i * FCPI * FPITCH) = fr[i];
        }
    };
    auto park = [&]() __attribute__((always_inline)) {
        if constexpr (GATHER) {
#pragma unroll
            for (int ks = 0; ks < CH; ++ks) {
#pragma unroll
                for (int m = 0; m < DIM; ++m) cx[ks][m] = nx[ks][m];
                cf[ks] = nf[ks];
            }
        } else park_from(xr, fr);
    };

    constexpr bool INVERSE = PART == 4;                               // (the moment pass of PART 4 is PART 0's)
    double mu[NM], nu[NO];
    double max_d2 = 0.0;
    // One pass over the neighbours.  MAXONLY: only the largest squared distance is computed; otherwise the moments, with `maxv` as the
    // largest squared distance of the lane's case.  MASKED (some case of the group is ragged, or K is not a multiple of CH) is decided
    // ONCE per pass: with the choice inside the chunk loop the two variants' 60 accumulators met in different registers and every
    // iteration paid ~50 copies per 4 neighbours at the join (first version: 179 instructions per neighbour).  warm: chunk 0 is
    // Q - 1 is already parked and chunk Q - 2 in flight (the prologue below).
    auto pass_impl = [&](auto masked_tag, auto mode_tag, const double maxv, const bool warm, const int Q) __attribute__((always_inline)) {      // Q: chunks of the pass (the row's, or a wave of short cases' own)
        constexpr bool MASKED = decltype(masked_tag)::value;
        constexpr int MODE = decltype(mode_tag)::value;               // 0: moments with the maximum `maxv`; 1: the maximum only
        constexpr bool MAXONLY = MODE != 0;                           // (no weight in this pass)
        if constexpr (MODE == 0) {
#pragma unroll
            for (int e = 0; e < NM; ++e) mu[e] = 0.0;
#pragma unroll
            for (int a = 0; a < NO; ++a) nu[a] = 0.0;
        }
        max_d2 = 0.0;
        const double inv_max = MAXONLY ? 0.0 : inverse_max(maxv);
        auto chunk = [&](const int q) __attribute__((always_inline)) {
            const double* const xrow = xrow_of(q, DMA);
            const double* const frow = frow_of(q, DMA);
#pragma unroll
            for (int g = CH / GRP - 1; g >= 0; --g) {
#pragma unroll
                for (int kk = GRP - 1; kk >= 0; --kk) {
                    const int ks = g * GRP + kk;
                    const bool live = MASKED ? (q * CH + ks < nkc) : true;
                    double d[DIM];
#pragma unroll
                    for (int m = 0; m < DIM; ++m) { d[m] = (GATHER ? cx[ks][m] : xrow[ks * DIM + m]) - xi[m]; if (MASKED) d[m] = live ? d[m] : 0.0; }
                    const double d2 = sqdist(d);
                    max_d2 = d2 > max_d2 ? d2 : max_d2;               // (a masked slot contributes 0)
                    if constexpr (MODE == 0) {
                        double w = weight(d2, inv_max, uniform);
                        double f = GATHER ? cf[ks] : frow[ks];
                        if (MASKED) { w = live ? w : 0.0; f = live ? f : 0.0; }
#ifndef WLSQM_STAGE_OUTER3D
#define WLSQM_STAGE_OUTER3D 1
#endif
#ifndef WLSQM_STAGE_OUTER3D_FROM
#define WLSQM_STAGE_OUTER3D_FROM 3
#endif
                        if constexpr (DIM == 3 && ORDER >= WLSQM_STAGE_OUTER3D_FROM && WLSQM_STAGE_OUTER3D) accumulate_moments_outer3d<ORDER, (PART == 1 || PART == 2) ? PART : 0>(mu, nu, d, w, f);
                        else accumulate_moments_best<DIM, ORDER>(mu, nu, d, w, f);
                    }
                }
                if constexpr (SCHED_BARRIER) __builtin_amdgcn_sched_barrier(0);      // GRP neighbours in flight at a time
            }
        };
        if constexpr (DMA) {
            // (warm: chunk Q - 1 landed, chunks Q - 2 .. Q - 1 - PF in flight: the prologue)
            if (!warm) dma_prime(Q);
            for (int q = Q - 1; q >= 0; --q) {
                if (!(warm && q == Q - 1)) {
                    if (q - PF >= 0) dma_fetch(q - PF);               // into the slot of chunk q + 1: consumed
                    dma_wait(q < PF ? q : PF);
                }
                chunk(q);
            }
        } else {
            if (!warm) fetch(Q - 1);
            for (int q = Q - 1; q >= 0; --q) {
                if (!(warm && q == Q - 1)) {
                    __syncthreads();                                  // the previous chunk has been read by every lane
                    park();
                    __syncthreads();
                    if (q > 0) fetch(q - 1);
                }
                chunk(q);
            }
        }
    };
    const bool full = (K % CH == 0) && __all(nkc >= K);               // wave-uniform: no ragged case in this group, whole chunks
    using M0 = std::integral_constant<int, 0>;
    using M1 = std::integral_constant<int, 1>;
    auto moments = [&](const double maxv, const bool warm) __attribute__((always_inline)) {
        if (full) pass_impl(std::false_type{}, M0{}, maxv, warm, Q); else pass_impl(std::true_type{}, M0{}, maxv, warm, Q);
    };
    // ---- prologue: the LAST chunk parked, the one before it requested; is this group's input SORTED by distance?  The speculation below pays only
    // then (a wrong guess costs a whole second pass: 1.9x): the squared distances of that chunk must be non-decreasing in every
    // lane — by chance for unsorted neighbours with probability 1 / 8! per case.  Unsorted input (a ball query) takes the plain two
    // passes instead: the largest squared distance first (a few instructions per neighbour), then the moments.
    if constexpr (DMA) {
        if constexpr (!EARLY) dma_prime(Q);
        if constexpr (!(EARLY && PF == 0)) { if (Q - 1 - PF >= 0) dma_fetch(Q - 1 - PF); }
        dma_wait(Q - 1 < PF ? Q - 1 : PF);
    } else {
        if constexpr (!EARLY) fetch(Q - 1);
        __syncthreads();
        park();
        __syncthreads();
        if (Q > 1) fetch(Q - 2);
    }
    bool mono = true;
    {
        const double* xrow = xrow_of(Q - 1, DMA);
        double prev = 0.0;
#pragma unroll
        for (int ks = 0; ks < CH; ++ks) {
            double d[DIM];
#pragma unroll
            for (int m = 0; m < DIM; ++m) d[m] = (GATHER ? cx[ks][m] : xrow[ks * DIM + m]) - xi[m];
            const double d2 = sqdist(d);
            if ((Q - 1) * CH + ks < nkc) { mono = mono && d2 >= prev; prev = d2; }
        }
        if (guess_staged) guess = prev;                               // the squared distance of neighbour nkc - 1
    }
    // Which arithmetic a case gets is a property of the case alone (its bits do not depend on its wave-mates): a case whose first staged
    // chunk is sorted — and every uniformly weighted case — gets the moments weighted with its largest squared distance (speculative
    // pass, repeated if the guess was wrong); a wave with an unsorted chunk runs the two plain passes (same bits either way).
    // (Round 5's one-pass form with three sets of sums — 0.270 against 0.249 ms and a digit lost to cancellation, profiles/r05b_unsorted.txt —,
    // its DEEP / AGPR forms of two chunks in flight — r05c_ab_stage_agpr.txt — are out of the source since round 6: git history, commit c9ef97d.)
    const bool elig_r = uniform || mono;
    if (__all(elig_r)) {
        moments(guess, true);
        // the guess must have been the largest squared distance, bit for bit (uniform weighting does not use it); otherwise the wave
        // repeats the pass with the true maxima — lanes whose guess was right get the same bits again
        if (!__all(uniform || max_d2 == guess)) moments(max_d2, false);
    } else {
        // unsorted neighbours: the largest squared distance first (coordinates only: the values travel with the second pass), then the moments
        if constexpr (!GATHER) want_f = false;
        if (full) pass_impl(std::false_type{}, M1{}, 0.0, true, Q); else pass_impl(std::true_type{}, M1{}, 0.0, true, Q);
        want_f = true;
        moments(max_d2, false);
    }

    if constexpr (PART == 1 || PART == 2) {
        // ---- this half of the moments to the workspace: 512 contiguous bytes per entry (tail lanes store their replayed case too: the
        // solve reads whole 16-case runs)
        double* const out = p.ws + (long long)grp * (200 * 64) + lane;
#pragma unroll
        for (int r = 0; r <= 2 * ORDER; ++r) {
            if (stage_part(r) != PART) continue;
#pragma unroll
            for (int q = 0; q + r <= 2 * ORDER; ++q) {
#pragma unroll
                for (int pp = 0; pp + q + r <= 2 * ORDER; ++pp) {
                    __builtin_nontemporal_store(mu[mom_index<3>(pp, q, r)], out + mom_index<3>(pp, q, r) * 64);
                    if (pp + q + r <= ORDER) __builtin_nontemporal_store(nu[mom_index<3>(pp, q, r)], out + (NM + mom_index<3>(pp, q, r)) * 64);
                }
            }
        }
        return;
    } else {
    // ---- one solve per lane: masked full system, unpivoted LDL^T
    constexpr unsigned long long FULL = (1ull << NO) - 1ull;
    double* const fio = p.fi + j * p.sfi_j;
    double rhs[NO];
    // A case with exactly the function value known (knowns = b?_F: the reference's default mask and BASELINE configs[2]) solves the
    // (NO - 1) x (NO - 1) system directly: 105 + 14 instead of 120 + 15 entries to expand, factor and substitute for 15 DOFs.  Chosen
    // by the case's own mask (a mixed wave runs both forms), so a case's bits do not depend on its wave-mates.
    // (only for the 15-unknown systems: the smaller ones gain nothing and their register allocation suffered — 2D order 3 went from
    // 330 registers to 512 + 704 B of scratch with the second form compiled in)
    constexpr bool REDUCED = NO == 15;
    const bool mine1 = REDUCED && known == 1ull && dropped == 0ull;
    double Mf[INVERSE ? NE : 1];                                      // PART 4: the factor the solve leaves (the reduced form's 105 entries first)
    if constexpr (REDUCED) {
        if (mine1) {
            constexpr int N1 = NO - 1, NE1 = N1 * (N1 + 1) / 2;
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
            if constexpr (INVERSE) {
#pragma unroll
                for (int e = 0; e < NE1; ++e) Mf[e] = M1[e];
            }
            rhs[0] = v0;                                              // (for the whole-row store below: its own bits)
#pragma unroll
            for (int a = 1; a < NO; ++a) rhs[a] = r1[a - 1];
        }
    }
    if constexpr (R0 > 0) {
        // ---- 20 unknowns: 210 + 20 entries are 460 registers, more than a lane has next to anything else, and a lone wave per
        // SIMD waits out every scratch access in full (all in registers: 2.3 KB of scratch per lane and 1.96 ms per 200k cases, half
        // the speed of the wave-per-case kernel this replaces).  The top R0 rows of the matrix — the ones the elimination is done
        // with first — live in LDS instead, entry e of lane l at lds[64 e + l] (conflict-free), where the staging rows were; the
        // trailing (NO - R0) x (NO - R0) block and the right-hand side stay in registers as for the smaller systems.
        constexpr int N2 = NO - R0, NE2 = N2 * (N2 + 1) / 2;
        __syncthreads();                                              // the last chunk has been read
        double* const L = lds + lane;
        double R[NE2];
        auto at = [&](int a, int b) __attribute__((always_inline)) -> double {          // a <= b
            return a < R0 ? L[tri<NO>(a, b) * 64] : R[tri<N2>(a - R0, b - R0)];
        };
        auto put = [&](int a, int b, double v) __attribute__((always_inline)) {
            if (a < R0) L[tri<NO>(a, b) * 64] = v; else R[tri<N2>(a - R0, b - R0)] = v;
        };
        // Entry (a, b), a <= b, of the masked full system straight from the moments (by entry: the moment-by-moment form of
        // expand_moments is 84 x 20 x 20 iterations here, beyond what the compiler unrolls — the moments would be indexed at run time
        // and live in scratch).  The matrix is never materialised before the elimination: step 0 of the factorisation takes its
        // operands from here, so the 84 moments die while the 136 register entries are born (both at once do not fit the lane).
        const unsigned k32 = (unsigned)known, t32 = (unsigned)(known & ~dropped);        // (20 bits)
        const bool any_known = __any(k32 != 0u);                      // wave-uniform: the common wave has no known DOF at all
        auto raw = [&](int a, int b) __attribute__((always_inline)) -> double {
            const int pa = Mono<DIM>::P[a], qa = Mono<DIM>::Q[a], ra = Mono<DIM>::R[a];
            const int pb = Mono<DIM>::P[b], qb = Mono<DIM>::Q[b], rb = Mono<DIM>::R[b];
            const double fa = mom_inv_fact(pa) * mom_inv_fact(qa) * mom_inv_fact(ra), fb = mom_inv_fact(pb) * mom_inv_fact(qb) * mom_inv_fact(rb);
            return mu[mom_index<DIM>(pa + pb, qa + qb, ra + rb)] * (fa * fb);
        };
        auto entry = [&](int a, int b) __attribute__((always_inline)) -> double {       // masked to identity in the known DOFs
            const double m = raw(a, b);
            if (!any_known) return m;
            return a == b ? (((k32 >> a) & 1u) ? 1.0 : m) : ((((k32 >> a) | (k32 >> b)) & 1u) ? 0.0 : m);
        };
        double sol[NO];
#pragma unroll
        for (int a = 0; a < NO; ++a) {
            const int pa = Mono<DIM>::P[a], qa = Mono<DIM>::Q[a], ra = Mono<DIM>::R[a];
            sol[a] = nu[mom_index<DIM>(pa, qa, ra)] * (mom_inv_fact(pa) * mom_inv_fact(qa) * mom_inv_fact(ra));
        }
        // knowns elimination (eliminate_knowns), branch-free per lane behind the wave-uniform test: the value of an unknown DOF
        // enters as 0.0 (fma(-m, 0, s) = s exactly), so a case's bits do not depend on its wave-mates' masks
        if (any_known) {
#pragma unroll
            for (int om = 0; om < NO; ++om) {
                const double v = ((t32 >> om) & 1u) ? fio[om] : 0.0;
#pragma unroll
                for (int a = 0; a < NO; ++a)
                    if (a != om) sol[a] = fma(-(a < om ? raw(a, om) : raw(om, a)), v, sol[a]);
            }
#pragma unroll
            for (int a = 0; a < NO; ++a) sol[a] = ((k32 >> a) & 1u) ? 0.0 : sol[a];
        }
        // LDL^T, right-looking: the LDS rows first (pivot row in registers, trailing updates into LDS or registers), then the
        // register block with the routine of the smaller systems
#pragma unroll
        for (int j = 0; j < R0; ++j) {
            double pr[NO], t[NO];
#pragma unroll
            for (int m = j; m < NO; ++m) pr[m] = j == 0 ? entry(0, m) : L[tri<NO>(j, m) * 64];
            const double inv = recip(pr[j]);
#pragma unroll
            for (int i = j + 1; i < NO; ++i) t[i] = pr[i] * inv;
#pragma unroll
            for (int i = j + 1; i < NO; ++i) {
#pragma unroll
                for (int m = i; m < NO; ++m) put(i, m, fma(-t[i], pr[m], j == 0 ? entry(i, m) : at(i, m)));
            }
#pragma unroll
            for (int i = j + 1; i < NO; ++i) L[tri<NO>(j, i) * 64] = t[i];
            L[tri<NO>(j, j) * 64] = inv;
            if (WLSQM_STAGE_SOLVE_BARRIERS) __builtin_amdgcn_sched_barrier(0);
        }
        // (ldlt_factor<N2>, one elimination step at a time: steps interleaved by the scheduler keep more of the matrix in flight
        // than the lane has registers for)
#pragma unroll
        for (int j = 0; j < N2; ++j) {
            const double inv = recip(R[tri<N2>(j, j)]);
#pragma unroll
            for (int i = j + 1; i < N2; ++i) {
                const double t = R[tri<N2>(j, i)] * inv;
#pragma unroll
                for (int m = i; m < N2; ++m) R[tri<N2>(i, m)] -= t * R[tri<N2>(j, m)];
                R[tri<N2>(j, i)] = t;
            }
            R[tri<N2>(j, j)] = inv;
            if (WLSQM_STAGE_SOLVE_BARRIERS) __builtin_amdgcn_sched_barrier(0);
        }
        // substitution: forward through the LDS rows, both directions in the register block, backward through the LDS rows
#pragma unroll
        for (int j = 0; j < R0; ++j) {
#pragma unroll
            for (int i = j + 1; i < NO; ++i) sol[i] = fma(-L[tri<NO>(j, i) * 64], sol[j], sol[i]);
        }
        if (WLSQM_STAGE_SOLVE_BARRIERS) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < N2; ++j) {
#pragma unroll
            for (int i = j + 1; i < N2; ++i) sol[R0 + i] -= R[tri<N2>(j, i)] * sol[R0 + j];
            if (WLSQM_STAGE_SOLVE_BARRIERS) __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int j = N2 - 1; j >= 0; --j) {
            double v = sol[R0 + j] * R[tri<N2>(j, j)];
#pragma unroll
            for (int i = j + 1; i < N2; ++i) v -= R[tri<N2>(j, i)] * sol[R0 + i];
            sol[R0 + j] = v;
            if (WLSQM_STAGE_SOLVE_BARRIERS) __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int j = R0 - 1; j >= 0; --j) {
            double v = sol[j] * L[tri<NO>(j, j) * 64];
#pragma unroll
            for (int i = j + 1; i < NO; ++i) v = fma(-L[tri<NO>(j, i) * 64], sol[i], v);
            sol[j] = v;
        }
        if (any_known) {                                              // (for the whole-row store below: a known DOF's own bits)
#pragma unroll
            for (int a = 0; a < NO; ++a) if ((t32 >> a) & 1u) sol[a] = fio[a];
        }
#pragma unroll
        for (int a = 0; a < NO; ++a) rhs[a] = sol[a];
    } else if (!mine1) {
        double M[NE];
        expand_moments<DIM, ORDER>(mu, nu, M, rhs);
        if (known) {
            double val[NO];
#pragma unroll
            for (int a = 0; a < NO; ++a) val[a] = (((known & ~dropped) >> a) & 1ull) ? fio[a] : 0.0;
            eliminate_knowns<NO>(M, rhs, known, val);
#pragma unroll
            for (int a = 0; a < NO; ++a) if (((known & ~dropped) >> a) & 1ull) rhs[a] = val[a];      // (for the whole-row store below: its own bits)
        }
        ldlt_factor<NO>(M);
        double sol[NO];
#pragma unroll
        for (int a = 0; a < NO; ++a) sol[a] = ((known >> a) & 1ull) ? 0.0 : rhs[a];
        ldlt_solve<NO>(M, sol);
#pragma unroll
        for (int a = 0; a < NO; ++a) if (!((known >> a) & 1ull)) rhs[a] = sol[a];
        if constexpr (INVERSE) {
#pragma unroll
            for (int e = 0; e < NE; ++e) Mf[e] = M[e];
        }
    }
    // ---- results.  A full group with contiguous fi rows and no dropped DOF: the wave's 64 rows are ONE run of 64 NO doubles; they go
    // through LDS and leave as whole 16-byte pieces, known DOFs re-written with their own bits (what the reference's Case_get_fi
    // does too, infra.pyx:780-795) — separate 8-byte stores at a row pitch are partial-sector writes.
    const bool whole = nvalid == 64 && p.sfi_j == NO && ((reinterpret_cast<uintptr_t>(p.fi) & 15u) == 0) && (64 * NO) % 2 == 0 &&
                       __all(dropped == 0ull && known != FULL);
    if (whole) {
        __syncthreads();                                              // the last chunk has been read
#pragma unroll
        for (int a = 0; a < NO; ++a) lds[lane * NO + a] = rhs[a];
        __syncthreads();
        d2_* out = reinterpret_cast<d2_*>(p.fi + t0 * NO);
        const d2_* src = reinterpret_cast<const d2_*>(lds);
#pragma unroll
        for (int q = lane; q < 64 * NO / 2; q += 64) __builtin_nontemporal_store(src[q], &out[q]);
    } else if (valid && known != FULL) {
#pragma unroll
        for (int a = 0; a < NO; ++a)
            if (!((known >> a) & 1ull)) fio[a] = rhs[a];
    }
    if constexpr (INVERSE) {
        // ---- the inverse, column by column: a unit vector through the factor that is in registers anyway (NO substitutions instead of
        // one per neighbour); a column of the wave's 64 cases is one contiguous run of 64 NO doubles, through LDS.  Rows and columns of
        // known DOFs are zero.  Tail lanes replay the last case into their own slot of the scratch block (sized in whole groups).
        static_assert(R0 == 0, "the inverse is emitted from a factor in registers");
        double* const blk = p.ws + (long long)grp * (64 * NO * NO);
#pragma unroll 1
        for (int col = 0; col < NO; ++col) {
            double o[NO];
            if (REDUCED && mine1) {
                constexpr int N1 = NO - 1, NE1 = N1 * (N1 + 1) / 2;
                double M1[NE1], sv[N1];
#pragma unroll
                for (int e = 0; e < NE1; ++e) M1[e] = Mf[e];
#pragma unroll
                for (int a = 0; a < N1; ++a) sv[a] = (a + 1 == col) ? 1.0 : 0.0;      // column 0 (the known DOF): zeros
                ldlt_solve<N1>(M1, sv);
                o[0] = 0.0;
#pragma unroll
                for (int a = 0; a < N1; ++a) o[a + 1] = sv[a];
            } else {
                const bool kcol = (known >> col) & 1ull;              // known: zero row and column (the rows are identity rows)
#pragma unroll
                for (int a = 0; a < NO; ++a) o[a] = (a == col && !kcol) ? 1.0 : 0.0;
                ldlt_solve<NO>(Mf, o);
            }
            __syncthreads();                                          // the previous column / the result rows have left the LDS
#pragma unroll
            for (int a = 0; a < NO; ++a) lds[lane * NO + a] = o[a];
            __syncthreads();
            double* const dst = blk + col * (64 * NO);
#pragma unroll
            for (int i = 0; i < NO; ++i) dst[lane + 64 * i] = lds[lane + 64 * i];
        }
    }
    }   // PART == 0
}

__host__ __device__ constexpr int stage_minw(int NO, int PART, bool GATHER) {
    return PART == 5 ? 1 : NO <= 6 ? (GATHER ? WLSQM_STAGE_MINW6G : WLSQM_STAGE_MINW6) : NO <= 10 ? (GATHER ? WLSQM_STAGE_MINW10G : WLSQM_STAGE_MINW10) :
           NO <= 15 ? WLSQM_STAGE_MINW15 : (NO == 35 && !GATHER) ? WLSQM_STAGE_MINW35 : 1;
}

template <int DIM, int ORDER, int PART = 0, bool GATHER = false, int RAG = 0>
__global__ __launch_bounds__(64, stage_minw(ndofs(DIM, ORDER), PART, GATHER)) void fit_stage_kernel(const KParams p, unsigned char* const status) {
    stage_group<DIM, ORDER, PART, GATHER, RAG>(p, blockIdx.x, status);
}

// the groups a RAG = 1 launch marked (none for a batch of full rows: ~2 us of idle waves): workgroup b looks at the groups b, b + gridDim.x, ...,
// 64 of them per step — one status byte per lane
template <int DIM, int ORDER>
__global__ __launch_bounds__(64, stage_minw(ndofs(DIM, ORDER), 0, false)) void fit_stage_ragged_kernel(const KParams p, const long long ngroups, const unsigned char* const status) {
    for (long long g0 = blockIdx.x; g0 < ngroups; g0 += (long long)gridDim.x * 64) {
        const long long g = g0 + (long long)threadIdx.x * gridDim.x;
        unsigned long long todo = __ballot(g < ngroups && status[g] != 0);
        while (todo) {                                                // wave-uniform
            const int q = __ffsll((long long)todo) - 1;
            todo &= todo - 1ull;
            stage_group<DIM, ORDER, 0, false, 2>(p, (unsigned)(g0 + (long long)q * gridDim.x), nullptr);
            __syncthreads();                                          // the LDS is reused by the next group
        }
    }
}

template <int DIM, int ORDER, bool GATHER = false>
static int launch_stage(const KParams& p, hipStream_t stream) {
    const long long groups = (p.ncases + 63) / 64;
    if (groups <= 0) return WLSQM_OK;
    if (groups > 0x7fffffffLL) { set_error("fit_stage: batch too large for one launch"); return WLSQM_EVALUE; }
    constexpr int NO = ndofs(DIM, ORDER);
    // RAGGED batches (round 6): p.ragged = 2 (the host entry points have seen the neighbour counts): the RAGGED copy for every group;
    // p.ragged = 0 (device-resident counts nobody has looked at): the plain kernel marks the waves none of whose cases reaches the row's last
    // chunk and the RAGGED copy runs those behind it — an idle launch for a batch of full rows; p.ragged = 1 (full rows, known): as before.
    constexpr bool RAG_SHAPE = !GATHER && NO <= 20 && !(DIM == 3 && ORDER == 4);
    const char* rg = getenv("WLSQM_HIP_STAGE_RAGGED");                // 0: off (A/B)
    const bool rag_on = RAG_SHAPE && !(rg && rg[0] == '0') && p.max_nk > stage::CH;
    if constexpr (RAG_SHAPE) {
        if (rag_on && p.ragged == 2) {
            hipLaunchKernelGGL((fit_stage_kernel<DIM, ORDER, 0, false, 2>), dim3((unsigned)groups), dim3(64), 0, stream, p, (unsigned char*)nullptr);
            WLSQM_HIP_CHECK(hipGetLastError());
            note_kernel("stage-ragged");
            return WLSQM_OK;
        }
    }
    const bool mark = rag_on && p.ragged == 0;
    CallScratch cs;
    unsigned char* status = nullptr;
    if (mark) {
        const int rc = call_scratch_acquire(&cs, (size_t)groups, stream);
        if (rc != WLSQM_OK) return rc;
        status = static_cast<unsigned char*>(cs.p);
    }
    auto finish = [&](const char* name) {
        hipError_t le = hipGetLastError();
        if constexpr (RAG_SHAPE) {
            if (mark && le == hipSuccess) {
                const long long want = groups < 2048 ? groups : 2048;
                hipLaunchKernelGGL((fit_stage_ragged_kernel<DIM, ORDER>), dim3((unsigned)want), dim3(64), 0, stream, p, groups, status);
                le = hipGetLastError();
            }
        }
        const int rc = mark ? call_scratch_release(&cs, stream) : (int)WLSQM_OK;
        if (le != hipSuccess) return hip_fail(le, "fit_stage_kernel");
        note_kernel(name);
        return rc;
    };
    if constexpr (!GATHER && NO <= 10) {
        // TWO FORMS of the dense systems up to 10 unknowns: two waves per SIMD (PART 0), or one wave that owns its SIMD (PART 5: register-staged
        // chunks; same bits per case).  Rows whose neighbours are NOT sorted by distance take two passes per group, and the second pass finds
        // the rows in L2 only with one wave per SIMD resident: 1M cases with shuffled rows, configs[1] 0.218 -> 0.200 ms, configs[4] 0.406 ->
        // 0.373 — while sorted rows lose 5-9 % there.  ROUND 6: the form is a function of the ARGUMENTS — the caller's word about its rows
        // (wlsqm_hip_set_order_hint; default: sorted, what every k-nearest-neighbour search returns) — not of what earlier launches on the
        // stream reported (round 5: host-mapped bytes written by sampled groups; the same call could be timed at 0.20 or 0.22 ms depending on
        // its predecessor).  WLSQM_HIP_STAGE_FORM=two / one forces a form (A/B, tests).
        const char* e = getenv("WLSQM_HIP_STAGE_FORM");
        const bool own_simd = (e && (e[0] == 't' || e[0] == 'o')) ? e[0] == 'o' : p.rows_sorted == 0;
        if (mark) {
            if (own_simd) hipLaunchKernelGGL((fit_stage_kernel<DIM, ORDER, 5, false, 1>), dim3((unsigned)groups), dim3(64), 0, stream, p, status);
            else hipLaunchKernelGGL((fit_stage_kernel<DIM, ORDER, 0, false, 1>), dim3((unsigned)groups), dim3(64), 0, stream, p, status);
        } else {
            if (own_simd) hipLaunchKernelGGL((fit_stage_kernel<DIM, ORDER, 5, false, 0>), dim3((unsigned)groups), dim3(64), 0, stream, p, status);
            else hipLaunchKernelGGL((fit_stage_kernel<DIM, ORDER, 0, false, 0>), dim3((unsigned)groups), dim3(64), 0, stream, p, status);
        }
        return finish(own_simd ? "stage-own" : "stage");
    }
    if constexpr (RAG_SHAPE) {
        if (mark) {
            hipLaunchKernelGGL((fit_stage_kernel<DIM, ORDER, 0, GATHER, 1>), dim3((unsigned)groups), dim3(64), 0, stream, p, status);
            return finish("stage");
        }
    }
    hipLaunchKernelGGL((fit_stage_kernel<DIM, ORDER, 0, GATHER, 0>), dim3((unsigned)groups), dim3(64), 0, stream, p, status);
    return finish(GATHER ? "stage-gather" : "stage");
}

int launch_quad_solve(const KParams& p, hipStream_t stream);          // fit_quad.hip

// The basic fit of a dense contiguous 2D order-4 batch that ALSO leaves every case's inverse normal matrix at inv[group][column][case][row]
// (first kernel of the sensitivities' path, fit_sens.hip; round 5: instead of the four-lanes-per-case tile kernel + moment_solve_kernel pair).
// *handled = false: the input is not the staged kernel's (strides, alignment, odd K): the caller keeps the pair.
int launch_fit_stage_inverse(int dimension, int order, const KParams& p, long long K, double* inv, hipStream_t stream, bool* handled) {
    *handled = false;
    const char* e = getenv("WLSQM_HIP_STAGE_INVERSE");
    if ((e && e[0] == '0') || dimension != 2 || order != 4) return WLSQM_OK;
    if (p.hoods || p.case_index || !p.xk || !p.fk || K < 8 || K % 2 != 0 || K > 65536) return WLSQM_OK;
    if (p.sxk_k != dimension || p.sxk_j != K * dimension || p.sfk_k != 1 || p.sfk_j != K) return WLSQM_OK;
    if ((reinterpret_cast<uintptr_t>(p.xk) | reinterpret_cast<uintptr_t>(p.fk)) & 15u) return WLSQM_OK;
    const long long groups = (p.ncases + 63) / 64;
    if (groups <= 0 || groups > 0x7fffffffLL) return WLSQM_OK;
    KParams q = p;
    q.ws = inv; q.do_sens = 0; q.sens = nullptr; q.iterative = 0;
    hipLaunchKernelGGL((fit_stage_kernel<2, 4, 4, false>), dim3((unsigned)groups), dim3(64), 0, stream, q, (unsigned char*)nullptr);
    WLSQM_HIP_CHECK(hipGetLastError());
    *handled = true;
    return WLSQM_OK;
}

// 3D order 4: two moment launches (half of the 200 sums each) and the four-lanes-per-case solve, in slices that bound the workspace
// (1 600 bytes per case, stream-ordered) at 1.7 GB.
template <bool GATHER = false>
static int launch_stage34(const KParams& p, hipStream_t stream) {
    long long SLICE = 1LL << 20;
    if (const char* e = getenv("WLSQM_HIP_QUAD_SLICE")) {             // (tests: small slices; whole 64-case groups)
        const long long v = atoll(e);
        if (v >= 64) SLICE = v / 64 * 64;
    }
    for (long long j0 = 0; j0 < p.ncases; j0 += SLICE) {
        KParams q = slice_cases(p, j0, p.ncases - j0 < SLICE ? p.ncases - j0 : SLICE);
        const long long groups = (q.ncases + 63) / 64;
        double* ws = nullptr;
        int rc = scratch_alloc_async(reinterpret_cast<void**>(&ws), (size_t)groups * 200 * 64 * sizeof(double), stream);
        if (rc != WLSQM_OK) return rc;
        q.ws = ws; q.ws_stride = 0;
        // (round 5: both halves in ONE launch — the second pass re-staging the rows from L2 — was built and dropped: the compiler's register
        // allocation of the fused kernel puts 380-780 accumulation-register moves into every 8-neighbour chunk where the two kernels have
        // ~60, and 848 B of scratch)
        hipLaunchKernelGGL((fit_stage_kernel<3, 4, 1, GATHER>), dim3((unsigned)groups), dim3(64), 0, stream, q, (unsigned char*)nullptr);
        hipError_t le = hipGetLastError();                            // (a failed moment launch must not run the solve on an uninitialised workspace)
        if (le == hipSuccess) {
            hipLaunchKernelGGL((fit_stage_kernel<3, 4, 2, GATHER>), dim3((unsigned)groups), dim3(64), 0, stream, q, (unsigned char*)nullptr);
            le = hipGetLastError();
        }
        rc = le == hipSuccess ? launch_quad_solve(q, stream) : hip_fail(le, "fit_stage_kernel<3, 4>");
        const int rc2 = scratch_free_async(ws, stream);
        if (rc != WLSQM_OK) return rc;
        if (rc2 != WLSQM_OK) return rc2;
    }
    note_kernel(GATHER ? "quad-gather" : "quad");
    return WLSQM_OK;
}

// Basic fits of dense contiguous batches with an even neighbour count, for the shapes listed below;
// everything else keeps its kernels.  WLSQM_HIP_STAGE=0 disables it, =all sends every covered shape here (A/B).
int launch_fit_stage(int dimension, int order, const KParams& p, long long K, hipStream_t stream, bool* handled) {
    *handled = false;
    const char* e = getenv("WLSQM_HIP_STAGE");
    if (e && e[0] == '0') return WLSQM_OK;
    const char* off = getenv("WLSQM_HIP_DISABLE_TILE");
    if (off && off[0] == '1') return WLSQM_OK;
    if (p.do_sens || p.iterative || p.case_index) return WLSQM_OK;
    if (p.hoods) {
        // index-based input: the gathering form (WLSQM_HIP_STAGE_GATHER=0 disables it, =all sends every covered shape here: A/B).
        // tools/time_cloud.py, 1M cases, points in Morton order / in Halton order (every gather a miss), against the gathering tile /
        // ring / wave-per-case kernels: 2D order 4 at 64 neighbours 0.42 / 0.44 against 0.57 / 0.69 ms, 3D order 2 at 40 0.28 / 0.40
        // against 0.37 / 0.41, 2D order 2 at 24 0.094 / 0.124 against 0.124 / 0.132, 2D order 3 at 30 0.21 / 0.21 against 0.24 / 0.26,
        // 3D order 3 at 40 (400k) 0.30 against 1.62, 3D order 4 1.18 against 4.13 (profiles/r04y_stage_gather.txt)
        const char* g = getenv("WLSQM_HIP_STAGE_GATHER");
        if ((g && g[0] == '0') || !p.S || !p.F || K < 8 || K > 65536) return WLSQM_OK;
        const bool gall = g && g[0] == 'a';
#define GCASE(D, O, COND) if (dimension == D && order == O && (gall || (COND))) { *handled = true; return launch_stage<D, O, true>(p, stream); }
        GCASE(3, 3, true)
        if (dimension == 3 && order == 4) { *handled = true; return launch_stage34<true>(p, stream); }
        // (the small systems at small neighbour counts keep the gathering tile kernels where those measured faster: 2D order 2 at
        // 12 / 20 / 28 neighbours 0.113 / 0.135 / 0.162 against 0.079 / 0.110 / 0.147 ms, 2D order 3 at 20 / 28 0.170 / 0.205 against
        // 0.145 / 0.187, 3D order 2 at 20 0.188 against 0.179 — the partial first chunk costs more than it carries)
        GCASE(2, 4, true)
        GCASE(3, 2, K >= 24)
        GCASE(2, 3, K >= 30)
        GCASE(2, 2, K >= 32 || K % 8 == 0)
#undef GCASE
        return WLSQM_OK;
    }
    if (!p.xk || !p.fk) return WLSQM_OK;
    if (K < 8 || K % 2 != 0 || K > 65536) return WLSQM_OK;          // (row bytes and the tile's offsets are 32-bit)
    if (p.sxk_k != dimension || p.sxk_j != K * dimension || p.sfk_k != 1 || p.sfk_j != K) return WLSQM_OK;
    if ((reinterpret_cast<uintptr_t>(p.xk) | reinterpret_cast<uintptr_t>(p.fk)) & 15u) return WLSQM_OK;
    const bool all = e && e[0] == 'a';
#define SCASE(D, O, COND) if (dimension == D && order == O && (all || (COND))) { *handled = true; return launch_stage<D, O>(p, stream); }
    // (tools/sweep_stage.py, profiles/r04l_sweep_stage.txt: 400k cases, against the fixed-K tile / ring / moment kernels, neighbour
    // counts 8 .. 128: 2D order 4 1.05-2.3x, 3D order 2 1.05-1.9x, 2D order 3 0.98-1.56x, 2D order 2 0.94-0.99x up to 24 neighbours
    // and 0.95-1.41x from 32 on)
    SCASE(2, 4, true)
    SCASE(3, 2, true)
    SCASE(2, 3, true)
    SCASE(2, 2, K >= 32)
    SCASE(3, 3, true)
    if (dimension == 3 && order == 4) { *handled = true; return launch_stage34<false>(p, stream); }
#undef SCASE
    return WLSQM_OK;
}

}  // namespace wlsqm
