#pragma once
// wlsqm_ring.hpp — ONE-kernel fit of the 15-unknown systems (2D order 4; BASELINE configs[2], "C3"): LDS-DMA ring + moments +
// register-parked lane-per-case solve.  Replaces the two-kernel moment path (tile pass -> 480 B/case workspace ->
// moment_solve_kernel: 960 B/case of extra HBM traffic, profiles/traffic_C3.json) for dense contiguous input.
//
// Reference arithmetic (file:line in /root/reference): make_c_2D impl.pyx:286-432, weights infra.pyx:668-702, make_A
// impl.pyx:566-602, RHS + knowns elimination impl.pyx:768-823, factor/solve lapackdrivers.pyx:1628-1665 (here: moments +
// unpivoted LDL^T, wlsqm_kernels.hpp / wlsqm_moments.hpp).
//
// Shape: one wave per workgroup, __launch_bounds__(64, 1) (the 120-entry matrix of the solve needs > 256 registers, so the
// kernel owns its SIMD and must hide its own memory latency):
//   * a tile = 16 consecutive cases, 4 lanes per case (lane = h * 16 + c), each lane sums KC / 4 neighbours into the 45 + 15
//     distinct moments (outer-product form), the 4 partial sums meet in a two-step xor butterfly;
//   * the NEXT tile's xk block is already on its way while the current one is computed: `global_load_lds_dwordx4` into the
//     other half of a two-tile LDS ring (no VGPR destination, no ds_write).  The DMA writes LDS lane-linearly, so rows cannot be
//     padded; bank conflicts are removed on the SOURCE side: row r is stored rotated by rot(r) = r (1 - K) mod 16 chunks, so
//     that lane (c, h) finds neighbour k of its case at chunk (k + rot(c)) mod K and 16 consecutive lanes hit 16 different
//     4-bank groups (ds_read_b128).  The next tile's fk values (16-byte loads of the lane's own share) and per-case scalars
//     travel through a second register set;
//   * after the butterfly all 4 lanes of a case hold the same sums; lane h keeps those of the tile with (iteration mod 4) == h
//     in a parked register set.  After 4 tiles the 64 lanes hold 64 DIFFERENT cases and the whole wave expands the matrix,
//     eliminates knowns, factors and substitutes — the solve runs once per 64 cases on all lanes instead of after every tile
//     on a quarter of them, and nothing but xk, fk, xi, the scalars and fi crosses HBM.
// A workgroup owns a CONTIGUOUS run of tiles (so a solve stores 64 consecutive fi rows).
#include <cstdlib>

#include <atomic>

#include "wlsqm_internal.hpp"
#include "wlsqm_kernels.hpp"
#include "wlsqm_moments.hpp"

namespace wlsqm {

typedef double rd2_ __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) void* ring_lds_ptr_t;
typedef const __attribute__((address_space(1))) void* ring_glb_ptr_t;

typedef unsigned ru2_ __attribute__((ext_vector_type(2)));
// v_permlane16_swap / v_permlane32_swap on a double (both halves): afterwards, in the lanes of the EVEN 16-lane rows (lower 32
// lanes) x is unchanged and y holds the partner lane's x; in the ODD rows (upper 32 lanes) y is unchanged and x holds the
// partner's y (tools/ubench/permlane_swap.hip) — so x + y is "x summed over the pair" in one half of the lanes and "y summed
// over the pair" in the other: one step of a reduce-SCATTER without selects, LDS traffic or copies.
__device__ __forceinline__ void swap16(double& x, double& y) {
    const ru2_ lo = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(x), (unsigned)__double2loint(y), false, false);
    const ru2_ hi = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(x), (unsigned)__double2hiint(y), false, false);
    x = __hiloint2double((int)hi.x, (int)lo.x); y = __hiloint2double((int)hi.y, (int)lo.y);
}
__device__ __forceinline__ void swap32(double& x, double& y) {
    const ru2_ lo = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(x), (unsigned)__double2loint(y), false, false);
    const ru2_ hi = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(x), (unsigned)__double2hiint(y), false, false);
    x = __hiloint2double((int)hi.x, (int)lo.x); y = __hiloint2double((int)hi.y, (int)lo.y);
}

#ifndef WLSQM_RING_IBUF_PAD
#define WLSQM_RING_IBUF_PAD 4    // ints of padding behind every pair of index rows (0: A/B)
#endif
#ifndef WLSQM_RING_FI_RUN
#define WLSQM_RING_FI_RUN 1     // 3D branch-free solve: the 64 fi rows leave as one run of non-temporal 16-byte pieces through LDS instead of 8-byte pieces at an 80-byte pitch: configs[4] at 1M, interleaved, 0.3035-0.3086 against 0.3113-0.3159 ms (profiles/r03i_ab_ring_c5.txt)
#endif
#ifndef WLSQM_RING_FI_NT
#define WLSQM_RING_FI_NT 0      // non-temporal for EVERY fi store of the ring: configs[2] unchanged (its rows already leave as a run), configs[4] 0.31 -> 0.46 ms with its 8-byte pieces at an 80-byte pitch — non-temporal pays for whole lines only
#endif
template <class T>
__device__ __forceinline__ void ring_store(T* dst, const T v) {
#if defined(WLSQM_RING_NO_STORE)      // experiment only: what the kernel costs without its results leaving (wrong output)
    if (reinterpret_cast<uintptr_t>(dst) == 8) *dst = v;
#elif WLSQM_RING_FI_NT
    __builtin_nontemporal_store(v, dst);
#else
    *dst = v;
#endif
}

template <int DIM, int K> struct RingGeom {
    static constexpr int WV = 64, TC = 16, LPC = 4;
    static constexpr int KC = (K + 7) / 8 * 8;              // slots the four shares cover (even share each); slots >= K are masked
    static constexpr int KPL = KC / LPC;
    static constexpr int ROWB = K * DIM * 8;                 // bytes of a row (a multiple of 16: K * DIM even)
    static constexpr int PARTS = (ROWB + 1023) / 1024;       // DMA instructions per row (64 lanes x 16 B = 1 KiB each)
    static constexpr int NI = TC * PARTS;                    // DMA instructions per tile
    // padded row stride in doubles, even (16-byte DMA destinations).  2D: == 2 (mod 4), so that 16 consecutive lanes reading one
    // neighbour of 16 consecutive rows with ds_read_b128 hit 16 different 4-bank groups; 3D (ds_read_b64 x 3): == 2 (mod 32)
    // spreads the 16 rows over the even bank pairs (the second share of a case lands 2-way on some of them: accepted).
    // Rows cover KC slots (a padded share reads up to KC - K slots past K).
    static constexpr int RS = (DIM == 2) ? ((2 * KC + 1) / 4) * 4 + 2 : ((KC * DIM + 31) / 32) * 32 + 2;
    static constexpr int SLOT = TC * RS + 2 * WV;            // doubles per ring slot (+ slack: the last row's DMA writes whole 16-B lanes only)
    static constexpr size_t LDS_BYTES = sizeof(double) * 2 * SLOT;
    // 3D: 64 fi rows of the branch-free solve leave as one contiguous run of 16-byte pieces through a staging area behind the ring
    // (that solve runs while BOTH ring slots are in use); 35.3 + 5 KB = 39.5 KB: still four workgroups per CU
    static constexpr bool FI_STAGE = (DIM == 3);
    static constexpr int FI_STAGE_NO = 10;                   // unknowns of the 3D instantiation (order 2)
    static_assert((K * DIM) % 2 == 0 && K % 2 == 0 && K >= 8, "rows must be multiples of 16 bytes");
    static_assert(RS >= DIM * KC && RS % 2 == 0, "row stride");
};

// GATHER: index-based input.  The neighbours of case j are rows hoods[j, k] of the point table S (16-byte rows: 2D) with values
// F: the SAME ring, filled by the same DMA instruction with PER-LANE global addresses S + 16 hoods[r, lane] instead of a uniform
// row base plus lane * 16 — row r of the slot ends up exactly as in the dense case, and nothing after the prefetch differs.
// The indices of a tile (TC * K int32, contiguous) travel one tile further ahead, as a lane-linear DMA copy into a 4 KB LDS
// buffer: at the top of tile t the indices of tile t + 1 have landed; they are read back (row r for the coordinate DMAs, the
// lane's own share for the values, which are gathered into the second register set like the dense values), and the indices of tile
// t + 2 are requested into the same buffer behind those reads.  Slots k >= nk of a row are never dereferenced (the padding of a
// ragged row may hold anything): they fetch point 0 instead; the nk of the tile ahead sits in a register a tile early for that.
template <int DIM, int ORDER, int K, int UNR, int MINW, bool GATHER = false>
__global__ __launch_bounds__(64, MINW) void fit_ring_kernel(const KParams p, const long long ntiles, const int tiles_per_wg) {
    using G = RingGeom<DIM, K>;
    static_assert(!GATHER || (DIM == 2 && G::PARTS == 1), "index-based ring: 16-byte point rows, one DMA per row");
    constexpr int WV = 64, TC = G::TC, KPL = G::KPL, RS = G::RS;
    constexpr int NO = ndofs(DIM, ORDER), NE = NO * (NO + 1) / 2, NM = mom_count<DIM>(2 * ORDER), NN = mom_count<DIM>(ORDER);
    static_assert(NN == NO, "one right-hand-side moment per DOF");
    extern __shared__ __attribute__((aligned(16))) double lds[];          // [2][SLOT]

    const int lane = threadIdx.x, c = lane % TC, h = lane / TC, k0 = h * KPL;
    // 3D: the four lanes of a case take INTERLEAVED neighbours (k = 4 kk + h) instead of contiguous quarters.  A point is three
    // doubles and a quarter 3 KPL = 30 of them: with contiguous quarters the ds_read_b64 of lanes (c, h) and (c, h + 1) — same 32-lane
    // group, addresses c RS + 30 h + ... with RS == 2 (mod 32) — land on the same bank pair for every c (both offsets are even in
    // 8-byte units: re-pitching the rows cannot separate them), a 2-way conflict on every coordinate read: SQ_LDS_BANK_CONFLICT 0.46
    // of the LDS-active cycles of configs[4] since round 2.  Interleaved, the offsets are 2 c + 3 h: 32 different bank pairs.
#ifndef WLSQM_RING_INTERLEAVE3D
#define WLSQM_RING_INTERLEAVE3D 1
#endif
    constexpr bool ILV = (DIM == 3) && (WLSQM_RING_INTERLEAVE3D != 0) && (G::KC == K) && !GATHER;
    auto slot_of = [&](int kk) { return ILV ? G::LPC * kk + h : k0 + kk; };      // neighbour slot of this lane's kk-th term

    struct Meta { int nk, wm; long long kn; double xi[DIM]; };
    Meta nxt;
    double fnext[KPL];
    int* const ibuf = reinterpret_cast<int*>(lds + 2 * G::SLOT);          // GATHER: indices of the tile ahead, rows in PAIRS at a pitch of 2 K + 4
    // (unpadded, the sixteen rows of a 64-slot tile start in the same bank: the lanes' reads of their own shares were 16-way conflicts —
    // SQ_LDS_BANK_CONFLICT 67 % of the LDS-active cycles of the kernel, profiles/r03k_pmc_new_kernels.txt; a pair of rows is the smallest
    // unit a 16-byte DMA piece never straddles for every even K)
    constexpr int IPP = 2 * K + WLSQM_RING_IBUF_PAD;                      // ints per row pair
    auto irow = [&](int r) { return (r >> 1) * IPP + (r & 1) * K; };      // first int of row r
    int nk_ahead = 0;                                                     // GATHER: nk of this lane's case in the tile whose indices are in ibuf

    // GATHER: request the indices of `tile` (lane-linear copy of TC * K int32 into ibuf) and the nk of its cases
    auto prefetch_indices = [&](long long tile) {
        const long long j0 = tile * TC;
        const int nvalid = (p.ncases - j0 < TC) ? (int)(p.ncases - j0) : TC;
        const char* hb = reinterpret_cast<const char*>(p.hoods + j0 * (long long)K);
        const int end = nvalid * K * 4;                                   // bytes of the tile's valid rows
        const int lim = end - 16;                                         // tail tile: pieces at or behind `end` replay the last 16 valid bytes (never dereferenced: rows >= nvalid count as nk = 0)
        static_assert(2 * K * 4 <= 1024, "a row pair per DMA instruction");
        if (lane * 16 < 2 * K * 4) {
#pragma unroll
            for (int pr = 0; pr < TC / 2; ++pr) {
                const int off = pr * 2 * K * 4 + lane * 16;
                __builtin_amdgcn_global_load_lds((ring_glb_ptr_t)(hb + (off < lim ? off : lim)),
                                                 (ring_lds_ptr_t)(reinterpret_cast<char*>(ibuf) + pr * IPP * 4), 16, 0, 0);
            }
        }
        // The ONE piece that straddles `end` (K == 2 mod 4 and an odd number of valid rows: end == 8 mod 16).  The DMA lands at the
        // lane's own LDS position whatever its source, so a clamped source put bytes [end - 16, end) where [end - 8, end + 8) belong:
        // the last two indices of the last valid row became copies of the two before them (round 3's tail bug: wrong neighbours for
        // the last case of a launch whenever nk >= K - 1).  Its valid half is fetched by a plain 8-byte load instead and written
        // behind the DMAs: the load's data arrives after every DMA issued before it (vmcnt is in order), so the write lands on top.
        if constexpr (K % 4 == 2) {
            if (end & 8) {                                                // wave-uniform: tail tiles only
                const int pr = (nvalid - 1) >> 1;                         // the pair whose first row is the last valid one
                asm volatile("" ::: "memory");
                if (lane == 0) {
                    const int* tail = reinterpret_cast<const int*>(hb + end - 8);
                    const int v0 = tail[0], v1 = tail[1];
                    __builtin_amdgcn_s_waitcnt(0x0f70);                   // vmcnt(0): the DMAs above and the two loads
                    asm volatile("" ::: "memory");
                    ibuf[pr * IPP + K - 2] = v0;
                    ibuf[pr * IPP + K - 1] = v1;
                }
                asm volatile("" ::: "memory");
            }
        }
        const int cc = c < nvalid ? c : nvalid - 1;
        nk_ahead = p.nk[(j0 + cc) * p.snk];
    };

    // Everything of tile `tile` that can be requested ahead of time.  One DMA instruction moves (at most) 64 neighbours = 1 KiB
    // of ONE row: its LDS base is the row's padded position (wave-uniform, M0), its global address a wave-uniform base plus
    // lane * 16 — no per-lane address arithmetic at all.  (A first version filled the ring lane-linearly with source-side
    // rotated rows; its sixteen hoisted 64-bit per-lane addresses were spilled next to the 120-entry solve, and the scratch
    // reloads BETWEEN the DMAs waited — vmcnt is in order — for the DMA issued just before: 0.73 ms instead of 0.56.)
    auto prefetch = [&](long long tile, int slot, bool more) {
        const long long j0 = tile * TC;
        const int nvalid = (p.ncases - j0 < TC) ? (int)(p.ncases - j0) : TC;      // wave-uniform
        double* dst = lds + slot * G::SLOT;
        if constexpr (GATHER) {
            // (called right behind the barrier at the top of a tile: ibuf holds this tile's indices, nk_ahead its nk.)  Every read of
            // ibuf comes first: an LDS read issued behind a DMA waits for that DMA (the compiler cannot tell the ring from ibuf)
            // (tail tile: the rows behind the last valid case hold replayed bytes of its row, padding included — nothing of them is
            // dereferenced: their nk counts as 0)
            const int nkl = c < nvalid ? min(nk_ahead, K) : 0;
            int ixr[TC], ixf[KPL];
#pragma unroll
            for (int r = 0; r < TC; ++r) ixr[r] = ibuf[irow(r) + (lane < K ? lane : K - 1)];
#pragma unroll
            for (int i = 0; i < KPL; ++i) ixf[i] = ibuf[irow(c) + ((k0 + i < K) ? k0 + i : K - 1)];   // a padded share replays the last slot (masked in the loop)
            __builtin_amdgcn_s_waitcnt(0xc07f);                                   // lgkmcnt(0): the reads have returned
            asm volatile("" ::: "memory");
            const char* Sb = reinterpret_cast<const char*>(p.S);
            if (lane < K) {
#pragma unroll
                for (int r = 0; r < TC; ++r) {
                    const int nkr = __builtin_amdgcn_readlane(nkl, r);            // lane r (h = 0) holds case r's nk
                    const int ix = lane < nkr ? ixr[r] : 0;
                    __builtin_amdgcn_global_load_lds((ring_glb_ptr_t)(Sb + (size_t)(unsigned)ix * 16u), (ring_lds_ptr_t)(dst + r * RS), 16, 0, 0);
                }
            }
            const int cc = c < nvalid ? c : nvalid - 1;
            const long long jc = j0 + cc;
            nxt.nk = nk_ahead; nxt.wm = p.wm[jc * p.swm]; nxt.kn = p.knowns[jc * p.sknowns];
            const long long pj = own_point(p, jc);
#pragma unroll
            for (int m = 0; m < DIM; ++m) nxt.xi[m] = p.S[pj * DIM + m];
#pragma unroll
            for (int i = 0; i < KPL; ++i) fnext[i] = p.F[(unsigned)((k0 + i < nkl) ? ixf[i] : 0)];
            // the next tile's indices overwrite ibuf: behind the reads above
            if (more) prefetch_indices(tile + 1);
            return;
        }
        const char* xbase = reinterpret_cast<const char*>(p.xk + j0 * (long long)(K * DIM));
        const unsigned lane16 = (unsigned)lane * 16u;
        // one region per 64-neighbour part, so that the active lanes of every DMA are a PREFIX of the wave (with the parts
        // interleaved per row the compiler threads the repeated lane condition into two paths and issues the full-row DMAs
        // once for the low and once for the high lanes: wrong rows for K > 64, measured)
#ifndef WLSQM_RING_IMM_OFFSETS
#define WLSQM_RING_IMM_OFFSETS 1
#endif
        bool dma_done = false;
#if defined(__HIP_DEVICE_COMPILE__)          // (the host pass drops the kernel's stub when it sees the non-zero immediate: device pass only)
        if constexpr (WLSQM_RING_IMM_OFFSETS != 0 && G::PARTS == 1 && 3 * G::ROWB < 4096) {
            if (nvalid == TC) {
                // full tile: four per-lane base addresses (rows 0, 4, 8, 12) and the instruction's immediate offset for the three rows
                // behind each (0 .. 3 ROWB < 4 KiB) instead of sixteen 64-bit multiply-adds; the immediate is added to the LDS address
                // as well, so the LDS base handed to the instruction is the row's position MINUS the offset
                if ((int)lane16 < G::ROWB) {
#pragma unroll
                    for (int g4 = 0; g4 < TC; g4 += 4) {
                        const char* gb = xbase + (size_t)g4 * G::ROWB + lane16;
                        char* lb = reinterpret_cast<char*>(dst + g4 * RS);
                        __builtin_amdgcn_global_load_lds((ring_glb_ptr_t)gb, (ring_lds_ptr_t)(lb), 16, 0, 0);
                        __builtin_amdgcn_global_load_lds((ring_glb_ptr_t)gb, (ring_lds_ptr_t)(lb + 1 * (RS * 8 - G::ROWB)), 16, 1 * G::ROWB, 0);
                        __builtin_amdgcn_global_load_lds((ring_glb_ptr_t)gb, (ring_lds_ptr_t)(lb + 2 * (RS * 8 - G::ROWB)), 16, 2 * G::ROWB, 0);
                        __builtin_amdgcn_global_load_lds((ring_glb_ptr_t)gb, (ring_lds_ptr_t)(lb + 3 * (RS * 8 - G::ROWB)), 16, 3 * G::ROWB, 0);
                    }
                }
                dma_done = true;
            }
        }
#endif
        if (!dma_done)
#pragma unroll
        for (int pp = 0; pp < G::PARTS; ++pp) {
            if ((pp + 1) * 1024 <= G::ROWB || pp * 1024 + (int)lane16 < G::ROWB) {
#pragma unroll
                for (int r = 0; r < TC; ++r) {
                    const int rs = r < nvalid ? r : nvalid - 1;                    // tail tile: replay the last valid row
                    const char* src = xbase + (size_t)rs * G::ROWB + (size_t)pp * 1024u;   // uniform
                    __builtin_amdgcn_global_load_lds((ring_glb_ptr_t)(src + lane16), (ring_lds_ptr_t)(dst + r * RS + pp * 128),
                                                     16, 0, 0);
                }
            }
        }
        const int cc = c < nvalid ? c : nvalid - 1;                                // tail tile: replay the last valid case
        const long long jc = j0 + cc;
        nxt.nk = p.nk[jc * p.snk]; nxt.wm = p.wm[jc * p.swm]; nxt.kn = p.knowns[jc * p.sknowns];
#pragma unroll
        for (int m = 0; m < DIM; ++m) nxt.xi[m] = p.xi[jc * p.sxi_j + m];
        const char* fbase = reinterpret_cast<const char*>(p.fk + j0 * (long long)K);
        if constexpr (ILV) {
#pragma unroll
            for (int i = 0; i < KPL; ++i)
                fnext[i] = *reinterpret_cast<const double*>(fbase + (unsigned)(cc * K + h) * 8u + (unsigned)(i * G::LPC) * 8u);
        } else {
#pragma unroll
        for (int i = 0; i < KPL / 2; ++i) {
            // a padded share's slots beyond the row replay the row's last pair (masked in the loop)
            const int kq = (G::KC == K) ? k0 + 2 * i : ((k0 + 2 * i < K) ? k0 + 2 * i : K - 2);
            const unsigned foff = (unsigned)(cc * K + kq) * 8u;
            const rd2_ v = *reinterpret_cast<const rd2_*>(fbase + foff);
            fnext[2 * i] = v.x; fnext[2 * i + 1] = v.y;
        }
        }
    };

    // Parked sums.  The four partial sums of a case meet in a reduce-scatter (two swap steps, below): afterwards lane (c, h)
    // holds the COMPLETE sums of one quarter of the 60 moments of case c (quarter QB[h] of the list mu[0..44], nu[0..14]) and
    // keeps them in PQ[it % 4].  After four tiles a 4 x 4 transpose between the four lanes of a case (two more swap steps, no
    // arithmetic) leaves lane (c, h) with all 60 moments of case c of the tile of iteration it % 4 == h: the 64 lanes hold 64
    // different cases and the whole wave solves.
    constexpr int NV = (NM + NO + 3) / 4 * 4, NQ = NV / 4;          // the list (mu, nu), padded with zeros to four equal quarters
    double PQ[4][NQ];
    long long jp = 0;
    unsigned long long knownp = 0, droppedp = 0;
    bool havep = false;
    constexpr unsigned long long FULL = (1ull << NO) - 1ull;

    auto solve_parked = [&](double* dead_slot) {
        // 4 x 4 transpose of the quarters (lanes x parked sets): rows of 16 lanes first, then the halves of the wave.  Lane h parked, for
        // tile t, quarter QB[h] = {0, 2, 1, 3}[h] in PQ[t].  swap16(PQ[t], PQ[t+1]), t = 0, 2: even-row lanes now hold two quarters of
        // tile t in PQ[t], PQ[t+1], odd-row lanes two quarters of tile t + 1; swap32(PQ[t], PQ[t+2]), t = 0, 1: lower lanes get the
        // other two quarters of their tile from the upper half in PQ[2], PQ[3], upper lanes theirs in PQ[0], PQ[1].  Lane h ends with
        // tile h complete, and in EVERY lane PQ[r] holds quarter {0, 2, 1, 3}[r] (checked for all four lanes).
#pragma unroll
        for (int e = 0; e < NQ; ++e) { swap16(PQ[0][e], PQ[1][e]); swap16(PQ[2][e], PQ[3][e]); }
#pragma unroll
        for (int e = 0; e < NQ; ++e) { swap32(PQ[0][e], PQ[2][e]); swap32(PQ[1][e], PQ[3][e]); }
        auto entry = [&](int i) -> double {                      // moment i of the list (mu, nu): quarter i / NQ sits in PQ[QRinv]
            const int qtr = i / NQ, e = i - qtr * NQ;
            const int r = (qtr == 0) ? 0 : (qtr == 1) ? 2 : (qtr == 2) ? 1 : 3;
            return PQ[r][e];
        };
        // Every case of the wave has exactly the function value known (knowns = b?_F, the reference's default and BASELINE
        // configs[2]): the (NO - 1) x (NO - 1) system is expanded and factored directly — 105 + 14 instead of 120 + 15 entries for 15
        // DOFs, which is what lets the matrix stay in the architectural registers (the full system overflows them by a few entries
        // and the compiler then shuttles ~1 400 values per solve through the accumulation registers).  The same linear system as the
        // generic path's (whose first elimination step is the identity row), not the same roundings: see `mine1`.
        // Which of the two forms a case gets depends on ITS mask alone (round 4): a case with exactly F known takes the reduced system
        // also when other cases of its wave do not — the two forms round differently (3e-9 relative on the 14 x 14 systems: different
        // instantiations of the factorisation, contracted differently), and a case's bits must not depend on its wave-mates
        // (tools/check_tile_mates.py).  A mixed wave runs both forms one after the other; a uniform wave one, as before.
        const bool mine1 = NO >= 3 && havep && knownp == 1ull && droppedp == 0ull;
        if constexpr (NO >= 3) {
            const bool all1 = __all(!havep || mine1);            // (asked outside the divergent region below: every lane votes)
            const bool all_have = __all(havep);
            {
                if (mine1) {
                    constexpr int N1 = NO - 1, NE1 = N1 * (N1 + 1) / 2;
                    double* fio = p.fi + jp * p.sfi_j;
                    const double v0 = fio[0];
                    double M1[NE1], r1[N1];
#pragma unroll
                    for (int a = 1; a < NO; ++a) {
                        const int pa = Mono<DIM>::P[a], qa = Mono<DIM>::Q[a], ra = Mono<DIM>::R[a];
                        r1[a - 1] = entry(NM + mom_index<DIM>(pa, qa, ra)) * (mom_inv_fact(pa) * mom_inv_fact(qa) * mom_inv_fact(ra));
                    }
#pragma unroll
                    for (int i = 0; i < NM; ++i) {
                        const double m = entry(i);
#pragma unroll
                        for (int a = 1; a < NO; ++a) {
                            const int pa = Mono<DIM>::P[a], qa = Mono<DIM>::Q[a], ra = Mono<DIM>::R[a];
                            const double fa = mom_inv_fact(pa) * mom_inv_fact(qa) * mom_inv_fact(ra);
                            if (mom_index<DIM>(pa, qa, ra) == i) r1[a - 1] -= (m * (1.0 * fa)) * v0;      // M[0, a] * fi[0] (impl.pyx:815-818)
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
                    // Results.  With all 64 cases valid and contiguous fi rows the wave's 64 rows are ONE run of 64 NO doubles
                    // (its cases are consecutive): they go through the ring slot that has just been consumed and leave as whole
                    // 16-byte pieces, the known value re-written with its own bits — what the reference's Case_get_fi does too
                    // (infra.pyx:780-795 copies all `no` doubles back).  Separate 8-byte stores at a 120-byte pitch with the
                    // known DOF left out made every row a partial-sector write: 200 instead of 120 MB written and 150 MB of
                    // extra sector fetches per 1M cases (profiles/r02c_C3_pmc_summary.json).
                    const bool whole_rows = dead_slot != nullptr && p.sfi_j == NO && all_have && all1 && ((reinterpret_cast<uintptr_t>(p.fi) & 15u) == 0);      // (wave-uniform; true only when no lane is outside this branch)
                    if (whole_rows) {
                        double* mine = dead_slot + lane * NO;
                        mine[0] = v0;
#pragma unroll
                        for (int a = 1; a < NO; ++a) mine[a] = r1[a - 1];
                        __syncthreads();
                        const long long jbase = jp - lane;                       // case of lane 0: the 64 cases are jbase + lane
                        rd2_* out = reinterpret_cast<rd2_*>(p.fi + jbase * NO);
                        const rd2_* src = reinterpret_cast<const rd2_*>(dead_slot);
#pragma unroll
                        for (int q = lane; q < 64 * NO / 2; q += 64) ring_store(&out[q], src[q]);
                    } else {
#pragma unroll
                        for (int a = 1; a < NO; ++a) ring_store(&fio[a], r1[a - 1]);
                    }
                }
            }
        }
        if (havep && !mine1 && knownp != FULL) {
            double* fio = p.fi + jp * p.sfi_j;
            double M[NE], rhs[NO];
            // (nu in graded order is the right-hand-side moment of DOF order for every (dimension, order) here: mom_index == DOF index
            // is not assumed — expand_moments_from asks for nu by moment index)
            expand_moments_from<DIM, ORDER>([&](int i) { return entry(i); }, [&](int i) { return entry(NM + i); }, M, rhs);
            if (knownp) {
                double val[NO];
#pragma unroll
                for (int a = 0; a < NO; ++a) val[a] = (((knownp & ~droppedp) >> a) & 1ull) ? fio[a] : 0.0;
                eliminate_knowns<NO>(M, rhs, knownp, val);
            }
            ldlt_factor<NO>(M);
            ldlt_solve<NO>(M, rhs);
#pragma unroll
            for (int a = 0; a < NO; ++a)
                if (!((knownp >> a) & 1ull)) ring_store(&fio[a], rhs[a]);
        }
        havep = false;
    };

    // The same solve without a single branch, for a wave whose 64 parked cases are all valid and have no known DOF (round 3:
    // C5 +1.8 %).  (Emitting it INSIDE the straight-line moment pass of the next tile, in the hope that the scheduler would fill the
    // stalls of its dependent chains with the independent moment FMAs, changed nothing — the ISA had the solve first and the moments
    // after it — and the second copy of the moment pass rounded differently from the first: a case's bits then depended on its
    // position in the launch.  One copy of every arithmetic path.)
    auto solve_simple = [&]() {
#pragma unroll
        for (int e = 0; e < NQ; ++e) { swap16(PQ[0][e], PQ[1][e]); swap16(PQ[2][e], PQ[3][e]); }
#pragma unroll
        for (int e = 0; e < NQ; ++e) { swap32(PQ[0][e], PQ[2][e]); swap32(PQ[1][e], PQ[3][e]); }
        auto entry = [&](int i) -> double {
            const int qtr = i / NQ, e = i - qtr * NQ;
            const int r = (qtr == 0) ? 0 : (qtr == 1) ? 2 : (qtr == 2) ? 1 : 3;
            return PQ[r][e];
        };
        double* fio = p.fi + jp * p.sfi_j;
        double M[NE], rhs[NO];
        expand_moments_from<DIM, ORDER>([&](int i) { return entry(i); }, [&](int i) { return entry(NM + i); }, M, rhs);
        ldlt_factor<NO>(M);
        ldlt_solve<NO>(M, rhs);
#if WLSQM_RING_FI_RUN
        if constexpr (G::FI_STAGE) {
            static_assert(!G::FI_STAGE || NO <= G::FI_STAGE_NO, "staging area behind the ring");
            // the 64 cases are consecutive (four consecutive tiles, all valid): their rows are ONE run of 64 * NO doubles
            if (p.sfi_j == NO && ((reinterpret_cast<uintptr_t>(p.fi) & 15u) == 0)) {      // wave-uniform
                double* stg = lds + 2 * G::SLOT;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
                for (int a = 0; a < NO; ++a) stg[lane * NO + a] = rhs[a];
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                rd2_* out = reinterpret_cast<rd2_*>(p.fi + (jp - lane) * NO);
                const rd2_* src = reinterpret_cast<const rd2_*>(stg);
#pragma unroll
                for (int q = lane; q < 64 * NO / 2; q += 64) __builtin_nontemporal_store(src[q], &out[q]);
                havep = false;
                return;
            }
        }
#endif
#pragma unroll
        for (int a = 0; a < NO; ++a) ring_store(&fio[a], rhs[a]);
        havep = false;
    };
#ifndef WLSQM_RING_FUSE_SOLVE
#define WLSQM_RING_FUSE_SOLVE 1
#endif
    constexpr bool FUSE = (WLSQM_RING_FUSE_SOLVE != 0) && DIM == 3;

    constexpr bool DELAY = (DIM == 3);
    bool pending = false;
    const long long tile0 = (long long)blockIdx.x * tiles_per_wg;
    long long tend = tile0 + tiles_per_wg;
    if (tend > ntiles) tend = ntiles;
    if constexpr (GATHER) {
        if (tile0 < tend) { prefetch_indices(tile0); __syncthreads(); }
    }
    if (tile0 < tend) prefetch(tile0, 0, tile0 + 1 < tend);
    int it = 0;
    for (long long tile = tile0; tile < tend; ++tile, ++it) {
        __syncthreads();                               // with a DMA in flight: s_waitcnt vmcnt(0) + s_barrier (one wave)
        const long long j = tile * TC + c;
        const bool valid = j < p.ncases;
        const int nkc = min(nxt.nk, K);
        const bool uniform = (nxt.wm == WLSQM_WEIGHT_UNIFORM);
        unsigned long long known, dropped;
        effective_mask<NO>(nxt.kn, known, dropped);
        double xi[DIM];
#pragma unroll
        for (int m = 0; m < DIM; ++m) xi[m] = nxt.xi[m];
        double f[KPL];
#pragma unroll
        for (int i = 0; i < KPL; ++i) f[i] = fnext[i];
        const double* row = lds + (it & 1) * G::SLOT + c * RS + (ILV ? h : k0) * DIM;       // this lane's share of its case's row
        if (tile + 1 < tend) prefetch(tile + 1, (it + 1) & 1, tile + 2 < tend);
        // DELAY: the solve of the previous four tiles runs HERE, behind the prefetch, so that its fi stores are acknowledged
        // while this tile accumulates instead of at the next barrier (vmcnt counts the stores too); no ring slot is dead at
        // this point, so the rows are stored directly
        const bool full = (G::KC == K) && __all(nkc >= K);     // wave-uniform: no ragged case in this tile
        if constexpr (DELAY) {
            if (pending) {
                // all 64 parked cases valid and without knowns: the branch-free copy (same operations on the same numbers)
                if (FUSE && __all(havep && knownp == 0ull && droppedp == 0ull)) solve_simple();
                else solve_parked(nullptr);
                pending = false;
            }
        }

        // (squared distances are written as explicit fma(dy, dy, dx * dx) everywhere: the ragged and the full-tile code paths
        // must round identically, or a case's result would depend on which other cases share its tile)
        auto offset = [&](int kk, double (&d)[DIM]) {  // neighbour k0 + kk of this lane's case (immediate offsets)
            if constexpr (DIM == 2) {
                const rd2_ xy = *reinterpret_cast<const rd2_*>(row + kk * DIM);      // ds_read_b128
                d[0] = xy.x - xi[0]; d[1] = xy.y - xi[1];
            } else {
#pragma unroll
                for (int m = 0; m < DIM; ++m) d[m] = row[(ILV ? G::LPC * kk : kk) * DIM + m] - xi[m];
            }
        };
        auto sqdist = [&](const double (&d)[DIM]) {    // one rounding sequence for every code path (see above)
            double d2 = d[0] * d[0];
#pragma unroll
            for (int m = 1; m < DIM; ++m) d2 = fma(d[m], d[m], d2);
            return d2;
        };
        // (keeping the offsets and squared distances of this pass in registers for the moment pass — 40 doubles for the 3D order-2
        // share — measured 1.6 % slower than reading and subtracting again: C5 ring / tile ratio 0.931 against 0.915)
        double max_d2 = 0.0;
        if (full) {
#pragma unroll
            for (int kk = 0; kk < KPL; ++kk) {
                double d[DIM];
                offset(kk, d);
                const double d2 = sqdist(d);
                max_d2 = d2 > max_d2 ? d2 : max_d2;
            }
        } else {
#pragma unroll
            for (int kk = 0; kk < KPL; ++kk) {
                double d[DIM];
                offset(kk, d);
                double d2 = sqdist(d);
                d2 = (slot_of(kk) < nkc) ? d2 : 0.0;
                max_d2 = d2 > max_d2 ? d2 : max_d2;
            }
        }
#pragma unroll
        for (int off = TC; off < WV; off <<= 1) { const double o = __shfl_xor(max_d2, off, WV); max_d2 = o > max_d2 ? o : max_d2; }
        const double inv_max = inverse_max(max_d2);

        double mu[NM], nu[NO];
#pragma unroll
        for (int e = 0; e < NM; ++e) mu[e] = 0.0;
#pragma unroll
        for (int a = 0; a < NO; ++a) nu[a] = 0.0;
        auto neighbour = [&](int kk, bool live) {
            double d[DIM];
            offset(kk, d);
#pragma unroll
            for (int m = 0; m < DIM; ++m) d[m] = live ? d[m] : 0.0;
            const double d2 = sqdist(d);
            const double w = live ? weight(d2, inv_max, uniform) : 0.0;
            accumulate_moments_best<DIM, ORDER>(mu, nu, d, w, live ? f[kk] : 0.0);
        };
        if (full) {
            // (requesting the coordinates of the next four neighbours from LDS before working on the current one — the compiler
            // issues the ds_reads in pairs and waits right behind the second — measured the same: 0.461 against 0.457 ms on C3)
#pragma unroll UNR
            for (int kk = 0; kk < KPL; ++kk) neighbour(kk, true);
        } else {
#pragma unroll
            for (int kk = 0; kk < KPL; ++kk) neighbour(kk, slot_of(kk) < nkc);
        }
        // reduce-scatter over the four lanes of a case.  V = (mu, nu), 60 entries.  Step 1 (rows of 16 lanes: h <-> h ^ 1) on the
        // pairs (e, e + 30): even rows keep the sums of the first half, odd rows of the second; step 2 (halves of the wave:
        // h <-> h ^ 2) on the pairs (e, e + 15) of what a lane kept.  Same association as the xor butterfly it replaces
        // ((h, h^1) first), so the sums are bit-identical to it; 135 instead of 360 instructions and nothing through LDS.
        {
            double Rr[NV / 2], Qq[NQ];
            double pad[NV - NM - NO + 1];
#pragma unroll
            for (int i = 0; i < NV - NM - NO + 1; ++i) pad[i] = 0.0;
            auto V = [&](int i) -> double& { return i < NM ? mu[i] : (i < NM + NO ? nu[i - NM] : pad[i - NM - NO]); };
#pragma unroll
            for (int e = 0; e < NV / 2; ++e) { swap16(V(e), V(e + NV / 2)); Rr[e] = V(e) + V(e + NV / 2); }
#pragma unroll
            for (int e = 0; e < NQ; ++e) { swap32(Rr[e], Rr[e + NQ]); Qq[e] = Rr[e] + Rr[e + NQ]; }
            // lane h now holds quarter {0, 2, 1, 3}[h] of the list (2D order 4: h = 0: entries 0..14, 1: 30..44, 2: 15..29, 3: 45..59)
            const int slot = it & 3;                   // wave-uniform
            if (slot == 0) {
#pragma unroll
                for (int e = 0; e < NQ; ++e) PQ[0][e] = Qq[e];
            } else if (slot == 1) {
#pragma unroll
                for (int e = 0; e < NQ; ++e) PQ[1][e] = Qq[e];
            } else if (slot == 2) {
#pragma unroll
                for (int e = 0; e < NQ; ++e) PQ[2][e] = Qq[e];
            } else {
#pragma unroll
                for (int e = 0; e < NQ; ++e) PQ[3][e] = Qq[e];
            }
        }
        if (h == (it & 3)) { jp = j; knownp = known; droppedp = dropped; havep = valid; }
        if ((it & 3) == 3) {                                            // the 64 lanes hold 64 different cases (this tile's slot is dead)
            if (DELAY && tile + 1 < tend) pending = true;
            else solve_parked(lds + (it & 1) * G::SLOT);
        }
    }
    if (it & 3) solve_parked(lds + ((it - 1) & 1) * G::SLOT);   // leftovers of a run that is not a multiple of 4 tiles
}

// (Two tiles ahead — the DMAs of tile t + 2 issued as soon as tile t has read its slot for the last time, the wait before a tile
// written as `s_waitcnt vmcnt(<DMAs of one prefetch>)` so that the younger prefetch stays in flight — does not survive the
// compiler: it puts its own `s_waitcnt vmcnt(0)` behind the hand-written one — for the prefetched scalar / fk REGISTER loads, whose
// pending count it no longer knows at the loop header (conditional or unconditional prefetch, with or without s_barrier) — and the
// prefetch in the middle of the tile spills 416-496 B per lane around the live moments.)
// (A persistent launch — one workgroup per resident slot, groups of four tiles drawn from a global counter one group ahead, the
// next group's first tile prefetched under the current group's last — measured SLOWER than one workgroup per group: C5 0.355
// against 0.328 ms; the dispatcher's balancing of many short workgroups is worth more than the saved first-tile latency.)
// tiles per workgroup: a multiple of 4 (one solve per 4 tiles); WLSQM_HIP_RING_TILES overrides (A/B: 1M C3 cases at
// 4 / 8 / 16 / 64 tiles per workgroup 0.532 / 0.539 / 0.536 / 0.539 ms — the dispatcher balances short workgroups best)
static int ring_tiles_per_wg() {
    const char* e = getenv("WLSQM_HIP_RING_TILES");
    const int v = e ? atoi(e) : 4;
    return v >= 1 ? v : 4;
}

template <int DIM, int K, bool GATHER> struct RingLaunchGeom {        // ring + (index-based) the index buffer of the tile ahead
    static constexpr int TC = RingGeom<DIM, K>::TC;
    static constexpr size_t LDS_BYTES = RingGeom<DIM, K>::LDS_BYTES + (GATHER ? (size_t)(TC / 2) * (2 * K + WLSQM_RING_IBUF_PAD) * 4 : 0) + (RingGeom<DIM, K>::FI_STAGE ? (size_t)64 * RingGeom<DIM, K>::FI_STAGE_NO * 8 : 0);
};

template <int DIM, int ORDER, int K, int UNR, int MINW, bool GATHER = false>
static int launch_ring_impl(const KParams& p, hipStream_t stream) {
    using G = RingLaunchGeom<DIM, K, GATHER>;
    const long long ntiles = (p.ncases + G::TC - 1) / G::TC;
    auto kern = fit_ring_kernel<DIM, ORDER, K, UNR, MINW, GATHER>;
    static std::atomic<bool> optin[16] = {};   // idempotent opt-in: a race only repeats it
    int dev = 0;
    WLSQM_HIP_CHECK(hipGetDevice(&dev));
    if (dev >= 0 && dev < 16 && !optin[dev]) {
        if (G::LDS_BYTES > 64 * 1024)
            WLSQM_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                                (int)G::LDS_BYTES));
        optin[dev] = true;
    }
    int T = ring_tiles_per_wg();
    if (DIM == 3 && !getenv("WLSQM_HIP_RING_TILES")) {
        // The 3D kernel solves a group of four tiles behind the NEXT tile's prefetch (its fi stores are acknowledged under that
        // tile's arithmetic), so long workgroups hide all but their last solve — if the launch still fills its rounds: with W
        // resident waves, ceil(ntiles / T / W) rounds should be nearly full.  1M C5 cases (62 500 tiles, 1 024 waves), T = 4 / 16 / 20 /
        // 24 / 28 / 32 / 48: 0.342 / 0.339 / 0.380 / 0.347 / 0.387 / 0.320 / 0.437 ms; T = 30 / 31 / 32 / 61 / 62: 0.419 / 0.330 / 0.330 / 0.436 / 0.324 on another
        // box.  Small launches keep four tiles per workgroup.
        static KernelSetup setup;
        long long slots = 0;
        int rc = persistent_grid(reinterpret_cast<const void*>(kern), 64, G::LDS_BYTES, 0, true, setup, &slots);
        if (rc != WLSQM_OK) return rc;
        slots = (long long)((double)slots / grid_multiple());
        if (slots >= 1 && ntiles >= 8 * slots) {
            double best = 0.0;
            for (int t = 16; t <= 64; ++t) {            // (a run that is not a multiple of 4 tiles ends with a partial solve group)
                const double rounds = (double)ntiles / t / (double)slots;
                const double eff = rounds / (double)(long long)(rounds + 0.999999);
                if (eff >= best - 0.01) { best = eff > best ? eff : best; T = t; }      // ties: the longer workgroup
            }
        }
    }
    const long long grid = (ntiles + T - 1) / T;
    if (grid > 0x7fffffffll) { set_error("fit_ring: batch too large for one launch"); return WLSQM_EVALUE; }
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(64), G::LDS_BYTES, stream, p, ntiles, T);
    WLSQM_HIP_CHECK(hipGetLastError());
    note_kernel(GATHER ? "tile-solve-gather" : "tile-solve");
    return WLSQM_OK;
}

}  // namespace wlsqm
