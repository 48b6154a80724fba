// fit_quad.hip — the solve of the 35-unknown systems (3D order 4) with FOUR LANES PER CASE (round 4).
//
// Reference path (file:line in /root/reference): the normal equations of make_A impl.pyx:566-602 and the right-hand side of
// impl.pyx:768-787 arrive as MOMENTS (wlsqm_moments.hpp) from fit_stage_kernel<3,4,PART> (csrc/fit_stage.hip: one lane per case, the
// 165 + 35 sums of a case in two launches of 104 / 96 accumulators each); here: knowns elimination impl.pyx:792-818, dgetrf / dgetrs
// lapackdrivers.pyx:1628-1665 as an unpivoted LDL^T (the fast mode's solve for every shape, wlsqm_kernels.hpp).
//
// Why four lanes.  630 + 35 entries are 1 330 registers: no lane holds a case.  The kernel this replaces gave a case a whole
// wavefront (fit_rows.hip: lane i = row i, pivot rows by v_readlane, 29 of 64 lanes idle): 4 780 vector instructions per case.  A quad
// is the widest group whose lanes exchange registers without LDS or readlanes — the DPP quad_perm modifier broadcasts one lane's
// register to its quad in one move per dword —, and 16 cases per wave means every instruction works for 16 cases:
//   * row i of the matrix belongs to lane i mod 4 of the quad, slot i / 4 of that lane: nine slots per lane (the last one of lane 3
//     is a dummy row), slot s stored from column 4 s on (the aligned start makes the stored shape the same for all four lanes; the
//     few entries left of the diagonal are dead weight), the right-hand side as column 35: 180 doubles per lane, whole groups of four
//     columns everywhere;
//   * elimination step j: the owner's row is broadcast to the quad, every lane updates its own later rows (the multiplier of row i is
//     column i of the pivot row — a select over four registers by the lane's position), the right-hand side rides along as a column
//     (= the forward substitution), the pivot row stays as U and its diagonal entry becomes 1 / d_j;
//   * back substitution by columns: x_m from its owner, broadcast, every lane takes it out of its rows' right-hand sides.
// Same operations for every lane, no divergence, no LDS traffic after the moments have been picked up.
#include <type_traits>

#include "wlsqm_internal.hpp"
#include "wlsqm_kernels.hpp"
#include "wlsqm_moments.hpp"

namespace wlsqm {

namespace quad {
constexpr int NO = 35, NP = 36, NMOM = 165, NW = 200, SLOTS = 9, PITCH = 201;
__host__ __device__ constexpr int seg(int s) { return NP - 4 * s; }                          // columns 4 s .. 34 and, as column 35, the right-hand side
__host__ __device__ constexpr int off(int s) { return NP * s - 2 * s * (s - 1); }            // sum of seg(0 .. s - 1)
constexpr int NR = off(SLOTS);                                                               // 180 doubles per lane

__host__ __device__ constexpr double fact_of(int a) {
    return a < NO ? mom_inv_fact(Mono<3>::P[a]) * mom_inv_fact(Mono<3>::Q[a]) * mom_inv_fact(Mono<3>::R[a]) : 1.0;
}
// where entry (i, m) of the matrix and entry i of the right-hand side sit in a case's block of NW moments (slot NW: a zero, for the
// padding row / column 35), and the factorial scale of row i
struct Tables {
    unsigned char idx[NP * NP];
    unsigned char nuidx[NP];
    double fact[NP];
};
constexpr Tables make_tables() {
    Tables t{};
    for (int i = 0; i < NP; ++i) {
        for (int m = 0; m < NP; ++m)
            t.idx[i * NP + m] = (i < NO && m < NO)
                ? (unsigned char)mom_index<3>(Mono<3>::P[i] + Mono<3>::P[m], Mono<3>::Q[i] + Mono<3>::Q[m], Mono<3>::R[i] + Mono<3>::R[m])
                : (unsigned char)NW;
        t.nuidx[i] = i < NO ? (unsigned char)(NMOM + mom_index<3>(Mono<3>::P[i], Mono<3>::Q[i], Mono<3>::R[i])) : (unsigned char)NW;
        t.fact[i] = fact_of(i);
    }
    return t;
}
__constant__ Tables g_tab = make_tables();

// the register of lane SRC of every quad, in all four lanes
template <int SRC> __device__ __forceinline__ double bcast(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_mov_dpp(lo, SRC * 0x55, 0xf, 0xf, true);
    hi = __builtin_amdgcn_mov_dpp(hi, SRC * 0x55, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ void pin(double& x) { asm volatile("" : "+v"(x)); }      // (see the elimination loop)
template <int V> using ic = std::integral_constant<int, V>;
template <int B, int E, class F> __device__ __forceinline__ void static_for(F&& f) {
    if constexpr (B < E) { f(ic<B>{}); static_for<B + 1, E>(f); }
}
}  // namespace quad

// p.ws: the moments of the launch, written by fit_stage_kernel<3,4,PART>: entry e of case t at ws[((t / 64) NW + e) 64 + t % 64].
__global__ __launch_bounds__(64, 1) void quad_solve_kernel(const KParams p) {
    using namespace quad;
    // the moments of the wave's 16 cases; after the rows have been built from them, the same memory holds slots 0 and 1 of every
    // lane's rows (68 of the 180 doubles: entry e of lane t at lds[64 e + t]) — the rows that are finished first and then only read
    // again by the back substitution.  All 180 in registers were 360 of the lane's 512 and the compiler put 120 doubles of them in
    // scratch (0.83 ms per 200k cases: a lone wave waits out every scratch access).
#ifndef WLSQM_QUAD_LROWS
#define WLSQM_QUAD_LROWS 2
#endif
    constexpr int LROWS = WLSQM_QUAD_LROWS, NL = off(LROWS);
    __shared__ __attribute__((aligned(16))) double mom[(16 * PITCH > 64 * NL) ? 16 * PITCH : 64 * NL];
    __shared__ unsigned int s_idx32[NP * NP / 4];
    const int lane = threadIdx.x, l = lane & 3, c = lane >> 2;
    const bool l_odd = (lane & 1) != 0, l_high = (lane & 2) != 0;
    const long long case0 = (long long)blockIdx.x * 16;
    for (int w = lane; w < NP * NP / 4; w += 64) s_idx32[w] = reinterpret_cast<const unsigned int*>(g_tab.idx)[w];
    {
        // the 16 cases of this wave are 16 consecutive lanes of one 64-case group of the moment kernels: 128 contiguous bytes per entry
        const double* src = p.ws + (case0 >> 6) * (long long)(NW * 64) + (case0 & 63);
        const int cc = lane & 15, e0 = lane >> 4;
#ifndef WLSQM_QUAD_LOAD_UNROLL
#define WLSQM_QUAD_LOAD_UNROLL 50       // all of a lane's 50 moments requested at once (10 at a time: 3Do4@400k 1.14 against 1.09 ms)
#endif
#pragma unroll WLSQM_QUAD_LOAD_UNROLL
        for (int it = 0; it < NW / 4; ++it) mom[cc * PITCH + it * 4 + e0] = src[(it * 4 + e0) * 64 + cc];
        if (lane < 16) mom[lane * PITCH + NW] = 0.0;
    }
    __syncthreads();
    const unsigned char* const s_idx = reinterpret_cast<const unsigned char*>(s_idx32);
    const long long t = case0 + c;
    const bool valid = t < p.ncases;
    const long long j = valid ? t : p.ncases - 1;                      // lanes past the end replay the last case (never stored)
    unsigned long long known, dropped;
    effective_mask<NO>(p.knowns[j * p.sknowns], known, dropped);
    const unsigned long long vals = known & ~dropped;                  // DOFs whose value is in fi
    const bool any_known = __any(known != 0ull);                       // wave-uniform: the common wave has no known DOF at all
    double* const fio = p.fi + j * p.sfi_j;
    const double* const mc = mom + c * PITCH;

    double R[NR];
    // ---- the lane's nine rows from the moments: masked to identity in the known DOFs (eliminate_knowns), the values of the known
    // DOFs taken out of the right-hand side first (with the unmasked entries; an unknown DOF enters as 0.0: fma(-m, 0, g) = g exactly).
    // Row 35 (lane 3, slot 8) does not exist: its table entries point at the zero slot of the moment block.
    double fa[SLOTS];
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) {
        const int i = 4 * s + l;
        fa[s] = g_tab.fact[i];
        R[off(s) + seg(s) - 1] = mc[g_tab.nuidx[i]] * fa[s];
    }
    if (any_known) {
        // (rolled, table look-ups at run time: unrolled over the 35 DOFs this rare path cost the whole kernel 243 spilled registers and
        // 976 B of scratch — without it the kernel needs 442 registers and no scratch; round 5)
#pragma nounroll
        for (int om = 0; om < NO; ++om) {
            const double v = ((vals >> om) & 1ull) ? fio[om] : 0.0;
            const double fom = g_tab.fact[om];
#pragma unroll
            for (int s = 0; s < SLOTS; ++s) {
                const int i = 4 * s + l;
                const double vi = (i == om) ? 0.0 : v;
                R[off(s) + seg(s) - 1] = fma(-(mc[s_idx[i * NP + om]] * (fa[s] * fom)), vi, R[off(s) + seg(s) - 1]);
            }
        }
#pragma unroll
        for (int s = 0; s < SLOTS; ++s) R[off(s) + seg(s) - 1] = ((known >> (4 * s + l)) & 1ull) ? 0.0 : R[off(s) + seg(s) - 1];
    }
    // (the wave-uniform choice hoisted out of the loops: tested per entry it put a branch between every two LDS reads and the 171
    // look-ups of a lane ran one after the other, each waiting out the LDS latency twice)
    auto build = [&](auto masked_tag) __attribute__((always_inline)) {
        constexpr bool MASKED = decltype(masked_tag)::value;
        // Every entry is two dependent LDS reads (table word, then the moment).  Left to itself the compiler — short of registers
        // with the rows filling up — issued them one at a time: 171 x 2 LDS round trips per lane, about 40 % of the kernel.  So in
        // batches, separated by scheduling barriers: the table words of a slot, then eight moments at a time, then their products.
        static_for<0, SLOTS>([&](auto s_) __attribute__((always_inline)) {
            constexpr int s = decltype(s_)::value;
            const int i = 4 * s + l;
            const bool bi = (known >> i) & 1ull;                       // (bit 35 is never set: effective_mask keeps NO bits)
            unsigned int w[SLOTS];
            static_for<s, SLOTS>([&](auto g_) __attribute__((always_inline)) { w[decltype(g_)::value] = s_idx32[i * (NP / 4) + decltype(g_)::value]; });
            __builtin_amdgcn_sched_barrier(0);
            static_for<0, (SLOTS - s + 1) / 2>([&](auto h_) __attribute__((always_inline)) {
                constexpr int g0 = s + 2 * decltype(h_)::value;        // column groups g0 and g0 + 1
                double v[8];
                static_for<0, 8>([&](auto e_) __attribute__((always_inline)) {
                    constexpr int e = decltype(e_)::value, g = g0 + e / 4, k = e % 4;
                    if constexpr (g < SLOTS && 4 * g + k < NO) v[e] = mc[(w[g] >> (8 * k)) & 0xffu];
                });
                __builtin_amdgcn_sched_barrier(0);
                static_for<0, 8>([&](auto e_) __attribute__((always_inline)) {
                    constexpr int e = decltype(e_)::value, g = g0 + e / 4, k = e % 4, m = 4 * g + k;
                    if constexpr (g < SLOTS && m < NO) {               // (column 35 of the stored row is the right-hand side)
                        double x = v[e] * (fa[s] * fact_of(m));
                        if (MASKED) {
                            const bool bm = (known >> m) & 1ull;
                            x = (bi || bm) ? ((i == m) ? 1.0 : 0.0) : x;
                        }
                        R[off(s) + 4 * (g - s) + k] = x;
                    }
                });
                __builtin_amdgcn_sched_barrier(0);
            });
        });
    };
    if (any_known) build(std::true_type{}); else build(std::false_type{});

    __syncthreads();                                                  // every lane is done with the moments
    double* const LR = mom + lane;
#pragma unroll
    for (int e = 0; e < NL; ++e) LR[e * 64] = R[e];
    // (slot and column are compile-time constants everywhere below: static_for instead of unrolled loops, whose indices the
    // optimizer has to discover — with the LDS / register choice inside, some of the 35 steps' inner loops stayed rolled and the
    // rows went to scratch as an indexed array)
    auto ld = [&](auto s_, auto q_) __attribute__((always_inline)) -> double {
        constexpr int s = decltype(s_)::value, q = decltype(q_)::value;
        if constexpr (s < LROWS) return LR[(off(s) + q) * 64]; else return R[off(s) + q];
    };
    auto st = [&](auto s_, auto q_, double v) __attribute__((always_inline)) {
        constexpr int s = decltype(s_)::value, q = decltype(q_)::value;
        if constexpr (s < LROWS) LR[(off(s) + q) * 64] = v; else R[off(s) + q] = v;
    };

    // ---- LDL^T by rows, right-looking; the right-hand side rides along as column 35 (= the forward substitution).  Step jj walks the
    // pivot row by groups of four columns: the four entries are broadcast, they hold the multiplier of the slot that starts at this
    // group (a select by the lane's position), and every slot that stores these columns is updated — nothing of the pivot row stays
    // live beyond its group.
    static_for<0, NO>([&](auto jj_) __attribute__((always_inline)) {
        constexpr int jj = decltype(jj_)::value, sj = jj / 4, lj = jj % 4;
        const double inv = recip(bcast<lj>(ld(ic<sj>{}, ic<lj>{})));
        double tt[SLOTS];
        static_for<sj, SLOTS>([&](auto sg_) __attribute__((always_inline)) {
            constexpr int sg = decltype(sg_)::value;
            double b[4];
            static_for<0, 4>([&](auto k_) __attribute__((always_inline)) {
                constexpr int k = decltype(k_)::value;
                b[k] = bcast<lj>(ld(ic<sj>{}, ic<4 * (sg - sj) + k>{}));
            });
            // (two levels of selects on the bits of the lane's position: an == chain becomes a switch with real branches, 1 500 basic
            // blocks in this kernel, and the register allocator gives up on the matrix)
            const double a01 = l_odd ? b[1] : b[0], a23 = l_odd ? b[3] : b[2];
            const double a = l_high ? a23 : a01;
            tt[sg] = a * inv;
            if (sg == sj) tt[sg] = (l > lj) ? tt[sg] : 0.0;            // rows up to the pivot row are finished
            if (sg == SLOTS - 1) tt[sg] = (l == 3) ? 0.0 : tt[sg];     // the dummy row (its "column" is the right-hand side)
            static_for<sj, sg + 1>([&](auto s_) __attribute__((always_inline)) {
                constexpr int s = decltype(s_)::value;
                static_for<0, 4>([&](auto k_) __attribute__((always_inline)) {
                    constexpr int k = decltype(k_)::value;
                    st(ic<s>{}, ic<4 * (sg - s) + k>{}, fma(-tt[s], b[k], ld(ic<s>{}, ic<4 * (sg - s) + k>{})));
                });
            });
        });
        st(ic<sj>{}, ic<lj>{}, (l == lj) ? inv : ld(ic<sj>{}, ic<lj>{}));      // the owner keeps 1 / d_j where d_j was
        // The step's updates happen IN this step: instruction selection works on the whole basic block and emitted the register
        // rows' updates lazily — all of steps 0 .. 7 right before step 8 read its pivot row —, keeping eight steps' multipliers and
        // pivot rows alive (in scratch) until then.  An empty asm statement per updated entry pins the order; no instruction.
        static_for<(sj > LROWS ? sj : LROWS), SLOTS>([&](auto s_) __attribute__((always_inline)) {
            constexpr int s = decltype(s_)::value;
            static_for<0, seg(s)>([&](auto q_) __attribute__((always_inline)) {
                pin(R[off(s) + decltype(q_)::value]);
            });
        });
        __builtin_amdgcn_sched_barrier(0);
    });

    // ---- back substitution by columns; the unknown DOFs leave from their row's owner
    static_for<0, NO>([&](auto r_) __attribute__((always_inline)) {
        constexpr int m = NO - 1 - decltype(r_)::value, sm = m / 4, lm = m % 4;
        const double x = bcast<lm>(ld(ic<sm>{}, ic<lm>{}) * ld(ic<sm>{}, ic<seg(sm) - 1>{}));
        if (l == lm && valid && !((known >> m) & 1ull)) fio[m] = x;
        static_for<0, sm + 1>([&](auto s_) __attribute__((always_inline)) {
            constexpr int s = decltype(s_)::value;
            double u = ld(ic<s>{}, ic<m - 4 * s>{});
            if (s == sm) u = (l < lm) ? u : 0.0;
            st(ic<s>{}, ic<seg(s) - 1>{}, fma(-u, x, ld(ic<s>{}, ic<seg(s) - 1>{})));
        });
        if (m % 4 == 0) __builtin_amdgcn_sched_barrier(0);
    });
}

// The solve of p.ncases cases whose moments are in p.ws (see the kernel).  (Round 5's sixteen-lanes-per-case form with v_fmac_f64_dpp
// row_newbcast — 1.417 against 1.207 ms per 400k cases: profiles/r05d_row16.txt — is out of the library: tools/experiments/fit_quad_row16.hip.txt.)
int launch_quad_solve(const KParams& p, hipStream_t stream) {
    const long long waves = (p.ncases + 15) / 16;
    if (waves <= 0) return WLSQM_OK;
    if (waves > 0x7fffffffLL) { set_error("fit_quad: batch too large for one launch"); return WLSQM_EVALUE; }
    hipLaunchKernelGGL(quad_solve_kernel, dim3((unsigned)waves), dim3(64), 0, stream, p);
    WLSQM_HIP_CHECK(hipGetLastError());
    return WLSQM_OK;
}

}  // namespace wlsqm
