// wlsqm_tile1.hpp — pieces of the one-wave tile shape shared by fit_tile1_kernel (fit_tilek.hip) and
// solve_many_kernel (solve_many.hip): a 64-lane workgroup owns a tile of 16 cases, 4 lanes per case, lane (c, h)
// takes the neighbours [h*KPL, (h+1)*KPL); xk goes through padded LDS rows, fk straight into registers.
#pragma once
#include "wlsqm_internal.hpp"
#include "wlsqm_kernels.hpp"

namespace wlsqm {

constexpr int K1_WV = 64, K1_TC = 16, K1_LPC = 4, K1_FMAX = 16, K1_ROUND = 8;
typedef double k1d2_ __attribute__((ext_vector_type(2)));

struct Tile1Geom {
    int K, KPL;            // neighbour slots per case; per lane (<= FMAX)
    int RS;                // LDS row stride (doubles)
    int XCH, CPRX;         // 16-byte chunks of the tile's xk block; per row
    float inv_cprx;        // 1 / CPRX: chunk -> row without an integer division (chunk numbers stay below 2^11)
    int fvec;              // this lane's fk values are 16-byte aligned pairs (K and KPL even)
};

// Host side: geometry for K neighbour slots; false when a lane would need more than fmax neighbours.
// (LPC lanes per case: 4 on 16-case tiles by default, 2 on 32-case tiles for the small neighbourhoods)
template <int DIM, int LPC = K1_LPC>
inline bool tile1_geometry(long long K, int fmax, Tile1Geom& G) {
    auto rup = [](int v, int m, int r) { return v + ((r - v % m) % m + m) % m; };
    G.K = (int)K; G.KPL = (int)((K + LPC - 1) / LPC);
    if (G.KPL > fmax) return false;
    G.fvec = (K % 2 == 0 && G.KPL % 2 == 0) ? 1 : 0;
    G.RS = DIM == 2 ? rup((int)K * DIM, 4, 2) : rup((int)K * DIM, 2, 1);   // conflict-free ds_read_b128 / b64
    G.XCH = (K1_WV / LPC) * (int)K * DIM / 2; G.CPRX = (int)K * DIM / 2; G.inv_cprx = 1.0f / (float)G.CPRX;
    return true;
}

// This lane's fk values of row `gr` (K doubles): slots [k0, k0 + KPL), clamped inside the row.
template <int FMAX>
__device__ __forceinline__ void tile1_load_f(double (&f)[FMAX], const double* gr, int k0, const Tile1Geom& G) {
    if (G.fvec) {
        const k1d2_* gv = reinterpret_cast<const k1d2_*>(gr + (k0 < G.K ? k0 : 0));
        const int npair = (min(k0 + G.KPL, G.K) - k0) / 2;      // <= 0 for a lane past the end of the row
#pragma unroll
        for (int i = 0; i < FMAX / 2; ++i)
            if (2 * i < G.KPL) { const k1d2_ v = gv[i < npair ? i : 0]; f[2 * i] = v.x; f[2 * i + 1] = v.y; }
    } else {
#pragma unroll
        for (int kk = 0; kk < FMAX; ++kk)
            if (kk < G.KPL) { const int k = k0 + kk; f[kk] = gr[k < G.K ? k : G.K - 1]; }
    }
}

// The tile's xk block (rows of `nvalid` cases, contiguous from gx0): coalesced 16 B per lane, K1_ROUND loads in flight,
// parked in padded LDS rows.
template <int DIM>
__device__ __forceinline__ void tile1_stage_x(double* sX, const double* gx0, long long nvalid, int lane, const Tile1Geom& G) {
    const k1d2_* gx = reinterpret_cast<const k1d2_*>(gx0);
    const long long xlim = nvalid * G.CPRX;
    for (int q0 = lane; q0 < G.XCH; q0 += K1_WV * K1_ROUND) {
        k1d2_ b[K1_ROUND];
#pragma unroll
        for (int i = 0; i < K1_ROUND; ++i) { const long long q = q0 + i * K1_WV; b[i] = gx[q < xlim ? q : xlim - 1]; }
#pragma unroll
        for (int i = 0; i < K1_ROUND; ++i) {
            const int q = q0 + i * K1_WV;
            if (q < G.XCH) {
                const int r = (int)(((float)q + 0.5f) * G.inv_cprx), c2 = q - r * G.CPRX;
                double* d = sX + r * G.RS + 2 * c2;
                if constexpr (DIM == 2) *reinterpret_cast<k1d2_*>(d) = b[i];      // RS even for DIM == 2
                else { d[0] = b[i].x; d[1] = b[i].y; }
            }
        }
    }
}

// The same for a tile whose 16 cases are picked by an index list (order buckets of a heterogeneous batch): row r of the
// tile is case `mycase` of lane r (every lane passes the case number of ITS row c = lane % 16; rows >= nvalid replay the last
// valid one).  Each row is still one contiguous run of K*DIM doubles, so the loads stay 16 B per lane and line-sized.
template <int DIM>
__device__ __forceinline__ void tile1_stage_x_indexed(double* sX, const double* xk, long long mycase, long long nvalid, int lane,
                                                      const Tile1Geom& G) {
    const long long row_doubles = (long long)G.K * DIM;
    for (int q0 = lane; q0 < G.XCH; q0 += K1_WV * K1_ROUND) {
        k1d2_ b[K1_ROUND];
#pragma unroll
        for (int i = 0; i < K1_ROUND; ++i) {
            const int q = min(q0 + i * K1_WV, G.XCH - 1);
            int r = (int)(((float)q + 0.5f) * G.inv_cprx);
            const int c2 = q - r * G.CPRX;
            r = r < nvalid ? r : (int)nvalid - 1;
            const long long src_case = __shfl(mycase, r, K1_WV);          // lane r holds row r's case number
            b[i] = *reinterpret_cast<const k1d2_*>(xk + src_case * row_doubles + 2 * c2);
        }
#pragma unroll
        for (int i = 0; i < K1_ROUND; ++i) {
            const int q = q0 + i * K1_WV;
            if (q < G.XCH) {
                const int r = (int)(((float)q + 0.5f) * G.inv_cprx), c2 = q - r * G.CPRX;
                double* d = sX + r * G.RS + 2 * c2;
                if constexpr (DIM == 2) *reinterpret_cast<k1d2_*>(d) = b[i];
                else { d[0] = b[i].x; d[1] = b[i].y; }
            }
        }
    }
}

// Largest squared distance of the case (impl.pyx:389-391) over all four lanes; neighbours k >= nkc count as 0.
template <int DIM, int FMAX, int LPC = K1_LPC>
__device__ __forceinline__ double tile1_max_d2(const double* xr, const double (&xi)[DIM], int k0, int nkc, const Tile1Geom& G) {
    double max_d2 = 0.0;
#pragma unroll
    for (int kk = 0; kk < FMAX; ++kk) {
        if (kk < G.KPL) {              // wave-uniform
            const int k = k0 + kk;
            const int kc = k < nkc ? k : 0;
            double d2 = 0.0;
#pragma unroll
            for (int m = 0; m < DIM; ++m) { const double dd = xr[kc * DIM + m] - xi[m]; d2 += dd * dd; }
            d2 = k < nkc ? d2 : 0.0;
            max_d2 = d2 > max_d2 ? d2 : max_d2;
        }
    }
#pragma unroll
    for (int off = K1_WV / LPC; off < K1_WV; off <<= 1) { const double o = __shfl_xor(max_d2, off, K1_WV); max_d2 = o > max_d2 ? o : max_d2; }
    return max_d2;
}

}  // namespace wlsqm
