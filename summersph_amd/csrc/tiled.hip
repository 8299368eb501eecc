// tiled.hip -- LDS-staged pair kernels for the fixed-h path (default; pairs.hip keeps the direct-gather
// versions for A/B runs, SPH_FLAG_NO_LDS_TILES).
//
// Why.  With direct gathers every pair visit fetches its neighbour's record through the texture
// addresser (TA): 2 (density) or 6 (forces) scattered 16-B loads per lane and visit.  rocprofv3 shows
// the TA ~76 % busy and the VALU ~40 % (profiles/r01_v2_rocprof_summary.txt): the gather path, not
// arithmetic, bounds the kernels.  But the neighbours of 256 consecutive cell-sorted particles are
// not scattered: they lie in THREE contiguous intervals of the sorted order, one per offset along the
// slowest grid axis (cells of neighbouring columns are adjacent in memory, the fastest axis is the
// short one).  So a workgroup stages those intervals chunk by chunk into LDS with fully coalesced
// loads -- each record is fetched once per workgroup instead of once per pair -- and the pair loop
// reads LDS.
//
// Neighbour list ("ELL, wave-strided, 4-packed"): entry k of particle i = (wave w, lane l) is component
// k%4 of the int4 at nlist4[(w*cap4 + k/4)*64 + l].  Entries are appended while the intervals are
// scanned in order, so each lane's list is ascending within an interval; the evaluation kernels walk
// the same intervals and consume, per staged chunk, the entries that fall into it.  (Any chunk size
// works: entries of one interval pass are ascending, and a lane that runs ahead only consumes entries
// whose records are in the staged chunk anyway.)
//
// Replaces (citations: /root/reference/SUMMER_SPH.f90, "[F]"): the same reference code as pairs.hip --
// density_tree_search/get_density [F]:398-457, get_pressure_and_sound_speed [F]:459-468,
// SPH_tree_search/get_SPH [F]:295-395, zero_rates + gas side of sink_gravforces [F]:779-793,559-576.
#include <cmath>
#include <cstdlib>

#include "pair_common.hpp"

namespace sph {

namespace {

constexpr int TB = 256;              // threads per workgroup = targets per workgroup
constexpr int WT_BS = 1024;         // whole-tile kernels: threads (= targets) per workgroup, one workgroup per CU
constexpr int WT_CAP = WT_TILE_RECORDS;   // ... and records per tile (116 KB of {x,y,z,m}; + 40 KB kernel table)
constexpr int T_NL = 512;            // staged records per chunk: neighbour-list build (32 B each)
// density / forces: chunk size and where the kernel table lives are template parameters, chosen by
// measurement (launch_*_tiled): TABLDS = table staged in LDS (40 KB, limits workgroups per CU),
// else read through L1 from a pair-packed copy {t[k], t[k+1]} (one 16-B load per visit).

// lerp from the pair-packed global table: tp[k] = {t[k], t[k+1]}
__device__ __forceinline__ double pair_lerp(const double2 *__restrict__ tp, double qi, double inv_dq, int nq) {
    const double t = qi * inv_dq;
    const int k = min((int)t, nq - 1);
    const double a = t - (double)k;
    const double2 v = tp[k];
    return (1.0 - a) * v.x + a * v.y;
}

__device__ __forceinline__ int sel4(const int4 &v, int k) {
    const int a = (k & 1) ? v.y : v.x, b = (k & 1) ? v.w : v.z;
    return (k & 2) ? b : a;
}

// The three candidate intervals [lo, hi) of a workgroup (one per offset o2 = -1, 0, +1 along the slowest
// axis), from the per-target row ranges.  s_lo/s_hi: LDS scratch of 3 ints each.
struct Rows {
    int jb[3], je[3];     // this target's three cell rows (o1 = -1, 0, +1) of the current o2
};

__device__ __forceinline__ void target_rows(const GridDesc &g, const int32_t *__restrict__ cell_start, const int cc[3], bool live,
                                            int o2, Rows &r) {
    const int d0 = g.dim[g.s[0]], d1 = g.dim[g.s[1]], d2 = g.dim[g.s[2]];
    const int c2 = cc[2] + o2;
    const int lo0 = max(cc[0] - 1, 0), hi0 = min(cc[0] + 1, d0 - 1);
#pragma unroll
    for (int o1 = -1; o1 <= 1; o1++) {
        const int c1 = cc[1] + o1;
        const bool ok = live && c2 >= 0 && c2 < d2 && c1 >= 0 && c1 < d1;
        if (ok) {
            const int64_t row = ((int64_t)c2 * d1 + c1) * d0;
            r.jb[o1 + 1] = cell_start[row + lo0];
            r.je[o1 + 1] = cell_start[row + hi0 + 1];
        } else {
            r.jb[o1 + 1] = 0; r.je[o1 + 1] = 0;
        }
    }
}

__device__ __forceinline__ void block_interval(const Rows &r, int *s_lo, int *s_hi, int &lo, int &hi) {
    int mn = 0x7fffffff, mx = 0;
#pragma unroll
    for (int k = 0; k < 3; k++)
        if (r.je[k] > r.jb[k]) { mn = min(mn, r.jb[k]); mx = max(mx, r.je[k]); }
    for (int o = 32; o > 0; o >>= 1) { mn = min(mn, __shfl_xor(mn, o, 64)); mx = max(mx, __shfl_xor(mx, o, 64)); }
    __syncthreads();                                   // protects s_lo/s_hi of the previous interval
    if ((threadIdx.x & 63) == 0) { s_lo[threadIdx.x >> 6] = mn; s_hi[threadIdx.x >> 6] = mx; }
    __syncthreads();
    lo = min(min(s_lo[0], s_lo[1]), min(s_lo[2], s_lo[3]));
    hi = max(max(s_hi[0], s_hi[1]), max(s_hi[2], s_hi[3]));
}

// ------------------------------------------------------------------------------------------
// neighbour list build: every j != i with |x_i - x_j|^2 <= rcut2
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(TB) void nlist_tiled(GridDesc g, const double4 *__restrict__ drec,
                                                  const int32_t *__restrict__ cell_start, int64_t n, double rcut2, int32_t cap,
                                                  int4 *__restrict__ nlist4, int32_t *__restrict__ ncount,
                                                  int32_t *__restrict__ wave_max, int32_t *__restrict__ flags,
                                                  const int32_t *__restrict__ orig, int32_t n_owned) {
    __shared__ double4 tile[T_NL];
    __shared__ int s_lo[4], s_hi[4];
    const int64_t i = (int64_t)xcd_chunk(blockIdx.x, gridDim.x) * TB + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const int64_t w = i >> 6;
    const bool live = i < n && orig[i] < n_owned;
    const double4 pi = drec[i < n ? i : n - 1];
    int cc[3];
    cell_coords(g, pi.x, pi.y, pi.z, cc);
    const int cap4 = cap >> 2;
    int4 *mine = nlist4 + ((size_t)w * cap4) * 64 + lane;
    int cnt = 0;
    int4 buf = make_int4(0, 0, 0, 0);

#pragma unroll
    for (int o2 = -1; o2 <= 1; o2++) {
        Rows r;
        target_rows(g, cell_start, cc, live, o2, r);
        int lo, hi;
        block_interval(r, s_lo, s_hi, lo, hi);
        for (int cb = lo; cb < hi; cb += T_NL) {
            const int ce = min(cb + T_NL, hi);
            __syncthreads();
            for (int t = threadIdx.x; t < ce - cb; t += TB) tile[t] = drec[cb + t];
            __syncthreads();
#pragma unroll
            for (int k = 0; k < 3; k++) {
                const int a = max(r.jb[k], cb), b = min(r.je[k], ce);
                for (int j = a; j < b; j++) {
                    const double4 pj = tile[j - cb];
                    const double dx = pi.x - pj.x, dy = pi.y - pj.y, dz = pi.z - pj.z;
                    const double r2 = dx * dx + dy * dy + dz * dz;
                    if (r2 <= rcut2 && j != (int)i) {
                        const int q = cnt & 3;              // selects, not branches: the accept path runs for every third candidate
                        buf.x = q == 0 ? j : buf.x; buf.y = q == 1 ? j : buf.y; buf.z = q == 2 ? j : buf.z; buf.w = q == 3 ? j : buf.w;
                        if (q == 3 && cnt < cap) mine[(size_t)(cnt >> 2) * 64] = buf;
                        cnt++;
                    }
                }
            }
        }
    }
    if ((cnt & 3) != 0 && cnt < cap) mine[(size_t)(cnt >> 2) * 64] = buf;     // last, partly filled quad
    if (i < n) ncount[i] = live ? cnt : 0;
    const int wm = wave_max_i32(live ? cnt : 0);
    if (lane == 0 && (w << 6) < n) {
        wave_max[w] = min(wm, cap);
        if (wm > 0) atomicMax(&flags[1], wm);
    }
}

// ---- the walk over a lane's packed list -----------------------------------------------------------
struct ListCursor {
    const int4 *mine;
    int4 buf;
    int k, cnt;
    __device__ __forceinline__ void init(const int4 *base, int count) {
        mine = base; k = 0; cnt = count;
        buf = count > 0 ? base[0] : make_int4(0, 0, 0, 0);
    }
    __device__ __forceinline__ int cur() const { return sel4(buf, k); }
    __device__ __forceinline__ void advance() {
        k++;
        if ((k & 3) == 0 && k < cnt) buf = mine[(size_t)(k >> 2) * 64];
    }
};

// force gather record (FREC doubles): x y z m | vx vy vz rho/2 | P/rho^2  c/2  alpha/2  0
__device__ __forceinline__ void write_frec_t(double *__restrict__ frec, int64_t i, const double4 &pi, double vx, double vy,
                                             double vz, double rho, double P, double c, double alpha) {
    double4 *fr = reinterpret_cast<double4 *>(frec + (size_t)i * FREC);
    fr[0] = pi;
    fr[1] = make_double4(vx, vy, vz, 0.5 * rho);
    fr[2] = make_double4(P / (rho * rho), 0.5 * c, 0.5 * alpha, 0.0);       // [F]:381: P/(rho*rho)
}

// ------------------------------------------------------------------------------------------
// density + EOS
// ------------------------------------------------------------------------------------------
template <int T_DE, bool TABLDS>
__global__ __launch_bounds__(TB) void density_tiled(GridDesc g, PairConst pc, const double4 *__restrict__ drec,
                                                    const int32_t *__restrict__ cell_start, const int4 *__restrict__ nlist4,
                                                    int32_t cap, const int32_t *__restrict__ ncount,
                                                    const double *__restrict__ w_tab, const double2 *__restrict__ w_pair,
                                                    int64_t n, const double *__restrict__ u,
                                                    const double *__restrict__ alpha, const double *__restrict__ vx,
                                                    const double *__restrict__ vy, const double *__restrict__ vz,
                                                    double *__restrict__ rho, double *__restrict__ P, double *__restrict__ cs,
                                                    double *__restrict__ frec, const int32_t *__restrict__ orig, int32_t n_owned) {
    extern __shared__ double lds_dyn[];
    double *lds_w = lds_dyn;                                               // nq+1 doubles (padded to even) if TABLDS
    double4 *tile = reinterpret_cast<double4 *>(lds_dyn + (TABLDS ? ((pc.nq + 2) & ~1) : 0));
    __shared__ int s_lo[4], s_hi[4];
    if (TABLDS)
        for (int k = threadIdx.x; k <= pc.nq; k += TB) lds_w[k] = w_tab[k];

    const int64_t i = (int64_t)xcd_chunk(blockIdx.x, gridDim.x) * TB + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const int64_t w = i >> 6;
    const bool live = i < n && orig[i] < n_owned;
    const double4 pi = drec[i < n ? i : n - 1];
    int cc[3];
    cell_coords(g, pi.x, pi.y, pi.z, cc);
    ListCursor lc;
    lc.init(nlist4 + ((size_t)w * (cap >> 2)) * 64 + lane, live ? min(ncount[i], cap) : 0);
    const double inv_h = 1.0 / pc.h, inv_dq = 1.0 / pc.dq;
    double acc = 0.0;

#pragma unroll
    for (int o2 = -1; o2 <= 1; o2++) {
        Rows r;
        target_rows(g, cell_start, cc, live, o2, r);
        int lo, hi;
        block_interval(r, s_lo, s_hi, lo, hi);
        for (int cb = lo; cb < hi; cb += T_DE) {
            const int ce = min(cb + T_DE, hi);
            __syncthreads();
            for (int t = threadIdx.x; t < ce - cb; t += TB) tile[t] = drec[cb + t];
            __syncthreads();
            while (true) {
                const int j = lc.cur();
                const bool has = lc.k < lc.cnt && j >= cb && j < ce;
                if (!__any(has)) break;
                if (has) {
                    const double4 pj = tile[j - cb];
                    const double n0 = pi.x - pj.x, n1 = pi.y - pj.y, n2 = pi.z - pj.z;     // [F]:445
                    double dr, rs;
                    fast_sqrt_rsqrt(n0 * n0 + n1 * n1 + n2 * n2, dr, rs);                   // [F]:446
                    const double qi = dr * inv_h;                                          // [F]:111
                    if (qi <= 2.0)                                                         // [F]:113-118,454
                        acc = fma(pj.w, TABLDS ? table_lerp(lds_w, qi, inv_dq, pc.nq) : pair_lerp(w_pair, qi, inv_dq, pc.nq), acc);
                    lc.advance();
                }
            }
        }
    }
    if (!live) return;
    acc = fma(pi.w, w_tab[0], acc);            // self term, r = 0 ([F]:443-455 visits the particle's own leaf)
    const double rhoi = acc / pc.wnorm;                                                    // [F]:125
    const double ui = u[i];
    const double Pi = pc.gamma_m1 * ui * rhoi;                                             // [F]:465
    const double ci = sqrt(pc.gamma * Pi / rhoi);                                          // [F]:466
    rho[i] = rhoi; P[i] = Pi; cs[i] = ci;
    write_frec_t(frec, i, pi, vx[i], vy[i], vz[i], rhoi, Pi, ci, alpha[i]);
}

// ------------------------------------------------------------------------------------------
// forces
// ------------------------------------------------------------------------------------------
template <int T_FO, bool TABLDS>
__global__ __launch_bounds__(TB) void forces_tiled(GridDesc g, PairConst pc, const double *__restrict__ frec,
                                                   const int32_t *__restrict__ cell_start, const int4 *__restrict__ nlist4,
                                                   int32_t cap, const int32_t *__restrict__ ncount,
                                                   const double *__restrict__ dw_tab, const double2 *__restrict__ dw_pair,
                                                   const double *__restrict__ sink, int64_t n,
                                                   double *__restrict__ ax, double *__restrict__ ay, double *__restrict__ az,
                                                   double *__restrict__ du, double *__restrict__ dalpha,
                                                   const int32_t *__restrict__ orig, int32_t n_owned) {
    extern __shared__ double lds_dyn[];
    double *lds_dw = lds_dyn;
    double4 *tile = reinterpret_cast<double4 *>(lds_dyn + (TABLDS ? ((pc.nq + 2) & ~1) : 0));   // T_FO * 3 double4
    __shared__ int s_lo[4], s_hi[4];
    if (TABLDS)
        for (int k = threadIdx.x; k <= pc.nq; k += TB) lds_dw[k] = dw_tab[k];

    const int64_t i = (int64_t)xcd_chunk(blockIdx.x, gridDim.x) * TB + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const int64_t w = i >> 6;
    const bool live = i < n && orig[i] < n_owned;
    const int64_t self = i < n ? i : n - 1;
    const double4 *fi = reinterpret_cast<const double4 *>(frec + (size_t)self * FREC);
    const double4 A = fi[0], B = fi[1], Cc = fi[2];   // x y z m | vx vy vz rho/2 | P/rho^2 c/2 alpha/2 -
    int cc[3];
    cell_coords(g, A.x, A.y, A.z, cc);
    ListCursor lc;
    lc.init(nlist4 + ((size_t)w * (cap >> 2)) * 64 + lane, live ? min(ncount[i], cap) : 0);
    const double inv_h = 1.0 / pc.h, inv_dq = 1.0 / pc.dq;
    const double4 *fg = reinterpret_cast<const double4 *>(frec);
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, sdu = 0.0, sdal = 0.0;

#pragma unroll
    for (int o2 = -1; o2 <= 1; o2++) {
        Rows r;
        target_rows(g, cell_start, cc, live, o2, r);
        int lo, hi;
        block_interval(r, s_lo, s_hi, lo, hi);
        for (int cb = lo; cb < hi; cb += T_FO) {
            const int ce = min(cb + T_FO, hi);
            __syncthreads();
            for (int t = threadIdx.x; t < (ce - cb) * 3; t += TB) tile[t] = fg[(size_t)cb * 3 + t];
            __syncthreads();
            while (true) {
                const int j = lc.cur();
                const bool has = lc.k < lc.cnt && j >= cb && j < ce;
                if (!__any(has)) break;
                if (has) {
                    const double4 *tj = tile + (j - cb) * 3;
                    const double4 Aj = tj[0], Bj = tj[1], Cj = tj[2];
                    const double n0 = A.x - Aj.x, n1 = A.y - Aj.y, n2 = A.z - Aj.z;           // [F]:356
                    const double r2 = n0 * n0 + n1 * n1 + n2 * n2;
                    double dr, rs;
                    fast_sqrt_rsqrt(r2, dr, rs);                                              // [F]:357
                    const double qi = dr * inv_h;
                    if (qi <= 2.0 && r2 > 0.0) {       // beyond 2h all terms are 0; r == 0: coincident points, DESIGN.md
                        const double v0 = B.x - Bj.x, v1 = B.y - Bj.y, v2 = B.z - Bj.z;       // [F]:358
                        const double vdotr = fmin(v0 * n0 + v1 * n1 + v2 * n2, 0.0);          // [F]:359-361
                        const double dWm = (TABLDS ? table_lerp(lds_dw, qi, inv_dq, pc.nq)
                                                   : pair_lerp(dw_pair, qi, inv_dq, pc.nq)) * rs;  // [F]:366; rs = 1/dr of [F]:363
                        const double g0 = n0 * dWm, g1 = n1 * dWm, g2 = n2 * dWm;             // [F]:363,368
                        const double vdotgradW = g0 * v0 + g1 * v1 + g2 * v2;                 // [F]:370
                        const double vis_nu = (pc.h * vdotr) * fast_rcp(r2 + pc.visc_eps_h2); // [F]:373
                        const double cbar = Cc.y + Cj.y, abar = Cc.z + Cj.z;                  // [F]:374,376 (halves stored)
                        const double visc = (abar * vis_nu) * (2.0 * vis_nu - cbar) * fast_rcp(B.w + Bj.w);   // [F]:378
                        const double Cf = Cc.x + Cj.x + visc;                                 // [F]:381-382
                        const double mC = Aj.w * Cf;
                        s0 = fma(mC, g0, s0); s1 = fma(mC, g1, s1); s2 = fma(mC, g2, s2);     // [F]:383
                        const double mv = Aj.w * vdotgradW;
                        sdu = fma(mv, Cc.x + 0.5 * visc, sdu);                                // [F]:387
                        sdal += mv;                                                           // [F]:390
                    }
                    lc.advance();
                }
            }
        }
    }
    if (!live) return;
    // zero_rates, then the gas side of sink_gravforces, [F]:567-576
    double a0 = pc.grav ? ax[i] : 0.0, a1 = pc.grav ? ay[i] : 0.0, a2 = pc.grav ? az[i] : 0.0;   // [F]:824-825
    for (int s = 0; s < pc.ns; s++) {
        const double v0 = A.x - sink[0 * MAX_SINKS + s], v1 = A.y - sink[1 * MAX_SINKS + s], v2 = A.z - sink[2 * MAX_SINKS + s];
        const double dr = sqrt(v0 * v0 + v1 * v1 + v2 * v2);
        const double d3 = dr * dr * dr;
        const double ms = sink[6 * MAX_SINKS + s];
        a0 = a0 - (ms * (pc.G * v0 / d3)); a1 = a1 - (ms * (pc.G * v1 / d3)); a2 = a2 - (ms * (pc.G * v2 / d3));
    }
    const double inv_dwn = 1.0 / pc.dwnorm;                                                   // [F]:126, applied once
    ax[i] = a0 - s0 * inv_dwn; ay[i] = a1 - s1 * inv_dwn; az[i] = a2 - s2 * inv_dwn;
    du[i] = sdu * inv_dwn;
    // [F]:317; rho_i = 2 B.w, c_i = 2 Cc.y, alpha_i = 2 Cc.z (exact)
    dalpha[i] = fmax((sdal * inv_dwn) / (2.0 * B.w), 0.0) + pc.alpha_decay * ((pc.alpha_floor - 2.0 * Cc.z) * (2.0 * Cc.y) / pc.h);
}

// ------------------------------------------------------------------------------------------
// "Whole tile" evaluation: the three candidate intervals of the workgroup are staged ONCE, side by side, and the
// pair loop is the lockstep list walk of pairs.hip with the record fetched from LDS instead of through the texture
// addresser.  A divergent 16-byte gather costs the vector-memory path about one lane per clock and CU
// (tests/tools/micro/gather_lds.hip: 2.8-3.5x slower than the same gather out of LDS, staging included), and that
// rate, not cache misses, bounds density_kernel / forces_kernel.  A list entry j becomes a tile slot by comparing it
// with the three interval starts.  A workgroup whose intervals do not fit the tile (dense regions, thick domains,
// workgroups that straddle two slices) walks its lists with the memory gathers of pairs.hip instead -- decided per
// workgroup, so the LDS loop contains no vector-memory gather at all: its only vector loads are the list rows, read
// as int4 (four entries) two rows ahead.  Same operations in the same order as pairs.hip: bitwise the same results.
// ------------------------------------------------------------------------------------------
struct TileMap {
    int lo[3], len[3], base[3];
    int need;             // records of the three intervals together
    // branch-free: the third interval is the default (every list entry of a workgroup whose tile fits lies in one of the three)
    __device__ __forceinline__ int slot(int j) const {
        const unsigned u0 = (unsigned)(j - lo[0]), u1 = (unsigned)(j - lo[1]);
        int s = base[2] + (j - lo[2]);
        s = u1 < (unsigned)len[1] ? base[1] + (int)u1 : s;
        s = u0 < (unsigned)len[0] ? (int)u0 : s;
        return s;
    }
};

// s_lo / s_hi: LDS scratch of 3 * NW ints each (NW = waves per workgroup); one barrier pair for the three intervals
template <int NW>
__device__ __forceinline__ void tile_map(const GridDesc &g, const int32_t *__restrict__ cell_start, const int cc[3], bool live,
                                         int *s_lo, int *s_hi, TileMap &m) {
    int mn[3], mx[3];
#pragma unroll
    for (int q = 0; q < 3; q++) {
        Rows r;
        target_rows(g, cell_start, cc, live, q - 1, r);
        mn[q] = 0x7fffffff; mx[q] = 0;
#pragma unroll
        for (int k = 0; k < 3; k++)
            if (r.je[k] > r.jb[k]) { mn[q] = min(mn[q], r.jb[k]); mx[q] = max(mx[q], r.je[k]); }
    }
#pragma unroll
    for (int q = 0; q < 3; q++)
        for (int o = 32; o > 0; o >>= 1) { mn[q] = min(mn[q], __shfl_xor(mn[q], o, 64)); mx[q] = max(mx[q], __shfl_xor(mx[q], o, 64)); }
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int q = 0; q < 3; q++) { s_lo[q * NW + (threadIdx.x >> 6)] = mn[q]; s_hi[q * NW + (threadIdx.x >> 6)] = mx[q]; }
    }
    __syncthreads();
    int total = 0;
#pragma unroll
    for (int q = 0; q < 3; q++) {
        int lo = 0x7fffffff, hi = 0;
        for (int k = 0; k < NW; k++) { lo = min(lo, s_lo[q * NW + k]); hi = max(hi, s_hi[q * NW + k]); }
        const int len = hi > lo ? hi - lo : 0;
        m.lo[q] = lo; m.len[q] = len; m.base[q] = total;
        total += len;
    }
    m.need = total;
}

__device__ __forceinline__ size_t poff(int k) { return (size_t)(k >> 2) * 256 + (k & 3); }

template <int TCAP, int BS>
__global__ __launch_bounds__(BS) void density_wt(GridDesc g, PairConst pc, const double4 *__restrict__ drec,
                                                 const int32_t *__restrict__ cell_start, const int32_t *__restrict__ nlist,
                                                 int32_t cap, const int32_t *__restrict__ ncount, const int32_t *__restrict__ wave_max,
                                                 const double *__restrict__ w_tab,
                                                 int64_t n, const double *__restrict__ u,
                                                 const double *__restrict__ alpha, const double *__restrict__ vx,
                                                 const double *__restrict__ vy, const double *__restrict__ vz,
                                                 double *__restrict__ rho, double *__restrict__ P, double *__restrict__ cs,
                                                 double *__restrict__ frec, const int32_t *__restrict__ orig, int32_t n_owned) {
    extern __shared__ double lds_dyn[];
    double *lds_w = lds_dyn;                                               // nq+1 doubles (padded to even)
    double4 *tile = reinterpret_cast<double4 *>(lds_dyn + ((pc.nq + 2) & ~1));
    __shared__ int s_lo[3 * (BS / 64)], s_hi[3 * (BS / 64)];
    for (int k = threadIdx.x; k <= pc.nq; k += BS) lds_w[k] = w_tab[k];

    const int64_t i = (int64_t)xcd_chunk(blockIdx.x, gridDim.x) * BS + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const int64_t w = i >> 6;
    const bool live = i < n && orig[i] < n_owned;
    const int self = i < n ? (int)i : (int)(n - 1);
    const double4 pi = drec[self];
    int cc[3];
    cell_coords(g, pi.x, pi.y, pi.z, cc);
    TileMap tm;
    tile_map<BS / 64>(g, cell_start, cc, live, s_lo, s_hi, tm);
    const bool fits = tm.need <= TCAP;                 // workgroup-uniform
    if (fits) {
#pragma unroll
        for (int q = 0; q < 3; q++)
            for (int t = threadIdx.x; t < tm.len[q]; t += BS) tile[tm.base[q] + t] = drec[tm.lo[q] + t];
    }
    __syncthreads();
    if ((i & ~(int64_t)63) >= n) return;

    const int cnt = live ? min(ncount[i], cap) : 0;
    const int kmax = __builtin_amdgcn_readfirstlane(wave_max[w]);
    const double inv_h = 1.0 / pc.h, inv_dq = 1.0 / pc.dq;
    double acc = 0.0;
    auto visit = [&](const double4 &pj, bool act) {
        const double n0 = pi.x - pj.x, n1 = pi.y - pj.y, n2 = pi.z - pj.z;     // [F]:445
        double dr, rs;
        fast_sqrt_rsqrt(n0 * n0 + n1 * n1 + n2 * n2, dr, rs);                   // [F]:446
        const double qi = dr * inv_h;                                          // [F]:111
        if (act && qi <= 2.0)                                                  // [F]:113-118,454
            acc = fma(pj.w, table_lerp(lds_w, qi, inv_dq, pc.nq), acc);
    };
    if (fits && kmax > 0) {
        // list rows as int4 (four entries), fetched two rows ahead with wave-uniform, unconditional loads
        const int4 *mine4 = reinterpret_cast<const int4 *>(nlist) + ((size_t)w * (cap >> 2)) * 64 + lane;
        const int nrow = (kmax + 3) >> 2;
        int4 qa = load_row(mine4);
        int4 qb = load_row(mine4 + (size_t)min(1, nrow - 1) * 64);
        double4 p1 = tile[0 < cnt ? tm.slot(qa.x) : 0];
        for (int r = 0; r < nrow; r++) {
            const int4 qc = load_row(mine4 + (size_t)min(r + 2, nrow - 1) * 64);
#pragma unroll
            for (int v = 0; v < 4; v++) {
                const int k = 4 * r + v;
                if (k < kmax) {                                     // wave-uniform
                    const double4 pj = p1;
                    p1 = tile[k + 1 < cnt ? tm.slot(v < 3 ? comp4(qa, v + 1) : qb.x) : 0];
                    visit(pj, k < cnt);
                }
            }
            qa = qb; qb = qc;
        }
    } else if (!fits) {
        const int32_t *mine = nlist + (((size_t)w * (cap >> 2)) * 64 + lane) * 4;
        int j1 = 0 < cnt ? load_entry(mine + poff(0)) : self;
        int j2 = 1 < cnt ? load_entry(mine + poff(1)) : self;
        double4 p1 = drec[j1];
        for (int k = 0; k < kmax; k++) {
            const double4 pj = p1;
            j1 = j2;
            if (k + 2 < cnt) j2 = load_entry(mine + poff(k + 2));
            if (k + 1 < cnt) p1 = drec[j1];
            visit(pj, k < cnt);
        }
    }
    if (!live) return;
    acc = fma(pi.w, lds_w[0], acc);            // self term, r = 0 ([F]:443-455 visits the particle's own leaf)
    const double rhoi = acc / pc.wnorm;                                                    // [F]:125
    const double ui = u[i];
    const double Pi = pc.gamma_m1 * ui * rhoi;                                             // [F]:465
    const double ci = sqrt(pc.gamma * Pi / rhoi);                                          // [F]:466
    rho[i] = rhoi; P[i] = Pi; cs[i] = ci;
    write_frec_t(frec, i, pi, vx[i], vy[i], vz[i], rhoi, Pi, ci, alpha[i]);
}

// forces: {x, y, z, m} of the neighbour from the LDS tile (the records density_wt stages), the other 64 bytes of its
// record (v, rho/2 | P/rho^2, c/2, alpha/2) through the vector-memory path: 4 instead of 6 divergent 16-byte loads per
// visit.  (The whole 96-byte record in LDS leaves room for one 256-thread workgroup per CU -- one wave per SIMD -- and
// was measured slower than pairs.hip: 0.65-0.75 vs 0.54 ms.)
template <int TCAP, int BS>
__global__ __launch_bounds__(BS) void forces_wt(GridDesc g, PairConst pc, const double4 *__restrict__ drec, const double *__restrict__ frec,
                                                const int32_t *__restrict__ cell_start, const int32_t *__restrict__ nlist,
                                                int32_t cap, const int32_t *__restrict__ ncount, const int32_t *__restrict__ wave_max,
                                                const double *__restrict__ dw_tab,
                                                const double *__restrict__ sink, int64_t n,
                                                double *__restrict__ ax, double *__restrict__ ay, double *__restrict__ az,
                                                double *__restrict__ du, double *__restrict__ dalpha,
                                                const int32_t *__restrict__ orig, int32_t n_owned,
                                                const int32_t *__restrict__ wave_class, int32_t want) {
    extern __shared__ double lds_dyn[];
    double *lds_dw = lds_dyn;
    double4 *tile = reinterpret_cast<double4 *>(lds_dyn + ((pc.nq + 2) & ~1));
    __shared__ int s_lo[3 * (BS / 64)], s_hi[3 * (BS / 64)];
    const int64_t i = (int64_t)xcd_chunk(blockIdx.x, gridDim.x) * BS + threadIdx.x;
    if (wave_class) {       // split evaluation (multi-GPU overlap): a block with no wave of class `want` leaves at once
        bool any = false;
        const int64_t w0 = (i - threadIdx.x) >> 6;
        for (int k = 0; k < BS / 64; k++)
            any |= ((w0 + k) << 6) < n && wave_class[w0 + k] == want;
        if (!any) return;
    }
    for (int k = threadIdx.x; k <= pc.nq; k += BS) lds_dw[k] = dw_tab[k];
    const int lane = threadIdx.x & 63;
    const int64_t w = i >> 6;
    const bool live = i < n && orig[i] < n_owned;
    const int self = i < n ? (int)i : (int)(n - 1);
    const double4 *fg = reinterpret_cast<const double4 *>(frec);
    const double4 A = fg[(size_t)self * 3], B = fg[(size_t)self * 3 + 1], Cc = fg[(size_t)self * 3 + 2];
    int cc[3];
    cell_coords(g, A.x, A.y, A.z, cc);
    TileMap tm;
    tile_map<BS / 64>(g, cell_start, cc, live, s_lo, s_hi, tm);
    const bool fits = tm.need <= TCAP;
    if (fits) {
#pragma unroll
        for (int q = 0; q < 3; q++)
            for (int t = threadIdx.x; t < tm.len[q]; t += BS) tile[tm.base[q] + t] = drec[tm.lo[q] + t];
    }
    __syncthreads();
    if ((i & ~(int64_t)63) >= n) return;
    if (wave_class && wave_class[w] != want) return;

    const int cnt = live ? min(ncount[i], cap) : 0;
    const int kmax = __builtin_amdgcn_readfirstlane(wave_max[w]);
    const double inv_h = 1.0 / pc.h, inv_dq = 1.0 / pc.dq;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, sdu = 0.0, sdal = 0.0;
    auto visit = [&](const double4 &Aj, const double4 &Bj, const double4 &Cj, bool act) {
        const double n0 = A.x - Aj.x, n1 = A.y - Aj.y, n2 = A.z - Aj.z;               // [F]:356
        const double r2 = n0 * n0 + n1 * n1 + n2 * n2;
        double dr, rs;
        fast_sqrt_rsqrt(r2, dr, rs);                                                  // [F]:357
        const double qi = dr * inv_h;
        if (act && qi <= 2.0 && r2 > 0.0) {
            const double v0 = B.x - Bj.x, v1 = B.y - Bj.y, v2 = B.z - Bj.z;           // [F]:358
            const double vdotr = fmin(v0 * n0 + v1 * n1 + v2 * n2, 0.0);              // [F]:359-361
            const double dWm = table_lerp(lds_dw, qi, inv_dq, pc.nq) * rs;            // [F]:366; rs = 1/dr of [F]:363
            const double g0 = n0 * dWm, g1 = n1 * dWm, g2 = n2 * dWm;                 // [F]:363,368
            const double vdotgradW = g0 * v0 + g1 * v1 + g2 * v2;                     // [F]:370
            const double vis_nu = (pc.h * vdotr) * fast_rcp(r2 + pc.visc_eps_h2);     // [F]:373
            const double cbar = Cc.y + Cj.y, abar = Cc.z + Cj.z;                      // [F]:374,376 (halves stored)
            const double visc = (abar * vis_nu) * (2.0 * vis_nu - cbar) * fast_rcp(B.w + Bj.w);   // [F]:378
            const double Cf = Cc.x + Cj.x + visc;                                     // [F]:381-382
            const double mC = Aj.w * Cf;
            s0 = fma(mC, g0, s0); s1 = fma(mC, g1, s1); s2 = fma(mC, g2, s2);         // [F]:383
            const double mv = Aj.w * vdotgradW;
            sdu = fma(mv, Cc.x + 0.5 * visc, sdu);                                    // [F]:387
            sdal += mv;                                                               // [F]:390
        }
    };
    if (fits && kmax > 0) {
        const int4 *mine4 = reinterpret_cast<const int4 *>(nlist) + ((size_t)w * (cap >> 2)) * 64 + lane;
        const int nrow = (kmax + 3) >> 2;
        int4 qa = load_row(mine4);
        int4 qb = load_row(mine4 + (size_t)min(1, nrow - 1) * 64);
        int jn = 0 < cnt ? qa.x : self;
        double4 A1 = tile[0 < cnt ? tm.slot(jn) : 0];
        double4 B1 = fg[(size_t)jn * 3 + 1], C1 = fg[(size_t)jn * 3 + 2];
        for (int r = 0; r < nrow; r++) {
            const int4 qc = load_row(mine4 + (size_t)min(r + 2, nrow - 1) * 64);
#pragma unroll
            for (int v = 0; v < 4; v++) {
                const int k = 4 * r + v;
                if (k < kmax) {                                     // wave-uniform
                    const double4 Aj = A1, Bj = B1, Cj = C1;
                    jn = v < 3 ? comp4(qa, v + 1) : qb.x;
                    A1 = tile[k + 1 < cnt ? tm.slot(jn) : 0];
                    if (k + 1 < cnt) { B1 = fg[(size_t)jn * 3 + 1]; C1 = fg[(size_t)jn * 3 + 2]; }     // idle lanes issue no gather
                    visit(Aj, Bj, Cj, k < cnt);
                }
            }
            qa = qb; qb = qc;
        }
    } else if (!fits) {
        const int32_t *mine = nlist + (((size_t)w * (cap >> 2)) * 64 + lane) * 4;
        int j1 = 0 < cnt ? load_entry(mine + poff(0)) : self;
        int j2 = 1 < cnt ? load_entry(mine + poff(1)) : self;
        const double4 *fj = fg + (size_t)j1 * 3;
        double4 A1 = fj[0], B1 = fj[1], C1 = fj[2];
        for (int k = 0; k < kmax; k++) {
            const double4 Aj = A1, Bj = B1, Cj = C1;
            j1 = j2;
            if (k + 2 < cnt) j2 = load_entry(mine + poff(k + 2));
            if (k + 1 < cnt) { fj = fg + (size_t)j1 * 3; A1 = fj[0]; B1 = fj[1]; C1 = fj[2]; }
            visit(Aj, Bj, Cj, k < cnt);
        }
    }
    if (!live) return;
    // zero_rates, then the gas side of sink_gravforces, [F]:567-576
    double a0 = pc.grav ? ax[i] : 0.0, a1 = pc.grav ? ay[i] : 0.0, a2 = pc.grav ? az[i] : 0.0;   // [F]:824-825
    for (int s = 0; s < pc.ns; s++) {
        const double v0 = A.x - sink[0 * MAX_SINKS + s], v1 = A.y - sink[1 * MAX_SINKS + s], v2 = A.z - sink[2 * MAX_SINKS + s];
        const double dr = sqrt(v0 * v0 + v1 * v1 + v2 * v2);
        const double d3 = dr * dr * dr;
        const double ms = sink[6 * MAX_SINKS + s];
        a0 = a0 - (ms * (pc.G * v0 / d3)); a1 = a1 - (ms * (pc.G * v1 / d3)); a2 = a2 - (ms * (pc.G * v2 / d3));
    }
    const double inv_dwn = 1.0 / pc.dwnorm;                                                   // [F]:126, applied once
    ax[i] = a0 - s0 * inv_dwn; ay[i] = a1 - s1 * inv_dwn; az[i] = a2 - s2 * inv_dwn;
    du[i] = sdu * inv_dwn;
    dalpha[i] = fmax((sdal * inv_dwn) / (2.0 * B.w), 0.0) + pc.alpha_decay * ((pc.alpha_floor - 2.0 * Cc.z) * (2.0 * Cc.y) / pc.h);
}

// workgroups (of the whole-tile kernels' size) whose three intervals do not fit the tile -> flags[4]
template <int TCAP, int BS>
__global__ __launch_bounds__(BS) void wt_fit_probe(GridDesc g, const double4 *__restrict__ drec, const int32_t *__restrict__ cell_start,
                                                   int64_t n, const int32_t *__restrict__ orig, int32_t n_owned, int32_t *__restrict__ flags) {
    __shared__ int s_lo[3 * (BS / 64)], s_hi[3 * (BS / 64)];
    const int64_t i = (int64_t)xcd_chunk(blockIdx.x, gridDim.x) * BS + threadIdx.x;
    const bool live = i < n && orig[i] < n_owned;
    const double4 pi = drec[i < n ? i : n - 1];
    int cc[3];
    cell_coords(g, pi.x, pi.y, pi.z, cc);
    TileMap tm;
    tile_map<BS / 64>(g, cell_start, cc, live, s_lo, s_hi, tm);
    if (threadIdx.x == 0 && tm.need > TCAP) atomicAdd(&flags[4], 1);
}

inline unsigned tb_blocks(int64_t n) { return (unsigned)((n + TB - 1) / TB); }

}  // namespace

#define TL_CHECK(expr)                                                      \
    do {                                                                    \
        hipError_t _e = (expr);                                             \
        if (_e != hipSuccess) {                                             \
            c->err = std::string(#expr) + ": " + hipGetErrorString(_e);     \
            return SPH_ERR_HIP;                                             \
        }                                                                   \
    } while (0)

int nlist_build_tiled(sph_ctx *c) {
    const int64_t n = c->n;
    if (n == 0) return SPH_OK;
    const PairConst pc = make_pair_const(c);
    const unsigned wt_blocks = (unsigned)((n + WT_BS - 1) / WT_BS);
    if (c->whole_tile) {
        TL_CHECK(hipMemsetAsync(c->d_flags + 4, 0, sizeof(int32_t), c->stream));
        wt_fit_probe<WT_CAP, WT_BS><<<dim3(wt_blocks), dim3(WT_BS), 0, c->stream>>>(
            c->grid, reinterpret_cast<const double4 *>(c->drec), c->cell_start, n, c->orig, (int32_t)c->n_owned, c->d_flags);
        TL_CHECK(hipGetLastError());
        TL_CHECK(hipMemcpyAsync(c->h_pinned + 10, c->d_flags + 4, sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    }
    for (int attempt = 0; attempt < 8; attempt++) {
        TL_CHECK(hipMemsetAsync(c->d_flags + 1, 0, sizeof(int32_t), c->stream));
        nlist_tiled<<<dim3(tb_blocks(n)), dim3(TB), 0, c->stream>>>(c->grid, reinterpret_cast<const double4 *>(c->drec), c->cell_start,
                                                                    n, pc.rcut2, c->nl_cap, reinterpret_cast<int4 *>(c->nlist),
                                                                    c->ncount, c->wave_max, c->d_flags, c->orig, (int32_t)c->n_owned);
        TL_CHECK(hipGetLastError());
        TL_CHECK(hipMemcpyAsync(c->h_pinned + 9, c->d_flags + 1, sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
        TL_CHECK(hipStreamSynchronize(c->stream));
        const int32_t mx = *reinterpret_cast<int32_t *>(c->h_pinned + 9);
        c->nl_max = mx;
        if (c->whole_tile) {
            // the whole-tile kernels pay a prologue and run their fall-back loop at one workgroup per CU: use them when
            // (nearly) every workgroup's intervals fit the tile -- thin discs and sheets; thick domains keep pairs.hip
            const int32_t misfit = *reinterpret_cast<int32_t *>(c->h_pinned + 10);
            c->wt_fit_pct = (int32_t)(100 - (100 * (int64_t)misfit) / std::max<int64_t>(wt_blocks, 1));
            c->wt_ok = (int64_t)misfit * 10 <= (int64_t)wt_blocks;
        }
        if (mx <= c->nl_cap) { c->nlist_builds++; return SPH_OK; }
        ctx_free(c, c->nlist);
        c->nl_cap = ((mx + mx / 8 + 8) + 3) & ~3;
        if (ctx_alloc(c, &c->nlist, (size_t)c->nl_waves_cap * c->nl_cap * 64, "neighbour list") != SPH_OK) { c->nl_cap = 0; return SPH_ERR_NOMEM; }
    }
    c->err = "neighbour list did not converge";
    return SPH_ERR_STATE;
}

static int tile_variant() {
    static int v = -1;
    if (v < 0) { const char *e = getenv("SPH_TILE_VARIANT"); v = e ? atoi(e) : 0; }
    return v;
}

template <int T, bool TABLDS>
static hipError_t density_tiled_launch(sph_ctx *c, const PairConst &pc) {
    const size_t lds = (TABLDS ? (size_t)((pc.nq + 2) & ~1) * sizeof(double) : 0) + (size_t)T * sizeof(double4);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&density_tiled<T, TABLDS>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    density_tiled<T, TABLDS><<<dim3(tb_blocks(c->n)), dim3(TB), lds, c->stream>>>(
        c->grid, pc, reinterpret_cast<const double4 *>(c->drec), c->cell_start, reinterpret_cast<const int4 *>(c->nlist), c->nl_cap,
        c->ncount, c->w_tab, reinterpret_cast<const double2 *>(c->w_pair), c->n, c->f[SPH_F_U], c->f[SPH_F_ALPHA], c->f[SPH_F_VX],
        c->f[SPH_F_VY], c->f[SPH_F_VZ], c->f[SPH_F_RHO], c->f[SPH_F_P], c->f[SPH_F_C], c->frec, c->orig, (int32_t)c->n_owned);
    return hipGetLastError();
}

template <int T, bool TABLDS>
static hipError_t forces_tiled_launch(sph_ctx *c, const PairConst &pc) {
    const size_t lds = (TABLDS ? (size_t)((pc.nq + 2) & ~1) * sizeof(double) : 0) + (size_t)T * 3 * sizeof(double4);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&forces_tiled<T, TABLDS>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    forces_tiled<T, TABLDS><<<dim3(tb_blocks(c->n)), dim3(TB), lds, c->stream>>>(
        c->grid, pc, c->frec, c->cell_start, reinterpret_cast<const int4 *>(c->nlist), c->nl_cap, c->ncount, c->dw_tab,
        reinterpret_cast<const double2 *>(c->dw_pair), c->sink, c->n, c->f[SPH_F_AX], c->f[SPH_F_AY], c->f[SPH_F_AZ], c->f[SPH_F_DU],
        c->f[SPH_F_DALPHA], c->orig, (int32_t)c->n_owned);
    return hipGetLastError();
}

hipError_t launch_density_tiled(sph_ctx *c, const PairConst &pc) {
    if (c->n == 0) return hipSuccess;
    switch (tile_variant()) {
        case 1: return density_tiled_launch<1216, true>(c, pc);
        case 2: return density_tiled_launch<512, false>(c, pc);
        case 3: return density_tiled_launch<512, true>(c, pc);
        default: return density_tiled_launch<1216, false>(c, pc);
    }
}

hipError_t launch_forces_tiled(sph_ctx *c, const PairConst &pc) {
    if (c->n == 0) return hipSuccess;
    switch (tile_variant()) {
        case 1: return forces_tiled_launch<400, true>(c, pc);
        case 2: return forces_tiled_launch<320, false>(c, pc);
        case 3: return forces_tiled_launch<256, true>(c, pc);
        default: return forces_tiled_launch<448, false>(c, pc);
    }
}


// ---- whole-tile kernels -------------------------------------------------------------------------------------
hipError_t launch_density_wt(sph_ctx *c, const PairConst &pc) {
    if (c->n == 0) return hipSuccess;
    const size_t lds = (size_t)((pc.nq + 2) & ~1) * sizeof(double) + (size_t)WT_CAP * sizeof(double4);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&density_wt<WT_CAP, WT_BS>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    density_wt<WT_CAP, WT_BS><<<dim3((unsigned)((c->n + WT_BS - 1) / WT_BS)), dim3(WT_BS), lds, c->stream>>>(
        c->grid, pc, reinterpret_cast<const double4 *>(c->drec), c->cell_start, c->nlist, c->nl_cap, c->ncount, c->wave_max,
        c->w_tab, c->n, c->f[SPH_F_U], c->f[SPH_F_ALPHA], c->f[SPH_F_VX],
        c->f[SPH_F_VY], c->f[SPH_F_VZ], c->f[SPH_F_RHO], c->f[SPH_F_P], c->f[SPH_F_C], c->frec, c->orig, (int32_t)c->n_owned);
    return hipGetLastError();
}

// part 0: every wave.  part 1 / 2: only the waves of class 0 (interior) / class 1, as launch_forces
hipError_t launch_forces_wt(sph_ctx *c, const PairConst &pc, int part) {
    if (c->n == 0) return hipSuccess;
    const size_t lds = (size_t)((pc.nq + 2) & ~1) * sizeof(double) + (size_t)WT_CAP * sizeof(double4);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&forces_wt<WT_CAP, WT_BS>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    forces_wt<WT_CAP, WT_BS><<<dim3((unsigned)((c->n + WT_BS - 1) / WT_BS)), dim3(WT_BS), lds, c->stream>>>(
        c->grid, pc, reinterpret_cast<const double4 *>(c->drec), c->frec, c->cell_start, c->nlist, c->nl_cap, c->ncount, c->wave_max,
        c->dw_tab, c->sink, c->n, c->f[SPH_F_AX], c->f[SPH_F_AY], c->f[SPH_F_AZ], c->f[SPH_F_DU],
        c->f[SPH_F_DALPHA], c->orig, (int32_t)c->n_owned, part ? c->wave_class : nullptr, part == 2 ? 1 : 0);
    return hipGetLastError();
}

}  // namespace sph
