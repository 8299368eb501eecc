// tiled.hip -- LDS-staged kernels of the fixed-h path: the neighbour-list build and the "whole tile" density / forces
// kernels (pairs.hip keeps the direct-gather versions: thick domains, SPH_FLAG_NO_WHOLE_TILE, A/B runs).
//
// Why.  With direct gathers every pair visit fetches its neighbour's record through the texture addresser (TA): 2
// (density) or 6 (forces) scattered 16-B loads per lane and visit, and a divergent 16-byte gather costs the
// vector-memory path about one lane per clock and CU however many lanes share a line (tests/tools/micro/): the TA, not
// arithmetic or cache misses, bounds density_kernel / forces_kernel.  But the neighbours of a run of consecutive
// cell-sorted particles are not scattered: they lie in THREE contiguous intervals of the sorted order, one per offset
// along the slowest grid axis (cells of neighbouring columns are adjacent in memory, the fastest axis is the short one).
// So a workgroup stages those intervals once, side by side, in LDS with coalesced loads and the pair loop -- the same
// lock-step list walk as pairs.hip -- reads its neighbours from the tile.
//
// Neighbour list ("ELL, wave-strided, 8-packed", tile_common.hpp): 16-bit entries = the neighbour's slot in the tile of the
// target's group of 256; entry k of particle i = (wave w, lane l) is halfword ent_pos(k%8) of the int4 at
// nlist4[(w*cap/8 + k/8)*64 + l].
//
// Replaces (citations: /root/reference/SUMMER_SPH.f90, "[F]"): the same reference code as pairs.hip --
// density_tree_search/get_density [F]:398-457, get_pressure_and_sound_speed [F]:459-468,
// SPH_tree_search/get_SPH [F]:295-395, zero_rates + gas side of sink_gravforces [F]:779-793,559-576.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <functional>

#include "pair_common.hpp"
#include "tile_common.hpp"

namespace sph {

namespace {

constexpr int TB = 256;              // neighbour-list build: threads (= targets) per workgroup
constexpr int T_NL = 512;            // ... and staged records per chunk (32 B each)
constexpr int DEAL_BINS = 1024;      // counting sort of a group's list lengths (longer lists share the last bin)
constexpr int WT_BS = 1024;          // density_wt: threads (= targets) per workgroup, one workgroup per CU
// the arguments of forces_q / density_wt that only their epilogues read (kept in LDS, see there)
struct alignas(16) DensEpiArgs { const double *u, *alpha, *vx, *vy, *vz; double *rho, *P, *cs, *frec; double wnorm, gamma, gamma_m1, h; };
struct alignas(16) EpiArgs { double G, alpha_floor, alpha_decay, inv_dwnorm; double *ax, *ay, *az, *du, *dalpha; int ns, grav; };

// -DSPH_PHASE_CLOCKS (profiles/phase_clocks.sh builds it beside the product library): forces_q / density_wt add up, per phase of
// a group, the ticks of the constant 100-MHz counter -- [0] groups, [1] top of the trip to the tile staged and synchronised,
// [2] the pair loop of wave 0 (the longest lists), [3] the pair loops of all waves, [4] the waves counted in [3], [5] reduction,
// epilogue and stores; +8: the same for density_wt
#ifdef SPH_PHASE_CLOCKS
__device__ unsigned long long g_phase_clocks[64];
#define PHASE_NOW() __builtin_readcyclecounter()
#define PHASE_ADD(slot, v) atomicAdd(&g_phase_clocks[slot], (unsigned long long)(v))
// forces_q, finer: PH_MARK(k) adds the ticks since the previous mark to phase k of wave 0 (an epilogue wave; [32 + k]) and of
// the last wave (a stager; [48 + k]), collected in LDS and added to g_phase_clocks once per workgroup
#define PH_MARK(k) do { const unsigned long long now_ = __builtin_readcyclecounter(); \
        if (ph_sel >= 0 && (threadIdx.x & 63) == 0) atomicAdd(&s_phase[ph_sel][k], now_ - ph_last); ph_last = now_; } while (0)
#else
#define PHASE_NOW() 0ull
#define PHASE_ADD(slot, v) ((void)0)
#define PH_MARK(k) ((void)0)
#endif
constexpr int LDS_BYTES = 160 * 1024;
constexpr int LDS_RESERVE = 1024;    // static LDS of the kernels (interval scratch) + slack

// The three candidate intervals [lo, hi) of a workgroup (one per offset o2 = -1, 0, +1 along the slowest
// axis), from the per-target row ranges.
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
// neighbour list build: every j != i with |x_i - x_j|^2 <= rcut2.  The workgroup stages its candidate intervals
// chunk-wise (coalesced), each lane scans its 9 cell rows out of LDS and appends 16-byte quads to its list column.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(TB) void nlist_tiled(GridDesc g, const double4 *__restrict__ drec,
                                                  const int32_t *__restrict__ cell_start, int64_t n, double rcut2, int32_t cap,
                                                  int4 *__restrict__ nlist4, int32_t *__restrict__ ncount,
                                                  int32_t *__restrict__ wave_max, int32_t *__restrict__ flags,
                                                  const int32_t *__restrict__ orig, int32_t n_owned, int2 *__restrict__ deal,
                                                  int32_t *__restrict__ plan_f, int32_t *__restrict__ plan_h) {
    __shared__ double4 tile[T_NL];
    __shared__ int4 rowbuf[TB];                       // per lane: the row being filled, eight 16-bit entries (packing them in
    __shared__ int s_lo[4], s_hi[4];                  // registers cost the accept path 17 vector instructions instead of 6)
    const int64_t i = (int64_t)xcd_chunk(blockIdx.x, gridDim.x) * TB + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const int64_t w = i >> 6;
    const bool live = i < n && orig[i] < n_owned;
    const double4 pi = drec[i < n ? i : n - 1];
    int cc[3];
    cell_coords(g, pi.x, pi.y, pi.z, cc);
    const int cap8 = cap >> 3;
    int4 *mine = nlist4 + ((size_t)w * cap8) * 64 + lane;
    uint16_t *myrow = reinterpret_cast<uint16_t *>(&rowbuf[threadIdx.x]);
    int cnt = 0;
    int plo[3], plen[3];
    int hlo[2][3], hlen[2][3];                        // thread 0: the intervals of the two half groups (128 targets: waves 0-1, 2-3)
    int base = 0;                                     // tile slot of the current interval's first record

#pragma unroll
    for (int o2 = -1; o2 <= 1; o2++) {
        Rows r;
        target_rows(g, cell_start, cc, live, o2, r);
        int lo, hi;
        block_interval(r, s_lo, s_hi, lo, hi);
        plo[o2 + 1] = lo; plen[o2 + 1] = hi > lo ? hi - lo : 0;
#pragma unroll
        for (int hf = 0; hf < 2; hf++) {              // (s_lo / s_hi: per-wave bounds, valid until the next block_interval)
            const int l = min(s_lo[2 * hf], s_lo[2 * hf + 1]), h = max(s_hi[2 * hf], s_hi[2 * hf + 1]);
            hlo[hf][o2 + 1] = l; hlen[hf][o2 + 1] = h > l ? h - l : 0;
        }
        const int slot0 = base - lo;                  // entry of candidate j: its slot in this group's tile
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
                        myrow[ent_pos(cnt & 7)] = (uint16_t)(j + slot0);
                        if ((cnt & 7) == 7 && cnt < cap) mine[(size_t)(cnt >> 3) * 64] = rowbuf[threadIdx.x];
                        cnt++;
                    }
                }
            }
        }
        base += plen[o2 + 1];
    }
    if ((cnt & 7) != 0 && cnt < cap) mine[(size_t)(cnt >> 3) * 64] = rowbuf[threadIdx.x];     // last, partly filled row (its tail: stale entries, never read)
    if (i < n) ncount[i] = live ? cnt : 0;
    const int wm = wave_max_i32(live ? cnt : 0);
    if (lane == 0 && (w << 6) < n) wave_max[w] = min(wm, cap);
    // the group's longest list and tile need go into its plan record; plan_reduce_kernel forms the maxima the host reads
    // (one atomic per WAVE on one address -- 15 625 of them at 1e6 particles -- is what the memory side serialises)
    __syncthreads();
    if (lane == 0) s_lo[threadIdx.x >> 6] = wm;
    __syncthreads();
    if (threadIdx.x == 0) {
        // the three intervals staged above: what the entries are relative to, and the tile of forces_q for this group
        int32_t *p = plan_f + 8 * (size_t)xcd_chunk(blockIdx.x, gridDim.x);
        const int need = plen[0] + plen[1] + plen[2];
        p[0] = plo[0]; p[1] = plo[1]; p[2] = plo[2]; p[3] = plen[0]; p[4] = plen[1]; p[5] = plen[2]; p[6] = need;
        p[7] = max(max(s_lo[0], s_lo[1]), max(s_lo[2], s_lo[3]));
        if (plan_h) {                                 // forces_q with eight lanes per target: groups of 128
#pragma unroll
            for (int hf = 0; hf < 2; hf++) {
                int32_t *q = plan_h + 8 * (2 * (size_t)xcd_chunk(blockIdx.x, gridDim.x) + hf);
                q[0] = hlo[hf][0]; q[1] = hlo[hf][1]; q[2] = hlo[hf][2]; q[3] = hlen[hf][0]; q[4] = hlen[hf][1]; q[5] = hlen[hf][2];
                q[6] = hlen[hf][0] + hlen[hf][1] + hlen[hf][2]; q[7] = 0;
            }
        }
    }
    if (deal) {
        // forces_q deals the 256 targets of this workgroup (= one of its groups) to its lanes in order of list length,
        // longest first: deal[base + rank] = {index within the group, list length}, non-targets (-1) last.  A counting sort
        // over the lengths in the tile's memory; among equal lengths the order is whatever the LDS atomics give, which no
        // result depends on (a target's sums do not depend on the lanes that form them).
        static_assert(TB == 256 && T_NL * sizeof(double4) >= 2 * DEAL_BINS * sizeof(int), "one group per workgroup; bins fit the tile");
        int *hist = reinterpret_cast<int *>(tile), *start = hist + DEAL_BINS;
        const int len = live ? min(cnt, cap) : -1;
        const int bin = len < 0 ? 0 : min(len, DEAL_BINS - 2) + 1;
        __syncthreads();                                  // the last chunk is no longer read
        for (int k = threadIdx.x; k < DEAL_BINS; k += TB) hist[k] = 0;
        __syncthreads();
        const int pos = atomicAdd(&hist[bin], 1);
        __syncthreads();
        // start[b] = number of targets in bins above b; thread t owns bins 4 t .. 4 t + 3
        constexpr int PER = DEAL_BINS / TB;
        int loc[PER], sum = 0;
#pragma unroll
        for (int k = 0; k < PER; k++) { loc[k] = hist[PER * threadIdx.x + k]; sum += loc[k]; }
        int above = sum;                                  // inclusive suffix sum over the lanes of the wave (lane 63 highest bins)
        for (int o = 1; o < 64; o <<= 1) { const int v = __shfl_down(above, o, 64); if (lane + o < 64) above += v; }
        if (lane == 0) s_lo[threadIdx.x >> 6] = above;    // total of this wave
        __syncthreads();
        int higher = 0;
        for (int q = (threadIdx.x >> 6) + 1; q < TB / 64; q++) higher += s_lo[q];
        int run = above - sum + higher;                   // targets in bins above this thread's highest bin
#pragma unroll
        for (int k = PER - 1; k >= 0; k--) { start[PER * threadIdx.x + k] = run; run += loc[k]; }
        __syncthreads();
        deal[(int64_t)xcd_chunk(blockIdx.x, gridDim.x) * TB + start[bin] + pos] = make_int2(threadIdx.x, len);
    }
}

// ------------------------------------------------------------------------------------------
// "Whole tile" evaluation: the three candidate intervals of the workgroup are staged ONCE, side by side.  A list entry j
// becomes a tile slot by comparing it with the three interval starts.  A workgroup whose intervals do not fit the tile
// (dense regions, thick domains, workgroups that straddle two slices) walks its lists with the memory gathers of
// pairs.hip instead -- decided per workgroup, so the LDS loop contains no fall-back gather: its vector loads are the list
// rows, read as int4 (four entries) two rows ahead, and (forces) the part of the neighbour's record the tile does not hold.
// Three conditions each cost a factor when violated (measured): no rarely-taken vector-memory branch inside
// the LDS loop (the compiler then waits for vmcnt(0) before every use); list rows fetched with wave-uniform,
// unconditional loads and a scalar trip count; enough waves per SIMD.
// ------------------------------------------------------------------------------------------
// s_lo / s_hi: LDS scratch of 3 * NW ints each (NW = waves per workgroup); one barrier for the three intervals
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

// TAB: the W table sits in LDS beside the tile (the default); !TAB: its knots are recomputed (w_knot, bitwise the table's values)
// and the tile gets the table's 40 KB -- for neighbourhoods so dense that the three intervals of a group do not fit otherwise
template <int BS, bool TAB>
__global__ __launch_bounds__(BS) void density_wt(PairConst pc, int32_t tcap, int32_t ngroups, const int32_t *__restrict__ plan,
                                                 const int32_t *__restrict__ plan_f,
                                                 const double4 *__restrict__ drec, const int32_t *__restrict__ nlist,
                                                 int32_t cap, const int32_t *__restrict__ ncount, const int32_t *__restrict__ wave_max,
                                                 const double *__restrict__ w_tab,
                                                 int64_t n, const double *__restrict__ u,
                                                 const double *__restrict__ alpha, const double *__restrict__ vx,
                                                 const double *__restrict__ vy, const double *__restrict__ vz,
                                                 double *__restrict__ rho, double *__restrict__ P, double *__restrict__ cs,
                                                 double *__restrict__ frec, const int32_t *__restrict__ orig, int32_t n_owned) {
    extern __shared__ __align__(16) double lds_dyn[];      // (16: the tile is read 16 bytes at a time; static LDS in front of it must not shift it by 8)
    double *lds_w = lds_dyn;                                               // TAB_LEN(nq) doubles (padded to even)
    double4 *tile = reinterpret_cast<double4 *>(lds_dyn + (TAB ? (TAB_LDS(pc.nq)) : 0));
    // persistent, as forces_q: one workgroup per CU walks over groups of BS targets; the table once, the plan one group ahead;
    // XCD x works on one contiguous eighth of the groups (its L2 then holds what neighbouring groups stage twice)
    if (TAB) for (int k = threadIdx.x; k < TAB_LEN(pc.nq); k += BS) lds_w[k] = w_tab[k];
    const int nx = min(8, (int)gridDim.x), xcd = blockIdx.x % nx, per = ((int)gridDim.x - xcd + nx - 1) / nx;
    const int64_t g_hi = (int64_t)ngroups * (xcd + 1) / nx;
    const int lane = threadIdx.x & 63;
    const double inv_h = pc.inv_h, inv_dq = pc.inv_dq;
    auto w_of = [&](double q) {
        const double t = knot_coord(q, inv_dq);
        return TAB ? table_knots_at(lds_w, t) : knot_knots_at([&](int k) { return w_knot(k, pc.dq, pc.nq); }, t);
    };
    // what only the epilogue needs of the arguments waits in LDS (26 scalar registers less across the pair loop; see forces_q)
    __shared__ DensEpiArgs s_epi;
    if (threadIdx.x == 0) s_epi = DensEpiArgs{u, alpha, vx, vy, vz, rho, P, cs, frec, pc.wnorm, pc.gamma, pc.gamma_m1, pc.h};
    // slot tcap of the tile: the sentinel, far away and massless -- what an idle lane visits instead of being masked
    if (threadIdx.x == 0) tile[tcap] = make_double4(SENTINEL_POS, SENTINEL_POS, SENTINEL_POS, 0.0);
    int64_t group = (int64_t)ngroups * xcd / nx + blockIdx.x / nx;
    TileMap tm_next;
    if (group < g_hi) load_plan(plan, group, tm_next);
#ifdef SPH_PHASE_CLOCKS
    unsigned long long pa_n = 0, pa_stage = 0, pa_pairs = 0, pa_epi = 0;
#endif
    for (; group < g_hi; group += per) {
        const TileMap tm = tm_next;
        if (group + per < g_hi) load_plan(plan, group + per, tm_next);
        const int64_t i = group * BS + threadIdx.x;
        const bool fits = tm.need <= tcap;                 // workgroup-uniform
        [[maybe_unused]] const unsigned long long ph0 = PHASE_NOW();
        __syncthreads();                                   // the previous group's tile is no longer read (first trip: the table is written)
        if (fits) stage_tile<BS, 8, 2, false>(reinterpret_cast<const double2 *>(drec), reinterpret_cast<double2 *>(tile), tm);
        const int64_t w = i >> 6;
        const bool live = i < n && orig[i] < n_owned;
        const int self = i < n ? (int)i : (int)(n - 1);
        const double4 pi = drec[self];
        const int cnt = live ? min(ncount[i], cap) : 0;
        const int kmax = (i & ~(int64_t)63) < n ? __builtin_amdgcn_readfirstlane(wave_max[w]) : 0;
        // the list entries are slots of the tile of the wave's own group of 256 (plan_f): into THIS tile (the union of four
        // such groups) or into the sorted order (direct gathers) with one constant per interval
        EntryMap em = entry_to_index(plan_f, __builtin_amdgcn_readfirstlane(self >> 8));      // wave-uniform: scalar loads
        if (fits) {
            em.a0 = (int)((unsigned)em.a0 + (unsigned)tm.base[0] - (unsigned)tm.lo[0]);
            em.a1 = (int)((unsigned)em.a1 + (unsigned)tm.base[1] - (unsigned)tm.lo[1]);
            em.a2 = (int)((unsigned)em.a2 + (unsigned)tm.base[2] - (unsigned)tm.lo[2]);
        }
        __syncthreads();
        [[maybe_unused]] const unsigned long long ph1 = PHASE_NOW();
        double acc = 0.0;
        const int4 *mine4 = reinterpret_cast<const int4 *>(nlist) + ((size_t)w * (cap >> 3)) * 64 + lane;
        const int nrow = (kmax + 7) >> 3;
        // list rows as int4 (eight 16-bit entries), fetched two rows ahead with wave-uniform, unconditional loads.  A row is
        // walked as two halves of four entries -- entries 4 hh .. 4 hh + 3 are the hh-th halfwords of its four words -- so that
        // the loop body stays four visits long (eight unrolled visits cost 35 more registers and spilled)
        auto half_entry = [](int wd, int sh) { return (int)(((unsigned)wd >> sh) & 0xffffu); };
        if (fits && kmax > 0) {
            int4 qa = load_row(mine4);
            int4 qb = load_row(mine4 + (size_t)min(1, nrow - 1) * 64);
            double4 p1 = tile[0 < cnt ? em(half_entry(qa.x, 0)) : tcap];
            for (int r = 0; r < nrow; r++) {
                const int4 qc = load_row(mine4 + (size_t)min(r + 2, nrow - 1) * 64);
#pragma unroll 1
                for (int hh = 0; hh < 2; hh++) {
                    const int sh = hh << 4;
                    const int nxt = hh == 0 ? half_entry(qa.x, 16) : half_entry(qb.x, 0);      // the entry after this half's last
#pragma unroll
                    for (int v = 0; v < 4; v++) {             // whole halves, no trip-count test: the trips past the wave's
                        const int k = 8 * r + 4 * hh + v;     // longest list are masked like any idle lane
                        const double4 pj = p1;
                        const int s1 = k + 1 < cnt ? em(v < 3 ? half_entry(comp4(qa, v + 1), sh) : nxt) : tcap;
                        density_visit_fn<false>(pi, pj, true, w_of, inv_h, acc, [&]() { p1 = tile[s1]; });     // the next record: behind the knots
                    }
                }
                qa = qb; qb = qc;
            }
        } else if (!fits && kmax > 0) {
            int4 qa = load_row(mine4);
            int4 qb = load_row(mine4 + (size_t)min(1, nrow - 1) * 64);
            double4 p1 = drec[0 < cnt ? em(half_entry(qa.x, 0)) : self];
            for (int r = 0; r < nrow; r++) {
                const int4 qc = load_row(mine4 + (size_t)min(r + 2, nrow - 1) * 64);
#pragma unroll 1
                for (int hh = 0; hh < 2; hh++) {
                    const int sh = hh << 4;
                    const int nxt = hh == 0 ? half_entry(qa.x, 16) : half_entry(qb.x, 0);
#pragma unroll
                    for (int v = 0; v < 4; v++) {
                        const int k = 8 * r + 4 * hh + v;
                        const double4 pj = p1;
                        if (k + 1 < cnt) p1 = drec[em(v < 3 ? half_entry(comp4(qa, v + 1), sh) : nxt)];
                        density_visit_fn(pi, pj, k < cnt, w_of, inv_h, acc);
                    }
                }
                qa = qb; qb = qc;
            }
        }
        [[maybe_unused]] const unsigned long long ph2 = PHASE_NOW();
        if (live) {
            const DensEpiArgs ea = s_epi;
            PairConst pe;
            pe.wnorm = ea.wnorm; pe.gamma = ea.gamma; pe.gamma_m1 = ea.gamma_m1; pe.h = ea.h;
            density_epilogue(pe, i, pi, acc, TAB ? lds_w[0] : w_knot(0, pc.dq, pc.nq), ea.u, ea.alpha, ea.vx, ea.vy, ea.vz, ea.rho, ea.P, ea.cs, ea.frec);
        }
#ifdef SPH_PHASE_CLOCKS
        pa_n++; pa_stage += ph1 - ph0; pa_pairs += ph2 - ph1; pa_epi += PHASE_NOW() - ph2;
#endif
    }
#ifdef SPH_PHASE_CLOCKS
    if ((threadIdx.x & 63) == 0) { PHASE_ADD(8 + 3, pa_pairs); PHASE_ADD(8 + 4, pa_n); }
    if (threadIdx.x == 0) { PHASE_ADD(8 + 0, pa_n); PHASE_ADD(8 + 1, pa_stage); PHASE_ADD(8 + 2, pa_pairs); PHASE_ADD(8 + 5, pa_epi); }
#endif
}

// forces, LPT lanes per target.  A workgroup of BS threads owns BS / LPT consecutive targets; the LPT lanes of a target take
// the LPT entries of a list row between them (lane s: component s of every row), so a trip is a row.  Why: (1) the tile of
// BS / LPT targets is small enough to hold the neighbours' WHOLE 96-byte records beside the dw table -- no vector-memory
// gather is left in the loop, only the list rows (4 bytes per lane, coalesced); (2) a workgroup still fills the CU's 16 wave
// slots (4 per SIMD), which the one-lane-per-target kernels with a full-record tile could not (their tile then leaves room
// for 384-512 targets = 6-8 waves, and they stall on LDS latency: measured 0.45-0.53 ms per launch for every tile record
// and workgroup size against 0.45 ms for the round-1 kernel); (3) a wave's trip count is the longest of 64 / LPT lists
// divided by LPT, not the longest of 64: fewer idle lane-trips.  Records sit at 96-byte stride; one extra 16-byte unit per
// eight records spreads a wave's scattered 16-byte reads over all banks (6 s mod 16 alone hits the even units only).  The LPT partial sums of a target are added in a fixed tree (lane order), so results are reproducible;
// they differ from the one-lane kernels' by summation order (parity tolerance, not bitwise).

template <int BS, int LPT, bool TAB>
__global__ __launch_bounds__(BS) void forces_q(PairConst pc, int32_t tcap, int32_t ngroups, const int32_t *__restrict__ plan,
                                               const int2 *__restrict__ deal,
                                               const double *__restrict__ frec, const int32_t *__restrict__ nlist,
                                               int32_t cap, const int32_t *__restrict__ ncount,
                                               const double *__restrict__ dw_tab, const double *__restrict__ sink, int64_t n,
                                               double *__restrict__ ax, double *__restrict__ ay, double *__restrict__ az,
                                               double *__restrict__ du, double *__restrict__ dalpha,
                                               const int32_t *__restrict__ orig, int32_t n_owned,
                                               const int32_t *__restrict__ wave_class, int32_t want, const int32_t *__restrict__ plan256) {
    // LPT = 4: groups of 256 targets, `plan` = plan_f (the groups the list entries are slots of), targets dealt by list length.
    // LPT = 8: groups of 128 targets (half the tile need: neighbourhoods of ~200 that do not fit otherwise), `plan` = the
    // half-group plans, plan256 = plan_f: an entry becomes a slot of the half group's tile with one constant per interval;
    // a trip is a whole list row (eight entries, one per lane); targets in their natural order.
    static_assert(LPT == 4 || LPT == 8, "a list row holds eight entries: two trips of four lanes or one of eight");
    constexpr int T = BS / LPT;
    constexpr int TPR = 8 / LPT;                           // trips per list row
    extern __shared__ __align__(16) double lds_dyn[];      // (16: the tile is read 16 bytes at a time; static LDS in front of it must not shift it by 8)
    double *lds_dw = lds_dyn;
    double2 *tile = reinterpret_cast<double2 *>(lds_dyn + (TAB ? (TAB_LDS(pc.nq)) : 0));      // !TAB: dW knots recomputed (density_wt)
    __shared__ double s_sink[4][MAX_SINKS];              // x, y, z, m of the sinks: the epilogue reads them here, not through serial scalar loads
    const int sub = threadIdx.x & (LPT - 1), tl = threadIdx.x / LPT;
    const double4 *fg = reinterpret_cast<const double4 *>(frec);
    // Waves w, w + 4, w + 8, w + 12 share a SIMD, and the deal hands out the targets longest list first, sixteen per wave: taken in
    // wave order SIMD 0 would get the longest sixteen of every quarter (19 % more trips than SIMD 3 for lists of 70 down to 25),
    // and the group lasts as long as its slowest SIMD.  Every second row of four waves takes its chunks in reverse.
    const int wv = threadIdx.x >> 6, chunk = (wv & 4) ? (wv ^ 3) : wv;
    const int rank = LPT == 4 ? chunk * 16 + (tl & 15) : tl;
    auto dealt = [&](int64_t g) -> int2 {                  // this thread's target in group g: {index within the group, list length or -1}
        const int64_t t = g * T + rank;
        if (t >= n) return make_int2(0, -1);
        if (LPT == 4) return deal[t];
        return make_int2(tl, orig[t] < n_owned ? min(ncount[t], cap) : -1);
    };
    // the dw table once per workgroup: the kernel is persistent, one workgroup per CU walks over many groups of T targets
    if (TAB) {
        for (int t = threadIdx.x; t < (TAB_LEN(pc.nq) >> 1); t += BS) reinterpret_cast<double2 *>(lds_dw)[t] = reinterpret_cast<const double2 *>(dw_tab)[t];
        if (threadIdx.x == 0 && (TAB_LEN(pc.nq) & 1)) lds_dw[TAB_LEN(pc.nq) - 1] = dw_tab[TAB_LEN(pc.nq) - 1];
    }
    // workgroups b, b + 8, .. share an XCD (round-robin dispatch, speed only): XCD x works on one contiguous eighth of the
    // groups, so that the up to nine workgroups that stage a record find it in that XCD's L2; its workgroups take the
    // groups of that eighth in turn
    const int nx = min(8, (int)gridDim.x), xcd = blockIdx.x % nx, per = ((int)gridDim.x - xcd + nx - 1) / nx;
    const int64_t g_hi = (int64_t)ngroups * (xcd + 1) / nx;
    const double inv_h = pc.inv_h, inv_dq = pc.inv_dq;
    auto dw_of = [&](double q) {
        const double t = knot_coord(q, inv_dq);
        return TAB ? table_knots_at(lds_dw, t) : knot_knots_at([&](int k) { return dw_knot(k, pc.dq, pc.nq); }, t);
    };
    // What only the epilogue needs of the kernel's arguments waits in LDS: 22 scalar registers less across the pair loop, which
    // has none to spare (the three fall-back variants of this kernel spilled 10-19 of them before)
    __shared__ EpiArgs s_epi;
    if (threadIdx.x == 0)
        s_epi = EpiArgs{pc.G, pc.alpha_floor, pc.alpha_decay, pc.inv_dwnorm, ax, ay, az, du, dalpha, pc.ns, pc.grav};
    if (threadIdx.x < 4 * MAX_SINKS) {
        const int row = threadIdx.x / MAX_SINKS, s = threadIdx.x % MAX_SINKS;
        s_sink[row][s] = s < pc.ns ? sink[(row == 3 ? 6 : row) * MAX_SINKS + s] : 0.0;
    }
    const SinkRows sk{s_sink[0], s_sink[1], s_sink[2], s_sink[3]};
    // slot tcap of the tile: the sentinel record (far away, massless, rho/2 = 1) of the idle lanes
    if (threadIdx.x == 0) {
        double2 *sp = tile + q_unit(tcap);
        sp[0] = make_double2(SENTINEL_POS, SENTINEL_POS); sp[1] = make_double2(SENTINEL_POS, 0.0); sp[2] = make_double2(0.0, 0.0);
        sp[3] = make_double2(0.0, 1.0); sp[4] = make_double2(0.0, 0.0); sp[5] = make_double2(0.0, 0.0);
    }
    // what does not need the tile is fetched one group ahead: the plan and the dealt target of this thread
    int64_t group = (int64_t)ngroups * xcd / nx + blockIdx.x / nx;
    TileMap tm_next;
    int2 deal_next = make_int2(0, -1);
    if (group < g_hi) {
        load_plan(plan, group, tm_next);
        deal_next = dealt(group);
    }
#ifdef SPH_PHASE_CLOCKS
    unsigned long long pa_n = 0;
    __shared__ unsigned long long s_phase[2][16];
    if (threadIdx.x < 32) s_phase[threadIdx.x >> 4][threadIdx.x & 15] = 0;
    const int ph_sel = (threadIdx.x >> 6) == 0 ? 0 : ((threadIdx.x >> 6) == BS / 64 - 1 ? 1 : -1);
    unsigned long long ph_last = __builtin_readcyclecounter();
#endif
    for (; group < g_hi; group += per) {
        const int64_t base = group * T;
        const TileMap tm = tm_next;
        const int2 dl = deal_next;
        if (group + per < g_hi) {
            load_plan(plan, group + per, tm_next);
            deal_next = dealt(group + per);
        }
        if (wave_class) {       // split evaluation (multi-GPU overlap): classes are per 64 targets
            bool any = false;
            for (int k = 0; k < T / 64; k++)
                any |= base + 64 * k < n && wave_class[(base >> 6) + k] == want;
            if (!any) continue;
        }
        const bool fits = tm.need <= tcap;
        PH_MARK(0);      // end of the previous trip (epilogue, prefetches) .. top
        // Targets are dealt to the waves in order of list length (deal_kernel, longest first): a wave's trip count is that
        // of its longest list, and 16 consecutive particles of a disc column span midplane and surface (mean 12 rows,
        // longest of 16: 19).  With every neighbour record in the tile the order costs nothing but the coalescing of the list
        // rows (4 bytes per lane).  Each target's sums are its own, so the order does not touch the results.
        const int64_t i = base + dl.x;
        const bool live = dl.y >= 0 && (!wave_class || wave_class[i >> 6] == want);
        const int self = i < n ? (int)i : (int)(n - 1);
        const int cnt = live ? dl.y : 0;
        // a trip = LPT consecutive entries of the target's list, one per lane.  LPT = 4: lane s reads word s of the list row (eight
        // 16-bit entries), its low half in the even trip, its high half in the odd one; LPT = 8: lane s reads word s & 3 and
        // takes half s >> 2, a row per trip (ent_pos, tile_common.hpp)
        const int ntrip = __builtin_amdgcn_readfirstlane(wave_max_i32((cnt + LPT - 1) / LPT));
        const int nrow = (ntrip + TPR - 1) / TPR;
        const uint32_t *lp = reinterpret_cast<const uint32_t *>(nlist) + (((size_t)(self >> 6) * (cap >> 3)) * 64 + (self & 63)) * 4 + (sub & 3);
        // the first two list rows are asked for ahead of the tile; the target's record is read from the tile: one round trip to
        // memory for the tile where there were four (tile in two trips, then the record, then the rows)
        uint32_t wa = lp[0];
        uint32_t wb = lp[(size_t)min(1, max(nrow, 1) - 1) * 256];
        pin_value(wa); pin_value(wb);           // (asked for before the barrier: the compiler would move the loads to their first use)
        PH_MARK(1);      // own list rows arrived
        __syncthreads();                        // the previous group's tile and sums are no longer read
        PH_MARK(2);      // barrier 1
        if (fits) stage_tile<BS, 8, 6, true>(reinterpret_cast<const double2 *>(frec), tile, tm);       // every load of the tile in flight at once
        PH_MARK(3);      // tile loads + writes
        __syncthreads();
        PH_MARK(4);      // barrier 2
        // the target's own record: it is in the tile (its cell is one of its neighbour cells) -- an LDS read, not a third trip to memory
        double4 A, B, Cc;
        if (fits) {
            const double2 *sp = q_record(tile, tm.slot(self));
            const double2 s0 = sp[0], s1 = sp[1], s2 = sp[2], s3 = sp[3], s4 = sp[4], s5 = sp[5];
            A = make_double4(s0.x, s0.y, s1.x, s1.y); B = make_double4(s2.x, s2.y, s3.x, s3.y); Cc = make_double4(s4.x, s4.y, s5.x, s5.y);
        } else {
            A = fg[(size_t)self * 3]; B = fg[(size_t)self * 3 + 1]; Cc = fg[(size_t)self * 3 + 2];
        }
        PH_MARK(5);      // own record out of the tile
        ForceSums f;
        auto ent_of = [&](uint32_t wd, int hf) { return (int)((wd >> ((LPT == 4 ? hf : (sub >> 2)) << 4)) & 0xffffu); };
        if (ntrip > 0) {
            // entries are slots of the 256-group's tile (LPT = 4: this tile) or, plus a constant per interval, of this half group's
            // tile / the sorted order
            if (fits) {
                EntryMap em{0, 0, 0, 0, 0};
                if (LPT != 4) {
                    em = entry_to_index(plan256, group >> 1);
                    em.a0 = (int)((unsigned)em.a0 + (unsigned)tm.base[0] - (unsigned)tm.lo[0]);
                    em.a1 = (int)((unsigned)em.a1 + (unsigned)tm.base[1] - (unsigned)tm.lo[1]);
                    em.a2 = (int)((unsigned)em.a2 + (unsigned)tm.base[2] - (unsigned)tm.lo[2]);
                }
                auto slot_of = [&](int e) { return LPT == 4 ? e : em(e); };       // LPT = 4: an entry IS the slot
                const double2 *rp = q_record(tile, sub < cnt ? slot_of(ent_of(wa, 0)) : tcap);
                double2 r0 = rp[0], r1 = rp[1], r2 = rp[2], r3 = rp[3], r4 = rp[4], r5 = rp[5];
                for (int r = 0; r < nrow; r++) {
                    const uint32_t wc = lp[(size_t)min(r + 2, nrow - 1) * 256];
#pragma unroll
                    for (int hf = 0; hf < TPR; hf++) {
                        if (TPR == 2 && hf == 1 && 2 * r + 1 >= ntrip) break;        // wave-uniform: the odd trip of the last row
                        const Nbr nb{r0.x, r0.y, r1.x, r1.y, r2.x, r2.y, r3.x, r3.y, r4.x, r4.y, r5.x};
                        const int k = LPT * (TPR * r + hf) + sub;
                        const int en = (TPR == 2 && hf == 0) ? ent_of(wa, 1) : ent_of(wb, 0);
                        rp = q_record(tile, k + LPT < cnt ? slot_of(en) : tcap);
                        // the next record is read behind this visit's two table knots: the LDS answers a wave in order, and the
                        // knots are what the visit waits for
                        force_visit<false>(pc, inv_h, A, B, Cc, nb, true, dw_of, f,
                                           [&]() { r0 = rp[0]; r1 = rp[1]; r2 = rp[2]; r3 = rp[3]; r4 = rp[4]; r5 = rp[5]; });
                    }
                    wa = wb; wb = wc;
                }
            } else {
                const EntryMap em = entry_to_index(plan256, LPT == 4 ? group : (group >> 1));
                int j = sub < cnt ? em(ent_of(wa, 0)) : self;
                double4 A1 = fg[(size_t)j * 3], B1 = fg[(size_t)j * 3 + 1], C1 = fg[(size_t)j * 3 + 2];
                for (int r = 0; r < nrow; r++) {
                    const uint32_t wc = lp[(size_t)min(r + 2, nrow - 1) * 256];
#pragma unroll
                    for (int hf = 0; hf < TPR; hf++) {
                        if (TPR == 2 && hf == 1 && 2 * r + 1 >= ntrip) break;
                        const Nbr nb = nbr_of(A1, B1, C1);
                        const int k = LPT * (TPR * r + hf) + sub;
                        const int en = (TPR == 2 && hf == 0) ? ent_of(wa, 1) : ent_of(wb, 0);
                        if (k + LPT < cnt) { j = em(en); A1 = fg[(size_t)j * 3]; B1 = fg[(size_t)j * 3 + 1]; C1 = fg[(size_t)j * 3 + 2]; }
                        force_visit(pc, inv_h, A, B, Cc, nb, k < cnt, dw_of, f);
                    }
                    wa = wb; wb = wc;
                }
            }
        }
        PH_MARK(6);      // pair loop
        // the target's sums, added in a fixed tree over its LPT lanes: the same value in all of them
#pragma unroll
        for (int o = 1; o < LPT; o <<= 1) {
            f.s0 += __shfl_xor(f.s0, o, 64); f.s1 += __shfl_xor(f.s1, o, 64); f.s2 += __shfl_xor(f.s2, o, 64);
            f.sdu += __shfl_xor(f.sdu, o, 64); f.sdal += __shfl_xor(f.sdal, o, 64);
        }
        PH_MARK(7);      // lane reduction
        // The epilogue, by the wave itself and at once -- no barrier, no hand-over through LDS: the waves that finish their lists
        // early (the oldest ones, which the SIMD serves first) do it while the others still walk theirs.  The first four lanes of
        // a target take one channel each (three acceleration components, du/dt) and store it; what the channels share (the
        // distance to each sink) is computed by all of them at the cost of one.
        if (live && sub < 4) {
            const EpiArgs ea = s_epi;
            PairConst pe;
            pe.G = ea.G; pe.alpha_floor = ea.alpha_floor; pe.alpha_decay = ea.alpha_decay; pe.inv_dwnorm = ea.inv_dwnorm; pe.ns = ea.ns;
            double *const out = sub == 0 ? ea.ax : (sub == 1 ? ea.ay : (sub == 2 ? ea.az : ea.du));
            const double start = (ea.grav && sub < 3) ? out[i] : 0.0;
            out[i] = force_channel(pe, sk, A, f, start, sub);
            if (sub == 3) ea.dalpha[i] = alpha_rate(pe, inv_h, B.w, Cc.x, Cc.y, f);
        }
#ifdef SPH_PHASE_CLOCKS
        pa_n++;
        PH_MARK(8);      // epilogue
#endif
    }
#ifdef SPH_PHASE_CLOCKS
    PH_MARK(0);
    if (threadIdx.x == 0) PHASE_ADD(0, pa_n);
    __syncthreads();
    if (threadIdx.x < 32) PHASE_ADD(32 + threadIdx.x, s_phase[threadIdx.x >> 4][threadIdx.x & 15]);
#endif
}


// What the host reads of a build, from the plan records nlist_tiled left (plan_f[8 g + ...] = {lo0, lo1, lo2, len0, len1, len2,
// need, longest list of the group}): the longest list, the groups of density_wt / forces_q whose tile does not fit, the largest
// tile need; and, for the whole-tile kernels, the plan of every 1024-target group of density_wt = the union of the plans of
// its four 256-target groups (plan_d).
// (Computing the intervals inside the evaluation kernels cost them 18 dependent cell-table reads per thread, two wave
// reductions and a barrier before the first byte could be staged: with one workgroup per CU nothing hides that.)
__global__ __launch_bounds__(1024) void plan_reduce_kernel(int64_t ngroups_d, int64_t ngroups_f, const int32_t *__restrict__ plan_f, int32_t tcap_d,
                                                           int32_t tcap_f, int32_t tcap_d_big, int32_t tcap_f_big, int32_t *__restrict__ plan_d,
                                                           const int32_t *__restrict__ plan_h, int32_t *__restrict__ report) {
    // ONE workgroup strides over the density groups (977 at 1e6 particles) and writes the four numbers straight into the host's
    // report slot (pinned memory mapped into the device's address space): no atomics, no device-to-host copy kernels
    __shared__ int s_red[8][16];
    int mx_list = 0, mx_need = 0, misfit_f = 0, misfit_d = 0, misfit_fb = 0, misfit_db = 0;      // ..b: against the table-free (bigger) tiles
    int misfit_h = 0, misfit_hb = 0;                                                              // half groups (128 targets) of forces_q
    for (int64_t gd = threadIdx.x; gd < ngroups_d; gd += 1024) {
        int lo[3] = {0x7fffffff, 0x7fffffff, 0x7fffffff}, hi[3] = {0, 0, 0};
        for (int64_t gf = 4 * gd; gf < std::min<int64_t>(4 * gd + 4, ngroups_f); gf++) {
            const int4 a = *reinterpret_cast<const int4 *>(plan_f + 8 * gf), b = *reinterpret_cast<const int4 *>(plan_f + 8 * gf + 4);
            const int l[3] = {a.x, a.y, a.z}, len[3] = {a.w, b.x, b.y};
            for (int q = 0; q < 3; q++)
                if (len[q] > 0) { lo[q] = min(lo[q], l[q]); hi[q] = max(hi[q], l[q] + len[q]); }
            mx_need = max(mx_need, b.z); mx_list = max(mx_list, b.w);
            misfit_f += b.z > tcap_f ? 1 : 0;
            misfit_fb += b.z > tcap_f_big ? 1 : 0;
            if (plan_h) {
                for (int hf = 0; hf < 2; hf++) {
                    const int nh = plan_h[8 * (2 * gf + hf) + 6];
                    misfit_h += nh > tcap_f ? 1 : 0; misfit_hb += nh > tcap_f_big ? 1 : 0;
                }
            }
        }
        if (plan_d) {
            int need = 0;
            int32_t *p = plan_d + 8 * (size_t)gd;
            for (int q = 0; q < 3; q++) {
                const int len = hi[q] > lo[q] ? hi[q] - lo[q] : 0;
                p[q] = lo[q]; p[3 + q] = len;
                need += len;
            }
            p[6] = need; p[7] = 0;
            misfit_d += need > tcap_d ? 1 : 0;
            misfit_db += need > tcap_d_big ? 1 : 0;
        }
    }
    for (int o = 32; o > 0; o >>= 1) {
        mx_list = max(mx_list, __shfl_xor(mx_list, o, 64)); mx_need = max(mx_need, __shfl_xor(mx_need, o, 64));
        misfit_f += __shfl_xor(misfit_f, o, 64); misfit_d += __shfl_xor(misfit_d, o, 64);
        misfit_fb += __shfl_xor(misfit_fb, o, 64); misfit_db += __shfl_xor(misfit_db, o, 64);
        misfit_h += __shfl_xor(misfit_h, o, 64); misfit_hb += __shfl_xor(misfit_hb, o, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        const int wv = threadIdx.x >> 6;
        s_red[0][wv] = mx_list; s_red[1][wv] = misfit_d; s_red[2][wv] = misfit_f; s_red[3][wv] = mx_need; s_red[4][wv] = misfit_db; s_red[5][wv] = misfit_fb;
        s_red[6][wv] = misfit_h; s_red[7][wv] = misfit_hb;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        int r0 = 0, r1 = 0, r2 = 0, r3 = 0, r4 = 0, r5 = 0, r6 = 0, r7 = 0;
        for (int k = 0; k < 16; k++) {
            r0 = max(r0, s_red[0][k]); r1 += s_red[1][k]; r2 += s_red[2][k]; r3 = max(r3, s_red[3][k]); r4 += s_red[4][k]; r5 += s_red[5][k];
            r6 += s_red[6][k]; r7 += s_red[7][k];
        }
        // {longest list, misfits density, misfits forces, largest need, misfits density / forces with the table-free tiles,
        //  misfits of the half groups of forces_q with the table / table-free tile}
        report[0] = r0; report[1] = r1; report[2] = r2; report[3] = r3; report[4] = r4; report[5] = r5; report[6] = r6; report[7] = r7;
    }
}

inline unsigned tb_blocks(int64_t n) { return (unsigned)((n + TB - 1) / TB); }

constexpr int FQ_T = 256;            // forces_q: targets per group

int32_t tile_cap(int nq, int rec, bool tablds) {
    const size_t tab = tablds ? (size_t)(TAB_LDS(nq)) * sizeof(double) : 0;
    if (tab + LDS_RESERVE >= (size_t)LDS_BYTES) return 0;
    return (int32_t)(((size_t)LDS_BYTES - LDS_RESERVE - tab) / ((size_t)rec * sizeof(double))) - 1;      // slot tcap: the sentinel record
}

// forces_q: 6 units of 16 bytes per record + 1 per eight records
int32_t tile_cap_q(int nq, bool tablds = true) {
    const size_t tab = tablds ? (size_t)(TAB_LDS(nq)) * sizeof(double) : 0;
    constexpr size_t reserve = 4096;            // static LDS of forces_q
    if (tab + reserve + 64 >= (size_t)LDS_BYTES) return 0;
    const size_t units = ((size_t)LDS_BYTES - reserve - tab) / 16 - 8;       // the sentinel record at q_unit(tcap): 6 units
    return (int32_t)((units * 8) / 49);
}


}  // namespace

#define TL_CHECK(expr)                                                      \
    do {                                                                    \
        hipError_t _e = (expr);                                             \
        if (_e != hipSuccess) {                                             \
            c->err = std::string(#expr) + ": " + hipGetErrorString(_e);     \
            return SPH_ERR_HIP;                                             \
        }                                                                   \
    } while (0)

// a context whose groups outgrow the 16-bit entries leaves the tiled path for good: the untiled 32-bit list of pairs.hip
static int leave_tiled_path(sph_ctx *c) {
    c->tiled = false; c->whole_tile = false; c->packed_list = false;
    c->wt_ok = c->wt_ok_f = false; c->wt_fit_pct = c->wt_fit_pct_f = -1;
    c->ring_nl_valid = false;
    ctx_free(c, c->nlist);
    if (ctx_alloc(c, &c->nlist, (size_t)c->nl_waves_cap * c->nl_cap * 64, "neighbour list") != SPH_OK) { c->nl_cap = 0; return SPH_ERR_NOMEM; }
    return nlist_build(c);
}

int nlist_build_tiled(sph_ctx *c) {
    const int64_t n = c->n;
    if (n == 0) return SPH_OK;
    const PairConst pc = make_pair_const(c);
    const unsigned d_blocks = (unsigned)((n + WT_BS - 1) / WT_BS), f_blocks = (unsigned)((n + FQ_T - 1) / FQ_T);
    auto regrow = [&](int32_t want) {
        ctx_free(c, c->nlist);
        c->nl_cap = (want + 7) & ~7;
        if (ctx_alloc(c, &c->nlist, (size_t)c->nl_waves_cap * c->nl_cap * 32, "neighbour list") != SPH_OK) { c->nl_cap = 0; return SPH_ERR_NOMEM; }     // 16-bit entries
        return SPH_OK;
    };
    // what a build reports: {longest list, misfits of the density geometry, misfits of the forces geometry, largest tile need}
    auto digest = [&](const int32_t *rep) {
        c->nl_max = rep[0];
        if (c->whole_tile) {
            // the whole-tile kernels pay a prologue and run their fall-back loop at one workgroup per CU: use them when
            // (nearly) every workgroup's intervals fit the tile -- thin discs and sheets; thick domains keep pairs.hip
            // ... with the kernel table beside the tile; where that fails but the table-free tile (40 KB larger, knots recomputed)
            // holds the intervals -- dense neighbourhoods -- the table-free variant runs (SPH_TILE_TABLE=regs: always, A/B)
            static const bool force_big = getenv("SPH_TILE_TABLE") && std::string(getenv("SPH_TILE_TABLE")) == "regs";
            const bool ok_d = (int64_t)rep[1] * 10 <= (int64_t)d_blocks, ok_db = (int64_t)rep[4] * 10 <= (int64_t)d_blocks;
            const bool ok_f = (int64_t)rep[2] * 10 <= (int64_t)f_blocks, ok_fb = (int64_t)rep[5] * 10 <= (int64_t)f_blocks;
            // forces: groups of 256 with the table (1.0), table-free (x1.1 on the bench disc), groups of 128 -- eight lanes per
            // target, every record staged more often -- with the table, table-free; the first whose tiles fit nine groups in ten
            static const int force_half = getenv("SPH_FORCES_HALF_GROUPS") ? atoi(getenv("SPH_FORCES_HALF_GROUPS")) : 0;     // A/B switch
            const bool ok_h = (int64_t)rep[6] * 10 <= 2 * (int64_t)f_blocks, ok_hb = (int64_t)rep[7] * 10 <= 2 * (int64_t)f_blocks;
            c->wt_big = force_big || (!ok_d && ok_db);
            c->wt_ok = c->wt_big ? ok_db : ok_d;
            c->wt_fit_pct = (int32_t)(100 - (100 * (int64_t)rep[c->wt_big ? 4 : 1]) / std::max<int64_t>(d_blocks, 1));
            c->wt_half_f = force_half != 0 || (!ok_f && !ok_fb && (ok_h || ok_hb));
            if (c->wt_half_f) {
                c->wt_big_f = force_big || !ok_h;
                c->wt_ok_f = c->wt_big_f ? ok_hb : ok_h;
                c->wt_fit_pct_f = (int32_t)(100 - (100 * (int64_t)rep[c->wt_big_f ? 7 : 6]) / std::max<int64_t>(2 * (int64_t)f_blocks, 1));
            } else {
                c->wt_big_f = force_big || (!ok_f && ok_fb);
                c->wt_ok_f = c->wt_big_f ? ok_fb : ok_f;
                c->wt_fit_pct_f = (int32_t)(100 - (100 * (int64_t)rep[c->wt_big_f ? 5 : 2]) / std::max<int64_t>(f_blocks, 1));
            }
        }
    };
    // Steady state: the report of the PREVIOUS build (it arrived long ago) is read instead of waiting for this one's.  The
    // list keeps a third of headroom, so a list that overflows within one step (which the dt control all but excludes) is
    // an error reported one build late, not a silent truncation.  The same for the 16-bit entries: the context leaves the
    // tiled path when a group's intervals reach HALF of what an entry can address.
    const bool trusted = c->ring_nl_valid && !c->no_stale;
    const int p = c->ring_nl;
    int32_t *slot = reinterpret_cast<int32_t *>(c->h_pinned + 240 + 8 * p);
    if (trusted) {
        TL_CHECK(hipEventSynchronize(c->ev_nl[1 - p]));
        const int32_t *prev = reinterpret_cast<const int32_t *>(c->h_pinned + 240 + 8 * (1 - p));
        if (prev[0] > c->nl_cap) { c->err = "neighbour list overflowed in the previous build (lists grew by more than a third within one step)"; return SPH_ERR_STATE; }
        if (prev[3] >= LIST16_MAX_NEED) { c->err = "neighbour list: a group's candidate intervals outgrew the 16-bit entries within one step"; return SPH_ERR_STATE; }
        if (2 * (int64_t)prev[3] >= LIST16_MAX_NEED) return leave_tiled_path(c);
        digest(prev);
        if (4 * (int64_t)prev[0] > 3 * (int64_t)c->nl_cap) { const int st = regrow(prev[0] + prev[0] / 2 + 8); if (st != SPH_OK) return st; }
    }
    for (int attempt = 0; attempt < 8; attempt++) {
        nlist_tiled<<<dim3(tb_blocks(n)), dim3(TB), 0, c->stream>>>(c->grid, reinterpret_cast<const double4 *>(c->drec), c->cell_start,
                                                                    n, pc.rcut2, c->nl_cap, reinterpret_cast<int4 *>(c->nlist),
                                                                    c->ncount, c->wave_max, c->d_flags, c->orig, (int32_t)c->n_owned,
                                                                    c->whole_tile ? reinterpret_cast<int2 *>(c->deal) : nullptr,
                                                                    c->plan_f, c->whole_tile ? c->plan_h : nullptr);
        TL_CHECK(hipGetLastError());
        {
            const int64_t ngd = (n + WT_BS - 1) / WT_BS, ngf = (n + FQ_T - 1) / FQ_T;
            plan_reduce_kernel<<<dim3(1), dim3(1024), 0, c->stream>>>(
                ngd, ngf, c->plan_f, c->whole_tile ? tile_cap(pc.nq, 4, true) : 0, c->whole_tile ? tile_cap_q(pc.nq) : LIST16_MAX_NEED,
                c->whole_tile ? tile_cap(pc.nq, 4, false) : 0, c->whole_tile ? tile_cap_q(pc.nq, false) : LIST16_MAX_NEED,
                c->whole_tile ? c->plan_d : nullptr, c->whole_tile ? c->plan_h : nullptr, slot);
            TL_CHECK(hipGetLastError());
        }
        TL_CHECK(hipEventRecord(c->ev_nl[p], c->stream));
        if (trusted) break;
        TL_CHECK(hipStreamSynchronize(c->stream));
        c->host_syncs++;
        if (2 * (int64_t)slot[3] >= LIST16_MAX_NEED) return leave_tiled_path(c);
        digest(slot);
        const int32_t mx = slot[0];
        if (4 * (int64_t)mx <= 3 * (int64_t)c->nl_cap) break;            // fits, with the headroom the steady state relies on
        if (attempt == 7) { c->err = "neighbour list did not converge"; return SPH_ERR_STATE; }
        { const int st = regrow(mx + mx / 2 + 8); if (st != SPH_OK) return st; }
    }
    c->ring_nl = 1 - p; c->ring_nl_valid = true;
    c->nlist_builds++;
    return SPH_OK;
}

// list entries / lane-trips of the forces kernel in use, for a list with the given lengths (sorted-slot order)
double forces_lane_efficiency(const std::vector<int32_t> &cnt) {
    // forces_q: groups of 256 dealt by length, waves of 16 targets x 4 lanes, a trip = a row of four
    double entries = 0.0, lane_trips = 0.0;
    std::vector<int32_t> g;
    for (size_t b = 0; b < cnt.size(); b += 256) {
        g.assign(cnt.begin() + b, cnt.begin() + std::min(cnt.size(), b + 256));
        std::sort(g.begin(), g.end(), std::greater<int32_t>());
        for (size_t w = 0; w < g.size(); w += 16) lane_trips += 64.0 * ((g[w] + 3) / 4);
        for (int32_t v : g) entries += v;
    }
    return lane_trips > 0.0 ? entries / lane_trips : 0.0;
}

// ---- whole-tile kernels -------------------------------------------------------------------------------------
// Workgroups of the persistent kernels: one per CU.  With several ranks four CUs stay free: a persistent workgroup holds the
// whole register file of its CU (128 VGPRs x 16 waves), so nothing else starts beside it -- measured with the sinks' acceleration
// on a second stream, which simply ran after forces_q -- and the send/receive kernels of the halo exchange are meant to run
// beside the density pass and the interior wavefronts of the forces (1.6 % fewer CUs for the pair kernels).
static unsigned persistent_grid(const sph_ctx *c, int64_t ngroups) {
    const int cus = std::max(c->num_cus - (c->nranks > 1 ? 4 : 0), 8);
    return (unsigned)std::min<int64_t>(ngroups, cus);
}

template <bool TAB>
static hipError_t density_wt_launch(sph_ctx *c, const PairConst &pc) {
    const int32_t tcap = tile_cap(pc.nq, 4, TAB);
    const size_t lds = (TAB ? (size_t)(TAB_LDS(pc.nq)) * sizeof(double) : 0) + ((size_t)tcap + 1) * sizeof(double4);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&density_wt<WT_BS, TAB>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    const int64_t ngroups = (c->n + WT_BS - 1) / WT_BS;
    const unsigned grid = persistent_grid(c, ngroups);
    density_wt<WT_BS, TAB><<<dim3(grid), dim3(WT_BS), lds, c->stream>>>(
        pc, tcap, (int32_t)ngroups, c->plan_d, c->plan_f, reinterpret_cast<const double4 *>(c->drec), c->nlist, c->nl_cap, c->ncount, c->wave_max,
        c->w_tab, c->n, c->f[SPH_F_U], c->f[SPH_F_ALPHA], c->f[SPH_F_VX],
        c->f[SPH_F_VY], c->f[SPH_F_VZ], c->f[SPH_F_RHO], c->f[SPH_F_P], c->f[SPH_F_C], c->frec, c->orig, (int32_t)c->n_owned);
    return hipGetLastError();
}

hipError_t launch_density_wt(sph_ctx *c, const PairConst &pc) {
    if (c->n == 0) return hipSuccess;
    return c->wt_big ? density_wt_launch<false>(c, pc) : density_wt_launch<true>(c, pc);
}

template <int LPT, bool TAB>
static hipError_t forces_q_launch(sph_ctx *c, const PairConst &pc, int part) {
    constexpr int BS = 1024;
    const int32_t tcap = tile_cap_q(pc.nq, TAB);
    const size_t lds = (TAB ? (size_t)(TAB_LDS(pc.nq)) * sizeof(double) : 0) + ((size_t)tcap * 6 + (tcap >> 3) + 8) * sizeof(double2);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&forces_q<BS, LPT, TAB>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    constexpr int T = BS / LPT;
    const int64_t ngroups = (c->n + T - 1) / T;
    const unsigned grid = persistent_grid(c, ngroups);
    forces_q<BS, LPT, TAB><<<dim3(grid), dim3(BS), lds, c->stream>>>(
        pc, tcap, (int32_t)ngroups, LPT == 4 ? c->plan_f : c->plan_h, LPT == 4 ? reinterpret_cast<const int2 *>(c->deal) : nullptr, c->frec, c->nlist, c->nl_cap,
        c->ncount, c->dw_tab, c->sink, c->n, c->f[SPH_F_AX],
        c->f[SPH_F_AY], c->f[SPH_F_AZ], c->f[SPH_F_DU], c->f[SPH_F_DALPHA], c->orig, (int32_t)c->n_owned,
        part ? c->wave_class : nullptr, part == 2 ? 1 : 0, c->plan_f);
    return hipGetLastError();
}

// part 0: every wave.  part 1 / 2: only the waves of class 0 (interior) / class 1, as launch_forces
hipError_t launch_forces_wt(sph_ctx *c, const PairConst &pc, int part) {
    if (c->n == 0) return hipSuccess;
    if (c->wt_half_f) return c->wt_big_f ? forces_q_launch<8, false>(c, pc, part) : forces_q_launch<8, true>(c, pc, part);
    return c->wt_big_f ? forces_q_launch<4, false>(c, pc, part) : forces_q_launch<4, true>(c, pc, part);
}

}  // namespace sph

#ifdef SPH_PHASE_CLOCKS
// profiling build only: read (and clear) the phase counters
extern "C" int sph_debug_phase_clocks(unsigned long long *out64) {
    if (hipMemcpyFromSymbol(out64, HIP_SYMBOL(sph::g_phase_clocks), 64 * sizeof(unsigned long long)) != hipSuccess) return 1;
    unsigned long long zero[64] = {};
    return hipMemcpyToSymbol(HIP_SYMBOL(sph::g_phase_clocks), zero, sizeof(zero)) == hipSuccess ? 0 : 1;
}
#endif
