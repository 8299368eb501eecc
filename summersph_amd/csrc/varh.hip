// varh.hip -- the variable-h path: per-particle smoothing length, grad-h terms, the reference's
// leaf-box neighbour rule and the h update.
//
// Replaces (citations: "/root/reference/SUMMER_SPH - Variable.f90", "[V]")
//   create_tree / build_tree (geometry only)    [V]:999-1020,163-267 -> leaf_keys, leaf_boxes
//   density_tree_search / get_density (rho, Om) [V]:440-496          -> density_v_kernel
//   get_pressure_and_sound_speed(gamma)         [V]:502-512          -> density_v_kernel epilogue
//   SPH_tree_search / get_SPH (grad-h form)     [V]:324-432          -> forces_v_kernel
//   calc_smoothing                              [V]:515-546          -> update_h_kernel
//
// The neighbour rule of [V] is not a sphere test.  The tree walk of a body at
// x reaches particle j's one-particle leaf iff on every axis |x - c_leaf(j)| < 2 h_j + edge_leaf(j)/2;
// the kernel then uses the BODY's h (density) or both (forces), and a force pair {a,b} (a > b by
// particle number) is evaluated iff a's walk reaches b's leaf.  To reproduce that, every particle's
// leaf box of the reference octree (bbox-midpoint root, edge = largest extent, strict '>' split) is
// computed here WITHOUT building the tree: the path of a particle down the octree is a 3-bit-per-
// level key; after a radix sort of those keys a particle's leaf level is 1 + its longest common
// prefix with its sorted neighbours, and the leaf centre follows from replaying the same
// centre +- edge/4 additions the reference performs (bitwise the same box).
//
// Neighbour search: uniform grid of edge 2 <h> (mean h) with a per-cell maximum of h; a particle
// scans the cells C with dist(x, C) <= 2 max(h_i, hmax_C), which finds every j with
// r <= 2 max(h_i, h_j) without making the cells as large as the largest h.
#include <algorithm>
#include <cstdlib>
#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>

#include <cmath>

#include "pair_common.hpp"

namespace sph {

namespace {

constexpr int VBLOCK = 256;
constexpr int LEVELS = 21;            // 63-bit path keys
constexpr uint32_t FLAG_D = 0x80000000u, FLAG_F = 0x40000000u, FLAG_R = 0x20000000u, IDX_MASK = 0x1fffffffu;
// calc_smoothing re-evaluates rho with a trial length h' > h: the list also keeps (behind the D/F entries, from the
// last row of the lane's column downwards) the particles whose leaf the body's walk reaches within 2 h (1 + margin)
constexpr double H_MARGIN = 1.1;

struct RootBox { double c[3]; double size; };

__device__ __forceinline__ double wave_maxd(double v) {
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ double wave_sumd(double v) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// ---- h statistics -------------------------------------------------------------------------------
// The re-flag pass (below) stands in for a list build at unchanged positions while no h has grown by more than this since
// the build (shrinking only removes pairs; a measured optimum: 1.05 is used less often, 1.1 makes build and margin larger)
constexpr double REFLAG_GROW = 1.07;

// max h, sum h and -- with h_prev, the lengths the current list was built with -- the largest growth h / h_prev and the
// largest shrinkage (stored as the largest h_prev / h)
__global__ __launch_bounds__(VBLOCK) void h_stats_partial(const double *__restrict__ h, const double *__restrict__ h_prev, int64_t n,
                                                          double *__restrict__ part) {
    __shared__ double sm[4][VBLOCK / WAVE];
    double mx = 0.0, su = 0.0, gr = 0.0, sh = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * VBLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * VBLOCK) {
        mx = fmax(mx, h[i]); su += h[i];
        if (h_prev) { gr = fmax(gr, h[i] / h_prev[i]); sh = fmax(sh, h_prev[i] / h[i]); }
    }
    mx = wave_maxd(mx); su = wave_sumd(su); gr = wave_maxd(gr); sh = wave_maxd(sh);
    if ((threadIdx.x & 63) == 0) { sm[0][threadIdx.x >> 6] = mx; sm[1][threadIdx.x >> 6] = su; sm[2][threadIdx.x >> 6] = gr; sm[3][threadIdx.x >> 6] = sh; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int k = 1; k < VBLOCK / WAVE; k++) { mx = fmax(mx, sm[0][k]); su += sm[1][k]; gr = fmax(gr, sm[2][k]); sh = fmax(sh, sm[3][k]); }
        part[4 * blockIdx.x] = mx; part[4 * blockIdx.x + 1] = su; part[4 * blockIdx.x + 2] = gr; part[4 * blockIdx.x + 3] = sh;
    }
}

__global__ void h_stats_final(const double *__restrict__ part, int nb, double *__restrict__ out) {
    double mx = 0.0, su = 0.0, gr = 0.0, sh = 0.0;
    for (int b = threadIdx.x; b < nb; b += 64) { mx = fmax(mx, part[4 * b]); su += part[4 * b + 1]; gr = fmax(gr, part[4 * b + 2]); sh = fmax(sh, part[4 * b + 3]); }
    mx = wave_maxd(mx); su = wave_sumd(su); gr = wave_maxd(gr); sh = wave_maxd(sh);
    if (threadIdx.x == 0) { out[0] = mx; out[1] = su; out[2] = gr; out[3] = sh; }
}

// ---- octree leaf boxes ----------------------------------------------------------------------------
// path of the particle down the reference's octree: child index = (x > cx) | (y > cy) << 1 | (z > cz) << 2
// ([V]:228-236), child centre = centre +- edge/4, child edge = edge/2 ([V]:212-222)
__global__ __launch_bounds__(VBLOCK) void leaf_keys(RootBox rb, const double4 *__restrict__ prec, int64_t n,
                                                    uint64_t *__restrict__ keys, uint32_t *__restrict__ vals) {
    const int64_t i = (int64_t)blockIdx.x * VBLOCK + threadIdx.x;
    if (i >= n) return;
    const double4 p = prec[i];
    double cx = rb.c[0], cy = rb.c[1], cz = rb.c[2], size = rb.size;
    uint64_t key = 0;
    for (int l = 0; l < LEVELS; l++) {
        const int bx = p.x > cx, by = p.y > cy, bz = p.z > cz;
        key = (key << 3) | (uint64_t)(bx | (by << 1) | (bz << 2));
        const double q = 0.25 * size;
        cx = cx + (bx ? q : -q); cy = cy + (by ? q : -q); cz = cz + (bz ? q : -q);
        size = size * 0.5;
    }
    keys[i] = key;
    vals[i] = (uint32_t)i;
}

__device__ __forceinline__ int common_levels(uint64_t a, uint64_t b) {
    const uint64_t x = a ^ b;
    if (x == 0) return LEVELS;
    return (__clzll((long long)x) - 1) / 3;
}

// s = position in key order.  lrec[slot] = {leaf centre, reach = 2 h + edge/2}
__global__ __launch_bounds__(VBLOCK) void leaf_boxes(RootBox rb, const uint64_t *__restrict__ keys, const uint32_t *__restrict__ vals,
                                                     int64_t n, const double4 *__restrict__ prec, double4 *__restrict__ lrec,
                                                     double *__restrict__ leaf_half) {
    const int64_t s = (int64_t)blockIdx.x * VBLOCK + threadIdx.x;
    if (s >= n) return;
    const uint64_t key = keys[s];
    int cp = 0;
    if (s > 0) cp = max(cp, common_levels(key, keys[s - 1]));
    if (s + 1 < n) cp = max(cp, common_levels(key, keys[s + 1]));
    // alone in the root box (n == 1): the root itself is the leaf; otherwise one level below the deepest
    // node shared with another particle
    const int level = n == 1 ? 0 : min(cp + 1, LEVELS);
    double cx = rb.c[0], cy = rb.c[1], cz = rb.c[2], size = rb.size;
    for (int l = 1; l <= level; l++) {
        const int ch = (int)((key >> (3 * (LEVELS - l))) & 7);
        const double q = 0.25 * size;
        cx = cx + ((ch & 1) ? q : -q); cy = cy + ((ch & 2) ? q : -q); cz = cz + ((ch & 4) ? q : -q);
        size = size * 0.5;
    }
    if (cp == LEVELS && n > 1) {
        // Keys identical over all 21 levels: somebody is closer than edge / 2^21 on every axis.  The reference goes on
        // splitting (to depth 1000, [V]:8,203); so does this branch, from the positions themselves: the deepest level this
        // particle shares with any member of the run of equal keys (a run is two or three particles), then its own path
        // down to one level below.  (cx, cy, cz, size) is the common level-21 cell here.  Coincident points never
        // separate: they end at level 1000 (the reference leaves them in a childless two-particle node its walks skip).
        const double4 ps = prec[vals[s]];
        int deepest = 0;
        for (int dir = -1; dir <= 1; dir += 2)
            for (int64_t t = s + dir; t >= 0 && t < n && keys[t] == key; t += dir) {
                const double4 pt = prec[vals[t]];
                double ax = cx, ay = cy, az = cz, as = size;
                int extra = 0;
                for (; extra < 1000 - LEVELS - 1; extra++) {
                    const int cs = (ps.x > ax) | ((ps.y > ay) << 1) | ((ps.z > az) << 2);
                    const int ct = (pt.x > ax) | ((pt.y > ay) << 1) | ((pt.z > az) << 2);
                    if (cs != ct) break;
                    const double q = 0.25 * as;
                    ax = ax + ((cs & 1) ? q : -q); ay = ay + ((cs & 2) ? q : -q); az = az + ((cs & 4) ? q : -q);
                    as = as * 0.5;
                }
                deepest = max(deepest, extra);
            }
        for (int l = 0; l <= deepest; l++) {              // levels 22 .. 22 + deepest of this particle's own path
            const int ch = (ps.x > cx) | ((ps.y > cy) << 1) | ((ps.z > cz) << 2);
            const double q = 0.25 * size;
            cx = cx + ((ch & 1) ? q : -q); cy = cy + ((ch & 2) ? q : -q); cz = cz + ((ch & 4) ? q : -q);
            size = size * 0.5;
        }
    }
    const uint32_t slot = vals[s];
    const double h = prec[slot].w;
    lrec[slot] = make_double4(cx, cy, cz, 2.0 * h + size / 2.0);          // [V]:380,479
    leaf_half[slot] = size / 2.0;
}

// Multi-GPU: the octree is that of ALL GPUs' particles.  gkeys = their sorted path keys (n_glob); a local slot finds its
// own key there (it is one of them) and takes its leaf level from the neighbours in THAT order.
__global__ __launch_bounds__(VBLOCK) void leaf_boxes_ext(RootBox rb, const uint64_t *__restrict__ gkeys, int64_t n_glob,
                                                         int64_t n, const double4 *__restrict__ prec, double4 *__restrict__ lrec,
                                                         double *__restrict__ leaf_half) {
    const int64_t i = (int64_t)blockIdx.x * VBLOCK + threadIdx.x;
    if (i >= n) return;
    const double4 p = prec[i];
    double cx = rb.c[0], cy = rb.c[1], cz = rb.c[2], size = rb.size;
    uint64_t key = 0;
    for (int l = 0; l < LEVELS; l++) {
        const int bx = p.x > cx, by = p.y > cy, bz = p.z > cz;
        key = (key << 3) | (uint64_t)(bx | (by << 1) | (bz << 2));
        const double q = 0.25 * size;
        cx = cx + (bx ? q : -q); cy = cy + (by ? q : -q); cz = cz + (bz ? q : -q);
        size = size * 0.5;
    }
    int64_t lo = 0, hi = n_glob;
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (gkeys[mid] < key) lo = mid + 1; else hi = mid;
    }
    int cp = 0;
    if (lo > 0) cp = max(cp, common_levels(key, gkeys[lo - 1]));
    if (lo + 1 < n_glob) cp = max(cp, common_levels(key, gkeys[lo + 1]));
    const int level = n_glob == 1 ? 0 : min(cp + 1, LEVELS);
    cx = rb.c[0]; cy = rb.c[1]; cz = rb.c[2]; size = rb.size;
    for (int l = 1; l <= level; l++) {
        const int ch = (int)((key >> (3 * (LEVELS - l))) & 7);
        const double q = 0.25 * size;
        cx = cx + ((ch & 1) ? q : -q); cy = cy + ((ch & 2) ? q : -q); cz = cz + ((ch & 4) ? q : -q);
        size = size * 0.5;
    }
    lrec[i] = make_double4(cx, cy, cz, 2.0 * p.w + size / 2.0);          // [V]:380,479
    leaf_half[i] = size / 2.0;
}

// only h changed since the last build (calc_smoothing at the end of a step: same positions, same sorted order, same
// leaf cells): new h into the gather records, new reaches 2 h + edge/2 from the stored half edges
__global__ __launch_bounds__(VBLOCK) void refresh_h_records(int64_t n, const double *__restrict__ h, const double *__restrict__ leaf_half,
                                                            double4 *__restrict__ prec, double4 *__restrict__ lrec) {
    const int64_t i = (int64_t)blockIdx.x * VBLOCK + threadIdx.x;
    if (i >= n) return;
    const double hi = h[i];
    prec[i].w = hi;
    lrec[i].w = 2.0 * hi + leaf_half[i];                                  // [V]:380,479
}

__global__ __launch_bounds__(VBLOCK) void cell_hmax_kernel(const int32_t *__restrict__ cell_start, int64_t ncells,
                                                           const double4 *__restrict__ prec, double *__restrict__ hmax) {
    const int64_t c = (int64_t)blockIdx.x * VBLOCK + threadIdx.x;
    if (c >= ncells) return;
    double m = 0.0;
    for (int j = cell_start[c]; j < cell_start[c + 1]; j++) m = fmax(m, prec[j].w);
    hmax[c] = 4.0 * m * m * (1.0 + 1e-12);          // stored as the squared reach (2 hmax)^2 (1 + 1e-12), what the build compares with
}

__device__ __forceinline__ bool reaches(const double4 &leaf, double x, double y, double z) {
    return ((int)(fabs(x - leaf.x) < leaf.w) & (int)(fabs(y - leaf.y) < leaf.w) & (int)(fabs(z - leaf.z) < leaf.w)) != 0;     // no short circuit: no branches
}

// distance^2 from coordinate p to the cell interval [lo, lo+e] along one axis
__device__ __forceinline__ double axis_gap2(double p, double lo, double e) {
    const double d = fmax(fmax(lo - p, p - (lo + e)), 0.0);
    return d * d;
}

// ---- neighbour list ---------------------------------------------------------------------------------
// entry = j | FLAG_D (j counts in i's density sum) | FLAG_F (pair {i,j} counts in the force sums) | FLAG_R (i's walk
// reaches j's leaf; what update_h needs to know for a trial h)
// layout: 4-packed, wave-strided (entry k of lane l in wave w = component k%4 of the int4 at
// nlist4[(w*cap/4 + k/4)*64 + l]); the evaluation kernels read it in lockstep.

// Built from LDS-staged tiles.  For every offset o2 along the slowest grid axis
// the candidates of a 256-particle workgroup lie in ONE contiguous interval of the sorted order (all
// columns within R of the workgroup's columns); it is staged chunk-wise with coalesced loads and each lane
// scans its own cells out of LDS instead of gathering every candidate through the TA.
constexpr int T_NV = 512;           // candidates per staged chunk: {x,y,z,h}, leaf box and id of each (68 B)

// one chunk [cb, ce) of the sorted order into LDS: {x, y, z, (2 h_j)^2 (1 + 1e-12)}, the leaf box and the particle number.
// Every test needs h_j only through that square, and x -> 4 x x c is monotone in floating point, so max(square_i, square_j)
// is bitwise the square of max(h_i, h_j)
__device__ __forceinline__ void stage_chunk(double4 *tile, double4 *tile_l, int32_t *tile_o, const double4 *__restrict__ prec,
                                            const double4 *__restrict__ lrec, const int32_t *__restrict__ orig,
                                            const int32_t *__restrict__ number, int cb, int ce) {
    for (int t = threadIdx.x; t < ce - cb; t += VBLOCK) {
        double4 v = prec[cb + t];
        v.w = 4.0 * v.w * v.w * (1.0 + 1e-12);
        tile[t] = v; tile_l[t] = lrec[cb + t]; tile_o[t] = number ? number[orig[cb + t]] : orig[cb + t];
    }
}

// the reference's rule for one candidate j of body i (both directions), written once for the build and the re-flag pass:
// rij = i's walk reaches j's leaf; D = [V]:479 + kernel support of h_i; F = [V]:383 (the higher-numbered partner's walk decides)
__device__ __forceinline__ void pair_flags(const double4 &pi, const double4 &li, int oi, double ri2, const double4 &pj, const double4 &lj,
                                           int oj, double r2, bool &inD, bool &inF, bool &rij) {
    rij = reaches(lj, pi.x, pi.y, pi.z);
    inD = ((int)rij & (int)(r2 <= ri2)) != 0;
    const bool rji = reaches(li, pj.x, pj.y, pj.z);
    inF = ((int)(r2 <= fmax(ri2, pj.w)) & (int)(oi > oj ? rij : rji)) != 0;
}

// grow2 = (largest growth of any h the list shall survive)^2: candidates within 2 grow max(h_i, h_j) that count for nothing
// now are kept in the margin shell too, so that nlist_v_reflag finds every pair a new h can switch on.
__global__ __launch_bounds__(VBLOCK) void nlist_v_tiled(GridDesc g, int R, double h_glob, double grow2, const double4 *__restrict__ prec,
                                                        const double4 *__restrict__ lrec, const int32_t *__restrict__ orig,
                                                        const int32_t *__restrict__ cell_start, const double *__restrict__ cell_hmax,
                                                        int64_t n, int32_t n_owned, int32_t cap, int32_t *__restrict__ nlist,
                                                        int32_t *__restrict__ ncount, int32_t *__restrict__ ntail,
                                                        int32_t *__restrict__ wave_max, int32_t *__restrict__ wave_need,
                                                        const int32_t *__restrict__ number) {
    __shared__ double4 tile[T_NV], tile_l[T_NV];
    __shared__ int32_t tile_o[T_NV];
    __shared__ int s_lo[VBLOCK / WAVE], s_hi[VBLOCK / WAVE];
    const int64_t blk = xcd_chunk(blockIdx.x, gridDim.x);
    const int64_t i = blk * VBLOCK + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const int64_t w = i >> 6;
    const bool live = i < n && orig[i] < n_owned;
    const int64_t self = i < n ? i : n - 1;
    const double4 pi = prec[self], li = lrec[self];
    // the reference's particle number decides which partner's walk counts for a force pair ([V]:383); on several
    // GPUs the caller supplies the global numbers (sph_set_numbers_dev), else it is the context's own numbering
    const int oi = number ? number[orig[self]] : orig[self];
    const double p[3] = {pi.x, pi.y, pi.z};
    int cc[3];
    cell_coords(g, pi.x, pi.y, pi.z, cc);
    const int s0 = g.s[0], s1 = g.s[1], s2 = g.s[2];
    const int d0 = g.dim[s0], d1 = g.dim[s1], d2 = g.dim[s2];
    const double e = 1.0 / g.inv_edge;
    const double hi = pi.w, him = pi.w * H_MARGIN;
    const double rg = 2.0 * fmax(him, h_glob), rg2 = rg * rg * (1.0 + 1e-12);
    const double ri2 = 4.0 * hi * hi * (1.0 + 1e-12), rim2 = 4.0 * him * him * (1.0 + 1e-12);
    const double rig2 = grow2 * ri2;                      // (2 grow h_i)^2: the re-flag margin on the body's own side
    const int c1lo = max(cc[1] - R, 0), c1hi = min(cc[1] + R, d1 - 1);
    const int c0lo = max(cc[0] - R, 0), c0hi = min(cc[0] + R, d0 - 1);
    const int cap4 = cap >> 2;
    int4 *mine = reinterpret_cast<int4 *>(nlist) + ((size_t)w * cap4) * 64 + lane;
    int4 buf = make_int4(0, 0, 0, 0), tbuf = make_int4(0, 0, 0, 0);
    int cnt = 0, tcnt = 0;

    for (int o2 = -R; o2 <= R; o2++) {
        const int c2 = cc[2] + o2;
        const bool in2 = live && c2 >= 0 && c2 < d2;
        const double g2 = in2 ? axis_gap2(p[s2], g.org[s2] + c2 * e, e) : 0.0;
        const bool use2 = in2 && g2 <= rg2;
        // the workgroup's interval for this offset
        int mn = 0x7fffffff, mx = 0;
        if (use2) {
            mn = cell_start[((int64_t)c2 * d1 + c1lo) * d0 + c0lo];
            mx = cell_start[((int64_t)c2 * d1 + c1hi) * d0 + c0hi + 1];
            if (mx <= mn) { mn = 0x7fffffff; mx = 0; }
        }
        for (int o = 32; o > 0; o >>= 1) { mn = min(mn, __shfl_xor(mn, o, 64)); mx = max(mx, __shfl_xor(mx, o, 64)); }
        __syncthreads();
        if (lane == 0) { s_lo[threadIdx.x >> 6] = mn; s_hi[threadIdx.x >> 6] = mx; }
        __syncthreads();
        const int lo = min(min(s_lo[0], s_lo[1]), min(s_lo[2], s_lo[3]));
        const int hiv = max(max(s_hi[0], s_hi[1]), max(s_hi[2], s_hi[3]));
        for (int cb = lo; cb < hiv; cb += T_NV) {
            const int ce = min(cb + T_NV, hiv);
            __syncthreads();
            stage_chunk(tile, tile_l, tile_o, prec, lrec, orig, number, cb, ce);
            __syncthreads();
            if (!use2) continue;
            // per-lane walk over this lane's own columns and cells (lanes of different columns advance in
            // parallel; a wave-uniform walk with scalar table loads was measured 2x slower because waves that
            // straddle two columns then serialise)
            for (int c1 = c1lo; c1 <= c1hi; c1++) {
                const double g21 = g2 + axis_gap2(p[s1], g.org[s1] + c1 * e, e);
                if (g21 > rg2) continue;
                const int64_t row = ((int64_t)c2 * d1 + c1) * d0;
                int js = cell_start[row + c0lo];
                if (cell_start[row + c0hi + 1] <= cb || js >= ce) continue;                       // row not in this chunk
                for (int c0 = c0lo; c0 <= c0hi; c0++) {
                    const int jn = cell_start[row + c0 + 1];             // one table read per cell: the end is the next start
                    const int jb = max(js, cb), je = min(jn, ce);
                    js = jn;
                    if (jb >= je) continue;
                    const double gap = g21 + axis_gap2(p[s0], g.org[s0] + c0 * e, e);
                    if (gap > fmax(rim2, grow2 * cell_hmax[row + c0])) continue;     // (2 max(1.1 h_i, grow hmax_C))^2 (1 + 1e-12)
                    for (int j = jb; j < je; j++) {
                        const double4 pj = tile[j - cb];
                        const double dx = pi.x - pj.x, dy = pi.y - pj.y, dz = pi.z - pj.z;
                        const double r2 = dx * dx + dy * dy + dz * dz;
                        if (r2 <= fmax(rim2, grow2 * pj.w) && j != (int)i) {
                            bool inD, inF, rij;
                            pair_flags(pi, li, oi, ri2, pj, tile_l[j - cb], tile_o[j - cb], r2, inD, inF, rij);
                            const int ent = (int32_t)((uint32_t)j | (inD ? FLAG_D : 0u) | (inF ? FLAG_F : 0u) | (rij ? FLAG_R : 0u));
                            if (inD || inF) {
                                const int q4 = cnt & 3;             // selects, not branches
                                buf.x = q4 == 0 ? ent : buf.x; buf.y = q4 == 1 ? ent : buf.y; buf.z = q4 == 2 ? ent : buf.z; buf.w = q4 == 3 ? ent : buf.w;
                                if (q4 == 3 && cnt < cap) mine[(size_t)(cnt >> 2) * 64] = buf;
                                cnt++;
                            } else if ((rij && r2 <= rim2) || r2 <= fmax(rig2, grow2 * pj.w)) {
                                // margin shell: a trial h of calc_smoothing (reached, within 2.2 h_i) or a grown h (within
                                // 2 grow max(h_i, h_j), reached or not: the reaches grow too) can need it
                                const int q4 = tcnt & 3;
                                tbuf.x = q4 == 0 ? ent : tbuf.x; tbuf.y = q4 == 1 ? ent : tbuf.y; tbuf.z = q4 == 2 ? ent : tbuf.z; tbuf.w = q4 == 3 ? ent : tbuf.w;
                                if (q4 == 3 && tcnt < cap) mine[(size_t)(cap4 - 1 - (tcnt >> 2)) * 64] = tbuf;
                                tcnt++;
                            }
                        }
                    }
                }
            }
        }
    }
    if ((cnt & 3) != 0 && cnt < cap) mine[(size_t)(cnt >> 2) * 64] = buf;
    if ((tcnt & 3) != 0 && tcnt < cap) mine[(size_t)(cap4 - 1 - (tcnt >> 2)) * 64] = tbuf;
    if (i < n) { ncount[i] = live ? cnt : 0; ntail[i] = live ? tcnt : 0; }
    const int wm = wave_max_i32(live ? cnt : 0);
    // rows needed: the D/F entries from the top, the margin entries from the bottom of the lane's column
    const int need = wave_max_i32(live ? 4 * (((cnt + 3) >> 2) + ((tcnt + 3) >> 2)) : 0);
    if (lane == 0 && (w << 6) < n) {
        wave_max[w] = min(wm, cap);
        wave_need[w] = need;          // reduced by max_to_host (an atomicMax per wave on ONE address is serialised at the memory side)
    }
}

// largest of n ints -> *host_out (pinned host memory mapped into the device's address space); one workgroup
__global__ __launch_bounds__(1024) void max_to_host(const int32_t *__restrict__ v, int64_t n, int32_t *__restrict__ host_out) {
    __shared__ int s_red[16];
    int m = 0;
    for (int64_t k = threadIdx.x; k < n; k += 1024) m = max(m, v[k]);
    m = wave_max_i32(m);
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int k = 1; k < 16; k++) m = max(m, s_red[k]);
        *host_out = m;
    }
}

// Same positions, same sorted order, new h (calc_smoothing ran): the list of the new lengths out of the list of the old
// ones, in place.  The D/F entries and the margin shell of a column are together every pair that a growth of any h by up
// to `grow` can switch on, and every pair that was on.  Pass 1 re-evaluates the D/F entries where they stand (an entry
// that counts for nothing any more keeps its slot without flags: a masked trip); pass 2 walks the margin rows from the
// lowest row upwards and appends what the new lengths switch on behind the D/F entries -- writes never pass the reads,
// because the build left rows(D/F) + rows(margin) <= rows of the column.  The wave walks its 64 columns in lock-step
// like the evaluation kernels (rows two ahead, the candidate's record one entry ahead); the leaf box is fetched only for
// candidates between the two kernel supports, the particle number only when the two walks disagree.
// Both passes are software pipelines: the candidate's record is requested one entry ahead, classified when it arrives
// (outside both supports: drop; inside both: D, F and R hold, see reflag_classify; in between: exact test), the leaf box and
// the particle number of the in-between class are requested then and used one entry later -- no trip waits for a
// memory round trip of its own.
struct ReflagStage {          // an entry between classification and its flags
    int j, cls;               // cls 0: nothing holds, 1: everything holds, 2: decide with the leaf box
    bool d, rji;              // r <= 2 h_i;  j's walk reaches i's leaf
    double4 lj;               // j's leaf box (cls 2)
    int oj;                   // orig[j] (cls 2)
};

__device__ __forceinline__ void reflag_classify(const double4 &pi, const double4 &li, double ri2, bool act, int j, const double4 &pj_raw,
                                                const double4 *__restrict__ lrec, const int32_t *__restrict__ orig, ReflagStage &st) {
    const double pjw = 4.0 * pj_raw.w * pj_raw.w * (1.0 + 1e-12);          // as stage_chunk
    const double dx = pi.x - pj_raw.x, dy = pi.y - pj_raw.y, dz = pi.z - pj_raw.z;
    const double r2 = dx * dx + dy * dy + dz * dz;
    // Inside both kernel supports (cls 1): a leaf box {|x - c| < 2 h + e/2} contains the sphere of radius 2 h around its own
    // particle (which lies within e/2 of c), so both walks reach the other's leaf and D and F hold without looking at a
    // box.  The 1 % keeps rounding at the sphere's surface out of this class (those pairs take the exact test).
    st.cls = !act ? 0 : (r2 < 0.99 * fmin(ri2, pjw) ? 1 : (r2 <= fmax(ri2, pjw) ? 2 : 0));
    st.j = j;
    st.d = r2 <= ri2;
    st.rji = reaches(li, pj_raw.x, pj_raw.y, pj_raw.z);
    if (st.cls == 2) { st.lj = lrec[j]; st.oj = orig[j]; }
}

// the entry with its new flags (pair_flags' rule)
__device__ __forceinline__ int reflag_finish(const double4 &pi, int oi, const ReflagStage &st, const int32_t *__restrict__ number) {
    bool inD = st.cls == 1, inF = st.cls == 1, rij = st.cls == 1;
    if (st.cls == 2) {
        rij = reaches(st.lj, pi.x, pi.y, pi.z);
        inD = ((int)rij & (int)st.d) != 0;                                  // [V]:479 + kernel support of h_i
        inF = rij;
        if (rij != st.rji) inF = oi > (number ? number[st.oj] : st.oj) ? rij : st.rji;      // [V]:383
    }
    return (int32_t)((uint32_t)st.j | (inD ? FLAG_D : 0u) | (inF ? FLAG_F : 0u) | (rij ? FLAG_R : 0u));
}

__global__ __launch_bounds__(VBLOCK) void nlist_v_reflag(const double4 *__restrict__ prec, const double4 *__restrict__ lrec,
                                                         const int32_t *__restrict__ orig, int64_t n, int32_t n_owned, int32_t cap,
                                                         int32_t *__restrict__ nlist, int32_t *__restrict__ ncount,
                                                         const int32_t *__restrict__ ntail, int32_t *__restrict__ wave_max,
                                                         const int32_t *__restrict__ number) {
    const int64_t i = (int64_t)xcd_chunk(blockIdx.x, gridDim.x) * VBLOCK + threadIdx.x;
    if ((i & ~(int64_t)63) >= n) return;
    const int lane = threadIdx.x & 63;
    const int64_t w = i >> 6;
    const bool live = i < n && orig[i] < n_owned;
    const int self = i < n ? (int)i : (int)(n - 1);
    const double4 pi = prec[self], li = lrec[self];
    const int oi = number ? number[orig[self]] : orig[self];
    const double ri2 = 4.0 * pi.w * pi.w * (1.0 + 1e-12);
    const int cap4 = cap >> 2;
    int4 *mine = reinterpret_cast<int4 *>(nlist) + ((size_t)w * cap4) * 64 + lane;
    const int cnt_a = live ? min(ncount[i], cap) : 0, tcnt_a = live ? ntail[i] : 0;

    // pass 1: the D/F entries, in place
    int4 last = make_int4(0, 0, 0, 0);                      // the row that holds entry cnt_a (where pass 2 appends)
    const int kmax = wave_max_i32(cnt_a);
    if (kmax > 0) {
        const int nrow = (kmax + 3) >> 2;
        int4 qa = load_row(mine);
        int4 qb = load_row(mine + (size_t)min(1, nrow - 1) * 64);
        int e1 = 0 < cnt_a ? qa.x : self;
        double4 p1 = prec[e1 & IDX_MASK];
        ReflagStage st;
        st.j = 0; st.cls = 0; st.d = false; st.rji = false; st.lj = li; st.oj = 0;
        int4 outp = make_int4(0, 0, 0, 0);                  // the previous row, waiting for its fourth entry
        int4 qprev = make_int4(0, 0, 0, 0);                 // ... as it stands in memory: most rows come out unchanged and are not written
        auto same = [](const int4 &a, const int4 &b) { return ((a.x ^ b.x) | (a.y ^ b.y) | (a.z ^ b.z) | (a.w ^ b.w)) == 0; };
        for (int r = 0; r < nrow; r++) {
            const int4 qc = load_row(mine + (size_t)min(r + 2, nrow - 1) * 64);
            int4 out = make_int4(0, 0, 0, 0);
#pragma unroll
            for (int v = 0; v < 4; v++) {
                const int k = 4 * r + v;
                // the entry classified one trip ago gets its flags
                if (v == 0) {
                    if (r > 0) {
                        outp.w = reflag_finish(pi, oi, st, number);
                        if (4 * (r - 1) < cnt_a && !same(outp, qprev)) mine[(size_t)(r - 1) * 64] = outp;
                        if (r - 1 == (cnt_a >> 2)) last = outp;
                    }
                } else {
                    const int ent = reflag_finish(pi, oi, st, number);
                    if (v == 1) out.x = ent; else if (v == 2) out.y = ent; else out.z = ent;
                }
                // this trip's entry: its record has arrived; the next one's is requested
                const int e0 = e1;
                const double4 pj = p1;
                if (k + 1 < cnt_a) {
                    e1 = v < 3 ? comp4(qa, v + 1) : qb.x;
                    p1 = prec[e1 & IDX_MASK];
                }
                reflag_classify(pi, li, ri2, k < cnt_a, e0 & IDX_MASK, pj, lrec, orig, st);
            }
            outp = out; qprev = qa;
            qa = qb; qb = qc;
        }
        outp.w = reflag_finish(pi, oi, st, number);
        if (4 * (nrow - 1) < cnt_a && !same(outp, qprev)) mine[(size_t)(nrow - 1) * 64] = outp;
        if (nrow - 1 == (cnt_a >> 2)) last = outp;
    }
    // pass 2: the margin shell, rows in ascending row index (= descending entry number); what the new lengths switch on is
    // appended behind the D/F entries
    int cnt = cnt_a;
    int4 buf = last;
    const int rows_t = (tcnt_a + 3) >> 2;
    const int smax = wave_max_i32(rows_t);
    int4 qn = rows_t > 0 ? mine[(size_t)(cap4 - rows_t) * 64] : make_int4(self, self, self, self);
    for (int s = 0; s < smax; s++) {
        const int tr = rows_t - 1 - s;                      // this lane's margin row (entries 4 tr .. 4 tr + 3), if it has one left
        const bool has = tr >= 0;
        const int4 q = qn;
        if (tr >= 1) qn = mine[(size_t)(cap4 - tr) * 64];   // the next row (tr - 1) one step ahead; rows are never written before they are read
        // the four records of the row in flight together, then the leaf boxes of those between the supports, then the flags
        double4 pr[4];
#pragma unroll
        for (int v = 0; v < 4; v++) pr[v] = prec[(has && 4 * tr + v < tcnt_a) ? (comp4(q, v) & IDX_MASK) : self];
        ReflagStage sv[4];
#pragma unroll
        for (int v = 0; v < 4; v++) reflag_classify(pi, li, ri2, has && 4 * tr + v < tcnt_a, comp4(q, v) & IDX_MASK, pr[v], lrec, orig, sv[v]);
#pragma unroll
        for (int v = 0; v < 4; v++) {
            const int ent = reflag_finish(pi, oi, sv[v], number);
            if ((uint32_t)ent & (FLAG_D | FLAG_F)) {
                const int q4 = cnt & 3;
                buf.x = q4 == 0 ? ent : buf.x; buf.y = q4 == 1 ? ent : buf.y; buf.z = q4 == 2 ? ent : buf.z; buf.w = q4 == 3 ? ent : buf.w;
                if (q4 == 3) mine[(size_t)(cnt >> 2) * 64] = buf;
                cnt++;
            }
        }
    }
    if (cnt > cnt_a && (cnt & 3) != 0) mine[(size_t)(cnt >> 2) * 64] = buf;
    if (i < n) ncount[i] = live ? cnt : 0;
    const int wm = wave_max_i32(live ? cnt : 0);
    if (lane == 0) wave_max[w] = min(wm, cap);
}

// ---- the variable-h pair terms, written once (gather kernels and tile kernels call these) ---------------------------
struct DensSumsV { double s1 = 0.0, s2 = 0.0; };     // sum m_j w(q),  sum m_j (q dw(q) - 3 w(q))

__device__ __forceinline__ void density_visit_v(const double4 &pi, const double4 &pj, bool act, const double *__restrict__ lw,
                                                const double *__restrict__ ldw, double inv_h, double inv_dq, int nq, DensSumsV &d) {
    const double n0 = pi.x - pj.x, n1 = pi.y - pj.y, n2 = pi.z - pj.z;              // [V]:481
    double dr, rs;
    rsqrt_sqrt(n0 * n0 + n1 * n1 + n2 * n2, dr, rs);                                 // [V]:482
    const double qi = dr * inv_h;
    if (act && qi <= 2.0) {
        double wl, dwl;
        table_lerp2(lw, ldw, qi, inv_dq, nq, wl, dwl);                               // [V]:486
        d.s1 = fma(pj.w, wl, d.s1);                                                  // [V]:492
        d.s2 = fma(pj.w, qi * dwl - 3.0 * wl, d.s2);                                 // [V]:487,493 (x -pi h^4)
    }
}

// rho, Omega, EOS and the force record of particle i from its sums (self term: the walk reaches the body's own leaf, r = 0)
__device__ __forceinline__ void density_epilogue_v(const PairConst &pc, int64_t i, const double4 &pi, double hi, DensSumsV d, double w0,
                                                   const double *__restrict__ u, const double *__restrict__ alpha,
                                                   const double *__restrict__ vx, const double *__restrict__ vy,
                                                   const double *__restrict__ vz, double *__restrict__ rho, double *__restrict__ omega,
                                                   double *__restrict__ P, double *__restrict__ cs, double *__restrict__ frec) {
    d.s1 = fma(pi.w, w0, d.s1);
    d.s2 = fma(pi.w, -3.0 * w0, d.s2);
    const double h3 = hi * hi * hi;
    const double rhoi = d.s1 / (pc.kernel_pi * h3);                                      // [V]:139
    const double om_acc = -d.s2 / (pc.kernel_pi * ((hi * hi) * (hi * hi)));             // [V]:140,487
    const double omi = 1.0 + (hi / (3.0 * rhoi)) * om_acc;                               // [V]:455
    const double Pi = pc.gamma_m1 * u[i] * rhoi;                                         // [V]:509
    const double ci = sqrt(pc.gamma * Pi / rhoi);                                        // [V]:510
    rho[i] = rhoi; omega[i] = omi; P[i] = Pi; cs[i] = ci;
    write_frec(frec, i, pi, vx[i], vy[i], vz[i], rhoi, Pi / (omi * rhoi * rhoi), ci, alpha[i], hi);           // [V]:413
}

// one visit of the grad-h force sums, [V]:385-427.  A, B, C: the target's record (C.w = h_i); inv_n4i = 1 / (pi h_i^4)
__device__ __forceinline__ void force_visit_v(const PairConst &pc, double hi, double inv_h, double inv_dq, double inv_pi, double inv_n4i,
                                              const double *__restrict__ lds_dw, const double4 &A, const double4 &B, const double4 &Cc,
                                              const double4 &Aj, const double4 &Bj, const double4 &Cj, bool act, ForceSums &f) {
    const double n0 = A.x - Aj.x, n1 = A.y - Aj.y, n2 = A.z - Aj.z;                   // [V]:385
    const double r2 = n0 * n0 + n1 * n1 + n2 * n2;
    double dr, rs;
    rsqrt_sqrt(r2, dr, rs);
    if (act && r2 > 0.0) {
        const double hj = Cj.w;
        const double inv_hj = fast_rcp(hj);
        const double qo = dr * inv_h, qn = dr * inv_hj;
        const double ihj2 = inv_hj * inv_hj;
        // dW(r, h_i) and dW(r, h_j), each normalised with its own h ([V]:395-396,140)
        const double dWo = qo <= 2.0 ? table_lerp(lds_dw, qo, inv_dq, pc.nq) * inv_n4i : 0.0;
        const double dWn = qn <= 2.0 ? table_lerp(lds_dw, qn, inv_dq, pc.nq) * (ihj2 * ihj2 * inv_pi) : 0.0;
        const double v0 = B.x - Bj.x, v1 = B.y - Bj.y, v2 = B.z - Bj.z;               // [V]:387
        const double vr = v0 * n0 + v1 * n1 + v2 * n2;
        const double vdotr = fmin(vr, 0.0);                                           // [V]:388-390
        const double dWs = 0.5 * (dWo + dWn);
        const double vdotgradW = (vr * rs) * dWs;                                     // [V]:401
        const double avg_len = 0.5 * (hi + hj);                                       // [V]:402
        double inv_r2e, inv_rho;                                                      // one reciprocal for the two denominators
        rcp_pair(r2 + pc.visc_eps_h2 * avg_len * avg_len, B.w + Bj.w, inv_r2e, inv_rho);
        const double vis_nu = (avg_len * vdotr) * inv_r2e;                            // [V]:405
        const double cbar = Cc.x + Cj.x, abar = Cc.y + Cj.y;
        const double visc = (abar * vis_nu) * (2.0 * vis_nu - cbar) * inv_rho;        // [V]:410
        const double S = (Cc.z * dWo + Cj.z * dWn + visc * dWs) * rs;                 // [V]:413-414 (along n)
        const double mS = Aj.w * S;
        f.s0 = fma(mS, n0, f.s0); f.s1 = fma(mS, n1, f.s1); f.s2 = fma(mS, n2, f.s2); // [V]:416
        const double mv = Aj.w * vdotgradW;
        f.sdu = fma(mv, Cc.z + 0.5 * visc, f.sdu);                                    // [V]:419-421
        f.sdal += mv;                                                                 // [V]:427
    }
}

// rates of particle i from its sums (variable h): sink gravity, the alpha rate of [V]:346
__device__ __forceinline__ void force_epilogue_v(const PairConst &pc, const double *__restrict__ sink, int64_t i, const double4 &A,
                                                 double rho_half, double c_half, double al_half, double hi, const ForceSums &f,
                                                 double *__restrict__ ax, double *__restrict__ ay, double *__restrict__ az,
                                                 double *__restrict__ du, double *__restrict__ dalpha) {
    double a0, a1, a2;
    sink_gas_accel(pc, sink, A, i, ax, ay, az, a0, a1, a2);      // zero_rates, [self-gravity], sink_gravforces ([V]:1028-1030, 691-)
    ax[i] = a0 - f.s0; ay[i] = a1 - f.s1; az[i] = a2 - f.s2;
    du[i] = f.sdu;
    dalpha[i] = fmax(f.sdal / (2.0 * rho_half), 0.0) + pc.alpha_decay * ((pc.alpha_floor - 2.0 * al_half) * (2.0 * c_half) / hi);
}

// ---- density + Omega + EOS ---------------------------------------------------------------------------
__global__ __launch_bounds__(VBLOCK) void density_v_kernel(PairConst pc, const double4 *__restrict__ drec,
                                                           const int32_t *__restrict__ nlist, int32_t cap,
                                                           const int32_t *__restrict__ ncount, const int32_t *__restrict__ wave_max,
                                                           const double *__restrict__ w_tab, const double *__restrict__ dw_tab,
                                                           int64_t n, const double *__restrict__ hh, const double *__restrict__ u,
                                                           const double *__restrict__ alpha, const double *__restrict__ vx,
                                                           const double *__restrict__ vy, const double *__restrict__ vz,
                                                           double *__restrict__ rho, double *__restrict__ omega,
                                                           double *__restrict__ P, double *__restrict__ cs, double *__restrict__ frec,
                                                           const int32_t *__restrict__ orig, int32_t n_owned) {
    extern __shared__ double lds[];
    double *lw = lds, *ldw = lds + TAB_LEN(pc.nq);
    for (int k = threadIdx.x; k < TAB_LEN(pc.nq); k += VBLOCK) { lw[k] = w_tab[k]; ldw[k] = dw_tab[k]; }
    __syncthreads();

    const int64_t i = (int64_t)xcd_chunk(blockIdx.x, gridDim.x) * VBLOCK + threadIdx.x;
    if ((i & ~(int64_t)63) >= n) return;
    const int lane = threadIdx.x & 63;
    const int64_t w = i >> 6;
    const bool live = i < n && orig[i] < n_owned;
    const int self = i < n ? (int)i : (int)(n - 1);
    const double4 pi = drec[self];
    const double hi = hh[self];
    const int cnt = live ? min(ncount[i], cap) : 0;
    const int kmax = __builtin_amdgcn_readfirstlane(wave_max[w]);
    const int4 *mine4 = reinterpret_cast<const int4 *>(nlist) + ((size_t)w * (cap >> 2)) * 64 + lane;
    const double inv_h = 1.0 / hi, inv_dq = 0.5 * pc.nq;

    // list rows as int4 (four entries), streamed two rows ahead with wave-uniform loads: a quarter of the list
    // instructions of a per-entry read, and the rows do not displace the gather records from the caches
    DensSumsV d;
    if (kmax > 0) {
        const int nrow = (kmax + 3) >> 2;
        int4 qa = load_row(mine4);
        int4 qb = load_row(mine4 + (size_t)min(1, nrow - 1) * 64);
        int e1 = 0 < cnt ? qa.x : self;
        double4 p1 = drec[e1 & IDX_MASK];
        for (int r = 0; r < nrow; r++) {
            const int4 qc = load_row(mine4 + (size_t)min(r + 2, nrow - 1) * 64);
#pragma unroll
            for (int v = 0; v < 4; v++) {                 // whole rows, no trip-count test (tiled.hip density_wt)
                const int k = 4 * r + v;
                const double4 pj = p1;
                const bool act = k < cnt && ((uint32_t)e1 & FLAG_D);
                if (k + 1 < cnt) {                              // idle lanes issue no gather, nor do entries that count for forces only
                    e1 = v < 3 ? comp4(qa, v + 1) : qb.x;
                    if ((uint32_t)e1 & FLAG_D) p1 = drec[e1 & IDX_MASK];
                }
                density_visit_v(pi, pj, act, lw, ldw, inv_h, inv_dq, pc.nq, d);
            }
            qa = qb; qb = qc;
        }
    }
    if (!live) return;
    density_epilogue_v(pc, i, pi, hi, d, lw[0], u, alpha, vx, vy, vz, rho, omega, P, cs, frec);
}

// ---- re-flag pass and density pass in ONE walk over the list (start of a step: only h is newer than the list) ---------
// nlist_v_reflag and density_v_kernel walk the same rows and gather the same neighbours one after the other; here every
// entry is visited once: its flags for the new lengths are settled as in nlist_v_reflag (and written back for forces_v),
// and when they include D the entry is added to the density sums on the spot.  Pass 1 = the D/F entries where they stand,
// pass 2 = the margin shell, whose promoted entries are appended behind them (and summed too).  The order in which a
// particle's neighbours are summed is the list's, with promoted entries last -- as for a re-flagged list.
// Measured (1e6 bench disc, ms per start of step): separate passes 0.985 + 0.33, this kernel 1.20.  It is bound by the latency
// of its dependent gathers at three waves per SIMD (160 VGPRs), not by their number: a variant in which the build marked the
// entries whose flags cannot change while every h stays within 3 % (no gather for them at all; forced on for the timing) ran
// 1.25 ms and cost the build 0.13 ms per step for the marking -- out of the tree again.
struct FusedStage {           // an entry between classification and its flags
    ReflagStage st;
    double4 pm;               // {x, y, z, m} of the partner (for the density visit)
};

__global__ __launch_bounds__(VBLOCK) void reflag_density_kernel(PairConst pc, const double4 *__restrict__ prec, const double4 *__restrict__ lrec,
                                                                const double4 *__restrict__ drec, const double *__restrict__ mass,
                                                                const int32_t *__restrict__ orig, int64_t n, int32_t n_owned, int32_t cap,
                                                                int32_t *__restrict__ nlist, int32_t *__restrict__ ncount,
                                                                const int32_t *__restrict__ ntail, int32_t *__restrict__ wave_max,
                                                                const int32_t *__restrict__ number, const double *__restrict__ w_tab,
                                                                const double *__restrict__ dw_tab, const double *__restrict__ u,
                                                                const double *__restrict__ alpha, const double *__restrict__ vx,
                                                                const double *__restrict__ vy, const double *__restrict__ vz,
                                                                double *__restrict__ rho, double *__restrict__ omega,
                                                                double *__restrict__ P, double *__restrict__ cs, double *__restrict__ frec) {
    extern __shared__ double lds[];
    double *lw = lds, *ldw = lds + TAB_LEN(pc.nq);
    for (int k = threadIdx.x; k < TAB_LEN(pc.nq); k += VBLOCK) { lw[k] = w_tab[k]; ldw[k] = dw_tab[k]; }
    __syncthreads();
    const int64_t i = (int64_t)xcd_chunk(blockIdx.x, gridDim.x) * VBLOCK + threadIdx.x;
    if ((i & ~(int64_t)63) >= n) return;
    const int lane = threadIdx.x & 63;
    const int64_t w = i >> 6;
    const bool live = i < n && orig[i] < n_owned;
    const int self = i < n ? (int)i : (int)(n - 1);
    const double4 pi = prec[self], li = lrec[self];
    const double4 pim = drec[self];                           // {x, y, z, m_i}: the density sums are formed as density_v_kernel forms them
    const int oi = number ? number[orig[self]] : orig[self];
    const double hi = pi.w;
    const double ri2 = 4.0 * hi * hi * (1.0 + 1e-12);
    const double inv_h = 1.0 / hi, inv_dq = 0.5 * pc.nq;
    const int cap4 = cap >> 2;
    int4 *mine = reinterpret_cast<int4 *>(nlist) + ((size_t)w * cap4) * 64 + lane;
    const int cnt_a = live ? min(ncount[i], cap) : 0, tcnt_a = live ? ntail[i] : 0;
    DensSumsV d;

    // the record an entry needs: {x, y, z, h_j} to settle its flags, m_j for the sum
    auto fetch = [&](int e, bool act, double4 &rec, double &mj) {
        const int j = e & IDX_MASK;
        rec = prec[act ? j : self];
        mj = act ? mass[j] : 0.0;
    };
    auto classify = [&](int e, bool act, const double4 &rec, double mj, FusedStage &fs) {
        reflag_classify(pi, li, ri2, act, e & IDX_MASK, rec, lrec, orig, fs.st);
        fs.pm = make_double4(rec.x, rec.y, rec.z, mj);
    };
    // flags of the entry classified a trip ago; a D entry joins the sums here
    auto finish = [&](const FusedStage &fs) -> int {
        const int ent = reflag_finish(pi, oi, fs.st, number);
        density_visit_v(pim, fs.pm, ((uint32_t)ent & FLAG_D) != 0, lw, ldw, inv_h, inv_dq, pc.nq, d);
        return ent;
    };

    // pass 1: the D/F entries, in place
    int4 last = make_int4(0, 0, 0, 0);                      // the row that holds entry cnt_a (where pass 2 appends)
    const int kmax = wave_max_i32(cnt_a);
    if (kmax > 0) {
        const int nrow = (kmax + 3) >> 2;
        int4 qa = load_row(mine);
        int4 qb = load_row(mine + (size_t)min(1, nrow - 1) * 64);
        int e1 = 0 < cnt_a ? qa.x : self;
        double4 p1; double m1;
        fetch(e1, 0 < cnt_a, p1, m1);
        FusedStage fs;
        fs.st.j = 0; fs.st.cls = 0; fs.st.d = false; fs.st.rji = false; fs.st.lj = li; fs.st.oj = 0; fs.pm = pim;
        int4 outp = make_int4(0, 0, 0, 0);                  // the previous row, waiting for its fourth entry
        int4 qprev = make_int4(0, 0, 0, 0);                 // ... as it stands in memory: most rows come out unchanged and are not written
        auto same = [](const int4 &a, const int4 &b) { return ((a.x ^ b.x) | (a.y ^ b.y) | (a.z ^ b.z) | (a.w ^ b.w)) == 0; };
        for (int r = 0; r < nrow; r++) {
            const int4 qc = load_row(mine + (size_t)min(r + 2, nrow - 1) * 64);
            int4 out = make_int4(0, 0, 0, 0);
#pragma unroll
            for (int v = 0; v < 4; v++) {
                const int k = 4 * r + v;
                if (v == 0) {
                    if (r > 0) {
                        outp.w = finish(fs);
                        if (4 * (r - 1) < cnt_a && !same(outp, qprev)) mine[(size_t)(r - 1) * 64] = outp;
                        if (r - 1 == (cnt_a >> 2)) last = outp;
                    }
                } else {
                    const int ent = finish(fs);
                    if (v == 1) out.x = ent; else if (v == 2) out.y = ent; else out.z = ent;
                }
                const int e0 = e1;
                const double4 pj = p1;
                const double mj = m1;
                if (k + 1 < cnt_a) {
                    e1 = v < 3 ? comp4(qa, v + 1) : qb.x;
                    fetch(e1, true, p1, m1);
                }
                classify(e0, k < cnt_a, pj, mj, fs);
            }
            outp = out; qprev = qa;
            qa = qb; qb = qc;
        }
        outp.w = finish(fs);
        if (4 * (nrow - 1) < cnt_a && !same(outp, qprev)) mine[(size_t)(nrow - 1) * 64] = outp;
        if (nrow - 1 == (cnt_a >> 2)) last = outp;
    }
    // pass 2: the margin shell, rows in ascending row index; what the new lengths switch on is appended behind the D/F
    // entries and, where it counts for the density, summed
    int cnt = cnt_a;
    int4 buf = last;
    const int rows_t = (tcnt_a + 3) >> 2;
    const int smax = wave_max_i32(rows_t);
    int4 qn = rows_t > 0 ? mine[(size_t)(cap4 - rows_t) * 64] : make_int4(self, self, self, self);
    for (int s = 0; s < smax; s++) {
        const int tr = rows_t - 1 - s;
        const bool has = tr >= 0;
        const int4 q = qn;
        if (tr >= 1) qn = mine[(size_t)(cap4 - tr) * 64];
        double4 pr[4];
        bool on[4];
#pragma unroll
        for (int v = 0; v < 4; v++) {
            const int e = comp4(q, v);
            on[v] = has && 4 * tr + v < tcnt_a;
            pr[v] = prec[on[v] ? (e & IDX_MASK) : self];
        }
        ReflagStage sv[4];
#pragma unroll
        for (int v = 0; v < 4; v++) reflag_classify(pi, li, ri2, on[v], comp4(q, v) & IDX_MASK, pr[v], lrec, orig, sv[v]);
#pragma unroll
        for (int v = 0; v < 4; v++) {
            const int ent = reflag_finish(pi, oi, sv[v], number);
            if ((uint32_t)ent & (FLAG_D | FLAG_F)) {
                if ((uint32_t)ent & FLAG_D)
                    density_visit_v(pim, make_double4(pr[v].x, pr[v].y, pr[v].z, mass[ent & IDX_MASK]), true, lw, ldw, inv_h, inv_dq, pc.nq, d);
                const int q4 = cnt & 3;
                buf.x = q4 == 0 ? ent : buf.x; buf.y = q4 == 1 ? ent : buf.y; buf.z = q4 == 2 ? ent : buf.z; buf.w = q4 == 3 ? ent : buf.w;
                if (q4 == 3) mine[(size_t)(cnt >> 2) * 64] = buf;
                cnt++;
            }
        }
    }
    if (cnt > cnt_a && (cnt & 3) != 0) mine[(size_t)(cnt >> 2) * 64] = buf;
    if (i < n) ncount[i] = live ? cnt : 0;
    const int wm = wave_max_i32(live ? cnt : 0);
    if (lane == 0) wave_max[w] = min(wm, cap);
    if (!live) return;
    density_epilogue_v(pc, i, pim, hi, d, lw[0], u, alpha, vx, vy, vz, rho, omega, P, cs, frec);
}

__global__ __launch_bounds__(256) void eos_only_v_kernel(PairConst pc, int64_t n, const double4 *__restrict__ drec,
                                                         const double *__restrict__ hh, const double *__restrict__ u,
                                                         const double *__restrict__ alpha, const double *__restrict__ vx,
                                                         const double *__restrict__ vy, const double *__restrict__ vz,
                                                         const double *__restrict__ rho, const double *__restrict__ omega,
                                                         double *__restrict__ P, double *__restrict__ cs, double *__restrict__ frec) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const double r = rho[i];
    const double Pi = pc.gamma_m1 * u[i] * r;
    const double ci = sqrt(pc.gamma * Pi / r);
    P[i] = Pi; cs[i] = ci;
    write_frec(frec, i, drec[i], vx[i], vy[i], vz[i], r, Pi / (omega[i] * r * r), ci, alpha[i], hh[i]);
}

// ---- forces, grad-h form ---------------------------------------------------------------------------------
__global__ __launch_bounds__(VBLOCK) void forces_v_kernel(PairConst pc, const double *__restrict__ frec,
                                                          const int32_t *__restrict__ nlist, int32_t cap,
                                                          const int32_t *__restrict__ ncount, const int32_t *__restrict__ wave_max,
                                                          const double *__restrict__ dw_tab, const double *__restrict__ sink, int64_t n,
                                                          double *__restrict__ ax, double *__restrict__ ay, double *__restrict__ az,
                                                          double *__restrict__ du, double *__restrict__ dalpha,
                                                          const int32_t *__restrict__ orig, int32_t n_owned) {
    extern __shared__ double lds_dw[];
    for (int k = threadIdx.x; k < TAB_LEN(pc.nq); k += VBLOCK) lds_dw[k] = dw_tab[k];
    __syncthreads();

    const int64_t i = (int64_t)xcd_chunk(blockIdx.x, gridDim.x) * VBLOCK + threadIdx.x;
    if ((i & ~(int64_t)63) >= n) return;
    const int lane = threadIdx.x & 63;
    const int64_t w = i >> 6;
    const bool live = i < n && orig[i] < n_owned;
    const int self = i < n ? (int)i : (int)(n - 1);
    const double4 *fi = reinterpret_cast<const double4 *>(frec + (size_t)self * FREC);
    const double4 A = fi[0], B = fi[1], Cc = fi[2];   // x y z m | vx vy vz rho/2 | c/2 alpha/2 P/(Om rho^2) h
    const int cnt = live ? min(ncount[i], cap) : 0;
    const int kmax = __builtin_amdgcn_readfirstlane(wave_max[w]);
    const int4 *mine4 = reinterpret_cast<const int4 *>(nlist) + ((size_t)w * (cap >> 2)) * 64 + lane;
    const double hi = Cc.w;
    const double inv_h = 1.0 / hi, inv_dq = 0.5 * pc.nq, inv_pi = 1.0 / pc.kernel_pi;
    const double inv_n4i = 1.0 / (pc.kernel_pi * ((hi * hi) * (hi * hi)));          // [V]:140 for h_i

    ForceSums f;
    const int nrow = (kmax + 3) >> 2;
    int4 qa = make_int4(0, 0, 0, 0), qb = qa;
    if (kmax > 0) { qa = load_row(mine4); qb = load_row(mine4 + (size_t)min(1, nrow - 1) * 64); }
    int e1 = 0 < cnt ? qa.x : self;
    const double4 *fj = reinterpret_cast<const double4 *>(frec + (size_t)(e1 & IDX_MASK) * FREC);
    double4 A1 = fj[0], B1 = fj[1], C1 = fj[2];
    for (int r = 0; r < nrow; r++) {                  // whole rows (four entries), rows streamed two ahead
        const int4 qc = load_row(mine4 + (size_t)min(r + 2, nrow - 1) * 64);
#pragma unroll
        for (int v = 0; v < 4; v++) {
            const int k = 4 * r + v;
            const double4 Aj = A1, Bj = B1, Cj = C1;
            const bool act = k < cnt && ((uint32_t)e1 & FLAG_F);
            if (k + 1 < cnt) {                          // idle lanes issue no gather
                e1 = v < 3 ? comp4(qa, v + 1) : qb.x;
                fj = reinterpret_cast<const double4 *>(frec + (size_t)(e1 & IDX_MASK) * FREC);
                A1 = fj[0]; B1 = fj[1]; C1 = fj[2];
            }
            force_visit_v(pc, hi, inv_h, inv_dq, inv_pi, inv_n4i, lds_dw, A, B, Cc, Aj, Bj, Cj, act, f);
        }
        qa = qb; qb = qc;
    }
    if (!live) return;
    force_epilogue_v(pc, sink, i, A, B.w, Cc.x, Cc.y, hi, f, ax, ay, az, du, dalpha);
}


// ---- calc_smoothing -------------------------------------------------------------------------------------
// One partner of a trial length hn: the two sums of density_visit_v (sum m w(q), sum m (q dw(q) - 3 w(q))), normalised once per
// trial in h_trial_finish -- [V]:482-493 divides every pair by pi hn^3, pi hn^4 and hn (with the square root and the three
// divisions of the table lookup: seven IEEE operations of ~30 instructions each per partner; round 3 measured calc_smoothing
// at 0.54 ms per step with them).  Beyond 2 hn both knots are the tables' final zeros: no test of q.
__device__ __forceinline__ void h_trial_visit(const double4 &pi, const double4 &pj, double inv_hn, double inv_dq,
                                              const double *__restrict__ w_tab, const double *__restrict__ dw_tab, DensSumsV &d) {
#pragma clang fp contract(off)
    const double n0 = pi.x - pj.x, n1 = pi.y - pj.y, n2 = pi.z - pj.z;
    double dr, rs;
    rsqrt_sqrt(fma(n2, n2, fma(n1, n1, n0 * n0)), dr, rs);
    const double qi = dr * inv_hn;
    const double t = knot_coord(qi, inv_dq);
    const int k = (int)t;
    const double a = __builtin_amdgcn_fract(t), b = 1.0 - a;
    const double wl = fma(a, w_tab[k + 1], b * w_tab[k]), dwl = fma(a, dw_tab[k + 1], b * dw_tab[k]);
    d.s1 = fma(pj.w, wl, d.s1);
    d.s2 = fma(pj.w, fma(qi, dwl, -3.0 * wl), d.s2);
}
__device__ __forceinline__ void h_trial_finish(const PairConst &pc, double hn, const DensSumsV &d, double &rho, double &om) {
    const double h3 = hn * hn * hn;
    rho = d.s1 / (pc.kernel_pi * h3);                                                       // [V]:139
    const double om_acc = -d.s2 / (pc.kernel_pi * ((hn * hn) * (hn * hn)));                 // [V]:140,487
    om = 1.0 + (hn / (3.0 * rho)) * om_acc;                                                 // [V]:535
}

// rho and Omega of ONE body with trial length hn on the tree of the last evaluation (leaf boxes and
// reaches hold the OLD h of every particle), [V]:531-535 -> density_tree_search
__device__ void density_one(const GridDesc &g, const double4 *__restrict__ drec, const double4 *__restrict__ lrec,
                            const int32_t *__restrict__ cell_start, const double *__restrict__ w_tab,
                            const double *__restrict__ dw_tab, const PairConst &pc, const double4 &pi, double hn, double &rho,
                            double &om) {
    int cc[3];
    cell_coords(g, pi.x, pi.y, pi.z, cc);
    const int R = max((int)ceil(2.0 * hn * g.inv_edge), 1);
    const int d0 = g.dim[g.s[0]], d1 = g.dim[g.s[1]], d2 = g.dim[g.s[2]];
    DensSumsV d;
    const double inv_hn = 1.0 / hn, inv_dq = 0.5 * pc.nq, rt2 = 4.0 * hn * hn * (1.0 + 1e-12);
    for (int c2 = max(cc[2] - R, 0); c2 <= min(cc[2] + R, d2 - 1); c2++)
        for (int c1 = max(cc[1] - R, 0); c1 <= min(cc[1] + R, d1 - 1); c1++) {
            const int64_t row = ((int64_t)c2 * d1 + c1) * d0;
            const int jb = cell_start[row + max(cc[0] - R, 0)], je = cell_start[row + min(cc[0] + R, d0 - 1) + 1];
            for (int j = jb; j < je; j++) {
                const double4 pj = drec[j];
                const double n0 = pi.x - pj.x, n1 = pi.y - pj.y, n2 = pi.z - pj.z;
                if (n0 * n0 + n1 * n1 + n2 * n2 > rt2) continue;        // cheap pre-test (beyond 2 hn the visit adds zeros anyway)
                const double4 lj = lrec[j];
                if (!reaches(lj, pi.x, pi.y, pi.z)) continue;
                h_trial_visit(pi, pj, inv_hn, inv_dq, w_tab, dw_tab, d);
            }
        }
    h_trial_finish(pc, hn, d, rho, om);
}

// the same sums from the body's neighbour list: valid while the trial length stays inside the margin shell the list
// was built with (hn <= H_MARGIN * h of the build); entries are tested as density_one tests its candidates
__device__ void density_list(const double4 *__restrict__ drec, const int4 *__restrict__ mine, int cap4, int cnt, int tcnt,
                             const double *__restrict__ w_tab, const double *__restrict__ dw_tab, const PairConst &pc,
                             const double4 &pi, double hn, double &rho, double &om) {
    DensSumsV d;
    const double inv_hn = 1.0 / hn, inv_dq = 0.5 * pc.nq;
    d.s1 = pi.w * w_tab[0];                            // the body itself (r = 0; its own leaf is always reached)
    d.s2 = pi.w * (-3.0 * w_tab[0]);
    auto visit = [&](uint32_t ent) {
        if (!(ent & FLAG_R)) return;
        h_trial_visit(pi, drec[ent & IDX_MASK], inv_hn, inv_dq, w_tab, dw_tab, d);
    };
    // one 16-byte row = four entries; the D/F entries from the top of the column, the margin shell from its bottom
    for (int row = 0; 4 * row < cnt; row++) {
        const int4 q = mine[(size_t)row * 64];
        const int left = cnt - 4 * row;
        visit((uint32_t)q.x);
        if (left > 1) visit((uint32_t)q.y);
        if (left > 2) visit((uint32_t)q.z);
        if (left > 3) visit((uint32_t)q.w);
    }
    for (int row = 0; 4 * row < tcnt; row++) {
        const int4 q = mine[(size_t)(cap4 - 1 - row) * 64];
        const int left = tcnt - 4 * row;
        visit((uint32_t)q.x);
        if (left > 1) visit((uint32_t)q.y);
        if (left > 2) visit((uint32_t)q.z);
        if (left > 3) visit((uint32_t)q.w);
    }
    h_trial_finish(pc, hn, d, rho, om);
}

__global__ __launch_bounds__(VBLOCK) void update_h_kernel(GridDesc g, PairConst pc, const double4 *__restrict__ drec,
                                                          const double4 *__restrict__ lrec, const int32_t *__restrict__ cell_start,
                                                          const double *__restrict__ w_tab, const double *__restrict__ dw_tab,
                                                          int64_t n, const double *__restrict__ h_old, double *__restrict__ h_out,
                                                          double *__restrict__ rho, double *__restrict__ omega,
                                                          const int32_t *__restrict__ orig, int32_t n_owned,
                                                          const int32_t *__restrict__ nlist, int32_t cap,
                                                          const int32_t *__restrict__ ncount, const int32_t *__restrict__ ntail,
                                                          int has_margin) {
    // Two phases.  Every lane takes the first Newton step of its own particle; most particles are done with it (h grew by
    // less than the tolerance, or shrank).  The ones that must re-evaluate rho -- a walk over their whole list -- are then
    // dealt densely to the lanes of the workgroup, so that the walks run in full wavefronts instead of in every wavefront
    // for a third of its lanes (0.49 -> 0.40 ms per step on the bench disc).  A particle's result does not depend on the lane
    // that computes it.
    __shared__ int s_list[VBLOCK];
    __shared__ int s_n;
    const int64_t base = (int64_t)xcd_chunk(blockIdx.x, gridDim.x) * VBLOCK;
    const int64_t i = base + threadIdx.x;
    if (threadIdx.x == 0) s_n = 0;
    __syncthreads();
    bool iterate = false;
    if (i < n) {
        const double h0 = h_old[i];
        if (orig[i] >= n_owned) {
            h_out[i] = h0;
        } else {
            const double mi = drec[i].w;
            const double t = pc.eta / h0;
            const double hn = h0 * (1.0 + ((mi * (t * t * t) / rho[i]) - 1.0) / (3.0 * omega[i]));      // [V]:527
            if (hn < pc.h_max_length && hn > pc.h_min_length) {                                       // [V]:528
                iterate = ((hn - h0) / h0) > pc.h_tol && hn < pc.h_iter_cap;                          // [V]:529, first test
                if (!iterate) h_out[i] = hn;
            } else {
                h_out[i] = h0;                                                                        // [V]:541
            }
        }
    }
    const unsigned long long mask = __ballot(iterate);
    int wbase = 0;
    if ((threadIdx.x & 63) == 0 && mask) wbase = atomicAdd(&s_n, __popcll(mask));
    wbase = __shfl(wbase, 0, 64);
    if (iterate) s_list[wbase + __popcll(mask & ((1ull << (threadIdx.x & 63)) - 1ull))] = threadIdx.x;
    __syncthreads();
    if ((int)threadIdx.x >= s_n) return;
    const int64_t ii = base + s_list[threadIdx.x];
    const double h0 = h_old[ii];
    const double4 pi = drec[ii];
    const int cap4 = cap >> 2;
    const int4 *mine = reinterpret_cast<const int4 *>(nlist) + ((size_t)(ii >> 6) * cap4) * 64 + (ii & 63);
    const int cnt = min(ncount[ii], cap), tcnt = ntail[ii];
    double r = rho[ii], om = omega[ii];
    double old_len = h0;
    double t = pc.eta / h0;
    double hn = h0 * (1.0 + ((pi.w * (t * t * t) / r) - 1.0) / (3.0 * om));              // [V]:527 (as in phase 1)
    while (((hn - old_len) / old_len) > pc.h_tol && hn < pc.h_iter_cap) {                // [V]:529
        old_len = hn;
        if (has_margin && hn <= H_MARGIN * h0) density_list(drec, mine, cap4, cnt, tcnt, w_tab, dw_tab, pc, pi, hn, r, om);
        else density_one(g, drec, lrec, cell_start, w_tab, dw_tab, pc, pi, hn, r, om);    // [V]:531-535
        t = pc.eta / hn;
        hn = hn * (1.0 + ((pi.w * (t * t * t)) / r - 1.0) / (3.0 * om));                 // [V]:538
    }
    rho[ii] = r; omega[ii] = om;
    h_out[ii] = hn;
}

}  // namespace

#define VH_CHECK(expr)                                                      \
    do {                                                                    \
        hipError_t _e = (expr);                                             \
        if (_e != hipSuccess) {                                             \
            c->err = std::string(#expr) + ": " + hipGetErrorString(_e);     \
            return SPH_ERR_HIP;                                             \
        }                                                                   \
    } while (0)

hipError_t varh_sort_tmp_bytes(int64_t n, size_t *bytes) {
    size_t b = 0;
    hipError_t e = rocprim::radix_sort_pairs(nullptr, b, (uint64_t *)nullptr, (uint64_t *)nullptr, (uint32_t *)nullptr,
                                             (uint32_t *)nullptr, (size_t)n, 0u, 63u, (hipStream_t) nullptr);
    *bytes = b;
    return e;
}

int varh_h_stats(sph_ctx *c, bool with_growth) {
    const int64_t n = std::max(c->n, c->n_slots);     // a pending ghost swap: every occupied slot (an upper bound on h)
    c->h_growth = INFINITY;
    if (n == 0) { c->h_max_glob = c->h_mean = c->p.h; return SPH_OK; }
    const int nb = (int)std::min<int64_t>((n + VBLOCK - 1) / VBLOCK, 512);
    double *part = c->bbox_part;     // reuse the bbox partial buffer (>= 1024*6 doubles)
    // with_growth: c->h_new holds the lengths the list in place was built with (launch_update_h swapped the two arrays)
    h_stats_partial<<<dim3(nb), dim3(VBLOCK), 0, c->stream>>>(c->f[SPH_F_H], with_growth ? c->h_new : nullptr, n, part);
    h_stats_final<<<dim3(1), dim3(64), 0, c->stream>>>(part, nb, part + 2048);
    VH_CHECK(hipGetLastError());
    VH_CHECK(hipMemcpyAsync(c->h_pinned + 28, part + 2048, 4 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    VH_CHECK(hipStreamSynchronize(c->stream));
    c->h_max_glob = c->h_pinned[28];
    c->h_mean = c->h_pinned[29] / (double)n;
    c->h_shrink = INFINITY;
    if (with_growth) { c->h_growth = c->h_pinned[30]; c->h_shrink = c->h_pinned[31]; }
    if (!(c->h_max_glob > 0.0) || !std::isfinite(c->h_max_glob)) { c->err = "variable h: non-positive or non-finite smoothing length"; return SPH_ERR_ARG; }
    return SPH_OK;
}

int varh_leaf_build(sph_ctx *c) {
    const int64_t n = c->n;
    if (n == 0) return SPH_OK;
    // root box of the reference's create_tree, [V]:1007-1012, from the grid's bounding box (same min/max)
    RootBox rb;
    double size = 0.0;
    for (int a = 0; a < 3; a++) {
        rb.c[a] = (c->bbox[3 + a] + c->bbox[a]) / 2.0;
        size = std::max(size, c->bbox[3 + a] - c->bbox[a]);
    }
    rb.size = size;
    for (int a = 0; a < 3; a++) c->root_box[a] = rb.c[a];
    c->root_box[3] = size;
    const unsigned gb = (unsigned)((n + VBLOCK - 1) / VBLOCK);
    const double4 *prec = reinterpret_cast<const double4 *>(c->prec);
    if (c->gx_src) {
        // the shared octree: root box and sorted keys of every GPU's particles (gravity.hip keeps them)
        size = 0.0;
        for (int a = 0; a < 3; a++) {
            rb.c[a] = (c->gx_box[3 + a] + c->gx_box[a]) / 2.0;
            size = std::max(size, c->gx_box[3 + a] - c->gx_box[a]);
        }
        rb.size = size;
        for (int a = 0; a < 3; a++) c->root_box[a] = rb.c[a];
        c->root_box[3] = size;
        { const int st = global_keys_sorted(c); if (st != SPH_OK) return st; }
        leaf_boxes_ext<<<dim3(gb), dim3(VBLOCK), 0, c->stream>>>(rb, c->g_keys_alt, c->gx_n, n, prec, reinterpret_cast<double4 *>(c->lrec), c->leaf_half);
        VH_CHECK(hipGetLastError());
    } else {
        leaf_keys<<<dim3(gb), dim3(VBLOCK), 0, c->stream>>>(rb, prec, n, c->mkeys, c->mvals);
        VH_CHECK(hipGetLastError());
        size_t tmp = c->msort_tmp_bytes;
        VH_CHECK(rocprim::radix_sort_pairs(c->msort_tmp, tmp, c->mkeys, c->mkeys_alt, c->mvals, c->mvals_alt, (size_t)n, 0u, 63u, c->stream));
        leaf_boxes<<<dim3(gb), dim3(VBLOCK), 0, c->stream>>>(rb, c->mkeys_alt, c->mvals_alt, n, prec, reinterpret_cast<double4 *>(c->lrec), c->leaf_half);
        VH_CHECK(hipGetLastError());
        c->path_keys_valid = true;           // the gravity tree and the accretion pass of this grid build take them from here
    }
    cell_hmax_kernel<<<dim3((unsigned)((c->grid.ncells + VBLOCK - 1) / VBLOCK)), dim3(VBLOCK), 0, c->stream>>>(
        c->cell_start, c->grid.ncells, prec, c->cell_hmax);
    VH_CHECK(hipGetLastError());
    return SPH_OK;
}

// h changed, positions did not: refresh the records that carry h instead of re-sorting and re-deriving the leaf cells
int varh_refresh_h(sph_ctx *c) {
    const int64_t n = c->n;
    if (n == 0) return SPH_OK;
    refresh_h_records<<<dim3((unsigned)((n + VBLOCK - 1) / VBLOCK)), dim3(VBLOCK), 0, c->stream>>>(
        n, c->f[SPH_F_H], c->leaf_half, reinterpret_cast<double4 *>(c->prec), reinterpret_cast<double4 *>(c->lrec));
    cell_hmax_kernel<<<dim3((unsigned)((c->grid.ncells + VBLOCK - 1) / VBLOCK)), dim3(VBLOCK), 0, c->stream>>>(
        c->cell_start, c->grid.ncells, reinterpret_cast<const double4 *>(c->prec), c->cell_hmax);
    VH_CHECK(hipGetLastError());
    return SPH_OK;
}

int varh_nlist_build(sph_ctx *c) {
    static const bool no_reflag_env = getenv("SPH_NO_REFLAG") != nullptr;      // A/B switches: always build, no re-flag margin
    const bool no_reflag = no_reflag_env || (c->p.flags & SPH_FLAG_NO_REFLAG) != 0;
    const int64_t n = c->n;
    if (n == 0) return SPH_OK;
    const int R = std::max(1, (int)std::ceil(2.0 * H_MARGIN * c->h_max_glob * c->grid.inv_edge * (1.0 + 1e-9)));
    const unsigned gb = (unsigned)((n + VBLOCK - 1) / VBLOCK);
    const double grow = no_reflag ? 1.0 : REFLAG_GROW;
    for (int attempt = 0; attempt < 8; attempt++) {
        // per-wave row needs go to wave_class (free in variable-h mode: no split force evaluation), their maximum straight to the host
        nlist_v_tiled<<<dim3(gb), dim3(VBLOCK), 0, c->stream>>>(
            c->grid, R, c->h_max_glob, grow * grow, reinterpret_cast<const double4 *>(c->prec), reinterpret_cast<const double4 *>(c->lrec), c->orig,
            c->cell_start, c->cell_hmax, n, (int32_t)c->n_owned, c->nl_cap, c->nlist, c->ncount, c->ntail, c->wave_max, c->wave_class,
            c->numbers_set ? c->number : nullptr);
        max_to_host<<<dim3(1), dim3(1024), 0, c->stream>>>(c->wave_class, (n + 63) / 64, reinterpret_cast<int32_t *>(c->h_pinned + 9));
        VH_CHECK(hipGetLastError());
        VH_CHECK(hipStreamSynchronize(c->stream));
        const int32_t mx = *reinterpret_cast<int32_t *>(c->h_pinned + 9);
        c->nl_max = mx;
        if (mx <= c->nl_cap) {
            c->nlist_builds++;
            c->list_has_margin = true;
            c->vl_grow = grow;
            return SPH_OK;
        }
        ctx_free(c, c->nlist);
        c->nl_cap = ((mx + mx / 8 + 8) + 3) & ~3;
        if (ctx_alloc(c, &c->nlist, (size_t)c->nl_waves_cap * c->nl_cap * 64, "neighbour list") != SPH_OK) { c->nl_cap = 0; return SPH_ERR_NOMEM; }
    }
    c->err = "neighbour list did not converge";
    return SPH_ERR_STATE;
}

// can the list in place (built for the same positions and order with the lengths now in c->h_new) be re-flagged for
// the new lengths?  needs its margin shell intact and no h grown beyond that margin (varh_h_stats(c, true) measured it)
bool varh_can_reflag(const sph_ctx *c) {
    return c->list_has_margin && c->vl_grow > 1.0 && c->h_growth <= c->vl_grow * (1.0 - 1e-12) && c->nl_cap > 0;
}

// the list of the new lengths out of the list of the old ones, in place (nlist_v_reflag); no read-back: a column cannot
// outgrow its rows
int varh_nlist_reflag(sph_ctx *c) {
    const int64_t n = c->n;
    if (n == 0) return SPH_OK;
    nlist_v_reflag<<<dim3((unsigned)((n + VBLOCK - 1) / VBLOCK)), dim3(VBLOCK), 0, c->stream>>>(
        reinterpret_cast<const double4 *>(c->prec), reinterpret_cast<const double4 *>(c->lrec), c->orig, n, (int32_t)c->n_owned, c->nl_cap,
        c->nlist, c->ncount, c->ntail, c->wave_max, c->numbers_set ? c->number : nullptr);
    VH_CHECK(hipGetLastError());
    c->list_has_margin = false;        // the margin rows may be overwritten: calc_smoothing falls back to its cell walk on this list
    c->nlist_reflags++;
    return SPH_OK;
}

// the re-flag pass and the start-of-step density pass in one kernel (reflag_density_kernel); as varh_nlist_reflag otherwise
int varh_reflag_density(sph_ctx *c, const PairConst &pc) {
    const int64_t n = c->n;
    if (n == 0) return SPH_OK;
    const size_t lds = (size_t)TAB_LEN(pc.nq) * 2 * sizeof(double);
    reflag_density_kernel<<<dim3((unsigned)((n + VBLOCK - 1) / VBLOCK)), dim3(VBLOCK), lds, c->stream>>>(
        pc, reinterpret_cast<const double4 *>(c->prec), reinterpret_cast<const double4 *>(c->lrec), reinterpret_cast<const double4 *>(c->drec),
        c->f[SPH_F_M], c->orig, n, (int32_t)c->n_owned, c->nl_cap, c->nlist, c->ncount, c->ntail, c->wave_max,
        c->numbers_set ? c->number : nullptr, c->w_tab, c->dw_tab, c->f[SPH_F_U], c->f[SPH_F_ALPHA], c->f[SPH_F_VX], c->f[SPH_F_VY],
        c->f[SPH_F_VZ], c->f[SPH_F_RHO], c->f[SPH_F_OMEGA], c->f[SPH_F_P], c->f[SPH_F_C], c->frec);
    VH_CHECK(hipGetLastError());
    c->list_has_margin = false;
    c->nlist_reflags++;
    return SPH_OK;
}

hipError_t launch_density_v(sph_ctx *c, const PairConst &pc) {
    if (c->n == 0) return hipSuccess;
    const size_t lds = (size_t)TAB_LEN(pc.nq) * 2 * sizeof(double);
    density_v_kernel<<<dim3((unsigned)((c->n + VBLOCK - 1) / VBLOCK)), dim3(VBLOCK), lds, c->stream>>>(
        pc, reinterpret_cast<const double4 *>(c->drec), c->nlist, c->nl_cap, c->ncount, c->wave_max, c->w_tab, c->dw_tab, c->n,
        c->f[SPH_F_H], c->f[SPH_F_U], c->f[SPH_F_ALPHA], c->f[SPH_F_VX], c->f[SPH_F_VY], c->f[SPH_F_VZ], c->f[SPH_F_RHO],
        c->f[SPH_F_OMEGA], c->f[SPH_F_P], c->f[SPH_F_C], c->frec, c->orig, (int32_t)c->n_owned);
    return hipGetLastError();
}

hipError_t launch_eos_only_v(sph_ctx *c, const PairConst &pc) {
    if (c->n == 0) return hipSuccess;
    eos_only_v_kernel<<<dim3((unsigned)((c->n + 255) / 256)), dim3(256), 0, c->stream>>>(
        pc, c->n, reinterpret_cast<const double4 *>(c->drec), c->f[SPH_F_H], c->f[SPH_F_U], c->f[SPH_F_ALPHA], c->f[SPH_F_VX],
        c->f[SPH_F_VY], c->f[SPH_F_VZ], c->f[SPH_F_RHO], c->f[SPH_F_OMEGA], c->f[SPH_F_P], c->f[SPH_F_C], c->frec);
    return hipGetLastError();
}

hipError_t launch_forces_v(sph_ctx *c, const PairConst &pc) {
    if (c->n == 0) return hipSuccess;
    const size_t lds = (size_t)TAB_LEN(pc.nq) * sizeof(double);
    forces_v_kernel<<<dim3((unsigned)((c->n + VBLOCK - 1) / VBLOCK)), dim3(VBLOCK), lds, c->stream>>>(
        pc, c->frec, c->nlist, c->nl_cap, c->ncount, c->wave_max, c->dw_tab, c->sink, c->n, c->f[SPH_F_AX], c->f[SPH_F_AY],
        c->f[SPH_F_AZ], c->f[SPH_F_DU], c->f[SPH_F_DALPHA], c->orig, (int32_t)c->n_owned);
    return hipGetLastError();
}

hipError_t launch_update_h(sph_ctx *c, const PairConst &pc) {
    if (c->n == 0) return hipSuccess;
    update_h_kernel<<<dim3((unsigned)((c->n + VBLOCK - 1) / VBLOCK)), dim3(VBLOCK), 0, c->stream>>>(
        c->grid, pc, reinterpret_cast<const double4 *>(c->drec), reinterpret_cast<const double4 *>(c->lrec), c->cell_start,
        c->w_tab, c->dw_tab, c->n, c->f[SPH_F_H], c->h_new, c->f[SPH_F_RHO], c->f[SPH_F_OMEGA], c->orig, (int32_t)c->n_owned,
        c->nlist, c->nl_cap, c->ncount, c->ntail, c->list_has_margin ? 1 : 0);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) std::swap(c->f[SPH_F_H], c->h_new);
    return e;
}

}  // namespace sph
