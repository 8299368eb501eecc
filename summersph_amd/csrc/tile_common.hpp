// tile_common.hpp -- building blocks of the whole-tile kernels (tiled.hip): the three candidate intervals of a group of
// consecutive targets, their plan in memory, the staging loop.
#pragma once
#include "pair_common.hpp"

namespace sph {

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
    // -1: j lies in none of the intervals (variable h: partners beyond the 27-cell stencil of the group)
    __device__ __forceinline__ int slot_checked(int j) const {
        const unsigned u0 = (unsigned)(j - lo[0]), u1 = (unsigned)(j - lo[1]), u2 = (unsigned)(j - lo[2]);
        int s = u2 < (unsigned)len[2] ? base[2] + (int)u2 : -1;
        s = u1 < (unsigned)len[1] ? base[1] + (int)u1 : s;
        s = u0 < (unsigned)len[0] ? (int)u0 : s;
        return s;
    }
};


__device__ __forceinline__ void load_plan(const int32_t *__restrict__ plan, int64_t group, TileMap &tm) {
    const int32_t *p = plan + 8 * (size_t)group;
    const int4 a = *reinterpret_cast<const int4 *>(p), b = *reinterpret_cast<const int4 *>(p + 4);      // two 16-byte loads
    tm.lo[0] = a.x; tm.lo[1] = a.y; tm.lo[2] = a.z; tm.len[0] = a.w; tm.len[1] = b.x; tm.len[2] = b.y;
    tm.base[0] = 0; tm.base[1] = a.w; tm.base[2] = a.w + b.x;
    tm.need = b.z;
}


// Staging: the three intervals form one index space, tile slot s <- record lo[q] + (s - base[q]); a record is UPR units
// of 16 bytes; unit t of the tile goes to dst[SWZ ? q_unit(s) + part : t].  U loads in flight per thread: staging is
// latency-bound (one workgroup per CU, nothing else to run), so what counts is the number of round trips.
__device__ __forceinline__ int q_unit(int s) { return 6 * s + (s >> 3); }
template <int BS, int U, int UPR, bool SWZ>
__device__ __forceinline__ void stage_tile(const double2 *__restrict__ src, double2 *dst, const TileMap &tm) {
    const int b1 = tm.base[1], b2 = tm.base[2], o0 = tm.lo[0], o1 = tm.lo[1] - b1, o2 = tm.lo[2] - b2;
    const int count = UPR * tm.need;
    for (int t0 = threadIdx.x; t0 < count; t0 += U * BS) {
        double2 v[U];
#pragma unroll
        for (int u = 0; u < U; u++) {                              // unconditional (clamped) loads: plain registers
            const int t = min(t0 + u * BS, count - 1), sl = t / UPR;
            v[u] = src[(size_t)UPR * (size_t)(sl + (sl >= b2 ? o2 : (sl >= b1 ? o1 : o0))) + (t - UPR * sl)];
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int t = t0 + u * BS, sl = t / UPR;
            if (t < count) dst[SWZ ? UPR * sl + (sl >> 3) + (t - UPR * sl) : t] = v[u];      // UPR = 6: q_unit(sl) + part
        }
    }
}

}  // namespace sph
