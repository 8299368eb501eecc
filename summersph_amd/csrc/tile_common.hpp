// tile_common.hpp -- building blocks of the whole-tile kernels (tiled.hip): the three candidate intervals of a group of
// consecutive targets, their plan in memory, the staging loop.
#pragma once
#include "pair_common.hpp"

namespace sph {

// ---- the fixed-h neighbour list of the tiled build: 16-bit entries --------------------------------------------------------
// An entry is the neighbour's SLOT IN THE TILE OF ITS GROUP of 256 consecutive targets (the group of forces_q; nlist_tiled is
// one workgroup per group): slot = base[q] + (j - lo[q]) for the interval q (offset o2 = q - 1 along the slowest grid axis)
// that holds j, base = {0, len0, len0 + len1}.  forces_q uses an entry as it is; the 1024-target groups of density_wt and the
// direct-gather kernels add a per-interval constant (found with two compares against base[1], base[2]).  Half the bytes of a
// 32-bit index per entry -- the list rows are the largest stream of density_wt, and the build writes them all.
// Layout ("ELL, wave-strided, 8-packed"): entry k of particle i = (wave w, lane l) is halfword ent_pos(k & 7) of the int4 at
// nlist4[(w * cap/8 + k/8) * 64 + l].  ent_pos puts entries 4t .. 4t+3 of a row into the SAME half of its four words, so the
// four lanes of a forces_q target (lane s = word s) consume a row in two trips of four consecutive entries each.
// The three intervals of a group must stay below 65536 records together (LIST16_MAX_NEED); a context whose groups outgrow
// that (a very dense thick domain) falls back to the untiled 32-bit list of pairs.hip for good (nlist_build_tiled).
constexpr int LIST16_MAX_NEED = 65536;
constexpr double SENTINEL_POS = 1.0e30;       // coordinates of the sentinel record of the whole-tile kernels (tiled.hip): q is astronomically > 2
__device__ __host__ __forceinline__ int ent_pos(int k) { return ((k & 3) << 1) | ((k >> 2) & 1); }
// halfword p (compile-time in the unrolled loops) of a row
__device__ __forceinline__ int row_entry(const int4 &q, int p) {
    const unsigned wd = (unsigned)((p >> 1) == 0 ? q.x : ((p >> 1) == 1 ? q.y : ((p >> 1) == 2 ? q.z : q.w)));
    return (int)((p & 1) ? (wd >> 16) : (wd & 0xffffu));
}
// entry -> sorted index (direct gathers) or -> slot of another tile: e + add[interval of e]
struct EntryMap {
    int b1, b2;           // base[1], base[2] of the forces group the list column belongs to
    int a0, a1, a2;       // what to add to an entry of interval 0, 1, 2 (scalars: an array member went to scratch memory)
    // written as two conditional increments: the nested select `e >= b2 ? a2 : (e >= b1 ? a1 : a0)` is turned into a
    // three-entry table in scratch memory by the compiler (one scratch load per visit)
    __device__ __forceinline__ int operator()(int e) const {
        return (int)((unsigned)e + (unsigned)a0 + (e >= b1 ? (unsigned)a1 - (unsigned)a0 : 0u) + (e >= b2 ? (unsigned)a2 - (unsigned)a1 : 0u));
    }
};
// plan_f record of group g: {lo0, lo1, lo2, len0, len1, len2, need, 0}; sorted index = e + lo[q] - base[q]
__device__ __forceinline__ EntryMap entry_to_index(const int32_t *__restrict__ plan_f, int64_t g) {
    const int32_t *p = plan_f + 8 * (size_t)g;
    const int4 a = *reinterpret_cast<const int4 *>(p);
    const int len1 = p[4];
    EntryMap m;
    m.b1 = a.w; m.b2 = a.w + len1;
    m.a0 = a.x; m.a1 = (int)((unsigned)a.y - (unsigned)m.b1); m.a2 = (int)((unsigned)a.z - (unsigned)m.b2);
    return m;
}

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
// the same as a byte offset, for a slot below 2^16: 16 (6 s + s/8) = 16 floor(49 s / 8) = (98 s) & ~15 -- a 24-bit multiply and a mask
__device__ __forceinline__ const double2 *q_record(const double2 *tile, int s) {
    return reinterpret_cast<const double2 *>(reinterpret_cast<const char *>(tile) + (__umul24((unsigned)s, 98u) & ~15u));
}
// The U loaded values, pinned between the loads and the stores.  Without it the compiler sinks each clamped load into the
// conditional block of its store and the tile arrives in U consecutive round trips per trip of the loop (seen in the ISA of
// round 3: global_load, s_waitcnt vmcnt(0), ds_write, U times); with it the U loads are issued back to back and waited for once.
template <int U>
__device__ __forceinline__ void pin_loaded(double2 (&v)[U]) {
    static_assert(U == 4 || U == 6 || U == 8 || U == 10, "stage_tile: 4, 6, 8 or 10 loads in flight");
    if constexpr (U == 4)
        asm volatile("" : "+v"(v[0].x), "+v"(v[0].y), "+v"(v[1].x), "+v"(v[1].y), "+v"(v[2].x), "+v"(v[2].y), "+v"(v[3].x), "+v"(v[3].y));
    else if constexpr (U == 6)
        asm volatile("" : "+v"(v[0].x), "+v"(v[0].y), "+v"(v[1].x), "+v"(v[1].y), "+v"(v[2].x), "+v"(v[2].y), "+v"(v[3].x), "+v"(v[3].y),
                          "+v"(v[4].x), "+v"(v[4].y), "+v"(v[5].x), "+v"(v[5].y));
    else if constexpr (U == 8)
        asm volatile("" : "+v"(v[0].x), "+v"(v[0].y), "+v"(v[1].x), "+v"(v[1].y), "+v"(v[2].x), "+v"(v[2].y), "+v"(v[3].x), "+v"(v[3].y),
                          "+v"(v[4].x), "+v"(v[4].y), "+v"(v[5].x), "+v"(v[5].y), "+v"(v[6].x), "+v"(v[6].y), "+v"(v[7].x), "+v"(v[7].y));
    else
        asm volatile("" : "+v"(v[0].x), "+v"(v[0].y), "+v"(v[1].x), "+v"(v[1].y), "+v"(v[2].x), "+v"(v[2].y), "+v"(v[3].x), "+v"(v[3].y),
                          "+v"(v[4].x), "+v"(v[4].y), "+v"(v[5].x), "+v"(v[5].y), "+v"(v[6].x), "+v"(v[6].y), "+v"(v[7].x), "+v"(v[7].y),
                          "+v"(v[8].x), "+v"(v[8].y), "+v"(v[9].x), "+v"(v[9].y));
}

// Staging in three parts, so that a kernel can have EVERY global load of a group in flight together -- the tile and what its
// threads need besides (their target's record, their first list rows): stage_issue sends the first U loads of the thread,
// stage_wait waits for them (pin_loaded: put it BEFORE the barrier that frees the tile -- the
// compiler moves loads whose values are first used behind a barrier to behind that barrier), stage_commit writes them to the tile, stage_rest does the trips beyond the first (none when
// UPR * need <= U * BS, which holds for the kernels with the table in LDS).
template <int U>
struct StageRegs { double2 v[U]; };

// (BS: the threads that stage, me: this thread's number among them -- all of the workgroup by default)
template <int BS, int U, int UPR>
__device__ __forceinline__ void stage_issue(StageRegs<U> &st, const double2 *__restrict__ src, const TileMap &tm, int me = threadIdx.x) {
    const int b1 = tm.base[1], b2 = tm.base[2], o0 = tm.lo[0], o1 = tm.lo[1] - b1, o2 = tm.lo[2] - b2;
    const int count = UPR * tm.need;
    int tid = me;                                                  // opaque, as in stage_commit
    asm volatile("" : "+v"(tid));
#pragma unroll
    for (int u = 0; u < U; u++) {                                  // unconditional (clamped) loads: plain registers
        const int t = max(min(tid + u * BS, count - 1), 0), sl = t / UPR;
        st.v[u] = src[(size_t)UPR * (size_t)(sl + (sl >= b2 ? o2 : (sl >= b1 ? o1 : o0))) + (t - UPR * sl)];
    }
}
template <int U>
__device__ __forceinline__ void stage_wait(StageRegs<U> &st) { pin_loaded<U>(st.v); }
template <int BS, int U, int UPR, bool SWZ>
__device__ __forceinline__ void stage_commit(StageRegs<U> &st, double2 *dst, const TileMap &tm, int me = threadIdx.x) {
    const int count = UPR * tm.need;
    // the thread's U tile addresses depend on threadIdx only: hoisted out of the kernel's group loop they cost U registers for the
    // whole kernel (spilled, and reloaded one by one here, in round 3) -- the opaque copy keeps their computation in place
    int tid = me;
    asm volatile("" : "+v"(tid));
#pragma unroll
    for (int u = 0; u < U; u++) {
        const int t = tid + u * BS, sl = t / UPR;
        if (t < count) dst[SWZ ? UPR * sl + (sl >> 3) + (t - UPR * sl) : t] = st.v[u];          // UPR = 6: q_unit(sl) + part
    }
}
template <int BS, int U, int UPR, bool SWZ>
__device__ __forceinline__ void stage_rest(const double2 *__restrict__ src, double2 *dst, const TileMap &tm, int me = threadIdx.x) {
    const int b1 = tm.base[1], b2 = tm.base[2], o0 = tm.lo[0], o1 = tm.lo[1] - b1, o2 = tm.lo[2] - b2;
    const int count = UPR * tm.need;
    for (int t0 = me + U * BS; t0 < count; t0 += U * BS) {
        double2 v[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int t = min(t0 + u * BS, count - 1), sl = t / UPR;
            v[u] = src[(size_t)UPR * (size_t)(sl + (sl >= b2 ? o2 : (sl >= b1 ? o1 : o0))) + (t - UPR * sl)];
        }
        pin_loaded<U>(v);
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int t = t0 + u * BS, sl = t / UPR;
            if (t < count) dst[SWZ ? UPR * sl + (sl >> 3) + (t - UPR * sl) : t] = v[u];
        }
    }
}
template <int BS, int U, int UPR, bool SWZ>
__device__ __forceinline__ void stage_tile(const double2 *__restrict__ src, double2 *dst, const TileMap &tm) {
    if (tm.need <= 0) return;         // (a group of ghosts only: nothing to stage, and the clamped loads of stage_issue would start at -1)
    StageRegs<U> st;
    stage_issue<BS, U, UPR>(st, src, tm);
    stage_wait<U>(st);
    stage_commit<BS, U, UPR, SWZ>(st, dst, tm);
    stage_rest<BS, U, UPR, SWZ>(src, dst, tm);
}

// values loaded from global memory, pinned like the staged ones: the loads are issued before this point and waited for here
__device__ __forceinline__ void pin_value(double4 &a) { asm volatile("" : "+v"(a.x), "+v"(a.y), "+v"(a.z), "+v"(a.w)); }
__device__ __forceinline__ void pin_value(int4 &a) { asm volatile("" : "+v"(a.x), "+v"(a.y), "+v"(a.z), "+v"(a.w)); }
__device__ __forceinline__ void pin_value(uint32_t &a) { asm volatile("" : "+v"(a)); }
__device__ __forceinline__ void pin_value(int &a) { asm volatile("" : "+v"(a)); }

}  // namespace sph
