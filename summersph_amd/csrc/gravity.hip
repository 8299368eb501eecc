// gravity.hip -- Barnes-Hut gas self-gravity, reproducing the reference's tree walk.
//
// Replaces (citations: /root/reference/SUMMER_SPH.f90 "[F]"; "SUMMER_SPH - Variable.f90" "[V]")
//   create_tree / build_tree (masses, centres of mass)      [F]:795-816,149-246
//   particle_gravforces / particle_gravforce_one, theta 0.5 [F]:249-290  ([V]:285-311: softening table
//                                                           looked up with the particle's own h)
//
// The reference walks a pointer octree (bbox-midpoint root, edge = largest extent, strict '>' split, one
// particle per leaf) and accepts a node when edge / sqrt(|x - com|^2 + 0.001*2.5) < theta or when it is
// a leaf.  Here the same tree exists only implicitly:
//   * every particle's path down that octree is a 63-bit key (3 bits per level, varh.hip uses the same
//     keys for the leaf boxes); the keys are radix-sorted;
//   * a binary radix tree (Karras 2012) over the sorted keys gives all ranges of particles that share a
//     key prefix.  A binary node whose prefix length crosses a multiple of 3 IS an octree node (and, with
//     it, the whole chain of single-child octree nodes above it: same particles, same centre of mass,
//     larger edges); the others are partial groups that the reference never tests.  The walk therefore
//     tests acceptance only on the former, with the smallest edge of the chain (if any node of the chain
//     is accepted the contribution is the same), and opens the latter unconditionally;
//   * masses and centres of mass are summed bottom-up (pairwise, fixed order -> reproducible);
//   * the walk is stackless (rope pointers), one lane per target particle in cell-sorted order, so the
//     lanes of a wave walk nearly the same nodes.
#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>

#include <cmath>
#include <cstdio>
#include <cstdlib>

#include "pair_common.hpp"

namespace sph {

namespace {

constexpr int GB = 256;
constexpr int LEVELS = 21;
constexpr int END = 0x7fffffff;          // rope terminator

struct RootBox { double c[3]; double size; };

__global__ __launch_bounds__(GB) void grav_keys(RootBox rb, const double4 *__restrict__ drec, int64_t n,
                                                uint64_t *__restrict__ keys, uint32_t *__restrict__ vals) {
    const int64_t i = (int64_t)blockIdx.x * GB + threadIdx.x;
    if (i >= n) return;
    const double4 p = drec[i];
    double cx = rb.c[0], cy = rb.c[1], cz = rb.c[2], size = rb.size;
    uint64_t key = 0;
    for (int l = 0; l < LEVELS; l++) {                        // [F]:208-217, 190-198
        const int bx = p.x > cx, by = p.y > cy, bz = p.z > cz;
        key = (key << 3) | (uint64_t)(bx | (by << 1) | (bz << 2));
        const double q = 0.25 * size;
        cx = cx + (bx ? q : -q); cy = cy + (by ? q : -q); cz = cz + (bz ? q : -q);
        size = size * 0.5;
    }
    keys[i] = key;
    vals[i] = (uint32_t)i;
}

// common prefix length (in bits of the 64-bit word) of sorted keys i and j; ties are broken by the index
__device__ __forceinline__ int delta(const uint64_t *__restrict__ k, int n, int i, int j) {
    if (j < 0 || j >= n) return -1;
    const uint64_t x = k[i] ^ k[j];
    if (x == 0) return 64 + __clz(i ^ j);
    return __clzll((long long)x);
}

// node ids: internal i in [0, n-2] as i; leaf j as ~j (negative)
struct TreeArrays {
    int32_t *left, *right, *parent;     // internal nodes
    int32_t *leaf_parent;               // leaves
    int32_t *prefix;                    // internal: common prefix length (bits of the 64-bit word)
    int32_t *other;                     // internal: the far end j of the node's leaf range [min(i,j), max(i,j)]
    double4 *seg;                       // segment tree over the leaves in key order (levels 1.., see seg_stage)
    double4 *sum;                       // internal: sum m x, sum m y, sum m z, sum m  -> later com + mass
    double4 *leafA;                     // leaves in key order: x y z m
    int32_t *slot;                      // leaf -> cell-sorted slot
    int32_t *lvl;                       // internal: octree level of the node's smallest box, or -1 (partial group)
    int32_t *rope, *leaf_rope;          // next node after this subtree in depth-first order
    int4 *walkB;                        // internal: {first octree node / leaf below, rope, level, -} with partial
                                        // (non-octree) binary nodes skipped on both pointers
    int2 *leafB;                        // leaves: {rope (partial nodes skipped), cell-sorted slot}
};

__global__ __launch_bounds__(GB) void radix_tree(const uint64_t *__restrict__ k, int n, TreeArrays t) {
    const int i = blockIdx.x * GB + threadIdx.x;
    if (i >= n - 1) return;
    const int d = delta(k, n, i, i + 1) > delta(k, n, i, i - 1) ? 1 : -1;
    const int dmin = delta(k, n, i, i - d);
    int lmax = 2;
    while (delta(k, n, i, i + lmax * d) > dmin) lmax <<= 1;
    int l = 0;
    for (int s = lmax >> 1; s >= 1; s >>= 1)
        if (delta(k, n, i, i + (l + s) * d) > dmin) l += s;
    const int j = i + l * d;
    const int dnode = delta(k, n, i, j);
    int s = 0;
    int tt = l;
    do {
        tt = (tt + 1) >> 1;
        if (delta(k, n, i, i + (s + tt) * d) > dnode) s += tt;
    } while (tt > 1);
    const int gamma = i + s * d + min(d, 0);
    const int lo = min(i, j), hi = max(i, j);
    const int L = lo == gamma ? ~gamma : gamma;
    const int Rr = hi == gamma + 1 ? ~(gamma + 1) : gamma + 1;
    t.left[i] = L; t.right[i] = Rr;
    t.prefix[i] = dnode;
    if (L < 0) t.leaf_parent[~L] = i; else t.parent[L] = i;
    if (Rr < 0) t.leaf_parent[~Rr] = i; else t.parent[Rr] = i;
    if (i == 0) t.parent[0] = -1;
    t.other[i] = j;
}

__global__ __launch_bounds__(GB) void leaf_data(const uint32_t *__restrict__ vals, const double4 *__restrict__ drec, int n,
                                                TreeArrays t) {
    const int j = blockIdx.x * GB + threadIdx.x;
    if (j >= n) return;
    const uint32_t s = vals[j];
    t.leafA[j] = drec[s];
    t.slot[j] = (int32_t)s;
}

// Node sums.  Every internal node of the radix tree covers a contiguous range of leaves (key order), so its mass
// and first moments are a range sum over the leaves.  A segment tree over the leaf array (level l entry k = sum of
// leaves [k 2^l, (k+1) 2^l), built in stages of SEG_LV levels per launch, LDS ping-pong) answers every range with
// 2 log2(length) entries, summed small to large on either side -- pairwise accuracy, a fixed order (reproducible),
// and no communication between workgroups (a bottom-up pass with arrival counters needs device-scope fences, which
// write back the XCD's L2 on this chip: 5.8 ms for 1e6 leaves, against 0.1 ms for this scheme).
constexpr int SEG_LV = 9;                       // levels per stage: a block reduces 512 entries
constexpr int SEG_MAX_LEVELS = 40;
struct SegLevels {
    int64_t off[SEG_MAX_LEVELS];                // off[l]: first entry of level l in seg[] (level 0 = the leaves, not stored)
    int64_t cnt[SEG_MAX_LEVELS];
    int levels;
};

__device__ __forceinline__ double4 leaf_moment(const double4 p) { return make_double4(p.w * p.x, p.w * p.y, p.w * p.z, p.w); }
__device__ __forceinline__ double4 add4(const double4 a, const double4 b) { return make_double4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }

// stage: from level L0 (the leaves if L0 == 0) up to level L0 + SEG_LV
__global__ __launch_bounds__(256) void seg_stage(SegLevels sl, int L0, const double4 *__restrict__ leafA, double4 *__restrict__ seg) {
    __shared__ double4 buf[2][512];
    const int64_t base = (int64_t)blockIdx.x * 512;
    const int64_t c0 = sl.cnt[L0];
    for (int e = threadIdx.x; e < 512; e += 256) {
        const int64_t k = base + e;
        double4 v = make_double4(0, 0, 0, 0);
        if (k < c0) v = L0 == 0 ? leaf_moment(leafA[k]) : seg[sl.off[L0] + k];
        buf[0][e] = v;
    }
    __syncthreads();
    int cur = 0;
    for (int l = 1; l <= SEG_LV && L0 + l < sl.levels; l++) {
        const int width = 512 >> l;
        const int64_t kb = base >> l;
        for (int e = threadIdx.x; e < width; e += 256) {
            const double4 v = add4(buf[cur][2 * e], buf[cur][2 * e + 1]);     // entries past the end are zero
            buf[cur ^ 1][e] = v;
            if (kb + e < sl.cnt[L0 + l]) seg[sl.off[L0 + l] + kb + e] = v;
        }
        __syncthreads();
        cur ^= 1;
    }
}

__global__ __launch_bounds__(GB) void node_sums_seg(int n, SegLevels sl, TreeArrays t) {
    const int i = blockIdx.x * GB + threadIdx.x;
    if (i >= n - 1) return;
    const int j = t.other[i];
    int64_t l = min(i, j), r = (int64_t)max(i, j) + 1;
    double4 accl = make_double4(0, 0, 0, 0), accr = make_double4(0, 0, 0, 0);
    int lev = 0;
    while (l < r) {
        if (l & 1) { accl = add4(accl, lev == 0 ? leaf_moment(t.leafA[l]) : t.seg[sl.off[lev] + l]); l++; }
        if (r & 1) { r--; accr = add4(accr, lev == 0 ? leaf_moment(t.leafA[r]) : t.seg[sl.off[lev] + r]); }
        l >>= 1; r >>= 1; lev++;
    }
    t.sum[i] = add4(accl, accr);
}

// centre of mass, octree level, ropes
__global__ __launch_bounds__(GB) void node_finish(int n, TreeArrays t) {
    const int i = blockIdx.x * GB + threadIdx.x;
    if (i < n - 1) {
        const double4 s = t.sum[i];
        t.sum[i] = s.w > 0.0 ? make_double4(s.x / s.w, s.y / s.w, s.z / s.w, s.w) : make_double4(0, 0, 0, 0);   // [F]:173-177
        // key bits shared by the node's particles: prefix - 1 (the 64-bit word has a leading zero bit)
        const int bits = min(t.prefix[i] - 1, 3 * LEVELS);
        const int par = t.parent[i];
        const int pbits = par < 0 ? -1 : min(t.prefix[par] - 1, 3 * LEVELS);
        const int lv = bits / 3;
        t.lvl[i] = (3 * lv > pbits) ? lv : -1;
        // rope: next node after this subtree
        int x = i, r = END;
        while (x != 0) {
            const int p = t.parent[x];
            if (t.left[p] == x) { r = t.right[p]; break; }
            x = p;
        }
        t.rope[i] = r;
    }
    if (i < n) {
        int r = END;
        if (n >= 2) {
            int p = t.leaf_parent[i];
            if (t.left[p] == ~i) r = t.right[p];
            else {
                int x = p;
                while (x != 0) {
                    const int pp = t.parent[x];
                    if (t.left[pp] == x) { r = t.right[pp]; break; }
                    x = pp;
                }
            }
        }
        t.leaf_rope[i] = r;
    }
}

// The walk only ever TESTS octree nodes and leaves; partial binary nodes are opened unconditionally, so both
// pointers of the walk records jump over them: eff(x) = x if x is a leaf or an octree node, else eff(left(x)).
__device__ __forceinline__ int eff_node(const TreeArrays &t, int x) {
    while (x != END && x >= 0 && t.lvl[x] < 0) x = t.left[x];
    return x;
}

__global__ __launch_bounds__(GB) void node_walk_records(int n, TreeArrays t) {
    const int i = blockIdx.x * GB + threadIdx.x;
    if (i < n - 1) t.walkB[i] = make_int4(eff_node(t, t.left[i]), eff_node(t, t.rope[i]), t.lvl[i], 0);
    if (i < n) t.leafB[i] = make_int2(eff_node(t, t.leaf_rope[i]), t.slot[i]);
}

// the walk, [F]:264-290.  Writes a = 0 - sum (the forces kernels continue from there, [F]:824-827).
__global__ __launch_bounds__(GB) void grav_walk(int n, TreeArrays t, RootBox rb, const double4 *__restrict__ drec,
                                                const double *__restrict__ hvar, double hfix, double soft2, double theta, double G,
                                                const double *__restrict__ gt, int nq, double dq, double *__restrict__ ax,
                                                double *__restrict__ ay, double *__restrict__ az, const int32_t *__restrict__ orig,
                                                int32_t n_owned) {
    const int i = xcd_chunk(blockIdx.x, gridDim.x) * GB + threadIdx.x;
    if (i >= n) return;
    if (orig[i] >= n_owned) return;
    const double4 p = drec[i];
    const double hp = hvar ? hvar[i] : hfix;
    const double inv_hp = 1.0 / hp, inv_dq = 1.0 / dq, theta2 = theta * theta;
    double a0 = 0.0, a1 = 0.0, a2 = 0.0;
    // the root is an octree node by construction (level >= 0)
    int node = n >= 2 ? 0 : END;
    while (node != END) {
        double4 c;
        int next_open, next_skip;
        bool leaf;
        double size = 0.0;
        if (node < 0) {
            const int j = ~node;
            const int2 lb = t.leafB[j];
            next_skip = lb.x; next_open = lb.x;
            if (lb.y == i) { node = next_skip; continue; }           // own leaf: direction = 0, contributes nothing
            c = t.leafA[j];
            leaf = true;
        } else {
            const int4 wb = t.walkB[node];
            c = t.sum[node];
            leaf = false;
            size = ldexp(rb.size, -wb.z);
            next_open = wb.x; next_skip = wb.y;
        }
        const double d0 = p.x - c.x, d1 = p.y - c.y, d2c = p.z - c.z;  // [F]:274
        const double d2 = fma(d2c, d2c, fma(d1, d1, fma(d0, d0, soft2)));    // [F]:275
        // [F]:278: size/dist < theta  <=>  size^2 < theta^2 d2 (all positive): opened nodes need neither the
        // square root nor the division (the two forms can only disagree within an ulp of the threshold)
        if (leaf || size * size < theta2 * d2) {
            if (c.w > 0.0) {
                const double rs = fast_rsqrt(d2);
                double f = (G * c.w) * ((rs * rs) * rs);              // [F]:281: G M [W] / dist^3
                const double qi = (d2 * rs) * inv_hp;                  // dist / h
                if (qi <= 2.0) {                                       // [F]:129-146: W = 1 beyond the softening support
                    const double tq = qi * inv_dq;
                    const int k = min((int)tq, nq - 1);
                    const double a = tq - (double)k;
                    f *= (1.0 - a) * gt[k] + a * gt[k + 1];
                }
                a0 = fma(-f, d0, a0); a1 = fma(-f, d1, a1); a2 = fma(-f, d2c, a2);
            }
            node = next_skip;
        } else {
            node = next_open;
        }
    }
    ax[i] = a0; ay[i] = a1; az[i] = a2;
}

// One 64-byte record per node for the wave walk, leaves behind the internal nodes (unified index: internal i -> i,
// leaf j -> n - 1 + j): centre of mass + G x mass, the squared edge of the node's smallest box (-1 for a leaf: always
// accepted), both successors, and for leaves the cell-sorted slot (to recognise the target's own leaf).
struct alignas(64) WalkRec {
    double cx, cy, cz, gm;              // centre of mass, G m (first 32 bytes: what the contribution needs, one scalar load)
    double size2;
    int32_t next_open, next_skip;       // unified indices, END terminates
    int32_t slot, has_mass;             // has_mass: m > 0 (a wave-uniform test in the walk, not a vector compare)
    double pad;
};

__device__ __forceinline__ int unified(int node, int n) { return node == END ? END : (node < 0 ? n - 1 + ~node : node); }

__global__ __launch_bounds__(GB) void node_wave_records(int n, TreeArrays t, RootBox rb, double theta2, double G, WalkRec *__restrict__ rec,
                                                        int32_t *__restrict__ leaf_of) {
    const int i = blockIdx.x * GB + threadIdx.x;
    if (i < n - 1) {
        const int4 wb = t.walkB[i];
        const double4 c = t.sum[i];
        const double size = ldexp(rb.size, -wb.z);
        WalkRec r{c.x, c.y, c.z, G * c.w, (size * size) / theta2, unified(wb.x, n), unified(wb.y, n), -1, c.w > 0.0 ? 1 : 0, 0.0};   // theta = 0.5: exact
        rec[i] = r;
    }
    if (i < n) {
        const int2 lb = t.leafB[i];
        const double4 c = t.leafA[i];
        const int nx = unified(lb.x, n);
        WalkRec r{c.x, c.y, c.z, G * c.w, -1.0, nx, nx, lb.y, c.w > 0.0 ? 1 : 0, 0.0};
        rec[n - 1 + i] = r;
        leaf_of[lb.y] = i;                   // cell-sorted slot -> leaf (key order)
    }
}

// The same walk, one WAVE per 64 consecutive targets of the cell-sorted order (a column of grid cells; taking them in
// key order instead makes the union of the 64 walks 24 % larger).  The wave visits the union of its lanes' walks in the same depth-first order: the
// current node is wave-uniform (its record comes through the scalar cache, one 64-byte read for 64 lanes), every
// active lane applies its own acceptance test, and the wave descends as soon as one lane needs to.  A lane that
// accepted the node sleeps until the walk leaves that subtree -- it wakes at the node's rope, which every exit
// from the subtree reaches.  Each lane therefore accumulates exactly the contributions of its own walk, in the same
// order as grav_walk (kept as the per-lane reference implementation, SPH_GRAV_WAVE=0).
template <bool STATS>
__global__ __launch_bounds__(GB) void grav_walk_wave(int nt, int n, const WalkRec *__restrict__ rec, const double4 *__restrict__ leafA,
                                                     const double4 *__restrict__ drec, const int32_t *__restrict__ leaf_of,
                                                     const double *__restrict__ hvar,
                                                     double hfix, double soft2, double theta, double G,
                                                     const double *__restrict__ gt, int nq, double dq, double *__restrict__ ax,
                                                     double *__restrict__ ay, double *__restrict__ az, const int32_t *__restrict__ orig,
                                                     int32_t n_owned, unsigned long long *__restrict__ stats, double rb_size) {
    // nt targets (the context's cell-sorted slots); the tree has n leaves: the same particles (leaf_of: slot -> leaf),
    // or an external source set (multi-GPU: every GPU's particles; leaf_of == nullptr).  A target's own leaf needs no
    // special case then: its direction is exactly 0 and it adds 0 * f.
    const int i = xcd_chunk(blockIdx.x, gridDim.x) * GB + threadIdx.x;       // target: cell-sorted slot
    const int self = i < nt ? i : nt - 1;
    const bool live = i < nt && orig[self] < n_owned;
    const int j = leaf_of ? leaf_of[self] : -1;                               // its leaf (key order)
    const double4 p = leaf_of ? leafA[j] : drec[self];
    const double hp = hvar ? hvar[self] : hfix;
    const double inv_hp = 1.0 / hp, inv_dq = 1.0 / dq, theta2 = theta * theta, rsoft2 = 4.0 * hp * hp;
    double a0 = 0.0, a1 = 0.0, a2 = 0.0;
    bool active = live;                       // the root is an octree node by construction
    int resume = END;
    int node = n >= 2 ? 0 : END;
    const int own = leaf_of ? n - 1 + j : END;    // unified index of the target's own leaf
    unsigned visits = 0, sums = 0;
    // (Requesting both successors' records before the acceptance tests run -- so that the scalar-load latency overlaps the
    // tests -- changes nothing: 4.06 ms per step either way; the walk is bound by vector issue, not by the pointer chase.)
    // Per visit the walk issues ~23 vector instructions where round 2's issued 33: which lanes are awake is a 64-bit mask in scalar
    // registers (compares deliver masks, __builtin_amdgcn_fcmp / icmp; a mask predicates vector code for free, inverse_ballot), G m
    // and "has mass" come with the record, W multiplies only inside the softening support, and the softening length enters the
    // squared distance through the first fma.
    constexpr int CMP_EQ = 32, CMP_NE = 33, CMP_OLT = 4;                  // llvm::CmpInst predicates of the two builtins
    unsigned long long act_m = __builtin_amdgcn_ballot_w64(active);       // lanes walking; the others sleep until their rope (resume)
    while (node != END) {
        node = __builtin_amdgcn_readfirstlane(node);
        const WalkRec r = rec[node];
        act_m |= __builtin_amdgcn_uicmp((unsigned)resume, (unsigned)node, CMP_EQ);      // a sleeper wakes at the rope of the node it accepted
        const int n_open = r.next_open, n_skip = r.next_skip;
        const double d0 = p.x - r.cx, d1 = p.y - r.cy, d2c = p.z - r.cz;  // [F]:274
        const double d2 = fma(d2c, d2c, fma(d1, d1, fma(d0, d0, soft2)));  // [F]:275
        // [F]:278, see grav_walk: size2 holds edge^2 / theta^2 (leaves: -1, always accepted)
        const unsigned long long acc_m = __builtin_amdgcn_fcmp(r.size2, d2, CMP_OLT);
        const unsigned long long done_m = act_m & acc_m;                  // awake and satisfied with this node
        const bool open_any = (act_m & ~acc_m) != 0;                      // some awake lane has to open it
        if (r.has_mass) {
            // own leaf: direction = 0, contributes nothing
            if (__builtin_amdgcn_inverse_ballot_w64(done_m & __builtin_amdgcn_uicmp((unsigned)own, (unsigned)node, CMP_NE))) {
                const double rs = fast_rsqrt(d2);
                double f = r.gm * ((rs * rs) * rs);                       // [F]:281: G M [W] / dist^3
                if (d2 <= rsoft2) {
                    const double qi = (d2 * rs) * inv_hp;                  // dist / h
                    if (qi <= 2.0) {                                       // [F]:129-146: W = 1 beyond the softening support
                        const double tq = qi * inv_dq;
                        const int k = min((int)tq, nq - 1);
                        const double a = tq - (double)k;
                        f *= (1.0 - a) * gt[k] + a * gt[k + 1];
                    }
                }
                a0 = fma(-f, d0, a0); a1 = fma(-f, d1, a1); a2 = fma(-f, d2c, a2);
                if (STATS) sums++;
            }
        }
        if (STATS) visits++;
        if (STATS && (threadIdx.x & 63) == 0) {          // debug histogram: visits by node size (levels below the root)
            const double ratio = r.size2 > 0.0 ? r.size2 * theta2 / (rb_size * rb_size) : 0.0;
            int lv = 0;
            while (lv < 15 && ratio > 0.0 && ratio < 1.0 / (double)(1ull << (2 * lv))) lv++;
            atomicAdd(&stats[2 + (r.size2 > 0.0 ? lv : 15)], 1ull);
        }
        if (open_any) {
            resume = __builtin_amdgcn_inverse_ballot_w64(done_m) ? n_skip : resume;      // done with this subtree: sleep until its rope
            act_m &= ~done_m;
            node = n_open;
        } else {
            node = n_skip;
        }
    }
    if (live) { ax[i] = a0; ay[i] = a1; az[i] = a2; }
    if (STATS) {
        if ((threadIdx.x & 63) == 0) atomicAdd(&stats[0], (unsigned long long)visits);
        atomicAdd(&stats[1], (unsigned long long)sums);
    }
}

}  // namespace

#define GR_CHECK2(expr)                                                     \
    do {                                                                    \
        hipError_t _e = (expr);                                             \
        if (_e != hipSuccess) {                                             \
            c->err = std::string(#expr) + ": " + hipGetErrorString(_e);     \
            return SPH_ERR_HIP;                                             \
        }                                                                   \
    } while (0)

hipError_t grav_sort_tmp_bytes(int64_t n, size_t *bytes) {
    size_t b = 0;
    hipError_t e = rocprim::radix_sort_pairs(nullptr, b, (uint64_t *)nullptr, (uint64_t *)nullptr, (uint32_t *)nullptr,
                                             (uint32_t *)nullptr, (size_t)n, 0u, 63u, (hipStream_t) nullptr);
    *bytes = b;
    return e;
}

static TreeArrays tree_arrays(sph_ctx *c) {
    TreeArrays t;
    t.left = c->g_left; t.right = c->g_right; t.parent = c->g_parent; t.leaf_parent = c->g_leaf_parent; t.prefix = c->g_prefix;
    t.other = c->g_flag; t.seg = reinterpret_cast<double4 *>(c->g_seg); t.sum = reinterpret_cast<double4 *>(c->g_sum); t.leafA = reinterpret_cast<double4 *>(c->g_leafA);
    t.slot = c->g_slot; t.lvl = c->g_lvl; t.rope = c->g_rope; t.leaf_rope = c->g_leaf_rope;
    t.walkB = reinterpret_cast<int4 *>(c->g_walkB); t.leafB = reinterpret_cast<int2 *>(c->g_leafB);
    return t;
}

void gravity_free(sph_ctx *c) {
    ctx_free(c, c->g_left); ctx_free(c, c->g_right); ctx_free(c, c->g_parent); ctx_free(c, c->g_leaf_parent);
    ctx_free(c, c->g_prefix); ctx_free(c, c->g_flag); ctx_free(c, c->g_slot); ctx_free(c, c->g_lvl); ctx_free(c, c->g_rope);
    ctx_free(c, c->g_leaf_rope); ctx_free(c, c->g_sum); ctx_free(c, c->g_seg); ctx_free(c, c->g_wrec); ctx_free(c, c->g_leaf_of);
    ctx_free(c, c->g_leafA); ctx_free(c, c->g_walkB); ctx_free(c, c->g_leafB);
    ctx_free(c, c->g_keys); ctx_free(c, c->g_keys_alt); ctx_free(c, c->g_vals); ctx_free(c, c->g_vals_alt);
    ctx_free_ptr(c, c->g_sort_tmp); c->g_sort_tmp = nullptr; c->g_sort_tmp_bytes = 0;
    c->g_cap = 0; c->gx_keys_valid = false;
}

// tree arrays for `need` leaves (the context's slots, or an external source set that may be much larger)
static int gravity_reserve(sph_ctx *c, int64_t need) {
    if (need <= c->g_cap) return SPH_OK;
    gravity_free(c);
    const size_t cap = (size_t)(need + need / 16 + 64);
#define G_ALLOC(p, cnt, what) do { int _s = ctx_alloc(c, &(p), (cnt), what); if (_s != SPH_OK) return _s; } while (0)
    G_ALLOC(c->g_left, cap, "tree left"); G_ALLOC(c->g_right, cap, "tree right"); G_ALLOC(c->g_parent, cap, "tree parent");
    G_ALLOC(c->g_leaf_parent, cap, "leaf parent"); G_ALLOC(c->g_prefix, cap, "tree prefix"); G_ALLOC(c->g_flag, cap, "range ends");
    G_ALLOC(c->g_slot, cap, "leaf slots"); G_ALLOC(c->g_lvl, cap, "tree levels"); G_ALLOC(c->g_rope, cap, "tree ropes");
    G_ALLOC(c->g_leaf_rope, cap, "leaf ropes"); G_ALLOC(c->g_sum, cap * 4, "node sums"); G_ALLOC(c->g_seg, (cap + 64) * 4, "segment tree");
    G_ALLOC(c->g_wrec, cap * 16, "wave walk records"); G_ALLOC(c->g_leaf_of, cap, "slot -> leaf"); G_ALLOC(c->g_leafA, cap * 4, "leaf records");
    G_ALLOC(c->g_walkB, cap * 4, "walk records"); G_ALLOC(c->g_leafB, cap * 2, "leaf walk records");
    G_ALLOC(c->g_keys, cap, "tree keys"); G_ALLOC(c->g_keys_alt, cap, "tree keys (alt)");
    G_ALLOC(c->g_vals, cap, "tree vals"); G_ALLOC(c->g_vals_alt, cap, "tree vals (alt)");
#undef G_ALLOC
    size_t tmp = 0;
    GR_CHECK2(grav_sort_tmp_bytes((int64_t)cap, &tmp));
    if (ctx_alloc_bytes(c, &c->g_sort_tmp, tmp ? tmp : 1, "tree sort scratch") != SPH_OK) return SPH_ERR_NOMEM;
    c->g_sort_tmp_bytes = tmp;
    c->g_cap = (int64_t)cap;
    return SPH_OK;
}

// path keys of the external source set, sorted (shared by the gravity tree, the variable-h leaf boxes and accretion)
int global_keys_sorted(sph_ctx *c) {
    if (!c->gx_src) { c->err = "no external source set"; return SPH_ERR_STATE; }
    if (c->gx_keys_valid) return SPH_OK;
    const int64_t n = c->gx_n;
    { const int st = gravity_reserve(c, n); if (st != SPH_OK) return st; }
    RootBox rb;
    double size = 0.0;
    for (int a = 0; a < 3; a++) {
        rb.c[a] = (c->gx_box[3 + a] + c->gx_box[a]) / 2.0;
        size = std::max(size, c->gx_box[3 + a] - c->gx_box[a]);
    }
    rb.size = size;
    grav_keys<<<dim3((unsigned)((n + GB - 1) / GB)), dim3(GB), 0, c->stream>>>(rb, reinterpret_cast<const double4 *>(c->gx_src), n, c->g_keys, c->g_vals);
    GR_CHECK2(hipGetLastError());
    size_t tmp = c->g_sort_tmp_bytes;
    GR_CHECK2(rocprim::radix_sort_pairs(c->g_sort_tmp, tmp, c->g_keys, c->g_keys_alt, c->g_vals, c->g_vals_alt, (size_t)n, 0u, 63u, c->stream));
    c->gx_keys_valid = true;
    return SPH_OK;
}

// builds the tree over the context's own particles (current cell-sorted positions) or over the external source set
int gravity_tree_build(sph_ctx *c) {
    const bool ext = c->gx_src != nullptr;
    const int64_t n = ext ? c->gx_n : c->n;
    if (n == 0) return SPH_OK;
    if (n > 2000000000LL) { c->err = "gravity: more than 2e9 sources"; return SPH_ERR_ARG; }
    { const int st = gravity_reserve(c, ext ? n : std::max(c->cap, n)); if (st != SPH_OK) return st; }
    const double *bb = ext ? c->gx_box : c->bbox;
    RootBox rb;
    double size = 0.0;
    for (int a = 0; a < 3; a++) {
        rb.c[a] = (bb[3 + a] + bb[a]) / 2.0;                            // [F]:803-805
        size = std::max(size, bb[3 + a] - bb[a]);                       // [F]:806-808
    }
    rb.size = size;
    for (int a = 0; a < 3; a++) c->root_box[a] = rb.c[a];
    c->root_box[3] = size;
    const unsigned gb = (unsigned)((n + GB - 1) / GB);
    const double4 *drec = reinterpret_cast<const double4 *>(ext ? c->gx_src : c->drec);
    if (ext) {
        const int st = global_keys_sorted(c);
        if (st != SPH_OK) return st;
    } else {
        if (c->path_keys_valid && c->mkeys_alt && c->mvals_alt) {
            // variable h: the leaf-box build of this grid build sorted the same keys (same positions, same root box)
            GR_CHECK2(hipMemcpyAsync(c->g_keys_alt, c->mkeys_alt, (size_t)n * sizeof(uint64_t), hipMemcpyDeviceToDevice, c->stream));
            GR_CHECK2(hipMemcpyAsync(c->g_vals_alt, c->mvals_alt, (size_t)n * sizeof(uint32_t), hipMemcpyDeviceToDevice, c->stream));
        } else {
            grav_keys<<<dim3(gb), dim3(GB), 0, c->stream>>>(rb, drec, n, c->g_keys, c->g_vals);
            GR_CHECK2(hipGetLastError());
            size_t tmp = c->g_sort_tmp_bytes;
            GR_CHECK2(rocprim::radix_sort_pairs(c->g_sort_tmp, tmp, c->g_keys, c->g_keys_alt, c->g_vals, c->g_vals_alt, (size_t)n, 0u, 63u, c->stream));
        }
    }
    TreeArrays t = tree_arrays(c);
    leaf_data<<<dim3(gb), dim3(GB), 0, c->stream>>>(c->g_vals_alt, drec, (int)n, t);
    if (n >= 2) {
        radix_tree<<<dim3(gb), dim3(GB), 0, c->stream>>>(c->g_keys_alt, (int)n, t);
        SegLevels sl{};
        sl.cnt[0] = n; sl.off[0] = 0; sl.levels = 1;
        int64_t off = 0;
        while (sl.cnt[sl.levels - 1] > 1 && sl.levels < SEG_MAX_LEVELS) {
            sl.cnt[sl.levels] = (sl.cnt[sl.levels - 1] + 1) / 2;
            sl.off[sl.levels] = off;
            off += sl.cnt[sl.levels];
            sl.levels++;
        }
        for (int L0 = 0; L0 + 1 < sl.levels; L0 += SEG_LV)
            seg_stage<<<dim3((unsigned)((sl.cnt[L0] + 511) / 512)), dim3(256), 0, c->stream>>>(sl, L0, t.leafA, t.seg);
        node_sums_seg<<<dim3(gb), dim3(GB), 0, c->stream>>>((int)n, sl, t);
    }
    node_finish<<<dim3(gb), dim3(GB), 0, c->stream>>>((int)n, t);
    node_walk_records<<<dim3(gb), dim3(GB), 0, c->stream>>>((int)n, t);
    node_wave_records<<<dim3(gb), dim3(GB), 0, c->stream>>>((int)n, t, rb, c->p.theta * c->p.theta, c->p.G, reinterpret_cast<WalkRec *>(c->g_wrec), c->g_leaf_of);
    GR_CHECK2(hipGetLastError());
    return SPH_OK;
}

hipError_t launch_gravity(sph_ctx *c) {
    const bool ext = c->gx_src != nullptr;
    const int64_t nt = c->n;                     // targets: the context's slots
    const int64_t n = ext ? c->gx_n : c->n;      // leaves of the tree
    if (nt == 0 || n == 0) return hipSuccess;
    RootBox rb;
    for (int a = 0; a < 3; a++) rb.c[a] = c->root_box[a];
    rb.size = c->root_box[3];
    const double soft2 = 0.001 * 2.5;                                   // 0.001_dp*smoothing, MODULE constant ([F]:275, [V]:296)
    static int wave_walk = -1;
    if (wave_walk < 0) { const char *e = getenv("SPH_GRAV_WAVE"); wave_walk = e ? atoi(e) : 1; }
    if (wave_walk || ext) {
        TreeArrays ta = tree_arrays(c);
        unsigned long long *stats = nullptr;
        if (wave_walk == 2) {                                           // debug: visit / contribution counts
            if (hipMalloc(reinterpret_cast<void **>(&stats), 18 * 8) != hipSuccess) return hipGetLastError();
            (void)hipMemsetAsync(stats, 0, 18 * 8, c->stream);
        }
        auto walk = stats ? grav_walk_wave<true> : grav_walk_wave<false>;
        walk<<<dim3((unsigned)((nt + GB - 1) / GB)), dim3(GB), 0, c->stream>>>(
            (int)nt, (int)n, reinterpret_cast<const WalkRec *>(c->g_wrec), ta.leafA, reinterpret_cast<const double4 *>(c->drec),
            ext ? nullptr : c->g_leaf_of, c->variable ? c->f[SPH_F_H] : nullptr, c->p.h, soft2,
            c->p.theta, c->p.G, c->grav_tab, c->p.nq, 2.0 / c->p.nq, c->f[SPH_F_AX], c->f[SPH_F_AY], c->f[SPH_F_AZ], c->orig,
            (int32_t)c->n_owned, stats, rb.size);
        if (stats) {
            unsigned long long h[18] = {0};
            (void)hipStreamSynchronize(c->stream);
            (void)hipMemcpy(h, stats, 18 * 8, hipMemcpyDeviceToHost);
            fprintf(stderr, "[grav_walk_wave] visits per wave by level (0..14, leaves):");
            for (int k = 0; k < 16; k++) fprintf(stderr, " %.1f", (double)h[2 + k] / (double)((nt + 63) / 64));
            fprintf(stderr, "\n");
            (void)hipFree(stats);
            fprintf(stderr, "[grav_walk_wave] n=%lld wave visits/wave=%.1f contributions/particle=%.1f\n", (long long)n,
                    (double)h[0] / (double)((n + 63) / 64), (double)h[1] / (double)n);
        }
        return hipGetLastError();
    }
    grav_walk<<<dim3((unsigned)((n + GB - 1) / GB)), dim3(GB), 0, c->stream>>>(
        (int)n, tree_arrays(c), rb, reinterpret_cast<const double4 *>(c->drec), c->variable ? c->f[SPH_F_H] : nullptr, c->p.h, soft2,
        c->p.theta, c->p.G, c->grav_tab, c->p.nq, 2.0 / c->p.nq, c->f[SPH_F_AX], c->f[SPH_F_AY], c->f[SPH_F_AZ], c->orig,
        (int32_t)c->n_owned);
    return hipGetLastError();
}

}  // namespace sph
