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
    int32_t *flag;                      // bottom-up arrival counters
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
    t.flag[i] = 0;
}

__global__ __launch_bounds__(GB) void leaf_data(const uint32_t *__restrict__ vals, const double4 *__restrict__ drec, int n,
                                                TreeArrays t) {
    const int j = blockIdx.x * GB + threadIdx.x;
    if (j >= n) return;
    const uint32_t s = vals[j];
    t.leafA[j] = drec[s];
    t.slot[j] = (int32_t)s;
}

// bottom-up sums: the second thread to arrive at a node finds both children finished
__global__ __launch_bounds__(GB) void node_sums(int n, TreeArrays t) {
    const int j = blockIdx.x * GB + threadIdx.x;
    if (j >= n || n < 2) return;
    int node = t.leaf_parent[j];
    while (node >= 0) {
        __threadfence();
        if (atomicAdd(&t.flag[node], 1) == 0) return;
        __threadfence();
        const int L = t.left[node], Rr = t.right[node];
        double4 a, b;
        if (L < 0) { const double4 p = t.leafA[~L]; a = make_double4(p.w * p.x, p.w * p.y, p.w * p.z, p.w); }
        else { const double *q = reinterpret_cast<const double *>(&t.sum[L]);
               a = make_double4(__hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT),
                                __hip_atomic_load(q + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), __hip_atomic_load(q + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)); }
        if (Rr < 0) { const double4 p = t.leafA[~Rr]; b = make_double4(p.w * p.x, p.w * p.y, p.w * p.z, p.w); }
        else { const double *q = reinterpret_cast<const double *>(&t.sum[Rr]);
               b = make_double4(__hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT),
                                __hip_atomic_load(q + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), __hip_atomic_load(q + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)); }
        double *o = reinterpret_cast<double *>(&t.sum[node]);
        __hip_atomic_store(o, a.x + b.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(o + 1, a.y + b.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(o + 2, a.z + b.z, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(o + 3, a.w + b.w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        node = t.parent[node];
    }
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
        const double d2 = (d0 * d0 + d1 * d1 + d2c * d2c) + soft2;    // [F]:275
        // [F]:278: size/dist < theta  <=>  size^2 < theta^2 d2 (all positive): opened nodes need neither the
        // square root nor the division (the two forms can only disagree within an ulp of the threshold)
        if (leaf || size * size < theta2 * d2) {
            if (c.w > 0.0) {
                double dist, rs;
                fast_sqrt_rsqrt(d2, dist, rs);
                const double qi = dist * inv_hp;
                double W = 1.0;                                        // [F]:129-146: 1 beyond the softening support
                if (qi <= 2.0) {
                    const double tq = qi * inv_dq;
                    const int k = min((int)tq, nq - 1);
                    const double a = tq - (double)k;
                    W = (1.0 - a) * gt[k] + a * gt[k + 1];
                }
                const double f = (G * c.w) * W * (rs * rs * rs);      // [F]:281: G M W / dist^3
                a0 = fma(-f, d0, a0); a1 = fma(-f, d1, a1); a2 = fma(-f, d2c, a2);
            }
            node = next_skip;
        } else {
            node = next_open;
        }
    }
    ax[i] = a0; ay[i] = a1; az[i] = a2;
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
    t.flag = c->g_flag; t.sum = reinterpret_cast<double4 *>(c->g_sum); t.leafA = reinterpret_cast<double4 *>(c->g_leafA);
    t.slot = c->g_slot; t.lvl = c->g_lvl; t.rope = c->g_rope; t.leaf_rope = c->g_leaf_rope;
    t.walkB = reinterpret_cast<int4 *>(c->g_walkB); t.leafB = reinterpret_cast<int2 *>(c->g_leafB);
    return t;
}

// builds the tree for the current (cell-sorted) positions
int gravity_tree_build(sph_ctx *c) {
    const int64_t n = c->n;
    if (n == 0) return SPH_OK;
    RootBox rb;
    double size = 0.0;
    for (int a = 0; a < 3; a++) {
        rb.c[a] = (c->bbox[3 + a] + c->bbox[a]) / 2.0;                  // [F]:803-805
        size = std::max(size, c->bbox[3 + a] - c->bbox[a]);             // [F]:806-808
    }
    rb.size = size;
    for (int a = 0; a < 3; a++) c->root_box[a] = rb.c[a];
    c->root_box[3] = size;
    const unsigned gb = (unsigned)((n + GB - 1) / GB);
    const double4 *drec = reinterpret_cast<const double4 *>(c->drec);
    grav_keys<<<dim3(gb), dim3(GB), 0, c->stream>>>(rb, drec, n, c->mkeys, c->mvals);
    GR_CHECK2(hipGetLastError());
    size_t tmp = c->msort_tmp_bytes;
    GR_CHECK2(rocprim::radix_sort_pairs(c->msort_tmp, tmp, c->mkeys, c->mkeys_alt, c->mvals, c->mvals_alt, (size_t)n, 0u, 63u, c->stream));
    TreeArrays t = tree_arrays(c);
    leaf_data<<<dim3(gb), dim3(GB), 0, c->stream>>>(c->mvals_alt, drec, (int)n, t);
    if (n >= 2) {
        radix_tree<<<dim3(gb), dim3(GB), 0, c->stream>>>(c->mkeys_alt, (int)n, t);
        node_sums<<<dim3(gb), dim3(GB), 0, c->stream>>>((int)n, t);
    }
    node_finish<<<dim3(gb), dim3(GB), 0, c->stream>>>((int)n, t);
    node_walk_records<<<dim3(gb), dim3(GB), 0, c->stream>>>((int)n, t);
    GR_CHECK2(hipGetLastError());
    return SPH_OK;
}

hipError_t launch_gravity(sph_ctx *c) {
    const int64_t n = c->n;
    if (n == 0) return hipSuccess;
    RootBox rb;
    for (int a = 0; a < 3; a++) rb.c[a] = c->root_box[a];
    rb.size = c->root_box[3];
    const double soft2 = 0.001 * 2.5;                                   // 0.001_dp*smoothing, MODULE constant ([F]:275, [V]:296)
    grav_walk<<<dim3((unsigned)((n + GB - 1) / GB)), dim3(GB), 0, c->stream>>>(
        (int)n, tree_arrays(c), rb, reinterpret_cast<const double4 *>(c->drec), c->variable ? c->f[SPH_F_H] : nullptr, c->p.h, soft2,
        c->p.theta, c->p.G, c->grav_tab, c->p.nq, 2.0 / c->p.nq, c->f[SPH_F_AX], c->f[SPH_F_AY], c->f[SPH_F_AZ], c->orig,
        (int32_t)c->n_owned);
    return hipGetLastError();
}

}  // namespace sph
