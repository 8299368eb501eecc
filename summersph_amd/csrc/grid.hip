// grid.hip -- the neighbour index: uniform cell grid + radix-sorted particle order.
//
// Replaces the reference's pointer octree (create_tree / build_tree,
// /root/reference/SUMMER_SPH.f90:795-816,149-246) for the fixed-h path.  Only the octree's
// SEMANTICS are kept: for fixed h the tree walk visits exactly {j : |x_i - x_j| <= 2h}
// (SURVEY.md 3.2), which a grid of edge 2h and a 27-cell stencil also covers.
//
// Pipeline per rebuild (all on ctx->stream):
//   bbox_partial/bbox_final  -> bounding box (one small read-back: the host sizes the grid)
//   cell_keys                -> 32-bit key per particle, axis with fewest cells fastest
//   rocprim radix sort       -> stable (key, slot) sort on only as many bits as the grid needs
//   cell_table               -> cell_start[c] = first sorted slot with key >= c
//   reorder                  -> gathers the 9 state arrays + ids into sorted order and writes
//                               the 32-byte density gather record {x,y,z,m}
#include <cstdlib>
#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#include <cmath>
#include <utility>

#include "sph_internal.hpp"

namespace sph {

namespace {

constexpr int BB_BLOCK = 256;
constexpr int BB_MAX_BLOCKS = 1024;

__device__ __forceinline__ double wave_min(double v) {
    for (int o = 32; o > 0; o >>= 1) v = fmin(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ double wave_max(double v) {
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, 64));
    return v;
}

// partial[b*6 + {0..2}] = min x,y,z ; {3..5} = max x,y,z over block b's grid-stride share
// mode 0: every slot except replaced ghosts (slot < dead_below with original id >= n_owned); mode 1: owned only
__global__ __launch_bounds__(BB_BLOCK) void bbox_partial(const double *__restrict__ x, const double *__restrict__ y,
                                                         const double *__restrict__ z, int64_t n,
                                                         double *__restrict__ partial, int32_t *__restrict__ flags,
                                                         const int32_t *__restrict__ orig, int32_t n_owned,
                                                         int64_t dead_below, int owned_only) {
    __shared__ double sm[6][BB_BLOCK / WAVE];
    double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    bool bad = false;
    for (int64_t i = (int64_t)blockIdx.x * BB_BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BB_BLOCK) {
        if ((owned_only || i < dead_below) && orig[i] >= n_owned) continue;
        double v[3] = {x[i], y[i], z[i]};
#pragma unroll
        for (int a = 0; a < 3; a++) {
            bad |= !isfinite(v[a]);
            lo[a] = fmin(lo[a], v[a]);
            hi[a] = fmax(hi[a], v[a]);
        }
    }
    if (bad) flags[0] = 1;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int a = 0; a < 3; a++) {
        double mn = wave_min(lo[a]), mx = wave_max(hi[a]);
        if (lane == 0) { sm[a][wv] = mn; sm[3 + a][wv] = mx; }
    }
    __syncthreads();
    if (threadIdx.x < 6) {
        double r = sm[threadIdx.x][0];
        for (int k = 1; k < BB_BLOCK / WAVE; k++)
            r = threadIdx.x < 3 ? fmin(r, sm[threadIdx.x][k]) : fmax(r, sm[threadIdx.x][k]);
        partial[(int64_t)blockIdx.x * 6 + threadIdx.x] = r;
    }
}

// 6 waves, one per bbox component; lanes stride over the per-block partials.  host_slot (pinned host memory, mapped into the
// device's address space) gets the box and the non-finite flag directly: a separate device-to-host copy of 52 bytes costs
// a copy kernel of ~16 us on the stream (two of them per grid build were 3 % of the bench step)
__global__ __launch_bounds__(384) void bbox_final(const double *__restrict__ partial, int nblocks, double *__restrict__ out,
                                                  double *__restrict__ host_slot = nullptr, int32_t *__restrict__ flags = nullptr) {
    const int comp = threadIdx.x >> 6, lane = threadIdx.x & 63;
    double r = comp < 3 ? INFINITY : -INFINITY;
    for (int b = lane; b < nblocks; b += 64) {
        const double v = partial[b * 6 + comp];
        r = comp < 3 ? fmin(r, v) : fmax(r, v);
    }
    r = comp < 3 ? wave_min(r) : wave_max(r);
    if (lane == 0) {
        out[comp] = r;
        if (host_slot) {
            host_slot[comp] = r;
            if (comp == 0) { *reinterpret_cast<int32_t *>(host_slot + 6) = flags[0]; flags[0] = 0; }     // ... and cleared for the next build
        }
    }
}

// count, sum and sum of squares per axis of the live particles inside box (lo, hi): partial[b*7 + ...].  For the trimmed
// grid box of very sparse domains (grid_rebuild); two stages with a fixed order, like the bounding box.
struct Box6 { double lo[3], hi[3]; };
__global__ __launch_bounds__(BB_BLOCK) void moment_partial(const double *__restrict__ x, const double *__restrict__ y,
                                                           const double *__restrict__ z, int64_t n, Box6 box,
                                                           double *__restrict__ partial, const int32_t *__restrict__ orig,
                                                           int32_t n_owned, int64_t dead_below) {
    __shared__ double sm[7][BB_BLOCK / WAVE];
    double acc[7] = {0, 0, 0, 0, 0, 0, 0};
    for (int64_t i = (int64_t)blockIdx.x * BB_BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BB_BLOCK) {
        if (i < dead_below && orig[i] >= n_owned) continue;
        const double v[3] = {x[i], y[i], z[i]};
        const bool in = v[0] >= box.lo[0] && v[0] <= box.hi[0] && v[1] >= box.lo[1] && v[1] <= box.hi[1] && v[2] >= box.lo[2] && v[2] <= box.hi[2];
        if (!in) continue;
        acc[0] += 1.0;
#pragma unroll
        for (int a = 0; a < 3; a++) { acc[1 + a] += v[a]; acc[4 + a] += v[a] * v[a]; }
    }
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < 7; k++) {
        double r = acc[k];
        for (int o = 32; o > 0; o >>= 1) r += __shfl_xor(r, o, 64);
        if (lane == 0) sm[k][wv] = r;
    }
    __syncthreads();
    if (threadIdx.x < 7) {
        double r = 0.0;
        for (int k = 0; k < BB_BLOCK / WAVE; k++) r += sm[threadIdx.x][k];
        partial[(int64_t)blockIdx.x * 7 + threadIdx.x] = r;
    }
}

__global__ __launch_bounds__(448) void moment_final(const double *__restrict__ partial, int nblocks, double *__restrict__ out) {
    const int comp = threadIdx.x >> 6, lane = threadIdx.x & 63;
    double r = 0.0;
    for (int b = lane; b < nblocks; b += 64) r += partial[b * 7 + comp];
    for (int o = 32; o > 0; o >>= 1) r += __shfl_xor(r, o, 64);
    if (lane == 0) out[comp] = r;
}

__device__ __forceinline__ uint32_t cell_key(const GridDesc &g, double px, double py, double pz, int cc[3]) {
    const double p[3] = {px, py, pz};
    int c[3];
#pragma unroll
    for (int a = 0; a < 3; a++) {
        // clamped as a double BEFORE the cast: a particle far outside the (stale or trimmed) box, or a NaN that is reported a
        // build late, must not reach an out-of-range float -> int conversion (undefined in C++); fmax drops a NaN -> cell 0
        c[a] = (int)fmin(fmax((p[a] - g.org[a]) * g.inv_edge, 0.0), (double)(g.dim[a] - 1));
    }
    cc[0] = c[g.s[0]]; cc[1] = c[g.s[1]]; cc[2] = c[g.s[2]];
    return ((uint32_t)cc[2] * (uint32_t)g.dim[g.s[1]] + (uint32_t)cc[1]) * (uint32_t)g.dim[g.s[0]] + (uint32_t)cc[0];
}

// replaced ghosts (slot < dead_below, original id >= n_owned) get the key ncells: they sort behind every cell
__global__ __launch_bounds__(256) void cell_keys(GridDesc g, const double *__restrict__ x, const double *__restrict__ y,
                                                 const double *__restrict__ z, int64_t n, uint32_t *__restrict__ keys,
                                                 uint32_t *__restrict__ vals, const int32_t *__restrict__ orig,
                                                 int32_t n_owned, int64_t dead_below) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    int cc[3];
    const bool dead = i < dead_below && orig[i] >= n_owned;
    keys[i] = dead ? (uint32_t)g.ncells : cell_key(g, x[i], y[i], z[i], cc);
    vals[i] = (uint32_t)i;
}

// ---- counting sort by cell (the default; the radix sort below remains for grids with far more cells than particles) ----
// The sorted order wanted is that of a STABLE sort of (cell key, slot): cells ascending, within a cell the slots in their
// previous order.  keys + histogram -> exclusive scan = cell table -> scatter with a per-cell cursor (arrival order, not
// reproducible) -> every entry is moved to its rank among the entries of its cell (reproducible again).
// ~70 us at 1e6 particles against ~165 us for rocprim's 21-launch merge sort of the same pairs + the table search.
// The slots are nearly sorted already (last step's order), so the lanes of a wave hold a few runs of equal keys: one atomic
// per run instead of one per lane (42 -> ~10 us for the histogram, 60 -> ~15 us for the scatter at 1e6 particles).
// Returns this lane's offset within its run and, for the run's first lane, the run length (0 for the others).
__device__ __forceinline__ int run_of_equal_keys(uint32_t k, bool valid, int &run_len, int &head_lane) {
    const int lane = threadIdx.x & 63;
    const uint32_t prev = (uint32_t)__shfl_up((int)k, 1, 64);
    const bool pvalid = __shfl_up(valid ? 1 : 0, 1, 64) != 0;
    const bool head = valid && (lane == 0 || !pvalid || prev != k);
    // a run ends where the next head starts or where the valid lanes end
    const uint64_t heads = __ballot(head), ends = heads | ~__ballot(valid);
    const uint64_t below = heads & ((lane == 63) ? ~0ull : ((2ull << lane) - 1ull));          // heads at or below this lane
    head_lane = below ? 63 - __clzll((long long)below) : lane;
    const uint64_t above = lane == 63 ? 0ull : (ends >> (lane + 1));                           // first boundary above this lane
    const int next = above ? lane + 1 + (__ffsll((long long)above) - 1) : 64;
    run_len = head ? next - lane : 0;
    return lane - head_lane;
}

__global__ __launch_bounds__(256) void cell_keys_count(GridDesc g, const double *__restrict__ x, const double *__restrict__ y,
                                                       const double *__restrict__ z, int64_t n, uint32_t *__restrict__ keys,
                                                       int32_t *__restrict__ count, const int32_t *__restrict__ orig,
                                                       int32_t n_owned, int64_t dead_below) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const bool valid = i < n;
    uint32_t k = 0;
    if (valid) {
        int cc[3];
        const bool dead = i < dead_below && orig[i] >= n_owned;
        k = dead ? (uint32_t)g.ncells : cell_key(g, x[i], y[i], z[i], cc);
        keys[i] = k;
    }
    int run_len, head_lane;
    run_of_equal_keys(k, valid, run_len, head_lane);
    if (run_len > 0) atomicAdd(&count[k], run_len);
}

__global__ __launch_bounds__(256) void cell_scatter(const uint32_t *__restrict__ keys, int64_t n, const int32_t *__restrict__ cell_start,
                                                    int32_t *__restrict__ fill, uint32_t *__restrict__ slots) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const bool valid = i < n;
    const uint32_t k = valid ? keys[i] : 0u;
    int run_len, head_lane;
    const int off = run_of_equal_keys(k, valid, run_len, head_lane);
    int base = 0;
    if (run_len > 0) base = cell_start[k] + atomicAdd(&fill[k], run_len);
    base = __shfl(base, head_lane, 64);
    if (valid) slots[base + off] = (uint32_t)i;
}

// p < n_live: entry slots[p] of cell k goes to cell_start[k] + (number of entries of the cell that are smaller)
__global__ __launch_bounds__(256) void cell_rank(const uint32_t *__restrict__ keys, int64_t n_live, const int32_t *__restrict__ cell_start,
                                                 const uint32_t *__restrict__ slots, uint32_t *__restrict__ sorted) {
    int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= n_live) return;
    const uint32_t v = slots[p];
    const uint32_t k = keys[v];
    const int s = cell_start[k], e = cell_start[k + 1];
    int rank = 0;
    for (int q = s; q < e; q++) rank += slots[q] < v ? 1 : 0;
    sorted[s + rank] = v;
}

// cell_start[c] = lower_bound(sorted keys, c), c in [0, ncells]
__global__ __launch_bounds__(256) void cell_table(const uint32_t *__restrict__ keys, int64_t n, int64_t ncells,
                                                  int32_t *__restrict__ cell_start) {
    int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (c > ncells) return;
    int64_t lo = 0, hi = n;
    while (lo < hi) {
        int64_t mid = (lo + hi) >> 1;
        if ((int64_t)keys[mid] < c) lo = mid + 1; else hi = mid;
    }
    cell_start[c] = (int32_t)lo;
}

struct ReorderArgs {
    const double *src[10];
    double *dst[10];
    int nf;            // 9 state fields, 10 with the smoothing length (variable-h path)
    double *prec;      // variable-h: {x,y,z,h} gather record, else nullptr
};

__global__ __launch_bounds__(256) void reorder(ReorderArgs a, const uint32_t *__restrict__ perm,
                                               const int32_t *__restrict__ orig_in, int32_t *__restrict__ orig_out,
                                               int32_t *__restrict__ inv, double *__restrict__ drec, int64_t n) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint32_t s = perm[i];
    double v[10];
#pragma unroll
    for (int k = 0; k < 10; k++) v[k] = k < a.nf ? a.src[k][s] : 0.0;
#pragma unroll
    for (int k = 0; k < 10; k++)
        if (k < a.nf) a.dst[k][i] = v[k];
    if (a.prec) reinterpret_cast<double4 *>(a.prec)[i] = make_double4(v[SPH_F_X], v[SPH_F_Y], v[SPH_F_Z], v[9]);
    const int32_t id = orig_in[s];
    orig_out[i] = id;
    inv[id] = (int32_t)i;                 // original id -> sorted slot
    double4 r = make_double4(v[SPH_F_X], v[SPH_F_Y], v[SPH_F_Z], v[SPH_F_M]);
    reinterpret_cast<double4 *>(drec)[i] = r;
}

__global__ __launch_bounds__(256) void iota_kernel(int32_t *p, int64_t n) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) p[i] = (int32_t)i;
}

__global__ __launch_bounds__(256) void unpermute(const double *__restrict__ src, const int32_t *__restrict__ orig,
                                                 double *__restrict__ dst, int64_t n) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) dst[orig[i]] = src[i];
}

__global__ __launch_bounds__(256) void fill_kernel(double *p, double v, int64_t n) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) p[i] = v;
}

// dst[slot] = vals[orig[slot] - first] for the slots whose original id lies in [first, first+count)
__global__ __launch_bounds__(256) void scatter_by_id(double *__restrict__ dst, const int32_t *__restrict__ orig,
                                                     const double *__restrict__ vals, int64_t n, int64_t first, int64_t count) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int64_t id = (int64_t)orig[i] - first;
    if (id >= 0 && id < count) dst[i] = vals[id];
}

struct FieldPtrs {
    double *p[SPH_F_COUNT];
    int32_t nf;
};

// out[f][k] = field_f[slot of original id ids[k]]   (ids == nullptr: ids[k] = k)
__global__ __launch_bounds__(256) void gather_by_id(FieldPtrs fp, const int32_t *__restrict__ inv,
                                                    const int64_t *__restrict__ ids, int64_t count,
                                                    double *__restrict__ out) {
    int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (k >= count) return;
    const int32_t slot = inv[ids ? ids[k] : k];
    for (int f = 0; f < fp.nf; f++) out[(size_t)f * count + k] = fp.p[f][slot];
}

// the same for a selection whose size only the device knows (sph_select_boxes_async): out[0] = count, out[1] = 0, then
// out[2 + f * count + k] for k < count -- if count <= capacity; else the header alone (the caller asks again, exactly)
__global__ __launch_bounds__(256) void gather_selected(FieldPtrs fp, const int32_t *__restrict__ inv, const int64_t *__restrict__ ids,
                                                       const int64_t *__restrict__ count_ptr, int64_t capacity, double *__restrict__ out) {
    const int64_t count = *count_ptr;
    const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (k == 0) { out[0] = (double)count; out[1] = 0.0; }
    if (k >= count || count > capacity) return;
    const int32_t slot = inv[ids[k]];
    for (int f = 0; f < fp.nf; f++) out[2 + (size_t)f * count + k] = fp.p[f][slot];
}

// field_f[slot of original id first+k] = vals[f][k]
__global__ __launch_bounds__(256) void scatter_many_by_id(FieldPtrs fp, const int32_t *__restrict__ inv, int64_t first,
                                                          int64_t count, const double *__restrict__ vals) {
    int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (k >= count) return;
    const int32_t slot = inv[first + k];
    for (int f = 0; f < fp.nf; f++) fp.p[f][slot] = vals[(size_t)f * count + k];
}

}  // namespace

hipError_t launch_gather_fields(sph_ctx *c, int nf, const int *fields, const int64_t *ids, int64_t count, double *out) {
    if (count <= 0 || nf <= 0) return hipSuccess;
    FieldPtrs fp{};
    fp.nf = nf;
    for (int f = 0; f < nf; f++) fp.p[f] = c->f[fields[f]];
    gather_by_id<<<dim3((unsigned)((count + 255) / 256)), dim3(256), 0, c->stream>>>(fp, c->inv, ids, count, out);
    return hipGetLastError();
}

hipError_t launch_gather_selected(sph_ctx *c, int nf, const int *fields, int box, int64_t capacity, double *out) {
    FieldPtrs fp{};
    fp.nf = nf;
    for (int f = 0; f < nf; f++) fp.p[f] = c->f[fields[f]];
    const int64_t threads = std::max<int64_t>(capacity, 1);
    gather_selected<<<dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, c->stream>>>(fp, c->inv, c->sel_ids + (size_t)box * c->sel_stride,
                                                                                        c->sel_count + box, capacity, out);
    return hipGetLastError();
}

hipError_t launch_scatter_fields(sph_ctx *c, int nf, const int *fields, int64_t first, int64_t count, const double *vals) {
    if (count <= 0 || nf <= 0) return hipSuccess;
    FieldPtrs fp{};
    fp.nf = nf;
    for (int f = 0; f < nf; f++) fp.p[f] = c->f[fields[f]];
    scatter_many_by_id<<<dim3((unsigned)((count + 255) / 256)), dim3(256), 0, c->stream>>>(fp, c->inv, first, count, vals);
    return hipGetLastError();
}

hipError_t launch_fill(sph_ctx *c, double *p, double v, int64_t n) {
    if (n <= 0) return hipSuccess;
    fill_kernel<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream>>>(p, v, n);
    return hipGetLastError();
}

hipError_t launch_scatter_field(sph_ctx *c, double *field, int64_t first, int64_t count, const double *vals) {
    if (c->n <= 0 || count <= 0) return hipSuccess;
    scatter_by_id<<<dim3((unsigned)((c->n + 255) / 256)), dim3(256), 0, c->stream>>>(field, c->orig, vals, c->n, first, count);
    return hipGetLastError();
}

hipError_t grid_sort_tmp_bytes(int64_t n, size_t *bytes) {
    size_t b = 0;
    hipError_t e = rocprim::radix_sort_pairs(nullptr, b, (uint32_t *)nullptr, (uint32_t *)nullptr, (uint32_t *)nullptr,
                                             (uint32_t *)nullptr, (size_t)n, 0u, 32u, (hipStream_t) nullptr);
    *bytes = b;
    return e;
}

hipError_t launch_iota(sph_ctx *c, int32_t *p, int64_t n) {
    if (n <= 0) return hipSuccess;
    iota_kernel<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream>>>(p, n);
    return hipGetLastError();
}

hipError_t launch_unpermute(sph_ctx *c, const double *src_sorted, double *dst_original) {
    if (c->n <= 0) return hipSuccess;
    unpermute<<<dim3((unsigned)((c->n + 255) / 256)), dim3(256), 0, c->stream>>>(src_sorted, c->orig, dst_original, c->n);
    return hipGetLastError();
}

// device-side cell key for the pair kernels lives in pairs.hip (same formula)

#define GR_CHECK(expr)                                                      \
    do {                                                                    \
        hipError_t _e = (expr);                                             \
        if (_e != hipSuccess) {                                             \
            c->err = std::string(#expr) + ": " + hipGetErrorString(_e);     \
            return SPH_ERR_HIP;                                             \
        }                                                                   \
    } while (0)

// bounding box of the owned particles at their current positions -> d_out6 (device) and/or h_out6 (host, synchronises)
int owned_bbox(sph_ctx *c, double *d_out6, double *h_out6) {
    hipStream_t st = c->stream;
    const int64_t ns = c->n_slots;
    int nb = (int)std::min<int64_t>((ns + BB_BLOCK - 1) / BB_BLOCK, BB_MAX_BLOCKS);
    if (nb < 1) nb = 1;
    double *res = c->bbox_part + (size_t)BB_MAX_BLOCKS * 6 + 8;
    bbox_partial<<<dim3(nb), dim3(BB_BLOCK), 0, st>>>(c->f[SPH_F_X], c->f[SPH_F_Y], c->f[SPH_F_Z], ns, c->bbox_part, c->d_flags + 2,
                                                      c->orig, (int32_t)c->n_owned, 0, 1);
    bbox_final<<<dim3(1), dim3(384), 0, st>>>(c->bbox_part, nb, res);
    GR_CHECK(hipGetLastError());
    if (d_out6) GR_CHECK(hipMemcpyAsync(d_out6, res, 6 * sizeof(double), hipMemcpyDeviceToDevice, st));
    if (h_out6) {
        GR_CHECK(hipMemcpyAsync(c->h_pinned + 48, res, 6 * sizeof(double), hipMemcpyDeviceToHost, st));
        GR_CHECK(hipStreamSynchronize(st));
        for (int a = 0; a < 6; a++) h_out6[a] = c->h_pinned[48 + a];
    }
    return SPH_OK;
}

int grid_rebuild(sph_ctx *c) {
    const int64_t n = c->n;                 // live particles after the build
    const int64_t ns = c->n_slots;          // occupied slots before it (> n while a ghost swap is pending)
    const bool swap = c->dead_below > 0;
    hipStream_t st = c->stream;
    if (ns == 0) { c->grid_valid = true; return SPH_OK; }

    // ---- bounding box ---------------------------------------------------------------
    int nb = (int)std::min<int64_t>((ns + BB_BLOCK - 1) / BB_BLOCK, BB_MAX_BLOCKS);
    // d_flags[0] (non-finite position seen) is zero here: cleared at creation and by every bbox_final
    bbox_partial<<<dim3(nb), dim3(BB_BLOCK), 0, st>>>(c->f[SPH_F_X], c->f[SPH_F_Y], c->f[SPH_F_Z], ns, c->bbox_part, c->d_flags,
                                                      c->orig, (int32_t)c->n_owned, c->dead_below, 0);
    const int p = c->ring_bbox;
    double *slot = c->h_pinned + 200 + 16 * p;
    bbox_final<<<dim3(1), dim3(384), 0, st>>>(c->bbox_part, nb, c->bbox_part + (size_t)BB_MAX_BLOCKS * 6, slot, c->d_flags);
    GR_CHECK(hipGetLastError());
    // the exact box of the current positions -> read-back slot p.  Who needs it NOW (octree root boxes: variable h,
    // self-gravity, accretion; the first build of a particle set) waits for it; the plain fixed-h path takes the box of the
    // previous build, which arrived long ago, widened by one cell: particles outside the grid's box are clamped into its
    // boundary cells and still meet all their neighbours there, so the box only has to be roughly right.
    GR_CHECK(hipEventRecord(c->ev_bbox[p], st));        // bbox_final wrote the slot itself (pinned memory)
    const bool stale = c->ring_bbox_valid && !c->no_stale && !c->variable && !c->gravity && !(c->p.flags & SPH_FLAG_ACCRETE_CULL);
    const double *bb = slot;
    if (stale) {
        GR_CHECK(hipEventSynchronize(c->ev_bbox[1 - p]));
        bb = c->h_pinned + 200 + 16 * (1 - p);
    } else {
        GR_CHECK(hipStreamSynchronize(st));
        c->host_syncs++;
    }
    c->ring_bbox = 1 - p; c->ring_bbox_valid = true; c->bbox_exact = !stale;
    if (*reinterpret_cast<const int32_t *>(bb + 6) != 0) {
        c->err = "non-finite particle position at grid build";
        return SPH_ERR_NONFINITE;
    }
    const double guard = stale ? 2.0 * c->p.h * (1.0 + 1e-6) : 0.0;
    for (int a = 0; a < 3; a++) { c->bbox[a] = bb[a] - guard; c->bbox[3 + a] = bb[3 + a] + guard; }
    bb = c->bbox;
    GridDesc g{};
    // fixed h: cells of edge 2h.  variable h: edge 2 <h> with a per-cell maximum of h (varh.hip)
    double edge = 2.0 * (c->variable ? c->h_mean : c->p.h) * (1.0 + 1e-6);
    if (c->variable) {
        // keep the cell table below ~2^27 cells: widen the cells if the box is huge compared with <h>
        double vol = 1.0;
        for (int a = 0; a < 3; a++) vol *= std::floor((bb[3 + a] - bb[a]) / edge) + 1.0;
        if (vol > 134217728.0) edge *= std::cbrt(vol / 134217728.0);
    }
    g.inv_edge = 1.0 / edge;
    // Very sparse domains (a particle that escaped to 1e5 AU, a diffuse halo): the cell table must not grow with the
    // VOLUME of the bounding box.  The grid's box need not hold every particle -- a particle outside it is clamped into
    // a boundary cell, where it still meets all its neighbours (cells are >= 2h wide, so everything within 2h of an
    // outside particle is clamped to the same layer or sits in the last one) -- so when the exact box would need more
    // than ~64 cells per particle the grid covers the bulk only: mean +- 6 sigma of the particles inside the current box,
    // trimmed repeatedly (a far outlier inflates sigma, the next round no longer sees it).  Results do not depend on the
    // box beyond summation order; only the boundary cells get crowded if MANY particles lie outside.
    double tb[6] = {bb[0], bb[1], bb[2], bb[3], bb[4], bb[5]};
    auto cells_of = [&](const double *b) {
        double v = 1.0;
        for (int a = 0; a < 3; a++) v *= std::floor((b[3 + a] - b[a]) * g.inv_edge) + 1.0;
        return v;
    };
    const double sparse_limit = 64.0 * (double)ns + 4.0e6;
    for (int round = 0; round < 8 && cells_of(tb) > sparse_limit; round++) {
        Box6 bx;
        for (int a = 0; a < 3; a++) { bx.lo[a] = tb[a]; bx.hi[a] = tb[3 + a]; }
        double *mpart = c->bbox_part;                                   // >= 1024 * 7 doubles
        moment_partial<<<dim3(nb), dim3(BB_BLOCK), 0, st>>>(c->f[SPH_F_X], c->f[SPH_F_Y], c->f[SPH_F_Z], ns, bx, mpart, c->orig,
                                                            (int32_t)c->n_owned, c->dead_below);
        moment_final<<<dim3(1), dim3(448), 0, st>>>(mpart, nb, mpart + (size_t)BB_MAX_BLOCKS * 7);
        GR_CHECK(hipGetLastError());
        GR_CHECK(hipMemcpyAsync(c->h_pinned + 280, mpart + (size_t)BB_MAX_BLOCKS * 7, 7 * sizeof(double), hipMemcpyDeviceToHost, st));
        GR_CHECK(hipStreamSynchronize(st));
        c->host_syncs++;
        const double *mo = c->h_pinned + 280;
        if (!(mo[0] >= 1.0)) break;
        bool shrunk = false;
        for (int a = 0; a < 3; a++) {
            const double mean = mo[1 + a] / mo[0];
            const double sig = std::sqrt(std::max(mo[4 + a] / mo[0] - mean * mean, 0.0));
            const double half = 6.0 * sig + 2.0 * edge;
            const double lo = std::max(tb[a], mean - half), hi = std::min(tb[3 + a], mean + half);
            if (lo > tb[a] || hi < tb[3 + a]) shrunk = true;
            tb[a] = lo; tb[3 + a] = hi;
        }
        if (!shrunk) break;
    }
    bb = tb;
    double ncell_d = 1.0;
    for (int a = 0; a < 3; a++) {
        g.org[a] = bb[a];
        double ext = bb[3 + a] - bb[a];
        double d = std::floor(ext * g.inv_edge) + 1.0;
        if (!(d >= 1.0) || d > 2.0e9) { c->err = "cell grid dimension out of range"; return SPH_ERR_GRID; }
        g.dim[a] = (int32_t)d;
        ncell_d *= d;
    }
    if (ncell_d >= 2147483647.0) { c->err = "cell grid exceeds 2^31 cells"; return SPH_ERR_GRID; }
    g.ncells = (int64_t)g.dim[0] * g.dim[1] * g.dim[2];
    // axis permutation: the axis with the fewest cells runs fastest (a column of cells is short: the thin direction of a
    // disc), the LONGEST of the other two is the middle one.  Rows are then as long as possible, a workgroup of consecutive
    // particles rarely wraps from one row into the next, and its three candidate intervals (tiled.hip) stay a few columns
    // wide -- also in the narrow x-slabs of a multi-GPU run, where the short axis in the middle would make every
    // interval span whole rows
    int s[3] = {0, 1, 2};
    for (int i = 0; i < 3; i++)
        for (int j = i + 1; j < 3; j++)
            if (g.dim[s[j]] < g.dim[s[i]]) std::swap(s[i], s[j]);
    std::swap(s[1], s[2]);
    g.s[0] = s[0]; g.s[1] = s[1]; g.s[2] = s[2];
    c->grid = g;

    if (g.ncells + 2 > c->cell_cap) {
        ctx_free(c, c->cell_start); c->cell_fill = nullptr;
        c->cell_cap = (g.ncells + 2) + (g.ncells + 2) / 4;
        // one allocation: the cell table and, right behind the part in use, the cursors of the counting sort (one fill zeroes both)
        if (ctx_alloc(c, &c->cell_start, 2 * (size_t)c->cell_cap + 16, "cell table + cursors") != SPH_OK) { c->cell_cap = 0; return SPH_ERR_NOMEM; }
        if (c->variable) {
            ctx_free(c, c->cell_hmax);
            if (ctx_alloc(c, &c->cell_hmax, (size_t)c->cell_cap, "cell hmax") != SPH_OK) { c->cell_cap = 0; return SPH_ERR_NOMEM; }
        }
    }

    // ---- keys, sort, cell table ---------------------------------------------------------
    const unsigned gb = (unsigned)((std::max<int64_t>(n, 1) + 255) / 256);
    const unsigned gbs = (unsigned)((ns + 255) / 256);
    static const bool force_radix = getenv("SPH_SORT_RADIX") != nullptr;             // A/B switch
    size_t scan_tmp = 0;
    bool counting = !force_radix && g.ncells <= 4 * ns + 1000000;
    if (counting) {
        GR_CHECK(rocprim::exclusive_scan(nullptr, scan_tmp, c->cell_start, c->cell_start, 0, (size_t)(g.ncells + 2), rocprim::plus<int32_t>(), st));
        counting = scan_tmp <= c->sort_tmp_bytes;
    }
    if (counting) {
        const size_t table = ((size_t)(g.ncells + 2) + 3) & ~(size_t)3, cursors = ((size_t)(g.ncells + 1) + 3) & ~(size_t)3;
        c->cell_fill = c->cell_start + table;
        GR_CHECK(hipMemsetAsync(c->cell_start, 0, sizeof(int32_t) * (table + cursors), st));        // multiples of 16 bytes: one fill kernel
        cell_keys_count<<<dim3(gbs), dim3(256), 0, st>>>(g, c->f[SPH_F_X], c->f[SPH_F_Y], c->f[SPH_F_Z], ns, c->keys, c->cell_start,
                                                        c->orig, (int32_t)c->n_owned, c->dead_below);
        GR_CHECK(hipGetLastError());
        // in place: cell_start[c] = first sorted slot of cell c; [ncells] = live particles; the replaced ghosts sort behind them
        GR_CHECK(rocprim::exclusive_scan(c->sort_tmp, scan_tmp, c->cell_start, c->cell_start, 0, (size_t)(g.ncells + 2), rocprim::plus<int32_t>(), st));
        cell_scatter<<<dim3(gbs), dim3(256), 0, st>>>(c->keys, ns, c->cell_start, c->cell_fill, c->vals);
        cell_rank<<<dim3(gb), dim3(256), 0, st>>>(c->keys, n, c->cell_start, c->vals, c->vals_alt);
        GR_CHECK(hipGetLastError());
        c->n_slots = n; c->dead_below = 0;
    } else {
        cell_keys<<<dim3(gbs), dim3(256), 0, st>>>(g, c->f[SPH_F_X], c->f[SPH_F_Y], c->f[SPH_F_Z], ns, c->keys,
                                                  c->vals, c->orig, (int32_t)c->n_owned, c->dead_below);
        GR_CHECK(hipGetLastError());
        unsigned bits = 1;
        while (bits < 32 && ((int64_t)1 << bits) < g.ncells + (swap ? 1 : 0)) bits++;
        size_t tmp = c->sort_tmp_bytes;
        GR_CHECK(rocprim::radix_sort_pairs(c->sort_tmp, tmp, c->keys, c->keys_alt, c->vals, c->vals_alt, (size_t)ns, 0u, bits, st));
        c->n_slots = n; c->dead_below = 0;     // the replaced ghosts sorted behind the n live entries and are dropped here
        cell_table<<<dim3((unsigned)((g.ncells + 1 + 255) / 256)), dim3(256), 0, st>>>(c->keys_alt, n, g.ncells, c->cell_start);
        GR_CHECK(hipGetLastError());
    }

    // ---- reorder state into sorted slots --------------------------------------------------
    ReorderArgs ra{};
    for (int k = 0; k < 9; k++) { ra.src[k] = c->f[k]; ra.dst[k] = c->f_alt[k]; }
    ra.nf = 9; ra.prec = nullptr;
    if (c->variable) { ra.src[9] = c->f[SPH_F_H]; ra.dst[9] = c->f_alt[9]; ra.nf = 10; ra.prec = c->prec; }
    reorder<<<dim3(gb), dim3(256), 0, st>>>(ra, c->vals_alt, c->orig, c->orig_alt, c->inv, c->drec, n);
    GR_CHECK(hipGetLastError());
    for (int k = 0; k < 9; k++) std::swap(c->f[k], c->f_alt[k]);
    if (c->variable) std::swap(c->f[SPH_F_H], c->f_alt[9]);
    std::swap(c->orig, c->orig_alt);
    c->grid_builds++;
    return SPH_OK;
}

}  // namespace sph
