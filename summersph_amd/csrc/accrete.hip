// accrete.hip -- sink accretion and boundary cull on the device (the particle set shrinks).
//
// Replaces (citations: /root/reference/SUMMER_SPH.f90 "[F]"; "SUMMER_SPH - Variable.f90" "[V]")
//   initiate_sink_accretion + sink2gasdists + pack_sinks   [F]:484-556  ([V]:616-688)
//   check_bounds                                           [F]:471-482  ([V]:599-614, gas part)
//
// The reference decides accretion on its octree: a sink's walk descends while the node centre is within
// radius + edge/2 of the sink on every axis, and at a one-particle leaf it applies ITS distance rule --
// [F]: leaf centre within 2 radius + edge/2, then sum_k sqrt(centre_k^2 - sink_k^2) < radius (a quirk of
//      the reference: it uses the LEAF CENTRE, and a negative argument gives NaN = "not accreted");
// [V]: leaf centre within radius + edge/2, then sum_k |x_k - sink_k| < radius.
// Both need the particle's chain of octree boxes, which is replayed here from the particle's path key
// (same keys as gravity.hip / varh.hip) and its leaf level (longest common prefix with its sorted
// neighbours + 1).  Survivors keep their relative order (the reference's pack()), i.e. the caller's
// particle numbering after the call is "rank among survivors".
#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#include <cmath>

#include "pair_common.hpp"

namespace sph {

namespace {

constexpr int AB = 256;
constexpr int LEVELS = 21;

struct RootBox { double c[3]; double size; };

__global__ __launch_bounds__(AB) void acc_keys(RootBox rb, const double4 *__restrict__ drec, int64_t n,
                                               uint64_t *__restrict__ keys, uint32_t *__restrict__ vals) {
    const int64_t i = (int64_t)blockIdx.x * AB + threadIdx.x;
    if (i >= n) return;
    const double4 p = drec[i];
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

// initiate_sink_accretion runs only if a sink has mass ([F]:919): decided on the device (the masses live there), so that the
// pass needs no read-back for it.  Accretion only adds mass, so the answer is the same before and after the updates of a pass
// in which it was false (nothing is marked then).
__device__ __forceinline__ bool any_sink_mass(const double *__restrict__ sink, int ns) {
    bool any = false;
    for (int k = 0; k < ns; k++) any |= sink[6 * MAX_SINKS + k] > 0.0;
    return any;
}

__device__ __forceinline__ int common_levels(uint64_t a, uint64_t b) {
    const uint64_t x = a ^ b;
    if (x == 0) return LEVELS;
    return (__clzll((long long)x) - 1) / 3;
}

// keep[id] = 1 unless accreted by some sink or outside the box.  acc[slot] = bit mask of accreting sinks.
__global__ __launch_bounds__(AB) void acc_mark(RootBox rb, const uint64_t *__restrict__ keys, const uint32_t *__restrict__ vals,
                                               int64_t n, const double4 *__restrict__ drec, const int32_t *__restrict__ orig,
                                               const double *__restrict__ sink, const double *__restrict__ srad, int ns, int variant,
                                               double bound, int32_t *__restrict__ keep,
                                               unsigned long long *__restrict__ accmask) {
    const int64_t s = (int64_t)blockIdx.x * AB + threadIdx.x;
    if (s >= n) return;
    const bool do_accrete = any_sink_mass(sink, ns);
    const uint64_t key = keys[s];
    int cp = 0;
    if (s > 0) cp = max(cp, common_levels(key, keys[s - 1]));
    if (s + 1 < n) cp = max(cp, common_levels(key, keys[s + 1]));
    const int level = n == 1 ? 0 : min(cp + 1, LEVELS);
    const uint32_t slot = vals[s];
    const double4 p = drec[slot];
    unsigned long long mask = 0ull;
    if (do_accrete) {
        for (int k = 0; k < ns; k++) {
            const double sx = sink[0 * MAX_SINKS + k], sy = sink[1 * MAX_SINKS + k], sz = sink[2 * MAX_SINKS + k];
            const double rad = srad[k];
            double cx = rb.c[0], cy = rb.c[1], cz = rb.c[2], size = rb.size;
            bool reached = true;
            for (int l = 1; l <= level; l++) {          // ancestors: levels 0 .. level-1 hold more than one particle
                const double lim = rad + size / 2.0;                                         // [F]:529
                if (!(fabs(cx - sx) < lim && fabs(cy - sy) < lim && fabs(cz - sz) < lim)) { reached = false; break; }
                const int ch = (int)((key >> (3 * (LEVELS - l))) & 7);
                const double q = 0.25 * size;
                cx = cx + ((ch & 1) ? q : -q); cy = cy + ((ch & 2) ? q : -q); cz = cz + ((ch & 4) ? q : -q);
                size = size * 0.5;
            }
            if (!reached) continue;
            const double lim = (variant ? rad : 2 * rad) + size / 2.0;                       // [F]:536 / [V]:668
            if (!(fabs(cx - sx) < lim && fabs(cy - sy) < lim && fabs(cz - sz) < lim)) continue;
            double dr;
            if (variant) dr = sqrt((p.x - sx) * (p.x - sx)) + sqrt((p.y - sy) * (p.y - sy)) + sqrt((p.z - sz) * (p.z - sz));   // [V]:669
            else dr = sqrt(cx * cx - sx * sx) + sqrt(cy * cy - sy * sy) + sqrt(cz * cz - sz * sz);                             // [F]:537
            if (dr < rad) mask |= 1ull << k;
        }
    }
    accmask[slot] = mask;
    const bool inside = fabs(p.x) <= bound && fabs(p.y) <= bound && fabs(p.z) <= bound;       // [F]:478
    keep[orig[slot]] = (mask == 0ull && inside) ? 1 : 0;
}

// Multi-GPU: the octree is that of ALL GPUs' particles (the external gravity sources: keys / vals are their sorted
// path keys and source indices).  Every GPU looks at all leaves but only marks its own particles: source indices
// [src_off, src_off + n_owned) are this GPU's owned particles in the caller's order.  keep[] / accmask[] were zeroed
// (ghosts are dropped with the accreted particles; they are exchanged again before the next evaluation).
__global__ __launch_bounds__(AB) void acc_mark_ext(RootBox rb, const uint64_t *__restrict__ keys, const uint32_t *__restrict__ vals,
                                                   int64_t n_glob, int64_t src_off, int64_t n_owned, const double4 *__restrict__ drec,
                                                   const int32_t *__restrict__ inv, const double *__restrict__ sink,
                                                   const double *__restrict__ srad, int ns, int variant, double bound,
                                                   int32_t *__restrict__ keep, unsigned long long *__restrict__ accmask) {
    const int64_t s = (int64_t)blockIdx.x * AB + threadIdx.x;
    if (s >= n_glob) return;
    const bool do_accrete = any_sink_mass(sink, ns);
    const int64_t id = (int64_t)vals[s] - src_off;
    if (id < 0 || id >= n_owned) return;
    const uint64_t key = keys[s];
    int cp = 0;
    if (s > 0) cp = max(cp, common_levels(key, keys[s - 1]));
    if (s + 1 < n_glob) cp = max(cp, common_levels(key, keys[s + 1]));
    const int level = n_glob == 1 ? 0 : min(cp + 1, LEVELS);
    const int32_t slot = inv[id];
    const double4 p = drec[slot];
    unsigned long long mask = 0ull;
    if (do_accrete) {
        for (int k = 0; k < ns; k++) {
            const double sx = sink[0 * MAX_SINKS + k], sy = sink[1 * MAX_SINKS + k], sz = sink[2 * MAX_SINKS + k];
            const double rad = srad[k];
            double cx = rb.c[0], cy = rb.c[1], cz = rb.c[2], size = rb.size;
            bool reached = true;
            for (int l = 1; l <= level; l++) {
                const double lim = rad + size / 2.0;                                         // [F]:529
                if (!(fabs(cx - sx) < lim && fabs(cy - sy) < lim && fabs(cz - sz) < lim)) { reached = false; break; }
                const int ch = (int)((key >> (3 * (LEVELS - l))) & 7);
                const double q = 0.25 * size;
                cx = cx + ((ch & 1) ? q : -q); cy = cy + ((ch & 2) ? q : -q); cz = cz + ((ch & 4) ? q : -q);
                size = size * 0.5;
            }
            if (!reached) continue;
            const double lim = (variant ? rad : 2 * rad) + size / 2.0;                       // [F]:536 / [V]:668
            if (!(fabs(cx - sx) < lim && fabs(cy - sy) < lim && fabs(cz - sz) < lim)) continue;
            double dr;
            if (variant) dr = sqrt((p.x - sx) * (p.x - sx)) + sqrt((p.y - sy) * (p.y - sy)) + sqrt((p.z - sz) * (p.z - sz));   // [V]:669
            else dr = sqrt(cx * cx - sx * sx) + sqrt(cy * cy - sy * sy) + sqrt(cz * cz - sz * sz);                             // [F]:537
            if (dr < rad) mask |= 1ull << k;
        }
    }
    accmask[slot] = mask;
    const bool inside = fabs(p.x) <= bound && fabs(p.y) <= bound && fabs(p.z) <= bound;       // [F]:478
    keep[id] = (mask == 0ull && inside) ? 1 : 0;
}

// sink k: sums over ranks (rank order) of the per-rank sums, then [F]:497-508
__global__ void acc_sink_update_ranks(int ns, const double *__restrict__ all, int nranks, int stride, double *__restrict__ sink) {
    __shared__ int s_any;
    if (threadIdx.x == 0) s_any = any_sink_mass(sink, ns) ? 1 : 0;       // as of before any update of this pass
    __syncthreads();
    const int k = threadIdx.x;
    if (k >= ns || !s_any) return;
    double v[7] = {0, 0, 0, 0, 0, 0, 0};
    for (int r = 0; r < nranks; r++)
        for (int q = 0; q < 7; q++) v[q] += all[(size_t)r * stride + (size_t)k * 7 + q];
    const double m0 = sink[6 * MAX_SINKS + k];
    const double nm = m0 + v[0];
    for (int a = 0; a < 3; a++) {
        sink[a * MAX_SINKS + k] = (m0 * sink[a * MAX_SINKS + k] + v[1 + a]) / nm;
        sink[(3 + a) * MAX_SINKS + k] = (m0 * sink[(3 + a) * MAX_SINKS + k] + v[4 + a]) / nm;
    }
    sink[6 * MAX_SINKS + k] = m0 + v[0];
}

// one block per sink: the fixed-order total of the per-block partial sums -> out[k*7 .. k*7+7)
__global__ void acc_partials_final(const double *__restrict__ part, int nb, double *__restrict__ out) {
    if (threadIdx.x >= 7) return;
    double r = 0.0;
    for (int b = 0; b < nb; b++) r += part[(size_t)b * 7 + threadIdx.x];
    out[threadIdx.x] = r;
}

__device__ __forceinline__ double wave_sumd(double v) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// per sink: sum of m, m x, m v over its accreted particles (two stages, fixed order)
__global__ __launch_bounds__(AB) void acc_sums_partial(int64_t n, int k, const unsigned long long *__restrict__ accmask,
                                                       const double4 *__restrict__ drec, const double *__restrict__ vx,
                                                       const double *__restrict__ vy, const double *__restrict__ vz,
                                                       double *__restrict__ part) {
    __shared__ double sm[7][AB / WAVE];
    double v[7] = {0, 0, 0, 0, 0, 0, 0};
    for (int64_t i = (int64_t)blockIdx.x * AB + threadIdx.x; i < n; i += (int64_t)gridDim.x * AB) {
        if ((accmask[i] >> k) & 1ull) {
            const double4 p = drec[i];
            v[0] += p.w; v[1] += p.w * p.x; v[2] += p.w * p.y; v[3] += p.w * p.z;
            v[4] += p.w * vx[i]; v[5] += p.w * vy[i]; v[6] += p.w * vz[i];
        }
    }
    for (int q = 0; q < 7; q++) {
        const double r = wave_sumd(v[q]);
        if ((threadIdx.x & 63) == 0) sm[q][threadIdx.x >> 6] = r;
    }
    __syncthreads();
    if (threadIdx.x < 7) {
        double r = 0.0;
        for (int w = 0; w < AB / WAVE; w++) r += sm[threadIdx.x][w];
        part[(size_t)blockIdx.x * 7 + threadIdx.x] = r;
    }
}

// [F]:497-508: new_mass, position and velocity become mass-weighted means, mass grows
__global__ void acc_sink_update(int k, int ns, const double *__restrict__ part, int nb, double *__restrict__ sink) {
    __shared__ double v[7];
    if (threadIdx.x < 7) {                       // one lane per sum, each over the blocks in the same fixed order as ever
        double r = 0.0;
        for (int b = 0; b < nb; b++) r += part[(size_t)b * 7 + threadIdx.x];
        v[threadIdx.x] = r;
    }
    __syncthreads();
    if (threadIdx.x != 0 || !any_sink_mass(sink, ns)) return;
    const double m0 = sink[6 * MAX_SINKS + k];
    const double nm = m0 + v[0];
    for (int a = 0; a < 3; a++) {
        sink[a * MAX_SINKS + k] = (m0 * sink[a * MAX_SINKS + k] + v[1 + a]) / nm;
        sink[(3 + a) * MAX_SINKS + k] = (m0 * sink[(3 + a) * MAX_SINKS + k] + v[4 + a]) / nm;
    }
    sink[6 * MAX_SINKS + k] = m0 + v[0];
}

struct CompactArgs {
    const double *src[10];
    double *dst[10];
    int nf;
};

// survivors, in the caller's order: new id = number of survivors before it
__global__ __launch_bounds__(AB) void acc_compact(CompactArgs a, const int32_t *__restrict__ keep, const int32_t *__restrict__ pos,
                                                  const int32_t *__restrict__ inv, int64_t n) {
    const int64_t id = (int64_t)blockIdx.x * AB + threadIdx.x;
    if (id >= n || !keep[id]) return;
    const int32_t slot = inv[id], o = pos[id];
    for (int f = 0; f < a.nf; f++) a.dst[f][o] = a.src[f][slot];
}

}  // namespace

#define AC_CHECK(expr)                                                      \
    do {                                                                    \
        hipError_t _e = (expr);                                             \
        if (_e != hipSuccess) {                                             \
            c->err = std::string(#expr) + ": " + hipGetErrorString(_e);     \
            return SPH_ERR_HIP;                                             \
        }                                                                   \
    } while (0)

__global__ void sink_cull(int ns, double bound, double *__restrict__ sink, double *__restrict__ srad, int32_t *__restrict__ ns_out);
int sinks_cull(sph_ctx *c);

// requires a valid grid (bbox, drec, orig/inv of the current positions).  On return the context holds only
// the survivors, in the caller's order (like a fresh upload); *removed = how many particles left.
int accrete_and_cull(sph_ctx *c, int64_t *removed, int32_t *d_keep_out) {
    const int64_t n = c->n;
    *removed = 0;
    if (n == 0) return sinks_cull(c);
    RootBox rb;
    double size = 0.0;
    for (int a = 0; a < 3; a++) {
        rb.c[a] = (c->bbox[3 + a] + c->bbox[a]) / 2.0;
        size = std::max(size, c->bbox[3 + a] - c->bbox[a]);
    }
    rb.size = size;
    const unsigned gb = (unsigned)((n + AB - 1) / AB);
    const double4 *drec = reinterpret_cast<const double4 *>(c->drec);
    // (whether accretion runs at all -- a sink with mass, [F]:919 -- is decided on the device: any_sink_mass)

    // the sorted path keys of the current grid build: left by the leaf-box build (variable h), copied from the self-gravity
    // tree of the last evaluation (same positions, same root box, same key), or computed and sorted here (0.2 ms at 1e6)
    if (c->path_keys_valid) {
    } else if (c->gravity && c->tree_valid && !c->gx_src && c->g_keys_alt && c->g_vals_alt) {
        AC_CHECK(hipMemcpyAsync(c->mkeys_alt, c->g_keys_alt, (size_t)n * sizeof(uint64_t), hipMemcpyDeviceToDevice, c->stream));
        AC_CHECK(hipMemcpyAsync(c->mvals_alt, c->g_vals_alt, (size_t)n * sizeof(uint32_t), hipMemcpyDeviceToDevice, c->stream));
        c->path_keys_valid = true;
    } else {
        acc_keys<<<dim3(gb), dim3(AB), 0, c->stream>>>(rb, drec, n, c->mkeys, c->mvals);
        size_t tmp = c->msort_tmp_bytes;
        AC_CHECK(rocprim::radix_sort_pairs(c->msort_tmp, tmp, c->mkeys, c->mkeys_alt, c->mvals, c->mvals_alt, (size_t)n, 0u, 63u, c->stream));
        c->path_keys_valid = true;
    }
    int32_t *keep = reinterpret_cast<int32_t *>(c->keys);          // the cell-key buffers are free between grid builds
    int32_t *pos = reinterpret_cast<int32_t *>(c->keys_alt);
    unsigned long long *accmask = reinterpret_cast<unsigned long long *>(c->scratch);
    acc_mark<<<dim3(gb), dim3(AB), 0, c->stream>>>(rb, c->mkeys_alt, c->mvals_alt, n, drec, c->orig, c->sink, c->sink_radius, c->ns,
                                                   c->variable ? 1 : 0, c->p.bounding_size, keep, accmask);
    AC_CHECK(hipGetLastError());
    if (d_keep_out) AC_CHECK(hipMemcpyAsync(d_keep_out, keep, sizeof(int32_t) * (size_t)n, hipMemcpyDeviceToDevice, c->stream));
    {
        const int nb = (int)std::min<int64_t>((n + AB - 1) / AB, 256);
        for (int k = 0; k < c->ns; k++) {
            acc_sums_partial<<<dim3(nb), dim3(AB), 0, c->stream>>>(n, k, accmask, drec, c->f[SPH_F_VX], c->f[SPH_F_VY], c->f[SPH_F_VZ], c->sink_part);
            acc_sink_update<<<dim3(1), dim3(64), 0, c->stream>>>(k, c->ns, c->sink_part, nb, c->sink);      // no-op without a massive sink
        }
        AC_CHECK(hipGetLastError());
    }
    // exclusive scan of keep[] over original ids -> new ids
    size_t sb = 0;
    AC_CHECK(rocprim::exclusive_scan(nullptr, sb, keep, pos, 0, (size_t)n, rocprim::plus<int32_t>(), c->stream));
    if (sb > c->sort_tmp_bytes) { c->err = "accrete: scan scratch too small"; return SPH_ERR_NOMEM; }
    AC_CHECK(rocprim::exclusive_scan(c->sort_tmp, sb, keep, pos, 0, (size_t)n, rocprim::plus<int32_t>(), c->stream));
    // ONE read-back per call: how many particles stay (and, variable h, how many sinks: [V]'s check_bounds culls sinks too)
    if (c->variable && c->ns > 0) {
        sink_cull<<<dim3(1), dim3(64), 0, c->stream>>>(c->ns, c->p.bounding_size, c->sink, c->sink_radius, c->d_flags + 2);
        AC_CHECK(hipGetLastError());
    }
    int32_t *last = reinterpret_cast<int32_t *>(c->h_pinned + 300);
    last[2] = c->ns;
    AC_CHECK(hipMemcpyAsync(&last[0], pos + (n - 1), sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    AC_CHECK(hipMemcpyAsync(&last[1], keep + (n - 1), sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    if (c->variable && c->ns > 0) AC_CHECK(hipMemcpyAsync(&last[2], c->d_flags + 2, sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    AC_CHECK(hipStreamSynchronize(c->stream));
    c->host_syncs++;
    if (last[2] != c->ns) { c->ns = last[2]; c->rates_valid = false; }
    const int64_t n_new = (int64_t)last[0] + last[1];
    if (n_new == n) return SPH_OK;                     // nobody left: sorted state stays as it is
    CompactArgs ca{};
    for (int k = 0; k < 9; k++) { ca.src[k] = c->f[k]; ca.dst[k] = c->f_alt[k]; }
    ca.nf = 9;
    if (c->variable) { ca.src[9] = c->f[SPH_F_H]; ca.dst[9] = c->f_alt[9]; ca.nf = 10; }
    acc_compact<<<dim3(gb), dim3(AB), 0, c->stream>>>(ca, keep, pos, c->inv, n);
    AC_CHECK(hipGetLastError());
    for (int k = 0; k < 9; k++) std::swap(c->f[k], c->f_alt[k]);
    if (c->variable) std::swap(c->f[SPH_F_H], c->f_alt[9]);
    // the reference's pack ([F]:481,554) keeps the survivors' density, pressure, accelerations and rates of the last
    // evaluation: compact those too (one field at a time through the scratch array, whose mask use is over), so that a
    // download after the last step of a run sees them
    const bool keep_derived = c->rates_valid;
    if (keep_derived) {
        for (int k = SPH_F_RHO; k < SPH_F_COUNT; k++) {
            if ((k == SPH_F_H) || (k == SPH_F_OMEGA && !c->variable)) continue;
            CompactArgs cd{};
            cd.src[0] = c->f[k]; cd.dst[0] = c->scratch; cd.nf = 1;
            acc_compact<<<dim3(gb), dim3(AB), 0, c->stream>>>(cd, keep, pos, c->inv, n);
            std::swap(c->f[k], c->scratch);
        }
        AC_CHECK(hipGetLastError());
    }
    c->n = n_new; c->n_slots = n_new; c->dead_below = 0;
    c->n_owned = n_new;
    AC_CHECK(launch_iota(c, c->orig, n_new));
    AC_CHECK(launch_iota(c, c->inv, n_new));
    c->grid_valid = c->rho_valid = c->eos_valid = c->rates_valid = c->tree_valid = c->order_valid = false;
    c->h_refresh_ok = false;
    c->path_keys_valid = false;
    c->derived_kept = keep_derived;
    *removed = n - n_new;
    return SPH_OK;
}

// ---- check_bounds for the sinks, [V]:610-613: sinks outside the box are packed away (variable-h variant only) ----------
__global__ void sink_cull(int ns, double bound, double *__restrict__ sink, double *__restrict__ srad, int32_t *__restrict__ ns_out) {
    if (threadIdx.x != 0) return;
    int o = 0;
    for (int j = 0; j < ns; j++) {
        const bool inside = fabs(sink[0 * MAX_SINKS + j]) <= bound && fabs(sink[1 * MAX_SINKS + j]) <= bound &&
                            fabs(sink[2 * MAX_SINKS + j]) <= bound;
        if (!inside) continue;
        if (o != j) {
            for (int r = 0; r < 10; r++) sink[r * MAX_SINKS + o] = sink[r * MAX_SINKS + j];
            srad[o] = srad[j];
        }
        o++;
    }
    *ns_out = o;
}

int sinks_cull(sph_ctx *c) {
    if (!c->variable || c->ns == 0) return SPH_OK;
    sink_cull<<<dim3(1), dim3(64), 0, c->stream>>>(c->ns, c->p.bounding_size, c->sink, c->sink_radius, c->d_flags + 2);
    AC_CHECK(hipGetLastError());
    int32_t ns_new = c->ns;
    AC_CHECK(hipMemcpyAsync(&ns_new, c->d_flags + 2, sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    AC_CHECK(hipStreamSynchronize(c->stream));
    if (ns_new != c->ns) { c->ns = ns_new; c->rates_valid = false; }
    return SPH_OK;
}

// ---- check_sink_creation, [V]:549-597 -------------------------------------------------------------------------------
// The FIRST particle (caller's order) with m (eta/h)^3 > 0.5 decides: if it lies within radius_j + 2h of an existing
// sink nothing happens at all (the reference returns), else a sink of mass 1e-11 and radius 2h is created at its
// position with its velocity (the particle stays).  At most one sink per call.
__global__ __launch_bounds__(AB) void sink_create_scan(const double4 *__restrict__ drec, const double *__restrict__ h,
                                                       const int32_t *__restrict__ orig, const int32_t *__restrict__ number,
                                                       int64_t n, int32_t n_owned, double eta, int32_t *__restrict__ first_no) {
    const int64_t i = (int64_t)blockIdx.x * AB + threadIdx.x;
    if (i >= n || orig[i] >= n_owned) return;
    const double t = eta / h[i];
    if (drec[i].w * (t * t * t) > 0.5) atomicMin(first_no, number ? number[orig[i]] : orig[i]);          // [V]:560
}

// candidate record: {particle number (+inf: none), x, y, z, vx, vy, vz, h, 0}
__global__ __launch_bounds__(AB) void sink_create_fetch(const double4 *__restrict__ drec, const double *__restrict__ h,
                                                        const double *__restrict__ vx, const double *__restrict__ vy,
                                                        const double *__restrict__ vz, const int32_t *__restrict__ orig,
                                                        const int32_t *__restrict__ number, int64_t n, int32_t n_owned,
                                                        const int32_t *__restrict__ first_no, double *__restrict__ cand) {
    const int64_t i = (int64_t)blockIdx.x * AB + threadIdx.x;
    const int32_t want = *first_no;
    if (i == 0 && want == 0x7fffffff) { cand[0] = INFINITY; for (int k = 1; k < 9; k++) cand[k] = 0.0; }
    if (i >= n || orig[i] >= n_owned || want == 0x7fffffff) return;
    if ((number ? number[orig[i]] : orig[i]) != want) return;
    const double4 p = drec[i];
    cand[0] = (double)want; cand[1] = p.x; cand[2] = p.y; cand[3] = p.z;
    cand[4] = vx[i]; cand[5] = vy[i]; cand[6] = vz[i]; cand[7] = h[i]; cand[8] = 0.0;
}

__global__ void sink_create_apply(const double *__restrict__ cand, int ns, double *__restrict__ sink, double *__restrict__ srad,
                                  int32_t *__restrict__ created) {
    if (threadIdx.x != 0) return;
    *created = 0;
    if (!(cand[0] < 1.0e300) || ns >= MAX_SINKS) return;
    const double px = cand[1], py = cand[2], pz = cand[3], hi = cand[7];
    for (int j = 0; j < ns; j++) {
        const double d0 = sink[0 * MAX_SINKS + j] - px, d1 = sink[1 * MAX_SINKS + j] - py, d2 = sink[2 * MAX_SINKS + j] - pz;
        const double dr = sqrt(d0 * d0 + d1 * d1 + d2 * d2);                      // [V]:562
        if (dr < srad[j] + 2 * hi) return;                                        // [V]:563-565
    }
    sink[0 * MAX_SINKS + ns] = px; sink[1 * MAX_SINKS + ns] = py; sink[2 * MAX_SINKS + ns] = pz;
    sink[3 * MAX_SINKS + ns] = cand[4]; sink[4 * MAX_SINKS + ns] = cand[5]; sink[5 * MAX_SINKS + ns] = cand[6];
    sink[6 * MAX_SINKS + ns] = 0.00000000001;                                     // [V]:581
    sink[7 * MAX_SINKS + ns] = 0.0; sink[8 * MAX_SINKS + ns] = 0.0; sink[9 * MAX_SINKS + ns] = 0.0;
    srad[ns] = 2 * hi;                                                            // [V]:582
    *created = 1;
}

// the first candidate among this context's owned particles -> d_cand (9 doubles, device)
int sink_candidate(sph_ctx *c, double *d_cand) {
    if (!c->variable) { c->err = "sink creation: variable-h contexts only"; return SPH_ERR_STATE; }
    if (c->n > 0 && !c->order_valid) { c->err = "sink creation: needs the sorted order of the current positions"; return SPH_ERR_STATE; }
    const int32_t big = 0x7fffffff;
    AC_CHECK(hipMemcpyAsync(c->d_flags + 3, &big, sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
    const unsigned gb = (unsigned)((std::max<int64_t>(c->n, 1) + AB - 1) / AB);
    const int32_t *num = c->numbers_set ? c->number : nullptr;
    sink_create_scan<<<dim3(gb), dim3(AB), 0, c->stream>>>(reinterpret_cast<const double4 *>(c->drec), c->f[SPH_F_H], c->orig, num, c->n,
                                                           (int32_t)c->n_owned, c->p.eta, c->d_flags + 3);
    sink_create_fetch<<<dim3(gb), dim3(AB), 0, c->stream>>>(reinterpret_cast<const double4 *>(c->drec), c->f[SPH_F_H], c->f[SPH_F_VX],
                                                            c->f[SPH_F_VY], c->f[SPH_F_VZ], c->orig, num, c->n, (int32_t)c->n_owned,
                                                            c->d_flags + 3, d_cand);
    AC_CHECK(hipGetLastError());
    return SPH_OK;
}

// distance test against the existing sinks, then the new sink ([V]:561-595); d_cand: the winning candidate record
int sink_add_checked(sph_ctx *c, const double *d_cand, int32_t *created) {
    *created = 0;
    sink_create_apply<<<dim3(1), dim3(64), 0, c->stream>>>(d_cand, c->ns, c->sink, c->sink_radius, c->d_flags + 2);
    AC_CHECK(hipGetLastError());
    int32_t flag = 0;
    AC_CHECK(hipMemcpyAsync(&flag, c->d_flags + 2, sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    AC_CHECK(hipStreamSynchronize(c->stream));
    if (flag) { c->ns += 1; *created = 1; c->rates_valid = false; }
    return SPH_OK;
}

int sink_creation(sph_ctx *c, int32_t *created) {
    *created = 0;
    if (!c->variable || c->n == 0) return SPH_OK;
    double *cand = c->bbox_part + 1024 * 6 + 16;       // spare doubles behind the bbox partials and results
    { const int st = sink_candidate(c, cand); if (st != SPH_OK) return st; }
    return sink_add_checked(c, cand, created);
}

// ---- multi-GPU: mark + per-rank sums, then (after the caller all-gathered the sums) sink update + compaction ----------
int accrete_mark_ext(sph_ctx *c, int64_t src_off, double *d_partials) {
    if (!c->gx_src) { c->err = "sph_accrete_mark_dev: needs the all-gathered sources (sph_set_gravity_sources_dev)"; return SPH_ERR_STATE; }
    { const int st = global_keys_sorted(c); if (st != SPH_OK) return st; }
    const int64_t n = c->n, no = c->n_owned, ng = c->gx_n;
    if (src_off < 0 || src_off + no > ng) { c->err = "sph_accrete_mark_dev: owned block outside the source set"; return SPH_ERR_ARG; }
    RootBox rb;
    double size = 0.0;
    for (int a = 0; a < 3; a++) {
        rb.c[a] = (c->gx_box[3 + a] + c->gx_box[a]) / 2.0;
        size = std::max(size, c->gx_box[3 + a] - c->gx_box[a]);
    }
    rb.size = size;
    int32_t *keep = reinterpret_cast<int32_t *>(c->keys);
    unsigned long long *accmask = reinterpret_cast<unsigned long long *>(c->scratch);
    AC_CHECK(hipMemsetAsync(keep, 0, sizeof(int32_t) * (size_t)std::max<int64_t>(n, 1), c->stream));
    AC_CHECK(hipMemsetAsync(accmask, 0, sizeof(unsigned long long) * (size_t)std::max<int64_t>(n, 1), c->stream));
    AC_CHECK(hipMemsetAsync(d_partials, 0, sizeof(double) * 7 * MAX_SINKS, c->stream));
    if (no > 0 && ng > 0)
        acc_mark_ext<<<dim3((unsigned)((ng + AB - 1) / AB)), dim3(AB), 0, c->stream>>>(
            rb, c->g_keys_alt, c->g_vals_alt, ng, src_off, no, reinterpret_cast<const double4 *>(c->drec), c->inv, c->sink,
            c->sink_radius, c->ns, c->variable ? 1 : 0, c->p.bounding_size, keep, accmask);
    AC_CHECK(hipGetLastError());
    if (n > 0) {             // (all zero without a massive sink: nothing is marked then)
        const int nb = (int)std::min<int64_t>((n + AB - 1) / AB, 256);
        for (int k = 0; k < c->ns; k++) {
            acc_sums_partial<<<dim3(nb), dim3(AB), 0, c->stream>>>(n, k, accmask, reinterpret_cast<const double4 *>(c->drec),
                                                                   c->f[SPH_F_VX], c->f[SPH_F_VY], c->f[SPH_F_VZ], c->sink_part);
            acc_partials_final<<<dim3(1), dim3(64), 0, c->stream>>>(c->sink_part, nb, d_partials + (size_t)k * 7);
        }
        AC_CHECK(hipGetLastError());
    }
    c->acc_marked = true;
    return SPH_OK;
}

int accrete_apply_ext(sph_ctx *c, const double *d_all, int nranks, int stride, int32_t *d_keep_out, int64_t *removed) {
    *removed = 0;
    if (!c->acc_marked) { c->err = "sph_accrete_apply_dev: call sph_accrete_mark_dev first"; return SPH_ERR_STATE; }
    c->acc_marked = false;
    const int64_t n = c->n, no = c->n_owned;
    if (c->ns > 0)
        acc_sink_update_ranks<<<dim3(1), dim3(64), 0, c->stream>>>(c->ns, d_all, nranks, stride, c->sink);
    AC_CHECK(hipGetLastError());
    int32_t *keep = reinterpret_cast<int32_t *>(c->keys);
    int32_t *pos = reinterpret_cast<int32_t *>(c->keys_alt);
    if (d_keep_out && no > 0) AC_CHECK(hipMemcpyAsync(d_keep_out, keep, sizeof(int32_t) * (size_t)no, hipMemcpyDeviceToDevice, c->stream));
    int64_t n_new = 0;
    if (n > 0) {
        size_t sb = 0;
        AC_CHECK(rocprim::exclusive_scan(nullptr, sb, keep, pos, 0, (size_t)n, rocprim::plus<int32_t>(), c->stream));
        if (sb > c->sort_tmp_bytes) { c->err = "accrete: scan scratch too small"; return SPH_ERR_NOMEM; }
        AC_CHECK(rocprim::exclusive_scan(c->sort_tmp, sb, keep, pos, 0, (size_t)n, rocprim::plus<int32_t>(), c->stream));
        int32_t last[2];
        AC_CHECK(hipMemcpyAsync(&last[0], pos + (n - 1), sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
        AC_CHECK(hipMemcpyAsync(&last[1], keep + (n - 1), sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
        AC_CHECK(hipStreamSynchronize(c->stream));
        n_new = (int64_t)last[0] + last[1];
        CompactArgs ca{};
        for (int k = 0; k < 9; k++) { ca.src[k] = c->f[k]; ca.dst[k] = c->f_alt[k]; }
        ca.nf = 9;
        if (c->variable) { ca.src[9] = c->f[SPH_F_H]; ca.dst[9] = c->f_alt[9]; ca.nf = 10; }
        acc_compact<<<dim3((unsigned)((n + AB - 1) / AB)), dim3(AB), 0, c->stream>>>(ca, keep, pos, c->inv, n);
        AC_CHECK(hipGetLastError());
        for (int k = 0; k < 9; k++) std::swap(c->f[k], c->f_alt[k]);
        if (c->variable) std::swap(c->f[SPH_F_H], c->f_alt[9]);
    }
    c->numbers_set = false;                            // the caller's numbering has changed
    *removed = no - n_new;                             // owned particles that left; the ghosts are dropped as well
    c->n = n_new; c->n_slots = n_new; c->dead_below = 0;
    c->n_owned = n_new;
    AC_CHECK(launch_iota(c, c->orig, n_new));
    AC_CHECK(launch_iota(c, c->inv, n_new));
    AC_CHECK(hipStreamSynchronize(c->stream));
    c->grid_valid = c->rho_valid = c->eos_valid = c->rates_valid = c->tree_valid = c->order_valid = false;
    return sinks_cull(c);                              // [V]:610-613; the sinks are replicated: every rank drops the same ones
}

}  // namespace sph
