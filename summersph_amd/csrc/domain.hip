// domain.hip -- device-side building blocks of the multi-GPU slab decomposition (summersph_amd/dist.py):
//
//   owned bounding box      -> what the other ranks need to know to pick my ghosts
//   box selection           -> which of my particles lie in a peer's (bounding box + 2h): its ghosts
//   ghost swap              -> replace the ghost slots of the context without re-uploading the owned particles
//   rank reductions         -> sink accelerations summed / dt candidate min-reduced over the ranks' all-gathered
//                              partials on the device, in rank order (bitwise reproducible), no host round trip
//
// The reference is a single process (SUMMER_SPH.f90:863-930); these have no counterpart there.  Everything is
// enqueued on ctx->stream; only the selection returns counts to the host (message sizes).
#include <rocprim/device/device_select.hpp>
#include <rocprim/iterator/counting_iterator.hpp>

#include <algorithm>
#include <cmath>

#include "sph_internal.hpp"

namespace sph {

namespace {

struct InBox {
    const int32_t *inv;
    const double *x, *y, *z;
    double lo[3], hi[3];
    __device__ bool operator()(const int64_t &id) const {
        const int32_t s = inv[id];
        const double px = x[s];
        if (!(px >= lo[0] && px <= hi[0])) return false;        // slabs: almost every particle stops here
        const double py = y[s], pz = z[s];
        return py >= lo[1] && py <= hi[1] && pz >= lo[2] && pz <= hi[2];
    }
};

// appends `count` ghosts behind the occupied slots: state fields from vals[f*count + k] (row 9 = h when hfield is
// given), original id n_owned + k
__global__ __launch_bounds__(256) void append_ghosts(FieldPtrs9 fp, double *__restrict__ hfield, int32_t *__restrict__ orig,
                                                     int64_t first_slot, int64_t n_owned, int64_t count,
                                                     const double *__restrict__ vals) {
    const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (k >= count) return;
#pragma unroll
    for (int f = 0; f < 9; f++) fp.p[f][first_slot + k] = vals[(size_t)f * count + k];
    if (hfield) hfield[first_slot + k] = vals[(size_t)9 * count + k];
    orig[first_slot + k] = (int32_t)(n_owned + k);
}

__global__ __launch_bounds__(256) void set_numbers(int32_t *__restrict__ number, int64_t first, int64_t count,
                                                   const int64_t *__restrict__ src) {
    const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (k < count) number[first + k] = (int32_t)src[k];
}

// out[0 .. 3*MAX_SINKS) = my partial sink accelerations, out[3*MAX_SINKS] = my pending dt candidate
__global__ void pack_partials(const double *__restrict__ sink, const double *__restrict__ dtbuf, double *__restrict__ out) {
    const int k = threadIdx.x;
    if (k < 3 * MAX_SINKS) out[k] = sink[7 * MAX_SINKS + k];
    if (k == 3 * MAX_SINKS) out[k] = dtbuf[2];
}

// Where will my particles be after the coming kick + drift?  x' = x + (v + a dt'/2) dt' with dt' one of the three values
// the dt rule can produce (0.5, 1, 1.5 times the current dt): the union of the three boxes contains the box after the
// drift, so the other GPUs can pick my ghosts without another exchange of bounding boxes.
constexpr int PB_BLOCKS = 512;
__global__ __launch_bounds__(256) void pred_bbox_partial(const double *__restrict__ x, const double *__restrict__ y,
                                                         const double *__restrict__ z, const double *__restrict__ vx,
                                                         const double *__restrict__ vy, const double *__restrict__ vz,
                                                         const double *__restrict__ ax, const double *__restrict__ ay,
                                                         const double *__restrict__ az, const int32_t *__restrict__ orig,
                                                         int32_t n_owned, int64_t n, const double *__restrict__ dtbuf,
                                                         double *__restrict__ part) {
    __shared__ double sm[6][4];
    const double dt0 = dtbuf[0];
    double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        if (orig[i] >= n_owned) continue;
        const double p[3] = {x[i], y[i], z[i]}, v[3] = {vx[i], vy[i], vz[i]}, a[3] = {ax[i], ay[i], az[i]};
#pragma unroll
        for (int f = 1; f <= 3; f++) {
            const double dt = 0.5 * f * dt0;
#pragma unroll
            for (int k = 0; k < 3; k++) {
                const double q = p[k] + (v[k] + 0.5 * a[k] * dt) * dt;
                lo[k] = fmin(lo[k], q); hi[k] = fmax(hi[k], q);
            }
        }
    }
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < 3; k++) {
        double mn = lo[k], mx = hi[k];
        for (int o = 32; o > 0; o >>= 1) { mn = fmin(mn, __shfl_xor(mn, o, 64)); mx = fmax(mx, __shfl_xor(mx, o, 64)); }
        if (lane == 0) { sm[k][wv] = mn; sm[3 + k][wv] = mx; }
    }
    __syncthreads();
    if (threadIdx.x < 6) {
        double r = sm[threadIdx.x][0];
        for (int w = 1; w < 4; w++) r = threadIdx.x < 3 ? fmin(r, sm[threadIdx.x][w]) : fmax(r, sm[threadIdx.x][w]);
        part[(size_t)blockIdx.x * 6 + threadIdx.x] = r;
    }
}

// one wave per component (launched with 6 waves): lanes stride over the partials, then a wave reduction
__global__ void pred_bbox_final(const double *__restrict__ part, int nb, double *__restrict__ out) {
    const int k = threadIdx.x >> 6, lane = threadIdx.x & 63;
    double r = k < 3 ? INFINITY : -INFINITY;
    for (int b = lane; b < nb; b += 64) r = k < 3 ? fmin(r, part[(size_t)b * 6 + k]) : fmax(r, part[(size_t)b * 6 + k]);
    for (int o = 32; o > 0; o >>= 1) r = k < 3 ? fmin(r, __shfl_xor(r, o, 64)) : fmax(r, __shfl_xor(r, o, 64));
    if (lane == 0) out[k] = r;
}

// sink accelerations = sum over ranks (rank order); optionally get_next_timestep's rule ([F]:851-859, t = t + dt of
// [F]:914 first) with the minimum of the ranks' candidates
__global__ void apply_partials(const double *__restrict__ all, int nranks, int stride, double *__restrict__ sink,
                               int apply_dt, double dt_max, double dt_min, double *__restrict__ dtbuf) {
    const int k = threadIdx.x;
    if (k < 3 * MAX_SINKS) {
        double s = 0.0;
        for (int r = 0; r < nranks; r++) s += all[(size_t)r * stride + k];
        sink[7 * MAX_SINKS + k] = s;
    }
    if (k == 3 * MAX_SINKS && apply_dt) {
        double cand = INFINITY;
        for (int r = 0; r < nranks; r++) cand = fmin(cand, all[(size_t)r * stride + k]);
        double dt = dtbuf[0];
        dtbuf[1] = dtbuf[1] + dt;
        if (cand > 2 * dt && 1.5 * dt < dt_max) dt = 1.5 * dt;
        else if (cand < 0.5 * dt && dt * 0.5 > dt_min) dt = 0.5 * dt;
        dtbuf[0] = dt;
        dtbuf[2] = cand;
    }
}

}  // namespace

#define DM_CHECK(expr)                                                      \
    do {                                                                    \
        hipError_t _e = (expr);                                             \
        if (_e != hipSuccess) {                                             \
            c->err = std::string(#expr) + ": " + hipGetErrorString(_e);     \
            return SPH_ERR_HIP;                                             \
        }                                                                   \
    } while (0)

// ids (ascending original id) of the owned particles inside each of nbox boxes -> c->sel_ids + b * n_owned; the counts stay
// on the device (c->sel_count) and travel to pinned memory behind the selection, without a wait
int domain_select_boxes_enqueue(sph_ctx *c, int nbox, const double *boxes) {
    const int64_t no = c->n_owned;
    c->sel_stride = no;
    int64_t *h = reinterpret_cast<int64_t *>(c->h_pinned + 64);
    for (int b = 0; b < nbox; b++) h[b] = 0;
    if (!c->sel_count) {
        if (ctx_alloc(c, &c->sel_count, (size_t)MAX_SEL_BOXES, "selection counts") != SPH_OK) return SPH_ERR_NOMEM;
    }
    if (nbox == 0) return SPH_OK;
    if (no == 0) { DM_CHECK(hipMemsetAsync(c->sel_count, 0, (size_t)nbox * sizeof(int64_t), c->stream)); return SPH_OK; }
    const size_t need = (size_t)nbox * (size_t)no;
    if (need > c->sel_cap) {
        ctx_free(c, c->sel_ids);
        c->sel_cap = 0;
        if (ctx_alloc(c, &c->sel_ids, need, "selection ids") != SPH_OK) return SPH_ERR_NOMEM;
        c->sel_cap = need;
    }
    rocprim::counting_iterator<int64_t> first(0);
    for (int b = 0; b < nbox; b++) {
        InBox pred{c->inv, c->f[SPH_F_X], c->f[SPH_F_Y], c->f[SPH_F_Z],
                   {boxes[b * 6 + 0], boxes[b * 6 + 1], boxes[b * 6 + 2]}, {boxes[b * 6 + 3], boxes[b * 6 + 4], boxes[b * 6 + 5]}};
        size_t tmp = 0;
        DM_CHECK(rocprim::select(nullptr, tmp, first, c->sel_ids + (size_t)b * no, c->sel_count + b, (size_t)no, pred, c->stream));
        if (tmp > c->sel_tmp_bytes) {
            DM_CHECK(hipStreamSynchronize(c->stream));
            ctx_free_ptr(c, c->sel_tmp);
            c->sel_tmp = nullptr; c->sel_tmp_bytes = 0;
            if (ctx_alloc_bytes(c, &c->sel_tmp, tmp, "selection scratch") != SPH_OK) return SPH_ERR_NOMEM;
            c->sel_tmp_bytes = tmp;
        }
        tmp = c->sel_tmp_bytes;
        DM_CHECK(rocprim::select(c->sel_tmp, tmp, first, c->sel_ids + (size_t)b * no, c->sel_count + b, (size_t)no, pred, c->stream));
    }
    DM_CHECK(hipMemcpyAsync(h, c->sel_count, (size_t)nbox * sizeof(int64_t), hipMemcpyDeviceToHost, c->stream));
    return SPH_OK;
}

void domain_selected_counts(sph_ctx *c, int nbox, int64_t *counts) {
    const int64_t *h = reinterpret_cast<const int64_t *>(c->h_pinned + 64);
    for (int b = 0; b < nbox; b++) counts[b] = h[b];
}

int domain_select_boxes(sph_ctx *c, int nbox, const double *boxes, int64_t *counts) {
    const int st = domain_select_boxes_enqueue(c, nbox, boxes);
    if (st != SPH_OK) return st;
    DM_CHECK(hipStreamSynchronize(c->stream));
    domain_selected_counts(c, nbox, counts);
    return SPH_OK;
}

int domain_replace_ghosts(sph_ctx *c, int64_t count, const double *d_vals) {
    const int64_t n_old = c->n_slots;
    if (n_old + count > c->cap) {
        c->err = "ghost capacity exceeded (sph_reserve more slots before sph_upload)";
        return SPH_ERR_NOMEM;
    }
    if (count > 0) {
        FieldPtrs9 fp{};
        for (int f = 0; f < 9; f++) fp.p[f] = c->f[f];
        append_ghosts<<<dim3((unsigned)((count + 255) / 256)), dim3(256), 0, c->stream>>>(fp, c->variable ? c->f[SPH_F_H] : nullptr, c->orig, n_old, c->n_owned, count, d_vals);
        DM_CHECK(hipGetLastError());
    }
    c->dead_below = n_old;             // slots below with an original id >= n_owned are the old ghosts
    c->n_slots = n_old + count;
    c->n = c->n_owned + count;
    return SPH_OK;
}

hipError_t launch_set_numbers(sph_ctx *c, int64_t first, int64_t count, const int64_t *d_numbers) {
    if (count <= 0) return hipSuccess;
    set_numbers<<<dim3((unsigned)((count + 255) / 256)), dim3(256), 0, c->stream>>>(c->number, first, count, d_numbers);
    return hipGetLastError();
}

hipError_t launch_pack_partials(sph_ctx *c, double *d_out, bool predict_box) {
    pack_partials<<<dim3(1), dim3(256), 0, c->stream>>>(c->sink, c->d_dt, d_out);
    // [193, 199): the predicted bounding box of the owned particles after the coming drift (needs current rates)
    if (predict_box && c->rates_valid && c->n_slots == c->n) {
        const int nb = (int)std::max<int64_t>(1, std::min<int64_t>((c->n + 255) / 256, PB_BLOCKS));
        pred_bbox_partial<<<dim3(nb), dim3(256), 0, c->stream>>>(c->f[SPH_F_X], c->f[SPH_F_Y], c->f[SPH_F_Z], c->f[SPH_F_VX], c->f[SPH_F_VY],
                                                                 c->f[SPH_F_VZ], c->f[SPH_F_AX], c->f[SPH_F_AY], c->f[SPH_F_AZ], c->orig,
                                                                 (int32_t)c->n_owned, c->n, c->d_dt, c->bbox_part);
        pred_bbox_final<<<dim3(1), dim3(384), 0, c->stream>>>(c->bbox_part, nb, d_out + 3 * MAX_SINKS + 1);
    } else {
        (void)hipMemsetAsync(d_out + 3 * MAX_SINKS + 1, 0xff, 6 * sizeof(double), c->stream);      // NaN: no prediction
    }
    return hipGetLastError();
}

hipError_t launch_apply_partials(sph_ctx *c, const double *d_all, int nranks, int stride, bool apply_dt) {
    apply_partials<<<dim3(1), dim3(256), 0, c->stream>>>(d_all, nranks, stride, c->sink, apply_dt ? 1 : 0, c->p.dt_max, c->p.dt_min, c->d_dt);
    return hipGetLastError();
}

}  // namespace sph
