// api.hip -- the C ABI of libsummersph_hip.so (declared in include/summersph.h).
// Owns the context, device memory and the call-order state machine; all arithmetic lives in
// grid.hip / pairs.hip / integrate.hip.  There is deliberately no CPU path here.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstring>
#include <new>

#include "sph_internal.hpp"

using namespace sph;

namespace {

#define API_HIP(expr)                                                       \
    do {                                                                    \
        hipError_t _e = (expr);                                             \
        if (_e != hipSuccess) {                                             \
            c->err = std::string(#expr) + ": " + hipGetErrorString(_e);     \
            return SPH_ERR_HIP;                                             \
        }                                                                   \
    } while (0)

#define API_TRY(expr)                          \
    do {                                       \
        int _s = (expr);                       \
        if (_s != SPH_OK) return _s;           \
    } while (0)

struct DeviceGuard {
    int prev = -1;
    explicit DeviceGuard(int dev) { (void)hipGetDevice(&prev); if (prev != dev) (void)hipSetDevice(dev); else prev = -1; }
    ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};

void free_particle_arrays(sph_ctx *c) {
    for (auto &p : c->f) ctx_free(c, p);
    for (auto &p : c->f_alt) ctx_free(c, p);
    ctx_free(c, c->orig); ctx_free(c, c->orig_alt); ctx_free(c, c->inv); ctx_free(c, c->scratch); ctx_free(c, c->number);
    ctx_free(c, c->drec); ctx_free(c, c->frec);
    ctx_free(c, c->keys); ctx_free(c, c->keys_alt); ctx_free(c, c->vals); ctx_free(c, c->vals_alt);
    ctx_free(c, c->sort_tmp);
    ctx_free(c, c->nlist); ctx_free(c, c->ncount); ctx_free(c, c->wave_max); ctx_free(c, c->ntail); ctx_free(c, c->wave_class); ctx_free(c, c->leaf_half);
    ctx_free(c, c->plan_d); ctx_free(c, c->plan_f); ctx_free(c, c->plan_h); ctx_free(c, c->deal);
    for (auto &g : c->g_cache) ctx_free(c, g);
    c->grav_valid = false;
    ctx_free(c, c->sel_ids); c->sel_cap = 0;
    ctx_free(c, c->prec); ctx_free(c, c->lrec); ctx_free(c, c->mkeys); ctx_free(c, c->mkeys_alt);
    ctx_free(c, c->mvals); ctx_free(c, c->mvals_alt); ctx_free(c, c->msort_tmp); ctx_free(c, c->h_new);
    gravity_free(c);                                     // tree arrays are (re)allocated by the next tree build
    c->msort_tmp_bytes = 0;
    c->cap = 0; c->nl_cap = 0; c->nl_waves_cap = 0; c->sort_tmp_bytes = 0;
    c->h_new_is_build = false; c->list_has_margin = false;
}

int ensure_capacity(sph_ctx *c, int64_t n) {
    if (std::max(n, c->reserve) <= c->cap) return SPH_OK;
    free_particle_arrays(c);
    const int64_t cap = std::max<int64_t>(n + n / 16 + 64, c->reserve);
    for (auto &p : c->f) API_TRY(ctx_alloc(c, &p, (size_t)cap, "state"));
    for (auto &p : c->f_alt) API_TRY(ctx_alloc(c, &p, (size_t)cap, "state (alt)"));
    API_TRY(ctx_alloc(c, &c->orig, (size_t)cap, "ids"));
    API_TRY(ctx_alloc(c, &c->orig_alt, (size_t)cap, "ids (alt)"));
    API_TRY(ctx_alloc(c, &c->inv, (size_t)cap, "ids (inverse)"));
    API_TRY(ctx_alloc(c, &c->number, (size_t)cap, "particle numbers"));
    API_TRY(ctx_alloc(c, &c->scratch, (size_t)cap, "scratch"));
    API_TRY(ctx_alloc(c, &c->drec, (size_t)cap * 4, "density records"));
    API_TRY(ctx_alloc(c, &c->frec, (size_t)cap * FREC, "force records"));
    API_TRY(ctx_alloc(c, &c->keys, (size_t)cap, "keys"));
    API_TRY(ctx_alloc(c, &c->keys_alt, (size_t)cap, "keys (alt)"));
    API_TRY(ctx_alloc(c, &c->vals, (size_t)cap, "vals"));
    API_TRY(ctx_alloc(c, &c->vals_alt, (size_t)cap, "vals (alt)"));
    size_t tmp = 0;
    API_HIP(grid_sort_tmp_bytes(cap, &tmp));
    c->sort_tmp_bytes = tmp;
    API_TRY(ctx_alloc_bytes(c, &c->sort_tmp, tmp ? tmp : 1, "sort scratch"));
    c->nl_waves_cap = (cap + 63) / 64;
    c->nl_cap = 96;   // grows on demand (nlist_build); the tiled fixed-h build writes 16-bit entries (tile_common.hpp)
    API_TRY(ctx_alloc(c, &c->nlist, (size_t)c->nl_waves_cap * c->nl_cap * (c->tiled ? 32 : 64), "neighbour list"));
    API_TRY(ctx_alloc(c, &c->ncount, (size_t)cap, "neighbour counts"));
    API_TRY(ctx_alloc(c, &c->wave_max, (size_t)c->nl_waves_cap, "wave max"));
    API_TRY(ctx_alloc(c, &c->wave_class, (size_t)c->nl_waves_cap, "wave classes"));
    API_TRY(ctx_alloc(c, &c->plan_d, (size_t)(cap / 256 + 2) * 8, "tile plans (density)"));
    API_TRY(ctx_alloc(c, &c->plan_f, (size_t)(cap / 256 + 2) * 8, "tile plans (forces)"));
    API_TRY(ctx_alloc(c, &c->plan_h, (size_t)(cap / 256 + 2) * 16, "tile plans (forces, half groups)"));
    API_TRY(ctx_alloc(c, &c->deal, 2 * ((size_t)cap + 256), "dealing order"));
    if (c->variable) {
        API_TRY(ctx_alloc(c, &c->prec, (size_t)cap * 4, "position+h records"));
        API_TRY(ctx_alloc(c, &c->lrec, (size_t)cap * 4, "leaf boxes"));
        API_TRY(ctx_alloc(c, &c->h_new, (size_t)cap, "h scratch"));
        API_TRY(ctx_alloc(c, &c->ntail, (size_t)cap, "margin counts"));
        API_TRY(ctx_alloc(c, &c->leaf_half, (size_t)cap, "leaf half edges"));
    }
    {   // octree path keys: leaf boxes (variable h), self-gravity tree, accretion
        API_TRY(ctx_alloc(c, &c->mkeys, (size_t)cap, "octree keys"));
        API_TRY(ctx_alloc(c, &c->mkeys_alt, (size_t)cap, "octree keys (alt)"));
        API_TRY(ctx_alloc(c, &c->mvals, (size_t)cap, "octree vals"));
        API_TRY(ctx_alloc(c, &c->mvals_alt, (size_t)cap, "octree vals (alt)"));
        size_t mt = 0;
        API_HIP(varh_sort_tmp_bytes(cap, &mt));
        c->msort_tmp_bytes = mt;
        API_TRY(ctx_alloc_bytes(c, &c->msort_tmp, mt ? mt : 1, "octree sort scratch"));
    }
    c->cap = cap;
    return SPH_OK;
}

#pragma clang fp contract(off)
void host_tables(int nq, std::vector<double> &w, std::vector<double> &dw) {
    // SUMMER_SPH.f90:55-79 (cubic spline M4 sampled on q in [0,2]); dq = 2.0_dp/nq
    w.assign((size_t)TAB_LEN(nq), 0.0);        // knots 0..nq + one zero of padding (table_lerp)
    dw.assign((size_t)TAB_LEN(nq), 0.0);
    const double dq = 2.0 / nq;
    for (int i = 0; i < nq; i++) {         // knot nq (q = 2, where W = dW = 0) stays an exact zero whatever nq * dq rounds to
        const double q = i * dq;
        if (q >= 0.0 && q <= 1.0) {
            w[i] = 1.0 - 1.5 * (q * q) + 0.75 * (q * q * q);
            dw[i] = -3.0 * q + 2.25 * (q * q);
        } else if (q > 1.0 && q <= 2.0) {
            const double t = 2.0 - q;
            w[i] = 0.25 * (t * t * t);
            dw[i] = -0.75 * (t * t);
        }
    }
}

void host_grav_table(int nq, std::vector<double> &g) {
    // SUMMER_SPH.f90:81-101: spline-softened force factor, 1 outside the support
    g.assign((size_t)nq + 1, 1.0);
    const double dq = 2.0 / nq;
    for (int i = 0; i <= nq; i++) {
        const double q = i * dq;
        if (q >= 0.0 && q <= 1.0) {
            const double q3 = q * q * q, q5 = q3 * q * q, q6 = q5 * q;
            g[i] = ((40.0 * q3) - (36.0 * q5) + (15.0 * q6)) / 30.0;
        } else if (q > 1.0 && q <= 2.0) {
            const double q3 = q * q * q, q4 = q3 * q, q5 = q4 * q, q6 = q5 * q;
            g[i] = ((80.0 * q3) - (90.0 * q4) + (36.0 * q5) - (5 * q6) - 2) / 30.0;
        }
    }
}

struct Timed {
    sph_ctx *c; int id; hipEvent_t e0 = nullptr, e1 = nullptr;
    Timed(sph_ctx *c_, int id_) : c(c_), id(id_) {
        if ((c->timing & (1u << id)) && (c->timing_seen[id]++ % c->timing_stride) == 0) {
            (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
            (void)hipEventRecord(e0, c->stream);
        }
    }
    ~Timed() {
        if (e0) {
            (void)hipEventRecord(e1, c->stream);
            c->tslot[id].pending.emplace_back(e0, e1);
        }
    }
};

void resolve_timing(sph_ctx *c) {
    for (auto &s : c->tslot) {
        for (auto &pr : s.pending) {
            float ms = 0.f;
            (void)hipEventSynchronize(pr.second);
            if (hipEventElapsedTime(&ms, pr.first, pr.second) == hipSuccess) { s.total_ms += ms; s.launches++; }
            (void)hipEventDestroy(pr.first); (void)hipEventDestroy(pr.second);
        }
        s.pending.clear();
    }
}

// The whole-tile kernels are persistent, one workgroup per CU.  density_wt walks over groups of 1024 targets: with fewer groups
// than CUs it leaves CUs idle, and below ~0.75 groups per CU the gather kernel of pairs.hip (a workgroup per 256 targets) is
// faster -- measured on MI355X (tests/tools/small_n_ab.sh, ms per pass): 12 000 particles 0.032 vs 0.069, 100 000: 0.048 vs
// 0.071, 300 000 (293 groups): 0.098 vs 0.077 for the tile kernel; 200 000 particles with ~200 neighbours each (196 groups,
// table-free tile): 0.205 vs 0.358 -- hence one group per CU at least.  forces_q (groups of 256, four lanes per target) wins at every
// size, 47 groups included (12 000 particles: 0.032 vs 0.049 ms).  SPH_TILE_MIN_GROUPS_D / _F: A/B switches (x0.01 groups per CU).
bool use_tile_kernel(const sph_ctx *c, bool forces) {
    static const int thr_d = getenv("SPH_TILE_MIN_GROUPS_D") ? atoi(getenv("SPH_TILE_MIN_GROUPS_D")) : 100;
    static const int thr_f = getenv("SPH_TILE_MIN_GROUPS_F") ? atoi(getenv("SPH_TILE_MIN_GROUPS_F")) : 0;
    if (!c->whole_tile || !(forces ? c->wt_ok_f : c->wt_ok)) return false;
    const int64_t groups = (c->n + (forces ? 255 : 1023)) / (forces ? 256 : 1024);
    return 100 * groups >= (int64_t)(forces ? thr_f : thr_d) * c->num_cus;
}

int do_density(sph_ctx *c) {
    static const bool no_refresh = getenv("SPH_NO_H_REFRESH") != nullptr;      // A/B switch
    if (!no_refresh && !c->grid_valid && c->variable && c->h_refresh_ok && c->order_valid && c->leaf_valid && c->n_slots == c->n) {
        // same positions, same particles, new h (calc_smoothing, Variable.f90:1152): the sorted order, the cell table and
        // the leaf cells stand; only what depends on h is redone -- and a self-gravity tree stays valid
        // growth of h against the lengths the list was built with -- known only when sph_update_h produced the new h (its
        // swap left the old lengths in h_new); after an upload / scatter of h the list is built, never re-flagged
        API_TRY(varh_h_stats(c, c->h_new_is_build));
        c->h_new_is_build = false;
        { Timed t(c, SPH_K_LEAF); API_TRY(varh_refresh_h(c)); }
        // the list of the new lengths: re-flagged from the list in place when no h outgrew its margin, else built.  The
        // re-flag pass and the density pass that follows it walk the same rows: one kernel does both (SPH_NO_FUSED_REFLAG: A/B)
        static const bool no_fused = getenv("SPH_NO_FUSED_REFLAG") != nullptr;
        c->rates_valid = false; c->rho_valid = false; c->eos_valid = false;
        if (varh_can_reflag(c) && !no_fused) {
            Timed t(c, SPH_K_REFLAG);
            API_TRY(varh_reflag_density(c, make_pair_const(c)));
            c->grid_valid = true; c->h_refresh_ok = false;
            c->density_passes++;
            c->rho_valid = true; c->eos_valid = true;
            return SPH_OK;
        }
        if (varh_can_reflag(c)) { Timed t(c, SPH_K_REFLAG); API_TRY(varh_nlist_reflag(c)); }
        else { Timed t(c, SPH_K_NLIST); API_TRY(varh_nlist_build(c)); }
        c->grid_valid = true;
    }
    c->h_refresh_ok = false;
    if (!c->grid_valid) {
        c->h_new_is_build = false;
        if (c->variable) API_TRY(varh_h_stats(c));
        { Timed t(c, SPH_K_GRID); API_TRY(grid_rebuild(c)); }
        c->order_valid = true; c->derived_kept = false; c->grav_valid = false;
        c->path_keys_valid = false;
        c->rates_valid = false; c->rho_valid = false; c->eos_valid = false; c->tree_valid = false;
        c->wave_class_valid = false; c->interior_done = false;
        if (c->variable) {
            c->leaf_valid = false;
            { Timed t(c, SPH_K_LEAF); API_TRY(varh_leaf_build(c)); }
            c->leaf_valid = true;
            { Timed t(c, SPH_K_NLIST); API_TRY(varh_nlist_build(c)); }
        } else {
            Timed t(c, SPH_K_NLIST); API_TRY(c->tiled ? nlist_build_tiled(c) : nlist_build(c));
        }
        c->grid_valid = true;
    }
    const PairConst pc = make_pair_const(c);
    Timed t(c, SPH_K_DENSITY);
    if ((c->p.flags & SPH_FLAG_REUSE_DENSITY) && c->rho_valid) {
        API_HIP(c->variable ? launch_eos_only_v(c, pc) : launch_eos_only(c, pc));
    } else {
        API_HIP(c->variable ? launch_density_v(c, pc) : (use_tile_kernel(c, false) ? launch_density_wt(c, pc) : launch_density(c, pc)));
        c->density_passes++;
    }
    c->rho_valid = true; c->eos_valid = true;
    return SPH_OK;
}

int do_forces(sph_ctx *c) {
    if (!c->eos_valid || !c->grid_valid) { c->err = "sph_forces: call sph_density first"; return SPH_ERR_STATE; }
    const PairConst pc = make_pair_const(c);
    if (c->gravity) {                                   // particle_gravforces, [F]:825 -- before the sink and SPH terms
        Timed t(c, SPH_K_GRAVITY);
        const bool reuse = (c->p.flags & SPH_FLAG_REUSE_GRAVITY) != 0;
        const size_t bytes = (size_t)c->n * sizeof(double);
        if (reuse && c->grav_valid && c->tree_valid) {
            // same positions, masses, h, tree and sorted order as at the last walk: the same accelerations, bit for bit
            API_HIP(hipMemcpyAsync(c->f[SPH_F_AX], c->g_cache[0], bytes, hipMemcpyDeviceToDevice, c->stream));
            API_HIP(hipMemcpyAsync(c->f[SPH_F_AY], c->g_cache[1], bytes, hipMemcpyDeviceToDevice, c->stream));
            API_HIP(hipMemcpyAsync(c->f[SPH_F_AZ], c->g_cache[2], bytes, hipMemcpyDeviceToDevice, c->stream));
        } else {
            if (!c->tree_valid) { API_TRY(gravity_tree_build(c)); c->tree_valid = true; }
            { Timed tw(c, SPH_K_GRAV_WALK); API_HIP(launch_gravity(c)); }
            if (reuse) {
                for (auto &g : c->g_cache)
                    if (!g) API_TRY(ctx_alloc(c, &g, (size_t)c->cap, "gravity cache"));
                API_HIP(hipMemcpyAsync(c->g_cache[0], c->f[SPH_F_AX], bytes, hipMemcpyDeviceToDevice, c->stream));
                API_HIP(hipMemcpyAsync(c->g_cache[1], c->f[SPH_F_AY], bytes, hipMemcpyDeviceToDevice, c->stream));
                API_HIP(hipMemcpyAsync(c->g_cache[2], c->f[SPH_F_AZ], bytes, hipMemcpyDeviceToDevice, c->stream));
                c->grav_valid = true;
            }
        }
    }
    { Timed t(c, SPH_K_SINKACC); API_HIP(launch_sink_accel(c, pc)); }
    { Timed t(c, SPH_K_FORCES); API_HIP(c->variable ? launch_forces_v(c, pc) : (use_tile_kernel(c, true) ? launch_forces_wt(c, pc, 0) : launch_forces(c, pc))); }
    c->force_passes++;
    c->rates_valid = true;
    return SPH_OK;
}

// forces in two parts (multi-GPU overlap): 1 = sink gravity + the waves that cannot see a ghost, while the ghost
// fields are still travelling; 2 = the remaining waves, after sph_refresh_eos
int do_forces_part(sph_ctx *c, int part) {
    if (c->variable || c->gravity) { c->err = "sph_forces_part: fixed-h contexts without self-gravity only"; return SPH_ERR_STATE; }
    if (!c->eos_valid || !c->grid_valid) { c->err = "sph_forces_part: call sph_density first"; return SPH_ERR_STATE; }
    const PairConst pc = make_pair_const(c);
    if (part == 1) {
        if (!c->wave_class_valid) { API_HIP(launch_classify_waves(c)); c->wave_class_valid = true; }
        { Timed t(c, SPH_K_SINKACC); API_HIP(launch_sink_accel(c, pc)); }
        { Timed t(c, SPH_K_FORCES); API_HIP(use_tile_kernel(c, true) ? launch_forces_wt(c, pc, 1) : launch_forces(c, pc, 1)); }
        c->interior_done = true;
        c->rates_valid = false;
        return SPH_OK;
    }
    if (!c->interior_done) { c->err = "sph_forces_part: part 2 before part 1"; return SPH_ERR_STATE; }
    { Timed t(c, SPH_K_FORCES); API_HIP(use_tile_kernel(c, true) ? launch_forces_wt(c, pc, 2) : launch_forces(c, pc, 2)); }
    c->interior_done = false;
    c->force_passes++;
    c->rates_valid = true;
    return SPH_OK;
}

int do_kick(sph_ctx *c, double dt, bool dev) {
    if (!c->rates_valid) { c->err = "sph_kick: rates are stale, call sph_forces first"; return SPH_ERR_STATE; }
    Timed t(c, SPH_K_KICK);
    API_HIP(launch_kick(c, dt, dev));
    c->eos_valid = false;
    return SPH_OK;
}

int do_drift(sph_ctx *c, double dt, bool dev) {
    Timed t(c, SPH_K_DRIFT);
    API_HIP(launch_drift(c, dt, dev));
    c->grid_valid = false; c->rho_valid = false; c->eos_valid = false; c->order_valid = false;
    return SPH_OK;
}

int put_dt(sph_ctx *c, double dt, double t) {
    c->h_pinned[16] = dt; c->h_pinned[17] = t; c->h_pinned[18] = 0.0;
    API_HIP(hipMemcpyAsync(c->d_dt, c->h_pinned + 16, 3 * sizeof(double), hipMemcpyHostToDevice, c->stream));
    return SPH_OK;
}

// The fixed-h build path reads its reports (longest list, non-finite positions) one build late so that the steady state never
// waits for the host.  Every entry point that synchronises anyway and hands results to the caller calls this afterwards: the
// reports of the LAST build have arrived by then, so a list that overflowed in the final build of a run (or in the only build
// after an upload) is an error here, never a silently truncated result.  Requires the stream to be idle.
int drain_reports(sph_ctx *c) {
    if (c->ring_nl_valid) {
        const int32_t *rep = reinterpret_cast<const int32_t *>(c->h_pinned + 240 + 8 * (1 - c->ring_nl));
        if (rep[0] > c->nl_cap || (c->tiled && rep[3] >= 65536)) {
            c->err = "neighbour list overflowed in the last build (lists or candidate intervals grew beyond their headroom within one step): results are incomplete";
            c->ring_nl_valid = false; c->grid_valid = false; c->rho_valid = false; c->eos_valid = false; c->rates_valid = false;
            return SPH_ERR_STATE;
        }
    }
    if (c->ring_bbox_valid && !c->bbox_exact) {
        const double *bb = c->h_pinned + 200 + 16 * (1 - c->ring_bbox);
        if (*reinterpret_cast<const int32_t *>(bb + 6) != 0) { c->err = "non-finite particle position at grid build"; return SPH_ERR_NONFINITE; }
    }
    return SPH_OK;
}

int get_dt(sph_ctx *c, double *dt, double *t) {
    API_HIP(hipMemcpyAsync(c->h_pinned + 20, c->d_dt, 3 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    API_HIP(hipStreamSynchronize(c->stream));
    API_TRY(drain_reports(c));
    if (dt) *dt = c->h_pinned[20];
    if (t) *t = c->h_pinned[21];
    return SPH_OK;
}

// calc_smoothing (Variable.f90:515-546) on the grid / leaf boxes of the last evaluation
int do_update_h(sph_ctx *c) {
    if (!c->variable) { c->err = "sph_update_h: context is not in variable-h mode"; return SPH_ERR_STATE; }
    if (!c->grid_valid || !c->rho_valid) { c->err = "sph_update_h: needs the density of the current positions"; return SPH_ERR_STATE; }
    const PairConst pc = make_pair_const(c);
    { Timed t(c, SPH_K_UPDATE_H); API_HIP(launch_update_h(c, pc)); }
    // h changed: reaches, neighbour sets, rho all depend on it -- but nothing else does (do_density's short path)
    c->grid_valid = false; c->rho_valid = false; c->eos_valid = false;
    c->h_refresh_ok = true;
    c->h_new_is_build = true;                           // launch_update_h swapped: h_new = the lengths of the list in place
    c->grav_valid = false;                              // the softening length of [V]:296 is the particle's own h
    return SPH_OK;
}

int do_accrete(sph_ctx *c, int64_t *removed, int32_t *d_keep = nullptr) {
    if (!c->order_valid) { c->err = "sph_accrete_and_cull: needs the grid of the current positions (call sph_density first)"; return SPH_ERR_STATE; }
    if (c->n_owned != c->n) { c->err = "sph_accrete_and_cull: not available with ghost particles"; return SPH_ERR_STATE; }
    if (!c->bbox_exact) {       // the octree's root box is the exact bounding box: take it from the last build's read-back slot
        API_HIP(hipStreamSynchronize(c->stream));
        const double *bb = c->h_pinned + 200 + 16 * (1 - c->ring_bbox);
        for (int a = 0; a < 6; a++) c->bbox[a] = bb[a];
        c->bbox_exact = true;
    }
    API_TRY(accrete_and_cull(c, removed, d_keep));      // includes [V]'s cull of the sinks (Variable.f90:610-613)
    if (*removed > 0) c->numbers_set = false;           // the caller's numbering changed with the pack()
    return SPH_OK;
}

int one_step_device_dt(sph_ctx *c) {
    // SUMMER_SPH.f90:889-916
    API_TRY(do_density(c));
    API_TRY(do_forces(c));
    {   // kick + drift, one pass over the state (bitwise what sph_kick + sph_drift give)
        if (!c->rates_valid) { c->err = "sph_step: rates are stale"; return SPH_ERR_STATE; }
        Timed t(c, SPH_K_KICK);
        API_HIP(launch_kick_drift(c));
        c->grid_valid = false; c->rho_valid = false; c->eos_valid = false; c->order_valid = false;
    }
    API_TRY(do_density(c));
    API_TRY(do_forces(c));
    {   // closing kick + get_next_timestep, one pass (bitwise what sph_kick + sph_next_dt give)
        if (!c->rates_valid) { c->err = "sph_step: rates are stale"; return SPH_ERR_STATE; }
        Timed t(c, SPH_K_DT);
        API_HIP(launch_kick_next_dt(c, true));
        c->eos_valid = false;
    }
    if (c->variable) API_TRY(do_update_h(c));          // Variable.f90:1152
    if (c->variable && (c->p.flags & SPH_FLAG_SINK_CREATION)) {      // Variable.f90:1155, before accretion and bounds
        int32_t created = 0;
        API_TRY(sink_creation(c, &created));
    }
    if (c->p.flags & SPH_FLAG_ACCRETE_CULL) {          // SUMMER_SPH.f90:919-920
        int64_t removed = 0;
        API_TRY(do_accrete(c, &removed));
    }
    return SPH_OK;
}

// What an overwritten field invalidates -- shared by sph_upload_field*, sph_scatter_field_dev and sph_scatter_fields_dev.
void field_written(sph_ctx *c, int field) {
    if (field <= SPH_F_Z || field == SPH_F_M || field == SPH_F_H) {
        c->h_refresh_ok = field == SPH_F_H && (c->grid_valid || c->h_refresh_ok);     // only h is newer than the grid
        c->grid_valid = false; c->rho_valid = false;
    }
    if (field == SPH_F_H) c->h_new_is_build = false;     // h_new no longer pairs with the new h: the next list is built, not re-flagged
    if (field <= SPH_F_Z || field == SPH_F_M) {
        // a new configuration: the one-build-late reports (list overflow, bounding box, non-finite flag) describe the old one,
        // so the next build waits for its own read-backs instead of trusting them
        c->ring_nl_valid = false; c->ring_bbox_valid = false;
    }
    if (field <= SPH_F_Z) c->order_valid = false;
    if (field <= SPH_F_ALPHA || field == SPH_F_RHO || field == SPH_F_H || field == SPH_F_OMEGA) c->eos_valid = false;
    c->derived_kept = false;
    c->grav_valid = false;
}

bool field_ready(const sph_ctx *c, int field) {
    if (field <= SPH_F_ALPHA) return true;
    if (field == SPH_F_H) return c->variable;
    // accretion / cull compacted the derived arrays along with the state: the survivors keep the values of the last
    // evaluation, as the reference's pack leaves them ([F]:481,554)
    if (c->derived_kept) return field != SPH_F_OMEGA || c->variable;
    // derived arrays are in the current slot order as long as no re-sort happened since they were
    // written (a re-sort clears rates_valid)
    if (field == SPH_F_OMEGA) return c->variable && (c->rho_valid || c->rates_valid);
    if (field == SPH_F_RHO) return c->rho_valid || c->rates_valid;
    if (field == SPH_F_P || field == SPH_F_C) return c->eos_valid || c->rates_valid;
    return c->rates_valid;
}

}  // namespace

namespace sph {

int ctx_alloc_bytes(sph_ctx *c, void **p, size_t bytes, const char *what) {
    *p = nullptr;
    if (bytes == 0) bytes = 8;
    if (hipMalloc(p, bytes) != hipSuccess) {
        (void)hipGetLastError();
        *p = nullptr;
        c->err = std::string("hipMalloc failed: ") + what;
        return SPH_ERR_NOMEM;
    }
    c->allocs[*p] = bytes;
    c->device_bytes += (int64_t)bytes;
    return SPH_OK;
}

void ctx_free_ptr(sph_ctx *c, void *p) {
    if (!p) return;
    auto it = c->allocs.find(p);
    if (it != c->allocs.end()) { c->device_bytes -= (int64_t)it->second; c->allocs.erase(it); }
    (void)hipFree(p);
}

}  // namespace sph

extern "C" {

int sph_abi_version(void) { return SPH_ABI_VERSION; }

int sph_get_params(const sph_ctx *c, sph_params *out) {
    if (!c || !out) return SPH_ERR_ARG;
    *out = c->p;
    return SPH_OK;
}

const char *sph_strerror(int s) {
    switch (s) {
        case SPH_OK: return "ok";
        case SPH_ERR_ARG: return "invalid argument";
        case SPH_ERR_NO_DEVICE: return "no usable HIP device";
        case SPH_ERR_HIP: return "HIP runtime error";
        case SPH_ERR_NOMEM: return "out of memory";
        case SPH_ERR_STATE: return "call order violated";
        case SPH_ERR_GRID: return "cell grid too large";
        case SPH_ERR_NONFINITE: return "non-finite particle position";
        default: return "unknown status";
    }
}

const char *sph_last_error(const sph_ctx *c) { return c ? c->err.c_str() : "null context"; }

int sph_params_default(sph_params *p) {
    if (!p) return SPH_ERR_ARG;
    std::memset(p, 0, sizeof(*p));
    p->h = 2.5;                               // SUMMER_SPH.f90:11
    p->gamma = 1.4; p->gamma_m1 = 0.4;        // :465-466
    p->nq = 5000;                             // :8
    p->flags = 0;
    p->kernel_pi = 3.14159265359;             // :125-126
    p->visc_eps = (double)0.01f;              // :373  (REAL(4) literal)
    p->alpha_floor = 0.1;                     // :317
    p->alpha_decay = (double)0.15f;           // :317  (REAL(4) literal)
    p->G = (double)39.47841760435743f;        // :7    (REAL(4) literal)
    p->dt_scale = 0.25;                       // :851
    p->dt_max = (double)0.1f;                 // :855
    p->dt_min = (double)0.0001f;              // :857
    p->bounding_size = 1500.0;                // :11
    p->eta = 1.2; p->h_tol = 1e-3; p->h_max_length = 10.0;     // variable-h only (no reference defaults exist)
    p->h_min_length = (double)0.01f;          // Variable.f90:528
    p->h_iter_cap = 10.0;                     // Variable.f90:529
    p->theta = 0.5;                           // :825
    return SPH_OK;
}

int sph_params_default_variable(sph_params *p) {
    if (!p) return SPH_ERR_ARG;
    sph_params_default(p);
    p->flags = SPH_FLAG_VARIABLE_H;
    p->nq = 2500;                                        // Variable.f90:8
    p->kernel_pi = (double)3.1415926535897932f;          // Variable.f90:7 (REAL(4) literal)
    p->gamma = 1.4; p->gamma_m1 = p->gamma - 1.0;        // Variable.f90:509 computes gamma - 1.0_dp
    return SPH_OK;
}

int sph_ctx_create(const sph_params *p, int device, sph_ctx **out) {
    if (!out) return SPH_ERR_ARG;
    *out = nullptr;
    sph_params dp;
    if (!p) { sph_params_default(&dp); p = &dp; }
    if (!(p->h > 0.0) || p->nq < 2 || p->nq > 19000 || !(p->gamma > 0.0)) return SPH_ERR_ARG;   // table must fit LDS
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { (void)hipGetLastError(); return SPH_ERR_NO_DEVICE; }
    if (device < 0 || device >= ndev) return SPH_ERR_NO_DEVICE;
    sph_ctx *c = new (std::nothrow) sph_ctx();
    if (!c) return SPH_ERR_NOMEM;
    c->p = *p;
    c->variable = (p->flags & SPH_FLAG_VARIABLE_H) != 0;
    c->gravity = (p->flags & SPH_FLAG_SELF_GRAVITY) != 0;
    c->tiled = !c->variable && (p->flags & SPH_FLAG_NO_LDS_TILES) == 0;
    // whole-tile kernels: 116 KB tile + the kernel table must fit the 160 KB of LDS (nq <= ~5500)
    c->whole_tile = c->tiled && (p->flags & SPH_FLAG_NO_WHOLE_TILE) == 0 &&
                    (size_t)TAB_LDS(p->nq) * sizeof(double) + (size_t)WT_TILE_RECORDS * 32 + 1024 <= (size_t)160 * 1024;
    c->packed_list = c->tiled;
    c->device = device;
    DeviceGuard g(device);
    { int cus = 0; if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && cus > 0) c->num_cus = cus; }
    int st = SPH_OK;
    auto fail = [&](int s) { sph_ctx_destroy(c); return s; };
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) return fail(SPH_ERR_HIP);
    for (int k = 0; k < 2; k++)
        if (hipEventCreateWithFlags(&c->ev_bbox[k], hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&c->ev_nl[k], hipEventDisableTiming) != hipSuccess) return fail(SPH_ERR_HIP);
    c->no_stale = getenv("SPH_SYNC_EVERY_BUILD") != nullptr;
    if (hipHostMalloc(reinterpret_cast<void **>(&c->h_pinned), 640 * sizeof(double), hipHostMallocDefault) != hipSuccess) return fail(SPH_ERR_NOMEM);
    std::memset(c->h_pinned, 0, 640 * sizeof(double));
    if ((st = ctx_alloc(c, &c->bbox_part, (size_t)1024 * 8 + 64, "bbox")) != SPH_OK) return fail(st);
    if ((st = ctx_alloc(c, &c->d_flags, 8, "flags")) != SPH_OK) return fail(st);
    if (hipMemset(c->d_flags, 0, 8 * sizeof(int32_t)) != hipSuccess) return fail(SPH_ERR_HIP);
    if ((st = ctx_alloc(c, &c->w_tab, (size_t)TAB_LEN(p->nq), "W table")) != SPH_OK) return fail(st);
    if ((st = ctx_alloc(c, &c->dw_tab, (size_t)TAB_LEN(p->nq), "dW table")) != SPH_OK) return fail(st);
    if ((st = ctx_alloc(c, &c->grav_tab, (size_t)p->nq + 1, "softening table")) != SPH_OK) return fail(st);
    if ((st = ctx_alloc(c, &c->w_pair, (size_t)2 * p->nq, "W pair table")) != SPH_OK) return fail(st);
    if ((st = ctx_alloc(c, &c->dw_pair, (size_t)2 * p->nq, "dW pair table")) != SPH_OK) return fail(st);
    if ((st = ctx_alloc(c, &c->sink, (size_t)10 * MAX_SINKS, "sinks")) != SPH_OK) return fail(st);
    if ((st = ctx_alloc(c, &c->sink_radius, (size_t)MAX_SINKS, "sink radii")) != SPH_OK) return fail(st);
    c->sink_blocks = 512;
    if ((st = ctx_alloc(c, &c->sink_part, (size_t)c->sink_blocks * MAX_SINKS * 3, "sink partials")) != SPH_OK) return fail(st);
    c->dt_blocks = 1024;
    if ((st = ctx_alloc(c, &c->dt_part, (size_t)c->dt_blocks, "dt partials")) != SPH_OK) return fail(st);
    if ((st = ctx_alloc(c, &c->d_dt, 4, "dt")) != SPH_OK) return fail(st);
    std::vector<double> w, dw;
    host_tables(p->nq, w, dw);
    {
        std::vector<double> gt;
        host_grav_table(p->nq, gt);
        if (hipMemcpy(c->grav_tab, gt.data(), gt.size() * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) return fail(SPH_ERR_HIP);
    }
    std::vector<double> wp((size_t)2 * p->nq), dwp((size_t)2 * p->nq);
    for (int k = 0; k < p->nq; k++) { wp[2 * k] = w[k]; wp[2 * k + 1] = w[k + 1]; dwp[2 * k] = dw[k]; dwp[2 * k + 1] = dw[k + 1]; }
    if (hipMemcpy(c->w_pair, wp.data(), wp.size() * sizeof(double), hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(c->dw_pair, dwp.data(), dwp.size() * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) return fail(SPH_ERR_HIP);
    if (hipMemcpy(c->w_tab, w.data(), w.size() * sizeof(double), hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(c->dw_tab, dw.data(), dw.size() * sizeof(double), hipMemcpyHostToDevice) != hipSuccess ||
        hipMemset(c->sink, 0, sizeof(double) * 10 * MAX_SINKS) != hipSuccess ||
        hipMemset(c->d_dt, 0, sizeof(double) * 4) != hipSuccess)
        return fail(SPH_ERR_HIP);
    *out = c;
    return SPH_OK;
}

int sph_ctx_destroy(sph_ctx *c) {
    if (!c) return SPH_OK;
    DeviceGuard g(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    resolve_timing(c);
    free_particle_arrays(c);
    ctx_free(c, c->cell_start); c->cell_fill = nullptr; ctx_free(c, c->cell_hmax); ctx_free(c, c->bbox_part); ctx_free(c, c->d_flags);
    ctx_free(c, c->grav_tab); ctx_free(c, c->sink_radius);
    ctx_free(c, c->w_tab); ctx_free(c, c->dw_tab); ctx_free(c, c->w_pair); ctx_free(c, c->dw_pair); ctx_free(c, c->sink); ctx_free(c, c->sink_part);
    ctx_free(c, c->dt_part); ctx_free(c, c->d_dt);
    if (c->h_pinned) (void)hipHostFree(c->h_pinned);
    for (int k = 0; k < 2; k++) { if (c->ev_bbox[k]) (void)hipEventDestroy(c->ev_bbox[k]); if (c->ev_nl[k]) (void)hipEventDestroy(c->ev_nl[k]); }
    ctx_free(c, c->sel_count); ctx_free_ptr(c, c->sel_tmp); ctx_free(c, c->bnd_boxes);
    if (c->stream && c->own_stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return SPH_OK;
}

int64_t sph_count(const sph_ctx *c) { return c ? c->n : -1; }
int32_t sph_sink_count(const sph_ctx *c) { return c ? c->ns : -1; }

int sph_check_sink_creation(sph_ctx *c, int32_t *created) {
    if (!c) return SPH_ERR_ARG;
    DeviceGuard g(c->device);
    int32_t cr = 0;
    const int st = sink_creation(c, &cr);
    if (created) *created = cr;
    return st;
}

static int upload_impl(sph_ctx *c, int64_t n, const double *const src[9], hipMemcpyKind kind) {
    if (!c) return SPH_ERR_ARG;
    if (n < 0 || n > (c->variable ? 500000000LL : 2000000000LL)) { c->err = "sph_upload: bad n"; return SPH_ERR_ARG; }
    for (int k = 0; k < 8; k++)
        if (n > 0 && !src[k]) { c->err = "sph_upload: null array"; return SPH_ERR_ARG; }
    DeviceGuard g(c->device);
    API_TRY(ensure_capacity(c, n));
    c->n = n; c->n_slots = n; c->dead_below = 0;
    c->numbers_set = false;
    c->n_owned = n;
    for (int k = 0; k < 9; k++) {
        if (n == 0) break;
        if (src[k]) API_HIP(hipMemcpyAsync(c->f[k], src[k], (size_t)n * sizeof(double), kind, c->stream));
        else API_HIP(hipMemsetAsync(c->f[k], 0, (size_t)n * sizeof(double), c->stream));   // alpha = 0, SUMMER_SPH.f90:681
    }
    API_HIP(launch_iota(c, c->orig, n));
    API_HIP(launch_iota(c, c->inv, n));
    if (c->variable) API_HIP(launch_fill(c, c->f[SPH_F_H], c->p.h, n));    // until sph_upload_field(SPH_F_H) sets it
    API_HIP(hipStreamSynchronize(c->stream));
    c->grid_valid = c->rho_valid = c->eos_valid = c->rates_valid = c->order_valid = c->tree_valid = false;
    c->derived_kept = false;
    c->ring_bbox_valid = c->ring_nl_valid = false;       // a new particle set: the next build waits for its own read-backs
    c->h_new_is_build = false; c->h_refresh_ok = false;
    return SPH_OK;
}

int sph_upload(sph_ctx *c, int64_t n, const double *x, const double *y, const double *z, const double *vx,
               const double *vy, const double *vz, const double *u, const double *m, const double *alpha) {
    const double *src[9] = {x, y, z, vx, vy, vz, u, m, alpha};
    return upload_impl(c, n, src, hipMemcpyHostToDevice);
}

int sph_upload_dev(sph_ctx *c, int64_t n, const double *x, const double *y, const double *z, const double *vx,
                   const double *vy, const double *vz, const double *u, const double *m, const double *alpha) {
    const double *src[9] = {x, y, z, vx, vy, vz, u, m, alpha};
    return upload_impl(c, n, src, hipMemcpyDeviceToDevice);
}

int sph_set_sinks(sph_ctx *c, int32_t ns, const double *sx, const double *sy, const double *sz, const double *svx,
                  const double *svy, const double *svz, const double *sm) {
    if (!c) return SPH_ERR_ARG;
    if (ns < 0 || ns > MAX_SINKS) { c->err = "sph_set_sinks: 0 <= ns <= 64"; return SPH_ERR_ARG; }
    const double *src[7] = {sx, sy, sz, svx, svy, svz, sm};
    for (auto p : src) if (ns > 0 && !p) { c->err = "sph_set_sinks: null array"; return SPH_ERR_ARG; }
    DeviceGuard g(c->device);
    std::vector<double> buf((size_t)10 * MAX_SINKS, 0.0);
    for (int k = 0; k < 7; k++) for (int s = 0; s < ns; s++) buf[(size_t)k * MAX_SINKS + s] = src[k][s];
    API_HIP(hipStreamSynchronize(c->stream));
    API_HIP(hipMemcpy(c->sink, buf.data(), buf.size() * sizeof(double), hipMemcpyHostToDevice));
    std::vector<double> rad((size_t)MAX_SINKS, c->variable ? 5.0 : 3.5);      // Variable.f90:830 / SUMMER_SPH.f90:694
    API_HIP(hipMemcpy(c->sink_radius, rad.data(), rad.size() * sizeof(double), hipMemcpyHostToDevice));
    c->ns = ns;
    c->rates_valid = false;
    return SPH_OK;
}

int sph_get_sinks(sph_ctx *c, int32_t ns, double *sx, double *sy, double *sz, double *svx, double *svy, double *svz,
                  double *sm, double *sax, double *say, double *saz) {
    if (!c || ns < 0 || ns > c->ns) return SPH_ERR_ARG;
    DeviceGuard g(c->device);
    std::vector<double> buf((size_t)10 * MAX_SINKS);
    API_HIP(hipStreamSynchronize(c->stream));
    API_HIP(hipMemcpy(buf.data(), c->sink, buf.size() * sizeof(double), hipMemcpyDeviceToHost));
    double *dst[10] = {sx, sy, sz, svx, svy, svz, sm, sax, say, saz};
    for (int k = 0; k < 10; k++) if (dst[k]) for (int s = 0; s < ns; s++) dst[k][s] = buf[(size_t)k * MAX_SINKS + s];
    return SPH_OK;
}

int sph_density(sph_ctx *c) { if (!c) return SPH_ERR_ARG; DeviceGuard g(c->device); return do_density(c); }
int sph_forces(sph_ctx *c) { if (!c) return SPH_ERR_ARG; DeviceGuard g(c->device); return do_forces(c); }
int sph_kick(sph_ctx *c, double dt) { if (!c) return SPH_ERR_ARG; DeviceGuard g(c->device); return do_kick(c, dt, false); }
int sph_drift(sph_ctx *c, double dt) { if (!c) return SPH_ERR_ARG; DeviceGuard g(c->device); return do_drift(c, dt, false); }

int sph_next_dt(sph_ctx *c, double *dt) {
    if (!c || !dt) return SPH_ERR_ARG;
    if (!c->rates_valid) { c->err = "sph_next_dt: rates are stale"; return SPH_ERR_STATE; }
    DeviceGuard g(c->device);
    API_TRY(put_dt(c, *dt, 0.0));
    { Timed t(c, SPH_K_DT); API_HIP(launch_next_dt(c, false)); }
    return get_dt(c, dt, nullptr);
}

int sph_run(sph_ctx *c, int32_t nsteps, double *dt, double *t) {
    if (!c || !dt || nsteps < 0) return SPH_ERR_ARG;
    DeviceGuard g(c->device);
    API_TRY(put_dt(c, *dt, t ? *t : 0.0));
    for (int k = 0; k < nsteps; k++) API_TRY(one_step_device_dt(c));
    return get_dt(c, dt, t);
}

int sph_step(sph_ctx *c, double *dt, double *t) { return sph_run(c, 1, dt, t); }

int sph_download_field_dev(sph_ctx *c, int field, double *d_out, int64_t n) {
    if (!c || field < 0 || field >= SPH_F_COUNT || n != c->n || (n > 0 && !d_out)) return SPH_ERR_ARG;
    if (!field_ready(c, field)) { c->err = "sph_download_field: field is stale"; return SPH_ERR_STATE; }
    DeviceGuard g(c->device);
    API_HIP(launch_unpermute(c, c->f[field], d_out));
    API_HIP(hipStreamSynchronize(c->stream));
    return drain_reports(c);
}

int sph_download_field(sph_ctx *c, int field, double *host, int64_t n) {
    if (!c || field < 0 || field >= SPH_F_COUNT || n != c->n || (n > 0 && !host)) return SPH_ERR_ARG;
    if (!field_ready(c, field)) { c->err = "sph_download_field: field is stale"; return SPH_ERR_STATE; }
    if (n == 0) return SPH_OK;
    DeviceGuard g(c->device);
    API_HIP(launch_unpermute(c, c->f[field], c->scratch));
    API_HIP(hipMemcpyAsync(host, c->scratch, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    API_HIP(hipStreamSynchronize(c->stream));
    return drain_reports(c);
}

int sph_download_state(sph_ctx *c, int64_t n, double *x, double *y, double *z, double *vx, double *vy, double *vz,
                       double *u, double *m, double *alpha) {
    double *dst[9] = {x, y, z, vx, vy, vz, u, m, alpha};
    for (int k = 0; k < 9; k++)
        if (dst[k]) API_TRY(sph_download_field(c, k, dst[k], n));
    return SPH_OK;
}

int sph_get_stats(sph_ctx *c, sph_stats *o) {
    if (!c || !o) return SPH_ERR_ARG;
    std::memset(o, 0, sizeof(*o));
    o->n = c->n; o->n_cells = c->grid.ncells;
    for (int a = 0; a < 3; a++) o->grid_dim[a] = c->grid.dim[a];
    o->nlist_capacity = c->nl_cap; o->nlist_max = c->nl_max;
    o->tile_fit_pct = c->whole_tile ? c->wt_fit_pct : -1;
    o->tile_fit_pct_forces = c->whole_tile ? c->wt_fit_pct_f : -1;
    o->host_syncs = (int32_t)c->host_syncs;
    o->grid_builds = c->grid_builds; o->nlist_builds = c->nlist_builds;
    o->density_passes = c->density_passes; o->force_passes = c->force_passes;
    o->device_bytes = c->device_bytes;
    o->nlist_reflags = c->nlist_reflags;
    if (c->n > 0 && c->nlist_builds > 0 && c->ncount && c->n <= c->cap) {      // counts of the last build
        DeviceGuard g(c->device);
        std::vector<int32_t> cnt((size_t)c->n);
        API_HIP(hipStreamSynchronize(c->stream));
        API_TRY(drain_reports(c));
        API_HIP(hipMemcpy(cnt.data(), c->ncount, cnt.size() * sizeof(int32_t), hipMemcpyDeviceToHost));
        double s = 0.0, sw = 0.0;
        for (int32_t v : cnt) s += v;
        o->nlist_mean = s / (double)c->n;
        for (size_t w = 0; w < cnt.size(); w += 64) {
            int32_t m = 0;
            for (size_t k = w; k < std::min(cnt.size(), w + 64); k++) m = std::max(m, cnt[k]);
            sw += m;
        }
        o->nlist_wave_mean = sw / (double)((cnt.size() + 63) / 64);
        o->lane_efficiency_forces = (c->whole_tile && c->wt_ok_f) ? forces_lane_efficiency(cnt) : (sw > 0.0 ? s / (64.0 * sw) : 0.0);
    }
    return SPH_OK;
}

// ---- multi-GPU building blocks (see summersph_amd/dist.py) ------------------------------------

static int upload_field_impl(sph_ctx *c, int field, const double *src, int64_t n, bool host) {
    if (!c || field < 0 || field >= SPH_F_COUNT || n != c->n || (n > 0 && !src)) return SPH_ERR_ARG;
    if ((field == SPH_F_H || field == SPH_F_OMEGA) && !c->variable) { c->err = "field needs SPH_FLAG_VARIABLE_H"; return SPH_ERR_ARG; }
    if (n == 0) return SPH_OK;
    DeviceGuard g(c->device);
    const double *dsrc = src;
    if (host) {
        API_HIP(hipMemcpyAsync(c->scratch, src, (size_t)n * sizeof(double), hipMemcpyHostToDevice, c->stream));
        dsrc = c->scratch;
    }
    const int f = field;
    API_HIP(launch_scatter_fields(c, 1, &f, 0, n, dsrc));
    API_HIP(hipStreamSynchronize(c->stream));
    field_written(c, field);
    return SPH_OK;
}

int sph_upload_field(sph_ctx *c, int field, const double *host, int64_t n) { return upload_field_impl(c, field, host, n, true); }
int sph_upload_field_dev(sph_ctx *c, int field, const double *d_vals, int64_t n) { return upload_field_impl(c, field, d_vals, n, false); }

int sph_set_sink_radii(sph_ctx *c, int32_t ns, const double *radius) {
    if (!c || ns != c->ns || (ns > 0 && !radius)) return SPH_ERR_ARG;
    DeviceGuard g(c->device);
    API_HIP(hipStreamSynchronize(c->stream));
    if (ns > 0) API_HIP(hipMemcpy(c->sink_radius, radius, (size_t)ns * sizeof(double), hipMemcpyHostToDevice));
    return SPH_OK;
}

int sph_get_sink_radii(sph_ctx *c, int32_t ns, double *radius) {
    if (!c || ns < 0 || ns > c->ns || (ns > 0 && !radius)) return SPH_ERR_ARG;
    DeviceGuard g(c->device);
    API_HIP(hipStreamSynchronize(c->stream));
    if (ns > 0) API_HIP(hipMemcpy(radius, c->sink_radius, (size_t)ns * sizeof(double), hipMemcpyDeviceToHost));
    return SPH_OK;
}

int sph_accrete_and_cull(sph_ctx *c, int64_t *n_removed) {
    if (!c) return SPH_ERR_ARG;
    DeviceGuard g(c->device);
    int64_t r = 0;
    const int st = do_accrete(c, &r);
    if (n_removed) *n_removed = r;
    return st;
}

int sph_accrete_and_cull_keep(sph_ctx *c, int32_t *d_keep, int64_t *n_removed) {
    if (!c || (c->n > 0 && !d_keep)) return SPH_ERR_ARG;
    DeviceGuard g(c->device);
    int64_t r = 0;
    const int st = do_accrete(c, &r, d_keep);
    if (n_removed) *n_removed = r;
    return st;
}

int sph_update_h(sph_ctx *c) { if (!c) return SPH_ERR_ARG; DeviceGuard g(c->device); return do_update_h(c); }

int sph_set_owned(sph_ctx *c, int64_t n_owned) {
    if (!c || n_owned < 0 || n_owned > c->n) return SPH_ERR_ARG;
    c->n_owned = n_owned;
    c->h_refresh_ok = false;
    c->grid_valid = c->rho_valid = c->eos_valid = c->rates_valid = false;
    return SPH_OK;
}

int sph_set_rank(sph_ctx *c, int32_t rank, int32_t nranks) {
    if (!c || nranks < 1 || rank < 0 || rank >= nranks) return SPH_ERR_ARG;
    c->rank = rank; c->nranks = nranks;
    return SPH_OK;
}

int sph_scatter_field_dev(sph_ctx *c, int field, int64_t first, int64_t count, const double *d_vals) {
    if (!c || field < 0 || field >= SPH_F_COUNT || first < 0 || count < 0 || first + count > c->n || (count > 0 && !d_vals))
        return SPH_ERR_ARG;
    DeviceGuard g(c->device);
    API_HIP(launch_scatter_field(c, c->f[field], first, count, d_vals));
    if ((field == SPH_F_H || field == SPH_F_OMEGA) && !c->variable) { c->err = "field needs SPH_FLAG_VARIABLE_H"; return SPH_ERR_ARG; }
    field_written(c, field);
    return SPH_OK;
}

static bool fields_ok(const sph_ctx *c, int nf, const int *fields, bool for_read) {
    if (nf < 1 || nf > SPH_F_COUNT || !fields) return false;
    for (int f = 0; f < nf; f++) {
        if (fields[f] < 0 || fields[f] >= SPH_F_COUNT) return false;
        if (for_read && !field_ready(c, fields[f])) return false;
    }
    return true;
}

int sph_gather_fields_dev(sph_ctx *c, int32_t nf, const int32_t *fields, int64_t count, const int64_t *d_ids, double *d_out) {
    if (!c || count < 0 || count > c->n || (count > 0 && !d_out)) return SPH_ERR_ARG;
    if (!fields_ok(c, nf, fields, true)) { c->err = "sph_gather_fields_dev: bad or stale field"; return SPH_ERR_ARG; }
    DeviceGuard g(c->device);
    API_HIP(launch_gather_fields(c, nf, fields, d_ids, count, d_out));
    if (c->own_stream) API_HIP(hipStreamSynchronize(c->stream));
    return SPH_OK;
}

int sph_scatter_fields_dev(sph_ctx *c, int32_t nf, const int32_t *fields, int64_t first, int64_t count, const double *d_vals) {
    if (!c || first < 0 || count < 0 || first + count > c->n || (count > 0 && !d_vals)) return SPH_ERR_ARG;
    if (!fields_ok(c, nf, fields, false)) return SPH_ERR_ARG;
    DeviceGuard g(c->device);
    API_HIP(launch_scatter_fields(c, nf, fields, first, count, d_vals));
    for (int f = 0; f < nf; f++) field_written(c, fields[f]);
    return SPH_OK;
}

static int refresh_eos_impl(sph_ctx *c, bool ghosts_only) {
    if (!c) return SPH_ERR_ARG;
    if (!c->grid_valid || !c->rho_valid) { c->err = "sph_refresh_eos: density is stale"; return SPH_ERR_STATE; }
    DeviceGuard g(c->device);
    const PairConst pc = make_pair_const(c);
    Timed t(c, SPH_K_DENSITY);
    if (c->variable) API_HIP(launch_eos_only_v(c, pc));
    else API_HIP(launch_eos_only(c, pc, ghosts_only));
    c->eos_valid = true;
    return SPH_OK;
}

int sph_refresh_eos(sph_ctx *c) { return refresh_eos_impl(c, false); }
int sph_refresh_eos_ghosts(sph_ctx *c) { return refresh_eos_impl(c, true); }

int sph_dt_candidate(sph_ctx *c, double *cand) {
    if (!c || !cand) return SPH_ERR_ARG;
    if (!c->rates_valid) { c->err = "sph_dt_candidate: rates are stale"; return SPH_ERR_STATE; }
    DeviceGuard g(c->device);
    { Timed t(c, SPH_K_DT); API_HIP(launch_dt_partial_only(c)); }
    API_HIP(hipMemcpyAsync(c->h_pinned + 24, c->d_dt + 2, sizeof(double), hipMemcpyDeviceToHost, c->stream));
    API_HIP(hipStreamSynchronize(c->stream));
    *cand = c->h_pinned[24];
    return SPH_OK;
}

int sph_set_sink_accel(sph_ctx *c, int32_t ns, const double *sax, const double *say, const double *saz) {
    if (!c || ns != c->ns || (ns > 0 && (!sax || !say || !saz))) return SPH_ERR_ARG;
    DeviceGuard g(c->device);
    API_HIP(hipStreamSynchronize(c->stream));
    const double *src[3] = {sax, say, saz};
    for (int k = 0; k < 3; k++)
        if (ns > 0) API_HIP(hipMemcpy(c->sink + (size_t)(7 + k) * MAX_SINKS, src[k], (size_t)ns * sizeof(double), hipMemcpyHostToDevice));
    return SPH_OK;
}

int sph_set_stream(sph_ctx *c, void *stream) {
    if (!c) return SPH_ERR_ARG;
    DeviceGuard g(c->device);
    API_HIP(hipStreamSynchronize(c->stream));
    resolve_timing(c);
    if (c->own_stream) { (void)hipStreamDestroy(c->stream); c->stream = nullptr; }
    c->stream = reinterpret_cast<hipStream_t>(stream);      // NULL = the device's default stream
    c->own_stream = false;
    return SPH_OK;
}

int sph_reserve(sph_ctx *c, int64_t n_slots) {
    if (!c || n_slots < 0) return SPH_ERR_ARG;
    c->reserve = n_slots;
    return SPH_OK;
}

int sph_owned_bbox(sph_ctx *c, double *lo_hi, double *d_lo_hi) {
    if (!c || (!lo_hi && !d_lo_hi)) return SPH_ERR_ARG;
    DeviceGuard g(c->device);
    return owned_bbox(c, d_lo_hi, lo_hi);
}

int sph_select_boxes(sph_ctx *c, int32_t nbox, const double *boxes, int64_t *counts) {
    if (!c || nbox < 0 || nbox > MAX_SEL_BOXES || (nbox > 0 && (!boxes || !counts))) return SPH_ERR_ARG;
    DeviceGuard g(c->device);
    c->sel_boxes = 0;
    API_TRY(domain_select_boxes(c, nbox, boxes, counts));
    c->sel_boxes = nbox;
    for (int b = 0; b < nbox; b++) c->sel_counts[b] = counts[b];
    return SPH_OK;
}

int sph_select_boxes_async(sph_ctx *c, int32_t nbox, const double *boxes) {
    if (!c || nbox < 0 || nbox > MAX_SEL_BOXES || (nbox > 0 && !boxes)) return SPH_ERR_ARG;
    DeviceGuard g(c->device);
    c->sel_boxes = 0;
    API_TRY(domain_select_boxes_enqueue(c, nbox, boxes));
    c->sel_boxes = nbox;
    for (int b = 0; b < nbox; b++) c->sel_counts[b] = -1;          // unknown until sph_selected_counts
    return SPH_OK;
}

int sph_selected_counts(sph_ctx *c, int32_t nbox, int64_t *counts) {
    if (!c || nbox != c->sel_boxes || (nbox > 0 && !counts)) return SPH_ERR_ARG;
    domain_selected_counts(c, nbox, counts);
    for (int b = 0; b < nbox; b++) c->sel_counts[b] = counts[b];
    return SPH_OK;
}

int sph_gather_selected_dev(sph_ctx *c, int32_t box, int32_t nf, const int32_t *fields, int64_t capacity, double *d_out) {
    if (!c || box < 0 || box >= c->sel_boxes || capacity < 0 || !d_out) return SPH_ERR_ARG;
    if (!fields_ok(c, nf, fields, true)) { c->err = "sph_gather_selected_dev: bad or stale field"; return SPH_ERR_ARG; }
    DeviceGuard g(c->device);
    if (c->n_owned == 0) {            // nothing selected, no id list: the header alone
        API_HIP(hipMemsetAsync(d_out, 0, 2 * sizeof(double), c->stream));
        return SPH_OK;
    }
    API_HIP(launch_gather_selected(c, nf, fields, box, capacity, d_out));
    return SPH_OK;
}

int sph_selected_ids_dev(sph_ctx *c, int32_t box, int64_t count, int64_t *d_ids) {
    if (!c || box < 0 || box >= c->sel_boxes || count != c->sel_counts[box] || (count > 0 && !d_ids)) return SPH_ERR_ARG;
    if (count == 0) return SPH_OK;
    DeviceGuard g(c->device);
    API_HIP(hipMemcpyAsync(d_ids, c->sel_ids + (size_t)box * c->sel_stride, (size_t)count * sizeof(int64_t), hipMemcpyDeviceToDevice, c->stream));
    if (c->own_stream) API_HIP(hipStreamSynchronize(c->stream));
    return SPH_OK;
}

int sph_replace_ghosts_dev(sph_ctx *c, int64_t count, const double *d_state) {
    if (!c || count < 0 || (count > 0 && !d_state)) return SPH_ERR_ARG;
    if (c->dead_below > 0) { c->err = "sph_replace_ghosts_dev: a ghost swap is already pending (call sph_density)"; return SPH_ERR_STATE; }
    DeviceGuard g(c->device);
    API_TRY(domain_replace_ghosts(c, count, d_state));
    if (c->own_stream) API_HIP(hipStreamSynchronize(c->stream));
    c->grid_valid = c->rho_valid = c->eos_valid = c->rates_valid = c->order_valid = c->tree_valid = false;
    return SPH_OK;
}

int sph_set_boundary_boxes(sph_ctx *c, int32_t nbox, const double *boxes) {
    if (!c || nbox < 0 || nbox > MAX_SEL_BOXES || (nbox > 0 && !boxes)) return SPH_ERR_ARG;
    DeviceGuard g(c->device);
    if (!c->bnd_boxes) API_TRY(ctx_alloc(c, &c->bnd_boxes, (size_t)6 * MAX_SEL_BOXES, "boundary boxes"));
    for (int k = 0; k < 6 * nbox; k++) c->h_pinned[128 + k] = boxes[k];      // pinned staging: no host synchronisation
    if (nbox > 0) API_HIP(hipMemcpyAsync(c->bnd_boxes, c->h_pinned + 128, (size_t)6 * nbox * sizeof(double), hipMemcpyHostToDevice, c->stream));
    c->n_bnd_boxes = nbox;
    c->wave_class_valid = false;
    return SPH_OK;
}

int sph_forces_part(sph_ctx *c, int32_t part) {
    if (!c || (part != 1 && part != 2)) return SPH_ERR_ARG;
    DeviceGuard g(c->device);
    return do_forces_part(c, part);
}

int sph_set_gravity_sources_dev(sph_ctx *c, int64_t n_src, const double *d_xyzm, const double *lo_hi) {
    if (!c || n_src < 0 || (n_src > 0 && (!d_xyzm || !lo_hi))) return SPH_ERR_ARG;
    if (!c->gravity && !c->variable) { c->err = "sph_set_gravity_sources_dev: needs SPH_FLAG_SELF_GRAVITY or SPH_FLAG_VARIABLE_H"; return SPH_ERR_STATE; }
    c->gx_src = n_src > 0 ? d_xyzm : nullptr;
    c->gx_n = n_src;
    for (int a = 0; a < 6; a++) c->gx_box[a] = n_src > 0 ? lo_hi[a] : 0.0;
    c->gx_keys_valid = false;
    c->tree_valid = false; c->grav_valid = false;
    c->leaf_valid = false;
    c->rates_valid = false;
    if (c->variable) { c->grid_valid = false; c->rho_valid = false; c->eos_valid = false; }      // the leaf boxes change
    return SPH_OK;
}

int sph_accrete_mark_dev(sph_ctx *c, int64_t src_offset, double *d_partials) {
    if (!c || !d_partials) return SPH_ERR_ARG;
    DeviceGuard g(c->device);
    return accrete_mark_ext(c, src_offset, d_partials);
}

int sph_accrete_apply_dev(sph_ctx *c, const double *d_all, int32_t nranks, int32_t stride, int32_t *d_keep, int64_t *n_removed) {
    if (!c || !d_all || nranks < 1 || stride < 7 * MAX_SINKS) return SPH_ERR_ARG;
    DeviceGuard g(c->device);
    int64_t r = 0;
    const int st = accrete_apply_ext(c, d_all, nranks, stride, d_keep, &r);
    if (n_removed) *n_removed = r;
    return st;
}

int sph_set_numbers_dev(sph_ctx *c, int64_t first, int64_t count, const int64_t *d_numbers) {
    if (!c || first < 0 || count < 0 || first + count > c->cap || (count > 0 && !d_numbers)) return SPH_ERR_ARG;
    DeviceGuard g(c->device);
    API_HIP(launch_set_numbers(c, first, count, d_numbers));
    c->numbers_set = true;
    if (c->variable) { c->grid_valid = false; c->rho_valid = false; c->eos_valid = false; }
    return SPH_OK;
}

int sph_sink_candidate_dev(sph_ctx *c, double *d_cand) {
    if (!c || !d_cand) return SPH_ERR_ARG;
    DeviceGuard g(c->device);
    return sink_candidate(c, d_cand);
}

int sph_add_sink_checked_dev(sph_ctx *c, const double *d_cand, int32_t *created) {
    if (!c || !d_cand) return SPH_ERR_ARG;
    DeviceGuard g(c->device);
    int32_t cr = 0;
    const int st = sink_add_checked(c, d_cand, &cr);
    if (created) *created = cr;
    return st;
}

int sph_set_dt(sph_ctx *c, double dt, double t) {
    if (!c) return SPH_ERR_ARG;
    DeviceGuard g(c->device);
    return put_dt(c, dt, t);
}

int sph_get_dt(sph_ctx *c, double *dt, double *t) {
    if (!c) return SPH_ERR_ARG;
    DeviceGuard g(c->device);
    return get_dt(c, dt, t);
}

int sph_kick_devdt(sph_ctx *c) { if (!c) return SPH_ERR_ARG; DeviceGuard g(c->device); return do_kick(c, 0.0, true); }
int sph_drift_devdt(sph_ctx *c) { if (!c) return SPH_ERR_ARG; DeviceGuard g(c->device); return do_drift(c, 0.0, true); }

// kick + drift in one pass over the state (bitwise sph_kick_devdt + sph_drift_devdt; what sph_step itself launches)
int sph_kick_drift_devdt(sph_ctx *c) {
    if (!c) return SPH_ERR_ARG;
    if (!c->rates_valid) { c->err = "sph_kick_drift_devdt: rates are stale, call sph_forces first"; return SPH_ERR_STATE; }
    DeviceGuard g(c->device);
    Timed t(c, SPH_K_KICK);
    API_HIP(launch_kick_drift(c));
    c->grid_valid = false; c->rho_valid = false; c->eos_valid = false; c->order_valid = false;
    return SPH_OK;
}

// the closing kick + the local dt candidate in one pass (bitwise sph_kick_devdt + sph_dt_candidate_dev)
int sph_kick_dt_candidate_dev(sph_ctx *c) {
    if (!c) return SPH_ERR_ARG;
    if (!c->rates_valid) { c->err = "sph_kick_dt_candidate_dev: rates are stale, call sph_forces first"; return SPH_ERR_STATE; }
    DeviceGuard g(c->device);
    Timed t(c, SPH_K_DT);
    API_HIP(launch_kick_dt_candidate(c));
    c->eos_valid = false;
    return SPH_OK;
}

// the same for the gas alone, and the sinks' half kick by itself: between the two the sinks' accelerations may still be on
// their way through the rank reduction (bitwise sph_kick_dt_candidate_dev when called back to back)
int sph_kick_dt_candidate_gas_dev(sph_ctx *c) {
    if (!c) return SPH_ERR_ARG;
    if (!c->rates_valid) { c->err = "sph_kick_dt_candidate_gas_dev: rates are stale, call sph_forces first"; return SPH_ERR_STATE; }
    DeviceGuard g(c->device);
    Timed t(c, SPH_K_DT);
    API_HIP(launch_kick_dt_candidate(c, false));
    c->eos_valid = false;
    return SPH_OK;
}

int sph_kick_sinks_devdt(sph_ctx *c) {
    if (!c) return SPH_ERR_ARG;
    DeviceGuard g(c->device);
    API_HIP(launch_kick_sinks(c));
    return SPH_OK;
}

int sph_dt_candidate_dev(sph_ctx *c) {
    if (!c) return SPH_ERR_ARG;
    if (!c->rates_valid) { c->err = "sph_dt_candidate_dev: rates are stale"; return SPH_ERR_STATE; }
    DeviceGuard g(c->device);
    Timed t(c, SPH_K_DT);
    API_HIP(launch_dt_partial_only(c));
    return SPH_OK;
}

int sph_pack_partials_ex_dev(sph_ctx *c, double *d_out, int32_t predict_box) {
    if (!c || !d_out) return SPH_ERR_ARG;
    DeviceGuard g(c->device);
    API_HIP(launch_pack_partials(c, d_out, predict_box != 0));
    if (c->own_stream) API_HIP(hipStreamSynchronize(c->stream));
    return SPH_OK;
}

int sph_pack_partials_dev(sph_ctx *c, double *d_out) { return sph_pack_partials_ex_dev(c, d_out, 1); }

int sph_apply_partials_dev(sph_ctx *c, const double *d_all, int32_t nranks, int32_t stride, int32_t apply_dt) {
    if (!c || !d_all || nranks < 1 || stride < SPH_PARTIALS) return SPH_ERR_ARG;
    DeviceGuard g(c->device);
    API_HIP(launch_apply_partials(c, d_all, nranks, stride, apply_dt != 0));
    return SPH_OK;
}

int sph_get_bbox(sph_ctx *c, double *lo, double *hi) {
    if (!c || !lo || !hi) return SPH_ERR_ARG;
    if (!c->grid_valid) { c->err = "sph_get_bbox: no grid built for the current positions"; return SPH_ERR_STATE; }
    DeviceGuard g(c->device);
    API_HIP(hipStreamSynchronize(c->stream));             // the exact box of the last build sits in its read-back slot
    const double *bb = c->h_pinned + 200 + 16 * (1 - c->ring_bbox);
    for (int a = 0; a < 3; a++) { lo[a] = bb[a]; hi[a] = bb[3 + a]; }
    return SPH_OK;
}

int sph_timing_enable(sph_ctx *c, int on) {
    if (!c || on < 0) return SPH_ERR_ARG;
    c->timing = on == 1 ? 0xffffffffu : ((unsigned)on >> 1);       // 1: every group; else bit k + 1 = group k
    for (auto &s : c->timing_seen) s = 0;
    return SPH_OK;
}

int sph_timing_stride(sph_ctx *c, int stride) {
    if (!c || stride < 1) return SPH_ERR_ARG;
    c->timing_stride = stride;
    return SPH_OK;
}

int sph_timing_reset(sph_ctx *c) {
    if (!c) return SPH_ERR_ARG;
    DeviceGuard g(c->device);
    resolve_timing(c);
    for (auto &s : c->tslot) { s.total_ms = 0.0; s.launches = 0; }
    return SPH_OK;
}

int sph_timing_get(sph_ctx *c, int id, double *total_ms, int64_t *launches) {
    if (!c || id < 0 || id >= SPH_K_COUNT) return SPH_ERR_ARG;
    DeviceGuard g(c->device);
    resolve_timing(c);
    if (total_ms) *total_ms = c->tslot[id].total_ms;
    if (launches) *launches = c->tslot[id].launches;
    return SPH_OK;
}

int sph_synchronize(sph_ctx *c) {
    if (!c) return SPH_ERR_ARG;
    DeviceGuard g(c->device);
    API_HIP(hipStreamSynchronize(c->stream));
    return drain_reports(c);
}

void *sph_stream(sph_ctx *c) { return c ? (void *)c->stream : nullptr; }

}  // extern "C"
