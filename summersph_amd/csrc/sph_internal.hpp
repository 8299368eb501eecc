// sph_internal.hpp -- context layout and kernel-launcher declarations shared by the
// translation units of libsummersph_hip.so.  gfx950 (MI355X) only.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/summersph.h"

namespace sph {

// W / dW table length in doubles (one entry of zero padding behind the knot at q = 2, see table_lerp in pair_common.hpp) and its LDS footprint
constexpr int TAB_LEN(int nq) { return nq + 2; }
constexpr int TAB_LDS(int nq) { return (nq + 3) & ~1; }

constexpr int WAVE = 64;           // CDNA wavefront
constexpr int FREC = 12;           // doubles per force gather record
constexpr int MAX_SINKS = 64;      // sinks handled by the in-kernel loops
constexpr int MAX_SEL_BOXES = 64;  // boxes per sph_select_boxes call (multi-GPU ghost selection)
struct FieldPtrs9 { double *p[9]; };

// Uniform cell grid over the particles' bounding box.  Axes are permuted so that
// axis s[0] (the one with the fewest cells) varies fastest in the cell key: for a disc
// that is z, which keeps the three key-contiguous cells of a neighbour row short and the
// nine rows of a 27-cell stencil close together in the sorted particle order.
struct GridDesc {
    double org[3];      // bbox minimum, natural x,y,z order
    double inv_edge;    // 1 / cell edge, edge = 2h (1 + 1e-6)
    int32_t dim[3];     // cells along x,y,z
    int32_t s[3];       // axis permutation: s[0] fastest ... s[2] slowest
    int64_t ncells;
};

// Constants every pair kernel needs; passed by value as a kernel argument (SGPRs).
struct PairConst {
    double h;            // smoothing length
    double inv_h;        // 1 / h
    double inv_dq;       // nq / 2 exactly (dq = 2 / nq, [F]:58): scalar registers for the kernels, not vector ones computed per thread
    double dq;           // 2/nq
    double wnorm;        // kernel_pi * h^3        (W  is DIVIDED by this, [F]:125)
    double inv_dwnorm;   // 1 / (kernel_pi * h^4)  (dW is divided by kernel_pi h^4, [F]:126: once, in the epilogue)
    double visc_eps_h2;  // (0.01f * h) * h        ([F]:373)
    double alpha_floor, alpha_decay, G, gamma, gamma_m1;
    double rcut2;        // (2h)^2 (1 + 1e-12): filter only, the evaluation re-tests q <= 2
    int32_t nq;
    int32_t ns;          // sinks
    int32_t grav;        // 1: ax, ay, az already hold the self-gravity term ([F]:825), continue from there
    // variable-h path
    double kernel_pi, eta, h_tol, h_max_length, h_min_length, h_iter_cap, dt_scale;
};

struct TimingSlot {
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
    double total_ms = 0.0;
    int64_t launches = 0;
};

}  // namespace sph

struct sph_ctx {
    sph_params p;
    int device = 0;
    int num_cus = 256;
    hipStream_t stream = nullptr;
    std::string err;

    int64_t n = 0;        // particles held (owned + ghosts)
    int64_t n_owned = 0;  // original ids [0, n_owned) are this GPU's particles; [n_owned, n) are ghost
                          // copies of other GPUs' particles: neighbours only, never targets
    int64_t cap = 0;      // allocated particle slots
    int64_t reserve = 0;  // minimum capacity requested by sph_reserve (room for ghost swaps)
    // multi-GPU ghost swap (domain.hip): between sph_replace_ghosts_dev and the next grid build the arrays hold
    // n_slots > n entries; slots below dead_below whose original id is >= n_owned are the replaced ghosts
    int64_t n_slots = 0, dead_below = 0;
    int64_t *sel_ids = nullptr; size_t sel_cap = 0;      // box selection results, nbox x n_owned
    int64_t *sel_count = nullptr;
    void *sel_tmp = nullptr; size_t sel_tmp_bytes = 0;
    int32_t sel_boxes = 0; int64_t sel_counts[64] = {};  // last sph_select_boxes
    int64_t sel_stride = 0;                              // ids of box b start at sel_ids + b * sel_stride (n_owned at select time)
    // split force evaluation (sph_forces_part): waves near the other GPUs' boxes wait for the ghost fields
    int32_t *wave_class = nullptr; double *bnd_boxes = nullptr; int32_t n_bnd_boxes = 0;
    bool wave_class_valid = false, interior_done = false;
    bool own_stream = true;                              // false: the caller's stream (sph_set_stream), *_dev calls do not synchronise

    // cell-sorted struct-of-arrays state + derived + rates (SPH_F_* order)
    double *f[SPH_F_COUNT] = {};
    double *f_alt[10] = {};          // ping-pong targets for the state fields on reorder (9, +h when variable)
    int32_t *orig = nullptr, *orig_alt = nullptr;   // sorted slot -> original index
    int32_t *inv = nullptr;                         // original index -> sorted slot
    double *scratch = nullptr;       // n doubles (un-permute on download)

    // gather records (array-of-structs: one particle = one or few cache lines)
    double *drec = nullptr;          // 4 doubles: x y z m
    double *frec = nullptr;          // FREC doubles: x y z m | vx vy vz rho/2 | c/2 alpha/2 P/rho^2 h (pair_common.hpp)

    // variable-h path ("SUMMER_SPH - Variable.f90"): extra gather records and the octree leaf boxes
    bool variable = false;
    bool tiled = true;               // fixed-h: LDS-staged neighbour-list build (tiled.hip) unless SPH_FLAG_NO_LDS_TILES
    bool whole_tile = false;         // fixed-h: density/forces read the neighbours' {x,y,z,m} from one LDS tile per workgroup (tiled.hip)
    bool wt_ok = false, wt_ok_f = false;   // ... and the last list build found that the workgroups' intervals fit the tile (density / forces geometry)
    bool wt_half_f = false;                // forces_q on groups of 128 targets (eight lanes each): half the tile need of a group of 256
    bool wt_big = false, wt_big_f = false; // ... only the table-free tile (the kernel table's knots recomputed, 40 KB more for the tile)
    int32_t wt_fit_pct = -1, wt_fit_pct_f = -1;   // percentage of workgroups that fit (-1: kernels off)
    bool packed_list = true;         // list layout: 4-packed (tiled build) or wave-strided dwords (nlist_kernel)
    double *prec = nullptr;          // 4 doubles: x y z h        (neighbour-list build)
    double *lrec = nullptr;          // 4 doubles: leaf centre x y z, reach = 2h + leaf_edge/2 (<0: unresolved)
    uint64_t *mkeys = nullptr, *mkeys_alt = nullptr;   // octree path keys (3 bits per level, 21 levels)
    uint32_t *mvals = nullptr, *mvals_alt = nullptr;
    void *msort_tmp = nullptr; size_t msort_tmp_bytes = 0;
    double *cell_hmax = nullptr;     // per cell: largest h of its particles
    double *h_new = nullptr;         // scratch for calc_smoothing
    double *leaf_half = nullptr;     // half edge of every slot's leaf cell (reach = 2 h + this)
    bool path_keys_valid = false;    // mkeys_alt / mvals_alt hold the sorted octree path keys of the current grid build (root box: c->bbox)
    bool h_refresh_ok = false;       // the only thing newer than the grid is h (sph_update_h / an upload of h)
    bool h_new_is_build = false;     // h_new holds the lengths the list in place was built with (set by sph_update_h's swap; an
                                     // upload / scatter of h, a new particle set or a reallocation clears it: no re-flag then)
    bool leaf_valid = false;         // leaf cells match the current sorted order and (external) octree
    double h_max_glob = 0.0, h_mean = 0.0;
    double root_box[4] = {0, 0, 0, 0};   // octree root centre + edge ([V]:1007-1012)

    // Barnes-Hut gas self-gravity (gravity.hip): binary radix tree over the octree path keys
    bool gravity = false;
    bool tree_valid = false;
    double *g_cache[3] = {nullptr, nullptr, nullptr};   // SPH_FLAG_REUSE_GRAVITY: the self-gravity term of the last walk (sorted order)
    bool grav_valid = false;                            // ... still that of the current positions, masses, h, tree and sorted order
    int32_t *g_left = nullptr, *g_right = nullptr, *g_parent = nullptr, *g_leaf_parent = nullptr, *g_prefix = nullptr;
    int32_t *g_flag = nullptr, *g_slot = nullptr, *g_lvl = nullptr, *g_rope = nullptr, *g_leaf_rope = nullptr;
    double *g_sum = nullptr, *g_leafA = nullptr;     // 4 doubles per node / leaf
    int32_t *g_leaf_of = nullptr;                    // cell-sorted slot -> leaf index
    double *g_wrec = nullptr;                        // 64-byte walk records, 2 cap of them (gravity.hip WalkRec)
    double *g_seg = nullptr;                         // segment tree of leaf moments: 4 doubles x (cap + 64)
    int32_t *g_walkB = nullptr, *g_leafB = nullptr;  // int4 per node, int2 per leaf: packed walk pointers
    double *grav_tab = nullptr;                      // softening table, [F]:81-101
    uint64_t *g_keys = nullptr, *g_keys_alt = nullptr; uint32_t *g_vals = nullptr, *g_vals_alt = nullptr;
    void *g_sort_tmp = nullptr; size_t g_sort_tmp_bytes = 0;
    bool acc_marked = false, acc_any_mass = false;  // multi-GPU accretion: between sph_accrete_mark_dev and _apply_dev
    int64_t g_cap = 0;                               // leaves the tree arrays hold
    // external gravity sources (multi-GPU: the particles of every GPU), caller-owned {x,y,z,m} records + their bounding box
    const double *gx_src = nullptr; int64_t gx_n = 0; double gx_box[6] = {0, 0, 0, 0, 0, 0};
    bool gx_keys_valid = false;                      // g_keys_alt / g_vals_alt hold the sorted path keys of gx_src
    int32_t *number = nullptr; bool numbers_set = false;   // variable h: the reference's particle numbers by original id

    // grid
    sph::GridDesc grid{};
    double bbox[6] = {0, 0, 0, 0, 0, 0};   // min xyz, max xyz at the last grid build
    uint32_t *keys = nullptr, *keys_alt = nullptr, *vals = nullptr, *vals_alt = nullptr;
    void *sort_tmp = nullptr; size_t sort_tmp_bytes = 0;
    int32_t *cell_start = nullptr; int64_t cell_cap = 0;
    int32_t *cell_fill = nullptr;    // counting sort of the grid build: per-cell cursor (inside the cell_start allocation, behind the table)
    double *bbox_part = nullptr;     // per-block partial min/max
    double *h_pinned = nullptr;      // pinned host scratch (bbox[6], flags, dt, ...)
    int32_t *d_flags = nullptr;      // [0] nonfinite, [1] nlist max count
    // Read-backs one build stale (no host wait on the step that is being enqueued): the fixed-h path sizes its grid from
    // the bounding box of the PREVIOUS build plus one cell (a particle outside the box is clamped into a boundary cell and
    // still finds every neighbour: the box is a matter of speed, not of correctness) and checks the list overflow of the
    // previous build.  Two pinned slots and two events each, used alternately.
    hipEvent_t ev_bbox[2] = {nullptr, nullptr}, ev_nl[2] = {nullptr, nullptr};
    int ring_bbox = 0, ring_nl = 0;
    bool ring_bbox_valid = false, ring_nl_valid = false;
    bool bbox_exact = true;          // c->bbox is the exact box of the last build (not the previous one widened)
    bool no_stale = false;           // SPH_SYNC_EVERY_BUILD: wait for every read-back (A/B switch)
    int64_t host_syncs = 0;          // stream synchronisations inside the build path (statistics)

    // neighbour list: slot k of particle i (wave w = i/64, lane = i%64) lives at
    // nlist[(w*nl_cap + k)*64 + lane]  -> a wave reads 64 consecutive ints per k
    int32_t *nlist = nullptr; int32_t nl_cap = 0; int64_t nl_waves_cap = 0;
    int32_t *ncount = nullptr; int32_t *wave_max = nullptr;
    int32_t *plan_h = nullptr;                      // ... and of every half group (128 targets) of forces_q
    int32_t *plan_d = nullptr, *plan_f = nullptr;   // whole-tile kernels: the tile intervals of every workgroup (density / forces geometry), per list build
    int32_t *deal = nullptr;                        // forces_q: order of the targets within their workgroup (by list length)
    int32_t *ntail = nullptr;        // variable h: entries of the margin shell, stored from the end of the lane's column
    // variable h, re-flag pass (varh.hip): the growth of h the list in place was built to survive, the growth measured
    double vl_grow = 1.0, h_growth = 0.0, h_shrink = 0.0;    // h_shrink: largest h_build / h_now
    bool list_has_margin = false;    // the list in place carries the margin shell calc_smoothing reads
    int64_t nlist_reflags = 0;
    int32_t nl_max = 0; double nl_mean = 0.0;

    // kernel tables on the device
    double *w_tab = nullptr, *dw_tab = nullptr;
    double *w_pair = nullptr, *dw_pair = nullptr;    // pair-packed copies: entry k = {t[k], t[k+1]} (16-B aligned)

    // sinks on the device: 10 arrays of MAX_SINKS doubles: x y z vx vy vz m ax ay az
    int32_t ns = 0;
    double *sink = nullptr;
    double *sink_radius = nullptr;   // MAX_SINKS accretion radii
    double *sink_part = nullptr;     // per-block partial sums for the sink accelerations
    int32_t sink_blocks = 0;

    // time step on the device: [0] dt, [1] t, [2] dt candidate
    double *d_dt = nullptr;
    double *dt_part = nullptr; int32_t dt_blocks = 0;

    bool order_valid = false;    // sorted order, drec, bbox, orig/inv match the current positions
    bool grid_valid = false;     // sorted order + cell table + neighbour list match positions
    bool rho_valid = false;      // rho matches positions and masses
    bool eos_valid = false;      // P, c, frec match rho, u, alpha, v
    bool rates_valid = false;
    bool derived_kept = false;   // after an accretion / cull: rho .. dalpha of the survivors, compacted with the state (downloads only)

    // statistics and timing
    int64_t grid_builds = 0, nlist_builds = 0, density_passes = 0, force_passes = 0;
    int64_t device_bytes = 0;
    std::unordered_map<void *, size_t> allocs;   // every device allocation of this context
    int32_t rank = 0, nranks = 1;   // multi-GPU: only rank 0 adds the sink-sink pair terms before the all-reduce
    unsigned timing = 0;             // bit k: kernel group k is bracketed by HIP events
    int timing_stride = 1;           // ... every timing_stride-th launch of it (sph_timing_stride)
    int64_t timing_seen[32] = {};    // launches of group k since timing was switched on
    sph::TimingSlot tslot[SPH_K_COUNT];
};

namespace sph {

// device memory owned by the context (tracked so that sph_stats.device_bytes is exact)
int ctx_alloc_bytes(sph_ctx *c, void **p, size_t bytes, const char *what);
void ctx_free_ptr(sph_ctx *c, void *p);
template <class T>
inline int ctx_alloc(sph_ctx *c, T **p, size_t count, const char *what) {
    return ctx_alloc_bytes(c, reinterpret_cast<void **>(p), (count ? count : 1) * sizeof(T), what);
}
template <class T>
inline void ctx_free(sph_ctx *c, T *&p) {
    ctx_free_ptr(c, p);
    p = nullptr;
}

// all launchers enqueue on ctx->stream and return a hipError_t (no synchronisation unless noted)
hipError_t grid_sort_tmp_bytes(int64_t n, size_t *bytes);
// builds sorted order + cell table from the current positions.  Synchronises once (bbox read-back).
int grid_rebuild(sph_ctx *c);
// builds the neighbour list; synchronises once (overflow check), grows the list if needed
int nlist_build(sph_ctx *c);
hipError_t launch_density(sph_ctx *c, const PairConst &pc);
hipError_t launch_eos_only(sph_ctx *c, const PairConst &pc, bool ghosts_only = false);
hipError_t launch_forces(sph_ctx *c, const PairConst &pc, int part = 0);
hipError_t launch_classify_waves(sph_ctx *c);
hipError_t launch_sink_accel(sph_ctx *c, const PairConst &pc);
hipError_t launch_kick(sph_ctx *c, double dt, bool dt_from_device);
hipError_t launch_drift(sph_ctx *c, double dt, bool dt_from_device);
hipError_t launch_next_dt(sph_ctx *c, bool advance_t);
hipError_t launch_unpermute(sph_ctx *c, const double *src_sorted, double *dst_original);
hipError_t launch_iota(sph_ctx *c, int32_t *p, int64_t n);
hipError_t launch_fill(sph_ctx *c, double *p, double v, int64_t n);
hipError_t launch_scatter_field(sph_ctx *c, double *field, int64_t first, int64_t count, const double *vals);
hipError_t launch_gather_fields(sph_ctx *c, int nf, const int *fields, const int64_t *ids, int64_t count, double *out);
hipError_t launch_gather_selected(sph_ctx *c, int nf, const int *fields, int box, int64_t capacity, double *out);
hipError_t launch_scatter_fields(sph_ctx *c, int nf, const int *fields, int64_t first, int64_t count, const double *vals);
hipError_t launch_dt_partial_only(sph_ctx *c);
hipError_t launch_kick_drift(sph_ctx *c);
hipError_t launch_kick_next_dt(sph_ctx *c, bool advance_t);   // closing kick + get_next_timestep in one pass      // kick + drift with the device dt in one pass (sph_step / sph_run)
hipError_t launch_kick_dt_candidate(sph_ctx *c, bool with_sinks = true);
hipError_t launch_kick_sinks(sph_ctx *c);
// multi-GPU building blocks (domain.hip, grid.hip)
int owned_bbox(sph_ctx *c, double *d_out6, double *h_out6);      // h_out6 != nullptr: synchronises
int domain_select_boxes(sph_ctx *c, int nbox, const double *boxes, int64_t *counts);
int domain_select_boxes_enqueue(sph_ctx *c, int nbox, const double *boxes);     // the same without waiting: counts to pinned memory
void domain_selected_counts(sph_ctx *c, int nbox, int64_t *counts);                // ... read after the caller's own synchronisation
int domain_replace_ghosts(sph_ctx *c, int64_t count, const double *d_vals);
hipError_t launch_pack_partials(sph_ctx *c, double *d_out, bool predict_box);
hipError_t launch_set_numbers(sph_ctx *c, int64_t first, int64_t count, const int64_t *d_numbers);
hipError_t launch_apply_partials(sph_ctx *c, const double *d_all, int nranks, int stride, bool apply_dt);
// LDS-tiled fixed-h kernels (tiled.hip; default)
int nlist_build_tiled(sph_ctx *c);
constexpr int WT_TILE_RECORDS = 3712;     // whole-tile kernels: {x,y,z,m} records per LDS tile (116 KB; + the kernel table <= 160 KB)
hipError_t launch_density_wt(sph_ctx *c, const PairConst &pc);
hipError_t launch_forces_wt(sph_ctx *c, const PairConst &pc, int part);
double forces_lane_efficiency(const std::vector<int32_t> &cnt);   // list entries / lane-trips of the forces kernel in use (host, statistics)
// self-gravity (gravity.hip)
hipError_t grav_sort_tmp_bytes(int64_t n, size_t *bytes);
int gravity_tree_build(sph_ctx *c);
int global_keys_sorted(sph_ctx *c);      // sorted path keys of the external source set -> c->g_keys_alt / g_vals_alt
void gravity_free(sph_ctx *c);
hipError_t launch_gravity(sph_ctx *c);
// accretion + boundary cull (accrete.hip)
int accrete_and_cull(sph_ctx *c, int64_t *removed, int32_t *d_keep_out = nullptr);
int sink_creation(sph_ctx *c, int32_t *created);
int sink_candidate(sph_ctx *c, double *d_cand);    // multi-GPU halves of sink_creation
int sink_add_checked(sph_ctx *c, const double *d_cand, int32_t *created);
int sinks_cull(sph_ctx *c);                         // [V] check_bounds for the sinks    // [V] check_sink_creation; may add one sink (c->ns grows)
int accrete_mark_ext(sph_ctx *c, int64_t src_off, double *d_partials);       // multi-GPU: marks + per-rank sink sums
int accrete_apply_ext(sph_ctx *c, const double *d_all, int nranks, int stride, int32_t *d_keep_out, int64_t *removed);
// variable-h path (varh.hip)
hipError_t varh_sort_tmp_bytes(int64_t n, size_t *bytes);
int varh_h_stats(sph_ctx *c, bool with_growth = false);   // h_max_glob, h_mean (+ h_growth against c->h_new): one read-back
bool varh_can_reflag(const sph_ctx *c);
int varh_nlist_reflag(sph_ctx *c);
int varh_reflag_density(sph_ctx *c, const PairConst &pc);   // the re-flag pass and the density pass in one walk over the list
int varh_leaf_build(sph_ctx *c);       // leaf boxes of all particles for the current positions + h
int varh_nlist_build(sph_ctx *c);
int varh_refresh_h(sph_ctx *c);        // only h changed: prec, reaches and per-cell max h from the new h
hipError_t launch_density_v(sph_ctx *c, const PairConst &pc);
hipError_t launch_eos_only_v(sph_ctx *c, const PairConst &pc);
hipError_t launch_forces_v(sph_ctx *c, const PairConst &pc);
hipError_t launch_update_h(sph_ctx *c, const PairConst &pc);   // leaves the local candidate (min * dt_scale) in d_dt[2]
PairConst make_pair_const(const sph_ctx *c);

}  // namespace sph
