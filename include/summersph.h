/* summersph.h -- C ABI of the MI355X-native SPH core (libsummersph_hip.so).
 *
 * Drop-in boundary for the hot path of graves-andrew-02/SUMMERSPH: the calls `simulate`
 * makes between tree allocation and the second `kick` (reference file
 * SUMMER_SPH.f90, "[F]", lines 886-916).  The reference has no FFI; its de-facto interface
 * is the set of module procedures listed beside each entry point below.  A Fortran
 * maintainer binds these with `bind(C)` interfaces (summersph_amd/host/sph_hip_binding.f90,
 * INTEGRATION.md).
 *
 * Conventions
 *   - every entry point returns an int status (SPH_OK == 0); nothing aborts or prints.
 *     sph_strerror() / sph_last_error() give text.
 *   - caller-owned HOST arrays of double, contiguous, length n (struct-of-arrays); the
 *     opaque context owns all device memory.  *_dev variants take DEVICE pointers instead
 *     (inputs already resident in HBM).
 *   - one context per GPU, one host thread per context (not re-entrant per context).
 *   - particle order: whatever order the caller uploaded ("original order").  Internally
 *     particles live cell-sorted; every download un-permutes.
 *   - there is NO CPU fallback: without a HIP device sph_ctx_create fails with
 *     SPH_ERR_NO_DEVICE.
 */
#ifndef SUMMERSPH_H
#define SUMMERSPH_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SPH_ABI_VERSION 1

typedef struct sph_ctx sph_ctx;

enum sph_status {
    SPH_OK = 0,
    SPH_ERR_ARG = 1,          /* null pointer, negative size, bad field id ...            */
    SPH_ERR_NO_DEVICE = 2,    /* no HIP device / device index out of range                */
    SPH_ERR_HIP = 3,          /* a HIP runtime call failed (text in sph_last_error)       */
    SPH_ERR_NOMEM = 4,        /* device or host allocation failed                         */
    SPH_ERR_STATE = 5,        /* call order violated (e.g. forces before density)         */
    SPH_ERR_GRID = 6,         /* cell grid would exceed 2^31 cells (non-finite positions?) */
    SPH_ERR_NONFINITE = 7     /* NaN/Inf in particle positions at grid build               */
};

/* Runtime parameters.  Defaults (sph_params_default) are the reference's compile-time
 * constants of SUMMER_SPH.f90, with REAL(4)-rounded literals reproduced bit for bit. */
typedef struct sph_params {
    double h;            /* smoothing length; [F]:11 smoothing = 2.5                      */
    double gamma;        /* 1.4    ([F]:466)                                              */
    double gamma_m1;     /* 0.4    ([F]:465 writes the literal 0.4_dp, not gamma-1)       */
    int32_t nq;          /* 5000   ([F]:8) kernel-table intervals on q in [0,2]           */
    int32_t flags;       /* SPH_FLAG_*                                                    */
    double kernel_pi;    /* 3.14159265359 ([F]:125-126)                                   */
    double visc_eps;     /* (double)0.01f  ([F]:373)                                      */
    double alpha_floor;  /* 0.1    ([F]:317)                                              */
    double alpha_decay;  /* (double)0.15f ([F]:317)                                       */
    double G;            /* (double)39.47841760435743f ([F]:7)                            */
    double dt_scale;     /* 0.25   ([F]:851)                                              */
    double dt_max;       /* (double)0.1f    ([F]:855)                                     */
    double dt_min;       /* (double)0.0001f ([F]:857)                                     */
    double bounding_size;/* 1500   ([F]:11), used by sph_cull_bounds                      */
    /* variable-h path only ("SUMMER_SPH - Variable.f90", "[V]"; flags & SPH_FLAG_VARIABLE_H)  */
    double eta;          /* h = eta (m/rho)^(1/3) target of calc_smoothing ([V]:527)          */
    double h_tol;        /* convergence_criteria of calc_smoothing ([V]:529)                  */
    double h_max_length; /* max_length ([V]:528)                                              */
    double h_min_length; /* (double)0.01f ([V]:528)                                           */
    double h_iter_cap;   /* 10.0 ([V]:529)                                                    */
    /* Barnes-Hut gas self-gravity (flags & SPH_FLAG_SELF_GRAVITY)                              */
    double theta;        /* 0.5: opening angle, hard coded at the call site [F]:825 / [V]:1029 */
} sph_params;

/* flags */
#define SPH_FLAG_VARIABLE_H 2   /* per-particle smoothing length, grad-h terms and the leaf-box
                                   neighbour rule of the reference's variable-h variant [V]:
                                   kernel normalised with h_i and REAL(4) pi, nq = 2500, Omega,
                                   h update after every step.  Upload h with sph_upload_field. */
#define SPH_FLAG_SELF_GRAVITY 16 /* find_forces WITH the Barnes-Hut gas self-gravity term (particle_gravforces,
                                   [F]:249-290, 825): sph_forces then equals find_forces as it is (several GPUs:
                                   sph_set_gravity_sources_dev) */
#define SPH_FLAG_ACCRETE_CULL 32 /* sph_step / sph_run also do the end-of-step sink accretion and boundary cull
                                   of simulate() ([F]:919-920): the particle count may shrink (sph_count) */
#define SPH_FLAG_SINK_CREATION 64 /* variable h: sph_step / sph_run also run check_sink_creation ([V]:549-597, 1155): the
                                    number of sinks may grow by one per step (sph_sink_count)                  */
#define SPH_FLAG_NO_LDS_TILES 4  /* fixed-h path: build the neighbour list with per-lane gathers (pairs.hip)
                                   instead of LDS-staged tiles (tiled.hip); A/B measurements        */
/* flag bit 8 (round 1: SPH_FLAG_LDS_TILE_EVAL, chunk-staged density/forces kernels) was measured slower in every
   configuration and is gone; the bit is ignored */
#define SPH_FLAG_NO_WHOLE_TILE 128 /* fixed-h path: density/forces with the memory gathers of pairs.hip only, never the
                                   whole-tile kernels of tiled.hip (bitwise the same results); A/B measurements */
#define SPH_FLAG_REUSE_GRAVITY 256 /* self-gravity: keep the Barnes-Hut term of the last walk and copy it instead of
                                    walking again while positions, masses, h and the tree are unchanged -- the case of
                                    the start-of-step evaluation, which sees the positions of the previous step's
                                    last evaluation (bitwise the same accelerations); OFF by default: the reference
                                    walks its tree in both evaluations ([F]:898,910) and the headline numbers do too */
#define SPH_FLAG_NO_REFLAG 512   /* variable h: build the neighbour list anew also when only h changed since the last build
                                    (start of a step, after calc_smoothing); by default that list is derived in place from the
                                    list of the old lengths (varh.hip nlist_v_reflag: the same neighbour sets, entries in a
                                    different order); A/B measurements                                */
#define SPH_FLAG_REUSE_DENSITY 1 /* skip the density pass when positions and masses did not
                                    change since the last one (bitwise the same rho); OFF by
                                    default: the reference recomputes it, [F]:896,908 */

/* field ids for sph_download_field / sph_field_dev */
enum sph_field {
    SPH_F_X = 0, SPH_F_Y, SPH_F_Z, SPH_F_VX, SPH_F_VY, SPH_F_VZ, SPH_F_U, SPH_F_M, SPH_F_ALPHA,
    SPH_F_RHO, SPH_F_P, SPH_F_C, SPH_F_AX, SPH_F_AY, SPH_F_AZ, SPH_F_DU, SPH_F_DALPHA,
    SPH_F_H,        /* smoothing length (state, variable-h path; [V]:24 s_length)             */
    SPH_F_OMEGA,    /* grad-h factor (derived, variable-h path; [V]:25 omega)                  */
    SPH_F_COUNT
};

/* kernels whose device time is recorded when timing is on */
enum sph_kernel_id {
    SPH_K_GRID = 0,    /* bbox + keys + sort + cell table + reorder                        */
    SPH_K_NLIST,       /* neighbour-list build                                            */
    SPH_K_DENSITY,     /* density + EOS                                                   */
    SPH_K_FORCES,      /* sink gravity + SPH pair forces + alpha rate                     */
    SPH_K_SINKACC,     /* acceleration of the sinks                                       */
    SPH_K_KICK, SPH_K_DRIFT, SPH_K_DT,
    SPH_K_LEAF,        /* variable-h: octree leaf boxes (Morton keys, sort, depth)             */
    SPH_K_UPDATE_H,    /* variable-h: calc_smoothing                                          */
    SPH_K_GRAVITY,     /* self-gravity: octree keys, radix tree, node sums, tree walk          */
    SPH_K_GRAV_WALK,   /* self-gravity: the tree walk alone (inside SPH_K_GRAVITY)             */
    SPH_K_REFLAG,      /* variable-h: the list of the new h derived from the list in place (nlist_v_reflag);
                          SPH_K_NLIST then counts list BUILDS only                              */
    SPH_K_COUNT
};

typedef struct sph_stats {
    int64_t n;              /* gas particles                                              */
    int64_t n_cells;        /* cells of the current grid                                  */
    int32_t grid_dim[3];    /* cells along x, y, z                                        */
    int32_t nlist_capacity; /* neighbour slots per particle currently allocated           */
    int32_t nlist_max;      /* largest neighbour count found at the last build            */
    int32_t tile_fit_pct;   /* fixed h: workgroups (%) whose neighbour intervals fit the LDS tile of the
                               whole-tile pair kernels at the last build; those kernels run when >= 90;
                               -1: kernels off (flags, variable h)                        */
    double  nlist_mean;     /* mean neighbour count (pairs inside 2h, self excluded)      */
    int64_t grid_builds, nlist_builds, density_passes, force_passes;
    int64_t device_bytes;   /* HBM held by the context                                    */
    double  nlist_wave_mean;/* mean over wavefronts of the longest list in the wave = trips the pair kernels run */
    int32_t tile_fit_pct_forces; /* as tile_fit_pct, for the forces kernel's workgroup size and tile record            */
    int32_t host_syncs;     /* stream synchronisations inside grid / list builds since the context was created: a
                               steady-state fixed-h step adds none (read-backs are taken one build late)       */
    double  lane_efficiency_forces; /* fixed h: list entries / lane-trips of the forces kernel in use (1 = no idle lanes):
                                       forces_q deals targets by list length, so this is not nlist_mean / nlist_wave_mean */
    int64_t nlist_reflags;  /* variable h: list builds replaced by the re-flag pass (same positions, new h: the list of the
                               new lengths derived from the list in place, entry for entry what a build would write) */
} sph_stats;

/* ---- life cycle: replaces init_kernel_table ([F]:55-79) and the tree (de)allocation
 *      in simulate ([F]:894,901,905,928) ------------------------------------------------ */
int sph_params_default(sph_params *p);
/* defaults of the variable-h variant: SPH_FLAG_VARIABLE_H, nq 2500, REAL(4) pi, gamma_m1 =
 * gamma - 1.0, eta 1.2, h_tol 1e-3, h_max_length 10 (the reference ships no parameters.txt) */
int sph_params_default_variable(sph_params *p);
int sph_ctx_create(const sph_params *p, int device, sph_ctx **out);
int sph_ctx_destroy(sph_ctx *ctx);
const char *sph_strerror(int status);
const char *sph_last_error(const sph_ctx *ctx);
int sph_abi_version(void);
/* the parameters the context was created with */
int sph_get_params(const sph_ctx *ctx, sph_params *out);

/* ---- state hand-over: replaces packing from `type(particle)` / `type(sink)` ([F]:14-37).
 *      alpha may be NULL (-> 0, as the reader initialises it, [F]:681). ------------------ */
int sph_upload(sph_ctx *ctx, int64_t n, const double *x, const double *y, const double *z,
               const double *vx, const double *vy, const double *vz,
               const double *u, const double *m, const double *alpha);
int sph_upload_dev(sph_ctx *ctx, int64_t n, const double *d_x, const double *d_y, const double *d_z,
                   const double *d_vx, const double *d_vy, const double *d_vz,
                   const double *d_u, const double *d_m, const double *d_alpha);
int sph_set_sinks(sph_ctx *ctx, int32_t ns, const double *sx, const double *sy, const double *sz,
                  const double *svx, const double *svy, const double *svz, const double *sm);
/* any output pointer may be NULL */
int sph_get_sinks(sph_ctx *ctx, int32_t ns, double *sx, double *sy, double *sz,
                  double *svx, double *svy, double *svz, double *sm,
                  double *sax, double *say, double *saz);
/* accretion radii of the sinks ([F]:694: 3.5, [V]:830: 5.0 for file sinks, 0 for the dummy sink); the
 * defaults set by sph_set_sinks are 3.5 (fixed h) / 5.0 (variable h) */
int sph_set_sink_radii(sph_ctx *ctx, int32_t ns, const double *radius);
int sph_get_sink_radii(sph_ctx *ctx, int32_t ns, double *radius);
int64_t sph_count(const sph_ctx *ctx);
int32_t sph_sink_count(const sph_ctx *ctx);
/* check_sink_creation of the variable-h reference ([V]:549-597) on the sorted order of the last sph_density: the first
 * particle (caller's order) with m (eta/h)^3 > 0.5 either lies within radius + 2h of a sink (nothing happens) or a
 * sink of mass 1e-11 and radius 2h is created at its position; *created = 0/1 */
int sph_check_sink_creation(sph_ctx *ctx, int32_t *created);
/* the two halves of it for several GPUs: the first candidate among the owned particles as a record of SPH_SINK_CAND
 * doubles {particle number or +inf, x, y, z, vx, vy, vz, h, 0} in device memory; the caller all-gathers the records,
 * picks the lowest number and hands that record to every context, which applies the distance test and the creation */
#define SPH_SINK_CAND 9
int sph_sink_candidate_dev(sph_ctx *ctx, double *d_cand);
int sph_add_sink_checked_dev(sph_ctx *ctx, const double *d_cand, int32_t *created);
/* one field in the caller's particle order, host or device source (e.g. SPH_F_H after sph_upload) */
int sph_upload_field(sph_ctx *ctx, int field, const double *host, int64_t n);
int sph_upload_field_dev(sph_ctx *ctx, int field, const double *d_vals, int64_t n);

/* ---- the hot path ------------------------------------------------------------------- */
/* create_tree + get_density + get_pressure_and_sound_speed   ([F]:894-897, 398-468)      */
int sph_density(sph_ctx *ctx);
/* find_forces ([F]:818-829): zero_rates, [particle_gravforces when SPH_FLAG_SELF_GRAVITY],
 * sink_gravforces, get_SPH   ([F]:249-290, 559-591, 295-395)                             */
int sph_forces(sph_ctx *ctx);
/* kick ([F]:742-759) and drift ([F]:762-776), gas and sinks                              */
int sph_kick(sph_ctx *ctx, double dt);
int sph_drift(sph_ctx *ctx, double dt);
/* get_next_timestep ([F]:831-860): in/out dt                                             */
int sph_next_dt(sph_ctx *ctx, double *dt);
/* initiate_sink_accretion (only if a sink has mass, [F]:919) + check_bounds ([F]:920) on the current
 * positions: removes accreted / escaped particles on the device, merges accreted mass and momentum
 * into the sinks.  Survivors keep their relative order: the caller's numbering becomes the rank among
 * survivors, as after the reference's pack().  Needs the grid of the current positions (call it after
 * sph_density / sph_forces / sph_kick, before the next sph_drift).                               */
int sph_accrete_and_cull(sph_ctx *ctx, int64_t *n_removed);
/* the same, and which particles stayed: d_keep (device, one int32 per particle held BEFORE the call, the caller's order;
 * 1 = survivor) -- for a caller that keeps per-particle data of its own beside the context (the global numbers of
 * libsummersph_halo.so) and has to pack() it the same way ([F]:481,554)                                          */
int sph_accrete_and_cull_keep(sph_ctx *ctx, int32_t *d_keep, int64_t *n_removed);
/* variable-h only: calc_smoothing ([V]:515-546) on the neighbour structure of the last evaluation */
int sph_update_h(sph_ctx *ctx);
/* one iteration of simulate's loop body, [F]:889-916 (variable-h: [V]:1120-1152 incl. the h update): density, forces, kick, drift,
 * density, forces, kick, t += dt, next dt.  Identical to the unfused call sequence.      */
int sph_step(sph_ctx *ctx, double *dt, double *t);
/* nsteps iterations without returning to the host in between (dt stays on the device)    */
int sph_run(sph_ctx *ctx, int32_t nsteps, double *dt, double *t);

/* ---- read-back, original particle order ---------------------------------------------- */
int sph_download_field(sph_ctx *ctx, int field, double *host, int64_t n);
int sph_download_field_dev(sph_ctx *ctx, int field, double *d_out, int64_t n);
int sph_download_state(sph_ctx *ctx, int64_t n, double *x, double *y, double *z,
                       double *vx, double *vy, double *vz, double *u, double *m, double *alpha);

/* ---- multi-GPU building blocks (one context per GPU; orchestration: summersph_amd/dist.py) --
 * A context may hold GHOST particles: copies of other GPUs' particles that lie within 2h of
 * this GPU's domain.  Upload owned particles first and ghosts after them, then declare how
 * many are owned: original ids [n_owned, n) are ghosts.  Ghosts act as neighbours in the
 * density and force sums but are never targets (no rho, rates, dt candidate or sink pull is
 * computed for them); their rho comes from their owner through sph_scatter_field_dev.      */
int sph_set_owned(sph_ctx *ctx, int64_t n_owned);
int sph_set_rank(sph_ctx *ctx, int32_t rank, int32_t nranks);
/* field[slot of original id first+k] = d_vals[k], k in [0,count): refreshes ghost rho after the
 * owners' density pass, or ghost v, u, alpha after a kick.                                 */
int sph_scatter_field_dev(sph_ctx *ctx, int field, int64_t first, int64_t count, const double *d_vals);
/* several fields at once (one kernel, one synchronisation):
 * gather : d_out[f*count + k] = field fields[f] of the particle with original id d_ids[k]
 *          (d_ids == NULL: ids 0..count-1, e.g. all owned particles)
 * scatter: field fields[f] of original id first+k = d_vals[f*count + k]                      */
int sph_gather_fields_dev(sph_ctx *ctx, int32_t nf, const int32_t *fields, int64_t count,
                          const int64_t *d_ids, double *d_out);
int sph_scatter_fields_dev(sph_ctx *ctx, int32_t nf, const int32_t *fields, int64_t first, int64_t count,
                           const double *d_vals);
/* ---- the same exchange kept on the device (no re-upload, no host reductions) ----------------
 * sph_set_stream      run the context on the caller's HIP stream (e.g. torch's current stream; NULL =
 *                     the default stream): calls that take device pointers then stop synchronising
 *                     and are ordered with the caller's own work on that stream.
 * sph_reserve         minimum slot capacity of the next sph_upload (room for ghosts).
 * sph_owned_bbox      min xyz, max xyz of the owned particles at their current positions, to the host
 *                     (lo_hi, synchronises) and/or to device memory (d_lo_hi).
 * sph_select_boxes    for each of nbox boxes {lo xyz, hi xyz}: the owned particles inside, ascending
 *                     original id; counts to the host; ids fetched with sph_selected_ids_dev.
 * sph_select_boxes_async  the same selection without the wait: the counts stay on the device (and travel to pinned memory
 *                     behind the selection); sph_gather_selected_dev packs a selection whose size only the device knows:
 *                     d_out[0] = count, d_out[1] = 0, then d_out[2 + f*count + k] -- if count <= capacity, else the header
 *                     alone; sph_selected_counts hands the counts over once the caller has synchronised with whatever
 *                     followed the selection on the context's stream (it does not wait itself).
 * sph_replace_ghosts_dev  drop the current ghosts and append `count` new ones (d_state[f*count + k],
 *                     f = x y z vx vy vz u m alpha); owned particles stay where they are, the next
 *                     sph_density re-sorts everything.  SPH_ERR_NOMEM if the slots do not suffice.
 * sph_set_dt / sph_get_dt, sph_kick_devdt / sph_drift_devdt: dt and t held on the device
 *                     (sph_run's mechanism), so that a step needs no host round trip for them.
 * sph_kick_drift_devdt / sph_kick_dt_candidate_dev: the same pairs of calls fused into one pass over the state each
 *                     (kick + drift; closing kick + local dt candidate), bitwise the separate calls.
 *                     sph_kick_dt_candidate_gas_dev + sph_kick_sinks_devdt split the latter into its gas and its sink part, so
 *                     that the gas is kicked while the sinks' accelerations still travel through the rank reduction.
 * sph_dt_candidate_dev    local dt candidate ([F]:845-851 over owned particles) kept on the device.
 * sph_pack_partials_dev   d_out[0..3*64) = this GPU's partial sink accelerations (ax[64] ay[64]
 *                     az[64]), d_out[192] = its dt candidate, d_out[193..199) = the bounding box its owned particles
 *                     will have after the coming kick + drift (union over the three dt values the dt rule can
 *                     produce; NaN when the rates are stale): SPH_PARTIALS doubles, to be
 *                     all-gathered by the caller.
 *                     sph_pack_partials_ex_dev(.., predict_box = 0) leaves the box out (NaN): it costs a pass over the
 *                     particles and only a reduction that is followed by a drift has a reader for it.
 * sph_apply_partials_dev  sink accelerations = sum over the nranks gathered blocks (rank order);
 *                     apply_dt != 0: t += dt, then [F]:855-858 with the minimum candidate.
 * sph_set_boundary_boxes / sph_forces_part: forces in two launches so that the exchange of ghost fields overlaps
 *                     the bulk of the work.  boxes = the other GPUs' bounding boxes {lo xyz, hi xyz} (every ghost lies
 *                     inside them).  part 1: sink gravity + every wavefront whose 64 particles are all farther than
 *                     2h from all boxes (they cannot have a ghost neighbour); part 2 (after the ghost fields arrived
 *                     and sph_refresh_eos ran): the remaining wavefronts.  Together identical to sph_forces.
 *                     Fixed-h contexts without self-gravity.
 * sph_set_gravity_sources_dev  self-gravity from a particle set other than the context's own: n_src records
 *                     {x, y, z, m} in device memory (caller-owned, must stay valid until replaced) and their bounding
 *                     box {min xyz, max xyz} (host).  The Barnes-Hut tree is then built over these sources -- with
 *                     every GPU's particles all-gathered into them it is the SAME tree on every GPU as the single
 *                     tree of the undecomposed run -- and walked for the context's owned particles.  n_src = 0: back
 *                     to the context's own particles.
 * sph_accrete_mark_dev / sph_accrete_apply_dev  sink accretion + boundary cull ([F]:471-556) when the octree is that of
 *                     all GPUs' particles (after sph_forces with sph_set_gravity_sources_dev; src_offset = position
 *                     of this GPU's owned particles, in the caller's order, inside the source set).  mark: decides for
 *                     the owned particles and writes this GPU's sums per sink (m, m x, m y, m z, m vx, m vy, m vz:
 *                     SPH_ACC_PARTIALS doubles) to d_partials; the caller all-gathers them.  apply: sink update from the
 *                     sums of all ranks (rank order), then the context keeps only the surviving OWNED particles, in
 *                     the caller's order (ghosts are dropped); d_keep (optional) receives keep[0..n_owned_before).
 * sph_set_numbers_dev  variable h: the reference's particle numbers (the rank in the input file) of original ids
 *                     [first, first + count).  The force pair {a, b} is evaluated iff the walk of the HIGHER-numbered
 *                     partner reaches the other's leaf ([V]:383), so on several GPUs the numbers must be the global
 *                     ones, for owned particles and ghosts alike.  Without this call: the context's own numbering.
 *                     With SPH_FLAG_VARIABLE_H sph_replace_ghosts_dev takes 10 rows (.. alpha, h) and
 *                     sph_set_gravity_sources_dev also supplies the octree the leaf boxes are taken from.            */
#define SPH_PARTIALS 199
int sph_set_numbers_dev(sph_ctx *ctx, int64_t first, int64_t count, const int64_t *d_numbers);
#define SPH_ACC_PARTIALS 448
int sph_accrete_mark_dev(sph_ctx *ctx, int64_t src_offset, double *d_partials);
int sph_accrete_apply_dev(sph_ctx *ctx, const double *d_all, int32_t nranks, int32_t stride, int32_t *d_keep, int64_t *n_removed);
int sph_set_gravity_sources_dev(sph_ctx *ctx, int64_t n_src, const double *d_xyzm, const double *lo_hi);
int sph_set_boundary_boxes(sph_ctx *ctx, int32_t nbox, const double *boxes);
int sph_forces_part(sph_ctx *ctx, int32_t part);
int sph_set_stream(sph_ctx *ctx, void *hip_stream);
int sph_reserve(sph_ctx *ctx, int64_t n_slots);
int sph_owned_bbox(sph_ctx *ctx, double *lo_hi, double *d_lo_hi);
int sph_select_boxes(sph_ctx *ctx, int32_t nbox, const double *boxes, int64_t *counts);
int sph_selected_ids_dev(sph_ctx *ctx, int32_t box, int64_t count, int64_t *d_ids);
int sph_select_boxes_async(sph_ctx *ctx, int32_t nbox, const double *boxes);
int sph_selected_counts(sph_ctx *ctx, int32_t nbox, int64_t *counts);
int sph_gather_selected_dev(sph_ctx *ctx, int32_t box, int32_t nf, const int32_t *fields, int64_t capacity, double *d_out);
int sph_replace_ghosts_dev(sph_ctx *ctx, int64_t count, const double *d_state);
int sph_set_dt(sph_ctx *ctx, double dt, double t);
int sph_get_dt(sph_ctx *ctx, double *dt, double *t);
int sph_kick_devdt(sph_ctx *ctx);
int sph_drift_devdt(sph_ctx *ctx);
int sph_dt_candidate_dev(sph_ctx *ctx);
int sph_kick_drift_devdt(sph_ctx *ctx);
int sph_kick_dt_candidate_dev(sph_ctx *ctx);
int sph_kick_dt_candidate_gas_dev(sph_ctx *ctx);
int sph_kick_sinks_devdt(sph_ctx *ctx);
int sph_pack_partials_dev(sph_ctx *ctx, double *d_out);
int sph_pack_partials_ex_dev(sph_ctx *ctx, double *d_out, int32_t predict_box);
int sph_apply_partials_dev(sph_ctx *ctx, const double *d_all, int32_t nranks, int32_t stride, int32_t apply_dt);
/* P, c and the force gather records of ALL slots from the current rho, u, alpha, v          */
int sph_refresh_eos(sph_ctx *ctx);
/* the same for the ghost slots only (the owned particles' records are written by sph_density itself) */
int sph_refresh_eos_ghosts(sph_ctx *ctx);
/* the local part of get_next_timestep ([F]:845-851): min over OWNED particles * dt_scale;
 * the caller min-reduces over ranks and applies [F]:855-858                                */
int sph_dt_candidate(sph_ctx *ctx, double *candidate);
/* overwrite the sink accelerations (after summing the per-GPU partial sums over ranks)     */
int sph_set_sink_accel(sph_ctx *ctx, int32_t ns, const double *sax, const double *say, const double *saz);

/* ---- diagnostics / measurement -------------------------------------------------------- */
int sph_get_stats(sph_ctx *ctx, sph_stats *out);
/* bounding box of the particle positions at the last grid build (= the current positions
 * after sph_density / sph_step): lo[3], hi[3].  Serves check_bounds ([F]:471-482) without a
 * download.                                                                              */
int sph_get_bbox(sph_ctx *ctx, double *lo, double *hi);
/* HIP events around kernel groups: on = 0 none, 1 every group, else a mask with bit (k + 1) set for sph_kernel_id k (timing
 * one group costs two event records per launch of that group; timing all of them ~4 % of a fixed-h step)                */
int sph_timing_enable(sph_ctx *ctx, int on);
/* bracket only every stride-th launch of a timed group (default 1): an event pair costs the stream ~14 us, so a run that is
 * itself being timed samples its dominant kernel instead of bracketing every launch; sph_timing_get then returns the
 * bracketed launches and their total                                                                                      */
int sph_timing_stride(sph_ctx *ctx, int stride);
int sph_timing_reset(sph_ctx *ctx);
int sph_timing_get(sph_ctx *ctx, int kernel_id, double *total_ms, int64_t *launches);
int sph_synchronize(sph_ctx *ctx);
/* the HIP stream all kernels of this context are launched on (hipStream_t as void*)      */
void *sph_stream(sph_ctx *ctx);

#ifdef __cplusplus
}
#endif
#endif /* SUMMERSPH_H */
