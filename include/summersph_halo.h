/* summersph_halo.h -- the loop body of simulate() on several GPUs, natively (libsummersph_halo.so).
 *
 * One process (or thread) per GPU, each with its own sph_ctx (include/summersph.h).  A sph_halo object ties the
 * context to a communicator and runs the reference's step sequence (/root/reference/SUMMER_SPH.f90:889-916) on the
 * domain decomposition of summersph_amd/dist.py -- slabs along x, ghost copies of the neighbours' particles within 2h
 * -- without Python: the ghost exchange and the field refreshes are packed on the context's stream, travel as grouped
 * ncclSend / ncclRecv on a SECOND stream and are unpacked behind an event the boundary wavefronts wait for, while
 * the interior wavefronts (sph_forces_part(1)) run; the per-evaluation reduction (sink accelerations, dt candidate,
 * predicted boxes) is one ncclAllGather of SPH_PARTIALS doubles; the ghost payload needs no size exchange (the room of a message is
 * agreed from the last one, a header carries the count, overflows go a second round).  Fixed-h contexts without self-gravity (the headline path);
 * the octree paths stay with dist.py.
 *
 * The reference has no counterpart (single process); what this replaces on the reference side is the body of
 * `simulate` between the reader and the writer, exactly as sph_run does on one GPU.
 *
 * Transports:
 *   sph_halo_create   RCCL: ncclCommInitRank from a 128-byte id every rank got from rank 0 (file, MPI, a socket ...),
 *                     own communication stream.
 *   sph_halo_attach   RCCL on the caller's ncclComm_t and hipStream_t (e.g. a communicator an MPI host already has).
 *   sph_halo_create_inproc  the ranks are THREADS of one process sharing one device (device-to-device copies through a hub):
 *                     how the orchestration is tested on a one-GPU box.  Not for production.
 * Status: the RCCL transport has run with ONE rank only (collectives and a send/recv to itself, tests/test_halo_gpu.py); the
 * test pool has no multi-GPU node.  The orchestration above it is the code the in-process transport runs with 2-4 ranks.
 *
 * Every entry point returns SPH_OK or an SPH_ERR_* status; sph_halo_last_error gives text.  All ranks must make the same
 * calls in the same order (the rule of any collective library).
 */
#ifndef SUMMERSPH_HALO_H
#define SUMMERSPH_HALO_H

#include "summersph.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct sph_halo sph_halo;

#define SPH_HALO_ID_BYTES 128

/* rank 0: a fresh communicator id (ncclGetUniqueId) */
int sph_halo_unique_id(void *id128);
int sph_halo_create(sph_ctx *ctx, const void *id128, int32_t rank, int32_t nranks, sph_halo **out);
int sph_halo_attach(sph_ctx *ctx, void *nccl_comm, void *comm_stream, int32_t rank, int32_t nranks, sph_halo **out);
/* in-process hub for `nranks` threads (tests) */
void *sph_halo_hub_create(int32_t nranks);
void sph_halo_hub_destroy(void *hub);
int sph_halo_create_inproc(sph_ctx *ctx, void *hub, int32_t rank, int32_t nranks, sph_halo **out);
int sph_halo_destroy(sph_halo *h);
const char *sph_halo_last_error(const sph_halo *h);

/* ownership: rank r owns x in [edges[r-1], edges[r]) (nranks - 1 interior edges, ascending; the same on every rank).
 * Every `migrate_every` steps the particles that left their slab change owner (0: never).                       */
int sph_halo_set_slabs(sph_halo *h, const double *edges, int32_t migrate_every);
/* this rank's particles (host arrays, alpha may be NULL) and their global numbers (NULL: 0..n-1)               */
int sph_halo_upload(sph_halo *h, int64_t n, const double *x, const double *y, const double *z,
                    const double *vx, const double *vy, const double *vz,
                    const double *u, const double *m, const double *alpha, const int64_t *gid);
/* the same for a variable-h context (SPH_FLAG_VARIABLE_H): + the smoothing lengths, the 10th column of the reader
 * of "SUMMER_SPH - Variable.f90":782                                                                               */
int sph_halo_upload_v(sph_halo *h, int64_t n, const double *x, const double *y, const double *z,
                      const double *vx, const double *vy, const double *vz,
                      const double *u, const double *m, const double *alpha, const double *hsml, const int64_t *gid);
/* nsteps iterations of the loop body; dt, t in and out as sph_run                                                */
int sph_halo_run(sph_halo *h, int32_t nsteps, double *dt, double *t);
int64_t sph_halo_count(const sph_halo *h);         /* owned particles of this rank                                  */
/* the owned particles (any pointer may be NULL); capacity >= sph_halo_count                                       */
int sph_halo_download(sph_halo *h, int64_t capacity, double *x, double *y, double *z,
                      double *vx, double *vy, double *vz, double *u, double *m, double *alpha, int64_t *gid);
int sph_halo_download_v(sph_halo *h, int64_t capacity, double *x, double *y, double *z,
                        double *vx, double *vy, double *vz, double *u, double *m, double *alpha, double *hsml, int64_t *gid);
/* every rank's owned particles on `root`, ordered by global number (for a single save file as the reference
 * writes it): collective; on root the arrays hold n_total entries (capacity >= n_total), elsewhere they are unused */
int sph_halo_gather_root(sph_halo *h, int32_t root, int64_t capacity, int64_t *n_total, double *x, double *y, double *z,
                         double *vx, double *vy, double *vz, double *u, double *m, double *alpha, int64_t *gid);
int sph_halo_gather_root_v(sph_halo *h, int32_t root, int64_t capacity, int64_t *n_total, double *x, double *y, double *z,
                           double *vx, double *vy, double *vz, double *u, double *m, double *alpha, double *hsml, int64_t *gid);

typedef struct sph_halo_stats {
    int64_t ghosts;        /* ghost particles held after the last exchange                                       */
    int64_t migrated;      /* particles that changed owner (all ranks, since creation)                           */
    int64_t exchanges;     /* grouped point-to-point rounds                                                       */
    int64_t collectives;   /* all-gathers                                                                         */
    int64_t migrations;    /* migration rounds                                                                    */
    int64_t host_waits;    /* times the host waited for the device inside sph_halo_run                            */
    int64_t removed;       /* particles of this rank accreted or culled (since creation)                          */
    int64_t sinks_created; /* sinks created by check_sink_creation (the same on every rank)                       */
    int64_t let_sent;      /* self-gravity, locally essential tree: particles + pseudo-particles this rank shipped  */
    int64_t let_received;  /* ... and received (32 bytes each), summed over the source updates                      */
    int64_t let_updates;   /* source updates (one per drift)                                                      */
} sph_halo_stats;
int sph_halo_get_stats(const sph_halo *h, sph_halo_stats *out);
/* transport check: every rank sends `count` doubles to every rank (itself included) and verifies what arrives;
 * then an all-gather of the same pattern.  SPH_OK iff all of it matched.                                          */
int sph_halo_selftest(sph_halo *h, int64_t count);

#ifdef __cplusplus
}
#endif
#endif /* SUMMERSPH_HALO_H */
