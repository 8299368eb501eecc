// halo.hip -- libsummersph_halo.so: the loop body of simulate() on several GPUs without Python (include/summersph_halo.h).
//
// A client of the public C ABI (include/summersph.h) only: every physics kernel is the context's; this file owns the
// orchestration of summersph_amd/dist.py (DistSim._step / evaluate / _exchange_ghosts / _migrate / _reduce /
// _gravity_sources / _accrete_and_cull: fixed h and variable h, Barnes-Hut self-gravity on the replicated tree, sink
// accretion + cull, sink creation) and the two transports it runs on.  Streams: s0 = the context's stream (set with sph_set_stream, so
// the *_dev calls do not synchronise), s1 = the communication stream.  Every message is packed on s0, an event lets
// s1 start, the grouped ncclSend / ncclRecv (or the all-gather) runs on s1, a second event lets s0 unpack -- and s0
// keeps computing between the two events (density, the interior wavefronts of the forces).
//
// The reference (/root/reference/SUMMER_SPH.f90:863-930) is one process; the step sequence reproduced is [F]:889-920
// ("SUMMER_SPH - Variable.f90":1120-1158 for variable-h contexts).
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_reduce_by_key.hpp>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/device/device_select.hpp>
#include <rocprim/iterator/counting_iterator.hpp>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <condition_variable>
#include <cstring>
#include <mutex>
#include <numeric>
#include <string>
#include <vector>

#include "../../include/summersph_halo.h"

namespace {

constexpr int MAXP = 64;                 // sph_select_boxes takes 64 boxes
constexpr int NF_MAX = 10;               // state fields that travel: 9, + h for variable-h contexts
constexpr int DT_SLOT = 192, PRED_SLOT = 193;
constexpr int ACC_PARTIALS = 7 * 64;     // SPH_ACC_PARTIALS: per sink m, m x, m y, m z, m vx, m vy, m vz of the accreted particles
const int32_t STATE[NF_MAX] = {SPH_F_X, SPH_F_Y, SPH_F_Z, SPH_F_VX, SPH_F_VY, SPH_F_VZ, SPH_F_U, SPH_F_M, SPH_F_ALPHA, SPH_F_H};

struct DevGuard {
    int prev = 0;
    explicit DevGuard(int d) { (void)hipGetDevice(&prev); (void)hipSetDevice(d); }
    ~DevGuard() { (void)hipSetDevice(prev); }
};

// grow-only device buffer
struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    hipError_t need(size_t bytes) {
        if (bytes <= cap) return hipSuccess;
        if (p) { hipError_t e = hipFree(p); p = nullptr; cap = 0; if (e != hipSuccess) return e; }      // hipFree waits for the device
        const size_t want = bytes + bytes / 4 + 4096;
        hipError_t e = hipMalloc(&p, want);
        if (e == hipSuccess) cap = want;
        return e;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
    template <class T> T *as() const { return static_cast<T *>(p); }
};

// ---- transports -----------------------------------------------------------------------------------------------------
struct Transport {
    std::string err;
    virtual ~Transport() {}
    // every rank contributes `bytes` from d_send; d_recv gets nranks blocks in rank order
    virtual int allgather(const void *d_send, void *d_recv, size_t bytes, hipStream_t s) = 0;
    // one grouped round: sb[q] bytes of d_send[q] to rank q, rb[q] bytes from rank q into d_recv[q] (0 = nothing; q may be
    // the caller itself)
    virtual int exchange(const void *const *d_send, const size_t *sb, void *const *d_recv, const size_t *rb, hipStream_t s) = 0;
};

struct RcclTransport : Transport {
    ncclComm_t comm = nullptr;
    bool own = false;
    int P = 1;
    ~RcclTransport() override { if (own && comm) (void)ncclCommDestroy(comm); }
    int fail(const char *what, ncclResult_t r) { err = std::string(what) + ": " + ncclGetErrorString(r); return SPH_ERR_HIP; }
    int allgather(const void *d_send, void *d_recv, size_t bytes, hipStream_t s) override {
        ncclResult_t r = ncclAllGather(d_send, d_recv, bytes, ncclChar, comm, s);
        return r == ncclSuccess ? SPH_OK : fail("ncclAllGather", r);
    }
    int exchange(const void *const *d_send, const size_t *sb, void *const *d_recv, const size_t *rb, hipStream_t s) override {
        ncclResult_t r = ncclGroupStart();
        if (r != ncclSuccess) return fail("ncclGroupStart", r);
        for (int q = 0; q < P; q++) {
            if (sb[q] > 0) { r = ncclSend(d_send[q], sb[q], ncclChar, q, comm, s); if (r != ncclSuccess) { (void)ncclGroupEnd(); return fail("ncclSend", r); } }
            if (rb[q] > 0) { r = ncclRecv(d_recv[q], rb[q], ncclChar, q, comm, s); if (r != ncclSuccess) { (void)ncclGroupEnd(); return fail("ncclRecv", r); } }
        }
        r = ncclGroupEnd();
        return r == ncclSuccess ? SPH_OK : fail("ncclGroupEnd", r);
    }
};

// the ranks are threads of one process on one device: a message is a device-to-device copy the receiver issues on its own
// stream once the sender's data is ready (event); the sender's stream in turn waits for the copies out of its buffers
struct Hub {
    int P;
    std::mutex m;
    std::condition_variable cv;
    int arrived = 0;
    uint64_t gen = 0;
    const void *ag[MAXP];
    const void *const *send[MAXP];
    const size_t *sb[MAXP];
    hipEvent_t ready[MAXP], done[MAXP];
    std::atomic<bool> failed{false};
    explicit Hub(int p) : P(p) {
        for (int q = 0; q < MAXP; q++) { ag[q] = nullptr; send[q] = nullptr; sb[q] = nullptr; ready[q] = nullptr; done[q] = nullptr; }
    }
    void barrier() {
        std::unique_lock<std::mutex> lk(m);
        const uint64_t g = gen;
        if (++arrived == P) { arrived = 0; gen++; cv.notify_all(); }
        else cv.wait(lk, [&] { return gen != g; });
    }
};

struct InprocTransport : Transport {
    Hub *hub = nullptr;
    int rank = 0;
    int finish(hipStream_t s, hipError_t e) {
        // everybody passes the same barriers whatever happened, so a failing rank cannot strand the others.  After the
        // barrier every rank's copies are queued and marked (done): this rank's stream then waits for the peers' copies out of
        // ITS buffers, so that -- as with RCCL -- the end of the operation on the stream means the send buffers are free.
        // No rank waits on the host.
        if (e == hipSuccess) e = hipEventRecord(hub->done[rank], s);
        if (e != hipSuccess) hub->failed = true;
        hub->barrier();
        for (int q = 0; q < hub->P && e == hipSuccess; q++) e = hipStreamWaitEvent(s, hub->done[q], 0);
        if (e != hipSuccess) { hub->failed = true; err = std::string("in-process transport: ") + hipGetErrorString(e); return SPH_ERR_HIP; }
        if (hub->failed) { err = "in-process transport: another rank failed"; return SPH_ERR_STATE; }
        return SPH_OK;
    }
    int allgather(const void *d_send, void *d_recv, size_t bytes, hipStream_t s) override {
        hipError_t e = hipEventRecord(hub->ready[rank], s);
        hub->ag[rank] = d_send;
        hub->barrier();
        for (int q = 0; q < hub->P && e == hipSuccess; q++) {
            e = hipStreamWaitEvent(s, hub->ready[q], 0);
            if (e == hipSuccess && bytes > 0)
                e = hipMemcpyAsync(static_cast<char *>(d_recv) + (size_t)q * bytes, hub->ag[q], bytes, hipMemcpyDeviceToDevice, s);
        }
        return finish(s, e);
    }
    int exchange(const void *const *d_send, const size_t *sb, void *const *d_recv, const size_t *rb, hipStream_t s) override {
        hipError_t e = hipEventRecord(hub->ready[rank], s);
        hub->send[rank] = d_send;
        hub->sb[rank] = sb;
        hub->barrier();
        bool mismatch = false;
        for (int q = 0; q < hub->P && e == hipSuccess; q++) {
            if (hub->sb[q][rank] != rb[q]) { mismatch = true; continue; }
            if (rb[q] == 0) continue;
            e = hipStreamWaitEvent(s, hub->ready[q], 0);
            if (e == hipSuccess) e = hipMemcpyAsync(d_recv[q], hub->send[q][rank], rb[q], hipMemcpyDeviceToDevice, s);
        }
        if (mismatch) hub->failed = true;
        const int st = finish(s, e);
        if (mismatch) { err = "in-process transport: send and receive sizes differ"; return SPH_ERR_STATE; }
        return st;
    }
};

// ---- small kernels of the orchestration ------------------------------------------------------------------------------
struct Edges { double e[MAXP]; int n; };

// owner of x: the number of interior edges <= x (torch.bucketize(x, edges, right=True) of dist.py)
__global__ void dest_kernel(const double *__restrict__ x, int64_t n, Edges ed, int32_t *__restrict__ dest, unsigned long long *__restrict__ counts) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double xi = x[i];
    int d = 0;
    for (int k = 0; k < ed.n; k++) d += (ed.e[k] <= xi) ? 1 : 0;
    dest[i] = d;
    atomicAdd(&counts[d], 1ULL);
}

__global__ void flag_kernel(const int32_t *__restrict__ dest, int64_t n, int32_t q, uint8_t *__restrict__ flags) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) flags[i] = dest[i] == q ? 1 : 0;
}

// dst[r * dst_stride + dst_off + k] = src[r * src_stride + (ids ? ids[k] : k)], r < nrows, k < cnt
__global__ void gather_rows(const double *__restrict__ src, int64_t src_stride, const int64_t *__restrict__ ids, int64_t cnt, int nrows,
                            double *__restrict__ dst, int64_t dst_stride, int64_t dst_off) {
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= cnt) return;
    const int64_t j = ids ? ids[k] : k;
    for (int r = 0; r < nrows; r++) dst[(int64_t)r * dst_stride + dst_off + k] = src[(int64_t)r * src_stride + j];
}

__global__ void gid_to_double(const int64_t *__restrict__ gid, const int64_t *__restrict__ ids, int64_t cnt, double *__restrict__ dst) {
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k < cnt) dst[k] = (double)gid[ids ? ids[k] : k];              // global numbers stay below 2^53: exact
}

__global__ void double_to_gid(const double *__restrict__ src, int64_t cnt, int64_t *__restrict__ dst) {
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k < cnt) dst[k] = (int64_t)src[k];
}

__global__ void gather_gid(const int64_t *__restrict__ gid, const int64_t *__restrict__ ids, int64_t cnt, int64_t *__restrict__ dst) {
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k < cnt) dst[k] = gid[ids[k]];
}

__global__ void iota64(int64_t *p, int64_t n) {
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n) p[k] = k;
}

// row = this rank's block with the dt candidate replaced by the minimum over ranks (end of a run, dist.py _finish_dt)
__global__ void min_dt_row(const double *__restrict__ all, int P, int rank, double *__restrict__ row) {
    const int k = threadIdx.x;
    if (k < SPH_PARTIALS) {
        double v = all[(size_t)rank * SPH_PARTIALS + k];
        if (k == DT_SLOT) for (int q = 0; q < P; q++) v = fmin(v, all[(size_t)q * SPH_PARTIALS + DT_SLOT]);
        row[k] = v;
    }
}

__global__ void pattern_kernel(double *p, int64_t n, double base) {
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n) p[k] = base + (double)k;
}

// largest of n positive doubles -> *out (bit patterns of positive doubles order like unsigned integers); *out zeroed before
__global__ void max_positive(const double *__restrict__ v, int64_t n, unsigned long long *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    double m = i < n ? v[i] : 0.0;
    for (int o = 32; o > 0; o >>= 1) m = fmax(m, __shfl_xor(m, o, 64));
    if ((threadIdx.x & 63) == 0 && m > 0.0) atomicMax(out, (unsigned long long)__double_as_longlong(m));
}

__global__ void set_header(double *p, double count) { if (threadIdx.x == 0) { p[0] = count; p[1] = 0.0; } }

// {x,y,z,m} records of one rank's block of the all-gathered sources: src[(off + k) * 4 + f] = blk[f * cnt + k]
__global__ void source_records(const double *__restrict__ blk, int64_t cnt, int64_t off, double *__restrict__ src) {
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= cnt) return;
    for (int f = 0; f < 4; f++) src[(off + k) * 4 + f] = blk[(int64_t)f * cnt + k];
}

// the candidate with the lowest particle number (first among equals, as torch.argmin): row of 9 doubles
__global__ void pick_candidate(const double *__restrict__ all, int P, double *__restrict__ out) {
    if (threadIdx.x != 0) return;
    int best = 0;
    for (int q = 1; q < P; q++) if (all[(size_t)q * 9] < all[(size_t)best * 9]) best = q;
    for (int k = 0; k < 9; k++) out[k] = all[(size_t)best * 9 + k];
}

// ---- locally essential tree (self-gravity on several ranks) -----------------------------------------------------------------
// The reference's octree over ALL particles ([F]:795-816: root = midpoint of the global bounding box, edge = its largest extent,
// 8-way splits) fixes every cell; a particle's path down it is a 63-bit key (3 bits per level, as csrc/gravity.hip computes it).
// A rank does not need the other ranks' particles one by one: a cell whose cube is so far from THIS rank's box that every
// target in the box accepts it whatever its centre of mass inside the cube (edge^2 / theta^2 < dmin^2, dmin = distance box--cube;
// [F]:278 tests edge / sqrt(|x - com|^2 + eps) < theta) is never opened here, so its sender ships it as ONE pseudo-particle
// {centre of mass, mass} -- the coarsest such cell on the path of each of its particles -- and ships single particles only where
// no cell on the path qualifies.  Acceptability is a property of (cell cube, receiver box): every sender cuts at the same cells,
// a cell is either opened by all or shipped whole by all, and the receiver's tree over {own particles, received particles and
// pseudo-particles} has, for every cell one of its targets opens, the same children occupancy, mass and centre of mass as the
// global tree (sums in another order: rounding).
struct LetRoot { double c[3]; double size; };
struct LetBoxes { double b[MAXP][6]; int n; };

__device__ __forceinline__ uint64_t let_key(const LetRoot &rb, double x, double y, double z) {
    double cx = rb.c[0], cy = rb.c[1], cz = rb.c[2], size = rb.size;
    uint64_t key = 0;
    for (int l = 0; l < 21; l++) {
        const int bx = x > cx, by = y > cy, bz = z > cz;
        key = (key << 3) | (uint64_t)(bx | (by << 1) | (bz << 2));
        const double q = 0.25 * size;
        cx = cx + (bx ? q : -q); cy = cy + (by ? q : -q); cz = cz + (bz ? q : -q);
        size = size * 0.5;
    }
    return key;
}

// xyzm: [4][n] (rows x, y, z, m)
__global__ void let_keys(LetRoot rb, const double *__restrict__ xyzm, int64_t n, uint64_t *__restrict__ keys, uint32_t *__restrict__ vals) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    keys[i] = let_key(rb, xyzm[i], xyzm[n + i], xyzm[2 * n + i]);
    vals[i] = (uint32_t)i;
}

// For receiver box b (blockIdx.y) and the particle at sorted position s: the level of the coarsest acceptable cell on its path
// (22: none, the particle travels as itself); head[b n + s] = 1 where a new exported item starts; mom = {m x, m y, m z, m}.
__global__ void let_cut(LetRoot rb, LetBoxes boxes, double inv_theta2, const uint64_t *__restrict__ keys, const uint32_t *__restrict__ vals,
                        const double *__restrict__ xyzm, int64_t n, int32_t *__restrict__ head, double4 *__restrict__ mom) {
    const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int b = blockIdx.y;
    if (s >= n) return;
    const double *bx = boxes.b[b];
    const uint64_t key = keys[s], prev = s > 0 ? keys[s - 1] : 0;
    // levels this particle shares with its predecessor in key order (the predecessor's cut is the same on shared cells)
    int lev = 22;
    double cx = rb.c[0], cy = rb.c[1], cz = rb.c[2], size = rb.size;
    for (int l = 0; l <= 21; l++) {
        // cell at level l: cube (cx, cy, cz) +- size / 2
        const double hx = 0.5 * size;
        const double dx = fmax(fmax(bx[0] - (cx + hx), (cx - hx) - bx[3]), 0.0), dy = fmax(fmax(bx[1] - (cy + hx), (cy - hx) - bx[4]), 0.0),
                     dz = fmax(fmax(bx[2] - (cz + hx), (cz - hx) - bx[5]), 0.0);
        const double dmin2 = dx * dx + dy * dy + dz * dz;
        if ((size * size) * inv_theta2 * (1.0 + 1e-9) < dmin2) { lev = l; break; }
        if (l == 21) break;
        const int ch = (int)((key >> (3 * (20 - l))) & 7);
        const double q = 0.25 * size;
        cx = cx + ((ch & 1) ? q : -q); cy = cy + ((ch & 2) ? q : -q); cz = cz + ((ch & 4) ? q : -q);
        size = size * 0.5;
    }
    // same item as the predecessor iff both lie in the same acceptable cell: they share the first 3 lev bits of the key (then the
    // predecessor's cut stops at the same cell)
    bool first = s == 0 || lev == 22;
    if (!first) first = lev == 0 ? false : ((key ^ prev) >> (3 * (21 - lev))) != 0;
    head[(int64_t)b * n + s] = first ? 1 : 0;
    const uint32_t i = vals[s];
    const double m = xyzm[3 * n + i];
    mom[(int64_t)b * n + s] = make_double4(m * xyzm[i], m * xyzm[n + i], m * xyzm[2 * n + i], m);
}

struct Add4 { __device__ double4 operator()(const double4 &a, const double4 &b) const { return make_double4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); } };

// segment sums {m x, m y, m z, m} -> records {x, y, z, m} (a single particle comes out as itself up to one rounding of m x / m)
__global__ void let_records(const double4 *__restrict__ sums, int64_t cnt, double4 *__restrict__ rec) {
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= cnt) return;
    const double4 v = sums[k];
    rec[k] = v.w > 0.0 ? make_double4(v.x / v.w, v.y / v.w, v.z / v.w, v.w) : make_double4(0.0, 0.0, 0.0, 0.0);
}

// own particles as {x, y, z, m} records: rec[k] = {xyzm[k], xyzm[n + k], ...}
__global__ void let_own_records(const double *__restrict__ xyzm, int64_t n, double4 *__restrict__ rec) {
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n) rec[k] = make_double4(xyzm[k], xyzm[n + k], xyzm[2 * n + k], xyzm[3 * n + k]);
}

inline dim3 blocks_for(int64_t n) { return dim3((unsigned)((n + 255) / 256)); }

}  // namespace

struct sph_halo {
    sph_ctx *c = nullptr;
    int rank = 0, P = 1, device = 0;
    Transport *tr = nullptr;
    hipStream_t s0 = nullptr, s1 = nullptr;
    bool own_s1 = true;
    hipEvent_t e01 = nullptr, e10 = nullptr, e_pred = nullptr;
    double h = 0.0;
    bool variable = false, gravity = false, accrete = false, sink_creation = false;   // the context's flags
    bool octree = false;                 // gravity || variable: the shared octree needs every rank's exact box and all particles
    int nf = 9;                          // state fields per particle (10 with h)
    int64_t counts_all[MAXP];            // owned particles of every rank
    bool counts_valid = false, sources_valid = false;
    std::vector<double> last_boxes;      // every rank's exact owned box at the last ghost exchange (host)
    Edges edges{};
    int migrate_every = 32, since_migrate = 0;
    int64_t n_owned = 0, reserved = 0;
    bool uploaded = false;
    bool pos_dirty = true, vel_dirty = false, dt_pending = false, pred_for_drift = false, pred_valid = false;
    DevBuf gid, gid_new, own, newbuf, part, allpart, row, box, boxes, cnt, cntall, ghosts, dest, flags, keep_ids, sel_tmp, sel_count;
    DevBuf src, srcmine, srcall, accp, accall, keep, cand, candall, gnum;
    // locally essential tree: this rank's sorted path keys, the cut per receiver, the exported items
    bool let = false;                    // self-gravity on several ranks through the LET exchange (fixed h) instead of the all-gather
    double theta = 0.5;
    DevBuf lkeys, lkeys2, lvals, lvals2, lsort, lhead, lseg, lmom, lsum, lukeys, lcount, lscan, lrec;
    DevBuf sendb[MAXP], recvb[MAXP], ids[MAXP];
    double *pin = nullptr;               // pinned host memory: the gathered partials, the boxes, the count matrix
    double *pin_part = nullptr, *pin_boxes = nullptr;
    int64_t *pin_cnt = nullptr, *pin_row = nullptr;
    int64_t send_count[MAXP], ghost_first[MAXP], ghost_count[MAXP];
    int64_t cap_send[MAXP], cap_recv[MAXP];   // ghost messages: particles the next message to / from a peer has room for (both sides agree)
    double *pin_hdr = nullptr;               // pinned: headers out [0, P), headers in [P, 2P)
    bool refresh_pending = false;
    int refresh_nf = 0;
    int32_t refresh_fields[NF_MAX];
    sph_halo_stats st{};
    std::string err;
};

namespace {

#define H_TRY(call)                                                                                              \
    do {                                                                                                         \
        const int st_ = (call);                                                                                  \
        if (st_ != SPH_OK) {                                                                                     \
            const char *m_ = sph_last_error(h->c);                                                               \
            h->err = std::string(#call) + ": " + ((m_ && *m_) ? m_ : sph_strerror(st_));                        \
            return st_;                                                                                          \
        }                                                                                                        \
    } while (0)
#define H_HIP(call)                                                                                              \
    do {                                                                                                         \
        const hipError_t e_ = (call);                                                                            \
        if (e_ != hipSuccess) { h->err = std::string(#call) + ": " + hipGetErrorString(e_); return SPH_ERR_HIP; } \
    } while (0)
#define H_TR(call)                                                                                               \
    do {                                                                                                         \
        const int st_ = (call);                                                                                  \
        if (st_ != SPH_OK) { h->err = h->tr->err; return st_; }                                                  \
    } while (0)

// s1 continues after what s0 has queued so far / s0 continues after what s1 has queued so far
int s0_then_s1(sph_halo *h) { H_HIP(hipEventRecord(h->e01, h->s0)); H_HIP(hipStreamWaitEvent(h->s1, h->e01, 0)); return SPH_OK; }
int s1_then_s0(sph_halo *h) { H_HIP(hipEventRecord(h->e10, h->s1)); H_HIP(hipStreamWaitEvent(h->s0, h->e10, 0)); return SPH_OK; }

int host_wait(sph_halo *h, hipStream_t s) {
    H_HIP(hipStreamSynchronize(s));
    h->st.host_waits++;
    return SPH_OK;
}

int common_init(sph_halo *h) {
    if (h->P < 1 || h->P > MAXP || h->rank < 0 || h->rank >= h->P) { h->err = "1 <= nranks <= 64, 0 <= rank < nranks"; return SPH_ERR_ARG; }
    sph_params p;
    H_TRY(sph_get_params(h->c, &p));
    h->variable = (p.flags & SPH_FLAG_VARIABLE_H) != 0;
    h->gravity = (p.flags & SPH_FLAG_SELF_GRAVITY) != 0;
    h->accrete = (p.flags & SPH_FLAG_ACCRETE_CULL) != 0;
    h->sink_creation = (p.flags & SPH_FLAG_SINK_CREATION) != 0 && h->variable;
    h->octree = h->gravity || h->variable;
    h->nf = h->variable ? 10 : 9;
    h->theta = p.theta;
    // fixed h: the gravity sources of the other ranks arrive as a locally essential tree (SPH_HALO_REPLICATED=1: every particle of
    // every rank, as dist.py does; variable h keeps that -- the leaf boxes of the ghosts need their key neighbours one by one)
    h->let = h->gravity && !h->variable && h->P > 1 && getenv("SPH_HALO_REPLICATED") == nullptr;
    if (h->accrete && h->P > 1 && !h->octree) {
        // the accretion test walks the octree of ALL particles: it needs the all-gathered sources of the self-gravity path
        h->err = "sph_halo: accretion on several ranks needs SPH_FLAG_SELF_GRAVITY or SPH_FLAG_VARIABLE_H (the shared octree)";
        return SPH_ERR_ARG;
    }
    h->h = p.h;
    int dev = 0;
    if (hipStreamGetDevice(reinterpret_cast<hipStream_t>(sph_stream(h->c)), &dev) != hipSuccess) (void)hipGetDevice(&dev);
    h->device = dev;
    DevGuard g(dev);
    H_HIP(hipStreamCreateWithFlags(&h->s0, hipStreamNonBlocking));
    if (!h->s1) { H_HIP(hipStreamCreateWithFlags(&h->s1, hipStreamNonBlocking)); h->own_s1 = true; }
    H_HIP(hipEventCreateWithFlags(&h->e01, hipEventDisableTiming));
    H_HIP(hipEventCreateWithFlags(&h->e10, hipEventDisableTiming));
    H_HIP(hipEventCreateWithFlags(&h->e_pred, hipEventDisableTiming));
    H_TRY(sph_set_stream(h->c, h->s0));
    H_TRY(sph_set_rank(h->c, h->rank, h->P));
    const size_t nd = (size_t)h->P * (SPH_PARTIALS + 8) + (size_t)h->P * h->P + h->P + 64 + 4 * (size_t)h->P;
    H_HIP(hipHostMalloc(reinterpret_cast<void **>(&h->pin), nd * sizeof(double), hipHostMallocDefault));
    h->pin_part = h->pin;
    h->pin_boxes = h->pin_part + (size_t)h->P * SPH_PARTIALS;
    h->pin_cnt = reinterpret_cast<int64_t *>(h->pin_boxes + (size_t)h->P * 8);
    h->pin_row = h->pin_cnt + (size_t)h->P * h->P;
    h->pin_hdr = reinterpret_cast<double *>(h->pin_row + h->P + 8);
    for (int q = 0; q < MAXP; q++) { h->send_count[q] = 0; h->ghost_first[q] = 0; h->ghost_count[q] = 0; h->cap_send[q] = 0; h->cap_recv[q] = 0; h->counts_all[q] = 0; }
    h->edges.n = 0;
    return SPH_OK;
}

// every rank's row of P int64 -> the P x P matrix on the host (row = sender)
int gather_counts(sph_halo *h, const int64_t *mine, int64_t *matrix) {
    const int P = h->P;
    if (P == 1) { matrix[0] = mine[0]; return SPH_OK; }
    H_HIP(h->cnt.need((size_t)P * 8));
    H_HIP(h->cntall.need((size_t)P * P * 8));
    for (int q = 0; q < P; q++) h->pin_row[q] = mine[q];
    H_HIP(hipMemcpyAsync(h->cnt.p, h->pin_row, (size_t)P * 8, hipMemcpyHostToDevice, h->s1));
    H_TR(h->tr->allgather(h->cnt.p, h->cntall.p, (size_t)P * 8, h->s1));
    h->st.collectives++;
    H_HIP(hipMemcpyAsync(h->pin_cnt, h->cntall.p, (size_t)P * P * 8, hipMemcpyDeviceToHost, h->s1));
    if (int st = host_wait(h, h->s1)) return st;
    for (int k = 0; k < P * P; k++) matrix[k] = h->pin_cnt[k];
    return SPH_OK;
}

// one grouped round of `width` rows per particle: sendb[q] holds width * send_n[q] doubles, recvb[q] gets width * recv_n[q]
int p2p(sph_halo *h, const int64_t *send_n, const int64_t *recv_n, int width) {
    const void *sp[MAXP];
    void *rp[MAXP];
    size_t sb[MAXP], rb[MAXP];
    for (int q = 0; q < h->P; q++) {
        sb[q] = (size_t)send_n[q] * width * 8;
        rb[q] = (size_t)recv_n[q] * width * 8;
        if (rb[q]) H_HIP(h->recvb[q].need(rb[q]));
        sp[q] = h->sendb[q].p;
        rp[q] = h->recvb[q].p;
    }
    H_TR(h->tr->exchange(sp, sb, rp, rb, h->s1));
    h->st.exchanges++;
    return SPH_OK;
}

// one grouped round with explicit pointers and byte counts per peer
int p2p_raw(sph_halo *h, const void *const *sp, const size_t *sb, void *const *rp, const size_t *rb) {
    H_TR(h->tr->exchange(sp, sb, rp, rb, h->s1));
    bool any = false;
    for (int q = 0; q < h->P; q++) any |= sb[q] > 0 || rb[q] > 0;
    if (any) h->st.exchanges++;                  // an empty group is not a round
    return SPH_OK;
}

inline int64_t ghost_capacity(int64_t count) { return count + count / 4 + 256; }

int ensure_reserve(sph_halo *h, int64_t n) {
    if (n + n / 8 + 32768 > h->reserved) {            // room for the ghost swaps; grows rarely (a re-allocation)
        h->reserved = n + n / 4 + 65536;
        H_TRY(sph_reserve(h->c, h->reserved));
    }
    return SPH_OK;
}

// particles that left their slab change owner (dist.py _migrate)
int migrate(sph_halo *h) {
    const int P = h->P, NF = h->nf;
    const int64_t n = h->n_owned;
    h->st.migrations++;
    H_HIP(h->own.need((size_t)std::max<int64_t>(n, 1) * NF * 8));
    H_HIP(h->dest.need((size_t)std::max<int64_t>(n, 1) * 4));
    H_HIP(h->cnt.need((size_t)std::max(P, 2) * 8));
    if (n > 0) H_TRY(sph_gather_fields_dev(h->c, NF, STATE, n, nullptr, h->own.as<double>()));
    H_HIP(hipMemsetAsync(h->cnt.p, 0, (size_t)P * 8, h->s0));
    if (n > 0) {
        dest_kernel<<<blocks_for(n), 256, 0, h->s0>>>(h->own.as<double>(), n, h->edges, h->dest.as<int32_t>(), h->cnt.as<unsigned long long>());
        H_HIP(hipGetLastError());
    }
    int64_t mine[MAXP];
    H_HIP(hipMemcpyAsync(h->pin_row, h->cnt.p, (size_t)P * 8, hipMemcpyDeviceToHost, h->s0));
    if (int st = host_wait(h, h->s0)) return st;
    for (int q = 0; q < P; q++) mine[q] = h->pin_row[q];
    std::vector<int64_t> cm((size_t)P * P);
    if (int st = gather_counts(h, mine, cm.data())) return st;
    int64_t moved = 0;
    for (int a = 0; a < P; a++) for (int b = 0; b < P; b++) if (a != b) moved += cm[(size_t)a * P + b];
    if (moved == 0) return SPH_OK;                     // nobody moved anywhere: the contexts stay as they are

    int64_t out[MAXP], inc[MAXP];
    for (int q = 0; q < P; q++) { out[q] = q == h->rank ? 0 : cm[(size_t)h->rank * P + q]; inc[q] = q == h->rank ? 0 : cm[(size_t)q * P + h->rank]; }
    const int64_t n_keep = cm[(size_t)h->rank * P + h->rank];
    H_HIP(h->flags.need((size_t)std::max<int64_t>(n, 1)));
    H_HIP(h->sel_count.need(8));
    size_t tmp_bytes = 0;
    if (n > 0) {
        H_HIP(rocprim::select(nullptr, tmp_bytes, rocprim::counting_iterator<int64_t>(0), h->flags.as<uint8_t>(), h->keep_ids.as<int64_t>(),
                              h->sel_count.as<size_t>(), (size_t)n, h->s0));
        H_HIP(h->sel_tmp.need(tmp_bytes));
    }
    auto select_dest = [&](int q, DevBuf &ids, int64_t expect) -> int {
        H_HIP(ids.need((size_t)std::max<int64_t>(expect, 1) * 8));
        if (expect == 0) return SPH_OK;
        flag_kernel<<<blocks_for(n), 256, 0, h->s0>>>(h->dest.as<int32_t>(), n, q, h->flags.as<uint8_t>());
        H_HIP(hipGetLastError());
        size_t tb = tmp_bytes;
        H_HIP(rocprim::select(h->sel_tmp.p, tb, rocprim::counting_iterator<int64_t>(0), h->flags.as<uint8_t>(), ids.as<int64_t>(),
                              h->sel_count.as<size_t>(), (size_t)n, h->s0));
        return SPH_OK;
    };
    // payload per destination: the 9 state rows + the global number as a 10th row
    for (int q = 0; q < P; q++) {
        if (out[q] == 0) continue;
        if (int st = select_dest(q, h->ids[q], out[q])) return st;
        H_HIP(h->sendb[q].need((size_t)out[q] * (NF + 1) * 8));
        gather_rows<<<blocks_for(out[q]), 256, 0, h->s0>>>(h->own.as<double>(), n, h->ids[q].as<int64_t>(), out[q], NF, h->sendb[q].as<double>(), out[q], 0);
        gid_to_double<<<blocks_for(out[q]), 256, 0, h->s0>>>(h->gid.as<int64_t>(), h->ids[q].as<int64_t>(), out[q], h->sendb[q].as<double>() + (size_t)NF * out[q]);
        H_HIP(hipGetLastError());
    }
    if (int st = select_dest(h->rank, h->keep_ids, n_keep)) return st;
    if (int st = s0_then_s1(h)) return st;
    if (int st = p2p(h, out, inc, NF + 1)) return st;
    if (int st = s1_then_s0(h)) return st;
    int64_t n_new = n_keep;
    for (int q = 0; q < P; q++) n_new += inc[q];
    H_HIP(h->newbuf.need((size_t)std::max<int64_t>(n_new, 1) * NF * 8));
    H_HIP(h->gid_new.need((size_t)std::max<int64_t>(n_new, 1) * 8));
    if (n_keep > 0) {
        gather_rows<<<blocks_for(n_keep), 256, 0, h->s0>>>(h->own.as<double>(), n, h->keep_ids.as<int64_t>(), n_keep, NF, h->newbuf.as<double>(), n_new, 0);
        gather_gid<<<blocks_for(n_keep), 256, 0, h->s0>>>(h->gid.as<int64_t>(), h->keep_ids.as<int64_t>(), n_keep, h->gid_new.as<int64_t>());
        H_HIP(hipGetLastError());
    }
    int64_t off = n_keep;
    for (int q = 0; q < P; q++) {
        if (inc[q] == 0) continue;
        gather_rows<<<blocks_for(inc[q]), 256, 0, h->s0>>>(h->recvb[q].as<double>(), inc[q], nullptr, inc[q], NF, h->newbuf.as<double>(), n_new, off);
        double_to_gid<<<blocks_for(inc[q]), 256, 0, h->s0>>>(h->recvb[q].as<double>() + (size_t)NF * inc[q], inc[q], h->gid_new.as<int64_t>() + off);
        H_HIP(hipGetLastError());
        off += inc[q];
    }
    if (int st = ensure_reserve(h, n_new)) return st;
    const double *r = h->newbuf.as<double>();
    H_TRY(sph_upload_dev(h->c, n_new, r, r + n_new, r + 2 * n_new, r + 3 * n_new, r + 4 * n_new, r + 5 * n_new, r + 6 * n_new, r + 7 * n_new, r + 8 * n_new));
    if (h->variable && n_new > 0) H_TRY(sph_upload_field_dev(h->c, SPH_F_H, r + 9 * n_new, n_new));
    std::swap(h->gid, h->gid_new);
    h->n_owned = n_new;
    h->st.migrated += moved;
    h->counts_valid = false; h->sources_valid = false;
    for (int q = 0; q < P; q++) { h->send_count[q] = 0; h->ghost_count[q] = 0; }
    return SPH_OK;
}

// who needs which of my particles, ship them, swap them in (dist.py _exchange_ghosts)
int exchange_ghosts(sph_halo *h) {
    const int P = h->P, NF = h->nf;
    const int W = NF + (h->variable ? 1 : 0);       // variable h: + the global particle number of every ghost ([V]:383 compares numbers)
    // the octree paths (self-gravity, variable h) need the exact global box: no predicted boxes for them
    bool use_pred = h->pred_for_drift && h->pred_valid && !h->octree;
    h->pred_for_drift = false;
    std::vector<double> boxes((size_t)P * 6);
    double hmax_all = 0.0;
    if (use_pred) {
        H_HIP(hipEventSynchronize(h->e_pred));
        h->st.host_waits++;
        for (int q = 0; q < P; q++)
            for (int a = 0; a < 6; a++) {
                const double v = h->pin_part[(size_t)q * SPH_PARTIALS + PRED_SLOT + a];
                if (std::isnan(v)) use_pred = false;             // some rank had no prediction
                boxes[(size_t)q * 6 + a] = v;
            }
    }
    if (!use_pred) {
        // every rank's box (variable h: and its largest h -- i and j interact within 2 max(h_i, h_j), so the ghost layer is as
        // wide as twice the largest h anywhere): 8 doubles per rank
        H_HIP(h->box.need(8 * 8));
        H_HIP(h->boxes.need((size_t)P * 8 * 8));
        H_HIP(hipMemsetAsync(h->box.p, 0, 8 * 8, h->s0));
        H_TRY(sph_owned_bbox(h->c, nullptr, h->box.as<double>()));
        if (h->variable && h->n_owned > 0) {
            static const int32_t HF[1] = {SPH_F_H};
            H_HIP(h->own.need((size_t)h->n_owned * 8));
            H_TRY(sph_gather_fields_dev(h->c, 1, HF, h->n_owned, nullptr, h->own.as<double>()));
            max_positive<<<blocks_for(h->n_owned), 256, 0, h->s0>>>(h->own.as<double>(), h->n_owned, reinterpret_cast<unsigned long long *>(h->box.as<double>() + 6));
            H_HIP(hipGetLastError());
        }
        if (int st = s0_then_s1(h)) return st;
        H_TR(h->tr->allgather(h->box.p, h->boxes.p, 8 * 8, h->s1));
        h->st.collectives++;
        H_HIP(hipMemcpyAsync(h->pin_boxes, h->boxes.p, (size_t)P * 8 * 8, hipMemcpyDeviceToHost, h->s1));
        if (int st = host_wait(h, h->s1)) return st;
        for (int q = 0; q < P; q++) {
            for (int a = 0; a < 6; a++) boxes[(size_t)q * 6 + a] = h->pin_boxes[(size_t)q * 8 + a];
            hmax_all = std::max(hmax_all, h->pin_boxes[(size_t)q * 8 + 6]);
        }
        h->last_boxes = boxes;
    }
    const double r = 2.0 * (h->variable ? hmax_all : h->h) * (1.0 + 1e-9);
    auto finite6 = [&](int q) { for (int a = 0; a < 6; a++) if (!std::isfinite(boxes[(size_t)q * 6 + a])) return false; return true; };
    const bool mine_ok = finite6(h->rank);
    const double *me = &boxes[(size_t)h->rank * 6];
    int peers[MAXP], npeers = 0;
    double sel[MAXP * 6];
    for (int q = 0; q < P; q++) {
        h->send_count[q] = 0;
        if (q == h->rank || !mine_ok || !finite6(q)) continue;
        const double *b = &boxes[(size_t)q * 6];
        // ONE expression, evaluated identically by both ranks of a pair (operands ordered by rank): each side posts a
        // receive for the other's header, so a decision that differed by an ulp at gap == r would leave a send without
        // its partner
        const double *lo_rank = q < h->rank ? b : me, *hi_rank = q < h->rank ? me : b;
        bool touch = true;
        for (int a = 0; a < 3; a++) {
            const double gap = std::max(hi_rank[a] - lo_rank[3 + a], lo_rank[a] - hi_rank[3 + a]);
            if (!(gap <= r)) touch = false;
        }
        if (!touch) continue;
        for (int a = 0; a < 3; a++) { sel[npeers * 6 + a] = b[a] - r; sel[npeers * 6 + 3 + a] = b[3 + a] + r; }
        peers[npeers++] = q;
    }
    // The selection does not wait for its counts (sph_select_boxes_async): the payload is packed for the agreed room by a
    // kernel that reads the count on the device, and the host learns both its own counts and the peers' at the one wait below.
    // (The octree paths wait for their counts -- sph_select_boxes -- because the ghosts' global numbers travel as an extra row
    // that is packed from this object's own list; they wait for the exact boxes anyway.)
    const bool sync_select = h->variable;
    int64_t counts[MAXP], selc[MAXP];
    for (int q = 0; q < P; q++) counts[q] = 0;
    if (npeers > 0) {
        if (sync_select) { H_TRY(sph_select_boxes(h->c, npeers, sel, selc)); h->st.host_waits++; }
        else H_TRY(sph_select_boxes_async(h->c, npeers, sel));
    }
    // The payload travels without a size exchange: both sides of a pair agree on the room the message has (cap_send here =
    // cap_recv there, derived from the count of the last message between the two, 0 at first), the first two doubles say
    // how many particles there are.  Round A: header + rows in that room.  Round B, for the pairs whose count did not fit
    // (a first contact, a jump): the rows again at their exact size -- both sides know, the sender from its count, the
    // receiver from the header.  Steady state: one round and one read-back (headers + own counts), no collective.
    bool touching[MAXP];
    for (int q = 0; q < P; q++) touching[q] = false;
    for (int b = 0; b < npeers; b++) touching[peers[b]] = true;
    const void *sp[MAXP];
    void *rp[MAXP];
    size_t sb[MAXP], rb[MAXP];
    for (int q = 0; q < P; q++) { sp[q] = nullptr; rp[q] = nullptr; sb[q] = 0; rb[q] = 0; }
    // rows of a message of `cnt` particles behind the 2-double header: NF state rows [+ the numbers], row stride cnt
    auto pack_rows = [&](int q, int64_t cnt, double *dst) -> int {
        H_TRY(sph_gather_fields_dev(h->c, NF, STATE, cnt, h->ids[q].as<int64_t>(), dst));
        if (W > NF) {
            gid_to_double<<<blocks_for(cnt), 256, 0, h->s0>>>(h->gid.as<int64_t>(), h->ids[q].as<int64_t>(), cnt, dst + (size_t)NF * cnt);
            H_HIP(hipGetLastError());
        }
        return SPH_OK;
    };
    for (int b = 0; b < npeers; b++) {
        const int q = peers[b];
        H_HIP(h->sendb[q].need((size_t)(2 + (W + 1) * std::max<int64_t>(h->cap_send[q], 1)) * 8));
        H_HIP(h->recvb[q].need((size_t)(2 + W * h->cap_recv[q]) * 8));
        if (sync_select) {
            counts[q] = selc[b];
            set_header<<<1, 64, 0, h->s0>>>(h->sendb[q].as<double>(), (double)counts[q]);
            H_HIP(hipGetLastError());
            if (counts[q] > 0) {
                H_HIP(h->ids[q].need((size_t)counts[q] * 8));
                H_TRY(sph_selected_ids_dev(h->c, b, counts[q], h->ids[q].as<int64_t>()));
                if (counts[q] <= h->cap_send[q]) if (int st = pack_rows(q, counts[q], h->sendb[q].as<double>() + 2)) return st;
            }
        } else {
            H_TRY(sph_gather_selected_dev(h->c, b, NF, STATE, h->cap_send[q], h->sendb[q].as<double>()));
        }
        sp[q] = h->sendb[q].p; sb[q] = (size_t)(2 + W * h->cap_send[q]) * 8;
        rp[q] = h->recvb[q].p; rb[q] = (size_t)(2 + W * h->cap_recv[q]) * 8;
    }
    if (int st = s0_then_s1(h)) return st;
    if (int st = p2p_raw(h, sp, sb, rp, rb)) return st;
    double *hdr_in = h->pin_hdr + 2 * P;
    for (int b = 0; b < npeers; b++)
        H_HIP(hipMemcpyAsync(hdr_in + peers[b], h->recvb[peers[b]].p, 8, hipMemcpyDeviceToHost, h->s1));
    if (npeers > 0) if (int st = host_wait(h, h->s1)) return st;
    if (npeers > 0 && !sync_select) {
        H_TRY(sph_selected_counts(h->c, npeers, selc));        // they arrived before the headers (same wait)
        for (int b = 0; b < npeers; b++) counts[peers[b]] = selc[b];
    }
    int64_t rc[MAXP], total = 0, roff[MAXP];
    for (int q = 0; q < P; q++) { rc[q] = touching[q] ? (int64_t)hdr_in[q] : 0; roff[q] = 2; total += rc[q]; }
    // the id lists of what the peers now hold as ghosts (the field refreshes of this evaluation gather by them)
    for (int b = 0; b < npeers; b++) {
        const int q = peers[b];
        h->send_count[q] = counts[q];
        if (counts[q] == 0 || sync_select) continue;
        H_HIP(h->ids[q].need((size_t)counts[q] * 8));
        H_TRY(sph_selected_ids_dev(h->c, b, counts[q], h->ids[q].as<int64_t>()));
    }
    // round B (every rank calls it, an empty group costs nothing): the pairs that did not fit
    for (int q = 0; q < P; q++) { sp[q] = nullptr; rp[q] = nullptr; sb[q] = 0; rb[q] = 0; }
    bool resend = false;
    for (int b = 0; b < npeers; b++) {
        const int q = peers[b];
        if (counts[q] > h->cap_send[q]) {
            H_HIP(h->sendb[q].need((size_t)(2 + (W + 1) * counts[q]) * 8));      // (re-allocation waits for the device: round A is complete)
            if (int st = pack_rows(q, counts[q], h->sendb[q].as<double>() + 2)) return st;
            sp[q] = h->sendb[q].as<double>() + 2; sb[q] = (size_t)W * counts[q] * 8;
            resend = true;
        }
        if (rc[q] > h->cap_recv[q]) {
            H_HIP(h->recvb[q].need((size_t)W * rc[q] * 8));
            rp[q] = h->recvb[q].p; rb[q] = (size_t)W * rc[q] * 8;
            roff[q] = 0;
        }
        h->cap_send[q] = ghost_capacity(counts[q]);
        h->cap_recv[q] = ghost_capacity(rc[q]);
    }
    if (resend) if (int st = s0_then_s1(h)) return st;
    if (int st = p2p_raw(h, sp, sb, rp, rb)) return st;
    if (int st = s1_then_s0(h)) return st;
    // every ghost lies inside its owner's box: particles farther than 2h from all of them cannot have a ghost neighbour
    // (their forces do not wait for the ghost fields)
    double bnd[MAXP * 6];
    int nb = 0;
    for (int q = 0; q < P; q++) if (rc[q] > 0) { for (int a = 0; a < 6; a++) bnd[nb * 6 + a] = boxes[(size_t)q * 6 + a]; nb++; }
    H_TRY(sph_set_boundary_boxes(h->c, nb, bnd));
    if (h->n_owned + total > h->reserved) {
        h->err = "sph_halo: more ghosts than the reserved slots hold (n_owned + ghosts > n_owned * 5/4 + 65536)";
        return SPH_ERR_NOMEM;
    }
    H_HIP(h->ghosts.need((size_t)std::max<int64_t>(total, 1) * W * 8));
    int64_t first = h->n_owned, off = 0;
    for (int q = 0; q < P; q++) {
        h->ghost_first[q] = first;
        h->ghost_count[q] = rc[q];
        if (rc[q] > 0) {
            gather_rows<<<blocks_for(rc[q]), 256, 0, h->s0>>>(h->recvb[q].as<double>() + roff[q], rc[q], nullptr, rc[q], W, h->ghosts.as<double>(), total, off);
            H_HIP(hipGetLastError());
        }
        first += rc[q];
        off += rc[q];
    }
    h->st.ghosts = total;
    H_TRY(sph_replace_ghosts_dev(h->c, total, h->ghosts.as<double>()));        // rows 0 .. NF-1 (row stride `total`)
    if (h->variable) {
        // the pair rule of [V]:383 compares the reference's particle numbers: the owned particles' and the ghosts'
        if (h->n_owned > 0) H_TRY(sph_set_numbers_dev(h->c, 0, h->n_owned, h->gid.as<int64_t>()));
        if (total > 0) {
            H_HIP(h->gnum.need((size_t)total * 8));
            double_to_gid<<<blocks_for(total), 256, 0, h->s0>>>(h->ghosts.as<double>() + (size_t)NF * total, total, h->gnum.as<int64_t>());
            H_HIP(hipGetLastError());
            H_TRY(sph_set_numbers_dev(h->c, h->n_owned, total, h->gnum.as<int64_t>()));
        }
    }
    return SPH_OK;
}

// Self-gravity is long range, and the variable-h neighbour rule needs the octree leaf of every particle: every rank gets
// {x, y, z, m} of ALL particles (one all-gather, padded to the largest rank) and builds the same octree as the undecomposed
// run would -- same particles, same root box, hence the same nodes and the same accepted set for every target -- and walks it
// for its own particles.  The tree build is replicated work; the walk, which dominates, is shared.  (dist.py _gravity_sources)
int gravity_sources(sph_halo *h) {
    static const int32_t XYZM[4] = {SPH_F_X, SPH_F_Y, SPH_F_Z, SPH_F_M};
    const int P = h->P;
    if (!h->counts_valid) {
        H_HIP(h->cnt.need((size_t)std::max(P, 2) * 8));
        H_HIP(h->cntall.need((size_t)P * P * 8));
        h->pin_row[0] = h->n_owned;
        H_HIP(hipMemcpyAsync(h->cnt.p, h->pin_row, 8, hipMemcpyHostToDevice, h->s1));
        H_TR(h->tr->allgather(h->cnt.p, h->cntall.p, 8, h->s1));
        h->st.collectives++;
        H_HIP(hipMemcpyAsync(h->pin_cnt, h->cntall.p, (size_t)P * 8, hipMemcpyDeviceToHost, h->s1));
        if (int st = host_wait(h, h->s1)) return st;
        for (int q = 0; q < P; q++) h->counts_all[q] = h->pin_cnt[q];
        h->counts_valid = true;
    }
    int64_t maxn = 1, total = 0;
    for (int q = 0; q < P; q++) { maxn = std::max(maxn, h->counts_all[q]); total += h->counts_all[q]; }
    const size_t blk = (size_t)4 * maxn * 8;
    H_HIP(h->srcmine.need(blk));
    H_HIP(h->srcall.need(blk * P));
    H_HIP(h->src.need((size_t)std::max<int64_t>(total, 1) * 4 * 8));
    if (h->n_owned > 0) H_TRY(sph_gather_fields_dev(h->c, 4, XYZM, h->n_owned, nullptr, h->srcmine.as<double>()));     // [4][n_owned], compact
    if (int st = s0_then_s1(h)) return st;
    H_TR(h->tr->allgather(h->srcmine.p, h->srcall.p, blk, h->s1));
    h->st.collectives++;
    if (int st = s1_then_s0(h)) return st;
    int64_t off = 0;
    double lo_hi[6] = {INFINITY, INFINITY, INFINITY, -INFINITY, -INFINITY, -INFINITY};
    for (int q = 0; q < P; q++) {
        const int64_t cq = h->counts_all[q];
        if (cq == 0) continue;
        source_records<<<blocks_for(cq), 256, 0, h->s0>>>(h->srcall.as<double>() + (size_t)q * 4 * maxn, cq, off, h->src.as<double>());
        H_HIP(hipGetLastError());
        off += cq;
        for (int a = 0; a < 3; a++) {
            lo_hi[a] = std::min(lo_hi[a], h->last_boxes[(size_t)q * 6 + a]);
            lo_hi[3 + a] = std::max(lo_hi[3 + a], h->last_boxes[(size_t)q * 6 + 3 + a]);
        }
    }
    H_TRY(sph_set_gravity_sources_dev(h->c, total, h->src.as<double>(), lo_hi));
    return SPH_OK;
}

// ship the listed fields of the particles my peers hold as ghosts: packed on s0, sent on s1 while s0 goes on
int refresh_start(sph_halo *h, int nf, const int32_t *fields) {
    for (int q = 0; q < h->P; q++) {
        if (h->send_count[q] == 0) continue;
        H_HIP(h->sendb[q].need((size_t)h->send_count[q] * nf * 8));
        H_TRY(sph_gather_fields_dev(h->c, nf, fields, h->send_count[q], h->ids[q].as<int64_t>(), h->sendb[q].as<double>()));
    }
    if (int st = s0_then_s1(h)) return st;
    if (int st = p2p(h, h->send_count, h->ghost_count, nf)) return st;
    H_HIP(hipEventRecord(h->e10, h->s1));
    h->refresh_pending = true;
    h->refresh_nf = nf;
    for (int f = 0; f < nf; f++) h->refresh_fields[f] = fields[f];
    return SPH_OK;
}

// ... and scatter what the peers sent into my ghost slots once it has arrived
int refresh_finish(sph_halo *h) {
    if (!h->refresh_pending) return SPH_OK;
    h->refresh_pending = false;
    H_HIP(hipStreamWaitEvent(h->s0, h->e10, 0));
    for (int q = 0; q < h->P; q++)
        if (h->ghost_count[q] > 0)
            H_TRY(sph_scatter_fields_dev(h->c, h->refresh_nf, h->refresh_fields, h->ghost_first[q], h->ghost_count[q], h->recvb[q].as<double>()));
    return SPH_OK;
}

// sink accelerations summed over ranks; a pending dt candidate min-reduced and the dt rule applied; the same message
// carries every rank's predicted box after the coming kick + drift (dist.py _reduce)
// In two halves, so that the caller can put work between them that does not need the sums: reduce_start packs and sends the
// partials off (all-gather on s1), reduce_finish lets s0 wait for them and applies them.
int reduce_start(sph_halo *h, bool before_drift) {
    H_HIP(h->part.need(SPH_PARTIALS * 8));
    // the predicted box costs a pass over the particles: only where a drift follows and other ranks read it
    H_TRY(sph_pack_partials_ex_dev(h->c, h->part.as<double>(), (before_drift && h->P > 1 && !h->octree) ? 1 : 0));
    if (h->P > 1) {
        H_HIP(h->allpart.need((size_t)h->P * SPH_PARTIALS * 8));
        if (int st = s0_then_s1(h)) return st;
        H_TR(h->tr->allgather(h->part.p, h->allpart.p, SPH_PARTIALS * 8, h->s1));
        h->st.collectives++;
        // to the host without stalling either stream: read after the drift
        if (before_drift) H_HIP(hipMemcpyAsync(h->pin_part, h->allpart.p, (size_t)h->P * SPH_PARTIALS * 8, hipMemcpyDeviceToHost, h->s1));
        H_HIP(hipEventRecord(h->e_pred, h->s1));
    }
    return SPH_OK;
}

int reduce_finish(sph_halo *h, bool before_drift) {
    if (h->P == 1) {
        H_TRY(sph_apply_partials_dev(h->c, h->part.as<double>(), 1, SPH_PARTIALS, h->dt_pending ? 1 : 0));
        h->pred_valid = false;
    } else {
        H_HIP(hipStreamWaitEvent(h->s0, h->e_pred, 0));
        H_TRY(sph_apply_partials_dev(h->c, h->allpart.as<double>(), h->P, SPH_PARTIALS, h->dt_pending ? 1 : 0));
        h->pred_valid = before_drift && !h->octree;
    }
    h->dt_pending = false;
    return SPH_OK;
}

// self-gravity sources as a locally essential tree: own particles + what every other rank cut for this rank's box
int gravity_sources_let(sph_halo *h) {
    static const int32_t XYZM[4] = {SPH_F_X, SPH_F_Y, SPH_F_Z, SPH_F_M};
    const int P = h->P;
    const int64_t n = h->n_owned;
    // the global root box: every rank's exact box of this exchange
    auto finite6 = [&](int q) { for (int a = 0; a < 6; a++) if (!std::isfinite(h->last_boxes[(size_t)q * 6 + a])) return false; return true; };
    double lo_hi[6] = {INFINITY, INFINITY, INFINITY, -INFINITY, -INFINITY, -INFINITY};
    LetBoxes lb{};
    int peer_of[MAXP];
    lb.n = 0;
    for (int q = 0; q < P; q++) {
        if (!finite6(q)) continue;
        for (int a = 0; a < 3; a++) {
            lo_hi[a] = std::min(lo_hi[a], h->last_boxes[(size_t)q * 6 + a]);
            lo_hi[3 + a] = std::max(lo_hi[3 + a], h->last_boxes[(size_t)q * 6 + 3 + a]);
        }
        if (q == h->rank) continue;
        for (int a = 0; a < 6; a++) lb.b[lb.n][a] = h->last_boxes[(size_t)q * 6 + a];
        peer_of[lb.n++] = q;
    }
    LetRoot rb;
    rb.size = 0.0;
    for (int a = 0; a < 3; a++) { rb.c[a] = (lo_hi[3 + a] + lo_hi[a]) / 2.0; rb.size = std::max(rb.size, lo_hi[3 + a] - lo_hi[a]); }
    const int nb = lb.n;
    int64_t sendc[MAXP];
    for (int q = 0; q < P; q++) sendc[q] = 0;
    H_HIP(h->srcmine.need((size_t)std::max<int64_t>(n, 1) * 4 * 8));
    if (n > 0) H_TRY(sph_gather_fields_dev(h->c, 4, XYZM, n, nullptr, h->srcmine.as<double>()));       // [4][n]
    if (n > 0 && nb > 0) {
        if ((int64_t)nb * n >= 2147483647LL) { h->err = "sph_halo: LET cut larger than 2^31 items"; return SPH_ERR_ARG; }
        const size_t items = (size_t)nb * (size_t)n;
        H_HIP(h->lkeys.need((size_t)n * 8)); H_HIP(h->lkeys2.need((size_t)n * 8));
        H_HIP(h->lvals.need((size_t)n * 4)); H_HIP(h->lvals2.need((size_t)n * 4));
        H_HIP(h->lhead.need(items * 4)); H_HIP(h->lseg.need(items * 4));
        H_HIP(h->lmom.need(items * 32)); H_HIP(h->lsum.need(items * 32)); H_HIP(h->lukeys.need(items * 4));
        H_HIP(h->lcount.need(64));
        let_keys<<<blocks_for(n), 256, 0, h->s0>>>(rb, h->srcmine.as<double>(), n, h->lkeys.as<uint64_t>(), h->lvals.as<uint32_t>());
        H_HIP(hipGetLastError());
        size_t t1 = 0, t2 = 0, t3 = 0;
        H_HIP(rocprim::radix_sort_pairs(nullptr, t1, h->lkeys.as<uint64_t>(), h->lkeys2.as<uint64_t>(), h->lvals.as<uint32_t>(), h->lvals2.as<uint32_t>(),
                                        (size_t)n, 0u, 63u, h->s0));
        H_HIP(rocprim::inclusive_scan(nullptr, t2, h->lhead.as<int32_t>(), h->lseg.as<int32_t>(), items, rocprim::plus<int32_t>(), h->s0));
        H_HIP(rocprim::reduce_by_key(nullptr, t3, h->lseg.as<int32_t>(), h->lmom.as<double4>(), items, h->lukeys.as<int32_t>(), h->lsum.as<double4>(),
                                     h->lcount.as<unsigned int>(), Add4(), rocprim::equal_to<int32_t>(), h->s0));
        H_HIP(h->lsort.need(std::max(t1, std::max(t2, t3))));
        H_HIP(rocprim::radix_sort_pairs(h->lsort.p, t1, h->lkeys.as<uint64_t>(), h->lkeys2.as<uint64_t>(), h->lvals.as<uint32_t>(), h->lvals2.as<uint32_t>(),
                                        (size_t)n, 0u, 63u, h->s0));
        let_cut<<<dim3(blocks_for(n).x, (unsigned)nb), 256, 0, h->s0>>>(rb, lb, 1.0 / (h->theta * h->theta), h->lkeys2.as<uint64_t>(), h->lvals2.as<uint32_t>(),
                                                                         h->srcmine.as<double>(), n, h->lhead.as<int32_t>(), h->lmom.as<double4>());
        H_HIP(hipGetLastError());
        H_HIP(rocprim::inclusive_scan(h->lsort.p, t2, h->lhead.as<int32_t>(), h->lseg.as<int32_t>(), items, rocprim::plus<int32_t>(), h->s0));
        H_HIP(rocprim::reduce_by_key(h->lsort.p, t3, h->lseg.as<int32_t>(), h->lmom.as<double4>(), items, h->lukeys.as<int32_t>(), h->lsum.as<double4>(),
                                     h->lcount.as<unsigned int>(), Add4(), rocprim::equal_to<int32_t>(), h->s0));
        // items per receiver: the running item number at the end of each receiver's block (one read-back)
        for (int b = 0; b < nb; b++)
            H_HIP(hipMemcpyAsync(h->pin_row + b, h->lseg.as<int32_t>() + ((size_t)(b + 1) * n - 1), 4, hipMemcpyDeviceToHost, h->s0));
        if (int st = host_wait(h, h->s0)) return st;
        int64_t before = 0;
        H_HIP(h->lrec.need((size_t)std::max<int64_t>((int64_t)reinterpret_cast<int32_t *>(h->pin_row + nb - 1)[0], 1) * 32));
        const int64_t total = reinterpret_cast<int32_t *>(h->pin_row + nb - 1)[0];
        let_records<<<blocks_for(total), 256, 0, h->s0>>>(h->lsum.as<double4>(), total, h->lrec.as<double4>());
        H_HIP(hipGetLastError());
        for (int b = 0; b < nb; b++) {
            const int64_t upto = reinterpret_cast<int32_t *>(h->pin_row + b)[0];
            sendc[peer_of[b]] = upto - before;
            before = upto;
        }
    }
    // counts: who sends how much to whom (P x P on the host), then one grouped round of {x, y, z, m} records
    std::vector<int64_t> cm((size_t)P * P);
    if (int st = gather_counts(h, sendc, cm.data())) return st;
    const void *sp[MAXP];
    void *rp[MAXP];
    size_t sb[MAXP], rbytes[MAXP];
    int64_t recv_total = 0, off_send = 0;
    for (int q = 0; q < P; q++) { sp[q] = nullptr; rp[q] = nullptr; sb[q] = 0; rbytes[q] = 0; }
    for (int q = 0; q < P; q++) if (q != h->rank) recv_total += cm[(size_t)q * P + h->rank];
    H_HIP(h->src.need((size_t)std::max<int64_t>(n + recv_total, 1) * 32));
    if (n > 0) { let_own_records<<<blocks_for(n), 256, 0, h->s0>>>(h->srcmine.as<double>(), n, h->src.as<double4>()); H_HIP(hipGetLastError()); }
    int64_t off_recv = n;
    for (int q = 0; q < P; q++) {
        if (q == h->rank) continue;
        if (sendc[q] > 0) { sp[q] = h->lrec.as<double4>() + off_send; sb[q] = (size_t)sendc[q] * 32; off_send += sendc[q]; }
        const int64_t rq = cm[(size_t)q * P + h->rank];
        if (rq > 0) { rp[q] = h->src.as<double4>() + off_recv; rbytes[q] = (size_t)rq * 32; off_recv += rq; }
    }
    if (int st = s0_then_s1(h)) return st;
    if (int st = p2p_raw(h, sp, sb, rp, rbytes)) return st;
    if (int st = s1_then_s0(h)) return st;
    h->st.let_sent += off_send; h->st.let_received += recv_total; h->st.let_updates++;
    H_TRY(sph_set_gravity_sources_dev(h->c, n + recv_total, h->src.as<double>(), lo_hi));
    return SPH_OK;
}

// one force evaluation: create_tree .. find_forces of the reference, [F]:894-898 (dist.py evaluate)
int evaluate(sph_halo *h, bool before_drift) {
    static const int32_t RHO[2] = {SPH_F_RHO, SPH_F_OMEGA};
    static const int32_t VEL[5] = {SPH_F_VX, SPH_F_VY, SPH_F_VZ, SPH_F_U, SPH_F_ALPHA};
    const bool multi = h->P > 1;
    if (h->pos_dirty) {
        if (multi && h->migrate_every > 0 && h->edges.n == h->P - 1 && h->since_migrate >= h->migrate_every) {
            if (int st = migrate(h)) return st;
            h->since_migrate = 0;
            h->pred_for_drift = false;          // ownership changed: the predicted boxes are void
        }
        if (multi) if (int st = exchange_ghosts(h)) return st;
        if (multi && h->octree && !h->sources_valid) {
            if (int st = h->let ? gravity_sources_let(h) : gravity_sources(h)) return st;
            h->sources_valid = true;
        }
        if (!multi && h->variable && h->n_owned > 0) H_TRY(sph_set_numbers_dev(h->c, 0, h->n_owned, h->gid.as<int64_t>()));
        H_TRY(sph_density(h->c));
        if (multi) if (int st = refresh_start(h, h->variable ? 2 : 1, RHO)) return st;      // variable h: rho and Omega
    } else {
        // the density sum needs positions and masses only: it runs while the ghosts' v, u, alpha travel
        if (multi && h->vel_dirty) if (int st = refresh_start(h, 5, VEL)) return st;
        H_TRY(sph_density(h->c));
    }
    h->pos_dirty = h->vel_dirty = false;
    if (multi && !h->octree) {
        // ... and so do the forces of the wavefronts that cannot see a ghost
        H_TRY(sph_forces_part(h->c, 1));
        if (int st = refresh_finish(h)) return st;
        H_TRY(sph_refresh_eos_ghosts(h->c));
        H_TRY(sph_forces_part(h->c, 2));
    } else if (multi) {
        // the octree paths evaluate in one piece (the tree walk and the grad-h pass have no interior / boundary split)
        if (int st = refresh_finish(h)) return st;
        H_TRY(sph_refresh_eos_ghosts(h->c));
        H_TRY(sph_forces(h->c));
    } else {
        H_TRY(sph_forces(h->c));
    }
    return reduce_start(h, before_drift);
}

// survivors of an accretion / cull keep their global numbers (the reference's pack(), [F]:481,554, applied to this object's list)
int compact_gid(sph_halo *h, int64_t n_before, int64_t n_after) {
    if (n_after == n_before) return SPH_OK;
    H_HIP(h->gid_new.need((size_t)std::max<int64_t>(n_after, 1) * 8));
    H_HIP(h->sel_count.need(8));
    size_t tmp_bytes = 0;
    H_HIP(rocprim::select(nullptr, tmp_bytes, h->gid.as<int64_t>(), h->keep.as<int32_t>(), h->gid_new.as<int64_t>(), h->sel_count.as<size_t>(),
                          (size_t)n_before, h->s0));
    H_HIP(h->sel_tmp.need(tmp_bytes));
    H_HIP(rocprim::select(h->sel_tmp.p, tmp_bytes, h->gid.as<int64_t>(), h->keep.as<int32_t>(), h->gid_new.as<int64_t>(), h->sel_count.as<size_t>(),
                          (size_t)n_before, h->s0));
    std::swap(h->gid, h->gid_new);
    return SPH_OK;
}

// initiate_sink_accretion + check_bounds, [F]:919-920 ([V]:1156-1158).  Several ranks: every rank decides for its own particles
// on the shared octree; the per-sink sums over the accreted particles are all-gathered and added in rank order, so every rank
// updates the (replicated) sinks identically (dist.py _accrete_and_cull).  One rank: the context's own pass.
int accrete_and_cull(sph_halo *h) {
    const int P = h->P;
    const int64_t n_before = h->n_owned;
    int64_t removed = 0;
    H_HIP(h->keep.need((size_t)std::max<int64_t>(n_before, 1) * 4));
    if (P == 1) {
        H_TRY(sph_accrete_and_cull_keep(h->c, h->keep.as<int32_t>(), &removed));
        if (removed > 0) h->pos_dirty = true;          // variable h: the survivors' numbers are set again before the next pass
    } else {
        int64_t off = 0;                            // where this rank's particles sit in the source set (LET: they lead it)
        if (!h->let) for (int q = 0; q < h->rank; q++) off += h->counts_all[q];
        H_HIP(h->accp.need(ACC_PARTIALS * 8));
        H_HIP(h->accall.need((size_t)P * ACC_PARTIALS * 8));
        H_TRY(sph_accrete_mark_dev(h->c, off, h->accp.as<double>()));
        if (int st = s0_then_s1(h)) return st;
        H_TR(h->tr->allgather(h->accp.p, h->accall.p, ACC_PARTIALS * 8, h->s1));
        h->st.collectives++;
        if (int st = s1_then_s0(h)) return st;
        H_TRY(sph_accrete_apply_dev(h->c, h->accall.as<double>(), P, ACC_PARTIALS, h->keep.as<int32_t>(), &removed));
        // the ghosts were dropped with the accreted particles: exchange before the next pass
        h->pos_dirty = true; h->vel_dirty = false;
        h->counts_valid = false; h->sources_valid = false;
        for (int q = 0; q < P; q++) { h->send_count[q] = 0; h->ghost_count[q] = 0; }
    }
    if (removed > 0) {
        if (int st = compact_gid(h, n_before, n_before - removed)) return st;
        h->n_owned = n_before - removed;
        h->pred_for_drift = false; h->pred_valid = false;
    }
    h->st.removed += removed;
    return SPH_OK;
}

// check_sink_creation, [V]:549-597: the first particle by global number that qualifies, found with one all-gather; every rank
// adds the same sink (the sinks are replicated)
int create_sink(sph_halo *h) {
    int32_t created = 0;
    if (h->P == 1) {
        H_TRY(sph_check_sink_creation(h->c, &created));
    } else {
        H_HIP(h->cand.need(9 * 8));
        H_HIP(h->candall.need((size_t)h->P * 9 * 8));
        H_TRY(sph_sink_candidate_dev(h->c, h->cand.as<double>()));
        if (int st = s0_then_s1(h)) return st;
        H_TR(h->tr->allgather(h->cand.p, h->candall.p, 9 * 8, h->s1));
        h->st.collectives++;
        if (int st = s1_then_s0(h)) return st;
        pick_candidate<<<1, 64, 0, h->s0>>>(h->candall.as<double>(), h->P, h->cand.as<double>());
        H_HIP(hipGetLastError());
        H_TRY(sph_add_sink_checked_dev(h->c, h->cand.as<double>(), &created));
    }
    h->st.sinks_created += created;
    return SPH_OK;
}

int step(sph_halo *h) {
    if (int st = evaluate(h, true)) return st;
    if (int st = reduce_finish(h, true)) return st;       // the kick needs the new dt: nothing to put in between
    H_TRY(sph_kick_drift_devdt(h->c));
    h->pos_dirty = true;
    h->sources_valid = false;
    h->pred_for_drift = true;                   // the reduction above predicted where this drift takes everybody
    h->since_migrate++;
    if (int st = evaluate(h, false)) return st;           // a kick follows, no drift
    // closing kick + get_next_timestep's local part ([F]:845-851; reduced with the next evaluation) for the gas while the sinks'
    // accelerations travel; the sinks are kicked when they have arrived
    H_TRY(sph_kick_dt_candidate_gas_dev(h->c));
    if (int st = reduce_finish(h, false)) return st;
    H_TRY(sph_kick_sinks_devdt(h->c));
    h->vel_dirty = true;
    h->dt_pending = true;
    if (h->variable) {
        H_TRY(sph_update_h(h->c));              // calc_smoothing, [V]:1152; the ghosts' h is stale now: full exchange next
        if (h->P > 1) { h->pos_dirty = true; h->vel_dirty = false; }
    }
    if (h->sink_creation) if (int st = create_sink(h)) return st;                 // [V]:1155
    if (h->accrete) if (int st = accrete_and_cull(h)) return st;                  // [F]:919-920
    return SPH_OK;
}

// reduce a pending dt candidate now (end of a run): t += dt and [F]:855-858 on every rank
int finish_dt(sph_halo *h) {
    if (!h->dt_pending) return SPH_OK;
    H_HIP(h->part.need(SPH_PARTIALS * 8));
    H_HIP(h->row.need(SPH_PARTIALS * 8));
    H_TRY(sph_pack_partials_ex_dev(h->c, h->part.as<double>(), 0));
    if (h->P == 1) {
        H_TRY(sph_apply_partials_dev(h->c, h->part.as<double>(), 1, SPH_PARTIALS, 1));
    } else {
        H_HIP(h->allpart.need((size_t)h->P * SPH_PARTIALS * 8));
        if (int st = s0_then_s1(h)) return st;
        H_TR(h->tr->allgather(h->part.p, h->allpart.p, SPH_PARTIALS * 8, h->s1));
        h->st.collectives++;
        if (int st = s1_then_s0(h)) return st;
        // the sink accelerations in the blocks are the totals every rank already holds: keep them, apply only the dt part
        min_dt_row<<<1, 256, 0, h->s0>>>(h->allpart.as<double>(), h->P, h->rank, h->row.as<double>());
        H_HIP(hipGetLastError());
        H_TRY(sph_apply_partials_dev(h->c, h->row.as<double>(), 1, SPH_PARTIALS, 1));
    }
    h->dt_pending = false;
    h->pred_valid = false;
    return SPH_OK;
}

void destroy_impl(sph_halo *h) {
    DevGuard g(h->device);
    if (h->s0) (void)hipStreamSynchronize(h->s0);
    if (h->s1) (void)hipStreamSynchronize(h->s1);
    if (h->c && h->s0) (void)sph_set_stream(h->c, nullptr);     // the context outlives this object: off our stream before it goes
    delete h->tr;
    for (DevBuf *b : {&h->gid, &h->gid_new, &h->own, &h->newbuf, &h->part, &h->allpart, &h->row, &h->box, &h->boxes, &h->cnt, &h->cntall,
                      &h->ghosts, &h->dest, &h->flags, &h->keep_ids, &h->sel_tmp, &h->sel_count, &h->src, &h->srcmine, &h->srcall, &h->accp,
                      &h->accall, &h->keep, &h->cand, &h->candall, &h->gnum, &h->lkeys, &h->lkeys2, &h->lvals, &h->lvals2, &h->lsort, &h->lhead,
                      &h->lseg, &h->lmom, &h->lsum, &h->lukeys, &h->lcount, &h->lscan, &h->lrec}) b->release();
    for (int q = 0; q < MAXP; q++) { h->sendb[q].release(); h->recvb[q].release(); h->ids[q].release(); }
    if (h->pin) (void)hipHostFree(h->pin);
    for (hipEvent_t e : {h->e01, h->e10, h->e_pred}) if (e) (void)hipEventDestroy(e);
    if (h->s0) (void)hipStreamDestroy(h->s0);
    if (h->s1 && h->own_s1) (void)hipStreamDestroy(h->s1);
    delete h;
}

thread_local std::string g_create_err;

}  // namespace

extern "C" {

int sph_halo_unique_id(void *id128) {
    if (!id128) return SPH_ERR_ARG;
    static_assert(sizeof(ncclUniqueId) == SPH_HALO_ID_BYTES, "ncclUniqueId is 128 bytes");
    ncclUniqueId id;
    if (ncclGetUniqueId(&id) != ncclSuccess) return SPH_ERR_HIP;
    std::memcpy(id128, &id, sizeof id);
    return SPH_OK;
}

static int create_common(sph_halo *h, sph_halo **out) {
    const int st = common_init(h);
    if (st != SPH_OK) { g_create_err = h->err; destroy_impl(h); return st; }
    *out = h;
    return SPH_OK;
}

int sph_halo_create(sph_ctx *ctx, const void *id128, int32_t rank, int32_t nranks, sph_halo **out) {
    if (!ctx || !id128 || !out || nranks < 1 || nranks > MAXP || rank < 0 || rank >= nranks) return SPH_ERR_ARG;
    int dev = 0;
    if (hipStreamGetDevice(reinterpret_cast<hipStream_t>(sph_stream(ctx)), &dev) != hipSuccess) (void)hipGetDevice(&dev);
    DevGuard g(dev);
    ncclUniqueId id;
    std::memcpy(&id, id128, sizeof id);
    auto *tr = new RcclTransport();
    tr->P = nranks;
    tr->own = true;
    const ncclResult_t r = ncclCommInitRank(&tr->comm, nranks, id, rank);
    if (r != ncclSuccess) { g_create_err = std::string("ncclCommInitRank: ") + ncclGetErrorString(r); tr->comm = nullptr; delete tr; return SPH_ERR_HIP; }
    auto *h = new sph_halo();
    h->c = ctx; h->rank = rank; h->P = nranks; h->tr = tr;
    return create_common(h, out);
}

int sph_halo_attach(sph_ctx *ctx, void *nccl_comm, void *comm_stream, int32_t rank, int32_t nranks, sph_halo **out) {
    if (!ctx || !nccl_comm || !out || nranks < 1 || nranks > MAXP || rank < 0 || rank >= nranks) return SPH_ERR_ARG;
    auto *tr = new RcclTransport();
    tr->P = nranks;
    tr->comm = reinterpret_cast<ncclComm_t>(nccl_comm);
    tr->own = false;
    auto *h = new sph_halo();
    h->c = ctx; h->rank = rank; h->P = nranks; h->tr = tr;
    if (comm_stream) { h->s1 = reinterpret_cast<hipStream_t>(comm_stream); h->own_s1 = false; }
    return create_common(h, out);
}

void *sph_halo_hub_create(int32_t nranks) {
    if (nranks < 1 || nranks > MAXP) return nullptr;
    return new Hub(nranks);
}

void sph_halo_hub_destroy(void *hub) {
    auto *hb = static_cast<Hub *>(hub);
    if (!hb) return;
    for (int q = 0; q < MAXP; q++) { if (hb->ready[q]) (void)hipEventDestroy(hb->ready[q]); if (hb->done[q]) (void)hipEventDestroy(hb->done[q]); }
    delete hb;
}

int sph_halo_create_inproc(sph_ctx *ctx, void *hub, int32_t rank, int32_t nranks, sph_halo **out) {
    auto *hb = static_cast<Hub *>(hub);
    if (!ctx || !hb || !out || nranks != hb->P || rank < 0 || rank >= nranks) return SPH_ERR_ARG;
    int dev = 0;
    if (hipStreamGetDevice(reinterpret_cast<hipStream_t>(sph_stream(ctx)), &dev) != hipSuccess) (void)hipGetDevice(&dev);
    DevGuard g(dev);
    if (hipEventCreateWithFlags(&hb->ready[rank], hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&hb->done[rank], hipEventDisableTiming) != hipSuccess) return SPH_ERR_HIP;
    auto *tr = new InprocTransport();
    tr->hub = hb;
    tr->rank = rank;
    auto *h = new sph_halo();
    h->c = ctx; h->rank = rank; h->P = nranks; h->tr = tr;
    return create_common(h, out);
}

int sph_halo_destroy(sph_halo *h) {
    if (!h) return SPH_ERR_ARG;
    destroy_impl(h);
    return SPH_OK;
}

const char *sph_halo_last_error(const sph_halo *h) { return h ? h->err.c_str() : g_create_err.c_str(); }

int sph_halo_set_slabs(sph_halo *h, const double *edges, int32_t migrate_every) {
    if (!h || migrate_every < 0 || (h->P > 1 && !edges)) return SPH_ERR_ARG;
    for (int k = 0; k + 1 < h->P; k++) {
        if (k > 0 && !(edges[k] >= edges[k - 1])) { h->err = "sph_halo_set_slabs: edges must ascend"; return SPH_ERR_ARG; }
        h->edges.e[k] = edges[k];
    }
    h->edges.n = h->P - 1;
    h->migrate_every = migrate_every;
    return SPH_OK;
}

static int upload_impl(sph_halo *h, int64_t n, const double *const st9[9], const double *hsml, const int64_t *gid) {
    if (!h || n < 0) return SPH_ERR_ARG;
    if (h->variable && n > 0 && !hsml) { h->err = "sph_halo_upload: a variable-h context needs the smoothing lengths (sph_halo_upload_v)"; return SPH_ERR_ARG; }
    DevGuard g(h->device);
    if (int st = ensure_reserve(h, n)) return st;
    H_TRY(sph_upload(h->c, n, st9[0], st9[1], st9[2], st9[3], st9[4], st9[5], st9[6], st9[7], st9[8]));
    if (h->variable && n > 0) H_TRY(sph_upload_field(h->c, SPH_F_H, hsml, n));
    H_HIP(h->gid.need((size_t)std::max<int64_t>(n, 1) * 8));
    if (n > 0) {
        if (gid) H_HIP(hipMemcpyAsync(h->gid.p, gid, (size_t)n * 8, hipMemcpyHostToDevice, h->s0));
        else { iota64<<<blocks_for(n), 256, 0, h->s0>>>(h->gid.as<int64_t>(), n); H_HIP(hipGetLastError()); }
        H_HIP(hipStreamSynchronize(h->s0));
    }
    h->n_owned = n;
    h->uploaded = true;
    h->pos_dirty = true; h->vel_dirty = false; h->dt_pending = false; h->pred_for_drift = false; h->pred_valid = false;
    h->counts_valid = false; h->sources_valid = false;
    h->since_migrate = 0;
    for (int q = 0; q < MAXP; q++) { h->send_count[q] = 0; h->ghost_count[q] = 0; h->cap_send[q] = 0; h->cap_recv[q] = 0; }
    return SPH_OK;
}

int sph_halo_upload(sph_halo *h, int64_t n, const double *x, const double *y, const double *z, const double *vx, const double *vy,
                    const double *vz, const double *u, const double *m, const double *alpha, const int64_t *gid) {
    const double *st9[9] = {x, y, z, vx, vy, vz, u, m, alpha};
    return upload_impl(h, n, st9, nullptr, gid);
}

int sph_halo_upload_v(sph_halo *h, int64_t n, const double *x, const double *y, const double *z, const double *vx, const double *vy,
                      const double *vz, const double *u, const double *m, const double *alpha, const double *hsml, const int64_t *gid) {
    const double *st9[9] = {x, y, z, vx, vy, vz, u, m, alpha};
    return upload_impl(h, n, st9, hsml, gid);
}

int sph_halo_run(sph_halo *h, int32_t nsteps, double *dt, double *t) {
    if (!h || !dt || !t || nsteps < 0) return SPH_ERR_ARG;
    if (!h->uploaded) { h->err = "sph_halo_run: call sph_halo_upload first"; return SPH_ERR_STATE; }
    DevGuard g(h->device);
    H_TRY(sph_set_dt(h->c, *dt, *t));
    for (int s = 0; s < nsteps; s++) if (int st = step(h)) return st;
    if (int st = finish_dt(h)) return st;
    H_TRY(sph_get_dt(h->c, dt, t));
    return SPH_OK;
}

int64_t sph_halo_count(const sph_halo *h) { return h ? h->n_owned : -1; }

static int download_impl(sph_halo *h, int64_t capacity, double *const dst[NF_MAX], int64_t *gid) {
    if (!h || capacity < h->n_owned) return SPH_ERR_ARG;
    DevGuard g(h->device);
    const int NF = h->nf;
    const int64_t n = h->n_owned;
    if (n == 0) return SPH_OK;
    H_HIP(h->own.need((size_t)n * NF * 8));
    H_TRY(sph_gather_fields_dev(h->c, NF, STATE, n, nullptr, h->own.as<double>()));
    for (int f = 0; f < NF; f++)
        if (dst[f]) H_HIP(hipMemcpyAsync(dst[f], h->own.as<double>() + (size_t)f * n, (size_t)n * 8, hipMemcpyDeviceToHost, h->s0));
    if (gid) H_HIP(hipMemcpyAsync(gid, h->gid.p, (size_t)n * 8, hipMemcpyDeviceToHost, h->s0));
    H_HIP(hipStreamSynchronize(h->s0));
    return SPH_OK;
}

int sph_halo_download(sph_halo *h, int64_t capacity, double *x, double *y, double *z, double *vx, double *vy, double *vz, double *u,
                      double *m, double *alpha, int64_t *gid) {
    double *dst[NF_MAX] = {x, y, z, vx, vy, vz, u, m, alpha, nullptr};
    return download_impl(h, capacity, dst, gid);
}

int sph_halo_download_v(sph_halo *h, int64_t capacity, double *x, double *y, double *z, double *vx, double *vy, double *vz, double *u,
                        double *m, double *alpha, double *hsml, int64_t *gid) {
    if (h && !h->variable && hsml) { h->err = "sph_halo_download_v: the context has no per-particle h"; return SPH_ERR_ARG; }
    double *dst[NF_MAX] = {x, y, z, vx, vy, vz, u, m, alpha, hsml};
    return download_impl(h, capacity, dst, gid);
}

static int gather_root_impl(sph_halo *h, int32_t root, int64_t capacity, int64_t *n_total, double *const dst[NF_MAX], int64_t *gid) {
    if (!h || root < 0 || root >= h->P || !n_total) return SPH_ERR_ARG;
    DevGuard g(h->device);
    const int P = h->P, NF = h->nf;
    const int64_t n = h->n_owned;
    int64_t mine[MAXP];
    for (int q = 0; q < P; q++) mine[q] = q == root ? n : 0;       // row r of the matrix: what rank r sends to each rank
    std::vector<int64_t> cm((size_t)P * P);
    if (int st = gather_counts(h, mine, cm.data())) return st;
    int64_t total = 0;
    for (int q = 0; q < P; q++) total += cm[(size_t)q * P + root];
    *n_total = total;
    const bool is_root = h->rank == root;
    const bool fits = !is_root || capacity >= total;
    // payload: the state rows + the global number
    H_HIP(h->own.need((size_t)std::max<int64_t>(n, 1) * (NF + 1) * 8));
    if (n > 0) {
        H_TRY(sph_gather_fields_dev(h->c, NF, STATE, n, nullptr, h->own.as<double>()));
        gid_to_double<<<blocks_for(n), 256, 0, h->s0>>>(h->gid.as<int64_t>(), nullptr, n, h->own.as<double>() + (size_t)NF * n);
        H_HIP(hipGetLastError());
    }
    int64_t send_n[MAXP], recv_n[MAXP];
    for (int q = 0; q < P; q++) { send_n[q] = 0; recv_n[q] = 0; }
    if (!is_root) {
        send_n[root] = n;
        // p2p sends out of sendb: let it alias the payload
        std::swap(h->sendb[root], h->own);
    } else {
        for (int q = 0; q < P; q++) if (q != root) recv_n[q] = cm[(size_t)q * P + root];
    }
    int st = s0_then_s1(h);
    if (st == SPH_OK) st = p2p(h, send_n, recv_n, NF + 1);
    if (st == SPH_OK) st = host_wait(h, h->s1);
    if (!is_root) std::swap(h->sendb[root], h->own);
    if (st != SPH_OK) return st;
    if (!is_root) return SPH_OK;
    if (!fits) { h->err = "sph_halo_gather_root: capacity below the total particle count"; return SPH_ERR_ARG; }
    // to the host block by block, then into global-number order
    std::vector<double> blk;
    std::vector<int64_t> order((size_t)total);
    std::vector<double> all((size_t)total * (NF + 1));
    int64_t off = 0;
    for (int q = 0; q < P; q++) {
        const int64_t nq = cm[(size_t)q * P + root];
        if (nq == 0) continue;
        const double *src = q == root ? h->own.as<double>() : h->recvb[q].as<double>();
        blk.resize((size_t)nq * (NF + 1));
        H_HIP(hipMemcpy(blk.data(), src, blk.size() * 8, hipMemcpyDeviceToHost));
        for (int r = 0; r <= NF; r++) std::memcpy(&all[(size_t)r * total + off], &blk[(size_t)r * nq], (size_t)nq * 8);
        off += nq;
    }
    std::iota(order.begin(), order.end(), (int64_t)0);
    const double *gd = &all[(size_t)NF * total];
    std::stable_sort(order.begin(), order.end(), [&](int64_t a, int64_t b) { return gd[a] < gd[b]; });
    for (int f = 0; f < NF; f++)
        if (dst[f]) for (int64_t k = 0; k < total; k++) dst[f][k] = all[(size_t)f * total + order[(size_t)k]];
    if (gid) for (int64_t k = 0; k < total; k++) gid[k] = (int64_t)gd[order[(size_t)k]];
    return SPH_OK;
}

int sph_halo_gather_root(sph_halo *h, int32_t root, int64_t capacity, int64_t *n_total, double *x, double *y, double *z, double *vx,
                         double *vy, double *vz, double *u, double *m, double *alpha, int64_t *gid) {
    double *dst[NF_MAX] = {x, y, z, vx, vy, vz, u, m, alpha, nullptr};
    return gather_root_impl(h, root, capacity, n_total, dst, gid);
}

int sph_halo_gather_root_v(sph_halo *h, int32_t root, int64_t capacity, int64_t *n_total, double *x, double *y, double *z, double *vx,
                           double *vy, double *vz, double *u, double *m, double *alpha, double *hsml, int64_t *gid) {
    double *dst[NF_MAX] = {x, y, z, vx, vy, vz, u, m, alpha, hsml};
    return gather_root_impl(h, root, capacity, n_total, dst, gid);
}

int sph_halo_get_stats(const sph_halo *h, sph_halo_stats *out) {
    if (!h || !out) return SPH_ERR_ARG;
    *out = h->st;
    return SPH_OK;
}

int sph_halo_selftest(sph_halo *h, int64_t count) {
    if (!h || count < 1) return SPH_ERR_ARG;
    DevGuard g(h->device);
    const int P = h->P;
    auto value = [](int from, int to) { return 1.0e6 * from + 1.0e3 * to; };
    int64_t cn[MAXP];
    for (int q = 0; q < P; q++) {
        cn[q] = count;
        H_HIP(h->sendb[q].need((size_t)count * 8));
        pattern_kernel<<<blocks_for(count), 256, 0, h->s0>>>(h->sendb[q].as<double>(), count, value(h->rank, q));
        H_HIP(hipGetLastError());
    }
    if (int st = s0_then_s1(h)) return st;
    if (int st = p2p(h, cn, cn, 1)) return st;
    if (int st = host_wait(h, h->s1)) return st;
    std::vector<double> got((size_t)count);
    for (int q = 0; q < P; q++) {
        H_HIP(hipMemcpy(got.data(), h->recvb[q].p, (size_t)count * 8, hipMemcpyDeviceToHost));
        for (int64_t k = 0; k < count; k++)
            if (got[(size_t)k] != value(q, h->rank) + (double)k) { h->err = "sph_halo_selftest: point-to-point payload differs"; return SPH_ERR_STATE; }
    }
    H_HIP(h->own.need((size_t)count * P * 8));
    if (int st = s0_then_s1(h)) return st;
    H_TR(h->tr->allgather(h->sendb[0].p, h->own.p, (size_t)count * 8, h->s1));
    h->st.collectives++;
    if (int st = host_wait(h, h->s1)) return st;
    for (int q = 0; q < P; q++) {
        H_HIP(hipMemcpy(got.data(), h->own.as<double>() + (size_t)q * count, (size_t)count * 8, hipMemcpyDeviceToHost));
        for (int64_t k = 0; k < count; k++)
            if (got[(size_t)k] != value(q, 0) + (double)k) { h->err = "sph_halo_selftest: all-gather payload differs"; return SPH_ERR_STATE; }
    }
    return SPH_OK;
}

}  // extern "C"
