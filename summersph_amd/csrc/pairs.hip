// pairs.hip -- the hot path: neighbour list, density + EOS, SPH pair forces.
//
// Replaces (citations: /root/reference/SUMMER_SPH.f90, "[F]")
//   density_tree_search / get_density        [F]:398-457   -> density_kernel
//   get_pressure_and_sound_speed             [F]:459-468   -> fused into density_kernel's epilogue
//   SPH_tree_search / get_SPH                [F]:295-395   -> forces_kernel (gather form)
//   zero_rates + sink_gravforces (gas side)  [F]:779-793,559-576 -> forces_kernel prologue
//   sink_gravforces (sink side)              [F]:567-591   -> sink_accel kernels
//
// Structure.  Positions change once per step (drift), but a step evaluates density and
// forces twice on each position set (end of step n and start of step n+1 see the same
// positions).  So the 27-cell candidate scan runs ONCE per position set and leaves a
// neighbour list; the four pair passes that follow run on true neighbours only, with all 64
// lanes of a wave doing pair arithmetic instead of distance tests.
//
// Neighbour list layout ("ELL, wave-strided"): particle i = (wave w, lane l); its k-th
// neighbour sits at nlist[(w*cap + k)*64 + l].  Each loop iteration of a wave therefore reads
// 64 consecutive ints (one 256-B request), and the neighbour data themselves are fetched as
// array-of-struct gather records (32 B for density, 96 B for forces) so that one neighbour is
// one or two cache lines instead of 4..11 scattered ones.  Records of spatially adjacent
// particles are adjacent in memory (cell-sorted order), which keeps the gathers in L1/L2.
//
// Arithmetic.  Every pair term is written in the reference's expression order (same
// divisions, same literal roundings, table look-up with linear interpolation, kernel tables
// staged in LDS).  Differences to the reference are summation order over neighbours and FMA
// contraction, both rounding-level (measured in tests/test_parity_gpu.py).
#include <cmath>

#include "pair_common.hpp"
#include "tile_common.hpp"

namespace sph {

namespace {

constexpr int PAIR_BLOCK = 256;

// ------------------------------------------------------------------------------------------
// neighbour list: every j != i with |x_i - x_j|^2 <= rcut2 (rcut2 a hair above (2h)^2; the
// evaluation kernels re-test q <= 2 exactly as lookup_kernel does, [F]:113)
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(PAIR_BLOCK) void nlist_kernel(GridDesc g, const double4 *__restrict__ drec,
                                                           const int32_t *__restrict__ cell_start, int64_t n,
                                                           double rcut2, int32_t cap, int32_t *__restrict__ nlist,
                                                           int32_t *__restrict__ ncount, int32_t *__restrict__ wave_max,
                                                           int32_t *__restrict__ wave_need, const int32_t *__restrict__ orig,
                                                           int32_t n_owned) {
    const int64_t i = (int64_t)xcd_chunk(blockIdx.x, gridDim.x) * PAIR_BLOCK + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const int64_t w = i >> 6;
    int cnt = 0;
    if (i < n && orig[i] >= n_owned) ncount[i] = 0;      // ghost: a neighbour only, never a target
    if (i < n && orig[i] < n_owned) {
        const double4 pi = drec[i];
        int cc[3];
        cell_coords(g, pi.x, pi.y, pi.z, cc);
        const int d0 = g.dim[g.s[0]], d1 = g.dim[g.s[1]], d2 = g.dim[g.s[2]];
        const int lo0 = max(cc[0] - 1, 0), hi0 = min(cc[0] + 1, d0 - 1);
        int32_t *mine = nlist + ((size_t)w * cap) * 64 + lane;
        for (int o2 = -1; o2 <= 1; o2++) {
            const int c2 = cc[2] + o2;
            if (c2 < 0 || c2 >= d2) continue;
            for (int o1 = -1; o1 <= 1; o1++) {
                const int c1 = cc[1] + o1;
                if (c1 < 0 || c1 >= d1) continue;
                const int64_t row = ((int64_t)c2 * d1 + c1) * d0;
                const int jb = cell_start[row + lo0], je = cell_start[row + hi0 + 1];
                // 4 candidates per trip: the loads are independent, so four gathers are in flight
                for (int j = jb; j < je; j += 4) {
                    double r2[4];
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        const double4 pj = drec[min(j + u, je - 1)];
                        const double dx = pi.x - pj.x, dy = pi.y - pj.y, dz = pi.z - pj.z;
                        r2[u] = dx * dx + dy * dy + dz * dz;
                    }
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        if (j + u < je && r2[u] <= rcut2 && j + u != (int)i) {
                            if (cnt < cap) mine[(size_t)cnt * 64] = j + u;
                            cnt++;
                        }
                    }
                }
            }
        }
        ncount[i] = cnt;
    }
    const int wm = wave_max_i32(cnt);
    if (lane == 0 && (w << 6) < n) {
        wave_max[w] = min(wm, cap);
        wave_need[w] = wm;            // reduced by max_to_host: an atomicMax per wave on one address is serialised at the memory side
    }
}

// largest of n ints -> *host_out (pinned host memory mapped into the device's address space); one workgroup
__global__ __launch_bounds__(1024) void max_to_host(const int32_t *__restrict__ v, int64_t n, int32_t *__restrict__ host_out) {
    __shared__ int s_red[16];
    int m = 0;
    for (int64_t k = threadIdx.x; k < n; k += 1024) m = max(m, v[k]);
    m = wave_max_i32(m);
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int k = 1; k < 16; k++) m = max(m, s_red[k]);
        *host_out = m;
    }
}

// ------------------------------------------------------------------------------------------
// density + EOS
// ------------------------------------------------------------------------------------------
// entry k of a lane's list -> sorted index of the neighbour.  PACKED = false: wave-strided dwords holding the index itself
// (nlist_kernel).  PACKED = true: the 16-bit tile slots of the tiled build (tile_common.hpp), turned into an index with the
// plan of the lane's group of 256.  Both are read in lockstep.
template <bool PACKED>
struct ListColumn {
    const int32_t *m32;
    const uint16_t *m16;
    EntryMap em;
    __device__ __forceinline__ ListColumn(const int32_t *nlist, const int32_t *plan_f, int64_t w, int32_t cap, int lane, int self) {
        if (PACKED) {
            m32 = nullptr;
            m16 = reinterpret_cast<const uint16_t *>(nlist) + (((size_t)w * (cap >> 3)) * 64 + lane) * 8;
            em = entry_to_index(plan_f, __builtin_amdgcn_readfirstlane(self >> 8));
        } else {
            m32 = nlist + ((size_t)w * cap) * 64 + lane;
            m16 = nullptr;
            em = EntryMap{0, 0, 0, 0, 0};
        }
    }
    __device__ __forceinline__ int operator()(int k) const {
        return PACKED ? em((int)m16[(size_t)(k >> 3) * 512 + ent_pos(k & 7)]) : m32[(size_t)k * 64];
    }
};

template <int BLOCK, bool PACKED>
__global__ __launch_bounds__(BLOCK) void density_kernel(PairConst pc, const double4 *__restrict__ drec,
                                                        const int32_t *__restrict__ nlist, int32_t cap,
                                                        const int32_t *__restrict__ ncount,
                                                        const int32_t *__restrict__ wave_max,
                                                        const double *__restrict__ w_tab, int64_t n,
                                                        const double *__restrict__ u, const double *__restrict__ alpha,
                                                        const double *__restrict__ vx, const double *__restrict__ vy,
                                                        const double *__restrict__ vz, double *__restrict__ rho,
                                                        double *__restrict__ P, double *__restrict__ cs,
                                                        double *__restrict__ frec, const int32_t *__restrict__ orig,
                                                        int32_t n_owned, const int32_t *__restrict__ plan_f) {
    extern __shared__ double lds_w[];
    for (int k = threadIdx.x; k < TAB_LEN(pc.nq); k += BLOCK) lds_w[k] = w_tab[k];
    __syncthreads();

    const int64_t i = (int64_t)xcd_chunk(blockIdx.x, gridDim.x) * BLOCK + threadIdx.x;
    if ((i & ~(int64_t)63) >= n) return;   // whole wave out of range
    const int lane = threadIdx.x & 63;
    const int64_t w = i >> 6;
    const bool live = i < n && orig[i] < n_owned;   // ghosts keep the rho their owner sent
    const int self = i < n ? (int)i : (int)(n - 1);
    const double4 pi = drec[self];
    const int cnt = live ? min(ncount[i], cap) : 0;
    const int kmax = wave_max[w];
    const ListColumn<PACKED> mine(nlist, plan_f, w, cap, lane, self);
    const double inv_h = pc.inv_h, inv_dq = pc.inv_dq;

    // Software pipeline: neighbour indices are fetched two trips ahead, records one trip ahead, so
    // the dependent index -> record gather chain overlaps the arithmetic of the current pair.
    // Lanes past their own count re-read their own record (a valid address) and are masked out.
    int j1 = 0 < cnt ? mine(0) : self;
    int j2 = 1 < cnt ? mine(1) : self;
    double4 p1 = drec[j1];
    double acc = 0.0;   // sum of m_j * w(q_ij), normalised once at the end
    for (int k = 0; k < kmax; k++) {
        const double4 pj = p1;
        const bool act = k < cnt;
        j1 = j2;
        if (k + 2 < cnt) j2 = mine(k + 2);
        if (k + 1 < cnt) p1 = drec[j1];          // idle lanes issue no gather
        density_visit(pi, pj, act, lds_w, inv_h, inv_dq, pc.nq, acc);
    }
    if (!live) return;
    density_epilogue(pc, i, pi, acc, lds_w[0], u, alpha, vx, vy, vz, rho, P, cs, frec);
}

// P, c and the force records from an unchanged rho (SPH_FLAG_REUSE_DENSITY)
__global__ __launch_bounds__(256) void eos_only_kernel(PairConst pc, int64_t n, const double4 *__restrict__ drec,
                                                       const double *__restrict__ u, const double *__restrict__ alpha,
                                                       const double *__restrict__ vx, const double *__restrict__ vy,
                                                       const double *__restrict__ vz, const double *__restrict__ rho,
                                                       double *__restrict__ P, double *__restrict__ cs,
                                                       double *__restrict__ frec, const int32_t *__restrict__ orig,
                                                       int32_t skip_below) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    if (orig && orig[i] < skip_below) return;          // ghosts only: the owned records are current (density epilogue)
    const double r = rho[i];
    const double Pi = pc.gamma_m1 * u[i] * r;
    const double ci = sqrt(pc.gamma * Pi / r);
    P[i] = Pi; cs[i] = ci;
    write_frec(frec, i, drec[i], vx[i], vy[i], vz[i], r, Pi / (r * r), ci, alpha[i], pc.h);
}

// ------------------------------------------------------------------------------------------
// forces: sink gravity on the gas, SPH pressure + artificial viscosity, du/dt, dalpha/dt
// ------------------------------------------------------------------------------------------
template <int BLOCK, bool PACKED>
__global__ __launch_bounds__(BLOCK) void forces_kernel(PairConst pc, const double *__restrict__ frec,
                                                       const int32_t *__restrict__ nlist, int32_t cap,
                                                       const int32_t *__restrict__ ncount,
                                                       const int32_t *__restrict__ wave_max,
                                                       const double *__restrict__ dw_tab,
                                                       const double *__restrict__ sink, int64_t n,
                                                       double *__restrict__ ax, double *__restrict__ ay,
                                                       double *__restrict__ az, double *__restrict__ du,
                                                       double *__restrict__ dalpha, const int32_t *__restrict__ orig,
                                                       int32_t n_owned, const int32_t *__restrict__ wave_class, int32_t want,
                                                       const int32_t *__restrict__ plan_f) {
    extern __shared__ double lds_dw[];
    const int64_t i = (int64_t)xcd_chunk(blockIdx.x, gridDim.x) * BLOCK + threadIdx.x;
    if (wave_class) {       // split evaluation (multi-GPU overlap): only the waves of class `want`; a block with none leaves
        bool any = false;   // before it loads the table (workgroup-uniform: every thread looks at the block's waves)
        const int64_t w0 = (i - threadIdx.x) >> 6;
        for (int k = 0; k < BLOCK / 64; k++)
            any |= ((w0 + k) << 6) < n && wave_class[w0 + k] == want;
        if (!any) return;
    }
    for (int k = threadIdx.x; k < TAB_LEN(pc.nq); k += BLOCK) lds_dw[k] = dw_tab[k];
    __syncthreads();

    if ((i & ~(int64_t)63) >= n) return;
    if (wave_class && wave_class[i >> 6] != want) return;
    const int lane = threadIdx.x & 63;
    const int64_t w = i >> 6;
    const bool live = i < n && orig[i] < n_owned;
    const int self = i < n ? (int)i : (int)(n - 1);
    const double4 *fi = reinterpret_cast<const double4 *>(frec + (size_t)self * FREC);
    const double4 A = fi[0], B = fi[1], Cc = fi[2];   // x y z m | vx vy vz rho/2 | c/2 alpha/2 P/rho^2 -
    const int cnt = live ? min(ncount[i], cap) : 0;
    const int kmax = wave_max[w];
    const ListColumn<PACKED> mine(nlist, plan_f, w, cap, lane, self);
    const double inv_h = pc.inv_h, inv_dq = pc.inv_dq;

    ForceSums f;
    auto dw_of = [&](double q) { return table_knots_at(lds_dw, knot_coord(q, inv_dq)); };
    int j1 = 0 < cnt ? mine(0) : self;
    int j2 = 1 < cnt ? mine(1) : self;
    const double4 *fj = reinterpret_cast<const double4 *>(frec + (size_t)j1 * FREC);
    double4 A1 = fj[0], B1 = fj[1], C1 = fj[2];
    for (int k = 0; k < kmax; k++) {
        const Nbr nb = nbr_of(A1, B1, C1);
        const bool act = k < cnt;
        j1 = j2;
        if (k + 2 < cnt) j2 = mine(k + 2);
        if (k + 1 < cnt) {                       // idle lanes issue no gather
            fj = reinterpret_cast<const double4 *>(frec + (size_t)j1 * FREC);
            A1 = fj[0]; B1 = fj[1]; C1 = fj[2];
        }
        force_visit(pc, inv_h, A, B, Cc, nb, act, dw_of, f);
    }
    if (!live) return;
    force_epilogue(pc, sink, i, inv_h, A, B, Cc, f, ax, ay, az, du, dalpha);
}

// ------------------------------------------------------------------------------------------
// acceleration of the sinks: sum over all gas particles, then sink-sink pairs ([F]:567-591)
// two stages, fixed order -> bitwise reproducible
// ------------------------------------------------------------------------------------------
constexpr int SA_BLOCK = 256;

__device__ __forceinline__ double wave_sum(double v) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__global__ __launch_bounds__(SA_BLOCK) void sink_accel_partial(PairConst pc, const double4 *__restrict__ drec, int64_t n,
                                                               const double *__restrict__ sink, double *__restrict__ part,
                                                               const int32_t *__restrict__ orig, int32_t n_owned) {
    __shared__ double sm[3][SA_BLOCK / WAVE];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int s = 0; s < pc.ns; s++) {
        const double sx = sink[0 * MAX_SINKS + s], sy = sink[1 * MAX_SINKS + s], sz = sink[2 * MAX_SINKS + s];
        double b0 = 0.0, b1 = 0.0, b2 = 0.0;
        for (int64_t j = (int64_t)blockIdx.x * SA_BLOCK + threadIdx.x; j < n; j += (int64_t)gridDim.x * SA_BLOCK) {
            if (orig[j] >= n_owned) continue;                                       // ghosts are summed by their owners
            const double4 p = drec[j];
            const double v0 = p.x - sx, v1 = p.y - sy, v2 = p.z - sz;              // [F]:569
            const double dr = sqrt(v0 * v0 + v1 * v1 + v2 * v2);                   // [F]:570
            const double d3 = dr * dr * dr;
            b0 = b0 + (p.w * (pc.G * v0 / d3)); b1 = b1 + (p.w * (pc.G * v1 / d3)); b2 = b2 + (p.w * (pc.G * v2 / d3));  // [F]:572-573
        }
        b0 = wave_sum(b0); b1 = wave_sum(b1); b2 = wave_sum(b2);
        if (lane == 0) { sm[0][wv] = b0; sm[1][wv] = b1; sm[2][wv] = b2; }
        __syncthreads();
        if (threadIdx.x < 3) {
            double r = 0.0;
            for (int k = 0; k < SA_BLOCK / WAVE; k++) r += sm[threadIdx.x][k];
            part[((size_t)blockIdx.x * MAX_SINKS + s) * 3 + threadIdx.x] = r;
        }
        __syncthreads();
    }
}

// one block per sink, one wave per component: lanes stride over the per-block partials, then a
// fixed-order wave reduction (bitwise reproducible)
__global__ __launch_bounds__(192) void sink_accel_final(const double *__restrict__ part, int nblocks, double *__restrict__ sink) {
    const int s = blockIdx.x, comp = threadIdx.x >> 6, lane = threadIdx.x & 63;
    double b = 0.0;
    for (int k = lane; k < nblocks; k += 64) b += part[((size_t)k * MAX_SINKS + s) * 3 + comp];
    b = wave_sum(b);
    if (lane == 0) sink[(7 + comp) * MAX_SINKS + s] = b;
}

// sink-sink pairs, [F]:578-590: serial, as in the reference (ns is tiny)
__global__ void sink_sink_kernel(PairConst pc, double *__restrict__ sink) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    for (int i = 0; i < pc.ns; i++) {
        for (int j = 0; j < i; j++) {
            const double v0 = sink[0 * MAX_SINKS + j] - sink[0 * MAX_SINKS + i];
            const double v1 = sink[1 * MAX_SINKS + j] - sink[1 * MAX_SINKS + i];
            const double v2 = sink[2 * MAX_SINKS + j] - sink[2 * MAX_SINKS + i];
            const double dr = sqrt(v0 * v0 + v1 * v1 + v2 * v2);
            const double d3 = dr * dr * dr;
            const double w0 = pc.G * v0 / d3, w1 = pc.G * v1 / d3, w2 = pc.G * v2 / d3;
            const double mi = sink[6 * MAX_SINKS + i], mj = sink[6 * MAX_SINKS + j];
            sink[7 * MAX_SINKS + i] += mj * w0; sink[8 * MAX_SINKS + i] += mj * w1; sink[9 * MAX_SINKS + i] += mj * w2;
            sink[7 * MAX_SINKS + j] -= mi * w0; sink[8 * MAX_SINKS + j] -= mi * w1; sink[9 * MAX_SINKS + j] -= mi * w2;
        }
    }
}

}  // namespace

PairConst make_pair_const(const sph_ctx *c) {
    PairConst pc{};
    const sph_params &p = c->p;
    pc.h = p.h;
    pc.nq = p.nq;
    pc.dq = 2.0 / p.nq;                                   // [F]:10
    pc.inv_h = 1.0 / p.h;
    pc.inv_dq = 0.5 * p.nq;
    pc.wnorm = p.kernel_pi * (p.h * p.h * p.h);           // [F]:125
    pc.inv_dwnorm = 1.0 / (p.kernel_pi * (p.h * p.h * p.h * p.h));    // [F]:126
    pc.visc_eps_h2 = p.visc_eps * p.h * p.h;              // [F]:373
    pc.alpha_floor = p.alpha_floor; pc.alpha_decay = p.alpha_decay;
    pc.G = p.G; pc.gamma = p.gamma; pc.gamma_m1 = p.gamma_m1;
    pc.rcut2 = 4.0 * p.h * p.h * (1.0 + 1e-12);
    pc.ns = c->ns;
    pc.grav = c->gravity ? 1 : 0;
    pc.kernel_pi = p.kernel_pi; pc.eta = p.eta; pc.h_tol = p.h_tol; pc.h_max_length = p.h_max_length;
    pc.h_min_length = p.h_min_length; pc.h_iter_cap = p.h_iter_cap; pc.dt_scale = p.dt_scale;
    if (c->variable) pc.visc_eps_h2 = p.visc_eps;        // multiplied by avg_len^2 per pair (Variable.f90:405)
    return pc;
}

static inline unsigned pair_blocks(int64_t n) { return (unsigned)((n + PAIR_BLOCK - 1) / PAIR_BLOCK); }

#define NL_CHECK(expr)                                                      \
    do {                                                                    \
        hipError_t _e = (expr);                                             \
        if (_e != hipSuccess) {                                             \
            c->err = std::string(#expr) + ": " + hipGetErrorString(_e);     \
            return SPH_ERR_HIP;                                             \
        }                                                                   \
    } while (0)

int nlist_build(sph_ctx *c) {
    const int64_t n = c->n;
    if (n == 0) return SPH_OK;
    const PairConst pc = make_pair_const(c);
    for (int attempt = 0; attempt < 8; attempt++) {
        // per-wave longest lists go to wave_class (written again by classify_waves only after the build), their maximum straight to the host
        nlist_kernel<<<dim3(pair_blocks(n)), dim3(PAIR_BLOCK), 0, c->stream>>>(
            c->grid, reinterpret_cast<const double4 *>(c->drec), c->cell_start, n, pc.rcut2, c->nl_cap, c->nlist,
            c->ncount, c->wave_max, c->wave_class, c->orig, (int32_t)c->n_owned);
        max_to_host<<<dim3(1), dim3(1024), 0, c->stream>>>(c->wave_class, (n + 63) / 64, reinterpret_cast<int32_t *>(c->h_pinned + 9));
        NL_CHECK(hipGetLastError());
        NL_CHECK(hipStreamSynchronize(c->stream));
        c->wave_class_valid = false;
        const int32_t mx = *reinterpret_cast<int32_t *>(c->h_pinned + 9);
        c->nl_max = mx;
        if (mx <= c->nl_cap) { c->nlist_builds++; return SPH_OK; }
        // grow: the list is sized for the densest wave; 288 GB of HBM make this cheap
        int32_t want = mx + mx / 8 + 8;
        ctx_free(c, c->nlist);
        c->nl_cap = want;
        if (ctx_alloc(c, &c->nlist, (size_t)c->nl_waves_cap * c->nl_cap * 64, "neighbour list") != SPH_OK) { c->nl_cap = 0; return SPH_ERR_NOMEM; }
    }
    c->err = "neighbour list did not converge";
    return SPH_ERR_STATE;
}

// workgroup size of the gather kernels: 256 threads.  64-thread workgroups (every CU busy at the reference's own problem sizes:
// 12 000 particles are 47 workgroups of 256) were measured and lose at every size -- 12 000 particles: density 0.037 vs 0.032 ms
// per pass, 100 000: 0.089 vs 0.048 (the 40-KB kernel table is loaded per workgroup); SPH_GATHER_BLOCK=64 keeps the A/B switch
static int gather_block(const sph_ctx *) {
    static const int forced = getenv("SPH_GATHER_BLOCK") ? atoi(getenv("SPH_GATHER_BLOCK")) : 0;
    return forced == 64 ? 64 : PAIR_BLOCK;
}

hipError_t launch_density(sph_ctx *c, const PairConst &pc) {
    if (c->n == 0) return hipSuccess;
    const size_t lds = (size_t)TAB_LEN(pc.nq) * sizeof(double);
    if (gather_block(c) == 64) {
        auto k64 = c->packed_list ? density_kernel<64, true> : density_kernel<64, false>;
        k64<<<dim3((unsigned)((c->n + 63) / 64)), dim3(64), lds, c->stream>>>(
            pc, reinterpret_cast<const double4 *>(c->drec), c->nlist, c->nl_cap, c->ncount, c->wave_max, c->w_tab, c->n,
            c->f[SPH_F_U], c->f[SPH_F_ALPHA], c->f[SPH_F_VX], c->f[SPH_F_VY], c->f[SPH_F_VZ], c->f[SPH_F_RHO], c->f[SPH_F_P],
            c->f[SPH_F_C], c->frec, c->orig, (int32_t)c->n_owned, c->plan_f);
        return hipGetLastError();
    }
    auto k = c->packed_list ? density_kernel<PAIR_BLOCK, true> : density_kernel<PAIR_BLOCK, false>;
    k<<<dim3(pair_blocks(c->n)), dim3(PAIR_BLOCK), lds, c->stream>>>(
        pc, reinterpret_cast<const double4 *>(c->drec), c->nlist, c->nl_cap, c->ncount, c->wave_max, c->w_tab, c->n,
        c->f[SPH_F_U], c->f[SPH_F_ALPHA], c->f[SPH_F_VX], c->f[SPH_F_VY], c->f[SPH_F_VZ], c->f[SPH_F_RHO], c->f[SPH_F_P],
        c->f[SPH_F_C], c->frec, c->orig, (int32_t)c->n_owned, c->plan_f);
    return hipGetLastError();
}

hipError_t launch_eos_only(sph_ctx *c, const PairConst &pc, bool ghosts_only) {
    if (c->n == 0) return hipSuccess;
    eos_only_kernel<<<dim3((unsigned)((c->n + 255) / 256)), dim3(256), 0, c->stream>>>(
        pc, c->n, reinterpret_cast<const double4 *>(c->drec), c->f[SPH_F_U], c->f[SPH_F_ALPHA], c->f[SPH_F_VX],
        c->f[SPH_F_VY], c->f[SPH_F_VZ], c->f[SPH_F_RHO], c->f[SPH_F_P], c->f[SPH_F_C], c->frec,
        ghosts_only ? c->orig : nullptr, (int32_t)c->n_owned);
    return hipGetLastError();
}

// part 0: every wave.  part 1 / 2: only the waves of class 0 (interior: no lane within 2h of a ghost box) / class 1
hipError_t launch_forces(sph_ctx *c, const PairConst &pc, int part) {
    if (c->n == 0) return hipSuccess;
    const size_t lds = (size_t)TAB_LEN(pc.nq) * sizeof(double);
    if (gather_block(c) == 64) {
        auto k64 = c->packed_list ? forces_kernel<64, true> : forces_kernel<64, false>;
        k64<<<dim3((unsigned)((c->n + 63) / 64)), dim3(64), lds, c->stream>>>(
            pc, c->frec, c->nlist, c->nl_cap, c->ncount, c->wave_max, c->dw_tab, c->sink, c->n, c->f[SPH_F_AX],
            c->f[SPH_F_AY], c->f[SPH_F_AZ], c->f[SPH_F_DU], c->f[SPH_F_DALPHA], c->orig, (int32_t)c->n_owned,
            part ? c->wave_class : nullptr, part == 2 ? 1 : 0, c->plan_f);
        return hipGetLastError();
    }
    auto k = c->packed_list ? forces_kernel<PAIR_BLOCK, true> : forces_kernel<PAIR_BLOCK, false>;
    k<<<dim3(pair_blocks(c->n)), dim3(PAIR_BLOCK), lds, c->stream>>>(
        pc, c->frec, c->nlist, c->nl_cap, c->ncount, c->wave_max, c->dw_tab, c->sink, c->n, c->f[SPH_F_AX],
        c->f[SPH_F_AY], c->f[SPH_F_AZ], c->f[SPH_F_DU], c->f[SPH_F_DALPHA], c->orig, (int32_t)c->n_owned,
        part ? c->wave_class : nullptr, part == 2 ? 1 : 0, c->plan_f);
    return hipGetLastError();
}

// wave_class[w] = 1 if a lane of wave w lies within `reach` of one of the boxes (the other GPUs' bounding boxes: all
// ghosts are inside them), else 0
__global__ __launch_bounds__(256) void classify_waves(const double4 *__restrict__ drec, int64_t n, const double *__restrict__ boxes,
                                                      int nbox, double reach, int32_t *__restrict__ wave_class) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    bool near = false;
    if (i < n) {
        const double4 p = drec[i];
        for (int b = 0; b < nbox; b++) {
            const double *q = boxes + 6 * b;
            const double dx = fmax(fmax(q[0] - p.x, p.x - q[3]), 0.0), dy = fmax(fmax(q[1] - p.y, p.y - q[4]), 0.0),
                         dz = fmax(fmax(q[2] - p.z, p.z - q[5]), 0.0);
            near |= dx <= reach && dy <= reach && dz <= reach;
        }
    }
    const bool any = __any(near);
    if ((threadIdx.x & 63) == 0 && (i & ~(int64_t)63) < n) wave_class[i >> 6] = any ? 1 : 0;
}

hipError_t launch_classify_waves(sph_ctx *c) {
    if (c->n == 0) return hipSuccess;
    classify_waves<<<dim3((unsigned)((c->n + 255) / 256)), dim3(256), 0, c->stream>>>(
        reinterpret_cast<const double4 *>(c->drec), c->n, c->bnd_boxes, c->n_bnd_boxes, 2.0 * c->p.h * (1.0 + 1e-9), c->wave_class);
    return hipGetLastError();
}

hipError_t launch_sink_accel(sph_ctx *c, const PairConst &pc) {
    if (pc.ns == 0) return hipSuccess;
    int nb = (int)std::min<int64_t>((c->n + SA_BLOCK - 1) / SA_BLOCK, c->sink_blocks);
    if (nb < 1) nb = 1;
    sink_accel_partial<<<dim3(nb), dim3(SA_BLOCK), 0, c->stream>>>(pc, reinterpret_cast<const double4 *>(c->drec), c->n,
                                                                    c->sink, c->sink_part, c->orig, (int32_t)c->n_owned);
    sink_accel_final<<<dim3(pc.ns), dim3(192), 0, c->stream>>>(c->sink_part, nb, c->sink);
    if (pc.ns >= 2 && c->rank == 0) sink_sink_kernel<<<dim3(1), dim3(64), 0, c->stream>>>(pc, c->sink);
    return hipGetLastError();
}

}  // namespace sph
