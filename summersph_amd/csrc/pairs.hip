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

#include "sph_internal.hpp"

namespace sph {

namespace {

constexpr int PAIR_BLOCK = 256;

__device__ __forceinline__ int xcd_chunk(int b, int nb) {
    // blocks b, b+8, b+16.. share an XCD (round-robin dispatch): give each XCD one
    // contiguous run of the cell-sorted particle order so neighbours hit the same L2.
    const int q = nb >> 3, r = nb & 7, x = b & 7, k = b >> 3;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + k;
}

__device__ __forceinline__ void cell_coords(const GridDesc &g, double px, double py, double pz, int cc[3]) {
    const double p[3] = {px, py, pz};
    int c[3];
#pragma unroll
    for (int a = 0; a < 3; a++) {
        int v = (int)((p[a] - g.org[a]) * g.inv_edge);
        c[a] = min(max(v, 0), g.dim[a] - 1);
    }
    cc[0] = c[g.s[0]]; cc[1] = c[g.s[1]]; cc[2] = c[g.s[2]];
}

__device__ __forceinline__ int wave_max_i32(int v) {
    for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o, 64));
    return v;
}

// ------------------------------------------------------------------------------------------
// neighbour list: every j != i with |x_i - x_j|^2 <= rcut2 (rcut2 a hair above (2h)^2; the
// evaluation kernels re-test q <= 2 exactly as lookup_kernel does, [F]:113)
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(PAIR_BLOCK) void nlist_kernel(GridDesc g, const double4 *__restrict__ drec,
                                                           const int32_t *__restrict__ cell_start, int64_t n,
                                                           double rcut2, int32_t cap, int32_t *__restrict__ nlist,
                                                           int32_t *__restrict__ ncount, int32_t *__restrict__ wave_max,
                                                           int32_t *__restrict__ flags) {
    const int64_t i = (int64_t)xcd_chunk(blockIdx.x, gridDim.x) * PAIR_BLOCK + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const int64_t w = i >> 6;
    int cnt = 0;
    if (i < n) {
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
                for (int j = jb; j < je; j++) {
                    const double4 pj = drec[j];
                    const double dx = pi.x - pj.x, dy = pi.y - pj.y, dz = pi.z - pj.z;
                    const double r2 = dx * dx + dy * dy + dz * dz;
                    if (r2 <= rcut2 && j != (int)i) {
                        if (cnt < cap) mine[(size_t)cnt * 64] = j;
                        cnt++;
                    }
                }
            }
        }
        ncount[i] = cnt;
    }
    const int wm = wave_max_i32(cnt);
    if (lane == 0 && (w << 6) < n) {
        wave_max[w] = min(wm, cap);
        if (wm > 0) atomicMax(&flags[1], wm);
    }
}

// [F]:105-127 with the table in LDS.  S-normalisation is applied by the caller.
__device__ __forceinline__ double table_lerp(const double *__restrict__ tab, double qi, double dq, int nq) {
    int k = (int)(qi / dq);
    k = min(k, nq - 1);
    const double a = (qi - k * dq) / dq;
    return (1.0 - a) * tab[k] + a * tab[k + 1];
}

// ------------------------------------------------------------------------------------------
// density + EOS
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(PAIR_BLOCK) void density_kernel(PairConst pc, const double4 *__restrict__ drec,
                                                             const int32_t *__restrict__ nlist, int32_t cap,
                                                             const int32_t *__restrict__ ncount,
                                                             const int32_t *__restrict__ wave_max,
                                                             const double *__restrict__ w_tab, int64_t n,
                                                             const double *__restrict__ u, const double *__restrict__ alpha,
                                                             const double *__restrict__ vx, const double *__restrict__ vy,
                                                             const double *__restrict__ vz, double *__restrict__ rho,
                                                             double *__restrict__ P, double *__restrict__ cs,
                                                             double *__restrict__ frec) {
    extern __shared__ double lds_w[];
    for (int k = threadIdx.x; k <= pc.nq; k += PAIR_BLOCK) lds_w[k] = w_tab[k];
    __syncthreads();

    const int64_t i = (int64_t)xcd_chunk(blockIdx.x, gridDim.x) * PAIR_BLOCK + threadIdx.x;
    if ((i & ~(int64_t)63) >= n) return;   // whole wave out of range
    const int lane = threadIdx.x & 63;
    const int64_t w = i >> 6;
    const bool live = i < n;
    const double4 pi = live ? drec[i] : make_double4(0, 0, 0, 0);
    const int cnt = live ? min(ncount[i], cap) : 0;
    const int kmax = wave_max[w];
    const int32_t *mine = nlist + ((size_t)w * cap) * 64 + lane;

    // self term: r = 0 -> W = w_table(0) = 1, [F]:443-455 visits the particle's own leaf too
    double acc = live ? pi.w * (lds_w[0] / pc.wnorm) : 0.0;
    for (int k = 0; k < kmax; k++) {
        if (k < cnt) {
            const int j = mine[(size_t)k * 64];
            const double4 pj = drec[j];
            const double n0 = pi.x - pj.x, n1 = pi.y - pj.y, n2 = pi.z - pj.z;    // [F]:445
            const double dr = sqrt(n0 * n0 + n1 * n1 + n2 * n2);                   // [F]:446
            const double qi = dr / pc.h;                                           // [F]:111
            if (qi <= 2.0) {                                                       // [F]:113
                const double Wj = table_lerp(lds_w, qi, pc.dq, pc.nq) / pc.wnorm;  // [F]:114-125
                acc = acc + pj.w * Wj;                                             // [F]:454
            }
        }
    }
    if (!live) return;
    // EOS, [F]:465-466
    const double ui = u[i];
    const double Pi = pc.gamma_m1 * ui * acc;
    const double ci = sqrt(pc.gamma * Pi / acc);
    rho[i] = acc; P[i] = Pi; cs[i] = ci;
    double *fr = frec + (size_t)i * FREC;
    reinterpret_cast<double4 *>(fr)[0] = pi;
    reinterpret_cast<double4 *>(fr)[1] = make_double4(vx[i], vy[i], vz[i], acc);
    reinterpret_cast<double4 *>(fr)[2] = make_double4(Pi / (acc * acc), ci, alpha[i], 0.0);   // [F]:381: P/(rho*rho)
}

// P, c and the force records from an unchanged rho (SPH_FLAG_REUSE_DENSITY)
__global__ __launch_bounds__(256) void eos_only_kernel(PairConst pc, int64_t n, const double4 *__restrict__ drec,
                                                       const double *__restrict__ u, const double *__restrict__ alpha,
                                                       const double *__restrict__ vx, const double *__restrict__ vy,
                                                       const double *__restrict__ vz, const double *__restrict__ rho,
                                                       double *__restrict__ P, double *__restrict__ cs,
                                                       double *__restrict__ frec) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const double r = rho[i];
    const double Pi = pc.gamma_m1 * u[i] * r;
    const double ci = sqrt(pc.gamma * Pi / r);
    P[i] = Pi; cs[i] = ci;
    double *fr = frec + (size_t)i * FREC;
    reinterpret_cast<double4 *>(fr)[0] = drec[i];
    reinterpret_cast<double4 *>(fr)[1] = make_double4(vx[i], vy[i], vz[i], r);
    reinterpret_cast<double4 *>(fr)[2] = make_double4(Pi / (r * r), ci, alpha[i], 0.0);
}

// ------------------------------------------------------------------------------------------
// forces: sink gravity on the gas, SPH pressure + artificial viscosity, du/dt, dalpha/dt
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(PAIR_BLOCK) void forces_kernel(PairConst pc, const double *__restrict__ frec,
                                                            const int32_t *__restrict__ nlist, int32_t cap,
                                                            const int32_t *__restrict__ ncount,
                                                            const int32_t *__restrict__ wave_max,
                                                            const double *__restrict__ dw_tab,
                                                            const double *__restrict__ sink, int64_t n,
                                                            double *__restrict__ ax, double *__restrict__ ay,
                                                            double *__restrict__ az, double *__restrict__ du,
                                                            double *__restrict__ dalpha) {
    extern __shared__ double lds_dw[];
    for (int k = threadIdx.x; k <= pc.nq; k += PAIR_BLOCK) lds_dw[k] = dw_tab[k];
    __syncthreads();

    const int64_t i = (int64_t)xcd_chunk(blockIdx.x, gridDim.x) * PAIR_BLOCK + threadIdx.x;
    if ((i & ~(int64_t)63) >= n) return;
    const int lane = threadIdx.x & 63;
    const int64_t w = i >> 6;
    const bool live = i < n;
    const int64_t ii = live ? i : 0;
    const double4 *fi = reinterpret_cast<const double4 *>(frec + (size_t)ii * FREC);
    const double4 A = fi[0], B = fi[1], Cc = fi[2];   // x y z m | vx vy vz rho | P/rho^2 c alpha -
    const int cnt = live ? min(ncount[i], cap) : 0;
    const int kmax = wave_max[w];
    const int32_t *mine = nlist + ((size_t)w * cap) * 64 + lane;

    // zero_rates, then the gas side of sink_gravforces, [F]:567-576 (same order as find_forces)
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, due = 0.0, dal = 0.0;
    for (int s = 0; s < pc.ns; s++) {
        const double v0 = A.x - sink[0 * MAX_SINKS + s], v1 = A.y - sink[1 * MAX_SINKS + s], v2 = A.z - sink[2 * MAX_SINKS + s];
        const double dr = sqrt(v0 * v0 + v1 * v1 + v2 * v2);
        const double d3 = dr * dr * dr;
        const double ms = sink[6 * MAX_SINKS + s];
        a0 = a0 - (ms * (pc.G * v0 / d3)); a1 = a1 - (ms * (pc.G * v1 / d3)); a2 = a2 - (ms * (pc.G * v2 / d3));
    }

    for (int k = 0; k < kmax; k++) {
        if (k < cnt) {
            const int j = mine[(size_t)k * 64];
            const double4 *fj = reinterpret_cast<const double4 *>(frec + (size_t)j * FREC);
            const double4 Aj = fj[0], Bj = fj[1], Cj = fj[2];
            double n0 = A.x - Aj.x, n1 = A.y - Aj.y, n2 = A.z - Aj.z;                 // [F]:356
            const double dr = sqrt(n0 * n0 + n1 * n1 + n2 * n2);                      // [F]:357
            const double qi = dr / pc.h;
            if (qi <= 2.0 && dr > 0.0) {    // beyond 2h every term is exactly 0; dr == 0: see DESIGN.md (coincident points)
                const double v0 = B.x - Bj.x, v1 = B.y - Bj.y, v2 = B.z - Bj.z;       // [F]:358
                double vdotr = v0 * n0 + v1 * n1 + v2 * n2;                           // [F]:359
                if (vdotr >= 0.0) vdotr = 0.0;                                        // [F]:361
                n0 = n0 / dr; n1 = n1 / dr; n2 = n2 / dr;                             // [F]:363
                const double dWm = table_lerp(lds_dw, qi, pc.dq, pc.nq) / pc.dwnorm;  // [F]:366,126
                const double g0 = n0 * dWm, g1 = n1 * dWm, g2 = n2 * dWm;             // [F]:368
                const double vdotgradW = g0 * v0 + g1 * v1 + g2 * v2;                 // [F]:370
                const double vis_nu = (pc.h * vdotr) / (dr * dr + pc.visc_eps_h2);    // [F]:373
                const double cbar = 0.5 * (Cc.y + Cj.y);                              // [F]:374
                const double abar = 0.5 * (Cc.z + Cj.z);                              // [F]:376
                const double visc = (-abar * cbar * vis_nu + 2.0 * abar * vis_nu * vis_nu) / (0.5 * (B.w + Bj.w));  // [F]:378
                const double Cf = Cc.x + Cj.x + visc;                                 // [F]:381-382
                const double mj = Aj.w;
                a0 = a0 - mj * (Cf * g0); a1 = a1 - mj * (Cf * g1); a2 = a2 - mj * (Cf * g2);   // [F]:383
                due = due + mj * vdotgradW * (Cc.x + 0.5 * visc);                     // [F]:387
                dal = dal + mj * vdotgradW;                                           // [F]:390
            }
        }
    }
    if (!live) return;
    ax[i] = a0; ay[i] = a1; az[i] = a2; du[i] = due;
    // [F]:317
    dalpha[i] = fmax(dal / B.w, 0.0) + pc.alpha_decay * ((pc.alpha_floor - Cc.z) * Cc.y / pc.h);
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
                                                               const double *__restrict__ sink, double *__restrict__ part) {
    __shared__ double sm[3][SA_BLOCK / WAVE];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int s = 0; s < pc.ns; s++) {
        const double sx = sink[0 * MAX_SINKS + s], sy = sink[1 * MAX_SINKS + s], sz = sink[2 * MAX_SINKS + s];
        double b0 = 0.0, b1 = 0.0, b2 = 0.0;
        for (int64_t j = (int64_t)blockIdx.x * SA_BLOCK + threadIdx.x; j < n; j += (int64_t)gridDim.x * SA_BLOCK) {
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

__global__ void sink_accel_final(PairConst pc, const double *__restrict__ part, int nblocks, double *__restrict__ sink) {
    const int s = threadIdx.x;
    if (s < pc.ns) {
        double b0 = 0.0, b1 = 0.0, b2 = 0.0;
        for (int b = 0; b < nblocks; b++) {
            const double *p = part + ((size_t)b * MAX_SINKS + s) * 3;
            b0 += p[0]; b1 += p[1]; b2 += p[2];
        }
        sink[7 * MAX_SINKS + s] = b0; sink[8 * MAX_SINKS + s] = b1; sink[9 * MAX_SINKS + s] = b2;
    }
    __syncthreads();
    // sink-sink pairs, [F]:578-590: serial, as in the reference (ns is tiny)
    if (threadIdx.x == 0 && pc.ns >= 2) {
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
}

}  // namespace

PairConst make_pair_const(const sph_ctx *c) {
    PairConst pc{};
    const sph_params &p = c->p;
    pc.h = p.h;
    pc.nq = p.nq;
    pc.dq = 2.0 / p.nq;                                   // [F]:10
    pc.wnorm = p.kernel_pi * (p.h * p.h * p.h);           // [F]:125
    pc.dwnorm = p.kernel_pi * (p.h * p.h * p.h * p.h);    // [F]:126
    pc.visc_eps_h2 = p.visc_eps * p.h * p.h;              // [F]:373
    pc.alpha_floor = p.alpha_floor; pc.alpha_decay = p.alpha_decay;
    pc.G = p.G; pc.gamma = p.gamma; pc.gamma_m1 = p.gamma_m1;
    pc.rcut2 = 4.0 * p.h * p.h * (1.0 + 1e-12);
    pc.ns = c->ns;
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
        NL_CHECK(hipMemsetAsync(c->d_flags + 1, 0, sizeof(int32_t), c->stream));
        nlist_kernel<<<dim3(pair_blocks(n)), dim3(PAIR_BLOCK), 0, c->stream>>>(
            c->grid, reinterpret_cast<const double4 *>(c->drec), c->cell_start, n, pc.rcut2, c->nl_cap, c->nlist,
            c->ncount, c->wave_max, c->d_flags);
        NL_CHECK(hipGetLastError());
        NL_CHECK(hipMemcpyAsync(c->h_pinned + 9, c->d_flags + 1, sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
        NL_CHECK(hipStreamSynchronize(c->stream));
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

hipError_t launch_density(sph_ctx *c, const PairConst &pc) {
    if (c->n == 0) return hipSuccess;
    const size_t lds = (size_t)(pc.nq + 1) * sizeof(double);
    density_kernel<<<dim3(pair_blocks(c->n)), dim3(PAIR_BLOCK), lds, c->stream>>>(
        pc, reinterpret_cast<const double4 *>(c->drec), c->nlist, c->nl_cap, c->ncount, c->wave_max, c->w_tab, c->n,
        c->f[SPH_F_U], c->f[SPH_F_ALPHA], c->f[SPH_F_VX], c->f[SPH_F_VY], c->f[SPH_F_VZ], c->f[SPH_F_RHO], c->f[SPH_F_P],
        c->f[SPH_F_C], c->frec);
    return hipGetLastError();
}

hipError_t launch_eos_only(sph_ctx *c, const PairConst &pc) {
    if (c->n == 0) return hipSuccess;
    eos_only_kernel<<<dim3((unsigned)((c->n + 255) / 256)), dim3(256), 0, c->stream>>>(
        pc, c->n, reinterpret_cast<const double4 *>(c->drec), c->f[SPH_F_U], c->f[SPH_F_ALPHA], c->f[SPH_F_VX],
        c->f[SPH_F_VY], c->f[SPH_F_VZ], c->f[SPH_F_RHO], c->f[SPH_F_P], c->f[SPH_F_C], c->frec);
    return hipGetLastError();
}

hipError_t launch_forces(sph_ctx *c, const PairConst &pc) {
    if (c->n == 0) return hipSuccess;
    const size_t lds = (size_t)(pc.nq + 1) * sizeof(double);
    forces_kernel<<<dim3(pair_blocks(c->n)), dim3(PAIR_BLOCK), lds, c->stream>>>(
        pc, c->frec, c->nlist, c->nl_cap, c->ncount, c->wave_max, c->dw_tab, c->sink, c->n, c->f[SPH_F_AX],
        c->f[SPH_F_AY], c->f[SPH_F_AZ], c->f[SPH_F_DU], c->f[SPH_F_DALPHA]);
    return hipGetLastError();
}

hipError_t launch_sink_accel(sph_ctx *c, const PairConst &pc) {
    if (pc.ns == 0) return hipSuccess;
    int nb = (int)std::min<int64_t>((c->n + SA_BLOCK - 1) / SA_BLOCK, c->sink_blocks);
    if (nb < 1) nb = 1;
    sink_accel_partial<<<dim3(nb), dim3(SA_BLOCK), 0, c->stream>>>(pc, reinterpret_cast<const double4 *>(c->drec), c->n,
                                                                    c->sink, c->sink_part);
    sink_accel_final<<<dim3(1), dim3(MAX_SINKS), 0, c->stream>>>(pc, c->sink_part, nb, c->sink);
    return hipGetLastError();
}

}  // namespace sph
