// integrate.hip -- leapfrog kick / drift and the adaptive global time step.
//
// Replaces (citations: /root/reference/SUMMER_SPH.f90, "[F]")
//   kick               [F]:742-759
//   drift              [F]:762-776
//   get_next_timestep  [F]:831-860
// All three are pure HBM streaming kernels (80+40, 48+24 and 80 B per particle).  dt lives in
// device memory (ctx->d_dt[0]) so that whole steps can be enqueued without a host round trip.
#include <cmath>

#include "sph_internal.hpp"
#include "integ_common.hpp"

namespace sph {

namespace {

constexpr int EW_BLOCK = 256;
constexpr int DT_BLOCK = 256;

struct KickArgs {
    double *vx, *vy, *vz, *u, *alpha;
    const double *ax, *ay, *az, *du, *dalpha;
};

__global__ __launch_bounds__(EW_BLOCK) void kick_kernel(KickArgs a, int64_t n, double dt_val, const double *__restrict__ dt_ptr,
                                                        double *__restrict__ sink, int ns) {
    const double dt = dt_ptr ? dt_ptr[0] : dt_val;
    const int64_t i = (int64_t)blockIdx.x * EW_BLOCK + threadIdx.x;
    if (i < n) {
        a.vx[i] = kick_half(a.vx[i], a.ax[i], dt);            // [F]:749-751
        a.vy[i] = kick_half(a.vy[i], a.ay[i], dt);
        a.vz[i] = kick_half(a.vz[i], a.az[i], dt);
        a.u[i] = kick_half(a.u[i], a.du[i], dt);              // [F]:757
        a.alpha[i] = kick_alpha(a.alpha[i], a.dalpha[i], dt); // [F]:758
    }
    if (blockIdx.x == 0 && threadIdx.x < ns) {                // [F]:753-755
        const int s = threadIdx.x;
        for (int k = 0; k < 3; k++)
            sink[(3 + k) * MAX_SINKS + s] = sink[(3 + k) * MAX_SINKS + s] + 0.5 * sink[(7 + k) * MAX_SINKS + s] * dt;
    }
}

__global__ __launch_bounds__(EW_BLOCK) void drift_kernel(double *__restrict__ x, double *__restrict__ y, double *__restrict__ z,
                                                         const double *__restrict__ vx, const double *__restrict__ vy,
                                                         const double *__restrict__ vz, int64_t n, double dt_val,
                                                         const double *__restrict__ dt_ptr, double *__restrict__ sink, int ns) {
    const double dt = dt_ptr ? dt_ptr[0] : dt_val;
    const int64_t i = (int64_t)blockIdx.x * EW_BLOCK + threadIdx.x;
    if (i < n) {
        x[i] = drift_pos(x[i], vx[i], dt);                    // [F]:769-771
        y[i] = drift_pos(y[i], vy[i], dt);
        z[i] = drift_pos(z[i], vz[i], dt);
    }
    if (blockIdx.x == 0 && threadIdx.x < ns) {                // [F]:773-775
        const int s = threadIdx.x;
        for (int k = 0; k < 3; k++)
            sink[k * MAX_SINKS + s] = sink[k * MAX_SINKS + s] + sink[(3 + k) * MAX_SINKS + s] * dt;
    }
}

// kick immediately followed by drift (first half of a step, [F]:899-900): one pass instead of two; same expressions,
// same results as kick_kernel + drift_kernel
__global__ __launch_bounds__(EW_BLOCK) void kick_drift_kernel(KickArgs a, double *__restrict__ x, double *__restrict__ y,
                                                              double *__restrict__ z, int64_t n, const double *__restrict__ dt_ptr,
                                                              double *__restrict__ sink, int ns) {
    const double dt = dt_ptr[0];
    const int64_t i = (int64_t)blockIdx.x * EW_BLOCK + threadIdx.x;
    if (i < n) {
        const double vx = kick_half(a.vx[i], a.ax[i], dt), vy = kick_half(a.vy[i], a.ay[i], dt), vz = kick_half(a.vz[i], a.az[i], dt);
        a.vx[i] = vx; a.vy[i] = vy; a.vz[i] = vz;
        a.u[i] = kick_half(a.u[i], a.du[i], dt);
        a.alpha[i] = kick_alpha(a.alpha[i], a.dalpha[i], dt);
        x[i] = drift_pos(x[i], vx, dt);
        y[i] = drift_pos(y[i], vy, dt);
        z[i] = drift_pos(z[i], vz, dt);
    }
    if (blockIdx.x == 0 && threadIdx.x < ns) {
        const int s = threadIdx.x;
        for (int k = 0; k < 3; k++) {
            const double v = sink[(3 + k) * MAX_SINKS + s] + 0.5 * sink[(7 + k) * MAX_SINKS + s] * dt;
            sink[(3 + k) * MAX_SINKS + s] = v;
            sink[k * MAX_SINKS + s] = sink[k * MAX_SINKS + s] + v * dt;
        }
    }
}

__device__ __forceinline__ double wave_min(double v) {
    for (int o = 32; o > 0; o >>= 1) v = fmin(v, __shfl_xor(v, o, 64));
    return v;
}

// per-block minimum of the four candidates of [F]:845-850
__global__ __launch_bounds__(DT_BLOCK) void dt_partial(const double *__restrict__ vx, const double *__restrict__ vy,
                                                       const double *__restrict__ vz, const double *__restrict__ ax,
                                                       const double *__restrict__ ay, const double *__restrict__ az,
                                                       const double *__restrict__ u, const double *__restrict__ du,
                                                       const double *__restrict__ cs, double h, int64_t n,
                                                       double *__restrict__ part, const int32_t *__restrict__ orig,
                                                       int32_t n_owned, const double *__restrict__ hvar) {
    __shared__ double sm[DT_BLOCK / WAVE];
    double mn = INFINITY;
    for (int64_t i = (int64_t)blockIdx.x * DT_BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * DT_BLOCK) {
        if (orig[i] >= n_owned) continue;                     // ghosts are timed by their owners
        const double hi = hvar ? hvar[i] : h;                 // per-particle h: Variable.f90:1053-1054
        mn = fmin(mn, dt_candidates(vx[i], vy[i], vz[i], ax[i], ay[i], az[i], u[i], du[i], cs[i], hi));      // [F]:846-849
    }
    mn = wave_min(mn);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = mn;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int k = 1; k < DT_BLOCK / WAVE; k++) mn = fmin(mn, sm[k]);
        part[blockIdx.x] = mn;
    }
}

// the closing kick of a step immediately followed by get_next_timestep's per-particle part ([F]:910-911,845-850): one
// pass; same expressions as kick_kernel + dt_partial (the minimum does not depend on the order)
__global__ __launch_bounds__(DT_BLOCK) void kick_dt_kernel(KickArgs a, const double *__restrict__ cs, double h, int64_t n,
                                                           const double *__restrict__ dt_ptr, double *__restrict__ sink, int ns,
                                                           double *__restrict__ part, const int32_t *__restrict__ orig,
                                                           int32_t n_owned, const double *__restrict__ hvar) {
    __shared__ double sm[DT_BLOCK / WAVE];
    const double dt = dt_ptr[0];
    double mn = INFINITY;
    for (int64_t i = (int64_t)blockIdx.x * DT_BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * DT_BLOCK) {
        const double ax = a.ax[i], ay = a.ay[i], az = a.az[i], du = a.du[i];
        const double vx = kick_half(a.vx[i], ax, dt), vy = kick_half(a.vy[i], ay, dt), vz = kick_half(a.vz[i], az, dt);
        const double u = kick_half(a.u[i], du, dt);
        a.vx[i] = vx; a.vy[i] = vy; a.vz[i] = vz; a.u[i] = u;
        a.alpha[i] = kick_alpha(a.alpha[i], a.dalpha[i], dt);
        if (orig[i] >= n_owned) continue;                     // ghosts are timed by their owners
        const double hi = hvar ? hvar[i] : h;
        mn = fmin(mn, dt_candidates(vx, vy, vz, ax, ay, az, u, du, cs[i], hi));      // [F]:846-849
    }
    mn = wave_min(mn);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = mn;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int k = 1; k < DT_BLOCK / WAVE; k++) mn = fmin(mn, sm[k]);
        part[blockIdx.x] = mn;
    }
    if (blockIdx.x == 0 && threadIdx.x < ns) {                // [F]:753-755
        const int s = threadIdx.x;
        for (int k = 0; k < 3; k++)
            sink[(3 + k) * MAX_SINKS + s] = sink[(3 + k) * MAX_SINKS + s] + 0.5 * sink[(7 + k) * MAX_SINKS + s] * dt;
    }
}

// dtbuf: [0] dt, [1] t, [2] dt candidate.  [F]:851-859 (+ t = t + dt of [F]:914 when advance_t)
__global__ void dt_final(const double *__restrict__ part, int nblocks, double dt_scale, double dt_max, double dt_min,
                         int advance_t, double *__restrict__ dtbuf) {
    double mn = INFINITY;
    for (int b = threadIdx.x; b < nblocks; b += 64) mn = fmin(mn, part[b]);
    mn = wave_min(mn);
    if (threadIdx.x == 0) {
        const double cand = mn * dt_scale;                    // [F]:851
        double dt = dtbuf[0];
        if (advance_t) dtbuf[1] = dtbuf[1] + dt;              // [F]:914 (before the dt update)
        if (cand > 2 * dt && 1.5 * dt < dt_max) dt = 1.5 * dt;            // [F]:855-856
        else if (cand < 0.5 * dt && dt * 0.5 > dt_min) dt = 0.5 * dt;     // [F]:857-858
        dtbuf[0] = dt;
        dtbuf[2] = cand;
    }
}

// multi-GPU: only the local candidate, the launcher reduces it over ranks and applies [F]:855-858
__global__ void dt_candidate_only(const double *__restrict__ part, int nblocks, double dt_scale, double *__restrict__ dtbuf) {
    double mn = INFINITY;
    for (int b = threadIdx.x; b < nblocks; b += 64) mn = fmin(mn, part[b]);
    mn = wave_min(mn);
    if (threadIdx.x == 0) dtbuf[2] = mn * dt_scale;
}

}  // namespace

hipError_t launch_dt_partial_only(sph_ctx *c) {
    int nb = (int)std::min<int64_t>((c->n + DT_BLOCK - 1) / DT_BLOCK, c->dt_blocks);
    if (nb < 1) nb = 1;
    dt_partial<<<dim3(nb), dim3(DT_BLOCK), 0, c->stream>>>(c->f[SPH_F_VX], c->f[SPH_F_VY], c->f[SPH_F_VZ], c->f[SPH_F_AX],
                                                           c->f[SPH_F_AY], c->f[SPH_F_AZ], c->f[SPH_F_U], c->f[SPH_F_DU],
                                                           c->f[SPH_F_C], c->p.h, c->n, c->dt_part, c->orig, (int32_t)c->n_owned,
                                                           c->variable ? c->f[SPH_F_H] : nullptr);
    dt_candidate_only<<<dim3(1), dim3(64), 0, c->stream>>>(c->dt_part, nb, c->p.dt_scale, c->d_dt);
    return hipGetLastError();
}

hipError_t launch_kick(sph_ctx *c, double dt, bool dt_from_device) {
    KickArgs a{c->f[SPH_F_VX], c->f[SPH_F_VY], c->f[SPH_F_VZ], c->f[SPH_F_U], c->f[SPH_F_ALPHA],
               c->f[SPH_F_AX], c->f[SPH_F_AY], c->f[SPH_F_AZ], c->f[SPH_F_DU], c->f[SPH_F_DALPHA]};
    unsigned nb = (unsigned)std::max<int64_t>((c->n + EW_BLOCK - 1) / EW_BLOCK, 1);
    kick_kernel<<<dim3(nb), dim3(EW_BLOCK), 0, c->stream>>>(a, c->n, dt, dt_from_device ? c->d_dt : nullptr, c->sink, c->ns);
    return hipGetLastError();
}

hipError_t launch_kick_drift(sph_ctx *c) {
    KickArgs a{c->f[SPH_F_VX], c->f[SPH_F_VY], c->f[SPH_F_VZ], c->f[SPH_F_U], c->f[SPH_F_ALPHA],
               c->f[SPH_F_AX], c->f[SPH_F_AY], c->f[SPH_F_AZ], c->f[SPH_F_DU], c->f[SPH_F_DALPHA]};
    unsigned nb = (unsigned)std::max<int64_t>((c->n + EW_BLOCK - 1) / EW_BLOCK, 1);
    kick_drift_kernel<<<dim3(nb), dim3(EW_BLOCK), 0, c->stream>>>(a, c->f[SPH_F_X], c->f[SPH_F_Y], c->f[SPH_F_Z], c->n, c->d_dt, c->sink, c->ns);
    return hipGetLastError();
}

hipError_t launch_drift(sph_ctx *c, double dt, bool dt_from_device) {
    unsigned nb = (unsigned)std::max<int64_t>((c->n + EW_BLOCK - 1) / EW_BLOCK, 1);
    drift_kernel<<<dim3(nb), dim3(EW_BLOCK), 0, c->stream>>>(c->f[SPH_F_X], c->f[SPH_F_Y], c->f[SPH_F_Z], c->f[SPH_F_VX],
                                                              c->f[SPH_F_VY], c->f[SPH_F_VZ], c->n, dt,
                                                              dt_from_device ? c->d_dt : nullptr, c->sink, c->ns);
    return hipGetLastError();
}

hipError_t launch_kick_next_dt(sph_ctx *c, bool advance_t) {
    KickArgs a{c->f[SPH_F_VX], c->f[SPH_F_VY], c->f[SPH_F_VZ], c->f[SPH_F_U], c->f[SPH_F_ALPHA],
               c->f[SPH_F_AX], c->f[SPH_F_AY], c->f[SPH_F_AZ], c->f[SPH_F_DU], c->f[SPH_F_DALPHA]};
    int nb = (int)std::min<int64_t>((c->n + DT_BLOCK - 1) / DT_BLOCK, c->dt_blocks);
    if (nb < 1) nb = 1;
    kick_dt_kernel<<<dim3(nb), dim3(DT_BLOCK), 0, c->stream>>>(a, c->f[SPH_F_C], c->p.h, c->n, c->d_dt, c->sink, c->ns, c->dt_part, c->orig,
                                                               (int32_t)c->n_owned, c->variable ? c->f[SPH_F_H] : nullptr);
    dt_final<<<dim3(1), dim3(64), 0, c->stream>>>(c->dt_part, nb, c->p.dt_scale, c->p.dt_max, c->p.dt_min, advance_t ? 1 : 0, c->d_dt);
    return hipGetLastError();
}

// the sinks' half kick alone ([F]:753-755), for a gas kick launched with ns = 0
__global__ void sink_kick_kernel(const double *__restrict__ dt_ptr, double *__restrict__ sink, int ns) {
    const int s = threadIdx.x;
    if (s >= ns) return;
    const double dt = dt_ptr[0];
    for (int k = 0; k < 3; k++)
        sink[(3 + k) * MAX_SINKS + s] = sink[(3 + k) * MAX_SINKS + s] + 0.5 * sink[(7 + k) * MAX_SINKS + s] * dt;
}

hipError_t launch_kick_sinks(sph_ctx *c) {
    if (c->ns > 0) sink_kick_kernel<<<dim3(1), dim3(64), 0, c->stream>>>(c->d_dt, c->sink, c->ns);
    return hipGetLastError();
}

// multi-GPU: the closing kick + the LOCAL dt candidate in one pass (kick_dt_kernel), the rule itself after the rank reduction;
// with_sinks = false leaves the sinks' velocities alone (their accelerations may still be travelling between the ranks)
hipError_t launch_kick_dt_candidate(sph_ctx *c, bool with_sinks) {
    KickArgs a{c->f[SPH_F_VX], c->f[SPH_F_VY], c->f[SPH_F_VZ], c->f[SPH_F_U], c->f[SPH_F_ALPHA],
               c->f[SPH_F_AX], c->f[SPH_F_AY], c->f[SPH_F_AZ], c->f[SPH_F_DU], c->f[SPH_F_DALPHA]};
    int nb = (int)std::min<int64_t>((c->n + DT_BLOCK - 1) / DT_BLOCK, c->dt_blocks);
    if (nb < 1) nb = 1;
    kick_dt_kernel<<<dim3(nb), dim3(DT_BLOCK), 0, c->stream>>>(a, c->f[SPH_F_C], c->p.h, c->n, c->d_dt, c->sink, with_sinks ? c->ns : 0, c->dt_part,
                                                               c->orig, (int32_t)c->n_owned, c->variable ? c->f[SPH_F_H] : nullptr);
    dt_candidate_only<<<dim3(1), dim3(64), 0, c->stream>>>(c->dt_part, nb, c->p.dt_scale, c->d_dt);
    return hipGetLastError();
}

hipError_t launch_next_dt(sph_ctx *c, bool advance_t) {
    int nb = (int)std::min<int64_t>((c->n + DT_BLOCK - 1) / DT_BLOCK, c->dt_blocks);
    if (nb < 1) nb = 1;
    dt_partial<<<dim3(nb), dim3(DT_BLOCK), 0, c->stream>>>(c->f[SPH_F_VX], c->f[SPH_F_VY], c->f[SPH_F_VZ], c->f[SPH_F_AX],
                                                           c->f[SPH_F_AY], c->f[SPH_F_AZ], c->f[SPH_F_U], c->f[SPH_F_DU],
                                                           c->f[SPH_F_C], c->p.h, c->n, c->dt_part, c->orig, (int32_t)c->n_owned,
                                                           c->variable ? c->f[SPH_F_H] : nullptr);
    dt_final<<<dim3(1), dim3(64), 0, c->stream>>>(c->dt_part, nb, c->p.dt_scale, c->p.dt_max, c->p.dt_min, advance_t ? 1 : 0, c->d_dt);
    return hipGetLastError();
}

}  // namespace sph
