// integ_common.hpp -- the leapfrog's per-particle expressions, written ONCE with explicit fused multiply-adds (contraction
// off): integrate.hip's streaming kernels and the epilogue of the whole-tile force pass (tiled.hip, which applies the kick that
// follows a force evaluation on the spot) evaluate the same expressions and give the same bits.  The forms are the ones
// the compiler had chosen for the plain expressions of kick / drift / get_next_timestep ([F]:742-776, 845-850).
#pragma once
#include <hip/hip_runtime.h>

#include <cmath>

namespace sph {

// v + 0.5 a dt, u + 0.5 du dt                                              [F]:749-751, 757
__device__ __forceinline__ double kick_half(double v, double a, double dt) {
#pragma clang fp contract(off)
    return fma(0.5 * a, dt, v);
}
// alpha + dalpha dt 0.5                                                    [F]:758
__device__ __forceinline__ double kick_alpha(double al, double dal, double dt) {
#pragma clang fp contract(off)
    return fma(dal * dt, 0.5, al);
}
// x + v dt                                                                 [F]:769-771
__device__ __forceinline__ double drift_pos(double x, double v, double dt) {
#pragma clang fp contract(off)
    return fma(v, dt, x);
}
// the smallest of the four time-step candidates of one particle            [F]:845-850 (fmin skips the NaN of a 0/0 candidate)
__device__ __forceinline__ double dt_candidates(double vx, double vy, double vz, double ax, double ay, double az, double u, double du,
                                                double cs, double h) {
#pragma clang fp contract(off)
    const double v2 = fma(vz, vz, fma(vx, vx, vy * vy));
    const double a2 = fma(az, az, fma(ax, ax, ay * ay));
    const double c1 = sqrt(v2 / a2);                      // [F]:846
    const double c2 = u / fabs(du);                       // [F]:847
    const double c3 = h / sqrt(v2);                       // [F]:848
    const double c4 = h / fma(1.2, cs, cs);               // [F]:849
    return fmin(fmin(c1, fmin(c2, c3)), c4);
}

}  // namespace sph
