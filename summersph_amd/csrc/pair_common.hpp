// pair_common.hpp -- device helpers shared by the fixed-h (pairs.hip) and variable-h (varh.hip)
// pair kernels.  gfx950 only.
#pragma once
#include <hip/hip_runtime.h>

#include "sph_internal.hpp"

namespace sph {

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

// Neighbour-list rows read as whole int4 quads (whole-tile kernels) are read exactly once per pass: a streaming
// (non-temporal) load keeps them from evicting the gather records, which ARE re-used, from L1 / L2 (forces_wt 0.485 ->
// 0.454 ms on the bench disc).  NOT for the per-entry dword reads of the other kernels, which touch each 1-KB row in four
// consecutive trips and lose their L1 hits with the hint (forces_kernel 0.535 -> 0.58 ms).
__device__ __forceinline__ int4 load_row(const int4 *p) {
    typedef int v4i __attribute__((ext_vector_type(4)));
    const v4i v = __builtin_nontemporal_load(reinterpret_cast<const v4i *>(p));
    return make_int4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ int load_entry(const int32_t *p) { return *p; }
__device__ __forceinline__ int comp4(const int4 &v, int u) { return u == 0 ? v.x : (u == 1 ? v.y : (u == 2 ? v.z : v.w)); }

// ---- fp64 reciprocal / square root helpers ---------------------------------------------------
// A full IEEE fp64 division costs ~20 VALU instructions on CDNA4 and the reference's pair term has
// nine of them.  The kernels use the hardware seed (v_rcp_f64 / v_rsq_f64) plus Newton / Goldschmidt
// steps instead: results are within ~1 ulp of the correctly rounded quotient, i.e. the same size as
// the summation-order noise that is there anyway (parity tolerance 1e-13, tests/test_parity_gpu.py).
__device__ __forceinline__ double fast_rcp(double x) {
    double r = __builtin_amdgcn_rcp(x);
    double e = fma(-x, r, 1.0);
    r = fma(e, r, r);
    e = fma(-x, r, 1.0);
    r = fma(e, r, r);
    return r;
}

// s = sqrt(x), rs = 1/sqrt(x) for x > 0 (x == 0 gives s = 0 and a finite rs the caller masks)
__device__ __forceinline__ void fast_sqrt_rsqrt(double x, double &s, double &rs) {
    const double xs = fmax(x, 1e-300);
    double y = __builtin_amdgcn_rsq(xs);
    double g = xs * y, h = 0.5 * y;
    double r = fma(-h, g, 0.5);
    g = fma(g, r, g); h = fma(h, r, h);
    r = fma(-h, g, 0.5);
    g = fma(g, r, g); h = fma(h, r, h);
    const double d = fma(-g, g, xs);
    g = fma(d, h, g);
    s = x > 0.0 ? g : 0.0;
    rs = h + h;
}

// lookup_kernel's interpolation (SUMMER_SPH.f90:114-118), table in LDS or global; returns the
// un-normalised value.  k = min(int(q/dq), nq-1), a = (q - k dq)/dq are evaluated as q*(1/dq):
// identical except within an ulp of a table knot, where the (continuous) interpolant changes by O(1e-16).
__device__ __forceinline__ double table_lerp(const double *__restrict__ tab, double qi, double inv_dq, int nq) {
    const double t = qi * inv_dq;
    const int k = min((int)t, nq - 1);
    const double a = t - (double)k;
    return (1.0 - a) * tab[k] + a * tab[k + 1];
}

// both tables at once (same knot, same weight)
__device__ __forceinline__ void table_lerp2(const double *__restrict__ tw, const double *__restrict__ tdw, double qi,
                                            double inv_dq, int nq, double &w, double &dw) {
    const double t = qi * inv_dq;
    const int k = min((int)t, nq - 1);
    const double a = t - (double)k, b = 1.0 - a;
    w = b * tw[k] + a * tw[k + 1];
    dw = b * tdw[k] + a * tdw[k + 1];
}

}  // namespace sph
