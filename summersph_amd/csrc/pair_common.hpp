// pair_common.hpp -- device helpers shared by the fixed-h (pairs.hip) and variable-h (varh.hip)
// pair kernels.  gfx950 only.
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>
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
        // clamped as a double BEFORE the cast: a particle far outside the (stale or trimmed) box, or a NaN that is reported a
        // build late, must not reach an out-of-range float -> int conversion (undefined in C++); fmax drops a NaN -> cell 0
        c[a] = (int)fmin(fmax((p[a] - g.org[a]) * g.inv_edge, 0.0), (double)(g.dim[a] - 1));
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
// (Seeds measured on gfx950, tests/tools/seed: v_rcp_f64 and v_rsq_f64 are good to 2^-24.  ONE third-order step therefore
// reaches 2^-70 -- e = 1 - x r, r (1 + e + e^2) -- in three instructions where two Newton steps take four; likewise
// y (1 + e/2 + 3 e^2/8) with e = 1 - x y^2 for the reciprocal square root, five instructions instead of eight.)
__device__ __forceinline__ double fast_rcp(double x) {
    const double r = __builtin_amdgcn_rcp(x);
    const double e = fma(-x, r, 1.0);
    return fma(r, fma(e, e, e), r);
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

// s = sqrt(x) and rs = 1/sqrt(x) from ONE reciprocal square root refined by one third-order step, s = x rs (x == 0: s = 0 up to
// 1e-150, rs finite; the caller masks it).  ~1.5 ulp in s against the ~0.5 of fast_sqrt_rsqrt, three vector instructions fewer
// per pair visit -- used by the fixed-h pair terms since round 3 (the pair sums' parity margin is 1e-13 against ~1e-15).
__device__ __forceinline__ void rsqrt_sqrt(double x, double &s, double &rs) {
    const double xs = fmax(x, 1e-300);
    const double y0 = __builtin_amdgcn_rsq(xs);
    const double e = fma(-(xs * y0), y0, 1.0);
    const double y = fma(y0, fma(0.375, e, 0.5) * e, y0);
    rs = y;
    s = x * y;
}

// 1/a and 1/b from ONE reciprocal: r = 1/(a b) (fast_rcp), 1/a = b r, 1/b = a r.  a, b > 0 and their
// product far from over- / underflow (squared distances and densities here).
__device__ __forceinline__ void rcp_pair(double a, double b, double &ra, double &rb) {
    const double r = fast_rcp(a * b);
    ra = b * r;
    rb = a * r;
}

// rs = 1/sqrt(x) alone, x > 0 (one third-order step on the hardware seed: ~1 ulp); the gravity walk needs the distance
// itself only inside the softening support
__device__ __forceinline__ double fast_rsqrt(double x) {
    const double y = __builtin_amdgcn_rsq(x);
    const double e = fma(-(x * y), y, 1.0);
    return fma(y, fma(0.375, e, 0.5) * e, y);
}

// lookup_kernel's interpolation (SUMMER_SPH.f90:114-118), table in LDS or global; returns the
// un-normalised value.  k = min(int(q/dq), nq-1), a = (q - k dq)/dq are evaluated as q*(1/dq):
// identical except within an ulp of a table knot, where the (continuous) interpolant changes by O(1e-16).
// The W / dW tables carry ONE ENTRY OF PADDING (TAB_LEN(nq) = nq + 2 values, the last one 0 like the knot at q = 2): the
// reference's clamp k <= nq-1 only matters at q = 2 exactly, where it yields t[nq] = 0; with the padding k = nq, a = 0 gives
// the same 0 without the clamp, and a = fract(t) replaces the int -> double conversion and a subtraction (callers pass
// q <= 2, so k <= nq up to the rounding of 2/dq, which lands on the two zero entries).
// (The pair terms are written with explicit fma() under contract(off): which products the compiler fuses would otherwise
// depend on the kernel a term is inlined into, and the kernel sets -- pairs.hip, the tile variants of tiled.hip -- could
// not be compared bitwise.)
// The lookup in two halves, so that a kernel can put work between the request of the two knots and their use: the knots of
// a table in LDS come back behind whatever the wave asked of the LDS before them.
struct Knots { double t0, t1, a; };
__device__ __forceinline__ Knots table_knots_at(const double *__restrict__ tab, double t) {      // t = q / dq, 0 <= t <= nq (+ an ulp)
    const int k = (int)t;
    return Knots{tab[k], tab[k + 1], __builtin_amdgcn_fract(t)};
}
__device__ __forceinline__ Knots table_knots(const double *__restrict__ tab, double qi, double inv_dq) {
    return table_knots_at(tab, qi * inv_dq);
}
// t for ANY q >= 0, with 1/dq = nq/2 EXACTLY (the kernels pass 0.5 * nq: dq = 2/nq, [F]:58): beyond the support t is nq exactly,
// where both knots are the zeros at the table's end -- a pair beyond 2h (and the far-away sentinel record the tile kernels give
// their idle lanes) then adds an exact zero with no test of q at all
__device__ __forceinline__ double knot_coord(double qi, double inv_dq) { return fmin(qi, 2.0) * inv_dq; }
__device__ __forceinline__ double knots_value(const Knots &kn) {
#pragma clang fp contract(off)
    return fma(kn.a, kn.t1, (1.0 - kn.a) * kn.t0);
}
__device__ __forceinline__ double table_lerp(const double *__restrict__ tab, double qi, double inv_dq, int nq) {
    (void)nq;
    return knots_value(table_knots(tab, qi, inv_dq));
}

// The table values themselves, recomputed: knot k of the W / dW table exactly as host_tables (api.hip, [F]:55-79) fills it --
// q = k dq, the same expressions under contract(off), so bitwise the table's values.  The whole-tile kernels use this when
// the 40-KB table would not leave the tile enough LDS (dense neighbourhoods): ~12 vector instructions per knot instead of an
// LDS read, which is still far cheaper than falling back to the direct gathers.
__device__ __forceinline__ double w_knot(int k, double dq, int nq) {
#pragma clang fp contract(off)
    const double q = k * dq;
    const double t = 2.0 - q;
    const double a = 1.0 - 1.5 * (q * q) + 0.75 * (q * q * q);
    const double b = 0.25 * (t * t * t);
    return k >= nq ? 0.0 : ((q >= 0.0 && q <= 1.0) ? a : ((q > 1.0 && q <= 2.0) ? b : 0.0));       // knot nq (q = 2) and the padding: 0
}
__device__ __forceinline__ double dw_knot(int k, double dq, int nq) {
#pragma clang fp contract(off)
    const double q = k * dq;
    const double t = 2.0 - q;
    const double a = -3.0 * q + 2.25 * (q * q);
    const double b = -0.75 * (t * t);
    return k >= nq ? 0.0 : ((q >= 0.0 && q <= 1.0) ? a : ((q > 1.0 && q <= 2.0) ? b : 0.0));
}
// lookup_kernel's interpolation with a knot function instead of a table in memory
template <class KnotFn>
__device__ __forceinline__ Knots knot_knots_at(KnotFn knot, double t) {
    const int k = (int)t;
    return Knots{knot(k), knot(k + 1), __builtin_amdgcn_fract(t)};
}
template <class KnotFn>
__device__ __forceinline__ Knots knot_knots(KnotFn knot, double qi, double inv_dq) { return knot_knots_at(knot, qi * inv_dq); }
template <class KnotFn>
__device__ __forceinline__ double knot_lerp(KnotFn knot, double qi, double inv_dq, int nq) {
    (void)nq;
    return knots_value(knot_knots(knot, qi, inv_dq));
}

// both tables at once (same knot, same weight)
__device__ __forceinline__ void table_lerp2(const double *__restrict__ tw, const double *__restrict__ tdw, double qi,
                                            double inv_dq, int nq, double &w, double &dw) {
    const double t = qi * inv_dq;
    const int k = (int)t;
    const double a = __builtin_amdgcn_fract(t), b = 1.0 - a;
    w = b * tw[k] + a * tw[k + 1];
    dw = b * tdw[k] + a * tdw[k + 1];
}

// ---- the pair terms, written ONCE (pairs.hip, tiled.hip and varh.hip all call these) -----------------------------------

// force gather record (FREC doubles): A = x y z m | B = vx vy vz rho/2 | C = c/2  alpha/2  P/rho^2  h
// (halved values: 0.5*(a_i + a_j) == a_i/2 + a_j/2 exactly, which saves three multiplies per pair; variable h stores
// P/(Omega rho^2) in C.z, Variable.f90:413, and its h in C.w)
__device__ __forceinline__ void write_frec(double *__restrict__ frec, int64_t i, const double4 &pm, double vx, double vy,
                                           double vz, double rho, double P_over_rho2, double c, double alpha, double h) {
    double4 *fr = reinterpret_cast<double4 *>(frec + (size_t)i * FREC);
    fr[0] = pm;
    fr[1] = make_double4(vx, vy, vz, 0.5 * rho);
    fr[2] = make_double4(0.5 * c, 0.5 * alpha, P_over_rho2, h);
}

// one visit of the density sum, [F]:443-455: acc += m_j w(q_ij) (un-normalised, [F]:125 is applied once at the end)
struct NoPrefetch { __device__ __forceinline__ void operator()() const {} };

// w_of(q): the two knots of the un-normalised W table around q (table_knots on a table in LDS, or knot_knots); next(): what the
// caller wants issued right behind that request -- the tile kernels read the NEXT neighbour's record there, so that the knots
// are at the head of the wave's LDS queue and not behind six record reads
// MASK: the caller's idle lanes carry an arbitrary record and act = false (their mass counts as 0); !MASK: they carry the far-away,
// massless sentinel record, and the visit has no predicate at all (act is ignored) -- the same sums bitwise, both add exact zeros
template <bool MASK = true, class WFn, class Next = NoPrefetch>
__device__ __forceinline__ void density_visit_fn(const double4 &pi, const double4 &pj, bool act, WFn w_of, double inv_h, double &acc,
                                                 Next next = Next()) {
#pragma clang fp contract(off)
    const double n0 = pi.x - pj.x, n1 = pi.y - pj.y, n2 = pi.z - pj.z;     // [F]:445
    double dr, rs;
    rsqrt_sqrt(fma(n2, n2, fma(n1, n1, n0 * n0)), dr, rs);                  // [F]:446
    const double qi = dr * inv_h;                                          // [F]:111
    // no control flow: a lane that does not count ([F]:113: q > 2; idle lanes) adds an exact zero, and consecutive visits
    // can overlap
    const Knots kn = w_of(qi);                                             // [F]:113: beyond 2h the knots are the table's final zeros
    next();
    const double mj = (!MASK || act) ? pj.w : 0.0;
    acc = fma(mj, knots_value(kn), acc);                                   // [F]:114-118,454
}
__device__ __forceinline__ void density_visit(const double4 &pi, const double4 &pj, bool act, const double *__restrict__ lds_w,
                                              double inv_h, double inv_dq, int nq, double &acc) {
    (void)nq;
    density_visit_fn(pi, pj, act, [&](double q) { return table_knots_at(lds_w, knot_coord(q, inv_dq)); }, inv_h, acc);
}

// self term, normalisation, EOS and the force record of particle i ([F]:443-455 visits the particle's own leaf: r = 0;
// [F]:125; get_pressure_and_sound_speed [F]:465-466)
__device__ __forceinline__ void density_epilogue(const PairConst &pc, int64_t i, const double4 &pi, double acc, double w0,
                                                 const double *__restrict__ u, const double *__restrict__ alpha,
                                                 const double *__restrict__ vx, const double *__restrict__ vy,
                                                 const double *__restrict__ vz, double *__restrict__ rho,
                                                 double *__restrict__ P, double *__restrict__ cs, double *__restrict__ frec) {
    acc = fma(pi.w, w0, acc);
    const double rhoi = acc / pc.wnorm;
    const double Pi = pc.gamma_m1 * u[i] * rhoi;
    const double ci = sqrt(pc.gamma * Pi / rhoi);
    rho[i] = rhoi; P[i] = Pi; cs[i] = ci;
    write_frec(frec, i, pi, vx[i], vy[i], vz[i], rhoi, Pi / (rhoi * rhoi), ci, alpha[i], pc.h);   // [F]:381: P/(rho*rho)
}

// SPH sums of one target, un-normalised: every term is linear in dW, so 1/(pi h^4) ([F]:126) is applied once at the end
struct ForceSums { double s0 = 0.0, s1 = 0.0, s2 = 0.0, sdu = 0.0, sdal = 0.0; };

// a neighbour as the force pair term needs it
struct Nbr { double x, y, z, m, vx, vy, vz, rho_h, c_h, al_h, P_r2; };

__device__ __forceinline__ Nbr nbr_of(const double4 &A, const double4 &B, const double4 &C) {
    return Nbr{A.x, A.y, A.z, A.w, B.x, B.y, B.z, B.w, C.x, C.y, C.z};
}

// one visit of the fixed-h force sums, gather form of [F]:356-391.  A, B, C: the target's record; dw_of(q): the two knots of the
// un-normalised dw table around q; next(): issued right behind that request (density_visit_fn).  Written without control flow:
// a lane that does not count (beyond 2h every term is exactly 0; r == 0: coincident points; idle lanes) adds exact
// zeros -- every intermediate is finite (r2 + eps > 0, rho > 0, the table is padded) -- and with no branch between them the
// dependent chains of consecutive visits overlap.  The viscosity chain, which does not need the table, is written between the
// request of the knots and their use.
template <bool MASK = true, class DwFn, class Next = NoPrefetch>
__device__ __forceinline__ void force_visit(const PairConst &pc, double inv_h, const double4 &A, const double4 &B, const double4 &C,
                                            const Nbr &j, bool act, DwFn dw_of, ForceSums &f, Next next = Next()) {
#pragma clang fp contract(off)
    const double n0 = A.x - j.x, n1 = A.y - j.y, n2 = A.z - j.z;                  // [F]:356
    const double r2 = fma(n2, n2, fma(n1, n1, n0 * n0));
    double dr, rs;
    rsqrt_sqrt(r2, dr, rs);                                                       // [F]:357
    const double qi = dr * inv_h;
    const Knots kn = dw_of(qi);               // beyond 2h: the table's final zeros; r == 0: dw(0) = 0 -- g, and with it every term, is 0
    next();
    if constexpr (!std::is_same<Next, NoPrefetch>::value) __builtin_amdgcn_sched_barrier(0);     // both requests leave before the rest
    const bool on = !MASK || act;
    const double v0 = B.x - j.vx, v1 = B.y - j.vy, v2 = B.z - j.vz;               // [F]:358
    const double vdotr = fmin(fma(v2, n2, fma(v1, n1, v0 * n0)), 0.0);            // [F]:359-361
    double inv_r2e, inv_rho;                                                      // 1 / (r^2 + 0.01 h^2), 1 / rho_bar: one reciprocal
    rcp_pair(r2 + pc.visc_eps_h2, B.w + j.rho_h, inv_r2e, inv_rho);
    const double vis_nu = (pc.h * vdotr) * inv_r2e;                               // [F]:373
    const double cbar = C.x + j.c_h;                                              // [F]:374 (halves stored)
    const double abar = C.y + j.al_h;                                             // [F]:376
    const double visc = ((abar * vis_nu) * fma(2.0, vis_nu, -cbar)) * inv_rho;    // [F]:378
    const double Cf = (C.z + j.P_r2) + visc;                                      // [F]:381-382
    const double mj = on ? j.m : 0.0;
    const double mC = mj * Cf;
    const double dWm = knots_value(kn) * rs;                                      // [F]:366; rs: the 1/dr of [F]:363
    const double g0 = n0 * dWm, g1 = n1 * dWm, g2 = n2 * dWm;                     // [F]:363,368
    const double vdotgradW = fma(g2, v2, fma(g1, v1, g0 * v0));                   // [F]:370
    f.s0 = fma(mC, g0, f.s0); f.s1 = fma(mC, g1, f.s1); f.s2 = fma(mC, g2, f.s2); // [F]:383
    const double mv = mj * vdotgradW;
    f.sdu = fma(mv, fma(0.5, visc, C.z), f.sdu);                                  // [F]:387
    f.sdal = f.sdal + mv;                                                         // [F]:390
}

// zero_rates [+ the self-gravity term already in ax..az, [F]:824-825], then the gas side of sink_gravforces, [F]:567-576
__device__ __forceinline__ void sink_gas_accel(const PairConst &pc, const double *__restrict__ sink, const double4 &A, int64_t i,
                                               const double *__restrict__ ax, const double *__restrict__ ay,
                                               const double *__restrict__ az, double &a0, double &a1, double &a2) {
    a0 = pc.grav ? ax[i] : 0.0; a1 = pc.grav ? ay[i] : 0.0; a2 = pc.grav ? az[i] : 0.0;
    for (int s = 0; s < pc.ns; s++) {
        const double v0 = A.x - sink[0 * MAX_SINKS + s], v1 = A.y - sink[1 * MAX_SINKS + s], v2 = A.z - sink[2 * MAX_SINKS + s];
        const double dr = sqrt(v0 * v0 + v1 * v1 + v2 * v2);
        const double d3 = dr * dr * dr;
        const double ms = sink[6 * MAX_SINKS + s];
        a0 = a0 - (ms * (pc.G * v0 / d3)); a1 = a1 - (ms * (pc.G * v1 / d3)); a2 = a2 - (ms * (pc.G * v2 / d3));
    }
}

// ---- rates of particle i from its sums (fixed h): sink gravity, normalisation, the alpha rate of [F]:316-318 -------------
// (rho_i = 2 B.w, c_i = 2 C.x, alpha_i = 2 C.y: exact).  Written per CHANNEL -- 0, 1, 2: the acceleration components, 3: du/dt --
// so that the whole-tile kernel can give the four channels of a target to the four lanes that summed it, each storing its
// own; the gather kernels evaluate the four channels in one thread (the compiler shares what they have in common).  No IEEE
// division or square root (~30 instructions each): the hardware seeds refined to ~1 ulp, as in the pair terms.
struct SinkRows { const double *x, *y, *z, *m; };          // positions and masses of the sinks: MAX_SINKS doubles each, global or LDS
__device__ __forceinline__ SinkRows sink_rows(const double *sink) {
    return SinkRows{sink, sink + MAX_SINKS, sink + 2 * MAX_SINKS, sink + 6 * MAX_SINKS};
}

// acc_in: the channel's start value -- zero_rates [+ the self-gravity term already in ax..az, [F]:824-825] (channel 3: unused)
__device__ __forceinline__ double force_channel(const PairConst &pc, const SinkRows &sk, const double4 &A, const ForceSums &f,
                                                double acc_in, int c) {
#pragma clang fp contract(off)
    const double sum_c = c == 0 ? f.s0 : (c == 1 ? f.s1 : f.s2);
    double a = acc_in;
    for (int s = 0; s < pc.ns; s++) {                                             // the gas side of sink_gravforces, [F]:567-576
        const double v0 = A.x - sk.x[s], v1 = A.y - sk.y[s], v2 = A.z - sk.z[s];
        const double vc = c == 0 ? v0 : (c == 1 ? v1 : v2);
        double dr, rs;
        rsqrt_sqrt(fma(v2, v2, fma(v1, v1, v0 * v0)), dr, rs);
        const double gm = (sk.m[s] * pc.G) * ((rs * rs) * rs);                    // G m_s / dr^3
        a = fma(-gm, vc, a);
    }
    const double acc = a - sum_c * pc.inv_dwnorm;                                 // [F]:383; [F]:126 applied once
    return c == 3 ? f.sdu * pc.inv_dwnorm : acc;                                  // [F]:387
}
// the alpha rate, [F]:316-318,390
__device__ __forceinline__ double alpha_rate(const PairConst &pc, double inv_h, double rho_half, double c_half, double al_half,
                                             const ForceSums &f) {
#pragma clang fp contract(off)
    return fmax((f.sdal * pc.inv_dwnorm) * fast_rcp(2.0 * rho_half), 0.0) +
           pc.alpha_decay * (((pc.alpha_floor - 2.0 * al_half) * (2.0 * c_half)) * inv_h);
}

__device__ __forceinline__ void force_epilogue(const PairConst &pc, const double *__restrict__ sink, int64_t i, double inv_h, const double4 &A,
                                               const double4 &B, const double4 &C, const ForceSums &f, double *__restrict__ ax,
                                               double *__restrict__ ay, double *__restrict__ az, double *__restrict__ du,
                                               double *__restrict__ dalpha) {
    const SinkRows sk = sink_rows(sink);
    const double a0 = pc.grav ? ax[i] : 0.0, a1 = pc.grav ? ay[i] : 0.0, a2 = pc.grav ? az[i] : 0.0;
    ax[i] = force_channel(pc, sk, A, f, a0, 0);
    ay[i] = force_channel(pc, sk, A, f, a1, 1);
    az[i] = force_channel(pc, sk, A, f, a2, 2);
    du[i] = force_channel(pc, sk, A, f, 0.0, 3);
    dalpha[i] = alpha_rate(pc, inv_h, B.w, C.x, C.y, f);
}

}  // namespace sph
