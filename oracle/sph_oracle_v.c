/* sph_oracle_v.c -- TEST INFRASTRUCTURE: CPU restatement of the VARIABLE-h reference path.
 *
 * Plain C, fp64, -ffp-contract=off.  Only tests/, smoke() and bench.py's cpu_baseline may use it.
 * Parity status: PINNED against fixtures dumped from the unmodified reference
 * ("/root/reference/SUMMER_SPH - Variable.f90", cited "[V]") -- tests/test_oracle_v.py.
 *
 *   orcv_leaves        create_tree + build_tree: each particle's 1-particle leaf box   [V]:999-1020,163-267
 *   orcv_lookup_kernel lookup_kernel(r, hi): normalised with hi and REAL(4) pi          [V]:119-141
 *   orcv_density       get_density + density_tree_search (rho, Omega)                   [V]:440-496
 *   orcv_eos           get_pressure_and_sound_speed(bodies, gamma)                      [V]:502-512
 *   orcv_sph_forces    get_SPH + SPH_tree_search, grad-h form, gather restatement       [V]:324-432
 *   orcv_update_h      calc_smoothing                                                   [V]:515-546
 *   orcv_dt_candidate  get_next_timestep (per-particle h)                               [V]:1035-1065
 *   orcv_step          one iteration of simulate's loop body ("sph" variant)            [V]:1120-1152
 *
 * The neighbour rule of [V] is NOT a sphere test: the tree walk reaches the
 * leaf of particle j for a body at x iff, on every axis, |x_k - c_leaf(j),k| < 2 h_j + size_leaf(j)/2
 * (ancestors' tests with their max_len are implied by the leaf's).  The kernel then uses the BODY's
 * h (density) or both (forces), and in the force pass only leaves with number_j < number_body are
 * used, with a symmetric update of both particles.  So
 *   density   rho_i   = sum_{j : reach(i; j)}  m_j W(r_ij, h_i)
 *   forces    pair {a,b}, a > b by particle number, contributes iff reach(a; b)
 * where reach(i; j) is the box test above.  This file builds the reference's octree geometry
 * (bbox-midpoint root, edge = largest extent, strict '>' split, one particle per leaf) to get each
 * particle's leaf box, finds candidates with a uniform grid (edge 2 h_max) and applies the rule.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static const double V_PI = (double)3.1415926535897932f;      /* [V]:7: REAL(4) literal */
static const double V_G = (double)39.47841760435743f;        /* [V]:7 */
static const double V_VISC_EPS = (double)0.01f;              /* [V]:405 */
static const double V_ALPHA_DECAY = (double)0.15f;           /* [V]:346 */
static const double V_DT_MAX = (double)0.1f, V_DT_MIN = (double)0.0001f;   /* [V]:1060,1062 */

/* ---- octree leaf geometry ------------------------------------------------------------------- */
typedef struct { const double *x, *y, *z; double *lc; double *ls; int max_depth; } leaf_ctx;

static void leaf_rec(leaf_ctx *c, int *idx, int n, double cx, double cy, double cz, double size, int depth, int *tmp)
{
    if (n <= 1 || depth == 0) {                    /* [V]:203: size(particles) <= 1 .or. depth == 0 */
        for (int k = 0; k < n; k++) {
            int p = idx[k];
            c->lc[3 * p] = cx; c->lc[3 * p + 1] = cy; c->lc[3 * p + 2] = cz;
            c->ls[p] = (n == 1) ? size : -size;    /* negative: unresolved multi-particle node (never a leaf) */
        }
        return;
    }
    int cnt[8] = {0}, start[8], fill[8];
    for (int k = 0; k < n; k++) {                  /* [V]:228-236: strict '>' on each axis */
        int p = idx[k], ch = 0;
        if (c->x[p] > cx) ch |= 1;
        if (c->y[p] > cy) ch |= 2;
        if (c->z[p] > cz) ch |= 4;
        cnt[ch]++;
    }
    start[0] = 0;
    for (int ch = 1; ch < 8; ch++) start[ch] = start[ch - 1] + cnt[ch - 1];
    memcpy(fill, start, sizeof(fill));
    for (int k = 0; k < n; k++) {
        int p = idx[k], ch = 0;
        if (c->x[p] > cx) ch |= 1;
        if (c->y[p] > cy) ch |= 2;
        if (c->z[p] > cz) ch |= 4;
        tmp[fill[ch]++] = p;
    }
    memcpy(idx, tmp, sizeof(int) * (size_t)n);
    for (int ch = 0; ch < 8; ch++) {
        if (!cnt[ch]) continue;
        double ox = (ch & 1) ? 0.25 * size : -0.25 * size;      /* [V]:216-220 */
        double oy = (ch & 2) ? 0.25 * size : -0.25 * size;
        double oz = (ch & 4) ? 0.25 * size : -0.25 * size;
        leaf_rec(c, idx + start[ch], cnt[ch], cx + ox, cy + oy, cz + oz, size * 0.5, depth - 1, tmp + start[ch]);
    }
}

/* leaf centre (lc[3n]) and edge (ls[n]) of every particle; also root centre/size */
void orcv_leaves(int n, const double *x, const double *y, const double *z, int max_depth, double *lc, double *ls, double *root)
{
    double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int i = 0; i < n; i++) {
        if (x[i] < lo[0]) lo[0] = x[i]; if (x[i] > hi[0]) hi[0] = x[i];
        if (y[i] < lo[1]) lo[1] = y[i]; if (y[i] > hi[1]) hi[1] = y[i];
        if (z[i] < lo[2]) lo[2] = z[i]; if (z[i] > hi[2]) hi[2] = z[i];
    }
    double cx = (hi[0] + lo[0]) / 2.0, cy = (hi[1] + lo[1]) / 2.0, cz = (hi[2] + lo[2]) / 2.0;   /* [V]:1007-1009 */
    double size = hi[0] - lo[0];
    if (hi[1] - lo[1] > size) size = hi[1] - lo[1];
    if (hi[2] - lo[2] > size) size = hi[2] - lo[2];                                             /* [V]:1010-1012 */
    if (root) { root[0] = cx; root[1] = cy; root[2] = cz; root[3] = size; }
    int *idx = (int *)malloc(sizeof(int) * (size_t)(n > 0 ? n : 1)), *tmp = (int *)malloc(sizeof(int) * (size_t)(n > 0 ? n : 1));
    for (int i = 0; i < n; i++) idx[i] = i;
    leaf_ctx c = {x, y, z, lc, ls, max_depth};
    leaf_rec(&c, idx, n, cx, cy, cz, size, max_depth, tmp);
    free(idx); free(tmp);
}

/* ---- kernel ------------------------------------------------------------------------------------ */
static inline void lookup_kernel_v(const double *w, const double *dw, int nq, double r, double hi, double *Wi, double *dWi)
{
    const double dq = 2.0 / nq;
    double qi = r / hi;
    if (qi >= 0.0 && qi <= 2.0) {
        int i = (int)(qi / dq);
        if (i > nq - 1) i = nq - 1;
        double a = (qi - i * dq) / dq;
        *Wi = (1.0 - a) * w[i] + a * w[i + 1];
        *dWi = (1.0 - a) * dw[i] + a * dw[i + 1];
    } else {
        *Wi = 0.0; *dWi = 0.0;
    }
    *Wi = *Wi / (V_PI * (hi * hi * hi));                /* [V]:139 */
    *dWi = *dWi / (V_PI * ((hi * hi) * (hi * hi)));     /* [V]:140 */
}

void orcv_lookup_kernel(const double *w, const double *dw, int nq, int n, const double *r, const double *h, double *W, double *dW)
{
    for (int k = 0; k < n; k++) lookup_kernel_v(w, dw, nq, r[k], h[k], &W[k], &dW[k]);
}

/* ---- grid ---------------------------------------------------------------------------------------- */
typedef struct { int nx, ny, nz; double ox, oy, oz, inv; int *start, *idx; } vgrid;

static void vgrid_build(vgrid *g, int n, const double *x, const double *y, const double *z, double edge)
{
    double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int i = 0; i < n; i++) {
        if (x[i] < lo[0]) lo[0] = x[i]; if (x[i] > hi[0]) hi[0] = x[i];
        if (y[i] < lo[1]) lo[1] = y[i]; if (y[i] > hi[1]) hi[1] = y[i];
        if (z[i] < lo[2]) lo[2] = z[i]; if (z[i] > hi[2]) hi[2] = z[i];
    }
    g->ox = lo[0]; g->oy = lo[1]; g->oz = lo[2]; g->inv = 1.0 / edge;
    g->nx = (int)((hi[0] - lo[0]) * g->inv) + 1; g->ny = (int)((hi[1] - lo[1]) * g->inv) + 1; g->nz = (int)((hi[2] - lo[2]) * g->inv) + 1;
    size_t nc = (size_t)g->nx * g->ny * g->nz;
    g->start = (int *)calloc(nc + 1, sizeof(int));
    g->idx = (int *)malloc(sizeof(int) * (size_t)(n > 0 ? n : 1));
    int *cell = (int *)malloc(sizeof(int) * (size_t)(n > 0 ? n : 1));
    for (int i = 0; i < n; i++) {
        int cx = (int)((x[i] - g->ox) * g->inv), cy = (int)((y[i] - g->oy) * g->inv), cz = (int)((z[i] - g->oz) * g->inv);
        if (cx >= g->nx) cx = g->nx - 1; if (cy >= g->ny) cy = g->ny - 1; if (cz >= g->nz) cz = g->nz - 1;
        cell[i] = (cz * g->ny + cy) * g->nx + cx;
        g->start[cell[i] + 1]++;
    }
    for (size_t c = 0; c < nc; c++) g->start[c + 1] += g->start[c];
    int *fill = (int *)malloc(sizeof(int) * (nc + 1));
    memcpy(fill, g->start, sizeof(int) * (nc + 1));
    for (int i = 0; i < n; i++) g->idx[fill[cell[i]]++] = i;
    free(fill); free(cell);
}
static void vgrid_free(vgrid *g) { free(g->start); free(g->idx); }

static inline int reach(const double *lc, const double *ls, const double *h, int j, double px, double py, double pz)
{
    if (ls[j] < 0.0) return 0;                          /* unresolved node: neither branch of the walk matches */
    const double lim = 2.0 * h[j] + ls[j] / 2.0;        /* [V]:380,479: 2*max_len + size/2 at the leaf */
    return fabs(px - lc[3 * j]) < lim && fabs(py - lc[3 * j + 1]) < lim && fabs(pz - lc[3 * j + 2]) < lim;
}

/* rho_i and Omega_i of ONE body with smoothing length hi (h[] of the others is what the tree holds) */
static void density_one(const vgrid *g, int n, const double *x, const double *y, const double *z, const double *m,
                        const double *h_tree, const double *lc, const double *ls, int nq, const double *w, const double *dw,
                        int i, double hi, double *rho, double *omega)
{
    int R = (int)ceil(2.0 * hi * g->inv);               /* cells to look at on each side */
    if (R < 1) R = 1;
    int cx = (int)((x[i] - g->ox) * g->inv), cy = (int)((y[i] - g->oy) * g->inv), cz = (int)((z[i] - g->oz) * g->inv);
    if (cx >= g->nx) cx = g->nx - 1; if (cy >= g->ny) cy = g->ny - 1; if (cz >= g->nz) cz = g->nz - 1;
    double r0 = 0.0, om = 0.0;
    for (int dz = -R; dz <= R; dz++) for (int dy = -R; dy <= R; dy++) for (int dx = -R; dx <= R; dx++) {
        int ax = cx + dx, ay = cy + dy, az = cz + dz;
        if (ax < 0 || ay < 0 || az < 0 || ax >= g->nx || ay >= g->ny || az >= g->nz) continue;
        int c = (az * g->ny + ay) * g->nx + ax;
        for (int k = g->start[c]; k < g->start[c + 1]; k++) {
            int j = g->idx[k];
            if (!reach(lc, ls, h_tree, j, x[i], y[i], z[i])) continue;
            double n0 = x[i] - x[j], n1 = y[i] - y[j], n2 = z[i] - z[j];      /* [V]:481 */
            double dr = sqrt(n0 * n0 + n1 * n1 + n2 * n2);
            double Wj, dWj;
            lookup_kernel_v(w, dw, nq, dr, hi, &Wj, &dWj);                    /* [V]:486 */
            double W_h = -(dr * dWj - 3 * Wj) / hi;                           /* [V]:487 */
            r0 = r0 + m[j] * Wj;                                              /* [V]:492 */
            om = om + m[j] * W_h;                                             /* [V]:493 */
        }
    }
    *rho = r0;
    *omega = 1.0 + (hi / (3 * r0)) * om;                                      /* [V]:455 */
}

void orcv_density(int n, const double *x, const double *y, const double *z, const double *m, const double *h,
                  const double *lc, const double *ls, int nq, const double *w, const double *dw,
                  double *rho, double *omega, int nthreads)
{
    double hmax = 0.0;
    for (int i = 0; i < n; i++) if (h[i] > hmax) hmax = h[i];
    vgrid g; vgrid_build(&g, n, x, y, z, 2.0 * hmax);
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 128) num_threads(nthreads > 0 ? nthreads : 1)
#endif
    for (int i = 0; i < n; i++) density_one(&g, n, x, y, z, m, h, lc, ls, nq, w, dw, i, h[i], &rho[i], &omega[i]);
    vgrid_free(&g);
}

/* [V]:509-510 */
void orcv_eos(int n, const double *u, const double *rho, double gamma, double *P, double *c)
{
    for (int i = 0; i < n; i++) {
        P[i] = (gamma - 1.0) * u[i] * rho[i];
        c[i] = sqrt(gamma * P[i] / rho[i]);
    }
}

/* gather restatement of [V]:352-432 (+ alpha clean-up [V]:346); a (in/out) holds the gravity terms.
 * number: the reference's particle numbers (NULL: the array index); they decide whose walk counts for a pair, [V]:383 --
 * a caller that holds only a subset of the particles (multi-rank tests) passes the global numbers */
void orcv_sph_forces_num(int n, const double *x, const double *y, const double *z, const double *vx, const double *vy, const double *vz,
                         const double *m, const double *h, const double *rho, const double *omega, const double *P, const double *c,
                         const double *alpha, const double *lc, const double *ls, int nq, const double *w, const double *dw,
                         double *ax, double *ay, double *az, double *du, double *dalpha, const long long *number, int nthreads)
{
    double hmax = 0.0;
    for (int i = 0; i < n; i++) if (h[i] > hmax) hmax = h[i];
    vgrid g; vgrid_build(&g, n, x, y, z, 2.0 * hmax);
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 128) num_threads(nthreads > 0 ? nthreads : 1)
#endif
    for (int i = 0; i < n; i++) {
        int cx = (int)((x[i] - g.ox) * g.inv), cy = (int)((y[i] - g.oy) * g.inv), cz = (int)((z[i] - g.oz) * g.inv);
        if (cx >= g.nx) cx = g.nx - 1; if (cy >= g.ny) cy = g.ny - 1; if (cz >= g.nz) cz = g.nz - 1;
        double a0 = ax[i], a1 = ay[i], a2 = az[i], due = 0.0, dal = 0.0;
        const double pri = P[i] / (omega[i] * rho[i] * rho[i]);                       /* [V]:413 */
        for (int dz = -1; dz <= 1; dz++) for (int dy = -1; dy <= 1; dy++) for (int dx = -1; dx <= 1; dx++) {
            int bx = cx + dx, by = cy + dy, bz = cz + dz;
            if (bx < 0 || by < 0 || bz < 0 || bx >= g.nx || by >= g.ny || bz >= g.nz) continue;
            int cc = (bz * g.ny + by) * g.nx + bx;
            for (int k = g.start[cc]; k < g.start[cc + 1]; k++) {
                int j = g.idx[k];
                if (j == i) continue;
                /* [V]:383: the walk of the higher-numbered body must reach the lower one's leaf */
                const int higher = number ? number[i] > number[j] : i > j;
                int ok = higher ? reach(lc, ls, h, j, x[i], y[i], z[i]) : reach(lc, ls, h, i, x[j], y[j], z[j]);
                if (!ok) continue;
                double n0 = x[i] - x[j], n1 = y[i] - y[j], n2 = z[i] - z[j];          /* [V]:385 */
                double dr = sqrt(n0 * n0 + n1 * n1 + n2 * n2);
                if (dr / h[i] > 2.0 && dr / h[j] > 2.0) continue;                     /* both kernels vanish */
                double v0 = vx[i] - vx[j], v1 = vy[i] - vy[j], v2 = vz[i] - vz[j];    /* [V]:387 */
                double vdotr = v0 * n0 + v1 * n1 + v2 * n2;
                if (vdotr >= 0) vdotr = 0.0;
                n0 = n0 / dr; n1 = n1 / dr; n2 = n2 / dr;                             /* [V]:392 */
                double Wo, dWo, Wn, dWn;
                lookup_kernel_v(w, dw, nq, dr, h[i], &Wo, &dWo);                      /* own h      ([V]:395 / 396) */
                lookup_kernel_v(w, dw, nq, dr, h[j], &Wn, &dWn);                      /* neighbour h */
                double go0 = n0 * dWo, go1 = n1 * dWo, go2 = n2 * dWo;
                double gn0 = n0 * dWn, gn1 = n1 * dWn, gn2 = n2 * dWn;
                double vdotgradW = ((go0 * v0 + go1 * v1 + go2 * v2) + (gn0 * v0 + gn1 * v1 + gn2 * v2)) / 2;   /* [V]:401 */
                double avg_len = (h[i] + h[j]) / 2;                                   /* [V]:402 */
                double vis_nu = (avg_len * vdotr) / (dr * dr + V_VISC_EPS * avg_len * avg_len);   /* [V]:405 */
                double cbar = 0.5 * (c[i] + c[j]);
                double abar = 0.5 * (alpha[i] + alpha[j]);
                double visc = (-abar * cbar * vis_nu + 2 * abar * vis_nu * vis_nu) / (0.5 * (rho[i] + rho[j]));   /* [V]:410 */
                double prj = P[j] / (omega[j] * rho[j] * rho[j]);
                /* [V]:413-414: P_i/(Om_i rho_i^2) dW(h_i) + P_j/(Om_j rho_j^2) dW(h_j) + visc (dW_i + dW_j)/2 */
                double c0 = pri * go0 + prj * gn0 + visc * (gn0 + go0) / 2;
                double c1 = pri * go1 + prj * gn1 + visc * (gn1 + go1) / 2;
                double c2 = pri * go2 + prj * gn2 + visc * (gn2 + go2) / 2;
                a0 = a0 - m[j] * c0; a1 = a1 - m[j] * c1; a2 = a2 - m[j] * c2;        /* [V]:416 */
                due = due + m[j] * vdotgradW * (pri + 0.5 * visc);                    /* [V]:419-421 */
                dal = dal + m[j] * vdotgradW;                                         /* [V]:427 */
            }
        }
        ax[i] = a0; ay[i] = a1; az[i] = a2; du[i] = due;
        double t = dal / rho[i];
        dalpha[i] = (t > 0.0 ? t : 0.0) + V_ALPHA_DECAY * ((0.1 - alpha[i]) * c[i] / h[i]);   /* [V]:346 */
    }
    vgrid_free(&g);
}

void orcv_sph_forces(int n, const double *x, const double *y, const double *z, const double *vx, const double *vy, const double *vz,
                     const double *m, const double *h, const double *rho, const double *omega, const double *P, const double *c,
                     const double *alpha, const double *lc, const double *ls, int nq, const double *w, const double *dw,
                     double *ax, double *ay, double *az, double *du, double *dalpha, int nthreads)
{
    orcv_sph_forces_num(n, x, y, z, vx, vy, vz, m, h, rho, omega, P, c, alpha, lc, ls, nq, w, dw, ax, ay, az, du, dalpha, 0, nthreads);
}

/* calc_smoothing, [V]:515-546.  h_tree = the smoothing lengths the tree holds (those of the last
 * evaluation); h (in/out) are the bodies' live values.  rho/omega are updated as the reference does. */
void orcv_update_h(int n, const double *x, const double *y, const double *z, const double *m, double *h,
                   double *rho, double *omega, const double *lc, const double *ls, int nq, const double *w, const double *dw,
                   double eta, double tol, double max_length, int nthreads)
{
    double *h_tree = (double *)malloc(sizeof(double) * (size_t)(n > 0 ? n : 1));
    memcpy(h_tree, h, sizeof(double) * (size_t)n);
    double hmax = 0.0;
    for (int i = 0; i < n; i++) if (h[i] > hmax) hmax = h[i];
    vgrid g; vgrid_build(&g, n, x, y, z, 2.0 * hmax);
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 128) num_threads(nthreads > 0 ? nthreads : 1)
#endif
    for (int i = 0; i < n; i++) {
        double old_len = h[i];
        double t = eta / h[i];
        double hn = h[i] * (1 + ((m[i] * (t * t * t) / rho[i]) - 1) / (3 * omega[i]));          /* [V]:527 */
        if (hn < max_length && hn > (double)0.01f) {                                           /* [V]:528 */
            while (((hn - old_len) / old_len) > tol && (hn < 10.0)) {                          /* [V]:529 */
                old_len = hn;
                density_one(&g, n, x, y, z, m, h_tree, lc, ls, nq, w, dw, i, hn, &rho[i], &omega[i]);   /* [V]:531-535 */
                t = eta / hn;
                hn = hn * (1 + ((m[i] * (t * t * t)) / rho[i] - 1) / (3 * omega[i]));          /* [V]:538 */
            }
            h[i] = hn;
        } else {
            h[i] = old_len;                                                                    /* [V]:541 */
        }
    }
    vgrid_free(&g);
    free(h_tree);
}

/* [V]:1050-1056 */
double orcv_dt_candidate(int n, const double *vx, const double *vy, const double *vz, const double *ax, const double *ay,
                         const double *az, const double *u, const double *du, const double *c, const double *h, double scale)
{
    double mn = INFINITY;
    for (int i = 0; i < n; i++) {
        double v2 = vx[i] * vx[i] + vy[i] * vy[i] + vz[i] * vz[i];
        double a2 = ax[i] * ax[i] + ay[i] * ay[i] + az[i] * az[i];
        double c1 = sqrt(v2 / a2), c2 = u[i] / fabs(du[i]), c3 = h[i] / sqrt(v2), c4 = h[i] / (c[i] + 1.2 * c[i]);
        if (c1 < mn) mn = c1; if (c2 < mn) mn = c2; if (c3 < mn) mn = c3; if (c4 < mn) mn = c4;
    }
    return mn * scale;
}

double orcv_dt_update(double cand, double dt)
{
    if (cand > 2 * dt && 1.5 * dt < V_DT_MAX) return 1.5 * dt;
    else if (cand < 0.5 * dt && dt * 0.5 > V_DT_MIN) return 0.5 * dt;
    return dt;
}

double orcv_G(void) { return V_G; }
double orcv_pi(void) { return V_PI; }
