/* sph_oracle.c -- TEST INFRASTRUCTURE: CPU restatement of the reference's hot path.
 *
 * Plain C, IEEE fp64, compiled with -ffp-contract=off.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may use this file; the product (summersph_amd/csrc +
 * summersph_amd/host) never links or calls it.
 *
 * Parity status: PINNED.  Every function here is checked in tests/test_oracle.py against
 * fixtures dumped from the unmodified reference (tests/golden/, made by
 * tests/golden/make_golden.py through oracle/_ref/ref_driver).
 *
 * What is restated (all citations: /root/reference/SUMMER_SPH.f90, "[F]"):
 *   orc_init_tables      init_kernel_table / init_grav_kernel_table          [F]:55-101
 *   orc_lookup_kernel    lookup_kernel                                       [F]:105-127
 *   orc_density          get_density + density_tree_search                   [F]:398-457
 *   orc_eos              get_pressure_and_sound_speed                        [F]:459-468
 *   orc_sink_gravity     zero_rates + sink_gravforces                        [F]:779-793,559-591
 *   orc_sph_forces       get_SPH + SPH_tree_search (gather form)             [F]:295-395
 *   orc_kick / orc_drift kick / drift                                        [F]:742-776
 *   orc_next_dt          get_next_timestep                                   [F]:831-860
 *   orc_step             one iteration of simulate's loop body               [F]:886-916
 *
 * The reference finds neighbours with an octree walk whose leaf test guarantees exactly the
 * set {j : |x_i - x_j| <= 2h} contributes (W == 0 beyond).  This restatement
 * finds the same set with a uniform cell grid (edge 2h) and evaluates each pair with the
 * reference's expression order; only the ORDER OF SUMMATION over neighbours differs
 * (tree order vs cell order), which is rounding-level.  The reference's scatter form
 * ("number_j < number_i", update both) is restated in the equivalent gather form: each
 * term is bitwise the same under i<->j, again only summation order differs.
 *
 * Literals without a _dp suffix in the reference are REAL(4) constants widened to fp64;
 * they are reproduced via float casts below.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORC_PI_LIT 3.14159265359 /* [F]:125-126 */
static const double ORC_G = (double)39.47841760435743f;   /* [F]:7: no kind suffix -> REAL(4) */
static const double ORC_VISC_EPS = (double)0.01f;          /* [F]:373 */
static const double ORC_ALPHA_DECAY = (double)0.15f;       /* [F]:317 */
static const double ORC_DT_MAX = (double)0.1f;             /* [F]:855 */
static const double ORC_DT_MIN = (double)0.0001f;          /* [F]:857 */

double orc_G(void) { return ORC_G; }

/* ---- kernel tables ------------------------------------------------------------------ */
/* [F]:55-79 and [F]:81-101.  dq = 2.0_dp/nq ([F]:10).  q**2, q**3 are q*q, q*q*q. */
void orc_init_tables(int nq, double *w, double *dw, double *grav)
{
    const double dq = 2.0 / nq;
    for (int i = 0; i <= nq; i++) {
        double q = i * dq;
        if (q >= 0.0 && q <= 1.0) {
            w[i] = 1.0 - 1.5 * (q * q) + 0.75 * (q * q * q);
            dw[i] = -3.0 * q + 2.25 * (q * q);
            if (grav) {
                double q3 = q * q * q, q5 = q3 * q * q, q6 = q5 * q;
                grav[i] = ((40.0 * q3) - (36.0 * q5) + (15.0 * q6)) / 30.0;
            }
        } else if (q > 1.0 && q <= 2.0) {
            double t = 2.0 - q;
            w[i] = 0.25 * (t * t * t);
            dw[i] = -0.75 * (t * t);
            if (grav) {
                double q3 = q * q * q, q4 = q3 * q, q5 = q4 * q, q6 = q5 * q;
                grav[i] = ((80.0 * q3) - (90.0 * q4) + (36.0 * q5) - (5 * q6) - 2) / 30.0;
            }
        } else {
            w[i] = 0.0;
            dw[i] = 0.0;
            if (grav) grav[i] = 1.0;
        }
    }
}

/* [F]:105-127.  S is the normalising length ([F]: the global `smoothing`, == hi there). */
static inline void lookup_kernel(const double *w, const double *dw, int nq, double r, double hi, double S,
                                 double *Wi, double *dWi)
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
        *Wi = 0.0;
        *dWi = 0.0;
    }
    *Wi = *Wi / (ORC_PI_LIT * (S * S * S));
    *dWi = *dWi / (ORC_PI_LIT * (S * S * S * S));
}

void orc_lookup_kernel(const double *w, const double *dw, int nq, int n, const double *r, double h,
                       double *W, double *dW)
{
    for (int k = 0; k < n; k++) lookup_kernel(w, dw, nq, r[k], h, h, &W[k], &dW[k]);
}

/* [F]:129-146 */
void orc_lookup_grav(const double *grav, int nq, int n, const double *r, double h, double *gW)
{
    const double dq = 2.0 / nq;
    for (int k = 0; k < n; k++) {
        double qi = r[k] / h;
        if (qi >= 0.0 && qi <= 2.0) {
            int i = (int)(qi / dq);
            if (i > nq - 1) i = nq - 1;
            double a = (qi - i * dq) / dq;
            gW[k] = (1.0 - a) * grav[i] + a * grav[i + 1];
        } else {
            gW[k] = 1.0;
        }
    }
}

/* ---- uniform cell grid (replaces the octree walk; same neighbour set) ------------------ */
typedef struct {
    int nx, ny, nz;
    double ox, oy, oz, inv;
    int *start;   /* ncell+1 */
    int *idx;     /* n: particle ids grouped by cell, ascending id inside a cell */
} grid_t;

static void grid_build(grid_t *g, int n, const double *x, const double *y, const double *z, double edge)
{
    double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int i = 0; i < n; i++) {
        if (x[i] < lo[0]) lo[0] = x[i]; if (x[i] > hi[0]) hi[0] = x[i];
        if (y[i] < lo[1]) lo[1] = y[i]; if (y[i] > hi[1]) hi[1] = y[i];
        if (z[i] < lo[2]) lo[2] = z[i]; if (z[i] > hi[2]) hi[2] = z[i];
    }
    g->ox = lo[0]; g->oy = lo[1]; g->oz = lo[2]; g->inv = 1.0 / edge;
    g->nx = (int)((hi[0] - lo[0]) * g->inv) + 1;
    g->ny = (int)((hi[1] - lo[1]) * g->inv) + 1;
    g->nz = (int)((hi[2] - lo[2]) * g->inv) + 1;
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
    free(fill);
    free(cell);
}

static void grid_free(grid_t *g) { free(g->start); free(g->idx); }

static inline void cell_of(const grid_t *g, double x, double y, double z, int *cx, int *cy, int *cz)
{
    *cx = (int)((x - g->ox) * g->inv); *cy = (int)((y - g->oy) * g->inv); *cz = (int)((z - g->oz) * g->inv);
    if (*cx >= g->nx) *cx = g->nx - 1; if (*cy >= g->ny) *cy = g->ny - 1; if (*cz >= g->nz) *cz = g->nz - 1;
}

/* ---- density ------------------------------------------------------------------------ */
/* [F]:398-457: rho_i = sum over {j: r_ij <= 2h} (self included) of m_j W(r_ij, h). */
void orc_density(int n, const double *x, const double *y, const double *z, const double *m, double h,
                 int nq, const double *w, const double *dw, double *rho, int nthreads)
{
    grid_t g;
    grid_build(&g, n, x, y, z, 2.0 * h);
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 256) num_threads(nthreads > 0 ? nthreads : 1)
#endif
    for (int i = 0; i < n; i++) {
        int cx, cy, cz;
        cell_of(&g, x[i], y[i], z[i], &cx, &cy, &cz);
        double acc = 0.0;
        for (int dz = -1; dz <= 1; dz++) for (int dy = -1; dy <= 1; dy++) for (int dx = -1; dx <= 1; dx++) {
            int ax = cx + dx, ay = cy + dy, az = cz + dz;
            if (ax < 0 || ay < 0 || az < 0 || ax >= g.nx || ay >= g.ny || az >= g.nz) continue;
            int c = (az * g.ny + ay) * g.nx + ax;
            for (int k = g.start[c]; k < g.start[c + 1]; k++) {
                int j = g.idx[k];
                double n0 = x[i] - x[j], n1 = y[i] - y[j], n2 = z[i] - z[j];   /* [F]:445 */
                double dr = sqrt(n0 * n0 + n1 * n1 + n2 * n2);                  /* [F]:446 */
                double Wj, dWj;
                lookup_kernel(w, dw, nq, dr, h, h, &Wj, &dWj);                  /* [F]:449 */
                acc = acc + m[j] * Wj;                                          /* [F]:454 */
            }
        }
        rho[i] = acc;
    }
    grid_free(&g);
}

/* [F]:459-468, gamma = 1.4 hard coded there */
void orc_eos(int n, const double *u, const double *rho, double *P, double *c)
{
    for (int i = 0; i < n; i++) {
        P[i] = (0.4) * u[i] * rho[i];
        c[i] = sqrt(1.4 * P[i] / rho[i]);
    }
}

/* ---- sink gravity: zero_rates then sink_gravforces ([F]:779-793, 559-591) --------------- */
void orc_sink_gravity(int n, const double *x, const double *y, const double *z, const double *m,
                      int ns, const double *sx, const double *sy, const double *sz, const double *sm,
                      double *ax, double *ay, double *az, double *sax, double *say, double *saz)
{
    for (int j = 0; j < n; j++) { ax[j] = 0.0; ay[j] = 0.0; az[j] = 0.0; }
    for (int i = 0; i < ns; i++) { sax[i] = 0.0; say[i] = 0.0; saz[i] = 0.0; }
    for (int i = 0; i < ns; i++) {
        for (int j = 0; j < n; j++) {
            double v0 = x[j] - sx[i], v1 = y[j] - sy[i], v2 = z[j] - sz[i];
            double dr = sqrt(v0 * v0 + v1 * v1 + v2 * v2);
            double d3 = dr * dr * dr;
            double w0 = ORC_G * v0 / d3, w1 = ORC_G * v1 / d3, w2 = ORC_G * v2 / d3;
            sax[i] = sax[i] + (m[j] * w0); say[i] = say[i] + (m[j] * w1); saz[i] = saz[i] + (m[j] * w2);
            ax[j] = ax[j] - (sm[i] * w0); ay[j] = ay[j] - (sm[i] * w1); az[j] = az[j] - (sm[i] * w2);
        }
    }
    if (ns < 2) return;
    for (int i = 0; i < ns; i++) {
        for (int j = 0; j < i; j++) {
            double v0 = sx[j] - sx[i], v1 = sy[j] - sy[i], v2 = sz[j] - sz[i];
            double dr = sqrt(v0 * v0 + v1 * v1 + v2 * v2);
            double d3 = dr * dr * dr;
            double w0 = ORC_G * v0 / d3, w1 = ORC_G * v1 / d3, w2 = ORC_G * v2 / d3;
            sax[i] = sax[i] + (sm[j] * w0); say[i] = say[i] + (sm[j] * w1); saz[i] = saz[i] + (sm[j] * w2);
            sax[j] = sax[j] - (sm[i] * w0); say[j] = say[j] - (sm[i] * w1); saz[j] = saz[j] - (sm[i] * w2);
        }
    }
}

/* ---- SPH pair forces, gather form of [F]:323-395 + the alpha clean-up [F]:316-318 ------- */
/* a (in/out) already holds the gravity terms, as in find_forces' order ([F]:824-827). */
void orc_sph_forces(int n, const double *x, const double *y, const double *z,
                    const double *vx, const double *vy, const double *vz, const double *m,
                    const double *rho, const double *P, const double *c, const double *alpha, double h,
                    int nq, const double *w, const double *dw,
                    double *ax, double *ay, double *az, double *du, double *dalpha, int nthreads)
{
    grid_t g;
    grid_build(&g, n, x, y, z, 2.0 * h);
    const double eps = ORC_VISC_EPS * h * h;   /* 0.01*smoothing*smoothing, [F]:373 */
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 256) num_threads(nthreads > 0 ? nthreads : 1)
#endif
    for (int i = 0; i < n; i++) {
        int cx, cy, cz;
        cell_of(&g, x[i], y[i], z[i], &cx, &cy, &cz);
        double a0 = ax[i], a1 = ay[i], a2 = az[i], due = 0.0, dal = 0.0;
        const double pri = P[i] / (rho[i] * rho[i]);
        for (int dz = -1; dz <= 1; dz++) for (int dy = -1; dy <= 1; dy++) for (int dx = -1; dx <= 1; dx++) {
            int bx = cx + dx, by = cy + dy, bz = cz + dz;
            if (bx < 0 || by < 0 || bz < 0 || bx >= g.nx || by >= g.ny || bz >= g.nz) continue;
            int cc = (bz * g.ny + by) * g.nx + bx;
            for (int k = g.start[cc]; k < g.start[cc + 1]; k++) {
                int j = g.idx[k];
                if (j == i) continue;                                   /* [F]:354: strict <, so never self */
                double n0 = x[i] - x[j], n1 = y[i] - y[j], n2 = z[i] - z[j];       /* [F]:356 */
                double r2 = n0 * n0 + n1 * n1 + n2 * n2;
                double dr = sqrt(r2);                                             /* [F]:357 */
                if (dr / h > 2.0) continue;                 /* W = dW = 0: every term below is 0 */
                double v0 = vx[i] - vx[j], v1 = vy[i] - vy[j], v2 = vz[i] - vz[j]; /* [F]:358 */
                double vdotr = v0 * n0 + v1 * n1 + v2 * n2;                       /* [F]:359 */
                if (vdotr >= 0) vdotr = 0.0;                                      /* [F]:361 */
                n0 = n0 / dr; n1 = n1 / dr; n2 = n2 / dr;                         /* [F]:363 */
                double Wj, dWm;
                lookup_kernel(w, dw, nq, dr, h, h, &Wj, &dWm);                    /* [F]:366 */
                double g0 = n0 * dWm, g1 = n1 * dWm, g2 = n2 * dWm;               /* [F]:368 */
                double vdotgradW = g0 * v0 + g1 * v1 + g2 * v2;                   /* [F]:370 */
                double vis_nu = (h * vdotr) / (dr * dr + eps);                    /* [F]:373 */
                double cbar = 0.5 * (c[i] + c[j]);                                /* [F]:374 */
                double abar = 0.5 * (alpha[i] + alpha[j]);                        /* [F]:376 */
                double visc = (-abar * cbar * vis_nu + 2 * abar * vis_nu * vis_nu) / (0.5 * (rho[i] + rho[j])); /* [F]:378 */
                double prj = P[j] / (rho[j] * rho[j]);
                double C = (pri + prj + visc);                                    /* [F]:381-382 */
                a0 = a0 - m[j] * (C * g0); a1 = a1 - m[j] * (C * g1); a2 = a2 - m[j] * (C * g2); /* [F]:383 */
                due = due + m[j] * vdotgradW * (pri + 0.5 * visc);                /* [F]:387 */
                dal = dal + m[j] * vdotgradW;                                     /* [F]:390 */
            }
        }
        ax[i] = a0; ay[i] = a1; az[i] = a2; du[i] = due;
        /* [F]:317 */
        double t = dal / rho[i];
        dalpha[i] = (t > 0.0 ? t : 0.0) + ORC_ALPHA_DECAY * ((0.1 - alpha[i]) * c[i] / h);
    }
    grid_free(&g);
}

/* ---- integrator ----------------------------------------------------------------------- */
/* [F]:742-759 */
void orc_kick(int n, double *vx, double *vy, double *vz, double *u, double *alpha,
              const double *ax, const double *ay, const double *az, const double *du, const double *dalpha,
              int ns, double *svx, double *svy, double *svz, const double *sax, const double *say, const double *saz,
              double dt)
{
    for (int i = 0; i < n; i++) {
        vx[i] = vx[i] + 0.5 * ax[i] * dt; vy[i] = vy[i] + 0.5 * ay[i] * dt; vz[i] = vz[i] + 0.5 * az[i] * dt;
        u[i] = u[i] + 0.5 * du[i] * dt;
        alpha[i] = alpha[i] + dalpha[i] * dt * 0.5;
    }
    for (int i = 0; i < ns; i++) {
        svx[i] = svx[i] + 0.5 * sax[i] * dt; svy[i] = svy[i] + 0.5 * say[i] * dt; svz[i] = svz[i] + 0.5 * saz[i] * dt;
    }
}

/* [F]:762-776 */
void orc_drift(int n, double *x, double *y, double *z, const double *vx, const double *vy, const double *vz,
               int ns, double *sx, double *sy, double *sz, const double *svx, const double *svy, const double *svz,
               double dt)
{
    for (int i = 0; i < n; i++) { x[i] = x[i] + vx[i] * dt; y[i] = y[i] + vy[i] * dt; z[i] = z[i] + vz[i] * dt; }
    for (int i = 0; i < ns; i++) { sx[i] = sx[i] + svx[i] * dt; sy[i] = sy[i] + svy[i] * dt; sz[i] = sz[i] + svz[i] * dt; }
}

/* [F]:831-860.  minval over the four candidate arrays; NaN candidates (0/0) are skipped the
 * way a '<' scan skips them. */
double orc_dt_candidate(int n, const double *vx, const double *vy, const double *vz,
                        const double *ax, const double *ay, const double *az,
                        const double *u, const double *du, const double *c, double h)
{
    double mn = INFINITY;
    for (int i = 0; i < n; i++) {
        double v2 = vx[i] * vx[i] + vy[i] * vy[i] + vz[i] * vz[i];
        double a2 = ax[i] * ax[i] + ay[i] * ay[i] + az[i] * az[i];
        double c1 = sqrt(v2 / a2);
        double c2 = u[i] / fabs(du[i]);
        double c3 = h / sqrt(v2);
        double c4 = h / (c[i] + 1.2 * c[i]);
        if (c1 < mn) mn = c1; if (c2 < mn) mn = c2; if (c3 < mn) mn = c3; if (c4 < mn) mn = c4;
    }
    return mn * 0.25;
}

double orc_dt_update(double dt_candidate, double dt)
{
    if (dt_candidate > 2 * dt && 1.5 * dt < ORC_DT_MAX) return 1.5 * dt;
    else if (dt_candidate < 0.5 * dt && dt * 0.5 > ORC_DT_MIN) return 0.5 * dt;
    return dt;
}

/* ---- one force evaluation and one simulate-loop iteration -------------------------------- */
/* state arrays, all length n (gas) / ns (sinks); BH gas self-gravity not included ("sph" variant
 * of the fixtures = [F]:824,826,827). */
typedef struct {
    int n, ns, nq;
    double h;
    double *x, *y, *z, *vx, *vy, *vz, *u, *m, *alpha;
    double *rho, *P, *c, *ax, *ay, *az, *du, *dalpha;
    double *sx, *sy, *sz, *svx, *svy, *svz, *sm, *sax, *say, *saz;
    const double *w, *dw;
} orc_state;

void orc_evaluate(orc_state *s, int nthreads)
{
    orc_density(s->n, s->x, s->y, s->z, s->m, s->h, s->nq, s->w, s->dw, s->rho, nthreads);
    orc_eos(s->n, s->u, s->rho, s->P, s->c);
    orc_sink_gravity(s->n, s->x, s->y, s->z, s->m, s->ns, s->sx, s->sy, s->sz, s->sm,
                     s->ax, s->ay, s->az, s->sax, s->say, s->saz);
    orc_sph_forces(s->n, s->x, s->y, s->z, s->vx, s->vy, s->vz, s->m, s->rho, s->P, s->c, s->alpha, s->h,
                   s->nq, s->w, s->dw, s->ax, s->ay, s->az, s->du, s->dalpha, nthreads);
}

/* [F]:889-916 (without accretion / bounds cull, which the "sph" fixtures also leave out) */
double orc_step(orc_state *s, double dt, int nthreads)
{
    orc_evaluate(s, nthreads);
    orc_kick(s->n, s->vx, s->vy, s->vz, s->u, s->alpha, s->ax, s->ay, s->az, s->du, s->dalpha,
             s->ns, s->svx, s->svy, s->svz, s->sax, s->say, s->saz, dt);
    orc_drift(s->n, s->x, s->y, s->z, s->vx, s->vy, s->vz, s->ns, s->sx, s->sy, s->sz, s->svx, s->svy, s->svz, dt);
    orc_evaluate(s, nthreads);
    orc_kick(s->n, s->vx, s->vy, s->vz, s->u, s->alpha, s->ax, s->ay, s->az, s->du, s->dalpha,
             s->ns, s->svx, s->svy, s->svz, s->sax, s->say, s->saz, dt);
    double cand = orc_dt_candidate(s->n, s->vx, s->vy, s->vz, s->ax, s->ay, s->az, s->u, s->du, s->c, s->h);
    return orc_dt_update(cand, dt);
}

int orc_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
