/* sph_oracle_grav.c -- TEST INFRASTRUCTURE: CPU restatement of the reference's Barnes-Hut gas
 * self-gravity, sink accretion and boundary cull.  Plain C, fp64, -ffp-contract=off.
 * Parity status: PINNED -- tests/test_oracle_grav.py checks it against the "full_*" fixtures that
 * were dumped from the unmodified reference (find_forces as is, simulate's accretion + cull).
 *
 * Citations: /root/reference/SUMMER_SPH.f90 "[F]" ("SUMMER_SPH - Variable.f90" "[V]" where it differs)
 *   orcg_tree_build   create_tree + build_tree (octree with mass / centre of mass)   [F]:795-816,149-246
 *   orcg_gravity      particle_gravforces / particle_gravforce_one (theta = 0.5)     [F]:249-290 ([V]:285-311)
 *   orcg_accrete      initiate_sink_accretion + sink2gasdists + pack_sinks           [F]:484-556
 *   orcg_accrete_v    the variable-h variant of sink2gasdists                        [V]:649-676
 *   orcg_cull         check_bounds                                                   [F]:471-482
 *
 * The tree is an explicit node array built by the same recursion as the reference: bbox-midpoint root,
 * edge = largest extent, children by strict '>' on the centre, one particle per leaf, particles kept in
 * the reference's order inside every node (stable partition), so node masses and centres of mass are
 * summed in the reference's order.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static const double GR_G = (double)39.47841760435743f;      /* [F]:7 */

typedef struct {
    double c[3], size, mass, com[3];
    int child[8];          /* -1: none */
    int first, count;      /* particles: order[first .. first+count) */
    int has_children;
} gnode;

typedef struct {
    gnode *nodes; int nnodes, cap;
    int *order;            /* particle ids, grouped per node in the reference's order */
    const double *x, *y, *z, *m;
} gtree;

static int new_node(gtree *t)
{
    if (t->nnodes == t->cap) { t->cap = t->cap * 2 + 64; t->nodes = (gnode *)realloc(t->nodes, sizeof(gnode) * (size_t)t->cap); }
    gnode *nd = &t->nodes[t->nnodes];
    memset(nd, 0, sizeof(*nd));
    for (int k = 0; k < 8; k++) nd->child[k] = -1;
    return t->nnodes++;
}

static void build_rec(gtree *t, int ni, int depth, int *tmp)
{
    /* mass and centre of mass, summed in node order ([F]:165-177) */
    {
        gnode *nd = &t->nodes[ni];
        double M = 0.0, cx = 0.0, cy = 0.0, cz = 0.0;
        for (int k = 0; k < nd->count; k++) {
            int p = t->order[nd->first + k];
            M = M + t->m[p];
            cx = cx + t->m[p] * t->x[p]; cy = cy + t->m[p] * t->y[p]; cz = cz + t->m[p] * t->z[p];
        }
        nd->mass = M;
        if (M > 0.0) { nd->com[0] = cx / M; nd->com[1] = cy / M; nd->com[2] = cz / M; }
        else { nd->com[0] = nd->c[0]; nd->com[1] = nd->c[1]; nd->com[2] = nd->c[2]; }
        if (nd->count <= 1 || depth == 0) return;                                   /* [F]:182 */
    }
    int first = t->nodes[ni].first, count = t->nodes[ni].count;
    double c0 = t->nodes[ni].c[0], c1 = t->nodes[ni].c[1], c2 = t->nodes[ni].c[2], size = t->nodes[ni].size;
    int cnt[8] = {0}, start[8], fill[8];
    for (int k = 0; k < count; k++) {
        int p = t->order[first + k], ch = 0;
        if (t->x[p] > c0) ch |= 1; if (t->y[p] > c1) ch |= 2; if (t->z[p] > c2) ch |= 4;     /* [F]:208-217 */
        cnt[ch]++;
    }
    start[0] = 0;
    for (int ch = 1; ch < 8; ch++) start[ch] = start[ch - 1] + cnt[ch - 1];
    memcpy(fill, start, sizeof(fill));
    for (int k = 0; k < count; k++) {
        int p = t->order[first + k], ch = 0;
        if (t->x[p] > c0) ch |= 1; if (t->y[p] > c1) ch |= 2; if (t->z[p] > c2) ch |= 4;
        tmp[fill[ch]++] = p;
    }
    memcpy(t->order + first, tmp, sizeof(int) * (size_t)count);
    t->nodes[ni].has_children = 1;
    for (int ch = 0; ch < 8; ch++) {
        if (!cnt[ch]) continue;
        int ci = new_node(t);
        gnode *cn = &t->nodes[ci];
        cn->size = size * 0.5;                                                              /* [F]:190 */
        cn->c[0] = c0 + ((ch & 1) ? 0.25 * size : -0.25 * size);                            /* [F]:194-198 */
        cn->c[1] = c1 + ((ch & 2) ? 0.25 * size : -0.25 * size);
        cn->c[2] = c2 + ((ch & 4) ? 0.25 * size : -0.25 * size);
        cn->first = first + start[ch]; cn->count = cnt[ch];
        t->nodes[ni].child[ch] = ci;
        build_rec(t, ci, depth - 1, tmp);
    }
}

gtree *orcg_tree_build(int n, const double *x, const double *y, const double *z, const double *m, int max_depth)
{
    gtree *t = (gtree *)calloc(1, sizeof(gtree));
    t->x = x; t->y = y; t->z = z; t->m = m;
    t->order = (int *)malloc(sizeof(int) * (size_t)(n > 0 ? n : 1));
    for (int i = 0; i < n; i++) t->order[i] = i;
    double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int i = 0; i < n; i++) {
        if (x[i] < lo[0]) lo[0] = x[i]; if (x[i] > hi[0]) hi[0] = x[i];
        if (y[i] < lo[1]) lo[1] = y[i]; if (y[i] > hi[1]) hi[1] = y[i];
        if (z[i] < lo[2]) lo[2] = z[i]; if (z[i] > hi[2]) hi[2] = z[i];
    }
    int r = new_node(t);
    gnode *root = &t->nodes[r];
    for (int a = 0; a < 3; a++) root->c[a] = (hi[a] + lo[a]) / 2.0;                            /* [F]:803-805 */
    root->size = hi[0] - lo[0];
    if (hi[1] - lo[1] > root->size) root->size = hi[1] - lo[1];
    if (hi[2] - lo[2] > root->size) root->size = hi[2] - lo[2];                                /* [F]:806-808 */
    root->first = 0; root->count = n;
    int *tmp = (int *)malloc(sizeof(int) * (size_t)(n > 0 ? n : 1));
    build_rec(t, r, max_depth, tmp);
    free(tmp);
    return t;
}

void orcg_tree_free(gtree *t) { if (t) { free(t->nodes); free(t->order); free(t); } }
int orcg_tree_nodes(const gtree *t) { return t->nnodes; }

static inline double lookup_grav(const double *grav, int nq, double r, double h)
{
    const double dq = 2.0 / nq;
    double qi = r / h;
    if (qi >= 0.0 && qi <= 2.0) {
        int i = (int)(qi / dq);
        if (i > nq - 1) i = nq - 1;
        double a = (qi - i * dq) / dq;
        return (1.0 - a) * grav[i] + a * grav[i + 1];
    }
    return 1.0;
}

/* [F]:264-290.  soft2 = 0.001_dp*smoothing (the MODULE constant 2.5 in both variants); hp = the length the
 * softening table is looked up with ([F]: smoothing, [V]: the particle's own s_length). */
static void grav_one(const gtree *t, int ni, double px, double py, double pz, double hp, double soft2, double theta,
                     const double *grav, int nq, double *a)
{
    const gnode *nd = &t->nodes[ni];
    double d0 = px - nd->com[0], d1 = py - nd->com[1], d2c = pz - nd->com[2];
    double d2 = (d0 * d0 + d1 * d1 + d2c * d2c) + soft2;
    double dist = sqrt(d2);
    if ((nd->size / dist) < theta || !nd->has_children) {
        if (nd->mass > 0.0 && dist > 0.0) {
            double W = lookup_grav(grav, nq, dist, hp);
            double d3 = dist * dist * dist;
            a[0] = a[0] - (GR_G * nd->mass * W * d0 / d3);
            a[1] = a[1] - (GR_G * nd->mass * W * d1 / d3);
            a[2] = a[2] - (GR_G * nd->mass * W * d2c / d3);
        }
    } else {
        for (int k = 0; k < 8; k++)
            if (nd->child[k] >= 0) grav_one(t, nd->child[k], px, py, pz, hp, soft2, theta, grav, nq, a);
    }
}

/* a += gravity of all gas on every particle.  h_var == NULL: fixed h (hp = h_fixed). */
void orcg_gravity(const gtree *t, int n, const double *x, const double *y, const double *z, double h_fixed, const double *h_var,
                  double theta, int nq, const double *grav, double *ax, double *ay, double *az, int nthreads)
{
    const double soft2 = 0.001 * 2.5;       /* 0.001_dp*smoothing with the module constant, [F]:275 / [V]:296 */
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 64) num_threads(nthreads > 0 ? nthreads : 1)
#endif
    for (int i = 0; i < n; i++) {
        double a[3] = {ax[i], ay[i], az[i]};
        grav_one(t, 0, x[i], y[i], z[i], h_var ? h_var[i] : h_fixed, soft2, theta, grav, nq, a);
        ax[i] = a[0]; ay[i] = a[1]; az[i] = a[2];
    }
}

/* ---- sink accretion ---------------------------------------------------------------------------------- */
/* [F]:517-544: walk with radius + size/2 at internal nodes, 2*radius + size/2 at leaves; the leaf's distance
 * is sum(sqrt(centre^2 - sink^2)) over the axes (NaN when negative -> not accreted).  keep[p] = 0 marks
 * accreted particles. */
static void accrete_rec_f(const gtree *t, int ni, const double s[3], double radius, unsigned char *keep)
{
    const gnode *nd = &t->nodes[ni];
    double o[3] = {nd->c[0] - s[0], nd->c[1] - s[1], nd->c[2] - s[2]};
    double lim1 = radius + nd->size / 2.0, lim2 = 2 * radius + nd->size / 2.0;
    if (nd->count > 1 && fabs(o[0]) < lim1 && fabs(o[1]) < lim1 && fabs(o[2]) < lim1 && nd->has_children) {
        for (int k = 0; k < 8; k++) if (nd->child[k] >= 0) accrete_rec_f(t, nd->child[k], s, radius, keep);
    } else if (nd->count == 1 && fabs(o[0]) < lim2 && fabs(o[1]) < lim2 && fabs(o[2]) < lim2) {
        double dr = sqrt(nd->c[0] * nd->c[0] - s[0] * s[0]) + sqrt(nd->c[1] * nd->c[1] - s[1] * s[1])
                  + sqrt(nd->c[2] * nd->c[2] - s[2] * s[2]);                                   /* [F]:537 */
        if (dr < radius) keep[t->order[nd->first]] = 0;
    }
}

/* [V]:649-676: radius + size/2 at both kinds of node, distance = sum |x_p - s| (sqrt of the square) */
static void accrete_rec_v(const gtree *t, int ni, const double s[3], double radius, unsigned char *keep)
{
    const gnode *nd = &t->nodes[ni];
    double o[3] = {nd->c[0] - s[0], nd->c[1] - s[1], nd->c[2] - s[2]};
    double lim = radius + nd->size / 2.0;
    int in = fabs(o[0]) < lim && fabs(o[1]) < lim && fabs(o[2]) < lim;
    if (nd->count > 1 && in && nd->has_children) {
        for (int k = 0; k < 8; k++) if (nd->child[k] >= 0) accrete_rec_v(t, nd->child[k], s, radius, keep);
    } else if (nd->count == 1 && in) {
        int p = t->order[nd->first];
        double dr = sqrt((t->x[p] - s[0]) * (t->x[p] - s[0])) + sqrt((t->y[p] - s[1]) * (t->y[p] - s[1]))
                  + sqrt((t->z[p] - s[2]) * (t->z[p] - s[2]));                                  /* [V]:669 */
        if (dr < radius) keep[p] = 0;
    }
}

/* initiate_sink_accretion, [F]:484-515: per sink (in order) mark, merge mass / position / velocity, then the
 * caller packs.  keep (n bytes) is the OR over sinks.  variant: 0 = [F], 1 = [V]. */
void orcg_accrete(const gtree *t, int n, const double *vx, const double *vy, const double *vz, int ns, double *sx, double *sy,
                  double *sz, double *svx, double *svy, double *svz, double *sm, const double *srad, int variant,
                  unsigned char *keep)
{
    unsigned char *k1 = (unsigned char *)malloc((size_t)(n > 0 ? n : 1));
    memset(keep, 1, (size_t)n);
    for (int i = 0; i < ns; i++) {
        memset(k1, 1, (size_t)n);
        double s[3] = {sx[i], sy[i], sz[i]};
        if (variant) accrete_rec_v(t, 0, s, srad[i], k1); else accrete_rec_f(t, 0, s, srad[i], k1);
        double dm = 0.0, px = 0.0, py = 0.0, pz = 0.0, qx = 0.0, qy = 0.0, qz = 0.0;
        for (int p = 0; p < n; p++) if (!k1[p]) {           /* sum(pack(...)) in particle order, [F]:497-506 */
            dm = dm + t->m[p];
            px = px + t->m[p] * t->x[p]; py = py + t->m[p] * t->y[p]; pz = pz + t->m[p] * t->z[p];
            qx = qx + t->m[p] * vx[p]; qy = qy + t->m[p] * vy[p]; qz = qz + t->m[p] * vz[p];
            keep[p] = 0;
        }
        double new_mass = sm[i] + dm;
        sx[i] = (sm[i] * sx[i] + px) / new_mass; sy[i] = (sm[i] * sy[i] + py) / new_mass; sz[i] = (sm[i] * sz[i] + pz) / new_mass;
        svx[i] = (sm[i] * svx[i] + qx) / new_mass; svy[i] = (sm[i] * svy[i] + qy) / new_mass; svz[i] = (sm[i] * svz[i] + qz) / new_mass;
        sm[i] = sm[i] + dm;
    }
    free(k1);
}

/* check_bounds, [F]:471-482: keep[p] &= all(|x| <= bound) */
void orcg_cull(int n, const double *x, const double *y, const double *z, double bound, unsigned char *keep)
{
    for (int p = 0; p < n; p++)
        if (!(fabs(x[p]) <= bound && fabs(y[p]) <= bound && fabs(z[p]) <= bound)) keep[p] = 0;
}
