/*
 * ho_voxel.c -- oracle restatement of Hare.Geometry.Voxel_Grid (Voxel_Grid.cs).
 *
 * TEST INFRASTRUCTURE ONLY (see hare_oracle.h).  PARITY UNPINNED.
 * Citations are file:line into /root/reference/.
 */
#include "hare_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>
#include <pthread.h>

struct ho_voxel_grid {
    int32_t ct;                 /* VoxelCtX = VoxelCtY = VoxelCtZ */
    int32_t M;
    double obox_min[3], obox_max[3];
    double box_dims[3], voxel_dims[3];
    double char_step;
    uint32_t **cell_start;      /* [M][ct^3+1]; cell = (x*ct + y)*ct + z == C# [x,y,z] */
    int32_t **cell_items;       /* [M][...] ascending polygon index per cell */
};

static const double EPSILON = 0.001; /* Voxel_Grid.cs:39 */

/* Voxel (x,y,z) padded box: Voxel_Grid.cs:283-285 with Point + Point (Primitives.cs:156-159). */
void ho_voxel_box(const ho_voxel_grid *g, int32_t x, int32_t y, int32_t z, double bmin[3], double bmax[3])
{
    const int32_t idx[3] = {x, y, z};
    for (int a = 0; a < 3; ++a) {
        double vmin = idx[a] * g->voxel_dims[a] - EPSILON;
        double vmax = (idx[a] + 1) * g->voxel_dims[a] + EPSILON;
        bmin[a] = vmin + g->obox_min[a];
        bmax[a] = vmax + g->obox_min[a];
    }
}

/* ctor prologue shared by both constructors: Voxel_Grid.cs:52-90 / :132-161 */
static void grid_bounds(ho_voxel_grid *g, const ho_topology *models, int32_t M)
{
    double MaxPT[3] = {-INFINITY, -INFINITY, -INFINITY};
    double MinPT[3] = {INFINITY, INFINITY, INFINITY};
    for (int32_t m = 0; m < M; ++m)
        for (int a = 0; a < 3; ++a) {
            if ((models[m].max[a] + 0.01) > MaxPT[a]) MaxPT[a] = (models[m].max[a] + EPSILON);
            if ((models[m].min[a] - 0.01) < MinPT[a]) MinPT[a] = (models[m].min[a] - EPSILON);
        }
    for (int a = 0; a < 3; ++a) {
        g->obox_min[a] = MinPT[a] - .1;
        g->obox_max[a] = MaxPT[a] + .1;
        g->box_dims[a] = g->obox_max[a] - g->obox_min[a];
    }
}

static void grid_set_ct(ho_voxel_grid *g, int32_t ct)
{
    g->ct = ct;
    for (int a = 0; a < 3; ++a) g->voxel_dims[a] = g->box_dims[a] / ct;
    const double *vd = g->voxel_dims;
    g->char_step = (vd[0] < vd[1]) ? ((vd[0] < vd[2]) ? vd[0] : vd[2]) : (vd[1] < vd[2] ? vd[1] : vd[2]);
}

/* An allocation that fails during a grid build sets this (thread-local) flag and the build goes on without writing; the
 * builders look at it before they return and hand back NULL instead of a grid (ho_last_error() says why). */
static __thread int vox_failed;
typedef struct ilist { int32_t *v; uint32_t n, cap; } ilist;
static void il_push(ilist *l, int32_t x)
{
    if (vox_failed) return;
    if (l->n == l->cap) {
        const uint32_t ncap = l->cap ? l->cap * 2 : 8;
        if (ho_grow((void **)&l->v, (size_t)ncap * sizeof(int32_t))) { vox_failed = 1; return; }
        l->cap = ncap;
    }
    l->v[l->n++] = x;
}
static void *vox_calloc(size_t n, size_t size)
{
    void *p = calloc(n ? n : 1, size);
    if (!p) {
        ho_set_error("oracle voxel grid: out of memory");
        vox_failed = 1;
    }
    return p;
}

static void lists_to_csr(ilist *lists, size_t ncell, uint32_t **start_out, int32_t **items_out)
{
    *start_out = NULL;
    *items_out = NULL;
    uint32_t *start = (uint32_t *)malloc((ncell + 1) * sizeof(uint32_t));
    size_t tot = 0;
    for (size_t c = 0; c < ncell; ++c) tot += lists[c].n;
    int32_t *items = (int32_t *)malloc((tot ? tot : 1) * sizeof(int32_t));
    if (!start || !items || tot > 0xFFFFFFF0u) {
        ho_set_error(tot > 0xFFFFFFF0u ? "oracle voxel grid: more than 2^32 list entries" : "oracle voxel grid: out of memory");
        vox_failed = 1;
        free(start);
        free(items);
        for (size_t c = 0; c < ncell; ++c) free(lists[c].v);
        return;
    }
    tot = 0;
    for (size_t c = 0; c < ncell; ++c) { start[c] = (uint32_t)tot; tot += lists[c].n; }
    start[ncell] = (uint32_t)tot;
    for (size_t c = 0; c < ncell; ++c) {
        if (lists[c].n) memcpy(items + start[c], lists[c].v, lists[c].n * sizeof(int32_t));
        free(lists[c].v);
    }
    *start_out = start;
    *items_out = items;
}

static int cmp_i32(const void *a, const void *b)
{
    int32_t x = *(const int32_t *)a, y = *(const int32_t *)b;
    return (x > y) - (x < y);
}

/* Voxel_Grid(Topology[], int Domain): Voxel_Grid.cs:48-121, Fill_Voxels :273-304. */
ho_voxel_grid *ho_voxel_build(const ho_topology *models, int32_t M, int32_t domain, int build_mode)
{
    vox_failed = 0;
    ho_voxel_grid *g = (ho_voxel_grid *)vox_calloc(1, sizeof *g);
    if (!g) return NULL;
    g->M = M;
    grid_bounds(g, models, M);
    grid_set_ct(g, domain);
    g->cell_start = (uint32_t **)vox_calloc((size_t)M, sizeof(uint32_t *));
    g->cell_items = (int32_t **)vox_calloc((size_t)M, sizeof(int32_t *));
    if (vox_failed) { ho_voxel_free(g); return NULL; }
    const int32_t ct = domain;
    const size_t ncell = (size_t)ct * ct * ct;

    for (int32_t m = 0; m < M && !vox_failed; ++m) {
        const ho_topology *T = &models[m];
        ilist *lists = (ilist *)vox_calloc(ncell, sizeof(ilist));
        if (!lists) break;
        if (build_mode == 0) {
            /* literal: for every voxel, for every polygon (Voxel_Grid.cs:276-294) */
            for (int32_t x = 0; x < ct; ++x)
                for (int32_t y = 0; y < ct; ++y)
                    for (int32_t z = 0; z < ct; ++z) {
                        double bmin[3], bmax[3];
                        ho_voxel_box(g, x, y, z, bmin, bmax);
                        ilist *l = &lists[((size_t)x * ct + y) * ct + z];
                        for (int32_t i = 0; i < T->P; ++i)
                            if (ho_poly_box_overlap(bmin, bmax, T->verts + (size_t)i * 12, T->nverts[i])) il_push(l, i);
                    }
        } else {
            /* triangle-major: same predicate on a conservative cell range (polygon AABB grown by
             * the 1 mm pad plus one cell each way, so the pad and any rounding of the index
             * estimate are covered), lists sorted ascending afterwards -> identical lists. */
            for (int32_t i = 0; i < T->P; ++i) {
                int32_t lo[3], hi[3];
                for (int a = 0; a < 3; ++a) {
                    double mn = INFINITY, mx = -INFINITY;
                    for (int c = 0; c < T->nverts[i]; ++c) {
                        double v = T->verts[(size_t)i * 12 + 3 * c + a];
                        if (v < mn) mn = v;
                        if (v > mx) mx = v;
                    }
                    double flo = floor((mn - 0.0011 - g->obox_min[a]) / g->voxel_dims[a]) - 1;
                    double fhi = floor((mx + 0.0011 - g->obox_min[a]) / g->voxel_dims[a]) + 1;
                    if (flo < 0) flo = 0;
                    if (fhi > ct - 1) fhi = ct - 1;
                    lo[a] = (int32_t)flo;
                    hi[a] = (int32_t)fhi;
                }
                for (int32_t x = lo[0]; x <= hi[0]; ++x)
                    for (int32_t y = lo[1]; y <= hi[1]; ++y)
                        for (int32_t z = lo[2]; z <= hi[2]; ++z) {
                            double bmin[3], bmax[3];
                            ho_voxel_box(g, x, y, z, bmin, bmax);
                            if (ho_poly_box_overlap(bmin, bmax, T->verts + (size_t)i * 12, T->nverts[i]))
                                il_push(&lists[((size_t)x * ct + y) * ct + z], i);
                        }
            }
            for (size_t c = 0; c < ncell; ++c)
                if (lists[c].n > 1) qsort(lists[c].v, lists[c].n, sizeof(int32_t), cmp_i32);
        }
        lists_to_csr(lists, ncell, &g->cell_start[m], &g->cell_items[m]);
        free(lists);
    }
    if (vox_failed) { ho_voxel_free(g); return NULL; }
    return g;
}

/* Voxel_Grid(Topology[], int MaxDomain, int Avg_polys): Voxel_Grid.cs:128-254. */
ho_voxel_grid *ho_voxel_build_adaptive(const ho_topology *models, int32_t M, int32_t max_domain, int32_t avg_polys)
{
    vox_failed = 0;
    ho_voxel_grid *g = (ho_voxel_grid *)vox_calloc(1, sizeof *g);
    if (!g) return NULL;
    g->M = M;
    grid_bounds(g, models, M);
    g->cell_start = (uint32_t **)vox_calloc((size_t)M, sizeof(uint32_t *));
    g->cell_items = (int32_t **)vox_calloc((size_t)M, sizeof(int32_t *));

    /* level "-1": one voxel holding every polygon (Voxel_Grid.cs:157-166) */
    int32_t ct = 1;
    ilist **lv = (ilist **)vox_calloc((size_t)M, sizeof(ilist *));
    for (int32_t m = 0; m < M && !vox_failed; ++m) {
        lv[m] = (ilist *)vox_calloc(1, sizeof(ilist));
        if (!lv[m]) break;
        for (int32_t j = 0; j < models[m].P; ++j) il_push(&lv[m][0], j);
    }
    if (vox_failed) {
        if (lv)
            for (int32_t m = 0; m < M; ++m)
                if (lv[m]) { free(lv[m][0].v); free(lv[m]); }
        free(lv);
        ho_voxel_free(g);
        return NULL;
    }
    g->ct = 1; /* if max_domain == 0 the C# keeps the 1x1x1 grid with VoxelDims unset; treat ct=1 */
    grid_set_ct(g, 1);

    for (int32_t k = 0; k < max_domain; ++k) {
        int32_t nct = 2 * ct;
        grid_set_ct(g, nct);
        double sum = 0;
        int cnt = 0;
        for (int32_t m = 0; m < M; ++m) {
            const ho_topology *T = &models[m];
            size_t ncell = (size_t)nct * nct * nct;
            ilist *nl = (ilist *)vox_calloc(ncell, sizeof(ilist));
            if (!nl) {            /* the finer level does not fit: give up cleanly (lists of the coarser level are freed below) */
                for (int32_t mm = 0; mm < M; ++mm) {
                    size_t oc = (size_t)(mm < m ? nct : ct);
                    oc = oc * oc * oc;
                    for (size_t c = 0; c < oc; ++c) free(lv[mm][c].v);
                    free(lv[mm]);
                }
                free(lv);
                ho_voxel_free(g);
                return NULL;
            }
            for (int32_t x = 0; x < nct; ++x)
                for (int32_t y = 0; y < nct; ++y)
                    for (int32_t z = 0; z < nct; ++z) {
                        double bmin[3], bmax[3];
                        ho_voxel_box(g, x, y, z, bmin, bmax);
                        int32_t xp = (int32_t)floor((double)x / 2), yp = (int32_t)floor((double)y / 2), zp = (int32_t)floor((double)z / 2);
                        const ilist *par = &lv[m][((size_t)xp * ct + yp) * ct + zp];
                        ilist *l = &nl[((size_t)x * nct + y) * nct + z];
                        for (uint32_t q = 0; q < par->n; ++q) {
                            int32_t i = par->v[q];
                            if (ho_poly_box_overlap(bmin, bmax, T->verts + (size_t)i * 12, T->nverts[i])) il_push(l, i);
                        }
                    }
            size_t ocell = (size_t)ct * ct * ct;
            for (size_t c = 0; c < ocell; ++c) free(lv[m][c].v);
            free(lv[m]);
            lv[m] = nl;
            for (size_t c = 0; c < ncell; ++c)
                if (nl[c].n > 0) { sum += nl[c].n; cnt++; }
        }
        ct = nct;
        if (k > 1 && sum / cnt < avg_polys) break; /* "We are done..." :252 */
    }
    for (int32_t m = 0; m < M; ++m) {
        lists_to_csr(lv[m], (size_t)ct * ct * ct, &g->cell_start[m], &g->cell_items[m]);
        free(lv[m]);
    }
    free(lv);
    if (vox_failed) { ho_voxel_free(g); return NULL; }
    return g;
}

void ho_voxel_free(ho_voxel_grid *g)
{
    if (!g) return;
    if (!g->cell_start || !g->cell_items) {
        free(g->cell_start);
        free(g->cell_items);
        free(g);
        return;
    }
    for (int32_t m = 0; m < g->M; ++m) {
        free(g->cell_start[m]);
        free(g->cell_items[m]);
    }
    free(g->cell_start);
    free(g->cell_items);
    free(g);
}

int32_t ho_voxel_ct(const ho_voxel_grid *g) { return g->ct; }
double ho_voxel_char_step(const ho_voxel_grid *g) { return g->char_step; }
void ho_voxel_geometry(const ho_voxel_grid *g, double omin[3], double omax[3], double vd[3])
{
    for (int a = 0; a < 3; ++a) {
        omin[a] = g->obox_min[a];
        omax[a] = g->obox_max[a];
        vd[a] = g->voxel_dims[a];
    }
}
const uint32_t *ho_voxel_cell_start(const ho_voxel_grid *g, int32_t m) { return g->cell_start[m]; }
const int32_t *ho_voxel_cell_items(const ho_voxel_grid *g, int32_t m) { return g->cell_items[m]; }

static void miss(ho_xevent *out)
{
    /* X_Event(): Hare_Geometry_Primitives.cs:454-462 */
    memset(out, 0, sizeof *out);
    out->poly_id = -1;
}

/* (int)Math.Floor(v) plus the range test of Voxel_Grid.cs:577, done in the double domain so that
 * NaN / out-of-int-range values land on the "outside" side exactly like int.MinValue does in C#. */
static int cell_index(double v, int32_t ct, int32_t *out)
{
    double f = floor(v);
    if (!(f >= 0.0 && f < (double)ct)) return 0;
    *out = (int32_t)f;
    return 1;
}

/* Voxel_Grid.Shoot: Voxel_Grid.cs:561-761; the poly_origin overload :351-552 only adds the
 * `continue` of :477 (pass -1,-1 for the plain overload: no polygon index is negative). */
int ho_voxel_shoot(const ho_voxel_grid *g, const ho_topology *models, ho_ray *R, int32_t top_index,
                   int32_t po1, int32_t po2, int32_t *mailbox, int32_t ray_id, ho_xevent *out, ho_counters *ctr)
{
    const ho_topology *T = &models[top_index];
    const uint32_t *cs = g->cell_start[top_index];
    const int32_t *ci = g->cell_items[top_index];
    const int32_t ct = g->ct;
    const double *omin = g->obox_min, *vd = g->voxel_dims;
    int32_t X = 0, Y = 0, Z = 0;
    double t_start = 0;

    int inside = cell_index((R->x - omin[0]) / vd[0], ct, &X) &
                 cell_index((R->y - omin[1]) / vd[1], ct, &Y) &
                 cell_index((R->z - omin[2]) / vd[2], ct, &Z);
    if (!inside) {
        if (!ho_aabb_intersect_move(g->obox_min, g->obox_max, R, &t_start)) {
            miss(out);
            return 0;
        }
        inside = cell_index((R->x - omin[0] + R->dx * 1E-6) / vd[0], ct, &X) &
                 cell_index((R->y - omin[1] + R->dy * 1E-6) / vd[1], ct, &Y) &
                 cell_index((R->z - omin[2] + R->dz * 1E-6) / vd[2], ct, &Z);
        if (!inside) { /* C#: IndexOutOfRangeException at :593; the build reports a miss */
            miss(out);
            return 0;
        }
    }

    double bmin[3], bmax[3];
    ho_voxel_box(g, X, Y, Z, bmin, bmax);
    int stepX, stepY, stepZ;
    double tMaxX, tMaxY, tMaxZ, tDeltaX, tDeltaY, tDeltaZ;
    if (R->dx < 0) {
        stepX = -1;
        tMaxX = (bmin[0] - R->x) / R->dx;
        tDeltaX = vd[0] / R->dx * stepX;
    } else {
        stepX = 1;
        tMaxX = (bmax[0] - R->x) / R->dx;
        tDeltaX = vd[0] / R->dx * stepX;
    }
    if (R->dy < 0) {
        stepY = -1;
        tMaxY = (bmin[1] - R->y) / R->dy;
        tDeltaY = vd[1] / R->dy * stepY;
    } else {
        stepY = 1;
        tMaxY = (bmax[1] - R->y) / R->dy;
        tDeltaY = vd[1] / R->dy * stepY;
    }
    if (R->dz < 0) {
        stepZ = -1;
        tMaxZ = (bmin[2] - R->z) / R->dz;
        tDeltaZ = vd[2] / R->dz * stepZ;
    } else {
        stepZ = 1;
        tMaxZ = (bmax[2] - R->z) / R->dz;
        tDeltaZ = vd[2] / R->dz * stepZ;
    }

    int have = 0; /* Xpt != null */
    double hx = 0, hy = 0, hz = 0;
    double tmin = DBL_MAX;
    int32_t pid = -1;

    for (;;) {
        size_t cell = ((size_t)X * ct + Y) * ct + Z;
        if (ctr) {
            ctr->cells++;
            ctr->entries += cs[cell + 1] - cs[cell];
        }
        for (uint32_t q = cs[cell]; q < cs[cell + 1]; ++q) {
            int32_t i = ci[q];
            if (i == po1 || i == po2) continue;
            if (mailbox[i] != ray_id) {
                mailbox[i] = ray_id;
                double x, y, z, t;
                if (ctr) ctr->tests++;
                if (ho_poly_intersect_fast(T, i, R, &x, &y, &z, &t) && t > 0.0000000001) {
                    if (t < tmin) {
                        have = 1;
                        hx = x;
                        hy = y;
                        hz = z;
                        tmin = t;
                        pid = i;
                    }
                }
            }
        }

        if (have) {
            ho_voxel_box(g, X, Y, Z, bmin, bmax);
            if (ho_is_point_in_box(bmin, bmax, hx, hy, hz)) {
                out->t = tmin + t_start;
                out->u = 0;
                out->v = 0;
                out->x = hx;
                out->y = hy;
                out->z = hz;
                out->poly_id = pid;
                out->hit = 1;
                return 1;
            }
        }

        if (tMaxX < tMaxY) {
            if (tMaxX < tMaxZ) {
                X += stepX;
                if (X < 0 || X >= ct) { miss(out); return 0; }
                tMaxX = tMaxX + tDeltaX;
            } else {
                Z += stepZ;
                if (Z < 0 || Z >= ct) { miss(out); return 0; }
                tMaxZ = tMaxZ + tDeltaZ;
            }
        } else {
            if (tMaxY < tMaxZ) {
                Y += stepY;
                if (Y < 0 || Y >= ct) { miss(out); return 0; }
                tMaxY = tMaxY + tDeltaY;
            } else {
                Z += stepZ;
                if (Z < 0 || Z >= ct) { miss(out); return 0; }
                tMaxZ = tMaxZ + tDeltaZ;
            }
        }
    }
}

/* ---- batch driver (threads over contiguous ray chunks; private mailbox per thread) ---- */
typedef struct vjob {
    const ho_voxel_grid *g;
    const ho_topology *models;
    int32_t top;
    int64_t lo, hi;
    ho_ray *rays;
    const int32_t *e1, *e2;
    int32_t first_id;
    int keep;
    ho_xevent *out;
    ho_counters ctr;
} vjob;

static void *vworker(void *arg)
{
    vjob *j = (vjob *)arg;
    int32_t P = j->models[j->top].P;
    int32_t *mb = (int32_t *)calloc((size_t)(P ? P : 1), sizeof(int32_t));
    memset(&j->ctr, 0, sizeof j->ctr);
    for (int64_t i = j->lo; i < j->hi; ++i) {
        ho_ray tmp = j->rays[i];
        ho_ray *R = j->keep ? &tmp : &j->rays[i];
        int32_t id = j->first_id + (int32_t)i;
        int h = ho_voxel_shoot(j->g, j->models, R, j->top, j->e1 ? j->e1[i] : -1, j->e2 ? j->e2[i] : -1,
                               mb, id, &j->out[i], &j->ctr);
        j->ctr.rays++;
        j->ctr.hits += (uint64_t)h;
    }
    free(mb);
    return NULL;
}

int ho_voxel_shoot_batch(const ho_voxel_grid *g, const ho_topology *models, int32_t top_index,
                         int64_t n, ho_ray *rays, const int32_t *excl1, const int32_t *excl2,
                         int32_t first_ray_id, int keep_rays, int nthreads, ho_xevent *out, ho_counters *ctr)
{
    if (nthreads < 1) nthreads = 1;
    if (nthreads > 256) nthreads = 256;
    vjob *jobs = (vjob *)calloc((size_t)nthreads, sizeof(vjob));
    pthread_t *th = (pthread_t *)calloc((size_t)nthreads, sizeof(pthread_t));
    for (int k = 0; k < nthreads; ++k) {
        jobs[k].g = g;
        jobs[k].models = models;
        jobs[k].top = top_index;
        jobs[k].lo = n * k / nthreads;
        jobs[k].hi = n * (k + 1) / nthreads;
        jobs[k].rays = rays;
        jobs[k].e1 = excl1;
        jobs[k].e2 = excl2;
        jobs[k].first_id = first_ray_id;
        jobs[k].keep = keep_rays;
        jobs[k].out = out;
        if (nthreads == 1)
            vworker(&jobs[k]);
        else
            pthread_create(&th[k], NULL, vworker, &jobs[k]);
    }
    ho_counters tot;
    memset(&tot, 0, sizeof tot);
    for (int k = 0; k < nthreads; ++k) {
        if (nthreads > 1) pthread_join(th[k], NULL);
        tot.rays += jobs[k].ctr.rays;
        tot.hits += jobs[k].ctr.hits;
        tot.cells += jobs[k].ctr.cells;
        tot.entries += jobs[k].ctr.entries;
        tot.tests += jobs[k].ctr.tests;
    }
    if (ctr) *ctr = tot;
    free(jobs);
    free(th);
    return 0;
}

/* ---- faithful mailbox pool: Poly_Ray_ID[Model.Length, 500][Polygon_Count] zero-initialised
 *      (Voxel_Grid.cs:54-62) + assign_id (:334-342).  Slots are allocated lazily. ---- */
struct ho_voxel_pool {
    const ho_voxel_grid *g;
    const ho_topology *models;
    uint32_t rayno;
    int32_t **slots; /* [M*500] */
};

ho_voxel_pool *ho_voxel_pool_new(const ho_voxel_grid *g, const ho_topology *models)
{
    ho_voxel_pool *p = (ho_voxel_pool *)calloc(1, sizeof *p);
    p->g = g;
    p->models = models;
    p->slots = (int32_t **)calloc((size_t)g->M * 500, sizeof(int32_t *));
    return p;
}

void ho_voxel_pool_free(ho_voxel_pool *p)
{
    if (!p) return;
    for (size_t i = 0; i < (size_t)p->g->M * 500; ++i) free(p->slots[i]);
    free(p->slots);
    free(p);
}

int ho_voxel_pool_shoot(ho_voxel_pool *p, ho_ray *R, int32_t ray_id, int32_t top_index,
                        int32_t po1, int32_t po2, ho_xevent *out)
{
    /* assign_id */
    p->rayno++;
    if (p->rayno == 500) p->rayno = 0;
    uint32_t rayid = p->rayno;
    size_t s = (size_t)top_index * 500 + rayid;
    if (!p->slots[s]) {
        int32_t P = p->models[top_index].P;
        p->slots[s] = (int32_t *)calloc((size_t)(P ? P : 1), sizeof(int32_t));
    }
    return ho_voxel_shoot(p->g, p->models, R, top_index, po1, po2, p->slots[s], ray_id, out, NULL);
}
