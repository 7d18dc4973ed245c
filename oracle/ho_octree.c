/*
 * ho_octree.c -- oracle restatement of the LIVE Hare.Geometry.Octree ("Octree - alt.cs";
 * Octree.cs is entirely commented out).
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

typedef struct onode {
    double bmin[3], bmax[3];
    int32_t first_child;   /* -1: leaf (Children[0] == null) */
    int32_t *polys;
    int32_t npolys, cap;
} onode;

struct ho_octree {
    onode *nodes;
    int32_t n, cap;
    int32_t max_depth, max_polys;
    const ho_topology *model0;
    int64_t live_items;   /* list entries currently held by nodes */
    int failed;           /* an allocation failed or a budget was exceeded: the build unwinds, ho_octree_build returns NULL */
};

static int32_t new_node(ho_octree *o, const double mn[3], const double mx[3])
{
    if (o->failed) return -1;
    if (o->n >= HO_MAX_TREE_NODES) {
        ho_set_error("oracle octree: more than 2^24 nodes (child boxes are padded by an absolute 0.1 m, \"Octree - alt.cs\":99-111: "
                     "below ~0.4 m node size the tree grows 8x per level -- lower maxDepth)");
        o->failed = 1;
        return -1;
    }
    if (o->n == o->cap) {
        const int32_t ncap = o->cap ? o->cap * 2 : 64;
        if (ho_grow((void **)&o->nodes, (size_t)ncap * sizeof(onode))) { o->failed = 1; return -1; }
        o->cap = ncap;
    }
    onode *nd = &o->nodes[o->n];
    memset(nd, 0, sizeof *nd);
    memcpy(nd->bmin, mn, sizeof nd->bmin);
    memcpy(nd->bmax, mx, sizeof nd->bmax);
    nd->first_child = -1;
    return o->n++;
}

static void node_push(ho_octree *o, onode *nd, int32_t p)
{
    if (o->failed) return;
    if (o->live_items >= HO_MAX_TREE_ITEMS) {
        ho_set_error("oracle octree: more than 2^28 polygon-list entries alive (the tree grows 8x per level once nodes are smaller "
                     "than the 0.1 m padding of \"Octree - alt.cs\":99-111 -- lower maxDepth)");
        o->failed = 1;
        return;
    }
    if (nd->npolys == nd->cap) {
        const int32_t ncap = nd->cap ? nd->cap * 2 : 8;
        if (ho_grow((void **)&nd->polys, (size_t)ncap * sizeof(int32_t))) { o->failed = 1; return; }
        nd->cap = ncap;
    }
    nd->polys[nd->npolys++] = p;
    o->live_items++;
}

/* BuildOctree: "Octree - alt.cs":91-138 */
static void build(ho_octree *o, int32_t ni, int depth)
{
    if (o->failed) return;
    if (depth >= o->max_depth || o->nodes[ni].npolys <= o->max_polys) return;

    double nmin[3], nmax[3], center[3];
    memcpy(nmin, o->nodes[ni].bmin, sizeof nmin);
    memcpy(nmax, o->nodes[ni].bmax, sizeof nmax);
    for (int a = 0; a < 3; ++a) center[a] = (nmax[a] + nmin[a]) / 2; /* AABB.Center, AABB_Main.cs:64 */

    int32_t first = -1;
    for (int i = 0; i < 8; ++i) {
        const int bit[3] = {4, 2, 1};
        double mn[3], mx[3];
        for (int a = 0; a < 3; ++a) {
            mn[a] = ((i & bit[a]) == 0 ? nmin[a] : center[a]) - 0.1;
            mx[a] = ((i & bit[a]) == 0 ? center[a] : nmax[a]) + 0.1;
        }
        int32_t c = new_node(o, mn, mx);
        if (c < 0) return;              /* budget / memory: the eight children are not all there, leave the node a leaf */
        if (i == 0) first = c;
    }
    o->nodes[ni].first_child = first;

    const ho_topology *T = o->model0; /* Model[0].Polygon_Vertices(polyId), :123 */
    for (int32_t q = 0; q < o->nodes[ni].npolys; ++q) {
        int32_t pid = o->nodes[ni].polys[q];
        for (int c = 0; c < 8; ++c) {
            onode *ch = &o->nodes[first + c];
            if (ho_poly_box_overlap(ch->bmin, ch->bmax, T->verts + (size_t)pid * 12, T->nverts[pid]))
                node_push(o, ch, pid);
        }
        if (o->failed) return;
        /* polygons that overlap no child are dropped (lostpolys is never used, :116,:129) */
    }
    o->live_items -= o->nodes[ni].npolys;
    free(o->nodes[ni].polys); /* node.Polygons.Clear() */
    o->nodes[ni].polys = NULL;
    o->nodes[ni].npolys = 0;
    o->nodes[ni].cap = 0;

    for (int c = 0; c < 8; ++c) build(o, first + c, depth + 1);
}

/* Octree ctor: "Octree - alt.cs":45-89.  Root per topology; the last topology wins; membership is
 * always tested against Model[0]'s vertices (:123).  Raw vertex bounds are taken over the polygon
 * corners (== Vertices_List after Build_Topology). */
ho_octree *ho_octree_build(const ho_topology *models, int32_t M, int32_t max_depth, int32_t max_polys)
{
    ho_octree *o = (ho_octree *)calloc(1, sizeof *o);
    if (!o) { ho_set_error("oracle: out of memory"); return NULL; }
    o->max_depth = max_depth;
    o->max_polys = max_polys;
    o->model0 = &models[0];
    for (int32_t m = 0; m < M; ++m) {
        const ho_topology *T = &models[m];
        double mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
        for (int32_t p = 0; p < T->P; ++p)
            for (int c = 0; c < T->nverts[p]; ++c)
                for (int a = 0; a < 3; ++a) {
                    double v = T->verts[(size_t)p * 12 + 3 * c + a];
                    if (v < mn[a]) mn[a] = v;
                    if (v > mx[a]) mx[a] = v;
                }
        double maxdim = ho_dotnet_max(mx[0] - mn[0], ho_dotnet_max(mx[1] - mn[1], mx[2] - mn[2]));
        double center[3];
        for (int a = 0; a < 3; ++a) center[a] = mx[a] + mn[a] / 2; /* `max + min / 2` (F8) */
        for (int a = 0; a < 3; ++a) {
            mn[a] = center[a] - maxdim - 1e-1;
            mx[a] = center[a] + maxdim + 1e-1;
        }
        /* a fresh root per topology: drop what an earlier topology built */
        for (int32_t i = 0; i < o->n; ++i) free(o->nodes[i].polys);
        o->n = 0;
        o->live_items = 0;
        int32_t root = new_node(o, mn, mx);
        if (root >= 0) {
            for (int32_t i = 0; i < T->P; ++i) node_push(o, &o->nodes[root], i);
            build(o, root, 0);
        }
        if (o->failed) {
            ho_octree_free(o);
            return NULL;
        }
    }
    return o;
}

void ho_octree_free(ho_octree *o)
{
    if (!o) return;
    for (int32_t i = 0; i < o->n; ++i) free(o->nodes[i].polys);
    free(o->nodes);
    free(o);
}

int32_t ho_octree_node_count(const ho_octree *o) { return o->n; }

int64_t ho_octree_item_total(const ho_octree *o)
{
    int64_t t = 0;
    for (int32_t i = 0; i < o->n; ++i) t += o->nodes[i].npolys;
    return t;
}

void ho_octree_export(const ho_octree *o, double *boxes, int32_t *first_child, int32_t *item_start,
                      int32_t *item_count, int32_t *items)
{
    int32_t pos = 0;
    for (int32_t i = 0; i < o->n; ++i) {
        const onode *nd = &o->nodes[i];
        for (int a = 0; a < 3; ++a) {
            boxes[6 * (size_t)i + a] = nd->bmin[a];
            boxes[6 * (size_t)i + 3 + a] = nd->bmax[a];
        }
        first_child[i] = nd->first_child;
        item_start[i] = pos;
        item_count[i] = nd->npolys;
        if (nd->npolys) memcpy(items + pos, nd->polys, (size_t)nd->npolys * sizeof(int32_t));
        pos += nd->npolys;
    }
}

static void miss(ho_xevent *out)
{
    memset(out, 0, sizeof *out);
    out->poly_id = -1;
}

typedef struct sentry { int32_t node; double tmin, tmax; } sentry;

/* Octree.Shoot: "Octree - alt.cs":159-284; ComputeTraversalOrder :286-306. */
int ho_octree_shoot(const ho_octree *o, const ho_topology *models, const ho_ray *ray, int32_t top_index,
                    int32_t po1, int32_t po2, ho_xevent *out, ho_counters *ctr)
{
    const ho_topology *T = &models[top_index];
    const onode *root = &o->nodes[0];

    double invDx = fabs(ray->dx) > 1e-16 ? 1.0 / ray->dx : 1e16;
    double invDy = fabs(ray->dy) > 1e-16 ? 1.0 / ray->dy : 1e16;
    double invDz = fabs(ray->dz) > 1e-16 ? 1.0 / ray->dz : 1e16;

    double tx0 = (root->bmin[0] - ray->x) * invDx;
    double tx1 = (root->bmax[0] - ray->x) * invDx;
    double ty0 = (root->bmin[1] - ray->y) * invDy;
    double ty1 = (root->bmax[1] - ray->y) * invDy;
    double tz0 = (root->bmin[2] - ray->z) * invDz;
    double tz1 = (root->bmax[2] - ray->z) * invDz;
    if (invDx < 0) { double temp = tx0; tx0 = tx1; tx1 = temp; }
    if (invDy < 0) { double temp = ty0; ty0 = ty1; ty1 = temp; }
    if (invDz < 0) { double temp = tz0; tz0 = tz1; tz1 = temp; }

    double tmin = ho_dotnet_max(ho_dotnet_max(tx0, ty0), tz0);
    double tmax = ho_dotnet_min(ho_dotnet_min(tx1, ty1), tz1);

    if (tmax < tmin || tmax < 0) {
        miss(out);
        return 0;
    }

    /* ComputeTraversalOrder */
    int order[8];
    {
        int xDir = ray->dx >= 0 ? 0 : 1;
        int yDir = ray->dy >= 0 ? 0 : 1;
        int zDir = ray->dz >= 0 ? 0 : 1;
        int i = 0;
        for (int ix = xDir; ix <= 1 && ix >= 0; ix += (ray->dx >= 0 ? 1 : -1))
            for (int iy = yDir; iy <= 1 && iy >= 0; iy += (ray->dy >= 0 ? 1 : -1))
                for (int iz = zDir; iz <= 1 && iz >= 0; iz += (ray->dz >= 0 ? 1 : -1))
                    order[i++] = (ix << 2) | (iy << 1) | iz;
    }

    int scap = ho_initial_stack_cap(8 * (o->max_depth + 2));
    sentry *stack = (sentry *)ho_alloc((size_t)scap * sizeof(sentry));
    if (!stack) { miss(out); return -1; }      /* an ERROR, not a miss: callers must not compare this record */
    int sp = 0;
    stack[sp].node = 0;
    stack[sp].tmin = tmin;
    stack[sp].tmax = tmax;
    sp++;

    int hit = 0;
    double closestT = DBL_MAX;
    ho_xevent best;
    miss(&best);

    while (sp > 0) {
        sentry e = stack[--sp];
        const onode *node = &o->nodes[e.node];
        double nodeTmin = e.tmin, nodeTmax = e.tmax;

        if (nodeTmax < nodeTmin || nodeTmax < 0) continue;
        if (hit && closestT <= nodeTmin) continue;
        if (ctr) ctr->cells++;

        if (node->first_child < 0) {
            if (ctr) ctr->entries += (uint64_t)node->npolys;
            for (int32_t q = 0; q < node->npolys; ++q) {
                int32_t polyId = node->polys[q];
                if (polyId == po1 || polyId == po2) continue;
                double x, y, z, u, v, t;
                if (ctr) ctr->tests++;
                if (ho_poly_intersect_full(T, polyId, ray, &x, &y, &z, &u, &v, &t) && t > 0.0000000001) {
                    if (t < closestT) {
                        closestT = t;
                        best.t = t;
                        best.u = u;
                        best.v = v;
                        best.x = x;
                        best.y = y;
                        best.z = z;
                        best.poly_id = polyId;
                        best.hit = 1;
                        hit = 1;
                        if (closestT <= nodeTmin) {
                            *out = best;
                            free(stack);
                            return 1;
                        }
                    }
                }
            }
        } else {
            for (int k = 0; k < 8; ++k) {
                int32_t ci = node->first_child + order[k];
                const onode *child = &o->nodes[ci];
                double cTx0 = (child->bmin[0] - ray->x) * invDx;
                double cTx1 = (child->bmax[0] - ray->x) * invDx;
                double cTy0 = (child->bmin[1] - ray->y) * invDy;
                double cTy1 = (child->bmax[1] - ray->y) * invDy;
                double cTz0 = (child->bmin[2] - ray->z) * invDz;
                double cTz1 = (child->bmax[2] - ray->z) * invDz;
                if (invDx < 0) { double temp = cTx0; cTx0 = cTx1; cTx1 = temp; }
                if (invDy < 0) { double temp = cTy0; cTy0 = cTy1; cTy1 = temp; }
                if (invDz < 0) { double temp = cTz0; cTz0 = cTz1; cTz1 = temp; }
                double childTmin = ho_dotnet_max(ho_dotnet_max(cTx0, cTy0), cTz0);
                double childTmax = ho_dotnet_min(ho_dotnet_min(cTx1, cTy1), cTz1);
                if (childTmax < childTmin || childTmax < 0 || childTmin > nodeTmax || childTmax < nodeTmin) continue;
                if (sp == scap) {
                    if (ho_grow((void **)&stack, (size_t)scap * 2 * sizeof(sentry))) {     /* cannot continue this ray: an ERROR (-1), never a miss */
                        free(stack);
                        miss(out);
                        return -1;
                    }
                    scap *= 2;
                }
                stack[sp].node = ci;
                stack[sp].tmin = ho_dotnet_max(childTmin, nodeTmin);
                stack[sp].tmax = ho_dotnet_min(childTmax, nodeTmax);
                sp++;
            }
        }
    }
    free(stack);
    if (hit) {
        *out = best;
        return 1;
    }
    miss(out);
    return 0;
}

typedef struct ojob {
    const ho_octree *o;
    const ho_topology *models;
    int32_t top;
    int64_t lo, hi;
    const ho_ray *rays;
    const int32_t *e1, *e2;
    ho_xevent *out;
    ho_counters ctr;
    int failed;                 /* a ray could not be traced (allocation failure): the batch reports an error */
} ojob;

static void *oworker(void *arg)
{
    ojob *j = (ojob *)arg;
    memset(&j->ctr, 0, sizeof j->ctr);
    for (int64_t i = j->lo; i < j->hi; ++i) {
        int h = ho_octree_shoot(j->o, j->models, &j->rays[i], j->top, j->e1 ? j->e1[i] : -1,
                                j->e2 ? j->e2[i] : -1, &j->out[i], &j->ctr);
        if (h < 0) { j->failed = 1; return NULL; }
        j->ctr.rays++;
        j->ctr.hits += (uint64_t)h;
    }
    return NULL;
}

int ho_octree_shoot_batch(const ho_octree *o, const ho_topology *models, int32_t top_index, int64_t n,
                          const ho_ray *rays, const int32_t *excl1, const int32_t *excl2, int nthreads,
                          ho_xevent *out, ho_counters *ctr)
{
    if (nthreads < 1) nthreads = 1;
    if (nthreads > 256) nthreads = 256;
    ojob *jobs = (ojob *)calloc((size_t)nthreads, sizeof(ojob));
    pthread_t *th = (pthread_t *)calloc((size_t)nthreads, sizeof(pthread_t));
    for (int k = 0; k < nthreads; ++k) {
        jobs[k].o = o;
        jobs[k].models = models;
        jobs[k].top = top_index;
        jobs[k].lo = n * k / nthreads;
        jobs[k].hi = n * (k + 1) / nthreads;
        jobs[k].rays = rays;
        jobs[k].e1 = excl1;
        jobs[k].e2 = excl2;
        jobs[k].out = out;
        if (nthreads == 1)
            oworker(&jobs[k]);
        else
            pthread_create(&th[k], NULL, oworker, &jobs[k]);
    }
    ho_counters tot;
    memset(&tot, 0, sizeof tot);
    int failed = 0;
    for (int k = 0; k < nthreads; ++k) {
        if (nthreads > 1) pthread_join(th[k], NULL);
        failed |= jobs[k].failed;
        tot.rays += jobs[k].ctr.rays;
        tot.hits += jobs[k].ctr.hits;
        tot.cells += jobs[k].ctr.cells;
        tot.entries += jobs[k].ctr.entries;
        tot.tests += jobs[k].ctr.tests;
    }
    if (ctr) *ctr = tot;
    free(jobs);
    free(th);
    if (failed) {               /* the worker's message is thread-local to the worker: say it on the caller's thread */
        ho_set_error("oracle octree: out of memory while tracing a ray (traversal stack); the batch's events are not valid");
        return -1;
    }
    return 0;
}
