/*
 * ho_kdtree.c -- oracle restatement of Hare.Geometry.KDTree (KDTree.cs).
 *
 * TEST INFRASTRUCTURE ONLY (see hare_oracle.h).  PARITY UNPINNED.
 * Citations are file:line into /root/reference/.
 *
 * Note: KDTree.Shoot pushes BOTH children at every interior node
 * (KDTree.cs:355-356), so it visits every leaf; only the visit order depends on the ray.
 */
#include "hare_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>
#include <pthread.h>

typedef struct knode {
    double bmin[3], bmax[3];
    double split;
    int32_t axis;
    int32_t left, right; /* -1,-1: leaf */
    int32_t *polys;
    int32_t npolys;
} knode;

struct ho_kdtree {
    knode *nodes;
    int32_t n, cap;
    int32_t max_depth, max_polys;
    const ho_topology *model0;
    double *centroids; /* Model[0].Polygon_Centroid */
    int64_t live_items;
    int failed;        /* allocation failure / budget: the build unwinds, ho_kdtree_build returns NULL */
};

static int32_t knew(ho_kdtree *k, const double mn[3], const double mx[3])
{
    if (k->failed) return -1;
    if (k->n >= HO_MAX_TREE_NODES) {
        ho_set_error("oracle kd-tree: more than 2^24 nodes -- lower maxDepth");
        k->failed = 1;
        return -1;
    }
    if (k->n == k->cap) {
        const int32_t ncap = k->cap ? k->cap * 2 : 64;
        if (ho_grow((void **)&k->nodes, (size_t)ncap * sizeof(knode))) { k->failed = 1; return -1; }
        k->cap = ncap;
    }
    knode *nd = &k->nodes[k->n];
    memset(nd, 0, sizeof *nd);
    memcpy(nd->bmin, mn, sizeof nd->bmin);
    memcpy(nd->bmax, mx, sizeof nd->bmax);
    nd->left = nd->right = -1;
    return k->n++;
}

typedef struct skey { double key; int32_t pos; int32_t id; } skey;

/* double.CompareTo ordering (NaN sorts first), ties broken by position => the stable order
 * Enumerable.OrderBy guarantees (KDTree.cs:98-101). */
static int cmp_skey(const void *a, const void *b)
{
    const skey *x = (const skey *)a, *y = (const skey *)b;
    if (x->key < y->key) return -1;
    if (x->key > y->key) return 1;
    if (x->key != y->key) { /* at least one NaN */
        int xn = isnan(x->key), yn = isnan(y->key);
        if (xn && !yn) return -1;
        if (!xn && yn) return 1;
    }
    return (x->pos > y->pos) - (x->pos < y->pos);
}

/* BuildKDTree: KDTree.cs:90-139 */
static void kbuild(ho_kdtree *k, int32_t ni, int depth, const double mn[3], const double mx[3])
{
    if (k->failed) return;
    if (depth >= k->max_depth || k->nodes[ni].npolys <= k->max_polys) return;
    const ho_topology *T = k->model0;
    int axis = depth % 3;
    int32_t cnt = k->nodes[ni].npolys;
    if (k->live_items + 2 * (int64_t)cnt > HO_MAX_TREE_ITEMS) {
        ho_set_error("oracle kd-tree: more than 2^28 polygon-list entries alive -- lower maxDepth");
        k->failed = 1;
        return;
    }
    skey *sk = (skey *)malloc((size_t)(cnt ? cnt : 1) * sizeof(skey));
    if (!sk) { ho_set_error("oracle: out of memory"); k->failed = 1; return; }
    for (int32_t q = 0; q < cnt; ++q) {
        int32_t id = k->nodes[ni].polys[q];
        sk[q].key = k->centroids[3 * (size_t)id + axis];
        sk[q].pos = q;
        sk[q].id = id;
    }
    qsort(sk, (size_t)cnt, sizeof(skey), cmp_skey);
    int32_t median = cnt / 2;
    double split = sk[median].key;
    k->nodes[ni].axis = axis;
    k->nodes[ni].split = split;

    double leftMax[3] = {mx[0], mx[1], mx[2]};
    leftMax[axis] = split;
    double rightMin[3] = {mn[0], mn[1], mn[2]};
    rightMin[axis] = split;

    int32_t L = knew(k, mn, leftMax);
    int32_t Rr = knew(k, rightMin, mx);
    if (L < 0 || Rr < 0) { free(sk); return; }
    k->nodes[L].polys = (int32_t *)malloc((size_t)(cnt ? cnt : 1) * sizeof(int32_t));
    k->nodes[Rr].polys = (int32_t *)malloc((size_t)(cnt ? cnt : 1) * sizeof(int32_t));
    if (!k->nodes[L].polys || !k->nodes[Rr].polys) {
        ho_set_error("oracle: out of memory");
        k->failed = 1;
        free(sk);
        return;
    }
    k->nodes[ni].left = L;
    k->nodes[ni].right = Rr;

    for (int32_t q = 0; q < cnt; ++q) {
        int32_t id = sk[q].id;
        int anyle = 0, anygt = 0;
        for (int c = 0; c < T->nverts[id]; ++c) {
            double v = T->verts[(size_t)id * 12 + 3 * c + axis];
            if (v <= split) anyle = 1;
            if (v > split) anygt = 1;
        }
        if (anyle) k->nodes[L].polys[k->nodes[L].npolys++] = id;
        if (anygt) k->nodes[Rr].polys[k->nodes[Rr].npolys++] = id;
    }
    free(sk);
    k->live_items += (int64_t)k->nodes[L].npolys + k->nodes[Rr].npolys - cnt;
    free(k->nodes[ni].polys); /* node.Polygons.Clear() */
    k->nodes[ni].polys = NULL;
    k->nodes[ni].npolys = 0;

    kbuild(k, L, depth + 1, mn, leftMax);
    kbuild(k, Rr, depth + 1, rightMin, mx);
}

/* KDTree ctor: KDTree.cs:51-88.  min/max accumulate over topologies; one root per topology,
 * the last wins; membership/centroids use Model[0] (:99,:125). */
ho_kdtree *ho_kdtree_build(const ho_topology *models, int32_t M, int32_t max_depth, int32_t max_polys)
{
    ho_kdtree *k = (ho_kdtree *)calloc(1, sizeof *k);
    if (!k) { ho_set_error("oracle: out of memory"); return NULL; }
    k->max_depth = max_depth;
    k->max_polys = max_polys;
    k->model0 = &models[0];
    k->centroids = (double *)malloc((size_t)(models[0].P ? models[0].P : 1) * 3 * sizeof(double));
    if (!k->centroids) { ho_set_error("oracle: out of memory"); free(k); return NULL; }
    ho_polygon_centroids(models[0].verts, models[0].nverts, models[0].P, k->centroids);
    double mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int32_t m = 0; m < M; ++m) {
        const ho_topology *T = &models[m];
        for (int32_t p = 0; p < T->P; ++p)
            for (int c = 0; c < T->nverts[p]; ++c)
                for (int a = 0; a < 3; ++a) {
                    double v = T->verts[(size_t)p * 12 + 3 * c + a];
                    if (v < mn[a]) mn[a] = v;
                    if (v > mx[a]) mx[a] = v;
                }
        for (int32_t i = 0; i < k->n; ++i) free(k->nodes[i].polys);
        k->n = 0;
        k->live_items = 0;
        int32_t root = knew(k, mn, mx);
        if (root >= 0) {
            k->nodes[root].polys = (int32_t *)malloc((size_t)(T->P ? T->P : 1) * sizeof(int32_t));
            if (!k->nodes[root].polys) { ho_set_error("oracle: out of memory"); k->failed = 1; }
        }
        if (!k->failed) {
            for (int32_t i = 0; i < T->P; ++i) k->nodes[root].polys[i] = i;
            k->nodes[root].npolys = T->P;
            k->live_items = T->P;
            kbuild(k, root, 0, mn, mx);
        }
        if (k->failed) {
            ho_kdtree_free(k);
            return NULL;
        }
    }
    return k;
}

void ho_kdtree_free(ho_kdtree *k)
{
    if (!k) return;
    for (int32_t i = 0; i < k->n; ++i) free(k->nodes[i].polys);
    free(k->nodes);
    free(k->centroids);
    free(k);
}

int32_t ho_kdtree_node_count(const ho_kdtree *k) { return k->n; }

int64_t ho_kdtree_item_total(const ho_kdtree *k)
{
    int64_t t = 0;
    for (int32_t i = 0; i < k->n; ++i) t += k->nodes[i].npolys;
    return t;
}

void ho_kdtree_export(const ho_kdtree *k, double *boxes, double *split, int32_t *axis, int32_t *left,
                      int32_t *right, int32_t *item_start, int32_t *item_count, int32_t *items)
{
    int32_t pos = 0;
    for (int32_t i = 0; i < k->n; ++i) {
        const knode *nd = &k->nodes[i];
        for (int a = 0; a < 3; ++a) {
            boxes[6 * (size_t)i + a] = nd->bmin[a];
            boxes[6 * (size_t)i + 3 + a] = nd->bmax[a];
        }
        split[i] = nd->split;
        axis[i] = nd->axis;
        left[i] = nd->left;
        right[i] = nd->right;
        item_start[i] = pos;
        item_count[i] = nd->npolys;
        if (nd->npolys) memcpy(items + pos, nd->polys, (size_t)nd->npolys * sizeof(int32_t));
        pos += nd->npolys;
    }
}

/* KDTree.Shoot: KDTree.cs:204-361 */
int ho_kdtree_shoot(const ho_kdtree *k, const ho_topology *models, const ho_ray *ray, int32_t top_index,
                    int32_t po1, int32_t po2, int32_t *mailbox, int32_t ray_id, ho_xevent *out, ho_counters *ctr)
{
    const ho_topology *T = &models[top_index];
    memset(out, 0, sizeof *out);
    out->poly_id = -1;
    int hit = 0;
    double closestT = DBL_MAX;

    int scap = ho_initial_stack_cap(k->max_depth + 8);
    if (scap < 2) scap = 2;
    int32_t *stack = (int32_t *)ho_alloc((size_t)scap * sizeof(int32_t));
    if (!stack) return -1;     /* an ERROR, not a miss (out holds the miss record, not to be compared) */
    int sp = 0;
    stack[sp++] = 0;
    const double o[3] = {ray->x, ray->y, ray->z};
    const double d[3] = {ray->dx, ray->dy, ray->dz};

    while (sp > 0) {
        const knode *cur = &k->nodes[stack[--sp]];
        if (ctr) ctr->cells++;
        if (cur->left < 0 && cur->right < 0) {
            if (ctr) ctr->entries += (uint64_t)cur->npolys;
            for (int32_t q = 0; q < cur->npolys; ++q) {
                int32_t polyId = cur->polys[q];
                if (polyId == po1 || polyId == po2) continue;
                if (mailbox[polyId] == ray_id) continue;
                mailbox[polyId] = ray_id;
                double x, y, z, u, v, t;
                if (ctr) ctr->tests++;
                if (ho_poly_intersect_full(T, polyId, ray, &x, &y, &z, &u, &v, &t) && t > 0.0000000001) {
                    if (t < closestT) {
                        closestT = t;
                        out->t = t;
                        out->u = u;
                        out->v = v;
                        out->x = x;
                        out->y = y;
                        out->z = z;
                        out->poly_id = polyId;
                        out->hit = 1;
                        hit = 1;
                    }
                }
            }
        } else {
            /* the three SplitAxis branches (:249-353) are one pattern with the two other axes
             * checked in ascending axis order: (y,z), (x,z), (x,y). */
            int a = cur->axis, b = (a == 0) ? 1 : 0, c = (a == 2) ? 1 : 2;
            double side = o[a] - cur->split;
            double tSplit = -side / d[a];
            double bSplit = o[b] + tSplit * d[b];
            double cSplit = o[c] + tSplit * d[c];
            int32_t first, second;
            if (bSplit <= cur->bmax[b] && bSplit >= cur->bmin[b] && cSplit <= cur->bmax[c] && cSplit >= cur->bmin[c]) {
                if (side >= 0) { first = cur->right; second = cur->left; }
                else { first = cur->left; second = cur->right; }
            } else {
                if (side >= 0) { first = cur->left; second = cur->right; }
                else { first = cur->right; second = cur->left; }
            }
            if (sp + 2 > scap) {
                if (ho_grow((void **)&stack, (size_t)scap * 2 * sizeof(int32_t))) {       /* cannot continue this ray: an ERROR (-1), never a miss */
                    free(stack);
                    memset(out, 0, sizeof *out);
                    out->poly_id = -1;
                    return -1;
                }
                scap *= 2;
            }
            stack[sp++] = second;
            stack[sp++] = first;
        }
    }
    free(stack);
    return hit;
}

typedef struct kjob {
    const ho_kdtree *k;
    const ho_topology *models;
    int32_t top;
    int64_t lo, hi;
    const ho_ray *rays;
    const int32_t *e1, *e2;
    int32_t first_id;
    ho_xevent *out;
    ho_counters ctr;
    int failed;                 /* a ray could not be traced (allocation failure): the batch reports an error */
} kjob;

static void *kworker(void *arg)
{
    kjob *j = (kjob *)arg;
    int32_t P = j->models[j->top].P;
    int32_t *mb = (int32_t *)calloc((size_t)(P ? P : 1), sizeof(int32_t));
    memset(&j->ctr, 0, sizeof j->ctr);
    if (!mb) { j->failed = 1; return NULL; }
    for (int64_t i = j->lo; i < j->hi; ++i) {
        int h = ho_kdtree_shoot(j->k, j->models, &j->rays[i], j->top, j->e1 ? j->e1[i] : -1,
                                j->e2 ? j->e2[i] : -1, mb, j->first_id + (int32_t)i, &j->out[i], &j->ctr);
        if (h < 0) { j->failed = 1; break; }
        j->ctr.rays++;
        j->ctr.hits += (uint64_t)h;
    }
    free(mb);
    return NULL;
}

int ho_kdtree_shoot_batch(const ho_kdtree *k, const ho_topology *models, int32_t top_index, int64_t n,
                          const ho_ray *rays, const int32_t *excl1, const int32_t *excl2, int32_t first_ray_id,
                          int nthreads, ho_xevent *out, ho_counters *ctr)
{
    if (nthreads < 1) nthreads = 1;
    if (nthreads > 256) nthreads = 256;
    kjob *jobs = (kjob *)calloc((size_t)nthreads, sizeof(kjob));
    pthread_t *th = (pthread_t *)calloc((size_t)nthreads, sizeof(pthread_t));
    for (int q = 0; q < nthreads; ++q) {
        jobs[q].k = k;
        jobs[q].models = models;
        jobs[q].top = top_index;
        jobs[q].lo = n * q / nthreads;
        jobs[q].hi = n * (q + 1) / nthreads;
        jobs[q].rays = rays;
        jobs[q].e1 = excl1;
        jobs[q].e2 = excl2;
        jobs[q].first_id = first_ray_id;
        jobs[q].out = out;
        if (nthreads == 1)
            kworker(&jobs[q]);
        else
            pthread_create(&th[q], NULL, kworker, &jobs[q]);
    }
    ho_counters tot;
    memset(&tot, 0, sizeof tot);
    int failed = 0;
    for (int q = 0; q < nthreads; ++q) {
        if (nthreads > 1) pthread_join(th[q], NULL);
        failed |= jobs[q].failed;
        tot.rays += jobs[q].ctr.rays;
        tot.hits += jobs[q].ctr.hits;
        tot.cells += jobs[q].ctr.cells;
        tot.entries += jobs[q].ctr.entries;
        tot.tests += jobs[q].ctr.tests;
    }
    if (ctr) *ctr = tot;
    free(jobs);
    free(th);
    if (failed) {
        ho_set_error("oracle kd-tree: out of memory while tracing a ray (traversal stack / mailbox); the batch's events are not valid");
        return -1;
    }
    return 0;
}
