/*
 * hare_oracle.h -- CPU restatement of Hare's ray-cast hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under hare_amd/ (the product) may include,
 * link, call or execute anything under oracle/.  Only tests/, the smoke check in
 * __graft_entry__.py and bench.py's cpu_baseline leg use it, and only as the
 * checker / the timed CPU baseline -- never as the thing shipped.
 *
 * PARITY UNPINNED: the reference (PachydermAcoustic/Hare, C#) ships no tests,
 * golden vectors or fixtures for this path, and no C#/.NET toolchain exists in
 * the build container, so the reference itself cannot be run.  This oracle is a
 * line-by-line restatement of the C# text (IEEE-754 binary64, no contraction,
 * reference operand order), pinned only by closed-form known answers
 * (tests/test_oracle_kat.py) and by its own brute-force cross-checks.
 *
 * All reference citations are file:line into /root/reference/.
 */
#ifndef HARE_ORACLE_H
#define HARE_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* X_Event wire record (Hare_Geometry_Primitives.cs:435-481). 56 bytes. */
typedef struct ho_xevent {
    double t, u, v;
    double x, y, z;      /* X_Point; 0,0,0 on a miss (X_Point is null in C#) */
    int32_t poly_id;     /* -1 on a miss (Primitives.cs:461) */
    int32_t hit;         /* 0 / 1 */
} ho_xevent;

/* Ray wire record (Hare_Geometry_Primitives.cs:393-429): origin, direction. */
typedef struct ho_ray {
    double x, y, z, dx, dy, dz;
} ho_ray;

/* Exact per-batch work counters of the REFERENCE algorithm (mailbox applied,
 * exclusions removed) -- these feed the algorithmic-bytes roofline formula
 * B_ray = 104 + 8*C + 4*L + 96*T. */
typedef struct ho_counters {
    uint64_t rays, hits;
    uint64_t cells;      /* C: grid cells (or tree nodes) visited   */
    uint64_t entries;    /* L: candidate-list entries scanned       */
    uint64_t tests;      /* T: Topology.intersect calls performed   */
} ho_counters;

/* A flattened Hare.Geometry.Topology: what a host reads back from the managed
 * object (Hare_Geometry_Topology.cs:418-424, :539, :50/:58). */
typedef struct ho_topology {
    int32_t P;                /* Polygon_Count                                   */
    const double *verts;      /* P x 4 x 3; corner 3 ignored when nverts[p]==3   */
    const int32_t *nverts;    /* P; 3 or 4                                       */
    const double *normals;    /* P x 3; Polygon.Normal (unit)                    */
    double min[3], max[3];    /* Topology.Min / Topology.Max                     */
} ho_topology;

/* ---------- primitives ---------- */
double ho_dot(double ax, double ay, double az, double bx, double by, double bz);
void   ho_cross(double ax, double ay, double az, double bx, double by, double bz, double out[3]);
double ho_dotnet_max(double a, double b);
double ho_dotnet_min(double a, double b);
double ho_dotnet_round(double x, int digits);

/* Polygon ctor normal + Finish_Topology bounds (host-side helpers). */
void ho_polygon_normals(const double *verts, const int32_t *nverts, int32_t P, double *normals_out);
void ho_finish_topology_bounds(const double *verts, const int32_t *nverts, int32_t P, double min_out[3], double max_out[3]);
void ho_polygon_centroids(const double *verts, const int32_t *nverts, int32_t P, double *centroids_out);

/* Topology(Point[][]) ingest: Math.Round(x,15) + Hash2 dedupe.  Writes the corner
 * coordinates a host would read back from the Topology.  Returns vertex count. */
int32_t ho_build_topology(const double *verts_in, const int32_t *nverts, int32_t P, double *verts_out);

/* Polygon tests: returns 1 on hit.  fast = Voxel_Grid path (no u,v). */
int ho_poly_intersect_fast(const ho_topology *T, int32_t i, const ho_ray *R, double *x, double *y, double *z, double *t);
int ho_poly_intersect_full(const ho_topology *T, int32_t i, const ho_ray *R, double *x, double *y, double *z, double *u, double *v, double *t);

/* AABB */
int ho_aabb_intersect_move(const double bmin[3], const double bmax[3], ho_ray *R, double *tmin);
int ho_is_point_in_box(const double bmin[3], const double bmax[3], double x, double y, double z);
int ho_poly_box_overlap(const double bmin[3], const double bmax[3], const double *poly_verts, int32_t nv);

/* ---------- Voxel_Grid ---------- */
typedef struct ho_voxel_grid ho_voxel_grid;

/* build_mode: 0 = literal reference loop (every voxel x every polygon), 1 = triangle-major
 * (conservative cell range + the same exact predicate + ascending lists).  Identical output. */
ho_voxel_grid *ho_voxel_build(const ho_topology *models, int32_t M, int32_t domain, int build_mode);
ho_voxel_grid *ho_voxel_build_adaptive(const ho_topology *models, int32_t M, int32_t max_domain, int32_t avg_polys);
void ho_voxel_free(ho_voxel_grid *g);
int32_t ho_voxel_ct(const ho_voxel_grid *g);
double  ho_voxel_char_step(const ho_voxel_grid *g);
void    ho_voxel_geometry(const ho_voxel_grid *g, double obox_min[3], double obox_max[3], double voxel_dims[3]);
/* CSR view of Voxel_Inv[x,y,z,m]; cell = (x*ct + y)*ct + z. */
const uint32_t *ho_voxel_cell_start(const ho_voxel_grid *g, int32_t m);
const int32_t  *ho_voxel_cell_items(const ho_voxel_grid *g, int32_t m);
void ho_voxel_box(const ho_voxel_grid *g, int32_t x, int32_t y, int32_t z, double bmin[3], double bmax[3]);

/* One Shoot (Voxel_Grid.cs:561-761 / :351-552).  R is mutated when the origin is
 * outside the grid (F11).  mailbox: P ints (Poly_Ray_ID[top,rayid]); ray_id = R.Ray_ID.
 * Returns hit. */
int ho_voxel_shoot(const ho_voxel_grid *g, const ho_topology *models, ho_ray *R, int32_t top_index,
                   int32_t poly_origin1, int32_t poly_origin2,
                   int32_t *mailbox, int32_t ray_id, ho_xevent *out, ho_counters *ctr);

/* Batch driver: Ray_ID = first_ray_id + i, one private zeroed mailbox per thread
 * (equivalent to the reference's locked 500-slot pool for unique non-zero ids).
 * rays are mutated like the reference does unless keep_rays != 0. */
int ho_voxel_shoot_batch(const ho_voxel_grid *g, const ho_topology *models, int32_t top_index,
                         int64_t n, ho_ray *rays, const int32_t *excl1, const int32_t *excl2,
                         int32_t first_ray_id, int keep_rays, int nthreads,
                         ho_xevent *out, ho_counters *ctr);

/* Faithful emulation of the locked mailbox pool (assign_id, Voxel_Grid.cs:334-342),
 * single-threaded; used by the quirk tests (Ray_ID == 0, slot collisions). */
typedef struct ho_voxel_pool ho_voxel_pool;
ho_voxel_pool *ho_voxel_pool_new(const ho_voxel_grid *g, const ho_topology *models);
void ho_voxel_pool_free(ho_voxel_pool *p);
int ho_voxel_pool_shoot(ho_voxel_pool *p, ho_ray *R, int32_t ray_id, int32_t top_index,
                        int32_t poly_origin1, int32_t poly_origin2, ho_xevent *out);

/* ---------- allocation failures and tree budgets ----------
 * Every builder returns NULL when an allocation fails or a tree outgrows its budget, and ho_last_error() (thread-local)
 * says which; nothing dereferences an unchecked realloc.  The budgets are not in the reference (which would run until the
 * process dies): a loose octree whose nodes shrink below ~0.4 m grows 8x per level whatever the polygon count
 * ("Octree - alt.cs":99-111 pads child boxes by an ABSOLUTE 0.1 m), so a careless maxDepth exhausts any machine. */
#define HO_MAX_TREE_NODES (1 << 24)          /* nodes of one octree / kd-tree                 */
#define HO_MAX_TREE_ITEMS (1ll << 28)        /* polygon-list entries alive at any one time    */
const char *ho_last_error(void);
void ho_set_error(const char *msg);
/* grow *p to `bytes` (realloc); 0 = ok, -1 = failed (*p unchanged, error set) */
int ho_grow(void **p, size_t bytes);
void *ho_alloc(size_t bytes);                 /* malloc that reports through ho_set_error (and honours the test hook below) */
/* test hooks: make the k-th following allocation fail; start the traversal stacks small (ho_prims.c) */
void ho_test_fail_alloc_after(int k);
void ho_test_stack_cap(int c);
int ho_initial_stack_cap(int usual);

/* ---------- Octree ("Octree - alt.cs") ---------- */
typedef struct ho_octree ho_octree;
ho_octree *ho_octree_build(const ho_topology *models, int32_t M, int32_t max_depth, int32_t max_polys);
void ho_octree_free(ho_octree *o);
int32_t ho_octree_node_count(const ho_octree *o);
/* node arrays (build order = pre-order of BuildOctree): box 6 doubles, first_child (-1 leaf), item range */
void ho_octree_export(const ho_octree *o, double *boxes /*n x 6*/, int32_t *first_child, int32_t *item_start, int32_t *item_count, int32_t *items);
int64_t ho_octree_item_total(const ho_octree *o);
int ho_octree_shoot(const ho_octree *o, const ho_topology *models, const ho_ray *R, int32_t top_index,
                    int32_t poly_origin1, int32_t poly_origin2, ho_xevent *out, ho_counters *ctr);
int ho_octree_shoot_batch(const ho_octree *o, const ho_topology *models, int32_t top_index,
                          int64_t n, const ho_ray *rays, const int32_t *excl1, const int32_t *excl2,
                          int nthreads, ho_xevent *out, ho_counters *ctr);

/* ---------- KDTree (KDTree.cs) ---------- */
typedef struct ho_kdtree ho_kdtree;
ho_kdtree *ho_kdtree_build(const ho_topology *models, int32_t M, int32_t max_depth, int32_t max_polys);
void ho_kdtree_free(ho_kdtree *k);
int32_t ho_kdtree_node_count(const ho_kdtree *k);
int64_t ho_kdtree_item_total(const ho_kdtree *k);
void ho_kdtree_export(const ho_kdtree *k, double *boxes, double *split, int32_t *axis, int32_t *left, int32_t *right,
                      int32_t *item_start, int32_t *item_count, int32_t *items);
int ho_kdtree_shoot(const ho_kdtree *k, const ho_topology *models, const ho_ray *R, int32_t top_index,
                    int32_t poly_origin1, int32_t poly_origin2, int32_t *mailbox, int32_t ray_id,
                    ho_xevent *out, ho_counters *ctr);
int ho_kdtree_shoot_batch(const ho_kdtree *k, const ho_topology *models, int32_t top_index,
                          int64_t n, const ho_ray *rays, const int32_t *excl1, const int32_t *excl2,
                          int32_t first_ray_id, int nthreads, ho_xevent *out, ho_counters *ctr);

/* ---------- brute force (not in the reference; cross-check only) ---------- */
int ho_brute_shoot(const ho_topology *T, const ho_ray *R, int32_t poly_origin1, int32_t poly_origin2, int full_uv, ho_xevent *out);

/* ---------- harness-defined specular bounce (not in the reference) ---------- */
void ho_reflect(const ho_topology *T, const ho_ray *R, const ho_xevent *ev, ho_ray *out);
void ho_reflect_batch(const ho_topology *T, int64_t n, const ho_ray *rays, const ho_xevent *ev, ho_ray *out);

#ifdef __cplusplus
}
#endif
#endif
