/*
 * hare_hip.h -- C-ABI of libhare_hip.so: the MI355X (gfx950) implementation of Hare's ray-cast
 * hot path, Spatial_Partition.Shoot.
 *
 * The reference (PachydermAcoustic/Hare, C#) has NO native boundary; its seam is the abstract
 * class Hare.Geometry.Spatial_Partition (Spatial_Partition.cs:27-35).  A drop-in is a C# class
 * deriving from it that P/Invokes the functions below (bindings/csharp/).  Each
 * entry point names the reference member it stands in for (file:line into the reference).
 *
 * Conventions
 *   - extern "C", cdecl, plain pointers and sizes; no C++/torch types cross this boundary.
 *   - every function returns HARE_OK (0) or a negative HARE_E_* code; the message of the last
 *     failure on the calling thread is hare_last_error().  No exception crosses the ABI.
 *   - the caller owns every host buffer; the library never keeps a host pointer after a call
 *     returns.  Everything behind hare_scene* belongs to the library until hare_scene_destroy.
 *   - per-ray conditions (origin outside the grid, NaN direction, ...) are not errors: they give
 *     the miss record X_Event() (Hare_Geometry_Primitives.cs:454-462).
 *   - threading: build/destroy calls are single-caller; hare_shoot_batch may be called from
 *     several host threads on one scene: up to four calls run side by side, each on its own
 *     staging buffers and streams, further callers wait for a free set (what the reference's
 *     Ray.ThreadID / mailbox pool serve, Voxel_Grid.cs:334-342); hare_shoot_device is
 *     stream-ordered: it enqueues and returns -- no allocation, no free, no wait on the device, a stream
 *     or an event on the host (every scratch it uses was reserved when the partition went to the device;
 *     hare_scene_get_option "hip_malloc_calls" ... let a caller check).  What it does take, for the few
 *     enqueues of one call, is the mutex of the launch slot (and scratch block) the call uses: calls from
 *     several threads contend only when they draw the same slot of the ring.  hare_shoot_one takes no lock.
 *   - devices: every call acts on the scene's own device and leaves the calling thread's current
 *     HIP device as it found it.
 *   - batches have NO CPU fallback: hare_shoot_batch / hare_shoot_device run the HIP kernels and fail
 *     with HARE_E_NODEVICE when no gfx950 device / HIP runtime is available.  The single-ray call
 *     hare_shoot_one -- Spatial_Partition.Shoot exactly as the reference exposes it -- runs on the
 *     calling host thread (a GPU round trip per ray would be ~100x slower than the reference).
 */
#ifndef HARE_HIP_H
#define HARE_HIP_H

#include <stdint.h>

#if defined(__GNUC__)
#define HARE_API __attribute__((visibility("default")))
#else
#define HARE_API
#endif

#ifdef __cplusplus
extern "C" {
#endif

#define HARE_OK 0
#define HARE_E_INVALID (-1)     /* bad argument                                   */
#define HARE_E_NOMEM (-2)       /* host or device allocation failed               */
#define HARE_E_HIP (-3)         /* a HIP runtime call failed                      */
#define HARE_E_NODEVICE (-4)    /* no HIP runtime / no usable GPU                 */
#define HARE_E_STATE (-5)       /* partition not built, scene not uploaded, ...   */
#define HARE_E_UNSUPPORTED (-6) /* e.g. polygon with more than 4 corners          */

/* Which Spatial_Partition subclass a shoot goes through */
#define HARE_KIND_VOXEL 0   /* Voxel_Grid  (Voxel_Grid.cs)       */
#define HARE_KIND_OCTREE 1  /* Octree      ("Octree - alt.cs")   */
#define HARE_KIND_KDTREE 2  /* KDTree      (KDTree.cs)           */

/* hare_shoot_* flags */
#define HARE_SHOOT_WRITEBACK_ORIGIN 1u /* also apply AABB.Intersect's origin move to rays[] (AABB_Main.cs:254-257) */
#define HARE_SHOOT_COUNT_WORK 2u       /* fill cells/entries/tests of hare_counters (slower diagnostic kernel)       */
#define HARE_SHOOT_SIMPLE_KERNEL 4u    /* voxel: one-ray-per-lane kernel instead of the persistent one (A/B testing) */
#define HARE_SHOOT_RETIRED_RAYS 8u     /* hare_shoot_device only (the bounce loop): poly_origin1 == -2 marks a ray that     */
                                       /* hare_reflect_device retired -> miss record, no traversal, not counted.  Without    */
                                       /* it -- and always in the host-buffer calls -- a negative poly_origin matches no      */
                                       /* polygon, as in the reference (Voxel_Grid.cs:477 compares indices only)              */

#define HARE_SHOOT_SLIM_EVENTS 16u     /* hare_shoot_batch / _sharded only: `out` receives slim records (below) instead of X_Events:   */
                                       /* 16 bytes per ray come back over the host link instead of 56.  Not together with                 */
                                       /* HARE_SHOOT_WRITEBACK_ORIGIN (HARE_E_INVALID): hare_expand_events redoes the origin move from    */
                                       /* the rays as they were passed in, which the write-back would have overwritten                   */

#define HARE_SHOOT_COUNT_OWN 64u       /* measurement: run the COUNTING BUILD of the production kernel the batch would get (the pool kernel of   */
                                       /* Voxel_Grid, hare_octree_dense, the kd-tree kernel) -- same events, slower -- and fill hare_counters     */
                                       /* with the work THAT kernel did: cells = voxels walked into / node records fetched, entries = list         */
                                       /* entries scanned, reserved[0] = candidates pre-culled, tests = exact polygon tests.  (COUNT_WORK counts   */
                                       /* the REFERENCE algorithm's work with a diagnostic kernel.)  HARE_E_UNSUPPORTED for a batch another        */
                                       /* kernel would serve.  No reference counterpart                                                            */
#define HARE_SHOOT_BOUNCE_LOOP 32u     /* hare_shoot_kernel_name only: name the kernel hare_bounce_device (<= 16 casts) launches for n rays           */

/* Hare.Geometry.Ray (Hare_Geometry_Primitives.cs:393-429): origin + direction.  Ray_ID/ThreadID
 * only serve the reference's mailbox pool and are not needed here. 48 bytes. */
typedef struct hare_ray {
    double x, y, z;
    double dx, dy, dz;
} hare_ray;

/* Hare.Geometry.X_Event (Hare_Geometry_Primitives.cs:435-481). 56 bytes.
 * Miss == X_Event(): t=u=v=0, X_Point null (0,0,0 here), Poly_id=-1, Hit=false. */
typedef struct hare_xevent {
    double t, u, v;
    double x, y, z;   /* X_Point */
    int32_t poly_id;  /* Poly_id */
    int32_t hit;      /* Hit     */
} hare_xevent;

/* Slim result records (HARE_SHOOT_SLIM_EVENTS): what an X_Event holds that the caller cannot recompute.
 *   X_Point is R.origin + R.direction * t by the reference's own expression (Hare_Geometry_Polygons.cs:652): evaluated by the
 *   caller in double precision without fusing it gives the same bits; Voxel_Grid.Shoot returns u = v = 0 (Voxel_Grid.cs:696-697).
 *   Voxel_Grid: hare_slim_event, 16 bytes.  hit = 1: t is X_Event.t and X_Point = o + d * t.  hit = 2: the ray started outside
 *   the grid and AABB.Intersect moved its origin (AABB_Main.cs:254-257): t is measured from the MOVED origin o'; X_Event.t =
 *   t + t_start and X_Point = o' + d * t -- hare_expand_events redoes the move and both sums bit for bit.
 *   Octree / KDTree: hare_slim_event_uv, 32 bytes (they return u, v; rays are never moved; hit is 0 or 1).
 * hare_expand_events turns n slim records back into full X_Events on the host (byte-identical to what the full call returns). */
typedef struct hare_slim_event {
    double t;
    int32_t poly_id;  /* -1 on a miss */
    int32_t hit;      /* 0 miss, 1 hit, 2 hit on a ray whose origin was moved (t from the moved origin) */
} hare_slim_event;
typedef struct hare_slim_event_uv {
    double t, u, v;
    int32_t poly_id;
    int32_t hit;
} hare_slim_event_uv;

/* Batch counters.  hits is what a multi-GPU run reduces across ranks. */
typedef struct hare_counters {
    uint64_t rays, hits;
    uint64_t cells;    /* grid cells / tree nodes visited (HARE_SHOOT_COUNT_WORK)           */
    uint64_t entries;  /* candidate-list entries scanned  (HARE_SHOOT_COUNT_WORK)           */
    uint64_t tests;    /* polygon tests performed; the GPU has no mailbox, so this counts   */
                       /* re-tests the reference's mailbox would skip                       */
    uint64_t reserved[3];
} hare_counters;

/* One Hare.Geometry.Topology as read back from the managed object:
 *   verts   = Model[m][poly, corner]          (Hare_Geometry_Topology.cs:418-424), P x 4 x 3 doubles,
 *             corner 3 ignored for triangles
 *   nverts  = Model[m].Polys[p].VertextCT     (Hare_Geometry_Polygons.cs:196), 3 or 4
 *   normals = Model[m].Normal(p)              (Hare_Geometry_Topology.cs:539), P x 3 doubles
 *   min/max = Model[m].Min / Model[m].Max     (Hare_Geometry_Topology.cs:50,58) after Finish_Topology() */
typedef struct hare_topology_desc {
    int32_t P;
    int32_t reserved;
    const double *verts;
    const int32_t *nverts;
    const double *normals;
    double min[3];
    double max[3];
} hare_topology_desc;

typedef struct hare_scene hare_scene;

/* ---- library / device ---- */
HARE_API const char *hare_version(void);
HARE_API const char *hare_last_error(void);
HARE_API int hare_device_count(int32_t *count);
/* Path of the HIP runtime the library bound to (diagnostics). */
HARE_API const char *hare_hip_runtime_path(void);

/* ---- helpers for hosts that do not go through the managed Topology ----
 * Polygon ctor normal (Hare_Geometry_Polygons.cs:159-171) and Finish_Topology bounds
 * (Hare_Geometry_Topology.cs:148-167), bit-identical to what the managed object would hold. */
HARE_API int hare_polygon_normals(const double *verts, const int32_t *nverts, int32_t P, double *normals_out);
HARE_API int hare_topology_bounds(const double *verts, const int32_t *nverts, int32_t P, double min_out[3], double max_out[3]);

/* Topology(Point[][]) ingest, for hosts that start from a raw polygon soup (Hare_Geometry_Topology.cs:120-142
 * ctor, :258-311 Build_Topology, :342-377 AddGetIndex; Point.Round / Point.Hash2
 * Hare_Geometry_Primitives.cs:230-250; MS_AABB Hare_Geometry_Topology.cs:677-697).  Every corner is rounded
 * with Math.Round(x, 15) and corners that fall into the same 1 mm Hash2 cell are merged onto the FIRST such
 * corner, in polygon order, exactly as the managed Topology does.
 *   soup           P x 4 x 3 corner coordinates (slot 3 ignored for triangles)
 *   verts_out      P x 4 x 3: the merged corner coordinates (unused slots zero) -- what hare_scene_create,
 *                  hare_polygon_normals and hare_topology_bounds expect
 *   corner_vertex  nullable, P x 4: index of each corner in Vertices_List order (-1 in unused slots)
 *   vertices_out   nullable, capacity sum(nverts) x 3: Vertices_List
 *   n_vertices_out nullable: length of Vertices_List
 * A polygon with nverts other than 3 or 4 is HARE_E_UNSUPPORTED (the reference throws NotImplementedException). */
HARE_API int hare_topology_ingest(const double *soup, const int32_t *nverts, int32_t P, double *verts_out,
                                  int32_t *corner_vertex, double *vertices_out, int32_t *n_vertices_out);

/* ---- scene = Spatial_Partition.Model (Spatial_Partition.cs:29) ----
 * Copies the topologies; `device` is the HIP device ordinal that will hold the scene. */
HARE_API int hare_scene_create(const hare_topology_desc *topos, int32_t n_topos, int32_t device, hare_scene **out);
HARE_API void hare_scene_destroy(hare_scene *s);

/* Diagnostics and A/B switches of ONE scene, for tests, profiling and tools; production callers never need them.  A scene takes
 * its defaults from the environment ONCE, inside hare_scene_create (HARE_BUILD=host; and, only in a process that opted in with
 * HARE_DEV=1, HARE_VOXEL_KERNEL = pool|persist, HARE_OCTREE_KERNEL = dense|group|persist|pool, HARE_TICKET, HARE_K1P_STATIC_RAYS, HARE_K2P_STATIC_RAYS,
 * HARE_BATCH_CHUNKS, HARE_OCTREE_TIGHT, HARE_VOXEL_TIGHT, HARE_TUNE): no call reads the environment afterwards.  Options:
 *   "build_host"      1: host builders even when a GPU is present (identical lists either way)
 *   "voxel_kernel"    0: the library's rule, 1: hare_voxel_persist_* (K1p), 2: hare_voxel_pool_* (K1q)
 *   "octree_kernel"   0: the library's rule (K2g below 81 920 rays on a 256-CU part -- 320 per CU --, K2d from there), 1: hare_octree_persist (K2p, one lane per ray),
 *                     2: hare_octree_pool (K2q), 3: hare_octree_group (K2g, eight lanes per ray), 4: hare_octree_dense (K2d: K2p with its leaf
 *                     entries spread densely over the wave and its exact tests deferred)
 *   "bounce_fused"    1: hare_bounce_device / hare_bounce_batch (last cast's events only) run a Voxel_Grid's bounce loop as ONE launch where they can; 0 (default): a launch per cast
 *   "bounce_pack"     1 (default): behind every reflection of a Voxel_Grid's launch-per-cast loop (last cast's events only) the blocks of 64 consecutive rays
 *                     in which a ray still lives are listed on the device, and the next cast walks that list: a cast costs nothing for rays retired in
 *                     whole blocks (open scenes: 72 -> 14 us per million retired rays).  One more one-workgroup launch per cast: 0 saves a closed room ~1 %
 *   "octree_tail"     what finishes the rays K2p's waves still walk at the end of a launch: 2 (default) hare_octree_group_tail (eight lanes per
 *                     ray, every ray a wave holds 32 rounds after its tickets ran dry), 1 hare_octree_tail (a wave per ray, a wave's last 16), 0 nothing
 *   "k2p_tail_max", "k2p_tail_patience"   the hand-over rule (0 / -1: the library's)
 *   "kdtree_kernel"   0 the library's rule (hare_kdtree_dense: persistent waves, one-line node records with both children's tight boxes,
 *                     leaves pre-culled densely, exact tests deferred), 1 hare_kdtree_shoot (one ray per lane: A/B baseline), 2 hare_kdtree_dense
 *   "ticket_rays", "k1p_static_rays" (both voxel kernels), "k2p_static_rays", "batch_chunks"   0: the library's rule, else the value
 *   "coop_tail"       1 (default): a wave that has drawn its last rays traces the last few with all 64 lanes (heavy rays); 0: off
 *   "wide_drain"      1 (default): the pool kernel spends the lanes its finished rays leave on the rays that remain (several lanes per
 *                     ray: its candidates four per lane, the occupied voxels ahead one per lane); 0: off.  Results never depend on it
 *   "voxel_walk"      1 (default): the pool kernel's DDA step loop (Voxel_Grid.cs:713-759) as written by hand for gfx950 -- the per-axis
 *                     updates under the axis' own EXEC mask; 0: the compiler's loop (A/B).  The same steps in the same order: results never depend on it
 *   "voxel_skip"      1: the pool kernel's walk crosses an EMPTY aligned block of 4 x 4 x 4 voxels in one operation -- the exact closed-form skip (the DDA as a
 *                     merge of three sequences of sequential adds): the same voxel, the same tMax bit patterns, the same results.  0 (default): it is
 *                     slower than the hand-written step on this hardware; kept as a tested option
 *   "octree_tight"    1 (default): the octree and kd-tree kernels drop a node whose subtree's polygons the ray cannot hit -- per node the box of
 *                     all polygons its subtree lists, built when the tree goes to the device; 0: every node the reference visits.  Results never
 *                     depend on it (an X_Event is the reference's bit for bit either way)
 *   "voxel_tight"     the same per voxel: a ray without a hit walks on past an occupied voxel whose polygons it cannot hit; 1 (default) / 0.
 *                     The boxes cost 32 B per voxel and topology and exist only while this is on and the pool kernel serves the grid (up to
 *                     512 voxels a side); switching it on later builds them then
 *   "voxel_order"     1 (default): the pool kernel takes the rays of a batch of primary rays (no exclusion arrays, from 1 572 864 rays), window by
 *                     window of 4 096, in the order of their estimated walk length -- a wave's rays then cost about the same; 0 never, 2 every
 *                     batch.  Rays and events stay where the caller has them; results never depend on it
 *   "voxel_order_max_rays"  the largest batch that pass serves (default 16 777 216): its scratch -- a ring of 4 blocks of that many 4-byte entries,
 *                     256 MiB by default -- is reserved when the grid goes to the device, so that no shoot ever allocates; a larger batch, a stream
 *                     under capture, or a failed reservation runs in the caller's order (same results).  0: no ring
 *   "voxel_tight_max_mb"  budget for those boxes in MiB (0, the default: none).  Over budget -- or out of device memory -- the grid is
 *                     built and traced without them: never an error, never a different result
 *   "dev"             1: developer flag bits of hare_shoot_* (timeline, phase profile, cull audit) pass
 * Single-caller like the build calls: not to be changed while shoots are in flight on the scene. */
HARE_API int hare_scene_set_option(hare_scene *s, const char *name, int64_t value);

/* Read an option back (any name hare_scene_set_option takes), or one of the read-only figures a host sizes its memory by:
 *   "voxel_tight_bytes"     device bytes the voxels' tight boxes take on this scene (32 B per voxel and topology; 0: none -- the option is
 *                           off, the grid is one the pool kernel does not serve, over "voxel_tight_max_mb", or their allocation failed:
 *                           the grid is then traced without them, same results)
 *   "voxel_order_bytes"     device bytes of the pool kernel's order ring (0: none -- "voxel_order" off, "voxel_order_max_rays" 0, a grid the
 *                           pool kernel does not serve, or the reservation failed)
 *   "hip_malloc_calls", "hip_free_calls", "hip_sync_calls"   process-wide counts of hipMalloc / hipFree / host-side waits (hipDeviceSynchronize,
 *                           hipStreamSynchronize, hipEventSynchronize) this library has made: a caller (or a test) can hold the stream-ordered
 *                           entry points to "none of these"
 *   "octree_scratch_bytes"  device bytes of the octree kernels' scratch ring (hand-over records and stack spill; 0 before the first
 *                           octree launch that needs one)
 * No reference counterpart: Hare has no device memory to account for. */
HARE_API int hare_scene_get_option(const hare_scene *s, const char *name, int64_t *value);

/* ---- partition constructors ----
 * The voxel grid and the octree (of a single topology) are built on the GPU when one is present (environment
 * HARE_BUILD=host when the scene is created, or option "build_host", forces the host builders); both builders produce identical lists.  The KDTree is built on the host.
 * Voxel_Grid(Topology[] Model_in, int Domain)                      Voxel_Grid.cs:48-121  */
HARE_API int hare_voxel_build(hare_scene *s, int32_t domain);
/* Voxel_Grid(Topology[] Model_in, int MaxDomain, int Avg_polys)    Voxel_Grid.cs:128-254 */
HARE_API int hare_voxel_build_adaptive(hare_scene *s, int32_t max_domain, int32_t avg_polys);
/* Octree(Topology[] Model_In, int maxDepth, int maxPolygonsPerNode) "Octree - alt.cs":45-89
 * Several topologies: as the reference is written -- the root cube and the ids 0..P-1 of the LAST topology, binned by the
 * vertices of topology 0 (:63-88,123); the kd-tree's box grows over all topologies (KDTree.cs:67-87).  HARE_E_INVALID where
 * the reference would index out of range: a topology that gets split and has more polygons than topology 0 (at build), a
 * top_index whose topology has fewer polygons than the last one (at shoot). */
HARE_API int hare_octree_build(hare_scene *s, int32_t max_depth, int32_t max_polys);
/* KDTree(Topology[] Model_In, int maxDepth, int maxPolygonsPerNode) KDTree.cs:51-88 */
HARE_API int hare_kdtree_build(hare_scene *s, int32_t max_depth, int32_t max_polys);

/* ---- partition introspection (Voxel_Grid public members + what the parity tests compare) ---- */
typedef struct hare_voxel_info {
    int32_t ct;              /* VoxelCtX = VoxelCtY = VoxelCtZ                          */
    int32_t n_topos;
    double obox_min[3];      /* MinPt (Voxel_Grid.cs:785-791)                            */
    double obox_max[3];
    double box_dims[3];      /* Xdim / Ydim / Zdim (Voxel_Grid.cs:763-783)               */
    double voxel_dims[3];
    double char_step;        /* Spatial_Partition.Char_Step (Spatial_Partition.cs:31)    */
    uint64_t total_items;    /* sum of Voxel_Inv[x,y,z,m].Count over the grid, topology 0 */
    int32_t built_on_device; /* 1: lists came from the GPU builder, 0: from the host builder  */
    int32_t reserved;
} hare_voxel_info;
HARE_API int hare_voxel_get_info(const hare_scene *s, hare_voxel_info *out);
/* Voxel_Inv[x,y,z,top] (Voxel_Grid.cs:33) as CSR: cell = (x*ct + y)*ct + z; cell_start has
 * ct^3 + 1 entries, items has cell_start[ct^3] entries (ascending polygon index per cell).
 * items may be NULL to fetch cell_start only (its last entry sizes the items buffer). */
HARE_API int hare_voxel_get_lists(const hare_scene *s, int32_t top_index, uint32_t *cell_start, int32_t *items);

typedef struct hare_tree_info {
    int32_t n_nodes;
    int32_t max_depth;
    int32_t max_polys;
    int32_t built_on_device;    /* 1: the membership tests ran on the GPU (octree); same arrays either way */
    uint64_t total_items;
} hare_tree_info;
HARE_API int hare_octree_get_info(const hare_scene *s, hare_tree_info *out);
/* nodes in creation order (root, then each split's 8 children, depth-first):
 * boxes n x 6 (min xyz, max xyz), first_child (-1 = leaf), item_start/item_count into items */
HARE_API int hare_octree_get_nodes(const hare_scene *s, double *boxes, int32_t *first_child, int32_t *item_start,
                          int32_t *item_count, int32_t *items);
HARE_API int hare_kdtree_get_info(const hare_scene *s, hare_tree_info *out);
HARE_API int hare_kdtree_get_nodes(const hare_scene *s, double *boxes, double *split, int32_t *axis, int32_t *left,
                          int32_t *right, int32_t *item_start, int32_t *item_count, int32_t *items);

/* ---- Shoot ----
 * bool Shoot(Ray R, int top_index, out X_Event Ret_event)                                  Spatial_Partition.cs:32
 * bool Shoot(Ray R, int top_index, out X_Event Ret_event, int poly_origin1, int poly_origin2 = -1)   :33
 * for n rays at once.  excl1/excl2 (nullable) are poly_origin1/poly_origin2 per ray, -1 = none.
 * rays is read (and, with HARE_SHOOT_WRITEBACK_ORIGIN, updated like the reference mutates R).
 * Host buffers; the call copies to the scene's device, runs the kernel and copies back.  With HARE_SHOOT_SLIM_EVENTS `out` is
 * an array of n hare_slim_event (Voxel_Grid) or hare_slim_event_uv (Octree, KDTree) instead of n hare_xevent. */
HARE_API int hare_shoot_batch(hare_scene *s, int32_t kind, int32_t top_index, int64_t n, hare_ray *rays,
                     const int32_t *excl1, const int32_t *excl2, uint32_t flags, hare_xevent *out,
                     hare_counters *ctr /* nullable */);

/* Slim records -> X_Events, on the calling host thread(s): `slim` is what a HARE_SHOOT_SLIM_EVENTS call on `s` wrote for `rays`
 * (the rays as they were passed in), kind as in that call.  Needs no GPU. */
HARE_API int hare_expand_events(const hare_scene *s, int32_t kind, int64_t n, const hare_ray *rays, const void *slim,
                                hare_xevent *out);

/* The same call over several devices of one node from ONE process (hosts without torch.distributed, e.g. the
 * .NET shim): scenes[k] is the scene on device k's ordinal of choice -- same topologies, same partition, built by
 * the caller with hare_scene_create(..., device_k, ...) + hare_*_build -- and rays [n*k/G, n*(k+1)/G) go to it
 * (the contiguous split of SURVEY.md 8(e); scene replicated, no exchange between devices).  One host thread per
 * scene drives its copies and kernel; outputs land in the same slices of `out`, so the result is byte-identical
 * to the one-device call; `ctr` is the sum over devices.  The first failing shard's code and message are returned. */
HARE_API int hare_shoot_batch_sharded(hare_scene *const *scenes, int32_t n_scenes, int32_t kind, int32_t top_index, int64_t n,
                                      hare_ray *rays, const int32_t *excl1, const int32_t *excl2, uint32_t flags,
                                      hare_xevent *out, hare_counters *ctr);

/* Same with DEVICE pointers on the scene's device and a caller stream (hipStream_t as void*,
 * NULL = default stream); stream-ordered, does not synchronise.  d_counters (nullable) points to
 * a device hare_counters that the kernel ACCUMULATES into.  Calls may be issued from several host
 * threads and on several streams; the scene keeps per-launch scratch (work tickets, counter shards) in a ring of
 * 64 slots: the 65th launch in flight is ordered behind the first (it waits for an event that launch recorded), so any
 * number of launches may be queued.  The same holds for the scratch rings some launches use beside it (the pool kernel's ray
 * order: 4 blocks; the octree kernels' hand-over records and stack spill: 8): a block's next user waits ON ITS STREAM for the event the previous
 * one recorded.  Under stream capture the order pass is skipped (results unchanged).  rays, the exclusion arrays, events and counters
 * must not overlap (HARE_E_INVALID). */
HARE_API int hare_shoot_device(hare_scene *s, int32_t kind, int32_t top_index, int64_t n, void *d_rays,
                      const void *d_excl1, const void *d_excl2, uint32_t flags, void *d_out,
                      void *d_counters, void *stream);

/* Name of the gfx950 kernel a hare_shoot_device / hare_shoot_batch call with these arguments launches (for profiles:
 * rocprofv3 lists kernels by this name).  The voxel path has two production kernels and picks by batch size. */
HARE_API const char *hare_shoot_kernel_name(const hare_scene *s, int32_t kind, int32_t top_index, int64_t n, uint32_t flags);

/* ---- Shoot, one ray (unchanged reference call sites) ----
 * bool Shoot(Ray R, int top_index, out X_Event Ret_event, int poly_origin1 = -1, int poly_origin2 = -1)
 * (Spatial_Partition.cs:32-33; Voxel_Grid.cs:351,561; "Octree - alt.cs":154; KDTree.cs:193) for ONE ray, on the calling
 * host thread: the same trace the simple HIP kernels run (hare_amd/csrc/hare_trace.h), instantiated for the host over a
 * host mirror of the scene that is built on first use.  Results are bit-identical to the batch calls.  *ray is
 * updated like the reference mutates R when the origin lies outside the grid (AABB_Main.cs:254-257).  Needs no GPU;
 * lock-free, callable concurrently from any number of threads (no mailbox: SURVEY.md F7).  out->hit is the return
 * value of the reference's Shoot. */
HARE_API int hare_shoot_one(hare_scene *s, int32_t kind, int32_t top_index, hare_ray *ray, int32_t poly_origin1,
                            int32_t poly_origin2, hare_xevent *out);

/* ---- occlusion predicate (harness-defined, SURVEY.md F13 / 8(a) A9: the reference has no any-hit API; the seam it
 * would sit beside is Spatial_Partition.cs:32-33) ----
 * occluded[i] = Shoot(rays[i]) hit something AND that closest hit has t < tmax[i]  (tmax NULL: any hit counts).
 * Defined on the CLOSEST hit the reference's Shoot would return, so that it is pinned by the same oracle as Shoot, including
 * the reference's miss-on-grid-exit rule (Voxel_Grid.cs:716-757) and the octree's early return ("Octree - alt.cs":233).
 *   events != NULL   the closest-hit cast as hare_shoot_*, the X_Events returned, the flags derived from them
 *   events == NULL   flags only, from kernels that stop a ray as soon as its flag is decided: the voxel walk ends when it has
 *                    passed t_max without a hit below it pending (the pending-hit confirmation of Voxel_Grid.cs:705-709 is
 *                    kept: a hit counts when the reference would return it); the octree walk ends at the first hit below
 *                    t_max (no node is skipped for lying beyond t_max: with the reference's far-to-near order and early
 *                    return that would change which hit is "the" hit); the kd-tree walk ends at the first accepted hit below
 *                    t_max and never enters a subtree whose tight box the ray reaches at or beyond t_max (KDTree.Shoot returns
 *                    the smallest accepted t over ALL polygons, so "some polygon is accepted below t_max" is its flag).  The
 *                    flags are identical either way; the host call
 *                    then brings back 4 bytes per ray instead of 56.  counters: rays, and hits = number of occluded rays.
 * rays[] is never written. */
HARE_API int hare_occluded_device(hare_scene *s, int32_t kind, int32_t top_index, int64_t n, void *d_rays,
                                  const void *d_excl1, const void *d_excl2, const void *d_tmax /* n doubles, nullable */,
                                  uint32_t flags, void *d_events /* n x 56 B, nullable */, void *d_occluded /* n int32 */,
                                  void *d_counters, void *stream);
HARE_API int hare_occluded_batch(hare_scene *s, int32_t kind, int32_t top_index, int64_t n, hare_ray *rays,
                                 const int32_t *excl1, const int32_t *excl2, const double *tmax /* nullable */,
                                 uint32_t flags, int32_t *occluded, hare_xevent *events /* nullable */,
                                 hare_counters *ctr /* nullable */);
/* the same over several devices from one process (contiguous ray shards, as hare_shoot_batch_sharded) */
HARE_API int hare_occluded_batch_sharded(hare_scene *const *scenes, int32_t n_scenes, int32_t kind, int32_t top_index, int64_t n,
                                         hare_ray *rays, const int32_t *excl1, const int32_t *excl2, const double *tmax,
                                         uint32_t flags, int32_t *occluded, hare_xevent *events, hare_counters *ctr);

/* ---- specular bounce (harness-defined; the reference leaves reflection to its caller, which
 * re-shoots with poly_origin1 = the previous Poly_id -- Voxel_Grid.cs:351,477) ----
 * For every ray with events[i].hit: origin <- X_Point, direction <- d - (2*(d.n))*n with
 * n = Model[top].Normal(Poly_id), excl_out[i] <- Poly_id.  Rays that missed keep their record
 * and get excl_out[i] = -2 (dead: a later hare_shoot_device with HARE_SHOOT_RETIRED_RAYS reports a
 * miss for them immediately and does not count them).
 * Device pointers, stream-ordered. */
HARE_API int hare_reflect_device(hare_scene *s, int32_t top_index, int64_t n, void *d_rays, const void *d_events,
                        void *d_excl_out, void *stream);

/* ---- the bounce loop on DEVICE buffers, stream-ordered, no host synchronisation (harness-defined like hare_reflect_device) ----
 * `bounces` casts per ray: Shoot, reflect about Model[top].Normal(Poly_id) (Hare_Geometry_Polygons.cs:161-171), Shoot again with
 * poly_origin1 = the polygon just hit (Spatial_Partition.cs:33; Voxel_Grid.cs:351,477); a ray that misses is retired (its later
 * events are the miss record X_Event(), it is not counted).
 * By default `bounces` x (shoot + reflect) launches, retired rays skipped.  With the scene option "bounce_fused" = 1, a Voxel_Grid
 * wherever the pool kernel serves a batch, and bounces <= 16: ONE persistent launch (hare_voxel_bounce_*) in which every ray runs
 * through its casts on its own -- rays are independent across casts too, so no cast waits for the slowest ray of the one before.
 * Results are identical either way; measured on MI355X the single launch gains 2.5 % in the 100k-triangle hall and loses up to
 * 10 % in the 1M-triangle cathedral (profiles/r04_experiments/EXPERIMENTS.md), hence the default.
 *   d_rays               n rays: READ AND OVERWRITTEN (work array; every ray's last reflection remains)
 *   d_excl1 / d_excl2    nullable, read only: poly_origin1 / poly_origin2 of cast 0 (a negative index excludes nothing)
 *   d_work               scratch, 2 n int32
 *   d_events_all         nullable: bounces x n X_Events, cast-major
 *   d_events_last        the n X_Events of the last cast (nullable when d_events_all is given)
 *   d_counters           nullable: totals, ACCUMULATED (rays = casts with a live ray, hits)
 *   d_counters_per_cast  nullable: `bounces` hare_counters, ACCUMULATED (rays = rays alive in that cast, hits = rays that live on)
 *   flags                HARE_SHOOT_COUNT_WORK / HARE_SHOOT_SIMPLE_KERNEL only (either forces the launch per cast) */
HARE_API int hare_bounce_device(hare_scene *s, int32_t kind, int32_t top_index, int64_t n, void *d_rays, const void *d_excl1,
                                const void *d_excl2, int32_t bounces, uint32_t flags, void *d_work, void *d_events_all,
                                void *d_events_last, void *d_counters, void *d_counters_per_cast, void *stream);

/* ---- the whole bounce loop behind one call, from host buffers (harness-defined like hare_reflect_device) ----
 * What a Pachyderm-style caller does per ray with the reference -- Shoot, reflect about Model[top].Normal(Poly_id)
 * (Hare_Geometry_Polygons.cs:161-171), Shoot again with poly_origin1 = the polygon just hit (Spatial_Partition.cs:33;
 * Voxel_Grid.cs:351,477) -- for n rays and `bounces` casts, device-resident: the rays go up once, every cast and every
 * reflection runs on the scene's GPU, and only the requested X_Events come down (the download of a cast overlaps the next cast).
 *   cast 0 shoots rays[] with excl1 / excl2 (nullable; poly_origin1 / poly_origin2 per ray as in hare_shoot_batch: a negative
 *   index excludes nothing); cast b > 0 shoots the reflections of the rays that hit in cast b - 1, excluding the polygon they
 *   left.  A ray that misses is retired: its X_Event in every later cast is the miss record X_Event(), and it is not counted.
 *   When a quarter or more of the rays in flight have died the survivors are packed (stably) and later casts run on them
 *   alone; their events are put back in the caller's order on the device.  Results do not depend on whether that happened.
 *   events_all     nullable: bounces x n records, cast-major (cast b at events_all + b * n)
 *   events_last    nullable: the n records of the last cast
 *   ctr            nullable: counters summed over the casts (rays = live casts)
 *   ctr_per_cast   nullable: `bounces` blocks, one per cast (rays = rays alive in that cast, hits = rays that live on)
 *   flags          HARE_SHOOT_COUNT_WORK / HARE_SHOOT_SIMPLE_KERNEL only; rays[] is never written
 * Threading as hare_shoot_batch (a staging context per call in flight, four per scene). */
HARE_API int hare_bounce_batch(hare_scene *s, int32_t kind, int32_t top_index, int64_t n, const hare_ray *rays,
                               const int32_t *excl1, const int32_t *excl2, int32_t bounces, uint32_t flags,
                               hare_xevent *events_all, hare_xevent *events_last, hare_counters *ctr,
                               hare_counters *ctr_per_cast);
/* The same over several devices from one process: rays [n*k/G, n*(k+1)/G) go to scenes[k] (as hare_shoot_batch_sharded); every
 * shard keeps its rays resident on its own device for all casts; outputs are byte-identical to the one-device call. */
HARE_API int hare_bounce_batch_sharded(hare_scene *const *scenes, int32_t n_scenes, int32_t kind, int32_t top_index, int64_t n,
                                       const hare_ray *rays, const int32_t *excl1, const int32_t *excl2, int32_t bounces,
                                       uint32_t flags, hare_xevent *events_all, hare_xevent *events_last, hare_counters *ctr,
                                       hare_counters *ctr_per_cast);

#ifdef __cplusplus
}
#endif
#endif /* HARE_HIP_H */
