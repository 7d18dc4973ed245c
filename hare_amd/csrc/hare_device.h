// hare_device.h -- kernel argument blocks shared by the host library (g++) and the gfx950
// kernels (hipcc).  Plain data only.  Product code; nothing from oracle/.
#pragma once
#include <stdint.h>
#include "hare_math.h"

namespace hare {

// Wire records of the C-ABI (include/hare_hip.h); duplicated here as plain structs so kernels do
// not include the public header.  Layout checked against the ABI in api.cpp.
struct RayRec {            // Hare.Geometry.Ray: Hare_Geometry_Primitives.cs:393-429
    double x, y, z, dx, dy, dz;
};
struct XEventRec {         // Hare.Geometry.X_Event: Hare_Geometry_Primitives.cs:435-481
    double t, u, v, x, y, z;
    int32_t poly_id;
    int32_t hit;
};
static_assert(sizeof(RayRec) == 48, "ray wire size");
static_assert(sizeof(XEventRec) == 56, "x_event wire size");

struct CellRec {           // one grid cell: [start, start+count) into items; 16 bytes = one load
    uint32_t start;
    uint32_t count;
    int32_t i0, i1;        // items[start], items[start+1] inlined: entering a cell costs ONE dependent load
};
static_assert(sizeof(CellRec) == 16, "cell record size");

enum : uint32_t {
    SHOOT_WRITEBACK_ORIGIN = 1u,  // reproduce AABB.Intersect's origin move on the caller's rays (F11)
    SHOOT_SIMPLE_KERNEL = 4u,     // use the one-ray-per-lane kernel instead of the persistent one (A/B, diagnostics)
    SHOOT_RETIRED_SILENT = 0x80000u,   // internal (bounce loop into ONE event buffer, launch.cpp): a retired ray's slot already holds its miss record -- the pool kernel writes nothing for it
    SHOOT_RETIRED_RAYS = 8u,      // bounce loop: excl1 == -2 marks a ray hare_reflect retired -> miss record, no traversal, not counted
    SHOOT_SLIM_EVENTS = 16u,      // host-buffer calls: 16 / 32-byte result records (hare_slim_event*).  To the voxel kernels it means: for a
                                  // ray whose origin AABB.Intersect moved, leave tmin (t measured from the moved origin) in X_Event.u, which
                                  // Voxel_Grid always returns as 0 -- hare_events_pack_slim reads it from there (t itself is tmin + t_start,
                                  // from which tmin cannot be recovered bit for bit)
};

// Device counters (one block per scene, accumulated with atomics; 8 x u64)
enum { CTR_RAYS = 0, CTR_HITS = 1, CTR_CELLS = 2, CTR_ENTRIES = 3, CTR_TESTS = 4, CTR_CULLS = 5 /* HARE_SHOOT_COUNT_OWN: candidates pre-culled */,
       CTR_STEPS = 6 /* HARE_SHOOT_COUNT_OWN, Voxel_Grid: walk operations executed (a step or a block jump) */, CTR_WORDS = 8 };
// what one lane of a counting build (HARE_SHOOT_COUNT_OWN, the *_own kernels) has seen its rays do
struct OwnWork { unsigned cells = 0, entries = 0, culls = 0, tests = 0, steps = 0; };     // steps: DDA steps / block jumps EXECUTED (= cells unless "voxel_skip")

// The FP32 pre-cull's operands (v0, e1f, e2f) live in a dense array of their own, apart from the 128-byte records the exact test
// reads.  Cell and leaf lists hold runs of consecutive polygon ids (82 % of neighbouring entries differ by one in the bench scenes,
// tools/list_locality.py), so the candidates a ray scans sit next to each other (round 2: -4 % kernel time at C2 and C3, +12 %
// casts/s in the bounce loop over gathering from the record heads).
//   HARE_CULL32 = 1 (default): 32 bytes per polygon, one 32-byte sector, TWO 16-byte gathers:
//       word 0-1   v0 quantised to 21 bits per axis on the topology's bounding box: q = round((v0 - org) / step), x | y << 21 | z << 42
//       word 2-7   e1f, e2f (FP32 edges as before)
//     A topology WITH quadrilaterals (round 5) has 48-byte records (CullFrame::stride), three gathers: + e3f = (float)(v3 - v0) and a
//     flag word.  Quadrilateral.Intersect tries the triangles (0,1,2) and (2,3,0) (Hare_Geometry_Polygons.cs:784-823), so a quadrilateral
//     is culled when cull_fp32 is certain to miss BOTH -- edges (e1, e2) and (e2, e3) from the same corner v0.  Until round 5 a
//     quadrilateral was never pre-culled (NaN in e1f[0]) and every list entry that named one cost an exact FP64 test: 5.6 of them
//     per ray on the hall with its lattices un-split, against 1.4 on the triangulated hall.
//     The quantisation (<= step / 2 per axis: 21 um on a 90 m cathedral) and the FP32 arithmetic that rebuilds tv = o - v0 from it
//     only widen the cull's margins (cull_fp32's tv_err, hare_math.h); the exact FP64 test that decides a hit still reads the
//     128-byte record.  Audited like the cull itself (hare_cull_audit: 0 rejected hits over every ray x polygon pair).
//   HARE_CULL32 = 0: 48 bytes, bytes 0..47 of the PolyRec (v0 as FP64 + the edges), THREE gathers -- round 2's layout, the A/B baseline.
#ifndef HARE_CULL32
#define HARE_CULL32 1
#endif
constexpr int kCullStride = HARE_CULL32 ? 32 : 48;
struct CullFrame {             // how tv = o - v0 is rebuilt from a 32-byte record (per topology; wave-uniform kernel arguments)
    double org[3];             // the box's min corner: o_rel = (float)(o - org)
    float step[3];             // quantisation step per axis (0 for a flat axis)
    float err0;                // step_max / 2 + 2^-22 * extent_max, rounded up: the ray-independent part of tv_err
    int32_t stride;            // bytes per record: 32, or 48 for a topology with quadrilaterals (+ e3f and the quadrilateral flag)
    int32_t pad;
};

struct VoxelArgs {
    const PolyRec* polys;
    const QuadRec* quads;      // null when the topology is all triangles
    const CellRec* cells;      // ct^3, cell = (x*ct + y)*ct + z
    const int32_t* items;
    const uint32_t* occ;       // occupancy bitmap, staged in LDS by the persistent kernel (<= 64 KB): one bit per
                               // block of (2^occ_shift)^3 voxels, occ_cd blocks per axis; shift 0 = one bit per voxel
    int32_t ct;
    int32_t occ_words;
    int32_t occ_shift;
    int32_t occ_cd;
    double omin[3], omax[3];   // OBox
    double vd[3];              // VoxelDims
    const unsigned char* cull; // the pre-cull's dense records, kCullStride bytes per polygon
    CullFrame cf;
    const float* cellbox;      // nullable: per voxel {lo xyz, hi xyz, 0, 0}: the box of ALL polygons its list holds, grown by a margin and rounded
                               // outwards (build_kernels.hip: hare_cell_boxes).  A ray that misses it cannot hit any of them (K1q, voxel_pool.hip)
    double cellbox_mid[3];     // ... for origins with |o - cellbox_mid|_inf <= cellbox_rad only (the margin is sized for those)
    double cellbox_rad;
    const uint32_t* bocc;      // nullable (scene option "voxel_skip"): one bit per aligned block of 4^3 voxels that holds an occupied voxel (build_kernels.hip:
                               // hare_block_occ), staged in LDS behind the pools by the pool kernel; bocc_nb blocks per axis
    int32_t bocc_nb;
    int32_t bocc_words;
};

struct OctNode {               // 64 bytes
    double bmin[3], bmax[3];
    int32_t first_child;       // interior: index of the first of eight consecutive children; leaf: negative.  On the host (builders,
                               // hare_shoot_one, hare_octree_get_nodes) a leaf holds -1; the DEVICE copy of a leaf holds -2 - items[start + 1]
                               // (-1 when the list has fewer than two entries) and `pad` items[start] (-1: empty list): entering a leaf
                               // costs no list load for its first pair of candidates (as CellRec does for the grid)
    int32_t item_start;
    int32_t item_count;
    int32_t pad;               // device copy: a leaf's items[start] (above); an interior node's mask of children that are empty leaves, by octant
                               // (an interior node's item_start / item_count there: the same mask in cursor order per direction mask, device_scene.cpp)
};
static_assert(sizeof(OctNode) == 64, "octree node size");

struct OctreeArgs {
    const PolyRec* polys;
    const QuadRec* quads;
    const OctNode* nodes;
    const int32_t* items;
    int32_t n_nodes;
    int32_t max_depth;
    const unsigned char* cull; // as in VoxelArgs
    CullFrame cf;
    const float* tight;        // nullable: per node {lo xyz, hi xyz, 0, 0}: the box of ALL polygons the node's subtree lists, grown by a margin and
                               // rounded outwards (device_scene.cpp: make_tight_boxes).  A tame ray (K2p / K2d) that misses it cannot hit any of them.
    double tight_mid[3];       // ... for origins with |o - tight_mid|_inf <= tight_rad only (the margin is sized for those)
    double tight_rad;
};

// ---- the pre-cull as the kernels use it: load a candidate's record (cull_load: the gathers), prepare the ray once per task
// (cull_ray), test (cull_test: true = the exact test is certain to miss).  One code path for K1p, K1q, K2p, K2q and the audit.
#if defined(__HIPCC__)
struct CullRaw {
#if HARE_CULL32
    uint4 w0;                  // q.lo q.hi | e1f.x e1f.y
    float4 w1;                 // e1f.z e2f.x e2f.y e2f.z
    float4 w2;                 // 48-byte records only: e3f.x e3f.y e3f.z | 1.0f for a quadrilateral, 0 for a triangle
#else
    double2 c0;                // v0.x v0.y
    uint4 r1;                  // v0.z | e1f.x e1f.y
    float4 fb;                 // e1f.z e2f.x e2f.y e2f.z
#endif
};
struct CullRay {
#if HARE_CULL32
    float ox, oy, oz;          // (float)(o - org)
    float err;                 // tv_err for this ray: err0 + 2^-22 * |o - org|_1
#else
    double ox, oy, oz;
#endif
    float dfx, dfy, dfz, dm;
};
// Q: 0 the topology has no quadrilaterals (32-byte records), 1 it has (48-byte records), -1 decided at run time from the frame
// (the tree kernels, which are not compiled per topology kind)
template <int Q = -1, class Args>
__device__ __forceinline__ CullRaw cull_load(const Args& g, int i)
{
#if HARE_CULL32
    const bool wide = Q == 1 || (Q < 0 && g.cf.stride == 48);
    const unsigned char* rec = g.cull + (Q == 0 ? (size_t)(unsigned)i * 32u : (Q == 1 ? (size_t)(unsigned)i * 48u : (size_t)(unsigned)i * (size_t)(unsigned)g.cf.stride));
#else
    const unsigned char* rec = g.cull + (size_t)(unsigned)i * (size_t)kCullStride;
#endif
    CullRaw r;
#if HARE_CULL32
    r.w0 = *reinterpret_cast<const uint4*>(rec);
    r.w1 = *reinterpret_cast<const float4*>(rec + 16);
    r.w2 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (wide) r.w2 = *reinterpret_cast<const float4*>(rec + 32);
#else
    r.c0 = *reinterpret_cast<const double2*>(rec);
    r.r1 = *reinterpret_cast<const uint4*>(rec + 16);
    r.fb = *reinterpret_cast<const float4*>(rec + 32);
#endif
    return r;
}
template <class Args>
__device__ __forceinline__ CullRay cull_ray(const Args& g, double ox, double oy, double oz, double dx, double dy, double dz)
{
    CullRay r;
#if HARE_CULL32
    r.ox = (float)(ox - g.cf.org[0]);
    r.oy = (float)(oy - g.cf.org[1]);
    r.oz = (float)(oz - g.cf.org[2]);
    r.err = __builtin_fmaf(2.3841858e-07f /* 2^-22 */, fabsf(r.ox) + fabsf(r.oy) + fabsf(r.oz), g.cf.err0);
#else
    r.ox = ox; r.oy = oy; r.oz = oz;
    (void)g;
#endif
    r.dfx = (float)dx; r.dfy = (float)dy; r.dfz = (float)dz;
    r.dm = fabsf(r.dfx) + fabsf(r.dfy) + fabsf(r.dfz);
    return r;
}
template <int Q = -1, class Args>
__device__ __forceinline__ bool cull_test(const Args& g, const CullRay& r, const CullRaw& c)
{
#if HARE_CULL32
    const float qx = (float)(c.w0.x & 0x1FFFFFu);
    const float qy = (float)((c.w0.x >> 21) | ((c.w0.y & 0x3FFu) << 11));
    const float qz = (float)(c.w0.y >> 10);
    const float tvx = __builtin_fmaf(-qx, g.cf.step[0], r.ox);
    const float tvy = __builtin_fmaf(-qy, g.cf.step[1], r.oy);
    const float tvz = __builtin_fmaf(-qz, g.cf.step[2], r.oz);
    const float e1[3] = {__uint_as_float(c.w0.z), __uint_as_float(c.w0.w), c.w1.x}, e2[3] = {c.w1.y, c.w1.z, c.w1.w};
    bool miss = cull_fp32(tvx, tvy, tvz, r.dfx, r.dfy, r.dfz, r.dm, e1, e2, r.err);
    if (Q != 0) {
        // a quadrilateral: certain to miss only when the second triangle (v0, v2, v3) is certain to be missed too
        if ((Q == 1 || g.cf.stride == 48) && c.w2.w != 0.0f) {
            const float e3[3] = {c.w2.x, c.w2.y, c.w2.z};
            miss = miss & cull_fp32(tvx, tvy, tvz, r.dfx, r.dfy, r.dfz, r.dm, e2, e3, r.err);
        }
    }
    return miss;
#else
    (void)g;
    const double v0z = __hiloint2double((int)c.r1.y, (int)c.r1.x);
    const float e1[3] = {__uint_as_float(c.r1.z), __uint_as_float(c.r1.w), c.fb.x}, e2[3] = {c.fb.y, c.fb.z, c.fb.w};
    return cull_fp32((float)(r.ox - c.c0.x), (float)(r.oy - c.c0.y), (float)(r.oz - v0z), r.dfx, r.dfy, r.dfz, r.dm, e1, e2);
#endif
}
#endif

struct KdNodeRec {             // 80 bytes
    double bmin[3], bmax[3];
    double split;
    int32_t axis;
    int32_t left, right;       // -1,-1 leaf
    int32_t item_start, item_count;
    int32_t pad;
};

// hare_cost_order (order_kernels.hip): window of consecutive rays that is ordered, bins of the counting sort, threads per window
constexpr int kOrderWindow = 4096;
constexpr int kOrderBins = 512;
constexpr int kOrderThreads = 1024;
constexpr long long kOrderMinRays = 1572864;      // "voxel_order" 1: batches of primary rays from this size (launch.cpp)

// K3d (hare_kdtree_dense, kdtree_dense.hip): the device copy of a kd-tree node, ONE 128-byte cache line, per topology (device_scene.cpp: make_kd_dev_nodes).
// What a visit needs and nothing else: the split, the node's box on the two axes that are NOT the split axis (what KDTree.cs:249-353 compares
// the crossing point with; ascending axis order), the children, a leaf's list -- and the tight boxes of BOTH children's subtrees
// ({x0,y0,z0,x1,y1,z1}, as make_tight_boxes rounds them: outwards), so that the visit decides for both children whether to push them.
struct KdDevNode {
    double split;
    double bb[4];              // {min_b, max_b, min_c, max_c}: axis 0 -> (y, z), 1 -> (x, z), 2 -> (x, y)
    int32_t axis;              // 0, 1, 2; -1: a leaf
    int32_t left, right;
    int32_t item_start, item_count;
    int32_t empty;             // bit 0 / 1: the left / right child's subtree lists no polygon at all (never pushed: popping it has no effect)
    float tl[6], tr[6];        // tight boxes of the left / right child's subtree
    int32_t pad[4];
};
static_assert(sizeof(KdDevNode) == 128, "kd device node: one cache line");
#ifndef HARE_K3D_PEND
#define HARE_K3D_PEND 1            // survivors a lane may hold before it waits for the exact phase.  Swept on the hall at 1M rays: 1 / 2 / 3 / 4 / 6:
                                   // 667 / 650 / 647 / 586 / 564 Mrays/s (shoebox 1 320 / 1 297 / 1 292; 262k rays 367 / 345 / 347) -- a kd leaf holds few
                                   // polygons and the walk behind a survivor is mostly pruned by its hit: testing it at once beats walking on without
                                   // (profiles/r05_experiments/k3d_variants*.log)
#endif
#ifndef HARE_K3D_AHEAD
#define HARE_K3D_AHEAD 0            // the dense windows as a pipeline, as K2d's (HARE_K2D_AHEAD): no gain here (four waves per SIMD hide the
                                   // gathers: hall 683 / 678, shoebox 1 334 / 1 340 Mrays/s without / with; at three waves per SIMD -8 % either way;
                                   // profiles/r06_experiments/k3d_windows_pipelined_not_kept.log)
#endif
#ifndef HARE_K3D_WAVES_PER_EU
#define HARE_K3D_WAVES_PER_EU 4
#endif
// dynamic LDS of a 256-lane workgroup of K3d: (depth + 2) stack entries of 8 bytes per lane, the pending survivors, a 64-word table per wave
constexpr unsigned kd_dense_lds(int max_depth) { return (unsigned)(max_depth + 2) * 256u * 8u + 256u * 4u * (unsigned)HARE_K3D_PEND + 4u * 64u * 4u; }

struct KdArgs {
    const PolyRec* polys;
    const QuadRec* quads;
    const KdNodeRec* nodes;
    const int32_t* items;
    int32_t n_nodes;
    int32_t max_depth;
    const unsigned char* cull; // as in VoxelArgs (device kernels; null on the host)
    CullFrame cf;
    const float* tight;        // as in OctreeArgs: per node the box of the polygons its subtree lists (device kernels; null on the host)
    double tight_mid[3];
    double tight_rad;
    const KdDevNode* dnodes;   // hare_kdtree_dense: the one-line node records of this topology (null: the kernel is not used)
};

struct BuildArgs {             // Voxel_Grid construction kernels (build_kernels.hip)
    const PolyRec* polys;
    const QuadRec* quads;      // null when the topology is all triangles
    int32_t P;
    int32_t ct;
    double omin[3];            // OBox.Min
    double vd[3];              // VoxelDims
};

// Octree construction on the GPU (build_kernels.hip: hare_ob_count / hare_ob_fill): one workgroup per
// task = one child box x one segment of its parent's polygon list.
struct OctTask {               // 64 bytes
    double bmin[3], bmax[3];   // the child's loose box ("Octree - alt.cs":99-114)
    uint32_t pstart, pcount;   // segment of the parent level's item array
    uint32_t ostart;           // where this task's survivors go in the child level's item array (fill pass)
    uint32_t pad;
};
static_assert(sizeof(OctTask) == 64, "octree task size");

// K1q (voxel_pool.hip): launch geometry shared by the kernel and the host launcher
#ifndef HARE_K1Q_WAVES
#define HARE_K1Q_WAVES 12         // waves per workgroup (one workgroup per CU: 3 waves per SIMD)
#endif
#ifndef HARE_K1Q_SLOTS
#define HARE_K1Q_SLOTS 128        // rays per wave (power of two): 2 per lane
#endif
constexpr int kPoolWaves = HARE_K1Q_WAVES;
constexpr int kPoolSlots = HARE_K1Q_SLOTS;
constexpr int kPoolRing = kPoolSlots <= 64 ? 64 : (kPoolSlots <= 128 ? 128 : 256);   // queue capacity: the power of two >= slots
constexpr int kPoolWaveBytes = kPoolSlots * (6 * 8 + 7 * 4) + 5 * kPoolRing;   // per wave: 6 doubles + 7 words per slot, 5 byte queues
// (origin and FP32 direction in LDS as well -- no ray re-read in the cull phase -- was measured in round 2: +2.7 % at 8 waves, far behind 12 waves)
// the fused bounce build of K1q (hare_voxel_bounce_*): per wave a rearm queue, a cast number per slot, rays / hits per cast
constexpr int kBounceMaxCasts = 16;
constexpr int kPoolBounceExtra = kPoolRing + kPoolSlots + 2 * 4 * kBounceMaxCasts;
static_assert(kPoolBounceExtra % 8 == 0, "K1q bounce block keeps its words aligned");
static_assert(kPoolSlots >= 64 && kPoolSlots <= 256 && kPoolSlots % 2 == 0, "K1q slots: even, 64..256");
static_assert(kPoolWaveBytes % 8 == 0, "K1q: per-wave LDS block keeps the doubles aligned");

// K2q (octree_pool.hip): launch geometry shared by the kernel and the host launcher
#ifndef HARE_K2Q_WAVES
#define HARE_K2Q_WAVES 12
#endif
#ifndef HARE_K2Q_SLOTS
#define HARE_K2Q_SLOTS 128
#endif
constexpr int kOctPoolWaves = HARE_K2Q_WAVES;
constexpr int kOctPoolSlots = HARE_K2Q_SLOTS;
constexpr int kOctPoolRing = kOctPoolSlots <= 64 ? 64 : (kOctPoolSlots <= 128 ? 128 : 256);
constexpr int kOctPoolWaveBytes = kOctPoolSlots * (5 * 8 + 11 * 4) + 4 * kOctPoolRing;   // per wave: 5 doubles + 11 words per slot, 4 byte queues
static_assert(kOctPoolSlots >= 64 && kOctPoolSlots <= 256 && kOctPoolSlots % 2 == 0 && kOctPoolWaveBytes % 8 == 0, "K2q pool geometry");
constexpr int kOctScratchRing = 4;    // launches of one scene's octree pool kernel that may be in flight (each owns a scratch block)

// K2d (hare_octree_dense: K2p's DENSE build, kernels.hip): per workgroup of 256 lanes, behind the frames: HARE_K2D_PEND pending survivors per
// lane (12 bytes each) and a 64-word table per wave
#ifndef HARE_K2D_PEND
#define HARE_K2D_PEND 1            // round 5, re-swept after K3d's sweep said the same: 1 / 2 / 3 survivors per lane: 768 / 747 / 740 Mrays/s at 1M rays, 1 015 / 981 / 956
                                   // at 4M, 314 / 313 / 305 at 262k (profiles/r05_experiments/k2d_pend.log): with one, a lane that holds a survivor is "blocked"
                                   // and the exact phase runs in the same round -- the hit it finds prunes the walk behind it at once
#define HARE_K2D_CAP 128
#define HARE_K2D_EXACT_MIN 24
#endif
#ifndef HARE_K2D_STEPS
#define HARE_K2D_STEPS 2           // pop steps per round of the dense build (K2p: 4); 3 until frames closed with their last child
#define HARE_K2D_REFILL 16
#endif
#ifndef HARE_K2D_STEPS_DRAIN
#define HARE_K2D_STEPS_DRAIN 2     // ... once the tickets are dry
#endif
#ifndef HARE_K2D_POP_MIN
#define HARE_K2D_POP_MIN 1         // lanes that make a second, third ... pop step of a round worth its instructions
#endif
#ifndef HARE_PIN_ARGS
#define HARE_PIN_ARGS 1
#endif
#ifndef HARE_K2D_SKIP_PID
#define HARE_K2D_SKIP_PID 1
#endif
#ifndef HARE_K2D_AHEAD
#define HARE_K2D_AHEAD 1           // the dense passes as a pipeline: list entries two windows ahead, pre-cull records one (round 6; with three
                                   // waves per SIMD: hall 1M rays 773 -> 808 Mrays/s, 4M 1 024 -> 1 058, cathedral 550 -> 601)
#endif
constexpr unsigned kOctDenseExtra = 256u * 12u * (unsigned)HARE_K2D_PEND + 4u * 64u * 4u;

// K2g (octree_group.hip): eight lanes per ray, eight rays per wave.  Per group in LDS: the top kGroupStack entries of the ray's
// stack (24 bytes: clamped interval + the node's child / list words) and kGroupPend pending survivors (16 bytes each).
#ifndef HARE_K2G_STACK
#define HARE_K2G_STACK 32
#endif
#ifndef HARE_K2G_WAVES_PER_EU
#define HARE_K2G_WAVES_PER_EU 4
#endif
constexpr int kGroupStack = HARE_K2G_STACK;
constexpr int kGroupPend = 16;
constexpr int kGroupBytes = kGroupStack * 24 + kGroupPend * 16;
constexpr int kGroupWaveBytes = 8 * kGroupBytes;

// Scratch of ONE persistent launch in flight, in device memory (the scene keeps a ring of kLaunchSlots of them):
//   ticket   next-ray ticket the waves draw from with atomicAdd
//   done     waves that have finished, counted per blockIdx % 8 (the workgroups that share an XCD), and in done[8] the groups that
//            have: 8 + 1 addresses instead of one keep the ~4000 end-of-kernel atomics from serialising on one word
//   acc      {rays, hits} batch counters, sharded 64 ways (wave index % 64); the last wave of the grid sums them into the
//            caller's counters
struct LaunchSlotMem {
    unsigned int ticket;
    unsigned int done[9];
    unsigned int oct_tail_count, oct_tail_next, oct_tail_done;      // K2p -> K2t hand-over (octree_coop.hip); K2t's last wave zeroes them
    unsigned int pad[3];
    unsigned long long acc[128];
};
static_assert(sizeof(LaunchSlotMem) == 64 + 1024, "launch slot layout (kernels index it by word)");
constexpr unsigned kLaunchSlots = 64;     // launches of one scene that may be in flight; a 65th waits for the first (launch.cpp)

// One ray a K2p wave handed to the cooperative tail kernel K2t (octree_coop.hip): the state of its depth-first walk.  A record is
// 64 bytes of scalars followed by (levels) frames of 20 bytes, padded to 16: kOctTailHead + 20 * levels rounded up.
struct OctTailRec {
    uint32_t ray;
    int32_t lvl;               // top frame (-1: none open)
    int32_t q, qe;             // the current leaf's remaining entries items[q .. qe)
    double leaf_ca;            // that leaf's entry parameter (nodeTmin)
    double closestT, bu, bv;   // the hit so far
    int32_t pid, hit;
    double pad;
};
static_assert(sizeof(OctTailRec) == 64, "octree tail record head");
constexpr int kOctTailHead = 64;
#ifndef HARE_K2P_TAIL_MAX
#define HARE_K2P_TAIL_MAX 16      // a drained K2p wave hands over its rays once this few are left and they have outlived the rest of
                                  // the batch by HARE_K2P_TAIL_PATIENCE rounds (kernels.hip); swept at C3: (4, 0) 2.82 ms, (8, 64) 2.68,
                                  // (16, 64) 2.69, (16, 96) 2.95, (32, 160) 3.07; without the hand-over 2.94
#endif
constexpr int kOctTailMax = HARE_K2P_TAIL_MAX;
#ifndef HARE_K2P_TAIL_PATIENCE
#define HARE_K2P_TAIL_PATIENCE 64
#endif
#ifndef HARE_K2T_GROUP
#define HARE_K2T_GROUP 64         // K2t: lanes per handed-over ray: 64 = a whole wave (8 / 16 / 32: measured, slower -- the groups of a wave diverge)
#endif
constexpr int kOctTailGroup = HARE_K2T_GROUP;
constexpr unsigned kOctTailGroupsPerBlock = 256u / (unsigned)kOctTailGroup;     // rays a 256-thread K2t workgroup walks at a time (LDS: frames for each)

struct ShootIO {
    RayRec* rays;              // n; written only with SHOOT_WRITEBACK_ORIGIN
    const int32_t* excl1;      // nullable: poly_origin1 per ray
    const int32_t* excl2;      // nullable: poly_origin2 per ray
    XEventRec* out;            // n
    unsigned long long* ctr;   // nullable: CTR_WORDS counters, atomically accumulated
    unsigned int* work;        // persistent kernels: this launch's LaunchSlotMem (below) -- word 0 is the next-ray ticket.  The slot
                               // is all zero when a launch starts and the launch's last wave leaves it all zero again
                               // (launch_epilogue, kernels.hip): no memset in front of a launch, no reduce kernel behind it
    unsigned long long* prof;  // developer profiling kernel: 17 x u64 phase statistics (else null)
    int64_t n;
    uint32_t flags;
    // scheduling knobs: read only by the developer profiling build (hare_voxel_persist_prof); the
    // production kernels use the tuned values as compile-time constants
    int32_t steps_per_round;   // DDA steps per scheduling round
    int32_t refill_min_idle;   // refill when this many lanes are idle
    int32_t ray_chunk;         // rays drawn per ticket
    int32_t exact_min_parked;  // run the FP64 phase when this many lanes hold a survivor
    int32_t audit_polys;       // hare_cull_audit: polygon count
    // persistent voxel kernel: rays per ticket after each wave's static first chunk.  Small tickets even
    // out the end of a small batch; large ones keep the (chip-wide serialised) ticket atomics rare.
    int32_t ticket_rays;
    // persistent voxel kernel: rays in every wave's static first chunk (a multiple of 32, <= 128): the host shrinks it for
    // batches too small to give every wave of the grid 128 rays -- idle waves cost more than a shorter static share
    int32_t static_rays;
    // occlusion predicate (hare_occluded_*; harness-defined): occluded[i] = Shoot(rays[i]) hits AND that closest
    // hit has t < tmax[i] (tmax null: any hit).  With `occluded` set and `out` null the kernels write only the flag; the
    // hare_*_occl_* kernels then also cut the traversal short where that cannot change the flag (kernels.hip).
    const double* tmax;        // nullable
    int32_t* occluded;         // nullable: n flags
    int32_t coop_tail;         // 1: a drained wave traces its last rays cooperatively (voxel_coop.hip); 0: as lanes of the pool to the end
    int32_t wide_drain;        // 1: K1q spreads a ray's candidates / the voxels ahead of it over several lanes once the tickets are dry and few rays are left
    unsigned char* oct_tail;   // K2p -> K2t / K2g-tail: this launch's hand-over records (waves of the K2p grid x oct_tail_max), null = every lane finishes its own
    int32_t oct_tail_stride;   // bytes per record
    int32_t oct_tail_levels;   // frames per record (= the levels K2p keeps in LDS)
    int32_t oct_tail_max;      // K2p: a drained wave hands its rays over once at most this many are alive (64: all of them, at once) ...
    int32_t oct_tail_patience; // ... and they have outlived the rest of the batch by this many rounds
    // the fused bounce kernels (hare_voxel_bounce_*, voxel_pool.hip): rays, excl1 (and excl2 when given) are WORK arrays there
    int32_t bounce_casts;      // casts per ray (1 .. kBounceMaxCasts)
    XEventRec* out_all;        // nullable: bounce_casts x out_stride records, cast-major: every cast's final X_Event (pre-filled with miss records)
    int64_t out_stride;
    unsigned long long* ctr_casts;   // nullable: bounce_casts counter blocks (rays = rays that started the cast, hits), accumulated
    const uint32_t* order;     // K1q, nullable: the ORDER in which the launch takes the batch's rays -- position k of the static chunks / tickets is ray
                               // order[k] (a permutation of 0 .. n-1; rays, events and exclusions stay where the caller has them)
    // K1q, a cast of the bounce loop behind hare_reflect + hare_live_blocks (launch.cpp; null otherwise):
    const uint32_t* blocks;    // the ascending LIST of the blocks of 64 consecutive rays in which a ray still lives: position k of the static chunks / tickets is
                               // ray blocks[k >> 6] * 64 + (k & 63); blocks in which every ray is retired are not in it and cost the cast nothing
    const uint32_t* blk_words; // [0]: the list's length, in device memory (no host round trip)
    int32_t walk_steps;        // K1q: DDA steps per walk task at most (0: HARE_K1Q_WALK_STEPS); the host's rule by batch size (launch.cpp)
    int32_t hand_walk;         // K1q: 1 = the DDA step loop written by hand (voxel_walk.h), 0 = the compiler's (scene option "voxel_walk")
    unsigned char* oct_spill;  // K2g: stack entries beyond kGroupStack, oct_spill_cap x 24 bytes per group of eight lanes (null: the stack fits LDS)
    int32_t oct_spill_cap;
};

#if defined(__HIPCC__)
// Every kernel argument as a scalar of its OWN (round 6).  The compiler fetches the 720-byte argument block in tuples of 8 and 16 SGPRs and,
// short of SGPRs in these kernels, spills and reloads them AS tuples: a block that needs the rays' pointer reloaded sixteen registers to get
// two (v_readlane each: a tenth of K1q's instructions were such reloads).  One s_mov per field at the top of the kernel cuts each loose from its
// tuple; what is reloaded then is what is used.  K1q: 1 242 -> 357 v_readlane in the code, C2 2 903 -> 2 949 Mrays/s, C5 shard 836 -> 848, 262 144 rays
// +3.5 % (profiles/r06_experiments/pin_args.log).  The tree kernels (K2d, K2g, K3d) and K1p were measured with it and gain nothing: not used there.
#if HARE_PIN_ARGS
template <class T> __device__ __forceinline__ void pin_s(T& x)          // a real move: a copy of a sub-register would be coalesced back into its tuple
{
    T y;
    if constexpr (sizeof(T) == 4) asm volatile("s_mov_b32 %0, %1" : "=s"(y) : "s"(x));
    else asm volatile("s_mov_b64 %0, %1" : "=s"(y) : "s"(x));
    x = y;
}
#else
template <class T> __device__ __forceinline__ void pin_s(T&) {}
#endif
__device__ __forceinline__ void pin_args(VoxelArgs& g)
{
    pin_s(g.polys); pin_s(g.quads); pin_s(g.cells); pin_s(g.items); pin_s(g.occ); pin_s(g.ct); pin_s(g.occ_words); pin_s(g.occ_shift); pin_s(g.occ_cd);
    for (int k = 0; k < 3; ++k) { pin_s(g.omin[k]); pin_s(g.omax[k]); pin_s(g.vd[k]); pin_s(g.cf.org[k]); pin_s(g.cf.step[k]); pin_s(g.cellbox_mid[k]); }
    pin_s(g.cull); pin_s(g.cf.err0); pin_s(g.cf.stride); pin_s(g.cellbox); pin_s(g.cellbox_rad); pin_s(g.bocc); pin_s(g.bocc_nb); pin_s(g.bocc_words);
}
__device__ __forceinline__ void pin_args(ShootIO& io)
{
    pin_s(io.rays); pin_s(io.excl1); pin_s(io.excl2); pin_s(io.out); pin_s(io.ctr); pin_s(io.work); pin_s(io.prof); pin_s(io.n); pin_s(io.flags);
    pin_s(io.steps_per_round); pin_s(io.refill_min_idle); pin_s(io.ray_chunk); pin_s(io.exact_min_parked); pin_s(io.audit_polys);
    pin_s(io.ticket_rays); pin_s(io.static_rays); pin_s(io.tmax); pin_s(io.occluded); pin_s(io.coop_tail); pin_s(io.wide_drain);
    pin_s(io.oct_tail); pin_s(io.oct_tail_stride); pin_s(io.oct_tail_levels); pin_s(io.oct_tail_max); pin_s(io.oct_tail_patience);
    pin_s(io.bounce_casts); pin_s(io.out_all); pin_s(io.out_stride); pin_s(io.ctr_casts); pin_s(io.order); pin_s(io.blocks); pin_s(io.blk_words);
    pin_s(io.walk_steps); pin_s(io.hand_walk); pin_s(io.oct_spill); pin_s(io.oct_spill_cap);
}
#endif

}  // namespace hare
