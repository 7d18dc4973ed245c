// scene.h -- host-side scene object behind the opaque hare_scene* of include/hare_hip.h.
// Product code; nothing from oracle/.
#pragma once
#include <stdint.h>
#include <atomic>
#include <condition_variable>
#include <mutex>
#include <string>
#include <vector>

#include "hare_device.h"
#include "hiprt.h"

namespace hare {

// Occupancy bitmap layout for a grid of ct^3 voxels: the finest power-of-two block size whose bitmap fits the
// 64 KB the persistent kernel stages in LDS.  shift 0 = one bit per voxel (ct <= 80); a coarser bit means "some
// voxel of this block is non-empty" and the kernel then reads the voxel's cell record to find out which.
inline void occ_layout(int ct, int32_t& shift, int32_t& cd, int32_t& words)
{
    for (shift = 0;; ++shift) {
        cd = (ct + (1 << shift) - 1) >> shift;
        const long long bits = (long long)cd * cd * cd;
        words = (int32_t)((bits + 31) / 32);
        if ((long long)((words + 3) / 4) * 16 <= 64 * 1024) return;
    }
}

// Diagnostics and A/B switches of one scene.  Filled ONCE, at hare_scene_create, from the environment (HARE_BUILD always; the
// developer variables HARE_VOXEL_KERNEL, HARE_OCTREE_KERNEL, HARE_TICKET, HARE_TUNE, HARE_K1P_STATIC_RAYS, HARE_K2P_STATIC_RAYS,
// HARE_BATCH_CHUNKS only when HARE_DEV=1), changed afterwards only through hare_scene_set_option: no getenv on any launch path,
// so a host that calls setenv from another thread cannot race with a shoot, and a stray variable cannot change kernel choice.
struct SceneOptions {
    int dev = 0;               // HARE_DEV=1: developer flag bits (timeline, phase profile, cull audit) pass sanitize_flags
    int build_host = 0;        // HARE_BUILD=host: host builders even when a GPU is present (identical output)
    int voxel_kernel = 0;      // 0 = the library's rule, 1 = K1p (persist), 2 = K1q (pool)
    int octree_kernel = 0;     // 0 = the library's rule (K2g below 426k rays, K2d above), 1 = K2p (persist), 2 = K2q (pool), 3 = K2g (group), 4 = K2d (dense)
    int ticket_rays = 0;       // rays per ticket of the persistent kernels (0 = the host's rule)
    int k1p_static_rays = 0;   // static first chunk per wave (0 = the host's rule)
    int k2p_static_rays = 0;
    int bounce_fused = 0;      // 1: the bounce loop of a Voxel_Grid runs as ONE launch where the pool kernel serves (hare_voxel_bounce_*); 0 (default): a launch
                               // per cast -- measured: the hall +2.5 %, the cathedral -0 ... -10 % (a chip that works on all casts at once loses the L2 locality of one band)
    int voxel_tight = 1;       // 1: K1q sends a ray on past an occupied voxel whose polygons it cannot hit (the voxels' tight boxes, device_scene.cpp: upload_cell_boxes); 0: every list the reference scans (A/B)
    int voxel_order_max_rays = 1 << 24;   // entries per block of the order ring reserved with the grid (scene.h: kOrderRing blocks x 4 B each); larger batches run unordered; 0: no ring
    int voxel_order = 1;       // the pool kernel takes a batch's rays, window by window, in the order of their estimated walk length (order_kernels.hip):
                               // 1 (default) batches of primary rays from 1 572 864 rays, 0 never, 2 every batch.  Results never depend on it
    int voxel_tight_max_mb = 0; // budget for those boxes (32 B per voxel and topology), MiB; 0 = none.  Over budget or out of memory: no boxes, same results
    int dev_fail_cellbox_alloc = 0;   // test hook: the boxes' allocation "fails" (tests/test_gpu_tight.py: a build must still succeed)
    int octree_tight = 1;      // 1: K2d / K2p skip a popped node whose subtree's polygons the ray cannot hit (the tight boxes of device_scene.cpp); 0: every node the reference visits (A/B)
    int kdtree_kernel = 0;     // 0 = the library's rule (K3d, hare_kdtree_dense, wherever its node records exist and its LDS fits), 1 = the one-ray-per-lane kernel, 2 = K3d
    int octree_tail = 2;       // what finishes the rays K2p gives up: 0 nothing (every lane finishes its own), 1 K2t (a wave per ray, a wave's last 16), 2 K2g-tail (eight lanes per ray, all of them)
    int k2p_tail_max = 0;      // developer sweeps: hand over once at most this many rays are alive in a wave (0 = the rule) ...
    int k2p_tail_patience = -1; // ... after this many rounds (-1 = the rule)
    int batch_chunks = 0;      // chunks hare_shoot_batch pipelines a batch over (0 = the host's rule)
    int coop_tail = 1;         // 1: a drained wave traces its last rays with all 64 lanes (voxel_coop.hip); 0: as lanes of the pool to the end (A/B)
    int voxel_skip = 0;        // 1: K1q's walk crosses an EMPTY aligned block of 4^3 voxels in one operation -- the exact closed-form skip (voxel_pool.hip); bit-identical
                               // results, and slower than the hand-written step: off by default, kept as a tested option
    int voxel_walk = 1;        // 1: K1q's DDA step loop as written by hand for gfx950 (voxel_walk.h: per-axis updates under EXEC masks); 0: the compiler's loop (A/B)
    int bounce_pack = 1;       // 1: the launch-per-cast bounce loop of a Voxel_Grid lists the blocks of 64 rays in which a ray still lives behind every reflection (one
                               // more one-workgroup launch per cast) and the next cast walks the list: open scenes; 0: off (a closed room saves ~1 %)
    int wide_drain = 1;        // 1: K1q's wide cull / wide walk in the drain of a launch (voxel_pool.hip); 0: the pool's ordinary phases to the end (A/B)
    long long dev_order_ptr = 0;   // developer experiments (a `dev` scene only): a device array of n uint32, the order K1q takes the rays in (ShootIO::order)
    int tune[5] = {0, 0, 0, 0, 0};   // HARE_TUNE: steps,refill,chunk,blocks_per_cu,exact (profiling build; blocks_per_cu: K1p)
};

struct Topo {
    int32_t P = 0;
    std::vector<double> verts;    // P x 12
    std::vector<int32_t> nverts;  // P
    std::vector<double> normals;  // P x 3
    double mn[3], mx[3];          // Topology.Min / Max
    bool has_quads = false;
};

struct VoxelHost {
    bool built = false;
    bool on_device = false;   // lists were produced by the GPU builder
    int32_t ct = 0;
    double omin[3], omax[3], box_dims[3], vd[3];
    double char_step = 0;
    std::vector<std::vector<uint32_t>> start;  // [topo][ct^3 + 1]
    std::vector<std::vector<int32_t>> items;   // [topo][...]
};

struct OctreeHost {
    bool built = false;
    bool built_on_device = false;
    int32_t max_depth = 0, max_polys = 0;
    int32_t id_count = 0;      // polygon ids in the lists are < id_count (= Polygon_Count of the LAST topology, see build_octree)
    std::vector<OctNode> nodes;
    std::vector<int32_t> items;
};

struct KdHost {
    bool built = false;
    int32_t max_depth = 0, max_polys = 0;
    int32_t id_count = 0;      // as OctreeHost::id_count
    int32_t depth_reached = 0;
    std::vector<KdNodeRec> nodes;
    std::vector<int32_t> items;
};

struct DeviceModule {
    hipModule_t mod = nullptr;
    hipFunction_t voxel_tri = nullptr, voxel_quad = nullptr, voxel_count = nullptr;
    hipFunction_t voxel_persist_tri = nullptr, voxel_persist_quad = nullptr;
    hipFunction_t voxel_persist_tri_g = nullptr, voxel_persist_quad_g = nullptr;
    hipFunction_t voxel_pool_tri = nullptr, voxel_pool_quad = nullptr, voxel_pool_tri_g = nullptr, voxel_pool_quad_g = nullptr;
    // counting builds of the production kernels (HARE_SHOOT_COUNT_OWN)
    hipFunction_t voxel_pool_tri_own = nullptr, voxel_pool_quad_own = nullptr, voxel_pool_tri_g_own = nullptr, voxel_pool_quad_g_own = nullptr;
    hipFunction_t octree_dense_own = nullptr;
    hipFunction_t cost_order = nullptr;                                    // order_kernels.hip
    hipFunction_t kdtree_dense = nullptr, kdtree_dense_own = nullptr, kdtree_occl = nullptr;      // K3d (kdtree_dense.hip) and its counting build
    hipFunction_t voxel_bounce_tri = nullptr, voxel_bounce_quad = nullptr, voxel_bounce_tri_g = nullptr, voxel_bounce_quad_g = nullptr, counters_sum = nullptr;
    hipFunction_t octree = nullptr, octree_count = nullptr, octree_persist = nullptr, octree_pool = nullptr, octree_tail = nullptr, octree_group = nullptr, octree_group_tail = nullptr, octree_dense = nullptr;
    hipFunction_t kdtree = nullptr, kdtree_count = nullptr;
    hipFunction_t reflect = nullptr, occlusion = nullptr;
    hipFunction_t voxel_occl_tri = nullptr, voxel_occl_quad = nullptr, voxel_occl_tri_g = nullptr, voxel_occl_quad_g = nullptr, octree_occl = nullptr, octree_occl_any = nullptr;
    hipFunction_t events_pack_slim = nullptr;
    hipFunction_t block_occ = nullptr;                                     // build_kernels.hip: hare_block_occ
    hipFunction_t live_blocks = nullptr;                                   // kernels.hip: hare_live_blocks (the bounce loop's block list)
    hipFunction_t live_count = nullptr, scan_tiles = nullptr, reflect_compact = nullptr, events_fill_miss = nullptr, events_expand = nullptr;
    hipFunction_t cull_audit = nullptr;
    hipFunction_t voxel_persist_prof = nullptr;
    hipFunction_t vb_count = nullptr, vb_fill = nullptr, vb_level_count = nullptr, vb_level_fill = nullptr;
    hipFunction_t scan_block = nullptr, scan_add = nullptr, vb_sort_small = nullptr, vb_sort_block = nullptr, vb_finalize = nullptr, cell_boxes = nullptr;
    hipFunction_t vb_find_big = nullptr, vb_fill_big = nullptr;
    hipFunction_t ob_count = nullptr, ob_fill = nullptr;
    int cu_count = 0;
};

// Makes `device` the calling thread's current HIP device for the guard's lifetime and restores the previous one
// afterwards: a host that juggles several devices (torch device guards, one scene per GPU) must find its current
// device unchanged after any call into the library, and every call must act on the scene's own device.
struct DeviceGuard {
    const HipApi* H;
    int prev = -1;
    bool switched = false;
    DeviceGuard(const HipApi* api, int device) : H(api)
    {
        if (!H) return;
        if (H->GetDevice(&prev) != hipSuccess) { prev = -1; (void)H->GetLastError(); }
        if (prev != device) {
            if (H->SetDevice(device) == hipSuccess) switched = true;
            else (void)H->GetLastError();   // e.g. an ordinal that does not exist: the entry point reports it; the host's
                                            // runtime must not be left with a sticky "last error" (torch checks it)
        }
    }
    ~DeviceGuard()
    {
        if (H && switched && prev >= 0) (void)H->SetDevice(prev);
    }
    DeviceGuard(const DeviceGuard&) = delete;
    DeviceGuard& operator=(const DeviceGuard&) = delete;
};

struct HostMirror;   // host_trace.cpp: the scene as hare_shoot_one reads it

struct Scene {
    int device = 0;
    std::vector<Topo> topos;
    VoxelHost vox;
    OctreeHost oct;
    KdHost kd;

    // device residents (all on `device`)
    std::vector<void*> d_polys;                  // per topo: PolyRec[P]
    std::vector<void*> d_cull;                   // per topo: the pre-cull's dense records, kCullStride bytes per polygon (hare_device.h)
    std::vector<CullFrame> cull_frames;          // per topo: how those records decode
    std::vector<void*> d_quads;                  // per topo: QuadRec[P] or null (all triangles)
    std::vector<void*> d_cells, d_items, d_occ;  // per topo (voxel)
    std::vector<void*> d_bocc;                   // per topo: occupancy of the aligned 4^3 blocks of voxels (option "voxel_skip"; device_scene.cpp: upload_block_occ); null = none
    int32_t bocc_nb = 0, bocc_words = 0;
    std::vector<void*> d_cellbox;                // per topo: the voxels' tight boxes, 8 floats per voxel (device_scene.cpp: upload_cell_boxes); null = none
    double cellbox_mid[3] = {0, 0, 0}, cellbox_rad = -1;   // ray origins they may be used for: |o - mid|_inf <= rad
    int32_t occ_words = 0;                       // words of the occupancy bitmap the persistent kernel stages in LDS
    int32_t occ_shift = 0, occ_cd = 0;           // bitmap resolution: one bit per (2^occ_shift)^3 voxels, occ_cd blocks per axis
    void* d_oct_nodes = nullptr;
    void* d_oct_items = nullptr;
    std::vector<void*> d_oct_tight;   // per topology: the box of the polygons every node's subtree lists, 8 floats per node (device_scene.cpp: make_tight_boxes); null = none
    double oct_tight_mid[3] = {0, 0, 0}, oct_tight_rad = -1;      // ray origins the boxes may be used for: |o - mid|_inf <= rad
    void* d_kd_nodes = nullptr;
    std::vector<void*> d_kd_tight;    // as d_oct_tight, for the kd-tree's nodes
    std::vector<void*> d_kd_dev;      // per topology: the one-line node records of hare_kdtree_dense (KdDevNode; device_scene.cpp: upload_kd_dev_nodes); null = none
    double kd_tight_mid[3] = {0, 0, 0}, kd_tight_rad = -1;
    void* d_kd_items = nullptr;
    void* d_work = nullptr;                      // LaunchSlotMem[kLaunchSlots]: scratch of the persistent launches in flight
    std::atomic<unsigned> work_slot{0};
    // host side of the launch-slot ring: a slot is re-used by every kLaunchSlots-th launch, which must not start before the
    // launch that used it last has finished (they may be on different streams): each launch records an event behind itself
    // and waits for the slot's previous one; `mu` keeps wait + launch + record of one slot together
    struct LaunchSlot {
        std::mutex mu;
        hipEvent_t ev = nullptr;
        bool used = false;
    };
    LaunchSlot slots[kLaunchSlots];
    SceneOptions opt;
    // K2p -> K2t hand-over records (octree_coop.hip): a ring of blocks, one per K2p launch in flight; a block comes round after
    // kOctTailRing launches and its previous user must have finished (event), wait + launches + record under the mutex
    static constexpr int kOctTailRing = 8;
    void* d_oct_tail = nullptr;
    size_t oct_tail_block_bytes = 0;
    unsigned oct_tail_seq = 0;
    hipEvent_t oct_tail_ev[kOctTailRing] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    bool oct_tail_used[kOctTailRing] = {false, false, false, false, false, false, false, false};
    std::mutex oct_tail_mu;
    // the ray orders of the pool kernel (order_kernels.hip): a ring of kOrderRing blocks of order_cap uint32 each, ONE allocation, RESERVED when
    // the grid goes to the device (device_scene.cpp: reserve_order_ring; option "voxel_order_max_rays", 16 Mi rays = 64 MiB a block by default)
    // -- a shoot never allocates, frees or synchronises (round 5 grew the blocks inside hare_shoot_device: ADVICE).  A launch takes the next
    // block (atomic counter), waits ON ITS STREAM for the event the block's previous user recorded, enqueues the order pass and the shoot and
    // records the event again, under the block's own mutex (held for those four enqueues only).  A batch larger than a block, a stream that
    // is being captured, or a failed reservation: the cast runs in the caller's order -- the same events.  More than kOrderRing ordered
    // launches in flight on one scene queue behind each other's blocks (stream waits, never a host wait).
    static constexpr int kOrderRing = 4;
    void* d_order = nullptr;
    size_t order_cap = 0;                                   // uint32 entries per block (0: no ring -- the order pass never runs)
    hipEvent_t order_ev[kOrderRing] = {nullptr, nullptr, nullptr, nullptr};
    bool order_used[kOrderRing] = {false, false, false, false};
    std::atomic<unsigned> order_seq{0};
    std::mutex order_blk_mu[kOrderRing];

    // staging for hare_shoot_batch: a small pool of contexts (device buffers + the three streams a batch is pipelined
    // over), so that host threads calling on one scene run side by side instead of queueing on one mutex; `mu` guards
    // the one-time device set-up and the hand-out of contexts only, never a transfer or a kernel
    // device buffers of one hare_bounce_batch call in flight (bounce.cpp): the rays and their exclusions double-buffered
    // (a packed copy is written while the previous one is read), events double-buffered (the download of one cast runs
    // beside the next cast), the map back to the caller's order, the expanded events of a packed cast, tile counts
    struct BounceBuf {
        void* rays[2] = {nullptr, nullptr};
        void* excl[2] = {nullptr, nullptr};
        void* excl2 = nullptr;
        void* idx[2] = {nullptr, nullptr};
        void* ev[2] = {nullptr, nullptr};
        void* full = nullptr;
        void* tiles = nullptr;
        void* ctr = nullptr;            // hare_counters per cast (+ one word for the packed count)
        int64_t cap = 0;
        int32_t ctr_cap = 0;
        hipStream_t copy_st = nullptr;
    };
    struct BatchCtx {
        BounceBuf bounce;
        hipStream_t st[16] = {};
        void* d_rays = nullptr;
        void* d_e1 = nullptr;
        void* d_e2 = nullptr;
        void* d_out = nullptr;
        void* d_ctr = nullptr;
        void* d_slim = nullptr;      // HARE_SHOOT_SLIM_EVENTS: packed result records (cap x 32 B)
        void* d_tmax = nullptr;      // occlusion queries: t_max per ray, flags per ray
        void* d_occ = nullptr;
        int64_t cap = 0;
        int64_t occ_cap = 0;
        bool busy = false;
    };
    static constexpr int kBatchCtx = 4;
    std::mutex mu;
    std::condition_variable cv;
    BatchCtx ctx[kBatchCtx];
    hipStream_t stream = nullptr;                // the scene's own stream (builders)

    const DeviceModule* module = nullptr;

    // K2q scratch: one block per launch in flight (ring), each guarded by an event recorded behind the launch that used it
    void* d_oct_scratch[kOctScratchRing] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t oct_scratch_ev[kOctScratchRing] = {nullptr, nullptr, nullptr, nullptr};
    size_t oct_scratch_bytes = 0;
    std::atomic<unsigned> oct_scratch_next{0};
    std::mutex oct_scratch_mu;

    // host mirror for hare_shoot_one (single-ray callers): built on first use, read-only afterwards
    std::mutex mirror_mu;
    std::atomic<HostMirror*> mirror{nullptr};
    int32_t oct_levels = 0;                      // interior levels the octree actually has (frames the kernels need)
};
void free_host_mirror(Scene& s);             // host_trace.cpp
void make_poly_records(const Topo& T, std::vector<PolyRec>& rec, std::vector<QuadRec>& quads);   // device_scene.cpp
void make_cull_records(const Topo& T, const std::vector<PolyRec>& rec, std::vector<unsigned char>& dense, CullFrame& cf);   // device_scene.cpp

// error plumbing (thread-local message)
void set_error(const std::string& msg);
const char* last_error();

// the launcher behind every shoot entry point (launch.cpp)
// d_occ != null: also (d_out != null) or only (d_out == null) the occlusion flags against d_tmax (nullable: any hit)
int shoot_device_impl(Scene& s, const HipApi* H, int32_t kind, int32_t top, int64_t n, void* d_rays, const void* d_e1, const void* d_e2,
                      uint32_t flags, void* d_out, void* d_ctr, hipStream_t st, const void* d_tmax = nullptr, void* d_occ = nullptr,
                      const struct ShootExtra* extra = nullptr);
// what this library's own loops add to a cast (never a caller): internal flag bits (SHOOT_RETIRED_SILENT) and the bounce loop's block list
struct ShootExtra {
    uint32_t internal_flags = 0;
    // the bounce loop's block list (ShootIO, hare_device.h); only the pool kernel K1q reads it, other kernels cast all n rays
    const uint32_t* blocks = nullptr;
    const uint32_t* blk_words = nullptr;
};
int bounce_device_impl(Scene& s, const HipApi* H, int32_t kind, int32_t top, int64_t n, void* d_rays, const void* d_e1, const void* d_e2,
                       int32_t casts, uint32_t flags, void* d_work, void* d_all, void* d_last, void* d_ctr, void* d_ctr_casts, hipStream_t st);
uint32_t sanitize_flags(const Scene& s, uint32_t flags);
int dev_free(const HipApi* H, void*& p);
void free_bounce_buffers(const HipApi* H, Scene& s);          // bounce.cpp

// device plumbing (device_scene.cpp) shared with api.cpp, launch.cpp and build_gpu.cpp
const HipApi* api_or_err();
int ensure_device(Scene& s, const HipApi*& H);
int upload(const HipApi* H, void** dst, const void* src, size_t bytes);
int upload_polys(Scene& s, const HipApi* H);
int upload_voxel(Scene& s, const HipApi* H);
int launch(const HipApi* H, hipFunction_t f, unsigned grid, unsigned block, unsigned lds, hipStream_t st, void** args);
int hip_fail(const HipApi* H, hipError_t e, const char* what);

// grid geometry shared by the host and GPU builders (Voxel_Grid.cs:52-90)
void voxel_grid_bounds(const Scene& s, VoxelHost& g);
void voxel_grid_set_ct(VoxelHost& g, int32_t ct);

// GPU builders (build_gpu.cpp); *used = false means "not applicable, use the host builder"
int gpu_build_voxel_fixed(Scene& s, const HipApi* H, int32_t domain, bool* used);
int gpu_build_voxel_adaptive(Scene& s, const HipApi* H, int32_t max_domain, int32_t avg_polys, bool* used);
int gpu_build_octree(Scene& s, const HipApi* H, int32_t max_depth, int32_t max_polys, bool* used);

// Budgets of one octree (host and GPU builder alike; HARE_E_NOMEM beyond them, before the machine is exhausted).  The reference
// has none and would run until the process dies: its child boxes are padded by an ABSOLUTE 0.1 m ("Octree - alt.cs":99-111),
// so once nodes are smaller than ~0.4 m every polygon lands in all eight children and the tree grows 8x per level whatever the
// polygon count -- a 60-triangle soup at maxDepth 17 asks for 8^17 nodes.
constexpr size_t kOctMaxNodes = (size_t)1 << 24;          // 64 B each on the device: 1 GB
constexpr size_t kOctMaxItems = (size_t)1 << 28;          // polygon-list entries alive at once: 1 GB
int octree_budget_error(const char* what);                // sets the message, returns HARE_E_NOMEM (build_host.cpp)

// builders (host); return HARE_* codes
int build_voxel_fixed(Scene& s, int32_t domain);
int build_voxel_adaptive(Scene& s, int32_t max_domain, int32_t avg_polys);
int build_octree(Scene& s, int32_t max_depth, int32_t max_polys);
int octree_check_args(const Scene& s, int32_t max_depth, int32_t max_polys);
void octree_root_box(const Topo& T, double bmin[3], double bmax[3]);                      // "Octree - alt.cs":63-88
void octree_child_box(const double nmin[3], const double nmax[3], int i, double cmin[3], double cmax[3]);   // :96-114
int build_kdtree(Scene& s, int32_t max_depth, int32_t max_polys);
int32_t octree_levels(const OctreeHost& o);   // device_scene.cpp

// host helpers
void polygon_normals(const double* verts, const int32_t* nverts, int32_t P, double* out);
void topology_bounds(const double* verts, const int32_t* nverts, int32_t P, double mn[3], double mx[3]);
int topology_ingest(const double* soup, const int32_t* nverts, int32_t P, double* verts_out, int32_t* corner_vertex,
                    std::vector<double>& vertices);   // ingest.cpp

}  // namespace hare

// the opaque handle of include/hare_hip.h
struct hare_scene : hare::Scene {};
