// launch.cpp -- which kernel serves a batch and how it is launched: the launch-slot ring, the scene options, the kernel choice
// (ONE function for the launcher and for hare_shoot_kernel_name), hare_shoot_device / hare_bounce_device behind the C-ABI.  (Split from api.cpp in round 5.)
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <cmath>
#include <limits>
#include <map>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "../../include/hare_hip.h"
#include "scene.h"
#include "launch.h"

namespace hare {

// One persistent launch (K1p, K1q, K2p, K2q) on the next slot of the scene's launch-slot ring.  The slot holds the launch's
// ticket word, done counters and counter shards (LaunchSlotMem); the launch's own last wave leaves it zeroed, so nothing is
// enqueued in front of the kernel or behind it.  A slot comes round again after kLaunchSlots launches, possibly on another
// stream: the new launch waits for the event the slot's previous launch recorded behind itself (a no-op when that launch has
// finished, which is the rule), so a 65th launch in flight waits for the first instead of sharing its ticket word.  The
// slot's mutex keeps wait + launch + record together when several host threads launch on one scene.
// args[1] must point to `io`.
// Octree launches get a block of the scene's octree scratch ring (one block per launch in flight, event-ordered):
//   tail_levels > 0     a K2p launch: hand-over records for the rays its waves give up (tail_max per wave), followed on the same
//                       stream, inside the slot's lock, by the tail kernel -- K2t (octree_coop.hip: a wave per ray, the last few
//                       rays of a wave) or K2g-tail (octree_group.hip: eight lanes per ray, ALL the rays a wave still holds when
//                       the tickets run dry)
//   spill_entries > 0   K2g's stack entries beyond what LDS holds (24 bytes x entries per group of eight lanes), for the K2g
//                       launch itself or for the K2g-tail behind K2p
struct OctScratch {
    int tail_levels = 0;
    int tail_max = 0, tail_patience = 0;
    bool group_tail = false;
    int spill_entries = 0;
};
// A stream under capture (hipStreamBeginCapture) takes kernel launches but not this library's cross-launch event protocol in a form a
// graph replay could honour twice: the order pass is skipped there (the cast runs in the caller's order: same events).
static bool stream_is_capturing(const HipApi* H, hipStream_t st)
{
    if (!H->StreamIsCapturing || !st) return false;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (H->StreamIsCapturing(st, &cs) != hipSuccess) { (void)H->GetLastError(); return false; }
    return cs != hipStreamCaptureStatusNone;
}
int launch_on_slot(Scene& s, const HipApi* H, hipFunction_t f, unsigned grid, unsigned block, unsigned lds, hipStream_t st, ShootIO& io,
                   void** args, bool coop_tail = false, const OctScratch& oc = OctScratch())
{
    const unsigned idx = s.work_slot.fetch_add(1) % kLaunchSlots;
    Scene::LaunchSlot& sl = s.slots[idx];
    std::lock_guard<std::mutex> lk(sl.mu);
    io.work = reinterpret_cast<unsigned int*>(static_cast<LaunchSlotMem*>(s.d_work) + idx);
    io.coop_tail = (coop_tail && s.opt.coop_tail) ? 1 : 0;
    io.wide_drain = s.opt.wide_drain ? 1 : 0;
    io.hand_walk = s.opt.voxel_walk ? 1 : 0;
    io.oct_tail = nullptr;
    io.oct_spill = nullptr;
    io.oct_spill_cap = 0;
    const hipFunction_t tail_fn = oc.group_tail ? s.module->octree_group_tail : s.module->octree_tail;
    const bool with_tail = oc.tail_levels > 0 && s.opt.coop_tail && tail_fn != nullptr;
    const unsigned cus = (unsigned)std::max(1, s.module->cu_count);
    // the tail kernel's grid: K2t a wave per ray of a typical hand-over; K2g-tail a chip full of groups (waves without a record end at once)
    const unsigned tgrid = !with_tail ? 0u : (oc.group_tail ? cus * (unsigned)HARE_K2G_WAVES_PER_EU : std::max(1u, std::min(grid, 4u * cus)));
    const unsigned spill_groups = oc.spill_entries <= 0 ? 0u : (with_tail && oc.group_tail ? tgrid * 4u * 8u : grid * (block / 64u) * 8u);
    const bool with_spill = spill_groups > 0 && (!oc.tail_levels || (with_tail && oc.group_tail));
    std::unique_lock<std::mutex> tail_lk(s.oct_tail_mu, std::defer_lock);
    int tail_ring = -1;
    if (with_tail || with_spill) {
        const size_t stride = !with_tail ? 0 : (((size_t)kOctTailHead + 20u * (size_t)oc.tail_levels + 15u) & ~(size_t)15u);
        const size_t rec_bytes = !with_tail ? 0 : (((size_t)grid * (block / 64u) * (size_t)oc.tail_max * stride + 255u) & ~(size_t)255u);
        const size_t spill_bytes = with_spill ? (size_t)spill_groups * (size_t)oc.spill_entries * 24u : 0;
        const size_t need = rec_bytes + spill_bytes;
        tail_lk.lock();                      // held until the launch (and the tail behind it) is enqueued and the block's event recorded
        if (need > s.oct_tail_block_bytes) {
            // larger blocks: only when reserve_oct_scratch could not allocate at build time (it sizes the ring for the largest launch
            // this tree can get); launches in flight may still use the old ones
            if (s.d_oct_tail) {
                HIP_TRY(H->DeviceSynchronize());
                dev_free(H, s.d_oct_tail);
            }
            s.oct_tail_block_bytes = 0;
            HIP_TRY(H->Malloc(&s.d_oct_tail, (size_t)Scene::kOctTailRing * need));
            s.oct_tail_block_bytes = need;
            for (bool& u : s.oct_tail_used) u = false;
        }
        tail_ring = (int)(s.oct_tail_seq++ % (unsigned)Scene::kOctTailRing);
        if (!s.oct_tail_ev[tail_ring]) HIP_TRY(H->EventCreateWithFlags(&s.oct_tail_ev[tail_ring], hipEventDisableTiming));
        if (s.oct_tail_used[tail_ring]) HIP_TRY(H->StreamWaitEvent(st, s.oct_tail_ev[tail_ring], 0));
        unsigned char* blockp = static_cast<unsigned char*>(s.d_oct_tail) + (size_t)tail_ring * s.oct_tail_block_bytes;
        if (with_tail) {
            io.oct_tail = blockp;
            io.oct_tail_stride = (int32_t)stride;
            io.oct_tail_levels = oc.tail_levels;
            io.oct_tail_max = oc.tail_max;
            io.oct_tail_patience = oc.tail_patience;
        }
        if (with_spill) {
            io.oct_spill = blockp + rec_bytes;
            io.oct_spill_cap = oc.spill_entries;
        }
    }
    if (!sl.ev) HIP_TRY(H->EventCreateWithFlags(&sl.ev, hipEventDisableTiming));
    if (sl.used) HIP_TRY(H->StreamWaitEvent(st, sl.ev, 0));
    int rc = launch(H, f, grid, block, lds, st, args);
    if (rc) return rc;
    // From here on a kernel is enqueued that will use the slot (and the scratch block).  If a later step fails, the slot must
    // not come round again in the state that kernel leaves it in with nothing to wait for: drain the stream, put the slot back to
    // the all-zero state the kernels start from, and forget the events that were never recorded.
    auto fail_after_launch = [&](int code) {
        const std::string msg = last_error();
        (void)H->StreamSynchronize(st);
        (void)H->MemsetAsync(static_cast<LaunchSlotMem*>(s.d_work) + idx, 0, sizeof(LaunchSlotMem), st);
        (void)H->StreamSynchronize(st);
        sl.used = false;
        if (tail_ring >= 0) s.oct_tail_used[tail_ring] = false;
        set_error(msg);
        return code;
    };
    if (with_tail) {
        // the tail reads the count K2p left behind.  K2t (HARE_K2T_GROUP = 64): a whole wave per handed-over ray, four rays per workgroup,
        // LDS one 20-byte frame per level for each; K2g-tail: the groups' stacks and pending lists, as K2g
        const unsigned tlds = oc.group_tail ? 4u * (unsigned)kGroupWaveBytes : kOctTailGroupsPerBlock * 20u * (unsigned)oc.tail_levels;
        rc = launch(H, tail_fn, tgrid, 256, tlds, st, args);
        if (rc) return fail_after_launch(rc);
    }
    if (tail_ring >= 0) {
        if (hipError_t e = H->EventRecord(s.oct_tail_ev[tail_ring], st); e != hipSuccess) return fail_after_launch(hip_fail(H, e, "hipEventRecord"));
        s.oct_tail_used[tail_ring] = true;
    }
    if (hipError_t e = H->EventRecord(sl.ev, st); e != hipSuccess) return fail_after_launch(hip_fail(H, e, "hipEventRecord"));
    sl.used = true;
    return HARE_OK;
}

// Public flag bits; the developer bits (0x1000 round trace, 0x2000 timeline, 0x4000 phase profile, 0x8000 cull audit: they write past the
// counters block into a buffer the developer tools size for it, or leave `out` unwritten) only pass on a scene whose
// `dev` option is set (HARE_DEV=1 when the scene was created, or hare_scene_set_option), so a stray bit from a caller can
// never reach a kernel.
// HARE_SHOOT_BOUNCE_LOOP is NOT among them: it is a question to hare_shoot_kernel_name (which reads it from the raw flags), never a mode
// of a cast, and must not travel into ShootIO::flags where a device-side bit 32 would one day collide with it (ADVICE, round 4).
constexpr uint32_t kFlagAnyHit = 0x40000u;         // internal (set below, never by a caller): a flags-only occlusion query without t_max
constexpr uint32_t kPublicFlags = HARE_SHOOT_WRITEBACK_ORIGIN | HARE_SHOOT_COUNT_WORK | HARE_SHOOT_SIMPLE_KERNEL | HARE_SHOOT_RETIRED_RAYS | HARE_SHOOT_SLIM_EVENTS |
                                 HARE_SHOOT_COUNT_OWN;
static_assert((kPublicFlags & HARE_SHOOT_BOUNCE_LOOP) == 0, "the kernel-name query bit never reaches a kernel");
uint32_t sanitize_flags(const Scene& s, uint32_t flags)
{
    return flags & (kPublicFlags | (s.opt.dev ? 0xF000u : 0u));
}

// The scene's options as the environment gives them; called once per scene, from hare_scene_create (single-caller by contract).
void read_env_options(SceneOptions& o)
{
    auto on = [](const char* e) { return e && *e && *e != '0'; };
    o = SceneOptions();
    if (const char* b = getenv("HARE_BUILD")) o.build_host = strcmp(b, "host") == 0;
    o.dev = on(getenv("HARE_DEV"));
    if (!o.dev) return;            // everything below is a developer override: ignored unless the process opted in
    if (const char* k = getenv("HARE_VOXEL_KERNEL")) o.voxel_kernel = strcmp(k, "persist") == 0 ? 1 : (strcmp(k, "pool") == 0 ? 2 : 0);
    if (const char* k = getenv("HARE_OCTREE_KERNEL")) o.octree_kernel = strcmp(k, "persist") == 0 ? 1 : (strcmp(k, "pool") == 0 ? 2 : (strcmp(k, "group") == 0 ? 3 : (strcmp(k, "dense") == 0 ? 4 : 0)));
    if (const char* t = getenv("HARE_OCTREE_TAIL")) o.octree_tail = atoi(t);
    if (const char* k = getenv("HARE_KDTREE_KERNEL")) o.kdtree_kernel = strcmp(k, "simple") == 0 ? 1 : (strcmp(k, "dense") == 0 ? 2 : 0);
    if (const char* t = getenv("HARE_OCTREE_TIGHT")) o.octree_tight = atoi(t) != 0;
    if (const char* t = getenv("HARE_VOXEL_TIGHT")) o.voxel_tight = atoi(t) != 0;
    if (const char* t = getenv("HARE_VOXEL_WALK")) o.voxel_walk = atoi(t) != 0;
    if (const char* t = getenv("HARE_VOXEL_SKIP")) o.voxel_skip = atoi(t) != 0;
    if (const char* t = getenv("HARE_BOUNCE_PACK")) o.bounce_pack = atoi(t) != 0;
    if (const char* t = getenv("HARE_VOXEL_ORDER")) o.voxel_order = std::max(0, std::min(2, atoi(t)));
    if (const char* t = getenv("HARE_VOXEL_ORDER_MAX_RAYS")) o.voxel_order_max_rays = std::max(0, atoi(t));
    if (const char* t = getenv("HARE_VOXEL_TIGHT_MAX_MB")) o.voxel_tight_max_mb = std::max(0, atoi(t));
    if (const char* t = getenv("HARE_FAIL_CELLBOX_ALLOC")) o.dev_fail_cellbox_alloc = atoi(t) != 0;
    if (const char* t = getenv("HARE_BOUNCE_FUSED")) o.bounce_fused = atoi(t) != 0;
    if (const char* t = getenv("HARE_K2P_TAIL_MAX")) o.k2p_tail_max = atoi(t);
    if (const char* t = getenv("HARE_K2P_TAIL_PATIENCE")) o.k2p_tail_patience = atoi(t);
    if (const char* t = getenv("HARE_TICKET")) o.ticket_rays = atoi(t);
    if (const char* t = getenv("HARE_K1P_STATIC_RAYS")) o.k1p_static_rays = atoi(t);
    if (const char* t = getenv("HARE_K2P_STATIC_RAYS")) o.k2p_static_rays = atoi(t);
    if (const char* t = getenv("HARE_BATCH_CHUNKS")) o.batch_chunks = atoi(t);
    if (const char* t = getenv("HARE_TUNE")) {
        int v[5] = {0, 0, 0, 0, 0};
        if (sscanf(t, "%d,%d,%d,%d,%d", &v[0], &v[1], &v[2], &v[3], &v[4]) >= 3)
            for (int k = 0; k < 5; ++k) o.tune[k] = v[k];
    }
}

// The persistent kernels (K1p, K2p) give every wave of the grid a static first chunk of rays and hand out the rest by tickets:
// 128 rays per wave when the batch has plenty, less for a batch that does not (a fixed 128 left half of the grid's waves
// without any work at 262k rays), in steps of 32 and at least 64.  How much less differs (measured): a voxel ray
// is cheap against the ~30 ns of a ticket draw, so K1p takes the whole per-wave share statically (393k rays: 0.286 ms, with a
// quarter kept for tickets 0.335); an octree ray costs ten times as much and the end of the batch matters more than the
// tickets, so K2p keeps a quarter of the share for them (524k rays: 2.34 ms against 2.61 all static).
int32_t static_chunk_rays(int64_t n, unsigned pgrid, bool keep_a_quarter_for_tickets, bool keep_half = false, bool half_at_every_size = false)
{
    int64_t per_wave = n / ((int64_t)std::max(1u, pgrid) * 4);
    if (keep_half && half_at_every_size) {
        // K2d (swept, tools/k2d_static_sweep.sh): the optimum is half the share at every size below 786k rays -- and (round 6) that holds for the
        // caller-sized batches too, where a floor of 64 rays used to make the whole batch static: a wave that starts HALF full runs short rounds, and a
        // short launch is the chain of its heaviest rays' rounds (131 072 rays: 32 static rays per wave 261 Mrays/s, 64: 208; 196 608: 330 / 271; 262 144:
        // 48 static rays 397, 64: 366; profiles/r06_experiments/k2d_static_chunk_small_batches.log)
        per_wave = per_wave / 2;
        return (int32_t)std::max<int64_t>(16, std::min<int64_t>(128, per_wave / 8 * 8));      // never MORE than half: a wave that starts fuller than that ends late
    }
    // (K3d keeps the floor: its 16 waves per CU hold cheaper rays -- the hall at 262 144 rays 64 static rays per wave 406 Mrays/s, 32: 355; 393 216: 501 / 436)
    if (keep_half) per_wave = per_wave / 2;
    else if (keep_a_quarter_for_tickets) per_wave = per_wave * 3 / 4;
    return (int32_t)std::max<int64_t>(64, std::min<int64_t>(128, per_wave / 32 * 32));
}
size_t voxel_scene_bytes(const Scene& s, size_t top)
{
    const size_t ncell = (size_t)s.vox.ct * s.vox.ct * s.vox.ct;
    const size_t items = top < s.vox.items.size() ? s.vox.items[top].size() : 0;
    return (size_t)s.topos[top].P * (sizeof(PolyRec) + (size_t)(s.topos[top].has_quads ? 48 : kCullStride)) + ncell * sizeof(CellRec) + items * sizeof(int32_t);
}
int ticket_rays_for(const Scene& s, int64_t n, bool pool)
{
    if (s.opt.ticket_rays > 0) return std::max(8, std::min(4096, s.opt.ticket_rays));            // developer sweeps
    // measured optimum on MI355X (tools/sweep_ticket.py): 32 rays up to ~1.5M rays, where the end of the batch dominates,
    // growing to 128 where the ~11 ns/ticket same-address atomic rate would start to bind
    // K1q: 64 rays -- one full round of set-ups -- at every size (re-swept on the final round-3 kernel, hall and cathedral, 524k ... 8M
    // rays: 1M rays 32 / 48 / 64 / 96 / 128 rays per ticket 0.4425 / 0.4086 / 0.4018 / 0.4297 / 0.4177 ms; the kernel had become fast
    // enough for 32-ray tickets to run into the same-address atomic rate, ~11 ns per draw; sizes that are not a multiple of 64 leave
    // part of a set-up round empty; profiles/r03_experiments/k1q_ticket_resweep.log)
    if (pool) return n < 12582912 ? 64 : 128;
    return n < 1572864 ? 32 : (n < 6291456 ? 64 : (n < 12582912 ? 96 : 128));
}

// Which kernel serves a shoot: ONE function, used by the launcher and by hare_shoot_kernel_name, so that the name a profile
// is read by is the kernel that ran -- including the fall-backs (kernel missing from the code object, LDS that does not fit).
// `M` may be null (no device yet): the rule alone, for a 256-CU part.
//
// The voxel path has two production kernels (measured on MI355X over 9 scene / grid combinations): K1q
// (hare_voxel_pool_*) once a launch is long enough for its steady state to outweigh its longer ramp and drain, K1p
// (hare_voxel_persist_*) below.  Where that is depends on whether the scene's records stay in the L2: K1q keeps 1.5x the rays
// in flight per CU and requests eight candidates' records per task, which is what covers miss latency --
//  * a scene far beyond the L2 (the 986k-triangle cathedral, ~200 MB): K1q from one pool fill of the whole chip
//    (CUs x 12 waves x 128 rays = 393 216 rays on the 256-CU MI355X; 524k: -18 ... -28 %; 262k: +12 ... +33 %);
//  * a cache-resident scene (the 100k-triangle hall at D = 16 ... 128, 18 - 52 MB): from two fills (786 432 rays) on a grid with one
//    occupancy bit per voxel, three on a coarser bitmap (round 2, before the cooperative tails: three fills everywhere).
// Both thresholds scale with the CU count of the device the scene lives on.
constexpr bool kOctreePoolDefault = false;
constexpr int64_t kK2dShortBatchPerCu = 1280;      // K2d: below this many rays per CU (327 680 on 256 CUs) a wave is refilled at 32 idle lanes instead of 16

// flags_only: an occlusion query without events (hare_occluded_* with events == NULL): the hare_*_occl kernels, which write the
// flag and cut the traversal short; the simple kernels (counting, forced, kd-tree) write the flag after the full trace.
KernChoice choose_kernel(const Scene& s, const DeviceModule* M, int32_t kind, size_t top, int64_t n, uint32_t flags, bool flags_only)
{
    KernChoice c;
    const bool count = (flags & HARE_SHOOT_COUNT_WORK) != 0, simple = (flags & HARE_SHOOT_SIMPLE_KERNEL) != 0;
    // HARE_SHOOT_COUNT_OWN: the production kernel's counting build (K1q, K2d, the kd-tree kernel); a batch another kernel would serve
    // has none -> no kernel (the caller reports HARE_E_UNSUPPORTED)
    const bool own = (flags & HARE_SHOOT_COUNT_OWN) != 0 && !count && !simple && !flags_only;
    const bool quads = s.topos[top].has_quads;
    const bool huge = n >= 0x7FFFFF00ll;                  // the persistent kernels index rays with 32 bits
    const int cus = (M && M->cu_count > 0) ? M->cu_count : 256;
    auto pick = [&](Kern k, const char* name, hipFunction_t DeviceModule::*f) {
        c.k = k;
        c.name = name;
        c.f = M ? M->*f : nullptr;
    };
    auto have = [&](hipFunction_t DeviceModule::*f) { return !M || (M->*f) != nullptr; };
    if (kind == HARE_KIND_VOXEL) {
        if (flags & 0x8000u) { pick(Kern::VoxelAudit, "hare_cull_audit", &DeviceModule::cull_audit); return c; }
        const bool persist_ok = have(&DeviceModule::voxel_persist_tri) && have(&DeviceModule::voxel_persist_quad) &&
                                have(&DeviceModule::voxel_persist_tri_g) && have(&DeviceModule::voxel_persist_quad_g);
        if (count) { pick(Kern::VoxelCount, "hare_voxel_shoot_count", &DeviceModule::voxel_count); return c; }
        if (simple || huge || !persist_ok) {
            if (quads) pick(Kern::VoxelSimple, "hare_voxel_shoot_quad", &DeviceModule::voxel_quad);
            else pick(Kern::VoxelSimple, "hare_voxel_shoot_tri", &DeviceModule::voxel_tri);
            return c;
        }
        const bool coarse = s.occ_shift > 0;
        if (flags_only) {
            hipFunction_t DeviceModule::*of = !coarse ? (quads ? &DeviceModule::voxel_occl_quad : &DeviceModule::voxel_occl_tri)
                                                       : (quads ? &DeviceModule::voxel_occl_quad_g : &DeviceModule::voxel_occl_tri_g);
            if (have(of)) {
                pick(Kern::VoxelOccl, !coarse ? (quads ? "hare_voxel_occl_quad" : "hare_voxel_occl_tri")
                                              : (quads ? "hare_voxel_occl_quad_g" : "hare_voxel_occl_tri_g"), of);
                return c;
            }
            if (quads) pick(Kern::VoxelSimple, "hare_voxel_shoot_quad", &DeviceModule::voxel_quad);
            else pick(Kern::VoxelSimple, "hare_voxel_shoot_tri", &DeviceModule::voxel_tri);
            return c;
        }
        const unsigned lds = (unsigned)((s.occ_words + 3) / 4) * 16u;
        const bool pool_fits = lds + (unsigned)kPoolWaves * (unsigned)kPoolWaveBytes <= kLdsMax && s.vox.ct <= 512 && !(flags & 0x4000u);
        const int64_t fill = (int64_t)cus * kPoolWaves * kPoolSlots;          // rays in flight when every pool of the chip is full
        // K1q for every batch size.  Round 2 needed three pool fills of the chip on a resident scene before K1q won, the cooperative
        // tails brought that to two, the wide drain modes (voxel_pool.hip) to one -- and a batch BELOW one fill is spread over all waves
        // of the chip (ShootIO::static_rays), whose few rays each get several lanes from their second round on: K1p / K1q, ms, hall
        // D = 64: 1k 0.135 / 0.101, 16k 0.153 / 0.121, 65k 0.187 / 0.135, 131k 0.203 / 0.154, 262k 0.224 / 0.209, 393k 0.278 / 0.264,
        // 524k 0.326 / 0.285, 1M 0.483 / 0.452; cathedral D = 128: 1k 0.231 / 0.170, 16k 0.251 / 0.196, 65k 0.293 / 0.245, 131k
        // 0.302 / 0.322, 262k 0.354 / 0.337, 524k 0.609 / 0.439 (profiles/r03_experiments/k1p_k1q_small_batches.log,
        // k1p_k1q_crossover_with_wide_drain.log).  K1p serves what the pool kernel cannot: grids beyond 512 voxels a side or a bitmap
        // that leaves no room for the pools, and the developer builds.
        (void)fill;
        const bool pool_wanted = s.opt.voxel_kernel == 2 || s.opt.voxel_kernel == 0;
        hipFunction_t DeviceModule::*pf = !coarse ? (quads ? &DeviceModule::voxel_pool_quad : &DeviceModule::voxel_pool_tri)
                                                   : (quads ? &DeviceModule::voxel_pool_quad_g : &DeviceModule::voxel_pool_tri_g);
        if (pool_wanted && pool_fits && have(pf)) {
            pick(Kern::VoxelPool, !coarse ? (quads ? "hare_voxel_pool_quad" : "hare_voxel_pool_tri")
                                          : (quads ? "hare_voxel_pool_quad_g" : "hare_voxel_pool_tri_g"), pf);
            if (own) {         // HARE_SHOOT_COUNT_OWN: the counting build of the SAME kernel, same launch geometry
                hipFunction_t DeviceModule::*of = !coarse ? (quads ? &DeviceModule::voxel_pool_quad_own : &DeviceModule::voxel_pool_tri_own)
                                                           : (quads ? &DeviceModule::voxel_pool_quad_g_own : &DeviceModule::voxel_pool_tri_g_own);
                pick(Kern::VoxelPool, !coarse ? (quads ? "hare_voxel_pool_quad_own" : "hare_voxel_pool_tri_own")
                                              : (quads ? "hare_voxel_pool_quad_g_own" : "hare_voxel_pool_tri_g_own"), of);
            }
            return c;
        }
        if (own) { c = KernChoice(); return c; }          // no counting build of K1p
        if ((flags & 0x4000u) && have(&DeviceModule::voxel_persist_prof) && (!M || M->voxel_persist_prof) && !coarse && !quads) {
            pick(Kern::VoxelProf, "hare_voxel_persist_prof", &DeviceModule::voxel_persist_prof);
            return c;
        }
        pick(Kern::VoxelPersist, !coarse ? (quads ? "hare_voxel_persist_quad" : "hare_voxel_persist_tri")
                                         : (quads ? "hare_voxel_persist_quad_g" : "hare_voxel_persist_tri_g"),
             !coarse ? (quads ? &DeviceModule::voxel_persist_quad : &DeviceModule::voxel_persist_tri)
                     : (quads ? &DeviceModule::voxel_persist_quad_g : &DeviceModule::voxel_persist_tri_g));
        return c;
    }
    if (kind == HARE_KIND_OCTREE) {
        if (count) { pick(Kern::OctCount, "hare_octree_shoot_count", &DeviceModule::octree_count); return c; }
        const int levels = std::max(1, s.oct_levels);
        const bool small_tree = (int64_t)s.oct.nodes.size() < (1 << 23);
        if (!simple && !huge && small_tree && flags_only) {
            if ((flags & kFlagAnyHit) && (unsigned)levels * 256u * 20u <= kLdsMax && have(&DeviceModule::octree_occl_any)) {
                pick(Kern::OctOccl, "hare_octree_occl_any", &DeviceModule::octree_occl_any);      // no t_max: K2p's OCC build (tests a leaf at once)
                return c;
            }
            if ((unsigned)levels * 256u * 20u + kOctDenseExtra <= kLdsMax && have(&DeviceModule::octree_occl)) {
                pick(Kern::OctOccl, "hare_octree_occl", &DeviceModule::octree_occl);
                return c;
            }
        } else if (!simple && !huge && small_tree) {
            const bool pool_wanted = s.opt.octree_kernel == 2 || (s.opt.octree_kernel == 0 && kOctreePoolDefault && n >= 65536);
            if (pool_wanted && have(&DeviceModule::octree_pool)) { pick(Kern::OctPool, "hare_octree_pool", &DeviceModule::octree_pool); return c; }
            // K2g (octree_group.hip): eight lanes per ray -- the production kernel for closest-hit batches of every size
            // Octree.Shoot has two production kernels (measured on MI355X, hall, 8 levels; profiles/r04_experiments/k2d_*.log, k2_crossover.log):
            //   K2g (octree_group.hip, eight lanes per ray): a ray lives < 100 us, so a launch has next to no drain -- 2.1x K2p at 65k rays --
            //       but it spends 1.5x K2p's instructions per ray: steady state 335 Mrays/s;
            //   K2d (hare_octree_dense: one lane per ray, leaf entries spread densely over the wave, exact tests deferred) from 425 984 rays (first half of round 4; see below):
            //       K2d / K2g Mrays/s at 262k 235 / 256, 393k 271 / 281, 524k 350 / 301, 786k 441 / 312, 1M 493 / 319, 4M 656 / 337.
            //   K2p (hare_octree_persist) is K2d's predecessor: the A/B baseline (octree_kernel = 1) and the fall-back where K2d's LDS does not fit.
            // The threshold scales with the CU count.
            const bool fits_p = (unsigned)levels * 256u * 20u <= kLdsMax && have(&DeviceModule::octree_persist);
            const bool group_ok = have(&DeviceModule::octree_group);
            // (second half of round 4: K2d no longer spends a pop step on an exhausted frame, forms its slabs in cursor order and keeps HALF
            //  of a wave's share for tickets -- K2d / K2g at 131k 159 / 180, 196k 235 / 225, 262k 293 / 255, 393k 418 / 283, 524k 474 / 304:
            //  the crossover is a ray for every lane of K2d's grid, 768 per CU)
            // (round 6: K2d's static first chunk is half a wave's share at EVERY size -- a floor of 64 rays used to make a small batch all static --
            //  and the crossover fell: K2d / K2g at 65 536 rays 123 / 135, 81 920 169 / 152, 98 304 188 / 168, 131 072 263 / 217 Mrays/s)
            const int64_t group_below = (int64_t)cus * 320;           // 81 920 rays on the 256-CU part
            const bool group_wanted = s.opt.octree_kernel == 3 || (s.opt.octree_kernel == 0 && (n < group_below || !fits_p));
            if (group_wanted && group_ok) {
                if (own) { c = KernChoice(); return c; }          // no counting build of K2g: the caller is told so
                pick(Kern::OctGroup, "hare_octree_group", &DeviceModule::octree_group);
                return c;
            }
            // K2d (K2p's DENSE build) wherever it exists and its LDS fits; K2p (octree_kernel = 1) is the A/B baseline and the fall-back
            if ((s.opt.octree_kernel == 4 || s.opt.octree_kernel == 0) && (unsigned)levels * 256u * 20u + kOctDenseExtra <= kLdsMax &&
                have(&DeviceModule::octree_dense)) {
                pick(Kern::OctDense, "hare_octree_dense", &DeviceModule::octree_dense);
                if (own) pick(Kern::OctDense, "hare_octree_dense_own", &DeviceModule::octree_dense_own);
                return c;
            }
            if (own) { c = KernChoice(); return c; }
            if ((unsigned)levels * 256u * 20u <= kLdsMax && have(&DeviceModule::octree_persist)) {
                pick(Kern::OctPersist, "hare_octree_persist", &DeviceModule::octree_persist);
                return c;
            }
        }
        pick(Kern::OctSimple, "hare_octree_shoot", &DeviceModule::octree);
        return c;
    }
    if (kind == HARE_KIND_KDTREE) {
        if (count) { pick(Kern::KdCount, "hare_kdtree_shoot_count", &DeviceModule::kdtree_count); return c; }
        // K3d (hare_kdtree_dense, kdtree_dense.hip): persistent waves, one-line node records with both children's tight boxes, leaves pre-culled
        // densely, exact tests deferred -- the production kernel of KDTree.Shoot since round 5 wherever its node records exist for the
        // topology and its stack fits LDS (any depth hare_kdtree_build allows does); the one-ray-per-lane kernel (kdtree_kernel = 1) is the
        // A/B baseline, the fall-back, and what the flags-only occlusion predicate runs
        const bool dense_fits = !simple && !huge && s.opt.kdtree_kernel != 1 && top < s.d_kd_dev.size() && s.d_kd_dev[top] != nullptr &&
                                kd_dense_lds(s.kd.depth_reached) <= kLdsMax;
        if (dense_fits && flags_only && have(&DeviceModule::kdtree_occl)) {      // the flag without events: K3d's occlusion build (stops at the first hit below t_max)
            pick(Kern::KdDense, "hare_kdtree_occl", &DeviceModule::kdtree_occl);
            return c;
        }
        const bool dense_ok = dense_fits && !flags_only && have(&DeviceModule::kdtree_dense);
        if (dense_ok) {
            pick(Kern::KdDense, "hare_kdtree_dense", &DeviceModule::kdtree_dense);
            if (own) pick(Kern::KdDense, "hare_kdtree_dense_own", &DeviceModule::kdtree_dense_own);
            return c;
        }
        if (own) return c;                           // (no counting build of the one-ray-per-lane kd kernel)
        pick(Kern::KdSimple, "hare_kdtree_shoot", &DeviceModule::kdtree);
    }
    return c;
}

// [a, a + na) and [b, b + nb) share a byte
bool ranges_overlap(const void* a, size_t na, const void* b, size_t nb)
{
    const uintptr_t x = (uintptr_t)a, y = (uintptr_t)b;
    return a && b && x < y + nb && y < x + na;
}

// The grid as the voxel kernels take it
static void fill_voxel_args(const Scene& s, int32_t top, VoxelArgs& g)
{
    memset(&g, 0, sizeof g);
    g.polys = (const PolyRec*)s.d_polys[top];
    g.cull = (const unsigned char*)s.d_cull[top];
    g.cf = s.cull_frames[(size_t)top];
    g.quads = (const QuadRec*)s.d_quads[top];
    g.cells = (const CellRec*)s.d_cells[top];
    g.items = (const int32_t*)s.d_items[top];
    g.occ = (const uint32_t*)s.d_occ[top];
    g.ct = s.vox.ct;
    g.occ_words = s.occ_words;
    g.occ_shift = s.occ_shift;
    g.occ_cd = s.occ_cd;
    g.bocc = nullptr;
    g.bocc_nb = 0;
    g.bocc_words = 0;
    if (s.opt.voxel_skip && (size_t)top < s.d_bocc.size() && s.d_bocc[(size_t)top] && s.bocc_nb > 0) {
        g.bocc = (const uint32_t*)s.d_bocc[(size_t)top];
        g.bocc_nb = s.bocc_nb;
        g.bocc_words = s.bocc_words;
    }
    if (s.opt.voxel_tight && (size_t)top < s.d_cellbox.size() && s.cellbox_rad > 0) {
        g.cellbox = (const float*)s.d_cellbox[(size_t)top];
        for (int a = 0; a < 3; ++a) g.cellbox_mid[a] = s.cellbox_mid[a];
        g.cellbox_rad = s.cellbox_rad;
    }
    for (int a = 0; a < 3; ++a) {
        g.omin[a] = s.vox.omin[a];
        g.omax[a] = s.vox.omax[a];
        g.vd[a] = s.vox.vd[a];
    }
}

// ---- the specular bounce loop on device buffers (hare_bounce_device, and the loop inside hare_bounce_batch) ------------------------
// `casts` casts per ray; between casts the ray is reflected about the polygon it hit and that polygon is excluded (hare_reflect).
// Voxel_Grid where the pool kernel serves (every grid up to 512 voxels a side whose bitmap leaves room for the pools) and casts <= 16:
// ONE launch of hare_voxel_bounce_* -- every ray runs through its casts on its own, no barrier between casts (voxel_pool.hip).
// Anything else: casts x (shoot + reflect) launches on the stream, retired rays skipped; no host synchronisation either way.
//   d_rays   n rays, READ AND OVERWRITTEN (a work array: a ray's last reflection remains)
//   d_work   2 n int32 of scratch (the exclusions of the casts behind the first)
//   d_all    nullable: casts x n events, cast-major;  d_last: nullable when d_all is given: the last cast's n events
//   d_ctr    nullable: totals, accumulated (rays = casts with a live ray);  d_ctr_casts: nullable, `casts` blocks, accumulated
int bounce_device_impl(Scene& s, const HipApi* H, int32_t kind, int32_t top, int64_t n, void* d_rays, const void* d_e1, const void* d_e2,
                       int32_t casts, uint32_t flags, void* d_work, void* d_all, void* d_last, void* d_ctr, void* d_ctr_casts, hipStream_t st)
{
    if (n < 0 || casts < 1 || casts > 4096 || top < 0 || top >= (int32_t)s.topos.size()) {
        set_error("hare_bounce: bad n, bounces or top_index");
        return HARE_E_INVALID;
    }
    if (n == 0) return HARE_OK;
    if (!d_rays || !d_work || (!d_all && !d_last)) {
        set_error("hare_bounce: null rays / work array / events");
        return HARE_E_INVALID;
    }
    if (n > 0x7FFFFF00ll) {
        set_error("hare_bounce: batch too large");
        return HARE_E_INVALID;
    }
    flags = sanitize_flags(s, flags) & (HARE_SHOOT_COUNT_WORK | HARE_SHOOT_SIMPLE_KERNEL | HARE_SHOOT_COUNT_OWN);
    const DeviceModule& M = *s.module;
    hare_xevent* const all = (hare_xevent*)d_all;
    hare_xevent* const last = d_last ? (hare_xevent*)d_last : all + (size_t)(casts - 1) * (size_t)n;
    int32_t* const work = (int32_t*)d_work;
    if (!M.reflect || !M.events_fill_miss) {
        set_error("hare_bounce: bounce kernels missing from code object");
        return HARE_E_STATE;
    }
    // ---- one launch?
    if (kind == HARE_KIND_VOXEL && casts <= kBounceMaxCasts && flags == 0 && s.vox.built && !s.d_cells.empty()) {
        const KernChoice kc = choose_kernel(s, &M, kind, (size_t)top, n, 0u);
        const bool quads = s.topos[top].has_quads, coarse = s.occ_shift > 0;
        hipFunction_t f = !coarse ? (quads ? M.voxel_bounce_quad : M.voxel_bounce_tri) : (quads ? M.voxel_bounce_quad_g : M.voxel_bounce_tri_g);
        const unsigned lds = (unsigned)((s.occ_words + 3) / 4) * 16u;
        const unsigned plds = lds + (unsigned)kPoolWaves * (unsigned)(kPoolWaveBytes + kPoolBounceExtra);
        if (kc.k == Kern::VoxelPool && f != nullptr && plds <= kLdsMax && s.opt.bounce_fused) {
            // the work arrays: exclusions of cast 0 (none: -1), rewritten per ray as it goes from cast to cast
            if (d_e1) HIP_TRY(H->MemcpyAsync(work, d_e1, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToDevice, st));
            else HIP_TRY(H->MemsetAsync(work, 0xFF, (size_t)n * sizeof(int32_t), st));
            if (d_e2) HIP_TRY(H->MemcpyAsync(work + n, d_e2, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToDevice, st));
            if (all) {       // a ray that dies leaves the events of its later casts untouched: they start as miss records
                void* full = all;
                long long nn = (long long)n * casts;
                void* a1[] = {&full, &nn};
                if (int rc = launch(H, M.events_fill_miss, (unsigned)((nn + 255) / 256), 256, 0, st, a1)) return rc;
            }
            VoxelArgs g;
            fill_voxel_args(s, top, g);
            ShootIO io;
            memset(&io, 0, sizeof io);
            io.rays = (RayRec*)d_rays;
            io.excl1 = work;
            io.excl2 = d_e2 ? work + n : nullptr;
            io.out = (XEventRec*)last;
            io.ctr = (unsigned long long*)d_ctr;
            io.n = n;
            io.bounce_casts = casts;
            io.out_all = (XEventRec*)all;
            io.out_stride = n;
            io.ctr_casts = (unsigned long long*)d_ctr_casts;
            const unsigned cus = (unsigned)std::max(1, M.cu_count);
            unsigned pgrid = std::min<unsigned>(cus, (unsigned)((n + kPoolWaves - 1) / kPoolWaves));
            if (pgrid == 0) pgrid = 1;
            const int64_t per_wave = (n + (int64_t)pgrid * kPoolWaves - 1) / ((int64_t)pgrid * kPoolWaves);
            io.static_rays = (int32_t)std::max<int64_t>(8, std::min<int64_t>(128, (per_wave + 7) / 8 * 8));
            io.ticket_rays = ticket_rays_for(s, n, true);
            void* args[] = {&g, &io};
            return launch_on_slot(s, H, f, pgrid, 64u * (unsigned)kPoolWaves, plds, st, io, args, true);
        }
    }
    // ---- a launch per cast
    auto sum_counters = [&](int sign) -> int {
        if (!(d_ctr_casts && d_ctr)) return HARE_OK;
        if (!M.counters_sum) {
            set_error("hare_bounce: hare_counters_sum missing from code object");
            return HARE_E_STATE;
        }
        const void* pc = d_ctr_casts;
        int cc = casts;
        void* a[] = {&pc, &cc, &d_ctr, &sign};
        return launch(H, M.counters_sum, 1, 64, 0, st, a);
    };
    // Open scenes (round 6): behind every reflection the live BLOCKS of 64 rays are listed (hare_reflect: a byte per block; hare_live_blocks: the list), and
    // the pool kernel's next cast walks that list instead of the ray array: a block in which every ray has been retired costs the cast nothing
    // -- not a ticket, not a byte.  Device-side throughout (the count stays in device memory; no host round trip), no launch more than
    // the plain loop makes.  The list lives in the second half of the caller's work array (2 n int32: the first half carries the marks; the
    // second only serves the one-launch loop): n / 64 bytes, n / 64 words, the count.  A closed room, where nothing dies, pays the one-workgroup launch per cast (~1 %; scene option "bounce_pack" 0 switches it off).  Events and counters are those of the plain loop
    // (tests: every cast against the oracle's loop, open soups and closed rooms).  Voxel_Grid batches the pool kernel serves only; other
    // kernels cast all n rays and skip the retired ones as before.
    const int64_t nblk = (n + 63) / 64;
    unsigned char* const blk_live = (unsigned char*)(work + n);
    uint32_t* const blk_list = (uint32_t*)(blk_live + ((nblk + 15) & ~(int64_t)15));
    uint32_t* const blk_words = blk_list + nblk;          // {list length, epoch of the last dead block, epoch whose list is ready}
    // (Not with a buffer per cast, `all`: there every slot of every cast is written, a retired ray's with its miss record.)
    const bool use_blocks = !all && kind == HARE_KIND_VOXEL && M.live_blocks != nullptr && s.opt.bounce_pack != 0 && n >= 4096 && nblk <= 0x7FFFFFF0ll &&
                            (size_t)((nblk + 15) & ~(int64_t)15) + (size_t)(nblk + 4) * 4u <= (size_t)n * 4u &&
                            choose_kernel(s, &M, kind, (size_t)top, n, flags | HARE_SHOOT_RETIRED_RAYS).k == Kern::VoxelPool;
    if (int rc = sum_counters(-1)) return rc;            // the per-cast blocks are accumulated into: the totals get what THIS loop adds
    // From here on the caller's totals have the per-cast blocks SUBTRACTED: whatever ends the loop early (a cast whose kernel has no counting
    // build under HARE_SHOOT_COUNT_OWN, a failed launch) must put them back, or the totals stay short by what earlier loops counted (ADVICE)
    auto fail = [&](int rc) {
        const std::string msg = last_error();
        (void)sum_counters(+1);
        set_error(msg);
        return rc;
    };
    for (int32_t c = 0; c < casts; ++c) {
        hare_xevent* out_c = all ? all + (size_t)c * (size_t)n : last;
        void* ctr_c = d_ctr_casts ? (void*)((hare_counters*)d_ctr_casts + c) : d_ctr;
        const uint32_t f = flags | (c > 0 ? (uint32_t)HARE_SHOOT_RETIRED_RAYS : 0u);
        // every cast into ONE event buffer (`last`): a ray retired before this cast left its miss record there in the cast it died in, and the
        // pool kernel need not write it again (SHOOT_RETIRED_SILENT); with a buffer per cast (`all`) every slot is written as before
        ShootExtra extra;
        extra.internal_flags = (c > 0 && !all) ? (uint32_t)SHOOT_RETIRED_SILENT : 0u;
        if (c > 0 && use_blocks) { extra.blocks = blk_list; extra.blk_words = blk_words; }
        if (int rc = shoot_device_impl(s, H, kind, top, n, d_rays, c == 0 ? d_e1 : work, c == 0 ? d_e2 : nullptr, f, out_c, ctr_c, st, nullptr, nullptr, &extra)) return fail(rc);
        if (c + 1 < casts) {
            const void* polys = s.d_polys[(size_t)top];
            const void* ev = out_c;
            void* ex = work;
            long long mm = n;
            int32_t marks_valid = c > 0 ? 1 : 0;       // `work` holds the previous reflection's marks from the second reflection on
            unsigned char* bl = use_blocks ? blk_live : nullptr;
            void* a[] = {&polys, &d_rays, &ev, &ex, &mm, &marks_valid, &bl};
            if (int rc = launch(H, M.reflect, (unsigned)((n + 255) / 256), 256, 0, st, a)) return fail(rc);
            if (use_blocks) {
                uint32_t nb = (uint32_t)nblk;
                const unsigned char* blc = blk_live;
                uint32_t* lst = blk_list;
                uint32_t* cnt = blk_words;
                void* a2[] = {&blc, &nb, &lst, &cnt};
                if (int rc = launch(H, M.live_blocks, 1, 1024, 0, st, a2)) return fail(rc);
            }
        }
    }
    if (all && d_last) HIP_TRY(H->MemcpyAsync(d_last, all + (size_t)(casts - 1) * (size_t)n, (size_t)n * sizeof(hare_xevent), hipMemcpyDeviceToDevice, st));
    if (int rc = sum_counters(+1)) return rc;
    return HARE_OK;
}

int shoot_device_impl(Scene& s, const HipApi* H, int32_t kind, int32_t top, int64_t n, void* d_rays,
                      const void* d_e1, const void* d_e2, uint32_t flags, void* d_out, void* d_ctr, hipStream_t st, const void* d_tmax,
                      void* d_occ, const ShootExtra* extra)
{
    if (d_out && d_occ) {
        // events AND flags: the closest-hit cast as it is, then one compare per ray on the events it wrote (hare_occlusion)
        int rc = shoot_device_impl(s, H, kind, top, n, d_rays, d_e1, d_e2, flags, d_out, d_ctr, st, nullptr, nullptr);
        if (rc || n <= 0) return rc;
        if (!s.module->occlusion) {
            set_error("hare_occluded: kernel missing from code object");
            return HARE_E_STATE;
        }
        const void* ev = d_out;
        void* args[] = {&ev, &d_tmax, &d_occ, &n};
        return launch(H, s.module->occlusion, (unsigned)((n + 255) / 256), 256, 0, st, args);
    }
    const bool flags_only = d_occ != nullptr;
    flags = sanitize_flags(s, flags);
    if (n < 0 || top < 0 || top >= (int32_t)s.topos.size()) {
        set_error("hare_shoot: bad n or top_index");
        return HARE_E_INVALID;
    }
    if (n == 0) return HARE_OK;
    if (n > 0x7FFFFFFFll * 64) {
        set_error("hare_shoot: batch too large");
        return HARE_E_INVALID;
    }
    if (!d_rays || (!d_out && !d_occ)) {
        set_error("hare_shoot: null rays/out");
        return HARE_E_INVALID;
    }
    if (flags_only) flags &= ~(uint32_t)SHOOT_WRITEBACK_ORIGIN & ~0xE000u;     // a predicate: rays are input only, no developer modes
    if (flags_only && d_tmax == nullptr) flags |= kFlagAnyHit;
    // A live ray's own X_Event slot is its scratch in the pool kernels, and rays[] is re-read while events are written: the
    // buffers of one call must not alias (each other, the exclusion arrays, or the counters)
    {
        const size_t rb = (size_t)n * sizeof(hare_ray), ob = (size_t)n * sizeof(hare_xevent), eb = (size_t)n * sizeof(int32_t);
        if (ranges_overlap(d_rays, rb, d_occ, eb) || ranges_overlap(d_tmax, (size_t)n * 8, d_occ, eb) || ranges_overlap(d_e1, eb, d_occ, eb) ||
            ranges_overlap(d_e2, eb, d_occ, eb) || ranges_overlap(d_ctr, sizeof(hare_counters), d_occ, eb)) {
            set_error("hare_occluded: rays, exclusions, t_max, flags and counters must not overlap");
            return HARE_E_INVALID;
        }
        if (ranges_overlap(d_rays, rb, d_out, ob) || ranges_overlap(d_e1, eb, d_out, ob) || ranges_overlap(d_e2, eb, d_out, ob) ||
            ranges_overlap(d_ctr, sizeof(hare_counters), d_out, ob) || ranges_overlap(d_ctr, sizeof(hare_counters), d_rays, rb)) {
            set_error("hare_shoot: rays, exclusions, events and counters must not overlap");
            return HARE_E_INVALID;
        }
    }
    ShootIO io;
    memset(&io, 0, sizeof io);
    io.rays = (RayRec*)d_rays;
    io.excl1 = (const int32_t*)d_e1;
    io.excl2 = (const int32_t*)d_e2;
    io.out = (XEventRec*)d_out;
    io.ctr = (unsigned long long*)d_ctr;
    io.work = (unsigned int*)s.d_work;
    io.tmax = (const double*)d_tmax;
    io.occluded = (int32_t*)d_occ;
    io.n = n;
    io.flags = flags | (extra ? extra->internal_flags : 0u);        // (internal bits -- SHOOT_RETIRED_SILENT -- come from this library's own loops, never through sanitize_flags)
    io.steps_per_round = 10;
    io.refill_min_idle = 16;
    io.ray_chunk = 128;
    io.exact_min_parked = 8;
    io.audit_polys = s.topos[top].P;
    unsigned tune_blocks_per_cu = 0;
    if (s.opt.tune[0] > 0 && s.opt.tune[1] > 0 && s.opt.tune[1] <= 64 && s.opt.tune[2] > 0) {
        // developer sweeps (tools/sweep.py, tools/phase_prof.py): steps,refill,chunk,blocks_per_cu,exact;
        // only blocks_per_cu reaches the production kernels, the rest the profiling build
        io.steps_per_round = s.opt.tune[0];
        io.refill_min_idle = s.opt.tune[1];
        io.ray_chunk = s.opt.tune[2];
        tune_blocks_per_cu = s.opt.tune[3] > 0 ? (unsigned)s.opt.tune[3] : 0u;
        if (s.opt.tune[4] > 0 && s.opt.tune[4] <= 64) io.exact_min_parked = s.opt.tune[4];
    }
    const DeviceModule& M = *s.module;
    auto no_own_build = [&]() {
        set_error("hare_shoot: HARE_SHOOT_COUNT_OWN -- the kernel this batch gets has no counting build (the pool kernel of Voxel_Grid, "
                  "hare_octree_dense from 768 rays per CU, the kd-tree kernel have one)");
        return HARE_E_UNSUPPORTED;
    };
    const unsigned block = 256;
    const unsigned grid = (unsigned)((n + block - 1) / block);
    const unsigned cus = (unsigned)std::max(1, M.cu_count);

    if (kind == HARE_KIND_VOXEL) {
        if (!s.vox.built || s.d_cells.empty()) {
            set_error("hare_shoot: voxel grid not built");
            return HARE_E_STATE;
        }
        VoxelArgs g;
        fill_voxel_args(s, top, g);
        const KernChoice kc = choose_kernel(s, &M, kind, (size_t)top, n, flags, flags_only);
        if (!kc.f && (flags & HARE_SHOOT_COUNT_OWN)) return no_own_build();
        if (!kc.f) {
            set_error(kc.k == Kern::VoxelAudit ? "hare_shoot: the cull audit kernel is missing from the code object"
                                               : "hare_shoot: kernel missing from code object");
            return HARE_E_STATE;
        }
        void* args[] = {&g, &io};
        if (kc.k == Kern::VoxelAudit || kc.k == Kern::VoxelCount || kc.k == Kern::VoxelSimple)
            return launch(H, kc.f, grid, block, 0, st, args);
        const unsigned lds = (unsigned)((s.occ_words + 3) / 4) * 16u;     // the occupancy bitmap, <= 64 KB (occ_layout)
        if ((flags & 0x3000u) && d_ctr) io.prof = (unsigned long long*)d_ctr + CTR_WORDS;   // developer timeline (0x2000) / round trace (0x1000)
        if (kc.k == Kern::VoxelPool) {
            // K1q (voxel_pool.hip): more rays than lanes, ray state in LDS, one workgroup per CU
            const unsigned plds = lds + (unsigned)kPoolWaves * (unsigned)kPoolWaveBytes + (g.bocc ? (unsigned)((g.bocc_words + 3) / 4) * 16u : 0u);     // + the block bits of "voxel_skip"
            // a workgroup per CU whenever the batch has a ray for every wave; the static first chunk is what the batch has for each wave,
            // in steps of 8, at most 128 (a small batch: few rays per wave, each with several lanes from its second round on)
            unsigned pgrid = std::min<unsigned>(cus, (unsigned)((n + kPoolWaves - 1) / kPoolWaves));
            if (pgrid == 0) pgrid = 1;
            const int64_t per_wave = (n + (int64_t)pgrid * kPoolWaves - 1) / ((int64_t)pgrid * kPoolWaves);
            io.static_rays = (int32_t)std::max<int64_t>(8, std::min<int64_t>(128, (per_wave + 7) / 8 * 8));
            if (s.opt.k1p_static_rays > 0) io.static_rays = std::max(8, std::min(1024, s.opt.k1p_static_rays / 8 * 8));   // developer sweeps (tools/k1q_ticket_sweep.py)
            io.ticket_rays = ticket_rays_for(s, n, true);
            io.walk_steps = n >= 2 * (int64_t)cus * kPoolWaves * kPoolSlots ? 32 : 16;      // long walk tasks once the batch is two pool fills of the chip (786 432 rays on 256 CUs)
            if (extra && extra->blocks) { io.blocks = extra->blocks; io.blk_words = extra->blk_words; }      // a cast of the bounce loop: live blocks only
            if (s.opt.dev && s.opt.dev_order_ptr && !io.blocks) io.order = (const uint32_t*)(uintptr_t)s.opt.dev_order_ptr;
            // The order in which K1q takes the rays (order_kernels.hip): inside every window of 4 096 consecutive rays, by an estimate of
            // the walk length -- a pool of rays of similar cost wastes fewer lane-steps (C4 shard -6.8 %, C2 -5.3 % of the kernel's time with
            // the order given; window_sort_*.log), the batch's own locality stays.  Rule ("voxel_order" 1, the default): batches of PRIMARY
            // rays -- no exclusion arrays, not a cast of the bounce loop: reflected rays gain nothing and pay for the indirection
            // (+5 ... +8 %, window_sort_cathedral_bounce5.log) -- from kOrderMinRays: the pass reads every ray once more (12 us per million
            // rays: half of HBM's rate), which at 1M rays is what the order gains.  The scratch is a block of the scene's order ring
            // (stream-ordered allocation was tried first: hipMallocAsync / hipFreeAsync cost the stream more than the pass itself).
            const bool order_rule = io.blocks ? false : s.opt.voxel_order == 2 || (s.opt.voxel_order == 1 && !d_e1 && !d_e2 && !(flags & SHOOT_RETIRED_RAYS) && n >= kOrderMinRays);
            if (!io.order && order_rule && M.cost_order && (size_t)n <= s.order_cap && (flags & 0xF000u) == 0 && !stream_is_capturing(H, st)) {
                // a block of the scene's order ring (scene.h), reserved when the grid went to the device: nothing is allocated, freed or
                // synchronised here.  The block's mutex is held across wait + launches + record (four enqueues)
                const int ob = (int)(s.order_seq.fetch_add(1) % (unsigned)Scene::kOrderRing);
                std::lock_guard<std::mutex> olk(s.order_blk_mu[ob]);
                if (s.order_used[ob]) HIP_TRY(H->StreamWaitEvent(st, s.order_ev[ob], 0));
                const void* rp = d_rays;
                void* d_order = static_cast<uint32_t*>(s.d_order) + (size_t)ob * s.order_cap;
                long long nn = n;
                float o0[3], o1[3], iv[3];
                for (int a = 0; a < 3; ++a) { o0[a] = (float)s.vox.omin[a]; o1[a] = (float)s.vox.omax[a]; iv[a] = (float)(1.0 / s.vox.vd[a]); }
                float bpv = (float)kOrderBins / (3.0f * (float)std::max(1, s.vox.ct));
                void* oargs[] = {&rp, &nn, &o0[0], &o0[1], &o0[2], &o1[0], &o1[1], &o1[2], &iv[0], &iv[1], &iv[2], &bpv, &d_order};
                int rc = launch(H, M.cost_order, (unsigned)((n + kOrderWindow - 1) / kOrderWindow), (unsigned)kOrderThreads, 0, st, oargs);
                if (rc == HARE_OK) {
                    io.order = (const uint32_t*)d_order;
                    rc = launch_on_slot(s, H, kc.f, pgrid, 64u * (unsigned)kPoolWaves, plds, st, io, args, true);
                }
                if (H->EventRecord(s.order_ev[ob], st) == hipSuccess) s.order_used[ob] = true;
                else (void)H->GetLastError();
                return rc;
            }
            return launch_on_slot(s, H, kc.f, pgrid, 64u * (unsigned)kPoolWaves, plds, st, io, args, true);
        }
        // persistent kernel K1p: a grid that just fills the chip; waves draw ray chunks from a ticket
#ifndef HARE_OCCL_WAVES_PER_EU
#define HARE_OCCL_WAVES_PER_EU 4
#endif
        unsigned per_cu = kc.k == Kern::VoxelOccl ? HARE_OCCL_WAVES_PER_EU : 4;       // what the occlusion build is compiled for (kernels.hip)
        if (tune_blocks_per_cu) per_cu = tune_blocks_per_cu;
        if (lds) per_cu = std::min<unsigned>(per_cu, (unsigned)(kLdsMax / lds));
        unsigned pgrid = cus * std::max(1u, per_cu);
        pgrid = std::min<unsigned>(pgrid, (unsigned)((n + 63) / 64 + 3) / 4);
        if (pgrid == 0) pgrid = 1;
        io.ticket_rays = ticket_rays_for(s, n, false);
        // static first chunk per wave (static_chunk_rays): at 262k rays, where 128 left half the grid's waves without work, 0.348 -> 0.239 ms
        io.static_rays = static_chunk_rays(n, pgrid, false);
        if (s.opt.k1p_static_rays > 0) io.static_rays = std::max(32, std::min(256, s.opt.k1p_static_rays / 32 * 32));   // developer sweeps
        unsigned lds_total = lds;
        if (kc.k == Kern::VoxelProf) {
            if (!d_ctr) {
                set_error("hare_shoot: the phase profile needs a counters block");
                return HARE_E_INVALID;
            }
            io.prof = (unsigned long long*)d_ctr + CTR_WORDS;      // phase statistics land in the 17 u64 words FOLLOWING the counters block
            lds_total += 4u * 18u * 8u;                            // + the profiling build's per-wave statistics
        }
        return launch_on_slot(s, H, kc.f, pgrid, block, lds_total, st, io, args, true);
    }
    if (kind == HARE_KIND_OCTREE) {
        if (!s.oct.built || !s.d_oct_nodes) {
            set_error("hare_shoot: octree not built");
            return HARE_E_STATE;
        }
        if (s.oct.id_count > s.topos[(size_t)top].P) {   // the reference would index Model[top_index] out of range ("Octree - alt.cs":216)
            set_error("hare_shoot: the octree holds polygon ids of the last topology that topology " + std::to_string(top) + " does not have");
            return HARE_E_INVALID;
        }
        OctreeArgs g;
        memset(&g, 0, sizeof g);
        g.polys = (const PolyRec*)s.d_polys[top];
        g.cull = (const unsigned char*)s.d_cull[top];
        g.cf = s.cull_frames[(size_t)top];
        g.quads = (const QuadRec*)s.d_quads[top];
        g.nodes = (const OctNode*)s.d_oct_nodes;
        g.items = (const int32_t*)s.d_oct_items;
        if (s.opt.octree_tight && (size_t)top < s.d_oct_tight.size() && s.oct_tight_rad > 0) {
            g.tight = (const float*)s.d_oct_tight[(size_t)top];
            for (int a = 0; a < 3; ++a) g.tight_mid[a] = s.oct_tight_mid[a];
            g.tight_rad = s.oct_tight_rad;
        }
        g.n_nodes = (int32_t)s.oct.nodes.size();
        g.max_depth = std::max(1, s.oct_levels);   // frames per lane = interior levels the tree really has (<= maxDepth)
        if ((size_t)g.max_depth * 64u * 24u > kLdsMax) {   // (24 bytes per level and lane: the simple kernel's frames)   // cannot happen while hare_octree_build caps maxDepth at 24
            set_error("hare_shoot: octree is deeper than the per-lane frames the kernels keep in LDS (" +
                      std::to_string(g.max_depth) + " levels)");
            return HARE_E_UNSUPPORTED;
        }
        const KernChoice kc = choose_kernel(s, &M, kind, (size_t)top, n, flags, flags_only);
        if (!kc.f && (flags & HARE_SHOOT_COUNT_OWN)) return no_own_build();
        if (!kc.f) {
            set_error("hare_shoot: octree kernel missing from code object");
            return HARE_E_STATE;
        }
        if ((flags & 0x3000u) && d_ctr) io.prof = (unsigned long long*)d_ctr + CTR_WORDS;   // developer timeline (0x2000) / round trace (0x1000)
        if (kc.k == Kern::OctPool) {
            // K2q (octree_pool.hip): more rays than lanes; frames below the top one in a device scratch block per launch in flight
            const unsigned stride = 24u + 24u * (unsigned)g.max_depth;
            unsigned pgrid = std::min<unsigned>(cus, (unsigned)((n + 64 * kOctPoolWaves - 1) / (64 * kOctPoolWaves)));
            if (pgrid == 0) pgrid = 1;
            const size_t need = (size_t)cus * kOctPoolWaves * kOctPoolSlots * stride;
            // The scratch ring: the block's previous user must have finished before this launch starts, and the event that says
            // so must have been RECORDED before a later launch waits on it -- so the lock is held across wait + launch + record
            // (several host threads may launch on one scene: hare_shoot_batch runs up to 12 chunk streams).
            std::lock_guard<std::mutex> lk(s.oct_scratch_mu);
            if (s.oct_scratch_bytes < need) {          // first use, or a deeper tree since: (re)allocate the ring
                for (int k = 0; k < kOctScratchRing; ++k) {
                    if (s.oct_scratch_ev[k]) (void)H->EventSynchronize(s.oct_scratch_ev[k]);
                    dev_free(H, s.d_oct_scratch[k]);
                }
                s.oct_scratch_bytes = 0;
                for (int k = 0; k < kOctScratchRing; ++k) {
                    HIP_TRY(H->Malloc(&s.d_oct_scratch[k], need));
                    if (!s.oct_scratch_ev[k]) HIP_TRY(H->EventCreateWithFlags(&s.oct_scratch_ev[k], hipEventDisableTiming));
                }
                s.oct_scratch_bytes = need;
                s.oct_scratch_next = 0;
            }
            const unsigned seq = s.oct_scratch_next.fetch_add(1);
            const unsigned ring = seq % (unsigned)kOctScratchRing;
            unsigned char* scratch = (unsigned char*)s.d_oct_scratch[ring];
            if (seq >= (unsigned)kOctScratchRing) HIP_TRY(H->StreamWaitEvent(st, s.oct_scratch_ev[ring], 0));
            io.ticket_rays = s.opt.ticket_rays > 0 ? std::max(8, std::min(4096, s.opt.ticket_rays)) : 32;
            unsigned stride_arg = stride;
            void* qargs[] = {&g, &io, &scratch, &stride_arg};
            const int rc = launch_on_slot(s, H, kc.f, pgrid, 64u * (unsigned)kOctPoolWaves, (unsigned)kOctPoolWaves * (unsigned)kOctPoolWaveBytes, st, io, qargs);
            if (rc == HARE_OK) HIP_TRY(H->EventRecord(s.oct_scratch_ev[ring], st));
            return rc;
        }
        // K2g (octree_group.hip) on rays [off, off + m): workgroups of four waves, eight rays per wave; LDS = the groups' stacks and
        // pending lists (hare_device.h).  A ray's stack can hold 7 x levels + 8 entries (the reference's LIFO, "Octree - alt.cs":268-272);
        // what LDS does not hold spills to a block of the scene's octree scratch ring
        auto launch_group = [&](int64_t off, int64_t m, hipStream_t stream) -> int {
            ShootIO sub = io;
            sub.rays = io.rays + off;
            sub.out = io.out + off;
            if (io.excl1) sub.excl1 = io.excl1 + off;
            if (io.excl2) sub.excl2 = io.excl2 + off;
            sub.n = m;
            const unsigned glds = 4u * (unsigned)kGroupWaveBytes;
            unsigned per_cu = std::min((unsigned)HARE_K2G_WAVES_PER_EU, std::max(1u, (unsigned)(kLdsMax / glds)));
            unsigned pgrid = cus * per_cu;
            pgrid = std::min<unsigned>(pgrid, (unsigned)((m + 7) / 8 + 3) / 4);            // a wave per eight rays at least
            if (pgrid == 0) pgrid = 1;
            sub.ticket_rays = s.opt.ticket_rays > 0 ? std::max(8, std::min(4096, s.opt.ticket_rays)) : 8;    // swept 8 / 16 / 32 / 64: 8 (1M rays), flat at 4M
            // static first chunk per wave: what the batch has for every wave, at most 32 rays (four rounds of eight), at least 8
            const int64_t per_wave = m / ((int64_t)pgrid * 4);
            sub.static_rays = (int32_t)std::max<int64_t>(8, std::min<int64_t>(32, per_wave / 2 / 8 * 8));
            if (s.opt.k2p_static_rays > 0) sub.static_rays = std::max(8, std::min(256, s.opt.k2p_static_rays / 8 * 8));   // developer sweeps
            OctScratch oc;
            oc.spill_entries = std::max(0, 7 * g.max_depth + 8 - kGroupStack);
            void* a[] = {&g, &sub};
            return launch_on_slot(s, H, M.octree_group, pgrid, 256, glds, stream, sub, a, false, oc);
        };
        // K2p (+ K2t behind it) or the occlusion build on rays [off, off + m)
        auto launch_persist = [&](hipFunction_t f, bool closest_hit, int64_t off, int64_t m, hipStream_t stream) -> int {
            ShootIO sub = io;
            sub.rays = io.rays + off;
            if (io.out) sub.out = io.out + off;
            if (io.excl1) sub.excl1 = io.excl1 + off;
            if (io.excl2) sub.excl2 = io.excl2 + off;
            if (io.tmax) sub.tmax = io.tmax + off;
            if (io.occluded) sub.occluded = io.occluded + off;
            sub.n = m;
            // the kernel is compiled for HARE_K2P_WAVES_PER_EU waves per SIMD (= workgroups of 4 waves per CU); a persistent
            // grid must not exceed what is resident, or the extra workgroups start when the others have finished
            // 20 bytes x levels x 256 lanes per workgroup (interval + child word); the dense build: + its pending survivors and tables
            const bool dense_k = f != nullptr && (f == M.octree_dense || f == M.octree_dense_own || f == M.octree_occl);   // (the flags-only kernel is K2d's OCC build)
            const unsigned plds = (unsigned)g.max_depth * 256u * 20u + (dense_k ? kOctDenseExtra : 0u);
            unsigned per_cu = std::min((unsigned)(dense_k ? HARE_K2D_WAVES_PER_EU : HARE_K2P_WAVES_PER_EU), std::max(1u, (unsigned)(kLdsMax / plds)));
            unsigned pgrid = cus * per_cu;
            pgrid = std::min<unsigned>(pgrid, (unsigned)((m + 63) / 64 + 3) / 4);      // (a wave for every 32 rays of a small batch was measured: no better)
            if (pgrid == 0) pgrid = 1;
            // an octree ray costs ~10x a voxel ray: ticket atomics never bind.  K2p: 32 rays; K2d finishes rays sooner and likes 16
            // (8 / 16 / 24 / 32 rays per ticket: 1M rays 478 / 502 / 466 / 489 Mrays/s, 1.5M 553 / 570 / 569 / 563, 524k 353 / 354 / 348 / 346)
            sub.ticket_rays = s.opt.ticket_rays > 0 ? std::max(8, std::min(4096, s.opt.ticket_rays)) : (dense_k ? 16 : 32);
            // K2d refills a wave when this many of its lanes are idle: 16 on long batches; on short ones (a wave draws only a few tickets) 32 --
            // swept (k2d_refill_by_size.log), 16 / 24 / 32 / 48 idle lanes: 196 608 rays 247 / 249 / 251 / 251 Mrays/s, 262 144 317 / 336 / 341 / 340,
            // 393 216 489 / 495 / 471 / 395, 524 288 559 / 556 / 558 / 530, 655 360 662 / 649 / 624 / 587, 1M 775 / 775 / 757 / 676
            if (dense_k && !(s.opt.tune[0] > 0 && s.opt.tune[1] > 0)) sub.refill_min_idle = m < (int64_t)cus * kK2dShortBatchPerCu ? 32 : 16;
            sub.static_rays = static_chunk_rays(m, pgrid, true, dense_k, dense_k);    // K2p: 262k rays 2.607 -> 1.861 ms, 524k 2.569 -> 2.336; K2d: half
                                                                             // the share (393k rays 338 -> 418 Mrays/s, 524k 421 -> 474, 655k 393 -> 515)
            if (s.opt.k2p_static_rays > 0) sub.static_rays = std::max(8, std::min(256, s.opt.k2p_static_rays / 8 * 8));   // developer sweeps
            void* a[] = {&g, &sub};
            // The closest-hit kernel hands rays to a tail kernel (the occlusion build keeps them).  Rule: K2g-tail takes EVERY ray a wave
            // still walks when the tickets run dry (option "octree_tail" 2, the default); K2t takes a wave's last sixteen after 64 rounds (1)
            OctScratch oc;
            // K2d hands nothing over by the rule: its dense passes put the whole wave on whatever entries its last rays hold, which is what
            // a tail kernel was for (1M rays: no tail 485 Mrays/s, K2g-tail after 8 / 16 / 32 / 64 / 96 rounds 428 / 451 / 465 / 454 / 463, K2t 464)
            const bool dense = dense_k;
            if (closest_hit && s.opt.octree_tail != 0 && !(dense && s.opt.k2p_tail_max == 0 && s.opt.k2p_tail_patience < 0)) {
                oc.tail_levels = g.max_depth;
                oc.group_tail = s.opt.octree_tail == 2 && M.octree_group_tail != nullptr;
                oc.tail_max = oc.group_tail ? 64 : kOctTailMax;
                // K2g-tail: every ray the wave still holds 32 rounds after its tickets ran dry (swept: (64, 0) 345 Mrays/s, (64, 8) 367,
                // (64, 24..48) 391-396, (64, 64) 377, (64, 128) 347; (24..40, x) the same within 1 %; K2t (16, 64) 384)
                oc.tail_patience = oc.group_tail ? 32 : HARE_K2P_TAIL_PATIENCE;
                if (s.opt.k2p_tail_max > 0) oc.tail_max = std::min(64, s.opt.k2p_tail_max);          // developer sweeps
                if (s.opt.k2p_tail_patience >= 0) oc.tail_patience = s.opt.k2p_tail_patience;
                if (oc.group_tail) oc.spill_entries = std::max(0, 7 * g.max_depth + 8 - kGroupStack);
            }
            return launch_on_slot(s, H, f, pgrid, 256, plds, stream, sub, a, false, oc);
        };
        if (kc.k == Kern::OctGroup) return launch_group(0, n, st);
        if (kc.k == Kern::OctPersist || kc.k == Kern::OctDense || kc.k == Kern::OctOccl) return launch_persist(kc.f, kc.k != Kern::OctOccl, 0, n, st);
        void* args[] = {&g, &io};
        // one frame per interior level and lane in LDS: 24 bytes x levels x block
        const unsigned levels = (unsigned)g.max_depth;
        unsigned ob = 256;
        while (ob > 64 && (size_t)levels * ob * 24 > 64 * 1024) ob >>= 1;   // ob = 64: up to 106 levels fit 160 KB
        const unsigned lds = levels * ob * 24;
        return launch(H, kc.f, (unsigned)((n + ob - 1) / ob), ob, lds, st, args);
    }
    if (kind == HARE_KIND_KDTREE) {
        if (!s.kd.built || !s.d_kd_nodes) {
            set_error("hare_shoot: kd-tree not built");
            return HARE_E_STATE;
        }
        if (s.kd.id_count > s.topos[(size_t)top].P) {
            set_error("hare_shoot: the kd-tree holds polygon ids of the last topology that topology " + std::to_string(top) + " does not have");
            return HARE_E_INVALID;
        }
        KdArgs g;
        memset(&g, 0, sizeof g);
        g.polys = (const PolyRec*)s.d_polys[top];
        g.quads = (const QuadRec*)s.d_quads[top];
        g.nodes = (const KdNodeRec*)s.d_kd_nodes;
        g.items = (const int32_t*)s.d_kd_items;
        g.n_nodes = (int32_t)s.kd.nodes.size();
        g.max_depth = s.kd.depth_reached;
        g.cull = (const unsigned char*)s.d_cull[top];
        g.cf = s.cull_frames[(size_t)top];
        if (s.opt.octree_tight && (size_t)top < s.d_kd_tight.size() && s.kd_tight_rad > 0) {     // the option name is the octree's: one switch for both trees
            g.tight = (const float*)s.d_kd_tight[(size_t)top];
            for (int a = 0; a < 3; ++a) g.tight_mid[a] = s.kd_tight_mid[a];
            g.tight_rad = s.kd_tight_rad;
        }
        const KernChoice kkc = choose_kernel(s, &M, kind, (size_t)top, n, flags, flags_only);
        hipFunction_t f = kkc.f;
        if (!f && (flags & HARE_SHOOT_COUNT_OWN)) return no_own_build();
        if (!f) {
            set_error("hare_shoot: kd-tree kernel missing from code object");
            return HARE_E_STATE;
        }
        if (kkc.k == Kern::KdDense) {
            // K3d: a grid that just fills the chip (what its LDS -- (depth + 2) stack entries of 8 bytes per lane -- allows per CU, at most
            // HARE_K3D_WAVES_PER_EU workgroups); static first chunk and tickets as K2d
            g.dnodes = (const KdDevNode*)s.d_kd_dev[(size_t)top];
            const unsigned klds = kd_dense_lds(g.max_depth);
            unsigned per_cu = std::min((unsigned)HARE_K3D_WAVES_PER_EU, std::max(1u, (unsigned)(kLdsMax / klds)));
            unsigned pgrid = cus * per_cu;
            pgrid = std::min<unsigned>(pgrid, (unsigned)((n + 63) / 64 + 3) / 4);
            if (pgrid == 0) pgrid = 1;
            io.ticket_rays = s.opt.ticket_rays > 0 ? std::max(8, std::min(4096, s.opt.ticket_rays)) : 16;
            io.static_rays = static_chunk_rays(n, pgrid, true, true);
            if (s.opt.k2p_static_rays > 0) io.static_rays = std::max(32, std::min(256, s.opt.k2p_static_rays / 32 * 32));   // developer sweeps
            if ((flags & 0x2000u) && d_ctr) io.prof = (unsigned long long*)d_ctr + CTR_WORDS;   // developer timeline
            void* kargs[] = {&g, &io};
            return launch_on_slot(s, H, f, pgrid, 256, klds, st, io, kargs);
        }
        // node stack in LDS: at most depth + 2 entries per lane
        const unsigned slots = (unsigned)g.max_depth + 2;
        const unsigned kb = 256;
        const unsigned lds = slots * kb * 4;
        void* args[] = {&g, &io};
        return launch(H, f, (unsigned)((n + kb - 1) / kb), kb, lds, st, args);
    }
    set_error("hare_shoot: unknown partition kind");
    return HARE_E_INVALID;
}

}  // namespace hare