// bounce.cpp -- hare_bounce_batch: the device-resident specular bounce loop behind ONE C-ABI call from host buffers.
//
// Harness-defined: the reference leaves reflection to its caller, who re-shoots with
// poly_origin1 = the polygon just hit (Spatial_Partition.cs:33; Voxel_Grid.cs:351,477) after reflecting about
// Polygon.Normal (Hare_Geometry_Polygons.cs:161-171).  A managed caller holds no device pointers: done through
// hare_shoot_batch it would cross the host link (104 B per ray) every bounce and reflect in managed code.  Here the rays go
// up once, every cast and every reflection runs on the device, and what comes down is the X_Events the caller asks for.
//
// A call that wants the LAST cast's events only is enqueued once and synchronises once (below).  A call that wants every cast's
// events reads back, per cast, the 64-byte counter block (one stream synchronisation): the number of rays that hit is the
// number that live on.  When a quarter or more of the rays in flight have died since the last packing, the survivors are
// PACKED (hare_live_count / hare_scan_tiles / hare_reflect_compact: stable, so results and order are deterministic) and the
// next cast is launched on the survivors only -- SURVEY.md 7.1 step 9; in a closed room nearly nothing dies and the rays stay
// where they are (hare_reflect marks the few dead -2, the kernels skip them: HARE_SHOOT_RETIRED_RAYS).  Events of a packed
// cast are expanded to the caller's order on the device before they are downloaded.  The download of cast b runs beside
// cast b + 1 (it is issued after that cast's launches).
//
// Product code; nothing from oracle/.
#include <string.h>
#include <algorithm>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "../../include/hare_hip.h"
#include "scene.h"

namespace hare {

#define HIP_TRY(expr)                                                                          \
    do {                                                                                       \
        hipError_t _e = (expr);                                                                \
        if (_e != hipSuccess) return hip_fail(H, _e, #expr);                                   \
    } while (0)

void free_bounce_buffers(const HipApi* H, Scene& s)
{
    for (Scene::BatchCtx& c : s.ctx) {
        Scene::BounceBuf& b = c.bounce;
        if (b.copy_st) { (void)H->StreamSynchronize(b.copy_st); (void)H->StreamDestroy(b.copy_st); b.copy_st = nullptr; }
        for (void** p : {&b.rays[0], &b.rays[1], &b.excl[0], &b.excl[1], &b.excl2, &b.idx[0], &b.idx[1], &b.ev[0], &b.ev[1], &b.full,
                         &b.tiles, &b.ctr})
            dev_free(H, *p);
        b.cap = 0;
        b.ctr_cap = 0;
    }
}

namespace {

constexpr int64_t kCompactTile = 2048;          // kernels.hip: events per workgroup of the packing kernels
constexpr int64_t kPackMinRays = 16384;         // below this a cast is all launch latency: packing buys nothing

int ensure_bounce_buffers(const HipApi* H, Scene::BounceBuf& b, int64_t n, int32_t bounces)
{
    if (n > b.cap) {
        for (void** p : {&b.rays[0], &b.rays[1], &b.excl[0], &b.excl[1], &b.excl2, &b.idx[0], &b.idx[1], &b.ev[0], &b.ev[1], &b.full, &b.tiles})
            dev_free(H, *p);
        b.cap = 0;
        for (int k = 0; k < 2; ++k) {
            HIP_TRY(H->Malloc(&b.rays[k], (size_t)n * sizeof(hare_ray)));
            HIP_TRY(H->Malloc(&b.excl[k], (size_t)n * sizeof(int32_t)));
            HIP_TRY(H->Malloc(&b.idx[k], (size_t)n * sizeof(int32_t)));
            HIP_TRY(H->Malloc(&b.ev[k], (size_t)n * sizeof(hare_xevent)));
        }
        HIP_TRY(H->Malloc(&b.excl2, (size_t)n * sizeof(int32_t)));
        HIP_TRY(H->Malloc(&b.full, (size_t)n * sizeof(hare_xevent)));
        HIP_TRY(H->Malloc(&b.tiles, (size_t)((n + kCompactTile - 1) / kCompactTile + 1) * sizeof(uint32_t)));
        b.cap = n;
    }
    if (bounces > b.ctr_cap) {
        dev_free(H, b.ctr);
        b.ctr_cap = 0;
        HIP_TRY(H->Malloc(&b.ctr, (size_t)bounces * sizeof(hare_counters)));
        b.ctr_cap = bounces;
    }
    if (!b.copy_st) HIP_TRY(H->StreamCreate(&b.copy_st));
    return HARE_OK;
}

void fill_miss_host(hare_xevent* e, int64_t n)
{
    memset(e, 0, (size_t)n * sizeof(hare_xevent));          // X_Event(): Hare_Geometry_Primitives.cs:454-462
    for (int64_t i = 0; i < n; ++i) e[i].poly_id = -1;
}

// The loop for one scene (one device).  `stride` = distance between the casts of events_all (the whole batch's ray count when this
// is one shard of a sharded call).
int bounce_on_scene(Scene& s, const HipApi* H, Scene::BatchCtx& c, int32_t kind, int32_t top, int64_t n, const hare_ray* rays,
                    const int32_t* excl1, const int32_t* excl2, int32_t bounces, uint32_t flags, hare_xevent* events_all, int64_t stride,
                    hare_xevent* events_last, hare_counters* per_cast /* bounces entries, zeroed */)
{
    const DeviceModule& M = *s.module;
    if (!M.reflect || !M.live_count || !M.scan_tiles || !M.reflect_compact || !M.events_fill_miss || !M.events_expand) {
        set_error("hare_bounce_batch: bounce kernels missing from code object");
        return HARE_E_STATE;
    }
    Scene::BounceBuf& b = c.bounce;
    if (int rc = ensure_bounce_buffers(H, b, n, bounces)) return rc;
    if (!c.st[0]) HIP_TRY(H->StreamCreate(&c.st[0]));
    hipStream_t st = c.st[0];
    const void* polys = s.d_polys[(size_t)top];

    if (!events_all) {
        // ---- only the last cast's events are wanted: the whole loop is enqueued ONCE, with no host round trip between casts
        // (bounce_device_impl, launch.cpp: a launch per cast with the retired rays skipped -- or, under the scene option `bounce_fused`,
        // one launch for a Voxel_Grid where the pool kernel serves), and the call synchronises once, at its end.  No packing: a
        // retired ray costs its launch a record read and a miss record.  b.ev[1] serves as the loop's work array (2 n int32).
        HIP_TRY(H->MemcpyAsync(b.rays[0], rays, (size_t)n * sizeof(hare_ray), hipMemcpyHostToDevice, st));
        if (excl1) HIP_TRY(H->MemcpyAsync(b.excl[0], excl1, (size_t)n * sizeof(int32_t), hipMemcpyHostToDevice, st));
        if (excl2) HIP_TRY(H->MemcpyAsync(b.excl2, excl2, (size_t)n * sizeof(int32_t), hipMemcpyHostToDevice, st));
        HIP_TRY(H->MemsetAsync(b.ctr, 0, (size_t)bounces * sizeof(hare_counters), st));
        if (int rc = bounce_device_impl(s, H, kind, top, n, b.rays[0], excl1 ? b.excl[0] : nullptr, excl2 ? b.excl2 : nullptr, bounces, flags,
                                        b.ev[1], nullptr, b.ev[0], nullptr, b.ctr, st))
            return rc;
        if (events_last) HIP_TRY(H->MemcpyAsync(events_last, b.ev[0], (size_t)n * sizeof(hare_xevent), hipMemcpyDeviceToHost, st));
        HIP_TRY(H->MemcpyAsync(per_cast, b.ctr, (size_t)bounces * sizeof(hare_counters), hipMemcpyDeviceToHost, st));
        HIP_TRY(H->StreamSynchronize(st));
        return HARE_OK;
    }
    HIP_TRY(H->MemcpyAsync(b.rays[0], rays, (size_t)n * sizeof(hare_ray), hipMemcpyHostToDevice, st));
    if (excl1) HIP_TRY(H->MemcpyAsync(b.excl[0], excl1, (size_t)n * sizeof(int32_t), hipMemcpyHostToDevice, st));
    if (excl2) HIP_TRY(H->MemcpyAsync(b.excl2, excl2, (size_t)n * sizeof(int32_t), hipMemcpyHostToDevice, st));
    HIP_TRY(H->MemsetAsync(b.ctr, 0, (size_t)bounces * sizeof(hare_counters), st));

    int cur = 0;                 // which copy of rays / excl / idx the next cast reads
    int64_t m = n;               // rays in flight (packed: all of them live)
    bool packed = false;         // true: the arrays hold survivors only, idx maps them to the caller's positions
    bool marks = false;          // some of the m rays are dead and marked -2 (reflected in place since the last packing)

    auto shoot = [&](int cast) -> int {
        const void* e1 = (cast == 0) ? (excl1 ? b.excl[0] : nullptr) : b.excl[cur];
        const void* e2 = (cast == 0 && excl2) ? b.excl2 : nullptr;
        // the retire mark (-2) only means something to casts behind a reflection; a caller's own negative poly_origin excludes nothing
        const uint32_t f = (flags & ~HARE_SHOOT_RETIRED_RAYS) | ((cast > 0 && marks) ? HARE_SHOOT_RETIRED_RAYS : 0u);
        return shoot_device_impl(s, H, kind, top, m, b.rays[cur], e1, e2, f, b.ev[cast & 1], (hare_counters*)b.ctr + cast, st);
    };
    int rc = shoot(0);
    if (rc) return rc;

    for (int32_t cast = 0; cast < bounces; ++cast) {
        hare_xevent* dst = events_all ? events_all + (size_t)cast * (size_t)stride : ((cast == bounces - 1) ? events_last : nullptr);
        const void* src = b.ev[cast & 1];
        if (dst && packed) {     // back to the caller's order on the device
            unsigned blk = 256;
            long long nn = n, mm = m;
            void* full = b.full;
            const void* ev = b.ev[cast & 1];
            const void* idx = b.idx[cur];
            void* a1[] = {&full, &nn};
            if ((rc = launch(H, M.events_fill_miss, (unsigned)((n + blk - 1) / blk), blk, 0, st, a1))) return rc;
            void* a2[] = {&ev, &idx, &mm, &full};
            if ((rc = launch(H, M.events_expand, (unsigned)((m + blk - 1) / blk), blk, 0, st, a2))) return rc;
            src = b.full;
        }
        HIP_TRY(H->MemcpyAsync(&per_cast[cast], (hare_counters*)b.ctr + cast, sizeof(hare_counters), hipMemcpyDeviceToHost, st));
        HIP_TRY(H->StreamSynchronize(st));                  // cast `cast` (and its expansion) has finished; per_cast[cast] is here
        const int64_t hits = (int64_t)per_cast[cast].hits;
        const bool more = cast + 1 < bounces;
        if (more && hits > 0) {
            // ---- reflect, pack when it pays, and launch the NEXT cast before this cast's events go down the link
            unsigned blk = 256;
            const bool pack = hits * 4 <= m * 3 && m >= kPackMinRays;
            if (pack) {
                const long long mm = m, nt = (m + kCompactTile - 1) / kCompactTile;
                const void* ev = b.ev[cast & 1];
                void* tiles = b.tiles;
                void* total = (uint32_t*)b.tiles + nt;
                void* a1[] = {&ev, (void*)&mm, &tiles};
                if ((rc = launch(H, M.live_count, (unsigned)nt, 256, 0, st, a1))) return rc;
                void* a2[] = {&tiles, (void*)&nt, &total};
                if ((rc = launch(H, M.scan_tiles, 1, 1024, 0, st, a2))) return rc;
                const void* rin = b.rays[cur];
                const void* iin = packed ? b.idx[cur] : nullptr;
                void* rout = b.rays[cur ^ 1];
                void* eout = b.excl[cur ^ 1];
                void* iout = b.idx[cur ^ 1];
                void* a3[] = {&polys, &rin, &ev, &iin, &tiles, (void*)&mm, &rout, &eout, &iout};
                if ((rc = launch(H, M.reflect_compact, (unsigned)nt, 256, 0, st, a3))) return rc;
                cur ^= 1;
                m = hits;
                packed = true;
                marks = false;
            } else {
                long long mm = m;
                void* r = b.rays[cur];
                const void* ev = b.ev[cast & 1];
                void* ex = b.excl[cur];
                int32_t marks_valid = marks ? 1 : 0;       // excl[cur] carries the previous in-place reflection's marks: retired rays are not read again
                unsigned char* no_bytes = nullptr;
                void* a[] = {&polys, &r, &ev, &ex, &mm, &marks_valid, &no_bytes};
                if ((rc = launch(H, M.reflect, (unsigned)((m + blk - 1) / blk), blk, 0, st, a))) return rc;
                marks = true;
            }
            if ((rc = shoot(cast + 1))) return rc;
        }
        if (dst) {
            HIP_TRY(H->MemcpyAsync(dst, src, (size_t)n * sizeof(hare_xevent), hipMemcpyDeviceToHost, b.copy_st));
            HIP_TRY(H->StreamSynchronize(b.copy_st));
        }
        if (more && hits == 0) {       // nothing lives on: every later cast is all miss records, no ray counted
            for (int32_t k = cast + 1; k < bounces; ++k) {
                if (events_all) fill_miss_host(events_all + (size_t)k * (size_t)stride, n);
                else if (k == bounces - 1 && events_last) fill_miss_host(events_last, n);
            }
            break;
        }
    }
    HIP_TRY(H->StreamSynchronize(st));
    if (events_all && events_last) memcpy(events_last, events_all + (size_t)(bounces - 1) * (size_t)stride, (size_t)n * sizeof(hare_xevent));
    return HARE_OK;
}

int check_args(const char* who, int64_t n, const hare_ray* rays, int32_t bounces, hare_xevent* events_all, hare_xevent* events_last)
{
    if (n < 0 || bounces < 1 || bounces > 4096 || (n > 0 && !rays)) {
        set_error(std::string(who) + ": bad arguments (n >= 0, 1 <= bounces <= 4096, rays)");
        return HARE_E_INVALID;
    }
    (void)events_all;
    (void)events_last;
    return HARE_OK;
}

int bounce_one(hare_scene* s, int32_t kind, int32_t top, int64_t n, const hare_ray* rays, const int32_t* excl1, const int32_t* excl2,
               int32_t bounces, uint32_t flags, hare_xevent* events_all, int64_t stride, hare_xevent* events_last, hare_counters* per_cast)
{
    if (top < 0 || top >= (int32_t)s->topos.size()) {
        set_error("hare_bounce_batch: bad top_index");
        return HARE_E_INVALID;
    }
    // host-buffer callers: the reference's meaning of poly_origin (a negative index excludes nothing); no developer bits, no
    // origin write-back (the rays are the caller's constant input here)
    flags = sanitize_flags(*s, flags) & (HARE_SHOOT_COUNT_WORK | HARE_SHOOT_SIMPLE_KERNEL);
    DeviceGuard dev_guard(hip_api(nullptr), s->device);
    const HipApi* H = nullptr;
    Scene::BatchCtx* c = nullptr;
    {
        std::unique_lock<std::mutex> lk(s->mu);
        int rc = ensure_device(*s, H);
        if (rc) return rc;
        rc = upload_polys(*s, H);
        if (rc) return rc;
        if (n == 0) return HARE_OK;
        s->cv.wait(lk, [&] { for (Scene::BatchCtx& x : s->ctx) if (!x.busy) return true; return false; });
        for (Scene::BatchCtx& x : s->ctx)
            if (!x.busy && x.bounce.cap >= n) { c = &x; break; }
        if (!c)
            for (Scene::BatchCtx& x : s->ctx)
                if (!x.busy) { c = &x; break; }
        c->busy = true;
    }
    struct Release {
        hare_scene* s; Scene::BatchCtx* c;
        ~Release() { { std::lock_guard<std::mutex> lk(s->mu); c->busy = false; } s->cv.notify_one(); }
    } release{s, c};
    const int rc = bounce_on_scene(*s, H, *c, kind, top, n, rays, excl1, excl2, bounces, flags, events_all, stride, events_last, per_cast);
    if (rc != HARE_OK) {          // copies into the caller's buffers may still be in flight: drain before the error returns
        if (c->st[0]) (void)H->StreamSynchronize(c->st[0]);
        if (c->bounce.copy_st) (void)H->StreamSynchronize(c->bounce.copy_st);
    }
    return rc;
}

void add_counters(hare_counters& dst, const hare_counters& src)
{
    dst.rays += src.rays;
    dst.hits += src.hits;
    dst.cells += src.cells;
    dst.entries += src.entries;
    dst.tests += src.tests;
}

}  // namespace
}  // namespace hare

using namespace hare;

#define GUARD_BEGIN try {
#define GUARD_END                                               \
    }                                                           \
    catch (const std::bad_alloc&)                               \
    {                                                           \
        set_error("out of host memory");                        \
        return HARE_E_NOMEM;                                    \
    }                                                           \
    catch (...)                                                 \
    {                                                           \
        set_error("unexpected C++ exception");                  \
        return HARE_E_INVALID;                                  \
    }

extern "C" {

int hare_bounce_batch(hare_scene* s, int32_t kind, int32_t top_index, int64_t n, const hare_ray* rays, const int32_t* excl1,
                      const int32_t* excl2, int32_t bounces, uint32_t flags, hare_xevent* events_all, hare_xevent* events_last,
                      hare_counters* ctr, hare_counters* ctr_per_cast)
{
    if (!s) {
        set_error("null scene");
        return HARE_E_INVALID;
    }
    if (int rc = check_args("hare_bounce_batch", n, rays, bounces, events_all, events_last)) return rc;
    GUARD_BEGIN
    std::vector<hare_counters> pc((size_t)bounces);
    memset(pc.data(), 0, pc.size() * sizeof(hare_counters));
    if (ctr) memset(ctr, 0, sizeof *ctr);
    if (ctr_per_cast) memset(ctr_per_cast, 0, (size_t)bounces * sizeof(hare_counters));
    const int rc = bounce_one(s, kind, top_index, n, rays, excl1, excl2, bounces, flags, events_all, n, events_last, pc.data());
    if (rc) return rc;
    for (int32_t b = 0; b < bounces; ++b) {
        if (ctr) add_counters(*ctr, pc[(size_t)b]);
        if (ctr_per_cast) ctr_per_cast[b] = pc[(size_t)b];
    }
    return HARE_OK;
    GUARD_END
}

int hare_bounce_batch_sharded(hare_scene* const* scenes, int32_t n_scenes, int32_t kind, int32_t top_index, int64_t n,
                              const hare_ray* rays, const int32_t* excl1, const int32_t* excl2, int32_t bounces, uint32_t flags,
                              hare_xevent* events_all, hare_xevent* events_last, hare_counters* ctr, hare_counters* ctr_per_cast)
{
    if (!scenes || n_scenes < 1 || n_scenes > 64) {
        set_error("hare_bounce_batch_sharded: need 1..64 scenes");
        return HARE_E_INVALID;
    }
    for (int32_t k = 0; k < n_scenes; ++k)
        if (!scenes[k]) {
            set_error("hare_bounce_batch_sharded: null scene");
            return HARE_E_INVALID;
        }
    if (int rc = check_args("hare_bounce_batch_sharded", n, rays, bounces, events_all, events_last)) return rc;
    GUARD_BEGIN
    const int G = n_scenes;
    std::vector<int> rcs((size_t)G, HARE_OK);
    std::vector<std::string> errs((size_t)G);
    std::vector<std::vector<hare_counters>> pcs((size_t)G, std::vector<hare_counters>((size_t)bounces));
    auto shard = [&](int k) {
        const int64_t lo = (int64_t)((__int128)n * k / G), hi = (int64_t)((__int128)n * (k + 1) / G);
        memset(pcs[(size_t)k].data(), 0, (size_t)bounces * sizeof(hare_counters));
        try {
            rcs[(size_t)k] = bounce_one(scenes[k], kind, top_index, hi - lo, rays ? rays + lo : nullptr, excl1 ? excl1 + lo : nullptr,
                                        excl2 ? excl2 + lo : nullptr, bounces, flags, events_all ? events_all + lo : nullptr, n,
                                        events_last ? events_last + lo : nullptr, pcs[(size_t)k].data());
        } catch (...) {
            rcs[(size_t)k] = HARE_E_NOMEM;
            set_error("hare_bounce_batch_sharded: exception in a shard");
        }
        if (rcs[(size_t)k] != HARE_OK) errs[(size_t)k] = hare_last_error();     // thread-local: carry it to the caller's thread
    };
    std::vector<std::thread> workers;
    workers.reserve((size_t)G);
    for (int k = 1; k < G; ++k) {
        try {
            workers.emplace_back(shard, k);
        } catch (...) {
            shard(k);                // no thread to be had: run the shard here
        }
    }
    shard(0);
    for (auto& w : workers) w.join();
    for (int k = 0; k < G; ++k)
        if (rcs[(size_t)k] != HARE_OK) {
            set_error("shard " + std::to_string(k) + ": " + errs[(size_t)k]);
            return rcs[(size_t)k];
        }
    if (ctr) memset(ctr, 0, sizeof *ctr);
    if (ctr_per_cast) memset(ctr_per_cast, 0, (size_t)bounces * sizeof(hare_counters));
    for (int k = 0; k < G; ++k)
        for (int32_t b = 0; b < bounces; ++b) {
            if (ctr) add_counters(*ctr, pcs[(size_t)k][(size_t)b]);
            if (ctr_per_cast) add_counters(ctr_per_cast[b], pcs[(size_t)k][(size_t)b]);
        }
    return HARE_OK;
    GUARD_END
}

}  // extern "C"
