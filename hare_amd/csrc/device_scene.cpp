// device_scene.cpp -- what a scene keeps on its device: the code object, the polygon / grid / tree records and their uploads, the tight
// boxes of voxels and subtrees, the kd-tree's one-line node records, the scratch rings sized at build time.  (Split from api.cpp in round 5.)
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

extern "C" const unsigned char hare_kernels_co[];
extern "C" const unsigned char hare_kernels_co_end[];

namespace hare {

static thread_local std::string t_err;
void set_error(const std::string& msg) { t_err = msg; }
const char* last_error() { return t_err.c_str(); }

const HipApi* api_or_err()
{
    std::string e;
    const HipApi* h = hip_api(&e);
    if (!h) set_error(e);
    return h;
}

int hip_fail(const HipApi* H, hipError_t e, const char* what)
{
    set_error(std::string(what) + " failed: " + (H->GetErrorString ? H->GetErrorString(e) : "?"));
    (void)H->GetLastError();
    return (e == hipErrorOutOfMemory) ? HARE_E_NOMEM : HARE_E_HIP;
}

std::mutex g_mod_mu;
std::map<int, std::unique_ptr<DeviceModule>> g_modules;

int get_module(const HipApi* H, int device, const DeviceModule** out)
{
    std::lock_guard<std::mutex> lk(g_mod_mu);
    auto it = g_modules.find(device);
    if (it != g_modules.end()) {
        *out = it->second.get();
        return HARE_OK;
    }
    HIP_TRY(H->SetDevice(device));
    std::unique_ptr<DeviceModule> m(new DeviceModule());
    HIP_TRY(H->ModuleLoadData(&m->mod, hare_kernels_co));
    struct { const char* name; hipFunction_t* fn; } table[] = {
        {"hare_voxel_shoot_tri", &m->voxel_tri},
        {"hare_voxel_shoot_quad", &m->voxel_quad},
        {"hare_voxel_shoot_count", &m->voxel_count},
        {"hare_voxel_persist_tri", &m->voxel_persist_tri},
        {"hare_voxel_persist_quad", &m->voxel_persist_quad},
        {"hare_voxel_persist_tri_g", &m->voxel_persist_tri_g},
        {"hare_voxel_persist_quad_g", &m->voxel_persist_quad_g},
        {"hare_voxel_pool_tri", &m->voxel_pool_tri},
        {"hare_voxel_pool_quad", &m->voxel_pool_quad},
        {"hare_voxel_pool_tri_g", &m->voxel_pool_tri_g},
        {"hare_voxel_pool_quad_g", &m->voxel_pool_quad_g},
        {"hare_voxel_pool_tri_own", &m->voxel_pool_tri_own},
        {"hare_voxel_pool_quad_own", &m->voxel_pool_quad_own},
        {"hare_voxel_pool_tri_g_own", &m->voxel_pool_tri_g_own},
        {"hare_voxel_pool_quad_g_own", &m->voxel_pool_quad_g_own},
        {"hare_octree_dense_own", &m->octree_dense_own},
        {"hare_voxel_bounce_tri", &m->voxel_bounce_tri},
        {"hare_voxel_bounce_quad", &m->voxel_bounce_quad},
        {"hare_voxel_bounce_tri_g", &m->voxel_bounce_tri_g},
        {"hare_voxel_bounce_quad_g", &m->voxel_bounce_quad_g},
        {"hare_counters_sum", &m->counters_sum},
        {"hare_octree_shoot", &m->octree},
        {"hare_octree_shoot_count", &m->octree_count},
        {"hare_octree_persist", &m->octree_persist},
        {"hare_octree_pool", &m->octree_pool},
        {"hare_octree_tail", &m->octree_tail},
        {"hare_octree_group", &m->octree_group},
        {"hare_octree_group_tail", &m->octree_group_tail},
        {"hare_octree_dense", &m->octree_dense},
        {"hare_kdtree_shoot", &m->kdtree},
        {"hare_cost_order", &m->cost_order},
        {"hare_kdtree_dense", &m->kdtree_dense},
        {"hare_kdtree_dense_own", &m->kdtree_dense_own},
        {"hare_kdtree_occl", &m->kdtree_occl},
        {"hare_kdtree_shoot_count", &m->kdtree_count},
        {"hare_reflect", &m->reflect},
        {"hare_occlusion", &m->occlusion},
        {"hare_voxel_occl_tri", &m->voxel_occl_tri},
        {"hare_voxel_occl_quad", &m->voxel_occl_quad},
        {"hare_voxel_occl_tri_g", &m->voxel_occl_tri_g},
        {"hare_voxel_occl_quad_g", &m->voxel_occl_quad_g},
        {"hare_octree_occl", &m->octree_occl},
        {"hare_octree_occl_any", &m->octree_occl_any},
        {"hare_events_pack_slim", &m->events_pack_slim},
        {"hare_live_blocks", &m->live_blocks},
        {"hare_live_count", &m->live_count},
        {"hare_scan_tiles", &m->scan_tiles},
        {"hare_reflect_compact", &m->reflect_compact},
        {"hare_events_fill_miss", &m->events_fill_miss},
        {"hare_events_expand", &m->events_expand},
        {"hare_cull_audit", &m->cull_audit},
        {"hare_voxel_persist_prof", &m->voxel_persist_prof},
        {"hare_vb_count", &m->vb_count},
        {"hare_vb_fill", &m->vb_fill},
        {"hare_vb_level_count", &m->vb_level_count},
        {"hare_vb_level_fill", &m->vb_level_fill},
        {"hare_scan_block", &m->scan_block},
        {"hare_scan_add", &m->scan_add},
        {"hare_vb_sort_small", &m->vb_sort_small},
        {"hare_vb_sort_block", &m->vb_sort_block},
        {"hare_vb_finalize", &m->vb_finalize},
        {"hare_cell_boxes", &m->cell_boxes},
        {"hare_block_occ", &m->block_occ},
        {"hare_vb_find_big", &m->vb_find_big},
        {"hare_vb_fill_big", &m->vb_fill_big},
        {"hare_ob_count", &m->ob_count},
        {"hare_ob_fill", &m->ob_fill},
    };
    for (auto& t : table) {
        hipError_t e = H->ModuleGetFunction(t.fn, m->mod, t.name);
        if (e != hipSuccess) *t.fn = nullptr;   // optional kernels may be absent in a given build
    }
    (void)H->GetLastError();   // a failed lookup must not stay behind as the host's "last error"
    if (!m->voxel_tri || !m->voxel_quad) {
        set_error("embedded code object lacks hare_voxel_shoot_* (not a gfx950 device?)");
        return HARE_E_HIP;
    }
    int cus = 0;
    if (H->DeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess) cus = 256;
    m->cu_count = cus;
    *out = m.get();
    g_modules[device] = std::move(m);
    return HARE_OK;
}

int dev_free(const HipApi* H, void*& p)
{
    if (p) (void)H->Free(p);
    p = nullptr;
    return 0;
}

int upload(const HipApi* H, void** dst, const void* src, size_t bytes)
{
    if (*dst) {
        (void)H->Free(*dst);
        *dst = nullptr;
    }
    HIP_TRY(H->Malloc(dst, bytes ? bytes : 16));
    if (bytes) HIP_TRY(H->Memcpy(*dst, src, bytes, hipMemcpyHostToDevice));
    return HARE_OK;
}

int ensure_device(Scene& s, const HipApi*& H)
{
    H = api_or_err();
    if (!H) return HARE_E_NODEVICE;
    int n = 0;
    if (H->GetDeviceCount(&n) != hipSuccess || n <= 0) {
        set_error("no HIP device visible");
        return HARE_E_NODEVICE;
    }
    if (s.device < 0 || s.device >= n) {
        set_error("scene device ordinal out of range");
        return HARE_E_INVALID;
    }
    HIP_TRY(H->SetDevice(s.device));
    if (!s.module) {
        int rc = get_module(H, s.device, &s.module);
        if (rc) return rc;
    }
    if (!s.stream) HIP_TRY(H->StreamCreate(&s.stream));
    if (!s.d_work) {
        // the launch-slot ring: zeroed ONCE, here; afterwards every launch leaves its slot zeroed (launch_epilogue, kernels.hip)
        HIP_TRY(H->Malloc(&s.d_work, (size_t)kLaunchSlots * sizeof(LaunchSlotMem)));
        HIP_TRY(H->Memset(s.d_work, 0, (size_t)kLaunchSlots * sizeof(LaunchSlotMem)));
        HIP_TRY(H->DeviceSynchronize());     // launches may come on any stream
    }
    return HARE_OK;
}

// next float >= |x| * (1 + 2^-20): error-bound factors must never be rounded down
float up(double x)
{
    float f = (float)(fabs(x) * 1.00000095367431640625);
    while ((double)f < fabs(x)) f = nextafterf(f, INFINITY);
    return f;
}

// The device (and host-mirror) polygon records of one topology: PolyRec per polygon, QuadRec side array only when
// the topology has quadrilaterals.
void make_poly_records(const Topo& T, std::vector<PolyRec>& rec, std::vector<QuadRec>& quads)
{
    rec.assign((size_t)std::max(T.P, 1), PolyRec());
    memset(rec.data(), 0, rec.size() * sizeof(PolyRec));
    quads.clear();
    if (T.has_quads) {
        quads.resize((size_t)T.P);
        memset(quads.data(), 0, quads.size() * sizeof(QuadRec));
    }
    for (int32_t p = 0; p < T.P; ++p) {
        const double* V = &T.verts[(size_t)p * 12];
        PolyRec& r = rec[p];
        double e1[3], e2[3], n1 = 0, emax = 0;
        for (int a = 0; a < 3; ++a) {
            r.v0[a] = V[a];
            r.v1[a] = V[3 + a];
            r.v2[a] = V[6 + a];
            r.n[a] = T.normals[(size_t)p * 3 + a];
            e1[a] = V[3 + a] - V[a];            // edge1 / edge2 of RayXtri (Polygons.cs:452-457)
            e2[a] = V[6 + a] - V[a];
            r.e1f[a] = (float)e1[a];
            r.e2f[a] = (float)e2[a];
            n1 += fabs(e1[a]);
            emax = std::max(emax, std::max(fabs(e1[a]), fabs(e2[a])));
        }
        r.emax = up(emax);
        r.ee = up(n1 * (double)r.emax);
        if (T.nverts[p] == 4) {
            r.emax = INFINITY;                  // the record HEAD of a quadrilateral says "never cull" (tools, the 48-byte A/B layout):
            r.ee = INFINITY;                    // the dense pre-cull records are built from the corners and cull both its triangles
            r.e1f[0] = NAN;                     // (make_cull_records; round 5)
            for (int a = 0; a < 3; ++a) quads[p].v3[a] = V[9 + a];
        }
        if (T.has_quads) quads[p].nverts = T.nverts[p];
    }
}

// The pre-cull's dense records of one topology (hare_device.h, HARE_CULL32) and the frame that decodes them.
void make_cull_records(const Topo& T, const std::vector<PolyRec>& rec, std::vector<unsigned char>& dense, CullFrame& cf)
{
    memset(&cf, 0, sizeof cf);
    cf.stride = (HARE_CULL32 && T.has_quads) ? 48 : kCullStride;
    dense.assign(rec.size() * (size_t)cf.stride, 0);
#if HARE_CULL32
    // the quantisation box: the polygons' own v0 range (inside Topology.Min / Max; taken from the records so that a caller's
    // stale bounds cannot put a corner outside)
    double lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0};
    for (int32_t p = 0; p < T.P; ++p)
        for (int a = 0; a < 3; ++a) {
            const double v = rec[(size_t)p].v0[a];
            if (p == 0 || v < lo[a]) lo[a] = v;
            if (p == 0 || v > hi[a]) hi[a] = v;
        }
    constexpr double kQMax = 2097151.0;        // 2^21 - 1
    float step_max = 0, ext_max = 0;
    for (int a = 0; a < 3; ++a) {
        cf.org[a] = lo[a];
        const double ext = hi[a] - lo[a];
        float st = (ext > 0 && std::isfinite(ext)) ? up(ext / kQMax) : 0.0f;     // rounded up: q never exceeds 2^21 - 1
        cf.step[a] = st;
        step_max = std::max(step_max, st);
        ext_max = std::max(ext_max, up(ext));
    }
    // per component: quantisation <= step / 2; rebuilding tv = (float)(o - org) - q * step in FP32 adds 2^-24 (|o - org| + |tv|)
    // <= 2^-23 (|o - org| + extent).  err0 holds the ray-independent part, cull_ray adds 2^-22 |o - org|_1.
    cf.err0 = up(0.5 * (double)step_max + 2.3841858e-07 * (double)ext_max);
    for (int32_t p = 0; p < T.P; ++p) {
        const PolyRec& r = rec[(size_t)p];
        uint64_t q[3];
        for (int a = 0; a < 3; ++a) {
            double v = cf.step[a] > 0 ? std::nearbyint((r.v0[a] - cf.org[a]) / (double)cf.step[a]) : 0.0;
            if (!(v >= 0)) v = 0;                 // NaN coordinates: the edges are NaN too, the candidate is never culled
            if (v > kQMax) v = kQMax;
            q[a] = (uint64_t)v;
        }
        const uint64_t packed = q[0] | (q[1] << 21) | (q[2] << 42);
        unsigned char* d = &dense[(size_t)p * (size_t)cf.stride];
        memcpy(d, &packed, 8);
        float e1f[3], e2f[3];
        const double* V = &T.verts[(size_t)p * 12];
        for (int a = 0; a < 3; ++a) {            // from the corners themselves: the PolyRec of a quadrilateral carries NaN in e1f[0] (tools)
            e1f[a] = (float)(V[3 + a] - V[a]);
            e2f[a] = (float)(V[6 + a] - V[a]);
        }
        memcpy(d + 8, e1f, 12);
        memcpy(d + 20, e2f, 12);
        if (cf.stride == 48) {
            float w2[4] = {0.0f, 0.0f, 0.0f, 0.0f};
            if (T.nverts[p] == 4) {
                for (int a = 0; a < 3; ++a) w2[a] = (float)(V[9 + a] - V[a]);      // e3f: the second triangle is (v0, v2, v3)
                w2[3] = 1.0f;
            }
            memcpy(d + 32, w2, 16);
        }
    }
#else
    static_assert(offsetof(PolyRec, ee) == 48, "the 48-byte pre-cull record is the head of the PolyRec");
    for (size_t p = 0; p < rec.size(); ++p) memcpy(&dense[p * 48], &rec[p], 48);
    (void)T;
#endif
}

int upload_polys(Scene& s, const HipApi* H)
{
    if (s.d_polys.size() == s.topos.size()) return HARE_OK;
    s.d_polys.assign(s.topos.size(), nullptr);
    s.d_quads.assign(s.topos.size(), nullptr);
    s.d_cull.assign(s.topos.size(), nullptr);
    // all or nothing: a scene whose record arrays are only partly on the device must not look uploaded to the next call
    auto fail = [&](int rc) {
        for (auto* v : {&s.d_polys, &s.d_quads, &s.d_cull}) {
            for (void*& p : *v) dev_free(H, p);
            v->clear();
        }
        return rc;
    };
    s.cull_frames.assign(s.topos.size(), CullFrame());
    for (size_t m = 0; m < s.topos.size(); ++m) {
        const Topo& T = s.topos[m];
        std::vector<PolyRec> rec;
        std::vector<QuadRec> quads;
        make_poly_records(T, rec, quads);
        int rc = upload(H, &s.d_polys[m], rec.data(), rec.size() * sizeof(PolyRec));
        if (rc) return fail(rc);
        std::vector<unsigned char> dense;
        make_cull_records(T, rec, dense, s.cull_frames[m]);
        rc = upload(H, &s.d_cull[m], dense.data(), dense.size());
        if (rc) return fail(rc);
        if (T.has_quads) {
            rc = upload(H, &s.d_quads[m], quads.data(), quads.size() * sizeof(QuadRec));
            if (rc) return fail(rc);
        }
    }
    return HARE_OK;
}

int upload_voxel(Scene& s, const HipApi* H)
{
    const VoxelHost& g = s.vox;
    const size_t M = s.topos.size();
    for (auto* v : {&s.d_cells, &s.d_items, &s.d_occ}) {
        for (void*& p : *v) dev_free(H, p);
        v->assign(M, nullptr);
    }
    const size_t ncell = (size_t)g.ct * g.ct * g.ct;
    occ_layout(g.ct, s.occ_shift, s.occ_cd, s.occ_words);
    for (size_t m = 0; m < M; ++m) {
        std::vector<CellRec> cells(ncell);
        std::vector<uint32_t> occ((size_t)((s.occ_words + 3) / 4) * 4, 0u);   // padded to 16 bytes for uint4 staging
        for (size_t c = 0; c < ncell; ++c) {
            cells[c].start = g.start[m][c];
            cells[c].count = g.start[m][c + 1] - g.start[m][c];
            cells[c].i0 = cells[c].count > 0 ? g.items[m][cells[c].start] : -1;
            cells[c].i1 = cells[c].count > 1 ? g.items[m][cells[c].start + 1] : -1;
            if (cells[c].count) {
                const size_t z = c % g.ct, y = (c / g.ct) % g.ct, x = c / ((size_t)g.ct * g.ct);
                const size_t b = (((x >> s.occ_shift) * s.occ_cd) + (y >> s.occ_shift)) * s.occ_cd + (z >> s.occ_shift);
                occ[b >> 5] |= 1u << (b & 31);
            }
        }
        int rc = upload(H, &s.d_cells[m], cells.data(), cells.size() * sizeof(CellRec));
        if (rc) return rc;
        rc = upload(H, &s.d_items[m], g.items[m].data(), g.items[m].size() * sizeof(int32_t));
        if (rc) return rc;
        rc = upload(H, &s.d_occ[m], occ.data(), occ.size() * sizeof(uint32_t));
        if (rc) return rc;
    }
    return HARE_OK;
}

int launch(const HipApi* H, hipFunction_t f, unsigned grid, unsigned block, unsigned lds, hipStream_t st, void** args)
{
    HIP_TRY(H->ModuleLaunchKernel(f, grid, 1, 1, block, 1, 1, lds, st, args, nullptr));
    return HARE_OK;
}

// The voxels' tight boxes (hare_cell_boxes, build_kernels.hip), per topology, from the grid as it stands on the device -- behind either
// builder.  Margin 2^-20 of the scene's extent; good for ray origins within 1 024 extents of the scene (the kernel's guard).
// They are an ACCELERATION, never a precondition, and cost 32 B per voxel and topology (twice the CellRec array: 4.3 GB at D = 512),
// so they exist only where they are used:
//   * the option voxel_tight is on (hare_scene_set_option("voxel_tight", 1) on a grid built without them builds them then);
//   * the pool kernel K1q, the only kernel that reads them, can serve the grid (pool_can_serve: ct <= 512, bitmap + pools fit LDS);
//   * they fit the budget `voxel_tight_max_mb` (0 = no budget) -- and an allocation that fails is "no boxes", not a failed build:
//     what was allocated is freed, cellbox_rad stays -1, the grid is traced exactly as before (every list scanned).
// Returns an error only for a kernel launch / synchronisation failure (the device is then in trouble whatever we do).
bool pool_can_serve(const Scene& s)
{
    const unsigned lds = (unsigned)((s.occ_words + 3) / 4) * 16u;
    return lds + (unsigned)kPoolWaves * (unsigned)kPoolWaveBytes <= 160u * 1024u && s.vox.ct <= 512;
}
// The order ring of the pool kernel (scene.h; launch.cpp: the order pass of hare_cost_order): reserved HERE, when the grid goes to the device
// (every builder ends in upload_cell_boxes) and when the options that size it change -- single-caller moments by contract, nothing of the
// scene in flight -- so that a shoot never allocates.  kOrderRing blocks of `voxel_order_max_rays` entries, one allocation; the ring
// exists only while "voxel_order" is on and the pool kernel serves the grid.  A failed allocation is "no ring" (casts run in the caller's
// order, same events), never a failed build; hare_scene_get_option "voxel_order_bytes" says what is held.
void reserve_order_ring(Scene& s, const HipApi* H)
{
    if (!H) return;
    const bool want_ring = s.opt.voxel_order != 0 && s.opt.voxel_order_max_rays > 0 && s.vox.built && pool_can_serve(s) && s.module && s.module->cost_order;
    const size_t cap = !want_ring ? 0 : (((size_t)s.opt.voxel_order_max_rays + 65535u) & ~(size_t)65535u);
    if (cap == s.order_cap && (cap == 0 || s.d_order)) return;
    for (int k = 0; k < Scene::kOrderRing; ++k) {
        if (s.order_ev[k] && s.order_used[k]) (void)H->EventSynchronize(s.order_ev[k]);
        s.order_used[k] = false;
    }
    dev_free(H, s.d_order);
    s.d_order = nullptr;
    s.order_cap = 0;
    if (cap == 0) return;
    for (int k = 0; k < Scene::kOrderRing; ++k)
        if (!s.order_ev[k] && H->EventCreateWithFlags(&s.order_ev[k], hipEventDisableTiming) != hipSuccess) { (void)H->GetLastError(); s.order_ev[k] = nullptr; return; }
    if (s.opt.dev_fail_cellbox_alloc || H->Malloc(&s.d_order, (size_t)Scene::kOrderRing * cap * sizeof(uint32_t)) != hipSuccess) {
        (void)H->GetLastError();
        s.d_order = nullptr;
        return;
    }
    s.order_cap = cap;
}
// The block-level occupancy behind the scene option "voxel_skip" (hare_block_occ, build_kernels.hip): per topology one bit per aligned block of 4^3
// voxels, from the grid as it stands on the device.  Exists only while the option is on, the pool kernel serves the grid, and the bits fit the
// LDS the pools leave (they are staged behind them); otherwise the option is simply without effect.  Never an error.
void upload_block_occ(Scene& s, const HipApi* H)
{
    for (void*& p : s.d_bocc) dev_free(H, p);
    s.d_bocc.assign(s.topos.size(), nullptr);
    s.bocc_nb = 0;
    s.bocc_words = 0;
    if (!s.opt.voxel_skip || !pool_can_serve(s) || !s.module || !s.module->block_occ || !s.vox.built || s.d_cells.size() != s.topos.size()) return;
    const int nb = (s.vox.ct + 3) / 4;
    const long long words = ((long long)nb * nb * nb + 31) / 32;
    const unsigned lds = (unsigned)((s.occ_words + 3) / 4) * 16u + (unsigned)kPoolWaves * (unsigned)kPoolWaveBytes;
    if (lds + (unsigned long long)((words + 3) / 4) * 16ull + 256ull > 160ull * 1024ull) return;       // (256: the kernel's static LDS)
    const long long ncell = (long long)s.vox.ct * s.vox.ct * s.vox.ct;
    for (size_t m = 0; m < s.topos.size(); ++m) {
        if (!s.d_cells[m]) continue;
        const size_t bytes = (size_t)((words + 3) / 4) * 16u;
        if (H->Malloc(&s.d_bocc[m], bytes) != hipSuccess) { (void)H->GetLastError(); s.d_bocc[m] = nullptr; continue; }
        if (H->Memset(s.d_bocc[m], 0, bytes) != hipSuccess) { (void)H->GetLastError(); dev_free(H, s.d_bocc[m]); continue; }
        const void* cells = s.d_cells[m];
        long long nc = ncell;
        int ct = s.vox.ct, nbb = nb;
        void* out = s.d_bocc[m];
        void* args[] = {&cells, &nc, &ct, &nbb, &out};
        if (launch(H, s.module->block_occ, (unsigned)((ncell + 255) / 256), 256, 0, nullptr, args)) { dev_free(H, s.d_bocc[m]); continue; }
    }
    if (H->StreamSynchronize(nullptr) != hipSuccess) {
        (void)H->GetLastError();
        for (void*& p : s.d_bocc) dev_free(H, p);
        return;
    }
    s.bocc_nb = nb;
    s.bocc_words = (int32_t)words;
}
int upload_cell_boxes(Scene& s, const HipApi* H)
{
    reserve_order_ring(s, H);                 // the other reservation every voxel build ends in
    upload_block_occ(s, H);
    for (void*& p : s.d_cellbox) dev_free(H, p);
    s.d_cellbox.assign(s.topos.size(), nullptr);
    s.cellbox_rad = -1;
    if (!s.opt.voxel_tight || !pool_can_serve(s)) return HARE_OK;
    if (!s.module || !s.module->cell_boxes || !s.vox.built || s.d_cells.size() != s.topos.size() || s.d_polys.size() != s.topos.size()) return HARE_OK;
    double ext = 0, mag = 0;
    for (int a = 0; a < 3; ++a) {
        ext = std::max(ext, s.vox.omax[a] - s.vox.omin[a]);
        mag = std::max(mag, std::max(std::fabs(s.vox.omin[a]), std::fabs(s.vox.omax[a])));
    }
    for (const Topo& T : s.topos)
        for (int a = 0; a < 3; ++a) {
            ext = std::max(ext, T.mx[a] - T.mn[a]);
            mag = std::max(mag, std::max(std::fabs(T.mn[a]), std::fabs(T.mx[a])));
        }
    if (!(ext > 0 && std::isfinite(ext) && ext < 1e100 && std::isfinite(mag))) return HARE_OK;
    // 2^-20 of the extent, or of the largest coordinate for a scene far from the origin of its coordinates (as for the trees' boxes)
    const double delta = std::ldexp(std::max(ext, mag), -20);
    const long long ncell = (long long)s.vox.ct * s.vox.ct * s.vox.ct;
    const size_t bytes = (size_t)ncell * 8 * sizeof(float);
    size_t live = 0;
    for (size_t m = 0; m < s.topos.size(); ++m)
        if (s.d_cells[m] && s.d_items[m] && s.d_polys[m]) ++live;
    auto give_up = [&]() {
        for (void*& p : s.d_cellbox) dev_free(H, p);
        s.cellbox_rad = -1;
        return HARE_OK;
    };
    if (s.opt.voxel_tight_max_mb > 0 && (double)bytes * (double)live > (double)s.opt.voxel_tight_max_mb * 1048576.0) return give_up();
    for (size_t m = 0; m < s.topos.size(); ++m) {
        if (!s.d_cells[m] || !s.d_items[m] || !s.d_polys[m]) continue;
        if (s.opt.dev_fail_cellbox_alloc || H->Malloc(&s.d_cellbox[m], bytes) != hipSuccess) {     // out of memory (or the test hook): no boxes
            s.d_cellbox[m] = nullptr;
            (void)H->GetLastError();
            return give_up();
        }
        const void* cells = s.d_cells[m];
        const void* items = s.d_items[m];
        const void* polys = s.d_polys[m];
        const void* quads = s.d_quads[m];
        long long nc = ncell;
        double dl = delta;
        void* out = s.d_cellbox[m];
        void* args[] = {&cells, &items, &polys, &quads, &nc, &dl, &out};
        if (int rc = launch(H, s.module->cell_boxes, (unsigned)((ncell + 255) / 256), 256, 0, nullptr, args)) { give_up(); return rc; }
    }
    if (hipError_t e = H->StreamSynchronize(nullptr); e != hipSuccess) { give_up(); return hip_fail(H, e, "hipStreamSynchronize"); }
    for (int a = 0; a < 3; ++a) s.cellbox_mid[a] = 0.5 * (s.vox.omin[a] + s.vox.omax[a]);
    s.cellbox_rad = 1024.0 * ext;
    return HARE_OK;
}

// Frames the octree kernels keep per lane: one per interior level the tree really has.
int32_t octree_levels(const OctreeHost& o)
{
    if (o.nodes.empty()) return 1;
    int32_t best = 0;
    std::vector<std::pair<int32_t, int32_t>> st;   // node, depth
    st.emplace_back(0, 0);
    while (!st.empty()) {
        const auto [ni, d] = st.back();
        st.pop_back();
        const OctNode& nd = o.nodes[(size_t)ni];
        if (nd.first_child < 0) continue;
        best = std::max(best, d + 1);
        for (int c = 0; c < 8; ++c) st.emplace_back(nd.first_child + c, d + 1);
    }
    return std::max(best, 1);
}

// The octree kernels' scratch ring (launch_on_slot: hand-over records K2p / K2d -> tail kernel, stack spill of K2g / K2g-tail), sized ONCE,
// when the tree goes to the device, for the largest launch this tree can get on this device: a full K2g grid, or a full K2p / K2d grid
// whose every wave hands over 64 rays to a full K2g-tail grid.  A shoot then never allocates -- round 4 grew the ring inside the launch
// path under hipDeviceSynchronize, a device-wide stall in a call documented as stream-ordered (ADVICE).  Cost, kOctTailRing = 8 blocks:
// about 0.7 GB for an 8-level tree on the 256-CU part, about 1.2 GB at 24 levels (hare_scene_get_option "octree_scratch_bytes";
// INTEGRATION.md).  A failed allocation here is not an error: the launch path still grows the ring on demand, as before.
void reserve_oct_scratch(Scene& s, const HipApi* H)
{
    if (!s.module || !H) return;
    const size_t cus = (size_t)std::max(1, s.module->cu_count);
    const size_t levels = (size_t)std::max(1, s.oct_levels);
    const size_t spill_entries = (size_t)std::max(0, 7 * (int)levels + 8 - kGroupStack);
    const size_t glds = 4u * (size_t)kGroupWaveBytes;
    const size_t g_per_cu = std::min<size_t>((size_t)HARE_K2G_WAVES_PER_EU, std::max<size_t>(1, kLdsMax / glds));
    const size_t need_group = cus * g_per_cu * 4u * 8u * spill_entries * 24u;
    const size_t plds = levels * 256u * 20u;
    const size_t p_per_cu = std::min<size_t>((size_t)HARE_K2P_WAVES_PER_EU, std::max<size_t>(1, kLdsMax / plds));
    const size_t stride = ((size_t)kOctTailHead + 20u * levels + 15u) & ~(size_t)15u;
    const size_t rec_bytes = (cus * p_per_cu * 4u * 64u * stride + 255u) & ~(size_t)255u;
    const size_t tail_spill = cus * (size_t)HARE_K2G_WAVES_PER_EU * 4u * 8u * spill_entries * 24u;
    // ... but only what the scene's OPTIONS can launch (ADVICE, round 5): by the library's rule -- K2g below 320 rays per CU, K2d above, K2d handing
    // nothing over -- no launch ever writes a hand-over record, and the blocks hold K2g's stack spill alone (nothing for trees up to
    // (kGroupStack - 8) / 7 levels: the 8-level bench tree reserves 0 bytes where round 5 held 0.7 GB per scene).  Records are reserved when
    // K2p is forced ("octree_kernel" 1), a developer hand-over rule is set, or the tree is too deep for K2d's LDS -- hare_scene_set_option
    // calls this again when one of those changes.
    const bool dense_fits = (unsigned)levels * 256u * 20u + kOctDenseExtra <= kLdsMax;
    const bool handover = s.opt.octree_tail != 0 && (s.opt.octree_kernel == 1 || s.opt.k2p_tail_max > 0 || s.opt.k2p_tail_patience >= 0 || !dense_fits);
    const size_t need = std::max(need_group, handover ? rec_bytes + tail_spill : (size_t)0);
    std::lock_guard<std::mutex> lk(s.oct_tail_mu);
    if (need <= s.oct_tail_block_bytes) return;
    if (s.d_oct_tail) {
        if (H->DeviceSynchronize() != hipSuccess) { (void)H->GetLastError(); return; }     // a build call: nothing of this scene is in flight by contract
        dev_free(H, s.d_oct_tail);
    }
    s.oct_tail_block_bytes = 0;
    if (H->Malloc(&s.d_oct_tail, (size_t)Scene::kOctTailRing * need) != hipSuccess) {
        (void)H->GetLastError();
        s.d_oct_tail = nullptr;
        return;
    }
    s.oct_tail_block_bytes = need;
    for (bool& u : s.oct_tail_used) u = false;
}

// ---- the tight boxes of the trees, the kd-tree's device nodes, and the push of a built partition to the device
// The TIGHT boxes of an octree over one topology: for every node, the bounding box of all polygons the lists of its subtree hold --
// whole polygons, not clipped to anything: Octree.Shoot accepts a hit wherever it lies on the polygon ("Octree - alt.cs":224-233, F15) --
// grown by `delta` and rounded outwards to floats.  A ray that misses that box cannot make RayXtri accept any of those polygons: an
// accepted hit lies on the polygon to within the rounding of the exact test (~1e-13 of the distances involved), and delta is 2^-20 of
// the scene's extent -- ten million times that -- as long as the origin stays within 1 024 extents of the scene (the guard the kernels
// apply; beyond it they test every node as before).  So K2p / K2d may skip a popped node whose box the ray misses: no accept is lost,
// and nothing else about the walk depends on that node.  8 floats per node: lo xyz, hi xyz, two spare.
// (One routine for both trees: `kids(k, c)` lists node k's children into c and returns how many -- 0 for a leaf --, `leaf(k, start, count)`
// gives a leaf's list.)
template <class Kids, class Leaf>
static void make_tight_boxes_of(size_t n, const std::vector<int32_t>& items, Kids kids, Leaf leaf, const Topo& T, double delta, std::vector<float>& out)
{
    out.clear();
    std::vector<double> box(n * 6);
    const double inf = std::numeric_limits<double>::infinity();
    for (size_t k = 0; k < n; ++k) {
        double* b = &box[k * 6];
        b[0] = b[1] = b[2] = inf;
        b[3] = b[4] = b[5] = -inf;
    }
    // children are stored behind their parent (every builder appends a node's children when it splits it): one backward sweep
    // folds every subtree into its root; a tree that is not laid out that way gets no boxes at all (out stays empty)
    for (size_t k = n; k-- > 0;) {
        double* b = &box[k * 6];
        int32_t ch[8];
        const int nc = kids(k, ch);
        if (nc == 0) {
            int32_t start = 0, count = 0;
            leaf(k, start, count);
            if (start < 0 || count < 0 || (size_t)start + (size_t)count > items.size()) return;
            for (int32_t q = 0; q < count; ++q) {
                const int32_t id = items[(size_t)start + (size_t)q];
                if (id < 0 || id >= T.P) return;
                const double* v = &T.verts[(size_t)id * 12];
                const int nv = T.nverts[(size_t)id] == 4 ? 4 : 3;
                for (int c = 0; c < nv; ++c)
                    for (int a = 0; a < 3; ++a) {
                        const double x = v[c * 3 + a];
                        if (!(x == x)) { b[a] = -inf; b[3 + a] = inf; continue; }      // a NaN corner: the box is everything
                        if (x < b[a]) b[a] = x;
                        if (x > b[3 + a]) b[3 + a] = x;
                    }
            }
        } else {
            for (int c = 0; c < nc; ++c) {
                if (ch[c] < 0 || (size_t)ch[c] <= k || (size_t)ch[c] >= n) return;
                const double* cb = &box[(size_t)ch[c] * 6];
                for (int a = 0; a < 3; ++a) {
                    if (cb[a] < b[a]) b[a] = cb[a];
                    if (cb[3 + a] > b[3 + a]) b[3 + a] = cb[3 + a];
                }
            }
        }
    }
    auto down = [](double x) { float f = (float)x; if ((double)f > x) f = std::nextafterf(f, -std::numeric_limits<float>::infinity()); return f; };
    auto upf = [](double x) { float f = (float)x; if ((double)f < x) f = std::nextafterf(f, std::numeric_limits<float>::infinity()); return f; };
    out.assign(n * 8, 0.0f);
    for (size_t k = 0; k < n; ++k) {
        const double* b = &box[k * 6];
        float* o = &out[k * 8];
        for (int a = 0; a < 3; ++a) {
            o[a] = down(b[a] - delta);
            o[3 + a] = upf(b[3 + a] + delta);
        }
    }
}
static void make_tight_boxes(const OctreeHost& oct, const Topo& T, double delta, std::vector<float>& out)
{
    make_tight_boxes_of(
        oct.nodes.size(), oct.items,
        [&](size_t k, int32_t* c) { const int32_t fc = oct.nodes[k].first_child; if (fc < 0) return 0; for (int j = 0; j < 8; ++j) c[j] = fc + j; return 8; },
        [&](size_t k, int32_t& st, int32_t& cn) { st = oct.nodes[k].item_start; cn = oct.nodes[k].item_count; }, T, delta, out);
}
static void make_tight_boxes(const KdHost& kd, const Topo& T, double delta, std::vector<float>& out)
{
    make_tight_boxes_of(
        kd.nodes.size(), kd.items,
        [&](size_t k, int32_t* c) { const KdNodeRec& nd = kd.nodes[k]; if (nd.left < 0 && nd.right < 0) return 0; c[0] = nd.left; c[1] = nd.right; return 2; },
        [&](size_t k, int32_t& st, int32_t& cn) { st = kd.nodes[k].item_start; cn = kd.nodes[k].item_count; }, T, delta, out);
}
// What both trees need around them: the margin (2^-20 of the scene's extent), the boxes of every topology a query may name, and the
// range of origins they are good for.  `tight` is freed and refilled.
template <class Tree>
static int upload_tight_boxes(Scene* s, const HipApi* H, const Tree& tree, int32_t id_count, std::vector<void*>& tight, double mid[3], double& rad)
{
    for (void*& p : tight) dev_free(H, p);
    tight.assign(s->topos.size(), nullptr);
    rad = -1;
    if (s->topos.empty()) return HARE_OK;
    double lo[3], hi[3];
    for (int a = 0; a < 3; ++a) { lo[a] = s->topos[0].mn[a]; hi[a] = s->topos[0].mx[a]; }
    for (const Topo& T : s->topos)
        for (int a = 0; a < 3; ++a) { lo[a] = std::min(lo[a], T.mn[a]); hi[a] = std::max(hi[a], T.mx[a]); }
    double ext = 0, mag = 0;
    for (int a = 0; a < 3; ++a) {
        ext = std::max(ext, hi[a] - lo[a]);
        mag = std::max(mag, std::max(std::fabs(lo[a]), std::fabs(hi[a])));
    }
    if (!(ext > 0 && std::isfinite(ext) && ext < 1e100 && std::isfinite(mag))) return HARE_OK;
    // the margin: 2^-20 of the scene's extent -- or of its largest coordinate when the scene lies far from the origin of its coordinates,
    // where the rounding of the exact test (and of this one) is that of the COORDINATES, not of the extent
    const double delta = std::ldexp(std::max(ext, mag), -20);
    for (size_t m = 0; m < s->topos.size(); ++m) {
        if (id_count > s->topos[m].P) continue;
        std::vector<float> tb;
        make_tight_boxes(tree, s->topos[m], delta, tb);
        if (tb.empty()) continue;
        if (int rc = upload(H, &tight[m], tb.data(), tb.size() * sizeof(float))) return rc;
    }
    for (int a = 0; a < 3; ++a) mid[a] = 0.5 * (lo[a] + hi[a]);
    rad = 1024.0 * ext;
    return HARE_OK;
}

// hare_kdtree_dense's node records (KdDevNode, hare_device.h), per topology a query may name: the host tree's node with the tight boxes
// of BOTH its children's subtrees inlined (the same boxes upload_tight_boxes sends: same margin, same outward rounding) and the mark of a
// child whose subtree lists no polygon.  A topology for which the boxes cannot be made (a tree not laid out parent-before-children)
// gets no records and is served by the one-ray-per-lane kernel.
static int upload_kd_dev_nodes(Scene* s, const HipApi* H)
{
    for (void*& p : s->d_kd_dev) dev_free(H, p);
    s->d_kd_dev.assign(s->topos.size(), nullptr);
    if (s->topos.empty() || s->kd.nodes.empty() || !(s->kd_tight_rad > 0)) return HARE_OK;
    double lo[3], hi[3];
    for (int a = 0; a < 3; ++a) { lo[a] = s->topos[0].mn[a]; hi[a] = s->topos[0].mx[a]; }
    for (const Topo& T : s->topos)
        for (int a = 0; a < 3; ++a) { lo[a] = std::min(lo[a], T.mn[a]); hi[a] = std::max(hi[a], T.mx[a]); }
    double ext = 0, mag = 0;
    for (int a = 0; a < 3; ++a) {
        ext = std::max(ext, hi[a] - lo[a]);
        mag = std::max(mag, std::max(std::fabs(lo[a]), std::fabs(hi[a])));
    }
    const double delta = std::ldexp(std::max(ext, mag), -20);              // as upload_tight_boxes
    const size_t n = s->kd.nodes.size();
    // subtrees without a polygon (a fact of the tree: the same for every topology)
    std::vector<unsigned char> has(n, 0);
    for (size_t k = n; k-- > 0;) {
        const KdNodeRec& nd = s->kd.nodes[k];
        if (nd.left < 0 && nd.right < 0) has[k] = nd.item_count > 0;
        else {
            if (nd.left < 0 || nd.right < 0 || (size_t)nd.left <= k || (size_t)nd.right <= k || (size_t)nd.left >= n || (size_t)nd.right >= n) return HARE_OK;
            has[k] = has[(size_t)nd.left] | has[(size_t)nd.right];
        }
    }
    for (size_t m = 0; m < s->topos.size(); ++m) {
        if (s->kd.id_count > s->topos[m].P || m >= s->d_kd_tight.size() || !s->d_kd_tight[m]) continue;
        std::vector<float> tb;
        make_tight_boxes(s->kd, s->topos[m], delta, tb);
        if (tb.size() != n * 8) continue;
        std::vector<KdDevNode> dev(n);
        for (size_t k = 0; k < n; ++k) {
            const KdNodeRec& nd = s->kd.nodes[k];
            KdDevNode& o = dev[k];
            memset(&o, 0, sizeof o);
            const bool leaf = nd.left < 0 && nd.right < 0;
            o.split = nd.split;
            o.axis = leaf ? -1 : nd.axis;
            o.left = nd.left;
            o.right = nd.right;
            o.item_start = nd.item_start;
            o.item_count = nd.item_count;
            if (!leaf) {
                const int a = nd.axis, b = (a == 0) ? 1 : 0, c = (a == 2) ? 1 : 2;          // KDTree.cs:249-353: the two other axes, ascending
                o.bb[0] = nd.bmin[b]; o.bb[1] = nd.bmax[b]; o.bb[2] = nd.bmin[c]; o.bb[3] = nd.bmax[c];
                for (int j = 0; j < 6; ++j) { o.tl[j] = tb[(size_t)nd.left * 8 + j]; o.tr[j] = tb[(size_t)nd.right * 8 + j]; }
                o.empty = (has[(size_t)nd.left] ? 0 : 1) | (has[(size_t)nd.right] ? 0 : 2);
            }
        }
        if (int rc = upload(H, &s->d_kd_dev[m], dev.data(), dev.size() * sizeof(KdDevNode))) return rc;
    }
    return HARE_OK;
}
// After a host build: push the partition to the device when one is available.  Builds succeed
// without a GPU (introspection works); shooting then fails with HARE_E_NODEVICE.
int sync_partition_to_device(Scene& scene, int kind)
{
    Scene* const s = &scene;
    std::string e;
    const HipApi* H = hip_api(&e);
    int n = 0;
    if (!H || H->GetDeviceCount(&n) != hipSuccess || n <= 0) return HARE_OK;
    int rc = ensure_device(*s, H);
    if (rc) return rc;
    rc = upload_polys(*s, H);
    if (rc) return rc;
    if (kind == HARE_KIND_VOXEL) {
        rc = upload_voxel(*s, H);
        return rc ? rc : upload_cell_boxes(*s, H);
    }
    if (kind == HARE_KIND_OCTREE) {
        // the device copy of a leaf carries its first two list entries; that of an interior node the mask of its children that are
        // EMPTY leaves, by octant (OctNode, hare_device.h): popping one has no effect, so K2p / K2d never push it
        std::vector<OctNode> dev(s->oct.nodes);
        const std::vector<OctNode>& host = s->oct.nodes;
        for (size_t k = 0; k < dev.size(); ++k) {
            OctNode& nd = dev[k];
            if (nd.first_child < 0) {
                nd.pad = nd.item_count > 0 ? s->oct.items[(size_t)nd.item_start] : -1;
                nd.first_child = nd.item_count > 1 ? -2 - s->oct.items[(size_t)nd.item_start + 1] : -1;
            } else {
                int32_t empty = 0;
                for (int oct = 0; oct < 8; ++oct) {
                    const OctNode& ch = host[(size_t)nd.first_child + (size_t)oct];
                    if (ch.first_child < 0 && ch.item_count == 0) empty |= 1 << oct;
                }
                nd.pad = empty;
                // ... and the same mask in CURSOR order for each of the eight direction masks m (cursor k examines octant k ^ m), one byte
                // each, in the two list words an interior node does not use: the fast visit of K2p / K2d takes byte m as it is
                uint64_t by_mask = 0;
                for (int m = 0; m < 8; ++m) {
                    uint64_t byte = 0;
                    for (int k = 0; k < 8; ++k) byte |= (uint64_t)((empty >> (k ^ m)) & 1) << k;
                    by_mask |= byte << (8 * m);
                }
                nd.item_start = (int32_t)(uint32_t)(by_mask & 0xFFFFFFFFull);
                nd.item_count = (int32_t)(uint32_t)(by_mask >> 32);
            }
        }
        rc = upload(H, &s->d_oct_nodes, dev.data(), dev.size() * sizeof(OctNode));
        if (rc) return rc;
        // the tight boxes, per topology a query may name (one whose polygon ids the lists stay inside)
        rc = upload_tight_boxes(s, H, s->oct, s->oct.id_count, s->d_oct_tight, s->oct_tight_mid, s->oct_tight_rad);
        if (rc) return rc;
        rc = upload(H, &s->d_oct_items, s->oct.items.data(), s->oct.items.size() * sizeof(int32_t));
        if (rc) return rc;
        reserve_oct_scratch(*s, H);            // the launch path never allocates (nor synchronises the device) after this
        return HARE_OK;
    }
    rc = upload(H, &s->d_kd_nodes, s->kd.nodes.data(), s->kd.nodes.size() * sizeof(KdNodeRec));
    if (rc) return rc;
    rc = upload_tight_boxes(s, H, s->kd, s->kd.id_count, s->d_kd_tight, s->kd_tight_mid, s->kd_tight_rad);
    if (rc) return rc;
    rc = upload_kd_dev_nodes(s, H);
    if (rc) return rc;
    return upload(H, &s->d_kd_items, s->kd.items.data(), s->kd.items.size() * sizeof(int32_t));
}

}  // namespace hare
