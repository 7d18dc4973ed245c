// api.cpp -- the C-ABI of include/hare_hip.h: scene lifetime, device upload, kernel launches.
// Product code; nothing from oracle/.  There is deliberately no CPU shoot path here.
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

// the embedded gfx950 code object (embed.S)
extern "C" const unsigned char hare_kernels_co[];
extern "C" const unsigned char hare_kernels_co_end[];

static_assert(sizeof(hare_ray) == sizeof(hare::RayRec), "hare_ray layout");
static_assert(sizeof(hare_xevent) == sizeof(hare::XEventRec), "hare_xevent layout");
static_assert(sizeof(hare_xevent) == 56 && sizeof(hare_ray) == 48, "wire sizes");
static_assert(sizeof(hare_counters) == hare::CTR_WORDS * 8, "hare_counters layout");

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

#define HIP_TRY(expr)                                                                          \
    do {                                                                                       \
        hipError_t _e = (expr);                                                                \
        if (_e != hipSuccess) {                                                                \
            set_error(std::string(#expr) + " failed: " + (H->GetErrorString ? H->GetErrorString(_e) : "?")); \
            (void)H->GetLastError();                                                           \
            return (_e == hipErrorOutOfMemory) ? HARE_E_NOMEM : HARE_E_HIP;                    \
        }                                                                                      \
    } while (0)

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
        {"hare_kdtree_shoot_count", &m->kdtree_count},
        {"hare_reflect", &m->reflect},
        {"hare_occlusion", &m->occlusion},
        {"hare_voxel_occl_tri", &m->voxel_occl_tri},
        {"hare_voxel_occl_quad", &m->voxel_occl_quad},
        {"hare_voxel_occl_tri_g", &m->voxel_occl_tri_g},
        {"hare_voxel_occl_quad_g", &m->voxel_occl_quad_g},
        {"hare_octree_occl", &m->octree_occl},
        {"hare_events_pack_slim", &m->events_pack_slim},
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
int upload_cell_boxes(Scene& s, const HipApi* H)
{
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
int launch_on_slot(Scene& s, const HipApi* H, hipFunction_t f, unsigned grid, unsigned block, unsigned lds, hipStream_t st, ShootIO& io,
                   void** args, bool coop_tail = false, const OctScratch& oc = OctScratch())
{
    const unsigned idx = s.work_slot.fetch_add(1) % kLaunchSlots;
    Scene::LaunchSlot& sl = s.slots[idx];
    std::lock_guard<std::mutex> lk(sl.mu);
    io.work = reinterpret_cast<unsigned int*>(static_cast<LaunchSlotMem*>(s.d_work) + idx);
    io.coop_tail = (coop_tail && s.opt.coop_tail) ? 1 : 0;
    io.wide_drain = s.opt.wide_drain ? 1 : 0;
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
    if (const char* t = getenv("HARE_VOXEL_ORDER")) o.voxel_order = std::max(0, std::min(2, atoi(t)));
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
int32_t static_chunk_rays(int64_t n, unsigned pgrid, bool keep_a_quarter_for_tickets, bool keep_half = false)
{
    int64_t per_wave = n / ((int64_t)std::max(1u, pgrid) * 4);
    if (keep_half) per_wave = per_wave / 2;          // K2d (swept, tools/k2d_static_sweep.sh): the optimum is half the share at every size below 786k rays
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
enum class Kern { VoxelSimple, VoxelCount, VoxelAudit, VoxelProf, VoxelPool, VoxelPersist, VoxelOccl, OctSimple, OctCount, OctPool, OctPersist, OctDense, OctGroup, OctOccl,
                  KdSimple, KdCount, KdDense, None };
struct KernChoice {
    Kern k = Kern::None;
    const char* name = "";
    hipFunction_t f = nullptr;
};
constexpr unsigned kLdsMax = 160u * 1024u;
constexpr bool kOctreePoolDefault = false;
#ifndef HARE_K2P_WAVES_PER_EU
#define HARE_K2P_WAVES_PER_EU 4
#endif

// flags_only: an occlusion query without events (hare_occluded_* with events == NULL): the hare_*_occl kernels, which write the
// flag and cut the traversal short; the simple kernels (counting, forced, kd-tree) write the flag after the full trace.
KernChoice choose_kernel(const Scene& s, const DeviceModule* M, int32_t kind, size_t top, int64_t n, uint32_t flags, bool flags_only = false)
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
            if ((unsigned)levels * 256u * 20u <= kLdsMax && have(&DeviceModule::octree_occl)) {
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
            const int64_t group_below = (int64_t)cus * 768;           // 196 608 rays on the 256-CU part
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
        const bool dense_ok = !simple && !huge && !flags_only && s.opt.kdtree_kernel != 1 && top < s.d_kd_dev.size() && s.d_kd_dev[top] != nullptr &&
                              kd_dense_lds(s.kd.depth_reached) <= kLdsMax && have(&DeviceModule::kdtree_dense);
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
    const size_t need = std::max(need_group, rec_bytes + tail_spill);
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
    if (int rc = sum_counters(-1)) return rc;            // the per-cast blocks are accumulated into: the totals get what THIS loop adds
    for (int32_t c = 0; c < casts; ++c) {
        hare_xevent* out_c = all ? all + (size_t)c * (size_t)n : last;
        void* ctr_c = d_ctr_casts ? (void*)((hare_counters*)d_ctr_casts + c) : d_ctr;
        const uint32_t f = flags | (c > 0 ? (uint32_t)HARE_SHOOT_RETIRED_RAYS : 0u);
        if (int rc = shoot_device_impl(s, H, kind, top, n, d_rays, c == 0 ? d_e1 : work, c == 0 ? d_e2 : nullptr, f, out_c, ctr_c, st)) return rc;
        if (c + 1 < casts) {
            const void* polys = s.d_polys[(size_t)top];
            const void* ev = out_c;
            void* ex = work;
            long long mm = n;
            void* a[] = {&polys, &d_rays, &ev, &ex, &mm};
            if (int rc = launch(H, M.reflect, (unsigned)((n + 255) / 256), 256, 0, st, a)) return rc;
        }
    }
    if (all && d_last) HIP_TRY(H->MemcpyAsync(d_last, all + (size_t)(casts - 1) * (size_t)n, (size_t)n * sizeof(hare_xevent), hipMemcpyDeviceToDevice, st));
    if (int rc = sum_counters(+1)) return rc;
    return HARE_OK;
}

int shoot_device_impl(Scene& s, const HipApi* H, int32_t kind, int32_t top, int64_t n, void* d_rays,
                      const void* d_e1, const void* d_e2, uint32_t flags, void* d_out, void* d_ctr, hipStream_t st, const void* d_tmax,
                      void* d_occ)
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
    io.flags = flags;
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
            const unsigned plds = lds + (unsigned)kPoolWaves * (unsigned)kPoolWaveBytes;
            // a workgroup per CU whenever the batch has a ray for every wave; the static first chunk is what the batch has for each wave,
            // in steps of 8, at most 128 (a small batch: few rays per wave, each with several lanes from its second round on)
            unsigned pgrid = std::min<unsigned>(cus, (unsigned)((n + kPoolWaves - 1) / kPoolWaves));
            if (pgrid == 0) pgrid = 1;
            const int64_t per_wave = (n + (int64_t)pgrid * kPoolWaves - 1) / ((int64_t)pgrid * kPoolWaves);
            io.static_rays = (int32_t)std::max<int64_t>(8, std::min<int64_t>(128, (per_wave + 7) / 8 * 8));
            if (s.opt.k1p_static_rays > 0) io.static_rays = std::max(8, std::min(1024, s.opt.k1p_static_rays / 8 * 8));   // developer sweeps (tools/k1q_ticket_sweep.py)
            io.ticket_rays = ticket_rays_for(s, n, true);
            if (s.opt.dev && s.opt.dev_order_ptr) io.order = (const uint32_t*)(uintptr_t)s.opt.dev_order_ptr;
            // The order in which K1q takes the rays (order_kernels.hip): inside every window of 4 096 consecutive rays, by an estimate of
            // the walk length -- a pool of rays of similar cost wastes fewer lane-steps (C4 shard -6.8 %, C2 -5.3 % of the kernel's time with
            // the order given; window_sort_*.log), the batch's own locality stays.  Rule ("voxel_order" 1, the default): batches of PRIMARY
            // rays -- no exclusion arrays, not a cast of the bounce loop: reflected rays gain nothing and pay for the indirection
            // (+5 ... +8 %, window_sort_cathedral_bounce5.log) -- from kOrderMinRays: the pass reads every ray once more (12 us per million
            // rays: half of HBM's rate), which at 1M rays is what the order gains.  The scratch is a block of the scene's order ring
            // (stream-ordered allocation was tried first: hipMallocAsync / hipFreeAsync cost the stream more than the pass itself).
            const bool order_rule = s.opt.voxel_order == 2 || (s.opt.voxel_order == 1 && !d_e1 && !d_e2 && !(flags & SHOOT_RETIRED_RAYS) && n >= kOrderMinRays);
            if (!io.order && order_rule && M.cost_order && n <= 0x7FFFFF00ll && (flags & 0xF000u) == 0) {
                // a block of the scene's order ring (scene.h); held under the ring's lock across wait + launches + record
                std::lock_guard<std::mutex> olk(s.order_mu);
                const int ob = (int)(s.order_seq++ % (unsigned)Scene::kOrderRing);
                bool have_block = true;
                if (!s.order_ev[ob] && H->EventCreateWithFlags(&s.order_ev[ob], hipEventDisableTiming) != hipSuccess) { (void)H->GetLastError(); have_block = false; }
                if (have_block && s.order_cap[ob] < (size_t)n) {
                    if (s.order_used[ob]) (void)H->EventSynchronize(s.order_ev[ob]);        // its previous user has finished before it is replaced
                    dev_free(H, s.d_order[ob]);
                    s.order_cap[ob] = 0;
                    s.order_used[ob] = false;
                    const size_t cap = ((size_t)n + 65535u) & ~(size_t)65535u;
                    if (H->Malloc(&s.d_order[ob], cap * sizeof(uint32_t)) == hipSuccess) s.order_cap[ob] = cap;
                    else { (void)H->GetLastError(); s.d_order[ob] = nullptr; have_block = false; }      // no scratch: the cast runs in the caller's order
                }
                if (have_block) {
                    if (s.order_used[ob]) HIP_TRY(H->StreamWaitEvent(st, s.order_ev[ob], 0));
                    const void* rp = d_rays;
                    void* d_order = s.d_order[ob];
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
            const bool dense_k = f != nullptr && (f == M.octree_dense || f == M.octree_dense_own);
            const unsigned plds = (unsigned)g.max_depth * 256u * 20u + (dense_k ? kOctDenseExtra : 0u);
            unsigned per_cu = std::min((unsigned)HARE_K2P_WAVES_PER_EU, std::max(1u, (unsigned)(kLdsMax / plds)));
            unsigned pgrid = cus * per_cu;
            pgrid = std::min<unsigned>(pgrid, (unsigned)((m + 63) / 64 + 3) / 4);
            if (pgrid == 0) pgrid = 1;
            // an octree ray costs ~10x a voxel ray: ticket atomics never bind.  K2p: 32 rays; K2d finishes rays sooner and likes 16
            // (8 / 16 / 24 / 32 rays per ticket: 1M rays 478 / 502 / 466 / 489 Mrays/s, 1.5M 553 / 570 / 569 / 563, 524k 353 / 354 / 348 / 346)
            sub.ticket_rays = s.opt.ticket_rays > 0 ? std::max(8, std::min(4096, s.opt.ticket_rays)) : (dense_k ? 16 : 32);
            sub.static_rays = static_chunk_rays(m, pgrid, true, dense_k);    // K2p: 262k rays 2.607 -> 1.861 ms, 524k 2.569 -> 2.336; K2d: half
                                                                             // the share (393k rays 338 -> 418 Mrays/s, 524k 421 -> 474, 655k 393 -> 515)
            if (s.opt.k2p_static_rays > 0) sub.static_rays = std::max(32, std::min(256, s.opt.k2p_static_rays / 32 * 32));   // developer sweeps
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

using namespace hare;

extern "C" {

const char* hare_version(void) { return "hare_hip 0.1 (gfx950)"; }
const char* hare_last_error(void) { return last_error(); }

const char* hare_hip_runtime_path(void)
{
    const HipApi* H = api_or_err();
    return H ? H->path.c_str() : "";
}

int hare_device_count(int32_t* count)
{
    if (!count) {
        set_error("hare_device_count: null argument");
        return HARE_E_INVALID;
    }
    *count = 0;
    const HipApi* H = api_or_err();
    if (!H) return HARE_E_NODEVICE;
    int n = 0;
    if (H->GetDeviceCount(&n) != hipSuccess) n = 0;
    *count = n;
    return HARE_OK;
}

int hare_polygon_normals(const double* verts, const int32_t* nverts, int32_t P, double* normals_out)
{
    if (P < 0 || (P > 0 && (!verts || !nverts || !normals_out))) {
        set_error("hare_polygon_normals: bad arguments");
        return HARE_E_INVALID;
    }
    polygon_normals(verts, nverts, P, normals_out);
    return HARE_OK;
}

int hare_topology_bounds(const double* verts, const int32_t* nverts, int32_t P, double mn[3], double mx[3])
{
    if (P < 0 || (P > 0 && (!verts || !nverts)) || !mn || !mx) {
        set_error("hare_topology_bounds: bad arguments");
        return HARE_E_INVALID;
    }
    topology_bounds(verts, nverts, P, mn, mx);
    return HARE_OK;
}

int hare_topology_ingest(const double* soup, const int32_t* nverts, int32_t P, double* verts_out, int32_t* corner_vertex,
                         double* vertices_out, int32_t* n_vertices_out)
{
    if (n_vertices_out) *n_vertices_out = 0;
    if (P < 0 || (P > 0 && (!soup || !nverts || !verts_out))) {
        set_error("hare_topology_ingest: bad arguments");
        return HARE_E_INVALID;
    }
    for (int32_t p = 0; p < P; ++p)
        if (nverts[p] != 3 && nverts[p] != 4) {
            set_error("hare_topology_ingest: Hare does not support polygons of other than 3 or 4 sides");
            return HARE_E_UNSUPPORTED;
        }
    try {
        std::vector<double> vertices;
        const int nv = topology_ingest(soup, nverts, P, verts_out, corner_vertex, vertices);
        if (vertices_out && nv > 0) memcpy(vertices_out, vertices.data(), (size_t)nv * 3 * sizeof(double));
        if (n_vertices_out) *n_vertices_out = nv;
    } catch (const std::bad_alloc&) {
        set_error("hare_topology_ingest: out of memory");
        return HARE_E_NOMEM;
    } catch (...) {
        set_error("hare_topology_ingest: unexpected exception");
        return HARE_E_INVALID;
    }
    return HARE_OK;
}

int hare_scene_create(const hare_topology_desc* topos, int32_t n_topos, int32_t device, hare_scene** out)
{
    if (!out) {
        set_error("hare_scene_create: null out");
        return HARE_E_INVALID;
    }
    *out = nullptr;
    if (!topos || n_topos < 1 || n_topos > 64 || device < 0) {
        set_error("hare_scene_create: need 1..64 topologies and device >= 0");
        return HARE_E_INVALID;
    }
    try {
        std::unique_ptr<hare_scene> s(new hare_scene());
        s->device = device;
        read_env_options(s->opt);          // the only place the environment is read for this scene
        s->topos.resize(n_topos);
        for (int32_t m = 0; m < n_topos; ++m) {
            const hare_topology_desc& d = topos[m];
            if (d.P < 0 || (d.P > 0 && (!d.verts || !d.nverts || !d.normals))) {
                set_error("hare_scene_create: topology with null arrays");
                return HARE_E_INVALID;
            }
            Topo& T = s->topos[m];
            T.P = d.P;
            T.verts.assign(d.verts, d.verts + (size_t)d.P * 12);
            T.nverts.assign(d.nverts, d.nverts + d.P);
            T.normals.assign(d.normals, d.normals + (size_t)d.P * 3);
            for (int a = 0; a < 3; ++a) {
                T.mn[a] = d.min[a];
                T.mx[a] = d.max[a];
            }
            for (int32_t p = 0; p < d.P; ++p) {
                if (T.nverts[p] == 4) T.has_quads = true;
                else if (T.nverts[p] != 3) {
                    // Topology.Build_Topology throws NotImplementedException (Hare_Geometry_Topology.cs:298)
                    set_error("Hare Does not yet support polygons of more than 4 sides.");
                    return HARE_E_UNSUPPORTED;
                }
            }
        }
        *out = s.release();
        return HARE_OK;
    } catch (const std::bad_alloc&) {
        set_error("hare_scene_create: out of host memory");
        return HARE_E_NOMEM;
    } catch (...) {
        set_error("hare_scene_create: unexpected failure");
        return HARE_E_INVALID;
    }
}

void hare_scene_destroy(hare_scene* s)
{
    if (!s) return;
    std::string e;
    const HipApi* H = hip_api(&e);
    free_host_mirror(*s);
    if (H && (s->module || s->stream)) {
        DeviceGuard dev_guard(H, s->device);   // act on the scene's device, leave the caller's current device as it was
        if (s->stream) (void)H->StreamSynchronize(s->stream);
        for (auto* v : {&s->d_polys, &s->d_cull, &s->d_quads, &s->d_cells, &s->d_items, &s->d_occ, &s->d_cellbox})
            for (void*& p : *v) dev_free(H, p);
        for (void** p : {&s->d_oct_nodes, &s->d_oct_items, &s->d_kd_nodes, &s->d_kd_items, &s->d_work, &s->d_oct_tail})
            dev_free(H, *p);
        for (void*& p : s->d_oct_tight) dev_free(H, p);
        for (void*& p : s->d_kd_tight) dev_free(H, p);
        for (void*& p : s->d_kd_dev) dev_free(H, p);
        free_bounce_buffers(H, *s);
        for (Scene::BatchCtx& c : s->ctx) {
            for (hipStream_t& x : c.st)
                if (x) { (void)H->StreamSynchronize(x); (void)H->StreamDestroy(x); x = nullptr; }
            for (void** p : {&c.d_rays, &c.d_e1, &c.d_e2, &c.d_out, &c.d_ctr, &c.d_tmax, &c.d_occ, &c.d_slim}) dev_free(H, *p);
        }
        for (int k = 0; k < kOctScratchRing; ++k) {
            dev_free(H, s->d_oct_scratch[k]);
            if (s->oct_scratch_ev[k]) (void)H->EventDestroy(s->oct_scratch_ev[k]);
        }
        for (Scene::LaunchSlot& sl : s->slots)
            if (sl.ev) { (void)H->EventSynchronize(sl.ev); (void)H->EventDestroy(sl.ev); sl.ev = nullptr; }
        for (hipEvent_t& e : s->oct_tail_ev)
            if (e) { (void)H->EventSynchronize(e); (void)H->EventDestroy(e); e = nullptr; }
        for (int k = 0; k < Scene::kOrderRing; ++k) {
            if (s->order_ev[k]) { (void)H->EventSynchronize(s->order_ev[k]); (void)H->EventDestroy(s->order_ev[k]); s->order_ev[k] = nullptr; }
            dev_free(H, s->d_order[k]);
        }
        if (s->stream) (void)H->StreamDestroy(s->stream);
    }
    delete s;
}

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

extern "C++" {
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
static int upload_tight_boxes(hare_scene* s, const HipApi* H, const Tree& tree, int32_t id_count, std::vector<void*>& tight, double mid[3], double& rad)
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
static int upload_kd_dev_nodes(hare_scene* s, const HipApi* H)
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
}  // extern "C++"

// After a host build: push the partition to the device when one is available.  Builds succeed
// without a GPU (introspection works); shooting then fails with HARE_E_NODEVICE.
static int sync_partition_to_device(hare_scene* s, int kind)
{
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

// Build the grid on the GPU when a device is present (HARE_BUILD=host forces the host builder; both
// produce identical lists).  *on_gpu = false: the caller runs the host builder.
static int try_gpu_voxel_build(hare_scene* s, int32_t domain, int32_t max_domain, int32_t avg_polys, bool* on_gpu)
{
    *on_gpu = false;
    if (s->opt.build_host) return HARE_OK;
    std::string e;
    const HipApi* H = hip_api(&e);
    int n = 0;
    if (!H || H->GetDeviceCount(&n) != hipSuccess || n <= 0) return HARE_OK;
    int rc = ensure_device(*s, H);
    if (rc) return rc;
    rc = upload_polys(*s, H);
    if (rc) return rc;
    if (domain > 0) return gpu_build_voxel_fixed(*s, H, domain, on_gpu);
    return gpu_build_voxel_adaptive(*s, H, max_domain, avg_polys, on_gpu);
}

int hare_voxel_build(hare_scene* s, int32_t domain)
{
    if (!s) {
        set_error("null scene");
        return HARE_E_INVALID;
    }
    GUARD_BEGIN
    DeviceGuard dev_guard(hip_api(nullptr), s->device);
    if (domain < 1 || domain > 1024) {
        set_error("hare_voxel_build: domain must be in [1, 1024]");
        return HARE_E_INVALID;
    }
    bool on_gpu = false;
    free_host_mirror(*s);
    int rc = try_gpu_voxel_build(s, domain, 0, 0, &on_gpu);
    if (rc) return rc;
    if (on_gpu) return upload_cell_boxes(*s, hip_api(nullptr));
    rc = build_voxel_fixed(*s, domain);
    if (rc) return rc;
    return sync_partition_to_device(s, HARE_KIND_VOXEL);
    GUARD_END
}

int hare_voxel_build_adaptive(hare_scene* s, int32_t max_domain, int32_t avg_polys)
{
    if (!s) {
        set_error("null scene");
        return HARE_E_INVALID;
    }
    GUARD_BEGIN
    DeviceGuard dev_guard(hip_api(nullptr), s->device);
    if (max_domain < 1 || max_domain > 10) {
        set_error("hare_voxel_build_adaptive: max_domain must be in [1, 10]");
        return HARE_E_INVALID;
    }
    bool on_gpu = false;
    free_host_mirror(*s);
    int rc = try_gpu_voxel_build(s, 0, max_domain, avg_polys, &on_gpu);
    if (rc) return rc;
    if (on_gpu) return upload_cell_boxes(*s, hip_api(nullptr));
    rc = build_voxel_adaptive(*s, max_domain, avg_polys);
    if (rc) return rc;
    return sync_partition_to_device(s, HARE_KIND_VOXEL);
    GUARD_END
}

int hare_octree_build(hare_scene* s, int32_t max_depth, int32_t max_polys)
{
    if (!s) {
        set_error("null scene");
        return HARE_E_INVALID;
    }
    GUARD_BEGIN
    DeviceGuard dev_guard(hip_api(nullptr), s->device);
    int rc = octree_check_args(*s, max_depth, max_polys);
    if (rc) return rc;
    bool on_gpu = false;
    {   // SAT binning on the GPU when there is one (HARE_BUILD=host forces the host builder; identical output)
        std::string e;
        const HipApi* H = s->opt.build_host ? nullptr : hip_api(&e);
        int n = 0;
        if (H && H->GetDeviceCount(&n) == hipSuccess && n > 0) {
            rc = ensure_device(*s, H);
            if (rc) return rc;
            rc = upload_polys(*s, H);
            if (rc) return rc;
            rc = gpu_build_octree(*s, H, max_depth, max_polys, &on_gpu);
            if (rc) return rc;
        }
    }
    if (!on_gpu) {
        rc = build_octree(*s, max_depth, max_polys);
        if (rc) return rc;
    }
    s->oct_levels = octree_levels(s->oct);
    free_host_mirror(*s);
    return sync_partition_to_device(s, HARE_KIND_OCTREE);
    GUARD_END
}

int hare_kdtree_build(hare_scene* s, int32_t max_depth, int32_t max_polys)
{
    if (!s) {
        set_error("null scene");
        return HARE_E_INVALID;
    }
    GUARD_BEGIN
    DeviceGuard dev_guard(hip_api(nullptr), s->device);
    free_host_mirror(*s);
    int rc = build_kdtree(*s, max_depth, max_polys);
    if (rc) return rc;
    return sync_partition_to_device(s, HARE_KIND_KDTREE);
    GUARD_END
}

int hare_voxel_get_info(const hare_scene* s, hare_voxel_info* out)
{
    if (!s || !out) {
        set_error("null argument");
        return HARE_E_INVALID;
    }
    if (!s->vox.built) {
        set_error("voxel grid not built");
        return HARE_E_STATE;
    }
    memset(out, 0, sizeof *out);
    out->ct = s->vox.ct;
    out->n_topos = (int32_t)s->topos.size();
    for (int a = 0; a < 3; ++a) {
        out->obox_min[a] = s->vox.omin[a];
        out->obox_max[a] = s->vox.omax[a];
        out->box_dims[a] = s->vox.box_dims[a];
        out->voxel_dims[a] = s->vox.vd[a];
    }
    out->char_step = s->vox.char_step;
    out->total_items = s->vox.items[0].size();
    out->built_on_device = s->vox.on_device ? 1 : 0;
    return HARE_OK;
}

int hare_voxel_get_lists(const hare_scene* s, int32_t top, uint32_t* cell_start, int32_t* items)
{
    if (!s || !cell_start || top < 0 || top >= (int32_t)s->topos.size()) {
        set_error("bad argument");
        return HARE_E_INVALID;
    }
    if (!s->vox.built) {
        set_error("voxel grid not built");
        return HARE_E_STATE;
    }
    memcpy(cell_start, s->vox.start[top].data(), s->vox.start[top].size() * sizeof(uint32_t));
    if (items && !s->vox.items[top].empty())
        memcpy(items, s->vox.items[top].data(), s->vox.items[top].size() * sizeof(int32_t));
    return HARE_OK;
}

int hare_octree_get_info(const hare_scene* s, hare_tree_info* out)
{
    if (!s || !out) {
        set_error("null argument");
        return HARE_E_INVALID;
    }
    if (!s->oct.built) {
        set_error("octree not built");
        return HARE_E_STATE;
    }
    memset(out, 0, sizeof *out);
    out->n_nodes = (int32_t)s->oct.nodes.size();
    out->max_depth = s->oct.max_depth;
    out->max_polys = s->oct.max_polys;
    out->built_on_device = s->oct.built_on_device ? 1 : 0;
    out->total_items = s->oct.items.size();
    return HARE_OK;
}

int hare_octree_get_nodes(const hare_scene* s, double* boxes, int32_t* first_child, int32_t* item_start,
                          int32_t* item_count, int32_t* items)
{
    if (!s || !boxes || !first_child || !item_start || !item_count || !items) {
        set_error("null argument");
        return HARE_E_INVALID;
    }
    if (!s->oct.built) {
        set_error("octree not built");
        return HARE_E_STATE;
    }
    for (size_t i = 0; i < s->oct.nodes.size(); ++i) {
        const OctNode& n = s->oct.nodes[i];
        for (int a = 0; a < 3; ++a) {
            boxes[6 * i + a] = n.bmin[a];
            boxes[6 * i + 3 + a] = n.bmax[a];
        }
        first_child[i] = n.first_child;
        item_start[i] = n.item_start;
        item_count[i] = n.item_count;
    }
    if (!s->oct.items.empty()) memcpy(items, s->oct.items.data(), s->oct.items.size() * sizeof(int32_t));
    return HARE_OK;
}

int hare_kdtree_get_info(const hare_scene* s, hare_tree_info* out)
{
    if (!s || !out) {
        set_error("null argument");
        return HARE_E_INVALID;
    }
    if (!s->kd.built) {
        set_error("kd-tree not built");
        return HARE_E_STATE;
    }
    memset(out, 0, sizeof *out);
    out->n_nodes = (int32_t)s->kd.nodes.size();
    out->max_depth = s->kd.max_depth;
    out->max_polys = s->kd.max_polys;
    out->total_items = s->kd.items.size();
    return HARE_OK;
}

int hare_kdtree_get_nodes(const hare_scene* s, double* boxes, double* split, int32_t* axis, int32_t* left,
                          int32_t* right, int32_t* item_start, int32_t* item_count, int32_t* items)
{
    if (!s || !boxes || !split || !axis || !left || !right || !item_start || !item_count || !items) {
        set_error("null argument");
        return HARE_E_INVALID;
    }
    if (!s->kd.built) {
        set_error("kd-tree not built");
        return HARE_E_STATE;
    }
    for (size_t i = 0; i < s->kd.nodes.size(); ++i) {
        const KdNodeRec& n = s->kd.nodes[i];
        for (int a = 0; a < 3; ++a) {
            boxes[6 * i + a] = n.bmin[a];
            boxes[6 * i + 3 + a] = n.bmax[a];
        }
        split[i] = n.split;
        axis[i] = n.left < 0 ? 0 : n.axis;   // the C# field defaults to 0 on leaves
        left[i] = n.left;
        right[i] = n.right;
        item_start[i] = n.item_start;
        item_count[i] = n.item_count;
    }
    if (!s->kd.items.empty()) memcpy(items, s->kd.items.data(), s->kd.items.size() * sizeof(int32_t));
    return HARE_OK;
}

int hare_shoot_device(hare_scene* s, int32_t kind, int32_t top_index, int64_t n, void* d_rays, const void* d_excl1,
                      const void* d_excl2, uint32_t flags, void* d_out, void* d_counters, void* stream)
{
    if (!s) {
        set_error("null scene");
        return HARE_E_INVALID;
    }
    GUARD_BEGIN
    const HipApi* H = api_or_err();
    if (!H) return HARE_E_NODEVICE;
    // the launch and the ticket memset must be issued with the scene's device current, whatever the calling thread had
    // selected (a torch device guard, another scene); the guard puts the caller's device back afterwards
    DeviceGuard dev_guard(H, s->device);
    if (!s->module) {
        int rc = ensure_device(*s, H);
        if (rc) return rc;
    }
    // slim records are a format of the host-buffer calls (they are packed from the events in a staging buffer)
    return shoot_device_impl(*s, H, kind, top_index, n, d_rays, d_excl1, d_excl2, flags & ~HARE_SHOOT_SLIM_EVENTS, d_out, d_counters,
                             (hipStream_t)stream);
    GUARD_END
}

// hare_shoot_batch and hare_occluded_batch: host buffers in, host buffers out, pipelined over up to eight chunks.
//   out != null, occluded == null   closest-hit events (hare_shoot_batch)
//   out != null, occluded != null   events and the occlusion flags derived from them
//   out == null, occluded != null   flags only: the t_max-bounded kernels; 4 bytes per ray come back instead of 56
static int batch_impl(hare_scene* s, int32_t kind, int32_t top_index, int64_t n, hare_ray* rays, const int32_t* excl1,
                      const int32_t* excl2, uint32_t flags, hare_xevent* out, hare_counters* ctr, const double* tmax, int32_t* occluded)
{
    GUARD_BEGIN
    // host-buffer callers get the reference's meaning of poly_origin: an index that matches no polygon (any negative
    // value) excludes nothing.  Only the device-resident bounce loop (hare_reflect_device + hare_shoot_device) may
    // retire rays, so the retire flag never passes here, nor do developer bits.
    // (The developer modes 0x1000 / 0x2000 / 0x4000 write past the 64-byte counter block they are given: here that block is one of 16
    // in a staging array, so they never pass.  The cull audit, 0x8000, stays inside the block -- words 5..7 -- and is what
    // tests/test_gpu_parity.py::test_fp32_cull_never_rejects_a_hit runs through this call on a scene with the `dev` option.)
    flags = sanitize_flags(*s, flags) & ~HARE_SHOOT_RETIRED_RAYS & ~0x7000u;
    if ((flags & HARE_SHOOT_SLIM_EVENTS) && (flags & HARE_SHOOT_WRITEBACK_ORIGIN) && out) {
        // a slim record of a moved ray (hit == 2) holds t from the MOVED origin and hare_expand_events redoes the move from the
        // ORIGINAL one; with the write-back the caller's rays[] would already hold the moved origins and the rebuilt t would
        // silently lack t_start.  One call cannot have both.
        set_error("hare_shoot_batch: HARE_SHOOT_SLIM_EVENTS cannot be combined with HARE_SHOOT_WRITEBACK_ORIGIN "
                  "(hare_expand_events needs the rays as they were passed in)");
        return HARE_E_INVALID;
    }
    DeviceGuard dev_guard(hip_api(nullptr), s->device);
    const HipApi* H = nullptr;
    Scene::BatchCtx* c = nullptr;
    {
        std::unique_lock<std::mutex> lk(s->mu);
        int rc = ensure_device(*s, H);
        if (rc) return rc;
        if (ctr) memset(ctr, 0, sizeof *ctr);
        if (n == 0) return HARE_OK;
        // a free staging context, or wait for one: concurrent callers (Pachyderm's worker threads) run side by side
        s->cv.wait(lk, [&] { for (Scene::BatchCtx& x : s->ctx) if (!x.busy) return true; return false; });
        for (Scene::BatchCtx& x : s->ctx)
            if (!x.busy && x.cap >= n) { c = &x; break; }          // prefer one that is already large enough
        if (!c)
            for (Scene::BatchCtx& x : s->ctx)
                if (!x.busy) { c = &x; break; }
        c->busy = true;
    }
    struct Release {
        hare_scene* s; Scene::BatchCtx* c;
        ~Release() { { std::lock_guard<std::mutex> lk(s->mu); c->busy = false; } s->cv.notify_one(); }
    } release{s, c};
    constexpr int kMaxChunks = 16;
    if (n > c->cap) {
        for (void** p : {&c->d_rays, &c->d_e1, &c->d_e2, &c->d_out, &c->d_tmax, &c->d_occ, &c->d_slim}) dev_free(H, *p);
        c->cap = 0;
        c->occ_cap = 0;
        HIP_TRY(H->Malloc(&c->d_rays, (size_t)n * sizeof(hare_ray)));
        HIP_TRY(H->Malloc(&c->d_e1, (size_t)n * sizeof(int32_t)));
        HIP_TRY(H->Malloc(&c->d_e2, (size_t)n * sizeof(int32_t)));
        HIP_TRY(H->Malloc(&c->d_out, (size_t)n * sizeof(hare_xevent)));
        c->cap = n;
    }
    const bool slim = out && (flags & HARE_SHOOT_SLIM_EVENTS) != 0;
    const bool slim_uv = kind != HARE_KIND_VOXEL;                 // the trees return u, v: 32-byte records
    const size_t slim_bytes = slim_uv ? sizeof(hare_slim_event_uv) : sizeof(hare_slim_event);
    if (slim) {
        if (!s->module->events_pack_slim) {
            set_error("hare_shoot_batch: slim-event kernel missing from code object");
            return HARE_E_STATE;
        }
        if (!c->d_slim) HIP_TRY(H->Malloc(&c->d_slim, (size_t)c->cap * sizeof(hare_slim_event_uv)));
    }
    if (occluded && n > c->occ_cap) {
        for (void** p : {&c->d_tmax, &c->d_occ}) dev_free(H, *p);
        c->occ_cap = 0;
        HIP_TRY(H->Malloc(&c->d_tmax, (size_t)n * sizeof(double)));
        HIP_TRY(H->Malloc(&c->d_occ, (size_t)n * sizeof(int32_t)));
        c->occ_cap = n;
    }
    if (!c->d_ctr) HIP_TRY(H->Malloc(&c->d_ctr, kMaxChunks * sizeof(hare_counters)));
    // A large batch is pipelined as chunks, each on its own stream and driven by its own host thread: upload, kernel and download of
    // different chunks overlap (both directions of the host link busy).  Round 2 measured three chunks as the best (402 -> 499 Mrays/s
    // for 1M rays) because small launches were inefficient; since the pool kernel serves small launches well (round 3) five chunks are
    // (1M rays: full records 525 -> 572 Mrays/s, 4M: 541 -> 596), and eight for the longest batches with 16-byte slim records (4M:
    // 870 -> 985); more lose again to thread and launch overheads (profiles/r03_experiments/host_batch_chunks.log).
    int K = n < 196608 ? 1 : (n < 393216 ? 3 : (n < 2097152 ? 5 : 8));
    if (s->opt.batch_chunks > 0) K = std::max(1, std::min(kMaxChunks, s->opt.batch_chunks));     // developer sweeps
    for (int k = 0; k < K; ++k)
        if (!c->st[k]) HIP_TRY(H->StreamCreate(&c->st[k]));
    hare_counters parts[kMaxChunks];
    memset(parts, 0, sizeof parts);
    int rcs[kMaxChunks];
    for (int& r : rcs) r = HARE_OK;
    std::string errs[kMaxChunks];
    auto chunk_body = [&](int k, hipStream_t st) -> int {
        const int64_t lo = (int64_t)((__int128)n * k / K), m = (int64_t)((__int128)n * (k + 1) / K) - lo;
        if (m == 0) return HARE_OK;
        hare_ray* dr = (hare_ray*)c->d_rays + lo;
        int32_t* de1 = (int32_t*)c->d_e1 + lo;
        int32_t* de2 = (int32_t*)c->d_e2 + lo;
        hare_xevent* dout = out ? (hare_xevent*)c->d_out + lo : nullptr;
        hare_counters* dctr = (hare_counters*)c->d_ctr + k;
        double* dtm = (occluded && tmax) ? (double*)c->d_tmax + lo : nullptr;
        int32_t* docc = occluded ? (int32_t*)c->d_occ + lo : nullptr;
        HIP_TRY(H->MemcpyAsync(dr, rays + lo, (size_t)m * sizeof(hare_ray), hipMemcpyHostToDevice, st));
        if (dtm) HIP_TRY(H->MemcpyAsync(dtm, tmax + lo, (size_t)m * sizeof(double), hipMemcpyHostToDevice, st));
        if (excl1) HIP_TRY(H->MemcpyAsync(de1, excl1 + lo, (size_t)m * sizeof(int32_t), hipMemcpyHostToDevice, st));
        if (excl2) HIP_TRY(H->MemcpyAsync(de2, excl2 + lo, (size_t)m * sizeof(int32_t), hipMemcpyHostToDevice, st));
        HIP_TRY(H->MemsetAsync(dctr, 0, sizeof(hare_counters), st));
        const int r = shoot_device_impl(*s, H, kind, top_index, m, dr, excl1 ? de1 : nullptr, excl2 ? de2 : nullptr, flags, dout, dctr, st, dtm, docc);
        if (r) return r;
        if (slim) {      // pack {t[, u, v], poly_id, hit} on the device; 16 (32) bytes per ray cross the link instead of 56
            unsigned char* dsl = (unsigned char*)c->d_slim + (size_t)lo * slim_bytes;
            const void* evp = dout;
            long long mm = m;
            int uv = slim_uv ? 1 : 0;
            void* pargs[] = {&evp, &mm, &uv, &dsl};
            if (int pr = launch(H, s->module->events_pack_slim, (unsigned)((m + 255) / 256), 256, 0, st, pargs)) return pr;
            HIP_TRY(H->MemcpyAsync((unsigned char*)out + (size_t)lo * slim_bytes, dsl, (size_t)m * slim_bytes, hipMemcpyDeviceToHost, st));
        } else if (out) HIP_TRY(H->MemcpyAsync(out + lo, dout, (size_t)m * sizeof(hare_xevent), hipMemcpyDeviceToHost, st));
        if (occluded) HIP_TRY(H->MemcpyAsync(occluded + lo, docc, (size_t)m * sizeof(int32_t), hipMemcpyDeviceToHost, st));
        if (flags & HARE_SHOOT_WRITEBACK_ORIGIN)
            HIP_TRY(H->MemcpyAsync(rays + lo, dr, (size_t)m * sizeof(hare_ray), hipMemcpyDeviceToHost, st));
        HIP_TRY(H->MemcpyAsync(&parts[k], dctr, sizeof(hare_counters), hipMemcpyDeviceToHost, st));
        HIP_TRY(H->StreamSynchronize(st));
        return HARE_OK;
    };
    auto chunk = [&](int k) -> int {
        DeviceGuard g(H, s->device);                            // the current device is per host thread
        hipStream_t st = c->st[k];
        const int r = chunk_body(k, st);
        // a failed step leaves earlier async copies into the caller's buffers in flight: drain them before the
        // error reaches a caller who may free those buffers
        if (r != HARE_OK) (void)H->StreamSynchronize(st);
        return r;
    };
    auto guarded = [&](int k) {
        try {
            rcs[k] = chunk(k);
        } catch (...) {
            rcs[k] = HARE_E_NOMEM;
            set_error("hare_shoot_batch: exception in a chunk");
        }
        if (rcs[k] != HARE_OK) errs[k] = hare_last_error();     // thread-local: carry it to the caller's thread
    };
    {
        std::vector<std::thread> workers;
        workers.reserve((size_t)K);
        for (int k = 1; k < K; ++k) {
            try {
                workers.emplace_back(guarded, k);
            } catch (...) {
                guarded(k);          // no thread to be had: run the chunk here (a started thread must never be left unjoined)
            }
        }
        guarded(0);
        for (auto& w : workers) w.join();
    }
    for (int k = 0; k < K; ++k)
        if (rcs[k] != HARE_OK) {
            set_error(errs[k]);
            return rcs[k];
        }
    if (ctr) {
        uint64_t* dst = reinterpret_cast<uint64_t*>(ctr);
        for (int k = 0; k < K; ++k) {
            const uint64_t* src = reinterpret_cast<const uint64_t*>(&parts[k]);
            for (size_t w = 0; w < sizeof(hare_counters) / 8; ++w) dst[w] += src[w];
        }
    }
    return HARE_OK;
    GUARD_END
}

int hare_shoot_batch(hare_scene* s, int32_t kind, int32_t top_index, int64_t n, hare_ray* rays, const int32_t* excl1,
                     const int32_t* excl2, uint32_t flags, hare_xevent* out, hare_counters* ctr)
{
    if (!s) {
        set_error("null scene");
        return HARE_E_INVALID;
    }
    if (n < 0 || (n > 0 && (!rays || !out))) {
        set_error("hare_shoot_batch: bad arguments");
        return HARE_E_INVALID;
    }
    return batch_impl(s, kind, top_index, n, rays, excl1, excl2, flags, out, ctr, nullptr, nullptr);
}

int hare_shoot_batch_sharded(hare_scene* const* scenes, int32_t n_scenes, int32_t kind, int32_t top_index, int64_t n,
                             hare_ray* rays, const int32_t* excl1, const int32_t* excl2, uint32_t flags, hare_xevent* out,
                             hare_counters* ctr)
{
    if (!scenes || n_scenes < 1 || n_scenes > 64) {
        set_error("hare_shoot_batch_sharded: need 1..64 scenes");
        return HARE_E_INVALID;
    }
    for (int32_t k = 0; k < n_scenes; ++k)
        if (!scenes[k]) {
            set_error("hare_shoot_batch_sharded: null scene");
            return HARE_E_INVALID;
        }
    if (n < 0 || (n > 0 && (!rays || !out))) {
        set_error("hare_shoot_batch_sharded: bad arguments");
        return HARE_E_INVALID;
    }
    GUARD_BEGIN
    flags &= ~HARE_SHOOT_RETIRED_RAYS;      // each shard's hare_shoot_batch masks the rest by its own scene's options
    if (ctr) memset(ctr, 0, sizeof *ctr);
    const int G = n_scenes;
    std::vector<int> rcs((size_t)G, HARE_OK);
    std::vector<std::string> errs((size_t)G);
    std::vector<hare_counters> parts((size_t)G);
    auto shard = [&](int k) {
        const int64_t lo = (int64_t)((__int128)n * k / G), hi = (int64_t)((__int128)n * (k + 1) / G);
        memset(&parts[k], 0, sizeof parts[k]);
        // with HARE_SHOOT_SLIM_EVENTS `out` is an array of 16- or 32-byte records, not of X_Events
        const size_t rec = !(flags & HARE_SHOOT_SLIM_EVENTS) ? sizeof(hare_xevent)
                                                              : (kind == HARE_KIND_VOXEL ? sizeof(hare_slim_event) : sizeof(hare_slim_event_uv));
        hare_xevent* o_k = out ? reinterpret_cast<hare_xevent*>(reinterpret_cast<unsigned char*>(out) + (size_t)lo * rec) : nullptr;
        rcs[k] = hare_shoot_batch(scenes[k], kind, top_index, hi - lo, rays ? rays + lo : nullptr, excl1 ? excl1 + lo : nullptr,
                                  excl2 ? excl2 + lo : nullptr, flags, o_k, &parts[k]);
        if (rcs[k] != HARE_OK) errs[k] = hare_last_error();     // thread-local: carry it to the caller's thread
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
        if (rcs[k] != HARE_OK) {
            set_error("shard " + std::to_string(k) + ": " + errs[k]);
            return rcs[k];
        }
    if (ctr)
        for (int k = 0; k < G; ++k) {
            ctr->rays += parts[k].rays;
            ctr->hits += parts[k].hits;
            ctr->cells += parts[k].cells;
            ctr->entries += parts[k].entries;
            ctr->tests += parts[k].tests;
        }
    return HARE_OK;
    GUARD_END
}

int hare_reflect_device(hare_scene* s, int32_t top_index, int64_t n, void* d_rays, const void* d_events,
                        void* d_excl_out, void* stream)
{
    if (!s) {
        set_error("null scene");
        return HARE_E_INVALID;
    }
    if (n < 0 || top_index < 0 || top_index >= (int32_t)s->topos.size() || (n > 0 && (!d_rays || !d_events || !d_excl_out))) {
        set_error("hare_reflect_device: bad arguments");
        return HARE_E_INVALID;
    }
    GUARD_BEGIN
    DeviceGuard dev_guard(hip_api(nullptr), s->device);
    const HipApi* H = nullptr;
    int rc = ensure_device(*s, H);
    if (rc) return rc;
    rc = upload_polys(*s, H);
    if (rc) return rc;
    if (n == 0) return HARE_OK;
    if (!s->module->reflect) {
        set_error("hare_reflect_device: kernel missing from code object");
        return HARE_E_STATE;
    }
    const void* polys = s->d_polys[top_index];
    void* args[] = {&polys, &d_rays, &d_events, &d_excl_out, &n};
    const unsigned block = 256;
    return launch(H, s->module->reflect, (unsigned)((n + block - 1) / block), block, 0, (hipStream_t)stream, args);
    GUARD_END
}

int hare_bounce_device(hare_scene* s, int32_t kind, int32_t top_index, int64_t n, void* d_rays, const void* d_excl1, const void* d_excl2,
                       int32_t bounces, uint32_t flags, void* d_work, void* d_events_all, void* d_events_last, void* d_counters,
                       void* d_counters_per_cast, void* stream)
{
    if (!s) {
        set_error("null scene");
        return HARE_E_INVALID;
    }
    GUARD_BEGIN
    const HipApi* H = api_or_err();
    if (!H) return HARE_E_NODEVICE;
    DeviceGuard dev_guard(H, s->device);
    if (!s->module) {
        int rc = ensure_device(*s, H);
        if (rc) return rc;
    }
    if (int rc = upload_polys(*s, H)) return rc;
    if (n > 0) {
        const size_t rb = (size_t)n * sizeof(hare_ray), ob = (size_t)n * sizeof(hare_xevent), wb = (size_t)n * 2 * sizeof(int32_t),
                     ab = ob * (size_t)std::max(1, bounces), eb = (size_t)n * sizeof(int32_t);
        if (ranges_overlap(d_rays, rb, d_work, wb) || ranges_overlap(d_rays, rb, d_events_all, ab) || ranges_overlap(d_rays, rb, d_events_last, ob) ||
            ranges_overlap(d_work, wb, d_events_all, ab) || ranges_overlap(d_work, wb, d_events_last, ob) ||
            ranges_overlap(d_events_all, ab, d_events_last, ob) || ranges_overlap(d_excl1, eb, d_work, wb) || ranges_overlap(d_excl2, eb, d_work, wb) ||
            ranges_overlap(d_excl1, eb, d_rays, rb) || ranges_overlap(d_excl2, eb, d_rays, rb)) {
            set_error("hare_bounce_device: rays, exclusions, work array and events must not overlap");
            return HARE_E_INVALID;
        }
    }
    return bounce_device_impl(*s, H, kind, top_index, n, d_rays, d_excl1, d_excl2, bounces, flags, d_work, d_events_all, d_events_last,
                              d_counters, d_counters_per_cast, (hipStream_t)stream);
    GUARD_END
}

const char* hare_shoot_kernel_name(const hare_scene* s, int32_t kind, int32_t top_index, int64_t n, uint32_t flags)
{
    if (!s || top_index < 0 || top_index >= (int32_t)s->topos.size() || kind < HARE_KIND_VOXEL || kind > HARE_KIND_KDTREE) return "";
    // the launcher's own selection (choose_kernel), fall-backs included
    const KernChoice kc = choose_kernel(*s, s->module, kind, (size_t)top_index, n, sanitize_flags(*s, flags));
    if ((flags & HARE_SHOOT_BOUNCE_LOOP) && kc.k == Kern::VoxelPool && s->opt.bounce_fused && (flags & (HARE_SHOOT_COUNT_WORK | HARE_SHOOT_SIMPLE_KERNEL)) == 0) {
        // hare_bounce_device (<= 16 casts): the fused build of the pool kernel, where it exists and fits (bounce_device_impl's rule)
        const bool quads = s->topos[(size_t)top_index].has_quads, coarse = s->occ_shift > 0;
        const unsigned plds = (unsigned)((s->occ_words + 3) / 4) * 16u + (unsigned)kPoolWaves * (unsigned)(kPoolWaveBytes + kPoolBounceExtra);
        const DeviceModule* M = s->module;
        const bool have = !M || (!coarse ? (quads ? M->voxel_bounce_quad : M->voxel_bounce_tri) : (quads ? M->voxel_bounce_quad_g : M->voxel_bounce_tri_g)) != nullptr;
        if (have && plds <= kLdsMax)
            return !coarse ? (quads ? "hare_voxel_bounce_quad" : "hare_voxel_bounce_tri") : (quads ? "hare_voxel_bounce_quad_g" : "hare_voxel_bounce_tri_g");
    }
    return kc.name;
}

// Slim records back to X_Events (include/hare_hip.h).  Same arithmetic as the kernels: hare_math.h is compiled for the host with
// -ffp-contract=off, so o + d * t and AABB.Intersect's origin move give the bits the device gave.
int hare_expand_events(const hare_scene* s, int32_t kind, int64_t n, const hare_ray* rays, const void* slim, hare_xevent* out)
{
    if (!s || n < 0 || (n > 0 && (!rays || !slim || !out)) || kind < HARE_KIND_VOXEL || kind > HARE_KIND_KDTREE) {
        set_error("hare_expand_events: bad arguments");
        return HARE_E_INVALID;
    }
    if (kind == HARE_KIND_VOXEL && !s->vox.built) {
        set_error("hare_expand_events: voxel grid not built");
        return HARE_E_STATE;
    }
    GUARD_BEGIN
    auto body = [&](int64_t lo, int64_t hi) {
        for (int64_t i = lo; i < hi; ++i) {
            hare_xevent e;
            memset(&e, 0, sizeof e);
            e.poly_id = -1;
            const hare_ray& r = rays[i];
            if (kind == HARE_KIND_VOXEL) {
                const hare_slim_event& q = static_cast<const hare_slim_event*>(slim)[i];
                if (q.hit == 1) {
                    e.t = q.t;
                    e.x = r.x + r.dx * q.t; e.y = r.y + r.dy * q.t; e.z = r.z + r.dz * q.t;      // Polygons.cs:652
                    e.poly_id = q.poly_id;
                    e.hit = 1;
                } else if (q.hit == 2) {      // the origin was moved: redo AABB.Intersect (AABB_Main.cs:173-260), then both sums
                    V3 o = {r.x, r.y, r.z};
                    const V3 d = {r.dx, r.dy, r.dz};
                    double t_start = 0;
                    (void)aabb_clip_move(s->vox.omin, s->vox.omax, o, d, t_start);
                    e.t = q.t + t_start;                                                           // Voxel_Grid.cs:707
                    e.x = o.x + d.x * q.t; e.y = o.y + d.y * q.t; e.z = o.z + d.z * q.t;
                    e.poly_id = q.poly_id;
                    e.hit = 1;
                }
            } else {
                const hare_slim_event_uv& q = static_cast<const hare_slim_event_uv*>(slim)[i];
                if (q.hit != 0) {
                    e.t = q.t; e.u = q.u; e.v = q.v;
                    e.x = r.x + r.dx * q.t; e.y = r.y + r.dy * q.t; e.z = r.z + r.dz * q.t;
                    e.poly_id = q.poly_id;
                    e.hit = 1;
                }
            }
            out[i] = e;
        }
    };
    const int64_t nt = std::max<int64_t>(1, std::min<int64_t>({(int64_t)std::thread::hardware_concurrency(), 16, n / 65536}));
    std::vector<std::thread> th;
    for (int64_t k = 1; k < nt; ++k) {
        try {
            th.emplace_back(body, n * k / nt, n * (k + 1) / nt);
        } catch (...) {
            body(n * k / nt, n * (k + 1) / nt);
        }
    }
    body(0, n / nt);
    for (auto& t : th) t.join();
    return HARE_OK;
    GUARD_END
}

// Diagnostics / A-B switches of one scene (SceneOptions, scene.h).  Not thread-safe against shoots in flight on the scene.
namespace {
struct OptionEntry { const char* name; int SceneOptions::*field; int64_t lo, hi; };
const OptionEntry kOptionTable[] = {
        {"dev", &SceneOptions::dev, 0, 1},
        {"build_host", &SceneOptions::build_host, 0, 1},
        {"voxel_kernel", &SceneOptions::voxel_kernel, 0, 2},
        {"octree_kernel", &SceneOptions::octree_kernel, 0, 4},
        {"octree_tail", &SceneOptions::octree_tail, 0, 2},
        {"kdtree_kernel", &SceneOptions::kdtree_kernel, 0, 2},
        {"octree_tight", &SceneOptions::octree_tight, 0, 1},
        {"voxel_tight", &SceneOptions::voxel_tight, 0, 1},
        {"voxel_order", &SceneOptions::voxel_order, 0, 2},
        {"voxel_tight_max_mb", &SceneOptions::voxel_tight_max_mb, 0, 1 << 30},
        {"dev_fail_cellbox_alloc", &SceneOptions::dev_fail_cellbox_alloc, 0, 1},
        {"bounce_fused", &SceneOptions::bounce_fused, 0, 1},
        {"k2p_tail_max", &SceneOptions::k2p_tail_max, 0, 64},
        {"k2p_tail_patience", &SceneOptions::k2p_tail_patience, -1, 100000},
        {"ticket_rays", &SceneOptions::ticket_rays, 0, 4096},
        {"k1p_static_rays", &SceneOptions::k1p_static_rays, 0, 1024},
        {"k2p_static_rays", &SceneOptions::k2p_static_rays, 0, 256},
        {"batch_chunks", &SceneOptions::batch_chunks, 0, 16},
        {"coop_tail", &SceneOptions::coop_tail, 0, 1},
        {"wide_drain", &SceneOptions::wide_drain, 0, 1},
};
}  // namespace

int hare_scene_get_option(const hare_scene* s, const char* name, int64_t* value)
{
    if (!s || !name || !value) {
        set_error("hare_scene_get_option: null argument");
        return HARE_E_INVALID;
    }
    if (strcmp(name, "voxel_tight_bytes") == 0) {
        int64_t bytes = 0;
        if (s->cellbox_rad > 0)
            for (void* p : s->d_cellbox)
                if (p) bytes += (int64_t)s->vox.ct * s->vox.ct * s->vox.ct * 8 * (int64_t)sizeof(float);
        *value = bytes;
        return HARE_OK;
    }
    if (strcmp(name, "octree_scratch_bytes") == 0) {
        *value = s->d_oct_tail ? (int64_t)Scene::kOctTailRing * (int64_t)s->oct_tail_block_bytes : 0;
        return HARE_OK;
    }
    for (const OptionEntry& t : kOptionTable)
        if (strcmp(t.name, name) == 0) {
            *value = s->opt.*(t.field);
            return HARE_OK;
        }
    set_error(std::string("hare_scene_get_option: unknown option ") + name);
    return HARE_E_INVALID;
}

int hare_scene_set_option(hare_scene* s, const char* name, int64_t value)
{
    if (!s || !name) {
        set_error("hare_scene_set_option: null argument");
        return HARE_E_INVALID;
    }
    if (strcmp(name, "dev_order_ptr") == 0) {       // developer experiments: see SceneOptions::dev_order_ptr
        s->opt.dev_order_ptr = (long long)value;
        return HARE_OK;
    }
    for (const OptionEntry& t : kOptionTable)
        if (strcmp(t.name, name) == 0) {
            if (value < t.lo || value > t.hi) {
                set_error(std::string("hare_scene_set_option: value out of range for ") + name);
                return HARE_E_INVALID;
            }
            s->opt.*(t.field) = (int)value;
            // the voxels' tight boxes exist only while the option asks for them (upload_cell_boxes): switching it on for a grid that is
            // on the device without them builds them now; a changed budget (or the test hook) re-decides.  Like every build call this
            // is not thread-safe against shoots on the same scene.
            const bool box_option = t.field == &SceneOptions::voxel_tight || t.field == &SceneOptions::voxel_tight_max_mb ||
                                    t.field == &SceneOptions::dev_fail_cellbox_alloc;
            if (box_option && s->vox.built && !s->d_cells.empty() && s->module) {
                const bool rebuild = t.field == &SceneOptions::voxel_tight ? (value != 0 && s->cellbox_rad <= 0) : true;
                if (rebuild) {
                    const HipApi* H = hip_api(nullptr);
                    if (!H) return HARE_OK;
                    DeviceGuard dev_guard(H, s->device);
                    return upload_cell_boxes(*s, H);
                }
            }
            return HARE_OK;
        }
    set_error(std::string("hare_scene_set_option: unknown option ") + name);
    return HARE_E_INVALID;
}

int hare_occluded_device(hare_scene* s, int32_t kind, int32_t top_index, int64_t n, void* d_rays, const void* d_excl1,
                         const void* d_excl2, const void* d_tmax, uint32_t flags, void* d_events, void* d_occluded,
                         void* d_counters, void* stream)
{
    if (!s) {
        set_error("null scene");
        return HARE_E_INVALID;
    }
    if (n < 0 || (n > 0 && !d_occluded)) {
        set_error("hare_occluded_device: bad arguments");
        return HARE_E_INVALID;
    }
    GUARD_BEGIN
    const HipApi* H = api_or_err();
    if (!H) return HARE_E_NODEVICE;
    DeviceGuard dev_guard(H, s->device);
    if (!s->module) {
        int rc = ensure_device(*s, H);
        if (rc) return rc;
    }
    // with events: the closest-hit cast + one compare per ray; without: the flags-only kernels, whose walk ends at t_max
    return shoot_device_impl(*s, H, kind, top_index, n, d_rays, d_excl1, d_excl2, flags & ~HARE_SHOOT_SLIM_EVENTS, d_events, d_counters,
                             (hipStream_t)stream, d_tmax, d_occluded);
    GUARD_END
}

int hare_occluded_batch(hare_scene* s, int32_t kind, int32_t top_index, int64_t n, hare_ray* rays, const int32_t* excl1,
                        const int32_t* excl2, const double* tmax, uint32_t flags, int32_t* occluded, hare_xevent* events,
                        hare_counters* ctr)
{
    if (!s) {
        set_error("null scene");
        return HARE_E_INVALID;
    }
    if (n < 0 || (n > 0 && (!rays || !occluded))) {
        set_error("hare_occluded_batch: bad arguments");
        return HARE_E_INVALID;
    }
    return batch_impl(s, kind, top_index, n, rays, excl1, excl2, flags & ~(HARE_SHOOT_WRITEBACK_ORIGIN | HARE_SHOOT_SLIM_EVENTS), events, ctr, tmax,
                      occluded);
}

int hare_occluded_batch_sharded(hare_scene* const* scenes, int32_t n_scenes, int32_t kind, int32_t top_index, int64_t n, hare_ray* rays,
                                const int32_t* excl1, const int32_t* excl2, const double* tmax, uint32_t flags, int32_t* occluded,
                                hare_xevent* events, hare_counters* ctr)
{
    if (!scenes || n_scenes < 1 || n_scenes > 64) {
        set_error("hare_occluded_batch_sharded: need 1..64 scenes");
        return HARE_E_INVALID;
    }
    for (int32_t k = 0; k < n_scenes; ++k)
        if (!scenes[k]) {
            set_error("hare_occluded_batch_sharded: null scene");
            return HARE_E_INVALID;
        }
    if (n < 0 || (n > 0 && (!rays || !occluded))) {
        set_error("hare_occluded_batch_sharded: bad arguments");
        return HARE_E_INVALID;
    }
    GUARD_BEGIN
    if (ctr) memset(ctr, 0, sizeof *ctr);
    const int G = n_scenes;
    std::vector<int> rcs((size_t)G, HARE_OK);
    std::vector<std::string> errs((size_t)G);
    std::vector<hare_counters> parts((size_t)G);
    auto shard = [&](int k) {
        const int64_t lo = (int64_t)((__int128)n * k / G), hi = (int64_t)((__int128)n * (k + 1) / G);
        memset(&parts[(size_t)k], 0, sizeof(hare_counters));
        rcs[(size_t)k] = hare_occluded_batch(scenes[k], kind, top_index, hi - lo, rays + lo, excl1 ? excl1 + lo : nullptr,
                                             excl2 ? excl2 + lo : nullptr, tmax ? tmax + lo : nullptr, flags, occluded + lo,
                                             events ? events + lo : nullptr, &parts[(size_t)k]);
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
    if (ctr)
        for (int k = 0; k < G; ++k) {
            ctr->rays += parts[(size_t)k].rays;
            ctr->hits += parts[(size_t)k].hits;
            ctr->cells += parts[(size_t)k].cells;
            ctr->entries += parts[(size_t)k].entries;
            ctr->tests += parts[(size_t)k].tests;
        }
    return HARE_OK;
    GUARD_END
}

}  // extern "C"
