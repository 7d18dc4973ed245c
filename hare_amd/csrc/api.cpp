// api.cpp -- the C-ABI of include/hare_hip.h: scene lifetime, build calls, introspection, the host-buffer batch calls, options.
// (What a scene keeps on its device: device_scene.cpp; which kernel serves a batch and how it is launched: launch.cpp.)
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
#include "launch.h"

// the embedded gfx950 code object (embed.S)
extern "C" const unsigned char hare_kernels_co[];
extern "C" const unsigned char hare_kernels_co_end[];

static_assert(sizeof(hare_ray) == sizeof(hare::RayRec), "hare_ray layout");
static_assert(sizeof(hare_xevent) == sizeof(hare::XEventRec), "hare_xevent layout");
static_assert(sizeof(hare_xevent) == 56 && sizeof(hare_ray) == 48, "wire sizes");
static_assert(sizeof(hare_counters) == hare::CTR_WORDS * 8, "hare_counters layout");

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
        for (auto* v : {&s->d_polys, &s->d_cull, &s->d_quads, &s->d_cells, &s->d_items, &s->d_occ, &s->d_cellbox, &s->d_bocc})
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
        }
        dev_free(H, s->d_order);
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
    return sync_partition_to_device(*s, HARE_KIND_VOXEL);
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
    return sync_partition_to_device(*s, HARE_KIND_VOXEL);
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
    return sync_partition_to_device(*s, HARE_KIND_OCTREE);
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
    return sync_partition_to_device(*s, HARE_KIND_KDTREE);
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
    int32_t marks_valid = 0;         // the caller's array: whatever it holds, it is output only here
    unsigned char* no_bytes = nullptr;
    void* args[] = {&polys, &d_rays, &d_events, &d_excl_out, &n, &marks_valid, &no_bytes};
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
        {"voxel_order_max_rays", &SceneOptions::voxel_order_max_rays, 0, 0x7FFF0000},
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
        {"bounce_pack", &SceneOptions::bounce_pack, 0, 1},
        {"voxel_walk", &SceneOptions::voxel_walk, 0, 1},
        {"voxel_skip", &SceneOptions::voxel_skip, 0, 1},
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
    if (strcmp(name, "voxel_order_bytes") == 0) {
        *value = (int64_t)((size_t)Scene::kOrderRing * s->order_cap * sizeof(uint32_t));
        return HARE_OK;
    }
    if (strcmp(name, "hip_malloc_calls") == 0 || strcmp(name, "hip_free_calls") == 0 || strcmp(name, "hip_sync_calls") == 0) {
        *value = (int64_t)hip_call_count(name[4] == 'm' ? 0 : (name[4] == 'f' ? 1 : 2));
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
            // the order ring follows "voxel_order" / "voxel_order_max_rays" the same way (reserve_order_ring: reserved, resized or released now,
            // never inside a shoot)
            const bool ring_option = t.field == &SceneOptions::voxel_order || t.field == &SceneOptions::voxel_order_max_rays;
            if (ring_option && s->vox.built && !s->d_cells.empty() && s->module) {
                const HipApi* H = hip_api(nullptr);
                if (!H) return HARE_OK;
                DeviceGuard dev_guard(H, s->device);
                reserve_order_ring(*s, H);
                return HARE_OK;
            }
            if (t.field == &SceneOptions::voxel_skip && s->vox.built && !s->d_cells.empty() && s->module) {     // the block-level occupancy exists only while the option is on
                const HipApi* H = hip_api(nullptr);
                if (!H) return HARE_OK;
                DeviceGuard dev_guard(H, s->device);
                upload_block_occ(*s, H);
                return HARE_OK;
            }
            const bool oct_option = t.field == &SceneOptions::octree_kernel || t.field == &SceneOptions::octree_tail ||
                                    t.field == &SceneOptions::k2p_tail_max || t.field == &SceneOptions::k2p_tail_patience;
            if (oct_option && s->oct.built && s->d_oct_nodes && s->module) {        // the octree scratch ring is sized by these (reserve_oct_scratch)
                const HipApi* H = hip_api(nullptr);
                if (!H) return HARE_OK;
                DeviceGuard dev_guard(H, s->device);
                reserve_oct_scratch(*s, H);
                return HARE_OK;
            }
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
