// host_trace.cpp -- hare_shoot_one: Spatial_Partition.Shoot for ONE ray on the calling host thread.
//
// Unchanged Pachyderm call sites cast one ray at a time from many worker threads (that is what Ray.ThreadID and the
// locked mailbox pool exist for: Hare_Geometry_Primitives.cs:422, Voxel_Grid.cs:334-342).  A GPU round trip per ray
// (H2D, launch, D2H: tens of microseconds) would make such a caller ~100x slower than the reference's ~0.4 us per
// ray, so the single-ray entry point runs on the host: the SAME trace_voxel / trace_octree / trace_kdtree the
// simple HIP kernels instantiate (hare_trace.h over hare_math.h), compiled here for the host and pointed at a host
// mirror of the scene.  No lock is taken per ray and there is no mailbox (re-testing a polygon cannot change a
// result: the accept is the strict `t < tmin`), so any number of threads may call it on one scene.
// Batches stay GPU-only: hare_shoot_batch / hare_shoot_device never come here.
//
// Product code; nothing from oracle/.
#include <string.h>
#include <algorithm>
#include <memory>
#include <new>
#include <vector>

#include "../../include/hare_hip.h"
#include "hare_trace.h"
#include "scene.h"

namespace hare {

struct HostMirror {
    std::vector<std::vector<PolyRec>> polys;   // per topology, the records the kernels read
    std::vector<std::vector<QuadRec>> quads;
    std::vector<std::vector<CellRec>> cells;   // per topology: Voxel_Inv as cell records (when a grid is built)
};

void free_host_mirror(Scene& s)
{
    HostMirror* m = s.mirror.exchange(nullptr, std::memory_order_acq_rel);
    delete m;
}

static const HostMirror* host_mirror(Scene& s)
{
    HostMirror* m = s.mirror.load(std::memory_order_acquire);
    if (m) return m;
    std::lock_guard<std::mutex> lk(s.mirror_mu);
    m = s.mirror.load(std::memory_order_acquire);
    if (m) return m;
    std::unique_ptr<HostMirror> nm(new HostMirror());
    const size_t M = s.topos.size();
    nm->polys.resize(M);
    nm->quads.resize(M);
    for (size_t t = 0; t < M; ++t) make_poly_records(s.topos[t], nm->polys[t], nm->quads[t]);
    if (s.vox.built) {
        nm->cells.resize(M);
        const size_t ncell = (size_t)s.vox.ct * s.vox.ct * s.vox.ct;
        for (size_t t = 0; t < M; ++t) {
            std::vector<CellRec>& c = nm->cells[t];
            c.resize(ncell);
            const std::vector<uint32_t>& st = s.vox.start[t];
            const std::vector<int32_t>& it = s.vox.items[t];
            for (size_t k = 0; k < ncell; ++k) {
                c[k].start = st[k];
                c[k].count = st[k + 1] - st[k];
                c[k].i0 = c[k].count > 0 ? it[st[k]] : -1;
                c[k].i1 = c[k].count > 1 ? it[st[k] + 1] : -1;
            }
        }
    }
    m = nm.release();
    s.mirror.store(m, std::memory_order_release);
    return m;
}

}  // namespace hare

using namespace hare;

extern "C" int hare_shoot_one(hare_scene* s, int32_t kind, int32_t top_index, hare_ray* ray, int32_t poly_origin1,
                              int32_t poly_origin2, hare_xevent* out)
{
    if (!s || !ray || !out) {
        set_error("hare_shoot_one: null argument");
        return HARE_E_INVALID;
    }
    if (top_index < 0 || top_index >= (int32_t)s->topos.size()) {
        set_error("hare_shoot_one: bad top_index");
        return HARE_E_INVALID;
    }
    try {
        const bool built = kind == HARE_KIND_VOXEL ? s->vox.built : kind == HARE_KIND_OCTREE ? s->oct.built
                           : kind == HARE_KIND_KDTREE ? s->kd.built : false;
        if (kind != HARE_KIND_VOXEL && kind != HARE_KIND_OCTREE && kind != HARE_KIND_KDTREE) {
            set_error("hare_shoot_one: unknown partition kind");
            return HARE_E_INVALID;
        }
        if (!built) {
            set_error("hare_shoot_one: partition not built");
            return HARE_E_STATE;
        }
        const size_t top = (size_t)top_index;
        if ((kind == HARE_KIND_OCTREE && s->oct.id_count > s->topos[top].P) || (kind == HARE_KIND_KDTREE && s->kd.id_count > s->topos[top].P)) {
            set_error("hare_shoot_one: the tree holds polygon ids of the last topology that topology " + std::to_string(top_index) + " does not have");
            return HARE_E_INVALID;
        }
        const HostMirror* hm = host_mirror(*s);
        const PolyRec* polys = hm->polys[top].data();
        const QuadRec* quads = hm->quads[top].empty() ? nullptr : hm->quads[top].data();
        V3 o = {ray->x, ray->y, ray->z};
        const V3 d = {ray->dx, ray->dy, ray->dz};
        XEventRec ev;
        Work w = {0, 0, 0};
        if (kind == HARE_KIND_VOXEL) {
            VoxelArgs g;
            memset(&g, 0, sizeof g);
            g.polys = polys;
            g.quads = quads;
            g.cells = hm->cells[top].data();
            g.items = s->vox.items[top].data();
            g.ct = s->vox.ct;
            for (int a = 0; a < 3; ++a) {
                g.omin[a] = s->vox.omin[a];
                g.omax[a] = s->vox.omax[a];
                g.vd[a] = s->vox.vd[a];
            }
            // AABB.Intersect moves the caller's Ray when it starts outside the grid (AABB_Main.cs:254-257, F11)
            if (trace_voxel<true, false>(g, o, d, poly_origin1, poly_origin2, ev, w)) {
                ray->x = o.x;
                ray->y = o.y;
                ray->z = o.z;
            }
        } else if (kind == HARE_KIND_OCTREE) {
            OctreeArgs g;
            memset(&g, 0, sizeof g);
            g.polys = polys;
            g.quads = quads;
            g.nodes = s->oct.nodes.data();
            g.items = s->oct.items.data();
            g.n_nodes = (int32_t)s->oct.nodes.size();
            g.max_depth = std::max(1, s->oct_levels);
            constexpr int kMaxLevels = 32;   // hare_octree_build caps maxDepth at 24
            if (g.max_depth > kMaxLevels) {
                set_error("hare_shoot_one: octree deeper than 32 levels");
                return HARE_E_UNSUPPORTED;
            }
            int first[kMaxLevels], cursor[kMaxLevels];
            double fa[kMaxLevels], fb[kMaxLevels];
            OctFrames fr;
            fr.first = first;
            fr.cursor = cursor;
            fr.a = fa;
            fr.b = fb;
            trace_octree<false, false>(g, fr, 0, 1, o, d, poly_origin1, poly_origin2, ev, w);
        } else {
            KdArgs g;
            memset(&g, 0, sizeof g);
            g.polys = polys;
            g.quads = quads;
            g.nodes = s->kd.nodes.data();
            g.items = s->kd.items.data();
            g.n_nodes = (int32_t)s->kd.nodes.size();
            g.max_depth = s->kd.depth_reached;
            int stack[96];                   // at most depth + 2 entries; hare_kdtree_build caps maxDepth at 60
            if (g.max_depth + 2 > 96) {
                set_error("hare_shoot_one: kd-tree deeper than the host stack");
                return HARE_E_UNSUPPORTED;
            }
            trace_kdtree<false>(g, stack, 0, 1, o, d, poly_origin1, poly_origin2, ev, w);
        }
        memcpy(out, &ev, sizeof ev);
        return HARE_OK;
    } catch (const std::bad_alloc&) {
        set_error("hare_shoot_one: out of host memory");
        return HARE_E_NOMEM;
    } catch (...) {
        set_error("hare_shoot_one: unexpected C++ exception");
        return HARE_E_INVALID;
    }
}
