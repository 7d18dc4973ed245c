// hare.hpp -- C++ host-side mirror of the reference interface for the ray-cast path, over the C-ABI
// (include/hare_hip.h).  The reference is a compiled C# library whose toolchain is absent from the
// build image, so the compiled-language host side is written in C++ with the reference's names,
// argument meaning and error behaviour:
//   Hare::Geometry::Ray / X_Event          Hare_Geometry_Primitives.cs:393-481
//   Hare::Geometry::Topology               the members a partition reads (Hare_Geometry_Topology.cs:418-424,482,539,50/58)
//   Hare::Geometry::Spatial_Partition      Spatial_Partition.cs:27-35
//   Voxel_Grid / Octree / KDTree           Voxel_Grid.cs:48,128  "Octree - alt.cs":45  KDTree.cs:51
// Header-only; link with -lhare_hip.  Errors become exceptions, like in .NET.
#pragma once
#include <array>
#include <cmath>
#include <stdexcept>
#include <string>
#include <vector>

#include "hare_hip.h"

namespace Hare {
namespace Geometry {

struct Ray {                       // Hare_Geometry_Primitives.cs:393-429
    double x, y, z, dx, dy, dz;
    int ThreadID = 0, Ray_ID = 0;
    Ray(double x_, double y_, double z_, double dx_, double dy_, double dz_, int ThreadID_IN = 0, int ID = 0)
        : x(x_), y(y_), z(z_), dx(dx_), dy(dy_), dz(dz_), ThreadID(ThreadID_IN), Ray_ID(ID) {}
    void Reverse() { dx *= -1; dy *= -1; dz *= -1; }
};

struct X_Event {                   // Hare_Geometry_Primitives.cs:435-481
    double u = 0, v = 0, t = 0;
    bool Hit = false;
    bool has_point = false;        // X_Point == null on a miss
    std::array<double, 3> X_Point{{0, 0, 0}};
    int Poly_id = -1;
    X_Event() = default;           // :454-462
    explicit X_Event(const hare_xevent& e)
    {
        if (e.hit) { u = e.u; v = e.v; t = e.t; Hit = true; has_point = true; X_Point = {{e.x, e.y, e.z}}; Poly_id = e.poly_id; }
    }
};

inline void check(int rc)
{
    if (rc == HARE_OK) return;
    const std::string msg = std::string("hare_hip error ") + std::to_string(rc) + ": " + hare_last_error();
    if (rc == HARE_E_INVALID) throw std::invalid_argument(msg);
    throw std::runtime_error(msg);
}

class Topology {
public:
    std::vector<double> verts;     // P x 4 x 3
    std::vector<int32_t> nverts;   // P
    std::vector<double> normals;   // P x 3
    double Min[3], Max[3];

    // polygons: P x 4 x 3 doubles (corner 3 ignored for triangles) + corner counts; normals and the
    // Min/Max box are computed by the library's restatements of the Polygon ctor / Finish_Topology
    Topology(const double* v, const int32_t* nv, int32_t P) : verts(v, v + (size_t)P * 12), nverts(nv, nv + P), normals((size_t)P * 3)
    {
        check(hare_polygon_normals(verts.data(), nverts.data(), P, normals.data()));
        check(hare_topology_bounds(verts.data(), nverts.data(), P, Min, Max));
    }
    // Topology(Point[][]) + Finish_Topology(): raw polygon soup through the reference's ingest (Round(15) and the
    // 1 mm Hash2 corner merge) first.  Vertices_List and the per-corner vertex index are returned on request.
    static Topology from_polygons(const double* soup, const int32_t* nv, int32_t P, std::vector<double>* vertices = nullptr,
                                  std::vector<int32_t>* corner_vertex = nullptr)
    {
        std::vector<double> merged((size_t)P * 12), vlist;
        std::vector<int32_t> cv((size_t)P * 4);
        size_t corners = 0;
        for (int32_t p = 0; p < P; ++p) corners += (size_t)(nv[p] > 0 ? nv[p] : 0);
        vlist.resize(corners * 3 + 3);
        int32_t n_vertices = 0;
        check(hare_topology_ingest(soup, nv, P, merged.data(), cv.data(), vlist.data(), &n_vertices));
        vlist.resize((size_t)n_vertices * 3);
        if (vertices) *vertices = std::move(vlist);
        if (corner_vertex) *corner_vertex = std::move(cv);
        return Topology(merged.data(), nv, P);
    }
    int Polygon_Count() const { return (int)nverts.size(); }
    std::array<double, 3> Normal(int Poly_ID) const { return {{normals[3 * Poly_ID], normals[3 * Poly_ID + 1], normals[3 * Poly_ID + 2]}}; }
    std::array<double, 3> operator()(int Poly_ID, int Corner_ID) const
    {
        const double* p = &verts[(size_t)Poly_ID * 12 + 3 * Corner_ID];
        return {{p[0], p[1], p[2]}};
    }
    hare_topology_desc desc() const
    {
        hare_topology_desc d{};
        d.P = Polygon_Count();
        d.verts = verts.data();
        d.nverts = nverts.data();
        d.normals = normals.data();
        for (int a = 0; a < 3; ++a) { d.min[a] = Min[a]; d.max[a] = Max[a]; }
        return d;
    }
};

class Spatial_Partition {          // Spatial_Partition.cs:27-35
public:
    std::vector<const Topology*> Model;
    double Char_Step = 0;

    virtual ~Spatial_Partition() { hare_scene_destroy(scene_); }
    Spatial_Partition(const Spatial_Partition&) = delete;
    Spatial_Partition& operator=(const Spatial_Partition&) = delete;

    // bool Shoot(Ray R, int top_index, out X_Event Ret_event[, int poly_origin1, int poly_origin2 = -1])
    bool Shoot(Ray& R, int top_index, X_Event& Ret_event, int poly_origin1 = -1, int poly_origin2 = -1)
    {
        // one ray = the reference call site: traced on the calling host thread (hare_shoot_one; lock-free, no GPU round trip)
        hare_ray r{R.x, R.y, R.z, R.dx, R.dy, R.dz};
        hare_xevent e;
        check(hare_shoot_one(scene_, kind_, top_index, &r, poly_origin1, poly_origin2, &e));
        R.x = r.x; R.y = r.y; R.z = r.z;     // the reference moves R when it starts outside the grid
        Ret_event = X_Event(e);
        return Ret_event.Hit;
    }

    // the batch entry: n rays at once through the HIP kernels; returns the number of hits
    uint64_t Shoot(std::vector<hare_ray>& rays, int top_index, std::vector<hare_xevent>& results,
                   const int32_t* poly_origin1 = nullptr, const int32_t* poly_origin2 = nullptr, bool move_origins = false)
    {
        results.resize(rays.size());
        hare_counters c{};
        check(hare_shoot_batch(scene_, kind_, top_index, (int64_t)rays.size(), rays.data(), poly_origin1, poly_origin2,
                               move_origins ? HARE_SHOOT_WRITEBACK_ORIGIN : 0u, results.data(), &c));
        return c.hits;
    }
    // occlusion predicate (harness-defined): occluded[i] = closest hit of rays[i] exists and t < t_max[i] (null: any hit)
    std::vector<int32_t> Occluded(std::vector<hare_ray>& rays, int top_index, const double* t_max = nullptr,
                                  const int32_t* poly_origin1 = nullptr, const int32_t* poly_origin2 = nullptr)
    {
        std::vector<int32_t> occ(rays.size());
        check(hare_occluded_batch(scene_, kind_, top_index, (int64_t)rays.size(), rays.data(), poly_origin1, poly_origin2, t_max, 0u,
                                  occ.data(), nullptr, nullptr));
        return occ;
    }
    // the bounce loop in one call (harness-defined; the reference leaves reflection to its caller, who re-shoots with
    // poly_origin1 = the polygon just hit, Spatial_Partition.cs:33): `bounces` casts, rays resident on the GPU throughout.
    // events_all: bounces x rays.size() records, cast-major.  Returns the hits over all casts.
    uint64_t Bounce(const std::vector<hare_ray>& rays, int top_index, int bounces, std::vector<hare_xevent>& events_all,
                    const int32_t* poly_origin1 = nullptr, const int32_t* poly_origin2 = nullptr, std::vector<hare_counters>* per_cast = nullptr)
    {
        events_all.resize(rays.size() * (size_t)(bounces > 0 ? bounces : 0));
        if (per_cast) per_cast->resize((size_t)(bounces > 0 ? bounces : 0));
        hare_counters c{};
        check(hare_bounce_batch(scene_, kind_, top_index, (int64_t)rays.size(), rays.data(), poly_origin1, poly_origin2, bounces, 0u,
                                events_all.data(), nullptr, &c, per_cast ? per_cast->data() : nullptr));
        return c.hits;
    }
    void SetOption(const char* name, int64_t value) { check(hare_scene_set_option(scene_, name, value)); }
    int64_t GetOption(const char* name) const { int64_t v = 0; check(hare_scene_get_option(scene_, name, &v)); return v; }
    hare_scene* native() const { return scene_; }

protected:
    Spatial_Partition(const std::vector<const Topology*>& Model_in, int kind, int device) : Model(Model_in), kind_(kind)
    {
        std::vector<hare_topology_desc> d;
        for (const Topology* t : Model) d.push_back(t->desc());
        check(hare_scene_create(d.data(), (int32_t)d.size(), device, &scene_));
    }
    hare_scene* scene_ = nullptr;
    int kind_;
};

class Voxel_Grid : public Spatial_Partition {
public:
    Voxel_Grid(const std::vector<const Topology*>& Model_in, int Domain, int device = 0) : Spatial_Partition(Model_in, HARE_KIND_VOXEL, device)
    {
        check(hare_voxel_build(scene_, Domain));
        refresh();
    }
    Voxel_Grid(const std::vector<const Topology*>& Model_in, int MaxDomain, int Avg_polys, int device) : Spatial_Partition(Model_in, HARE_KIND_VOXEL, device)
    {
        check(hare_voxel_build_adaptive(scene_, MaxDomain, Avg_polys));
        refresh();
    }
    double Xdim() const { return info_.box_dims[0]; }
    double Ydim() const { return info_.box_dims[1]; }
    double Zdim() const { return info_.box_dims[2]; }
    std::array<double, 3> MinPt() const { return {{info_.obox_min[0], info_.obox_min[1], info_.obox_min[2]}}; }
    int VoxelCode(int X, int Y, int Z) const { return info_.ct * info_.ct * Z + info_.ct * X + Y; }          // Voxel_Grid.cs:264-267
    void VoxelDecode(int Code, int& X, int& Y, int& Z) const                                                    // :256-262
    {
        const int XYTot = info_.ct * info_.ct;
        Z = Code / XYTot;
        Code -= Z * XYTot;
        Y = Code / info_.ct;
        X = Code - Y * info_.ct;
    }
    void PointInVoxel(const double Pt[3], int& X, int& Y, int& Z) const                                         // :322-327
    {
        X = (int)std::floor((Pt[0] - info_.obox_min[0]) / info_.voxel_dims[0]);
        Y = (int)std::floor((Pt[1] - info_.obox_min[1]) / info_.voxel_dims[1]);
        Z = (int)std::floor((Pt[2] - info_.obox_min[2]) / info_.voxel_dims[2]);
    }
    int PointInVoxel(const double Pt[3]) const                                                                   // :329-332
    {
        int X, Y, Z;
        PointInVoxel(Pt, X, Y, Z);
        return VoxelCode(X, Y, Z);
    }
    const hare_voxel_info& info() const { return info_; }

private:
    void refresh()
    {
        check(hare_voxel_get_info(scene_, &info_));
        Char_Step = info_.char_step;
    }
    hare_voxel_info info_{};
};

class Octree : public Spatial_Partition {
public:
    Octree(const std::vector<const Topology*>& Model_In, int maxDepth, int maxPolygonsPerNode, int device = 0)
        : Spatial_Partition(Model_In, HARE_KIND_OCTREE, device)
    {
        check(hare_octree_build(scene_, maxDepth, maxPolygonsPerNode));
    }
};

class KDTree : public Spatial_Partition {
public:
    KDTree(const std::vector<const Topology*>& Model_In, int maxDepth, int maxPolygonsPerNode, int device = 0)
        : Spatial_Partition(Model_In, HARE_KIND_KDTREE, device)
    {
        check(hare_kdtree_build(scene_, maxDepth, maxPolygonsPerNode));
    }
};

}  // namespace Geometry
}  // namespace Hare
