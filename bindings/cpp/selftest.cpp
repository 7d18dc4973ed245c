// selftest.cpp -- sanitizer driver for the HOST side of libhare_hip (scene, host builders, ABI error paths).
// Built by `make -C hare_amd/csrc asan-test` with -fsanitize=address,undefined, sources linked in directly.
// Needs no GPU: every shoot must come back HARE_E_NODEVICE (or run, when a device is present).
#include <cmath>
#include <cstdio>
#include <vector>

#include "hare.hpp"

using namespace Hare::Geometry;

int main()
{
    const int n = 7;
    const double L[3] = {5.0, 4.0, 3.0};
    std::vector<double> verts;
    std::vector<int32_t> nverts;
    auto push = [&](const double (*q)[3], int nv) {
        for (int k = 0; k < 4; ++k)
            for (int a = 0; a < 3; ++a) verts.push_back(k < nv ? q[k][a] : 0.0);
        nverts.push_back(nv);
    };
    for (int f = 0; f < 6; ++f) {
        const int a = f / 2, b = (a + 1) % 3, c = (a + 2) % 3;
        const double w = (f & 1) ? L[a] : 0.0;
        for (int i = 0; i < n; ++i)
            for (int j = 0; j < n; ++j) {
                double q[4][3];
                for (int k = 0; k < 4; ++k) {
                    const int di = (k == 1 || k == 2), dj = (k >= 2);
                    q[k][a] = w;
                    q[k][b] = L[b] * (i + di) / n;
                    q[k][c] = L[c] * (j + dj) / n;
                }
                if ((i + j) % 4 == 0) push(q, 4);
                else {
                    const double t0[3][3] = {{q[0][0], q[0][1], q[0][2]}, {q[1][0], q[1][1], q[1][2]}, {q[2][0], q[2][1], q[2][2]}};
                    const double t1[3][3] = {{q[0][0], q[0][1], q[0][2]}, {q[2][0], q[2][1], q[2][2]}, {q[3][0], q[3][1], q[3][2]}};
                    push(t0, 3);
                    push(t1, 3);
                }
            }
    }
    Topology topo(verts.data(), nverts.data(), (int32_t)nverts.size());
    int failures = 0;
    try {
        {   // ingest: a copy of polygon 0 whose corners are 0.2 mm off merges back onto polygon 0's vertices
            std::vector<double> soup(verts.begin(), verts.begin() + 12);
            soup.insert(soup.end(), verts.begin(), verts.begin() + 12);
            for (int c = 0; c < nverts[0]; ++c) soup[12 + 3 * c] += 0.0002;
            const int32_t nv2[2] = {nverts[0], nverts[0]};
            std::vector<double> vlist;
            std::vector<int32_t> cv;
            Topology merged = Topology::from_polygons(soup.data(), nv2, 2, &vlist, &cv);
            bool same = true;
            for (int k = 0; k < 3 * nverts[0]; ++k) same = same && merged.verts[12 + k] == merged.verts[k];
            if (!same || (int)vlist.size() != 3 * nverts[0] || cv[4] != cv[0]) {
                std::printf("selftest: ingest did not merge near-duplicate corners\n");
                ++failures;
            }
            const int32_t bad[1] = {5};
            if (hare_topology_ingest(soup.data(), bad, 1, soup.data(), nullptr, nullptr, nullptr) != HARE_E_UNSUPPORTED) ++failures;
        }
        Voxel_Grid fixed({&topo}, 9);
        Voxel_Grid adaptive({&topo}, 5, 6, 0);
        Octree oct({&topo}, 5, 6);
        KDTree kd({&topo}, 7, 5);
        std::vector<uint32_t> start((size_t)9 * 9 * 9 + 1);
        check(hare_voxel_get_lists(fixed.native(), 0, start.data(), nullptr));
        std::vector<int32_t> items(start.back() + 1);
        check(hare_voxel_get_lists(fixed.native(), 0, start.data(), items.data()));
        hare_tree_info ti;
        check(hare_octree_get_info(oct.native(), &ti));
        std::vector<double> boxes((size_t)ti.n_nodes * 6);
        std::vector<int32_t> fc(ti.n_nodes), is(ti.n_nodes), ic(ti.n_nodes), it(ti.total_items + 1);
        check(hare_octree_get_nodes(oct.native(), boxes.data(), fc.data(), is.data(), ic.data(), it.data()));
        std::printf("selftest: %d polys, grid ct=%d/%d, %u list entries, octree %d nodes\n", topo.Polygon_Count(), fixed.info().ct,
                    adaptive.info().ct, start.back(), ti.n_nodes);
        {   // options: set / read back / range check / unknown name; the memory figures are 0 on a scene that never saw a device
            kd.SetOption("kdtree_kernel", 2);
            fixed.SetOption("voxel_order", 0);
            if (kd.GetOption("kdtree_kernel") != 2 || fixed.GetOption("voxel_order") != 0 || fixed.GetOption("voxel_tight") != 1) ++failures;
            int64_t v = -1;
            if (hare_scene_get_option(fixed.native(), "no_such_option", &v) != HARE_E_INVALID) ++failures;
            if (hare_scene_get_option(fixed.native(), "voxel_tight", nullptr) != HARE_E_INVALID) ++failures;
            if (hare_scene_set_option(fixed.native(), "voxel_order", 3) != HARE_E_INVALID) ++failures;
            if (fixed.GetOption("voxel_tight_bytes") < 0 || oct.GetOption("octree_scratch_bytes") < 0) ++failures;
            fixed.SetOption("voxel_order", 1);
            kd.SetOption("kdtree_kernel", 0);
        }
        std::vector<hare_ray> rays(1000);
        for (size_t i = 0; i < rays.size(); ++i) rays[i] = {2.5, 2.0, 1.5, std::cos(0.01 * i), std::sin(0.01 * i), 0.3};
        std::vector<hare_xevent> ev;
        for (Spatial_Partition* sp : {(Spatial_Partition*)&fixed, (Spatial_Partition*)&oct, (Spatial_Partition*)&kd}) {
            try {
                const uint64_t hits = sp->Shoot(rays, 0, ev);
                if (hits != rays.size()) { std::printf("unexpected hit count %llu\n", (unsigned long long)hits); ++failures; }
            } catch (const std::runtime_error& e) {
                if (std::string(e.what()).find("-4") == std::string::npos) { std::printf("unexpected: %s\n", e.what()); ++failures; }
            }
        }
        // error paths
        hare_scene* bad = nullptr;
        if (hare_scene_create(nullptr, 1, 0, &bad) != HARE_E_INVALID) ++failures;
        if (hare_voxel_build(fixed.native(), 0) != HARE_E_INVALID) ++failures;
        if (hare_shoot_batch(fixed.native(), 7, 0, 1, rays.data(), nullptr, nullptr, 0, ev.data(), nullptr) == HARE_OK) ++failures;
    } catch (const std::exception& e) {
        std::printf("exception: %s\n", e.what());
        return 2;
    }
    std::printf("selftest: %d failures\n", failures);
    return failures != 0;
}
