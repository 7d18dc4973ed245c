// example.cpp -- the C++ mirror in use: a unit cube, one ray, printed like a Hare caller would.
// Build:  g++ -std=c++17 -I include -I bindings/cpp bindings/cpp/example.cpp -L hare_amd -lhare_hip -Wl,-rpath,$PWD/hare_amd -o /tmp/hare_example
// Without a GPU the constructor (host build) and the single-ray Shoot (host trace) work; the batch Shoot, which is
// GPU-only, throws "no HIP device visible".
#include <cstdio>

#include "hare.hpp"

using namespace Hare::Geometry;

int main()
{
    // 12 triangles of the cube [0,2]^3
    const double c[8][3] = {{0, 0, 0}, {2, 0, 0}, {2, 2, 0}, {0, 2, 0}, {0, 0, 2}, {2, 0, 2}, {2, 2, 2}, {0, 2, 2}};
    const int f[12][3] = {{0, 1, 2}, {0, 2, 3}, {4, 6, 5}, {4, 7, 6}, {0, 5, 1}, {0, 4, 5}, {3, 2, 6}, {3, 6, 7}, {0, 3, 7}, {0, 7, 4}, {1, 5, 6}, {1, 6, 2}};
    std::vector<double> verts(12 * 12, 0.0);
    std::vector<int32_t> nverts(12, 3);
    for (int p = 0; p < 12; ++p)
        for (int k = 0; k < 3; ++k)
            for (int a = 0; a < 3; ++a) verts[p * 12 + 3 * k + a] = c[f[p][k]][a];
    Topology topo(verts.data(), nverts.data(), 12);
    try {
        Voxel_Grid grid({&topo}, 4);
        std::printf("Char_Step = %.17g\n", grid.Char_Step);
        Ray R(0.5, 0.75, 1.0, 1.0, 0.0, 0.0, 0, 1);
        X_Event ev;
        if (grid.Shoot(R, 0, ev)) std::printf("hit poly %d at t = %.17g (%.3f, %.3f, %.3f)\n", ev.Poly_id, ev.t, ev.X_Point[0], ev.X_Point[1], ev.X_Point[2]);
        else std::printf("miss\n");
        std::vector<hare_ray> rays = {{0.5, 0.75, 1.0, 1.0, 0.0, 0.0}, {0.5, 0.75, 1.0, 0.0, 0.0, -1.0}};
        std::vector<hare_xevent> evs;
        const unsigned long long hits = grid.Shoot(rays, 0, evs);          // the batch entry: HIP kernels, no CPU fallback
        std::printf("batch: %llu hits, t = %.17g and %.17g\n", hits, evs[0].t, evs[1].t);
        // the bounce loop in one call: ray 0 runs along +x inside the cube, 1.5 to the wall, then 2 from wall to wall
        std::vector<hare_xevent> all;
        std::vector<hare_counters> per_cast;
        const unsigned long long bh = grid.Bounce(rays, 0, 3, all, nullptr, nullptr, &per_cast);
        std::printf("bounce: %llu hits, ray 0 t = %.17g, %.17g, %.17g; cast 2: %llu rays\n", bh, all[0].t, all[2].t, all[4].t,
                    (unsigned long long)per_cast[2].rays);
    } catch (const std::exception& e) {
        std::printf("exception: %s\n", e.what());
        return 2;
    }
    return 0;
}
