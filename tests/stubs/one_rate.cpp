// Drives hare_shoot_one from compiled code (tests/test_shoot_one.py): args = dir P n domain min[3] max[3].
// Reads verts/nverts/normals/rays from <dir>, writes events.bin, prints mrays_1t=... mrays_4t=...
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <thread>
#include <vector>

#include "hare_hip.h"

template <class T>
static std::vector<T> slurp(const std::string& p, size_t count)
{
    std::vector<T> v(count);
    FILE* f = fopen(p.c_str(), "rb");
    if (!f || fread(v.data(), sizeof(T), count, f) != count) { fprintf(stderr, "cannot read %s\n", p.c_str()); exit(2); }
    fclose(f);
    return v;
}

int main(int argc, char** argv)
{
    if (argc != 11) return 2;
    const std::string dir = argv[1];
    const int P = atoi(argv[2]);
    const long n = atol(argv[3]);
    const int domain = atoi(argv[4]);
    auto verts = slurp<double>(dir + "/verts.bin", (size_t)P * 12);
    auto nverts = slurp<int32_t>(dir + "/nverts.bin", (size_t)P);
    auto normals = slurp<double>(dir + "/normals.bin", (size_t)P * 3);
    auto rays0 = slurp<hare_ray>(dir + "/rays.bin", (size_t)n);
    hare_topology_desc d{};
    d.P = P;
    d.verts = verts.data();
    d.nverts = nverts.data();
    d.normals = normals.data();
    for (int a = 0; a < 3; ++a) { d.min[a] = atof(argv[5 + a]); d.max[a] = atof(argv[8 + a]); }
    hare_scene* s = nullptr;
    if (hare_scene_create(&d, 1, 0, &s) || hare_voxel_build(s, domain)) { fprintf(stderr, "%s\n", hare_last_error()); return 3; }
    std::vector<hare_xevent> out((size_t)n);
    auto pass = [&](int threads) {
        std::vector<hare_ray> rays = rays0;
        auto t0 = std::chrono::steady_clock::now();
        std::vector<std::thread> th;
        for (int k = 0; k < threads; ++k)
            th.emplace_back([&, k] {
                for (long i = n * k / threads; i < n * (k + 1) / threads; ++i)
                    if (hare_shoot_one(s, HARE_KIND_VOXEL, 0, &rays[i], -1, -1, &out[i])) abort();
            });
        for (auto& t : th) t.join();
        return n / std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() / 1e6;
    };
    pass(1);   // builds the host mirror, warms the caches
    double r1 = 0, r4 = 0;
    for (int k = 0; k < 5; ++k) r1 = std::max(r1, pass(1));   // best of 5: the box is shared
    for (int k = 0; k < 3; ++k) r4 = std::max(r4, pass(4));
    FILE* f = fopen((dir + "/events.bin").c_str(), "wb");
    fwrite(out.data(), sizeof(hare_xevent), (size_t)n, f);
    fclose(f);
    printf("mrays_1t=%.3f mrays_4t=%.3f\n", r1, r4);
    hare_scene_destroy(s);
    return 0;
}
