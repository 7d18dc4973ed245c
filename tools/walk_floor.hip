// Developer microbenchmark: the floor of the walk phase.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/walk_floor.hip -o /tmp/walk_floor && /tmp/walk_floor
// Every lane walks one ray through a 64^3 grid with the production step arithmetic (FP64 tMax compares, selects,
// occupancy bit from an LDS bitmap, a dependent 16-byte cell-record load on occupied voxels) until it leaves the
// grid -- no polygon work, all lanes busy.  Reports steps/s for several launch shapes, i.e. what the walk costs
// when lane occupancy and wave occupancy are not the limit.  ~10 % of voxels occupied, like the hall at D = 64.
#include <hip/hip_runtime.h>
#pragma clang diagnostic ignored "-Wunused-value"
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

struct Cell { unsigned start, count; int i0, i1; };

template <int EXTRA_VGPR>
__global__ __launch_bounds__(256) void walk(const double* rays, const unsigned* occ, const Cell* cells, unsigned long long* out,
                                            int n, int ct, double vd)
{
    extern __shared__ unsigned locc[];
    for (int k = threadIdx.x; k < ct * ct * ct / 32; k += blockDim.x) locc[k] = occ[k];
    __syncthreads();
    unsigned long long steps = 0, sum = 0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const double ox = rays[6 * i], oy = rays[6 * i + 1], oz = rays[6 * i + 2];
        const double dx = rays[6 * i + 3], dy = rays[6 * i + 4], dz = rays[6 * i + 5];
        int X = (int)floor(ox / vd), Y = (int)floor(oy / vd), Z = (int)floor(oz / vd);
        int cell = (X * ct + Y) * ct + Z;
        const int sx = dx < 0 ? -1 : 1, sy = dy < 0 ? -1 : 1, sz = dz < 0 ? -1 : 1;
        const int cx = sx * ct * ct, cy = sy * ct, cz = sz;
        double tMaxX = ((dx < 0 ? X : X + 1) * vd - ox) / dx, tMaxY = ((dy < 0 ? Y : Y + 1) * vd - oy) / dy,
               tMaxZ = ((dz < 0 ? Z : Z + 1) * vd - oz) / dz;
        const double tDX = vd / dx * sx, tDY = vd / dy * sy, tDZ = vd / dz * sz;
        for (;;) {
            const bool cxy = tMaxX < tMaxY, cxz = tMaxX < tMaxZ, cyz = tMaxY < tMaxZ;
            const bool bx = cxy & cxz, by = (!cxy) & cyz, bz = !(bx | by);
            const double nX = tMaxX + tDX, nY = tMaxY + tDY, nZ = tMaxZ + tDZ;
            X += bx ? sx : 0; Y += by ? sy : 0; Z += bz ? sz : 0;
            tMaxX = bx ? nX : tMaxX; tMaxY = by ? nY : tMaxY; tMaxZ = bz ? nZ : tMaxZ;
            cell += bx ? cx : (by ? cy : cz);
            if (((unsigned)X >= (unsigned)ct) | ((unsigned)Y >= (unsigned)ct) | ((unsigned)Z >= (unsigned)ct)) break;
            ++steps;
            if ((locc[cell >> 5] >> (cell & 31)) & 1u) {
                const Cell c = cells[cell];
                sum += c.start + c.count + (unsigned)c.i0;
            }
        }
    }
    if constexpr (EXTRA_VGPR > 0) {   // burn registers to lower the wave occupancy like the production kernel's 124 VGPRs do
        double pad[EXTRA_VGPR > 0 ? EXTRA_VGPR : 1];
#pragma unroll
        for (int k = 0; k < EXTRA_VGPR; ++k) pad[k] = rays[k] * (double)(steps + k);
#pragma unroll
        for (int k = 0; k < EXTRA_VGPR; ++k) asm volatile("" : "+v"(pad[k]));
        double s2 = 0;
#pragma unroll
        for (int k = 0; k < EXTRA_VGPR; ++k) s2 += pad[k];
        sum += (unsigned long long)s2;
    }
    atomicAdd(&out[0], steps);
    atomicAdd(&out[1], sum);
}

// Same walk, but a lane that steps into an occupied voxel only notes it ("arrived") and stops stepping; the cell
// records of all lanes that arrived during a round of STEPS iterations are loaded together after the round --
// one wait per round instead of one per iteration (what the production round structure could do).
template <int STEPS>
__global__ __launch_bounds__(256) void walk_deferred(const double* rays, const unsigned* occ, const Cell* cells, unsigned long long* out,
                                                     int n, int ct, double vd)
{
    extern __shared__ unsigned locc[];
    for (int k = threadIdx.x; k < ct * ct * ct / 32; k += blockDim.x) locc[k] = occ[k];
    __syncthreads();
    unsigned long long steps = 0, sum = 0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const double ox = rays[6 * i], oy = rays[6 * i + 1], oz = rays[6 * i + 2];
        const double dx = rays[6 * i + 3], dy = rays[6 * i + 4], dz = rays[6 * i + 5];
        int X = (int)floor(ox / vd), Y = (int)floor(oy / vd), Z = (int)floor(oz / vd);
        int cell = (X * ct + Y) * ct + Z;
        const int sx = dx < 0 ? -1 : 1, sy = dy < 0 ? -1 : 1, sz = dz < 0 ? -1 : 1;
        const int cx = sx * ct * ct, cy = sy * ct, cz = sz;
        double tMaxX = ((dx < 0 ? X : X + 1) * vd - ox) / dx, tMaxY = ((dy < 0 ? Y : Y + 1) * vd - oy) / dy,
               tMaxZ = ((dz < 0 ? Z : Z + 1) * vd - oz) / dz;
        const double tDX = vd / dx * sx, tDY = vd / dy * sy, tDZ = vd / dz * sz;
        bool alive = true;
        while (__ballot(alive)) {
            bool arrived = false;
#pragma unroll 1
            for (int k = 0; k < STEPS; ++k) {
                if (alive && !arrived) {
                    const bool cxy = tMaxX < tMaxY, cxz = tMaxX < tMaxZ, cyz = tMaxY < tMaxZ;
                    const bool bx = cxy & cxz, by = (!cxy) & cyz, bz = !(bx | by);
                    const double nX = tMaxX + tDX, nY = tMaxY + tDY, nZ = tMaxZ + tDZ;
                    X += bx ? sx : 0; Y += by ? sy : 0; Z += bz ? sz : 0;
                    tMaxX = bx ? nX : tMaxX; tMaxY = by ? nY : tMaxY; tMaxZ = bz ? nZ : tMaxZ;
                    cell += bx ? cx : (by ? cy : cz);
                    if (((unsigned)X >= (unsigned)ct) | ((unsigned)Y >= (unsigned)ct) | ((unsigned)Z >= (unsigned)ct)) alive = false;
                    else {
                        ++steps;
                        arrived = (locc[cell >> 5] >> (cell & 31)) & 1u;
                    }
                }
            }
            if (arrived) {
                const Cell c = cells[cell];
                sum += c.start + c.count + (unsigned)c.i0;
            }
        }
    }
    atomicAdd(&out[0], steps);
    atomicAdd(&out[1], sum);
}

int main()
{
    const int ct = 64, n = 1 << 20;
    const double L = 40.0, vd = L / ct;
    std::vector<double> rays((size_t)n * 6);
    for (int i = 0; i < n; ++i) {
        const double z = 1.0 - (2.0 * i + 1.0) / n, phi = i * M_PI * (3.0 - std::sqrt(5.0)), r = std::sqrt(1 - z * z);
        rays[6 * i] = 0.31 * L; rays[6 * i + 1] = 0.42 * L; rays[6 * i + 2] = 0.37 * L;
        rays[6 * i + 3] = r * std::cos(phi); rays[6 * i + 4] = r * std::sin(phi); rays[6 * i + 5] = z;
    }
    std::vector<unsigned> occ((size_t)ct * ct * ct / 32, 0u);
    std::vector<Cell> cells((size_t)ct * ct * ct);
    unsigned long long h = 88172645463325252ull;
    for (size_t c = 0; c < cells.size(); ++c) {
        h ^= h << 13; h ^= h >> 7; h ^= h << 17;
        cells[c] = {(unsigned)c, 3u, (int)c, (int)c + 1};
        if (h % 10 == 0) occ[c >> 5] |= 1u << (c & 31);
    }
    double* d_rays; unsigned* d_occ; Cell* d_cells; unsigned long long* d_out;
    hipMalloc(&d_rays, rays.size() * 8); hipMalloc(&d_occ, occ.size() * 4); hipMalloc(&d_cells, cells.size() * sizeof(Cell)); hipMalloc(&d_out, 16);
    hipMemcpy(d_rays, rays.data(), rays.size() * 8, hipMemcpyHostToDevice);
    hipMemcpy(d_occ, occ.data(), occ.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(d_cells, cells.data(), cells.size() * sizeof(Cell), hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](const char* name, auto kern, int blocks) {
        float best = 1e9f; unsigned long long out[2] = {0, 0};
        for (int rep = 0; rep < 5; ++rep) {
            hipMemset(d_out, 0, 16);
            hipEventRecord(e0);
            hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 32768, 0, d_rays, d_occ, d_cells, d_out, n, ct, vd);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
        }
        hipMemcpy(out, d_out, 16, hipMemcpyDeviceToHost);
        printf("%-34s %4d workgroups: %.3f ms  %.1f steps/ray  %.1f Gsteps/s  (= %.0f Mrays/s at 52.7 steps per ray)\n", name, blocks, best,
               (double)out[0] / n, out[0] / best / 1e6, out[0] / best / 1e6 / 52.7 * 1e3);
    };
    run("lean (few VGPRs), grid 4096", walk<0>, 4096);
    run("lean (few VGPRs), 4 per CU", walk<0>, 1024);
    run("lean (few VGPRs), 2 per CU", walk<0>, 512);
    run("lean (few VGPRs), 1 per CU", walk<0>, 256);
    run("deferred record load, 3 steps/round", walk_deferred<3>, 1024);
    run("deferred record load, 4 steps/round", walk_deferred<4>, 1024);
    run("deferred record load, 8 steps/round", walk_deferred<8>, 1024);
    run("deferred record load, 1 step/round", walk_deferred<1>, 1024);
    return 0;
}
