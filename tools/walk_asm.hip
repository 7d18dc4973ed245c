// Developer microbenchmark (round 6): the walk task of K1q as the compiler writes it (selects) against the hand-written
// exec-masked step loop (hare_amd/csrc/voxel_walk.h), in the production shape: tasks of at most 16 steps, a lane stops at an
// occupied voxel or when it leaves the grid, twelve waves per CU, all lanes walking.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -I hare_amd/csrc tools/walk_asm.hip -o /tmp/walk_asm && /tmp/walk_asm
// Both kernels must report the same steps and the same checksum (the voxels visited and the tMax bits at every stop).
// MODE 2 (round 6): the EXACT closed-form skip over empty aligned 4^3 blocks that round 4 wrote down -- the DDA is a merge of
// three sequences tMax + j tDelta (sequential adds) under "smaller first, the later axis on a tie", so the axis that leaves the block first
// and the steps the other two have taken by then follow from <= 3 adds and compares per axis -- beside the compiler's step (a lane jumps when
// its block is empty, steps otherwise).  Same checksum (it IS exact); its rate against the hand-written step is the measurement asked for.
#include <hip/hip_runtime.h>
#pragma clang diagnostic ignored "-Wunused-value"
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "voxel_walk.h"

template <int MODE>
__global__ __launch_bounds__(768) void walk(const double* rays, const unsigned* occ, unsigned long long* out, int n, int ct, double vd, int occ_shift, int occ_cd)
{
    extern __shared__ unsigned locc[];
    const int cdw = occ_shift ? (occ_cd * occ_cd * occ_cd + 31) / 32 : ct * ct * ct / 32;
    for (int k = threadIdx.x; k < cdw; k += blockDim.x) locc[k] = occ[k];
    if (MODE == 2) {
        const int nb = (ct + 3) >> 2, nbw = (nb * nb * nb + 31) / 32;
        for (int k = threadIdx.x; k < nbw; k += blockDim.x) locc[cdw + k] = 0u;
        __syncthreads();
        for (int c = threadIdx.x; c < ct * ct * ct; c += blockDim.x)
            if ((locc[c >> 5] >> (c & 31)) & 1u) {
                const int z = c % ct, y = (c / ct) % ct, x = c / (ct * ct), b = ((x >> 2) * nb + (y >> 2)) * nb + (z >> 2);
                atomicOr(&locc[cdw + (b >> 5)], 1u << (b & 31));
            }
    }
    __syncthreads();
    unsigned long long steps = 0, sum = 0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const double ox = rays[6 * i], oy = rays[6 * i + 1], oz = rays[6 * i + 2];
        const double dx = rays[6 * i + 3], dy = rays[6 * i + 4], dz = rays[6 * i + 5];
        int X = (int)floor(ox / vd), Y = (int)floor(oy / vd), Z = (int)floor(oz / vd);
        const int dx1 = dx < 0 ? -1 : 1, dy1 = dy < 0 ? -1 : 1, dz1 = dz < 0 ? -1 : 1;
        double tMaxX = ((dx < 0 ? X : X + 1) * vd - ox) / dx, tMaxY = ((dy < 0 ? Y : Y + 1) * vd - oy) / dy,
               tMaxZ = ((dz < 0 ? Z : Z + 1) * vd - oz) / dz;
        const double tDeltaX = vd / dx * dx1, tDeltaY = vd / dy * dy1, tDeltaZ = vd / dz * dz1;
        bool alive = true;
        while (__ballot(alive)) {
            bool walking = alive;
            if (MODE == 0) {
                const int n0 = __popcll(__ballot(walking));
                const int walk_min = n0 / 3 < 20 ? n0 / 3 : 20;
#pragma unroll 1
                for (int k = 0; k < 16; ++k) {
                    const unsigned long long wm = __ballot(walking);
                    if (wm == 0 || (k > 0 && __popcll(wm) < walk_min)) break;
                    if (walking) {
                        const bool cxy = tMaxX < tMaxY, cxz = tMaxX < tMaxZ, cyz = tMaxY < tMaxZ;
                        const bool sx = cxy & cxz, sy = (!cxy) & cyz, sz = !(sx | sy);
                        const double nX = tMaxX + tDeltaX, nY = tMaxY + tDeltaY, nZ = tMaxZ + tDeltaZ;
                        X += sx ? dx1 : 0; Y += sy ? dy1 : 0; Z += sz ? dz1 : 0;
                        tMaxX = sx ? nX : tMaxX; tMaxY = sy ? nY : tMaxY; tMaxZ = sz ? nZ : tMaxZ;
                        const bool o = ((unsigned)X >= (unsigned)ct) | ((unsigned)Y >= (unsigned)ct) | ((unsigned)Z >= (unsigned)ct);
                        const int cell = o ? 0 : (X * ct + Y) * ct + Z;
                        const unsigned bit = occ_shift ? (unsigned)((((o ? 0 : X) >> occ_shift) * occ_cd + ((o ? 0 : Y) >> occ_shift)) * occ_cd + ((o ? 0 : Z) >> occ_shift)) : (unsigned)cell;
                        const bool oc = (locc[bit >> 5] >> (bit & 31)) & 1u;
                        walking = !o && !oc;
                        if (!o) ++steps;
                    }
                }
            } else if (MODE == 2) {
                // exact block skip: `bocc` (LDS, behind the voxel bitmap) has a bit per aligned 4^3 block that holds an occupied voxel
                const unsigned* bocc = locc + cdw;
                const int nb = (ct + 3) >> 2;
#pragma unroll 1
                for (int k = 0; k < 16; ++k) {
                    const unsigned long long wm = __ballot(walking);
                    if (wm == 0) break;
                    if (walking) {
                        const int bxi = ((X >> 2) * nb + (Y >> 2)) * nb + (Z >> 2);
                        const bool finite = fabs(tMaxX) < 1e300 && fabs(tMaxY) < 1e300 && fabs(tMaxZ) < 1e300 && fabs(tDeltaX) < 1e300 && fabs(tDeltaY) < 1e300 && fabs(tDeltaZ) < 1e300;
                        const bool empty_block = finite && !((bocc[bxi >> 5] >> (bxi & 31)) & 1u);
                        if (empty_block) {
                            const int kx = dx1 > 0 ? 4 - (X & 3) : (X & 3) + 1, ky = dy1 > 0 ? 4 - (Y & 3) : (Y & 3) + 1, kz = dz1 > 0 ? 4 - (Z & 3) : (Z & 3) + 1;
                            const double ax1 = tMaxX + tDeltaX, ax2 = ax1 + tDeltaX, ax3 = ax2 + tDeltaX, ax4 = ax3 + tDeltaX;
                            const double ay1 = tMaxY + tDeltaY, ay2 = ay1 + tDeltaY, ay3 = ay2 + tDeltaY, ay4 = ay3 + tDeltaY;
                            const double az1 = tMaxZ + tDeltaZ, az2 = az1 + tDeltaZ, az3 = az2 + tDeltaZ, az4 = az3 + tDeltaZ;
                            auto sel = [](int j, double a0, double a1, double a2, double a3, double a4) { return j == 0 ? a0 : (j == 1 ? a1 : (j == 2 ? a2 : (j == 3 ? a3 : a4))); };
                            const double Ex = sel(kx - 1, tMaxX, ax1, ax2, ax3, ax4), Ey = sel(ky - 1, tMaxY, ay1, ay2, ay3, ay4), Ez = sel(kz - 1, tMaxZ, az1, az2, az3, az4);
                            // the exit axis: the DDA's own choice among the three exit elements (x iff Ex < Ey && Ex < Ez; else y iff Ey < Ez; else z)
                            const bool ex = (Ex < Ey) & (Ex < Ez), ey = (!(Ex < Ey)) & (Ey < Ez), ez = !(ex | ey);
                            const double E = ex ? Ex : (ey ? Ey : Ez);
                            // elements of another axis that come BEFORE the exit element: a < E, or a == E when that axis is the later one
                            auto cnt = [&](bool later, int kk, double a0, double a1, double a2) {
                                int c = 0;
                                c += (kk > 1 && (a0 < E || (later && a0 == E))) ? 1 : 0;
                                c += (kk > 2 && (a1 < E || (later && a1 == E))) ? 1 : 0;
                                c += (kk > 3 && (a2 < E || (later && a2 == E))) ? 1 : 0;
                                return c;
                            };
                            const int nx = ex ? kx : cnt(false, kx, tMaxX, ax1, ax2);                 // x is never the later axis
                            const int ny = ey ? ky : cnt(ex, ky, tMaxY, ay1, ay2);                    // y is later than x only
                            const int nz = ez ? kz : cnt(true, kz, tMaxZ, az1, az2);                  // z is later than both
                            X += dx1 * nx; Y += dy1 * ny; Z += dz1 * nz;
                            tMaxX = sel(nx, tMaxX, ax1, ax2, ax3, ax4); tMaxY = sel(ny, tMaxY, ay1, ay2, ay3, ay4); tMaxZ = sel(nz, tMaxZ, az1, az2, az3, az4);
                            steps += (unsigned)(nx + ny + nz) - 1u;        // the voxels passed on the way count as walked (the arrival is counted below)
                        } else {
                            const bool cxy = tMaxX < tMaxY, cxz = tMaxX < tMaxZ, cyz = tMaxY < tMaxZ;
                            const bool sx = cxy & cxz, sy = (!cxy) & cyz, sz = !(sx | sy);
                            const double nX = tMaxX + tDeltaX, nY = tMaxY + tDeltaY, nZ = tMaxZ + tDeltaZ;
                            X += sx ? dx1 : 0; Y += sy ? dy1 : 0; Z += sz ? dz1 : 0;
                            tMaxX = sx ? nX : tMaxX; tMaxY = sy ? nY : tMaxY; tMaxZ = sz ? nZ : tMaxZ;
                        }
                        const bool o = ((unsigned)X >= (unsigned)ct) | ((unsigned)Y >= (unsigned)ct) | ((unsigned)Z >= (unsigned)ct);
                        const int cell = o ? 0 : (X * ct + Y) * ct + Z;
                        const bool oc = (locc[cell >> 5] >> (cell & 31)) & 1u;
                        walking = !o && !oc;
                        if (!o) ++steps;
                    }
                }
            } else {
                const int n0 = __popcll(__ballot(walking));
                const int walk_min = n0 / 3 < 20 ? n0 / 3 : 20;
                unsigned taken = 0, iters = 0;
                if (occ_shift) hare_walk::walk_steps<true, true>(tMaxX, tMaxY, tMaxZ, tDeltaX, tDeltaY, tDeltaZ, X, Y, Z, dx1, dy1, dz1, walking, (unsigned)ct, 16u, (unsigned)walk_min, 0u, (unsigned)occ_shift, (unsigned)occ_cd, taken, iters);
                else           hare_walk::walk_steps<false, true>(tMaxX, tMaxY, tMaxZ, tDeltaX, tDeltaY, tDeltaZ, X, Y, Z, dx1, dy1, dz1, walking, (unsigned)ct, 16u, (unsigned)walk_min, 0u, 0u, 0u, taken, iters);
                steps += taken;
            }
            const bool o = ((unsigned)X >= (unsigned)ct) | ((unsigned)Y >= (unsigned)ct) | ((unsigned)Z >= (unsigned)ct);
            if (alive && o) alive = false;
            if (alive && !walking) sum += (unsigned long long)((X * ct + Y) * ct + Z) + (unsigned long long)(__double_as_longlong(tMaxX) ^ __double_as_longlong(tMaxY) ^ __double_as_longlong(tMaxZ)) % 1000003ull;
        }
    }
    atomicAdd(&out[0], steps);
    atomicAdd(&out[1], sum);
}

int main()
{
    const int n = 1 << 20;
    for (int cfg = 0; cfg < 2; ++cfg) {
        const int ct = cfg ? 128 : 64, occ_shift = cfg ? 1 : 0, occ_cd = ct >> occ_shift;
        const double L = 40.0, vd = L / ct;
        std::vector<double> rays((size_t)n * 6);
        for (int i = 0; i < n; ++i) {
            const double z = 1.0 - (2.0 * i + 1.0) / n, phi = i * M_PI * (3.0 - std::sqrt(5.0)), r = std::sqrt(1 - z * z);
            rays[6 * i] = 0.31 * L; rays[6 * i + 1] = 0.42 * L; rays[6 * i + 2] = 0.37 * L;
            rays[6 * i + 3] = r * std::cos(phi); rays[6 * i + 4] = r * std::sin(phi); rays[6 * i + 5] = z;
        }
        std::vector<unsigned> occ((size_t)occ_cd * occ_cd * occ_cd / 32, 0u);
        // a room: walls one voxel (block) inside the grid's faces, a balcony slab, 1 % clutter -- most aligned 4^3 blocks are empty, as in the hall
        unsigned long long h = 88172645463325252ull;
        const int cd = occ_cd;
        for (size_t c = 0; c < (size_t)cd * cd * cd; ++c) {
            h ^= h << 13; h ^= h >> 7; h ^= h << 17;
            const int z = (int)(c % cd), y = (int)((c / cd) % cd), x = (int)(c / ((size_t)cd * cd));
            const bool wall = x == 1 || y == 1 || z == 1 || x == cd - 2 || y == cd - 2 || z == cd - 2;
            const bool balcony = z == cd / 3 && x < cd / 3;
            if (wall || balcony || h % 100 == 0) occ[c >> 5] |= 1u << (c & 31);
        }
        double* d_rays; unsigned* d_occ; unsigned long long* d_out;
        hipMalloc(&d_rays, rays.size() * 8); hipMalloc(&d_occ, occ.size() * 4); hipMalloc(&d_out, 16);
        hipMemcpy(d_rays, rays.data(), rays.size() * 8, hipMemcpyHostToDevice);
        hipMemcpy(d_occ, occ.data(), occ.size() * 4, hipMemcpyHostToDevice);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        auto run = [&](const char* name, auto kern) {
            float best = 1e9f; unsigned long long out[2] = {0, 0};
            for (int rep = 0; rep < 5; ++rep) {
                hipMemset(d_out, 0, 16);
                hipEventRecord(e0);
                hipLaunchKernelGGL(kern, dim3(256), dim3(768), 32768 + 120 * 1024, 0, d_rays, d_occ, d_out, n, ct, vd, occ_shift, occ_cd);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
            }
            hipMemcpy(out, d_out, 16, hipMemcpyDeviceToHost);
            printf("D=%d shift %d  %-28s %.3f ms  %.2f steps/ray  %.1f Gsteps/s  checksum %llu\n", ct, occ_shift, name, best, (double)out[0] / n, out[0] / best / 1e6, out[1]);
        };
        hipFuncSetAttribute((const void*)walk<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        hipFuncSetAttribute((const void*)walk<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        run("compiler (selects)", walk<0>);
        run("hand-written step loop", walk<1>);
        if (!occ_shift) {
            hipFuncSetAttribute((const void*)walk<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            run("exact 4^3 block skip + step", walk<2>);
        }
        hipFree(d_rays); hipFree(d_occ); hipFree(d_out);
    }
    return 0;
}
