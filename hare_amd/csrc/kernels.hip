// kernels.hip -- gfx950 (CDNA4, wave64) kernels for Hare's ray-cast path.
//
// Built with: hipcc --offload-arch=gfx950 --genco -O3 -ffp-contract=off
// (contraction OFF is a correctness requirement: the reference is .NET FP64, which never fuses
//  a*b+c; X_Event parity on near-ties depends on it -- SURVEY.md F5.)
//
// One ray per lane.  The traversal arithmetic is a restatement of
//   Voxel_Grid.Shoot           Voxel_Grid.cs:561-761 (+ :351-552, the poly_origin overload)
//   AABB.Intersect/IsPointInBox AABB_Main.cs:173-260, :75-84
//   Triangle/Quadrilateral.Intersect + RayXtri  Hare_Geometry_Polygons.cs:449-510, :637-660, :784-823
// in hare_math.h.  There is no mailbox on the GPU: re-testing a polygon can never change the
// result because the accept is the strict `t < tmin`.
#include <hip/hip_runtime.h>
#include "hare_device.h"

using namespace hare;

namespace {

constexpr double kTMin = 0.0000000001;           // Voxel_Grid.cs:691
constexpr double kDblMax = 1.7976931348623157e308;

struct Work {
    unsigned int cells, entries, tests;
};

__device__ __forceinline__ void set_miss(XEventRec& e)
{
    // X_Event(): Hare_Geometry_Primitives.cs:454-462
    e.t = 0; e.u = 0; e.v = 0; e.x = 0; e.y = 0; e.z = 0;
    e.poly_id = -1;
    e.hit = 0;
}

__device__ __forceinline__ unsigned long long wave_sum_u32(unsigned int v)
{
    unsigned long long s = v;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    return s;  // valid in lane 0
}

// Per-wave accumulation of the batch counters: one atomic per counter per wave.
__device__ __forceinline__ void flush_counters(unsigned long long* ctr, bool valid, bool hit, const Work& w, bool detailed)
{
    if (!ctr) return;
    const unsigned long long mv = __ballot(valid), mh = __ballot(valid && hit);
    unsigned long long c = 0, e = 0, t = 0;
    if (detailed) {
        c = wave_sum_u32(valid ? w.cells : 0u);
        e = wave_sum_u32(valid ? w.entries : 0u);
        t = wave_sum_u32(valid ? w.tests : 0u);
    }
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(&ctr[CTR_RAYS], (unsigned long long)__popcll(mv));
        atomicAdd(&ctr[CTR_HITS], (unsigned long long)__popcll(mh));
        if (detailed) {
            atomicAdd(&ctr[CTR_CELLS], c);
            atomicAdd(&ctr[CTR_ENTRIES], e);
            atomicAdd(&ctr[CTR_TESTS], t);
        }
    }
}

// Voxel_Grid.Shoot for one ray.  `o` is updated in place when the origin is clipped to OBox
// (AABB.Intersect moves the caller's Ray, F11); returns true when that happened.
template <bool QUADS, bool COUNT>
__device__ __forceinline__ bool trace_voxel(const VoxelArgs& g, V3& o, const V3& d, int e1, int e2,
                                            XEventRec& ev, Work& w)
{
    const int ct = g.ct;
    const double fct = (double)ct;
    double t_start = 0;
    bool moved = false;

    // origin cell: Voxel_Grid.cs:567-569; the range test of :577 is done on the floor() value so
    // NaN / out-of-int-range land on the "outside" side, as int.MinValue does in C#.
    double fx = floor((o.x - g.omin[0]) / g.vd[0]);
    double fy = floor((o.y - g.omin[1]) / g.vd[1]);
    double fz = floor((o.z - g.omin[2]) / g.vd[2]);
    bool inside = (fx >= 0.0 && fx < fct) & (fy >= 0.0 && fy < fct) & (fz >= 0.0 && fz < fct);
    if (!inside) {
        if (!aabb_clip_move(g.omin, g.omax, o, d, t_start)) {   // :579
            set_miss(ev);
            return false;
        }
        moved = true;
        fx = floor((o.x - g.omin[0] + d.x * 1E-6) / g.vd[0]);   // :584-586
        fy = floor((o.y - g.omin[1] + d.y * 1E-6) / g.vd[1]);
        fz = floor((o.z - g.omin[2] + d.z * 1E-6) / g.vd[2]);
        inside = (fx >= 0.0 && fx < fct) & (fy >= 0.0 && fy < fct) & (fz >= 0.0 && fz < fct);
        if (!inside) {   // C# would throw IndexOutOfRangeException at :593; reported as a miss
            set_miss(ev);
            return moved;
        }
    }
    int X = (int)fx, Y = (int)fy, Z = (int)fz;

    // padded box of the current voxel (kept per axis; only the stepped axis is recomputed)
    double lox = voxel_lo(X, g.vd[0], g.omin[0]), hix = voxel_hi(X, g.vd[0], g.omin[0]);
    double loy = voxel_lo(Y, g.vd[1], g.omin[1]), hiy = voxel_hi(Y, g.vd[1], g.omin[1]);
    double loz = voxel_lo(Z, g.vd[2], g.omin[2]), hiz = voxel_hi(Z, g.vd[2], g.omin[2]);

    // DDA setup: Voxel_Grid.cs:589-632
    int stepX, stepY, stepZ;
    double tMaxX, tMaxY, tMaxZ, tDeltaX, tDeltaY, tDeltaZ;
    if (d.x < 0) { stepX = -1; tMaxX = (lox - o.x) / d.x; tDeltaX = g.vd[0] / d.x * -1.0; }
    else         { stepX = 1;  tMaxX = (hix - o.x) / d.x; tDeltaX = g.vd[0] / d.x * 1.0; }
    if (d.y < 0) { stepY = -1; tMaxY = (loy - o.y) / d.y; tDeltaY = g.vd[1] / d.y * -1.0; }
    else         { stepY = 1;  tMaxY = (hiy - o.y) / d.y; tDeltaY = g.vd[1] / d.y * 1.0; }
    if (d.z < 0) { stepZ = -1; tMaxZ = (loz - o.z) / d.z; tDeltaZ = g.vd[2] / d.z * -1.0; }
    else         { stepZ = 1;  tMaxZ = (hiz - o.z) / d.z; tDeltaZ = g.vd[2] / d.z * 1.0; }

    bool have = false;                       // Xpt != null
    double hx = 0, hy = 0, hz = 0, tmin = kDblMax;
    int pid = -1;

    for (;;) {
        const CellRec c = g.cells[(X * ct + Y) * ct + Z];
        if (COUNT) { w.cells++; w.entries += c.count; }
        for (unsigned int q = c.start, qe = c.start + c.count; q < qe; ++q) {
            const int i = g.items[q];
            if (i == e1 || i == e2) continue;                     // :477
            if (COUNT) w.tests++;
            const PolyRec& p = g.polys[i];
            double t;
            const bool quad = QUADS && p.nverts == 4;
            if (poly_fast(p, quad, o, d, t) && t > kTMin) {       // :691
                if (t < tmin) {                                   // :693
                    have = true;
                    hx = o.x + d.x * t;                           // Polygons.cs:652
                    hy = o.y + d.y * t;
                    hz = o.z + d.z * t;
                    tmin = t;
                    pid = i;
                }
            }
        }
        // :705  IsPointInBox on the CURRENT padded voxel (AABB_Main.cs:75-84)
        if (have && !(hx < lox) && !(hy < loy) && !(hz < loz) && !(hx > hix) && !(hy > hiy) && !(hz > hiz)) {
            ev.t = tmin + t_start;                                // :707
            ev.u = 0; ev.v = 0;
            ev.x = hx; ev.y = hy; ev.z = hz;
            ev.poly_id = pid;
            ev.hit = 1;
            return moved;
        }
        // next voxel: :713-759 (strict '<'; ties go to Z, then Y); leaving the grid is a miss even
        // with a pending hit (F12)
        if (tMaxX < tMaxY) {
            if (tMaxX < tMaxZ) {
                X += stepX;
                if (X < 0 || X >= ct) break;
                tMaxX = tMaxX + tDeltaX;
                lox = voxel_lo(X, g.vd[0], g.omin[0]); hix = voxel_hi(X, g.vd[0], g.omin[0]);
            } else {
                Z += stepZ;
                if (Z < 0 || Z >= ct) break;
                tMaxZ = tMaxZ + tDeltaZ;
                loz = voxel_lo(Z, g.vd[2], g.omin[2]); hiz = voxel_hi(Z, g.vd[2], g.omin[2]);
            }
        } else {
            if (tMaxY < tMaxZ) {
                Y += stepY;
                if (Y < 0 || Y >= ct) break;
                tMaxY = tMaxY + tDeltaY;
                loy = voxel_lo(Y, g.vd[1], g.omin[1]); hiy = voxel_hi(Y, g.vd[1], g.omin[1]);
            } else {
                Z += stepZ;
                if (Z < 0 || Z >= ct) break;
                tMaxZ = tMaxZ + tDeltaZ;
                loz = voxel_lo(Z, g.vd[2], g.omin[2]); hiz = voxel_hi(Z, g.vd[2], g.omin[2]);
            }
        }
    }
    set_miss(ev);
    return moved;
}

template <bool QUADS, bool COUNT>
__device__ __forceinline__ void voxel_shoot_body(const VoxelArgs& g, const ShootIO& io)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool valid = i < io.n;
    XEventRec ev;
    set_miss(ev);
    Work w = {0, 0, 0};
    if (valid) {
        const RayRec r = io.rays[i];
        V3 o = {r.x, r.y, r.z};
        const V3 d = {r.dx, r.dy, r.dz};
        const int e1 = io.excl1 ? io.excl1[i] : -1;
        const int e2 = io.excl2 ? io.excl2[i] : -1;
        const bool moved = trace_voxel<QUADS, COUNT>(g, o, d, e1, e2, ev, w);
        io.out[i] = ev;
        if (moved && (io.flags & SHOOT_WRITEBACK_ORIGIN)) {
            io.rays[i].x = o.x;
            io.rays[i].y = o.y;
            io.rays[i].z = o.z;
        }
    }
    flush_counters(io.ctr, valid, ev.hit != 0, w, COUNT);
}

}  // namespace

extern "C" {

// K1: Voxel_Grid.Shoot, triangles only
__global__ __launch_bounds__(256) void hare_voxel_shoot_tri(VoxelArgs g, ShootIO io)
{
    voxel_shoot_body<false, false>(g, io);
}
// K1 with quadrilaterals present
__global__ __launch_bounds__(256) void hare_voxel_shoot_quad(VoxelArgs g, ShootIO io)
{
    voxel_shoot_body<true, false>(g, io);
}
// K1 + exact work counters (cells / list entries / tests) for diagnostics
__global__ __launch_bounds__(256) void hare_voxel_shoot_count(VoxelArgs g, ShootIO io)
{
    voxel_shoot_body<true, true>(g, io);
}

}  // extern "C"
