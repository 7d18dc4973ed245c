// hare_trace.h -- the reference's Shoot loops, one ray at a time, written once for host (g++) and device (hipcc).
//
//   trace_voxel    Voxel_Grid.Shoot     Voxel_Grid.cs:561-761 (+ :351-552, the poly_origin overload)
//   trace_octree   Octree.Shoot         "Octree - alt.cs":159-306
//   trace_kdtree   KDTree.Shoot         KDTree.cs:204-361
//
// The simple one-ray-per-lane kernels (hare_voxel_shoot_*, hare_octree_shoot*, hare_kdtree_shoot*: the counting
// and A/B kernels) instantiate these on the device; hare_shoot_one (host_trace.cpp) instantiates the SAME code on
// the host for single-ray callers (Spatial_Partition.Shoot as the reference exposes it, Spatial_Partition.cs:32-33).
// The argument blocks (VoxelArgs, OctreeArgs, KdArgs) hold plain pointers: device memory in a kernel, the scene's
// host mirror on the host.  The production batch kernels (hare_voxel_persist_*, hare_octree_persist) are separate
// state machines in kernels.hip that visit the same candidates in the same order.
//
// Product code.  Must not include anything from oracle/.
#pragma once
#include "hare_device.h"

namespace hare {

constexpr double kTMin = 0.0000000001;           // Voxel_Grid.cs:691
constexpr double kDblMax = 1.7976931348623157e308;

struct Work {
    unsigned int cells, entries, tests;
};

HARE_HD void set_miss(XEventRec& e)
{
    // X_Event(): Hare_Geometry_Primitives.cs:454-462
    e.t = 0; e.u = 0; e.v = 0; e.x = 0; e.y = 0; e.z = 0;
    e.poly_id = -1;
    e.hit = 0;
}

// Voxel_Grid.Shoot for one ray.  `o` is updated in place when the origin is clipped to OBox
// (AABB.Intersect moves the caller's Ray, F11); returns true when that happened.
template <bool QUADS, bool COUNT>
HARE_HD bool trace_voxel(const VoxelArgs& g, V3& o, const V3& d, int e1, int e2,
                                            XEventRec& ev, Work& w, double* tmin_local = nullptr)
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
            const double* v3 = (QUADS && g.quads && g.quads[i].nverts == 4) ? g.quads[i].v3 : nullptr;
            if (poly_fast(p, v3, o, d, t) && t > kTMin) {         // :691
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
            if (tmin_local) *tmin_local = tmin;
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

// ------------------------------------------------------------------------------------------------
// Octree.Shoot: "Octree - alt.cs":159-284.
//
// The reference keeps a LIFO Stack<(node,tmin,tmax)>: an interior node pushes its surviving children
// in ComputeTraversalOrder order (:286-306) and they pop in reverse.  Whether a child is pushed
// (:268) depends only on the ray, the child's box and the parent's interval -- never on the hit
// found so far -- so the same visit sequence is produced by a depth-first walk that keeps ONE frame
// per level {first_child, cursor, parent interval} and enumerates children lazily from order[7]
// down to order[0].  That needs (max_depth) frames per lane instead of 7*max_depth+1 stack entries;
// frames live in LDS, laid out [level][lane] so that a wave's accesses never bank-conflict.
// order[k] = k ^ mask with mask = (dx<0)<<2 | (dy<0)<<1 | (dz<0).
struct OctFrames {
    int* first;      // [levels][blockDim]
    int* cursor;     // [levels][blockDim]
    double* a;       // [levels][blockDim]
    double* b;       // [levels][blockDim]
};

template <bool COUNT, bool CULL>
HARE_HD void trace_octree(const OctreeArgs& g, const OctFrames& fr, int tid, int nt, const V3& o, const V3& d,
                                             int e1, int e2, XEventRec& ev, Work& w)
{
    const double invDx = fabs(d.x) > 1e-16 ? 1.0 / d.x : 1e16;   // :165-167
    const double invDy = fabs(d.y) > 1e-16 ? 1.0 / d.y : 1e16;
    const double invDz = fabs(d.z) > 1e-16 ? 1.0 / d.z : 1e16;
    const bool nx = invDx < 0, ny = invDy < 0, nz = invDz < 0;
    const int mask = ((d.x >= 0 ? 0 : 1) << 2) | ((d.y >= 0 ? 0 : 1) << 1) | (d.z >= 0 ? 0 : 1);

    auto slab = [&](const OctNode& n, double& tmin, double& tmax) {
        double tx0 = (n.bmin[0] - o.x) * invDx, tx1 = (n.bmax[0] - o.x) * invDx;
        double ty0 = (n.bmin[1] - o.y) * invDy, ty1 = (n.bmax[1] - o.y) * invDy;
        double tz0 = (n.bmin[2] - o.z) * invDz, tz1 = (n.bmax[2] - o.z) * invDz;
        if (nx) { const double s = tx0; tx0 = tx1; tx1 = s; }
        if (ny) { const double s = ty0; ty0 = ty1; ty1 = s; }
        if (nz) { const double s = tz0; tz0 = tz1; tz1 = s; }
        tmin = net_max(net_max(tx0, ty0), tz0);
        tmax = net_min(net_min(tx1, ty1), tz1);
    };

    set_miss(ev);
    const float dfx = (float)d.x, dfy = (float)d.y, dfz = (float)d.z;
    const float dm = fabsf(dfx) + fabsf(dfy) + fabsf(dfz);
    int m0 = -1, m1 = -1, m2 = -1, m3 = -1;       // the four polygons this ray tested last
    double rmin, rmax;
    slab(g.nodes[0], rmin, rmax);
    if (rmax < rmin || rmax < 0) return;                          // :185

    bool hit = false;
    double closestT = kDblMax;
    int lvl = -1;                  // top frame
    // the item "popped" next: starts with the root
    int cur = 0;
    double ca = rmin, cb = rmax;
    bool have_item = true;

    for (;;) {
        if (!have_item) {
            // pop: next surviving child of the deepest open frame, scanning order[7] .. order[0]
            while (lvl >= 0) {
                int k = fr.cursor[lvl * nt + tid];
                const int first = fr.first[lvl * nt + tid];
                const double pa = fr.a[lvl * nt + tid], pb = fr.b[lvl * nt + tid];
                while (k >= 0) {
                    const int c = first + (k ^ mask);
                    --k;
                    double tmn, tmx;
                    slab(g.nodes[c], tmn, tmx);
                    if (tmx < tmn || tmx < 0 || tmn > pb || tmx < pa) continue;      // :268
                    cur = c;
                    ca = net_max(tmn, pa);                                            // :271
                    cb = net_min(tmx, pb);
                    have_item = true;
                    break;
                }
                fr.cursor[lvl * nt + tid] = k;
                if (have_item) break;
                --lvl;
            }
            if (!have_item) break;   // stack empty
        }
        have_item = false;
        if (cb < ca || cb < 0) continue;                          // :207
        if (hit && closestT <= ca) continue;                      // :210
        if (COUNT) w.cells++;
        const OctNode& node = g.nodes[cur];
        const int fc = node.first_child;
        if (fc < 0) {
            const int is = node.item_start, ic = node.item_count;
            if (COUNT) w.entries += ic;
            for (int q = is; q < is + ic; ++q) {
                const int i = g.items[q];
                if (i == e1 || i == e2) continue;                 // :218
                if (COUNT) w.tests++;
                const PolyRec& p = g.polys[i];
                if (CULL) {
                    // Not in the reference (its mailbox is commented out, :221-222): loose leaves overlap, so a
                    // ray meets the same polygon in several leaves.  Skipping one it has just tested, and
                    // candidates the conservative FP32 cull proves to be misses, cannot change any accepted
                    // hit (strict `t < closestT`), only saves the FP64 test.
                    if (i == m0 || i == m1 || i == m2 || i == m3) continue;
                    m3 = m2; m2 = m1; m1 = m0; m0 = i;
                    if (cull_fp32((float)(o.x - p.v0[0]), (float)(o.y - p.v0[1]), (float)(o.z - p.v0[2]), dfx, dfy, dfz, dm,
                                  p.e1f, p.e2f))
                        continue;
                }
                double t, u, v;
                const double* v3 = (g.quads && g.quads[i].nverts == 4) ? g.quads[i].v3 : nullptr;
                if (poly_full(p, v3, o, d, t, u, v) && t > kTMin) {              // :224
                    if (t < closestT) {
                        closestT = t;
                        ev.t = t; ev.u = u; ev.v = v;
                        ev.x = o.x + d.x * t; ev.y = o.y + d.y * t; ev.z = o.z + d.z * t;
                        ev.poly_id = i;
                        ev.hit = 1;
                        hit = true;
                        if (closestT <= ca) return;               // :233 early termination
                    }
                }
            }
        } else {
            ++lvl;
            fr.first[lvl * nt + tid] = fc;
            fr.cursor[lvl * nt + tid] = 7;
            fr.a[lvl * nt + tid] = ca;
            fr.b[lvl * nt + tid] = cb;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// KDTree.Shoot: KDTree.cs:204-361.  Both children of every interior node are pushed (:355-356), so
// every leaf is visited; the split-plane logic only fixes the ORDER, which decides
// exact-t ties.  Explicit node stack in LDS, [slot][lane].
// CULL (device only, round 4): the conservative FP32 pre-cull in front of the exact test, as in every other kernel -- the reference's
// query visits EVERY leaf, so all P polygons are candidates of every ray and nearly all of them fail: a candidate the cull rejects
// is one the exact test is certain to reject, results unchanged.
template <bool COUNT, bool CULL = false>
HARE_HD void trace_kdtree(const KdArgs& g, int* stack, int tid, int nt, const V3& o, const V3& d, int e1, int e2,
                                             XEventRec& ev, Work& w)
{
    set_miss(ev);
#if defined(__HIPCC__)
    CullRay cray = {};
    if (CULL) cray = cull_ray(g, o.x, o.y, o.z, d.x, d.y, d.z);
#endif
    double closestT = kDblMax;
    int sp = 0;
    stack[tid] = 0;
    sp = 1;
    const double oo[3] = {o.x, o.y, o.z};
    const double dd[3] = {d.x, d.y, d.z};
#if defined(__HIPCC__)
    // The subtrees' tight boxes (device_scene.cpp: make_tight_boxes; as in the octree kernels): KDTree.Shoot visits EVERY leaf and lets RayXtri
    // say no to all but a few polygons.  A ray that misses the box of all polygons below a node -- or, holding a hit, reaches that box
    // behind it -- cannot make RayXtri accept (t > 1e-10 && t < closestT, :233) any of them: the node is dropped, the event unchanged.
    // Only for finite rays with the origin within the range the boxes' margin is sized for; 1/d may be infinite (a zero component):
    // fmax / fmin then leave that slab out unless the origin lies outside it, where +-inf says miss -- which it is.
    bool tight_ok = false;
    double ivx = 0, ivy = 0, ivz = 0;
    if (CULL && g.tight != nullptr) {
        tight_ok = fabs(o.x - g.tight_mid[0]) <= g.tight_rad && fabs(o.y - g.tight_mid[1]) <= g.tight_rad && fabs(o.z - g.tight_mid[2]) <= g.tight_rad &&
                   fabs(d.x) < 1e300 && fabs(d.y) < 1e300 && fabs(d.z) < 1e300;
        ivx = 1.0 / d.x; ivy = 1.0 / d.y; ivz = 1.0 / d.z;
    }
#endif
    while (sp > 0) {
        --sp;
        const int node = stack[sp * nt + tid];
        const KdNodeRec& cur = g.nodes[node];
#if defined(__HIPCC__)
        if (CULL && tight_ok) {
            const float4* tp = reinterpret_cast<const float4*>(g.tight) + 2 * (size_t)node;
            const float4 tb0 = tp[0], tb1 = tp[1];
            double ux0 = ((double)tb0.x - o.x) * ivx, ux1 = ((double)tb0.w - o.x) * ivx;
            double uy0 = ((double)tb0.y - o.y) * ivy, uy1 = ((double)tb1.x - o.y) * ivy;
            double uz0 = ((double)tb0.z - o.z) * ivz, uz1 = ((double)tb1.y - o.z) * ivz;
            const double un = __builtin_fmax(__builtin_fmax(__builtin_fmin(ux0, ux1), __builtin_fmin(uy0, uy1)), __builtin_fmin(uz0, uz1));
            const double uf = __builtin_fmin(__builtin_fmin(__builtin_fmax(ux0, ux1), __builtin_fmax(uy0, uy1)), __builtin_fmax(uz0, uz1));
            if ((uf < un) | (uf < 0) | ((ev.hit != 0) & (closestT <= un))) continue;
        }
#endif
        if (COUNT) w.cells++;
        if (cur.left < 0 && cur.right < 0) {
            const int is = cur.item_start, ic = cur.item_count;
            if (COUNT) w.entries += ic;
            for (int q = is; q < is + ic; ++q) {
                const int i = g.items[q];
                if (i == e1 || i == e2) continue;                 // :221
                if (COUNT) w.tests++;
#if defined(__HIPCC__)
                if (CULL && cull_test(g, cray, cull_load(g, i))) continue;
#endif
                const PolyRec& p = g.polys[i];
                double t, u, v;
                const double* v3 = (g.quads && g.quads[i].nverts == 4) ? g.quads[i].v3 : nullptr;
                if (poly_full(p, v3, o, d, t, u, v) && t > kTMin) {              // :233
                    if (t < closestT) {
                        closestT = t;
                        ev.t = t; ev.u = u; ev.v = v;
                        ev.x = o.x + d.x * t; ev.y = o.y + d.y * t; ev.z = o.z + d.z * t;
                        ev.poly_id = i;
                        ev.hit = 1;
                    }
                }
            }
        } else {
            // :249-353: the three SplitAxis branches are one pattern; the other two axes are checked
            // in ascending axis order
            const int a = cur.axis;
            const int b = (a == 0) ? 1 : 0, c = (a == 2) ? 1 : 2;
            double oa, da, ob, db, oc, dc, bmaxb, bminb, bmaxc, bminc;
            // select without dynamic register indexing
            oa = a == 0 ? oo[0] : (a == 1 ? oo[1] : oo[2]);
            da = a == 0 ? dd[0] : (a == 1 ? dd[1] : dd[2]);
            ob = b == 0 ? oo[0] : oo[1];
            db = b == 0 ? dd[0] : dd[1];
            oc = c == 1 ? oo[1] : oo[2];
            dc = c == 1 ? dd[1] : dd[2];
            bmaxb = b == 0 ? cur.bmax[0] : cur.bmax[1];
            bminb = b == 0 ? cur.bmin[0] : cur.bmin[1];
            bmaxc = c == 1 ? cur.bmax[1] : cur.bmax[2];
            bminc = c == 1 ? cur.bmin[1] : cur.bmin[2];
            const double side = oa - cur.split;
            const double tSplit = -side / da;
            const double bS = ob + tSplit * db;
            const double cS = oc + tSplit * dc;
            int first, second;
            if (bS <= bmaxb && bS >= bminb && cS <= bmaxc && cS >= bminc) {
                if (side >= 0) { first = cur.right; second = cur.left; }
                else { first = cur.left; second = cur.right; }
            } else {
                if (side >= 0) { first = cur.left; second = cur.right; }
                else { first = cur.right; second = cur.left; }
            }
            stack[sp * nt + tid] = second;
            stack[(sp + 1) * nt + tid] = first;
            sp += 2;
        }
    }
}

}  // namespace hare
