// hare_math.h -- FP64 arithmetic of Hare's ray-cast path, written once for host (g++) and device
// (hipcc, gfx950).  Every function states the reference lines whose ORDER OF OPERATIONS it
// reproduces (file:line into PachydermAcoustic/Hare); both compilers are run with
// -ffp-contract=off so that a*b+c is a rounded multiply followed by a rounded add, as .NET does it.
//
// Product code.  Must not include anything from oracle/.
#pragma once
#include <stdint.h>
#include <math.h>

#if defined(__HIPCC__)
#define HARE_HD __host__ __device__ __forceinline__
#else
#define HARE_HD inline
#endif

namespace hare {

struct V3 {
    double x, y, z;
};

// One polygon = one 128-byte record = one L2 line: a lane that tests polygon i touches exactly one
// line.  The first 48 bytes serve the conservative FP32 pre-cull (v0 + the two edges from v0 as
// floats: three 16-byte gathers); the exact FP64 test reads the rest.
struct alignas(128) PolyRec {
    double v0[3];      //   0
    float e1f[3];      //  24  (float)(v1 - v0)
    float e2f[3];      //  36  (float)(v2 - v0)
    float ee;          //  48  |e1|_1 * emax, rounded up  } kept for tools; the kernels form both factors from e1f / e2f
    float emax;        //  52  max(|e1|_inf, |e2|_inf)     } (cull_fp32); a quadrilateral has NaN in e1f[0]: never culled
    double v1[3];      //  56
    double v2[3];      //  80
    double n[3];       // 104  Polygon.Normal
};
static_assert(sizeof(PolyRec) == 128, "PolyRec must be one 128-byte line");

// Fourth corner + corner count, kept in a side array that only exists for topologies with
// quadrilaterals (Hare's room meshes are almost always triangles).
struct QuadRec {
    double v3[3];
    int32_t nverts;
    int32_t pad;
};
static_assert(sizeof(QuadRec) == 32, "QuadRec size");

// Hare_math.Dot: Hare_Geometry_Math.cs:43-46
HARE_HD double dot3(double ax, double ay, double az, double bx, double by, double bz)
{
    return (ax * bx) + (ay * by) + (az * bz);
}

// System.Math.Max / Min (double): NaN propagates, +0 > -0 (used at AABB_Main.cs:198-199 and
// "Octree - alt.cs":182-183,265-266,271).
HARE_HD double net_max(double a, double b)
{
    if (a != b) {
        if (!(a != a)) return b < a ? a : b;
        return a;
    }
    return signbit(b) ? a : b;
}
HARE_HD double net_min(double a, double b)
{
    if (a != b) {
        if (!(a != a)) return a < b ? a : b;
        return a;
    }
    return signbit(a) ? a : b;
}

// RayXtri(ref Ray, ref v0, ref v1, ref v2, ref t): Hare_Geometry_Polygons.cs:449-510.
// a,b,c are the three corners in the order the caller passes them.
HARE_HD bool tri_fast(const V3& o, const V3& d, const double* a, const double* b, const double* c, double& t)
{
    const double e1x = b[0] - a[0], e1y = b[1] - a[1], e1z = b[2] - a[2];
    const double e2x = c[0] - a[0], e2y = c[1] - a[1], e2z = c[2] - a[2];
    const double px = d.y * e2z - d.z * e2y;
    const double py = d.z * e2x - d.x * e2z;
    const double pz = d.x * e2y - d.y * e2x;
    const double det = dot3(e1x, e1y, e1z, px, py, pz);
    const double tx = o.x - a[0], ty = o.y - a[1], tz = o.z - a[2];
    const double qx = ty * e1z - tz * e1y;
    const double qy = tz * e1x - tx * e1z;
    const double qz = tx * e1y - ty * e1x;
    if (det > 0.000001) {
        const double u = dot3(tx, ty, tz, px, py, pz);
        if (u < 0.0 || u > det) return false;
        const double v = dot3(d.x, d.y, d.z, qx, qy, qz);
        if (v < 0.0 || u + v > det) return false;
    } else if (det < -0.000001) {
        const double u = dot3(tx, ty, tz, px, py, pz);
        if (u > 0.0 || u < det) return false;
        const double v = dot3(d.x, d.y, d.z, qx, qy, qz);
        if (v > 0.0 || u + v < det) return false;
    } else {
        return false;
    }
    const double invdet = 1.0 / det;
    t = dot3(e2x, e2y, e2z, qx, qy, qz) * invdet;
    return true;
}

// RayXtri(Ray, v0, v1, v2, ref t, ref u, ref v): Hare_Geometry_Polygons.cs:385-435.  The cross
// products go through Hare_math.Cross (Hare_Geometry_Math.cs:66-69), whose y component is
// written -(ax*bz - az*bx).  u and v are `ref`: they keep whatever a failing test wrote.
HARE_HD bool tri_full(const V3& o, const V3& d, const double* a, const double* b, const double* c,
                      double& t, double& u, double& v)
{
    const double e1x = b[0] - a[0], e1y = b[1] - a[1], e1z = b[2] - a[2];
    const double e2x = c[0] - a[0], e2y = c[1] - a[1], e2z = c[2] - a[2];
    const double px = d.y * e2z - d.z * e2y;
    const double py = -(d.x * e2z - d.z * e2x);
    const double pz = d.x * e2y - d.y * e2x;
    const double det = dot3(e1x, e1y, e1z, px, py, pz);
    const double tx = o.x - a[0], ty = o.y - a[1], tz = o.z - a[2];
    const double qx = ty * e1z - tz * e1y;
    const double qy = -(tx * e1z - tz * e1x);
    const double qz = tx * e1y - ty * e1x;
    if (det > 0.000001) {
        u = dot3(tx, ty, tz, px, py, pz);
        if (u < 0.0 || u > det) return false;
        v = dot3(d.x, d.y, d.z, qx, qy, qz);
        if (v < 0.0 || u + v > det) return false;
    } else if (det < -0.000001) {
        u = dot3(tx, ty, tz, px, py, pz);
        if (u > 0.0 || u < det) return false;
        v = dot3(d.x, d.y, d.z, qx, qy, qz);
        if (v > 0.0 || u + v < det) return false;
    } else {
        return false;
    }
    const double invdet = 1.0 / det;
    t = dot3(e2x, e2y, e2z, qx, qy, qz) * invdet;
    u = u * invdet;
    v = v * invdet;
    return true;
}

// Ray_Side: Hare_Geometry_Polygons.cs:601-606 (n < 0 -> false; NaN -> true)
HARE_HD bool ray_side(const V3& d, const double* n)
{
    return !(dot3(d.x, d.y, d.z, n[0], n[1], n[2]) < 0);
}

// Triangle.Intersect fast (Polygons.cs:637-660) / Quadrilateral.Intersect fast (:784-823).
// On a hit returns t; the caller forms the hit point O + d*t (:652).
// v3 = fourth corner for a quadrilateral, nullptr for a triangle.
HARE_HD bool poly_fast(const PolyRec& p, const double* v3, const V3& o, const V3& d, double& t)
{
    t = 0;
    if (ray_side(d, p.n)) {
        if (tri_fast(o, d, p.v0, p.v1, p.v2, t)) return true;
        return v3 && tri_fast(o, d, p.v2, v3, p.v0, t);
    }
    if (tri_fast(o, d, p.v2, p.v1, p.v0, t)) return true;
    return v3 && tri_fast(o, d, p.v0, v3, p.v2, t);
}

// Triangle.Intersect full (Polygons.cs:662-688) / Quadrilateral.Intersect full (:731-782).
HARE_HD bool poly_full(const PolyRec& p, const double* v3, const V3& o, const V3& d, double& t, double& u, double& v)
{
    u = 0;
    v = 0;
    t = 0;
    if (ray_side(d, p.n)) {
        if (tri_full(o, d, p.v0, p.v1, p.v2, t, u, v)) return true;
        return v3 && tri_full(o, d, p.v2, v3, p.v0, t, u, v);
    }
    if (tri_full(o, d, p.v2, p.v1, p.v0, t, u, v)) return true;
    return v3 && tri_full(o, d, p.v0, v3, p.v2, t, u, v);
}

// ---- conservative FP32 pre-cull of one ray/triangle pair -------------------------------------
// Not in the reference: a filter in front of RayXtri.  It may only reject a candidate the exact
// FP64 test is certain to reject, so the set of accepted hits -- and therefore every X_Event -- is
// unchanged.  RayXtri accepts iff |det| > 1e-6 and the three barycentric weights
// (det-u-v)/det, u/det, v/det are all >= 0 (for either corner order Ray_Side picks: reversing the
// order maps (u,v,det) to (-u, u+v-det, -det), the same three weights).  u, v and det are triple
// products a.(b x c): six terms, each a product of three components.  Evaluated in FP32 from
// inputs rounded to FP32, each term carries at most ~7 roundings of 2^-24; the sum of the |terms|
// is at most 2*|a|_1*|b|_inf*|c|_inf.  With G = 2^-18 (>= 4x the worst case) the margins are
//   M_uv  = G * |tv|_1 * |d|_1 * emax          (u, v;   emax >= |e1|_inf, |e2|_inf)
//   M_det = G * |d|_1 * ee                     (det;    ee   >= |e1|_1 * emax)
// plus 1e-30 against underflow.  Candidates whose determinant sign is not certain (|det| <= M_det)
// are kept.  NaN/inf anywhere makes every comparison false: the candidate is kept.
// The two factors of the margins are formed here from the FP32 edges themselves (8 instructions) rather than read from the
// record: a fourth 16-byte gather per candidate costs more in the texture-address unit than the arithmetic does in the
// VALU.  Rounding of the edges to FP32 and of the sums below loses at most ~8 x 2^-24 relative; both factors are inflated
// by 2^-20 twice over that.  A quadrilateral's record carries NaN in e1f[0]: every comparison fails, it is never culled.
// `tv_err`: an additional ABSOLUTE error bound on each component of tv (the 32-byte records hold v0 quantised to 21 bits per
// axis, hare_device.h: tv is then off by up to half a quantisation step plus its own FP32 roundings).  u = tv . (d x e2) and
// v = d . (tv x e1) move by at most |dtv|_1 |d x e2|_inf <= 3 tv_err * 2 |d|_inf emax and |d|_1 |dtv x e1|_inf <= |d|_1 * 2 tv_err
// emax: both within 6 * tv_err * |d|_1 * emax, which is added to M_uv (det does not involve tv).
HARE_HD bool cull_fp32(float tvx, float tvy, float tvz, float dx, float dy, float dz, float dm /*|d|_1*/,
                       const float* e1, const float* e2, float tv_err = 0.0f)
{
    // Explicit fused multiply-adds (the build contracts nothing by itself): this is the filter, not the
    // reference arithmetic -- a fused term has one rounding instead of two, so the bound above still holds.
    const float G = 3.814697265625e-06f;   // 2^-18
    const float UP = 1.00000095367431640625f;   // 1 + 2^-20
    const float emax = fmaxf(fmaxf(fmaxf(fabsf(e1[0]), fabsf(e1[1])), fabsf(e1[2])), fmaxf(fmaxf(fabsf(e2[0]), fabsf(e2[1])), fabsf(e2[2]))) * UP;
    const float ee = (fabsf(e1[0]) + fabsf(e1[1]) + fabsf(e1[2])) * emax * UP;
    const float px = __builtin_fmaf(dy, e2[2], -(dz * e2[1]));
    const float py = __builtin_fmaf(dz, e2[0], -(dx * e2[2]));
    const float pz = __builtin_fmaf(dx, e2[1], -(dy * e2[0]));
    const float det = __builtin_fmaf(e1[0], px, __builtin_fmaf(e1[1], py, e1[2] * pz));
    const float u = __builtin_fmaf(tvx, px, __builtin_fmaf(tvy, py, tvz * pz));
    const float qx = __builtin_fmaf(tvy, e1[2], -(tvz * e1[1]));
    const float qy = __builtin_fmaf(tvz, e1[0], -(tvx * e1[2]));
    const float qz = __builtin_fmaf(tvx, e1[1], -(tvy * e1[0]));
    const float v = __builtin_fmaf(dx, qx, __builtin_fmaf(dy, qy, dz * qz));
    const float tvm = fabsf(tvx) + fabsf(tvy) + fabsf(tvz);
    const float gd = G * dm;
    const float muv = __builtin_fmaf(__builtin_fmaf(gd, tvm, 6.001f * tv_err * dm), emax, 1e-30f);
    const float md = __builtin_fmaf(gd, ee, 1e-30f);
    const float adet = fabsf(det);
    const float su = det < 0.0f ? -u : u;
    const float sv = det < 0.0f ? -v : v;
    const bool certain = adet > md;
    const bool out = (adet < 0.000001f - md) | (su < -muv) | (sv < -muv) | (adet - su - sv < -(md + 2.0f * muv));
    return certain & out;   // true = safe to skip the FP64 test
}

// AABB.Intersect(ref Ray, ref tmin): AABB_Main.cs:173-260.  Moves the origin on success.
HARE_HD bool aabb_clip_move(const double* bmin, const double* bmax, V3& o, const V3& d, double& tmin_out)
{
    double tmin = 0;
    double tmax = 1.7976931348623157e308;
    const double oo[3] = {o.x, o.y, o.z};
    const double dd[3] = {d.x, d.y, d.z};
#if defined(__HIPCC__)
#pragma unroll
#endif
    for (int a = 0; a < 3; ++a) {
        if (fabs(dd[a]) < 4.9406564584124654e-324) {
            if (oo[a] < bmin[a] || oo[a] > bmax[a]) return false;
        } else {
            const double ood = (1 / dd[a]);
            double t1 = (bmin[a] - oo[a]) * ood;
            double t2 = (bmax[a] - oo[a]) * ood;
            if (t1 > t2) {
                const double s = t1;
                t1 = t2;
                t2 = s;
            }
            tmin = net_max(tmin, t1);
            tmax = net_min(tmax, t2);
            if (tmin > tmax) return false;
        }
    }
    o.x = o.x + d.x * tmin;
    o.y = o.y + d.y * tmin;
    o.z = o.z + d.z * tmin;
    tmin_out = tmin;
    return true;
}

// Padded voxel box along one axis: Voxel_Grid.cs:283-285 (+ Point + Point, Primitives.cs:156-159)
HARE_HD double voxel_lo(int i, double vd, double omin) { return (i * vd - 0.001) + omin; }
HARE_HD double voxel_hi(int i, double vd, double omin) { return ((i + 1) * vd + 0.001) + omin; }

// ---- AABB.PolyBoxOverlap: AABB_Tri_Int.cs:165-260 (Akenine-Moller SAT as Hare translated it) ----
// One fan triangle against a box given by centre c and half-width h (AABB ctor, AABB_Main.cs:64-67).
HARE_HD bool tri_box_sat(const double* c, const double* h, const double* A, const double* B, const double* C)
{
    const double v0x = A[0] - c[0], v0y = A[1] - c[1], v0z = A[2] - c[2];
    const double v1x = B[0] - c[0], v1y = B[1] - c[1], v1z = B[2] - c[2];
    const double v2x = C[0] - c[0], v2y = C[1] - c[1], v2z = C[2] - c[2];
    const double e0x = v1x - v0x, e0y = v1y - v0y, e0z = v1z - v0z;
    const double e1x = v2x - v1x, e1y = v2y - v1y, e1z = v2z - v1z;
    const double e2x = v0x - v2x, e2y = v0y - v2y, e2z = v0z - v2z;
    double fex, fey, fez, pa, pb, mn, mx, rad;

#define HARE_AXIS(PA, PB, RAD)                                        \
    pa = (PA);                                                        \
    pb = (PB);                                                        \
    if (pa < pb) { mn = pa; mx = pb; } else { mn = pb; mx = pa; }     \
    rad = (RAD);                                                      \
    if (mn > rad || mx < -rad) return false;
    // Z12 orders its pair with (p2 < p1) instead of (p1 < p2) (AABB_Tri_Int.cs:148); same min/max.
#define HARE_AXIS_Z12(P1, P2, RAD)                                    \
    pa = (P1);                                                        \
    pb = (P2);                                                        \
    if (pb < pa) { mn = pb; mx = pa; } else { mn = pa; mx = pb; }     \
    rad = (RAD);                                                      \
    if (mn > rad || mx < -rad) return false;

    fex = fabs(e0x); fey = fabs(e0y); fez = fabs(e0z);
    HARE_AXIS(e0z * v0y - e0y * v0z, e0z * v2y - e0y * v2z, fez * h[1] + fey * h[2])          // X01
    HARE_AXIS(-e0z * v0x + e0x * v0z, -e0z * v2x + e0x * v2z, fez * h[0] + fex * h[2])        // Y02
    HARE_AXIS_Z12(e0y * v1x - e0x * v1y, e0y * v2x - e0x * v2y, fey * h[0] + fex * h[1])      // Z12
    fex = fabs(e1x); fey = fabs(e1y); fez = fabs(e1z);
    HARE_AXIS(e1z * v0y - e1y * v0z, e1z * v2y - e1y * v2z, fez * h[1] + fey * h[2])          // X01
    HARE_AXIS(-e1z * v0x + e1x * v0z, -e1z * v2x + e1x * v2z, fez * h[0] + fex * h[2])        // Y02
    HARE_AXIS(e1y * v0x - e1x * v0y, e1y * v1x - e1x * v1y, fey * h[0] + fex * h[1])          // Z0
    fex = fabs(e2x); fey = fabs(e2y); fez = fabs(e2z);
    HARE_AXIS(e2z * v0y - e2y * v0z, e2z * v1y - e2y * v1z, fez * h[1] + fey * h[2])          // X2
    HARE_AXIS(-e2z * v0x + e2x * v0z, -e2z * v1x + e2x * v1z, fez * h[0] + fex * h[2])        // Y1
    HARE_AXIS_Z12(e2y * v1x - e2x * v1y, e2y * v2x - e2x * v2y, fey * h[0] + fex * h[1])      // Z12
#undef HARE_AXIS
#undef HARE_AXIS_Z12

    // FINDMINMAX + box-axis tests (AABB_Tri_Int.cs:240-249)
#define HARE_MM(a0, a1, a2, hw)                 \
    mn = a0;                                    \
    mx = a0;                                    \
    if (a1 < mn) mn = a1;                       \
    if (a1 > mx) mx = a1;                       \
    if (a2 < mn) mn = a2;                       \
    if (a2 > mx) mx = a2;                       \
    if (mn > hw || mx < -hw) return false;
    HARE_MM(v0x, v1x, v2x, h[0])
    HARE_MM(v0y, v1y, v2y, h[1])
    HARE_MM(v0z, v1z, v2z, h[2])
#undef HARE_MM

    // planeBoxOverlap(Cross(e0,e1), v0, halfwidth): AABB_Tri_Int.cs:51-95,255-256
    const double nx = e0y * e1z - e0z * e1y;
    const double ny = -(e0x * e1z - e0z * e1x);
    const double nz = e0x * e1y - e0y * e1x;
    double mnx, mny, mnz, mxx, mxy, mxz;
    if (nx > 0.0) { mnx = -h[0] - v0x; mxx = h[0] - v0x; } else { mnx = h[0] - v0x; mxx = -h[0] - v0x; }
    if (ny > 0.0) { mny = -h[1] - v0y; mxy = h[1] - v0y; } else { mny = h[1] - v0y; mxy = -h[1] - v0y; }
    if (nz > 0.0) { mnz = -h[2] - v0z; mxz = h[2] - v0z; } else { mnz = h[2] - v0z; mxz = -h[2] - v0z; }
    if (dot3(nx, ny, nz, mnx, mny, mnz) > 0.0) return false;
    if (dot3(nx, ny, nz, mxx, mxy, mxz) >= 0.0) return true;
    return false;
}

// PolyBoxOverlap on a polygon of nv corners (fan (P0,Pj,Pj+1); true at the first overlapping fan
// triangle).  bmin/bmax are the box corners; centre and half-width as the AABB ctor derives them.
HARE_HD bool poly_box_overlap(const double* bmin, const double* bmax, const double* P /*nv x 3*/, int nv)
{
    double c[3], h[3];
    for (int a = 0; a < 3; ++a) {
        c[a] = (bmax[a] + bmin[a]) / 2;
        const double w = bmax[a] - bmin[a];
        h[a] = w / 2;
    }
    for (int j = 1; j + 1 < nv; ++j)
        if (tri_box_sat(c, h, P, P + 3 * j, P + 3 * (j + 1))) return true;
    return false;
}

}  // namespace hare
