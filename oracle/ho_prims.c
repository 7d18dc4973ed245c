/*
 * ho_prims.c -- oracle primitives: Dot/Cross, polygon normals, ray/polygon tests,
 * AABB slab clip, point-in-box, triangle/box SAT, Topology ingest.
 *
 * TEST INFRASTRUCTURE ONLY (see hare_oracle.h).  PARITY UNPINNED (no reference
 * fixtures exist; the C# cannot be run here).  Citations are file:line into
 * /root/reference/.
 *
 * Build: gcc -O2 -std=c11 -ffp-contract=off -fno-fast-math (no FMA, no
 * reassociation): every expression below is evaluated exactly as written, left
 * to right, in IEEE-754 binary64, like the .NET JIT does on x64.
 */
#include "hare_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>

/* Hare_Geometry_Math.cs:43-46 */
double ho_dot(double ax, double ay, double az, double bx, double by, double bz)
{
    return (ax * bx) + (ay * by) + (az * bz);
}

/* Hare_Geometry_Math.cs:66-69 -- note the y component is written -(ax*bz - az*bx). */
void ho_cross(double ax, double ay, double az, double bx, double by, double bz, double out[3])
{
    out[0] = ay * bz - az * by;
    out[1] = -(ax * bz - az * bx);
    out[2] = ax * by - ay * bx;
}

/* System.Math.Max(double,double) as implemented by .NET (net7.0, the first TFM of
 * Hare.csproj:3): NaN propagates, +0 beats -0.  Used by AABB_Main.cs:198 and
 * "Octree - alt.cs":182. */
double ho_dotnet_max(double a, double b)
{
    if (a != b) {
        if (!isnan(a)) return b < a ? a : b;
        return a;
    }
    return signbit(b) ? a : b;
}

/* System.Math.Min(double,double), .NET semantics (AABB_Main.cs:199). */
double ho_dotnet_min(double a, double b)
{
    if (a != b) {
        if (!isnan(a)) return a < b ? a : b;
        return a;
    }
    return signbit(a) ? a : b;
}

/* System.Math.Round(double, int) with the default MidpointRounding.ToEven, as .NET
 * computes it: scale by 10^digits, round-half-even, unscale (Hare_Geometry_Primitives.cs:230-235
 * calls it with Prec = 15, Hare_Geometry_Topology.cs:70,345). */
double ho_dotnet_round(double x, int digits)
{
    static const double p10[16] = {1e0, 1e1, 1e2, 1e3, 1e4, 1e5, 1e6, 1e7, 1e8,
                                   1e9, 1e10, 1e11, 1e12, 1e13, 1e14, 1e15};
    if (fabs(x) < 1e16) {
        double p = p10[digits];
        x = x * p;
        x = rint(x);
        x = x / p;
    }
    return x;
}

/* Polygon ctor normal (Hare_Geometry_Polygons.cs:159-171): first non-zero
 * Cross(V1-V0, Vj-V0), j >= 2, then Vector.Normalize (Primitives.cs:49-57: three
 * divisions by sqrt(f), nothing when f == 0). */
void ho_polygon_normals(const double *verts, const int32_t *nverts, int32_t P, double *normals_out)
{
    for (int32_t p = 0; p < P; ++p) {
        const double *V = verts + (size_t)p * 12;
        double n[3] = {0, 0, 0};
        for (int j = 2; j < nverts[p]; ++j) {
            double ax = V[3] - V[0], ay = V[4] - V[1], az = V[5] - V[2];
            double bx = V[3 * j] - V[0], by = V[3 * j + 1] - V[1], bz = V[3 * j + 2] - V[2];
            ho_cross(ax, ay, az, bx, by, bz, n);
            /* IsZeroVector: (dx*dx + dy*dy + dz*dz) < double.Epsilon  (Primitives.cs:121-125) */
            if (!((n[0] * n[0] + n[1] * n[1] + n[2] * n[2]) < 4.9406564584124654e-324)) break;
        }
        double f = n[0] * n[0] + n[1] * n[1] + n[2] * n[2];
        if (f != 0) {
            f = sqrt(f);
            n[0] /= f;
            n[1] /= f;
            n[2] /= f;
        }
        normals_out[3 * p + 0] = n[0];
        normals_out[3 * p + 1] = n[1];
        normals_out[3 * p + 2] = n[2];
    }
}

/* Finish_Topology bounds (Hare_Geometry_Topology.cs:148-167): vertex bounds -/+ 1e-12.
 * Every vertex of Vertices_List is a corner of some polygon after Build_Topology, so the
 * corner bounds are the vertex-list bounds. */
void ho_finish_topology_bounds(const double *verts, const int32_t *nverts, int32_t P, double mn[3], double mx[3])
{
    double lo[3] = {DBL_MAX, DBL_MAX, DBL_MAX}, hi[3] = {-DBL_MAX, -DBL_MAX, -DBL_MAX};
    for (int32_t p = 0; p < P; ++p)
        for (int c = 0; c < nverts[p]; ++c)
            for (int a = 0; a < 3; ++a) {
                double v = verts[(size_t)p * 12 + 3 * c + a];
                if (lo[a] > v) lo[a] = v;
                if (hi[a] < v) hi[a] = v;
            }
    for (int a = 0; a < 3; ++a) {
        mn[a] = lo[a] - 0.000000000001;
        mx[a] = hi[a] + 0.000000000001;
    }
}

/* Topology.Polygon_Centroid (Hare_Geometry_Topology.cs:566-574): running Point sum from
 * (0,0,0), then Point / VertexCount. */
void ho_polygon_centroids(const double *verts, const int32_t *nverts, int32_t P, double *cent)
{
    for (int32_t p = 0; p < P; ++p) {
        double s[3] = {0, 0, 0};
        for (int c = 0; c < nverts[p]; ++c)
            for (int a = 0; a < 3; ++a) s[a] = s[a] + verts[(size_t)p * 12 + 3 * c + a];
        for (int a = 0; a < 3; ++a) cent[3 * p + a] = s[a] / (double)nverts[p];
    }
}

/* ---- Topology(Point[][]) ingest: Hare_Geometry_Topology.cs:121-142, :258-311, :342-377;
 *      Point.Hash2 Hare_Geometry_Primitives.cs:237-250; MS_AABB Topology.cs:677-697 ---- */
typedef struct vkey { uint64_t bucket, pos; int32_t index; } vkey;

static uint64_t hmix(uint64_t a, uint64_t b)
{
    uint64_t h = a * 0x9E3779B97F4A7C15ull ^ (b + 0x7F4A7C15ull + (a << 6) + (a >> 2));
    h ^= h >> 29;
    h *= 0xBF58476D1CE4E5B9ull;
    h ^= h >> 32;
    return h;
}

int32_t ho_build_topology(const double *vin, const int32_t *nverts, int32_t P, double *vout)
{
    /* bounds of the raw input points -/+ 1e-12 => Modspace (Topology.cs:124-139) */
    double lo[3] = {DBL_MAX, DBL_MAX, DBL_MAX}, hi[3] = {-DBL_MAX, -DBL_MAX, -DBL_MAX};
    for (int32_t p = 0; p < P; ++p)
        for (int c = 0; c < nverts[p]; ++c)
            for (int a = 0; a < 3; ++a) {
                double v = vin[(size_t)p * 12 + 3 * c + a];
                if (lo[a] > v) lo[a] = v;
                if (hi[a] < v) hi[a] = v;
            }
    double mmin[3], mmax[3];
    for (int a = 0; a < 3; ++a) {
        mmin[a] = lo[a] - 0.000000000001;
        mmax[a] = hi[a] + 0.000000000001;
    }
    double xl = mmax[0] - mmin[0], yl = mmax[1] - mmin[1], zl = mmax[2] - mmin[2];
    int cx = (int)ceil(xl), cy = (int)ceil(yl), cz = (int)ceil(zl);
    int mxd = cx > (cy > cz ? cy : cz) ? cx : (cy > cz ? cy : cz);
    uint64_t ydim = (uint64_t)mxd, XYTot = (uint64_t)mxd * (uint64_t)mxd;

    size_t total = 0;
    for (int32_t p = 0; p < P; ++p) total += (size_t)nverts[p];
    size_t cap = 16;
    while (cap < total * 2) cap <<= 1;
    vkey *tab = (vkey *)malloc(cap * sizeof(vkey));
    double *vlist = (double *)malloc((total ? total : 1) * 3 * sizeof(double));
    for (size_t i = 0; i < cap; ++i) tab[i].index = -1;
    int32_t nv = 0;

    for (int32_t p = 0; p < P; ++p) {
        for (int c = 0; c < 4; ++c)
            for (int a = 0; a < 3; ++a) vout[(size_t)p * 12 + 3 * c + a] = 0.0;
        for (int c = 0; c < nverts[p]; ++c) {
            /* AddGetIndex: x.Round(Prec) then Hash2 */
            double x = ho_dotnet_round(vin[(size_t)p * 12 + 3 * c + 0], 15);
            double y = ho_dotnet_round(vin[(size_t)p * 12 + 3 * c + 1], 15);
            double z = ho_dotnet_round(vin[(size_t)p * 12 + 3 * c + 2], 15);
            double Xoff = x - mmin[0], Yoff = y - mmin[1], Zoff = z - mmin[2];
            uint64_t xloc = (uint64_t)floor(Xoff), yloc = (uint64_t)floor(Yoff), zloc = (uint64_t)floor(Zoff);
            uint64_t bucket = XYTot * zloc + ydim * xloc + yloc;
            uint64_t xpos = (uint64_t)((Xoff - (double)xloc) * 1000);
            uint64_t ypos = (uint64_t)((Yoff - (double)yloc) * 1000);
            uint64_t zpos = (uint64_t)((Zoff - (double)zloc) * 1000);
            uint64_t pos = 1000000 * zpos + 1000 * xpos + ypos;
            size_t h = (size_t)hmix(bucket, pos) & (cap - 1);
            int32_t found = -1;
            while (tab[h].index >= 0) {
                if (tab[h].bucket == bucket && tab[h].pos == pos) { found = tab[h].index; break; }
                h = (h + 1) & (cap - 1);
            }
            if (found < 0) {
                tab[h].bucket = bucket;
                tab[h].pos = pos;
                tab[h].index = nv;
                vlist[3 * (size_t)nv + 0] = x;
                vlist[3 * (size_t)nv + 1] = y;
                vlist[3 * (size_t)nv + 2] = z;
                found = nv++;
            }
            for (int a = 0; a < 3; ++a) vout[(size_t)p * 12 + 3 * c + a] = vlist[3 * (size_t)found + a];
        }
    }
    free(tab);
    free(vlist);
    return nv;
}

/* ---- RayXtri, fast ("High Performance") variant: Hare_Geometry_Polygons.cs:449-510 ---- */
static int ray_x_tri_fast(const ho_ray *R, const double *v0, const double *v1, const double *v2, double *t)
{
    double edge1x = v1[0] - v0[0];
    double edge1y = v1[1] - v0[1];
    double edge1z = v1[2] - v0[2];
    double edge2x = v2[0] - v0[0];
    double edge2y = v2[1] - v0[1];
    double edge2z = v2[2] - v0[2];
    double u, v;

    double pvecx = R->dy * edge2z - R->dz * edge2y;
    double pvecy = R->dz * edge2x - R->dx * edge2z;
    double pvecz = R->dx * edge2y - R->dy * edge2x;

    double det = ho_dot(edge1x, edge1y, edge1z, pvecx, pvecy, pvecz);

    double tvecx = R->x - v0[0];
    double tvecy = R->y - v0[1];
    double tvecz = R->z - v0[2];
    double invdet = 1.0 / det;

    double qvecx = tvecy * edge1z - tvecz * edge1y;
    double qvecy = tvecz * edge1x - tvecx * edge1z;
    double qvecz = tvecx * edge1y - tvecy * edge1x;

    if (det > 0.000001) {
        u = ho_dot(tvecx, tvecy, tvecz, pvecx, pvecy, pvecz);
        if (u < 0.0 || u > det) return 0;
        v = ho_dot(R->dx, R->dy, R->dz, qvecx, qvecy, qvecz);
        if (v < 0.0 || u + v > det) return 0;
    } else if (det < -0.000001) {
        u = ho_dot(tvecx, tvecy, tvecz, pvecx, pvecy, pvecz);
        if (u > 0.0 || u < det) return 0;
        v = ho_dot(R->dx, R->dy, R->dz, qvecx, qvecy, qvecz);
        if (v > 0.0 || u + v < det) return 0;
    } else
        return 0;

    *t = ho_dot(edge2x, edge2y, edge2z, qvecx, qvecy, qvecz) * invdet;
    return 1;
}

/* ---- RayXtri, full variant with u,v: Hare_Geometry_Polygons.cs:385-435.  u and v are
 *      `ref` parameters: they are overwritten even on the failing paths. ---- */
static int ray_x_tri_full(const ho_ray *R, const double *v0, const double *v1, const double *v2,
                          double *t, double *u, double *v)
{
    double e1[3] = {v1[0] - v0[0], v1[1] - v0[1], v1[2] - v0[2]};
    double e2[3] = {v2[0] - v0[0], v2[1] - v0[1], v2[2] - v0[2]};
    double pvec[3], qvec[3];
    ho_cross(R->dx, R->dy, R->dz, e2[0], e2[1], e2[2], pvec);
    double det = ho_dot(e1[0], e1[1], e1[2], pvec[0], pvec[1], pvec[2]);
    double tvecx = R->x - v0[0];
    double tvecy = R->y - v0[1];
    double tvecz = R->z - v0[2];
    double invdet = 1.0 / det;
    ho_cross(tvecx, tvecy, tvecz, e1[0], e1[1], e1[2], qvec);

    if (det > 0.000001) {
        *u = ho_dot(tvecx, tvecy, tvecz, pvec[0], pvec[1], pvec[2]);
        if (*u < 0.0 || *u > det) return 0;
        *v = ho_dot(R->dx, R->dy, R->dz, qvec[0], qvec[1], qvec[2]);
        if (*v < 0.0 || *u + *v > det) return 0;
    } else if (det < -0.000001) {
        *u = ho_dot(tvecx, tvecy, tvecz, pvec[0], pvec[1], pvec[2]);
        if (*u > 0.0 || *u < det) return 0;
        *v = ho_dot(R->dx, R->dy, R->dz, qvec[0], qvec[1], qvec[2]);
        if (*v > 0.0 || *u + *v < det) return 0;
    } else
        return 0;

    *t = ho_dot(e2[0], e2[1], e2[2], qvec[0], qvec[1], qvec[2]) * invdet;
    *u = *u * invdet;
    *v = *v * invdet;
    return 1;
}

/* Ray_Side (Hare_Geometry_Polygons.cs:601-606): n < 0 -> false, otherwise (incl. NaN) true. */
static int ray_side(const ho_topology *T, int32_t i, const ho_ray *R)
{
    const double *N = T->normals + 3 * (size_t)i;
    double n = ho_dot(R->dx, R->dy, R->dz, N[0], N[1], N[2]);
    if (n < 0) return 0;
    return 1;
}

/* Topology.intersect fast (Hare_Geometry_Topology.cs:459-462) ->
 * Triangle.Intersect (Polygons.cs:637-660) / Quadrilateral.Intersect (Polygons.cs:784-823). */
int ho_poly_intersect_fast(const ho_topology *T, int32_t i, const ho_ray *R, double *x, double *y, double *z, double *t)
{
    const double *P = T->verts + (size_t)i * 12;
    const double *P0 = P, *P1 = P + 3, *P2 = P + 6, *P3 = P + 9;
    int ok;
    *t = 0;
    if (T->nverts[i] == 3) {
        if (ray_side(T, i, R))
            ok = ray_x_tri_fast(R, P0, P1, P2, t);
        else
            ok = ray_x_tri_fast(R, P2, P1, P0, t);
    } else {
        if (ray_side(T, i, R))
            ok = ray_x_tri_fast(R, P0, P1, P2, t) || ray_x_tri_fast(R, P2, P3, P0, t);
        else
            ok = ray_x_tri_fast(R, P2, P1, P0, t) || ray_x_tri_fast(R, P0, P3, P2, t);
    }
    if (ok) {
        *x = R->x + R->dx * *t;
        *y = R->y + R->dy * *t;
        *z = R->z + R->dz * *t;
        return 1;
    }
    *x = 0;
    *y = 0;
    *z = 0;
    return 0;
}

/* Topology.intersect full (Hare_Geometry_Topology.cs:450-453) ->
 * Triangle.Intersect (Polygons.cs:662-688) / Quadrilateral.Intersect (Polygons.cs:731-782). */
int ho_poly_intersect_full(const ho_topology *T, int32_t i, const ho_ray *R, double *x, double *y, double *z,
                           double *u, double *v, double *t)
{
    const double *P = T->verts + (size_t)i * 12;
    const double *P0 = P, *P1 = P + 3, *P2 = P + 6, *P3 = P + 9;
    int ok;
    *u = 0;
    *v = 0;
    *t = 0;
    if (T->nverts[i] == 3) {
        if (ray_side(T, i, R))
            ok = ray_x_tri_full(R, P0, P1, P2, t, u, v);
        else
            ok = ray_x_tri_full(R, P2, P1, P0, t, u, v);
    } else {
        if (ray_side(T, i, R))
            ok = ray_x_tri_full(R, P0, P1, P2, t, u, v) || ray_x_tri_full(R, P2, P3, P0, t, u, v);
        else
            ok = ray_x_tri_full(R, P2, P1, P0, t, u, v) || ray_x_tri_full(R, P0, P3, P2, t, u, v);
    }
    if (ok) {
        *x = R->x + R->dx * *t;
        *y = R->y + R->dy * *t;
        *z = R->z + R->dz * *t;
        return 1;
    }
    *x = 0;
    *y = 0;
    *z = 0;
    return 0;
}

/* AABB.Intersect(ref Ray R, ref double tmin): AABB_Main.cs:173-260.  Moves R's origin. */
int ho_aabb_intersect_move(const double bmin[3], const double bmax[3], ho_ray *R, double *tmin_out)
{
    double tmin = 0;
    double tmax = DBL_MAX;
    const double o[3] = {R->x, R->y, R->z};
    const double d[3] = {R->dx, R->dy, R->dz};
    for (int a = 0; a < 3; ++a) {
        if (fabs(d[a]) < 4.9406564584124654e-324) { /* double.Epsilon */
            if (o[a] < bmin[a] || o[a] > bmax[a]) { *tmin_out = tmin; return 0; }
        } else {
            double ood = (1 / d[a]);
            double t1 = (bmin[a] - o[a]) * ood;
            double t2 = (bmax[a] - o[a]) * ood;
            if (t1 > t2) {
                double tswap = t1;
                t1 = t2;
                t2 = tswap;
            }
            tmin = ho_dotnet_max(tmin, t1);
            tmax = ho_dotnet_min(tmax, t2);
            if (tmin > tmax) { *tmin_out = tmin; return 0; }
        }
    }
    R->x = R->x + R->dx * tmin;
    R->y = R->y + R->dy * tmin;
    R->z = R->z + R->dz * tmin;
    *tmin_out = tmin;
    return 1;
}

/* AABB.IsPointInBox: AABB_Main.cs:75-84 */
int ho_is_point_in_box(const double bmin[3], const double bmax[3], double x, double y, double z)
{
    if (x < bmin[0]) return 0;
    if (y < bmin[1]) return 0;
    if (z < bmin[2]) return 0;
    if (x > bmax[0]) return 0;
    if (y > bmax[1]) return 0;
    if (z > bmax[2]) return 0;
    return 1;
}

/* ---- AABB.PolyBoxOverlap: AABB_Tri_Int.cs:165-260 (axis tests :101-162, planeBoxOverlap :51-95).
 *      Box-derived quantities from the AABB ctor, AABB_Main.cs:57-68. ---- */
typedef struct sat_ctx {
    double v0[3], v1[3], v2[3];
    double h[3];
    double mn, mx, rad, p0, p1, p2;
} sat_ctx;

static int ax_x01(sat_ctx *s, double a, double b, double fa, double fb)
{
    s->p0 = a * s->v0[1] - b * s->v0[2];
    s->p2 = a * s->v2[1] - b * s->v2[2];
    if (s->p0 < s->p2) { s->mn = s->p0; s->mx = s->p2; } else { s->mn = s->p2; s->mx = s->p0; }
    s->rad = fa * s->h[1] + fb * s->h[2];
    if (s->mn > s->rad || s->mx < -s->rad) return 0;
    return 1;
}
static int ax_x2(sat_ctx *s, double a, double b, double fa, double fb)
{
    s->p0 = a * s->v0[1] - b * s->v0[2];
    s->p1 = a * s->v1[1] - b * s->v1[2];
    if (s->p0 < s->p1) { s->mn = s->p0; s->mx = s->p1; } else { s->mn = s->p1; s->mx = s->p0; }
    s->rad = fa * s->h[1] + fb * s->h[2];
    if (s->mn > s->rad || s->mx < -s->rad) return 0;
    return 1;
}
static int ax_y02(sat_ctx *s, double a, double b, double fa, double fb)
{
    s->p0 = -a * s->v0[0] + b * s->v0[2];
    s->p2 = -a * s->v2[0] + b * s->v2[2];
    if (s->p0 < s->p2) { s->mn = s->p0; s->mx = s->p2; } else { s->mn = s->p2; s->mx = s->p0; }
    s->rad = fa * s->h[0] + fb * s->h[2];
    if (s->mn > s->rad || s->mx < -s->rad) return 0;
    return 1;
}
static int ax_y1(sat_ctx *s, double a, double b, double fa, double fb)
{
    s->p0 = -a * s->v0[0] + b * s->v0[2];
    s->p1 = -a * s->v1[0] + b * s->v1[2];
    if (s->p0 < s->p1) { s->mn = s->p0; s->mx = s->p1; } else { s->mn = s->p1; s->mx = s->p0; }
    s->rad = fa * s->h[0] + fb * s->h[2];
    if (s->mn > s->rad || s->mx < -s->rad) return 0;
    return 1;
}
static int ax_z12(sat_ctx *s, double a, double b, double fa, double fb)
{
    s->p1 = a * s->v1[0] - b * s->v1[1];
    s->p2 = a * s->v2[0] - b * s->v2[1];
    if (s->p2 < s->p1) { s->mn = s->p2; s->mx = s->p1; } else { s->mn = s->p1; s->mx = s->p2; }
    s->rad = fa * s->h[0] + fb * s->h[1];
    if (s->mn > s->rad || s->mx < -s->rad) return 0;
    return 1;
}
static int ax_z0(sat_ctx *s, double a, double b, double fa, double fb)
{
    s->p0 = a * s->v0[0] - b * s->v0[1];
    s->p1 = a * s->v1[0] - b * s->v1[1];
    if (s->p0 < s->p1) { s->mn = s->p0; s->mx = s->p1; } else { s->mn = s->p1; s->mx = s->p0; }
    s->rad = fa * s->h[0] + fb * s->h[1];
    if (s->mn > s->rad || s->mx < -s->rad) return 0;
    return 1;
}

static void findminmax(double x0, double x1, double x2, double *mn, double *mx)
{
    *mn = x0;
    *mx = x0;
    if (x1 < *mn) *mn = x1;
    if (x1 > *mx) *mx = x1;
    if (x2 < *mn) *mn = x2;
    if (x2 > *mx) *mx = x2;
}

static int plane_box_overlap(const double n[3], const double vert[3], const double maxbox[3])
{
    double vmin[3], vmax[3];
    for (int a = 0; a < 3; ++a) {
        double v = vert[a];
        if (n[a] > 0.0) {
            vmin[a] = -maxbox[a] - v;
            vmax[a] = maxbox[a] - v;
        } else {
            vmin[a] = maxbox[a] - v;
            vmax[a] = -maxbox[a] - v;
        }
    }
    if (ho_dot(n[0], n[1], n[2], vmin[0], vmin[1], vmin[2]) > 0.0) return 0;
    if (ho_dot(n[0], n[1], n[2], vmax[0], vmax[1], vmax[2]) >= 0.0) return 1;
    return 0;
}

int ho_poly_box_overlap(const double bmin[3], const double bmax[3], const double *P, int32_t nv)
{
    sat_ctx s;
    double center[3];
    for (int a = 0; a < 3; ++a) {
        center[a] = (bmax[a] + bmin[a]) / 2;   /* Center = (Max + Min) / 2 */
        double width = bmax[a] - bmin[a];      /* Width = Max - Min        */
        s.h[a] = width / 2;                    /* halfwidth = Width / 2    */
    }
    /* fan (P0, Pj, Pj+1), j = 1..n-2 (AABB_Tri_Int.cs:167-172) */
    for (int j = 1, k = 2; k < nv; ++j, ++k) {
        const double *t0 = P, *t1 = P + 3 * j, *t2 = P + 3 * k;
        for (int a = 0; a < 3; ++a) {
            s.v0[a] = t0[a] - center[a];
            s.v1[a] = t1[a] - center[a];
            s.v2[a] = t2[a] - center[a];
        }
        double e0[3], e1[3], e2[3];
        for (int a = 0; a < 3; ++a) {
            e0[a] = s.v1[a] - s.v0[a];
            e1[a] = s.v2[a] - s.v1[a];
            e2[a] = s.v0[a] - s.v2[a];
        }
        double fex, fey, fez;
        fex = fabs(e0[0]);
        fey = fabs(e0[1]);
        fez = fabs(e0[2]);
        if (!ax_x01(&s, e0[2], e0[1], fez, fey)) continue;
        if (!ax_y02(&s, e0[2], e0[0], fez, fex)) continue;
        if (!ax_z12(&s, e0[1], e0[0], fey, fex)) continue;

        fex = fabs(e1[0]);
        fey = fabs(e1[1]);
        fez = fabs(e1[2]);
        if (!ax_x01(&s, e1[2], e1[1], fez, fey)) continue;
        if (!ax_y02(&s, e1[2], e1[0], fez, fex)) continue;
        if (!ax_z0(&s, e1[1], e1[0], fey, fex)) continue;

        fex = fabs(e2[0]);
        fey = fabs(e2[1]);
        fez = fabs(e2[2]);
        if (!ax_x2(&s, e2[2], e2[1], fez, fey)) continue;
        if (!ax_y1(&s, e2[2], e2[0], fez, fex)) continue;
        if (!ax_z12(&s, e2[1], e2[0], fey, fex)) continue;

        findminmax(s.v0[0], s.v1[0], s.v2[0], &s.mn, &s.mx);
        if (s.mn > s.h[0] || s.mx < -s.h[0]) continue;
        findminmax(s.v0[1], s.v1[1], s.v2[1], &s.mn, &s.mx);
        if (s.mn > s.h[1] || s.mx < -s.h[1]) continue;
        findminmax(s.v0[2], s.v1[2], s.v2[2], &s.mn, &s.mx);
        if (s.mn > s.h[2] || s.mx < -s.h[2]) continue;

        double normal[3];
        ho_cross(e0[0], e0[1], e0[2], e1[0], e1[1], e1[2], normal);
        if (!plane_box_overlap(normal, s.v0, s.h)) continue;
        return 1;
    }
    return 0;
}

/* Brute force nearest hit over all polygons, ascending index, strict '<', t > 1e-10.
 * NOT in the reference: used only to cross-check the partitions. */
int ho_brute_shoot(const ho_topology *T, const ho_ray *R, int32_t po1, int32_t po2, int full_uv, ho_xevent *out)
{
    double tmin = DBL_MAX;
    ho_xevent best;
    memset(&best, 0, sizeof best);
    best.poly_id = -1;
    for (int32_t i = 0; i < T->P; ++i) {
        if (i == po1 || i == po2) continue;
        double x, y, z, t, u = 0, v = 0;
        int ok = full_uv ? ho_poly_intersect_full(T, i, R, &x, &y, &z, &u, &v, &t)
                         : ho_poly_intersect_fast(T, i, R, &x, &y, &z, &t);
        if (ok && t > 0.0000000001 && t < tmin) {
            tmin = t;
            best.t = t;
            best.u = u;
            best.v = v;
            best.x = x;
            best.y = y;
            best.z = z;
            best.poly_id = i;
            best.hit = 1;
        }
    }
    *out = best;
    return best.hit;
}

/* Harness-defined specular bounce (no reference code):
 * o' = X_Point, d' = d - (2*(d.n))*n with n = Polys[Poly_id].Normal. */
void ho_reflect(const ho_topology *T, const ho_ray *R, const ho_xevent *ev, ho_ray *out)
{
    const double *N = T->normals + 3 * (size_t)ev->poly_id;
    double dn = ho_dot(R->dx, R->dy, R->dz, N[0], N[1], N[2]);
    double k = 2.0 * dn;
    out->x = ev->x;
    out->y = ev->y;
    out->z = ev->z;
    out->dx = R->dx - k * N[0];
    out->dy = R->dy - k * N[1];
    out->dz = R->dz - k * N[2];
}

/* The same bounce over a batch (harness-defined): rays that hit are reflected,
 * rays that missed keep their record (the caller retires them). */
void ho_reflect_batch(const ho_topology *T, int64_t n, const ho_ray *rays, const ho_xevent *ev, ho_ray *out)
{
    for (int64_t i = 0; i < n; ++i) {
        if (ev[i].hit) ho_reflect(T, &rays[i], &ev[i], &out[i]);
        else out[i] = rays[i];
    }
}


/* ---- allocation failures and budgets (hare_oracle.h) ---- */
static __thread char ho_err[256];
const char *ho_last_error(void) { return ho_err; }
void ho_set_error(const char *msg)
{
    strncpy(ho_err, msg ? msg : "", sizeof ho_err - 1);
    ho_err[sizeof ho_err - 1] = 0;
}
/* Test hooks (tests/test_tree_budget.py): the checker's own failure paths must be reachable without exhausting the machine.
 *   ho_test_fail_alloc_after(k)  the k-th following ho_grow / ho_alloc of ANY thread fails (k = 0: the next one); negative: off
 *   ho_test_stack_cap(c)         the traversal stacks of Octree.Shoot / KDTree.Shoot start with c entries (0: their usual size), so
 *                                that the growth path runs on an ordinary tree */
static volatile int ho_fail_countdown = -1;
static volatile int ho_stack_cap_override = 0;
void ho_test_fail_alloc_after(int k) { ho_fail_countdown = k; }
void ho_test_stack_cap(int c) { ho_stack_cap_override = c; }
int ho_initial_stack_cap(int usual) { return ho_stack_cap_override > 0 ? ho_stack_cap_override : usual; }
static int ho_alloc_must_fail(void)
{
    if (ho_fail_countdown < 0) return 0;
    return __sync_fetch_and_sub(&ho_fail_countdown, 1) == 0;
}
int ho_grow(void **p, size_t bytes)
{
    void *q = ho_alloc_must_fail() ? NULL : realloc(*p, bytes ? bytes : 1);
    if (!q) {
        ho_set_error("oracle: out of memory");
        return -1;
    }
    *p = q;
    return 0;
}
void *ho_alloc(size_t bytes)
{
    void *q = ho_alloc_must_fail() ? NULL : malloc(bytes ? bytes : 1);
    if (!q) ho_set_error("oracle: out of memory");
    return q;
}
