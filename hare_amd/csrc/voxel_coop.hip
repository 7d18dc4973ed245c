// voxel_coop.hip -- the cooperative tail of the voxel kernels (included by kernels.hip, in front of voxel_pool.hip).
//
// Why.  The cost of a ray is heavy-tailed: a ray that skims a tessellated surface crosses a hundred occupied voxels and
// scans thousands of list entries where the average ray scans 76 (late casts of the bounce loop, config 5).  A lane carries
// such a ray as ONE chain of dependent steps -- voxel, list entries eight at a time, exact test, next voxel -- and the launch
// ends when the longest chain does: in a late bounce cast 90 % of the waves had finished at 60 % of the kernel's duration
// and nine waves ran to the end (tools/c5_timeline.py: 3.83 ms, of which the last 1.5 ms served ~100 rays).
//
// What.  A wave that has drawn its last ticket and is down to its last few rays stops running them as lanes of a pool and
// traces them one after the other with ALL 64 lanes (coop_trace): the lanes take one list entry each, so a voxel's whole
// candidate list costs one pre-cull and one exact test in time instead of one per eight entries.  The chain of a heavy ray
// gets ~3x shorter; nothing leaves the wave, so there is no protocol between waves and nothing to wait for.
// (Handing the rays to OTHER, idle waves through a queue in device memory was built first, bit-exact, and removed: thousands of
// idle waves polling the queue cost the working ones more memory bandwidth than the help was worth, and the chain of one ray
// cannot be split anyway -- DESIGN.md section 9.)
//
// Per ray the candidates, their order and the arithmetic are Voxel_Grid.Shoot's (Voxel_Grid.cs:561-761): a voxel's list is
// scanned in chunks of 64 in list order; within a chunk the smallest t wins and equal t go to the earlier entry, which is what
// the reference's sequential `t < tmin` scan gives; the pending-hit rule (:705) and the miss on grid exit are applied voxel by
// voxel.  Entries already tested before the switch are tested again (a re-test never changes a result: strict `<`).
namespace {


// K1q's packed voxel word (voxel_pool.hip keeps the same constants)
constexpr uint32_t kTF_NX = 1u << 27, kTF_NY = 1u << 28, kTF_NZ = 1u << 29, kTF_MOVED = 1u << 30, kTF_HIT = 1u << 31;

// Voxel_Grid.Shoot from the state in (xf, tMax*, tmin, pid) to the end, by one whole wave.  Returns true on a hit; tmin / pid
// are the returned hit (t measured from the origin the walk uses: the moved one when kTF_MOVED).  Every lane returns the same.
template <bool QUADS, bool COARSE>
__device__ __forceinline__ bool coop_trace(const VoxelArgs& g, const ShootIO& io, const uint32_t* locc, unsigned ray, uint32_t xf,
                                           double tMaxX, double tMaxY, double tMaxZ, double& tmin, int& pid, OwnWork* own = nullptr)
{
    const unsigned lane = threadIdx.x & 63u;
    const int ct = g.ct;
    const RayRec r = io.rays[ray];                       // one address for the wave: a broadcast load
    V3 o = {r.x, r.y, r.z};
    const V3 d = {r.dx, r.dy, r.dz};
    if (xf & kTF_MOVED) {                                // AABB.Intersect's move, the set-up's own expression (AABB_Main.cs:254-256): same bits
        double ts;
        (void)aabb_clip_move(g.omin, g.omax, o, d, ts);
    }
    const double tDeltaX = (xf & kTF_NX) ? g.vd[0] / d.x * -1.0 : g.vd[0] / d.x * 1.0;      // Voxel_Grid.cs:589-632
    const double tDeltaY = (xf & kTF_NY) ? g.vd[1] / d.y * -1.0 : g.vd[1] / d.y * 1.0;
    const double tDeltaZ = (xf & kTF_NZ) ? g.vd[2] / d.z * -1.0 : g.vd[2] / d.z * 1.0;
    const int dx1 = (xf & kTF_NX) ? -1 : 1, dy1 = (xf & kTF_NY) ? -1 : 1, dz1 = (xf & kTF_NZ) ? -1 : 1;
    int X = (int)(xf & 511u), Y = (int)((xf >> 9) & 511u), Z = (int)((xf >> 18) & 511u);
    const int e1 = io.excl1 ? io.excl1[ray] : -1, e2 = io.excl2 ? io.excl2[ray] : -1;      // Voxel_Grid.cs:477
    const CullRay cray = cull_ray(g, o.x, o.y, o.z, d.x, d.y, d.z);
    double hx = 0, hy = 0, hz = 0;
    if (pid >= 0) { hx = o.x + d.x * tmin; hy = o.y + d.y * tmin; hz = o.z + d.z * tmin; }  // Polygons.cs:652
    // bounded by the grid: every iteration moves on by one voxel
    for (int steps = 0; steps < 3 * 512 + 8; ++steps) {
        const int cell = (X * ct + Y) * ct + Z;
        const uint32_t bit = COARSE ? (uint32_t)(((X >> g.occ_shift) * g.occ_cd + (Y >> g.occ_shift)) * g.occ_cd + (Z >> g.occ_shift)) : (uint32_t)cell;
        if ((locc[bit >> 5] >> (bit & 31)) & 1u) {
            const CellRec c = g.cells[cell];
            for (unsigned base = 0; base < c.count; base += 64u) {
                const unsigned k = base + lane;
                const bool valid = k < c.count;
                int i = -1;
                if (valid) i = k == 0 ? c.i0 : (k == 1 ? c.i1 : g.items[c.start + k]);
                bool test = valid && i != e1 && i != e2;
                if (own) { own->entries += valid ? 1u : 0u; own->culls += test ? 1u : 0u; }
                if (test) test = !cull_test<QUADS ? 1 : 0>(g, cray, cull_load<QUADS ? 1 : 0>(g, i));
                double t = kDblMax;
                if (own) own->tests += test ? 1u : 0u;
                if (test) {
                    const PolyRec& p = g.polys[i];
                    const double v0[3] = {p.v0[0], p.v0[1], p.v0[2]}, v1[3] = {p.v1[0], p.v1[1], p.v1[2]};
                    const double v2[3] = {p.v2[0], p.v2[1], p.v2[2]}, nn[3] = {p.n[0], p.n[1], p.n[2]};
                    const bool side = ray_side(d, nn);                          // Polygons.cs:641-648
                    double a[3], cc[3];
#pragma unroll
                    for (int m = 0; m < 3; ++m) { a[m] = side ? v0[m] : v2[m]; cc[m] = side ? v2[m] : v0[m]; }
                    double tt = 0;
                    bool ok = tri_fast(o, d, a, v1, cc, tt);
                    if (QUADS) {
                        if (!ok && g.quads) {
                            const QuadRec& qr = g.quads[i];
                            if (qr.nverts == 4) {
                                const double v3[3] = {qr.v3[0], qr.v3[1], qr.v3[2]};
                                ok = tri_fast(o, d, cc, v3, a, tt);             // (P2,P3,P0) / (P0,P3,P2)
                            }
                        }
                    }
                    if (ok && tt > kTMin) t = tt;                               // Voxel_Grid.cs:691
                }
                // the chunk's best: smallest t, the earlier list entry on a tie (what the sequential strict '<' scan keeps)
                double bt = t;
                unsigned bk = k;
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) {
                    const double ot = __shfl_xor(bt, off, 64);
                    const unsigned ok2 = (unsigned)__shfl_xor((int)bk, off, 64);
                    const bool take = ot < bt || (ot == bt && ok2 < bk);
                    bt = take ? ot : bt;
                    bk = take ? ok2 : bk;
                }
                if (bt < tmin) {                                                // :693 (strict: a hit held from an earlier voxel or chunk stays on a tie)
                    tmin = bt;
                    pid = __shfl(i, (int)(bk - base), 64);
                    hx = o.x + d.x * tmin; hy = o.y + d.y * tmin; hz = o.z + d.z * tmin;
                }
            }
        }
        // Voxel_Grid.cs:705: the hit point inside the CURRENT padded voxel?
        if (pid >= 0) {
            const double lox = voxel_lo(X, g.vd[0], g.omin[0]), hix = voxel_hi(X, g.vd[0], g.omin[0]);
            const double loy = voxel_lo(Y, g.vd[1], g.omin[1]), hiy = voxel_hi(Y, g.vd[1], g.omin[1]);
            const double loz = voxel_lo(Z, g.vd[2], g.omin[2]), hiz = voxel_hi(Z, g.vd[2], g.omin[2]);
            if (!(hx < lox) && !(hy < loy) && !(hz < loz) && !(hx > hix) && !(hy > hiy) && !(hz > hiz)) return true;
        }
        // :713-759 (selects, as in the kernels)
        const bool cxy = tMaxX < tMaxY, cxz = tMaxX < tMaxZ, cyz = tMaxY < tMaxZ;
        const bool sx = cxy & cxz;
        const bool sy = (!cxy) & cyz;
        const bool sz = !(sx | sy);
        const double nX = tMaxX + tDeltaX, nY = tMaxY + tDeltaY, nZ = tMaxZ + tDeltaZ;
        X += sx ? dx1 : 0;
        Y += sy ? dy1 : 0;
        Z += sz ? dz1 : 0;
        tMaxX = sx ? nX : tMaxX;
        tMaxY = sy ? nY : tMaxY;
        tMaxZ = sz ? nZ : tMaxZ;
        if (((unsigned)X >= (unsigned)ct) | ((unsigned)Y >= (unsigned)ct) | ((unsigned)Z >= (unsigned)ct)) break;   // leaving the grid: miss (F12)
        if (own && lane == 0) own->cells++;
    }
    pid = -1;
    return false;
}

}  // namespace
