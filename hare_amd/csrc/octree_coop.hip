// octree_coop.hip -- the cooperative tail of the octree kernel K2p: kernel K2t `hare_octree_tail` (included by kernels.hip).
//
// A lane of K2p carries a ray as ONE chain of dependent steps -- pop a child, fetch its node record, test it, at a leaf fetch
// list entries and their records, now and then an exact test -- ~150 of them for the average ray of config 3 and ten times that
// for the heaviest (nodes up to 245 against 41, list entries up to 2 200 against 102: DESIGN.md section 9), and a launch ends
// when the longest chain does: at 1M rays the tickets are dry at 1.49 ms and the last wave ends at 3.0 ms (tools/timeline_oct.py).
// A K2p wave that has drawn its last ticket and is down to its last few rays therefore stops: it writes their state -- the frames of
// the depth-first walk and the hit so far -- to a per-launch array in device memory and ends.  K2t, launched behind K2p on the
// same stream, walks each of those rays with a group of G lanes -- as shipped G = 64, a WHOLE WAVE per ray (HARE_K2T_GROUP, hare_device.h):
// the up-to-eight children of a frame are fetched and tested at once by eight of the lanes, a leaf's list is pre-culled and tested
// in one go, 64 entries at a time.  Every CU is free by then: the left-over rays of a launch all run at the same time.  (Groups of 8,
// 16 and 32 lanes -- several rays per wave -- were measured and are slower: the groups of a wave are in different loops at different
// times, DESIGN.md section 9b.  The code below is written for any power-of-two G >= 8.)
//
// The walk is Octree.Shoot's ("Octree - alt.cs":159-284), in the frame form of K2p (kernels.hip):
//   * children of a frame are examined from cursor 7 down to 0 (= popped far to near, :286-306), one at a time and in that
//     order as far as decisions go -- the pop-time tests (:207-211) see the closestT of the moment;
//   * the PUSH test of :268 is made on the child's own stored box (K2p derives the same planes from the parent's box; frames K2p
//     opened carry its mask of pushed children, frames opened here carry all eight and are filtered when examined);
//   * inside a leaf the reference scans the list in order, accepts t > 1e-10 && t < closestT (strict) and returns at once when the
//     accepted t is <= the leaf's entry parameter (:233).  The lanes evaluate every entry of a chunk of 64, then the scan is
//     replayed on the results: an entry "improves" when its t is below closestT and below every earlier entry's t (exclusive
//     prefix minimum over the lanes); the first improving entry with t <= nodeTmin ends the query; otherwise the last improving
//     entry is the new closest hit (improving t are strictly decreasing, so ties stay with the earlier entry).
// Candidates are filtered by the same conservative FP32 pre-cull as everywhere else; u, v come from the full RayXtri.
namespace {

__device__ __forceinline__ double coop_min(double a, double b) { return (a < b || a != a) ? a : b; }   // NaN-propagating like Math.Min (omin of K2p)
__device__ __forceinline__ double coop_max(double a, double b) { return (b < a || a != a) ? a : b; }

// One ray from the state a K2p lane left it in, by a GROUP of G consecutive lanes (shipped: G = 64, one ray per wave; G < 64: the groups
// of a wave run independently, each lane's control flow is that of its group): frames [0 .. lvl] (fa, fb, fpk: this group's arrays), the
// current leaf's remaining entries items[q .. qe) with entry parameter leaf_ca, and the hit so far.  All arguments group-uniform.
template <int G>
__device__ __forceinline__ void coop_octree(const OctreeArgs& g, const ShootIO& io, double* fa, double* fb, int* fpk, unsigned ray, int lvl, int q,
                                            int qe, double leaf_ca, double& closestT, double& bu, double& bv, int& pid, bool& hit)
{
    static_assert(G >= 8 && G <= 64 && (G & (G - 1)) == 0, "a group holds the eight children of a frame");
    const int wl = threadIdx.x & 63;
    const int lane = wl & (G - 1);                       // position in the group
    const int gshift = wl - lane;
    auto gballot = [&](bool p) -> unsigned long long {   // the group's lanes for which p holds (bit = position in the group)
        const unsigned long long b = __ballot(p) >> gshift;
        return G == 64 ? b : (b & ((1ull << (G & 63)) - 1ull));
    };
    const RayRec r = io.rays[ray];                       // one address for the wave: a broadcast load
    const V3 o = {r.x, r.y, r.z}, d = {r.dx, r.dy, r.dz};
    const double invDx = fabs(d.x) > 1e-16 ? 1.0 / d.x : 1e16;      // "Octree - alt.cs":165-167
    const double invDy = fabs(d.y) > 1e-16 ? 1.0 / d.y : 1e16;
    const double invDz = fabs(d.z) > 1e-16 ? 1.0 / d.z : 1e16;
    const int mask = ((d.x >= 0 ? 0 : 1) << 2) | ((d.y >= 0 ? 0 : 1) << 1) | (d.z >= 0 ? 0 : 1);
    const int e1 = io.excl1 ? io.excl1[ray] : -1, e2 = io.excl2 ? io.excl2[ray] : -1;      // :218
    const CullRay cray = cull_ray(g, o.x, o.y, o.z, d.x, d.y, d.z);

    // a leaf's entries items[q0 .. qe0): true when the query ended inside it (:233)
    auto leaf = [&](int q0, int qe0, double lca) -> bool {
        for (int base = q0; base < qe0; base += G) {
            const int k = base + lane;
            const bool valid = k < qe0;
            int i = -1;
            if (valid) i = g.items[k];
            bool test = valid && i != e1 && i != e2;
            if (test) test = !cull_test(g, cray, cull_load(g, i));
            double t = kDblMax, u = 0, v = 0;
            if (test) {
                const PolyRec& p = g.polys[i];
                const double* v3 = (g.quads && g.quads[i].nverts == 4) ? g.quads[i].v3 : nullptr;
                double tt, uu, vv;
                if (poly_full(p, v3, o, d, tt, uu, vv) && tt > kTMin) { t = tt; u = uu; v = vv; }      // :224
            }
            // exclusive prefix minimum of t in list order
            double inc = t;
#pragma unroll
            for (int off = 1; off < G; off <<= 1) {
                const double y = __shfl_up(inc, off, G);
                if (lane >= off) inc = y < inc ? y : inc;
            }
            double exc = __shfl_up(inc, 1, G);
            if (lane == 0) exc = kDblMax;
            const double before = closestT < exc ? closestT : exc;
            const bool improving = t < before;                                   // :225, strict: an earlier entry keeps a tie
            const unsigned long long rm = gballot(improving && t <= lca);        // :233
            const unsigned long long im = gballot(improving);
            if (rm) {
                const int l0 = (int)__builtin_ctzll(rm);
                closestT = __shfl(t, l0, G); bu = __shfl(u, l0, G); bv = __shfl(v, l0, G); pid = __shfl(i, l0, G);
                hit = true;
                return true;
            }
            if (im) {
                const int l1 = 63 - (int)__builtin_clzll(im);                   // improving t are strictly decreasing: the last one is the minimum
                closestT = __shfl(t, l1, G); bu = __shfl(u, l1, G); bv = __shfl(v, l1, G); pid = __shfl(i, l1, G);
                hit = true;
            }
        }
        return false;
    };

    if (q < qe && leaf(q, qe, leaf_ca)) return;
    // bounded: every pass either examines the children of a frame (each child once per frame) or drops a frame
    for (int guard = 0; guard < (1 << 22) && lvl >= 0; ++guard) {
        const int pk = fpk[lvl];
        unsigned rem = (unsigned)pk & 255u;
        if (rem == 0) { --lvl; continue; }
        const int first = (int)((unsigned)pk >> 8);
        const double pa = fa[lvl], pb = fb[lvl];
        // the remaining children of this frame, one per lane (cursor position = lane): record, slab interval from its own box (:253-266)
        double ca = 0, cb = 0;
        bool pushed = false;
        int fc = -1, is = 0, ic = 0;
        if (lane < 8 && ((rem >> lane) & 1u)) {
            const OctNode& nd = g.nodes[first + (lane ^ mask)];
            double tx0 = (nd.bmin[0] - o.x) * invDx, tx1 = (nd.bmax[0] - o.x) * invDx;
            double ty0 = (nd.bmin[1] - o.y) * invDy, ty1 = (nd.bmax[1] - o.y) * invDy;
            double tz0 = (nd.bmin[2] - o.z) * invDz, tz1 = (nd.bmax[2] - o.z) * invDz;
            if (invDx < 0) { const double sw = tx0; tx0 = tx1; tx1 = sw; }
            if (invDy < 0) { const double sw = ty0; ty0 = ty1; ty1 = sw; }
            if (invDz < 0) { const double sw = tz0; tz0 = tz1; tz1 = sw; }
            const double tmn = coop_max(coop_max(tx0, ty0), tz0), tmx = coop_min(coop_min(tx1, ty1), tz1);
            pushed = !(tmx < tmn || tmx < 0 || tmn > pb || tmx < pa);            // :268
            ca = coop_max(tmn, pa);                                              // :271
            cb = coop_min(tmx, pb);
            fc = nd.first_child; is = nd.item_start; ic = nd.item_count;
        }
        const unsigned pm = (unsigned)gballot(pushed) & 255u;
        bool descended = false;
        while (rem) {
            const int cur = 31 - __builtin_clz(rem);                             // pop order: cursor 7 down to 0
            rem &= ~(1u << cur);
            if (!((pm >> cur) & 1u)) continue;                                   // never pushed
            const double cca = __shfl(ca, cur, G), ccb = __shfl(cb, cur, G);
            if (ccb < cca || ccb < 0) continue;                                  // :207
            if (hit && closestT <= cca) continue;                                // :210
            const int cfc = __shfl(fc, cur, G);
            if (cfc < 0) {                                                       // a leaf (the device copy's first_child is any negative value)
                const int cis = __shfl(is, cur, G), cic = __shfl(ic, cur, G);
                if (leaf(cis, cis + cic, cca)) return;
            } else {                                                             // an interior node: its frame goes on top, the rest of this one waits
                if (lane == 0) {
                    fpk[lvl] = (int)(((unsigned)first << 8) | rem);
                    fa[lvl + 1] = cca;
                    fb[lvl + 1] = ccb;
                    fpk[lvl + 1] = (int)(((unsigned)cfc << 8) | 255u);           // all eight still to look at; the push test is made when they are
                }
                ++lvl;
                descended = true;
                break;
            }
        }
        if (!descended && lane == 0) fpk[lvl] = (int)((unsigned)first << 8);     // frame exhausted
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");                   // lane 0's frame words, before every lane reads them back
        __builtin_amdgcn_wave_barrier();
    }
}

}  // namespace
