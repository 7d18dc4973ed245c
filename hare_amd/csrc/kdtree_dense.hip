// kdtree_dense.hip -- K3d `hare_kdtree_dense`: KDTree.Shoot (KDTree.cs:198-361) as a production kernel (included by kernels.hip; round 5).
//
// The reference's query visits EVERY leaf of the tree (F4): it pops a node, pushes both children of an interior node -- which one first
// is decided by where the ray crosses the split plane (:249-353) -- and tests every polygon of every leaf with the full RayXtri, keeping
// the smallest t (strict `<`, t > 1e-10; :233).  Nothing about that walk depends on the hits found so far except the accept itself, so
// its RESULT is: the smallest t over all polygons, ties to the polygon met first in the reference's visiting order.  Round 4 made the
// one-ray-per-lane kernel a real tree walk by dropping subtrees whose polygons the ray cannot hit (the subtree TIGHT BOXES, device_scene.cpp:
// make_tight_boxes); this kernel gives that walk the shape the other two partitions' production kernels have (K2d, kernels.hip):
//   * persistent waves, a static first chunk of rays per wave and tickets behind it; a lane owns a ray and is refilled when it finishes;
//   * the node records are a device copy of ONE cache line each (KdDevNode, 128 B: split, the four box bounds the crossing test reads,
//     children, list, and the tight boxes of BOTH children) -- visiting an interior node decides for both children whether they can
//     still matter and pushes only those: `second` with the parameter at which the ray enters its box (a float rounded DOWN), `first`
//     stays in a register and is visited in the next step without a trip through the stack;
//   * a node popped from the stack is dropped WITHOUT being fetched when the ray, by then, holds a hit in front of its box
//     (closestT <= entry: every t in there fails `t < closestT`) -- the far child behind a hit in the near one, the prune a kd-tree is for;
//   * the entries of all leaves the wave's lanes hold are spread densely over the 64 lanes for the FP32 pre-cull (as K2d: an exclusive
//     scan of the counts, owners found by a max-scan over segment starts); survivors are only NOTED, per lane and in list order, and the
//     exact tests (poly_full: RayXtri with u, v) run when enough lanes hold one -- each lane its own, in the order the reference meets
//     them, so the accepted polygon on a tie is the reference's.  The walk runs on with the closestT it has (stale = prunes less).
// Visiting order, candidates per leaf and their order, the arithmetic of the crossing test and of RayXtri are the reference's; only WHICH
// nodes are looked at is smaller, and a node that is skipped is one whose polygons RayXtri would all have rejected or found behind the hit.
// The box tests are used for tame rays only (finite, origin within 1 024 extents of the scene: the range the boxes' margin is sized for,
// as in K2d); other rays -- and every ray when the option octree_tight is off -- visit every node, as the reference does.
//
// OWN (hare_kdtree_dense_own; HARE_SHOOT_COUNT_OWN): the same kernel counting its own node fetches, list entries pre-culled and exact tests.
// OCC (hare_kdtree_occl): the occlusion predicate without events (harness-defined, SURVEY.md 8(a) A9: occluded = the closest hit exists and
// its t is below t_max).  KDTree.Shoot returns the smallest accepted t over ALL polygons, so the flag is "some polygon is accepted with
// t < t_max": the walk ends at the first such hit, and a subtree whose tight box the ray enters at or beyond t_max cannot hold one (every
// accepted t in there is >= the entry) -- the same prune as behind a hit, with t_max as the hit.  (The octree's flag kernel may NOT prune
// by t_max: its reference walk is not a closest-hit query, F15; the kd-tree's is.)
#ifndef HARE_K3D_STEPS
#define HARE_K3D_STEPS 3          // node visits per round at most
#endif
#ifndef HARE_K3D_POP_MIN
#define HARE_K3D_POP_MIN 1        // a second / third visit step only while this many lanes take it (1 / 12 / 24: 598 / 586 / 583 Mrays/s on the hall)
#endif
#ifndef HARE_K3D_REFILL
#define HARE_K3D_REFILL 16        // refill when this many lanes are idle
#endif
#ifndef HARE_K3D_CAP
#define HARE_K3D_CAP 64           // list entries of one leaf that go into one round's dense passes
#endif
#ifndef HARE_K3D_EXACT_MIN
#define HARE_K3D_EXACT_MIN 24     // lanes holding a survivor that make the exact phase run (or one that cannot go on)
#endif

namespace {

// the largest float that is <= x (x finite or -inf / +inf): the stack keeps a node's box entry parameter as a float and may only ever
// UNDER-state it (a node is dropped when closestT <= that float, hence closestT <= the true entry)
__device__ __forceinline__ float float_below(double x)
{
    float f = (float)x;                                   // round to nearest
    if ((double)f > x) {
        const int b = __float_as_int(f);
        f = f > 0.0f ? __int_as_float(b - 1) : (f < 0.0f ? __int_as_float(b + 1) : -1.401298464e-45f);
    }
    return f;
}

template <bool OWN, bool OCC = false>
__device__ __forceinline__ void kdtree_dense_body(const KdArgs& g, const ShootIO& io)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    constexpr int nt = 256;
    constexpr int P = HARE_K3D_PEND;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int slots = g.max_depth + 2;
    int* const st_node = reinterpret_cast<int*>(lds);                          // [slots][nt] nodes pushed and not yet popped (the `second`s)
    float* const st_un = reinterpret_cast<float*>(st_node + (size_t)slots * nt);   // [slots][nt] lower bound of the entry into their tight box
    int* const pend_w = reinterpret_cast<int*>(st_un + (size_t)slots * nt);    // [P][nt] survivors noted and not yet tested
    int* const seg_mark = pend_w + (size_t)P * nt + (tid >> 6) * 64;           // per wave: segment starts of the dense pass
    OwnWork ownw;

    const int RAY_CHUNK = io.static_rays > 0 ? io.static_rays : 128;
    const unsigned int n32 = (unsigned int)io.n;
    const unsigned int n_static = gridDim.x * 4u * (unsigned int)RAY_CHUNK;
    unsigned int chunk_id = blockIdx.x * 4u + (threadIdx.x >> 6);
    if ((gridDim.x & 7u) == 0) chunk_id = ((blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3)) * 4u + (threadIdx.x >> 6);   // XCD-contiguous
    unsigned int cn = chunk_id * (unsigned int)RAY_CHUNK, ce = cn + (unsigned int)RAY_CHUNK;
    if (cn > n32) cn = n32;
    if (ce > n32) ce = n32;
    bool drained = false;
    unsigned int dead_run = 0;          // SHOOT_RETIRED_RAYS: tickets in a row whose rays were all retired
    bool chunk_live = true;

    bool alive = false, hit = false, tight_ok = false;
    unsigned int ray = 0;
    V3 o = {0, 0, 0}, d = {0, 0, 0};
    double ivx = 0, ivy = 0, ivz = 0;
    CullRay cray = {};
    int e1 = -1, e2 = -1;
    int sp = 0;                     // entries on this lane's stack
    int cur = -1;                   // the node to visit next (-1: pop one)
    int q = 0, qe = 0;              // the leaf in hand: items[q .. qe) still to pre-cull
    int np = 0;                     // survivors noted
    double closestT = kDblMax, bu = 0, bv = 0;
    double tlim = kDblMax;          // OCC: t_max of this ray (the bound a subtree's entry is held against while no hit is held)
    int pid = -1;
    unsigned int nhits = 0, nrays = 0;

    auto timeline = [&](int slot) {
        if (__builtin_expect((io.flags & 0x2000u) != 0 && io.prof != nullptr, 0)) {
            if (lane == 0) io.prof[32 + 4ull * (blockIdx.x * 4u + (threadIdx.x >> 6)) + slot] = __builtin_amdgcn_s_memrealtime();
        }
    };
    timeline(0);

    auto finish = [&]() {
        if (OCC) {
            const bool occ = hit && (io.tmax == nullptr || closestT < io.tmax[ray]);
            io.occluded[ray] = occ ? 1 : 0;
            if (occ) nhits++;                               // flags only: the batch counter `hits` counts occluded rays
            alive = false;
            return;
        }
        XEventRec ev;
        if (hit) {
            ev.t = closestT; ev.u = bu; ev.v = bv;
            ev.x = o.x + d.x * closestT; ev.y = o.y + d.y * closestT; ev.z = o.z + d.z * closestT;     // as trace_kdtree at the accept: same expression
            ev.poly_id = pid;
            ev.hit = 1;
            nhits++;
        } else {
            set_miss(ev);
        }
        store_event_streaming(&io.out[ray], ev);
        alive = false;
    };
    // entry / exit parameter of the ray through a tight box {x0,y0,z0,x1,y1,z1}: the expressions of trace_kdtree (hare_trace.h) -- fmax / fmin
    // drop a NaN operand, so a slab whose 1/d is infinite says nothing unless the origin lies outside it
    auto box_entry = [&](const float* b, double& un, double& uf) {
        const double ux0 = ((double)b[0] - o.x) * ivx, ux1 = ((double)b[3] - o.x) * ivx;
        const double uy0 = ((double)b[1] - o.y) * ivy, uy1 = ((double)b[4] - o.y) * ivy;
        const double uz0 = ((double)b[2] - o.z) * ivz, uz1 = ((double)b[5] - o.z) * ivz;
        un = __builtin_fmax(__builtin_fmax(__builtin_fmin(ux0, ux1), __builtin_fmin(uy0, uy1)), __builtin_fmin(uz0, uz1));
        uf = __builtin_fmin(__builtin_fmin(__builtin_fmax(ux0, ux1), __builtin_fmax(uy0, uy1)), __builtin_fmax(uz0, uz1));
    };

    for (;;) {
        // ------------------------------------------------------------------ refill idle lanes
        const unsigned long long idle = __ballot(!alive);
        if (__builtin_expect(!drained && (__popcll(idle) >= HARE_K3D_REFILL || idle == ~0ull), 0)) {
            bool want = !alive;
            while (true) {
                const unsigned long long wm = __ballot(want);
                if (wm == 0) break;
                if (cn >= ce) {
                    unsigned int base = 0;
                    unsigned int dyn = (unsigned int)io.ticket_rays;
                    if (io.flags & SHOOT_RETIRED_RAYS) {
                        // a cast of the bounce loop: a ticket whose rays the loop had all retired cost nothing but its draw -- every such
                        // ticket in a row doubles the next one (up to 64 x), the first live ray puts the size back (as K1q, voxel_pool.hip)
                        dead_run = chunk_live ? 0u : (dead_run < 6u ? dead_run + 1u : 6u);
                        chunk_live = false;
                        dyn <<= dead_run;
                    }
                    if (lane == 0) base = atomicAdd(io.work, dyn);
                    base = __shfl(base, 0, 64);
                    cn = base + n_static;
                    if (cn >= n32) { drained = true; timeline(1); break; }
                    ce = (n32 - cn > dyn) ? cn + dyn : n32;
                }
                const unsigned int mine = cn + rank_below(wm);
                const bool got = want && mine < ce;
                cn += (unsigned int)__popcll(__ballot(got));
                bool live_lane = false;
                if (got) {
                    want = false;
                    ray = mine;
                    const RayRec r = io.rays[ray];
                    o.x = r.x; o.y = r.y; o.z = r.z;
                    d.x = r.dx; d.y = r.dy; d.z = r.dz;
                    e1 = io.excl1 ? io.excl1[ray] : -1;
                    e2 = io.excl2 ? io.excl2[ray] : -1;
                    hit = false; alive = true;
                    closestT = kDblMax; pid = -1; bu = 0; bv = 0;
                    sp = 0; cur = -1; q = 0; qe = 0; np = 0;
                    if (e1 == -2 && (io.flags & SHOOT_RETIRED_RAYS)) {
                        finish();                                           // retired by the bounce loop: miss record, not counted
                    } else {
                        nrays++;
                        live_lane = true;
                        if (OCC) {
                            tlim = io.tmax ? io.tmax[ray] : kDblMax;
                            if (!(tlim == tlim)) tlim = kDblMax;            // a NaN t_max compares false with every t: never occluded, nothing to prune by
                        }
                        cray = cull_ray(g, o.x, o.y, o.z, d.x, d.y, d.z);
                        tight_ok = g.tight != nullptr && fabs(o.x - g.tight_mid[0]) <= g.tight_rad && fabs(o.y - g.tight_mid[1]) <= g.tight_rad &&
                                   fabs(o.z - g.tight_mid[2]) <= g.tight_rad && fabs(d.x) < 1e300 && fabs(d.y) < 1e300 && fabs(d.z) < 1e300;
                        ivx = 1.0 / d.x; ivy = 1.0 / d.y; ivz = 1.0 / d.z;
                        cur = 0;                                            // the root (KDTree.cs:211)
                        if (tight_ok) {
                            const float4* tp = reinterpret_cast<const float4*>(g.tight);
                            const float4 t0 = tp[0], t1 = tp[1];
                            const float rb[6] = {t0.x, t0.y, t0.z, t0.w, t1.x, t1.y};
                            double un, uf;
                            box_entry(rb, un, uf);
                            if ((uf < un) | (uf < 0)) finish();             // the ray misses every polygon of the tree
                        }
                    }
                }
                if (io.flags & SHOOT_RETIRED_RAYS) {
                    chunk_live = chunk_live || __ballot(live_lane) != 0ull;
                    want = want || (got && !alive);          // a lane that drew a retired ray draws again: its record is written, it holds nothing
                }
            }
        }
        {
            const unsigned long long am = __ballot(alive);
            if (am == 0) {
                if (drained) break;
                continue;
            }
        }

        // ------------------------------------------------------------------ phase P: visit a node per step
#pragma unroll 1
        for (int k = 0; k < HARE_K3D_STEPS; ++k) {
            // a lane whose walk is over and that holds no survivor is done; one that holds survivors waits for the exact phase
            if (alive && q == qe && cur < 0 && sp == 0 && np == 0) finish();
            const bool step = alive && np < P && q == qe && (cur >= 0 || sp > 0);
            {
                const unsigned long long pm = __ballot(step);
                if (pm == 0 || (k > 0 && __popcll(pm) < HARE_K3D_POP_MIN)) break;
            }
            if (step) {
                if (cur < 0) {                                              // pop (KDTree.cs:215)
                    --sp;
                    const int node = st_node[sp * nt + tid];
                    const float un = st_un[sp * nt + tid];
                    // holding a hit in front of the node's box: nothing in there can be accepted (t < closestT) -- dropped unfetched
                    cur = ((hit && closestT <= (double)un) || (OCC && tlim <= (double)un)) ? -1 : node;
                }
                if (cur >= 0) {
                    const KdDevNode nd = g.dnodes[cur];                     // one 128-byte line
                    if (OWN) ownw.cells++;
                    if (nd.axis < 0) {                                      // a leaf: its list goes to the dense pre-cull
                        q = nd.item_start; qe = nd.item_start + nd.item_count;
                        cur = -1;
                    } else {
                        // :249-353 -- the three SplitAxis branches are one pattern; the two other axes in ascending order (bb, device_scene.cpp)
                        const int a = nd.axis;
                        const double oa = a == 0 ? o.x : (a == 1 ? o.y : o.z);
                        const double da = a == 0 ? d.x : (a == 1 ? d.y : d.z);
                        const double ob = a == 0 ? o.y : o.x;
                        const double db = a == 0 ? d.y : d.x;
                        const double oc = a == 2 ? o.y : o.z;
                        const double dc = a == 2 ? d.y : d.z;
                        const double side = oa - nd.split;
                        const double tSplit = -side / da;
                        const double bS = ob + tSplit * db;
                        const double cS = oc + tSplit * dc;
                        const bool inside = bS <= nd.bb[1] && bS >= nd.bb[0] && cS <= nd.bb[3] && cS >= nd.bb[2];
                        const bool first_right = inside ? (side >= 0) : !(side >= 0);
                        const int first = first_right ? nd.right : nd.left, second = first_right ? nd.left : nd.right;
                        bool ok1 = true, ok2 = true;
                        double un2 = -kDblMax;
                        if (tight_ok) {
                            double unL, ufL, unR, ufR;
                            box_entry(nd.tl, unL, ufL);
                            box_entry(nd.tr, unR, ufR);
                            const bool okL = !((ufL < unL) | (ufL < 0) | (hit & (closestT <= unL)) | (OCC & (tlim <= unL)));
                            const bool okR = !((ufR < unR) | (ufR < 0) | (hit & (closestT <= unR)) | (OCC & (tlim <= unR)));
                            ok1 = first_right ? okR : okL;
                            ok2 = first_right ? okL : okR;
                            un2 = first_right ? unL : unR;
                        }
                        // a subtree that lists no polygon at all: visiting it has no effect whatever the ray (a fact of the tree, not a box test)
                        ok1 = ok1 && !(nd.empty & (first_right ? 2 : 1));
                        ok2 = ok2 && !(nd.empty & (first_right ? 1 : 2));
                        if (ok2) {                                          // :355-356: second is pushed first, i.e. popped after first's subtree
                            st_node[sp * nt + tid] = second;
                            st_un[sp * nt + tid] = float_below(un2);
                            ++sp;
                        }
                        cur = ok1 ? first : -1;                             // first: visited next, without a trip through the stack
                    }
                }
            }
        }

        // ------------------------------------------------------------------ B1: every leaf entry in hand, one per lane (as K2d)
        {
            const bool own = alive && np < P && q < qe;
            const int cnt = own ? (qe - q < HARE_K3D_CAP ? qe - q : HARE_K3D_CAP) : 0;
            const int inc = wave_scan_add(cnt);
            const int off = inc - cnt;
            const int total = __builtin_amdgcn_readlane(inc, 63);
            const int q0 = q;
            bool stop = false;
            // the owner of item (base + lane) and the item's list position (as K2d, kernels.hip: lanes past the end read entry 0)
            auto window = [&](int base, int& ow) -> int {
                seg_mark[lane] = -1;
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
                if (cnt > 0 && off < base + 64 && off + cnt > base) seg_mark[off > base ? off - base : 0] = lane;
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
                const int owner = wave_scan_max(seg_mark[lane]);
                const bool valid = base + lane < total;
                ow = valid ? owner : lane;
                const int rel_o = __shfl(q0 - off, ow, 64);
                return valid ? rel_o + base + lane : 0;
            };
#if HARE_K3D_AHEAD
            // the windows as a pipeline (round 6, as K2d): list entries two windows ahead, pre-cull records one
            int ow1 = lane, ow2 = lane, i1 = 0, i2 = 0;
            CullRaw rec1;
            if (total > 0) {
                const int p0 = window(0, ow1);
                const int p1 = total > 64 ? window(64, ow2) : 0;
                i1 = g.items[p0];
                i2 = g.items[p1];
                rec1 = cull_load(g, i1);
            }
#endif
#pragma unroll 1
            for (int base = 0; base < total; base += 64) {
                const bool valid = base + lane < total;
#if HARE_K3D_AHEAD
                const int ow = ow1, i_now = i1;
                const CullRaw rec = rec1;
                ow1 = ow2; i1 = i2;
                rec1 = cull_load(g, i1);
                {
                    int own_ = lane;
                    const int p2 = base + 128 < total ? window(base + 128, own_) : 0;
                    ow2 = own_;
                    i2 = g.items[p2];
                }
#else
                int ow;
                const int i_now = g.items[window(base, ow)];
                const CullRaw rec = cull_load(g, i_now);
#endif
                CullRay cr;
                cr.ox = __shfl(cray.ox, ow, 64); cr.oy = __shfl(cray.oy, ow, 64); cr.oz = __shfl(cray.oz, ow, 64);
                cr.dfx = __shfl(cray.dfx, ow, 64); cr.dfy = __shfl(cray.dfy, ow, 64); cr.dfz = __shfl(cray.dfz, ow, 64);
#if HARE_CULL32
                cr.err = __builtin_fmaf(2.3841858e-07f /* 2^-22 */, fabsf(cr.ox) + fabsf(cr.oy) + fabsf(cr.oz), g.cf.err0);   // as cull_ray
#endif
                cr.dm = fabsf(cr.dfx) + fabsf(cr.dfy) + fabsf(cr.dfz);
                const int i = valid ? i_now : -1;
                const bool surv = valid && !cull_test(g, cr, rec);
                if (OWN && valid) { ownw.entries++; ownw.culls++; }
                const unsigned long long sb = __ballot(surv);
                const int lo = off > base ? off - base : 0;
                const int hi = off + cnt - base < 64 ? off + cnt - base : 64;
                const bool mine = cnt > 0 && !stop && lo < hi && hi > 0 && lo < 64;
                unsigned long long seg = 0;
                if (mine) {
                    const unsigned long long m_hi = hi >= 64 ? ~0ull : ((1ull << hi) - 1ull);
                    seg = sb & m_hi & ~((1ull << lo) - 1ull);
                }
                int consumed = mine ? hi - lo : 0;
#pragma unroll
                for (int r = 0; r < P; ++r) {
                    const bool take = mine && seg != 0 && np < P;
                    const int pos = take ? (int)__builtin_ctzll(seg) : lane;
                    const int poly = __shfl(i, pos, 64);
                    if (take) {
                        // :221, applied by the owner -- and not the polygon the ray's hit lies on AGAIN (HARE_K2D_SKIP_PID, as K2d: the same t, never < closestT)
                        if (poly != e1 && poly != e2 && !(HARE_K2D_SKIP_PID && hit && poly == pid)) {
                            pend_w[np * nt + tid] = poly;
                            ++np;
                        }
                        seg &= seg - 1ull;
                    }
                }
                if (mine && seg != 0) {                                      // survivors left over: the list is full -- they are scanned again later
                    consumed = (int)__builtin_ctzll(seg) - lo;
                    stop = true;
                }
                if (mine) q += consumed;
            }
        }
        // ------------------------------------------------------------------ B2: the noted survivors, each lane its own, in order (:225-241)
        {
            const bool over = alive && q == qe && cur < 0 && sp == 0;
            const unsigned long long holding = __ballot(alive && np > 0);
            const unsigned long long blocked = __ballot(alive && np > 0 && (np >= P || over));
            if (holding != 0 && (blocked != 0 || __popcll(holding) >= HARE_K3D_EXACT_MIN)) {
                bool decided = false;
#pragma unroll 1
                for (int k = 0; k < P; ++k) {
                    const bool act = alive && k < np && !decided;
                    if (__ballot(act) == 0) break;
                    if (act) {
                        const int i = pend_w[k * nt + tid];
                        const PolyRec& p = g.polys[i];
                        const double* v3 = (g.quads && g.quads[i].nverts == 4) ? g.quads[i].v3 : nullptr;
                        double t, u, v;
                        if (OWN) ownw.tests++;
                        if (poly_full(p, v3, o, d, t, u, v) && t > kTMin) {              // :233
                            if (t < closestT) {
                                closestT = t; bu = u; bv = v; pid = i;
                                hit = true;
                            }
                            if (OCC && (io.tmax == nullptr || closestT < io.tmax[ray])) decided = true;      // some polygon lies in front of t_max: the flag is 1
                        }
                    }
                }
                np = alive ? 0 : np;
                if (OCC && decided) finish();
            }
            if (alive && q == qe && cur < 0 && sp == 0 && np == 0) finish();
        }
    }
    timeline(2);
    if (OWN) flush_own(io.ctr, ownw);
    launch_epilogue(io, nrays, nhits, 4u);
}

}  // namespace

extern "C" {
__global__ __launch_bounds__(256, HARE_K3D_WAVES_PER_EU) void hare_kdtree_dense(hare::KdArgs g, hare::ShootIO io) { kdtree_dense_body<false>(g, io); }
__global__ __launch_bounds__(256, HARE_K3D_WAVES_PER_EU) void hare_kdtree_dense_own(hare::KdArgs g, hare::ShootIO io) { kdtree_dense_body<true>(g, io); }
// the occlusion predicate, flags only (hare_occluded_* with events == NULL)
__global__ __launch_bounds__(256, HARE_K3D_WAVES_PER_EU) void hare_kdtree_occl(hare::KdArgs g, hare::ShootIO io) { kdtree_dense_body<false, true>(g, io); }
}
