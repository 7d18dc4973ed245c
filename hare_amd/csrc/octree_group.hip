// octree_group.hip -- K2g `hare_octree_group`: Octree.Shoot ("Octree - alt.cs":154-306) with a GROUP OF EIGHT LANES per ray,
// eight rays per wave (included by kernels.hip).
//
// Why.  K2p gives a ray one lane: ~150 dependent steps (pop a child, fetch it, test it, at a leaf fetch list entries one pair at
// a time, now and then an exact test) of 4 us each -- the life of a ray is 600 us, a 1M-ray launch is four generations of rays
// deep and a third of it is the last generation draining.  Here the eight children of a node are tested by
// the group's eight lanes AT ONCE and a leaf's list goes eight entries at a time: ~30 steps per ray instead of 150, 1/16 of the
// rays in flight for the same throughput, and the end of a launch shrinks with the life of a ray.
//
// What makes that possible is that almost nothing in Octree.Shoot depends on the hits found so far:
//   * WHICH nodes are pushed, in which order, with which clamped interval [tmin, tmax] (:245-272) depends on the ray and the boxes
//     only; so does the first pop test (:207).  The group keeps the reference's explicit LIFO stack in LDS: entry = the clamped
//     interval + the node's child / list words; the eight children of a popped interior node are tested by lanes 0..7 (lane k =
//     order[k], :286-306) and the accepted ones are pushed in order[0..7], so that order[7] is popped first, as in the reference.
//     Entries that the hit-independent pop test (:207) would drop are never pushed, nor are leaves with an empty list (popping
//     one has no effect).
//   * the value of RayXtri (t, u, v) for a (ray, polygon) pair depends on nothing else, and the conservative FP32 pre-cull in front
//     of it is stateless.
//   * the ONLY state is {hit, closestT, the best event}: it prunes (:210 `hit && closestT <= nodeTmin` -> skip the node), accepts
//     (:225 strict <) and ends the query (:233 accepted t <= the leaf's nodeTmin).  A larger (older) closestT prunes less, never
//     more: walking with a stale closestT visits a superset of the reference's nodes, in the reference's order.
// So the walk runs ahead with the closestT it has: a leaf's entries are pre-culled eight at a time and the survivors are only
// NOTED -- polygon, the leaf's nodeTmin, the leaf's visit number -- in a per-ray pending list, in walk order.  When enough are
// pending in the wave (or a ray cannot go on without them) the exact tests of all pending survivors run together, one per lane,
// and their results are REPLAYED in order against the state, exactly as the reference meets them:
//     entering a new leaf (first pending survivor with a new visit number):  skipped = hit && closestT <= leafTmin      (:210)
//     not skipped, t > 1e-10 && t < closestT:  accept;  accepted t <= leafTmin:  the query ends                        (:224-236)
// (the skip decision is made at the leaf's first VALID survivor instead of at its pop: the state cannot change in between.)
// A leaf the reference would have skipped or never reached was pre-culled in vain; its survivors are discarded by the replay.
// Results are bit-identical to the sequential walk (tests: every octree test of the suite runs against this kernel).
//
// LDS per group: the stack's top kGStack entries (24 B each) + 16 pending survivors (16 B each); entries below the top kGStack
// spill to a per-launch block in device memory (the scene's octree scratch ring, launch.cpp), so a stack of any depth the tree
// allows (7 x levels + 8) works, at LDS speed for all but pathological rays.
namespace {

#ifndef HARE_K2G_EXACT_MIN
#define HARE_K2G_EXACT_MIN 12      // run the exact phase when this many survivors are pending in the wave (or a ray is blocked on its own)
#endif
#ifndef HARE_K2G_REFILL
#define HARE_K2G_REFILL 2          // set up new rays when this many groups of the wave are idle
#endif
constexpr int kGStack = kGroupStack;                              // stack entries per ray kept in LDS (hare_device.h); deeper ones spill
constexpr int kGPend = kGroupPend;                                // pending survivors per ray (two chunks' worth)
constexpr int kGGroupBytes = kGroupBytes;
constexpr int kGWaveBytes = kGroupWaveBytes;

// TAIL = false: the rays of a batch (static first chunk per wave, then tickets).
// TAIL = true (`hare_octree_group_tail`, launched behind K2p on its stream): the rays K2p's waves handed over when the tickets ran dry
// (OctTailRec + frames, kernels.hip) -- ALL the rays a K2p wave was still walking, not only its last few: as lanes of K2p they would
// take the life of a ray, ~600 us, to finish at falling occupancy; here each gets eight lanes and they are done in a third of that.
// A K2p frame (first_child, children still to pop, interval) becomes a stack entry whose eight-bit child mask restricts the F phase
// to those children; frames go on the stack bottom-up, so that the deepest frame's children are popped first, as K2p would have.
template <bool TAIL>
__device__ __forceinline__ void octree_group_body(const OctreeArgs& g, const ShootIO& io)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    constexpr int W = 8;                                           // lanes per ray
    const int wl = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = wl & (W - 1);                                    // position in the group = cursor position of the child this lane tests
    const int gshift = wl - j;
    const unsigned jbit = 1u << j, jlt = jbit - 1u;
    auto gballot = [&](bool p) -> unsigned { return (unsigned)(__ballot(p) >> gshift) & 255u; };

    unsigned char* const gl = lds + (size_t)wave * kGWaveBytes + (size_t)(wl >> 3) * kGGroupBytes;
    double* const sa = reinterpret_cast<double*>(gl);             // [kGStack] clamped tmin of the entry's node
    double* const sb = sa + kGStack;                               // [kGStack] clamped tmax
    int2* const sw = reinterpret_cast<int2*>(sb + kGStack);        // [kGStack] interior: {first_child, mask of children to examine}; leaf: {-1 - item_start, item_count}
    double* const plca = reinterpret_cast<double*>(sw + kGStack);  // [kGPend] pending survivor: its leaf's nodeTmin
    int2* const ppw = reinterpret_cast<int2*>(plca + kGPend);      // [kGPend] {polygon, leaf visit number}
    // entries kGStack.. of this group's stack: its block of the launch's spill area (sized by the host for 7 * levels + 8 entries)
    const int spill_cap = io.oct_spill_cap;                        // entries per group in the spill block
    unsigned char* const spill = io.oct_spill == nullptr ? nullptr :
        io.oct_spill + ((size_t)(blockIdx.x * (blockDim.x >> 6) + wave) * 8u + (size_t)(wl >> 3)) * (size_t)spill_cap * 24u;
    auto st_store = [&](int e, double a, double b, int2 w) {
        if (e < kGStack) { sa[e] = a; sb[e] = b; sw[e] = w; }
        else {
            unsigned char* p = spill + (size_t)(e - kGStack) * 24u;
            *reinterpret_cast<double*>(p) = a; *reinterpret_cast<double*>(p + 8) = b; *reinterpret_cast<int2*>(p + 16) = w;
        }
    };
    auto st_load = [&](int e, double& a, double& b, int2& w) {
        if (e < kGStack) { a = sa[e]; b = sb[e]; w = sw[e]; }
        else {
            const unsigned char* p = spill + (size_t)(e - kGStack) * 24u;
            a = *reinterpret_cast<const double*>(p); b = *reinterpret_cast<const double*>(p + 8); w = *reinterpret_cast<const int2*>(p + 16);
        }
    };
    auto lds_sync = [&]() {                                        // one lane's LDS / spill writes, before other lanes of the wave read them
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    };

    // ---- rays: a static first chunk per wave, then tickets (as K1p / K2p)
    const unsigned int n32 = (unsigned int)io.n;
    const unsigned int waves_per_block = blockDim.x >> 6;
    const int RAY_CHUNK = io.static_rays > 0 ? io.static_rays : 32;
    const unsigned int n_static = gridDim.x * waves_per_block * (unsigned int)RAY_CHUNK;
    unsigned int chunk_id = blockIdx.x * waves_per_block + (unsigned)wave;
    if ((gridDim.x & 7u) == 0) chunk_id = ((blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3)) * waves_per_block + (unsigned)wave;   // XCD-contiguous
    unsigned int cn = chunk_id * (unsigned int)RAY_CHUNK, ce = cn + (unsigned int)RAY_CHUNK;
    if (cn > n32) cn = n32;
    if (ce > n32) ce = n32;
    bool drained = false;
    LaunchSlotMem* const sm = reinterpret_cast<LaunchSlotMem*>(io.work);
    const unsigned tail_count = TAIL ? sm->oct_tail_count : 0u;   // final: K2p has ended (stream order)

    // ---- per-ray state, uniform over the group's eight lanes
    bool alive = false, hit = false, tame = true;
    bool tight_ok = false;              // the subtrees' tight boxes may be used for this ray (g.tight, device_scene.cpp: make_tight_boxes): tame, origin near the scene
    unsigned int ray = 0;
    V3 o = {0, 0, 0}, d = {0, 0, 0};
    double invDx = 0, invDy = 0, invDz = 0;
    CullRay cray = {};
    int mask = 0, e1 = -1, e2 = -1;
    int sp = 0;                         // stack entries
    int q = 0, qe = 0;                  // the current leaf's remaining entries items[q .. qe)
    double lca = 0;                     // its nodeTmin
    int visit = 0;                      // leaves visited so far (numbers the pending survivors' leaves)
    int np = 0;                         // pending survivors
    int cur_visit = -1;                 // replay: the leaf the last replayed survivor belonged to ...
    bool cur_skip = false;              // ... and whether the reference skipped it (:210)
    double closestT = kDblMax, bu = 0, bv = 0;
    int pid = -1;
    unsigned int nhits = 0, nrays = 0;  // counted in lane 0 of the group
#ifdef HARE_K2G_STATS                   // developer build: what an iteration of the loop is made of (tools/k2g_stats.py); lane 0 counts
    unsigned long long st_iter = 0, st_f = 0, st_fg = 0, st_l = 0, st_lg = 0, st_e = 0, st_el = 0, st_pop = 0, st_popg = 0, st_ent = 0, st_alive = 0, st_valid = 0;
#define K2G_STAT(x) x
#else
#define K2G_STAT(x)
#endif

    auto finish = [&]() {
        if (j == 0) {
            XEventRec ev;
            if (hit) {
                ev.t = closestT; ev.u = bu; ev.v = bv;
                ev.x = o.x + d.x * closestT; ev.y = o.y + d.y * closestT; ev.z = o.z + d.z * closestT;
                ev.poly_id = pid;
                ev.hit = 1;
                nhits++;
            } else {
                set_miss(ev);
            }
            store_event_streaming(&io.out[ray], ev);
        }
        alive = false;
        sp = 0; q = 0; qe = 0; np = 0;
    };

    for (;;) {
        // ------------------------------------------------------------------ new rays for idle groups
        {
            const unsigned long long idle = __ballot(!alive);
            const int nidle = __popcll(idle) >> 3;
            if (__builtin_expect(!drained && (nidle >= HARE_K2G_REFILL || nidle == 8), 0)) {
                bool want = !alive;
                while (true) {
                    const unsigned long long wm = __ballot(want);
                    if (wm == 0) break;
                    if (TAIL) {
                        // one draw for all the groups that want a record
                        const unsigned int k = (unsigned int)__popcll(wm) >> 3;
                        unsigned int base = 0;
                        if (wl == 0) base = atomicAdd(&sm->oct_tail_next, k);
                        base = (unsigned)__builtin_amdgcn_readfirstlane((int)base);
                        cn = base < tail_count ? base : tail_count;
                        ce = base + k < tail_count ? base + k : tail_count;
                        if (cn >= ce) { drained = true; break; }
                        if (ce - cn < k) drained = true;              // the list ends inside this draw
                    } else if (cn >= ce) {
                        unsigned int base = 0;
                        const unsigned int dyn = (unsigned int)io.ticket_rays;
                        if (wl == 0) base = atomicAdd(io.work, dyn);
                        base = (unsigned)__builtin_amdgcn_readfirstlane((int)base);
                        cn = base + n_static;
                        if (cn >= n32) { drained = true; break; }
                        ce = (n32 - cn > dyn) ? cn + dyn : n32;
                    }
                    const unsigned int mine = cn + ((unsigned int)__popcll(wm & ((1ull << gshift) - 1ull)) >> 3);   // idle groups in front of this one
                    const bool got = want && mine < ce;
                    cn += (unsigned int)__popcll(__ballot(got)) >> 3;
                    if (TAIL && !got) want = false;                    // no record left for this group
                    const unsigned char* trec = nullptr;
                    OctTailRec th = {};
                    if (TAIL && got) {
                        trec = io.oct_tail + (size_t)mine * (size_t)io.oct_tail_stride;
                        th = *reinterpret_cast<const OctTailRec*>(trec);       // one address per group
                    }
                    if (got) {
                        want = false;
                        ray = TAIL ? th.ray : mine;
                        const RayRec r = io.rays[ray];             // one address per group
                        o.x = r.x; o.y = r.y; o.z = r.z;
                        d.x = r.dx; d.y = r.dy; d.z = r.dz;
                        e1 = io.excl1 ? io.excl1[ray] : -1;
                        e2 = io.excl2 ? io.excl2[ray] : -1;
                        hit = false; alive = true;
                        tame = fabs(o.x) < 1e300 && fabs(o.y) < 1e300 && fabs(o.z) < 1e300 && fabs(d.x) < 1e300 && fabs(d.y) < 1e300 && fabs(d.z) < 1e300;
                        tight_ok = tame && g.tight != nullptr && fabs(o.x - g.tight_mid[0]) <= g.tight_rad && fabs(o.y - g.tight_mid[1]) <= g.tight_rad &&
                                   fabs(o.z - g.tight_mid[2]) <= g.tight_rad;
                        closestT = kDblMax; pid = -1; bu = 0; bv = 0;
                        sp = 0; q = 0; qe = 0; np = 0; visit = 0; cur_visit = -1; cur_skip = false;
                        if (!TAIL && e1 == -2 && (io.flags & SHOOT_RETIRED_RAYS)) {
                            finish();
                        } else {
                            if (!TAIL && j == 0) nrays++;          // a handed-over ray was counted by K2p when it set it up
                            invDx = fabs(d.x) > 1e-16 ? 1.0 / d.x : 1e16;      // "Octree - alt.cs":165-167
                            invDy = fabs(d.y) > 1e-16 ? 1.0 / d.y : 1e16;
                            invDz = fabs(d.z) > 1e-16 ? 1.0 / d.z : 1e16;
                            mask = ((d.x >= 0 ? 0 : 1) << 2) | ((d.y >= 0 ? 0 : 1) << 1) | (d.z >= 0 ? 0 : 1);
                            cray = cull_ray(g, o.x, o.y, o.z, d.x, d.y, d.z);
                            if (TAIL) {
                                // the walk state K2p left: the hit so far, the leaf in hand, the frames bottom-up
                                closestT = th.closestT; bu = th.bu; bv = th.bv; pid = th.pid; hit = th.hit != 0;
                                q = th.q; qe = th.qe; lca = th.leaf_ca;
                                if (q < qe) visit = 1;
                                const int levels = io.oct_tail_levels;
                                const double* ra = reinterpret_cast<const double*>(trec + kOctTailHead);
                                const double* rb = ra + levels;
                                const int* rk = reinterpret_cast<const int*>(rb + levels);
                                for (int base = 0; base <= th.lvl; base += W) {
                                    const int k = base + j;
                                    int pk = 0;
                                    double fa_ = 0, fb_ = 0;
                                    if (k <= th.lvl) { pk = rk[k]; fa_ = ra[k]; fb_ = rb[k]; }
                                    const bool open = (pk & 255) != 0;                        // children still to pop
                                    const unsigned om = gballot(open);
                                    if (open) st_store(sp + __builtin_popcount(om & jlt), fa_, fb_, make_int2((int)((unsigned)pk >> 8), pk & 255));
                                    sp += __builtin_popcount(om);
                                }
                            } else {
                            const OctNode& root = g.nodes[0];
                            double tx0 = (root.bmin[0] - o.x) * invDx, tx1 = (root.bmax[0] - o.x) * invDx;
                            double ty0 = (root.bmin[1] - o.y) * invDy, ty1 = (root.bmax[1] - o.y) * invDy;
                            double tz0 = (root.bmin[2] - o.z) * invDz, tz1 = (root.bmax[2] - o.z) * invDz;
                            if (invDx < 0) { const double s = tx0; tx0 = tx1; tx1 = s; }
                            if (invDy < 0) { const double s = ty0; ty0 = ty1; ty1 = s; }
                            if (invDz < 0) { const double s = tz0; tz0 = tz1; tz1 = s; }
                            const double rmin = omax(omax(tx0, ty0), tz0), rmax = omin(omin(tx1, ty1), tz1);   // :182-183
                            if (rmax < rmin || rmax < 0) {
                                finish();                            // :185 (and the identical pop test :207)
                            } else {
                                const int rfc = root.first_child;
                                const bool rleaf = rfc < 0;
                                if (rleaf && root.item_count == 0) {
                                    finish();
                                } else {
                                    if (j == 0) { sa[0] = rmin; sb[0] = rmax; sw[0] = rleaf ? make_int2(-1 - root.item_start, root.item_count) : make_int2(rfc, 255); }
                                    sp = 1;
                                }
                            }
                            }
                        }
                    }
                }
                lds_sync();
            }
            if (__ballot(alive) == 0) {
                if (drained) break;
                continue;
            }
        }

#ifndef HARE_K2G_REPS
#define HARE_K2G_REPS 3            // POP / F / L rounds per pass of the loop: the refill test, the exact-phase test and the loop itself are paid once per pass
#endif
#pragma unroll 1
        for (int rep = 0; rep < HARE_K2G_REPS; ++rep) {
        // ------------------------------------------------------------------ POP: the next node of every group that has no leaf in hand
        // The group's lanes look at the top eight entries at once; entries the state prunes (:210) are dropped, the first one it
        // does not prune is the node (an interior node: phase F below; a leaf: phase L).
        bool interior = false;
        int fc = 0, cmask = 255;
        double pa = 0, pb = 0;
        K2G_STAT(st_iter++; st_alive += __popcll(__ballot(alive)) >> 3;)
        {
            const bool popping = alive && q == qe && sp > 0;
            if (__ballot(popping)) {
                K2G_STAT(st_pop++; st_popg += __popcll(__ballot(popping)) >> 3;)
                double ea = 0, eb = 0;
                int2 ew = make_int2(0, 0);
                const int e = sp - 1 - j;
                const bool have = popping && e >= 0;
                if (have) st_load(e, ea, eb, ew);
                const bool keep = have && !(hit && closestT <= ea);                       // :210 with the state of the moment
                const unsigned km = gballot(keep);
                const int js = km ? __builtin_ctz(km) : 7;
                const double ta = __shfl(ea, js, W), tb = __shfl(eb, js, W);
                const int tw0 = __shfl(ew.x, js, W), tw1 = __shfl(ew.y, js, W);
                if (popping) {
                    if (km == 0) {
                        sp = sp > W ? sp - W : 0;                                         // all eight pruned
                    } else {
                        sp -= js + 1;
                        if (tw0 < 0) { q = -1 - tw0; qe = q + tw1; lca = ta; ++visit; }    // a leaf: its list, its nodeTmin
                        else { interior = true; fc = tw0; cmask = tw1; pa = ta; pb = tb; }
                    }
                }
            }
        }

        // ------------------------------------------------------------------ F: the eight children of an interior node, one per lane
        if (__ballot(interior)) {
            K2G_STAT(st_f++; st_fg += __popcll(__ballot(interior)) >> 3;)
            double ca = 0, cb = 0;
            int2 cw = make_int2(0, 0);
            bool push = false;
            auto fstep = [&](auto fast_tag) {
                constexpr bool FAST = decltype(fast_tag)::value;
                auto mx = [](double a, double b) { return FAST ? __builtin_fmax(a, b) : omax(a, b); };
                auto mn = [](double a, double b) { return FAST ? __builtin_fmin(a, b) : omin(a, b); };
                const OctNode& nd = g.nodes[fc + (j ^ mask)];                             // lane k examines order[k] = k ^ mask (:286-306)
                double tx0 = (nd.bmin[0] - o.x) * invDx, tx1 = (nd.bmax[0] - o.x) * invDx;     // :253-258
                double ty0 = (nd.bmin[1] - o.y) * invDy, ty1 = (nd.bmax[1] - o.y) * invDy;
                double tz0 = (nd.bmin[2] - o.z) * invDz, tz1 = (nd.bmax[2] - o.z) * invDz;
                if (invDx < 0) { const double s = tx0; tx0 = tx1; tx1 = s; }
                if (invDy < 0) { const double s = ty0; ty0 = ty1; ty1 = s; }
                if (invDz < 0) { const double s = tz0; tz0 = tz1; tz1 = s; }
                const double tmn = mx(mx(tx0, ty0), tz0), tmx = mn(mn(tx1, ty1), tz1);    // :265-266
                push = !(tmx < tmn || tmx < 0 || tmn > pb || tmx < pa);                   // :268
                ca = mx(tmn, pa);                                                         // :271
                cb = mn(tmx, pb);
                const int cfc = nd.first_child;
                const bool cleaf = cfc < 0;
                cw = cleaf ? make_int2(-1 - nd.item_start, nd.item_count) : make_int2(cfc, 255);
                push = push && ((cmask >> j) & 1) != 0;                                 // a frame K2p had opened: only the children it had not popped yet
                push = push && !(cb < ca || cb < 0);                                     // the pop test :207 is hit-independent: made here
                push = push && !(cleaf && cw.y == 0);                                    // popping an empty leaf has no effect
                push = push && !(hit && closestT <= ca);                                 // :210 true now stays true (closestT only falls)
                // the tight box of the child's subtree (as in K2d, kernels.hip): a ray that misses the box of ALL the polygons below --
                // or, holding a hit, reaches it behind that hit -- cannot make RayXtri accept any of them: the child is not pushed
                if (FAST && g.tight != nullptr) {
                    const float4* tp = reinterpret_cast<const float4*>(g.tight) + 2 * (size_t)(fc + (j ^ mask));
                    const float4 tb0 = tp[0], tb1 = tp[1];
                    double ux0 = ((double)tb0.x - o.x) * invDx, ux1 = ((double)tb0.w - o.x) * invDx;
                    double uy0 = ((double)tb0.y - o.y) * invDy, uy1 = ((double)tb1.x - o.y) * invDy;
                    double uz0 = ((double)tb0.z - o.z) * invDz, uz1 = ((double)tb1.y - o.z) * invDz;
                    if (invDx < 0) { const double s = ux0; ux0 = ux1; ux1 = s; }
                    if (invDy < 0) { const double s = uy0; uy0 = uy1; uy1 = s; }
                    if (invDz < 0) { const double s = uz0; uz0 = uz1; uz1 = s; }
                    const double un = mx(mx(ux0, uy0), uz0), uf = mn(mn(ux1, uy1), uz1);
                    push = push && !(tight_ok && ((uf < un) | (uf < 0) | (hit & (closestT <= un))));
                }
            };
            // Rays whose components are all finite and far from overflow never produce a NaN here, so for them Math.Max / Math.Min
            // are v_max_f64 / v_min_f64 (the sign of a zero result is only ever compared); anything else: NaN-propagating selects
            const bool all_tame = __ballot(interior && !tame) == 0;
            if (interior) {
                if (all_tame) fstep(std::true_type{});
                else fstep(std::false_type{});
            }
            const unsigned pm = gballot(interior && push);
            if (interior) {
                if (push) st_store(sp + __builtin_popcount(pm & jlt), ca, cb, cw);           // order[0] lowest ... order[7] on top: popped first
                sp += __builtin_popcount(pm);
            }
            lds_sync();
        }

        // ------------------------------------------------------------------ L: eight entries of the leaf in hand, pre-culled; survivors noted
        {
            const bool scanning = alive && q < qe && np <= kGPend - W;
            if (__ballot(scanning)) {
                const int k = q + j;
                const bool valid = scanning && k < qe;
                K2G_STAT(st_l++; st_lg += __popcll(__ballot(scanning)) >> 3; st_ent += __popcll(__ballot(valid));)
                int i = -1;
                if (valid) i = g.items[k];
                // :218 -- and not the polygon the ray's hit lies on AGAIN (HARE_K2D_SKIP_PID, kernels.hip: the same ray against the same polygon yields the same t,
                // which is never < closestT: nothing would change)
                bool test = valid && i != e1 && i != e2 && !(HARE_K2D_SKIP_PID && hit && i == pid);
                if (test) test = !cull_test(g, cray, cull_load(g, i));
                const unsigned sm = gballot(test);
                if (test) {
                    const int at = np + __builtin_popcount(sm & jlt);
                    plca[at] = lca;
                    ppw[at] = make_int2(i, visit);
                }
                if (scanning) {
                    np += __builtin_popcount(sm);
                    q = q + W < qe ? q + W : qe;
                }
                lds_sync();
            }
        }

        }

        // ------------------------------------------------------------------ E: exact tests of the pending survivors, replayed in order
        {
            const bool walk_over = alive && sp == 0 && q == qe;                            // nothing left to visit
            if (walk_over && np == 0) finish();
            const bool blocked = alive && np > 0 && (walk_over || np > kGPend - W);
            const int total = __popcll(__ballot(alive && j < np));                         // min(np, 8) summed over the groups
            if (__ballot(blocked) != 0 || total >= HARE_K2G_EXACT_MIN) {
                const int take = np < W ? np : W;
                const bool mine = alive && j < take;
                K2G_STAT(st_e++; st_el += __popcll(__ballot(mine));)
                double t = kDblMax, u = 0, v = 0, slca = 0;
                int spoly = -1, svisit = -1;
                if (mine) {
                    const int2 w = ppw[j];
                    spoly = w.x; svisit = w.y; slca = plca[j];
                    const PolyRec& p = g.polys[spoly];
                    const double* v3 = (g.quads && g.quads[spoly].nverts == 4) ? g.quads[spoly].v3 : nullptr;
                    double tt, uu, vv;
                    if (poly_full(p, v3, o, d, tt, uu, vv) && tt > kTMin) { t = tt; u = uu; v = vv; }      // :224
                }
                // replay, in list order, the survivors with a valid t (the others change nothing)
                unsigned vm = gballot(mine && t < kDblMax);
                K2G_STAT(st_valid += __popcll(__ballot(mine && t < kDblMax));)
                bool ended = false;
                while (__ballot(vm != 0)) {
                    const int k = vm ? __builtin_ctz(vm) : 0;
                    const double tk = __shfl(t, k, W), uk = __shfl(u, k, W), vk = __shfl(v, k, W), lk = __shfl(slca, k, W);
                    const int pk = __shfl(spoly, k, W), vis = __shfl(svisit, k, W);
                    if (vm != 0 && !ended) {
                        if (vis != cur_visit) { cur_visit = vis; cur_skip = hit && closestT <= lk; }      // :210 at that leaf's pop
                        if (!cur_skip && tk < closestT) {                                                 // :225
                            closestT = tk; bu = uk; bv = vk; pid = pk; hit = true;
                            if (closestT <= lk) ended = true;                                             // :233
                        }
                    }
                    vm &= vm - 1u;
                }
                if (alive) {
                    if (ended) {
                        finish();
                    } else {
                        // the survivors beyond the first eight move down
                        const int rest = np - take;
                        double mlca = 0;
                        int2 mw = make_int2(0, 0);
                        if (j < rest) { mlca = plca[W + j]; mw = ppw[W + j]; }
                        lds_sync();
                        if (j < rest) { plca[j] = mlca; ppw[j] = mw; }
                        np = rest;
                        if (np == 0 && sp == 0 && q == qe) finish();
                    }
                }
                lds_sync();
            }
        }
    }
#ifdef HARE_K2G_STATS
    if (wl == 0 && io.prof) {           // 12 words behind the counters block (the tool sizes the buffer)
        const unsigned long long v[12] = {st_iter, st_alive, st_pop, st_popg, st_f, st_fg, st_l, st_lg, st_ent, st_e, st_el, st_valid};
        for (int k = 0; k < 12; ++k) atomicAdd(&io.prof[k], v[k]);
    }
#endif
    if (TAIL) {
        // the rays were counted by K2p when it set them up; their hits are counted here.  The last wave of this grid leaves the hand-over
        // counters zeroed for the launch that uses the slot next.
        const unsigned long long wh = wave_sum_u32(nhits);          // valid in lane 0 of the wave
        if (wl == 0) {
            if (io.ctr && wh) atomicAdd(&io.ctr[CTR_HITS], wh);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (atomicAdd(&sm->oct_tail_done, 1u) == gridDim.x * waves_per_block - 1u) {
                atomicExch(&sm->oct_tail_count, 0u);
                atomicExch(&sm->oct_tail_next, 0u);
                atomicExch(&sm->oct_tail_done, 0u);
            }
        }
        return;
    }
    launch_epilogue(io, nrays, nhits, waves_per_block);
}

}  // namespace
