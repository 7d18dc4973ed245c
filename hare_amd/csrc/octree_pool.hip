// octree_pool.hip -- K2q: Octree.Shoot with MORE RAYS THAN LANES (included by kernels.hip; same idea as voxel_pool.hip).
//
// Why: K2p (hare_octree_persist) keeps one ray per lane and alternates, every round, a few child-box tests for the
// lanes that are descending with a few culls for the lanes that sit in a leaf; measured lane utilisation 17 % --
// the reference algorithm needs ~200 child tests and ~100 leaf candidates per ray, and the two populations starve
// each other.  Here a wave owns a pool of SLOTS rays and three queues of slot numbers:
//   desc   rays that have to look at a node: enter it (leaf -> its list; interior -> open a frame) and then test the
//          frame's children in the reference's pop order until one is accepted      ("Octree - alt.cs":203-272)
//   cull   rays inside a leaf with candidates for the conservative FP32 pre-cull
//   exact  rays whose candidate survived the cull: the reference's full RayXtri with u, v           (:224-237)
// Each round the wave pops up to 64 rays of ONE queue and runs that phase on them at (nearly) full lane occupancy.
// Pools and queues are private to a wave (no atomics, no barriers); queue heads and counts are wave-uniform scalars.
//
// Visit order.  As in K2p, the LIFO stack of (node, tmin, tmax) is replaced by one frame per level {interval of the
// frame's node, first_child, children left}: whether a child is pushed (:268) depends only on ray, child box and parent
// interval, never on the hit so far, so enumerating the children lazily from order[7] down to order[0] reproduces the
// reference's pop sequence, with the pop-time tests (:207-211) applied when the cursor reaches the child.  Child boxes
// are derived from the parent box with BuildOctree's own expressions (:96-111), as in K2p.
//
// State.  LDS per slot (96 B): closestT, the top frame {a, b, first_child|cursor, node}, the node accepted but not yet
// entered {node, a, b} / the current leaf {q, qe, idx, nexti, nodeTmin}, ray index, level + flags, a two-entry
// mailbox.  The frames BELOW the top one and 1/d live in a per-slot scratch block in device memory (cache-resident):
// they are touched once per push / pop, in the same batch of loads as the node record the task needs anyway.
// The best hit so far is the ray's own X_Event record, rewritten whenever a closer hit is accepted.
#ifndef HARE_K2Q_CULL_PAIRS
#define HARE_K2Q_CULL_PAIRS 2     // pairs of candidates per cull task
#endif
#ifndef HARE_K2Q_EXACT_MIN
#define HARE_K2Q_EXACT_MIN 32     // run the exact phase when this many rays wait for it (or nothing else can run)
#endif
#ifndef HARE_K2Q_REFILL_MIN
#define HARE_K2Q_REFILL_MIN 64    // set up new rays when this many slots are free
#endif
#ifndef HARE_K2Q_TAIL
#define HARE_K2Q_TAIL 128         // tickets dry and at most this many rays left: every non-empty phase runs each round
#endif

namespace {

__device__ __forceinline__ void octree_pool_body(const OctreeArgs& g, const ShootIO& io, unsigned char* scratch, unsigned scratch_stride)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    constexpr unsigned S = kOctPoolSlots, R = kOctPoolRing, SM = kOctPoolRing - 1;
    const int wave = threadIdx.x >> 6;
    const unsigned lane = threadIdx.x & 63;
    unsigned char* const wb = lds_raw + (size_t)wave * kOctPoolWaveBytes;
    double* const L_ct = reinterpret_cast<double*>(wb);      // closestT
    double* const L_fa = L_ct + S;                           // top frame: interval of its node
    double* const L_fb = L_fa + S;
    double* const L_pa = L_fb + S;                           // accepted node not yet entered: its interval; in a leaf: nodeTmin
    double* const L_pb = L_pa + S;
    uint32_t* const L_ray = reinterpret_cast<uint32_t*>(L_pb + S);
    int32_t* const L_fpk = reinterpret_cast<int32_t*>(L_ray + S);      // top frame: new << 31 | first_child << 8 | children still to look at (by cursor position)
    int32_t* const L_fnode = L_fpk + S;                                // top frame: its node
    int32_t* const L_pnode = L_fnode + S;                              // accepted node not yet entered
    uint32_t* const L_q = reinterpret_cast<uint32_t*>(L_pnode + S);
    uint32_t* const L_qe = L_q + S;
    int32_t* const L_idx = reinterpret_cast<int32_t*>(L_qe + S);
    int32_t* const L_nexti = L_idx + S;
    uint32_t* const L_fl = reinterpret_cast<uint32_t*>(L_nexti + S);   // (level + 1) | mask << 8 | flags
    int32_t* const L_m0 = reinterpret_cast<int32_t*>(L_fl + S);
    int32_t* const L_m1 = L_m0 + S;
    uint8_t* const Q_desc = reinterpret_cast<uint8_t*>(L_m1 + S);
    uint8_t* const Q_cull = Q_desc + R;
    uint8_t* const Q_exact = Q_cull + R;
    uint8_t* const Q_free = Q_exact + R;
    constexpr uint32_t F_NEW = 1u << 16;      // L_pnode / L_pa / L_pb hold a node that was accepted and is entered by the next desc task
    constexpr uint32_t F_HIT = 1u << 17;      // a hit has been accepted (closestT, and the ray's X_Event record, hold it)

    for (unsigned k = lane; k < S; k += 64) Q_free[k] = (uint8_t)k;
    // nothing in LDS is shared between waves: no barrier

    // per-slot scratch in device memory: 1/d (3 doubles), then one 24-byte frame per level below the top one
    struct Frame { double a, b; int32_t pk, node; };
    unsigned char* const wscr = scratch + ((size_t)(blockIdx.x * (unsigned)kOctPoolWaves + (unsigned)wave) * S) * scratch_stride;
    auto scr_inv = [&](unsigned slot) { return reinterpret_cast<double*>(wscr + (size_t)slot * scratch_stride); };
    auto scr_frame = [&](unsigned slot, int lvl) { return reinterpret_cast<Frame*>(wscr + (size_t)slot * scratch_stride + 24) + lvl; };

    unsigned hD = 0, nD = 0, hC = 0, nC = 0, hE = 0, nE = 0, hF = 0, nF = S;
    auto push = [&](uint8_t* Q, unsigned head, unsigned& cnt, bool flag, unsigned slot) {
        const unsigned long long m = __ballot(flag);
        if (flag) Q[(head + cnt + rank_below(m)) & SM] = (uint8_t)slot;
        cnt += (unsigned)__popcll(m);
    };
    auto pop = [&](const uint8_t* Q, unsigned& head, unsigned& cnt, bool& active) -> unsigned {
        const unsigned n = cnt < 64u ? cnt : 64u;
        active = lane < n;
        const unsigned slot = Q[(head + (active ? lane : 0u)) & SM];
        head = (head + n) & SM;
        cnt -= n;
        return slot;
    };

    const unsigned n32 = (unsigned)io.n;
    const unsigned RAY_CHUNK = 128;
    const unsigned n_static = gridDim.x * (unsigned)kOctPoolWaves * RAY_CHUNK;
    unsigned chunk_id = blockIdx.x * (unsigned)kOctPoolWaves + (unsigned)wave;
    if ((gridDim.x & 7u) == 0) chunk_id = ((blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3)) * (unsigned)kOctPoolWaves + (unsigned)wave;
    unsigned cn = chunk_id * RAY_CHUNK, ce = cn + RAY_CHUNK;
    if (cn > n32) cn = n32;
    if (ce > n32) ce = n32;
    bool drained = false;
    unsigned nhits = 0, nrays = 0;

    auto store_miss = [&](unsigned ray) {
        XEventRec ev;
        set_miss(ev);
        store_event_streaming(&io.out[ray], ev);
    };

    for (unsigned round = 0; round < (1u << 26); ++round) {     // the cap only bounds a defect (see voxel_pool.hip)
        // ------------------------------------------------------------------ set-up of new rays into free slots
        if (!drained && (nF >= (unsigned)HARE_K2Q_REFILL_MIN || nD + nC + nE == 0)) {
            if (cn >= ce) {
                unsigned base = 0;
                const unsigned dyn = (unsigned)io.ticket_rays;
                if (lane == 0) base = atomicAdd(io.work, dyn);
                base = __shfl(base, 0, 64);
                cn = base + n_static;
                if (cn >= n32) { drained = true; cn = ce = n32; }
                else ce = (n32 - cn > dyn) ? cn + dyn : n32;
            }
            unsigned m = ce - cn;
            if (m > 64u) m = 64u;
            if (m > nF) m = nF;
            if (m > 0) {
                const bool act = lane < m;
                const unsigned slot = Q_free[(hF + (act ? lane : 0u)) & SM];
                hF = (hF + m) & SM;
                nF -= m;
                const unsigned ray = cn + lane;
                cn += m;
                bool to_desc = false, freed = false;
                if (act) {
                    const RayRec r = io.rays[ray];
                    if ((io.flags & SHOOT_RETIRED_RAYS) && io.excl1 && io.excl1[ray] == -2) {   // retired by the bounce loop: miss, not counted
                        store_miss(ray);
                        freed = true;
                    } else {
                        nrays++;
                        const double invDx = fabs(r.dx) > 1e-16 ? 1.0 / r.dx : 1e16;      // "Octree - alt.cs":165-167
                        const double invDy = fabs(r.dy) > 1e-16 ? 1.0 / r.dy : 1e16;
                        const double invDz = fabs(r.dz) > 1e-16 ? 1.0 / r.dz : 1e16;
                        const uint32_t mask = ((r.dx >= 0 ? 0u : 1u) << 2) | ((r.dy >= 0 ? 0u : 1u) << 1) | (r.dz >= 0 ? 0u : 1u);
                        const OctNode& root = g.nodes[0];
                        double tx0 = (root.bmin[0] - r.x) * invDx, tx1 = (root.bmax[0] - r.x) * invDx;
                        double ty0 = (root.bmin[1] - r.y) * invDy, ty1 = (root.bmax[1] - r.y) * invDy;
                        double tz0 = (root.bmin[2] - r.z) * invDz, tz1 = (root.bmax[2] - r.z) * invDz;
                        if (invDx < 0) { const double s = tx0; tx0 = tx1; tx1 = s; }
                        if (invDy < 0) { const double s = ty0; ty0 = ty1; ty1 = s; }
                        if (invDz < 0) { const double s = tz0; tz0 = tz1; tz1 = s; }
                        const double rmin = omax(omax(tx0, ty0), tz0), rmax = omin(omin(tx1, ty1), tz1);   // :182-183
                        if (rmax < rmin || rmax < 0) {                       // :185 (and the identical pop test :207)
                            store_miss(ray);
                            freed = true;
                        } else {
                            double* iv = scr_inv(slot);
                            iv[0] = invDx; iv[1] = invDy; iv[2] = invDz;
                            L_ct[slot] = kDblMax;
                            L_ray[slot] = ray;
                            L_pnode[slot] = 0;                               // the root is entered by the first desc task
                            L_pa[slot] = rmin; L_pb[slot] = rmax;
                            L_fl[slot] = 0u | (mask << 8) | F_NEW;           // level + 1 = 0: no frame open
                            L_m0[slot] = -1; L_m1[slot] = -1;
                            L_fpk[slot] = 0; L_fnode[slot] = 0;
                            to_desc = true;
                        }
                    }
                }
                push(Q_desc, hD, nD, to_desc, slot);
                push(Q_free, hF, nF, freed, slot);
            }
        }
        if (nD + nC + nE == 0) {
            if (drained) break;
            continue;
        }

        // ------------------------------------------------------------------ pick the phase(s) for this round
        const unsigned big = nD > nC ? nD : nC;
        const bool tail = drained && nD + nC + nE <= (unsigned)HARE_K2Q_TAIL;
        const int sel = (nE >= (unsigned)HARE_K2Q_EXACT_MIN || big == 0) ? 0 : (nC >= nD ? 1 : 2);
        if (tail ? nD > 0 : sel == 2) {
            // -------------------------------------------------------------- enter a node / test the children of the top frame
            bool act;
            const unsigned slot = pop(Q_desc, hD, nD, act);
            bool to_desc = false, to_cull = false, freed = false;
            if (act) {
                uint32_t fl = L_fl[slot];
                const unsigned ray = L_ray[slot];
                const double closestT = L_ct[slot];
                int lvl1 = (int)(fl & 255u);                                // level + 1 (0: no frame open)
                const uint32_t mask = (fl >> 8) & 7u;
                const bool isnew = (fl & F_NEW) != 0;
                const bool hit = (fl & F_HIT) != 0;
                double fa = L_fa[slot], fb = L_fb[slot];
                int fpk = L_fpk[slot], fnode = L_fnode[slot];
                const int pnode = L_pnode[slot];
                const double pa_new = L_pa[slot], pb_new = L_pb[slot];
                // one batch of loads: the node's record, the ray origin, 1/d
                const OctNode nd = g.nodes[isnew ? pnode : fnode];
                const double* iv = scr_inv(slot);
                const double invDx = iv[0], invDy = iv[1], invDz = iv[2];
                const RayRec r = io.rays[ray];
                bool searching = true;
                if (isnew) {
                    fl &= ~F_NEW;
                    if (nd.first_child < 0) {
                        // a leaf (:213): its list; nodeTmin for the early return of :233
                        const unsigned q = (unsigned)nd.item_start, qe = q + (unsigned)nd.item_count;
                        L_q[slot] = q; L_qe[slot] = qe;
                        if (q < qe) {
                            L_idx[slot] = g.items[q];
                            L_nexti[slot] = g.items[q + 1 < qe ? q + 1 : q];
                            to_cull = true;
                        } else {
                            to_desc = true;                                  // empty leaf: back to the frame
                        }
                        searching = false;
                    } else {
                        // an interior node: the frame that was on top goes to the scratch block, this one becomes the top
                        if (lvl1 > 0) {
                            Frame* f = scr_frame(slot, lvl1 - 1);
                            f->a = fa; f->b = fb; f->pk = fpk; f->node = fnode;
                        }
                        ++lvl1;
                        fa = pa_new; fb = pb_new; fnode = pnode;
                        fpk = (int)(((uint32_t)nd.first_child << 8) | 0x800000FFu);   // sign bit: the children have not been looked at yet
                    }
                } else if (lvl1 == 0) {
                    searching = false;                                      // the root was a leaf and is done: stack empty
                    freed = true;
                }
                // Rays whose components are all finite and far from overflow never produce a NaN below (1/d is finite and
                // non-zero, boxes are finite), so Math.Max / Math.Min reduce to the hardware's v_max_f64 / v_min_f64 for them
                // (the sign of a zero result is only ever compared).  Anything else takes the NaN-propagating compare-selects.
                const bool tame = fabs(r.x) < 1e300 && fabs(r.y) < 1e300 && fabs(r.z) < 1e300 &&
                                  fabs(r.dx) < 1e300 && fabs(r.dy) < 1e300 && fabs(r.dz) < 1e300;
                const bool all_tame = __ballot(searching && !tame) == 0;
                auto descend = [&](auto fast_tag) {
                    constexpr bool FAST = decltype(fast_tag)::value;
                    auto mx = [](double a, double b) { return FAST ? __builtin_fmax(a, b) : omax(a, b); };
                    auto mn = [](double a, double b) { return FAST ? __builtin_fmin(a, b) : omin(a, b); };
                    // child planes of the frame's node and their ray parameters ("Octree - alt.cs":96-111, :253-263):
                    // n[axis][half] / f[axis][half] = entry / exit parameter of the low (0) and high (1) child slab
                    double nx[2], fx[2], ny[2], fy[2], nz[2], fz[2];
                    {
                        const double c = (nd.bmax[0] + nd.bmin[0]) / 2;
                        const double a0 = ((nd.bmin[0] - 0.1) - r.x) * invDx, a1 = ((c + 0.1) - r.x) * invDx;
                        const double b0 = ((c - 0.1) - r.x) * invDx, b1 = ((nd.bmax[0] + 0.1) - r.x) * invDx;
                        const bool neg = invDx < 0;
                        nx[0] = neg ? a1 : a0; fx[0] = neg ? a0 : a1; nx[1] = neg ? b1 : b0; fx[1] = neg ? b0 : b1;
                    }
                    {
                        const double c = (nd.bmax[1] + nd.bmin[1]) / 2;
                        const double a0 = ((nd.bmin[1] - 0.1) - r.y) * invDy, a1 = ((c + 0.1) - r.y) * invDy;
                        const double b0 = ((c - 0.1) - r.y) * invDy, b1 = ((nd.bmax[1] + 0.1) - r.y) * invDy;
                        const bool neg = invDy < 0;
                        ny[0] = neg ? a1 : a0; fy[0] = neg ? a0 : a1; ny[1] = neg ? b1 : b0; fy[1] = neg ? b0 : b1;
                    }
                    {
                        const double c = (nd.bmax[2] + nd.bmin[2]) / 2;
                        const double a0 = ((nd.bmin[2] - 0.1) - r.z) * invDz, a1 = ((c + 0.1) - r.z) * invDz;
                        const double b0 = ((c - 0.1) - r.z) * invDz, b1 = ((nd.bmax[2] + 0.1) - r.z) * invDz;
                        const bool neg = invDz < 0;
                        nz[0] = neg ? a1 : a0; fz[0] = neg ? a0 : a1; nz[1] = neg ? b1 : b0; fz[1] = neg ? b0 : b1;
                    }
                    const int first = (fpk >> 8) & 0x7FFFFF;                // 23 bits: the host keeps trees with more nodes on K2p
                    unsigned rem = (unsigned)fpk & 255u;                    // children (by cursor position) still to be looked at
                    if (fpk < 0) {
                        // first look at this frame: the push test of :268 for all eight children at once (it depends on the ray,
                        // the child box and the frame's interval only), kept as a mask over cursor positions
                        double nxy[2][2], fxy[2][2];
#pragma unroll
                        for (int i = 0; i < 2; ++i)
#pragma unroll
                            for (int j = 0; j < 2; ++j) { nxy[i][j] = mx(nx[i], ny[j]); fxy[i][j] = mn(fx[i], fy[j]); }
                        unsigned pushed = 0;
#pragma unroll
                        for (int oct = 0; oct < 8; ++oct) {
                            const double tmn = mx(nxy[(oct >> 2) & 1][(oct >> 1) & 1], nz[oct & 1]);
                            const double tmx = mn(fxy[(oct >> 2) & 1][(oct >> 1) & 1], fz[oct & 1]);
                            const bool p = !(tmx < tmn || tmx < 0 || tmn > fb || tmx < fa);
                            pushed |= p ? (1u << oct) : 0u;
                        }
                        // octant bit -> cursor bit: cursor k examines octant k ^ mask
                        unsigned byc = 0;
#pragma unroll
                        for (int k = 0; k < 8; ++k) byc |= ((pushed >> (k ^ (int)mask)) & 1u) << k;
                        rem = byc;
                    }
                    bool found = false;
                    int cnode = 0;
                    double ca = 0, cb = 0;
#pragma unroll 1
                    while (rem != 0 && !found) {
                        const int cur = 31 - __builtin_clz(rem);            // children pop from order[7] down to order[0] (:286-306)
                        rem &= ~(1u << cur);
                        const int oct = cur ^ (int)mask;
                        const double tmn = mx(mx(nx[(oct >> 2) & 1], ny[(oct >> 1) & 1]), nz[oct & 1]);
                        const double tmx = mn(mn(fx[(oct >> 2) & 1], fy[(oct >> 1) & 1]), fz[oct & 1]);
                        ca = mx(tmn, fa); cb = mn(tmx, fb);                                       // :271
                        if (!(cb < ca || cb < 0) && !(hit && closestT <= ca)) {                   // popped and kept (:207-211)
                            cnode = first + oct;
                            found = true;
                        }
                    }
                    fpk = (first << 8) | (int)rem;
                    if (found) {
                        L_pnode[slot] = cnode; L_pa[slot] = ca; L_pb[slot] = cb;
                        fl |= F_NEW;
                        to_desc = true;
                    } else {
                        // frame exhausted: back to the parent frame (or the stack is empty: :276-283)
                        --lvl1;
                        if (lvl1 == 0) {
                            freed = true;
                        } else {
                            const Frame* f = scr_frame(slot, lvl1 - 1);
                            fa = f->a; fb = f->b; fpk = f->pk; fnode = f->node;
                            to_desc = true;
                        }
                    }
                };
                if (searching) {
                    if (all_tame) descend(std::true_type{});
                    else descend(std::false_type{});
                }
                if (freed) {
                    if (hit) nhits++;                                        // the record was written when the hit was accepted
                    else store_miss(ray);
                } else {
                    L_fa[slot] = fa; L_fb[slot] = fb; L_fpk[slot] = fpk; L_fnode[slot] = fnode;
                    if (to_cull) L_pa[slot] = pa_new;                        // nodeTmin of the leaf just entered
                    L_fl[slot] = (fl & ~255u) | (uint32_t)lvl1;
                }
            }
            push(Q_desc, hD, nD, to_desc, slot);
            push(Q_cull, hC, nC, to_cull, slot);
            push(Q_free, hF, nF, freed, slot);
        }
        if (tail ? nC > 0 : sel == 1) {
            // -------------------------------------------------------------- FP32 pre-cull of leaf candidates, 2 x CULL_PAIRS per ray at most
            bool act;
            const unsigned slot = pop(Q_cull, hC, nC, act);
            bool to_desc = false, to_cull = false, to_exact = false;
            if (act) {
                const unsigned ray = L_ray[slot];
                unsigned q = L_q[slot];
                const unsigned qe = L_qe[slot];
                int idx = L_idx[slot], nexti = L_nexti[slot], m0 = L_m0[slot], m1 = L_m1[slot];
                int e1 = -1, e2 = -1;
                if (io.excl1) e1 = io.excl1[ray];                           // :218
                if (io.excl2) e2 = io.excl2[ray];
                const RayRec r = io.rays[ray];
                const CullRay cray = cull_ray(g, r.x, r.y, r.z, r.dx, r.dy, r.dz);
                bool culling = true, parked = false;
#pragma unroll
                for (int kp = 0; kp < HARE_K2Q_CULL_PAIRS; ++kp) {
                    const bool has1 = q + 1 < qe;
                    const unsigned qa = q + 2 < qe ? q + 2 : qe - 1, qb = q + 3 < qe ? q + 3 : qe - 1;
                    const int i2 = g.items[qa], i3 = g.items[qb];
                    const int ia = idx >= 0 ? idx : 0, ib = (has1 && nexti >= 0) ? nexti : ia;
                    const CullRaw ra = cull_load(g, ia), rb = cull_load(g, ib);
                    const bool ca = cull_test(g, cray, ra);
                    const bool cb = cull_test(g, cray, rb);
                    // Not in the reference (its mailbox is commented out, :221-222): loose leaves overlap, so a ray meets the
                    // same polygon in several leaves; skipping one it has just tested, and candidates the conservative cull
                    // proves to be misses, cannot change any accepted hit (strict `t < closestT`)
                    const bool sk0 = idx == e1 || idx == e2 || idx == m0 || idx == m1;
                    const bool keep0 = culling && !sk0 && !ca;
                    const bool step0 = culling && !keep0;
                    const bool tested0 = culling && !sk0;
                    const int n0 = tested0 ? idx : m0, n1 = tested0 ? m0 : m1;            // mailbox after candidate 0
                    const bool go1 = step0 && has1;
                    const bool sk1 = nexti == e1 || nexti == e2 || nexti == n0 || nexti == n1;
                    const bool keep1 = go1 && !sk1 && !cb;
                    const bool step1 = go1 && !keep1;
                    const bool tested1 = go1 && !sk1;
                    m0 = tested1 ? nexti : n0;
                    m1 = tested1 ? n0 : n1;
                    q += (step0 ? 1u : 0u) + (step1 ? 1u : 0u);
                    const int nidx = step1 ? i2 : (step0 ? nexti : idx);
                    const int nnext = step1 ? i3 : (step0 ? i2 : nexti);
                    idx = nidx; nexti = nnext;
                    parked = parked || keep0 || keep1;
                    culling = culling && !keep0 && !keep1 && q < qe;
                }
                L_q[slot] = q; L_idx[slot] = idx; L_nexti[slot] = nexti; L_m0[slot] = m0; L_m1[slot] = m1;
                to_exact = parked;
                to_cull = culling;
                to_desc = !parked && !culling;                              // leaf exhausted: back to the frame
            }
            push(Q_desc, hD, nD, to_desc, slot);
            push(Q_cull, hC, nC, to_cull, slot);
            push(Q_exact, hE, nE, to_exact, slot);
        }
        if (tail ? nE > 0 : sel == 0) {
            // -------------------------------------------------------------- exact test with u, v of one candidate per ray (:224-237)
            bool act;
            const unsigned slot = pop(Q_exact, hE, nE, act);
            bool to_desc = false, to_cull = false, freed = false;
            if (act) {
                const unsigned ray = L_ray[slot];
                const int i = L_idx[slot];
                unsigned q = L_q[slot];
                const unsigned qe = L_qe[slot];
                const double closestT = L_ct[slot];
                const unsigned qa = q + 2 < qe ? q + 2 : qe - 1;
                const int after = g.items[qa];
                const RayRec r = io.rays[ray];
                const PolyRec& p = g.polys[i];
                const double* v3 = (g.quads && g.quads[i].nverts == 4) ? g.quads[i].v3 : nullptr;
                const V3 o = {r.x, r.y, r.z};
                const V3 d = {r.dx, r.dy, r.dz};
                double t, u, v;
                ++q;
                L_idx[slot] = L_nexti[slot];
                L_nexti[slot] = after;
                L_q[slot] = q;
                if (poly_full(p, v3, o, d, t, u, v) && t > kTMin && t < closestT) {   // :224-226
                    XEventRec ev;
                    ev.t = t; ev.u = u; ev.v = v;
                    ev.x = o.x + d.x * t; ev.y = o.y + d.y * t; ev.z = o.z + d.z * t;
                    ev.poly_id = i;
                    ev.hit = 1;
                    io.out[ray] = ev;
                    L_ct[slot] = t;
                    L_fl[slot] |= F_HIT;
                    if (t <= L_pa[slot]) {                                   // :233 closestT <= nodeTmin: the reference returns at once
                        freed = true;
                        nhits++;
                    }
                }
                if (!freed) {
                    to_cull = q < qe;
                    to_desc = !to_cull;
                }
            }
            push(Q_desc, hD, nD, to_desc, slot);
            push(Q_cull, hC, nC, to_cull, slot);
            push(Q_free, hF, nF, freed, slot);
        }
    }

    launch_epilogue(io, nrays, nhits, (unsigned)kOctPoolWaves);     // batch counters + the launch slot left zeroed
}

}  // namespace

extern "C" {
// K2q: dynamic LDS = waves x kOctPoolWaveBytes; scratch = grid x waves x slots x scratch_stride bytes of device memory
__global__ __launch_bounds__(64 * HARE_K2Q_WAVES) void hare_octree_pool(OctreeArgs g, ShootIO io, unsigned char* scratch, unsigned scratch_stride)
{
    octree_pool_body(g, io, scratch, scratch_stride);
}
}
