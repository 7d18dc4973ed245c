// voxel_pool.hip -- K1q: Voxel_Grid.Shoot with MORE RAYS THAN LANES (included by kernels.hip).
//
// Why: in K1p (hare_voxel_persist_*) a lane owns one ray, and a ray alternates between walking (~10 empty voxels),
// culling (~3 candidates) and, rarely, an exact test -- so whatever phase the wave executes, about half of its
// lanes hold a ray that is in another phase (measured: 40 % lane occupancy, 2.9x the instructions the work needs).
// Here a wave owns a POOL of SLOTS rays (2 per lane) whose traversal state lives in LDS, and queues of slot numbers:
//   walk   rays that cross empty voxels until the next non-empty one (no hit pending)
//   cull   rays that hold candidates for the conservative FP32 pre-cull
//   exact  rays whose candidate survived the cull and needs the reference's FP64 RayXtri
//   pend   rays that hold a hit and walk on until a voxel contains the hit point (Voxel_Grid.cs:705), or the grid ends
// Each round the wave picks a queue, pops up to 64 rays and runs that ONE phase on them at (nearly) full lane
// occupancy; rays move between queues as their phase changes, finished rays free their slot and new rays are set up
// 64 at a time.  Pools and queues are private to a wave: no atomics, no barriers, no inter-wave protocol -- queue
// heads and counts are wave-uniform scalars.
//
// A ray's own sequence of operations is exactly K1p's (and therefore the reference's, Voxel_Grid.cs:561-761): same
// cells in the same order, same candidate order, FP32 pre-cull in front of the exact RayXtri, pending-hit /
// IsPointInBox rule, miss on grid exit; only the interleaving BETWEEN rays differs.
//
// State.  LDS per slot, 81 B: tMax, tDelta (6 doubles), ray index, packed voxel + flags, q, qe, idx, nexti, the last
// polygon tested (7 words), one byte in each of the five queues.  o and d are not kept: cull and exact re-read the
// 48-byte ray record (cache-resident for the ray's short life) in the same batch of loads as the polygon records.
// The ray's own X_Event slot is its scratch until it finishes: .t = tmin of the hit so far (DBL_MAX: none),
// .x .y .z = that hit's point, the {Hit, Poly_id} word = its polygon, .u = t_start of a ray whose origin
// AABB.Intersect moved (the moved origin itself is o + d * t_start, the expression of AABB_Main.cs:254-256).
// INVARIANT: that scratch (and, with origin write-back, the ray record) is written by one lane of the wave with plain stores
// and read in a later phase by whichever lane pops the ray -- a store -> load hand-over between LANES OF ONE WAVEFRONT, never between
// waves.  What it rests on (LLVM AMDGPU backend user guide, "Memory Model", the GFX90A / GFX942 family sections, which gfx950 shares):
//   * the hardware description there: every CU has ONE vector L1, write-through, and a wavefront's vector-memory operations are
//     performed in order through it; "no special action is required for coherence between the lanes of a single wavefront, or for
//     coherence between wavefronts in the same work-group" (work-groups not in tgsplit mode -- these kernels are built without
//     -mtgsplit);
//   * the code-sequence table of the same section: `fence acq_rel` at syncscope "wavefront" expands to NO instruction (no s_waitcnt,
//     no buffer_inv / buffer_wbl2), i.e. program order of the wave's own loads and stores is all the memory model asks for there.
// So the fence in front of every phase (HARE_K1Q_PHASE_FENCE) is there for the COMPILER -- it may not move the scratch loads of a
// phase above the scratch stores of the previous one, nor keep a stale copy in registers -- and costs nothing at run time.  The
// loads that follow are ordinary global_loads, and the s_waitcnt vmcnt the compiler puts in front of their first use is the only
// wait involved; the earlier store needs none of its own, because the later load of the same wave queues behind it in the same L1.
// The invariant holds only while no OTHER wave touches those bytes (another CU's L1 is never refreshed by this one's stores: that
// would need the agent-scope forms): the host refuses calls whose rays / events / exclusion buffers overlap (launch.cpp), and a ray
// belongs to exactly one wave from its set-up to its final event.
#ifndef HARE_K1Q_WALK_STEPS
#define HARE_K1Q_WALK_STEPS 16    // DDA steps per walk task at most
#endif
#ifndef HARE_K1Q_WALK_MIN
#define HARE_K1Q_WALK_MIN 20      // a walk task ends early when fewer lanes than this are still walking ...
#endif
#ifndef HARE_K1Q_WALK_DIV
#define HARE_K1Q_WALK_DIV 3       // ... or than this fraction of the lanes it started with, whichever is less
#endif
#ifndef HARE_K1Q_CULL_PAIRS
#define HARE_K1Q_CULL_PAIRS 4     // pairs of candidates per cull task (swept 1..8: DESIGN.md section 9)
#endif
#ifndef HARE_K1Q_CULL_AHEAD
#define HARE_K1Q_CULL_AHEAD 1     // a cull task requests all its list entries, then all its records, then scans (0: pair by pair, each pair's entries from the previous pair's loads)
#endif
#ifndef HARE_K1Q_EXACT_MIN
#define HARE_K1Q_EXACT_MIN 24     // run the exact phase when this many rays wait for it (or nothing else can run); swept 8..56
#endif
#ifndef HARE_K1Q_PEND_MIN
#define HARE_K1Q_PEND_MIN 16      // the same for the pending-hit walk
#endif
#ifndef HARE_K1Q_TAIL
#define HARE_K1Q_TAIL 128         // tickets dry and at most this many rays left: every non-empty phase runs each round
#endif
#ifndef HARE_K1Q_TAIL_STEPS
#define HARE_K1Q_TAIL_STEPS 32    // DDA steps per walk task in that regime
#endif
#ifndef HARE_K1Q_MAILBOX
#define HARE_K1Q_MAILBOX 1        // skip the polygon this ray tested last
#endif
#ifndef HARE_K1Q_COOP_NOW
#define HARE_K1Q_COOP_NOW 2       // tickets dry and this few rays left: the wave traces them cooperatively at once (voxel_coop.hip)
#define HARE_K1Q_COOP_MAX 8       // ... or this few, once they have outlived the rest of the batch by
#define HARE_K1Q_COOP_PATIENCE 48 // this many rounds (heavy rays)
#endif
#ifndef HARE_K1Q_WIDE_MAX
#define HARE_K1Q_WIDE_MAX 64      // tickets dry and at most this many rays left (<= 64): the cull runs WIDE -- up to 16 lanes per ray, four candidates per
#endif                            // lane, their list entries and records requested together (two dependent round trips per task instead of five)
#ifndef HARE_K1Q_WIDE_WALK
#define HARE_K1Q_WIDE_WALK 1      // ... and the walk looks several occupied voxels ahead, one per lane of the ray's group
#endif
#ifndef HARE_K1Q_WALK_SEGS
#define HARE_K1Q_WALK_SEGS 3      // segments of steps per walk task at most: lanes that an occupied voxel sends on (empty list, tight box) walk on in the same task ...
#define HARE_K1Q_WALK_RESUME_MIN 8   // ... when at least this many of the task's lanes were sent on
#endif
#ifndef HARE_K1Q_HAND_WALK
#define HARE_K1Q_HAND_WALK 1      // the DDA step loop of the walk phases as written by hand (voxel_walk.h); 0: the compiler's (A/B)
#endif
#ifndef HARE_K1Q_REFILL_MIN
#define HARE_K1Q_REFILL_MIN 64    // set up new rays when this many slots are free (a full wave of set-ups)
#endif

namespace {

#ifdef HARE_K1Q_STATS
constexpr bool kK1qStats = true;
#else
constexpr bool kK1qStats = false;
#endif

// BOUNCE (hare_voxel_bounce_*, round 4): the whole specular bounce loop of a ray inside ONE launch.  Rays are independent, also across
// casts: ray i's cast c + 1 needs nothing but ray i's cast c.  The launch-per-cast loop puts a chip-wide barrier between casts and
// pays the end of a launch -- a quarter of a late cast in the cathedral -- once per cast.  Here a ray whose hit stands is not
// freed: its slot goes to a fifth queue, `rearm`, whose phase reflects the ray about the polygon's normal (the expressions of
// hare_reflect), writes the reflected ray and the polygon it left into the caller's work arrays (io.rays, io.excl1) and sets the
// slot up again -- Voxel_Grid.cs:563-632, the same code as for a new ray.  A ray that misses dies and frees its slot: no retired
// rays are carried, no packing.  Per ray the sequence of operations is that of the launch-per-cast loop; the events of every cast
// are identical to it (tests).  The event slot io.out[ray] is the ray's scratch through all its casts and ends up holding the last
// cast's X_Event (a miss record once the ray has died); io.out_all, if given, receives every cast's final event.
//
// OWN (hare_voxel_pool_*_own, round 5; flag HARE_SHOOT_COUNT_OWN): the same kernel counting ITS OWN work per lane -- voxels it walked into,
// list entries it scanned, candidates it pre-culled, exact tests it made -- into words 2 .. 5 of the counters block: the numerator of
// bench.py's `roofline.own` (the reference-priced figure counts voxels, entries and tests this kernel skips).  Events are identical.
template <bool QUADS, bool COARSE, bool BOUNCE = false, bool OWN = false>
__device__ __forceinline__ void voxel_pool_body(const VoxelArgs& g_in, const ShootIO& io_in)
{
    VoxelArgs g = g_in;
    ShootIO io = io_in;
    pin_args(g);                     // hare_device.h: every argument a scalar of its own
    pin_args(io);
    OwnWork own;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    constexpr unsigned S = kPoolSlots, R = kPoolRing, SM = kPoolRing - 1;   // slots; queue (ring) capacity and its mask
    uint32_t* const locc = reinterpret_cast<uint32_t*>(lds_raw);
    // the bitmap's byte address in LDS, for the hand-written step loop (voxel_walk.h)
    const unsigned lds_bitmap = (unsigned)(unsigned long long)(__attribute__((address_space(3))) unsigned char*)lds_raw;
    const int nw4 = (g.occ_words + 3) >> 2;
    {
        const uint4* src = reinterpret_cast<const uint4*>(g.occ);
        uint4* dst = reinterpret_cast<uint4*>(locc);
        for (int k = threadIdx.x; k < nw4; k += blockDim.x) dst[k] = src[k];
    }
    // wave-uniform BY CONSTRUCTION, and said so to the compiler (readfirstlane): everything derived from the wave's number -- its ray chunk,
    // `drained`, through them every queue head and count -- would otherwise count as divergent and live in vector registers, each `if`
    // on it a v_cmp + saveexec + branch (round 6: that bookkeeping was two fifths of a wave's time)
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const unsigned lane = threadIdx.x & 63;
    unsigned char* const wb = lds_raw + ((size_t)nw4 << 4) + (size_t)wave * kPoolWaveBytes;
    double* const L_tmx = reinterpret_cast<double*>(wb);
    double* const L_tmy = L_tmx + S;
    double* const L_tmz = L_tmy + S;
    double* const L_tdx = L_tmz + S;
    double* const L_tdy = L_tdx + S;
    double* const L_tdz = L_tdy + S;
    uint32_t* const L_ray = reinterpret_cast<uint32_t*>(L_tdz + S);
    uint32_t* const L_xyzf = L_ray + S;       // X | Y << 9 | Z << 18 | flags (below)
    uint32_t* const L_q = L_xyzf + S;
    uint32_t* const L_qe = L_q + S;
    int32_t* const L_idx = reinterpret_cast<int32_t*>(L_qe + S);
    int32_t* const L_nexti = L_idx + S;
    int32_t* const L_d1 = L_nexti + S;
    uint8_t* const Q_walk = reinterpret_cast<uint8_t*>(L_d1 + S);
    uint8_t* const Q_cull = Q_walk + R;
    uint8_t* const Q_exact = Q_cull + R;
    uint8_t* const Q_pend = Q_exact + R;
    uint8_t* const Q_free = Q_pend + R;
    // BOUNCE: per wave, behind the twelve pools: the rearm queue, each slot's cast number, rays started / hits per cast
    unsigned char* const xb = lds_raw + ((size_t)nw4 << 4) + (size_t)kPoolWaves * kPoolWaveBytes + (size_t)wave * kPoolBounceExtra;
    uint8_t* const Q_rearm = xb;
    uint8_t* const L_cast = Q_rearm + R;
    uint32_t* const C_rays = reinterpret_cast<uint32_t*>(L_cast + S);
    uint32_t* const C_hits = C_rays + kBounceMaxCasts;
    const int n_casts = BOUNCE ? io.bounce_casts : 1;
    if (BOUNCE) for (unsigned k = lane; k < 2u * (unsigned)kBounceMaxCasts; k += 64) C_rays[k] = 0u;
    constexpr uint32_t F_NX = 1u << 27, F_NY = 1u << 28, F_NZ = 1u << 29;   // direction component < 0 (Voxel_Grid.cs:589-632)
    constexpr uint32_t F_MOVED = 1u << 30;                                   // origin clipped to OBox: t_start in the scratch
    constexpr uint32_t F_HIT = 1u << 31;                                     // a hit is pending (tmin, point, polygon in the scratch)
    // Round 6: the pending hit's point lies in the voxel it was FOUND in -- bit 31 of the slot's list end L_qe (list positions stay below 2^31),
    // set by the exact phase when it accepts a hit, gone with the next voxel's list.  Voxel_Grid.cs:705 asks that question when the voxel's list
    // has been scanned, of the voxel the ray is still in: the same voxel, the same point, so the answer may be formed at the accept.  A ray whose
    // list ends with the flag up is FINISHED there and then (its event already holds t, the point and the polygon; u and v are zeroed) instead of
    // going through the pending-hit walk's own round: 1.38 -> 0.79 pend tasks per ray in the hall, 1.27 -> 0.58 in the cathedral.  Not for rays whose origin was moved (t = tmin + t_start is formed by the pend phase) nor in the BOUNCE build.
    constexpr uint32_t QE_HERE = 1u << 31;

    for (unsigned k = lane; k < S; k += 64) Q_free[k] = (uint8_t)k;
    // scene option "voxel_skip": the occupancy of the aligned 4^3 blocks of voxels, behind the pools (the host sized the launch's LDS for it)
    const bool skip_on = !BOUNCE && g.bocc != nullptr;
    uint32_t* const lbocc = reinterpret_cast<uint32_t*>(lds_raw + ((size_t)nw4 << 4) + (size_t)kPoolWaves * kPoolWaveBytes);
    if (skip_on) {
        const uint4* src = reinterpret_cast<const uint4*>(g.bocc);
        uint4* dst = reinterpret_cast<uint4*>(lbocc);
        for (int k = threadIdx.x; k < (g.bocc_words + 3) >> 2; k += blockDim.x) dst[k] = src[k];
    }
    __syncthreads();      // the bitmap is shared by the workgroup; everything after this point is wave-private

    const int ct = g.ct;
    const double fct = (double)ct;
    // wave-uniform queue state
    unsigned hW = 0, nW = 0, hC = 0, nC = 0, hE = 0, nE = 0, hP = 0, nP = 0, hF = 0, nF = S;
    unsigned hR = 0, nR = 0;            // BOUNCE: slots whose hit stands and whose ray goes on to its next cast
#ifdef HARE_K1Q_STATS                   // developer build (tools/k1q_stats.py): executions and active lanes of every phase; lane 0 counts
    unsigned long long kq_n[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, kq_l[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, kq_steps = 0;   // [8]: steps inside pend-walk tasks
    // ... and where a wave's TIME goes: every phase start closes the previous phase's interval on the shader clock ([9]: the round's own
    // bookkeeping -- queue choice, refill rule, fences; [7]: the step loop inside walk tasks)
    unsigned long long kq_t[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, kq_last_t = __builtin_amdgcn_s_memtime();
    int kq_last_i = 9;
    // (static indices under wave-uniform compares, and a second stamp when the bookkeeping is done: the clock's own cost is in no interval)
#define K1Q_CLOCK(i) { const unsigned long long d_ = __builtin_amdgcn_s_memtime() - kq_last_t; _Pragma("unroll") for (int k_ = 0; k_ < 10; ++k_) if (kq_last_i == k_) kq_t[k_] += d_; \
                       kq_last_i = (i); kq_last_t = __builtin_amdgcn_s_memtime(); }
#define K1Q_STAT(i, act) { kq_n[i]++; kq_l[i] += (unsigned long long)__popcll(__ballot(act)); K1Q_CLOCK(i) }
#else
#define K1Q_STAT(i, act)
#define K1Q_CLOCK(i)
#endif
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

    // ray chunks: static first chunk per wave (XCD-contiguous), then tickets -- as in K1p
    // the batch: io.n rays, or -- a cast of the bounce loop behind hare_reflect's block list -- the rays of the listed blocks (n_real guards the batch's
    // last, partial block)
    const unsigned n_real = (unsigned)io.n;
    unsigned n32 = n_real;
    // ... or -- a cast of the bounce loop behind hare_live_blocks (kernels.hip) -- the rays of the listed blocks of 64: a block in which every ray
    // has been retired is not in the list and costs this cast nothing.  A list that holds EVERY block (a closed room: nothing dies) is not
    // consulted at all.
    const uint32_t* blocks = nullptr;
    if (io.blocks) {
        const unsigned nb = (unsigned)__builtin_amdgcn_readfirstlane((int)io.blk_words[0]);
        if (nb < (n_real + 63u) / 64u) { blocks = io.blocks; n32 = nb * 64u; }     // (the batch's last block may reach past n_real: those lanes hold no ray)
    }
#ifndef HARE_K1Q_STATIC
#define HARE_K1Q_STATIC 128
#endif
    // the host shrinks the static first chunk for batches too small to give every wave of the grid 128 rays (ShootIO::static_rays): a
    // small batch is spread over ALL waves of the chip, which are then in their drain -- the wide modes -- from the second round on
    const unsigned RAY_CHUNK = io.static_rays > 0 ? (unsigned)io.static_rays : (unsigned)HARE_K1Q_STATIC;
    const unsigned n_static = gridDim.x * (unsigned)kPoolWaves * RAY_CHUNK;
    unsigned chunk_id = blockIdx.x * (unsigned)kPoolWaves + (unsigned)wave;
    if ((gridDim.x & 7u) == 0) chunk_id = ((blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3)) * (unsigned)kPoolWaves + (unsigned)wave;
    unsigned cn = chunk_id * RAY_CHUNK, ce = cn + RAY_CHUNK;
    if (cn > n32) cn = n32;
    if (ce > n32) ce = n32;
    bool drained = false;
    unsigned dead_run = 0;           // SHOOT_RETIRED_RAYS: tickets in a row whose rays were all retired (the next ticket is 2^dead_run x as large)
    bool chunk_live = true;          // ... a live ray seen since the last draw (the static first chunk counts as live: it is not a ticket)
    unsigned nhits = 0, nrays = 0;
    const bool writeback = (io.flags & SHOOT_WRITEBACK_ORIGIN) != 0;
    // the cooperative tail (voxel_coop.hip); with origin write-back the ray record holds the MOVED origin, which coop_trace would move again
    const bool coop = io.coop_tail != 0 && !writeback;
    unsigned tail_rounds = 0;        // rounds since the tickets ran dry
    const bool wide_on = io.wide_drain != 0;
    const bool hand_walk = io.hand_walk != 0;     // scene option "voxel_walk": the hand-written step loop (voxel_walk.h) / the compiler's

    auto store_miss = [&](unsigned ray) {
        XEventRec ev;
        set_miss(ev);
        store_event_streaming(&io.out[ray], ev);
    };
    auto occupied = [&](int X, int Y, int Z, int cell) -> bool {
        const uint32_t bit = COARSE ? (uint32_t)(((X >> g.occ_shift) * g.occ_cd + (Y >> g.occ_shift)) * g.occ_cd + (Z >> g.occ_shift))
                                    : (uint32_t)cell;
        return (locc[bit >> 5] >> (bit & 31)) & 1u;
    };
    // The voxel's tight box (hare_cell_boxes; device_scene.cpp: upload_cell_boxes): does the ray (origin as the walk uses it, i.e. moved) miss the box
    // of ALL polygons of the voxel's list?  Then the exact test cannot accept any of them and scanning the list would leave the ray as
    // it is.  FP64 slab test on reciprocals refined once (2^-47); the box is grown by 2^-20 of the scene's extent, a million times the
    // rounding of either test; rays that are not finite or start beyond 1 024 extents of the scene are never said to miss.  fmax / fmin
    // drop a NaN operand: a slab whose 1/d is infinite says nothing unless the origin lies outside it, where both products are the same
    // infinity and the ray is rightly found to miss.
    auto misses_cell_box = [&](int cell, double ox, double oy, double oz, double dx, double dy, double dz) -> bool {
        const float4* bp = reinterpret_cast<const float4*>(g.cellbox) + 2 * (size_t)cell;
        const float4 b0 = bp[0], b1 = bp[1];
        auto recip = [](double d) { const double y = __builtin_amdgcn_rcp(d); return __builtin_fma(y, __builtin_fma(-d, y, 1.0), y); };
        const double ix = recip(dx), iy = recip(dy), iz = recip(dz);
        const double x0 = ((double)b0.x - ox) * ix, x1 = ((double)b0.w - ox) * ix;
        const double y0 = ((double)b0.y - oy) * iy, y1 = ((double)b1.x - oy) * iy;
        const double z0 = ((double)b0.z - oz) * iz, z1 = ((double)b1.y - oz) * iz;
        const double tn = __builtin_fmax(__builtin_fmax(__builtin_fmin(x0, x1), __builtin_fmin(y0, y1)), __builtin_fmin(z0, z1));
        const double tf = __builtin_fmin(__builtin_fmin(__builtin_fmax(x0, x1), __builtin_fmax(y0, y1)), __builtin_fmax(z0, z1));
        const bool near_scene = fabs(ox - g.cellbox_mid[0]) <= g.cellbox_rad && fabs(oy - g.cellbox_mid[1]) <= g.cellbox_rad &&
                                fabs(oz - g.cellbox_mid[2]) <= g.cellbox_rad && fabs(dx) < 1e300 && fabs(dy) < 1e300 && fabs(dz) < 1e300;
        return near_scene && ((tf < tn) | (tf < 0));
    };
    // Voxel_Grid.cs:705 / AABB_Main.cs:75-84: the point inside the padded box of voxel (X, Y, Z), inclusive (six strict-reject compares)
    auto point_in_voxel = [&](int X, int Y, int Z, double hx, double hy, double hz) -> bool {
        const double lox = voxel_lo(X, g.vd[0], g.omin[0]), hix = voxel_hi(X, g.vd[0], g.omin[0]);
        const double loy = voxel_lo(Y, g.vd[1], g.omin[1]), hiy = voxel_hi(Y, g.vd[1], g.omin[1]);
        const double loz = voxel_lo(Z, g.vd[2], g.omin[2]), hiz = voxel_hi(Z, g.vd[2], g.omin[2]);
        return !(hx < lox) && !(hy < loy) && !(hz < loz) && !(hx > hix) && !(hy > hiy) && !(hz > hiz);
    };
    // one DDA step, Voxel_Grid.cs:713-759 written with selects (same booleans, same order; see K1p)
#define HARE_K1Q_STEP()                                                                          \
    {                                                                                            \
        const bool cxy = tMaxX < tMaxY, cxz = tMaxX < tMaxZ, cyz = tMaxY < tMaxZ;                \
        const bool sx = cxy & cxz;                                                               \
        const bool sy = (!cxy) & cyz;                                                            \
        const bool sz = !(sx | sy);                                                              \
        const double nX = tMaxX + tDeltaX, nY = tMaxY + tDeltaY, nZ = tMaxZ + tDeltaZ;           \
        X += sx ? dx1 : 0;                                                                       \
        Y += sy ? dy1 : 0;                                                                       \
        Z += sz ? dz1 : 0;                                                                       \
        tMaxX = sx ? nX : tMaxX;                                                                 \
        tMaxY = sy ? nY : tMaxY;                                                                 \
        tMaxZ = sz ? nZ : tMaxZ;                                                                 \
    }

    // ---- per-ray set-up into a slot: Voxel_Grid.cs:563-632 (a new ray of the batch; BOUNCE: also a reflected ray, rearm phase)
    auto arm = [&](unsigned slot, unsigned ray, V3 o, const V3 d, bool& to_walk, bool& to_cull, bool& freed) {
        bool alive = true, moved = false;
        double t_start = 0;
        double fx = floor((o.x - g.omin[0]) / g.vd[0]);
        double fy = floor((o.y - g.omin[1]) / g.vd[1]);
        double fz = floor((o.z - g.omin[2]) / g.vd[2]);
        bool inside = (fx >= 0.0 && fx < fct) & (fy >= 0.0 && fy < fct) & (fz >= 0.0 && fz < fct);
        if (!inside) {
            if (!aabb_clip_move(g.omin, g.omax, o, d, t_start)) {
                alive = false;
            } else {
                moved = true;
                if (writeback) { io.rays[ray].x = o.x; io.rays[ray].y = o.y; io.rays[ray].z = o.z; }
                fx = floor((o.x - g.omin[0] + d.x * 1E-6) / g.vd[0]);
                fy = floor((o.y - g.omin[1] + d.y * 1E-6) / g.vd[1]);
                fz = floor((o.z - g.omin[2] + d.z * 1E-6) / g.vd[2]);
                inside = (fx >= 0.0 && fx < fct) & (fy >= 0.0 && fy < fct) & (fz >= 0.0 && fz < fct);
                if (!inside) alive = false;
            }
        }
        if (alive) {
            const int X = (int)fx, Y = (int)fy, Z = (int)fz;
            const int cell = (X * ct + Y) * ct + Z;
            double tMaxX, tMaxY, tMaxZ, tDeltaX, tDeltaY, tDeltaZ;
            if (d.x < 0) { tMaxX = (voxel_lo(X, g.vd[0], g.omin[0]) - o.x) / d.x; tDeltaX = g.vd[0] / d.x * -1.0; }
            else         { tMaxX = (voxel_hi(X, g.vd[0], g.omin[0]) - o.x) / d.x; tDeltaX = g.vd[0] / d.x * 1.0; }
            if (d.y < 0) { tMaxY = (voxel_lo(Y, g.vd[1], g.omin[1]) - o.y) / d.y; tDeltaY = g.vd[1] / d.y * -1.0; }
            else         { tMaxY = (voxel_hi(Y, g.vd[1], g.omin[1]) - o.y) / d.y; tDeltaY = g.vd[1] / d.y * 1.0; }
            if (d.z < 0) { tMaxZ = (voxel_lo(Z, g.vd[2], g.omin[2]) - o.z) / d.z; tDeltaZ = g.vd[2] / d.z * -1.0; }
            else         { tMaxZ = (voxel_hi(Z, g.vd[2], g.omin[2]) - o.z) / d.z; tDeltaZ = g.vd[2] / d.z * 1.0; }
            L_tmx[slot] = tMaxX; L_tmy[slot] = tMaxY; L_tmz[slot] = tMaxZ;
            L_tdx[slot] = tDeltaX; L_tdy[slot] = tDeltaY; L_tdz[slot] = tDeltaZ;
            L_ray[slot] = ray;
            L_xyzf[slot] = (uint32_t)X | ((uint32_t)Y << 9) | ((uint32_t)Z << 18) | (d.x < 0 ? F_NX : 0u) | (d.y < 0 ? F_NY : 0u) |
                           (d.z < 0 ? F_NZ : 0u) | (moved ? F_MOVED : 0u);
            L_d1[slot] = -1;
            // the ray's scratch is its event slot (see the header).  Nothing is written at set-up: "no hit pending" is the
            // absence of F_HIT in the slot flags (round 4: one store and one visit of the record less per ray) -- except
            // for a ray that AABB.Intersect moved, whose t_start waits in the u field
            if (moved) reinterpret_cast<double*>(&io.out[ray])[1] = t_start;
            unsigned q = 0, qe = 0;
            int idx = -1, nexti = -1;
            if (occupied(X, Y, Z, cell)) {
                const CellRec c = g.cells[cell];
                q = c.start; qe = c.start + c.count; idx = c.i0; nexti = c.i1;
            }
            L_q[slot] = q; L_qe[slot] = qe; L_idx[slot] = idx; L_nexti[slot] = nexti;
            to_cull = q < qe;
            to_walk = !to_cull;
        } else {
            store_miss(ray);
            freed = true;
        }
    };

    // developer timeline (flag 0x2000, tools/timeline_prof.py): per wave {start, tickets dry, end, rounds} on the 100 MHz clock
    // (every lane stores the same word: an `if (lane == 0)` here would be a DIVERGENT branch whose join the compiler's uniformity analysis
    // shares with the wave-uniform state set beside it -- `drained` -- and through it every queue counter would count as divergent)
    auto timeline = [&](int slot, unsigned long long v) {
        if (__builtin_expect((io.flags & 0x2000u) != 0 && io.prof != nullptr, 0)) {
            io.prof[32 + 4ull * (blockIdx.x * (unsigned)kPoolWaves + (unsigned)wave) + slot] = v;
        }
    };
    timeline(0, __builtin_amdgcn_s_memrealtime());
    unsigned rounds_done = 0;
    unsigned long long stat_lanes = 0, stat_distinct = 0, stat_batches = 0;
    // a wave serves ~n / (waves in the grid) rays in a few rounds each; the cap only exists so that a defect can never
    // turn into a wave that does not finish (rays it left behind would keep their scratch values and fail every parity test)
#define HARE_K1Q_PHASE_FENCE() __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront")
#define HARE_K1Q_TAIL_FENCE() __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront")
#define HARE_K1Q_FINAL_FENCE() __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront")
    unsigned helped = 0;
    unsigned long long t_coop = 0;
    for (;;) {          // (BOUNCE: the cooperative tail can send a ray back to the pool for its next cast)
    for (unsigned round = 0; round < (1u << 24); ++round) {
        K1Q_CLOCK(9)
        // ------------------------------------------------------------------ set-up of new rays into free slots
        if (!drained && (nF >= (unsigned)HARE_K1Q_REFILL_MIN || nW + nC + nE + nP + nR == 0)) {
            if (cn >= ce && n_static >= n32) {
                // the static first chunks cover the whole batch (a small batch; a bounce cast whose block list is short or empty): there is no
                // ticket to draw -- 3 072 waves drawing one each from one address is 34 us, half of what an EMPTY cast used to cost
                drained = true;
                cn = ce = n32;
                timeline(1, __builtin_amdgcn_s_memrealtime());
            } else if (cn >= ce) {
                unsigned base = 0;
                unsigned dyn = (unsigned)io.ticket_rays;
                if (io.flags & SHOOT_RETIRED_RAYS) {
                    // a cast of the bounce loop over rays most of which the loop has retired (an open scene): a ticket whose rays were ALL
                    // retired costs this wave nothing but the draw, and the draws -- ~11 ns each on one address, chip-wide -- are then the
                    // whole launch (measured: 192 us per million retired rays, three quarters of a cast of live rays;
                    // profiles/r05_experiments/bounce_open_scene.log).  Every such ticket in a row doubles the next one, up to 64 x;
                    // the first live ray puts the size back.  A closed room never sees a dead ticket.
                    dead_run = chunk_live ? 0u : (dead_run < 6u ? dead_run + 1u : 6u);
                    chunk_live = false;
                    dyn <<= dead_run;
                }
                K1Q_CLOCK(8)
                if (lane == 0) base = atomicAdd(io.work, dyn);
                base = (unsigned)__builtin_amdgcn_readfirstlane((int)base);      // lane 0's ticket, in a scalar register
                K1Q_CLOCK(9)
                cn = base + n_static;
                if (cn >= n32) { drained = true; cn = ce = n32; timeline(1, __builtin_amdgcn_s_memrealtime()); }
                else ce = (n32 - cn > dyn) ? cn + dyn : n32;
            }
            unsigned m = ce - cn;
            if (m > 64u) m = 64u;
            if (m > nF) m = nF;
            if (m > 0) {
                const bool act = lane < m;
                K1Q_STAT(0, act)
                const unsigned slot = Q_free[(hF + (act ? lane : 0u)) & SM];
                hF = (hF + m) & SM;
                nF -= m;
                unsigned ray = (io.order != nullptr && act) ? io.order[cn + lane] : cn + lane;
                if (blocks != nullptr && act) ray = (blocks[(cn + lane) >> 6] << 6) | ((cn + lane) & 63u);
                cn += m;
                bool to_walk = false, to_cull = false, freed = false, retired_lane = false;
                if (act) {
                    // a ray the bounce loop retired (hare_reflect marked it -2): a miss record, no traversal, not counted -- and NOTHING ELSE of it
                    // is touched under SHOOT_RETIRED_SILENT -- the loop writes every
                    // cast into ONE event buffer, so the slot already holds the miss record of the cast the ray died in -- nothing is written
                    // (round 6: a cast over a million retired rays 72 -> see profiles/r06_experiments/bounce_open_scene.log)
                    const bool past_end = ray >= n_real;                 // the tail of the batch's last block (a block list only)
                    // (the record is requested together with the mark, not behind it: a set-up round waits for ONE round of loads.  Blocks in
                    //  which every ray is retired never get here -- the block list -- so the record of a retired ray is read only next to live ones)
                    const RayRec r = io.rays[past_end ? 0u : ray];
                    retired_lane = past_end || ((io.flags & SHOOT_RETIRED_RAYS) && io.excl1 && io.excl1[ray] == -2);
                    if (retired_lane) {
                        if (!past_end && !(io.flags & SHOOT_RETIRED_SILENT)) store_miss(ray);
                        freed = true;
                    } else {
                        const V3 o = {r.x, r.y, r.z};
                        const V3 d = {r.dx, r.dy, r.dz};
                        nrays++;
                        if (BOUNCE) { L_cast[slot] = 0; atomicAdd(&C_rays[0], 1u); }
                        arm(slot, ray, o, d, to_walk, to_cull, freed);
                    }
                }
                if (OWN && (to_walk || to_cull)) own.cells++;            // the voxel the ray starts in
                if (io.flags & SHOOT_RETIRED_RAYS) chunk_live = chunk_live || __ballot(act && !retired_lane) != 0ull;
                push(Q_walk, hW, nW, to_walk, slot);
                push(Q_cull, hC, nC, to_cull, slot);
                push(Q_free, hF, nF, freed, slot);
            }
        }
        K1Q_CLOCK(9)
#ifndef HARE_K1Q_REARM_MIN
#define HARE_K1Q_REARM_MIN 8      // BOUNCE: reflect + set up again when this many rays wait for it (or nothing else can run, or the launch drains); swept
                                  // 1 / 8 / 16 / 32 / 48: C5 shard 675 / 677 / 664 / 610 / 520 Mcasts/s -- a ray parked in this queue is a slot of the pool not working
#endif
        if (BOUNCE && nR > 0 && (nR >= (unsigned)HARE_K1Q_REARM_MIN || nW + nC + nE + nP == 0 || drained)) {
            // -------------------------------------------------------------- BOUNCE: the hit stands -- next cast of the same ray
            HARE_K1Q_PHASE_FENCE();      // the confirmed event in the scratch, before it is read here
            bool act;
            const unsigned slot = pop(Q_rearm, hR, nR, act);
            bool to_walk = false, to_cull = false, freed = false;
            if (act) {
                const unsigned ray = L_ray[slot];
                const int c = (int)L_cast[slot];
                double* const sc = reinterpret_cast<double*>(&io.out[ray]);
                const double hx = sc[3], hy = sc[4], hz = sc[5];                 // X_Point of the hit that stands
                const double w6 = sc[6];
                const int pid = __double2loint(w6);
                if (io.out_all) {                                               // every cast's event, cast-major
                    double* const dst = reinterpret_cast<double*>(&io.out_all[(int64_t)c * io.out_stride + ray]);
                    const double t0 = sc[0], u0 = sc[1], v0 = sc[2];
                    __builtin_nontemporal_store(t0, dst + 0); __builtin_nontemporal_store(u0, dst + 1); __builtin_nontemporal_store(v0, dst + 2);
                    __builtin_nontemporal_store(hx, dst + 3); __builtin_nontemporal_store(hy, dst + 4); __builtin_nontemporal_store(hz, dst + 5);
                    __builtin_nontemporal_store(w6, dst + 6);
                }
                if (c + 1 < n_casts) {
                    // hare_reflect's expressions: o' = X_Point, d' = d - (2 * (d . n)) * n, the next cast excludes the polygon just left
                    const RayRec r = io.rays[ray];
                    const PolyRec& p = g.polys[pid];
                    const double dn = dot3(r.dx, r.dy, r.dz, p.n[0], p.n[1], p.n[2]);
                    const double k = 2.0 * dn;
                    RayRec nr;
                    nr.x = hx; nr.y = hy; nr.z = hz;
                    nr.dx = r.dx - k * p.n[0];
                    nr.dy = r.dy - k * p.n[1];
                    nr.dz = r.dz - k * p.n[2];
                    io.rays[ray] = nr;
                    const_cast<int32_t*>(io.excl1)[ray] = pid;
                    if (io.excl2) const_cast<int32_t*>(io.excl2)[ray] = -1;
                    L_cast[slot] = (uint8_t)(c + 1);
                    atomicAdd(&C_rays[c + 1], 1u);
                    nrays++;
                    const V3 o2 = {nr.x, nr.y, nr.z}, d2 = {nr.dx, nr.dy, nr.dz};
                    arm(slot, ray, o2, d2, to_walk, to_cull, freed);
                    if (OWN && (to_walk || to_cull)) own.cells++;
                } else {
                    freed = true;                                               // its last cast: the event is where it belongs
                }
            }
            push(Q_walk, hW, nW, to_walk, slot);
            push(Q_cull, hC, nC, to_cull, slot);
            push(Q_free, hF, nF, freed, slot);
        }
        {
            const unsigned left = nW + nC + nE + nP + nR;
            if (drained) {
                if (left == 0) break;
                // down to the last rays: one or two at once; a handful when they have outlived the others by HARE_K1Q_COOP_PATIENCE
                // rounds (heavy rays) -- from here on the whole wave traces them one after the other (voxel_coop.hip)
                if (coop && nR == 0 && (left <= (unsigned)HARE_K1Q_COOP_NOW || (left <= (unsigned)HARE_K1Q_COOP_MAX && tail_rounds >= (unsigned)HARE_K1Q_COOP_PATIENCE))) break;
                ++tail_rounds;
            } else if (left == 0) continue;
            if (BOUNCE && nW + nC + nE + nP == 0) continue;              // only the rearm queue holds rays: its phase runs at the top of the round
        }
        ++rounds_done;
        if (__builtin_expect((io.flags & 0x1000u) != 0 && io.prof != nullptr, 0)) {
            // developer round trace (flag 0x1000, tools/round_trace.py): one wave in 256 stamps every round with the clock and its queue lengths
            const unsigned gw = blockIdx.x * (unsigned)kPoolWaves + (unsigned)wave;
            const unsigned long long stamp = ((unsigned long long)__builtin_amdgcn_s_memrealtime() << 34) |
                                             ((unsigned long long)(drained ? 1u : 0u) << 32) | nW | (nC << 8) | (nE << 16) | (nP << 24);
            if ((gw & 255u) == 5u && rounds_done < 1024u) io.prof[32 + 4 * 4096 + (gw >> 8) * 1024 + rounds_done] = stamp;      // (every lane the same word: see timeline())
            // ... and EVERY wave its first 48 rounds after the tickets ran dry (what the latest waves of a launch are doing)
            if (drained && tail_rounds <= 48u && tail_rounds > 0u) io.prof[32 + 4 * 4096 + 16 * 1024 + gw * 48u + (tail_rounds - 1u)] = stamp;
        }

        // ------------------------------------------------------------------ pick the phase for this round
        // Normally ONE phase per round, the one that fills the lanes best.  Once the tickets are dry and few rays are
        // left, every non-empty phase runs each round, in the order a ray passes through them (walk -> cull -> exact ->
        // pend), so that a ray advances several phases per round: at the end of a launch latency is all that counts.
        const unsigned big = nW > nC ? nW : nC;
        const bool tail = drained && nW + nC + nE + nP + nR <= (unsigned)HARE_K1Q_TAIL;
        // ... and once the pool is down to a few rays, a ray's candidates are spread over several lanes (the wide cull below).  Both
        // conditions only ever go from false to true (no ray is set up after the tickets ran dry), which the wide cull relies on.
        const bool wide = HARE_K1Q_WIDE_MAX > 0 && wide_on && drained && nW + nC + nE + nP + nR <= (unsigned)HARE_K1Q_WIDE_MAX;
        const int sel = (nE >= (unsigned)HARE_K1Q_EXACT_MIN || (big == 0 && nP == 0)) ? 0
                        : ((nP >= (unsigned)HARE_K1Q_PEND_MIN || big == 0) ? 1 : (nC >= nW ? 2 : 3));
        HARE_K1Q_PHASE_FENCE();      // the set-up's scratch stores, before any phase reads them
        if (HARE_K1Q_WIDE_WALK && wide && nW > 0) {
            K1Q_STAT(5, true)
            // -------------------------------------------------------------- the WIDE walk of the drain: several occupied voxels ahead
            // What is left at the very end of a launch are rays that cross many occupied voxels whose candidates all fail the pre-cull:
            // walk -> cull -> walk ..., one voxel per round, three dependent round trips each (tools/round_trace.py: the latest waves
            // of a 1M-ray launch spend 8 - 10 rounds of ~5 us on their last three or four rays; a late bounce cast in the cathedral
            // 35 - 40 rounds).  Here every lane of a ray's group of G runs the SAME walk (identical state: no divergence inside a
            // group) and lane `sub` keeps the sub-th occupied voxel the walk meets; the G voxels' cell records are fetched together,
            // and each lane scans ITS voxel's list four candidates at a time until one survives the pre-cull.  The ray stops at the
            // FIRST of these voxels (in walk order) with a survivor, at that survivor; voxels in front of it whose candidates all
            // failed are exactly the ones the one-voxel-per-round sequence would have passed (the pre-cull has no state; the walk queue
            // holds rays WITHOUT a pending hit, so there is no IsPointInBox rule to apply on the way).  No survivor anywhere and the
            // grid left: miss (Voxel_Grid.cs:716-757).  Work on voxels behind the stop is wasted, nothing else.
            const unsigned nq = nW;
            unsigned gsh = 0;
            while (gsh < 4u && (nq << (gsh + 1u)) <= 64u) ++gsh;
            const unsigned G = 1u << gsh, grp = lane >> gsh, sub = lane & (G - 1u);
            const bool act = grp < nq;
            const unsigned slot = Q_walk[(hW + (act ? grp : 0u)) & SM];
            hW = (hW + nq) & SM;
            nW = 0;
            double tMaxX = 0, tMaxY = 0, tMaxZ = 0, tDeltaX = 0, tDeltaY = 0, tDeltaZ = 0, sX = 0, sY = 0, sZ = 0;
            int X = 0, Y = 0, Z = 0, dx1 = 1, dy1 = 1, dz1 = 1, vX = 0, vY = 0, vZ = 0;
            uint32_t xf = 0;
            if (act) {
                tMaxX = L_tmx[slot]; tMaxY = L_tmy[slot]; tMaxZ = L_tmz[slot];
                tDeltaX = L_tdx[slot]; tDeltaY = L_tdy[slot]; tDeltaZ = L_tdz[slot];
                xf = L_xyzf[slot];
                X = (int)(xf & 511u); Y = (int)((xf >> 9) & 511u); Z = (int)((xf >> 18) & 511u);
                dx1 = (xf & F_NX) ? -1 : 1; dy1 = (xf & F_NY) ? -1 : 1; dz1 = (xf & F_NZ) ? -1 : 1;
            }
            bool walking = act, mine = false, exited = false;
            unsigned lead_sub = 0u;                  // the lane of a group that holds the state the walk ENDED in
            if (HARE_K1Q_HAND_WALK && hand_walk) {
                // by hand (voxel_walk.h): lane `sub` passes `sub` occupied voxels and stops AT the next one -- its own -- or outside the grid;
                // the lanes of a group run the same walk, so the group's last lane ends where the walk of all G voxels ends
                unsigned taken = 0, iters = 0;
                hare_walk::walk_steps<COARSE, OWN, true>(tMaxX, tMaxY, tMaxZ, tDeltaX, tDeltaY, tDeltaZ, X, Y, Z, dx1, dy1, dz1, walking, (unsigned)ct,
                                                         (unsigned)HARE_K1Q_TAIL_STEPS, 1u, lds_bitmap, (unsigned)g.occ_shift, (unsigned)g.occ_cd, taken, iters, sub);
                lead_sub = G - 1u;
                exited = act && (((unsigned)X >= (unsigned)ct) | ((unsigned)Y >= (unsigned)ct) | ((unsigned)Z >= (unsigned)ct));
                mine = act && !walking && !exited;
                sX = tMaxX; sY = tMaxY; sZ = tMaxZ; vX = X; vY = Y; vZ = Z;
                if (OWN && sub == lead_sub) own.cells += taken;              // the group's lanes run the same walk: counted once
            } else {
            unsigned seen = 0;
#pragma unroll 1
            for (int k = 0; k < HARE_K1Q_TAIL_STEPS; ++k) {
                if (__ballot(walking) == 0) break;
                if (walking) {
                    HARE_K1Q_STEP();
                    const bool out = ((unsigned)X >= (unsigned)ct) | ((unsigned)Y >= (unsigned)ct) | ((unsigned)Z >= (unsigned)ct);
                    if (OWN && !out && sub == 0u) own.cells++;                // the group's lanes run the same walk: counted once
                    if (out) {
                        exited = true;
                        walking = false;
                    } else if (occupied(X, Y, Z, (X * ct + Y) * ct + Z)) {
                        if (seen == sub) { mine = true; sX = tMaxX; sY = tMaxY; sZ = tMaxZ; vX = X; vY = Y; vZ = Z; }
                        ++seen;
                        walking = seen < G;
                    }
                }
            }
            }
            // the G voxels of a ray, one per lane: cell record, then its list in chunks of four (entries, then their records)
            bool surv = false;
            unsigned q_stop = 0, qe_stop = 0;
            int it_stop = -1;
            {
                unsigned start = 0, cnt = 0;
                int i0 = 0, i1 = 0, e1 = -1, e2 = -1, done1 = -1;
                CullRay cray = {};
                if (mine) {
                    const CellRec c = g.cells[(vX * ct + vY) * ct + vZ];
                    const unsigned ray = L_ray[slot];
                    done1 = L_d1[slot];
                    if (io.excl1) e1 = io.excl1[ray];
                    if (io.excl2) e2 = io.excl2[ray];
                    const RayRec r = io.rays[ray];
                    double ox = r.x, oy = r.y, oz = r.z;
                    if ((xf & F_MOVED) && !writeback) {
                        const double ts = reinterpret_cast<const double*>(&io.out[ray])[1];
                        ox = ox + r.dx * ts; oy = oy + r.dy * ts; oz = oz + r.dz * ts;
                    }
                    cray = cull_ray(g, ox, oy, oz, r.dx, r.dy, r.dz);
                    start = c.start; cnt = c.count; i0 = c.i0; i1 = c.i1;     // cnt == 0 on a coarse bitmap: the block is occupied, this voxel is not
                    qe_stop = start + cnt;
                }
                bool scanning = mine && cnt > 0u;
                // the chunk's four list entries are requested one iteration ahead, together with the previous chunk's records: one round
                // trip per chunk, not two (a list of one or two entries travels in the cell record)
                int n0 = i0, n1 = cnt > 1u ? i1 : i0, n2 = i0, n3 = i0;
                auto entries = [&](unsigned base) {
                    const unsigned last = cnt - 1u;
                    n0 = g.items[start + (base < last ? base : last)];
                    n1 = g.items[start + (base + 1u < last ? base + 1u : last)];
                    n2 = g.items[start + (base + 2u < last ? base + 2u : last)];
                    n3 = g.items[start + (base + 3u < last ? base + 3u : last)];
                };
                if (scanning && cnt > 2u) entries(0u);
#pragma unroll 1
                for (unsigned base = 0; base < 4096u; base += 4u) {
                    if (__ballot(scanning) == 0) break;
                    if (scanning) {
                        const int it0 = n0, it1 = n1, it2 = n2, it3 = n3;
                        if (OWN) { const unsigned m4 = cnt - base < 4u ? cnt - base : 4u; own.entries += m4; own.culls += m4; }
                        if (base + 4u < cnt) entries(base + 4u);
                        const CullRaw r0 = cull_load<QUADS ? 1 : 0>(g, it0), r1 = cull_load<QUADS ? 1 : 0>(g, it1), r2 = cull_load<QUADS ? 1 : 0>(g, it2), r3 = cull_load<QUADS ? 1 : 0>(g, it3);
                        auto keep = [&](unsigned k, int it, const CullRaw& rr) {
                            return k < cnt && it != e1 && it != e2 && !(HARE_K1Q_MAILBOX && it == done1) && !cull_test<QUADS ? 1 : 0>(g, cray, rr);
                        };
                        const bool k0 = keep(base, it0, r0), k1 = keep(base + 1u, it1, r1), k2 = keep(base + 2u, it2, r2), k3 = keep(base + 3u, it3, r3);
                        if (k0 | k1 | k2 | k3) {
                            surv = true;
                            q_stop = start + base + (k0 ? 0u : (k1 ? 1u : (k2 ? 2u : 3u)));
                            it_stop = k0 ? it0 : (k1 ? it1 : (k2 ? it2 : it3));
                            scanning = false;
                        } else if (base + 4u >= cnt) {
                            scanning = false;
                        }
                    }
                }
            }
            const unsigned sh = grp << gsh;
            const unsigned gm = (1u << G) - 1u;
            const unsigned ms = (unsigned)(__ballot(surv) >> sh) & gm;
            const bool has_stop = ms != 0u;
            const unsigned stop_sub = has_stop ? (unsigned)__builtin_ctz(ms) : 0u;
            const bool writer = act && has_stop && sub == stop_sub;          // the lane that holds the voxel the ray stops in
            const bool lead = act && !has_stop && sub == lead_sub;           // no stop: the ray walks on from where the walk ended, or has left the grid
            if (writer) {
                L_tmx[slot] = sX; L_tmy[slot] = sY; L_tmz[slot] = sZ;
                L_xyzf[slot] = (xf & 0xF8000000u) | (uint32_t)vX | ((uint32_t)vY << 9) | ((uint32_t)vZ << 18);
                L_q[slot] = q_stop; L_qe[slot] = qe_stop;
                L_idx[slot] = it_stop;
            }
            if (lead) {
                if (exited) {
                    store_miss(L_ray[slot]);
                } else {
                    L_tmx[slot] = tMaxX; L_tmy[slot] = tMaxY; L_tmz[slot] = tMaxZ;
                    L_xyzf[slot] = (xf & 0xF8000000u) | (uint32_t)X | ((uint32_t)Y << 9) | ((uint32_t)Z << 18);
                }
            }
            push(Q_walk, hW, nW, lead && !exited, slot);
            push(Q_exact, hE, nE, writer, slot);
            push(Q_free, hF, nF, lead && exited, slot);
        } else if (tail ? nW > 0 : sel == 3) {
            // -------------------------------------------------------------- DDA walk over empty voxels (no hit pending)
            bool act;
            const unsigned slot = pop(Q_walk, hW, nW, act);
            K1Q_STAT(1, act)
            bool walking = act;
            double tMaxX = 0, tMaxY = 0, tMaxZ = 0, tDeltaX = 0, tDeltaY = 0, tDeltaZ = 0;
            int X = 0, Y = 0, Z = 0, dx1 = 1, dy1 = 1, dz1 = 1;
            uint32_t xf = 0;
            if (act) {
                tMaxX = L_tmx[slot]; tMaxY = L_tmy[slot]; tMaxZ = L_tmz[slot];
                tDeltaX = L_tdx[slot]; tDeltaY = L_tdy[slot]; tDeltaZ = L_tdz[slot];
                xf = L_xyzf[slot];
                X = (int)(xf & 511u); Y = (int)((xf >> 9) & 511u); Z = (int)((xf >> 18) & 511u);
                dx1 = (xf & F_NX) ? -1 : 1; dy1 = (xf & F_NY) ? -1 : 1; dz1 = (xf & F_NZ) ? -1 : 1;
            }
            // the task ends early once few of its lanes still walk (the others have found their voxel): a third of what it
            // started with, so that a thin batch -- the end of the launch -- is not cut down to one step per round
            const int n0 = __popcll(__ballot(walking));
            const int walk_min = tail ? 1 : (n0 / HARE_K1Q_WALK_DIV < HARE_K1Q_WALK_MIN ? n0 / HARE_K1Q_WALK_DIV : HARE_K1Q_WALK_MIN);
            // (steps per task: 16, or -- the host's rule for batches of a pool fill's double and more, ShootIO::walk_steps -- 32: with the hand-written
            //  step a task's set-up weighs more than its steps; C2 +1.3 %, 4M rays +1.9 %, C4 shard +2.8 %, but -2.7 % at 262k rays)
            const int walk_steps = tail ? HARE_K1Q_TAIL_STEPS : (io.walk_steps > 0 ? io.walk_steps : HARE_K1Q_WALK_STEPS);   // end of the launch: fewer, longer tasks
            // A task is up to HARE_K1Q_WALK_SEGS SEGMENTS of steps (round 6).  A lane that stops in an occupied voxel fetches the cell record and tests the
            // voxel's tight box at the end of its segment; when that sends it on -- the list is empty (a per-block bitmap) or the ray cannot hit anything
            // of it -- it used to go back to the walk queue for its next ROUND (state to LDS, a push, a pop, the state back: a reflected ray in the
            // cathedral is sent on from 2.4 of the 5.2 occupied voxels it meets, and made 15 walk tasks at 25 lanes a step).  Now, when enough lanes
            // of the task are sent on, they walk on in the same task with what is left of its step budget.
            bool exited = false, to_cull = false;
            unsigned q = 0, qe = 0;
            int idx = -1, nexti = -1;
            int steps_left = walk_steps;
#pragma unroll 1
            for (int seg = 0; seg < HARE_K1Q_WALK_SEGS; ++seg) {
            K1Q_CLOCK(7)
            if (skip_on) {
                // ---- the EXACT multi-voxel skip (scene option "voxel_skip"; round 4's construction, built in round 6).  The DDA is a merge
                // of three sequences a_c[j] = tMax_c + j tDelta_c -- each formed by SEQUENTIAL adds, as the reference forms them -- under the order
                // "smaller value first, the later axis on a tie" (Voxel_Grid.cs:713-757: x steps iff tx < ty && tx < tz; else y iff ty < tz; else z).
                // A lane whose aligned 4^3 block holds no occupied voxel leaves it in ONE operation: k_c = steps on axis c that leave the block; the
                // exit element E = the first, in that order, of the three a_c[k_c - 1]; every other axis has taken as many steps as it has elements
                // in front of E (a < E, or a == E on the later axis); the exit axis all k, and one more add.  The voxels in between lie in the
                // empty block -- no list, and a ray of the walk queue holds no hit -- so nothing else would have happened there: the lane stands
                // where the stepping loop would stand, with the same three tMax bit patterns.  Used only when all six values are finite (then
                // tDelta > 0 and the sequences rise); other lanes step.  A block that reaches past the grid's edge is no exception: a ray
                // that leaves the grid inside it has left it for good, and ends as the miss it is.
                // It is exact -- and slower than the hand-written step (38 vector instructions per voxel crossed against 17: DESIGN.md section 5,
                // profiles/r06_experiments): off by default.
                const int nb = g.bocc_nb;
#pragma unroll 1
                for (int k = 0; k < steps_left; ++k) {
                    const unsigned long long wm = __ballot(walking);
                    if (wm == 0 || (k > 0 && __popcll(wm) < (seg == 0 ? walk_min : 1))) break;
                    if (walking) {
                        const int bxi = ((X >> 2) * nb + (Y >> 2)) * nb + (Z >> 2);
                        const bool finite = fabs(tMaxX) < 1e300 && fabs(tMaxY) < 1e300 && fabs(tMaxZ) < 1e300 && fabs(tDeltaX) < 1e300 && fabs(tDeltaY) < 1e300 &&
                                            fabs(tDeltaZ) < 1e300;
                        const bool jump = finite && !((lbocc[bxi >> 5] >> (bxi & 31)) & 1u);
                        unsigned crossed = 1;
                        if (jump) {
                            const int kx = dx1 > 0 ? 4 - (X & 3) : (X & 3) + 1, ky = dy1 > 0 ? 4 - (Y & 3) : (Y & 3) + 1, kz = dz1 > 0 ? 4 - (Z & 3) : (Z & 3) + 1;
                            const double ax1 = tMaxX + tDeltaX, ax2 = ax1 + tDeltaX, ax3 = ax2 + tDeltaX, ax4 = ax3 + tDeltaX;
                            const double ay1 = tMaxY + tDeltaY, ay2 = ay1 + tDeltaY, ay3 = ay2 + tDeltaY, ay4 = ay3 + tDeltaY;
                            const double az1 = tMaxZ + tDeltaZ, az2 = az1 + tDeltaZ, az3 = az2 + tDeltaZ, az4 = az3 + tDeltaZ;
                            auto sel5 = [](int j, double a0, double a1, double a2, double a3, double a4) { return j == 0 ? a0 : (j == 1 ? a1 : (j == 2 ? a2 : (j == 3 ? a3 : a4))); };
                            const double Ex = sel5(kx - 1, tMaxX, ax1, ax2, ax3, ax4), Ey = sel5(ky - 1, tMaxY, ay1, ay2, ay3, ay4), Ez = sel5(kz - 1, tMaxZ, az1, az2, az3, az4);
                            const bool ex = (Ex < Ey) & (Ex < Ez), ey = (!(Ex < Ey)) & (Ey < Ez), ez = !(ex | ey);
                            const double E = ex ? Ex : (ey ? Ey : Ez);
                            auto before = [&](bool later, int kk, double a0, double a1, double a2) {
                                int c = 0;
                                c += (kk > 1 && (a0 < E || (later && a0 == E))) ? 1 : 0;
                                c += (kk > 2 && (a1 < E || (later && a1 == E))) ? 1 : 0;
                                c += (kk > 3 && (a2 < E || (later && a2 == E))) ? 1 : 0;
                                return c;
                            };
                            const int nx = ex ? kx : before(false, kx, tMaxX, ax1, ax2);              // x is never the later axis
                            const int ny = ey ? ky : before(ex, ky, tMaxY, ay1, ay2);                 // y is later than x only
                            const int nz = ez ? kz : before(true, kz, tMaxZ, az1, az2);               // z is later than both
                            X += dx1 * nx; Y += dy1 * ny; Z += dz1 * nz;
                            tMaxX = sel5(nx, tMaxX, ax1, ax2, ax3, ax4); tMaxY = sel5(ny, tMaxY, ay1, ay2, ay3, ay4); tMaxZ = sel5(nz, tMaxZ, az1, az2, az3, az4);
                            crossed = (unsigned)(nx + ny + nz);
                        } else {
                            HARE_K1Q_STEP();
                        }
                        const bool out = ((unsigned)X >= (unsigned)ct) | ((unsigned)Y >= (unsigned)ct) | ((unsigned)Z >= (unsigned)ct);
                        const int cell = out ? 0 : (X * ct + Y) * ct + Z;
                        const bool occ = occupied(out ? 0 : X, out ? 0 : Y, out ? 0 : Z, cell);
                        walking = !out && !occ;
                        if (OWN) { own.cells += out ? crossed - 1u : crossed; own.steps++; }       // voxels walked into; operations executed
                    }
                }
                steps_left = 0;
            } else if (HARE_K1Q_HAND_WALK && hand_walk) {
                // the step loop written by hand (voxel_walk.h): the same steps, the per-axis updates under the axis' own EXEC mask
                unsigned taken = 0, iters = 0;
                hare_walk::walk_steps<COARSE, OWN || kK1qStats>(tMaxX, tMaxY, tMaxZ, tDeltaX, tDeltaY, tDeltaZ, X, Y, Z, dx1, dy1, dz1, walking, (unsigned)ct,
                                                             (unsigned)steps_left, (unsigned)(seg == 0 ? walk_min : 1), lds_bitmap, (unsigned)g.occ_shift, (unsigned)g.occ_cd, taken, iters);
                if (OWN) { own.cells += taken; own.steps += taken; }
                steps_left -= (int)iters;
#ifdef HARE_K1Q_STATS
                kq_n[7] += iters; kq_l[7] += wave_sum_u32(taken);      // executions of the step; voxels walked into (lane 0 holds the sum)
#endif
            } else {
#pragma unroll 1
            for (int k = 0; k < steps_left; ++k) {
                const unsigned long long wm = __ballot(walking);
                if (wm == 0 || (k > 0 && __popcll(wm) < (seg == 0 ? walk_min : 1))) break;
#ifdef HARE_K1Q_STATS
                kq_n[7]++; kq_l[7] += (unsigned long long)__popcll(wm);
#endif
                if (walking) {
                    HARE_K1Q_STEP();
                    const bool out = ((unsigned)X >= (unsigned)ct) | ((unsigned)Y >= (unsigned)ct) | ((unsigned)Z >= (unsigned)ct);
                    const int cell = out ? 0 : (X * ct + Y) * ct + Z;
                    const bool occ = occupied(out ? 0 : X, out ? 0 : Y, out ? 0 : Z, cell);
                    walking = !out && !occ;
                    if (OWN && !out) own.cells++;
                }
            }
            steps_left = 0;          // (the compiler's loop, A/B: one segment)
            }
            K1Q_CLOCK(1)
            const bool out_now = act && !exited && !to_cull && (((unsigned)X >= (unsigned)ct) | ((unsigned)Y >= (unsigned)ct) | ((unsigned)Z >= (unsigned)ct));
            const bool stopped = act && !exited && !to_cull && !walking && !out_now;       // in an occupied voxel (block), not yet looked at
            if (out_now) store_miss(L_ray[slot]);                           // leaving the grid: miss
            exited = exited || out_now;
            bool sent_on = false;
            if (stopped) {
                const int cell = (X * ct + Y) * ct + Z;
                const CellRec c = g.cells[cell];
                q = c.start; qe = c.start + c.count; idx = c.i0; nexti = c.i1;
                bool keep = true;
                if (COARSE && c.count == 0) keep = false;                       // the block is occupied, this voxel is not: walk on
                // the voxel's tight box (above): the ray cannot hit anything of this list -- it walks on as if the voxel were empty.  (A wall
                // next to the ray's path: its polygons lie in the voxels the ray crosses, but their box is a thin slab the ray never
                // reaches -- two in five of the list entries a reflected ray scans in the cathedral.)  Only here, where a ray without
                // a hit meets an occupied voxel in the pool's ordinary walk: the same test in the wide walk of the drain and in the
                // cooperative tail was measured and is worth nothing (C4 shard -2.5 % with it in the tail; k1q_box_variants.log).
                if (g.cellbox != nullptr && keep) {
                    const unsigned ray = L_ray[slot];
                    const RayRec r = io.rays[ray];
                    double ox = r.x, oy = r.y, oz = r.z;
                    if ((xf & F_MOVED) && !writeback) {
                        const double ts = reinterpret_cast<const double*>(&io.out[ray])[1];
                        ox = ox + r.dx * ts; oy = oy + r.dy * ts; oz = oz + r.dz * ts;
                    }
                    if (misses_cell_box(cell, ox, oy, oz, r.dx, r.dy, r.dz)) keep = false;
                }
                to_cull = keep;
                sent_on = !keep;
                walking = walking || sent_on;
            }
            // walk on in this task?  Only the lanes just sent on and those the budget stopped; not worth it for a few
            if (seg + 1 >= HARE_K1Q_WALK_SEGS || steps_left < 4 || (int)__popcll(__ballot(sent_on)) < HARE_K1Q_WALK_RESUME_MIN) break;
            }
            if (act && !exited) {
                L_tmx[slot] = tMaxX; L_tmy[slot] = tMaxY; L_tmz[slot] = tMaxZ;
                L_xyzf[slot] = (xf & 0xF8000000u) | (uint32_t)X | ((uint32_t)Y << 9) | ((uint32_t)Z << 18);
                if (to_cull) { L_q[slot] = q; L_qe[slot] = qe; L_idx[slot] = idx; L_nexti[slot] = nexti; }
            }
            push(Q_walk, hW, nW, walking, slot);                            // still walking: next round
            push(Q_cull, hC, nC, to_cull, slot);
            push(Q_free, hF, nF, exited, slot);
        }
        K1Q_CLOCK(9)
        HARE_K1Q_PHASE_FENCE();
        if (wide && nC > 0) {
            K1Q_STAT(6, true)
            // -------------------------------------------------------------- the WIDE pre-cull of the drain
            // The launch ends with its longest chains, and in the drain those are list scans: a ray in a voxel with a hundred entries
            // needs a dozen cull tasks of eight candidates, each task five dependent round trips (tools/round_trace.py: ~9 us per round,
            // the cull queue 49 -> 23 -> 13 -> 8 -> 4 -> 3 rays).  With few rays left the lanes are free: every ray of the queue gets
            // G = 2 .. 16 lanes, lane `sub` takes the four entries q + 4 sub .. + 3 -- list entries first, then all four records at once,
            // two round trips -- and the group's first survivor IN LIST ORDER goes to the exact phase, exactly the candidate the
            // sequential scan would have stopped at (the pre-cull has no state: it depends on the ray and the polygon only; skipping
            // the polygon tested last stays exact for the rest of the ray's life, see the mailbox note below).  L_idx is written for
            // a survivor only and L_nexti not at all: the exact phase re-reads its successor from the list while `wide` holds, and the
            // sequential cull never runs again (`wide` is monotonic).
            const unsigned nq = nC;                                         // <= HARE_K1Q_WIDE_MAX <= 64: the whole queue, every round
            unsigned gsh = 0;
            while (gsh < 4u && (nq << (gsh + 1u)) <= 64u) ++gsh;
            const unsigned G = 1u << gsh, grp = lane >> gsh, sub = lane & (G - 1u);
            const bool act = grp < nq;
            const unsigned slot = Q_cull[(hC + (act ? grp : 0u)) & SM];
            hC = (hC + nq) & SM;
            nC = 0;
            bool s0 = false, s1 = false, s2 = false, s3 = false;
            int it0 = 0, it1 = 0, it2 = 0, it3 = 0;
            unsigned q = 0, qe = 1;
            uint32_t xf = 0;
            if (act) {
                const unsigned ray = L_ray[slot];
                q = L_q[slot];
                qe = L_qe[slot] & ~QE_HERE;
                xf = L_xyzf[slot];
                const int done1 = L_d1[slot];
                int e1 = -1, e2 = -1;
                if (io.excl1) e1 = io.excl1[ray];
                if (io.excl2) e2 = io.excl2[ray];
                const RayRec r = io.rays[ray];
                double ox = r.x, oy = r.y, oz = r.z;
                if ((xf & F_MOVED) && !writeback) {
                    const double ts = reinterpret_cast<const double*>(&io.out[ray])[1];
                    ox = ox + r.dx * ts; oy = oy + r.dy * ts; oz = oz + r.dz * ts;
                }
                const CullRay cray = cull_ray(g, ox, oy, oz, r.dx, r.dy, r.dz);
                const unsigned k0 = q + 4u * sub, last = qe - 1u;           // q < qe for every ray in the cull queue
                it0 = g.items[k0 < last ? k0 : last];
                it1 = g.items[k0 + 1u < last ? k0 + 1u : last];
                it2 = g.items[k0 + 2u < last ? k0 + 2u : last];
                it3 = g.items[k0 + 3u < last ? k0 + 3u : last];
                const CullRaw r0 = cull_load<QUADS ? 1 : 0>(g, it0), r1 = cull_load<QUADS ? 1 : 0>(g, it1), r2 = cull_load<QUADS ? 1 : 0>(g, it2), r3 = cull_load<QUADS ? 1 : 0>(g, it3);
                auto keep = [&](unsigned k, int it, const CullRaw& rr) {
                    return k < qe && it != e1 && it != e2 && !(HARE_K1Q_MAILBOX && it == done1) && !cull_test<QUADS ? 1 : 0>(g, cray, rr);
                };
                s0 = keep(k0, it0, r0);
                s1 = keep(k0 + 1u, it1, r1);
                s2 = keep(k0 + 2u, it2, r2);
                s3 = keep(k0 + 3u, it3, r3);
                if (OWN) { const unsigned m4 = k0 >= qe ? 0u : (qe - k0 < 4u ? qe - k0 : 4u); own.entries += m4; own.culls += m4; }
            }
            const unsigned long long b0 = __ballot(s0), b1 = __ballot(s1), b2 = __ballot(s2), b3 = __ballot(s3);
            const unsigned sh = grp << gsh;
            const unsigned gm = (1u << G) - 1u;                             // G <= 16
            const unsigned m0 = (unsigned)(b0 >> sh) & gm, m1 = (unsigned)(b1 >> sh) & gm, m2 = (unsigned)(b2 >> sh) & gm, m3 = (unsigned)(b3 >> sh) & gm;
            unsigned first = 0xFFFFu;                                       // the group's first survivor, as an offset from q
            if (m0) first = 4u * (unsigned)__builtin_ctz(m0);
            if (m1) { const unsigned f = 4u * (unsigned)__builtin_ctz(m1) + 1u; first = f < first ? f : first; }
            if (m2) { const unsigned f = 4u * (unsigned)__builtin_ctz(m2) + 2u; first = f < first ? f : first; }
            if (m3) { const unsigned f = 4u * (unsigned)__builtin_ctz(m3) + 3u; first = f < first ? f : first; }
            const bool found = first != 0xFFFFu;
            const unsigned window = (qe - q) < 4u * G ? (qe - q) : 4u * G;
            const unsigned q_new = q + (found ? first : window);
            if (act && found && sub == (first >> 2)) {                      // the lane that holds the survivor: the exact phase tests L_idx
                const unsigned j = first & 3u;
                L_idx[slot] = j == 0u ? it0 : (j == 1u ? it1 : (j == 2u ? it2 : it3));
            }
            const bool lead = act && sub == 0u;
            if (lead) L_q[slot] = q_new;
            const bool exhausted = !found && q_new >= qe;
            push(Q_walk, hW, nW, lead && exhausted && !(xf & F_HIT), slot);
            push(Q_cull, hC, nC, lead && !found && !exhausted, slot);
            push(Q_exact, hE, nE, lead && found, slot);
            push(Q_pend, hP, nP, lead && exhausted && (xf & F_HIT) != 0u, slot);
        } else if (tail ? nC > 0 : sel == 2) {
            // -------------------------------------------------------------- FP32 pre-cull, 2 x CULL_PAIRS candidates per ray at most
            bool act;
            const unsigned slot = pop(Q_cull, hC, nC, act);
            K1Q_STAT(2, act)
            bool to_walk = false, to_cull = false, to_exact = false, to_pend = false, done_here = false;
            int share_idx = -1;
            if (act) {
                const unsigned ray = L_ray[slot];
                unsigned q = L_q[slot];
                const unsigned qe_word = L_qe[slot];
                const unsigned qe = qe_word & ~QE_HERE;
                int idx = L_idx[slot], nexti = L_nexti[slot], done1 = L_d1[slot];
                const uint32_t xf = L_xyzf[slot];
                int e1 = -1, e2 = -1;
                if (io.excl1) e1 = io.excl1[ray];                           // poly_origin1 / 2 (Voxel_Grid.cs:477); wave-uniform branches
                if (io.excl2) e2 = io.excl2[ray];
                const RayRec r = io.rays[ray];
                double ox = r.x, oy = r.y, oz = r.z;
                if ((xf & F_MOVED) && !writeback) {
                    const double ts = reinterpret_cast<const double*>(&io.out[ray])[1];
                    ox = ox + r.dx * ts; oy = oy + r.dy * ts; oz = oz + r.dz * ts;
                }
                const CullRay cray = cull_ray(g, ox, oy, oz, r.dx, r.dy, r.dz);
                bool culling = true, parked = false;
                share_idx = idx;
#if HARE_K1Q_CULL_AHEAD
                // All of the task's list entries FIRST (idx, nexti are here; the eight after them in four 8-byte gathers, their window
                // sliding back at the end of the list so that it stays inside it), then all eight pre-cull records, then the
                // sequential scan on the eight results: three rounds of dependent loads per task (slot state -> ray record + entries ->
                // records) where the pair-by-pair form below had six.  The same gathers as before -- the straight-line pairs
                // requested theirs whether or not the scan had ended -- and the same candidates, in the same order, kept or dropped by
                // the same rules.
                constexpr int NC = 2 * HARE_K1Q_CULL_PAIRS;
                int E[NC + 2];
                E[0] = idx;
                E[1] = (q + 1u < qe && nexti >= 0) ? nexti : idx;
                {
                    const unsigned left = qe - q;                                // >= 1
#pragma unroll
                    for (int m = 0; m < NC / 2; ++m) {
                        const unsigned k0 = 2u + 2u * (unsigned)m;              // entries k0, k0 + 1 of the task
                        if (left >= 2u) {
                            const unsigned a = q + k0 + 1u < qe ? q + k0 : qe - 2u;     // the pair's window, inside [q, qe)
                            const int2 w = *reinterpret_cast<const int2*>(g.items + a);
                            E[k0] = (q + k0 + 1u == qe) ? w.y : w.x;            // window slid back by one: the entry wanted is its second
                            E[k0 + 1] = w.y;
                        } else {
                            E[k0] = idx; E[k0 + 1] = idx;
                        }
                    }
                }
                bool T[NC];
#ifndef HARE_K1Q_CULL_BATCH
#define HARE_K1Q_CULL_BATCH (NC % 4 == 0 ? 4 : 2)    // records requested together (8 VGPRs each): 4 measured better than 8 (registers) and 2
#endif
#ifndef HARE_K1Q_CULL_BATCH_QUADS
#define HARE_K1Q_CULL_BATCH_QUADS HARE_K1Q_CULL_BATCH   // the 48-byte records of a topology with quadrilaterals: 12 VGPRs each
#endif
                constexpr int NB = QUADS ? HARE_K1Q_CULL_BATCH_QUADS : HARE_K1Q_CULL_BATCH;
#pragma unroll
                for (int b0 = 0; b0 < NC; b0 += NB) {
                    CullRaw R[NB];
#pragma unroll
                    for (int k = 0; k < NB; ++k) R[k] = cull_load<QUADS ? 1 : 0>(g, E[b0 + k]);
#pragma unroll
                    for (int k = 0; k < NB; ++k) T[b0 + k] = cull_test<QUADS ? 1 : 0>(g, cray, R[k]);
                    if (NB < NC) __builtin_amdgcn_sched_barrier(0);          // keep the next batch's requests behind this batch's tests (registers)
                }
                unsigned consumed = 0;
#pragma unroll
                for (int k = 0; k < NC; ++k) {
                    const bool valid = q + (unsigned)k < qe;
                    const int c = E[k];
                    const bool sk = c == e1 || c == e2 || (HARE_K1Q_MAILBOX && c == done1);   // Voxel_Grid.cs:477 + the mailbox (see below)
                    const bool keep = culling && valid && !sk && !T[k];         // survives: it goes to the exact phase
                    const bool step = culling && valid && !keep;                // consumed
                    done1 = (step && !sk) ? c : done1;                          // a certain miss counts as tested
                    consumed += step ? 1u : 0u;
                    if (OWN) { own.entries += (step || keep) ? 1u : 0u; own.culls += ((step || keep) && !sk) ? 1u : 0u; }
                    parked = parked || keep;
                    culling = culling && valid && !keep;
                }
                q += consumed;
                culling = culling && q < qe;
                {
                    int ni = E[0], nn = E[1];
#pragma unroll
                    for (int k = 1; k <= NC; ++k) {
                        ni = consumed == (unsigned)k ? E[k] : ni;
                        nn = consumed == (unsigned)k ? E[k + 1] : nn;
                    }
                    idx = ni; nexti = nn;
                }
#else
#pragma unroll
                for (int kp = 0; kp < HARE_K1Q_CULL_PAIRS; ++kp) {
                    // candidates idx (at q) and nexti (at q + 1): both records and the two list entries after them are
                    // requested together; everything below is straight-line selects (a lane that is done computes on its
                    // stale, still valid indices and keeps nothing)
                    const bool has1 = q + 1 < qe;
                    // the two list entries after the pair: ONE 8-byte gather (4-byte aligned is all the hardware asks for); near
                    // the end of the list the window slides back so that it stays inside it
                    int i2, i3;
                    if (qe - q >= 4u) {
                        const int2 w = *reinterpret_cast<const int2*>(g.items + q + 2);
                        i2 = w.x; i3 = w.y;
                    } else {
                        const unsigned qa = q + 2 < qe ? q + 2 : qe - 1;
                        i2 = g.items[qa]; i3 = i2;                           // q + 3 >= qe: i3 is never a candidate
                    }
                    const int ia = idx >= 0 ? idx : 0, ib = (has1 && nexti >= 0) ? nexti : ia;   // a finished lane may hold -1: stay inside the array
                    const CullRaw ra = cull_load<QUADS ? 1 : 0>(g, ia), rb = cull_load<QUADS ? 1 : 0>(g, ib);
                    const bool ca = cull_test<QUADS ? 1 : 0>(g, cray, ra);
                    const bool cb = cull_test<QUADS ? 1 : 0>(g, cray, rb);
                    // Re-testing a polygon can never change the result (strict `t < tmin`), so skipping the one this ray
                    // tested last is exact (Voxel_Grid.cs:477 + K1p's register mailbox)
                    const bool sk0 = idx == e1 || idx == e2 || (HARE_K1Q_MAILBOX && idx == done1);
                    const bool keep0 = culling && !sk0 && !ca;              // candidate 0 survives: it goes to the exact phase
                    const bool step0 = culling && !keep0;                   // candidate 0 consumed
                    const int dn0 = (step0 && !sk0) ? idx : done1;          // a certain miss counts as tested
                    const bool go1 = step0 && has1;
                    const bool sk1 = nexti == e1 || nexti == e2 || (HARE_K1Q_MAILBOX && nexti == dn0);
                    const bool keep1 = go1 && !sk1 && !cb;
                    const bool step1 = go1 && !keep1;
                    done1 = (step1 && !sk1) ? nexti : dn0;
                    q += (step0 ? 1u : 0u) + (step1 ? 1u : 0u);
                    const int nidx = step1 ? i2 : (step0 ? nexti : idx);
                    const int nnext = step1 ? i3 : (step0 ? i2 : nexti);
                    idx = nidx; nexti = nnext;
                    parked = parked || keep0 || keep1;
                    culling = culling && !keep0 && !keep1 && q < qe;
                }
#endif
                L_q[slot] = q; L_idx[slot] = idx; L_nexti[slot] = nexti; L_d1[slot] = done1;
                to_exact = parked;
                to_cull = culling;                                          // quota used up, list not exhausted
                const bool exhausted = !parked && !culling;
                to_pend = exhausted && (xf & F_HIT);
                to_walk = exhausted && !(xf & F_HIT);
                if (!BOUNCE && to_pend && (qe_word & QE_HERE) && !(xf & F_MOVED)) {
                    // the list is scanned and the pending hit lies in this voxel (QE_HERE, above): Voxel_Grid.cs:705-709 returns it.  The event
                    // slot holds t, the point, {Hit, Poly_id} since the accept; u and v are the record's (0, 0)
                    double* sc = reinterpret_cast<double*>(&io.out[ray]);
                    sc[1] = 0.0;
                    sc[2] = 0.0;
                    nhits++;
                    to_pend = false;
                    done_here = true;
                }
            }
            if (__builtin_expect((io.flags & 0x3000u) == 0x3000u && io.prof != nullptr, 0)) {
                // developer statistic (tools/share_stat.py; both developer bits, so that the timeline alone stays cheap): how many DIFFERENT polygons do the lanes of one cull batch look at?
                // (what a wave-shared LDS tile of polygon records could save: lanes - distinct record fetches)
                unsigned long long left = __ballot(act);
                unsigned distinct = 0;
                const unsigned lanes = (unsigned)__popcll(left);
                while (left) {
                    const int v = __builtin_amdgcn_readlane(share_idx, (int)__builtin_ctzll(left));
                    left &= ~__ballot(act && share_idx == v);
                    ++distinct;
                }
                stat_lanes += lanes; stat_distinct += distinct; stat_batches += 1;
            }
            push(Q_walk, hW, nW, to_walk, slot);
            push(Q_cull, hC, nC, to_cull, slot);
            push(Q_exact, hE, nE, to_exact, slot);
            push(Q_pend, hP, nP, to_pend, slot);
            push(Q_free, hF, nF, done_here, slot);
        }
        K1Q_CLOCK(9)
        HARE_K1Q_PHASE_FENCE();
        if (tail ? nE > 0 : sel == 0) {
            // -------------------------------------------------------------- exact FP64 test of one candidate per ray
            bool act;
            const unsigned slot = pop(Q_exact, hE, nE, act);
            K1Q_STAT(3, act)
            bool to_walk = false, to_cull = false, to_pend = false, done_here = false;
            if (act) {
                const unsigned ray = L_ray[slot];
                const int i = L_idx[slot];
                unsigned q = L_q[slot];
                unsigned qe_word = L_qe[slot];
                const unsigned qe = qe_word & ~QE_HERE;
                uint32_t xf = L_xyzf[slot];
                double* sc = reinterpret_cast<double*>(&io.out[ray]);
                const unsigned qa = q + 2 < qe ? q + 2 : qe - 1;
                const int after = g.items[qa];                              // items[q + 2]: nexti once this candidate is done
                // items[q + 1]: in the pool's slot state, except behind the wide cull, which leaves only L_idx (the candidate) current
                const int succ = wide ? g.items[q + 1u < qe ? q + 1u : qe - 1u] : L_nexti[slot];
                const double tmin = (xf & F_HIT) ? sc[0] : kDblMax;             // Voxel_Grid.cs:688: tmin starts at double.MaxValue
                const RayRec r = io.rays[ray];
                const PolyRec& p = g.polys[i];
                const double v0[3] = {p.v0[0], p.v0[1], p.v0[2]}, v1[3] = {p.v1[0], p.v1[1], p.v1[2]};
                const double v2[3] = {p.v2[0], p.v2[1], p.v2[2]}, nn[3] = {p.n[0], p.n[1], p.n[2]};
                double q3x = 0, q3y = 0, q3z = 0;
                int qnv = 3;
                if (QUADS) {
                    if (g.quads) {
                        const QuadRec& qr = g.quads[i];
                        q3x = qr.v3[0]; q3y = qr.v3[1]; q3z = qr.v3[2];
                        qnv = qr.nverts;
                    }
                }
                V3 o = {r.x, r.y, r.z};
                const V3 d = {r.dx, r.dy, r.dz};
                if ((xf & F_MOVED) && !writeback) {                         // AABB_Main.cs:254-256, same expression => same bits
                    const double ts = sc[1];
                    o.x = o.x + d.x * ts; o.y = o.y + d.y * ts; o.z = o.z + d.z * ts;
                }
                const bool side = ray_side(d, nn);                          // Polygons.cs:641-648
                double a[3], c[3];
#pragma unroll
                for (int m = 0; m < 3; ++m) { a[m] = side ? v0[m] : v2[m]; c[m] = side ? v2[m] : v0[m]; }
                double t = 0;
                if (OWN) own.tests++;
                bool ok = tri_fast(o, d, a, v1, c, t);
                if (QUADS) {
                    const double v3[3] = {q3x, q3y, q3z};
                    if (!ok && qnv == 4) ok = tri_fast(o, d, c, v3, a, t);     // (P2,P3,P0) / (P0,P3,P2)
                }
                if (ok && t > kTMin && t < tmin) {                              // Voxel_Grid.cs:691-693
                    const double hx = o.x + d.x * t, hy = o.y + d.y * t, hz = o.z + d.z * t;      // X_Point (Polygons.cs:652)
                    sc[0] = t;
                    sc[3] = hx;
                    sc[4] = hy;
                    sc[5] = hz;
                    sc[6] = __hiloint2double(1, i);
                    xf |= F_HIT;
                    L_xyzf[slot] = xf;
                    // ... and is that point in the voxel the ray is in (QE_HERE, above)?
                    const bool here = point_in_voxel((int)(xf & 511u), (int)((xf >> 9) & 511u), (int)((xf >> 18) & 511u), hx, hy, hz);
                    qe_word = here ? (qe_word | QE_HERE) : (qe_word & ~QE_HERE);
                    L_qe[slot] = qe_word;
                }
                L_d1[slot] = i;
                // next_candidate()
                ++q;
                L_idx[slot] = succ;
                L_nexti[slot] = after;
                L_q[slot] = q;
                to_cull = q < qe;
                to_pend = !to_cull && (xf & F_HIT);
                to_walk = !to_cull && !to_pend;
                if (!BOUNCE && to_pend && (qe_word & QE_HERE) && !(xf & F_MOVED)) {     // the list ends with this candidate and the hit lies here: finished (see the cull phase)
                    sc[1] = 0.0;
                    sc[2] = 0.0;
                    nhits++;
                    to_pend = false;
                    done_here = true;
                }
            }
            push(Q_walk, hW, nW, to_walk, slot);
            push(Q_cull, hC, nC, to_cull, slot);
            push(Q_pend, hP, nP, to_pend, slot);
            push(Q_free, hF, nF, done_here, slot);
        }
        K1Q_CLOCK(9)
        HARE_K1Q_PHASE_FENCE();      // the exact phase's hit record, before the pending-hit walk reads it
        if (tail ? nP > 0 : sel == 1) {
            // -------------------------------------------------------------- walk with a pending hit (Voxel_Grid.cs:705-759)
            bool act;
            const unsigned slot = pop(Q_pend, hP, nP, act);
            K1Q_STAT(4, act)
            bool to_cull = false, freed = false;
            bool walking = act;
            double tMaxX = 0, tMaxY = 0, tMaxZ = 0, tDeltaX = 0, tDeltaY = 0, tDeltaZ = 0, tmin = 0, hx = 0, hy = 0, hz = 0;
            int X = 0, Y = 0, Z = 0, dx1 = 1, dy1 = 1, dz1 = 1;
            unsigned ray = 0;
            uint32_t xf = 0;
            if (act) {
                tMaxX = L_tmx[slot]; tMaxY = L_tmy[slot]; tMaxZ = L_tmz[slot];
                tDeltaX = L_tdx[slot]; tDeltaY = L_tdy[slot]; tDeltaZ = L_tdz[slot];
                xf = L_xyzf[slot];
                ray = L_ray[slot];
                X = (int)(xf & 511u); Y = (int)((xf >> 9) & 511u); Z = (int)((xf >> 18) & 511u);
                dx1 = (xf & F_NX) ? -1 : 1; dy1 = (xf & F_NY) ? -1 : 1; dz1 = (xf & F_NZ) ? -1 : 1;
                const double* sc = reinterpret_cast<const double*>(&io.out[ray]);
                tmin = sc[0]; hx = sc[3]; hy = sc[4]; hz = sc[5];
            }
            bool exited = false;
            const int pend_steps = tail ? HARE_K1Q_TAIL_STEPS : HARE_K1Q_WALK_STEPS;
#pragma unroll 1
            for (int k = 0; k < pend_steps; ++k) {
                if (__ballot(walking) == 0) break;
#ifdef HARE_K1Q_STATS
                kq_n[8]++; kq_l[8] += (unsigned long long)__popcll(__ballot(walking));
#endif
                if (walking) {
                    // Voxel_Grid.cs:705: hit point inside the CURRENT padded voxel?
                    const double lox = voxel_lo(X, g.vd[0], g.omin[0]), hix = voxel_hi(X, g.vd[0], g.omin[0]);
                    const double loy = voxel_lo(Y, g.vd[1], g.omin[1]), hiy = voxel_hi(Y, g.vd[1], g.omin[1]);
                    const double loz = voxel_lo(Z, g.vd[2], g.omin[2]), hiz = voxel_hi(Z, g.vd[2], g.omin[2]);
                    const bool in = !(hx < lox) && !(hy < loy) && !(hz < loz) && !(hx > hix) && !(hy > hiy) && !(hz > hiz);
                    if (in) {
                        freed = true;
                        walking = false;
                    } else {
                        HARE_K1Q_STEP();
                        const bool out = ((unsigned)X >= (unsigned)ct) | ((unsigned)Y >= (unsigned)ct) | ((unsigned)Z >= (unsigned)ct);
                        const int cell = out ? 0 : (X * ct + Y) * ct + Z;
                        const bool occ = occupied(out ? 0 : X, out ? 0 : Y, out ? 0 : Z, cell);
                        if (OWN && !out) own.cells++;
                        exited = out;                                       // leaving the grid: miss, even with the hit pending (F12)
                        to_cull = !out && occ;
                        walking = !out && !occ;
                    }
                }
            }
            if (freed) {                                                    // the hit stands: Voxel_Grid.cs:707
                double* sc = reinterpret_cast<double*>(&io.out[ray]);
                double t_start = 0;
                if (xf & F_MOVED) t_start = sc[1];
                sc[0] = tmin + t_start;
                sc[1] = ((xf & F_MOVED) && (io.flags & SHOOT_SLIM_EVENTS)) ? tmin : 0.0;     // u: 0 as the reference returns it (see SHOOT_SLIM_EVENTS)
                sc[2] = 0;
                nhits++;
                if (BOUNCE) atomicAdd(&C_hits[L_cast[slot]], 1u);
            }
            const bool rearm = BOUNCE && freed && !exited;                 // the ray goes on to its next cast (or hands its event over): rearm phase
            if (exited) {
                store_miss(ray);
                freed = true;
            }
            unsigned q = 0, qe = 0;
            int idx = -1, nexti = -1;
            if (to_cull) {
                const CellRec c = g.cells[(X * ct + Y) * ct + Z];
                q = c.start; qe = c.start + c.count; idx = c.i0; nexti = c.i1;
                if (COARSE && c.count == 0) { to_cull = false; walking = true; }   // the block is occupied, this voxel is not
            }
            if (act && !freed) {
                L_tmx[slot] = tMaxX; L_tmy[slot] = tMaxY; L_tmz[slot] = tMaxZ;
                L_xyzf[slot] = (xf & 0xF8000000u) | (uint32_t)X | ((uint32_t)Y << 9) | ((uint32_t)Z << 18);
                if (to_cull) { L_q[slot] = q; L_qe[slot] = qe; L_idx[slot] = idx; L_nexti[slot] = nexti; }
            }
            push(Q_pend, hP, nP, walking, slot);
            push(Q_cull, hC, nC, to_cull, slot);
            push(Q_free, hF, nF, freed && !rearm, slot);
            if (BOUNCE) push(Q_rearm, hR, nR, rearm, slot);
        }
    }
    t_coop = __builtin_amdgcn_s_memrealtime();       // developer timeline: when the pool rounds ended
    if (coop && nW + nC + nE + nP > 0) {
        // ---- the cooperative tail: what is left (at most HARE_K1Q_COOP_MAX rays, in whatever queue) is traced by the whole wave,
        // one ray after the other, from the state the pool left it in
        HARE_K1Q_TAIL_FENCE();
        const uint8_t* const Qs[4] = {Q_walk, Q_cull, Q_exact, Q_pend};
        unsigned* const heads[4] = {&hW, &hC, &hE, &hP};
        unsigned* const cnts[4] = {&nW, &nC, &nE, &nP};
#pragma unroll
        for (int qn = 0; qn < 4; ++qn) {
            while (*cnts[qn] > 0) {
                const unsigned slot = Qs[qn][*heads[qn] & SM];                     // wave-uniform: every lane reads the same queue entry
                *heads[qn] = (*heads[qn] + 1u) & SM;
                *cnts[qn] -= 1u;
                const unsigned ray = L_ray[slot];
                const uint32_t xf = L_xyzf[slot];
                double tmin = kDblMax;
                int pid = -1;
                double* sc = reinterpret_cast<double*>(&io.out[ray]);              // the ray's scratch: this wave wrote it
                if (xf & F_HIT) { tmin = sc[0]; pid = __double2loint(sc[6]); }
                const bool hit = coop_trace<QUADS, COARSE>(g, io, locc, ray, xf, L_tmx[slot], L_tmy[slot], L_tmz[slot], tmin, pid, OWN ? &own : nullptr);
                ++helped;
                if (lane == 0) {
                    XEventRec ev;
                    if (hit) {
                        const RayRec r = io.rays[ray];
                        double t_start = 0, ox = r.x, oy = r.y, oz = r.z;
                        if (xf & F_MOVED) {                                        // the set-up left t_start in the scratch; o' = o + d * t_start (AABB_Main.cs:254-256)
                            t_start = sc[1];
                            ox = ox + r.dx * t_start; oy = oy + r.dy * t_start; oz = oz + r.dz * t_start;
                        }
                        ev.t = tmin + t_start;                                     // Voxel_Grid.cs:707
                        ev.u = ((xf & F_MOVED) && (io.flags & SHOOT_SLIM_EVENTS)) ? tmin : 0.0;
                        ev.v = 0;
                        ev.x = ox + r.dx * tmin; ev.y = oy + r.dy * tmin; ev.z = oz + r.dz * tmin;      // Polygons.cs:652
                        ev.poly_id = pid;
                        ev.hit = 1;
                        nhits++;
                        if (BOUNCE) atomicAdd(&C_hits[L_cast[slot]], 1u);
                    } else {
                        set_miss(ev);
                    }
                    if (BOUNCE && hit) io.out[ray] = ev;                         // read back by the rearm phase: plain stores, through the L1
                    else store_event_streaming(&io.out[ray], ev);
                    if (BOUNCE && hit) Q_rearm[(hR + nR) & SM] = (uint8_t)slot;  // the hit stands: on to the ray's next cast
                }
                if (BOUNCE && hit) nR += 1u;
            }
        }
    }
    if (!BOUNCE || nR == 0) break;
    HARE_K1Q_TAIL_FENCE();           // lane 0's rearm entries and events, before the rearm phase reads them
    }
#undef HARE_K1Q_PHASE_FENCE
#undef HARE_K1Q_TAIL_FENCE
    if (BOUNCE && io.ctr_casts) {
        // rays started / hits per cast: this wave's share, added to the caller's per-cast blocks
        HARE_K1Q_FINAL_FENCE();
        if (lane < (unsigned)n_casts) {
            const uint32_t r = C_rays[lane], h = C_hits[lane];
            if (r) atomicAdd(&io.ctr_casts[(size_t)lane * CTR_WORDS + CTR_RAYS], (unsigned long long)r);
            if (h) atomicAdd(&io.ctr_casts[(size_t)lane * CTR_WORDS + CTR_HITS], (unsigned long long)h);
        }
    }
#ifdef HARE_K1Q_STATS
    if (lane == 0 && io.prof) {
        for (int k = 0; k < 8; ++k) { atomicAdd(&io.prof[2 * k], kq_n[k]); atomicAdd(&io.prof[2 * k + 1], kq_l[k]); }
        atomicAdd(&io.prof[16], (unsigned long long)rounds_done);
        atomicAdd(&io.prof[18], kq_n[8]); atomicAdd(&io.prof[19], kq_l[8]);
        K1Q_CLOCK(9)
        for (int k = 0; k < 10; ++k) atomicAdd(&io.prof[20 + k], kq_t[k]);
    }
    (void)kq_steps;
#endif
    timeline(2, __builtin_amdgcn_s_memrealtime());
    timeline(3, (rounds_done & 0xFFFFu) | ((unsigned long long)(helped & 0xFFu) << 16) | (t_coop << 24));
    if (__builtin_expect((io.flags & 0x3000u) == 0x3000u && io.prof != nullptr, 0)) {
        if (lane == 0) {
            atomicAdd(&io.prof[0], stat_lanes);
            atomicAdd(&io.prof[1], stat_distinct);
            atomicAdd(&io.prof[2], stat_batches);
        }
    }

    if (OWN) flush_own(io.ctr, own);
    launch_epilogue(io, nrays, nhits, (unsigned)kPoolWaves);     // batch counters + the launch slot left zeroed
}

}  // namespace

extern "C" {
__global__ __launch_bounds__(64 * HARE_K1Q_WAVES) void hare_voxel_pool_tri(VoxelArgs g, ShootIO io) { voxel_pool_body<false, false>(g, io); }
__global__ __launch_bounds__(64 * HARE_K1Q_WAVES) void hare_voxel_pool_quad(VoxelArgs g, ShootIO io) { voxel_pool_body<true, false>(g, io); }
__global__ __launch_bounds__(64 * HARE_K1Q_WAVES) void hare_voxel_pool_tri_g(VoxelArgs g, ShootIO io) { voxel_pool_body<false, true>(g, io); }
__global__ __launch_bounds__(64 * HARE_K1Q_WAVES) void hare_voxel_pool_quad_g(VoxelArgs g, ShootIO io) { voxel_pool_body<true, true>(g, io); }
// the counting builds (HARE_SHOOT_COUNT_OWN): the same events, plus the kernel's own voxels / list entries / pre-culls / exact tests
__global__ __launch_bounds__(64 * HARE_K1Q_WAVES) void hare_voxel_pool_tri_own(VoxelArgs g, ShootIO io) { voxel_pool_body<false, false, false, true>(g, io); }
__global__ __launch_bounds__(64 * HARE_K1Q_WAVES) void hare_voxel_pool_quad_own(VoxelArgs g, ShootIO io) { voxel_pool_body<true, false, false, true>(g, io); }
__global__ __launch_bounds__(64 * HARE_K1Q_WAVES) void hare_voxel_pool_tri_g_own(VoxelArgs g, ShootIO io) { voxel_pool_body<false, true, false, true>(g, io); }
__global__ __launch_bounds__(64 * HARE_K1Q_WAVES) void hare_voxel_pool_quad_g_own(VoxelArgs g, ShootIO io) { voxel_pool_body<true, true, false, true>(g, io); }
// the whole specular bounce loop of every ray in one launch (BOUNCE, above): dynamic LDS = bitmap + waves x (kPoolWaveBytes + kPoolBounceExtra)
__global__ __launch_bounds__(64 * HARE_K1Q_WAVES) void hare_voxel_bounce_tri(VoxelArgs g, ShootIO io) { voxel_pool_body<false, false, true>(g, io); }
__global__ __launch_bounds__(64 * HARE_K1Q_WAVES) void hare_voxel_bounce_quad(VoxelArgs g, ShootIO io) { voxel_pool_body<true, false, true>(g, io); }
__global__ __launch_bounds__(64 * HARE_K1Q_WAVES) void hare_voxel_bounce_tri_g(VoxelArgs g, ShootIO io) { voxel_pool_body<false, true, true>(g, io); }
__global__ __launch_bounds__(64 * HARE_K1Q_WAVES) void hare_voxel_bounce_quad_g(VoxelArgs g, ShootIO io) { voxel_pool_body<true, true, true>(g, io); }
}
