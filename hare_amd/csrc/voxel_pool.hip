// voxel_pool.hip -- K1q: Voxel_Grid.Shoot with MORE RAYS THAN LANES (included by kernels.hip).
//
// Why: in K1p (hare_voxel_persist_*) a lane owns one ray, and a ray alternates between walking (~10 empty voxels),
// culling (~3 candidates) and, rarely, an exact test -- so whatever phase the wave executes, about half of its
// lanes hold a ray that is in another phase (measured: 40 % lane occupancy, 2.9x the instructions the work needs).
// Here a wave owns a POOL of SLOTS rays (2 per lane) whose state lives in LDS, and three queues of slot numbers --
// rays that have to walk, rays that hold candidates to cull, rays whose candidate survived the cull and needs the
// exact test.  Each round the wave picks the fullest queue, pops up to 64 rays, and runs that ONE phase on them at
// (nearly) full lane occupancy; rays move between queues as their phase changes, finished rays free their slot and
// new rays are set up 64 at a time.  Pools and queues are private to a wave: no atomics, no barriers, no
// inter-wave protocol -- queue heads and counts are wave-uniform scalars.
//
// A ray's own sequence of operations is exactly K1p's (and therefore the reference's, Voxel_Grid.cs:561-761): same
// cells in the same order, same candidate order, FP32 pre-cull in front of the exact RayXtri, pending-hit /
// IsPointInBox rule, miss on grid exit; only the interleaving BETWEEN rays differs.
//
// LDS per workgroup: the occupancy bitmap (<= 64 KB) + per wave SLOTS x 116 B (7 doubles, 13 words, 4 queue entries).
// o and d are NOT kept: the phases that need them (cull, exact) re-read the 48-byte ray record, which is
// cache-resident for the ray's short life.  The ray's own X_Event slot is its scratch until it finishes: the pending
// hit point, and for a ray whose origin AABB.Intersect moved, t_start and the moved origin.
#ifndef HARE_K1Q_WALK_STEPS
#define HARE_K1Q_WALK_STEPS 16    // DDA steps per walk task at most
#endif
#ifndef HARE_K1Q_WALK_MIN
#define HARE_K1Q_WALK_MIN 20      // a walk task ends early when fewer lanes than this are still walking
#endif
#ifndef HARE_K1Q_CULL_PAIRS
#define HARE_K1Q_CULL_PAIRS 2     // pairs of candidates per cull task
#endif
#ifndef HARE_K1Q_EXACT_MIN
#define HARE_K1Q_EXACT_MIN 24     // run the exact phase when this many rays wait for it (or nothing else can run)
#endif
#ifndef HARE_K1Q_REFILL_MIN
#define HARE_K1Q_REFILL_MIN 32    // set up new rays when this many slots are free
#endif

namespace {

template <bool QUADS, bool COARSE>
__device__ __forceinline__ void voxel_pool_body(const VoxelArgs& g, const ShootIO& io)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    constexpr unsigned S = kPoolSlots, SM = kPoolSlots - 1;
    uint32_t* const locc = reinterpret_cast<uint32_t*>(lds_raw);
    const int nw4 = (g.occ_words + 3) >> 2;
    {
        const uint4* src = reinterpret_cast<const uint4*>(g.occ);
        uint4* dst = reinterpret_cast<uint4*>(locc);
        for (int k = threadIdx.x; k < nw4; k += blockDim.x) dst[k] = src[k];
    }
    const int wave = threadIdx.x >> 6;
    const unsigned lane = threadIdx.x & 63;
    const unsigned long long lane_lt = (1ull << lane) - 1ull;
    unsigned char* const wb = lds_raw + ((size_t)nw4 << 4) + (size_t)wave * kPoolWaveBytes;
    double* const L_tmx = reinterpret_cast<double*>(wb);
    double* const L_tmy = L_tmx + S;
    double* const L_tmz = L_tmy + S;
    double* const L_tdx = L_tmz + S;
    double* const L_tdy = L_tdx + S;
    double* const L_tdz = L_tdy + S;
    double* const L_tmin = L_tdz + S;
    uint32_t* const L_ray = reinterpret_cast<uint32_t*>(L_tmin + S);
    uint32_t* const L_xyz = L_ray + S;        // X | Y << 10 | Z << 20 | (dx<0) << 30 | (dy<0) << 31 ; (dz<0) is in L_flags
    int32_t* const L_cell = reinterpret_cast<int32_t*>(L_xyz + S);
    uint32_t* const L_q = reinterpret_cast<uint32_t*>(L_cell + S);
    uint32_t* const L_qe = L_q + S;
    int32_t* const L_idx = reinterpret_cast<int32_t*>(L_qe + S);
    int32_t* const L_nexti = L_idx + S;
    int32_t* const L_pid = L_nexti + S;
    int32_t* const L_e1 = L_pid + S;
    int32_t* const L_e2 = L_e1 + S;
    int32_t* const L_d1 = L_e2 + S;
    int32_t* const L_d2 = L_d1 + S;
    uint32_t* const L_flags = reinterpret_cast<uint32_t*>(L_d2 + S);   // bit0: dz<0, bit1: moved (t_start parked in out[ray].t)
    uint16_t* const Q_walk = reinterpret_cast<uint16_t*>(L_flags + S);
    uint16_t* const Q_cull = Q_walk + S;
    uint16_t* const Q_exact = Q_cull + S;
    uint16_t* const Q_free = Q_exact + S;

    for (unsigned k = lane; k < S; k += 64) Q_free[k] = (uint16_t)k;
    __syncthreads();      // the bitmap is shared by the workgroup; everything after this point is wave-private

    const int ct = g.ct;
    const double fct = (double)ct;
    // wave-uniform queue state
    unsigned hW = 0, nW = 0, hC = 0, nC = 0, hE = 0, nE = 0, hF = 0, nF = S;
    auto push = [&](uint16_t* Q, unsigned head, unsigned& cnt, bool flag, unsigned slot) {
        const unsigned long long m = __ballot(flag);
        if (flag) Q[(head + cnt + (unsigned)__popcll(m & lane_lt)) & SM] = (uint16_t)slot;
        cnt += (unsigned)__popcll(m);
    };
    auto pop = [&](const uint16_t* Q, unsigned& head, unsigned& cnt, unsigned n, bool& active) -> unsigned {
        active = lane < n;
        const unsigned slot = active ? Q[(head + lane) & SM] : 0u;
        head = (head + n) & SM;
        cnt -= n;
        return slot;
    };

    // ray chunks: static first chunk per wave (XCD-contiguous), then tickets -- as in K1p
    const unsigned n32 = (unsigned)io.n;
    const unsigned RAY_CHUNK = 128;
    const unsigned n_static = gridDim.x * (unsigned)kPoolWaves * RAY_CHUNK;
    unsigned chunk_id = blockIdx.x * (unsigned)kPoolWaves + (unsigned)wave;
    if ((gridDim.x & 7u) == 0) chunk_id = ((blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3)) * (unsigned)kPoolWaves + (unsigned)wave;
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
    // hit: X_Point is already in the record (written when the hit was accepted); t_start was parked in .t by the set-up
    auto store_hit = [&](unsigned ray, bool moved, double tmin, int pid) {
        double* q = reinterpret_cast<double*>(&io.out[ray]);
        double t_start = 0;
        if (moved) t_start = q[0];
        q[0] = tmin + t_start;                                      // Voxel_Grid.cs:707
        q[1] = 0;
        q[2] = 0;
        q[6] = __hiloint2double(1, pid);
    };
    auto occupied = [&](int X, int Y, int Z, int cell) -> bool {
        const uint32_t bit = COARSE ? (uint32_t)(((X >> g.occ_shift) * g.occ_cd + (Y >> g.occ_shift)) * g.occ_cd + (Z >> g.occ_shift))
                                    : (uint32_t)cell;
        return (locc[bit >> 5] >> (bit & 31)) & 1u;
    };

    // a wave serves ~n / (waves in the grid) rays in a few rounds each; the cap only exists so that a defect can never
    // turn into a wave that does not finish (rays it left behind would keep their scratch values and fail every parity test)
    for (unsigned round = 0; round < (1u << 24); ++round) {
        // ------------------------------------------------------------------ set-up of new rays into free slots
        if (!drained && (nF >= (unsigned)HARE_K1Q_REFILL_MIN || nW + nC + nE == 0)) {
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
                bool act;
                const unsigned slot = pop(Q_free, hF, nF, m, act);
                const unsigned ray = cn + lane;
                cn += m;
                bool to_walk = false, to_cull = false, freed = false;
                if (act) {
                    // ---------------- per-ray set-up: Voxel_Grid.cs:563-632
                    const RayRec r = io.rays[ray];
                    V3 o = {r.x, r.y, r.z};
                    const V3 d = {r.dx, r.dy, r.dz};
                    const int e1 = io.excl1 ? io.excl1[ray] : -1;
                    const int e2 = io.excl2 ? io.excl2[ray] : -1;
                    bool alive = true, moved = false;
                    if (e1 == -2 && (io.flags & SHOOT_RETIRED_RAYS)) {      // retired by the bounce loop: miss, not counted
                        store_miss(ray);
                        alive = false;
                    } else {
                        nrays++;
                        double fx = floor((o.x - g.omin[0]) / g.vd[0]);
                        double fy = floor((o.y - g.omin[1]) / g.vd[1]);
                        double fz = floor((o.z - g.omin[2]) / g.vd[2]);
                        bool inside = (fx >= 0.0 && fx < fct) & (fy >= 0.0 && fy < fct) & (fz >= 0.0 && fz < fct);
                        if (!inside) {
                            double t_start = 0;
                            if (!aabb_clip_move(g.omin, g.omax, o, d, t_start)) {
                                store_miss(ray);
                                alive = false;
                            } else {
                                moved = true;
                                if (io.flags & SHOOT_WRITEBACK_ORIGIN) {
                                    io.rays[ray].x = o.x; io.rays[ray].y = o.y; io.rays[ray].z = o.z;
                                }
                                fx = floor((o.x - g.omin[0] + d.x * 1E-6) / g.vd[0]);
                                fy = floor((o.y - g.omin[1] + d.y * 1E-6) / g.vd[1]);
                                fz = floor((o.z - g.omin[2] + d.z * 1E-6) / g.vd[2]);
                                inside = (fx >= 0.0 && fx < fct) & (fy >= 0.0 && fy < fct) & (fz >= 0.0 && fz < fct);
                                if (!inside) {
                                    store_miss(ray);
                                    alive = false;
                                } else {
                                    // the ray's own X_Event slot is its scratch until it finishes: .t = t_start (added to the
                                    // hit's t, Voxel_Grid.cs:707), .u .v + the {Poly_id, Hit} word = the moved origin (the cull and
                                    // exact phases re-read the origin instead of keeping it), .x .y .z = the pending hit point
                                    double* sc = reinterpret_cast<double*>(&io.out[ray]);
                                    sc[0] = t_start;
                                    sc[1] = o.x;
                                    sc[2] = o.y;
                                    sc[6] = o.z;
                                }
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
                            L_tmin[slot] = kDblMax;
                            L_ray[slot] = ray;
                            L_xyz[slot] = (uint32_t)X | ((uint32_t)Y << 10) | ((uint32_t)Z << 20) | (d.x < 0 ? 1u << 30 : 0u) | (d.y < 0 ? 1u << 31 : 0u);
                            L_cell[slot] = cell;
                            L_pid[slot] = -1;
                            L_e1[slot] = e1; L_e2[slot] = e2; L_d1[slot] = -1; L_d2[slot] = -1;
                            L_flags[slot] = (d.z < 0 ? 1u : 0u) | (moved ? 2u : 0u);
                            unsigned q = 0, qe = 0;
                            int idx = -1, nexti = -1;
                            if (occupied(X, Y, Z, cell)) {
                                const CellRec c = g.cells[cell];
                                q = c.start; qe = c.start + c.count; idx = c.i0; nexti = c.i1;
                            }
                            L_q[slot] = q; L_qe[slot] = qe; L_idx[slot] = idx; L_nexti[slot] = nexti;
                            to_cull = q < qe;
                            to_walk = !to_cull;
                        }
                    }
                    freed = !alive;
                }
                push(Q_walk, hW, nW, to_walk, slot);
                push(Q_cull, hC, nC, to_cull, slot);
                push(Q_free, hF, nF, freed, slot);
            }
        }
        if (nW + nC + nE == 0) {
            if (drained) break;
            continue;
        }

        // ------------------------------------------------------------------ pick the phase for this round
        const bool do_exact = nE >= (unsigned)HARE_K1Q_EXACT_MIN || (nW + nC == 0);
        const bool do_cull = !do_exact && nC > 0 && (nC >= nW || nC >= 64u);
        if (do_exact) {
            // -------------------------------------------------------------- exact FP64 test of one candidate per ray
            bool act;
            const unsigned slot = pop(Q_exact, hE, nE, nE < 64u ? nE : 64u, act);
            bool to_walk = false, to_cull = false;
            if (act) {
                const unsigned ray = L_ray[slot];
                const int i = L_idx[slot];
                unsigned q = L_q[slot];
                const unsigned qe = L_qe[slot];
                const double tmin = L_tmin[slot];
                int after = -1;                                             // items[q + 2]: nexti once this candidate is done
                if (q + 2 < qe) after = g.items[q + 2];
                const RayRec r = io.rays[ray];
                V3 o = {r.x, r.y, r.z};
                if (L_flags[slot] & 2u) {                                   // origin was clipped to OBox (AABB_Main.cs:254-257)
                    const double* sc = reinterpret_cast<const double*>(&io.out[ray]);
                    o.x = sc[1]; o.y = sc[2]; o.z = sc[6];
                }
                const V3 d = {r.dx, r.dy, r.dz};
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
                const bool side = ray_side(d, nn);                          // Polygons.cs:641-648
                double a[3], c[3];
#pragma unroll
                for (int m = 0; m < 3; ++m) { a[m] = side ? v0[m] : v2[m]; c[m] = side ? v2[m] : v0[m]; }
                double t = 0;
                bool ok = tri_fast(o, d, a, v1, c, t);
                if (QUADS) {
                    const double v3[3] = {q3x, q3y, q3z};
                    if (!ok && qnv == 4) ok = tri_fast(o, d, c, v3, a, t);     // (P2,P3,P0) / (P0,P3,P2)
                }
                if (ok && t > kTMin && t < tmin) {                              // Voxel_Grid.cs:691-693
                    L_tmin[slot] = t;
                    L_pid[slot] = i;
                    XEventRec* e = &io.out[ray];
                    // X_Point of the pending hit (Polygons.cs:652); .t is left alone: it may hold t_start
                    e->x = o.x + d.x * t;
                    e->y = o.y + d.y * t;
                    e->z = o.z + d.z * t;
                }
                L_d2[slot] = L_d1[slot];
                L_d1[slot] = i;
                // next_candidate()
                ++q;
                if (q < qe) {
                    L_idx[slot] = L_nexti[slot];
                    if (q + 1 < qe) L_nexti[slot] = after;
                }
                L_q[slot] = q;
                to_cull = q < qe;
                to_walk = !to_cull;
            }
            push(Q_walk, hW, nW, to_walk, slot);
            push(Q_cull, hC, nC, to_cull, slot);
        } else if (do_cull) {
            // -------------------------------------------------------------- FP32 pre-cull, up to 2 x CULL_PAIRS candidates per ray
            bool act;
            const unsigned slot = pop(Q_cull, hC, nC, nC < 64u ? nC : 64u, act);
            bool to_walk = false, to_cull = false, to_exact = false;
            unsigned q = 0, qe = 0;
            int idx = -1, nexti = -1, e1 = -1, e2 = -1, done1 = -1, done2 = -1;
            double ox = 0, oy = 0, oz = 0;
            float dfx = 0, dfy = 0, dfz = 0, dm = 0;
            bool culling = act;
            if (act) {
                const unsigned ray = L_ray[slot];
                q = L_q[slot]; qe = L_qe[slot]; idx = L_idx[slot]; nexti = L_nexti[slot];
                e1 = L_e1[slot]; e2 = L_e2[slot]; done1 = L_d1[slot]; done2 = L_d2[slot];
                const RayRec r = io.rays[ray];
                ox = r.x; oy = r.y; oz = r.z;
                if (L_flags[slot] & 2u) {                                   // origin was clipped to OBox
                    const double* sc = reinterpret_cast<const double*>(&io.out[ray]);
                    ox = sc[1]; oy = sc[2]; oz = sc[6];
                }
                dfx = (float)r.dx; dfy = (float)r.dy; dfz = (float)r.dz;
                dm = fabsf(dfx) + fabsf(dfy) + fabsf(dfz);
            }
            struct CullRec { double2 c0; uint4 r1; float4 fb; float2 fc; };
            auto load_rec = [&](int i) {
                const unsigned char* rec = reinterpret_cast<const unsigned char*>(g.polys + i);
                CullRec r;
                r.c0 = *reinterpret_cast<const double2*>(rec);          // v0.x v0.y
                r.r1 = *reinterpret_cast<const uint4*>(rec + 16);       // v0.z | e1f.x e1f.y
                r.fb = *reinterpret_cast<const float4*>(rec + 32);      // e1f.z e2f.x e2f.y e2f.z
                r.fc = *reinterpret_cast<const float2*>(rec + 48);      // ee emax
                return r;
            };
            auto culled = [&](const CullRec& r) {
                const double c1x = __hiloint2double((int)r.r1.y, (int)r.r1.x);
                const float e1f[3] = {__uint_as_float(r.r1.z), __uint_as_float(r.r1.w), r.fb.x}, e2f[3] = {r.fb.y, r.fb.z, r.fb.w};
                return cull_fp32((float)(ox - r.c0.x), (float)(oy - r.c0.y), (float)(oz - c1x), dfx, dfy, dfz, dm, e1f, e2f, r.fc.x, r.fc.y);
            };
            // Re-testing a polygon can never change the result (strict `t < tmin`), so skipping the two this ray tested
            // last is exact (Voxel_Grid.cs:477 + the register mailbox of K1p)
            auto skip = [&](int i) { return i == e1 || i == e2 || i == done1 || i == done2; };
#pragma unroll 1
            for (int kp = 0; kp < HARE_K1Q_CULL_PAIRS; ++kp) {
                if (__ballot(culling) == 0) break;
                if (culling) {
                    // candidates idx (at q) and nexti (at q + 1); the one after those is requested now, used next iteration
                    const bool has1 = q + 1 < qe;
                    const bool has2 = q + 2 < qe;
                    const bool sk0 = skip(idx);
                    const bool sk1 = !has1 || skip(nexti) || nexti == idx;
                    int i2 = -1, i3 = -1;
                    if (has2) i2 = g.items[q + 2];
                    if (q + 3 < qe) i3 = g.items[q + 3];
                    CullRec ra, rb;
                    if (!sk0) ra = load_rec(idx);
                    if (!sk1) rb = load_rec(nexti);
                    bool parked = false;
                    if (!sk0) {
                        if (culled(ra)) { done2 = done1; done1 = idx; }
                        else parked = true;                                 // idx (at q) goes to the exact phase
                    }
                    if (!parked) {
                        ++q;                                                // candidate 0 consumed
                        if (has1) {
                            bool keep1 = false;
                            if (!sk1) {
                                if (culled(rb)) { done2 = done1; done1 = nexti; }
                                else keep1 = true;
                            }
                            if (keep1) {                                    // nexti (now at q) goes to the exact phase
                                idx = nexti; nexti = i2;
                                parked = true;
                            } else {
                                ++q;                                        // candidate 1 consumed
                                idx = i2; nexti = i3;
                            }
                        }
                    }
                    if (parked) { to_exact = true; culling = false; }
                    else if (q >= qe) { to_walk = true; culling = false; }
                }
            }
            if (act) {
                if (culling) to_cull = true;                                // quota used up, list not exhausted
                L_q[slot] = q; L_idx[slot] = idx; L_nexti[slot] = nexti;
                L_d1[slot] = done1; L_d2[slot] = done2;
            }
            push(Q_walk, hW, nW, to_walk, slot);
            push(Q_cull, hC, nC, to_cull, slot);
            push(Q_exact, hE, nE, to_exact, slot);
        } else {
            // -------------------------------------------------------------- DDA walk until the next non-empty voxel
            bool act;
            const unsigned slot = pop(Q_walk, hW, nW, nW < 64u ? nW : 64u, act);
            bool to_cull = false, freed = false;
            bool walking = act;
            double tMaxX = 0, tMaxY = 0, tMaxZ = 0, tDeltaX = 0, tDeltaY = 0, tDeltaZ = 0, tmin = 0;
            double hx = 0, hy = 0, hz = 0;
            int X = 0, Y = 0, Z = 0, cell = 0, pid = -1, dx1 = 1, dy1 = 1, dz1 = 1;
            unsigned ray = 0, fl = 0, q = 0, qe = 0;
            int idx = -1, nexti = -1;
            if (act) {
                tMaxX = L_tmx[slot]; tMaxY = L_tmy[slot]; tMaxZ = L_tmz[slot];
                tDeltaX = L_tdx[slot]; tDeltaY = L_tdy[slot]; tDeltaZ = L_tdz[slot];
                const uint32_t xyz = L_xyz[slot];
                X = (int)(xyz & 1023u); Y = (int)((xyz >> 10) & 1023u); Z = (int)((xyz >> 20) & 1023u);
                fl = L_flags[slot];
                dx1 = (xyz >> 30) & 1u ? -1 : 1; dy1 = (xyz >> 31) & 1u ? -1 : 1; dz1 = (fl & 1u) ? -1 : 1;
                cell = L_cell[slot];
                pid = L_pid[slot];
                ray = L_ray[slot];
                if (pid >= 0) {
                    tmin = L_tmin[slot];
                    const XEventRec* e = &io.out[ray];
                    hx = e->x; hy = e->y; hz = e->z;
                }
            }
            const int dcx = dx1 * ct * ct, dcy = dy1 * ct, dcz = dz1;
#pragma unroll 1
            for (int k = 0; k < HARE_K1Q_WALK_STEPS; ++k) {
                const unsigned long long wm = __ballot(walking);
                if (wm == 0 || (k > 0 && __popcll(wm) < HARE_K1Q_WALK_MIN)) break;
                if (walking) {
                    // Voxel_Grid.cs:705: pending hit inside the CURRENT padded voxel?
                    bool done = false;
                    if (pid >= 0) {
                        const double lox = voxel_lo(X, g.vd[0], g.omin[0]), hix = voxel_hi(X, g.vd[0], g.omin[0]);
                        const double loy = voxel_lo(Y, g.vd[1], g.omin[1]), hiy = voxel_hi(Y, g.vd[1], g.omin[1]);
                        const double loz = voxel_lo(Z, g.vd[2], g.omin[2]), hiz = voxel_hi(Z, g.vd[2], g.omin[2]);
                        if (!(hx < lox) && !(hy < loy) && !(hz < loz) && !(hx > hix) && !(hy > hiy) && !(hz > hiz)) {
                            store_hit(ray, (fl & 2u) != 0, tmin, pid);
                            nhits++;
                            done = true;
                            walking = false;
                            freed = true;
                        }
                    }
                    if (!done) {
                        // Voxel_Grid.cs:713-759 with selects (same booleans, same order; see K1p)
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
                        cell += sx ? dcx : (sy ? dcy : dcz);
                        const bool out = ((unsigned)X >= (unsigned)ct) | ((unsigned)Y >= (unsigned)ct) | ((unsigned)Z >= (unsigned)ct);
                        if (out) {                                          // leaving the grid: miss, even with a pending hit (F12)
                            store_miss(ray);
                            walking = false;
                            freed = true;
                        } else if (occupied(X, Y, Z, cell)) {
                            walking = false;                                // the cell record is fetched after the loop: a load
                            to_cull = true;                                 // in here would stall all 64 lanes at every step
                        }
                    }
                }
            }
            if (to_cull) {
                const CellRec c = g.cells[cell];
                q = c.start; qe = c.start + c.count; idx = c.i0; nexti = c.i1;
                if (COARSE && c.count == 0) { to_cull = false; walking = true; }   // the block is occupied, this voxel is not: walk on
            }
            if (act && !freed) {
                L_tmx[slot] = tMaxX; L_tmy[slot] = tMaxY; L_tmz[slot] = tMaxZ;
                L_xyz[slot] = (uint32_t)X | ((uint32_t)Y << 10) | ((uint32_t)Z << 20) | (dx1 < 0 ? 1u << 30 : 0u) | (dy1 < 0 ? 1u << 31 : 0u);
                L_cell[slot] = cell;
                if (to_cull) { L_q[slot] = q; L_qe[slot] = qe; L_idx[slot] = idx; L_nexti[slot] = nexti; }
            }
            push(Q_walk, hW, nW, walking, slot);                            // still walking: next round
            push(Q_cull, hC, nC, to_cull, slot);
            push(Q_free, hF, nF, freed, slot);
        }
    }

    // batch counters: per-wave partials, summed by hare_ctr_reduce
    if (io.ctr) {
        const unsigned long long r = wave_sum_u32(nrays), h = wave_sum_u32(nhits);
        if (lane == 0) {
            unsigned long long* sl = io.part + 2ull * (blockIdx.x * (unsigned)kPoolWaves + (unsigned)wave);
            sl[0] = r;
            sl[1] = h;
        }
    }
}

}  // namespace

extern "C" {
__global__ __launch_bounds__(64 * HARE_K1Q_WAVES) void hare_voxel_pool_tri(VoxelArgs g, ShootIO io) { voxel_pool_body<false, false>(g, io); }
__global__ __launch_bounds__(64 * HARE_K1Q_WAVES) void hare_voxel_pool_quad(VoxelArgs g, ShootIO io) { voxel_pool_body<true, false>(g, io); }
__global__ __launch_bounds__(64 * HARE_K1Q_WAVES) void hare_voxel_pool_tri_g(VoxelArgs g, ShootIO io) { voxel_pool_body<false, true>(g, io); }
__global__ __launch_bounds__(64 * HARE_K1Q_WAVES) void hare_voxel_pool_quad_g(VoxelArgs g, ShootIO io) { voxel_pool_body<true, true>(g, io); }
}
