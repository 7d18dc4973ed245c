// kernels.hip -- gfx950 (CDNA4, wave64) kernels for Hare's ray-cast path.
//
// Built with: hipcc --offload-arch=gfx950 --genco -O3 -ffp-contract=off
// (contraction OFF is a correctness requirement: the reference is .NET FP64, which never fuses
//  a*b+c; X_Event parity on near-ties depends on it -- SURVEY.md F5.  The one place that fuses, explicitly,
//  is the conservative FP32 pre-cull, which is a filter and not reference arithmetic.)
//
// Kernels (one ray per lane everywhere):
//   hare_voxel_persist_*   K1p  production Voxel_Grid.Shoot: persistent waves, per-lane state machine,
//                               LDS occupancy bitmap, FP32 pre-cull + exact FP64 test
//   hare_voxel_shoot_*     K1   the reference loop structure as is (work counters, A/B baseline)
//   hare_octree_persist    K2p  production Octree.Shoot; hare_octree_shoot* K2 simple/counting form
//   hare_kdtree_shoot*          KDTree.Shoot (visits every leaf, like the reference)
//   hare_reflect           K3   specular bounce between casts (harness-defined); hare_live_count / hare_scan_tiles /
//                               hare_reflect_compact / hare_events_*: the same with the survivors packed (hare_bounce_batch)
//   hare_cull_audit             tests only: FP32 cull vs exact test on every ray x polygon pair
//   hare_vb_*, hare_scan_*, hare_ob_*  Voxel_Grid / Octree construction (build_kernels.hip, included at the end)
//
// The arithmetic is a restatement of
//   Voxel_Grid.Shoot           Voxel_Grid.cs:561-761 (+ :351-552, the poly_origin overload)
//   AABB.Intersect/IsPointInBox AABB_Main.cs:173-260, :75-84
//   Triangle/Quadrilateral.Intersect + RayXtri  Hare_Geometry_Polygons.cs:385-510, :637-688, :731-823
//   Octree.Shoot "Octree - alt.cs":159-306, KDTree.Shoot KDTree.cs:204-361
// kept in hare_math.h.  There is no Poly_Ray_ID mailbox on the GPU: re-testing a polygon can never
// change the result because the accept is the strict `t < tmin`; the production kernels
// only remember the last few polygons a ray tested, in registers.
#include <hip/hip_runtime.h>
#include <type_traits>
#include "hare_device.h"
#include "hare_trace.h"
#include "voxel_walk.h"

using namespace hare;

// Lanes of `m` below the calling lane: v_mbcnt_lo + v_mbcnt_hi on the (usually scalar) mask -- two instructions and no register held, where
// `__popcll(m & lane_lt)` kept a 64-bit per-lane constant alive through the whole kernel (two VGPRs, spilled to scratch in K2d: round 6).
__device__ __forceinline__ unsigned rank_below(unsigned long long m)
{
    return __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
}

#include "voxel_coop.hip"      // coop_trace: one ray traced by a whole wave (the cooperative tail of K1p and K1q)
#include "octree_coop.hip"     // coop_octree: the same for Octree.Shoot (kernel K2t behind K2p)

namespace {


// Events are written exactly once per launch (59 MB per 1M rays) while the scene is re-read all the time: stream
// them past the caches (non-temporal) so that they do not evict it (+1.3 %; the same for the ray loads measured worse).
// (Seven 8-byte stores: a record is only 8-byte aligned.  Three 16-byte stores + one 8-byte store, chosen by the record's
//  alignment, were measured: the extra address arithmetic took K1p from 124 to 133 VGPRs, i.e. from 4 to 3 waves per SIMD,
//  and 0.50 -> 0.74 ms; the bytes saved never mattered at 6 % HBM utilisation.)
__device__ __forceinline__ void store_event_streaming(XEventRec* dst, const XEventRec& e)
{
    double* q = reinterpret_cast<double*>(dst);      // 56-byte records: 8-byte aligned only
    __builtin_nontemporal_store(e.t, q + 0);
    __builtin_nontemporal_store(e.u, q + 1);
    __builtin_nontemporal_store(e.v, q + 2);
    __builtin_nontemporal_store(e.x, q + 3);
    __builtin_nontemporal_store(e.y, q + 4);
    __builtin_nontemporal_store(e.z, q + 5);
    __builtin_nontemporal_store(__hiloint2double(e.hit, e.poly_id), q + 6);
}

__device__ __forceinline__ unsigned long long wave_sum_u32(unsigned int v)
{
    unsigned long long s = v;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    return s;  // valid in lane 0
}

__device__ __forceinline__ unsigned long long wave_sum_u64(unsigned long long s)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    return s;  // valid in lane 0
}

// End of a persistent launch, called by EVERY wave of the grid once it has no ray left (K1p, K1q, K2p, K2q).
// Replaces what used to surround each launch on the stream -- a memset of the ticket word in front, a reduce kernel over
// per-wave partials behind (2-3 % of a 0.49 ms launch) -- by work of the launch's own waves on its LaunchSlotMem (io.work):
//   1. the wave adds its {rays, hits} to one of 64 accumulator shards (RETURNING agent-scope atomics: when they have
//      returned, they have been performed at the memory side, whatever XCD the wave runs on);
//   2. it counts itself done in the counter of its blockIdx % 8 group; the group's last wave counts the group done; every
//      wave passes through here exactly once, so exactly one wave of the grid sees both counts complete;
//   3. that wave fetches-and-zeroes the shards (atomic exchanges: memory side again, so it sees every add of step 1, each of
//      which returned before its wave's done-count was issued), adds the sums to the caller's counters, and zeroes the done
//      counters and the ticket word: the slot is all zero again for the launch that uses it next (the host orders launches
//      on one slot, launch.cpp).  No fence is needed anywhere: every word of the slot is only ever touched by atomics.
// Cost: three dependent atomics per wave, off the ray path; ~4,000 done-counts spread over 8 + 1 addresses.
__device__ __forceinline__ void launch_epilogue(const ShootIO& io, unsigned int nrays, unsigned int nhits, unsigned int waves_per_block)
{
    const unsigned lane = threadIdx.x & 63u;
    unsigned int* const done = io.work + 1;                                                   // LaunchSlotMem::done
    unsigned long long* const acc = reinterpret_cast<unsigned long long*>(io.work + 16);    // LaunchSlotMem::acc
    const unsigned long long r = wave_sum_u32(nrays), h = wave_sum_u32(nhits);               // valid in lane 0
    int last = 0;
    if (lane == 0) {
        if (io.ctr && (r | h) != 0ull) {                  // (a wave that cast nothing has nothing to add: an empty cast of the bounce loop is all epilogue)
            const unsigned shard = (blockIdx.x * waves_per_block + (threadIdx.x >> 6)) & 63u;
            unsigned long long a = atomicAdd(&acc[2u * shard], r);
            unsigned long long b = atomicAdd(&acc[2u * shard + 1u], h);
            asm volatile("" ::"v"(a), "v"(b));          // the old values are "used": returning atomics, waited for ...
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // ... before the done-count below is issued
        }
        const unsigned xc = blockIdx.x & 7u;
        const unsigned in_group = ((gridDim.x - xc + 7u) >> 3) * waves_per_block;            // waves of the blocks b with b % 8 == xc
        if (atomicAdd(&done[xc], 1u) == in_group - 1u) {
            atomicExch(&done[xc], 0u);
            const unsigned groups = gridDim.x < 8u ? gridDim.x : 8u;
            if (atomicAdd(&done[8], 1u) == groups - 1u) {
                atomicExch(&done[8], 0u);
                last = 1;
            }
        }
    }
    last = __shfl(last, 0, 64);
    if (last) {
        unsigned long long vr = 0, vh = 0;
        if (io.ctr) {
            vr = atomicExch(&acc[2u * lane], 0ull);
            vh = atomicExch(&acc[2u * lane + 1u], 0ull);
        }
        vr = wave_sum_u64(vr);
        vh = wave_sum_u64(vh);
        if (lane == 0) {
            if (io.ctr) {
                atomicAdd(&io.ctr[CTR_RAYS], vr);
                atomicAdd(&io.ctr[CTR_HITS], vh);
            }
            atomicExch(io.work, 0u);                    // the ticket word: every wave has long stopped drawing
        }
    }
}

// HARE_SHOOT_COUNT_OWN: what a lane of a counting build (the *_own kernels) saw its rays do; summed over the wave into words 2 .. 5
__device__ __forceinline__ void flush_own(unsigned long long* ctr, const OwnWork& w)
{
    if (!ctr) return;
    const unsigned long long c = wave_sum_u32(w.cells), e = wave_sum_u32(w.entries), k = wave_sum_u32(w.culls), t = wave_sum_u32(w.tests);
    const unsigned long long st = wave_sum_u32(w.steps);
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(&ctr[CTR_CELLS], c);
        atomicAdd(&ctr[CTR_ENTRIES], e);
        atomicAdd(&ctr[CTR_TESTS], t);
        atomicAdd(&ctr[CTR_CULLS], k);
        if (st) atomicAdd(&ctr[CTR_STEPS], st);
    }
}

// Per-wave accumulation of the batch counters: one atomic per counter per wave.
__device__ __forceinline__ void flush_counters(unsigned long long* ctr, bool valid, bool hit, const Work& w, bool detailed)
{
    if (!ctr) return;
    const unsigned long long mv = __ballot(valid), mh = __ballot(valid && hit);
    unsigned long long c = 0, e = 0, t = 0;
    if (detailed) {
        c = wave_sum_u32(valid ? w.cells : 0u);
        e = wave_sum_u32(valid ? w.entries : 0u);
        t = wave_sum_u32(valid ? w.tests : 0u);
    }
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(&ctr[CTR_RAYS], (unsigned long long)__popcll(mv));
        atomicAdd(&ctr[CTR_HITS], (unsigned long long)__popcll(mh));
        if (detailed) {
            atomicAdd(&ctr[CTR_CELLS], c);
            atomicAdd(&ctr[CTR_ENTRIES], e);
            atomicAdd(&ctr[CTR_TESTS], t);
        }
    }
}


template <bool QUADS, bool COUNT>
__device__ __forceinline__ void voxel_shoot_body(const VoxelArgs& g, const ShootIO& io)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool valid = i < io.n;
    XEventRec ev;
    set_miss(ev);
    Work w = {0, 0, 0};
    bool live = false;
    if (valid) {
        const RayRec r = io.rays[i];
        V3 o = {r.x, r.y, r.z};
        const V3 d = {r.dx, r.dy, r.dz};
        const int e1 = io.excl1 ? io.excl1[i] : -1;
        const int e2 = io.excl2 ? io.excl2[i] : -1;
        live = !(e1 == -2 && (io.flags & SHOOT_RETIRED_RAYS));   // a ray the bounce loop retired (hare_reflect); else -2 is an ordinary "none"
        double tmin_local = 0;
        const bool moved = live && trace_voxel<QUADS, COUNT>(g, o, d, e1, e2, ev, w, &tmin_local);
        if (moved && ev.hit != 0 && (io.flags & SHOOT_SLIM_EVENTS)) ev.u = tmin_local;      // see SHOOT_SLIM_EVENTS
        if (io.out) io.out[i] = ev;
        if (io.occluded) {
            const bool occ = ev.hit != 0 && (io.tmax == nullptr || ev.t < io.tmax[i]);
            io.occluded[i] = occ ? 1 : 0;
            if (!io.out) ev.hit = occ ? 1 : 0;        // flags only: the batch counter `hits` counts occluded rays
        }
        if (moved && (io.flags & SHOOT_WRITEBACK_ORIGIN)) {
            io.rays[i].x = o.x;
            io.rays[i].y = o.y;
            io.rays[i].z = o.z;
        }
    }
    flush_counters(io.ctr, valid && live, ev.hit != 0, w, COUNT);
}



// The live blocks of a bounce cast, ascending, and their number, from hare_reflect's byte per block: ONE workgroup of NT lanes
// (hare_live_blocks), each lane a contiguous share of the bytes (a byte per 64 rays: 64 KB for four million rays) -- count, one
// workgroup-wide exclusive scan, write.  Two barriers whatever the batch size.  Deterministic -- no atomics: the waves meet the same rays in
// the same positions run after run.  (wsum: NT / 64 words of LDS.)
template <int NT>
__device__ __forceinline__ void build_live_list(const unsigned char* block_live, uint32_t nblk, uint32_t* list, uint32_t* count, uint32_t* wsum)
{
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const uint32_t per = ((nblk + (uint32_t)NT - 1u) / (uint32_t)NT + 15u) & ~15u;         // bytes per lane, whole 16-byte words
    const uint32_t b0 = threadIdx.x * per, b1 = b0 + per < nblk ? b0 + per : nblk;
    uint32_t c = 0;
    for (uint32_t b = b0; b < b1; b += 16u) {
        if (b + 16u <= b1) {
            const uint4 w = *reinterpret_cast<const uint4*>(block_live + b);       // bytes are 0 / 1: a popcount per word
            c += (uint32_t)__popc(w.x) + (uint32_t)__popc(w.y) + (uint32_t)__popc(w.z) + (uint32_t)__popc(w.w);
        } else {
            for (uint32_t k = b; k < b1; ++k) c += block_live[k] != 0 ? 1u : 0u;
        }
    }
    uint32_t inc = c;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t t = __shfl_up(inc, off, 64);
        if (lane >= off) inc += t;
    }
    if (lane == 63) wsum[wid] = inc;
    __syncthreads();
    uint32_t o = inc - c;
    for (int w = 0; w < wid; ++w) o += wsum[w];
    for (uint32_t b = b0; b < b1; b += 16u) {               // again sixteen bytes per load (a byte at a time: 64 dependent loads a lane, 30 us)
        if (b + 16u <= b1) {
            const uint4 w4 = *reinterpret_cast<const uint4*>(block_live + b);
            const uint32_t ws[4] = {w4.x, w4.y, w4.z, w4.w};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                uint32_t m = ws[q];
                while (m) {
                    const int bit = __ffs((int)m) - 1;      // bytes are 0 / 1: the set bit is bit 0 of its byte
                    list[o++] = b + 4u * (uint32_t)q + ((uint32_t)bit >> 3);
                    m &= m - 1u;
                }
            }
        } else {
            for (uint32_t k = b; k < b1; ++k)
                if (block_live[k] != 0) list[o++] = k;
        }
    }
    if (threadIdx.x == NT - 1) count[0] = o;          // the last lane's end = the total
    __syncthreads();
}

// ------------------------------------------------------------------------------------------------
// K1p: Voxel_Grid.Shoot as a persistent, wave-scheduled kernel.
//
// Why: in the straightforward kernel a wave executes, for EVERY cell step, the longest candidate list
// any of its 64 lanes holds -- SIMD utilisation of the polygon test was ~1.5 % on the 100k-tri hall --
// and every candidate costs a full FP64 Moller-Trumbore (~90 quarter-rate instructions) although
// ~17 of 18 candidates per ray are misses.  Here every lane is a small state machine and the wave
// alternates uniform phases:
//   A.  lanes whose candidate list is exhausted run up to `steps_per_round` DDA steps (pending-hit
//       check, step, occupancy test of the new cell);
//   B1. every lane that holds a candidate runs the conservative FP32 pre-cull (cull_fp32,
//       hare_math.h) on it; rejected candidates are dropped, survivors park the lane;
//   B2. when enough lanes are parked (or nothing else can progress) the parked lanes run the exact
//       FP64 test -- the reference's RayXtri, the only thing that decides a hit.
// Lanes that finish are refilled from a chunk of rays the wave drew with one atomic ticket
// (ballot + popcount compaction of the idle lanes), so waves stay full until the batch drains.
// The cell-occupancy bitmap (1 bit per voxel; per 2x2x2 .. 16x16x16 block above ~80^3) is staged in LDS once per workgroup, so the ~90 % of
// DDA steps that cross empty voxels never leave the CU.  Each polygon is one 128-byte line; the
// cull reads its first 64 bytes, the exact test the rest, both requested one phase before use.
//
// Per ray, the sequence of EXACT tests is the reference's candidate sequence with some certain
// misses removed; accepted hits, their order and their arithmetic are unchanged (Voxel_Grid.cs:561-761).
#ifndef HARE_K1P_STEPS
#define HARE_K1P_STEPS 10
#define HARE_K1P_CULLS 4
#endif
// OCC (hare_voxel_occl_*): the occlusion predicate instead of the X_Event.  The flag is "Shoot hits and the returned t is below
// t_max", so the walk may stop as soon as that is decided: a hit the reference would RETURN needs its point inside a voxel
// the walk reaches (Voxel_Grid.cs:705), and a polygon whose hit point lies in a voxel is in that voxel's list -- so every hit
// not found yet lies in the current voxel or beyond.  The current voxel was entered through the padded face of the previous
// one at parameter te; the next voxel's own padded box starts 2 mm before that face, i.e. at te - 0.002 / |d_axis|.  Once
// that is beyond t_max - t_start, and no hit below t_max is pending, nothing the rest of the walk finds can be below t_max:
// not occluded.  (Margins: 2.5 mm instead of 2, and t_max + 1e-9 relative, so that rounding can only make the walk longer;
// a pending hit is never acted on before the reference would confirm it, leaving the grid stays a miss: F12.)
#ifndef HARE_K1P_COOP_NOW
#define HARE_K1P_COOP_NOW 2        // tickets dry and this few lanes alive: the wave traces their rays cooperatively at once (voxel_coop.hip)
#define HARE_K1P_COOP_MAX 16       // ... or this few, once they have outlived the rest of the batch by
#define HARE_K1P_COOP_PATIENCE 24  // this many rounds (heavy rays)
#endif
#ifndef HARE_K1P_COOP
#define HARE_K1P_COOP 1            // 0: build K1p without the cooperative tail (A/B)
#endif
template <bool QUADS, bool COARSE, bool PROF = false, int STEPS = HARE_K1P_STEPS, int CULLS = HARE_K1P_CULLS, bool OCC = false>
__device__ __forceinline__ void voxel_persist_body(const VoxelArgs& g, const ShootIO& io)
{
    constexpr bool COOP = HARE_K1P_COOP && !OCC && !PROF;       // the occlusion and profiling builds keep every ray in its lane to the end
    // with origin write-back the ray record holds the MOVED origin, which coop_trace would move again
    const bool coop_on = io.coop_tail != 0 && !(io.flags & SHOOT_WRITEBACK_ORIGIN);
    int tail_rounds = 0;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    uint32_t* const locc = reinterpret_cast<uint32_t*>(lds_raw);   // address space known: ds_read, not flat
    // PROF: developer build with cycle stamps per phase (never the timed kernel).  Its statistics live in LDS behind the
    // bitmap (17 u64 per wave; the host adds the space), not in registers: the build must keep the production kernel's
    // register footprint, or it runs at a different occupancy and measures something else.
    unsigned long long* const pf = reinterpret_cast<unsigned long long*>(lds_raw + ((size_t)((g.occ_words + 3) >> 2) << 4)) + (threadIdx.x >> 6) * 18;
    unsigned long long* const pn = pf;              // slots 5..15: event / lane counts; slot 17: last stamp
    auto bump = [&](int slot, unsigned long long v) {
        if (PROF) { if ((threadIdx.x & 63) == 0) pf[slot] += v; }
    };
    auto stamp = [&](int slot) {
        if (PROF) {
            if ((threadIdx.x & 63) == 0) {
                const unsigned long long now = __builtin_readcyclecounter();
                pf[slot] += now - pf[17];
                pf[17] = now;
            }
        }
    };
    if (PROF) {
        if ((threadIdx.x & 63) == 0) {
            for (int k = 0; k < 17; ++k) pf[k] = 0;
            pf[17] = __builtin_readcyclecounter();
        }
    }
    (void)pn;
    {
        const int nw4 = (g.occ_words + 3) >> 2;           // the device buffer is padded to 16 bytes
        const uint4* src = reinterpret_cast<const uint4*>(g.occ);
        uint4* dst = reinterpret_cast<uint4*>(locc);
        for (int k = threadIdx.x; k < nw4; k += blockDim.x) dst[k] = src[k];
        __syncthreads();
    }

    const int ct = g.ct;
    const double fct = (double)ct;
    const int lane = threadIdx.x & 63;
    // developer timeline (flag 0x2000, tools/timeline_prof.py): per wave {start, last refill, end} on the
    // 100 MHz wall clock, stored straight to memory so that nothing stays live across the loop
    auto timeline = [&](int slot) {
        if (__builtin_expect((io.flags & 0x2000u) != 0 && io.prof != nullptr, 0)) {
            if (lane == 0) io.prof[32 + 4ull * (blockIdx.x * 4u + (threadIdx.x >> 6)) + slot] = __builtin_amdgcn_s_memrealtime();
        }
    };
    timeline(0);
    // scheduling knobs: compile-time in the production kernels (fewer live SGPRs), run-time in the
    // developer profiling build so that sweeps need no rebuild
    // tuned on MI355X at 1M and 16M rays (tools/ab_libs.py).  Steps and culls per round must grow TOGETHER: (8..12, 4) beat
    // (3, 1) by 6 % at 1M rays, while (8, 1), (3, 4), (8, 8) or (16, 8) all lose to it; the refill / exact thresholds are flat.
#ifndef HARE_K1P_REFILL
#define HARE_K1P_REFILL 16
#define HARE_K1P_EXACT 8
#endif
    const int STEPS_PER_ROUND = STEPS;
    const int REFILL_MIN_IDLE = HARE_K1P_REFILL;
#ifndef HARE_K1P_STATIC
#define HARE_K1P_STATIC 128
#endif
    const int RAY_CHUNK = io.static_rays > 0 ? io.static_rays : HARE_K1P_STATIC;
    const int EXACT_MIN_PARKED = HARE_K1P_EXACT;

    // wave-uniform work chunk [cn, ce).  The first chunk of every wave is static (wave w owns rays
    // [w*RAY_CHUNK, (w+1)*RAY_CHUNK)); tickets hand out the rays after those.  Same-address atomics
    // serialise chip-wide (~30 ns each), so a start-up draw by every wave would cost ~100 us of ramp.
    const unsigned int n32 = (unsigned int)io.n;   // 32-bit: the host routes n >= 2^31 to the simple kernel
    const unsigned int n_static = gridDim.x * 4u * (unsigned int)RAY_CHUNK;     // the host launches 4 waves per workgroup
    // XCD-aware static chunks: the dispatcher puts workgroup b on XCD b % 8 and every XCD has its own 4 MB L2, while the scene
    // is larger than that.  With chunk = wave index every XCD would cover the whole ray range at once; instead each XCD gets one
    // contiguous eighth of the static region (a contiguous slice of a ray front touches only part of the scene): +6 % at 1M rays,
    // +16 % for an all-static 512k batch.  The ticketed remainder needs nothing of the kind: one counter hands the rays out in
    // index order, so at any moment all XCDs work inside the same sliding window (per-XCD ticket ranges measured no better).
    unsigned int chunk_id = blockIdx.x * 4u + (threadIdx.x >> 6);
    if ((gridDim.x & 7u) == 0) chunk_id = ((blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3)) * 4u + (threadIdx.x >> 6);
    unsigned int cn = chunk_id * (unsigned int)RAY_CHUNK;
    unsigned int ce = cn + (unsigned int)RAY_CHUNK;
    if (cn > n32) cn = n32;
    if (ce > n32) ce = n32;
    bool drained = false;

    // ---- per-lane ray state
    bool alive = false;
    bool parked = false;             // holds a candidate that survived the cull, waiting for phase B2
    bool moved = false;              // origin was clipped to OBox: t_start is parked in out[ray].t
    unsigned int ray = 0;
    V3 o = {0, 0, 0}, d = {0, 0, 0};
    double tMaxX = 0, tMaxY = 0, tMaxZ = 0, tDeltaX = 0, tDeltaY = 0, tDeltaZ = 0;
    int X = 0, Y = 0, Z = 0, cell = 0;
    int dx1 = 1, dy1 = 1, dz1 = 1;   // stepX/Y/Z (Voxel_Grid.cs:589-632)
    int dcx = 0, dcy = 0, dcz = 0;   // the same steps as cell-index strides
    CullRay cray = {};               // the ray as the pre-cull reads it ((float)d, |d|_1; origin relative to the cull frame)
    int e1 = -1, e2 = -1;
    unsigned int q = 0, qe = 0;      // q: position of the CURRENT candidate `idx` in items
    int idx = -1, nexti = -1;        // current / following candidate polygon
    int done1 = -1, done2 = -1;      // the two polygons this ray tested last (register mailbox)
    double tmin = kDblMax;
    int pid = -1;
    unsigned int nhits = 0, nrays = 0;
    double occ_tstart = 0, occ_lim = kDblMax;      // OCC only: t_start of a clipped origin; t_max - t_start with its margin
    const double occ_px = 0.0025 / g.vd[0], occ_py = 0.0025 / g.vd[1], occ_pz = 0.0025 / g.vd[2];   // OCC only (wave-uniform)

    // Re-testing a polygon can never change the result (strict `t < tmin`), so skipping the ones
    // this ray has just tested is exact; it replaces the reference's Poly_Ray_ID mailbox
    // (Voxel_Grid.cs:687-689) for the common case of a polygon listed in consecutive voxels.
#ifndef HARE_K1P_MAILBOX
#define HARE_K1P_MAILBOX 0    // measured at C2: the compares on every candidate cost more than re-culling the 15 % repeats (0.502 -> 0.492 ms)
#endif
    auto skip = [&](int i) { return i == e1 || i == e2 || (HARE_K1P_MAILBOX >= 1 && i == done1) || (HARE_K1P_MAILBOX >= 2 && i == done2); };
    auto finish = [&](bool hit) {
        if (OCC) {       // the flag only: hit && (tmin + t_start) < t_max, the comparison hare_occlusion makes on the event's t
            const bool occ = hit && (io.tmax == nullptr || (tmin + occ_tstart) < io.tmax[ray]);
            io.occluded[ray] = occ ? 1 : 0;
            if (occ) nhits++;
            alive = false;
            return;
        }
        XEventRec ev;
        if (hit) {
            double t_start = 0;
            if (moved) t_start = io.out[ray].t;                  // parked there by the setup
            ev.t = tmin + t_start;                               // Voxel_Grid.cs:707
            ev.u = (moved && (io.flags & SHOOT_SLIM_EVENTS)) ? tmin : 0.0;      // 0 as the reference returns it, unless slim records are asked for
            ev.v = 0;
            ev.x = o.x + d.x * tmin;                             // Polygons.cs:652 (same operands => same bits)
            ev.y = o.y + d.y * tmin;
            ev.z = o.z + d.z * tmin;
            ev.poly_id = pid;
            ev.hit = 1;
            nhits++;
        } else {
            set_miss(ev);
        }
        store_event_streaming(&io.out[ray], ev);
        alive = false;
    };
    // the voxel just entered: occupancy bit from LDS; only a non-empty voxel touches memory
    auto enter_cell = [&]() {
        // COARSE (grids above ~80^3): the bit covers a block of voxels; the cell record says whether THIS one is empty
        const uint32_t bit = COARSE ? (uint32_t)(((X >> g.occ_shift) * g.occ_cd + (Y >> g.occ_shift)) * g.occ_cd + (Z >> g.occ_shift))
                                    : (uint32_t)cell;
        const uint32_t word = locc[bit >> 5];
        if ((word >> (bit & 31)) & 1u) {
            const CellRec c = g.cells[cell];
            q = c.start;
            qe = c.start + c.count;
            idx = c.i0;
            nexti = c.i1;
        }
    };
    auto next_candidate = [&]() {
        ++q;
        if (q < qe) {
            idx = nexti;
            if (q + 1 < qe) nexti = g.items[q + 1];
        }
    };

    stamp(0);
    for (;;) {
        bump(5, 1);
        // ------------------------------------------------------------------ refill idle lanes
        const unsigned long long idle = __ballot(!alive);
        if (__builtin_expect(!drained && (__popcll(idle) >= REFILL_MIN_IDLE || idle == ~0ull), 0)) {
            bool want = !alive;
            if (PROF) { bump(12, 1); bump(13, __popcll(idle)); }
            while (true) {
                const unsigned long long wm = __ballot(want);
                if (wm == 0) break;
                if (cn >= ce) {   // draw a new chunk: one atomic per RAY_CHUNK rays per wave
                    unsigned int base = 0;
                    const unsigned int dyn = (unsigned int)io.ticket_rays;     // run-time: the host sizes tickets to the batch
                    if (lane == 0) base = atomicAdd(io.work, dyn);
                    base = __shfl(base, 0, 64);
                    cn = base + n_static;
                    if (cn >= n32) { drained = true; break; }
                    ce = (n32 - cn > dyn) ? cn + dyn : n32;
                }
                const unsigned int rank = rank_below(wm);
                const unsigned int mine = cn + rank;
                const bool got = want && mine < ce;
                cn += (unsigned int)__popcll(__ballot(got));
                timeline(1);
                if (got) {
                    want = false;
                    ray = (unsigned int)mine;
                    // ---------------- per-ray setup: Voxel_Grid.cs:563-632
                    const RayRec r = io.rays[ray];
                    o.x = r.x; o.y = r.y; o.z = r.z;
                    d.x = r.dx; d.y = r.dy; d.z = r.dz;
                    e1 = io.excl1 ? io.excl1[ray] : -1;
                    e2 = io.excl2 ? io.excl2[ray] : -1;
                    tmin = kDblMax;
                    pid = -1;
                    done1 = -1; done2 = -1;
                    q = 0; qe = 0;
                    parked = false;
                    moved = false;
                    alive = true;
                    if (OCC) occ_tstart = 0;
                    if (e1 == -2 && (io.flags & SHOOT_RETIRED_RAYS)) {           // retired by the bounce loop: miss, not counted
                        finish(false);
                    } else {
                        nrays++;
                        double fx = floor((o.x - g.omin[0]) / g.vd[0]);
                        double fy = floor((o.y - g.omin[1]) / g.vd[1]);
                        double fz = floor((o.z - g.omin[2]) / g.vd[2]);
                        bool inside = (fx >= 0.0 && fx < fct) & (fy >= 0.0 && fy < fct) & (fz >= 0.0 && fz < fct);
                        if (!inside) {
                            double t_start = 0;
                            if (!aabb_clip_move(g.omin, g.omax, o, d, t_start)) {
                                finish(false);
                            } else {
                                moved = true;
                                if (OCC) occ_tstart = t_start;
                                else io.out[ray].t = t_start;     // read back by finish(); keeps 2 VGPRs free
                                if (io.flags & SHOOT_WRITEBACK_ORIGIN) {
                                    io.rays[ray].x = o.x; io.rays[ray].y = o.y; io.rays[ray].z = o.z;
                                }
                                fx = floor((o.x - g.omin[0] + d.x * 1E-6) / g.vd[0]);
                                fy = floor((o.y - g.omin[1] + d.y * 1E-6) / g.vd[1]);
                                fz = floor((o.z - g.omin[2] + d.z * 1E-6) / g.vd[2]);
                                inside = (fx >= 0.0 && fx < fct) & (fy >= 0.0 && fy < fct) & (fz >= 0.0 && fz < fct);
                                if (!inside) finish(false);
                            }
                        }
                        if (alive) {
                            X = (int)fx; Y = (int)fy; Z = (int)fz;
                            cell = (X * ct + Y) * ct + Z;
                            dx1 = d.x < 0 ? -1 : 1; dy1 = d.y < 0 ? -1 : 1; dz1 = d.z < 0 ? -1 : 1;
                            dcx = dx1 * ct * ct; dcy = dy1 * ct; dcz = dz1;
                            cray = cull_ray(g, o.x, o.y, o.z, d.x, d.y, d.z);      // after the clip: the origin the tests use
                            if (OCC) {
                                occ_lim = kDblMax;                           // no t_max: only the walk's own end decides
                                if (io.tmax) {
                                    const double tl = io.tmax[ray] - occ_tstart;
                                    occ_lim = tl + (fabs(tl) * 1e-9 + 1e-12);   // NaN t_max: every comparison below is false, the walk runs its course
                                }
                            }
                            if (d.x < 0) { tMaxX = (voxel_lo(X, g.vd[0], g.omin[0]) - o.x) / d.x; tDeltaX = g.vd[0] / d.x * -1.0; }
                            else         { tMaxX = (voxel_hi(X, g.vd[0], g.omin[0]) - o.x) / d.x; tDeltaX = g.vd[0] / d.x * 1.0; }
                            if (d.y < 0) { tMaxY = (voxel_lo(Y, g.vd[1], g.omin[1]) - o.y) / d.y; tDeltaY = g.vd[1] / d.y * -1.0; }
                            else         { tMaxY = (voxel_hi(Y, g.vd[1], g.omin[1]) - o.y) / d.y; tDeltaY = g.vd[1] / d.y * 1.0; }
                            if (d.z < 0) { tMaxZ = (voxel_lo(Z, g.vd[2], g.omin[2]) - o.z) / d.z; tDeltaZ = g.vd[2] / d.z * -1.0; }
                            else         { tMaxZ = (voxel_hi(Z, g.vd[2], g.omin[2]) - o.z) / d.z; tDeltaZ = g.vd[2] / d.z * 1.0; }
                            enter_cell();
                        }
                    }
                }
                // lanes that asked but found the chunk exhausted loop once more (new chunk)
            }
        }
        stamp(1);
        {
            const unsigned long long am = __ballot(alive);
            if (am == 0) {
                if (drained) break;
                continue;
            }
            // tickets dry and down to the last rays (voxel_coop.hip): one or two at once, a handful once they have outlived the
            // rest of the batch by COOP_PATIENCE rounds -- the whole wave then traces them one after the other
            if (COOP && drained && coop_on) {
                const int left = __popcll(am);
                if (left <= HARE_K1P_COOP_NOW || (left <= HARE_K1P_COOP_MAX && tail_rounds >= HARE_K1P_COOP_PATIENCE)) break;
                ++tail_rounds;
            }
        }
        if (PROF) bump(14, __popcll(__ballot(alive)));

        // ------------------------------------------------------------------ phase A: DDA steps
#pragma unroll 1
        for (int k = 0; k < STEPS_PER_ROUND; ++k) {
            const bool walk = alive && q == qe;
            if (__ballot(walk) == 0) break;
            if (PROF) { bump(6, 1); bump(7, __popcll(__ballot(walk))); }
            if (walk) {
                // Voxel_Grid.cs:705: pending hit inside the CURRENT padded voxel?
                bool done = false;
                if (pid >= 0) {
                    const double hx = o.x + d.x * tmin, hy = o.y + d.y * tmin, hz = o.z + d.z * tmin;
                    const double lox = voxel_lo(X, g.vd[0], g.omin[0]), hix = voxel_hi(X, g.vd[0], g.omin[0]);
                    const double loy = voxel_lo(Y, g.vd[1], g.omin[1]), hiy = voxel_hi(Y, g.vd[1], g.omin[1]);
                    const double loz = voxel_lo(Z, g.vd[2], g.omin[2]), hiz = voxel_hi(Z, g.vd[2], g.omin[2]);
                    if (!(hx < lox) && !(hy < loy) && !(hz < loz) && !(hx > hix) && !(hy > hiy) && !(hz > hiz)) {
                        finish(true);
                        done = true;
                    }
                }
                if (!done) {
                    // Voxel_Grid.cs:713-759, written with selects instead of the nested branches (same
                    // booleans, same order): X iff (tMaxX<tMaxY && tMaxX<tMaxZ); Y iff (!(tMaxX<tMaxY) &&
                    // tMaxY<tMaxZ); otherwise Z.  Step direction = sign of d (:589-632).
                    const bool cxy = tMaxX < tMaxY, cxz = tMaxX < tMaxZ, cyz = tMaxY < tMaxZ;
                    const bool sx = cxy & cxz;
                    const bool sy = (!cxy) & cyz;
                    const bool sz = !(sx | sy);
                    const double nX = tMaxX + tDeltaX, nY = tMaxY + tDeltaY, nZ = tMaxZ + tDeltaZ;
                    const double te = OCC ? (sx ? tMaxX : (sy ? tMaxY : tMaxZ)) : 0.0;     // parameter at which the next voxel is entered
                    X += sx ? dx1 : 0;
                    Y += sy ? dy1 : 0;
                    Z += sz ? dz1 : 0;
                    tMaxX = sx ? nX : tMaxX;
                    tMaxY = sy ? nY : tMaxY;
                    tMaxZ = sz ? nZ : tMaxZ;
                    cell += sx ? dcx : (sy ? dcy : dcz);
                    const bool out = ((unsigned)X >= (unsigned)ct) | ((unsigned)Y >= (unsigned)ct) | ((unsigned)Z >= (unsigned)ct);
                    bool beyond = false;
                    if (OCC) {
                        // entered through the stepped axis' padded face at te (the tMax just consumed); 0.0025 / |d_axis| = 0.0025 * tDelta / vd
                        const double pad = sx ? tDeltaX * occ_px : (sy ? tDeltaY * occ_py : tDeltaZ * occ_pz);
                        beyond = (pid < 0 || tmin > occ_lim) && (te - pad > occ_lim);
                    }
                    if (out | beyond) finish(false);     // leaving the grid: miss, even with a pending hit (F12); beyond t_max: not occluded
                    else enter_cell();
                }
            }
        }

        stamp(2);
        // ------------------------------------------------------------------ phase B1: FP32 cull
        if (PROF) { const unsigned long long m = __ballot(alive && !parked && q < qe); if (m) { bump(8, 1); bump(9, __popcll(m)); } }
#pragma unroll 1
        for (int kc = 0; kc < CULLS; ++kc) {
          if (CULLS > 1 && kc > 0 && __ballot(alive && !parked && q < qe) == 0) break;
          if (alive && !parked && q < qe) {
            if (skip(idx)) {                                                // Voxel_Grid.cs:477 (+ mailbox)
                next_candidate();
            } else {
                // the candidate's pre-cull record (hare_device.h: two or three 16-byte gathers)
                const CullRaw cr = cull_load<QUADS ? 1 : 0>(g, idx);
                if (cull_test<QUADS ? 1 : 0>(g, cray, cr)) {
                    done2 = done1;       // a certain miss counts as tested
                    done1 = idx;
                    next_candidate();
                } else {
                    parked = true;
                }
            }
          }
        }

        stamp(3);
        // ------------------------------------------------------------------ phase B2: exact FP64 test
        {
            const unsigned long long pm = __ballot(alive && parked);
            const unsigned long long busy = __ballot(alive && !parked);   // lanes that can still walk or cull
            if (pm != 0 && (__popcll(pm) >= EXACT_MIN_PARKED || busy == 0)) {
                if (PROF) { bump(10, 1); bump(11, __popcll(pm)); }
                if (alive && parked) {
                    const int i = idx;
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
                    // Ray_Side picks the corner order (Polygons.cs:641-648): (P0,P1,P2) or (P2,P1,P0)
                    const bool side = ray_side(d, nn);
                    double a[3], c[3];
#pragma unroll
                    for (int m = 0; m < 3; ++m) { a[m] = side ? v0[m] : v2[m]; c[m] = side ? v2[m] : v0[m]; }
                    double t = 0;
                    bool ok = tri_fast(o, d, a, v1, c, t);
                    if (QUADS) {
                        const double v3[3] = {q3x, q3y, q3z};
                        if (!ok && qnv == 4) ok = tri_fast(o, d, c, v3, a, t);     // (P2,P3,P0) / (P0,P3,P2)
                    }
                    if (ok && t > kTMin && t < tmin) {                      // :691-693
                        tmin = t;
                        pid = i;
                    }
                    done2 = done1;
                    done1 = i;
                    parked = false;
                    next_candidate();
                }
            }
        }
        stamp(4);
    }
    if (COOP) {
        // ---- the cooperative tail: each ray still alive is traced by the whole wave from where its lane left it.  The lane's walk
        // state goes through the ray's own event slot (scratch until the ray finishes, as in K1q): nothing of it stays in registers.
        if (alive) {
            double* sc = reinterpret_cast<double*>(&io.out[ray]);            // sc[0] holds t_start of a moved ray (the set-up put it there)
            sc[1] = tMaxX; sc[2] = tMaxY; sc[3] = tMaxZ;
            sc[4] = tmin;
            const uint32_t xf = (uint32_t)X | ((uint32_t)Y << 9) | ((uint32_t)Z << 18) | (dx1 < 0 ? kTF_NX : 0u) | (dy1 < 0 ? kTF_NY : 0u) |
                                (dz1 < 0 ? kTF_NZ : 0u) | (moved ? kTF_MOVED : 0u);
            sc[5] = __hiloint2double(pid, (int)xf);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        unsigned long long left = __ballot(alive);
        while (left) {
            const int l = (int)__builtin_ctzll(left);
            left &= left - 1ull;
            const unsigned cray_ray = (unsigned)__builtin_amdgcn_readlane((int)ray, l);
            double* sc = reinterpret_cast<double*>(&io.out[cray_ray]);
            const double w5 = sc[5];
            const uint32_t xf = (uint32_t)__double2loint(w5);
            double ctmin = sc[4];
            int cpid = __double2hiint(w5);
            const bool hit = coop_trace<QUADS, COARSE>(g, io, locc, cray_ray, xf, sc[1], sc[2], sc[3], ctmin, cpid);
            if (lane == l) {           // the ray's own lane finishes it (o, d, moved are its registers; t_start is read back from sc[0])
                tmin = ctmin;
                pid = cpid;
                finish(hit);
            }
        }
    }
    if (PROF) {
        if (lane == 0 && io.prof) {
#pragma unroll
            for (int k = 0; k < 16; ++k) atomicAdd(&io.prof[k], pf[k]);
            atomicAdd(&io.prof[16], 1ull);

        }
    }

    timeline(2);
    launch_epilogue(io, nrays, nhits, 4u);     // batch counters + the launch slot left zeroed (4 waves per workgroup)
}

// Filter audit (tests only): for every candidate the reference algorithm would test, run BOTH the FP32
// cull and the exact test and count (culled && exact test accepts) -- must be zero.  ctr[5] += violations,
// ctr[6] += culled candidates, ctr[7] += candidates.
__device__ __forceinline__ void audit_body(const VoxelArgs& g, const ShootIO& io)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned int viol = 0, culled = 0, cands = 0;
    if (i < io.n) {
        const RayRec r = io.rays[i];
        const V3 o = {r.x, r.y, r.z};
        const V3 d = {r.dx, r.dy, r.dz};
        const CullRay cray = cull_ray(g, o.x, o.y, o.z, d.x, d.y, d.z);
        // brute force over every polygon of the topology (n_polys in io.pad): the audit is about the
        // filter, not the traversal -- through the very load / decode / test the production kernels use
        const int P = io.audit_polys;
        for (int k = 0; k < P; ++k) {
            const PolyRec& p = g.polys[k];
            const bool c = cull_test(g, cray, cull_load(g, k));
            double t;
            const double* v3 = (g.quads && g.quads[k].nverts == 4) ? g.quads[k].v3 : nullptr;
            const bool hit = poly_fast(p, v3, o, d, t);
            cands++;
            if (c) {
                culled++;
                if (hit) viol++;
            }
        }
    }
    if (io.ctr) {
        const unsigned long long v = wave_sum_u32(viol), c = wave_sum_u32(culled), n = wave_sum_u32(cands);
        if ((threadIdx.x & 63) == 0) {
            atomicAdd(&io.ctr[5], v);
            atomicAdd(&io.ctr[6], c);
            atomicAdd(&io.ctr[7], n);
        }
    }
}


template <bool COUNT>
__device__ __forceinline__ void octree_shoot_body(const OctreeArgs& g, const ShootIO& io)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int nt = blockDim.x;
    const int levels = g.max_depth > 0 ? g.max_depth : 1;
    OctFrames fr;
    fr.a = reinterpret_cast<double*>(lds);
    fr.b = fr.a + (size_t)levels * nt;
    fr.first = reinterpret_cast<int*>(fr.b + (size_t)levels * nt);
    fr.cursor = fr.first + (size_t)levels * nt;

    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool valid = i < io.n;
    XEventRec ev;
    set_miss(ev);
    Work w = {0, 0, 0};
    bool live = false;
    if (valid) {
        const RayRec r = io.rays[i];
        const V3 o = {r.x, r.y, r.z};
        const V3 d = {r.dx, r.dy, r.dz};
        const int e1 = io.excl1 ? io.excl1[i] : -1;
        const int e2 = io.excl2 ? io.excl2[i] : -1;
        live = !(e1 == -2 && (io.flags & SHOOT_RETIRED_RAYS));
        if (live) trace_octree<COUNT, !COUNT>(g, fr, threadIdx.x, blockDim.x, o, d, e1, e2, ev, w);
        if (io.out) io.out[i] = ev;
        if (io.occluded) {
            const bool occ = ev.hit != 0 && (io.tmax == nullptr || ev.t < io.tmax[i]);
            io.occluded[i] = occ ? 1 : 0;
            if (!io.out) ev.hit = occ ? 1 : 0;        // flags only: the batch counter `hits` counts occluded rays
        }
    }
    flush_counters(io.ctr, valid && live, ev.hit != 0, w, COUNT);
}

// ------------------------------------------------------------------------------------------------
// K2p: Octree.Shoot as a persistent, wave-scheduled kernel (same skeleton as K1p).
//
// Each lane is a state machine over the depth-first walk described above (one LDS frame per level):
//   P. a lane without a leaf examines ONE child of its top frame per step (push-time test of
//      "Octree - alt.cs":268, pop-time tests of :207-211), up to STEPS children per round; an accepted
//      interior child opens a frame, an accepted leaf hands the lane its candidate list;
//   B1. a lane in a leaf runs the conservative FP32 pre-cull on one candidate;
//   B2. parked survivors run the exact FP64 RayXtri with u,v (the reference's full intersect), batched.
//
// Child boxes are not loaded.  BuildOctree derives them from the parent box ("Octree - alt.cs":96-111:
// min = (low ? node.Min : center) - 0.1, max = (low ? center : node.Max) + 0.1, center = (Max+Min)/2),
// so per axis the eight children share four planes.  When a frame opens, the lane computes those four
// planes from the node's own stored box with the same expressions (bit-identical to the stored child
// boxes -- checked on the host in tests/test_host_builders.py) and their twelve ray parameters once;
// a child test is then six selects and the max/min of :265-266.  Only accepted children fetch their
// 64-byte record.  Math.Max/Min are replaced by plain compare-selects that agree with them except for
// the sign of a zero result, which is only ever compared, never returned.
__device__ __forceinline__ double omax(double a, double b) { return (b < a || a != a) ? a : b; }   // NaN-propagating like Math.Max
__device__ __forceinline__ double omin(double a, double b) { return (a < b || a != a) ? a : b; }

// OCC (hare_octree_occl): the occlusion predicate instead of the X_Event.  Octree.Shoot has no pending-hit rule: the first hit
// found already makes Hit true and the returned t can only get smaller, so the walk stops at the first hit below t_max (the
// classic any-hit early out, exact here).  What is NOT exact, and therefore not done: skipping nodes whose entry lies beyond
// t_max.  The reference pops the FAR children first and returns early as soon as a hit lies in front of the current leaf's
// entry ("Octree - alt.cs":233), so a far leaf can end the query with a hit beyond t_max (not occluded) that a
// walk without that leaf would replace by a nearer one (occluded): the flag would differ from the reference's closest hit.
// Inclusive scans over the 64 lanes of a wave on the DPP network (row shifts inside rows of 16, then the two row broadcasts of gfx9):
// six VALU instructions, no LDS -- __shfl_up is a ds_bpermute per step.  `ident` is what a lane without a source keeps.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int wave_dpp(int ident, int v) { return __builtin_amdgcn_update_dpp(ident, v, CTRL, ROW_MASK, 0xF, false); }
__device__ __forceinline__ int wave_scan_add(int v)
{
    v += wave_dpp<0x111, 0xF>(0, v);      // row_shr:1
    v += wave_dpp<0x112, 0xF>(0, v);      // row_shr:2
    v += wave_dpp<0x114, 0xF>(0, v);      // row_shr:4
    v += wave_dpp<0x118, 0xF>(0, v);      // row_shr:8
    v += wave_dpp<0x142, 0xA>(0, v);      // row_bcast:15 into rows 1 and 3
    v += wave_dpp<0x143, 0xC>(0, v);      // row_bcast:31 into rows 2 and 3
    return v;
}
__device__ __forceinline__ int wave_scan_max(int v)   // values >= -1
{
    auto mx = [](int a, int b) { return a > b ? a : b; };
    v = mx(v, wave_dpp<0x111, 0xF>(-1, v));
    v = mx(v, wave_dpp<0x112, 0xF>(-1, v));
    v = mx(v, wave_dpp<0x114, 0xF>(-1, v));
    v = mx(v, wave_dpp<0x118, 0xF>(-1, v));
    v = mx(v, wave_dpp<0x142, 0xA>(-1, v));
    v = mx(v, wave_dpp<0x143, 0xC>(-1, v));
    return v;
}

// DENSE (hare_octree_dense, round 4): the same walk -- one lane owns a ray, frames in LDS, phase P as it is -- with the two things
// K2g taught about Octree.Shoot applied to it:
//   * a leaf's entries are not scanned by the owning lane, a pair per iteration at a third of the wave's lanes: ALL entries of all
//     leaves the wave's lanes hold are spread densely over the 64 lanes (exclusive scan of the counts; an item finds its owner by a
//     max-scan over segment starts in LDS and fetches the owner's pre-cull operands by ds_bpermute);
//   * a survivor of the pre-cull does not park its lane: it is NOTED (polygon, the leaf's nodeTmin, a bit on the first one noted from
//     its leaf; two per lane, in LDS: three were measured, 483 against 494 Mrays/s) and the lane walks on with the closestT it has (stale = prunes less, never more); the exact tests run when
//     enough lanes hold one, each lane on its own survivors in order, with the reference's rules replayed: entering a new leaf,
//     skipped = hit && closestT <= leafTmin (:210); not skipped and t < closestT: accept (:225); accepted t <= leafTmin: return (:233).
// Knobs (hare_device.h): HARE_K2D_PEND survivors a lane may hold before it has to wait for the exact phase; HARE_K2D_CAP list entries of one
// leaf that go into one round's dense passes; HARE_K2D_EXACT_MIN lanes holding a survivor that make the exact phase run (or one that
// cannot go on); HARE_K2D_STEPS pop steps per round.
// OWN (hare_octree_dense_own; flag HARE_SHOOT_COUNT_OWN): the same kernel counting ITS OWN work -- node records fetched (pops and the
// root), list entries put through the dense pre-cull, exact tests -- into words 2 .. 5 of the counters block (bench.py: `roofline.own`).
template <bool OCC, bool DENSE = false, bool OWN = false>
__device__ __forceinline__ void octree_persist_body(const OctreeArgs& g, const ShootIO& io)
{
    OwnWork ownw;
    int tail_rounds = 0;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    constexpr int nt = 256;          // the launch's block size (launch.cpp: launch_persist): frame indices are shifts, not 32-bit multiplies (quarter rate)
    const int tid = threadIdx.x;
    const int levels = g.max_depth > 0 ? g.max_depth : 1;
    double* const fa = reinterpret_cast<double*>(lds);              // [levels][nt] interval start of the frame's node
    double* const fb = fa + (size_t)levels * nt;                    // [levels][nt] interval end
    int* const fpk = reinterpret_cast<int*>(fb + (size_t)levels * nt);   // [levels][nt] first_child << 8 | children still to pop (by cursor position)
    // DENSE: behind the frames, the lanes' pending survivors and a 64-word table per wave (segment starts of the dense cull)
    constexpr int P = HARE_K2D_PEND;
    double* const pend_lca = reinterpret_cast<double*>(fpk + (size_t)levels * nt);     // [P][nt]
    int* const pend_w = reinterpret_cast<int*>(pend_lca + (size_t)P * nt);             // [P][nt] polygon | bit 31: the first survivor noted from its leaf
    int* const seg_mark = pend_w + (size_t)P * nt + (tid >> 6) * 64;

    const int lane = tid & 63;
    // developer timeline (flag 0x2000, tools/timeline_oct.py): per wave {start, tickets dry, end} on the 100 MHz wall clock
    auto timeline = [&](int slot) {
        if (__builtin_expect((io.flags & 0x2000u) != 0 && io.prof != nullptr, 0)) {
            if (lane == 0) io.prof[32 + 4ull * (blockIdx.x * 4u + (threadIdx.x >> 6)) + slot] = __builtin_amdgcn_s_memrealtime();
        }
    };
    timeline(0);
    // tuned on C3 (tools/sweep_oct.py): parking survivors does not pay here, several culls per round do
#ifndef HARE_K2P_STEPS
#define HARE_K2P_STEPS 4
#define HARE_K2P_CULLS 12
#define HARE_K2P_REFILL 8
#define HARE_K2P_EXACT 1
#endif
    // (DENSE: the refill threshold comes from the host -- launch.cpp: 16 idle lanes on long batches, 32 on short ones, where a wave's few
    //  refills are better made of whole tickets)
    const int STEPS = DENSE ? HARE_K2D_STEPS : HARE_K2P_STEPS, CULLS = HARE_K2P_CULLS,
              REFILL_MIN_IDLE = DENSE ? ((io.refill_min_idle > 0 && io.refill_min_idle <= 64) ? io.refill_min_idle : HARE_K2D_REFILL) : HARE_K2P_REFILL,
              EXACT_MIN_PARKED = HARE_K2P_EXACT;
    const int RAY_CHUNK = io.static_rays > 0 ? io.static_rays : 128;      // the host sizes it by the batch (ShootIO::static_rays)
    const unsigned int n32 = (unsigned int)io.n;
    // static first chunk per wave, tickets of io.ticket_rays after those (as in the voxel kernel)
    const unsigned int n_static = gridDim.x * 4u * (unsigned int)RAY_CHUNK;
    // XCD-contiguous static chunks, as in the voxel kernel
    unsigned int chunk_id = blockIdx.x * 4u + (threadIdx.x >> 6);
    if ((gridDim.x & 7u) == 0) chunk_id = ((blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3)) * 4u + (threadIdx.x >> 6);
    unsigned int cn = chunk_id * (unsigned int)RAY_CHUNK;
    unsigned int ce = cn + (unsigned int)RAY_CHUNK;
    if (cn > n32) cn = n32;
    if (ce > n32) ce = n32;
    bool drained = false;
    unsigned int dead_run = 0;          // SHOOT_RETIRED_RAYS: tickets in a row whose rays were all retired
    bool chunk_live = true;

    bool alive = false, parked = false, hit = false, tame = true;
    bool tight_ok = false;              // this ray may be tested against the subtrees' tight boxes (g.tight: a tame ray with its origin near the scene)
    unsigned int ray = 0;
    V3 o = {0, 0, 0}, d = {0, 0, 0};
    double invDx = 0, invDy = 0, invDz = 0;
    CullRay cray = {};
    int mask = 0, lvl = -1;
    int e1 = -1, e2 = -1;
    int q = 0, qe = 0;                  // remaining candidates of the current leaf: items[q .. qe)
    int idx = -1, nexti = -1;           // items[q], items[q + 1], already here: a cull iteration waits for ONE round of loads, not two
    double leaf_ca = 0;                 // nodeTmin of the current leaf
    double closestT = kDblMax, bu = 0, bv = 0;
    int pid = -1;
#ifndef HARE_K2P_MAILBOX
#define HARE_K2P_MAILBOX 0    // measured: 96 of the 100 list entries a ray scans are distinct polygons; the compares cost more than the 4 repeats
#endif
    int m0 = -1, m1 = -1, m2 = -1, m3 = -1;       // the polygons tested last (HARE_K2P_MAILBOX of them)
    unsigned int nhits = 0, nrays = 0;
    int np = 0;                         // DENSE: survivors noted and not yet tested
    bool leaf_fresh = false;            // DENSE: no survivor has been noted from the leaf in hand yet
    bool cur_skip = false;              // DENSE, replay: the reference skipped the leaf of the survivors being replayed (:210)
    bool leaving = false;               // DENSE: the wave hands its rays over after one more exact phase
#ifdef HARE_K2P_STATS                   // developer build (tools/k2p_stats.py): what a round of the loop is made of; lane 0 counts
    unsigned long long dr_t[4] = {0, 0, 0, 0}, dr_last = 0;   // ticks (100 MHz) in the pop steps / dense windows / exact phase / rest of a round, rounds with <= 8 rays after dry
    bool dr_few = false;
    unsigned dr_round = 0, dr_p = 0, dr_c = 0, dr_e = 0;   // the same per wave, after its tickets ran dry (timeline slot 3)
    unsigned long long sp_round = 0, sp_alive = 0, sp_p = 0, sp_pl = 0, sp_c = 0, sp_cl = 0, sp_e = 0, sp_el = 0, sp_visit = 0;
    unsigned long long sp_is = 0, sp_il = 0, sp_ls = 0, sp_ll = 0, sp_xl = 0, sp_fl = 0;   // steps with an interior visit / their lanes; leaf visits; exhausted frames; failed pops
    int st_kind = 0;                    // what this lane's last pop step did: 1 frame exhausted, 2 popped and dropped, 3 leaf, 4 interior
#define K2P_STAT(x) x
#else
#define K2P_STAT(x)
#endif

    auto finish = [&]() {
        if (OCC) {
            const bool occ = hit && (io.tmax == nullptr || closestT < io.tmax[ray]);
            io.occluded[ray] = occ ? 1 : 0;
            if (occ) nhits++;
            alive = false;
            return;
        }
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
        alive = false;
    };
    // A node that passed the pop-time tests with interval [ca, cb].  A leaf hands the lane its list.  An interior node opens
    // a frame: the PUSH test of :268 -- it depends on the ray, the child box and this interval only, never on the hit so far
    // -- is made for all eight children at once, on planes derived from the node's own box with BuildOctree's expressions
    // (:96-111; bit-identical to the stored child boxes), and kept as a mask over cursor positions.  Children are then
    // popped from order[7] down to order[0] (:286-306); each popped child's interval comes from ITS OWN record, which the
    // visit needs anyway -- so nothing of the parent has to be re-read or recomputed when the walk comes back to a frame.
    auto visit = [&](const OctNode& nd, double ca, double cb, auto fast_tag) {
        constexpr bool FAST = decltype(fast_tag)::value;
        auto mx = [](double a, double b) { return FAST ? __builtin_fmax(a, b) : omax(a, b); };
        auto mn = [](double a, double b) { return FAST ? __builtin_fmin(a, b) : omin(a, b); };
        if (nd.first_child < 0) {
            q = nd.item_start; qe = nd.item_start + nd.item_count; leaf_ca = ca;
            idx = nd.pad;                                  // a leaf's first two list entries travel in its node record (OctNode, hare_device.h)
            nexti = -2 - nd.first_child;
            if (DENSE) leaf_fresh = true;
        } else if (FAST) {
            // A tame ray (finite, far from overflow, and the sign of 1/d is the sign of d on every axis -- `tame`, set-up): no NaN can
            // arise, so Math.Max / Math.Min are v_max_f64 / v_min_f64 and the four conditions of :268 fold.  The slabs are formed in
            // CURSOR order (cursor index a along an axis is the octant slab a ^ (d < 0)), so the eight tests yield the frame's mask as it
            // is stored -- no octant -> cursor permutation -- and the children that are empty leaves come as a cursor-ordered byte of
            // the device copy (one per direction mask, device_scene.cpp).  Same expressions on the same operands as below: same bits.
            double ncx[2], fcx[2], ncy[2], fcy[2], ncz[2], fcz[2];
            {
                const double c = (nd.bmax[0] + nd.bmin[0]) / 2;
                const double a0 = ((nd.bmin[0] - 0.1) - o.x) * invDx, a1 = ((c + 0.1) - o.x) * invDx;
                const double b0 = ((c - 0.1) - o.x) * invDx, b1 = ((nd.bmax[0] + 0.1) - o.x) * invDx;
                const bool m = (mask & 4) != 0;
                ncx[0] = m ? b1 : a0; fcx[0] = m ? b0 : a1; ncx[1] = m ? a1 : b0; fcx[1] = m ? a0 : b1;
            }
            {
                const double c = (nd.bmax[1] + nd.bmin[1]) / 2;
                const double a0 = ((nd.bmin[1] - 0.1) - o.y) * invDy, a1 = ((c + 0.1) - o.y) * invDy;
                const double b0 = ((c - 0.1) - o.y) * invDy, b1 = ((nd.bmax[1] + 0.1) - o.y) * invDy;
                const bool m = (mask & 2) != 0;
                ncy[0] = m ? b1 : a0; fcy[0] = m ? b0 : a1; ncy[1] = m ? a1 : b0; fcy[1] = m ? a0 : b1;
            }
            {
                const double c = (nd.bmax[2] + nd.bmin[2]) / 2;
                const double a0 = ((nd.bmin[2] - 0.1) - o.z) * invDz, a1 = ((c + 0.1) - o.z) * invDz;
                const double b0 = ((c - 0.1) - o.z) * invDz, b1 = ((nd.bmax[2] + 0.1) - o.z) * invDz;
                const bool m = (mask & 1) != 0;
                ncz[0] = m ? b1 : a0; fcz[0] = m ? b0 : a1; ncz[1] = m ? a1 : b0; fcz[1] = m ? a0 : b1;
            }
            double nxy[2][2], fxy[2][2];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) { nxy[i][j] = __builtin_fmax(ncx[i], ncy[j]); fxy[i][j] = __builtin_fmin(fcx[i], fcy[j]); }
            // :268 is !(tmx < tmn || tmx < 0 || tmn > cb || tmx < ca): without NaN, tmx >= max(tmn, 0, ca) && tmn <= cb.  The prune of
            // :210 at the child's pop, hit && closestT <= max(tmn, ca): an accepted t is > 1e-10, so max(tmn, ca, 0) decides the same.
            const double m0 = __builtin_fmax(ca, 0.0);
            unsigned byc = 0;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const double tmn = __builtin_fmax(nxy[(k >> 2) & 1][(k >> 1) & 1], ncz[k & 1]);
                const double tmx = __builtin_fmin(fxy[(k >> 2) & 1][(k >> 1) & 1], fcz[k & 1]);
                const double e = __builtin_fmax(tmn, m0);
                const bool p = (tmx >= e) & (tmn <= cb) & !(hit & (closestT <= e));
                byc |= p ? (1u << k) : 0u;
            }
            // children that are empty leaves, in cursor order for this ray's direction mask (popping one has no effect, :213)
            const unsigned long long empties = ((unsigned long long)(unsigned)nd.item_count << 32) | (unsigned)nd.item_start;
            byc &= ~(unsigned)(empties >> (mask * 8)) & 255u;
            if (byc != 0) {                                        // a node that pushes nothing opens no frame: every open frame has a child to pop
                ++lvl;
                fa[lvl * nt + tid] = ca;
                fb[lvl * nt + tid] = cb;
                fpk[lvl * nt + tid] = (int)(((unsigned)nd.first_child << 8) | byc);
            }
        } else {
            double nx[2], fx[2], ny[2], fy[2], nz[2], fz[2];     // entry / exit parameter of the low (0) and high (1) child slab
            {
                const double c = (nd.bmax[0] + nd.bmin[0]) / 2;
                const double a0 = ((nd.bmin[0] - 0.1) - o.x) * invDx, a1 = ((c + 0.1) - o.x) * invDx;
                const double b0 = ((c - 0.1) - o.x) * invDx, b1 = ((nd.bmax[0] + 0.1) - o.x) * invDx;
                const bool neg = invDx < 0;
                nx[0] = neg ? a1 : a0; fx[0] = neg ? a0 : a1; nx[1] = neg ? b1 : b0; fx[1] = neg ? b0 : b1;
            }
            {
                const double c = (nd.bmax[1] + nd.bmin[1]) / 2;
                const double a0 = ((nd.bmin[1] - 0.1) - o.y) * invDy, a1 = ((c + 0.1) - o.y) * invDy;
                const double b0 = ((c - 0.1) - o.y) * invDy, b1 = ((nd.bmax[1] + 0.1) - o.y) * invDy;
                const bool neg = invDy < 0;
                ny[0] = neg ? a1 : a0; fy[0] = neg ? a0 : a1; ny[1] = neg ? b1 : b0; fy[1] = neg ? b0 : b1;
            }
            {
                const double c = (nd.bmax[2] + nd.bmin[2]) / 2;
                const double a0 = ((nd.bmin[2] - 0.1) - o.z) * invDz, a1 = ((c + 0.1) - o.z) * invDz;
                const double b0 = ((c - 0.1) - o.z) * invDz, b1 = ((nd.bmax[2] + 0.1) - o.z) * invDz;
                const bool neg = invDz < 0;
                nz[0] = neg ? a1 : a0; fz[0] = neg ? a0 : a1; nz[1] = neg ? b1 : b0; fz[1] = neg ? b0 : b1;
            }
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
                bool p = !(tmx < tmn || tmx < 0 || tmn > cb || tmx < ca);                         // :268
                // a child the state prunes already -- hit && closestT <= its clamped entry, :210 -- stays pruned until it would be popped
                // (closestT only falls): it is not pushed.  (tmn is the child's own entry parameter, bit for bit: derived planes, above.)
                p = p && !(hit && closestT <= mx(tmn, ca));
                pushed |= p ? (1u << oct) : 0u;
            }
            pushed &= ~((unsigned)nd.pad & 255u);                   // children that are empty leaves (the device copy's mask, device_scene.cpp): popping one
                                                                   // has no effect ("Octree - alt.cs":213: the list loop does not run), a third of K2p's pops
            unsigned byc = 0;                                      // octant bit -> cursor bit: cursor k examines octant k ^ mask
#pragma unroll
            for (int k = 0; k < 8; ++k) byc |= ((pushed >> (k ^ mask)) & 1u) << k;
            if (byc != 0) {                                        // a node that pushes nothing opens no frame: every open frame has a child to pop
                ++lvl;
                fa[lvl * nt + tid] = ca;
                fb[lvl * nt + tid] = cb;
                fpk[lvl * nt + tid] = (int)(((unsigned)nd.first_child << 8) | byc);
            }
        }
    };

    for (;;) {
        // ------------------------------------------------------------------ refill idle lanes
        const unsigned long long idle = __ballot(!alive);
        if (__builtin_expect(!drained && (__popcll(idle) >= REFILL_MIN_IDLE || idle == ~0ull), 0)) {
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
                    hit = false; parked = false; alive = true;
                    tame = fabs(o.x) < 1e300 && fabs(o.y) < 1e300 && fabs(o.z) < 1e300 && fabs(d.x) < 1e300 && fabs(d.y) < 1e300 && fabs(d.z) < 1e300;
                    // ... and 1/d has the sign of d on every axis (a component within 1e-16 of zero gets +1e16 whatever its sign, :165-167):
                    // the fast visit forms its slabs in cursor order from the direction mask alone
                    tame = tame && (d.x < 0) == (fabs(d.x) > 1e-16 && d.x < 0) && (d.y < 0) == (fabs(d.y) > 1e-16 && d.y < 0) &&
                           (d.z < 0) == (fabs(d.z) > 1e-16 && d.z < 0);
                    tight_ok = tame && g.tight != nullptr && fabs(o.x - g.tight_mid[0]) <= g.tight_rad && fabs(o.y - g.tight_mid[1]) <= g.tight_rad &&
                               fabs(o.z - g.tight_mid[2]) <= g.tight_rad;
                    closestT = kDblMax; pid = -1; bu = 0; bv = 0;
                    m0 = m1 = m2 = m3 = -1;
                    lvl = -1; q = 0; qe = 0;
                    np = 0; leaf_fresh = false; cur_skip = false;
                    if (e1 == -2 && (io.flags & SHOOT_RETIRED_RAYS)) {
                        finish();
                    } else {
                        nrays++;
                        live_lane = true;
                        invDx = fabs(d.x) > 1e-16 ? 1.0 / d.x : 1e16;      // "Octree - alt.cs":165-167
                        invDy = fabs(d.y) > 1e-16 ? 1.0 / d.y : 1e16;
                        invDz = fabs(d.z) > 1e-16 ? 1.0 / d.z : 1e16;
                        mask = ((d.x >= 0 ? 0 : 1) << 2) | ((d.y >= 0 ? 0 : 1) << 1) | (d.z >= 0 ? 0 : 1);
                        cray = cull_ray(g, o.x, o.y, o.z, d.x, d.y, d.z);
                        const OctNode& root = g.nodes[0];
                        double tx0 = (root.bmin[0] - o.x) * invDx, tx1 = (root.bmax[0] - o.x) * invDx;
                        double ty0 = (root.bmin[1] - o.y) * invDy, ty1 = (root.bmax[1] - o.y) * invDy;
                        double tz0 = (root.bmin[2] - o.z) * invDz, tz1 = (root.bmax[2] - o.z) * invDz;
                        if (invDx < 0) { const double s = tx0; tx0 = tx1; tx1 = s; }
                        if (invDy < 0) { const double s = ty0; ty0 = ty1; ty1 = s; }
                        if (invDz < 0) { const double s = tz0; tz0 = tz1; tz1 = s; }
                        const double rmin = omax(omax(tx0, ty0), tz0), rmax = omin(omin(tx1, ty1), tz1);   // :182-183
                        if (OWN) ownw.cells++;                                // the root's record
                        if (rmax < rmin || rmax < 0) finish();               // :185 (and the identical pop test :207)
                        else visit(root, rmin, rmax, std::false_type{});
                        // (Sending the root through the pop phase instead -- a level-0 frame with node 0 as its only child -- takes the
                        //  root's visit out of this set-up path, where the compiler spills ~20 registers around it, and was measured:
                        //  bit-exact, 2.88 -> 3.32 ms at 1M rays.  The spills sit here and in the kernel's prologue only, once per ray.)
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
            // tickets dry and down to the last few rays, which have outlived the rest of the batch by HARE_K2P_TAIL_PATIENCE rounds:
            // they go to the cooperative tail kernel (octree_coop.hip), this wave ends
            if (!OCC && drained && io.oct_tail != nullptr) {
                if (__popcll(am) <= io.oct_tail_max && tail_rounds >= io.oct_tail_patience) {
                    if (!DENSE || __ballot(alive && np > 0) == 0) break;
                    leaving = true;                          // DENSE: the records hold no pending survivors -- one exact phase first
                }
                ++tail_rounds;
            }
        }

        K2P_STAT(sp_round++; sp_alive += __popcll(__ballot(alive)); if (drained) dr_round++;)
        K2P_STAT({ const unsigned long long now = __builtin_amdgcn_s_memrealtime(); if (dr_few) dr_t[3] += now - dr_last; dr_last = now;
                   dr_few = drained && __popcll(__ballot(alive)) <= 8; })
        // ------------------------------------------------------------------ phase P: one child per step
        // (DENSE) once the tickets are dry the chip empties and a wave's time is the chain of dependent loads of its last rays, a node
        // record per pop: more pop steps per round then (HARE_K2D_STEPS_DRAIN), which the steady state cannot afford
        const int steps_now = (DENSE && drained) ? HARE_K2D_STEPS_DRAIN : STEPS;
#pragma unroll 1
        for (int k = 0; k < steps_now; ++k) {
            // DENSE: a lane whose walk is over but which still holds survivors waits for the exact phase; so does one whose list is full
            if (DENSE && alive && q == qe && lvl < 0 && np == 0) finish();
            const bool pop = DENSE ? (alive && !leaving && np < P && q == qe && lvl >= 0) : (alive && !parked && q == qe);
            {
                const unsigned long long pm = __ballot(pop);
                if (pm == 0 || (DENSE && k > 0 && __popcll(pm) < HARE_K2D_POP_MIN)) break;     // a further step only when enough lanes take it
            }
            K2P_STAT(sp_p++; sp_pl += __popcll(__ballot(pop)); if (drained) dr_p++;)
            // Rays whose components are all finite and far from overflow never produce a NaN here (1/d is finite and non-zero,
            // boxes are finite), so for them Math.Max / Math.Min are the hardware's v_max_f64 / v_min_f64 (the sign of a zero
            // result is only ever compared); anything else takes the NaN-propagating compare-selects.
            auto pstep = [&](auto fast_tag) {
                constexpr bool FAST = decltype(fast_tag)::value;
                auto mx = [](double a, double b) { return FAST ? __builtin_fmax(a, b) : omax(a, b); };
                auto mn = [](double a, double b) { return FAST ? __builtin_fmin(a, b) : omin(a, b); };
                if (lvl < 0) {
                    finish();                                                // stack empty: :276-283
                } else {
                    const int pk = fpk[lvl * nt + tid];
                    const unsigned rem = (unsigned)pk & 255u;
                    if (rem == 0) {
                        --lvl;                                               // (never since round 4: frames are closed with their last child, below)
                        K2P_STAT(st_kind = 1;)
                    } else {
                        const int cur = 31 - __builtin_clz(rem);             // pop order: order[7] down to order[0]
                        const double pa = fa[lvl * nt + tid], pb = fb[lvl * nt + tid];
                        // the frame's last child closes the frame: what that child opens takes its place, and no step is spent on finding
                        // a frame empty (a third of the pop steps before: tools/k2p_stats.py).  The stack is the reference's all the same --
                        // a frame without children left holds nothing that is ever popped.
                        if ((rem & (rem - 1u)) == 0u) --lvl;
                        else fpk[lvl * nt + tid] = pk & ~(1 << cur);
                        const int c = (int)((unsigned)pk >> 8) + (cur ^ mask);
                        // the whole 64-byte record in ONE batch of loads: left to itself the compiler fetches the last four words -- child
                        // index, list words -- in the branches that use them, a second and a third wait per pop
                        OctNode nd = g.nodes[c];
                        if (OWN) ownw.cells++;
                        // ... and, in the same batch, the box of the polygons its subtree lists (device_scene.cpp: make_tight_boxes)
                        float4 tb0 = make_float4(0, 0, 0, 0), tb1 = tb0;
                        if (FAST && g.tight != nullptr) {
                            const float4* tp = reinterpret_cast<const float4*>(g.tight) + 2 * (size_t)c;
                            tb0 = tp[0]; tb1 = tp[1];
                        }
                        asm volatile("" : "+v"(nd.first_child), "+v"(nd.item_start), "+v"(nd.item_count), "+v"(nd.pad));
                        // the child's slab interval from its own box (:253-266)
                        double tx0 = (nd.bmin[0] - o.x) * invDx, tx1 = (nd.bmax[0] - o.x) * invDx;
                        double ty0 = (nd.bmin[1] - o.y) * invDy, ty1 = (nd.bmax[1] - o.y) * invDy;
                        double tz0 = (nd.bmin[2] - o.z) * invDz, tz1 = (nd.bmax[2] - o.z) * invDz;
                        if (invDx < 0) { const double sw = tx0; tx0 = tx1; tx1 = sw; }
                        if (invDy < 0) { const double sw = ty0; ty0 = ty1; ty1 = sw; }
                        if (invDz < 0) { const double sw = tz0; tz0 = tz1; tz1 = sw; }
                        const double tmn = mx(mx(tx0, ty0), tz0), tmx = mn(mn(tx1, ty1), tz1);
                        const double ca = mx(tmn, pa), cb = mn(tmx, pb);                     // :271
                        K2P_STAT(st_kind = 2;)
                        // The tight box: a ray that misses it -- or, holding a hit, reaches it behind that hit -- cannot make RayXtri accept
                        // (:224-225, t > 1e-10 && t < closestT) any polygon of the subtree: the node is dropped as if it had been visited in
                        // vain.  FP64 on the reference's own slab expressions; the box is grown by 2^-20 of the scene's extent, a million
                        // times the rounding of either test, and rounded outwards (make_tight_boxes).  A stale closestT only drops less.
                        bool vain = false;
                        if (FAST && g.tight != nullptr) {
                            double ux0 = ((double)tb0.x - o.x) * invDx, ux1 = ((double)tb0.w - o.x) * invDx;
                            double uy0 = ((double)tb0.y - o.y) * invDy, uy1 = ((double)tb1.x - o.y) * invDy;
                            double uz0 = ((double)tb0.z - o.z) * invDz, uz1 = ((double)tb1.y - o.z) * invDz;
                            if (invDx < 0) { const double sw = ux0; ux0 = ux1; ux1 = sw; }
                            if (invDy < 0) { const double sw = uy0; uy0 = uy1; uy1 = sw; }
                            if (invDz < 0) { const double sw = uz0; uz0 = uz1; uz1 = sw; }
                            const double un = mx(mx(ux0, uy0), uz0), uf = mn(mn(ux1, uy1), uz1);
                            vain = tight_ok && ((uf < un) | (uf < 0) | (hit & (closestT <= un)));
                        }
                        if (!vain && !(cb < ca || cb < 0) && !(hit && closestT <= ca)) {   // popped and kept (:207-211)
                            K2P_STAT(st_kind = nd.first_child < 0 ? 3 : 4;)
                            visit(nd, ca, cb, fast_tag);
                        }
                    }
                }
            };
            const bool all_tame = __ballot(pop && !tame) == 0;
            K2P_STAT(st_kind = 0;)
            if (pop) {
                if (all_tame) pstep(std::true_type{});
                else pstep(std::false_type{});
            }
            K2P_STAT({ const int ni = __popcll(__ballot(st_kind == 4)); sp_is += ni ? 1 : 0; sp_il += ni; const int nl = __popcll(__ballot(st_kind == 3));
                       sp_ls += nl ? 1 : 0; sp_ll += nl; sp_xl += __popcll(__ballot(st_kind == 1)); sp_fl += __popcll(__ballot(st_kind == 2)); })
        }

        K2P_STAT({ const unsigned long long now = __builtin_amdgcn_s_memrealtime(); if (dr_few) dr_t[0] += now - dr_last; dr_last = now; })
        if (DENSE) {
        // ------------------------------------------------------------------ DENSE B1: every leaf entry in hand, one per lane
        {
            const bool own = alive && !leaving && np < P && q < qe;
            const int cnt = own ? (qe - q < HARE_K2D_CAP ? qe - q : HARE_K2D_CAP) : 0;
            const int inc = wave_scan_add(cnt);                              // inclusive scan of the counts over the lanes
            const int off = inc - cnt;
            const int total = __builtin_amdgcn_readlane(inc, 63);
            const int q0 = q;                                                // where this lane's segment starts in its list
            bool stop = false;                                               // owner: its list is full, the rest of its segment waits
            // which owner does item (base + lane) belong to?  Owners mark the start of their segment inside this window (or position
            // 0 when the segment began before it); an inclusive max-scan spreads the owner's lane number over its items.  Returns the
            // item's list position (a lane past the end: position 0 -- total > 0, so the list has an entry to load)
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
                const int rel_o = __shfl(q0 - off, ow, 64);                   // list index of item x = rel_o + x
                return valid ? rel_o + base + lane : 0;
            };
#if HARE_K2D_AHEAD
            // Round 6: the windows as a pipeline.  A window is two dependent gathers -- the list entry, then that polygon's pre-cull
            // record -- and a caller-sized batch leaves a wave nothing to hide them behind (profiles/r06_experiments: a round is ~9
            // dependent round trips, 5.7 of them these).  What a window reads depends on nothing the windows before it decide (counts
            // and offsets are fixed at the top), so the entries are requested two windows ahead and the records one: a window's
            // loads fly during the window before it.  Loads are unconditional (lanes past the end read entry 0) so that no branch
            // sits between an issue and its use.
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
                K2P_STAT(sp_c++; sp_cl += (total - base < 64 ? total - base : 64); if (drained) dr_c++;)
                const bool valid = base + lane < total;
#if HARE_K2D_AHEAD
                const int ow = ow1, i_now = i1;
                const CullRaw rec = rec1;
                ow1 = ow2; i1 = i2;
                rec1 = cull_load(g, i1);                                      // the next window's records (entry 0 again when there is none)
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
                // the owner's pre-cull operands: six shuffles (what can be rebuilt from them is rebuilt: cull_ray's |d|_1 and error
                // bound; the owner's exclusions are applied by the owner, below)
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
                // owner side: the survivors of its segment, in list order, as far as its pending list has room
                const int lo = off > base ? off - base : 0;
                const int hi = off + cnt - base < 64 ? off + cnt - base : 64;
                const bool mine = cnt > 0 && !stop && lo < hi && hi > 0 && lo < 64;
                unsigned long long seg = 0;
                if (mine) {
                    const unsigned long long m_hi = hi >= 64 ? ~0ull : ((1ull << hi) - 1ull);
                    seg = sb & m_hi & ~((1ull << lo) - 1ull);
                }
                int consumed = mine ? hi - lo : 0;                           // entries of this window the owner is done with
#pragma unroll
                for (int r = 0; r < P; ++r) {
                    const bool take = mine && seg != 0 && np < P;
                    const int pos = take ? (int)__builtin_ctzll(seg) : lane;
                    const int poly = __shfl(i, pos, 64);                     // every lane executes the shuffle
                    if (take) {
                        // :218, applied by the owner -- and (round 6, HARE_K2D_SKIP_PID) the polygon the ray's hit lies on is not tested AGAIN: a polygon is
                        // listed in every leaf it touches, the walk goes on behind a hit (F15), and the same ray against the same polygon yields
                        // the same t bit for bit -- `t < closestT` (:225) fails, nothing changes.  Three of four exact tests were such repeats.
                        if (poly != e1 && poly != e2 && !(HARE_K2D_SKIP_PID && hit && poly == pid)) {
                            pend_lca[np * nt + tid] = leaf_ca;
                            pend_w[np * nt + tid] = poly | (leaf_fresh ? (int)0x80000000 : 0);
                            leaf_fresh = false;
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
        K2P_STAT({ const unsigned long long now = __builtin_amdgcn_s_memrealtime(); if (dr_few) dr_t[1] += now - dr_last; dr_last = now; })
        // ------------------------------------------------------------------ DENSE B2: the noted survivors, each lane its own, in order
        {
            const bool over = alive && lvl < 0 && q == qe;                   // nothing left to visit
            const unsigned long long holding = __ballot(alive && np > 0);
            const unsigned long long blocked = __ballot(alive && np > 0 && (np >= P || over));
            if (holding != 0 && (blocked != 0 || leaving || __popcll(holding) >= HARE_K2D_EXACT_MIN)) {
                K2P_STAT(sp_e++; sp_el += __popcll(holding); if (drained) dr_e++;)
                bool ended = false;
#pragma unroll 1
                for (int k = 0; k < P; ++k) {
                    const bool act = alive && !ended && k < np;
                    if (__ballot(act) == 0) break;
                    if (act) {
                        const int w = pend_w[k * nt + tid];
                        const double lk = pend_lca[k * nt + tid];
                        const int i = w & 0x7FFFFFFF;
                        const PolyRec& p = g.polys[i];
                        const double* v3 = (g.quads && g.quads[i].nverts == 4) ? g.quads[i].v3 : nullptr;
                        double t, u, v;
                        if (OWN) ownw.tests++;
                        const bool ok = poly_full(p, v3, o, d, t, u, v) && t > kTMin;                  // :224
                        if (w < 0) cur_skip = hit && closestT <= lk;              // a new leaf: :210 as at its pop (no accept lies in between)
                        if (ok && !cur_skip && t < closestT) {                                          // :225
                            closestT = t; bu = u; bv = v; pid = i;
                            hit = true;
                            if (closestT <= lk) ended = true;                                           // :233
                            else if (OCC && (io.tmax == nullptr || closestT < io.tmax[ray])) ended = true;   // any hit below t_max decides the flag
                        }
                    }
                }
                if (alive && np > 0) {
                    np = 0;
                    if (ended) finish();
                }
            }
            if (alive && lvl < 0 && q == qe && np == 0) finish();             // the walk is over and nothing is pending
            K2P_STAT({ const unsigned long long now = __builtin_amdgcn_s_memrealtime(); if (dr_few) dr_t[2] += now - dr_last; dr_last = now; })
            if (leaving && __ballot(alive && np > 0) == 0) break;             // the hand-over records can be written now
        }
        } else {
        // ------------------------------------------------------------------ phase B1: FP32 cull (leaf candidates)
        // Two candidates per iteration: both list entries, then both polygon records, are requested together and the culls
        // run back to back -- two culls per pair of dependent loads instead of one.  (This kernel is held to three workgroups
        // per CU by its LDS frames, so the second record in flight costs no occupancy; the voxel kernel has no such room.)
        auto load_rec = [&](int i) { return cull_load(g, i); };
        auto culled = [&](const CullRaw& r) { return cull_test(g, cray, r); };
        auto recently = [&](int i) {                                                                    // :218 (+ mailbox)
            return i == e1 || i == e2 || (HARE_K2D_SKIP_PID && hit && i == pid) || (HARE_K2P_MAILBOX >= 1 && i == m0) || (HARE_K2P_MAILBOX >= 2 && i == m1) ||
                   (HARE_K2P_MAILBOX >= 4 && (i == m2 || i == m3));
        };
#pragma unroll 1
        for (int kc = 0; kc < CULLS / 2; ++kc) {
            const bool culling = alive && !parked && q < qe;
            if (__ballot(culling) == 0) break;
            K2P_STAT(sp_c++; sp_cl += __popcll(__ballot(culling));)
            if (culling) {
                const bool has1 = q + 1 < qe;
                // the two entries AFTER this pair, for the next iteration: one 8-byte gather (4-byte alignment is all the hardware asks
                // for), requested together with the pair's records -- so an iteration has one round of dependent loads, not two
                int i2 = -1, i3 = -1;
                if (qe - q >= 4) {
                    const int2 w = *reinterpret_cast<const int2*>(g.items + q + 2);
                    i2 = w.x; i3 = w.y;
                } else if (q + 2 < qe) {
                    i2 = g.items[q + 2];                                     // q + 3 >= qe: i3 is never a candidate
                }
                const int i0 = idx, i1 = has1 ? nexti : -1;
                const bool sk0 = recently(i0);
                const bool sk1 = !has1 || recently(i1) || i1 == i0;
                CullRaw ra, rb;
                if (!sk0) ra = load_rec(i0);
                if (!sk1) rb = load_rec(i1);
                const int q_before = q;
                bool consumed0 = true;
                if (!sk0) {
                    if (HARE_K2P_MAILBOX >= 4) { m3 = m2; m2 = m1; }
                    if (HARE_K2P_MAILBOX >= 2) m1 = m0;
                    if (HARE_K2P_MAILBOX >= 1) m0 = i0;
                    if (culled(ra)) ++q;
                    else { parked = true; consumed0 = false; }           // phase B2 tests idx (= items[q])
                } else {
                    ++q;
                }
                if (consumed0 && has1) {
                    if (sk1) {
                        ++q;
                    } else {
                        if (HARE_K2P_MAILBOX >= 4) { m3 = m2; m2 = m1; }
                        if (HARE_K2P_MAILBOX >= 2) m1 = m0;
                        if (HARE_K2P_MAILBOX >= 1) m0 = i1;
                        if (culled(rb)) ++q;
                        else parked = true;
                    }
                }
                const int adv = q - q_before;                                // 0, 1 or 2 entries consumed: slide the window
                idx = adv == 2 ? i2 : (adv == 1 ? nexti : idx);
                nexti = adv == 2 ? i3 : (adv == 1 ? i2 : nexti);
            }
        }

        // ------------------------------------------------------------------ phase B2: exact test with u,v
        {
            const unsigned long long pm = __ballot(alive && parked);
            const unsigned long long busy = __ballot(alive && !parked);
            if (pm != 0 && (__popcll(pm) >= EXACT_MIN_PARKED || busy == 0)) {
                K2P_STAT(sp_e++; sp_el += __popcll(pm);)
                if (alive && parked) {
                    const int i = idx;                                       // == items[q]
                    const PolyRec& p = g.polys[i];
                    const double* v3 = (g.quads && g.quads[i].nverts == 4) ? g.quads[i].v3 : nullptr;
                    double t, u, v;
                    parked = false;
                    ++q;
                    idx = nexti;
                    nexti = q + 1 < qe ? g.items[q + 1] : -1;                // requested with the polygon record: not a round of its own
                    if (poly_full(p, v3, o, d, t, u, v) && t > kTMin && t < closestT) {   // :224-226
                        closestT = t; bu = u; bv = v; pid = i;
                        hit = true;
                        if (closestT <= leaf_ca) finish();                                // :233
                        else if (OCC && (io.tmax == nullptr || closestT < io.tmax[ray])) finish();   // any hit below t_max decides the flag
                    }
                }
            }
        }
        }
    }

    if (!OCC && io.oct_tail != nullptr) {
        // ---- hand-over to K2t: every lane still alive writes its walk state -- scalars, then its frames from LDS
        const unsigned long long am = __ballot(alive);
        if (am) {
            unsigned base = 0;
            if (lane == 0) base = atomicAdd(&reinterpret_cast<LaunchSlotMem*>(io.work)->oct_tail_count, (unsigned)__popcll(am));
            base = (unsigned)__builtin_amdgcn_readfirstlane((int)base);
            if (alive) {
                unsigned char* rec = io.oct_tail + (size_t)(base + rank_below(am)) * (size_t)io.oct_tail_stride;
                OctTailRec h;
                h.ray = ray; h.lvl = lvl; h.q = q; h.qe = qe; h.leaf_ca = leaf_ca;
                h.closestT = closestT; h.bu = bu; h.bv = bv; h.pid = pid; h.hit = hit ? 1 : 0; h.pad = 0;
                *reinterpret_cast<OctTailRec*>(rec) = h;
                double* ra = reinterpret_cast<double*>(rec + kOctTailHead);
                double* rb = ra + io.oct_tail_levels;
                int* rk = reinterpret_cast<int*>(rb + io.oct_tail_levels);
                for (int k = 0; k <= lvl; ++k) { ra[k] = fa[k * nt + tid]; rb[k] = fb[k * nt + tid]; rk[k] = fpk[k * nt + tid]; }
            }
        }
    }
    timeline(2);
#ifdef HARE_K2P_STATS
    if ((io.flags & 0x2000u) != 0 && io.prof != nullptr && lane == 0)       // rounds | pop steps | dense windows | exact phases after the tickets ran dry
        io.prof[32 + 4ull * (blockIdx.x * 4u + (threadIdx.x >> 6)) + 3] = (unsigned long long)(dr_round & 0xFFFFu) | ((unsigned long long)(dr_p & 0xFFFFu) << 16) |
                                                                            ((unsigned long long)(dr_c & 0xFFFFu) << 32) | ((unsigned long long)(dr_e & 0xFFFFu) << 48);
    if ((io.flags & 0x2000u) != 0 && io.prof != nullptr && lane == 0)
        for (int k = 0; k < 4; ++k) io.prof[32 + 4ull * 4096ull + 4ull * (blockIdx.x * 4u + (threadIdx.x >> 6)) + k] = dr_t[k];
    if (lane == 0 && io.prof) {
        const unsigned long long v[14] = {sp_round, sp_alive, sp_p, sp_pl, sp_c, sp_cl, sp_e, sp_el, sp_is, sp_il, sp_ls, sp_ll, sp_xl, sp_fl};
        for (int k = 0; k < 14; ++k) atomicAdd(&io.prof[k], v[k]);
    }
#endif
    if (OWN) flush_own(io.ctr, ownw);
    launch_epilogue(io, nrays, nhits, 4u);
}

// K2t: the rays K2p handed over, one per group of kOctTailGroup lanes (hare_device.h: 64, a whole wave per ray; octree_coop.hip).
// Grid: any number of 256-thread workgroups; dynamic LDS = kOctTailGroupsPerBlock groups x levels x 20 bytes.
__device__ __forceinline__ void octree_tail_body(const OctreeArgs& g, const ShootIO& io)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    constexpr int G = kOctTailGroup, NG = 256 / G;        // groups per workgroup
    const int gl = threadIdx.x & (G - 1), grp = threadIdx.x / G;
    const int levels = io.oct_tail_levels;
    double* const fa = reinterpret_cast<double*>(lds) + (size_t)grp * levels;
    double* const fb = reinterpret_cast<double*>(lds) + (size_t)(NG + grp) * levels;
    int* const fpk = reinterpret_cast<int*>(reinterpret_cast<double*>(lds) + (size_t)2 * NG * levels) + (size_t)grp * levels;
    LaunchSlotMem* const sm = reinterpret_cast<LaunchSlotMem*>(io.work);
    const unsigned count = sm->oct_tail_count;            // final: K2p has ended (stream order)
    unsigned nhits = 0;
    for (;;) {                                            // every group draws its own tickets: a group never waits for its neighbours
        unsigned i = 0;
        if (gl == 0) i = atomicAdd(&sm->oct_tail_next, 1u);
        i = (unsigned)__shfl((int)i, 0, G);
        if (i >= count) break;
        const unsigned char* rec = io.oct_tail + (size_t)i * (size_t)io.oct_tail_stride;
        const OctTailRec h = *reinterpret_cast<const OctTailRec*>(rec);          // one address for the group
        const double* ra = reinterpret_cast<const double*>(rec + kOctTailHead);
        const double* rb = ra + levels;
        const int* rk = reinterpret_cast<const int*>(rb + levels);
        for (int k = gl; k <= h.lvl; k += G) { fa[k] = ra[k]; fb[k] = rb[k]; fpk[k] = rk[k]; }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        double closestT = h.closestT, bu = h.bu, bv = h.bv;
        int pid = h.pid;
        bool hit = h.hit != 0;
        coop_octree<G>(g, io, fa, fb, fpk, h.ray, h.lvl, h.q, h.qe, h.leaf_ca, closestT, bu, bv, pid, hit);
        if (gl == 0) {
            XEventRec ev;
            if (hit) {
                const RayRec r = io.rays[h.ray];
                ev.t = closestT; ev.u = bu; ev.v = bv;
                ev.x = r.x + r.dx * closestT; ev.y = r.y + r.dy * closestT; ev.z = r.z + r.dz * closestT;
                ev.poly_id = pid;
                ev.hit = 1;
                nhits++;
            } else {
                set_miss(ev);
            }
            store_event_streaming(&io.out[h.ray], ev);
        }
    }
    // the rays were counted by K2p when it set them up; their hits are counted here.  The last wave of this grid leaves the hand-over
    // counters zeroed for the launch that uses the slot next.
    const unsigned long long wh = wave_sum_u32(gl == 0 ? nhits : 0u);          // valid in lane 0 of the wave
    if ((threadIdx.x & 63) == 0) {
        if (io.ctr && wh) atomicAdd(&io.ctr[CTR_HITS], wh);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (atomicAdd(&sm->oct_tail_done, 1u) == gridDim.x * 4u - 1u) {
            atomicExch(&sm->oct_tail_count, 0u);
            atomicExch(&sm->oct_tail_next, 0u);
            atomicExch(&sm->oct_tail_done, 0u);
        }
    }
}


template <bool COUNT>
__device__ __forceinline__ void kdtree_shoot_body(const KdArgs& g, const ShootIO& io)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    int* stack = reinterpret_cast<int*>(lds);
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool valid = i < io.n;
    XEventRec ev;
    set_miss(ev);
    Work w = {0, 0, 0};
    bool live = false;
    if (valid) {
        const RayRec r = io.rays[i];
        const V3 o = {r.x, r.y, r.z};
        const V3 d = {r.dx, r.dy, r.dz};
        const int e1 = io.excl1 ? io.excl1[i] : -1;
        const int e2 = io.excl2 ? io.excl2[i] : -1;
        live = !(e1 == -2 && (io.flags & SHOOT_RETIRED_RAYS));
        if (live) trace_kdtree<COUNT, !COUNT>(g, stack, threadIdx.x, blockDim.x, o, d, e1, e2, ev, w);     // the counting build runs the reference's walk as it is
        if (io.out) io.out[i] = ev;
        if (io.occluded) {
            const bool occ = ev.hit != 0 && (io.tmax == nullptr || ev.t < io.tmax[i]);
            io.occluded[i] = occ ? 1 : 0;
            if (!io.out) ev.hit = occ ? 1 : 0;        // flags only: the batch counter `hits` counts occluded rays
        }
    }
    flush_counters(io.ctr, valid && live, ev.hit != 0, w, COUNT);
}

}  // namespace

extern "C" {

// K1: Voxel_Grid.Shoot, triangles only
__global__ __launch_bounds__(256) void hare_voxel_shoot_tri(VoxelArgs g, ShootIO io)
{
    voxel_shoot_body<false, false>(g, io);
}
// K1 with quadrilaterals present
__global__ __launch_bounds__(256) void hare_voxel_shoot_quad(VoxelArgs g, ShootIO io)
{
    voxel_shoot_body<true, false>(g, io);
}
// K1 + exact work counters (cells / list entries / tests) for diagnostics
__global__ __launch_bounds__(256) void hare_voxel_shoot_count(VoxelArgs g, ShootIO io)
{
    voxel_shoot_body<true, true>(g, io);
}

// K1p: persistent Voxel_Grid.Shoot (default voxel kernel); dynamic LDS = the occupancy bitmap, one bit per voxel
__global__ __launch_bounds__(256, 4) void hare_voxel_persist_tri(VoxelArgs g, ShootIO io) { voxel_persist_body<false, false>(g, io); }
__global__ __launch_bounds__(256, 4) void hare_voxel_persist_quad(VoxelArgs g, ShootIO io) { voxel_persist_body<true, false>(g, io); }
// same for grids above ~80^3, whose bitmap has one bit per block of 2^k voxels per axis so that it still fits 64 KB of LDS
__global__ __launch_bounds__(256, 4) void hare_voxel_persist_tri_g(VoxelArgs g, ShootIO io) { voxel_persist_body<false, true>(g, io); }
__global__ __launch_bounds__(256, 4) void hare_voxel_persist_quad_g(VoxelArgs g, ShootIO io) { voxel_persist_body<true, true>(g, io); }

#ifndef HARE_OCCL_WAVES_PER_EU
#define HARE_OCCL_WAVES_PER_EU 4
#endif
// the occlusion predicate on K1p's walk, cut short at t_max (flags only: ShootIO::occluded, ShootIO::tmax); same launch geometry
__global__ __launch_bounds__(256, HARE_OCCL_WAVES_PER_EU) void hare_voxel_occl_tri(VoxelArgs g, ShootIO io) { voxel_persist_body<false, false, false, HARE_K1P_STEPS, HARE_K1P_CULLS, true>(g, io); }
__global__ __launch_bounds__(256, HARE_OCCL_WAVES_PER_EU) void hare_voxel_occl_quad(VoxelArgs g, ShootIO io) { voxel_persist_body<true, false, false, HARE_K1P_STEPS, HARE_K1P_CULLS, true>(g, io); }
__global__ __launch_bounds__(256, HARE_OCCL_WAVES_PER_EU) void hare_voxel_occl_tri_g(VoxelArgs g, ShootIO io) { voxel_persist_body<false, true, false, HARE_K1P_STEPS, HARE_K1P_CULLS, true>(g, io); }
__global__ __launch_bounds__(256, HARE_OCCL_WAVES_PER_EU) void hare_voxel_occl_quad_g(VoxelArgs g, ShootIO io) { voxel_persist_body<true, true, false, HARE_K1P_STEPS, HARE_K1P_CULLS, true>(g, io); }

// developer profiling build of the persistent kernel (phase stamps into ShootIO::prof)
__global__ __launch_bounds__(256) void hare_voxel_persist_prof(VoxelArgs g, ShootIO io) { voxel_persist_body<false, false, true>(g, io); }

// tests only: FP32-cull audit against the exact test, all polygons x all rays
__global__ __launch_bounds__(256) void hare_cull_audit(VoxelArgs g, ShootIO io) { audit_body(g, io); }

// K2: Octree.Shoot ("Octree - alt.cs":159-284); dynamic LDS = levels * blockDim * 24 bytes
__global__ __launch_bounds__(256) void hare_octree_shoot(OctreeArgs g, ShootIO io) { octree_shoot_body<false>(g, io); }
__global__ __launch_bounds__(256) void hare_octree_shoot_count(OctreeArgs g, ShootIO io) { octree_shoot_body<true>(g, io); }

// K2p: persistent Octree.Shoot (default octree kernel); dynamic LDS = levels * blockDim * 24 bytes
#ifndef HARE_K2D_WAVES_PER_EU
#define HARE_K2D_WAVES_PER_EU 3
#endif
#ifndef HARE_K2P_WAVES_PER_EU
#define HARE_K2P_WAVES_PER_EU 4
#endif
__global__ __launch_bounds__(256, HARE_K2P_WAVES_PER_EU) void hare_octree_persist(OctreeArgs g, ShootIO io) { octree_persist_body<false>(g, io); }
// K2t: the cooperative tail behind K2p (octree_coop.hip)
__global__ __launch_bounds__(256) void hare_octree_tail(OctreeArgs g, ShootIO io) { octree_tail_body(g, io); }
// the occlusion predicate on the same walk (flags only, any-hit early out); same launch geometry
// (round 5: K2d's OCC build -- the dense walk with the any-hit early out; K2p's, which this was until then, had fallen behind the closest-hit
//  kernel it was meant to beat: 457 against 784 Mrays/s at t_max = half the mean free path)
__global__ __launch_bounds__(256, HARE_K2D_WAVES_PER_EU) void hare_octree_occl(OctreeArgs g, ShootIO io) { octree_persist_body<true, true>(g, io); }
// ... and K2p's stays for queries WITHOUT a t_max (any hit at all decides): there the first accepted test ends the ray, and testing a leaf's
// entries at once beats deferring them -- 2356 against the dense build's 1627 Mrays/s
__global__ __launch_bounds__(256, HARE_K2P_WAVES_PER_EU) void hare_octree_occl_any(OctreeArgs g, ShootIO io) { octree_persist_body<true>(g, io); }
// K2d: K2p with its leaf entries spread densely over the wave and its exact tests deferred (octree_persist_body<.., DENSE>);
// dynamic LDS = K2p's frames + kOctDenseExtra bytes per workgroup.  Three waves per SIMD (round 6): from seven levels on the frames admit no
// more than three workgroups per CU anyway, and with 168 registers the kernel spills nothing (at 128: 16 VGPRs, reloaded inside the pop loop;
// 1M rays +2 %, profiles/r06_experiments/k2d_windows_pipelined.log)
__global__ __launch_bounds__(256, HARE_K2D_WAVES_PER_EU) void hare_octree_dense(OctreeArgs g, ShootIO io) { octree_persist_body<false, true>(g, io); }
// ... and its counting build (HARE_SHOOT_COUNT_OWN)
__global__ __launch_bounds__(256, HARE_K2D_WAVES_PER_EU) void hare_octree_dense_own(OctreeArgs g, ShootIO io) { octree_persist_body<false, true, true>(g, io); }

// KDTree.Shoot (KDTree.cs:204-361); dynamic LDS = (depth + 2) * blockDim * 4 bytes
__global__ __launch_bounds__(256) void hare_kdtree_shoot(KdArgs g, ShootIO io) { kdtree_shoot_body<false>(g, io); }
__global__ __launch_bounds__(256) void hare_kdtree_shoot_count(KdArgs g, ShootIO io) { kdtree_shoot_body<true>(g, io); }

// K3: specular bounce (harness-defined): o' = X_Point, d' = d - (2*(d.n))*n, next exclusion = the polygon just hit; rays
// that missed are marked dead (-2).  Returns whether the ray lives on.
// marks_valid (the bounce loop from its second reflection on): excl_out already holds the previous reflection's marks, and a ray marked -2 is
// retired -- its event is a miss record whatever it says, so neither the event (56 B) nor anything else of it is read.
__device__ __forceinline__ bool reflect_one(const PolyRec* polys, RayRec* rays, const XEventRec* ev, int32_t* excl_out, int64_t i, bool marks_valid)
{
    if (marks_valid && excl_out[i] == -2) return false;
    const XEventRec e = ev[i];
    if (!e.hit) {
        excl_out[i] = -2;
        return false;
    }
    const RayRec r = rays[i];
    const PolyRec& p = polys[e.poly_id];
    const double dn = dot3(r.dx, r.dy, r.dz, p.n[0], p.n[1], p.n[2]);
    const double k = 2.0 * dn;
    RayRec o;
    o.x = e.x; o.y = e.y; o.z = e.z;
    o.dx = r.dx - k * p.n[0];
    o.dy = r.dy - k * p.n[1];
    o.dz = r.dz - k * p.n[2];
    rays[i] = o;
    excl_out[i] = e.poly_id;
    return true;
}
// block_live (nullable; the bounce loop's launch-per-cast path on a grid the pool kernel serves, launch.cpp): one byte per block of 64
// consecutive rays (a wave of this kernel), 1 when any ray of the block lives on.  hare_live_blocks turns the bytes into the list of live
// blocks the pool kernel's next cast walks instead of the ray array: blocks in which every ray is retired cost it nothing -- open scenes,
// round 6.
__global__ __launch_bounds__(256) void hare_reflect(const PolyRec* polys, RayRec* rays, const XEventRec* ev,
                                                    int32_t* excl_out, int64_t n, int32_t marks_valid, unsigned char* block_live)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = i < n && reflect_one(polys, rays, ev, excl_out, i, marks_valid != 0);
    if (block_live) {
        const unsigned long long bm = __ballot(live);
        const int64_t blk = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
        if ((threadIdx.x & 63) == 0 && blk * 64 < n) block_live[blk] = bm != 0ull ? 1 : 0;
    }
}
// One workgroup, behind hare_reflect on the stream: 5 us for four million rays.  What was tried instead, to save this launch where nothing
// dies (a closed room pays ~1 % for it): the list built by hare_reflect's last workgroup behind a done-count (a fence + an atomic on one word
// per workgroup: 1.1 ms per million rays -- an agent-scope release writes back an XCD's L2); the reflection itself as <= 512 persistent
// workgroups that exchange their counts (0.8 ms as a chain, and with the counts read all at once still slower than the plain kernel's many small
// workgroups on LIVE rays: C5 -4 %); the list built by the pool kernel's first workgroup while the others wait (the same fences: a cast over
// retired rays 119 us instead of 59) -- profiles/r06_experiments/README.md.
__global__ __launch_bounds__(1024) void hare_live_blocks(const unsigned char* block_live, uint32_t nblk, uint32_t* list, uint32_t* count)
{
    __shared__ uint32_t wsum[16];
    build_live_list<1024>(block_live, nblk, list, count, wsum);
}

// ---- dead-ray compaction of the bounce loop (hare_bounce_batch; SURVEY.md 7.1 step 9, 8(a) A9 "reflect_compact") ----
// In an open scene most rays leave after a few bounces; instead of streaming the retired rays' records through every later
// cast, the loop packs the survivors: (1) live rays counted per tile of 2048 events, (2) the tile counts scanned by one
// workgroup, (3) reflection written to the packed position.  The packing is STABLE (survivors keep their relative order, so
// the burst's locality survives and so does run-to-run determinism); `idx` carries each packed ray's position in the caller's
// arrays for the way back (hare_events_expand).
constexpr int kCompactTile = 2048;      // events per workgroup: 8 passes of 256 lanes

__global__ __launch_bounds__(256) void hare_live_count(const XEventRec* ev, long long m, uint32_t* tile_counts)
{
    __shared__ uint32_t ws[4];
    const long long base = (long long)blockIdx.x * kCompactTile;
    uint32_t c = 0;
#pragma unroll
    for (int k = 0; k < kCompactTile / 256; ++k) {
        const long long i = base + k * 256 + threadIdx.x;
        if (i < m && ev[i].hit != 0) ++c;
    }
    const unsigned long long s = wave_sum_u32(c);
    if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = (uint32_t)s;
    __syncthreads();
    if (threadIdx.x == 0) tile_counts[blockIdx.x] = ws[0] + ws[1] + ws[2] + ws[3];
}

// exclusive scan of the tile counts in place, by ONE workgroup (a million rays are 512 tiles); total[0] = the sum
__global__ __launch_bounds__(1024) void hare_scan_tiles(uint32_t* counts, long long nt, uint32_t* total)
{
    __shared__ uint32_t wsum[16];
    __shared__ uint32_t carry;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (long long base = 0; base < nt; base += 1024) {
        const long long i = base + threadIdx.x;
        const uint32_t v = i < nt ? counts[i] : 0u;
        uint32_t inc = v;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t t = __shfl_up(inc, off, 64);
            if (lane >= off) inc += t;
        }
        if (lane == 63) wsum[wid] = inc;
        __syncthreads();
        uint32_t woff = carry;
        for (int w = 0; w < wid; ++w) woff += wsum[w];
        if (i < nt) counts[i] = woff + inc - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry = woff + inc;
        __syncthreads();
    }
    if (threadIdx.x == 0) total[0] = carry;
}

// hare_reflect for the survivors only, written to their packed positions; idx_in null = the rays are still in caller order
__global__ __launch_bounds__(256) void hare_reflect_compact(const PolyRec* polys, const RayRec* rays_in, const XEventRec* ev,
                                                            const int32_t* idx_in, const uint32_t* tile_offsets, long long m,
                                                            RayRec* rays_out, int32_t* excl_out, int32_t* idx_out)
{
    __shared__ uint32_t wc[4];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const long long base = (long long)blockIdx.x * kCompactTile;
    uint32_t run = tile_offsets[blockIdx.x];
#pragma unroll 1
    for (int k = 0; k < kCompactTile / 256; ++k) {
        const long long i = base + k * 256 + threadIdx.x;
        XEventRec e;
        e.hit = 0;
        if (i < m) e = ev[i];
        const bool live = i < m && e.hit != 0;
        const unsigned long long bm = __ballot(live);
        if (lane == 0) wc[wid] = (uint32_t)__popcll(bm);
        __syncthreads();
        uint32_t off = run + rank_below(bm);
        for (int w = 0; w < wid; ++w) off += wc[w];
        run += wc[0] + wc[1] + wc[2] + wc[3];
        __syncthreads();
        if (live) {
            const RayRec r = rays_in[i];
            const PolyRec& p = polys[e.poly_id];
            const double dn = dot3(r.dx, r.dy, r.dz, p.n[0], p.n[1], p.n[2]);
            const double kk = 2.0 * dn;
            RayRec o;
            o.x = e.x; o.y = e.y; o.z = e.z;
            o.dx = r.dx - kk * p.n[0];
            o.dy = r.dy - kk * p.n[1];
            o.dz = r.dz - kk * p.n[2];
            rays_out[off] = o;
            excl_out[off] = e.poly_id;
            idx_out[off] = idx_in ? idx_in[i] : (int32_t)i;
        }
    }
}

// The way back: events of the packed rays to their positions in the caller's order; every other position is the miss record
__global__ __launch_bounds__(256) void hare_events_fill_miss(XEventRec* full, long long n)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    XEventRec e;
    set_miss(e);
    full[i] = e;
}
__global__ __launch_bounds__(256) void hare_events_expand(const XEventRec* ev, const int32_t* idx, long long m, XEventRec* full)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    full[idx[i]] = ev[i];
}

// Slim result records of the host-buffer calls (HARE_SHOOT_SLIM_EVENTS, include/hare_hip.h): what an X_Event holds that the caller
// cannot recompute.  X_Point is o + d * t by the reference's own expression (Hare_Geometry_Polygons.cs:652), so it need not cross
// the host link; the voxel path returns u = v = 0 (Voxel_Grid.cs:696-697).  uv = 0: 16 bytes {t, poly_id, hit}; uv = 1: 32 bytes
// {t, u, v, poly_id, hit}.  hit = 2 marks a voxel hit on a ray whose origin AABB.Intersect moved: t is then tmin, measured from
// the moved origin (hare_expand_events re-derives the move, t = tmin + t_start and X_Point = o' + d * tmin bit for bit).
__global__ __launch_bounds__(256) void hare_events_pack_slim(const XEventRec* ev, long long n, int uv, unsigned char* slim)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const XEventRec e = ev[i];
    if (uv) {
        double* q = reinterpret_cast<double*>(slim + i * 32);
        q[0] = e.t; q[1] = e.u; q[2] = e.v;
        q[3] = __hiloint2double(e.hit, e.poly_id);
    } else {
        const bool moved = e.hit != 0 && e.u != 0.0;
        double2 r;
        r.x = moved ? e.u : e.t;
        r.y = __hiloint2double(moved ? 2 : e.hit, e.poly_id);
        *reinterpret_cast<double2*>(slim + i * 16) = r;
    }
}

// A9 occlusion predicate (harness-defined): a ray is occluded when its CLOSEST hit -- the
// X_Event the shoot kernel just wrote, so the closest-hit oracle pins it -- lies before t_max.  t_max null: any hit.
// per-cast counter blocks of a bounce loop summed into the caller's totals (the launch-per-cast path of hare_bounce_device).  Both are
// ACCUMULATED by contract, so the call subtracts the blocks as they stand before the loop (sign = -1) and adds them after it (sign = +1):
// the totals move by what the loop added (arithmetic modulo 2^64)
__global__ __launch_bounds__(64) void hare_counters_sum(const unsigned long long* per_cast, int casts, unsigned long long* total, int sign)
{
    const int w = threadIdx.x;
    if (w >= CTR_WORDS) return;
    unsigned long long s = 0;
    for (int c = 0; c < casts; ++c) s += per_cast[(size_t)c * CTR_WORDS + w];
    if (s) atomicAdd(&total[w], sign < 0 ? (0ull - s) : s);
}

__global__ __launch_bounds__(256) void hare_occlusion(const XEventRec* ev, const double* tmax, int32_t* occluded, int64_t n)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double t = ev[i].t;
    const int hit = ev[i].hit;
    occluded[i] = (hit != 0 && (tmax == nullptr || t < tmax[i])) ? 1 : 0;
}

}  // extern "C"

#include "voxel_pool.hip"
#include "octree_pool.hip"
#include "octree_group.hip"
// K2g: Octree.Shoot, eight lanes per ray (octree_group.hip) -- the production kernel of the octree path
extern "C" __global__ __launch_bounds__(256, HARE_K2G_WAVES_PER_EU) void hare_octree_group(hare::OctreeArgs g, hare::ShootIO io) { octree_group_body<false>(g, io); }
// K2g as the tail of K2p: the rays K2p's waves were still walking when the tickets ran dry, continued from their walk state
extern "C" __global__ __launch_bounds__(256, HARE_K2G_WAVES_PER_EU) void hare_octree_group_tail(hare::OctreeArgs g, hare::ShootIO io) { octree_group_body<true>(g, io); }
#include "kdtree_dense.hip"
#include "order_kernels.hip"
#include "build_kernels.hip"
