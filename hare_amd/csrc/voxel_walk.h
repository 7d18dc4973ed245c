// voxel_walk.h -- the DDA step loop of K1q's walk phases, written by hand for gfx950 (round 6).
//
// What it computes is Voxel_Grid.cs:713-759, step for step, exactly as HARE_K1Q_STEP() in voxel_pool.hip states it:
//     cxy = tMaxX < tMaxY, cxz = tMaxX < tMaxZ, cyz = tMaxY < tMaxZ
//     x steps iff cxy & cxz;  y steps iff !cxy & cyz;  z steps otherwise        (NaN compares false: z steps)
//     the stepping axis:  index += +-1,  tMax = tMax + tDelta                    (one IEEE add, no contraction)
// then "left the grid" (any index outside [0, ct)) and the voxel's occupancy bit from the LDS bitmap; a lane stops at the first
// occupied voxel or outside the grid, with the index and the three tMax bit patterns the reference has at that point.
//
// Why by hand: the compiler's version of this loop is 40 vector + 45 scalar instructions per step (selects for every
// component of every axis, v_cndmask pairs for the doubles, a structured-control-flow mask dance around each exit); the walk is a
// third (hall) to a half (cathedral) of K1q's vector instructions (profiles/r05_experiments/k1q_round_stats_final.log).  Here the
// three per-axis updates run under the axis' own EXEC mask -- an add that is masked off costs nothing extra, where a select costs an
// instruction per 32 bits -- and a lane that has stopped simply leaves EXEC: 17 vector instructions per step on a per-voxel bitmap,
// 20 on a per-block one, ~14 scalar.  Same arithmetic, same order, same bits (tests: walk on / off / oracle).
#pragma once
#include <cstdint>

namespace hare_walk {

// Steps every lane with `walking` set until it stands in an occupied voxel or outside the grid, at most `max_steps` steps; the
// loop also ends once fewer than `min_lanes` lanes still walk (never before the first step).  On return `walking` is true for
// lanes that have not stopped (the task's step budget ran out).  `lds_bitmap` = LDS byte address of the occupancy bitmap
// (scene.h: occ_layout; COARSE: one bit per block of (2^occ_shift)^3 voxels, occ_cd blocks per axis).  COUNT: `taken` += voxels
// walked into (inside the grid) by this lane; `iters` += executions of the step (wave-uniform).  SKIP (the wide walk of the drain): a lane
// passes `skip` occupied voxels and stops at the next one.  Must be called in wave-uniform control flow.
template <bool COARSE, bool COUNT, bool SKIP = false>
__device__ __forceinline__ void walk_steps(double& tMaxX, double& tMaxY, double& tMaxZ, const double tDeltaX, const double tDeltaY,
                                           const double tDeltaZ, int& X, int& Y, int& Z, const int dx1, const int dy1, const int dz1,
                                           bool& walking, const unsigned ct, const unsigned max_steps, const unsigned min_lanes,
                                           const unsigned lds_bitmap, const unsigned occ_shift, const unsigned occ_cd, unsigned& taken, unsigned& iters, unsigned skip = 0u)
{
    unsigned long long wm = __ballot(walking);
    if (wm == 0ull) return;
    // wave-uniform operands, in scalar registers whatever the compiler thinks of their uniformity
    const unsigned s_ct = (unsigned)__builtin_amdgcn_readfirstlane((int)ct), s_steps = (unsigned)__builtin_amdgcn_readfirstlane((int)max_steps);
    const unsigned s_min = (unsigned)__builtin_amdgcn_readfirstlane((int)min_lanes), s_base = (unsigned)__builtin_amdgcn_readfirstlane((int)lds_bitmap);
    const unsigned s_sh = (unsigned)__builtin_amdgcn_readfirstlane((int)occ_shift), s_cd = (unsigned)__builtin_amdgcn_readfirstlane((int)occ_cd);
    unsigned long long sv, ma, mb, mw;
    unsigned t0, t1, k = 0u, cnt, still;
    unsigned nt = taken;
    // every read-write operand is early-clobber ("+&"): the loop writes them while the plain inputs are still live, and without the
    // mark the register allocator may give an input that happens to hold the same VALUE (a zero step counter and a zero LDS
    // address, say) the same register
#define HARE_WALK_HEAD                                                                                          \
        "s_mov_b64 %[sv], exec\n\t"                                                                             \
        "s_mov_b64 exec, %[wm]\n"                                                                               \
        "1:\n\t"                                                                                                \
        "s_add_u32 %[k], %[k], 1\n\t"                                                                           \
        "v_cmp_lt_f64_e64 %[ma], %[tx], %[ty]\n\t"                                                              \
        "v_cmp_lt_f64_e64 %[mb], %[tx], %[tz]\n\t"                                                              \
        "v_cmp_lt_f64_e32 vcc, %[ty], %[tz]\n\t"                                                                \
        "s_mov_b64 %[mw], exec\n\t"                                                                             \
        "s_and_b64 %[mb], %[ma], %[mb]\n\t"          /* x steps */                                             \
        "s_andn2_b64 vcc, vcc, %[ma]\n\t"            /* y steps */                                             \
        "s_or_b64 %[ma], %[mb], vcc\n\t"                                                                        \
        "s_andn2_b64 %[ma], %[mw], %[ma]\n\t"        /* z steps: every other walking lane */                   \
        "s_mov_b64 exec, %[mb]\n\t"                                                                             \
        "v_add_f64 %[tx], %[tx], %[dtx]\n\t"                                                                    \
        "v_add_u32_e32 %[X], %[X], %[dx1]\n\t"                                                                  \
        "s_mov_b64 exec, vcc\n\t"                                                                               \
        "v_add_f64 %[ty], %[ty], %[dty]\n\t"                                                                    \
        "v_add_u32_e32 %[Y], %[Y], %[dy1]\n\t"                                                                  \
        "s_mov_b64 exec, %[ma]\n\t"                                                                             \
        "v_add_f64 %[tz], %[tz], %[dtz]\n\t"                                                                    \
        "v_add_u32_e32 %[Z], %[Z], %[dz1]\n\t"                                                                  \
        "s_mov_b64 exec, %[mw]\n\t"                                                                             \
        "v_max3_u32 %[t0], %[X], %[Y], %[Z]\n\t"     /* an index of -1 is 0xFFFFFFFF */                        \
        "v_cmp_gt_u32_e32 vcc, %[ct], %[t0]\n\t"                                                                \
        "s_and_b64 exec, exec, vcc\n\t"              /* lanes that left the grid stop */                       \
        "s_cbranch_execz 2f\n\t"
#define HARE_WALK_COUNT "v_add_u32_e32 %[nt], 1, %[nt]\n\t"
#define HARE_WALK_BIT_FINE                                                                                      \
        "v_mad_u32_u24 %[t0], %[X], %[ct], %[Y]\n\t"                                                            \
        "v_mad_u32_u24 %[t0], %[t0], %[ct], %[Z]\n\t"
#define HARE_WALK_BIT_COARSE                                                                                    \
        "v_lshrrev_b32_e32 %[t0], %[sh], %[X]\n\t"                                                              \
        "v_lshrrev_b32_e32 %[t1], %[sh], %[Y]\n\t"                                                              \
        "v_mad_u32_u24 %[t0], %[t0], %[cd], %[t1]\n\t"                                                          \
        "v_lshrrev_b32_e32 %[t1], %[sh], %[Z]\n\t"                                                              \
        "v_mad_u32_u24 %[t0], %[t0], %[cd], %[t1]\n\t"
#define HARE_WALK_LOOKUP                                                                                        \
        "v_lshrrev_b32_e32 %[t1], 5, %[t0]\n\t"                                                                 \
        "v_lshl_add_u32 %[t1], %[t1], 2, %[base]\n\t"                                                           \
        "ds_read_b32 %[t1], %[t1]\n\t"                                                                          \
        "s_waitcnt lgkmcnt(0)\n\t"                                                                              \
        "v_bfe_u32 %[t1], %[t1], %[t0], 1\n\t"       /* bit t0 & 31 of the word */
#define HARE_WALK_STOP_FIRST                         /* a lane stops at the first occupied voxel (block) */     \
        "v_cmp_eq_u32_e32 vcc, 0, %[t1]\n\t"                                                                    \
        "s_and_b64 exec, exec, vcc\n\t"                                                                         \
        "s_cbranch_execz 2f\n\t"
#define HARE_WALK_STOP_SKIP                          /* ... at the occupied voxel after `skip` others */        \
        "v_cmp_ne_u32_e32 vcc, 0, %[t1]\n\t"                                                                    \
        "s_and_b64 %[ma], exec, vcc\n\t"             /* lanes in an occupied voxel */                          \
        "s_cbranch_scc0 3f\n\t"                                                                                 \
        "s_mov_b64 %[mw], exec\n\t"                                                                             \
        "s_mov_b64 exec, %[ma]\n\t"                                                                             \
        "v_cmp_eq_u32_e32 vcc, 0, %[skip]\n\t"       /* it is theirs: they stop */                             \
        "v_add_u32_e32 %[skip], -1, %[skip]\n\t"     /* the others have one fewer to pass */                   \
        "s_andn2_b64 exec, %[mw], vcc\n\t"                                                                      \
        "s_cbranch_execz 2f\n"                                                                                  \
        "3:\n\t"
#define HARE_WALK_END                                                                                           \
        "s_cmp_ge_u32 %[k], %[steps]\n\t"                                                                       \
        "s_cbranch_scc1 2f\n\t"                                                                                 \
        "s_bcnt1_i32_b64 %[cnt], exec\n\t"                                                                      \
        "s_cmp_ge_u32 %[cnt], %[wmin]\n\t"                                                                      \
        "s_cbranch_scc1 1b\n"                                                                                   \
        "2:\n\t"                                                                                                \
        "s_mov_b64 %[wm], exec\n\t"                                                                             \
        "s_mov_b64 exec, %[sv]\n\t"                                                                             \
        "v_cndmask_b32_e64 %[still], 0, 1, %[wm]\n\t"
#define HARE_WALK_OPERANDS                                                                                      \
        : [tx] "+&v"(tMaxX), [ty] "+&v"(tMaxY), [tz] "+&v"(tMaxZ), [X] "+&v"(X), [Y] "+&v"(Y), [Z] "+&v"(Z), [nt] "+&v"(nt),               \
          [wm] "+&s"(wm), [k] "+&s"(k), [sv] "=&s"(sv), [ma] "=&s"(ma), [mb] "=&s"(mb), [mw] "=&s"(mw), [cnt] "=&s"(cnt),             \
          [t0] "=&v"(t0), [t1] "=&v"(t1), [still] "=&v"(still), [skip] "+&v"(skip)                                                                    \
        : [dtx] "v"(tDeltaX), [dty] "v"(tDeltaY), [dtz] "v"(tDeltaZ), [dx1] "v"(dx1), [dy1] "v"(dy1), [dz1] "v"(dz1),               \
          [ct] "s"(s_ct), [steps] "s"(s_steps), [wmin] "s"(s_min), [base] "s"(s_base), [sh] "s"(s_sh), [cd] "s"(s_cd)             \
        : "vcc", "scc"
#define HARE_WALK_ASM(COUNT_, BIT_, STOP_) asm volatile(HARE_WALK_HEAD COUNT_ BIT_ HARE_WALK_LOOKUP STOP_ HARE_WALK_END HARE_WALK_OPERANDS)
    if (SKIP) {
        if (COARSE) { if (COUNT) HARE_WALK_ASM(HARE_WALK_COUNT, HARE_WALK_BIT_COARSE, HARE_WALK_STOP_SKIP); else HARE_WALK_ASM("", HARE_WALK_BIT_COARSE, HARE_WALK_STOP_SKIP); }
        else        { if (COUNT) HARE_WALK_ASM(HARE_WALK_COUNT, HARE_WALK_BIT_FINE, HARE_WALK_STOP_SKIP);   else HARE_WALK_ASM("", HARE_WALK_BIT_FINE, HARE_WALK_STOP_SKIP); }
    } else {
        if (COARSE) { if (COUNT) HARE_WALK_ASM(HARE_WALK_COUNT, HARE_WALK_BIT_COARSE, HARE_WALK_STOP_FIRST); else HARE_WALK_ASM("", HARE_WALK_BIT_COARSE, HARE_WALK_STOP_FIRST); }
        else        { if (COUNT) HARE_WALK_ASM(HARE_WALK_COUNT, HARE_WALK_BIT_FINE, HARE_WALK_STOP_FIRST);   else HARE_WALK_ASM("", HARE_WALK_BIT_FINE, HARE_WALK_STOP_FIRST); }
    }
#undef HARE_WALK_ASM
#undef HARE_WALK_LOOKUP
#undef HARE_WALK_STOP_FIRST
#undef HARE_WALK_STOP_SKIP
#undef HARE_WALK_END
#undef HARE_WALK_HEAD
#undef HARE_WALK_COUNT
#undef HARE_WALK_BIT_FINE
#undef HARE_WALK_BIT_COARSE
#undef HARE_WALK_OPERANDS
    walking = still != 0u;
    taken = nt;
    iters += k;
}

}  // namespace hare_walk
