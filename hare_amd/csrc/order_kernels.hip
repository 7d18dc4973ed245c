// order_kernels.hip -- the ORDER in which the pool kernel K1q takes a batch's rays (included by kernels.hip; round 5).
//
// Measured (profiles/r05_experiments/c2_heavy_first.log, window_sort_*.log): K1q runs 5 - 7 % faster when the rays a wave's pool holds at
// one time cost about the same -- its walk tasks run until few of their lanes still walk, so a pool of rays with similar walk lengths
// wastes fewer lane-steps -- as long as the batch's own locality is kept (a random permutation costs 11 - 13 %, a global sort by cost
// gains 1 - 2 %: heavy rays bunched together thrash what the burst's order leaves alone).  Both at once: the rays of every WINDOW of
// kOrderWindow consecutive rays are taken in the order of an estimate of their walk length.  Rays, events and exclusions stay where
// the caller has them (ShootIO::order is read once per ray, at its set-up); every ray's X_Event is computed from the ray alone, so the
// order cannot change a result.
//
// The estimate: voxels from the origin (or the entry into the grid's box) to the exit, sum_a |d_a| * len / VoxelDims_a -- the set-up
// arithmetic of Voxel_Grid.cs:567-632 in FP32 (it only orders) -- counted into kOrderBins bins over [0, 3 ct]; a counting sort in LDS
// per window: histogram, scan, placement by LDS atomics.  The placement inside a bin is whatever order the atomics return: the order of
// rays of (almost) equal estimate is not defined and does not matter.
// (kOrderWindow, kOrderBins, kOrderThreads: hare_device.h)
extern "C" __global__ __launch_bounds__(1024) void hare_cost_order(const hare::RayRec* rays, long long n, float ox0, float oy0, float oz0, float ox1, float oy1,
                                                                   float oz1, float ivx, float ivy, float ivz, float bins_per_voxel, uint32_t* order)
{
    using namespace hare;
    __shared__ unsigned hist[kOrderBins];
    __shared__ unsigned wsum[kOrderBins / 64];
    const long long base = (long long)blockIdx.x * kOrderWindow;
    const int m = (int)(n - base < (long long)kOrderWindow ? n - base : (long long)kOrderWindow);
    const int tid = threadIdx.x;
    if (tid < kOrderBins) hist[tid] = 0u;
    __syncthreads();
    constexpr int PER = kOrderWindow / kOrderThreads;
    unsigned key[PER];
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        const int idx = tid + j * kOrderThreads;
        key[j] = 0u;
        if (idx < m) {
            const RayRec r = rays[base + idx];
            const float o[3] = {(float)r.x, (float)r.y, (float)r.z}, d[3] = {(float)r.dx, (float)r.dy, (float)r.dz};
            const float lo[3] = {ox0, oy0, oz0}, hi[3] = {ox1, oy1, oz1}, iv[3] = {ivx, ivy, ivz};
            float t_in = 0.0f, t_out = 3.0e38f, w = 0.0f;
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                const float inv = 1.0f / d[a];
                const float t0 = (lo[a] - o[a]) * inv, t1 = (hi[a] - o[a]) * inv;
                t_in = fmaxf(t_in, fminf(t0, t1));          // fmaxf / fminf drop a NaN (0 * inf): that slab says nothing
                t_out = fminf(t_out, fmaxf(t0, t1));
                w += fabsf(d[a]) * iv[a];                   // voxels per unit of the ray parameter
            }
            const float cells = (t_out > t_in) ? (t_out - t_in) * w : 0.0f;
            const float b = cells * bins_per_voxel;
            key[j] = (b >= 0.0f && b < (float)(kOrderBins - 1)) ? (unsigned)b : (b >= (float)(kOrderBins - 1) ? (unsigned)(kOrderBins - 1) : 0u);   // NaN -> 0
            atomicAdd(&hist[key[j]], 1u);
        }
    }
    __syncthreads();
    // exclusive scan of the histogram: eight waves scan 64 bins each, then add the sums of the waves in front
    unsigned mine = 0, incl = 0;
    if (tid < kOrderBins) {
        mine = hist[tid];
        incl = mine;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const unsigned v = __shfl_up(incl, off, 64);
            if ((tid & 63) >= off) incl += v;
        }
        if ((tid & 63) == 63) wsum[tid >> 6] = incl;
    }
    __syncthreads();
    if (tid < kOrderBins) {
        unsigned before = 0;
        for (int w = 0; w < (tid >> 6); ++w) before += wsum[w];
        hist[tid] = before + incl - mine;                   // where the bin starts inside the window
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        const int idx = tid + j * kOrderThreads;
        if (idx < m) {
            const unsigned pos = atomicAdd(&hist[key[j]], 1u);
            order[base + pos] = (uint32_t)(base + idx);
        }
    }
}
