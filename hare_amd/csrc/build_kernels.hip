// build_kernels.hip -- Voxel_Grid construction on the GPU.
//
// Reference: Voxel_Grid ctor + Fill_Voxels (Voxel_Grid.cs:48-121, :273-304) test every voxel against every
// polygon with AABB.PolyBoxOverlap (AABB_Tri_Int.cs:165-260); the hierarchical ctor (:128-254) tests a
// child voxel against its parent's list.  Membership fixes the ORDER in which Shoot tests candidates
// and therefore who wins an exact-t tie, so the lists must come out identical: same predicate
// (hare_math.h: poly_box_overlap / tri_box_sat, FP64, no contraction), same padded boxes, ascending
// polygon index per voxel.
//
// Fixed grid:   polygon-major  count -> exclusive scan -> fill (atomic cursor) -> per-voxel sort.
// Hierarchical: voxel-major over the parent's (ascending) list: count -> scan -> fill, no sort needed.
// Included at the end of kernels.hip (one code object).
#include <hip/hip_runtime.h>
#include "hare_device.h"

using namespace hare;

namespace {

constexpr uint32_t kBigList = 8192;   // longest list the LDS bitonic sorter takes

__device__ __forceinline__ int load_corners(const BuildArgs& b, int p, double* V)
{
    const PolyRec& r = b.polys[p];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        V[a] = r.v0[a];
        V[3 + a] = r.v1[a];
        V[6 + a] = r.v2[a];
    }
    int nv = 3;
    if (b.quads) {
        const QuadRec& q = b.quads[p];
        nv = q.nverts;
#pragma unroll
        for (int a = 0; a < 3; ++a) V[9 + a] = q.v3[a];
    }
    return nv;
}

__device__ __forceinline__ void cell_box(const BuildArgs& b, int x, int y, int z, double* bmin, double* bmax)
{
    bmin[0] = voxel_lo(x, b.vd[0], b.omin[0]); bmax[0] = voxel_hi(x, b.vd[0], b.omin[0]);
    bmin[1] = voxel_lo(y, b.vd[1], b.omin[1]); bmax[1] = voxel_hi(y, b.vd[1], b.omin[1]);
    bmin[2] = voxel_lo(z, b.vd[2], b.omin[2]); bmax[2] = voxel_hi(z, b.vd[2], b.omin[2]);
}

// conservative voxel range of a polygon: its AABB grown by the 1 mm pad (+10 %) and one voxel each way
__device__ __forceinline__ void cell_range(const BuildArgs& b, const double* V, int nv, int* lo, int* hi)
{
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        double mn = V[a], mx = V[a];
        for (int c = 1; c < nv; ++c) {
            mn = fmin(mn, V[3 * c + a]);
            mx = fmax(mx, V[3 * c + a]);
        }
        double flo = floor((mn - 0.0011 - b.omin[a]) / b.vd[a]) - 1;
        double fhi = floor((mx + 0.0011 - b.omin[a]) / b.vd[a]) + 1;
        if (!(flo >= 0)) flo = 0;
        if (!(fhi <= b.ct - 1)) fhi = b.ct - 1;
        lo[a] = (int)flo;
        hi[a] = (int)fhi;
    }
}

template <bool FILL>
__device__ __forceinline__ void poly_major(const BuildArgs& b, uint32_t* count_or_cursor, const uint32_t* start, int32_t* items)
{
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= b.P) return;
    double V[12];
    const int nv = load_corners(b, p, V);
    int lo[3], hi[3];
    cell_range(b, V, nv, lo, hi);
    const int ct = b.ct;
    for (int x = lo[0]; x <= hi[0]; ++x)
        for (int y = lo[1]; y <= hi[1]; ++y)
            for (int z = lo[2]; z <= hi[2]; ++z) {
                double bmin[3], bmax[3];
                cell_box(b, x, y, z, bmin, bmax);
                if (poly_box_overlap(bmin, bmax, V, nv)) {
                    const uint32_t cell = (uint32_t)((x * ct + y) * ct + z);
                    if (FILL) {
                        // voxels with a very long list are filled in order by hare_vb_fill_big instead
                        if (start[cell + 1] - start[cell] > kBigList) continue;
                        const uint32_t k = atomicAdd(&count_or_cursor[cell], 1u);
                        items[start[cell] + k] = p;
                    } else {
                        atomicAdd(&count_or_cursor[cell], 1u);
                    }
                }
            }
}

// Hierarchical level: one thread per child voxel walks its parent's list in order.
template <bool FILL>
__device__ __forceinline__ void voxel_major(const BuildArgs& b, const uint32_t* pstart, const int32_t* pitems,
                                            uint32_t* count, const uint32_t* start, int32_t* items)
{
    const int ct = b.ct, pct = ct >> 1;
    const long long c = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= (long long)ct * ct * ct) return;
    const int z = (int)(c % ct), y = (int)((c / ct) % ct), x = (int)(c / ((long long)ct * ct));
    double bmin[3], bmax[3];
    cell_box(b, x, y, z, bmin, bmax);
    const long long par = ((long long)(x >> 1) * pct + (y >> 1)) * pct + (z >> 1);
    uint32_t n = 0;
    const uint32_t base = FILL ? start[c] : 0u;
    for (uint32_t q = pstart[par]; q < pstart[par + 1]; ++q) {
        const int p = pitems[q];
        double V[12];
        const int nv = load_corners(b, p, V);
        if (poly_box_overlap(bmin, bmax, V, nv)) {
            if (FILL) items[base + n] = p;
            ++n;
        }
    }
    if (!FILL) count[c] = n;
}

// ---- Octree.BuildOctree ("Octree - alt.cs":91-138) -------------------------------------------------
// A splitting node hands each polygon of its list to every child whose loose box it overlaps, in list
// order.  One workgroup per task (child box x <= 8192-entry segment of the parent's list): 256 entries
// at a time, exact PolyBoxOverlap, survivors appended in order with a block-wide prefix sum.
template <bool FILL>
__device__ __forceinline__ void octree_task(const BuildArgs& b, const OctTask* tasks, const int32_t* pitems, uint32_t* counts,
                                            int32_t* oitems)
{
    __shared__ uint32_t wcount[4];
    __shared__ uint32_t base_s;
    const OctTask t = tasks[blockIdx.x];
    const double bmin[3] = {t.bmin[0], t.bmin[1], t.bmin[2]}, bmax[3] = {t.bmax[0], t.bmax[1], t.bmax[2]};
    if (threadIdx.x == 0) base_s = FILL ? t.ostart : 0u;
    __syncthreads();
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    for (uint32_t q0 = 0; q0 < t.pcount; q0 += 256) {
        const uint32_t q = q0 + threadIdx.x;
        bool in = false;
        int p = -1;
        if (q < t.pcount) {
            p = pitems[t.pstart + q];
            double V[12];
            const int nv = load_corners(b, p, V);
            in = poly_box_overlap(bmin, bmax, V, nv);
        }
        const unsigned long long m = __ballot(in);
        if (lane == 0) wcount[wid] = (uint32_t)__popcll(m);
        __syncthreads();
        if (FILL) {
            uint32_t off = base_s;
            for (int w = 0; w < wid; ++w) off += wcount[w];
            if (in) oitems[off + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = p;
        }
        __syncthreads();
        if (threadIdx.x == 0) base_s += wcount[0] + wcount[1] + wcount[2] + wcount[3];
        __syncthreads();
    }
    if (!FILL && threadIdx.x == 0) counts[blockIdx.x] = base_s;
}

}  // namespace

extern "C" {

__global__ __launch_bounds__(256) void hare_vb_count(BuildArgs b, uint32_t* count) { poly_major<false>(b, count, nullptr, nullptr); }
__global__ __launch_bounds__(256) void hare_vb_fill(BuildArgs b, uint32_t* cursor, const uint32_t* start, int32_t* items)
{
    poly_major<true>(b, cursor, start, items);
}
__global__ __launch_bounds__(256) void hare_vb_level_count(BuildArgs b, const uint32_t* pstart, const int32_t* pitems, uint32_t* count)
{
    voxel_major<false>(b, pstart, pitems, count, nullptr, nullptr);
}
__global__ __launch_bounds__(256) void hare_vb_level_fill(BuildArgs b, const uint32_t* pstart, const int32_t* pitems,
                                                          const uint32_t* start, int32_t* items)
{
    voxel_major<true>(b, pstart, pitems, nullptr, start, items);
}

// ---- exclusive scan of uint32, 2048 elements per block (256 threads x 8) ---------------------------
__global__ __launch_bounds__(256) void hare_scan_block(const uint32_t* in, uint32_t* out, uint32_t* block_sums, long long n)
{
    __shared__ uint32_t wsum[4];
    const long long base = (long long)blockIdx.x * 2048 + (long long)threadIdx.x * 8;
    uint32_t v[8], s = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        v[k] = (base + k < n) ? in[base + k] : 0u;
        s += v[k];
    }
    // inclusive scan of the per-thread sums inside the wave, then across the 4 waves
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    uint32_t inc = s;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t t = __shfl_up(inc, off, 64);
        if (lane >= off) inc += t;
    }
    if (lane == 63) wsum[wid] = inc;
    __syncthreads();
    uint32_t woff = 0;
    for (int w = 0; w < wid; ++w) woff += wsum[w];
    uint32_t run = woff + inc - s;   // exclusive prefix of this thread
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        if (base + k < n) out[base + k] = run;
        run += v[k];
    }
    if (threadIdx.x == 255 && block_sums) block_sums[blockIdx.x] = run;
}

__global__ __launch_bounds__(256) void hare_scan_add(uint32_t* out, const uint32_t* block_offsets, long long n)
{
    const long long base = (long long)blockIdx.x * 2048 + (long long)threadIdx.x * 8;
    const uint32_t off = block_offsets[blockIdx.x];
#pragma unroll
    for (int k = 0; k < 8; ++k)
        if (base + k < n) out[base + k] += off;
}

// ---- per-voxel ascending sort of the filled lists ---------------------------------------------------
// small lists: one thread each, insertion sort in place; longer ones are queued for the block sorter
__global__ __launch_bounds__(256) void hare_vb_sort_small(const uint32_t* start, int32_t* items, long long ncell,
                                                          uint32_t* big_cells, uint32_t* big_count, uint32_t small_max)
{
    const long long c = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= ncell) return;
    const uint32_t s = start[c], n = start[c + 1] - s;
    if (n <= 1 || n > kBigList) return;   // > kBigList: already in order (hare_vb_fill_big)
    if (n > small_max) {
        big_cells[atomicAdd(big_count, 1u)] = (uint32_t)c;
        return;
    }
    int32_t* a = items + s;
    for (uint32_t i = 1; i < n; ++i) {
        const int32_t key = a[i];
        uint32_t j = i;
        while (j > 0 && a[j - 1] > key) {
            a[j] = a[j - 1];
            --j;
        }
        a[j] = key;
    }
}

// one block per queued voxel: bitonic sort in LDS (up to 8192 entries); longer lists set *overflow
__global__ __launch_bounds__(256) void hare_vb_sort_block(const uint32_t* start, int32_t* items, const uint32_t* big_cells,
                                                          uint32_t* overflow)
{
    __shared__ int32_t buf[8192];
    const uint32_t c = big_cells[blockIdx.x];
    const uint32_t s = start[c], n = start[c + 1] - s;
    if (n > 8192) {
        if (threadIdx.x == 0) atomicAdd(overflow, 1u);
        return;
    }
    uint32_t m = 1;
    while (m < n) m <<= 1;
    for (uint32_t i = threadIdx.x; i < m; i += blockDim.x) buf[i] = i < n ? items[s + i] : 0x7FFFFFFF;
    __syncthreads();
    for (uint32_t k = 2; k <= m; k <<= 1)
        for (uint32_t j = k >> 1; j > 0; j >>= 1) {
            for (uint32_t i = threadIdx.x; i < m; i += blockDim.x) {
                const uint32_t l = i ^ j;
                if (l > i) {
                    const bool up = (i & k) == 0;
                    const int32_t x = buf[i], y = buf[l];
                    if ((x > y) == up) {
                        buf[i] = y;
                        buf[l] = x;
                    }
                }
            }
            __syncthreads();
        }
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) items[s + i] = buf[i];
}

// voxels whose list exceeds kBigList: queue them (after the scan, before the fill)
__global__ __launch_bounds__(256) void hare_vb_find_big(const uint32_t* start, long long ncell, uint32_t* big_cells, uint32_t* big_count)
{
    const long long c = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= ncell) return;
    if (start[c + 1] - start[c] > kBigList) big_cells[atomicAdd(big_count, 1u)] = (uint32_t)c;
}

// one block per queued voxel: walk ALL polygons in index order, 256 at a time, and append the overlapping
// ones with a block-wide prefix sum -- an ordered (ascending) fill that needs no sort
__global__ __launch_bounds__(256) void hare_vb_fill_big(BuildArgs b, const uint32_t* start, int32_t* items, const uint32_t* big_cells)
{
    __shared__ uint32_t wcount[4];
    __shared__ uint32_t base_s;
    const uint32_t c = big_cells[blockIdx.x];
    const int ct = b.ct;
    const int z = (int)(c % ct), y = (int)((c / ct) % ct), x = (int)(c / ((uint32_t)ct * ct));
    double bmin[3], bmax[3];
    cell_box(b, x, y, z, bmin, bmax);
    if (threadIdx.x == 0) base_s = start[c];
    __syncthreads();
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    for (int p0 = 0; p0 < b.P; p0 += 256) {
        const int p = p0 + threadIdx.x;
        bool in = false;
        if (p < b.P) {
            double V[12];
            const int nv = load_corners(b, p, V);
            in = poly_box_overlap(bmin, bmax, V, nv);
        }
        const unsigned long long m = __ballot(in);
        if (lane == 0) wcount[wid] = (uint32_t)__popcll(m);
        __syncthreads();
        uint32_t off = base_s;
        for (int w = 0; w < wid; ++w) off += wcount[w];
        if (in) items[off + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = p;
        __syncthreads();
        if (threadIdx.x == 0) base_s += wcount[0] + wcount[1] + wcount[2] + wcount[3];
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void hare_ob_count(BuildArgs b, const OctTask* tasks, const int32_t* pitems, uint32_t* counts)
{
    octree_task<false>(b, tasks, pitems, counts, nullptr);
}
__global__ __launch_bounds__(256) void hare_ob_fill(BuildArgs b, const OctTask* tasks, const int32_t* pitems, int32_t* oitems)
{
    octree_task<true>(b, tasks, pitems, nullptr, oitems);
}

// The block-level occupancy of the pool kernel's exact multi-voxel skip (scene option "voxel_skip", voxel_pool.hip): one bit per ALIGNED block of
// 4 x 4 x 4 voxels, set when any voxel of the block has a list.  nb = ceil(ct / 4) blocks per axis; the words are zeroed by the host first.
__global__ __launch_bounds__(256) void hare_block_occ(const CellRec* cells, long long ncell, int ct, int nb, uint32_t* bocc)
{
    const long long c = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= ncell || cells[c].count == 0) return;
    const int z = (int)(c % ct), y = (int)((c / ct) % ct), x = (int)(c / ((long long)ct * ct));
    const int b = ((x >> 2) * nb + (y >> 2)) * nb + (z >> 2);
    atomicOr(&bocc[b >> 5], 1u << (b & 31));
}

// The voxels' TIGHT boxes (device_scene.cpp: upload_cell_boxes; used by K1q, voxel_pool.hip): per voxel the bounding box of ALL polygons its list
// holds -- whole polygons, not clipped to the voxel: Voxel_Grid.Shoot records a hit wherever it lies on the polygon (Voxel_Grid.cs:691-699)
// -- grown by `delta` and rounded outwards to floats.  8 floats per voxel: lo xyz, hi xyz, two spare; an empty voxel gets an empty box.
__global__ __launch_bounds__(256) void hare_cell_boxes(const CellRec* cells, const int32_t* items, const PolyRec* polys, const QuadRec* quads,
                                                       long long ncell, double delta, float* out)
{
    const long long c = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= ncell) return;
    const CellRec cr = cells[c];
    const double inf = __builtin_huge_val();
    double lo[3] = {inf, inf, inf}, hi[3] = {-inf, -inf, -inf};
    for (uint32_t q = 0; q < cr.count; ++q) {
        const int i = items[cr.start + q];
        const PolyRec& p = polys[i];
        auto corner = [&](const double* v) {
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                const double x = v[a];
                if (!(x == x)) { lo[a] = -inf; hi[a] = inf; }            // a NaN corner: the box is everything
                else { lo[a] = x < lo[a] ? x : lo[a]; hi[a] = x > hi[a] ? x : hi[a]; }
            }
        };
        corner(p.v0); corner(p.v1); corner(p.v2);
        if (quads && quads[i].nverts == 4) corner(quads[i].v3);
    }
    float* o = out + 8 * c;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        o[a] = __double2float_rd(lo[a] - delta);
        o[3 + a] = __double2float_ru(hi[a] + delta);
    }
    o[6] = 0.0f; o[7] = 0.0f;
}


// cell records (start, count, first two entries inlined) + occupancy bitmap (zeroed beforehand)
__global__ __launch_bounds__(256) void hare_vb_finalize(const uint32_t* start, const int32_t* items, CellRec* cells,
                                                        uint32_t* occ, long long ncell, unsigned long long* stats,
                                                        int ct, int occ_shift, int occ_cd)
{
    const long long c = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t n = 0;
    if (c < ncell) {
        const uint32_t s = start[c];
        n = start[c + 1] - s;
        CellRec r;
        r.start = s;
        r.count = n;
        r.i0 = n > 0 ? items[s] : -1;
        r.i1 = n > 1 ? items[s + 1] : -1;
        cells[c] = r;
        if (n) {   // one bit per block of (2^occ_shift)^3 voxels (scene.h: occ_layout)
            const int z = (int)(c % ct), y = (int)((c / ct) % ct), x = (int)(c / ((long long)ct * ct));
            const uint32_t b = (uint32_t)(((x >> occ_shift) * occ_cd + (y >> occ_shift)) * occ_cd + (z >> occ_shift));
            atomicOr(&occ[b >> 5], 1u << (b & 31));
        }
    }
    // stats[0] += sum of counts over non-empty voxels, stats[1] += number of non-empty voxels (Voxel_Grid.cs:249-252)
    const unsigned long long ne = __ballot(n > 0);
    unsigned long long sum = n;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) sum += __shfl_down(sum, off, 64);
    if ((threadIdx.x & 63) == 0 && stats && ne) {
        atomicAdd(&stats[0], sum);
        atomicAdd(&stats[1], (unsigned long long)__popcll(ne));
    }
}

}  // extern "C"
