// build_gpu.cpp -- host orchestration of the GPU Voxel_Grid builders (kernels in build_kernels.hip).
// Produces the device-resident grid (cells / items / occupancy bitmap) directly and mirrors the lists
// to the host vectors for introspection (hare_voxel_get_lists).  Product code; nothing from oracle/.
#include <string.h>
#include <algorithm>
#include <vector>

#include "../../include/hare_hip.h"
#include "scene.h"

namespace hare {
namespace {

#define HIP_TRY(expr)                                        \
    do {                                                     \
        hipError_t _e = (expr);                              \
        if (_e != hipSuccess) return hip_fail(H, _e, #expr); \
    } while (0)

struct DevMem {   // frees on scope exit unless released
    const HipApi* H;
    std::vector<void*> ptrs;
    explicit DevMem(const HipApi* h) : H(h) {}
    ~DevMem() { for (void* p : ptrs) if (p) (void)H->Free(p); }
    int alloc(void** out, size_t bytes, bool zero)
    {
        *out = nullptr;
        hipError_t e = H->Malloc(out, bytes ? bytes : 16);
        if (e != hipSuccess) return hip_fail(H, e, "hipMalloc (grid build)");
        ptrs.push_back(*out);
        if (zero) {
            e = H->MemsetAsync(*out, 0, bytes ? bytes : 16, nullptr);
            if (e != hipSuccess) return hip_fail(H, e, "hipMemsetAsync (grid build)");
        }
        return HARE_OK;
    }
    void release(void* p) { for (void*& q : ptrs) if (q == p) q = nullptr; }
};

// exclusive scan of n uint32 (in -> out), recursive over 2048-element blocks
int scan_u32(const HipApi* H, const DeviceModule& M, DevMem& mem, const void* in, void* out, long long n)
{
    const long long nb = (n + 2047) / 2048;
    void* sums = nullptr;
    void* sums_scanned = nullptr;
    if (nb > 1) {
        int rc = mem.alloc(&sums, (size_t)nb * 4, false);
        if (rc) return rc;
        rc = mem.alloc(&sums_scanned, (size_t)nb * 4, false);
        if (rc) return rc;
    }
    {
        void* args[] = {(void*)&in, &out, &sums, &n};
        int rc = launch(H, M.scan_block, (unsigned)nb, 256, 0, nullptr, args);
        if (rc) return rc;
    }
    if (nb > 1) {
        int rc = scan_u32(H, M, mem, sums, sums_scanned, nb);
        if (rc) return rc;
        void* args[] = {&out, &sums_scanned, &n};
        rc = launch(H, M.scan_add, (unsigned)nb, 256, 0, nullptr, args);
        if (rc) return rc;
    }
    return HARE_OK;
}

bool have_build_kernels(const DeviceModule& M)
{
    return M.vb_count && M.vb_fill && M.vb_level_count && M.vb_level_fill && M.scan_block && M.scan_add &&
           M.vb_sort_small && M.vb_sort_block && M.vb_finalize && M.vb_find_big && M.vb_fill_big;
}

void fill_args(BuildArgs& b, const Scene& s, size_t m, const VoxelHost& g)
{
    memset(&b, 0, sizeof b);
    b.polys = (const PolyRec*)s.d_polys[m];
    b.quads = (const QuadRec*)s.d_quads[m];
    b.P = s.topos[m].P;
    b.ct = g.ct;
    for (int a = 0; a < 3; ++a) {
        b.omin[a] = g.omin[a];
        b.vd[a] = g.vd[a];
    }
}

// cells + occupancy from (start, items); takes ownership of nothing, allocates d_cells / d_occ
int finalize_level(const HipApi* H, const DeviceModule& M, DevMem& mem, const void* d_start, const void* d_items, long long ncell,
                   void** d_cells, void** d_occ, int32_t* occ_words, unsigned long long stats_out[2])
{
    *occ_words = (int32_t)((ncell + 31) / 32);
    const size_t occ_bytes = (size_t)((*occ_words + 3) / 4) * 16;
    int rc = mem.alloc(d_cells, (size_t)ncell * sizeof(CellRec), false);
    if (rc) return rc;
    rc = mem.alloc(d_occ, occ_bytes, true);
    if (rc) return rc;
    void* d_stats = nullptr;
    rc = mem.alloc(&d_stats, 16, true);
    if (rc) return rc;
    void* args[] = {(void*)&d_start, (void*)&d_items, d_cells, d_occ, &ncell, &d_stats};
    rc = launch(H, M.vb_finalize, (unsigned)((ncell + 255) / 256), 256, 0, nullptr, args);
    if (rc) return rc;
    HIP_TRY(H->Memcpy(stats_out, d_stats, 16, hipMemcpyDeviceToHost));
    return HARE_OK;
}

void replace_buffer(const HipApi* H, std::vector<void*>& v, size_t m, void* p)
{
    if (v[m]) (void)H->Free(v[m]);
    v[m] = p;
}

}  // namespace

int gpu_build_voxel_fixed(Scene& s, const HipApi* H, int32_t domain, bool* used)
{
    *used = false;
    const DeviceModule& M = *s.module;
    if (!have_build_kernels(M) || domain > 512) return HARE_OK;   // 512^3 cells = 2 GB of counters: host path
    VoxelHost g;
    voxel_grid_bounds(s, g);
    voxel_grid_set_ct(g, domain);
    const long long ncell = (long long)domain * domain * domain;
    const size_t NT = s.topos.size();
    g.start.resize(NT);
    g.items.resize(NT);
    std::vector<void*> cells(NT, nullptr), items(NT, nullptr), occ(NT, nullptr);
    DevMem mem(H);
    int32_t occ_words = 0;
    for (size_t m = 0; m < NT; ++m) {
        BuildArgs b;
        fill_args(b, s, m, g);
        void *d_count, *d_start, *d_items = nullptr;
        int rc = mem.alloc(&d_count, (size_t)(ncell + 1) * 4, true);
        if (rc) return rc;
        rc = mem.alloc(&d_start, (size_t)(ncell + 1) * 4, false);
        if (rc) return rc;
        const unsigned pgrid = (unsigned)std::max(1, (b.P + 255) / 256);
        {
            void* args[] = {&b, &d_count};
            rc = launch(H, M.vb_count, pgrid, 256, 0, nullptr, args);
            if (rc) return rc;
        }
        rc = scan_u32(H, M, mem, d_count, d_start, ncell + 1);
        if (rc) return rc;
        uint32_t total = 0;
        HIP_TRY(H->Memcpy(&total, (const char*)d_start + (size_t)ncell * 4, 4, hipMemcpyDeviceToHost));
        rc = mem.alloc(&d_items, (size_t)total * 4, false);
        if (rc) return rc;
        HIP_TRY(H->MemsetAsync(d_count, 0, (size_t)(ncell + 1) * 4, nullptr));   // reuse as the fill cursor
        {
            void* args[] = {&b, &d_count, &d_start, &d_items};
            rc = launch(H, M.vb_fill, pgrid, 256, 0, nullptr, args);
            if (rc) return rc;
        }
        void *d_big, *d_bigcount;
        const uint32_t small_max = 48;
        rc = mem.alloc(&d_big, ((size_t)total / (small_max + 1) + 1) * 4, false);
        if (rc) return rc;
        rc = mem.alloc(&d_bigcount, 8, true);
        if (rc) return rc;
        {   // voxels with more than 8192 polygons: ordered fill, one block each
            long long nc = ncell;
            void* args[] = {&d_start, &nc, &d_big, &d_bigcount};
            rc = launch(H, M.vb_find_big, (unsigned)((ncell + 255) / 256), 256, 0, nullptr, args);
            if (rc) return rc;
            uint32_t nbig = 0;
            HIP_TRY(H->Memcpy(&nbig, d_bigcount, 4, hipMemcpyDeviceToHost));
            if (nbig > 0) {
                void* fargs[] = {&b, &d_start, &d_items, &d_big};
                rc = launch(H, M.vb_fill_big, nbig, 256, 0, nullptr, fargs);
                if (rc) return rc;
            }
            HIP_TRY(H->MemsetAsync(d_bigcount, 0, 8, nullptr));
        }
        // per-voxel ascending order of the atomically filled lists
        {
            long long nc = ncell;
            uint32_t smax = small_max;
            void* args[] = {&d_start, &d_items, &nc, &d_big, &d_bigcount, &smax};
            rc = launch(H, M.vb_sort_small, (unsigned)((ncell + 255) / 256), 256, 0, nullptr, args);
            if (rc) return rc;
        }
        uint32_t big[2] = {0, 0};
        HIP_TRY(H->Memcpy(big, d_bigcount, 8, hipMemcpyDeviceToHost));
        if (big[0] > 0) {
            void* d_overflow = (char*)d_bigcount + 4;
            void* args[] = {&d_start, &d_items, &d_big, &d_overflow};
            rc = launch(H, M.vb_sort_block, big[0], 256, 0, nullptr, args);
            if (rc) return rc;
            HIP_TRY(H->Memcpy(big, d_bigcount, 8, hipMemcpyDeviceToHost));
            if (big[1] > 0) return HARE_OK;   // a voxel with > 8192 polygons: leave it to the host builder
        }
        unsigned long long stats[2];
        rc = finalize_level(H, M, mem, d_start, d_items, ncell, &cells[m], &occ[m], &occ_words, stats);
        if (rc) return rc;
        items[m] = d_items;
        // mirror the lists on the host (introspection / parity tests)
        g.start[m].resize((size_t)ncell + 1);
        g.items[m].resize(total);
        HIP_TRY(H->Memcpy(g.start[m].data(), d_start, (size_t)(ncell + 1) * 4, hipMemcpyDeviceToHost));
        if (total) HIP_TRY(H->Memcpy(g.items[m].data(), d_items, (size_t)total * 4, hipMemcpyDeviceToHost));
    }
    // commit
    s.d_cells.resize(NT, nullptr);
    s.d_items.resize(NT, nullptr);
    s.d_occ.resize(NT, nullptr);
    for (size_t m = 0; m < NT; ++m) {
        mem.release(cells[m]);
        mem.release(items[m]);
        mem.release(occ[m]);
        replace_buffer(H, s.d_cells, m, cells[m]);
        replace_buffer(H, s.d_items, m, items[m]);
        replace_buffer(H, s.d_occ, m, occ[m]);
    }
    s.occ_words = occ_words;
    g.built = true;
    g.on_device = true;
    s.vox = std::move(g);
    *used = true;
    return HARE_OK;
}

// Hierarchical ctor on the GPU: Voxel_Grid.cs:128-254, level by level, voxel-major over the parent's list.
int gpu_build_voxel_adaptive(Scene& s, const HipApi* H, int32_t max_domain, int32_t avg_polys, bool* used)
{
    *used = false;
    const DeviceModule& M = *s.module;
    if (!have_build_kernels(M) || max_domain > 9) return HARE_OK;
    VoxelHost g;
    voxel_grid_bounds(s, g);
    const size_t NT = s.topos.size();
    DevMem mem(H);
    std::vector<void*> pstart(NT, nullptr), pitems(NT, nullptr), cells(NT, nullptr), occ(NT, nullptr);
    for (size_t m = 0; m < NT; ++m) {   // level "-1": one voxel listing every polygon (:157-166)
        const int32_t P = s.topos[m].P;
        std::vector<int32_t> iota((size_t)std::max(P, 1));
        for (int32_t j = 0; j < P; ++j) iota[j] = j;
        const uint32_t st[2] = {0u, (uint32_t)P};
        int rc = mem.alloc(&pstart[m], 8, false);
        if (rc) return rc;
        rc = mem.alloc(&pitems[m], iota.size() * 4, false);
        if (rc) return rc;
        HIP_TRY(H->Memcpy(pstart[m], st, 8, hipMemcpyHostToDevice));
        HIP_TRY(H->Memcpy(pitems[m], iota.data(), iota.size() * 4, hipMemcpyHostToDevice));
    }
    int32_t ct = 1, occ_words = 0;
    std::vector<uint32_t> totals(NT, 0);
    for (int32_t k = 0; k < max_domain; ++k) {
        const int32_t nct = 2 * ct;
        voxel_grid_set_ct(g, nct);
        const long long ncell = (long long)nct * nct * nct;
        double sum = 0;
        long long cnt = 0;
        for (size_t m = 0; m < NT; ++m) {
            BuildArgs b;
            fill_args(b, s, m, g);
            void *d_count, *d_start, *d_items = nullptr;
            int rc = mem.alloc(&d_count, (size_t)(ncell + 1) * 4, true);
            if (rc) return rc;
            rc = mem.alloc(&d_start, (size_t)(ncell + 1) * 4, false);
            if (rc) return rc;
            const unsigned cgrid = (unsigned)((ncell + 255) / 256);
            {
                void* args[] = {&b, &pstart[m], &pitems[m], &d_count};
                rc = launch(H, M.vb_level_count, cgrid, 256, 0, nullptr, args);
                if (rc) return rc;
            }
            rc = scan_u32(H, M, mem, d_count, d_start, ncell + 1);
            if (rc) return rc;
            uint32_t total = 0;
            HIP_TRY(H->Memcpy(&total, (const char*)d_start + (size_t)ncell * 4, 4, hipMemcpyDeviceToHost));
            rc = mem.alloc(&d_items, (size_t)total * 4, false);
            if (rc) return rc;
            {
                void* args[] = {&b, &pstart[m], &pitems[m], &d_start, &d_items};
                rc = launch(H, M.vb_level_fill, cgrid, 256, 0, nullptr, args);
                if (rc) return rc;
            }
            unsigned long long stats[2];
            void *d_cells, *d_occ;
            rc = finalize_level(H, M, mem, d_start, d_items, ncell, &d_cells, &d_occ, &occ_words, stats);
            if (rc) return rc;
            sum += (double)stats[0];
            cnt += (long long)stats[1];
            // this level becomes the parent of the next; drop the previous level's buffers
            for (void* old : {pstart[m], pitems[m], cells[m], occ[m], d_count}) {
                if (old) {
                    mem.release(old);
                    (void)H->Free(old);
                }
            }
            pstart[m] = d_start;
            pitems[m] = d_items;
            cells[m] = d_cells;
            occ[m] = d_occ;
            totals[m] = total;
        }
        ct = nct;
        if (k > 1 && sum / (double)cnt < avg_polys) break;   // "We are done..." (:252); 0/0 = NaN compares false
    }
    g.start.resize(NT);
    g.items.resize(NT);
    const long long ncell = (long long)ct * ct * ct;
    for (size_t m = 0; m < NT; ++m) {
        g.start[m].resize((size_t)ncell + 1);
        g.items[m].resize(totals[m]);
        HIP_TRY(H->Memcpy(g.start[m].data(), pstart[m], (size_t)(ncell + 1) * 4, hipMemcpyDeviceToHost));
        if (totals[m]) HIP_TRY(H->Memcpy(g.items[m].data(), pitems[m], (size_t)totals[m] * 4, hipMemcpyDeviceToHost));
    }
    s.d_cells.resize(NT, nullptr);
    s.d_items.resize(NT, nullptr);
    s.d_occ.resize(NT, nullptr);
    for (size_t m = 0; m < NT; ++m) {
        mem.release(cells[m]);
        mem.release(pitems[m]);
        mem.release(occ[m]);
        replace_buffer(H, s.d_cells, m, cells[m]);
        replace_buffer(H, s.d_items, m, pitems[m]);
        replace_buffer(H, s.d_occ, m, occ[m]);
    }
    s.occ_words = occ_words;
    g.built = true;
    g.on_device = true;
    s.vox = std::move(g);
    *used = true;
    return HARE_OK;
}

}  // namespace hare
