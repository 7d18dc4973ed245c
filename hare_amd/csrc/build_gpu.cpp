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
int finalize_level(const HipApi* H, const DeviceModule& M, DevMem& mem, const void* d_start, const void* d_items, int ct,
                   void** d_cells, void** d_occ, int32_t* occ_words, int32_t* occ_shift, int32_t* occ_cd, unsigned long long stats_out[2])
{
    long long ncell = (long long)ct * ct * ct;
    occ_layout(ct, *occ_shift, *occ_cd, *occ_words);
    const size_t occ_bytes = (size_t)((*occ_words + 3) / 4) * 16;
    int rc = mem.alloc(d_cells, (size_t)ncell * sizeof(CellRec), false);
    if (rc) return rc;
    rc = mem.alloc(d_occ, occ_bytes, true);
    if (rc) return rc;
    void* d_stats = nullptr;
    rc = mem.alloc(&d_stats, 16, true);
    if (rc) return rc;
    void* args[] = {(void*)&d_start, (void*)&d_items, d_cells, d_occ, &ncell, &d_stats, &ct, occ_shift, occ_cd};
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
    int32_t occ_words = 0, occ_shift = 0, occ_cd = 0;
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
        rc = finalize_level(H, M, mem, d_start, d_items, g.ct, &cells[m], &occ[m], &occ_words, &occ_shift, &occ_cd, stats);
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
    s.occ_shift = occ_shift;
    s.occ_cd = occ_cd;
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
    int32_t ct = 1, occ_words = 0, occ_shift = 0, occ_cd = 0;
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
            rc = finalize_level(H, M, mem, d_start, d_items, nct, &d_cells, &d_occ, &occ_words, &occ_shift, &occ_cd, stats);
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
    s.occ_shift = occ_shift;
    s.occ_cd = occ_cd;
    g.built = true;
    g.on_device = true;
    s.vox = std::move(g);
    *used = true;
    return HARE_OK;
}

// ---------------------------------------------------------------------------------------------------
// Octree.BuildOctree on the GPU ("Octree - alt.cs":91-138).  Level by level: the host decides which nodes
// of the level split (depth < maxDepth and more than maxPolygonsPerNode polygons, :93) and writes the
// children's boxes; the GPU runs every PolyBoxOverlap of the level (hare_ob_count -> host prefix ->
// hare_ob_fill, ordered, so a child's list keeps its parent's ascending order exactly like the
// reference's foreach).  Nodes are numbered afterwards in the host builder's (depth-first) order, so both
// builders return identical arrays.
int gpu_build_octree(Scene& s, const HipApi* H, int32_t max_depth, int32_t max_polys, bool* used)
{
    *used = false;
    const DeviceModule& M = *s.module;
    if (!M.ob_count || !M.ob_fill) return HARE_OK;
    if (s.topos.size() != 1) return HARE_OK;   // several topologies (root of the last, membership by the first): host builder
    const Topo& T = s.topos[0];
    DevMem mem(H);
    BuildArgs b;
    memset(&b, 0, sizeof b);
    b.polys = (const PolyRec*)s.d_polys[0];
    b.quads = (const QuadRec*)s.d_quads[0];
    b.P = T.P;

    struct BNode {                 // breadth-first working node
        double bmin[3], bmax[3];
        int32_t first_child;       // BFS index, -1 leaf
        int32_t level;             // which level array holds its list
        uint32_t start, count;     // list = level_items[level][start .. start + count)
    };
    std::vector<BNode> nodes;
    std::vector<std::vector<int32_t>> level_items;     // host mirrors of the per-level item arrays
    {
        BNode r;
        memset(&r, 0, sizeof r);
        octree_root_box(T, r.bmin, r.bmax);
        r.first_child = -1;
        r.level = 0;
        r.start = 0;
        r.count = (uint32_t)T.P;
        nodes.push_back(r);
        level_items.emplace_back((size_t)T.P);
        for (int32_t i = 0; i < T.P; ++i) level_items[0][i] = i;
    }
    void* d_cur = nullptr;         // device copy of the current level's item array
    int rc = mem.alloc(&d_cur, (size_t)T.P * 4, false);
    if (rc) return rc;
    if (T.P > 0) HIP_TRY(H->Memcpy(d_cur, level_items[0].data(), (size_t)T.P * 4, hipMemcpyHostToDevice));

    constexpr uint32_t kSeg = 8192;                    // parent-list entries per task
    size_t level_begin = 0;                            // nodes of the current level: [level_begin, nodes.size())
    for (int depth = 0; depth < max_depth; ++depth) {
        const size_t level_end = nodes.size();
        std::vector<OctTask> tasks;
        std::vector<uint32_t> task_child;              // BFS index of the child a task belongs to
        for (size_t n = level_begin; n < level_end; ++n) {
            if ((int64_t)nodes[n].count <= (int64_t)max_polys) continue;          // :93
            if (nodes.size() + 8 > kOctMaxNodes) return octree_budget_error("nodes");
            nodes[n].first_child = (int32_t)nodes.size();
            const BNode parent = nodes[n];
            for (int i = 0; i < 8; ++i) {
                BNode c;
                memset(&c, 0, sizeof c);
                octree_child_box(parent.bmin, parent.bmax, i, c.bmin, c.bmax);
                c.first_child = -1;
                c.level = depth + 1;
                const uint32_t child = (uint32_t)nodes.size();
                nodes.push_back(c);
                for (uint32_t q = 0; q < parent.count; q += kSeg) {
                    OctTask t;
                    memset(&t, 0, sizeof t);
                    for (int a = 0; a < 3; ++a) {
                        t.bmin[a] = c.bmin[a];
                        t.bmax[a] = c.bmax[a];
                    }
                    t.pstart = parent.start + q;
                    t.pcount = std::min(kSeg, parent.count - q);
                    tasks.push_back(t);
                    task_child.push_back(child);
                }
            }
        }
        if (tasks.empty()) break;
        if (tasks.size() > 0x7FFFFFFFull) {
            set_error("hare_octree_build: level too large for one launch");
            return HARE_E_UNSUPPORTED;
        }
        void* d_tasks = nullptr;
        void* d_counts = nullptr;
        rc = mem.alloc(&d_tasks, tasks.size() * sizeof(OctTask), false);
        if (rc) return rc;
        rc = mem.alloc(&d_counts, tasks.size() * 4, false);
        if (rc) return rc;
        HIP_TRY(H->Memcpy(d_tasks, tasks.data(), tasks.size() * sizeof(OctTask), hipMemcpyHostToDevice));
        {
            void* args[] = {&b, &d_tasks, &d_cur, &d_counts};
            rc = launch(H, M.ob_count, (unsigned)tasks.size(), 256, 0, nullptr, args);
            if (rc) return rc;
        }
        std::vector<uint32_t> counts(tasks.size());
        HIP_TRY(H->Memcpy(counts.data(), d_counts, tasks.size() * 4, hipMemcpyDeviceToHost));
        // prefix over tasks (tasks of one child are consecutive and in list order) = the next level's item array
        uint64_t total = 0;
        for (size_t k = 0; k < tasks.size(); ++k) {
            BNode& c = nodes[task_child[k]];
            if (k == 0 || task_child[k] != task_child[k - 1]) c.start = (uint32_t)total;
            tasks[k].ostart = (uint32_t)total;
            c.count += counts[k];
            total += counts[k];
            if (total > kOctMaxItems) return octree_budget_error("polygon-list entries");
        }
        void* d_next = nullptr;
        rc = mem.alloc(&d_next, (size_t)total * 4, false);
        if (rc) return rc;
        HIP_TRY(H->Memcpy(d_tasks, tasks.data(), tasks.size() * sizeof(OctTask), hipMemcpyHostToDevice));
        {
            void* args[] = {&b, &d_tasks, &d_cur, &d_next};
            rc = launch(H, M.ob_fill, (unsigned)tasks.size(), 256, 0, nullptr, args);
            if (rc) return rc;
        }
        level_items.emplace_back((size_t)total);
        if (total) HIP_TRY(H->Memcpy(level_items.back().data(), d_next, (size_t)total * 4, hipMemcpyDeviceToHost));
        // the parent level's device array and this level's scratch are no longer needed
        for (void* p : {d_cur, d_tasks, d_counts}) {
            mem.release(p);
            (void)H->Free(p);
        }
        d_cur = d_next;
        level_begin = level_end;
    }

    // depth-first renumbering = the order in which the recursive reference / host builder allocates nodes:
    // a node's eight children get consecutive numbers when the node is processed, then each subtree in turn
    std::vector<int32_t> order;                 // order[new] = BFS index
    order.reserve(nodes.size());
    std::vector<int32_t> renum(nodes.size(), -1);
    {
        order.push_back(0);
        renum[0] = 0;
        std::vector<std::pair<int32_t, int>> stack;      // (BFS node, next child to descend into)
        auto open = [&](int32_t n) {
            if (nodes[n].first_child >= 0) {
                for (int i = 0; i < 8; ++i) {
                    renum[nodes[n].first_child + i] = (int32_t)order.size();
                    order.push_back(nodes[n].first_child + i);
                }
                stack.emplace_back(n, 0);
            }
        };
        open(0);
        while (!stack.empty()) {
            auto& top = stack.back();
            if (top.second == 8) {
                stack.pop_back();
                continue;
            }
            const int32_t child = nodes[top.first].first_child + top.second++;
            open(child);
        }
    }
    OctreeHost o;
    o.max_depth = max_depth;
    o.max_polys = max_polys;
    o.nodes.resize(nodes.size());
    size_t tot = 0;
    for (const BNode& n : nodes)
        if (n.first_child < 0) tot += n.count;
    if (tot > 0x7FFFFFF0ull) {
        set_error("hare_octree_build: more than 2^31 leaf entries");
        return HARE_E_UNSUPPORTED;
    }
    o.items.reserve(tot);
    for (size_t k = 0; k < order.size(); ++k) {
        const BNode& n = nodes[order[k]];
        OctNode r;
        memset(&r, 0, sizeof r);
        for (int a = 0; a < 3; ++a) {
            r.bmin[a] = n.bmin[a];
            r.bmax[a] = n.bmax[a];
        }
        r.first_child = n.first_child >= 0 ? renum[n.first_child] : -1;
        r.item_start = (int32_t)o.items.size();
        if (n.first_child < 0) {                 // a node that split has handed its list on (node.Polygons.Clear(), :132)
            r.item_count = (int32_t)n.count;
            const int32_t* src = level_items[n.level].data() + n.start;
            o.items.insert(o.items.end(), src, src + n.count);
        }
        o.nodes[k] = r;
    }
    o.id_count = T.P;
    o.built = true;
    o.built_on_device = true;
    s.oct = std::move(o);
    *used = true;
    return HARE_OK;
}

}  // namespace hare
