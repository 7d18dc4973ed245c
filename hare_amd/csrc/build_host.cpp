// build_host.cpp -- host construction of the partitions (Voxel_Grid / Octree / KDTree ctors).
//
// Membership decides candidate ORDER, and order decides which polygon wins an exact-t tie, so the
// lists must equal the reference's: same predicate (AABB.PolyBoxOverlap, hare_math.h), same padded
// boxes, ascending polygon index per cell.  The reference visits every voxel x every polygon
// (Voxel_Grid.cs:273-304, O(D^3 P)); here polygons are binned triangle-major over a conservative
// cell range with the SAME predicate, which yields identical lists in O(P * cells-per-polygon).
//
// Compiled by g++ with -ffp-contract=off.  Product code; nothing from oracle/.
#include <math.h>
#include <string.h>
#include <algorithm>
#include <functional>
#include <thread>

#include "../../include/hare_hip.h"
#include "scene.h"

namespace hare {

// Polygon ctor normal: Hare_Geometry_Polygons.cs:159-171 + Vector.Normalize (Primitives.cs:49-57)
void polygon_normals(const double* verts, const int32_t* nverts, int32_t P, double* out)
{
    for (int32_t p = 0; p < P; ++p) {
        const double* V = verts + (size_t)p * 12;
        double nx = 0, ny = 0, nz = 0;
        for (int j = 2; j < nverts[p]; ++j) {
            const double ax = V[3] - V[0], ay = V[4] - V[1], az = V[5] - V[2];
            const double bx = V[3 * j] - V[0], by = V[3 * j + 1] - V[1], bz = V[3 * j + 2] - V[2];
            nx = ay * bz - az * by;             // Hare_math.Cross, Hare_Geometry_Math.cs:62-65
            ny = -(ax * bz - az * bx);
            nz = ax * by - ay * bx;
            if (!((nx * nx + ny * ny + nz * nz) < 4.9406564584124654e-324)) break;  // !IsZeroVector
        }
        double f = nx * nx + ny * ny + nz * nz;
        if (f != 0) {
            f = sqrt(f);
            nx /= f;
            ny /= f;
            nz /= f;
        }
        out[3 * (size_t)p + 0] = nx;
        out[3 * (size_t)p + 1] = ny;
        out[3 * (size_t)p + 2] = nz;
    }
}

// Finish_Topology: Hare_Geometry_Topology.cs:148-167
void topology_bounds(const double* verts, const int32_t* nverts, int32_t P, double mn[3], double mx[3])
{
    double lo[3] = {1.7976931348623157e308, 1.7976931348623157e308, 1.7976931348623157e308};
    double hi[3] = {-1.7976931348623157e308, -1.7976931348623157e308, -1.7976931348623157e308};
    for (int32_t p = 0; p < P; ++p)
        for (int c = 0; c < nverts[p]; ++c)
            for (int a = 0; a < 3; ++a) {
                const double v = verts[(size_t)p * 12 + 3 * c + a];
                if (lo[a] > v) lo[a] = v;
                if (hi[a] < v) hi[a] = v;
            }
    for (int a = 0; a < 3; ++a) {
        mn[a] = lo[a] - 0.000000000001;
        mx[a] = hi[a] + 0.000000000001;
    }
}

namespace {

int hw_threads()
{
    unsigned n = std::thread::hardware_concurrency();
    if (n == 0) n = 4;
    if (n > 64) n = 64;
    return (int)n;
}

void parallel_for(int64_t n, int nthreads, const std::function<void(int, int64_t, int64_t)>& fn)
{
    if (nthreads > n) nthreads = (int)std::max<int64_t>(1, n);
    if (nthreads <= 1) {
        fn(0, 0, n);
        return;
    }
    std::vector<std::thread> th;
    for (int k = 0; k < nthreads; ++k) th.emplace_back(fn, k, n * k / nthreads, n * (k + 1) / nthreads);
    for (auto& t : th) t.join();
}

// ctor prologue: Voxel_Grid.cs:52-90
void grid_bounds(const Scene& s, VoxelHost& g)
{
    voxel_grid_bounds(s, g);
}
void grid_set_ct(VoxelHost& g, int32_t ct) { voxel_grid_set_ct(g, ct); }

}  // namespace

void voxel_grid_bounds(const Scene& s, VoxelHost& g)
{
    double MaxPT[3] = {-INFINITY, -INFINITY, -INFINITY}, MinPT[3] = {INFINITY, INFINITY, INFINITY};
    for (const Topo& t : s.topos)
        for (int a = 0; a < 3; ++a) {
            if ((t.mx[a] + 0.01) > MaxPT[a]) MaxPT[a] = (t.mx[a] + 0.001);
            if ((t.mn[a] - 0.01) < MinPT[a]) MinPT[a] = (t.mn[a] - 0.001);
        }
    for (int a = 0; a < 3; ++a) {
        g.omin[a] = MinPT[a] - .1;
        g.omax[a] = MaxPT[a] + .1;
        g.box_dims[a] = g.omax[a] - g.omin[a];
    }
}

void voxel_grid_set_ct(VoxelHost& g, int32_t ct)
{
    g.ct = ct;
    for (int a = 0; a < 3; ++a) g.vd[a] = g.box_dims[a] / ct;
    const double* v = g.vd;
    g.char_step = (v[0] < v[1]) ? ((v[0] < v[2]) ? v[0] : v[2]) : (v[1] < v[2] ? v[1] : v[2]);  // :90
}

namespace {

inline void cell_box(const VoxelHost& g, int x, int y, int z, double bmin[3], double bmax[3])
{
    bmin[0] = voxel_lo(x, g.vd[0], g.omin[0]);
    bmax[0] = voxel_hi(x, g.vd[0], g.omin[0]);
    bmin[1] = voxel_lo(y, g.vd[1], g.omin[1]);
    bmax[1] = voxel_hi(y, g.vd[1], g.omin[1]);
    bmin[2] = voxel_lo(z, g.vd[2], g.omin[2]);
    bmax[2] = voxel_hi(z, g.vd[2], g.omin[2]);
}

struct Pair {
    uint32_t cell;
    int32_t poly;
};

}  // namespace

int build_voxel_fixed(Scene& s, int32_t domain)
{
    if (domain < 1 || domain > 1024) {
        set_error("hare_voxel_build: domain must be in [1, 1024]");
        return HARE_E_INVALID;
    }
    VoxelHost g;
    grid_bounds(s, g);
    grid_set_ct(g, domain);
    const int32_t ct = domain;
    const size_t ncell = (size_t)ct * ct * ct;
    g.start.resize(s.topos.size());
    g.items.resize(s.topos.size());
    const int nth = hw_threads();

    for (size_t m = 0; m < s.topos.size(); ++m) {
        const Topo& T = s.topos[m];
        std::vector<std::vector<Pair>> per(nth);
        parallel_for(T.P, nth, [&](int k, int64_t lo, int64_t hi) {
            std::vector<Pair>& out = per[k];
            for (int64_t i = lo; i < hi; ++i) {
                const double* V = &T.verts[(size_t)i * 12];
                const int nv = T.nverts[i];
                int clo[3], chi[3];
                for (int a = 0; a < 3; ++a) {
                    double mn = INFINITY, mx = -INFINITY;
                    for (int c = 0; c < nv; ++c) {
                        mn = std::min(mn, V[3 * c + a]);
                        mx = std::max(mx, V[3 * c + a]);
                    }
                    // conservative: polygon AABB grown by the 1 mm voxel pad (+10 %), one more cell each way
                    double flo = floor((mn - 0.0011 - g.omin[a]) / g.vd[a]) - 1;
                    double fhi = floor((mx + 0.0011 - g.omin[a]) / g.vd[a]) + 1;
                    if (!(flo >= 0)) flo = 0;
                    if (!(fhi <= ct - 1)) fhi = ct - 1;
                    clo[a] = (int)flo;
                    chi[a] = (int)fhi;
                }
                for (int x = clo[0]; x <= chi[0]; ++x)
                    for (int y = clo[1]; y <= chi[1]; ++y)
                        for (int z = clo[2]; z <= chi[2]; ++z) {
                            double bmin[3], bmax[3];
                            cell_box(g, x, y, z, bmin, bmax);
                            if (poly_box_overlap(bmin, bmax, V, nv))
                                out.push_back({(uint32_t)(((size_t)x * ct + y) * ct + z), (int32_t)i});
                        }
            }
        });
        // stable counting sort by cell; threads hold ascending polygon ranges in thread order
        std::vector<uint32_t>& start = g.start[m];
        start.assign(ncell + 1, 0);
        size_t total = 0;
        for (auto& v : per) {
            total += v.size();
            for (const Pair& p : v) start[p.cell + 1]++;
        }
        if (total > 0xFFFFFFF0ull) {
            set_error("hare_voxel_build: more than 2^32 cell entries");
            return HARE_E_UNSUPPORTED;
        }
        for (size_t c = 0; c < ncell; ++c) start[c + 1] += start[c];
        std::vector<uint32_t> cur(start.begin(), start.end() - 1);
        std::vector<int32_t>& items = g.items[m];
        items.resize(total);
        for (auto& v : per)
            for (const Pair& p : v) items[cur[p.cell]++] = p.poly;
    }
    g.built = true;
    s.vox = std::move(g);
    return HARE_OK;
}

// Hierarchical ctor: Voxel_Grid.cs:128-254 -- each level tests a child voxel only against its
// parent's list (:207-215); stop after level k > 1 once the mean list length over non-empty
// voxels drops below Avg_polys (:249-252).
int build_voxel_adaptive(Scene& s, int32_t max_domain, int32_t avg_polys)
{
    if (max_domain < 1 || max_domain > 10) {
        set_error("hare_voxel_build_adaptive: max_domain must be in [1, 10]");
        return HARE_E_INVALID;
    }
    VoxelHost g;
    grid_bounds(s, g);
    const size_t M = s.topos.size();
    int32_t ct = 1;
    std::vector<std::vector<uint32_t>> start(M);
    std::vector<std::vector<int32_t>> items(M);
    for (size_t m = 0; m < M; ++m) {
        start[m] = {0u, (uint32_t)s.topos[m].P};
        items[m].resize(s.topos[m].P);
        for (int32_t j = 0; j < s.topos[m].P; ++j) items[m][j] = j;
    }
    const int nth = hw_threads();
    for (int32_t k = 0; k < max_domain; ++k) {
        const int32_t nct = 2 * ct;
        grid_set_ct(g, nct);
        double sum = 0;
        int cnt = 0;
        for (size_t m = 0; m < M; ++m) {
            const Topo& T = s.topos[m];
            const size_t ncell = (size_t)nct * nct * nct;
            std::vector<std::vector<int32_t>> titems(nth);
            std::vector<uint32_t> count(ncell, 0);
            parallel_for(nct, nth, [&](int kth, int64_t xlo, int64_t xhi) {
                std::vector<int32_t>& out = titems[kth];
                for (int x = (int)xlo; x < (int)xhi; ++x)
                    for (int y = 0; y < nct; ++y)
                        for (int z = 0; z < nct; ++z) {
                            double bmin[3], bmax[3];
                            cell_box(g, x, y, z, bmin, bmax);
                            const size_t par = ((size_t)(x / 2) * ct + (y / 2)) * ct + (z / 2);
                            uint32_t c = 0;
                            for (uint32_t q = start[m][par]; q < start[m][par + 1]; ++q) {
                                const int32_t i = items[m][q];
                                if (poly_box_overlap(bmin, bmax, &T.verts[(size_t)i * 12], T.nverts[i])) {
                                    out.push_back(i);
                                    ++c;
                                }
                            }
                            count[((size_t)x * nct + y) * nct + z] = c;
                        }
            });
            std::vector<uint32_t> nstart(ncell + 1, 0);
            for (size_t c = 0; c < ncell; ++c) {
                nstart[c + 1] = nstart[c] + count[c];
                if (count[c] > 0) {
                    sum += count[c];
                    cnt++;
                }
            }
            std::vector<int32_t> nitems;
            nitems.reserve(nstart[ncell]);
            for (auto& v : titems) nitems.insert(nitems.end(), v.begin(), v.end());
            start[m] = std::move(nstart);
            items[m] = std::move(nitems);
        }
        ct = nct;
        if (k > 1 && sum / cnt < avg_polys) break;
    }
    g.start = std::move(start);
    g.items = std::move(items);
    g.built = true;
    s.vox = std::move(g);
    return HARE_OK;
}

// ---------------------------------------------------------------- Octree ("Octree - alt.cs":45-138)
namespace {

struct OBuild {
    const Topo* T0;
    int max_depth, max_polys;
    std::vector<OctNode> nodes;
    std::vector<std::vector<int32_t>> lists;  // per node
    size_t live_items = 0;                    // entries held by the lists
    int failed = 0;                           // HARE_E_NOMEM: a budget was exceeded, the recursion unwinds
};

void oct_split(OBuild& b, int32_t ni, int depth)
{
    if (b.failed) return;
    if (depth >= b.max_depth || (int)b.lists[ni].size() <= b.max_polys) return;   // :93
    if (b.nodes.size() + 8 > kOctMaxNodes) {
        b.failed = octree_budget_error("nodes");
        return;
    }
    double nmin[3], nmax[3];
    for (int a = 0; a < 3; ++a) {
        nmin[a] = b.nodes[ni].bmin[a];
        nmax[a] = b.nodes[ni].bmax[a];
    }
    const int32_t first = (int32_t)b.nodes.size();
    for (int i = 0; i < 8; ++i) {
        OctNode c;
        memset(&c, 0, sizeof c);
        octree_child_box(nmin, nmax, i, c.bmin, c.bmax);
        c.first_child = -1;
        b.nodes.push_back(c);
        b.lists.emplace_back();
    }
    b.nodes[ni].first_child = first;
    const Topo& T = *b.T0;                       // Model[0].Polygon_Vertices(polyId), :123
    std::vector<int32_t> mine;
    mine.swap(b.lists[ni]);                      // node.Polygons.Clear(), :132
    for (int32_t pid : mine) {
        for (int c = 0; c < 8; ++c) {
            const OctNode& ch = b.nodes[first + c];
            if (poly_box_overlap(ch.bmin, ch.bmax, &T.verts[(size_t)pid * 12], T.nverts[pid])) {
                b.lists[first + c].push_back(pid);
                ++b.live_items;
            }
        }
        if (b.live_items > kOctMaxItems) {
            b.failed = octree_budget_error("polygon-list entries");
            return;
        }
    }
    b.live_items -= mine.size();
    for (int c = 0; c < 8; ++c) oct_split(b, first + c, depth + 1);
}

}  // namespace

// the eight loose children of a node: bit 4 -> x, 2 -> y, 1 -> z ("Octree - alt.cs":96-114)
void octree_child_box(const double nmin[3], const double nmax[3], int i, double cmin[3], double cmax[3])
{
    const int bit[3] = {4, 2, 1};
    for (int a = 0; a < 3; ++a) {
        const double center = (nmax[a] + nmin[a]) / 2;     // AABB.Center, AABB_Main.cs:64
        cmin[a] = ((i & bit[a]) == 0 ? nmin[a] : center) - 0.1;
        cmax[a] = ((i & bit[a]) == 0 ? center : nmax[a]) + 0.1;
    }
}

// root cube ("Octree - alt.cs":63-88)
void octree_root_box(const Topo& T, double bmin[3], double bmax[3])
{
    double mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int32_t p = 0; p < T.P; ++p)
        for (int c = 0; c < T.nverts[p]; ++c)
            for (int a = 0; a < 3; ++a) {
                const double v = T.verts[(size_t)p * 12 + 3 * c + a];
                if (v < mn[a]) mn[a] = v;
                if (v > mx[a]) mx[a] = v;
            }
    const double maxdim = net_max(mx[0] - mn[0], net_max(mx[1] - mn[1], mx[2] - mn[2]));   // :78
    for (int a = 0; a < 3; ++a) {
        const double center = mx[a] + mn[a] / 2;   // `max + min / 2` as written (:79)
        bmin[a] = center - maxdim - 1e-1;
        bmax[a] = center + maxdim + 1e-1;
    }
}

int octree_budget_error(const char* what)
{
    set_error(std::string("hare_octree_build: the tree outgrows its budget of ") + what + " (2^24 nodes, 2^28 list entries).  The reference "
              "pads child boxes by an absolute 0.1 m (\"Octree - alt.cs\":99-111): once nodes are smaller than ~0.4 m every polygon lands in "
              "all eight children and the tree grows 8x per level -- lower maxDepth (a node of the root's size halves per level)");
    return HARE_E_NOMEM;
}

int octree_check_args(const Scene& s, int32_t max_depth, int32_t max_polys)
{
    if (max_depth < 0 || max_depth > 24 || max_polys < 0) {
        set_error("hare_octree_build: max_depth must be in [0, 24], max_polys >= 0");
        return HARE_E_INVALID;
    }
    if (s.topos.empty()) {
        set_error("hare_octree_build: no topology");
        return HARE_E_INVALID;
    }
    // Several topologies, as the reference has it ("Octree - alt.cs":63-88): a fresh root per topology, the LAST one's
    // stays; its polygon ids 0..P-1 are binned by the vertices of Model[0] (:123).  A topology that is split and has
    // more polygons than Model[0] makes the reference index Model[0] out of range.
    for (size_t m = 1; m < s.topos.size(); ++m)
        if (s.topos[m].P > s.topos[0].P && max_depth > 0 && s.topos[m].P > max_polys) {
            set_error("hare_octree_build: topology " + std::to_string(m) + " has more polygons than topology 0, whose vertices the "
                      "reference bins every topology's polygon ids by (\"Octree - alt.cs\":123): IndexOutOfRangeException there");
            return HARE_E_INVALID;
        }
    return HARE_OK;
}

int build_octree(Scene& s, int32_t max_depth, int32_t max_polys)
{
    if (int rc = octree_check_args(s, max_depth, max_polys)) return rc;
    const Topo& T = s.topos.back();      // root cube and id range: the last topology's (:63-88)
    OBuild b;
    b.T0 = &s.topos[0];                  // membership: Model[0]'s vertices (:123)
    b.max_depth = max_depth;
    b.max_polys = max_polys;
    OctNode root;
    memset(&root, 0, sizeof root);
    octree_root_box(T, root.bmin, root.bmax);
    root.first_child = -1;
    b.nodes.push_back(root);
    b.lists.emplace_back();
    b.lists[0].resize(T.P);
    for (int32_t i = 0; i < T.P; ++i) b.lists[0][i] = i;
    b.live_items = (size_t)T.P;
    oct_split(b, 0, 0);
    if (b.failed) return b.failed;

    OctreeHost o;
    o.max_depth = max_depth;
    o.max_polys = max_polys;
    o.nodes = std::move(b.nodes);
    size_t tot = 0;
    for (auto& l : b.lists) tot += l.size();
    if (tot > 0x7FFFFFF0ull) {
        set_error("hare_octree_build: more than 2^31 leaf entries");
        return HARE_E_UNSUPPORTED;
    }
    o.items.reserve(tot);
    for (size_t i = 0; i < o.nodes.size(); ++i) {
        o.nodes[i].item_start = (int32_t)o.items.size();
        o.nodes[i].item_count = (int32_t)b.lists[i].size();
        o.items.insert(o.items.end(), b.lists[i].begin(), b.lists[i].end());
    }
    o.id_count = T.P;
    o.built = true;
    s.oct = std::move(o);
    return HARE_OK;
}

// ---------------------------------------------------------------- KDTree (KDTree.cs:51-139)
namespace {

struct KBuild {
    const Topo* T0;
    int max_depth, max_polys, depth_reached;
    std::vector<double> cent;  // Polygon_Centroid of Model[0], Hare_Geometry_Topology.cs:566-574
    std::vector<KdNodeRec> nodes;
    std::vector<std::vector<int32_t>> lists;
};

int32_t kd_new(KBuild& b, const double mn[3], const double mx[3])
{
    KdNodeRec n;
    memset(&n, 0, sizeof n);
    for (int a = 0; a < 3; ++a) {
        n.bmin[a] = mn[a];
        n.bmax[a] = mx[a];
    }
    n.left = n.right = -1;
    n.axis = -1;
    b.nodes.push_back(n);
    b.lists.emplace_back();
    return (int32_t)b.nodes.size() - 1;
}

void kd_split(KBuild& b, int32_t ni, int depth, const double mn[3], const double mx[3])
{
    if (depth > b.depth_reached) b.depth_reached = depth;
    if (depth >= b.max_depth || (int)b.lists[ni].size() <= b.max_polys) return;   // :92
    const Topo& T = *b.T0;
    const int axis = depth % 3;
    std::vector<int32_t> sorted;
    sorted.swap(b.lists[ni]);
    // Enumerable.OrderBy is a stable sort on double.CompareTo (NaN first)
    std::stable_sort(sorted.begin(), sorted.end(), [&](int32_t x, int32_t y) {
        const double kx = b.cent[3 * (size_t)x + axis], ky = b.cent[3 * (size_t)y + axis];
        if (kx < ky) return true;
        if (kx > ky || kx == ky) return false;
        return (kx != kx) && !(ky != ky);
    });
    const double split = b.cent[3 * (size_t)sorted[sorted.size() / 2] + axis];     // :104-105
    b.nodes[ni].axis = axis;
    b.nodes[ni].split = split;
    double leftMax[3] = {mx[0], mx[1], mx[2]}, rightMin[3] = {mn[0], mn[1], mn[2]};
    leftMax[axis] = split;
    rightMin[axis] = split;
    const int32_t L = kd_new(b, mn, leftMax);
    const int32_t R = kd_new(b, rightMin, mx);
    b.nodes[ni].left = L;
    b.nodes[ni].right = R;
    for (int32_t id : sorted) {                                                      // :123-133
        bool le = false, gt = false;
        for (int c = 0; c < T.nverts[id]; ++c) {
            const double v = T.verts[(size_t)id * 12 + 3 * c + axis];
            if (v <= split) le = true;
            if (v > split) gt = true;
        }
        if (le) b.lists[L].push_back(id);
        if (gt) b.lists[R].push_back(id);
    }
    kd_split(b, L, depth + 1, mn, leftMax);
    kd_split(b, R, depth + 1, rightMin, mx);
}

}  // namespace

int build_kdtree(Scene& s, int32_t max_depth, int32_t max_polys)
{
    if (max_depth < 0 || max_depth > 60 || max_polys < 0) {
        set_error("hare_kdtree_build: max_depth must be in [0, 60], max_polys >= 0");
        return HARE_E_INVALID;
    }
    if (s.topos.empty()) {
        set_error("hare_kdtree_build: no topology");
        return HARE_E_INVALID;
    }
    // Several topologies, as the reference has it (KDTree.cs:67-87): the bounding box grows over the topologies, a fresh
    // root per topology, the LAST one's stays; its polygon ids are split by the centroids and vertices of Model[0] (:98-133).
    for (size_t m = 1; m < s.topos.size(); ++m)
        if (s.topos[m].P > s.topos[0].P && max_depth > 0 && s.topos[m].P > max_polys) {
            set_error("hare_kdtree_build: topology " + std::to_string(m) + " has more polygons than topology 0, whose centroids the "
                      "reference splits every topology's polygon ids by (KDTree.cs:98-105): IndexOutOfRangeException there");
            return HARE_E_INVALID;
        }
    const Topo& T = s.topos[0];          // centroids and membership: Model[0]
    const Topo& TL = s.topos.back();     // id range: the last topology's
    KBuild b;
    b.T0 = &T;
    b.max_depth = max_depth;
    b.max_polys = max_polys;
    b.depth_reached = 0;
    b.cent.resize((size_t)std::max(T.P, 1) * 3);
    for (int32_t p = 0; p < T.P; ++p) {
        double sx = 0, sy = 0, sz = 0;
        for (int c = 0; c < T.nverts[p]; ++c) {
            sx = sx + T.verts[(size_t)p * 12 + 3 * c + 0];
            sy = sy + T.verts[(size_t)p * 12 + 3 * c + 1];
            sz = sz + T.verts[(size_t)p * 12 + 3 * c + 2];
        }
        b.cent[3 * (size_t)p + 0] = sx / T.nverts[p];
        b.cent[3 * (size_t)p + 1] = sy / T.nverts[p];
        b.cent[3 * (size_t)p + 2] = sz / T.nverts[p];
    }
    double mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (const Topo& Tm : s.topos)       // min / max are not reset between topologies (:67-82)
        for (int32_t p = 0; p < Tm.P; ++p)
            for (int c = 0; c < Tm.nverts[p]; ++c)
                for (int a = 0; a < 3; ++a) {
                    const double v = Tm.verts[(size_t)p * 12 + 3 * c + a];
                    if (v < mn[a]) mn[a] = v;
                    if (v > mx[a]) mx[a] = v;
                }
    const int32_t root = kd_new(b, mn, mx);
    b.lists[root].resize(TL.P);
    for (int32_t i = 0; i < TL.P; ++i) b.lists[root][i] = i;
    kd_split(b, root, 0, mn, mx);

    KdHost k;
    k.max_depth = max_depth;
    k.max_polys = max_polys;
    k.depth_reached = b.depth_reached;
    k.nodes = std::move(b.nodes);
    size_t tot = 0;
    for (auto& l : b.lists) tot += l.size();
    if (tot > 0x7FFFFFF0ull) {
        set_error("hare_kdtree_build: more than 2^31 leaf entries");
        return HARE_E_UNSUPPORTED;
    }
    k.items.reserve(tot);
    for (size_t i = 0; i < k.nodes.size(); ++i) {
        k.nodes[i].item_start = (int32_t)k.items.size();
        k.nodes[i].item_count = (int32_t)b.lists[i].size();
        k.items.insert(k.items.end(), b.lists[i].begin(), b.lists[i].end());
    }
    k.id_count = TL.P;
    k.built = true;
    s.kd = std::move(k);
    return HARE_OK;
}

}  // namespace hare
