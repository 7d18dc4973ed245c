// ingest.cpp -- Topology(Point[][]) ingest for hosts that start from a raw polygon soup.
//
// The managed Topology rounds every incoming corner (Point.Round(15)) and looks it up in a two-level
// dictionary keyed by Point.Hash2: a 1 m bucket of the model box and a 1 mm position inside the
// bucket.  A corner whose key is already present is REPLACED by the vertex stored under that key (the
// first corner that landed there), so two corners closer than about a millimetre usually share one
// Vertex object -- and it is that shared coordinate the polygon tests later read.  A host that feeds
// the library from its own mesh has to reproduce the merge or its polygons differ from the managed
// ones by up to a millimetre.
//
// Follows Hare_Geometry_Topology.cs:120-142 (ctor), :258-311 (Build_Topology), :342-377
// (AddGetIndex), :677-697 (MS_AABB); Hare_Geometry_Primitives.cs:230-250 (Round, Hash2).
// Product code; nothing from oracle/.  g++ -ffp-contract=off.
#include <math.h>
#include <stdint.h>
#include <unordered_map>
#include <vector>

#include "../../include/hare_hip.h"
#include "scene.h"

namespace hare {

namespace {

// System.Math.Round(double, 15), MidpointRounding.ToEven: scale, round half to even, unscale; values
// of magnitude >= 1e16 are returned unchanged.
double round15(double v)
{
    if (fabs(v) < 1e16) {
        v = v * 1e15;
        v = nearbyint(v);      // default rounding mode = to nearest even
        v = v / 1e15;
    }
    return v;
}

struct Key {
    uint64_t bucket, pos;
    bool operator==(const Key& o) const { return bucket == o.bucket && pos == o.pos; }
};

struct KeyHash {
    size_t operator()(const Key& k) const
    {
        uint64_t h = (k.bucket + 0x9E3779B97F4A7C15ull) * 0xD6E8FEB86659FD93ull;
        h ^= k.pos + (h >> 31);
        h *= 0xFF51AFD7ED558CCDull;
        return (size_t)(h ^ (h >> 33));
    }
};

}  // namespace

int topology_ingest(const double* soup, const int32_t* nverts, int32_t P, double* verts_out, int32_t* corner_vertex,
                    std::vector<double>& vertices)
{
    // ctor: bounds of the raw corners -/+ 1e-12 define Modspace
    double lo[3] = {1.7976931348623157e308, 1.7976931348623157e308, 1.7976931348623157e308};
    double hi[3] = {-1.7976931348623157e308, -1.7976931348623157e308, -1.7976931348623157e308};
    for (int32_t p = 0; p < P; ++p)
        for (int c = 0; c < nverts[p]; ++c)
            for (int a = 0; a < 3; ++a) {
                const double v = soup[(size_t)p * 12 + 3 * c + a];
                if (lo[a] > v) lo[a] = v;
                if (hi[a] < v) hi[a] = v;
            }
    double ms_min[3], ms_max[3];
    for (int a = 0; a < 3; ++a) {
        ms_min[a] = lo[a] - 0.000000000001;
        ms_max[a] = hi[a] + 0.000000000001;
    }
    // MS_AABB: one cubic lattice of 1 m buckets, edge = the longest side rounded up.  xdim*ydim is an
    // `int` product in the reference and reaches Hash2 through (ulong): keep both conversions.
    const int cx = (int)ceil(ms_max[0] - ms_min[0]), cy = (int)ceil(ms_max[1] - ms_min[1]), cz = (int)ceil(ms_max[2] - ms_min[2]);
    const int32_t dim = std::max(cx, std::max(cy, cz));
    const uint64_t ydim = (uint64_t)(int64_t)dim;
    const uint64_t xytot = (uint64_t)(int64_t)(int32_t)((uint32_t)dim * (uint32_t)dim);

    std::unordered_map<Key, int32_t, KeyHash> seen;
    size_t corners = 0;
    for (int32_t p = 0; p < P; ++p) corners += (size_t)nverts[p];
    seen.reserve(corners);
    vertices.clear();
    vertices.reserve(corners * 3);

    for (int32_t p = 0; p < P; ++p) {
        for (int c = 0; c < 4; ++c) {
            for (int a = 0; a < 3; ++a) verts_out[(size_t)p * 12 + 3 * c + a] = 0.0;
            if (corner_vertex) corner_vertex[(size_t)p * 4 + c] = -1;
        }
        for (int c = 0; c < nverts[p]; ++c) {
            const double x = round15(soup[(size_t)p * 12 + 3 * c + 0]);
            const double y = round15(soup[(size_t)p * 12 + 3 * c + 1]);
            const double z = round15(soup[(size_t)p * 12 + 3 * c + 2]);
            // Hash2
            const double xoff = x - ms_min[0], yoff = y - ms_min[1], zoff = z - ms_min[2];
            const uint64_t xl = (uint64_t)floor(xoff), yl = (uint64_t)floor(yoff), zl = (uint64_t)floor(zoff);
            Key k;
            k.bucket = xytot * zl + ydim * xl + yl;
            const uint64_t xp = (uint64_t)((xoff - (double)xl) * 1000), yp = (uint64_t)((yoff - (double)yl) * 1000),
                           zp = (uint64_t)((zoff - (double)zl) * 1000);
            k.pos = 1000000 * zp + 1000 * xp + yp;
            // AddGetIndex
            int32_t index;
            auto it = seen.find(k);
            if (it != seen.end()) {
                index = it->second;
            } else {
                index = (int32_t)(vertices.size() / 3);
                vertices.push_back(x);
                vertices.push_back(y);
                vertices.push_back(z);
                seen.emplace(k, index);
            }
            for (int a = 0; a < 3; ++a) verts_out[(size_t)p * 12 + 3 * c + a] = vertices[3 * (size_t)index + a];
            if (corner_vertex) corner_vertex[(size_t)p * 4 + c] = index;
        }
    }
    return (int)(vertices.size() / 3);
}

}  // namespace hare
