// bandsolve.hip -- the two boundary systems of the row-band protocol (host C++, no device work).
//
// Every rank solves them on the gathered seam rows of ALL bands (2 rows of W cells per band), identically, once per step; as
// NumPy loops over R * 2 * W nodes (ufunc.at, unique, masked gathers per level) they cost 100-300 ms at 8 bands of 65536 columns
// -- more than the band's own GPU work.  Here each is one O(n) pass.
//   mhip_band_forest_solve   malstroem_amd/distributed.py: solve_band_accum -- accumulation over the forest of seam crossings
//   mhip_band_ws_resolve     BandPipeline.watershed -- pseudo labels of the seam rows resolved through the other bands' rows
#include "common.hpp"
#include <vector>

// val[i] > 0: the node's own (known) contribution; parent[i] >= 0: the node its flux continues in.  A node is FINAL once all
// its children are final and its own contribution is known; val[i] then is own + everything upstream.  Every other node (own
// contribution unknown, an unknown child somewhere upstream, a cycle) ends as 0 = unknown.  Sums are integers below 2**53.
extern "C" int mhip_band_forest_solve(int64_t n, const int64_t *parent, double *val)
{
    MH_ARG(n >= 0 && (n == 0 || (parent && val)), "band_forest_solve(n, parent, val)");
    std::vector<int32_t> nchild((size_t)n, 0);
    std::vector<uint8_t> state((size_t)n, 0);      // bit 0: own contribution known, bit 1: final
    for (int64_t i = 0; i < n; ++i) {
        const int64_t p = parent[i];
        MH_ARG(p >= -1 && p < n, "band_forest_solve: parent out of range");
        if (p >= 0) ++nchild[(size_t)p];
        state[(size_t)i] = val[i] > 0.0 ? 1 : 0;
    }
    for (int64_t i = 0; i < n; ++i) {
        if (nchild[(size_t)i] || (state[(size_t)i] & 3) != 1) continue;     // not a leaf, unknown, or completed by an earlier walk
        // a leaf with a known contribution: walk up for as long as the walk completes its parents
        int64_t k = i;
        for (;;) {
            state[(size_t)k] |= 2;
            const int64_t p = parent[k];
            if (p < 0) break;
            val[p] += val[k];
            if (--nchild[(size_t)p] != 0 || !(state[(size_t)p] & 1)) break;   // (an unknown parent never completes)
            k = p;
        }
    }
    for (int64_t i = 0; i < n; ++i)
        if (!(state[(size_t)i] & 2)) val[i] = 0.0;
    return MHIP_OK;
}

// vals[i] >= 0: a resolved label (0: none); vals[i] < 0: "whatever node -vals[i] - 1 resolves to".  Chains are followed to
// their end (with path compression); a cycle resolves to 0 (a flow cycle across bands stays unassigned).
extern "C" int mhip_band_ws_resolve(int64_t n, int64_t *vals)
{
    MH_ARG(n >= 0 && (n == 0 || vals), "band_ws_resolve(n, vals)");
    std::vector<int64_t> path;
    for (int64_t i = 0; i < n; ++i) {
        if (vals[i] >= 0) continue;
        path.clear();
        int64_t k = i, res = 0;
        for (;;) {
            const int64_t v = vals[k];
            if (v >= 0) { res = v; break; }
            const int64_t nx = -v - 1;
            MH_ARG(nx < n, "band_ws_resolve: pointer out of range");
            path.push_back(k);
            if ((int64_t)path.size() > n) { res = 0; break; }         // (cannot happen: the marks below end every cycle)
            vals[k] = INT64_MIN;                                        // on the current path
            if (vals[nx] == INT64_MIN) { res = 0; break; }              // a cycle
            k = nx;
        }
        for (int64_t q : path) vals[q] = res;
    }
    return MHIP_OK;
}

// Connected classes of an undirected graph on nodes 0 .. n-1 given by m edges (a[k], b[k]): cls[i] = the smallest node of i's
// class (the label merge of BandPipeline.label works on the few thousand seam pairs the bands publish, not on their rows).
extern "C" int mhip_band_union_find(int64_t n, int64_t m, const int64_t *a, const int64_t *b, int64_t *cls)
{
    MH_ARG(n >= 0 && m >= 0 && (n == 0 || cls) && (m == 0 || (a && b)), "band_union_find(n, m, a, b, cls)");
    for (int64_t i = 0; i < n; ++i) cls[i] = i;
    auto find = [&](int64_t x) {
        while (cls[x] != x) {
            cls[x] = cls[cls[x]];
            x = cls[x];
        }
        return x;
    };
    for (int64_t k = 0; k < m; ++k) {
        MH_ARG(a[k] >= 0 && a[k] < n && b[k] >= 0 && b[k] < n, "band_union_find: node out of range");
        const int64_t x = find(a[k]), y = find(b[k]);
        if (x != y) cls[x < y ? y : x] = x < y ? x : y;      // the smaller node is the root
    }
    for (int64_t i = 0; i < n; ++i) cls[i] = find(i);
    return MHIP_OK;
}
