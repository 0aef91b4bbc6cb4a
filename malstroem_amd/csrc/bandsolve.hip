// bandsolve.hip -- the two boundary systems of the row-band protocol (host C++, no device work).
//
// Every rank solves them on the gathered seam rows of ALL bands (2 rows of W cells per band), identically, once per step; as
// NumPy loops over R * 2 * W nodes (ufunc.at, unique, masked gathers per level) they cost 100-300 ms at 8 bands of 65536 columns
// -- more than the band's own GPU work.  Here each is one O(n) pass.
//   mhip_band_forest_solve   malstroem_amd/distributed.py: solve_band_accum -- accumulation over the forest of seam crossings
//   mhip_band_ws_resolve     BandPipeline.watershed -- pseudo labels of the seam rows resolved through the other bands' rows
#include "common.hpp"
#include <algorithm>
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

// ---- the host sections of BandPipeline.accum / label / watershed as single passes ----------------------------------------------
// Every rank runs them once per step on the seam rows (2 * W cells per band) and on the gathered pairs of all bands.  As NumPy
// expressions (flatnonzero, unique with inverse, searchsorted, ufunc.at over 10^5 .. 10^6 elements) they took 25-30 ms per step and
// rank at 8 bands of 65536 columns; the functions below do the same work in one or two passes each.

// accum, before the gather: one (child, parent) pair per cell of a neighbour's edge row whose flux crosses this band and leaves it
// again.  exit_half[k] = side * W + column of the leaving edge cell or -1; child = child_base + k, parent = parent_base + exit;
// a pair = {int32 child, int32 parent, double own contribution of the child, of the parent} (24 bytes).
struct BandAccumPair { int32_t child, parent; double own_child, own_parent; };

extern "C" int mhip_band_accum_pairs(int64_t W, const int32_t *exit_half, const double *nbr, const double *own_edge, int64_t child_base,
                                     int64_t parent_base, void *pairs, int64_t *n)
{
    MH_ARG(W >= 0 && n && (W == 0 || (exit_half && nbr && own_edge && pairs)), "band_accum_pairs(...)");
    MH_ARG(child_base >= 0 && parent_base >= 0 && child_base + W < (int64_t(1) << 31) && parent_base + 2 * W < (int64_t(1) << 31),
           "band_accum_pairs: node ids are 31-bit");
    BandAccumPair *out = (BandAccumPair *)pairs;
    int64_t k = 0;
    for (int64_t j = 0; j < W; ++j) {
        const int32_t e = exit_half[j];
        if (e < 0) continue;
        MH_ARG(e < 2 * W, "band_accum_pairs: exit cell out of range");
        out[k].child = (int32_t)(child_base + j);
        out[k].parent = (int32_t)(parent_base + e);
        out[k].own_child = nbr[j];
        out[k].own_parent = own_edge[e];
        ++k;
    }
    *n = k;
    return MHIP_OK;
}

static void forest_solve(int64_t n, const int32_t *parent, double *val)
{
    std::vector<int32_t> nchild((size_t)n, 0);
    std::vector<uint8_t> state((size_t)n, 0);
    for (int64_t i = 0; i < n; ++i) {
        if (parent[i] >= 0) ++nchild[(size_t)parent[i]];
        state[(size_t)i] = val[i] > 0.0 ? 1 : 0;
    }
    for (int64_t i = 0; i < n; ++i) {
        if (nchild[(size_t)i] || (state[(size_t)i] & 3) != 1) continue;
        int64_t k = i;
        for (;;) {
            state[(size_t)k] |= 2;
            const int64_t p = parent[k];
            if (p < 0) break;
            val[p] += val[k];
            if (--nchild[(size_t)p] != 0 || !(state[(size_t)p] & 1)) break;
            k = p;
        }
    }
    for (int64_t i = 0; i < n; ++i)
        if (!(state[(size_t)i] & 2)) val[i] = 0.0;
}

// accum, after the gather: the forest of the m pass-through pairs of all bands (nodes are seam cells, node ids below nspace) is
// solved like mhip_band_forest_solve; the values of the nodes base_top + [0, W) go to top[], those of base_bot + [0, W) to bot[]
// (the neighbours' edge rows as this band needs them as known sources; cells that are not nodes keep what the rows hold).
extern "C" int mhip_band_accum_solve(int64_t nspace, int64_t m, const void *pairs, int64_t W, int64_t base_top, double *top, int64_t base_bot,
                                     double *bot)
{
    MH_ARG(nspace >= 0 && nspace < (int64_t(1) << 31) && m >= 0 && W >= 0 && (m == 0 || pairs), "band_accum_solve(...)");
    const BandAccumPair *pr = (const BandAccumPair *)pairs;
    if (m == 0) return MHIP_OK;
    std::vector<int32_t> id((size_t)nspace, -1);
    std::vector<int64_t> orig;
    std::vector<int32_t> par;
    std::vector<double> val;
    orig.reserve((size_t)(2 * m)); par.reserve((size_t)(2 * m)); val.reserve((size_t)(2 * m));
    auto node = [&](int64_t o, double own) {
        int32_t &slot = id[(size_t)o];
        if (slot < 0) {
            slot = (int32_t)orig.size();
            orig.push_back(o); par.push_back(-1); val.push_back(own);
        } else
            val[(size_t)slot] = own;
        return slot;
    };
    for (int64_t k = 0; k < m; ++k) {
        MH_ARG(pr[k].child >= 0 && pr[k].child < nspace && pr[k].parent >= 0 && pr[k].parent < nspace, "band_accum_solve: node out of range");
        const int32_t c = node(pr[k].child, pr[k].own_child);
        const int32_t p = node(pr[k].parent, pr[k].own_parent);
        par[(size_t)c] = p;
    }
    const int64_t nn = (int64_t)orig.size();
    forest_solve(nn, par.data(), val.data());
    for (int64_t i = 0; i < nn; ++i) {
        const int64_t o = orig[(size_t)i];
        if (top && o >= base_top && o < base_top + W) top[o - base_top] = val[(size_t)i];
        if (bot && o >= base_bot && o < base_bot + W) bot[o - base_bot] = val[(size_t)i];
    }
    return MHIP_OK;
}

// label, before the gather: one (mine, theirs) pair per run of equal (halo label, neighbour's label) along a halo row, and the
// phantoms of the row: local components with cells in the halo row but none in the adjacent owned row `edge`.
extern "C" int mhip_band_label_pairs(int64_t W, const int32_t *halo, const int32_t *edge, const int32_t *nbr, int64_t key_mine, int64_t key_nbr,
                                     int64_t *ea, int64_t *eb, int64_t *npairs, int64_t *ph, int64_t *nph)
{
    MH_ARG(W >= 0 && npairs && nph && (W == 0 || (halo && edge && nbr && ea && eb && ph)), "band_label_pairs(...)");
    int64_t k = 0;
    std::vector<int32_t> hv, ev;
    for (int64_t j = 0; j < W; ++j) {
        const int32_t h = halo[j], t = nbr[j], e = edge[j];
        if (h > 0 && t > 0 && (j == 0 || h != halo[j - 1] || t != nbr[j - 1])) {
            ea[k] = key_mine | (int64_t)h;
            eb[k] = key_nbr | (int64_t)t;
            ++k;
        }
        if (h > 0 && (j == 0 || h != halo[j - 1])) hv.push_back(h);
        if (e > 0 && (j == 0 || e != edge[j - 1])) ev.push_back(e);
    }
    *npairs = k;
    std::sort(hv.begin(), hv.end());
    hv.erase(std::unique(hv.begin(), hv.end()), hv.end());
    std::sort(ev.begin(), ev.end());
    int64_t q = 0;
    size_t i = 0;
    for (int32_t h : hv) {
        while (i < ev.size() && ev[i] < h) ++i;
        if (i == ev.size() || ev[i] != h) ph[q++] = h;
    }
    *nph = q;
    return MHIP_OK;
}

// (key, origin) pairs sorted by key: LSD radix sort, 11 bits per pass, as many passes as the largest key needs
static void radix_sort_pairs(std::vector<uint64_t> &key, std::vector<uint32_t> &org)
{
    const size_t n = key.size();
    uint64_t all = 0;
    for (uint64_t k : key) all |= k;
    std::vector<uint64_t> key2(n);
    std::vector<uint32_t> org2(n);
    for (int shift = 0; shift < 64 && (all >> shift); shift += 11) {
        size_t count[2049] = {0};
        for (size_t i = 0; i < n; ++i) ++count[((key[i] >> shift) & 2047) + 1];
        for (int d = 0; d < 2048; ++d) count[d + 1] += count[d];
        for (size_t i = 0; i < n; ++i) {
            const size_t pos = count[(key[i] >> shift) & 2047]++;
            key2[pos] = key[i];
            org2[pos] = org[i];
        }
        key.swap(key2);
        org.swap(org2);
    }
}

// label, after the gather: the classes of the seam pairs of all bands and, from them, the numbering of EVERY band.
//   nloc[r]      local labels of band r (1 .. nloc[r])
//   EA, EB       the m gathered pairs; keys (band << 32) | local label
//   PH           the gathered phantoms (keys): labels without an owned cell, never representatives
// A class is represented by its smallest real (rank, label) member and gets that member's new label; the other members and all
// phantoms are DROPPED from their band's numbering: band r's kept label l becomes offsets[r] + l - #(dropped labels of r below l).
//   offsets[R + 1]             offsets[R] = the global number of labels
//   dropped / target [*ndrop]  band `me`: its dropped labels (ascending) and the global labels they become (0: none)
//   shared [*nshared]          global labels with cells in more than one band (ascending): their records need a merge
extern "C" int mhip_band_label_merge(int32_t R, int32_t me, const int64_t *nloc, int64_t m, const int64_t *EA, const int64_t *EB, int64_t nph,
                                     const int64_t *PH, int64_t *offsets, int32_t *dropped, int32_t *target, int64_t *ndrop, int64_t *shared,
                                     int64_t *nshared)
{
    MH_ARG(R >= 1 && me >= 0 && me < R && nloc && m >= 0 && nph >= 0 && (m == 0 || (EA && EB)) && (nph == 0 || PH) && offsets && ndrop && nshared,
           "band_label_merge(...)");
    const int64_t tot = 2 * m + nph;
    MH_ARG(tot < (int64_t(1) << 31), "band_label_merge: too many pairs");
    MH_ARG(tot == 0 || (dropped && target && shared), "band_label_merge: output arrays");
    std::vector<uint64_t> key((size_t)tot);
    std::vector<uint32_t> org((size_t)tot);
    uint64_t maxlab = 1;
    for (int64_t i = 0; i < tot; ++i) {
        const int64_t k = i < m ? EA[i] : i < 2 * m ? EB[i - m] : PH[i - 2 * m];
        const int64_t r = k >> 32, l = k & 0xffffffffll;
        MH_ARG(r >= 0 && r < R && l >= 1 && l <= nloc[r], "band_label_merge: key out of range");
        if ((uint64_t)l > maxlab) maxlab = (uint64_t)l;
    }
    const int lb = 64 - __builtin_clzll(maxlab);
    for (int64_t i = 0; i < tot; ++i) {
        const int64_t k = i < m ? EA[i] : i < 2 * m ? EB[i - m] : PH[i - 2 * m];
        key[(size_t)i] = ((uint64_t)(k >> 32) << lb) | (uint64_t)(k & 0xffffffffll);
        org[(size_t)i] = (uint32_t)i;
    }
    radix_sort_pairs(key, org);
    // the nodes: distinct keys in (band, label) order
    std::vector<int32_t> node_of((size_t)tot), nrank, nlab;
    std::vector<uint8_t> phantom;
    for (int64_t i = 0; i < tot; ++i) {
        if (i == 0 || key[(size_t)i] != key[(size_t)i - 1]) {
            nrank.push_back((int32_t)(key[(size_t)i] >> lb));
            nlab.push_back((int32_t)(key[(size_t)i] & ((uint64_t(1) << lb) - 1)));
            phantom.push_back(0);
        }
        node_of[org[(size_t)i]] = (int32_t)nrank.size() - 1;
        if ((int64_t)org[(size_t)i] >= 2 * m) phantom.back() = 1;
    }
    const int64_t nn = (int64_t)nrank.size();
    std::vector<int32_t> cls((size_t)nn);
    for (int64_t i = 0; i < nn; ++i) cls[(size_t)i] = (int32_t)i;
    auto find = [&](int32_t x) {
        while (cls[(size_t)x] != x) {
            cls[(size_t)x] = cls[(size_t)cls[(size_t)x]];
            x = cls[(size_t)x];
        }
        return x;
    };
    for (int64_t k = 0; k < m; ++k) {
        const int32_t x = find(node_of[(size_t)k]), y = find(node_of[(size_t)(m + k)]);
        if (x != y) cls[(size_t)(x < y ? y : x)] = x < y ? x : y;
    }
    // representative of a class = its first real node in (band, label) order
    std::vector<int32_t> rep((size_t)nn, -1), first_rank((size_t)nn, -1);
    std::vector<uint8_t> multi((size_t)nn, 0);
    std::vector<int64_t> ndrop_r((size_t)R, 0);
    for (int64_t i = 0; i < nn; ++i) {
        const int32_t c = find((int32_t)i);
        cls[(size_t)i] = c;
        if (!phantom[(size_t)i]) {
            if (rep[(size_t)c] < 0) rep[(size_t)c] = (int32_t)i;
            if (first_rank[(size_t)c] < 0) first_rank[(size_t)c] = nrank[(size_t)i];
            else if (first_rank[(size_t)c] != nrank[(size_t)i]) multi[(size_t)c] = 1;
        }
        if (rep[(size_t)c] != (int32_t)i) ++ndrop_r[(size_t)nrank[(size_t)i]];
    }
    offsets[0] = 0;
    for (int32_t r = 0; r < R; ++r) {
        MH_ARG(nloc[r] >= ndrop_r[(size_t)r], "band_label_merge: more dropped labels than labels");
        offsets[r + 1] = offsets[r] + nloc[r] - ndrop_r[(size_t)r];
    }
    // the new label of every class, from its representative's band's numbering
    std::vector<int64_t> class_label((size_t)nn, 0);
    int32_t cur = -1;
    int64_t below = 0;
    for (int64_t i = 0; i < nn; ++i) {
        if (nrank[(size_t)i] != cur) { cur = nrank[(size_t)i]; below = 0; }
        const int32_t c = cls[(size_t)i];
        if (rep[(size_t)c] == (int32_t)i) class_label[(size_t)c] = offsets[cur] + nlab[(size_t)i] - below;
        else ++below;
    }
    int64_t nd = 0, ns = 0;
    for (int64_t i = 0; i < nn; ++i) {
        const int32_t c = cls[(size_t)i];
        if (nrank[(size_t)i] == me && rep[(size_t)c] != (int32_t)i) {
            dropped[nd] = nlab[(size_t)i];
            target[nd] = (int32_t)class_label[(size_t)c];
            ++nd;
        }
        if (c == (int32_t)i && multi[(size_t)c]) shared[ns++] = class_label[(size_t)c];
    }
    std::sort(shared, shared + ns);
    *ndrop = nd;
    *nshared = ns;
    return MHIP_OK;
}

// watershed, before the gather.  mine[e], e = side * W + column: the band's edge rows after the local pass (a label, 0, or the
// pseudo label -(1 + k) of the halo cell its path leaves the band through; k >= W: bottom halo).  up / dn: the neighbours' edge
// rows next to the band (null: none).  An edge cell with a pseudo label is PUBLISHED -- (node, entry) with entry = the label its
// path ends at right there in the neighbour's row, or -(1 + node) of the neighbour's cell it continues in -- when a neighbour's
// path points at it or when its own target is unresolved as well.  Node of (band, side, column) = (2 * band + side) * W + column.
extern "C" int mhip_band_ws_publish(int64_t W, int32_t me, const int32_t *mine, const int32_t *up, const int32_t *dn, int64_t *N, int64_t *V,
                                    int64_t *n)
{
    MH_ARG(W >= 0 && me >= 0 && n && (W == 0 || (mine && N && V)), "band_ws_publish(...)");
    std::vector<uint8_t> pointed((size_t)(2 * W), 0);
    for (int64_t j = 0; j < W; ++j) {
        if (up && up[j] < 0) {           // their pseudo labels >= W point at their bottom halo = my first row
            const int64_t t = -(int64_t)up[j] - 1;
            MH_ARG(t < 2 * W, "band_ws_publish: pseudo label out of range");
            if (t >= W) pointed[(size_t)(t - W)] = 1;
        }
        if (dn && dn[j] < 0) {           // their pseudo labels < W point at their top halo = my last row
            const int64_t t = -(int64_t)dn[j] - 1;
            MH_ARG(t < 2 * W, "band_ws_publish: pseudo label out of range");
            if (t < W) pointed[(size_t)(W + t)] = 1;
        }
    }
    int64_t k = 0;
    for (int64_t e = 0; e < 2 * W; ++e) {
        if (mine[e] >= 0) continue;
        const int64_t idx = -(int64_t)mine[e] - 1;
        MH_ARG(idx < 2 * W, "band_ws_publish: pseudo label out of range");
        const bool to_up = idx < W;
        const int64_t tgt_node = to_up ? (2 * (int64_t)(me - 1) + 1) * W + idx : (2 * (int64_t)(me + 1)) * W + (idx - W);
        const int64_t tgt_val = to_up ? (up ? up[idx] : 0) : (dn ? dn[idx - W] : 0);
        if (pointed[(size_t)e] || tgt_val < 0) {
            N[k] = 2 * (int64_t)me * W + e;
            V[k] = tgt_val >= 0 ? tgt_val : -(tgt_node + 1);
            ++k;
        }
    }
    *n = k;
    return MHIP_OK;
}

// watershed, after the gather: the published (node, entry) pairs of all bands are followed to their ends (mhip_band_ws_resolve)
// and the look-up table of this band's halo cells is written: lut[side * W + k] = the label a path through halo cell k ends at
// (the neighbour's row where it holds a label, else what its published chain resolved to, 0: nowhere).
extern "C" int mhip_band_ws_lut(int64_t W, int32_t me, int64_t n, const int64_t *N, const int64_t *V, const int32_t *up, const int32_t *dn,
                                int32_t *lut)
{
    MH_ARG(W >= 0 && me >= 0 && n >= 0 && (n == 0 || (N && V)) && (W == 0 || lut), "band_ws_lut(...)");
    std::vector<int64_t> node(N, N + n), val(V, V + n);
    bool sorted = true;
    for (int64_t i = 1; i < n; ++i) sorted &= node[(size_t)i - 1] < node[(size_t)i];
    if (!sorted) {
        std::vector<int64_t> order((size_t)n);
        for (int64_t i = 0; i < n; ++i) order[(size_t)i] = i;
        std::sort(order.begin(), order.end(), [&](int64_t a, int64_t b) { return N[a] < N[b]; });
        for (int64_t i = 0; i < n; ++i) { node[(size_t)i] = N[order[(size_t)i]]; val[(size_t)i] = V[order[(size_t)i]]; }
    }
    auto lookup = [&](int64_t t) -> int64_t {
        const auto it = std::lower_bound(node.begin(), node.end(), t);
        return (it != node.end() && *it == t) ? (int64_t)(it - node.begin()) : -1;
    };
    for (int64_t i = 0; i < n; ++i)
        if (val[(size_t)i] < 0) {
            const int64_t pos = lookup(-val[(size_t)i] - 1);
            val[(size_t)i] = pos >= 0 ? -(pos + 1) : 0;           // (a target nobody published: nobody needs it either)
        }
    if (n) {
        const int rc = mhip_band_ws_resolve(n, val.data());
        if (rc != MHIP_OK) return rc;
    }
    for (int side = 0; side < 2; ++side) {
        const int32_t *row = side ? dn : up;
        const int64_t base = side ? (2 * (int64_t)(me + 1)) * W : (2 * (int64_t)(me - 1) + 1) * W;
        for (int64_t k = 0; k < W; ++k) {
            int32_t r = row ? row[k] : 0;
            if (r < 0) {
                const int64_t pos = lookup(base + k);
                r = pos >= 0 ? (int32_t)val[(size_t)pos] : 0;
            }
            lut[side * W + k] = r;
        }
    }
    return MHIP_OK;
}

// records of the labels that live in several bands, merged over the bands' partial records in band (= raster) order.
//   kind 0: label_stats records {min, max, sum, count}; sums are added in band order
//   kind 2: label_max_index records {value, row, col}: the larger value wins, the earlier band on ties; row < 0 = no cell here
//   kind 3: label_min_index records: the smaller value wins
struct BandStat { double mn, mx, sum; int64_t count; };
struct BandIndex { double value; int64_t row, col; };
extern "C" int mhip_band_merge_records(int32_t kind, int32_t R, int64_t n, const void *const *parts, void *out)
{
    MH_ARG((kind == 0 || kind == 2 || kind == 3) && R >= 1 && n >= 0 && parts && (n == 0 || out), "band_merge_records(kind, R, n, parts, out)");
    for (int32_t r = 0; r < R; ++r) MH_ARG(n == 0 || parts[r], "band_merge_records: null part");
    if (kind == 0) {
        BandStat *m = (BandStat *)out;
        for (int64_t i = 0; i < n; ++i) m[i] = ((const BandStat *)parts[0])[i];
        for (int32_t r = 1; r < R; ++r) {
            const BandStat *p = (const BandStat *)parts[r];
            for (int64_t i = 0; i < n; ++i) {
                m[i].mn = (p[i].mn < m[i].mn || p[i].mn != p[i].mn) ? p[i].mn : m[i].mn;      // (np.minimum / np.maximum: a NaN stays)
                m[i].mx = (p[i].mx > m[i].mx || p[i].mx != p[i].mx) ? p[i].mx : m[i].mx;
                m[i].sum = m[i].sum + p[i].sum;
                m[i].count = m[i].count + p[i].count;
            }
        }
    } else {
        BandIndex *m = (BandIndex *)out;
        for (int64_t i = 0; i < n; ++i) m[i] = ((const BandIndex *)parts[0])[i];
        for (int32_t r = 1; r < R; ++r) {
            const BandIndex *p = (const BandIndex *)parts[r];
            for (int64_t i = 0; i < n; ++i) {
                bool better = kind == 2 ? p[i].value > m[i].value : p[i].value < m[i].value;
                better |= m[i].row < 0 && p[i].row >= 0;
                better &= p[i].row >= 0;
                if (better) m[i] = p[i];
            }
        }
    }
    return MHIP_OK;
}

// ---- several bands of ONE process (host threads): their rendezvous in the library ---------------------------------------------
// The 1- and 2-GPU shapes of a raster beyond 2**31 cells run k bands per process, a host thread each (distributed.ThreadComm /
// HybridComm).  Their halo exchanges and votes are rendezvous of those threads -- 40 to 60 per step -- and as Python objects
// (queue.Queue, threading.Barrier: a handful of GIL hand-overs per thread and wait) each cost ~0.1 ms with the device idle.  Here a
// thread blocks inside a ctypes call (no GIL), spins for a few microseconds first and sleeps on a condition variable after that.
// A wait that lasts longer than `timeout_ms` returns MHIP_ECOMM: a band that died must not leave the others waiting for ever.
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstring>
#include <mutex>
#include <thread>

namespace {
struct ThreadGroup {
    int n = 0;
    std::mutex mu;
    std::condition_variable cv;
    int arrived = 0;
    std::atomic<unsigned long long> generation{0};
    bool broken = false;
    std::vector<double> value;
    std::vector<const void *> up, down;     // the rows rank r offers to r - 1 / r + 1 in the current exchange
    std::vector<int64_t> meta_up, meta_down;      // eight words each: [0] bytes (-1: nothing), the rest is the callers' (dtype, shape)
    explicit ThreadGroup(int n_) : n(n_), value(n_), up(n_), down(n_), meta_up((size_t)n_ * 8), meta_down((size_t)n_ * 8) {}

    int barrier(int timeout_ms)
    {
        std::unique_lock<std::mutex> lk(mu);
        if (broken) return MHIP_ECOMM;
        const unsigned long long gen = generation.load(std::memory_order_relaxed);
        if (++arrived == n) {
            arrived = 0;
            generation.store(gen + 1, std::memory_order_release);
            lk.unlock();
            cv.notify_all();
            return MHIP_OK;
        }
        lk.unlock();
        const auto t0 = std::chrono::steady_clock::now();
        for (;;) {      // the others are usually microseconds away
            if (generation.load(std::memory_order_acquire) != gen) return MHIP_OK;
            if (std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(50)) break;
        }
        lk.lock();
        const bool ok = cv.wait_for(lk, std::chrono::milliseconds(timeout_ms),
                                    [&] { return generation.load(std::memory_order_acquire) != gen || broken; });
        if (!ok || (broken && generation.load(std::memory_order_acquire) == gen)) {
            broken = true;      // nobody gets past a barrier of this group any more
            lk.unlock();
            cv.notify_all();
            return MHIP_ECOMM;
        }
        return MHIP_OK;
    }
};
#define MH_TG(expr)                                                                                                         \
    do {                                                                                                                    \
        if ((expr) != MHIP_OK) {                                                                                            \
            mh::set_error("a band thread of this process did not arrive within the timeout (or the group is broken)");     \
            return MHIP_ECOMM;                                                                                              \
        }                                                                                                                   \
    } while (0)
}  // namespace

extern "C" int mhip_tg_create(int32_t n, void **group)
{
    MH_ARG(n >= 1 && group, "tg_create(n, &group)");
    *group = new ThreadGroup(n);
    return MHIP_OK;
}

extern "C" int mhip_tg_destroy(void *group)
{
    delete static_cast<ThreadGroup *>(group);
    return MHIP_OK;
}

extern "C" int mhip_tg_barrier(void *group, int32_t timeout_ms)
{
    MH_ARG(group && timeout_ms > 0, "tg_barrier(group, timeout_ms)");
    MH_TG(static_cast<ThreadGroup *>(group)->barrier(timeout_ms));
    return MHIP_OK;
}

/* the maximum of every thread's value (all threads call it; rank in [0, n)) */
extern "C" int mhip_tg_allreduce_max(void *group, int32_t rank, double value, double *out, int32_t timeout_ms)
{
    ThreadGroup *g = static_cast<ThreadGroup *>(group);
    MH_ARG(g && out && rank >= 0 && rank < g->n && timeout_ms > 0, "tg_allreduce_max(group, rank, value, &out, timeout_ms)");
    g->value[rank] = value;
    MH_TG(g->barrier(timeout_ms));
    double m = g->value[0];
    for (int r = 1; r < g->n; ++r) m = g->value[r] > m ? g->value[r] : m;      // (a NaN vote loses: the callers vote 0 / 1 / 2)
    *out = m;
    MH_TG(g->barrier(timeout_ms));      // everybody has read: the slots may be written again
    return MHIP_OK;
}

/* Neighbour exchange between the threads in two calls, because a thread does not know what its neighbours will hand it (halo rows
 * have its own rows' shape; the raster writer passes blocks of rows of any height).  offer: rank r leaves `to_up` (for r - 1) and
 * `to_down` (for r + 1) with eight words of description each (meta[0] = bytes, the rest is the caller's: item size, kind, shape);
 * NULL = nothing.  It learns what r - 1 offers downwards (meta_from_up) and r + 1 upwards (meta_from_down); [0] = -1: nothing.
 * take: the bytes into buffers of at least those sizes (NULL where nothing comes); the offered rows must stay valid until it returns. */
extern "C" int mhip_tg_offer(void *group, int32_t rank, const void *to_up, const int64_t *meta_up, const void *to_down, const int64_t *meta_down,
                             int64_t *meta_from_up, int64_t *meta_from_down, int32_t timeout_ms)
{
    ThreadGroup *g = static_cast<ThreadGroup *>(group);
    MH_ARG(g && rank >= 0 && rank < g->n && meta_from_up && meta_from_down && timeout_ms > 0 && (!to_up || meta_up) && (!to_down || meta_down),
           "tg_offer(group, rank, to_up, meta_up[8], to_down, meta_down[8], meta_from_up[8], meta_from_down[8], timeout_ms)");
    g->up[rank] = to_up;
    g->down[rank] = to_down;
    for (int k = 0; k < 8; ++k) {
        g->meta_up[(size_t)rank * 8 + k] = to_up ? meta_up[k] : -1;
        g->meta_down[(size_t)rank * 8 + k] = to_down ? meta_down[k] : -1;
    }
    MH_TG(g->barrier(timeout_ms));
    for (int k = 0; k < 8; ++k) {
        meta_from_up[k] = rank > 0 ? g->meta_down[(size_t)(rank - 1) * 8 + k] : -1;
        meta_from_down[k] = rank < g->n - 1 ? g->meta_up[(size_t)(rank + 1) * 8 + k] : -1;
    }
    return MHIP_OK;
}

extern "C" int mhip_tg_take(void *group, int32_t rank, void *from_up, void *from_down, int32_t timeout_ms)
{
    ThreadGroup *g = static_cast<ThreadGroup *>(group);
    MH_ARG(g && rank >= 0 && rank < g->n && timeout_ms > 0, "tg_take(group, rank, from_up, from_down, timeout_ms)");
    int rc = MHIP_OK;
    if (rank > 0 && g->meta_down[(size_t)(rank - 1) * 8] >= 0) {
        if (!from_up) rc = MHIP_EINVAL;
        else memcpy(from_up, g->down[rank - 1], (size_t)g->meta_down[(size_t)(rank - 1) * 8]);
    }
    if (rank < g->n - 1 && g->meta_up[(size_t)(rank + 1) * 8] >= 0) {
        if (!from_down) rc = MHIP_EINVAL;
        else memcpy(from_down, g->up[rank + 1], (size_t)g->meta_up[(size_t)(rank + 1) * 8]);
    }
    MH_TG(g->barrier(timeout_ms));      // everybody has copied: the offered rows may go
    if (rc != MHIP_OK) mh::set_error("tg_take: no buffer for a row a neighbour offers");
    return rc;
}

// ---- ranks that are PROCESSES of one host: their barrier in a shared-memory segment -------------------------------------------
// The control plane of a band chain is 50-60 small collectives per step (votes, neighbour rows of labels / accumulation / watersheds,
// all-gathers of seam pairs); between the processes of ONE node they went through gloo (TCP on the loopback interface, 0.15-0.5 ms
// each: 4 processes x 1 band of 32768^2 on one GPU 116 ms against 105 ms for 4 bands in one process).  distributed.ShmComm keeps the
// payloads in a POSIX shared-memory segment; what it needs from here is the barrier: two 64-bit words at `base` (arrivals, generation),
// atomics of the hardware (the segment is mapped by every process), a short spin and then short sleeps.  timeout: MHIP_ECOMM.
extern "C" int mhip_shm_barrier(void *base, int32_t n, int32_t timeout_ms)
{
    MH_ARG(base && n >= 1 && timeout_ms > 0 && ((uintptr_t)base & 7u) == 0, "shm_barrier(base, n, timeout_ms)");
    unsigned long long *arrived = static_cast<unsigned long long *>(base), *generation = arrived + 1;
    const unsigned long long gen = __atomic_load_n(generation, __ATOMIC_ACQUIRE);
    if (__atomic_add_fetch(arrived, 1ull, __ATOMIC_ACQ_REL) == (unsigned long long)n) {
        __atomic_store_n(arrived, 0ull, __ATOMIC_RELAXED);
        __atomic_store_n(generation, gen + 1, __ATOMIC_RELEASE);
        return MHIP_OK;
    }
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
        if (__atomic_load_n(generation, __ATOMIC_ACQUIRE) != gen) return MHIP_OK;
        const auto dt = std::chrono::steady_clock::now() - t0;
        if (dt > std::chrono::milliseconds(timeout_ms)) break;
        if (dt > std::chrono::microseconds(200)) std::this_thread::sleep_for(std::chrono::microseconds(20));      // (the others are usually microseconds away)
    }
    mh::set_error("a rank of this node did not arrive at the shared-memory barrier within the timeout");
    return MHIP_ECOMM;
}
