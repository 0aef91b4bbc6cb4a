// rain.hip -- rain events on the stream network (host C++: the work is O(#nodes) per event on a forest).
//
// Reference: malstroem/network.py:75-129 (Network._calc_node / _calc_stream_tree / rain_event) and rain.py:48-87
// (RainTool evaluates every requested event in a Python loop over the whole network).  Here all events of a call are
// evaluated in one pass over the forest, in the reference's own order and with its arithmetic:
//   rainv  = wshed_area * mm * 0.001                      (left to right)
//   inflow = sum(spillv of the upstream nodes, in the order they were added), starting from 0
//   total  = rainv + inflow;  v = min(total, bspot_vol);  spillv = max(0, total - bspot_vol)
//   pctv   = None when bspot_vol is falsy, else 100 * v / bspot_vol
// so the results are bit-identical float64 values.  Only nodes below a root (downstream id None) are evaluated, like in the
// reference; `order` returns the sequence in which the reference would have evaluated (and listed) them.
#include "common.hpp"
#include <cmath>
#include <vector>

extern "C" int mhip_rain_events(int64_t n, const int64_t *down_index, const double *wshed_area, const double *bspot_vol, int32_t nevents,
                                const double *mmrain, double *rainv, double *spillv, double *v, double *pctv, int64_t *order,
                                int64_t *ncomputed)
{
    MH_ARG(n >= 0 && nevents >= 0 && (n == 0 || (down_index && wshed_area && bspot_vol)) && (nevents == 0 || mmrain) && ncomputed,
           "rain_events(n, down_index, wshed_area, bspot_vol, nevents, mmrain, ...)");
    MH_ARG(n == 0 || nevents == 0 || (rainv && spillv && v && pctv), "rain_events output arrays");
    // upstream lists in insertion order (CSR)
    std::vector<int64_t> start((size_t)n + 1, 0), kids((size_t)n), roots;
    for (int64_t i = 0; i < n; ++i) {
        const int64_t d = down_index[i];
        MH_ARG(d >= -2 && d < n, "rain_events: downstream index out of range");
        if (d >= 0) ++start[(size_t)d + 1];
        else if (d == -1) roots.push_back(i);
    }
    for (int64_t i = 0; i < n; ++i) start[(size_t)i + 1] += start[(size_t)i];
    {
        std::vector<int64_t> fill(start.begin(), start.end() - 1);
        for (int64_t i = 0; i < n; ++i)
            if (down_index[i] >= 0) kids[(size_t)fill[(size_t)down_index[i]]++] = i;
    }
    // evaluation order of network.py:104-113: per root a LIFO pre-order, evaluated back to front
    std::vector<int64_t> seq, tree, stack;
    std::vector<char> seen((size_t)n, 0);
    seq.reserve((size_t)n);
    for (int64_t root : roots) {
        tree.clear();
        stack.assign(1, root);
        while (!stack.empty()) {
            const int64_t x = stack.back();
            stack.pop_back();
            if (seen[(size_t)x]) {
                mh::set_error("rain_events: the node graph is not a forest (node %lld is reached twice)", (long long)x);
                return MHIP_EINVAL;
            }
            seen[(size_t)x] = 1;
            tree.push_back(x);
            for (int64_t k = start[(size_t)x]; k < start[(size_t)x + 1]; ++k) stack.push_back(kids[(size_t)k]);
        }
        for (size_t k = tree.size(); k-- > 0;) seq.push_back(tree[k]);
    }
    *ncomputed = (int64_t)seq.size();
    if (order) {
        for (int64_t i = 0; i < n; ++i) order[i] = i < (int64_t)seq.size() ? seq[(size_t)i] : -1;
    }
    const double nan = std::nan("");
    for (int32_t e = 0; e < nevents; ++e) {
        double *rv = rainv + (size_t)e * n, *sv = spillv + (size_t)e * n, *vv = v + (size_t)e * n, *pv = pctv + (size_t)e * n;
        for (int64_t i = 0; i < n; ++i) rv[i] = sv[i] = vv[i] = pv[i] = nan;
        const double mm = mmrain[e];
        for (int64_t x : seq) {
            const double wshed = wshed_area[x] * mm * 0.001;
            const double capacity = bspot_vol[x];
            double upstream = 0.0;
            for (int64_t k = start[(size_t)x]; k < start[(size_t)x + 1]; ++k) upstream += sv[kids[(size_t)k]];
            const double total = wshed + upstream;
            const double filled = capacity < total ? capacity : total;   // Python min(total, capacity)
            const double over = total - capacity;
            rv[x] = wshed;
            sv[x] = over > 0 ? over : 0.0;                                 // Python max(0, total - capacity)
            vv[x] = filled;
            pv[x] = (capacity == 0.0) ? nan : 100.0 * filled / capacity;   // `not bspot_capacity`
        }
    }
    return MHIP_OK;
}
