"""The host sections of the band protocol (csrc/bandsolve.hip: mhip_band_accum_pairs / accum_solve / label_pairs / label_merge /
ws_publish / ws_lut) against their plain NumPy statements on random seam rows.  Host code: no GPU needed."""
import ctypes

import numpy as np
import pytest

from malstroem_amd import _lib
from malstroem_amd.distributed import ACCUM_PAIR_DTYPE

pytestmark = pytest.mark.skipif(not _lib.LIB_PATH.exists(), reason="library not built")


def _c64():
    return ctypes.c_int64(0)


# ---- accumulation ---------------------------------------------------------------------------------------------------------------
def np_accum_pairs(W, exit_half, nbr, own_edge, cbase, pbase):
    k = np.flatnonzero(exit_half >= 0)
    e = exit_half[k].astype(np.int64)
    return cbase + k, pbase + e, nbr[k], own_edge[e]


def np_accum_solve(C, P, OC, OP, W, base_top, top, base_bot, bot):
    m = C.size
    if not m:
        return
    nodes, inv = np.unique(np.concatenate([C, P]), return_inverse=True)
    val = np.zeros(nodes.size, np.float64)
    val[inv[:m]] = OC
    val[inv[m:]] = OP
    parent = np.full(nodes.size, -1, np.int64)
    parent[inv[:m]] = inv[m:]
    # the forest solved the slow way: a node is final when it is known and all its children are final
    known = val > 0
    final = np.zeros(nodes.size, bool)
    total = val.copy()
    nchild = np.bincount(parent[parent >= 0], minlength=nodes.size)
    done_children = np.zeros(nodes.size, np.int64)
    changed = True
    while changed:
        changed = False
        for i in range(nodes.size):
            if not final[i] and known[i] and done_children[i] == nchild[i]:
                final[i] = True
                changed = True
                if parent[i] >= 0:
                    total[parent[i]] += total[i]
                    done_children[parent[i]] += 1
    total[~final] = 0
    for row, base in ((top, base_top), (bot, base_bot)):
        if row is not None:
            lo, hi = np.searchsorted(nodes, [base, base + W])
            row[nodes[lo:hi] - base] = total[lo:hi]


@pytest.mark.parametrize("seed", range(6))
def test_accum_sections(seed):
    rng = np.random.default_rng(seed)
    W, R = int(rng.integers(1, 60)), int(rng.integers(2, 5))
    own = [rng.integers(0, 50, 2 * W).astype(np.float64) * (rng.random(2 * W) < 0.85) for _ in range(R)]     # 0 = unknown contribution
    parts, tops, bots = [], {}, {}
    for me in range(R):
        exit_map = np.where(rng.random(2 * W) < 0.4, rng.integers(0, 2 * W, 2 * W), -1).astype(np.int32)
        got = np.empty(2 * W, ACCUM_PAIR_DTYPE)
        want = [[], [], [], []]
        k = 0
        for half, r_nbr, side in ((exit_map[:W], me - 1, 1), (exit_map[W:], me + 1, 0)):
            if r_nbr < 0 or r_nbr >= R:
                continue
            nbr = np.ascontiguousarray(own[r_nbr][side * W:(side + 1) * W])
            base = (2 * r_nbr + side) * W
            n = _c64()
            _lib.call("mhip_band_accum_pairs", _lib.i64(W), _lib.ptr(half), _lib.ptr(nbr), _lib.ptr(own[me]), _lib.i64(base), _lib.i64(2 * me * W),
                      _lib.ptr(got[k:]), ctypes.byref(n))
            k += n.value
            for lst, a in zip(want, np_accum_pairs(W, half, nbr, own[me], base, 2 * me * W)):
                lst.append(a)
        for name, w in zip(("child", "parent", "own_child", "own_parent"), want):
            w = np.concatenate(w) if w else np.zeros(0)
            assert np.array_equal(got[name][:k], w)
        parts.append(got[:k].copy())
    pairs = np.concatenate(parts)
    C, P, OC, OP = (pairs[n] for n in ("child", "parent", "own_child", "own_parent"))
    # one child per node like in the protocol (a cell's flux leaves through ONE edge cell): drop the repeated children
    _, firsts = np.unique(C, return_index=True)
    firsts.sort()
    pairs = np.ascontiguousarray(pairs[firsts])
    C, P, OC, OP = (pairs[n].astype(t) for n, t in (("child", np.int64), ("parent", np.int64), ("own_child", np.float64), ("own_parent", np.float64)))
    for me in range(R):
        top0 = own[me - 1][W:].copy() if me > 0 else None
        bot0 = own[me + 1][:W].copy() if me + 1 < R else None
        got_t, got_b = (None if top0 is None else top0.copy()), (None if bot0 is None else bot0.copy())
        _lib.call("mhip_band_accum_solve", _lib.i64(2 * R * W), _lib.i64(pairs.size), _lib.ptr(pairs), _lib.i64(W),
                  _lib.i64((2 * (me - 1) + 1) * W), _lib.ptr(got_t) if got_t is not None else None,
                  _lib.i64(2 * (me + 1) * W), _lib.ptr(got_b) if got_b is not None else None)
        np_accum_solve(C, P, OC, OP, W, (2 * (me - 1) + 1) * W, top0, 2 * (me + 1) * W, bot0)
        for g, w in ((got_t, top0), (got_b, bot0)):
            if w is not None:
                assert np.array_equal(g, w)


# ---- labels ---------------------------------------------------------------------------------------------------------------------
def np_label_pairs(W, halo, edge, nbr, me, r_nbr):
    key = lambda r, lab: (np.int64(r) << 32) | lab.astype(np.int64)

    def run_values(row):
        keep = row > 0
        if row.size > 1:
            keep[1:] &= row[1:] != row[:-1]
        return np.unique(row[keep])
    m = (halo > 0) & (nbr > 0)
    if W > 1:
        m[1:] &= (halo[1:] != halo[:-1]) | (nbr[1:] != nbr[:-1])
    return key(me, halo[m]), key(r_nbr, nbr[m]), np.setdiff1d(run_values(halo), run_values(edge), assume_unique=True)


def np_label_merge(R, me, nlocs, EA, EB, PH):
    nodes, inv = np.unique(np.concatenate([EA, EB, PH]), return_inverse=True)
    nn = nodes.size
    root = np.arange(nn)
    ia, ib = inv[:EA.size], inv[EA.size:2 * EA.size]
    while True:                         # label propagation to the smallest node of the class
        new = root.copy()
        np.minimum.at(new, ia, root[ib])
        np.minimum.at(new, ib, root[ia])
        new = new[new]
        if np.array_equal(new, root):
            break
        root = new
    node_rank, node_lab = (nodes >> 32).astype(np.int64), (nodes & 0xffffffff).astype(np.int64)
    phantom = np.isin(nodes, PH)
    rep = {}
    for i in range(nn):
        if not phantom[i] and root[i] not in rep:
            rep[root[i]] = i
    is_rep = np.array([rep.get(root[i], -1) == i for i in range(nn)], bool) if nn else np.zeros(0, bool)
    drop_of = [node_lab[(node_rank == r) & ~is_rep] for r in range(R)]
    offsets = np.concatenate([[0], np.cumsum([nlocs[r] - drop_of[r].size for r in range(R)])]).astype(np.int64)
    class_label = {}
    for i in np.flatnonzero(is_rep):
        r = node_rank[i]
        class_label[root[i]] = offsets[r] + node_lab[i] - np.searchsorted(drop_of[r], node_lab[i])
    mine = np.flatnonzero((node_rank == me) & ~is_rep)
    target = np.array([class_label.get(root[i], 0) for i in mine], np.int64)
    shared = sorted(class_label[c] for c in class_label if np.unique(node_rank[(root == c) & ~phantom]).size > 1)
    return offsets, node_lab[mine], target, np.array(shared, np.int64)


@pytest.mark.parametrize("seed", range(8))
def test_label_sections(seed):
    rng = np.random.default_rng(100 + seed)
    W, R = int(rng.integers(1, 80)), int(rng.integers(2, 5))
    nlocs = rng.integers(5, 40, R).astype(np.int64)

    def runs_row(nmax, p_fg):
        """a row of runs of equal labels with background in between"""
        row = np.zeros(W, np.int32)
        j = 0
        while j < W:
            ln = int(rng.integers(1, 6))
            if rng.random() < p_fg:
                row[j:j + ln] = rng.integers(1, nmax + 1)
            j += ln
        return row
    # the seam between band r and r + 1: two raster rows; each band has its own labels of both (same foreground)
    allp = []
    rows = {}
    for r in range(R - 1):
        for which in ("last", "first"):         # raster row "last" of r (= top halo of r + 1), raster row "first" of r + 1 (= bottom halo of r)
            fg = runs_row(1, 0.6) > 0
            rows[(r, which, "upper")] = np.where(fg, runs_row(int(nlocs[r]), 1.0), 0).astype(np.int32)       # labels of band r
            rows[(r, which, "lower")] = np.where(fg, runs_row(int(nlocs[r + 1]), 1.0), 0).astype(np.int32)   # labels of band r + 1
    for me in range(R):
        ea, eb, ph = np.empty(2 * W, np.int64), np.empty(2 * W, np.int64), np.empty(2 * W, np.int64)
        k = q = 0
        wa, wb, wp = [], [], []
        cases = []
        if me > 0:      # top halo = raster row "last" of the seam above in MY labels; the adjacent owned row = "first"; theirs = upper labels
            cases.append((rows[(me - 1, "last", "lower")], rows[(me - 1, "first", "lower")], rows[(me - 1, "last", "upper")], me - 1))
        if me + 1 < R:
            cases.append((rows[(me, "first", "upper")], rows[(me, "last", "upper")], rows[(me, "first", "lower")], me + 1))
        for halo, edge, nbr, r_nbr in cases:
            n, nq = _c64(), _c64()
            _lib.call("mhip_band_label_pairs", _lib.i64(W), _lib.ptr(halo), _lib.ptr(edge), _lib.ptr(nbr), _lib.i64(me << 32), _lib.i64(r_nbr << 32),
                      _lib.ptr(ea[k:]), _lib.ptr(eb[k:]), ctypes.byref(n), _lib.ptr(ph[q:]), ctypes.byref(nq))
            a, b_, p = np_label_pairs(W, halo.copy(), edge.copy(), nbr.copy(), me, r_nbr)
            assert np.array_equal(ea[k:k + n.value], a) and np.array_equal(eb[k:k + n.value], b_)
            assert np.array_equal(ph[q:q + nq.value], p)
            k += n.value
            q += nq.value
        allp.append(dict(ea=ea[:k].copy(), eb=eb[:k].copy(), ph=ph[:q].copy() | (np.int64(me) << 32)))
    EA, EB, PH = (np.ascontiguousarray(np.concatenate([p[n] for p in allp])) for n in ("ea", "eb", "ph"))
    # (random rows make "phantoms" that are real elsewhere: keep the protocol's invariant that a phantom has a pair)
    PH = np.ascontiguousarray(PH[np.isin(PH, np.concatenate([EA, EB]))])
    for me in range(R):
        cap = max(2 * EA.size + PH.size, 1)
        offsets = np.zeros(R + 1, np.int64)
        dropped, target, shared = np.empty(cap, np.int32), np.empty(cap, np.int32), np.empty(cap, np.int64)
        nd, ns = _c64(), _c64()
        _lib.call("mhip_band_label_merge", R, me, _lib.ptr(nlocs), _lib.i64(EA.size), _lib.ptr(EA), _lib.ptr(EB), _lib.i64(PH.size), _lib.ptr(PH),
                  _lib.ptr(offsets), _lib.ptr(dropped), _lib.ptr(target), ctypes.byref(nd), _lib.ptr(shared), ctypes.byref(ns))
        w_off, w_drop, w_tgt, w_shared = np_label_merge(R, me, nlocs, EA, EB, PH)
        assert np.array_equal(offsets, w_off)
        assert np.array_equal(dropped[:nd.value], w_drop) and np.array_equal(target[:nd.value], w_tgt)
        assert np.array_equal(shared[:ns.value], w_shared)


def test_label_merge_refuses_bad_keys():
    nlocs = np.array([3, 3], np.int64)
    EA, EB = np.array([(0 << 32) | 4], np.int64), np.array([(1 << 32) | 1], np.int64)      # label 4 of a band with 3 labels
    out = np.zeros(3, np.int64)
    d, t, s = np.zeros(4, np.int32), np.zeros(4, np.int32), np.zeros(4, np.int64)
    with pytest.raises(Exception):
        _lib.call("mhip_band_label_merge", 2, 0, _lib.ptr(nlocs), _lib.i64(1), _lib.ptr(EA), _lib.ptr(EB), _lib.i64(0), None, _lib.ptr(out), _lib.ptr(d),
                  _lib.ptr(t), ctypes.byref(_c64()), _lib.ptr(s), ctypes.byref(_c64()))


# ---- watersheds -----------------------------------------------------------------------------------------------------------------
def np_ws_publish(W, me, mine, up, dn):
    mine = mine.astype(np.int64)
    up64 = None if up is None else up.astype(np.int64)
    dn64 = None if dn is None else dn.astype(np.int64)
    neg = np.flatnonzero(mine < 0)
    idx = -mine[neg] - 1
    to_up = idx < W
    tgt_node = np.where(to_up, (2 * (me - 1) + 1) * W + idx, (2 * (me + 1)) * W + (idx - W))
    tgt_val = np.zeros(neg.size, np.int64)
    if up64 is not None:
        tgt_val[to_up] = up64[idx[to_up]]
    if dn64 is not None:
        tgt_val[~to_up] = dn64[idx[~to_up] - W]
    entry = np.where(tgt_val >= 0, tgt_val, -(tgt_node + 1))
    pointed = np.zeros(2 * W, bool)
    if up64 is not None:
        j = -up64[up64 < 0] - 1
        pointed[j[j >= W] - W] = True
    if dn64 is not None:
        j = -dn64[dn64 < 0] - 1
        pointed[W + j[j < W]] = True
    pub = pointed[neg] | (tgt_val < 0)
    return 2 * me * W + neg[pub], entry[pub]


def np_ws_lut(W, me, N, V, up, dn):
    order = np.argsort(N)
    N, V = N[order], V[order].copy()
    res = {}

    def resolve(i, seen):
        if V[i] >= 0:
            return V[i]
        if i in seen:
            return 0
        seen.add(i)
        t = -V[i] - 1
        pos = np.searchsorted(N, t)
        if pos >= N.size or N[pos] != t:
            return 0
        return resolve(pos, seen)
    out = np.zeros(2 * W, np.int64)
    for row, base, lo in ((up, (2 * (me - 1) + 1) * W, 0), (dn, (2 * (me + 1)) * W, W)):
        if row is None:
            continue
        for k in range(W):
            r = int(row[k])
            if r < 0:
                pos = np.searchsorted(N, base + k)
                r = resolve(pos, set()) if pos < N.size and N[pos] == base + k else 0
            out[lo + k] = r
    return out


@pytest.mark.parametrize("seed", range(8))
def test_watershed_sections(seed):
    rng = np.random.default_rng(200 + seed)
    W, R = int(rng.integers(1, 50)), int(rng.integers(2, 5))

    def edge_rows(r):
        """first and last owned row of band r after the local pass: labels, 0, or pseudo labels of halo cells that exist"""
        v = rng.integers(0, 9, 2 * W).astype(np.int32)
        pseudo = rng.random(2 * W) < 0.5
        k = rng.integers(0, 2 * W, 2 * W)
        if r == 0:
            k = W + k % W            # no top halo
        if r == R - 1:
            k = k % W                # no bottom halo
        return np.where(pseudo, -(1 + k), v).astype(np.int32)
    rows = [edge_rows(r) for r in range(R)]
    parts = []
    for me in range(R):
        up = np.ascontiguousarray(rows[me - 1][W:]) if me > 0 else None
        dn = np.ascontiguousarray(rows[me + 1][:W]) if me + 1 < R else None
        N, V, n = np.empty(2 * W, np.int64), np.empty(2 * W, np.int64), _c64()
        _lib.call("mhip_band_ws_publish", _lib.i64(W), me, _lib.ptr(rows[me]), _lib.ptr(up) if up is not None else None,
                  _lib.ptr(dn) if dn is not None else None, _lib.ptr(N), _lib.ptr(V), ctypes.byref(n))
        wn, wv = np_ws_publish(W, me, rows[me], up, dn)
        assert np.array_equal(N[:n.value], wn) and np.array_equal(V[:n.value], wv)
        parts.append((N[:n.value].copy(), V[:n.value].copy()))
    N = np.ascontiguousarray(np.concatenate([p[0] for p in parts]))
    V = np.ascontiguousarray(np.concatenate([p[1] for p in parts]))
    for me in range(R):
        up = np.ascontiguousarray(rows[me - 1][W:]) if me > 0 else None
        dn = np.ascontiguousarray(rows[me + 1][:W]) if me + 1 < R else None
        for NN, VV in ((N, V), (np.ascontiguousarray(N[::-1]), np.ascontiguousarray(V[::-1]))):       # (any order of the gathered pairs)
            lut = np.empty(2 * W, np.int32)
            _lib.call("mhip_band_ws_lut", _lib.i64(W), me, _lib.i64(NN.size), _lib.ptr(NN), _lib.ptr(VV), _lib.ptr(up) if up is not None else None,
                      _lib.ptr(dn) if dn is not None else None, _lib.ptr(lut))
            assert np.array_equal(lut, np_ws_lut(W, me, N, V, up, dn))


# ---- records --------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("kind", [0, 2, 3])
def test_merge_records(kind):
    rng = np.random.default_rng(300 + kind)
    R, n = 5, 200
    parts = []
    for r in range(R):
        if kind == 0:
            a = np.zeros(n, _lib.STAT_DTYPE)
            has = rng.random(n) < 0.5
            a["min"] = np.where(has, rng.random(n), np.inf)
            a["max"] = np.where(has, rng.random(n) + 1, -np.inf)
            a["sum"] = np.where(has, rng.random(n) * 1e3, 0)
            a["count"] = np.where(has, rng.integers(1, 99, n), 0)
            if r == 2:
                a["min"][0] = a["max"][0] = np.nan              # (the background record may see NaN cells)
        else:
            a = np.zeros(n, _lib.INDEX_DTYPE)
            has = rng.random(n) < 0.5
            a["value"] = rng.integers(0, 4, n)                   # many ties
            a["row"] = np.where(has, rng.integers(0, 50, n) + 50 * r, -1)
            a["col"] = np.where(has, rng.integers(0, 50, n), -1)
        parts.append(a)
    m = parts[0].copy()
    for p in parts[1:]:
        if kind == 0:
            m["min"] = np.minimum(m["min"], p["min"])
            m["max"] = np.maximum(m["max"], p["max"])
            m["sum"] = m["sum"] + p["sum"]
            m["count"] = m["count"] + p["count"]
        else:
            better = (p["value"] > m["value"]) if kind == 2 else (p["value"] < m["value"])
            better |= (m["row"] < 0) & (p["row"] >= 0)
            better &= p["row"] >= 0
            m[better] = p[better]
    out = np.empty_like(parts[0])
    pp = (ctypes.c_void_p * R)(*[p.ctypes.data for p in parts])
    _lib.call("mhip_band_merge_records", kind, R, _lib.i64(n), pp, _lib.ptr(out))
    assert out.tobytes() == m.tobytes()
