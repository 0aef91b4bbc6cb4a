#!/usr/bin/env python3
"""Deterministic CPU model of the worklist protocol of csrc/fill.hip (the iterative tile schedule, plain fill, MT == 2).

What is modelled, rule by rule (fill.hip line numbers as of round 3):
  * tiles of TI x TI cells, windows of (TI + 2)^2 with a one-cell halo ring; a MACRO tile = 2 x 2 tiles = one workgroup
  * a visit loads the four windows, then iterates at most MAXIT times (visit_macro, `for (int it ...)`):
        it == 0: a tile runs a cycle only if its bit is set in the macro tile's mark word (`need = flags >> wave & 1`); a tile
                 that does not run publishes the edges of its window as loaded
        it >= 1: every tile runs a cycle; before the vertical (horizontal) passes the halo row (column) that faces the sibling is
                 lowered to the edge row (column) the sibling published at the END of the previous iteration -- all 64 lanes of
                 it, so the sibling's own halo lanes travel along: that is how a diagonal sibling's corner cell arrives, one
                 iteration later than an edge cell
        a cycle = passes down, up (row layout), right, left (column layout); a pass updates one row after the other, all cells
        of a row at once: nv = max(dem, min over the 3 x 3 neighbourhood)  (pass_plain)
        the loop ends after an iteration in which no tile changed a cell (`capped = false`) or after MAXIT iterations
  * after the loop every tile probes its four halo edges ("which halo cells would drop given my edge cells", probe_plain) and
    sets the bit of each of its 8 neighbour tiles in the NEXT round's mark word of that neighbour's macro tile.  For a sibling
    (neighbour inside the same macro tile):
        rule "r02":  want = capped                 -- the sibling "has seen my edges unless the exchange was capped"
        rule "r03":  want = capped or probe bit    -- the fix
  * rounds: a round visits every macro tile whose mark word is non-zero; the first round visits all of them, a window that
    touches no raster border cell skips its first cycle (INIT_INF); converged when a round marks nothing.

`search()` runs the schedule on random small integer terrains and compares with the true greatest fixed point.
Finding of round 3 (the "lost wake-up" of DESIGN 4.2, one cell of 1.07 G on 16384 x 65536 as four bands): with rule "r02" a
cell at the centre of a macro tile whose only lower neighbour is the DIAGONAL sibling's corner cell stays too high when that
corner drops in iteration 0 and iteration 1 is quiet -- the corner needs two iterations to reach the diagonal sibling through
the halo lanes of the two edge siblings, the loop ends after one quiet iteration, and the probe that would have re-queued
the sibling is discarded.  tests/test_fill_protocol_model.py keeps both facts: "r02" loses wake-ups, "r03" does not.
"""
import numpy as np

INF = np.float32(np.inf)


def fixed_point(dem):
    """Greatest fixed point of W = max(dem, min(W, 8 nbrs)), border = dem (reference fill.py:112-171), by plain iteration."""
    H, Wd = dem.shape
    w = np.full_like(dem, INF)
    w[0], w[-1], w[:, 0], w[:, -1] = dem[0], dem[-1], dem[:, 0], dem[:, -1]
    while True:
        p = np.pad(w, 1, constant_values=INF)
        m = np.minimum.reduce([p[1 + dr:1 + dr + H, 1 + dc:1 + dc + Wd] for dr in (-1, 0, 1) for dc in (-1, 0, 1)])
        nw = w.copy()
        nw[1:-1, 1:-1] = np.maximum(dem[1:-1, 1:-1], m[1:-1, 1:-1])
        if np.array_equal(nw, w):
            return w
        w = nw


def _hmin3(row):
    p = np.concatenate([[INF], row, [INF]])
    return np.minimum(np.minimum(p[:-2], p[1:-1]), p[2:])


def _probe(w, d, upd):
    """halo rows 0 / WN-1: which cells would drop given interior rows 1 / WN-2 (probe_plain)"""
    a, b = np.where(upd, w[1], INF), np.where(upd, w[-2], INF)
    ma, mb = _hmin3(a), _hmin3(b)
    return np.maximum(d[0], np.minimum(w[0], ma)) < w[0], np.maximum(d[-1], np.minimum(w[-1], mb)) < w[-1]


class Model(object):
    def __init__(self, dem, TI=3, MAXIT=2, rule="r03", rng=None, p_stale=0.3):
        self.dem = np.asarray(dem, dtype=np.float32)
        self.H, self.Wd = self.dem.shape
        self.TI, self.WN, self.MAXIT, self.rule = TI, TI + 2, MAXIT, rule
        self.ntr, self.ntc = -(-max(self.H - 2, 1) // TI), -(-max(self.Wd - 2, 1) // TI)
        self.mtr, self.mtc = -(-self.ntr // 2), -(-self.ntc // 2)
        self.rng = rng or np.random.default_rng(0)
        self.p_stale = p_stale
        self.W = np.full_like(self.dem, INF)       # _initialize_filled: +inf, borders = dem
        self.W[0], self.W[-1], self.W[:, 0], self.W[:, -1] = self.dem[0], self.dem[-1], self.dem[:, 0], self.dem[:, -1]
        self.visits = 0

    def _load(self, src, ti, tj):
        WN, TI = self.WN, self.TI
        w, d = np.full((WN, WN), INF, np.float32), np.full((WN, WN), INF, np.float32)
        r0, c0 = ti * TI, tj * TI
        r1, c1 = min(r0 + WN, self.H), min(c0 + WN, self.Wd)
        w[:r1 - r0, :c1 - c0] = src[r0:r1, c0:c1]
        d[:r1 - r0, :c1 - c0] = self.dem[r0:r1, c0:c1]
        # raster border cells never move: the kernel pins them through dem == W and the lane / row masks
        return w, d

    def _movable(self, ti, tj):
        """window cells this tile may update: interior of the window, inside the raster, not a raster border cell"""
        WN, TI = self.WN, self.TI
        rr = ti * TI + np.arange(WN)[:, None]
        cc = tj * TI + np.arange(WN)[None, :]
        m = (rr >= 1) & (rr <= self.H - 2) & (cc >= 1) & (cc <= self.Wd - 2)
        m[0], m[-1], m[:, 0], m[:, -1] = False, False, False, False
        return m

    def visit(self, macro, flags, src):
        """One macro-tile visit.  Returns the list of (macro tile, bit) marks for the next round."""
        WN, TI = self.WN, self.TI
        mi, mj = divmod(macro, self.mtc)
        tiles = {}
        for q in range(4):
            ti, tj = 2 * mi + (q >> 1), 2 * mj + (q & 1)
            if ti < self.ntr and tj < self.ntc:
                w, d = self._load(src, ti, tj)
                tiles[q] = dict(ti=ti, tj=tj, w=w, d=d, mv=self._movable(ti, tj), changed=False)
        # edges[parity][q] = (row1, rowTI, col1, colTI), each with all WN lanes (halo lanes included)
        edges = [dict(), dict()]
        inf_edge = (np.full(WN, INF, np.float32),) * 4
        capped = True
        for it in range(self.MAXIT):
            cur, prev = it & 1, (it & 1) ^ 1
            any_chg = False
            newedges = {}
            for q in range(4):
                if q not in tiles:
                    newedges[q] = inf_edge
                    continue
                t = tiles[q]
                w, d, mv = t["w"], t["d"], t["mv"]
                need = bool((flags >> q) & 1) if it == 0 else True
                if need:
                    qi, qj = q >> 1, q & 1
                    chg = False
                    # half 0: vertical sibling's edge row, then down / up passes in the row layout
                    if it > 0:
                        sib = edges[prev].get(q ^ 2, inf_edge)
                        if qi == 1:
                            w[0] = np.minimum(w[0], sib[1])          # its row TI
                        else:
                            w[-1] = np.minimum(w[-1], sib[0])        # its row 1
                    for down in (True, False):
                        for r in (range(1, WN - 1) if down else range(WN - 2, 0, -1)):
                            s = 1 if down else -1
                            nv = np.maximum(d[r], np.minimum(np.minimum(_hmin3(w[r - s]), _hmin3(w[r])), _hmin3(w[r + s])))
                            nv = np.where(mv[r], nv, w[r])
                            chg |= bool(np.any(nv != w[r]))
                            w[r] = nv
                    # half 1: horizontal sibling's edge column, then right / left passes (column layout)
                    wt, dt, mt = w.T, d.T, mv.T    # views: updates go through to w
                    if it > 0:
                        sib = edges[prev].get(q ^ 1, inf_edge)
                        if qj == 1:
                            wt[0] = np.minimum(wt[0], sib[3])        # its column TI
                        else:
                            wt[-1] = np.minimum(wt[-1], sib[2])      # its column 1
                    for down in (True, False):
                        for r in (range(1, WN - 1) if down else range(WN - 2, 0, -1)):
                            s = 1 if down else -1
                            nv = np.maximum(dt[r], np.minimum(np.minimum(_hmin3(wt[r - s]), _hmin3(wt[r])), _hmin3(wt[r + s])))
                            nv = np.where(mt[r], nv, wt[r])
                            chg |= bool(np.any(nv != wt[r]))
                            wt[r] = nv
                    t["changed"] |= chg
                    any_chg |= chg
                newedges[q] = (w[1].copy(), w[-2].copy(), w[:, 1].copy(), w[:, -2].copy())
            edges[cur] = newedges
            if not any_chg:
                capped = False
                break
        # stage-out: the cells a tile owns
        for t in tiles.values():
            r0, c0 = t["ti"] * TI, t["tj"] * TI
            rr, cc = np.nonzero(t["mv"])
            self.W[r0 + rr, c0 + cc] = t["w"][rr, cc]
        # probes -> marks
        marks = []
        for q, t in tiles.items():
            w, d = t["w"], t["d"]
            upd_cols = np.zeros(WN, bool)
            upd_cols[1:-1] = True
            topN, botN = _probe(w, d, upd_cols)
            leftT, rightT = _probe(w.T, d.T, upd_cols)
            inner = slice(1, WN - 1)
            bits = {(-1, -1): topN[0] or leftT[0], (-1, 0): topN[inner].any(), (-1, 1): topN[-1] or rightT[0],
                    (0, -1): leftT[inner].any(), (0, 1): rightT[inner].any(),
                    (1, -1): botN[0] or leftT[-1], (1, 0): botN[inner].any(), (1, 1): botN[-1] or rightT[-1]}
            for dp in (-1, 0, 1):
                for dq in (-1, 0, 1):
                    p, qq = t["ti"] + dp, t["tj"] + dq
                    if not (0 <= p < self.ntr and 0 <= qq < self.ntc):
                        continue
                    tm = (p >> 1) * self.mtc + (qq >> 1)
                    bit = bool(bits.get((dp, dq), False))
                    if tm == macro:
                        want = capped if self.rule == "r02" else (capped or bit)
                    else:
                        want = bit
                    if want:
                        marks.append((tm, 1 << ((p & 1) * 2 + (qq & 1))))
        self.visits += 1
        return marks

    def run(self, max_rounds=10000):
        nmt = self.mtr * self.mtc
        # first round: every macro tile; a window without a raster border cell skips its first cycle (INIT_INF)
        flags = {}
        for m in range(nmt):
            mi, mj = divmod(m, self.mtc)
            f = 0
            for q in range(4):
                r0, c0 = (2 * mi + (q >> 1)) * self.TI, (2 * mj + (q & 1)) * self.TI
                interior = r0 > 0 and r0 + self.WN - 1 < self.H - 1 and c0 > 0 and c0 + self.WN - 1 < self.Wd - 1
                if not interior:
                    f |= 1 << q
            flags[m] = f
        todo = dict(flags)
        first = True
        for rnd in range(max_rounds):
            nxt = {}
            order = list(todo)
            self.rng.shuffle(order)
            snapshot = self.W.copy()
            for m in order:
                # concurrency: a visit may have loaded its windows before the other visits of the round stored theirs
                src = snapshot if self.rng.random() < self.p_stale else self.W
                if src is snapshot:
                    # its OWN cells are current (nobody else writes them): overlay them
                    src = snapshot.copy()
                    mi, mj = divmod(m, self.mtc)
                    r0, c0 = 2 * mi * self.TI + 1, 2 * mj * self.TI + 1
                    src[r0:r0 + 2 * self.TI, c0:c0 + 2 * self.TI] = self.W[r0:r0 + 2 * self.TI, c0:c0 + 2 * self.TI]
                for tm, bit in self.visit(m, todo[m] if not first else flags[m], src):
                    nxt[tm] = nxt.get(tm, 0) | bit
            first = False
            if not nxt:
                return rnd + 1
            todo = nxt
        raise RuntimeError("no convergence")


def search(rule, trials=300, seed=0, shape=(14, 14), TI=3, levels=6, stop_at_first=False):
    """-> (number of terrains on which the schedule ends above the fixed point, first failing (dem, result, truth) or None)"""
    rng = np.random.default_rng(seed)
    bad, first = 0, None
    for _ in range(trials):
        dem = rng.integers(0, levels, size=shape).astype(np.float32)
        m = Model(dem, TI=TI, rule=rule, rng=np.random.default_rng(int(rng.integers(1 << 30))))
        m.run()
        truth = fixed_point(dem)
        assert np.all(m.W >= truth), "the schedule went BELOW the fixed point: the model itself is broken"
        if not np.array_equal(m.W, truth):
            bad += 1
            if first is None:
                first = (dem, m.W.copy(), truth)
                if stop_at_first:
                    break
    return bad, first


if __name__ == "__main__":
    for rule in ("r02", "r03"):
        bad, first = search(rule)
        print("rule %s: %d of 300 random terrains end above the fixed point" % (rule, bad))
        if first is not None:
            dem, got, truth = first
            rr, cc = np.nonzero(got != truth)
            print("  first failure: cells", list(zip(rr.tolist(), cc.tolist())), "tile coordinates (TI = 3):",
                  [((r - 1) // 3, (c - 1) // 3, (r - 1) % 3 + 1, (c - 1) % 3 + 1) for r, c in zip(rr, cc)])
