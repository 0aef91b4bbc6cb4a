#!/usr/bin/env python3
"""CPU model of the flood's queue-driven seed-graph solve (csrc/pflood.hip: pf_solve_queue_body) under RANDOM interleavings.

The kernel: a resident grid of workgroups takes BLOCK visits from one queue; a visit loads the levels of the block's seeds and of
the seeds of its neighbouring blocks, relaxes the block to its fixed point (dst <- max(w, src), minimax), writes back what dropped
and wakes the neighbours that hold a link to a seed it lowered.  What is modelled, step by step (every step is a point where the
scheduler may switch to another worker -- the only atomicity assumed is that of the single atomics the kernel uses):

    take    ticket = head++ ; wait until slot[ticket] is filled (or finished == tail: leave)
    mark    mark[blk] <- RUNNING                     (atomicExch: clears "queued / woken again")
    load    snapshot of the levels of the block's region (the block's own as `old`)
    relax   local fixed point on the snapshot
    store   per dropped seed:   variant "min":   Lv[s] = min(Lv[s], v)       (agent-scope atomicMin -- the kernel)
                                variant "store": Lv[s] = v                   (a plain store -- the first version)
    wake    per neighbour with a link to a dropped seed: old = mark[nb] |= QUEUED; if old == 0: append nb   (atomicOr)
                                variant "exch" (first version): old = exchange(mark[nb], QUEUED): a RUNNING block is queued at once
    end     old = mark[blk] &= ~RUNNING; if old & QUEUED: append blk (woken while it ran) ; finished++
                                variant "exch": no running bit at all (mark cleared at `mark`)

Variants:  "r04" = min + running bit (what pflood.hip does);  "first" = store + exch (the first version of round 4: a block woken
while a visit of it is running is taken by a second worker, and the visit that started from the older levels puts a seed back UP);
"late_old" = "r04" with the `old` levels copied AFTER the first relaxations of a visit (the race that had been in the rounds' kernel
since round 2: a thread lowered a level before another thread had copied it -- the drop is never written back, nobody is woken).

`search(variant, ...)` builds random seed graphs on a grid of blocks (edges inside a block and to the 8 neighbouring blocks, OCEAN
edges on the outline blocks), runs the protocol with several workers under a seeded random scheduler and compares the levels with
the exact minimax distances (Dijkstra).  tests/test_queue_protocol_model.py pins: "r04" is exact on every schedule tried, the other
two are caught leaving a level too high -- which is what the flood's run-time proof (check.hip) caught on the GPU.
"""
import heapq
import random

INF = float("inf")
QUEUED, RUNNING = 1, 2


def make_graph(rng, nbr, nbc, seeds_per_block=3, p_link=0.7):
    """-> (nblocks, seeds: block -> list of seed ids, edges: list of (a, b, w) undirected, ocean: list of (s, w))"""
    nb = nbr * nbc
    seeds = {b: [b * seeds_per_block + k for k in range(seeds_per_block)] for b in range(nb)}
    edges, ocean = [], []
    for b in range(nb):
        s = seeds[b]
        for i in range(len(s)):          # spill edges inside the block
            for j in range(i + 1, len(s)):
                if rng.random() < 0.8:
                    edges.append((s[i], s[j], rng.randint(1, 30)))
        bi, bj = divmod(b, nbc)
        for di in (-1, 0, 1):            # links to the neighbouring blocks (each unordered pair of blocks once)
            for dj in (-1, 0, 1):
                ni, nj = bi + di, bj + dj
                if (di, dj) <= (0, 0) or not (0 <= ni < nbr and 0 <= nj < nbc):
                    continue
                for a in s:
                    for c in seeds[ni * nbc + nj]:
                        if rng.random() < p_link / 3:
                            edges.append((a, c, rng.randint(1, 30)))
        if bi in (0, nbr - 1) or bj in (0, nbc - 1):
            ocean.append((rng.choice(s), rng.randint(1, 30)))
    return nb, seeds, edges, ocean


def exact(nseeds, edges, ocean):
    adj = {}
    for a, b, w in edges:
        adj.setdefault(a, []).append((b, w))
        adj.setdefault(b, []).append((a, w))
    L = [INF] * nseeds
    pq = []
    for s, w in ocean:
        if w < L[s]:
            L[s] = w
            heapq.heappush(pq, (w, s))
    while pq:
        d, s = heapq.heappop(pq)
        if d > L[s]:
            continue
        for t, w in adj.get(s, ()):
            v = max(d, w)
            if v < L[t]:
                L[t] = v
                heapq.heappush(pq, (v, t))
    return L


def run(variant, rng, nbr, nbc, nworkers=4, **graph_kw):
    nb, seeds, edges, ocean = make_graph(rng, nbr, nbc, **graph_kw)
    nseeds = sum(len(v) for v in seeds.values())
    block_of = {s: b for b, ss in seeds.items() for s in ss}
    # directed relaxations by the block of their destination
    rel = {b: [] for b in range(nb)}
    for a, b_, w in edges:
        rel[block_of[a]].append((a, b_, w))
        rel[block_of[b_]].append((b_, a, w))
    for s, w in ocean:
        rel[block_of[s]].append((s, None, w))
    Lv = [INF] * nseeds
    mark = [0] * nb
    outline = [b for b in range(nb) if (b // nbc) in (0, nbr - 1) or (b % nbc) in (0, nbc - 1)]
    slots = list(outline)                    # the queue: slots[ticket]
    for b in outline:
        mark[b] = QUEUED
    state = {"head": 0, "finished": 0}

    def worker():
        while True:
            ticket = state["head"]
            state["head"] += 1
            yield
            while ticket >= len(slots):
                if state["finished"] == len(slots):      # (finished read first, then tail: one step here)
                    return
                yield
            blk = slots[ticket]
            mark[blk] = RUNNING if variant != "first" else 0
            yield
            mine = seeds[blk]
            region = {}
            old = {}
            order = [(d, s, w) for d, s, w in rel[blk]]
            rng.shuffle(order)
            for d, s, w in order:                            # the loads, a seed at a time (other workers run in between)
                for x in (d, s):
                    if x is not None and x not in region:
                        region[x] = Lv[x]
                        yield
            old = {s: region.get(s, Lv[s]) for s in mine}
            late = [s for s in mine if variant == "late_old" and rng.random() < 0.15]     # the copies of a slow thread
            changed = True
            first = True
            while changed:                                   # the block's fixed point on the snapshot
                changed = False
                for d, s, w in order:
                    v = max(w, region[s]) if s is not None else w
                    if v < region.get(d, INF):
                        region[d] = v
                        changed = True
                if first:                                    # ... happen after a fast thread's first relaxations
                    for s in late:
                        old[s] = region.get(s, INF)
                first = False
            yield
            dropped = [s for s in mine if region.get(s, INF) < old[s]]
            for s in dropped:
                Lv[s] = min(Lv[s], region[s]) if variant != "first" else region[s]
                yield
            wake = set()
            for d, s, w in rel[blk]:
                if s is not None and block_of[s] != blk and d in dropped:
                    wake.add(block_of[s])
            for nbk in sorted(wake):
                if variant == "first":
                    o, mark[nbk] = mark[nbk], QUEUED
                else:
                    o = mark[nbk]
                    mark[nbk] |= QUEUED
                if o == 0:
                    slots.append(nbk)
                yield
            if variant != "first":
                o = mark[blk]
                mark[blk] &= ~RUNNING
                if o & QUEUED:
                    slots.append(blk)
            state["finished"] += 1
            yield

    workers = [worker() for _ in range(nworkers)]
    alive = list(range(nworkers))
    steps = 0
    while alive:
        k = rng.choice(alive)
        for _ in range(rng.choice((1, 1, 2, 5, 40, 200))):      # bursts: a worker may stall for a long while between two of its steps
            try:
                next(workers[k])
            except StopIteration:
                alive.remove(k)
                break
            steps += 1
        if steps > 2_000_000:
            raise RuntimeError("the model did not terminate")
    return Lv, exact(nseeds, edges, ocean), len(slots)


def search(variant, trials, seed, nbr=4, nbc=4, nworkers=4, stop_at_first=False):
    """-> (number of runs that ended with a level too high, first failing (levels, exact) or None, a level below the exact one ever seen)"""
    bad, first, below = 0, None, False
    for t in range(trials):
        rng = random.Random(seed * 100003 + t)
        got, want, _ = run(variant, rng, nbr, nbc, nworkers)
        if any(g < w for g, w in zip(got, want)):
            below = True
        if got != want:
            bad += 1
            if first is None:
                first = (got, want)
            if stop_at_first:
                break
    return bad, first, below


if __name__ == "__main__":
    for v in ("r04", "first", "late_old"):
        print(v, search(v, 300, 1)[0], "of 300 runs end with a level too high")
