"""Where the wall clock of a run with SEVERAL contexts on one device goes (row bands on one GPU: their kernels run side by side, a
kernel's own duration then says little): `python tools/band_attribution.py <rocprofv3 results.db> <contexts>` cuts the time from the
first pf_tile_kernel to the last dispatch into slices between dispatch starts / ends, gives each slice in equal shares to the kernels
running in it, and prints per kernel the share of the wall clock it ends up with (ms per step; a step = <contexts> pf_tile_kernel
launches), the slices nothing ran in ("idle": host sections, votes, exchanges) and how many kernels ran side by side on average."""
import re
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    nctx = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rows = list(db.execute("select name, start, end from kernels order by start"))
    first = next(i for i, r in enumerate(rows) if "pf_tile_kernel" in r[0])
    rows = rows[first:]
    steps = max(1, sum(1 for r in rows if "pf_tile_kernel" in r[0]) // nctx)
    ev = []
    for i, (n, s, e) in enumerate(rows):
        ev.append((s, 1, i))
        ev.append((e, 0, i))
    ev.sort()
    def short(n):
        n = re.sub(r"mh::\(anonymous namespace\)::|\(anonymous namespace\)::|void ", "", n)
        return re.sub(r"\(.*", "", n)

    active = set()
    gaps = {}
    last_ended = None
    share = {}
    idle = 0.0
    weighted = 0.0
    busy = 0.0
    prev = ev[0][0]
    for t, kind, i in ev:
        dt = t - prev
        if dt > 0:
            if active:
                w = dt / len(active)
                for j in active:
                    share[j] = share.get(j, 0.0) + w
                busy += dt
                weighted += dt * len(active)
            else:
                idle += dt
        prev = t
        if kind:
            if not active and dt > 0 and last_ended is not None:      # this dispatch ends a stretch with nothing running
                key = (short(rows[last_ended][0]), short(rows[i][0]))
                g = gaps.setdefault(key, [0, 0.0])
                g[0] += 1
                g[1] += dt
            active.add(i)
        else:
            active.discard(i)
            last_ended = i
    per = {}
    calls = {}
    own = {}
    for j, w in share.items():
        n = short(rows[j][0])
        per[n] = per.get(n, 0.0) + w
        calls[n] = calls.get(n, 0) + 1
        own[n] = own.get(n, 0.0) + (rows[j][2] - rows[j][1])
    span = ev[-1][0] - ev[0][0]
    print("kernel,calls_per_step,wall_share_ms_per_step,own_duration_ms_per_step")
    for n in sorted(per, key=lambda k: -per[k]):
        print("%s,%.1f,%.3f,%.3f" % (n, calls[n] / steps, per[n] / steps / 1e6, own[n] / steps / 1e6))
    print("# stretches with nothing running, by (last kernel to end -> first kernel to start): count per step, ms per step")
    for key in sorted(gaps, key=lambda k: -gaps[k][1])[:40]:
        print("# gap %s -> %s,%.1f,%.3f" % (key[0], key[1], gaps[key][0] / steps, gaps[key][1] / steps / 1e6))
    print("# %d steps; span %.2f ms per step; nothing running %.2f ms per step (%.1f %%); %.2f kernels side by side while busy"
          % (steps, span / steps / 1e6, idle / steps / 1e6, 100.0 * idle / span, weighted / max(busy, 1.0)))


if __name__ == "__main__":
    main()
