"""One step of a rocprofv3 --kernel-trace run as a timeline: `python tools/kernel_timeline.py <results.db> [step]` prints, for the
step's dispatches in start order, start (us from the step's first dispatch), duration, the idle gap before it (no kernel running
on the device) and the kernel name; then the busy / idle totals.  A step = the dispatches between two pf_tile_kernel launches."""
import re
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    which = int(sys.argv[2]) if len(sys.argv) > 2 else -2
    cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
    rows = list(db.execute("select name, start, end from kernels order by start"))
    marks = [i for i, r in enumerate(rows) if "pf_tile_kernel" in r[0]]
    lo = marks[which]
    hi = marks[which + 1] if which + 1 < 0 and which + 1 != 0 else (marks[which + 1] if which + 1 < len(marks) and which >= 0 else len(rows))
    # the step starts a few dispatches before pf_tile (memsets): take everything after the previous step's last big kernel
    step = rows[lo:hi]
    t0 = step[0][1]
    busy_until = t0
    idle = 0
    out = []
    for n, s, e in step:
        gap = max(0, s - busy_until)
        idle += gap
        busy_until = max(busy_until, e)
        n = re.sub(r"mh::\(anonymous namespace\)::", "", n)
        n = re.sub(r"\(.*", "", n)
        out.append((s - t0, e - s, gap, n))
    total = busy_until - t0
    print("start_us,duration_us,idle_gap_before_us,kernel")
    for s, d, g, n in out:
        print("%.1f,%.1f,%.1f,%s" % (s / 1e3, d / 1e3, g / 1e3, n))
    print("# span %.1f us, device idle %.1f us (%.1f %%), %d dispatches" % (total / 1e3, idle / 1e3, 100.0 * idle / total, len(step)))


if __name__ == "__main__":
    main()
