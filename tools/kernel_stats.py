"""Per-kernel summary (calls, total / average duration, share) of a rocprofv3 --kernel-trace --stats run, from the rocpd
database it writes (<dir>/<name>_results.db): `python tools/kernel_stats.py gpurun_out/prof/x_results.db > profiles/<round>_kernel_stats.csv`"""
import re
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
    name = "name" if "name" in cols else cols[0]
    dur = "duration" if "duration" in cols else None
    if dur:
        q = "select %s, count(*), sum(duration), avg(duration) from kernels group by %s order by sum(duration) desc" % (name, name)
    else:
        q = "select %s, count(*), sum(end - start), avg(end - start) from kernels group by %s order by sum(end - start) desc" % (name, name)
    rows = list(db.execute(q))
    total = sum(r[2] for r in rows) or 1
    print("name,total_calls,total_duration_us,average_us,percentage")
    for n, c, t, a in rows:
        n = re.sub(r"mh::\(anonymous namespace\)::", "", n)
        print('"%s",%d,%.3f,%.3f,%.3f' % (n, c, t / 1e3, a / 1e3, 100.0 * t / total))


if __name__ == "__main__":
    main()
