"""On the GPU box, after tools/lab/ktime.sh: start-to-start times and durations of the no-flats rounds of one step, launch by launch
(python3 tools/lab/round_times.py /tmp/kt/k_results.db)."""
import re, sqlite3, sys
db = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
name = "name" if "name" in cols else cols[0]
rows = list(db.execute("select %s, start, end from kernels order by start" % name))
i0 = [i for i, r in enumerate(rows) if "ng_first" in r[0]]
i = i0[len(i0) // 2]
out, prev = [], None
for n, s, e in rows[i:]:
    if "ng_finish" in n:
        out.append(("finish", (s - prev) / 1e3 if prev else 0, (e - s) / 1e3)); break
    if "ng_" in n:
        out.append((re.search(r"ng_\w+", n).group(0)[:10], (s - prev) / 1e3 if prev else 0, (e - s) / 1e3)); prev = s
print("kernel, us since the previous ng_ launch started, duration us")
for o in out: print("%-10s %8.1f %8.1f" % o)
