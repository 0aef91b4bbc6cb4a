"""Merge two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE: they do not fit one pass on gfx950) into the per-kernel HBM
traffic table bench.py reads.  On the GPU box:

    cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
    rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/pmc_f -o f -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline
    rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/pmc_w -o w -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline
    python3 tools/pmc_traffic.py gpurun_out/pmc_f/f_results.db gpurun_out/pmc_w/w_results.db 16384 > profiles/<round>_pmc_hbm_traffic.json

FETCH_SIZE / WRITE_SIZE are in KiB per dispatch.  Corrections (MI355X_MICROARCH.md, "HBM"): FETCH_SIZE counts half the
bytes of 16-byte-per-lane streaming reads -- bench.py doubles it for d8_kernel only; other widths are left as reported
(uncalibrated, fine for ratios and for spotting re-reads)."""
import json
import re
import sqlite3
import sys


def per_kernel(path, counter):
    db = sqlite3.connect(path)
    out = {}
    for name, val in db.execute("select name, counter_value from pmc_events where counter_name = ?", (counter,)):
        m = re.search(r"(?:mh::\(anonymous namespace\)::)?(\w+(?:<[^>]*>)?)\(", name)
        k = m.group(1) if m else name
        e = out.setdefault(k, [0, 0.0])
        e[0] += 1
        e[1] += float(val)
    return out


def main():
    fdb, wdb, n = sys.argv[1], sys.argv[2], int(sys.argv[3])
    f, w = per_kernel(fdb, "FETCH_SIZE"), per_kernel(wdb, "WRITE_SIZE")
    cells = float(n) * n
    rows = []
    for k in sorted(set(f) | set(w), key=lambda k: -(f.get(k, [0, 0])[1] + w.get(k, [0, 0])[1])):
        if k.startswith("__amd_rocclr"):
            continue
        fl, fk = f.get(k, [0, 0.0])
        wl, wk = w.get(k, [0, 0.0])
        rows.append({"kernel": k, "launches": max(fl, wl), "fetch_size_kb": fk, "write_size_kb": wk,
                     "fetch_bytes_per_cell_raw": fk * 1024 / cells, "write_bytes_per_cell": wk * 1024 / cells})
    print(json.dumps(rows, indent=1))


if __name__ == "__main__":
    main()
