"""Merge two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE: they do not fit one pass on gfx950) into the per-kernel HBM
traffic table bench.py reads.  On the GPU box:

    cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
    rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/pmc_f -o f -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline
    rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/pmc_w -o w -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline
    python3 tools/pmc_traffic.py gpurun_out/pmc_f/f_results.db gpurun_out/pmc_w/w_results.db 16384 > profiles/<round>_pmc_hbm_traffic.json

FETCH_SIZE / WRITE_SIZE are in KiB per dispatch.  Calibration (MI355X_MICROARCH.md, "HBM": FETCH_SIZE = TCC_EA0_RDREQ x 64 B
while a coalesced streaming read is served in 128-byte requests): every kernel of this run whose compulsory reads are known
exactly reports HALF of them, whatever its load width -- read16_kernel and copy16_kernel (16 B per lane, 2**30 bytes per call:
0.537 GB reported), minmax_kernel (16 B per lane, 4 B/cell: 2.00), count_kernel and ccl_tile_kernel (4 B per lane, 4 B/cell: 2.00),
d8s_kernel (16 B per lane, 8 B/cell + halo rows: 4.74) -- while WRITE_SIZE is exact (copy16_kernel: 1.074 GB per call).  So
FETCH_SIZE is doubled for EVERY kernel (`fetch_bytes_per_cell` = 2 x raw; all reads on this path are full-line streams or
L2-resident tables; a scattered gather would be over-corrected, i.e. the figure is an upper bound there).  A kernel whose corrected
fetch still lies below its compulsory reads is flagged (`below_compulsory`): its inputs came out of the Infinity Cache / L2 (written
by the kernel before it) rather than from HBM."""
import json
import re
import sqlite3
import sys


# compulsory HBM bytes per cell and step of the streaming kernels (each input read once, each output written once)
COMPULSORY = {
    "d8s_kernel": (8, 1), "d8_kernel": (8, 1), "minmax_kernel": (4, 0), "pf_apply_check_kernel": (6, 8), "pf_apply_kernel": (6, 8),
    "pf_tile_kernel": (4, 2), "ng_first_kernel": (4, 5.1), "ng_finish_kernel": (12, 8), "ccl_tile_kernel": (4, 4),
    "ccl_emit_ranked_kernel": (4, 4), "stats_kernel": (8, 0), "count_kernel": (4, 0), "ws_tile_kernel": (5, 4), "ws_assign_hop_kernel": (8, 4),
    "arg_packed_kernel": (12, 0), "accum_tile_kernel": (1, 2), "accum_final_walk_kernel": (3, 8), "fill_check_kernel": (8, 0), "depths_kernel": (8, 4),
}


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
        row = {"kernel": k, "launches": max(fl, wl), "fetch_size_kb": fk, "write_size_kb": wk,
               "fetch_bytes_per_cell_raw": round(fk * 1024 / cells, 3), "fetch_bytes_per_cell": round(2 * fk * 1024 / cells, 3),
               "write_bytes_per_cell": round(wk * 1024 / cells, 3)}
        base = re.sub(r"<.*", "", k)
        if base in COMPULSORY and n * n == cells:
            cr, cw = COMPULSORY[base]
            row["compulsory_read_bytes_per_cell"], row["compulsory_write_bytes_per_cell"] = cr, cw
            row["below_compulsory"] = bool(row["fetch_bytes_per_cell"] < 0.97 * cr)
        rows.append(row)
    # the library the counters were collected with (bench.py only attaches counter figures to a live time of the SAME build) and
    # the number of chain steps in the profiled command (`--steps 1 --warmup 0`: one)
    import hashlib
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    h = hashlib.sha256()       # (the same fingerprint as bench.lib_fingerprint: the library's sources, not the shared object's bytes)
    for f in sorted((root / "malstroem_amd" / "csrc").glob("*.hip")) + [root / "malstroem_amd" / "csrc" / "common.hpp",
                                                                        root / "malstroem_amd" / "csrc" / "Makefile", root / "include" / "malstroem_hip.h"]:
        h.update(f.name.encode() + b"\0" + f.read_bytes() + b"\0")
    meta = {"library_sources_sha256_16": h.hexdigest()[:16],
            "steps": int(sys.argv[4]) if len(sys.argv) > 4 else 1, "cells": n * n,
            "fetch_correction": "fetch_bytes_per_cell = 2 x FETCH_SIZE: exact for coalesced streams, an UPPER BOUND for scattered gathers"}
    print(json.dumps({"meta": meta, "kernels": rows}, indent=1))


if __name__ == "__main__":
    main()
