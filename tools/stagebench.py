#!/usr/bin/env python3
"""Time individual pipeline stages on one GPU (development helper): python tools/stagebench.py --size 8192 --stages fill noflat"""
import argparse, sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
from bench import fbm
from malstroem_amd.pipeline import HydroPipeline

ap = argparse.ArgumentParser()
ap.add_argument("--size", type=int, default=8192)
ap.add_argument("--beta", type=float, default=2.0)
ap.add_argument("--reps", type=int, default=2)
ap.add_argument("--stages", nargs="+", default=["fill"])
args = ap.parse_args()
dem = fbm(args.size, beta=args.beta)
with HydroPipeline(dem.shape) as p:
    p.upload("dem", dem)
    for rep in range(args.reps):
        for s in args.stages:
            if s == "watershed" and rep >= 0:
                pass
            p.run(s)
            if s == "label":
                p.apply_keep(None)
        p.sync()
        out = {s: round(p.stage_ms(s), 3) for s in args.stages}
        for k in ("fill_rounds", "fill_visits", "fill_cycles", "noflat_rounds", "noflat_visits", "noflat_cycles", "fill_tiles"):
            out[k] = p.get_int(k)
        print(out, flush=True)
