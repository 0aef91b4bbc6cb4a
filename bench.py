#!/usr/bin/env python3
"""bench.py -- whole-job throughput of the raster hydrology hot path on MI355X.

Metric (BASELINE.json): Mcells/s for fill -> D8 -> accumulation -> bluespot label (-> watershed, pour
points) on a 16384^2 float32 fBm DEM, inputs resident in HBM when the timed region starts, plus the
achieved fraction of the HBM roofline per stage (D8 stencil = the metric's second headline).

A "step" is one pass of the whole chain over the DEM:
    fill (+depths) -> minimum_safe_short_and_diag + no-flats fill -> D8 -> accumulation ->
    connected components + label_stats -> (keep all) -> watersheds + label_count -> pour points (argmax accum)

    python bench.py [--gpus N] [--steps K] [--warmup W] [--size 16384] [--beta 2.0]

N > 1 is launched by torch.distributed.run (one process per GPU).  torch is used ONLY for rank plumbing
(gloo: barrier, max-over-ranks of the timing, moving a few halo rows); the product itself is torch-free.
N > 1 is WEAK scaling of ONE DEM: the global raster is (N * size) x size, row-banded over the GPUs
(malstroem_amd.distributed.BandPipeline: halo-row exchange for the two fills, D8, accumulation, labelling with
cross-band equivalences, watersheds).  Each band is the same fBm tile, mirrored on odd ranks so that the global
surface is continuous across band boundaries.  The per-label record reductions (label_stats, label_count,
pour points) are not band-merged yet and are left out of the N > 1 step (they are ~7 % of the N = 1 step).
Should the band path fail on a box, the run falls back to one independent DEM per GPU and says so.

Only the `cpu_baseline` leg touches oracle/ (the single-thread C restatement of the reference path),
on a bounded sample, on rank 0 at N = 1.
"""
import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

# algorithmic (compulsory) HBM bytes per cell, SURVEY.md 8(d) / BASELINE.md section 4
# one request for the whole chain: the bluespot depths (12 B/cell) are computed on the label branch, not in the fill stage
ALG_BYTES = {"fill": 8, "noflat": 12, "flowdir": 9, "accum": 9, "label": 12 + 8 + 8, "watershed": 9 + 4, "pourpoints": 12}
#   fill      : fill 8 B (4 R + 4 W) + depths 12 B (4 + 4 R, 4 W)
#   label     : CCL 8 B (4 R + 4 W) + label_stats 8 B (4 + 4 R)
#   watershed : 9 B (1 + 4 R, 4 W) + label_count 4 B
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is the measured copy ceiling
STAGES = ["fill", "noflat", "flowdir", "accum", "label", "watershed", "pourpoints"]


def fbm(n, beta=2.0, seed=42):
    """SURVEY.md 8(d) spectral fBm recipe (scipy.fft with threads: same recipe, faster than numpy.fft)."""
    import scipy.fft
    rng = np.random.default_rng(seed)
    kx = np.fft.fftfreq(n)[:, None]
    ky = np.fft.rfftfreq(n)[None, :]
    k = np.hypot(kx, ky)
    k[0, 0] = 1
    amp = k ** (-(beta + 1) / 2)
    del k
    amp[0, 0] = 0
    spec = amp * (rng.normal(size=amp.shape) + 1j * rng.normal(size=amp.shape))
    del amp
    z = scipy.fft.irfft2(spec, s=(n, n), workers=os.cpu_count() or 1)
    del spec
    z -= z.min()
    z *= 100.0 / z.max()
    return z.astype(np.float32)


KERNEL_OF_STAGE = {"fill": "fill_round_kernel<float, ...> (one launch per round)", "noflat": "fill_round_kernel<double, ...> (one launch per round)",
                   "flowdir": "d8_kernel", "accum": "accum_tile_kernel<false/true> + perimeter graph", "label": "ccl_* + stats_kernel",
                   "watershed": "ws_* + count_kernel", "pourpoints": "arg_packed_kernel"}


def pmc_traffic(stage, n):
    """HBM bytes of the stage's kernels from the newest committed rocprofv3 PMC table (profiles/*_pmc_hbm_traffic.json,
    made by tools/pmc_traffic.py: FETCH_SIZE and WRITE_SIZE collected in separate runs at 16384^2; FETCH_SIZE doubled for
    the 16-byte-per-lane streaming reads of d8 / depths as MI355X_MICROARCH.md prescribes, left as reported for the
    4-byte tile loads of the fills)."""
    tables = sorted((ROOT / "profiles").glob("*_pmc_hbm_traffic.json"))
    if n != 16384 or not tables:
        return None
    path = tables[-1]
    kernels = {"flowdir": (("d8_kernel",), 2.0), "fill": (("fill_round_kernel<float",), 1.0),
               "noflat": (("fill_round_kernel<double",), 1.0), "accum": (("accum_",), 1.0), "label": (("ccl_", "stats_"), 1.0),
               "watershed": (("ws_", "count_kernel"), 1.0), "pourpoints": (("arg_",), 1.0)}
    if stage not in kernels:
        return None
    prefixes, fcorr = kernels[stage]
    tot = 0.0
    for row in json.loads(path.read_text()):
        if row["kernel"].startswith(prefixes):
            tot += (row["fetch_size_kb"] * fcorr + row["write_size_kb"]) * 1024.0
    return {"hbm_bytes_per_stage": round(tot), "bytes_per_cell": round(tot / (float(n) * n), 2), "source": "profiles/" + path.name}


def cpu_baseline(dem, sample):
    """Single-thread CPU port (oracle/) of the same chain on a sample x sample crop of the same DEM."""
    import oracle
    crop = np.ascontiguousarray(dem[:sample, :sample])
    t0 = time.perf_counter()
    filled = oracle.fill_terrain(crop)
    depths = oracle.depths(filled, crop)
    short, diag = oracle.minimum_safe_short_and_diag(crop)
    fnf = oracle.fill_terrain_no_flats(crop, short, diag)
    fd = oracle.terrain_flowdirection(fnf)
    acc = oracle.accumulated_flow(fd)
    lab, n = oracle.connected_components(depths)
    oracle.label_stats(depths, lab, n)
    ws = lab.copy()
    oracle.watersheds_from_labels(fd, ws, 0)
    oracle.label_count(ws)
    oracle.label_max_index(acc, lab, n)
    dt = time.perf_counter() - t0
    return {"value": round(crop.size / dt / 1e6, 3), "unit": "Mcells/s", "cores": 1, "kind": "port",
            "sample": "%dx%d crop of the benchmark DEM, full chain, %.1f s" % (sample, sample, dt)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--size", type=int, default=16384)
    ap.add_argument("--beta", type=float, default=2.0)
    ap.add_argument("--cpu-sample", type=int, default=8192)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    if world > 1:
        import torch.distributed as dist  # plumbing only: barrier + max over ranks (gloo, CPU tensors)
        import torch
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)

    from malstroem_amd import _lib
    from malstroem_amd.pipeline import HydroPipeline
    if _lib.device_count() < 1:
        raise SystemExit("bench.py needs an MI355X: libmalstroem_hip has no CPU fallback")

    n = args.size
    t_gen = time.perf_counter()
    dem = fbm(n, beta=args.beta, seed=42)   # every rank builds the same tile (bands mirror it on odd ranks)
    t_gen = time.perf_counter() - t_gen

    ndev = _lib.device_count()
    device = local_rank % ndev
    parallelism = "1 GPU"
    stage_names = STAGES
    band = None
    if world > 1:
        try:
            from malstroem_amd.distributed import BandPipeline, RcclComm, TorchComm
            # halo rows GPU -> GPU over RCCL when it passes its self test on every rank, else host-staged gloo
            transport = os.environ.get("MALSTROEM_BAND_TRANSPORT", "auto")
            if world > ndev:
                transport = "gloo"   # ranks share a GPU (rehearsal on a small box): RCCL wants one device per rank
            comm = TorchComm() if transport == "gloo" else RcclComm.create(device)
            band = BandPipeline(comm, (n * world, n), device=device)
            band.upload_dem(dem if rank % 2 == 0 else dem[::-1])
            parallelism = "row bands of one %dx%d DEM over %d GPUs, halo rows %s" % (
                n * world, n, world, "GPU->GPU over RCCL send/recv" if getattr(comm, "device_rows", False) else "host-staged over gloo")
            stage_names = ["fill", "noflat", "flowdir", "accum", "label", "watershed", "pourpoints"]
        except Exception as e:  # pragma: no cover - only on a multi-GPU node
            import traceback
            traceback.print_exc()
            band = None
            parallelism = "replicas only (band mode failed: %s)" % (str(e)[:80],)
    pipe = None
    if band is None:
        pipe = HydroPipeline(dem.shape, device=device)
        pipe.upload("dem", dem)

    stage_ms = {s: 0.0 for s in stage_names}

    def step(record):
        if band is not None:
            # the whole chain with the labelling branch on a second host thread / side stream (BandPipeline.run_chain); the
            # per-label records are computed and merged across bands like in the single-GPU chain, the slice of a band's
            # own labels stays on the device (nobody downloads records inside the timed loop at N = 1 either)
            tm = {}
            band.run_chain(records=True, fetch_own=False, overlap=os.environ.get("MALSTROEM_BAND_OVERLAP", "1") != "0", timings=tm)
            if record:
                for name in stage_names:
                    stage_ms[name] += tm.get(name, 0.0)
        else:
            # one request for the whole chain: the library runs the bluespot branch (label, watershed) on a second
            # stream next to no-flats fill -> D8 -> accumulation ("stage DAG"); all labels are kept
            pipe.run("fill", "noflat", "flowdir", "accum", "label", "watershed", "pourpoints")
            pipe.sync()
            if record:
                for s in stage_names:
                    stage_ms[s] += pipe.stage_ms(s)

    def barrier():
        if pipe is not None:
            pipe.sync()
        if dist is not None:
            dist.barrier()

    for _ in range(args.warmup):
        step(False)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(True)
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        import torch
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t[0])

    cells = float(n) * n
    ms_per_step = elapsed * 1e3 / args.steps
    value = cells * world * args.steps / elapsed / 1e6
    for s in stage_names:
        stage_ms[s] /= args.steps

    if band is not None:
        info = {"nlabels": band.nlabels, "halo_exchanges": dict(band.exchanges)}
    else:
        info = {k: pipe.get_int(k) for k in ("fill_rounds", "noflat_rounds", "nlabels", "fill_tiles", "fill_visits", "fill_cycles",
                                            "noflat_visits", "noflat_cycles")}
    if rank == 0:
        stages = {}
        for s in stage_names:
            gbs = ALG_BYTES[s] * cells / (stage_ms[s] * 1e-3) / 1e9 if stage_ms[s] > 0 else 0.0
            stages[s] = {"ms": round(stage_ms[s], 3), "alg_bytes_per_cell": ALG_BYTES[s], "achieved_GBs": round(gbs, 1),
                         "frac_of_hbm_peak": round(gbs / HBM_PEAK_GBS, 4)}
        dominant = max(stage_names, key=lambda s: stage_ms[s])
        d8 = stages["flowdir"]
        out = {
            "metric": "Mcells/s fill->D8->accum->label on %d^2 f32 DEM" % n,   # BASELINE.json; the step also runs watersheds + pour points
            "value": round(value, 2), "unit": "Mcells/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32 fill / f64 no-flats+D8+accum / u8 flowdir / i32 labels", "data": "synthetic",
            "config": {"workload": "%dx%d fBm beta=%g float32 DEM per GPU: fill+depths -> no-flats fill -> D8 -> accumulation -> "
                                   "CCL+label_stats -> watersheds+label_count -> pour points" % (n, n, args.beta),
                       "parallelism": parallelism, "dem_generation_s": round(t_gen, 1), **info},
            "roofline": {"bound": "hbm", "kernel": KERNEL_OF_STAGE.get(dominant, dominant), "stage": dominant,
                         "launches_per_step": info.get({"fill": "fill_rounds", "noflat": "noflat_rounds"}.get(dominant, ""), 1),
                         "achieved": stages[dominant]["achieved_GBs"], "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": stages[dominant]["frac_of_hbm_peak"], "traffic": pmc_traffic(dominant, n),
                         "note": "dominant stage by device time; algorithmic bytes of the whole stage / stage time (HIP events)"},
            "d8_roofline": {"bound": "hbm", "kernel": "d8_kernel", "achieved": d8["achieved_GBs"], "peak": HBM_PEAK_GBS,
                            "unit": "GB/s", "frac": d8["frac_of_hbm_peak"], "traffic": pmc_traffic("flowdir", n)},
            "stages": stages,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(dem, min(args.cpu_sample, n))
        print(json.dumps(out), flush=True)
    if pipe is not None:
        pipe.close()
    if band is not None:
        band.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
