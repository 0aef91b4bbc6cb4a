#!/usr/bin/env python3
"""bench.py -- whole-job throughput of the raster hydrology hot path on MI355X.

Metric (BASELINE.json): Mcells/s for fill -> D8 -> accumulation -> bluespot label (-> watershed, pour points) with the
inputs resident in HBM when the timed region starts, plus the achieved fraction of the HBM roofline per stage (the D8
stencil is the metric's second headline).  A "step" is one pass of the whole chain over the DEM:
    fill (+depths) -> minimum_safe_short_and_diag + no-flats fill -> D8 -> accumulation ->
    connected components + label_stats -> (keep all) -> watersheds + label_count -> pour points (argmax accum)

    python bench.py [--gpus N] [--steps K] [--warmup W] [--size S] [--beta 2.0] [--bands B] [--config 2|3]

Workloads (BASELINE.json `configs`):
  N = 1 (default)    configs[2]: one 16384^2 fBm DEM on one GPU, one device context -- the configuration the metric is quoted on.
  N > 1 (default)    configs[3]: ONE 65536^2 DEM row-banded over the N GPUs: STRONG scaling (total work fixed).  A band context
                     addresses < 2**31 cells (int32 cell indices), so the raster is cut into B = max(N, 4) bands; with N < 4 a
                     process drives B / N bands on its GPU as threads (N = 1 or 2: the 1- and 2-GPU points of the curve; pass
                     --size 65536 to get the N = 1 point).  One band per GPU (N >= 4): halo rows travel GPU -> GPU over RCCL
                     inside the library; several bands per GPU: through the host communicator.
  --config 2         configs[1]: 4096^2, fill + no-flats + D8 only (stencil roofline bring-up), one GPU.
The 65536^2 DEM is a two-octave variant of the SURVEY 8(d) recipe (an FFT of that size does not fit a rank's share of host
memory): 0.5 x the periodic 16384^2 spectral fBm tile (seed 42, tiled 4 x 4) + 0.5 x a 1024^2 spectral fBm (seed 43)
bilinearly upsampled x 64, every rank building only its own rows.

N > 1 is launched by torch.distributed.run (one process per GPU).  torch is used ONLY by this launcher for rank plumbing
(gloo: barrier, max over ranks of the timing, the control-plane collectives of the band protocol and handing the
ncclUniqueId to the ranks); the product package is torch-free.  If the band path fails on any rank the run exits non-zero:
there is no fall-back to independent replicas.

Only the `cpu_baseline` leg touches oracle/ (the single-thread C restatement of the reference path), on a bounded sample,
on rank 0 at N = 1.
"""
import argparse
import json
import os
import sys
import threading
import time
import traceback
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tools"))

# algorithmic (compulsory) HBM bytes per cell, SURVEY.md 8(d) / BASELINE.md section 4
# one request for the whole chain: the bluespot depths (12 B/cell) are computed on the label branch, not in the fill stage
ALG_BYTES = {"fill": 8, "noflat": 12, "flowdir": 9, "accum": 9, "label": 12 + 8 + 8, "watershed": 9 + 4, "pourpoints": 12}
#   fill      : fill 8 B (4 R + 4 W)
#   label     : depths 12 B (4 + 4 R, 4 W) + CCL 8 B (4 R + 4 W) + label_stats 8 B (4 + 4 R)
#   watershed : 9 B (1 + 4 R, 4 W) + label_count 4 B
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); the device-copy ceiling is measured below
STAGES = ["fill", "noflat", "flowdir", "accum", "label", "watershed", "pourpoints"]
ENGINE_NOTE = ("fill_algorithm 1 = tiled priority-flood (0 iterative schedule, 4 flood + iterative repair); noflat_algorithm 2 = integer "
               "geodesic transform (3 + float64 relaxation of irregular flats, 0 float64 relaxation); pour_algorithm 1 = keys out of the "
               "accumulation's final pass (0 pass over accumulation + labels)")
BAND_CELL_LIMIT = 2 ** 30   # cells per band context the bench aims for (hard limit of a context: 2**31 - 2)


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


class DemSource(object):
    """Rows [row0, row0 + nrows) of the benchmark DEM of edge `size`, identical bits on every rank."""
    TILE, COARSE = 16384, 1024

    def __init__(self, size, beta):
        self.size, self.beta = int(size), float(beta)
        self.two_octave = self.size > self.TILE
        if self.two_octave:
            assert self.size % self.TILE == 0 and self.size % self.COARSE == 0, "large DEMs are multiples of 16384"
            self.tile = fbm(self.TILE, beta=beta, seed=42)
            self.coarse = fbm(self.COARSE, beta=beta, seed=43).astype(np.float64)
            self.recipe = ("two-octave fBm beta=%g: 0.5 x periodic %d^2 spectral tile (seed 42) tiled %dx%d + 0.5 x %d^2 spectral "
                           "fBm (seed 43) bilinearly upsampled x%d" % (beta, self.TILE, self.size // self.TILE, self.size // self.TILE,
                                                                      self.COARSE, self.size // self.COARSE))
        else:
            self.full = fbm(self.size, beta=beta, seed=42)
            self.recipe = "spectral fBm beta=%g seed 42 (SURVEY 8d recipe)" % beta

    def rows(self, row0, nrows):
        if not self.two_octave:
            return np.ascontiguousarray(self.full[row0:row0 + nrows])
        n, f, nc = self.size, self.size // self.COARSE, self.COARSE
        out = np.empty((nrows, n), dtype=np.float32)
        cx = np.arange(n, dtype=np.float64) / f
        c0 = np.floor(cx).astype(np.int64) % nc
        c1, ct = (c0 + 1) % nc, (cx - np.floor(cx))[None, :]
        reps = n // self.TILE
        for a in range(0, nrows, 1024):
            r = np.arange(row0 + a, min(row0 + nrows, row0 + a + 1024), dtype=np.int64)
            rx = r.astype(np.float64) / f
            r0 = np.floor(rx).astype(np.int64) % nc
            r1, rt = (r0 + 1) % nc, (rx - np.floor(rx))[:, None]
            lr = (1.0 - rt) * self.coarse[r0] + rt * self.coarse[r1]                 # rows interpolated: len(r) x 1024
            low = (1.0 - ct) * lr[:, c0] + ct * lr[:, c1]                             # columns interpolated: len(r) x n
            det = np.tile(self.tile[r % self.TILE], (1, reps))
            out[a:a + len(r)] = (0.5 * low + 0.5 * det.astype(np.float64)).astype(np.float32)
        return out


KERNEL_OF_STAGE = {"fill": "pf_tile_kernel", "noflat": "ng_round_kernel", "flowdir": "d8s_kernel", "accum": "accum_tile_kernel + perimeter graph + accum_final_walk_kernel", "label": "depths + ccl_* + stats_kernel",
                   "watershed": "ws_* + count_kernel", "pourpoints": "pour_finish_kernel (keys out of accum_final_walk_kernel; arg_packed_kernel for the general pass)"}
# FETCH_SIZE reports half of the bytes of a coalesced streaming read on gfx950 whatever the load width (calibrated per run on kernels
# with known reads: tools/pmc_traffic.py); the committed table carries the corrected figure (`fetch_bytes_per_cell`)
STAGE_KERNELS = {"flowdir": ("d8_kernel", "d8s_kernel"), "fill": ("fill_round_kernel<float", "pf_"), "noflat": ("fill_round_kernel<double", "noflat_", "ng_"),
                 "accum": ("accum_",), "label": ("ccl_", "stats_", "depths_kernel"), "watershed": ("ws_", "count_kernel"),
                 "pourpoints": ("arg_", "pour_")}


FETCH_NOTE = ("FETCH_SIZE x 2 (calibrated on the streaming kernels of the same run, tools/pmc_traffic.py): exact for coalesced "
              "streams, an UPPER BOUND for kernels that gather scattered sectors (ccl_*, ws_*, accum_*, ng_round, pf_ring)")


def lib_fingerprint():
    """sha256 (16 hex digits) over the library's SOURCES (csrc/*.hip, common.hpp, the Makefile, the C-ABI header, in name order): what
    the PMC table records of the library it was collected with.  (The shared object's own bytes differ from build to build --
    hipcc embeds build paths in the code object bundle --, the sources say which kernels ran.)"""
    import hashlib
    h = hashlib.sha256()
    try:
        src = sorted((ROOT / "malstroem_amd" / "csrc").glob("*.hip")) + [ROOT / "malstroem_amd" / "csrc" / "common.hpp",
                                                                         ROOT / "malstroem_amd" / "csrc" / "Makefile", ROOT / "include" / "malstroem_hip.h"]
        for f in src:
            h.update(f.name.encode() + b"\0" + f.read_bytes() + b"\0")
        return h.hexdigest()[:16]
    except Exception:
        return None


def pmc_table(n):
    """(rows, meta, path) of the newest committed rocprofv3 PMC table (profiles/*_pmc_hbm_traffic.json, made by
    tools/pmc_traffic.py: FETCH_SIZE and WRITE_SIZE collected in separate runs of `bench.py --steps 1 --warmup 0` at 16384^2)."""
    tables = sorted((ROOT / "profiles").glob("*_pmc_hbm_traffic.json"))
    if n != 16384 or not tables:
        return None, None, None
    path = tables[-1]
    doc = json.loads(path.read_text())
    if isinstance(doc, dict):                      # round 4 on: {"meta": {...}, "kernels": [...]}
        return doc["kernels"], doc.get("meta", {}), path
    return doc, {}, path


def pmc_traffic(stage, n):
    """HBM bytes of the stage's kernels from the committed PMC table: per STEP for a stage (the table is one step of the chain),
    per LAUNCH for the D8 stencil (every d8s_kernel launch of the profiled command is one full pass over the surface)."""
    rows, meta, path = pmc_table(n)
    if rows is None or stage not in STAGE_KERNELS:
        return None
    tot, launches = 0.0, 0
    for row in rows:
        if row["kernel"].startswith(STAGE_KERNELS[stage]):
            tot += (row["fetch_size_kb"] * 2.0 + row["write_size_kb"]) * 1024.0
            launches += int(row.get("launches", 1))
    if tot == 0.0:
        return None
    fp = lib_fingerprint()
    out = {"source": "profiles/" + path.name, "fetch_correction": FETCH_NOTE,
           "table_library": meta.get("library_sources_sha256_16"), "table_matches_this_library": bool(fp and meta.get("library_sources_sha256_16") == fp)}
    if stage == "flowdir":
        tot /= max(launches, 1)
        out.update({"hbm_bytes_per_launch": round(tot), "bytes_per_cell": round(tot / (float(n) * n), 2), "launches_in_table": launches})
    else:
        tot /= max(int(meta.get("steps", 1)), 1)
        out.update({"hbm_bytes_per_step": round(tot), "bytes_per_cell": round(tot / (float(n) * n), 2)})
    return out


def host_cpu():
    model = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return model, os.cpu_count() or 0


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
    model, ncores = host_cpu()
    return {"value": round(crop.size / dt / 1e6, 3), "unit": "Mcells/s", "cores": 1, "kind": "port",
            "host_cpu": model, "host_cores": ncores,
            "sample": "%dx%d crop of the benchmark DEM, full chain, %.1f s, 1 thread (the reference path is single-threaded)" % (sample, sample, dt)}


class BandWorker(threading.Thread):
    """One band of the row-banded DEM: owns a BandPipeline, runs one chain per step on request."""

    def __init__(self, comm, shape, device, rccl, dem_rows_fn, gate):
        super(BandWorker, self).__init__(daemon=True)
        self.comm, self.shape, self.device, self.rccl, self.dem_rows_fn, self.gate = comm, shape, device, rccl, dem_rows_fn, gate
        self.error, self.pipe, self.timings, self.host_ms, self.stop = None, None, {}, {}, False

    def run(self):
        from malstroem_amd.distributed import BandPipeline
        try:
            self.pipe = BandPipeline(self.comm, self.shape, device=self.device, rccl=self.rccl)
            self.pipe.upload_dem(self.dem_rows_fn(self.pipe.row0, self.pipe.nrows))
        except Exception as e:
            traceback.print_exc()
            self.error = e
        self.gate.wait()                      # setup done (or failed) on every band of this process
        while True:
            self.gate.wait()                  # start of a step (or stop)
            if self.stop:
                break
            if self.error is None:
                try:
                    tm = {}
                    self.pipe.host_ms = {}
                    self.pipe.run_chain(records=True, fetch_own=False, overlap=True, timings=tm)
                    for k, v in tm.items():
                        self.timings[k] = self.timings.get(k, 0.0) + v
                    for k, v in self.pipe.host_ms.items():      # thread CPU time of the host-only sections (boundary systems)
                        self.host_ms[k] = self.host_ms.get(k, 0.0) + v
                except Exception as e:
                    traceback.print_exc()
                    self.error = e
            self.gate.wait()                  # end of the step
        if self.pipe is not None:
            self.pipe.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--size", type=int, default=0, help="DEM edge; default 16384 on one GPU, 65536 on several")
    ap.add_argument("--beta", type=float, default=2.0)
    ap.add_argument("--bands", type=int, default=0, help="row bands in total (default: as many as the int32 cell domain needs, >= N)")
    ap.add_argument("--config", type=int, default=3, choices=(2, 3), help="2: 4096^2, fill + D8 only (BASELINE configs[1])")
    ap.add_argument("--cpu-sample", type=int, default=8192)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--check", action="store_true",
                    help="band runs: after the timed steps compare every band's rasters (sha256 of its owned rows) and the label count with ONE "
                         "undivided context on rank 0 (the DEM must fit one context: < 2**31 cells); exit code 6 on a difference")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    if world > 1:
        import torch.distributed as dist  # launcher plumbing only (gloo, CPU tensors)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)

    from malstroem_amd import _lib
    from malstroem_amd.pipeline import HydroPipeline
    if _lib.device_count() < 1:
        raise SystemExit("bench.py needs an MI355X: libmalstroem_hip has no CPU fallback")
    ndev = _lib.device_count()
    device = local_rank % ndev

    config2 = args.config == 2
    n = args.size or (4096 if config2 else (16384 if world == 1 else 65536))
    cells = float(n) * n
    nbands = args.bands or max(world, -(-int(cells) // BAND_CELL_LIMIT))
    nbands = -(-nbands // world) * world
    band_mode = (world > 1 or nbands > 1) and not config2
    stage_names = ["fill", "noflat", "flowdir"] if config2 else STAGES

    t_gen = time.perf_counter()
    src = DemSource(n, args.beta)
    t_gen = time.perf_counter() - t_gen

    stage_ms = {s: 0.0 for s in stage_names}
    hot_ms, hot_launches = {"fill": 0.0, "noflat": 0.0}, {"fill": 0, "noflat": 0}
    pipe, workers, gate = None, [], None
    parallelism = "1 GPU, one device context"
    if band_mode:
        from launch_comm import TorchComm
        from malstroem_amd.distributed import HybridComm, SingleComm
        k = nbands // world
        proc_comm = TorchComm() if world > 1 else SingleComm()
        control_plane = "torch.distributed (gloo)" if world > 1 else "one process"
        if world > 1 and os.environ.get("MALSTROEM_CONTROL_PLANE", "shm") != "gloo":
            # the ranks of one node (the launch contract: --nnodes=1): votes, neighbour rows of the boundary systems and the small
            # all-gathers through a shared-memory segment instead of TCP on the loopback interface (set up through gloo; ranks that do
            # not share rank 0's host keep gloo, all of them together)
            from malstroem_amd.distributed import ShmComm
            proc_comm = ShmComm.over(proc_comm)
            if isinstance(proc_comm, ShmComm):
                control_plane = "shared-memory segment of the node (malstroem_amd.distributed.ShmComm), set up over gloo"
        host_rows_asked = os.environ.get("MALSTROEM_BAND_TRANSPORT", "rccl") == "host"     # explicit override: rows through the host communicator
        rccl = k == 1 and world > 1 and world <= ndev and not host_rows_asked
        comms = [proc_comm] if k == 1 else HybridComm.world(proc_comm, k)
        gate = threading.Barrier(k + 1)
        workers = [BandWorker(c, (n, n), device, rccl, src.rows, gate) for c in comms]
        for w in workers:
            w.start()
        gate.wait()
        used_rccl = bool(workers[0].pipe is not None and workers[0].pipe.rccl)
        rccl_ranks = world if used_rccl else 0
        if rccl and not used_rccl:
            # one band per GPU and the rows are NOT on RCCL: what a scaling run would measure is the host communicator.  Refuse
            # (MALSTROEM_BAND_TRANSPORT=host asks for that configuration explicitly and is reported as such).
            if rank == 0:
                print("bench.py: RCCL is not carrying the band rows (%s); set MALSTROEM_BAND_TRANSPORT=host to measure the host path on purpose"
                      % (workers[0].pipe.rccl_error if workers[0].pipe is not None else "band setup failed"), file=sys.stderr)
            for w in workers:
                w.error = w.error or RuntimeError("RCCL transport unavailable")
        rccl_note = ""
        if rccl and not used_rccl and workers[0].pipe is not None:
            rccl_note = " (RCCL setup failed: %s)" % workers[0].pipe.rccl_error
        parallelism = "%d row bands of one %dx%d DEM over %d GPU(s), %d band(s) per GPU; halo rows %s" % (
            nbands, n, n, world, k, "GPU->GPU over RCCL inside the library (ncclSend/ncclRecv)" if used_rccl else "through the host communicator" + rccl_note)
    else:
        pipe = HydroPipeline((n, n), device=device)
        pipe.upload("dem", src.rows(0, n))

    def failed():
        bad = 1.0 if any(w.error is not None for w in workers) else 0.0
        if dist is not None:
            import torch
            t = torch.tensor([bad], dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            bad = float(t[0])
        return bad > 0.0

    def shutdown(code):
        for w in workers:
            w.stop = True
        if workers:
            gate.wait()
            for w in workers:
                w.join(60)
        if pipe is not None:
            pipe.close()
        if band_mode and hasattr(proc_comm, "close") and not isinstance(proc_comm, TorchComm):
            try:
                proc_comm.close()      # (the shared-memory control plane: rank 0 removes the segment)
            except Exception:
                pass
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        sys.exit(code)

    if failed():
        if rank == 0:
            print("bench.py: band setup failed on at least one rank (no replica fall-back); see the tracebacks above", file=sys.stderr)
        shutdown(3)

    nsampled = [0]
    band_step_ms = []

    def step(record, sample=True):
        if band_mode:
            t_s = time.perf_counter()
            gate.wait()     # go
            gate.wait()     # all bands of this process are through
            if record:
                band_step_ms.append(round((time.perf_counter() - t_s) * 1e3, 1))      # (this process's bands; the line's time is the span of all steps, max over ranks)
        else:
            # one request for the whole chain: the library runs the bluespot branch (label, watershed) on a second
            # stream next to no-flats fill -> D8 -> accumulation ("stage DAG"); all labels are kept
            pipe.run(*stage_names)
            pipe.sync()
            if record:
                for k in engines_seen:            # EVERY timed step, not just the last one: an intermittent fall-back must not hide
                    engines_seen[k].add(pipe.get_int(k))
            if record and sample:
                # (the stage times of about eight of the timed steps, evenly spaced: fourteen ctypes calls cost ~80 us of a 22 ms step
                # when every step pays them -- the timeline showed the device idle that long between two steps)
                nsampled[0] += 1
                for s in stage_names:
                    stage_ms[s] += pipe.stage_ms(s)
                for s in ("fill", "noflat"):      # the stage's dominant kernel, HIP events around its launches inside the library
                    hot_ms[s] += pipe.get_float(s + "_hot_ms")
                    hot_launches[s] = pipe.get_int(s + "_hot_launches")

    engines_seen = {"fill_algorithm": set(), "noflat_algorithm": set()}
    # development knob of the library (MHIP_DEVELOPER=1 MHIP_SERIAL=1): every stage on the context's stream, nothing overlapped --
    # the per-stage times then mean "alone on the GPU"; the line carries "streams": "serial"
    serial_streams = os.environ.get("MHIP_DEVELOPER", "") == "1" and os.environ.get("MHIP_SERIAL", "")[:1] == "1"
    if not config2:
        engines_seen["pour_algorithm"] = set()

    def barrier():
        if pipe is not None:
            pipe.sync()
        if dist is not None:
            dist.barrier()

    for _ in range(args.warmup):
        step(False)
    for w in workers:
        w.timings, w.host_ms = {}, {}
    barrier()
    t0 = time.perf_counter()
    stride = max(1, args.steps // 8)
    for i in range(args.steps):
        step(True, i % stride == 0)
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        import torch
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t[0])
    # A silent engine fall-back (capacity overflow of the flood, a rejected geodesic transform, pour points by the general pass)
    # gives right results 2-3x slower: the line must not publish such a run as the design's number.  MALSTROEM_BENCH_ALLOW_FALLBACK=1
    # reports it anyway (deliberate A/B runs; DEMs other than the default ones may legitimately take another engine).
    allow_fallback = os.environ.get("MALSTROEM_BENCH_ALLOW_FALLBACK", "") == "1"
    fell_back = []
    if pipe is not None:
        want = {"fill_algorithm": (1,), "noflat_algorithm": (2,)}
        if not config2 and not serial_streams:      # (one stream: the pour points are a pass of their own by design, not a fall-back)
            want["pour_algorithm"] = (1,)
        fell_back = ["%s=%s" % (k, sorted(engines_seen[k])) for k, ok in want.items() if not engines_seen[k] <= set(ok)]
    for w in workers:
        if w.error is None and w.pipe is not None:
            e = w.pipe.engines()
            if e.get("fill") not in (1, None) or e.get("noflat") not in (2, 3, None):
                fell_back.append("band %d: %r" % (w.pipe.comm.rank, e))
    if fell_back and not allow_fallback:
        print("bench.py: rank %d: a stage fell back to its slow engine (%s); no metric is reported "
              "(MALSTROEM_BENCH_ALLOW_FALLBACK=1 reports it anyway)" % (rank, ", ".join(fell_back)), file=sys.stderr)
        if workers:
            workers[0].error = workers[0].error or RuntimeError("engine fall-back")
        else:
            shutdown(5)
    if failed():
        if rank == 0:
            print("bench.py: the band chain failed (or fell back to a slow engine) on at least one rank; no metric is reported", file=sys.stderr)
        shutdown(4)

    check_note = None
    if args.check and band_mode:
        # the launch shapes the driver uses (N processes x k bands) rehearsed at a size one context holds: every band's rasters against
        # the undivided result, bit for bit
        import hashlib
        names = ("filled", "depths", "noflat", "flowdir", "accum", "labels", "watersheds")
        mine = []
        for w in workers:
            p = w.pipe
            mine.append((int(p.row0), int(p.nrows), int(p.nlabels), {k: hashlib.sha256(np.ascontiguousarray(p.download(k)).tobytes()).hexdigest() for k in names}))
        everyone = [mine]
        if dist is not None:
            everyone = [None] * world
            dist.all_gather_object(everyone, mine)
        bad = []
        if rank == 0:
            assert cells < 2 ** 31 - 2, "--check needs a DEM that fits one context"
            with HydroPipeline((n, n), device=device) as one:
                one.upload("dem", src.rows(0, n))
                one.run(*stage_names)
                one.sync()
                nlab = one.get_int("nlabels")
                whole = {k: one.download(k) for k in names}
            for per_rank in everyone:
                for row0, nrows, nl, digests in per_rank:
                    if nl != nlab:
                        bad.append("rows %d..%d: %d labels, one context %d" % (row0, row0 + nrows, nl, nlab))
                    for k in names:
                        if hashlib.sha256(np.ascontiguousarray(whole[k][row0:row0 + nrows]).tobytes()).hexdigest() != digests[k]:
                            bad.append("rows %d..%d: %s differs from the undivided context" % (row0, row0 + nrows, k))
            check_note = "%d bands x 7 rasters + label count == one undivided context" % nbands if not bad else "; ".join(bad)
        flag = 1.0 if bad else 0.0
        if dist is not None:
            import torch
            t = torch.tensor([flag], dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            flag = float(t[0])
        if flag:
            if rank == 0:
                print("bench.py --check: the band run differs from one undivided context: %s" % check_note, file=sys.stderr)
            shutdown(6)

    ms_per_step = elapsed * 1e3 / args.steps
    value = cells * args.steps / elapsed / 1e6           # ONE DEM whatever N is: whole-job cells per second
    # The D8 stencil on its own (BASELINE's second figure): in the chain the no-flats fill's finishing pass writes the flow directions
    # from the surface it holds in registers and the `flowdir` stage is an empty interval, so the kernel is timed as a request of
    # its own on the resident surface, after the timed steps (HIP events on its stream, like every stage).
    d8_alone_ms = d8_single_ms = None
    if pipe is not None:
        # steady-state stencil throughput: 16 launches back to back between ONE pair of HIP events on the kernel's stream
        # (mhip_ctx_kernel_ms "d8_steady"), best-of-3 is NOT taken: the mean of three such batches
        acc = 0.0
        for _ in range(3):
            ms, nl = pipe.kernel_ms("d8_steady")
            acc += ms / nl
        d8_alone_ms = acc / 3
        # ... and, for comparison, one request bracketed by its own pair of events (what round 3 reported: + the event bracket)
        reps, acc = 5, 0.0
        for _ in range(reps):
            for _ in range(4):
                pipe.run("flowdir")
            pipe.sync()
            acc += pipe.stage_ms("flowdir")
        d8_single_ms = acc / reps
    if band_mode:
        for s in stage_names:
            stage_ms[s] = max(w.timings.get(s, 0.0) for w in workers) / args.steps   # host wall clock per stage, slowest band of this rank
        host_serial = {}
        for w in workers:
            for k2, v in w.host_ms.items():
                host_serial[k2] = max(host_serial.get(k2, 0.0), v / args.steps)
        # which transport carries the rows of each seam (seam s lies between band s and band s + 1; band b runs on rank b // k)
        seam_transports = []
        kb = nbands // world      # bands per process
        for sm in range(nbands - 1):
            if sm // kb == (sm + 1) // kb:
                seam_transports.append("memory of process %d (two bands of one process)" % (sm // kb))
            elif used_rccl:
                seam_transports.append("RCCL, rank %d <-> %d (ncclSend / ncclRecv on the bands' streams)" % (sm // kb, (sm + 1) // kb))
            else:
                seam_transports.append("host communicator (gloo point-to-point), process %d <-> %d" % (sm // kb, (sm + 1) // kb))
        info = {"nlabels": workers[0].pipe.nlabels, "halo_exchanges": dict(workers[0].pipe.exchanges), "bands": nbands,
                "seam_transports": seam_transports, "control_plane": control_plane,
                "band_engines": [w.pipe.engines() for w in workers], "engines": ENGINE_NOTE,
                "step_ms_rank0": band_step_ms, "rccl_ranks": rccl_ranks, "rccl_side_communicator": bool(workers[0].pipe.rccl_side), "check": check_note,
                # thread CPU time per step of the host-only sections of the boundary systems (label / accumulation / watershed
                # seams), maximum over the bands of rank 0: what does not shrink with the number of GPUs
                "host_serial_ms": {k2: round(v, 2) for k2, v in sorted(host_serial.items())},
                "host_serial_ms_total": round(sum(host_serial.values()), 2)}
    else:
        for s in stage_names:
            stage_ms[s] /= max(nsampled[0], 1)
        keys = ("fill_rounds", "noflat_rounds", "fill_tiles", "fill_visits", "fill_cycles", "noflat_visits", "noflat_cycles") + (() if config2 else ("nlabels",))
        info = {k: pipe.get_int(k) for k in keys}
        for k in ("fill_algorithm", "fill_launches", "noflat_algorithm") + (() if config2 else ("pour_algorithm",)):
            info[k] = pipe.get_int(k)
        info["engines_seen_in_timed_steps"] = {k: sorted(v) for k, v in engines_seen.items()}
        info["engines"] = ENGINE_NOTE
        if info["noflat_algorithm"] not in (2, 3):
            info["noflat_reject"] = {k: pipe.get_int("noflat_reject" + k) for k in ("", "_irregular", "_unreached", "_mismatch")}
    if rank == 0:
        stages = {}
        for s in stage_names:
            gbs = ALG_BYTES[s] * cells / (stage_ms[s] * 1e-3) / 1e9 if stage_ms[s] > 0 else 0.0
            stages[s] = {"ms": round(stage_ms[s], 3), "alg_bytes_per_cell": ALG_BYTES[s], "achieved_GBs": round(gbs, 1),
                         "frac_of_hbm_peak": round(gbs / (HBM_PEAK_GBS * world), 4)}
        # The chain runs fill -> no-flats fill -> D8 alone on the GPU, then {accumulation} next to {labelling -> watersheds} and the
        # pour points: the event times of those tail stages include the time they waited for CUs the other branch held (alone:
        # accumulation 5.6, labelling 5.5, watersheds 3.6 ms at 16384^2).  The dominant stage is taken among the stages
        # that had the GPU to themselves, whose time means the same thing in the chain and alone.
        overlapped = () if (band_mode or config2 or serial_streams) else ("accum", "label", "watershed", "pourpoints")
        for s in overlapped:
            stages[s]["overlapped"] = True
        dominant = max((s for s in stage_names if s not in overlapped), key=lambda s: stage_ms[s])
        d8 = dict(stages["flowdir"])
        if d8_alone_ms:
            if stage_ms["flowdir"] < 0.5 * d8_alone_ms:
                stages["flowdir"].update({"achieved_GBs": None, "frac_of_hbm_peak": None,
                                          "fused_into": "noflat (ng_finish_kernel writes the directions; d8_roofline times the stencil on its own)"})
            gbs = ALG_BYTES["flowdir"] * cells / (d8_alone_ms * 1e-3) / 1e9
            d8 = {"ms": round(d8_alone_ms, 4), "achieved_GBs": round(gbs, 1), "frac_of_hbm_peak": round(gbs / HBM_PEAK_GBS, 4)}
        if "pourpoints" in stages and info.get("pour_algorithm") == 1:
            # the 12 B/cell pass over accumulation + labels does not run: the keys come out of the accumulation's final pass and the
            # stage is pour_finish_kernel over one key per label
            stages["pourpoints"].update({"achieved_GBs": None, "frac_of_hbm_peak": None, "alg_bytes_per_cell": None,
                                         "fused_into": "accum (accum_final_walk_kernel writes one key per label; this stage turns keys into records)"})
        copy_gbs = read_gbs = None
        try:
            copy_gbs = round(pipe.copy_bandwidth(), 1) if pipe is not None else None
            read_gbs = round(pipe.read_bandwidth(), 1) if pipe is not None else None
        except Exception:
            copy_gbs = read_gbs = None
        chain = "fill -> no-flats fill -> D8" if config2 else ("fill+depths -> no-flats fill -> D8 -> accumulation -> CCL+label_stats -> "
                                                               "watersheds+label_count -> pour points")
        launches_key = {"fill": "fill_launches" if "fill_launches" in info else "fill_rounds", "noflat": "noflat_rounds"}.get(dominant, "")

        def kernel_detail(stage):
            """the dominant KERNEL of the dominant stage: its own device time (HIP events around its launches, live in this run) and
            its own HBM bytes (the committed PMC table of the same command), next to the stage figure"""
            if band_mode or stage not in hot_ms or hot_ms[stage] <= 0.0:
                return None
            kms = hot_ms[stage] / max(nsampled[0], 1)
            out_ = {"name": KERNEL_OF_STAGE[stage], "launches_per_step": hot_launches[stage], "ms_per_step": round(kms, 3),
                    "avg_launch_us": round(1e3 * kms / max(hot_launches[stage], 1), 1), "timed_by": "HIP events around the launches, on their stream"}
            rows, meta, path = pmc_table(n)
            fp = lib_fingerprint()
            if rows is not None and fp and meta.get("library_sources_sha256_16") == fp:
                # counter-based figures (HBM bytes the kernel MOVED / its time), only when the table was collected with this very
                # library: they are traffic, not the algorithmic fraction the `frac` keys of this line carry
                for row in rows:
                    if row["kernel"].startswith(KERNEL_OF_STAGE[stage]):
                        own = (2.0 * row["fetch_size_kb"] + row["write_size_kb"]) * 1024.0 / max(int(meta.get("steps", 1)), 1)
                        out_.update({"traffic_bytes_per_step": round(own), "traffic_bytes_per_cell": round(own / cells, 2),
                                     "traffic_GBs": round(own / (kms * 1e-3) / 1e9, 1),
                                     "traffic_frac_of_hbm_peak": round(own / (kms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                     "traffic_source": "profiles/" + path.name, "fetch_correction": FETCH_NOTE})
                        break
            elif rows is not None:
                out_["traffic_note"] = "profiles/%s was collected with another build of the library (%s != %s): no counter figures" % (
                    path.name, meta.get("library_sources_sha256_16"), fp)
            return out_
        out = {
            "metric": ("Mcells/s fill->D8 on %d^2 f32 DEM" if config2 else "Mcells/s fill->D8->accum->label on %d^2 f32 DEM") % n,
            "value": round(value, 2), "unit": "Mcells/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "strong" if band_mode else "single",
            "vs_baseline": None,
            "dtype": "f32 fill / f64 no-flats+D8+accum / u8 flowdir / i32 labels", "data": "synthetic",
            "config": {"workload": "%dx%d float32 DEM (%s): %s" % (n, n, src.recipe, chain),
                       "parallelism": parallelism, "dem_generation_s": round(t_gen, 1),
                       **({"streams": "serial (MHIP_SERIAL=1: every stage alone on the GPU, the pour points as a pass of their own)"} if serial_streams else {}),
                       **info},
            "roofline": {"bound": "hbm", "kernel": KERNEL_OF_STAGE.get(dominant, dominant), "stage": dominant,
                         "kernel_detail": kernel_detail(dominant),
                         "launches_per_step": info.get(launches_key, 1),
                         "achieved": stages[dominant]["achieved_GBs"], "peak": HBM_PEAK_GBS * world,
                         "unit": "GB/s", "frac": stages[dominant]["frac_of_hbm_peak"], "traffic": pmc_traffic(dominant, n) if not band_mode else None,
                         "measured_copy_peak_GBs": copy_gbs, "measured_read_peak_GBs": read_gbs,
                         "note": "dominant stage by %s; algorithmic bytes of the whole stage / stage time" % (
                             "host wall clock of the slowest band (rank 0)" if band_mode else "device time (HIP events on the stage's stream)")},
            "d8_roofline": {"bound": "hbm", "kernel": "d8s_kernel", "ms": d8["ms"],
                            "timed_by": "16 launches back to back between one pair of HIP events on the kernel's stream, mean of 3 batches",
                            "ms_single_request": round(d8_single_ms, 4) if d8_single_ms else None,
                            "achieved": d8["achieved_GBs"], "peak": HBM_PEAK_GBS * world,
                            "unit": "GB/s", "frac": d8["frac_of_hbm_peak"], "frac_of_measured_copy_peak": round(d8["achieved_GBs"] / copy_gbs, 4) if copy_gbs else None,
                            "frac_of_measured_read_peak": round(d8["achieved_GBs"] / read_gbs, 4) if read_gbs else None,
                            "traffic": pmc_traffic("flowdir", n) if not band_mode else None},
            "stages": stages, "stage_times_sampled_steps": nsampled[0] if not band_mode else None,
        }
        if world == 1 and not band_mode and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(src.full, min(args.cpu_sample, n))
        print(json.dumps(out), flush=True)
    shutdown(0)


if __name__ == "__main__":
    main()
