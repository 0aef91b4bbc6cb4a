"""GPU: the launch shapes of the scaling bench, rehearsed on ONE device.

The driver starts ``bench.py --gpus N`` as N processes under ``torch.distributed.run`` on an 8-GPU node; a one-GPU box cannot
run that over xGMI, but it can run the same processes, the same ``nbands`` / bands-per-process arithmetic, the same votes and
the same control plane with every rank on device 0 and the rows through the host communicator (``bench.py`` picks that
transport by itself when there are more ranks than devices).  ``--check`` then compares every band's rasters and the label
count with one undivided context, bit for bit, and the engine gate of the bench (flood / geodesic transform, no silent
fall-back) is live as in the real run.  At most 4 ranks: the box admits 6 processes on its card, this one included.
"""
import json
import os
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(900)]

ROOT = Path(__file__).resolve().parents[1]


@pytest.mark.parametrize("nproc,bands", [(4, 4), (4, 8), (2, 4)], ids=["gpus4_one_band_each", "eight_bands_on_4_ranks", "gpus2_two_bands_each"])
def test_bench_launch_shape_equals_one_context(nproc, bands, tmp_path):
    port = 29600 + (os.getpid() + 7 * nproc + bands) % 300
    env = dict(os.environ, MALSTROEM_BAND_TRANSPORT="host", OMP_NUM_THREADS="2")
    env.pop("MHIP_DEVELOPER", None)       # the product configuration: no development knobs
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(ROOT / "bench.py"), "--gpus", str(nproc), "--size", "2048", "--bands", str(bands), "--steps", "1",
           "--warmup", "1", "--no-cpu-baseline", "--check"]
    out = subprocess.run(cmd, env=env, cwd=str(ROOT), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    err = out.stderr.decode(errors="replace")
    assert out.returncode == 0, err[-3000:]
    lines = [ln for ln in out.stdout.decode().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout.decode()[-2000:]
    line = json.loads(lines[0])
    cfg = line["config"]
    assert line["n_gpus"] == nproc and cfg["bands"] == bands and line["scaling"] == "strong"
    assert cfg["rccl_ranks"] == 0                                  # one device: the rows went through the host communicator, and the line says so
    seams = cfg["seam_transports"]                                 # ... seam by seam: in memory between two bands of one process, gloo between processes
    k = bands // nproc
    assert len(seams) == bands - 1
    for sm, text in enumerate(seams):
        assert text.startswith("memory of process") if sm // k == (sm + 1) // k else text.startswith("host communicator"), seams
    assert cfg["check"].endswith("== one undivided context"), cfg["check"]
    assert all(e["fill"] == 1 and e["noflat"] in (2, 3) for e in cfg["band_engines"]), cfg["band_engines"]
