"""lab: C undivided contexts of the benchmark DEM run the whole chain at the same time on ONE device (a host thread each): how much
of a step's time is latency the device could fill with a second DEM?  usage: python tools/lab/concurrent_dems.py C [size] [steps]"""
import sys
import threading
import time

sys.path.insert(0, ".")
from bench import DemSource
from malstroem_amd.pipeline import HydroPipeline

C = int(sys.argv[1])
n = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
src = DemSource(n, 2.0)
dem = src.rows(0, n)
pipes = [HydroPipeline((n, n), device=0) for _ in range(C)]
for p in pipes:
    p.upload("dem", dem)
names = ["fill", "noflat", "flowdir", "accum", "label", "watershed", "pourpoints"]
gate = threading.Barrier(C + 1)
stop = []


def work(p):
    while True:
        gate.wait()
        if stop:
            return
        p.run(*names)
        p.sync()
        gate.wait()


threads = [threading.Thread(target=work, args=(p,), daemon=True) for p in pipes]
[t.start() for t in threads]
for _ in range(2):
    gate.wait(); gate.wait()
t0 = time.perf_counter()
for _ in range(steps):
    gate.wait(); gate.wait()
dt = (time.perf_counter() - t0) / steps
stop.append(1)
gate.wait()
print("%d DEMs of %d^2 at a time: %.2f ms per round = %.2f ms per DEM = %.0f Mcells/s; engines %s" % (
    C, n, dt * 1e3, dt * 1e3 / C, C * n * n / dt / 1e6, [(p.get_int("fill_algorithm"), p.get_int("noflat_algorithm")) for p in pipes]))
