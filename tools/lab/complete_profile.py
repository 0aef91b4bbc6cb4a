"""lab: `complete.process_all` on a synthetic DEM (n x n fBm written as a GeoTIFF first) with a bluespot filter and two rain events --
for `rocprofv3 --kernel-trace --stats`: which kernels does the f1-f4 path (tools, filter, stream walk, rain events, raster I/O) spend
device time in, and how long does the whole sequence take on the host?   usage: python tools/lab/complete_profile.py [n] [outdir]"""
import os
import sys
import tempfile
import time

sys.path.insert(0, ".")
from bench import fbm
from malstroem_amd.complete import process_all
from malstroem_amd.io import RasterWriter

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
out = sys.argv[2] if len(sys.argv) > 2 else tempfile.mkdtemp()
import shutil
shutil.rmtree(out, ignore_errors=True)
os.makedirs(out)
dem = fbm(n, beta=2.0, seed=42) * 20.0          # (metres: depths of centimetres to metres, so that the filter keeps something)
src = os.path.join(tempfile.mkdtemp(), "dem.tif")      # (process_all wants an empty output directory)
RasterWriter(src, (500000.0, 0.4, 0.0, 6200000.0, 0.0, -0.4), None, nodata=-9999.0).write(dem)
t0 = time.perf_counter()
res = process_all(src, out, [10, 100], accum=True, filter="maxdepth > 0.05 and volume > 0.5")
dt = time.perf_counter() - t0
print("complete on %d^2: %.2f s host wall clock, %d bluespots kept" % (n, dt, res["nlabels"]))
