"""File-inclusive rate of the C3 workload (GPU only): the 32 planted 960x960 pages are held as encoded PNG (or JPEG)
files in memory; batch i+1 is decoded on host threads (rt_decode_image) while batch i runs on the GPU
(rt_run_batch with host pages, i.e. PCIe-inclusive).  Prints images/s for device-resident pages, host pages and
encoded files.

    python tools/bench_files.py [png|jpeg] [steps] [threads]
"""
import io, os, sys, time
from concurrent.futures import ThreadPoolExecutor
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from PIL import Image
import retto_amd
from retto_amd import workload

fmt = (sys.argv[1] if len(sys.argv) > 1 else "png").upper()
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
threads = int(sys.argv[3]) if len(sys.argv) > 3 else 16
s = retto_amd.RettoSession(retto_amd.synthetic_session_config(0))
files, maps = [], []
for i in range(32):
    page, rects = workload.planted_page(960, 960, 32, seed=i)
    b = io.BytesIO(); Image.fromarray(page).save(b, fmt, **({"quality": 90} if fmt == "JPEG" else {}))
    files.append(b.getvalue())
    maps.append(workload.planted_map(960, 960, 960, 960, rects))
print("%s: %.0f KB per page" % (fmt, sum(len(f) for f in files) / 32 / 1024))
pool = ThreadPoolExecutor(threads)
decode = lambda: list(pool.map(retto_amd.decode_image, files))
t = time.time(); pages = decode(); print("decode of 32 pages on %d threads: %.1f ms" % (threads, (time.time() - t) * 1e3))
import ctypes as C
lib, h = s._hd.lib, s._hd.h
d_pages = []
for pg in pages:
    p = C.c_void_p()
    assert lib.rt_device_malloc(h, pg.nbytes, C.byref(p)) == 0 and lib.rt_memcpy_h2d(h, p, pg.ctypes.data, pg.nbytes) == 0
    d_pages.append(p.value)
hs = [960] * 32
# No planted-map override here (it would add a 3.7 MB f32 upload per page that the real path does not have): the
# boxes come from the random-weight detector's own map, so only the three rates below are comparable with each other.
def timed(fn, label):
    for _ in range(3):
        fn()
    t = time.time()
    for _ in range(steps):
        fn()
    print("%-44s %.1f images/s" % (label, 32 * steps / (time.time() - t)))
timed(lambda: lib.rt_results_free(s.run_batch_raw(d_pages, hs, hs, retto_amd.RT_MEM_DEVICE, None)), "pages resident in HBM:")
timed(lambda: s.run_batch(pages), "host pages (PCIe-inclusive):")
t = time.time()
fut = pool.submit(decode)
for i in range(steps):
    pages = fut.result()
    fut = pool.submit(decode)  # next batch decodes while this one runs
    s.run_batch(pages)
fut.result()
print("%-44s %.1f images/s" % ("encoded files, decode overlapped with the GPU:", 32 * steps / (time.time() - t)))
s.close()
