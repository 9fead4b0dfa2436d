"""Soak test (GPU): many rt_run_batch calls on changing batches; device memory must stay flat after warm-up."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import retto_amd
from retto_amd import workload

s = retto_amd.RettoSession(retto_amd.synthetic_session_config(0))
rng = np.random.default_rng(0)
sizes = [(640, 640), (960, 960), (720, 1280), (416, 608)]
pages, maps = [], []
for i in range(24):
    h, w = sizes[i % len(sizes)]
    p, r = workload.planted_page(h, w, 4 + i % 9, seed=i)
    dh, dw = s.det_preprocess(p).shape[2:]
    pages.append(p); maps.append(workload.planted_map(dh, dw, h, w, r))
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 400
free0 = None
t0 = time.time()
for it in range(steps):
    k = int(rng.integers(1, 17))
    idx = rng.choice(len(pages), k, replace=False)
    res = s.run_batch([pages[i] for i in idx], det_map_override=[maps[i] for i in idx])
    assert all(len(r.det_result) > 0 for r in res)
    if it % 25 == 24:
        free, total = torch.cuda.mem_get_info()
        if it >= 199 and free0 is None:  # (the arenas reach their high-water mark once the largest batch combinations have been seen)
            free0 = free
        print("step %4d  free %.2f GB  (%.1f s)" % (it + 1, free / 2**30, time.time() - t0), flush=True)
free, _ = torch.cuda.mem_get_info()
assert free0 is None or abs(free - free0) < 512 * 2**20, "device memory drifted by %.1f MB" % ((free0 - free) / 2**20)
print("soak ok")
