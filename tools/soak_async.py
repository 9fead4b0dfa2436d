"""Soak test of the submit-ahead path (GPU): random batches of changing composition submitted 1-4 ahead (rt_submit_batch), waited for
in random order (rt_wait_batch); every page's discrete results (boxes, labels, token ids) must equal the ones the same page
gives in a synchronous call of its own, whatever batch / lane / position it was in; device memory must stay flat after warm-up.

    python tools/soak_async.py [iterations]
"""
import ctypes as C, hashlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import retto_amd
from retto_amd import workload

s = retto_amd.RettoSession(retto_amd.synthetic_session_config(0))
lib = s._hd.lib
rng = np.random.default_rng(1)
sizes = [(640, 640), (960, 960), (720, 1280), (416, 608), (320, 480)]
if os.environ.get("SOAK_BIG"):   # C4-like: A4 scans at 300 dpi and mixed small pages
    sizes = [(3508, 2480), (960, 960), (2100, 1300), (640, 640), (1984, 1408)]
pages, maps = [], []
for i in range(30 if not os.environ.get("SOAK_BIG") else 15):
    h, w = sizes[i % len(sizes)]
    p, r = workload.planted_page(h, w, 3 + i % 9, seed=100 + i)
    dh, dw = s.det_preprocess(p).shape[2:]
    pages.append(p); maps.append(workload.planted_map(dh, dw, h, w, r))


def digest(r, i):
    n = lib.rt_results_count(r, i)
    hsh = hashlib.sha256()
    if n:
        hsh.update(np.ctypeslib.as_array(lib.rt_results_boxes(r, i), (n, 8)).tobytes())
        hsh.update(np.ctypeslib.as_array(lib.rt_results_cls_labels(r, i), (n,)).tobytes())
        for k in range(n):
            tp = C.POINTER(C.c_int32)()
            nt = lib.rt_results_rec_tokens(r, i, k, C.byref(tp))
            hsh.update(bytes(np.ctypeslib.as_array(tp, (nt,)).tobytes()) if nt else b"-")
    return hsh.hexdigest()[:16], n


lib.rt_results_det_checksum.restype = C.c_double
lib.rt_results_det_checksum.argtypes = [C.c_void_p]
ref, ref_sum = [], []   # (the det network's output reaches the results only through the map checksum: the boxes come from the planted maps)
for p, m in zip(pages, maps):
    r = s.run_batch_raw([p], [p.shape[0]], [p.shape[1]], retto_amd.RT_MEM_HOST, [m])
    ref.append(digest(r, 0)); ref_sum.append(lib.rt_results_det_checksum(r)); lib.rt_results_free(r)
assert sum(n for _d, n in ref) > 0 and sum(1 for _d, n in ref if n > 0) >= len(ref) * 2 // 3, [n for _d, n in ref]
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 400
free0 = None
t0 = time.time()
checked = 0
for it in range(steps):
    inflight = int(rng.integers(1, 5))
    tickets = []
    for _ in range(inflight):
        k = int(rng.integers(1, 13))
        idx = [int(v) for v in rng.choice(len(pages), k, replace=False)]
        t = s.submit_batch_raw([pages[i] for i in idx], [pages[i].shape[0] for i in idx], [pages[i].shape[1] for i in idx],
                               retto_amd.RT_MEM_HOST, [maps[i] for i in idx])
        tickets.append((idx, t))
    for j in rng.permutation(len(tickets)):
        idx, t = tickets[int(j)]
        r = s.wait_batch_raw(t)
        want = float(np.sum([ref_sum[i] for i in idx], dtype=np.float64)); got_sum = lib.rt_results_det_checksum(r)
        # (1e-6: the sums are fp64, but at levels whose batch total reaches the wide-GEMM sizes the squeeze-excite means come from
        #  the depthwise kernel's fused partial sums instead of k_pool_partial's -- another fp32 summation order, ~1e-8 on the map)
        assert abs(got_sum - want) <= 1e-6 * abs(want), "iteration %d: det map checksum %.17g of a batch of %d, its pages alone sum to %.17g" % (it, got_sum, len(idx), want)
        for pos, i in enumerate(idx):
            got = digest(r, pos)
            assert got == ref[i], "iteration %d: page %d at position %d of a batch of %d differs from its own run" % (it, i, pos, len(idx))
            checked += 1
        lib.rt_results_free(r)
    if it % 50 == 49:
        free, total = torch.cuda.mem_get_info()
        if it >= 199 and free0 is None:
            free0 = free
        print("iteration %4d  %6d pages checked  free %.2f GB  (%.1f s)" % (it + 1, checked, free / 2**30, time.time() - t0), flush=True)
free, _ = torch.cuda.mem_get_info()
assert free0 is None or abs(free - free0) < 512 * 2**20, "device memory drifted by %.1f MB" % ((free0 - free) / 2**20)
print("async soak ok: %d pages in %d iterations" % (checked, steps))
