"""Debug probe: a C3-shaped batch through rt_run_batch with the split-bf16 kernels on, against the same batch with them off."""
import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import retto_amd
from retto_amd import workload
from oracle import ref_lib as R
n = int(os.environ.get("PAGES", "32"))
sess = retto_amd.RettoSession(retto_amd.synthetic_session_config(0))
lib, h = sess._hd.lib, sess._hd.h
if os.environ.get("LANES"):
    lib.rt_set_lanes(h, int(os.environ["LANES"]))
pages, maps = [], []
for i in range(n):
    page, rects = workload.planted_page(960, 960, 32, seed=100 + i)
    dh, dw = R.resize_either_dims(960, 960)
    pages.append(page); maps.append(workload.planted_map(dh, dw, 960, 960, rects))
ref = sess.run_batch(pages, det_map_override=maps)
print("ref ok", sum(len(p.rec_result) for p in ref), flush=True)
lib.rt_debug_set_variants(0, 0, 4096)
for it in range(int(os.environ.get("ITERS", "3"))):
    got = sess.run_batch(pages, det_map_override=maps)
    same = sum(int(np.array_equal(a.tokens, b.tokens)) for p, q in zip(ref, got) for a, b in zip(p.rec_result, q.rec_result))
    tot = sum(len(p.rec_result) for p in ref)
    sc = max(abs(a.score - b.score) for p, q in zip(ref, got) for a, b in zip(p.rec_result, q.rec_result) if a.score == a.score)
    print("split run %d: %d / %d lines with equal tokens, max score diff %.2e" % (it, same, tot, sc), flush=True)
if os.environ.get("INFLIGHT"):
    import ctypes as C
    d_pages, d_maps = [], []
    for arr, lst in [(p_, d_pages) for p_ in pages] + [(m_, d_maps) for m_ in maps]:
        ptr = C.c_void_p()
        assert lib.rt_device_malloc(h, arr.nbytes, C.byref(ptr)) == 0
        assert lib.rt_memcpy_h2d(h, ptr, arr.ctypes.data, arr.nbytes) == 0
        lst.append(ptr.value)
    hs = [960] * n; ws = [960] * n
    q = []
    for it in range(int(os.environ.get("STEPS", "8"))):
        q.append(sess.submit_batch_raw(d_pages, hs, ws, retto_amd.RT_MEM_DEVICE, d_maps))
        if len(q) >= int(os.environ["INFLIGHT"]):
            lib.rt_results_free(sess.wait_batch_raw(q.pop(0)))
    while q:
        lib.rt_results_free(sess.wait_batch_raw(q.pop(0)))
    print("in-flight steps ok", flush=True)
lib.rt_debug_set_variants(0, 0, 0)
sess.close()
