import sys, os
import functools; print = functools.partial(print, flush=True)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, retto_amd
from retto_amd import workload
from oracle import ref_lib as R
sess = retto_amd.RettoSession(retto_amd.synthetic_session_config(0))
good, gmaps = [], []
for i in range(4):
    page, rects = workload.planted_page(320, 480, 3, seed=60 + i)
    dh, dw = R.resize_either_dims(320, 480)
    good.append(page); gmaps.append(workload.planted_map(dh, dw, 320, 480, rects))
bad = list(good); bad[2] = np.zeros((1, 4000, 3), np.uint8)
def H(ps): return [p.shape[0] for p in ps]
def W(ps): return [p.shape[1] for p in ps]
for name, pages, maps in (("good", good, gmaps), ("bad", bad, None), ("good", good, gmaps), ("good", good, gmaps)):
    try:
        r = sess.run_batch_raw(pages, H(pages), W(pages), retto_amd.RT_MEM_HOST, maps); print("sync", name, "ok"); sess._hd.lib.rt_results_free(r)
    except Exception as e:
        print("sync", name, "ERR", type(e).__name__, e)
for order in ("bad,good", "good,bad,good"):
    ts = []
    for nm in order.split(","):
        pages, maps = (good, gmaps) if nm == "good" else (bad, None)
        ts.append((nm, sess.submit_batch_raw(pages, H(pages), W(pages), retto_amd.RT_MEM_HOST, maps)))
    for nm, t in ts:
        try:
            r = sess.wait_batch_raw(t); print("async", order, nm, "ok"); sess._hd.lib.rt_results_free(r)
        except Exception as e:
            print("async", order, nm, "ERR", type(e).__name__, e)
