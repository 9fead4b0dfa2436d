"""Race soak at C3 size (GPU): batches of 32 pages x 32 lines submitted two ahead from host memory (staged by rt_submit_batch),
three lanes, with the split-bf16 kernels on (RT_GEMM_SPLIT=1 is set here) or off (SOAK_FP32=1): every batch's results must be
bit-identical to the first run of that batch -- tile order varies with the dynamic queues, the values must not.

    python tools/soak_split.py [iterations]
"""
import ctypes as C, hashlib, os, sys, time
if not os.environ.get("SOAK_FP32"):
    os.environ["RT_GEMM_SPLIT"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import retto_amd
from retto_amd import workload

C5 = bool(os.environ.get("SOAK_C5"))   # server graphs, fp16 (the split switch does nothing there)
NP = int(os.environ.get("SOAK_PAGES", "32"))
s = retto_amd.RettoSession(retto_amd.synthetic_session_config(0, lanes=int(os.environ.get("SOAK_LANES", "0")), **({"server": True, "dtype": "f16"} if C5 else {})))
lib = s._hd.lib
batches = []
for b in range(3):
    pages, maps = [], []
    for i in range(NP):
        p, r = workload.planted_page(960, 960, 32, seed=1000 * b + i)
        pages.append(p); maps.append(workload.planted_map(960, 960, 960, 960, r))
    batches.append((pages, maps))


def detail(r, n_pages):
    out = []
    for i in range(n_pages):
        n = lib.rt_results_count(r, i)
        sc = np.ctypeslib.as_array(lib.rt_results_rec_scores(r, i), (n,)).copy() if n else np.zeros(0, np.float32)
        bx = np.ctypeslib.as_array(lib.rt_results_boxes(r, i), (n, 8)).copy() if n else np.zeros((0, 8), np.float32)
        toks = []
        for k in range(n):
            tp = C.POINTER(C.c_int32)()
            nt = lib.rt_results_rec_tokens(r, i, k, C.byref(tp))
            toks.append(tuple(np.ctypeslib.as_array(tp, (nt,)).tolist()) if nt else ())
        out.append((bx, sc, toks))
    return out


lib.rt_results_det_checksum.restype = C.c_double
lib.rt_results_det_checksum.argtypes = [C.c_void_p]


def digest(r, n_pages):
    h = hashlib.sha256()
    lines = 0
    # the det network's output only reaches the results through this sum (the boxes come from the planted maps): without it a
    # race in a det kernel would be invisible here
    h.update(np.float64(lib.rt_results_det_checksum(r)).tobytes())
    for i in range(n_pages):
        n = lib.rt_results_count(r, i)
        lines += n
        if n:
            h.update(np.ctypeslib.as_array(lib.rt_results_boxes(r, i), (n, 8)).tobytes())
            h.update(np.ctypeslib.as_array(lib.rt_results_cls_labels(r, i), (n,)).tobytes())
            h.update(np.ctypeslib.as_array(lib.rt_results_rec_scores(r, i), (n,)).tobytes())
            for k in range(n):
                tp = C.POINTER(C.c_int32)()
                nt = lib.rt_results_rec_tokens(r, i, k, C.byref(tp))
                h.update(np.ctypeslib.as_array(tp, (nt,)).tobytes() if nt else b"-")
    return h.hexdigest()[:16], lines


def submit(b):
    pages, maps = batches[b]
    return s.submit_batch_raw(pages, [960] * NP, [960] * NP, retto_amd.RT_MEM_HOST, maps)


steps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
ref = {}
refd = {}
q = []
t0 = time.time()
bad = 0
for it in range(steps + 2):
    if it < steps:
        nb = 1 if os.environ.get("SOAK_ONE") else 3
        q.append((it % nb, submit(it % nb)))
    if len(q) == 2 or it >= steps:
        if not q:
            break
        b, t = q.pop(0)
        r = s.wait_batch_raw(t)
        d = digest(r, NP)
        det = detail(r, NP) if (b not in ref or d != ref[b]) else None
        lib.rt_results_free(r)
        assert d[1] >= NP * 30, d
        if b not in ref:
            ref[b] = d; refd[b] = det
        elif d != ref[b]:
            bad += 1
            if bad <= 6:
                msg = []
                for pg, ((bx0, sc0, tk0), (bx1, sc1, tk1)) in enumerate(zip(refd[b], det)):
                    if len(sc0) != len(sc1) or not np.array_equal(bx0, bx1): msg.append("page %d: boxes differ" % pg); continue
                    for k in range(len(sc0)):
                        if sc0[k].tobytes() != sc1[k].tobytes() or tk0[k] != tk1[k]:
                            msg.append("page %d line %d (%d tokens, box w %.0f): score %.7f vs %.7f, tokens %s" % (pg, k, len(tk0[k]), bx0[k, 2] - bx0[k, 0], sc0[k], sc1[k], "equal" if tk0[k] == tk1[k] else "DIFFER"))
                print("DIFF batch %d at iteration %d: %d lines: %s" % (b, it, len(msg), "; ".join(msg[:8])), flush=True)
    if it % 20 == 19:
        print("iteration %4d  (%.1f s)  %s" % (it + 1, time.time() - t0, "ok" if not bad else "%d bad" % bad), flush=True)
s.close()
print("split soak done: %d iterations, %d differing batches, lines per batch %s" % (steps, bad, [v[1] for v in ref.values()]))
sys.exit(1 if bad else 0)
