"""Race soak of the barrier-free thin-block kernels (GPU only): every shape of tools/bench_lc.py plus ragged ones, REPS
repetitions of 5 launches each, the last launch's output compared bit for bit with k_lc_thin / the unfused pair.
    python tools/soak_lc.py [reps]"""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import retto_amd
s = retto_amd.RettoSession(retto_amd.synthetic_session_config(0))
lib, h = s._hd.lib, s._hd.h
lib.rt_bench_lc.argtypes = [C.c_void_p] + [C.c_int] * 8 + [C.POINTER(C.c_float), C.POINTER(C.c_float)]
shapes = [(32, 480, 480, 16, 32, 1), (32, 480, 480, 32, 48, 2), (32, 240, 240, 48, 48, 1), (32, 240, 240, 48, 96, 2),
          (1024, 24, 200, 16, 32, 1), (1024, 24, 200, 32, 64, 1), (1024, 24, 200, 64, 64, 1), (1024, 24, 200, 64, 128, 21),
          (37, 13, 37, 64, 64, 1), (19, 5, 333, 48, 96, 2), (300, 24, 77, 64, 128, 21)]
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
bad = 0
for rep in range(reps):
    for (n, hh, ww, ci, co, st) in shapes:
        ms, md = C.c_float(), C.c_float(-1)
        rc = lib.rt_bench_lc(h, n, hh, ww, ci, co, st, 3, 5, C.byref(ms), C.byref(md))
        if rc != 0 or md.value != 0.0:
            bad += 1
            print("rep %d shape %s: rc %d max |diff| %g" % (rep, (n, hh, ww, ci, co, st), rc, md.value), flush=True)
print("%d repetitions x %d shapes x 5 launches: %d mismatches" % (reps, len(shapes), bad))
sys.exit(1 if bad else 0)
