"""Race soak for the LDS-DMA fp16 kernels: repeats a few many-tile launches and checks that every run is bit-identical to the first.
    python tools/soak_conv16.py [repetitions]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import retto_amd
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
s = retto_amd.RettoSession(retto_amd.synthetic_session_config(0, dtype="f16"))
lib, hd = s.worker._hd.lib, s.worker._hd.h
lib.rt_debug_conv16.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
shapes = [(16, 128, 128, 3, 120, 120), (16, 192, 192, 3, 60, 96), (64, 224, 224, 3, 3, 100), (8, 256, 64, 9, 60, 60), (1, 1664, 768, 1, 240, 320),
          (1, 480, 240, 1, 180, 333), (48, 160, 160, 3, 12, 100), (8, 80, 64, 3, 64, 64)]
bad = 0
for (n, cin, cout, k, h, w) in shapes:
    rng = np.random.default_rng(cin + cout + k)
    x = rng.standard_normal((n, cin, h, w)).astype(np.float16).astype(np.float32)
    wt = (rng.standard_normal((cout, cin, k, k)) * np.sqrt(2.0 / (cin * k * k))).astype(np.float16).astype(np.float32)
    b = (rng.standard_normal(cout) * 0.1).astype(np.float32)
    outs = []
    first = None
    for r in range(reps):
        out = np.empty((n, cout, h, w), np.float32)
        rc = lib.rt_debug_conv16(hd, x.ctypes.data, n, cin, h, w, wt.ctypes.data, cout, k, k, 1, 1, b.ctypes.data, 1, out.ctypes.data)
        assert rc == 0
        if first is None:
            first = out
        elif not np.array_equal(first, out):
            bad += 1
            print("DIFF shape", (n, cin, cout, k, h, w), "run", r, "values", int((first != out).sum()))
    print("shape", (n, cin, cout, k, h, w), "ok" if bad == 0 else "bad so far %d" % bad, "finite", bool(np.isfinite(first).all()))
s.close()
print("soak done: %d differing runs" % bad)
sys.exit(1 if bad else 0)
