"""Diagnostics for one fp16 conv / 1x1 GEMM launch through rt_debug_conv16:
    RT_CONV_STAMPS=1 python tools/conv16_stamps.py [n cin cout k h w]
prints the time of the launch (HIP events, 5th of 5 repetitions) and, when the library was built with `make STAMPS=1`
(retto_amd/csrc: in-kernel s_memtime stamps, a diagnostic build that de-pipelines the loops -- never ship or time it as the
product), the per-stage stamps of one workgroup.  A production build prints the launch time only."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import retto_amd
n, cin, cout, k, h, w = [int(v) for v in sys.argv[1:7]] if len(sys.argv) > 6 else (16, 128, 128, 3, 240, 240)
s = retto_amd.RettoSession(retto_amd.synthetic_session_config(0, dtype="f16"))
lib, hd = s.worker._hd.lib, s.worker._hd.h
rng = np.random.default_rng(0)
x = rng.standard_normal((n, cin, h, w)).astype(np.float32)
wt = (rng.standard_normal((cout, cin, k, k)) * 0.05).astype(np.float32)
out = np.empty((n, cout, h, w), np.float32)
lib.rt_debug_conv16.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
rc = lib.rt_debug_conv16(hd, x.ctypes.data, n, cin, h, w, wt.ctypes.data, cout, k, k, 1, 1, None, 1, out.ctypes.data)
print("rc", rc)
s.close()
