"""Kernel micro-benchmark: the fused thin LCNetV3 blocks (rt_bench_lc) on the det-network shapes of the C3 workload (GPU only).
    python tools/bench_lc.py [forms ...]        SHAPES=n,h,w,cin,cout,stride;... overrides the list"""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import retto_amd
s = retto_amd.RettoSession(retto_amd.synthetic_session_config(0))
lib, h = s._hd.lib, s._hd.h
lib.rt_bench_lc.argtypes = [C.c_void_p] + [C.c_int] * 8 + [C.POINTER(C.c_float), C.POINTER(C.c_float)]
shapes = [(32, 480, 480, 16, 32, 1), (32, 480, 480, 32, 48, 2), (32, 240, 240, 48, 48, 1), (32, 240, 240, 48, 96, 2),   # det backbone, 32 pages of 960 x 960
          (1024, 24, 200, 16, 32, 1), (1024, 24, 200, 32, 64, 1), (1024, 24, 200, 64, 64, 1), (1024, 24, 200, 64, 128, 21)]                              # rec backbone, 1024 lines of 48 x 400
if os.environ.get("SHAPES"):
    shapes = [tuple(int(x) for x in t.split(",")) for t in os.environ["SHAPES"].split(";")]
forms = [int(v) for v in sys.argv[1:]] or [0, 1]
for (n, hh, ww, ci, co, st) in shapes:
    s_h, s_w = (2, 1) if st == 21 else (st, st)
    px = n * ((hh + s_h - 1) // s_h) * ((ww + s_w - 1) // s_w)
    gb = (n * hh * ww * ci + px * co) * 4 / 1e9
    line = "%4d x %4d x %4d  %2d -> %3d /%-2d" % (n, hh, ww, ci, co, st)
    for f in forms:
        ms, md = C.c_float(), C.c_float()
        rc = lib.rt_bench_lc(h, n, hh, ww, ci, co, st, f, 5, C.byref(ms), C.byref(md))
        if rc != 0:
            line += " f%d: ERR %s" % (f, lib.rt_last_error(h)); continue
        line += " | f%d %6.3f ms %5.2f TB/s d=%.1e" % (f, ms.value, gb / ms.value, md.value)
    print(line, flush=True)
