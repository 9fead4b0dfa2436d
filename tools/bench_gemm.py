"""Kernel micro-benchmark: nn::gemm variants on the hot shapes (GPU only)."""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import retto_amd
s = retto_amd.RettoSession(retto_amd.synthetic_session_config(0))
lib, h = s._hd.lib, s._hd.h
lib.rt_bench_gemm.argtypes = [C.c_void_p, C.c_longlong, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float)]
shapes = [(738432, 240, 240), (369216, 480, 480), (738432, 128, 240), (460800, 48, 48), (460800, 96, 96), (115200, 192, 192), (28800, 384, 384), (76000, 120, 6625)]
variants = [int(v) for v in sys.argv[1:]] or [1, 2, 3, 4]
if os.environ.get('SHAPES'):
    shapes = [tuple(int(x) for x in t.split('x')) for t in os.environ['SHAPES'].split(',')]
for (M, K, N) in shapes:
    line = "M=%7d K=%3d N=%4d " % (M, K, N)
    for v in variants:
        ms, md = C.c_float(), C.c_float()
        rc = lib.rt_bench_gemm(h, M, K, N, v, 5, C.byref(ms), C.byref(md))
        if rc != 0:
            line += " v%d: ERR %s" % (v, lib.rt_last_error(h)); continue
        tf = 2.0 * M * K * N / (ms.value * 1e-3) / 1e12
        line += " | v%d %7.3f ms %6.1f TF d=%.1e" % (v, ms.value, tf, md.value)
    print(line, flush=True)
