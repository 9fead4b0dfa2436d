"""Diagnostics: fp32 GEMM time against K at fixed M, N (rt_bench_gemm): the intercept is the per-tile cost outside the K loop.
    python tools/gemm32_sweep.py [M] [N] [variant]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import retto_amd
M = int(sys.argv[1]) if len(sys.argv) > 1 else 1843200
N = int(sys.argv[2]) if len(sys.argv) > 2 else 96
variant = int(sys.argv[3]) if len(sys.argv) > 3 else 0
s = retto_amd.RettoSession(retto_amd.synthetic_session_config(0))
lib, h = s._hd.lib, s._hd.h
lib.rt_bench_gemm.argtypes = [C.c_void_p, C.c_longlong, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float)]
print("M %d N %d variant %d" % (M, N, variant))
for K in (16, 32, 64, 96, 128, 192, 256, 384, 512):
    ms, md = C.c_float(), C.c_float()
    rc = lib.rt_bench_gemm(h, M, K, N, variant, 5, C.byref(ms), C.byref(md))
    by = M * (K + N) * 4.0
    print("K %4d  %.3f ms  %.1f TFLOP/s  %.2f TB/s  rc %d maxdiff %.2g" % (K, ms.value, 2.0 * M * K * N / ms.value / 1e9, by / ms.value / 1e9, rc, md.value))
s.close()
