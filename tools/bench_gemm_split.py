"""Split-bf16 GEMM against the fp32-MFMA kernels: time per launch and error against an fp64 product (GPU only).

  python tools/bench_gemm_split.py            # the rec network's wide layers at C3 size
  SHAPES=1230432x240x240 python tools/bench_gemm_split.py 30 40

Variants: 1 = narrow fp32-MFMA kernel (k_gemm<NT>), 30 = k_gemm32p (fp32 MFMA, production), 40 = split-bf16 (k_gemm_split).
The error columns are max |err| and rms err of the bare product (bias 0, no activation) over 3 x 512 rows against
sum_k (double)a (double)w on the host, operands with full 24-bit significands (rt_bench_gemm_err).
"""
import ctypes as C, sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import retto_amd
s = retto_amd.RettoSession(retto_amd.synthetic_session_config(0))
lib, h = s._hd.lib, s._hd.h
lib.rt_bench_gemm.argtypes = [C.c_void_p, C.c_longlong, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float)]
lib.rt_bench_gemm_err.argtypes = [C.c_void_p, C.c_longlong, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint, C.POINTER(C.c_double)]
shapes = [(1230432, 240, 240), (307608, 480, 480), (1230432, 128, 240)]
if os.environ.get('SHAPES'):
    shapes = [tuple(int(x) for x in t.split('x')) for t in os.environ['SHAPES'].split(',')]
variants = [int(v) for v in sys.argv[1:]] or [30, 40]
iters = int(os.environ.get('ITERS', '10'))
rows = []
for (M, K, N) in shapes:
    for v in variants:
        ms, md = C.c_float(), C.c_float()
        rc = lib.rt_bench_gemm(h, M, K, N, v, iters, C.byref(ms), C.byref(md))
        if rc != 0:
            print("M=%d K=%d N=%d v%d: ERR %s" % (M, K, N, v, lib.rt_last_error(h))); continue
        errs = []
        for seed in (() if os.environ.get('NOERR') else (1, 2, 3)):
            o = (C.c_double * 4)()
            rc = lib.rt_bench_gemm_err(h, min(M, 65536 + 77), K, N, v, 512, 0, seed, o)
            if rc != 0:
                print("err run failed:", lib.rt_last_error(h)); break
            errs.append(list(o))
        tf = 2.0 * M * K * N / (ms.value * 1e-3) / 1e12
        row = {"M": M, "K": K, "N": N, "variant": v, "ms": round(ms.value, 4), "tflops_fp32_equiv": round(tf, 1),
               "maxdiff_vs_v1_with_epilogue": md.value,
               "max_abs_err_vs_fp64": max(e[0] for e in errs) if errs else None,
               "rms_err_vs_fp64": max(e[1] for e in errs) if errs else None,
               "rms_ref": errs[0][3] if errs else None}
        rows.append(row)
        print(json.dumps(row), flush=True)
