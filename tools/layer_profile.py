"""Per-layer timing table of the C3 workload (GPU only): run with RT_PROFILE_DETAIL=1 so the
HIP-event profiler labels every LCNet block kernel "family@shape"; prints time per launch with
the algorithmic bytes and FLOPs of that launch.

    RT_PROFILE_DETAIL=1 python tools/layer_profile.py [pages] [steps] [lines]     (lines = 0: det network + DB post only)

Columns: ms per step, family, shape, launches per step, ms per launch, algorithmic TB/s and TFLOP/s of a launch (input read once +
output written once, fp32; 2 x MACs), and the fraction of the roofline that binds it: max(TB/s / 8.0, TFLOP/s / 157.3) with the
binding side named (MI355X_MICROARCH.md: HBM3E 8 TB/s spec -- about 6.3 achievable --, fp32 MFMA 157.3 TFLOP/s, fp16 MFMA
2500 TFLOP/s dense for the conv16 / gemm16 families).  WORKLOAD=c5 profiles the server models in fp16, DTYPE=f16 the mobile ones.
"""
import ctypes as C, os, sys
os.environ.setdefault("RT_PROFILE_DETAIL", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import retto_amd
from retto_amd import workload

pages_n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
lines_n = int(sys.argv[3]) if len(sys.argv) > 3 else 32
F16 = os.environ.get("WORKLOAD", "c3") == "c5" or os.environ.get("DTYPE") == "f16"   # WORKLOAD=c5: server models, fp16
cfg = retto_amd.synthetic_session_config(0, server=os.environ.get("WORKLOAD", "c3") == "c5", dtype="f16" if F16 else "f32")
cfg.lanes = 1
s = retto_amd.RettoSession(cfg)
lib, h = s._hd.lib, s._hd.h
if os.environ.get("VARIANTS"):
    lib.rt_debug_set_variants(*[int(v) for v in os.environ["VARIANTS"].split(",")])
d_pages, d_maps = [], []
for i in range(pages_n):
    page, rects = workload.planted_page(960, 960, lines_n, seed=i)
    m = workload.planted_map(960, 960, 960, 960, rects)
    for arr, lst in ((page, d_pages), (m, d_maps)):
        p = C.c_void_p()
        assert lib.rt_device_malloc(h, arr.nbytes, C.byref(p)) == 0
        assert lib.rt_memcpy_h2d(h, p, arr.ctypes.data, arr.nbytes) == 0
        lst.append(p.value)
hs = [960] * pages_n
for _ in range(2):
    lib.rt_results_free(s.run_batch_raw(d_pages, hs, hs, retto_amd.RT_MEM_DEVICE, d_maps))
# families whose ProfScope carries no shape are priced from the work model of the workload that just ran (retto_amd/workmodel.py):
# the family's algorithmic bytes / FLOPs per step over its launches per step
fam_work = {}
if not F16:
    from retto_amd import workmodel
    res1 = s.run_batch([workload.planted_page(960, 960, lines_n, seed=i)[0] for i in range(pages_n)],
                       det_map_override=[workload.planted_map(960, 960, 960, 960, workload.planted_page(960, 960, lines_n, seed=i)[1]) for i in range(pages_n)]) if lines_n else []
    crops, widths = workmodel.line_geometry(lib, res1) if lines_n else ([], [])
    fam_work = workmodel.step_work([(960, 960)] * pages_n, crops, widths)
s.profile_enable(True)
for _ in range(steps):
    lib.rt_results_free(s.run_batch_raw(d_pages, hs, hs, retto_amd.RT_MEM_DEVICE, d_maps))
rows = []
for name, (ms, calls) in s.profile_get().items():
    if calls == 0 or name.startswith("net/"):
        continue
    fam, _, shape = name.partition("@")
    per = ms / calls
    by = fl = 0.0
    if shape:
        a, b, c, d = (int(v) for v in shape.split(","))
        if fam.startswith("dwconv16"):
            k = int(fam[-1]); by = (a + b) * c * 2.0; fl = 2.0 * b * c * k * k
        elif fam.startswith("conv16") or fam.startswith("gemm16"):
            fl = 2.0 * a * b * c          # fp16 family: the MFMA side is what binds; bytes depend on the window and are left out
        elif fam.startswith("dwconv"):
            k = int(fam[-1]); by = (a + b) * c * 4.0; fl = 2.0 * b * c * k * k
        elif fam == "conv3x3_phase":      # a output pixels, b = depth per output (9 cf + 4 cc), 24 outputs; d = 1: the head conv
            fl = 2.0 * a * b * c
            by = (a * (72.0 if d else (b - 384) / 9.0 + 24) + a / 4.0 * (24.0 if d else 96.0)) * 4.0
        elif fam == "lc_thin":            # a OUTPUT pixels, b -> c channels, d = 10 sh + sw: the input has sh * sw times the pixels
            sh, sw = d // 10, d % 10
            by = (a * sh * sw * b + a * c) * 4.0
            fl = 2.0 * a * b * (9 + c)
        elif fam == "cls_block":          # a INPUT pixels, b = expanded channels, c = output channels, d = cin * 1000 + k * 100 + sh * 10 + se
            cin, k, sh, se = d // 1000, (d // 100) % 10, (d // 10) % 10, d % 10
            ao = a / sh
            by = (a * cin + ao * c) * 4.0 + (2.0 * ao * b * 4.0 if se else 0.0)   # (+ the depthwise scratch round trip of an SE block)
            fl = 2.0 * a * cin * b + 2.0 * ao * b * (k * k + c)
        else:
            by = a * (b + c) * 4.0 if not fam.startswith("conv3x3") else a * (96 + 24) * 4.0
            fl = 2.0 * a * b * c
    elif fam in fam_work:
        by = fam_work[fam]["bytes"] * steps / calls
        fl = fam_work[fam]["flops"] * steps / calls
    rows.append((ms / steps, fam, shape, calls // steps, per, by / per / 1e6 if by else 0, fl / per / 1e6 if fl else 0))
rows.sort(reverse=True)
tot = sum(r[0] for r in rows)
print("total profiled %.2f ms/step" % tot)
print("%8s %-34s %-28s %5s %9s %8s %8s %6s %s" % ("ms/step", "family", "shape", "n/stp", "ms/launch", "TB/s", "TFLOP/s", "frac", "bound"))
only = os.environ.get("ONLY")
for r in rows:
    if only and only not in r[1]:
        continue
    fh, fm = r[5] / 1e3 / 8.0, r[6] / 1e3 / (2500.0 if "16" in r[1] else 157.3)
    print("%8.3f %-34s %-28s %5d %9.3f %8.2f %8.1f %6.2f %s" % (r[0], r[1], r[2], r[3], r[4], r[5] / 1e3, r[6] / 1e3, max(fh, fm),
                                                              "" if not (r[5] or r[6]) else ("hbm" if fh >= fm else "mfma")))
