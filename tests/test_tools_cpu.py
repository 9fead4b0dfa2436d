"""The measurement tooling parses rocprofv3 csv files by kernel name: a template argument added to a kernel silently drops it from a
family (round 5: `k_conv16v2<..., NW, BD>` vanished from `family:conv16_3x3` until the pattern was widened).  These tests feed the
tools small synthetic counter files with the kernel names of the current build."""
import csv
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _write(path, rows):
    with open(path, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Kernel_Name", "Counter_Name", "Counter_Value", "Start_Timestamp", "End_Timestamp"])
        for r in rows:
            w.writerow(r)


def test_pmc_summary_families_know_the_current_kernel_names(tmp_path):
    names = ["void rt::nh::k_conv16v2<2, 9, 3, 0, 4, true>(rt::nh::ConvArgs2)", "void rt::nh::k_conv16v2<4, 3, 3, 0, 4, true>(rt::nh::ConvArgs2)",
             "void rt::nh::k_conv16v2<2, 9, 9, 0, 8, true>(rt::nh::ConvArgs2)", "void rt::nh::k_gemm16p<4, 4>(rt::nh::GemmArgs16)",
             "void rt::nn::k_gemm32p<2, 1, true, 0, false>(rt::nn::GemmPArgs)", "void rt::nn::k_gemm32p<2, 1, false, 0, true>(rt::nn::GemmPArgs)"]
    f, w, s = tmp_path / "f.csv", tmp_path / "w.csv", tmp_path / "s.csv"
    _write(f, [(n, "FETCH_SIZE", 1000, 0, 100) for n in names for _ in range(2)])
    _write(w, [(n, "WRITE_SIZE", 500, 0, 100) for n in names for _ in range(2)])
    _write(s, [(n, c, v, 0, 1000) for n in names for c, v in (("GRBM_GUI_ACTIVE", 8000), ("SQ_VALU_MFMA_BUSY_CYCLES", 512000), ("SQ_WAVE_CYCLES", 1000))])
    out = tmp_path / "o.json"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pmc_summary.py"), str(f), str(w), str(out), str(s)], capture_output=True, text=True, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-1500:]
    k = json.load(open(out))["kernels"]
    for fam, members in (("family:conv16_3x3", 2), ("family:conv16_9x9", 1), ("family:gemm16", 1), ("family:gemm32p", 1), ("family:gemm32p_se", 1)):
        assert fam in k and len(k[fam]["members"]) == members, (fam, k.get(fam))
    assert k["family:conv16_3x3"]["fetch_bytes_per_launch"] == 1000 * 1024 * 2 and k["family:conv16_3x3"]["write_bytes_per_launch"] == 500 * 1024
    assert abs(k["family:conv16_3x3"]["sq"]["mfma_util"] - 0.5) < 1e-6      # 512000 / (1024 SIMDs x 8000 / 8)


def test_pmc_det_bytes_sums_the_det_launches_only(tmp_path):
    det = ["void rt::nn::k_stem_mfma<1>(rt::nn::StemArgs)", "void rt::nn::k_lc_lds<2, 3, 2, 2, 2, false, 2>(rt::nn::LcwArgs)", "void rt::nn::k_fpn_phase<6, 1, 1>(rt::nn::FpnArgs)"]
    other = ["rt::pp::k_ccl_rows(rt::pp::DbPage const*, float, int)", "rt::pp::k_contour_boxes(rt::pp::DbPage const*, rt::pp::DbParams)", "__amd_rocclr_copyBuffer", "rt::pp::k_sum_partial(float const*, long long, double*)"]
    f, w = tmp_path / "f.csv", tmp_path / "w.csv"
    passes = 3
    _write(f, [(n, "FETCH_SIZE", 100, 0, 1) for n in det + other for _ in range(passes)])
    _write(w, [(n, "WRITE_SIZE", 40, 0, 1) for n in det + other for _ in range(passes)])
    out = tmp_path / "d.json"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pmc_det_bytes.py"), str(f), str(w), str(out), "32"], capture_output=True, text=True, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-1500:]
    j = json.load(open(out))
    assert j["passes"] == passes
    assert j["det_fetch_bytes_per_pass"] == len(det) * 100 * 1024 * 2 and j["det_write_bytes_per_pass"] == len(det) * 40 * 1024
    assert set(j["left_out"]) >= {"k_ccl_rows", "k_contour_boxes", "k_sum_partial"}


def test_bench_workload_defaults(monkeypatch):
    """bench.py's per-workload defaults: the driver runs it with --gpus / --steps / --warmup only, so what a workload needs beyond
    those must come from parse() -- C3 two batches in flight, C2 (one page per call) four, an explicit --inflight wins."""
    sys.path.insert(0, ROOT) if ROOT not in sys.path else None
    import importlib
    bench = importlib.import_module("bench")

    def args(*argv):
        monkeypatch.setattr(sys, "argv", ["bench.py", *argv])
        return bench.parse()

    a = args("--gpus", "1", "--steps", "20", "--warmup", "5")
    assert (a.workload, a.pages, a.lines, a.inflight, a.dtype, a.models) == ("c3", 32, 32, 2, "f32", "mobile")
    a = args("--workload", "c2")
    assert (a.pages, a.lines, a.inflight, a.steps) == (1, 0, 4, 200)
    assert args("--workload", "c2", "--inflight", "1").inflight == 1
    a = args("--workload", "c5")
    assert (a.dtype, a.models, a.inflight) == ("f16", "server", 2)
