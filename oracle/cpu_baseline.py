"""oracle/cpu_baseline.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

bench.py's `cpu_baseline` leg: the CPU restatement of the whole path -- C++ pre/post (oracle/retto_oracle.cpp) around
the C++ / OpenMP fp32 networks (oracle/nets_cpu.cpp) -- timed on the GPU box's host cores on a bounded sample of the
same workload.  Protocol: a fresh interpreter (no HIP runtime, no torch thread pools, passive
OpenMP waiting), P worker processes x T OpenMP threads covering every core of the host (count stated), one warm-up page
per worker, then >= 3 timed repetitions (median reported), a 1-thread figure, and a per-stage breakdown.  It stands in
for retto's ort-CPU path, which cannot be run (no Rust / ONNX Runtime / model files); torch-CPU (oracle/nets_torch.py,
oneDNN kernels) is the optional second column.  A reported, non-target baseline.

    python -m oracle.cpu_baseline --size 960 --lines 32 --pages 8 --budget 20     # prints one JSON object
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import multiprocessing as mp
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_HERE)
_lib = None
_lib_kind = None


def _load():
    """libretto_oracle_nets.so: rebuilt with -march=native for THIS host when g++ is here, else the shipped AVX2 build."""
    global _lib, _lib_kind
    if _lib is not None:
        return _lib
    src = os.path.join(_HERE, "nets_cpu.cpp")
    shipped = os.path.join(_HERE, "libretto_oracle_nets.so")
    path, kind = shipped, "x86-64-v3 build"
    try:
        out = os.path.join(tempfile.gettempdir(), "retto_oracle_nets_native_%d.so" % os.getuid())
        if not os.path.exists(out) or os.path.getmtime(out) < os.path.getmtime(src):
            tmp = "%s.%d.tmp" % (out, os.getpid())
            subprocess.check_call(["g++", "-O3", "-std=c++17", "-fPIC", "-fopenmp", "-march=native", "-fno-math-errno", "-shared",
                                   "-o", tmp, src], stderr=subprocess.DEVNULL)
            os.replace(tmp, out)
        path, kind = out, "-march=native build"
    except Exception:
        if not os.path.exists(shipped):
            subprocess.check_call(["make", "-C", _HERE, "-s", "libretto_oracle_nets.so"])
    lib = C.CDLL(path)
    lib.ocpu_create.restype = C.c_void_p
    lib.ocpu_create.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t]
    lib.ocpu_destroy.argtypes = [C.c_void_p]
    lib.ocpu_det.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
    lib.ocpu_cls.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    lib.ocpu_rec.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    _lib, _lib_kind = lib, kind
    return lib


class CpuNets:
    """The three worker functions (det / cls / rec on host NCHW tensors) on the C++ / OpenMP networks."""

    def __init__(self, det_blob: bytes, cls_blob: bytes, rec_blob: bytes, classes: int = 6625):
        self.lib = _load()
        self.h = self.lib.ocpu_create(det_blob, len(det_blob), cls_blob, len(cls_blob), rec_blob, len(rec_blob))
        if not self.h:
            raise RuntimeError("ocpu_create failed (bad RTWB blob)")
        self.classes = classes
        self.t = {"det": 0.0, "cls": 0.0, "rec": 0.0}

    def close(self):
        if self.h:
            self.lib.ocpu_destroy(self.h); self.h = None

    def set_threads(self, n):
        self.lib.ocpu_set_threads(int(n))

    def det(self, x):
        x = np.ascontiguousarray(x, np.float32); n, _, h, w = x.shape
        out = np.empty((n, 1, h, w), np.float32)
        t0 = time.perf_counter(); self.lib.ocpu_det(self.h, x.ctypes.data, n, h, w, out.ctypes.data); self.t["det"] += time.perf_counter() - t0
        return out

    def cls(self, x):
        x = np.ascontiguousarray(x, np.float32); n = x.shape[0]
        out = np.empty((n, 2), np.float32)
        t0 = time.perf_counter(); self.lib.ocpu_cls(self.h, x.ctypes.data, n, out.ctypes.data); self.t["cls"] += time.perf_counter() - t0
        return out

    def rec(self, x):
        x = np.ascontiguousarray(x, np.float32); n, _, _, w = x.shape
        T = self.lib.ocpu_rec(self.h, x.ctypes.data, n, w, None)
        out = np.empty((n, T, self.classes), np.float32)
        t0 = time.perf_counter(); self.lib.ocpu_rec(self.h, x.ctypes.data, n, w, out.ctypes.data); self.t["rec"] += time.perf_counter() - t0
        return out


def host_cpus():
    """CPUs this process may actually use: the affinity mask, capped by the cgroup CPU quota (the GPU box runs the bench in a
    pod whose quota -- e.g. 16 CPUs of 256 visible -- is what bounds any CPU baseline: more threads than that only throttle)."""
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        n = os.cpu_count() or 1
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]            # cgroup v2: "max 100000" or "<quota> <period>"
        if q != "max":
            quota = float(q) / float(per)
    except Exception:
        try:
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read()); per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / per
        except Exception:
            pass
    if quota:
        n = max(1, min(n, int(quota + 0.5)))
    return n


def _make_pages(size, lines, n, seed0=0):
    from retto_amd import workload
    from oracle import ref_lib as R
    out = []
    for i in range(n):
        page, rects = workload.planted_page(size, size, lines, seed=seed0 + i)
        dh, dw = R.resize_either_dims(*R.resize_both(page).shape[:2])
        out.append((page, workload.planted_map(dh, dw, size, size, rects)))
    return out


def _worker(rank, n_workers, threads, size, lines, n_pages, reps, barrier, q, torch_nets):
    """One worker process: its own networks, `threads` OpenMP threads, pages rank, rank + P, ... of every repetition."""
    try:
        from retto_amd import synth
        from oracle.pipeline import OracleSession
        det_b, cls_b, rec_b, dict_b = synth.synth_models(0)
        o = OracleSession(det_b, cls_b, rec_b, dict_b)
        nets = None
        if torch_nets:
            import torch
            torch.set_num_threads(threads)
        else:
            nets = CpuNets(det_b, cls_b, rec_b, o.wr["rec.head.fc.w"].shape[1])
            nets.set_threads(threads)
            o.det_worker, o.cls_worker, o.rec_worker = nets.det, nets.cls, nets.rec
        pages = _make_pages(size, lines, max(1, min(n_pages, 4)), seed0=100 * rank)
        o.run(pages[0][0], det_map_override=pages[0][1])      # warm-up (untimed)
        mine = len(range(rank, n_pages, n_workers))
        walls, t_tot = [], 0.0
        if nets:
            for k in nets.t:
                nets.t[k] = 0.0
        for _ in range(reps):
            barrier.wait()
            t0 = time.perf_counter()
            for i in range(mine):
                o.run(pages[i % len(pages)][0], det_map_override=pages[i % len(pages)][1])
            dt = time.perf_counter() - t0
            t_tot += dt
            barrier.wait()
            walls.append(time.perf_counter() - t0)   # includes waiting for the slowest worker: the repetition's wall time
        q.put({"rank": rank, "walls": walls, "pages": mine * reps, "busy": t_tot, "stages": dict(nets.t) if nets else None})
    except Exception as e:  # surface the failure instead of hanging the barrier
        q.put({"rank": rank, "error": repr(e)})
        try:
            barrier.abort()
        except Exception:
            pass


def _run_config(n_workers, threads, size, lines, n_pages, reps, torch_nets=False):
    ctx = mp.get_context("fork")
    barrier = ctx.Barrier(n_workers)
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, n_workers, threads, size, lines, n_pages, reps, barrier, q, torch_nets)) for r in range(n_workers)]
    for p in procs:
        p.start()
    res = [q.get() for _ in procs]
    for p in procs:
        p.join()
    errs = [r["error"] for r in res if "error" in r]
    if errs:
        raise RuntimeError("cpu baseline worker failed: " + errs[0])
    rates = [n_pages / max(r["walls"][k] for r in res) for k in range(reps)]
    pages = sum(r["pages"] for r in res)
    stages = None
    if res[0]["stages"] is not None:
        stages = {s: round(1000.0 * sum(r["stages"][s] for r in res) / pages, 2) for s in ("det", "cls", "rec")}
        stages["pre_post"] = round(1000.0 * (sum(r["busy"] for r in res) - sum(sum(r["stages"].values()) for r in res)) / pages, 2)
    return rates, stages


def measure(size=960, lines=32, budget_s=20.0, with_torch=False):
    """Runs inside the fresh interpreter.  Returns the cpu_baseline JSON object."""
    cpus = host_cpus()
    # calibrate on one process x 4 threads: seconds per page -> pages per repetition that fit the budget
    t0 = time.perf_counter()
    (r1,), _ = _run_config(1, min(4, cpus), size, lines, 1, 1)
    cal_wall = time.perf_counter() - t0
    sec_page_4t = 1.0 / r1
    T = 4 if cpus >= 8 else max(1, cpus // 2)
    P = max(1, cpus // T)
    reps = 3
    # every repetition gives each worker the same number of pages; bounded by the budget (setup + warm-up included)
    per_worker = max(1, min(8, int((budget_s * 0.55 / reps) / (sec_page_4t * 1.5))))
    n_pages = P * per_worker
    rates, stages = _run_config(P, T, size, lines, n_pages, reps)
    one_thread = None
    if sec_page_4t * 4 * 1.2 < max(10.0, budget_s * 0.6):
        (r, ), _ = _run_config(1, 1, size, lines, 1, 1)
        one_thread = r
    out = {"value": round(float(np.median(rates)), 4), "unit": "images/s", "cores": cpus, "kind": "port",
           "sample": "%d pages of %dx%d / %d planted lines per repetition through the CPU oracle (oracle/retto_oracle.cpp pre/post + "
                     "oracle/nets_cpu.cpp C++/OpenMP fp32 networks, %s) in a fresh interpreter: %d worker processes x %d OpenMP threads = "
                     "%d of the %d CPUs this process may use (affinity mask capped by the cgroup CPU quota; the host shows %d logical CPUs), 1 warm-up page per worker, %d timed repetitions (rates %s images/s, median "
                     "reported). Reference ort-CPU itself is not runnable here (no Rust / ONNX Runtime / model files)" % (
                         n_pages, size, size, lines, _lib_kind or "see oracle/Makefile", P, T, P * T, cpus, os.cpu_count() or 0, reps,
                         "/".join("%.2f" % r for r in rates)),
           "workers": P, "threads_per_worker": T, "nproc": os.cpu_count(),
           "one_thread_images_per_s": round(one_thread, 5) if one_thread else None,
           "one_worker_4_threads_images_per_s": round(r1, 4),
           "stage_cpu_ms_per_page": stages}
    if with_torch:
        tr, _ = _run_config(max(1, cpus // 16), min(16, cpus), size, lines, max(1, cpus // 16), 3, torch_nets=True)
        out["torch_cpu"] = {"value": round(float(np.median(tr)), 4), "unit": "images/s", "cores": cpus,
                            "sample": "same pages, oracle/nets_torch.py (torch-CPU fp32, oneDNN) + C++ pre/post: %d processes x %d torch threads, "
                                      "1 warm-up + 3 repetitions (median)" % (max(1, cpus // 16), min(16, cpus))}
    return out


def run_subprocess(size, lines, budget_s=20.0, with_torch=False, timeout_s=300.0):
    """Called by bench.py: runs `measure` in a fresh interpreter (no HIP runtime / torch pools in the timed process)."""
    env = dict(os.environ)
    env.update({"OMP_WAIT_POLICY": "PASSIVE", "OMP_PROC_BIND": "false", "PYTHONPATH": _ROOT + os.pathsep + env.get("PYTHONPATH", "")})
    env.pop("OMP_NUM_THREADS", None)
    cmd = [sys.executable, "-m", "oracle.cpu_baseline", "--size", str(size), "--lines", str(lines), "--budget", str(budget_s)]
    if with_torch:
        cmd.append("--torch")
    try:
        out = subprocess.run(cmd, cwd=_ROOT, env=env, capture_output=True, text=True, timeout=timeout_s)
        if out.returncode != 0:
            return {"value": None, "unit": "images/s", "cores": host_cpus(), "kind": "port", "sample": "cpu baseline failed: " + out.stderr[-400:]}
        return json.loads(out.stdout.strip().splitlines()[-1])
    except subprocess.TimeoutExpired:
        return {"value": None, "unit": "images/s", "cores": host_cpus(), "kind": "port", "sample": "cpu baseline exceeded %.0f s" % timeout_s}


def run_torch_server(det_b, cls_b, rec_b, dict_b, pages, maps, budget_s=20.0, describe=""):
    """C5 (server graphs): the only CPU restatement of the PPHGNet networks is oracle/nets_torch.py -- torch-CPU fp32
    (oneDNN), 1 warm-up page + timed pages inside the budget (at least one)."""
    import torch
    from oracle import nets_torch as N
    from oracle.pipeline import OracleSession
    cpus = host_cpus()
    torch.set_num_threads(cpus)   # the CPUs this process may use (cgroup quota), not the 128+ threads torch picks from the host's count
    o = OracleSession(det_b, cls_b, rec_b, dict_b)
    o.det_worker = lambda t: N.sdet_forward(o.wd, torch.from_numpy(t)).numpy()
    o.rec_worker = lambda t: N.srec_forward(o.wr, torch.from_numpy(t)).numpy()
    t0 = time.perf_counter(); o.run(pages[0], det_map_override=maps[0]); warm = time.perf_counter() - t0
    rates = []
    t_end = time.perf_counter() + max(0.0, budget_s - warm)
    while len(rates) < 1 or (len(rates) < 3 and time.perf_counter() + warm < t_end):
        i = len(rates) % len(pages)
        t1 = time.perf_counter(); o.run(pages[i], det_map_override=maps[i]); rates.append(1.0 / (time.perf_counter() - t1))
    return {"value": round(float(np.median(rates)), 4), "unit": "images/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": "%s through oracle/pipeline.py on torch-CPU fp32 (server graphs of oracle/nets_torch.py, oneDNN, %d threads = the CPUs "
                      "this process may use (affinity mask capped by the cgroup quota), one page at a time) + C++ pre/post: 1 warm-up page, "
                      "%d timed page(s), median" % (describe, torch.get_num_threads(), len(rates)),
            "nproc": os.cpu_count()}


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=960)
    ap.add_argument("--lines", type=int, default=32)
    ap.add_argument("--budget", type=float, default=20.0)
    ap.add_argument("--torch", action="store_true")
    a = ap.parse_args()
    sys.path.insert(0, _ROOT)
    _load()
    print(json.dumps(measure(a.size, a.lines, a.budget, a.torch)))
