#!/usr/bin/env python3
"""bench.py -- images/sec of the end-to-end PP-OCRv4 det+cls+rec path on MI355X.

Metric (BASELINE.json): images/sec end-to-end PP-OCRv4 det+rec @960x960; 1 -> 8 GPU scaling.
Default workload at every N: BASELINE config C3 -- 32 synthetic 960x960 RGB pages per GPU per step (weak
scaling), 32 planted text lines per page, full pipeline (resize/normalise -> DBNet -> DB post -> crops ->
angle cls -> SVTR/CTC rec) through libretto_hip's rt_run_batch, fp32, pages resident in HBM when the timed
region starts.  One process per GPU; weights are broadcast once over RCCL; there is no per-step collective.

Other workloads (never substituted for the C3 headline; `config.workload` names what ran):
  --workload c2   det only, one 960x960 page, batch 1 (launch-gap accounting in `c2`)
  --workload c4   mixed page sizes; with --global-batch B one list of B pages is sharded over the ranks
                  (retto_amd.dist.shard_pages), results are gathered in input order and checked for rank invariance
  --workload c5   PP-OCRv4 server det+rec in fp16 (BASELINE config 5), own roofline against the fp16 MFMA peak

Prints ONE JSON line on rank 0 (driver contract) with `roofline` for the dominant kernel family (HIP-event
times measured live on the session's stream) and `cpu_baseline` (rank 0, N=1 only).
"""
from __future__ import annotations

import argparse
import ctypes as C
import hashlib
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
FP32_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: f32 MFMA / vector peak
FP16_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: BF16/FP16 MFMA ~2.5 PF dense

C4_SIZES = [(640, 640), (960, 960), (720, 1280), (1080, 1920), (1754, 1240), (3508, 2480)]


def _flush_c_stdio():
    try:
        C.CDLL(None).fflush(None)
    except Exception:
        pass


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--pages", type=int, default=None, help="pages per GPU per step (C3: 32, C5: 128, C2: 1)")
    ap.add_argument("--size", type=int, default=960)
    ap.add_argument("--lines", type=int, default=None, help="planted text lines per page (default 32; C2: 0)")
    ap.add_argument("--workload", default="c3", choices=["c2", "c3", "c4", "c5"])
    ap.add_argument("--global-batch", type=int, default=0,
                    help="c4: one list of this many mixed-size pages sharded over the ranks (strong scaling); 0 = per-rank pages")
    ap.add_argument("--dtype", default=None, choices=["f32", "f16"], help="arithmetic of the networks (default f32; c5: f16)")
    ap.add_argument("--models", default=None, choices=["mobile", "server"], help="PP-OCRv4 graphs (default mobile; c5: server)")
    ap.add_argument("--pages-on", default="hbm", choices=["hbm", "host"],
                    help="where the pages are when the timed region starts (value is always quoted with hbm; the host rate is "
                         "reported beside it in `pages_on_host`)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-torch", action="store_true", help="also time the torch-CPU oracle (second column of cpu_baseline)")
    ap.add_argument("--det-sub-batch", type=int, default=0)
    ap.add_argument("--variants", type=str, default="", help="debug: gemm,dw,flags kernel variants")
    ap.add_argument("--lanes", type=int, default=0, help="concurrent page streams inside rt_run_batch (0 = library default)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only for CPU-side tests)")
    ap.add_argument("--share-gpu", action="store_true", help="test only: every rank uses device 0")
    ap.add_argument("--bcast", default="torch", choices=["torch", "cabi"],
                    help="weight broadcast: torch.distributed (default) or libretto_hip's own RCCL entry point rt_broadcast_blobs")
    ap.add_argument("--cpu-seconds", type=float, default=20.0, help="budget of the CPU baseline leg")
    ap.add_argument("--profile-all", action="store_true", help="print the per-family table to stderr")
    ap.add_argument("--dry-run", action="store_true", help="rank plumbing only (no GPU needed): launch / rendezvous / weight broadcast / gather, then one JSON line")
    ap.add_argument("--no-split-leg", action="store_true", help="skip the split-bf16 leg (opt-in kernels, reported as `split_bf16`)")
    ap.add_argument("--no-c5", action="store_true", help="skip the C5 leg (server graphs, fp16) that the default C3 run appends as `c5`")
    ap.add_argument("--c5-pages", type=int, default=32)
    ap.add_argument("--c5-steps", type=int, default=5)
    ap.add_argument("--inflight", type=int, default=None,
                    help="batches submitted ahead in the timed region (rt_submit_batch / rt_wait_batch; default 2, C2: 4); 1 = one synchronous rt_run_batch per step")
    ap.add_argument("--repeat", type=int, default=2, help="extra repetitions of the K timed steps after the timed region (spread, reported in `repeat`)")
    a = ap.parse_args()
    if a.workload == "c5":
        a.dtype = a.dtype or "f16"; a.models = a.models or "server"
        a.pages = a.pages or 128; a.steps = a.steps or 3; a.warmup = a.warmup if a.warmup is not None else 3   # (the arenas of the 3 lanes settle after 3 calls)
    if a.workload == "c2":
        a.pages = a.pages or 1; a.lines = 0 if a.lines is None else a.lines
        a.steps = a.steps or 200; a.warmup = a.warmup if a.warmup is not None else 20
        # (one-page calls are 78 short kernels: with 4 calls in flight their lanes fill the CUs a single call leaves idle --
        #  2240 pages/s at 2 in flight, 2960 at 3, 2980 at 4; C3's 32-page batches gain nothing beyond 2)
        a.inflight = 4 if a.inflight is None else a.inflight
    a.inflight = 2 if a.inflight is None else a.inflight
    a.dtype = a.dtype or "f32"; a.models = a.models or "mobile"
    a.pages = a.pages or 32; a.lines = 32 if a.lines is None else a.lines
    a.steps = a.steps or 40; a.warmup = 10 if a.warmup is None else a.warmup   # (round 4: a 1.1 s timed region instead of 0.6 s)
    return a


RC_RENDEZVOUS = 75   # a rank's exit code when the rendezvous port was taken (EX_TEMPFAIL): launch_ranks() retries once


def page_digest(lib, r, i):
    """Stable digest of one page's discrete results (boxes, labels, token ids): what must not depend on the rank / batch."""
    n = lib.rt_results_count(r, i)
    hsh = hashlib.sha256()
    if n:
        hsh.update(np.ctypeslib.as_array(lib.rt_results_boxes(r, i), (n, 8)).tobytes())
        hsh.update(np.ctypeslib.as_array(lib.rt_results_cls_labels(r, i), (n,)).tobytes())
        for k in range(n):
            tp = C.POINTER(C.c_int32)()
            nt = lib.rt_results_rec_tokens(r, i, k, C.byref(tp))
            hsh.update(bytes(np.ctypeslib.as_array(tp, (nt,)).tobytes()) if nt else b"-")
    return hsh.hexdigest()[:16], n


def launch_ranks(n):
    """`python bench.py --gpus N` without a launcher: N child interpreters with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set
    (what torch.distributed.run would export), rank 0's stdout relayed as this process's stdout (the ONE JSON line), the other
    ranks' stdout sent to stderr.  All ranks are polled together: the first rank that exits non-zero ends the job at once (its
    peers would otherwise sit in init_process_group / the broadcast until a timeout) and ITS exit code is returned; 0 if all
    ranks succeed.  RT_BENCH_RANK_TIMEOUT (default 900 s: several times a default run) is only the backstop for a job in which
    every rank hangs.  The rendezvous port is kept bound (SO_REUSEADDR) until the children are started, and a rendezvous that
    fails on a port somebody else took in between is retried once on a fresh port."""
    import socket
    deadline = time.time() + float(os.environ.get("RT_BENCH_RANK_TIMEOUT", "900"))

    def attempt():
        sk = socket.socket()
        sk.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
        sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]
        procs = []
        t_start = time.time()
        try:
            for r in range(n):
                env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                           MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
                procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                              stdout=None if r == 0 else sys.stderr))
        finally:
            sk.close()   # the interpreters need seconds to reach the rendezvous: the port stayed reserved while they were started
        rc = 0
        while True:
            codes = [pr.poll() for pr in procs]
            bad = [c for c in codes if c not in (None, 0)]
            if bad:
                rc = bad[0]
                break
            if all(c == 0 for c in codes):
                break
            if time.time() > deadline:
                rc = 124
                break
            time.sleep(0.05)
        if rc:
            for q in procs:
                if q.poll() is None:
                    q.terminate()
            t_kill = time.time() + 5.0
            for q in procs:
                try:
                    q.wait(timeout=max(0.1, t_kill - time.time()))
                except subprocess.TimeoutExpired:
                    q.kill(); q.wait()
        return rc, time.time() - t_start

    rc, took = attempt()
    if rc == RC_RENDEZVOUS and took < 60 and time.time() < deadline:   # the port was taken between close() and the store's bind
        print("bench.py: rendezvous port was taken, retrying once on a fresh port", file=sys.stderr)
        rc, _ = attempt()
    return rc


def c5_leg(a):
    """BASELINE config 5 on this GPU, driver-visible: a short `--workload c5` run of this script in a child process after the
    C3 measurement (32 pages per step so that it stays well inside the default run's minutes), its line condensed."""
    cmd = [sys.executable, os.path.abspath(__file__), "--workload", "c5", "--pages", str(a.c5_pages), "--steps", str(a.c5_steps),
           "--warmup", "3", "--no-cpu-baseline", "--no-c5"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE")}
    t0 = time.perf_counter()
    try:
        pr = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    except subprocess.TimeoutExpired:
        return {"error": "c5 leg timed out"}
    lines = [ln for ln in pr.stdout.splitlines() if ln.startswith("{")]
    if pr.returncode != 0 or not lines:
        return {"error": "c5 leg failed (rc %d): %s" % (pr.returncode, pr.stderr[-400:])}
    j = json.loads(lines[-1])
    return {"value": j["value"], "unit": j["unit"], "ms_per_step": j["ms_per_step"], "steps": j["steps"], "warmup": j["warmup"],
            "dtype": j["dtype"], "workload": j["config"]["workload"], "pages_per_step": j["config"]["pages_per_gpu_per_step"],
            "roofline": j["roofline"], "networks": j["networks"], "selfcheck": j["selfcheck"], "repeat": j.get("repeat"),
            "leg_wall_s": round(time.perf_counter() - t0, 1),
            "note": "same script, `--workload c5`, run as a child process after the C3 timed region; the headline fields above stay C3"}


def main():
    a = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        # not under torch.distributed.run: start the N ranks ourselves, one fresh child process per GPU.  Nothing in THIS
        # process has touched HIP or torch yet (children are started, never exec'ed over a GPU-initialised process).
        sys.exit(launch_ranks(a.gpus))
    if world != a.gpus:
        print("bench.py: --gpus %d but WORLD_SIZE=%d" % (a.gpus, world), file=sys.stderr)
        sys.exit(2)
    if os.environ.get("RT_BENCH_FAIL_RANK") == str(rank) and "WORLD_SIZE" in os.environ:   # test hook: a rank that dies before the rendezvous
        print("bench.py: rank %d fails on request (RT_BENCH_FAIL_RANK)" % rank, file=sys.stderr)
        sys.exit(3)
    import torch
    import torch.distributed as dist
    dist_on = world > 1 or os.environ.get("RT_BENCH_FORCE_DIST") == "1"  # (test hook: run the RCCL path with a single rank)
    tdev = "cuda" if a.backend == "nccl" else "cpu"
    if dist_on:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        try:
            if a.backend == "nccl":
                torch.cuda.set_device(local_rank)
                dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
            else:
                dist.init_process_group(a.backend, rank=rank, world_size=world)
        except Exception as e:   # noqa: BLE001 -- only to tell launch_ranks() that a retry on another port can help
            if rank == 0 and ("address already in use" in str(e).lower() or "eaddrinuse" in str(e).lower()):
                print("bench.py: rendezvous port %s is taken: %s" % (os.environ.get("MASTER_PORT"), e), file=sys.stderr)
                sys.exit(RC_RENDEZVOUS)
            raise
    device = 0 if (a.share_gpu or not dist_on) else local_rank

    import retto_amd
    from retto_amd import synth, workload, workmodel
    from retto_amd.dist import broadcast_blobs, shard_pages, process_shard, run_global_batch

    # ---- weights: generated on rank 0, RCCL-broadcast once -------------------------------
    if rank == 0:
        blobs = list(synth.synth_server_models(0) if a.models == "server" else synth.synth_models(0))
    else:
        blobs = [None] * 4
    t_bc = time.perf_counter()
    if dist_on and a.bcast == "cabi":   # the C-ABI hook a Rust / C++ host would use: RCCL id from rank 0, shared out of band
        from retto_amd.dist import broadcast_blobs_cabi, rccl_unique_id
        box = [rccl_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        blobs = broadcast_blobs_cabi(blobs, 4, rank, world, device, box[0])
    elif dist_on:
        blobs = broadcast_blobs(blobs, 4, rank, device=tdev)  # RCCL over xGMI, once
    bcast_ms = 1000.0 * (time.perf_counter() - t_bc) if dist_on else None
    rccl_ranks = dist.get_world_size() if dist_on else 1   # (the size of the communicator the weights went over)
    if a.dry_run:   # CPU-side check of the multi-rank plumbing (tests/test_dist_gloo.py): everything up to the session
        from retto_amd.dist import digest
        sizes = [C4_SIZES[(7 * i) % len(C4_SIZES)] for i in range(19)]
        res = run_global_batch(sizes, rank, world if dist_on else 1, lambda ids: ["%d:%s" % (i, digest(blobs)[:8]) for i in ids],
                               est_lines=[a.lines] * len(sizes), chunk=4)
        if dist_on:
            dist.barrier()
            dist.destroy_process_group()
        if rank == 0:
            print(json.dumps({"dry_run": True, "n_gpus": world, "rccl_ranks": rccl_ranks, "bcast_ms": None if bcast_ms is None else round(bcast_ms, 2),
                              "backend": a.backend, "blob_digest": digest(blobs)[:16], "gathered": res}), flush=True)
        return
    det_b, cls_b, rec_b, dict_b = blobs
    cfg = retto_amd.RettoSessionConfig()
    cfg.det_sub_batch = a.det_sub_batch
    cfg.lanes = a.lanes
    cfg.dtype = a.dtype
    cfg.worker_config = retto_amd.RettoHipWorkerConfig(device=device, models=retto_amd.RettoWorkerModelProvider(
        det=retto_amd.RettoWorkerModelSource.Blob(det_b), rec=retto_amd.RettoWorkerModelSource.Blob(rec_b),
        cls=retto_amd.RettoWorkerModelSource.Blob(cls_b)))
    cfg.rec_processor_config.character_source = retto_amd.RettoWorkerModelSource.Blob(dict_b)
    sess = retto_amd.RettoSession(cfg)
    lib, h = sess._hd.lib, sess._hd.h
    lib.rt_model_info.restype = C.c_char_p
    model_info = lib.rt_model_info(h).decode()
    if a.variants:
        lib.rt_debug_set_variants(*[int(v) for v in a.variants.split(',')])

    # ---- the page list of this rank ---------------------------------------------------------
    S = a.size
    global_mode = a.workload == "c4" and a.global_batch > 0
    if global_mode:
        # ONE list of B pages (same seeds on every rank), LPT-sharded by estimated work; each rank keeps its shard in input order
        rng_sizes = np.random.default_rng(77)
        all_sizes = [C4_SIZES[int(rng_sizes.integers(0, len(C4_SIZES)))] for _ in range(a.global_batch)]
        my_ids = sorted(shard_pages(all_sizes, world, rank, est_lines=[a.lines] * a.global_batch))
        sizes = [all_sizes[i] for i in my_ids]
        seeds = [5000 + i for i in my_ids]
    else:
        rng_sizes = np.random.default_rng(77 + rank)
        sizes = [(S, S) if a.workload != "c4" else C4_SIZES[int(rng_sizes.integers(0, len(C4_SIZES)))] for _ in range(a.pages)]
        seeds = [1000 * rank + i for i in range(a.pages)]
        my_ids = list(range(a.pages))
    n_my = len(sizes)

    def make_page(sz, seed):
        ph, pw = sz
        page, rects = workload.planted_page(ph, pw, a.lines, seed=seed)
        rh, rw, dh, dw = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        assert lib.rt_resize_both_dims(h, ph, pw, C.byref(rh), C.byref(rw)) == 0      # a2: session size limits
        assert lib.rt_det_input_dims(h, rh.value, rw.value, C.byref(dh), C.byref(dw)) == 0  # a3: det input size
        m = workload.planted_map(dh.value, dw.value, ph, pw, rects)
        return page, m, (dh.value, dw.value)

    pages, maps, d_pages, d_maps, det_dims = [], [], [], [], []
    for sz, seed in zip(sizes, seeds):
        page, m, dd = make_page(sz, seed)
        pages.append(page); maps.append(m); det_dims.append(dd)
        for arr, lst in ((page, d_pages), (m, d_maps)):
            p = C.c_void_p()
            assert lib.rt_device_malloc(h, arr.nbytes, C.byref(p)) == 0
            assert lib.rt_memcpy_h2d(h, p, arr.ctypes.data, arr.nbytes) == 0
            lst.append(p.value)
    hs = [p.shape[0] for p in pages]; ws = [p.shape[1] for p in pages]
    h_pages = [p.ctypes.data for p in pages]   # host-resident variant (pinned by the library's own staging)
    chunk = a.pages if global_mode else n_my   # a step of the global mode walks the shard in calls of --pages pages
    local = {gid: k for k, gid in enumerate(my_ids)}   # page id -> index into this rank's resident pages

    def run_ids(ids, on_host=False, want="free"):
        """One rt_run_batch over the pages `ids` (ids of this rank).  want: "free" -> [None] * n, "digest" -> [(digest, lines)],
        "raw" -> [(results handle, n)] (caller frees)."""
        ks = [local[i] for i in ids]
        if on_host:
            r = sess.run_batch_raw([h_pages[k] for k in ks], [hs[k] for k in ks], [ws[k] for k in ks], retto_amd.RT_MEM_HOST,
                                   [maps[k].ctypes.data for k in ks])
        else:
            r = sess.run_batch_raw([d_pages[k] for k in ks], [hs[k] for k in ks], [ws[k] for k in ks], retto_amd.RT_MEM_DEVICE,
                                   [d_maps[k] for k in ks])
        if want == "raw":
            return [(r, len(ks))] + [None] * (len(ks) - 1)
        out = [page_digest(lib, r, j) for j in range(len(ks))] if want == "digest" else [None] * len(ks)
        lib.rt_results_free(r)
        return out

    def step(on_host=False, keep=False):
        # (the per-rank half of retto_amd.dist.run_global_batch: the shard in calls of `chunk` pages, no collective)
        res = process_shard(my_ids, lambda ids: run_ids(ids, on_host, "raw" if keep else "free"), chunk)
        return [x for _i, x in res if x is not None] if keep else []

    def run_steps(k, on_host=False):
        """k steps.  --inflight N > 1 (default 2): batches are submitted ahead (rt_submit_batch / rt_wait_batch), at most N in
        flight -- every step still processes its whole batch; the results of step i are collected while step i + 1 runs.
        N = 1: one synchronous rt_run_batch per step (what rounds 1-3 timed)."""
        if a.inflight <= 1 or global_mode:
            for _ in range(k):
                step(on_host)
            return
        ks = list(range(n_my))
        q = []
        for _ in range(k):
            if on_host:
                q.append(sess.submit_batch_raw([h_pages[j] for j in ks], hs, ws, retto_amd.RT_MEM_HOST, [maps[j].ctypes.data for j in ks]))
            else:
                q.append(sess.submit_batch_raw(d_pages, hs, ws, retto_amd.RT_MEM_DEVICE, d_maps))
            if len(q) >= a.inflight:
                lib.rt_results_free(sess.wait_batch_raw(q.pop(0)))
        while q:
            lib.rt_results_free(sess.wait_batch_raw(q.pop(0)))

    def barrier():
        lib.rt_synchronize(h)
        if dist_on:
            dist.barrier()
            if tdev == "cuda":
                torch.cuda.synchronize()

    # warmup (also sizes the arenas)
    n_lines = 0
    checksum = 0.0
    # (the first warm-up step keeps its results: line count and det checksum of the workload; the others run the way the timed
    #  steps do -- batches submitted ahead -- so that the lanes are in their steady phase offsets when the clock starts)
    outs = step(keep=True)
    n_lines = sum(lib.rt_results_count(r, i) for r, n in outs for i in range(n))
    checksum = sum(lib.rt_results_det_checksum(r) for r, n in outs)
    for r, n in outs:
        lib.rt_results_free(r)
    if a.warmup > 1:
        run_steps(a.warmup - 1)
    # ---- timed region: EXACTLY K steps at the production setting (concurrent lanes) -------
    on_host = a.pages_on == "host"
    barrier()
    import resource

    def thread_cpu():
        out = {}
        try:
            tck = os.sysconf("SC_CLK_TCK")
            for tid in os.listdir("/proc/self/task"):
                f = open("/proc/self/task/%s/stat" % tid).read()
                name = f[f.index("(") + 1:f.rindex(")")]
                fld = f[f.rindex(")") + 2:].split()
                out[tid] = (name, (int(fld[11]) + int(fld[12])) / tck)
        except Exception:
            pass
        return out
    th0 = thread_cpu()
    ru0 = resource.getrusage(resource.RUSAGE_SELF)
    t0 = time.perf_counter()
    run_steps(a.steps, on_host)
    barrier()
    elapsed = time.perf_counter() - t0
    ru1 = resource.getrusage(resource.RUSAGE_SELF)
    th1 = thread_cpu()
    per_thread = sorted(((th1[t][1] - th0.get(t, (None, 0.0))[1], th1[t][0]) for t in th1), reverse=True)[:6]
    # host budget of one rank: CPU time (user + system, all threads) per step and the threads it keeps -- eight ranks share the
    # node's CPUs (the pod on the MI355X box: 16), so a rank that needs more than cpus / 8 of a core-second per second is the limit
    host_cpu_ms = 1000.0 * ((ru1.ru_utime - ru0.ru_utime) + (ru1.ru_stime - ru0.ru_stime)) / a.steps
    try:
        host_threads = int([ln for ln in open("/proc/self/status") if ln.startswith("Threads:")][0].split()[1])
    except Exception:
        host_threads = None
    # spread: the same K steps again, `--repeat` more times (each bracketed like the timed region); `value` stays the FIRST region
    rep_ms = [1000.0 * elapsed / a.steps]
    for _ in range(max(0, a.repeat)):
        barrier()
        tr = time.perf_counter()
        run_steps(a.steps, on_host)
        barrier()
        rep_ms.append(1000.0 * (time.perf_counter() - tr) / a.steps)
    # the same K steps as synchronous calls (one rt_run_batch per step, nothing submitted ahead): what rounds 1-3 reported
    sync_ms = None
    if a.inflight > 1 and not global_mode:
        barrier()
        tr = time.perf_counter()
        for _ in range(a.steps):
            step(on_host)
        barrier()
        sync_ms = 1000.0 * (time.perf_counter() - tr) / a.steps
    # the same steps with the pages starting in host memory (PCIe inside the timed region): reported beside `value`, never as it.
    # What crosses PCIe is what RettoSession::run would be handed (session.rs:108-131): the RGB pages.  The planted maps are
    # benchmark scaffolding and stay in HBM (RT_MEM_HOST_MAPS_DEVICE); the leg has its own warm-up and at least 20
    # steps whatever --steps says (the first host-fed calls size the staging; 5 steps after one cold call measured that).
    other_rate = None
    if not global_mode:
        ks_all = list(range(n_my))

        def host_steps(k):
            q = []
            for _ in range(k):
                q.append(sess.submit_batch_raw([h_pages[j] for j in ks_all], hs, ws, retto_amd.RT_MEM_HOST_MAPS_DEVICE, d_maps))
                if len(q) >= max(1, a.inflight):
                    lib.rt_results_free(sess.wait_batch_raw(q.pop(0)))
            while q:
                lib.rt_results_free(sess.wait_batch_raw(q.pop(0)))
        if not on_host:
            host_steps(3)
            lib.rt_synchronize(h)
            k2 = max(20, a.steps) if a.workload != "c2" else max(100, a.steps)
            t2 = time.perf_counter()
            host_steps(k2)
            lib.rt_synchronize(h)
            other_rate = n_my * k2 / (time.perf_counter() - t2)
        else:
            run_steps(2, False)
            lib.rt_synchronize(h)
            k2 = max(20, a.steps)
            t2 = time.perf_counter()
            run_steps(k2, False)
            lib.rt_synchronize(h)
            other_rate = n_my * k2 / (time.perf_counter() - t2)
    # ---- split-bf16 leg (round 6; opt-in kernels, NEVER `value`): the wide rec-net GEMMs that have no squeeze-excite operand on
    # k_gemm_split -- three bf16 planes per fp32 operand, six v_mfma_f32_16x16x32_bf16 products per fp32 product, fp32
    # accumulation (error against an fp64 product measured BELOW the fp32-MFMA kernel's: tests/test_gpu_parity.py
    # test_split_bf16_gemm_error_against_fp64, tools/bench_gemm_split.py).  Same steps, same batches in flight.
    split_leg = None
    if a.workload == "c3" and a.dtype == "f32" and not global_mode and not on_host and not a.no_split_leg:
        vv = [int(v) for v in a.variants.split(',')] if a.variants else [0, 0, 0]
        lib.rt_debug_set_variants(vv[0], vv[1], vv[2] | 4096)
        try:
            run_steps(3)
            barrier()
            ts_ = time.perf_counter()
            run_steps(a.steps)
            barrier()
            sp_ms = 1000.0 * (time.perf_counter() - ts_) / a.steps
            split_leg = {"value": round(n_my * 1000.0 / sp_ms * (world if dist_on else 1), 3), "unit": "images/s", "ms_per_step": round(sp_ms, 3),
                         "arithmetic": "fp32 operands split exactly into 3 bf16 terms each; 6 of the 9 bf16 x bf16 products (the dropped ones are below 2^-26 of "
                                       "the product) on v_mfma_f32_16x16x32_bf16 with fp32 accumulation",
                         "layers": "the 7 wide pointwise convs of the rec network (k_gemm32p's: 4 x 1230432x240x240, 2 x 307608x480x480, 1 x 1230432x128x240 "
                                   "at C3 size and the two squeeze-excite launches 615216x480x480 / x240x480, whose per-image scale multiplies the fp32 pixel "
                                   "operand before it is split)",
                         "note": "opt-in (RT_GEMM_SPLIT=1 / rt_debug_set_variants bit 12); `value`, `dtype` and `roofline` are the fp32-MFMA kernels"}
        finally:
            lib.rt_debug_set_variants(vv[0], vv[1], vv[2])
    # ---- global mode: gather every page's digest in INPUT order; rank invariance is checked on rank 0 below -------
    gathered = None
    if global_mode:   # every rank ends with every page's digest in input order (retto_amd.dist.run_global_batch)
        digs = run_global_batch(all_sizes, rank, world if dist_on else 1, lambda ids: run_ids(ids, False, "digest"),
                                est_lines=[a.lines] * a.global_batch, chunk=chunk)
        gathered = [(gid,) + tuple(d) for gid, d in enumerate(digs)]
    # ---- roofline pass (rank 0 only): the same K steps strictly serial on one stream with HIP
    # events around every launch.  Concurrent lanes share the GPU and stretch each other's
    # kernels, so a kernel's own duration can only be read from a serial pass.
    prof, prof_nets, serial_ms = {}, {}, None
    if rank == 0:
        lib.rt_set_lanes(h, 1)
        for _ in range(2):  # lane 0's arenas re-size for the whole batch
            step()
        sess.profile_enable(True)
        lib.rt_synchronize(h)
        ts = time.perf_counter()
        psteps = min(a.steps, 20)
        for _ in range(psteps):
            step()
        lib.rt_synchronize(h)
        serial_ms = 1000.0 * (time.perf_counter() - ts) / psteps
        prof = sess.profile_get()
        # whole-network device times from a pass WITHOUT the per-launch event pairs (~65 pairs per det pass are work on the
        # stream themselves): only the enclosing net/det, net/cls, net/rec scopes are recorded
        sess.profile_enable(2)
        for _ in range(psteps):
            step()
        lib.rt_synchronize(h)
        prof_nets = {name: v for name, v in sess.profile_get().items() if name.startswith("net/")}
        sess.profile_enable(False)
        # the same serial pass with the split-bf16 kernels on: launch times of the two families of the opt-in leg
        prof_split = {}
        if split_leg:
            vv = [int(v) for v in a.variants.split(',')] if a.variants else [0, 0, 0]
            lib.rt_debug_set_variants(vv[0], vv[1], vv[2] | 4096)
            try:
                step()
                sess.profile_enable(True)
                lib.rt_synchronize(h)
                for _ in range(psteps):
                    step()
                lib.rt_synchronize(h)
                prof_split = {name: v for name, v in sess.profile_get().items() if name.startswith("gemm_pw/") or name == "net/rec"}
                sess.profile_enable(False)
            finally:
                lib.rt_debug_set_variants(vv[0], vv[1], vv[2])
        lib.rt_set_lanes(h, 1 << 20)
    if dist_on:
        t = torch.tensor([elapsed], dtype=torch.float64, device=tdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        cnt = torch.tensor([n_lines, n_my], dtype=torch.int64, device=tdev)
        dist.all_reduce(cnt)
        n_lines_total, pages_total = int(cnt[0].item()), int(cnt[1].item())
    else:
        n_lines_total, pages_total = n_lines, n_my

    # RCCL prints a version banner through C stdio at init; in a pipe it would only come out at process exit,
    # i.e. after (rank 0) or interleaved with (other ranks) the JSON line.  Push it out now, on every rank.
    _flush_c_stdio()
    if rank != 0:
        if dist_on:
            dist.destroy_process_group()
        _flush_c_stdio()
        return

    value = pages_total * a.steps / elapsed
    ms_per_step = 1000.0 * elapsed / a.steps

    # ---- rank invariance of the global batch: pages that other ranks processed, re-run here one by one ---------------
    rank_invariance = None
    if global_mode:
        foreign = [g for g in gathered if g[0] not in set(my_ids)] or gathered
        sample = foreign[:: max(1, len(foreign) // 6)][:6]
        bad = []
        for gid, dig, nl in sample:
            page, m, _dd = make_page(all_sizes[gid], 5000 + gid)
            r = sess.run_batch_raw([page.ctypes.data], [page.shape[0]], [page.shape[1]], retto_amd.RT_MEM_HOST, [m.ctypes.data])
            d2 = page_digest(lib, r, 0)
            lib.rt_results_free(r)
            if d2 != (dig, nl):
                bad.append(gid)
        if bad:
            raise RuntimeError("bench rank-invariance check failed for pages %s" % bad)
        rank_invariance = {"pages_checked": [g[0] for g in sample], "ok": True,
                           "digest_of_all_pages": hashlib.sha256("".join(g[1] for g in gathered).encode()).hexdigest()[:16]}

    # ---- work model of what ran (rank 0's pages) ------------------------------------------------------------
    widths, crops = [], []
    if a.lines > 0:
        res = []
        for c0 in range(0, n_my, 32):
            res += sess.run_batch(pages[c0:c0 + 32], det_map_override=maps[c0:c0 + 32])
        crops, widths = workmodel.line_geometry(lib, res)
    if a.models == "server":
        det_w, rec_w = workmodel.sdet_work(det_dims), workmodel.srec_work(widths)
    elif a.dtype == "f16":
        det_w, rec_w = workmodel.det16_work(det_dims), workmodel.rec16_work(widths)
    else:
        det_w, rec_w = workmodel.det_work(det_dims), workmodel.rec_work(widths)
    work = {k: dict(v) for k, v in det_w.items()}
    extra = [rec_w]
    if a.models == "mobile" and a.dtype == "f32":   # the classifier and the u8 <-> f32 stages are priced for the fp32 path
        extra += [workmodel.cls_work(len(crops)),
                  workmodel.prepost_work([(p.shape[0], p.shape[1]) for p in pages], det_dims, crops, widths,
                                         sum(workmodel.tokens_for_width(w_) for w_ in widths))]
    for part in extra:
        for k, v in part.items():
            if k in work:
                work[k]["bytes"] += v["bytes"]; work[k]["flops"] += v["flops"]
            else:
                work[k] = dict(v)
    psteps = min(a.steps, 20)
    nets_ev = {name[4:]: ms / psteps for name, (ms, calls) in prof.items() if calls and name.startswith("net/")}   # with the per-launch events
    nets = {name[4:]: ms / psteps for name, (ms, calls) in prof_nets.items() if calls} or nets_ev
    fams = sorted(((ms, calls, name) for name, (ms, calls) in prof.items() if calls and not name.startswith("net/")), reverse=True)
    total_ms = sum(f[0] for f in fams)
    if a.profile_all:
        for ms, calls, name in fams:
            wk = work.get(name, {"bytes": 0.0, "flops": 0.0})
            per_step_ms = ms / psteps
            print("%-18s %9.3f ms/step %6d launches/step  %8.1f GB/s  %8.2f TFLOP/s" % (
                name, per_step_ms, calls // psteps, wk["bytes"] / per_step_ms / 1e6 if per_step_ms else 0,
                wk["flops"] / per_step_ms / 1e9 if per_step_ms else 0), file=sys.stderr)
        print("sum of kernel families: %.2f ms/step; serial wall %.2f ms/step; production wall %.2f ms/step" % (
            total_ms / psteps, serial_ms, ms_per_step), file=sys.stderr)
    if split_leg and rank == 0 and prof_split:
        # the pointwise-conv GEMM families of the rec / det networks in the two serial passes: fp32-MFMA kernels vs split-bf16 on
        def gemm_rows(pr):
            return {name: {"ms_per_step": round(v[0] / psteps, 3), "launches_per_step": v[1] / psteps, "avg_launch_ms": round(v[0] / v[1], 4)}
                    for name, v in sorted(pr.items()) if name.startswith("gemm_pw/") and v[1]}
        f32_rows, sp_rows = gemm_rows(prof), gemm_rows(prof_split)
        moved = [n for n in f32_rows if n not in sp_rows or abs(f32_rows[n]["launches_per_step"] - sp_rows[n]["launches_per_step"]) > 1e-9]
        wk_f = sum(work[n]["flops"] for n in moved if n in work); wk_b = sum(work[n]["bytes"] for n in moved if n in work)
        t_f32 = sum(f32_rows[n]["ms_per_step"] for n in moved) - sum(sp_rows[n]["ms_per_step"] for n in moved if n in sp_rows)
        t_sp = sum(v["ms_per_step"] for n, v in sp_rows.items() if "k_gemm_split" in n)
        split_leg["kernels"] = {"fp32_mfma_pass": f32_rows, "split_pass": sp_rows,
                                "layers_moved_ms_per_step": {"fp32_mfma": round(t_f32, 3), "split_bf16": round(t_sp, 3)},
                                "split_fp32_equivalent_tflops": round(wk_f / (t_sp * 1e-3) / 1e12, 1) if t_sp and wk_f else None,
                                "split_algorithmic_gbs": round(wk_b / (t_sp * 1e-3) / 1e9, 1) if t_sp and wk_b else None,
                                "measured": "HIP events, serial passes (lanes=1) of %d steps: default kernels | split kernels on" % psteps}
        if "net/rec" in prof_split and prof_split["net/rec"][1]:
            split_leg["rec_net_ms"] = round(prof_split["net/rec"][0] / psteps, 3)
    roofline = None
    for ms, calls, name in fams:
        if name in work:
            wk = work[name]
            launches_per_step = calls / psteps
            avg_ms = ms / calls
            bytes_per_launch = wk["bytes"] / launches_per_step
            flops_per_launch = wk["flops"] / launches_per_step
            gbs = bytes_per_launch / (avg_ms * 1e-3) / 1e9
            tfs = flops_per_launch / (avg_ms * 1e-3) / 1e12
            mfma_peak = FP16_PEAK_TFLOPS if "16" in name.split("/")[0] else FP32_PEAK_TFLOPS  # fp16 families carry "16" in their label
            hbm_frac, mfma_frac = gbs / HBM_PEAK_GBS, tfs / mfma_peak
            if hbm_frac >= mfma_frac:
                roofline = {"bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                            "frac": round(hbm_frac, 4), "traffic": None}
            else:
                roofline = {"bound": "mfma", "achieved": round(tfs, 2), "peak": mfma_peak, "unit": "TFLOP/s",
                            "frac": round(mfma_frac, 4), "traffic": None}
            # HBM bytes per launch come from separate rocprofv3 --pmc passes (tools/pmc_summary.py: FETCH_SIZE / WRITE_SIZE,
            # gfx950 corrections) committed as profiles/pmc_traffic_<workload>.json together with the workload flags and a digest
            # of the kernel sources they were taken on: used only when both match this run.
            pmc_path = os.path.join(ROOT, "profiles", "pmc_traffic_%s.json" % a.workload)
            if os.path.exists(pmc_path):
                from retto_amd import _lib as _L
                pj = json.load(open(pmc_path))
                same_workload = pj.get("workload") == {"workload": a.workload, "pages": a.pages, "size": a.size, "lines": a.lines,
                                                       "dtype": a.dtype, "models": a.models}
                k = pj.get("kernels", {}).get(pj.get("labels", {}).get(name, ""))
                if same_workload and k:
                    now = _L.source_digest()
                    stale = now != pj.get("csrc_digest")
                    # a profile of other sources is not evidence for this build: reported as stale, never copied into `traffic`
                    roofline["traffic"] = None if stale else k["fetch_bytes_per_launch"] + k["write_bytes_per_launch"]
                    roofline["traffic_source"] = {"file": "profiles/" + os.path.basename(pmc_path), "collected_on_csrc_digest": pj.get("csrc_digest"),
                                                  "this_build_csrc_digest": now, "stale": stale,
                                                  "how": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command (FETCH_SIZE x2 on gfx950), "
                                                         "tools/pmc_summary.py"}
                    if "sq" in k and not stale:  # SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8) and the clock of that pass
                        roofline["mfma_util_pmc"] = k["sq"]["mfma_util"]
                        roofline["clock_ghz_pmc"] = k["sq"]["clock_ghz"]
            roofline.update({"kernel": name, "avg_launch_ms": round(avg_ms, 4), "launches_per_step": launches_per_step,
                             "measured": "HIP events on the session stream, serial pass (lanes=1) of %d steps, %.2f ms/step" % (psteps, serial_ms),
                             "share_of_kernel_time": round(ms / total_ms, 3),
                             "algorithmic_bytes_per_launch": int(bytes_per_launch),
                             "algorithmic_flops_per_launch": int(flops_per_launch)})
            break

    # ---- the whole step against both roofs: every launch family's algorithmic FLOPs / bytes (retto_amd/workmodel.py) over the
    #      PRODUCTION step time (the timed region, all lanes, batches in flight)
    roofline_e2e = None
    if work:
        fl = sum(v["flops"] for v in work.values()); by = sum(v["bytes"] for v in work.values())
        mfma_peak_e2e = FP16_PEAK_TFLOPS if a.dtype == "f16" else FP32_PEAK_TFLOPS
        unpriced = sorted(name for _ms, _calls, name in fams if name not in work)
        roofline_e2e = {"flops_per_step": int(fl), "bytes_per_step": int(by), "ms_per_step": round(ms_per_step, 3),
                        "tflops": round(fl / (ms_per_step * 1e-3) / 1e12, 2), "frac_of_mfma_peak": round(fl / (ms_per_step * 1e-3) / 1e12 / mfma_peak_e2e, 4),
                        "gbs": round(by / (ms_per_step * 1e-3) / 1e9, 1), "frac_of_hbm_peak": round(by / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                        "families_without_a_price": unpriced,
                        "note": "sum over every launch family of the step as executed; the det FPN convs are priced by the upsampling-aware "
                                "algorithm that runs (7.6 GFLOP per 960x960 page, 10.4 for the reference graph's convs)"}

    # ---- whole networks (HIP events around the det / cls / rec forward in the same serial pass) -----------
    # north_star's "DBNet-backbone achieved HBM": B_layer = 500 MB per 960x960 page (every conv layer's
    # input read + output written once, fp32) over the det network's device time, against the 8 TB/s peak.
    networks = None
    if nets:
        det_flops = sum(v["flops"] for v in det_w.values())
        rec_flops = sum(v["flops"] for v in rec_w.values())
        det_bytes = sum(v["bytes"] for v in det_w.values())
        mfma_peak = FP16_PEAK_TFLOPS if a.dtype == "f16" else FP32_PEAK_TFLOPS
        networks = {"det_ms": round(nets.get("det", 0.0), 3), "cls_ms": round(nets.get("cls", 0.0), 3),
                    "rec_ms": round(nets.get("rec", 0.0), 3),
                    "det_tflops": round(det_flops / (nets["det"] * 1e-3) / 1e12, 2) if nets.get("det") else None,
                    "rec_tflops": round(rec_flops / (nets["rec"] * 1e-3) / 1e12, 2) if nets.get("rec") else None,
                    "det_layer_bytes_gbs": round(det_bytes / (nets["det"] * 1e-3) / 1e9, 1) if nets.get("det") else None,
                    "with_per_launch_events": {k: round(v, 3) for k, v in nets_ev.items()},
                    "note": "per step of %d pages, one lane, HIP events around each network in a pass that records nothing else "
                            "(`with_per_launch_events`: the same scopes in the per-launch profile pass, whose ~65 event pairs per det pass "
                            "add their own stream time); %s MFMA peak %.1f TFLOP/s, HBM peak %.0f GB/s; det_layer_bytes = per-launch "
                            "algorithmic bytes of the det net as executed (retto_amd/workmodel.py)" % (n_my, a.dtype, mfma_peak, HBM_PEAK_GBS)}
        if a.models == "mobile" and a.dtype == "f32" and nets.get("det"):
            b_layer = 500e6 * sum(dh_ * dw_ for dh_, dw_ in det_dims) / (960.0 * 960.0)
            networks["det_b_layer_gbs"] = round(b_layer / (nets["det"] * 1e-3) / 1e9, 1)
            networks["det_b_layer_frac_of_hbm_peak"] = round(b_layer / (nets["det"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
            # the same network by the COUNTERS: FETCH_SIZE (x2 on gfx950) + WRITE_SIZE summed over the det launches of one pass
            # (tools/pmc_det_bytes.py, committed as profiles/pmc_det_c3.json with the digest of the sources it was taken on)
            # over this run's det_ms -- what the HBM actually moved, next to the formula figure above
            dpath = os.path.join(ROOT, "profiles", "pmc_det_c3.json")
            if os.path.exists(dpath) and a.size == 960:
                from retto_amd import _lib as _L2
                dj = json.load(open(dpath))
                now = _L2.source_digest()
                stale = now != dj.get("csrc_digest")
                per_page = dj["det_hbm_bytes_per_pass"] / float(dj["workload"]["pages"])
                networks["det_pmc"] = {"file": "profiles/pmc_det_c3.json", "stale": stale, "collected_on_csrc_digest": dj.get("csrc_digest"),
                                       "hbm_bytes_per_page": int(per_page),
                                       "fetch_bytes_per_page": int(dj["det_fetch_bytes_per_pass"] / float(dj["workload"]["pages"])),
                                       "write_bytes_per_page": int(dj["det_write_bytes_per_pass"] / float(dj["workload"]["pages"]))}
                if not stale:
                    gbs_pmc = per_page * n_my / (nets["det"] * 1e-3) / 1e9
                    networks["det_pmc_gbs"] = round(gbs_pmc, 1)
                    networks["det_pmc_frac_of_hbm_peak"] = round(gbs_pmc / HBM_PEAK_GBS, 4)

    # ---- C2: launch-gap accounting of the batch-1 det path ----------------------------------------------------
    c2 = None
    if a.workload == "c2":
        launches = sum(calls for _ms, calls, _n in fams) / psteps
        c2 = {"det_net_ms": round(nets.get("det", 0.0), 4), "kernel_ms_per_step": round(total_ms / psteps, 4),
              "serial_wall_ms_per_step": round(serial_ms, 4), "production_wall_ms_per_step": round(ms_per_step, 4),
              "launches_per_step": round(launches, 1),
              "gap_ms_per_step": round(serial_ms - total_ms / psteps, 4),
              "note": "one 960x960 page, no text lines: det pre-process + DBNet + DB post; gap = serial wall - sum of kernel time "
                      "(host launch overhead, the count round trip, event bookkeeping)"}

    # ---- CPU baseline (rank 0, N = 1): the C++ / OpenMP restatement of the same pipeline on the host cores ----
    cpu_baseline = None
    if world == 1 and not a.no_cpu_baseline:
        from oracle import cpu_baseline as CB
        if a.models == "mobile":   # fresh interpreter: no HIP runtime or torch thread pools inside the timed CPU process
            cpu_baseline = CB.run_subprocess(S, a.lines, budget_s=a.cpu_seconds, with_torch=a.cpu_torch)
        else:
            cpu_baseline = CB.run_torch_server(det_b, cls_b, rec_b, dict_b, pages, maps, budget_s=a.cpu_seconds,
                                               describe="%dx%d / %d-line pages of this workload" % (S, S, a.lines))

    # ---- self-check outside the timed region (rank 0): a page's result must not depend on what else is in the batch.
    # Page 0 alone (small launches, other kernel shapes) against page 0 inside the batch -- boxes and token ids equal,
    # scores to fp32 rounding.  A mismatch is a bug in a size-dependent kernel path; the run fails instead of reporting.
    def _page0(r):
        n0 = lib.rt_results_count(r, 0)
        boxes = np.ctypeslib.as_array(lib.rt_results_boxes(r, 0), (n0, 8)).copy() if n0 else np.zeros((0, 8), np.float32)
        sc = np.ctypeslib.as_array(lib.rt_results_rec_scores(r, 0), (n0,)).copy() if n0 else np.zeros(0, np.float32)
        toks = []
        for k in range(n0):
            tp = C.POINTER(C.c_int32)()
            nt = lib.rt_results_rec_tokens(r, 0, k, C.byref(tp))
            toks.append([tp[t] for t in range(nt)])
        return boxes, sc, toks
    c1 = min(n_my, chunk)
    r_b = sess.run_batch_raw(d_pages[:c1], hs[:c1], ws[:c1], retto_amd.RT_MEM_DEVICE, d_maps[:c1]); in_batch = _page0(r_b); lib.rt_results_free(r_b)
    r_a = sess.run_batch_raw(d_pages[:1], hs[:1], ws[:1], retto_amd.RT_MEM_DEVICE, d_maps[:1]); alone = _page0(r_a); lib.rt_results_free(r_a)
    tol = dict(rtol=1e-4, atol=1e-6) if a.dtype == "f32" else dict(rtol=5e-2, atol=5e-3)
    if not (np.array_equal(in_batch[0], alone[0]) and in_batch[2] == alone[2] and np.allclose(in_batch[1], alone[1], equal_nan=True, **tol)):
        raise RuntimeError("bench self-check failed: page 0 differs between the batch and a run of its own")
    selfcheck = {"batch_invariance_page0": True, "lines": int(len(in_batch[2]))}

    names = {"c2": "C2: PP-OCRv4 %s det only (DBNet + DB post, no text lines), one %dx%d page per call (batch 1), " % (a.models, S, S),
             "c3": "C3: PP-OCRv4 %s det+cls+rec full pipeline, batch=%d pages of %dx%d per GPU, " % (a.models, n_my, S, S),
             "c4": ("C4: PP-OCRv4 %s det+cls+rec full pipeline, ONE global batch of %d mixed-size pages (640x640 .. 2480x3508) sharded over "
                    "%d rank(s) by estimated work (LPT), results gathered in input order, " % (a.models, a.global_batch, world)) if global_mode else
                   "C4: PP-OCRv4 %s det+cls+rec full pipeline, batch=%d mixed-size pages per GPU (640x640 .. 2480x3508), " % (a.models, n_my),
             "c5": "C5: PP-OCRv4 SERVER det (PPHGNet_small + LKPAN + PFHeadLocal) + mobile cls + SERVER rec (PPHGNet_small + SVTR/CTC), "
                   "fp16 MFMA, batch=%d pages of %dx%d per GPU, " % (n_my, S, S)}
    wl = names[a.workload] + ("%d planted lines/page (planted DB map drives box extraction; det net fully executed, checksum %.6g); "
                               "pages %s when the timed region starts" % (
                                   a.lines, checksum, "resident in HBM" if not on_host else "in host memory (H2D inside the timed region)"))
    out = {
        "metric": "images/sec end-to-end PP-OCRv4 det+rec @960x960",
        "value": round(value, 3), "unit": "images/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "strong" if global_mode else "weak",
        "vs_baseline": None, "dtype": a.dtype, "data": "synthetic",
        "config": {"workload": wl, "pages_per_gpu_per_step": n_my, "lines_per_step_all_gpus": n_lines_total,
                   "weights": "seeded synthetic, PP-OCRv4 %s shapes" % a.models, "networks": model_info,
                   "parallelism": "dp%d (pages sharded, no per-step collective)" % world,
                   "batches_in_flight": 1 if global_mode else max(1, a.inflight)},
        "roofline": roofline,
        "roofline_e2e": roofline_e2e,
        "cpu_baseline": cpu_baseline,
        "networks": networks,
        "selfcheck": selfcheck,
    }
    out["repeat"] = {"ms_per_step": [round(x, 3) for x in rep_ms], "min": round(min(rep_ms), 3), "median": round(float(np.median(rep_ms)), 3),
                     "note": "the timed region (first entry, = ms_per_step) and %d more regions of %d steps on this rank" % (len(rep_ms) - 1, a.steps)}
    out["host"] = {"cpu_ms_per_step": round(host_cpu_ms, 2), "cpu_cores_busy": round(host_cpu_ms / ms_per_step, 2), "threads": host_threads,
                   "cpu_budget": lib.rt_host_cpu_budget(),
                   "busiest_threads_cores": [[n_, round(dt / elapsed, 2)] for dt, n_ in per_thread if dt > 0],
                   "note": "rank 0, timed region: process CPU time (user + system, all threads) per step, the same as a fraction of one core "
                           "(cpu_ms / ms_per_step), threads of the process, and rt_host_cpu_budget() = CPUs this process may plan with "
                           "(affinity mask capped by the cgroup quota, / LOCAL_WORLD_SIZE)"}
    if sync_ms is not None:
        out["synchronous_calls"] = {"ms_per_step": round(sync_ms, 3), "value": round(n_my * world * 1000.0 / sync_ms, 3), "unit": "images/s",
                                    "note": "the same %d steps on this rank as one synchronous rt_run_batch per step (nothing submitted ahead: what rounds "
                                            "1-3 reported); `value` is timed with %d batches in flight through rt_submit_batch / rt_wait_batch -- every "
                                            "step still runs its whole batch, the host-side result assembly of step i overlaps step i + 1" % (a.steps, a.inflight)}
    if dist_on:
        out["rccl_ranks"] = rccl_ranks
        out["bcast_ms"] = round(bcast_ms, 2)
    if a.workload == "c3" and world == 1 and not a.no_c5 and a.dtype == "f32" and a.models == "mobile":
        sess.close(); sess = None
        out["c5"] = c5_leg(a)
    if other_rate is not None:
        out["pages_on_host" if not on_host else "pages_on_hbm"] = {
            "value": round(other_rate * (world if not global_mode else 1), 3), "unit": "images/s",
            "note": "rank 0, x%d ranks: %s" % (world, ("the pages start in host memory: 2.76 MB per 960^2 page cross PCIe inside the timed region (pageable memory; rt_submit_batch "
                    "stages them into HBM on a copy stream, one batch ahead of the lanes); own warm-up, >= 20 steps, %d batches in flight; the planted maps (benchmark scaffolding) stay in HBM" % max(1, a.inflight))
                    if not on_host else "the pages start in HBM")}
    if split_leg:
        out["split_bf16"] = split_leg
    if c2:
        out["c2"] = c2
    if rank_invariance:
        out["rank_invariance"] = rank_invariance
    if sess is not None:
        sess.close()
    if dist_on:
        dist.destroy_process_group()
    _flush_c_stdio()
    sys.stdout.flush()
    print(json.dumps(out), flush=True)  # the ONE line of the contract, last thing on stdout


if __name__ == "__main__":
    main()
