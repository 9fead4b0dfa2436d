#!/usr/bin/env python3
"""bench.py -- images/sec of the end-to-end PP-OCRv4 det+cls+rec path on MI355X.

Metric (BASELINE.json): images/sec end-to-end PP-OCRv4 det+rec @960x960; 1 -> 8 GPU scaling.
Workload at every N: BASELINE config C3 -- 32 synthetic 960x960 RGB pages per GPU per
step (weak scaling), 32 planted text lines per page, full pipeline
(resize/normalise -> DBNet -> DB post -> crops -> angle cls -> SVTR/CTC rec) through
libretto_hip's rt_run_batch with pages resident in HBM.  One process per GPU; weights are
broadcast once over RCCL (torch.distributed "nccl"); there is no per-step collective.

Prints ONE JSON line on rank 0 (see the driver contract) with `roofline` for the dominant
kernel family (HIP-event times measured live on the session's stream during the timed
region) and `cpu_baseline` (the CPU oracle timed on a bounded sample, rank 0, N=1 only).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
FP32_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: f32 MFMA / vector peak


def _flush_c_stdio():
    import ctypes
    try:
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--pages", type=int, default=32, help="pages per GPU per step (C3: 32)")
    ap.add_argument("--size", type=int, default=960)
    ap.add_argument("--lines", type=int, default=32, help="planted text lines per page")
    ap.add_argument("--workload", default="c3", choices=["c3", "c4"],
                    help="c3 (default, the metric's configuration): pages of --size x --size; c4: mixed page sizes "
                         "drawn (seeded) from SURVEY 8d's set, 2480x3508 scans included (resize_both shrinks them)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--det-sub-batch", type=int, default=0)
    ap.add_argument("--variants", type=str, default="", help="debug: gemm,dw,fuse kernel variants")
    ap.add_argument("--lanes", type=int, default=0, help="concurrent page streams inside rt_run_batch (0 = library default)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only for single-GPU smoke tests of the multi-rank path)")
    ap.add_argument("--share-gpu", action="store_true", help="test only: every rank uses device 0")
    ap.add_argument("--cpu-pages", type=int, default=3, help="pages of the same workload timed on the CPU oracle")
    ap.add_argument("--profile-all", action="store_true", help="print the per-family table to stderr")
    return ap.parse_args()


def main():
    a = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        if world == 1 and a.gpus > 1:
            print("bench.py: --gpus %d needs torch.distributed.run with %d ranks" % (a.gpus, a.gpus), file=sys.stderr)
            sys.exit(2)
    import torch
    import torch.distributed as dist
    dist_on = world > 1 or os.environ.get("RT_BENCH_FORCE_DIST") == "1"  # (test hook: run the RCCL path with a single rank)
    tdev = "cuda" if a.backend == "nccl" else "cpu"
    if dist_on:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if a.backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(a.backend, rank=rank, world_size=world)
    device = 0 if (a.share_gpu or not dist_on) else local_rank

    import retto_amd
    from retto_amd import synth, workload, workmodel

    # ---- weights: generated on rank 0, RCCL-broadcast once -------------------------------
    if rank == 0:
        blobs = list(synth.synth_models(0))
    else:
        blobs = [None] * 4
    if dist_on:
        from retto_amd.dist import broadcast_blobs
        blobs = broadcast_blobs(blobs, 4, rank, device=tdev)  # RCCL over xGMI, once
    det_b, cls_b, rec_b, dict_b = blobs
    cfg = retto_amd.RettoSessionConfig()
    cfg.det_sub_batch = a.det_sub_batch
    cfg.lanes = a.lanes
    cfg.worker_config = retto_amd.RettoHipWorkerConfig(device=device, models=retto_amd.RettoWorkerModelProvider(
        det=retto_amd.RettoWorkerModelSource.Blob(det_b), rec=retto_amd.RettoWorkerModelSource.Blob(rec_b),
        cls=retto_amd.RettoWorkerModelSource.Blob(cls_b)))
    cfg.rec_processor_config.character_source = retto_amd.RettoWorkerModelSource.Blob(dict_b)
    sess = retto_amd.RettoSession(cfg)
    lib, h = sess._hd.lib, sess._hd.h
    if a.variants:
        lib.rt_debug_set_variants(*[int(v) for v in a.variants.split(',')])

    # ---- synthetic pages + planted maps, staged to HBM once -------------------------------
    import ctypes as C
    S = a.size
    C4_SIZES = [(640, 640), (960, 960), (720, 1280), (1080, 1920), (1754, 1240), (3508, 2480)]
    rng_sizes = np.random.default_rng(77 + rank)
    pages, maps, d_pages, d_maps, det_dims = [], [], [], [], []
    for i in range(a.pages):
        ph, pw = (S, S) if a.workload == "c3" else C4_SIZES[int(rng_sizes.integers(0, len(C4_SIZES)))]
        page, rects = workload.planted_page(ph, pw, a.lines, seed=1000 * rank + i)
        rh, rw, dh, dw = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        assert lib.rt_resize_both_dims(h, ph, pw, C.byref(rh), C.byref(rw)) == 0      # a2: session size limits
        assert lib.rt_det_input_dims(h, rh.value, rw.value, C.byref(dh), C.byref(dw)) == 0  # a3: det input size
        m = workload.planted_map(dh.value, dw.value, ph, pw, rects)
        pages.append(page); maps.append(m); det_dims.append((dh.value, dw.value))
        for arr, lst in ((page, d_pages), (m, d_maps)):
            p = C.c_void_p()
            assert lib.rt_device_malloc(h, arr.nbytes, C.byref(p)) == 0
            assert lib.rt_memcpy_h2d(h, p, arr.ctypes.data, arr.nbytes) == 0
            lst.append(p.value)
    hs = [p.shape[0] for p in pages]; ws = [p.shape[1] for p in pages]

    def step():
        r = sess.run_batch_raw(d_pages, hs, ws, retto_amd.RT_MEM_DEVICE, d_maps)
        return r

    def barrier():
        lib.rt_synchronize(h)
        if dist_on:
            dist.barrier()
            if tdev == "cuda":
                torch.cuda.synchronize()

    # warmup (also sizes the arenas)
    n_lines = 0
    checksum = 0.0
    for _ in range(max(a.warmup, 1)):
        r = step()
        n_lines = sum(lib.rt_results_count(r, i) for i in range(a.pages))
        checksum = lib.rt_results_det_checksum(r)
        widths_probe = r
        lib.rt_results_free(r)
    # ---- timed region: EXACTLY K steps at the production setting (concurrent lanes) -------
    barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        r = step()
        lib.rt_results_free(r)
    barrier()
    t1 = time.perf_counter()
    elapsed = t1 - t0
    # ---- roofline pass (rank 0 only): the same K steps strictly serial on one stream with HIP
    # events around every launch.  Concurrent lanes share the GPU and stretch each other's
    # kernels, so a kernel's own duration can only be read from a serial pass.
    prof, serial_ms = {}, None
    if rank == 0:
        lib.rt_set_lanes(h, 1)
        for _ in range(2):  # lane 0's arenas re-size for the whole batch
            r = step(); lib.rt_results_free(r)
        sess.profile_enable(True)
        lib.rt_synchronize(h)
        ts = time.perf_counter()
        for _ in range(a.steps):
            r = step()
            lib.rt_results_free(r)
        lib.rt_synchronize(h)
        serial_ms = 1000.0 * (time.perf_counter() - ts) / a.steps
        prof = sess.profile_get()
        sess.profile_enable(False)
        lib.rt_set_lanes(h, 1 << 20)
    if dist_on:
        t = torch.tensor([elapsed], dtype=torch.float64, device=tdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        cnt = torch.tensor([n_lines], dtype=torch.int64, device=tdev)
        dist.all_reduce(cnt)
        n_lines_total = int(cnt.item())
    else:
        n_lines_total = n_lines

    # RCCL prints a version banner through C stdio at init; in a pipe it would only come out at process exit,
    # i.e. after (rank 0) or interleaved with (other ranks) the JSON line.  Push it out now, on every rank.
    _flush_c_stdio()
    if rank != 0:
        if dist_on:
            dist.destroy_process_group()
        _flush_c_stdio()
        return

    total_pages = world * a.pages * a.steps
    value = total_pages / elapsed
    ms_per_step = 1000.0 * elapsed / a.steps

    # ---- roofline of the dominant kernel family -------------------------------------------
    # rec line widths of this rank's pages, from the same planning rule as the pipeline
    from retto_amd import _lib as L
    res = sess.run_batch(pages[:a.pages], det_map_override=maps[:a.pages])
    widths = []
    for pr in res:
        dims = []
        for d in pr.det_result:
            b = d.boxes.as_array().reshape(1, 8).astype(np.float32)
            wv = np.zeros(1, np.int32); hv = np.zeros(1, np.int32)
            lib.rt_crop_dims(b.ctypes.data, 1, wv.ctypes.data, hv.ctypes.data)
            dims.append((int(hv[0]), int(wv[0])))
        order = sorted(range(len(dims)), key=lambda i: -(dims[i][0] / dims[i][1]))
        ratio = np.float32(320) / np.float32(48)
        for s0 in range(0, len(order), 6):
            idx = order[s0:s0 + 6]
            for i in idx:
                ratio = max(ratio, np.float32(dims[i][1]) / np.float32(dims[i][0]))
            widths += [lib.rt_resize_norm_width(48, 320, float(ratio))] * len(idx)
    work = workmodel.det_work(det_dims)
    for k, v in workmodel.rec_work(widths).items():
        if k in work:
            work[k]["bytes"] += v["bytes"]; work[k]["flops"] += v["flops"]
        else:
            work[k] = dict(v)
    nets = {name[4:]: ms / a.steps for name, (ms, calls) in prof.items() if calls and name.startswith("net/")}
    fams = sorted(((ms, calls, name) for name, (ms, calls) in prof.items() if calls and not name.startswith("net/")), reverse=True)
    total_ms = sum(f[0] for f in fams)
    if a.profile_all:
        for ms, calls, name in fams:
            wk = work.get(name, {"bytes": 0.0, "flops": 0.0})
            per_step_ms = ms / a.steps
            print("%-16s %9.3f ms/step %6d launches/step  %8.1f GB/s  %7.2f TFLOP/s" % (
                name, per_step_ms, calls // a.steps, wk["bytes"] / per_step_ms / 1e6 if per_step_ms else 0,
                wk["flops"] / per_step_ms / 1e9 if per_step_ms else 0), file=sys.stderr)
        print("sum of kernel families: %.2f ms/step; wall %.2f ms/step" % (total_ms / a.steps, ms_per_step), file=sys.stderr)
    roofline = None
    for ms, calls, name in fams:
        if name in work:
            wk = work[name]
            launches_per_step = calls / a.steps
            avg_ms = ms / calls
            bytes_per_launch = wk["bytes"] / launches_per_step
            flops_per_launch = wk["flops"] / launches_per_step
            gbs = bytes_per_launch / (avg_ms * 1e-3) / 1e9
            tfs = flops_per_launch / (avg_ms * 1e-3) / 1e12
            hbm_frac, mfma_frac = gbs / HBM_PEAK_GBS, tfs / FP32_PEAK_TFLOPS
            if hbm_frac >= mfma_frac:
                roofline = {"bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                            "frac": round(hbm_frac, 4), "traffic": None}
            else:
                roofline = {"bound": "mfma", "achieved": round(tfs, 2), "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s",
                            "frac": round(mfma_frac, 4), "traffic": None}
            # HBM bytes per launch from the committed PMC passes of this same command (profiles/pmc_traffic.json,
            # tools/pmc_summary.py: separate FETCH_SIZE / WRITE_SIZE passes, gfx950 corrections); only valid for
            # the default workload the passes were taken on
            pmc_symbol = {"gemm_pw/k_gemm_wide<4,5,4,3>": "k_gemm_wide<4, 5, 4, 3, 0, 0, 0, 0, 0>",
                          "gemm_pw/k_gemm_wide<2,5,4,3>+se": "k_gemm_wide<2, 5, 4, 3, 0, 0, 1, 0, 0>",
                          "gemm_pw/k_gemm_wide<2,4,4,2>": "k_gemm_wide<2, 4, 4, 2, 0, 0, 0, 0, 0>"}.get(name)
            pmc_path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "pmc_traffic.json")
            if pmc_symbol and (a.workload, a.pages, a.size, a.lines) == ("c3", 32, 960, 32) and os.path.exists(pmc_path):
                k = json.load(open(pmc_path))["kernels"].get(pmc_symbol)
                if k:
                    roofline["traffic"] = k["fetch_bytes_per_launch"] + k["write_bytes_per_launch"]
                    roofline["traffic_source"] = "profiles/pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this workload, FETCH_SIZE x2)"
                    if "sq" in k:  # SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8) and the clock of that pass
                        roofline["mfma_util_pmc"] = k["sq"]["mfma_util"]
                        roofline["clock_ghz_pmc"] = k["sq"]["clock_ghz"]
            roofline.update({"kernel": name, "avg_launch_ms": round(avg_ms, 4), "launches_per_step": launches_per_step,
                             "measured": "HIP events on the session stream, serial pass (lanes=1) of the same %d steps, %.2f ms/step" % (a.steps, serial_ms),
                             "share_of_kernel_time": round(ms / total_ms, 3),
                             "algorithmic_bytes_per_launch": int(bytes_per_launch),
                             "algorithmic_flops_per_launch": int(flops_per_launch)})
            break

    # ---- whole networks (HIP events around DetNet / ClsNet / RecNet::run in the same serial pass) -----------
    # north_star's "DBNet-backbone achieved HBM": B_layer = 500 MB per 960x960 page (every conv layer's
    # input read + output written once, fp32) over the det network's device time, against the 8 TB/s peak.
    networks = None
    if nets:
        det_flops = sum(v["flops"] for v in workmodel.det_work(det_dims).values())
        rec_flops = sum(v["flops"] for v in workmodel.rec_work(widths).values())
        b_layer = 500e6 * sum(dh_ * dw_ for dh_, dw_ in det_dims) / (960.0 * 960.0)
        networks = {"det_ms": round(nets.get("det", 0.0), 3), "cls_ms": round(nets.get("cls", 0.0), 3),
                    "rec_ms": round(nets.get("rec", 0.0), 3),
                    "det_tflops": round(det_flops / (nets["det"] * 1e-3) / 1e12, 2) if nets.get("det") else None,
                    "rec_tflops": round(rec_flops / (nets["rec"] * 1e-3) / 1e12, 2) if nets.get("rec") else None,
                    "det_b_layer_gbs": round(b_layer / (nets["det"] * 1e-3) / 1e9, 1) if nets.get("det") else None,
                    "det_b_layer_frac_of_hbm_peak": round(b_layer / (nets["det"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if nets.get("det") else None,
                    "note": "per step of %d pages, one lane; fp32 MFMA peak %.1f TFLOP/s, HBM peak %.0f GB/s" % (a.pages, FP32_PEAK_TFLOPS, HBM_PEAK_GBS)}

    # ---- CPU baseline: the oracle on a bounded sample of the same workload -------------------
    cpu_baseline = None
    if world == 1 and not a.no_cpu_baseline:
        import torch as _t
        from oracle.pipeline import OracleSession
        o = OracleSession(det_b, cls_b, rec_b, dict_b)
        k = max(1, min(a.cpu_pages, a.pages))
        t0c = time.perf_counter()
        for i in range(k):
            o.run(pages[i], det_map_override=maps[i])
        dt = time.perf_counter() - t0c
        cpu_baseline = {"value": round(k / dt, 4), "unit": "images/s", "cores": _t.get_num_threads(), "kind": "port",
                        "sample": "%d page(s) of the same %dx%d / %d-line workload through the CPU oracle "
                                  "(oracle/pipeline.py: torch-CPU fp32 nets + C++ pre/post restatement); "
                                  "reference ort-CPU itself is not runnable here" % (k, S, S, a.lines)}

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
    r_b = step(); in_batch = _page0(r_b); lib.rt_results_free(r_b)
    r_a = sess.run_batch_raw(d_pages[:1], hs[:1], ws[:1], retto_amd.RT_MEM_DEVICE, d_maps[:1]); alone = _page0(r_a); lib.rt_results_free(r_a)
    if not (np.array_equal(in_batch[0], alone[0]) and in_batch[2] == alone[2] and
            np.allclose(in_batch[1], alone[1], rtol=1e-4, atol=1e-6, equal_nan=True)):
        raise RuntimeError("bench self-check failed: page 0 differs between the batch and a run of its own")
    selfcheck = {"batch_invariance_page0": True, "lines": int(len(in_batch[2]))}

    out = {
        "metric": "images/sec end-to-end PP-OCRv4 det+rec @960x960",
        "value": round(value, 3), "unit": "images/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": ("C3: PP-OCRv4 mobile det+cls+rec full pipeline, batch=%d pages of %dx%d per GPU, " % (a.pages, S, S)
                                if a.workload == "c3" else
                                "C4: PP-OCRv4 mobile det+cls+rec full pipeline, batch=%d mixed-size pages per GPU (640x640 .. 2480x3508), " % a.pages) +
                               "%d planted lines/page (planted DB map drives box extraction; det net fully executed, "
                               "checksum %.6g)" % (a.lines, checksum),
                   "pages_per_gpu_per_step": a.pages, "lines_per_step_all_gpus": n_lines_total,
                   "weights": "seeded synthetic, PP-OCRv4 mobile shapes", "parallelism": "dp%d (pages sharded, no per-step collective)" % world},
        "roofline": roofline,
        "cpu_baseline": cpu_baseline,
        "networks": networks,
        "selfcheck": selfcheck,
    }
    sess.close()
    if dist_on:
        dist.destroy_process_group()
    _flush_c_stdio()
    sys.stdout.flush()
    print(json.dumps(out), flush=True)  # the ONE line of the contract, last thing on stdout


if __name__ == "__main__":
    main()
