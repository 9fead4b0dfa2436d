"""GPU tests against the committed golden fixtures (tests/golden, produced by the CPU oracle)
and of the error behaviour of the boundary."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")


def test_golden_preprocess(hip_session):
    import retto_amd
    d = np.load(os.path.join(G, "preprocess.npz"))
    cfg = retto_amd.synthetic_session_config(0)
    cfg.det_processor_config.limit_side_len = 64
    s = retto_amd.RettoSession(cfg)
    try:
        assert np.array_equal(s.det_preprocess(d["img"]).view(np.uint32), d["det_input"].view(np.uint32))
    finally:
        s.close()


def test_golden_dbpost(hip_session):
    d = np.load(os.path.join(G, "dbpost.npz"))
    for k in ("rot", "nested"):
        m = d["map_" + k]
        b, s = hip_session.det_postprocess(m, *m.shape)
        assert np.array_equal(b, d["boxes_" + k])
        assert np.array_equal(s.view(np.uint32), d["scores_" + k].view(np.uint32))


def test_golden_crops(hip_session):
    c = np.load(os.path.join(G, "crops.npz"))
    crops = hip_session.crop_images(c["page"], c["boxes"])
    for i, crop in enumerate(crops):
        assert np.array_equal(crop, c["crop%d" % i])
        a = hip_session.resize_norm_image(crop, crop.shape[0], crop.shape[1], 48, 192, 0.0)
        b = hip_session.resize_norm_image(crop, crop.shape[0], crop.shape[1], 48, 320, 9.5)
        assert np.array_equal(a.view(np.uint32), c["cls%d" % i].view(np.uint32))
        assert np.array_equal(b.view(np.uint32), c["rec%d" % i].view(np.uint32))


def test_golden_ctc(hip_session):
    d = np.load(os.path.join(G, "ctc.npz"))
    ids, top = d["ids"], d["top"]
    n, t = ids.shape
    probs = np.full((n, t, 6625), 1e-5, np.float32)
    for i in range(n):
        for k in range(t):
            probs[i, k, ids[i, k]] = top[i, k]
    i_, k_, a_, b_ = d["tie"]
    probs[i_, k_, a_] = probs[i_, k_, b_]
    idx, pr, toks, sc = hip_session.ctc_decode(probs)
    assert np.array_equal(idx, d["idx"]) and np.array_equal(pr.view(np.uint32), d["prob"].view(np.uint32))
    for i in range(3):
        assert np.array_equal(toks[i], d["tok%d" % i])
    assert np.array_equal(sc.view(np.uint32), d["score"].view(np.uint32))


def test_errors_mirror_reference(models):
    """worker.rs:33-47: missing path / empty blob -> ModelNotFoundError; shape errors are ShapeError."""
    import retto_amd
    cfg = retto_amd.synthetic_session_config(0)
    cfg.worker_config.models.det = retto_amd.RettoWorkerModelSource.Path("/nonexistent/ch_PP-OCRv4_det_infer.rtwb")
    with pytest.raises(retto_amd.ModelNotFoundError):
        retto_amd.RettoSession(cfg)
    cfg = retto_amd.synthetic_session_config(0)
    cfg.worker_config.models.rec = retto_amd.RettoWorkerModelSource.Blob(b"")
    with pytest.raises(retto_amd.ModelNotFoundError):
        retto_amd.RettoSession(cfg)
    cfg = retto_amd.synthetic_session_config(0)
    cfg.rec_processor_config.character_source = retto_amd.RettoWorkerModelSource.Blob(b"\xff\xfe\n")
    with pytest.raises(retto_amd.Utf8Error):
        retto_amd.RettoSession(cfg)
    s = retto_amd.RettoSession(retto_amd.synthetic_session_config(0))
    try:
        with pytest.raises(retto_amd.ShapeError):
            s.worker.det(np.zeros((1, 3, 100, 100), np.float32))      # not a multiple of 32
        with pytest.raises(retto_amd.ShapeError):
            s.worker.cls(np.zeros((1, 3, 48, 100), np.float32))
        assert s.run_batch([]) == []
        r = s.run(np.zeros((64, 64, 3), np.uint8))                     # nothing detected -> empty results
        assert isinstance(r.det_result, list)
    finally:
        s.close()


def test_stage_json_shape(hip_session):
    import json
    from retto_amd import workload
    from oracle import ref_lib as R
    page, rects = workload.planted_page(160, 320, 2, seed=1)
    js = hip_session.stage_json(page)
    det, cls, rec = (json.loads(j) for j in js)
    assert isinstance(det, list) and isinstance(cls, list) and isinstance(rec, list)
    for d in det:
        assert set(d) == {"boxes", "score"} and len(d["boxes"]["inner"]) == 4 and set(d["boxes"]["inner"][0]) == {"x", "y"}
    for c in cls:
        assert set(c) == {"label"} and set(c["label"]) == {"label", "score"}
    for r in rec:
        assert set(r) == {"text", "score"}


def test_cli_directory_run(tmp_path):
    """retto-cli's loop (main.rs:72-93) over PNG files through the HIP session."""
    import json
    from PIL import Image
    from retto_amd import cli, workload
    for i in range(3):
        page, _ = workload.planted_page(160, 320, 2, seed=i)
        Image.fromarray(page).save(tmp_path / ("p%d.png" % i))
    out = tmp_path / "out.jsonl"
    assert cli.main(["--images", str(tmp_path), "--synthetic", "--batch", "2", "--json", str(out)]) == 0
    lines = [json.loads(l) for l in open(out)]
    assert len(lines) == 3 and all(set(l) == {"file", "det", "cls", "rec"} for l in lines)


def test_session_from_onnx_sources(tmp_path):
    """rt_create takes the .onnx files themselves (Path or Blob sources, worker.rs:18-27): the importer output
    drives the same kernels as the RTWB blob it was un-folded from."""
    import sys
    sys.path.insert(0, os.path.dirname(__file__))
    import retto_amd
    from retto_amd import synth, workload
    from onnx_writer import build_model_onnx
    blobs = synth.synth_models(0)
    tens = [synth.det_tensors(), synth.cls_tensors(), synth.rec_tensors()]
    # same seeds as synth_models: the blobs and the tensor dicts describe the same weights
    assert all(np.array_equal(synth.unpack_blob(b)[k], t[k]) for b, t in zip(blobs[:3], tens) for k in list(t)[:3])
    kinds = [retto_amd.MODEL_DET, retto_amd.MODEL_CLS, retto_amd.MODEL_REC]
    onnx = [build_model_onnx(retto_amd.model_manifest(k), t, seed=9, style=i % 2) for i, (k, t) in enumerate(zip(kinds, tens))]
    (tmp_path / "det.onnx").write_bytes(onnx[0])
    S = retto_amd.RettoWorkerModelSource
    cfg = retto_amd.synthetic_session_config(0)
    cfg.worker_config.models = retto_amd.RettoWorkerModelProvider(det=S.Path(str(tmp_path / "det.onnx")), rec=S.Blob(onnx[2]), cls=S.Blob(onnx[1]))
    a = retto_amd.RettoSession(cfg)
    b = retto_amd.RettoSession(retto_amd.synthetic_session_config(0))
    try:
        page, rects = workload.planted_page(160, 320, 3, seed=4)
        x = b.det_preprocess(page)
        wa, wb = retto_amd.RettoHipWorker(cfg, a._hd), retto_amd.RettoHipWorker(cfg, b._hd)
        assert np.abs(wa.det(x) - wb.det(x)).max() < 1e-4
        m = workload.planted_map(x.shape[2], x.shape[3], 160, 320, rects)
        ra, rb = a.run_batch([page], det_map_override=[m])[0], b.run_batch([page], det_map_override=[m])[0]
        assert [t.text for t in ra.rec_result] == [t.text for t in rb.rec_result]
        assert [c.label.label for c in ra.cls_result] == [c.label.label for c in rb.cls_result]
    finally:
        a.close(); b.close()


def test_native_directory_driver(tmp_path):
    """examples/retto_dir.cpp: the retto-cli loop in C++ over the C ABI alone (model files by path -- RTWB and
    .onnx --, PPM / PNG / JPEG files from a directory, decoded by the library); its JSON lines equal the Python mirror's stage JSON."""
    import json
    import subprocess
    import sys
    sys.path.insert(0, os.path.dirname(__file__))
    import retto_amd
    from retto_amd import synth, workload
    from onnx_writer import build_model_onnx
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "examples", "retto_dir")
    if not os.path.exists(exe):  # normally built by __graft_entry__.build(); host-only C++, seconds
        subprocess.check_call(["g++", "-std=c++17", "-O2", "-I" + os.path.join(root, "include"), os.path.join(root, "examples", "retto_dir.cpp"),
                               "-L" + os.path.join(root, "retto_amd"), "-lretto_hip", "-Wl,-rpath,$ORIGIN/../retto_amd", "-pthread", "-o", exe])
    det, cls, rec, dic = synth.synth_models(0)
    (tmp_path / "det.rtwb").write_bytes(det)
    (tmp_path / "cls.onnx").write_bytes(build_model_onnx(retto_amd.model_manifest(retto_amd.MODEL_CLS), synth.cls_tensors(), seed=2, style=1))
    (tmp_path / "rec.rtwb").write_bytes(rec)
    (tmp_path / "keys.txt").write_bytes(dic)
    pages_dir = tmp_path / "pages"
    pages_dir.mkdir()
    pages = []
    for i in range(3):
        page, _ = workload.planted_page(96 + 32 * i, 320, 2, seed=20 + i)
        # bright text lines on black give the random-weight detector nothing; the planted rectangles do not matter here
        if i == 0:
            (pages_dir / "p0.ppm").write_bytes(b"P6\n%d %d\n255\n" % (page.shape[1], page.shape[0]) + page.tobytes())
        elif i == 1:
            from PIL import Image
            Image.fromarray(page).save(pages_dir / "p1.png")
        else:  # lossy: the page the driver sees is the decoded one
            from PIL import Image
            Image.fromarray(page).save(pages_dir / "p2.jpg", quality=90)
            page = retto_amd.decode_image((pages_dir / "p2.jpg").read_bytes())
        pages.append(page)
    out = subprocess.run([exe, "--det", str(tmp_path / "det.rtwb"), "--cls", str(tmp_path / "cls.onnx"), "--rec", str(tmp_path / "rec.rtwb"),
                          "--keys", str(tmp_path / "keys.txt"), "--images", str(pages_dir), "--batch", "2"],
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    assert [int(l.split(" ", 1)[0]) for l in out.stdout.strip().split("\n")] == [0, 1, 2]   # index in the sorted file list
    lines = [json.loads(l.split(" ", 1)[1]) for l in out.stdout.strip().split("\n")]
    assert [os.path.basename(l["file"]) for l in lines] == ["p0.ppm", "p1.png", "p2.jpg"]
    assert "Successfully processed 3 images" in out.stderr
    # --ranks 1: the sharded form (model blobs broadcast over RCCL through rt_broadcast_blobs, files dealt by size) with the one
    # GPU of this box -- same lines
    out1 = subprocess.run([exe, "--det", str(tmp_path / "det.rtwb"), "--cls", str(tmp_path / "cls.onnx"), "--rec", str(tmp_path / "rec.rtwb"),
                           "--keys", str(tmp_path / "keys.txt"), "--images", str(pages_dir), "--batch", "2", "--ranks", "1"],
                          capture_output=True, text=True, timeout=300, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert out1.returncode == 0, out1.stderr
    import re   # (librccl prints a version banner on stdout at communicator creation: result lines are "<index> {json}")
    res1 = [l for l in out1.stdout.strip().split("\n") if re.match(r"^\d+ \{", l)]
    assert sorted(res1) == sorted(out.stdout.strip().split("\n"))
    S = retto_amd.RettoWorkerModelSource
    cfg = retto_amd.RettoSessionConfig()
    cfg.worker_config = retto_amd.RettoHipWorkerConfig(device=0, models=retto_amd.RettoWorkerModelProvider(
        det=S.Path(str(tmp_path / "det.rtwb")), rec=S.Path(str(tmp_path / "rec.rtwb")), cls=S.Path(str(tmp_path / "cls.onnx"))))
    cfg.rec_processor_config.character_source = S.Path(str(tmp_path / "keys.txt"))
    s = retto_amd.RettoSession(cfg)
    try:
        for page, l in zip(pages, lines):
            det_j, cls_j, rec_j = (json.loads(j) for j in s.stage_json(page))
            assert l["det"] == det_j and l["cls"] == cls_j and l["rec"] == rec_j
    finally:
        s.close()


def test_run_encoded_batch_equals_decoded_pages(hip_session):
    """RettoSession::run takes encoded bytes (session.rs:108): rt_run_encoded_batch over PNG / JPEG files gives the
    results of the decoded pages; a corrupt file fails the call with ImageError like image::load_from_memory's `?`."""
    import io
    from PIL import Image
    import retto_amd
    from retto_amd import workload
    files, pages = [], []
    for i, (fmt, kw) in enumerate((("PNG", {}), ("JPEG", {"quality": 92}), ("JPEG", {"quality": 80, "subsampling": 0}), ("BMP", {}))):
        page, _ = workload.planted_page(128 + 32 * i, 352, 3, seed=40 + i)
        b = io.BytesIO(); Image.fromarray(page).save(b, fmt, **kw)
        files.append(b.getvalue())
        pages.append(np.asarray(Image.open(io.BytesIO(files[-1])).convert("RGB")))
    got = hip_session.run_encoded_batch(files)
    want = hip_session.run_batch(pages)
    assert len(got) == len(want) == 4
    for g, w in zip(got, want):
        assert len(g.det_result) == len(w.det_result)
        for a, b_ in zip(g.det_result, w.det_result):
            assert np.array_equal(a.boxes.as_array(), b_.boxes.as_array()) and a.score == b_.score
        assert [c.label.label for c in g.cls_result] == [c.label.label for c in w.cls_result]
        assert [r.text for r in g.rec_result] == [r.text for r in w.rec_result]
    one = hip_session.run(files[0])
    assert [r.text for r in one.rec_result] == [r.text for r in want[0].rec_result]
    with pytest.raises(retto_amd.ImageError):
        hip_session.run_encoded_batch([files[0], files[1][:100]])
    assert hip_session.run_encoded_batch([]) == []


def test_run_stream_order_and_payloads(hip_session):
    """session.rs:133-143: per image Det, then Cls, then Rec; payloads equal the stage JSON of a plain run.  With
    several pages (3 lanes) every page still sees its own stages in order and each exactly once."""
    import json
    from retto_amd import workload
    pages, maps = [], []
    for i in range(5):
        page, rects = workload.planted_page(160 + 32 * i, 320, 2 + i % 2, seed=60 + i)
        pages.append(page)
        dh, dw = hip_session.det_preprocess(page).shape[2:]
        maps.append(workload.planted_map(dh, dw, page.shape[0], page.shape[1], rects))
    events = []
    hip_session.run_batch_stream(pages, lambda p, stage, payload: events.append((p, stage, payload)), det_map_override=maps)
    assert len(events) == 15
    for p in range(5):
        mine = [(s, pl) for q, s, pl in events if q == p]
        assert [s for s, _ in mine] == ["Det", "Cls", "Rec"]
        plain = hip_session.run_batch([pages[p]], det_map_override=[maps[p]])[0]
        assert len(mine[0][1]) == len(plain.det_result) > 0
        assert [d["score"] for d in mine[0][1]] == [float(np.float32(d.score)) for d in plain.det_result] or \
            np.allclose([d["score"] for d in mine[0][1]], [d.score for d in plain.det_result], rtol=1e-6)
        assert [[(pt["x"], pt["y"]) for pt in d["boxes"]["inner"]] for d in mine[0][1]] == \
            [[(pt.x, pt.y) for pt in d.boxes.inner] for d in plain.det_result]
        assert [c["label"]["label"] for c in mine[1][1]] == [c.label.label for c in plain.cls_result]
        assert [t["text"] for t in mine[2][1]] == [t.text for t in plain.rec_result]
    # the reference entry: one image, a sender
    got = []
    hip_session.run_stream(pages[0], lambda stage, payload: got.append(stage))
    assert got == ["Det", "Cls", "Rec"]


def test_det_stem_from_u8_pages_matches_tensor_path(hip_session):
    """rt_run_batch feeds the det stem with the RGB8 pages (normalisation folded in); rt_det gets the normalised f32
    tensor of rt_det_preprocess.  Both must produce the same probability map (checked through its checksum)."""
    import ctypes as C
    page = np.random.default_rng(77).integers(0, 256, (224, 352, 3), dtype=np.uint8)
    x = hip_session.det_preprocess(page)
    ref = float(hip_session.worker.det(x).astype(np.float64).sum())
    lib = hip_session._hd.lib
    r = hip_session.run_batch_raw([page], [page.shape[0]], [page.shape[1]])
    try:
        got = lib.rt_results_det_checksum(r)
    finally:
        lib.rt_results_free(r)
    assert abs(got - ref) <= 1e-9 * abs(ref) + 1e-6, (got, ref)


def test_rccl_weight_broadcast_single_rank():
    """The N > 1 bench path over RCCL (backend "nccl"), exercised with the one GPU this box has: process group on
    the device, weight-blob broadcast through device tensors, MAX / SUM all-reduce and barrier as bench.py uses them."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, %r)
from retto_amd import synth
from retto_amd.dist import broadcast_blobs, digest
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
blobs = list(synth.synth_models(0))
got = broadcast_blobs(blobs, 4, 0, device="cuda")
assert digest(got) == digest(blobs)
t = torch.tensor([1.5], dtype=torch.float64, device="cuda"); dist.all_reduce(t, op=dist.ReduceOp.MAX)
c = torch.tensor([7], dtype=torch.int64, device="cuda"); dist.all_reduce(c)
dist.barrier(); torch.cuda.synchronize()
assert float(t.item()) == 1.5 and int(c.item()) == 7
dist.destroy_process_group()
print("rccl ok")
''' % root
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0 and "rccl ok" in out.stdout, out.stderr[-2000:]


def test_cabi_rccl_broadcast_single_rank():
    """rt_rccl_unique_id + rt_broadcast_blobs (the C-ABI form of the one collective: what a Rust / C++ host calls) with the
    one GPU of this box: communicator of world 1, sizes + blobs through the device staging buffer, bytes unchanged."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r'''
import sys
sys.path.insert(0, %r)
from retto_amd import synth
from retto_amd.dist import broadcast_blobs_cabi, rccl_unique_id, digest
blobs = list(synth.synth_models(0))
uid = rccl_unique_id()
assert len(uid) == 128 and any(uid)
got = broadcast_blobs_cabi(blobs, 4, 0, 1, 0, uid)
assert digest(got) == digest(blobs)
print("cabi rccl ok")
''' % root
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0 and "cabi rccl ok" in out.stdout, (out.stdout[-500:], out.stderr[-2000:])


def test_bench_runs_the_cabi_broadcast_every_round():
    """bench.py's own multi-rank path with the one GPU of the box: RT_BENCH_FORCE_DIST=1 makes a world-1
    run take the distributed branch -- process group over RCCL, rt_rccl_unique_id + rt_broadcast_blobs for the weights (the
    C-ABI collective a Rust host would call), barrier + MAX all-reduce around the timed region -- so the collective code runs
    on the driver's box every round even though no 8-GPU node is available."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, RT_BENCH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29547", RANK="0", LOCAL_RANK="0",
               WORLD_SIZE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--bcast", "cabi", "--steps", "2", "--warmup", "1",
                          "--pages", "3", "--size", "480", "--lines", "4", "--no-c5", "--no-cpu-baseline", "--repeat", "0"],
                         capture_output=True, text=True, timeout=900, env=env, cwd=root)
    assert out.returncode == 0, out.stderr[-3000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert line["rccl_ranks"] == 1 and line["bcast_ms"] is not None and line["bcast_ms"] > 0
    assert line["n_gpus"] == 1 and line["value"] > 0 and line["selfcheck"]["batch_invariance_page0"] is True


@pytest.mark.parametrize("style", [0, 1, 2])
def test_onnx_import_against_unfolded_torch_forward(style):
    """The importer against an INDEPENDENT evaluation of what the file says: the .onnx files are written with un-folded
    parameters (Conv weights + separate BatchNormalization statistics, LearnableAffineBlock scalars in front of the activation,
    bias as its own Add, three spellings of hardswish); the torch oracle is fed those un-folded tensors as they stand in the
    file (conv -> bias -> BN with the file's epsilon -> affine), the HIP session is created from the .onnx bytes.  Agreement
    to 1e-4 checks the importer's matching AND its folding arithmetic, not the importer against itself."""
    import sys
    sys.path.insert(0, os.path.dirname(__file__))
    import torch
    import torch.nn.functional as F
    import retto_amd
    from oracle import nets_torch as N
    from retto_amd import synth
    from onnx_writer import build_model_onnx
    tens = [synth.det_tensors(1), synth.cls_tensors(3), synth.rec_tensors(2)]
    kinds = [retto_amd.MODEL_DET, retto_amd.MODEL_CLS, retto_amd.MODEL_REC]
    unf = [{}, {}, {}]
    onnx = [build_model_onnx(retto_amd.model_manifest(k), t, seed=30 + style, style=style, unfolded=u) for k, t, u in zip(kinds, tens, unf)]
    S = retto_amd.RettoWorkerModelSource
    cfg = retto_amd.synthetic_session_config(0)
    cfg.worker_config.models = retto_amd.RettoWorkerModelProvider(det=S.Blob(onnx[0]), rec=S.Blob(onnx[2]), cls=S.Blob(onnx[1]))
    sess = retto_amd.RettoSession(cfg)

    def unfolded_conv(u):
        def conv(w, name, x, stride=(1, 1), pad=(0, 0), groups=1):
            r = u.get(name)
            if r is None:                       # not a conv written by GraphWriter.conv (never happens for 4-d weights)
                return F.conv2d(x, w[name + ".w"], w.get(name + ".b"), stride=stride, padding=pad, groups=groups)
            y = F.conv2d(x, torch.from_numpy(r["w"]), None if r["b"] is None else torch.from_numpy(r["b"]), stride=stride, padding=pad, groups=groups)
            if r["bn"] is not None:
                s_, B, mean, var, eps = r["bn"]
                y = F.batch_norm(y, torch.from_numpy(mean), torch.from_numpy(var), torch.from_numpy(s_), torch.from_numpy(B), False, 0.0, eps)
            if r["lab"] is not None:
                y = y * float(r["lab"][0]) + float(r["lab"][1])
            return y
        return conv
    saved = N._conv
    try:
        rng = np.random.default_rng(5 + style)
        x = rng.uniform(-1, 1, (1, 3, 96, 160)).astype(np.float32)
        N._conv = unfolded_conv(unf[0])
        w = {k: torch.from_numpy(v) for k, v in tens[0].items()}
        ref = N.det_forward(w, torch.from_numpy(x)).numpy()
        # (the two transposed convs of the DB head are not routed through _conv: they stay folded in the oracle)
        assert np.abs(sess.worker.det(x) - ref).max() <= 1e-4
        x = rng.uniform(-1, 1, (3, 3, 48, 192)).astype(np.float32)
        N._conv = unfolded_conv(unf[1])
        ref = N.cls_forward({k: torch.from_numpy(v) for k, v in tens[1].items()}, torch.from_numpy(x)).numpy()
        assert np.abs(sess.worker.cls(x) - ref).max() <= 1e-4
        x = rng.uniform(-1, 1, (2, 3, 48, 320)).astype(np.float32)
        N._conv = unfolded_conv(unf[2])
        ref = N.rec_forward({k: torch.from_numpy(v) for k, v in tens[2].items()}, torch.from_numpy(x)).numpy()
        assert np.abs(sess.worker.rec(x) - ref).max() <= 2e-4
        n_bn = sum(1 for u in unf for r in u.values() if r["bn"] is not None)
        assert n_bn == 0 if style == 1 else n_bn > 10      # style 1 writes folded convs with the bias as a separate Add
        assert sum(1 for u in unf for r in u.values() if r["lab"] is not None) > 20
    finally:
        N._conv = saved
        sess.close()
