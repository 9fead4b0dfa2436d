"""GPU tests of the half-precision path (rt_config.dtype = RT_DTYPE_F16): the fp16 kernel family against torch,
the PP-OCRv4 mobile networks in fp16 and the PP-OCRv4 server networks (BASELINE.json config 5) against the fp32
torch-CPU oracle.  Tolerances are stated at each test: fp16 storage rounds every activation to 11 significant
bits, so the bar is a tolerance on the probabilities plus exact token ids / labels wherever the fp32 oracle's
decision margin exceeds that tolerance."""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import nets_torch as N
from oracle import ref_lib as R
import retto_amd
from retto_amd import synth, workload

pytestmark = pytest.mark.gpu

# Stated fp16 tolerances (absolute, on probabilities in [0, 1])
DET_ATOL_F16 = 4e-2       # DB probability map, mobile and server: max over the map; the mean error must stay below DET_MEAN_F16
DET_MEAN_F16 = 3e-3
CLS_ATOL_F16 = 2e-2
REC_ATOL_F16 = 3e-2       # per-class softmax probabilities
REC_MARGIN_F16 = 6e-2     # token ids must agree wherever the fp32 top-2 margin exceeds this
REC_EQUAL_FLOOR_F16 = 0.9  # ... and overall on at least this fraction of the time steps (measured: 0.99-1.0)


@pytest.fixture(scope="module")
def hip16():
    s = retto_amd.RettoSession(retto_amd.synthetic_session_config(0, dtype="f16"))
    yield s
    s.close()


@pytest.fixture(scope="module")
def server_models():
    return synth.synth_server_models(0)


@pytest.fixture(scope="module")
def hip_server():
    s = retto_amd.RettoSession(retto_amd.synthetic_session_config(0, server=True, dtype="f16"))
    yield s
    s.close()


def _model_info(sess):
    lib = sess.worker._hd.lib
    lib.rt_model_info.restype = C.c_char_p
    lib.rt_model_info.argtypes = [C.c_void_p]
    return lib.rt_model_info(sess.worker._hd.h).decode()


def test_model_selection(hip_session, hip16, hip_server):
    assert _model_info(hip_session) == "mobile/f32 f32 mobile/f32"
    assert _model_info(hip16) == "mobile/f16 f16 mobile/f16"
    assert _model_info(hip_server) == "server/f16 f16 server/f16"
    with pytest.raises(retto_amd.RettoError):   # the server graphs exist in fp16 only
        retto_amd.RettoSession(retto_amd.synthetic_session_config(0, server=True))


# ---------------------------------------------------------------- the implicit-GEMM conv kernel on its own
def _conv16(sess, x, w, b, sh, sw, act):
    lib, h = sess.worker._hd.lib, sess.worker._hd.h
    n, cin, H, W = x.shape
    cout, _, kh, kw = w.shape
    ho, wo = (H - 1) // sh + 1, (W - 1) // sw + 1
    out = np.empty((n, cout, ho, wo), np.float32)
    lib.rt_debug_conv16.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                    C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
    rc = lib.rt_debug_conv16(h, x.ctypes.data, n, cin, H, W, w.ctypes.data, cout, kh, kw, sh, sw,
                             b.ctypes.data if b is not None else None, act, out.ctypes.data)
    assert rc == 0, lib.rt_last_error(h)
    return out


@pytest.mark.parametrize("n,cin,cout,k,stride,hw,act", [
    (1, 8, 16, (3, 3), (2, 2), (64, 96), 0),        # stem-like: 8 input channels (one short slab)
    (2, 32, 32, (3, 3), (1, 1), (20, 37), 1),       # odd map sizes, one slab
    (1, 128, 128, (3, 3), (1, 1), (48, 48), 1),     # PPHGNet stage-1 layer
    (1, 160, 160, (3, 3), (1, 1), (24, 40), 1),     # 5 column tiles
    (1, 192, 192, (3, 3), (1, 1), (17, 33), 1),     # two 96-column blocks
    (2, 768, 224, (3, 3), (1, 1), (6, 40), 1),      # 224 = 128 + 96 columns, short maps (rec stage 4)
    (1, 256, 64, (9, 9), (1, 1), (30, 30), 0),      # LKPAN 9x9
    (1, 64, 64, (3, 3), (2, 2), (32, 48), 0),       # LKPAN stride-2 3x3
    (3, 32, 32, (7, 7), (1, 1), (15, 15), 0),       # IntraCL 7x7
    (1, 240, 480, (1, 1), (1, 1), (12, 80), 2),     # 1x1, hardswish
    (4, 480, 60, (1, 3), (1, 1), (1, 50), 3),       # SVTR 1x3, swish, N = 60 -> pitch 64
    (1, 80, 64, (2, 2), (1, 1), (16, 16), 1),       # 2.5 slabs
    (2, 96, 24, (3, 3), (1, 1), (40, 56), 0),       # mobile FPN 3x3
    (1, 1216, 512, (1, 1), (1, 1), (8, 20), 1),     # HG aggregation conv
    (2, 896, 256, (1, 1), (1, 1), (50, 61), 1),     # HG aggregation conv on >= 4096 pixels: the LDS-DMA GEMM kernel, ragged last tile
    (1, 1664, 768, (1, 1), (1, 1), (40, 120), 1),   # 3 channel blocks of 256, K = 52 slabs
    (1, 72, 128, (1, 1), (1, 1), (64, 80), 2),      # K = 2.25 slabs (odd stage tail + partial slab), 128-channel blocks
    (1, 2112, 1024, (1, 1), (1, 1), (30, 140), 0),  # largest aggregation conv (stage 4)
    (1, 384, 192, (1, 1), (1, 1), (64, 100), 1),    # GEMM kernel with one 192-wide channel block (3 fragments per wave)
    (1, 128, 64, (1, 1), (1, 1), (80, 83), 2),      # ... with a 64-wide block, ragged pixel tail
    (1, 480, 480, (1, 1), (1, 1), (72, 100), 2),    # mobile rec 1x1: 480 -> two 256-wide blocks, the second with 32 pad columns
    (1, 200, 240, (1, 1), (1, 1), (96, 64), 0),     # K = 6.25 slabs (partial last slab), N = 240 in one 256-wide block
])
def test_conv16_kernel(hip16, n, cin, cout, k, stride, hw, act):
    """fp16 inputs / weights, fp32 accumulation: against torch conv2d on the SAME fp16-rounded operands the only
    differences are the accumulation order and the final rounding to fp16 (2^-11 relative)."""
    rng = np.random.default_rng(cin * 131 + cout)
    H, W = hw
    kh, kw = k
    if (kh, kw) == (2, 2):
        pytest.skip("even kernels have no symmetric 'same' padding (used only with explicit pads by the PFHeadLocal path)")
    x = rng.standard_normal((n, cin, H, W)).astype(np.float16).astype(np.float32)
    w = (rng.standard_normal((cout, cin, kh, kw)) * np.sqrt(2.0 / (cin * kh * kw))).astype(np.float16).astype(np.float32)
    b = (rng.standard_normal(cout) * 0.1).astype(np.float32)
    got = _conv16(hip16, x, w, b, stride[0], stride[1], act)
    ref = F.conv2d(torch.from_numpy(x).double(), torch.from_numpy(w).double(), torch.from_numpy(b).double(), stride=stride,
                   padding=(kh // 2, kw // 2))
    ref = {0: lambda t: t, 1: F.relu, 2: N.hswish, 3: N.swish, 4: torch.sigmoid}[act](ref).numpy()
    assert got.shape == ref.shape
    err = np.abs(got - ref)
    tol = 2e-3 * np.abs(ref) + 2e-3     # one fp16 rounding of the result + fp32 accumulation noise
    assert (err <= tol).all(), f"max err {err.max()} at {np.unravel_index(err.argmax(), err.shape)} (ref {ref.flat[err.argmax()]})"


@pytest.mark.parametrize("n,cin,cout,k,hw", [
    (8, 128, 128, (3, 3), (120, 120)),     # row-wise LDS-DMA form: ring of 3 weight rows, two halo buffers, 57 tiles per image
    (8, 192, 192, (3, 3), (60, 96)),       # nine-tap form on 64-channel blocks (ring of 2)
    (4, 256, 64, (9, 9), (60, 60)),        # 9x9 rows, single halo buffer
    (1, 1664, 768, (1, 1), (160, 240)),    # pipelined GEMM: 150 pixel tiles x 3 channel blocks, 52 slabs through the ring of 4
    (1, 480, 240, (1, 1), (90, 333)),      # ragged last tile, 240 -> one 256-wide block
])
def test_conv16_dma_kernels_are_repeatable(hip16, n, cin, cout, k, hw):
    """Race screen for the hand-counted vmcnt / lgkmcnt waits of the LDS-DMA kernels: a read that overtakes its DMA, or a DMA
    that overwrites a buffer still being read, shows up as run-to-run differences long before it shows up against a tolerance.
    Many tiles per CU, the same launch repeated: every run must be bit-identical, and right."""
    rng = np.random.default_rng(cin + cout + k[0])
    H, W = hw
    x = rng.standard_normal((n, cin, H, W)).astype(np.float16).astype(np.float32)
    w = (rng.standard_normal((cout, cin, k[0], k[1])) * np.sqrt(2.0 / (cin * k[0] * k[1]))).astype(np.float16).astype(np.float32)
    b = (rng.standard_normal(cout) * 0.1).astype(np.float32)
    first = _conv16(hip16, x, w, b, 1, 1, 1)
    for rep in range(7):
        again = _conv16(hip16, x, w, b, 1, 1, 1)
        assert np.array_equal(first, again), f"run {rep + 2} differs from run 1 in {(first != again).sum()} values"
    ref = F.relu(F.conv2d(torch.from_numpy(x[:1]).double(), torch.from_numpy(w).double(), torch.from_numpy(b).double(),
                          padding=(k[0] // 2, k[1] // 2))).numpy()
    err = np.abs(first[:1] - ref)
    assert (err <= 2e-3 * np.abs(ref) + 2e-3).all(), f"max err {err.max()}"


# ---------------------------------------------------------------- mobile networks in fp16 vs the fp32 oracle
@pytest.mark.parametrize("n,h,w", [(1, 64, 96), (2, 160, 128), (1, 320, 320), (2, 960, 960)])
def test_det_net_f16(hip16, oracle_session, n, h, w):
    x = np.random.default_rng(h + w).uniform(-1, 1, (n, 3, h, w)).astype(np.float32)
    got = hip16.worker.det(x)
    ref = N.det_forward(oracle_session.wd, torch.from_numpy(x)).numpy()
    assert got.shape == ref.shape and np.isfinite(got).all()
    err = np.abs(got - ref)
    assert err.max() <= DET_ATOL_F16 and err.mean() <= DET_MEAN_F16, f"det map max abs err {err.max()} (mean {err.mean()})"
    # the thresholded mask agrees wherever the fp32 map is not within the tolerance of the 0.3 threshold
    far = np.abs(ref - 0.3) > DET_ATOL_F16
    assert ((got > 0.3) == (ref > 0.3))[far].all()


@pytest.mark.parametrize("n", [7, 300])
def test_cls_net_f16(hip16, oracle_session, n):
    x = np.random.default_rng(5).uniform(-1, 1, (n, 3, 48, 192)).astype(np.float32)
    got = hip16.worker.cls(x)
    ref = N.cls_forward(oracle_session.wc, torch.from_numpy(x)).numpy()
    assert np.abs(got - ref).max() <= CLS_ATOL_F16
    decisive = np.abs(ref[:, 1] - ref[:, 0]) > 2 * CLS_ATOL_F16
    assert (got.argmax(1) == ref.argmax(1))[decisive].all()


def _check_rec(got, ref, atol, margin):
    assert got.shape == ref.shape and np.isfinite(got).all()
    err = np.abs(got - ref).max()
    assert err <= atol, f"rec prob max abs err {err}"
    ga, ra = got.argmax(-1), ref.argmax(-1)
    top2 = np.sort(ref, -1)[..., -2:]
    decisive = (top2[..., 1] - top2[..., 0]) > margin
    assert (ga[decisive] == ra[decisive]).all()
    return float(decisive.mean()), float((ga == ra).mean())


# (120 x 400: the fp16 twin of test_rec_net's production-size case -- 144 000 rows at the 240-channel stages, k_gemm16p on its full-size dispatch)
@pytest.mark.parametrize("n,w", [(1, 320), (3, 321), (2, 487), (6, 640), (1, 1600), (120, 400)])
def test_rec_net_f16(hip16, oracle_session, n, w):
    x = np.random.default_rng(w).uniform(-1, 1, (n, 3, 48, w)).astype(np.float32)
    x[:, :, :, w // 2:] = 0.0
    got = hip16.worker.rec(x)
    ref = N.rec_forward(oracle_session.wr, torch.from_numpy(x)).numpy()
    frac_decisive, frac_equal = _check_rec(got, ref, REC_ATOL_F16, REC_MARGIN_F16)
    print(f"rec f16 n={n} w={w}: decisive {frac_decisive:.3f}, argmax equal {frac_equal:.3f}")
    assert frac_equal > REC_EQUAL_FLOOR_F16


# ---------------------------------------------------------------- server networks (config 5) vs the fp32 oracle
@pytest.mark.parametrize("n,h,w", [(1, 64, 96), (2, 160, 128), (1, 320, 416)])
def test_server_det_net(hip_server, server_models, n, h, w):
    wd = N.read_blob(server_models[0])
    x = np.random.default_rng(h * 7 + w).uniform(-1, 1, (n, 3, h, w)).astype(np.float32)
    got = hip_server.worker.det(x)
    ref = N.sdet_forward(wd, torch.from_numpy(x)).numpy()
    assert got.shape == ref.shape == (n, 1, h, w) and np.isfinite(got).all()
    err = np.abs(got - ref)
    assert err.max() <= DET_ATOL_F16 and err.mean() <= DET_MEAN_F16, f"server det map max abs err {err.max()} (mean {err.mean()})"
    print(f"server det {n}x{h}x{w}: max err {err.max():.4f}, mean {err.mean():.5f}")
    far = np.abs(ref - 0.3) > DET_ATOL_F16
    assert ((got > 0.3) == (ref > 0.3))[far].all()


@pytest.mark.parametrize("n,w", [(1, 320), (3, 333), (2, 640)])
def test_server_rec_net(hip_server, server_models, n, w):
    wr = N.read_blob(server_models[2])
    x = np.random.default_rng(w + 1).uniform(-1, 1, (n, 3, 48, w)).astype(np.float32)
    x[:, :, :, (2 * w) // 3:] = 0.0
    got = hip_server.worker.rec(x)
    ref = N.srec_forward(wr, torch.from_numpy(x)).numpy()
    frac_decisive, frac_equal = _check_rec(got, ref, REC_ATOL_F16, REC_MARGIN_F16)
    print(f"server rec n={n} w={w}: decisive {frac_decisive:.3f}, argmax equal {frac_equal:.3f}")
    # C5 is a TOLERANCE configuration: token ids are only required equal where the fp32 margin exceeds REC_MARGIN_F16;
    # the floor below is what the fp16 arithmetic keeps of the fp32 argmax overall on these weights
    assert frac_equal > REC_EQUAL_FLOOR_F16


# ---------------------------------------------------------------- whole pipeline in fp16: discrete stages stay bit-exact
@pytest.mark.parametrize("which", ["mobile", "server"])
def test_pipeline_f16_teacher_forced(hip16, hip_server, models, server_models, which):
    """The stages around the networks are the same kernels in every dtype: with the oracle's three workers replaced by
    the fp16 HIP worker (identical tensors on both sides), boxes / labels / tokens must be bit-exact."""
    from oracle.pipeline import OracleSession
    sess = hip16 if which == "mobile" else hip_server
    det, cls, rec, dic = models if which == "mobile" else server_models
    o = OracleSession(*(models[:3] + (dic,)))   # the oracle's own nets are unused: all three workers are replaced
    o.det_worker, o.cls_worker, o.rec_worker = sess.worker.det, sess.worker.cls, sess.worker.rec
    h, w, lines = 320, 480, 5
    page, rects = workload.planted_page(h, w, lines, seed=11)
    dh, dw = R.resize_either_dims(h, w)
    pmap = workload.planted_map(dh, dw, h, w, rects)
    got = sess.run_batch([page], det_map_override=[pmap])[0]
    ref = o.run(page, det_map_override=pmap)
    assert len(got.det_result) == len(ref.det_boxes) == lines
    assert np.array_equal(np.stack([d.boxes.as_array() for d in got.det_result]), ref.det_boxes)
    assert [c.label.label for c in got.cls_result] == list(ref.cls_labels)
    for g, t in zip(got.rec_result, ref.rec_tokens):
        assert np.array_equal(g.tokens, t)


# ---------------------------------------------------------------- BASELINE config 5 at its real size
def test_server_det_net_full_page(hip_server, server_models):
    """The server det net on one full 960 x 960 page (the C5 page size: the LDS-DMA kernels pick their tile shapes,
    ring depths and the XCD remap from the launch size) against the fp32 torch oracle, same tolerances as the small cases."""
    wd = N.read_blob(server_models[0])
    x = np.random.default_rng(960).uniform(-1, 1, (1, 3, 960, 960)).astype(np.float32)
    got = hip_server.worker.det(x)
    ref = N.sdet_forward(wd, torch.from_numpy(x)).numpy()
    assert got.shape == ref.shape == (1, 1, 960, 960) and np.isfinite(got).all()
    err = np.abs(got - ref)
    print(f"server det 1x960x960: max err {err.max():.4f}, mean {err.mean():.5f}")
    assert err.max() <= DET_ATOL_F16 and err.mean() <= DET_MEAN_F16
    far = np.abs(ref - 0.3) > DET_ATOL_F16
    assert ((got > 0.3) == (ref > 0.3))[far].all()


def test_server_rec_net_full_batch(hip_server, server_models):
    """The server rec net at a full line group (24 lines x 640: the GEMM / conv shapes of a C5 step) against the oracle."""
    wr = N.read_blob(server_models[2])
    n, w = 24, 640
    x = np.random.default_rng(2464).uniform(-1, 1, (n, 3, 48, w)).astype(np.float32)
    for i in range(n):
        x[i, :, :, 320 + 13 * i:] = 0.0   # ragged zero padding like resize_norm_image
    got = hip_server.worker.rec(x)
    ref = N.srec_forward(wr, torch.from_numpy(x)).numpy()
    frac_decisive, frac_equal = _check_rec(got, ref, REC_ATOL_F16, REC_MARGIN_F16)
    print(f"server rec n={n} w={w}: decisive {frac_decisive:.3f}, argmax equal {frac_equal:.3f}")
    assert frac_equal > REC_EQUAL_FLOOR_F16


def test_c5_full_size_properties(hip_server, models, server_models):
    """BASELINE config 5 at its per-GPU size (1024 pages / 8 GPUs = 128 pages of 960 x 960 in ONE batch -- the dispatch sizes a
    `bench.py --workload c5` step produces --, 32 planted lines each, PP-OCRv4 server graphs in fp16, 3 lanes) through
    size-independent properties, the fp16 counterpart of test_c3_full_size_properties:
    (1) batch / lane / order composition does not change a page's discrete results -- the batch equals the same pages
        shuffled and pages run alone: boxes, box scores, cls labels and token ids bit for bit; line / label scores equal to the
        fp16 tolerance (a page alone runs other kernel shapes: tile sizes follow the launch size);
    (2) two pages of the batch equal the oracle pipeline teacher-forced by the fp16 worker in the reference's batches of 6;
    (3) the planted lines come back: 32 boxes per page, each inside its planted rectangle grown by the unclip offset."""
    n = 128
    pages, maps, rects = [], [], []
    for i in range(n):
        page, rc = workload.planted_page(960, 960, 32, seed=100 + i)
        pages.append(page); rects.append(rc)
        maps.append(workload.planted_map(960, 960, 960, 960, rc))
    full = hip_server.run_batch(pages, det_map_override=maps)
    assert np.isfinite(hip_server.last_det_checksum)

    def same(a, b, strict):
        assert len(a.det_result) == len(b.det_result) == 32
        assert np.array_equal(np.stack([d.boxes.as_array() for d in a.det_result]), np.stack([d.boxes.as_array() for d in b.det_result]))
        assert [d.score for d in a.det_result] == [d.score for d in b.det_result]
        assert [c.label.label for c in a.cls_result] == [c.label.label for c in b.cls_result]
        for x, y in zip(a.rec_result, b.rec_result):
            assert np.array_equal(x.tokens, y.tokens) and x.text == y.text
        sa, sb = np.array([x.score for x in a.rec_result]), np.array([y.score for y in b.rec_result])
        ca, cb = np.array([c.label.score for c in a.cls_result]), np.array([c.label.score for c in b.cls_result])
        if strict:
            assert np.array_equal(sa, sb, equal_nan=True) and np.array_equal(ca, cb)
        else:
            np.testing.assert_allclose(sa, sb, atol=REC_ATOL_F16, equal_nan=True)
            np.testing.assert_allclose(ca, cb, atol=CLS_ATOL_F16)

    from oracle.pipeline import OracleSession
    o = OracleSession(*(models[:3] + (server_models[3],)))   # the oracle's own nets are unused: all three workers are replaced
    o.det_worker, o.cls_worker, o.rec_worker = hip_server.worker.det, hip_server.worker.cls, hip_server.worker.rec
    for j in (0, 21, 100):
        ref = o.run(pages[j], det_map_override=maps[j])
        assert len(ref.det_boxes) == 32
        assert np.array_equal(np.stack([d.boxes.as_array() for d in full[j].det_result]), ref.det_boxes)
        assert [c.label.label for c in full[j].cls_result] == list(ref.cls_labels)
        for g, t in zip(full[j].rec_result, ref.rec_tokens):
            assert np.array_equal(g.tokens, t)
        np.testing.assert_allclose([g.score for g in full[j].rec_result], ref.rec_scores, atol=REC_ATOL_F16, equal_nan=True)

    perm = np.random.default_rng(5).permutation(n)
    shuffled = hip_server.run_batch([pages[j] for j in perm], det_map_override=[maps[j] for j in perm])
    for k, j in enumerate(perm):
        same(shuffled[k], full[j], strict=False)
    again = hip_server.run_batch(pages, det_map_override=maps)      # the same call twice: bit-identical (no races in the LDS-DMA kernels)
    for a, b in zip(again, full):
        same(a, b, strict=True)
    for j in (0, 9, 31, 127):
        alone = hip_server.run_batch([pages[j]], det_map_override=[maps[j]])[0]
        same(alone, full[j], strict=False)
    for r, rc in zip(full, rects):
        got = np.stack([d.boxes.as_array() for d in r.det_result]).reshape(-1, 4, 2)
        for x0, y0, x1, y1 in rc:
            inside = [(b[:, 0].min() >= x0 - 12 and b[:, 0].max() <= x1 + 12 and b[:, 1].min() >= y0 - 12 and b[:, 1].max() <= y1 + 12)
                      for b in got]
            assert sum(inside) == 1, (x0, y0, x1, y1)


def test_server_session_from_onnx_files(hip_server, server_models):
    """rt_create takes the server .onnx files themselves through the det / rec sources (ort_worker.rs:120-135: give it an
    .onnx, it runs).  The files are written with UN-FOLDED parameters (Conv + BatchNormalization statistics, bias as a
    separate Add in style 1); the session built from them must agree with (a) a torch evaluation of what the files say --
    conv -> BN with the file's epsilon -- within the fp16 tolerances and (b) the session built from the RTWB blobs to fp16
    rounding of the re-folded weights."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(__file__))
    from onnx_writer import build_model_onnx
    tens = [synth.sdet_tensors(4), synth.cls_tensors(3), synth.srec_tensors(5)]
    assert synth.pack_blob(tens[0]) == server_models[0] and synth.pack_blob(tens[2]) == server_models[2]
    unf = [{}, {}, {}]
    kinds = [retto_amd.MODEL_SDET, retto_amd.MODEL_CLS, retto_amd.MODEL_SREC]
    onnx = [build_model_onnx(retto_amd.model_manifest(k), t, seed=50 + i, style=i % 2 * 1, unfolded=u) for i, (k, t, u) in enumerate(zip(kinds, tens, unf))]
    S = retto_amd.RettoWorkerModelSource
    cfg = retto_amd.synthetic_session_config(0, server=True, dtype="f16")
    cfg.worker_config.models = retto_amd.RettoWorkerModelProvider(det=S.Blob(onnx[0]), rec=S.Blob(onnx[2]), cls=S.Blob(onnx[1]))
    sess = retto_amd.RettoSession(cfg)

    def unfolded_conv(u):
        def conv(w, name, x, stride=(1, 1), pad=(0, 0), groups=1):
            r = u[name]
            y = F.conv2d(x, torch.from_numpy(r["w"]), None if r["b"] is None else torch.from_numpy(r["b"]), stride=stride, padding=pad, groups=groups)
            if r["bn"] is not None:
                s_, B, mean, var, eps = r["bn"]
                y = F.batch_norm(y, torch.from_numpy(mean), torch.from_numpy(var), torch.from_numpy(s_), torch.from_numpy(B), False, 0.0, eps)
            return y
        return conv
    saved = N._conv
    try:
        assert _model_info(sess) == "server/f16 f16 server/f16"
        rng = np.random.default_rng(77)
        x = rng.uniform(-1, 1, (1, 3, 160, 224)).astype(np.float32)
        N._conv = unfolded_conv(unf[0])
        ref = N.sdet_forward({k: torch.from_numpy(v) for k, v in tens[0].items()}, torch.from_numpy(x)).numpy()
        got = sess.worker.det(x)
        err = np.abs(got - ref)
        assert err.max() <= DET_ATOL_F16 and err.mean() <= DET_MEAN_F16
        assert np.abs(got - hip_server.worker.det(x)).max() <= 2e-2       # same graph from the RTWB blob
        x = rng.uniform(-1, 1, (2, 3, 48, 352)).astype(np.float32)
        x[:, :, :, 250:] = 0.0
        N._conv = unfolded_conv(unf[2])
        ref = N.srec_forward({k: torch.from_numpy(v) for k, v in tens[2].items()}, torch.from_numpy(x)).numpy()
        frac_decisive, frac_equal = _check_rec(sess.worker.rec(x), ref, REC_ATOL_F16, REC_MARGIN_F16)
        assert frac_equal > REC_EQUAL_FLOOR_F16
        assert sum(1 for r in unf[0].values() if r["bn"] is not None) > 50     # the det file really carried un-folded BatchNorms
    finally:
        N._conv = saved
        sess.close()
