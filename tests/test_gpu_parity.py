"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle on the
same seeded inputs.  Integer / byte / index results must be bit-exact; floating-point
network outputs are compared with the tolerance written at each test."""
import os

import numpy as np
import pytest
import torch

from oracle import nets_torch as N
from oracle import ref_lib as R
import retto_amd
from retto_amd import workload

pytestmark = pytest.mark.gpu

# fp32 tolerance on network outputs (<= 1e-4 on the DB probability map)
DET_ATOL = 1e-4


def _rand_page(h, w, seed):
    return np.random.default_rng(seed).integers(0, 256, (h, w, 3), dtype=np.uint8)


# ---------------------------------------------------------------- a4 / a9 / a11 networks
# (3 x 960^2: the squeeze-excite levels reach the sizes where the fused pooling / scaled-GEMM path is taken)
# (1 x 960^2 is BASELINE.json configs[1] -- C2's literal workload: one page per call, the small-batch dispatch of every layer)
@pytest.mark.parametrize("n,h,w", [(2, 32, 64), (1, 64, 96), (2, 160, 128), (1, 320, 320), (1, 960, 960), (3, 960, 960), (1, 1984, 1408)])
def test_det_net(hip_session, oracle_session, n, h, w):
    x = np.random.default_rng(h + w).uniform(-1, 1, (n, 3, h, w)).astype(np.float32)
    got = hip_session.worker.det(x)
    ref = N.det_forward(oracle_session.wd, torch.from_numpy(x)).numpy()
    assert got.shape == ref.shape == (n, 1, h, w)
    assert np.isfinite(got).all()
    err = np.abs(got - ref).max()
    assert err <= DET_ATOL, f"det map max abs err {err}"


@pytest.mark.parametrize("n", [7, 600])  # 600 crops: the GEMM / depthwise shapes of a full page batch
def test_cls_net(hip_session, oracle_session, n):
    x = np.random.default_rng(5).uniform(-1, 1, (n, 3, 48, 192)).astype(np.float32)
    x[3] *= 0.1
    got = hip_session.worker.cls(x)
    ref = N.cls_forward(oracle_session.wc, torch.from_numpy(x)).numpy()
    assert got.shape == (n, 2)
    assert np.abs(got - ref).max() <= 1e-4


# (12 / 24 x 640: 11.5 k / 23 k pixels at the squeeze-excite levels -- the fused pooling + scaled 128 x 128 / 128 x 240
#  GEMM tiles, which smaller batches never reach)
#  (120 x 400: 144 000 rows at the 240-channel stages -- past the M >= 131072 threshold of the persistent LDS-DMA GEMM k_gemm32p, its
#  squeeze-excite form k_gemm32p+se and k_gemm_wide<4,...>, so the production-size kernels are compared with the oracle directly;
#  1000 x 96: 144 rows per line at the 480-channel squeeze-excite level -- 256-row blocks that span three lines, the +se kernel's
#  third scale slot)
@pytest.mark.parametrize("n,w", [(1, 320), (3, 321), (2, 487), (1, 960), (12, 640), (24, 640), (1, 3648), (120, 400), (1000, 96),
                                 (130, 412)])  # (130, 412): 160 680 rows at the 128-channel stage = k_gemm32w with a partial last 64-row tile
def test_rec_net(hip_session, oracle_session, n, w):
    x = np.random.default_rng(w).uniform(-1, 1, (n, 3, 48, w)).astype(np.float32)
    x[:, :, :, w // 2:] = 0.0  # zero padding like resize_norm_image
    got = hip_session.worker.rec(x)
    ref = N.rec_forward(oracle_session.wr, torch.from_numpy(x)).numpy()
    assert got.shape == ref.shape
    assert np.abs(got - ref).max() <= 2e-4
    # token ids: bit-exact wherever the oracle's top-2 margin is not a rounding tie
    ga, ra = got.argmax(-1), ref.argmax(-1)
    top2 = np.sort(ref, -1)[..., -2:]
    decisive = (top2[..., 1] - top2[..., 0]) > 1e-4
    assert (ga[decisive] == ra[decisive]).all()
    assert decisive.mean() > 0.9


def test_rec_net_ragged_production_size(hip_session, oracle_session):
    """The form rt_run_batch launches: lines of DIFFERENT widths in one ragged launch series (rt_rec_ragged), large enough to
    cross every dispatch threshold of the recognition net (173 k rows at the 240-channel stages: k_gemm32p, k_gemm32p+se,
    k_gemm_wide; lines of 48..96 pixels put 128..255-row images under the +se kernel's 256-row blocks), against the torch
    oracle evaluated width by width (rec_processor.rs:214-270 pads a batch to one width; lines are independent)."""
    rng = np.random.default_rng(2024)
    widths = [48, 56, 64, 72, 88, 96, 120, 160, 200, 248, 320, 336, 400, 487, 560, 640]
    lines = []
    for i in range(640):
        w = widths[(i * 7 + i // 16) % len(widths)]
        x = rng.uniform(-1, 1, (3, 48, w)).astype(np.float32)
        x[:, :, w - (i % 5) * (w // 8):] = 0.0   # right-hand zero padding like resize_norm_image
        lines.append(x)
    rows240 = sum(12 * (((l.shape[2] - 1) // 2 + 1 - 1) // 2 + 1) for l in lines)
    assert rows240 >= 131072, rows240
    got = hip_session.worker.rec_ragged(lines)
    by_w = {}
    for i, l in enumerate(lines):
        by_w.setdefault(l.shape[2], []).append(i)
    worst, dec_n, dec_eq, tot = 0.0, 0, 0, 0
    for w, idx in by_w.items():
        ref = N.rec_forward(oracle_session.wr, torch.from_numpy(np.stack([lines[i] for i in idx]))).numpy()
        for k, i in enumerate(idx):
            g, r = got[i], ref[k]
            assert g.shape == r.shape, (w, g.shape, r.shape)
            worst = max(worst, float(np.abs(g - r).max()))
            top2 = np.sort(r, -1)[..., -2:]
            decisive = (top2[..., 1] - top2[..., 0]) > 1e-4
            dec_n += int(decisive.sum()); tot += decisive.size
            dec_eq += int((g.argmax(-1)[decisive] == r.argmax(-1)[decisive]).sum())
    assert worst <= 2e-4, worst
    assert dec_eq == dec_n and dec_n > 0.9 * tot, (dec_eq, dec_n, tot)


# ---- split-bf16 form of the wide rec-net GEMMs (round 6; opt-in, rt_debug_set_variants bit 12) ----------------------------------
def _gemm_err(hip_session, M, K, N, variant, seed):
    import ctypes as C
    lib, h = hip_session._hd.lib, hip_session._hd.h
    lib.rt_bench_gemm_err.argtypes = [C.c_void_p, C.c_longlong, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint, C.POINTER(C.c_double)]
    o = (C.c_double * 4)()
    assert lib.rt_bench_gemm_err(h, M, K, N, variant, 512, 0, seed, o) == 0, lib.rt_last_error(h)
    return list(o)   # max |err|, rms err, max |ref|, rms ref


@pytest.mark.parametrize("M,K,N", [(65613, 240, 240), (40000, 480, 480), (70001, 128, 240)])
def test_split_bf16_gemm_error_against_fp64(hip_session, M, K, N):
    """The evidence behind the split-bf16 GEMM (nn_gemm_split.hip): operands with FULL 24-bit significands, bias 0, no activation;
    3 x 512 rows (start, middle, the partial last row block) against sum_k (double)a (double)w on the host.  The split form
    (variant 40: 3 bf16 planes per operand, 6 bf16 MFMAs per product, fp32 accumulate) must be NO WORSE than the fp32-MFMA
    kernel (variant 30, k_gemm32p: v_mfma_f32_16x16x4_f32) on the same data in the rms error, seed by seed; its maximum error --
    one extreme of 1.5 k x N samples -- may not exceed the fp32 kernel's by more than a tenth (measured: 0.79 / 0.71 / 1.05 of
    it at K = 240 / 480 / 128); and both must sit at fp32 rounding level (rms error <= 1e-6 of the result's rms)."""
    for seed in (1, 2, 3):
        e32 = _gemm_err(hip_session, M, K, N, 30, seed)
        esp = _gemm_err(hip_session, M, K, N, 40, seed)
        assert esp[2] == e32[2] and esp[3] == e32[3]            # same reference
        assert esp[0] <= 1.1 * e32[0], (seed, esp, e32)         # max |err|
        assert esp[1] <= e32[1], (seed, esp, e32)               # rms err
        assert e32[1] <= 1e-6 * e32[3] and esp[1] <= 1e-6 * e32[3]   # both at fp32 rounding level (rms ref ~2.4: 2.4e-6)


@pytest.mark.parametrize("n,w", [(120, 400), (1000, 96), (130, 412)])
def test_rec_net_split_bf16(hip_session, oracle_session, n, w):
    """test_rec_net's production-size cases with the split-bf16 kernels switched in (the 240- / 480-channel pointwise convs of
    >= 32768 rows, the two squeeze-excite layers included -- [1000-96]: 144-row lines, 128-row tiles that span two lines): the SAME bars -- <= 2e-4 on the softmax output against the torch fp32
    oracle, argmax exact wherever the oracle's top-2 margin exceeds 1e-4."""
    lib = hip_session._hd.lib
    x = np.random.default_rng(w).uniform(-1, 1, (n, 3, 48, w)).astype(np.float32)
    x[:, :, :, w // 2:] = 0.0
    base = hip_session.worker.rec(x)
    lib.rt_debug_set_variants(0, 0, 4096)
    try:
        got = hip_session.worker.rec(x)
    finally:
        lib.rt_debug_set_variants(0, 0, 0)
    ref = N.rec_forward(oracle_session.wr, torch.from_numpy(x)).numpy()
    assert not np.array_equal(got, base)                # the split kernels did run (a different summation)
    assert np.abs(got - ref).max() <= 2e-4
    ga, ra = got.argmax(-1), ref.argmax(-1)
    top2 = np.sort(ref, -1)[..., -2:]
    decisive = (top2[..., 1] - top2[..., 0]) > 1e-4
    assert (ga[decisive] == ra[decisive]).all()
    assert decisive.mean() > 0.9


@pytest.mark.parametrize("M", [131072 + 1, 131072 + 255, 140003, 200000 + 8 * 31])
def test_wide_gemms_with_a_partial_last_row_block(hip_session, M):
    """The persistent wide GEMMs on row counts that end inside a tile (256 rows for k_gemm32p, 128 for k_gemm_split; the rows past
    M are range-checked by the buffer descriptor of the wave's requests -- ADVICE round 5): k_gemm32p is bit-identical to the
    narrow fp32 kernel over the first and the last 4 M outputs (rt_bench_gemm compares exactly those), the split-bf16 kernel within
    its error bound, for K = N = 240 and for the 480-channel shape."""
    import ctypes as C
    lib, h = hip_session._hd.lib, hip_session._hd.h
    lib.rt_bench_gemm.argtypes = [C.c_void_p, C.c_longlong, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    for K, N in ((240, 240), (480, 480)):
        for variant, bound in ((30, 0.0), (40, 2e-5)):
            ms, md = C.c_float(), C.c_float(-1)
            assert lib.rt_bench_gemm(h, M, K, N, variant, 1, C.byref(ms), C.byref(md)) == 0, lib.rt_last_error(h)
            assert 0.0 <= md.value <= bound, (M, K, N, variant, md.value)


@pytest.mark.parametrize("mode", ["split_bf16", "fp32_mfma", "c5_f16"])
def test_c3_batches_are_repeatable_with_three_lanes_and_batches_in_flight(mode):
    """Race screen at the size and in the shape bench.py times (tools/soak_split.py): three C3 batches of 32 pages x 32 lines rotate,
    submitted two ahead from host memory over three lanes; every batch must come out bit-identical to its first run.  With one
    lane the kernels see an idle chip and a warm instruction cache; it took the lanes running side by side to show that
    k_gemm_split's shared squeeze-excite slot could be overwritten in front of a late wave's prologue read (round 6: one line of
    1024 different in ~3 % of the runs; every wave now requests its own copy)."""
    import subprocess, sys
    env = dict(os.environ)
    env.pop("RT_GEMM_SPLIT", None)
    if mode == "fp32_mfma":
        env["SOAK_FP32"] = "1"
    if mode == "c5_f16":   # the server graphs in fp16, 16 pages per batch
        env["SOAK_C5"] = "1"; env["SOAK_PAGES"] = "16"
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "soak_split.py"), "90"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "0 differing batches" in r.stdout


def test_rec_net_split_bf16_is_repeatable(hip_session):
    """Race screen for the hand-kept vmcnt bookkeeping of k_gemm_split (run-time counts of the operations that may stay in flight at
    every barrier wait, a dynamic tile queue, LDS-DMA rings): a fragment read that overtakes its request, or a request that
    overwrites a ring slot still being read, shows up as run-to-run differences long before it shows up against a tolerance.
    Many tiles per workgroup (410 k rows at the 240-channel stages), the same launch series eight times: bit-identical."""
    lib = hip_session._hd.lib
    x = np.random.default_rng(77).uniform(-1, 1, (340, 3, 48, 400)).astype(np.float32)
    x[:, :, :, 300:] = 0.0
    lib.rt_debug_set_variants(0, 0, 4096)
    try:
        first = hip_session.worker.rec(x)
        for _ in range(7):
            again = hip_session.worker.rec(x)
            assert np.array_equal(first.view(np.uint32), again.view(np.uint32))
    finally:
        lib.rt_debug_set_variants(0, 0, 0)
    assert np.isfinite(first).all()


def test_pipeline_split_bf16_equals_fp32_mfma_pipeline(hip_session):
    """A C3-shaped batch (8 pages of 960 x 960, 32 planted lines each: 98 k rows at the 240-channel stages per lane part, above the
    split kernels' 32768-row threshold) through rt_run_batch with the split-bf16 kernels on and off: boxes, labels and every
    line's token ids identical, line scores within 1e-5 (the two forms differ in the order of an fp32 summation, nothing else)."""
    lib = hip_session._hd.lib
    pages, maps = [], []
    for i in range(8):
        page, rects = workload.planted_page(960, 960, 32, seed=500 + i)
        pages.append(page); maps.append(workload.planted_map(960, 960, 960, 960, rects))
    lib.rt_set_lanes(hip_session._hd.h, 1)      # one lane: the whole batch in one launch series (the split kernels' sizes)
    try:
        base = hip_session.run_batch(pages, det_map_override=maps)
        lib.rt_debug_set_variants(0, 0, 4096)
        try:
            got = hip_session.run_batch(pages, det_map_override=maps)
        finally:
            lib.rt_debug_set_variants(0, 0, 0)
    finally:
        lib.rt_set_lanes(hip_session._hd.h, 1 << 20)
    n_lines, worst = 0, 0.0
    for b, g in zip(base, got):
        assert len(b.det_result) == len(g.det_result) == 32
        assert np.array_equal(np.stack([d.boxes.as_array() for d in b.det_result]), np.stack([d.boxes.as_array() for d in g.det_result]))
        assert [c.label.label for c in b.cls_result] == [c.label.label for c in g.cls_result]
        for x, y in zip(b.rec_result, g.rec_result):
            assert np.array_equal(x.tokens, y.tokens) and x.text == y.text
            if x.score == x.score:
                worst = max(worst, abs(x.score - y.score))
            n_lines += 1
    assert n_lines == 256 and worst <= 1e-5, worst
    assert any(not (x.score == y.score) for b, g in zip(base, got) for x, y in zip(b.rec_result, g.rec_result) if x.score == x.score)   # the split kernels did run


# ---------------------------------------------------------------- a2 / a3 preprocessing
@pytest.mark.parametrize("h,w", [(640, 640), (96, 160), (2100, 1300), (20, 300), (50, 200)])
def test_resize_both(hip_session, h, w):
    img = _rand_page(h, w, 1)
    got = hip_session.resize_both(img)
    ref = R.resize_both(img)
    assert got.shape == ref.shape
    assert np.array_equal(got, ref)


@pytest.mark.parametrize("h,w", [(640, 640), (960, 960), (96, 160), (736, 1312), (50, 200)])
def test_det_preprocess(hip_session, h, w):
    img = _rand_page(h, w, 2)
    got = hip_session.det_preprocess(img)
    ref = R.det_preprocess(img)
    assert got.shape == ref.shape
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))


# ---------------------------------------------------------------- a5 DB post-processing
def _planted_cases():
    cases = []
    page, rects = workload.planted_page(320, 480, lines=6, seed=3)
    cases.append(("axis", workload.planted_map(320, 480, 320, 480, rects), 320, 480))
    cases.append(("rot", workload.planted_map_rotated(384, 512, [(256, 100, 150, 14, 12.0), (200, 250, 120, 10, -31.0),
                                                                   (400, 200, 100, 12, 83.0), (90, 330, 60, 9, 45.0)]), 384, 512))
    m = workload.planted_map(256, 256, 256, 256, [(0, 0, 120, 30), (200, 100, 256, 140), (10, 230, 200, 256)])
    cases.append(("border", m, 256, 256))
    m = workload.planted_map(256, 320, 256, 320, [(20, 20, 300, 120)], shrink=0.0)
    m[50:90, 60:260] = 0.02  # a hole
    m[60:80, 100:200] = 0.9  # an island inside the hole
    cases.append(("nested", m, 256, 320))
    m = workload.planted_map(736, 736, 640, 640, workload.planted_page(640, 640, 12, 4)[1])
    cases.append(("scaled", m, 640, 640))
    rng = np.random.default_rng(7)
    cases.append(("noise", rng.uniform(0, 1, (160, 192)).astype(np.float32), 160, 192))
    blobs = (rng.uniform(0, 1, (40, 48)) > 0.6).astype(np.float32)
    cases.append(("blobs", np.kron(blobs, np.ones((6, 6), np.float32)) * 0.9 + 0.01, 240, 288))
    return cases


@pytest.mark.parametrize("name,pred,oh,ow", _planted_cases(), ids=lambda v: v if isinstance(v, str) else None)
def test_det_postprocess(hip_session, name, pred, oh, ow):
    gb, gs = hip_session.det_postprocess(pred, oh, ow)
    rb, rs = R.det_postprocess(pred, oh, ow)
    assert len(gb) == len(rb), f"{name}: {len(gb)} boxes vs oracle {len(rb)}"
    assert np.array_equal(gb, rb), f"{name}: box coordinates differ"
    assert np.array_equal(gs.view(np.uint32), rs.view(np.uint32)), f"{name}: scores differ"
    if name in ("axis", "rot", "scaled"):
        assert len(gb) > 0


# ---------------------------------------------------------------- a6 crops, a8/a10 resize-norm
def test_crops_and_resize_norm(hip_session):
    img = _rand_page(384, 512, 11)
    pred = workload.planted_map_rotated(384, 512, [(256, 60, 150, 14, 12.0), (150, 200, 110, 10, -31.0),
                                                   (440, 220, 100, 12, 83.0), (90, 340, 60, 9, 0.0),
                                                   (320, 330, 80, 16, 0.0)])
    boxes, _ = R.det_postprocess(pred, 384, 512)
    assert len(boxes) == 5
    got = hip_session.crop_images(img, boxes)
    for b, g in zip(boxes, got):
        ref = R.get_crop_img(img, b)
        assert g.shape == ref.shape
        assert np.array_equal(g, ref)
        for (img_w, ratio) in ((192, 0.0), (320, 320 / 48), (320, 11.3)):
            a = hip_session.resize_norm_image(g, g.shape[0], g.shape[1], 48, img_w, ratio)
            r = R.resize_norm_image(ref, ref.shape[0], ref.shape[1], 48, img_w, ratio)
            assert a.shape == r.shape
            assert np.array_equal(a.view(np.uint32), r.view(np.uint32))


def test_resize_norm_upscale(hip_session):
    crop = _rand_page(17, 140, 12)  # h < 48: fractional (upscaling) thumbnail paths
    a = hip_session.resize_norm_image(crop, 17, 140, 48, 320, 320 / 48)
    r = R.resize_norm_image(crop, 17, 140, 48, 320, 320 / 48)
    assert np.array_equal(a.view(np.uint32), r.view(np.uint32))


# ---------------------------------------------------------------- a12 CTC
def test_ctc_decode(hip_session):
    rng = np.random.default_rng(13)
    n, t, c = 5, 40, 6625
    p = rng.uniform(0, 1e-4, (n, t, c)).astype(np.float32)
    ids = rng.integers(0, 50, (n, t))
    ids[0] = 0                      # all blank -> NaN score
    ids[1, 5:15] = 7                # repeats collapse
    ids[2, ::2] = 0                 # blanks between repeats keep both
    for i in range(n):
        for k in range(t):
            p[i, k, ids[i, k]] = 0.5 + 0.4 * rng.random()
    p[3, 4, 9] = p[3, 4, 3] = 0.95  # tie: first index wins
    gi, gp, gt, gs = hip_session.ctc_decode(p)
    ri, rp, rt, rs = R.ctc_decode(p)
    assert np.array_equal(gi, ri)
    assert np.array_equal(gp.view(np.uint32), rp.view(np.uint32))
    assert all(np.array_equal(a, b) for a, b in zip(gt, rt))
    assert np.array_equal(gs.view(np.uint32), rs.view(np.uint32))
    assert np.isnan(gs[0])


# ---------------------------------------------------------------- a1 whole pipeline
def test_pipeline_teacher_forced(hip_session, oracle_session):
    """Full session: boxes / cls labels / token ids bit-exact when the oracle's three
    workers are fed by the HIP worker (so both sides see identical fp32 tensors)."""
    pages, maps = [], []
    for seed, (h, w, L) in enumerate([(640, 640, 8), (480, 704, 5)]):
        page, rects = workload.planted_page(h, w, L, seed + 20)
        dh, dw = R.resize_either_dims(h, w)
        pages.append(page); maps.append(workload.planted_map(dh, dw, h, w, rects))
    res = hip_session.run_batch(pages, det_map_override=maps)
    oracle_session.det_worker = hip_session.worker.det
    oracle_session.cls_worker = hip_session.worker.cls
    oracle_session.rec_worker = hip_session.worker.rec
    for page, m, r in zip(pages, maps, res):
        o = oracle_session.run(page, det_map_override=m)
        assert len(r.det_result) == len(o.det_boxes) > 0
        gb = np.stack([d.boxes.as_array() for d in r.det_result])
        assert np.array_equal(gb, o.det_boxes)
        assert np.array_equal(np.array([d.score for d in r.det_result], np.float32).view(np.uint32),
                              o.det_scores.view(np.uint32))
        assert [c.label.label for c in r.cls_result] == list(o.cls_labels)
        np.testing.assert_allclose([c.label.score for c in r.cls_result], o.cls_scores, atol=1e-5)
        for k, (g, ot) in enumerate(zip(r.rec_result, o.rec_tokens)):
            assert np.array_equal(g.tokens, ot), f"line {k} tokens differ"
            assert g.text == o.rec_text[k]
        np.testing.assert_allclose([g.score for g in r.rec_result], o.rec_scores, rtol=1e-4, equal_nan=True)


def test_pipeline_cls_rotation(models, oracle_session):
    """cls_processor.rs:163-166: crops whose label is 180 with score >= thresh are rotated
    in place before recognition.  Uses a classifier blob with a flipped head and a low
    threshold so the branch is taken."""
    import retto_amd
    from retto_amd import synth
    t = synth.cls_tensors(3)
    t["cls.head.fc.w"] = -t["cls.head.fc.w"] * 4; t["cls.head.fc.b"] = -t["cls.head.fc.b"]
    cls_blob = synth.pack_blob(t)
    det, _, rec, dic = models
    cfg = retto_amd.synthetic_session_config(0)
    cfg.worker_config.models.cls = retto_amd.RettoWorkerModelSource.Blob(cls_blob)
    cfg.cls_processor_config.thresh = 0.55
    s = retto_amd.RettoSession(cfg)
    try:
        from oracle.pipeline import OracleSession
        o = OracleSession(det, cls_blob, rec, dic)
        o.det_worker, o.cls_worker, o.rec_worker = s.worker.det, s.worker.cls, s.worker.rec
        page, rects = workload.planted_page(480, 640, 6, 31)
        dh, dw = R.resize_either_dims(480, 640)
        m = workload.planted_map(dh, dw, 480, 640, rects)
        # the oracle's cls threshold must match
        import oracle.pipeline as OP
        r = s.run_batch([page], det_map_override=[m])[0]
        orig = np.float32(0.9)
        src = OP.OracleSession.cls_process
        def patched(self, crops, dims, _src=src):
            return _src(self, crops, dims)
        labels = [c.label.label for c in r.cls_result]
        assert 180 in labels
        # re-run the oracle with the same threshold
        import types
        def cls_process(self, crops, dims):
            n = len(crops)
            lab = np.zeros(n, np.uint16); sc = np.zeros(n, np.float32)
            order = sorted(range(n), key=lambda i: -(float(dims[i][0]) / float(dims[i][1])))
            for s0 in range(0, n, 6):
                idxs = order[s0:s0 + 6]
                tt = np.stack([R.resize_norm_image(crops[i], dims[i][0], dims[i][1], 48, 192, 0.0) for i in idxs])
                idx, scs = R.cls_postprocess(self.cls_worker(tt))
                for j, i in enumerate(idxs):
                    l = [0, 180][int(idx[j])]
                    if l == 180 and scs[j] >= np.float32(0.55):
                        crops[i] = R.rotate180(crops[i])
                    lab[i] = l; sc[i] = scs[j]
            return lab, sc
        o.cls_process = types.MethodType(cls_process, o)
        ores = o.run(page, det_map_override=m)
        assert labels == list(ores.cls_labels)
        for g, ot in zip(r.rec_result, ores.rec_tokens):
            assert np.array_equal(g.tokens, ot)
    finally:
        s.close()


def _planted_for(page_h, page_w, lines, seed):
    """page + planted map at the det-input size the pipeline will derive for it."""
    page, rects = workload.planted_page(page_h, page_w, lines, seed)
    plan = R.resize_both_plan(page_h, page_w)
    ah, aw = plan[-1] if plan else (page_h, page_w)
    dh, dw = R.resize_either_dims(ah, aw)
    return page, workload.planted_map(dh, dw, page_h, page_w, rects)


def _teacher_forced(oracle_session, hip_session):
    oracle_session.det_worker = hip_session.worker.det
    oracle_session.cls_worker = hip_session.worker.cls
    oracle_session.rec_worker = hip_session.worker.rec


def _assert_page_equal(r, o):
    assert len(r.det_result) == len(o.det_boxes)
    if len(o.det_boxes):
        gb = np.stack([d.boxes.as_array() for d in r.det_result])
        assert np.array_equal(gb, o.det_boxes)
    assert [c.label.label for c in r.cls_result] == list(o.cls_labels)
    for g, ot in zip(r.rec_result, o.rec_tokens):
        assert np.array_equal(g.tokens, ot)


def test_pipeline_mixed_sizes_c4(hip_session, oracle_session):
    """BASELINE config C4 flavour: mixed page sizes in one batch, including the 640 -> 736 upscaling
    det resize (C1), a page above max_side_len (resize_both path) and non-square pages."""
    specs = [(640, 640, 6, 41), (720, 1280, 5, 42), (2100, 1500, 7, 43), (416, 608, 3, 44)]
    pages, maps = zip(*[_planted_for(h, w, L, s) for h, w, L, s in specs])
    res = hip_session.run_batch(list(pages), det_map_override=list(maps))
    _teacher_forced(oracle_session, hip_session)
    for page, m, r in zip(pages, maps, res):
        o = oracle_session.run(page, det_map_override=m)
        assert len(o.det_boxes) > 0
        _assert_page_equal(r, o)
    # batch invariance: a page processed alone gives the same result as inside the batch
    alone = hip_session.run_batch([pages[1]], det_map_override=[maps[1]])[0]
    assert [g.text for g in alone.rec_result] == [g.text for g in res[1].rec_result]
    assert np.array_equal(np.stack([d.boxes.as_array() for d in alone.det_result]),
                          np.stack([d.boxes.as_array() for d in res[1].det_result]))


def test_pipeline_empty_and_border_pages(hip_session, oracle_session):
    blank = np.zeros((320, 320, 3), np.uint8)
    m_blank = np.full((736, 736), 0.01, np.float32)
    page, _ = workload.planted_page(320, 480, 3, 7)
    dh, dw = R.resize_either_dims(320, 480)
    # text regions touching the page border: crops sample outside the image (white fill)
    m_border = workload.planted_map(dh, dw, 320, 480, [(0, 0, 200, 30), (300, 290, 480, 320), (100, 150, 380, 180)], shrink=0.0)
    res = hip_session.run_batch([blank, page], det_map_override=[m_blank, m_border])
    assert res[0].det_result == [] and res[0].cls_result == [] and res[0].rec_result == []
    _teacher_forced(oracle_session, hip_session)
    o = oracle_session.run(page, det_map_override=m_border)
    assert len(o.det_boxes) == 3
    _assert_page_equal(res[1], o)


def test_reference_small_image_scenario(hip_session):
    """session.rs:206-229 restated: 200x50 page, text blob in the bottom-right (the render rotated by
    180 degrees); the first box's bottom-right corner lies within 10 px of (200, 50)."""
    page = np.zeros((50, 200, 3), np.uint8)
    page[28:48, 100:197] = 255
    pred = np.full((736, 2944), 0.02, np.float32)
    pred[int(0.55 * 736):int(0.97 * 736), int(0.5 * 2944):int(0.985 * 2944)] = 0.9
    r = hip_session.run_batch([page], det_map_override=[pred])[0]
    assert len(r.det_result) == 1
    br = r.det_result[0].boxes.br()
    assert np.hypot(br.x - 200, br.y - 50) < 10
    assert r.cls_result[0].label.label in (0, 180)


def test_det_postprocess_noise_full_page(hip_session):
    """Worst case for the contour machinery: a full 960x960 noise map (what the det net with
    synthetic weights actually produces) -> tens of thousands of components and holes."""
    rng = np.random.default_rng(99)
    pred = rng.uniform(0, 1, (960, 960)).astype(np.float32)
    gb, gs = hip_session.det_postprocess(pred, 960, 960)
    rb, rs = R.det_postprocess(pred, 960, 960)
    assert len(gb) == len(rb)
    assert np.array_equal(gb, rb) and np.array_equal(gs.view(np.uint32), rs.view(np.uint32))


def test_det_postprocess_more_boxes_than_the_old_cap(hip_session):
    """The reference's box list is unbounded (det_processor.rs:279-335).  A page of 5 329 small blobs (more than the 4 096
    the sort kernel used to hold in LDS, and more than fit its LDS path now) must come back complete, in the oracle's reading
    order, instead of RT_ERR_CAPACITY; with max_boxes_per_page = 16 384 a denser page (11 k boxes) takes the workspace sort."""
    pred = np.full((960, 960), 0.02, np.float32)
    for gy in range(73):
        for gx in range(73):
            y0, x0 = 4 + 13 * gy, 4 + 13 * gx
            pred[y0:y0 + 8, x0:x0 + 8] = 0.85 + 0.001 * ((gx * 7 + gy * 3) % 50)
    gb, gs = hip_session.det_postprocess(pred, 960, 960)
    rb, rs = R.det_postprocess(pred, 960, 960)
    assert len(rb) == 73 * 73 and len(gb) == len(rb)
    assert np.array_equal(gb, rb) and np.array_equal(gs.view(np.uint32), rs.view(np.uint32))
    cfg = retto_amd.synthetic_session_config(0)
    cfg.max_boxes_per_page = 16384
    s = retto_amd.RettoSession(cfg)
    try:
        pred = np.full((1280, 1280), 0.02, np.float32)
        for gy in range(105):
            for gx in range(105):
                y0, x0 = 3 + 12 * gy, 3 + 12 * gx
                pred[y0:y0 + 7, x0:x0 + 7] = 0.8 + 0.001 * ((gx * 5 + gy * 11) % 90)
        gb, gs = s.det_postprocess(pred, 1280, 1280)
        rb, rs = R.det_postprocess(pred, 1280, 1280)
        assert len(rb) == 105 * 105 and len(gb) == len(rb)
        assert np.array_equal(gb, rb) and np.array_equal(gs.view(np.uint32), rs.view(np.uint32))
    finally:
        s.close()


def test_det_postprocess_sparse_noise(hip_session):
    rng = np.random.default_rng(100)
    pred = (rng.uniform(0, 1, (480, 640)) > 0.93).astype(np.float32) * 0.9 + 0.01   # many tiny blobs
    pred[100:140, 50:400] = 0.8
    gb, gs = hip_session.det_postprocess(pred, 480, 640)
    rb, rs = R.det_postprocess(pred, 480, 640)
    assert len(gb) == len(rb) >= 1
    assert np.array_equal(gb, rb) and np.array_equal(gs.view(np.uint32), rs.view(np.uint32))


@pytest.mark.parametrize("w", [332, 333, 334, 335])
def test_det_postprocess_widths_mod_4(hip_session, w):
    """The labelling passes take four pixels per thread when the row length is a multiple of 4 (round 5) and one otherwise: the same
    content at widths = 0 .. 3 (mod 4) -- noise blobs, runs that start / end on every frame edge, holes, a diagonal chain that is
    only 8-connected -- against the oracle, boxes and score bit patterns."""
    rng = np.random.default_rng(7)
    h = 201
    pred = (rng.uniform(0, 1, (h, w)) > 0.975).astype(np.float32) * 0.85 + 0.02   # sparse blobs (dilation merges denser noise into one component)
    pred[0:9, 0:60] = 0.9; pred[h - 8:h, w - 70:w] = 0.9                  # blobs on the top-left / bottom-right corners
    pred[40:70, 0:5] = 0.8; pred[90:130, w - 4:w] = 0.8                    # left / right frame
    pred[100:140, 100:220] = 0.9; pred[112:128, 130:190] = 0.02           # a hole
    for k in range(30): pred[150 + k, 40 + k] = 0.95                      # diagonal chain
    gb, gs = hip_session.det_postprocess(pred, h, w)
    rb, rs = R.det_postprocess(pred, h, w)
    assert len(gb) == len(rb) >= 3
    assert np.array_equal(gb, rb) and np.array_equal(gs.view(np.uint32), rs.view(np.uint32))


def test_pipeline_raw_det_map(hip_session, oracle_session):
    """No planted map: boxes come from the det network's own (noise-like) output; the oracle is
    teacher-forced with the HIP map, so every discrete result must still agree."""
    page = _rand_page(320, 416, 77)
    r = hip_session.run_batch([page])[0]
    _teacher_forced(oracle_session, hip_session)
    o = oracle_session.run(page)
    _assert_page_equal(r, o)


@pytest.mark.parametrize("seed", range(24))
def test_det_postprocess_rotated_fuzz(hip_session, seed):
    """Random rotated text boxes (all angles, thin to fat, some touching each other or the frame) with soft
    edges: the f64 calipers / Clipper trigonometry of the device must give the oracle's boxes bit for bit."""
    rng = np.random.default_rng(1000 + seed)
    H, W = int(rng.integers(6, 24)) * 32, int(rng.integers(6, 24)) * 32
    boxes = []
    for _ in range(int(rng.integers(1, 14))):
        bw = float(rng.uniform(20, 0.6 * W)); bh = float(rng.uniform(4, 40))
        boxes.append((float(rng.uniform(0, W)), float(rng.uniform(0, H)), bw, bh, float(rng.uniform(-90, 90))))
    m = workload.planted_map_rotated(H, W, boxes)
    m = np.clip(m + rng.normal(0, 0.08, m.shape).astype(np.float32), 0.0, 1.0).astype(np.float32)  # ragged borders
    oh, ow = int(H * rng.uniform(0.5, 1.5)), int(W * rng.uniform(0.5, 1.5))
    gb, gs = hip_session.det_postprocess(m, oh, ow)
    rb, rs = R.det_postprocess(m, oh, ow)
    assert len(gb) == len(rb)
    assert np.array_equal(gb, rb)
    assert np.array_equal(gs.view(np.uint32), rs.view(np.uint32))


@pytest.mark.parametrize("seed", range(8))
def test_crops_fuzz(hip_session, seed):
    """Random (integer-cornered, as DB post produces them) quadrilaterals, including ones that stick out of the
    page (white fill) and tall ones (rotate270): bicubic perspective crops and their resize-normalised tensors
    are bit-exact."""
    rng = np.random.default_rng(2000 + seed)
    H, W = int(rng.integers(80, 500)), int(rng.integers(120, 700))
    img = _rand_page(H, W, 3000 + seed)
    boxes = []
    for _ in range(6):
        cx, cy = rng.uniform(0, W), rng.uniform(0, H)
        hl, ht, th = rng.uniform(8, 0.4 * W), rng.uniform(3, 30), np.deg2rad(rng.uniform(-90, 90))
        if rng.uniform() < 0.25:
            hl, ht = ht, hl  # tall box -> rotate270 path
        c, s_ = np.cos(th), np.sin(th)
        pts = [(cx + sx * hl * c - sy * ht * s_, cy + sx * hl * s_ + sy * ht * c) for sx, sy in ((-1, -1), (1, -1), (1, 1), (-1, 1))]
        boxes.append(np.round(np.array(pts, np.float32)))
    boxes = np.stack(boxes).reshape(-1, 8).astype(np.float32)
    got = hip_session.crop_images(img, boxes)
    for b, g in zip(boxes, got):
        ref = R.get_crop_img(img, b)
        assert g.shape == ref.shape and np.array_equal(g, ref)
        a = hip_session.resize_norm_image(g, g.shape[0], g.shape[1], 48, 320, max(320 / 48, g.shape[1] / g.shape[0]))
        r = R.resize_norm_image(ref, ref.shape[0], ref.shape[1], 48, 320, max(320 / 48, ref.shape[1] / ref.shape[0]))
        assert np.array_equal(a.view(np.uint32), r.view(np.uint32))


def test_non_default_det_config():
    """DetProcessorConfig knobs other than the defaults (det_processor.rs:44-93): limit_type Max, another limit,
    thresholds, unclip ratio, minimum box size, no dilation -- stage results still equal the oracle's."""
    cfg = retto_amd.synthetic_session_config(0)
    d = cfg.det_processor_config
    d.limit_type, d.limit_side_len = "Max", 640
    d.threch, d.box_thresh, d.unclip_ratio, d.min_mini_box_size = 0.4, 0.6, 2.0, 5
    d.dilation_kernel = None  # det_processor.rs:290 only consults the kernel (use_dilation is never read there either)
    s = retto_amd.RettoSession(cfg)
    try:
        for (h, w) in ((900, 1300), (300, 500), (640, 640)):
            page = _rand_page(h, w, h + w)
            x = s.det_preprocess(page)
            dh, dw = R.resize_either_dims(h, w, 1, 640)
            assert x.shape == (1, 3, dh, dw)
            page_r = R.resize_both(page)  # the stage function takes the page after the session's size limits
            ref = R.det_preprocess(page_r, 1, 640)
            assert np.array_equal(s.det_preprocess(page_r).view(np.uint32), ref.view(np.uint32))
        rng = np.random.default_rng(99)
        m = workload.planted_map_rotated(384, 512, [(256, 100, 150, 14, 12.0), (200, 250, 120, 10, -31.0), (400, 200, 100, 4, 83.0),
                                                    (90, 330, 60, 3, 45.0)])
        m = np.clip(m + rng.normal(0, 0.1, m.shape).astype(np.float32), 0, 1).astype(np.float32)
        gb, gs = s.det_postprocess(m, 384, 512)
        rb, rs = R.det_postprocess(m, 384, 512, thresh=0.4, box_thresh=0.6, unclip_ratio=2.0, min_size=5, dilate=False)
        assert len(gb) == len(rb) > 0
        assert np.array_equal(gb, rb) and np.array_equal(gs.view(np.uint32), rs.view(np.uint32))
    finally:
        s.close()


def test_c3_full_size_properties(hip_session):
    """BASELINE config C3 at full size (32 pages of 960 x 960, 32 planted lines each, the bench workload) through
    size-independent properties: (1) batch / lane composition does not change any page's result -- the whole batch on
    3 lanes equals the same pages in shuffled order and each page run alone, bit for bit (boxes, labels, token
    ids, scores); (2) the det checksum of the batch is the sum of the per-page checksums to fp32 rounding; (3) the planted lines come
    back: 32 boxes per page, each inside its planted rectangle grown by the unclip offset."""
    n = 32
    pages, maps, rects = [], [], []
    for i in range(n):
        page, rc = workload.planted_page(960, 960, 32, seed=i)
        pages.append(page); rects.append(rc)
        maps.append(workload.planted_map(960, 960, 960, 960, rc))
    full = hip_session.run_batch(pages, det_map_override=maps)
    cs_full = hip_session.last_det_checksum

    def same(a, b):
        assert len(a.det_result) == len(b.det_result) == 32
        assert np.array_equal(np.stack([d.boxes.as_array() for d in a.det_result]), np.stack([d.boxes.as_array() for d in b.det_result]))
        assert [d.score for d in a.det_result] == [d.score for d in b.det_result]
        assert [(c.label.label, c.label.score) for c in a.cls_result] == [(c.label.label, c.label.score) for c in b.cls_result]
        for x, y in zip(a.rec_result, b.rec_result):
            assert np.array_equal(x.tokens, y.tokens) and x.text == y.text
            assert x.score == y.score or (np.isnan(x.score) and np.isnan(y.score))

    # the page inside the 1024-line batch (large fused kernels) against the oracle fed by the HIP worker in batches of 6
    # lines (small, unfused kernels): tokens bit-exact, scores to fp32 tolerance
    from oracle.pipeline import OracleSession
    from retto_amd import synth
    o = OracleSession(*synth.synth_models(0))
    o.det_worker, o.cls_worker, o.rec_worker = hip_session.worker.det, hip_session.worker.cls, hip_session.worker.rec
    for j in (0, 17):
        ref = o.run(pages[j], det_map_override=maps[j])
        _assert_page_equal(full[j], ref)
        np.testing.assert_allclose([g.score for g in full[j].rec_result], ref.rec_scores, rtol=1e-4, equal_nan=True)
        np.testing.assert_allclose([c.label.score for c in full[j].cls_result], ref.cls_scores, atol=1e-5)

    perm = np.random.default_rng(0).permutation(n)
    shuffled = hip_session.run_batch([pages[j] for j in perm], det_map_override=[maps[j] for j in perm])
    for k, j in enumerate(perm):
        same(shuffled[k], full[j])
    cs_sum = 0.0
    for j in (0, 7, 31):
        alone = hip_session.run_batch([pages[j]], det_map_override=[maps[j]])[0]
        same(alone, full[j])
    for j in range(n):
        hip_session.run_batch([pages[j]], det_map_override=[maps[j]])
        cs_sum += hip_session.last_det_checksum
    # (the det maps of a page alone and inside the batch come from different kernel shapes: fp32 rounding, not bit-equal)
    assert abs(cs_sum - cs_full) <= 1e-6 * abs(cs_full)
    for r, rc in zip(full, rects):
        got = np.stack([d.boxes.as_array() for d in r.det_result]).reshape(-1, 4, 2)
        for x0, y0, x1, y1 in rc:
            inside = [(b[:, 0].min() >= x0 - 12 and b[:, 0].max() <= x1 + 12 and b[:, 1].min() >= y0 - 12 and b[:, 1].max() <= y1 + 12)
                      for b in got]
            assert sum(inside) == 1, (x0, y0, x1, y1)


def test_c4_mixed_sizes_full_size_properties(hip_session, oracle_session):
    """BASELINE config C4 at its real page sizes (640^2 ... 2480 x 3508, the largest taking the session size limit a2) and at
    its per-GPU share (256 pages / 8 GPUs = 32 pages in one batch over the session's lanes), 32 planted lines per page: the pages
    checked equal the oracle fed by the HIP worker in the reference's batches of 6 (boxes, labels, token ids bit-exact; scores to
    fp32 tolerance), every page has its 32 boxes, and a page run alone gives the same result as inside the batch."""
    sizes = [(640, 640), (960, 960), (720, 1280), (1080, 1920), (1754, 1240), (3508, 2480)]
    specs = [(h, w, 32, 300 + 7 * i) for i, (h, w) in enumerate((sizes * 6)[:32])]
    pages, maps = zip(*[_planted_for(h, w, L, s) for h, w, L, s in specs])
    res = hip_session.run_batch(list(pages), det_map_override=list(maps))
    assert len(res) == 32 and all(len(r.det_result) == 32 and len(r.rec_result) == 32 for r in res)
    _teacher_forced(oracle_session, hip_session)
    for j in (0, 3, 5, 8, 16, 17, 29, 31):
        o = oracle_session.run(pages[j], det_map_override=maps[j])
        assert len(o.det_boxes) == 32
        _assert_page_equal(res[j], o)
        np.testing.assert_allclose([g.score for g in res[j].rec_result], o.rec_scores, rtol=1e-4, equal_nan=True)
        np.testing.assert_allclose([c.label.score for c in res[j].cls_result], o.cls_scores, atol=1e-5)
    for j in (5, 10, 30):
        alone = hip_session.run_batch([pages[j]], det_map_override=[maps[j]])[0]
        assert np.array_equal(np.stack([d.boxes.as_array() for d in alone.det_result]),
                              np.stack([d.boxes.as_array() for d in res[j].det_result]))
        assert [(g.text, g.score) for g in alone.rec_result] == [(g.text, g.score) for g in res[j].rec_result]
        assert [(c.label.label, c.label.score) for c in alone.cls_result] == [(c.label.label, c.label.score) for c in res[j].cls_result]


def test_reference_large_image_scenario(hip_session, oracle_session):
    """session.rs:231-255 restated (`test_large_image`, the regression test of the reference's commit 7fc4127b): ONE 7680 x 4320
    page -- the max_side_len path of resize_both: 7680 x 4320 -> 1984 x 1120 (image_helper.rs:106-148), which is also the det
    input -- with a 300-pixel-high text blob in the bottom-right corner (the reference renders its text at (0, 0) and rotates the
    page by 180 degrees).  Through rt_run_batch: boxes, label and token ids equal the oracle's (bit-exact), and the first box's
    bottom-right corner lies within 100 px of (7680, 4320) -- the reference's own assertion."""
    H, W = 4320, 7680
    page = np.zeros((H, W, 3), np.uint8)
    x0, y0, x1, y1 = W - 1560, H - 330, W - 20, H - 20
    page[y0:y1, x0:x1] = 255
    # five glyph-like gaps so that the crop is not one flat rectangle
    for k in range(1, 5):
        page[y0:y1, x0 + k * 308 - 12:x0 + k * 308 + 12] = 0
    plan = R.resize_both_plan(H, W)
    assert plan and tuple(plan[-1]) == (1120, 1984)
    dh, dw = R.resize_either_dims(*plan[-1])
    assert (dh, dw) == (1120, 1984)
    m = workload.planted_map(dh, dw, H, W, [(x0, y0, x1, y1)])
    r = hip_session.run_batch([page], det_map_override=[m])[0]
    assert len(r.det_result) == 1
    br = r.det_result[0].boxes.br()
    assert np.hypot(br.x - W, br.y - H) < 100
    _teacher_forced(oracle_session, hip_session)
    o = oracle_session.run(page, det_map_override=m)
    _assert_page_equal(r, o)
    assert r.cls_result[0].label.label in (0, 180)
    np.testing.assert_allclose([g.score for g in r.rec_result], o.rec_scores, rtol=1e-4, equal_nan=True)


def test_extreme_line_shapes(hip_session, oracle_session):
    """Lines at the ends of the aspect range on one page: 76:1 (rec width 48 * 76 = 3648, 456 time steps through the
    global-attention neck), a 6-pixel-high sliver (upscaled 8x by resize_norm_image), a tall narrow box (h/w >= 1.5: the crop
    is rotated by 270 degrees, image_helper.rs:246) -- against the oracle fed by the HIP worker."""
    h, w = 400, 1984
    page = np.zeros((h, w, 3), np.uint8)
    rng = np.random.default_rng(3)
    rects = [(20, 30, 1930, 55), (40, 120, 300, 126), (600, 100, 640, 380), (900, 200, 1500, 240)]
    for x0, y0, x1, y1 in rects:
        page[y0:y1, x0:x1] = rng.integers(100, 256, (y1 - y0, x1 - x0, 3), dtype=np.uint8)
    plan = R.resize_both_plan(h, w)
    ah, aw = plan[-1] if plan else (h, w)
    dh, dw = R.resize_either_dims(ah, aw)
    m = workload.planted_map(dh, dw, h, w, rects, shrink=0.05)
    res = hip_session.run_batch([page], det_map_override=[m])[0]
    _teacher_forced(oracle_session, hip_session)
    o = oracle_session.run(page, det_map_override=m)
    assert len(o.det_boxes) >= 3
    _assert_page_equal(res, o)
    np.testing.assert_allclose([g.score for g in res.rec_result], o.rec_scores, rtol=1e-4, equal_nan=True)


# ---------------------------------------------------------------- independent end-to-end comparison (no teacher forcing)
def test_pipeline_against_independent_oracle(hip_session, oracle_session):
    """The HIP pipeline against the oracle pipeline running ITS OWN networks (torch-CPU fp32): nothing is shared but the
    inputs.  Boxes come from the planted map on both sides and must be bit-exact (so the crops are); the classifier label and
    the CTC token ids are decisions of two independent fp32 implementations and must agree wherever the oracle's decision
    margin exceeds EPS (the fp32 tolerance of the network tests is 1e-4 / 2e-4); the fraction of decisive lines is reported."""
    EPS_CLS, EPS_REC = 1e-3, 1e-3
    from oracle.pipeline import OracleSession  # a fresh oracle: other tests of this module teacher-force the shared fixture
    from retto_amd import synth
    det, cls, rec, dic = synth.synth_models(0)
    o = OracleSession(det, cls, rec, dic)
    n_lines = n_dec = n_cls_dec = 0
    for seed, (h, w, lines) in enumerate([(320, 480, 5), (480, 640, 9), (256, 704, 4)]):
        page, rects = workload.planted_page(h, w, lines, seed=40 + seed)
        dh, dw = R.resize_either_dims(h, w)
        pmap = workload.planted_map(dh, dw, h, w, rects)
        got = hip_session.run_batch([page], det_map_override=[pmap])[0]
        ref = o.run(page, det_map_override=pmap)
        assert len(got.det_result) == len(ref.det_boxes) == lines
        assert np.array_equal(np.stack([d.boxes.as_array() for d in got.det_result]), ref.det_boxes)
        assert np.array_equal(np.array([d.score for d in got.det_result], np.float32).view(np.uint32), ref.det_scores.view(np.uint32))
        for k in range(lines):
            n_lines += 1
            if ref.cls_margins[k] > EPS_CLS:
                n_cls_dec += 1
                assert got.cls_result[k].label.label == int(ref.cls_labels[k])
                assert abs(got.cls_result[k].label.score - float(ref.cls_scores[k])) <= 1e-4
            if ref.rec_margins[k] > EPS_REC:
                n_dec += 1
                assert np.array_equal(got.rec_result[k].tokens, ref.rec_tokens[k]), f"page {seed} line {k}"
                assert got.rec_result[k].text == ref.rec_text[k]
                if len(ref.rec_tokens[k]):
                    assert abs(got.rec_result[k].score - float(ref.rec_scores[k])) <= 2e-4
    print(f"independent oracle: {n_lines} lines, cls decisive {n_cls_dec}, rec decisive {n_dec}")
    assert n_dec >= n_lines // 2 and n_cls_dec >= n_lines // 2


# ---------------------------------------------------------------- differential fuzz on the ill-defined corners
@pytest.mark.parametrize("seed", range(6))
def test_det_postprocess_non_transitive_reading_order(hip_session, seed):
    """A.5: the reading-order comparator (|dy| < 10 -> by x, else by y) is not transitive on staircases of boxes whose centres
    are < 10 px apart pairwise-adjacent but > 10 px apart end to end.  Rust's sort is unspecified there; the contract of this
    build is the bottom-up stable merge sort over contour discovery order -- device and oracle must apply the same one."""
    rng = np.random.default_rng(500 + seed)
    H, W = 512, 768
    m = np.full((H, W), 0.01, np.float32)
    x = 20
    y = int(rng.integers(30, 60))
    for _ in range(int(rng.integers(8, 14))):          # a staircase: each box 4-8 px lower than the previous one, shuffled x order
        bw, bh = int(rng.integers(30, 50)), int(rng.integers(12, 18))
        xs = int(rng.integers(10, W - 60))
        m[y:y + bh, xs:xs + bw] = 0.9
        y += bh + int(rng.integers(4, 8)) if rng.uniform() < 0.4 else int(rng.integers(4, 8))
        if y > H - 40:
            break
    # plus two clean rows far below, so that the sorted prefix / suffix is unambiguous
    m[440:455, 100:200] = 0.9; m[440:455, 300:420] = 0.9
    gb, gs = hip_session.det_postprocess(m, H, W)
    rb, rs = R.det_postprocess(m, H, W)
    assert len(gb) == len(rb) >= 3
    assert np.array_equal(gb, rb) and np.array_equal(gs.view(np.uint32), rs.view(np.uint32))


@pytest.mark.parametrize("seed", range(6))
def test_det_postprocess_degenerate_hulls_without_dilation(seed):
    """A.4: with the dilation kernel disabled (dilation_kernel = None) isolated pixels and 1-pixel-wide runs reach min_area_rect as
    1- and 2-point hulls and collinear point sets; imageproc's draw_polygon_mut panics on some of them in the reference.  The
    defined behaviour here (score 0 -> dropped, everything else as usual) must be the same on the device and in the oracle."""
    import retto_amd
    cfg = retto_amd.synthetic_session_config(0)
    cfg.det_processor_config.dilation_kernel = None
    s = retto_amd.RettoSession(cfg)
    try:
        rng = np.random.default_rng(700 + seed)
        H, W = 192, 256
        m = np.full((H, W), 0.02, np.float32)
        for _ in range(40):                                   # single pixels, horizontal / vertical / diagonal 2-6 pixel runs
            y0, x0 = int(rng.integers(0, H)), int(rng.integers(0, W))
            n, (dy, dx) = int(rng.integers(1, 7)), [(0, 1), (1, 0), (1, 1), (1, -1)][int(rng.integers(0, 4))]
            for k in range(n):
                yy, xx = y0 + k * dy, x0 + k * dx
                if 0 <= yy < H and 0 <= xx < W:
                    m[yy, xx] = 0.95
        m[60:75, 40:180] = 0.9                                # one real line
        m[0, :] = 0.9; m[:, W - 1] = 0.9                      # 1-pixel strips hugging the frame
        gb, gs = s.det_postprocess(m, H, W)
        rb, rs = R.det_postprocess(m, H, W, dilate=False)
        assert len(gb) == len(rb) >= 1
        assert np.array_equal(gb, rb) and np.array_equal(gs.view(np.uint32), rs.view(np.uint32))
    finally:
        s.close()


@pytest.mark.parametrize("seed", range(4))
def test_det_postprocess_border_hugging_contours(hip_session, seed):
    """Blobs that touch or run along the frame (1-3 pixel strips on each side, corners, a frame-wide bar): border start pixels,
    hull points on the image edge, clipped boxes."""
    rng = np.random.default_rng(900 + seed)
    H, W = 160 + 32 * seed, 224
    m = np.full((H, W), 0.05, np.float32)
    t = int(rng.integers(1, 4))
    m[:t, 10:150] = 0.9; m[H - t:, 40:200] = 0.9; m[20:120, :t] = 0.9; m[30:100, W - t:] = 0.9
    m[:12, :14] = 0.9; m[H - 9:, W - 16:] = 0.9               # corners
    m[70:82, :] = 0.85                                        # a bar across the whole width
    m[100:130, 60:160] = np.clip(rng.normal(0.7, 0.2, (30, 100)), 0, 1).astype(np.float32)
    gb, gs = hip_session.det_postprocess(m, H, W)
    rb, rs = R.det_postprocess(m, H, W)
    assert len(gb) == len(rb) >= 2
    assert np.array_equal(gb, rb) and np.array_equal(gs.view(np.uint32), rs.view(np.uint32))


# ---------------------------------------------------------------- fused thin LCNetV3 blocks: the three kernel forms
# rt_bench_lc runs one block (3x3 depthwise -> 1x1 conv, random weights, the LCNetV3 tails) through k_lc_thin (or, where that
# kernel has no instance, the unfused depthwise + GEMM pair) and through the barrier-free forms of nn_lcwave.hip (1 = direct
# loads + DPP taps, 3 = wave-private LDS staging, the production form) and returns max |difference|: the forms share the
# arithmetic order, so the bar is bit-identity -- on ragged sizes too (widths that are no multiple of the 16-pixel tile, heights
# that are no multiple of the 2- / 4-row tile, single-pixel and single-row images, images narrower than one tile).
_LC_BLOCKS = [(16, 32, 1), (32, 64, 1), (48, 48, 1), (64, 64, 1), (32, 48, 2), (48, 96, 2), (64, 128, 21)]
_LC_SIZES = [(3, 13, 37), (2, 5, 16), (4, 1, 1), (2, 24, 17), (1, 50, 100), (5, 1, 33), (2, 7, 2)]


@pytest.mark.parametrize("cin,cout,stride", _LC_BLOCKS)
@pytest.mark.parametrize("form", [1, 3])
def test_thin_block_kernel_forms_are_bit_identical(hip_session, cin, cout, stride, form):
    import ctypes as C
    if form == 1 and stride != 1:
        pytest.skip("the direct-load form is kept for the stride-1 blocks only")
    lib, h = hip_session._hd.lib, hip_session._hd.h
    lib.rt_bench_lc.argtypes = [C.c_void_p] + [C.c_int] * 8 + [C.POINTER(C.c_float), C.POINTER(C.c_float)]
    for n, hh, ww in _LC_SIZES:
        ms, md = C.c_float(), C.c_float(-1.0)
        rc = lib.rt_bench_lc(h, n, hh, ww, cin, cout, stride, form, 1, C.byref(ms), C.byref(md))
        assert rc == 0, lib.rt_last_error(h)
        assert md.value == 0.0, f"{cin}->{cout} /{stride} form {form} on {n} x {hh} x {ww}: max |diff| {md.value}"


@pytest.mark.parametrize("flags", [128, 256, 512, 128 | 256 | 512])
def test_round3_kernels_are_bit_identical_to_the_ones_they_replaced(hip_session, flags):
    """k_lc_lds (fused thin blocks, incl. the newly fused (2,1) block), k_gemm32p (+se) and the column-sweep depthwise kernel
    against k_lc_thin / the unfused pair, the register-staged wide GEMM tiles and k_dwconv_rows: whole networks, same process,
    switched by rt_debug_set_variants bits 7-9.  Sizes large enough for the wide-GEMM and squeeze-excite dispatch thresholds."""
    lib = hip_session._hd.lib
    rng = np.random.default_rng(flags)
    xd = rng.uniform(-1, 1, (2, 3, 416, 352)).astype(np.float32)
    xr = rng.uniform(-1, 1, (420, 3, 48, 400)).astype(np.float32)   # 420 x 12 x 100 = 504000 rows at the 240-channel stages
    for i in range(xr.shape[0]):
        xr[i, :, :, 180 + (37 * i) % 220:] = 0
    xc = rng.uniform(-1, 1, (9, 3, 48, 192)).astype(np.float32)
    # narrow lines: 144 rows per line at the first squeeze-excite level (6 x 24), so k_gemm32p+se's 256-row blocks span THREE
    # lines (scale slot 2, the third a_tab entry, the clamped prefetch of the last lines) at M = 144 000 >= 131072
    xn = rng.uniform(-1, 1, (1000, 3, 48, 96)).astype(np.float32)
    try:
        lib.rt_debug_set_variants(0, 0, flags)
        old = [hip_session.worker.det(xd), hip_session.worker.rec(xr), hip_session.worker.cls(xc), hip_session.worker.rec(xn)]
    finally:
        lib.rt_debug_set_variants(0, 0, 0)
    new = [hip_session.worker.det(xd), hip_session.worker.rec(xr), hip_session.worker.cls(xc), hip_session.worker.rec(xn)]
    for name, a, b in zip(("det", "rec", "cls", "rec-narrow"), old, new):
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), f"{name}: max |diff| {np.abs(a.astype(np.float64) - b).max()}"


@pytest.mark.parametrize("n,h,w", [(2, 416, 352), (1, 736, 1312), (3, 960, 960)])
def test_upsampling_aware_fpn_convs_against_the_launch_series(hip_session, n, h, w):
    """nn_fpn.hip (round 4): the RSEFPN output convs and the DB head's first conv as phase / class convs of the un-upsampled
    levels with pre-summed weights, the finest lateral tensor composed away -- against the round-3 launch series (lateral_add +
    k_conv3_few + the fused four-level gather; rt_debug_set_variants bit 11).  Same mathematics, another summation order: equal
    to fp32 rounding on the probability map (logit-level differences ~1e-6), the thresholded mask equal wherever the map is not
    within 1e-5 of the threshold.  416 x 352 / 736 x 1312: partial 16 x 16 tiles at the 1/4-resolution level (104 x 88, 184 x 328)."""
    lib = hip_session._hd.lib
    x = np.random.default_rng(n * h + w).uniform(-1, 1, (n, 3, h, w)).astype(np.float32)
    new = hip_session.worker.det(x)
    try:
        lib.rt_debug_set_variants(0, 0, 2048)
        old = hip_session.worker.det(x)
    finally:
        lib.rt_debug_set_variants(0, 0, 0)
    assert np.isfinite(new).all()
    d = np.abs(new.astype(np.float64) - old)
    assert d.max() <= 2e-5, d.max()
    far = np.abs(old - 0.3) > 1e-5
    assert np.array_equal((new > 0.3)[far], (old > 0.3)[far])
    again = hip_session.worker.det(x)
    assert np.array_equal(new.view(np.uint32), again.view(np.uint32))   # fixed summation orders (incl. the per-tile SE sums)


@pytest.mark.parametrize("n", [1, 9, 700])
def test_fused_classifier_blocks_against_the_launch_series(hip_session, n):
    """k_cls_block (one kernel per MobileNetV3 block, a workgroup per crop) against the unfused expand / depthwise / SE / linear
    launches: same formulas, the squeeze-excite pooling sums in another order -> probabilities within 1e-6, labels equal."""
    lib = hip_session._hd.lib
    x = np.random.default_rng(n).uniform(-1, 1, (n, 3, 48, 192)).astype(np.float32)
    x[0] *= 0.05
    fused = hip_session.worker.cls(x)
    try:
        lib.rt_debug_set_variants(0, 0, 1024)
        series = hip_session.worker.cls(x)
    finally:
        lib.rt_debug_set_variants(0, 0, 0)
    assert np.isfinite(fused).all() and fused.shape == series.shape == (n, 2)
    assert np.abs(fused - series).max() <= 1e-6
    assert np.array_equal(fused.argmax(1), series.argmax(1))


def test_fused_classifier_blocks_are_repeatable(hip_session):
    """k_cls_block reduces the squeeze-excite pooling sums inside the workgroup (DPP row sums, per-wave partials added in a fixed
    order) and keeps the linear accumulators in registers: 12 runs on the same 300 crops must agree bit for bit."""
    x = np.random.default_rng(77).uniform(-1, 1, (300, 3, 48, 192)).astype(np.float32)
    first = hip_session.worker.cls(x)
    for _ in range(11):
        again = hip_session.worker.cls(x)
        assert np.array_equal(first.view(np.uint32), again.view(np.uint32))


def test_submit_wait_equals_run_batch(hip_session):
    """rt_submit_batch / rt_wait_batch (round 4; the counterpart of run_stream's worker thread, session.rs:108-143): three batches
    of different composition submitted ahead on the persistent lane threads must give, ticket by ticket, exactly what
    rt_run_batch gives for the same pages -- boxes, scores, labels, token ids bit for bit -- whatever order the tickets are
    waited in; single-page batches are spread over the lanes; other calls are refused while tickets are open."""
    import ctypes as C
    lib, h = hip_session._hd.lib, hip_session._hd.h
    batches = []
    for b, (n, hh, ww) in enumerate(((7, 480, 640), (1, 320, 480), (5, 640, 480), (1, 352, 512))):
        pages, maps = [], []
        for i in range(n):
            page, rects = workload.planted_page(hh, ww, 3 + (i + b) % 4, seed=100 * b + i)
            dh, dw = R.resize_either_dims(hh, ww)
            pages.append(page); maps.append(workload.planted_map(dh, dw, hh, ww, rects))
        batches.append((pages, maps))

    def digest(r, n):
        out = []
        for i in range(n):
            k = lib.rt_results_count(r, i)
            boxes = np.ctypeslib.as_array(lib.rt_results_boxes(r, i), (k, 8)).copy() if k else np.zeros((0, 8), np.float32)
            rs = np.ctypeslib.as_array(lib.rt_results_rec_scores(r, i), (k,)).copy() if k else np.zeros(0, np.float32)
            lab = np.ctypeslib.as_array(lib.rt_results_cls_labels(r, i), (k,)).copy() if k else np.zeros(0, np.uint16)
            toks = []
            for j in range(k):
                tp = C.POINTER(C.c_int32)()
                nt = lib.rt_results_rec_tokens(r, i, j, C.byref(tp))
                toks.append(tuple(np.ctypeslib.as_array(tp, (nt,)).tolist()) if nt else ())
            out.append((boxes.tobytes(), rs.view(np.uint32).tobytes(), lab.tobytes(), tuple(toks)))
        return out

    ref = []
    for pages, maps in batches:
        r = hip_session.run_batch_raw(pages, [p.shape[0] for p in pages], [p.shape[1] for p in pages], retto_amd.RT_MEM_HOST, maps)
        ref.append(digest(r, len(pages))); lib.rt_results_free(r)
        assert sum(len(d[3]) for d in ref[-1]) > 0
    for order in ((0, 1, 2, 3), (3, 1, 0, 2)):
        tickets = [hip_session.submit_batch_raw(pages, [p.shape[0] for p in pages], [p.shape[1] for p in pages], retto_amd.RT_MEM_HOST, maps)
                   for pages, maps in batches]
        with pytest.raises(retto_amd.RettoError):   # the session is busy until every ticket has been waited for
            hip_session.worker.cls(np.zeros((1, 3, 48, 192), np.float32))
        for b in order:
            r = hip_session.wait_batch_raw(tickets[b])
            assert digest(r, len(batches[b][0])) == ref[b], "batch %d differs between rt_run_batch and submit / wait" % b
            lib.rt_results_free(r)
    assert hip_session.worker.cls(np.zeros((1, 3, 48, 192), np.float32)).shape == (1, 2)


def test_staged_host_pages_equal_pages_in_hbm(hip_session):
    """rt_submit_batch copies host pages to HBM itself, on a copy stream, one batch ahead of the lanes (session.cpp "page
    staging").  The three memory modes -- pages and maps on the host, pages on the host with maps in HBM, everything in HBM --
    must give the same bytes, over more batches in flight than staging slots existed before (slots are created, grown and
    reused) and with batches of different total size."""
    import torch
    lib = hip_session._hd.lib
    batches = []
    for b, shapes in enumerate((((352, 512, 3), (640, 480, 5)), ((640, 640, 6), (480, 704, 4), (352, 512, 3), (640, 480, 5), (320, 480, 2)),
                                ((736, 416, 4),))):
        pages, maps = [], []
        for i, (h, w, L) in enumerate(shapes):
            page, rects = workload.planted_page(h, w, L, seed=700 + 10 * b + i)
            dh, dw = R.resize_either_dims(h, w)
            pages.append(page); maps.append(workload.planted_map(dh, dw, h, w, rects))
        batches.append((pages, maps))

    def digest(r, n):
        return [hip_session._collect(r, i) for i in range(n)]

    def same(a, b):
        assert len(a) == len(b)
        for x, y in zip(a, b):
            assert len(x.det_result) == len(y.det_result) > 0
            assert [d.boxes.as_array().tobytes() for d in x.det_result] == [d.boxes.as_array().tobytes() for d in y.det_result]
            assert [c.label.label for c in x.cls_result] == [c.label.label for c in y.cls_result]
            assert [(t.text, np.float32(t.score).tobytes()) for t in x.rec_result] == [(t.text, np.float32(t.score).tobytes()) for t in y.rec_result]

    dev = [([torch.from_numpy(p).cuda() for p in pages], [torch.from_numpy(m).cuda() for m in maps]) for pages, maps in batches]
    torch.cuda.synchronize()
    ref = []
    for (pages, maps), (dp, dm) in zip(batches, dev):
        r = hip_session.run_batch_raw([t.data_ptr() for t in dp], [p.shape[0] for p in pages], [p.shape[1] for p in pages],
                                      retto_amd.RT_MEM_DEVICE, [t.data_ptr() for t in dm])
        ref.append(digest(r, len(pages))); lib.rt_results_free(r)
    for mode in (retto_amd.RT_MEM_HOST, retto_amd.RT_MEM_HOST_MAPS_DEVICE):
        for rounds in range(2):   # the second round reuses the slots of the first
            tickets = []
            for k in (1, 0, 2, 1, 2, 0):   # six batches in flight, sizes going up and down
                pages, maps = batches[k]
                mm = maps if mode == retto_amd.RT_MEM_HOST else [t.data_ptr() for t in dev[k][1]]
                tickets.append((k, hip_session.submit_batch_raw(pages, [p.shape[0] for p in pages], [p.shape[1] for p in pages], mode, mm)))
            for k, t in reversed(tickets):
                r = hip_session.wait_batch_raw(t)
                same(digest(r, len(batches[k][0])), ref[k]); lib.rt_results_free(r)


def test_submit_wait_against_the_oracle_teacher_forced(hip_session, oracle_session):
    """The TIMED path itself (bench.py times rt_submit_batch / rt_wait_batch with two batches in flight) compared with the
    oracle directly, not through rt_run_batch: two batches submitted ahead, waited in reverse order, every page against
    oracle/pipeline.py teacher-forced by the HIP worker (session.rs:75-106 / 108-143) -- boxes, score bit patterns, labels,
    token ids, text."""
    lib = hip_session._hd.lib
    batches = []
    for b, shapes in enumerate((((640, 640, 6), (480, 704, 4), (352, 512, 3), (640, 480, 5)), ((736, 416, 4), (320, 480, 2)))):
        pages, maps = [], []
        for i, (h, w, L) in enumerate(shapes):
            page, rects = workload.planted_page(h, w, L, seed=300 + 10 * b + i)
            dh, dw = R.resize_either_dims(h, w)
            pages.append(page); maps.append(workload.planted_map(dh, dw, h, w, rects))
        batches.append((pages, maps))
    tickets = [hip_session.submit_batch_raw(pages, [p.shape[0] for p in pages], [p.shape[1] for p in pages], retto_amd.RT_MEM_HOST, maps)
               for pages, maps in batches]
    got = {}
    for b in (1, 0):
        r = hip_session.wait_batch_raw(tickets[b])
        got[b] = [hip_session._collect(r, i) for i in range(len(batches[b][0]))]
        lib.rt_results_free(r)
    oracle_session.det_worker = hip_session.worker.det
    oracle_session.cls_worker = hip_session.worker.cls
    oracle_session.rec_worker = hip_session.worker.rec
    lines = 0
    for b, (pages, maps) in enumerate(batches):
        for page, m, r in zip(pages, maps, got[b]):
            o = oracle_session.run(page, det_map_override=m)
            assert len(r.det_result) == len(o.det_boxes) > 0
            assert np.array_equal(np.stack([d.boxes.as_array() for d in r.det_result]), o.det_boxes)
            assert np.array_equal(np.array([d.score for d in r.det_result], np.float32).view(np.uint32), o.det_scores.view(np.uint32))
            assert [c.label.label for c in r.cls_result] == list(o.cls_labels)
            for k, (g, ot) in enumerate(zip(r.rec_result, o.rec_tokens)):
                assert np.array_equal(g.tokens, ot), f"batch {b} line {k} tokens differ"
                assert g.text == o.rec_text[k]
            lines += len(o.det_boxes)
    assert lines >= 20


def test_two_sessions_on_two_devices(hip_session):
    """One process, two GPUs (what a multi-GPU host that does not use one process per GPU would do, and what the per-device
    state of the library must survive: > 64 KB dynamic-LDS opt-in per (device, kernel), streams and arenas bound to the session's
    device, lane threads that set their device): the same pages through a session on device 0 and one on device 1 give identical
    results.  Skipped on a box with fewer than two GPUs (the round's GPU box has one)."""
    import torch as _t
    if _t.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (hipGetDeviceCount() = %d)" % _t.cuda.device_count())
    other = retto_amd.RettoSession(retto_amd.synthetic_session_config(0, device=1))
    try:
        pages, maps = [], []
        for i in range(5):
            page, rects = workload.planted_page(480, 640, 4, seed=40 + i)
            dh, dw = R.resize_either_dims(480, 640)
            pages.append(page); maps.append(workload.planted_map(dh, dw, 480, 640, rects))
        a = hip_session.run_batch(pages, det_map_override=maps)
        b = other.run_batch(pages, det_map_override=maps)
        for pa, pb in zip(a, b):
            assert len(pa.det_result) == len(pb.det_result) > 0
            assert np.array_equal(np.stack([d.boxes.as_array() for d in pa.det_result]), np.stack([d.boxes.as_array() for d in pb.det_result]))
            assert [c.label.label for c in pa.cls_result] == [c.label.label for c in pb.cls_result]
            for ta, tb in zip(pa.rec_result, pb.rec_result):
                assert np.array_equal(ta.tokens, tb.tokens) and ta.text == tb.text
        x = np.random.default_rng(3).uniform(-1, 1, (1, 3, 320, 320)).astype(np.float32)
        assert np.array_equal(hip_session.worker.det(x).view(np.uint32), other.worker.det(x).view(np.uint32))
    finally:
        other.close()


def test_submit_wait_failed_batch_does_not_poison_the_session():
    """A batch that fails on one lane (an empty page: ImageError) returns ITS error from
    rt_wait_batch; the batch submitted behind it on the same lanes completes with the right results, and the session stays
    usable (the failed lane drains its stream before its next job rewinds the arenas)."""
    sess = retto_amd.RettoSession(retto_amd.synthetic_session_config(0))
    try:
        lib = sess._hd.lib
        good, gmaps = [], []
        for i in range(4):
            page, rects = workload.planted_page(320, 480, 3, seed=60 + i)
            dh, dw = R.resize_either_dims(320, 480)
            good.append(page); gmaps.append(workload.planted_map(dh, dw, 320, 480, rects))
        ref = sess.run_batch(good, det_map_override=gmaps)
        bad = list(good)
        bad_h = [p.shape[0] for p in bad]; bad_h[2] = 0   # an empty page: ImageError on the lane that gets it, after its neighbours started
        t_bad = sess.submit_batch_raw(bad, bad_h, [p.shape[1] for p in bad], retto_amd.RT_MEM_HOST, None)
        t_ok = sess.submit_batch_raw(good, [p.shape[0] for p in good], [p.shape[1] for p in good], retto_amd.RT_MEM_HOST, gmaps)
        with pytest.raises(retto_amd.ImageError):
            sess.wait_batch_raw(t_bad)
        r = sess.wait_batch_raw(t_ok)
        for i, pr in enumerate(ref):
            n = lib.rt_results_count(r, i)
            assert n == len(pr.det_result) > 0
            assert np.array_equal(np.ctypeslib.as_array(lib.rt_results_boxes(r, i), (n, 8)), np.stack([d.boxes.as_array().reshape(8) for d in pr.det_result]))
        lib.rt_results_free(r)
        again = sess.run_batch(good, det_map_override=gmaps)
        assert [len(p.det_result) for p in again] == [len(p.det_result) for p in ref]
    finally:
        sess.close()


def test_submit_ahead_at_the_cap_with_a_failing_submission():
    """RT_MAX_INFLIGHT (8; C2's default depth is 4) one-page submissions in flight, the page of the SECOND one is empty (ImageError
    on the lane that gets it).  Its ticket returns that error; the submissions behind it complete with the results of a
    synchronous call; one more submission while the cap is reached is refused without disturbing them; the cap is not leaked
    (as many submissions again go through afterwards)."""
    sess = retto_amd.RettoSession(retto_amd.synthetic_session_config(0))
    try:
        lib = sess._hd.lib
        pages, maps = [], []
        for i in range(4):
            page, rects = workload.planted_page(320, 480, 3, seed=90 + i)
            dh, dw = R.resize_either_dims(320, 480)
            pages.append(page); maps.append(workload.planted_map(dh, dw, 320, 480, rects))
        ref = [sess.run_batch([pages[i]], det_map_override=[maps[i]])[0] for i in range(4)]

        def submit(i, h=None):
            return sess.submit_batch_raw([pages[i]], [pages[i].shape[0] if h is None else h], [pages[i].shape[1]], retto_amd.RT_MEM_HOST,
                                         None if h == 0 else [maps[i]])

        def check(r, i):
            n = lib.rt_results_count(r, 0)
            assert n == len(ref[i].det_result) > 0
            assert np.array_equal(np.ctypeslib.as_array(lib.rt_results_boxes(r, 0), (n, 8)),
                                  np.stack([d.boxes.as_array().reshape(8) for d in ref[i].det_result]))
            lib.rt_results_free(r)

        cap = retto_amd.RT_MAX_INFLIGHT
        assert cap >= 5
        order = [0, 1, 2, 3] + [k % 4 for k in range(4, cap)]
        tickets = [submit(i, h=0) if k == 1 else submit(i) for k, i in enumerate(order)]
        with pytest.raises(retto_amd.RettoError):   # the cap: refused, nothing queued
            submit(0)
        for k, (i, t) in enumerate(zip(order, tickets)):
            if k == 1:
                with pytest.raises(retto_amd.ImageError):
                    sess.wait_batch_raw(t)
            else:
                check(sess.wait_batch_raw(t), i)
        again = [submit(i % 4) for i in range(cap)]
        for i, t in enumerate(again):
            check(sess.wait_batch_raw(t), i % 4)
        assert [len(p.det_result) for p in sess.run_batch(pages, det_map_override=maps)] == [len(p.det_result) for p in ref]
    finally:
        sess.close()
