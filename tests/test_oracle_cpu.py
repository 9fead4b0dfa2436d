"""CPU tests of the oracle: hand-derived known answers (from the reference's source
arithmetic, /root/reference/retto-core/src, and the traced examples of SURVEY.md
Appendix A) and the committed golden fixtures.  The reference's own tests hold no
golden vectors (session.rs:206-255 need network + ORT), so parity is UNPINNED; these
tests pin the oracle itself."""
import os

import numpy as np
import pytest

from oracle import ref_lib as R
from retto_amd import workload

G = os.path.join(os.path.dirname(__file__), "golden")


# ---- sizes: SURVEY A.1 traced examples (image_helper.rs:106-174) --------------------
@pytest.mark.parametrize("hw,after,det", [
    ((640, 640), (640, 640), (736, 736)), ((960, 960), (960, 960), (960, 960)),
    ((50, 200), (50, 200), (736, 2944)), ((4320, 7680), (1120, 1984), (1120, 1984)),
    ((1754, 1240), (1754, 1240), (1760, 1248)), ((3508, 2480), (1984, 1408), (1984, 1408)),
    ((1080, 1920), (1080, 1920), (1088, 1920)), ((720, 1280), (720, 1280), (736, 1312)),
    ((2000, 2000), (2000, 2000), (2016, 2016))])
def test_size_arithmetic(hw, after, det):
    plan = R.resize_both_plan(*hw)
    got_after = plan[-1] if plan else hw
    assert got_after == after
    assert R.resize_either_dims(*after) == det


def test_resize_both_min_side_uses_original_dims():
    # image_helper.rs:131-145: 20x300 -> scale 1.5 -> floor(30)/32 rounds to 1 -> 32, floor(450)/32=14.06 -> 14*32
    assert R.resize_both_plan(20, 300) == [(32, 448)]


# ---- thumbnail (image 0.25.6) --------------------------------------------------------
def test_thumbnail_identity_and_box_average():
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, (12, 16, 3), dtype=np.uint8)
    assert np.array_equal(R.thumbnail(img, 12, 16), img)
    half = R.thumbnail(img, 6, 8)
    blocks = img.reshape(6, 2, 8, 2, 3).astype(np.uint32).sum(axis=(1, 3))
    assert np.array_equal(half, ((blocks + 2) // 4).astype(np.uint8))  # (sum + n/2) / n


def test_thumbnail_upscale_constant_image_is_constant():
    img = np.full((5, 7, 3), 200, np.uint8)
    up = R.thumbnail(img, 11, 15)
    # fractional paths compute fact_a*v + fact_b*v with f32 weights summing to ~1, then truncate
    assert up.min() >= 199 and up.max() <= 200


# ---- det normalise (det_processor.rs:151-160) -----------------------------------------
def test_det_normalize_values_and_bgr_order():
    img = np.zeros((32, 32, 3), np.uint8)
    img[..., 0] = 255; img[..., 1] = 128; img[..., 2] = 0  # R, G, B
    x = R.det_preprocess(img, limit_len=32)
    s = np.float32(1.0) / np.float32(255.0)
    exp = lambda v: (np.float32(v) * s - np.float32(0.5)) / np.float32(0.5)
    assert x.shape == (1, 3, 32, 32)
    assert x[0, 0, 0, 0] == exp(0) and x[0, 1, 0, 0] == exp(128) and x[0, 2, 0, 0] == exp(255)  # B, G, R


# ---- contours / boxes ---------------------------------------------------------------------
def test_find_contours_known_shapes():
    m = np.zeros((8, 10), np.uint8)
    m[1, 1] = 255                 # isolated pixel
    m[3:7, 3:8] = 255             # 4x5 block with a 1-pixel hole
    m[4, 5] = 0
    cs = R.find_contours(m)
    assert [bt for _, bt in cs] == [0, 0, 1]                  # outer, outer, hole (raster order of start pixels)
    assert cs[0][0].tolist() == [[1, 1]]
    assert cs[1][0][0].tolist() == [3, 3]                     # outer border starts at the top-left pixel
    assert set(map(tuple, cs[2][0].tolist())) == {(4, 4), (5, 3), (6, 4), (5, 5)}  # 4-neighbours of the hole


def test_dilate_offsets():
    p = np.zeros((5, 5), np.float32); p[2, 2] = 1.0
    m = R.threshold_dilate(p, 0.3, True)
    assert sorted(map(tuple, np.argwhere(m > 0).tolist())) == [(2, 2), (2, 3), (3, 2), (3, 3)]
    assert (R.threshold_dilate(np.full((3, 3), 0.3, np.float32)) == 0).all()  # strictly greater


def test_min_area_rect_axis_aligned_and_degenerate():
    pts = [(x, y) for x in range(3, 10) for y in range(2, 6)]
    assert R.min_area_rect(pts).tolist() == [[3, 2], [9, 2], [9, 5], [3, 5]]
    assert R.min_area_rect([(4, 4)]).tolist() == [[4, 4]] * 4
    assert R.min_area_rect([(1, 1), (5, 1)]).tolist() == [[1, 1], [5, 1], [5, 1], [1, 1]]


def test_unclip_distance_axis_aligned():
    # 220x30 box: area*1.6/perimeter = 6600*1.6/500 = 21.12 -> offset by round(21.12) on straight edges
    pts = R.unclip(np.array([[30, 40], [250, 40], [250, 70], [30, 70]], np.int32))
    assert pts[:, 0].min() == 30 - 21 and pts[:, 0].max() == 250 + 21
    assert pts[:, 1].min() == 40 - 21 and pts[:, 1].max() == 70 + 21


def test_det_postprocess_planted_rectangle():
    pred = np.full((320, 320), 0.01, np.float32)
    pred[40:70, 30:250] = 0.9
    boxes, scores = R.det_postprocess(pred, 320, 320)
    assert boxes.reshape(-1, 8).tolist() == [[9, 19, 271, 19, 271, 91, 9, 91]]
    assert 0.85 < scores[0] < 0.9   # the dilated row/column (0.01) is inside the scored polygon
    # threshold / filters
    assert len(R.det_postprocess(np.full((64, 64), 0.2, np.float32), 64, 64)[0]) == 0
    tiny = np.zeros((64, 64), np.float32); tiny[10, 10:12] = 0.9
    assert len(R.det_postprocess(tiny, 64, 64)[0]) == 0      # min side < 3


def test_reference_small_image_scenario():
    """session.rs:206-229 restated without font/network: text-like blob in the bottom-right of a
    200x50 page (the 180-degree rotated render); first box's bottom-right within 10 px of (200, 50)."""
    pred = np.full((736, 2944), 0.02, np.float32)           # 50x200 -> det input 736x2944 (A.1)
    pred[int(0.55 * 736):int(0.97 * 736), int(0.5 * 2944):int(0.985 * 2944)] = 0.9
    boxes, _ = R.det_postprocess(pred, 50, 200)
    assert len(boxes) == 1
    br = boxes[0, 2]
    assert np.hypot(br[0] - 200, br[1] - 50) < 10


# ---- points / crops -------------------------------------------------------------------------
def test_scale_and_clip_round_and_clamp():
    b = np.array([[10.5, -3], [99.6, 2], [200, 80], [0.4, 79.5]], np.float32)
    out = R.scale_and_clip(b, 100, 80, 200, 160)
    assert out.tolist() == [[21, 0], [199, 4], [199, 159], [1, 159]]   # round half away, clamp to [0, ori-1]


def test_crop_axis_aligned_interior_matches_source():
    rng = np.random.default_rng(3)
    img = rng.integers(0, 256, (60, 80, 3), dtype=np.uint8)
    box = np.array([[10, 12], [50, 12], [50, 30], [10, 30]], np.float32)
    crop = R.get_crop_img(img, box)
    assert crop.shape == (18, 40, 3)
    # identity-scale homography: bicubic at integer positions reproduces the source pixel
    assert np.array_equal(crop, img[12:30, 10:50])
    tall = np.array([[10, 5], [20, 5], [20, 45], [10, 45]], np.float32)
    assert R.crop_dims(tall) == (40, 10, True)            # h/w >= 1.5 -> rotate270
    c2 = R.get_crop_img(img, tall)
    assert np.array_equal(c2, np.rot90(img[5:45, 10:20], 1))


def test_crop_outside_is_white():
    img = np.zeros((40, 40, 3), np.uint8)
    box = np.array([[-5, 5], [20, 5], [20, 20], [-5, 20]], np.float32)
    crop = R.get_crop_img(img, box)
    assert (crop[:, 0] == 255).all()                      # footprint leaves the image -> default white


def test_resize_norm_padding_and_range():
    crop = np.full((24, 100, 3), 255, np.uint8)
    t = R.resize_norm_image(crop, 24, 100, 48, 320, 320 / 48)
    assert t.shape == (3, 48, 320)
    assert (t[:, :, :200] == 1.0).all() and (t[:, :, 200:] == 0.0).all()   # ceil(48*100/24)=200, zero padded
    assert R.resize_norm_image(crop, 24, 100, 48, 192, 0.0).shape == (3, 48, 192)


# ---- CTC (rec_processor.rs:48-97) -----------------------------------------------------------
def test_ctc_known_answers():
    d = np.load(os.path.join(G, "ctc.npz"))
    ids, top = d["ids"], d["top"]
    n, t = ids.shape
    probs = np.full((n, t, 6625), 1e-5, np.float32)
    for i in range(n):
        for k in range(t):
            probs[i, k, ids[i, k]] = top[i, k]
    i_, k_, a_, b_ = d["tie"]
    probs[i_, k_, a_] = probs[i_, k_, b_]
    idx, pr, toks, sc = R.ctc_decode(probs)
    assert toks[0].tolist() == [] and np.isnan(sc[0])                      # all blank -> 0/0
    assert toks[1].tolist() == [5, 5, 3, 7, 9, 6624]                        # repeats collapse, blanks split, tie -> first
    assert toks[2].tolist() == list(range(1, 13))
    assert idx[1, 4] == 3
    assert np.array_equal(idx, d["idx"]) and np.array_equal(pr, d["prob"]) and np.array_equal(sc.view(np.uint32), d["score"].view(np.uint32))


# ---- golden regression ----------------------------------------------------------------------
def test_golden_preprocess():
    d = np.load(os.path.join(G, "preprocess.npz"))
    assert np.array_equal(R.thumbnail(d["img"], 20, 33), d["thumb_down"])
    assert np.array_equal(R.thumbnail(d["img"], 64, 96), d["thumb_up"])
    assert np.array_equal(R.thumbnail(d["img"], 60, 40), d["thumb_mixed"])
    assert np.array_equal(R.det_preprocess(d["img"], limit_len=64), d["det_input"])


def test_golden_dbpost_and_crops():
    d = np.load(os.path.join(G, "dbpost.npz"))
    for k in ("rot", "nested"):
        b, s = R.det_postprocess(d["map_" + k], *d["map_" + k].shape)
        assert np.array_equal(b, d["boxes_" + k]) and np.array_equal(s, d["scores_" + k])
    assert len(d["boxes_rot"]) == 3 and len(d["boxes_nested"]) >= 2
    c = np.load(os.path.join(G, "crops.npz"))
    for i, b in enumerate(c["boxes"]):
        crop = R.get_crop_img(c["page"], b)
        assert np.array_equal(crop, c["crop%d" % i])
        assert np.array_equal(R.resize_norm_image(crop, crop.shape[0], crop.shape[1], 48, 192, 0.0), c["cls%d" % i])
        assert np.array_equal(R.resize_norm_image(crop, crop.shape[0], crop.shape[1], 48, 320, 9.5), c["rec%d" % i])


def test_oracle_pipeline_runs(oracle_session):
    page, rects = workload.planted_page(160, 320, 2, seed=1)
    dh, dw = R.resize_either_dims(160, 320)
    r = oracle_session.run(page, det_map_override=workload.planted_map(dh, dw, 160, 320, rects))
    assert len(r.det_boxes) == 2 and len(r.rec_tokens) == 2
    assert all(w >= 320 for w in r.rec_widths)
    assert set(r.cls_labels.tolist()) <= {0, 180}


# ---- independent cross-checks of the restated third-party algorithms ------------------------------
def test_contours_agree_with_scipy_labelling():
    """Suzuki-Abe's result set = one outer border per 8-connected foreground component + one hole border per
    4-connected background component that does not touch the frame (imageproc 0.25 find_contours); counted
    independently with scipy.ndimage.label.  Every border pixel is foreground and belongs to the right component."""
    from scipy import ndimage
    rng = np.random.default_rng(3)
    for trial in range(12):
        h, w = int(rng.integers(5, 40)), int(rng.integers(5, 60))
        m = (rng.uniform(0, 1, (h, w)) < rng.uniform(0.2, 0.8)).astype(np.uint8) * 255
        cs = R.find_contours(m)
        fg, n_fg = ndimage.label(m > 0, structure=np.ones((3, 3)))
        bg, n_bg = ndimage.label(m == 0)  # 4-connected
        frame = set(np.unique(np.concatenate([bg[0], bg[-1], bg[:, 0], bg[:, -1]]))) - {0}
        assert sum(1 for _, bt in cs if bt == 0) == n_fg
        assert sum(1 for _, bt in cs if bt == 1) == n_bg - len(frame)
        seen = set()
        for pts, bt in cs:
            assert all(m[y, x] for x, y in pts)
            if bt == 0:
                labels = {int(fg[y, x]) for x, y in pts}
                assert len(labels) == 1 and not (labels & seen)  # one outer border per component
                seen |= labels


def test_min_area_rect_against_brute_force():
    """imageproc's rotating calipers must find the minimum over hull-edge directions: compare the area with a
    brute-force scan of every hull edge in numpy (the corner order / flooring is the restatement's own)."""
    from scipy.spatial import ConvexHull
    rng = np.random.default_rng(5)
    for trial in range(20):
        pts = rng.integers(0, 200, (int(rng.integers(5, 60)), 2)).astype(np.float64)
        hull = pts[ConvexHull(pts).vertices]
        best = np.inf
        for i in range(len(hull)):
            e = hull[(i + 1) % len(hull)] - hull[i]
            e /= np.hypot(*e)
            u, v = hull @ e, hull @ np.array([-e[1], e[0]])
            best = min(best, (u.max() - u.min()) * (v.max() - v.min()))
        r = R.min_area_rect(pts)
        a = np.hypot(*(r[1] - r[0])) * np.hypot(*(r[2] - r[1]))
        # corners are truncated to integers (min_area_rect returns Point<i32>): allow the perimeter's worth of slack
        assert abs(a - best) <= 2.0 * (np.hypot(*(r[1] - r[0])) + np.hypot(*(r[2] - r[1]))) + 4.0
        # and every input point lies inside the (1-pixel dilated) rectangle
        c = r.mean(0); ex = (r[1] - r[0]); ey = (r[3] - r[0])
        lx, ly = np.hypot(*ex), np.hypot(*ey)
        if lx > 0 and ly > 0:
            pu, pv = (pts - c) @ (ex / lx), (pts - c) @ (ey / ly)
            assert np.abs(pu).max() <= lx / 2 + 1.5 and np.abs(pv).max() <= ly / 2 + 1.5


def test_unclip_area_matches_minkowski_sum():
    """Round-join offsetting of a convex polygon by d is its Minkowski sum with a disc: area = A + P*d + pi*d^2
    (Clipper's arcs are chords within arc tolerance 0.5 and its coordinates are integers: ~1 % slack)."""
    rng = np.random.default_rng(8)
    for trial in range(12):
        w, h = rng.uniform(40, 300), rng.uniform(10, 60)
        th = np.deg2rad(rng.uniform(-90, 90))
        c, s = np.cos(th), np.sin(th)
        quad = np.array([(500 + sx * w / 2 * c - sy * h / 2 * s, 500 + sx * w / 2 * s + sy * h / 2 * c)
                         for sx, sy in ((-1, -1), (1, -1), (1, 1), (-1, 1))]).round().astype(np.int32)
        def shoelace(p):
            x, y = p[:, 0].astype(np.float64), p[:, 1].astype(np.float64)
            return 0.5 * abs(np.dot(x, np.roll(y, -1)) - np.dot(y, np.roll(x, -1)))
        A = shoelace(quad)
        P = sum(np.hypot(*(quad[(i + 1) % 4] - quad[i]).astype(np.float64)) for i in range(4))
        d = A * 1.6 / P  # det_processor.rs:233-239
        out = R.unclip(quad)
        expect = A + P * d + np.pi * d * d
        assert abs(shoelace(out) - expect) <= 0.012 * expect + 2 * P


def test_cpp_openmp_nets_match_the_torch_oracle(models):
    """oracle/nets_cpu.cpp (the C++ / OpenMP networks of bench.py's cpu_baseline leg) is the same fp32 graph as
    oracle/nets_torch.py: outputs agree to fp32 accumulation-order noise."""
    import torch
    from oracle import cpu_baseline as CB
    from oracle import nets_torch as N
    det, cls, rec, _dic = models
    nets = CB.CpuNets(det, cls, rec)
    try:
        rng = np.random.default_rng(3)
        x = rng.uniform(-1, 1, (2, 3, 64, 96)).astype(np.float32)
        assert np.abs(nets.det(x) - N.det_forward(N.read_blob(det), torch.from_numpy(x)).numpy()).max() <= 1e-4
        x = rng.uniform(-1, 1, (4, 3, 48, 192)).astype(np.float32)
        assert np.abs(nets.cls(x) - N.cls_forward(N.read_blob(cls), torch.from_numpy(x)).numpy()).max() <= 1e-4
        x = rng.uniform(-1, 1, (2, 3, 48, 333)).astype(np.float32)
        x[:, :, :, 200:] = 0.0
        got, ref = nets.rec(x), N.rec_forward(N.read_blob(rec), torch.from_numpy(x)).numpy()
        assert got.shape == ref.shape and np.abs(got - ref).max() <= 2e-4
    finally:
        nets.close()


def test_cpu_baseline_protocol():
    """bench.py's cpu_baseline leg: fresh interpreter, warm-up + 3 repetitions, CPU count, 1-thread figure, per-stage breakdown."""
    from oracle import cpu_baseline as CB
    out = CB.run_subprocess(160, 2, budget_s=4.0)
    assert out["kind"] == "port" and out["unit"] == "images/s" and out["value"] and out["value"] > 0 and out["cores"] >= 1
    assert set(out["stage_cpu_ms_per_page"]) == {"det", "cls", "rec", "pre_post"}
    assert "3 timed repetitions" in out["sample"] and out["one_thread_images_per_s"] > 0
