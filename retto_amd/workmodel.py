"""Algorithmic work (bytes moved, FLOPs) of each kernel family, derived from the layer
tables: for every launch, input activations read
once + output activations written once (fp32, real channel counts) + weights once.
This is the `B_layer` accounting of BASELINE.md, split per kernel family so that the
roofline line in bench.py can price the dominant kernel.  Pure arithmetic; no GPU.
"""
from __future__ import annotations

import os

from collections import defaultdict
from typing import Dict, Iterable, Tuple

from . import synth

F = 4  # bytes per fp32


def _down(v: int, s: int) -> int:
    return (v - 1) // s + 1


def gemm_pw_label(M: int, N: int, se: bool = False, K: int = 0, min_pix: int = 1 << 30) -> str:
    """Mirror of nn::gemm_pw_label + the predicates behind it (retto_amd/csrc/nn_kernels.hip gemm_dispatch / gemm_se_tile_rows,
    nn_gemm_dma.hip gemm_dma_supported): which kernel symbol the dispatcher picks for a pointwise conv of M rows, K input and
    N output channels.  `se`: the block has a squeeze-excite whose scale is folded into the GEMM's A staging; `min_pix`: rows
    of the smallest image of the level (the fused form needs every image to cover a row tile).  K = 0: unknown, assumed to
    satisfy the K conditions (callers inside this module always pass it)."""
    npad = (N + 15) // 16 * 16
    dma_on = os.environ.get("RT_GEMM_DMA", "1") != "0"
    # k_gemm32p: N == Npad16 a multiple of 240 (<= 480 per bias table), whole 16-deep K groups, more than 3 slabs of 32, M >= 131072
    dma_shape = npad == N and N % 240 == 0 and M >= 131072 and (K == 0 or (K % 16 == 0 and K > 96))
    if se:
        if dma_on and dma_shape and min_pix >= 128 and (K == 0 or K <= 512):
            return "gemm_pw/k_gemm32p+se"
        wide = npad % 240 == 0 and M >= 16384
        rows_ok = min_pix >= 128
        if rows_ok and (K == 0 or K <= 512) and (wide or (M >= 8192 and npad >= 128)):
            return "gemm_pw/k_gemm_wide<2,5,4,3>+se" if wide else "gemm_pw/k_gemm_wide<2,4,4,2>+se"
        # no fused form: the tensor is scaled in a pass of its own and the GEMM is the plain one
    if npad % 240 == 0 and M >= 131072:   # (the persistent LDS-DMA form unless RT_GEMM_DMA=0 / an unsupported K keeps the register-staged tile)
        return "gemm_pw/k_gemm32p" if (dma_on and dma_shape) else "gemm_pw/k_gemm_wide<4,5,4,3>"
    if npad % 240 == 0 and M >= 16384:
        return "gemm_pw/k_gemm_wide<2,5,4,3>"
    return "gemm_pw/thin"


def lc_thin_fused(k: int, sh: int, sw: int, cin: int, cout: int, se: bool) -> bool:
    """Mirror of nn::lc_block_supported: thin 3x3 blocks run as ONE kernel (k_lc_lds, nn_lcwave.hip; k_lc_thin with
    RT_LC_WAVE=0), whose algorithmic traffic is the block's input + output (the depthwise result never reaches HBM)."""
    if se or k != 3 or cin % 16 or cout % 16:
        return False
    g, nt = cin // 16, cout // 16
    if (sh, sw) == (1, 1):
        return (g, nt) in ((1, 2), (2, 4), (3, 3), (4, 4))
    if (sh, sw) == (2, 2):
        return (g, nt) in ((2, 3), (3, 6))
    if (sh, sw) == (2, 1):   # the rec net's 64 -> 128 block: k_lc_lds only
        return (g, nt) == (4, 8) and os.environ.get("RT_LC_WAVE", "3") != "0"
    return False


def _adder():
    w = defaultdict(lambda: {"bytes": 0.0, "flops": 0.0})

    def add(fam, b, f=0.0):
        w[fam]["bytes"] += b; w[fam]["flops"] += f
    return w, add


DET_GROUP_PX = 32 * 960 * 960   # session.cpp: det launch-group budget (det-input pixels)
REC_GROUP_PX = 24000000         # session.cpp: rec launch-group budget (48 x W pixels)


def _groups(sizes, budget):
    out, cur, acc = [], [], 0
    for i, px in enumerate(sizes):
        if cur and acc + px > budget:
            out.append(cur); cur, acc = [], 0
        cur.append(i); acc += px
    if cur:
        out.append(cur)
    return out


def det_work(pages: Iterable[Tuple[int, int]], phase=None) -> Dict[str, Dict[str, float]]:
    """pages: det-input (H, W) per page.  Returns family -> {bytes, flops}.  phase: True = the launch series as executed since round
    4 (nn_fpn.hip: upsampling-aware FPN convs, 7.59 GFLOP per 960x960 page), False = the reference graph's convs as rounds 1-3 ran
    them (10.36 GFLOP); None = what the library does (RT_FPN_PHASE)."""
    w = defaultdict(lambda: {"bytes": 0.0, "flops": 0.0})

    def add(fam, b, f=0.0):
        w[fam]["bytes"] += b; w[fam]["flops"] += f

    pages = list(pages)
    group_of = {}
    for g in _groups([h * w_ for h, w_ in pages], DET_GROUP_PX):
        for i in g:
            group_of[i] = g
    for pi, (H, W) in enumerate(pages):
        grp = [pages[i] for i in group_of[pi]]
        h, ww = _down(H, 2), _down(W, 2)
        # the det stem reads the RGB8 page itself (normalisation folded in): 3 bytes per input pixel
        add("stem", H * W * 3 + h * ww * 16 * F, 2 * h * ww * 27 * 16)
        taps = {}
        for name, k, cin, cout, sh, sw, se in synth.DET_BLOCKS:
            ho, wo = _down(h, sh), _down(ww, sw)
            if lc_thin_fused(k, sh, sw, cin, cout, se):
                add("lc_thin", (h * ww * cin + ho * wo * cout) * F + (k * k * cin + cin * cout) * F, 2 * ho * wo * cin * (k * k + cout))
            else:
                add("dwconv%d" % k, (h * ww + ho * wo) * cin * F + k * k * cin * F, 2 * ho * wo * cin * k * k)
                scale = (ho * wo) / float(H * W)
                m_group = int(round(sum(gh * gw for gh, gw in grp) * scale))
                min_pix = int(round(min(gh * gw for gh, gw in grp) * scale))
                add(gemm_pw_label(m_group, cout, se, cin, min_pix), ho * wo * (cin + cout) * F + cin * cout * F, 2 * ho * wo * cin * cout)
            h, ww = ho, wo
            for j, (tn, tc, oc) in enumerate(synth.DET_TAPS):
                if tn == name:
                    add("gemm_misc", h * ww * (tc + oc) * F + tc * oc * F, 2 * h * ww * tc * oc)
                    taps[j] = (h, ww, oc)
        if phase is None:
            phase = os.environ.get("RT_FPN_PHASE", "1") != "0"   # nn_fpn.hip: upsampling-aware forms of inp0 / inp1 / the head conv
        for j in range(4):
            h, ww, oc = taps[j]
            cf = (oc + 3) // 4 * 4
            if j == 3:  # coarsest level: lateral GEMM, squeeze-excite pooled on its output and applied in place
                add("gemm_misc", h * ww * (oc + 96) * F + oc * 96 * F, 2 * h * ww * oc * 96)
                add("se_pool_fc", h * ww * 96 * F)
                add("scale_channels", 2 * h * ww * 96 * F)
            else:  # lateral conv + SE factor + top-down add in one pass (squeeze from the narrow tap tensor)
                h2, w2, _ = taps[j + 1]
                add("se_pool_fc", h * ww * oc * F)
                if not (phase and j == 0):   # the finest lateral tensor is never built on the phase path
                    add("lateral_add", (h * ww * (oc + 96) + h2 * w2 * 96) * F + oc * 96 * F, 2 * h * ww * oc * 96)
            if phase and j < 2:
                # p_j = conv3x3(tap_j; per-image composed weights) + four 2x2 phase convs of in_{j+1}: FLOPs of the algorithm as
                # executed (9 cf + 4 x 96 deep instead of 9 x 96)
                h2, w2, _ = taps[j + 1]
                add("fpn_compose", (oc * 96 + 9 * 24 * 96 + 9 * 24 * cf) * F, 2 * 9 * 24 * oc * 96)
                add("conv3x3_phase", (h * ww * (cf + 24) + h2 * w2 * 96) * F + (9 * 24 * cf + 16 * 96 * 24) * F,
                    2 * h * ww * 24 * (9 * cf + 4 * 96))
                add("se_pool_fc", ((h + 15) // 16) * ((ww + 15) // 16) * 24 * F)
            else:
                add("conv3x3", h * ww * (96 + 24) * F + 9 * 96 * 24 * F, 2 * h * ww * 9 * 96 * 24)
                add("se_pool_fc", h * ww * 24 * F)
        h4, w4, _ = taps[0]
        if phase:
            # head conv: p2 at its own resolution (9 x 24 deep), p3 as phase convs (4 x 24), p4 / p5 through their class tensors
            # (9 classes x <= 4 taps x 24 x 24 MACs per coarse pixel; written once, gathered once per output pixel)
            (h3, w3, _), (h16, w16, _), (h32, w32, _) = taps[1], taps[2], taps[3]
            add("fpn_class", ((h16 * w16 + h32 * w32) * (24 + 9 * 24) + h16 * w16 * 24) * F + 2 * 81 * 576 * F,
                2 * (h16 * w16 + h32 * w32) * 25 * 576)
            add("conv3x3_phase", (h4 * w4 * (24 + 24 + 24) + h3 * w3 * 24) * F + (9 * 24 * 24 + 16 * 24 * 24) * F,
                2 * h4 * w4 * 24 * (9 * 24 + 4 * 24))
        else:
            add("fpn_concat", sum(taps[j][0] * taps[j][1] for j in range(4)) * 24 * F + h4 * w4 * 96 * F)
            add("conv3x3", h4 * w4 * (96 + 24) * F + 9 * 96 * 24 * F, 2 * h4 * w4 * 9 * 96 * 24)
        add("db_head_tail", h4 * w4 * 24 * F + H * W * F, 2 * h4 * w4 * 4 * (24 * 24 + 4 * 24))
    return dict(w)


def _rows_at(W: int, block: str) -> int:
    """Pixels of a 48 x W line at the output of recognition block `block` (the stem halves both sides first)."""
    h, ww = _down(48, 2), _down(W, 2)
    for name, _k, _cin, _cout, sh, sw, _se in synth.REC_BLOCKS:
        h, ww = _down(h, sh), _down(ww, sw)
        if name == block:
            break
    return h * ww


def rec_work(widths: Iterable[int], classes: int = synth.REC_CLASSES) -> Dict[str, Dict[str, float]]:
    """widths: padded width W of every 48-high line tensor."""
    w = defaultdict(lambda: {"bytes": 0.0, "flops": 0.0})

    def add(fam, b, f=0.0):
        w[fam]["bytes"] += b; w[fam]["flops"] += f

    widths = list(widths)
    group_of = {}
    for g in _groups([48 * w_ for w_ in widths], REC_GROUP_PX):
        for i in g:
            group_of[i] = g
    for wi, W in enumerate(widths):
        H = 48
        grp_px = sum(48 * widths[i] for i in group_of[wi])
        h, ww = _down(H, 2), _down(W, 2)
        add("stem", H * W * 3 * F + h * ww * 16 * F, 2 * h * ww * 27 * 16)
        for name, k, cin, cout, sh, sw, se in synth.REC_BLOCKS:
            ho, wo = _down(h, sh), _down(ww, sw)
            if lc_thin_fused(k, sh, sw, cin, cout, se):
                add("lc_thin", (h * ww * cin + ho * wo * cout) * F + (k * k * cin + cin * cout) * F, 2 * ho * wo * cin * (k * k + cout))
            else:
                add("dwconv%d" % k, (h * ww + ho * wo) * cin * F + k * k * cin * F, 2 * ho * wo * cin * k * k)
                m_group = int(round(grp_px * (ho * wo) / float(H * W)))
                min_pix = min(_rows_at(widths[i], name) for i in group_of[wi])
                add(gemm_pw_label(m_group, cout, se, cin, min_pix), ho * wo * (cin + cout) * F + cin * cout * F, 2 * ho * wo * cin * cout)
            h, ww = ho, wo
        T = (ww - 2) // 2 + 1
        add("avgpool", (h * ww + T) * 480 * F)
        add("conv1x3", T * (480 + 60) * F + 3 * 480 * 60 * F, 2 * T * 3 * 480 * 60)
        add("conv1x3", T * (960 + 60) * F + 3 * 960 * 60 * F, 2 * T * 3 * 960 * 60)
        for cin, cout, cnt in ((60, 120, 2), (120, 360, 2), (120, 120, 2), (120, 240, 2), (240, 120, 2), (120, 480, 1)):
            add("gemm_neck", cnt * (T * (cin + cout) * F + cin * cout * F), cnt * 2 * T * cin * cout)
        add("attention", 2 * T * (360 + 120) * F, 2 * 2 * 2 * T * T * 120)
        add("layernorm", 5 * 3 * T * 120 * F)
        tiles = ((classes + 15) // 16 * 16 + 127) // 128  # fused CTC head: softmax statistics per 128-column tile, no logits
        add("gemm_ctc_fc", T * 120 * F + 120 * classes * F + T * tiles * 12, 2 * T * 120 * classes)
        add("ctc_argmax", T * tiles * 12 + T * 8)
    return dict(w)


def cls_work(n_crops: int, H: int = 48, W: int = 192) -> Dict[str, Dict[str, float]]:
    """Angle classifier (MobileNetV3-small x0.35) over n_crops crops of H x W: one fused kernel per block (k_cls_block):
    block input read + output written once (+ the depthwise scratch round trip of the squeeze-excite blocks), expand +
    depthwise + linear FLOPs."""
    w, add = _adder()
    h, ww = _down(H, 2), _down(W, 2)
    add("stem", n_crops * (H * W * 4 * F + h * ww * synth.CLS_STEM * F), n_crops * 2 * h * ww * 27 * synth.CLS_STEM)
    cin = synth.CLS_STEM
    for k, mid, cout, se, _act, sh, sw in synth.CLS_BLOCKS:
        ho, wo = _down(h, sh), _down(ww, sw)
        by = (h * ww * cin + ho * wo * cout) * F + (2 * ho * wo * mid * F if se else 0) + (cin * mid + k * k * mid + mid * cout) * F
        fl = 2 * h * ww * cin * mid + 2 * ho * wo * mid * (k * k + cout)
        add("cls_block", n_crops * by, n_crops * fl)
        h, ww, cin = ho, wo, cout
    add("gemm_cls", n_crops * (h * ww * (cin + synth.CLS_LAST) * F + 202 * F) + cin * synth.CLS_LAST * F,
        n_crops * (2 * h * ww * cin * synth.CLS_LAST + 2 * synth.CLS_LAST * 2))
    hp, wp = h // 2, ww // 2
    add("maxpool", n_crops * (h * ww + hp * wp) * synth.CLS_LAST * F)
    add("global_mean", n_crops * (hp * wp + 1) * synth.CLS_LAST * F)
    add("softmax", n_crops * 6 * F)
    return dict(w)


def prepost_work(pages: Iterable[Tuple[int, int]], det_pages: Iterable[Tuple[int, int]], crops: Iterable[Tuple[int, int]],
                 rec_widths: Iterable[int], tokens: int) -> Dict[str, Dict[str, float]]:
    """The u8 <-> f32 stages around the networks, as executed.  pages: (H, W) after the size limits; det_pages: det-input dims;
    crops: (h, w) of every text-line crop; rec_widths: padded width of every line tensor; tokens: time steps over all lines.
    db_postprocess: the full-map passes of dbpost_kernels.hip per det pixel -- map read (4 B), mask written and read (1 + 1),
    labels written (4) and read by the link / statistics / row-extent passes (3 x 4): 22 B; the per-contour geometry is not
    bandwidth."""
    w, add = _adder()
    for (H, W), (dh, dw) in zip(pages, det_pages):
        if (H, W) != (dh, dw):
            add("thumbnail", (H * W + dh * dw) * 3)
        add("db_postprocess", dh * dw * 22)
    crops = list(crops)
    crop_bytes = sum(h * ww_ * 3 for h, ww_ in crops)
    add("warp_crops", 2 * crop_bytes, 0.0)                       # source pixels under the quad read once, crop written once
    add("resize_norm", 2 * crop_bytes + len(crops) * 48 * 192 * 4 * F + sum(48 * wv * 4 * F for wv in rec_widths))  # cls + rec launches
    add("cls_post_rotate", len(crops) * 16)
    add("ctc_decode", tokens * 12)
    return dict(w)


def line_geometry(lib, results_per_page, rec_batch_num: int = 6, rec_h: int = 48, rec_w: int = 320):
    """(crops, widths): the (h, w) of every text-line crop and the padded width of its recognition tensor, from the boxes of
    RettoWorkerResult pages -- the session's own batching (rec_processor.rs:214-270: aspect-sorted chunks of rec_batch_num, running
    max_wh_ratio), restated here so that the work model prices exactly the tensors the session built."""
    import numpy as np
    crops, widths = [], []
    for pr in results_per_page:
        dims = []
        for d in pr.det_result:
            b = d.boxes.as_array().reshape(1, 8).astype(np.float32)
            wv = np.zeros(1, np.int32); hv = np.zeros(1, np.int32)
            lib.rt_crop_dims(b.ctypes.data, 1, wv.ctypes.data, hv.ctypes.data)
            dims.append((int(hv[0]), int(wv[0])))
        crops += dims
        order = sorted(range(len(dims)), key=lambda i: -(dims[i][0] / dims[i][1]))
        ratio = np.float32(rec_w) / np.float32(rec_h)
        for s0 in range(0, len(order), rec_batch_num):
            idx = order[s0:s0 + rec_batch_num]
            for i in idx:
                ratio = max(ratio, np.float32(dims[i][1]) / np.float32(dims[i][0]))
            widths += [lib.rt_resize_norm_width(rec_h, rec_w, float(ratio))] * len(idx)
    return crops, widths


def tokens_for_width(w: int) -> int:
    wa = (w - 1) // 2 + 1
    wc = (wa - 1) // 2 + 1
    return (wc - 2) // 2 + 1 if wc >= 2 else 0


def step_work(det_pages, crops, widths, pages=None) -> Dict[str, Dict[str, float]]:
    """Every launch family of a mobile fp32 step: det + cls + rec networks and the u8 <-> f32 stages around them."""
    det_pages = list(det_pages)
    work: Dict[str, Dict[str, float]] = {}
    parts = [det_work(det_pages), rec_work(widths), cls_work(len(crops)),
             prepost_work(pages if pages is not None else det_pages, det_pages, crops, widths, sum(tokens_for_width(w) for w in widths))]
    for part in parts:
        for k, v in part.items():
            if k in work:
                work[k]["bytes"] += v["bytes"]; work[k]["flops"] += v["flops"]
            else:
                work[k] = dict(v)
    return work


def page_flops(det_hw: Tuple[int, int], widths: Iterable[int]) -> float:
    d = sum(v["flops"] for v in det_work([det_hw]).values())
    r = sum(v["flops"] for v in rec_work(widths).values())
    return d + r


# ---------------------------------------------------------------------------------------------------------
# fp16 family (rt_config.dtype = f16): labels of nets_f16.cpp's profiler scopes.  Bytes: every launch reads its input
# activations once and writes its output once at 2 bytes per element (real channel counts) + the fp16 weights once.
# ---------------------------------------------------------------------------------------------------------
H2 = 2  # bytes per fp16


def conv16_label(kh: int, kw: int, n: int, cin: int = 32) -> str:
    """Mirror of nh::conv16_label (retto_amd/csrc/nn_f16.hip)."""
    if kh == 1 and kw == 1:
        return "gemm16/thin" if n <= 64 else "gemm16"
    if (kh, kw) == (3, 3):
        return "conv16_stem" if cin < 32 else "conv16_3x3"
    if kh == 9:
        return "conv16_9x9"
    return "conv16_kxk"


def _conv16(add, pix_in, pix_out, cin, cout, kh, kw):
    add(conv16_label(kh, kw, cout, cin), (pix_in * cin + pix_out * cout) * H2 + kh * kw * cin * cout * H2, 2.0 * pix_out * kh * kw * cin * cout)


def _lc16(add, h, ww, blocks):
    for name, k, cin, cout, sh, sw, se in blocks:
        ho, wo = _down(h, sh), _down(ww, sw)
        add("dwconv16_%d" % k, (h * ww + ho * wo) * cin * H2 + k * k * cin * H2, 2.0 * ho * wo * cin * k * k)
        if se:
            add("se_pool_fc16", ho * wo * cin * H2)
            add("scale_channels16", 2 * ho * wo * cin * H2)
        _conv16(add, ho * wo, ho * wo, cin, cout, 1, 1)
        h, ww = ho, wo
        yield name, h, ww


def _neck16(add, T, C, classes):
    D = 120
    add("avgpool16", (6 * T + T) * C * H2)
    _conv16(add, T, T, C, C // 8, 1, 3)
    _conv16(add, T, T, C // 8, D, 1, 1)
    add("f16_to_f32", 2 * T * D * 6); add("f32_to_f16", T * D * 6)
    for cin, cout, cnt in ((120, 360, 2), (120, 120, 2), (120, 240, 2), (240, 120, 2)):
        add("gemm_neck", cnt * (T * (cin + cout) * F + cin * cout * F), cnt * 2.0 * T * cin * cout)
    add("attention", 2 * T * (360 + 120) * F, 2 * 2 * 2.0 * T * T * 120)
    add("layernorm", 5 * 3 * T * 120 * F)
    _conv16(add, T, T, D, C, 1, 1)
    _conv16(add, T, T, 2 * C, C // 8, 1, 3)
    _conv16(add, T, T, C // 8, D, 1, 1)
    tiles = ((classes + 15) // 16 * 16 + 127) // 128
    add("gemm_ctc_fc", T * 120 * F + 120 * classes * F + T * tiles * 12, 2.0 * T * 120 * classes)
    add("ctc_argmax", T * tiles * 12 + T * 8)


def det16_work(pages: Iterable[Tuple[int, int]]) -> Dict[str, Dict[str, float]]:
    """PP-OCRv4 mobile det in fp16 (DetNetH)."""
    w, add = _adder()
    for H, W in pages:
        add("u8_to_f16", H * W * 3 + H * W * 8 * H2)
        h, ww = _down(H, 2), _down(W, 2)
        _conv16(add, H * W, h * ww, 8, 16, 3, 3)
        taps = {}
        for name, h, ww in _lc16(add, h, ww, synth.DET_BLOCKS):
            for j, (tn, tc, oc) in enumerate(synth.DET_TAPS):
                if tn == name:
                    _conv16(add, h * ww, h * ww, tc, oc, 1, 1)
                    taps[j] = (h, ww, oc)
        for j in range(4):
            h, ww, oc = taps[j]
            _conv16(add, h * ww, h * ww, oc, 96, 1, 1)
            add("se_pool_fc16", h * ww * 96 * H2)
            add("scale_channels16" if j == 3 else "upsample_add16", (2 * h * ww * 96 + (0 if j == 3 else taps[j + 1][0] * taps[j + 1][1] * 96)) * H2)
            _conv16(add, h * ww, h * ww, 96, 24, 3, 3)
            add("se_pool_fc16", h * ww * 24 * H2)
        h4, w4, _ = taps[0]
        add("fpn_concat16", (sum(taps[j][0] * taps[j][1] for j in range(4)) * 24 + h4 * w4 * 96) * H2)
        _conv16(add, h4 * w4, h4 * w4, 96, 24, 3, 3)
        _conv16(add, h4 * w4, h4 * w4, 24, 96, 1, 1)
        add("pixel_shuffle16", 2 * h4 * w4 * 96 * H2)
        add("db_head_tail16", 4 * h4 * w4 * 24 * H2 + H * W * F, 2.0 * 4 * h4 * w4 * 24 * 4)
    return dict(w)


def rec16_work(widths: Iterable[int], classes: int = synth.REC_CLASSES) -> Dict[str, Dict[str, float]]:
    """PP-OCRv4 mobile rec in fp16 (RecNetH)."""
    w, add = _adder()
    for W in widths:
        H = 48
        add("f32_to_f16", H * W * (16 + 16))
        h, ww = _down(H, 2), _down(W, 2)
        _conv16(add, H * W, h * ww, 8, 16, 3, 3)
        for _name, h, ww in _lc16(add, h, ww, synth.REC_BLOCKS):
            pass
        _neck16(add, (ww - 2) // 2 + 1, 480, classes)
    return dict(w)


def _hgnet_work(add, h, ww, stages, strides):
    """PPHGNet_small from the stem output resolution (h, ww) on; yields (stage index, h, w, channels)."""
    for si, ((name, cin, mid, cout, blocks, down), (sh, sw)) in enumerate(zip(stages, strides)):
        if down:
            ho, wo = _down(h, sh), _down(ww, sw)
            add("dwconv16_3", (h * ww + ho * wo) * cin * H2, 2.0 * ho * wo * cin * 9)
            h, ww = ho, wo
        for b in range(blocks):
            bin_ = cin if b == 0 else cout
            c = bin_
            for _l in range(synth.HG_LAYERS):
                _conv16(add, h * ww, h * ww, c, mid, 3, 3)
                c = mid
            _conv16(add, h * ww, h * ww, bin_ + synth.HG_LAYERS * mid, cout, 1, 1)
            add("se_pool_fc16", h * ww * cout * H2 + cout * cout * F)
            add("scale_channels16", (2 + (1 if b > 0 else 0)) * h * ww * cout * H2)
        yield si, h, ww, cout


def sdet_work(pages: Iterable[Tuple[int, int]]) -> Dict[str, Dict[str, float]]:
    """PP-OCRv4 server det (DetServerH): PPHGNet_small + LKPAN(256, intracl) + PFHeadLocal, as executed -- the IntraCL branch
    triples are folded into one k x k conv, PFHeadLocal's 3x3 on the upsampled feature runs as four 2x2 phase convs
    (K = 4 x 80) at half resolution."""
    w, add = _adder()
    for H, W in pages:
        add("u8_to_f16", H * W * 3 + H * W * 8 * H2)
        h, ww = _down(H, 2), _down(W, 2)
        _conv16(add, H * W, h * ww, 8, 64, 3, 3)
        _conv16(add, h * ww, h * ww, 64, 64, 3, 3)
        _conv16(add, h * ww, h * ww, 64, 128, 3, 3)
        h4, w4 = _down(h, 2), _down(ww, 2)
        add("maxpool16", (h * ww + h4 * w4) * 128 * H2)
        lv = []
        for si, hh, wv, c in _hgnet_work(add, h4, w4, synth.HG_STAGES_DET, [(1, 1)] + synth.HG_STRIDES_DET[1:]):
            lv.append((hh, wv, c))
        for j, (hh, wv, c) in enumerate(lv):
            px = hh * wv
            _conv16(add, px, px, c, 256, 1, 1)
            if j < 3:
                add("upsample_add16", (2 * px + lv[j + 1][0] * lv[j + 1][1]) * 256 * H2)
            _conv16(add, px, px, 256, 64, 9, 9)
            if j > 0:
                _conv16(add, lv[j - 1][0] * lv[j - 1][1], px, 64, 64, 3, 3)
            _conv16(add, px, px, 64, 64, 9, 9)
            _conv16(add, px, px, 64, 32, 1, 1)
            for k in (7, 5, 3):
                add("conv16_kxk", 2 * px * 32 * H2 + k * k * 32 * 32 * H2, 2.0 * px * k * k * 32 * 32)
            _conv16(add, px, px, 32, 64, 1, 1)
            if j > 0:
                add("fpn_concat16", (px + h4 * w4) * 64 * H2)
        p4 = h4 * w4
        _conv16(add, p4, p4, 256, 64, 3, 3)
        _conv16(add, p4, p4, 64, 256, 1, 1)
        add("pixel_shuffle16", 2 * p4 * 256 * H2)
        p2 = h * ww
        add("db_head_tail16", p2 * 64 * H2 + H * W * F, 2.0 * p2 * 64 * 4)
        add("map_window16", H * W * F + p2 * 16 * H2)
        add("conv16_local", 4 * (p2 * 80 * H2 + p2 * 2 * F) + 4 * 4 * 80 * 64 * H2, 4 * 2.0 * p2 * (4 * 80 * 64 + 64))
    return dict(w)


def srec_work(widths: Iterable[int], classes: int = synth.REC_CLASSES) -> Dict[str, Dict[str, float]]:
    """PP-OCRv4 server rec (RecServerH)."""
    w, add = _adder()
    for W in widths:
        H = 48
        add("f32_to_f16", H * W * (16 + 16))
        h, ww = _down(H, 2), _down(W, 2)
        _conv16(add, H * W, h * ww, 8, 64, 3, 3)
        _conv16(add, h * ww, h * ww, 64, 64, 3, 3)
        _conv16(add, h * ww, h * ww, 64, 128, 3, 3)
        for _si, h, ww, _c in _hgnet_work(add, h, ww, synth.HG_STAGES_REC, synth.HG_STRIDES_REC):
            pass
        _neck16(add, (ww - 2) // 2 + 1, 1024, classes)
    return dict(w)
