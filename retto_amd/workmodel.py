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


def gemm_pw_label(M: int, N: int, se: bool = False) -> str:
    """Mirror of nn::gemm_pw_label (retto_amd/csrc/nn_kernels.hip): which kernel symbol the
    dispatcher picks for a pointwise conv of M rows and N output channels.  `se`: the block has a
    squeeze-excite whose scale is folded into the GEMM's A staging (wide tiles, every image
    >= 128 rows at that level -- true for the bench workloads)."""
    npad = (N + 15) // 16 * 16
    if se and npad % 240 == 0 and M >= 131072 and os.environ.get("RT_GEMM_DMA", "1") != "0":
        return "gemm_pw/k_gemm32p+se"   # (images of >= 128 rows at that level: true for the bench workloads)
    if se and M >= 8192 and npad >= 128:
        return "gemm_pw/k_gemm_wide<2,5,4,3>+se" if npad % 240 == 0 and M >= 16384 else "gemm_pw/k_gemm_wide<2,4,4,2>+se"
    if npad % 240 == 0 and M >= 131072:   # (the persistent LDS-DMA form unless RT_GEMM_DMA=0 keeps the register-staged tile)
        return "gemm_pw/k_gemm32p" if os.environ.get("RT_GEMM_DMA", "1") != "0" else "gemm_pw/k_gemm_wide<4,5,4,3>"
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


def det_work(pages: Iterable[Tuple[int, int]]) -> Dict[str, Dict[str, float]]:
    """pages: det-input (H, W) per page.  Returns family -> {bytes, flops}."""
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
                add(gemm_pw_label(m_group, cout, se), ho * wo * (cin + cout) * F + cin * cout * F, 2 * ho * wo * cin * cout)
            h, ww = ho, wo
            for j, (tn, tc, oc) in enumerate(synth.DET_TAPS):
                if tn == name:
                    add("gemm_misc", h * ww * (tc + oc) * F + tc * oc * F, 2 * h * ww * tc * oc)
                    taps[j] = (h, ww, oc)
        for j in range(4):
            h, ww, oc = taps[j]
            if j == 3:  # coarsest level: lateral GEMM, squeeze-excite pooled on its output and applied in place
                add("gemm_misc", h * ww * (oc + 96) * F + oc * 96 * F, 2 * h * ww * oc * 96)
                add("se_pool_fc", h * ww * 96 * F)
                add("scale_channels", 2 * h * ww * 96 * F)
            else:  # lateral conv + SE factor + top-down add in one pass (squeeze from the narrow tap tensor)
                h2, w2, _ = taps[j + 1]
                add("se_pool_fc", h * ww * oc * F)
                add("lateral_add", (h * ww * (oc + 96) + h2 * w2 * 96) * F + oc * 96 * F, 2 * h * ww * oc * 96)
            add("conv3x3", h * ww * (96 + 24) * F + 9 * 96 * 24 * F, 2 * h * ww * 9 * 96 * 24)
            add("se_pool_fc", h * ww * 24 * F)
        h4, w4, _ = taps[0]
        add("fpn_concat", sum(taps[j][0] * taps[j][1] for j in range(4)) * 24 * F + h4 * w4 * 96 * F)
        add("conv3x3", h4 * w4 * (96 + 24) * F + 9 * 96 * 24 * F, 2 * h4 * w4 * 9 * 96 * 24)
        add("db_head_tail", h4 * w4 * 24 * F + H * W * F, 2 * h4 * w4 * 4 * (24 * 24 + 4 * 24))
    return dict(w)


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
                add(gemm_pw_label(m_group, cout, se), ho * wo * (cin + cout) * F + cin * cout * F, 2 * ho * wo * cin * cout)
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


def page_flops(det_hw: Tuple[int, int], widths: Iterable[int]) -> float:
    d = sum(v["flops"] for v in det_work([det_hw]).values())
    r = sum(v["flops"] for v in rec_work(widths).values())
    return d + r


# ---------------------------------------------------------------------------------------------------------
# fp16 family (rt_config.dtype = f16): labels of nets_f16.cpp's profiler scopes.  Bytes: every launch reads its input
# activations once and writes its output once at 2 bytes per element (real channel counts) + the fp16 weights once.
# ---------------------------------------------------------------------------------------------------------
H2 = 2  # bytes per fp16


def conv16_label(kh: int, kw: int, n: int) -> str:
    """Mirror of nh::conv16_label (retto_amd/csrc/nn_f16.hip)."""
    if kh == 1 and kw == 1:
        return "gemm16/thin" if n <= 64 else "gemm16"
    if (kh, kw) == (3, 3):
        return "conv16_3x3"
    if kh == 9:
        return "conv16_9x9"
    return "conv16_kxk"


def _conv16(add, pix_in, pix_out, cin, cout, kh, kw):
    add(conv16_label(kh, kw, cout), (pix_in * cin + pix_out * cout) * H2 + kh * kw * cin * cout * H2, 2.0 * pix_out * kh * kw * cin * cout)


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


def _adder():
    w = defaultdict(lambda: {"bytes": 0.0, "flops": 0.0})

    def add(fam, b, f=0.0):
        w[fam]["bytes"] += b; w[fam]["flops"] += f
    return w, add


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
