"""Algorithmic work (bytes moved, FLOPs) of each kernel family, derived from the layer
tables: for every launch, input activations read
once + output activations written once (fp32, real channel counts) + weights once.
This is the `B_layer` accounting of BASELINE.md, split per kernel family so that the
roofline line in bench.py can price the dominant kernel.  Pure arithmetic; no GPU.
"""
from __future__ import annotations

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
    if se and M >= 8192 and npad >= 128:
        return "gemm_pw/k_gemm_wide<2,5,4,3>+se" if npad % 240 == 0 and M >= 16384 else "gemm_pw/k_gemm_wide<2,4,4,2>+se"
    if npad % 240 == 0 and M >= 131072:
        return "gemm_pw/k_gemm_wide<4,5,4,3>"
    if npad % 240 == 0 and M >= 16384:
        return "gemm_pw/k_gemm_wide<2,5,4,3>"
    return "gemm_pw/thin"


def lc_thin_fused(k: int, sh: int, sw: int, cin: int, cout: int, se: bool) -> bool:
    """Mirror of nn::lc_thin_supported: thin stride-1 3x3 blocks run as ONE kernel (k_lc_thin), whose algorithmic
    traffic is the block's input + output (the depthwise result never reaches HBM)."""
    if se or k != 3 or cin % 4:
        return False
    c4, nt = cin // 4, ((cout + 15) // 16 * 16 + 31) // 32
    if (sh, sw) == (1, 1):
        return (c4, nt) in ((4, 1), (8, 2), (12, 2), (16, 2))
    if (sh, sw) == (2, 2):
        return (c4, nt) in ((8, 2), (12, 3))
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
