"""Synthetic OCR pages for benchmarks and tests.

A *planted page* is a black RGB page with ``L`` white text-line-like rounded
rectangles whose heights and aspect ratios come from a seeded distribution.  Because
the det network only has synthetic (random) weights, its probability map carries no
structure; the matching *planted probability map* (what a trained DBNet would emit for
such a page: ~1 inside shrunk text regions, ~0 elsewhere) is generated next to the
page and injected through ``det_map_override`` so that box extraction, cropping and
recognition run on a realistic, deterministic number of lines per page while the det
network is still executed in full on the page pixels.
"""
from __future__ import annotations

from typing import List, Tuple

import numpy as np


def planted_page(h: int, w: int, lines: int = 32, seed: int = 0, ratio_range=(3.0, 20.0),
                 rotate_deg: float = 0.0) -> Tuple[np.ndarray, List[Tuple[int, int, int, int]]]:
    """Returns (page uint8 [h,w,3], list of (x0, y0, x1, y1) line rectangles)."""
    rng = np.random.default_rng(seed)
    page = np.zeros((h, w, 3), np.uint8)
    rects = []
    margin = 12
    band = (h - 2 * margin) / max(lines, 1)
    for i in range(lines):
        lh = int(min(max(band * 0.45, 10), 40))
        ratio = rng.uniform(*ratio_range)
        lw = int(min(lh * ratio, w - 2 * margin - 4))
        x0 = margin + int(rng.integers(0, max(1, w - 2 * margin - lw)))
        y0 = int(margin + i * band + (band - lh) / 2)
        rects.append((x0, y0, x0 + lw, y0 + lh))
        # glyph-like texture inside the line so the crops are not flat
        tex = rng.integers(120, 256, (lh, lw, 3), dtype=np.uint8)
        tex[:, ::7] //= 3
        page[y0:y0 + lh, x0:x0 + lw] = tex
    if rotate_deg:
        raise NotImplementedError("rotated planted pages are built by planted_map_rotated()")
    return page, rects


def planted_map(det_h: int, det_w: int, page_h: int, page_w: int, rects, inside: float = 0.92,
                outside: float = 0.02, shrink: float = 0.12) -> np.ndarray:
    """Probability map (f32 [det_h, det_w]) for the rectangles of a planted page.  DB
    predicts a *shrunk* text kernel; ``shrink`` is the fraction of the line height
    removed on every side (the unclip step grows the box back)."""
    m = np.full((det_h, det_w), outside, np.float32)
    sy, sx = det_h / page_h, det_w / page_w
    for (x0, y0, x1, y1) in rects:
        d = shrink * (y1 - y0)
        a, b = int(round((y0 + d) * sy)), int(round((y1 - d) * sy))
        c, e = int(round((x0 + d) * sx)), int(round((x1 - d) * sx))
        if b > a and e > c:
            m[a:b, c:e] = inside
    return m


def planted_map_rotated(det_h: int, det_w: int, boxes, inside: float = 0.92, outside: float = 0.02) -> np.ndarray:
    """boxes: list of (cx, cy, half_len, half_thick, angle_deg) rotated rectangles."""
    yy, xx = np.mgrid[0:det_h, 0:det_w].astype(np.float32)
    m = np.full((det_h, det_w), outside, np.float32)
    for cx, cy, hl, ht, deg in boxes:
        th = np.deg2rad(deg)
        u = (xx - cx) * np.cos(th) + (yy - cy) * np.sin(th)
        v = -(xx - cx) * np.sin(th) + (yy - cy) * np.cos(th)
        m[(np.abs(u) < hl) & (np.abs(v) < ht)] = inside
    return m


def noise_page(h: int, w: int, seed: int = 0) -> np.ndarray:
    return np.random.default_rng(seed).integers(0, 256, (h, w, 3), dtype=np.uint8)
