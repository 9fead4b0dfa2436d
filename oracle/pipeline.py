"""oracle/pipeline.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

CPU restatement of ``RettoSession::process_pipeline``
(/root/reference/retto-core/src/session.rs:75-106) and of the cls / rec
processor loops (/root/reference/retto-core/src/processor/cls_processor.rs:127-172,
/root/reference/retto-core/src/processor/rec_processor.rs:214-270) on top of
oracle/ref_lib.py (C++ pre/post restatement) and oracle/nets_torch.py (torch
fp32 networks).  PARITY UNPINNED -- see oracle/retto_oracle.cpp header.

The three ``worker`` callables play the role of ``RettoInnerWorker::{det,cls,
rec}``; tests may substitute the HIP worker's outputs there ("teacher forcing")
to compare the discrete stages bit-exactly on identical fp32 inputs.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Callable, List, Optional

import numpy as np
import torch

from . import nets_torch as nets
from . import ref_lib as R


@dataclass
class OracleResult:
    det_boxes: np.ndarray          # [n,4,2] f32, original-image coordinates (session.rs:94-97)
    det_scores: np.ndarray         # [n] f32
    cls_labels: np.ndarray         # [n] u16 (0 / 180)
    cls_scores: np.ndarray         # [n] f32
    rec_tokens: List[np.ndarray]   # kept CTC token ids per line
    rec_scores: np.ndarray         # [n] f32 (NaN when nothing kept)
    rec_text: List[str]
    # intermediates for stage-level parity tests
    boxes_after: np.ndarray = None     # boxes in after_* coordinates (crop source)
    crops: List[np.ndarray] = field(default_factory=list)  # after cls rotation
    rec_widths: List[int] = field(default_factory=list)    # per line: W of its batch tensor
    det_map: np.ndarray = None
    # decision margins of the fp32 networks (for tests that compare against an INDEPENDENT implementation of the networks):
    cls_margins: np.ndarray = None                              # |p(0) - p(180)| per crop
    rec_margins: List[float] = field(default_factory=list)      # per line: min over time steps of (top-1 - top-2) probability


# char::is_whitespace (Unicode White_Space), the set str::trim strips
_RUST_WS = frozenset([0x09, 0x0A, 0x0B, 0x0C, 0x0D, 0x20, 0x85, 0xA0, 0x1680, 0x2028, 0x2029, 0x202F, 0x205F, 0x3000]
                     + list(range(0x2000, 0x200B)))


def load_dictionary(dict_bytes: bytes) -> List[str]:
    """RecCharacter::new (rec_processor.rs:29-46): String::from_utf8 (strict: Python's utf-8 codec rejects the
    same overlong / surrogate / > U+10FFFF forms), str::lines (split after "\\n", drop it and one "\\r" before it,
    no empty last line), str::trim (Unicode White_Space -- NOT Python's str.strip, which also strips U+001C-001F),
    "blank" first, " " last."""
    txt = dict_bytes.decode("utf-8")  # raises UnicodeDecodeError where Rust returns Utf8Error
    lines = txt.split("\n")
    if lines and lines[-1] == "":
        lines.pop()
    out = ["blank"]
    for ln in lines:
        a, b = 0, len(ln)
        while a < b and ord(ln[a]) in _RUST_WS:
            a += 1
        while b > a and ord(ln[b - 1]) in _RUST_WS:
            b -= 1
        out.append(ln[a:b])
    return out + [" "]


class OracleSession:
    def __init__(self, det_blob: bytes, cls_blob: bytes, rec_blob: bytes, dict_bytes: bytes,
                 max_side_len=2000, min_side_len=30):
        self.wd = nets.read_blob(det_blob)
        self.wc = nets.read_blob(cls_blob)
        self.wr = nets.read_blob(rec_blob)
        # RecCharacter::new (rec_processor.rs:29-46)
        self.dict = load_dictionary(dict_bytes)
        self.max_side_len, self.min_side_len = max_side_len, min_side_len
        self.det_worker: Callable = lambda t: nets.det_forward(self.wd, torch.from_numpy(t)).numpy()
        self.cls_worker: Callable = lambda t: nets.cls_forward(self.wc, torch.from_numpy(t)).numpy()
        self.rec_worker: Callable = lambda t: nets.rec_forward(self.wr, torch.from_numpy(t)).numpy()

    # cls_processor.rs:127-172
    def cls_process(self, crops: List[np.ndarray], dims):
        n = len(crops)
        labels = np.zeros(n, np.uint16); scores = np.zeros(n, np.float32)
        self._cls_margins = np.zeros(n, np.float32)
        # sort_by_key(Reverse(ori_ratio)): stable, descending h/w (f64)
        order = sorted(range(n), key=lambda i: -(float(dims[i][0]) / float(dims[i][1])))
        batches = []
        for s in range(0, n, 6):
            idxs = order[s:s + 6]
            t = np.stack([R.resize_norm_image(crops[i], dims[i][0], dims[i][1], 48, 192, 0.0) for i in idxs])
            batches.append((idxs, t))
        for idxs, t in batches:
            out = self.cls_worker(t)
            idx, sc = R.cls_postprocess(out)
            for j, i in enumerate(idxs):
                self._cls_margins[i] = abs(float(out[j, 0]) - float(out[j, 1]))
                label = [0, 180][int(idx[j])]
                if label == 180 and sc[j] >= np.float32(0.9):
                    crops[i] = R.rotate180(crops[i])
                labels[i] = label; scores[i] = sc[j]
        return labels, scores

    # rec_processor.rs:214-270
    def rec_process(self, crops: List[np.ndarray], dims):
        n = len(crops)
        toks: List[Optional[np.ndarray]] = [None] * n
        scores = np.zeros(n, np.float32); widths = [0] * n
        self._rec_margins = [0.0] * n
        order = sorted(range(n), key=lambda i: -(float(dims[i][0]) / float(dims[i][1])))
        max_wh_ratio = np.float32(320) / np.float32(48)
        for s in range(0, n, 6):
            idxs = order[s:s + 6]
            for i in idxs:
                h, w = crops[i].shape[:2]
                max_wh_ratio = max(max_wh_ratio, np.float32(w) / np.float32(h))
            t = np.stack([R.resize_norm_image(crops[i], dims[i][0], dims[i][1], 48, 320, float(max_wh_ratio))
                          for i in idxs])
            probs = self.rec_worker(t)
            _, _, tk, sc = R.ctc_decode(probs)
            top2 = np.sort(probs, axis=-1)[..., -2:]
            for j, i in enumerate(idxs):
                toks[i] = tk[j]; scores[i] = sc[j]; widths[i] = t.shape[3]
                self._rec_margins[i] = float((top2[j, :, 1] - top2[j, :, 0]).min())
        return toks, scores, widths

    # session.rs:75-106
    def run(self, page_rgb: np.ndarray, det_map_override: Optional[np.ndarray] = None) -> OracleResult:
        ori_h, ori_w = page_rgb.shape[:2]
        image = R.resize_both(page_rgb, self.max_side_len, self.min_side_len)
        after_h, after_w = image.shape[:2]
        x = R.det_preprocess(image)
        det_map = self.det_worker(x)[0, 0]
        if det_map_override is not None:
            assert det_map_override.shape == det_map.shape
            det_map = np.ascontiguousarray(det_map_override, np.float32)
        boxes, scores = R.det_postprocess(det_map, after_h, after_w)
        crops = [R.get_crop_img(image, b) for b in boxes]
        dims = [c.shape[:2] for c in crops]  # ImageHelper ori_h/ori_w = construction-time dims
        boxes_after = boxes.copy()
        boxes_ori = np.stack([R.scale_and_clip(b, after_w, after_h, ori_w, ori_h) for b in boxes]) \
            if len(boxes) else boxes.reshape(0, 4, 2)
        self._cls_margins = np.zeros(len(crops), np.float32)   # (a test may swap cls_process / rec_process for its own)
        self._rec_margins = [0.0] * len(crops)
        labels, cscores = self.cls_process(crops, dims)
        toks, rscores, widths = self.rec_process(crops, dims)
        text = ["".join(self.dict[t] for t in tk) for tk in toks]
        return OracleResult(boxes_ori, scores, labels, cscores, toks, rscores, text, boxes_after, crops, widths,
                            det_map, self._cls_margins.copy() if len(crops) else np.zeros(0, np.float32), list(self._rec_margins))
