#!/usr/bin/env python3
"""Writes tests/golden/pin/: the intermediates a maintainer WITH the reference toolchain needs to pin this backend's
restatement of the third-party algorithms, in formats a Rust test reads without numpy:

  pin.json            every structured value (integers as numbers, f32 / f64 values as decimal strings that round-trip
                      AND as hex bit patterns), one top-level key per section-9 row
  *.u8 / *.f32        raw little-endian arrays named <what>_<H>x<W>[x<C>].<dtype>

Inputs are seeded; expectations come from the CPU oracle (oracle/retto_oracle.cpp -- PARITY UNPINNED: this kit is how it
gets pinned).  Fixtures are data only.

    python tests/golden/make_pin_kit.py
"""
import json
import os
import struct
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle import ref_lib as R  # noqa: E402
from retto_amd import workload  # noqa: E402

PIN = os.path.join(HERE, "pin")


def f32(v):
    v = np.float32(v)
    return {"value": repr(float(v)), "bits": "0x%08x" % struct.unpack("<I", struct.pack("<f", float(v)))[0]}


def f64(v):
    return {"value": repr(float(v)), "bits": "0x%016x" % struct.unpack("<Q", struct.pack("<d", float(v)))[0]}


def raw(name, arr):
    arr = np.ascontiguousarray(arr)
    dt = {"uint8": "u8", "float32": "f32"}[str(arr.dtype)]
    fn = "%s_%s.%s" % (name, "x".join(str(d) for d in arr.shape), dt)
    arr.tofile(os.path.join(PIN, fn))
    return fn


def build():
    os.makedirs(PIN, exist_ok=True)
    for f in os.listdir(PIN):
        os.remove(os.path.join(PIN, f))
    rng = np.random.default_rng(2024)
    out = {"_readme": "INTEGRATION.md section 9; regenerate with tests/golden/make_pin_kit.py"}

    # ---- row 1: imageops::thumbnail (image 0.25.6) ---------------------------------------------------------------
    img = rng.integers(0, 256, (45, 70, 3), dtype=np.uint8)
    cases = []
    for tag, (nh, nw) in (("down", (20, 33)), ("up", (64, 96)), ("mixed", (60, 40))):
        cases.append({"case": tag, "new_width": nw, "new_height": nh, "expect": raw("thumb_" + tag, R.thumbnail(img, nh, nw))})
    out["thumbnail"] = {"input": raw("thumb_input", img), "layout": "HWC RGB8, row-major", "cases": cases}

    # ---- rows 2-5: find_contours, min_area_rect, box_score_fast, unclip on two planted maps ----------------------------
    maps = {
        "rot": workload.planted_map_rotated(160, 224, [(70, 30, 52, 8, 10.0), (70, 112, 52, 8, -20.0), (190, 80, 45, 7, 80.0)]),
        "nested": workload.planted_map(128, 160, 128, 160, [(4, 4, 150, 40), (20, 60, 140, 120)], shrink=0.0),
    }
    maps["nested"][70:110, 40:120] = 0.02
    maps["nested"][80:100, 60:100] = 0.9
    out["dbpost"] = {}
    for tag, m in maps.items():
        m = np.ascontiguousarray(m, np.float32)
        mask = R.threshold_dilate(m)
        conts = R.find_contours(mask)
        entry = {"pred": raw("pred_" + tag, m), "mask": raw("mask_" + tag, mask),
                 "mask_note": "pred > 0.3 (f32 compare) then grayscale_dilate with the 2x2 kernel of det_processor.rs:128-138; 255 = foreground",
                 "contours": []}
        for pts, bt in conts:
            rect = R.min_area_rect(pts.astype(np.float64))
            box = rect.astype(np.int32)   # the oracle floors in f64; the values are whole numbers
            c = {"border_type": "Outer" if bt == 0 else "Hole", "start": [int(pts[0][0]), int(pts[0][1])],
                 "n_points": int(len(pts)), "points": pts.astype(int).tolist(),
                 "min_area_rect": box.tolist()}
            s1 = np.hypot(*(box[0] - box[1]).astype(np.float32)); s2 = np.hypot(*(box[3] - box[2]).astype(np.float32))
            if min(s1, s2) >= 3:
                c["box_score_fast"] = f32(R.box_score_fast(m, box.reshape(8)))
                c["unclip_distance"] = f32(R.unclip_distance(box.reshape(8), 1.6))
                ring = R.unclip(box.reshape(8), 1.6)
                c["unclip_ring"] = ring.astype(int).tolist()    # Clipper works on i64: whole numbers, first point repeated last
            entry["contours"].append(c)
        boxes, scores = R.det_postprocess(m, m.shape[0], m.shape[1])
        entry["final_boxes"] = np.asarray(boxes, np.float32).reshape(-1, 4, 2).astype(int).tolist()
        entry["final_scores"] = [f32(s) for s in np.asarray(scores, np.float32)]
        out["dbpost"][tag] = entry

    # ---- row 6: Projection::from_control_points + warp_into(Bicubic) + rotate270 --------------------------------------
    page = rng.integers(0, 256, (160, 224, 3), dtype=np.uint8)
    boxes_rot, _ = R.det_postprocess(maps["rot"], 160, 224)
    crops = []
    for i, b in enumerate(np.asarray(boxes_rot, np.float32).reshape(-1, 8)):
        t, inv = R.crop_projection(b)
        crop = R.get_crop_img(page, b)
        crops.append({"box": [f32(v) for v in b], "crop": raw("crop%d" % i, crop), "crop_height": int(crop.shape[0]), "crop_width": int(crop.shape[1]),
                      "projection_forward_f32": [f32(v) for v in t.reshape(9)], "projection_inverse_f32": [f32(v) for v in inv.reshape(9)],
                      "note": "forward maps box corners to (0,0),(w,0),(w,h),(0,h); warp_into samples the page at inverse * (x, y, 1)"})
    out["crops"] = {"page": raw("crop_page", page), "cases": crops}

    # ---- row 7: reading-order sort on a non-transitive staircase (det_processor.rs:324-333) ----------------------------
    stairs = [[float(40 * (i % 5)), float(6 * i)] for i in range(12)]   # centres 6 px apart in y: "same line" is not transitive
    order = sorted(range(12), key=lambda i: i)   # placeholder: the defined order is computed below
    idx = list(range(12))

    def less(a, b):
        if abs(np.float32(stairs[a][1]) - np.float32(stairs[b][1])) < np.float32(10.0):
            return np.float32(stairs[a][0]) < np.float32(stairs[b][0])
        return np.float32(stairs[a][1]) < np.float32(stairs[b][1])
    width = 1
    while width < len(idx):       # bottom-up stable merge: the backend's defined behaviour on non-transitive input
        nxt = []
        for lo in range(0, len(idx), 2 * width):
            left, right = idx[lo:lo + width], idx[lo + width:lo + 2 * width]
            i = j = 0
            while i < len(left) and j < len(right):
                if less(right[j], left[i]):
                    nxt.append(right[j]); j += 1
                else:
                    nxt.append(left[i]); i += 1
            nxt += left[i:] + right[j:]
        idx, width = nxt, width * 2
    del order
    out["reading_order"] = {"centres_xy": stairs, "expected_order": idx,
                            "note": "boxes listed in discovery order; comparator: |dy| < 10 ? x : y.  slice::sort_by may return another order "
                                    "(or, from Rust 1.81, panic) on this input: record what it does"}

    # ---- row 8: RecCharacter::new on a dictionary with CR / LF / Unicode spaces ---------------------------------------
    data = b"a\r\n\xe3\x80\x80\n  b \t\nc"
    out["dictionary"] = {"bytes_hex": data.hex(), "expected_classes": ["blank", "a", "", "b", "c", " "],
                         "note": "String::from_utf8 -> lines() -> trim(); 'blank' inserted at 0, ' ' appended (rec_processor.rs:29-46)"}

    # ---- row 9: serde_json f32 formatting ---------------------------------------------------------------------------------
    vals = [0.9, 1e-7, 16777216.0, 123.0, 0.00001234, 1e30, -2.5, 0.1, 3.4028235e38, 1.17549435e-38]
    fmt = {0.9: "0.9", 1e-7: "1e-7", 16777216.0: "16777216.0", 123.0: "123.0", 0.00001234: "0.00001234", 1e30: "1e30", -2.5: "-2.5",
           0.1: "0.1", 3.4028235e38: "3.4028235e38", 1.17549435e-38: "1.1754944e-38"}
    out["json_f32"] = [{"bits": f32(v)["bits"], "expected": fmt[v]} for v in vals]

    with open(os.path.join(PIN, "pin.json"), "w") as f:
        json.dump(out, f, indent=1)
    return out


if __name__ == "__main__":
    build()
    total = sum(os.path.getsize(os.path.join(PIN, f)) for f in os.listdir(PIN))
    print("pin kit written to", PIN, "(%d files, %.1f KB)" % (len(os.listdir(PIN)), total / 1024))
