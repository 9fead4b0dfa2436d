#!/usr/bin/env python3
"""Regenerates tests/golden/*.npz from the CPU oracle (oracle/, PARITY UNPINNED: the
reference holds no golden vectors of its own, see oracle/retto_oracle.cpp).  The
fixtures are data only: seeded inputs and the oracle's outputs.  They pin the oracle
against accidental drift and give the GPU tests a second, committed reference.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle import ref_lib as R  # noqa: E402
from retto_amd import workload  # noqa: E402


def main():
    rng = np.random.default_rng(2024)
    # a2/a3: thumbnail paths (box average, fractional upscaling) + det input tensor
    img = rng.integers(0, 256, (45, 70, 3), dtype=np.uint8)
    np.savez_compressed(os.path.join(HERE, "preprocess.npz"), img=img,
                        thumb_down=R.thumbnail(img, 20, 33), thumb_up=R.thumbnail(img, 64, 96),
                        thumb_mixed=R.thumbnail(img, 60, 40), det_input=R.det_preprocess(img, limit_len=64))
    # a5: DB post-processing on planted maps
    m1 = workload.planted_map_rotated(160, 224, [(70, 30, 52, 8, 10.0), (70, 112, 52, 8, -20.0), (190, 80, 45, 7, 80.0)])
    b1, s1 = R.det_postprocess(m1, 160, 224)
    m2 = workload.planted_map(128, 160, 128, 160, [(4, 4, 150, 40), (20, 60, 140, 120)], shrink=0.0)
    m2[70:110, 40:120] = 0.02
    m2[80:100, 60:100] = 0.9
    b2, s2 = R.det_postprocess(m2, 128, 160)
    np.savez_compressed(os.path.join(HERE, "dbpost.npz"), map_rot=m1, boxes_rot=b1, scores_rot=s1,
                        map_nested=m2, boxes_nested=b2, scores_nested=s2)
    # a6 + a8/a10: crops and resize-norm tensors
    page = rng.integers(0, 256, (160, 224, 3), dtype=np.uint8)
    crops = [R.get_crop_img(page, b) for b in b1]
    d = {"page": page, "boxes": b1}
    for i, c in enumerate(crops):
        d["crop%d" % i] = c
        d["cls%d" % i] = R.resize_norm_image(c, c.shape[0], c.shape[1], 48, 192, 0.0)
        d["rec%d" % i] = R.resize_norm_image(c, c.shape[0], c.shape[1], 48, 320, 9.5)
    np.savez_compressed(os.path.join(HERE, "crops.npz"), **d)
    # a12: CTC greedy decode on engineered rows (sparse encoding: row-wise top entries)
    n, t, c = 3, 12, 6625
    ids = np.array([[0] * 12, [5, 5, 0, 5, 7, 7, 7, 0, 0, 9, 6624, 6624], [1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12]], np.int32)
    top = (0.5 + 0.4 * rng.random((n, t))).astype(np.float32)
    probs = np.full((n, t, c), 1e-5, np.float32)
    for i in range(n):
        for k in range(t):
            probs[i, k, ids[i, k]] = top[i, k]
    probs[1, 4, 3] = probs[1, 4, 7]  # tie -> first index (3) wins
    idx, pr, toks, sc = R.ctc_decode(probs)
    np.savez_compressed(os.path.join(HERE, "ctc.npz"), ids=ids, top=top, tie=np.array([1, 4, 3, 7]), idx=idx, prob=pr,
                        tok0=toks[0], tok1=toks[1], tok2=toks[2], score=sc)
    print("golden fixtures written to", HERE)


if __name__ == "__main__":
    main()
