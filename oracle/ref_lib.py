"""oracle/ref_lib.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

ctypes binding of oracle/libretto_oracle.so (the C++ CPU restatement in
oracle/retto_oracle.cpp; see its header: PARITY UNPINNED).  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libretto_oracle.so")


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "retto_oracle.cpp")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        _lib.orc_box_score_fast.restype = C.c_float
    return _lib


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


u8p = lambda a: _p(a, C.c_uint8)
f32p = lambda a: _p(a, C.c_float)
i32p = lambda a: _p(a, C.c_int)
f64p = lambda a: _p(a, C.c_double)


def resize_both_plan(h, w, max_side=2000, min_side=30):
    out = np.zeros(4, np.int32)
    n = lib().orc_resize_both_plan(h, w, max_side, min_side, i32p(out))
    return [(int(out[2 * i]), int(out[2 * i + 1])) for i in range(n)]


def resize_either_dims(h, w, limit_type=0, limit_len=736):
    rh, rw = C.c_int(), C.c_int()
    lib().orc_resize_either_dims(h, w, limit_type, limit_len, C.byref(rh), C.byref(rw))
    return rh.value, rw.value


def thumbnail(img: np.ndarray, nh: int, nw: int) -> np.ndarray:
    img = np.ascontiguousarray(img, np.uint8)
    h, w = img.shape[:2]
    out = np.zeros((nh, nw, 3), np.uint8)
    rc = lib().orc_thumbnail(u8p(img), h, w, u8p(out), nh, nw)
    if rc != 0:
        raise RuntimeError("thumbnail: reference would panic (out-of-bounds sample)")
    return out


def resize_both(img, max_side=2000, min_side=30):
    for nh, nw in resize_both_plan(img.shape[0], img.shape[1], max_side, min_side):
        img = thumbnail(img, nh, nw)
    return img


def det_preprocess(img, limit_type=0, limit_len=736, scale=1.0 / 255.0, mean=(0.5, 0.5, 0.5), std=(0.5, 0.5, 0.5)):
    """a3: returns f32 [1,3,H,W]."""
    rh, rw = resize_either_dims(img.shape[0], img.shape[1], limit_type, limit_len)
    r = thumbnail(img, rh, rw)
    out = np.zeros((1, 3, rh, rw), np.float32)
    m = np.asarray(mean, np.float32); s = np.asarray(std, np.float32)
    lib().orc_det_normalize(u8p(r), rh, rw, C.c_float(np.float32(scale)), f32p(m), f32p(s), f32p(out))
    return out


def threshold_dilate(pred, thresh=0.3, dilate=True):
    pred = np.ascontiguousarray(pred, np.float32)
    h, w = pred.shape
    m = np.zeros((h, w), np.uint8)
    lib().orc_threshold_dilate(f32p(pred), h, w, C.c_float(thresh), int(dilate), u8p(m))
    return m


def find_contours(mask):
    mask = np.ascontiguousarray(mask, np.uint8)
    h, w = mask.shape
    n = lib().orc_find_contours(u8p(mask), h, w, -1, None, 0, None, None)
    res = []
    cap = 4 * (h * w + 16)
    buf = np.zeros(2 * cap, np.int32)
    for k in range(n):
        npts, bt = C.c_int(), C.c_int()
        lib().orc_find_contours(u8p(mask), h, w, k, i32p(buf), cap, C.byref(npts), C.byref(bt))
        res.append((buf[:2 * npts.value].reshape(-1, 2).copy(), bt.value))
    return res


def min_area_rect(pts):
    pts = np.ascontiguousarray(pts, np.float64).reshape(-1, 2)
    out = np.zeros(8, np.float64)
    lib().orc_min_area_rect(f64p(pts), len(pts), f64p(out))
    return out.reshape(4, 2)


def box_score_fast(pred, box_i32):
    pred = np.ascontiguousarray(pred, np.float32)
    b = np.ascontiguousarray(box_i32, np.int32).reshape(8)
    return float(lib().orc_box_score_fast(f32p(pred), pred.shape[0], pred.shape[1], i32p(b)))


def unclip(box_i32, ratio=1.6):
    b = np.ascontiguousarray(box_i32, np.int32).reshape(8)
    out = np.zeros(2 * 4096, np.float32)
    n = lib().orc_unclip(i32p(b), C.c_float(ratio), f32p(out), 4096)
    return out[:2 * n].reshape(-1, 2).copy()


def unclip_distance(box_i32, ratio=1.6):
    b = np.ascontiguousarray(np.asarray(box_i32, np.int32).reshape(8))
    lib().orc_unclip_distance.restype = C.c_float
    return np.float32(lib().orc_unclip_distance(i32p(b), C.c_float(ratio)))


def crop_projection(box):
    """(forward, inverse) 3x3 f32 matrices of the projection get_crop_img warps with."""
    b = np.ascontiguousarray(np.asarray(box, np.float32).reshape(8))
    t, inv = np.zeros(9, np.float32), np.zeros(9, np.float32)
    if lib().orc_crop_projection(f32p(b), f32p(t), f32p(inv)) != 0:
        raise ValueError("singular homography")
    return t.reshape(3, 3), inv.reshape(3, 3)


def det_postprocess(pred, ori_h, ori_w, thresh=0.3, box_thresh=0.5, unclip_ratio=1.6, min_size=3, dilate=True,
                    max_out=65536):
    """a5: returns (boxes [n,4,2] f32, scores [n] f32)."""
    pred = np.ascontiguousarray(pred, np.float32)
    h, w = pred.shape
    boxes = np.zeros((max_out, 8), np.float32); scores = np.zeros(max_out, np.float32)
    n = lib().orc_det_postprocess(f32p(pred), h, w, ori_h, ori_w, C.c_float(thresh), C.c_float(box_thresh),
                                  C.c_float(unclip_ratio), min_size, int(dilate), f32p(boxes), f32p(scores), max_out)
    if n > max_out:
        raise RuntimeError("det_postprocess: more boxes than max_out")
    return boxes[:n].reshape(n, 4, 2).copy(), scores[:n].copy()


def scale_and_clip(box, bw, bh, ow, oh):
    b = np.ascontiguousarray(box, np.float32).reshape(8).copy()
    lib().orc_scale_and_clip(f32p(b), C.c_double(bw), C.c_double(bh), C.c_double(ow), C.c_double(oh))
    return b.reshape(4, 2)


def crop_dims(box):
    b = np.ascontiguousarray(box, np.float32).reshape(8)
    w, h, r = C.c_int(), C.c_int(), C.c_int()
    fw, fh = C.c_float(), C.c_float()
    lib().orc_crop_dims(f32p(b), C.byref(w), C.byref(h), C.byref(r), C.byref(fw), C.byref(fh))
    return w.value, h.value, bool(r.value)


def get_crop_img(img, box):
    img = np.ascontiguousarray(img, np.uint8)
    b = np.ascontiguousarray(box, np.float32).reshape(8)
    w, h, _ = crop_dims(b)
    out = np.zeros((h, w, 3), np.uint8)
    rc = lib().orc_get_crop_img(u8p(img), img.shape[0], img.shape[1], f32p(b), u8p(out))
    if rc != 0:
        raise RuntimeError("get_crop_img: singular homography (reference unwrap() panics)")
    return out


def rotate180(img):
    out = np.ascontiguousarray(img, np.uint8).copy()
    lib().orc_rotate180(u8p(out), out.shape[0], out.shape[1])
    return out


def resize_norm_image(crop, ori_h, ori_w, img_h=48, img_w=320, max_wh_ratio=0.0):
    """a8/a10: returns f32 [3, img_h, W]."""
    crop = np.ascontiguousarray(crop, np.uint8)
    W = lib().orc_resize_norm_width(img_h, img_w, C.c_float(max_wh_ratio))
    out = np.zeros((3, img_h, W), np.float32)
    rc = lib().orc_resize_norm_image(u8p(crop), crop.shape[0], crop.shape[1], ori_h, ori_w, img_h, img_w,
                                     C.c_float(max_wh_ratio), f32p(out))
    if rc != 0:
        raise RuntimeError("resize_norm_image: reference would panic")
    return out


def ctc_decode(probs):
    """a12: returns (idx [n,T], prob [n,T], tokens list[np.ndarray], scores [n])."""
    probs = np.ascontiguousarray(probs, np.float32)
    n, T, Cc = probs.shape
    idx = np.zeros((n, T), np.int32); pr = np.zeros((n, T), np.float32)
    tok = np.zeros((n, T), np.int32); tn = np.zeros(n, np.int32); sc = np.zeros(n, np.float32)
    lib().orc_ctc_decode(f32p(probs), n, T, Cc, i32p(idx), f32p(pr), i32p(tok), i32p(tn), f32p(sc))
    return idx, pr, [tok[i, :tn[i]].copy() for i in range(n)], sc


def cls_postprocess(probs):
    probs = np.ascontiguousarray(probs, np.float32)
    n, Cc = probs.shape
    idx = np.zeros(n, np.int32); sc = np.zeros(n, np.float32)
    lib().orc_cls_postprocess(f32p(probs), n, Cc, i32p(idx), f32p(sc))
    return idx, sc
