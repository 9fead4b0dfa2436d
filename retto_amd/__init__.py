"""retto_amd -- Python host-side mirror of retto-core's worker / session interface on
top of libretto_hip.so (MI355X / gfx950).

Names follow the reference (paths relative to /root/reference/retto-core/src):
    RettoWorkerModelSource  worker.rs:18-27        RettoHipWorker        worker.rs:69-98 (+ ort_worker.rs)
    RettoSessionConfig      session.rs:17-40       RettoSession.run      session.rs:108-131
    RettoWorkerResult       session.rs:44-48       RettoSession.run_stream  session.rs:133-143
    Point / PointBox        points.rs:16-68        Det/Cls/RecProcessorResult  processor/*.rs
Everything is computed by the HIP library; nothing here falls back to the CPU.
"""
from __future__ import annotations

import ctypes as C
import io
from dataclasses import dataclass, field
from typing import Callable, List, Optional, Sequence, Union

import numpy as np

from . import _lib
from ._lib import Config, ModelSource, RT_MEM_DEVICE, RT_MEM_HOST, RT_MEM_HOST_MAPS_DEVICE, RT_MAX_INFLIGHT


# ---- errors (error.rs:2-21) ------------------------------------------------------------
class RettoError(Exception):
    code = 4


class IOError_(RettoError): code = 1
class ImageError(RettoError): code = 2
class ShapeError(RettoError): code = 3
class BackendError(RettoError): code = 4
class Utf8Error(RettoError): code = 5
class ModelNotFoundError(RettoError): code = 7
class InvalidArgument(RettoError): code = 8
class CapacityError(RettoError): code = 9


_ERRS = {c.code: c for c in (IOError_, ImageError, ShapeError, BackendError, Utf8Error, ModelNotFoundError,
                             InvalidArgument, CapacityError)}


def _check(rc: int, handle) -> None:
    if rc != 0:
        msg = _lib.load().rt_last_error(handle)
        raise _ERRS.get(rc, RettoError)((msg or b"").decode("utf-8", "replace"))


# ---- model sources (worker.rs:18-27) -----------------------------------------------------
@dataclass
class RettoWorkerModelSource:
    path: Optional[str] = None
    blob: Optional[bytes] = None

    @staticmethod
    def Path(p: str) -> "RettoWorkerModelSource":
        return RettoWorkerModelSource(path=p)

    @staticmethod
    def Blob(b: bytes) -> "RettoWorkerModelSource":
        return RettoWorkerModelSource(blob=bytes(b))


@dataclass
class RettoWorkerModelProvider:  # worker.rs:61-65
    det: RettoWorkerModelSource
    rec: RettoWorkerModelSource
    cls: RettoWorkerModelSource


@dataclass
class RettoHipWorkerConfig:  # RettoOrtWorkerConfig analogue (ort_worker.rs:53-56)
    device: int = 0
    models: Optional[RettoWorkerModelProvider] = None


@dataclass
class DetProcessorConfig:  # det_processor.rs:44-93
    limit_side_len: int = 736
    limit_type: str = "Min"
    mean: Sequence[float] = (0.5, 0.5, 0.5)
    std: Sequence[float] = (0.5, 0.5, 0.5)
    scale: float = float(np.float32(1.0) / np.float32(255.0))
    threch: float = 0.3
    box_thresh: float = 0.5
    max_candidates: int = 1000  # declared but never read by the reference either
    unclip_ratio: float = 1.6
    use_dilation: bool = True
    min_mini_box_size: int = 3
    dilation_kernel: Optional[np.ndarray] = field(default_factory=lambda: np.ones((2, 2), np.uint64))


@dataclass
class ClsProcessorConfig:  # cls_processor.rs:14-36
    image_shape: Sequence[int] = (3, 48, 192)
    batch_num: int = 6
    thresh: float = 0.9
    label: Sequence[int] = (0, 180)


@dataclass
class RecProcessorConfig:  # rec_processor.rs:102-136
    character_source: Optional[RettoWorkerModelSource] = None
    image_shape: Sequence[int] = (3, 48, 320)
    batch_num: int = 6


@dataclass
class RettoSessionConfig:  # session.rs:17-40
    worker_config: RettoHipWorkerConfig = field(default_factory=RettoHipWorkerConfig)
    max_side_len: int = 2000
    min_side_len: int = 30
    det_processor_config: DetProcessorConfig = field(default_factory=DetProcessorConfig)
    cls_processor_config: ClsProcessorConfig = field(default_factory=ClsProcessorConfig)
    rec_processor_config: RecProcessorConfig = field(default_factory=RecProcessorConfig)
    max_boxes_per_page: int = 0
    det_sub_batch: int = 0
    lanes: int = 0
    dtype: str = "f32"   # "f32" (the reference's arithmetic) or "f16" (fp16 storage / MFMA, fp32 accumulation)


# ---- result types (points.rs, processor/*.rs) ---------------------------------------------
@dataclass
class Point:
    x: float
    y: float


@dataclass
class PointBox:
    inner: List[Point]  # clockwise from top-left

    def tl(self): return self.inner[0]
    def tr(self): return self.inner[1]
    def br(self): return self.inner[2]
    def bl(self): return self.inner[3]

    def as_array(self) -> np.ndarray:
        return np.array([[p.x, p.y] for p in self.inner], np.float32)


@dataclass
class DetProcessorInnerResult:
    boxes: PointBox
    score: float


@dataclass
class ClsPostProcessLabel:
    label: int
    score: float


@dataclass
class ClsProcessorSingleResult:
    label: ClsPostProcessLabel


@dataclass
class RecProcessorSingleResult:
    text: str
    score: float
    tokens: np.ndarray = None  # kept CTC token ids (not in the reference struct; exposed for parity checks)


@dataclass
class RettoWorkerResult:  # session.rs:44-48
    det_result: List[DetProcessorInnerResult]
    cls_result: List[ClsProcessorSingleResult]
    rec_result: List[RecProcessorSingleResult]


def _src(s: Optional[RettoWorkerModelSource], keep: list) -> ModelSource:
    m = ModelSource()
    if s is None:
        return m
    if s.path is not None:
        b = s.path.encode("utf-8"); keep.append(b); m.path = b
    elif s.blob is not None:
        buf = C.create_string_buffer(s.blob, len(s.blob)); keep.append(buf)
        m.data = C.cast(buf, C.c_void_p); m.len = len(s.blob)
    return m


def _as_f32(a, ndim):
    a = np.ascontiguousarray(a, np.float32)  # as_standard_layout (ort_worker.rs:191,202,213)
    if a.ndim != ndim:
        raise ShapeError(f"expected {ndim}-d array, got {a.ndim}-d")
    return a


class _Handle:
    """Owns one rt_session (one GPU)."""

    def __init__(self, cfg: RettoSessionConfig):
        lib = _lib.load()
        c = Config()
        lib.rt_config_default(C.byref(c))
        keep: list = []
        wc = cfg.worker_config
        if wc.models is None:
            raise ModelNotFoundError("no model provider configured (worker_config.models)")
        c.device_id = wc.device
        c.det = _src(wc.models.det, keep); c.cls = _src(wc.models.cls, keep); c.rec = _src(wc.models.rec, keep)
        c.dict = _src(cfg.rec_processor_config.character_source, keep)
        c.max_side_len, c.min_side_len = cfg.max_side_len, cfg.min_side_len
        d = cfg.det_processor_config
        c.det_limit_side_len = d.limit_side_len
        c.det_limit_type = 0 if d.limit_type == "Min" else 1
        for i in range(3):
            c.det_mean[i] = d.mean[i]; c.det_std[i] = d.std[i]
        c.det_scale = d.scale; c.det_thresh = d.threch; c.det_box_thresh = d.box_thresh
        c.det_unclip_ratio = d.unclip_ratio; c.det_min_mini_box_size = d.min_mini_box_size
        if d.dilation_kernel is None:
            c.det_dilation = 0
        else:
            k = np.asarray(d.dilation_kernel)
            if k.shape != (2, 2) or not np.all(k != 0):
                raise InvalidArgument("only the reference's default 2x2 all-ones dilation kernel (or None) is supported")
            c.det_dilation = 1
        cl, rc = cfg.cls_processor_config, cfg.rec_processor_config
        for i in range(3):
            c.cls_image_shape[i] = cl.image_shape[i]; c.rec_image_shape[i] = rc.image_shape[i]
        c.cls_batch_num, c.cls_thresh = cl.batch_num, cl.thresh
        c.rec_batch_num = rc.batch_num
        if tuple(cl.label) != (0, 180):
            raise InvalidArgument("cls label set other than [0, 180] is not supported")
        c.max_boxes_per_page = cfg.max_boxes_per_page; c.det_sub_batch = cfg.det_sub_batch; c.lanes = cfg.lanes
        if cfg.dtype not in ("f32", "f16"):
            raise InvalidArgument("dtype must be 'f32' or 'f16'")
        c.dtype = 1 if cfg.dtype == "f16" else 0
        h = C.c_void_p()
        _check(lib.rt_create(C.byref(c), C.byref(h)), None)
        self.lib, self.h = lib, h

    def close(self):
        if getattr(self, "h", None):
            self.lib.rt_destroy(self.h); self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class RettoHipWorker:
    """``impl RettoWorker + RettoInnerWorker`` for the HIP backend (worker.rs:69-98)."""

    def __init__(self, cfg: Union[RettoSessionConfig, RettoHipWorkerConfig], _handle: Optional[_Handle] = None):
        if _handle is None:
            if isinstance(cfg, RettoHipWorkerConfig):
                raise InvalidArgument("construct RettoHipWorker from a RettoSessionConfig (the dictionary is needed)")
            _handle = _Handle(cfg)
        self._hd = _handle

    def init(self) -> None:  # worker.rs:97 (no-op like ort_worker.rs:183-185)
        return None

    def det(self, x: np.ndarray) -> np.ndarray:
        x = _as_f32(x, 4); n, c, h, w = x.shape
        out = np.empty((n, 1, h, w), np.float32)
        _check(self._hd.lib.rt_det(self._hd.h, x.ctypes.data, n, c, h, w, out.ctypes.data), self._hd.h)
        return out

    def cls(self, x: np.ndarray) -> np.ndarray:
        x = _as_f32(x, 4); n, c, h, w = x.shape
        out = np.empty((n, 2), np.float32)
        _check(self._hd.lib.rt_cls(self._hd.h, x.ctypes.data, n, c, h, w, out.ctypes.data), self._hd.h)
        return out

    def rec(self, x: np.ndarray) -> np.ndarray:
        x = _as_f32(x, 4); n, c, h, w = x.shape
        t = C.c_int()
        _check(self._hd.lib.rt_rec(self._hd.h, x.ctypes.data, n, c, h, w, None, C.byref(t)), self._hd.h)
        out = np.empty((n, t.value, self._hd.lib.rt_rec_classes(self._hd.h)), np.float32)
        _check(self._hd.lib.rt_rec(self._hd.h, x.ctypes.data, n, c, h, w, out.ctypes.data, C.byref(t)), self._hd.h)
        return out


    def rec_ragged(self, lines) -> list:
        """rt_rec_ragged: lines = [3,48,w_i] arrays of different widths, ONE launch series (the form rt_run_batch's rec groups
        take); returns the per-line [T_i, classes] probabilities."""
        lines = [_as_f32(l, 3) for l in lines]
        n = len(lines)
        widths = (C.c_int * n)(*[l.shape[2] for l in lines])
        ts = (C.c_int * n)()
        flat = np.ascontiguousarray(np.concatenate([l.reshape(-1) for l in lines]))
        _check(self._hd.lib.rt_rec_ragged(self._hd.h, flat.ctypes.data, n, widths, None, ts), self._hd.h)
        ncls = self._hd.lib.rt_rec_classes(self._hd.h)
        out = np.empty((sum(ts), ncls), np.float32)
        _check(self._hd.lib.rt_rec_ragged(self._hd.h, flat.ctypes.data, n, widths, out.ctypes.data, ts), self._hd.h)
        res, o = [], 0
        for t in ts:
            res.append(out[o:o + t]); o += t
        return res

def decode_image(data: bytes) -> np.ndarray:
    """ImageHelper::new_from_raw_img_flow (image_helper.rs:34-44): encoded bytes -> RGB8 [H,W,3] through the
    library's host decoder (rt_decode_image: PNG, JPEG, PNM, BMP); raises ImageError otherwise."""
    lib = _lib.load()
    data = bytes(data)
    out = C.c_void_p(); h = C.c_int(); w = C.c_int()
    err = C.create_string_buffer(512)
    rc = lib.rt_decode_image(data, len(data), C.byref(out), C.byref(h), C.byref(w), err, len(err))
    if rc != 0:
        raise _ERRS.get(rc, RettoError)(err.value.decode("utf-8", "replace"))
    try:
        return np.ctypeslib.as_array(C.cast(out, C.POINTER(C.c_uint8)), (h.value, w.value, 3)).copy()
    finally:
        lib.rt_buffer_free(out)


class RettoSession:
    """RettoSession<RettoHipWorker> (session.rs:58-144), batched."""

    def __init__(self, cfg: RettoSessionConfig):
        self.config = cfg
        self._hd = _Handle(cfg)
        self.worker = RettoHipWorker(cfg, self._hd)
        self.worker.init()

    # -- stage functions ------------------------------------------------------------------
    def resize_both(self, img: np.ndarray) -> np.ndarray:
        img = np.ascontiguousarray(img, np.uint8); h, w = img.shape[:2]
        oh, ow = C.c_int(), C.c_int()
        _check(self._hd.lib.rt_resize_both_dims(self._hd.h, h, w, C.byref(oh), C.byref(ow)), self._hd.h)
        out = np.empty((oh.value, ow.value, 3), np.uint8)
        _check(self._hd.lib.rt_resize_both(self._hd.h, img.ctypes.data, h, w, out.ctypes.data, oh.value, ow.value), self._hd.h)
        return out

    def det_preprocess(self, img: np.ndarray) -> np.ndarray:
        img = np.ascontiguousarray(img, np.uint8); h, w = img.shape[:2]
        oh, ow = C.c_int(), C.c_int()
        _check(self._hd.lib.rt_det_input_dims(self._hd.h, h, w, C.byref(oh), C.byref(ow)), self._hd.h)
        out = np.empty((1, 3, oh.value, ow.value), np.float32)
        _check(self._hd.lib.rt_det_preprocess(self._hd.h, img.ctypes.data, h, w, out.ctypes.data), self._hd.h)
        return out

    def det_postprocess(self, pred: np.ndarray, ori_h: int, ori_w: int, max_out: int = 65536):
        pred = np.ascontiguousarray(pred, np.float32); h, w = pred.shape
        boxes = np.zeros((max_out, 8), np.float32); scores = np.zeros(max_out, np.float32); n = C.c_int()
        _check(self._hd.lib.rt_det_postprocess(self._hd.h, pred.ctypes.data, h, w, ori_h, ori_w, boxes.ctypes.data,
                                               scores.ctypes.data, max_out, C.byref(n)), self._hd.h)
        return boxes[:n.value].reshape(-1, 4, 2).copy(), scores[:n.value].copy()

    def crop_images(self, img: np.ndarray, boxes: np.ndarray) -> List[np.ndarray]:
        img = np.ascontiguousarray(img, np.uint8); h, w = img.shape[:2]
        b = np.ascontiguousarray(boxes, np.float32).reshape(-1, 8); n = len(b)
        ws = np.zeros(max(n, 1), np.int32); hs = np.zeros(max(n, 1), np.int32)
        _check(self._hd.lib.rt_crop_dims(b.ctypes.data, n, ws.ctypes.data, hs.ctypes.data), self._hd.h)
        total = int(sum(int(ws[i]) * int(hs[i]) * 3 for i in range(n)))
        out = np.zeros(max(total, 1), np.uint8)
        _check(self._hd.lib.rt_crop_images(self._hd.h, img.ctypes.data, h, w, b.ctypes.data, n, out.ctypes.data, total), self._hd.h)
        res, o = [], 0
        for i in range(n):
            sz = int(ws[i]) * int(hs[i]) * 3
            res.append(out[o:o + sz].reshape(int(hs[i]), int(ws[i]), 3).copy()); o += sz
        return res

    def resize_norm_image(self, crop: np.ndarray, ori_h: int, ori_w: int, img_h=48, img_w=320, max_wh_ratio=0.0):
        crop = np.ascontiguousarray(crop, np.uint8); h, w = crop.shape[:2]
        W = self._hd.lib.rt_resize_norm_width(img_h, img_w, max_wh_ratio)
        out = np.empty((3, img_h, W), np.float32)
        _check(self._hd.lib.rt_resize_norm_image(self._hd.h, crop.ctypes.data, h, w, ori_h, ori_w, img_h, img_w,
                                                 max_wh_ratio, out.ctypes.data), self._hd.h)
        return out

    def ctc_decode(self, probs: np.ndarray):
        probs = np.ascontiguousarray(probs, np.float32); n, t, c = probs.shape
        idx = np.zeros((n, t), np.int32); pr = np.zeros((n, t), np.float32)
        tok = np.zeros((n, t), np.int32); tn = np.zeros(n, np.int32); sc = np.zeros(n, np.float32)
        _check(self._hd.lib.rt_ctc_decode(self._hd.h, probs.ctypes.data, n, t, c, idx.ctypes.data, pr.ctypes.data,
                                          tok.ctypes.data, tn.ctypes.data, sc.ctypes.data), self._hd.h)
        return idx, pr, [tok[i, :tn[i]].copy() for i in range(n)], sc

    # -- pipeline ---------------------------------------------------------------------------
    def run_batch_raw(self, pages, hs, ws, mem=RT_MEM_HOST, det_map_override=None, submit=False):
        """pages: sequence of host arrays or device pointers (ints). Returns an opaque results handle."""
        lib, h = self._hd.lib, self._hd.h
        n = len(pages)
        arr_p = (C.c_void_p * max(n, 1))(); arr_h = (C.c_int * max(n, 1))(); arr_w = (C.c_int * max(n, 1))()
        keep = []
        for i, p in enumerate(pages):
            if isinstance(p, np.ndarray):
                p = np.ascontiguousarray(p, np.uint8); keep.append(p); arr_p[i] = p.ctypes.data
            else:
                arr_p[i] = int(p)
            arr_h[i], arr_w[i] = int(hs[i]), int(ws[i])
        ov = None
        if det_map_override is not None:
            ov = (C.c_void_p * max(n, 1))()
            for i, m in enumerate(det_map_override):
                if m is None:
                    ov[i] = None
                elif isinstance(m, np.ndarray):
                    m = np.ascontiguousarray(m, np.float32); keep.append(m); ov[i] = m.ctypes.data
                else:
                    ov[i] = int(m)
        out = C.c_void_p()
        if submit:   # rt_submit_batch: returns (ticket, the arrays that must stay alive until wait_batch)
            _check(lib.rt_submit_batch(h, arr_p, arr_h, arr_w, n, mem, ov, C.byref(out)), h)
            return out, keep
        _check(lib.rt_run_batch(h, arr_p, arr_h, arr_w, n, mem, ov, C.byref(out)), h)
        return out

    def submit_batch_raw(self, pages, hs, ws, mem=RT_MEM_HOST, det_map_override=None):
        """rt_submit_batch (the counterpart of RettoSession::run_stream's worker thread, session.rs:108-143): the batch is
        split over the session's lanes and this returns at once with a ticket; wait_batch_raw(ticket) gives the results
        handle.  Up to RT_MAX_INFLIGHT (8) batches ahead; nothing else may be called on the session in between."""
        return self.run_batch_raw(pages, hs, ws, mem, det_map_override, submit=True)

    def wait_batch_raw(self, ticket):
        t, _keep = ticket
        out = C.c_void_p()
        _check(self._hd.lib.rt_wait_batch(self._hd.h, t, C.byref(out)), self._hd.h)
        return out

    def _collect(self, r, page: int) -> RettoWorkerResult:
        lib = self._hd.lib
        n = lib.rt_results_count(r, page)
        boxes = np.ctypeslib.as_array(lib.rt_results_boxes(r, page), (n, 8)).copy() if n else np.zeros((0, 8), np.float32)
        ds = np.ctypeslib.as_array(lib.rt_results_det_scores(r, page), (n,)).copy() if n else np.zeros(0, np.float32)
        cl = np.ctypeslib.as_array(lib.rt_results_cls_labels(r, page), (n,)).copy() if n else np.zeros(0, np.uint16)
        cs = np.ctypeslib.as_array(lib.rt_results_cls_scores(r, page), (n,)).copy() if n else np.zeros(0, np.float32)
        rs = np.ctypeslib.as_array(lib.rt_results_rec_scores(r, page), (n,)).copy() if n else np.zeros(0, np.float32)
        det, cls, rec = [], [], []
        for k in range(n):
            det.append(DetProcessorInnerResult(PointBox([Point(float(boxes[k, 2 * q]), float(boxes[k, 2 * q + 1]))
                                                         for q in range(4)]), float(ds[k])))
            cls.append(ClsProcessorSingleResult(ClsPostProcessLabel(int(cl[k]), float(cs[k]))))
            tp = C.POINTER(C.c_int32)()
            nt = lib.rt_results_rec_tokens(r, page, k, C.byref(tp))
            toks = np.ctypeslib.as_array(tp, (nt,)).copy() if nt else np.zeros(0, np.int32)
            rec.append(RecProcessorSingleResult(lib.rt_results_rec_text(r, page, k).decode("utf-8"), float(rs[k]), toks))
        return RettoWorkerResult(det, cls, rec)

    def run_batch(self, pages: Sequence[np.ndarray], det_map_override=None) -> List[RettoWorkerResult]:
        pages = [np.ascontiguousarray(p, np.uint8) for p in pages]
        r = self.run_batch_raw(pages, [p.shape[0] for p in pages], [p.shape[1] for p in pages], RT_MEM_HOST,
                               det_map_override)
        try:
            self.last_det_checksum = self._hd.lib.rt_results_det_checksum(r)
            return [self._collect(r, i) for i in range(len(pages))]
        finally:
            self._hd.lib.rt_results_free(r)

    def run(self, image: Union[bytes, np.ndarray]) -> RettoWorkerResult:
        """session.rs:108-131.  ``image`` is an encoded image (bytes) or an RGB8 array."""
        if isinstance(image, (bytes, bytearray, memoryview)):
            return self.run_encoded_batch([bytes(image)])[0]
        return self.run_batch([image])[0]

    def run_encoded_batch(self, files: Sequence[bytes]) -> List[RettoWorkerResult]:
        """RettoSession::run over a batch of encoded images (rt_run_encoded_batch: decode on host threads, then
        the batch pipeline)."""
        n = len(files)
        files = [bytes(f) for f in files]
        ptrs = (C.c_char_p * n)(*files); lens = (C.c_size_t * n)(*[len(f) for f in files])
        out = C.c_void_p()
        _check(self._hd.lib.rt_run_encoded_batch(self._hd.h, ptrs, lens, n, None, None, C.byref(out)), self._hd.h)
        try:
            return [self._collect(out, i) for i in range(n)]
        finally:
            self._hd.lib.rt_results_free(out)

    def run_stream(self, image, sender: Callable[[str, list], None]) -> None:
        """session.rs:133-143: emits ("Det", ...), ("Cls", ...), ("Rec", ...) in that order.  Det arrives while
        the crops are still being classified / read (rt_run_batch_stream); the payloads are the parsed
        RettoWorkerStageResult JSON of the stage."""
        import json
        page = decode_image(image) if isinstance(image, (bytes, bytearray, memoryview)) else np.ascontiguousarray(image, np.uint8)
        self.run_batch_stream([page], lambda _page, stage, payload: sender(stage, payload))

    def run_batch_stream(self, pages: Sequence[np.ndarray], on_stage: Callable[[int, str, list], None], det_map_override=None) -> None:
        """on_stage(page index, "Det" | "Cls" | "Rec", parsed stage JSON), called from the library's lane threads."""
        import json
        pages = [np.ascontiguousarray(p, np.uint8) for p in pages]
        n = len(pages)
        ptrs = (C.c_void_p * n)(*[p.ctypes.data for p in pages])
        hs = (C.c_int * n)(*[p.shape[0] for p in pages]); ws = (C.c_int * n)(*[p.shape[1] for p in pages])
        maps = None
        if det_map_override is not None:
            keep = [None if m is None else np.ascontiguousarray(m, np.float32) for m in det_map_override]
            maps = (C.c_void_p * n)(*[None if m is None else m.ctypes.data for m in keep])
        errors = []

        def cb(_user, page, stage, js):
            try:
                on_stage(page, ("Det", "Cls", "Rec")[stage], json.loads(js.decode("utf-8")))
            except BaseException as e:  # never unwind through the C frames
                errors.append(e)

        ccb = _lib.STAGE_CALLBACK(cb)
        out = C.c_void_p()
        _check(self._hd.lib.rt_run_batch_stream(self._hd.h, ptrs, hs, ws, n, RT_MEM_HOST, maps, ccb, None, C.byref(out)), self._hd.h)
        self._hd.lib.rt_results_free(out)
        if errors:
            raise errors[0]

    def stage_json(self, page: np.ndarray) -> List[str]:
        """RettoWorkerStageResult JSON strings (Det, Cls, Rec) in retto-wasm's serde shape."""
        page = np.ascontiguousarray(page, np.uint8)
        r = self.run_batch_raw([page], [page.shape[0]], [page.shape[1]])
        try:
            return [self._hd.lib.rt_results_json(r, 0, s).decode("utf-8") for s in range(3)]
        finally:
            self._hd.lib.rt_results_free(r)

    # -- profiling --------------------------------------------------------------------------
    def profile_enable(self, on: bool = True):
        _check(self._hd.lib.rt_profile_enable(self._hd.h, int(on)), self._hd.h)

    def profile_get(self):
        names = C.POINTER(C.c_char_p)(); ms = C.POINTER(C.c_float)(); calls = C.POINTER(C.c_int)(); n = C.c_int()
        _check(self._hd.lib.rt_profile_get(self._hd.h, C.byref(names), C.byref(ms), C.byref(calls), C.byref(n)), self._hd.h)
        return {names[i].decode(): (float(ms[i]), int(calls[i])) for i in range(n.value)}

    def close(self):
        self._hd.close()


MODEL_DET, MODEL_CLS, MODEL_REC, MODEL_SDET, MODEL_SREC = 0, 1, 2, 3, 4   # (DET / REC sources also take the server .onnx files)


def onnx_to_rtwb(which: int, onnx_bytes: bytes) -> bytes:
    """PP-OCRv4 .onnx file -> RTWB blob (rt_onnx_to_rtwb; replaces ort_worker.rs:120-135's model loading).
    RettoSession accepts the .onnx bytes / path directly as well."""
    lib = _lib.load()
    out = C.c_void_p(); n = C.c_size_t(); err = C.create_string_buffer(1024)
    rc = lib.rt_onnx_to_rtwb(which, onnx_bytes, len(onnx_bytes), C.byref(out), C.byref(n), err, len(err))
    if rc != 0:
        raise _ERRS.get(rc, RettoError)(err.value.decode("utf-8", "replace"))
    try:
        return C.string_at(out, n.value)
    finally:
        lib.rt_buffer_free(out)


def model_manifest(which: int) -> str:
    lib = _lib.load()
    n = lib.rt_model_manifest(which, None, 0)
    buf = C.create_string_buffer(n)
    lib.rt_model_manifest(which, buf, n)
    return buf.value.decode()


def synthetic_session_config(seed: int = 0, device: int = 0, server: bool = False, **kw) -> RettoSessionConfig:
    """Session config over seeded synthetic PP-OCRv4-shaped weights (see retto_amd.synth): the mobile graphs, or with
    ``server=True`` the PP-OCRv4 server det / rec graphs (fp16 only: pass ``dtype="f16"``)."""
    from . import synth
    det, cls, rec, dic = synth.synth_server_models(seed) if server else synth.synth_models(seed)
    cfg = RettoSessionConfig(**kw)
    cfg.worker_config = RettoHipWorkerConfig(device=device, models=RettoWorkerModelProvider(
        det=RettoWorkerModelSource.Blob(det), rec=RettoWorkerModelSource.Blob(rec), cls=RettoWorkerModelSource.Blob(cls)))
    cfg.rec_processor_config.character_source = RettoWorkerModelSource.Blob(dic)
    return cfg
