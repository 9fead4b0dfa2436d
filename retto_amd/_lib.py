"""ctypes binding of libretto_hip.so (include/retto_hip.h).

There is no fallback: if the shared library is missing or cannot be loaded this
module raises, and every compute entry point fails with RT_ERR_BACKEND when no
gfx950 device is visible.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libretto_hip.so")

RT_OK = 0
RT_MEM_HOST, RT_MEM_DEVICE, RT_MEM_HOST_MAPS_DEVICE = 0, 1, 2
RT_MAX_INFLIGHT = 8   # include/retto_hip.h
RT_DTYPE_F32, RT_DTYPE_F16 = 0, 1
STATUS_NAMES = {0: "OK", 1: "IOError", 2: "ImageError", 3: "ShapeError", 4: "BackendError", 5: "Utf8Error",
                7: "ModelNotFoundError", 8: "InvalidArgument", 9: "CapacityError"}


class ModelSource(C.Structure):
    _fields_ = [("path", C.c_char_p), ("data", C.c_void_p), ("len", C.c_size_t)]


class Config(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32), ("device_id", C.c_int32),
        ("det", ModelSource), ("cls", ModelSource), ("rec", ModelSource), ("dict", ModelSource),
        ("max_side_len", C.c_int32), ("min_side_len", C.c_int32),
        ("det_limit_side_len", C.c_int32), ("det_limit_type", C.c_int32),
        ("det_mean", C.c_float * 3), ("det_std", C.c_float * 3), ("det_scale", C.c_float),
        ("det_thresh", C.c_float), ("det_box_thresh", C.c_float), ("det_unclip_ratio", C.c_float),
        ("det_min_mini_box_size", C.c_int32), ("det_dilation", C.c_int32),
        ("cls_image_shape", C.c_int32 * 3), ("cls_batch_num", C.c_int32), ("cls_thresh", C.c_float),
        ("rec_image_shape", C.c_int32 * 3), ("rec_batch_num", C.c_int32),
        ("max_boxes_per_page", C.c_int32), ("det_sub_batch", C.c_int32), ("lanes", C.c_int32), ("dtype", C.c_int32),
    ]


# every symbol include/retto_hip.h declares (tests check that each is exported)
EXPORTS = [
    "rt_config_default", "rt_create", "rt_destroy", "rt_last_error", "rt_version",
    "rt_det", "rt_cls", "rt_rec", "rt_rec_ragged", "rt_rec_classes", "rt_model_info",
    "rt_resize_both_dims", "rt_resize_both", "rt_det_input_dims", "rt_det_preprocess", "rt_det_postprocess",
    "rt_crop_dims", "rt_crop_images", "rt_scale_and_clip", "rt_resize_norm_width", "rt_resize_norm_image",
    "rt_ctc_decode",
    "rt_run_batch", "rt_run_batch_stream", "rt_submit_batch", "rt_wait_batch", "rt_host_cpu_budget", "rt_results_free", "rt_results_pages", "rt_results_count", "rt_results_boxes",
    "rt_results_det_scores", "rt_results_cls_labels", "rt_results_cls_scores", "rt_results_rec_scores",
    "rt_results_rec_tokens", "rt_results_rec_text", "rt_results_det_checksum", "rt_results_json",
    "rt_device_malloc", "rt_device_free", "rt_memcpy_h2d", "rt_memcpy_d2h", "rt_synchronize",
    "rt_set_lanes", "rt_profile_enable", "rt_profile_get",
    "rt_onnx_to_rtwb", "rt_buffer_free", "rt_model_manifest", "rt_decode_image", "rt_run_encoded_batch",
    "rt_debug_set_variants", "rt_bench_gemm", "rt_bench_gemm_err", "rt_bench_lc", "rt_debug_conv16", "rt_parse_dictionary", "rt_format_f32", "rt_rccl_unique_id", "rt_broadcast_blobs",
]

STAGE_CALLBACK = C.CFUNCTYPE(None, C.c_void_p, C.c_int, C.c_int, C.c_char_p)  # rt_stage_callback

_lib = None


def source_digest() -> str:
    """sha256 over the kernel / host sources of libretto_hip (retto_amd/csrc/*.hip, *.cpp, *.h and the public header): what a
    committed profile (profiles/pmc_traffic.json) was collected on.  Content-based, so it is the same in a git checkout and in
    the snapshot a GPU box receives (which has no .git)."""
    import glob
    import hashlib
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(_HERE, "csrc", "*.hip")) + glob.glob(os.path.join(_HERE, "csrc", "*.cpp")) +
                   glob.glob(os.path.join(_HERE, "csrc", "*.h")) + [os.path.join(os.path.dirname(_HERE), "include", "retto_hip.h")])
    for f in files:
        h.update(os.path.basename(f).encode()); h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def load():
    """Loads the HIP library; raises (never falls back) when it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `make -C retto_amd/csrc` (python -c 'import __graft_entry__ as g; "
            "g.build()'). retto_amd has no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    P = C.POINTER
    lib.rt_last_error.restype = C.c_char_p
    lib.rt_last_error.argtypes = [C.c_void_p]
    lib.rt_version.restype = C.c_char_p
    lib.rt_create.argtypes = [P(Config), P(C.c_void_p)]
    lib.rt_destroy.argtypes = [C.c_void_p]
    lib.rt_destroy.restype = None
    lib.rt_results_free.argtypes = [C.c_void_p]
    lib.rt_results_free.restype = None
    lib.rt_results_boxes.restype = P(C.c_float)
    lib.rt_results_det_scores.restype = P(C.c_float)
    lib.rt_results_cls_labels.restype = P(C.c_uint16)
    lib.rt_results_cls_scores.restype = P(C.c_float)
    lib.rt_results_rec_scores.restype = P(C.c_float)
    lib.rt_results_rec_text.restype = C.c_char_p
    lib.rt_results_json.restype = C.c_char_p
    lib.rt_results_det_checksum.restype = C.c_double
    for name in ("rt_results_boxes", "rt_results_det_scores", "rt_results_cls_labels", "rt_results_cls_scores",
                 "rt_results_rec_scores", "rt_results_count"):
        getattr(lib, name).argtypes = [C.c_void_p, C.c_int]
    lib.rt_results_pages.argtypes = [C.c_void_p]
    lib.rt_results_det_checksum.argtypes = [C.c_void_p]
    lib.rt_results_rec_tokens.argtypes = [C.c_void_p, C.c_int, C.c_int, P(P(C.c_int32))]
    lib.rt_results_rec_text.argtypes = [C.c_void_p, C.c_int, C.c_int]
    lib.rt_results_json.argtypes = [C.c_void_p, C.c_int, C.c_int]
    lib.rt_scale_and_clip.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_double, C.c_double, C.c_double]
    lib.rt_resize_norm_width.argtypes = [C.c_int, C.c_int, C.c_float]
    lib.rt_device_malloc.argtypes = [C.c_void_p, C.c_size_t, P(C.c_void_p)]
    lib.rt_device_free.argtypes = [C.c_void_p, C.c_void_p]
    lib.rt_memcpy_h2d.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
    lib.rt_memcpy_d2h.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
    lib.rt_synchronize.argtypes = [C.c_void_p]
    lib.rt_profile_enable.argtypes = [C.c_void_p, C.c_int]
    lib.rt_set_lanes.argtypes = [C.c_void_p, C.c_int]
    lib.rt_det.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
    lib.rt_cls.argtypes = lib.rt_det.argtypes
    lib.rt_rec.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, P(C.c_int)]
    lib.rt_rec_ragged.argtypes = [C.c_void_p, C.c_void_p, C.c_int, P(C.c_int), C.c_void_p, P(C.c_int)]
    lib.rt_rec_classes.argtypes = [C.c_void_p]
    lib.rt_resize_both_dims.argtypes = [C.c_void_p, C.c_int, C.c_int, P(C.c_int), P(C.c_int)]
    lib.rt_det_input_dims.argtypes = lib.rt_resize_both_dims.argtypes
    lib.rt_resize_both.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int]
    lib.rt_det_preprocess.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    lib.rt_det_postprocess.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                       C.c_void_p, C.c_int, P(C.c_int)]
    lib.rt_crop_dims.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    lib.rt_crop_images.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p,
                                   C.c_size_t]
    lib.rt_resize_norm_image.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                         C.c_float, C.c_void_p]
    lib.rt_ctc_decode.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                  C.c_void_p, C.c_void_p, C.c_void_p]
    lib.rt_run_batch.argtypes = [C.c_void_p, P(C.c_void_p), P(C.c_int), P(C.c_int), C.c_int, C.c_int, P(C.c_void_p),
                                 P(C.c_void_p)]
    lib.rt_submit_batch.argtypes = [C.c_void_p, P(C.c_void_p), P(C.c_int), P(C.c_int), C.c_int, C.c_int, P(C.c_void_p), P(C.c_void_p)]
    lib.rt_wait_batch.argtypes = [C.c_void_p, C.c_void_p, P(C.c_void_p)]
    lib.rt_run_batch_stream.argtypes = [C.c_void_p, P(C.c_void_p), P(C.c_int), P(C.c_int), C.c_int, C.c_int, P(C.c_void_p),
                                        STAGE_CALLBACK, C.c_void_p, P(C.c_void_p)]
    lib.rt_profile_get.argtypes = [C.c_void_p, P(P(C.c_char_p)), P(P(C.c_float)), P(P(C.c_int)), P(C.c_int)]
    lib.rt_onnx_to_rtwb.argtypes = [C.c_int, C.c_void_p, C.c_size_t, P(C.c_void_p), P(C.c_size_t), C.c_char_p, C.c_size_t]
    lib.rt_buffer_free.argtypes = [C.c_void_p]
    lib.rt_buffer_free.restype = None
    lib.rt_model_manifest.argtypes = [C.c_int, C.c_char_p, C.c_size_t]
    lib.rt_decode_image.argtypes = [C.c_char_p, C.c_size_t, P(C.c_void_p), P(C.c_int), P(C.c_int), C.c_char_p, C.c_size_t]
    lib.rt_run_encoded_batch.argtypes = [C.c_void_p, P(C.c_char_p), P(C.c_size_t), C.c_int, C.c_void_p, C.c_void_p, P(C.c_void_p)]
    lib.rt_model_manifest.restype = C.c_size_t
    _lib = lib
    return lib
