/*
 * retto_hip.h -- C ABI of libretto_hip.so, the MI355X (gfx950) backend for
 * retto-core's OCR hot path.  Plain pointers and sizes only; no C++/torch types.
 *
 * What each entry point replaces in the reference (paths relative to
 * /root/reference):
 *
 *   L1 tensor-level worker  (trait RettoInnerWorker, retto-core/src/worker.rs:69-73;
 *                            ORT implementation retto-core/src/worker/ort_worker.rs:189-220)
 *       rt_det / rt_cls / rt_rec
 *   worker construction     (trait RettoWorker::new/init, worker.rs:91-98; model source
 *                            resolution worker.rs:18-56; ort_worker.rs:120-181)
 *       rt_create / rt_destroy
 *   L2 pipeline             (RettoSession::run / run_stream, retto-core/src/session.rs:75-143)
 *       rt_run_batch (+ rt_results_* accessors); stage order Det -> Cls -> Rec
 *   stage functions, exported so each row of SURVEY.md section 8(a) can be checked
 *   against the oracle on its own:
 *       rt_resize_both        ImageHelper::resize_both            image_helper.rs:106-148   (a2)
 *       rt_det_preprocess     DetProcessor::preprocess            det_processor.rs:256-274  (a3)
 *       rt_det_postprocess    DetProcessor::postprocess           det_processor.rs:279-335  (a5)
 *       rt_crop_images        ImageHelper::get_crop_img           image_helper.rs:223-249   (a6)
 *       rt_scale_and_clip     PointBox::scale_and_clip            points.rs:179-194         (a7)
 *       rt_resize_norm_image  ImageHelper::resize_norm_image      image_helper.rs:176-209   (a8/a10)
 *       rt_ctc_decode         RecProcessor::postprocess + decode  rec_processor.rs:48-97,190-208 (a12)
 *
 * Error convention (retto-core/src/error.rs:2-21): every call returns an rt_status;
 * rt_last_error() gives the message of the last failure on that session (or of the
 * last failed rt_create on this thread when session == NULL).
 *
 * Threading (worker.rs:70-72, session.rs:108,133 take &mut self): one call in flight
 * per session; sessions are independent (one per GPU); a session may be used from a
 * thread other than the one that created it.
 *
 * Ownership: caller-owned inputs stay caller-owned; outputs are written into
 * caller-allocated buffers, except rt_results which is library-owned until
 * rt_results_free().
 *
 * All computation happens on the GPU.  There is no CPU fallback: if no gfx950 device
 * is visible rt_create fails with RT_ERR_BACKEND.
 */
#ifndef RETTO_HIP_H
#define RETTO_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RT_API __attribute__((visibility("default")))

typedef enum rt_status {
  RT_OK = 0,
  RT_ERR_IO = 1,              /* RettoError::IOError */
  RT_ERR_IMAGE = 2,           /* RettoError::ImageError */
  RT_ERR_SHAPE = 3,           /* RettoError::ShapeError */
  RT_ERR_BACKEND = 4,         /* RettoError::OrtError's slot: HIP / device failures */
  RT_ERR_UTF8 = 5,            /* RettoError::Utf8Error */
  RT_ERR_MODEL_NOT_FOUND = 7, /* RettoError::ModelNotFoundError */
  RT_ERR_INVALID = 8,         /* NULL / out-of-range argument */
  RT_ERR_CAPACITY = 9         /* a device-side work list overflowed its configured capacity */
} rt_status;

/* RettoWorkerModelSource::{Path,Blob} (worker.rs:18-27): path != NULL selects Path,
 * otherwise (data,len) is a Blob.  A missing path or an empty blob is
 * RT_ERR_MODEL_NOT_FOUND (worker.rs:33-47).  Model format: RTWB (retto_amd/synth.py). */
typedef struct rt_model_source {
  const char* path;
  const void* data;
  size_t len;
} rt_model_source;

/* RettoSessionConfig + Det/Cls/RecProcessorConfig defaults (session.rs:28-39,
 * det_processor.rs:75-93, cls_processor.rs:27-36, rec_processor.rs:111-136). */
typedef struct rt_config {
  uint32_t struct_size;      /* sizeof(rt_config) of the header the caller was compiled against: set by rt_config_default,
                                checked by rt_create (a host built against an older, shorter struct gets RT_ERR_INVALID instead
                                of having fields read past the end of its struct) */
  int32_t device_id;         /* HIP device ordinal (RettoOrtWorkerDevice::Cuda(id) analogue) */
  rt_model_source det, cls, rec, dict;
  int32_t max_side_len;      /* 2000 */
  int32_t min_side_len;      /* 30 */
  /* det */
  int32_t det_limit_side_len; /* 736 */
  int32_t det_limit_type;     /* 0 = Min (default), 1 = Max */
  float det_mean[3];          /* 0.5 */
  float det_std[3];           /* 0.5 */
  float det_scale;            /* 1/255 */
  float det_thresh;           /* 0.3 */
  float det_box_thresh;       /* 0.5 */
  float det_unclip_ratio;     /* 1.6 */
  int32_t det_min_mini_box_size; /* 3 */
  int32_t det_dilation;       /* 1 = 2x2 ones kernel (default), 0 = none */
  /* cls */
  int32_t cls_image_shape[3]; /* 3,48,192 */
  int32_t cls_batch_num;      /* 6 */
  float cls_thresh;           /* 0.9 */
  /* rec */
  int32_t rec_image_shape[3]; /* 3,48,320 */
  int32_t rec_batch_num;      /* 6 */
  /* backend knobs (no reference counterpart) */
  int32_t max_boxes_per_page; /* capacity of the per-page box list (the reference's list is unbounded); 0 = default 8192, at most 65536 */
  int32_t det_sub_batch;      /* pages per det launch group; 0 = default */
  int32_t lanes;              /* concurrent page streams inside rt_run_batch (1..4); 0 = default 3 */
  int32_t dtype;              /* rt_dtype: arithmetic of the three networks.  RT_DTYPE_F32 (default) = what the reference's
                                 ort-CPU path computes in; RT_DTYPE_F16 = fp16 storage / MFMA, fp32 accumulation
                                 (BASELINE.json config 5).  The PP-OCRv4 server graphs are built in fp16 only. */
} rt_config;
typedef enum rt_dtype { RT_DTYPE_F32 = 0, RT_DTYPE_F16 = 1 } rt_dtype;

typedef struct rt_session rt_session;
typedef struct rt_results rt_results;

RT_API void rt_config_default(rt_config* cfg);
RT_API int rt_create(const rt_config* cfg, rt_session** out);
RT_API void rt_destroy(rt_session* s);
RT_API const char* rt_last_error(const rt_session* s);
RT_API const char* rt_version(void);

/* ---- L1: RettoInnerWorker (host tensors in, host tensors out) ------------------- */
/* det: f32 NCHW [n,3,h,w] (h,w multiples of 32) -> f32 [n,1,h,w] */
RT_API int rt_det(rt_session* s, const float* nchw, int n, int c, int h, int w, float* out);
/* cls: f32 [n,3,48,192] -> f32 [n,2] (softmax) */
RT_API int rt_cls(rt_session* s, const float* nchw, int n, int c, int h, int w, float* out);
/* rec: f32 [n,3,48,w] -> f32 [n,T,6625] (softmax); *t_out = T; call with out == NULL to
 * query T only. */
RT_API int rt_rec(rt_session* s, const float* nchw, int n, int c, int h, int w, float* out, int* t_out);
/* rec over lines of DIFFERENT widths in one launch series: line i is f32 [3,48,widths[i]] (lines concatenated); out = the lines'
 * [T_i,6625] softmax rows concatenated, t_out[i] = T_i (out == NULL: query the T_i only).  No reference counterpart as a worker
 * call (ort_worker.rs:211-220 takes one width per call); it is the tensor-level view of what rt_run_batch does with
 * rec_processor.rs:214-270's batches -- each keeps the width the running max_wh_ratio gives it, all in one ragged launch group --
 * so that the parity tests can compare exactly that form with the oracle, line by line. */
RT_API int rt_rec_ragged(rt_session* s, const float* nchw, int n, const int* widths, float* out, int* t_out);
RT_API int rt_rec_classes(const rt_session* s);
/* "<det arch>/<det dtype> <cls dtype> <rec arch>/<rec dtype>", e.g. "mobile/f32 f32 mobile/f32" or "server/f16 f16 server/f16":
 * which graphs the model sources held and the arithmetic they run in (library-owned string) */
RT_API const char* rt_model_info(const rt_session* s);

/* ---- stage functions (host buffers; computed on the GPU) ------------------------- */
RT_API int rt_resize_both_dims(const rt_session* s, int h, int w, int* out_h, int* out_w);
RT_API int rt_resize_both(rt_session* s, const uint8_t* rgb, int h, int w, uint8_t* out, int out_h, int out_w);
RT_API int rt_det_input_dims(const rt_session* s, int h, int w, int* out_h, int* out_w);
/* a3: u8 HWC RGB (after resize_both) -> f32 [1,3,out_h,out_w] */
RT_API int rt_det_preprocess(rt_session* s, const uint8_t* rgb, int h, int w, float* out_nchw);
/* a5: f32 [h,w] map -> boxes (n x 8 f32: TL,TR,BR,BL x,y in ori_* coordinates) + scores. */
RT_API int rt_det_postprocess(rt_session* s, const float* pred, int h, int w, int ori_h, int ori_w,
                              float* boxes, float* scores, int max_out, int* n_out);
/* a6: crop sizes for n boxes (w,h after the optional rotate270), then the crops packed
 * back to back (RGB8) into out (capacity out_cap bytes). */
RT_API int rt_crop_dims(const float* boxes, int n, int* ws, int* hs);
RT_API int rt_crop_images(rt_session* s, const uint8_t* rgb, int h, int w, const float* boxes, int n,
                          uint8_t* out, size_t out_cap);
RT_API int rt_scale_and_clip(float* boxes, int n, double bitmap_w, double bitmap_h, double ori_w, double ori_h);
/* a8/a10: one crop -> f32 [3,img_h,W]; W = img_w when max_wh_ratio <= 0 else (int)(img_h*max_wh_ratio). */
RT_API int rt_resize_norm_width(int img_h, int img_w, float max_wh_ratio);
RT_API int rt_resize_norm_image(rt_session* s, const uint8_t* crop, int h, int w, int ori_h, int ori_w,
                                int img_h, int img_w, float max_wh_ratio, float* out_chw);
/* a12: probs [n,T,C] -> argmax idx [n,T], max prob [n,T], kept tokens [n,T] (+count), score [n] */
RT_API int rt_ctc_decode(rt_session* s, const float* probs, int n, int t, int c, int32_t* idx, float* prob,
                         int32_t* tokens, int32_t* n_tokens, float* scores);

/* ---- L2: RettoSession::run over a batch of pages --------------------------------- */
#define RT_MEM_HOST 0
#define RT_MEM_DEVICE 1
#define RT_MEM_HOST_MAPS_DEVICE 2   /* pages in host memory (what RettoSession::run is handed, session.rs:108-131), override maps in HBM */
/* rgb[i]: RGB8 HWC page i (hs[i] x ws[i]); mem says where the pixels live.
 * det_map_override (may be NULL, entries may be NULL): f32 [H,W] probability map at the
 * det input size of page i, in the memory space of the pages (RT_MEM_HOST_MAPS_DEVICE: in
 * device memory beside host pages), used INSTEAD of the det network's output when building
 * boxes (the network still runs).  This is the hook the synthetic-weights benchmark and the
 * teacher-forced parity tests use; a deployment has no such maps, so a host-fed measurement
 * keeps them in HBM and sends only the pages over PCIe. */
RT_API int rt_run_batch(rt_session* s, const uint8_t* const* rgb, const int* hs, const int* ws, int n_pages,
                        int mem, const float* const* det_map_override, rt_results** out);
/* RettoSession::run_stream (session.rs:133-143): as rt_run_batch, and cb(user, page, stage, json) is called for
 * every page with the stage's RettoWorkerStageResult JSON (stage 0 = Det, 1 = Cls, 2 = Rec; the reference's
 * mpsc::Sender order Det -> Cls -> Rec per image is kept).  Det is delivered as soon as the boxes of the page
 * are known -- before its crops are classified or read; Cls and Rec when the call completes.  Callbacks come
 * from the library's lane threads, one at a time; json is only valid during the call. */
typedef void (*rt_stage_callback)(void* user, int page, int stage, const char* json);
RT_API int rt_run_batch_stream(rt_session* s, const uint8_t* const* rgb, const int* hs, const int* ws, int n_pages,
                               int mem, const float* const* det_map_override, rt_stage_callback cb, void* user,
                               rt_results** out);
/* Asynchronous form of rt_run_batch (round 4) -- the counterpart of RettoSession::run_stream's worker thread + channel
 * (session.rs:108-143): rt_submit_batch splits the pages over the session's lanes exactly as rt_run_batch does and returns at
 * once; rt_wait_batch blocks until that batch is complete and returns its results (same object, same order, same values as
 * rt_run_batch).  Up to RT_MAX_INFLIGHT batches may be submitted ahead: a lane works through its parts of consecutive batches
 * back to back, so the host-side result assembly of batch i and the first kernels of batch i + 1 overlap with the other lanes'
 * work.  The argument ARRAYS are copied by rt_submit_batch; the PAGES (and override maps) they point to must stay valid and
 * unchanged until rt_wait_batch has returned for that ticket.  Every ticket must be waited for exactly once (any order); until
 * then every other call on the session except rt_submit_batch / rt_wait_batch fails with RT_ERR_INVALID.  rt_wait_batch of a
 * failed batch returns the failing stage's status (the other batches in flight are not affected).
 * Host pages (RT_MEM_HOST, RT_MEM_HOST_MAPS_DEVICE) are copied to HBM by rt_submit_batch itself, on a copy stream of the session,
 * part by part right before each lane's job is queued: with a batch submitted ahead the transfer of batch i + 1 runs under the
 * kernels of batch i, and no lane waits for PCIe with an empty stream.  The session keeps one staging buffer per batch in flight
 * (the batch's page bytes; reused).  From pageable memory the copy has completed when rt_submit_batch returns; from pinned memory
 * it is asynchronous -- either way the rule above (pages unchanged until rt_wait_batch) is the contract. */
#define RT_MAX_INFLIGHT 8   /* (a cap on queued tickets only: a lane works on one part at a time, whatever is queued behind it) */
typedef struct rt_ticket rt_ticket;
RT_API int rt_submit_batch(rt_session* s, const uint8_t* const* rgb, const int* hs, const int* ws, int n_pages, int mem,
                           const float* const* det_map_override, rt_ticket** out);
RT_API int rt_wait_batch(rt_session* s, rt_ticket* ticket, rt_results** out);
RT_API void rt_results_free(rt_results* r);
RT_API int rt_results_pages(const rt_results* r);
RT_API int rt_results_count(const rt_results* r, int page);
/* det: boxes in ORIGINAL image coordinates (session.rs:94-97) */
RT_API const float* rt_results_boxes(const rt_results* r, int page);      /* n x 8 */
RT_API const float* rt_results_det_scores(const rt_results* r, int page); /* n */
RT_API const uint16_t* rt_results_cls_labels(const rt_results* r, int page);
RT_API const float* rt_results_cls_scores(const rt_results* r, int page);
RT_API const float* rt_results_rec_scores(const rt_results* r, int page);
RT_API int rt_results_rec_tokens(const rt_results* r, int page, int line, const int32_t** tokens);
RT_API const char* rt_results_rec_text(const rt_results* r, int page, int line); /* UTF-8 */
/* f32 sum of every det probability map produced in the call (keeps the network's
 * output observable when det_map_override is used) */
RT_API double rt_results_det_checksum(const rt_results* r);
/* RettoWorkerStageResult JSON of one page in the serde shape retto-wasm emits
 * (retto-wasm/fe/index.ts:5-42); stage 0 = det, 1 = cls, 2 = rec. Library-owned. */
RT_API const char* rt_results_json(rt_results* r, int page, int stage);

/* ---- device memory + timing helpers for harnesses -------------------------------- */
RT_API int rt_device_malloc(rt_session* s, size_t bytes, void** out);
RT_API int rt_device_free(rt_session* s, void* p);
RT_API int rt_memcpy_h2d(rt_session* s, void* dst, const void* src, size_t bytes);
RT_API int rt_memcpy_d2h(rt_session* s, void* dst, const void* src, size_t bytes);
RT_API int rt_synchronize(rt_session* s);
/* Per-kernel-family device time of the last rt_run_batch / L1 call, measured with HIP
 * events on the session's own stream.  names/ms are library-owned arrays of *n entries. */
/* Upper bound on the concurrent lanes rt_run_batch uses from now on (<= the number created by
 * rt_config.lanes).  1 = strictly serial on the session's own stream (what the per-kernel
 * profile wants: concurrent lanes share the GPU and stretch each other's kernels). */
RT_API int rt_set_lanes(rt_session* s, int lanes);
/* on: 0 off; 1 every launch family ("gemm_pw/...", "dwconv5", ...) and the enclosing network scopes ("net/det", "net/cls",
 * "net/rec"); 2 the network scopes only (the per-launch event pairs are themselves work on the stream: whole-network device
 * times are read from a pass without them). */
RT_API int rt_profile_enable(rt_session* s, int on);
RT_API int rt_profile_get(rt_session* s, const char* const** names, const float** ms, const int** calls, int* n);

/* ---- model files -----------------------------------------------------
 * Replaces the model loading of retto-core/src/worker/ort_worker.rs:120-135 (the .onnx bytes
 * resolved by worker.rs:30-56 go to ONNX Runtime there).  rt_config's det / cls / rec sources
 * may hold either an RTWB blob or the PP-OCRv4 .onnx file itself: non-RTWB bytes are imported
 * by rt_create through the same code as rt_onnx_to_rtwb.  which: 0 det, 1 cls, 2 rec.
 * Host-only (no GPU needed).  *out is library-owned until rt_buffer_free; on failure err
 * (optional, err_cap bytes) receives the message and the RT_ERR_* code is returned. */
RT_API int rt_onnx_to_rtwb(int which, const void* onnx, size_t len, void** out, size_t* out_len, char* err, size_t err_cap);
RT_API void rt_buffer_free(void* p);

/* ---- encoded pages ----------------------------------------------------
 * rt_decode_image replaces ImageHelper::new_from_raw_img_flow (retto-core/src/image_helper.rs:34-44:
 * image::load_from_memory(bytes)?.to_rgb8()): PNG (all colour types / depths, Adam7), Huffman
 * JPEG (sequential and progressive; grey / YCbCr, any sampling), PNM, uncompressed BMP -> tightly packed RGB8 [h][w][3],
 * alpha dropped, 16-bit samples as (v + 128) / 257.  Host-only.  *rgb is library-owned until
 * rt_buffer_free.  Unknown / corrupt input: RT_ERR_IMAGE with the reason in err (optional).
 * rt_run_encoded_batch is RettoSession::run / run_stream (session.rs:108-143) over encoded
 * bytes: pages are decoded on host threads, then processed as by rt_run_batch_stream (cb may be NULL). */
/* Host CPUs this process should plan with: the affinity mask capped by the cgroup CPU quota, divided by LOCAL_WORLD_SIZE (one
 * process per GPU on a node).  rt_run_encoded_batch sizes its decode-thread pool with it (at most 16); std::thread::
 * hardware_concurrency() would report the machine (256 on the MI355X box) to a pod that owns 16 CPUs.  No reference counterpart
 * (retto-cli decodes on the calling thread, retto-cli/src/main.rs:80-86). */
RT_API int rt_host_cpu_budget(void);
RT_API int rt_decode_image(const void* data, size_t len, uint8_t** rgb, int* h, int* w, char* err, size_t err_cap);
RT_API int rt_run_encoded_batch(rt_session* s, const void* const* files, const size_t* lens, int n_pages,
                                rt_stage_callback cb, void* user, rt_results** out);
/* RecCharacter::new (retto-core/src/processor/rec_processor.rs:29-46) on the bytes of ppocr_keys_v1.txt, with
 * Rust's semantics: strict String::from_utf8 (RT_ERR_UTF8 on overlong forms, surrogates, > U+10FFFF), str::lines,
 * str::trim over the Unicode White_Space set (a U+3000-only line becomes ""), "blank" inserted at 0 and " "
 * appended.  *out = the entries joined by '\n' (library-owned until rt_buffer_free), *n_entries their count.
 * Host-only; rt_create runs the same code on rt_config.dict. */
RT_API int rt_parse_dictionary(const void* data, size_t len, char** out, size_t* out_len, int* n_entries, char* err, size_t err_cap);
/* The JSON text of one f32 as serde_json (ryu) writes it -- shortest round-trip digits, "1.0" / "0.9" /
 * "1.234e-7", null for non-finite -- which rt_results_json uses for every number.  Returns the length. */
RT_API int rt_format_f32(float v, char* buf, size_t cap);
/* The tensor list (RTWB names and shapes, forward order, -1 = read from the file) the importer
 * fills for model `which`, one "name d0 d1 ..." line per tensor; returns the length needed. */
RT_API size_t rt_model_manifest(int which, char* buf, size_t cap);

/* ---- multi-GPU: the one collective of the design ------------------------------------------------------------
 * Pages shard over the GPUs of a node, one process (one rt_session) per GPU, with no data-path collective; the model blobs are
 * broadcast ONCE from `root` over RCCL (xGMI).  The reference has no counterpart (retto-cli/src/main.rs:80-86 is a serial loop
 * in one process).  rt_rccl_unique_id: rank `root` creates the 128-byte RCCL id and hands it to the other ranks out of band
 * (file / pipe / environment / MPI).  rt_broadcast_blobs: every rank calls it with the same id, world and root; on `root`
 * data[i] / lens[i] are the blobs to send (caller-owned), on the other ranks they are OUTPUTS: data[i] is library-allocated
 * (release with rt_buffer_free) and lens[i] its size.  The blobs then go into rt_config.{det,cls,rec,dict}.  librccl.so is
 * loaded on first use only.  err (optional) receives the failure message. */
RT_API int rt_rccl_unique_id(void* id, size_t cap, char* err, size_t err_cap);
RT_API int rt_broadcast_blobs(const void* id, int rank, int world, int device_id, int root, int n_blobs, void** data, size_t* lens,
                              char* err, size_t err_cap);

/* ---- diagnostics for tools/ (kernel A/B switches and the GEMM micro-benchmark; no reference
 * counterpart, not needed by a drop-in host) ------------------------------------------------
 * rt_debug_set_variants: gemm_variant 0 = production dispatch, 8 / 10 / 15 / 20 force the
 * 128x128 / 128x240 / 256x240 wide tiles / the streaming kernel; dw_variant 0 = production,
 * 4 = 2-row depthwise strips; flags bits: 1 fused thin blocks OFF, 2 thin blocks on the
 * 128-pixel tile, 3 plain (not XCD-aware) depthwise block order, 4 32-channel depthwise slabs
 * only, 5 128- instead of 64-channel wide slabs, 6 CTC head on the 128x128 wide tile, 7 thin
 * LCNetV3 blocks back on k_lc_thin / the unfused pair (instead of k_lc_lds), 8 the wide fp32 GEMM
 * back on the register-staged tile (instead of k_gemm32p), 9 5x5 depthwise back on
 * k_dwconv_rows (instead of the column sweep), 10 the angle classifier's blocks as the unfused launch
 * series (instead of k_cls_block), 11 the RSEFPN output convs and the DB head conv as the round-3 launch
 * series over materialised upsampled tensors (instead of the upsampling-aware k_fpn_phase / k_fpn_class /
 * k_fpn_compose).
 * Process-wide; every setting but bits 10 and 11 computes bit-identical results (tests/test_gpu_parity.py
 * checks bits 7-9 on whole networks; bit 10 changes the order of the squeeze-excite pooling sums:
 * equal within 1e-6 on the class probabilities; bit 11 changes the summation order of the pre-summed phase
 * weights: equal within 2e-5 on the DB probability map). */
RT_API void rt_debug_set_variants(int gemm_variant, int dw_variant, int flags);
/* one launch of the fp16 implicit-GEMM conv kernel on host tensors: x [n, cin, h, w], wt [cout, cin, kh, kw], bias [cout] or
 * NULL, "same" padding k/2, stride (sh, sw), act = 0 none / 1 relu / 2 hardswish / 3 swish / 4 sigmoid -> out [n, cout, ho, wo] */
RT_API int rt_debug_conv16(rt_session* s, const float* x, int n, int cin, int h, int w, const float* wt, int cout, int kh, int kw,
                           int sh, int sw, const float* bias, int act, float* out);
/* times nn::gemm (M x K x N, random data) over `iters` launches on the session's stream and
 * returns the average ms and the max |diff| against variant 0 */
RT_API int rt_bench_gemm(rt_session* s, long long M, int K, int N, int variant, int iters, float* ms_out, float* maxdiff_out);
/* Error of one nn::gemm variant against an fp64 product on operands with full 24-bit significands (bias 0): out4 = {max |err|,
 * rms err, max |ref|, rms ref} over `rows` rows from the start, the middle and the end of the M rows.  variant 1 = narrow fp32-MFMA
 * kernel, 30 = k_gemm32p (fp32 MFMA), 40 = split-bf16 (three bf16 planes per operand, six bf16 MFMAs per product, fp32 accumulate). */
RT_API int rt_bench_gemm_err(rt_session* s, long long M, int K, int N, int variant, int rows, int act, unsigned seed, double* out4);
/* times the fused thin LCNetV3 block (3x3 depthwise -> 1x1 conv; n images of h x w, random data).  form: 0 = k_lc_thin
 * (workgroup-staged; the unfused depthwise + GEMM pair where it has no instance), 1 = k_lc_wave (direct loads, stride 1 only),
 * 3 = k_lc_lds (production), 5 = k_lc_lds incl. the opt-in 128 -> 128 split.  stride: 1, 2, or 21 = (2, 1).  Returns the average
 * ms and the max |diff| against form RT_BENCH_LC_REF (environment, default 0); RT_BENCH_LC_DUMP=1 prints where they differ. */
RT_API int rt_bench_lc(rt_session* s, int n, int h, int w, int cin, int cout, int stride, int form, int iters, float* ms_out, float* maxdiff_out);

#ifdef __cplusplus
}
#endif
#endif /* RETTO_HIP_H */
