// extern "C" surface of libretto_hip.so (include/retto_hip.h).
#include <thread>
#include <sched.h>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>

#include "geom_math.h"
#include "session.h"
#include "onnx_import.h"
#include "image_decode.h"

using namespace rt;

static thread_local std::string g_create_error;
static void capture_variant_defaults();   // the A/B switches' load-time values (rt_debug_set_variants restores to them)
const char* rt_results_json_impl(rt_results* r, int page, int stage);

// A call that fails after work was enqueued must not leave kernels or H2D copies in flight: the next
// begin_call() rewinds the pinned staging and the arenas they read.  Errors of the drain itself are dropped
// (the first failure is the one reported).
static void quiesce(rt_session* s) {
  if (!s) return;
  (void)hipSetDevice(s->device);
  if (s->st) (void)hipStreamSynchronize(s->st);
  for (auto& h : s->helpers)
    if (h->st) (void)hipStreamSynchronize(h->st);
  (void)hipGetLastError();
}
// ALLOW_INFLIGHT: only rt_submit_batch / rt_wait_batch may run while submitted batches are in flight -- every other entry point
// uses the main lane's stream and arenas, which lane 0's worker thread owns until the last ticket has been waited for.
template <bool ALLOW_INFLIGHT = false, typename F>
static int guarded(rt_session* s, F&& f) {
  if (!ALLOW_INFLIGHT && s && s->inflight.load() > 0) {
    s->last_error = "batches submitted with rt_submit_batch are in flight: call rt_wait_batch for every ticket first";
    return RT_ERR_INVALID;
  }
  // rt_session::last_error is written and cleared on the API caller's thread only (here, RT_REQUIRE, the shape checks): lane
  // threads keep their failure in the ticket (rt_ticket::errs) and it surfaces through rt_wait_batch's rethrow below.
  // On the ALLOW_INFLIGHT path nothing is drained here: the lane that failed has drained its own stream in the worker, and the
  // streams of the other lanes carry OTHER batches that a failed ticket must not stall.
  if (s) s->last_error.clear();
  try {
    f();
    return RT_OK;
  } catch (const RtError& e) {
    if (!ALLOW_INFLIGHT) quiesce(s);
    if (s) s->last_error = e.what(); else g_create_error = e.what();
    return e.code;
  } catch (const std::bad_alloc&) {
    if (!ALLOW_INFLIGHT) quiesce(s);
    if (s) s->last_error = "out of host memory"; else g_create_error = "out of host memory";
    return RT_ERR_BACKEND;
  } catch (const std::exception& e) {
    if (!ALLOW_INFLIGHT) quiesce(s);
    if (s) s->last_error = e.what(); else g_create_error = e.what();
    return RT_ERR_BACKEND;
  }
}
#define RT_REQUIRE(cond, s, msg)                                             \
  do {                                                                        \
    if (!(cond)) {                                                            \
      if (s) (s)->last_error = msg; else g_create_error = msg;                \
      return RT_ERR_INVALID;                                                  \
    }                                                                         \
  } while (0)

// RAII for the diagnostic hooks below: a process-wide A/B switch is put back and the scratch buffers are freed on EVERY way
// out of the hook (an RT_HIP_CHECK that throws used to leave the switch at the benchmark's value for every later call).
namespace {
struct RestoreInt { int& ref; int old; explicit RestoreInt(int& r) : ref(r), old(r) {} ~RestoreInt() { ref = old; } };
struct DevBufs {
  std::vector<void*> p;
  template <typename T> T* alloc(size_t n) { void* q = nullptr; RT_HIP_CHECK(hipMalloc(&q, std::max<size_t>(n, 1) * sizeof(T))); p.push_back(q); return (T*)q; }
  ~DevBufs() { for (void* q : p) (void)hipFree(q); }
};
struct ForgetSplit { const float* w; ~ForgetSplit() { nn::gemm_split_forget(w); } };
struct Events { hipEvent_t a = nullptr, b = nullptr; ~Events() { if (a) (void)hipEventDestroy(a); if (b) (void)hipEventDestroy(b); } };
}  // namespace

extern "C" {

void rt_config_default(rt_config* c) {
  if (!c) return;
  memset(c, 0, sizeof(*c));
  c->struct_size = (uint32_t)sizeof(rt_config);
  c->device_id = 0;
  c->max_side_len = 2000; c->min_side_len = 30;
  c->det_limit_side_len = 736; c->det_limit_type = 0;
  for (int i = 0; i < 3; i++) { c->det_mean[i] = 0.5f; c->det_std[i] = 0.5f; }
  c->det_scale = 1.0f / 255.0f;
  c->det_thresh = 0.3f; c->det_box_thresh = 0.5f; c->det_unclip_ratio = 1.6f;
  c->det_min_mini_box_size = 3; c->det_dilation = 1;
  c->cls_image_shape[0] = 3; c->cls_image_shape[1] = 48; c->cls_image_shape[2] = 192;
  c->cls_batch_num = 6; c->cls_thresh = 0.9f;
  c->rec_image_shape[0] = 3; c->rec_image_shape[1] = 48; c->rec_image_shape[2] = 320;
  c->rec_batch_num = 6;
  c->max_boxes_per_page = 0; c->det_sub_batch = 0; c->lanes = 0; c->dtype = RT_DTYPE_F32;
}

int rt_create(const rt_config* cfg, rt_session** out) {
  RT_REQUIRE(cfg && out, (rt_session*)nullptr, "rt_create: null argument");
  *out = nullptr;
  RT_REQUIRE(cfg->struct_size == sizeof(rt_config), (rt_session*)nullptr,
             "rt_config.struct_size does not match this library's rt_config: fill the struct with rt_config_default() of the same header");
  RT_REQUIRE(cfg->rec_batch_num > 0 && cfg->cls_batch_num > 0, (rt_session*)nullptr, "batch_num must be positive");
  RT_REQUIRE(cfg->cls_image_shape[0] == 3 && cfg->rec_image_shape[0] == 3 && cfg->rec_image_shape[1] == 48 &&
                 cfg->cls_image_shape[1] == 48 && cfg->cls_image_shape[2] == 192,
             (rt_session*)nullptr, "unsupported cls/rec image_shape for the PP-OCRv4 mobile graphs");
  RT_REQUIRE(cfg->lanes >= 0 && cfg->lanes <= 4, (rt_session*)nullptr, "lanes must be in [0, 4]");
  RT_REQUIRE(cfg->dtype == RT_DTYPE_F32 || cfg->dtype == RT_DTYPE_F16, (rt_session*)nullptr, "dtype must be RT_DTYPE_F32 or RT_DTYPE_F16");
  RT_REQUIRE(cfg->max_boxes_per_page >= 0 && cfg->max_boxes_per_page <= 65536, (rt_session*)nullptr,
             "max_boxes_per_page must be in [0, 65536]");
  capture_variant_defaults();
  return guarded(nullptr, [&] { *out = rt_session_create(cfg); });
}
void rt_destroy(rt_session* s) {
  if (!s) return;
  (void)hipSetDevice(s->device);
  for (auto& w : s->workers) w->shutdown();   // lanes finish what was submitted (tickets never waited for are leaked, not raced)
  s->workers.clear();
  if (s->st) { (void)hipStreamSynchronize(s->st); }
  for (auto& h : s->helpers) {
    if (h->st) (void)hipStreamSynchronize(h->st);
    if (h->d_flags) (void)hipFree(h->d_flags);
    if (h->ev_block) (void)hipEventDestroy(h->ev_block);
    if (h->st_part) { rt::forget_stream(h->st_part); (void)hipStreamDestroy(h->st_part); }
    if (h->st_full) (void)hipStreamDestroy(h->st_full);
  }
  s->helpers.clear();
  s->free_stage();
  s->det.reset(); s->cls.reset(); s->rec.reset();
  if (s->d_flags) (void)hipFree(s->d_flags);
  if (s->ev_block) (void)hipEventDestroy(s->ev_block);
  if (s->st_part) { rt::forget_stream(s->st_part); (void)hipStreamDestroy(s->st_part); }
  if (s->st_full) (void)hipStreamDestroy(s->st_full);
  delete s;
}
const char* rt_last_error(const rt_session* s) { return s ? s->last_error.c_str() : g_create_error.c_str(); }
const char* rt_version(void) { return "retto_hip 0.1.0 (gfx950)"; }

int rt_det(rt_session* s, const float* nchw, int n, int c, int h, int w, float* out) {
  RT_REQUIRE(s && nchw && out, s, "rt_det: null argument");
  if (c != 3 || n <= 0 || h <= 0 || w <= 0 || h % 32 || w % 32) { s->last_error = "rt_det: expected [n,3,h,w] with h,w multiples of 32"; return RT_ERR_SHAPE; }
  return guarded(s, [&] { s->det_forward(nchw, n, h, w, out); });
}
int rt_cls(rt_session* s, const float* nchw, int n, int c, int h, int w, float* out) {
  RT_REQUIRE(s && nchw && out, s, "rt_cls: null argument");
  if (c != 3 || n <= 0 || h != 48 || w != 192) { s->last_error = "rt_cls: expected [n,3,48,192]"; return RT_ERR_SHAPE; }
  return guarded(s, [&] { s->cls_forward(nchw, n, h, w, out); });
}
int rt_rec(rt_session* s, const float* nchw, int n, int c, int h, int w, float* out, int* t_out) {
  RT_REQUIRE(s, s, "rt_rec: null session");
  if (c != 3 || n <= 0 || h != 48 || w < 8) { s->last_error = "rt_rec: expected [n,3,48,w>=8]"; return RT_ERR_SHAPE; }
  RT_REQUIRE(out == nullptr || nchw != nullptr, s, "rt_rec: null input");
  return guarded(s, [&] { s->rec_forward(nchw, n, h, w, out, t_out); });
}
int rt_rec_ragged(rt_session* s, const float* nchw, int n, const int* widths, float* out, int* t_out) {
  RT_REQUIRE(s && widths && n > 0, s, "rt_rec_ragged: bad argument");
  RT_REQUIRE(out == nullptr || nchw != nullptr, s, "rt_rec_ragged: null input");
  return guarded(s, [&] { s->rec_forward_ragged(nchw, n, widths, out, t_out); });
}
int rt_rec_classes(const rt_session* s) { return s ? s->rec->classes() : 0; }
const char* rt_model_info(const rt_session* s) { return s ? s->model_info.c_str() : ""; }

int rt_resize_both_dims(const rt_session* s, int h, int w, int* out_h, int* out_w) {
  if (!s || !out_h || !out_w) return RT_ERR_INVALID;
  int plan[4];
  int n = gm::resize_both_plan(h, w, s->cfg.max_side_len, s->cfg.min_side_len, plan);
  *out_h = n ? plan[2 * (n - 1)] : h; *out_w = n ? plan[2 * (n - 1) + 1] : w;
  return RT_OK;
}
int rt_resize_both(rt_session* s, const uint8_t* rgb, int h, int w, uint8_t* out, int out_h, int out_w) {
  RT_REQUIRE(s && rgb && out && h > 0 && w > 0, s, "rt_resize_both: bad argument");
  return guarded(s, [&] { s->resize_both(rgb, h, w, out, out_h, out_w); });
}
int rt_det_input_dims(const rt_session* s, int h, int w, int* out_h, int* out_w) {
  if (!s || !out_h || !out_w) return RT_ERR_INVALID;
  gm::resize_either_dims(h, w, s->cfg.det_limit_type, s->cfg.det_limit_side_len, out_h, out_w);
  return RT_OK;
}
int rt_det_preprocess(rt_session* s, const uint8_t* rgb, int h, int w, float* out_nchw) {
  RT_REQUIRE(s && rgb && out_nchw && h > 0 && w > 0, s, "rt_det_preprocess: bad argument");
  return guarded(s, [&] { s->det_preprocess(rgb, h, w, out_nchw); });
}
int rt_det_postprocess(rt_session* s, const float* pred, int h, int w, int ori_h, int ori_w, float* boxes, float* scores,
                       int max_out, int* n_out) {
  RT_REQUIRE(s && pred && boxes && scores && n_out && h > 0 && w > 0, s, "rt_det_postprocess: bad argument");
  return guarded(s, [&] { s->det_postprocess(pred, h, w, ori_h, ori_w, boxes, scores, max_out, n_out); });
}
int rt_crop_dims(const float* boxes, int n, int* ws, int* hs) {
  if (!boxes || !ws || !hs) return RT_ERR_INVALID;
  for (int i = 0; i < n; i++) {
    gm::CropDims d = gm::crop_dims(boxes + 8 * i);
    ws[i] = d.rot ? d.h : d.w; hs[i] = d.rot ? d.w : d.h;
  }
  return RT_OK;
}
int rt_crop_images(rt_session* s, const uint8_t* rgb, int h, int w, const float* boxes, int n, uint8_t* out, size_t out_cap) {
  RT_REQUIRE(s && rgb && boxes && out && h > 0 && w > 0 && n >= 0, s, "rt_crop_images: bad argument");
  return guarded(s, [&] { s->crop_images(rgb, h, w, boxes, n, out, out_cap); });
}
int rt_scale_and_clip(float* boxes, int n, double bitmap_w, double bitmap_h, double ori_w, double ori_h) {
  if (!boxes) return RT_ERR_INVALID;
  for (int i = 0; i < n; i++) gm::scale_and_clip(boxes + 8 * i, bitmap_w, bitmap_h, ori_w, ori_h);
  return RT_OK;
}
int rt_resize_norm_width(int img_h, int img_w, float max_wh_ratio) { return gm::resize_norm_width(img_h, img_w, max_wh_ratio); }
int rt_resize_norm_image(rt_session* s, const uint8_t* crop, int h, int w, int ori_h, int ori_w, int img_h, int img_w,
                         float max_wh_ratio, float* out_chw) {
  RT_REQUIRE(s && crop && out_chw && h > 0 && w > 0 && img_h > 0, s, "rt_resize_norm_image: bad argument");
  return guarded(s, [&] { s->resize_norm_image(crop, h, w, ori_h, ori_w, img_h, img_w, max_wh_ratio, out_chw); });
}
int rt_ctc_decode(rt_session* s, const float* probs, int n, int t, int c, int32_t* idx, float* prob, int32_t* tokens,
                  int32_t* n_tokens, float* scores) {
  RT_REQUIRE(s && probs && idx && prob && tokens && n_tokens && scores && n > 0 && t > 0 && c > 0, s, "rt_ctc_decode: bad argument");
  return guarded(s, [&] { s->ctc_decode(probs, n, t, c, idx, prob, tokens, n_tokens, scores); });
}

int rt_run_batch(rt_session* s, const uint8_t* const* rgb, const int* hs, const int* ws, int n_pages, int mem,
                 const float* const* det_map_override, rt_results** out) {
  RT_REQUIRE(s && out && n_pages >= 0 && (n_pages == 0 || (rgb && hs && ws)), s, "rt_run_batch: bad argument");
  RT_REQUIRE(mem == RT_MEM_HOST || mem == RT_MEM_DEVICE || mem == RT_MEM_HOST_MAPS_DEVICE, s, "rt_run_batch: bad mem kind");
  *out = nullptr;
  return guarded(s, [&] { *out = s->run_batch(rgb, hs, ws, n_pages, mem, det_map_override); });
}
int rt_run_batch_stream(rt_session* s, const uint8_t* const* rgb, const int* hs, const int* ws, int n_pages, int mem,
                        const float* const* det_map_override, rt_stage_callback cb, void* user, rt_results** out) {
  RT_REQUIRE(s && out && cb && n_pages >= 0 && (n_pages == 0 || (rgb && hs && ws)), s, "rt_run_batch_stream: bad argument");
  RT_REQUIRE(mem == RT_MEM_HOST || mem == RT_MEM_DEVICE || mem == RT_MEM_HOST_MAPS_DEVICE, s, "rt_run_batch_stream: bad mem kind");
  *out = nullptr;
  return guarded(s, [&] { *out = s->run_batch(rgb, hs, ws, n_pages, mem, det_map_override, cb, user); });
}
int rt_submit_batch(rt_session* s, const uint8_t* const* rgb, const int* hs, const int* ws, int n_pages, int mem,
                    const float* const* det_map_override, rt_ticket** out) {
  RT_REQUIRE(s && out && n_pages >= 0 && (n_pages == 0 || (rgb && hs && ws)), s, "rt_submit_batch: bad argument");
  RT_REQUIRE(mem == RT_MEM_HOST || mem == RT_MEM_DEVICE || mem == RT_MEM_HOST_MAPS_DEVICE, s, "rt_submit_batch: bad mem kind");
  RT_REQUIRE(s->inflight.load() < RT_MAX_INFLIGHT, s, "rt_submit_batch: too many batches in flight (RT_MAX_INFLIGHT)");
  *out = nullptr;
  return guarded<true>(s, [&] { *out = s->submit_batch(rgb, hs, ws, n_pages, mem, det_map_override); });
}
int rt_wait_batch(rt_session* s, rt_ticket* ticket, rt_results** out) {
  RT_REQUIRE(s && ticket && out, s, "rt_wait_batch: bad argument");
  *out = nullptr;
  // (a failed batch: its own lanes have drained their streams in the worker; quiesce() in guarded() drains the rest)
  return guarded<true>(s, [&] { *out = s->wait_batch(ticket); });
}
int rt_host_cpu_budget(void) {
  int n = (int)std::thread::hardware_concurrency();
  cpu_set_t set;
  CPU_ZERO(&set);
  if (sched_getaffinity(0, sizeof(set), &set) == 0) { const int c = CPU_COUNT(&set); if (c > 0) n = c; }
  double quota = 0.0;
  if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {   // cgroup v2: "max <period>" or "<quota> <period>"
    char q[64]; double per = 0.0;
    if (fscanf(f, "%63s %lf", q, &per) == 2 && strcmp(q, "max") != 0 && per > 0) quota = atof(q) / per;
    fclose(f);
  } else {
    double q = -1, per = 0;
    if (FILE* fq = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) { if (fscanf(fq, "%lf", &q) != 1) q = -1; fclose(fq); }
    if (FILE* fp = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (fscanf(fp, "%lf", &per) != 1) per = 0; fclose(fp); }
    if (q > 0 && per > 0) quota = q / per;
  }
  if (quota > 0) n = std::max(1, std::min(n, (int)(quota + 0.5)));
  int ranks = 1;
  if (const char* e = getenv("LOCAL_WORLD_SIZE")) ranks = std::max(1, atoi(e));
  return std::max(1, n / ranks);
}
int rt_decode_image(const void* data, size_t len, uint8_t** rgb, int* h, int* w, char* err, size_t err_cap) {
  if (err && err_cap) err[0] = 0;
  if (!data || !rgb || !h || !w) { if (err && err_cap) snprintf(err, err_cap, "rt_decode_image: null argument"); return RT_ERR_INVALID; }
  *rgb = nullptr;
  try {
    std::vector<uint8_t> px;
    rt::decode_image((const uint8_t*)data, len, &px, h, w);
    uint8_t* p = (uint8_t*)malloc(px.size());
    if (!p) throw RtError(RT_ERR_BACKEND, "out of memory");
    memcpy(p, px.data(), px.size());
    *rgb = p;
    return RT_OK;
  } catch (const RtError& e) {
    if (err && err_cap) snprintf(err, err_cap, "%s", e.what());
    return e.code;
  } catch (const std::exception& e) {
    if (err && err_cap) snprintf(err, err_cap, "%s", e.what());
    return RT_ERR_BACKEND;
  }
}
int rt_parse_dictionary(const void* data, size_t len, char** out, size_t* out_len, int* n_entries, char* err, size_t err_cap) {
  if (err && err_cap) err[0] = 0;
  if ((!data && len) || !out || !out_len || !n_entries) { if (err && err_cap) snprintf(err, err_cap, "rt_parse_dictionary: null argument"); return RT_ERR_INVALID; }
  *out = nullptr; *out_len = 0; *n_entries = 0;
  try {
    std::vector<uint8_t> bytes((const uint8_t*)data, (const uint8_t*)data + len);
    std::vector<std::string> d = rt::load_dictionary(bytes);
    std::string j;
    for (size_t i = 0; i < d.size(); i++) { if (i) j += '\n'; j += d[i]; }
    char* p = (char*)malloc(j.size() + 1);
    if (!p) throw RtError(RT_ERR_BACKEND, "out of memory");
    memcpy(p, j.data(), j.size()); p[j.size()] = 0;
    *out = p; *out_len = j.size(); *n_entries = (int)d.size();
    return RT_OK;
  } catch (const RtError& e) {
    if (err && err_cap) snprintf(err, err_cap, "%s", e.what());
    return e.code;
  } catch (const std::exception& e) {
    if (err && err_cap) snprintf(err, err_cap, "%s", e.what());
    return RT_ERR_BACKEND;
  }
}
int rt_format_f32(float v, char* buf, size_t cap) {
  std::string s = rt_format_f32_impl(v);
  if (buf && cap) { size_t n = std::min(cap - 1, s.size()); memcpy(buf, s.data(), n); buf[n] = 0; }
  return (int)s.size();
}
// RettoSession::run / run_stream take the encoded bytes (session.rs:108,133): decode on host threads, then the batch path
int rt_run_encoded_batch(rt_session* s, const void* const* files, const size_t* lens, int n_pages, rt_stage_callback cb,
                         void* user, rt_results** out) {
  RT_REQUIRE(s && out && n_pages >= 0 && (n_pages == 0 || (files && lens)), s, "rt_run_encoded_batch: bad argument");
  *out = nullptr;
  return guarded(s, [&] {
    std::vector<std::vector<uint8_t>> px((size_t)n_pages);
    std::vector<int> hs((size_t)n_pages), ws((size_t)n_pages);
    std::vector<std::exception_ptr> errs((size_t)n_pages);
    std::atomic<int> next{0};
    auto work = [&] {
      for (int i; (i = next.fetch_add(1)) < n_pages;) {
        try {
          if (!files[i]) throw RtError(RT_ERR_IMAGE, "image decode: null input");
          rt::decode_image((const uint8_t*)files[i], lens[i], &px[(size_t)i], &hs[(size_t)i], &ws[(size_t)i]);
        } catch (...) { errs[(size_t)i] = std::current_exception(); }
      }
    };
    // decode threads: the CPUs this PROCESS may use (affinity mask capped by the cgroup quota -- the GPU box shows 256 logical
    // CPUs to a pod that owns 16), shared among the ranks of the node (LOCAL_WORLD_SIZE: one process per GPU), at most 16
    const int nt = std::max(1, std::min<int>(n_pages, std::min<int>(16, rt_host_cpu_budget())));
    std::vector<std::thread> th;
    for (int t = 1; t < nt; t++) th.emplace_back(work);
    work();
    for (auto& t : th) t.join();
    for (auto& e : errs) if (e) std::rethrow_exception(e);  // first failing page in page order, like the reference's `?`
    std::vector<const uint8_t*> ptrs((size_t)n_pages);
    for (int i = 0; i < n_pages; i++) ptrs[(size_t)i] = px[(size_t)i].data();
    *out = s->run_batch(ptrs.data(), hs.data(), ws.data(), n_pages, RT_MEM_HOST, nullptr, cb, user);
  });
}
void rt_results_free(rt_results* r) { delete r; }
int rt_results_pages(const rt_results* r) { return r ? (int)r->pages.size() : 0; }
#define RT_PAGE(r, page) ((r) && (page) >= 0 && (size_t)(page) < (r)->pages.size() ? &(r)->pages[(size_t)(page)] : nullptr)
int rt_results_count(const rt_results* r, int page) { auto* p = RT_PAGE(r, page); return p ? (int)p->det_scores.size() : 0; }
const float* rt_results_boxes(const rt_results* r, int page) { auto* p = RT_PAGE(r, page); return p ? p->boxes.data() : nullptr; }
const float* rt_results_det_scores(const rt_results* r, int page) { auto* p = RT_PAGE(r, page); return p ? p->det_scores.data() : nullptr; }
const uint16_t* rt_results_cls_labels(const rt_results* r, int page) { auto* p = RT_PAGE(r, page); return p ? p->cls_labels.data() : nullptr; }
const float* rt_results_cls_scores(const rt_results* r, int page) { auto* p = RT_PAGE(r, page); return p ? p->cls_scores.data() : nullptr; }
const float* rt_results_rec_scores(const rt_results* r, int page) { auto* p = RT_PAGE(r, page); return p ? p->rec_scores.data() : nullptr; }
int rt_results_rec_tokens(const rt_results* r, int page, int line, const int32_t** tokens) {
  auto* p = RT_PAGE(r, page);
  if (!p || line < 0 || (size_t)line >= p->tokens.size()) return 0;
  if (tokens) *tokens = p->tokens[(size_t)line].data();
  return (int)p->tokens[(size_t)line].size();
}
const char* rt_results_rec_text(const rt_results* r, int page, int line) {
  auto* p = RT_PAGE(r, page);
  if (!p || line < 0 || (size_t)line >= p->text.size()) return nullptr;
  return p->text[(size_t)line].c_str();
}
double rt_results_det_checksum(const rt_results* r) { return r ? r->det_checksum : 0.0; }
const char* rt_results_json(rt_results* r, int page, int stage) {
  if (!RT_PAGE(r, page) || stage < 0 || stage > 2) return nullptr;
  return rt_results_json_impl(r, page, stage);
}

int rt_device_malloc(rt_session* s, size_t bytes, void** out) {
  RT_REQUIRE(s && out, s, "rt_device_malloc: null argument");
  return guarded(s, [&] { RT_HIP_CHECK(hipSetDevice(s->device)); RT_HIP_CHECK(hipMalloc(out, bytes)); });
}
int rt_device_free(rt_session* s, void* p) {
  RT_REQUIRE(s, s, "rt_device_free: null session");
  return guarded(s, [&] { RT_HIP_CHECK(hipSetDevice(s->device)); RT_HIP_CHECK(hipFree(p)); });
}
int rt_memcpy_h2d(rt_session* s, void* dst, const void* src, size_t bytes) {
  RT_REQUIRE(s && dst && src, s, "rt_memcpy_h2d: null argument");
  return guarded(s, [&] { RT_HIP_CHECK(hipSetDevice(s->device)); RT_HIP_CHECK(hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice)); });
}
int rt_memcpy_d2h(rt_session* s, void* dst, const void* src, size_t bytes) {
  RT_REQUIRE(s && dst && src, s, "rt_memcpy_d2h: null argument");
  return guarded(s, [&] { RT_HIP_CHECK(hipSetDevice(s->device)); RT_HIP_CHECK(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost)); });
}
int rt_synchronize(rt_session* s) {
  RT_REQUIRE(s, s, "rt_synchronize: null session");
  return guarded(s, [&] { RT_HIP_CHECK(hipSetDevice(s->device)); RT_HIP_CHECK(hipDeviceSynchronize()); });
}
int rt_set_lanes(rt_session* s, int lanes) {
  RT_REQUIRE(s && lanes >= 1, s, "rt_set_lanes: bad argument");
  s->active_lanes = lanes;
  return RT_OK;
}
int rt_profile_enable(rt_session* s, int on) {
  RT_REQUIRE(s, s, "rt_profile_enable: null session");
  return guarded(s, [&] {
    const int mode = on == 2 ? 2 : (on != 0 ? 1 : 0);   // 2: the enclosing network scopes only
    s->prof.clear(); s->prof.on = mode;
    for (auto& h : s->helpers) { h->prof.clear(); h->prof.on = mode; }
  });
}
int rt_profile_get(rt_session* s, const char* const** names, const float** ms, const int** calls, int* n) {
  RT_REQUIRE(s && names && ms && calls && n, s, "rt_profile_get: null argument");
  for (auto& h : s->helpers) s->prof.merge(h->prof);
  *names = s->prof.names.data(); *ms = s->prof.ms.data(); *calls = s->prof.calls.data(); *n = (int)s->prof.names.size();
  return RT_OK;
}

int rt_onnx_to_rtwb(int which, const void* onnx, size_t len, void** out, size_t* out_len, char* err, size_t err_cap) {
  if (err && err_cap) err[0] = 0;
  if (!onnx || !len || !out || !out_len) { if (err && err_cap) snprintf(err, err_cap, "rt_onnx_to_rtwb: null argument"); return RT_ERR_INVALID; }
  try {
    std::vector<uint8_t> b = rt::onnx_to_rtwb(which, (const uint8_t*)onnx, len);
    void* p = malloc(b.size());
    if (!p) throw RtError(RT_ERR_BACKEND, "out of memory");
    memcpy(p, b.data(), b.size());
    *out = p; *out_len = b.size();
    return RT_OK;
  } catch (const RtError& e) {
    if (err && err_cap) snprintf(err, err_cap, "%s", e.what());
    return e.code;
  } catch (const std::exception& e) {
    if (err && err_cap) snprintf(err, err_cap, "%s", e.what());
    return RT_ERR_BACKEND;
  }
}
void rt_buffer_free(void* p) { free(p); }
size_t rt_model_manifest(int which, char* buf, size_t cap) {
  std::string s;
  try {
    for (const rt::ManifestEntry& m : rt::model_manifest(which)) {
      s += m.name;
      for (int d : m.dims) s += " " + std::to_string(d);
      s += "\n";
    }
  } catch (const std::exception&) { return 0; }
  if (buf && cap) { size_t n = std::min(cap - 1, s.size()); memcpy(buf, s.data(), n); buf[n] = 0; }
  return s.size() + 1;
}

// A/B switches for tools/ (include/retto_hip.h, diagnostics section)
// what the environment selected when the library was loaded (dynamic initialisation runs after the nn:: globals of the other
// translation units only by luck of link order, so these are read on the first call of rt_create -- before any hook can have
// changed them -- see capture_variant_defaults())
static int g_default_lc_wave = 3, g_default_gemm_dma = 1, g_default_dw_sweep = 4, g_default_cls_fused = 1, g_default_gemm_split = 0;
static void capture_variant_defaults() {
  static const bool once = [] {
    g_default_lc_wave = nn::g_lc_wave; g_default_gemm_dma = nn::g_gemm_dma; g_default_dw_sweep = nn::g_dw_sweep; g_default_cls_fused = nn::g_cls_fused; g_default_gemm_split = nn::g_gemm_split;
    return true;
  }();
  (void)once;
}
RT_API void rt_debug_set_variants(int gemm_variant, int dw_variant, int flags) {
  capture_variant_defaults();
  nn::g_gemm_variant = gemm_variant;
  nn::g_dw_variant = dw_variant;
  nn::g_lc_thin = (flags & 2) ? 0 : ((flags & 4) ? 2 : 4);
  nn::set_dw_xcd((flags & 8) ? 0 : 1);
  nn::g_dw_wide_slab_min = (flags & 16) ? (1 << 30) : 192;
  nn::g_dw_wide3_min = (flags & 16) ? (1 << 30) : 128;
  nn::g_dw_wide_lp = (flags & 32) ? 32 : 16;
  nn::g_argmax_wide = (flags & 64) ? 2 : 0;
  // round-3 kernels: bits 7-9 send their layers back to the kernels they replaced (defaults = what the environment selected at load)
  const int lc_wave0 = g_default_lc_wave, gemm_dma0 = g_default_gemm_dma, dw_sweep0 = g_default_dw_sweep, cls_fused0 = g_default_cls_fused;
  nn::g_lc_wave = (flags & 128) ? 0 : lc_wave0;
  nn::g_gemm_dma = (flags & 256) ? 0 : gemm_dma0;
  nn::g_dw_sweep = (flags & 512) ? 0 : dw_sweep0;
  nn::g_fpn_phase_off = (flags & 2048) ? 1 : 0;   // (bit 11: RSEFPN / DB-head convs as the round-3 launch series; equal to fp32 rounding, not bit-identical)
  nn::g_gemm_split = (flags & 4096) ? 1 : g_default_gemm_split;   // (bit 12: the split-bf16 form of the wide rec-net GEMMs, opt-in)
  nn::g_cls_fused = (flags & 1024) ? 0 : cls_fused0;   // (bit 10: the classifier's blocks as the unfused launch series; fp32-tolerance equal, not bit-identical)
}
// Runs one nh::conv16 launch on host tensors (diagnostics: the numerics tests compare it with torch conv2d).
// x [n, cin, h, w] f32, w [cout, cin, kh, kw] f32, bias [cout] or null, "same" padding k/2, stride (sh, sw); out [n, cout, ho, wo] f32.
RT_API int rt_debug_conv16(rt_session* s, const float* x, int n, int cin, int h, int w, const float* wt, int cout, int kh, int kw,
                           int sh, int sw, const float* bias, int act, float* out) {
  RT_REQUIRE(s && x && wt && out && n > 0 && cin > 0 && h > 0 && w > 0 && cout > 0, s, "rt_debug_conv16: bad argument");
  return guarded(s, [&] {
    using nh::half_t;
    s->begin_call();
    const int cp = nh::pitch8(cin), op = nh::pitch8(cout), npad = round_up(cout, 32), nslab = (cp + 31) / 32;
    const int ho = (h - 1) / sh + 1, wo = (w - 1) / sw + 1;
    std::vector<half_t> hx((size_t)n * h * w * cp, (half_t)0.f), hw((size_t)nslab * kh * kw * npad * 32, (half_t)0.f);
    for (int i = 0; i < n; i++)
      for (int c = 0; c < cin; c++)
        for (int p = 0; p < h * w; p++) hx[((size_t)i * h * w + p) * cp + c] = (half_t)x[((size_t)i * cin + c) * h * w + p];
    for (int o = 0; o < cout; o++)
      for (int c = 0; c < cin; c++)
        for (int t = 0; t < kh * kw; t++)
          hw[((((size_t)(c / 32) * kh + t / kw) * kw + t % kw) * npad + o) * 32 + c % 32] = (half_t)wt[((size_t)o * cin + c) * kh * kw + t];
    std::vector<float> hb(npad, 0.f);
    if (bias) memcpy(hb.data(), bias, (size_t)cout * sizeof(float));
    half_t* dx = s->arena.alloc<half_t>(hx.size()); half_t* dw = s->arena.alloc<half_t>(hw.size());
    float* db = s->arena.alloc<float>(hb.size()); half_t* dy = s->arena.alloc<half_t>((size_t)n * ho * wo * op);
    RT_HIP_CHECK(hipMemcpyAsync(dx, hx.data(), hx.size() * 2, hipMemcpyHostToDevice, s->st));
    RT_HIP_CHECK(hipMemcpyAsync(dw, hw.data(), hw.size() * 2, hipMemcpyHostToDevice, s->st));
    RT_HIP_CHECK(hipMemcpyAsync(db, hb.data(), hb.size() * 4, hipMemcpyHostToDevice, s->st));
    Level Li = make_level(std::vector<std::pair<int, int>>((size_t)n, {h, w})), Lo = make_level(std::vector<std::pair<int, int>>((size_t)n, {ho, wo}));
    const bool flat = kh == 1 && kw == 1 && sh == 1 && sw == 1;   // as the networks run their 1x1 layers: one GEMM over all pixels
    Level Lf = flat ? flat_level(Li) : Level();
    RunCtx c = s->ctx(&s->arena);
    if (flat) upload_levels(c, {&Li, &Lo, &Lf}); else upload_levels(c, {&Li, &Lo});
    nh::Epi16 e; e.bias = db; e.act = act;
    long long* d_st = nullptr;
    const bool stamps = getenv("RT_CONV_STAMPS") != nullptr;
    if (stamps && !nh::conv_stamps_compiled()) fprintf(stderr, "RT_CONV_STAMPS: this library was built without the stamp code (make STAMPS=1): times only\n");
    if (stamps) { d_st = s->arena.alloc<long long>(4096); RT_HIP_CHECK(hipMemsetAsync(d_st, 0, 4096 * 8, s->st)); nh::g_conv_stamps = d_st; }
    hipEvent_t ev0, ev1;
    RT_HIP_CHECK(hipEventCreate(&ev0)); RT_HIP_CHECK(hipEventCreate(&ev1));
    const int reps = stamps ? 5 : 1;
    for (int rep = 0; rep < reps; rep++) {
      if (rep == reps - 1) RT_HIP_CHECK(hipEventRecord(ev0, s->st));
      if (flat) nh::conv16(s->st, dx, cp, Lf.d, Lf.d, 1, 1, Lf.maxW, cp, 1, 1, 1, 1, 0, 0, dw, cout, npad, dy, op, 0, e);
      else nh::conv16(s->st, dx, cp, Li.d, Lo.d, n, Lo.maxH, Lo.maxW, cp, kh, kw, sh, sw, kh / 2, kw / 2, dw, cout, npad, dy, op, 0, e);
    }
    RT_HIP_CHECK(hipEventRecord(ev1, s->st));
    std::vector<half_t> hy((size_t)n * ho * wo * op);
    RT_HIP_CHECK(hipMemcpyAsync(hy.data(), dy, hy.size() * 2, hipMemcpyDeviceToHost, s->st));
    s->sync();
    if (!stamps) { (void)hipEventDestroy(ev0); (void)hipEventDestroy(ev1); }
    if (stamps) {
      nh::g_conv_stamps = nullptr;
      std::vector<long long> hs(4096);
      RT_HIP_CHECK(hipMemcpy(hs.data(), d_st, 4096 * 8, hipMemcpyDeviceToHost));
      const int nrows = ((cp + 31) / 32) * kh;
      float ms = 0.f;
      RT_HIP_CHECK(hipEventElapsedTime(&ms, ev0, ev1));
      fprintf(stderr, "launch %.3f ms = %.1f TFLOP/s; one workgroup (k_conv16v2): entry->requests %lld, ->data landed %lld, main loop %lld, epilogue %lld ticks (barrier %lld, math + transpose + store issue %lld, store drain %lld)\n",
              ms, 2.0 * n * ho * wo * (double)cout * cin * kh * kw / ms / 1e9, hs[4001] - hs[4000], hs[4002] - hs[4001], hs[4003] - hs[4002], hs[4004] - hs[4003], hs[4005] - hs[4003], hs[4006] - hs[4005], hs[4004] - hs[4006]);
      (void)hipEventDestroy(ev0); (void)hipEventDestroy(ev1);
      fprintf(stderr, "conv16 stamps (s_memtime ticks): stage: t1-t0 | t2-t1 | t3-t2 | t4-t3 | next t0 - t0   (k_conv16: barrier, staging, barrier, MFMAs; k_conv16v2: DMA issue, MFMAs, vmcnt wait, barrier)\n");
      for (int r = 0; r < nrows && r < 790; r++) {
        const long long* t = &hs[(size_t)r * 5];
        const long long nxt = r + 1 < nrows ? hs[(size_t)(r + 1) * 5] : t[4];
        fprintf(stderr, "  %3d: %6lld %6lld %6lld %6lld | %6lld\n", r, t[1] - t[0], t[2] - t[1], t[3] - t[2], t[4] - t[3], nxt - t[0]);
      }
    }
    for (int i = 0; i < n; i++)
      for (int o = 0; o < cout; o++)
        for (int p = 0; p < ho * wo; p++) out[((size_t)i * cout + o) * ho * wo + p] = (float)hy[((size_t)i * ho * wo + p) * op + o];
  });
}
// Kernel micro-benchmark (not part of the drop-in surface): times nn::gemm on random data.
RT_API int rt_bench_gemm(rt_session* s, long long M, int K, int N, int variant, int iters, float* ms_out, float* maxdiff_out) {
  RT_REQUIRE(s && ms_out, s, "rt_bench_gemm: null argument");
  return guarded(s, [&] {
    RT_HIP_CHECK(hipSetDevice(s->device));
    // operand pitches as the networks have them (chan_pitch: 240 -> 256), padding channels zero
    const int Kp = round_up(K, 4), lda = chan_pitch(K), Np = round_up(N, 16), ldc = chan_pitch(N), nkc = (Kp + nn::KC - 1) / nn::KC;
    std::vector<float> ha((size_t)M * lda, 0.f), hw((size_t)nkc * Np * nn::KC, 0.f), hb(Np, 0.1f);
    uint64_t st = 0x2545F4914F6CDD1Dull;   // (xorshift64: 24 live significand bits per value -- the matrix pipe's clock depends on the data)
    auto rnd = [&]() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return (float)((double)(int64_t)(st >> 11) * (1.0 / 4503599627370496.0)) - 1.0f; };
    for (long long m = 0; m < M; m++) for (int k = 0; k < K; k++) ha[(size_t)m * lda + k] = rnd();
    for (int k = 0; k < K; k++) for (int n = 0; n < N; n++) hw[((size_t)(k / nn::KC) * Np + n) * nn::KC + k % nn::KC] = rnd() * 0.1f;
    DevBufs bufs;
    RestoreInt keep_variant(nn::g_gemm_variant);
    float *dA = bufs.alloc<float>(ha.size()), *dW = bufs.alloc<float>(hw.size()), *dB = bufs.alloc<float>(hb.size()),
          *dC = bufs.alloc<float>((size_t)M * ldc), *dC0 = bufs.alloc<float>((size_t)M * ldc);
    ForgetSplit forget{dW};
    RT_HIP_CHECK(hipMemset(dC, 0, (size_t)M * ldc * 4)); RT_HIP_CHECK(hipMemset(dC0, 0, (size_t)M * ldc * 4));
    RT_HIP_CHECK(hipMemcpy(dA, ha.data(), ha.size() * 4, hipMemcpyHostToDevice));
    RT_HIP_CHECK(hipMemcpy(dW, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
    RT_HIP_CHECK(hipMemcpy(dB, hb.data(), hb.size() * 4, hipMemcpyHostToDevice));
    Epilogue e{dB, ACT_HSWISH, 1, 1.01f, 0.02f, nullptr, 0};
    nn::g_gemm_variant = 1; nn::gemm(s->st, dA, lda, M, Kp, dW, N, Np, dC0, ldc, 0, e);
    nn::g_gemm_variant = variant;
    nn::gemm(s->st, dA, lda, M, Kp, dW, N, Np, dC, ldc, 0, e);
    Events ev; RT_HIP_CHECK(hipEventCreate(&ev.a)); RT_HIP_CHECK(hipEventCreate(&ev.b));
    hipEvent_t a = ev.a, b = ev.b;
    RT_HIP_CHECK(hipEventRecord(a, s->st));
    for (int i = 0; i < iters; i++) nn::gemm(s->st, dA, lda, M, Kp, dW, N, Np, dC, ldc, 0, e);
    RT_HIP_CHECK(hipEventRecord(b, s->st));
    RT_HIP_CHECK(hipStreamSynchronize(s->st));
    float ms = 0; RT_HIP_CHECK(hipEventElapsedTime(&ms, a, b)); *ms_out = ms / iters;
    if (maxdiff_out) {
      // the first and the last 4 M elements (the last row block is the partial one)
      const size_t total = (size_t)M * ldc, cnt = std::min<size_t>(total, (size_t)1 << 22);
      std::vector<float> c0(cnt), c1(cnt);
      float md = 0;
      for (size_t off : {(size_t)0, total - cnt}) {
        RT_HIP_CHECK(hipMemcpy(c0.data(), dC0 + off, cnt * 4, hipMemcpyDeviceToHost)); RT_HIP_CHECK(hipMemcpy(c1.data(), dC + off, cnt * 4, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < cnt; i++) { const float d = std::fabs(c0[i] - c1[i]); md = (d > md || d != d) ? (d != d ? INFINITY : d) : md; }
      }
      *maxdiff_out = md;
    }
  });
}

// Error of one nn::gemm variant against an fp64 product (round 6: the evidence behind the split-bf16 form): random operands with
// FULL 24-bit significands (a few binades each), bias 0, no activation, so the output is the bare product; the first `rows` rows are
// compared with sum_k (double)a * (double)w on the host.  out4 = {max |err|, rms err, max |ref|, rms ref}; variant as rt_bench_gemm
// (1 = narrow fp32-MFMA kernel, 30 = k_gemm32p, 40 = split-bf16).  seed != 0 reseeds the operands; act = an Act value.
RT_API int rt_bench_gemm_err(rt_session* s, long long M, int K, int N, int variant, int rows, int act, unsigned seed, double* out4) {
  RT_REQUIRE(s && out4 && M > 0 && K > 0 && N > 0 && rows > 0, s, "rt_bench_gemm_err: bad argument");
  return guarded(s, [&] {
    RT_HIP_CHECK(hipSetDevice(s->device));
    const int Kp = round_up(K, 4), lda = chan_pitch(K), Np = round_up(N, 16), ldc = chan_pitch(N), nkc = (Kp + nn::KC - 1) / nn::KC;
    std::vector<float> ha((size_t)M * lda, 0.f), hw((size_t)nkc * Np * nn::KC, 0.f), hwd((size_t)K * N), hb(Np, 0.f);
    uint64_t st = 0x9E3779B97F4A7C15ull ^ ((uint64_t)(seed ? seed : 1u) * 0xD1B54A32D192ED03ull);
    auto next = [&]() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return st; };
    // sign * (1 + 23 random bits) * 2^e, e in [-4, 0] (pixels) / [-6, -2] (weights): every significand bit is live
    auto rnd = [&](int e_hi) {
      const uint64_t v = next();
      const uint32_t bits = (uint32_t)((v >> 63) << 31) | (uint32_t)((127 + e_hi - (int)((v >> 40) % 5)) << 23) | (uint32_t)(v & 0x7fffff);
      float f; memcpy(&f, &bits, 4); return f;
    };
    for (long long m = 0; m < M; m++) for (int k = 0; k < K; k++) ha[(size_t)m * lda + k] = rnd(0);
    for (int k = 0; k < K; k++) for (int n = 0; n < N; n++) { const float w = rnd(-2); hwd[(size_t)k * N + n] = w; hw[((size_t)(k / nn::KC) * Np + n) * nn::KC + k % nn::KC] = w; }
    DevBufs bufs;
    RestoreInt keep_variant(nn::g_gemm_variant);
    float *dA = bufs.alloc<float>(ha.size()), *dW = bufs.alloc<float>(hw.size()), *dB = bufs.alloc<float>(hb.size()), *dC = bufs.alloc<float>((size_t)M * ldc);
    ForgetSplit forget{dW};
    RT_HIP_CHECK(hipMemset(dC, 0, (size_t)M * ldc * 4));
    RT_HIP_CHECK(hipMemcpy(dA, ha.data(), ha.size() * 4, hipMemcpyHostToDevice));
    RT_HIP_CHECK(hipMemcpy(dW, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
    RT_HIP_CHECK(hipMemcpy(dB, hb.data(), hb.size() * 4, hipMemcpyHostToDevice));
    Epilogue e; e.bias = dB; e.act = act;
    nn::g_gemm_variant = variant;
    nn::gemm(s->st, dA, lda, M, Kp, dW, N, Np, dC, ldc, 0, e);
    RT_HIP_CHECK(hipStreamSynchronize(s->st));
    // rows from the start, the middle and the end (the last row block is the partial one)
    const long long R = std::min<long long>(rows, M);
    std::vector<float> hc((size_t)R * ldc);
    double max_err = 0, sq_err = 0, max_ref = 0, sq_ref = 0; long long cnt = 0;
    for (int part = 0; part < 3; part++) {
      const long long r0 = part == 0 ? 0 : part == 1 ? std::max<long long>(0, M / 2 - R / 2) : M - R;
      RT_HIP_CHECK(hipMemcpy(hc.data(), dC + r0 * ldc, hc.size() * 4, hipMemcpyDeviceToHost));
      std::vector<double> ref(N);
      for (long long m = 0; m < R; m++) {
        std::fill(ref.begin(), ref.end(), 0.0);
        const float* a = &ha[(size_t)(r0 + m) * lda];
        for (int k = 0; k < K; k++) { const double av = a[k]; const float* w = &hwd[(size_t)k * N]; for (int n = 0; n < N; n++) ref[n] += av * (double)w[n]; }
        for (int n = 0; n < N; n++) {
          double rv = ref[n];
          if (act == ACT_HSWISH) rv = rv * std::min(std::max(rv + 3.0, 0.0), 6.0) / 6.0;
          else if (act == ACT_RELU) rv = std::max(rv, 0.0);
          const double d = std::fabs((double)hc[(size_t)m * ldc + n] - rv);
          if (!(d == d)) max_err = INFINITY;
          max_err = std::max(max_err, d); sq_err += d * d; max_ref = std::max(max_ref, std::fabs(rv)); sq_ref += rv * rv; cnt++;
        }
      }
    }
    out4[0] = max_err; out4[1] = std::sqrt(sq_err / cnt); out4[2] = max_ref; out4[3] = std::sqrt(sq_ref / cnt);
  });
}

// Kernel micro-benchmark: the fused thin LCNetV3 block (3x3 depthwise -> pointwise) on n images of h x w pixels, random data.
// form = nn::g_lc_wave for the timed launches: 0 = k_lc_thin (workgroup-staged; the unfused depthwise + GEMM pair where it has no
// instance), 1 = k_lc_wave (direct loads, stride 1), 3 = k_lc_lds (production), 5 = also the opt-in 128 -> 128 split; stride 21
// means (2, 1).  maxdiff compares with form RT_BENCH_LC_REF (default 0); RT_BENCH_LC_DUMP prints where the two differ.
RT_API int rt_bench_lc(rt_session* s, int n, int h, int w, int cin, int cout, int stride, int form, int iters, float* ms_out, float* maxdiff_out) {
  RT_REQUIRE(s && ms_out && n > 0 && h > 0 && w > 0 && (stride == 1 || stride == 2 || stride == 21), s, "rt_bench_lc: bad argument");
  return guarded(s, [&] {
    RT_HIP_CHECK(hipSetDevice(s->device));
    const int Cp = round_up(cin, 4), Np = round_up(cout, 16), ldy = chan_pitch(cout), nkc = (Cp + nn::KC - 1) / nn::KC;
    const int sh = stride == 21 ? 2 : stride, sw = stride == 21 ? 1 : stride;   // 21: stride (2, 1)
    const int ho = (h + sh - 1) / sh, wo = (w + sw - 1) / sw;
    std::vector<ImgGeom> gi(n), go(n);
    for (int i = 0; i < n; i++) { gi[i] = ImgGeom{(long long)i * h * w, h, w, 0}; go[i] = ImgGeom{(long long)i * ho * wo, ho, wo, 0}; }
    const size_t nin = (size_t)n * h * w * Cp, nout = (size_t)n * ho * wo * ldy;
    std::vector<float> hx(nin), hwd(9 * Cp), hbd(Cp, 0.05f), hw((size_t)nkc * Np * nn::KC, 0.f), hb(Np, 0.1f);
    uint32_t st = 777;
    auto rnd = [&]() { st = st * 1664525u + 1013904223u; return ((st >> 8) & 0xffff) / 32768.0f - 1.0f; };
    for (auto& v : hx) v = rnd();
    for (auto& v : hwd) v = rnd() * 0.3f;
    for (int k = 0; k < cin; k++) for (int c = 0; c < cout; c++) hw[((size_t)(k / nn::KC) * Np + c) * nn::KC + k % nn::KC] = rnd() * 0.1f;
    DevBufs bufs;
    RestoreInt keep_form(nn::g_lc_wave);
    float *dx = bufs.alloc<float>(nin), *dwd = bufs.alloc<float>(hwd.size()), *dbd = bufs.alloc<float>(hbd.size()), *dw = bufs.alloc<float>(hw.size()),
          *db = bufs.alloc<float>(hb.size()), *dy = bufs.alloc<float>(nout), *dy0 = bufs.alloc<float>(nout);
    ImgGeom *dgi = bufs.alloc<ImgGeom>(n), *dgo = bufs.alloc<ImgGeom>(n);
    RT_HIP_CHECK(hipMemset(dy, 0, nout * 4)); RT_HIP_CHECK(hipMemset(dy0, 0, nout * 4));
    RT_HIP_CHECK(hipMemcpy(dx, hx.data(), nin * 4, hipMemcpyHostToDevice)); RT_HIP_CHECK(hipMemcpy(dwd, hwd.data(), hwd.size() * 4, hipMemcpyHostToDevice));
    RT_HIP_CHECK(hipMemcpy(dbd, hbd.data(), hbd.size() * 4, hipMemcpyHostToDevice)); RT_HIP_CHECK(hipMemcpy(dw, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
    RT_HIP_CHECK(hipMemcpy(db, hb.data(), hb.size() * 4, hipMemcpyHostToDevice));
    RT_HIP_CHECK(hipMemcpy(dgi, gi.data(), n * sizeof(ImgGeom), hipMemcpyHostToDevice)); RT_HIP_CHECK(hipMemcpy(dgo, go.data(), n * sizeof(ImgGeom), hipMemcpyHostToDevice));
    Epilogue e; e.bias = db; e.act = ACT_HSWISH; e.has_lab = 1; e.lab_a = 1.01f; e.lab_c = 0.02f;
    const int dw_act = (sh == 2 && sw == 2) ? ACT_NONE : ACT_HSWISH, dw_lab = !(sh == 2 && sw == 2);   // (depthwise tail as in the LCNetV3 blocks)
    float* dy1 = nullptr;   // unfused reference: depthwise output
    auto run = [&](float* out) {
      if (nn::g_lc_wave == 0 && !nn::lc_thin_supported(3, sh, sw, Cp, cin, Np)) {   // no k_lc_thin instance: depthwise + GEMM
        if (!dy1) dy1 = bufs.alloc<float>((size_t)n * ho * wo * Cp);
        nn::dwconv(s->st, 3, sh, sw, dx, dgi, dgo, n, ho, wo, Cp, cin, dwd, dbd, dw_act, dw_lab, 0.99f, 0.01f, dy1, nullptr);
        nn::gemm(s->st, dy1, Cp, (long long)n * ho * wo, Cp, dw, cout, Np, out, ldy, 0, e);
        return;
      }
      nn::lc_thin(s->st, sh, sw, dx, dgi, dgo, n, ho, wo, Cp, cin, dwd, dbd, dw_act, dw_lab, 0.99f, 0.01f, dw, cout, Np, out, ldy, e);
    };
    nn::g_lc_wave = getenv("RT_BENCH_LC_REF") ? atoi(getenv("RT_BENCH_LC_REF")) : 0; run(dy0);
    nn::g_lc_wave = form; run(dy);
    Events ev; RT_HIP_CHECK(hipEventCreate(&ev.a)); RT_HIP_CHECK(hipEventCreate(&ev.b));
    hipEvent_t a = ev.a, b = ev.b;
    RT_HIP_CHECK(hipEventRecord(a, s->st));
    for (int i = 0; i < iters; i++) run(dy);
    RT_HIP_CHECK(hipEventRecord(b, s->st));
    RT_HIP_CHECK(hipStreamSynchronize(s->st));
    float ms = 0; RT_HIP_CHECK(hipEventElapsedTime(&ms, a, b)); *ms_out = ms / iters;
    if (maxdiff_out) {
      const size_t cnt = std::min<size_t>(nout, (size_t)1 << 22);
      std::vector<float> c0(cnt), c1(cnt);
      float md = 0;
      for (size_t off : {(size_t)0, nout - cnt}) {
        RT_HIP_CHECK(hipMemcpy(c0.data(), dy0 + off, cnt * 4, hipMemcpyDeviceToHost)); RT_HIP_CHECK(hipMemcpy(c1.data(), dy + off, cnt * 4, hipMemcpyDeviceToHost));
        size_t bad = 0;
        std::map<std::string, int> hist;
        for (size_t i = 0; i < cnt; i++) {
          const float d = std::fabs(c0[i] - c1[i]); md = (d > md || d != d) ? (d != d ? INFINITY : d) : md;
          if (d != 0 && getenv("RT_BENCH_LC_DUMP")) {
            const size_t e = off + i, px = e / ldy; const int ch = (int)(e % ldy), img = (int)(px / ((size_t)ho * wo)), yy = (int)(px % ((size_t)ho * wo)) / wo, xx = (int)(px % wo);
            if (bad++ < 4) fprintf(stderr, "  diff img %d y %d x %d ch %d: %g vs %g\n", img, yy, xx, ch, c0[i], c1[i]);
            hist["x%16=" + std::to_string(xx % 16)]++; hist["ch%4=" + std::to_string(ch % 4)]++; hist["ch/16=" + std::to_string(ch / 16)]++; hist["y%2=" + std::to_string(yy % 2)]++;
            hist["tx=" + std::to_string(xx / 16)]++; hist["q=" + std::to_string((ch % 16) / 4)]++;
          }
        }
        if (bad) { fprintf(stderr, "  %zu of %zu differ:", bad, cnt); for (auto& kv : hist) fprintf(stderr, " %s:%d", kv.first.c_str(), kv.second); fprintf(stderr, "\n"); }
      }
      *maxdiff_out = md;
    }
  });
}

}  // extern "C"
