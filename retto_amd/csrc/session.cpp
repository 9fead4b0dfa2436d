#include "session.h"

#include <charconv>
#include "onnx_import.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <sstream>
#include <thread>

#include <chrono>

#include "geom_math.h"

using namespace rt;

static const bool g_trace = getenv("RT_TRACE") != nullptr;
struct HostTick {
  std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
  void lap(const char* what) {
    if (!g_trace) return;
    auto n = std::chrono::steady_clock::now();
    fprintf(stderr, "[rt host] %-28s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(n - t).count());
    t = n;
  }
};

// ---------------------------------------------------------------------------
// RecCharacter::new (rec_processor.rs:29-46): String::from_utf8(bytes)? -> lines().map(|l| l.trim()) -> push " "
// -> insert "blank" at 0.  Rust semantics, restated exactly:
//  * String::from_utf8 is strict: overlong forms, surrogates (U+D800-DFFF) and code points above U+10FFFF
//    are errors (Utf8Error);
//  * str::lines splits after every '\n', drops that '\n' and one '\r' in front of it; a trailing '\n' does not
//    open a last empty line;
//  * str::trim strips every char with the Unicode White_Space property (char::is_whitespace): U+0009-000D,
//    U+0020, U+0085, U+00A0, U+1680, U+2000-200A, U+2028, U+2029, U+202F, U+205F, U+3000 -- a dictionary line
//    that holds only U+3000 becomes the empty string.
// ---------------------------------------------------------------------------
namespace rt {
static int utf8_decode_strict(const unsigned char* p, size_t n, uint32_t* cp) {  // bytes consumed, 0 = invalid
  if (n == 0) return 0;
  const unsigned char c = p[0];
  if (c < 0x80) { *cp = c; return 1; }
  auto cont = [&](size_t i) { return i < n && (p[i] & 0xC0) == 0x80; };
  if (c >= 0xC2 && c <= 0xDF) { if (!cont(1)) return 0; *cp = ((c & 0x1F) << 6) | (p[1] & 0x3F); return 2; }
  if (c >= 0xE0 && c <= 0xEF) {
    if (!cont(1) || !cont(2)) return 0;
    if (c == 0xE0 && p[1] < 0xA0) return 0;   // overlong
    if (c == 0xED && p[1] >= 0xA0) return 0;  // surrogate
    *cp = ((c & 0x0F) << 12) | ((p[1] & 0x3F) << 6) | (p[2] & 0x3F); return 3;
  }
  if (c >= 0xF0 && c <= 0xF4) {
    if (!cont(1) || !cont(2) || !cont(3)) return 0;
    if (c == 0xF0 && p[1] < 0x90) return 0;   // overlong
    if (c == 0xF4 && p[1] >= 0x90) return 0;  // above U+10FFFF
    *cp = ((c & 0x07) << 18) | ((p[1] & 0x3F) << 12) | ((p[2] & 0x3F) << 6) | (p[3] & 0x3F); return 4;
  }
  return 0;  // 0x80-0xC1 (continuation / overlong lead), 0xF5-0xFF
}
static bool is_rust_whitespace(uint32_t c) {
  return (c >= 0x09 && c <= 0x0D) || c == 0x20 || c == 0x85 || c == 0xA0 || c == 0x1680 || (c >= 0x2000 && c <= 0x200A) ||
         c == 0x2028 || c == 0x2029 || c == 0x202F || c == 0x205F || c == 0x3000;
}
std::vector<std::string> load_dictionary(const std::vector<uint8_t>& bytes) {
  const unsigned char* p = bytes.data();
  const size_t n = bytes.size();
  std::vector<std::string> dict;
  dict.push_back("blank");
  size_t pos = 0;
  while (pos < n) {
    // one line: [pos, e) without its terminator
    size_t e = pos;
    while (e < n && p[e] != '\n') e++;
    size_t le = e;
    if (e < n && le > pos && p[le - 1] == '\r') le--;
    // trim: first / last non-whitespace char, validating as we decode
    size_t first = std::string::npos, last_end = 0;
    for (size_t i = pos; i < le;) {
      uint32_t cp;
      int k = utf8_decode_strict(p + i, le - i, &cp);
      if (k == 0) throw RtError(RT_ERR_UTF8, "dictionary is not valid UTF-8 (byte offset " + std::to_string(i) + ")");
      if (!is_rust_whitespace(cp)) { if (first == std::string::npos) first = i; last_end = i + (size_t)k; }
      i += (size_t)k;
    }
    dict.push_back(first == std::string::npos ? std::string() : std::string((const char*)p + first, last_end - first));
    if (e >= n) break;
    pos = e + 1;
  }
  dict.push_back(" ");
  return dict;
}
}  // namespace rt

// ---------------------------------------------------------------------------
// construction (RettoSession::new, session.rs:62-73; RettoWorker::new, worker.rs:91-98)
// ---------------------------------------------------------------------------
rt_session* rt_session_create(const rt_config* cfg) {
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
    throw RtError(RT_ERR_BACKEND, "no HIP device visible: libretto_hip has no CPU fallback");
  if (cfg->device_id < 0 || cfg->device_id >= ndev) throw RtError(RT_ERR_INVALID, "device_id out of range");
  RT_HIP_CHECK(hipSetDevice(cfg->device_id));
  hipDeviceProp_t prop;
  RT_HIP_CHECK(hipGetDeviceProperties(&prop, cfg->device_id));
  if (std::string(prop.gcnArchName).rfind("gfx950", 0) != 0)
    throw RtError(RT_ERR_BACKEND, std::string("libretto_hip is built for gfx950 only, found ") + prop.gcnArchName);
  std::unique_ptr<rt_session> s(new rt_session());
  s->cfg = *cfg;
  s->device = cfg->device_id;
  // model sources are consumed here; do not keep caller pointers
  Blob bd = Blob::from_source(cfg->det.path, cfg->det.data, cfg->det.len, "det", MODEL_DET);
  Blob bc = Blob::from_source(cfg->cls.path, cfg->cls.data, cfg->cls.len, "cls", MODEL_CLS);
  Blob br = Blob::from_source(cfg->rec.path, cfg->rec.data, cfg->rec.len, "rec", MODEL_REC);
  std::vector<uint8_t> dict = read_source_bytes(cfg->dict.path, cfg->dict.data, cfg->dict.len, "dict");
  s->cfg.det = s->cfg.cls = s->cfg.rec = s->cfg.dict = rt_model_source{nullptr, nullptr, 0};
  RT_HIP_CHECK(hipStreamCreateWithFlags(&s->st_full, hipStreamNonBlocking));
  s->st = s->st_full;
  // which graph a source holds is read off its tensor names; rt_config.dtype picks the arithmetic
  const bool f16 = cfg->dtype == RT_DTYPE_F16;
  const bool sdet = blob_is_server_det(bd), srec = blob_is_server_rec(br);
  if ((sdet || srec) && !f16)
    throw RtError(RT_ERR_INVALID, "the PP-OCRv4 server graphs are built in fp16 only: set rt_config.dtype = RT_DTYPE_F16");
  if (sdet) s->det.reset(new DetServerH(bd)); else if (f16) s->det.reset(new DetNetH(bd)); else s->det.reset(new DetNet(bd));
  if (f16) s->cls.reset(new ClsNetH(bc)); else s->cls.reset(new ClsNet(bc));
  if (srec) s->rec.reset(new RecServerH(br)); else if (f16) s->rec.reset(new RecNetH(br)); else s->rec.reset(new RecNet(br));
  s->model_info = std::string(s->det->arch()) + "/" + s->det->dtype() + " " + s->cls->dtype() + " " + s->rec->arch() + "/" + s->rec->dtype();
  s->dict = rt::load_dictionary(dict);
  if ((int)s->dict.size() != s->rec->classes())
    throw RtError(RT_ERR_SHAPE, "dictionary has " + std::to_string(s->dict.size()) + " entries but the rec head has " +
                                    std::to_string(s->rec->classes()) + " classes");
  RT_HIP_CHECK(hipMalloc((void**)&s->d_flags, 64));
  RT_HIP_CHECK(hipMemset(s->d_flags, 0, 64));
  RT_HIP_CHECK(hipEventCreateWithFlags(&s->ev_block, hipEventBlockingSync | hipEventDisableTiming));
  const int lanes = cfg->lanes > 0 ? cfg->lanes : 3;  // measured on C3: 1 -> 640, 2 -> 681, 3 -> 700, 4 -> 652 images/s
  for (int l = 1; l < lanes; l++) {
    std::unique_ptr<rt_session> h(new rt_session());
    h->cfg = s->cfg; h->device = s->device;
    h->det = s->det; h->cls = s->cls; h->rec = s->rec; h->dict = s->dict; h->model_info = s->model_info;
    RT_HIP_CHECK(hipStreamCreateWithFlags(&h->st_full, hipStreamNonBlocking));
    h->st = h->st_full;
    RT_HIP_CHECK(hipMalloc((void**)&h->d_flags, 64));
    RT_HIP_CHECK(hipMemset(h->d_flags, 0, 64));
    RT_HIP_CHECK(hipEventCreateWithFlags(&h->ev_block, hipEventBlockingSync | hipEventDisableTiming));
    s->helpers.push_back(std::move(h));
  }
  // CU partitions (RT_LANE_CUMASK=1, A/B only): every lane of a multi-lane session gets its own slice of every XCD.  Measured
  // slower than whole-device streams on C3 (runtime.h "CU partitions"), so the default keeps the lanes time-slicing the chip.
  static const bool part_on = getenv("RT_LANE_CUMASK") && atoi(getenv("RT_LANE_CUMASK")) != 0;
  if (part_on && lanes > 1) {
    std::vector<hipStream_t> ps((size_t)lanes, nullptr);
    std::vector<int> pc((size_t)lanes, 0);
    std::vector<std::vector<unsigned>> ids((size_t)lanes);
    bool ok = true;
    for (int l = 0; l < lanes && ok; l++) ok = (ps[(size_t)l] = rt::partition_stream(l, lanes, &pc[(size_t)l], &ids[(size_t)l])) != nullptr;
    for (int a = 0; a < lanes && ok; a++)
      for (int b = a + 1; b < lanes && ok; b++)
        for (unsigned v : ids[(size_t)a]) if (std::find(ids[(size_t)b].begin(), ids[(size_t)b].end(), v) != ids[(size_t)b].end()) { ok = false; break; }
    if (ok) {
      for (int l = 0; l < lanes; l++) {
        rt_session* ln = l == 0 ? s.get() : s->helpers[(size_t)l - 1].get();
        ln->st_part = ps[(size_t)l]; ln->part_cus = pc[(size_t)l];
      }
    } else {
      for (hipStream_t p : ps) if (p) { rt::forget_stream(p); (void)hipStreamDestroy(p); }
      if (getenv("RT_TRACE")) fprintf(stderr, "[rt] CU partitions could not be verified on this device: lanes keep whole-device streams\n");
    }
  }
  return s.release();
}

void rt_session::begin_call() {
  RT_HIP_CHECK(hipSetDevice(device));
  (void)hipGetLastError();   // HIP's last error is sticky per host thread: a failure of the PREVIOUS call on this thread (a refused
                             // hipMalloc, say) must not be what the first RT_LAUNCH of this call reports
  if (failed) { arena.abandon_pass(); scratch.abandon_pass(); dbws.abandon_pass(); failed = false; }
  arena.reset(); scratch.reset(); pinned.reset(); dbws.reset();   // (dbws too: its mark below must see the previous pass folded in)
  arena.mark_call(); scratch.mark_call(); dbws.mark_call();
  // (last_error belongs to the API caller's thread -- api.cpp guarded(); lane threads run this function and never touch it)
}
// Waiting for the lane's stream.  hipStreamSynchronize spins on the CPU (HIP's default scheduling when there are more CPUs than
// GPUs): three lanes = three cores at 100 % per rank for the whole step (measured: 3.9 cores busy per rank), which eight ranks
// on the GPU box's 16-CPU pod do not have.  Here: a blocking-sync event is recorded behind the work, polled for a short while (a
// wait that ends within ~50 us -- single pages, metadata copies -- pays no sleep / wake-up), then polled every 100 us with the
// thread asleep in between.  RT_SYNC_SPIN=1 restores hipStreamSynchronize.
static const bool g_sync_spin = getenv("RT_SYNC_SPIN") && atoi(getenv("RT_SYNC_SPIN")) != 0;
void rt_session::sync() {
  if (g_sync_spin || !ev_block) {
    RT_HIP_CHECK(hipStreamSynchronize(st));
  } else {
    RT_HIP_CHECK(hipEventRecord(ev_block, st));
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
      const hipError_t q = hipEventQuery(ev_block);
      if (q == hipSuccess) break;
      if (q != hipErrorNotReady) RT_HIP_CHECK(q);
      // (hipEventSynchronize on a hipEventBlockingSync event still kept the thread at 100 % of a core on this ROCm: measured
      //  3.0 cores busy with it, 3.9 with hipStreamSynchronize; so the sleeping is done here: poll, sleep 100 us, poll ...)
      //  The spin window is chosen per call: 50 us for multi-page throughput batches, the whole wait (up to 5 ms) for the
      //  single-page / stage-level calls, whose several sync points per call would otherwise pay ~150 us of sleep each.)
      if (std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(spin_us)) std::this_thread::sleep_for(std::chrono::microseconds(100));
    }
  }
  if (prof.on) prof.collect();
}
void rt_session::check_flags() {
  int f[2] = {0, 0};
  RT_HIP_CHECK(hipMemcpy(f, d_flags, sizeof(f), hipMemcpyDeviceToHost));
  if (f[0]) {
    RT_HIP_CHECK(hipMemset(d_flags, 0, 64));
    throw RtError(RT_ERR_IMAGE, "thumbnail sampled outside the source image (the reference panics here)");
  }
}

static std::vector<std::pair<int, int>> uniform_hw(int n, int h, int w) {
  return std::vector<std::pair<int, int>>((size_t)n, std::make_pair(h, w));
}

// ---------------------------------------------------------------------------
// L1 worker functions
// ---------------------------------------------------------------------------
void rt_session::det_forward(const float* nchw, int n, int h, int w, float* out) {
  begin_call();
  size_t ne = (size_t)n * 3 * h * w;
  float* d_in = arena.alloc<float>(ne);
  RT_HIP_CHECK(hipMemcpyAsync(d_in, nchw, ne * 4, hipMemcpyHostToDevice, st));
  float* x = arena.alloc<float>((size_t)n * h * w * 4);
  nn::nchw3_to_nhwc4(st, d_in, n, h, w, x);
  Level L0 = make_level(uniform_hw(n, h, w));
  RunCtx c = ctx(&scratch);
  float* map = det->run(c, x, L0);
  RT_HIP_CHECK(hipMemcpyAsync(out, map, (size_t)n * h * w * 4, hipMemcpyDeviceToHost, st));
  sync();
}
void rt_session::cls_forward(const float* nchw, int n, int h, int w, float* out) {
  begin_call();
  size_t ne = (size_t)n * 3 * h * w;
  float* d_in = arena.alloc<float>(ne);
  RT_HIP_CHECK(hipMemcpyAsync(d_in, nchw, ne * 4, hipMemcpyHostToDevice, st));
  float* x = arena.alloc<float>((size_t)n * h * w * 4);
  nn::nchw3_to_nhwc4(st, d_in, n, h, w, x);
  Level L0 = make_level(uniform_hw(n, h, w));
  RunCtx c = ctx(&scratch);
  float* probs = cls->run(c, x, L0);
  RT_HIP_CHECK(hipMemcpyAsync(out, probs, (size_t)n * 2 * 4, hipMemcpyDeviceToHost, st));
  sync();
}
void rt_session::rec_forward(const float* nchw, int n, int h, int w, float* out, int* t_out) {
  int T = RecNet::tokens_for_width(w);
  if (t_out) *t_out = T;
  if (!out) return;
  begin_call();
  size_t ne = (size_t)n * 3 * h * w;
  float* d_in = arena.alloc<float>(ne);
  RT_HIP_CHECK(hipMemcpyAsync(d_in, nchw, ne * 4, hipMemcpyHostToDevice, st));
  float* x = arena.alloc<float>((size_t)n * h * w * 4);
  nn::nchw3_to_nhwc4(st, d_in, n, h, w, x);
  Level L0 = make_level(uniform_hw(n, h, w)), Lt;
  RunCtx c = ctx(&scratch);
  float* logits = rec->run(c, x, L0, Lt);
  const int C = rec->classes();
  float* probs = scratch.alloc<float>((size_t)Lt.total * C);
  nn::softmax_rows(st, logits, rec->logits_ld(), Lt.total, C, probs);
  RT_HIP_CHECK(hipMemcpyAsync(out, probs, (size_t)Lt.total * C * 4, hipMemcpyDeviceToHost, st));
  sync();
}

// Ragged form of rec_forward: line i is [3,48,widths[i]] (NCHW, lines concatenated), every line keeps its own width inside ONE
// launch series -- what rt_run_batch's rec groups do (rec_processor.rs:214-270 gives each batch of 6 its width; the session
// concatenates the batches).  out = the lines' [T_i, classes] probabilities concatenated, t_out[i] = T_i.
void rt_session::rec_forward_ragged(const float* nchw, int n, const int* widths, float* out, int* t_out) {
  const int h = 48;
  size_t ne = 0, nt = 0;
  for (int i = 0; i < n; i++) {
    if (widths[i] < 8) throw RtError(RT_ERR_SHAPE, "rt_rec_ragged: every line must be at least 8 wide");
    ne += (size_t)3 * h * widths[i];
    int T = RecNet::tokens_for_width(widths[i]);
    if (t_out) t_out[i] = T;
    nt += (size_t)T;
  }
  if (!out) return;
  begin_call();
  float* d_in = arena.alloc<float>(ne);
  RT_HIP_CHECK(hipMemcpyAsync(d_in, nchw, ne * 4, hipMemcpyHostToDevice, st));
  float* x = arena.alloc<float>(ne / 3 * 4);
  std::vector<std::pair<int, int>> hw;
  size_t off = 0;
  for (int i = 0; i < n; i++) {
    nn::nchw3_to_nhwc4(st, d_in + off * 3, 1, h, widths[i], x + off * 4);
    off += (size_t)h * widths[i];
    hw.emplace_back(h, widths[i]);
  }
  Level L0 = make_level(hw), Lt;
  RunCtx c = ctx(&scratch);
  float* logits = rec->run(c, x, L0, Lt);
  if ((size_t)Lt.total != nt) throw RtError(RT_ERR_BACKEND, "rt_rec_ragged: token count mismatch");
  const int C = rec->classes();
  float* probs = scratch.alloc<float>((size_t)Lt.total * C);
  nn::softmax_rows(st, logits, rec->logits_ld(), Lt.total, C, probs);
  RT_HIP_CHECK(hipMemcpyAsync(out, probs, (size_t)Lt.total * C * 4, hipMemcpyDeviceToHost, st));
  sync();
}

// ---------------------------------------------------------------------------
// stage functions
// ---------------------------------------------------------------------------
// image_helper.rs:106-148 on a device image; returns the (possibly new) device buffer
static const uint8_t* dev_resize_both(rt_session* s, const uint8_t* img, int h, int w, int* oh, int* ow) {
  int plan[4];
  int n = gm::resize_both_plan(h, w, s->cfg.max_side_len, s->cfg.min_side_len, plan);
  const uint8_t* cur = img; int ch = h, cw = w;
  for (int i = 0; i < n; i++) {
    int nh = plan[2 * i], nw = plan[2 * i + 1];
    uint8_t* dst = s->arena.alloc<uint8_t>(std::max<size_t>((size_t)nh * nw * 3, 4));
    ProfScope ps(&s->prof, s->st, "thumbnail");
    pp::thumbnail_rgb8(s->st, cur, ch, cw, dst, nh, nw, s->d_flags);
    cur = dst; ch = nh; cw = nw;
  }
  *oh = ch; *ow = cw;
  return cur;
}

void rt_session::resize_both(const uint8_t* rgb, int h, int w, uint8_t* out, int oh, int ow) {
  begin_call();
  uint8_t* d = arena.alloc<uint8_t>((size_t)h * w * 3);
  RT_HIP_CHECK(hipMemcpyAsync(d, rgb, (size_t)h * w * 3, hipMemcpyHostToDevice, st));
  int rh, rw;
  const uint8_t* r = dev_resize_both(this, d, h, w, &rh, &rw);
  if (rh != oh || rw != ow) throw RtError(RT_ERR_SHAPE, "resize_both: output buffer dims do not match rt_resize_both_dims");
  RT_HIP_CHECK(hipMemcpyAsync(out, r, (size_t)rh * rw * 3, hipMemcpyDeviceToHost, st));
  sync(); check_flags();
}

void rt_session::det_preprocess(const uint8_t* rgb, int h, int w, float* out) {
  begin_call();
  uint8_t* d = arena.alloc<uint8_t>((size_t)h * w * 3);
  RT_HIP_CHECK(hipMemcpyAsync(d, rgb, (size_t)h * w * 3, hipMemcpyHostToDevice, st));
  int dh, dw;
  gm::resize_either_dims(h, w, cfg.det_limit_type, cfg.det_limit_side_len, &dh, &dw);
  if (dh <= 0 || dw <= 0) throw RtError(RT_ERR_SHAPE, "det input collapses to zero size");
  uint8_t* r = arena.alloc<uint8_t>((size_t)dh * dw * 3);
  pp::thumbnail_rgb8(st, d, h, w, r, dh, dw, d_flags);
  float* o = arena.alloc<float>((size_t)dh * dw * 3);
  pp::det_normalize(st, r, dh, dw, cfg.det_scale, cfg.det_mean, cfg.det_std, 1, o);
  RT_HIP_CHECK(hipMemcpyAsync(out, o, (size_t)dh * dw * 3 * 4, hipMemcpyDeviceToHost, st));
  sync(); check_flags();
}

static pp::DbParams db_params(const rt_config& c) {
  pp::DbParams p;
  p.thresh = c.det_thresh; p.box_thresh = c.det_box_thresh; p.unclip_ratio = c.det_unclip_ratio;
  p.min_size = c.det_min_mini_box_size; p.dilate = c.det_dilation;
  return p;
}
static int max_boxes_of(const rt_config& c) { return c.max_boxes_per_page > 0 ? c.max_boxes_per_page : 8192; }

void rt_session::det_postprocess(const float* pred, int h, int w, int ori_h, int ori_w, float* boxes, float* scores,
                                 int max_out, int* n_out) {
  begin_call();
  float* d = arena.alloc<float>((size_t)h * w);
  RT_HIP_CHECK(hipMemcpyAsync(d, pred, (size_t)h * w * 4, hipMemcpyHostToDevice, st));
  const int mb = max_boxes_of(cfg);
  void* ws = dbws.alloc_bytes(pp::db_workspace_bytes(h, w, mb));
  pp::DbBox* db = arena.alloc<pp::DbBox>(mb);
  int* cnt = arena.alloc<int>(2);
  {
    pp::DbPageIn in{d, h, w, ori_h, ori_w};
    void* hd = pinned.alloc_bytes(pp::db_page_desc_bytes());
    void* dd = arena.alloc_bytes(pp::db_page_desc_bytes());
    pp::db_postprocess_batch(st, 1, &in, db_params(cfg), &ws, mb, &db, &cnt, hd, dd);
  }
  int hc[2];
  RT_HIP_CHECK(hipMemcpyAsync(hc, cnt, 8, hipMemcpyDeviceToHost, st));
  sync();
  if (hc[1]) throw RtError(RT_ERR_CAPACITY, "DB post-processing work list overflow (raise max_boxes_per_page)");
  *n_out = hc[0];
  int n = std::min(hc[0], max_out);
  std::vector<pp::DbBox> hb((size_t)std::max(n, 1));
  if (n > 0) RT_HIP_CHECK(hipMemcpy(hb.data(), db, (size_t)n * sizeof(pp::DbBox), hipMemcpyDeviceToHost));
  for (int i = 0; i < n; i++) { memcpy(boxes + 8 * i, hb[i].pts, 32); scores[i] = hb[i].score; }
}

struct CropPlan {
  std::vector<pp::CropDesc> descs;
  std::vector<pp::CropRef> refs;
  size_t pool_bytes = 0;
  int max_pix = 0;
};
// image_helper.rs:223-249 planning for the boxes of one page (src = device page after resize_both)
static void plan_crops(CropPlan& plan, const uint8_t* src, int sh, int sw, const float* boxes, int n) {
  for (int i = 0; i < n; i++) {
    const float* b = boxes + 8 * i;
    gm::CropDims d = gm::crop_dims(b);
    if (d.w <= 0 || d.h <= 0) throw RtError(RT_ERR_IMAGE, "zero-sized crop");
    pp::CropDesc cd;
    cd.src = src; cd.sh = sh; cd.sw = sw; cd.w = d.w; cd.h = d.h; cd.rot = d.rot;
    if (!gm::projection_inverse(b, d.cw, d.ch, cd.inv))
      throw RtError(RT_ERR_IMAGE, "singular crop homography (Projection::from_control_points -> None; the reference unwraps)");
    cd.out_off = (long long)plan.pool_bytes;
    pp::CropRef r; r.off = cd.out_off; r.h = d.rot ? d.w : d.h; r.w = d.rot ? d.h : d.w; r.pad_ = 0;
    plan.descs.push_back(cd); plan.refs.push_back(r);
    plan.pool_bytes += ((size_t)d.w * d.h * 3 + 63) & ~(size_t)63;
    plan.max_pix = std::max(plan.max_pix, d.w * d.h);
  }
}

void rt_session::crop_images(const uint8_t* rgb, int h, int w, const float* boxes, int n, uint8_t* out, size_t out_cap) {
  begin_call();
  uint8_t* d = arena.alloc<uint8_t>((size_t)h * w * 3);
  RT_HIP_CHECK(hipMemcpyAsync(d, rgb, (size_t)h * w * 3, hipMemcpyHostToDevice, st));
  CropPlan plan;
  plan_crops(plan, d, h, w, boxes, n);
  size_t need = 0;
  for (auto& r : plan.refs) need += (size_t)r.h * r.w * 3;
  if (need > out_cap) throw RtError(RT_ERR_INVALID, "crop_images: output buffer too small");
  uint8_t* pool = arena.alloc<uint8_t>(std::max<size_t>(plan.pool_bytes, 64));
  pp::CropDesc* dd = arena.alloc<pp::CropDesc>(std::max(n, 1));
  if (n > 0) RT_HIP_CHECK(hipMemcpyAsync(dd, plan.descs.data(), (size_t)n * sizeof(pp::CropDesc), hipMemcpyHostToDevice, st));
  pp::warp_crops(st, dd, n, plan.max_pix, pool);
  size_t o = 0;
  for (int i = 0; i < n; i++) {
    size_t bytes = (size_t)plan.refs[i].h * plan.refs[i].w * 3;
    RT_HIP_CHECK(hipMemcpyAsync(out + o, pool + plan.refs[i].off, bytes, hipMemcpyDeviceToHost, st));
    o += bytes;
  }
  sync();
}

void rt_session::resize_norm_image(const uint8_t* crop, int h, int w, int ori_h, int ori_w, int img_h, int img_w,
                                   float ratio, float* out) {
  begin_call();
  uint8_t* d = arena.alloc<uint8_t>(std::max<size_t>((size_t)h * w * 3, 4));
  RT_HIP_CHECK(hipMemcpyAsync(d, crop, (size_t)h * w * 3, hipMemcpyHostToDevice, st));
  pp::LineDesc L;
  L.crop_off = 0; L.h = h; L.w = w;
  L.W = gm::resize_norm_width(img_h, img_w, ratio);
  L.resized_w = gm::resize_norm_resized_w(img_h, L.W, ori_h, ori_w);
  L.out_off = 0;
  pp::LineDesc* dl = arena.alloc<pp::LineDesc>(1);
  RT_HIP_CHECK(hipMemcpyAsync(dl, &L, sizeof(L), hipMemcpyHostToDevice, st));
  float* o = arena.alloc<float>((size_t)3 * img_h * std::max(L.W, 1));
  pp::resize_norm(st, dl, 1, img_h, L.W, d, 1, o, d_flags);
  RT_HIP_CHECK(hipMemcpyAsync(out, o, (size_t)3 * img_h * L.W * 4, hipMemcpyDeviceToHost, st));
  sync(); check_flags();
}

void rt_session::ctc_decode(const float* probs, int n, int t, int c, int32_t* idx, float* prob, int32_t* tokens,
                            int32_t* n_tokens, float* scores) {
  begin_call();
  size_t rows = (size_t)n * t;
  float* d = arena.alloc<float>(rows * c);
  RT_HIP_CHECK(hipMemcpyAsync(d, probs, rows * c * 4, hipMemcpyHostToDevice, st));
  int* di = arena.alloc<int>(rows); float* dp = arena.alloc<float>(rows);
  int* dt = arena.alloc<int>(rows); int* dn = arena.alloc<int>(n); float* ds = arena.alloc<float>(n);
  // per (n,t): first argmax and max of the probabilities themselves (rec_processor.rs:198-199)
  Level Lt = make_level(uniform_hw(n, 1, t));
  RunCtx cx = ctx(&arena);
  upload_levels(cx, {&Lt});
  nn::argmax_rows(st, d, c, (long long)rows, c, di, dp);
  pp::ctc_decode(st, di, dp, Lt.d, n, dt, dn, ds);
  RT_HIP_CHECK(hipMemcpyAsync(idx, di, rows * 4, hipMemcpyDeviceToHost, st));
  RT_HIP_CHECK(hipMemcpyAsync(prob, dp, rows * 4, hipMemcpyDeviceToHost, st));
  RT_HIP_CHECK(hipMemcpyAsync(tokens, dt, rows * 4, hipMemcpyDeviceToHost, st));
  RT_HIP_CHECK(hipMemcpyAsync(n_tokens, dn, (size_t)n * 4, hipMemcpyDeviceToHost, st));
  RT_HIP_CHECK(hipMemcpyAsync(scores, ds, (size_t)n * 4, hipMemcpyDeviceToHost, st));
  sync();
}

// ---------------------------------------------------------------------------
// L2: process_pipeline over a batch of pages
// ---------------------------------------------------------------------------
namespace {
struct PageState {
  int ori_h, ori_w, after_h, after_w, det_h, det_w;
  const uint8_t* img;  // device, after resize_both
  const float* map;    // device det map used for boxes
  pp::DbBox* d_boxes; int* d_count;
  int n_boxes = 0; int first_line = 0;
  std::vector<pp::DbBox> boxes;
};

std::string json_escape(const std::string& s) {
  std::string o;
  for (char ch : s) {
    switch (ch) {
      case '"': o += "\\\""; break;
      case '\\': o += "\\\\"; break;
      case '\n': o += "\\n"; break;
      case '\r': o += "\\r"; break;
      case '\t': o += "\\t"; break;
      default:
        if ((unsigned char)ch < 0x20) { char b[8]; snprintf(b, sizeof b, "\\u%04x", ch); o += b; }
        else o += ch;
    }
  }
  return o;
}
// serde_json writes a finite f32 through ryu: the shortest decimal string that parses back to the same f32
// (0.9f -> "0.9", not "0.899999976"), laid out by ryu's f32 rules -- plain decimals while the decimal point
// position kk is in (-6, 13] ("123.0", "0.00001234"), otherwise d[.ddd]e[-]x ("1e30", "1.234e-7").
std::string fnum(float v) {
  if (v != v || std::isinf(v)) return "null";  // serde_json writes non-finite floats as null
  if (v == 0.0f) return std::signbit(v) ? "-0.0" : "0.0";
  // std::to_chars (scientific, no precision) yields exactly the shortest digit string that round-trips, and -- unlike
  // snprintf / strtof -- does not depend on LC_NUMERIC (a host that called setlocale() with a comma-decimal locale would
  // otherwise get invalid JSON)
  char b[48];
  const std::to_chars_result tr = std::to_chars(b, b + sizeof b - 1, v, std::chars_format::scientific);
  *tr.ptr = 0;
  std::string digits; int exp10 = 0; bool neg = false;
  {
    const char* p = b;
    if (*p == '-') { neg = true; p++; }
    for (; *p && *p != 'e'; p++) if (*p != '.') digits += *p;
    exp10 = atoi(p + 1);
  }
  while (digits.size() > 1 && digits.back() == '0') digits.pop_back();
  const int len = (int)digits.size();
  const int k = exp10 - (len - 1);  // value = digits * 10^k
  const int kk = len + k;           // position of the decimal point
  std::string o = neg ? "-" : "";
  if (0 <= k && kk <= 13) { o += digits; o.append((size_t)k, '0'); o += ".0"; }
  else if (0 < kk && kk <= 13) { o += digits.substr(0, (size_t)kk); o += '.'; o += digits.substr((size_t)kk); }
  else if (-6 < kk && kk <= 0) { o += "0."; o.append((size_t)(-kk), '0'); o += digits; }
  else {
    o += digits[0];
    if (len > 1) { o += '.'; o += digits.substr(1); }
    o += 'e'; o += std::to_string(kk - 1);
  }
  return o;
}
}  // namespace
std::string rt_format_f32_impl(float v) { return fnum(v); }

rt_results* rt_session::run_pages(const uint8_t* const* rgb, const int* hs, const int* ws, int n_pages, int mem,
                                  const float* const* det_map_override) {
  if (g_trace) fprintf(stderr, "[rt host] %-28s %8.3f ms\n", "between calls", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - last_exit).count());
  HostTick tick0;
  // a one-page call is the reference's real mode (retto-cli/src/main.rs:80-86): it polls through its waits; a multi-page batch
  // is a throughput run whose lane threads must not hold a core each (8 ranks x 3 lanes on a 16-CPU pod)
  struct SpinScope { int& r; int old; ~SpinScope() { r = old; } } spin_scope{spin_us, spin_us};
  // (a lane worker's one-page part -- a batch of <= lanes pages, or several one-page submissions in flight -- is a throughput run
  //  too: it polls for a bounded 250 us per wait, not 5 ms)
  spin_us = n_pages > 1 ? 50 : on_lane_worker ? 250 : 5000;
  begin_call();
  tick0.lap("begin_call + arena reset");
  std::unique_ptr<rt_results> res(new rt_results());
  res->pages.resize((size_t)n_pages);
  if (n_pages == 0) return res.release();
  std::vector<PageState> pg((size_t)n_pages);
  const int mb = max_boxes_of(cfg);
  pp::DbBox *d_boxes_all = nullptr, *d_boxes_packed = nullptr;
  int* d_counts_all = nullptr;

  // ---- a2 + a3: size limits, det resize, normalise ---------------------------------
  std::vector<std::pair<int, int>> det_hw;
  std::vector<const uint8_t*> det_img((size_t)n_pages);
  for (int i = 0; i < n_pages; i++) {
    PageState& p = pg[i];
    p.ori_h = hs[i]; p.ori_w = ws[i];
    if (hs[i] <= 0 || ws[i] <= 0 || rgb[i] == nullptr) throw RtError(RT_ERR_IMAGE, "empty page");
    const uint8_t* raw = rgb[i];
    if (mem == RT_MEM_HOST || mem == RT_MEM_HOST_MAPS_DEVICE) {   // the pages cross PCIe here (a submitted batch staged them already)
      uint8_t* d = arena.alloc<uint8_t>((size_t)hs[i] * ws[i] * 3);
      RT_HIP_CHECK(hipMemcpyAsync(d, rgb[i], (size_t)hs[i] * ws[i] * 3, hipMemcpyHostToDevice, st));
      raw = d;
    }
    p.img = dev_resize_both(this, raw, hs[i], ws[i], &p.after_h, &p.after_w);
    if (p.after_h <= 0 || p.after_w <= 0) throw RtError(RT_ERR_SHAPE, "page collapses to zero size in resize_both");
    gm::resize_either_dims(p.after_h, p.after_w, cfg.det_limit_type, cfg.det_limit_side_len, &p.det_h, &p.det_w);
    if (p.det_h <= 0 || p.det_w <= 0) throw RtError(RT_ERR_SHAPE, "det input collapses to zero size");
    if (p.det_h == p.after_h && p.det_w == p.after_w) det_img[i] = p.img;  // thumbnail at ratio 1 is the identity
    else {
      uint8_t* d = arena.alloc<uint8_t>((size_t)p.det_h * p.det_w * 3);
      ProfScope ps(&prof, st, "thumbnail");
      pp::thumbnail_rgb8(st, p.img, p.after_h, p.after_w, d, p.det_h, p.det_w, d_flags);
      det_img[i] = d;
    }
    det_hw.push_back({p.det_h, p.det_w});
  }

  HostTick tick;
  tick.lap("pre (resize, upload)");
  // ---- a4: det network in launch groups -------------------------------------------
  long long group_px = cfg.det_sub_batch > 0 ? 0 : (long long)32 * 960 * 960;  // measured: larger launch groups win (launch-bound small layers)
  std::vector<double*> sum_parts; std::vector<int> sum_counts;
  for (int g0 = 0; g0 < n_pages;) {
    int g1 = g0; long long px = 0;
    while (g1 < n_pages) {
      long long add = (long long)det_hw[g1].first * det_hw[g1].second;
      if (g1 > g0 && ((cfg.det_sub_batch > 0 && g1 - g0 >= cfg.det_sub_batch) || (cfg.det_sub_batch <= 0 && px + add > group_px))) break;
      px += add; g1++;
    }
    std::vector<std::pair<int, int>> hw(det_hw.begin() + g0, det_hw.begin() + g1);
    Level L0 = make_level(hw);
    scratch.rewind();
    // a3 normalise is folded into the det stem (DetNet::run_u8): the pages stay RGB8 until the first conv reads them
    const int gn = g1 - g0;
    pp::NormDesc* hd = pinned.alloc<pp::NormDesc>((size_t)gn);
    pp::NormDesc* dd = scratch.alloc<pp::NormDesc>((size_t)gn);
    for (int i = g0; i < g1; i++)
      hd[i - g0] = pp::NormDesc{det_img[i], (long long)pg[i].det_h * pg[i].det_w, L0.h[i - g0].off};
    RT_HIP_CHECK(hipMemcpyAsync(dd, hd, (size_t)gn * sizeof(pp::NormDesc), hipMemcpyHostToDevice, st));
    RunCtx c = ctx(&scratch);
    float* map;
    static const bool f32_input = getenv("RT_DET_F32_INPUT") != nullptr;  // A/B: build the normalised tensor first
    if (f32_input) {
      float* x = scratch.alloc<float>((size_t)L0.total * 4);
      long long max_pix = 0;
      for (int i = 0; i < gn; i++) max_pix = std::max(max_pix, hd[i].npix);
      { ProfScope ps(&prof, st, "det_normalize");
        pp::det_normalize_batch(st, dd, gn, max_pix, cfg.det_scale, cfg.det_mean, cfg.det_std, x); }
      ProfOuter po(&prof, st, "net/det"); map = det->run(c, x, L0);
    } else {
      ProfOuter po(&prof, st, "net/det"); map = det->run_u8(c, dd, cfg.det_scale, cfg.det_mean, cfg.det_std, L0);
    }
    // keep the maps beyond the next group's scratch rewind; the last group's map is consumed by the DB post-processing
    // (same stream) before the classifier reuses the scratch arena, so it stays where the network left it
    float* keep = map;
    if (g1 < n_pages) {
      keep = arena.alloc<float>((size_t)L0.total);
      RT_HIP_CHECK(hipMemcpyAsync(keep, map, (size_t)L0.total * 4, hipMemcpyDeviceToDevice, st));
    }
    for (int i = g0; i < g1; i++) pg[i].map = keep + L0.h[i - g0].off;
    int nb = pp::sum_blocks(L0.total);
    double* parts = arena.alloc<double>(nb);
    pp::sum_partial(st, keep, L0.total, parts);
    sum_parts.push_back(parts); sum_counts.push_back(nb);
    g0 = g1;
  }

  tick.lap("det enqueue");
  // ---- a5: DB post-processing per page (on device; stream-ordered workspace reuse) --
  {
    std::vector<pp::DbPageIn> in((size_t)n_pages);
    std::vector<void*> wsp((size_t)n_pages);
    std::vector<pp::DbBox*> bo((size_t)n_pages);
    std::vector<int*> co((size_t)n_pages);
    // box lists and counts of all pages are contiguous: one pack launch + two copies bring them to the host
    d_boxes_all = arena.alloc<pp::DbBox>((size_t)mb * std::max(n_pages, 1));
    d_counts_all = arena.alloc<int>((size_t)2 * std::max(n_pages, 1));
    d_boxes_packed = arena.alloc<pp::DbBox>((size_t)mb * std::max(n_pages, 1));
    for (int i = 0; i < n_pages; i++) {
      PageState& p = pg[i];
      const float* pred = p.map;
      if (det_map_override && det_map_override[i]) {
        if (mem == RT_MEM_HOST || mem == RT_MEM_STAGED_MAPS_HOST) {
          float* d = arena.alloc<float>((size_t)p.det_h * p.det_w);
          RT_HIP_CHECK(hipMemcpyAsync(d, det_map_override[i], (size_t)p.det_h * p.det_w * 4, hipMemcpyHostToDevice, st));
          pred = d;
        } else pred = det_map_override[i];
      }
      p.d_boxes = d_boxes_all + (size_t)i * mb;
      p.d_count = d_counts_all + 2 * i;
      in[i] = pp::DbPageIn{pred, p.det_h, p.det_w, p.after_h, p.after_w};
      wsp[i] = dbws.alloc_bytes(pp::db_workspace_bytes(p.det_h, p.det_w, mb));
      bo[i] = p.d_boxes; co[i] = p.d_count;
    }
    void* hd = pinned.alloc_bytes((size_t)n_pages * pp::db_page_desc_bytes());
    void* dd = arena.alloc_bytes((size_t)n_pages * pp::db_page_desc_bytes());
    ProfScope ps(&prof, st, "db_postprocess");
    pp::db_postprocess_batch(st, n_pages, in.data(), db_params(cfg), wsp.data(), mb, bo.data(), co.data(), hd, dd);
    pp::pack_boxes(st, n_pages, d_boxes_all, d_counts_all, mb, d_boxes_packed);
  }
  tick.lap("dbpost enqueue");
  // metadata round trip #1: box lists (a few KB per page); pixels and tensors stay on the device
  int* h_counts = pinned.alloc<int>((size_t)2 * std::max(n_pages, 1));
  if (n_pages > 0) RT_HIP_CHECK(hipMemcpyAsync(h_counts, d_counts_all, (size_t)2 * n_pages * sizeof(int), hipMemcpyDeviceToHost, st));
  // (the map checksum partials ride to pinned memory with the box counts)
  std::vector<double*> h_sum(sum_parts.size());
  for (size_t g = 0; g < sum_parts.size(); g++) {
    h_sum[g] = pinned.alloc<double>((size_t)sum_counts[g]);
    RT_HIP_CHECK(hipMemcpyAsync(h_sum[g], sum_parts[g], (size_t)sum_counts[g] * 8, hipMemcpyDeviceToHost, st));
  }
  sync(); check_flags();
  int total_lines = 0;
  for (int i = 0; i < n_pages; i++) {
    if (h_counts[2 * i + 1]) throw RtError(RT_ERR_CAPACITY, "DB post-processing work list overflow (raise max_boxes_per_page)");
    pg[i].n_boxes = std::min(std::max(h_counts[2 * i], 0), mb);
    pg[i].first_line = total_lines;
    total_lines += pg[i].n_boxes;
  }
  pp::DbBox* h_boxes = nullptr;
  if (total_lines > 0) {
    h_boxes = pinned.alloc<pp::DbBox>((size_t)total_lines);
    RT_HIP_CHECK(hipMemcpyAsync(h_boxes, d_boxes_packed, (size_t)total_lines * sizeof(pp::DbBox), hipMemcpyDeviceToHost, st));
  }
  if (total_lines > 0) sync();
  if (total_lines > 0)
    for (int i = 0; i < n_pages; i++) pg[i].boxes.assign(h_boxes + pg[i].first_line, h_boxes + pg[i].first_line + pg[i].n_boxes);
  for (size_t g = 0; g < sum_parts.size(); g++)
    for (int i = 0; i < sum_counts[g]; i++) res->det_checksum += h_sum[g][i];

  if (stage_cb) {  // run_stream: the Det stage is complete here (session.rs:98)
    for (int i = 0; i < n_pages; i++) {
      rt_results::Page P;
      P.boxes.resize((size_t)pg[i].n_boxes * 8); P.det_scores.resize((size_t)pg[i].n_boxes);
      for (int k = 0; k < pg[i].n_boxes; k++) {
        float b[8]; memcpy(b, pg[i].boxes[k].pts, 32);
        gm::scale_and_clip(b, (double)pg[i].after_w, (double)pg[i].after_h, (double)pg[i].ori_w, (double)pg[i].ori_h);
        memcpy(&P.boxes[8 * k], b, 32);
        P.det_scores[k] = pg[i].boxes[k].score;
      }
      emit_stage(i, 0, P);
    }
  }
  tick.lap("sync #1 + box D2H");
  // ---- a6: crops --------------------------------------------------------------------
  CropPlan plan;
  for (int i = 0; i < n_pages; i++) {
    std::vector<float> b((size_t)pg[i].n_boxes * 8);
    for (int k = 0; k < pg[i].n_boxes; k++) memcpy(&b[8 * k], pg[i].boxes[k].pts, 32);
    plan_crops(plan, pg[i].img, pg[i].after_h, pg[i].after_w, b.data(), pg[i].n_boxes);
  }
  const int NL = total_lines;
  // per-line results come back in two copies into pinned memory: {label, cls score, token count, rec score} and the tokens
  const int NLp = std::max(NL, 1);
  int* h_meta = pinned.alloc<int>((size_t)4 * NLp);
  memset(h_meta, 0, (size_t)4 * NLp * sizeof(int));
  const int* h_label = h_meta; const float* h_cscore = reinterpret_cast<const float*>(h_meta + NLp);
  const int* h_ntok = h_meta + 2 * NLp; const float* h_rscore = reinterpret_cast<const float*>(h_meta + 3 * NLp);
  const int* h_tokens = nullptr; std::vector<long long> tok_off((size_t)NL + 1, 0);
  if (NL > 0) {
    uint8_t* pool = arena.alloc<uint8_t>(plan.pool_bytes + 64);
    pp::CropDesc* d_desc = arena.alloc<pp::CropDesc>(NL);
    pp::CropRef* d_refs = arena.alloc<pp::CropRef>(NL);
    {
      pp::CropDesc* hd = pinned.alloc<pp::CropDesc>(NL);
      pp::CropRef* hr = pinned.alloc<pp::CropRef>(NL);
      memcpy(hd, plan.descs.data(), (size_t)NL * sizeof(pp::CropDesc));
      memcpy(hr, plan.refs.data(), (size_t)NL * sizeof(pp::CropRef));
      RT_HIP_CHECK(hipMemcpyAsync(d_desc, hd, (size_t)NL * sizeof(pp::CropDesc), hipMemcpyHostToDevice, st));
      RT_HIP_CHECK(hipMemcpyAsync(d_refs, hr, (size_t)NL * sizeof(pp::CropRef), hipMemcpyHostToDevice, st));
    }
    { ProfScope ps(&prof, st, "warp_crops");
      pp::warp_crops(st, d_desc, NL, plan.max_pix, pool); }

    tick.lap("crop plan + warp enqueue");
    // ---- a8 + a9: angle classifier over every crop -------------------------------
    // (cls_processor.rs:127-172: batches of 6 sorted by aspect; the classifier is
    //  per-crop independent, so batch composition does not change any value)
    const int ch = cfg.cls_image_shape[1], cw = cfg.cls_image_shape[2];
    int* d_meta = arena.alloc<int>((size_t)4 * NLp);
    int* d_label = d_meta; float* d_cscore = reinterpret_cast<float*>(d_meta + NLp);
    int* d_ntok = d_meta + 2 * NLp; float* d_rscore = reinterpret_cast<float*>(d_meta + 3 * NLp);
    {
      const int CG = 2048;
      for (int c0 = 0; c0 < NL; c0 += CG) {
        int cn = std::min(CG, NL - c0);
        scratch.rewind();
        pp::LineDesc* hl = pinned.alloc<pp::LineDesc>(cn);
        int* hrow = pinned.alloc<int>(cn);
        for (int k = 0; k < cn; k++) {
          const pp::CropRef& r = plan.refs[c0 + k];
          pp::LineDesc L;
          L.crop_off = r.off; L.h = r.h; L.w = r.w; L.W = cw;
          L.resized_w = gm::resize_norm_resized_w(ch, cw, r.h, r.w);
          L.out_off = (long long)k * ch * cw * 4;
          hl[k] = L; hrow[k] = c0 + k;
        }
        pp::LineDesc* dl = scratch.alloc<pp::LineDesc>(cn);
        int* drow = scratch.alloc<int>(cn);
        RT_HIP_CHECK(hipMemcpyAsync(dl, hl, (size_t)cn * sizeof(pp::LineDesc), hipMemcpyHostToDevice, st));
        RT_HIP_CHECK(hipMemcpyAsync(drow, hrow, (size_t)cn * 4, hipMemcpyHostToDevice, st));
        float* x = scratch.alloc<float>((size_t)cn * ch * cw * 4);
        { ProfScope ps(&prof, st, "resize_norm");
          pp::resize_norm(st, dl, cn, ch, cw, pool, 0, x, d_flags); }
        Level L0 = make_level(uniform_hw(cn, ch, cw));
        RunCtx c = ctx(&scratch);
        float* probs;
        { ProfOuter po(&prof, st, "net/cls"); probs = cls->run(c, x, L0); }
        ProfScope ps(&prof, st, "cls_post_rotate");
        pp::cls_post_rotate(st, probs, drow, cn, cfg.cls_thresh, d_refs, pool, plan.max_pix, d_label, d_cscore);
      }
    }

    tick.lap("cls enqueue");
    // ---- a10 + a11 + a12: recognition ----------------------------------------------
    // per page: order by h/w descending (stable), chunks of batch_num, running max_wh_ratio
    const int rh = cfg.rec_image_shape[1], rw = cfg.rec_image_shape[2];
    std::vector<pp::LineDesc> lines((size_t)NL);
    std::vector<int> line_W((size_t)NL);
    for (int i = 0; i < n_pages; i++) {
      int nb = pg[i].n_boxes, f = pg[i].first_line;
      std::vector<int> order((size_t)nb);
      std::iota(order.begin(), order.end(), 0);
      std::stable_sort(order.begin(), order.end(), [&](int a, int b) {
        double ra = (double)plan.refs[f + a].h / (double)plan.refs[f + a].w;
        double rb = (double)plan.refs[f + b].h / (double)plan.refs[f + b].w;
        return ra > rb;  // Reverse(OrderedFloat(ori_ratio))
      });
      float max_wh_ratio = (float)rw / (float)rh;
      for (int s0 = 0; s0 < nb; s0 += cfg.rec_batch_num) {
        int s1 = std::min(nb, s0 + cfg.rec_batch_num);
        for (int k = s0; k < s1; k++) {
          const pp::CropRef& r = plan.refs[f + order[k]];
          float wh = (float)r.w / (float)r.h;
          if (wh > max_wh_ratio) max_wh_ratio = wh;
        }
        int W = gm::resize_norm_width(rh, rw, max_wh_ratio);
        for (int k = s0; k < s1; k++) {
          int li = f + order[k];
          const pp::CropRef& r = plan.refs[li];
          pp::LineDesc L;
          L.crop_off = r.off; L.h = r.h; L.w = r.w; L.W = W;
          L.resized_w = gm::resize_norm_resized_w(rh, W, r.h, r.w);
          L.out_off = 0;
          lines[li] = L; line_W[li] = W;
        }
      }
    }
    for (int l = 0; l < NL; l++) tok_off[l + 1] = tok_off[l] + RecNet::tokens_for_width(line_W[l]);
    const long long total_tok = tok_off[NL];
    int* d_idx = arena.alloc<int>(std::max<long long>(total_tok, 1));
    float* d_prob = arena.alloc<float>(std::max<long long>(total_tok, 1));
    int* d_tok = arena.alloc<int>(std::max<long long>(total_tok, 1));
    static const long long REC_GROUP_PX = getenv("RT_REC_GROUP_PX") ? atoll(getenv("RT_REC_GROUP_PX")) : (long long)24000000;  // measured sweet spot (profiles/README.md)
    for (int l0 = 0; l0 < NL;) {
      int l1 = l0; long long px = 0;
      while (l1 < NL) { long long add = (long long)rh * line_W[l1]; if (l1 > l0 && px + add > REC_GROUP_PX) break; px += add; l1++; }
      int ln = l1 - l0;
      scratch.rewind();
      std::vector<std::pair<int, int>> hw;
      pp::LineDesc* hl = pinned.alloc<pp::LineDesc>(ln);
      long long off = 0;
      for (int k = 0; k < ln; k++) {
        hl[k] = lines[l0 + k];
        hl[k].out_off = off * 4;
        off += (long long)rh * line_W[l0 + k];
        hw.push_back({rh, line_W[l0 + k]});
      }
      pp::LineDesc* dl = scratch.alloc<pp::LineDesc>(ln);
      RT_HIP_CHECK(hipMemcpyAsync(dl, hl, (size_t)ln * sizeof(pp::LineDesc), hipMemcpyHostToDevice, st));
      float* x = scratch.alloc<float>((size_t)off * 4);
      int maxW = 0; for (auto& p : hw) maxW = std::max(maxW, p.second);
      { ProfScope ps(&prof, st, "resize_norm");
        pp::resize_norm(st, dl, ln, rh, maxW, pool, 0, x, d_flags); }
      Level L0 = make_level(hw), Lt;
      RunCtx c = ctx(&scratch);
      // (the token count per line only depends on the widths, so the offsets are known before the net runs)
      { ProfOuter po(&prof, st, "net/rec");
        rec->run(c, x, L0, Lt, d_idx + tok_off[l0], d_prob + tok_off[l0]); }  // fused CTC head: logits never reach HBM
      if (Lt.total != tok_off[l1] - tok_off[l0]) throw RtError(RT_ERR_SHAPE, "token count mismatch");
      { ProfScope ps(&prof, st, "ctc_decode");
        pp::ctc_decode(st, d_idx + tok_off[l0], d_prob + tok_off[l0], Lt.d, ln, d_tok + tok_off[l0], d_ntok + l0, d_rscore + l0); }
      l0 = l1;
    }
    tick.lap("rec enqueue");
    // metadata round trip #2: labels, scores, token ids
    int* h_tok = pinned.alloc<int>((size_t)std::max<long long>(total_tok, 1));
    h_tokens = h_tok;
    RT_HIP_CHECK(hipMemcpyAsync(h_meta, d_meta, (size_t)4 * NLp * sizeof(int), hipMemcpyDeviceToHost, st));
    if (total_tok > 0) RT_HIP_CHECK(hipMemcpyAsync(h_tok, d_tok, (size_t)total_tok * 4, hipMemcpyDeviceToHost, st));
    sync(); check_flags();
  }

  tick.lap("sync #2 + D2H");
  // ---- results (session.rs:94-105) ---------------------------------------------------
  static const uint16_t LABELS[2] = {0, 180};
  for (int i = 0; i < n_pages; i++) {
    rt_results::Page& P = res->pages[i];
    const PageState& p = pg[i];
    int nb = p.n_boxes;
    P.boxes.resize((size_t)nb * 8); P.det_scores.resize(nb); P.cls_labels.resize(nb); P.cls_scores.resize(nb);
    P.rec_scores.resize(nb); P.tokens.resize(nb); P.text.resize(nb);
    for (int k = 0; k < nb; k++) {
      float b[8]; memcpy(b, p.boxes[k].pts, 32);
      gm::scale_and_clip(b, (double)p.after_w, (double)p.after_h, (double)p.ori_w, (double)p.ori_h);
      memcpy(&P.boxes[8 * k], b, 32);
      P.det_scores[k] = p.boxes[k].score;
      int li = p.first_line + k;
      P.cls_labels[k] = LABELS[h_label[li] ? 1 : 0]; P.cls_scores[k] = h_cscore[li];
      P.rec_scores[k] = h_rscore[li];
      P.tokens[k].assign(h_tokens + tok_off[li], h_tokens + tok_off[li] + h_ntok[li]);
      std::string& t = P.text[k];
      t.reserve(P.tokens[k].size() * 3);  // CJK dictionary entries are 3 UTF-8 bytes
      for (int id : P.tokens[k]) t += dict[(size_t)id];
    }
  }
  if (stage_cb)
    for (int i = 0; i < n_pages; i++) { emit_stage(i, 1, res->pages[i]); emit_stage(i, 2, res->pages[i]); }
  tick.lap("results");
  if (g_trace) last_exit = std::chrono::steady_clock::now();
  return res.release();
}

// Splits the pages over the lanes (contiguous ranges, order preserved) and runs the lanes on
// concurrent host threads; every lane is a full pipeline on its own stream, so one lane's
// small kernels, launch gaps and host sync points overlap the other lanes' large kernels.
// ---------------------------------------------------------------------------
// Lanes.  rt_run_batch splits the pages over the lanes; round 4: the lanes' host threads are persistent (round 3 created and
// joined them on every call) and batches can be SUBMITTED ahead (rt_submit_batch / rt_wait_batch, the counterpart of
// RettoSession::run_stream's worker thread + channel, session.rs:108-143): a lane that has finished its part of batch i starts
// on batch i + 1 at once, so the lanes drift apart in phase and the call boundary (result assembly on the host, descriptors of
// the next call) no longer idles the GPU.
// ---------------------------------------------------------------------------
void LaneWorker::start(int device) {
  th = std::thread([this, device] {
    (void)hipSetDevice(device);
    for (;;) {
      std::function<void()> f;
      {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return stop || !q.empty(); });
        if (q.empty()) return;   // stop requested and nothing left
        f = std::move(q.front()); q.pop_front();
      }
      f();
    }
  });
}
void LaneWorker::push(std::function<void()> f) {
  { std::lock_guard<std::mutex> lk(mu); q.push_back(std::move(f)); }
  cv.notify_one();
}
void LaneWorker::shutdown() {
  { std::lock_guard<std::mutex> lk(mu); stop = true; }
  cv.notify_all();
  if (th.joinable()) th.join();
}
void rt_session::ensure_workers() {
  while (workers.size() < helpers.size() + 1) {
    std::unique_ptr<LaneWorker> w(new LaneWorker());
    w->start(device);
    workers.push_back(std::move(w));
  }
}

// ---- page staging ----------------------------------------------------------------------------------------------------------
// A lane that uploads its own pages blocks its host thread in hipMemcpyAsync (pageable memory: the call returns when the copy is
// done) with nothing queued on its stream: about 3 ms per 11-page part.  rt_submit_batch therefore copies the host pages of the
// whole batch itself, on a copy stream, into a slot of HBM the session keeps per batch in flight; the lanes' streams wait for the
// slot's event.  With one batch submitted ahead the copy of batch i+1 runs on the DMA engines under the kernels of batch i.
void rt_session::stage_pages(rt_ticket* t) {
  if (t->mem != RT_MEM_HOST && t->mem != RT_MEM_HOST_MAPS_DEVICE) return;
  size_t total = 0;
  std::vector<size_t> off((size_t)t->n_pages, (size_t)-1);
  for (int i = 0; i < t->n_pages; i++) {
    if (t->hs[i] <= 0 || t->ws[i] <= 0 || !t->rgb[i]) continue;   // left to the lane, which reports it as the reference does
    off[(size_t)i] = total;
    total += ((size_t)t->hs[i] * t->ws[i] * 3 + 255) & ~(size_t)255;
  }
  if (!total) return;
  RT_HIP_CHECK(hipSetDevice(device));   // (the caller's thread: nothing else has chosen the device on the submit path)
  int slot = -1;
  for (size_t k = 0; k < stage_slots.size(); k++) if (!stage_slots[k].busy) { slot = (int)k; break; }
  if (slot < 0) { stage_slots.emplace_back(); slot = (int)stage_slots.size() - 1; }
  StageSlot& S = stage_slots[(size_t)slot];
  if (!st_copy) RT_HIP_CHECK(hipStreamCreateWithFlags(&st_copy, hipStreamNonBlocking));
  while ((int)S.ev.size() < t->nl) {
    hipEvent_t e = nullptr;
    RT_HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    S.ev.push_back(e);
  }
  if (S.cap < total) {
    if (S.p) { (void)hipFree(S.p); S.p = nullptr; S.cap = 0; }
    if (hipMalloc((void**)&S.p, total) != hipSuccess) { (void)hipGetLastError(); S.p = nullptr; return; }   // the lanes copy, as before
    S.cap = total;
  }
  S.busy = true;
  t->stage_slot = slot;
  t->stage_off = std::move(off);
  t->ev_up.assign(S.ev.begin(), S.ev.begin() + t->nl);
  t->mem_lane = t->mem == RT_MEM_HOST ? RT_MEM_STAGED_MAPS_HOST : RT_MEM_DEVICE;
}
// the part's pages go up right before its lane job is queued: the first lane starts after its own pages, not after the batch's
void rt_session::stage_part(rt_ticket* t, int l) {
  if (t->stage_slot < 0) return;
  StageSlot& S = stage_slots[(size_t)t->stage_slot];
  for (int i = t->first[l]; i < t->first[l + 1]; i++) {
    const size_t o = t->stage_off[(size_t)i];
    if (o == (size_t)-1) continue;
    RT_HIP_CHECK(hipMemcpyAsync(S.p + o, t->rgb[i], (size_t)t->hs[i] * t->ws[i] * 3, hipMemcpyHostToDevice, st_copy));
    t->rgb[i] = S.p + o;
  }
  RT_HIP_CHECK(hipEventRecord(t->ev_up[(size_t)l], st_copy));
}
void rt_session::release_stage(rt_ticket* t) {
  if (t->stage_slot >= 0) stage_slots[(size_t)t->stage_slot].busy = false;
  t->stage_slot = -1;
}
void rt_session::free_stage() {
  if (st_copy) (void)hipStreamSynchronize(st_copy);
  for (auto& S : stage_slots) { if (S.p) (void)hipFree(S.p); for (hipEvent_t e : S.ev) (void)hipEventDestroy(e); }
  stage_slots.clear();
  if (st_copy) { (void)hipStreamDestroy(st_copy); st_copy = nullptr; }
}

rt_ticket* rt_session::submit_batch(const uint8_t* const* rgb, const int* hs, const int* ws, int n_pages, int mem,
                                    const float* const* det_map_override, rt_stage_callback cb, void* user) {
  ensure_workers();
  std::unique_ptr<rt_ticket> t(new rt_ticket());
  const int nl = std::max(1, std::min<int>(std::min<int>((int)helpers.size() + 1, active_lanes), std::max(n_pages, 1)));
  t->nl = nl; t->n_pages = n_pages; t->mem = mem; t->cb = cb; t->user = user;
  t->rgb.assign(rgb, rgb + n_pages); t->hs.assign(hs, hs + n_pages); t->ws.assign(ws, ws + n_pages);
  if (det_map_override) t->maps.assign(det_map_override, det_map_override + n_pages);
  t->parts.assign((size_t)nl, nullptr); t->errs.resize((size_t)nl); t->first.assign((size_t)nl + 1, 0);
  {
    // contiguous ranges of about equal work: det pixels after the session size limit (a2) plus a constant per page
    // for its lines; equal page counts when the pages are all one size
    std::vector<double> cost((size_t)n_pages);
    double total = 0;
    for (int i = 0; i < n_pages; i++) {
      const double h = std::max(hs[i], 1), w = std::max(ws[i], 1);
      const double r = std::min(1.0, (double)cfg.max_side_len / std::max(h, w));
      total += cost[(size_t)i] = h * w * r * r + 250000.0;
    }
    double acc = 0;
    int l = 1;
    for (int i = 0; i < n_pages && l < nl; i++) {
      acc += cost[(size_t)i];
      // close range l-1 after page i once its share is reached, leaving at least one page for each later lane
      if (acc >= total * l / nl - 1e-6 || n_pages - (i + 1) <= nl - l) t->first[l++] = i + 1;  // one range per page: none empty
    }
    for (; l <= nl; l++) t->first[l] = n_pages;
    for (int k = 1; k <= nl; k++) t->first[k] = std::max(t->first[k], t->first[k - 1]);
  }
  t->remaining = nl;
  t->mem_lane = mem;
  rt_ticket* tp = t.get();
  static const bool no_stage = getenv("RT_NO_STAGE") && atoi(getenv("RT_NO_STAGE"));   // (A/B switch of the measurement in DESIGN.md)
  if (!no_stage) stage_pages(tp);
  // parts go to consecutive lanes starting behind the previous batch's last one: batches that fill fewer lanes than the session
  // has (single pages: one lane each) run side by side instead of queueing on lane 0
  const int total_lanes = (int)helpers.size() + 1;
  // (with rt_set_lanes below the session's lane count the partitions would leave CUs unused: whole-device streams then)
  const bool use_parts = active_lanes >= total_lanes;
  const int base = next_lane;
  next_lane = (next_lane + nl) % total_lanes;
  int queued = 0;
  try {
  for (int l = 0; l < nl; l++) {
    const int li = (base + l) % total_lanes;
    rt_session* s = li == 0 ? this : helpers[(size_t)li - 1].get();
    stage_part(tp, l);
    workers[(size_t)li]->push([tp, s, l, use_parts] {
      const int f0 = tp->first[l], f1 = tp->first[l + 1];
      s->stage_cb = tp->cb; s->stage_user = tp->user; s->stage_mu = &tp->cb_mu; s->page_base = f0;
      // a lane inside a submitted batch works on its own CU partition (every call ends with its stream drained, so the lane's
      // arenas and pinned staging can change streams between calls)
      s->st = (s->st_part && use_parts) ? s->st_part : s->st_full;
      s->on_lane_worker = true;
      try {
        if (!tp->ev_up.empty()) RT_HIP_CHECK(hipStreamWaitEvent(s->st, tp->ev_up[(size_t)l], 0));
        tp->parts[l] = s->run_pages(tp->rgb.data() + f0, tp->hs.data() + f0, tp->ws.data() + f0, f1 - f0, tp->mem_lane,
                                    tp->maps.empty() ? nullptr : tp->maps.data() + f0);
      } catch (...) {
        tp->errs[l] = std::current_exception();
        s->failed = true;
        // work of the failed part may still be queued on the lane's stream: drain it before the lane's next job rewinds the
        // arenas and the pinned staging it reads
        if (s->st) (void)hipStreamSynchronize(s->st);
        (void)hipGetLastError();
      }
      s->stage_cb = nullptr; s->stage_mu = nullptr;   // the callback never outlives the batch
      s->st = s->st_full; s->on_lane_worker = false;
      // notify while holding the mutex: rt_wait_batch owns the ticket and deletes it as soon as it sees remaining == 0, so
      // nothing of *tp may be touched once the decrement is visible outside the lock
      { std::lock_guard<std::mutex> lk(tp->mu); tp->remaining--; tp->cv.notify_all(); }
    });
    queued++;
  }
  } catch (...) {
    // a push that threw (bad_alloc): the parts already queued hold tp -- take the un-queued ones off the count, wait for the
    // queued ones, and leave the session as it was (inflight is only counted once every part is queued)
    {
      std::unique_lock<std::mutex> lk(tp->mu);
      tp->remaining -= nl - queued;
      tp->cv.wait(lk, [&] { return tp->remaining == 0; });
    }
    for (auto* r : tp->parts) delete r;
    release_stage(tp);
    throw;
  }
  inflight.fetch_add(1);
  return t.release();
}

rt_results* rt_session::wait_batch(rt_ticket* tp) {
  std::unique_ptr<rt_ticket> t(tp);
  {
    std::unique_lock<std::mutex> lk(t->mu);
    t->cv.wait(lk, [&] { return t->remaining == 0; });
  }
  inflight.fetch_sub(1);
  release_stage(t.get());   // (every lane has drained its stream: nothing reads the slot any more)
  std::unique_ptr<rt_results> res(new rt_results());
  std::exception_ptr err;
  for (int l = 0; l < t->nl; l++) {
    if (t->errs[l] && !err) err = t->errs[l];
    if (t->parts[l]) {
      for (auto& p : t->parts[l]->pages) res->pages.push_back(std::move(p));
      res->det_checksum += t->parts[l]->det_checksum;
      delete t->parts[l];
    }
  }
  if (err) std::rethrow_exception(err);
  return res.release();
}

rt_results* rt_session::run_batch(const uint8_t* const* rgb, const int* hs, const int* ws, int n_pages, int mem,
                                  const float* const* det_map_override, rt_stage_callback cb, void* user) {
  const int nl = std::max(1, std::min<int>(std::min<int>((int)helpers.size() + 1, active_lanes), std::max(n_pages, 1)));
  if (nl <= 1) {   // one lane: on the caller's thread (no hand-over in the single-page latency path)
    std::mutex cb_mu;
    struct Disarm {  // the callback never outlives the call
      rt_session* self;
      ~Disarm() { self->stage_cb = nullptr; self->stage_mu = nullptr; }
    } disarm{this};
    stage_cb = cb; stage_user = user; stage_mu = &cb_mu; page_base = 0;
    try {
      return run_pages(rgb, hs, ws, n_pages, mem, det_map_override);
    } catch (...) { failed = true; throw; }
  }
  return wait_batch(submit_batch(rgb, hs, ws, n_pages, mem, det_map_override, cb, user));
}

// RettoWorkerStageResult JSON (serde derive shapes; retto-wasm/fe/index.ts:5-42)
static std::string stage_json(const rt_results::Page& P, int stage) {
  std::ostringstream o;
  size_t n = P.det_scores.size();
  if (stage == 0) {
    o << "[";
    for (size_t k = 0; k < n; k++) {
      if (k) o << ",";
      o << "{\"boxes\":{\"inner\":[";
      for (int q = 0; q < 4; q++) { if (q) o << ","; o << "{\"x\":" << fnum(P.boxes[8 * k + 2 * q]) << ",\"y\":" << fnum(P.boxes[8 * k + 2 * q + 1]) << "}"; }
      o << "]},\"score\":" << fnum(P.det_scores[k]) << "}";
    }
    o << "]";
  } else if (stage == 1) {
    o << "[";
    for (size_t k = 0; k < n; k++) { if (k) o << ","; o << "{\"label\":{\"label\":" << P.cls_labels[k] << ",\"score\":" << fnum(P.cls_scores[k]) << "}}"; }
    o << "]";
  } else {
    o << "[";
    for (size_t k = 0; k < n; k++) { if (k) o << ","; o << "{\"text\":\"" << json_escape(P.text[k]) << "\",\"score\":" << fnum(P.rec_scores[k]) << "}"; }
    o << "]";
  }
  return o.str();
}
const char* rt_results_json_impl(rt_results* r, int page, int stage) {
  rt_results::Page& P = r->pages[(size_t)page];
  P.json[stage] = stage_json(P, stage);
  return P.json[stage].c_str();
}
void rt_session::emit_stage(int page, int stage, const rt_results::Page& P) {
  if (!stage_cb) return;
  const std::string j = stage_json(P, stage);
  std::unique_lock<std::mutex> lk;
  if (stage_mu) lk = std::unique_lock<std::mutex>(*stage_mu);
  stage_cb(stage_user, page_base + page, stage, j.c_str());
}
