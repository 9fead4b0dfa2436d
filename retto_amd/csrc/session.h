// Host scheduler: the MI355X counterpart of RettoSession::process_pipeline
// (/root/reference/retto-core/src/session.rs:75-106) over a batch of pages, plus the
// tensor-level worker entry points (worker.rs:69-73) and the stage functions.
#pragma once
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/retto_hip.h"
#include "nets.h"
#include "nets_f16.h"
#include "prepost.h"
#include "runtime.h"

struct rt_results {
  struct Page {
    std::vector<float> boxes;        // n x 8, original-image coordinates
    std::vector<float> det_scores;
    std::vector<uint16_t> cls_labels;
    std::vector<float> cls_scores;
    std::vector<float> rec_scores;
    std::vector<std::vector<int32_t>> tokens;
    std::vector<std::string> text;
    std::string json[3];
  };
  std::vector<Page> pages;
  double det_checksum = 0.0;
};

// A lane's host thread: created once per session (rt_session::ensure_workers), parked on a condition variable between jobs.
// Jobs run in submission order; a lane works through the parts of consecutive batches back to back, so the result assembly of
// batch i (host) and the det phase of batch i + 1 overlap with the other lanes' kernels.
struct LaneWorker {
  std::thread th;
  std::mutex mu;
  std::condition_variable cv;
  std::deque<std::function<void()>> q;
  bool stop = false;
  void start(int device);
  void push(std::function<void()> f);
  void shutdown();
  ~LaneWorker() { shutdown(); }
};

// One submitted batch (rt_submit_batch): the argument arrays are copied (the PAGES must stay valid until rt_wait_batch), the
// pages are split over the lanes exactly as rt_run_batch splits them, each lane leaves its part here.
struct rt_ticket {
  int nl = 0, n_pages = 0, mem = 0;
  std::vector<const uint8_t*> rgb;
  std::vector<int> hs, ws;
  std::vector<const float*> maps;   // empty: no override
  std::vector<int> first;
  std::vector<rt_results*> parts;
  std::vector<std::exception_ptr> errs;
  rt_stage_callback cb = nullptr;
  void* user = nullptr;
  std::mutex cb_mu;                 // callbacks of concurrent lanes are serialised
  std::mutex mu;
  std::condition_variable cv;
  int remaining = 0;
  // host pages staged by rt_submit_batch itself (session.cpp "page staging"): rgb[] then holds device addresses inside the
  // session's staging slot `stage_slot`, and the lanes wait for `ev_up` on their streams instead of copying
  int stage_slot = -1;
  int mem_lane = 0;                 // what the lanes are told: mem, or pages-on-device once staged
  std::vector<size_t> stage_off;    // per page: offset inside the slot, (size_t)-1 = not staged (empty page)
  std::vector<hipEvent_t> ev_up;    // per lane part; owned by the staging slot
};

// internal to the library (never accepted from a caller): pages already in HBM, det map overrides still host pointers
#define RT_MEM_STAGED_MAPS_HOST 3

struct rt_session {
  rt_config cfg{};
  int device = 0;
  hipStream_t st = nullptr;        // the stream the lane's current call runs on: st_full, or st_part inside a multi-lane batch
  hipStream_t st_full = nullptr;   // whole device
  hipStream_t st_part = nullptr;   // this lane's CU partition (runtime.h "CU partitions"); nullptr: not partitioned
  int part_cus = 0;
  hipEvent_t ev_block = nullptr;   // blocking-sync event behind sync(): the lane's host thread sleeps instead of spinning
  rt::Arena arena;      // lives for one API call: pages, maps, crops, descriptors, outputs
  rt::Arena scratch;    // network activations; rewound per launch group
  rt::Arena dbws;       // DB post-processing workspace (stream-ordered reuse across pages)
  rt::Pinned pinned;
  rt::Profiler prof;
  std::shared_ptr<rt::DetModel> det;   // weights are shared with the helper lanes
  std::shared_ptr<rt::ClsModel> cls;
  std::shared_ptr<rt::RecModel> rec;
  std::string model_info;
  // extra lanes: same networks and config, own stream / arenas; rt_run_batch splits the pages
  // over the lanes and runs them on concurrent host threads
  std::vector<std::unique_ptr<rt_session>> helpers;
  int active_lanes = 1 << 30;  // rt_set_lanes: upper bound on the lanes rt_run_batch uses
  std::vector<std::string> dict;  // RecCharacter (rec_processor.rs:29-46)
  std::string last_error;
  int* d_flags = nullptr;         // [0] thumbnail/resize error flag
  // run_stream (session.rs:133-143): stage results are handed to the callback as soon as the stage is complete --
  // Det after the box round trip (before any crop is classified or read), Cls and Rec when the call ends.
  rt_stage_callback stage_cb = nullptr;
  void* stage_user = nullptr;
  std::mutex* stage_mu = nullptr;  // callbacks of concurrent lanes are serialised
  int page_base = 0;               // global index of this lane's first page
  std::chrono::steady_clock::time_point last_exit = std::chrono::steady_clock::now();  // RT_TRACE only
  void emit_stage(int page, int stage, const rt_results::Page& P);

  rt::RunCtx ctx(rt::Arena* a) { return rt::RunCtx{st, a, &pinned, &prof}; }
  bool on_lane_worker = false;    // set by a lane's worker thread around run_pages (submitted batches): bounded polling there
  int spin_us = 5000;              // sync(): how long to poll before sleeping; run_pages drops it to 50 for multi-page batches
  void begin_call();
  void sync();
  void check_flags();

  // L1
  void det_forward(const float* nchw, int n, int h, int w, float* out);
  void cls_forward(const float* nchw, int n, int h, int w, float* out);
  void rec_forward(const float* nchw, int n, int h, int w, float* out, int* t_out);
  void rec_forward_ragged(const float* nchw, int n, const int* widths, float* out, int* t_out);
  // stages
  void resize_both(const uint8_t* rgb, int h, int w, uint8_t* out, int oh, int ow);
  void det_preprocess(const uint8_t* rgb, int h, int w, float* out);
  void det_postprocess(const float* pred, int h, int w, int ori_h, int ori_w, float* boxes, float* scores, int max_out,
                       int* n_out);
  void crop_images(const uint8_t* rgb, int h, int w, const float* boxes, int n, uint8_t* out, size_t out_cap);
  void resize_norm_image(const uint8_t* crop, int h, int w, int ori_h, int ori_w, int img_h, int img_w, float ratio,
                         float* out);
  void ctc_decode(const float* probs, int n, int t, int c, int32_t* idx, float* prob, int32_t* tokens,
                  int32_t* n_tokens, float* scores);
  // L2
  // persistent lane threads (index 0 = this session's own lane) and the number of submitted, not yet waited batches
  std::vector<std::unique_ptr<LaneWorker>> workers;
  std::atomic<int> inflight{0};
  int next_lane = 0;               // first lane of the next submitted batch
  bool failed = false;             // the lane's previous call threw: its arena statistics are discarded at the next begin_call
  // page staging: host pages of a submitted batch are copied to HBM by the submitting thread on a copy stream of their own, one
  // batch ahead of the lanes that read them (slots are reused; one per batch in flight)
  struct StageSlot { uint8_t* p = nullptr; size_t cap = 0; std::vector<hipEvent_t> ev; bool busy = false; };
  std::vector<StageSlot> stage_slots;
  hipStream_t st_copy = nullptr;
  void stage_pages(rt_ticket* t);          // picks the slot
  void stage_part(rt_ticket* t, int l);    // copies the pages of lane part l
  void release_stage(rt_ticket* t);
  void free_stage();
  void ensure_workers();
  rt_ticket* submit_batch(const uint8_t* const* rgb, const int* hs, const int* ws, int n_pages, int mem,
                          const float* const* det_map_override, rt_stage_callback cb = nullptr, void* user = nullptr);
  rt_results* wait_batch(rt_ticket* t);   // consumes the ticket
  rt_results* run_batch(const uint8_t* const* rgb, const int* hs, const int* ws, int n_pages, int mem,
                        const float* const* det_map_override, rt_stage_callback cb = nullptr, void* user = nullptr);
  rt_results* run_pages(const uint8_t* const* rgb, const int* hs, const int* ws, int n_pages, int mem,
                        const float* const* det_map_override);  // one lane
};

rt_session* rt_session_create(const rt_config* cfg);
std::string rt_format_f32_impl(float v);  // serde_json / ryu form of an f32
namespace rt {
// RecCharacter::new (rec_processor.rs:29-46) with Rust's from_utf8 / lines / trim semantics
std::vector<std::string> load_dictionary(const std::vector<uint8_t>& bytes);
}
