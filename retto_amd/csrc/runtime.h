// Small device runtime pieces shared by the nets and the session: a growing bump
// arena for activations, a pinned staging buffer for descriptor tables, an RTWB blob
// reader, and a per-kernel-family HIP-event profiler.
#pragma once
#include <cstdlib>
#include <map>
#include <string>
#include <vector>

#include "common.h"

namespace rt {

// ---- CU partitions (round 5; an experiment that LOST -- opt-in with RT_LANE_CUMASK=1, kept for A/B) -----------------------
// Measured on MI355X (tools/scratch/cumask.hip): two kernels on different streams whose CU masks OVERLAP do not run side by
// side -- a copy kernel launched beside a 128-workgroup MFMA kernel waits for it even with half of the CUs idle --, while
// streams with DISJOINT masks (hipExtStreamCreateWithCUMask) run truly concurrently, each at its partition's rate (128 | 128
// CUs: an MFMA loop at its full per-CU rate beside copies at 5.2 TB/s; a copy reaches ~43 GB/s per CU, 2.9 TB/s on 64 CUs).
// Mask bit i = CU slot i / 8 of XCD i % 8; slot k lies on shader engine k % 4; an XCD whose part of the mask is empty falls
// back to 8 CUs.  Giving every lane of a session its own slice of every XCD (lane l of L: slots [32 l / L, 32 (l + 1) / L))
// was measured on C3: 3 lanes 37.4 ms per step, 4 lanes 31.5, against 27.4 with whole-device streams -- the lanes run the same
// chain and meet in the same HBM-bound phases, where a third of the CUs cannot pull a third of the bandwidth, and every
// kernel's tail idles its partition.  The lanes therefore stay on whole-device streams (they time-slice the chip and fill
// each other's launch gaps and host round trips, nothing more).
// Returns a new stream restricted to partition `part` of `parts` on the current device, or nullptr when the mask cannot be
// verified on this device (the caller then keeps a plain stream).  cus_out = CUs of the partition.
hipStream_t partition_stream(int part, int parts, int* cus_out, std::vector<unsigned>* cu_ids = nullptr);
// CUs a kernel launched on `st` can occupy (persistent kernels size their grids with it); the device's CU count for any
// stream that is not a partition stream.
int stream_cus(hipStream_t st);
void forget_stream(hipStream_t st);

// ---- RTWB weight blob (format: retto_amd/synth.py) -------------------------------
struct BlobTensor {
  std::vector<int> dims;
  const float* data;  // points into the owning Blob's bytes
  size_t numel() const { size_t n = 1; for (int d : dims) n *= (size_t)d; return n; }
};
class Blob {
 public:
  // worker.rs:30-56: Path must exist / Blob must be non-empty, else ModelNotFound
  // model_kind >= 0 (onnx_import.h ModelKind): bytes that are not RTWB are taken as an .onnx file and imported
  static Blob from_source(const char* path, const void* data, size_t len, const char* what, int model_kind = -1);
  const BlobTensor& get(const std::string& name) const;
  bool has(const std::string& name) const { return t_.count(name) != 0; }
  const std::vector<uint8_t>& bytes() const { return bytes_; }
 private:
  void parse();
  std::vector<uint8_t> bytes_;
  std::map<std::string, BlobTensor> t_;
};
std::vector<uint8_t> read_source_bytes(const char* path, const void* data, size_t len, const char* what);

// ---- device bump arena --------------------------------------------------------------
// alloc() never frees; reset() rewinds.  When a run needs more than the current
// capacity the arena records the shortfall, and the owner re-runs after grow().
class Arena {
 public:
  explicit Arena(size_t initial = 0) { if (initial) reserve(initial); }
  ~Arena();
  Arena(const Arena&) = delete;
  Arena& operator=(const Arena&) = delete;
  void reserve(size_t bytes);
  // Rewinds.  If the last pass had to chain blocks, the caller has synchronised the
  // stream (end of an API call), so the chain is replaced by one block that fits the
  // largest pass seen; steady state never allocates.
  void reset();
  void rewind() { if (pass_ > need_) need_ = pass_; pass_ = 0; off_ = 0; }  // stream-ordered reuse inside a call
  // A call that FAILED must not size the arena for the calls after it (a page whose size limits explode -- 1 x 4000 becomes
  // 736 x 2.76 M det pixels, as in the reference -- asked for tens of GB before hipMalloc refused): forget the pass.
  // need_ goes back to its value at the start of the failed call too (mark_call): rewind()s inside that call have already folded
  // the exploding pass into it.
  void abandon_pass() { pass_ = 0; need_ = need_mark_; }
  void mark_call() { need_mark_ = need_; }
  size_t used() const { return off_; }
  size_t peak() const { return need_; }
  size_t capacity() const { return cap_; }
  template <typename T>
  T* alloc(size_t count) { return reinterpret_cast<T*>(alloc_bytes(count * sizeof(T))); }
  void* alloc_bytes(size_t bytes);
 private:
  char* base_ = nullptr;
  size_t cap_ = 0, off_ = 0, pass_ = 0, need_ = 0, need_mark_ = 0;
  std::vector<void*> old_;  // superseded blocks, kept until destruction (in-flight kernels may still read them)
};

// ---- pinned host staging (descriptor tables, results) -------------------------------
class Pinned {
 public:
  ~Pinned();
  void reset() { cur_ = 0; off_ = 0; }  // back to the first block: staging of the previous call has been consumed
  void* alloc_bytes(size_t bytes);
  template <typename T>
  T* alloc(size_t count) { return reinterpret_cast<T*>(alloc_bytes(count * sizeof(T))); }
 private:
  struct Block { char* p; size_t cap; };
  std::vector<Block> blocks_;
  size_t cur_ = 0, off_ = 0;
};

// ---- profiler -----------------------------------------------------------------------
class Profiler {
 public:
  ~Profiler();
  int on = 0;   // 0 off, 1 every launch family + network scopes, 2 network scopes only
  bool detail = getenv("RT_PROFILE_DETAIL") != nullptr;  // per-layer labels "family@shape" (tools/layer_profile.py)
  void begin(hipStream_t st, const char* name);
  void end(hipStream_t st);
  // a second, enclosing level ("net/det", "net/cls", "net/rec"): whole-network device time next to the families
  hipEvent_t outer_begin(hipStream_t st);
  void outer_end(hipStream_t st, const char* name, hipEvent_t a);
  void collect();  // after a stream sync: fold event pairs into the totals
  void clear();
  void merge(Profiler& other);  // adds other's totals into this one and zeroes other's
  std::vector<const char*> names;
  std::vector<float> ms;
  std::vector<int> calls;
 private:
  struct Rec { int id; hipEvent_t a, b; };
  std::vector<Rec> recs_;
  std::vector<hipEvent_t> pool_;
  std::vector<std::string> name_store_;
  int id_of(const char* name);
  hipEvent_t get_event();
  int cur_ = -1;
  hipEvent_t cur_a_{};
};

struct ProfScope {
  Profiler* p; hipStream_t st;
  // (p->on: 1 = every launch family + the enclosing network scopes; 2 = the network scopes only -- ~65 event pairs per det pass
  //  are themselves work on the stream, so whole-network times are read from a pass without them)
  ProfScope(Profiler* p_, hipStream_t s, const char* name) : p(p_), st(s) { if (p && p->on == 1) p->begin(st, name); }
  ProfScope(Profiler* p_, hipStream_t s, const char* name, const std::string& shape) : p(p_), st(s) {
    if (p && p->on == 1) p->begin(st, p->detail ? (std::string(name) + "@" + shape).c_str() : name);
  }
  ~ProfScope() { if (p && p->on == 1) p->end(st); }
};
struct ProfOuter {
  Profiler* p; hipStream_t st; const char* name; hipEvent_t a{};
  ProfOuter(Profiler* p_, hipStream_t s, const char* n) : p(p_), st(s), name(n) { if (p && p->on) a = p->outer_begin(st); }
  ~ProfOuter() { if (p && p->on) p->outer_end(st, name, a); }
};

}  // namespace rt
