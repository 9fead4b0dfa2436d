#include "runtime.h"
#include "onnx_import.h"

#include <sys/stat.h>

#include <cstdio>
#include <cstring>
#include <atomic>
#include <map>
#include <mutex>
#include <set>
#include <vector>
#include <utility>

namespace rt {

void allow_big_lds(const void* kernel, int bytes) {
  static std::mutex mu;
  static std::set<std::pair<int, const void*>> done;
  int dev = 0;
  RT_HIP_CHECK(hipGetDevice(&dev));
  // fast path without the lock: what THIS thread has already seen set (a lane thread launches these kernels thousands of
  // times per second; the three lanes of a session used to meet on the mutex eleven times per fp16 conv launch)
  thread_local std::set<std::pair<int, const void*>> seen;
  if (seen.count({dev, kernel})) return;
  {
    std::lock_guard<std::mutex> lk(mu);
    if (!done.count({dev, kernel})) {
      RT_HIP_CHECK(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
      done.insert({dev, kernel});
    }
  }
  seen.insert({dev, kernel});   // only once the attribute is known to be set: a throw above must be retried by the next call
}

// ---- CU partitions ---------------------------------------------------------------------
namespace {
__global__ void k_where_am_i(unsigned* out) {
  unsigned hw, xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  if (threadIdx.x == 0) out[blockIdx.x] = ((xcc & 15u) << 8) | ((hw >> 8) & 0xffu);   // XCD | shader engine, array, CU
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  while (__builtin_amdgcn_s_memtime() - t0 < 4000) {}   // (stay a moment so that the blocks spread over every CU the mask allows)
}
std::mutex g_part_mu;
std::map<std::pair<int, hipStream_t>, int> g_stream_cus;
std::map<int, int> g_dev_cus;
}  // namespace
static int stream_cus_locked(int dev, hipStream_t st);

// (on the launch path of every narrow gemm(), gemm_w, gemm_dma: the last answer is kept per thread -- a lane thread launches on
//  one stream of one device -- so the process-wide mutex is met once per (thread, stream), not per launch; forget_stream bumps
//  the epoch that invalidates every thread's copy)
static std::atomic<unsigned> g_part_epoch{1};
int stream_cus(hipStream_t st) {
  int dev = 0;
  RT_HIP_CHECK(hipGetDevice(&dev));
  struct Seen { unsigned epoch = 0; int dev = -1; hipStream_t st = nullptr; int cus = 0; };
  static thread_local Seen seen;
  const unsigned ep = g_part_epoch.load(std::memory_order_acquire);
  if (seen.epoch == ep && seen.dev == dev && seen.st == st) return seen.cus;
  const int c = stream_cus_locked(dev, st);
  seen = Seen{ep, dev, st, c};
  return c;
}
int stream_cus_locked(int dev, hipStream_t st) {
  std::lock_guard<std::mutex> lk(g_part_mu);
  auto it = g_stream_cus.find({dev, st});
  if (it != g_stream_cus.end()) return it->second;
  int& c = g_dev_cus[dev];
  if (!c) {
    hipDeviceProp_t p;
    RT_HIP_CHECK(hipGetDeviceProperties(&p, dev));
    c = p.multiProcessorCount;
  }
  return c;
}
void forget_stream(hipStream_t st) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return;
  std::lock_guard<std::mutex> lk(g_part_mu);
  g_stream_cus.erase({dev, st});
  g_part_epoch.fetch_add(1, std::memory_order_release);
}
hipStream_t partition_stream(int part, int parts, int* cus_out, std::vector<unsigned>* cu_ids) {
  int dev = 0;
  RT_HIP_CHECK(hipGetDevice(&dev));
  hipDeviceProp_t p;
  RT_HIP_CHECK(hipGetDeviceProperties(&p, dev));
  const int ncu = p.multiProcessorCount;
  if (ncu != 256 || parts < 2 || parts > 8 || part < 0 || part >= parts) return nullptr;   // (the bit layout below is the 8-XCD x 32-CU part's)
  // whole groups of 4 slots (slot k lies on shader engine k % 4; workgroups are dealt evenly over the shader engines whatever
  // their CU counts, so a partition with 3 / 3 / 2 / 2 CUs per engine runs like one with 2 / 2 / 2 / 2)
  const int k0 = 4 * (8 * part / parts), k1 = 4 * (8 * (part + 1) / parts);
  std::vector<uint32_t> mask(8, 0u);
  for (int k = k0; k < k1; k++)
    for (int x = 0; x < 8; x++) { const int bit = k * 8 + x; mask[(size_t)bit / 32] |= 1u << (bit % 32); }
  hipStream_t st = nullptr;
  if (hipExtStreamCreateWithCUMask(&st, (uint32_t)mask.size(), mask.data()) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
  // verify: 2048 one-wave workgroups must land on exactly (k1 - k0) CUs of every XCD
  bool ok = false;
  unsigned* d = nullptr;
  const int nb = 2048;
  if (hipMalloc((void**)&d, nb * 4) == hipSuccess) {
    hipLaunchKernelGGL(k_where_am_i, dim3(nb), dim3(64), 0, st, d);
    std::vector<unsigned> h((size_t)nb);
    if (hipGetLastError() == hipSuccess && hipStreamSynchronize(st) == hipSuccess &&
        hipMemcpy(h.data(), d, nb * 4, hipMemcpyDeviceToHost) == hipSuccess) {
      std::set<unsigned> seen(h.begin(), h.end());
      int per_xcd[16] = {0};
      for (unsigned v : seen) per_xcd[(v >> 8) & 15]++;
      ok = (int)seen.size() == 8 * (k1 - k0);
      for (int x = 0; x < 8; x++) ok = ok && per_xcd[x] == k1 - k0;
      if (cu_ids) cu_ids->assign(seen.begin(), seen.end());
    }
    (void)hipFree(d);
  }
  if (!ok) { (void)hipGetLastError(); (void)hipStreamDestroy(st); return nullptr; }
  {
    std::lock_guard<std::mutex> lk(g_part_mu);
    g_stream_cus[{dev, st}] = 8 * (k1 - k0);
    g_part_epoch.fetch_add(1, std::memory_order_release);   // (a recycled stream handle must not meet a thread's stale answer)
  }
  if (cus_out) *cus_out = 8 * (k1 - k0);
  return st;
}

// ---- sources / blob ---------------------------------------------------------------
std::vector<uint8_t> read_source_bytes(const char* path, const void* data, size_t len, const char* what) {
  std::vector<uint8_t> out;
  if (path != nullptr) {
    struct stat sb;
    if (stat(path, &sb) != 0) throw RtError(7, std::string("Model not found: ") + path);
    FILE* f = fopen(path, "rb");
    if (!f) throw RtError(1, std::string("cannot open ") + path);
    out.resize((size_t)sb.st_size);
    size_t got = out.empty() ? 0 : fread(out.data(), 1, out.size(), f);
    fclose(f);
    if (got != out.size()) throw RtError(1, std::string("short read on ") + path);
    return out;
  }
  if (data == nullptr || len == 0) throw RtError(7, std::string("Model not found: Empty model blob! (") + what + ")");
  out.assign((const uint8_t*)data, (const uint8_t*)data + len);
  return out;
}

Blob Blob::from_source(const char* path, const void* data, size_t len, const char* what, int model_kind) {
  Blob b;
  b.bytes_ = read_source_bytes(path, data, len, what);
  if (model_kind >= 0 && !looks_like_rtwb(b.bytes_)) b.bytes_ = onnx_to_rtwb(model_kind, b.bytes_.data(), b.bytes_.size());
  b.parse();
  return b;
}

void Blob::parse() {
  const uint8_t* p = bytes_.data();
  size_t n = bytes_.size();
  auto need = [&](size_t off, size_t cnt) {
    if (cnt > n || off > n - cnt) throw RtError(4, "RTWB blob truncated");  // (no wrap-around on hostile offsets)
  };
  need(0, 16);
  if (memcmp(p, "RTWB", 4) != 0) throw RtError(4, "not an RTWB weight blob");
  uint32_t ver, cnt;
  memcpy(&ver, p + 4, 4); memcpy(&cnt, p + 8, 4);
  if (ver != 1) throw RtError(4, "unsupported RTWB version");
  size_t off = 16;
  struct Ent { std::string name; std::vector<int> dims; uint64_t o, nb; };
  std::vector<Ent> ents;
  for (uint32_t i = 0; i < cnt; i++) {
    need(off, 2);
    uint16_t ln; memcpy(&ln, p + off, 2); off += 2;
    need(off, ln + 4);
    Ent e; e.name.assign((const char*)p + off, ln); off += ln;
    uint8_t ndim = p[off], dt = p[off + 1]; off += 4;
    if (dt != 0) throw RtError(4, "RTWB: only f32 tensors supported");
    need(off, 4u * ndim + 16);
    for (int d = 0; d < ndim; d++) { uint32_t v; memcpy(&v, p + off, 4); off += 4; e.dims.push_back((int)v); }
    memcpy(&e.o, p + off, 8); memcpy(&e.nb, p + off + 8, 8); off += 16;
    ents.push_back(std::move(e));
  }
  size_t base = (off + 63) / 64 * 64;
  for (auto& e : ents) {
    if (e.o > n) throw RtError(4, "RTWB blob truncated");
    need(base + e.o, e.nb);
    BlobTensor t; t.dims = e.dims; t.data = reinterpret_cast<const float*>(p + base + e.o);
    if (t.numel() * 4 != e.nb) throw RtError(4, "RTWB: size mismatch for " + e.name);
    t_[e.name] = t;
  }
}

const BlobTensor& Blob::get(const std::string& name) const {
  auto it = t_.find(name);
  if (it == t_.end()) throw RtError(4, "RTWB: missing tensor " + name);
  return it->second;
}

// ---- arena ------------------------------------------------------------------------
Arena::~Arena() {
  if (base_) (void)hipFree(base_);
  for (void* p : old_) (void)hipFree(p);
}
void Arena::reserve(size_t bytes) {
  if (bytes <= cap_) return;
  void* p = nullptr;
  RT_HIP_CHECK(hipMalloc(&p, bytes));   // (first: a refused allocation leaves the arena as it was -- the old block must not end up
                                        //  both current and on the superseded list, which reset() frees)
  if (base_) old_.push_back(base_);
  base_ = (char*)p; cap_ = bytes; off_ = 0;
}
void* Arena::alloc_bytes(size_t bytes) {
  bytes = (bytes + 255) & ~(size_t)255;
  if (off_ + bytes > cap_) {
    // allocations made earlier in this pass stay valid in the superseded block
    size_t want = std::max<size_t>(std::max<size_t>(cap_ * 2, bytes * 2), (size_t)64 << 20);
    reserve(want);
  }
  void* p = base_ + off_;
  off_ += bytes;
  pass_ += bytes;
  return p;
}
void Arena::reset() {
  if (pass_ > need_) need_ = pass_;
  if (!old_.empty()) {
    (void)hipDeviceSynchronize();
    for (void* q : old_) (void)hipFree(q);
    old_.clear();
    size_t want = need_ + need_ / 8 + ((size_t)1 << 20);
    if (want > cap_) {
      (void)hipFree(base_);
      base_ = nullptr; cap_ = 0; pass_ = 0; off_ = 0;   // (consistent even if the allocation below throws)
      void* q = nullptr;
      RT_HIP_CHECK(hipMalloc(&q, want));
      base_ = (char*)q; cap_ = want;
    }
  }
  pass_ = 0; off_ = 0;
}

Pinned::~Pinned() { for (auto& b : blocks_) (void)hipHostFree(b.p); }
void* Pinned::alloc_bytes(size_t bytes) {
  bytes = (bytes + 63) & ~(size_t)63;
  while (true) {
    if (cur_ < blocks_.size() && off_ + bytes <= blocks_[cur_].cap) {
      void* p = blocks_[cur_].p + off_;
      off_ += bytes;
      return p;
    }
    if (cur_ + 1 < blocks_.size()) { cur_++; off_ = 0; continue; }
    size_t cap = std::max<size_t>(bytes, (size_t)4 << 20);
    void* p = nullptr;
    RT_HIP_CHECK(hipHostMalloc(&p, cap, hipHostMallocDefault));
    blocks_.push_back(Block{(char*)p, cap});
    cur_ = blocks_.size() - 1; off_ = 0;
  }
}

// ---- profiler ---------------------------------------------------------------------
Profiler::~Profiler() {
  for (auto& r : recs_) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
  for (auto e : pool_) (void)hipEventDestroy(e);
}
int Profiler::id_of(const char* name) {
  for (size_t i = 0; i < name_store_.size(); i++) if (name_store_[i] == name) return (int)i;
  name_store_.reserve(256);
  name_store_.push_back(name);
  ms.push_back(0.f); calls.push_back(0);
  names.clear();
  for (auto& s : name_store_) names.push_back(s.c_str());
  return (int)name_store_.size() - 1;
}
hipEvent_t Profiler::get_event() {
  if (!pool_.empty()) { hipEvent_t e = pool_.back(); pool_.pop_back(); return e; }
  hipEvent_t e; RT_HIP_CHECK(hipEventCreate(&e)); return e;
}
void Profiler::begin(hipStream_t st, const char* name) {
  cur_ = id_of(name);
  cur_a_ = get_event();
  RT_HIP_CHECK(hipEventRecord(cur_a_, st));
}
void Profiler::end(hipStream_t st) {
  hipEvent_t b = get_event();
  RT_HIP_CHECK(hipEventRecord(b, st));
  recs_.push_back(Rec{cur_, cur_a_, b});
}
hipEvent_t Profiler::outer_begin(hipStream_t st) {
  hipEvent_t a = get_event();
  RT_HIP_CHECK(hipEventRecord(a, st));
  return a;
}
void Profiler::outer_end(hipStream_t st, const char* name, hipEvent_t a) {
  hipEvent_t b = get_event();
  RT_HIP_CHECK(hipEventRecord(b, st));
  recs_.push_back(Rec{id_of(name), a, b});
}
void Profiler::collect() {
  for (auto& r : recs_) {
    float t = 0.f;
    if (hipEventElapsedTime(&t, r.a, r.b) == hipSuccess) { ms[r.id] += t; calls[r.id] += 1; }
    pool_.push_back(r.a); pool_.push_back(r.b);
  }
  recs_.clear();
}
void Profiler::merge(Profiler& o) {
  o.collect();
  for (size_t i = 0; i < o.name_store_.size(); i++) {
    int id = id_of(o.name_store_[i].c_str());
    ms[id] += o.ms[i]; calls[id] += o.calls[i];
    o.ms[i] = 0.f; o.calls[i] = 0;
  }
}
void Profiler::clear() {
  collect();
  for (auto& v : ms) v = 0.f;
  for (auto& c : calls) c = 0;
}

}  // namespace rt
