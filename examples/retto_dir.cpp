// Native directory driver over the C ABI (include/retto_hip.h): the loop of retto-cli
// (/root/reference/retto-cli/src/main.rs:41-95) without Python -- walk a directory, read every page,
// run the pages through rt_run_batch in batches, print the three stage results per image in the wire
// format retto-wasm emits (retto-wasm/fe/index.ts:5-42) and the average time per image.
// Files are PNG / JPEG / PNM / BMP: the library's host decoder (rt_decode_image; image_helper.rs:34-44 is host code in
// the reference too) runs on a few threads for batch i+1 while batch i is on the GPU.
//
//   g++ -std=c++17 -Iinclude examples/retto_dir.cpp -Lretto_amd -lretto_hip -Wl,-rpath,$PWD/retto_amd -o examples/retto_dir
//   examples/retto_dir --det det.onnx --cls cls.onnx --rec rec.onnx --keys ppocr_keys_v1.txt --images DIR
//
// --ranks N: the same loop sharded over N GPUs of one node -- N processes (fork, one per GPU, device id = rank),
// rank 0 reads the model files and broadcasts them ONCE over RCCL through rt_rccl_unique_id / rt_broadcast_blobs (the id
// travels through a pipe), files are dealt to the ranks by size (greedy longest-processing-time), every output line carries
// the file's index in the sorted list so that `sort -n` restores the input order.  No per-batch communication.
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <filesystem>
#include <functional>
#include <future>
#include <memory>
#include <mutex>
#include <thread>
#include <string>
#include <vector>

#include <sys/wait.h>
#include <unistd.h>

#include "retto_hip.h"

struct Page { std::string path; int h = 0, w = 0; uint8_t* rgb = nullptr; };
struct Batch {
  std::vector<Page> pages;
  std::string error;
  ~Batch() { for (auto& p : pages) rt_buffer_free(p.rgb); }
};

static bool read_file(const std::string& path, std::vector<uint8_t>* bytes) {  // fs::read in main.rs:83
  FILE* f = fopen(path.c_str(), "rb");
  if (!f) return false;
  uint8_t buf[1 << 16];
  for (size_t n; (n = fread(buf, 1, sizeof(buf), f)) > 0;) bytes->insert(bytes->end(), buf, buf + n);
  const bool ok = !ferror(f);
  fclose(f);
  return ok;
}

// Reads and decodes files[b0, b1) on a few host threads (rt_decode_image); runs while the GPU works on the previous batch.
static std::unique_ptr<Batch> load_batch(const std::vector<std::string>& files, size_t b0, size_t b1, int threads) {
  std::unique_ptr<Batch> b(new Batch());
  b->pages.resize(b1 - b0);
  std::atomic<size_t> next{0};
  std::mutex mu;
  auto work = [&] {
    for (size_t i; (i = next.fetch_add(1)) < b1 - b0;) {
      Page& p = b->pages[i];
      p.path = files[b0 + i];
      std::vector<uint8_t> bytes;
      char err[256] = {0};
      std::string why;
      if (!read_file(p.path, &bytes)) why = "Failed to read image file " + p.path;
      else if (rt_decode_image(bytes.data(), bytes.size(), &p.rgb, &p.h, &p.w, err, sizeof(err)) != RT_OK) why = p.path + ": " + err;
      if (!why.empty()) { std::lock_guard<std::mutex> lk(mu); if (b->error.empty()) b->error = why; }
    }
  };
  std::vector<std::thread> th;
  for (int t = 1; t < threads; t++) th.emplace_back(work);
  work();
  for (auto& t : th) t.join();
  return b;
}

int main(int argc, char** argv) {
  std::string det, cls, rec, keys, images;
  int batch = 32, device = 0, ranks = 0;
  bool quiet = false;
  for (int i = 1; i + 1 < argc; i += 2) {
    std::string k = argv[i], v = argv[i + 1];
    if (k == "--det") det = v; else if (k == "--cls") cls = v; else if (k == "--rec") rec = v;
    else if (k == "--keys") keys = v; else if (k == "--images") images = v;
    else if (k == "--quiet") quiet = atoi(v.c_str()) != 0;
    else if (k == "--batch") batch = atoi(v.c_str()); else if (k == "--device-id") device = atoi(v.c_str());
    else if (k == "--ranks") ranks = atoi(v.c_str());
    else { fprintf(stderr, "unknown option %s\n", k.c_str()); return 2; }
  }
  if (det.empty() || cls.empty() || rec.empty() || keys.empty() || images.empty() || batch <= 0) {
    fprintf(stderr, "usage: retto_dir --det M --cls M --rec M --keys K --images DIR [--batch N] [--device-id D] [--ranks N] [--quiet 1]\n");
    return 2;
  }
  // the file list is the same on every rank (sorted walk), so is its split
  std::vector<std::string> files;
  for (auto& e : std::filesystem::recursive_directory_iterator(images))
    if (e.is_regular_file()) files.push_back(e.path().string());
  std::sort(files.begin(), files.end());
  std::vector<size_t> index(files.size());
  for (size_t i = 0; i < files.size(); i++) index[i] = i;

  int rank = 0, world = 1;
  std::vector<std::vector<uint8_t>> blobs(4);   // det, cls, rec, dict after the broadcast
  std::vector<void*> recv(4, nullptr);
  rt_config cfg;
  rt_config_default(&cfg);
  if (ranks > 0) {
    world = ranks;
    char uid[128];
    int fds[2];
    if (pipe(fds) != 0) { perror("pipe"); return 1; }
    std::vector<pid_t> kids;
    for (int r = 1; r < world; r++) {   // fork BEFORE anything touches the GPU
      pid_t p = fork();
      if (p < 0) { perror("fork"); return 1; }
      if (p == 0) { rank = r; break; }
      kids.push_back(p);
    }
    char err[256] = {0};
    if (rank == 0) {
      if (rt_rccl_unique_id(uid, sizeof(uid), err, sizeof(err)) != RT_OK) { fprintf(stderr, "rt_rccl_unique_id: %s\n", err); return 1; }
      for (int r = 1; r < world; r++) if (write(fds[1], uid, sizeof(uid)) != (ssize_t)sizeof(uid)) { perror("write"); return 1; }
      const std::string* src[4] = {&det, &cls, &rec, &keys};
      for (int i = 0; i < 4; i++) if (!read_file(*src[i], &blobs[(size_t)i])) { fprintf(stderr, "cannot read %s\n", src[i]->c_str()); return 1; }
    } else if (read(fds[0], uid, sizeof(uid)) != (ssize_t)sizeof(uid)) { perror("read"); return 1; }
    void* data[4]; size_t lens[4];
    for (int i = 0; i < 4; i++) { data[i] = blobs[(size_t)i].data(); lens[i] = blobs[(size_t)i].size(); }
    if (rt_broadcast_blobs(uid, rank, world, /*device*/ rank, /*root*/ 0, 4, data, lens, err, sizeof(err)) != RT_OK) {
      fprintf(stderr, "rank %d: rt_broadcast_blobs: %s\n", rank, err); return 1;
    }
    if (rank != 0) for (int i = 0; i < 4; i++) recv[(size_t)i] = data[i];
    device = rank;
    rt_model_source* ms[4] = {&cfg.det, &cfg.cls, &cfg.rec, &cfg.dict};
    for (int i = 0; i < 4; i++) { ms[i]->path = nullptr; ms[i]->data = data[i]; ms[i]->len = lens[i]; }
    // greedy longest-processing-time split by file size (a stand-in for the page's pixel count)
    std::vector<std::pair<uintmax_t, size_t>> by_size;
    for (size_t i = 0; i < files.size(); i++) by_size.push_back({std::filesystem::file_size(files[i]), i});
    std::sort(by_size.begin(), by_size.end(), [](auto& a, auto& b) { return a.first != b.first ? a.first > b.first : a.second < b.second; });
    std::vector<uintmax_t> load((size_t)world, 0);
    std::vector<size_t> mine;
    for (auto& f : by_size) {
      size_t o = (size_t)(std::min_element(load.begin(), load.end()) - load.begin());
      load[o] += f.first + 1;
      if ((int)o == rank) mine.push_back(f.second);
    }
    std::sort(mine.begin(), mine.end());
    std::vector<std::string> my_files;
    for (size_t i : mine) my_files.push_back(files[i]);
    files.swap(my_files); index.swap(mine);
    if (rank == 0) {   // the parent also waits for its children at exit
      struct Reaper { std::vector<pid_t> k; ~Reaper() { for (pid_t p : k) { int st; waitpid(p, &st, 0); } } };
      static Reaper reaper; reaper.k = kids;
    }
  } else {
    cfg.det.path = det.c_str(); cfg.cls.path = cls.c_str(); cfg.rec.path = rec.c_str(); cfg.dict.path = keys.c_str();
  }
  cfg.device_id = device;
  rt_session* s = nullptr;
  int rc = rt_create(&cfg, &s);
  for (void* p : recv) rt_buffer_free(p);
  if (rc != RT_OK) { fprintf(stderr, "rt_create failed (%d): %s\n", rc, rt_last_error(nullptr)); return 1; }
  fprintf(stderr, "[rank %d/%d] %zu files, processing...\n", rank, world, files.size());
  size_t done = 0;
  const int threads = std::max(1, std::min(16, (int)std::thread::hardware_concurrency()));
  const auto t0 = std::chrono::steady_clock::now();
  auto range_end = [&](size_t b0) { return std::min(files.size(), b0 + (size_t)batch); };
  std::future<std::unique_ptr<Batch>> next;
  if (!files.empty()) next = std::async(std::launch::async, load_batch, std::cref(files), (size_t)0, range_end(0), threads);
  for (size_t b0 = 0; b0 < files.size(); b0 += (size_t)batch) {
    std::unique_ptr<Batch> cur = next.get();
    if (range_end(b0) < files.size())  // decode the next batch while this one is on the GPU
      next = std::async(std::launch::async, load_batch, std::cref(files), range_end(b0), range_end(range_end(b0)), threads);
    if (!cur->error.empty()) { fprintf(stderr, "%s\n", cur->error.c_str()); if (next.valid()) next.wait(); rt_destroy(s); return 1; }
    std::vector<const uint8_t*> ptr; std::vector<int> hs, ws;
    for (auto& p : cur->pages) { ptr.push_back(p.rgb); hs.push_back(p.h); ws.push_back(p.w); }
    rt_results* r = nullptr;
    rc = rt_run_batch(s, ptr.data(), hs.data(), ws.data(), (int)cur->pages.size(), RT_MEM_HOST, nullptr, &r);
    if (rc != RT_OK) { fprintf(stderr, "rt_run_batch failed (%d): %s\n", rc, rt_last_error(s)); if (next.valid()) next.wait(); rt_destroy(s); return 1; }
    if (!quiet)
      for (int i = 0; i < rt_results_pages(r); i++)
        printf("%zu {\"file\":\"%s\",\"det\":%s,\"cls\":%s,\"rec\":%s}\n", index[b0 + (size_t)i], cur->pages[(size_t)i].path.c_str(),
               rt_results_json(r, i, 0), rt_results_json(r, i, 1), rt_results_json(r, i, 2));
    done += cur->pages.size();
    rt_results_free(r);
  }
  const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  if (done) fprintf(stderr, "Successfully processed %zu images, avg time: %.2fms\n", done, ms / done);
  rt_destroy(s);
  return 0;
}
