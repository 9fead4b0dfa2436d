// Native directory driver over the C ABI (include/retto_hip.h): the loop of retto-cli
// (/root/reference/retto-cli/src/main.rs:41-95) without Python -- walk a directory, read every page,
// run the pages through rt_run_batch in batches, print the three stage results per image in the wire
// format retto-wasm emits (retto-wasm/fe/index.ts:5-42) and the average time per image.
// Files are PNG / JPEG / PNM / BMP: the library's host decoder (rt_decode_image; image_helper.rs:34-44 is host code in
// the reference too) runs on a few threads for batch i+1 while batch i is on the GPU.
//
//   g++ -std=c++17 -Iinclude examples/retto_dir.cpp -Lretto_amd -lretto_hip -Wl,-rpath,$PWD/retto_amd -o examples/retto_dir
//   examples/retto_dir --det det.onnx --cls cls.onnx --rec rec.onnx --keys ppocr_keys_v1.txt --images DIR
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
  int batch = 32, device = 0;
  bool quiet = false;
  for (int i = 1; i + 1 < argc; i += 2) {
    std::string k = argv[i], v = argv[i + 1];
    if (k == "--det") det = v; else if (k == "--cls") cls = v; else if (k == "--rec") rec = v;
    else if (k == "--keys") keys = v; else if (k == "--images") images = v;
    else if (k == "--quiet") quiet = atoi(v.c_str()) != 0;
    else if (k == "--batch") batch = atoi(v.c_str()); else if (k == "--device-id") device = atoi(v.c_str());
    else { fprintf(stderr, "unknown option %s\n", k.c_str()); return 2; }
  }
  if (det.empty() || cls.empty() || rec.empty() || keys.empty() || images.empty() || batch <= 0) {
    fprintf(stderr, "usage: retto_dir --det M --cls M --rec M --keys K --images DIR [--batch N] [--device-id D] [--quiet 1]\n");
    return 2;
  }
  rt_config cfg;
  rt_config_default(&cfg);
  cfg.device_id = device;
  cfg.det.path = det.c_str(); cfg.cls.path = cls.c_str(); cfg.rec.path = rec.c_str(); cfg.dict.path = keys.c_str();
  rt_session* s = nullptr;
  int rc = rt_create(&cfg, &s);
  if (rc != RT_OK) { fprintf(stderr, "rt_create failed (%d): %s\n", rc, rt_last_error(nullptr)); return 1; }

  std::vector<std::string> files;
  for (auto& e : std::filesystem::recursive_directory_iterator(images))
    if (e.is_regular_file()) files.push_back(e.path().string());
  std::sort(files.begin(), files.end());
  fprintf(stderr, "Found %zu files, processing...\n", files.size());
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
        printf("{\"file\":\"%s\",\"det\":%s,\"cls\":%s,\"rec\":%s}\n", cur->pages[(size_t)i].path.c_str(), rt_results_json(r, i, 0),
               rt_results_json(r, i, 1), rt_results_json(r, i, 2));
    done += cur->pages.size();
    rt_results_free(r);
  }
  const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  if (done) fprintf(stderr, "Successfully processed %zu images, avg time: %.2fms\n", done, ms / done);
  rt_destroy(s);
  return 0;
}
