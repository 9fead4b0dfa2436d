// Native directory driver over the C ABI (include/retto_hip.h): the loop of retto-cli
// (/root/reference/retto-cli/src/main.rs:41-95) without Python -- walk a directory, read every page,
// run the pages through rt_run_batch in batches, print the three stage results per image in the wire
// format retto-wasm emits (retto-wasm/fe/index.ts:5-42) and the average time per image.
// Files are handed over encoded, as RettoSession::run takes them (session.rs:108): the library decodes PNG /
// JPEG / PNM / BMP on host threads (rt_run_encoded_batch; image_helper.rs:34-44 is host code in the reference too).
//
//   g++ -std=c++17 -Iinclude examples/retto_dir.cpp -Lretto_amd -lretto_hip -Wl,-rpath,$PWD/retto_amd -o examples/retto_dir
//   examples/retto_dir --det det.onnx --cls cls.onnx --rec rec.onnx --keys ppocr_keys_v1.txt --images DIR
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <filesystem>
#include <string>
#include <vector>

#include "retto_hip.h"

struct Page { std::string path; std::vector<uint8_t> bytes; };

static bool read_file(const std::string& path, Page* p) {  // fs::read in main.rs:83
  FILE* f = fopen(path.c_str(), "rb");
  if (!f) return false;
  p->path = path;
  uint8_t buf[1 << 16];
  for (size_t n; (n = fread(buf, 1, sizeof(buf), f)) > 0;) p->bytes.insert(p->bytes.end(), buf, buf + n);
  const bool ok = !ferror(f);
  fclose(f);
  return ok;
}

int main(int argc, char** argv) {
  std::string det, cls, rec, keys, images;
  int batch = 32, device = 0;
  for (int i = 1; i + 1 < argc; i += 2) {
    std::string k = argv[i], v = argv[i + 1];
    if (k == "--det") det = v; else if (k == "--cls") cls = v; else if (k == "--rec") rec = v;
    else if (k == "--keys") keys = v; else if (k == "--images") images = v;
    else if (k == "--batch") batch = atoi(v.c_str()); else if (k == "--device-id") device = atoi(v.c_str());
    else { fprintf(stderr, "unknown option %s\n", k.c_str()); return 2; }
  }
  if (det.empty() || cls.empty() || rec.empty() || keys.empty() || images.empty() || batch <= 0) {
    fprintf(stderr, "usage: retto_dir --det M --cls M --rec M --keys K --images DIR [--batch N] [--device-id D]\n");
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
  const auto t0 = std::chrono::steady_clock::now();
  for (size_t b0 = 0; b0 < files.size(); b0 += (size_t)batch) {
    std::vector<Page> pages;
    for (size_t i = b0; i < std::min(files.size(), b0 + (size_t)batch); i++) {
      Page p;
      if (!read_file(files[i], &p)) { fprintf(stderr, "Failed to read image file %s\n", files[i].c_str()); rt_destroy(s); return 1; }
      pages.push_back(std::move(p));
    }
    std::vector<const void*> ptr; std::vector<size_t> lens;
    for (auto& p : pages) { ptr.push_back(p.bytes.data()); lens.push_back(p.bytes.size()); }
    rt_results* r = nullptr;
    rc = rt_run_encoded_batch(s, ptr.data(), lens.data(), (int)pages.size(), nullptr, nullptr, &r);  // decode (host threads) + pipeline
    if (rc != RT_OK) { fprintf(stderr, "rt_run_encoded_batch failed (%d): %s\n", rc, rt_last_error(s)); rt_destroy(s); return 1; }
    for (int i = 0; i < rt_results_pages(r); i++)
      printf("{\"file\":\"%s\",\"det\":%s,\"cls\":%s,\"rec\":%s}\n", pages[i].path.c_str(), rt_results_json(r, i, 0),
             rt_results_json(r, i, 1), rt_results_json(r, i, 2));
    done += pages.size();
    rt_results_free(r);
  }
  const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  if (done) fprintf(stderr, "Successfully processed %zu images, avg time: %.2fms\n", done, ms / done);
  rt_destroy(s);
  return 0;
}
