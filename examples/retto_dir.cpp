// Native directory driver over the C ABI (include/retto_hip.h): the loop of retto-cli
// (/root/reference/retto-cli/src/main.rs:41-95) without Python -- walk a directory, read every page,
// run the pages through rt_run_batch in batches, print the three stage results per image in the wire
// format retto-wasm emits (retto-wasm/fe/index.ts:5-42) and the average time per image.
// Pages are binary PPM (P6, maxval 255): this image has no PNG/JPEG library, and the reference decodes
// on the host as well (image_helper.rs:34-44), so decode stays outside the library.
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

struct Page { std::string path; int h = 0, w = 0; std::vector<uint8_t> rgb; };

static bool read_ppm(const std::string& path, Page* p) {
  FILE* f = fopen(path.c_str(), "rb");
  if (!f) return false;
  char magic[3] = {0};
  int w = 0, h = 0, maxv = 0;
  bool ok = fscanf(f, "%2s", magic) == 1 && strcmp(magic, "P6") == 0;
  for (int* v : {&w, &h, &maxv}) {
    if (!ok) break;
    int c = fgetc(f);
    while (c == ' ' || c == '\n' || c == '\r' || c == '\t' || c == '#') {
      if (c == '#') while (c != '\n' && c != EOF) c = fgetc(f);
      c = fgetc(f);
    }
    ungetc(c, f);
    ok = fscanf(f, "%d", v) == 1;
  }
  ok = ok && maxv == 255 && w > 0 && h > 0 && fgetc(f) != EOF;
  if (ok) {
    p->path = path; p->h = h; p->w = w;
    p->rgb.resize((size_t)h * w * 3);
    ok = fread(p->rgb.data(), 1, p->rgb.size(), f) == p->rgb.size();
  }
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
      if (!read_ppm(files[i], &p)) { fprintf(stderr, "Failed to decode image %s\n", files[i].c_str()); rt_destroy(s); return 1; }
      pages.push_back(std::move(p));
    }
    std::vector<const uint8_t*> ptr; std::vector<int> hs, ws;
    for (auto& p : pages) { ptr.push_back(p.rgb.data()); hs.push_back(p.h); ws.push_back(p.w); }
    rt_results* r = nullptr;
    rc = rt_run_batch(s, ptr.data(), hs.data(), ws.data(), (int)pages.size(), RT_MEM_HOST, nullptr, &r);
    if (rc != RT_OK) { fprintf(stderr, "rt_run_batch failed (%d): %s\n", rc, rt_last_error(s)); rt_destroy(s); return 1; }
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
