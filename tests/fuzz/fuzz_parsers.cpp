// Test infrastructure (CPU only): drives the two parsers of untrusted bytes -- the image decoder and the ONNX
// importer -- with mutated inputs under AddressSanitizer / UBSan.  Every outcome other than "decoded" or
// "RtError" (crash, sanitizer report, foreign exception) fails the run.
//   fuzz_parsers image|onnx<which> ITER SEED file...
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>

#include "../../retto_amd/csrc/common.h"
#include "../../retto_amd/csrc/image_decode.h"
#include "../../retto_amd/csrc/onnx_import.h"

// The importer matches against the layer manifest, which lives next to the network code; the harness takes it from the
// built library's C ABI so that only the two parsers are compiled under the sanitizers.
extern "C" size_t rt_model_manifest(int which, char* buf, size_t cap);
namespace rt {
std::vector<ManifestEntry> model_manifest(int which) {
  std::vector<char> buf(rt_model_manifest(which, nullptr, 0) + 1);
  rt_model_manifest(which, buf.data(), buf.size());
  std::vector<ManifestEntry> out;
  for (char* line = strtok(buf.data(), "\n"); line; line = strtok(nullptr, "\n")) {
    ManifestEntry e;
    char* sp = strchr(line, ' ');
    e.name = sp ? std::string(line, sp) : std::string(line);
    while (sp) { e.dims.push_back(atoi(sp + 1)); sp = strchr(sp + 1, ' '); }
    out.push_back(e);
  }
  return out;
}
}  // namespace rt

static std::vector<uint8_t> slurp(const char* path) {
  std::vector<uint8_t> b;
  FILE* f = fopen(path, "rb");
  if (!f) { fprintf(stderr, "cannot open %s\n", path); exit(2); }
  uint8_t buf[65536];
  for (size_t n; (n = fread(buf, 1, sizeof(buf), f)) > 0;) b.insert(b.end(), buf, buf + n);
  fclose(f);
  return b;
}

int main(int argc, char** argv) {
  if (argc < 5) return 2;
  const std::string mode = argv[1];
  const int iters = atoi(argv[2]);
  std::mt19937_64 rng((uint64_t)atoll(argv[3]));
  std::vector<std::vector<uint8_t>> seeds;
  for (int i = 4; i < argc; i++) seeds.push_back(slurp(argv[i]));
  long ok = 0, rejected = 0;
  for (int it = 0; it < iters; it++) {
    std::vector<uint8_t> b = seeds[rng() % seeds.size()];
    const int kind = (int)(rng() % 6);
    if (kind == 0 && !b.empty()) b.resize(rng() % b.size());                                    // truncate
    else if (kind == 1) { for (int k = 0, n = 1 + (int)(rng() % 8); k < n && !b.empty(); k++) b[rng() % b.size()] ^= (uint8_t)(1u << (rng() % 8)); }
    else if (kind == 2) { for (int k = 0, n = 1 + (int)(rng() % 4); k < n && !b.empty(); k++) b[rng() % b.size()] = (uint8_t)rng(); }
    else if (kind == 3 && b.size() > 8) { size_t p = rng() % (b.size() - 4); uint32_t v = (rng() & 1) ? 0xffffffffu : (uint32_t)rng(); memcpy(&b[p], &v, 4); }  // wild length / dimension
    else if (kind == 4 && b.size() > 16) { size_t p = rng() % (b.size() - 8), n = 1 + rng() % 8; b.erase(b.begin() + (long)p, b.begin() + (long)(p + n)); }
    else if (kind == 5 && b.size() > 16) { size_t p = rng() % b.size(); b.insert(b.begin() + (long)p, (size_t)(1 + rng() % 16), (uint8_t)rng()); }
    try {
      if (mode == "image") {
        std::vector<uint8_t> rgb; int h = 0, w = 0;
        rt::decode_image(b.data(), b.size(), &rgb, &h, &w);
        if (rgb.size() != (size_t)h * w * 3) { fprintf(stderr, "size mismatch\n"); return 1; }
      } else {
        rt::onnx_to_rtwb(mode.back() - '0', b.data(), b.size());
      }
      ok++;
    } catch (const rt::RtError&) { rejected++; }
  }
  printf("%ld accepted, %ld rejected\n", ok, rejected);
  return 0;
}
