// oracle/nets_cpu.cpp -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
//
// C++ / OpenMP fp32 restatement of the three PP-OCRv4 mobile networks the reference runs through ONNX Runtime
// (RettoInnerWorker::{det,cls,rec}, /root/reference/retto-core/src/worker.rs:69-73; ORT calls
// /root/reference/retto-core/src/worker/ort_worker.rs:189-220), driven by the same RTWB weight blobs as the HIP path
// (format: retto_amd/synth.py).  It exists for ONE purpose: bench.py's `cpu_baseline` leg -- the stand-in SURVEY.md
// section 8(d) specifies for the reference's ort-CPU path, which cannot be run here (no Rust, no ONNX Runtime, no model
// files).  Same graphs as oracle/nets_torch.py (the fp32 oracle of record; tests/test_oracle_cpu.py checks this file
// against it).  PARITY UNPINNED, like every oracle in this directory.
//
// Layout NHWC; every conv is a direct loop nest over (row block | 8 output pixels | taps | input channel | output channels)
// with the output-channel loop innermost and contiguous, so the compiler vectorises it; rows are spread over the OpenMP
// team of the calling thread.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this.
#include <math.h>
#include <omp.h>
#include <stdint.h>
#include <string.h>

#include <algorithm>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>

namespace {

struct Tn { std::vector<int> dims; const float* d = nullptr; size_t numel = 0; };
struct Blob {
  std::vector<uint8_t> bytes;
  std::map<std::string, Tn> t;
  void parse(const void* p, size_t n) {
    bytes.assign((const uint8_t*)p, (const uint8_t*)p + n);
    const uint8_t* b = bytes.data();
    if (n < 16 || memcmp(b, "RTWB", 4) != 0) throw std::runtime_error("not an RTWB blob");
    uint32_t ver, cnt; memcpy(&ver, b + 4, 4); memcpy(&cnt, b + 8, 4);
    if (ver != 1) throw std::runtime_error("RTWB version");
    size_t q = 16;
    struct E { std::string name; std::vector<int> dims; uint64_t off, nb; };
    std::vector<E> es;
    for (uint32_t i = 0; i < cnt; i++) {
      uint16_t ln; memcpy(&ln, b + q, 2); q += 2;
      E e; e.name.assign((const char*)b + q, ln); q += ln;
      uint8_t nd = b[q]; q += 4;
      for (int k = 0; k < nd; k++) { uint32_t d; memcpy(&d, b + q, 4); q += 4; e.dims.push_back((int)d); }
      memcpy(&e.off, b + q, 8); memcpy(&e.nb, b + q + 8, 8); q += 16;
      es.push_back(e);
    }
    size_t base = (q + 63) / 64 * 64;
    for (auto& e : es) {
      if (base + e.off + e.nb > n) throw std::runtime_error("RTWB truncated");
      Tn x; x.dims = e.dims; x.d = (const float*)(b + base + e.off); x.numel = e.nb / 4;
      t[e.name] = x;
    }
  }
  const Tn& get(const std::string& n) const { auto it = t.find(n); if (it == t.end()) throw std::runtime_error("missing tensor " + n); return it->second; }
  bool has(const std::string& n) const { return t.count(n) != 0; }
};

struct T4 {  // NHWC activation
  int n = 0, h = 0, w = 0, c = 0;
  std::vector<float> d;
  T4() {}
  T4(int n_, int h_, int w_, int c_) : n(n_), h(h_), w(w_), c(c_), d((size_t)n_ * h_ * w_ * c_) {}
  float* at(int i, int y, int x) { return d.data() + (((size_t)i * h + y) * w + x) * c; }
  const float* at(int i, int y, int x) const { return d.data() + (((size_t)i * h + y) * w + x) * c; }
};

enum { ACT_NONE = 0, ACT_RELU, ACT_HSWISH, ACT_SWISH, ACT_SIGMOID };
static inline float actf(float v, int a) {
  switch (a) {
    case ACT_RELU: return v > 0.f ? v : 0.f;
    case ACT_HSWISH: return v * std::min(std::max(v + 3.f, 0.f), 6.f) / 6.f;
    case ACT_SWISH: return v / (1.f + expf(-v));
    case ACT_SIGMOID: return 1.f / (1.f + expf(-v));
    default: return v;
  }
}
struct Lab { bool has = false; float a = 1.f, c = 0.f; };
static Lab get_lab(const Blob& b, const std::string& n) {
  Lab l; if (b.has(n + ".a")) { l.has = true; l.a = b.get(n + ".a").d[0]; l.c = b.get(n + ".c").d[0]; } return l;
}

// dense conv, weights [cout][cin][kh][kw] (repacked to [kh][kw][cin][cout] per call site once, cached by name)
struct ConvW { int cout, cin, kh, kw; std::vector<float> w; std::vector<float> b; };
static ConvW pack_conv(const Blob& bl, const std::string& name) {
  const Tn& t = bl.get(name + ".w");
  ConvW c; c.cout = t.dims[0]; c.cin = t.dims[1]; c.kh = t.dims[2]; c.kw = t.dims[3];
  c.w.resize(t.numel);
  for (int o = 0; o < c.cout; o++)
    for (int i = 0; i < c.cin; i++)
      for (int k = 0; k < c.kh * c.kw; k++) c.w[((size_t)k * c.cin + i) * c.cout + o] = t.d[((size_t)o * c.cin + i) * c.kh * c.kw + k];
  c.b.assign(c.cout, 0.f);
  if (bl.has(name + ".b")) memcpy(c.b.data(), bl.get(name + ".b").d, c.cout * sizeof(float));
  return c;
}
static ConvW pack_linear(const Blob& bl, const std::string& name) {  // [in][out]
  const Tn& t = bl.get(name + ".w");
  ConvW c; c.cin = t.dims[0]; c.cout = t.dims[1]; c.kh = c.kw = 1;
  c.w.assign(t.d, t.d + t.numel);
  c.b.assign(bl.get(name + ".b").d, bl.get(name + ".b").d + c.cout);
  return c;
}

constexpr int PB = 8;  // output pixels per register block

static T4 conv(const T4& x, const ConvW& cw, int sh, int sw, int ph, int pw, int act, const Lab* lab = nullptr, const T4* residual = nullptr) {
  const int oh = (x.h + 2 * ph - cw.kh) / sh + 1, ow = (x.w + 2 * pw - cw.kw) / sw + 1;
  T4 y(x.n, oh, ow, cw.cout);
  const int N = cw.cout, K = cw.cin;
  const long long rows = (long long)x.n * oh;
#pragma omp parallel
  {
    std::vector<float> acc((size_t)PB * N);
#pragma omp for schedule(dynamic, 1)
    for (long long r = 0; r < rows; r++) {
      const int img = (int)(r / oh), oy = (int)(r % oh);
      for (int ox0 = 0; ox0 < ow; ox0 += PB) {
        const int np = std::min(PB, ow - ox0);
        for (int p = 0; p < np; p++) memcpy(&acc[(size_t)p * N], cw.b.data(), N * sizeof(float));
        for (int dy = 0; dy < cw.kh; dy++) {
          const int iy = oy * sh + dy - ph;
          if (iy < 0 || iy >= x.h) continue;
          for (int dx = 0; dx < cw.kw; dx++) {
            const float* wt = cw.w.data() + (size_t)(dy * cw.kw + dx) * K * N;
            const float* src[PB]; bool ok[PB];
            for (int p = 0; p < np; p++) {
              const int ix = (ox0 + p) * sw + dx - pw;
              ok[p] = ix >= 0 && ix < x.w;
              src[p] = ok[p] ? x.at(img, iy, ix) : nullptr;
            }
            for (int k = 0; k < K; k++) {
              const float* wr = wt + (size_t)k * N;
              for (int p = 0; p < np; p++) {
                if (!ok[p]) continue;
                const float a = src[p][k];
                float* ac = &acc[(size_t)p * N];
#pragma omp simd
                for (int n = 0; n < N; n++) ac[n] += a * wr[n];
              }
            }
          }
        }
        for (int p = 0; p < np; p++) {
          float* o = y.at(img, oy, ox0 + p);
          const float* rs = residual ? residual->at(img, oy, ox0 + p) : nullptr;
          for (int n = 0; n < N; n++) {
            float v = actf(acc[(size_t)p * N + n], act);
            if (lab && lab->has) v = v * lab->a + lab->c;
            if (rs) v += rs[n];
            o[n] = v;
          }
        }
      }
    }
  }
  return y;
}

struct DwW { int c, k; std::vector<float> w, b; };  // [k*k][c]
static DwW pack_dw(const Blob& bl, const std::string& name) {
  const Tn& t = bl.get(name + ".w");
  DwW d; d.c = t.dims[0]; d.k = t.dims[2];
  d.w.resize((size_t)d.k * d.k * d.c);
  for (int c = 0; c < d.c; c++)
    for (int q = 0; q < d.k * d.k; q++) d.w[(size_t)q * d.c + c] = t.d[(size_t)c * d.k * d.k + q];
  d.b.assign(bl.get(name + ".b").d, bl.get(name + ".b").d + d.c);
  return d;
}
static T4 dwconv(const T4& x, const DwW& d, int sh, int sw, int act, const Lab& lab) {
  const int p = d.k / 2, oh = (x.h + 2 * p - d.k) / sh + 1, ow = (x.w + 2 * p - d.k) / sw + 1, Cn = d.c;
  T4 y(x.n, oh, ow, Cn);
  const long long rows = (long long)x.n * oh;
#pragma omp parallel for schedule(dynamic, 1)
  for (long long r = 0; r < rows; r++) {
    const int img = (int)(r / oh), oy = (int)(r % oh);
    for (int ox = 0; ox < ow; ox++) {
      float* o = y.at(img, oy, ox);
      memcpy(o, d.b.data(), Cn * sizeof(float));
      for (int dy = 0; dy < d.k; dy++) {
        const int iy = oy * sh + dy - p;
        if (iy < 0 || iy >= x.h) continue;
        for (int dx = 0; dx < d.k; dx++) {
          const int ix = ox * sw + dx - p;
          if (ix < 0 || ix >= x.w) continue;
          const float* s = x.at(img, iy, ix);
          const float* wv = d.w.data() + (size_t)(dy * d.k + dx) * Cn;
#pragma omp simd
          for (int c = 0; c < Cn; c++) o[c] += s[c] * wv[c];
        }
      }
      for (int c = 0; c < Cn; c++) { float v = actf(o[c], act); o[c] = lab.has ? v * lab.a + lab.c : v; }
    }
  }
  return y;
}

struct SeW { int c, cr; std::vector<float> w1, b1, w2, b2; };  // w1 [cr][c], w2 [c][cr]
static SeW pack_se(const Blob& bl, const std::string& name) {
  SeW s; const Tn& a = bl.get(name + ".fc1.w"); s.cr = a.dims[0]; s.c = a.dims[1];
  s.w1.assign(a.d, a.d + a.numel); s.b1.assign(bl.get(name + ".fc1.b").d, bl.get(name + ".fc1.b").d + s.cr);
  const Tn& b = bl.get(name + ".fc2.w"); s.w2.assign(b.d, b.d + b.numel); s.b2.assign(bl.get(name + ".fc2.b").d, bl.get(name + ".fc2.b").d + s.c);
  return s;
}
// x *= hardsigmoid(fc2(relu(fc1(mean(x)))))  (+ x when residual: RSELayer)
static void se_apply(T4& x, const SeW& s, float slope, bool residual) {
  for (int i = 0; i < x.n; i++) {
    std::vector<double> m(x.c, 0.0);
    for (int y = 0; y < x.h; y++) for (int xx = 0; xx < x.w; xx++) { const float* p = x.at(i, y, xx); for (int c = 0; c < x.c; c++) m[c] += p[c]; }
    std::vector<float> mean(x.c), hid(s.cr), g(x.c);
    for (int c = 0; c < x.c; c++) mean[c] = (float)(m[c] / ((double)x.h * x.w));
    for (int j = 0; j < s.cr; j++) { float v = s.b1[j]; for (int c = 0; c < x.c; c++) v += mean[c] * s.w1[(size_t)j * x.c + c]; hid[j] = v > 0.f ? v : 0.f; }
    for (int c = 0; c < x.c; c++) {
      float v = s.b2[c]; for (int j = 0; j < s.cr; j++) v += hid[j] * s.w2[(size_t)c * s.cr + j];
      g[c] = std::min(std::max(v * slope + 0.5f, 0.f), 1.f) + (residual ? 1.f : 0.f);
    }
#pragma omp parallel for
    for (int y = 0; y < x.h; y++) for (int xx = 0; xx < x.w; xx++) { float* p = x.at(i, y, xx); for (int c = 0; c < x.c; c++) p[c] *= g[c]; }
  }
}

static T4 from_nchw(const float* src, int n, int c, int h, int w) {
  T4 t(n, h, w, c);
  for (int i = 0; i < n; i++)
    for (int k = 0; k < c; k++)
      for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) t.at(i, y, x)[k] = src[(((size_t)i * c + k) * h + y) * w + x];
  return t;
}
static T4 up_nearest(const T4& x, int s) {
  T4 y(x.n, x.h * s, x.w * s, x.c);
  for (int i = 0; i < x.n; i++) for (int yy = 0; yy < y.h; yy++) for (int xx = 0; xx < y.w; xx++) memcpy(y.at(i, yy, xx), x.at(i, yy / s, xx / s), x.c * sizeof(float));
  return y;
}
static void add_inplace(T4& a, const T4& b) { for (size_t i = 0; i < a.d.size(); i++) a.d[i] += b.d[i]; }

struct LcBlk { DwW dw; Lab dw_lab; bool dw_act; bool se; SeW sew; ConvW pw; Lab pw_lab; int sh, sw; };
struct Spec { const char* name; int k, cin, cout, sh, sw; bool se; };
static const Spec DET_SPEC[] = {
    {"s2.0", 3, 16, 32, 1, 1, false}, {"s3.0", 3, 32, 48, 2, 2, false}, {"s3.1", 3, 48, 48, 1, 1, false},
    {"s4.0", 3, 48, 96, 2, 2, false}, {"s4.1", 3, 96, 96, 1, 1, false}, {"s5.0", 3, 96, 192, 2, 2, false},
    {"s5.1", 5, 192, 192, 1, 1, false}, {"s5.2", 5, 192, 192, 1, 1, false}, {"s5.3", 5, 192, 192, 1, 1, false},
    {"s5.4", 5, 192, 192, 1, 1, false}, {"s6.0", 5, 192, 384, 2, 2, true}, {"s6.1", 5, 384, 384, 1, 1, true},
    {"s6.2", 5, 384, 384, 1, 1, false}, {"s6.3", 5, 384, 384, 1, 1, false}};
static const Spec REC_SPEC[] = {
    {"s2.0", 3, 16, 32, 1, 1, false}, {"s3.0", 3, 32, 64, 1, 1, false}, {"s3.1", 3, 64, 64, 1, 1, false},
    {"s4.0", 3, 64, 128, 2, 1, false}, {"s4.1", 3, 128, 128, 1, 1, false}, {"s5.0", 3, 128, 240, 1, 2, false},
    {"s5.1", 5, 240, 240, 1, 1, false}, {"s5.2", 5, 240, 240, 1, 1, false}, {"s5.3", 5, 240, 240, 1, 1, false},
    {"s5.4", 5, 240, 240, 1, 1, false}, {"s6.0", 5, 240, 480, 2, 1, true}, {"s6.1", 5, 480, 480, 1, 1, true},
    {"s6.2", 5, 480, 480, 2, 1, false}, {"s6.3", 5, 480, 480, 1, 1, false}};
static LcBlk build_lc(const Blob& b, const std::string& p, const Spec& s) {
  LcBlk k; k.dw = pack_dw(b, p + ".dw"); k.dw_lab = get_lab(b, p + ".dw"); k.dw_act = k.dw_lab.has;
  k.se = s.se; if (s.se) k.sew = pack_se(b, p + ".se");
  k.pw = pack_conv(b, p + ".pw"); k.pw_lab = get_lab(b, p + ".pw"); k.sh = s.sh; k.sw = s.sw;
  return k;
}
static T4 run_lc(const LcBlk& k, const T4& x) {
  T4 y = dwconv(x, k.dw, k.sh, k.sw, k.dw_act ? ACT_HSWISH : ACT_NONE, k.dw_lab);
  if (k.se) se_apply(y, k.sew, 0.1666667f, false);
  return conv(y, k.pw, 1, 1, 0, 0, ACT_HSWISH, &k.pw_lab);
}

struct Nets {
  // det
  ConvW d_stem; std::vector<LcBlk> d_blk; ConvW d_out[4], d_ins[4], d_inp[4], d_head; SeW d_ins_se[4], d_inp_se[4];
  std::vector<float> dc1_w, dc1_b, dc2_w; float dc2_b = 0.f;
  // rec
  ConvW r_stem; std::vector<LcBlk> r_blk; ConvW r_c1, r_c2, r_c3, r_c4, r_c11, r_fc;
  struct Mix { ConvW qkv, proj, fc1, fc2; std::vector<float> n1g, n1b, n2g, n2b; } mix[2];
  std::vector<float> ng, nb;
  // cls
  ConvW c_stem, c_conv2, c_fc;
  struct CB { ConvW expand, linear; DwW dw; bool se; SeW sew; int act, sh, sw; bool shortcut; };
  std::vector<CB> c_blk;
  Blob bd, bc, br;
};
static std::vector<float> vec(const Blob& b, const std::string& n) { const Tn& t = b.get(n); return std::vector<float>(t.d, t.d + t.numel); }

static void layer_norm(float* x, int rows, int C, const std::vector<float>& g, const std::vector<float>& b, float eps) {
#pragma omp parallel for
  for (int r = 0; r < rows; r++) {
    float* p = x + (size_t)r * C;
    float m = 0.f; for (int c = 0; c < C; c++) m += p[c]; m /= C;
    float v = 0.f; for (int c = 0; c < C; c++) v += (p[c] - m) * (p[c] - m); v /= C;
    const float inv = 1.f / sqrtf(v + eps);
    for (int c = 0; c < C; c++) p[c] = (p[c] - m) * inv * g[c] + b[c];
  }
}

}  // namespace

#define OCPU_API extern "C" __attribute__((visibility("default")))

OCPU_API void* ocpu_create(const void* det, size_t dn, const void* cls, size_t cn, const void* rec, size_t rn) {
  try {
    Nets* N = new Nets();
    N->bd.parse(det, dn); N->bc.parse(cls, cn); N->br.parse(rec, rn);
    const Blob& d = N->bd;
    N->d_stem = pack_conv(d, "det.stem");
    for (const Spec& s : DET_SPEC) N->d_blk.push_back(build_lc(d, std::string("det.") + s.name, s));
    for (int j = 0; j < 4; j++) {
      const std::string js = std::to_string(j);
      N->d_out[j] = pack_conv(d, "det.out" + js); N->d_ins[j] = pack_conv(d, "det.fpn.ins" + js); N->d_inp[j] = pack_conv(d, "det.fpn.inp" + js);
      N->d_ins_se[j] = pack_se(d, "det.fpn.ins" + js + ".se"); N->d_inp_se[j] = pack_se(d, "det.fpn.inp" + js + ".se");
    }
    N->d_head = pack_conv(d, "det.head.conv1");
    N->dc1_w = vec(d, "det.head.deconv1.w"); N->dc1_b = vec(d, "det.head.deconv1.b");
    N->dc2_w = vec(d, "det.head.deconv2.w"); N->dc2_b = d.get("det.head.deconv2.b").d[0];
    const Blob& r = N->br;
    N->r_stem = pack_conv(r, "rec.stem");
    for (const Spec& s : REC_SPEC) N->r_blk.push_back(build_lc(r, std::string("rec.") + s.name, s));
    N->r_c1 = pack_conv(r, "rec.neck.conv1"); N->r_c2 = pack_conv(r, "rec.neck.conv2"); N->r_c3 = pack_conv(r, "rec.neck.conv3");
    N->r_c4 = pack_conv(r, "rec.neck.conv4"); N->r_c11 = pack_conv(r, "rec.neck.conv1x1"); N->r_fc = pack_linear(r, "rec.head.fc");
    for (int i = 0; i < 2; i++) {
      const std::string p = "rec.neck.blk" + std::to_string(i);
      N->mix[i].qkv = pack_linear(r, p + ".qkv"); N->mix[i].proj = pack_linear(r, p + ".proj");
      N->mix[i].fc1 = pack_linear(r, p + ".fc1"); N->mix[i].fc2 = pack_linear(r, p + ".fc2");
      N->mix[i].n1g = vec(r, p + ".norm1.g"); N->mix[i].n1b = vec(r, p + ".norm1.beta");
      N->mix[i].n2g = vec(r, p + ".norm2.g"); N->mix[i].n2b = vec(r, p + ".norm2.beta");
    }
    N->ng = vec(r, "rec.neck.norm.g"); N->nb = vec(r, "rec.neck.norm.beta");
    const Blob& c = N->bc;
    N->c_stem = pack_conv(c, "cls.stem");
    struct CS { int k, mid, cout; bool se; int act, sh, sw; };
    static const CS CLS[] = {{3, 8, 8, true, ACT_RELU, 2, 1},      {3, 24, 8, false, ACT_RELU, 2, 1},    {3, 32, 8, false, ACT_RELU, 1, 1},
                             {5, 32, 16, true, ACT_HSWISH, 2, 1},  {5, 88, 16, true, ACT_HSWISH, 1, 1},  {5, 88, 16, true, ACT_HSWISH, 1, 1},
                             {5, 40, 16, true, ACT_HSWISH, 1, 1},  {5, 48, 16, true, ACT_HSWISH, 1, 1},  {5, 104, 32, true, ACT_HSWISH, 2, 1},
                             {5, 200, 32, true, ACT_HSWISH, 1, 1}, {5, 200, 32, true, ACT_HSWISH, 1, 1}};
    int cin = 8, i = 0;
    for (const CS& s : CLS) {
      const std::string p = "cls.b" + std::to_string(i++);
      Nets::CB b; b.expand = pack_conv(c, p + ".expand"); b.dw = pack_dw(c, p + ".dw"); b.se = s.se;
      if (s.se) b.sew = pack_se(c, p + ".se");
      b.linear = pack_conv(c, p + ".linear"); b.act = s.act; b.sh = s.sh; b.sw = s.sw; b.shortcut = (s.sh == 1 && s.sw == 1 && cin == s.cout);
      N->c_blk.push_back(b); cin = s.cout;
    }
    N->c_conv2 = pack_conv(c, "cls.conv2"); N->c_fc = pack_linear(c, "cls.head.fc");
    return N;
  } catch (const std::exception&) { return nullptr; }
}
OCPU_API void ocpu_destroy(void* h) { delete (Nets*)h; }
OCPU_API void ocpu_set_threads(int n) { omp_set_num_threads(n > 0 ? n : 1); }
OCPU_API int ocpu_max_threads() { return omp_get_max_threads(); }

// x [n,3,h,w] (BGR, normalised) -> [n,1,h,w]
OCPU_API int ocpu_det(void* hd, const float* x_nchw, int n, int h, int w, float* out) {
  Nets& N = *(Nets*)hd;
  T4 x = from_nchw(x_nchw, n, 3, h, w);
  Lab nolab;
  T4 t = conv(x, N.d_stem, 2, 2, 1, 1, ACT_NONE);
  T4 taps[4];
  const int tap_after[4] = {2, 4, 9, 13};
  for (size_t i = 0; i < N.d_blk.size(); i++) {
    t = run_lc(N.d_blk[i], t);
    for (int j = 0; j < 4; j++) if (tap_after[j] == (int)i) taps[j] = conv(t, N.d_out[j], 1, 1, 0, 0, ACT_NONE);
  }
  T4 in[4];
  for (int j = 3; j >= 0; j--) {
    in[j] = conv(taps[j], N.d_ins[j], 1, 1, 0, 0, ACT_NONE);
    se_apply(in[j], N.d_ins_se[j], 0.2f, true);
    if (j < 3) { T4 u = up_nearest(in[j + 1], 2); add_inplace(in[j], u); }
  }
  T4 fuse(n, in[0].h, in[0].w, 96);
  for (int j = 3; j >= 0; j--) {
    T4 p = conv(in[j], N.d_inp[j], 1, 1, 1, 1, ACT_NONE);
    se_apply(p, N.d_inp_se[j], 0.2f, true);
    T4 u = j ? up_nearest(p, 1 << j) : p;
    for (int i = 0; i < n; i++) for (int y = 0; y < fuse.h; y++) for (int xx = 0; xx < fuse.w; xx++) memcpy(fuse.at(i, y, xx) + (3 - j) * 24, u.at(i, y, xx), 24 * sizeof(float));
  }
  T4 h1 = conv(fuse, N.d_head, 1, 1, 1, 1, ACT_RELU);
  // deconv1 (24->24, 2x2 s2) + relu, deconv2 (24->1) + sigmoid
  const int H4 = h1.h, W4 = h1.w;
#pragma omp parallel for
  for (int r = 0; r < n * H4; r++) {
    const int i = r / H4, y = r % H4;
    for (int xx = 0; xx < W4; xx++) {
      const float* s = h1.at(i, y, xx);
      for (int q = 0; q < 4; q++) {
        float f[24];
        for (int co = 0; co < 24; co++) {
          float v = N.dc1_b[co];
          for (int ci = 0; ci < 24; ci++) v += s[ci] * N.dc1_w[((size_t)ci * 24 + co) * 4 + q];
          f[co] = v > 0.f ? v : 0.f;
        }
        for (int q2 = 0; q2 < 4; q2++) {
          float v = N.dc2_b;
          for (int ci = 0; ci < 24; ci++) v += f[ci] * N.dc2_w[(size_t)ci * 4 + q2];
          const int oy = 4 * y + 2 * (q >> 1) + (q2 >> 1), ox = 4 * xx + 2 * (q & 1) + (q2 & 1);
          out[((size_t)i * h + oy) * w + ox] = 1.f / (1.f + expf(-v));
        }
      }
    }
  }
  (void)nolab;
  return 0;
}

OCPU_API int ocpu_cls(void* hd, const float* x_nchw, int n, float* out) {
  Nets& N = *(Nets*)hd;
  T4 x = from_nchw(x_nchw, n, 3, 48, 192);
  T4 t = conv(x, N.c_stem, 2, 2, 1, 1, ACT_HSWISH);
  Lab nolab;
  for (const Nets::CB& b : N.c_blk) {
    T4 e = conv(t, b.expand, 1, 1, 0, 0, b.act);
    T4 d = dwconv(e, b.dw, b.sh, b.sw, b.act, nolab);
    if (b.se) se_apply(d, b.sew, 0.2f, false);
    t = conv(d, b.linear, 1, 1, 0, 0, ACT_NONE, nullptr, b.shortcut ? &t : nullptr);
  }
  T4 f = conv(t, N.c_conv2, 1, 1, 0, 0, ACT_HSWISH);
  for (int i = 0; i < n; i++) {
    std::vector<double> m(200, 0.0);
    const int ph = f.h / 2, pw = f.w / 2;
    for (int y = 0; y < ph; y++) for (int xx = 0; xx < pw; xx++)
      for (int c = 0; c < 200; c++) {
        float v = std::max(std::max(f.at(i, 2 * y, 2 * xx)[c], f.at(i, 2 * y, 2 * xx + 1)[c]), std::max(f.at(i, 2 * y + 1, 2 * xx)[c], f.at(i, 2 * y + 1, 2 * xx + 1)[c]));
        m[c] += v;
      }
    float l[2];
    for (int o = 0; o < 2; o++) { float v = N.c_fc.b[o]; for (int c = 0; c < 200; c++) v += (float)(m[c] / (ph * pw)) * N.c_fc.w[(size_t)c * 2 + o]; l[o] = v; }
    const float mx = std::max(l[0], l[1]), e0 = expf(l[0] - mx), e1 = expf(l[1] - mx);
    out[2 * i] = e0 / (e0 + e1); out[2 * i + 1] = e1 / (e0 + e1);
  }
  return 0;
}

// x [n,3,48,w] -> softmax probs [n,T,classes]; returns T
OCPU_API int ocpu_rec(void* hd, const float* x_nchw, int n, int w, float* out) {
  Nets& N = *(Nets*)hd;
  T4 x = from_nchw(x_nchw, n, 3, 48, w);
  T4 t = conv(x, N.r_stem, 2, 2, 1, 1, ACT_NONE);
  for (const LcBlk& b : N.r_blk) t = run_lc(b, t);
  const int T = (t.w - 2) / 2 + 1, C = 480, D = 120, classes = N.r_fc.cout;
  if (!out) return T;
  T4 hpool(n, 1, T, C);
  for (int i = 0; i < n; i++) for (int xx = 0; xx < T; xx++) for (int c = 0; c < C; c++) {
    float s = 0.f; for (int dy = 0; dy < 3; dy++) for (int dx = 0; dx < 2; dx++) s += t.at(i, dy, 2 * xx + dx)[c];
    hpool.at(i, 0, xx)[c] = s / 6.f;
  }
  T4 z = conv(conv(hpool, N.r_c1, 1, 1, 0, 1, ACT_SWISH), N.r_c2, 1, 1, 0, 0, ACT_SWISH);  // [n,1,T,120]
  auto linear = [&](const float* in, int rows, const ConvW& L, int act) {
    std::vector<float> o((size_t)rows * L.cout);
#pragma omp parallel for
    for (int r = 0; r < rows; r++) {
      float* y = &o[(size_t)r * L.cout];
      memcpy(y, L.b.data(), L.cout * sizeof(float));
      for (int k = 0; k < L.cin; k++) { const float a = in[(size_t)r * L.cin + k]; const float* wr = &L.w[(size_t)k * L.cout];
#pragma omp simd
        for (int c = 0; c < L.cout; c++) y[c] += a * wr[c]; }
      for (int c = 0; c < L.cout; c++) y[c] = actf(y[c], act);
    }
    return o;
  };
  const int rows = n * T;
  std::vector<float> zz(z.d);
  for (int bi = 0; bi < 2; bi++) {
    const Nets::Mix& M = N.mix[bi];
    std::vector<float> qkv = linear(zz.data(), rows, M.qkv, ACT_NONE), att((size_t)rows * D);
    const int nh = 8, hd_ = D / 8; const float sc = 1.f / sqrtf((float)hd_);
#pragma omp parallel for collapse(2)
    for (int i = 0; i < n; i++) for (int hh = 0; hh < nh; hh++) {
      std::vector<float> p(T);
      for (int a = 0; a < T; a++) {
        const float* q = &qkv[((size_t)(i * T + a)) * 3 * D + hh * hd_];
        float mx = -1e30f;
        for (int b = 0; b < T; b++) { const float* k = &qkv[((size_t)(i * T + b)) * 3 * D + D + hh * hd_]; float s = 0.f; for (int e = 0; e < hd_; e++) s += q[e] * sc * k[e]; p[b] = s; mx = std::max(mx, s); }
        float sum = 0.f; for (int b = 0; b < T; b++) { p[b] = expf(p[b] - mx); sum += p[b]; }
        float* o = &att[((size_t)(i * T + a)) * D + hh * hd_];
        for (int e = 0; e < hd_; e++) o[e] = 0.f;
        for (int b = 0; b < T; b++) { const float* v = &qkv[((size_t)(i * T + b)) * 3 * D + 2 * D + hh * hd_]; const float pw = p[b] / sum; for (int e = 0; e < hd_; e++) o[e] += pw * v[e]; }
      }
    }
    std::vector<float> pr = linear(att.data(), rows, M.proj, ACT_NONE);
    for (size_t i = 0; i < zz.size(); i++) zz[i] += pr[i];
    layer_norm(zz.data(), rows, D, M.n1g, M.n1b, 1e-5f);
    std::vector<float> m1 = linear(zz.data(), rows, M.fc1, ACT_SWISH), m2 = linear(m1.data(), rows, M.fc2, ACT_NONE);
    for (size_t i = 0; i < zz.size(); i++) zz[i] += m2[i];
    layer_norm(zz.data(), rows, D, M.n2g, M.n2b, 1e-5f);
  }
  layer_norm(zz.data(), rows, D, N.ng, N.nb, 1e-6f);
  T4 zt(n, 1, T, D); zt.d = zz;
  T4 z3 = conv(zt, N.r_c3, 1, 1, 0, 0, ACT_SWISH);
  T4 cat(n, 1, T, 2 * C);
  for (int i = 0; i < n; i++) for (int xx = 0; xx < T; xx++) { memcpy(cat.at(i, 0, xx), hpool.at(i, 0, xx), C * sizeof(float)); memcpy(cat.at(i, 0, xx) + C, z3.at(i, 0, xx), C * sizeof(float)); }
  T4 z5 = conv(conv(cat, N.r_c4, 1, 1, 0, 1, ACT_SWISH), N.r_c11, 1, 1, 0, 0, ACT_SWISH);
  std::vector<float> lg = linear(z5.d.data(), rows, N.r_fc, ACT_NONE);
#pragma omp parallel for
  for (int r = 0; r < rows; r++) {
    float* l = &lg[(size_t)r * classes]; float mx = l[0];
    for (int c = 1; c < classes; c++) mx = std::max(mx, l[c]);
    float s = 0.f; for (int c = 0; c < classes; c++) { l[c] = expf(l[c] - mx); s += l[c]; }
    for (int c = 0; c < classes; c++) out[(size_t)r * classes + c] = l[c] / s;
  }
  return T;
}
