#include "nets.h"
#include "onnx_import.h"

#include <algorithm>
#include <cstdlib>
#include <cstring>

namespace rt {

// ---------------------------------------------------------------------------
// levels
// ---------------------------------------------------------------------------
static void finish_level(Level& L) {
  long long off = 0;
  L.maxH = L.maxW = 0; L.maxPix = 0;
  for (auto& g : L.h) {
    g.off = off; g.pad_ = 0;
    off += (long long)g.H * g.W;
    L.maxH = std::max(L.maxH, g.H); L.maxW = std::max(L.maxW, g.W);
    L.maxPix = std::max(L.maxPix, (long long)g.H * g.W);
  }
  L.total = off;
}
Level make_level(const std::vector<std::pair<int, int>>& hw) {
  Level L;
  for (auto& p : hw) L.h.push_back(ImgGeom{0, p.first, p.second, 0});
  finish_level(L);
  return L;
}
Level down_level(const Level& in, int sh, int sw) {
  Level L;
  for (auto& g : in.h) L.h.push_back(ImgGeom{0, (g.H - 1) / sh + 1, (g.W - 1) / sw + 1, 0});
  finish_level(L);
  return L;
}
Level flat_level(const Level& in) {
  if (in.total > 0x7fffffffLL) throw RtError(3, "level too large for a flat view");
  return make_level({{1, (int)in.total}});
}
Level pool_level(const Level& in, int kh, int kw) {
  Level L;
  for (auto& g : in.h) L.h.push_back(ImgGeom{0, g.H >= kh ? (g.H - kh) / kh + 1 : 0, g.W >= kw ? (g.W - kw) / kw + 1 : 0, 0});
  finish_level(L);
  return L;
}
void upload_levels(RunCtx& c, std::vector<Level*> levels) {
  size_t total = 0;
  for (auto* L : levels) total += L->h.size();
  if (total == 0) return;
  ImgGeom* hbuf = c.pinned->alloc<ImgGeom>(total);
  ImgGeom* dbuf = c.arena->alloc<ImgGeom>(total);
  size_t o = 0;
  for (auto* L : levels) {
    memcpy(hbuf + o, L->h.data(), L->h.size() * sizeof(ImgGeom));
    L->d = dbuf + o;
    o += L->h.size();
  }
  RT_HIP_CHECK(hipMemcpyAsync(dbuf, hbuf, total * sizeof(ImgGeom), hipMemcpyHostToDevice, c.st));
}

// ---------------------------------------------------------------------------
// weights
// ---------------------------------------------------------------------------
WeightStore::~WeightStore() {
  // (the split-bf16 planes nn::gemm_split derived from a pack go with it: a later allocation may reuse the address)
  for (void* p : bufs_) { nn::gemm_split_forget((const float*)p); (void)hipFree(p); }
}
void* WeightStore::upload_bytes(const void* host, size_t bytes) {
  void* p = nullptr;
  size_t cap = std::max<size_t>(bytes, 16);
  RT_HIP_CHECK(hipMalloc(&p, cap));
  RT_HIP_CHECK(hipMemcpy(p, host, bytes, hipMemcpyHostToDevice));
  bufs_.push_back(p); total_ += cap;
  return p;
}
float* WeightStore::upload(const std::vector<float>& host) {
  void* p = nullptr;
  size_t bytes = std::max<size_t>(host.size(), 4) * sizeof(float);
  RT_HIP_CHECK(hipMalloc(&p, bytes));
  RT_HIP_CHECK(hipMemcpy(p, host.data(), host.size() * sizeof(float), hipMemcpyHostToDevice));
  bufs_.push_back(p); total_ += bytes;
  return (float*)p;
}

static void expect_dims(const BlobTensor& t, std::initializer_list<int> d, const std::string& name) {
  if (t.dims != std::vector<int>(d)) throw RtError(3, "RTWB: unexpected shape for " + name);
}

// conv weight [cout, cin, kh, kw] -> [nkc][taps][Npad][KC]
PackedDense pack_conv(WeightStore& ws, const Blob& b, const std::string& name, int cout, int cin, int kh, int kw) {
  const BlobTensor& w = b.get(name + ".w");
  expect_dims(w, {cout, cin, kh, kw}, name + ".w");
  PackedDense p;
  p.K = round_up(cin, 4); p.N = cout; p.Npad = round_up(cout, 16); p.kh = kh; p.kw = kw;
  const int taps = kh * kw, nkc = (p.K + nn::KC - 1) / nn::KC;
  std::vector<float> host((size_t)nkc * taps * p.Npad * nn::KC, 0.f);
  for (int n = 0; n < cout; n++)
    for (int k = 0; k < cin; k++)
      for (int t = 0; t < taps; t++) {
        int kc = k / nn::KC, kk = k % nn::KC;
        host[(((size_t)kc * taps + t) * p.Npad + n) * nn::KC + kk] = w.data[((size_t)n * cin + k) * taps + t];
      }
  p.w = ws.upload(host);
  std::vector<float> bias(p.Npad, 0.f);
  if (b.has(name + ".b")) {
    const BlobTensor& bt = b.get(name + ".b");
    expect_dims(bt, {cout}, name + ".b");
    memcpy(bias.data(), bt.data, cout * sizeof(float));
  }
  p.b = ws.upload(bias);
  return p;
}
// linear weight [in, out]
PackedDense pack_linear(WeightStore& ws, const Blob& b, const std::string& name, int cin, int cout) {
  const BlobTensor& w = b.get(name + ".w");
  expect_dims(w, {cin, cout}, name + ".w");
  PackedDense p;
  p.K = round_up(cin, 4); p.N = cout; p.Npad = round_up(cout, 16);
  const int nkc = (p.K + nn::KC - 1) / nn::KC;
  std::vector<float> host((size_t)nkc * p.Npad * nn::KC, 0.f);
  for (int k = 0; k < cin; k++)
    for (int n = 0; n < cout; n++)
      host[((size_t)(k / nn::KC) * p.Npad + n) * nn::KC + (k % nn::KC)] = w.data[(size_t)k * cout + n];
  p.w = ws.upload(host);
  std::vector<float> bias(p.Npad, 0.f);
  const BlobTensor& bt = b.get(name + ".b");
  expect_dims(bt, {cout}, name + ".b");
  memcpy(bias.data(), bt.data, cout * sizeof(float));
  p.b = ws.upload(bias);
  return p;
}
static PackedDw pack_dw(WeightStore& ws, const Blob& b, const std::string& name, int C, int k) {
  const BlobTensor& w = b.get(name + ".w");
  expect_dims(w, {C, 1, k, k}, name + ".w");
  PackedDw p; p.k = k; p.C = C; p.Cp = chan_pitch(C);
  std::vector<float> host((size_t)k * k * p.Cp, 0.f), bias(p.Cp, 0.f);
  for (int c = 0; c < C; c++)
    for (int t = 0; t < k * k; t++) host[(size_t)t * p.Cp + c] = w.data[(size_t)c * k * k + t];
  const BlobTensor& bt = b.get(name + ".b");
  expect_dims(bt, {C}, name + ".b");
  memcpy(bias.data(), bt.data, C * sizeof(float));
  p.w = ws.upload(host); p.b = ws.upload(bias);
  return p;
}
static void pack_stem(WeightStore& ws, const Blob& b, const std::string& name, int cout, float** w_out, float** b_out) {
  const BlobTensor& w = b.get(name + ".w");
  expect_dims(w, {cout, 3, 3, 3}, name + ".w");
  std::vector<float> host(27 * cout), bias(cout);
  for (int n = 0; n < cout; n++)
    for (int ci = 0; ci < 3; ci++)
      for (int t = 0; t < 9; t++) host[(t * 3 + ci) * cout + n] = w.data[(n * 3 + ci) * 9 + t];
  const BlobTensor& bt = b.get(name + ".b");
  expect_dims(bt, {cout}, name + ".b");
  memcpy(bias.data(), bt.data, cout * sizeof(float));
  *w_out = ws.upload(host); *b_out = ws.upload(bias);
}
static Lab get_lab(const Blob& b, const std::string& name) {
  Lab l;
  if (b.has(name + ".a")) { l.has = 1; l.a = b.get(name + ".a").data[0]; l.c = b.get(name + ".c").data[0]; }
  return l;
}
float* upload_raw(WeightStore& ws, const Blob& b, const std::string& name, size_t expect_numel) {
  const BlobTensor& t = b.get(name);
  if (t.numel() != expect_numel) throw RtError(3, "RTWB: unexpected size for " + name);
  return ws.upload(std::vector<float>(t.data, t.data + t.numel()));
}
static SeW get_se(WeightStore& ws, const Blob& b, const std::string& name, int C) {
  SeW s; s.C = C; s.Cr = C / 4;
  s.w1 = upload_raw(ws, b, name + ".fc1.w", (size_t)s.Cr * C); s.b1 = upload_raw(ws, b, name + ".fc1.b", s.Cr);
  s.w2 = upload_raw(ws, b, name + ".fc2.w", (size_t)s.Cr * C); s.b2 = upload_raw(ws, b, name + ".fc2.b", C);
  return s;
}

struct LcSpec { const char* name; int k, cin, cout, sh, sw; bool se; };
static const LcSpec DET_SPEC[] = {
    {"s2.0", 3, 16, 32, 1, 1, false}, {"s3.0", 3, 32, 48, 2, 2, false}, {"s3.1", 3, 48, 48, 1, 1, false},
    {"s4.0", 3, 48, 96, 2, 2, false}, {"s4.1", 3, 96, 96, 1, 1, false}, {"s5.0", 3, 96, 192, 2, 2, false},
    {"s5.1", 5, 192, 192, 1, 1, false}, {"s5.2", 5, 192, 192, 1, 1, false}, {"s5.3", 5, 192, 192, 1, 1, false},
    {"s5.4", 5, 192, 192, 1, 1, false}, {"s6.0", 5, 192, 384, 2, 2, true}, {"s6.1", 5, 384, 384, 1, 1, true},
    {"s6.2", 5, 384, 384, 1, 1, false}, {"s6.3", 5, 384, 384, 1, 1, false}};
static const LcSpec REC_SPEC[] = {
    {"s2.0", 3, 16, 32, 1, 1, false}, {"s3.0", 3, 32, 64, 1, 1, false}, {"s3.1", 3, 64, 64, 1, 1, false},
    {"s4.0", 3, 64, 128, 2, 1, false}, {"s4.1", 3, 128, 128, 1, 1, false}, {"s5.0", 3, 128, 240, 1, 2, false},
    {"s5.1", 5, 240, 240, 1, 1, false}, {"s5.2", 5, 240, 240, 1, 1, false}, {"s5.3", 5, 240, 240, 1, 1, false},
    {"s5.4", 5, 240, 240, 1, 1, false}, {"s6.0", 5, 240, 480, 2, 1, true}, {"s6.1", 5, 480, 480, 1, 1, true},
    {"s6.2", 5, 480, 480, 2, 1, false}, {"s6.3", 5, 480, 480, 1, 1, false}};

static LcBlock build_lc(WeightStore& ws, const Blob& b, const std::string& prefix, const LcSpec& s) {
  LcBlock blk;
  std::string p = prefix + "." + s.name;
  blk.dw = pack_dw(ws, b, p + ".dw", s.cin, s.k);
  blk.dw_lab = get_lab(b, p + ".dw");
  blk.dw_act = blk.dw_lab.has ? ACT_HSWISH : ACT_NONE;  // LearnableRepLayer: act only when stride != 2
  blk.se = s.se;
  if (s.se) blk.sew = get_se(ws, b, p + ".se", s.cin);
  blk.pw = pack_conv(ws, b, p + ".pw", s.cout, s.cin, 1, 1);
  blk.pw_lab = get_lab(b, p + ".pw");
  blk.sh = s.sh; blk.sw = s.sw; blk.cin = s.cin; blk.cout = s.cout;
  return blk;
}

Epilogue make_epi(const PackedDense& p, int act, const Lab* lab, const float* residual, int ld_res) {
  Epilogue e;
  e.bias = p.b; e.act = act;
  e.has_lab = lab ? lab->has : 0; e.lab_a = lab ? lab->a : 1.f; e.lab_c = lab ? lab->c : 0.f;
  e.residual = residual; e.ld_res = ld_res;
  return e;
}

// Squeeze-excite scales of x ([image][Cp]); with apply the tensor is rescaled in place, otherwise the
// caller folds the returned factors into the kernel that consumes x.
static float* run_se(RunCtx& c, float* x, const Level& L, const SeW& se, float slope, int residual, bool apply = true) {
  int Cp = chan_pitch(se.C);
  float* partial = c.arena->alloc<float>((size_t)L.n() * nn::pool_chunks(L.maxPix) * Cp);
  float* scale = c.arena->alloc<float>((size_t)L.n() * Cp);
  { ProfScope ps(c.prof, c.st, "se_pool_fc");
    nn::se_scale(c.st, x, L.d, L.n(), L.maxPix, se.C, Cp, se.w1, se.b1, se.w2, se.b2, se.Cr, slope, residual, partial, scale); }
  if (apply) {
    ProfScope ps(c.prof, c.st, "scale_channels");
    nn::scale_channels(c.st, x, L.d, L.n(), L.maxPix, Cp, scale);
  }
  return scale;
}

static std::string shape_str(long long a, long long b, long long c, long long d) {
  return std::to_string(a) + "," + std::to_string(b) + "," + std::to_string(c) + "," + std::to_string(d);
}

static const float HSIG_LCNET = 0.1666667f;  // paddle nn.Hardsigmoid
static const float HSIG_MBV3 = 0.2f;         // F.hardsigmoid(slope=0.2, offset=0.5)

static float* run_lc(RunCtx& c, const LcBlock& b, const float* x, const Level& Lin, const Level& Lout) {
  if (!b.se && b.pw.K == b.dw.Cp &&
      nn::lc_block_supported(b.dw.k, b.sh, b.sw, b.dw.Cp, b.dw.C, b.pw.N, b.pw.Npad, b.dw_act, b.dw_lab.has, make_epi(b.pw, ACT_HSWISH, &b.pw_lab),
                             Lout.maxH, Lout.maxW)) {
    int Cpo = chan_pitch(b.cout);
    float* y2 = c.arena->alloc<float>((size_t)Lout.total * Cpo);
    ProfScope ps(c.prof, c.st, "lc_thin", shape_str(Lout.total, b.dw.Cp, b.pw.N, b.sh * 10 + b.sw));
    nn::lc_thin(c.st, b.sh, b.sw, x, Lin.d, Lout.d, Lout.n(), Lout.maxH, Lout.maxW, b.dw.Cp, b.dw.C, b.dw.w, b.dw.b, b.dw_act,
                b.dw_lab.has, b.dw_lab.a, b.dw_lab.c, b.pw.w, b.pw.N, b.pw.Npad, y2, Cpo, make_epi(b.pw, ACT_HSWISH, &b.pw_lab));
    return y2;
  }
  float* y1 = c.arena->alloc<float>((size_t)Lout.total * b.dw.Cp);
  // Squeeze-excite without extra passes over y1: the depthwise kernel leaves per-block channel sums,
  // the FC turns them into scales, and the pointwise GEMM multiplies them in while staging its A
  // rows (a row tile spans at most two images when every image has >= tile rows).
  long long min_pix = Lout.maxPix;
  for (const ImgGeom& g : Lout.h) min_pix = std::min<long long>(min_pix, (long long)g.H * g.W);
  // 256: k_gemm32p (3-int table entries), 128: register-staged wide tiles (2-int entries), 0: no fused form
  const int tile_rows = b.se ? nn::gemm_se_tile_rows(b.dw.Cp, Lout.total, b.pw.K, b.pw.N, b.pw.Npad, ACT_HSWISH, min_pix) : 0;
  static const bool no_se_fusion = getenv("RT_NO_SE_FUSION") != nullptr;  // A/B switch
  const bool fuse_se = b.se && tile_rows > 0 && b.pw.K <= 512 && (b.dw.k == 3 || b.dw.k == 5) && !no_se_fusion;
  float* pool = nullptr; int chunks = 0, strip_R = 0, strips_pb = 32;
  if (fuse_se) {
    nn::dwconv_pool_layout(b.dw.k, b.sh, b.sw, b.dw.Cp, Lout.maxH, Lout.maxW, &chunks, &strip_R, &strips_pb);
    pool = c.arena->alloc<float>((size_t)Lout.n() * chunks * b.dw.Cp);
  }
  { ProfScope ps(c.prof, c.st, b.dw.k == 3 ? "dwconv3" : "dwconv5", shape_str(Lin.total, Lout.total, b.dw.Cp, b.sh * 10 + b.sw));
    nn::dwconv(c.st, b.dw.k, b.sh, b.sw, x, Lin.d, Lout.d, Lout.n(), Lout.maxH, Lout.maxW, b.dw.Cp, b.dw.C, b.dw.w, b.dw.b,
               b.dw_act, b.dw_lab.has, b.dw_lab.a, b.dw_lab.c, y1, pool); }
  Epilogue epi = make_epi(b.pw, ACT_HSWISH, &b.pw_lab);
  if (fuse_se) {
    float* scale = c.arena->alloc<float>((size_t)Lout.n() * b.dw.Cp);
    { ProfScope ps(c.prof, c.st, "se_pool_fc");
      nn::se_fc_from_dw(c.st, pool, Lout.d, Lout.n(), chunks, strip_R, strips_pb, b.sew.C, b.dw.Cp, b.sew.w1, b.sew.b1, b.sew.w2,
                        b.sew.b2, b.sew.Cr, HSIG_LCNET, 0, scale); }
    const int*& dtab = Lout.a_tabs[tile_rows];
    const int stride = tile_rows == 256 ? 3 : 2;   // per row block: image of its first row, first row of the next image (, of the one after)
    if (!dtab) {
      const long long tiles = (Lout.total + tile_rows - 1) / tile_rows;
      int* htab = c.pinned->alloc<int>((size_t)tiles * stride);
      int* dt = c.arena->alloc<int>((size_t)tiles * stride);
      size_t img = 0;
      for (long long t = 0; t < tiles; t++) {
        const long long m0 = t * tile_rows;
        while (img + 1 < Lout.h.size() && Lout.h[img + 1].off <= m0) img++;
        htab[stride * t] = (int)img;
        htab[stride * t + 1] = img + 1 < Lout.h.size() ? (int)Lout.h[img + 1].off : 0x7fffffff;  // no next image: never crossed
        if (stride == 3) htab[stride * t + 2] = img + 2 < Lout.h.size() ? (int)Lout.h[img + 2].off : 0x7fffffff;
      }
      RT_HIP_CHECK(hipMemcpyAsync(dt, htab, (size_t)tiles * stride * sizeof(int), hipMemcpyHostToDevice, c.st));
      dtab = dt;
    }
    epi.a_scale = scale; epi.ld_scale = b.dw.Cp; epi.a_tab = dtab; epi.a_tab_stride = stride; epi.n_img = Lout.n();
  } else if (b.se) {
    run_se(c, y1, Lout, b.sew, HSIG_LCNET, 0);
  }
  int Cpo = chan_pitch(b.cout);
  float* y2 = c.arena->alloc<float>((size_t)Lout.total * Cpo);
  { ProfScope ps(c.prof, c.st, nn::gemm_pw_label(Lout.total, b.pw.Npad, fuse_se, tile_rows), shape_str(Lout.total, b.pw.K, b.pw.N, 0));
    nn::gemm(c.st, y1, b.dw.Cp, Lout.total, b.pw.K, b.pw.w, b.pw.N, b.pw.Npad, y2, Cpo, 0, epi); }
  return y2;
}

// ---------------------------------------------------------------------------
// Weights of the upsampling-aware FPN convs (nn_fpn.hip).  w = raw conv weight [24][cin_total][3][3].
// A 3x3 conv over up_s(z): output row y = s Y + py reads row Y - 1 of z through tap dy = 0 only when py == 0, row Y + 1
// through tap dy = 2 only when py == s - 1, and row Y otherwise.
// ---------------------------------------------------------------------------
// [9 taps][24 n][cf] (cf >= cc: zero padded): the channels [c0, c0 + cc) that are convolved at their own resolution
static std::vector<float> fpn_fine_weights(const float* w, int cin_total, int c0, int cc, int cf) {
  std::vector<float> o((size_t)9 * 24 * cf, 0.f);
  for (int t = 0; t < 9; t++)
    for (int n = 0; n < 24; n++)
      for (int c = 0; c < cc; c++) o[((size_t)t * 24 + n) * cf + c] = w[((size_t)n * cin_total + c0 + c) * 9 + t];
  return o;
}
// taps of the 3-tap axis that land on the 2-tap phase axis: phase 0 = {row - 1: tap 0; row 0: taps 1, 2}, phase 1 = {row 0: taps 0, 1; row + 1: tap 2}
static void phase_taps(int ph, int t, int* lo, int* hi) {
  if (ph == 0) { *lo = t == 0 ? 0 : 1; *hi = t == 0 ? 0 : 2; }
  else { *lo = t == 0 ? 0 : 2; *hi = t == 0 ? 1 : 2; }
}
// [cc / 24 slabs][4 phases (py, px)][4 taps (ty, tx)][24 n][24 k]: conv3x3(up2(z)) as four 2 x 2 convs of z
static std::vector<float> fpn_phase_weights(const float* w, int cin_total, int c0, int cc) {
  const int ns = cc / 24;
  std::vector<float> o((size_t)ns * 16 * 576, 0.f);
  for (int s = 0; s < ns; s++)
    for (int py = 0; py < 2; py++)
      for (int px = 0; px < 2; px++)
        for (int ty = 0; ty < 2; ty++)
          for (int tx = 0; tx < 2; tx++) {
            int y0, y1, x0, x1;
            phase_taps(py, ty, &y0, &y1); phase_taps(px, tx, &x0, &x1);
            for (int n = 0; n < 24; n++)
              for (int k = 0; k < 24; k++) {
                double acc = 0.0;
                for (int dy = y0; dy <= y1; dy++)
                  for (int dx = x0; dx <= x1; dx++) acc += w[((size_t)n * cin_total + c0 + s * 24 + k) * 9 + dy * 3 + dx];
                o[((((size_t)s * 4 + py * 2 + px) * 4 + ty * 2 + tx) * 24 + n) * 24 + k] = (float)acc;
              }
          }
  return o;
}
// [9 classes (row class x 3 + column class)][9 taps (ry + 1, rx + 1)][24 n][24 k] for the 24 channels at c0: class 0 = first row /
// column of an upsampling block, 1 = interior, 2 = last
static std::vector<float> fpn_class_weights(const float* w, int cin_total, int c0) {
  auto span = [](int cls, int r, int* lo, int* hi) {   // taps of the 3-tap axis that land on relative row r (-1, 0, +1); empty: lo > hi
    *lo = 1; *hi = 0;
    if (cls == 0) { if (r == -1) { *lo = 0; *hi = 0; } else if (r == 0) { *lo = 1; *hi = 2; } }
    else if (cls == 1) { if (r == 0) { *lo = 0; *hi = 2; } }
    else { if (r == 0) { *lo = 0; *hi = 1; } else if (r == 1) { *lo = 2; *hi = 2; } }
  };
  std::vector<float> o((size_t)81 * 576, 0.f);
  for (int rc = 0; rc < 3; rc++)
    for (int cc = 0; cc < 3; cc++)
      for (int ry = -1; ry <= 1; ry++)
        for (int rx = -1; rx <= 1; rx++) {
          int y0, y1, x0, x1;
          span(rc, ry, &y0, &y1); span(cc, rx, &x0, &x1);
          for (int k = 0; k < 24; k++)
            for (int n = 0; n < 24; n++) {
              double acc = 0.0;
              for (int dy = y0; dy <= y1; dy++)
                for (int dx = x0; dx <= x1; dx++) acc += w[((size_t)n * cin_total + c0 + k) * 9 + dy * 3 + dx];
              o[((((size_t)rc * 3 + cc) * 9 + (ry + 1) * 3 + (rx + 1)) * 24 + n) * 24 + k] = (float)acc;
            }
        }
  return o;
}

// ---------------------------------------------------------------------------
// DetNet
// ---------------------------------------------------------------------------
DetNet::DetNet(const Blob& b) {
  pack_stem(ws_, b, "det.stem", 16, &stem_w_, &stem_b_);
  for (const LcSpec& s : DET_SPEC) blocks_.push_back(build_lc(ws_, b, "det", s));
  tap_after_[0] = 2; tap_after_[1] = 4; tap_after_[2] = 9; tap_after_[3] = 13;
  const int tap_c[4] = {48, 96, 192, 384}, out_c[4] = {12, 18, 42, 360};
  for (int j = 0; j < 4; j++) {
    std::string js = std::to_string(j);
    out_[j] = pack_conv(ws_, b, "det.out" + js, out_c[j], tap_c[j], 1, 1);
    ins_[j] = pack_conv(ws_, b, "det.fpn.ins" + js, 96, out_c[j], 1, 1);
    {  // the same weights as [cin][96] (bias-free conv: checked by the manifest)
      const BlobTensor& w = b.get("det.fpn.ins" + js + ".w");
      std::vector<float> lin((size_t)out_c[j] * 96);
      for (int n = 0; n < 96; n++)
        for (int k = 0; k < out_c[j]; k++) lin[(size_t)k * 96 + n] = w.data[(size_t)n * out_c[j] + k];
      ins_lin_[j] = ws_.upload(lin);
      has_bias_[j] = b.has("det.fpn.ins" + js + ".b");
    }
    ins_se_[j] = get_se(ws_, b, "det.fpn.ins" + js + ".se", 96);
    inp_[j] = pack_conv(ws_, b, "det.fpn.inp" + js, 24, 96, 3, 3);
    inp_se_[j] = get_se(ws_, b, "det.fpn.inp" + js + ".se", 24);
  }
  head_conv1_ = pack_conv(ws_, b, "det.head.conv1", 24, 96, 3, 3);
  {  // operands of the upsampling-aware forms (nn_fpn.hip): pre-summed phase / class weights of the head conv and of inp0 / inp1
    const BlobTensor& hw = b.get("det.head.conv1.w");   // [24][96 = p5 | p4 | p3 | p2][3][3]
    head_wf_ = ws_.upload(fpn_fine_weights(hw.data, 96, 72, 24, 24));
    head_wc_ = ws_.upload(fpn_phase_weights(hw.data, 96, 48, 24));
    head_cls4_ = ws_.upload(fpn_class_weights(hw.data, 96, 24));
    head_cls5_ = ws_.upload(fpn_class_weights(hw.data, 96, 0));
    for (int j = 0; j < 2; j++) {
      const BlobTensor& iw = b.get("det.fpn.inp" + std::to_string(j) + ".w");   // [24][96][3][3]
      inp_wc_[j] = ws_.upload(fpn_phase_weights(iw.data, 96, 0, 96));
      std::vector<float> wm((size_t)9 * 24 * 96);
      for (int t = 0; t < 9; t++)
        for (int n = 0; n < 24; n++)
          for (int m = 0; m < 96; m++) wm[((size_t)t * 24 + n) * 96 + m] = iw.data[((size_t)n * 96 + m) * 9 + t];
      inp_wm_[j] = ws_.upload(wm);
    }
  }
  dc1_w_ = upload_raw(ws_, b, "det.head.deconv1.w", 24 * 24 * 4); dc1_b_ = upload_raw(ws_, b, "det.head.deconv1.b", 24);
  dc2_w_ = upload_raw(ws_, b, "det.head.deconv2.w", 24 * 4); dc2_b_ = upload_raw(ws_, b, "det.head.deconv2.b", 1);
}

float* DetNet::run(RunCtx& c, const float* x, Level& L0, const nn::U8Page* pages, float scale, const float* mean3,
                   const float* std3) {
  for (auto& g : L0.h)
    if (g.H % 32 != 0 || g.W % 32 != 0 || g.H == 0 || g.W == 0) throw RtError(3, "det input sides must be non-zero multiples of 32");
  Level L2 = down_level(L0, 2, 2), L4 = down_level(L2, 2, 2), L8 = down_level(L4, 2, 2), L16 = down_level(L8, 2, 2),
        L32 = down_level(L16, 2, 2);
  upload_levels(c, {&L0, &L2, &L4, &L8, &L16, &L32});
  Level* lv[6] = {&L0, &L2, &L4, &L8, &L16, &L32};
  float* t = c.arena->alloc<float>((size_t)L2.total * 16);
  { ProfScope ps(c.prof, c.st, "stem");
    if (pages) nn::stem_conv_u8(c.st, pages, scale, mean3, std3, L0.d, L2.d, L2.n(), L2.maxH, L2.maxW, 16, stem_w_, stem_b_, ACT_NONE, t);
    else nn::stem_conv(c.st, x, L0.d, L2.d, L2.n(), L2.maxH, L2.maxW, 16, stem_w_, stem_b_, ACT_NONE, t); }
  int li = 1;
  float* taps[4] = {nullptr, nullptr, nullptr, nullptr};
  Level* tap_lv[4] = {nullptr, nullptr, nullptr, nullptr};
  for (size_t i = 0; i < blocks_.size(); i++) {
    const LcBlock& b = blocks_[i];
    Level* Lin = lv[li];
    if (b.sh == 2) li++;
    t = run_lc(c, b, t, *Lin, *lv[li]);
    for (int j = 0; j < 4; j++)
      if (tap_after_[j] == (int)i) {
        int Cpo = round_up(out_[j].N, 4);
        float* o = c.arena->alloc<float>((size_t)lv[li]->total * Cpo);
        ProfScope ps(c.prof, c.st, "gemm_misc", shape_str(lv[li]->total, out_[j].K, out_[j].N, 0));
        nn::gemm(c.st, t, chan_pitch(b.cout), lv[li]->total, out_[j].K, out_[j].w, out_[j].N, out_[j].Npad, o, Cpo, 0,
                 make_epi(out_[j], ACT_NONE));
        taps[j] = o; tap_lv[j] = lv[li];
      }
  }
  // RSEFPN.  The squeeze-excite factors of the lateral (ins) and output (inp) convs are not applied in a pass
  // of their own: ins[j]'s go into the top-down add that consumes it (out = ins[j] * s + up(ins[j+1]); the
  // coarsest level, which has no add, is rescaled in place), inp[j]'s into the concat gather.
  // Upsampling-aware forms (nn_fpn.hip): every level exactly half of the finer one (det inputs are multiples of 32), lateral
  // convs without bias.  RT_FPN_PHASE=0 / rt_debug_set_variants bit 11 keep the round-3 launch series.
  bool phase = nn::fpn_phase_supported(out_[0].N, 96) && nn::fpn_phase_supported(out_[1].N, 96) && nn::fpn_phase_supported(24, 24) &&
               !nn::g_fpn_phase_off && !has_bias_[0] && !has_bias_[1];
  for (int j = 0; j < 3 && phase; j++)
    for (size_t i = 0; i < tap_lv[j]->h.size(); i++)
      if (tap_lv[j]->h[i].H != 2 * tap_lv[j + 1]->h[i].H || tap_lv[j]->h[i].W != 2 * tap_lv[j + 1]->h[i].W) phase = false;
  float* in[4] = {nullptr, nullptr, nullptr, nullptr};
  float* lat_scale[4] = {nullptr, nullptr, nullptr, nullptr};
  for (int j = 3; j >= 0; j--) {
    const int cin = out_[j].N, cin_p = round_up(cin, 4);
    if (j == 3 || has_bias_[j]) {  // coarsest level (no add to fold into), or a lateral conv that carries a bias: GEMM + SE in place
      in[j] = c.arena->alloc<float>((size_t)tap_lv[j]->total * 96);
      { ProfScope ps(c.prof, c.st, "gemm_misc", shape_str(tap_lv[j]->total, ins_[j].K, 96, 0));
        nn::gemm(c.st, taps[j], cin_p, tap_lv[j]->total, ins_[j].K, ins_[j].w, 96, ins_[j].Npad, in[j], 96, 0, make_epi(ins_[j], ACT_NONE)); }
      float* sc = run_se(c, in[j], *tap_lv[j], ins_se_[j], HSIG_MBV3, 1, j == 3);
      if (j < 3) {
        ProfScope ps(c.prof, c.st, "upsample_add");
        nn::upsample_add(c.st, in[j], in[j + 1], tap_lv[j]->d, tap_lv[j + 1]->d, tap_lv[j]->n(), tap_lv[j]->maxPix, 96, in[j], sc);
      }
      continue;
    }
    // lateral conv (cin -> 96, no bias) + SE factor + top-down add in one pass over the 96-channel tensor: the squeeze
    // only needs the channel means, and mean(x . W) = mean(x) . W
    const Level& L = *tap_lv[j];
    float* partial = c.arena->alloc<float>((size_t)L.n() * nn::pool_chunks(L.maxPix) * cin_p);
    float* scale = c.arena->alloc<float>((size_t)L.n() * 96);
    lat_scale[j] = scale;
    { ProfScope ps(c.prof, c.st, "se_pool_fc");
      nn::se_scale_projected(c.st, taps[j], L.d, L.n(), L.maxPix, cin, cin_p, ins_lin_[j], 96, 96, ins_se_[j].w1, ins_se_[j].b1,
                             ins_se_[j].w2, ins_se_[j].b2, ins_se_[j].Cr, HSIG_MBV3, 1, partial, scale); }
    if (phase && j == 0) continue;   // the finest lateral tensor is never built: inp0 convolves the tap tensor itself (below)
    in[j] = c.arena->alloc<float>((size_t)L.total * 96);
    { ProfScope ps(c.prof, c.st, "lateral_add", shape_str(L.total, cin, 96, 0));
      nn::lateral_add(c.st, taps[j], cin, cin_p, ins_lin_[j], 96, scale, in[j + 1], L.d, tap_lv[j + 1]->d, L.n(), L.maxPix, in[j]); }
  }
  float* p[4];
  const float* p_scale[4];
  for (int j = 3; j >= 0; j--) {
    const Level& L = *tap_lv[j];
    p[j] = c.arena->alloc<float>((size_t)L.total * 24);
    if (phase && j < 2) {
      // p_j = conv3x3(lateral(c_j) * s + up2(in_{j+1})) = conv3x3(c_j; per-image composed weights) + four 2 x 2 phase convs of in_{j+1}
      const int cin = out_[j].N, cf = round_up(cin, 4);
      float* wimg = c.arena->alloc<float>((size_t)L.n() * 9 * 24 * cf);
      const int tiles = ((L.maxW + 15) / 16) * ((L.maxH + 15) / 16);
      float* pool = c.arena->alloc<float>((size_t)L.n() * tiles * 24);
      float* scale = c.arena->alloc<float>((size_t)L.n() * 24);
      { ProfScope ps(c.prof, c.st, "fpn_compose");
        nn::fpn_compose(c.st, ins_lin_[j], cin, 96, lat_scale[j], inp_wm_[j], cf, L.n(), wimg); }
      { ProfScope ps(c.prof, c.st, "conv3x3_phase", shape_str(L.total, 9 * cf + 4 * 96, 24, 0));
        nn::FpnPhaseArgs a;
        a.fine = taps[j]; a.ld_fine = cf; a.Wf = wimg; a.wf_img = (long long)9 * 24 * cf;
        a.coarse = in[j + 1]; a.ld_coarse = 96; a.Wc = inp_wc_[j];
        a.bias = inp_[j].b; a.y = p[j]; a.ldy = 24; a.pool = pool; a.pool_tiles = tiles; a.act = ACT_NONE;
        nn::fpn_phase(c.st, a, cin, 96, L.d, tap_lv[j + 1]->d, L.n(), L.maxH, L.maxW); }
      { ProfScope ps(c.prof, c.st, "se_pool_fc");
        nn::se_fc_from_tiles(c.st, pool, L.d, L.n(), tiles, 24, 24, inp_se_[j].w1, inp_se_[j].b1, inp_se_[j].w2, inp_se_[j].b2,
                             inp_se_[j].Cr, HSIG_MBV3, 1, scale); }
      p_scale[3 - j] = scale;
      continue;
    }
    { ProfScope ps(c.prof, c.st, "conv3x3", shape_str(L.total, 9 * 96, 24, 0));
      nn::conv_sp(c.st, 3, 3, in[j], 96, L.d, L.n(), L.maxH, L.maxW, 96, inp_[j].w, 24,
                  inp_[j].Npad, p[j], 24, make_epi(inp_[j], ACT_NONE)); }
    p_scale[3 - j] = run_se(c, p[j], L, inp_se_[j], HSIG_MBV3, 1, false);  // order p5, p4, p3, p2
  }
  float* h1 = c.arena->alloc<float>((size_t)L4.total * 24);
  if (phase) {
    // head conv over concat(up8(p5), up4(p4), up2(p3), p2) * scales: p2 at its own resolution, p3 as phase convs, p4 / p5 through
    // their class tensors (bias folded in)
    float* V5 = c.arena->alloc<float>((size_t)L32.total * 9 * 24);
    float* V45 = c.arena->alloc<float>((size_t)L16.total * 9 * 24);
    { ProfScope ps(c.prof, c.st, "fpn_class");
      nn::fpn_class(c.st, p[3], 24, p_scale[0], 24, L32.d, L32.n(), L32.maxPix, head_cls5_, nullptr, nullptr, nullptr, 0, V5, L32.total);
      nn::fpn_class(c.st, p[2], 24, p_scale[1], 24, L16.d, L16.n(), L16.maxPix, head_cls4_, head_conv1_.b, V5, L32.d, L32.total, V45, L16.total); }
    ProfScope ps(c.prof, c.st, "conv3x3_phase", shape_str(L4.total, 9 * 24 + 4 * 24, 24, 1));
    nn::FpnPhaseArgs a;
    a.fine = p[0]; a.ld_fine = 24; a.fine_scale = p_scale[3]; a.ld_fs = 24; a.Wf = head_wf_; a.wf_img = 0;
    a.coarse = p[1]; a.ld_coarse = 24; a.coarse_scale = p_scale[2]; a.ld_cs = 24; a.Wc = head_wc_;
    a.G = V45; a.gg = L16.d; a.g_plane = L16.total; a.y = h1; a.ldy = 24; a.act = ACT_RELU;
    nn::fpn_phase(c.st, a, 24, 24, L4.d, L8.d, L4.n(), L4.maxH, L4.maxW);
  } else if (nn::conv3_fpn_fused_supported(24, 24)) {   // the head conv gathers the four levels itself: no 96-channel fuse tensor
    ProfScope ps(c.prof, c.st, "conv3x3", shape_str(L4.total, 9 * 96, 24, 1));
    nn::conv3_fpn_fused(c.st, p[3], p[2], p[1], p[0], L32.d, L16.d, L8.d, L4.d, L4.n(), L4.maxH, L4.maxW, 24, p_scale,
                        head_conv1_.w, 24, head_conv1_.Npad, h1, 24, make_epi(head_conv1_, ACT_RELU));
  } else {
    float* fuse = c.arena->alloc<float>((size_t)L4.total * 96);
    { ProfScope ps(c.prof, c.st, "fpn_concat");
      nn::fpn_concat(c.st, p[3], p[2], p[1], p[0], L32.d, L16.d, L8.d, L4.d, L4.n(), L4.maxPix, 24, fuse, p_scale); }
    ProfScope ps(c.prof, c.st, "conv3x3", shape_str(L4.total, 9 * 96, 24, 1));
    nn::conv_sp(c.st, 3, 3, fuse, 96, L4.d, L4.n(), L4.maxH, L4.maxW, 96, head_conv1_.w, 24, head_conv1_.Npad, h1, 24,
                make_epi(head_conv1_, ACT_RELU));
  }
  float* map = c.arena->alloc<float>((size_t)L0.total);
  { ProfScope ps(c.prof, c.st, "db_head_tail");
    nn::db_head_tail(c.st, h1, L4.d, L0.d, L4.n(), L4.maxPix, dc1_w_, dc1_b_, dc2_w_, dc2_b_, map); }
  return map;
}

// ---------------------------------------------------------------------------
// RecNet
// ---------------------------------------------------------------------------
RecNet::RecNet(const Blob& b) {
  pack_stem(ws_, b, "rec.stem", 16, &stem_w_, &stem_b_);
  for (const LcSpec& s : REC_SPEC) blocks_.push_back(build_lc(ws_, b, "rec", s));
  const int C = 480, D = 120;
  conv1_ = pack_conv(ws_, b, "rec.neck.conv1", C / 8, C, 1, 3);
  conv2_ = pack_conv(ws_, b, "rec.neck.conv2", D, C / 8, 1, 1);
  conv3_ = pack_conv(ws_, b, "rec.neck.conv3", C, D, 1, 1);
  conv4_ = pack_conv(ws_, b, "rec.neck.conv4", C / 8, 2 * C, 1, 3);
  conv1x1_ = pack_conv(ws_, b, "rec.neck.conv1x1", D, C / 8, 1, 1);
  core_.load(ws_, b, "rec");
}

int RecNet::tokens_for_width(int w) {
  int wa = (w - 1) / 2 + 1, wc = (wa - 1) / 2 + 1;
  return wc >= 2 ? (wc - 2) / 2 + 1 : 0;
}

float* RecNet::run(RunCtx& c, const float* x, Level& L0, Level& Lt, int* idx_out, float* prob_out) {
  for (auto& g : L0.h) if (g.H != 48 || g.W < 8) throw RtError(3, "rec input must be 48 high and at least 8 wide");
  Level La = down_level(L0, 2, 2);
  std::vector<Level> lv; lv.reserve(16);
  lv.push_back(La);
  for (const LcBlock& b : blocks_) lv.push_back(down_level(lv.back(), b.sh, b.sw));
  Lt = pool_level(lv.back(), 3, 2);
  std::vector<Level*> ups = {&L0};
  for (auto& l : lv) ups.push_back(&l);
  ups.push_back(&Lt);
  upload_levels(c, ups);
  float* t = c.arena->alloc<float>((size_t)lv[0].total * 16);
  { ProfScope ps(c.prof, c.st, "stem");
    nn::stem_conv(c.st, x, L0.d, lv[0].d, lv[0].n(), lv[0].maxH, lv[0].maxW, 16, stem_w_, stem_b_, ACT_NONE, t); }
  for (size_t i = 0; i < blocks_.size(); i++) t = run_lc(c, blocks_[i], t, lv[i], lv[i + 1]);
  const long long rows = Lt.total;
  const int C = 480, D = 120;
  float* cat = c.arena->alloc<float>((size_t)rows * 2 * C);
  { ProfScope ps(c.prof, c.st, "avgpool");
    nn::avgpool_3x2(c.st, t, lv.back().d, Lt.d, Lt.n(), Lt.maxPix, C, cat, 2 * C); }
  float* z1 = c.arena->alloc<float>((size_t)rows * 60);
  // line-boundary flags of the flat token list (the two 1x3 convs run over it as one sequence: nn::conv13_flat)
  unsigned char* tok_flags = nullptr;
  if (nn::conv13_flat_supported(60, conv1_.Npad) && Lt.maxH == 1 && rows > 0) {
    unsigned char* hf = c.pinned->alloc<unsigned char>((size_t)rows);
    memset(hf, 0, (size_t)rows);
    for (const ImgGeom& g : Lt.h)
      if (g.H == 1 && g.W > 0) { hf[g.off] |= 1; hf[g.off + g.W - 1] |= 2; }
    tok_flags = c.arena->alloc<unsigned char>((size_t)rows);
    RT_HIP_CHECK(hipMemcpyAsync(tok_flags, hf, (size_t)rows, hipMemcpyHostToDevice, c.st));
  }
  { ProfScope ps(c.prof, c.st, "conv1x3");
    if (tok_flags) nn::conv13_flat(c.st, cat, 2 * C, rows, tok_flags, C, conv1_.w, 60, conv1_.Npad, z1, 60, make_epi(conv1_, ACT_SWISH));
    else
    nn::conv_sp(c.st, 1, 3, cat, 2 * C, Lt.d, Lt.n(), Lt.maxH, Lt.maxW, C, conv1_.w, 60, conv1_.Npad, z1, 60,
                make_epi(conv1_, ACT_SWISH)); }
  float* z = c.arena->alloc<float>((size_t)rows * D);
  { ProfScope ps(c.prof, c.st, "gemm_neck");
    nn::gemm(c.st, z1, 60, rows, conv2_.K, conv2_.w, D, conv2_.Npad, z, D, 0, make_epi(conv2_, ACT_SWISH)); }
  float* zf = core_.mixer(c, z, Lt);
  { ProfScope ps(c.prof, c.st, "gemm_neck");
    nn::gemm(c.st, zf, D, rows, conv3_.K, conv3_.w, C, conv3_.Npad, cat, 2 * C, C, make_epi(conv3_, ACT_SWISH)); }
  float* z4 = c.arena->alloc<float>((size_t)rows * 60);
  { ProfScope ps(c.prof, c.st, "conv1x3");
    if (tok_flags) nn::conv13_flat(c.st, cat, 2 * C, rows, tok_flags, 2 * C, conv4_.w, 60, conv4_.Npad, z4, 60, make_epi(conv4_, ACT_SWISH));
    else
    nn::conv_sp(c.st, 1, 3, cat, 2 * C, Lt.d, Lt.n(), Lt.maxH, Lt.maxW, 2 * C, conv4_.w, 60, conv4_.Npad, z4, 60,
                make_epi(conv4_, ACT_SWISH)); }
  float* z5 = c.arena->alloc<float>((size_t)rows * D);
  { ProfScope ps(c.prof, c.st, "gemm_neck");
    nn::gemm(c.st, z4, 60, rows, conv1x1_.K, conv1x1_.w, D, conv1x1_.Npad, z5, D, 0, make_epi(conv1x1_, ACT_SWISH)); }
  return core_.head(c, z5, rows, idx_out, prob_out);
}

// ---------------------------------------------------------------------------
// SvtrCore: EncoderWithSVTR's mixing blocks (post-norm: x = LN(x + MHA(x)); x = LN(x + MLP(x))), final LN, CTC FC
// ---------------------------------------------------------------------------
void SvtrCore::load(WeightStore& ws, const Blob& b, const std::string& prefix) {
  for (int i = 0; i < 2; i++) {
    std::string p = prefix + ".neck.blk" + std::to_string(i);
    blk[i].qkv = pack_linear(ws, b, p + ".qkv", D, 3 * D);
    blk[i].proj = pack_linear(ws, b, p + ".proj", D, D);
    blk[i].fc1 = pack_linear(ws, b, p + ".fc1", D, 2 * D);
    blk[i].fc2 = pack_linear(ws, b, p + ".fc2", 2 * D, D);
    blk[i].n1g = upload_raw(ws, b, p + ".norm1.g", D); blk[i].n1b = upload_raw(ws, b, p + ".norm1.beta", D);
    blk[i].n2g = upload_raw(ws, b, p + ".norm2.g", D); blk[i].n2b = upload_raw(ws, b, p + ".norm2.beta", D);
  }
  ng = upload_raw(ws, b, prefix + ".neck.norm.g", D); nb = upload_raw(ws, b, prefix + ".neck.norm.beta", D);
  const BlobTensor& fw = b.get(prefix + ".head.fc.w");
  if (fw.dims.size() != 2 || fw.dims[0] != D) throw RtError(3, "RTWB: unexpected shape for " + prefix + ".head.fc.w");
  classes = fw.dims[1];
  fc = pack_linear(ws, b, prefix + ".head.fc", D, classes);
}

float* SvtrCore::mixer(RunCtx& c, float* z, const Level& Lt) const {
  const long long rows = Lt.total;
  float* qkv = c.arena->alloc<float>((size_t)rows * 3 * D);
  float* a = c.arena->alloc<float>((size_t)rows * D);
  float* a2 = c.arena->alloc<float>((size_t)rows * D);
  float* m = c.arena->alloc<float>((size_t)rows * 2 * D);
  for (int i = 0; i < 2; i++) {
    const Blk& k = blk[i];
    { ProfScope ps(c.prof, c.st, "gemm_neck");
      nn::gemm(c.st, z, D, rows, k.qkv.K, k.qkv.w, 3 * D, k.qkv.Npad, qkv, 3 * D, 0, make_epi(k.qkv, ACT_NONE)); }
    { ProfScope ps(c.prof, c.st, "attention");
      nn::attention(c.st, qkv, Lt.d, Lt.n(), Lt.maxW * Lt.maxH, 8, D / 8, a); }
    { ProfScope ps(c.prof, c.st, "gemm_neck");
      nn::gemm(c.st, a, D, rows, k.proj.K, k.proj.w, D, k.proj.Npad, a2, D, 0, make_epi(k.proj, ACT_NONE)); }
    float* zn = c.arena->alloc<float>((size_t)rows * D);
    { ProfScope ps(c.prof, c.st, "layernorm");
      nn::add_layernorm(c.st, z, a2, rows, D, k.n1g, k.n1b, 1e-5f, zn); }
    { ProfScope ps(c.prof, c.st, "gemm_neck");
      nn::gemm(c.st, zn, D, rows, k.fc1.K, k.fc1.w, 2 * D, k.fc1.Npad, m, 2 * D, 0, make_epi(k.fc1, ACT_SWISH));
      nn::gemm(c.st, m, 2 * D, rows, k.fc2.K, k.fc2.w, D, k.fc2.Npad, a2, D, 0, make_epi(k.fc2, ACT_NONE)); }
    float* zo = c.arena->alloc<float>((size_t)rows * D);
    { ProfScope ps(c.prof, c.st, "layernorm");
      nn::add_layernorm(c.st, zn, a2, rows, D, k.n2g, k.n2b, 1e-5f, zo); }
    z = zo;
  }
  float* zf = c.arena->alloc<float>((size_t)rows * D);
  { ProfScope ps(c.prof, c.st, "layernorm");
    nn::add_layernorm(c.st, z, nullptr, rows, D, ng, nb, 1e-6f, zf); }
  return zf;
}

float* SvtrCore::head(RunCtx& c, const float* z5, long long rows, int* idx_out, float* prob_out) const {
  if (idx_out) {
    const int tiles = nn::gemm_argmax_tiles(fc.Npad);
    Epilogue e = make_epi(fc, ACT_NONE);
    e.am_max = c.arena->alloc<float>((size_t)rows * tiles);
    e.am_sum = c.arena->alloc<float>((size_t)rows * tiles);
    e.am_idx = c.arena->alloc<int>((size_t)rows * tiles);
    e.am_tiles = tiles;
    { ProfScope ps(c.prof, c.st, "gemm_ctc_fc", shape_str(rows, fc.K, classes, 1));
      nn::gemm(c.st, z5, D, rows, fc.K, fc.w, classes, fc.Npad, nullptr, 0, 0, e); }
    { ProfScope ps(c.prof, c.st, "ctc_argmax");
      nn::argmax_merge(c.st, e.am_max, e.am_idx, e.am_sum, tiles, rows, idx_out, prob_out); }
    return nullptr;
  }
  const int ld = round_up(classes, 4);
  float* logits = c.arena->alloc<float>((size_t)rows * ld);
  { ProfScope ps(c.prof, c.st, "gemm_ctc_fc", shape_str(rows, fc.K, classes, 0));
    nn::gemm(c.st, z5, D, rows, fc.K, fc.w, classes, fc.Npad, logits, ld, 0, make_epi(fc, ACT_NONE)); }
  return logits;
}

// ---------------------------------------------------------------------------
// ClsNet
// ---------------------------------------------------------------------------
struct ClsSpec { int k, mid, cout; bool se; int act, sh, sw; };
static const ClsSpec CLS_SPEC[] = {
    {3, 8, 8, true, ACT_RELU, 2, 1},      {3, 24, 8, false, ACT_RELU, 2, 1},    {3, 32, 8, false, ACT_RELU, 1, 1},
    {5, 32, 16, true, ACT_HSWISH, 2, 1},  {5, 88, 16, true, ACT_HSWISH, 1, 1},  {5, 88, 16, true, ACT_HSWISH, 1, 1},
    {5, 40, 16, true, ACT_HSWISH, 1, 1},  {5, 48, 16, true, ACT_HSWISH, 1, 1},  {5, 104, 32, true, ACT_HSWISH, 2, 1},
    {5, 200, 32, true, ACT_HSWISH, 1, 1}, {5, 200, 32, true, ACT_HSWISH, 1, 1}};

ClsNet::ClsNet(const Blob& b) {
  pack_stem(ws_, b, "cls.stem", 8, &stem_w_, &stem_b_);
  int cin = 8, i = 0;
  for (const ClsSpec& s : CLS_SPEC) {
    std::string p = "cls.b" + std::to_string(i++);
    B blk;
    blk.expand = pack_conv(ws_, b, p + ".expand", s.mid, cin, 1, 1);
    blk.dw = pack_dw(ws_, b, p + ".dw", s.mid, s.k);
    blk.se = s.se;
    if (s.se) blk.sew = get_se(ws_, b, p + ".se", s.mid);
    blk.linear = pack_conv(ws_, b, p + ".linear", s.cout, s.mid, 1, 1);
    blk.act = s.act; blk.sh = s.sh; blk.sw = s.sw;
    blk.shortcut = (s.sh == 1 && s.sw == 1 && cin == s.cout);
    blocks_.push_back(blk);
    cin = s.cout;
  }
  conv2_ = pack_conv(ws_, b, "cls.conv2", 200, cin, 1, 1);
  fc_ = pack_linear(ws_, b, "cls.head.fc", 200, 2);
}

float* ClsNet::run(RunCtx& c, const float* x, Level& L0) {
  std::vector<Level> lv; lv.reserve(16);
  lv.push_back(down_level(L0, 2, 2));
  for (const B& b : blocks_) lv.push_back(down_level(lv.back(), b.sh, b.sw));
  Level Lp = pool_level(lv.back(), 2, 2);
  std::vector<Level*> ups = {&L0};
  for (auto& l : lv) ups.push_back(&l);
  ups.push_back(&Lp);
  upload_levels(c, ups);
  float* t = c.arena->alloc<float>((size_t)lv[0].total * 8);
  { ProfScope ps(c.prof, c.st, "stem");
    nn::stem_conv(c.st, x, L0.d, lv[0].d, lv[0].n(), lv[0].maxH, lv[0].maxW, 8, stem_w_, stem_b_, ACT_HSWISH, t); }
  int cin = 8;
  for (size_t i = 0; i < blocks_.size(); i++) {
    const B& b = blocks_[i];
    const Level &Lin = lv[i], &Lout = lv[i + 1];
    if (nn::g_cls_fused && nn::cls_block_supported(b.dw.k, b.sh, b.sw, cin, b.dw.C, b.linear.N, b.act, Lin.maxH, Lin.maxW, (int)Lout.maxPix)) {
      // the whole block in one launch, a workgroup per crop (nn_clsblock.hip)
      const int co = b.linear.N;
      float* y = c.arena->alloc<float>((size_t)Lout.total * co);
      float* dscr = b.se ? c.arena->alloc<float>((size_t)Lout.total * round_up(b.dw.C, 16)) : nullptr;
      ProfScope ps(c.prof, c.st, "cls_block", shape_str(Lin.total, b.dw.C, co, cin * 1000 + b.dw.k * 100 + b.sh * 10 + (b.se ? 1 : 0)));
      nn::cls_block(c.st, b.dw.k, b.sh, b.se, b.act, t, Lin.d, Lout.d, Lout.n(), Lin.maxH, Lin.maxW, (int)Lout.maxPix, cin, b.dw.C, b.dw.Cp, co, b.expand.w,
                    b.expand.b, b.dw.w, b.dw.b, b.se ? b.sew.w1 : nullptr, b.se ? b.sew.b1 : nullptr, b.se ? b.sew.w2 : nullptr,
                    b.se ? b.sew.b2 : nullptr, b.se ? b.sew.Cr : 0, HSIG_MBV3, b.linear.w, b.linear.b, b.shortcut, y, dscr);
      t = y; cin = co;
      continue;
    }
    int mid = b.dw.Cp;
    float* e = c.arena->alloc<float>((size_t)Lin.total * mid);
    { ProfScope ps(c.prof, c.st, "gemm_cls");
      nn::gemm(c.st, t, cin, Lin.total, b.expand.K, b.expand.w, b.expand.N, b.expand.Npad, e, mid, 0,
               make_epi(b.expand, b.act)); }
    float* d = c.arena->alloc<float>((size_t)Lout.total * mid);
    { ProfScope ps(c.prof, c.st, b.dw.k == 3 ? "dwconv3" : "dwconv5");
      nn::dwconv(c.st, b.dw.k, b.sh, b.sw, e, Lin.d, Lout.d, Lout.n(), Lout.maxH, Lout.maxW, mid, b.dw.C, b.dw.w, b.dw.b, b.act, 0,
                 1.f, 0.f, d); }
    if (b.se) run_se(c, d, Lout, b.sew, HSIG_MBV3, 0);
    int co = b.linear.N;
    float* y = c.arena->alloc<float>((size_t)Lout.total * co);
    { ProfScope ps(c.prof, c.st, "gemm_cls");
      nn::gemm(c.st, d, mid, Lout.total, b.linear.K, b.linear.w, co, b.linear.Npad, y, co, 0,
               make_epi(b.linear, ACT_NONE, nullptr, b.shortcut ? t : nullptr, cin)); }
    t = y; cin = co;
  }
  const Level& Ll = lv.back();
  float* f = c.arena->alloc<float>((size_t)Ll.total * 200);
  { ProfScope ps(c.prof, c.st, "gemm_cls");
    nn::gemm(c.st, t, cin, Ll.total, conv2_.K, conv2_.w, 200, conv2_.Npad, f, 200, 0, make_epi(conv2_, ACT_HSWISH)); }
  float* mp = c.arena->alloc<float>((size_t)Lp.total * 200);
  { ProfScope ps(c.prof, c.st, "maxpool");
    nn::maxpool_2x2(c.st, f, Ll.d, Lp.d, Lp.n(), Lp.maxPix, 200, mp); }
  float* partial = c.arena->alloc<float>((size_t)Lp.n() * nn::pool_chunks(Lp.maxPix) * 200);
  float* gm = c.arena->alloc<float>((size_t)Lp.n() * 200);
  { ProfScope ps(c.prof, c.st, "global_mean");
    nn::global_mean(c.st, mp, Lp.d, Lp.n(), Lp.maxPix, 200, partial, gm); }
  float* logits = c.arena->alloc<float>((size_t)Lp.n() * 4);
  float* probs = c.arena->alloc<float>((size_t)Lp.n() * 2);
  { ProfScope ps(c.prof, c.st, "gemm_cls");
    nn::gemm(c.st, gm, 200, Lp.n(), fc_.K, fc_.w, 2, fc_.Npad, logits, 4, 0, make_epi(fc_, ACT_NONE)); }
  { ProfScope ps(c.prof, c.st, "softmax");
    nn::softmax_rows(c.st, logits, 4, Lp.n(), 2, probs); }
  return probs;
}

// ---------------------------------------------------------------------------
// Model manifest: the RTWB tensor list of each network (onnx_import.h), in the order the
// parameters appear in the forward graph of the PaddleOCR modules (same-shaped layers are told
// apart by that order only: the RSEFPN runs its levels top-down, ins3..ins0 then inp3..inp0).
// Must stay in sync with the constructors above and with retto_amd/synth.py (tests check both).
// ---------------------------------------------------------------------------
static void mf_conv(std::vector<ManifestEntry>& m, const std::string& n, int cout, int cin_g, int kh, int kw, bool bias = true) {
  m.push_back({n + ".w", {cout, cin_g, kh, kw}});
  if (bias) m.push_back({n + ".b", {cout}});
}
static void mf_lab(std::vector<ManifestEntry>& m, const std::string& n) { m.push_back({n + ".a", {1}}); m.push_back({n + ".c", {1}}); }
static void mf_se(std::vector<ManifestEntry>& m, const std::string& n, int c) {
  mf_conv(m, n + ".fc1", c / 4, c, 1, 1);
  mf_conv(m, n + ".fc2", c, c / 4, 1, 1);
}
static void mf_linear(std::vector<ManifestEntry>& m, const std::string& n, int cin, int cout) {
  m.push_back({n + ".w", {cin, cout}}); m.push_back({n + ".b", {cout}});
}
static void mf_ln(std::vector<ManifestEntry>& m, const std::string& n, int c) { m.push_back({n + ".g", {c}}); m.push_back({n + ".beta", {c}}); }
template <size_t N>
static void mf_lcnet(std::vector<ManifestEntry>& m, const std::string& prefix, const LcSpec (&spec)[N], bool det) {
  mf_conv(m, prefix + ".stem", 16, 3, 3, 3);
  for (const LcSpec& s : spec) {
    const std::string p = prefix + "." + s.name;
    mf_conv(m, p + ".dw", s.cin, 1, s.k, s.k);
    if (!det || !(s.sh == 2 && s.sw == 2)) mf_lab(m, p + ".dw");  // LearnableRepLayer: act (and its LAB) unless stride == 2
    if (s.se) mf_se(m, p + ".se", s.cin);
    mf_conv(m, p + ".pw", s.cout, s.cin, 1, 1);
    mf_lab(m, p + ".pw");
  }
}

std::vector<ManifestEntry> model_manifest(int which) {
  std::vector<ManifestEntry> m;
  if (which == MODEL_DET) {
    mf_lcnet(m, "det", DET_SPEC, true);
    const int tap_c[4] = {48, 96, 192, 384}, out_c[4] = {12, 18, 42, 360};
    for (int j = 0; j < 4; j++) mf_conv(m, "det.out" + std::to_string(j), out_c[j], tap_c[j], 1, 1);
    for (int j = 3; j >= 0; j--) {
      mf_conv(m, "det.fpn.ins" + std::to_string(j), 96, out_c[j], 1, 1, false);
      mf_se(m, "det.fpn.ins" + std::to_string(j) + ".se", 96);
    }
    for (int j = 3; j >= 0; j--) {
      mf_conv(m, "det.fpn.inp" + std::to_string(j), 24, 96, 3, 3, false);
      mf_se(m, "det.fpn.inp" + std::to_string(j) + ".se", 24);
    }
    mf_conv(m, "det.head.conv1", 24, 96, 3, 3);
    m.push_back({"det.head.deconv1.w", {24, 24, 2, 2}}); m.push_back({"det.head.deconv1.b", {24}});
    m.push_back({"det.head.deconv2.w", {24, 1, 2, 2}}); m.push_back({"det.head.deconv2.b", {1}});
  } else if (which == MODEL_REC) {
    mf_lcnet(m, "rec", REC_SPEC, false);
    const int C = 480, D = 120;
    mf_conv(m, "rec.neck.conv1", C / 8, C, 1, 3);
    mf_conv(m, "rec.neck.conv2", D, C / 8, 1, 1);
    for (int i = 0; i < 2; i++) {
      const std::string p = "rec.neck.blk" + std::to_string(i);
      mf_linear(m, p + ".qkv", D, 3 * D);
      mf_linear(m, p + ".proj", D, D);
      mf_ln(m, p + ".norm1", D);
      mf_linear(m, p + ".fc1", D, 2 * D);
      mf_linear(m, p + ".fc2", 2 * D, D);
      mf_ln(m, p + ".norm2", D);
    }
    mf_ln(m, "rec.neck.norm", D);
    mf_conv(m, "rec.neck.conv3", C, D, 1, 1);
    mf_conv(m, "rec.neck.conv4", C / 8, 2 * C, 1, 3);
    mf_conv(m, "rec.neck.conv1x1", D, C / 8, 1, 1);
    m.push_back({"rec.head.fc.w", {D, -1}}); m.push_back({"rec.head.fc.b", {-1}});
  } else if (which == MODEL_CLS) {
    mf_conv(m, "cls.stem", 8, 3, 3, 3);
    int cin = 8, i = 0;
    for (const ClsSpec& s : CLS_SPEC) {
      const std::string p = "cls.b" + std::to_string(i++);
      mf_conv(m, p + ".expand", s.mid, cin, 1, 1);
      mf_conv(m, p + ".dw", s.mid, 1, s.k, s.k);
      if (s.se) mf_se(m, p + ".se", s.mid);
      mf_conv(m, p + ".linear", s.cout, s.mid, 1, 1);
      cin = s.cout;
    }
    mf_conv(m, "cls.conv2", 200, cin, 1, 1);
    mf_linear(m, "cls.head.fc", 200, 2);
  } else if (which == MODEL_SDET || which == MODEL_SREC) {
    // PP-OCRv4 server graphs (nets_f16.cpp: DetServerH / RecServerH; layer names as retto_amd/synth.py writes them), in the
    // forward order of the PaddleOCR modules: PPHGNet_small (stem, then per stage the depthwise down-sample, per block the six
    // 3x3 layers, the 1x1 aggregation and the ESE gate's 1x1), LKPAN with IntraCL blocks, PFHeadLocal.
    const bool det = which == MODEL_SDET;
    const std::string pre = det ? "sdet" : "srec";
    const int stem[3] = {64, 64, 128};
    int cin = 3;
    for (int i = 0; i < 3; i++) { mf_conv(m, pre + ".stem" + std::to_string(i), stem[i], cin, 3, 3); cin = stem[i]; }
    struct St { const char* name; int cin, mid, cout, blocks; bool down; };
    const St stages[4] = {{"st1", 128, 128, 256, 1, !det}, {"st2", 256, 160, 512, 1, true}, {"st3", 512, 192, 768, 2, true}, {"st4", 768, 224, 1024, 1, true}};
    for (const St& st : stages) {
      const std::string p = pre + "." + st.name;
      if (st.down) mf_conv(m, p + ".ds", st.cin, 1, 3, 3);
      for (int b = 0; b < st.blocks; b++) {
        const int bin = b == 0 ? st.cin : st.cout;
        int c = bin;
        const std::string pb = p + ".b" + std::to_string(b);
        for (int l = 0; l < 6; l++) { mf_conv(m, pb + ".l" + std::to_string(l), st.mid, c, 3, 3); c = st.mid; }
        mf_conv(m, pb + ".agg", st.cout, bin + 6 * st.mid, 1, 1);
        mf_conv(m, pb + ".ese", st.cout, st.cout, 1, 1);
      }
    }
    if (det) {
      const int C = 256, Q = 64, R = 32, outs[4] = {256, 512, 768, 1024};
      for (int j = 3; j >= 0; j--) mf_conv(m, "sdet.neck.ins" + std::to_string(j), C, outs[j], 1, 1, false);
      for (int j = 3; j >= 0; j--) mf_conv(m, "sdet.neck.inp" + std::to_string(j), Q, C, 9, 9, false);
      for (int j = 0; j < 3; j++) mf_conv(m, "sdet.neck.panhead" + std::to_string(j), Q, Q, 3, 3, false);
      for (int j = 0; j < 4; j++) mf_conv(m, "sdet.neck.panlat" + std::to_string(j), Q, Q, 9, 9, false);
      for (int j = 4; j >= 1; j--) {
        const std::string p = "sdet.neck.incl" + std::to_string(j);
        mf_conv(m, p + ".reduce", R, Q, 1, 1);
        for (int k : {7, 5, 3}) {
          mf_conv(m, p + ".c" + std::to_string(k), R, R, k, k);
          mf_conv(m, p + ".v" + std::to_string(k), R, R, k, 1);
          mf_conv(m, p + ".q" + std::to_string(k), R, R, 1, k);
        }
        mf_conv(m, p + ".ret", Q, R, 1, 1);
      }
      mf_conv(m, "sdet.head.conv1", Q, C, 3, 3);
      m.push_back({"sdet.head.deconv1.w", {Q, Q, 2, 2}}); m.push_back({"sdet.head.deconv1.b", {Q}});
      m.push_back({"sdet.head.deconv2.w", {Q, 1, 2, 2}}); m.push_back({"sdet.head.deconv2.b", {1}});
      mf_conv(m, "sdet.head.local3", Q, Q + 1, 3, 3);
      mf_conv(m, "sdet.head.local1", 1, Q, 1, 1);
    } else {
      const int C = 1024, D = 120;
      mf_conv(m, "srec.neck.conv1", C / 8, C, 1, 3);
      mf_conv(m, "srec.neck.conv2", D, C / 8, 1, 1);
      for (int i = 0; i < 2; i++) {
        const std::string p = "srec.neck.blk" + std::to_string(i);
        mf_linear(m, p + ".qkv", D, 3 * D);
        mf_linear(m, p + ".proj", D, D);
        mf_ln(m, p + ".norm1", D);
        mf_linear(m, p + ".fc1", D, 2 * D);
        mf_linear(m, p + ".fc2", 2 * D, D);
        mf_ln(m, p + ".norm2", D);
      }
      mf_ln(m, "srec.neck.norm", D);
      mf_conv(m, "srec.neck.conv3", C, D, 1, 1);
      mf_conv(m, "srec.neck.conv4", C / 8, 2 * C, 1, 3);
      mf_conv(m, "srec.neck.conv1x1", D, C / 8, 1, 1);
      m.push_back({"srec.head.fc.w", {D, -1}}); m.push_back({"srec.head.fc.b", {-1}});
    }
  } else {
    throw RtError(8, "model_manifest: unknown model kind");
  }
  return m;
}

}  // namespace rt
