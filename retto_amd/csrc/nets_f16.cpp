#include "nets_f16.h"

#include <algorithm>
#include <cstring>

namespace rt {

using nh::half_t;
using nh::pitch8;
using nh::Epi16;

static const float HSIG_LCNET = 0.1666667f;  // paddle nn.Hardsigmoid
static const float HSIG_MBV3 = 0.2f;         // F.hardsigmoid(slope=0.2, offset=0.5)
static const float GATE_SIGMOID = -1.f;      // ESEModule: sigmoid gate

bool blob_is_server_det(const Blob& b) { return b.has("sdet.stem0.w"); }
bool blob_is_server_rec(const Blob& b) { return b.has("srec.stem0.w"); }

// ---------------------------------------------------------------------------
// weights: fp32 RTWB tensors -> fp16 device layouts
// ---------------------------------------------------------------------------
static void expect_dims16(const BlobTensor& t, std::initializer_list<int> d, const std::string& name) {
  if (t.dims != std::vector<int>(d)) throw RtError(3, "RTWB: unexpected shape for " + name);
}
// host image of a conv16 weight: [nslab][kh][kw][npad][32], element (n, k, dy, dx) given by w(n, k, dy, dx)
template <class F>
static Conv16 upload_conv16(WeightStore& ws, int cout, int cin, int kh, int kw, F&& wfun, const float* bias) {
  Conv16 p;
  p.cin = cin; p.cin_p = pitch8(cin); p.cout = cout; p.npad = round_up(cout, 32); p.kh = kh; p.kw = kw;
  const int nslab = (p.cin_p + nh::KS - 1) / nh::KS;
  std::vector<half_t> host((size_t)nslab * kh * kw * p.npad * nh::KS, (half_t)0.f);
  for (int n = 0; n < cout; n++)
    for (int k = 0; k < cin; k++)
      for (int dy = 0; dy < kh; dy++)
        for (int dx = 0; dx < kw; dx++) {
          const int s = k / nh::KS, kk = k % nh::KS;
          host[((((size_t)s * kh + dy) * kw + dx) * p.npad + n) * nh::KS + kk] = (half_t)wfun(n, k, dy, dx);
        }
  p.w = (half_t*)ws.upload_bytes(host.data(), host.size() * sizeof(half_t));
  std::vector<float> b(p.npad, 0.f);
  if (bias) memcpy(b.data(), bias, (size_t)cout * sizeof(float));
  p.b = ws.upload(b);
  return p;
}
static Conv16 pack_conv16(WeightStore& ws, const Blob& b, const std::string& name, int cout, int cin, int kh, int kw) {
  const BlobTensor& w = b.get(name + ".w");
  expect_dims16(w, {cout, cin, kh, kw}, name + ".w");
  const float* bias = nullptr;
  if (b.has(name + ".b")) { const BlobTensor& bt = b.get(name + ".b"); expect_dims16(bt, {cout}, name + ".b"); bias = bt.data; }
  const float* d = w.data;
  return upload_conv16(ws, cout, cin, kh, kw,
                       [&](int n, int k, int dy, int dx) { return d[(((size_t)n * cin + k) * kh + dy) * kw + dx]; }, bias);
}
// ConvTranspose 2x2 stride 2 (weight [cin, cout, 2, 2]) as a 1x1 conv to 4 * cout channels ordered (dy, dx, co)
static Conv16 pack_deconv16(WeightStore& ws, const Blob& b, const std::string& name, int cin, int cout) {
  const BlobTensor& w = b.get(name + ".w");
  expect_dims16(w, {cin, cout, 2, 2}, name + ".w");
  const BlobTensor& bt = b.get(name + ".b");
  expect_dims16(bt, {cout}, name + ".b");
  std::vector<float> bias((size_t)4 * cout);
  for (int q = 0; q < 4; q++) memcpy(bias.data() + (size_t)q * cout, bt.data, (size_t)cout * sizeof(float));
  const float* d = w.data;
  return upload_conv16(ws, 4 * cout, cin, 1, 1,
                       [&](int n, int k, int, int) { const int q = n / cout, co = n % cout; return d[((size_t)k * cout + co) * 4 + q]; },
                       bias.data());
}
static Dw16 pack_dw16(WeightStore& ws, const Blob& b, const std::string& name, int C, int k) {
  const BlobTensor& w = b.get(name + ".w");
  expect_dims16(w, {C, 1, k, k}, name + ".w");
  Dw16 p; p.k = k; p.C = C; p.Cp = pitch8(C);
  std::vector<half_t> host((size_t)k * k * p.Cp, (half_t)0.f);
  std::vector<float> bias(p.Cp, 0.f);
  for (int c = 0; c < C; c++)
    for (int t = 0; t < k * k; t++) host[(size_t)t * p.Cp + c] = (half_t)w.data[(size_t)c * k * k + t];
  const BlobTensor& bt = b.get(name + ".b");
  expect_dims16(bt, {C}, name + ".b");
  memcpy(bias.data(), bt.data, (size_t)C * sizeof(float));
  p.w = (half_t*)ws.upload_bytes(host.data(), host.size() * sizeof(half_t));
  p.b = ws.upload(bias);
  return p;
}
// fp32 FC for nn::gemm from a conv-style weight [cout][cin] (1x1): packed [K/32][Npad16][32]
static PackedDense pack_fc32(WeightStore& ws, const BlobTensor& w, int cout, int cin, const float* bias) {
  if (w.numel() != (size_t)cout * cin) throw RtError(3, "RTWB: unexpected size of a squeeze-excite weight");
  PackedDense p;
  p.K = round_up(cin, 4); p.N = cout; p.Npad = round_up(cout, 16);
  const int nkc = (p.K + nn::KC - 1) / nn::KC;
  std::vector<float> host((size_t)nkc * p.Npad * nn::KC, 0.f);
  for (int n = 0; n < cout; n++)
    for (int k = 0; k < cin; k++) host[((size_t)(k / nn::KC) * p.Npad + n) * nn::KC + (k % nn::KC)] = w.data[(size_t)n * cin + k];
  p.w = ws.upload(host);
  std::vector<float> b(p.Npad, 0.f);
  memcpy(b.data(), bias, (size_t)cout * sizeof(float));
  p.b = ws.upload(b);
  return p;
}
// squeeze-excite: fc1 [Cr, C, 1, 1], fc2 [C, Cr, 1, 1]
static Se16 get_se16(WeightStore& ws, const Blob& b, const std::string& name, int C) {
  Se16 s; s.C = C; s.Cr = C / 4; s.has_fc1 = true;
  s.fc1 = pack_fc32(ws, b.get(name + ".fc1.w"), s.Cr, C, b.get(name + ".fc1.b").data);
  s.fc2 = pack_fc32(ws, b.get(name + ".fc2.w"), C, s.Cr, b.get(name + ".fc2.b").data);
  return s;
}
// ESEModule: one 1x1 conv C -> C on the channel means, sigmoid gate
static Se16 get_ese16(WeightStore& ws, const Blob& b, const std::string& name, int C) {
  Se16 s; s.C = C; s.Cr = C;
  const BlobTensor& w = b.get(name + ".w");
  expect_dims16(w, {C, C, 1, 1}, name + ".w");
  s.fc2 = pack_fc32(ws, w, C, C, b.get(name + ".b").data);
  return s;
}
static Lab get_lab16(const Blob& b, const std::string& name) {
  Lab l;
  if (b.has(name + ".a")) { l.has = 1; l.a = b.get(name + ".a").data[0]; l.c = b.get(name + ".c").data[0]; }
  return l;
}

// ---------------------------------------------------------------------------
// run helpers
// ---------------------------------------------------------------------------
static H16 alloc16(RunCtx& c, const Level& L, int C) {
  H16 t; t.C = C; t.ld = pitch8(C);
  t.p = c.arena->alloc<half_t>((size_t)std::max<long long>(L.total, 1) * t.ld);
  return t;
}
static H16 view16(H16 t, int coff, int C) { H16 v; v.p = t.p + coff; v.C = C; v.ld = t.ld; return v; }
static std::string shp(long long a, long long b, long long c, long long d) {
  return std::to_string(a) + "," + std::to_string(b) + "," + std::to_string(c) + "," + std::to_string(d);
}
static Epi16 epi16(const Conv16& w, int act, const Lab* lab = nullptr, const half_t* res = nullptr, int ld_res = 0) {
  Epi16 e; e.bias = w.b; e.act = act;
  if (lab && lab->has) { e.has_lab = 1; e.lab_a = lab->a; e.lab_c = lab->c; }
  e.residual = res; e.ld_res = ld_res;
  return e;
}
// spatial conv (pad = k/2 "same" unless given), x at level Lin -> y at level Lout
static void conv_sp16(RunCtx& c, const Conv16& w, H16 x, const Level& Lin, const Level& Lout, int sh, int sw, H16 y, int coff,
                      const Epi16& e, int pt = -1, int pl = -1) {
  if (x.C != w.cin) throw RtError(3, "conv16: input has " + std::to_string(x.C) + " channels, weights expect " + std::to_string(w.cin));
  ProfScope ps(c.prof, c.st, nh::conv16_label(w.kh, w.kw, w.cout, w.cin), shp(Lout.total, (long long)w.kh * w.kw * w.cin, w.cout, sh * 10 + sw));
  nh::conv16(c.st, x.p, x.ld, Lin.d, Lout.d, Lout.n(), Lout.maxH, Lout.maxW, w.cin_p, w.kh, w.kw, sh, sw, pt < 0 ? w.kh / 2 : pt,
             pl < 0 ? w.kw / 2 : pl, w.w, w.cout, w.npad, y.p, y.ld, coff, e);
}
// 1x1 conv over every pixel of a level as one GEMM (F = flat_level of the tensor's level)
static void conv_pw16(RunCtx& c, const Conv16& w, H16 x, const Level& F, H16 y, int coff, const Epi16& e) {
  if (x.C != w.cin) throw RtError(3, "conv16 (1x1): input has " + std::to_string(x.C) + " channels, weights expect " + std::to_string(w.cin));
  ProfScope ps(c.prof, c.st, nh::conv16_label(1, 1, w.cout), shp(F.total, w.cin, w.cout, 0));
  nh::conv16(c.st, x.p, x.ld, F.d, F.d, 1, 1, F.maxW, w.cin_p, 1, 1, 1, 1, 0, 0, w.w, w.cout, w.npad, y.p, y.ld, coff, e);
}
static void dw16(RunCtx& c, const Dw16& w, H16 x, const Level& Lin, const Level& Lout, int sh, int sw, int act, const Lab& lab, H16 y) {
  ProfScope ps(c.prof, c.st, w.k == 3 ? "dwconv16_3" : "dwconv16_5", shp(Lin.total, Lout.total, w.Cp, sh * 10 + sw));
  nh::dwconv16(c.st, w.k, sh, sw, x.p, x.ld, Lin.d, Lout.d, Lout.n(), Lout.maxH, Lout.maxW, w.Cp, w.w, w.b, act, lab.has, lab.a, lab.c,
               y.p, y.ld);
}
// gate factors of a squeeze-excite / ESE layer: deterministic spatial mean (fp32), the FCs as fp32 GEMMs over all images of the
// level at once (rows = images), then the gate
static float* se16(RunCtx& c, const Se16& s, H16 x, const Level& L, float slope, int residual) {
  const int Cp = pitch8(s.C), n = L.n();
  float* partial = c.arena->alloc<float>((size_t)n * nh::pool_chunks16(L.maxPix) * Cp);
  float* mean = c.arena->alloc<float>((size_t)n * Cp);
  float* scale = c.arena->alloc<float>((size_t)n * Cp);
  ProfScope ps(c.prof, c.st, "se_pool_fc16");
  nh::global_mean16(c.st, x.p, x.ld, L.d, n, L.maxPix, Cp, partial, mean);
  const float* in = mean; int ldin = Cp;
  if (s.has_fc1) {
    const int Crp = round_up(s.Cr, 4);
    float* hid = c.arena->alloc<float>((size_t)n * Crp);
    nn::gemm(c.st, mean, Cp, n, s.fc1.K, s.fc1.w, s.Cr, s.fc1.Npad, hid, Crp, 0, make_epi(s.fc1, ACT_RELU));
    in = hid; ldin = Crp;
  }
  float* sv = c.arena->alloc<float>((size_t)n * Cp);
  nn::gemm(c.st, in, ldin, n, s.fc2.K, s.fc2.w, s.C, s.fc2.Npad, sv, Cp, 0, make_epi(s.fc2, ACT_NONE));
  nh::gate16(c.st, sv, Cp, n, s.C, Cp, slope, residual, scale);
  return scale;
}
static void scale16(RunCtx& c, H16 x, const Level& L, const float* scale, const H16* res, H16 y) {
  ProfScope ps(c.prof, c.st, "scale_channels16");
  nh::scale_channels16(c.st, x.p, x.ld, L.d, L.n(), L.maxPix, pitch8(x.C), scale, res ? res->p : nullptr, res ? res->ld : 0, y.p, y.ld);
}

static LcBlock16 build_lc16(WeightStore& ws, const Blob& b, const std::string& p, int k, int cin, int cout, int sh, int sw, bool se) {
  LcBlock16 blk;
  blk.dw = pack_dw16(ws, b, p + ".dw", cin, k);
  blk.dw_lab = get_lab16(b, p + ".dw");
  blk.dw_act = blk.dw_lab.has ? ACT_HSWISH : ACT_NONE;  // LearnableRepLayer: act only when stride != 2
  blk.se = se;
  if (se) blk.sew = get_se16(ws, b, p + ".se", cin);
  blk.pw = pack_conv16(ws, b, p + ".pw", cout, cin, 1, 1);
  blk.pw_lab = get_lab16(b, p + ".pw");
  blk.sh = sh; blk.sw = sw; blk.cin = cin; blk.cout = cout;
  return blk;
}
// x at Lin -> LCNetV3 block output at Lout (Fout = flat view of Lout)
static H16 run_lc16(RunCtx& c, const LcBlock16& b, H16 x, const Level& Lin, const Level& Lout, const Level& Fout) {
  H16 y1 = alloc16(c, Lout, b.cin);
  dw16(c, b.dw, x, Lin, Lout, b.sh, b.sw, b.dw_act, b.dw_lab, y1);
  if (b.se) {
    float* s = se16(c, b.sew, y1, Lout, HSIG_LCNET, 0);
    scale16(c, y1, Lout, s, nullptr, y1);
  }
  H16 y2 = alloc16(c, Lout, b.cout);
  conv_pw16(c, b.pw, y1, Fout, y2, 0, epi16(b.pw, ACT_HSWISH, &b.pw_lab));
  return y2;
}

struct LcSpec16 { const char* name; int k, cin, cout, sh, sw; bool se; };
static const LcSpec16 DET_SPEC16[] = {
    {"s2.0", 3, 16, 32, 1, 1, false}, {"s3.0", 3, 32, 48, 2, 2, false}, {"s3.1", 3, 48, 48, 1, 1, false},
    {"s4.0", 3, 48, 96, 2, 2, false}, {"s4.1", 3, 96, 96, 1, 1, false}, {"s5.0", 3, 96, 192, 2, 2, false},
    {"s5.1", 5, 192, 192, 1, 1, false}, {"s5.2", 5, 192, 192, 1, 1, false}, {"s5.3", 5, 192, 192, 1, 1, false},
    {"s5.4", 5, 192, 192, 1, 1, false}, {"s6.0", 5, 192, 384, 2, 2, true}, {"s6.1", 5, 384, 384, 1, 1, true},
    {"s6.2", 5, 384, 384, 1, 1, false}, {"s6.3", 5, 384, 384, 1, 1, false}};
static const LcSpec16 REC_SPEC16[] = {
    {"s2.0", 3, 16, 32, 1, 1, false}, {"s3.0", 3, 32, 64, 1, 1, false}, {"s3.1", 3, 64, 64, 1, 1, false},
    {"s4.0", 3, 64, 128, 2, 1, false}, {"s4.1", 3, 128, 128, 1, 1, false}, {"s5.0", 3, 128, 240, 1, 2, false},
    {"s5.1", 5, 240, 240, 1, 1, false}, {"s5.2", 5, 240, 240, 1, 1, false}, {"s5.3", 5, 240, 240, 1, 1, false},
    {"s5.4", 5, 240, 240, 1, 1, false}, {"s6.0", 5, 240, 480, 2, 1, true}, {"s6.1", 5, 480, 480, 1, 1, true},
    {"s6.2", 5, 480, 480, 2, 1, false}, {"s6.3", 5, 480, 480, 1, 1, false}};

// det input: RGB8 pages -> [pix][8] halves (B, G, R, 0...), or the f32 NHWC-4 tensor of the L1 entry point
static H16 det_input_u8(RunCtx& c, const nn::U8Page* pages, float scale, const float* mean3, const float* std3, const Level& L0) {
  H16 x; x.C = 3; x.ld = 8;
  x.p = c.arena->alloc<half_t>((size_t)L0.total * 8);
  static_assert(sizeof(nn::U8Page) == sizeof(nh::U8Page16), "page descriptors must match");
  ProfScope ps(c.prof, c.st, "u8_to_f16");
  nh::u8_to_h8(c.st, reinterpret_cast<const nh::U8Page16*>(pages), L0.n(), L0.maxPix, scale, mean3, std3, x.p);
  return x;
}
static H16 input_f32(RunCtx& c, const float* x4, const Level& L0) {
  H16 x; x.C = 3; x.ld = 8;
  x.p = c.arena->alloc<half_t>((size_t)L0.total * 8);
  ProfScope ps(c.prof, c.st, "f32_to_f16");
  nh::f32x4_to_h8(c.st, x4, L0.total, x.p);
  return x;
}

// ---------------------------------------------------------------------------
// DetNetH: PP-OCRv4 mobile det in fp16 (same graph as DetNet, nets.cpp)
// ---------------------------------------------------------------------------
DetNetH::DetNetH(const Blob& b) {
  stem_ = pack_conv16(ws_, b, "det.stem", 16, 3, 3, 3);
  for (const LcSpec16& s : DET_SPEC16) blocks_.push_back(build_lc16(ws_, b, std::string("det.") + s.name, s.k, s.cin, s.cout, s.sh, s.sw, s.se));
  tap_after_[0] = 2; tap_after_[1] = 4; tap_after_[2] = 9; tap_after_[3] = 13;
  const int tap_c[4] = {48, 96, 192, 384}, out_c[4] = {12, 18, 42, 360};
  for (int j = 0; j < 4; j++) {
    const std::string js = std::to_string(j);
    out_[j] = pack_conv16(ws_, b, "det.out" + js, out_c[j], tap_c[j], 1, 1);
    ins_[j] = pack_conv16(ws_, b, "det.fpn.ins" + js, 96, out_c[j], 1, 1);
    ins_se_[j] = get_se16(ws_, b, "det.fpn.ins" + js + ".se", 96);
    inp_[j] = pack_conv16(ws_, b, "det.fpn.inp" + js, 24, 96, 3, 3);
    inp_se_[j] = get_se16(ws_, b, "det.fpn.inp" + js + ".se", 24);
  }
  head_conv1_ = pack_conv16(ws_, b, "det.head.conv1", 24, 96, 3, 3);
  dc1_ = pack_deconv16(ws_, b, "det.head.deconv1", 24, 24);
  dc2_w_ = upload_raw(ws_, b, "det.head.deconv2.w", 24 * 4);
  dc2_b_ = b.get("det.head.deconv2.b").data[0];
}

float* DetNetH::run(RunCtx& c, const float* x, Level& L0) {
  // (the levels are uploaded by forward(); the input conversion only needs the pixel count)
  return forward(c, input_f32(c, x, L0), L0);
}
float* DetNetH::run_u8(RunCtx& c, const nn::U8Page* pages, float scale, const float* mean3, const float* std3, Level& L0) {
  return forward(c, det_input_u8(c, pages, scale, mean3, std3, L0), L0);
}

float* DetNetH::forward(RunCtx& c, H16 x, Level& L0) {
  for (auto& g : L0.h)
    if (g.H % 32 != 0 || g.W % 32 != 0 || g.H == 0 || g.W == 0) throw RtError(3, "det input sides must be non-zero multiples of 32");
  Level L2 = down_level(L0, 2, 2), L4 = down_level(L2, 2, 2), L8 = down_level(L4, 2, 2), L16 = down_level(L8, 2, 2),
        L32 = down_level(L16, 2, 2);
  Level F2 = flat_level(L2), F4 = flat_level(L4), F8 = flat_level(L8), F16 = flat_level(L16), F32 = flat_level(L32);
  upload_levels(c, {&L0, &L2, &L4, &L8, &L16, &L32, &F2, &F4, &F8, &F16, &F32});
  Level* lv[6] = {&L0, &L2, &L4, &L8, &L16, &L32};
  Level* fl[6] = {nullptr, &F2, &F4, &F8, &F16, &F32};
  H16 t = alloc16(c, L2, 16);
  conv_sp16(c, stem_, x, L0, L2, 2, 2, t, 0, epi16(stem_, ACT_NONE));
  int li = 1;
  H16 taps[4]; Level* tap_lv[4] = {nullptr, nullptr, nullptr, nullptr}; Level* tap_fl[4] = {nullptr, nullptr, nullptr, nullptr};
  for (size_t i = 0; i < blocks_.size(); i++) {
    const LcBlock16& b = blocks_[i];
    Level* Lin = lv[li];
    if (b.sh == 2) li++;
    t = run_lc16(c, b, t, *Lin, *lv[li], *fl[li]);
    for (int j = 0; j < 4; j++)
      if (tap_after_[j] == (int)i) {
        taps[j] = alloc16(c, *lv[li], out_[j].cout);
        conv_pw16(c, out_[j], t, *fl[li], taps[j], 0, epi16(out_[j], ACT_NONE));
        tap_lv[j] = lv[li]; tap_fl[j] = fl[li];
      }
  }
  // RSEFPN: in_j = ins_j(tap_j) * (1 + SE); top-down nearest-2x adds; p_j = inp_j(.) * (1 + SE); concat of the upsampled p_j
  H16 in[4];
  for (int j = 3; j >= 0; j--) {
    in[j] = alloc16(c, *tap_lv[j], 96);
    conv_pw16(c, ins_[j], taps[j], *tap_fl[j], in[j], 0, epi16(ins_[j], ACT_NONE));
    float* sc = se16(c, ins_se_[j], in[j], *tap_lv[j], HSIG_MBV3, 1);
    if (j == 3) scale16(c, in[j], *tap_lv[j], sc, nullptr, in[j]);
    else {
      ProfScope ps(c.prof, c.st, "upsample_add16");
      nh::upsample_add16(c.st, in[j].p, in[j + 1].p, tap_lv[j]->d, tap_lv[j + 1]->d, tap_lv[j]->n(), tap_lv[j]->maxPix, 96, in[j].p, sc);
    }
  }
  H16 fuse = alloc16(c, L4, 96);
  for (int j = 3; j >= 0; j--) {
    H16 p = alloc16(c, *tap_lv[j], 24);
    conv_sp16(c, inp_[j], in[j], *tap_lv[j], *tap_lv[j], 1, 1, p, 0, epi16(inp_[j], ACT_NONE));
    float* sc = se16(c, inp_se_[j], p, *tap_lv[j], HSIG_MBV3, 1);
    ProfScope ps(c.prof, c.st, "fpn_concat16");
    nh::upsample_into16(c.st, p.p, p.ld, tap_lv[j]->d, L4.d, L4.n(), L4.maxPix, 24, j, fuse.p, fuse.ld, (3 - j) * 24, sc);  // order p5, p4, p3, p2
  }
  H16 h1 = alloc16(c, L4, 24);
  conv_sp16(c, head_conv1_, fuse, L4, L4, 1, 1, h1, 0, epi16(head_conv1_, ACT_RELU));
  H16 d1 = alloc16(c, L4, 96);   // deconv1 as a 1x1 conv to (dy, dx, c)
  conv_pw16(c, dc1_, h1, F4, d1, 0, epi16(dc1_, ACT_RELU));
  H16 f = alloc16(c, L2, 24);
  { ProfScope ps(c.prof, c.st, "pixel_shuffle16");
    nh::pixel_shuffle16(c.st, d1.p, d1.ld, L4.d, L2.d, L2.n(), L2.maxPix, 24, f.p, f.ld, 0); }
  float* map = c.arena->alloc<float>((size_t)L0.total);
  { ProfScope ps(c.prof, c.st, "db_head_tail16");
    nh::deconv_to_map16(c.st, f.p, f.ld, L2.d, L0.d, L2.n(), L2.maxPix, 24, dc2_w_, dc2_b_, map); }
  return map;
}

// ---------------------------------------------------------------------------
// ClsNetH
// ---------------------------------------------------------------------------
struct ClsSpec16 { int k, mid, cout; bool se; int act, sh, sw; };
static const ClsSpec16 CLS_SPEC16[] = {
    {3, 8, 8, true, ACT_RELU, 2, 1},      {3, 24, 8, false, ACT_RELU, 2, 1},    {3, 32, 8, false, ACT_RELU, 1, 1},
    {5, 32, 16, true, ACT_HSWISH, 2, 1},  {5, 88, 16, true, ACT_HSWISH, 1, 1},  {5, 88, 16, true, ACT_HSWISH, 1, 1},
    {5, 40, 16, true, ACT_HSWISH, 1, 1},  {5, 48, 16, true, ACT_HSWISH, 1, 1},  {5, 104, 32, true, ACT_HSWISH, 2, 1},
    {5, 200, 32, true, ACT_HSWISH, 1, 1}, {5, 200, 32, true, ACT_HSWISH, 1, 1}};

ClsNetH::ClsNetH(const Blob& b) {
  stem_ = pack_conv16(ws_, b, "cls.stem", 8, 3, 3, 3);
  int cin = 8, i = 0;
  for (const ClsSpec16& s : CLS_SPEC16) {
    const std::string p = "cls.b" + std::to_string(i++);
    B blk;
    blk.expand = pack_conv16(ws_, b, p + ".expand", s.mid, cin, 1, 1);
    blk.dw = pack_dw16(ws_, b, p + ".dw", s.mid, s.k);
    blk.se = s.se;
    if (s.se) blk.sew = get_se16(ws_, b, p + ".se", s.mid);
    blk.linear = pack_conv16(ws_, b, p + ".linear", s.cout, s.mid, 1, 1);
    blk.act = s.act; blk.sh = s.sh; blk.sw = s.sw;
    blk.shortcut = (s.sh == 1 && s.sw == 1 && cin == s.cout);
    blocks_.push_back(blk);
    cin = s.cout;
  }
  conv2_ = pack_conv16(ws_, b, "cls.conv2", 200, cin, 1, 1);
  fc_ = pack_linear(ws_, b, "cls.head.fc", 200, 2);
}

float* ClsNetH::run(RunCtx& c, const float* x4, Level& L0) {
  std::vector<Level> lv; lv.reserve(16);
  lv.push_back(down_level(L0, 2, 2));
  for (const B& b : blocks_) lv.push_back(down_level(lv.back(), b.sh, b.sw));
  Level Lp = pool_level(lv.back(), 2, 2);
  std::vector<Level> fl; fl.reserve(16);
  for (auto& l : lv) fl.push_back(flat_level(l));
  std::vector<Level*> ups = {&L0};
  for (auto& l : lv) ups.push_back(&l);
  for (auto& l : fl) ups.push_back(&l);
  ups.push_back(&Lp);
  upload_levels(c, ups);
  H16 x = input_f32(c, x4, L0);
  H16 t = alloc16(c, lv[0], 8);
  conv_sp16(c, stem_, x, L0, lv[0], 2, 2, t, 0, epi16(stem_, ACT_HSWISH));
  for (size_t i = 0; i < blocks_.size(); i++) {
    const B& b = blocks_[i];
    const Level &Lin = lv[i], &Lout = lv[i + 1];
    H16 e = alloc16(c, Lin, b.expand.cout);
    conv_pw16(c, b.expand, t, fl[i], e, 0, epi16(b.expand, b.act));
    H16 d = alloc16(c, Lout, b.expand.cout);
    Lab nolab;
    dw16(c, b.dw, e, Lin, Lout, b.sh, b.sw, b.act, nolab, d);
    if (b.se) {
      float* s = se16(c, b.sew, d, Lout, HSIG_MBV3, 0);
      scale16(c, d, Lout, s, nullptr, d);
    }
    H16 y = alloc16(c, Lout, b.linear.cout);
    conv_pw16(c, b.linear, d, fl[i + 1], y, 0, epi16(b.linear, ACT_NONE, nullptr, b.shortcut ? t.p : nullptr, t.ld));
    t = y;
  }
  const Level& Ll = lv.back();
  H16 f = alloc16(c, Ll, 200);
  conv_pw16(c, conv2_, t, fl.back(), f, 0, epi16(conv2_, ACT_HSWISH));
  H16 mp = alloc16(c, Lp, 200);
  { ProfScope ps(c.prof, c.st, "maxpool16");
    nh::maxpool16(c.st, f.p, f.ld, Ll.d, Lp.d, Lp.n(), Lp.maxPix, 200, 2, 2, 2, 2, 0, 0, mp.p, mp.ld); }
  float* partial = c.arena->alloc<float>((size_t)Lp.n() * nh::pool_chunks16(Lp.maxPix) * 200);
  float* gm = c.arena->alloc<float>((size_t)Lp.n() * 200);
  { ProfScope ps(c.prof, c.st, "global_mean16");
    nh::global_mean16(c.st, mp.p, mp.ld, Lp.d, Lp.n(), Lp.maxPix, 200, partial, gm); }
  float* logits = c.arena->alloc<float>((size_t)Lp.n() * 4);
  float* probs = c.arena->alloc<float>((size_t)Lp.n() * 2);
  { ProfScope ps(c.prof, c.st, "gemm_cls");
    nn::gemm(c.st, gm, 200, Lp.n(), fc_.K, fc_.w, 2, fc_.Npad, logits, 4, 0, make_epi(fc_, ACT_NONE)); }
  { ProfScope ps(c.prof, c.st, "softmax");
    nn::softmax_rows(c.st, logits, 4, Lp.n(), 2, probs); }
  return probs;
}

// ---------------------------------------------------------------------------
// RecNeck16: avg-pool (3,2) -> EncoderWithSVTR -> CTC head; the 1x3 / 1x1 convs on C and 2C channels run in fp16,
// the mixing blocks on D = 120 and the CTC FC (fused argmax) in fp32 (SvtrCore)
// ---------------------------------------------------------------------------
void RecNeck16::load(WeightStore& ws, const Blob& b, const std::string& prefix, int C_) {
  C = C_;
  const int D = 120;
  conv1 = pack_conv16(ws, b, prefix + ".neck.conv1", C / 8, C, 1, 3);
  conv2 = pack_conv16(ws, b, prefix + ".neck.conv2", D, C / 8, 1, 1);
  conv3 = pack_conv16(ws, b, prefix + ".neck.conv3", C, D, 1, 1);
  conv4 = pack_conv16(ws, b, prefix + ".neck.conv4", C / 8, 2 * C, 1, 3);
  conv1x1 = pack_conv16(ws, b, prefix + ".neck.conv1x1", D, C / 8, 1, 1);
  core.load(ws, b, prefix);
}

float* RecNeck16::run(RunCtx& c, H16 t, const Level& Lb, Level& Lt, const Level& LtFlat, int* idx_out, float* prob_out) const {
  const long long rows = Lt.total;
  const int D = 120;
  H16 cat; cat.C = 2 * C; cat.ld = 2 * C;
  cat.p = c.arena->alloc<half_t>((size_t)std::max<long long>(rows, 1) * cat.ld);
  { ProfScope ps(c.prof, c.st, "avgpool16");
    nh::avgpool16(c.st, t.p, t.ld, Lb.d, Lt.d, Lt.n(), Lt.maxPix, C, 3, 2, cat.p, cat.ld); }
  H16 h = view16(cat, 0, C);
  H16 z1 = alloc16(c, Lt, C / 8);
  conv_sp16(c, conv1, h, Lt, Lt, 1, 1, z1, 0, epi16(conv1, ACT_SWISH));
  H16 z = alloc16(c, Lt, D);
  conv_pw16(c, conv2, z1, LtFlat, z, 0, epi16(conv2, ACT_SWISH));
  float* zf32 = c.arena->alloc<float>((size_t)std::max<long long>(rows, 1) * D);
  { ProfScope ps(c.prof, c.st, "f16_to_f32");
    nh::h_to_f32(c.st, z.p, z.ld, rows, D, zf32, D, 0); }
  float* zm = core.mixer(c, zf32, Lt);
  H16 zh = alloc16(c, Lt, D);
  { ProfScope ps(c.prof, c.st, "f32_to_f16");
    nh::f32_to_h(c.st, zm, D, rows, D, zh.p, zh.ld, 0); }
  conv_pw16(c, conv3, zh, LtFlat, cat, C, epi16(conv3, ACT_SWISH));
  H16 z4 = alloc16(c, Lt, C / 8);
  conv_sp16(c, conv4, cat, Lt, Lt, 1, 1, z4, 0, epi16(conv4, ACT_SWISH));
  H16 z5 = alloc16(c, Lt, D);
  conv_pw16(c, conv1x1, z4, LtFlat, z5, 0, epi16(conv1x1, ACT_SWISH));
  float* z5f = c.arena->alloc<float>((size_t)std::max<long long>(rows, 1) * D);
  { ProfScope ps(c.prof, c.st, "f16_to_f32");
    nh::h_to_f32(c.st, z5.p, z5.ld, rows, D, z5f, D, 0); }
  return core.head(c, z5f, rows, idx_out, prob_out);
}

// ---------------------------------------------------------------------------
// RecNetH: PP-OCRv4 mobile rec in fp16
// ---------------------------------------------------------------------------
RecNetH::RecNetH(const Blob& b) {
  stem_ = pack_conv16(ws_, b, "rec.stem", 16, 3, 3, 3);
  for (const LcSpec16& s : REC_SPEC16) blocks_.push_back(build_lc16(ws_, b, std::string("rec.") + s.name, s.k, s.cin, s.cout, s.sh, s.sw, s.se));
  neck_.load(ws_, b, "rec", 480);
}

float* RecNetH::run(RunCtx& c, const float* x4, Level& L0, Level& Lt, int* idx_out, float* prob_out) {
  for (auto& g : L0.h) if (g.H != 48 || g.W < 8) throw RtError(3, "rec input must be 48 high and at least 8 wide");
  std::vector<Level> lv; lv.reserve(16);
  lv.push_back(down_level(L0, 2, 2));
  for (const LcBlock16& b : blocks_) lv.push_back(down_level(lv.back(), b.sh, b.sw));
  Lt = pool_level(lv.back(), 3, 2);
  std::vector<Level> fl; fl.reserve(17);
  for (auto& l : lv) fl.push_back(flat_level(l));
  Level LtF = flat_level(Lt);
  std::vector<Level*> ups = {&L0, &Lt, &LtF};
  for (auto& l : lv) ups.push_back(&l);
  for (auto& l : fl) ups.push_back(&l);
  upload_levels(c, ups);
  H16 x = input_f32(c, x4, L0);
  H16 t = alloc16(c, lv[0], 16);
  conv_sp16(c, stem_, x, L0, lv[0], 2, 2, t, 0, epi16(stem_, ACT_NONE));
  for (size_t i = 0; i < blocks_.size(); i++) t = run_lc16(c, blocks_[i], t, lv[i], lv[i + 1], fl[i + 1]);
  return neck_.run(c, t, lv.back(), Lt, LtF, idx_out, prob_out);
}

// ---------------------------------------------------------------------------
// PPHGNet_small (rec_hgnet.py): stem 3 x ConvBNAct(3x3), [max-pool for det], 4 stages of
//   [depthwise 3x3 downsample] + HG_Block* ; HG_Block = 6 x ConvBNAct(3x3) whose outputs are concatenated with the
//   block input, 1x1 aggregation conv, ESE gate, residual when identity.
// Every layer output is written straight into the block's concat buffer (channel offset), nothing is copied.
// ---------------------------------------------------------------------------
void HgNet16::load(WeightStore& ws, const Blob& b, const std::string& prefix, bool det) {
  const int stem_c[3] = {64, 64, 128};
  int cin = 3;
  for (int i = 0; i < 3; i++) { stem[i] = pack_conv16(ws, b, prefix + ".stem" + std::to_string(i), stem_c[i], cin, 3, 3); cin = stem_c[i]; }
  struct St { const char* name; int cin, mid, cout, blocks; bool down; int sh, sw; };
  const St det_st[4] = {{"st1", 128, 128, 256, 1, false, 2, 2}, {"st2", 256, 160, 512, 1, true, 2, 2},
                        {"st3", 512, 192, 768, 2, true, 2, 2}, {"st4", 768, 224, 1024, 1, true, 2, 2}};
  const St rec_st[4] = {{"st1", 128, 128, 256, 1, true, 2, 1}, {"st2", 256, 160, 512, 1, true, 1, 2},
                        {"st3", 512, 192, 768, 2, true, 2, 1}, {"st4", 768, 224, 1024, 1, true, 2, 1}};
  for (const St& s : det ? det_st : rec_st) {
    HgStage16 st; st.down = s.down; st.sh = s.sh; st.sw = s.sw; st.cin = s.cin; st.cout = s.cout;
    const std::string p = prefix + "." + s.name;
    if (s.down) st.ds = pack_dw16(ws, b, p + ".ds", s.cin, 3);
    for (int k = 0; k < s.blocks; k++) {
      HgBlock16 blk; blk.cin = k == 0 ? s.cin : s.cout; blk.mid = s.mid; blk.cout = s.cout; blk.identity = k > 0;
      const std::string q = p + ".b" + std::to_string(k);
      int c = blk.cin;
      for (int l = 0; l < 6; l++) { blk.l[l] = pack_conv16(ws, b, q + ".l" + std::to_string(l), s.mid, c, 3, 3); c = s.mid; }
      blk.agg = pack_conv16(ws, b, q + ".agg", s.cout, blk.cin + 6 * s.mid, 1, 1);
      blk.ese = get_ese16(ws, b, q + ".ese", s.cout);
      st.blocks.push_back(blk);
    }
    stages.push_back(st);
  }
}

// Concat buffer of a block whose input has `cin` channels: [cin | 6 x mid]
static H16 hg_cat(RunCtx& c, const HgBlock16& b, const Level& L) {
  H16 t; t.C = b.cin + 6 * b.mid; t.ld = t.C;
  t.p = c.arena->alloc<half_t>((size_t)std::max<long long>(L.total, 1) * t.ld);
  return t;
}
// cat[:, :cin] already holds the block input; returns the block output written to `out` (a view, possibly of the next concat)
static void run_hg_block(RunCtx& c, const HgBlock16& b, H16 cat, const Level& L, const Level& F, H16 out) {
  int off = 0, cw = b.cin;
  for (int l = 0; l < 6; l++) {
    H16 in = view16(cat, off, cw);
    off += cw; cw = b.mid;
    conv_sp16(c, b.l[l], in, L, L, 1, 1, cat, off, epi16(b.l[l], ACT_RELU));
  }
  H16 t = alloc16(c, L, b.cout);
  conv_pw16(c, b.agg, cat, F, t, 0, epi16(b.agg, ACT_RELU));
  float* gate = se16(c, b.ese, t, L, GATE_SIGMOID, 0);
  H16 res = view16(cat, 0, b.cin);
  scale16(c, t, L, gate, b.identity ? &res : nullptr, out);
}

// Runs stem + stages from the [pix][8] input.  feat[i] = output of stage i.  Lpool (det only) is the level after the
// 3x3 stride-2 max-pool that follows the stem; Lst / Fst are the stage levels and their flat views.
struct HgRun { H16 feat[4]; };
static HgRun run_hgnet(RunCtx& c, const HgNet16& n, H16 x, const Level& L0, const Level& Lstem, const Level* Lpool,
                       const std::vector<Level*>& Lst, const std::vector<Level*>& Fst) {
  HgRun r;
  H16 s0 = alloc16(c, Lstem, 64), s1 = alloc16(c, Lstem, 64), s2 = alloc16(c, Lstem, 128);
  conv_sp16(c, n.stem[0], x, L0, Lstem, 2, 2, s0, 0, epi16(n.stem[0], ACT_RELU));
  conv_sp16(c, n.stem[1], s0, Lstem, Lstem, 1, 1, s1, 0, epi16(n.stem[1], ACT_RELU));
  conv_sp16(c, n.stem[2], s1, Lstem, Lstem, 1, 1, s2, 0, epi16(n.stem[2], ACT_RELU));
  H16 cur = s2;                 // tensor feeding the next stage, at level *Lcur
  const Level* Lcur = &Lstem;
  H16 ready_cat; bool have_cat = false;  // a stage without downsample finds its input already in its first concat buffer
  if (Lpool) {  // det: max-pool 3x3 s2 p1 straight into stage 1's concat (stage 1 has no downsample)
    ready_cat = hg_cat(c, n.stages[0].blocks[0], *Lpool);
    ProfScope ps(c.prof, c.st, "maxpool16");
    nh::maxpool16(c.st, cur.p, cur.ld, Lstem.d, Lpool->d, Lpool->n(), Lpool->maxPix, 128, 3, 3, 2, 2, 1, 1, ready_cat.p, ready_cat.ld);
    have_cat = true; Lcur = Lpool;
  }
  for (size_t si = 0; si < n.stages.size(); si++) {
    const HgStage16& st = n.stages[si];
    const Level& L = *Lst[si];
    const Level& F = *Fst[si];
    H16 cat;
    if (st.down) {
      cat = hg_cat(c, st.blocks[0], L);
      Lab nolab;
      dw16(c, st.ds, cur, *Lcur, L, st.sh, st.sw, ACT_NONE, nolab, view16(cat, 0, st.cin));
    } else {
      if (!have_cat) throw RtError(8, "hgnet: a stage without downsample needs its input in a concat buffer");
      cat = ready_cat;
    }
    have_cat = false;
    for (size_t k = 0; k < st.blocks.size(); k++) {
      const HgBlock16& b = st.blocks[k];
      const bool last = k + 1 == st.blocks.size();
      H16 out, next_cat;
      if (!last) { next_cat = hg_cat(c, st.blocks[k + 1], L); out = view16(next_cat, 0, b.cout); }  // the producer writes the next block's input in place
      else out = alloc16(c, L, b.cout);
      run_hg_block(c, b, cat, L, F, out);
      if (!last) cat = next_cat; else cur = out;
    }
    r.feat[si] = cur;
    Lcur = &L;
  }
  return r;
}

// ---------------------------------------------------------------------------
// DetServerH
// ---------------------------------------------------------------------------
DetServerH::DetServerH(const Blob& b) {
  bb_.load(ws_, b, "sdet", true);
  const int feat_c[4] = {256, 512, 768, 1024};
  for (int i = 0; i < 4; i++) {
    const std::string is = std::to_string(i);
    ins_[i] = pack_conv16(ws_, b, "sdet.neck.ins" + is, 256, feat_c[i], 1, 1);
    inp_[i] = pack_conv16(ws_, b, "sdet.neck.inp" + is, 64, 256, 9, 9);
    panlat_[i] = pack_conv16(ws_, b, "sdet.neck.panlat" + is, 64, 64, 9, 9);
    if (i < 3) panhead_[i] = pack_conv16(ws_, b, "sdet.neck.panhead" + is, 64, 64, 3, 3);
  }
  for (int i = 0; i < 4; i++) {  // IntraCLBlock: the k x k, k x 1 and 1 x k convs of one level see the same input -> one k x k conv
    const std::string p = "sdet.neck.incl" + std::to_string(i + 1);
    Incl& I = incl_[i];
    I.reduce = pack_conv16(ws_, b, p + ".reduce", 32, 64, 1, 1);
    Conv16* dst[3] = {&I.c7, &I.c5, &I.c3};
    const int ks[3] = {7, 5, 3};
    for (int q = 0; q < 3; q++) {
      const int k = ks[q], R = 32;
      const std::string ksx = std::to_string(k);
      const BlobTensor& wc = b.get(p + ".c" + ksx + ".w"); const BlobTensor& wv = b.get(p + ".v" + ksx + ".w");
      const BlobTensor& wq = b.get(p + ".q" + ksx + ".w");
      expect_dims16(wc, {R, R, k, k}, p + ".c.w"); expect_dims16(wv, {R, R, k, 1}, p + ".v.w"); expect_dims16(wq, {R, R, 1, k}, p + ".q.w");
      std::vector<float> bias(R);
      for (int n = 0; n < R; n++)
        bias[n] = b.get(p + ".c" + ksx + ".b").data[n] + b.get(p + ".v" + ksx + ".b").data[n] + b.get(p + ".q" + ksx + ".b").data[n];
      *dst[q] = upload_conv16(ws_, R, R, k, k, [&](int n, int ci, int dy, int dx) {
        float v = wc.data[(((size_t)n * R + ci) * k + dy) * k + dx];
        if (dx == k / 2) v += wv.data[((size_t)n * R + ci) * k + dy];
        if (dy == k / 2) v += wq.data[((size_t)n * R + ci) * k + dx];
        return v;
      }, bias.data());
    }
    I.ret = pack_conv16(ws_, b, p + ".ret", 64, 32, 1, 1);
  }
  head_conv1_ = pack_conv16(ws_, b, "sdet.head.conv1", 64, 256, 3, 3);
  dc1_ = pack_deconv16(ws_, b, "sdet.head.deconv1", 64, 64);
  dc2_w_ = upload_raw(ws_, b, "sdet.head.deconv2.w", 64 * 4);
  dc2_b_ = b.get("sdet.head.deconv2.b").data[0];
  // PFHeadLocal's LocalModule: 3x3 conv over concat[shrink map (1), up2(f) (64)] at full resolution, evaluated per output
  // phase (a, b) = (row & 1, col & 1) at HALF resolution: the taps that fall on the same pixel of f are summed (2x2 kernel over
  // f), the 3x3 window of the map comes in as 9 of the 16 "window" channels (nh::map_window16) at the tap that sits on (y, x).
  {
    const BlobTensor& w = b.get("sdet.head.local3.w");
    expect_dims16(w, {64, 65, 3, 3}, "sdet.head.local3.w");
    const BlobTensor& bt = b.get("sdet.head.local3.b");
    auto W = [&](int n, int ci, int dy, int dx) { return w.data[(((size_t)n * 65 + ci) * 3 + dy) * 3 + dx]; };
    for (int a = 0; a < 2; a++)
      for (int bb = 0; bb < 2; bb++) {
        // tap row r of the 2x2 kernel reads f row y + r + a - 1; full-res tap dy in {-1,0,1} lands on f row y + floor((a+dy)/2)
        auto fl2 = [](int v) { return v >= 0 ? v / 2 : -((-v + 1) / 2); };
        local_[2 * a + bb] = upload_conv16(ws_, 64, 80, 2, 2, [&](int n, int k, int r, int cc) {
          float v = 0.f;
          if (k < 64) {
            for (int dy = -1; dy <= 1; dy++)
              for (int dx = -1; dx <= 1; dx++)
                if (fl2(a + dy) - a + 1 == r && fl2(bb + dx) - bb + 1 == cc) v += W(n, 1 + k, dy + 1, dx + 1);
          } else if (r == 1 - a && cc == 1 - bb) {
            const int i = (k - 64) / 4, j = (k - 64) % 4;  // window pixel (2y - 1 + i, 2x - 1 + j)
            const int dy = i - a - 1, dx = j - bb - 1;
            if (dy >= -1 && dy <= 1 && dx >= -1 && dx <= 1) v = W(n, 0, dy + 1, dx + 1);
          }
          return v;
        }, bt.data);
      }
    const BlobTensor& w1 = b.get("sdet.head.local1.w");
    expect_dims16(w1, {1, 64, 1, 1}, "sdet.head.local1.w");
    std::vector<float> dw(64, 0.f);
    memcpy(dw.data(), w1.data, 64 * sizeof(float));
    local1_w_ = ws_.upload(dw);
    local1_b_ = b.get("sdet.head.local1.b").data[0];
  }
}

float* DetServerH::run(RunCtx& c, const float* x, Level& L0) { return forward(c, input_f32(c, x, L0), L0); }
float* DetServerH::run_u8(RunCtx& c, const nn::U8Page* pages, float scale, const float* mean3, const float* std3, Level& L0) {
  return forward(c, det_input_u8(c, pages, scale, mean3, std3, L0), L0);
}

float* DetServerH::forward(RunCtx& c, H16 x, Level& L0) {
  for (auto& g : L0.h)
    if (g.H % 32 != 0 || g.W % 32 != 0 || g.H == 0 || g.W == 0) throw RtError(3, "det input sides must be non-zero multiples of 32");
  Level L2 = down_level(L0, 2, 2), L4 = down_level(L2, 2, 2), L8 = down_level(L4, 2, 2), L16 = down_level(L8, 2, 2),
        L32 = down_level(L16, 2, 2);
  Level F2 = flat_level(L2), F4 = flat_level(L4), F8 = flat_level(L8), F16 = flat_level(L16), F32 = flat_level(L32);
  upload_levels(c, {&L0, &L2, &L4, &L8, &L16, &L32, &F2, &F4, &F8, &F16, &F32});
  Level* lv[4] = {&L4, &L8, &L16, &L32};
  Level* fl[4] = {&F4, &F8, &F16, &F32};
  HgRun bb = run_hgnet(c, bb_, x, L0, L2, &L4, {&L4, &L8, &L16, &L32}, {&F4, &F8, &F16, &F32});
  // ---- LKPAN ----
  H16 in[4];
  for (int j = 3; j >= 0; j--) {
    in[j] = alloc16(c, *lv[j], 256);
    conv_pw16(c, ins_[j], bb.feat[j], *fl[j], in[j], 0, epi16(ins_[j], ACT_NONE));
    if (j < 3) {
      ProfScope ps(c.prof, c.st, "upsample_add16");
      nh::upsample_add16(c.st, in[j].p, in[j + 1].p, lv[j]->d, lv[j + 1]->d, lv[j]->n(), lv[j]->maxPix, 256, in[j].p, nullptr);
    }
  }
  H16 f[4];
  for (int j = 3; j >= 0; j--) {
    f[j] = alloc16(c, *lv[j], 64);
    conv_sp16(c, inp_[j], in[j], *lv[j], *lv[j], 1, 1, f[j], 0, epi16(inp_[j], ACT_NONE));
  }
  // bottom-up path: pan_{j+1} = f_{j+1} + panhead_j(pan_j) (3x3 stride 2), pan_0 = f_0
  H16 pan[4]; pan[0] = f[0];
  for (int j = 0; j < 3; j++) {
    pan[j + 1] = alloc16(c, *lv[j + 1], 64);
    conv_sp16(c, panhead_[j], pan[j], *lv[j], *lv[j + 1], 2, 2, pan[j + 1], 0, epi16(panhead_[j], ACT_NONE, nullptr, f[j + 1].p, f[j + 1].ld));
  }
  H16 fuse = alloc16(c, L4, 256);
  for (int j = 0; j < 4; j++) {
    const Level& L = *lv[j];
    H16 p = alloc16(c, L, 64);
    conv_sp16(c, panlat_[j], pan[j], L, L, 1, 1, p, 0, epi16(panlat_[j], ACT_NONE));
    // IntraCLBlock(64, reduce 2): p + relu(bn(ret(c3(c5(c7(reduce(p)))))))
    const Incl& I = incl_[j];
    H16 r0 = alloc16(c, L, 32), r1 = alloc16(c, L, 32);
    conv_pw16(c, I.reduce, p, *fl[j], r0, 0, epi16(I.reduce, ACT_NONE));
    conv_sp16(c, I.c7, r0, L, L, 1, 1, r1, 0, epi16(I.c7, ACT_NONE));
    conv_sp16(c, I.c5, r1, L, L, 1, 1, r0, 0, epi16(I.c5, ACT_NONE));
    conv_sp16(c, I.c3, r0, L, L, 1, 1, r1, 0, epi16(I.c3, ACT_NONE));
    if (j == 0) {  // p2 is written straight into the concat
      conv_pw16(c, I.ret, r1, *fl[j], fuse, 192, epi16(I.ret, ACT_RELU, nullptr, p.p, p.ld));
    } else {
      H16 q = alloc16(c, L, 64);
      conv_pw16(c, I.ret, r1, *fl[j], q, 0, epi16(I.ret, ACT_RELU, nullptr, p.p, p.ld));
      ProfScope ps(c.prof, c.st, "fpn_concat16");
      nh::upsample_into16(c.st, q.p, q.ld, L.d, L4.d, L4.n(), L4.maxPix, 64, j, fuse.p, fuse.ld, (3 - j) * 64, nullptr);  // order p5, p4, p3, p2
    }
  }
  // ---- PFHeadLocal ----
  H16 h1 = alloc16(c, L4, 64);
  conv_sp16(c, head_conv1_, fuse, L4, L4, 1, 1, h1, 0, epi16(head_conv1_, ACT_RELU));
  H16 d1 = alloc16(c, L4, 256);
  conv_pw16(c, dc1_, h1, F4, d1, 0, epi16(dc1_, ACT_RELU));
  H16 g; g.C = 80; g.ld = 80;   // [f (64) | 4x4 window of the shrink map (16)] at half resolution
  g.p = c.arena->alloc<half_t>((size_t)L2.total * 80);
  { ProfScope ps(c.prof, c.st, "pixel_shuffle16");
    nh::pixel_shuffle16(c.st, d1.p, d1.ld, L4.d, L2.d, L2.n(), L2.maxPix, 64, g.p, g.ld, 0); }
  float* map = c.arena->alloc<float>((size_t)L0.total);
  { ProfScope ps(c.prof, c.st, "db_head_tail16");
    nh::deconv_to_map16(c.st, g.p, g.ld, L2.d, L0.d, L2.n(), L2.maxPix, 64, dc2_w_, dc2_b_, map); }
  { ProfScope ps(c.prof, c.st, "map_window16");
    nh::map_window16(c.st, map, L0.d, L2.d, L2.n(), L2.maxPix, g.p, g.ld, 64); }
  for (int a = 0; a < 2; a++)
    for (int b = 0; b < 2; b++) {
      const Conv16& w = local_[2 * a + b];
      Epi16 e = epi16(w, ACT_RELU);
      e.dot_w = local1_w_; e.dot_b = local1_b_; e.dot_map = map; e.gmap = L0.d; e.dot_py = a; e.dot_px = b;
      ProfScope ps(c.prof, c.st, "conv16_local", shp(L2.total, 4 * 80, 64, 2 * a + b));
      nh::conv16(c.st, g.p, g.ld, L2.d, L2.d, L2.n(), L2.maxH, L2.maxW, 80, 2, 2, 1, 1, 1 - a, 1 - b, w.w, 64, w.npad, nullptr, 8, 0, e);
    }
  return map;
}

// ---------------------------------------------------------------------------
// RecServerH
// ---------------------------------------------------------------------------
RecServerH::RecServerH(const Blob& b) {
  bb_.load(ws_, b, "srec", false);
  neck_.load(ws_, b, "srec", 1024);
}

float* RecServerH::run(RunCtx& c, const float* x4, Level& L0, Level& Lt, int* idx_out, float* prob_out) {
  for (auto& g : L0.h) if (g.H != 48 || g.W < 8) throw RtError(3, "rec input must be 48 high and at least 8 wide");
  Level Ls = down_level(L0, 2, 2);
  Level S1 = down_level(Ls, 2, 1), S2 = down_level(S1, 1, 2), S3 = down_level(S2, 2, 1), S4 = down_level(S3, 2, 1);
  Lt = pool_level(S4, 3, 2);
  Level F1 = flat_level(S1), F2 = flat_level(S2), F3 = flat_level(S3), F4 = flat_level(S4), LtF = flat_level(Lt);
  upload_levels(c, {&L0, &Ls, &S1, &S2, &S3, &S4, &Lt, &F1, &F2, &F3, &F4, &LtF});
  H16 x = input_f32(c, x4, L0);
  HgRun bb = run_hgnet(c, bb_, x, L0, Ls, nullptr, {&S1, &S2, &S3, &S4}, {&F1, &F2, &F3, &F4});
  return neck_.run(c, bb.feat[3], S4, Lt, LtF, idx_out, prob_out);
}

}  // namespace rt
