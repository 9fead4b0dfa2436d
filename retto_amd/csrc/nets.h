// The three networks behind RettoInnerWorker::{det,cls,rec}
// (/root/reference/retto-core/src/worker.rs:69-73), built from RTWB blobs and run as
// sequences of nn:: kernel launches over ragged NHWC batches.  Architecture: PP-OCRv4
// mobile det (PPLCNetV3 x0.75 + RSEFPN + DBHead), ch_ppocr_mobile_v2.0 cls
// (MobileNetV3-small x0.35) and PP-OCRv4 rec (PPLCNetV3 x0.95 + SVTR neck + CTC head);
// see SURVEY.md Appendix C.
#pragma once
#include <memory>
#include <string>
#include <map>
#include <vector>

#include "nn.h"
#include "runtime.h"

namespace rt {

// A pyramid level of a ragged batch: host copy + device copy of the per-image geometry.
struct Level {
  std::vector<ImgGeom> h;
  const ImgGeom* d = nullptr;
  long long total = 0;    // pixels over all images
  int maxH = 0, maxW = 0;
  long long maxPix = 0;   // max H*W over images
  int n() const { return (int)h.size(); }
  // row tile -> image tables of the SE-scaled GEMMs (nets.cpp run_lc), by tile height: built and uploaded once per pass
  mutable std::map<int, const int*> a_tabs;
};

struct RunCtx {
  hipStream_t st;
  Arena* arena;     // activations + descriptor tables of this pass
  Pinned* pinned;   // host staging for descriptor uploads
  Profiler* prof;
};

Level make_level(const std::vector<std::pair<int, int>>& hw);
Level flat_level(const Level& in);                             // one "image" of 1 x total pixels: a 1x1 conv as a GEMM over rows
Level down_level(const Level& in, int sh, int sw);             // conv k odd, pad k/2
Level pool_level(const Level& in, int kh, int kw);             // stride = kernel, no pad
void upload_levels(RunCtx& c, std::vector<Level*> levels);     // one H2D copy for all tables

struct PackedDense {  // gemm / conv_sp weights on device
  float* w = nullptr; float* b = nullptr;
  int K = 0, N = 0, Npad = 0, kh = 1, kw = 1;
};
struct PackedDw { float* w = nullptr; float* b = nullptr; int k = 3, C = 0, Cp = 0; };
struct Lab { int has = 0; float a = 1.f, c = 0.f; };
struct SeW { float *w1 = nullptr, *b1 = nullptr, *w2 = nullptr, *b2 = nullptr; int C = 0, Cr = 0; };

class WeightStore {  // owns one device allocation per network
 public:
  ~WeightStore();
  float* upload(const std::vector<float>& host);
  void* upload_bytes(const void* host, size_t bytes);
  size_t bytes() const { return total_; }
 private:
  std::vector<void*> bufs_;
  size_t total_ = 0;
};

PackedDense pack_conv(WeightStore& ws, const Blob& b, const std::string& name, int cout, int cin, int kh, int kw);
PackedDense pack_linear(WeightStore& ws, const Blob& b, const std::string& name, int cin, int cout);
float* upload_raw(WeightStore& ws, const Blob& b, const std::string& name, size_t expect_numel);
Epilogue make_epi(const PackedDense& p, int act, const Lab* lab = nullptr, const float* residual = nullptr, int ld_res = 0);

struct LcBlock {
  PackedDw dw; Lab dw_lab; int dw_act = 0;
  bool se = false; SeW sew;
  PackedDense pw; Lab pw_lab;
  int sh = 1, sw = 1, cin = 0, cout = 0;
};

// The three worker functions as interfaces (RettoInnerWorker::{det,cls,rec}, worker.rs:69-73): the fp32 mobile
// networks below and the fp16 mobile / server networks of nets_f16.h implement them; the session only sees these.
class DetModel {
 public:
  virtual ~DetModel() {}
  // x: f32 NHWC pitch-4 (B,G,R,0) at level L0 (every H,W a multiple of 32).
  // Returns the probability maps, one float per L0 pixel (arena memory).
  virtual float* run(RunCtx& c, const float* x, Level& L0) = 0;
  // Same from the RGB8 pages themselves (device descriptors, one per image of L0): the normalisation of
  // DetProcessor::preprocess is folded into the first kernel, the f32 input tensor is never built.
  virtual float* run_u8(RunCtx& c, const nn::U8Page* pages, float scale, const float* mean3, const float* std3, Level& L0) = 0;
  virtual const char* arch() const = 0;   // "mobile" / "server"
  virtual const char* dtype() const = 0;  // "f32" / "f16"
  virtual size_t weight_bytes() const = 0;
};
class ClsModel {
 public:
  virtual ~ClsModel() {}
  // x: f32 NHWC pitch-4, n images of 48 x 192. Returns softmax probs [n,2] (arena).
  virtual float* run(RunCtx& c, const float* x, Level& L0) = 0;
  virtual const char* dtype() const = 0;
};
class RecModel {
 public:
  virtual ~RecModel() {}
  virtual int classes() const = 0;
  int logits_ld() const { return round_up(classes(), 4); }
  // x: f32 NHWC pitch-4 (R,G,B,0), level L0 = lines of height 48; Lt (out) is the token level (H=1, W=T_i).
  // Returns the logits [Lt.total, logits_ld()], or -- with idx_out / prob_out (Lt.total each) -- runs the fused
  // CTC head (argmax + softmax probability of the argmax per time step, no logits in HBM) and returns nullptr.
  virtual float* run(RunCtx& c, const float* x, Level& L0, Level& Lt, int* idx_out = nullptr, float* prob_out = nullptr) = 0;
  virtual const char* arch() const = 0;
  virtual const char* dtype() const = 0;
  virtual size_t weight_bytes() const = 0;
};

// The part of the recognition head that is the same in every variant (mobile / server, fp32 / fp16): the two
// SVTR mixing blocks + final norm on D = 120 channels, and the CTC FC with its fused argmax epilogue.  fp32.
struct SvtrCore {
  struct Blk { PackedDense qkv, proj, fc1, fc2; float *n1g, *n1b, *n2g, *n2b; } blk[2];
  float *ng = nullptr, *nb = nullptr;
  PackedDense fc;
  int classes = 0, D = 120;
  void load(WeightStore& ws, const Blob& b, const std::string& prefix);  // <prefix>.neck.blk*, .neck.norm, .head.fc
  // z [rows, D] -> LN(blocks(z)) [rows, D]
  float* mixer(RunCtx& c, float* z, const Level& Lt) const;
  // z5 [rows, D] -> fused argmax (idx_out / prob_out) and nullptr, or the logits [rows, round_up(classes, 4)]
  float* head(RunCtx& c, const float* z5, long long rows, int* idx_out, float* prob_out) const;
};

class DetNet : public DetModel {
 public:
  explicit DetNet(const Blob& b);
  float* run(RunCtx& c, const float* x, Level& L0) override { return run(c, x, L0, nullptr, 0.f, nullptr, nullptr); }
  float* run_u8(RunCtx& c, const nn::U8Page* pages, float scale, const float* mean3, const float* std3, Level& L0) override {
    return run(c, nullptr, L0, pages, scale, mean3, std3);
  }
  const char* arch() const override { return "mobile"; }
  const char* dtype() const override { return "f32"; }
  size_t weight_bytes() const override { return ws_.bytes(); }
 private:
  float* run(RunCtx& c, const float* x, Level& L0, const nn::U8Page* pages, float scale, const float* mean3, const float* std3);
  WeightStore ws_;
  float* stem_w_; float* stem_b_;
  std::vector<LcBlock> blocks_;
  int tap_after_[4];
  PackedDense out_[4], ins_[4], inp_[4], head_conv1_;
  float* ins_lin_[4];  // lateral 1x1 weights as [cin][96] for the fused lateral + top-down add
  bool has_bias_[4] = {false, false, false, false};
  SeW ins_se_[4], inp_se_[4];
  float *dc1_w_, *dc1_b_, *dc2_w_, *dc2_b_;
  // nn_fpn.hip operands: head conv split by level (p2 fine / p3 phase / p4, p5 class), inp0 / inp1 phase weights + [tap][n][m] form
  float *head_wf_ = nullptr, *head_wc_ = nullptr, *head_cls4_ = nullptr, *head_cls5_ = nullptr;
  float *inp_wc_[2] = {nullptr, nullptr}, *inp_wm_[2] = {nullptr, nullptr};
};

class RecNet : public RecModel {
 public:
  explicit RecNet(const Blob& b);
  int classes() const override { return core_.classes; }
  float* run(RunCtx& c, const float* x, Level& L0, Level& Lt, int* idx_out = nullptr, float* prob_out = nullptr) override;
  const char* arch() const override { return "mobile"; }
  const char* dtype() const override { return "f32"; }
  size_t weight_bytes() const override { return ws_.bytes(); }
  // time steps of a line of width w (stem stride 2, one (1,2) stage, avg-pool 2): the same for the mobile and server graphs
  static int tokens_for_width(int w);
 private:
  WeightStore ws_;
  float* stem_w_; float* stem_b_;
  std::vector<LcBlock> blocks_;
  PackedDense conv1_, conv2_, conv3_, conv4_, conv1x1_;
  SvtrCore core_;
};

class ClsNet : public ClsModel {
 public:
  explicit ClsNet(const Blob& b);
  float* run(RunCtx& c, const float* x, Level& L0) override;
  const char* dtype() const override { return "f32"; }
 private:
  struct B { PackedDense expand, linear; PackedDw dw; bool se; SeW sew; int act, sh, sw; bool shortcut; };
  WeightStore ws_;
  float* stem_w_; float* stem_b_;
  std::vector<B> blocks_;
  PackedDense conv2_, fc_;
};

}  // namespace rt
