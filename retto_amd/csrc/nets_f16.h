// The worker networks in half precision (rt_config.dtype = RT_DTYPE_F16), built on the nn_f16 kernel family:
//   * the PP-OCRv4 mobile det / cls / rec graphs of nets.h with fp16 activations and weights (same RTWB blobs);
//   * the PP-OCRv4 server graphs of BASELINE.json config 5 -- det: PPHGNet_small + LKPAN(256, intracl) + PFHeadLocal,
//     rec: PPHGNet_small + the SVTR neck / CTC head on 1024 channels (SURVEY.md Appendix C "Server"; tensors sdet.* / srec.*,
//     retto_amd/synth.py).  The server graphs exist in fp16 only.
// Everything a network returns to the session is fp32 (probability map, class probabilities, argmax / probability
// per time step): the discrete stages behind it are the same bit-exact kernels as in the fp32 build.
#pragma once
#include "nets.h"
#include "nn_f16.h"

namespace rt {

struct Conv16 {  // conv16 weights on device: [ceil(cin_p/32)][kh][kw][npad][32] halves + fp32 bias [npad]
  nh::half_t* w = nullptr; float* b = nullptr;
  int cin = 0, cin_p = 0, cout = 0, npad = 0, kh = 1, kw = 1;
};
struct Dw16 { nh::half_t* w = nullptr; float* b = nullptr; int k = 3, C = 0, Cp = 0; };
struct Se16 { PackedDense fc1, fc2; bool has_fc1 = false; int C = 0, Cr = 0; };  // fp32 FCs on the channel means (nn::gemm)
struct H16 { nh::half_t* p = nullptr; int C = 0, ld = 0; };  // view of an fp16 NHWC tensor: C channels at pitch ld

struct LcBlock16 {
  Dw16 dw; Lab dw_lab; int dw_act = 0;
  bool se = false; Se16 sew;
  Conv16 pw; Lab pw_lab;
  int sh = 1, sw = 1, cin = 0, cout = 0;
};

class DetNetH : public DetModel {  // PP-OCRv4 mobile det, fp16
 public:
  explicit DetNetH(const Blob& b);
  float* run(RunCtx& c, const float* x, Level& L0) override;
  float* run_u8(RunCtx& c, const nn::U8Page* pages, float scale, const float* mean3, const float* std3, Level& L0) override;
  const char* arch() const override { return "mobile"; }
  const char* dtype() const override { return "f16"; }
  size_t weight_bytes() const override { return ws_.bytes(); }
 private:
  float* forward(RunCtx& c, H16 x, Level& L0);
  WeightStore ws_;
  Conv16 stem_;
  std::vector<LcBlock16> blocks_;
  int tap_after_[4];
  Conv16 out_[4], ins_[4], inp_[4], head_conv1_, dc1_;
  Se16 ins_se_[4], inp_se_[4];
  float* dc2_w_ = nullptr; float dc2_b_ = 0.f;
};

class ClsNetH : public ClsModel {  // ch_ppocr_mobile_v2.0 cls, fp16
 public:
  explicit ClsNetH(const Blob& b);
  float* run(RunCtx& c, const float* x, Level& L0) override;
  const char* dtype() const override { return "f16"; }
 private:
  struct B { Conv16 expand, linear; Dw16 dw; bool se; Se16 sew; int act, sh, sw; bool shortcut; };
  WeightStore ws_;
  Conv16 stem_, conv2_;
  std::vector<B> blocks_;
  PackedDense fc_;
};

// SVTR neck convs in fp16 around the fp32 SvtrCore; shared by the mobile and the server recognition nets
struct RecNeck16 {
  Conv16 conv1, conv2, conv3, conv4, conv1x1;
  SvtrCore core;
  int C = 0;  // backbone channels (480 mobile, 1024 server)
  void load(WeightStore& ws, const Blob& b, const std::string& prefix, int C);
  // t: backbone output at level Lb (H = 3 rows); pools to Lt and runs neck + head
  float* run(RunCtx& c, H16 t, const Level& Lb, Level& Lt, const Level& LtFlat, int* idx_out, float* prob_out) const;
};

class RecNetH : public RecModel {  // PP-OCRv4 mobile rec, fp16
 public:
  explicit RecNetH(const Blob& b);
  int classes() const override { return neck_.core.classes; }
  float* run(RunCtx& c, const float* x, Level& L0, Level& Lt, int* idx_out = nullptr, float* prob_out = nullptr) override;
  const char* arch() const override { return "mobile"; }
  const char* dtype() const override { return "f16"; }
  size_t weight_bytes() const override { return ws_.bytes(); }
 private:
  WeightStore ws_;
  Conv16 stem_;
  std::vector<LcBlock16> blocks_;
  RecNeck16 neck_;
};

// ---- PPHGNet_small -----------------------------------------------------------------------------------------------------
struct HgBlock16 { Conv16 l[6]; Conv16 agg; Se16 ese; int cin = 0, mid = 0, cout = 0; bool identity = false; };
struct HgStage16 { bool down = false; Dw16 ds; int sh = 1, sw = 1; std::vector<HgBlock16> blocks; int cin = 0, cout = 0; };
struct HgNet16 {
  Conv16 stem[3];
  std::vector<HgStage16> stages;
  void load(WeightStore& ws, const Blob& b, const std::string& prefix, bool det);
};

class DetServerH : public DetModel {  // PP-OCRv4 server det: PPHGNet_small + LKPAN + PFHeadLocal
 public:
  explicit DetServerH(const Blob& b);
  float* run(RunCtx& c, const float* x, Level& L0) override;
  float* run_u8(RunCtx& c, const nn::U8Page* pages, float scale, const float* mean3, const float* std3, Level& L0) override;
  const char* arch() const override { return "server"; }
  const char* dtype() const override { return "f16"; }
  size_t weight_bytes() const override { return ws_.bytes(); }
 private:
  float* forward(RunCtx& c, H16 x, Level& L0);
  WeightStore ws_;
  HgNet16 bb_;
  Conv16 ins_[4], inp_[4], panlat_[4], panhead_[3];
  struct Incl { Conv16 reduce, c7, c5, c3, ret; } incl_[4];  // c*: the kxk + kx1 + 1xk branches folded into one kxk conv
  Conv16 head_conv1_, dc1_, local_[4];                       // local_[2a+b]: PFHeadLocal's 3x3 on up2(f) as the 2x2 conv of phase (a,b)
  float* dc2_w_ = nullptr; float dc2_b_ = 0.f;
  float* local1_w_ = nullptr; float local1_b_ = 0.f;
};

class RecServerH : public RecModel {  // PP-OCRv4 server rec: PPHGNet_small + SVTR neck + CTC head
 public:
  explicit RecServerH(const Blob& b);
  int classes() const override { return neck_.core.classes; }
  float* run(RunCtx& c, const float* x, Level& L0, Level& Lt, int* idx_out = nullptr, float* prob_out = nullptr) override;
  const char* arch() const override { return "server"; }
  const char* dtype() const override { return "f16"; }
  size_t weight_bytes() const override { return ws_.bytes(); }
 private:
  WeightStore ws_;
  HgNet16 bb_;
  RecNeck16 neck_;
};

// Which graph a blob holds, by its tensor names.
bool blob_is_server_det(const Blob& b);
bool blob_is_server_rec(const Blob& b);

}  // namespace rt
