// fp16-storage / fp32-accumulate kernel family (nn_f16.hip), selected by rt_config.dtype = RT_DTYPE_F16
// and used by every network of nets_f16.cpp: the PP-OCRv4 mobile graphs in half precision and the
// PP-OCRv4 server graphs (PPHGNet_small), BASELINE.json config 5.
//
// Activations are fp16 NHWC with a channel pitch that is a multiple of 8 (16-byte vectors); batches are
// ragged lists of images described by rt::ImgGeom tables exactly as in the fp32 family (nn.h).  All
// contractions run on v_mfma_f32_32x32x16_f16 with fp32 accumulators; bias / activation / LAB / residual
// are applied in fp32 before the single rounding to fp16 on store.
#pragma once
#include "common.h"

namespace rt {
namespace nh {

typedef _Float16 half_t;

constexpr int KS = 32;  // channels per K slab of the conv kernel (weights are packed in KS slabs)

static inline int pitch8(int c) { return round_up(c, 8); }

// Fused conv epilogue, all in fp32: v = acc + bias[n]; v = act(v); v = lab_a * v + lab_c (has_lab);
// v += residual[pix][n] (fp16 tensor); then one of
//   store   y[pix][coff + n] = (half) v
//   dot     (EPI_DOT, the PFHeadLocal tail) s[pix] = sum_n v[n] * dot_w[n] + dot_b over ALL N channels of the block,
//           map[opix] = 0.5 * (map[opix] + sigmoid(s)) with opix = the full-resolution pixel of phase (dot_py, dot_px)
struct Epi16 {
  const float* bias = nullptr;  // [Npad] or null
  int act = ACT_NONE;
  int has_lab = 0;
  float lab_a = 1.f, lab_c = 0.f;
  const half_t* residual = nullptr;
  int ld_res = 0;
  const float* dot_w = nullptr;  // [Npad]: enables the dot epilogue
  float dot_b = 0.f;
  float* dot_map = nullptr;      // fp32 map at 2x the output resolution (one float per pixel, image-major like gmap)
  const ImgGeom* gmap = nullptr;
  int dot_py = 0, dot_px = 0;
};

// Dense convolution as implicit GEMM.  x: [pixels][ldx] halves, Cin channels read (multiple of 8, <= ldx);
// kernel (KH, KW), stride (SH, SW), zero padding (PT, PL) at the top / left (the bottom / right padding is whatever the
// output geometry implies); weights packed [ceil(Cin/32)][KH][KW][Npad][32] (pack_conv16), Npad = N rounded up to 32.
// Output geometry gout (per image) decides the spatial extent; y: [pixels][ldy] at channel offset coff.
// A 1x1 conv over a whole batch can be launched as one image of H = 1, W = total pixels.
void conv16(hipStream_t st, const half_t* x, int ldx, const ImgGeom* gin, const ImgGeom* gout, int n_img, int maxHo, int maxWo,
            int Cin, int KH, int KW, int SH, int SW, int PT, int PL, const half_t* Wp, int N, int Npad, half_t* y, int ldy, int coff,
            const Epi16& epi);
// same contraction as a plain GEMM over rows: y[M][ldy] = epi(x[M][ldx] (K = Cin) . W); geometry tables are built on device
// memory the caller provides (one ImgGeom: {0, 1, M})
const char* conv16_label(int KH, int KW, int N, int Cin = 32);
bool conv_stamps_compiled();   // true in a `make STAMPS=1` build
extern long long* g_conv_stamps;  // diagnostics: when set, one workgroup of every conv16 launch records s_memtime per stage

// Depthwise KxK (K odd), stride (sh, sw), pad K/2.  Wd packed [K*K][Cp] halves, bias fp32 [Cp].
void dwconv16(hipStream_t st, int K, int sh, int sw, const half_t* x, int ldx, const ImgGeom* gin, const ImgGeom* gout, int n_img,
              int maxHo, int maxWo, int Cp, const half_t* Wd, const float* bias, int act, int has_lab, float lab_a, float lab_c,
              half_t* y, int ldy);

// ---- layout / dtype conversion -------------------------------------------------------------------------------------
// One RGB8 page of a det launch group (same descriptor as nn::U8Page): (x * scale - mean[c]) / std[c] in BGR order,
// written as [pix][8] halves (channels 3..7 zero)
struct U8Page16 { const uint8_t* rgb; long long npix; long long out_pix; };
void u8_to_h8(hipStream_t st, const U8Page16* pages, int n, long long max_pix, float scale, const float* mean3, const float* std3,
              half_t* out);
// fp32 NHWC pitch-4 -> fp16 pitch-8 (channels 4..7 zero)
void f32x4_to_h8(hipStream_t st, const float* in, long long npix, half_t* out);
// fp16 [rows][lds] (first C channels) -> fp32 [rows][ldd] at channel offset coff
void h_to_f32(hipStream_t st, const half_t* src, int lds, long long rows, int C, float* dst, int ldd, int coff);
void f32_to_h(hipStream_t st, const float* src, int lds, long long rows, int C, half_t* dst, int ldd, int coff);

// ---- squeeze-excite / ESE -------------------------------------------------------------------------------------------
// Deterministic two-stage spatial mean, then: hid = relu(w1t . mean + b1) (skipped when w1t == null: hid = mean, Cr = C),
// s = w2t . hid + b2, gate = hardsigmoid(slope) or sigmoid (slope < 0), + 1 when residual.  w1t [C][Cr], w2t [Cr][C]
// (transposed so that consecutive threads read consecutive floats).  scale: fp32 [n_img][Cp].
int pool_chunks16(long long max_pix);
void se_scale16(hipStream_t st, const half_t* x, int ldx, const ImgGeom* geom, int n_img, long long max_pix, int C, int Cp, const float* w1t,
                const float* b1, const float* w2t, const float* b2, int Cr, float slope, int residual, float* partial,
                float* scale);
// gate of a squeeze-excite: scale[img][c] = hardsigmoid(slope) or sigmoid (slope < 0) of s[img][c] (pitch lds), + 1 when residual
void gate16(hipStream_t st, const float* s, int lds, int n_img, int C, int Cp, float slope, int residual, float* scale);
// y = x * scale[image] (+ res); in place allowed; x / res / y have their own channel pitches (views into concat buffers)
void scale_channels16(hipStream_t st, const half_t* x, int ldx, const ImgGeom* geom, int n_img, long long max_pix, int Cp, const float* scale,
                      const half_t* res, int ldr, half_t* y, int ldy);
// global mean only: out fp32 [n_img][Cp]
void global_mean16(hipStream_t st, const half_t* x, int ldx, const ImgGeom* geom, int n_img, long long max_pix, int Cp, float* partial,
                   float* out);

// ---- spatial glue ---------------------------------------------------------------------------------------------------
// out = a * scale_a[image] + nearest_up2(b)   (scale_a optional; in place on a allowed)
void upsample_add16(hipStream_t st, const half_t* a, const half_t* b, const ImgGeom* ga, const ImgGeom* gb, int n_img,
                    long long max_pix, int Cp, half_t* out, const float* scale_a);
// dst[pix][coff .. coff + C) = src[pix >> shift][0 .. C) * scale[image]   (nearest upsample by 1 << shift into a concat)
void upsample_into16(hipStream_t st, const half_t* src, int lds, const ImgGeom* gsrc, const ImgGeom* gdst, int n_img,
                     long long max_pix, int C, int shift, half_t* dst, int ldd, int coff, const float* scale);
// out = a + b (same geometry, pitch Cp)
void add16(hipStream_t st, const half_t* a, const half_t* b, long long n_halves, half_t* out);
// max / average pooling, window (kh, kw), stride (sh, sw), pad (ph, pw) (max: -inf padding; avg: no padding allowed)
void maxpool16(hipStream_t st, const half_t* x, int ldx, const ImgGeom* gin, const ImgGeom* gout, int n_img, long long max_pix, int Cp,
               int kh, int kw, int sh, int sw, int ph, int pw, half_t* y, int ldy);
void avgpool16(hipStream_t st, const half_t* x, int ldx, const ImgGeom* gin, const ImgGeom* gout, int n_img, long long max_pix, int Cp,
               int kh, int kw, half_t* y, int ldy);   // window = stride, no padding
void avgpool16_to_f32(hipStream_t st, const half_t* x, const ImgGeom* gin, const ImgGeom* gout, int n_img, long long max_pix, int C,
                      int Cp, int kh, int kw, float* y, int ldy);
// ConvTranspose 2x2 stride 2 = 1x1 conv to 4 * C channels (conv16, columns ordered (dy, dx, c)) + this pixel shuffle:
// dst[(2y+dy, 2x+dx)][coff + c] = src[(y, x)][(dy * 2 + dx) * C + c]
void pixel_shuffle16(hipStream_t st, const half_t* src, int lds, const ImgGeom* gsrc, const ImgGeom* gdst, int n_img,
                     long long max_pix, int C, half_t* dst, int ldd, int coff);
// DB head tail: map[(2y+dy, 2x+dx)] = sigmoid(b + sum_c f[(y,x)][c] * w[c][dy*2+dx])  (ConvTranspose 2x2 s2, C -> 1)
void deconv_to_map16(hipStream_t st, const half_t* f, int ldf, const ImgGeom* gf, const ImgGeom* gmap, int n_img, long long max_pix,
                     int C, const float* w, float b, float* map);
// PFHeadLocal: the 4 x 4 neighbourhood of full-resolution map pixels around every half-resolution pixel,
// S[(y, x)][coff + 4 * i + j] = map[(2y - 1 + i, 2x - 1 + j)] (0 outside), as 16 extra fp16 channels of the feature tensor
void map_window16(hipStream_t st, const float* map, const ImgGeom* gmap, const ImgGeom* gf, int n_img, long long max_pix,
                  half_t* dst, int ldd, int coff);

}  // namespace nh
}  // namespace rt
