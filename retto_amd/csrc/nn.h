// Launchers of the neural-network kernels (nn_kernels.hip).  All activations are
// fp32 NHWC with a channel pitch that is a multiple of 4; batches are ragged lists of
// images described by device arrays of rt::ImgGeom.
#pragma once
#include "common.h"

namespace rt {
namespace nn {

constexpr int KC = 32;  // K-chunk of the MFMA GEMM / conv kernels (weights are packed in KC slabs)

// C[M, ldc] (+coff) = epi(A[M, lda] x W), W packed as [ceil(K/KC)][Npad16][KC].
void gemm(hipStream_t st, const float* A, int lda, long long M, int K, const float* Wp, int N, int Npad16, float* C,
          int ldc, int coff, const Epilogue& epi);
const char* gemm_pw_label(long long M, int Npad16, bool a_scale = false, int se_tile_rows = 0);   // se_tile_rows: gemm_se_tile_rows() of the layer
// Persistent LDS-DMA form of the 256 x 240 tile (nn_gemm_dma.hip): N a multiple of 240, K whole 16-deep groups, plain
// bias / activation / LAB epilogue.  gemm() takes it for the large 240- / 480-channel layers.
// k_gemm32w: K = N = 128 with the weights resident in LDS (nn_gemm_dma.hip)
bool gemm_w_supported(int lda, long long M, int K, int N, int Npad16, const Epilogue& epi, int ldc = 0, int coff = 0);
void gemm_w(hipStream_t st, const float* A, int lda, long long M, int K, const float* Wp, int N, int Npad16, float* C, int ldc, int coff,
            const Epilogue& epi);
bool gemm_dma_supported(int lda, long long M, int K, int N, int Npad16, const Epilogue& epi);
void gemm_dma(hipStream_t st, const float* A, int lda, long long M, int K, const float* Wp, int N, int Npad16, float* C,
              int ldc, int coff, const Epilogue& epi);
// Split-bf16 form of the same layers (nn_gemm_split.hip): three bf16 planes per operand, six v_mfma_f32_16x16x32_bf16 products
// per fp32 product, fp32 accumulation.  Opt-in: g_gemm_split (RT_GEMM_SPLIT=1, rt_debug_set_variants flag bit 12).
extern int g_gemm_split;
bool gemm_split_supported(int lda, long long M, int K, int N, int Npad16, const Epilogue& epi);
void gemm_split(hipStream_t st, const float* A, int lda, long long M, int K, const float* Wp, int N, int Npad16, float* C,
                int ldc, int coff, const Epilogue& epi);
void gemm_split_forget(const float* Wp);   // drops the cached split planes of a weight pack (before its memory is freed / reused)
void set_dw_xcd(int v);  // A/B: XCD-aware block order of the depthwise kernel (default on)
extern int g_dw_wide_slab_min, g_dw_wide3_min, g_dw_wide_lp;
extern int g_dw_variant;    // same for dwconv
extern int g_gemm_variant;  // kernel micro-benchmark hook (0 = production dispatch)
extern int g_gemm_dma;      // A/B: persistent LDS-DMA wide GEMM on (default) / off
extern int g_dw_sweep;      // A/B: column-sweep 5x5 depthwise kernel: pixels per thread (0: off)

// Dense stride-1 "same" convolution, kernel (KH,KW) in {(3,3),(1,3)}; W packed as
// [ceil(Cin/KC)][KH*KW][Npad16][KC].
// 1x3 conv over the flat token list of a level of text lines (H = 1 images, contiguous rows): flags[t] bit 0 = token t is the first
// of its line, bit 1 = the last (nn_kernels.hip k_conv13_flat).  N <= 64.
bool conv13_flat_supported(int N, int Npad16);
void conv13_flat(hipStream_t st, const float* x, int ldx, long long rows, const unsigned char* flags, int Cin, const float* Wp, int N,
                 int Npad16, float* y, int ldy, const Epilogue& epi);
void conv_sp(hipStream_t st, int KH, int KW, const float* x, int ldx, const ImgGeom* geom, int n_img, int maxH,
             int maxW, int Cin, const float* Wp, int N, int Npad16, float* y, int ldy, const Epilogue& epi);

// Depthwise KxK (K in {3,5}), stride (sh,sw), pad K/2.  Wd packed [K*K][Cp].
void dwconv(hipStream_t st, int K, int sh, int sw, const float* x, const ImgGeom* gin, const ImgGeom* gout, int n_img,
            int maxHo, int maxWo, int Cp, int C, const float* Wd, const float* bias, int act, int has_lab,
            float lab_a, float lab_c, float* y, float* pool = nullptr);  // Cp = channel pitch (chan_pitch), C = real channels
// Fused squeeze-excite pooling: with `pool` (n_img * chunks * Cp floats, dwconv_pool_layout) the depthwise
// kernel also writes per-block channel sums of its output; se_fc_from_dw turns them into the scales.
void dwconv_pool_layout(int K, int sh, int sw, int Cp, int maxHo, int maxWo, int* chunks, int* strip_R, int* strips_per_block);
void se_fc_from_dw(hipStream_t st, const float* partial, const ImgGeom* geom, int n_img, int chunks, int strip_R,
                   int strips_per_block, int C, int Cp, const float* w1, const float* b1, const float* w2, const float* b2,
                   int Cr, float slope, int residual, float* scale);
// Row-tile height of the wide GEMM the dispatcher picks for (M, Npad16), 0 when it picks the narrow
// kernel: Epilogue::a_scale (squeeze-excite scale folded into the A staging) needs a wide tile and
// every image at least that many rows.
int gemm_tile_rows(long long M, int Npad16);
// ... and which fused squeeze-excite form gemm() will take for this layer: 256 (k_gemm32p: a_tab entries of 3 ints per
// 256-row block, Epilogue::a_tab_stride = 3, n_img set), 128 (wide register-staged tiles: 2 ints per 128-row block) or 0 (none:
// scale the tensor in a pass of its own).  min_pix: rows of the smallest image.
int gemm_se_tile_rows(int lda, long long M, int K, int N, int Npad16, int act, long long min_pix);
// Fused CTC head: gemm() with Epilogue::am_* set (am_tiles = gemm_argmax_tiles(Npad16), buffers of
// M * am_tiles elements) leaves softmax statistics per column tile; argmax_merge gives, per row, the
// argmax over the N logits and softmax(logits)[argmax] -- the [M, 6625] logits never reach HBM.
extern int g_argmax_wide;
int gemm_argmax_tiles(int Npad16);
void argmax_merge(hipStream_t st, const float* pm, const int* pi, const float* ps, int tiles, long long rows, int* idx,
                  float* prob);

// One MobileNetV3 block of the angle classifier (expand 1x1 -> depthwise k x k -> squeeze-excite -> linear 1x1 [+ x]) as one
// kernel, a workgroup per crop (nn_clsblock.hip).  Weights in the PackedDense / PackedDw layouts of nets.cpp.
extern int g_cls_fused;
bool cls_block_supported(int k, int sh, int sw, int cin, int mid, int cout, int act, int maxH_in, int maxW, int max_pix_out);
void cls_block(hipStream_t st, int k, int sh, bool se, int act, const float* x, const ImgGeom* gin, const ImgGeom* gout, int n_img,
               int maxH_in, int maxW, int max_pix_out, int cin, int mid, int mid_cp, int cout, const float* Wexp, const float* bexp,
               const float* Wdw, const float* bdw, const float* w1, const float* b1, const float* w2, const float* b2, int cr, float slope,
               const float* Wlin, const float* blin, bool shortcut, float* y, float* dscr);   // dscr: n_pixels_out x round_up(mid, 16) floats when se
// Fused thin LCNetV3 block (3x3 depthwise -> 1x1 conv, C_in <= 64, no SE): see k_lc_thin.
extern int g_lc_thin;
bool lc_thin_supported(int K, int sh, int sw, int Cp, int C, int Npad16);
void lc_thin(hipStream_t st, int sh, int sw, const float* x, const ImgGeom* gin, const ImgGeom* gout, int n_img, int maxHo,
             int maxWo, int Cp, int C, const float* Wd, const float* bd, int dw_act, int dw_has_lab, float dw_a, float dw_c,
             const float* Wp, int N, int Npad16, float* y, int ldy, const Epilogue& epi);
// one-wave-per-tile form of the same block (nn_lcwave.hip): activations never touch LDS; bit-identical to lc_thin
extern int g_lc_wave;
// either form has an instance for the block (what run_lc asks before it takes the fused path)
bool lc_block_supported(int K, int sh, int sw, int Cp, int C, int N, int Npad16, int dw_act, int dw_has_lab, const Epilogue& epi,
                        int maxHo, int maxWo);
bool lc_wave_supported(int K, int sh, int sw, int Cp, int C, int N, int Npad16, int dw_act, int dw_has_lab, const Epilogue& epi);
void lc_wave(hipStream_t st, int sh, int sw, const float* x, const ImgGeom* gin, const ImgGeom* gout, int n_img, int maxHo,
             int maxWo, int Cp, int C, const float* Wd, const float* bd, int dw_act, int dw_has_lab, float dw_a, float dw_c,
             const float* Wp, int N, int Npad16, float* y, int ldy, const Epilogue& epi);

// 3x3 stride-2 stem on a 3(+1 pad)-channel f32 NHWC input. Ws packed [27][COUT]. COUT in {8,16}.
// One RGB8 page of a det launch group (device pointer; npix = H*W; out_pix = its pixel offset in the group).
struct U8Page { const uint8_t* rgb; long long npix; long long out_pix; };
void stem_conv_u8(hipStream_t st, const U8Page* pages, float scale, const float* mean3, const float* std3, const ImgGeom* gin,
                  const ImgGeom* gout, int n_img, int maxHo, int maxWo, int COUT, const float* Ws, const float* bias, int act,
                  float* y);
void stem_conv(hipStream_t st, const float* x, const ImgGeom* gin, const ImgGeom* gout, int n_img, int maxHo, int maxWo,
               int COUT, const float* Ws, const float* bias, int act, float* y);

// NCHW f32 [n,3,H,W] (uniform dims) -> NHWC pitch 4 (pad = 0)
void nchw3_to_nhwc4(hipStream_t st, const float* in, int n, int H, int W, float* out);

// Squeeze-excite: deterministic two-stage spatial mean, then fc1-relu-fc2-hardsigmoid.
// scale[n][Cp] = hsig(...) (+1 when residual: RSELayer's x + x*s).  `partial` scratch holds
// n_img*chunks*Cp floats with chunks = pool_chunks(maxPix).
int pool_chunks(long long max_pix);
void se_scale(hipStream_t st, const float* x, const ImgGeom* geom, int n_img, long long max_pix, int C, int Cp,
              const float* w1, const float* b1, const float* w2, const float* b2, int Cr, float slope, int residual,
              float* partial, float* scale);
void scale_channels(hipStream_t st, float* x, const ImgGeom* geom, int n_img, long long max_pix, int Cp,
                    const float* scale);
// mean over H*W per image -> out[n][Cp] (uses the same partial scratch)
void global_mean(hipStream_t st, const float* x, const ImgGeom* geom, int n_img, long long max_pix, int Cp,
                 float* partial, float* out);

// FPN lateral without materialising the lateral conv: se_scale_projected pools the narrow tap tensor x (pitch Cin_p)
// and maps the means through Wlin [Cin][C] before the SE FCs; lateral_add then writes (x . Wlin) * scale + up2(b).
void se_scale_projected(hipStream_t st, const float* x_in, const ImgGeom* geom, int n_img, long long max_pix, int Cin,
                        int Cin_p, const float* Wlin, int C, int Cp, const float* w1, const float* b1, const float* w2,
                        const float* b2, int Cr, float slope, int residual, float* partial, float* scale);
void lateral_add(hipStream_t st, const float* x, int Cin, int Cin_p, const float* Wlin, int C, const float* scale,
                 const float* b, const ImgGeom* ga, const ImgGeom* gb, int n_img, long long max_pix, float* out);
// out = a * scale_a[image] + nearest_up2(b)   (in place on a allowed; scale_a [image][Cp] optional)
void upsample_add(hipStream_t st, const float* a, const float* b, const ImgGeom* ga, const ImgGeom* gb, int n_img,
                  long long max_pix, int Cp, float* out, const float* scale_a = nullptr);
// fuse = concat(up8(p5), up4(p4), up2(p3), p2) along channels; every input has pitch Cq, output pitch 4*Cq
void fpn_concat(hipStream_t st, const float* p5, const float* p4, const float* p3, const float* p2, const ImgGeom* g5,
                const ImgGeom* g4, const ImgGeom* g3, const ImgGeom* g2, int n_img, long long max_pix, int Cq,
                float* out, const float* const* scales = nullptr);  // scales[4]: per-level [img][Cq] factors (p5..p2) or null
// The DB head's first 3x3 conv reading concat(up8(p5), up4(p4), up2(p3), p2) (x per-level scales) directly from the four levels
// (fpn_concat + conv_sp in one kernel; the pair's result to fp32 rounding: the K slabs are the four levels).
bool conv3_fpn_fused_supported(int Cq, int N);
void conv3_fpn_fused(hipStream_t st, const float* p5, const float* p4, const float* p3, const float* p2, const ImgGeom* g5,
                     const ImgGeom* g4, const ImgGeom* g3, const ImgGeom* g2, int n_img, int maxH, int maxW, int Cq,
                     const float* const* scales, const float* Wp, int N, int Npad16, float* y, int ldy, const Epilogue& epi);
// RSEFPN / DB-head 3x3 convs restructured around the nearest-neighbour upsampling (nn_fpn.hip).  k_fpn_phase computes, for the
// pixels of a level gf whose next coarser level gc is exactly half its size,
//   y = act( conv3x3(fine * fine_scale; Wf) + conv3x3(up2(coarse * coarse_scale)) [as four 2 x 2 phase convs; Wc pre-summed]
//            + (G ? G[pixel >> 2][class of (y & 3, x & 3)] : bias) )                                  onto 24 output channels.
struct FpnPhaseArgs {
  const float* fine = nullptr; int ld_fine = 0;        // fine tensor, cf channels (cf4 = ceil(cf / 4) 16-byte chunks read per pixel)
  const float* fine_scale = nullptr; int ld_fs = 0;    // [image][ld_fs] or null
  const float* Wf = nullptr; long long wf_img = 0;     // [9 taps][24][4 cf4] (+ image * wf_img: per-image composed weights)
  const float* coarse = nullptr; int ld_coarse = 0;    // coarse tensor, cc = 24 or 96 channels
  const float* coarse_scale = nullptr; int ld_cs = 0;
  const float* Wc = nullptr;                           // [cc / 24 slabs][4 phases][4 taps][24 n][24 k]
  const float* G = nullptr; const ImgGeom* gg = nullptr;  // class tensor [9][pixel of gg = gf / 4][24] (fpn_class), bias included
  long long g_plane = 0;                               // pixels of the level behind G (stride between its class planes)
  const float* bias = nullptr;                         // [24], used when G is null
  float* y = nullptr; int ldy = 0;
  float* pool = nullptr; int pool_tiles = 0;           // optional [image][pool_tiles][24]: per-tile channel sums of y (16 x 16 tiles, raster order)
  int act = ACT_NONE;                                  // ACT_NONE / ACT_RELU
};
extern int g_fpn_phase_off;   // A/B (rt_debug_set_variants bit 11): 1 = the round-3 launch series
bool fpn_phase_supported(int cf, int cc);
void fpn_phase(hipStream_t st, const FpnPhaseArgs& a, int cf, int cc, const ImgGeom* gf, const ImgGeom* gc, int n_img, int maxH, int maxW);
// V[9][pixel][24] (plane = pixels of the level) = class tensor of z (24 channels, x scale) for the head conv's up4 / up8 inputs; Wcls [9][9][24 n][24 k];
// + bias, + lower[implied class][(y >> 1, x >> 1)] (the next coarser level's class tensor) when given.
void fpn_class(hipStream_t st, const float* z, int ldz, const float* scale, int ld_s, const ImgGeom* geom, int n_img,
               long long max_pix, const float* Wcls, const float* bias, const float* lower, const ImgGeom* glow, long long low_plane,
               float* V, long long plane);
// out[img][9][24][cf] = sum_m Wlat[c][m] * scale[img][m] * Wm[tap][n][m]: the lateral 1x1 conv, its squeeze-excite factor and the
// 3x3 conv that follows composed into one 3x3 conv of the narrow tap tensor
void fpn_compose(hipStream_t st, const float* Wlat, int cin, int C, const float* scale, const float* Wm, int cf, int n_img, float* out);
// squeeze-excite FCs from per-tile channel sums (FpnPhaseArgs::pool; tiles = 16 x 16 pixels in raster order)
void se_fc_from_tiles(hipStream_t st, const float* partial, const ImgGeom* geom, int n_img, int tiles_alloc, int C, int Cp,
                      const float* w1, const float* b1, const float* w2, const float* b2, int Cr, float slope, int residual,
                      float* scale);
// DB head tail: convT2x2s2(24->24)+relu, convT2x2s2(24->1), sigmoid. in [.,.,24] at 1/4 res,
// out f32 map at full res (geometry gout, one float per pixel).
void db_head_tail(hipStream_t st, const float* x, const ImgGeom* gin, const ImgGeom* gout, int n_img, long long max_pix,
                  const float* w1, const float* b1, const float* w2, const float* b2, float* out);

void avgpool_3x2(hipStream_t st, const float* x, const ImgGeom* gin, const ImgGeom* gout, int n_img, long long max_pix,
                 int Cp, float* y, int ldy);
void maxpool_2x2(hipStream_t st, const float* x, const ImgGeom* gin, const ImgGeom* gout, int n_img, long long max_pix,
                 int Cp, float* y);
// y = LayerNorm(x + r) over the last dim C (r may be null), rows contiguous with pitch C
void add_layernorm(hipStream_t st, const float* x, const float* r, long long rows, int C, const float* g,
                   const float* beta, float eps, float* y);
// Global multi-head attention over each image's tokens. qkv [rows, 3*C] (q|k|v, channel = head*hd + d)
void attention(hipStream_t st, const float* qkv, const ImgGeom* geom, int n_img, int maxT, int heads, int hd,
               float* out);
void copy_channels(hipStream_t st, const float* src, int lds, long long rows, int C, float* dst, int ldd, int coff);
// row softmax: in [rows, ld] (first C valid) -> out [rows, C] dense
void softmax_rows(hipStream_t st, const float* in, int ld, long long rows, int C, float* out);
// per-row (argmax logit, softmax max prob) without materialising the softmax: in [rows, ld]
void argmax_prob_rows(hipStream_t st, const float* in, int ld, long long rows, int C, int* idx, float* prob);
// per-row (first argmax, max value) of rows that already hold probabilities (ndarray-stats argmax/max)
void argmax_rows(hipStream_t st, const float* in, int ld, long long rows, int C, int* idx, float* maxval);

}  // namespace nn
}  // namespace rt
