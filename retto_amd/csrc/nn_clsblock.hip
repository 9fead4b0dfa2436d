// One MobileNetV3 inverted-residual block of the angle classifier as ONE kernel, a workgroup per crop (gfx950).
//
//   y = W_lin . ( se( act( dw_kxk( act( W_exp . x + b_exp ) ) + b_dw ) ) ) + b_lin  [+ x]
//
// The classifier's tensors are tiny (a 48 x 192 crop is 24 x 96 x 8 floats after the stem, 2 x 96 x 32 at the end) and its
// eleven blocks ran as ~55 launches of 20-50 us (expand GEMM, depthwise, pooling, SE FC, channel scaling, linear GEMM): 2.5 ms of a
// 31 ms C3 step for 33 GFLOP.  Here a workgroup of 8 waves owns a crop for the whole block:
//   * per 16-channel slice of the expansion: v_mfma_f32_16x16x4_f32 with the crop's pixels straight from global memory (a
//     pixel is 32-128 bytes), bias + activation, the slice to LDS as [pixel][64 bytes] (chunk XOR-swizzled like k_lc_lds);
//     barrier; depthwise k x k from LDS with lane (r, q) = pixel r of a 16-pixel row tile, chunk q -- the result is the MFMA
//     pixel operand of the linear 1x1, accumulated over the slices in registers; barrier;
//   * squeeze-excite is local to the workgroup: a first pass over the slices only pools the depthwise outputs (per-wave partial
//     sums in LDS, added in a fixed order: deterministic), the FC runs on the first threads, a second pass recomputes the slices
//     and scales them in front of the linear MFMAs -- recomputing is cheaper than 154 KB of depthwise output per crop;
//   * weights and taps are read from global memory in operand layout where they are used (a few KB per block: L1 / L2).
// Same formulas and the same k order inside every sum as the unfused kernels; the pooling sums are ordered differently
// (fp32 tolerance, tests/test_gpu_parity.py::test_cls_net).  Row width must be a multiple of 16 (the classifier's 96).
#include "common.h"
#include "nn_dev.h"

namespace rt {
namespace nn {

struct ClsBlkArgs {
  const float* x; float* y; const ImgGeom* gin; const ImgGeom* gout;
  const float* Wexp; const float* bexp; const float* Wdw; const float* bdw;
  const float* w1; const float* b1; const float* w2; const float* b2;
  const float* Wlin; const float* blin; float* dscr;   // dscr: [crop][pixel][npad_e] depthwise outputs of the squeeze-excite blocks
  int cin, mid, mid_cp, cout, npad_e, npad_l, cr, shortcut, es_bytes;
  float slope;
};

namespace {
template <int ACT>
__device__ __forceinline__ f32x4 act4(const f32x4& v) {
  f32x4 o;
#pragma unroll
  for (int e = 0; e < 4; e++) {
    const float t = v[e];
    o[e] = ACT == ACT_RELU ? fmaxf(t, 0.0f) : t * fminf(fmaxf(t + 3.0f, 0.0f), 6.0f) * 0.16666667f;
  }
  return o;
}
__device__ __forceinline__ int swz64(int px, int c) { return (px * 4 + (c ^ ((px >> 1) & 3))) * 16; }   // byte offset of chunk c of pixel px
}  // namespace

int g_cls_fused = getenv("RT_CLS_FUSED") ? atoi(getenv("RT_CLS_FUSED")) : 1;   // A/B: 0 = the unfused launch series

// NW waves per workgroup, each owning up to MAXU output row tiles (their linear accumulators live in registers): 8 x 9 for the
// first block's 72 tiles, 8 x 5 for the later ones (fewer registers: more crops per CU; at 8 x 9 everywhere the 200-channel blocks
// ran one crop per CU, 0.35 ms)
// PADROW: the LDS slice is [row][W + 2 PAD pixels][80 bytes] with zero columns left and right (no column checks, no swizzle:
// 80-byte pixels are conflict-free for the 16-byte reads of 8 neighbouring lanes) -- every block but the first, whose 24 x 96
// input does not fit that way and keeps 64-byte pixels with the XOR swizzle and checked columns.
template <int KS, int SH, bool SE, int ACT, int NW, int MAXU, bool PADROW>
__global__ __launch_bounds__(64 * NW) void k_cls_block(ClsBlkArgs p) {
  constexpr int CLS_MAXU = MAXU, NTHR = 64 * NW;
  extern __shared__ __attribute__((aligned(16))) char smem_cb[];
  const ImgGeom gi = p.gin[blockIdx.x], go = p.gout[blockIdx.x];
  const int W = gi.W, tiles_x = W >> 4, nt_in = gi.H * tiles_x, nt_out = go.H * tiles_x;
  char* es = smem_cb;
  float* poolw = reinterpret_cast<float*>(smem_cb + p.es_bytes);   // [NW waves][16 channels]
  float* pool = poolw + NW * 16;                                      // [npad_e] sums, then means
  float* scale = pool + p.npad_e;                                    // [npad_e]
  float* hid = scale + p.npad_e;                                     // [128]
  float* xs = hid + 128 + (KS * KS + 1) * 16;                         // PADROW: the crop's input [pixel][cin + 4]
  float* taps = hid + 128;                                           // [KS * KS + 1][16]: depthwise taps and bias of the current slice, zero past the channel pitch
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 15, q = lane >> 4;
  const int NS = p.npad_e >> 4, GI = (p.cin + 15) >> 4, NTL = p.npad_l >> 4;
  const float* xc = p.x + gi.off * p.cin;

  f32x4 accy[CLS_MAXU][2];
#pragma unroll
  for (int j = 0; j < CLS_MAXU; j++)
#pragma unroll
    for (int nt = 0; nt < 2; nt++) accy[j][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
  constexpr int PAD = KS / 2;
  const int WP = PADROW ? gi.W + 2 * PAD : gi.W;   // pixels per LDS row
  if (PADROW) {   // the zero columns (never written again)
    for (int i = tid; i < gi.H * 2 * PAD * 5; i += NTHR) {
      const int c = i % 5, pc = (i / 5) % (2 * PAD), row = i / (10 * PAD);
      *reinterpret_cast<f32x4*>(es + ((size_t)row * WP + (pc < PAD ? pc : gi.W + pc)) * 80 + c * 16) = f32x4{0.f, 0.f, 0.f, 0.f};
    }
  }
  const int xp = p.cin + 4;   // pixel pitch of xs in floats: conflict-free 16-byte reads of 8 neighbouring pixels
  if (PADROW) {   // the input of the block, read once (every slice of the expansion uses all of it)
    const int c4n = p.cin >> 2;
    for (int i = tid; i < gi.H * W * c4n; i += NTHR) {
      const int px = i / c4n, c4 = i - px * c4n;
      *reinterpret_cast<f32x4*>(xs + (size_t)px * xp + c4 * 4) = *reinterpret_cast<const f32x4*>(xc + (size_t)px * p.cin + c4 * 4);
    }
  }
  float* dcrop = SE ? p.dscr + (size_t)go.off * p.npad_e : nullptr;
  // weight fragments and biases of a slice are requested one slice ahead (their L2 round trip was on every slice's critical path)
  f32x4 wa_n[2], be_n, wl_n[2];
  auto load_frags = [&](int s) __attribute__((always_inline)) {
#pragma unroll
    for (int g = 0; g < 2; g++)
      wa_n[g] = g < GI ? *reinterpret_cast<const f32x4*>(p.Wexp + (size_t)(16 * s + r) * KC + 16 * g + 4 * q) : f32x4{0.f, 0.f, 0.f, 0.f};
    be_n = *reinterpret_cast<const f32x4*>(p.bexp + 16 * s + 4 * q);
#pragma unroll
    for (int nt = 0; nt < 2; nt++)
      wl_n[nt] = nt < NTL ? *reinterpret_cast<const f32x4*>(p.Wlin + ((size_t)(s >> 1) * p.npad_l + nt * 16 + r) * KC + (s & 1) * 16 + 4 * q)
                          : f32x4{0.f, 0.f, 0.f, 0.f};
  };
  load_frags(0);
  __syncthreads();

  {
    constexpr int pass = 0;
    for (int s = 0; s < NS; s++) {
      const int ch = 16 * s + 4 * q;   // this lane's 4 channels of the slice
      // ---- expansion of slice s: every pixel of the crop -> LDS ----
      {
        const f32x4 wa[2] = {wa_n[0], wa_n[1]};
        const f32x4 be = be_n;
        for (int t = wave; t < nt_in; t += NW) {
          const int px = t * 16 + r;
          f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int g = 0; g < 2; g++) {
            if (g < GI) {
              f32x4 xa = {0.f, 0.f, 0.f, 0.f};
              if (16 * g + 4 * q < p.cin) xa = PADROW ? *reinterpret_cast<const f32x4*>(xs + (size_t)px * xp + 16 * g + 4 * q)
                                                      : *reinterpret_cast<const f32x4*>(xc + (size_t)px * p.cin + 16 * g + 4 * q);
#pragma unroll
              for (int st = 0; st < 4; st++) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[g][st], xa[st], acc, 0, 0, 0);
            }
          }
          if (PADROW) { const int py = px / W, pxx = px - py * W; *reinterpret_cast<f32x4*>(es + ((size_t)py * WP + pxx + PAD) * 80 + q * 16) = act4<ACT>(acc + be); }
          else *reinterpret_cast<f32x4*>(es + swz64(px, q)) = act4<ACT>(acc + be);
        }
      }
      if (tid < (KS * KS + 1) * 4) {   // taps and bias of the slice -> LDS (their last readers are behind the barrier at the end of the previous slice)
        const int t = tid >> 2, c = 16 * s + 4 * (tid & 3);
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (c < p.mid_cp) v = t < KS * KS ? *reinterpret_cast<const f32x4*>(p.Wdw + (size_t)t * p.mid_cp + c) : *reinterpret_cast<const f32x4*>(p.bdw + c);
        *reinterpret_cast<f32x4*>(taps + tid * 4) = v;
      }
      __syncthreads();
      // ---- depthwise k x k of the slice, then pooling (first squeeze-excite pass) or the linear 1x1 ----
      {
        const f32x4 bd = *reinterpret_cast<const f32x4*>(taps + KS * KS * 16 + 4 * q);
        const f32x4 wl[2] = {wl_n[0], wl_n[1]};
        if (s + 1 < NS) load_frags(s + 1);
        f32x4 psum = {0.f, 0.f, 0.f, 0.f};
        // all row tiles of the wave advance together through the kernel rows: the taps of a row are read once, and the LDS
        // reads of the tiles are in flight together (tile by tile, the rolled row loop was one latency chain per tile)
        f32x4 dsum[CLS_MAXU];
        int oys[CLS_MAXU], oxs[CLS_MAXU];
#pragma unroll
        for (int j = 0; j < CLS_MAXU; j++) {
          const int u = wave + NW * j;
          oys[j] = u / tiles_x; oxs[j] = (u - oys[j] * tiles_x) * 16 + r;
          dsum[j] = bd;
        }
#pragma unroll 1   // (rolled: unrolled, the 25 taps of the 5x5 blocks are hoisted into 100 registers and the kernel spills)
        for (int dy = 0; dy < KS; dy++) {
          f32x4 w[KS];
#pragma unroll
          for (int dx = 0; dx < KS; dx++) w[dx] = *reinterpret_cast<const f32x4*>(taps + (dy * KS + dx) * 16 + 4 * q);
#pragma unroll
          for (int j = 0; j < CLS_MAXU; j++) {
            const int iy = oys[j] * SH + dy - PAD;
            if (wave + NW * j < nt_out && (unsigned)iy < (unsigned)gi.H) {   // (uniform per wave)
              if (PADROW) {
                const char* rowp = es + ((size_t)iy * WP + oxs[j]) * 80 + q * 16;   // padded column ox + dx = image column ox + dx - PAD
#pragma unroll
                for (int dx = 0; dx < KS; dx++) dsum[j] = __builtin_elementwise_fma(*reinterpret_cast<const f32x4*>(rowp + dx * 80), w[dx], dsum[j]);
              } else {
#pragma unroll
                for (int dx = 0; dx < KS; dx++) {
                  const int ix = oxs[j] + dx - PAD;
                  f32x4 v = {0.f, 0.f, 0.f, 0.f};
                  if ((unsigned)ix < (unsigned)W) v = *reinterpret_cast<const f32x4*>(es + swz64(iy * W + ix, q));
                  dsum[j] = __builtin_elementwise_fma(v, w[dx], dsum[j]);
                }
              }
            }
          }
        }
#pragma unroll
        for (int j = 0; j < CLS_MAXU; j++) {
          const int u = wave + NW * j;
          if (u < nt_out) {   // (uniform per wave)
            f32x4 d = act4<ACT>(dsum[j]);
            if (SE && pass == 0) {
              psum += d;
              *reinterpret_cast<f32x4*>(dcrop + (size_t)(u * 16 + r) * p.npad_e + ch) = d;   // (read back by the same lane in the second pass)
            } else {
#pragma unroll
              for (int nt = 0; nt < 2; nt++)
                if (nt < NTL) {
#pragma unroll
                  for (int st = 0; st < 4; st++) accy[j][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wl[nt][st], d[st], accy[j][nt], 0, 0, 0);
                }
            }
          }
        }
        if (SE && pass == 0) {
          // sum over the 16 pixel lanes of a row in registers (DPP, fixed order), lane r = 15 holds the total
#pragma unroll
          for (int e = 0; e < 4; e++) {
            float v = psum[e];
            v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x118, 0xf, 0xf, true));   // row_shr:8
            v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x114, 0xf, 0xf, true));   // row_shr:4
            v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x112, 0xf, 0xf, true));   // row_shr:2
            v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x111, 0xf, 0xf, true));   // row_shr:1
            psum[e] = v;
          }
          if (r == 15) *reinterpret_cast<f32x4*>(poolw + wave * 16 + 4 * q) = psum;
        }
      }
      __syncthreads();
      if (SE && pass == 0 && tid < 16) {   // channel 16 s + tid: the waves' partial sums in a fixed order
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < NW; k++) t += poolw[k * 16 + tid];
        pool[16 * s + tid] = t;
      }
      // (the next write of poolw / es is behind the barrier in the middle of the next slice)
    }
    if (SE) {
      __syncthreads();
      const float inv = 1.0f / (float)(go.H * W);
      // FC1, w1 [cr][mid]: a wave per hidden unit, lanes across the channels (coalesced rows; a thread per unit walked its row
      // with one dependent L2 round trip per channel: ~100 us per crop on the 200-channel blocks)
      for (int j = wave; j < p.cr; j += NW) {
        float t = 0.f;
        for (int c4 = lane * 4; c4 < p.mid; c4 += 256) {
          const f32x4 m4 = *reinterpret_cast<const f32x4*>(pool + c4), w4 = *reinterpret_cast<const f32x4*>(p.w1 + (size_t)j * p.mid + c4);
          t = fmaf(m4[0] * inv, w4[0], t); t = fmaf(m4[1] * inv, w4[1], t); t = fmaf(m4[2] * inv, w4[2], t); t = fmaf(m4[3] * inv, w4[3], t);
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o, 64);
        if (lane == 0) hid[j] = fmaxf(t + p.b1[j], 0.f);
      }
      __syncthreads();
      for (int c = tid; c < p.npad_e; c += NTHR) {   // FC2, w2 [mid][cr]: a thread per channel, its row's loads issued together
        float o = 0.f;
        if (c < p.mid) {
          float t = p.b2[c];
          const float* wr = p.w2 + (size_t)c * p.cr;
          int j = 0;
          for (; j + 8 <= p.cr; j += 8) {
            float wv[8];
#pragma unroll
            for (int i = 0; i < 8; i++) wv[i] = wr[j + i];
#pragma unroll
            for (int i = 0; i < 8; i++) t = fmaf(hid[j + i], wv[i], t);
          }
          for (; j < p.cr; j++) t = fmaf(hid[j], wr[j], t);
          o = fminf(fmaxf(fmaf(t, p.slope, 0.5f), 0.f), 1.f);
        }
        scale[c] = o;
      }
      __syncthreads();
      // second pass: the depthwise outputs come back from the scratch (each lane its own values: no barrier), scaled, into the linear 1x1
      for (int s = 0; s < NS; s++) {
        const int ch = 16 * s + 4 * q;
        const f32x4 sc = *reinterpret_cast<const f32x4*>(scale + ch);
        f32x4 wl[2];
#pragma unroll
        for (int nt = 0; nt < 2; nt++)
          wl[nt] = nt < NTL ? *reinterpret_cast<const f32x4*>(p.Wlin + ((size_t)(s >> 1) * p.npad_l + nt * 16 + r) * KC + (s & 1) * 16 + 4 * q)
                            : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < CLS_MAXU; j++) {
          const int u = wave + NW * j;
          if (u < nt_out) {
            const f32x4 d = *reinterpret_cast<const f32x4*>(dcrop + (size_t)(u * 16 + r) * p.npad_e + ch) * sc;
#pragma unroll
            for (int nt = 0; nt < 2; nt++)
              if (nt < NTL) {
#pragma unroll
                for (int st = 0; st < 4; st++) accy[j][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wl[nt][st], d[st], accy[j][nt], 0, 0, 0);
              }
          }
        }
      }
    }
  }
  // ---- epilogue: lane (r, q) holds output channels 16 nt + 4 q .. + 3 of pixel r of its row tiles ----
  float* yc = p.y + go.off * p.cout;
#pragma unroll
  for (int j = 0; j < CLS_MAXU; j++) {
    const int u = wave + NW * j;
    if (u < nt_out) {
      const int px = u * 16 + r;
#pragma unroll
      for (int nt = 0; nt < 2; nt++) {
        const int co = nt * 16 + 4 * q;
        if (nt < NTL && co < p.cout) {
          f32x4 o = accy[j][nt] + *reinterpret_cast<const f32x4*>(p.blin + co);
          if (p.shortcut) o += *reinterpret_cast<const f32x4*>(xc + (size_t)px * p.cin + co);
          *reinterpret_cast<f32x4*>(yc + (size_t)px * p.cout + co) = o;
        }
      }
    }
  }
}

bool cls_block_supported(int k, int sh, int sw, int cin, int mid, int cout, int act, int maxH_in, int maxW, int max_pix_out) {
  if (!(k == 3 || k == 5) || !(sh == 1 || sh == 2) || sw != 1) return false;
  if (!(act == ACT_RELU || act == ACT_HSWISH)) return false;
  if (cin % 4 || cin > 32 || cout % 4 || cout > 32 || mid > 512) return false;
  if (maxW % 16 || maxW <= 0) return false;
  if ((max_pix_out + 15) / 16 > 8 * 9) return false;
  return (long long)maxH_in * maxW * 64 + (8 * 16 + 2 * round_up(mid, 16) + 128 + (k * k + 1) * 16) * 4 <= 160 * 1024;
}

void cls_block(hipStream_t st, int k, int sh, bool se, int act, const float* x, const ImgGeom* gin, const ImgGeom* gout, int n_img,
               int maxH_in, int maxW, int max_pix_out, int cin, int mid, int mid_cp, int cout, const float* Wexp, const float* bexp,
               const float* Wdw, const float* bdw, const float* w1, const float* b1, const float* w2, const float* b2, int cr, float slope,
               const float* Wlin, const float* blin, bool shortcut, float* y, float* dscr) {
  if (n_img <= 0) return;
  static const int dbg = getenv("RT_CLS_DBG") ? atoi(getenv("RT_CLS_DBG")) : 0;   // timing experiments only (wrong results)
  if (dbg & 1) se = false;
  if (se && !dscr) throw RtError(8, "cls_block: a squeeze-excite block needs the scratch tensor");
  ClsBlkArgs a{x, y, gin, gout, Wexp, bexp, Wdw, bdw, w1, b1, w2, b2, Wlin, blin, dscr,
               cin, mid, mid_cp, cout, round_up(mid, 16), round_up(cout, 16), cr, shortcut ? 1 : 0, maxH_in * maxW * 64, slope};
  const int nt_out = (max_pix_out + 15) / 16;
  // 8 waves per crop everywhere (4 on the small maps measured 1.31 vs 1.11 ms for the eleven blocks: a launch has only ~1-4
  // crops per CU, so a crop's latency chain, not the CU's wave slots, is what counts); 9 row tiles per wave where needed
  const int shape = nt_out > 40 ? 0 : 1;   // 8 waves x 9 tiles | 8 x 5
  const int nw = 8;
  if (dbg & 2) a.npad_e = 16;   // one slice only
  const int small = (nw * 16 + 2 * a.npad_e + 128 + (k * k + 1) * 16) * 4;
  const long long padded = (long long)maxH_in * (maxW + 2 * (k / 2)) * 80;
  const long long xs_bytes = (long long)maxH_in * maxW * (cin + 4) * 4;
  const bool padrow = padded + small + xs_bytes <= 96 * 1024;   // (the first block's 24 x 96 input: 188 KB that way)
  if (padrow) a.es_bytes = (int)padded;
  const int lds = a.es_bytes + small + (padrow ? (int)xs_bytes : 0);
#define RT_CB(KK, SS, EE, AA, NWW, MU, PP)                                                               \
  do {                                                                                                   \
    allow_big_lds((const void*)k_cls_block<KK, SS, EE, AA, NWW, MU, PP>, 160 * 1024);                    \
    RT_LAUNCH((k_cls_block<KK, SS, EE, AA, NWW, MU, PP>), dim3(n_img), dim3(64 * NWW), lds, st, a);     \
  } while (0)
#define RT_CB_S(KK, SS, EE, AA) do { if (!padrow) RT_CB(KK, SS, EE, AA, 8, 9, false); else if (shape == 0) RT_CB(KK, SS, EE, AA, 8, 9, true); else RT_CB(KK, SS, EE, AA, 8, 5, true); } while (0)
#define RT_CB_A(KK, SS, EE) do { if (act == ACT_RELU) RT_CB_S(KK, SS, EE, ACT_RELU); else RT_CB_S(KK, SS, EE, ACT_HSWISH); } while (0)
#define RT_CB_E(KK, SS) do { if (se) RT_CB_A(KK, SS, true); else RT_CB_A(KK, SS, false); } while (0)
  if (k == 3 && sh == 1) RT_CB_E(3, 1);
  else if (k == 3) RT_CB_E(3, 2);
  else if (sh == 1) RT_CB_E(5, 1);
  else RT_CB_E(5, 2);
#undef RT_CB_E
#undef RT_CB_A
#undef RT_CB_S
#undef RT_CB
}

}  // namespace nn
}  // namespace rt
