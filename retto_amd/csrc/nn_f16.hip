// fp16-storage / fp32-accumulate kernels for gfx950 (see nn_f16.h).  Written for CDNA4 only:
// v_mfma_f32_32x32x16_f16, 64-wide waves, ds_read_b128 fragment reads from padded LDS rows.
#include "nn_f16_dev.h"

#include <algorithm>
#include <cmath>
#include <type_traits>

namespace rt {
namespace nh {

constexpr int WPRE = 8;  // 16-byte chunks of the next weight row a thread can hold in registers

template <int NTN, int NTP, int WN, int WP, int DOT>
__global__ __launch_bounds__(64 * WN * WP, 2) void k_conv16(const ConvArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int NTHR = 64 * WN * WP, BN = 32 * NTN * WN;
  const int TH = a.TH, TW = a.TW;
  const ImgGeom go = a.gout[blockIdx.y];
  const int tiles_x = (go.W + TW - 1) / TW, tiles_y = (go.H + TH - 1) / TH;
  const int zb = blockIdx.x % a.nzb, tile = blockIdx.x / a.nzb;
  if (tile >= tiles_x * tiles_y) return;
  const ImgGeom gi = a.gin[blockIdx.y];
  const int ty0 = (tile / tiles_x) * TH, tx0 = (tile % tiles_x) * TW;
  const int nb0 = zb * BN;
  const int HH = (TH - 1) * a.SH + a.KH, HW = (TW - 1) * a.SW + a.KW;
  const int lp = a.lp;
  half_t* halo = reinterpret_cast<half_t*>(smem);
  half_t* wl = halo + (((size_t)HH * HW * lp + 7) & ~(size_t)7);
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wn = wid % WN, wp = wid / WN;
  const int r = lane & 31, h = lane >> 5;

  int pbase[NTP], oys[NTP], oxs[NTP];  // oys < 0: lane has no pixel in this tile
#pragma unroll
  for (int j = 0; j < NTP; j++) {
    int q = (wp * NTP + j) * 32 + r;
    int ty = q / TW, tx = q - ty * TW;
    bool ok = ty < TH;
    if (!ok) { ty = 0; tx = 0; }
    pbase[j] = (ty * a.SH * HW + tx * a.SW) * lp + h * 8;
    oys[j] = ok ? ty0 + ty : -1;
    oxs[j] = tx0 + tx;
  }
  const int abase = (wn * NTN * 32 + r) * lp + h * 8;

  f32x16 acc[NTN][NTP];
#pragma unroll
  for (int i = 0; i < NTN; i++)
#pragma unroll
    for (int j = 0; j < NTP; j++)
#pragma unroll
      for (int e = 0; e < 16; e++) acc[i][j][e] = 0.f;

  const int iy0 = ty0 * a.SH - a.PT, ix0 = tx0 * a.SW - a.PL;
  const int nslab = (a.Cin + KS - 1) / KS;
  const int nrows = nslab * a.KH;
  const int cpp = (lp >> 3) - 1;               // 16-byte chunks per staged LDS row (pixel or output channel) that hold data
  const int wchunks = a.KW * BN * cpp;         // chunks of one staged weight row
  const int hchunks = HH * HW * cpp;           // chunks of one staged halo tile
  const bool prefetch = wchunks <= WPRE * NTHR;
  const size_t row_halves = (size_t)a.KW * a.Npad * KS;
  // The index arithmetic of the staging loops (run-time divisions by the halo width and the chunks per row) cost as many
  // issue cycles as the MFMAs of a stage: a thread's chunks are the same in every stage, so their offsets are computed once.
  //   weights: chunk c = tid + i * NTHR -> LDS offset wdst[i], global offset wsrc[i] inside a kernel row (-1: padding / none)
  //   halo:    the first HFIX chunks of a thread -> LDS offset hdst[i], global offset hsrc[i] at slab 0 (-1: outside the image)
  int wdst[WPRE], wsrc[WPRE];
#pragma unroll
  for (int i = 0; i < WPRE; i++) {
    const int c = tid + i * NTHR;
    wdst[i] = -1; wsrc[i] = -1;
    if (c < wchunks) {
      const int row = c / cpp, qd = c - row * cpp;
      const int dx = row / BN, n = row - dx * BN;
      wdst[i] = row * lp + qd * 8;
      if (nb0 + n < a.Npad) wsrc[i] = (dx * a.Npad + nb0 + n) * KS + qd * 8;
    }
  }
  constexpr int HFIX = 6;
  int hdst[HFIX], hsrc[HFIX];
#pragma unroll
  for (int i = 0; i < HFIX; i++) {
    const int c = tid + i * NTHR;
    hdst[i] = -1; hsrc[i] = -1;
    if (c < hchunks) {
      const int p = c / cpp, qd = c - p * cpp;
      const int hy = p / HW, hx = p - hy * HW;
      const int iy = iy0 + hy, ix = ix0 + hx;
      hdst[i] = (p * lp + qd * 8) | (qd << 24);   // (chunk index kept in the top bits: the slab's valid channels are tested per stage)
      if (iy >= 0 && iy < gi.H && ix >= 0 && ix < gi.W) hsrc[i] = (hy * gi.W + hx) * a.ldx + qd * 8;  // relative to the tile's halo origin
    }
  }
  const half_t* ximg = a.x + gi.off * a.ldx;
  const half_t* xtile = ximg + ((long long)iy0 * gi.W + ix0) * a.ldx;   // (only dereferenced at offsets of pixels inside the image)
  h8 pre[WPRE];
  auto load_row = [&](int rr) {
    const half_t* wg = a.w + (size_t)rr * row_halves;
#pragma unroll
    for (int i = 0; i < WPRE; i++) {
      h8 v = {0, 0, 0, 0, 0, 0, 0, 0};
      if (wsrc[i] >= 0) v = *reinterpret_cast<const h8*>(wg + wsrc[i]);
      pre[i] = v;
    }
  };
  if (prefetch) load_row(0);

  for (int rr = 0; rr < nrows; rr++) {
    const int s = rr / a.KH, dy = rr - s * a.KH;
    const int cvalid = min(KS, a.Cin - s * KS);
    const int ksteps = (cvalid + 15) >> 4;
    const bool stamp = RT_STAMP_ON(a.stamps && blockIdx.y == 0 && blockIdx.x == gridDim.x / 2 && tid == 0);
    if (stamp) a.stamps[rr * 5 + 0] = __builtin_amdgcn_s_memtime();
    __syncthreads();  // every wave is done reading the previous weight row (and, at dy == 0, the previous halo)
    if (stamp) a.stamps[rr * 5 + 1] = __builtin_amdgcn_s_memtime();
    if (dy == 0) {
      // all of a thread's halo loads are issued before the first LDS write (branch-free: chunks outside the image read the
      // tensor's first bytes and are zeroed by a select), so their latencies overlap instead of adding up
      h8 hv[HFIX];
#pragma unroll
      for (int i = 0; i < HFIX; i++) {
        const bool ok = hsrc[i] >= 0 && (hdst[i] >> 24) * 8 < cvalid;
        const half_t* src = ok ? xtile + hsrc[i] + s * KS : a.x;
        hv[i] = *reinterpret_cast<const h8*>(src);
        if (!ok) hv[i] = h8{0, 0, 0, 0, 0, 0, 0, 0};
      }
#pragma unroll
      for (int i = 0; i < HFIX; i++)
        if (hdst[i] >= 0) *reinterpret_cast<h8*>(halo + (hdst[i] & 0xffffff)) = hv[i];
      for (int c = tid + HFIX * NTHR; c < hchunks; c += NTHR) {  // larger halos (9x9 kernels, strided tiles): the remaining chunks
        int p = c / cpp, qd = c - p * cpp;
        int hy = p / HW, hx = p - hy * HW;
        int iy = iy0 + hy, ix = ix0 + hx;
        h8 v = {0, 0, 0, 0, 0, 0, 0, 0};
        if (iy >= 0 && iy < gi.H && ix >= 0 && ix < gi.W && qd * 8 < cvalid)
          v = *reinterpret_cast<const h8*>(ximg + ((long long)iy * gi.W + ix) * a.ldx + s * KS + qd * 8);
        *reinterpret_cast<h8*>(halo + (size_t)p * lp + qd * 8) = v;
      }
    }
    if (prefetch) {
#pragma unroll
      for (int i = 0; i < WPRE; i++)
        if (wdst[i] >= 0) *reinterpret_cast<h8*>(wl + wdst[i]) = pre[i];
      if (rr + 1 < nrows) load_row(rr + 1);
    } else {
      const half_t* wg = a.w + (size_t)rr * row_halves;
      for (int c = tid; c < wchunks; c += NTHR) {
        int row = c / cpp, qd = c - row * cpp;
        int dx = row / BN, n = row - dx * BN;
        h8 v = {0, 0, 0, 0, 0, 0, 0, 0};
        if (nb0 + n < a.Npad) v = *reinterpret_cast<const h8*>(wg + ((size_t)dx * a.Npad + nb0 + n) * KS + qd * 8);
        *reinterpret_cast<h8*>(wl + (size_t)row * lp + qd * 8) = v;
      }
    }
    if (stamp) a.stamps[rr * 5 + 2] = __builtin_amdgcn_s_memtime();
    __syncthreads();
    if (stamp) a.stamps[rr * 5 + 3] = __builtin_amdgcn_s_memtime();
    for (int dx = 0; dx < a.KW; dx++) {
      const int toff = (dy * HW + dx) * lp;
      const half_t* wrow = wl + (size_t)dx * BN * lp + abase;
      for (int ks = 0; ks < ksteps; ks++) {
        h8 A[NTN], B[NTP];
#pragma unroll
        for (int i = 0; i < NTN; i++) A[i] = *reinterpret_cast<const h8*>(wrow + i * 32 * lp + ks * 16);
#pragma unroll
        for (int j = 0; j < NTP; j++) B[j] = *reinterpret_cast<const h8*>(halo + pbase[j] + toff + ks * 16);
#pragma unroll
        for (int i = 0; i < NTN; i++)
#pragma unroll
          for (int j = 0; j < NTP; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[i], B[j], acc[i][j], 0, 0, 0);
      }
    }
    if (stamp) a.stamps[rr * 5 + 4] = __builtin_amdgcn_s_memtime();
  }

  // ---- epilogue: lane = one pixel (column r of the tile), registers = channels (reg & 3) + 8 * (reg >> 2) + 4 * h ----
  static_assert(WN == 1, "the epilogue assumes that a wave holds all BN channels of its pixels");
  if (DOT) {
    dot_tile16<NTN, NTP>(a, acc, lane, oys, oxs, go, (int)blockIdx.y);
    return;
  }
  __syncthreads();   // every wave is done with the last stage's LDS: it becomes the waves' transpose scratch
  store_tile16<NTN, NTP>(a, acc, reinterpret_cast<half_t*>(smem) + (size_t)wid * epi_scratch_halves<NTN * WN>(), lane, nb0, oys, oxs, go);
}


int g_conv16_v2 = 1;   // 1 = LDS-DMA kernels where they apply (default); 0 = register-staged k_conv16 everywhere (A/B)

long long* g_conv_stamps = nullptr;  // diagnostics: device buffer for stage time stamps (rt_debug_conv16 with RT_CONV_STAMPS)
bool conv_stamps_compiled() {
#ifdef RT_CONV_STAMPS_BUILD
  return true;
#else
  return false;
#endif
}

template <int NTN, int DOT>
static void launch_conv16(hipStream_t st, const ConvArgs& a, dim3 grid, size_t lds) {
  auto kfn = k_conv16<NTN, 2, 1, 4, DOT>;
  allow_big_lds(reinterpret_cast<const void*>(kfn), 160 * 1024);   // once per (device, instantiation)
  RT_LAUNCH(kfn, grid, dim3(256), lds, st, a);
}

static int choose_bn(int Npad) {
  int best = 32, best_cost = 1 << 30;
  for (int bn : {160, 128, 96, 64, 32}) {
    int nb = (Npad + bn - 1) / bn;
    int cost = nb * bn + 16 * nb;
    if (cost < best_cost) { best_cost = cost; best = bn; }
  }
  return best;
}

// pixel tile (TH x TW <= 256) for maps of at most maxHo x maxWo
static void choose_tile(int maxHo, int maxWo, int* TH, int* TW) {
  const int BP = 256;
  if (maxHo >= 16 && maxWo >= 16) {
    const int ny = (maxHo + 15) / 16;
    int th = (maxHo + ny - 1) / ny;  // <= 16, even split of the rows
    if (maxHo >= 64) th = 16;
    *TH = th; *TW = std::min(BP / th, std::max(maxWo, 1));
    return;
  }
  int th = std::max(1, std::min(maxHo, 16));
  int tw = std::min(BP / th, std::max(maxWo, 1));
  if (tw == maxWo && th < maxHo) th = std::min(maxHo, BP / std::max(tw, 1));  // narrow maps: spend the tile on rows
  *TH = th; *TW = tw;
}

const char* conv16_label(int KH, int KW, int N, int Cin) {
  if (KH == 1 && KW == 1) return N <= 64 ? "gemm16/thin" : "gemm16";
  // (the stems -- fewer than 32 input channels -- never take the LDS-DMA kernel: their own family, so that `conv16_3x3` is priced and
  //  counter-measured over the same launches)
  if (KH == 3 && KW == 3) return Cin < 32 ? "conv16_stem" : "conv16_3x3";
  if (KH == 9) return "conv16_9x9";
  return "conv16_kxk";
}

void conv16(hipStream_t st, const half_t* x, int ldx, const ImgGeom* gin, const ImgGeom* gout, int n_img, int maxHo, int maxWo,
            int Cin, int KH, int KW, int SH, int SW, int PT, int PL, const half_t* Wp, int N, int Npad, half_t* y, int ldy, int coff,
            const Epi16& epi) {
  if (n_img <= 0 || maxHo <= 0 || maxWo <= 0) return;
  if (Cin % 8 || ldx % 8 || (ldy % 4) || (coff % 4) || Npad % 32) throw RtError(8, "conv16: channel counts / pitches must be multiples of 8 (input) and 4 (output)");
  ConvArgs a;
  a.x = x; a.ldx = ldx; a.gin = gin; a.gout = gout; a.Cin = Cin; a.KH = KH; a.KW = KW; a.SH = SH; a.SW = SW; a.PT = PT; a.PL = PL;
  a.w = Wp; a.N = N; a.Npad = Npad; a.y = y; a.ldy = ldy; a.coff = coff; a.epi = epi;
  a.stamps = g_conv_stamps;
  static const int xcd_env = getenv("RT_XCD") ? atoi(getenv("RT_XCD")) : 1;   // 0: hardware block order (A/B)
  a.xcd = xcd_env;
  a.lp = round_up(std::min(Cin, KS), 16) + 8;  // whole 16-deep k-steps of real (zero-filled) data + one pad chunk
  const bool dot = epi.dot_w != nullptr;
  // ---- LDS-DMA kernels (nn_f16_dma.hip): the 3x3-class layers and the big 1x1 layers ----
  if (g_conv16_v2 && conv16_dma(st, a, n_img, maxHo, maxWo)) return;
  int bn = dot ? Npad : choose_bn(Npad);
  if (dot && Npad > 160) throw RtError(8, "conv16: the dot epilogue needs all output channels in one block (N <= 160)");
  choose_tile(maxHo, maxWo, &a.TH, &a.TW);
  // keep the halo + one weight row within the LDS of a CU (strided / large kernels shrink the tile)
  auto lds_bytes = [&](int th, int tw) {
    size_t hh = (size_t)(th - 1) * SH + KH, hw = (size_t)(tw - 1) * SW + KW;
    return ((hh * hw * a.lp + 7) & ~(size_t)7) * 2 + (size_t)KW * bn * a.lp * 2;
  };
  while (lds_bytes(a.TH, a.TW) > 150 * 1024 && (a.TH > 1 || a.TW > 8)) {
    if (a.TH >= a.TW && a.TH > 1) a.TH = (a.TH + 1) / 2; else a.TW = (a.TW + 1) / 2;
  }
  const size_t lds = std::max(lds_bytes(a.TH, a.TW), (size_t)4 * (32 * (bn + 8) + 128) * 2);  // main loop | epilogue transpose scratch
  if (lds > 160 * 1024) throw RtError(8, "conv16: kernel row does not fit in LDS");
  a.nzb = (Npad + bn - 1) / bn;
  const long long tiles = (long long)((maxWo + a.TW - 1) / a.TW) * ((maxHo + a.TH - 1) / a.TH);
  if (n_img > RT_MAX_GRID_Y) throw RtError(8, "conv16: too many images in one launch");
  dim3 grid((unsigned)(tiles * a.nzb), (unsigned)n_img);
  switch (bn / 32) {
    case 1: dot ? launch_conv16<1, 1>(st, a, grid, lds) : launch_conv16<1, 0>(st, a, grid, lds); break;
    case 2: dot ? launch_conv16<2, 1>(st, a, grid, lds) : launch_conv16<2, 0>(st, a, grid, lds); break;
    case 3: dot ? launch_conv16<3, 1>(st, a, grid, lds) : launch_conv16<3, 0>(st, a, grid, lds); break;
    case 4: dot ? launch_conv16<4, 1>(st, a, grid, lds) : launch_conv16<4, 0>(st, a, grid, lds); break;
    case 5: dot ? launch_conv16<5, 1>(st, a, grid, lds) : launch_conv16<5, 0>(st, a, grid, lds); break;
    default: throw RtError(8, "conv16: unsupported channel block");
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Depthwise KxK: thread = (output pixel, 8 channels), fp32 accumulation in (dy, dx) order, taps / bias from L1.
// ---------------------------------------------------------------------------------------------------------------------
template <int K, int SW>
__global__ __launch_bounds__(256) void k_dw16(int sh, const half_t* __restrict__ x, int ldx, const ImgGeom* __restrict__ gin,
                                              const ImgGeom* __restrict__ gout, int Cp, const half_t* __restrict__ Wd,
                                              const float* __restrict__ bias, int act, int has_lab, float lab_a, float lab_c,
                                              half_t* __restrict__ y, int ldy) {
  // thread = 4 adjacent output pixels x 8 channels: a kernel row needs (4 - 1) * SW + K input vectors for 4 * K taps, so every
  // input vector is loaded once per kernel row instead of once per tap (K = 5: 2 loads per output instead of 5 per row)
  constexpr int PX = 4, NV = (PX - 1) * SW + K, P = K / 2;
  const ImgGeom gi = gin[blockIdx.y], go = gout[blockIdx.y];
  const int C8 = Cp >> 3, wq = (go.W + PX - 1) / PX;
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long long)go.H * wq * C8) return;
  const int c8 = (int)(idx % C8);
  const long long pq = idx / C8;
  const int oy = (int)(pq / wq), ox0 = (int)(pq - (long long)oy * wq) * PX;
  float acc[PX][8];
#pragma unroll
  for (int j = 0; j < PX; j++)
#pragma unroll
    for (int t = 0; t < 8; t++) acc[j][t] = bias[c8 * 8 + t];
#pragma unroll
  for (int dy = 0; dy < K; dy++) {
    const int iy = oy * sh + dy - P;
    if (iy < 0 || iy >= gi.H) continue;
    const half_t* row = x + (gi.off + (long long)iy * gi.W) * ldx + c8 * 8;
    h8 v[NV];
#pragma unroll
    for (int j = 0; j < NV; j++) {
      const int ix = ox0 * SW + j - P;
      v[j] = h8{0, 0, 0, 0, 0, 0, 0, 0};
      if (ix >= 0 && ix < gi.W) v[j] = *reinterpret_cast<const h8*>(row + (long long)ix * ldx);
    }
#pragma unroll
    for (int dx = 0; dx < K; dx++) {
      const h8 w = *reinterpret_cast<const h8*>(Wd + (size_t)(dy * K + dx) * Cp + c8 * 8);
#pragma unroll
      for (int j = 0; j < PX; j++)
#pragma unroll
        for (int t = 0; t < 8; t++) acc[j][t] = fmaf((float)v[j * SW + dx][t], (float)w[t], acc[j][t]);
    }
  }
#pragma unroll
  for (int j = 0; j < PX; j++) {
    if (ox0 + j >= go.W) break;
    h8 o;
#pragma unroll
    for (int t = 0; t < 8; t++) {
      float v = act_f(acc[j][t], act);
      if (has_lab) v = fmaf(v, lab_a, lab_c);
      o[t] = (half_t)v;
    }
    *reinterpret_cast<h8*>(y + (go.off + (long long)oy * go.W + ox0 + j) * ldy + c8 * 8) = o;
  }
}

void dwconv16(hipStream_t st, int K, int sh, int sw, const half_t* x, int ldx, const ImgGeom* gin, const ImgGeom* gout, int n_img,
              int maxHo, int maxWo, int Cp, const half_t* Wd, const float* bias, int act, int has_lab, float lab_a, float lab_c,
              half_t* y, int ldy) {
  if (n_img <= 0) return;
  if (Cp % 8 || ldx % 8 || ldy % 8) throw RtError(8, "dwconv16: channel pitches must be multiples of 8");
  if ((K != 3 && K != 5) || sw < 1 || sw > 2) throw RtError(8, "dwconv16: unsupported kernel size / stride");
  const long long total = (long long)maxHo * ((maxWo + 3) / 4) * (Cp / 8);
  for (int y0 = 0; y0 < n_img; y0 += RT_MAX_GRID_Y) {
    dim3 grid((unsigned)((total + 255) / 256), std::min(n_img - y0, RT_MAX_GRID_Y));
#define RT_DW16(KK, SS) RT_LAUNCH((k_dw16<KK, SS>), grid, dim3(256), 0, st, sh, x, ldx, gin + y0, gout + y0, Cp, Wd, bias, act, has_lab, lab_a, lab_c, y, ldy)
    if (K == 3 && sw == 1) RT_DW16(3, 1); else if (K == 3) RT_DW16(3, 2); else if (sw == 1) RT_DW16(5, 1); else RT_DW16(5, 2);
#undef RT_DW16
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// conversions
// ---------------------------------------------------------------------------------------------------------------------
struct Norm3h { float scale, mean[3], stdv[3]; };
__global__ __launch_bounds__(256) void k_u8_to_h8(const U8Page16* __restrict__ pages, Norm3h nm, half_t* __restrict__ out) {
  const U8Page16 d = pages[blockIdx.y];
  const long long p = (long long)blockIdx.x * 256 + threadIdx.x;
  if (p >= d.npix) return;
  const uint8_t* s = d.rgb + p * 3;
  h8 o = {0, 0, 0, 0, 0, 0, 0, 0};
  // rgb2bgr + normalize (det_processor.rs:151-155, image_helper.rs:211-221): channel 0 = B
#pragma unroll
  for (int c = 0; c < 3; c++) {
    float xv = (float)s[2 - c];
    o[c] = (half_t)((xv * nm.scale - nm.mean[c]) / nm.stdv[c]);
  }
  *reinterpret_cast<h8*>(out + (d.out_pix + p) * 8) = o;
}
void u8_to_h8(hipStream_t st, const U8Page16* pages, int n, long long max_pix, float scale, const float* mean3, const float* std3,
              half_t* out) {
  if (n <= 0 || max_pix <= 0) return;
  Norm3h nm; nm.scale = scale;
  for (int i = 0; i < 3; i++) { nm.mean[i] = mean3[i]; nm.stdv[i] = std3[i]; }
  RT_LAUNCH(k_u8_to_h8, dim3((unsigned)((max_pix + 255) / 256), n), dim3(256), 0, st, pages, nm, out);
}
__global__ __launch_bounds__(256) void k_f32x4_to_h8(const float* __restrict__ in, long long npix, half_t* __restrict__ out) {
  const long long p = (long long)blockIdx.x * 256 + threadIdx.x;
  if (p >= npix) return;
  const f32x4 v = *reinterpret_cast<const f32x4*>(in + p * 4);
  h8 o = {(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3], 0, 0, 0, 0};
  *reinterpret_cast<h8*>(out + p * 8) = o;
}
void f32x4_to_h8(hipStream_t st, const float* in, long long npix, half_t* out) {
  if (npix <= 0) return;
  RT_LAUNCH(k_f32x4_to_h8, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, st, in, npix, out);
}
__global__ __launch_bounds__(256) void k_h_to_f32(const half_t* __restrict__ src, int lds, long long rows, int C, float* __restrict__ dst,
                                                  int ldd, int coff) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= rows * C) return;
  const long long rrow = i / C; const int c = (int)(i - rrow * C);
  dst[rrow * ldd + coff + c] = (float)src[rrow * lds + c];
}
void h_to_f32(hipStream_t st, const half_t* src, int lds, long long rows, int C, float* dst, int ldd, int coff) {
  if (rows <= 0) return;
  RT_LAUNCH(k_h_to_f32, dim3((unsigned)((rows * C + 255) / 256)), dim3(256), 0, st, src, lds, rows, C, dst, ldd, coff);
}
__global__ __launch_bounds__(256) void k_f32_to_h(const float* __restrict__ src, int lds, long long rows, int C, half_t* __restrict__ dst,
                                                  int ldd, int coff) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= rows * C) return;
  const long long rrow = i / C; const int c = (int)(i - rrow * C);
  dst[rrow * ldd + coff + c] = (half_t)src[rrow * lds + c];
}
void f32_to_h(hipStream_t st, const float* src, int lds, long long rows, int C, half_t* dst, int ldd, int coff) {
  if (rows <= 0) return;
  RT_LAUNCH(k_f32_to_h, dim3((unsigned)((rows * C + 255) / 256)), dim3(256), 0, st, src, lds, rows, C, dst, ldd, coff);
}

// ---------------------------------------------------------------------------------------------------------------------
// squeeze-excite / ESE: deterministic partial sums (fixed pixel chunks, fixed order), then the small FCs per image
// ---------------------------------------------------------------------------------------------------------------------
constexpr int POOL_PIX16 = 1024;
int pool_chunks16(long long max_pix) { return (int)((max_pix + POOL_PIX16 - 1) / POOL_PIX16); }

__global__ __launch_bounds__(256) void k_pool_partial16(const half_t* __restrict__ x, int ldx, const ImgGeom* __restrict__ geom, int Cp,
                                                        int chunks, float* __restrict__ partial) {
  __shared__ float red[256 * 8];
  const ImgGeom g = geom[blockIdx.y];
  const long long npix = (long long)g.H * g.W;
  const long long p0 = (long long)blockIdx.x * POOL_PIX16;
  const int C8 = Cp >> 3;
  float* out = partial + ((long long)blockIdx.y * chunks + blockIdx.x) * Cp;
  for (int cbase = 0; cbase < C8; cbase += 256) {
    const int cgroups = min(256, C8 - cbase);
    const int PL = 256 / cgroups;
    const int c8 = cbase + (threadIdx.x % cgroups), pl = threadIdx.x / cgroups;
    float s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (pl < PL && p0 < npix) {
      const long long pend = min(npix, p0 + POOL_PIX16);
      // four pixels per trip: four independent 16-byte loads in flight per thread (one load per trip kept the kernel at
      // 3 TB/s with the waves parked on the load 93 % of their cycles), summed in a fixed order
      const half_t* xp = x + (g.off + p0 + pl) * ldx + c8 * 8;
      const long long step = (long long)PL * ldx;
      long long p = p0 + pl;
      for (; p + 3 * PL < pend; p += 4 * PL, xp += 4 * step) {
        const h8 v0 = *reinterpret_cast<const h8*>(xp), v1 = *reinterpret_cast<const h8*>(xp + step);
        const h8 v2 = *reinterpret_cast<const h8*>(xp + 2 * step), v3 = *reinterpret_cast<const h8*>(xp + 3 * step);
#pragma unroll
        for (int t = 0; t < 8; t++) s[t] += ((float)v0[t] + (float)v1[t]) + ((float)v2[t] + (float)v3[t]);
      }
      for (; p < pend; p += PL, xp += step) {
        const h8 v = *reinterpret_cast<const h8*>(xp);
#pragma unroll
        for (int t = 0; t < 8; t++) s[t] += (float)v[t];
      }
    }
#pragma unroll
    for (int t = 0; t < 8; t++) red[threadIdx.x * 8 + t] = s[t];
    __syncthreads();
    if (threadIdx.x < cgroups) {
      float t8[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      for (int l = 0; l < PL; l++)
#pragma unroll
        for (int t = 0; t < 8; t++) t8[t] += red[(l * cgroups + threadIdx.x) * 8 + t];
#pragma unroll
      for (int t = 0; t < 8; t++) out[(cbase + threadIdx.x) * 8 + t] = t8[t];
    }
    __syncthreads();
  }
}

// block per group of IPB images: mean -> [fc1 -> relu] -> fc2 -> gate.  The FC weights are read once per block and applied to
// all of its images (the recognition net's ESE layers have C x C weights of up to 4 MB and ~1000 images per launch: one
// block per image re-read them from L2 a thousand times).
constexpr int SE_IPB = 8;
__global__ __launch_bounds__(256) void k_se_fc16(const float* __restrict__ partial, const ImgGeom* __restrict__ geom, int n_img,
                                                 int chunks_alloc, int C, int Cp, const float* __restrict__ w1t,
                                                 const float* __restrict__ b1, const float* __restrict__ w2t,
                                                 const float* __restrict__ b2, int Cr, float slope, int residual,
                                                 float* __restrict__ scale) {
  extern __shared__ float sm16[];  // mean[IPB][Cp] + hid[IPB][Cr]
  float* mean = sm16;
  float* hid = sm16 + SE_IPB * Cp;
  const int img0 = blockIdx.x * SE_IPB, ni = min(SE_IPB, n_img - img0);
  for (int i = 0; i < ni; i++) {
    const ImgGeom g = geom[img0 + i];
    const long long npix = (long long)g.H * g.W;
    const int chunks = (int)((npix + POOL_PIX16 - 1) / POOL_PIX16);
    const float inv = 1.0f / (float)npix;
    for (int c = threadIdx.x; c < Cp; c += 256) {
      float s = 0.f;
      for (int k = 0; k < chunks; k++) s += partial[((long long)(img0 + i) * chunks_alloc + k) * Cp + c];
      mean[i * Cp + c] = s * inv;
    }
  }
  __syncthreads();
  if (w2t == nullptr) {  // plain global mean
    for (int i = 0; i < ni; i++)
      for (int c = threadIdx.x; c < Cp; c += 256) scale[(long long)(img0 + i) * Cp + c] = mean[i * Cp + c];
    return;
  }
  const float* hin = mean;
  int hpitch = Cp;
  if (w1t) {
    for (int j = threadIdx.x; j < Cr; j += 256) {  // w1t [C][Cr]
      float s[SE_IPB];
#pragma unroll
      for (int i = 0; i < SE_IPB; i++) s[i] = b1[j];
      for (int c = 0; c < C; c++) {
        const float w = w1t[(size_t)c * Cr + j];
#pragma unroll
        for (int i = 0; i < SE_IPB; i++) s[i] = fmaf(mean[i * Cp + c], w, s[i]);
      }
#pragma unroll
      for (int i = 0; i < SE_IPB; i++) hid[i * Cr + j] = fmaxf(s[i], 0.f);
    }
    __syncthreads();
    hin = hid; hpitch = Cr;
  }
  for (int c = threadIdx.x; c < Cp; c += 256) {  // w2t [Cr][C]
    float s[SE_IPB];
#pragma unroll
    for (int i = 0; i < SE_IPB; i++) s[i] = c < C ? b2[c] : 0.f;
    if (c < C)
      for (int j = 0; j < Cr; j++) {
        const float w = w2t[(size_t)j * C + c];
#pragma unroll
        for (int i = 0; i < SE_IPB; i++) s[i] = fmaf(hin[i * hpitch + j], w, s[i]);
      }
    for (int i = 0; i < ni; i++) {
      float o = 0.f;
      if (c < C) {
        o = slope < 0.f ? 1.f / (1.f + __expf(-s[i])) : fminf(fmaxf(fmaf(s[i], slope, 0.5f), 0.f), 1.f);
        if (residual) o += 1.0f;
      }
      scale[(long long)(img0 + i) * Cp + c] = o;
    }
  }
}

void se_scale16(hipStream_t st, const half_t* x, int ldx, const ImgGeom* geom, int n_img, long long max_pix, int C, int Cp, const float* w1t,
                const float* b1, const float* w2t, const float* b2, int Cr, float slope, int residual, float* partial,
                float* scale) {
  if (n_img <= 0) return;
  const int chunks = pool_chunks16(max_pix);
  for (int y0 = 0; y0 < n_img; y0 += RT_MAX_GRID_Y) {
    const int ny = std::min(n_img - y0, RT_MAX_GRID_Y);
    RT_LAUNCH(k_pool_partial16, dim3(chunks, ny), dim3(256), 0, st, x, ldx, geom + y0, Cp, chunks, partial + (size_t)y0 * chunks * Cp);
  }
  const size_t lds = (size_t)SE_IPB * (Cp + Cr + 4) * sizeof(float);
  allow_big_lds(reinterpret_cast<const void*>(k_se_fc16), 96 * 1024);
  RT_LAUNCH(k_se_fc16, dim3((n_img + SE_IPB - 1) / SE_IPB), dim3(256), lds, st, partial, geom, n_img, chunks, C, Cp, w1t, b1, w2t,
            b2, Cr, slope, residual, scale);
}
// scale[img][c] = gate(s[img][c]) (+1 when residual), 0 on the pad channels
__global__ __launch_bounds__(256) void k_gate16(const float* __restrict__ sv, int lds_, int n_img, int C, int Cp, float slope, int residual,
                                                float* __restrict__ scale) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n_img * Cp) return;
  const int img = i / Cp, c = i - img * Cp;
  float o = 0.f;
  if (c < C) {
    const float v = sv[(size_t)img * lds_ + c];
    o = slope < 0.f ? 1.f / (1.f + __expf(-v)) : fminf(fmaxf(fmaf(v, slope, 0.5f), 0.f), 1.f);
    if (residual) o += 1.0f;
  }
  scale[i] = o;
}
void gate16(hipStream_t st, const float* s, int lds_, int n_img, int C, int Cp, float slope, int residual, float* scale) {
  if (n_img <= 0) return;
  RT_LAUNCH(k_gate16, dim3((unsigned)((n_img * Cp + 255) / 256)), dim3(256), 0, st, s, lds_, n_img, C, Cp, slope, residual, scale);
}
void global_mean16(hipStream_t st, const half_t* x, int ldx, const ImgGeom* geom, int n_img, long long max_pix, int Cp, float* partial,
                   float* out) {
  se_scale16(st, x, ldx, geom, n_img, max_pix, Cp, Cp, nullptr, nullptr, nullptr, nullptr, 0, 0.f, 0, partial, out);
}

__global__ __launch_bounds__(256) void k_scale_channels16(const half_t* __restrict__ x, int ldx, const ImgGeom* __restrict__ geom, int Cp,
                                                          const float* __restrict__ scale, const half_t* __restrict__ res, int ldr,
                                                          half_t* __restrict__ y, int ldy) {
  const ImgGeom g = geom[blockIdx.y];
  const int C8 = Cp >> 3;
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long long)g.H * g.W * C8) return;
  const int c8 = (int)(idx % C8);
  const long long pix = g.off + idx / C8;
  const h8 v = *reinterpret_cast<const h8*>(x + pix * ldx + c8 * 8);
  const float* s = scale + (long long)blockIdx.y * Cp + c8 * 8;
  h8 o;
  if (res) {
    const h8 rv = *reinterpret_cast<const h8*>(res + pix * ldr + c8 * 8);
#pragma unroll
    for (int t = 0; t < 8; t++) o[t] = (half_t)fmaf((float)v[t], s[t], (float)rv[t]);
  } else {
#pragma unroll
    for (int t = 0; t < 8; t++) o[t] = (half_t)((float)v[t] * s[t]);
  }
  *reinterpret_cast<h8*>(y + pix * ldy + c8 * 8) = o;
}
void scale_channels16(hipStream_t st, const half_t* x, int ldx, const ImgGeom* geom, int n_img, long long max_pix, int Cp, const float* scale,
                      const half_t* res, int ldr, half_t* y, int ldy) {
  if (n_img <= 0) return;
  const long long total = max_pix * (Cp / 8);
  for (int y0 = 0; y0 < n_img; y0 += RT_MAX_GRID_Y)
    RT_LAUNCH(k_scale_channels16, dim3((unsigned)((total + 255) / 256), std::min(n_img - y0, RT_MAX_GRID_Y)), dim3(256), 0, st, x, ldx,
              geom + y0, Cp, scale + (size_t)y0 * Cp, res, ldr, y, ldy);
}

// ---------------------------------------------------------------------------------------------------------------------
// spatial glue
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_upsample_add16(const half_t* __restrict__ a, const half_t* __restrict__ b,
                                                        const ImgGeom* __restrict__ ga, const ImgGeom* __restrict__ gb, int Cp,
                                                        half_t* __restrict__ out, const float* __restrict__ scale_a) {
  const ImgGeom A = ga[blockIdx.y], B = gb[blockIdx.y];
  const int C8 = Cp >> 3;
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long long)A.H * A.W * C8) return;
  const int c8 = (int)(idx % C8);
  const long long p = idx / C8;
  const int y = (int)(p / A.W), x = (int)(p - (long long)y * A.W);
  const int by = min(y >> 1, B.H - 1), bx = min(x >> 1, B.W - 1);
  const h8 va = *reinterpret_cast<const h8*>(a + (A.off + p) * Cp + c8 * 8);
  const h8 vb = *reinterpret_cast<const h8*>(b + (B.off + (long long)by * B.W + bx) * Cp + c8 * 8);
  h8 o;
  if (scale_a) {
    const float* s = scale_a + (long long)blockIdx.y * Cp + c8 * 8;
#pragma unroll
    for (int t = 0; t < 8; t++) o[t] = (half_t)fmaf((float)va[t], s[t], (float)vb[t]);
  } else {
#pragma unroll
    for (int t = 0; t < 8; t++) o[t] = (half_t)((float)va[t] + (float)vb[t]);
  }
  *reinterpret_cast<h8*>(out + (A.off + p) * Cp + c8 * 8) = o;
}
void upsample_add16(hipStream_t st, const half_t* a, const half_t* b, const ImgGeom* ga, const ImgGeom* gb, int n_img,
                    long long max_pix, int Cp, half_t* out, const float* scale_a) {
  if (n_img <= 0) return;
  const long long total = max_pix * (Cp / 8);
  RT_LAUNCH(k_upsample_add16, dim3((unsigned)((total + 255) / 256), n_img), dim3(256), 0, st, a, b, ga, gb, Cp, out, scale_a);
}

__global__ __launch_bounds__(256) void k_upsample_into16(const half_t* __restrict__ src, int lds, const ImgGeom* __restrict__ gsrc,
                                                         const ImgGeom* __restrict__ gdst, int C, int shift, half_t* __restrict__ dst,
                                                         int ldd, int coff, const float* __restrict__ scale) {
  const ImgGeom S = gsrc[blockIdx.y], D = gdst[blockIdx.y];
  const int C8 = C >> 3;
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long long)D.H * D.W * C8) return;
  const int c8 = (int)(idx % C8);
  const long long p = idx / C8;
  const int y = (int)(p / D.W), x = (int)(p - (long long)y * D.W);
  const int sy = min(y >> shift, S.H - 1), sx = min(x >> shift, S.W - 1);
  h8 v = *reinterpret_cast<const h8*>(src + (S.off + (long long)sy * S.W + sx) * lds + c8 * 8);
  if (scale) {
    const float* s = scale + (long long)blockIdx.y * lds + c8 * 8;
#pragma unroll
    for (int t = 0; t < 8; t++) v[t] = (half_t)((float)v[t] * s[t]);
  }
  *reinterpret_cast<h8*>(dst + (D.off + p) * ldd + coff + c8 * 8) = v;
}
void upsample_into16(hipStream_t st, const half_t* src, int lds, const ImgGeom* gsrc, const ImgGeom* gdst, int n_img,
                     long long max_pix, int C, int shift, half_t* dst, int ldd, int coff, const float* scale) {
  if (n_img <= 0) return;
  if (C % 8 || coff % 8 || ldd % 8 || lds % 8) throw RtError(8, "upsample_into16: channels must be multiples of 8");
  const long long total = max_pix * (C / 8);
  RT_LAUNCH(k_upsample_into16, dim3((unsigned)((total + 255) / 256), n_img), dim3(256), 0, st, src, lds, gsrc, gdst, C, shift, dst,
            ldd, coff, scale);
}

__global__ __launch_bounds__(256) void k_add16(const half_t* __restrict__ a, const half_t* __restrict__ b, long long n8,
                                               half_t* __restrict__ out) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n8) return;
  const h8 va = reinterpret_cast<const h8*>(a)[i], vb = reinterpret_cast<const h8*>(b)[i];
  h8 o;
#pragma unroll
  for (int t = 0; t < 8; t++) o[t] = (half_t)((float)va[t] + (float)vb[t]);
  reinterpret_cast<h8*>(out)[i] = o;
}
void add16(hipStream_t st, const half_t* a, const half_t* b, long long n_halves, half_t* out) {
  if (n_halves <= 0) return;
  const long long n8 = n_halves / 8;
  RT_LAUNCH(k_add16, dim3((unsigned)((n8 + 255) / 256)), dim3(256), 0, st, a, b, n8, out);
}

__global__ __launch_bounds__(256) void k_maxpool16(const half_t* __restrict__ x, int ldx, const ImgGeom* __restrict__ gin,
                                                   const ImgGeom* __restrict__ gout, int Cp, int kh, int kw, int sh, int sw, int ph,
                                                   int pw, half_t* __restrict__ y, int ldy) {
  const ImgGeom gi = gin[blockIdx.y], go = gout[blockIdx.y];
  const int C8 = Cp >> 3;
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long long)go.H * go.W * C8) return;
  const int c8 = (int)(idx % C8);
  const long long p = idx / C8;
  const int oy = (int)(p / go.W), ox = (int)(p - (long long)oy * go.W);
  float m[8];
#pragma unroll
  for (int t = 0; t < 8; t++) m[t] = -INFINITY;
  for (int dy = 0; dy < kh; dy++) {
    const int iy = oy * sh + dy - ph;
    if (iy < 0 || iy >= gi.H) continue;
    for (int dx = 0; dx < kw; dx++) {
      const int ix = ox * sw + dx - pw;
      if (ix < 0 || ix >= gi.W) continue;
      const h8 v = *reinterpret_cast<const h8*>(x + (gi.off + (long long)iy * gi.W + ix) * ldx + c8 * 8);
#pragma unroll
      for (int t = 0; t < 8; t++) m[t] = fmaxf(m[t], (float)v[t]);
    }
  }
  h8 o;
#pragma unroll
  for (int t = 0; t < 8; t++) o[t] = (half_t)m[t];
  *reinterpret_cast<h8*>(y + (go.off + p) * ldy + c8 * 8) = o;
}
void maxpool16(hipStream_t st, const half_t* x, int ldx, const ImgGeom* gin, const ImgGeom* gout, int n_img, long long max_pix, int Cp,
               int kh, int kw, int sh, int sw, int ph, int pw, half_t* y, int ldy) {
  if (n_img <= 0) return;
  const long long total = max_pix * (Cp / 8);
  for (int y0 = 0; y0 < n_img; y0 += RT_MAX_GRID_Y)
    RT_LAUNCH(k_maxpool16, dim3((unsigned)((total + 255) / 256), std::min(n_img - y0, RT_MAX_GRID_Y)), dim3(256), 0, st, x, ldx,
              gin + y0, gout + y0, Cp, kh, kw, sh, sw, ph, pw, y, ldy);
}

__global__ __launch_bounds__(256) void k_avgpool16_to_f32(const half_t* __restrict__ x, const ImgGeom* __restrict__ gin,
                                                          const ImgGeom* __restrict__ gout, int C, int Cp, int kh, int kw,
                                                          float* __restrict__ y, int ldy) {
  const ImgGeom gi = gin[blockIdx.y], go = gout[blockIdx.y];
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long long)go.H * go.W * C) return;
  const int c = (int)(idx % C);
  const long long p = idx / C;
  const int oy = (int)(p / go.W), ox = (int)(p - (long long)oy * go.W);
  float s = 0.f;
  for (int dy = 0; dy < kh; dy++)
    for (int dx = 0; dx < kw; dx++) s += (float)x[(gi.off + (long long)(oy * kh + dy) * gi.W + ox * kw + dx) * Cp + c];
  y[(go.off + p) * ldy + c] = s / (float)(kh * kw);
}
void avgpool16_to_f32(hipStream_t st, const half_t* x, const ImgGeom* gin, const ImgGeom* gout, int n_img, long long max_pix, int C,
                      int Cp, int kh, int kw, float* y, int ldy) {
  if (n_img <= 0 || max_pix <= 0) return;
  const long long total = max_pix * C;
  for (int y0 = 0; y0 < n_img; y0 += RT_MAX_GRID_Y)
    RT_LAUNCH(k_avgpool16_to_f32, dim3((unsigned)((total + 255) / 256), std::min(n_img - y0, RT_MAX_GRID_Y)), dim3(256), 0, st, x,
              gin + y0, gout + y0, C, Cp, kh, kw, y, ldy);
}

__global__ __launch_bounds__(256) void k_avgpool16(const half_t* __restrict__ x, int ldx, const ImgGeom* __restrict__ gin,
                                                   const ImgGeom* __restrict__ gout, int Cp, int kh, int kw, half_t* __restrict__ y,
                                                   int ldy) {
  const ImgGeom gi = gin[blockIdx.y], go = gout[blockIdx.y];
  const int C8 = Cp >> 3;
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long long)go.H * go.W * C8) return;
  const int c8 = (int)(idx % C8);
  const long long p = idx / C8;
  const int oy = (int)(p / go.W), ox = (int)(p - (long long)oy * go.W);
  float s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int dy = 0; dy < kh; dy++)
    for (int dx = 0; dx < kw; dx++) {
      const h8 v = *reinterpret_cast<const h8*>(x + (gi.off + (long long)(oy * kh + dy) * gi.W + ox * kw + dx) * ldx + c8 * 8);
#pragma unroll
      for (int t = 0; t < 8; t++) s[t] += (float)v[t];
    }
  const float inv = 1.f / (float)(kh * kw);
  h8 o;
#pragma unroll
  for (int t = 0; t < 8; t++) o[t] = (half_t)(s[t] * inv);
  *reinterpret_cast<h8*>(y + (go.off + p) * ldy + c8 * 8) = o;
}
void avgpool16(hipStream_t st, const half_t* x, int ldx, const ImgGeom* gin, const ImgGeom* gout, int n_img, long long max_pix, int Cp,
               int kh, int kw, half_t* y, int ldy) {
  if (n_img <= 0 || max_pix <= 0) return;
  const long long total = max_pix * (Cp / 8);
  for (int y0 = 0; y0 < n_img; y0 += RT_MAX_GRID_Y)
    RT_LAUNCH(k_avgpool16, dim3((unsigned)((total + 255) / 256), std::min(n_img - y0, RT_MAX_GRID_Y)), dim3(256), 0, st, x, ldx,
              gin + y0, gout + y0, Cp, kh, kw, y, ldy);
}

__global__ __launch_bounds__(256) void k_pixel_shuffle16(const half_t* __restrict__ src, int lds, const ImgGeom* __restrict__ gsrc,
                                                         const ImgGeom* __restrict__ gdst, int C, half_t* __restrict__ dst, int ldd,
                                                         int coff) {
  const ImgGeom S = gsrc[blockIdx.y], D = gdst[blockIdx.y];
  const int C8 = C >> 3;
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long long)D.H * D.W * C8) return;
  const int c8 = (int)(idx % C8);
  const long long p = idx / C8;
  const int y = (int)(p / D.W), x = (int)(p - (long long)y * D.W);
  const int q = (y & 1) * 2 + (x & 1);
  const h8 v = *reinterpret_cast<const h8*>(src + (S.off + (long long)(y >> 1) * S.W + (x >> 1)) * lds + q * C + c8 * 8);
  *reinterpret_cast<h8*>(dst + (D.off + p) * ldd + coff + c8 * 8) = v;
}
void pixel_shuffle16(hipStream_t st, const half_t* src, int lds, const ImgGeom* gsrc, const ImgGeom* gdst, int n_img,
                     long long max_pix, int C, half_t* dst, int ldd, int coff) {
  if (n_img <= 0) return;
  if (C % 8 || coff % 8 || ldd % 8 || lds % 8) throw RtError(8, "pixel_shuffle16: channels must be multiples of 8");
  const long long total = max_pix * (C / 8);
  RT_LAUNCH(k_pixel_shuffle16, dim3((unsigned)((total + 255) / 256), n_img), dim3(256), 0, st, src, lds, gsrc, gdst, C, dst, ldd, coff);
}

// 8 lanes = one pixel of f (half resolution), lane g of the group takes the channels 8g .. 8g+7 (+64, ...): a pixel's channels are
// read as one contiguous run (one thread per pixel walking its own row touched 64 cache lines per load: 7x over-fetch by PMC);
// the four partial sums are combined by three xor-shuffles (fixed order) and lanes 0-3 of the group store one output each
__global__ __launch_bounds__(256) void k_deconv_to_map16(const half_t* __restrict__ f, int ldf, const ImgGeom* __restrict__ gf,
                                                         const ImgGeom* __restrict__ gmap, int C, const float* __restrict__ w, float b,
                                                         float* __restrict__ map) {
  const ImgGeom F = gf[blockIdx.y], M = gmap[blockIdx.y];
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long p = t >> 3;
  const int g = (int)(t & 7);
  const bool live = p < (long long)F.H * F.W;     // (whole groups are live or not: no divergence inside a shuffle group)
  float s[4] = {0.f, 0.f, 0.f, 0.f};
  if (live) {
    const half_t* src = f + (F.off + p) * ldf;
    for (int c = g * 8; c < C; c += 64) {
      const h8 v = *reinterpret_cast<const h8*>(src + c);
#pragma unroll
      for (int k = 0; k < 8; k++) {
        const f32x4 wv = *reinterpret_cast<const f32x4*>(w + (size_t)(c + k) * 4);
#pragma unroll
        for (int q = 0; q < 4; q++) s[q] = fmaf((float)v[k], wv[q], s[q]);
      }
    }
  }
#pragma unroll
  for (int d = 1; d < 8; d <<= 1)
#pragma unroll
    for (int q = 0; q < 4; q++) s[q] += __shfl_xor(s[q], d);
  if (live && g < 4) {
    const int y = (int)(p / F.W), x = (int)(p - (long long)y * F.W);
    const float v = g == 0 ? s[0] : g == 1 ? s[1] : g == 2 ? s[2] : s[3];
    map[M.off + (long long)(2 * y + (g >> 1)) * M.W + 2 * x + (g & 1)] = 1.f / (1.f + __expf(-(v + b)));
  }
}
void deconv_to_map16(hipStream_t st, const half_t* f, int ldf, const ImgGeom* gf, const ImgGeom* gmap, int n_img, long long max_pix,
                     int C, const float* w, float b, float* map) {
  if (n_img <= 0) return;
  if (C % 8) throw RtError(8, "deconv_to_map16: C must be a multiple of 8");
  RT_LAUNCH(k_deconv_to_map16, dim3((unsigned)((max_pix * 8 + 255) / 256), n_img), dim3(256), 0, st, f, ldf, gf, gmap, C, w, b, map);
}

__global__ __launch_bounds__(256) void k_map_window16(const float* __restrict__ map, const ImgGeom* __restrict__ gmap,
                                                      const ImgGeom* __restrict__ gf, half_t* __restrict__ dst, int ldd, int coff) {
  const ImgGeom F = gf[blockIdx.y], M = gmap[blockIdx.y];
  const long long p = (long long)blockIdx.x * 256 + threadIdx.x;
  if (p >= (long long)F.H * F.W) return;
  const int y = (int)(p / F.W), x = (int)(p - (long long)y * F.W);
  h8 lo, hi;
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int my = 2 * y - 1 + i, mx = 2 * x - 1 + j;
      float v = 0.f;
      if (my >= 0 && my < M.H && mx >= 0 && mx < M.W) v = map[M.off + (long long)my * M.W + mx];
      const int k = 4 * i + j;
      if (k < 8) lo[k] = (half_t)v; else hi[k - 8] = (half_t)v;
    }
  half_t* d = dst + (F.off + p) * ldd + coff;
  *reinterpret_cast<h8*>(d) = lo;
  *reinterpret_cast<h8*>(d + 8) = hi;
}
void map_window16(hipStream_t st, const float* map, const ImgGeom* gmap, const ImgGeom* gf, int n_img, long long max_pix,
                  half_t* dst, int ldd, int coff) {
  if (n_img <= 0) return;
  if (coff % 8 || ldd % 8) throw RtError(8, "map_window16: channel offsets must be multiples of 8");
  RT_LAUNCH(k_map_window16, dim3((unsigned)((max_pix + 255) / 256), n_img), dim3(256), 0, st, map, gmap, gf, dst, ldd, coff);
}

}  // namespace nh
}  // namespace rt
