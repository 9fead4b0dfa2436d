// fp16-storage / fp32-accumulate kernels for gfx950 (see nn_f16.h).  Written for CDNA4 only:
// v_mfma_f32_32x32x16_f16, 64-wide waves, ds_read_b128 fragment reads from padded LDS rows.
#include "nn_f16.h"

#include <algorithm>
#include <cmath>

namespace rt {
namespace nh {

// In-kernel time stamps (tools/conv16_stamps.py) exist only in a diagnostic build (make STAMPS=1): even behind a false
// run-time flag an s_memtime in the stage loop is a pending scalar-memory event to hipcc's wait-count pass, which then
// writes lgkmcnt(0) before every MFMA group and serialises the software-pipelined fragment reads.
#ifdef RT_CONV_STAMPS_BUILD
#define RT_STAMP_ON(expr) (expr)
#else
#define RT_STAMP_ON(expr) false
#endif

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float act_f(float v, int act) {
  switch (act) {
    case ACT_RELU: return fmaxf(v, 0.f);
    case ACT_HSWISH: return v * fminf(fmaxf(v + 3.f, 0.f), 6.f) * (1.f / 6.f);
    case ACT_SWISH: return v / (1.f + __expf(-v));
    case ACT_SIGMOID: return 1.f / (1.f + __expf(-v));
    default: return v;
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Dense convolution as implicit GEMM on v_mfma_f32_32x32x16_f16.
//
//   D[n][pixel] += W[n][k] . X[k][pixel]      A operand = weights (rows = output channels), B operand = activations
//
// so that a lane ends up with 16 output channels of ONE pixel (4 groups of 4 consecutive channels -> 8-byte stores into the
// NHWC result).  A workgroup owns a TH x TW pixel tile (TH * TW <= BP = 32 * NTP * WP) of one image and BN = 32 * NTN * WN
// output channels; its waves are WN x WP, each 32 * NTN channels x 32 * NTP pixels.  K runs over 32-channel slabs of the
// input and, inside a slab, over the kernel rows:
//   * the (TH-1)*SH+KH x (TW-1)*SW+KW halo tile of the slab is staged once in LDS ([pixel][32 + 8 halves]: 80-byte rows make
//     the 16-byte fragment reads of 16 consecutive pixels conflict-free) and shared by all KH * KW taps;
//   * the weights of one kernel row (KW taps x BN channels x 32) are staged per row, the next row prefetched into registers
//     while the MFMAs of the current one run;
//   * per tap and 16-deep k-step a wave reads NTN + NTP fragments for NTN * NTP MFMAs.
// Pixel tiles are TH x TW with run-time TW (not a power of two: the 3 / 6 / 12 / 24-row maps of the recognition net take
// full-height tiles), a 1x1 conv over a whole batch runs as one "image" of 1 x M pixels.
// ---------------------------------------------------------------------------------------------------------------------
struct ConvArgs {
  const half_t* x; int ldx;
  const ImgGeom* gin; const ImgGeom* gout;
  int Cin, KH, KW, SH, SW, PT, PL;
  const half_t* w; int N, Npad;
  half_t* y; int ldy, coff;
  int TH, TW, nzb;   // pixel tile, number of channel blocks (fastest block coordinate: neighbours share the input in L2)
  int lp;            // LDS row pitch in halves: min(Cin, 32) rounded up to 16, + 8
  long long* stamps; // diagnostics (RT_CONV_STAMPS): s_memtime of wave 0 at 5 points of every stage of one workgroup, or null
  Epi16 epi;
};

// Epilogue of both conv kernels: bias / activation / LAB / residual on the accumulators, fp16 NHWC store.
// After the MFMAs a lane holds 16 channels of ONE pixel in four runs of 4: storing them directly is 8-byte pieces at a pixel
// pitch (every wave-store touches 64 different cache lines; measured 40 k cycles for a 512 x 128 tile, as long as the 12
// MFMA stages of a 3x3 128->128 layer).  So each wave transposes its 32-pixel fragments through a private LDS scratch
// ([pixel][BN + 8] halves) and stores 16 bytes per lane with consecutive lanes on consecutive channels of one pixel: whole
// 64 ... 256-byte channel runs per pixel.  Taken when the output pitch and channel offset are multiples of 8 and there is
// no residual (the residual form adds in fp32 before the one rounding, per lane, as before); same values either way.
// scratch: wave-private, 32 * (BN + 8) halves + 32 long long; the caller has synchronised the workgroup after its last
// main-loop LDS read.
template <int NTN>
__device__ __forceinline__ size_t epi_scratch_halves() { return (size_t)32 * (32 * NTN + 8) + 128; }

template <int ACT>
__device__ __forceinline__ float act_c(float v) {   // activation known at compile time: no per-element branch
  if (ACT == ACT_RELU) return fmaxf(v, 0.f);
  if (ACT == ACT_HSWISH) return v * fminf(fmaxf(v + 3.f, 0.f), 6.f) * (1.f / 6.f);
  if (ACT == ACT_SWISH) return v / (1.f + __expf(-v));
  if (ACT == ACT_SIGMOID) return 1.f / (1.f + __expf(-v));
  return v;
}

// ACT: the activation; EDGE: this channel block holds the last real output channel (values beyond N are forced to zero);
// WIDE: LDS-transposed 16-byte stores (no residual, pitches multiples of 8), else 8-byte stores per lane with the residual.
// All three are workgroup-uniform and resolved ONCE (store_tile16 below): with run-time tests inside the 16 * NTN * NTP
// element loops the compiler emitted ~1300 scalar branches and the epilogue of a 512 x 128 tile ran 40 k cycles.
template <int NTN, int NTP, int ACT, bool EDGE, bool WIDE>
__device__ __forceinline__ void store_tile16_t(const ConvArgs& a, const f32x16 (&acc)[NTN][NTP], half_t* scratch, int lane, int nb0,
                                               const int (&oys)[NTP], const int (&oxs)[NTP], const ImgGeom& go) {
  constexpr int BN = 32 * NTN, PITCH = BN + 8, CPP = 4 * NTN;   // 16-byte chunks per pixel
  const Epi16& e = a.epi;
  const int r = lane & 31, h = lane >> 5;
  const int nstore = (a.N + 7) & ~7;  // the channel pitch is a multiple of 8: the pad channels are written too (zeros), consumers read them
  const float lab_a = e.has_lab ? e.lab_a : 1.f, lab_c = e.has_lab ? e.lab_c : 0.f;
  const bool has_lab = e.has_lab != 0;
  long long* ptab = reinterpret_cast<long long*>(scratch + 32 * PITCH);
  const bool has_bias = e.bias != nullptr;
#pragma unroll
  for (int j = 0; j < NTP; j++) {
    const int oy = oys[j], ox = oxs[j];
    const bool okp = oy >= 0 && oy < go.H && ox < go.W;
    const long long pixo = go.off + (long long)oy * go.W + ox;
    if (!WIDE && !okp) continue;
#pragma unroll
    for (int i = 0; i < NTN; i++)
#pragma unroll
      for (int g = 0; g < 4; g++) {
        const int nl = i * 32 + 8 * g + 4 * h, n = nb0 + nl;
        if (!WIDE && EDGE && n >= nstore) continue;
        f32x4 v;
#pragma unroll
        for (int t = 0; t < 4; t++) v[t] = acc[i][j][4 * g + t];
        if (has_bias && (!EDGE || n < a.Npad)) v += *reinterpret_cast<const f32x4*>(e.bias + n);
#pragma unroll
        for (int t = 0; t < 4; t++) v[t] = act_c<ACT>(v[t]);
        if (has_lab) {   // (uniform; LAB follows an activation in the PPLCNet blocks only)
#pragma unroll
          for (int t = 0; t < 4; t++) v[t] = fmaf(v[t], lab_a, lab_c);
        }
        if (!WIDE && e.residual) {
          h4 rs = *reinterpret_cast<const h4*>(e.residual + pixo * e.ld_res + n);
#pragma unroll
          for (int t = 0; t < 4; t++) v[t] += (float)rs[t];
        }
        h4 o;
#pragma unroll
        for (int t = 0; t < 4; t++) o[t] = (!EDGE || n + t < a.N) ? (half_t)v[t] : (half_t)0.f;
        if (WIDE) *reinterpret_cast<h4*>(scratch + r * PITCH + nl) = o;
        else *reinterpret_cast<h4*>(a.y + pixo * a.ldy + a.coff + n) = o;
      }
    if (!WIDE) continue;
    if (h == 0) ptab[r] = okp ? pixo : -1;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the wave's own LDS writes are complete (in order) and visible to its lanes
#pragma unroll
    for (int k = 0; k < 2 * NTN; k++) {
      const int idx = lane + 64 * k, pp = idx / CPP, ch = idx - pp * CPP;
      const long long po = ptab[pp];
      const int n = nb0 + ch * 8;
      if (po >= 0 && (!EDGE || n < nstore)) {
        const h8 v = *reinterpret_cast<const h8*>(scratch + pp * PITCH + ch * 8);
        *reinterpret_cast<h8*>(a.y + po * a.ldy + a.coff + n) = v;
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // reads returned before the next fragment overwrites the scratch
  }
}

template <int NTN, int NTP, int ACT>
__device__ __forceinline__ void store_tile16_a(const ConvArgs& a, const f32x16 (&acc)[NTN][NTP], half_t* scratch, int lane, int nb0,
                                               const int (&oys)[NTP], const int (&oxs)[NTP], const ImgGeom& go) {
  const bool wide = !a.epi.residual && !((a.ldy | a.coff) & 7);
  const bool edge = nb0 + 32 * NTN > a.N;
  if (wide) {
    if (edge) store_tile16_t<NTN, NTP, ACT, true, true>(a, acc, scratch, lane, nb0, oys, oxs, go);
    else store_tile16_t<NTN, NTP, ACT, false, true>(a, acc, scratch, lane, nb0, oys, oxs, go);
  } else {
    store_tile16_t<NTN, NTP, ACT, true, false>(a, acc, scratch, lane, nb0, oys, oxs, go);
  }
}

template <int NTN, int NTP>
__device__ __forceinline__ void store_tile16(const ConvArgs& a, const f32x16 (&acc)[NTN][NTP], half_t* scratch, int lane, int nb0,
                                             const int (&oys)[NTP], const int (&oxs)[NTP], const ImgGeom& go) {
  switch (a.epi.act) {
    case ACT_RELU: store_tile16_a<NTN, NTP, ACT_RELU>(a, acc, scratch, lane, nb0, oys, oxs, go); break;
    case ACT_HSWISH: store_tile16_a<NTN, NTP, ACT_HSWISH>(a, acc, scratch, lane, nb0, oys, oxs, go); break;
    case ACT_SWISH: store_tile16_a<NTN, NTP, ACT_SWISH>(a, acc, scratch, lane, nb0, oys, oxs, go); break;
    case ACT_SIGMOID: store_tile16_a<NTN, NTP, ACT_SIGMOID>(a, acc, scratch, lane, nb0, oys, oxs, go); break;
    default: store_tile16_a<NTN, NTP, ACT_NONE>(a, acc, scratch, lane, nb0, oys, oxs, go); break;
  }
}

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
  const Epi16& e = a.epi;
  if (DOT) {
#pragma unroll
    for (int j = 0; j < NTP; j++) {
      float sdot = 0.f;
#pragma unroll
      for (int i = 0; i < NTN; i++)
#pragma unroll
        for (int g = 0; g < 4; g++)
#pragma unroll
          for (int t = 0; t < 4; t++) {
            const int n = (wn * NTN + i) * 32 + 8 * g + 4 * h + t;
            float v = acc[i][j][4 * g + t] + (e.bias ? e.bias[n] : 0.f);
            v = act_f(v, e.act);
            sdot = fmaf(v, e.dot_w[n], sdot);  // dot_w is zero beyond N
          }
      sdot += __shfl_xor(sdot, 32);
      if (h == 0 && oys[j] >= 0) {
        const int oy = oys[j], ox = oxs[j];
        if (oy < go.H && ox < go.W) {
          const ImgGeom gm = e.gmap[blockIdx.y];
          float* m = e.dot_map + gm.off + (long long)(2 * oy + e.dot_py) * gm.W + 2 * ox + e.dot_px;
          *m = 0.5f * (*m + 1.f / (1.f + __expf(-(sdot + e.dot_b))));
        }
      }
    }
    return;
  }
  static_assert(WN == 1, "the epilogue assumes that a wave holds all BN channels of its pixels");
  __syncthreads();   // every wave is done with the last stage's LDS: it becomes the waves' transpose scratch
  store_tile16<NTN, NTP>(a, acc, reinterpret_cast<half_t*>(smem) + (size_t)wid * epi_scratch_halves<NTN * WN>(), lane, nb0, oys, oxs, go);
}

// ---------------------------------------------------------------------------------------------------------------------
// k_conv16v2: the same implicit GEMM for the 3x3 / 1x3 / 3x1 layers that dominate the server graphs, staged by LDS-DMA.
// The first form stages through registers and needs its VGPRs for the accumulators, so it can only prefetch one kernel row
// of weights and nothing of the halo: the in-kernel stamps show 2300 (row) to 5800 (row + halo) cycles of exposed L2 / HBM
// latency per 3500-cycle MFMA phase.  Here both operands go global -> LDS by global_load_lds (no VGPR staging):
//   * 8 waves on a 512-pixel tile x 32 * NTN channels (weights are re-read per 512 instead of 256 pixels);
//   * weights: a ring of 3 kernel-row buffers -- row r + 2 is requested while row r is multiplied (two MFMA phases of
//     latency cover); halo: two buffers, the next slab's tile is requested at the first row of the current slab;
//   * one raw s_barrier per row, counted s_waitcnt vmcnt (hipcc's __syncthreads would drain the DMA queue);
//   * LDS rows are un-padded 64-byte slabs (the DMA writes lane-linear), conflicts are avoided by an XOR swizzle of the
//     16-byte chunk index with bits 2-3 of the row, applied to the per-lane SOURCE address and again on the fragment reads.
// Same fragment maps, K order and epilogue as k_conv16: results are bit-identical.
// ---------------------------------------------------------------------------------------------------------------------
// The LDS-DMA request is written as inline asm: with the builtin (__builtin_amdgcn_global_load_lds) in a loop hipcc's
// wait-count pass treats the LDS counter as out of order and emits lgkmcnt(0) before every MFMA group -- which also waits
// for the fragment reads just issued for the NEXT k-step (checked on a reduced kernel: counted lgkmcnt(5/4/1) without the
// DMA or with this form, lgkmcnt(0) everywhere with the builtin).  M0 = wave-uniform LDS byte address; lane i writes
// 16 bytes at M0 + 16 i.  The kernel counts vmcnt for these requests by hand (nothing else loads inside the loop).
__device__ __forceinline__ void glds16(const void* g, void* l) {
  const unsigned la = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) char*)l);
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(g), "s"(la) : "memory", "m0");
}
#define RT_GLDS16(gp, lp) glds16((gp), (lp))

struct ConvArgs2 {
  ConvArgs a;
  const half_t* zeros;   // >= 16 zero bytes in device memory: DMA source of padding pixels / channels
  int hbuf_halves;       // size of one halo buffer (halves, multiple of 8)
  int hbufs;             // 2: next slab's halo prefetched; 1: single buffer (large halos)
};
constexpr int V2_HMAX = 6;   // DMA instructions per thread for one halo tile (6 * 512 * 16 B = 48 KB)

template <int NTN, int KW>
__global__ __launch_bounds__(512, 1) void k_conv16v2(const ConvArgs2 c2) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem2[];
  const ConvArgs& a = c2.a;
  const bool kstamp = RT_STAMP_ON(a.stamps && blockIdx.y == 0 && blockIdx.x == gridDim.x / 2 && threadIdx.x == 0);
  if (kstamp) a.stamps[4000] = __builtin_amdgcn_s_memtime();
  constexpr int NTHR = 512, NTP = 2, BN = 32 * NTN, ROW = KS;  // LDS row = 32 halves (64 bytes), un-padded
  const int TH = a.TH, TW = a.TW;
  const ImgGeom go = a.gout[blockIdx.y];
  const int tiles_x = (go.W + TW - 1) / TW, tiles_y = (go.H + TH - 1) / TH;
  const int zb = blockIdx.x % a.nzb, tile = blockIdx.x / a.nzb;
  if (tile >= tiles_x * tiles_y) return;
  const ImgGeom gi = a.gin[blockIdx.y];
  const int ty0 = (tile / tiles_x) * TH, tx0 = (tile % tiles_x) * TW;
  const int nb0 = zb * BN;
  const int HH = (TH - 1) * a.SH + a.KH, HW = (TW - 1) * a.SW + KW;
  half_t* hbuf = reinterpret_cast<half_t*>(smem2);
  half_t* wring = hbuf + (size_t)c2.hbufs * c2.hbuf_halves;
  constexpr int wbuf_halves = ((KW * BN * 4 + 511) & ~511) * 8;   // whole groups of 512 DMA slots
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int r = lane & 31, h = lane >> 5;

  int pix[NTP], oys[NTP], oxs[NTP];
#pragma unroll
  for (int j = 0; j < NTP; j++) {
    int q = (wid * NTP + j) * 32 + r;
    int ty = q / TW, tx = q - ty * TW;
    const bool ok = ty < TH;
    if (!ok) { ty = 0; tx = 0; }
    pix[j] = ty * a.SH * HW + tx * a.SW;   // halo pixel of tap (0, 0)
    oys[j] = ok ? ty0 + ty : -1;
    oxs[j] = tx0 + tx;
  }
  const int aswz = (r >> 2) & 3;            // rows of the weight tile: (row >> 2) & 3 == (r >> 2) & 3 (BN, 32 multiples of 16)

  f32x16 acc[NTN][NTP];
#pragma unroll
  for (int i = 0; i < NTN; i++)
#pragma unroll
    for (int j = 0; j < NTP; j++)
#pragma unroll
      for (int e = 0; e < 16; e++) acc[i][j][e] = 0.f;

  const int iy0 = ty0 * a.SH - a.PT, ix0 = tx0 * a.SW - a.PL;
  const int nslab = (a.Cin + KS - 1) / KS, nrows = nslab * a.KH;
  const int hchunks = HH * HW * 4; constexpr int wchunks = KW * BN * 4;
  const half_t* xtile = a.x + gi.off * a.ldx + ((long long)iy0 * gi.W + ix0) * a.ldx;
  // per-thread DMA sources, computed once: slot e = tid + 512 * i of a buffer holds (row e >> 2, physical chunk e & 3),
  // i.e. the logical chunk (e & 3) ^ ((row >> 2) & 3) of that row
  int hsrc[V2_HMAX];
#pragma unroll
  for (int i = 0; i < V2_HMAX; i++) {
    const int e = tid + i * NTHR;
    hsrc[i] = -1;
    if (e < hchunks) {
      const int p = e >> 2, cl = (e & 3) ^ ((p >> 2) & 3);
      const int hy = p / HW, hx = p - hy * HW;
      const int iy = iy0 + hy, ix = ix0 + hx;
      if (iy >= 0 && iy < gi.H && ix >= 0 && ix < gi.W) hsrc[i] = ((hy * gi.W + hx) * a.ldx + cl * 8) | (cl << 28);
    }
  }
  int wsrc[3];   // a kernel row: KW * BN * 4 <= 3 * 128 * 4 = 1536 chunks = 3 per thread
#pragma unroll
  for (int i = 0; i < 3; i++) {
    const int e = tid + i * NTHR;
    wsrc[i] = -1;
    if (e < wchunks) {
      const int row = e >> 2, cl = (e & 3) ^ ((row >> 2) & 3);
      const int dx = row / BN, n = row - dx * BN;
      if (nb0 + n < a.Npad) wsrc[i] = (dx * a.Npad + nb0 + n) * KS + cl * 8;
    }
  }
  const size_t row_halves = (size_t)KW * a.Npad * KS;
  const int wave_slot = wid * 64 * 8;   // halves: this wave's 64 consecutive 16-byte slots inside a group of 512
  auto dma_halo = [&](int s) {          // slab s -> halo buffer s % hbufs
    half_t* dst = hbuf + (size_t)(s % c2.hbufs) * c2.hbuf_halves;
    const int cvalid = min(KS, a.Cin - s * KS);
#pragma unroll
    for (int i = 0; i < V2_HMAX; i++) {
      if (i * NTHR >= hchunks) break;   // (uniform)
      const bool ok = hsrc[i] >= 0 && (hsrc[i] >> 28) * 8 < cvalid;
      const half_t* src = ok ? xtile + (hsrc[i] & 0x0fffffff) + s * KS : c2.zeros;
      RT_GLDS16(src, dst + (size_t)i * NTHR * 8 + wave_slot);
    }
  };
  auto dma_wrow = [&](int rr) {         // kernel row rr -> ring slot rr % 3
    half_t* dst = wring + (size_t)(rr % 3) * wbuf_halves;
    const half_t* wg = a.w + (size_t)rr * row_halves;
#pragma unroll
    for (int i = 0; i < 3; i++) {
      if (i * NTHR >= wchunks) break;   // (uniform)
      const half_t* src = wsrc[i] >= 0 ? wg + wsrc[i] : c2.zeros;
      RT_GLDS16(src, dst + (size_t)i * NTHR * 8 + wave_slot);
    }
  };
  // prologue: halo 0, rows 0 and 1 (and halo 1 with two buffers) in flight; wait for everything once
  dma_halo(0);
  dma_wrow(0);
  if (nrows > 1) dma_wrow(1);
  if (kstamp) a.stamps[4001] = __builtin_amdgcn_s_memtime();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  if (kstamp) a.stamps[4002] = __builtin_amdgcn_s_memtime();
  const int nw = (wchunks + NTHR - 1) / NTHR, nh = (hchunks + NTHR - 1) / NTHR;   // DMA instructions per row / per halo, per thread

  for (int rr = 0; rr < nrows; rr++) {
    const int s = rr / a.KH, dy = rr - s * a.KH;
    const int cvalid = min(KS, a.Cin - s * KS);
    const int ksteps = (cvalid + 15) >> 4;
    const bool stamp = RT_STAMP_ON(a.stamps && blockIdx.y == 0 && blockIdx.x == gridDim.x / 2 && tid == 0);
    if (stamp) { a.stamps[rr * 5 + 0] = __builtin_amdgcn_s_memtime(); a.stamps[rr * 5 + 1] = a.stamps[rr * 5 + 0]; }
    // ---- multiply kernel row rr: the k-steps (dx, 16 channels) of the row in one software-pipelined sequence -- the
    // fragments of step i + 1 are requested from LDS before the MFMAs of step i are issued, so only the first read of a
    // stage is exposed; the DMA requests for later stages go out behind the first MFMA group ----
    const half_t* wl = wring + (size_t)(rr % 3) * wbuf_halves;
    const half_t* halo = hbuf + (size_t)(s % c2.hbufs) * c2.hbuf_halves;
    // (a slab with fewer than 32 real channels still runs both 16-deep k-steps: its LDS rows and the packed weights are
    // zero-filled, and a fixed step count keeps the sequence below straight-line code -- with the steps behind run-time
    // tests hipcc waits lgkmcnt(0) before every MFMA group, which also waits for the prefetch just issued)
    (void)ksteps;
    constexpr int NK = KW * 2;
    const int tap0 = dy * HW;
    auto frags = [&](int it, h8 (&A)[NTN], h8 (&B)[NTP]) {
      const int dx = it >> 1, ks = it & 1;
      const int cl = ks * 2 + h;
      const half_t* wrow = wl + (size_t)(dx * BN + r) * ROW + ((cl ^ aswz) << 3);
#pragma unroll
      for (int i = 0; i < NTN; i++) A[i] = *reinterpret_cast<const h8*>(wrow + i * 32 * ROW);
#pragma unroll
      for (int j = 0; j < NTP; j++) {
        const int p = pix[j] + tap0 + dx;
        B[j] = *reinterpret_cast<const h8*>(halo + p * ROW + ((cl ^ ((p >> 2) & 3)) << 3));
      }
    };
    auto mfmas = [&](const h8 (&A)[NTN], const h8 (&B)[NTP]) {
#pragma unroll
      for (int i = 0; i < NTN; i++)
#pragma unroll
        for (int j = 0; j < NTP; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[i], B[j], acc[i][j], 0, 0, 0);
    };
    // (sched_barrier: hipcc otherwise sinks every fragment read down to its first use and waits lgkmcnt(0) there)
    h8 Af[2][NTN], Bf[2][NTP];
    frags(0, Af[0], Bf[0]);
    frags(1, Af[1], Bf[1]);
    __builtin_amdgcn_sched_barrier(0);
    mfmas(Af[0], Bf[0]);
    __builtin_amdgcn_sched_barrier(0);
    // ---- requests for later stages (the buffers they overwrite were last read before the barrier this wave just passed) ----
    if (rr + 2 < nrows) dma_wrow(rr + 2);
    bool halo_now = false;
    if (c2.hbufs == 2) { if (dy == 0 && s + 1 < nslab) { dma_halo(s + 1); halo_now = true; } }
    (void)halo_now;
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int it = 1; it < NK; it++) {
      if (it + 1 < NK) frags(it + 1, Af[(it + 1) & 1], Bf[(it + 1) & 1]);
      __builtin_amdgcn_sched_barrier(0);
      mfmas(Af[it & 1], Bf[it & 1]);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (stamp) { a.stamps[rr * 5 + 2] = __builtin_amdgcn_s_memtime(); a.stamps[rr * 5 + 3] = a.stamps[rr * 5 + 2]; }
    if (rr + 1 == nrows) break;
    // ---- the next row's data must have landed: everything except the requests made in THIS iteration ----
    const int ns = (rr + 1) / a.KH, ndy = (rr + 1) - ns * a.KH;
    if (c2.hbufs == 1 && ndy == 0) {
      // single halo buffer: every wave is done with the old tile only after the barrier; request and wait here (exposed)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      dma_halo(ns);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      continue;
    }
    // outstanding and allowed to stay in flight: row rr + 2 (nw instructions, if requested); the halo requested in this
    // iteration is only needed KH rows later, but it was issued AFTER row rr + 2, so it may stay in flight as well
    // (with one-row kernels the halo requested in this iteration is needed by the very next row: nothing may stay in flight)
    const int keep = (halo_now && ndy == 0) ? 0 : (rr + 2 < nrows ? nw : 0) + (halo_now ? nh : 0);
    // (a halo requested in an earlier iteration of this slab is older than row rr + 1's weights and therefore retired with them)
    switch (keep) {
      case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
      case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
      case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
      case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
      case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
      case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
      case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
      case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
      case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
      default: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
    }
    if (stamp) a.stamps[rr * 5 + 3] = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_s_barrier();
    if (stamp) a.stamps[rr * 5 + 4] = __builtin_amdgcn_s_memtime();
  }

  if (kstamp) a.stamps[4003] = __builtin_amdgcn_s_memtime();
  __builtin_amdgcn_s_barrier();   // every wave is done with the last stage's LDS (no DMA is in flight any more)
  if (kstamp) a.stamps[4005] = __builtin_amdgcn_s_memtime();
  store_tile16<NTN, NTP>(a, acc, reinterpret_cast<half_t*>(smem2) + (size_t)wid * epi_scratch_halves<NTN>(), lane, nb0, oys, oxs, go);
  if (kstamp) a.stamps[4006] = __builtin_amdgcn_s_memtime();
  if (kstamp) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); a.stamps[4004] = __builtin_amdgcn_s_memtime(); }
}

int g_conv16_v2 = 1;   // 1 = LDS-DMA kernel for the 3x3-class layers (default); 0 = register-staged k_conv16 everywhere (A/B)
static const half_t* zero_page16() {   // per device: DMA source of padding (one allocation per process and device)
  static const half_t* z[16] = {nullptr};
  int dev = 0;
  RT_HIP_CHECK(hipGetDevice(&dev));
  if (dev < 0 || dev >= 16) throw RtError(8, "conv16: device index out of range");
  if (!z[dev]) {
    void* p = nullptr;
    RT_HIP_CHECK(hipMalloc(&p, 256));
    RT_HIP_CHECK(hipMemset(p, 0, 256));
    z[dev] = (const half_t*)p;
  }
  return z[dev];
}

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
  static size_t lds_set = 0;  // per instantiation
  if (lds > lds_set) {
    RT_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(160 * 1024)));
    lds_set = 160 * 1024;
  }
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

const char* conv16_label(int KH, int KW, int N) {
  if (KH == 1 && KW == 1) return N <= 64 ? "gemm16/thin" : "gemm16";
  if (KH == 3 && KW == 3) return "conv16_3x3";
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
  a.lp = round_up(std::min(Cin, KS), 16) + 8;  // whole 16-deep k-steps of real (zero-filled) data + one pad chunk
  const bool dot = epi.dot_w != nullptr;
  // ---- LDS-DMA kernel for the 3x3-class layers ----
  if (g_conv16_v2 && !dot && KH <= 3 && (KW == 1 || KW == 3) && KH * KW > 1 && Cin >= 32 && n_img <= RT_MAX_GRID_Y) {
    int bn2 = 32, best = 1 << 30;
    for (int bn : {128, 96, 64, 32}) {
      const int nb = (Npad + bn - 1) / bn, cost = nb * bn + 16 * nb;
      if (cost < best) { best = cost; bn2 = bn; }
    }
    // 512-pixel tile: full-height tiles on short maps; the halo tile must fit V2_HMAX DMA instructions per thread
    int th, tw;
    if (maxHo >= 16) { const int ny = (maxHo + 15) / 16; th = maxHo >= 64 ? 16 : (maxHo + ny - 1) / ny; } else th = std::max(maxHo, 1);
    tw = std::max(1, std::min(512 / th, maxWo));
    auto hpix = [&](int t_h, int t_w) { return ((t_h - 1) * SH + KH) * ((t_w - 1) * SW + KW); };
    while (hpix(th, tw) * 4 > V2_HMAX * 512 && tw > 8) tw--;
    if (hpix(th, tw) * 4 <= V2_HMAX * 512) {
      ConvArgs2 c2;
      c2.a = a; c2.a.TH = th; c2.a.TW = tw; c2.a.lp = KS; c2.a.nzb = (Npad + bn2 - 1) / bn2;
      c2.zeros = zero_page16();
      c2.hbuf_halves = ((hpix(th, tw) * 4 + 511) & ~511) * 8;
      const size_t wbytes = (size_t)((KW * bn2 * 4 + 511) & ~511) * 16 * 3;
      c2.hbufs = (2 * (size_t)c2.hbuf_halves * 2 + wbytes <= 160 * 1024) ? 2 : 1;
      const size_t lds2 = std::max((size_t)c2.hbufs * c2.hbuf_halves * 2 + wbytes, (size_t)8 * (32 * (bn2 + 8) + 128) * 2);  // main loop | epilogue scratch
      const long long tiles2 = (long long)((maxWo + tw - 1) / tw) * ((maxHo + th - 1) / th);
      dim3 grid2((unsigned)(tiles2 * c2.a.nzb), (unsigned)n_img);
      static bool attr2 = false;
      if (!attr2) {
        for (const void* f : {(const void*)k_conv16v2<1, 1>, (const void*)k_conv16v2<2, 1>, (const void*)k_conv16v2<3, 1>, (const void*)k_conv16v2<4, 1>,
                              (const void*)k_conv16v2<1, 3>, (const void*)k_conv16v2<2, 3>, (const void*)k_conv16v2<3, 3>, (const void*)k_conv16v2<4, 3>})
          RT_HIP_CHECK(hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr2 = true;
      }
#define RT_V2_LAUNCH(NT) \
      switch (KW) { case 1: RT_LAUNCH((k_conv16v2<NT, 1>), grid2, dim3(512), lds2, st, c2); break; \
                    default: RT_LAUNCH((k_conv16v2<NT, 3>), grid2, dim3(512), lds2, st, c2); break; }
      switch (bn2 / 32) {
        case 1: RT_V2_LAUNCH(1); break;
        case 2: RT_V2_LAUNCH(2); break;
        case 3: RT_V2_LAUNCH(3); break;
        default: RT_V2_LAUNCH(4); break;
      }
#undef RT_V2_LAUNCH
      return;
    }
  }
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
      for (long long p = p0 + pl; p < pend; p += PL) {
        const h8 v = *reinterpret_cast<const h8*>(x + (g.off + p) * ldx + c8 * 8);
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
  static bool attr = false;
  if (!attr) { RT_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_se_fc16), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024)); attr = true; }
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

// thread = one pixel of f (half resolution): 4 outputs
__global__ __launch_bounds__(256) void k_deconv_to_map16(const half_t* __restrict__ f, int ldf, const ImgGeom* __restrict__ gf,
                                                         const ImgGeom* __restrict__ gmap, int C, const float* __restrict__ w, float b,
                                                         float* __restrict__ map) {
  const ImgGeom F = gf[blockIdx.y], M = gmap[blockIdx.y];
  const long long p = (long long)blockIdx.x * 256 + threadIdx.x;
  if (p >= (long long)F.H * F.W) return;
  const int y = (int)(p / F.W), x = (int)(p - (long long)y * F.W);
  float s[4] = {b, b, b, b};
  const half_t* src = f + (F.off + p) * ldf;
  for (int c = 0; c < C; c += 8) {
    const h8 v = *reinterpret_cast<const h8*>(src + c);
#pragma unroll
    for (int t = 0; t < 8; t++) {
      const f32x4 wv = *reinterpret_cast<const f32x4*>(w + (size_t)(c + t) * 4);
#pragma unroll
      for (int q = 0; q < 4; q++) s[q] = fmaf((float)v[t], wv[q], s[q]);
    }
  }
#pragma unroll
  for (int q = 0; q < 4; q++)
    map[M.off + (long long)(2 * y + (q >> 1)) * M.W + 2 * x + (q & 1)] = 1.f / (1.f + __expf(-s[q]));
}
void deconv_to_map16(hipStream_t st, const half_t* f, int ldf, const ImgGeom* gf, const ImgGeom* gmap, int n_img, long long max_pix,
                     int C, const float* w, float b, float* map) {
  if (n_img <= 0) return;
  if (C % 8) throw RtError(8, "deconv_to_map16: C must be a multiple of 8");
  RT_LAUNCH(k_deconv_to_map16, dim3((unsigned)((max_pix + 255) / 256), n_img), dim3(256), 0, st, f, ldf, gf, gmap, C, w, b, map);
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
