// Device-side pieces shared by the fp16 convolution kernels (nn_f16.hip: register-staged k_conv16; nn_f16_dma.hip: the
// LDS-DMA kernels k_conv16v2 and k_gemm16): vector types, activations, the kernel argument block and the epilogue.
// Included by .hip files only.
#pragma once
#include "nn_f16.h"

#include <type_traits>

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
  int xcd;           // 1: logical workgroup ids are remapped so that neighbours (the channel blocks of one pixel tile, adjacent tiles) run on ONE XCD and share its L2
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
// Workgroups are dispatched round-robin over the 8 XCDs (hardware id mod 8), each with its own L2.  This maps hardware id ->
// logical id so that XCD x works on one contiguous range of logical ids (bijective for any count).
__device__ __forceinline__ unsigned xcd_remap(unsigned bid, unsigned nwg) {
  const unsigned q = nwg >> 3, r = nwg & 7, xcd = bid & 7, local = bid >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + local;
}

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
  // this lane's bias values, loaded once up front (all 4 * NTN loads in flight together; inside the element loops each load
  // is waited for where it is used and the epilogue of a 512 x 128 tile doubles).  5-fragment tiles (k_conv16 only) have
  // no registers left for them.
  constexpr bool PRE = NTN <= 4;
  f32x4 bv[PRE ? NTN : 1][4];
  if (PRE) {
#pragma unroll
    for (int i = 0; i < NTN; i++)
#pragma unroll
      for (int g = 0; g < 4; g++) {
        const int n = nb0 + i * 32 + 8 * g + 4 * h;
        bv[PRE ? i : 0][g] = (has_bias && n < a.Npad) ? *reinterpret_cast<const f32x4*>(e.bias + n) : f32x4{0.f, 0.f, 0.f, 0.f};
      }
  }
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
        if (PRE) v += bv[PRE ? i : 0][g];
        else if (has_bias && (!EDGE || n < a.Npad)) v += *reinterpret_cast<const f32x4*>(e.bias + n);
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

// DOT epilogue (PFHeadLocal): the 64 -> 1 conv over the activated channels of a pixel, sigmoid, and 0.5 * (map + .) into
// the fp32 probability map at phase (dot_py, dot_px) of the 2x up-sampled grid.  The block holds ALL output channels.
template <int NTN, int NTP>
__device__ __forceinline__ void dot_tile16(const ConvArgs& a, const f32x16 (&acc)[NTN][NTP], int lane, const int (&oys)[NTP],
                                           const int (&oxs)[NTP], const ImgGeom& go, int img) {
  const Epi16& e = a.epi;
  const int h = lane >> 5;
  // (activation resolved once per workgroup, as in store_tile16: a run-time switch per element is a scalar branch each)
  auto dot_epi = [&](auto actc) {
    constexpr int ACT = decltype(actc)::value;
#pragma unroll
    for (int j = 0; j < NTP; j++) {
      float sdot = 0.f;
#pragma unroll
      for (int i = 0; i < NTN; i++)
#pragma unroll
        for (int g = 0; g < 4; g++) {
          const int n = i * 32 + 8 * g + 4 * h;
          f32x4 v;
#pragma unroll
          for (int t = 0; t < 4; t++) v[t] = acc[i][j][4 * g + t];
          if (e.bias) v += *reinterpret_cast<const f32x4*>(e.bias + n);
          const f32x4 dw = *reinterpret_cast<const f32x4*>(e.dot_w + n);  // dot_w is zero beyond N
#pragma unroll
          for (int t = 0; t < 4; t++) sdot = fmaf(act_c<ACT>(v[t]), dw[t], sdot);
        }
      sdot += __shfl_xor(sdot, 32);
      if (h == 0 && oys[j] >= 0) {
        const int oy = oys[j], ox = oxs[j];
        if (oy < go.H && ox < go.W) {
          const ImgGeom gm = e.gmap[img];
          float* m = e.dot_map + gm.off + (long long)(2 * oy + e.dot_py) * gm.W + 2 * ox + e.dot_px;
          *m = 0.5f * (*m + 1.f / (1.f + __expf(-(sdot + e.dot_b))));
        }
      }
    }
  };
  switch (e.act) {
    case ACT_RELU: dot_epi(std::integral_constant<int, ACT_RELU>{}); break;
    case ACT_HSWISH: dot_epi(std::integral_constant<int, ACT_HSWISH>{}); break;
    case ACT_SWISH: dot_epi(std::integral_constant<int, ACT_SWISH>{}); break;
    case ACT_SIGMOID: dot_epi(std::integral_constant<int, ACT_SIGMOID>{}); break;
    default: dot_epi(std::integral_constant<int, ACT_NONE>{}); break;
  }
}

// host side, nn_f16_dma.hip: the LDS-DMA paths.  Return false when the layer is not theirs (the caller then takes k_conv16).
bool conv16_dma(hipStream_t st, const ConvArgs& a, int n_img, int maxHo, int maxWo);

}  // namespace nh
}  // namespace rt
