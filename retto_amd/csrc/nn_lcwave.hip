// Fused thin LCNetV3 blocks, one wave per tile, no barrier after the weights are staged (gfx950).
//
//   y = epi_pw( W_pw . lab(act(dw3x3(x) + b_dw)) )        C_in = 16 G, N = 16 NT
//
// k_lc_thin (nn_kernels.hip) stages an input patch and the depthwise result through LDS with two barriers per tile; at
// 2 workgroups per CU the phases (loads, depthwise VALU, MFMA, stores) of a tile run one after the other and the layers sit
// at 0.3 of HBM.  Two forms that drop the barriers, both bit-identical to k_lc_thin and to k_dwconv_rows + k_gemm (arithmetic
// order per output: bias, then taps (dy, dx) ascending with fmaf; k ascending in the MFMA chain):
//
// k_lc_lds (production, strides (1,1), (2,2), (2,1)): a wave stages the 16-channel slice of its input patch through its own LDS
// buffer -- see the comment at the kernel.
//
// k_lc_wave (stride 1, kept as the A/B form RT_LC_WAVE=1): no LDS for activations at all.
//   * the MFMA pixel operand of v_mfma_f32_16x16x4_f32 wants lane (r = lane & 15, q = lane >> 4) to hold channels
//     16 g + 4 q .. + 3 of pixel r for the 16-deep k group g (nn_kernels.hip, mma_chunk).  That is one f32x4 of an NHWC
//     pixel: the lane loads it straight from global memory, for the MT + 2 input rows of the tile;
//   * the horizontal taps are the neighbouring lanes' registers: DPP row_shr:1 / row_shl:1 move them inside the 16-lane row
//     (= the same q), and the two lanes at the row ends keep the `old` operand, which a sparse load (lanes r = 0 and 15
//     only, by exec mask) has filled with the halo pixel;
//   * the depthwise result of group g is the MFMA operand as it stands; the weights' fragments come from LDS (staged once
//     per workgroup), the 9 taps of the lane's channel group too (4 distinct addresses per wave: broadcast reads);
//   * the loads of group g + 1 are issued before the MFMAs of group g.
//   It lost to k_lc_lds on every shape: 16 different pixels in the 16 lanes of a row make the texture addresser walk 64
//   separate 16-byte pieces per load / store (TA_BUSY 0.72).
//
// What the measurements behind these kernels showed (tools/scratch/*.hip):
//   * the VALU does not run under an executing fp32 MFMA, not even from another wave of the SIMD (64 MFMAs + 128 v_fma take
//     the sum of both): every VALU op of the epilogues and of the depthwise half is paid in MFMA time -- hence packed
//     v_pk_fma / v_pk_mul / v_pk_add, v_med3 for the clamp, buffer addressing with scalar offsets, no predicates;
//   * __builtin_bit_cast on an ELEMENT of an ext_vector miscompiles with this hipcc (all four lanes became element 0);
//   * buffer_store_dwordx4 with an SGPR soffset: hipcc's hazard recognizer assumes no store-data hazard and lets the next VALU
//     op overwrite the data registers; on gfx950 dword 1 of lanes 12-15 of each 16-lane row was stored corrupted.  The stores
//     here keep soffset = 0.
#include "common.h"
#include "nn_dev.h"

namespace rt {
namespace nn {

namespace {
constexpr int LROW = KC + 4;
constexpr int DPP_ROW_SHL1 = 0x101, DPP_ROW_SHR1 = 0x111;

template <int CTRL>
__device__ __forceinline__ f32x4 dpp_row(const f32x4& old, const f32x4& src) {  // lanes without a source in their row keep `old`
  f32x4 o;
#pragma unroll
  for (int e = 0; e < 4; e++) {   // (__builtin_bit_cast on a vector ELEMENT miscompiles with this hipcc: every lane of the result became element 0)
    const float fo = old[e], fs = src[e];
    o[e] = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(fo), __float_as_int(fs), CTRL, 0xf, 0xf, false));
  }
  return o;
}
}  // namespace

typedef float f32x2 __attribute__((ext_vector_type(2)));

// hardswish + LearnableAffineBlock on 4 values, same expression and rounding as epi_val<ACT_HSWISH, 1>: packed add / mul / fma
// (2 values per VALU op) and one v_med3 per value for the clamp
__device__ __forceinline__ f32x4 hswish_lab4(const f32x4& v, float lab_a, float lab_c) {
  f32x4 t = v + 3.0f;
#pragma unroll
  for (int e = 0; e < 4; e++) { const float te = t[e]; t[e] = __builtin_amdgcn_fmed3f(te, 0.0f, 6.0f); }
  f32x4 o = v * t;
  o = o * 0.16666667f;
  const f32x4 a4 = {lab_a, lab_a, lab_a, lab_a}, c4 = {lab_c, lab_c, lab_c, lab_c};
  return __builtin_elementwise_fma(o, a4, c4);
}

struct LcwArgs {   // (compact: the kernel keeps its scalars in SGPRs; a spilled SGPR costs VALU writelane / readlane ops)
  const float* x; const ImgGeom* gin; const ImgGeom* gout;
  const float* Wd; const float* bd; const float* Wp; const float* bias; float* y;
  float dw_a, dw_c, pw_a, pw_c;
  int Npad, ldy, tiles_per_wave;
};

// DWA: the depthwise half ends in hardswish + LAB (stride 1) or in nothing (stride 2).  The pointwise half always ends in
// hardswish + LAB (a block without LAB passes a = 1, c = 0).  C_in = 16 G and N = 16 NT exactly: no channel masks.
template <int G, int NT, int MT, bool DWA>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) void k_lc_wave(LcwArgs p) {
  constexpr int CP = G * 16, NKC = (CP + KC - 1) / KC, NCOL = NT * 16;
  constexpr int IR = MT + 2;   // input rows of a tile (stride 1)
  constexpr int NV = 2;        // registers per row: {centre, halo}
  __shared__ __attribute__((aligned(16))) float wt[NKC * NCOL * LROW];
  __shared__ __attribute__((aligned(16))) float tp[10 * CP + NCOL];  // 9 taps + depthwise bias, [t][CP]; pointwise bias [NCOL]
  const ImgGeom g = p.gout[blockIdx.y], gi = p.gin[blockIdx.y];
  const int tiles_x = (g.W + 15) / 16, tiles_y = (g.H + MT - 1) / MT, n_tiles = tiles_x * tiles_y;
  const int first = blockIdx.x * 4 * p.tiles_per_wave;
  if (first >= n_tiles) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), r = lane & 15, q = lane >> 4;
  for (int idx = tid; idx < NKC * NCOL * 8; idx += 256) {
    const int c4i = idx & 7, row = (idx >> 3) % NCOL, kc = (idx >> 3) / NCOL;
    *reinterpret_cast<f32x4*>(wt + (kc * NCOL + row) * LROW + c4i * 4) =
        *reinterpret_cast<const f32x4*>(p.Wp + ((long long)kc * p.Npad + row) * KC + c4i * 4);
  }
  for (int idx = tid; idx < 10 * CP / 4; idx += 256)
    *reinterpret_cast<f32x4*>(tp + idx * 4) =
        idx < 9 * CP / 4 ? *reinterpret_cast<const f32x4*>(p.Wd + idx * 4) : *reinterpret_cast<const f32x4*>(p.bd + (idx - 9 * CP / 4) * 4);
  for (int idx = tid; idx < NCOL / 4; idx += 256) *reinterpret_cast<f32x4*>(tp + 10 * CP + idx * 4) = *reinterpret_cast<const f32x4*>(p.bias + idx * 4);
  __syncthreads();   // the only barrier: from here on the waves run on their own

  // Buffer descriptors of the image: an offset (VGPR part) past num_records reads zeros / drops the store, so the padding
  // columns, the tile overhang and "this lane is not a row end" cost no branch and no select.  The row and the channel group
  // go into the scalar offset (not range checked: rows outside the image are a scalar branch).
  const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(p.x + gi.off * CP), 0, (unsigned)gi.H * (unsigned)gi.W * (CP * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(
      p.y + g.off * p.ldy, 0, (unsigned)g.H * (unsigned)g.W * (unsigned)p.ldy * 4, 0x00020000);
  const unsigned row_bytes = (unsigned)gi.W * CP * 4, yrow_bytes = (unsigned)g.W * (unsigned)p.ldy * 4;
  constexpr unsigned OOB = 0x80000000u;

  int tile = first + wave;
  if (tile >= n_tiles) return;
  int ty = tile / tiles_x, tx = tile - ty * tiles_x;   // (uniform; advanced by 4 tiles per iteration without dividing)
  ty = __builtin_amdgcn_readfirstlane(ty); tx = __builtin_amdgcn_readfirstlane(tx);

  f32x4 in[IR][NV];
  unsigned c_off, h_off, y_off;
  auto lane_offsets = [&](int txx) __attribute__((always_inline)) {
    const int ox = txx * 16 + r, ix = ox;
    const int hx = r == 0 ? ox - 1 : ox + 1;
    const bool h_ok = (r == 0 || r == 15) && (unsigned)hx < (unsigned)gi.W;
    c_off = ix < gi.W ? (unsigned)ix * (CP * 4) + q * 16 : OOB;
    h_off = h_ok ? (unsigned)hx * (CP * 4) + q * 16 : OOB;
    y_off = ox < g.W ? (unsigned)ox * ((unsigned)p.ldy * 4) + q * 16 : OOB;
  };
  // requests channel group GG of the tile in tile row tyy (lane offsets of its tile column are current)
  const bool edge_lane = r == 0 || r == 15;
  auto fetch = [&](int tyy, auto gg_tag) __attribute__((always_inline)) {
    constexpr int GG = decltype(gg_tag)::value;
#pragma unroll
    for (int i = 0; i < IR; i++) {
      const int iy = tyy * MT - 1 + i;
      if ((unsigned)iy < (unsigned)gi.H) {   // (uniform)
        const unsigned so = (unsigned)iy * row_bytes + GG * 64;
        in[i][0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs, c_off, so, 0));
      } else {
#pragma unroll
        for (int v = 0; v < NV - 1; v++) in[i][v] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
    // halo pixels: only the row-end lanes take part (the texture addresser spends its cycles per active lane; the other
    // lanes' halo registers are never looked at: DPP overwrites them)
    if (edge_lane) {
#pragma unroll
      for (int i = 0; i < IR; i++) {
        const int iy = tyy * MT - 1 + i;
        if ((unsigned)iy < (unsigned)gi.H) in[i][NV - 1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs, h_off, (unsigned)iy * row_bytes + GG * 64, 0));
        else in[i][NV - 1] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
  };

  lane_offsets(tx);
  fetch(ty, IntTag<0>{});
  for (int it = 0; it < p.tiles_per_wave; it++) {
    // the tile after this one (4 tiles on: the other three waves of the workgroup take the ones between)
    int ntx = tx + 4, nty = ty;
    while (ntx >= tiles_x) { ntx -= tiles_x; nty++; }
    const bool more = it + 1 < p.tiles_per_wave && nty < tiles_y;
    const unsigned y_off_t = y_off;
    f32x4 acc[MT][NT];
#pragma unroll
    for (int gg = 0; gg < G; gg++) {
      // ---- depthwise 3x3 of group gg: the MFMA pixel operand ----
      f32x4 a[MT];
      {
        const float* tq = tp + (4 * gg + q) * 4;
        f32x4 dsum[MT];
        const f32x4 dwb = *reinterpret_cast<const f32x4*>(tq + 9 * CP);
#pragma unroll
        for (int mt = 0; mt < MT; mt++) dsum[mt] = dwb;
#pragma unroll
        for (int i = 0; i < IR; i++) {
          f32x4 col[3];
          col[0] = dpp_row<DPP_ROW_SHR1>(in[i][1], in[i][0]);
          col[1] = in[i][0];
          col[2] = dpp_row<DPP_ROW_SHL1>(in[i][1], in[i][0]);
          // input row i is tap row dy = i - mt of output row mt; rows arrive in ascending i, i.e. ascending dy for
          // every mt: per output the order stays (dy, dx) ascending
#pragma unroll
          for (int mt = 0; mt < MT; mt++) {
            const int dy = i - mt;
            if (dy < 0 || dy > 2) continue;
#pragma unroll
            for (int dx = 0; dx < 3; dx++) {
              const f32x4 w = *reinterpret_cast<const f32x4*>(tq + (dy * 3 + dx) * CP);
              dsum[mt] = __builtin_elementwise_fma(col[dx], w, dsum[mt]);
            }
          }
        }
#pragma unroll
        for (int mt = 0; mt < MT; mt++) a[mt] = DWA ? hswish_lab4(dsum[mt], p.dw_a, p.dw_c) : dsum[mt];
      }
      __builtin_amdgcn_sched_barrier(0);
      // ---- next group's (or the next tile's first group's) input: in flight under the MFMAs ----
      if (gg + 1 < G) {
        if (gg == 0) fetch(ty, IntTag<1 % G>{}); else if (gg == 1) fetch(ty, IntTag<2 % G>{}); else fetch(ty, IntTag<3 % G>{});
      } else if (more) {
        lane_offsets(ntx);
        fetch(nty, IntTag<0>{});
      }
      __builtin_amdgcn_sched_barrier(0);
      // ---- pointwise: 4 MFMA steps per column tile and row ----
      const float* wg = wt + ((gg >> 1) * NCOL + r) * LROW + (gg & 1) * 16 + 4 * q;
#pragma unroll
      for (int nt = 0; nt < NT; nt++) {
        const f32x4 b = *reinterpret_cast<const f32x4*>(wg + nt * 16 * LROW);
#pragma unroll
        for (int s = 0; s < 4; s++)
#pragma unroll
          for (int mt = 0; mt < MT; mt++) {
            if (gg == 0 && s == 0) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[s], a[mt][s], f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
            else acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[s], a[mt][s], acc[mt][nt], 0, 0, 0);
          }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    // ---- epilogue: lane holds output channels 16 nt + 4 q .. + 3 of pixel (ty * MT + mt, ox) ----
#pragma unroll
    for (int mt = 0; mt < MT; mt++) {
      const int oy = ty * MT + mt;
      if (oy < g.H) {   // (uniform)
        // (the row goes into the VGPR offset, soffset stays 0: with a register in soffset hipcc's hazard recognizer assumes the
        //  store has no data hazard and lets the next VALU op overwrite the data registers straight away -- on gfx950 that
        //  corrupted dword 1 of lanes 12-15 of each row of 16; with soffset = 0 it inserts the wait state itself)
        const unsigned y_row = y_off_t == OOB ? OOB : y_off_t + (unsigned)oy * yrow_bytes;
#pragma unroll
        for (int nt = 0; nt < NT; nt++) {
          const f32x4 bias = *reinterpret_cast<const f32x4*>(tp + 10 * CP + nt * 16 + q * 4);
          const f32x4 o = hswish_lab4(acc[mt][nt] + bias, p.pw_a, p.pw_c);
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned, o), yrs, y_row + nt * 64, 0, 0);
        }
      }
    }
    if (!more) break;
    tx = ntx; ty = nty;
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// k_lc_lds: the same block with the input staged through a WAVE-PRIVATE LDS buffer.
// k_lc_wave's loads and stores put 16 different pixels into the 16 lanes of a row (the MFMA operand layout): the texture
// addresser then walks 64 separate 16-byte pieces per instruction (~60 cycles against ~18 for a contiguous KB, measured with
// tools/scratch/ta_pattern.hip) and is the busiest unit of that kernel (TA_BUSY 0.72).  Here a wave
//   * loads the 16-channel slice g of its (MT + 2) x 18 pixel input patch with lane l -> 16-byte slot l (4 lanes = the 64 bytes
//     of one pixel's slice: contiguous per quad), 7 instructions for MT = 4, zero padding by out-of-range offsets;
//   * writes the slots to its own LDS buffer (XOR swizzle of the chunk with bits 1-2 of the pixel: the reads below are
//     conflict-free), and reads them back in the MFMA layout -- lane (r, q) reads chunk q of pixels r, r + 1, r + 2 of each row:
//     the three horizontal taps are three ds_read_b128, no DPP, no halo registers;
//   * has the loads of slice g + 1 in flight (in the staging registers just written out) under the depthwise and MFMA work of g.
// No barrier: LDS operations of a wave execute in order, the buffer belongs to the wave.  MT = 4 rows per tile: the taps and the
// weight fragments (LDS reads, too) are amortised over four MFMA row tiles.  Same arithmetic order as k_lc_wave / k_lc_thin.
// ---------------------------------------------------------------------------------------------------------------------
template <int G, int NT, int MT, int SH, int SW, bool DWA, int WPE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void k_lc_lds(LcwArgs p) {
  constexpr int CP = G * 16, NKC = (CP + KC - 1) / KC, NCOL = NT * 16;
  constexpr int IR = (MT - 1) * SH + 3, PXR = 15 * SW + 3, RP = PXR * 64;     // input rows of a tile, pixels per row, row pitch in bytes
  constexpr int SLOTS = IR * PXR * 4, NJ = (SLOTS + 63) / 64, XB = IR * RP + 512;   // (+ 512 bytes: the slots past the patch land there)
  __shared__ __attribute__((aligned(16))) float wt[NKC * NCOL * LROW];
  __shared__ __attribute__((aligned(16))) float tp[10 * CP + NCOL];  // 9 taps + depthwise bias, [t][CP]; pointwise bias [NCOL]
  __shared__ __attribute__((aligned(16))) char xb[4 * XB];
  const ImgGeom g = p.gout[blockIdx.y], gi = p.gin[blockIdx.y];
  const int tiles_x = (g.W + 15) / 16, tiles_y = (g.H + MT - 1) / MT, n_tiles = tiles_x * tiles_y;
  const int first = blockIdx.x * 4 * p.tiles_per_wave;
  if (first >= n_tiles) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), r = lane & 15, q = lane >> 4;
  const int n0 = blockIdx.z * NCOL;   // this workgroup's block of output channels (N = gridDim.z * NCOL)
  for (int idx = tid; idx < NKC * NCOL * 8; idx += 256) {
    const int c4i = idx & 7, row = (idx >> 3) % NCOL, kc = (idx >> 3) / NCOL;
    *reinterpret_cast<f32x4*>(wt + (kc * NCOL + row) * LROW + c4i * 4) =
        *reinterpret_cast<const f32x4*>(p.Wp + ((long long)kc * p.Npad + n0 + row) * KC + c4i * 4);
  }
  for (int idx = tid; idx < 10 * CP / 4; idx += 256)
    *reinterpret_cast<f32x4*>(tp + idx * 4) =
        idx < 9 * CP / 4 ? *reinterpret_cast<const f32x4*>(p.Wd + idx * 4) : *reinterpret_cast<const f32x4*>(p.bd + (idx - 9 * CP / 4) * 4);
  for (int idx = tid; idx < NCOL / 4; idx += 256) *reinterpret_cast<f32x4*>(tp + 10 * CP + idx * 4) = *reinterpret_cast<const f32x4*>(p.bias + n0 + idx * 4);
  __syncthreads();   // the only barrier

  const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(p.x + gi.off * CP), 0, (unsigned)gi.H * (unsigned)gi.W * (CP * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(
      p.y + g.off * p.ldy, 0, (unsigned)g.H * (unsigned)g.W * (unsigned)p.ldy * 4, 0x00020000);
  const unsigned row_bytes = (unsigned)gi.W * CP * 4, yrow_bytes = (unsigned)g.W * (unsigned)p.ldy * 4;
  constexpr unsigned OOB = 0x80000000u;
  char* const xw = xb + wave * XB;
  auto swz = [](int px, int c) { return (px * 4 + (c ^ ((px >> 1) & 3))) * 16; };
  // slot of this lane in load instruction j: row, pixel of the row, chunk -> LDS byte offset (fixed) and image offset (per tile)
  unsigned wr[NJ];   // LDS byte address | row << 24 | pixel << 16 (the image offsets of a tile are rebuilt from these)
#pragma unroll
  for (int j = 0; j < NJ; j++) {
    const int t = lane + 64 * j, row = t / (PXR * 4), rem = t - row * (PXR * 4), px = rem >> 2, c = rem & 3;
    const unsigned la = (unsigned)(size_t)(__attribute__((address_space(3))) char*)xw + (t < SLOTS ? row * RP + swz(px, c) : IR * RP + (lane & 31) * 16);
    wr[j] = la | (unsigned)(t < SLOTS ? row : 31) << 25 | (unsigned)(px * 4 + c) << 17;   // (LDS addresses stay below 2^17; pixel * 4 + chunk < 256)
  }
  const char* rd[3];
#pragma unroll
  for (int dx = 0; dx < 3; dx++) rd[dx] = xw + swz(r * SW + dx, q);

  int tile = first + wave;
  if (tile >= n_tiles) return;
  int ty = tile / tiles_x, tx = tile - ty * tiles_x;
  ty = __builtin_amdgcn_readfirstlane(ty); tx = __builtin_amdgcn_readfirstlane(tx);

  // Prefetch distance: one slice (the loads of slice n + 1 fly under the depthwise + MFMA work of slice n, ~1.3 us) or, where
  // the registers allow it (MT = 4 runs 2 waves per SIMD) and G is even, two slices with two sets of staging registers.
  constexpr int DEPTH = (MT == 4 && G % 2 == 0 && SH == 1 && SW == 1) ? 2 : 1;
  unsigned goff[NJ], y_off;
  f32x4 st[DEPTH][NJ];
  auto lane_offsets = [&](int tyy, int txx) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < NJ; j++) {
      const int row = wr[j] >> 25, pc = (wr[j] >> 17) & 255;
      const int iy = tyy * (MT * SH) - 1 + row, ix = txx * (16 * SW) - 1 + (pc >> 2);
      const bool ok = row < IR && (unsigned)iy < (unsigned)gi.H && (unsigned)ix < (unsigned)gi.W;
      goff[j] = ok ? (unsigned)iy * row_bytes + (unsigned)ix * (CP * 4) + (pc & 3) * 16 : OOB;
    }
    const int ox = txx * 16 + r;
    y_off = ox < g.W ? (unsigned)ox * ((unsigned)p.ldy * 4) + n0 * 4 + q * 16 : OOB;
  };
  auto fetch = [&](auto gg_tag, auto buf_tag) __attribute__((always_inline)) {   // slice GG of the tile `goff` describes
    constexpr int GG = decltype(gg_tag)::value, B = decltype(buf_tag)::value;
#pragma unroll
    for (int j = 0; j < NJ; j++) st[B][j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs, goff[j], GG * 64, 0));
  };

  lane_offsets(ty, tx);
  fetch(IntTag<0>{}, IntTag<0>{});
  if (DEPTH == 2) fetch(IntTag<1 % G>{}, IntTag<DEPTH - 1>{});
  for (int it = 0; it < p.tiles_per_wave; it++) {
    int ntx = tx + 4, nty = ty;
    while (ntx >= tiles_x) { ntx -= tiles_x; nty++; }
    const bool more = it + 1 < p.tiles_per_wave && nty < tiles_y;
    const unsigned y_off_t = y_off;
    f32x4 acc[MT][NT];
#pragma unroll
    for (int gg = 0; gg < G; gg++) {
      // ---- slice gg: staging registers -> the wave's LDS buffer (the reads of slice gg - 1 are ahead of these writes in the
      //      wave's LDS queue), then a later slice's loads into the same registers ----
      const int b = DEPTH == 2 ? (gg & 1) : 0;
#pragma unroll
      for (int j = 0; j < NJ; j++) *reinterpret_cast<__attribute__((address_space(3))) f32x4*>(wr[j] & 0x1ffff) = st[b][j];
      __builtin_amdgcn_sched_barrier(0);
      {
        const int nx = gg + DEPTH;   // the slice to request: of this tile, or of the next one
        if (nx < G) {
          switch (nx) {   // (gg is a constant after unrolling: one case survives)
            case 1: fetch(IntTag<1 % G>{}, IntTag<(1 % G) % DEPTH>{}); break;
            case 2: fetch(IntTag<2 % G>{}, IntTag<(2 % G) % DEPTH>{}); break;
            case 3: fetch(IntTag<3 % G>{}, IntTag<(3 % G) % DEPTH>{}); break;
            case 4: fetch(IntTag<4 % G>{}, IntTag<(4 % G) % DEPTH>{}); break;
            case 5: fetch(IntTag<5 % G>{}, IntTag<(5 % G) % DEPTH>{}); break;
            case 6: fetch(IntTag<6 % G>{}, IntTag<(6 % G) % DEPTH>{}); break;
            default: fetch(IntTag<7 % G>{}, IntTag<(7 % G) % DEPTH>{}); break;
          }
        } else if (more) {
          if (nx == G) lane_offsets(nty, ntx);
          if (nx == G) fetch(IntTag<0>{}, IntTag<0>{}); else fetch(IntTag<1 % G>{}, IntTag<DEPTH - 1>{});
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      // ---- depthwise 3x3 of the slice: the MFMA pixel operand ----
      f32x4 a[MT];
      {
        const float* tq = tp + (4 * gg + q) * 4;
        f32x4 dsum[MT];
        const f32x4 dwb = *reinterpret_cast<const f32x4*>(tq + 9 * CP);
#pragma unroll
        for (int mt = 0; mt < MT; mt++) dsum[mt] = dwb;
#pragma unroll
        for (int i = 0; i < IR; i++) {
          f32x4 col[3];
#pragma unroll
          for (int dx = 0; dx < 3; dx++) col[dx] = *reinterpret_cast<const f32x4*>(rd[dx] + i * RP);
#pragma unroll
          for (int mt = 0; mt < MT; mt++) {
            const int dy = i - mt * SH;
            if (dy < 0 || dy > 2) continue;
#pragma unroll
            for (int dx = 0; dx < 3; dx++) {
              const f32x4 w = *reinterpret_cast<const f32x4*>(tq + (dy * 3 + dx) * CP);
              dsum[mt] = __builtin_elementwise_fma(col[dx], w, dsum[mt]);
            }
          }
        }
#pragma unroll
        for (int mt = 0; mt < MT; mt++) a[mt] = DWA ? hswish_lab4(dsum[mt], p.dw_a, p.dw_c) : dsum[mt];
      }
      __builtin_amdgcn_sched_barrier(0);
      // ---- pointwise ----
      const float* wg = wt + ((gg >> 1) * NCOL + r) * LROW + (gg & 1) * 16 + 4 * q;
#pragma unroll
      for (int nt = 0; nt < NT; nt++) {
        const f32x4 b = *reinterpret_cast<const f32x4*>(wg + nt * 16 * LROW);
#pragma unroll
        for (int s = 0; s < 4; s++)
#pragma unroll
          for (int mt = 0; mt < MT; mt++) {
            if (gg == 0 && s == 0) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[s], a[mt][s], f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
            else acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[s], a[mt][s], acc[mt][nt], 0, 0, 0);
          }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    // ---- epilogue ----
#pragma unroll
    for (int mt = 0; mt < MT; mt++) {
      const int oy = ty * MT + mt;
      if (oy < g.H) {   // (uniform)
        // (the row goes into the VGPR offset, soffset stays 0: with a register in soffset hipcc's hazard recognizer assumes the
        //  store has no data hazard and lets the next VALU op overwrite the data registers straight away -- on gfx950 that
        //  corrupted dword 1 of lanes 12-15 of each row of 16; with soffset = 0 it inserts the wait state itself)
        const unsigned y_row = y_off_t == OOB ? OOB : y_off_t + (unsigned)oy * yrow_bytes;
#pragma unroll
        for (int nt = 0; nt < NT; nt++) {
          const f32x4 bias = *reinterpret_cast<const f32x4*>(tp + 10 * CP + nt * 16 + q * 4);
          const f32x4 o = hswish_lab4(acc[mt][nt] + bias, p.pw_a, p.pw_c);
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned, o), yrs, y_row + nt * 64, 0, 0);
        }
      }
    }
    if (!more) break;
    tx = ntx; ty = nty;
  }
}

// instantiated (stride, C_in / 16, N / 16): the thin blocks of the two LCNetV3 backbones
static int lc_wave_code(int sh, int sw, int Cp, int Npad16) {
  const int gq = Cp / 16, nt = Npad16 / 16;
  if (Cp % 16 || Npad16 % 16) return 0;
  if (sh == 1 && sw == 1) {
    if (gq == 1 && nt == 2) return 1;   // 16 -> 32
    if (gq == 2 && nt == 4) return 2;   // 32 -> 64
    if (gq == 3 && nt == 3) return 3;   // 48 -> 48
    if (gq == 4 && nt == 4) return 4;   // 64 -> 64
    // 128 -> 128 in two blocks of 64 output channels: 1.35 vs 1.56 ms on uniform images, but 1.69 vs 1.52 ms on the ragged C3
    // batch (every block recomputes the depthwise half): opt-in (RT_LC_WAVE=5)
    if (gq == 8 && nt == 8 && g_lc_wave == 5) return 9;
  } else if (sh == 2 && sw == 2) {
    if (gq == 2 && nt == 3) return 6;   // 32 -> 48
    if (gq == 3 && nt == 6) return 7;   // 48 -> 96
  } else if (sh == 2 && sw == 1) {
    if (gq == 4 && nt == 8) return 8;   // 64 -> 128 (rec s4.0; k_lc_lds only)
  }
  return 0;
}
bool lc_wave_supported(int K, int sh, int sw, int Cp, int C, int N, int Npad16, int dw_act, int dw_has_lab, const Epilogue& epi) {
  if (K != 3 || Cp != C || N != Npad16 || lc_wave_code(sh, sw, Cp, Npad16) == 0) return false;
  if (epi.residual || epi.a_scale || epi.am_max || !epi.bias || epi.act != ACT_HSWISH) return false;
  // depthwise tail: hardswish + LAB (LearnableRepLayer at stride 1) or nothing (stride 2)
  // depthwise tail (LearnableRepLayer: the activation is skipped when stride == 2; the rec net's (2, 1) is not 2)
  return (sh == 2 && sw == 2) ? (dw_act == ACT_NONE && !dw_has_lab) : (dw_act == ACT_HSWISH && dw_has_lab);
}

void lc_wave(hipStream_t st, int sh, int sw, const float* x, const ImgGeom* gin, const ImgGeom* gout, int n_img, int maxHo,
             int maxWo, int Cp, int C, const float* Wd, const float* bd, int dw_act, int dw_has_lab, float dw_a, float dw_c,
             const float* Wp, int N, int Npad16, float* y, int ldy, const Epilogue& epi) {
  if (n_img <= 0) return;
  if (!lc_wave_supported(3, sh, sw, Cp, C, N, Npad16, dw_act, dw_has_lab, epi)) throw RtError(8, "lc_wave: unsupported block (check lc_wave_supported)");
  static const int tpw_env = getenv("RT_LCW_TPW") ? atoi(getenv("RT_LCW_TPW")) : 0;
  static const int mt_env = getenv("RT_LCW_MT") ? atoi(getenv("RT_LCW_MT")) : 0;
  const int code = lc_wave_code(sh, sw, Cp, Npad16);
  const int tpw = tpw_env > 0 ? tpw_env : 4;
  if (g_lc_wave >= 3 || code >= 6) {   // (the direct-load form below is kept for the stride-1 blocks only: A/B, RT_LC_WAVE=1)
    // LDS-staged form: 4-row tiles at stride 1 (2 waves per SIMD), 2-row tiles at stride 2
    const int mtl = (sh == 1 && mt_env != 2) ? 4 : 2;
    const int tiles = ((maxWo + 15) / 16) * ((maxHo + mtl - 1) / mtl);
    dim3 grid((tiles + 4 * tpw - 1) / (4 * tpw), n_img);
    LcwArgs a{x, gin, gout, Wd, bd, Wp, epi.bias, y, dw_a, dw_c, epi.has_lab ? epi.lab_a : 1.f, epi.has_lab ? epi.lab_c : 0.f, Npad16, ldy, tpw};
#define RT_LCL(GG, NN) do { if (mtl == 2) RT_LAUNCH((k_lc_lds<GG, NN, 2, 1, 1, true, 3>), grid, dim3(256), 0, st, a); else RT_LAUNCH((k_lc_lds<GG, NN, 4, 1, 1, true, 2>), grid, dim3(256), 0, st, a); } while (0)
    switch (code) {
      case 1: RT_LCL(1, 2); break;
      case 2: RT_LCL(2, 4); break;
      case 3: RT_LCL(3, 3); break;
      case 4: RT_LCL(4, 4); break;
      case 6: RT_LAUNCH((k_lc_lds<2, 3, 2, 2, 2, false, 2>), grid, dim3(256), 0, st, a); break;
      case 7: RT_LAUNCH((k_lc_lds<3, 6, 2, 2, 2, false, 2>), grid, dim3(256), 0, st, a); break;
      case 9: grid.z = 2; RT_LAUNCH((k_lc_lds<8, 4, 4, 1, 1, true, 2>), grid, dim3(256), 0, st, a); break;
      default: RT_LAUNCH((k_lc_lds<4, 8, 2, 2, 1, true, 2>), grid, dim3(256), 0, st, a); break;
    }
#undef RT_LCL
    return;
  }
  const int mt = mt_env == 1 ? 1 : 2;
  const int tiles = ((maxWo + 15) / 16) * ((maxHo + mt - 1) / mt);
  dim3 grid((tiles + 4 * tpw - 1) / (4 * tpw), n_img);
  LcwArgs a{x, gin, gout, Wd, bd, Wp, epi.bias, y, dw_a, dw_c, epi.has_lab ? epi.lab_a : 1.f, epi.has_lab ? epi.lab_c : 0.f, Npad16, ldy, tpw};
#define RT_LCW_T(GG, NN, MM, SS) RT_LAUNCH((k_lc_wave<GG, NN, MM, true>), grid, dim3(256), 0, st, a)
#define RT_LCW(GG, NN, SS) do { if (mt == 1) RT_LCW_T(GG, NN, 1, SS); else RT_LCW_T(GG, NN, 2, SS); } while (0)
  switch (code) {
    case 1: RT_LCW(1, 2, 1); break;
    case 2: RT_LCW(2, 4, 1); break;
    case 3: RT_LCW(3, 3, 1); break;
    case 4: RT_LCW(4, 4, 1); break;
    default: throw RtError(8, "lc_wave: unsupported shape (check lc_wave_supported)");
  }
#undef RT_LCW
#undef RT_LCW_T
}

}  // namespace nn
}  // namespace rt
