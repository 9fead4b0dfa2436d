// RSEFPN / DB-head 3x3 convs of the det network, restructured around the nearest-neighbour upsampling (round 4).
//
// The two big det layers (`fpn.inp0` and `head.conv1`, 3x3 96 -> 24 at 1/4 resolution, 46 % of the
// network's MACs) convolve tensors that are mostly UPSAMPLED coarser levels:
//     in2  = lateral(c2) * s + up2(in3)                            (12 -> 96 channels by a bias-free 1x1 conv, + top-down add)
//     fuse = concat(up8(p5), up4(p4), up2(p3), p2) * scales        (4 x 24 channels)
// A 3x3 conv over up_s(z) reads, for an output pixel at phase (y mod s, x mod s), at most 2 x 2 distinct pixels of z, and
// which ones -- and with which sums of the nine taps -- depends on the phase only.  So
//   * conv3x3(up2(z)) is, per phase, a 2 x 2 conv of z with pre-summed weights: 4 instead of 9 taps (k_fpn_phase, coarse part);
//   * conv3x3(up4(z)) / conv3x3(up8(z)) take one of 9 row / column classes (first, interior, last row of a block): a
//     [coarse pixel][9][24] class tensor is computed at 1/16 (1/32) resolution (k_fpn_class) and gathered per output pixel;
//   * conv3x3(lateral(c2) * s) is a 3x3 conv of the 12-channel tap tensor itself with per-image composed weights
//     W'[tap][c][n] = sum_m Wlat[c][m] s[m] Wconv[n][m][tap] (k_fpn_compose): the 96-channel tensor in2 is never built.
// MFMA work per output pixel: head 864 -> 312 deep, inp0 864 -> 492 deep (inp1: 864 -> 546); the 0.7 GB lateral tensor of the
// finest level and its write + read disappear.  Same mathematics as the reference graph; the pre-summed weights change the
// rounding order only (test_det_net: <= 1e-4 on the probability map, as before).
//
// Replaces (together with nets.cpp) ONNX Runtime's Session::run for the det graph,
// /root/reference/retto-core/src/worker/ort_worker.rs:189-198.
#include "nn.h"
#include "nn_dev.h"

#include <cstdlib>

namespace rt {
namespace nn {

namespace {
constexpr int ROW = 28;                        // floats per LDS row: 24 channels + 4 pad = 7 x 16 bytes (odd: consecutive rows
                                               // never share a 16-byte bank group within 8 lanes)
constexpr int FT = 18 * 18, CT = 10 * 10;      // pixels of the fine halo tile (4 phase planes of 9 x 9) / the coarse halo tile
constexpr int CW = 16 * 24;                    // weight rows of a coarse slab: 4 phases x 4 taps x 24 outputs (fine: 9 taps x 24 outputs)
// Tile images in LDS: a tile row (9 pixels of a phase plane, 10 of the coarse tile) has a pitch of 72 sixteen-byte slots = 288
// floats, a pixel 7 slots.  ds_read_b128 serves the lanes in fixed groups of 16 ({0-3, 12-15, 20-27}, ...: MI355X_MICROARCH.md,
// LDS): with lane = Y * 8 + X a group holds half-rows of four consecutive Y; their slots (8 Y + 7 X) mod 16 are all different
// exactly when the row pitch is 8 mod 16 slots.  (The first version packed the rows at 63 / 70 slots: PMC showed 45 % of the
// LDS cycles of these kernels as bank conflicts.)
constexpr int TPITCH = 288, FPLANE = 9 * TPITCH;
constexpr int FT_FLOATS = 4 * FPLANE, CT_FLOATS = 10 * TPITCH;
// (round 5: the weights no longer pass through LDS -- a lane reads its channel's row of the tap straight from L1 / L2 into the A
//  operand registers, one tap ahead of the MFMAs that use it: 65.7 -> 41.5 KB of LDS, three workgroups per CU instead of two, and
//  the weight stash -- nine 16-byte LDS writes per thread and stage -- is gone)
constexpr int LDS_FLOATS = FT_FLOATS > CT_FLOATS ? FT_FLOATS : CT_FLOATS;
constexpr int NPF = 8, NPS = 8;                // prefetch registers (16-byte vectors) per thread: pixel operands / scale vectors
}  // namespace

// One workgroup = a 16 x 16 tile of the fine level = 8 x 8 coarse pixels.  Wave w owns phase (w >> 1, w & 1): its 64 lanes are
// the 8 x 8 pixels of that phase (lane = coarse Y * 8 + X -> fine pixel (2 Y + py, 2 X + px)), so that all lanes of a wave share
// the phase's weights -- what v_mfma_f32_4x4x1_16B_f32 with the A-operand broadcast needs (lane = pixel, D[i] = channel 4 g + i,
// as k_conv3_few).  The fine halo tile is stored as four phase planes so that the lanes of a wave read consecutive LDS rows for
// every tap (one wave-uniform offset per tap).  K is walked in stages: the fine tensor (CF4 x 4 channels, nine taps), then NS
// slabs of 24 coarse channels (four taps each); the operands of the next stage are requested into registers before the MFMAs of
// the current one.
template <int CF4, int NS, int HASG>
__global__ __launch_bounds__(256, 3) void k_fpn_phase(FpnPhaseArgs a, const ImgGeom* __restrict__ gf,
                                                      const ImgGeom* __restrict__ gc) {
  extern __shared__ __attribute__((aligned(16))) float lds[];   // LDS_FLOATS (41.5 KB: three workgroups per CU)
  const int img = blockIdx.y;
  const ImgGeom g = gf[img], gcs = gc[img];
  const int tiles_x = (g.W + 15) >> 4, tiles_y = (g.H + 15) >> 4;
  if ((int)blockIdx.x >= tiles_x * tiles_y) return;
  const int ty = blockIdx.x / tiles_x, tx = blockIdx.x - ty * tiles_x;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int py = wave >> 1, px = wave & 1;
  const int Y = lane >> 3, X = lane & 7;
  const int oy = ty * 16 + 2 * Y + py, ox = tx * 16 + 2 * X + px;
  const bool valid = oy < g.H && ox < g.W;

  f32x4 acc[6];
  if (HASG) {
    // start values: the class tensor of the two coarsest levels (bias included), [9 classes][quarter-resolution pixel][24]
    const ImgGeom gq = a.gg[img];
    const int qy = min(oy >> 2, gq.H - 1), qx = min(ox >> 2, gq.W - 1);
    const int ry = oy & 3, rx = ox & 3;
    const int cls = (ry == 0 ? 0 : ry == 3 ? 2 : 1) * 3 + (rx == 0 ? 0 : rx == 3 ? 2 : 1);
    const float* gp = a.G + (cls * a.g_plane + gq.off + (long long)qy * gq.W + qx) * 24;
#pragma unroll
    for (int i = 0; i < 6; i++) acc[i] = *reinterpret_cast<const f32x4*>(gp + i * 4);
  } else {
#pragma unroll
    for (int i = 0; i < 6; i++) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  }

  // Operand staging, global -> registers -> LDS.  Every load of a stage is issued UNCONDITIONALLY (out-of-range pixels read a safe
  // address and are zeroed when they are written to LDS; the per-image scale vectors are loaded beside the pixels and multiplied
  // in at stash time): with the loads inside `if (in range)` blocks hipcc waits vmcnt(0) behind every one of them -- a dozen
  // serialised round trips per tile, 13 us of the first version's 28 us per block.
  f32x4 pf[NPF], sf[NPS];
  unsigned vmask = 0;
  constexpr int FXL = (FT * CF4 + 255) / 256;
  constexpr int CXL = (CT * 6 + 255) / 256;
  static_assert(FXL <= NPF && CXL <= NPF && FXL <= NPS && CXL <= NPS, "prefetch registers");

  auto fetch_fine = [&]() {
    vmask = 0;
#pragma unroll
    for (int i = 0; i < FXL; i++) {
      const int idx = min(tid + 256 * i, FT * CF4 - 1);
      const int hp = idx / CF4, c4 = idx - hp * CF4;
      const int hy = hp / 18, hx = hp - hy * 18;
      const int gy = ty * 16 + hy - 1, gx = tx * 16 + hx - 1;
      const bool ok = gy >= 0 && gy < g.H && gx >= 0 && gx < g.W;
      const long long pix = ok ? g.off + (long long)gy * g.W + gx : g.off;
      pf[i] = *reinterpret_cast<const f32x4*>(a.fine + pix * a.ld_fine + c4 * 4);
      vmask |= (ok ? 1u : 0u) << i;
    }
    if (a.fine_scale) {
#pragma unroll
      for (int i = 0; i < FXL; i++) {
        const int idx = min(tid + 256 * i, FT * CF4 - 1);
        sf[i] = *reinterpret_cast<const f32x4*>(a.fine_scale + (long long)img * a.ld_fs + (idx % CF4) * 4);
      }
    }
  };
  auto stash_fine = [&]() {
#pragma unroll
    for (int i = 0; i < FXL; i++) {
      const int idx = tid + 256 * i;
      if (idx < FT * CF4) {
        const int hp = idx / CF4, c4 = idx - hp * CF4;
        const int hy = hp / 18, hx = hp - hy * 18;
        const int e = ((hy & 1) * 2 + (hx & 1)) * FPLANE + (hy >> 1) * TPITCH + (hx >> 1) * ROW;
        f32x4 v = pf[i];
        if (a.fine_scale) v *= sf[i];
        if (!((vmask >> i) & 1)) v = f32x4{0.f, 0.f, 0.f, 0.f};
        *reinterpret_cast<f32x4*>(lds + e + c4 * 4) = v;
      }
    }
  };
  // Coarse slabs (round 5): everything that does not depend on the slab is computed ONCE -- per prefetch register the byte offset
  // of its 16 bytes in the coarse tensor (0x80000000 = outside the image: the buffer load returns zeros, no select at stash time),
  // its LDS address and its place in the scale vector; a slab advances by 96 bytes, a scalar operand.  The per-slab form (two
  // divisions, bounds, a 64-bit address and four selects per register, for fetch and again for stash) was ~180 of the ~290 VALU
  // instructions a wave spent per slab beside its 576 MFMAs -- and an fp32 MFMA loop pays each of them in MFMA time.
  unsigned c_src[CXL], c_dst[CXL], c_sc[CXL];
#pragma unroll
  for (int i = 0; i < CXL; i++) {
    const int idx = min(tid + 256 * i, CT * 6 - 1);
    const int cp = idx / 6, c4 = idx - cp * 6;
    const int cy = cp / 10, cx = cp - cy * 10;
    const int gy = ty * 8 + cy - 1, gx = tx * 8 + cx - 1;
    const bool ok = gy >= 0 && gy < gcs.H && gx >= 0 && gx < gcs.W && tid + 256 * i < CT * 6;
    c_src[i] = ok ? (unsigned)((gy * gcs.W + gx) * a.ld_coarse + c4 * 4) * 4u : 0x80000000u;
    c_dst[i] = (unsigned)(cy * TPITCH + cx * ROW + c4 * 4);
    c_sc[i] = (unsigned)(c4 * 4);
  }
  // (descriptor over this image's part of the coarse tensor; an image is far below 2 GB)
  const __amdgpu_buffer_rsrc_t crs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.coarse + gcs.off * a.ld_coarse), 0, 0x7fffffffu, 0x00020000);
  auto fetch_coarse = [&](int s) {
#pragma unroll
    for (int i = 0; i < CXL; i++) pf[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(crs, c_src[i], s * 96, 0));
    if (a.coarse_scale) {
      const float* sp = a.coarse_scale + (long long)img * a.ld_cs + s * 24;
#pragma unroll
      for (int i = 0; i < CXL; i++) sf[i] = *reinterpret_cast<const f32x4*>(sp + c_sc[i]);
    }
  };
  auto stash_coarse = [&]() {
#pragma unroll
    for (int i = 0; i < CXL; i++) {
      if (tid + 256 * i < CT * 6) {
        f32x4 v = pf[i];
        if (a.coarse_scale) v *= sf[i];
        *reinterpret_cast<f32x4*>(lds + c_dst[i]) = v;
      }
    }
  };
  // A operand: lane 4 g + i holds output channel 4 g + i (lanes >= 24 are never selected by abid; they read rows that their
  // ds_read_b128 lane group does not touch otherwise: no bank conflict from the idle lanes either)
  const int wl = (lane & 31) < 24 ? (lane & 31) : (lane & 15);
#define RT_FPN_MFMA6(av, bv)                                                             \
  acc[0] = __builtin_amdgcn_mfma_f32_4x4x1f32(av, bv, acc[0], 4, 0, 0);                  \
  acc[1] = __builtin_amdgcn_mfma_f32_4x4x1f32(av, bv, acc[1], 4, 1, 0);                  \
  acc[2] = __builtin_amdgcn_mfma_f32_4x4x1f32(av, bv, acc[2], 4, 2, 0);                  \
  acc[3] = __builtin_amdgcn_mfma_f32_4x4x1f32(av, bv, acc[3], 4, 3, 0);                  \
  acc[4] = __builtin_amdgcn_mfma_f32_4x4x1f32(av, bv, acc[4], 4, 4, 0);                  \
  acc[5] = __builtin_amdgcn_mfma_f32_4x4x1f32(av, bv, acc[5], 4, 5, 0);

  // weight rows of this lane: fine [tap][n][CF4 * 4] (per-image composed weights), coarse [slab][phase * 96 + tap * 24 + n][24]
  const float* wfine = a.Wf + (long long)img * a.wf_img + wl * (CF4 * 4);
  const float* wcoarse = a.Wc + (wave * 96 + wl) * 24;
  f32x4 wq[2][6];   // the tap being multiplied and the next one
  auto ldw_fine = [&](int tap, f32x4 (&d)[6]) __attribute__((always_inline)) {
#pragma unroll
    for (int kk = 0; kk < CF4; kk++) d[kk] = *reinterpret_cast<const f32x4*>(wfine + (tap * 24 * CF4 + kk) * 4);
  };
  auto ldw_coarse = [&](int s, int t, f32x4 (&d)[6]) __attribute__((always_inline)) {
    const float* wr = wcoarse + ((long long)s * CW + t * 24) * 24;
#pragma unroll
    for (int kk = 0; kk < 6; kk++) d[kk] = *reinterpret_cast<const f32x4*>(wr + kk * 4);
  };
  fetch_fine();
  ldw_fine(0, wq[0]);
  stash_fine();
  __syncthreads();
  fetch_coarse(0);
  __builtin_amdgcn_sched_barrier(0);
  {
    const float* xb = lds + Y * TPITCH + X * ROW;
#pragma unroll
    for (int dy = 0; dy < 3; dy++)
#pragma unroll
      for (int dx = 0; dx < 3; dx++) {
        const int tap = dy * 3 + dx;
        if (tap + 1 < 9) ldw_fine(tap + 1, wq[(tap + 1) & 1]);
        else ldw_coarse(0, 0, wq[(tap + 1) & 1]);   // (tap 8 multiplies wq[0]: the first coarse tap goes to wq[1])
        __builtin_amdgcn_sched_barrier(0);
        const int sy = py + dy, sx = px + dx;   // wave-uniform: halo-tile coordinates of this tap = (2 Y + sy, 2 X + sx)
        const int toff = ((sy & 1) * 2 + (sx & 1)) * FPLANE + (sy >> 1) * TPITCH + (sx >> 1) * ROW;
        const float* xr = xb + toff;
#pragma unroll
        for (int kk = 0; kk < CF4; kk++) {
          const f32x4 b = *reinterpret_cast<const f32x4*>(xr + kk * 4);
          const f32x4 w = wq[tap & 1][kk];
#pragma unroll
          for (int s2 = 0; s2 < 4; s2++) { RT_FPN_MFMA6(w[s2], b[s2]) }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
  }
  __syncthreads();
  // coarse taps run through the same two registers sets: global tap index g = 4 s + t uses wq[(g + 1) & 1]
#pragma unroll 1
  for (int s = 0; s < NS; s++) {
    stash_coarse();
    __syncthreads();
    if (s + 1 < NS) fetch_coarse(s + 1);
    __builtin_amdgcn_sched_barrier(0);
    const float* cb = lds + Y * TPITCH + X * ROW;
#pragma unroll
    for (int t = 0; t < 4; t++) {
      // the next tap's weights (the next slab's first tap behind the last one; past the end: a repeat of this slab's, unused)
      if (t + 1 < 4) ldw_coarse(s, t + 1, wq[t & 1]);
      else ldw_coarse(min(s + 1, NS - 1), 0, wq[t & 1]);
      __builtin_amdgcn_sched_barrier(0);
      const int toff = (py + (t >> 1)) * TPITCH + (px + (t & 1)) * ROW;   // coarse pixel (Y + py + ty - 1, X + px + tx - 1), tile origin at -1
      const float* xr = cb + toff;
#pragma unroll
      for (int kk = 0; kk < 6; kk++) {
        const f32x4 b = *reinterpret_cast<const f32x4*>(xr + kk * 4);
        const f32x4 w = wq[(t + 1) & 1][kk];
#pragma unroll
        for (int s2 = 0; s2 < 4; s2++) { RT_FPN_MFMA6(w[s2], b[s2]) }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();
  }
#undef RT_FPN_MFMA6
  // epilogue: bias (unless it came with the class tensor), activation, 96 contiguous bytes per pixel
  f32x4 o[6];
#pragma unroll
  for (int gi = 0; gi < 6; gi++) {
    f32x4 v = acc[gi];
    if (!HASG && a.bias) v += *reinterpret_cast<const f32x4*>(a.bias + gi * 4);
    if (a.act == ACT_RELU) {
#pragma unroll
      for (int j = 0; j < 4; j++) v[j] = fmaxf(v[j], 0.f);
    }
    o[gi] = valid ? v : f32x4{0.f, 0.f, 0.f, 0.f};
  }
  if (valid) {
    float* yr = a.y + (g.off + (long long)oy * g.W + ox) * a.ldy;
#pragma unroll
    for (int gi = 0; gi < 6; gi++) *reinterpret_cast<f32x4*>(yr + gi * 4) = o[gi];
  }
  if (a.pool) {
    // channel sums of this tile's outputs for the squeeze-excite that follows (se_fc_from_tiles): values through LDS, then a
    // fixed summation order (thread (c, part) adds 64 pixels in index order, four parts in order): repeatable and independent
    // of the batch the page is in
    float* ps = lds;   // [256 pixels][24] (the stage buffers are free: the last MFMA stage ended with a barrier)
#pragma unroll
    for (int gi = 0; gi < 6; gi++) *reinterpret_cast<f32x4*>(ps + tid * 24 + gi * 4) = o[gi];
    __syncthreads();
    float part = 0.f;
    if (tid < 96) {
      const int c = tid % 24, q = tid / 24;
      for (int p = 0; p < 64; p++) part += ps[(q * 64 + p) * 24 + c];
    }
    __syncthreads();
    if (tid < 96) ps[tid] = part;
    __syncthreads();
    if (tid < 24) a.pool[((long long)img * a.pool_tiles + blockIdx.x) * 24 + tid] = (ps[tid] + ps[24 + tid]) + (ps[48 + tid] + ps[72 + tid]);
  }
}

int g_fpn_phase_off = 0;
bool fpn_phase_supported(int cf, int cc) {
  static const bool off = getenv("RT_FPN_PHASE") && atoi(getenv("RT_FPN_PHASE")) == 0;
  if (off) return false;
  const int cf4 = (cf + 3) / 4;
  return (cf4 == 3 || cf4 == 5 || cf4 == 6) && (cc == 24 || cc == 96);
}

void fpn_phase(hipStream_t st, const FpnPhaseArgs& a, int cf, int cc, const ImgGeom* gf, const ImgGeom* gc, int n_img, int maxH,
               int maxW) {
  if (n_img <= 0) return;
  const int cf4 = (cf + 3) / 4, ns = cc / 24;
  dim3 grid(((maxW + 15) / 16) * ((maxH + 15) / 16), n_img);
  constexpr int LB = LDS_FLOATS * 4;
  for (const void* f : {(const void*)k_fpn_phase<6, 1, 1>, (const void*)k_fpn_phase<3, 4, 0>, (const void*)k_fpn_phase<5, 4, 0>, (const void*)k_fpn_phase<6, 1, 0>})
    allow_big_lds(f, LB);
  if (a.G) {
    if (cf4 == 6 && ns == 1) { RT_LAUNCH((k_fpn_phase<6, 1, 1>), grid, dim3(256), LB, st, a, gf, gc); return; }
  } else {
    if (cf4 == 3 && ns == 4) { RT_LAUNCH((k_fpn_phase<3, 4, 0>), grid, dim3(256), LB, st, a, gf, gc); return; }
    if (cf4 == 5 && ns == 4) { RT_LAUNCH((k_fpn_phase<5, 4, 0>), grid, dim3(256), LB, st, a, gf, gc); return; }
    if (cf4 == 6 && ns == 1) { RT_LAUNCH((k_fpn_phase<6, 1, 0>), grid, dim3(256), LB, st, a, gf, gc); return; }
  }
  throw RtError(8, "fpn_phase: no instance for this channel split");
}

// ---------------------------------------------------------------------------
// Class tensor of a level that reaches the head conv upsampled by S = 4 or 8: V[class (row class x 3 + column class)][pixel][24]
//   = sum over the <= 2 x 2 pixels of z the class touches of Wcls[class][tap] . (z * scale)          (+ bias, + the class tensor
//   of the next coarser level at the class this pixel's position inside ITS block implies).
// Row class 0 = first row of an S-block (taps: row above with W[-1], own row with W[0] + W[+1]), 1 = interior (own row, all
// three summed), 2 = last row (own row with W[-1] + W[0], row below with W[+1]); columns alike.  Wcls [9 classes][9 taps][24 n][24 k]
// holds the pre-summed weights (taps a class does not touch are skipped).  Same MFMA form as k_fpn_phase (lane = pixel, A operand
// = the class's weights broadcast to all sixteen 4 x 4 blocks), operands straight from global memory into registers: a wave = 64
// consecutive pixels of an image and one class (blockIdx.z).  1/16 and 1/32 resolution: ~2 GMAC per 32 pages.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_fpn_class(const float* __restrict__ z, int ldz, const float* __restrict__ scale, int ld_s,
                                                   const ImgGeom* __restrict__ geom, const float* __restrict__ Wcls,
                                                   const float* __restrict__ bias, const float* __restrict__ lower,
                                                   const ImgGeom* __restrict__ glow, long long low_plane, float* __restrict__ V,
                                                   long long plane) {
  // a workgroup = 256 consecutive pixels of an image x ONE class: its <= 2 x 2 taps are four slots (ty, tx); the slots' weights
  // go through LDS, every pixel operand is requested up front (one round trip per workgroup: with the loads next to their uses
  // hipcc waited vmcnt(0) behind each 16-byte load, ~60 serialised round trips per wave)
  __shared__ __attribute__((aligned(16))) float wlds[4 * 24 * ROW];
  const int img = blockIdx.y, cls = blockIdx.z;
  const int rc = cls / 3, cc = cls - rc * 3;
  const ImgGeom g = geom[img];
  const long long npix = (long long)g.H * g.W;
  if ((long long)blockIdx.x * 256 >= npix) return;
  const int tid = threadIdx.x;
  const long long p = (long long)blockIdx.x * 256 + tid;
  const bool valid = p < npix;
  const long long pc = valid ? p : npix - 1;
  const int y = (int)(pc / g.W), x = (int)(pc - (long long)y * g.W);
  const int lane = tid & 63, wl = (lane & 31) < 24 ? (lane & 31) : (lane & 15);   // (idle lanes on rows their ds_read_b128 group does not use: no bank conflict)
  const int ry0 = rc == 0 ? -1 : 0, ry1 = rc == 2 ? 1 : 0, rx0 = cc == 0 ? -1 : 0, rx1 = cc == 2 ? 1 : 0;
  // slot (ty, tx) -> relative pixel (ty ? ry1 : ry0, tx ? rx1 : rx0); a slot that repeats the previous one is skipped
  const bool two_y = ry1 != ry0, two_x = rx1 != rx0;
  f32x4 b[4][6], wreg[3];
  // (weights first: loads return in order, and the weights are needed first -- for the LDS stash)
#pragma unroll
  for (int i = 0; i < 3; i++) {   // 4 slots x 24 rows x 6 chunks = 576 chunks
    const int idx = min(tid + 256 * i, 575);
    const int t = idx / 144, r = idx - t * 144, n = r / 6, c4 = r - n * 6;
    const int ry = (t >> 1) ? ry1 : ry0, rx = (t & 1) ? rx1 : rx0;
    wreg[i] = *reinterpret_cast<const f32x4*>(Wcls + (((long long)(cls * 9 + (ry + 1) * 3 + (rx + 1))) * 24 + n) * 24 + c4 * 4);
  }
  unsigned okmask = 0;
#pragma unroll
  for (int t = 0; t < 4; t++) {
    const int ry = (t >> 1) ? ry1 : ry0, rx = (t & 1) ? rx1 : rx0;
    const int yy = y + ry, xx = x + rx;
    const bool ok = yy >= 0 && yy < g.H && xx >= 0 && xx < g.W;
    const float* src = z + (g.off + (ok ? (long long)yy * g.W + xx : 0)) * ldz;
#pragma unroll
    for (int i = 0; i < 6; i++) b[t][i] = *reinterpret_cast<const f32x4*>(src + i * 4);
    okmask |= (ok ? 1u : 0u) << t;
  }
  f32x4 sc[6], acc[6];
#pragma unroll
  for (int i = 0; i < 6; i++) {
    sc[i] = scale ? *reinterpret_cast<const f32x4*>(scale + (long long)img * ld_s + i * 4) : f32x4{1.f, 1.f, 1.f, 1.f};
    acc[i] = bias ? *reinterpret_cast<const f32x4*>(bias + i * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
  }
  f32x4 low[6];
  if (lower) {
    // this level's block is half of the next coarser level's block: the class there follows from the class here and the
    // parity of the pixel
    const ImgGeom gl = glow[img];
    const int lr = (rc == 0 && !(y & 1)) ? 0 : (rc == 2 && (y & 1)) ? 2 : 1;
    const int lc = (cc == 0 && !(x & 1)) ? 0 : (cc == 2 && (x & 1)) ? 2 : 1;
    const float* lp = lower + ((lr * 3 + lc) * low_plane + gl.off + (long long)min(y >> 1, gl.H - 1) * gl.W + min(x >> 1, gl.W - 1)) * 24;
#pragma unroll
    for (int i = 0; i < 6; i++) low[i] = *reinterpret_cast<const f32x4*>(lp + i * 4);
  }
#pragma unroll
  for (int i = 0; i < 3; i++) {
    const int idx = tid + 256 * i;
    if (idx < 576) { const int row = idx / 6, c4 = idx - row * 6; *reinterpret_cast<f32x4*>(wlds + row * ROW + c4 * 4) = wreg[i]; }
  }
  __syncthreads();
#pragma unroll
  for (int t = 0; t < 4; t++) {
    if (((t >> 1) && !two_y) || ((t & 1) && !two_x)) continue;   // uniform
    const bool ok = (okmask >> t) & 1;
#pragma unroll
    for (int i = 0; i < 6; i++) {
      f32x4 bv = b[t][i] * sc[i];
      if (!ok) bv = f32x4{0.f, 0.f, 0.f, 0.f};
      const f32x4 wv = *reinterpret_cast<const f32x4*>(wlds + (t * 24 + wl) * ROW + i * 4);
#pragma unroll
      for (int e = 0; e < 4; e++) {
        acc[0] = __builtin_amdgcn_mfma_f32_4x4x1f32(wv[e], bv[e], acc[0], 4, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_4x4x1f32(wv[e], bv[e], acc[1], 4, 1, 0);
        acc[2] = __builtin_amdgcn_mfma_f32_4x4x1f32(wv[e], bv[e], acc[2], 4, 2, 0);
        acc[3] = __builtin_amdgcn_mfma_f32_4x4x1f32(wv[e], bv[e], acc[3], 4, 3, 0);
        acc[4] = __builtin_amdgcn_mfma_f32_4x4x1f32(wv[e], bv[e], acc[4], 4, 4, 0);
        acc[5] = __builtin_amdgcn_mfma_f32_4x4x1f32(wv[e], bv[e], acc[5], 4, 5, 0);
      }
    }
  }
  if (valid) {
    float* o = V + (cls * plane + g.off + p) * 24;   // class-major planes: a wave stores 64 x 96 contiguous bytes
#pragma unroll
    for (int i = 0; i < 6; i++) *reinterpret_cast<f32x4*>(o + i * 4) = lower ? acc[i] + low[i] : acc[i];
  }
}
void fpn_class(hipStream_t st, const float* z, int ldz, const float* scale, int ld_s, const ImgGeom* geom, int n_img,
               long long max_pix, const float* Wcls, const float* bias, const float* lower, const ImgGeom* glow, long long low_plane,
               float* V, long long plane) {
  if (n_img <= 0) return;
  RT_LAUNCH(k_fpn_class, dim3((unsigned)((max_pix + 255) / 256), n_img, 9), dim3(256), 0, st, z, ldz, scale, ld_s, geom, Wcls,
            bias, lower, glow, low_plane, V, plane);
}

// W'[img][tap][n][c] = sum_m Wlat[c][m] * s[img][m] * Wm[tap][n][m]   (c < cin; padding channels of the cf-wide rows are zero).
// A workgroup per (tap, image): the scaled lateral matrix and the tap's [24][C] slice in LDS, a thread per output.
__global__ __launch_bounds__(256) void k_fpn_compose(const float* __restrict__ Wlat, int cin, int C, const float* __restrict__ s,
                                                     const float* __restrict__ Wm, int cf, float* __restrict__ out) {
  extern __shared__ float sm[];   // [cin][C + 1] | [24][C + 1]
  const int img = blockIdx.y, tap = blockIdx.x, ldc = C + 1;
  float* wl = sm;
  float* wm = sm + cin * ldc;
  for (int i = threadIdx.x; i < cin * C; i += 256) { const int c = i / C, m = i - c * C; wl[c * ldc + m] = Wlat[i] * s[(long long)img * C + m]; }
  for (int i = threadIdx.x; i < 24 * C; i += 256) { const int n = i / C, m = i - n * C; wm[n * ldc + m] = Wm[(long long)tap * 24 * C + i]; }
  __syncthreads();
  for (int idx = threadIdx.x; idx < 24 * cf; idx += 256) {   // n * cf + c
    const int n = idx / cf, c = idx - n * cf;
    float acc = 0.f;
    if (c < cin)
      for (int m = 0; m < C; m++) acc = fmaf(wl[c * ldc + m], wm[n * ldc + m], acc);
    out[((long long)img * 9 + tap) * 24 * cf + idx] = acc;
  }
}
void fpn_compose(hipStream_t st, const float* Wlat, int cin, int C, const float* scale, const float* Wm, int cf, int n_img, float* out) {
  if (n_img <= 0) return;
  RT_LAUNCH(k_fpn_compose, dim3(9, n_img), dim3(256), (size_t)(cin + 24) * (C + 1) * sizeof(float), st, Wlat, cin, C, scale, Wm, cf, out);
}

}  // namespace nn
}  // namespace rt
