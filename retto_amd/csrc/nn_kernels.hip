// Neural-network kernels for gfx950 (CDNA4): fp32 NHWC, 64-wide wavefronts,
// v_mfma_f32_16x16x4_f32 for the dense 1x1 / 3x3 / 1x3 contractions (exact fp32
// fmaf chains: no reduced-precision path exists on gfx950 and parity is fp32),
// VALU for the depthwise / pooling / normalisation work which is HBM bound.
//
// The networks these serve replace ONNX Runtime's Session::run in
// /root/reference/retto-core/src/worker/ort_worker.rs:189-220.
#include "nn.h"
#include "nn_dev.h"

#include <algorithm>
#include <cstdlib>

namespace rt {
namespace nn {

// ---------------------------------------------------------------------------
// MFMA micro-kernel shared by gemm and conv_sp.
// LDS rows are KC(32)+4 floats.  Weights are the MFMA "A" operand (M dim = cout),
// pixels the "B" operand (N dim = pixel), so each lane ends up holding 4 consecutive
// output channels of one pixel -> one 16-byte store per accumulator.
// Lane l: r = l & 15, q = l >> 4.  Within a 16-deep k group lane quarter q reads
// k = 4q..4q+3 as one ds_read_b128 and feeds element s to MFMA step s; the weight
// operand uses the same (q, s) -> k map, which is all the instruction requires.
// ---------------------------------------------------------------------------
constexpr int LROW = KC + 4;

template <int NT>
__device__ __forceinline__ void mma_chunk(const float* __restrict__ xs0, const float* __restrict__ xs1,
                                          const float* __restrict__ ws, int nt_valid, f32x4 (&acc)[2][NT], int r,
                                          int q) {
#pragma unroll
  for (int g = 0; g < KC / 16; g++) {
    f32x4 a0 = *reinterpret_cast<const f32x4*>(xs0 + g * 16 + 4 * q);
    f32x4 a1 = *reinterpret_cast<const f32x4*>(xs1 + g * 16 + 4 * q);
#pragma unroll
    for (int nt = 0; nt < NT; nt++) {
      // (kept as a run-time test per column tile: the branch-free form -- what the wide kernel does -- measured 10 %
      //  slower here, hipcc then hoists every weight fragment read ahead of the MFMAs)
      if (nt < nt_valid) {
        f32x4 b = *reinterpret_cast<const f32x4*>(ws + (nt * 16 + r) * LROW + g * 16 + 4 * q);
#pragma unroll
        for (int s = 0; s < 4; s++) {
          acc[0][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[s], a0[s], acc[0][nt], 0, 0, 0);
          acc[1][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[s], a1[s], acc[1][nt], 0, 0, 0);
        }
      }
    }
  }
}

template <int NT>
__device__ __forceinline__ void epilogue_store(f32x4 (&acc)[2][NT], int nt_valid, const Epilogue& epi, int n0, int N,
                                               int nstore, float* __restrict__ yrow0, float* __restrict__ yrow1,
                                               bool v0, bool v1, const float* res0, const float* res1, int q) {
  act_dispatch(epi.act, epi.has_lab, res0 != nullptr, [&](auto at, auto lt, auto rt_) {
    constexpr int A = decltype(at)::value, L = decltype(lt)::value, RES = decltype(rt_)::value;
#pragma unroll
    for (int nt = 0; nt < NT; nt++) {
      if (nt >= nt_valid) continue;
      int col = n0 + nt * 16 + q * 4;
      if (col >= nstore) continue;
      f32x4 bias = {0.f, 0.f, 0.f, 0.f};
      if (epi.bias) bias = *reinterpret_cast<const f32x4*>(epi.bias + col);
#pragma unroll
      for (int mt = 0; mt < 2; mt++) {
        bool valid = mt == 0 ? v0 : v1;
        if (!valid) continue;
        float* yr = mt == 0 ? yrow0 : yrow1;
        const float* rr = mt == 0 ? res0 : res1;
        f32x4 v = acc[mt][nt];
        f32x4 o;
#pragma unroll
        for (int j = 0; j < 4; j++) {
          float t = epi_val<A, L>(v[j] + bias[j], epi.act, epi.has_lab, epi.lab_a, epi.lab_c);
          if (RES) t += rr[col + j];
          o[j] = (col + j < N) ? t : 0.0f;
        }
        *reinterpret_cast<f32x4*>(yr + col) = o;
      }
    }
  });
}

// ---------------------------------------------------------------------------
// Flat GEMM: rows = pixels (any ragged batch is just a longer M).
// Block 256 threads (4 waves), tile 128 rows x 16*NT cols, K in 32-chunks.
// ---------------------------------------------------------------------------
// BF (round 5): the slab fetch through buffer resources -- fixed per-thread byte offsets, the slab advance as the scalar offset, rows
// beyond M / Npad and the K tail out of range (zeros) -- instead of a predicated 64-bit-address load per vector: every VALU
// instruction of an fp32 MFMA loop is paid in MFMA time, and the address / predicate arithmetic was ~100 of them per slab.
template <int NT, int EPIM = 0, bool BF = false>
__global__ __launch_bounds__(256, 4) void k_gemm(const float* __restrict__ A, int lda, long long M, int K,
                                              const float* __restrict__ Wp, int N, int Npad, float* __restrict__ C,
                                              int ldc, int coff, Epilogue epi) {
  __shared__ __attribute__((aligned(16))) float lds[(128 + 16 * NT) * LROW];
  float* xs = lds;
  float* ws = lds + 128 * LROW;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 15, q = lane >> 4;
  const long long m0 = (long long)blockIdx.x * 128;
  const int n0 = blockIdx.y * 16 * NT;
  const int nt_valid = min(NT, (Npad - n0) / 16);
  f32x4 acc[2][NT];
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int j = 0; j < NT; j++) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nkc = (K + KC - 1) / KC;
  // next K-slab (128 activation rows + 16*NT weight rows) in registers while the current one is multiplied
  constexpr int W_LD = (16 * NT * 8 + 255) / 256;
  f32x4 pa[4], pw[W_LD];
  // buffer form: descriptors over this tile's rows / this block's weight rows, per-thread offsets fixed for the whole K loop
  unsigned aoff[4], woff[W_LD];
  __amdgpu_buffer_rsrc_t ars, wrs;
  if (BF) {
    const long long rows_here = min((long long)128, M - m0);
    ars = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(A + m0 * lda), 0, (unsigned)(rows_here * lda * 4), 0x00020000);
    wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Wp + (long long)n0 * KC), 0, 0x7fffffffu, 0x00020000);
#pragma unroll
    for (int i = 0; i < 4; i++) { const int idx = tid + 256 * i; aoff[i] = (unsigned)((idx >> 3) * lda * 4 + (idx & 7) * 16); }
#pragma unroll
    for (int i = 0; i < W_LD; i++) {
      const int idx = tid + 256 * i, row = idx >> 3;
      woff[i] = (idx < 16 * NT * 8 && n0 + row < Npad) ? (unsigned)(row * KC * 4 + (idx & 7) * 16) : 0x80000000u;
    }
  }
  auto fetch = [&](int kc) {
    const int k0 = kc * KC;
    if (BF) {
      const bool tail = k0 + KC > K;   // (uniform) the last slab of a K that is not a multiple of 32: vectors beyond K are zeros
#pragma unroll
      for (int i = 0; i < 4; i++) {
        const unsigned o = (tail && k0 + (int)((tid + 256 * i) & 7) * 4 >= K) ? 0x80000000u : aoff[i];
        pa[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ars, o, k0 * 4, 0));
      }
#pragma unroll
      for (int i = 0; i < W_LD; i++) pw[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wrs, woff[i], kc * Npad * KC * 4, 0));
      return;
    }
#pragma unroll
    for (int i = 0; i < 4; i++) {
      int idx = tid + 256 * i;
      int row = idx >> 3, c4 = idx & 7;
      pa[i] = f32x4{0.f, 0.f, 0.f, 0.f};
      long long m = m0 + row;
      if (m < M && k0 + c4 * 4 < K) pa[i] = *reinterpret_cast<const f32x4*>(A + m * lda + k0 + c4 * 4);
    }
#pragma unroll
    for (int i = 0; i < W_LD; i++) {
      int idx = tid + 256 * i, row = idx >> 3, c4 = idx & 7;
      pw[i] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (idx < 16 * NT * 8 && n0 + row < Npad) pw[i] = *reinterpret_cast<const f32x4*>(Wp + ((long long)kc * Npad + n0 + row) * KC + c4 * 4);
    }
  };
  fetch(0);
  for (int kc = 0; kc < nkc; kc++) {
#pragma unroll
    for (int i = 0; i < 4; i++) {
      int idx = tid + 256 * i;
      *reinterpret_cast<f32x4*>(xs + (idx >> 3) * LROW + (idx & 7) * 4) = pa[i];
    }
#pragma unroll
    for (int i = 0; i < W_LD; i++) {
      int idx = tid + 256 * i;
      if (idx < 16 * NT * 8) *reinterpret_cast<f32x4*>(ws + (idx >> 3) * LROW + (idx & 7) * 4) = pw[i];
    }
    __syncthreads();
    if (kc + 1 < nkc) fetch(kc + 1);
    mma_chunk<NT>(xs + (wave * 32 + r) * LROW, xs + (wave * 32 + 16 + r) * LROW, ws, nt_valid, acc, r, q);
    __syncthreads();
  }
  long long ma = m0 + wave * 32 + r, mb = ma + 16;
  if (EPIM) {
    // CTC head (see k_gemm_wide's EPIM): softmax statistics of this block's 16*NT columns per row; a wave owns
    // whole rows here, so the reduction is in-lane over the columns and across the 4 lane quarters only
#pragma unroll
    for (int mt = 0; mt < 2; mt++) {
      float m = -INFINITY, sum = 0.f;
      int mi = 0x7fffffff;
#pragma unroll
      for (int nt = 0; nt < NT; nt++) {
        if (nt >= nt_valid) continue;
        const int col = n0 + nt * 16 + q * 4;
        f32x4 bias = {0.f, 0.f, 0.f, 0.f};
        if (epi.bias) bias = *reinterpret_cast<const f32x4*>(epi.bias + col);
#pragma unroll
        for (int j = 0; j < 4; j++) {
          if (col + j >= N) continue;
          const float v = acc[mt][nt][j] + bias[j];
          if (v > m) { sum = sum * __expf(m - v) + 1.0f; m = v; mi = col + j; }
          else sum += __expf(v - m);
        }
      }
#pragma unroll
      for (int d = 16; d < 64; d <<= 1) {
        const float om = __shfl_xor(m, d), os = __shfl_xor(sum, d);
        const int oi = __shfl_xor(mi, d);
        const float nm = fmaxf(m, om);
        sum = sum * (m == nm ? 1.0f : __expf(m - nm)) + os * (om == nm ? 1.0f : __expf(om - nm));
        if (om > m || (om == m && oi < mi)) mi = oi;
        m = nm;
      }
      const long long row = mt == 0 ? ma : mb;
      if (q == 0 && row < M) {
        const long long o = row * epi.am_tiles + blockIdx.y;
        epi.am_max[o] = m; epi.am_sum[o] = sum; epi.am_idx[o] = mi;
      }
    }
    return;
  }
  const int nstore = (N + 3) & ~3;
  epilogue_store<NT>(acc, nt_valid, epi, n0, N, nstore, C + ma * ldc + coff, C + mb * ldc + coff, ma < M, mb < M,
                     epi.residual ? epi.residual + ma * epi.ld_res : nullptr,
                     epi.residual ? epi.residual + mb * epi.ld_res : nullptr, q);
}

// ---------------------------------------------------------------------------
// Streaming GEMM for the thin layers (K <= 16*KG <= 128, N <= 16*NT <= 128), which are HBM bound:
// the whole weight matrix sits in LDS for the life of the block and every wave walks over 32-row
// tiles on its own -- the activation fragments go from global memory straight into the MFMA
// operand registers (lane (r, q) loads A[m0 + r][16g + 4q .. +3], the same (q, s) -> k map the LDS
// path uses), the next tile's fragments are in flight while the current one is multiplied, and
// there is no barrier after the weights are staged.  Same k order per accumulator as k_gemm:
// bit-identical results.
// ---------------------------------------------------------------------------
template <int NT, int KG>
__global__ __launch_bounds__(256) void k_gemm_stream(const float* __restrict__ A, int lda, long long M, int K,
                                                     const float* __restrict__ Wp, int N, int Npad,
                                                     float* __restrict__ C, int ldc, int coff, Epilogue epi) {
  constexpr int NKC = (KG + 1) / 2;
  __shared__ __attribute__((aligned(16))) float ws[NKC * 16 * NT * LROW];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 15, q = lane >> 4;
  for (int idx = tid; idx < NKC * 16 * NT * 8; idx += 256) {
    const int c4 = idx & 7, row = (idx >> 3) % (16 * NT), kc = (idx >> 3) / (16 * NT);
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (row < Npad) v = *reinterpret_cast<const f32x4*>(Wp + ((long long)kc * Npad + row) * KC + c4 * 4);
    *reinterpret_cast<f32x4*>(ws + (kc * 16 * NT + row) * LROW + c4 * 4) = v;
  }
  __syncthreads();
  const int nt_valid = min(NT, Npad / 16);
  const long long n_tiles = (M + 31) >> 5, stride = (long long)gridDim.x * 4;
  const int nstore = (N + 3) & ~3;
  constexpr int PD = 1;  // tiles of A fragments in flight per wave (2 and 4 measured slower: 0.57 -> 0.68 ms at K = 32)
  f32x4 an[PD][2][KG];
  auto fetch = [&](long long tile, f32x4 (&dst)[2][KG]) {
#pragma unroll
    for (int mt = 0; mt < 2; mt++) {
      const long long m = tile * 32 + mt * 16 + r;
      const float* row = A + (m < M ? m : 0) * lda + 4 * q;
#pragma unroll
      for (int g = 0; g < KG; g++) {
        dst[mt][g] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (m < M && g * 16 + 4 * q < K) dst[mt][g] = *reinterpret_cast<const f32x4*>(row + g * 16);
      }
    }
  };
  const long long tile0 = (long long)blockIdx.x * 4 + wave;
#pragma unroll
  for (int u = 0; u < PD; u++)
    if (tile0 + u * stride < n_tiles) fetch(tile0 + u * stride, an[u]);
  for (long long base = tile0; base < n_tiles; base += stride * PD) {
#pragma unroll
    for (int u = 0; u < PD; u++) {
      const long long tile = base + u * stride;
      if (tile >= n_tiles) break;
      f32x4 a[2][KG];
#pragma unroll
      for (int mt = 0; mt < 2; mt++)
#pragma unroll
        for (int g = 0; g < KG; g++) a[mt][g] = an[u][mt][g];
      if (tile + stride * PD < n_tiles) fetch(tile + stride * PD, an[u]);
      f32x4 acc[2][NT];
#pragma unroll
      for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < NT; j++) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int g = 0; g < KG; g++) {
#pragma unroll
        for (int nt = 0; nt < NT; nt++) {
          if (nt < nt_valid) {
            const f32x4 b = *reinterpret_cast<const f32x4*>(ws + ((g >> 1) * 16 * NT + nt * 16 + r) * LROW + (g & 1) * 16 + 4 * q);
#pragma unroll
            for (int s2 = 0; s2 < 4; s2++) {
              acc[0][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[s2], a[0][g][s2], acc[0][nt], 0, 0, 0);
              acc[1][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[s2], a[1][g][s2], acc[1][nt], 0, 0, 0);
            }
          }
        }
      }
      const long long ma = tile * 32 + r, mb = ma + 16;
      epilogue_store<NT>(acc, nt_valid, epi, 0, N, nstore, C + ma * ldc + coff, C + mb * ldc + coff, ma < M, mb < M,
                         epi.residual ? epi.residual + ma * epi.ld_res : nullptr,
                         epi.residual ? epi.residual + mb * epi.ld_res : nullptr, q);
    }
  }
}

// ---------------------------------------------------------------------------
// Wide GEMM for N > 128: block tile (16*MT*WM) rows x (16*NT*WN) cols so that a pass reads
// each activation row once (the narrow kernel re-reads A once per 128 columns), WM x WN waves,
// and the next K-slab is prefetched into registers while the MFMAs of the current one run.
// ---------------------------------------------------------------------------
template <int MT, int NT, int WM, int WN, int RASTER = 0, int DBG = 0, int ASC = 0, int EPIM = 0, int DBUF = 0>
__global__ __launch_bounds__(64 * WM * WN, (NT > 8 ? 2 : 1)) void k_gemm_wide(const float* __restrict__ A, int lda, long long M, int K,
                                                             const float* __restrict__ Wp, int N, int Npad,
                                                             float* __restrict__ C, int ldc, int coff, Epilogue epi) {
  constexpr int NTHR = 64 * WM * WN, BM = 16 * MT * WM, BN = 16 * NT * WN;
  constexpr int A_LD = (BM * 8 + NTHR - 1) / NTHR, W_LD = (BN * 8 + NTHR - 1) / NTHR;
  constexpr int SC_MAXK = 512;  // ASC: the squeeze-excite scales of the (at most 2) images under this row tile
  // DBUF: two LDS stages -- slab kc+1 is written while other waves still multiply slab kc, one barrier per slab
  constexpr int STAGE = (BM + BN) * LROW;
  __shared__ __attribute__((aligned(16))) float lds[(DBUF ? 2 : 1) * STAGE + (ASC ? 2 * SC_MAXK : 0)];
  float* sc = lds + (DBUF ? 2 : 1) * STAGE;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 15, q = lane >> 4;
  const int wm = wave / WN, wn = wave % WN;
  long long mb = blockIdx.x;
  int nb = blockIdx.y;
  if (RASTER) {
    // XCD-aware order: block ids are dealt round-robin to the 8 XCDs, so the column blocks of
    // one row block are given consecutive slots on the SAME XCD and share its L2 for the A rows.
    const int ncol = (Npad + BN - 1) / BN;
    const long long id = blockIdx.x, slot = id >> 3;
    mb = (slot / ncol) * 8 + (id & 7);
    nb = (int)(slot % ncol);
    if (mb * BM >= M) return;
  }
  const long long m0 = mb * BM;
  const int n0 = nb * BN;
  const int nt_valid = max(0, min(NT, (Npad - n0) / 16 - wn * NT));
  f32x4 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; i++)
#pragma unroll
    for (int j = 0; j < NT; j++) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x4 pa[A_LD], pw[W_LD];
  const int nkc = (K + KC - 1) / KC;
  auto fetch = [&](int kc) {
    const int k0 = kc * KC;
#pragma unroll
    for (int i = 0; i < A_LD; i++) {
      int idx = tid + NTHR * i, row = idx >> 3, c4 = idx & 7;
      long long m = m0 + row;
      pa[i] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (row < BM && m < M && k0 + c4 * 4 < K) pa[i] = *reinterpret_cast<const f32x4*>(A + m * lda + k0 + c4 * 4);
    }
#pragma unroll
    for (int i = 0; i < W_LD; i++) {
      int idx = tid + NTHR * i, row = idx >> 3, c4 = idx & 7;
      pw[i] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (row < BN && n0 + row < Npad)
        pw[i] = *reinterpret_cast<const f32x4*>(Wp + ((long long)kc * Npad + n0 + row) * KC + c4 * 4);
    }
  };
  int sc_bnd = 0;  // tile-relative row where the second image starts (>= BM: none)
  if (ASC) {
    const int img = epi.a_tab[2 * mb];
    const long long sc_boundary = epi.a_tab[2 * mb + 1];
    sc_bnd = (int)min((long long)BM, sc_boundary - m0);
    for (int i = tid; i < 2 * (K >> 2); i += NTHR) {
      const int which = i >= (K >> 2), k4 = i - which * (K >> 2);
      // (the second image only exists when the boundary falls inside the tile)
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (!which || sc_boundary < m0 + BM) v = *reinterpret_cast<const f32x4*>(epi.a_scale + (long long)(img + which) * epi.ld_scale + k4 * 4);
      *reinterpret_cast<f32x4*>(sc + which * SC_MAXK + k4 * 4) = v;
    }
    __syncthreads();
  }
  auto stash = [&](int kc) {
    float* xs = lds + (DBUF ? (kc & 1) * STAGE : 0);
    float* ws = xs + BM * LROW;
#pragma unroll
    for (int i = 0; i < A_LD; i++) {
      int idx = tid + NTHR * i, row = idx >> 3, c4 = idx & 7;
      if (row < BM) {
        f32x4 v = pa[i];
        if (ASC) {
          const int k = kc * KC + c4 * 4;
          if (k < K) v *= *reinterpret_cast<const f32x4*>(sc + (row >= sc_bnd ? SC_MAXK : 0) + k);
        }
        *reinterpret_cast<f32x4*>(xs + row * LROW + c4 * 4) = v;
      }
    }
#pragma unroll
    for (int i = 0; i < W_LD; i++) {
      int idx = tid + NTHR * i, row = idx >> 3, c4 = idx & 7;
      if (row < BN) *reinterpret_cast<f32x4*>(ws + row * LROW + c4 * 4) = pw[i];
    }
  };
  fetch(0);
  stash(0);
  __syncthreads();
  // One K slab.  NG = 16-deep groups multiplied (compile time): the slabs run as a branch-free loop over the full ones
  // plus, when K leaves a tail of <= 16 (K = 240), one last slab with a single group.  A run-time test between the two
  // MFMA groups (the earlier form) splits the scheduling region: 5-9 % slower on every shape.  (Compiling the tail out
  // for K % 32 == 0 was tried: hipcc then spills 22 instead of 8 VGPRs on the 256-row tile and loses the gain.)
  auto slab = [&](int kc, auto ngtag) {
    constexpr int NG = decltype(ngtag)::value;
    if (DBG == 0 && kc + 1 < nkc) fetch(kc + 1);
    const float* xs = lds + (DBUF ? (kc & 1) * STAGE : 0);
    const float* ws = xs + BM * LROW;
    const float* xr = xs + (wm * MT * 16 + r) * LROW;
    const float* wr = ws + (wn * NT * 16 + r) * LROW;
    // Fragments of both 16-deep groups are requested up front; the MFMA block of group 0 only
    // waits for its own reads (in-order LDS returns), hiding group 1's latency.  Branch-free:
    // padding tiles multiply zero weights (W rows past Npad are zero in LDS) and are dropped at
    // the store.
    if constexpr (NT > 8) {
      // fat waves (few rows x all columns): the weight fragments are streamed one column tile at a time so that
      // the accumulators (MT x NT x 4 registers) leave room for two workgroups per CU
#pragma unroll
      for (int g = 0; g < NG; g++) {
        f32x4 a[MT];
#pragma unroll
        for (int mt = 0; mt < MT; mt++) a[mt] = *reinterpret_cast<const f32x4*>(xr + mt * 16 * LROW + g * 16 + 4 * q);
#pragma unroll
        for (int nt = 0; nt < NT; nt++) {
          const f32x4 b = *reinterpret_cast<const f32x4*>(wr + nt * 16 * LROW + g * 16 + 4 * q);
#pragma unroll
          for (int s = 0; s < 4; s++)
#pragma unroll
            for (int mt = 0; mt < MT; mt++)
              acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[s], a[mt][s], acc[mt][nt], 0, 0, 0);
        }
      }
    } else {
    f32x4 a[NG][MT], b[NG][NT];
#pragma unroll
    for (int g = 0; g < NG; g++) {
#pragma unroll
      for (int mt = 0; mt < MT; mt++) a[g][mt] = *reinterpret_cast<const f32x4*>(xr + mt * 16 * LROW + g * 16 + 4 * q);
#pragma unroll
      for (int nt = 0; nt < NT; nt++) b[g][nt] = *reinterpret_cast<const f32x4*>(wr + nt * 16 * LROW + g * 16 + 4 * q);
    }
#pragma unroll
    for (int g = 0; g < NG; g++) {
#pragma unroll
      for (int s = 0; s < 4; s++)
#pragma unroll
        for (int nt = 0; nt < NT; nt++)
#pragma unroll
          for (int mt = 0; mt < MT; mt++)
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[g][nt][s], a[g][mt][s], acc[mt][nt], 0, 0, 0);
    }
    }
    if (DBUF) {
      if (kc + 1 < nkc) stash(kc + 1);
      __syncthreads();
    } else {
      if (DBG != 2) __syncthreads();
      if (DBG == 0 && kc + 1 < nkc) { stash(kc + 1); __syncthreads(); }
      if (DBG == 1) __syncthreads();
    }
  };
  {
    const int tail = K - (nkc - 1) * KC;               // depth of the last slab: 1..32
    const int n2 = tail > 16 ? nkc : nkc - 1;          // slabs with both 16-deep groups
    for (int kc = 0; kc < n2; kc++) slab(kc, IntTag<KC / 16>{});
    if (n2 < nkc) slab(nkc - 1, IntTag<1>{});
  }
  if (EPIM) {
    // Softmax statistics of this column tile, per row: (max, first column of the max, sum exp(v - max)).
    // In-lane over the lane's 4*NT columns (ascending), then the 4 lane quarters (ascending columns),
    // then the WN waves through LDS -- fixed order, lowest column wins ties like the reference's argmax.
    float* rm = lds;
    float* rs = lds + BM * WN;
    int* ri = reinterpret_cast<int*>(lds + 2 * BM * WN);
#pragma unroll
    for (int mt = 0; mt < MT; mt++) {
      float m = -INFINITY, sum = 0.f;
      int mi = 0x7fffffff;
#pragma unroll
      for (int nt = 0; nt < NT; nt++) {
        if (nt >= nt_valid) continue;
        const int col = n0 + (wn * NT + nt) * 16 + q * 4;
        f32x4 bias = {0.f, 0.f, 0.f, 0.f};
        if (epi.bias) bias = *reinterpret_cast<const f32x4*>(epi.bias + col);
#pragma unroll
        for (int j = 0; j < 4; j++) {
          if (col + j >= N) continue;
          const float v = acc[mt][nt][j] + bias[j];
          if (v > m) { sum = sum * __expf(m - v) + 1.0f; m = v; mi = col + j; }
          else sum += __expf(v - m);
        }
      }
#pragma unroll
      for (int d = 16; d < 64; d <<= 1) {
        const float om = __shfl_xor(m, d), os = __shfl_xor(sum, d);
        const int oi = __shfl_xor(mi, d);
        const float nm = fmaxf(m, om);
        sum = sum * (m == nm ? 1.0f : __expf(m - nm)) + os * (om == nm ? 1.0f : __expf(om - nm));
        if (om > m || (om == m && oi < mi)) mi = oi;
        m = nm;
      }
      if (q == 0) {
        const int row = (wm * MT + mt) * 16 + r;
        rm[row * WN + wn] = m; rs[row * WN + wn] = sum; ri[row * WN + wn] = mi;
      }
    }
    __syncthreads();
    for (int row = tid; row < BM; row += NTHR) {
      if (m0 + row >= M) break;
      float m = rm[row * WN], sum = rs[row * WN];
      int mi = ri[row * WN];
      for (int w = 1; w < WN; w++) {
        const float om = rm[row * WN + w], os = rs[row * WN + w];
        const int oi = ri[row * WN + w];
        const float nm = fmaxf(m, om);
        sum = sum * (m == nm ? 1.0f : __expf(m - nm)) + os * (om == nm ? 1.0f : __expf(om - nm));
        if (om > m || (om == m && oi < mi)) mi = oi;
        m = nm;
      }
      const long long o = (m0 + row) * epi.am_tiles + nb;
      epi.am_max[o] = m; epi.am_sum[o] = sum; epi.am_idx[o] = mi;
    }
    return;
  }
  const int nstore = (N + 3) & ~3;
  act_dispatch(epi.act, epi.has_lab, epi.residual != nullptr, [&](auto at, auto lt, auto rt_) {
    constexpr int A = decltype(at)::value, L = decltype(lt)::value, RES = decltype(rt_)::value;
#pragma unroll
    for (int nt = 0; nt < NT; nt++) {
      if (nt >= nt_valid) continue;
      int col = n0 + (wn * NT + nt) * 16 + q * 4;
      if (col >= nstore) continue;
      f32x4 bias = {0.f, 0.f, 0.f, 0.f};
      if (epi.bias) bias = *reinterpret_cast<const f32x4*>(epi.bias + col);
#pragma unroll
      for (int mt = 0; mt < MT; mt++) {
        long long m = m0 + (wm * MT + mt) * 16 + r;
        if (m >= M) continue;
        f32x4 o;
#pragma unroll
        for (int j = 0; j < 4; j++) {
          float t = epi_val<A, L>(acc[mt][nt][j] + bias[j], epi.act, epi.has_lab, epi.lab_a, epi.lab_c);
          if (RES) t += epi.residual[m * epi.ld_res + col + j];
          o[j] = (col + j < N) ? t : 0.0f;
        }
        *reinterpret_cast<f32x4*>(C + m * ldc + coff + col) = o;
      }
    }
  });
}

// ---------------------------------------------------------------------------
// Fused thin LCNetV3 block (3x3 depthwise, strides (1,1) / (2,1) / (2,2), no SE, C_in <= 64, N <= 128):
//   y = epi_pw( W_pw . lab(act(dw3x3(x) + b_dw)) )
// These layers are HBM bound (a few channels per pixel): run separately they move 3*C_in + C_out floats per
// pixel (depthwise read + write, GEMM read + write); fused, the depthwise result only ever exists as the
// GEMM's A tile in LDS and the block moves C_in (+ halo) + C_out.  One workgroup = TH waves on a TH x 16 pixel
// tile with ALL input channels: input patch ((TH+2) x 18 pixels) -> LDS, depthwise from LDS (taps in registers) -> A tile
// [K/32][TH*16][36] in LDS, then the same mma_chunk / epilogue as k_gemm (TH/2 x 2 waves, 32 pixels x 16*NT
// columns each) against the whole pointwise weight matrix, resident in LDS for the `tiles_per_block` tiles a
// block walks.  TH = 4 (64 pixels, 256 threads, 22-66 KB of LDS) keeps several workgroups per CU.
// Same accumulation order as k_dwconv_rows + k_gemm: bit-identical to the unfused pair.
// ---------------------------------------------------------------------------
template <int C4, int NT, int TH, int SH, int SW>
__global__ __launch_bounds__(64 * TH) void k_lc_thin(const float* __restrict__ x, const ImgGeom* __restrict__ gin,
                                                     const ImgGeom* __restrict__ gout, int C,
                                                     const float* __restrict__ Wd, const float* __restrict__ bd, int dw_act,
                                                     int dw_has_lab, float dw_a, float dw_c, const float* __restrict__ Wp,
                                                     int N, int Npad, float* __restrict__ y, int ldy, Epilogue epi,
                                                     int tiles_per_block) {
  constexpr int CP = C4 * 4, TW = 16, ROWS = TH * TW, NTHR = 64 * TH, PPITCH = CP + 4;
  constexpr int PH = (TH - 1) * SH + 3, PW = (TW - 1) * SW + 3;  // input patch of a TH x 16 output tile
  constexpr int NKC = (CP + KC - 1) / KC, NCOL = 32 * NT;
  constexpr int NPF = (PH * PW * C4 + NTHR - 1) / NTHR;
  __shared__ __attribute__((aligned(16))) float patch[PH * PW * PPITCH];
  __shared__ __attribute__((aligned(16))) float at[NKC * ROWS * LROW];
  __shared__ __attribute__((aligned(16))) float wt[NKC * NCOL * LROW];
  const ImgGeom g = gout[blockIdx.y], gi = gin[blockIdx.y];
  const int tiles_x = (g.W + TW - 1) / TW, n_tiles = tiles_x * ((g.H + TH - 1) / TH);
  int tile = blockIdx.x * tiles_per_block;
  if (tile >= n_tiles) return;
  const int tile_end = min(n_tiles, tile + tiles_per_block);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 15, q = lane >> 4;
  const int wm = wave >> 1, wn = wave & 1;
  for (int idx = tid; idx < NKC * NCOL * 8; idx += NTHR) {
    const int c4i = idx & 7, row = (idx >> 3) % NCOL, kc = (idx >> 3) / NCOL;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (row < Npad) v = *reinterpret_cast<const f32x4*>(Wp + ((long long)kc * Npad + row) * KC + c4i * 4);
    *reinterpret_cast<f32x4*>(wt + (kc * NCOL + row) * LROW + c4i * 4) = v;
  }
  for (int idx = tid; idx < NKC * ROWS * LROW / 4; idx += NTHR) *reinterpret_cast<f32x4*>(at + idx * 4) = f32x4{0.f, 0.f, 0.f, 0.f};
  // depthwise work items: strips of SL pixels x C4 channel groups, one per thread (SL shrinks with C4 so that
  // the threads stay busy: 16 groups -> 4 pixels, 8 -> 2, 4 -> 1); taps and bias of the item's group in registers
  constexpr int SL = C4 >= 12 ? 4 : (C4 == 8 ? 2 : 1), ITEMS = (ROWS / SL) * C4;
  static_assert(ITEMS <= NTHR, "one depthwise item per thread");
  const int ic4 = tid % C4, ip0 = (tid / C4) * SL, ipy = ip0 / TW, ipx = ip0 % TW;
  f32x4 dww[9], dwb;
#pragma unroll
  for (int t = 0; t < 9; t++) dww[t] = *reinterpret_cast<const f32x4*>(Wd + t * CP + ic4 * 4);
  dwb = *reinterpret_cast<const f32x4*>(bd + ic4 * 4);
  const int nt_valid = max(0, min(NT, Npad / 16 - wn * NT));
  const int nstore = (N + 3) & ~3;
  // Input patch of a tile -> registers (zero outside the image).  The next tile's patch is requested right
  // after the depthwise phase, so its latency overlaps the MFMAs and the stores of the current tile (hipcc
  // drains vmcnt to 0 before the LDS writes anyway, so a longer prefetch distance buys nothing).
  f32x4 pf[NPF];
  auto fetch = [&](int t) {
    const int ty0 = (t / tiles_x) * TH, tx0 = (t % tiles_x) * TW;
#pragma unroll
    for (int i = 0; i < NPF; i++) {
      const int e = tid + NTHR * i;
      pf[i] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (e < PH * PW * C4) {
        const int c4i = e % C4, px = e / C4, iy = ty0 * SH - 1 + px / PW, ix = tx0 * SW - 1 + px % PW;
        if (iy >= 0 && iy < gi.H && ix >= 0 && ix < gi.W)
          pf[i] = *reinterpret_cast<const f32x4*>(x + (gi.off + (long long)iy * gi.W + ix) * CP + c4i * 4);
      }
    }
  };
  fetch(tile);
  for (; tile < tile_end; tile++) {
    const int oy0 = (tile / tiles_x) * TH, ox0 = (tile % tiles_x) * TW;
#pragma unroll
    for (int i = 0; i < NPF; i++) {
      const int e = tid + NTHR * i;
      if (e < PH * PW * C4) *reinterpret_cast<f32x4*>(patch + (e / C4) * PPITCH + (e % C4) * 4) = pf[i];
    }
    __syncthreads();  // patch complete; also: every wave is past the previous tile's reads of `at`
    // depthwise: a 3 x ((SL-1)*SW + 3) window of patch vectors feeds SL outputs
    act_dispatch(dw_act, dw_has_lab, false, [&](auto atag, auto ltag, auto) {
      constexpr int A = decltype(atag)::value, L = decltype(ltag)::value;
      if (tid < ITEMS) {
        f32x4 acc[SL];
#pragma unroll
        for (int j = 0; j < SL; j++) acc[j] = dwb;
#pragma unroll
        for (int dy = 0; dy < 3; dy++) {
          constexpr int NVW = (SL - 1) * SW + 3;
          f32x4 v[NVW];
#pragma unroll
          for (int j = 0; j < NVW; j++)
            v[j] = *reinterpret_cast<const f32x4*>(patch + ((ipy * SH + dy) * PW + ipx * SW + j) * PPITCH + ic4 * 4);
#pragma unroll
          for (int dx = 0; dx < 3; dx++)
#pragma unroll
            for (int j = 0; j < SL; j++)
#pragma unroll
              for (int e = 0; e < 4; e++) acc[j][e] = fmaf(v[j * SW + dx][e], dww[dy * 3 + dx][e], acc[j][e]);
        }
#pragma unroll
        for (int j = 0; j < SL; j++) {
          f32x4 o;
#pragma unroll
          for (int e = 0; e < 4; e++) o[e] = (ic4 * 4 + e < C) ? epi_val<A, L>(acc[j][e], dw_act, dw_has_lab, dw_a, dw_c) : 0.f;
          *reinterpret_cast<f32x4*>(at + (((ic4 * 4) / KC) * ROWS + ip0 + j) * LROW + (ic4 * 4) % KC) = o;
        }
      }
    });
    __syncthreads();
    if (tile + 1 < tile_end) fetch(tile + 1);
    f32x4 acc[2][NT];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
      for (int j = 0; j < NT; j++) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kc = 0; kc < NKC; kc++)
      mma_chunk<NT>(at + (kc * ROWS + wm * 32 + r) * LROW, at + (kc * ROWS + wm * 32 + 16 + r) * LROW,
                    wt + (kc * NCOL + wn * NT * 16) * LROW, nt_valid, acc, r, q);
    const int p0 = wm * 32 + r, p1 = p0 + 16;
    const int oya = oy0 + p0 / TW, oxa = ox0 + p0 % TW, oyb = oy0 + p1 / TW, oxb = ox0 + p1 % TW;
    const long long pa = g.off + (long long)oya * g.W + oxa, pb = g.off + (long long)oyb * g.W + oxb;
    epilogue_store<NT>(acc, nt_valid, epi, wn * NT * 16, N, nstore, y + pa * ldy, y + pb * ldy, oya < g.H && oxa < g.W,
                       oyb < g.H && oxb < g.W, nullptr, nullptr, q);
  }
}

int g_lc_wave = getenv("RT_LC_WAVE") ? atoi(getenv("RT_LC_WAVE")) : 3;   // 3 = wave-private LDS form for the stride-1 blocks (nn_lcwave.hip, default); 1 = direct-load form, every shape (A/B); 0 = k_lc_thin
int g_lc_thin = 4;  // 4 = fused thin blocks (default); 2 / 3 = force the 128- / 64-pixel tile; 0 = separate depthwise + GEMM kernels (A/B)
static int lc_thin_code(int sh, int sw, int Cp, int Npad16) {  // instantiated (stride, C_in/4, column tiles) combinations
  const int c4 = Cp / 4, nt = (Npad16 + 31) / 32;
  if (sh == 1 && sw == 1) {
    if (c4 == 4 && nt == 1) return 1;
    if (c4 == 8 && nt == 2) return 2;
    if (c4 == 12 && nt == 2) return 3;
    if (c4 == 16 && nt == 2) return 4;
  } else if (sh == 2 && sw == 2) {  // ((2,1) 64 -> 128 of the rec net measured slower fused: 1.37 vs 1.05 ms)
    if (c4 == 8 && nt == 2) return 6;   // det s3.0: 32 -> 48
    if (c4 == 12 && nt == 3) return 7;  // det s4.0: 48 -> 96
  }
  return 0;
}
bool lc_thin_supported(int K, int sh, int sw, int Cp, int C, int Npad16) {
  return g_lc_thin && K == 3 && Cp == round_up(C, 4) && lc_thin_code(sh, sw, Cp, Npad16) != 0;
}
// (k_lc_lds / k_lc_wave address an image through a 32-bit buffer offset, out-of-range marker 0x80000000: images up to 1 GB)
static bool lc_wave_fits(int maxHo, int maxWo, int sh, int sw, int Cp, int ldy) {
  const long long in_bytes = (long long)(maxHo * sh + 2) * (maxWo * sw + 2) * Cp * 4, out_bytes = (long long)maxHo * maxWo * ldy * 4;
  return in_bytes < (1ll << 30) && out_bytes < (1ll << 30);
}
bool lc_block_supported(int K, int sh, int sw, int Cp, int C, int N, int Npad16, int dw_act, int dw_has_lab, const Epilogue& epi,
                        int maxHo, int maxWo) {
  if (lc_thin_supported(K, sh, sw, Cp, C, Npad16)) return true;
  return g_lc_wave && g_lc_thin == 4 && lc_wave_supported(K, sh, sw, Cp, C, N, Npad16, dw_act, dw_has_lab, epi) &&
         lc_wave_fits(maxHo, maxWo, sh, sw, Cp, chan_pitch(N));
}
void lc_thin(hipStream_t st, int sh, int sw, const float* x, const ImgGeom* gin, const ImgGeom* gout, int n_img, int maxHo,
             int maxWo, int Cp, int C, const float* Wd, const float* bd, int dw_act, int dw_has_lab, float dw_a, float dw_c,
             const float* Wp, int N, int Npad16, float* y, int ldy, const Epilogue& epi) {
  if (n_img <= 0) return;
  if (epi.residual || epi.a_scale) throw RtError(8, "lc_thin: residual / a_scale epilogues are not supported");
  if (g_lc_wave && g_lc_thin == 4 && lc_wave_supported(3, sh, sw, Cp, C, N, Npad16, dw_act, dw_has_lab, epi) && lc_wave_fits(maxHo, maxWo, sh, sw, Cp, ldy)) {
    lc_wave(st, sh, sw, x, gin, gout, n_img, maxHo, maxWo, Cp, C, Wd, bd, dw_act, dw_has_lab, dw_a, dw_c, Wp, N, Npad16, y, ldy, epi);
    return;
  }
  // measured per shape: 64-pixel tiles (more workgroups per CU) win from 48 channels up, 128-pixel tiles below
  const int code = lc_thin_code(sh, sw, Cp, Npad16);
  const int TH = code >= 6 ? 4 : (g_lc_thin == 2 ? 8 : (g_lc_thin == 3 ? 4 : (Cp >= 48 ? 4 : 8)));  // 2 / 3 force a variant (A/B)
  static const int tpb_env = getenv("RT_LCT_TPB") ? atoi(getenv("RT_LCT_TPB")) : 0;
  const int tiles = ((maxWo + 15) / 16) * ((maxHo + TH - 1) / TH), tpb = tpb_env > 0 ? tpb_env : 8;
  dim3 grid((tiles + tpb - 1) / tpb, n_img);
#define RT_LCT_T(CC, NN, TT, S1, S2) RT_LAUNCH((k_lc_thin<CC, NN, TT, S1, S2>), grid, dim3(64 * TT), 0, st, x, gin, gout, C, Wd, bd, dw_act, dw_has_lab, dw_a, dw_c, Wp, N, Npad16, y, ldy, epi, tpb)
#define RT_LCT(CC, NN) do { if (TH == 4) RT_LCT_T(CC, NN, 4, 1, 1); else RT_LCT_T(CC, NN, 8, 1, 1); } while (0)
  switch (code) {
    case 1: RT_LCT(4, 1); break;
    case 2: RT_LCT(8, 2); break;
    case 3: RT_LCT(12, 2); break;
    case 4: RT_LCT(16, 2); break;
    case 6: RT_LCT_T(8, 2, 4, 2, 2); break;
    case 7: RT_LCT_T(12, 3, 4, 2, 2); break;
    default: throw RtError(8, "lc_thin: unsupported shape (check lc_thin_supported)");
  }
#undef RT_LCT
#undef RT_LCT_T
}

int g_gemm_variant = 0;  // 0 = production choice; others are forced by the kernel micro-benchmark
int g_gemm_dma = getenv("RT_GEMM_DMA") ? atoi(getenv("RT_GEMM_DMA")) : 1;   // A/B: 0 keeps the register-staged 256 x 240 tile

// production dispatch (tools/bench_gemm.py): wide tiles once N and M are large; 0 = narrow k_gemm<NT>
static int gemm_dispatch(long long M, int Npad16) {
  if (Npad16 % 240 == 0 && M >= 131072) return 15;  // 256 x 240 tile: halves the weight re-fetch per row
  // (round 5: the 128 x 240 tile only where it fills the chip twice -- 38 k rows x 240 channels, a one-page batch, are 300
  //  workgroups of 768 threads on 256 CUs: two rounds for 1.17 rounds of work; the narrow kernel's 600 workgroups sit four to a CU)
  static const int mid_env = getenv("RT_GEMM_MID") ? atoi(getenv("RT_GEMM_MID")) : 1;   // A/B: 0 = the 128 x 240 tile from 16384 rows
  if (Npad16 % 240 == 0 && M >= 16384 && (!mid_env || (M + 127) / 128 * (Npad16 / 240) >= 512)) return 10;
  // (the 128 x 128 tile, variant 8, lost to the narrow kernel once that prefetched its next K-slab: 192 x 192 at
  // 115200 rows 85 vs 65 TFLOP/s, 128 x 128 at 2.4 M rows 76 vs 72; it remains the squeeze-excite (a_scale) and CTC tile)
  return 0;  // gemm() picks the streaming kernel when K, N <= 64, else the narrow LDS kernel
}
// Profiler label of a pointwise-conv GEMM: family + the kernel symbol the dispatcher picks, so the
// per-kernel numbers of bench.py can be compared with rocprofv3's kernel stats one to one.
const char* gemm_pw_label(long long M, int Npad16, bool a_scale, int se_tile_rows) {
  if (!g_gemm_variant && g_gemm_split && Npad16 % 240 == 0 && M >= 32768 && (!a_scale || se_tile_rows > 0)) return a_scale ? "gemm_pw/k_gemm_split+se" : "gemm_pw/k_gemm_split";   // (K > 96 is what the rec net's layers have)
  if (a_scale && se_tile_rows == 256) return "gemm_pw/k_gemm32p+se";
  if (a_scale && !g_gemm_variant) return gemm_dispatch(M, Npad16) == 0 ? "gemm_pw/k_gemm_wide<2,4,4,2>+se" : "gemm_pw/k_gemm_wide<2,5,4,3>+se";
  switch (g_gemm_variant ? -1 : gemm_dispatch(M, Npad16)) {
    case 15: return g_gemm_dma ? "gemm_pw/k_gemm32p" : "gemm_pw/k_gemm_wide<4,5,4,3>";
    case 10: return "gemm_pw/k_gemm_wide<2,5,4,3>";
    case 0: return "gemm_pw/thin";  // k_gemm_stream (K, N <= 64) or k_gemm<NT>
    default: return "gemm_pw/variant";
  }
}

int g_argmax_wide = 0;  // CTC head: 0 = narrow kernel with 128-column blocks; 2 = 128 x 128 wide tile (0.92 ms); 1 = 256 x 240 tile (0.97 ms)
int gemm_argmax_tiles(int Npad16) { return g_argmax_wide == 1 ? (Npad16 + 239) / 240 : (Npad16 + 127) / 128; }

// one thread per row: fold the column tiles in ascending order -> argmax (first maximum) and softmax(max) = 1 / sum
__global__ __launch_bounds__(256) void k_argmax_merge(const float* __restrict__ pm, const int* __restrict__ pi,
                                                      const float* __restrict__ ps, int tiles, long long rows,
                                                      int* __restrict__ idx, float* __restrict__ prob) {
  const long long row = (long long)blockIdx.x * 256 + threadIdx.x;
  if (row >= rows) return;
  float m = -INFINITY;
  int mi = 0x7fffffff;
  for (int t = 0; t < tiles; t++) {
    const float v = pm[row * tiles + t];
    if (v > m) { m = v; mi = pi[row * tiles + t]; }
  }
  float sum = 0.f;
  for (int t = 0; t < tiles; t++) sum += ps[row * tiles + t] * expf(pm[row * tiles + t] - m);
  idx[row] = mi;
  prob[row] = 1.0f / sum;
}
void argmax_merge(hipStream_t st, const float* pm, const int* pi, const float* ps, int tiles, long long rows, int* idx,
                  float* prob) {
  if (rows <= 0) return;
  RT_LAUNCH(k_argmax_merge, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, st, pm, pi, ps, tiles, rows, idx, prob);
}

// Row-block height of the squeeze-excite (a_scale) form the dispatcher will take: 256 = k_gemm32p (3-int a_tab entries, images
// of >= 128 rows), 128 = the register-staged wide tiles (2-int entries, images >= 128 rows), 0 = no fused form.
int gemm_se_tile_rows(int lda, long long M, int K, int N, int Npad16, int act, long long min_pix) {
  if (g_gemm_variant) return 0;
  Epilogue probe; probe.act = act;
  if (g_gemm_dma && gemm_dispatch(M, Npad16) == 15 && min_pix >= 128 && K <= 512 && act == ACT_HSWISH &&
      gemm_dma_supported(lda, M, K, N, Npad16, probe))
    return 256;
  const int r = gemm_tile_rows(M, Npad16);
  return (r > 0 && min_pix >= r) ? r : 0;
}
int gemm_tile_rows(long long M, int Npad16) {
  if (g_gemm_variant) return 0;
  if (gemm_dispatch(M, Npad16) != 0) return 128;  // (a_scale runs the 256 x 240 shapes on the 128 x 240 tile)
  return (Npad16 >= 128 && M >= 8192) ? 128 : 0;  // a_scale on the 128 x 128 tile
}

void gemm(hipStream_t st, const float* A, int lda, long long M, int K, const float* Wp, int N, int Npad16, float* C,
          int ldc, int coff, const Epilogue& epi) {
  if (M <= 0) return;
  int v = g_gemm_variant;
  if (v == 0) v = gemm_dispatch(M, Npad16);
  if (epi.am_max) {  // CTC head: softmax statistics per column tile instead of the logits (argmax_merge folds them)
    if (epi.am_tiles != gemm_argmax_tiles(Npad16)) throw RtError(8, "gemm: am_tiles must be gemm_argmax_tiles(Npad16)");
    if (g_argmax_wide == 1) {
      dim3 grid((unsigned)((M + 255) / 256), (unsigned)((Npad16 + 239) / 240));
      RT_LAUNCH((k_gemm_wide<4, 5, 4, 3, 0, 0, 0, 1>), grid, dim3(768), 0, st, A, lda, M, K, Wp, N, Npad16, C, ldc, coff, epi);
    } else if (g_argmax_wide == 2) {
      dim3 grid((unsigned)((M + 127) / 128), (unsigned)((Npad16 + 127) / 128));
      RT_LAUNCH((k_gemm_wide<2, 4, 4, 2, 0, 0, 0, 1>), grid, dim3(512), 0, st, A, lda, M, K, Wp, N, Npad16, C, ldc, coff, epi);
    } else {  // narrow kernel, 128-column blocks
      dim3 grid((unsigned)((M + 127) / 128), (unsigned)((Npad16 + 127) / 128));
      static const int bf_env1 = getenv("RT_GEMM_BF") ? atoi(getenv("RT_GEMM_BF")) : 1;
      if (bf_env1 && (long long)128 * lda * 4 < (1ll << 31) && (long long)((K + KC - 1) / KC) * Npad16 * KC * 4 < (1ll << 31))
        RT_LAUNCH((k_gemm<8, 1, true>), grid, dim3(256), 0, st, A, lda, M, K, Wp, N, Npad16, C, ldc, coff, epi);
      else
      RT_LAUNCH((k_gemm<8, 1>), grid, dim3(256), 0, st, A, lda, M, K, Wp, N, Npad16, C, ldc, coff, epi);
    }
    return;
  }
  if ((v == 40 || (!g_gemm_variant && g_gemm_split)) && gemm_split_supported(lda, M, K, N, Npad16, epi)) {
    gemm_split(st, A, lda, M, K, Wp, N, Npad16, C, ldc, coff, epi);
    return;
  }
  if (v == 40) v = 30;
  if (!g_gemm_variant && gemm_w_supported(lda, M, K, N, Npad16, epi, ldc, coff)) { gemm_w(st, A, lda, M, K, Wp, N, Npad16, C, ldc, coff, epi); return; }
  // persistent LDS-DMA form of the 256 x 240 tile (variant 30; production for the large N = 240 / 480 layers)
  if ((v == 30 || (v == 15 && !g_gemm_variant && g_gemm_dma)) && gemm_dma_supported(lda, M, K, N, Npad16, epi)) {
    gemm_dma(st, A, lda, M, K, Wp, N, Npad16, C, ldc, coff, epi);
    return;
  }
  if (v == 30) v = 15;
  if (epi.a_scale && epi.a_tab_stride != 2) throw RtError(8, "gemm: a 3-int a_tab is only understood by k_gemm32p (gemm_se_tile_rows() == 256)");
  if (epi.a_scale) {  // squeeze-excite scale folded into the A staging: wide tiles only (gemm_tile_rows)
    if (K > 512 || !epi.a_tab) throw RtError(8, "gemm: a_scale needs K <= 512 and a row-tile table");
    if (v == 15) v = 10;  // the 256-row tile has no registers to spare for the scaling (spills): 128 x 240 measured faster
    if (v == 0 && Npad16 >= 128) v = 8;
    if (v == 10) {
      dim3 grid((unsigned)((M + 127) / 128), (unsigned)((Npad16 + 239) / 240));
      RT_LAUNCH((k_gemm_wide<2, 5, 4, 3, 0, 0, 1>), grid, dim3(768), 0, st, A, lda, M, K, Wp, N, Npad16, C, ldc, coff, epi);
    } else if (v == 8) {
      dim3 grid((unsigned)((M + 127) / 128), (unsigned)((Npad16 + 127) / 128));
      RT_LAUNCH((k_gemm_wide<2, 4, 4, 2, 0, 0, 1>), grid, dim3(512), 0, st, A, lda, M, K, Wp, N, Npad16, C, ldc, coff, epi);
    } else {
      throw RtError(8, "gemm: a_scale is only implemented for the wide tiles");
    }
    return;
  }
  if (v == 8) {
    dim3 grid((unsigned)((M + 127) / 128), (unsigned)((Npad16 + 127) / 128));
    RT_LAUNCH((k_gemm_wide<2, 4, 4, 2>), grid, dim3(512), 0, st, A, lda, M, K, Wp, N, Npad16, C, ldc, coff, epi);
    return;
  }
  if ((v == 20 && Npad16 <= 128 && K <= 128) || (v == 0 && !g_gemm_variant && Npad16 <= 64 && K <= 64 && M >= 65536)) {  // streaming kernel for the thin layers
    const int ntl = Npad16 / 16, kg = (K + 15) / 16;
    const long long tiles = (M + 31) / 32;
    const unsigned blocks = (unsigned)std::min<long long>((tiles + 3) / 4, 256 * 8);
#define RT_GS(NTV, KGV) RT_LAUNCH((k_gemm_stream<NTV, KGV>), dim3(blocks), dim3(256), 0, st, A, lda, M, K, Wp, N, Npad16, C, ldc, coff, epi)
#define RT_GS_K(NTV) do { if (kg <= 1) RT_GS(NTV, 1); else if (kg <= 2) RT_GS(NTV, 2); else if (kg <= 4) RT_GS(NTV, 4); else if (kg <= 6) RT_GS(NTV, 6); else RT_GS(NTV, 8); } while (0)
    if (ntl <= 1) RT_GS_K(1); else if (ntl <= 2) RT_GS_K(2); else if (ntl <= 3) RT_GS_K(3); else if (ntl <= 4) RT_GS_K(4);
    else if (ntl <= 6) RT_GS_K(6); else RT_GS_K(8);
#undef RT_GS_K
#undef RT_GS
    return;
  }
  if (v == 15) {
    dim3 grid((unsigned)((M + 255) / 256), (unsigned)((Npad16 + 239) / 240));
    RT_LAUNCH((k_gemm_wide<4, 5, 4, 3>), grid, dim3(768), 0, st, A, lda, M, K, Wp, N, Npad16, C, ldc, coff, epi);
    return;
  }
  if (v == 10) {
    dim3 grid((unsigned)((M + 127) / 128), (unsigned)((Npad16 + 239) / 240));
    RT_LAUNCH((k_gemm_wide<2, 5, 4, 3>), grid, dim3(768), 0, st, A, lda, M, K, Wp, N, Npad16, C, ldc, coff, epi);
    return;
  }
  int ntiles = Npad16 / 16;
  int NT = ntiles >= 8 ? 8 : ntiles;
  if (ntiles > 8) {  // pick the split with least padding among 8 / 6 / 5
    int best = 8, waste = round_up(ntiles, 8) - ntiles;
    for (int c : {6, 5, 4}) {
      int w = round_up(ntiles, c) - ntiles;
      if (w < waste) { waste = w; best = c; }
    }
    NT = best;
  }
  // small M (one page, the coarse pyramid levels): fewer columns per workgroup until there is a workgroup per CU -- 3600 x 480 x
  // 480 on 128 x 128 tiles is 116 workgroups walking 15 slabs each, 35 us; the column tiles are independent: same bits
  {
    const long long rbs = (M + 127) / 128;
    static const int occ_env = getenv("RT_GEMM_OCC") ? atoi(getenv("RT_GEMM_OCC")) : 2;   // workgroups per CU to aim for
    const long long want = (long long)stream_cus(st) * occ_env;
    while (NT > 1 && rbs * ((ntiles + NT - 1) / NT) < want) NT = (NT + 1) / 2;
  }
  dim3 grid((unsigned)((M + 127) / 128), (unsigned)((ntiles + NT - 1) / NT));
  // buffer-resource fetch (k_gemm<NT, 0, true>): a tile's 128 rows and the packed weights within the 2-GB offset range
  static const int bf_env = getenv("RT_GEMM_BF") ? atoi(getenv("RT_GEMM_BF")) : 1;
  const bool bf = bf_env && (long long)128 * lda * 4 < (1ll << 31) && (long long)((K + KC - 1) / KC) * Npad16 * KC * 4 < (1ll << 31);
#define RT_GEMM_CASE(n) \
  case n: if (bf) RT_LAUNCH((k_gemm<n, 0, true>), grid, dim3(256), 0, st, A, lda, M, K, Wp, N, Npad16, C, ldc, coff, epi); \
          else RT_LAUNCH(k_gemm<n>, grid, dim3(256), 0, st, A, lda, M, K, Wp, N, Npad16, C, ldc, coff, epi); break;
  switch (NT) {
    RT_GEMM_CASE(1) RT_GEMM_CASE(2) RT_GEMM_CASE(3) RT_GEMM_CASE(4) RT_GEMM_CASE(5) RT_GEMM_CASE(6) RT_GEMM_CASE(7)
    RT_GEMM_CASE(8)
  }
#undef RT_GEMM_CASE
}

// ---------------------------------------------------------------------------
// Spatial dense conv (stride 1, zero "same" padding) as implicit GEMM.
// One block = one TH x TW (=128 pixel) output tile of one image x 16*NT couts.
// The halo tile of a 32-channel slab is staged once in LDS and reused by all taps.
// ---------------------------------------------------------------------------
template <int KH, int KW, int TH, int TW, int NT>
__global__ __launch_bounds__(256) void k_conv_sp(const float* __restrict__ x, int ldx, const ImgGeom* __restrict__ geom,
                                                 int Cin, const float* __restrict__ Wp, int N, int Npad,
                                                 float* __restrict__ y, int ldy, Epilogue epi) {
  constexpr int HH = TH + KH - 1, HW = TW + KW - 1, TAPS = KH * KW;
  __shared__ __attribute__((aligned(16))) float lds[(HH * HW + TAPS * 16 * NT) * LROW];
  float* xs = lds;
  float* ws = lds + HH * HW * LROW;
  const ImgGeom g = geom[blockIdx.y];
  const int tiles_x = (g.W + TW - 1) / TW, tiles_y = (g.H + TH - 1) / TH;
  if ((int)blockIdx.x >= tiles_x * tiles_y) return;
  const int ty = blockIdx.x / tiles_x, tx = blockIdx.x % tiles_x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 15, q = lane >> 4;
  const int n0 = blockIdx.z * 16 * NT;
  const int nt_valid = min(NT, (Npad - n0) / 16);
  f32x4 acc[2][NT];
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int j = 0; j < NT; j++) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  // this lane's two pixels (tile-local)
  const int p0 = wave * 32 + r, p1 = p0 + 16;
  const int py0 = p0 / TW, px0 = p0 % TW, py1 = p1 / TW, px1 = p1 % TW;

  const int nkc = (Cin + KC - 1) / KC;
  // The next 32-channel slab (halo pixels + taps) is requested into registers before the MFMAs of the current one
  // and written to LDS after them: its latency hides behind TAPS x mma_chunk instead of sitting between barriers.
  constexpr int XLD = (HH * HW * 8 + 255) / 256, WLD = (TAPS * 16 * NT * 8 + 255) / 256;
  f32x4 px_[XLD], pw_[WLD];
  auto fetch = [&](int kc) {
    const int k0 = kc * KC;
#pragma unroll
    for (int i = 0; i < XLD; i++) {
      const int idx = tid + 256 * i;
      px_[i] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (idx < HH * HW * 8) {
        int hp = idx >> 3, c4 = idx & 7;
        int hy = hp / HW, hx = hp % HW;
        int gy = ty * TH + hy - KH / 2, gx = tx * TW + hx - KW / 2;
        if (gy >= 0 && gy < g.H && gx >= 0 && gx < g.W && k0 + c4 * 4 < Cin)
          px_[i] = *reinterpret_cast<const f32x4*>(x + (g.off + (long long)gy * g.W + gx) * ldx + k0 + c4 * 4);
      }
    }
#pragma unroll
    for (int i = 0; i < WLD; i++) {
      const int idx = tid + 256 * i;
      pw_[i] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (idx < TAPS * 16 * NT * 8) {
        int row = idx >> 3, c4 = idx & 7;
        int tap = row / (16 * NT), n = row % (16 * NT);
        if (n0 + n < Npad)
          pw_[i] = *reinterpret_cast<const f32x4*>(Wp + (((long long)kc * TAPS + tap) * Npad + n0 + n) * KC + c4 * 4);
      }
    }
  };
  auto stash = [&]() {
#pragma unroll
    for (int i = 0; i < XLD; i++) {
      const int idx = tid + 256 * i;
      if (idx < HH * HW * 8) *reinterpret_cast<f32x4*>(xs + (idx >> 3) * LROW + (idx & 7) * 4) = px_[i];
    }
#pragma unroll
    for (int i = 0; i < WLD; i++) {
      const int idx = tid + 256 * i;
      if (idx < TAPS * 16 * NT * 8) *reinterpret_cast<f32x4*>(ws + (idx >> 3) * LROW + (idx & 7) * 4) = pw_[i];
    }
  };
  fetch(0);
  for (int kc = 0; kc < nkc; kc++) {
    stash();
    __syncthreads();
    if (kc + 1 < nkc) fetch(kc + 1);
#pragma unroll
    for (int dy = 0; dy < KH; dy++)
#pragma unroll
      for (int dx = 0; dx < KW; dx++)
        mma_chunk<NT>(xs + ((py0 + dy) * HW + px0 + dx) * LROW, xs + ((py1 + dy) * HW + px1 + dx) * LROW,
                      ws + (dy * KW + dx) * 16 * NT * LROW, nt_valid, acc, r, q);
    __syncthreads();
  }
  const int oy0 = ty * TH + py0, ox0 = tx * TW + px0, oy1 = ty * TH + py1, ox1 = tx * TW + px1;
  const bool v0 = oy0 < g.H && ox0 < g.W, v1 = oy1 < g.H && ox1 < g.W;
  const long long pa = g.off + (long long)oy0 * g.W + ox0, pb = g.off + (long long)oy1 * g.W + ox1;
  const int nstore = (N + 3) & ~3;
  epilogue_store<NT>(acc, nt_valid, epi, n0, N, nstore, y + pa * ldy, y + pb * ldy, v0, v1,
                     epi.residual ? epi.residual + pa * epi.ld_res : nullptr,
                     epi.residual ? epi.residual + pb * epi.ld_res : nullptr, q);
}

// ---------------------------------------------------------------------------
// 3x3 dense conv onto FEW output channels (N = 4 * NG <= 64, e.g. the RSEFPN / DB-head 96 -> 24 layers) without
// padding the channel dimension to MFMA tiles.  On the 16-wide tiles of k_conv_sp 24 channels run as 32: a quarter of the
// matrix work multiplies zeros, and these layers are MFMA bound (PMC: 82 TFLOP/s of real work = 110 padded).
// v_mfma_f32_4x4x1_16B_f32 computes sixteen independent 4 x 4 outer products per instruction at the full f32 MFMA rate
// (8 cycles, 512 FLOP); with the A-operand broadcast (cbsz = 4, abid = g) every block takes the SAME four weights --
// channel group g of this k -- and its own four pixels: lane l = pixel l of the wave's 64, D[i] = channel 4 g + i.  One
// ds_read_b128 per operand feeds four k steps x NG channel groups = 4 NG MFMAs; a lane ends with all N channels of its
// pixel (N * 4 contiguous bytes per store).  Work: exactly N channels.  Workgroup = 4 waves on a 16 x 16 pixel tile,
// halo tile + nine taps of a 32-channel slab in LDS, next slab prefetched into registers (as k_conv_sp), 2 workgroups / CU.
// ---------------------------------------------------------------------------
// FUSE: the input is the RSEFPN's concat(up8(p5) * s5, up4(p4) * s4, up2(p3) * s3, p2 * s2) gathered on the fly (FpnSrc): the
// 4 x Cq-channel fuse tensor (k_fpn_concat: a 0.7 GB write + read per 32 pages) is never built.  Same values as the
// two-step form's operands (scale multiply first); the K slabs are the four levels (24 channels each) instead of three
// 32-channel slabs, so the sum runs in another order: equal to the pair to fp32 rounding, not bit for bit.
struct FpnSrc {
  const float* p[4];        // p5, p4, p3, p2 (pitch Cq each)
  const ImgGeom* g[4];
  const float* scale[4];    // [image][Cq] or null
  int Cq;
};
template <int NG, int FUSE = 0>
__global__ __launch_bounds__(256, 2) void k_conv3_few(const float* __restrict__ x, int ldx, const ImgGeom* __restrict__ geom,
                                                      int Cin, const float* __restrict__ Wp, int N, int Npad,
                                                      float* __restrict__ y, int ldy, Epilogue epi, FpnSrc fs) {
  constexpr int TH = 16, TW = 16, HH = TH + 2, HW = TW + 2, TAPS = 9, NCH = 4 * NG;
  // channels per K slab, in 4-channel chunks: a 32-channel slab of the input tensor, or (FUSE) one whole FPN level of
  // Cq = 24 channels -- all threads of a slab then gather from the same level (uniform geometry and pointers)
  constexpr int CH4 = FUSE ? 6 : KC / 4;
  __shared__ __attribute__((aligned(16))) float lds[(HH * HW + TAPS * NCH) * LROW];
  float* xs = lds;
  float* ws = lds + HH * HW * LROW;
  const ImgGeom g = geom[blockIdx.y];
  const int tiles_x = (g.W + TW - 1) / TW, tiles_y = (g.H + TH - 1) / TH;
  if ((int)blockIdx.x >= tiles_x * tiles_y) return;
  const int ty = blockIdx.x / tiles_x, tx = blockIdx.x % tiles_x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int py = wave * 4 + (lane >> 4), px = lane & 15;   // this lane's pixel (tile-local)
  f32x4 acc[NG];
#pragma unroll
  for (int i = 0; i < NG; i++) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int nkc = FUSE ? 4 : (Cin + KC - 1) / KC;
  constexpr int XLD = (HH * HW * CH4 + 255) / 256, WLD = (TAPS * NCH * CH4 + 255) / 256;
  f32x4 px_[XLD], pw_[WLD];
  auto fetch = [&](int kc) {
    const int k0 = kc * CH4 * 4;
    // FUSE: level kc (p5, p4, p3, p2): nearest-neighbour source pixel (gy >> sh, gx >> sh), times the level's SE scale
    const int sh = 3 - kc;
    const ImgGeom G = FUSE ? fs.g[kc][blockIdx.y] : g;
    const float* src = FUSE ? fs.p[kc] : x;
    const float* scl = FUSE ? fs.scale[kc] : nullptr;
#pragma unroll
    for (int i = 0; i < XLD; i++) {
      const int idx = tid + 256 * i;
      px_[i] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (idx < HH * HW * CH4) {
        const int hp = idx / CH4, c4 = idx - hp * CH4;
        const int hy = hp / HW, hx = hp % HW;
        const int gy = ty * TH + hy - 1, gx = tx * TW + hx - 1;
        if (gy >= 0 && gy < g.H && gx >= 0 && gx < g.W && (FUSE || k0 + c4 * 4 < Cin)) {
          if (FUSE) {
            const int sy = min(gy >> sh, G.H - 1), sx = min(gx >> sh, G.W - 1);
            f32x4 v = *reinterpret_cast<const f32x4*>(src + (G.off + (long long)sy * G.W + sx) * fs.Cq + c4 * 4);
            if (scl) v *= *reinterpret_cast<const f32x4*>(scl + (long long)blockIdx.y * fs.Cq + c4 * 4);
            px_[i] = v;
          } else {
            px_[i] = *reinterpret_cast<const f32x4*>(x + (g.off + (long long)gy * g.W + gx) * ldx + k0 + c4 * 4);
          }
        }
      }
    }
#pragma unroll
    for (int i = 0; i < WLD; i++) {
      const int idx = tid + 256 * i;
      pw_[i] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (idx < TAPS * NCH * CH4) {
        const int row = idx / CH4, c4 = idx - row * CH4;
        const int tap = row / NCH, n = row % NCH;
        const int ch = k0 + c4 * 4;   // input channel of the chunk; the packed weights are in 32-channel slabs
        if (n < Npad) pw_[i] = *reinterpret_cast<const f32x4*>(Wp + (((long long)(ch >> 5) * TAPS + tap) * Npad + n) * KC + (ch & 31));
      }
    }
  };
  auto stash = [&]() {
#pragma unroll
    for (int i = 0; i < XLD; i++) {
      const int idx = tid + 256 * i;
      if (idx < HH * HW * CH4) *reinterpret_cast<f32x4*>(xs + (idx / CH4) * LROW + (idx % CH4) * 4) = px_[i];
    }
#pragma unroll
    for (int i = 0; i < WLD; i++) {
      const int idx = tid + 256 * i;
      if (idx < TAPS * NCH * CH4) *reinterpret_cast<f32x4*>(ws + (idx / CH4) * LROW + (idx % CH4) * 4) = pw_[i];
    }
  };
  const float* wrow = ws + (lane < NCH ? lane : 0) * LROW;   // A operand: lane 4 g + i holds channel 4 g + i (lanes >= N are never selected)
  fetch(0);
  for (int kc = 0; kc < nkc; kc++) {
    stash();
    __syncthreads();
    if (kc + 1 < nkc) fetch(kc + 1);
#pragma unroll
    for (int dy = 0; dy < 3; dy++)
#pragma unroll
      for (int dx = 0; dx < 3; dx++) {
        const float* xr = xs + ((py + dy) * HW + px + dx) * LROW;
        const float* wr = wrow + (dy * 3 + dx) * NCH * LROW;
#pragma unroll
        for (int kk = 0; kk < CH4; kk++) {
          const f32x4 b = *reinterpret_cast<const f32x4*>(xr + kk * 4);
          const f32x4 a = *reinterpret_cast<const f32x4*>(wr + kk * 4);
#pragma unroll
          for (int s2 = 0; s2 < 4; s2++) {
            // (abid must be an immediate: one call per channel group)
            if (NG > 0) acc[0] = __builtin_amdgcn_mfma_f32_4x4x1f32(a[s2], b[s2], acc[0], 4, 0, 0);
            if (NG > 1) acc[NG > 1 ? 1 : 0] = __builtin_amdgcn_mfma_f32_4x4x1f32(a[s2], b[s2], acc[NG > 1 ? 1 : 0], 4, 1, 0);
            if (NG > 2) acc[NG > 2 ? 2 : 0] = __builtin_amdgcn_mfma_f32_4x4x1f32(a[s2], b[s2], acc[NG > 2 ? 2 : 0], 4, 2, 0);
            if (NG > 3) acc[NG > 3 ? 3 : 0] = __builtin_amdgcn_mfma_f32_4x4x1f32(a[s2], b[s2], acc[NG > 3 ? 3 : 0], 4, 3, 0);
            if (NG > 4) acc[NG > 4 ? 4 : 0] = __builtin_amdgcn_mfma_f32_4x4x1f32(a[s2], b[s2], acc[NG > 4 ? 4 : 0], 4, 4, 0);
            if (NG > 5) acc[NG > 5 ? 5 : 0] = __builtin_amdgcn_mfma_f32_4x4x1f32(a[s2], b[s2], acc[NG > 5 ? 5 : 0], 4, 5, 0);
            if (NG > 6) acc[NG > 6 ? 6 : 0] = __builtin_amdgcn_mfma_f32_4x4x1f32(a[s2], b[s2], acc[NG > 6 ? 6 : 0], 4, 6, 0);
            if (NG > 7) acc[NG > 7 ? 7 : 0] = __builtin_amdgcn_mfma_f32_4x4x1f32(a[s2], b[s2], acc[NG > 7 ? 7 : 0], 4, 7, 0);
          }
        }
      }
    __syncthreads();
  }
  const int oy = ty * TH + py, ox = tx * TW + px;
  if (oy >= g.H || ox >= g.W) return;
  float* yr = y + (g.off + (long long)oy * g.W + ox) * ldy;
  act_dispatch(epi.act, epi.has_lab, false, [&](auto at, auto lt, auto) {
    constexpr int A = decltype(at)::value, L = decltype(lt)::value;
#pragma unroll
    for (int gi = 0; gi < NG; gi++) {
      f32x4 bias = {0.f, 0.f, 0.f, 0.f};
      if (epi.bias) bias = *reinterpret_cast<const f32x4*>(epi.bias + gi * 4);
      f32x4 o;
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const float t = epi_val<A, L>(acc[gi][j] + bias[j], epi.act, epi.has_lab, epi.lab_a, epi.lab_c);
        o[j] = (gi * 4 + j < N) ? t : 0.0f;
      }
      *reinterpret_cast<f32x4*>(yr + gi * 4) = o;
    }
  });
}

static const int g_conv3_few = getenv("RT_CONV3_FEW") ? atoi(getenv("RT_CONV3_FEW")) : 1;   // A/B: 0 keeps the 16-wide tiles for every N

// DB head's first 3x3 conv (4 * Cq -> N) straight from the four FPN levels: conv_sp over fpn_concat's result without building it.
bool conv3_fpn_fused_supported(int Cq, int N) { return g_conv3_few && Cq == 24 && N == 24 && getenv("RT_NO_FPN_FUSE") == nullptr; }
void conv3_fpn_fused(hipStream_t st, const float* p5, const float* p4, const float* p3, const float* p2, const ImgGeom* g5,
                     const ImgGeom* g4, const ImgGeom* g3, const ImgGeom* g2, int n_img, int maxH, int maxW, int Cq,
                     const float* const* scales, const float* Wp, int N, int Npad16, float* y, int ldy, const Epilogue& epi) {
  if (n_img <= 0) return;
  FpnSrc fs;
  fs.p[0] = p5; fs.p[1] = p4; fs.p[2] = p3; fs.p[3] = p2;
  fs.g[0] = g5; fs.g[1] = g4; fs.g[2] = g3; fs.g[3] = g2;
  for (int i = 0; i < 4; i++) fs.scale[i] = scales ? scales[i] : nullptr;
  fs.Cq = Cq;
  dim3 gridf(((maxW + 15) / 16) * ((maxH + 15) / 16), n_img);
  RT_LAUNCH((k_conv3_few<6, 1>), gridf, dim3(256), 0, st, nullptr, 4 * Cq, g2, 4 * Cq, Wp, N, Npad16, y, ldy, epi, fs);
}

// ---------------------------------------------------------------------------
// k_conv13_flat (round 5): the SVTR neck's 1x3 convs (480 / 960 -> 60 channels) over the FLAT token list.
// The token sequences of a launch's text lines are contiguous in memory ([line 0 | line 1 | ...], one row per token), so a 1x3
// conv over all of them is one conv over the flat list in which a tap must not cross a line boundary.  k_conv_sp gives every
// line its own 128-token tile: lines of ~50 tokens leave 61 % of every tile's MFMAs on pixels that do not exist (PMC: matrix pipe
// 0.76 busy at 0.20 / 0.41 ms per launch for 8.8 / 17.7 GFLOP), and five ways of shrinking the tiles did not help (DESIGN.md
// 5.4).  Here a tile is 128 CONSECUTIVE tokens whatever lines they belong to; `flags` (one byte per token: 1 = first of its
// line, 2 = last) zero the left / right tap's operand where the neighbour belongs to another line.  Halo rows are contiguous
// (130 rows of the activation matrix), every tile but the last is full.  Same slab / tap / (q, s) -> k order per accumulator as
// k_conv_sp and exact zeros for the masked taps: bit-identical.
// ---------------------------------------------------------------------------
template <int NT>
__global__ __launch_bounds__(256) void k_conv13_flat(const float* __restrict__ x, int ldx, long long rows, const unsigned char* __restrict__ flags,
                                                     int Cin, const float* __restrict__ Wp, int N, int Npad, float* __restrict__ y, int ldy,
                                                     Epilogue epi) {
  constexpr int TW = 128, HW = TW + 2;
  __shared__ __attribute__((aligned(16))) float lds[(HW + 3 * 16 * NT) * LROW];
  float* xs = lds;
  float* ws = lds + HW * LROW;
  const long long P0 = (long long)blockIdx.x * TW;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 15, q = lane >> 4;
  const int n0 = blockIdx.y * 16 * NT;
  const int nt_valid = min(NT, (Npad - n0) / 16);
  f32x4 acc[2][NT];
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int j = 0; j < NT; j++) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int p0 = wave * 32 + r, p1 = p0 + 16;
  const long long t0 = P0 + p0, t1 = P0 + p1;
  const unsigned f0 = t0 < rows ? flags[t0] : 3u, f1 = t1 < rows ? flags[t1] : 3u;
  const int nkc = (Cin + KC - 1) / KC;
  constexpr int XLD = (HW * 8 + 255) / 256, WLD = (3 * 16 * NT * 8 + 255) / 256;
  f32x4 px_[XLD], pw_[WLD];
  auto fetch = [&](int kc) {
    const int k0 = kc * KC;
#pragma unroll
    for (int i = 0; i < XLD; i++) {
      const int idx = tid + 256 * i;
      px_[i] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (idx < HW * 8) {
        const int hx = idx >> 3, c4 = idx & 7;
        const long long tok = P0 - 1 + hx;
        if (tok >= 0 && tok < rows && k0 + c4 * 4 < Cin) px_[i] = *reinterpret_cast<const f32x4*>(x + tok * ldx + k0 + c4 * 4);
      }
    }
#pragma unroll
    for (int i = 0; i < WLD; i++) {
      const int idx = tid + 256 * i;
      pw_[i] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (idx < 3 * 16 * NT * 8) {
        const int row = idx >> 3, c4 = idx & 7;
        const int tap = row / (16 * NT), n = row - tap * (16 * NT);
        if (n0 + n < Npad) pw_[i] = *reinterpret_cast<const f32x4*>(Wp + (((long long)kc * 3 + tap) * Npad + n0 + n) * KC + c4 * 4);
      }
    }
  };
  fetch(0);
  for (int kc = 0; kc < nkc; kc++) {
#pragma unroll
    for (int i = 0; i < XLD; i++) {
      const int idx = tid + 256 * i;
      if (idx < HW * 8) *reinterpret_cast<f32x4*>(xs + (idx >> 3) * LROW + (idx & 7) * 4) = px_[i];
    }
#pragma unroll
    for (int i = 0; i < WLD; i++) {
      const int idx = tid + 256 * i;
      if (idx < 3 * 16 * NT * 8) *reinterpret_cast<f32x4*>(ws + (idx >> 3) * LROW + (idx & 7) * 4) = pw_[i];
    }
    __syncthreads();
    if (kc + 1 < nkc) fetch(kc + 1);
#pragma unroll
    for (int dx = 0; dx < 3; dx++) {
      // tap dx reads halo row p + dx (= token p + dx - 1); the left tap of a line's first token and the right tap of its last are zero
      const bool z0 = (dx == 0 && (f0 & 1u)) || (dx == 2 && (f0 & 2u)), z1 = (dx == 0 && (f1 & 1u)) || (dx == 2 && (f1 & 2u));
      const float* xr0 = xs + (p0 + dx) * LROW;
      const float* xr1 = xs + (p1 + dx) * LROW;
      const float* wt = ws + dx * 16 * NT * LROW;
#pragma unroll
      for (int g = 0; g < KC / 16; g++) {
        f32x4 a0 = *reinterpret_cast<const f32x4*>(xr0 + g * 16 + 4 * q);
        f32x4 a1 = *reinterpret_cast<const f32x4*>(xr1 + g * 16 + 4 * q);
        if (dx != 1) { if (z0) a0 = f32x4{0.f, 0.f, 0.f, 0.f}; if (z1) a1 = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
        for (int nt = 0; nt < NT; nt++) {
          if (nt < nt_valid) {
            const f32x4 b = *reinterpret_cast<const f32x4*>(wt + (nt * 16 + r) * LROW + g * 16 + 4 * q);
#pragma unroll
            for (int s4 = 0; s4 < 4; s4++) {
              acc[0][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[s4], a0[s4], acc[0][nt], 0, 0, 0);
              acc[1][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[s4], a1[s4], acc[1][nt], 0, 0, 0);
            }
          }
        }
      }
    }
    __syncthreads();
  }
  const int nstore = (N + 3) & ~3;
  epilogue_store<NT>(acc, nt_valid, epi, n0, N, nstore, y + t0 * ldy, y + t1 * ldy, t0 < rows, t1 < rows,
                     epi.residual ? epi.residual + t0 * epi.ld_res : nullptr, epi.residual ? epi.residual + t1 * epi.ld_res : nullptr, q);
}
int g_conv13_flat = getenv("RT_CONV13_FLAT") ? atoi(getenv("RT_CONV13_FLAT")) : 1;   // A/B: 0 = k_conv_sp<1,3,1,128,NT> (a tile per line)
bool conv13_flat_supported(int N, int Npad16) { return g_conv13_flat && Npad16 <= 64 && N > 0; }
void conv13_flat(hipStream_t st, const float* x, int ldx, long long rows, const unsigned char* flags, int Cin, const float* Wp, int N,
                 int Npad16, float* y, int ldy, const Epilogue& epi) {
  if (rows <= 0) return;
  const int ntiles = Npad16 / 16;
  int NT = ntiles >= 4 ? 4 : ntiles;
  // few tokens (one page: 13 tiles of 128 tokens walking 45 slabs each, 96 us): a column tile per workgroup until there is a
  // workgroup per CU -- the column tiles are independent: same bits (as the narrow GEMM's launch rule)
  const int cus13 = stream_cus(st);
  while (NT > 1 && (rows + 127) / 128 * ((ntiles + NT - 1) / NT) < cus13) NT = (NT + 1) / 2;
  dim3 grid((unsigned)((rows + 127) / 128), (unsigned)((ntiles + NT - 1) / NT));
  switch (NT) {
    case 1: RT_LAUNCH((k_conv13_flat<1>), grid, dim3(256), 0, st, x, ldx, rows, flags, Cin, Wp, N, Npad16, y, ldy, epi); break;
    case 2: RT_LAUNCH((k_conv13_flat<2>), grid, dim3(256), 0, st, x, ldx, rows, flags, Cin, Wp, N, Npad16, y, ldy, epi); break;
    case 3: RT_LAUNCH((k_conv13_flat<3>), grid, dim3(256), 0, st, x, ldx, rows, flags, Cin, Wp, N, Npad16, y, ldy, epi); break;
    default: RT_LAUNCH((k_conv13_flat<4>), grid, dim3(256), 0, st, x, ldx, rows, flags, Cin, Wp, N, Npad16, y, ldy, epi); break;
  }
}

void conv_sp(hipStream_t st, int KH, int KW, const float* x, int ldx, const ImgGeom* geom, int n_img, int maxH,
             int maxW, int Cin, const float* Wp, int N, int Npad16, float* y, int ldy, const Epilogue& epi) {
  if (n_img <= 0) return;
  int ntiles = Npad16 / 16;
  // few output channels that do not fill 16-wide tiles (N = 24: a quarter of k_conv_sp's MFMA work would be padding)
  if (KH == 3 && KW == 3 && g_conv3_few && N % 16 != 0 && N <= 32 && !epi.residual && ldy >= round_up(N, 4)) {
    dim3 gridf(((maxW + 15) / 16) * ((maxH + 15) / 16), n_img);
    switch ((N + 3) / 4) {
#define RT_C3F(n) case n: RT_LAUNCH((k_conv3_few<n>), gridf, dim3(256), 0, st, x, ldx, geom, Cin, Wp, N, Npad16, y, ldy, epi, FpnSrc{}); return;
      RT_C3F(1) RT_C3F(2) RT_C3F(3) RT_C3F(5) RT_C3F(6)
#undef RT_C3F
      default: break;
    }
  }
  if (KH == 3 && KW == 3) {
    int NT = ntiles >= 2 ? 2 : 1;
    dim3 grid(((maxW + 15) / 16) * ((maxH + 7) / 8), n_img, (ntiles + NT - 1) / NT);
    if (NT == 2)
      RT_LAUNCH((k_conv_sp<3, 3, 8, 16, 2>), grid, dim3(256), 0, st, x, ldx, geom, Cin, Wp, N, Npad16, y, ldy,
                         epi);
    else
      RT_LAUNCH((k_conv_sp<3, 3, 8, 16, 1>), grid, dim3(256), 0, st, x, ldx, geom, Cin, Wp, N, Npad16, y, ldy,
                         epi);
  } else if (KH == 1 && KW == 3) {
    int NT = ntiles >= 4 ? 4 : ntiles;
    dim3 grid(((maxW + 127) / 128) * maxH, n_img, (ntiles + NT - 1) / NT);
#define RT_C13(n)                                                                                                   \
  case n:                                                                                                           \
    RT_LAUNCH((k_conv_sp<1, 3, 1, 128, n>), grid, dim3(256), 0, st, x, ldx, geom, Cin, Wp, N, Npad16, y, ldy, \
                       epi);                                                                                        \
    break;
    switch (NT) { RT_C13(1) RT_C13(2) RT_C13(3) RT_C13(4) }
#undef RT_C13
  } else {
    throw RtError(8, "conv_sp: unsupported kernel size");
  }
}

// ---------------------------------------------------------------------------
// Row-streaming depthwise conv, strides (SH, SW) in {1,2}: a thread owns an R-row x 4-pixel output
// patch of 4 channels and streams the (R-1)*SH+K input rows through registers once (K=5, R=4,
// stride 1: 4 16-byte loads per output instead of 25), with the KxK weights of the block's
// 32-channel slab in LDS (3.2 KB).  Strips are numbered column-major so that the vertically
// adjacent strips (which share K-SH halo rows) sit in the same workgroup and meet in L1/L2
// instead of re-reading HBM.  Accumulation order per output: bias, then taps in (dy, dx) order.
// 128 VGPRs (4 waves/SIMD): +15 % over 3 waves at C = 480.  Measured and rejected: lane groups
// spanning whole pixels with all the weights in LDS (1.2-1.5x slower), one-row-ahead register
// prefetch (hipcc hoists every load: spills).  On short maps (<= 24 rows, no pooling) the 5x5 stride-1 layers run on
// k_dwconv_sweep below instead (every input row fetched once).
__device__ int g_dw_xcd_dev = 1;  // XCD-aware block order of k_dwconv_rows (A/B: set_dw_xcd)
template <int K, int R, int SH, int SW, int POOL, int LP = 8>  // LP lanes (16 bytes each) side by side on a pixel: 32- or 64-channel slabs
__global__ __launch_bounds__(256, 4) void k_dwconv_rows(const float* __restrict__ x, const ImgGeom* __restrict__ gin,
                                                     const ImgGeom* __restrict__ gout, int Cp, int C,
                                                     const float* __restrict__ Wd, const float* __restrict__ bias, int act,
                                                     int has_lab, float lab_a, float lab_c, float* __restrict__ y,
                                                     float* __restrict__ pool) {
  constexpr int NV = 3 * SW + K;       // input columns feeding 4 output pixels
  constexpr int NI = (R - 1) * SH + K; // input rows feeding R output rows
  constexpr int SPB = 256 / LP;  // strips per block
  __shared__ __attribute__((aligned(16))) float wl[K * K * LP * 4];
  // XCD-aware block order: workgroups are dealt round-robin to the 8 XCDs (own L2 each), so neighbouring block
  // ids -- which share halo columns / rows -- would sit on different L2s and each fetch the overlap.  Block id
  // (xcd, seq) is mapped to work item xcd * (total / 8) + seq: every XCD walks one contiguous eighth of the
  // (strip block, image, slab) space, neighbours meet in the same L2 shortly after each other.
  unsigned bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
  if (g_dw_xcd_dev) {
    // (within one channel slab: an XCD per slab -- the same remap over all three grid dimensions -- measured
    // 1.4x slower at C = 256)
    const unsigned gx = gridDim.x, gy = gridDim.y, total = gx * gy, per = total >> 3;
    const unsigned lin = by * gx + bx, xcd = (lin + bz * total) & 7;
    if (lin < per * 8 && (total & 7) == 0) {
      const unsigned w = xcd * per + (lin >> 3);
      bx = w % gx; by = w / gx;
    }
  }
  const ImgGeom gi = gin[by], go = gout[by];
  const int strips_x = (go.W + 3) >> 2, strips_y = (go.H + R - 1) / R;
  // (Measured and rejected: channel slab as the fastest block coordinate -- 1.4x slower at C = 256.)
  const int nbx = gridDim.x;
  if ((long long)bx * SPB >= (long long)strips_x * strips_y) return;
  const int cbase = bz * LP * 4;
  const int tid = threadIdx.x;
  for (int i = tid; i < K * K * LP; i += 256) {
    int t = i / LP, cc = i % LP;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (cbase + cc * 4 < Cp) v = *reinterpret_cast<const f32x4*>(Wd + t * Cp + cbase + cc * 4);
    *reinterpret_cast<f32x4*>(wl + i * 4) = v;
  }
  __syncthreads();
  const int c4 = tid % LP, ch = cbase + c4 * 4;
  const long long strip = (long long)bx * SPB + tid / LP;
  const bool active = ch < Cp && strip < (long long)strips_x * strips_y;
  f32x4 psum;  // squeeze-excite pooling: this thread's share of the channel sums (summed only in the store loop)
  if (active) {
  const int oy0 = (int)(strip % strips_y) * R, ox0 = (int)(strip / strips_y) * 4;  // column-major
  const f32x4 b = *reinterpret_cast<const f32x4*>(bias + ch);
  f32x4 acc[R][4];
#pragma unroll
  for (int r = 0; r < R; r++)
#pragma unroll
    for (int j = 0; j < 4; j++) acc[r][j] = b;
  // (A branch-free form of this loop lets hipcc hoist every load: 256 VGPRs or spills, 2-10x
  // slower.  The per-row branches keep one input row in flight per wave.)
#pragma unroll
  for (int i = 0; i < NI; i++) {
    const int iy = oy0 * SH + i - K / 2;
    if (iy < 0 || iy >= gi.H) continue;
    const float* row = x + (gi.off + (long long)iy * gi.W) * Cp + ch;
    f32x4 v[NV];
#pragma unroll
    for (int j = 0; j < NV; j++) {
      int ix = ox0 * SW + j - K / 2;
      v[j] = (ix >= 0 && ix < gi.W) ? *reinterpret_cast<const f32x4*>(row + (long long)ix * Cp) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int r = 0; r < R; r++) {
      const int dy = i - r * SH;  // input row i feeds output row r through tap row dy
      if (dy < 0 || dy >= K) continue;
#pragma unroll
      for (int dx = 0; dx < K; dx++) {
        const f32x4 w = *reinterpret_cast<const f32x4*>(wl + ((dy * K + dx) * LP + c4) * 4);
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
          for (int e = 0; e < 4; e++) acc[r][j][e] = fmaf(v[j * SW + dx][e], w[e], acc[r][j][e]);
      }
    }
  }
  f32x4 ps = {0.f, 0.f, 0.f, 0.f};
  act_dispatch(act, has_lab, false, [&](auto at, auto lt, auto) {
    constexpr int A = decltype(at)::value, L = decltype(lt)::value;
#pragma unroll
    for (int r = 0; r < R; r++) {
      const int oy = oy0 + r;
      if (oy >= go.H) break;
#pragma unroll
      for (int j = 0; j < 4; j++) {
        if (ox0 + j >= go.W) break;
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; e++) {
          const float t = epi_val<A, L>(acc[r][j][e], act, has_lab, lab_a, lab_c);
          o[e] = (ch + e < C) ? t : 0.f;  // pitch padding (chan_pitch) holds zeros whatever the input padding held
        }
        *reinterpret_cast<f32x4*>(y + (go.off + (long long)oy * go.W + ox0 + j) * Cp + ch) = o;
        if (POOL) ps += o;
      }
    }
  });
  psum = ps;
  } else {
    psum = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  if (POOL) {  // fixed-order reduction: the strips of a wave by shuffles, 4 waves through LDS -> pool[image][block][channel]
    __shared__ __attribute__((aligned(16))) float red[4 * LP * 4];
#pragma unroll
    for (int d = LP; d < 64; d <<= 1)
#pragma unroll
      for (int e = 0; e < 4; e++) psum[e] += __shfl_xor(psum[e], d);
    if ((tid & 63) < LP) *reinterpret_cast<f32x4*>(red + ((tid >> 6) * LP + (tid & (LP - 1))) * 4) = psum;
    __syncthreads();
    if (tid < LP && cbase + tid * 4 < Cp) {
      f32x4 t = *reinterpret_cast<const f32x4*>(red + tid * 4);
#pragma unroll
      for (int w = 1; w < 4; w++) t += *reinterpret_cast<const f32x4*>(red + (w * LP + tid) * 4);
      *reinterpret_cast<f32x4*>(pool + ((long long)by * nbx + bx) * Cp + cbase + tid * 4) = t;
    }
  }
}

// ---------------------------------------------------------------------------
// Column-sweep depthwise conv (stride 1) for short, wide maps (the recognition net's 12- / 6-row stages): a thread owns a
// PX-pixel x 4-channel column of the WHOLE image and streams the input rows top to bottom once, holding the K output rows
// an input row feeds in a rotating window of accumulators (slot = output row % K, compile time: rows walked in groups of
// K); an output row is stored and its slot restarted from the bias as soon as its last input row has gone through.
// k_dwconv_rows re-reads the K - 1 rows that vertically adjacent strips share -- 20 row reads for 12 rows, 1.63x the output
// bytes at the fabric (PMC), which is what bounds it (4.1 TB/s algorithmic = 6.7 TB/s of fetches).  Same accumulation
// order per output (bias, taps in (dy, dx) order): bit-identical.  Measured on the 1.23 M-pixel 256-channel maps (k_dwconv_rows
// 0.621 ms): PX = 4 at 3 waves / SIMD 0.543 ms (4.64 TB/s), PX = 2 at 4 waves / SIMD 0.567; a first form that loaded the
// columns one tap column ahead (dx-outer loop) ran 0.674 -- the eight loads of a row must be in flight together.
// Round 6: buffer-descriptor addressing (below) 0.533 -> 0.519 ms; PX = 4 at 4 waves / SIMD still spills (300 bytes of scratch).
// ---------------------------------------------------------------------------
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
template <int K, int LP, int PX, int WAVES>
__global__ __launch_bounds__(256, WAVES) void k_dwconv_sweep(const float* __restrict__ x, const ImgGeom* __restrict__ gin,
                                                           const ImgGeom* __restrict__ gout, int Cp, int C,
                                                           const float* __restrict__ Wd, const float* __restrict__ bias, int act,
                                                           int has_lab, float lab_a, float lab_c, float* __restrict__ y) {
  constexpr int NV = PX - 1 + K, P = K / 2, SPB = 256 / LP;
  __shared__ __attribute__((aligned(16))) float wl[K * K * LP * 4];
  const ImgGeom gi = gin[blockIdx.y], go = gout[blockIdx.y];
  const int strips_x = (go.W + PX - 1) / PX;
  if ((int)blockIdx.x * SPB >= strips_x) return;
  const int cbase = blockIdx.z * LP * 4;
  const int tid = threadIdx.x;
  for (int i = tid; i < K * K * LP; i += 256) {
    int t = i / LP, cc = i % LP;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (cbase + cc * 4 < Cp) v = *reinterpret_cast<const f32x4*>(Wd + t * Cp + cbase + cc * 4);
    *reinterpret_cast<f32x4*>(wl + i * 4) = v;
  }
  __syncthreads();
  const int c4 = tid % LP, ch = cbase + c4 * 4;
  const int strip = (int)blockIdx.x * SPB + tid / LP;
  if (ch >= Cp || strip >= strips_x) return;
  const int ox0 = strip * PX, H = gi.H;   // (stride 1: output and input maps have the same size)
  const f32x4 b = *reinterpret_cast<const f32x4*>(bias + ch);
  f32x4 acc[K][PX];
#pragma unroll
  for (int r = 0; r < K; r++)
#pragma unroll
    for (int j = 0; j < PX; j++) acc[r][j] = b;
  // Addresses: one buffer descriptor per ROW (its base is uniform: scalar registers; num_records = the row's bytes) + a 32-bit
  // per-lane byte offset that does not depend on the row.  Columns right of the map are out of the descriptor's range -- loads
  // return zeros, stores are dropped -- and columns left of it carry the out-of-range mark: no per-load test or branch is left.
  // (The first form kept 64-bit per-lane pointers and laundered the running column pointer through an asm operand to keep hipcc
  // from precomputing and spilling the 64-bit column addresses of every row; a pointer that went through an asm operand loses its
  // address space, so every load was a flat_load -- counted in lgkmcnt as well, which made the waits for the taps' LDS reads wait
  // for the row in flight.)
  unsigned loff[NV];   // column ox0 - P + j, this lane's 4 channels
#pragma unroll
  for (int j = 0; j < NV; j++) loff[j] = ox0 - P + j >= 0 ? (unsigned)((ox0 - P + j) * Cp + ch) * 4u : 0x80000000u;
  auto uni_ptr = [](const float* p_) {   // (the image geometry comes from a load indexed by blockIdx: made scalar explicitly)
    const unsigned long long v = (unsigned long long)p_;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return (float*)(((unsigned long long)hi << 32) | lo);
  };
  const float* xim = uni_ptr(x + gi.off * Cp);
  float* yim = uni_ptr(y + go.off * Cp);
  const unsigned row_bytes = (unsigned)__builtin_amdgcn_readfirstlane(gi.W * Cp * 4);
  act_dispatch(act, has_lab, false, [&](auto at, auto lt, auto) {
    constexpr int A = decltype(at)::value, L = decltype(lt)::value;
    auto store_row = [&](int oy, f32x4 (&a)[PX]) __attribute__((always_inline)) {
      const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(yim + (long long)oy * go.W * Cp, 0, row_bytes, 0x00020000);
#pragma unroll
      for (int j = 0; j < PX; j++) {
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; e++) {
          const float t = epi_val<A, L>(a[j][e], act, has_lab, lab_a, lab_c);
          o[e] = (ch + e < C) ? t : 0.f;  // pitch padding (chan_pitch) holds zeros whatever the input padding held
        }
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), yrs, loff[j + P], 0, 0);
        a[j] = b;
      }
    };
    for (int i0 = 0; i0 < H; i0 += K) {
#pragma unroll
      for (int u = 0; u < K; u++) {
        const int i = i0 + u;   // input row; feeds output rows i - P .. i + P through tap rows dy = K - 1 .. 0
        if (i < H) {
          const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xim) + (long long)i * gi.W * Cp, 0, row_bytes, 0x00020000);
          f32x4 v[NV];
#pragma unroll
          for (int j = 0; j < NV; j++) v[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs, loff[j], 0, 0));
#pragma unroll
          for (int dy = 0; dy < K; dy++) {
            const int r = i + P - dy;                  // output row fed through tap row dy
            const int slot = (u + P - dy + K) % K;     // == r mod K (i0 is a multiple of K): compile time
            if (r >= 0 && r < H) {
#pragma unroll
              for (int dx = 0; dx < K; dx++) {
                const f32x4 w = *reinterpret_cast<const f32x4*>(wl + ((dy * K + dx) * LP + c4) * 4);
#pragma unroll
                for (int j = 0; j < PX; j++)
#pragma unroll
                  for (int e = 0; e < 4; e++) acc[slot][j][e] = fmaf(v[j + dx][e], w[e], acc[slot][j][e]);
              }
            }
          }
          if (i - P >= 0) store_row(i - P, acc[(u - P + K) % K]);   // output row i - P has seen its last input row
        }
      }
    }
#pragma unroll
    for (int t = 0; t < P; t++) {   // the last P output rows end at the bottom padding
      const int r = H - P + t;
      if (r >= 0) {
#pragma unroll
        for (int sl = 0; sl < K; sl++)
          if (r % K == sl) store_row(r, acc[sl]);
      }
    }
  });
}

void set_dw_xcd(int v) { RT_HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_dw_xcd_dev), &v, sizeof(int))); }
int g_dw_wide_slab_min = 192;      // channel pitch from which the 5x5 kernels use wide slabs (1 << 30: never, A/B)
static int dw_lanes_per_pixel(int K, int sh, int sw, int R, int Cp, bool pool);
int g_dw_wide3_min = 128;          // same for the 3x3 kernels (64-channel slabs)
int g_dw_wide_lp = 16;             // 16 = 64-channel slabs, 32 = 128-channel slabs
int g_dw_variant = 0;
int g_dw_sweep = getenv("RT_DW_SWEEP") ? atoi(getenv("RT_DW_SWEEP")) : 4;   // column-sweep 5x5 kernel on short maps: pixels per thread (0: off)
// Output rows per thread of k_dwconv_rows: 4 (stride 1) or 2 (stride 2); 3 for the 3- and 6-row maps of the
// recognition net's last stages, where 4-row (2-row) strips would leave a quarter of the lanes' rows empty.
static int dw_strip_rows(int sh, int maxHo) {
  if (g_dw_variant == 4) return 2;
  if (maxHo == 6 || (maxHo == 3 && sh == 1)) return 3;  // (stride 2 onto 3 rows: 2-row strips measured faster)
  return sh == 1 ? 4 : 2;
}

void dwconv(hipStream_t st, int K, int sh, int sw, const float* x, const ImgGeom* gin, const ImgGeom* gout, int n_img,
            int maxHo, int maxWo, int Cp, int C, const float* Wd, const float* bias, int act, int has_lab, float lab_a,
            float lab_c, float* y, float* pool) {
  if (n_img <= 0) return;
  if ((K == 3 || K == 5) && sh >= 1 && sh <= 2 && sw >= 1 && sw <= 2 && (g_dw_variant == 0 || g_dw_variant == 4)) {  // 4 = 2-row strips (A/B)
    const int R = dw_strip_rows(sh, maxHo);  // input rows streamed: stride 1 -> R+K-1, stride 2 -> 2R+K-2
    long long strips = (long long)((maxWo + 3) / 4) * ((maxHo + R - 1) / R);
    // 64- / 128-channel slabs (256 / 512 contiguous bytes per pixel and load) for wide tensors: the 5x5 kernel on
    // 256 channels goes from 2.9 to 4.1 TB/s with 64-channel slabs; 32-channel slabs otherwise
    const int lp = dw_lanes_per_pixel(K, sh, sw, R, Cp, pool != nullptr);
    // short, wide maps: one thread column sweeps the whole height (RT_DW_SWEEP: 0 off; 2 / 3 / 4 = form, A/B)
    if (g_dw_sweep && K == 5 && sh == 1 && sw == 1 && !pool && lp == 16 && maxHo >= 5 && maxHo <= 24) {
      const int spb = 256 / 16, px = g_dw_sweep == 4 ? 4 : 2;
      dim3 grids((unsigned)(((maxWo + px - 1) / px + spb - 1) / spb), n_img, (Cp + 63) / 64);
      if (px == 4) RT_LAUNCH((k_dwconv_sweep<5, 16, 4, 3>), grids, dim3(256), 0, st, x, gin, gout, Cp, C, Wd, bias, act, has_lab, lab_a, lab_c, y);
      else if (g_dw_sweep == 3) RT_LAUNCH((k_dwconv_sweep<5, 16, 2, 5>), grids, dim3(256), 0, st, x, gin, gout, Cp, C, Wd, bias, act, has_lab, lab_a, lab_c, y);
      else RT_LAUNCH((k_dwconv_sweep<5, 16, 2, 4>), grids, dim3(256), 0, st, x, gin, gout, Cp, C, Wd, bias, act, has_lab, lab_a, lab_c, y);
      return;
    }
    // (round 4) the same sweep for the 3x3 stride-1 layer on the 12-row, 128-channel maps of the recognition net: k_dwconv_rows
    // fetched 1.33x its output there (6-row patches of 4-row strips, PMC)
    if (g_dw_sweep && K == 3 && sh == 1 && sw == 1 && !pool && lp == 16 && maxHo >= 3 && maxHo <= 24) {
      const int spb = 256 / 16;
      dim3 grids((unsigned)(((maxWo + 3) / 4 + spb - 1) / spb), n_img, (Cp + 63) / 64);
      RT_LAUNCH((k_dwconv_sweep<3, 16, 4, 4>), grids, dim3(256), 0, st, x, gin, gout, Cp, C, Wd, bias, act, has_lab, lab_a, lab_c, y);
      return;
    }
    if (lp != 8) {
      const int spb = 256 / lp;
      dim3 gridw((unsigned)((strips + spb - 1) / spb), n_img, (Cp + lp * 4 - 1) / (lp * 4));
#define RT_DWW(KK, RR, SH_, SW_, LL)                                                                                         \
  do {                                                                                                                       \
    if (pool) RT_LAUNCH((k_dwconv_rows<KK, RR, SH_, SW_, 1, LL>), gridw, dim3(256), 0, st, x, gin, gout, Cp, C, Wd, bias, act, has_lab, lab_a, lab_c, y, pool); \
    else RT_LAUNCH((k_dwconv_rows<KK, RR, SH_, SW_, 0, LL>), gridw, dim3(256), 0, st, x, gin, gout, Cp, C, Wd, bias, act, has_lab, lab_a, lab_c, y, pool); \
  } while (0)
#define RT_DWW_L(KK, RR, SH_, SW_) do { if (lp == 16) RT_DWW(KK, RR, SH_, SW_, 16); else RT_DWW(KK, RR, SH_, SW_, 32); } while (0)
      if (K == 5 && sh == 1 && sw == 1 && R == 4) { RT_DWW_L(5, 4, 1, 1); return; }
      if (K == 5 && sh == 1 && sw == 1 && R == 3) { RT_DWW_L(5, 3, 1, 1); return; }
      if (K == 5 && sh == 2 && sw == 1 && R == 2) { RT_DWW_L(5, 2, 2, 1); return; }
      if (K == 5 && sh == 2 && sw == 1 && R == 3) { RT_DWW_L(5, 3, 2, 1); return; }
      if (K == 5 && sh == 2 && sw == 2 && R == 2) { RT_DWW_L(5, 2, 2, 2); return; }
      if (K == 3 && sh == 1 && sw == 1 && R == 4 && !pool) { RT_DWW(3, 4, 1, 1, 16); return; }
      if (K == 3 && sh == 1 && sw == 2 && R == 4 && !pool) { RT_DWW(3, 4, 1, 2, 16); return; }
#undef RT_DWW_L
#undef RT_DWW
      throw RtError(8, "dwconv: no wide-slab instantiation for a shape dw_lanes_per_pixel() lists");
    }
    dim3 grid((unsigned)((strips + 31) / 32), n_img, (Cp + 31) / 32);
#define RT_DWR(KK, RR, SH_, SW_)                                                                                              \
  do {                                                                                                                       \
    if (pool) RT_LAUNCH((k_dwconv_rows<KK, RR, SH_, SW_, 1>), grid, dim3(256), 0, st, x, gin, gout, Cp, C, Wd, bias, act, has_lab, lab_a, lab_c, y, pool); \
    else RT_LAUNCH((k_dwconv_rows<KK, RR, SH_, SW_, 0>), grid, dim3(256), 0, st, x, gin, gout, Cp, C, Wd, bias, act, has_lab, lab_a, lab_c, y, pool); \
  } while (0)
    const int code = (K == 5 ? 4 : 0) + (sh == 2 ? 2 : 0) + (sw == 2 ? 1 : 0);
#define RT_DWR_R(KK, SH_, SW_, RBIG) \
  do { if (R == 3) RT_DWR(KK, 3, SH_, SW_); else if (R == RBIG) RT_DWR(KK, RBIG, SH_, SW_); else RT_DWR(KK, 2, SH_, SW_); } while (0)
    switch (code) {
      case 0: RT_DWR_R(3, 1, 1, 4); break; case 1: RT_DWR_R(3, 1, 2, 4); break;
      case 2: RT_DWR_R(3, 2, 1, 2); break; case 3: RT_DWR_R(3, 2, 2, 2); break;
      case 4: RT_DWR_R(5, 1, 1, 4); break; case 5: RT_DWR_R(5, 1, 2, 4); break;
      case 6: RT_DWR_R(5, 2, 1, 2); break; default: RT_DWR_R(5, 2, 2, 2); break;
    }
#undef RT_DWR_R
#undef RT_DWR
    return;
  }
  throw RtError(8, "dwconv: unsupported kernel size / stride");
}

// ---------------------------------------------------------------------------
// Stem: 3x3 stride 2 pad 1, 3 -> COUT, thread per output pixel.
// ---------------------------------------------------------------------------
template <int COUT>
__global__ __launch_bounds__(256) void k_stem(const float* __restrict__ x, const ImgGeom* __restrict__ gin,
                                              const ImgGeom* __restrict__ gout, const float* __restrict__ Ws,
                                              const float* __restrict__ bias, int act, float* __restrict__ y) {
  __shared__ float w[27 * COUT + COUT];
  for (int i = threadIdx.x; i < 27 * COUT; i += 256) w[i] = Ws[i];
  for (int i = threadIdx.x; i < COUT; i += 256) w[27 * COUT + i] = bias[i];
  __syncthreads();
  const ImgGeom gi = gin[blockIdx.y], go = gout[blockIdx.y];
  long long p = (long long)blockIdx.x * 256 + threadIdx.x;
  if (p >= (long long)go.H * go.W) return;
  int oy = (int)(p / go.W), ox = (int)(p % go.W);
  float acc[COUT];
#pragma unroll
  for (int c = 0; c < COUT; c++) acc[c] = w[27 * COUT + c];
#pragma unroll
  for (int dy = 0; dy < 3; dy++) {
    int iy = oy * 2 - 1 + dy;
    if (iy < 0 || iy >= gi.H) continue;
#pragma unroll
    for (int dx = 0; dx < 3; dx++) {
      int ix = ox * 2 - 1 + dx;
      if (ix < 0 || ix >= gi.W) continue;
      f32x4 v = *reinterpret_cast<const f32x4*>(x + (gi.off + (long long)iy * gi.W + ix) * 4);
      const float* wt = w + (dy * 3 + dx) * 3 * COUT;
#pragma unroll
      for (int ci = 0; ci < 3; ci++)
#pragma unroll
        for (int c = 0; c < COUT; c++) acc[c] = fmaf(v[ci], wt[ci * COUT + c], acc[c]);
    }
  }
  float* o = y + (go.off + p) * COUT;
#pragma unroll
  for (int c = 0; c < COUT; c += 4) {
    f32x4 t = {act_apply(acc[c], act), act_apply(acc[c + 1], act), act_apply(acc[c + 2], act),
               act_apply(acc[c + 3], act)};
    *reinterpret_cast<f32x4*>(o + c) = t;
  }
}

// Det stem straight from the RGB8 pages: DetProcessor::preprocess's normalise (det_processor.rs:151-155: BGR order,
// (v * scale - mean) / std, each operation rounded on its own like k_det_normalize) is applied while the 3x3 window
// is read, so the [H, W, 4] f32 input tensor (16 bytes per pixel written and read back) never exists.
template <int COUT>
__global__ __launch_bounds__(256) void k_stem_u8(const U8Page* __restrict__ pages, float scale, float m0, float m1, float m2,
                                                 float s0, float s1, float s2, const ImgGeom* __restrict__ gin,
                                                 const ImgGeom* __restrict__ gout, const float* __restrict__ Ws,
                                                 const float* __restrict__ bias, int act, float* __restrict__ y) {
  __shared__ float w[27 * COUT + COUT];
  for (int i = threadIdx.x; i < 27 * COUT; i += 256) w[i] = Ws[i];
  for (int i = threadIdx.x; i < COUT; i += 256) w[27 * COUT + i] = bias[i];
  __syncthreads();
  const ImgGeom gi = gin[blockIdx.y], go = gout[blockIdx.y];
  const uint8_t* rgb = pages[blockIdx.y].rgb;
  long long p = (long long)blockIdx.x * 256 + threadIdx.x;
  if (p >= (long long)go.H * go.W) return;
  int oy = (int)(p / go.W), ox = (int)(p % go.W);
  const float mean[3] = {m0, m1, m2}, stdv[3] = {s0, s1, s2};
  float acc[COUT];
#pragma unroll
  for (int c = 0; c < COUT; c++) acc[c] = w[27 * COUT + c];
#pragma unroll
  for (int dy = 0; dy < 3; dy++) {
    int iy = oy * 2 - 1 + dy;
    if (iy < 0 || iy >= gi.H) continue;
#pragma unroll
    for (int dx = 0; dx < 3; dx++) {
      int ix = ox * 2 - 1 + dx;
      if (ix < 0 || ix >= gi.W) continue;
      const uint8_t* px = rgb + ((long long)iy * gi.W + ix) * 3;
      const float* wt = w + (dy * 3 + dx) * 3 * COUT;
#pragma unroll
      for (int ci = 0; ci < 3; ci++) {  // channel ci of BGR
        float v;
        {
          // this file is compiled with -ffp-contract=fast; HIP's __fmul_rn / __fsub_rn are plain operators and
          // `#pragma clang fp contract(off)` did not stop the fusion either (caught by the checksum test): the
          // multiply and the subtract are pinned as separate instructions so that the value is k_det_normalize's
          float t, u;
          const float xf = (float)px[2 - ci], mc = mean[ci];
          asm volatile("v_mul_f32_e32 %0, %1, %2" : "=v"(t) : "v"(xf), "v"(scale));
          asm volatile("v_sub_f32_e32 %0, %1, %2" : "=v"(u) : "v"(t), "v"(mc));
          v = u / stdv[ci];
        }
#pragma unroll
        for (int c = 0; c < COUT; c++) acc[c] = fmaf(v, wt[ci * COUT + c], acc[c]);
      }
    }
  }
  float* o = y + (go.off + p) * COUT;
#pragma unroll
  for (int c = 0; c < COUT; c += 4) {
    f32x4 t = {act_apply(acc[c], act), act_apply(acc[c + 1], act), act_apply(acc[c + 2], act),
               act_apply(acc[c + 3], act)};
    *reinterpret_cast<f32x4*>(o + c) = t;
  }
}
// ---------------------------------------------------------------------------
// Stem on the matrix cores (round 4): 3x3 stride 2, 3 -> 16 channels = a [pixels, 27] x [27, 16] contraction, seven
// v_mfma_f32_16x16x4_f32 steps per 16 pixels.  The thread-per-pixel kernels above ran at 2.1 TB/s of their bytes: 27 strided
// byte loads + 27 IEEE divisions (U8) and 432 scalar FMAs per output pixel.  Here a workgroup stages the (2 TH + 1) x (2 TW + 1)
// input patch of a TH x TW = 8 x 32 output tile ONCE into LDS as normalised floats (coalesced byte / 16-byte loads, each input
// element converted once), and lane (r, q) of a wave reads operand k = 4 step + q of pixel r with one ds_read_b32 per step
// (consecutive pixels are 6 floats apart: 16 distinct banks); the weights are the MFMA "A" operand (seven registers for the
// whole kernel), so a lane ends with 4 consecutive channels of a pixel and a wave stores 1 KB of contiguous output.
// U8 = 1: the RGB8 page with DetProcessor::preprocess's normalise (det_processor.rs:151-155) applied while staging, same three
// separately rounded operations as k_det_normalize; U8 = 0: the f32 NHWC-4 tensor.  Both stage the same floats and run the same
// MFMA code: bit-identical results (test_det_stem_from_u8_pages_matches_tensor_path).
// ---------------------------------------------------------------------------
template <int U8>
__global__ __launch_bounds__(256) void k_stem_mfma(const float* __restrict__ x, const U8Page* __restrict__ pages, float scale, float m0,
                                                   float m1, float m2, float s0, float s1, float s2, const ImgGeom* __restrict__ gin,
                                                   const ImgGeom* __restrict__ gout, const float* __restrict__ Ws,
                                                   const float* __restrict__ bias, int act, float* __restrict__ y) {
  constexpr int TH = 8, TW = 32, PH = 2 * TH + 1, PW = 2 * TW + 1, PITCH = PW * 3 + 1;   // 196 floats per patch row
  __shared__ float patch[PH * PITCH];
  const ImgGeom gi = gin[blockIdx.y], go = gout[blockIdx.y];
  const int tiles_x = (go.W + TW - 1) / TW, tiles_y = (go.H + TH - 1) / TH;
  if ((int)blockIdx.x >= tiles_x * tiles_y) return;
  const int ty = blockIdx.x / tiles_x, tx = blockIdx.x - ty * tiles_x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 15, q = lane >> 4;
  const int iy0 = ty * TH * 2 - 1, ix0 = tx * TW * 2 - 1;   // input coordinates of the patch origin (pad 1)
  // staging: a thread per patch PIXEL (index arithmetic once per pixel, carried from one pixel to the next without divisions)
  constexpr int NP = PH * PW, NL = (NP + 255) / 256;
  int pr = tid / PW, pc = tid - pr * PW;
  if (U8) {
    // the normalised value of a page byte depends on (channel, byte) only: a 3 x 256 table per workgroup, every entry computed
    // with DetProcessor::preprocess's three separately rounded operations, replaces ~15 VALU ops (an IEEE division) per byte
    __shared__ float lut[3 * 256];
    {
      const float mean[3] = {m0, m1, m2}, stdv[3] = {s0, s1, s2};
#pragma unroll
      for (int ci = 0; ci < 3; ci++) {
        float t, u;
        const float xf = (float)tid, mc = mean[ci];
        // (this file is compiled with -ffp-contract=fast: the multiply and the subtract are pinned as separate instructions so
        //  that the value is k_det_normalize's)
        asm volatile("v_mul_f32_e32 %0, %1, %2" : "=v"(t) : "v"(xf), "v"(scale));
        asm volatile("v_sub_f32_e32 %0, %1, %2" : "=v"(u) : "v"(t), "v"(mc));
        lut[ci * 256 + tid] = u / stdv[ci];
      }
    }
    const uint8_t* rgb = pages[blockIdx.y].rgb;
    uint8_t raw[NL][3];
    unsigned okm = 0;
    {
      int r_ = pr, c_ = pc;
#pragma unroll
      for (int i = 0; i < NL; i++) {
        const int iy = iy0 + min(r_, PH - 1), ix = ix0 + c_;
        const bool ok = iy >= 0 && iy < gi.H && ix >= 0 && ix < gi.W;
        const uint8_t* px = rgb + (ok ? ((long long)iy * gi.W + ix) * 3 : 0);
        raw[i][0] = px[0]; raw[i][1] = px[1]; raw[i][2] = px[2];
        okm |= (ok ? 1u : 0u) << i;
        r_ += 256 / PW; c_ += 256 % PW;
        if (c_ >= PW) { c_ -= PW; r_++; }
      }
    }
    __syncthreads();   // the table
    {
      int r_ = pr, c_ = pc;
#pragma unroll
      for (int i = 0; i < NL; i++) {
        if (r_ < PH) {
          const bool ok = (okm >> i) & 1;
          float* d = patch + r_ * PITCH + c_ * 3;   // tensor channel ci = page byte 2 - ci (BGR)
          d[0] = ok ? lut[raw[i][2]] : 0.f; d[1] = ok ? lut[256 + raw[i][1]] : 0.f; d[2] = ok ? lut[512 + raw[i][0]] : 0.f;
        }
        r_ += 256 / PW; c_ += 256 % PW;
        if (c_ >= PW) { c_ -= PW; r_++; }
      }
    }
  } else {
    f32x4 raw[NL];
    unsigned okm = 0;
    {
      int r_ = pr, c_ = pc;
#pragma unroll
      for (int i = 0; i < NL; i++) {
        const int iy = iy0 + min(r_, PH - 1), ix = ix0 + c_;
        const bool ok = iy >= 0 && iy < gi.H && ix >= 0 && ix < gi.W;
        raw[i] = *reinterpret_cast<const f32x4*>(x + (gi.off + (ok ? (long long)iy * gi.W + ix : 0)) * 4);
        okm |= (ok ? 1u : 0u) << i;
        r_ += 256 / PW; c_ += 256 % PW;
        if (c_ >= PW) { c_ -= PW; r_++; }
      }
    }
    {
      int r_ = pr, c_ = pc;
#pragma unroll
      for (int i = 0; i < NL; i++) {
        if (r_ < PH) {
          const bool ok = (okm >> i) & 1;
          float* d = patch + r_ * PITCH + c_ * 3;
          d[0] = ok ? raw[i][0] : 0.f; d[1] = ok ? raw[i][1] : 0.f; d[2] = ok ? raw[i][2] : 0.f;
        }
        r_ += 256 / PW; c_ += 256 % PW;
        if (c_ >= PW) { c_ -= PW; r_++; }
      }
    }
  }
  // operand maps: k = 4 step + q = (dy * 3 + dx) * 3 + ci; k = 27 is padding (zero weight, any finite pixel value)
  float wa[7];
  int off[7];
#pragma unroll
  for (int sidx = 0; sidx < 7; sidx++) {
    const int k = 4 * sidx + q, kk = min(k, 26);
    const int tap = kk / 3, ci = kk - tap * 3, dy = tap / 3, dx = tap - dy * 3;
    wa[sidx] = k < 27 ? Ws[k * 16 + r] : 0.f;
    off[sidx] = dy * PITCH + dx * 3 + ci;
  }
  const f32x4 bv = *reinterpret_cast<const f32x4*>(bias + 4 * q);
  __syncthreads();
#pragma unroll
  for (int gidx = 0; gidx < 4; gidx++) {   // this wave: rows 2 wave, 2 wave + 1; 32 pixels each = 2 groups of 16
    const int ly = 2 * wave + (gidx >> 1), lx = (gidx & 1) * 16 + r;
    const float* pb = patch + (2 * ly) * PITCH + (2 * lx) * 3;
    f32x4 acc = bv;
#pragma unroll
    for (int sidx = 0; sidx < 7; sidx++) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[sidx], pb[off[sidx]], acc, 0, 0, 0);
    const int oy = ty * TH + ly, ox = tx * TW + lx;
    if (oy < go.H && ox < go.W) {
      f32x4 o = {act_apply(acc[0], act), act_apply(acc[1], act), act_apply(acc[2], act), act_apply(acc[3], act)};
      *reinterpret_cast<f32x4*>(y + (go.off + (long long)oy * go.W + ox) * 16 + 4 * q) = o;
    }
  }
}
static const int g_stem_mfma = getenv("RT_STEM_MFMA") ? atoi(getenv("RT_STEM_MFMA")) : 1;   // A/B: 0 = thread-per-pixel stems

void stem_conv_u8(hipStream_t st, const U8Page* pages, float scale, const float* mean3, const float* std3, const ImgGeom* gin,
                  const ImgGeom* gout, int n_img, int maxHo, int maxWo, int COUT, const float* Ws, const float* bias, int act,
                  float* y) {
  if (n_img <= 0) return;
  if (COUT != 16) throw RtError(8, "stem_conv_u8: unsupported COUT");
  if (g_stem_mfma) {
    dim3 gridm((unsigned)(((maxWo + 31) / 32) * ((maxHo + 7) / 8)), n_img);
    RT_LAUNCH(k_stem_mfma<1>, gridm, dim3(256), 0, st, (const float*)nullptr, pages, scale, mean3[0], mean3[1], mean3[2], std3[0],
              std3[1], std3[2], gin, gout, Ws, bias, act, y);
    return;
  }
  dim3 grid((unsigned)(((long long)maxHo * maxWo + 255) / 256), n_img);
  RT_LAUNCH(k_stem_u8<16>, grid, dim3(256), 0, st, pages, scale, mean3[0], mean3[1], mean3[2], std3[0], std3[1],
                     std3[2], gin, gout, Ws, bias, act, y);
}

void stem_conv(hipStream_t st, const float* x, const ImgGeom* gin, const ImgGeom* gout, int n_img, int maxHo, int maxWo,
               int COUT, const float* Ws, const float* bias, int act, float* y) {
  if (n_img <= 0) return;
  if (COUT == 16 && g_stem_mfma) {
    dim3 gridm((unsigned)(((maxWo + 31) / 32) * ((maxHo + 7) / 8)), n_img);
    RT_LAUNCH(k_stem_mfma<0>, gridm, dim3(256), 0, st, x, (const U8Page*)nullptr, 0.f, 0.f, 0.f, 0.f, 1.f, 1.f, 1.f, gin, gout, Ws,
              bias, act, y);
    return;
  }
  dim3 grid((unsigned)(((long long)maxHo * maxWo + 255) / 256), n_img);
  if (COUT == 16) RT_LAUNCH(k_stem<16>, grid, dim3(256), 0, st, x, gin, gout, Ws, bias, act, y);
  else if (COUT == 8) RT_LAUNCH(k_stem<8>, grid, dim3(256), 0, st, x, gin, gout, Ws, bias, act, y);
  else throw RtError(8, "stem_conv: unsupported COUT");
}

__global__ void k_nchw3_to_nhwc4(const float* __restrict__ in, int n, int H, int W, float* __restrict__ out) {
  long long p = (long long)blockIdx.x * 256 + threadIdx.x;
  long long plane = (long long)H * W;
  if (p >= plane * n) return;
  long long im = p / plane, pp = p % plane;
  const float* b = in + im * 3 * plane + pp;
  f32x4 v = {b[0], b[plane], b[2 * plane], 0.0f};
  *reinterpret_cast<f32x4*>(out + p * 4) = v;
}
void nchw3_to_nhwc4(hipStream_t st, const float* in, int n, int H, int W, float* out) {
  long long total = (long long)n * H * W;
  if (total <= 0) return;
  RT_LAUNCH(k_nchw3_to_nhwc4, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, in, n, H, W, out);
}

// ---------------------------------------------------------------------------
// Squeeze-excite / global mean: deterministic two-stage reduction.
// ---------------------------------------------------------------------------
// (round 5: 128 pixels per block, was 1024 -- with 480 channels a thread summed 512 pixels one dependent load after the other, 27 us
//  per launch whatever the batch: 8 launches = 18 % of the one-page C2 call; k_se_fc adds the chunks 16 bytes wide on up to 1024 threads)
constexpr int POOL_PIX = 128;  // pixels per partial block
int pool_chunks(long long max_pix) { return (int)((max_pix + POOL_PIX - 1) / POOL_PIX); }

__global__ __launch_bounds__(256) void k_pool_partial(const float* __restrict__ x, const ImgGeom* __restrict__ geom,
                                                      int Cp, int chunks, float* __restrict__ partial) {
  __shared__ __attribute__((aligned(16))) float red[256 * 4];
  const ImgGeom g = geom[blockIdx.y];
  const long long npix = (long long)g.H * g.W;
  const long long p0 = (long long)blockIdx.x * POOL_PIX;
  const int C4 = Cp >> 2;
  float* out = partial + ((long long)blockIdx.y * chunks + blockIdx.x) * Cp;
  // channel groups are processed in passes of up to 256/PL groups... keep it simple:
  // PL pixel lanes per channel group so that PL * C4pass <= 256.
  for (int cbase = 0; cbase < C4; cbase += 256) {
    int cgroups = min(256, C4 - cbase);
    int PL = 256 / cgroups;
    int c4 = cbase + (threadIdx.x % cgroups), pl = threadIdx.x / cgroups;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    if (pl < PL && p0 < npix) {
      long long pend = min(npix, p0 + POOL_PIX);
#pragma unroll 4
      for (long long p = p0 + pl; p < pend; p += PL) {
        f32x4 v = *reinterpret_cast<const f32x4*>(x + (g.off + p) * Cp + c4 * 4);
        s += v;
      }
    }
    *reinterpret_cast<f32x4*>(red + threadIdx.x * 4) = s;
    __syncthreads();
    if (threadIdx.x < cgroups) {
      f32x4 t = {0.f, 0.f, 0.f, 0.f};
      for (int l = 0; l < PL; l++) t += *reinterpret_cast<const f32x4*>(red + (l * cgroups + threadIdx.x) * 4);
      *reinterpret_cast<f32x4*>(out + (cbase + threadIdx.x) * 4) = t;
    }
    __syncthreads();
  }
}

// block per image: mean -> fc1 -> relu -> fc2 -> hardsigmoid.  Round 4: every phase on all threads of the block (the first form
// summed the chunks with one thread per channel and ran each hidden unit's C-long dot product on one thread: 14 us per launch).
// Round 5: the launch was still 31 us whatever the batch -- 10 of them are a quarter of the C2 call -- because every phase was a
// chain of dependent-latency iterations (chunks, then C / 8 per hidden unit, then Cr per channel: ~490 load -> fma steps).  Now up
// to 1024 threads, 16-byte loads everywhere, 16 lanes per hidden unit and 4 per channel: ~35 steps.  Every order of summation is
// fixed by the thread layout and the block size, which depends on nothing but the layer: repeatable, and independent of the batch
// the image is in.
__global__ __launch_bounds__(1024) void k_se_fc(const float* __restrict__ partial, const ImgGeom* __restrict__ geom,
                                                int chunks_alloc, int C, int Cp, const float* __restrict__ w1,
                                                const float* __restrict__ b1, const float* __restrict__ w2,
                                                const float* __restrict__ b2, int Cr, float slope, int residual,
                                                float* __restrict__ scale, int strip_R, int strips_per_block,
                                                const float* __restrict__ Wlin, int Cin, int Cin_p) {
  extern __shared__ __attribute__((aligned(16))) float sm[];  // mean[Cp] | hid[Cr4 + 4] | mean_in[Cin_p + 4] (when projecting) | red[4 NT]
  const int NT = (int)blockDim.x, Cr4 = (Cr + 3) & ~3;
  float* mean = sm;
  float* hid = sm + Cp;
  float* mean_in = hid + Cr4 + 4;
  float* red = mean_in + (Wlin ? Cin_p + 4 : 0);
  const ImgGeom g = geom[blockIdx.x];
  const long long npix = (long long)g.H * g.W;
  const int tid = threadIdx.x;
  // partial sums come from k_pool_partial (POOL_PIX pixels each) or, strip_R > 0, from the blocks of the
  // depthwise kernel that produced the tensor (32 strips of strip_R x 4 pixels each)
  // ... or, strip_R < 0, from the 16 x 16-pixel tiles of the kernel that produced it (k_fpn_phase)
  const int chunks = strip_R > 0 ? (((g.W + 3) >> 2) * ((g.H + strip_R - 1) / strip_R) + strips_per_block - 1) / strips_per_block
                     : strip_R < 0 ? ((g.W + 15) >> 4) * ((g.H + 15) >> 4)
                                 : (int)((npix + POOL_PIX - 1) / POOL_PIX);
  const float inv = 1.0f / (float)npix;
  {
    // channel sums: Cs channels (of the narrow tensor when projecting) as Cs / 4 columns of 16 bytes, NT / (Cs / 4) threads per
    // column each adding every PL-th chunk, then the parts in order
    const int Cs = Wlin ? Cin_p : Cp, C4 = Cs >> 2;
    float* dst = Wlin ? mean_in : mean;
    const float* part = partial + (long long)blockIdx.x * chunks_alloc * Cs;
    f32x4* red4 = reinterpret_cast<f32x4*>(red);
    for (int cb = 0; cb < C4; cb += NT) {
      const int cg = min(NT, C4 - cb), PL = NT / cg, col = cb + tid % cg, pl = tid / cg;
      f32x4 s0 = {0.f, 0.f, 0.f, 0.f};
      if (pl < PL) {
#pragma unroll 4
        for (int k = pl; k < chunks; k += PL) s0 += *reinterpret_cast<const f32x4*>(part + (long long)k * Cs + col * 4);
      }
      red4[tid] = s0;
      __syncthreads();
      if (tid < cg) {
        f32x4 t = {0.f, 0.f, 0.f, 0.f};
        for (int i = 0; i < PL; i++) t += red4[i * cg + tid];
        *reinterpret_cast<f32x4*>(dst + (cb + tid) * 4) = t * inv;
      }
      __syncthreads();
    }
  }
  if (Wlin) {
    // The pooled tensor is a bias-free 1x1 conv of a narrower one (FPN lateral: y = x . Wlin): mean(y) = mean(x) . Wlin,
    // so the partial sums are those of x (pitch Cin_p) and y itself never has to exist for the squeeze.
    if (Cp <= NT) {   // NT / Cp threads per output channel, each every G-th input channel; the parts in order
      const int G = NT / Cp, gk = tid / Cp, c = tid - gk * Cp;
      float s0 = 0.f;
      if (gk < G && c < C) {
#pragma unroll 4
        for (int k = gk; k < Cin; k += G) s0 = fmaf(mean_in[k], Wlin[k * C + c], s0);
      }
      red[tid] = s0;
      __syncthreads();
      if (tid < Cp) {
        float t = 0.f;
        for (int i = 0; i < G; i++) t += red[i * Cp + tid];
        mean[tid] = tid < C ? t : 0.f;
      }
    } else {
      for (int c = tid; c < Cp; c += NT) {
        float s0 = 0.f;
        if (c < C) for (int k = 0; k < Cin; k++) s0 = fmaf(mean_in[k], Wlin[k * C + c], s0);
        mean[c] = s0;
      }
    }
    __syncthreads();
  }
  if (w1 == nullptr) {  // plain global mean
    for (int c = tid; c < Cp; c += NT) scale[(long long)blockIdx.x * Cp + c] = mean[c];
    return;
  }
  if (tid < Cr4 + 4 - Cr) hid[Cr + tid] = 0.f;
  if (((C & 3) | (int)((size_t)w1 & 15)) == 0) {  // fc1, w1 [Cr][C]: 16 adjacent lanes per hidden unit, 16-byte loads, NT / 16 units per pass
    const int l = tid & 15, UP = NT >> 4, n4 = C >> 2;
    const f32x4* m4 = reinterpret_cast<const f32x4*>(mean);
    for (int j0 = 0; j0 < Cr; j0 += UP) {
      const int j = j0 + (tid >> 4);
      float s0 = 0.f;
      if (j < Cr) {
        const f32x4* wr = reinterpret_cast<const f32x4*>(w1 + (long long)j * C);
#pragma unroll 4
        for (int c4 = l; c4 < n4; c4 += 16) {
          const f32x4 w = wr[c4], m = m4[c4];
          s0 = fmaf(m[0], w[0], s0); s0 = fmaf(m[1], w[1], s0); s0 = fmaf(m[2], w[2], s0); s0 = fmaf(m[3], w[3], s0);
        }
      }
      s0 += __shfl_xor(s0, 8); s0 += __shfl_xor(s0, 4); s0 += __shfl_xor(s0, 2); s0 += __shfl_xor(s0, 1);
      if (j < Cr && l == 0) hid[j] = fmaxf(s0 + b1[j], 0.f);
    }
  } else {  // (C not a multiple of 4: 8 lanes per unit, scalar loads)
    const int l = tid & 7, UP = NT >> 3;
    for (int j0 = 0; j0 < Cr; j0 += UP) {
      const int j = j0 + (tid >> 3);
      float s0 = 0.f;
      if (j < Cr) {
        const float* wr = w1 + (long long)j * C;
        for (int c = l; c < C; c += 8) s0 = fmaf(mean[c], wr[c], s0);
      }
      s0 += __shfl_xor(s0, 4); s0 += __shfl_xor(s0, 2); s0 += __shfl_xor(s0, 1);
      if (j < Cr && l == 0) hid[j] = fmaxf(s0 + b1[j], 0.f);
    }
  }
  __syncthreads();
  if (((Cr & 3) | (int)((size_t)w2 & 15)) == 0) {  // fc2, w2 [C][Cr]: 4 adjacent lanes per channel, 16-byte loads
    const int l = tid & 3, CPP = NT >> 2, n4 = Cr >> 2;
    const f32x4* h4 = reinterpret_cast<const f32x4*>(hid);
    for (int c0 = 0; c0 < Cp; c0 += CPP) {
      const int c = c0 + (tid >> 2);
      float s0 = 0.f;
      if (c < C) {
        const f32x4* wr = reinterpret_cast<const f32x4*>(w2 + (long long)c * Cr);
#pragma unroll 4
        for (int j4 = l; j4 < n4; j4 += 4) {
          const f32x4 w = wr[j4], h = h4[j4];
          s0 = fmaf(h[0], w[0], s0); s0 = fmaf(h[1], w[1], s0); s0 = fmaf(h[2], w[2], s0); s0 = fmaf(h[3], w[3], s0);
        }
      }
      s0 += __shfl_xor(s0, 2); s0 += __shfl_xor(s0, 1);
      if (l == 0 && c < Cp) {
        float o = 0.f;
        if (c < C) {
          o = fminf(fmaxf(fmaf(s0 + b2[c], slope, 0.5f), 0.f), 1.f);
          if (residual) o += 1.0f;
        }
        scale[(long long)blockIdx.x * Cp + c] = o;
      }
    }
  } else {
    for (int c = tid; c < Cp; c += NT) {
      float o = 0.f;
      if (c < C) {
        float s0 = b2[c];
        const float* wr = w2 + (long long)c * Cr;
#pragma unroll 4
        for (int j = 0; j < Cr; j++) s0 = fmaf(hid[j], wr[j], s0);
        o = fminf(fmaxf(fmaf(s0, slope, 0.5f), 0.f), 1.f);
        if (residual) o += 1.0f;
      }
      scale[(long long)blockIdx.x * Cp + c] = o;
    }
  }
}
// block size of k_se_fc: decided by the layer alone (the summation orders depend on it): 1024 threads for the wide layers, whose
// fc matrices are 30-230 KB; 256 where a block has nothing to spread (C <= 64)
static int se_fc_threads(int Cp, int Cin_p) { return std::max(Cp, Cin_p) > 64 ? 1024 : 256; }
static size_t se_fc_lds(int Cp, int Cr, int Cin_p, bool proj, int nt) {
  return (size_t)(Cp + ((Cr + 3) & ~3) + 4 + (proj ? Cin_p + 4 : 0) + 4 * nt) * sizeof(float);
}

void se_scale(hipStream_t st, const float* x, const ImgGeom* geom, int n_img, long long max_pix, int C, int Cp,
              const float* w1, const float* b1, const float* w2, const float* b2, int Cr, float slope, int residual,
              float* partial, float* scale) {
  if (n_img <= 0) return;
  int chunks = pool_chunks(max_pix);
  RT_LAUNCH(k_pool_partial, dim3(chunks, n_img), dim3(256), 0, st, x, geom, Cp, chunks, partial);
  const int nt = se_fc_threads(Cp, 0);
  RT_LAUNCH(k_se_fc, dim3(n_img), dim3(nt), se_fc_lds(Cp, Cr, 0, false, nt), st, partial, geom, chunks, C, Cp,
                     w1, b1, w2, b2, Cr, slope, residual, scale, 0, 32, (const float*)nullptr, 0, 0);
}
void se_scale_projected(hipStream_t st, const float* x_in, const ImgGeom* geom, int n_img, long long max_pix, int Cin,
                        int Cin_p, const float* Wlin, int C, int Cp, const float* w1, const float* b1, const float* w2,
                        const float* b2, int Cr, float slope, int residual, float* partial, float* scale) {
  if (n_img <= 0) return;
  int chunks = pool_chunks(max_pix);
  RT_LAUNCH(k_pool_partial, dim3(chunks, n_img), dim3(256), 0, st, x_in, geom, Cin_p, chunks, partial);
  const int nt = se_fc_threads(Cp, Cin_p);
  RT_LAUNCH(k_se_fc, dim3(n_img), dim3(nt), se_fc_lds(Cp, Cr, Cin_p, true, nt), st, partial, geom, chunks, C, Cp,
                     w1, b1, w2, b2, Cr, slope, residual, scale, 0, 32, Wlin, Cin, Cin_p);
}
// Lanes side by side on a pixel in k_dwconv_rows for this layer: 16 / 32 (64- / 128-channel slabs) where the tensor is
// wide AND a wide instantiation exists for the shape, else 8 (32-channel slabs).  The one place that decides it: the
// launch (dwconv) and the layout of the fused pooling partials (dwconv_pool_layout -> k_se_fc) must agree.
static int dw_lanes_per_pixel(int K, int sh, int sw, int R, int Cp, bool pool) {
  int lp = 8;
  if (K == 5 && Cp >= g_dw_wide_slab_min) lp = g_dw_wide_lp;
  else if (K == 3 && Cp >= g_dw_wide3_min) lp = 16;
  if (lp == 8) return 8;
  if (K == 5 && sh == 1 && sw == 1 && (R == 4 || R == 3)) return lp;
  if (K == 5 && sh == 2 && sw == 1 && (R == 2 || R == 3)) return lp;
  if (K == 5 && sh == 2 && sw == 2 && R == 2) return lp;
  if (K == 3 && sh == 1 && (sw == 1 || sw == 2) && R == 4 && !pool) return 16;
  return 8;
}
void dwconv_pool_layout(int K, int sh, int sw, int Cp, int maxHo, int maxWo, int* chunks, int* strip_R, int* strips_per_block) {
  const int R = dw_strip_rows(sh, maxHo), spb = 256 / dw_lanes_per_pixel(K, sh, sw, R, Cp, true);
  *strip_R = R; *strips_per_block = spb;
  *chunks = (int)(((long long)((maxWo + 3) / 4) * ((maxHo + R - 1) / R) + spb - 1) / spb);
}
void se_fc_from_dw(hipStream_t st, const float* partial, const ImgGeom* geom, int n_img, int chunks, int strip_R,
                   int strips_per_block, int C, int Cp, const float* w1, const float* b1, const float* w2, const float* b2,
                   int Cr, float slope, int residual, float* scale) {
  if (n_img <= 0) return;
  const int nt = se_fc_threads(Cp, 0);
  RT_LAUNCH(k_se_fc, dim3(n_img), dim3(nt), se_fc_lds(Cp, Cr, 0, false, nt), st, partial, geom, chunks, C, Cp,
                     w1, b1, w2, b2, Cr, slope, residual, scale, strip_R, strips_per_block, (const float*)nullptr, 0, 0);
}
void se_fc_from_tiles(hipStream_t st, const float* partial, const ImgGeom* geom, int n_img, int tiles_alloc, int C, int Cp,
                      const float* w1, const float* b1, const float* w2, const float* b2, int Cr, float slope, int residual,
                      float* scale) {
  if (n_img <= 0) return;
  const int nt = se_fc_threads(Cp, 0);
  RT_LAUNCH(k_se_fc, dim3(n_img), dim3(nt), se_fc_lds(Cp, Cr, 0, false, nt), st, partial, geom, tiles_alloc, C, Cp,
                     w1, b1, w2, b2, Cr, slope, residual, scale, -16, 32, (const float*)nullptr, 0, 0);
}
void global_mean(hipStream_t st, const float* x, const ImgGeom* geom, int n_img, long long max_pix, int Cp,
                 float* partial, float* out) {
  if (n_img <= 0) return;
  int chunks = pool_chunks(max_pix);
  RT_LAUNCH(k_pool_partial, dim3(chunks, n_img), dim3(256), 0, st, x, geom, Cp, chunks, partial);
  const int nt = se_fc_threads(Cp, 0);
  RT_LAUNCH(k_se_fc, dim3(n_img), dim3(nt), se_fc_lds(Cp, 0, 0, false, nt), st, partial, geom, chunks, Cp, Cp,
                     (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, 0,
                     0.f, 0, out, 0, 32, (const float*)nullptr, 0, 0);
}

__global__ __launch_bounds__(256) void k_scale_channels(float* __restrict__ x, const ImgGeom* __restrict__ geom, int Cp,
                                                        const float* __restrict__ scale) {
  const ImgGeom g = geom[blockIdx.y];
  const int C4 = Cp >> 2;
  long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long long)g.H * g.W * C4) return;
  int c4 = (int)(idx % C4);
  f32x4 s = *reinterpret_cast<const f32x4*>(scale + (long long)blockIdx.y * Cp + c4 * 4);
  f32x4* p = reinterpret_cast<f32x4*>(x + g.off * Cp) + idx;
  f32x4 v = *p;
  v *= s;
  *p = v;
}
void scale_channels(hipStream_t st, float* x, const ImgGeom* geom, int n_img, long long max_pix, int Cp,
                    const float* scale) {
  if (n_img <= 0) return;
  long long total = max_pix * (Cp / 4);
  RT_LAUNCH(k_scale_channels, dim3((unsigned)((total + 255) / 256), n_img), dim3(256), 0, st, x, geom, Cp,
                     scale);
}

// ---------------------------------------------------------------------------
// FPN glue
// ---------------------------------------------------------------------------
// FPN lateral + top-down add in one pass: out = (x . Wlin) * s + nearest-2x(b).  x is the narrow tap tensor (pitch
// Cin_p), Wlin [Cin][C] the bias-free 1x1 lateral conv, s [image][C] its squeeze-excite factor (from
// se_scale_projected); the C-channel lateral tensor is written once, already scaled and summed.
__global__ __launch_bounds__(256) void k_lateral_add(const float* __restrict__ x, int Cin, int Cin_p,
                                                     const float* __restrict__ Wlin, int C, const float* __restrict__ s,
                                                     const float* __restrict__ b, const ImgGeom* __restrict__ ga,
                                                     const ImgGeom* __restrict__ gb, float* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) float wl[];  // [Cin][C]
  for (int i = threadIdx.x; i < Cin * C / 4; i += 256) reinterpret_cast<f32x4*>(wl)[i] = reinterpret_cast<const f32x4*>(Wlin)[i];
  __syncthreads();
  const ImgGeom A = ga[blockIdx.y];
  const int C4 = C >> 2;
  constexpr int PPT = 4;  // consecutive pixels per thread: every weight vector read from LDS feeds 4 pixels
  const long long npix = (long long)A.H * A.W;
  long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  const int c4 = (int)(idx % C4);
  const long long p0 = (idx / C4) * PPT;
  if (p0 >= npix) return;
  f32x4 acc[PPT];
#pragma unroll
  for (int i = 0; i < PPT; i++) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int k4 = 0; k4 < Cin_p; k4 += 4) {
    f32x4 xv[PPT];
#pragma unroll
    for (int i = 0; i < PPT; i++)
      xv[i] = (p0 + i < npix) ? *reinterpret_cast<const f32x4*>(x + (A.off + p0 + i) * Cin_p + k4) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int e = 0; e < 4; e++) {
      if (k4 + e < Cin) {
        const f32x4 w = *reinterpret_cast<const f32x4*>(wl + (k4 + e) * C + c4 * 4);
#pragma unroll
        for (int i = 0; i < PPT; i++)
#pragma unroll
          for (int j = 0; j < 4; j++) acc[i][j] = fmaf(xv[i][e], w[j], acc[i][j]);
      }
    }
  }
  const f32x4 sc = *reinterpret_cast<const f32x4*>(s + (long long)blockIdx.y * C + c4 * 4);
  const ImgGeom B = b ? gb[blockIdx.y] : A;
#pragma unroll
  for (int i = 0; i < PPT; i++) {
    const long long p = p0 + i;
    if (p >= npix) break;
    f32x4 o = acc[i] * sc;
    if (b) {
      const int y = (int)(p / A.W), xx = (int)(p % A.W);
      const int by = min(y >> 1, B.H - 1), bx = min(xx >> 1, B.W - 1);
      o += *reinterpret_cast<const f32x4*>(b + (B.off + (long long)by * B.W + bx) * C + c4 * 4);
    }
    *reinterpret_cast<f32x4*>(out + (A.off + p) * C + c4 * 4) = o;
  }
}
void lateral_add(hipStream_t st, const float* x, int Cin, int Cin_p, const float* Wlin, int C, const float* scale,
                 const float* b, const ImgGeom* ga, const ImgGeom* gb, int n_img, long long max_pix, float* out) {
  if (n_img <= 0) return;
  long long total = ((max_pix + 3) / 4) * (C / 4);  // 4 pixels per thread
  RT_LAUNCH(k_lateral_add, dim3((unsigned)((total + 255) / 256), n_img), dim3(256), (size_t)Cin * C * sizeof(float), st,
                     x, Cin, Cin_p, Wlin, C, scale, b, ga, gb, out);
}

// out = a * sa + nearest-2x(b); sa (optional, [image][Cp]) is the squeeze-excite scale of `a`, folded in here
// instead of a separate read+write pass over `a`.
__global__ __launch_bounds__(256) void k_upsample_add(const float* __restrict__ a, const float* __restrict__ b,
                                                      const ImgGeom* __restrict__ ga, const ImgGeom* __restrict__ gb,
                                                      int Cp, float* __restrict__ out, const float* __restrict__ sa) {
  const ImgGeom A = ga[blockIdx.y], B = gb[blockIdx.y];
  const int C4 = Cp >> 2;
  long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long long)A.H * A.W * C4) return;
  int c4 = (int)(idx % C4);
  long long p = idx / C4;
  int y = (int)(p / A.W), x = (int)(p % A.W);
  int by = min(y >> 1, B.H - 1), bx = min(x >> 1, B.W - 1);
  f32x4 va = *reinterpret_cast<const f32x4*>(a + (A.off + p) * Cp + c4 * 4);
  if (sa) va *= *reinterpret_cast<const f32x4*>(sa + (long long)blockIdx.y * Cp + c4 * 4);
  f32x4 vb = *reinterpret_cast<const f32x4*>(b + (B.off + (long long)by * B.W + bx) * Cp + c4 * 4);
  *reinterpret_cast<f32x4*>(out + (A.off + p) * Cp + c4 * 4) = va + vb;
}
void upsample_add(hipStream_t st, const float* a, const float* b, const ImgGeom* ga, const ImgGeom* gb, int n_img,
                  long long max_pix, int Cp, float* out, const float* scale_a) {
  if (n_img <= 0) return;
  long long total = max_pix * (Cp / 4);
  RT_LAUNCH(k_upsample_add, dim3((unsigned)((total + 255) / 256), n_img), dim3(256), 0, st, a, b, ga, gb, Cp,
                     out, scale_a);
}

__global__ __launch_bounds__(256) void k_fpn_concat(const float* __restrict__ p5, const float* __restrict__ p4,
                                                    const float* __restrict__ p3, const float* __restrict__ p2,
                                                    const ImgGeom* __restrict__ g5, const ImgGeom* __restrict__ g4,
                                                    const ImgGeom* __restrict__ g3, const ImgGeom* __restrict__ g2,
                                                    int Cq, float* __restrict__ out, const float* __restrict__ s5,
                                                    const float* __restrict__ s4, const float* __restrict__ s3,
                                                    const float* __restrict__ s2) {
  const ImgGeom G2 = g2[blockIdx.y];
  const int Q4 = Cq >> 2, C4 = Q4 * 4;
  long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long long)G2.H * G2.W * C4) return;
  int c4 = (int)(idx % C4);
  long long p = idx / C4;
  int y = (int)(p / G2.W), x = (int)(p % G2.W);
  int lvl = c4 / Q4, cc = c4 % Q4;
  const float* src;
  const float* sc;  // optional squeeze-excite scale of that level ([image][Cq]), folded into the gather
  ImgGeom G;
  int sh;
  if (lvl == 0) { src = p5; sc = s5; G = g5[blockIdx.y]; sh = 3; }
  else if (lvl == 1) { src = p4; sc = s4; G = g4[blockIdx.y]; sh = 2; }
  else if (lvl == 2) { src = p3; sc = s3; G = g3[blockIdx.y]; sh = 1; }
  else { src = p2; sc = s2; G = G2; sh = 0; }
  int sy = min(y >> sh, G.H - 1), sx = min(x >> sh, G.W - 1);
  f32x4 v = *reinterpret_cast<const f32x4*>(src + (G.off + (long long)sy * G.W + sx) * Cq + cc * 4);
  if (sc) v *= *reinterpret_cast<const f32x4*>(sc + (long long)blockIdx.y * Cq + cc * 4);
  *reinterpret_cast<f32x4*>(out + (G2.off + p) * (4 * Cq) + c4 * 4) = v;
}
void fpn_concat(hipStream_t st, const float* p5, const float* p4, const float* p3, const float* p2, const ImgGeom* g5,
                const ImgGeom* g4, const ImgGeom* g3, const ImgGeom* g2, int n_img, long long max_pix, int Cq,
                float* out, const float* const* scales) {
  if (n_img <= 0) return;
  long long total = max_pix * Cq;  // 4 levels * Cq/4 groups
  const float* nul = nullptr;
  RT_LAUNCH(k_fpn_concat, dim3((unsigned)((total + 255) / 256), n_img), dim3(256), 0, st, p5, p4, p3, p2, g5,
                     g4, g3, g2, Cq, out, scales ? scales[0] : nul, scales ? scales[1] : nul, scales ? scales[2] : nul,
                     scales ? scales[3] : nul);
}

// DB head tail. w1 [24][24][2][2] (cin, cout, dy, dx), w2 [24][1][2][2].
__global__ __launch_bounds__(256) void k_db_head_tail(const float* __restrict__ x, const ImgGeom* __restrict__ gin,
                                                      const ImgGeom* __restrict__ gout, const float* __restrict__ w1,
                                                      const float* __restrict__ b1, const float* __restrict__ w2,
                                                      const float* __restrict__ b2, float* __restrict__ out) {
  __shared__ float sw1[4 * 24 * 24];  // [pos][ci][co]
  __shared__ float sw2[4 * 24];       // [pos][co]
  __shared__ float sb1[24];
  for (int i = threadIdx.x; i < 4 * 24 * 24; i += 256) {
    int pos = i / 576, ci = (i / 24) % 24, co = i % 24;
    sw1[i] = w1[(ci * 24 + co) * 4 + pos];
  }
  for (int i = threadIdx.x; i < 96; i += 256) { int pos = i / 24, co = i % 24; sw2[i] = w2[co * 4 + pos]; }
  if (threadIdx.x < 24) sb1[threadIdx.x] = b1[threadIdx.x];
  __syncthreads();
  const ImgGeom gi = gin[blockIdx.y], go = gout[blockIdx.y];
  long long p = (long long)blockIdx.x * 256 + threadIdx.x;
  if (p >= (long long)gi.H * gi.W) return;
  int iy = (int)(p / gi.W), ix = (int)(p % gi.W);
  float in[24];
  const f32x4* src = reinterpret_cast<const f32x4*>(x + (gi.off + p) * 24);
#pragma unroll
  for (int i = 0; i < 6; i++) { f32x4 v = src[i]; in[4 * i] = v[0]; in[4 * i + 1] = v[1]; in[4 * i + 2] = v[2]; in[4 * i + 3] = v[3]; }
  const float bias2 = b2[0];
  float o[4][4];
#pragma unroll
  for (int pos1 = 0; pos1 < 4; pos1++) {
    float mid[24];
#pragma unroll
    for (int co = 0; co < 24; co++) mid[co] = sb1[co];
#pragma unroll
    for (int ci = 0; ci < 24; ci++)
#pragma unroll
      for (int co = 0; co < 24; co++) mid[co] = fmaf(in[ci], sw1[(pos1 * 24 + ci) * 24 + co], mid[co]);
#pragma unroll
    for (int pos2 = 0; pos2 < 4; pos2++) {
      float s = bias2;
#pragma unroll
      for (int co = 0; co < 24; co++) s = fmaf(fmaxf(mid[co], 0.f), sw2[pos2 * 24 + co], s);
      int oy = (pos1 >> 1) * 2 + (pos2 >> 1), ox = (pos1 & 1) * 2 + (pos2 & 1);
      o[oy][ox] = 1.0f / (1.0f + expf(-s));
    }
  }
#pragma unroll
  for (int r = 0; r < 4; r++) {
    f32x4 v = {o[r][0], o[r][1], o[r][2], o[r][3]};
    *reinterpret_cast<f32x4*>(out + go.off + (long long)(iy * 4 + r) * go.W + ix * 4) = v;
  }
}
// The same tail on the matrix cores (round 4).  Both transposed convs are per-pixel maps (2 x 2 kernels at stride 2 do not
// overlap): 24 -> 4 x 24 -> 4 x 4 values, 2688 MACs per input pixel, which the VALU form above pays as scalar FMAs (0.22 ms per 32
// pages for 0.3 GB of traffic).  v_mfma_f32_4x4x1_16B_f32 with the A-operand broadcast (as k_conv3_few: lane = pixel, D[i] =
// output 4 g + i): the pixel's own 24 channels are the B operands straight from its registers, the weights come from LDS
// ([position][output][k] rows, one ds_read_b128 per four k), the ReLU'd first-stage accumulators are the B operands of the second
// stage.  672 MFMAs per 64 pixels; loads and stores are whole contiguous runs per wave.
__global__ __launch_bounds__(256) void k_db_head_tail_mfma(const float* __restrict__ x, const ImgGeom* __restrict__ gin,
                                                           const ImgGeom* __restrict__ gout, const float* __restrict__ w1,
                                                           const float* __restrict__ b1, const float* __restrict__ w2,
                                                           const float* __restrict__ b2, float* __restrict__ out) {
  constexpr int RW = 28;
  __shared__ __attribute__((aligned(16))) float sw1[4 * 24 * RW];  // [pos][co][ci]
  __shared__ __attribute__((aligned(16))) float sw2[4 * RW];       // [pos2][co]
  __shared__ __attribute__((aligned(16))) float sb1[24];
  for (int i = threadIdx.x; i < 4 * 24 * 24; i += 256) {
    const int pos = i / 576, co = (i / 24) % 24, ci = i % 24;
    sw1[(pos * 24 + co) * RW + ci] = w1[(ci * 24 + co) * 4 + pos];
  }
  for (int i = threadIdx.x; i < 96; i += 256) { const int pos = i / 24, co = i % 24; sw2[pos * RW + co] = w2[co * 4 + pos]; }
  if (threadIdx.x < 24) sb1[threadIdx.x] = b1[threadIdx.x];
  const ImgGeom gi = gin[blockIdx.y], go = gout[blockIdx.y];
  const long long npix = (long long)gi.H * gi.W;
  const long long p = (long long)blockIdx.x * 256 + threadIdx.x;
  const bool valid = p < npix;
  const long long pc = valid ? p : npix - 1;
  f32x4 in[6];
  const f32x4* src = reinterpret_cast<const f32x4*>(x + (gi.off + pc) * 24);
#pragma unroll
  for (int i = 0; i < 6; i++) in[i] = src[i];
  const float bias2 = b2[0];
  __syncthreads();
  if ((long long)blockIdx.x * 256 + (threadIdx.x & ~63) >= npix) return;   // whole wave past the image (after the only barrier)
  const int lane = threadIdx.x & 63;
  const float* w1r = sw1 + (lane < 24 ? lane : 0) * RW;
  const float* w2r = sw2 + (lane < 4 ? lane : 0) * RW;
  f32x4 bq[6];
#pragma unroll
  for (int g = 0; g < 6; g++) bq[g] = *reinterpret_cast<const f32x4*>(sb1 + g * 4);
  f32x4 o[4];
#pragma unroll
  for (int pos1 = 0; pos1 < 4; pos1++) {
    f32x4 mid[6];
#pragma unroll
    for (int g = 0; g < 6; g++) mid[g] = bq[g];
#pragma unroll
    for (int kk = 0; kk < 6; kk++) {
      const f32x4 wv = *reinterpret_cast<const f32x4*>(w1r + pos1 * 24 * RW + kk * 4);
#pragma unroll
      for (int e = 0; e < 4; e++) {
        mid[0] = __builtin_amdgcn_mfma_f32_4x4x1f32(wv[e], in[kk][e], mid[0], 4, 0, 0);
        mid[1] = __builtin_amdgcn_mfma_f32_4x4x1f32(wv[e], in[kk][e], mid[1], 4, 1, 0);
        mid[2] = __builtin_amdgcn_mfma_f32_4x4x1f32(wv[e], in[kk][e], mid[2], 4, 2, 0);
        mid[3] = __builtin_amdgcn_mfma_f32_4x4x1f32(wv[e], in[kk][e], mid[3], 4, 3, 0);
        mid[4] = __builtin_amdgcn_mfma_f32_4x4x1f32(wv[e], in[kk][e], mid[4], 4, 4, 0);
        mid[5] = __builtin_amdgcn_mfma_f32_4x4x1f32(wv[e], in[kk][e], mid[5], 4, 5, 0);
      }
    }
    f32x4 s = {bias2, bias2, bias2, bias2};   // the four second-stage positions of this first-stage position
#pragma unroll
    for (int kk = 0; kk < 6; kk++) {
      const f32x4 wv = *reinterpret_cast<const f32x4*>(w2r + kk * 4);
#pragma unroll
      for (int e = 0; e < 4; e++) s = __builtin_amdgcn_mfma_f32_4x4x1f32(wv[e], fmaxf(mid[kk][e], 0.f), s, 4, 0, 0);
    }
#pragma unroll
    for (int e = 0; e < 4; e++) o[pos1][e] = __builtin_amdgcn_rcpf(1.0f + __expf(-s[e]));   // act_apply's sigmoid (v_exp + v_rcp, <= 2 ulp)
  }
  if (!valid) return;
  const int iy = (int)(p / gi.W), ix = (int)(p - (long long)iy * gi.W);
#pragma unroll
  for (int r = 0; r < 4; r++) {   // output row r of the pixel's 4 x 4 block: first-stage row r >> 1, second-stage row r & 1
    const int pa = (r >> 1) * 2, sb = (r & 1) * 2;
    const f32x4 v = {o[pa][sb], o[pa][sb + 1], o[pa + 1][sb], o[pa + 1][sb + 1]};
    *reinterpret_cast<f32x4*>(out + go.off + (long long)(iy * 4 + r) * go.W + ix * 4) = v;
  }
}
static const int g_tail_mfma = getenv("RT_TAIL_MFMA") ? atoi(getenv("RT_TAIL_MFMA")) : 1;   // A/B: 0 = the VALU form

void db_head_tail(hipStream_t st, const float* x, const ImgGeom* gin, const ImgGeom* gout, int n_img, long long max_pix,
                  const float* w1, const float* b1, const float* w2, const float* b2, float* out) {
  if (n_img <= 0) return;
  if (g_tail_mfma) {
    RT_LAUNCH(k_db_head_tail_mfma, dim3((unsigned)((max_pix + 255) / 256), n_img), dim3(256), 0, st, x, gin, gout, w1, b1, w2, b2, out);
    return;
  }
  RT_LAUNCH(k_db_head_tail, dim3((unsigned)((max_pix + 255) / 256), n_img), dim3(256), 0, st, x, gin, gout, w1,
                     b1, w2, b2, out);
}

// ---------------------------------------------------------------------------
// Pooling
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_avgpool_3x2(const float* __restrict__ x, const ImgGeom* __restrict__ gin,
                                                     const ImgGeom* __restrict__ gout, int Cp, float* __restrict__ y,
                                                     int ldy) {
  const ImgGeom gi = gin[blockIdx.y], go = gout[blockIdx.y];
  const int C4 = Cp >> 2;
  long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long long)go.H * go.W * C4) return;
  int c4 = (int)(idx % C4);
  long long p = idx / C4;
  int oy = (int)(p / go.W), ox = (int)(p % go.W);
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int dy = 0; dy < 3; dy++)
#pragma unroll
    for (int dx = 0; dx < 2; dx++)
      s += *reinterpret_cast<const f32x4*>(x + (gi.off + (long long)(oy * 3 + dy) * gi.W + ox * 2 + dx) * Cp + c4 * 4);
  s *= (1.0f / 6.0f);
  *reinterpret_cast<f32x4*>(y + (go.off + p) * ldy + c4 * 4) = s;
}
void avgpool_3x2(hipStream_t st, const float* x, const ImgGeom* gin, const ImgGeom* gout, int n_img, long long max_pix,
                 int Cp, float* y, int ldy) {
  if (n_img <= 0) return;
  long long total = max_pix * (Cp / 4);
  RT_LAUNCH(k_avgpool_3x2, dim3((unsigned)((total + 255) / 256), n_img), dim3(256), 0, st, x, gin, gout, Cp, y,
                     ldy);
}
__global__ __launch_bounds__(256) void k_maxpool_2x2(const float* __restrict__ x, const ImgGeom* __restrict__ gin,
                                                     const ImgGeom* __restrict__ gout, int Cp, float* __restrict__ y) {
  const ImgGeom gi = gin[blockIdx.y], go = gout[blockIdx.y];
  const int C4 = Cp >> 2;
  long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long long)go.H * go.W * C4) return;
  int c4 = (int)(idx % C4);
  long long p = idx / C4;
  int oy = (int)(p / go.W), ox = (int)(p % go.W);
  f32x4 m;
  bool first = true;
#pragma unroll
  for (int dy = 0; dy < 2; dy++)
#pragma unroll
    for (int dx = 0; dx < 2; dx++) {
      f32x4 v = *reinterpret_cast<const f32x4*>(x + (gi.off + (long long)(oy * 2 + dy) * gi.W + ox * 2 + dx) * Cp + c4 * 4);
      if (first) { m = v; first = false; }
      else { for (int j = 0; j < 4; j++) m[j] = fmaxf(m[j], v[j]); }
    }
  *reinterpret_cast<f32x4*>(y + (go.off + p) * Cp + c4 * 4) = m;
}
void maxpool_2x2(hipStream_t st, const float* x, const ImgGeom* gin, const ImgGeom* gout, int n_img, long long max_pix,
                 int Cp, float* y) {
  if (n_img <= 0) return;
  long long total = max_pix * (Cp / 4);
  RT_LAUNCH(k_maxpool_2x2, dim3((unsigned)((total + 255) / 256), n_img), dim3(256), 0, st, x, gin, gout, Cp, y);
}

// ---------------------------------------------------------------------------
// SVTR pieces
// ---------------------------------------------------------------------------
// one wavefront per row; wave-shuffle reductions (two-pass mean / variance like F.layer_norm)
__global__ __launch_bounds__(256) void k_add_layernorm(const float* __restrict__ x, const float* __restrict__ r,
                                                       long long rows, int C, const float* __restrict__ g,
                                                       const float* __restrict__ beta, float eps,
                                                       float* __restrict__ y) {
  long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  int lane = threadIdx.x & 63;
  if (row >= rows) return;
  float v[4];  // C <= 256
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 4; i++) {
    int c = lane + 64 * i;
    float t = 0.f;
    if (c < C) { t = x[row * C + c]; if (r) t += r[row * C + c]; }
    v[i] = t; s += t;
  }
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  float mean = s / (float)C;
  float ss = 0.f;
#pragma unroll
  for (int i = 0; i < 4; i++) { int c = lane + 64 * i; if (c < C) { float d = v[i] - mean; ss = fmaf(d, d, ss); } }
  for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o);
  float rstd = 1.0f / sqrtf(ss / (float)C + eps);
#pragma unroll
  for (int i = 0; i < 4; i++) {
    int c = lane + 64 * i;
    if (c < C) y[row * C + c] = fmaf((v[i] - mean) * rstd, g[c], beta[c]);
  }
}
void add_layernorm(hipStream_t st, const float* x, const float* r, long long rows, int C, const float* g,
                   const float* beta, float eps, float* y) {
  if (rows <= 0) return;
  if (C > 256) throw RtError(8, "add_layernorm: C > 256");
  RT_LAUNCH(k_add_layernorm, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, x, r, rows, C, g, beta, eps, y);
}

// block = (query chunk of 64, head, image); K/V of the head streamed through LDS in tiles of 64 keys
template <int HD>
__global__ __launch_bounds__(64) void k_attention(const float* __restrict__ qkv, const ImgGeom* __restrict__ geom,
                                                  int heads, float* __restrict__ out) {
  const ImgGeom g = geom[blockIdx.z];
  const int T = g.H * g.W, C = heads * HD, head = blockIdx.y;
  const int t = blockIdx.x * 64 + threadIdx.x;
  if (blockIdx.x * 64 >= T) return;
  __shared__ float ks[64 * HD], vs[64 * HD];
  const float scale = 1.0f / sqrtf((float)HD);
  float qv[HD], acc[HD];
  const bool valid = t < T;
#pragma unroll
  for (int d = 0; d < HD; d++) {
    qv[d] = valid ? qkv[(g.off + t) * 3 * C + head * HD + d] * scale : 0.f;
    acc[d] = 0.f;
  }
  float m = -INFINITY, l = 0.f;
  for (int k0 = 0; k0 < T; k0 += 64) {
    int kn = min(64, T - k0);
    __syncthreads();
    for (int i = threadIdx.x; i < kn * HD; i += 64) {
      int kk = i / HD, d = i % HD;
      ks[i] = qkv[(g.off + k0 + kk) * 3 * C + C + head * HD + d];
      vs[i] = qkv[(g.off + k0 + kk) * 3 * C + 2 * C + head * HD + d];
    }
    __syncthreads();
    for (int kk = 0; kk < kn; kk++) {
      float s = 0.f;
#pragma unroll
      for (int d = 0; d < HD; d++) s = fmaf(qv[d], ks[kk * HD + d], s);
      float mn = fmaxf(m, s);
      float corr = expf(m - mn), p = expf(s - mn);
      l = l * corr + p;
#pragma unroll
      for (int d = 0; d < HD; d++) acc[d] = fmaf(acc[d], corr, p * vs[kk * HD + d]);
      m = mn;
    }
  }
  if (valid) {
    float inv = 1.0f / l;
#pragma unroll
    for (int d = 0; d < HD; d++) out[(g.off + t) * C + head * HD + d] = acc[d] * inv;
  }
}
// Round 5: one workgroup per text line.  k_attention above is one wave per (64 queries, head, line): a line of T ~ 50 tokens
// gives 8 one-wave blocks that each gather their head's K / V with 4-byte loads at a 1440-byte stride (24 scattered load
// instructions per thread in a dependent loop) -- 140 us per launch for 1.2 GFLOP.  Here the 256 threads of a line's workgroup
// stage a 64-key tile of K and V for ALL heads with 16-byte loads of whole rows (the three projections of a token are 1440
// contiguous bytes), re-laid per head into 16-float LDS rows, and a thread owns one (query, head) pair: a key costs 8
// ds_read_b128 (lanes of a wave are consecutive queries of one head: broadcast reads) instead of 30 ds_read_b32.  Same keys in
// the same order through the same online-softmax recurrence; exp by v_exp_f32 (__expf) where k_attention calls expf: equal to
// ~2 ulp of the exponentials, NOT bit-identical (RT_ATT_LINE=0/1 is an fp32-tolerance A/B).
template <int HD>
__global__ __launch_bounds__(256) void k_attention_line(const float* __restrict__ qkv, const ImgGeom* __restrict__ geom,
                                                        int heads, float* __restrict__ out) {
  static_assert(HD <= 16, "a head's row is padded to 16 floats");
  extern __shared__ __attribute__((aligned(16))) float att_lds[];
  const ImgGeom g = geom[blockIdx.x];
  const int T = g.H * g.W, C = heads * HD;
  if (T <= 0) return;
  float* ks = att_lds;                       // [64 keys][heads][16]
  float* vs = att_lds + 64 * heads * 16;
  const int tid = threadIdx.x;
  const float scale = 1.0f / sqrtf((float)HD);
  const int pairs = T * heads;
  const int C4 = C / 4;                      // 16-byte chunks of one projection of a token
  for (int p0 = 0; p0 < pairs; p0 += 256) {
    const int p = p0 + tid;
    const bool valid = p < pairs;
    const int head = valid ? p / T : 0, t = valid ? p - head * T : 0;
    float qv[HD], acc[HD];
#pragma unroll
    for (int d = 0; d < HD; d++) {
      qv[d] = valid ? qkv[(g.off + t) * 3 * C + head * HD + d] * scale : 0.f;
      acc[d] = 0.f;
    }
    float m = -INFINITY, l = 0.f;
    for (int k0 = 0; k0 < T; k0 += 64) {
      const int kn = min(64, T - k0);
      __syncthreads();
      // stage: chunk j of key kk (K and V): 4 consecutive channels c .. c + 3 -> [kk][c / HD][c % HD]
      for (int i = tid; i < kn * C4; i += 256) {
        const int kk = i / C4, j = i - kk * C4;
        const float* row = qkv + (g.off + k0 + kk) * 3 * C;
        const f32x4 kq = *reinterpret_cast<const f32x4*>(row + C + 4 * j);
        const f32x4 vq = *reinterpret_cast<const f32x4*>(row + 2 * C + 4 * j);
#pragma unroll
        for (int e = 0; e < 4; e++) {
          const int c = 4 * j + e, hh = c / HD, d = c - hh * HD;
          ks[(kk * heads + hh) * 16 + d] = kq[e];
          vs[(kk * heads + hh) * 16 + d] = vq[e];
        }
      }
      __syncthreads();
      for (int kk = 0; kk < kn; kk++) {
        const f32x4* kr = reinterpret_cast<const f32x4*>(ks + (kk * heads + head) * 16);
        const f32x4* vr = reinterpret_cast<const f32x4*>(vs + (kk * heads + head) * 16);
        float kv[16], vv[16];
#pragma unroll
        for (int c4 = 0; c4 < 4; c4++) {
          const f32x4 a = kr[c4], b = vr[c4];
#pragma unroll
          for (int e = 0; e < 4; e++) { kv[4 * c4 + e] = a[e]; vv[4 * c4 + e] = b[e]; }
        }
        float sdot = 0.f;
#pragma unroll
        for (int d = 0; d < HD; d++) sdot = fmaf(qv[d], kv[d], sdot);
        // (v_exp_f32 forms: 2 instructions instead of ~15 each -- the loop is VALU-bound: 27 M (query, key) pairs per step)
        const float mn = fmaxf(m, sdot);
        const float corr = __expf(m - mn), pe = __expf(sdot - mn);
        l = l * corr + pe;
#pragma unroll
        for (int d = 0; d < HD; d++) acc[d] = fmaf(acc[d], corr, pe * vv[d]);
        m = mn;
      }
    }
    if (valid) {
      const float inv = 1.0f / l;
#pragma unroll
      for (int d = 0; d < HD; d++) out[(g.off + t) * C + head * HD + d] = acc[d] * inv;
    }
  }
}
// Round 6: the same attention on the matrix pipe (v_mfma_f32_16x16x4_f32), lines of <= 128 tokens (longer ones keep
// k_attention_line).  A workgroup per line, two passes of four heads (K and V of four heads x <= 128 keys = 64 KB of LDS as
// [key][head][16 floats], slot 15 zero), wave w = head 4 pass + w.  Per (16-query, 16-key) tile:
//   S^T = K_tile Q^T      4 MFMAs: lane (r, q4) feeds K[key r][4 q4 .. + 3] (one ds_read_b128) and Q[query r][4 q4 .. + 3] -- MFMA
//                         step s pairs channel 4 q4 + s of both operands, any pairing is a dot product -- and receives
//                         S[keys 4 q4 .. + 3][query r];
//   online softmax        per query = per lane column: tile maximum over the lane's four keys and the three other lane rows of
//                         its column (two shuffles), one correction factor and four exponentials per lane;
//   O^T += V^T P          4 MFMAs: P is the MFMA's pixel operand AS IT STANDS in the accumulator layout (step e takes element e:
//                         key 4 q4 + e), V^T comes from LDS with the same key order; lane (r, q4) holds O[query r][4 q4 .. + 3].
// The per-lane partial sums of the softmax denominator meet once at the end.  Same mathematics as k_attention_line in another
// summation order (fp32 tolerance; tests/test_gpu_parity.py::test_rec_net).
template <int HD>
__global__ __launch_bounds__(512) void k_attention_mfma(const float* __restrict__ qkv, const ImgGeom* __restrict__ geom,
                                                        int heads, float* __restrict__ out) {
  static_assert(HD <= 16, "a head's row is padded to 16 floats");
  extern __shared__ __attribute__((aligned(16))) float att_lds[];
  const ImgGeom g = geom[blockIdx.x];
  const int T = g.H * g.W, C = heads * HD;
  if (T <= 0) return;
  const int Tp = (T + 15) & ~15;
  // [Tp keys][4 heads x 16 floats + 4 pad]: 68 floats per key (padding the key rows to 68 floats against the fragment reads' bank conflicts measured
  // slower: 0.069 vs 0.061 ms per launch)
  constexpr int KP = 64;
  float* ks = att_lds;
  float* vs = att_lds + (size_t)Tp * KP;
  // 8 waves: wave = head of the pass (w & 3) x half of the query-tile pairs (w >> 2): a long line's chain per wave halves
  const int tid = threadIdx.x, lane = tid & 63, wave = (tid >> 6) & 3, qhalf = tid >> 8, r = lane & 15, q4 = lane >> 4;
  const float scale = 1.0f / sqrtf((float)HD);
  const int qt = Tp >> 4;
  // slot 15 of every row and the rows of keys >= T stay zero for the whole kernel (the staging below writes real entries only)
  for (int i = tid; i < Tp * (2 * KP / 4); i += 512) reinterpret_cast<f32x4*>(att_lds)[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int pass = 0; pass * 4 < heads; pass++) {
    const int h0 = pass * 4, nh = min(4, heads - h0);
    __syncthreads();   // (the zero fill / the previous pass's readers are done)
    // stage K and V of the pass's heads: their channels are nh * HD contiguous floats of a token's K / V projection, fetched as
    // 16-byte chunks (a token's three projections are 3 C contiguous floats; h0 * HD * 4 bytes is a multiple of 16 for HD = 15,
    // h0 = 0 / 4) and scattered to [key][head][d]
    const int nch = (nh * HD + 3) >> 2;
    for (int i = tid; i < T * nch; i += 512) {
      const int kk = i / nch, j = i - kk * nch;
      const float* row = qkv + (g.off + kk) * 3 * C + h0 * HD + 4 * j;
      const f32x4 kq = *reinterpret_cast<const f32x4*>(row + C), vq = *reinterpret_cast<const f32x4*>(row + 2 * C);
#pragma unroll
      for (int e = 0; e < 4; e++) {
        const int c = 4 * j + e, hh = c / HD, d = c - hh * HD;
        if (hh < nh) { ks[kk * KP + hh * 16 + d] = kq[e]; vs[kk * KP + hh * 16 + d] = vq[e]; }
      }
    }
    __syncthreads();
    if (wave >= nh) continue;
    const int head = h0 + wave;
    // two query tiles at a time: their chains (MFMA -> shuffles -> exponentials -> MFMA, serial over the key tiles) are independent
    // and interleave in the wave's instruction stream
    for (int qi = 2 * qhalf; qi < qt; qi += 4) {
      int tq[2];
      f32x4 qf[2], o[2];
      float m[2], l[2];
#pragma unroll
      for (int u = 0; u < 2; u++) {
        tq[u] = (qi + u) * 16 + r;          // this lane's query (column of every tile)
        qf[u] = f32x4{0.f, 0.f, 0.f, 0.f}; o[u] = qf[u]; m[u] = -INFINITY; l[u] = 0.f;
        if (tq[u] < T) {
          const float* qrow = qkv + (g.off + tq[u]) * 3 * C + head * HD + 4 * q4;
#pragma unroll
          for (int e = 0; e < 4; e++)
            if (4 * q4 + e < HD) qf[u][e] = qrow[e] * scale;
        }
      }
      for (int ki = 0; ki < qt; ki++) {
        const f32x4 kf = *reinterpret_cast<const f32x4*>(ks + (ki * 16 + r) * KP + wave * 16 + 4 * q4);
        float ve[4];
#pragma unroll
        for (int e = 0; e < 4; e++) ve[e] = vs[(ki * 16 + 4 * q4 + e) * KP + wave * 16 + r];   // V^T[channel r][key 4 q4 + e]
        const int kb = ki * 16 + 4 * q4;
        f32x4 sv[2], pv[2];
#pragma unroll
        for (int u = 0; u < 2; u++) {
          sv[u] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int st = 0; st < 4; st++) sv[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[st], qf[u][st], sv[u], 0, 0, 0);
        }
#pragma unroll
        for (int u = 0; u < 2; u++) {
#pragma unroll
          for (int e = 0; e < 4; e++)
            if (kb + e >= T) sv[u][e] = -INFINITY;   // keys beyond T do not take part
          float tm = fmaxf(fmaxf(sv[u][0], sv[u][1]), fmaxf(sv[u][2], sv[u][3]));
          tm = fmaxf(tm, __shfl_xor(tm, 16, 64));
          tm = fmaxf(tm, __shfl_xor(tm, 32, 64));
          const float mn = fmaxf(m[u], tm);     // (finite: key 0 of tile 0 exists)
          const float corr = __expf(m[u] - mn);
#pragma unroll
          for (int e = 0; e < 4; e++) pv[u][e] = __expf(sv[u][e] - mn);
          l[u] = l[u] * corr + ((pv[u][0] + pv[u][1]) + (pv[u][2] + pv[u][3]));
          o[u] *= corr;
          m[u] = mn;
        }
#pragma unroll
        for (int e = 0; e < 4; e++)
#pragma unroll
          for (int u = 0; u < 2; u++) o[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(ve[e], pv[u][e], o[u], 0, 0, 0);
      }
#pragma unroll
      for (int u = 0; u < 2; u++) {
        l[u] += __shfl_xor(l[u], 16, 64);
        l[u] += __shfl_xor(l[u], 32, 64);
        if (tq[u] < T) {
          const float inv = 1.0f / l[u];
          float* orow = out + (g.off + tq[u]) * C + head * HD + 4 * q4;
#pragma unroll
          for (int e = 0; e < 4; e++)
            if (4 * q4 + e < HD) orow[e] = o[u][e] * inv;
        }
      }
    }
  }
}
int g_attention_line = getenv("RT_ATT_LINE") ? atoi(getenv("RT_ATT_LINE")) : 1;   // A/B: 0 = k_attention (one wave per 64 queries and head)
void attention(hipStream_t st, const float* qkv, const ImgGeom* geom, int n_img, int maxT, int heads, int hd,
               float* out) {
  if (n_img <= 0) return;
  if (hd != 15) throw RtError(8, "attention: head dim must be 15");
  static const int att_mfma = getenv("RT_ATT_MFMA") ? atoi(getenv("RT_ATT_MFMA")) : 1;   // A/B: 0 = k_attention_line for every line
  if (att_mfma && g_attention_line && maxT <= 128 && heads <= 8) {
    const size_t lds = (size_t)2 * ((maxT + 15) & ~15) * 64 * 4;   // K and V of four heads: 64 KB at 128 tokens
    allow_big_lds((const void*)k_attention_mfma<15>, 64 * 1024);
    RT_LAUNCH(k_attention_mfma<15>, dim3((unsigned)n_img), dim3(512), lds, st, qkv, geom, heads, out);
    return;
  }
  if (g_attention_line && (heads * hd) % 4 == 0 && heads <= 8) {
    const size_t lds = (size_t)2 * 64 * heads * 16 * 4;   // 64 KB at 8 heads
    allow_big_lds((const void*)k_attention_line<15>, 64 * 1024);
    RT_LAUNCH(k_attention_line<15>, dim3((unsigned)n_img), dim3(256), lds, st, qkv, geom, heads, out);
    return;
  }
  RT_LAUNCH(k_attention<15>, dim3((maxT + 63) / 64, heads, n_img), dim3(64), 0, st, qkv, geom, heads, out);
}

__global__ __launch_bounds__(256) void k_copy_channels(const float* __restrict__ src, int lds, long long rows, int C4,
                                                       float* __restrict__ dst, int ldd, int coff) {
  long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= rows * C4) return;
  long long row = idx / C4;
  int c4 = (int)(idx % C4);
  *reinterpret_cast<f32x4*>(dst + row * ldd + coff + c4 * 4) = *reinterpret_cast<const f32x4*>(src + row * lds + c4 * 4);
}
void copy_channels(hipStream_t st, const float* src, int lds, long long rows, int C, float* dst, int ldd, int coff) {
  if (rows <= 0) return;
  long long total = rows * (C / 4);
  RT_LAUNCH(k_copy_channels, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, src, lds, rows, C / 4, dst,
                     ldd, coff);
}

// block per row; max, sum(exp), then write (MODE 0) or argmax + max prob (MODE 1)
template <int MODE>
__global__ __launch_bounds__(256) void k_softmax_rows(const float* __restrict__ in, int ld, long long rows, int C,
                                                      float* __restrict__ out, int* __restrict__ idx_out,
                                                      float* __restrict__ prob_out) {
  __shared__ float sv[4];
  __shared__ int si[4];
  __shared__ float bc[2];
  const long long row = blockIdx.x;
  const float* x = in + row * ld;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float m = -INFINITY;
  int mi = 0x7fffffff;
  for (int c = threadIdx.x; c < C; c += 256) {
    float v = x[c];
    if (v > m) { m = v; mi = c; }
  }
  for (int o = 32; o > 0; o >>= 1) {
    float om = __shfl_xor(m, o);
    int oi = __shfl_xor(mi, o);
    if (om > m || (om == m && oi < mi)) { m = om; mi = oi; }
  }
  if (lane == 0) { sv[wave] = m; si[wave] = mi; }
  __syncthreads();
  if (threadIdx.x == 0) {
    float bm = sv[0]; int bi = si[0];
    for (int w = 1; w < 4; w++) if (sv[w] > bm || (sv[w] == bm && si[w] < bi)) { bm = sv[w]; bi = si[w]; }
    bc[0] = bm; si[0] = bi;
  }
  __syncthreads();
  const float gm = bc[0];
  const int gi = si[0];
  if (MODE == 2) {  // argmax + the maximum itself (rows that are already probabilities)
    if (threadIdx.x == 0) { idx_out[row] = gi; prob_out[row] = gm; }
    return;
  }
  float s = 0.f;
  for (int c = threadIdx.x; c < C; c += 256) s += expf(x[c] - gm);
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  __syncthreads();
  if (lane == 0) sv[wave] = s;
  __syncthreads();
  const float tot = (sv[0] + sv[1]) + (sv[2] + sv[3]);
  if (MODE == 0) {
    for (int c = threadIdx.x; c < C; c += 256) out[row * C + c] = expf(x[c] - gm) / tot;
  } else if (threadIdx.x == 0) {
    idx_out[row] = gi;
    prob_out[row] = 1.0f / tot;  // exp(0) / sum
  }
}
void softmax_rows(hipStream_t st, const float* in, int ld, long long rows, int C, float* out) {
  if (rows <= 0) return;
  RT_LAUNCH(k_softmax_rows<0>, dim3((unsigned)rows), dim3(256), 0, st, in, ld, rows, C, out, (int*)nullptr,
                     (float*)nullptr);
}
void argmax_prob_rows(hipStream_t st, const float* in, int ld, long long rows, int C, int* idx, float* prob) {
  if (rows <= 0) return;
  RT_LAUNCH(k_softmax_rows<1>, dim3((unsigned)rows), dim3(256), 0, st, in, ld, rows, C, (float*)nullptr, idx,
                     prob);
}

void argmax_rows(hipStream_t st, const float* in, int ld, long long rows, int C, int* idx, float* maxval) {
  if (rows <= 0) return;
  RT_LAUNCH(k_softmax_rows<2>, dim3((unsigned)rows), dim3(256), 0, st, in, ld, rows, C, (float*)nullptr, idx,
                     maxval);
}

}  // namespace nn
}  // namespace rt
