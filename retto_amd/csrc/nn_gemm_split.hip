// Split-bf16 form of the wide fp32 GEMMs (round 6): fp32-width arithmetic on the bf16 matrix pipe of gfx950.
//
// The 240- / 480-channel 1x1 convolutions of the recognition network (k_gemm32p's job; they replace ONNX Runtime's Conv at
// /root/reference/retto-core/src/worker/ort_worker.rs:211-220) run on v_mfma_f32_16x16x4_f32 at 1/16 of the chip's bf16
// matrix rate.  An fp32 value x splits EXACTLY into three bf16 terms, x = h + m + l (h = bf16(x), m = bf16(x - h),
// l = bf16(x - h - m), round-to-nearest: 3 x 8 significant bits + signs cover the 24), every bf16 x bf16 product is exact
// in fp32, and v_mfma_f32_16x16x32_bf16 accumulates in fp32.  So
//     a * w  =  ah wh + ah wm + am wh + am wm + ah wl + al wh   + (am wl + al wm + al wl),
// and the six kept terms leave |error| <= 2^-26 |a w| per product (the dropped three), a quarter of the half-ulp an fp32
// accumulate rounds away anyway.  Six bf16 MFMAs do the work of sixteen fp32 ones: 2.7x the fp32 matrix rate, at which
// point the launch is bound by HBM (input once, output once).
//
//   * weights: split once on the device into three bf16 planes, packed in fragment order
//     [column block of 240][K slab of 32][16-channel tile][plane h, m, l][lane][8 bf16]  (1 KB per (tile, plane));
//   * pixels: fp32 rows go global -> LDS by buffer_load ... lds (each wave requests ITS OWN 32 rows: no workgroup barrier on
//     that operand, two K slabs of 32 in flight per wave), are read as fp32 fragments and split in registers (11 VALU
//     instructions per pair of values, issued between the MFMAs of the previous slab's second half);
//   * a wave owns 32 pixel rows x all 240 channels of the column block (120 accumulator registers), so every pixel value is
//     split once; 4 waves = 128 rows per workgroup, persistent over the row blocks, TWO workgroups per CU (80 KB of LDS
//     each).  Two independent workgroups, not one of eight waves: with one barrier domain the two waves of a SIMD leave
//     every barrier together and run the same program in lockstep -- both in their MFMAs, then both in their requests /
//     splits / stores -- and in-kernel stamps showed the SIMD's time to be the SUM of the two waves' streams (10.9 k
//     cycles per slab for 2 x 2880 of MFMAs at 1.99 GHz).  Two workgroups drift apart by themselves;
//   * the weight stream (45 KB per K slab) passes through a ring of THREE 15 KB group buffers (tiles 0..4 | 5..9 | 10..14),
//     one workgroup barrier per group placed before the group's last 12 MFMAs are issued.  Three, not two: vmcnt retires
//     in order, so a wait for a weight request forces every OLDER pixel request of the wave to have landed; with the
//     weights two groups ahead (and the pixels behind them) a pixel slab has more than a slab time to arrive;
//   * the previous tile's epilogue (bias / hardswish / LAB / 16-byte stores) rides in front of the next tile's first
//     MFMA steps, one 16-channel tile at a time (its accumulators restart from zero there).
#include "nn.h"
#include "nn_dev.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <mutex>
#include <string>

namespace rt {
namespace nn {

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
// LDS-DMA through a buffer resource: lane i writes 16 bytes at M0 + 16 i; source = base + per-lane offset + scalar offset; a
// lane whose per-lane offset is beyond the resource's range writes zeros.
// s_nop 4 first: this kernel spills scalar registers to VGPR lanes, and a descriptor word restored by v_readlane (a VALU write
// of an SGPR) right in front of the asm block needs 5 wait states before a VMEM instruction reads it -- the hazard recogniser
// does not see the buffer_load inside the block.  Without it: stale descriptor words, memory faults that came and went with
// the register allocation.
__device__ __forceinline__ void blds16(unsigned voff, __amdgpu_buffer_rsrc_t rs, unsigned lds_sgpr, unsigned soff) {
  asm volatile("s_nop 4\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %3 offen lds" ::"v"(voff), "s"(rs), "s"(lds_sgpr), "s"(soff) : "memory", "m0");
}
#pragma clang diagnostic pop
// 4-byte form: lane i writes one dword at M0 + 4 i (the squeeze-excite scale vectors of a slab: 2 images x 32 channels per instruction)
__device__ __forceinline__ void blds4(unsigned voff, __amdgpu_buffer_rsrc_t rs, unsigned lds_sgpr) {
  asm volatile("s_nop 4\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dword %0, %1, 0 offen lds" ::"v"(voff), "s"(rs), "s"(lds_sgpr) : "memory", "m0");
}
__device__ __forceinline__ unsigned lds_addr32(const void* p) { return (unsigned)(size_t)(__attribute__((address_space(3))) const char*)p; }
#pragma clang diagnostic pop
template <int OFF>
__device__ __forceinline__ u32x4 lds_read16u(unsigned byte_addr) {   // address + compile-time offset in the instruction
  static_assert(OFF >= 0 && OFF < 65536, "16-bit offset field");
  u32x4 v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(byte_addr), "n"(OFF));
  return v;
}
template <int N>
__device__ __forceinline__ void lgkm_wait() {   // leaves the newest N LDS operations in flight and pins the order around it
  asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory");
  __builtin_amdgcn_sched_barrier(0);
}
__device__ __forceinline__ int lane_id() { return (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }

__device__ __forceinline__ unsigned cvt_pk_bf16(float a, float b) {   // v_cvt_pk_bf16_f32: a -> low half, b -> high half (RNE)
  const f32x2 v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}
// (a, b) -> three packed bf16 pairs with a = ah + am + al exactly (and b likewise): the subtractions are exact
__device__ __forceinline__ void split_pair(float a, float b, unsigned& h, unsigned& m, unsigned& l) {
  h = cvt_pk_bf16(a, b);
  const float ra = a - __uint_as_float(h << 16), rb = b - __uint_as_float(h & 0xffff0000u);
  m = cvt_pk_bf16(ra, rb);
  const float sa = ra - __uint_as_float(m << 16), sb = rb - __uint_as_float(m & 0xffff0000u);
  l = cvt_pk_bf16(sa, sb);
}
__device__ __forceinline__ void split8(const f32x4 x0, const f32x4 x1, u32x4& h, u32x4& m, u32x4& l) {
  unsigned hh[4], mm[4], ll[4];
  split_pair(x0[0], x0[1], hh[0], mm[0], ll[0]);
  split_pair(x0[2], x0[3], hh[1], mm[1], ll[1]);
  split_pair(x1[0], x1[1], hh[2], mm[2], ll[2]);
  split_pair(x1[2], x1[3], hh[3], mm[3], ll[3]);
  h = u32x4{hh[0], hh[1], hh[2], hh[3]}; m = u32x4{mm[0], mm[1], mm[2], mm[3]}; l = u32x4{ll[0], ll[1], ll[2], ll[3]};
}
template <bool OFF = false>
__device__ __forceinline__ f32x4 mfma_bf16(const u32x4 w, const u32x4 a, const f32x4 c) {
  if (OFF) { f32x4 d = c; d[0] += __uint_as_float(w[0] ^ a[0]); return d; }
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w), __builtin_bit_cast(bf16x8, a), c, 0, 0, 0);
}

}  // namespace

constexpr int S_BM = 128, S_BN = 240, S_NT = 15, S_NW = 4, S_NTHR = 256;
constexpr int S_GT = 5;                                         // tiles per weight group: a slab = 3 groups (0..4 | 5..9 | 10..14)
constexpr unsigned S_WSLAB = S_NT * 3 * 1024;                   // 46080 bytes of split weights per K slab and column block
constexpr unsigned S_WGRP = S_GT * 3 * 1024;                    // ring buffer of one group: 15 KB
constexpr unsigned S_AWAVE = 32 * 128;                          // a wave's 32 rows of one 32-deep fp32 slab
constexpr unsigned S_ASLOT = S_NW * S_AWAVE;                    // 16 KB
constexpr unsigned S_OFF_W = 2 * S_ASLOT, S_OFF_BIAS = S_OFF_W + 3 * S_WGRP;
constexpr int S_BIAS_MAX = 240;                                 // one column block's bias (restaged per tile)
constexpr unsigned S_OFF_TQ = S_OFF_BIAS + S_BIAS_MAX * 4;      // ids of the workgroup's tiles j, j + 1, ... (slot j & 3)
constexpr unsigned S_OFF_SC = S_OFF_TQ + 16;                     // ASC: scale vectors of two slabs in flight, [wave][slot][2 images][32] floats
constexpr size_t S_LDS = S_OFF_TQ + 16, S_LDS_ASC = S_OFF_SC + 2048;           // 32 KB pixels | 45 KB weights | bias | tile ids = 79824 bytes: two workgroups per CU

struct GemmSArgs {
  const float* A; const unsigned short* Ws; float* C;
  long long M;
  int lda, nslab, N, ldc, coff;
  int n_rb, n_cb;
  int dyn;           // 1: tile ids from the queue; 2 (debugging): the queue runs, the ids stay static
  // ASC (squeeze-excite scale folded into the pixel operand): row m of image i is multiplied by a_scale[i * ld_scale + k] before
  // it is split.  a_tab: per 256-row block {image of its first row, first row of the next image, of the one after} (stride 3,
  // what k_gemm32p+se reads) or per 128-row block {image, first row of the next image} (stride 2); every image has >= 128 rows.
  const float* a_scale; const int* a_tab; int ld_scale, n_img, tab_stride, K;
  unsigned* sched;   // [0] tiles handed out beyond the workgroups' first ones, [1] workgroups done; zero between launches
  Epilogue epi;
};

// fp32 packed weights [K / 32][Npad][32] -> split planes [cb][slab][nt][plane][lane][8]; lane (r, q) element e holds channel
// 16 nt + r, k = 32 slab + (e < 4 ? 4 q + e : 16 + 4 q + e - 4) -- the k order of the pixel fragments below
__global__ void k_split_pack(const float* __restrict__ Wp, int nslab, int Npad, int n_cb, unsigned short* __restrict__ out) {
  const long long total = (long long)n_cb * nslab * S_NT * 64;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int lane = (int)(i & 63);
    long long t = i >> 6;
    const int nt = (int)(t % S_NT); t /= S_NT;
    const int slab = (int)(t % nslab);
    const int cb = (int)(t / nslab);
    const int r = lane & 15, q = lane >> 4;
    const int n = cb * S_BN + nt * 16 + r;
    unsigned short* o = out + (((long long)(cb * nslab + slab) * S_NT + nt) * 3) * 512 + lane * 8;
    for (int e = 0; e < 8; e++) {
      const int k = e < 4 ? 4 * q + e : 16 + 4 * q + (e - 4);
      const float w = Wp[((long long)slab * Npad + n) * KC + k];
      const unsigned h = cvt_pk_bf16(w, 0.f) & 0xffffu;
      const float r1 = w - __uint_as_float(h << 16);
      const unsigned m = cvt_pk_bf16(r1, 0.f) & 0xffffu;
      const float r2 = r1 - __uint_as_float(m << 16);
      const unsigned l = cvt_pk_bf16(r2, 0.f) & 0xffffu;
      o[e] = (unsigned short)h; o[512 + e] = (unsigned short)m; o[1024 + e] = (unsigned short)l;
    }
  }
}

// DBG (timing experiments, wrong results): 1 no stores, 2 no pixel requests after the prologue, 4 no weight requests after it,
// 8 no split arithmetic, 16 no MFMAs
template <int ACT, int LAB, int DBG = 0, bool ASC = false>
__global__ __launch_bounds__(S_NTHR, 2) void k_gemm_split(const GemmSArgs g) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_s[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 15, q = lane >> 4;
  const unsigned lds_b = __builtin_amdgcn_readfirstlane(lds_addr32(smem_s));
  float* bias_l = reinterpret_cast<float*>(smem_s + S_OFF_BIAS);
  // (the bias of the column block whose epilogue is pending; restaged between two barriers of a tile's last slab when the
  //  next tile's column block differs)
  int bias_cb = -1;
  auto stage_bias = [&](int cb) __attribute__((always_inline)) {
    if (cb != bias_cb) { for (int i = tid; i < S_BIAS_MAX; i += S_NTHR) bias_l[i] = g.epi.bias ? g.epi.bias[cb * S_BN + i] : 0.f; bias_cb = cb; }
  };
  const int G = gridDim.x, n_tiles = g.n_rb * g.n_cb, nslab = g.nslab;
  const unsigned pitch = (unsigned)(g.lda * 4);

  // ---- request side ---------------------------------------------------------------------------------------------------------
  // pixels: piece p (8 rows) of the wave's 32 rows; lane i -> row 8 p + i / 8, physical 16-byte chunk i % 8 holds the logical
  // chunk (i % 8) ^ ((row >> 1) & 7); (row >> 1) & 7 = ((i >> 4) & 3) | ((p & 1) << 2): one offset for even, one for odd pieces
  const unsigned rq_row = (unsigned)(lane >> 3) * pitch;
  const unsigned rq_a0 = rq_row + (unsigned)((((lane & 7) ^ ((lane >> 4) & 3))) << 4);
  const unsigned rq_a1 = rq_row + (unsigned)((((lane & 7) ^ (((lane >> 4) & 3) | 4))) << 4);
  const unsigned rq_w = (unsigned)lane * 16u;
  // The descriptors of the tile a request stream is in are rebuilt when the stream enters the next tile, not per request (an
  // integer division and 64-bit address arithmetic on the scalar unit: ~100 instructions the wave issues instead of MFMAs).
  // Tiles are handed out dynamically: workgroup b starts with tile b, every further one comes from an atomic counter (block end
  // times of the static form: min 720, mean 840, max 970 us -- the CUs do not run at one speed).  Wave 0 fetches the id of the
  // workgroup's tile j + 1 in the first slab of tile j and publishes it through tileq[]; the request streams need it when they
  // leave tile j (its last two slabs), the MFMA side when tile j ends.
  // (explicit DS instructions: through a volatile generic pointer hipcc emits FLAT loads / stores, each followed by vmcnt(0) --
  //  every queue access would drain the wave's requests)
  const unsigned tq_addr = lds_b + S_OFF_TQ;
  auto tq_read = [&](int j) __attribute__((always_inline)) {
    int v;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(tq_addr + (unsigned)((j & 3) * 4)) : "memory");
    return (int)__builtin_amdgcn_readfirstlane(v);
  };
  auto tq_write = [&](int j, int v) __attribute__((always_inline)) {
    asm volatile("ds_write_b32 %0, %1" ::"v"(tq_addr + (unsigned)((j & 3) * 4)), "v"(v) : "memory");
  };
  if (tid == 0) tq_write(0, (int)blockIdx.x);
  int aq_j = 0, wq_j = 0;   // tile ordinals of the two request streams
  int aq_t = (int)blockIdx.x, aq_s = 0, aq_slot = 0;        // next pixel slab to request: tile, slab, ring slot
  int wq_t = (int)blockIdx.x, wq_h = 0, wq_buf = 0; unsigned wq_so = 0; // next weight group to request: tile, group 0 .. 3 nslab - 1, ring buffer, byte offset of the group
  // (measured and removed: walking a tile's K slabs in an order rotated by its row block, so that the workgroups of the chip do not
  //  all read the same 128 bytes of their 1-KB rows at the same time -- requests only, no MFMAs: 0.489 vs 0.501 ms, whole kernel
  //  0.966 vs 0.979: not channel camping)
  int aq_p = 0, wq_p = 0, wq_g = 0;   // slab of the next pixel / weight request, weight group within it
  bool dbg_pro = true;   // (DBG: requests of the prologue are always issued)
  auto a_desc = [&](int t) __attribute__((always_inline)) {
    const bool live = t < n_tiles;
    const int rb = live ? t / g.n_cb : 0;
    const long long m0 = (long long)rb * S_BM + 32 * wid;
    const unsigned rows_here = live ? (unsigned)max(0ll, min(32ll, g.M - m0)) : 0u;
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.A + (rows_here ? m0 * g.lda : 0)), 0, rows_here * pitch, 0x00020000);
  };
  auto w_desc = [&](int t) __attribute__((always_inline)) {
    const bool live = t < n_tiles;
    const int cb = live ? t % g.n_cb : 0;
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(g.Ws + (long long)cb * nslab * (S_WSLAB / 2)), 0, live ? 0x7fffffffu : 0u, 0x00020000);
  };
  __amdgpu_buffer_rsrc_t ars = a_desc(aq_t), wrs = w_desc(wq_t);
  // ASC: image of a tile's first row and the tile-relative row at which the next image begins (a 128-row tile meets at most two
  // images: every image has >= 128 rows).  The scale vectors of a slab (2 images x 32 channels = 256 bytes) are requested by
  // wave 0 with the slab's pixels, one 4-byte-per-lane instruction, into the slot of the pixel ring's parity.
  int sc_img = 0, sc_bnd[2] = {0x7fffffff, 0x7fffffff};   // of the request stream's tile; boundary by ring slot (read when that slab is split)
  auto tile_images = [&](int t, int* img, int* bnd) __attribute__((always_inline)) {
    *img = 0; *bnd = 0x7fffffff;
    if (ASC && t < n_tiles) {
      const int rb = t / g.n_cb;
      const long long m0 = (long long)rb * S_BM;
      if (g.tab_stride == 2) { *img = g.a_tab[2 * rb]; const long long b1 = g.a_tab[2 * rb + 1]; *bnd = (int)min(0x7fffffffll, max(0ll, b1 - m0)); }
      else {
        const int* e = g.a_tab + 3 * (rb >> 1);
        const long long b1 = e[1], b2 = e[2];
        *img = e[0] + (m0 >= b1 ? 1 : 0) + (m0 >= b2 ? 1 : 0);
        const long long nb = m0 < b1 ? b1 : (m0 < b2 ? b2 : 0x7fffffffll + m0);
        *bnd = (int)min(0x7fffffffll, nb - m0);
      }
    }
  };
  int sc_bnd_t = 0x7fffffff;
  if (ASC) tile_images(aq_t, &sc_img, &sc_bnd_t);
  const __amdgpu_buffer_rsrc_t srs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(ASC ? g.a_scale : g.A), 0, ASC ? (unsigned)g.n_img * (unsigned)g.ld_scale * 4u : 0u, 0x00020000);
  const unsigned sc_dst0 = lds_b + S_OFF_SC;
  const unsigned a_dst0 = lds_b + (unsigned)wid * S_AWAVE, w_dst0 = lds_b + S_OFF_W + (unsigned)wid * 1024u;
  // Requests go out ONE PIECE PER MFMA STEP, not as a burst behind the barrier: an LDS-DMA instruction holds the issuing wave for
  // 60-180 cycles, the two waves of a SIMD leave a barrier together, and seven requests each right there are ~1000 cycles in
  // which neither issues an MFMA (measured: each request stream removed took ~0.09 ms off a 0.94 ms launch).  The waves 0..3
  // issue their piece BEFORE the step's MFMAs, their SIMD partners 4..7 AFTER them: one's request under the other's MFMAs.
  // vmcnt bookkeeping at run time: vc_next / vc_later = vector-memory instructions issued after the last piece of the weight
  // group the next / the next-but-one barrier wait needs (that many may stay in flight at that wait).
  // (one running count of issued operations and snapshots of it: an add per operation instead of three -- a wave issues one scalar
  //  instruction per ~4 cycles, and in-kernel stamps put the ~15 scalar instructions around a request at 110 cycles per piece)
  int vc_ops = 0, vs_next = 0, vs_later = 0, vs_at = 0;   // vs_at: ... at the tile-queue atomic of wave 0
  auto vm_note = [&](int n) __attribute__((always_inline)) { vc_ops += n; };
  // destination / source offsets of the current request group, advanced when a group's last piece has been issued
  unsigned aq_dst = a_dst0, aq_so = 0, wq_dst = w_dst0;
  unsigned wq_sow = (unsigned)wid * 1024u;   // wq_so + this wave's first piece
  const unsigned pitch8 = 8u * pitch;
  auto a_piece = [&](auto ktag) __attribute__((always_inline)) {
    constexpr int k = decltype(ktag)::value;
    if (!(DBG & 2) || dbg_pro) { blds16(((k & 1) ? rq_a1 : rq_a0) + (unsigned)k * pitch8, ars, aq_dst + k * 1024u, aq_so); vm_note(1); }
    if (ASC && k == 0) {
      sc_bnd[aq_slot] = sc_bnd_t;
      {   // lane i: image sc_img + i / 32 (clamped), channel 32 slab + i % 32 (beyond K: out of range, zero).
          // EVERY wave requests its own copy into its own slots, so that -- like the pixels -- the wave's vmcnt is the only
          // hand-over.  (First form: wave 0 alone, one shared slot per ring parity, published by the group barriers.  After
          // the prologue's __syncthreads the four waves read slot 0 for slab 0 and wave 0 went straight on to request slab 2's
          // vectors INTO slot 0: with the session's other lanes running (cold instruction cache, contended memory) a wave could
          // still be in front of its prologue reads when that request landed -- one text line of a 1024-line C3 batch came out
          // different in ~3 % of the runs, tools/soak_split.py.  Four 256-byte requests per slab instead of one: +1 % on the launch.)
        const int kk = aq_p * 32 + (lane & 31);
        const int im = min(sc_img + (lane >> 5), g.n_img - 1);
        const unsigned off = kk < g.K ? (unsigned)(im * g.ld_scale + kk) * 4u : 0x80000000u;
        blds4(off, srs, sc_dst0 + (unsigned)(wid * 2 + aq_slot) * 256u); vm_note(1);
      }
    }
    if (k == 3) {
      aq_slot ^= 1;
      aq_p = aq_p + 1 == nslab ? 0 : aq_p + 1;
      if (++aq_s == nslab) { aq_s = 0; aq_j++; { const int q_ = (g.dyn & 8) ? tq_read(aq_j) : 0; aq_t = (g.dyn & 1) ? q_ : aq_t + G; } ars = a_desc(aq_t); aq_p = 0; if (ASC) tile_images(aq_t, &sc_img, &sc_bnd_t); }
      aq_dst = a_dst0 + (unsigned)aq_slot * S_ASLOT; aq_so = (unsigned)aq_p * 128u;
    }
    __builtin_amdgcn_sched_barrier(0);
  };
  // (a group has 15 pieces: wave w takes w, w + 4, w + 8, w + 12 -- wave 3 has three; its bookkeeping still runs at k == 3)
  auto w_piece = [&](auto ktag) __attribute__((always_inline)) {
    constexpr int k = decltype(ktag)::value;
    if ((!(DBG & 4) || dbg_pro) && (k < 3 || wid < 3)) { blds16(rq_w, wrs, wq_dst + k * 4096u, wq_sow + k * 4096u); vm_note(1); }
    if (k == 3) {
      vs_later = vc_ops;
      wq_buf = wq_buf == 2 ? 0 : wq_buf + 1;
      if (++wq_g == 3) { wq_g = 0; wq_p = wq_p + 1 == nslab ? 0 : wq_p + 1; }
      if (++wq_h == 3 * nslab) { wq_h = 0; wq_j++; { const int q_ = (g.dyn & 8) ? tq_read(wq_j) : 0; wq_t = (g.dyn & 1) ? q_ : wq_t + G; } wrs = w_desc(wq_t); wq_p = 0; wq_g = 0; }
      wq_so = (unsigned)wq_p * S_WSLAB + (unsigned)wq_g * S_WGRP;
      wq_sow = wq_so + (unsigned)wid * 1024u; wq_dst = w_dst0 + (unsigned)wq_buf * S_WGRP;
    }
    __builtin_amdgcn_sched_barrier(0);
  };
  auto a_issue = [&]() __attribute__((always_inline)) { a_piece(IntTag<0>{}); a_piece(IntTag<1>{}); a_piece(IntTag<2>{}); a_piece(IntTag<3>{}); };
  auto w_issue = [&]() __attribute__((always_inline)) { w_piece(IntTag<0>{}); w_piece(IntTag<1>{}); w_piece(IntTag<2>{}); w_piece(IntTag<3>{}); };
  // s_waitcnt vmcnt(n), n at run time (the instruction takes an immediate; a smaller n only waits for more)
  auto vm_wait_n = [&](int n) __attribute__((always_inline)) {
#define RT_VW(N) case N: asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory"); break;
    switch (min(n, 47)) {
      RT_VW(0) RT_VW(1) RT_VW(2) RT_VW(3) RT_VW(4) RT_VW(5) RT_VW(6) RT_VW(7) RT_VW(8) RT_VW(9) RT_VW(10) RT_VW(11) RT_VW(12) RT_VW(13) RT_VW(14) RT_VW(15)
      RT_VW(16) RT_VW(17) RT_VW(18) RT_VW(19) RT_VW(20) RT_VW(21) RT_VW(22) RT_VW(23) RT_VW(24) RT_VW(25) RT_VW(26) RT_VW(27) RT_VW(28) RT_VW(29) RT_VW(30) RT_VW(31)
      RT_VW(32) RT_VW(33) RT_VW(34) RT_VW(35) RT_VW(36) RT_VW(37) RT_VW(38) RT_VW(39) RT_VW(40) RT_VW(41) RT_VW(42) RT_VW(43) RT_VW(44) RT_VW(45) RT_VW(46) RT_VW(47)
    }
#undef RT_VW
    __builtin_amdgcn_sched_barrier(0);
  };
  // the wait that closes a weight group: everything issued after the last piece of the group it needs may stay in flight.  EXPECT =
  // the count of the steady state at that site (one compare instead of the switch's tree)
  auto vm_wait = [&](auto expect_tag) __attribute__((always_inline)) {
    constexpr int EXPECT = decltype(expect_tag)::value;
    const int n = vc_ops - vs_next;
    if (n == EXPECT) { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(EXPECT) : "memory"); __builtin_amdgcn_sched_barrier(0); }
    else vm_wait_n(n);
    vs_next = vs_later;
  };

  // ---- fragment side --------------------------------------------------------------------------------------------------------
  // pixels: lane (r, q) of row tile mt reads row 16 mt + r of the wave's rows, logical chunks q and 4 + q (k = 4 q .. + 3 and
  // 16 + 4 q .. + 3 of the slab): the second read is the first's byte address ^ 64
  const unsigned a_fr = (unsigned)wid * S_AWAVE + (unsigned)(r * 128) + ((((unsigned)q) ^ ((unsigned)(r >> 1) & 7u)) << 4);
  const unsigned w_fr = lds_b + S_OFF_W + (unsigned)lane * 16u;
  f32x4 acc[2][S_NT];
  u32x4 Ah[2], Am[2], Al[2];     // split pixel fragments of the current slab
  u32x4 Bf[2][3];                // weight fragments (planes h, m, l) of two consecutive tiles
  f32x4 raw[2][2];
  // (p + (x ^ 64)) is not (p + x) ^ 64 in general: the chunk field of a_fr is bits 4..6 and mt * 2048 leaves them alone, so
  //  the address of the second read is formed on the byte offset itself
  auto read_raw2 = [&](int slot) __attribute__((always_inline)) {
    const unsigned o0 = (unsigned)slot * S_ASLOT + a_fr, o1 = o0 ^ 64u;
#pragma unroll
    for (int mt = 0; mt < 2; mt++) {
      raw[mt][0] = *reinterpret_cast<const f32x4*>(smem_s + o0 + mt * 2048);
      raw[mt][1] = *reinterpret_cast<const f32x4*>(smem_s + o1 + mt * 2048);
    }
  };
  // (inline asm: as plain loads hipcc re-used the registers of the fragment in use and sank the reads behind the MFMAs that
  //  freed them -- 4 MFMAs before their first use; every wait that follows leaves exactly the newest three reads in flight)
  auto read_b = [&](auto slot_tag, auto off_tag, unsigned wf) __attribute__((always_inline)) {   // wf: w_fr + ring buffer of the half
    constexpr int SL = decltype(slot_tag)::value, OFF = decltype(off_tag)::value;
    Bf[SL][0] = lds_read16u<OFF>(wf);
    Bf[SL][1] = lds_read16u<OFF + 1024>(wf);
    Bf[SL][2] = lds_read16u<OFF + 2048>(wf);
  };

  // ---- epilogue of the previous tile, one 16-channel tile at a time ------------------------------------------------------------
  char* pend_c = nullptr;     // &C[first row of the wave][first channel of the column block] of the tile waiting for its epilogue
  int pend_rows = 0;
  bool pend_full = false;   // every lane of the wave stores (32 valid rows): the chunk's two stores are certainly issued
  const unsigned mt_step = (unsigned)(16 * g.ldc * 4);
  auto epi_chunk = [&](auto nttag) __attribute__((always_inline)) {
    constexpr int nt = decltype(nttag)::value;
    int ln = lane_id();
    asm volatile("" : "+v"(ln));
    const int rr = ln & 15;
    const unsigned lo = (unsigned)((rr * g.ldc + nt * 16 + (ln >> 4) * 4) * 4);
    const f32x4 bias = *reinterpret_cast<const f32x4*>(bias_l + nt * 16 + (ln >> 4) * 4);
#pragma unroll
    for (int mt = 0; mt < 2; mt++) {
      f32x4 o;
      if constexpr (ACT == ACT_HSWISH && (LAB == 0 || LAB == 1)) {
        const f32x4 v = acc[mt][nt] + bias;
        f32x4 t = v + 3.0f;
#pragma unroll
        for (int j = 0; j < 4; j++) { const float tj = t[j]; t[j] = __builtin_amdgcn_fmed3f(tj, 0.0f, 6.0f); }
        o = v * t;
        o = o * 0.16666667f;
        if (LAB == 1) { const f32x4 a4 = {g.epi.lab_a, g.epi.lab_a, g.epi.lab_a, g.epi.lab_a}, c4 = {g.epi.lab_c, g.epi.lab_c, g.epi.lab_c, g.epi.lab_c}; o = __builtin_elementwise_fma(o, a4, c4); }
      } else {
#pragma unroll
        for (int j = 0; j < 4; j++) o[j] = epi_val<ACT, LAB>(acc[mt][nt][j] + bias[j], g.epi.act, g.epi.has_lab, g.epi.lab_a, g.epi.lab_c);
      }
      if (DBG & 64) { *reinterpret_cast<f32x4*>(pend_c + ((nt * 2 + mt) * (g.ldc * 4) + ln * 16)) = o; }   // (timing: 1 KB contiguous per instruction)
      else if (DBG & 128) { constexpr int k_ = nt * 2; *reinterpret_cast<f32x4*>(pend_c + (((ln >> 3) + 8 * ((k_ + mt) & 3)) * (g.ldc * 4) + (ln & 7) * 16 + 128 * ((k_ + mt) >> 2))) = o; }   // (timing: 8 rows x 128 bytes)
      else if ((DBG & 1) ? o[0] == 123.456f : rr + mt * 16 < pend_rows) *reinterpret_cast<f32x4*>(pend_c + (lo + mt * mt_step)) = o;
    }
    if (pend_full) vm_note(2);   // (stores a partial row block may skip are not counted: the waits then wait for more)
  };

  // 12 MFMAs of one weight tile (16 output channels) against the wave's 32 pixels; ZERO: the accumulators start from zero
#define RT_SMF(SL, nt, ZERO) do { \
    _Pragma("unroll") for (int mt_ = 0; mt_ < 2; mt_++) \
      acc[mt_][nt] = mfma_bf16<(DBG & 16) != 0>(Bf[SL][2], Ah[mt_], (ZERO) ? f32x4{0.f, 0.f, 0.f, 0.f} : acc[mt_][nt]); \
    _Pragma("unroll") for (int mt_ = 0; mt_ < 2; mt_++) acc[mt_][nt] = mfma_bf16<(DBG & 16) != 0>(Bf[SL][0], Al[mt_], acc[mt_][nt]); \
    _Pragma("unroll") for (int mt_ = 0; mt_ < 2; mt_++) acc[mt_][nt] = mfma_bf16<(DBG & 16) != 0>(Bf[SL][1], Am[mt_], acc[mt_][nt]); \
    _Pragma("unroll") for (int mt_ = 0; mt_ < 2; mt_++) acc[mt_][nt] = mfma_bf16<(DBG & 16) != 0>(Bf[SL][1], Ah[mt_], acc[mt_][nt]); \
    _Pragma("unroll") for (int mt_ = 0; mt_ < 2; mt_++) acc[mt_][nt] = mfma_bf16<(DBG & 16) != 0>(Bf[SL][0], Am[mt_], acc[mt_][nt]); \
    _Pragma("unroll") for (int mt_ = 0; mt_ < 2; mt_++) acc[mt_][nt] = mfma_bf16<(DBG & 16) != 0>(Bf[SL][0], Ah[mt_], acc[mt_][nt]); \
  } while (0)

  // DBG & 256: in-kernel clock and the cycles one wave spends in the two waits of a slab (diagnostic instantiation only)
  constexpr bool ST = (DBG & 256) != 0;
  const bool st_on = ST && blockIdx.x == 7 && wid == (int)g.epi.am_tiles;
  unsigned long long st_c0 = 0, st_r0 = 0, st_a = 0, st_sum[4] = {0, 0, 0, 0};
  if (ST) { st_c0 = __builtin_amdgcn_s_memtime(); st_r0 = __builtin_amdgcn_s_memrealtime(); }
  // DBG & 512: the same wave's time inside a step: fragment request + wait | request pieces | 12 MFMAs (+ woven VALU)
  constexpr bool ST2 = ST && (DBG & 512) != 0;
  unsigned long long st2_t = 0, st2_sum[4] = {0, 0, 0, 0};
#define RT_ST2(i) do { if (ST2) { if (st_on) { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); if (i) st2_sum[i] += n_ - st2_t; st2_t = n_; } __builtin_amdgcn_sched_barrier(0); } } while (0)
#define RT_STA() do { if (ST) { if (st_on) st_a = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); } } while (0)
#define RT_STB(i) do { if (ST) { if (st_on) { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); st_sum[i] += n_ - st_a; st_a = n_; } __builtin_amdgcn_sched_barrier(0); } } while (0)
  // ---- prologue ---------------------------------------------------------------------------------------------------------------
  w_issue();
  w_issue();
  a_issue();
  a_issue();
  dbg_pro = false;
  vs_next = vs_later = vc_ops;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  const float* sc_lds = reinterpret_cast<const float*>(smem_s + S_OFF_SC);
  auto split_into = [&](int mt, u32x4& h, u32x4& m, u32x4& l, int slot) __attribute__((always_inline)) {
    if (ASC) {   // the slab's scale vector of this lane's row's image: channels 4 q .. + 3 and 16 + 4 q .. + 3
      const int row = 32 * wid + 16 * mt + r;
      const float* t = sc_lds + (wid * 2 + slot) * 64 + (row >= sc_bnd[slot] ? 32 : 0) + 4 * q;
      raw[mt][0] *= *reinterpret_cast<const f32x4*>(t);
      raw[mt][1] *= *reinterpret_cast<const f32x4*>(t + 16);
    }
    if (DBG & 8) { h = __builtin_bit_cast(u32x4, raw[mt][0]); m = __builtin_bit_cast(u32x4, raw[mt][1]); l = h ^ m; }
    else split8(raw[mt][0], raw[mt][1], h, m, l);
  };
  read_raw2(0);
  split_into(0, Ah[0], Am[0], Al[0], 0);
  split_into(1, Ah[1], Am[1], Al[1], 0);
  read_b(IntTag<0>{}, IntTag<0>{}, w_fr);
  __builtin_amdgcn_sched_barrier(0);
  int a_slot = 0;   // ring slot of the current slab's pixels
  int tile_j = 0;   // ordinal of the workgroup's current tile
  int wb = 0;       // ring buffer of the current half's weights
  u32x4 Nh[2], Nm[2], Nl[2];   // the next slab's split pixel fragments (built during the current slab's second half)

  // One 32-deep slab: 15 steps of 12 MFMAs in three groups of five.  EPI: first slab of a tile -- the accumulators start from zero
  // and, if a tile is pending, its epilogue chunks ride along (chunk nt + 1 woven into the MFMAs of step nt).
  // Request schedule of a slab (at most one weight and one pixel piece per step): steps 0-3, 5-8, 10-13 the weights two groups
  // ahead (the ring buffer of the group the last barrier retired), steps 0-3 also the pixels two slabs ahead (the slot whose
  // fragments were read a slab ago).  The barrier that closes a group needs the weights of the next one, requested two groups
  // earlier: everything issued since may stay in flight.
  auto slab = [&](auto epi_tag, bool pend) __attribute__((always_inline)) {
    constexpr bool EPI = decltype(epi_tag)::value;
    unsigned wf = w_fr + (unsigned)wb * S_WGRP;
#define RT_WEAVE() do { \
      _Pragma("unroll") for (int i_ = 0; i_ < 12; i_++) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, 4, 0); } } while (0)
    // one step: request the next tile's weight fragments, wait for this tile's, this step's request pieces, 12 MFMAs -- with the
    // VALU work that rides along woven in, four instructions behind each MFMA: the epilogue chunk of the NEXT step's tile (whose
    // accumulators this step does not touch) and WORK (the split)
#define RT_STEP_V(SL, nt, NOFF, REQ, WORK) do { \
      RT_ST2(0); \
      read_b(IntTag<1 - SL>{}, IntTag<NOFF>{}, wf); \
      lgkm_wait<3>(); \
      RT_ST2(1); \
      REQ; \
      RT_ST2(2); \
      RT_SMF(SL, nt, EPI); \
      if (EPI && pend) epi_chunk(IntTag<(nt) + 1>{}); \
      WORK; \
      RT_WEAVE(); \
      __builtin_amdgcn_sched_barrier(0); \
      RT_ST2(3); } while (0)
#define RT_STEP(SL, nt, NOFF, REQ) RT_STEP_V(SL, nt, NOFF, REQ, (void)0)
    // last step of a group: this tile's fragments are in registers; the next group's weights have landed (this wave's pieces:
    // vm_wait; everybody's: the barrier), its first fragments are requested, then the 12 MFMAs
#define RT_LAST(SL, nt, REQ, WORK, STI, VME) do { \
      RT_STA(); \
      lgkm_wait<0>(); \
      vm_wait(IntTag<VME>{}); \
      RT_STB(STI); \
      __builtin_amdgcn_s_barrier(); \
      __builtin_amdgcn_sched_barrier(0); \
      RT_STB(STI + 1); \
      wb = wb == 2 ? 0 : wb + 1; \
      wf = w_fr + (unsigned)wb * S_WGRP; \
      read_b(IntTag<1 - SL>{}, IntTag<0>{}, wf); \
      __builtin_amdgcn_sched_barrier(0); \
      REQ; \
      RT_SMF(SL, nt, EPI); \
      if (EPI && pend && (nt) + 1 < S_NT) { epi_chunk(IntTag<((nt) + 1 < S_NT ? (nt) + 1 : 0)>{}); } \
      WORK; \
      RT_WEAVE(); \
      __builtin_amdgcn_sched_barrier(0); } while (0)
#define RT_WA(k) do { w_piece(IntTag<k>{}); a_piece(IntTag<k>{}); } while (0)
    // The id of the workgroup's next tile: one returning atomic of wave 0 at the start of a tile's first slab, published through
    // tileq[] before the barrier that closes the slab's second group.  The request streams read it when they leave the tile --
    // for K = 128 (four slabs) that is as early as the second slab's fourth step, so TWO barriers of the first slab must lie
    // behind the write (published at the slab's end, with no barrier in between, the other waves raced it: memory faults that
    // depended on how the session's lanes happened to line up).
    unsigned fetched = 0;
    if (EPI && (g.dyn & 2)) {
      if (wid == 0) {
        if (lane_id() == 0) asm volatile("s_nop 4\n\tglobal_atomic_add %0, %1, %2, %3 sc0" : "=v"(fetched) : "v"(0u), "v"(1u), "s"(g.sched) : "memory");
        vm_note(1); vs_at = vc_ops;
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (EPI && pend) epi_chunk(IntTag<0>{});
    // ---- group 0: tiles 0..4
    RT_STEP(0, 0, 1 * 3072, RT_WA(0)); RT_STEP(1, 1, 2 * 3072, RT_WA(1)); RT_STEP(0, 2, 3 * 3072, RT_WA(2)); RT_STEP(1, 3, 4 * 3072, RT_WA(3));
    RT_LAST(0, 4, (void)0, (void)0, 0, 8);
    // ---- group 1: tiles 5..9
    RT_STEP(1, 5, 1 * 3072, w_piece(IntTag<0>{})); RT_STEP(0, 6, 2 * 3072, w_piece(IntTag<1>{})); RT_STEP(1, 7, 3 * 3072, w_piece(IntTag<2>{}));
    RT_STEP(0, 8, 4 * 3072, w_piece(IntTag<3>{}));
    if (EPI && (g.dyn & 4)) {   // the next tile's id has returned (everything issued since may stay in flight)
      if (wid == 0) {
        vm_wait_n(vc_ops - vs_at);
        if (lane_id() == 0) tq_write(tile_j + 1, G + (int)fetched);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    RT_LAST(1, 9, (void)0, (void)0, 2, 5);
    // ---- group 2: tiles 10..14.  The next slab's pixels have landed: their request is older than the weights the barrier
    // above waited for (requested in this slab's first steps, behind them).
    read_raw2(a_slot ^ 1);
    __builtin_amdgcn_sched_barrier(0);
    RT_STEP_V(0, 10, 1 * 3072, w_piece(IntTag<0>{}), split_into(0, Nh[0], Nm[0], Nl[0], a_slot ^ 1));
    RT_STEP_V(1, 11, 2 * 3072, w_piece(IntTag<1>{}), split_into(1, Nh[1], Nm[1], Nl[1], a_slot ^ 1));
    RT_STEP(0, 12, 3 * 3072, w_piece(IntTag<2>{})); RT_STEP(1, 13, 4 * 3072, w_piece(IntTag<3>{}));
    RT_LAST(0, 14, (void)0, (void)0, 2, 4);
#undef RT_STEP
#undef RT_STEP_V
#undef RT_LAST
#undef RT_WA
#undef RT_WEAVE
    // hand the fragments over (register moves; the second set keeps the 12 MFMAs above independent of the split).  The weight
    // fragment just requested must have landed before it is moved: the 12 MFMAs above are in the pipe meanwhile.
    lgkm_wait<0>();
#pragma unroll
    for (int mt = 0; mt < 2; mt++) { Ah[mt] = Nh[mt]; Am[mt] = Nm[mt]; Al[mt] = Nl[mt]; }
    Bf[0][0] = Bf[1][0]; Bf[0][1] = Bf[1][1]; Bf[0][2] = Bf[1][2];
    a_slot ^= 1;
  };

  for (int t = (int)blockIdx.x; t < n_tiles; ) {
    const int rb = t / g.n_cb, cb = t - rb * g.n_cb;
    const bool pend = pend_c != nullptr;
    pend_full = pend && pend_rows >= 32 && !(DBG & 1);
    slab(std::true_type{}, pend);
    // (the pending tile's last bias read lies before the last barrier of the slab above; this tile's epilogue -- the next reader --
    //  begins with the next tile's first slab or the tail below, barriers away)
    stage_bias(cb);
    for (int s = 1; s < nslab; s++) slab(std::false_type{}, false);
    const long long m0 = (long long)rb * S_BM + 32 * wid;
    pend_c = reinterpret_cast<char*>(g.C + m0 * g.ldc + g.coff + cb * S_BN);
    pend_rows = (int)max(0ll, min(32ll, g.M - m0));
    tile_j++;
    { const int q_ = (g.dyn & 8) ? tq_read(tile_j) : 0;
      t = (g.dyn & 1) ? q_ : t + G; }   // (published in this tile's first slab, barriers ago)
  }
  __syncthreads();
  if (pend_c) {
    epi_chunk(IntTag<0>{}); epi_chunk(IntTag<1>{}); epi_chunk(IntTag<2>{}); epi_chunk(IntTag<3>{}); epi_chunk(IntTag<4>{});
    epi_chunk(IntTag<5>{}); epi_chunk(IntTag<6>{}); epi_chunk(IntTag<7>{}); epi_chunk(IntTag<8>{}); epi_chunk(IntTag<9>{});
    epi_chunk(IntTag<10>{}); epi_chunk(IntTag<11>{}); epi_chunk(IntTag<12>{}); epi_chunk(IntTag<13>{}); epi_chunk(IntTag<14>{});
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // (requests issued for tiles that do not exist write zeros into LDS)
  // the last workgroup to finish leaves the two counters at zero for the next launch on this stream
  if (tid == 0 && (g.dyn & 16)) {
    const unsigned done = __hip_atomic_fetch_add(g.sched + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (done == (unsigned)G - 1) {
      __hip_atomic_store(g.sched, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(g.sched + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  if (ST && wid == 0 && lane == 0) {
    unsigned long long* o = reinterpret_cast<unsigned long long*>(g.epi.am_max);
    o[8 + 2 * blockIdx.x] = st_r0; o[9 + 2 * blockIdx.x] = __builtin_amdgcn_s_memrealtime();
  }
  if (ST && st_on && lane == 0) {
    unsigned long long* o = reinterpret_cast<unsigned long long*>(g.epi.am_max);
    o[0] = __builtin_amdgcn_s_memtime() - st_c0; o[1] = __builtin_amdgcn_s_memrealtime() - st_r0;
    for (int i = 0; i < 4; i++) o[2 + i] = st_sum[i];
    for (int i = 0; i < 4; i++) o[1040 + i] = st2_sum[i];
  }
#undef RT_STA
#undef RT_ST2
#undef RT_STB
#undef RT_SMF
}

// ---- host side ----------------------------------------------------------------------------------------------------------------
int g_gemm_split = getenv("RT_GEMM_SPLIT") ? atoi(getenv("RT_GEMM_SPLIT")) : 0;   // opt-in (RT_GEMM_SPLIT=1 / rt_debug_set_variants): gemm() takes the split-bf16 kernel where it applies

bool gemm_split_supported(int lda, long long M, int K, int N, int Npad16, const Epilogue& epi) {
  if (epi.am_max || epi.residual) return false;
  static const int only = getenv("RT_GS_ONLY") ? atoi(getenv("RT_GS_ONLY")) : 3;   // (triage: 1 = plain launches only, 2 = +se only)
  if (!(only & (epi.a_scale ? 2 : 1))) return false;
  // squeeze-excite scale: a row-block table of either form, hardswish epilogue, images of >= 128 rows (what gemm_se_tile_rows() > 0 says)
  if (epi.a_scale && (!epi.a_tab || (epi.a_tab_stride != 2 && epi.a_tab_stride != 3) || epi.act != ACT_HSWISH || epi.n_img <= 0 || epi.ld_scale < K)) return false;
  if (Npad16 != N || N % S_BN != 0 || N > 960) return false;
  if (lda < round_up(K, KC) || (lda & 3)) return false;          // whole 32-deep slabs readable (padding channels hold zeros)
  if ((long long)lda * 4 * 32 >= (1ll << 31)) return false;
  // (large launches only: the small-batch dispatch of a one-page call stays on the narrow fp32 kernel)
  return M >= 32768 && K > 3 * KC;   // (>= 4 slabs: the request streams read the next tile's id two slabs before a tile ends; it is published in the tile's first slab)
}

// the split planes of a packed fp32 weight matrix, built on first use and kept for the life of the process (keyed by the
// device pointer of the fp32 pack: the networks' weights live as long as their session)
static std::mutex g_split_mu;
static std::map<std::pair<int, const float*>, unsigned short*> g_split_cache;
static const unsigned short* split_pack_of(hipStream_t st, const float* Wp, int nslab, int Npad, int n_cb) {
  int dev = 0;
  RT_HIP_CHECK(hipGetDevice(&dev));
  std::lock_guard<std::mutex> lk(g_split_mu);
  auto it = g_split_cache.find({dev, Wp});
  if (it != g_split_cache.end()) return it->second;
  // First use of this pack on this device: build the planes and WAIT for them before the entry becomes visible -- the session's
  // lanes launch the same layers from their own threads on their own streams, and a lane that found the entry while the pack
  // kernel was still queued on another lane's stream multiplied by whatever the allocation held.
  unsigned short* p = nullptr;
  const size_t bytes = (size_t)n_cb * nslab * S_WSLAB;
  RT_HIP_CHECK(hipMalloc((void**)&p, bytes + 65536));   // (+ slack behind the last group)
  hipError_t e = hipMemsetAsync(p, 0, bytes + 65536, st);
  if (e == hipSuccess) {
    const long long total = (long long)n_cb * nslab * S_NT * 64;
    hipLaunchKernelGGL(k_split_pack, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, Wp, nslab, Npad, n_cb, p);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  if (e != hipSuccess) { (void)hipFree(p); throw RtError(4, std::string("gemm_split: building the split planes failed: ") + hipGetErrorString(e)); }
  g_split_cache[{dev, Wp}] = p;
  return p;
}

void gemm_split_forget(const float* Wp) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return;
  std::lock_guard<std::mutex> lk(g_split_mu);
  auto it = g_split_cache.find({dev, Wp});
  if (it != g_split_cache.end()) { (void)hipFree(it->second); g_split_cache.erase(it); }
}

void gemm_split(hipStream_t st, const float* A, int lda, long long M, int K, const float* Wp, int N, int Npad16, float* C,
                int ldc, int coff, const Epilogue& epi) {
  GemmSArgs g;
  g.A = A; g.C = C; g.M = M; g.lda = lda; g.nslab = (K + KC - 1) / KC; g.N = N; g.ldc = ldc; g.coff = coff;
  g.n_rb = (int)((M + S_BM - 1) / S_BM); g.n_cb = N / S_BN; g.epi = epi;
  g.a_scale = epi.a_scale; g.a_tab = epi.a_tab; g.ld_scale = epi.ld_scale; g.n_img = epi.n_img; g.tab_stride = epi.a_tab_stride; g.K = K;
  g.Ws = split_pack_of(st, Wp, g.nslab, Npad16, g.n_cb);
  {
    int dev = 0;
    RT_HIP_CHECK(hipGetDevice(&dev));
    static std::mutex mu;
    static std::map<std::pair<int, hipStream_t>, unsigned*> counters;   // zero between launches (the kernel resets them)
    std::lock_guard<std::mutex> lk(mu);
    unsigned*& c = counters[{dev, st}];
    if (!c) { RT_HIP_CHECK(hipMalloc((void**)&c, 256)); RT_HIP_CHECK(hipMemset(c, 0, 256)); }
    g.sched = c;
    static const int stat = getenv("RT_GS_STATIC") ? atoi(getenv("RT_GS_STATIC")) : 0;   // A/B: static tile lists
    g.dyn = stat ? 0 : 31;   // (bits: 1 ids from the queue, 2 atomic, 4 publish, 8 queue reads, 16 end-of-kernel counters)
  }
  const int grid0 = std::min(g.n_rb * g.n_cb, 2 * stream_cus(st));   // two workgroups per CU
#define RT_GS(ACTV, LABV) do { allow_big_lds((const void*)k_gemm_split<ACTV, LABV>, 160 * 1024); \
    RT_LAUNCH((k_gemm_split<ACTV, LABV>), dim3((unsigned)grid), dim3(S_NTHR), S_LDS, st, g); } while (0)
  static const int dbg = getenv("RT_GS_DBG") ? atoi(getenv("RT_GS_DBG")) : 0;   // timing experiments only (wrong results)
  const int grid_env = getenv("RT_GS_GRID") ? atoi(getenv("RT_GS_GRID")) : 0;   // (experiments: e.g. 256 = one workgroup per CU)
  const int grid = grid_env > 0 ? std::min(grid0, grid_env) : grid0;
  if (dbg & 256) {
    static unsigned long long* dst = nullptr;
    if (!dst) RT_HIP_CHECK(hipMalloc((void**)&dst, 64 + 16 * 1024));
    RT_HIP_CHECK(hipMemsetAsync(dst, 0, 64 + 16 * 1024, st));
    g.epi.am_max = reinterpret_cast<float*>(dst);
    g.epi.am_tiles = getenv("RT_GS_WAVE") ? atoi(getenv("RT_GS_WAVE")) : 0;
    if (dbg & 512) { allow_big_lds((const void*)k_gemm_split<ACT_HSWISH, 1, 768>, 160 * 1024);
      RT_LAUNCH((k_gemm_split<ACT_HSWISH, 1, 768>), dim3((unsigned)grid), dim3(S_NTHR), S_LDS, st, g); }
    else { allow_big_lds((const void*)k_gemm_split<ACT_HSWISH, 1, 256>, 160 * 1024);
      RT_LAUNCH((k_gemm_split<ACT_HSWISH, 1, 256>), dim3((unsigned)grid), dim3(S_NTHR), S_LDS, st, g); }
    unsigned long long h[8 + 2048];
    RT_HIP_CHECK(hipMemcpyAsync(h, dst, sizeof(h), hipMemcpyDeviceToHost, st));
    RT_HIP_CHECK(hipStreamSynchronize(st));
    {
      int occ = -1;
      (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (const void*)k_gemm_split<ACT_HSWISH, 1, 256>, S_NTHR, S_LDS);
      unsigned long long t0 = ~0ull, t1 = 0;
      for (int b = 0; b < grid && b < 1024; b++) { t0 = std::min(t0, h[8 + 2 * b]); t1 = std::max(t1, h[9 + 2 * b]); }
      int late_start = 0; double s_end = 0, mn = 1e30, mx = 0;
      for (int b = 0; b < grid && b < 1024; b++) {
        if ((h[8 + 2 * b] - t0) > 5000) late_start++;
        const double e = (h[9 + 2 * b] - t0) / 100.0; s_end += e; mn = std::min(mn, e); mx = std::max(mx, e);
      }
      fprintf(stderr, "gsplit blocks: occupancy API %d per CU; grid %d; span %.1f us; %d blocks started > 50 us after the first; block end min %.1f mean %.1f max %.1f us\n",
              occ, grid, (t1 - t0) / 100.0, late_start, mn, s_end / std::min(grid, 1024), mx);
    }
    if (dbg & 512) fprintf(stderr, "gsplit step stamps: fragment request + wait %llu | request pieces %llu | MFMAs + woven work %llu cycles per launch\n", h[1041], h[1042], h[1043]);
    if (h[1]) fprintf(stderr, "gsplit wave %d of block 7: %llu cycles in %.1f us = %.3f GHz; per launch: mid wait %llu + barrier %llu, end wait %llu + barrier %llu cycles (%d slabs per tile)\n",
                      g.epi.am_tiles, h[0], h[1] / 100.0, h[0] / (h[1] * 10.0), h[2], h[3], h[4], h[5], g.nslab);
    return;
  }
  if (dbg) {
#define RT_GSD(D) case D: allow_big_lds((const void*)k_gemm_split<ACT_HSWISH, 1, D>, 160 * 1024); \
    RT_LAUNCH((k_gemm_split<ACT_HSWISH, 1, D>), dim3((unsigned)grid), dim3(S_NTHR), S_LDS, st, g); break;
    switch (dbg) { RT_GSD(1) RT_GSD(2) RT_GSD(4) RT_GSD(6) RT_GSD(8) RT_GSD(7) RT_GSD(24) RT_GSD(25) RT_GSD(26) RT_GSD(28) RT_GSD(30) RT_GSD(94) RT_GSD(158) RT_GSD(64) RT_GSD(128)
      default: throw RtError(8, "gemm_split: unknown RT_GS_DBG"); }
#undef RT_GSD
    return;
  }
  if (epi.a_scale) {
#define RT_GSE(LABV) do { allow_big_lds((const void*)k_gemm_split<ACT_HSWISH, LABV, 0, true>, 160 * 1024); \
    RT_LAUNCH((k_gemm_split<ACT_HSWISH, LABV, 0, true>), dim3((unsigned)grid), dim3(S_NTHR), S_LDS_ASC, st, g); } while (0)
    if (epi.has_lab) RT_GSE(1); else RT_GSE(0);
#undef RT_GSE
    return;
  }
  if (epi.act == ACT_HSWISH && epi.has_lab) RT_GS(ACT_HSWISH, 1);
  else if (epi.act == ACT_HSWISH) RT_GS(ACT_HSWISH, 0);
  else RT_GS(-1, -1);
#undef RT_GS
}

}  // namespace nn
}  // namespace rt
