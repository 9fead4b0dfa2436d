// fp32 wide GEMM staged by LDS-DMA (global_load_lds) with persistent tiles, for gfx950.
//
// k_gemm32p serves the 240- / 480-channel 1x1 convolutions of the recognition network (a quarter of the C3 step's
// kernel time; they replace ONNX Runtime's Conv at /root/reference/retto-core/src/worker/ort_worker.rs:211-220).
// Same tile, fragment maps and per-accumulator K order as k_gemm_wide<4,5,4,3> (nn_kernels.hip): results are
// bit-identical.  What changes is how the operands reach LDS and what overlaps:
//   * both operands go global -> LDS by global_load_lds_dwordx4 (no VGPR staging: the 32 prefetch registers and the
//     stash phase between two barriers are gone), 32-deep K slabs in a ring of two stage buffers, un-padded 128-byte
//     rows with an XOR swizzle applied to the per-lane DMA source and again on the fragment reads (conflict-free
//     ds_read_b128);
//   * ONE barrier per slab, in its middle (the k_gemm16p scheme): reads(s, g1) | MFMAs(s, g0) | slab s + 1 landed |
//     BARRIER | request slab s + 2 into the buffer of slab s | reads(s + 1, g0) | MFMAs(s, g1) -- after the barrier every
//     wave still has 80 MFMAs queued, so the matrix pipe never drains at a stage boundary;
//   * persistent workgroups (one per CU) walk their tiles without leaving the pipeline: the first slabs of tile t + 1
//     are requested while tile t is multiplied, and the 20 stores a wave issues for tile t drain under the MFMAs of
//     tile t + 1 (counted vmcnt: the first wait after an epilogue leaves exactly those stores in flight).
#include "nn.h"
#include "nn_dev.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <mutex>
#include <string>
#include <type_traits>

namespace rt {
namespace nn {

namespace {

#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"   // M0 is named as clobbered on purpose: nothing else in this kernel uses it
// LDS-DMA request: lane i writes 16 bytes at M0 + 16 i; source = 64-bit scalar base + 32-bit per-lane byte offset.
// Inline asm, not the builtin: with the builtin in a loop hipcc waits lgkmcnt(0) before every MFMA group (nn_f16_dma.hip).
__device__ __forceinline__ void glds16_so(unsigned voff, const void* sbase, unsigned lds_sgpr) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(lds_sgpr) : "memory", "m0");
}
// The same through a buffer resource (round 5): source = resource base + per-lane byte offset + scalar offset; a lane whose offset
// is beyond the resource's range writes zeros.
__device__ __forceinline__ void blds16(unsigned voff, __amdgpu_buffer_rsrc_t rs, unsigned lds_sgpr, unsigned soff) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %3 offen lds" ::"v"(voff), "s"(rs), "s"(lds_sgpr), "s"(soff) : "memory", "m0");
}
#pragma clang diagnostic pop
#ifndef RT_G32P_BUF
#define RT_G32P_BUF 1   // 0 (make EXTRA=-DRT_G32P_BUF=0): the round-4 request form, per-lane offsets re-derived at every request
#endif
__device__ __forceinline__ unsigned lds_addr32(const void* p) { return (unsigned)(size_t)(__attribute__((address_space(3))) const char*)p; }
template <int OFF>
__device__ __forceinline__ f32x4 lds_read16f(unsigned byte_addr) {   // address + compile-time offset in the instruction
  static_assert(OFF >= 0 && OFF < 65536, "16-bit offset field");
  f32x4 v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(byte_addr), "n"(OFF));
  return v;
}
__device__ __forceinline__ int lane_id() { return (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }
template <int N>
__device__ __forceinline__ void lgkm_wait() {   // leaves the newest N LDS reads in flight and pins the order around it
  static_assert(N >= 0 && N <= 15, "lgkmcnt is a 4-bit counter");
  asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory");
  __builtin_amdgcn_sched_barrier(0);
}

}  // namespace

struct GemmPArgs {
  const float* A; const float* Wp; float* C;
  long long M;
  int lda, K, N, Npad, ldc, coff;
  int n_rb, n_cb;      // row blocks of 256, column blocks of 240
  // ASC (squeeze-excite scale folded into the pixel operand): row m of image i is multiplied by a_scale[i * ld_scale + k].
  // a_tab holds 3 ints per row block: {image of the block's first row, first row of the next image, of the one after}
  // (INT_MAX when there is none); every image has >= 128 rows, so a 256-row block touches at most 3.
  const float* a_scale; const int* a_tab; int ld_scale, n_img;
  unsigned* sched;     // tile counters ([16 q]: queue q, tiles handed out beyond the workgroups' first ones) and [128] workgroups done; zero between launches
  int xcdq;            // 1: one tile queue per XCD (workgroup b is on XCD b % 8 and takes the row blocks rb % 8 == b % 8, both column
                       // tiles of a row block consecutively: they meet in that XCD's L2); 0: one queue, tile ids in launch order
  Epilogue epi;
};

constexpr int P_BM = 256, P_BN = 240, P_MT = 4, P_NT = 5, P_WN = 3, P_NW = 12, P_NTHR = 768;
constexpr unsigned P_ABYTES = P_BM * 128, P_WBYTES = P_BN * 128, P_STAGE = P_ABYTES + P_WBYTES;   // 63488 bytes per slab
constexpr int P_AJ = P_BM / 8, P_WJ = P_BN / 8;              // 1-KB DMA pieces (8 rows of 128 bytes) per operand and slab
constexpr int P_NSTORE = P_MT * P_NT;                        // store instructions of one wave's epilogue
constexpr int P_BIAS_MAX = 960;                              // bias vector kept in LDS (N <= 960)
constexpr int P_SCK = 512;                                   // ASC: K <= 512; scale table = 2 tile parities x 3 images x P_SCK floats
constexpr size_t P_LDS = 2 * (size_t)P_STAGE + P_BIAS_MAX * 4 + 16;   // slabs | bias | tile queue
constexpr size_t P_LDS_ASC = P_LDS + 2 * 3 * P_SCK * 4;               // ... | scale tables

// ACT / LAB: compile-time epilogue (-1 = decided per element); HALF: K = 32 j + 16, the last slab holds one group
// DBG (timing experiments, wrong results): 1 no stores, 2 no requests after the first two, 4 no epilogue math, 8 stamps
template <int ACT, int LAB, bool HALF, int DBG = 0, bool ASC = false>
__global__ __launch_bounds__(P_NTHR, 1) void k_gemm32p(const GemmPArgs g) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem32p[];
  constexpr int MT = P_MT, NT = P_NT, WN = P_WN;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);   // (SGPR: everything derived from it stays scalar)
  const int r = lane & 15, q = lane >> 4;
  const int wm = wid / WN, wn = wid - wm * WN;
  const unsigned lds_b = __builtin_amdgcn_readfirstlane(lds_addr32(smem32p));
  float* bias_l = reinterpret_cast<float*>(smem32p + 2 * P_STAGE);
  const int nkc = (g.K + KC - 1) / KC;
  const int G = gridDim.x;
  unsigned long long ck_t0 = 0, ck_r0 = 0;
  if (DBG & 16) { ck_t0 = __builtin_amdgcn_s_memtime(); ck_r0 = __builtin_amdgcn_s_memrealtime(); }

  // bias -> LDS once (the epilogue must not issue loads of its own: the vmcnt bookkeeping below counts its stores)
  for (int i = tid; i < P_BIAS_MAX; i += P_NTHR) bias_l[i] = (g.epi.bias && i < g.Npad) ? g.epi.bias[i] : 0.f;

  // ---- request side: slabs are requested in the order they are multiplied, two slabs ahead of the MFMAs -------------
  // A slab image is 62 pieces of 1 KB (8 rows of 128 bytes): pieces 0..31 the 256 pixel rows, 32..61 the 240 weight rows.
  // Lane i of a request writes (row 8 p + i / 8, physical chunk i % 8) = logical chunk (i % 8) ^ ((row >> 1) & 7) of that row.
  // Wave w requests pieces w + 12 i, i = 0..4 (i < 2: pixels, i > 2: weights, i = 2: pixels for w < 8), waves 0 / 1 also 60 / 61.
  // The per-lane source offsets are re-derived from the lane id at every request (~30 VALU instructions per slab and wave):
  // kept in registers they cost six VGPRs the first slab of a tile does not have (it spills accumulators).
  // Tiles are handed out dynamically: workgroup b starts with tile b, every further tile comes from an atomic counter.  (With
  // a static tile list a workgroup that starts late -- the session's lanes run other kernels on the same GPU, and this one
  // needs most of a CU's LDS -- still carries its full share, and the launch ends that much later: in the three-lane
  // production setting the static form gave back everything the kernel had gained.)  Wave 0 fetches the id of the
  // workgroup's (j + 2)-th tile while tile j's first slab is multiplied and publishes it through tileq[] at the next
  // hand-over barrier; the request side needs it 2 slabs before tile j + 1 ends, the MFMA side when tile j + 1 ends.
  const int n_tiles = g.n_rb * g.n_cb;
  volatile int* tileq = reinterpret_cast<volatile int*>(smem32p + 2 * P_STAGE + P_BIAS_MAX * 4);   // ids of the tiles j, j + 1, .. (slot j & 3)
  // (round 4) per-XCD queues: PMC showed the N = 480 layers fetching 1.93x their input -- the two 240-column tiles of a row
  // block went to workgroups on different XCDs, each L2 read the 256 x K pixel block from the fabric.  Queue q holds the row
  // blocks rb = 8 i + q with their column tiles in order; local index t -> tile (8 (t / n_cb) + q) * n_cb + t % n_cb.
  const int xq = g.xcdq ? (int)(blockIdx.x & 7) : 0;
  const int q_wgs = g.xcdq ? (G - xq + 7) >> 3 : G;                                   // workgroups of this queue (their first tiles: local 0 .. q_wgs - 1)
  const int q_tiles = g.xcdq ? ((g.n_rb - xq + 7) >> 3) * g.n_cb : n_tiles;           // tiles of this queue
  auto tile_of = [&](int t) __attribute__((always_inline)) {
    if (!g.xcdq) return t;
    if (t >= q_tiles) return n_tiles;   // (dead: past the queue's end)
    const int i = t / g.n_cb;
    return (8 * i + xq) * g.n_cb + (t - i * g.n_cb);
  };
  const int first_tile = tile_of(g.xcdq ? (int)(blockIdx.x >> 3) : (int)blockIdx.x);
  if (tid == 0) tileq[0] = first_tile;
  int it_tile = first_tile, it_j = 0, it_kc = 0;
  int it_rb = it_tile / g.n_cb, it_cb = it_tile - it_rb * g.n_cb;
  unsigned it_buf = 0;
  bool it_live = it_tile < n_tiles;
  bool fetch_now = false;     // the next slab (the second of its tile) fetches the id of the workgroup's next tile
  int pub_slot = 0;
  const bool p2_is_a = wid + 2 * P_NW < P_AJ;
  const unsigned pitch = (unsigned)(g.lda * 4);
  int dbg_issued = 0;
#if RT_G32P_BUF
  // per-lane request offsets: lane i of a wave's piece addresses row 8 wid + i / 8 of the piece group, physical chunk i % 8
  constexpr bool REQ_HELD = ASC;
  unsigned held_a = 0, held_w = 0;
  if (REQ_HELD) {
    const int rsub = lane >> 3, c0 = lane & 7, row = 8 * wid + rsub;
    const unsigned ch = (unsigned)((c0 ^ ((row >> 1) & 7)) * 16);
    held_a = (unsigned)row * pitch + ch;
    held_w = (unsigned)(row * 128) + ch;
  }
#endif
  auto dma_issue = [&]() __attribute__((always_inline)) {
    if (it_live && !((DBG & 2) && dbg_issued >= 2)) {
      dbg_issued++;
      const long long m0 = (long long)it_rb * P_BM;
      const unsigned dst = lds_b + it_buf * P_STAGE + (unsigned)wid * 1024;
#if RT_G32P_BUF
      // Buffer form: every piece a wave requests lies 96 rows (12 KB of slab image) behind its previous one, and 96 rows do not
      // change the swizzle term -- so ONE per-lane offset per operand (req_a / req_w, fixed for the kernel) serves all six
      // requests; the piece, slab and column-block advances are scalar.  The pixel descriptor ends with the row block's last
      // valid row: rows beyond M arrive as zeros (never stored) without the per-lane clamp.  ~8 VALU instructions per slab and
      // wave instead of ~30 -- an fp32 MFMA loop pays each of them in MFMA time.
      {
        // (the plain instantiations re-derive the per-lane offset WITHOUT the wave's row base 8 wid: that base goes into the
        //  descriptor -- start and length -- not into the scalar offset, which the range check of a raw buffer does not see:
        //  rows beyond M must come back as zeros for every wave)
        const int wrow = REQ_HELD ? 0 : 8 * wid;
        const unsigned rows_here = (unsigned)max(0ll, min((long long)P_BM, g.M - m0) - wrow);
        const __amdgpu_buffer_rsrc_t ars = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.A + (m0 + (rows_here ? wrow : 0)) * g.lda), 0, rows_here * pitch, 0x00020000);
        // (weights: the descriptor starts 8 pieces BEFORE the column block's rows -- piece 2 of the waves 8..11 is weight piece
        //  wid - 8 -- so that every scalar offset is >= 0; nothing below the first weight row is ever addressed)
        const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.Wp + (long long)it_cb * P_BN * KC - 8 * 256), 0, 0x7fffffffu, 0x00020000);
        // (the two per-lane offsets: held in registers by the squeeze-excite instantiations, re-derived from the lane id -- ~10
        //  VALU instructions -- by the plain ones: each form spills accumulators in the other family, allocation is at 168 of 168)
        unsigned req_a = held_a, req_w = held_w;
        unsigned sa = (unsigned)it_kc * (KC * 4), swk = (unsigned)it_kc * (unsigned)g.Npad * (KC * 4);
        if (!REQ_HELD) {
          int ln_ = lane_id();
          asm volatile("" : "+v"(ln_));
          const unsigned ch_ = (unsigned)(((ln_ & 7) ^ (((ln_ >> 4) & 3) | ((wid & 1) << 2))) << 4);
          req_a = (unsigned)(ln_ >> 3) * pitch + ch_; req_w = (unsigned)((ln_ >> 3) * 128) + ch_;
          swk += (unsigned)wid * 1024u;
        }
        blds16(req_a, ars, dst, sa);
        blds16(req_a + 96u * pitch, ars, dst + 1 * (P_NW * 1024), sa);
        if (p2_is_a) blds16(req_a + 192u * pitch, ars, dst + 2 * (P_NW * 1024), sa);
        else blds16(req_w, wrs, dst + 2 * (P_NW * 1024), swk);
        blds16(req_w, wrs, dst + 3 * (P_NW * 1024), swk + 12 * 1024);
        blds16(req_w, wrs, dst + 4 * (P_NW * 1024), swk + 24 * 1024);
        if (wid + 5 * P_NW < P_AJ + P_WJ) blds16(req_w, wrs, dst + 5 * (P_NW * 1024), swk + 36 * 1024);
      }
      int ln = 0;
      if (ASC) { ln = lane_id(); asm volatile("" : "+v"(ln)); }
#else
      const int last = (int)min((long long)P_BM - 1, g.M - 1 - m0);   // rows beyond M re-read the last valid row (never stored)
      const float* abase = g.A + m0 * g.lda + it_kc * KC;
      const float* wbase = g.Wp + ((long long)it_kc * g.Npad + it_cb * P_BN) * KC;
      int ln = lane_id();
      asm volatile("" : "+v"(ln));
      const int rsub = ln >> 3, c0 = ln & 7;
      auto off_a = [&](int i) __attribute__((always_inline)) {
        const int row = 8 * (wid + P_NW * i) + rsub;
        return (unsigned)min(row, last) * pitch + (unsigned)((c0 ^ ((row >> 1) & 7)) * 16);
      };
      auto off_w = [&](int i) __attribute__((always_inline)) {
        const int row = 8 * (wid + P_NW * i - P_AJ) + rsub;
        return (unsigned)(row * 128 + (c0 ^ ((row >> 1) & 7)) * 16);
      };
      glds16_so(off_a(0), abase, dst);
      glds16_so(off_a(1), abase, dst + 1 * (P_NW * 1024));
      if (p2_is_a) glds16_so(off_a(2), abase, dst + 2 * (P_NW * 1024));
      else glds16_so(off_w(2), wbase, dst + 2 * (P_NW * 1024));
      glds16_so(off_w(3), wbase, dst + 3 * (P_NW * 1024));
      glds16_so(off_w(4), wbase, dst + 4 * (P_NW * 1024));
      if (wid + 5 * P_NW < P_AJ + P_WJ) glds16_so(off_w(5), wbase, dst + 5 * (P_NW * 1024));
#endif
      if (ASC && it_kc == 0 && wid >= 2 && wid < 8) {
        // the tile's scale vectors (3 images x K floats) ride with its first slab: waves 2..7 request one 1-KB piece each
        // (image slot (wid - 2) / 2, half (wid - 2) % 2) into the table of this tile's parity
        const int slot = (wid - 2) >> 1, half = (wid - 2) & 1;
        const int img = min(g.a_tab[3 * it_rb] + slot, g.n_img - 1);
        const float* sbase = g.a_scale + (long long)img * g.ld_scale;
        const unsigned soff = (unsigned)min(half * 1024 + ln * 16, g.K * 4 - 16);   // (K = 240: the second half re-reads the tail)
        glds16_so(soff, sbase, lds_b + (unsigned)(2 * P_STAGE + P_BIAS_MAX * 4 + 16) + (unsigned)(((it_j & 1) * 3 + slot) * P_SCK * 4 + half * 1024));
      }
      it_buf ^= 1;
      if (++it_kc == nkc) {
        it_kc = 0;
        it_j++;
        it_tile = __builtin_amdgcn_readfirstlane(tileq[it_j & 3]);   // (published at least one barrier ago; made scalar again)
        it_rb = it_tile / g.n_cb; it_cb = it_tile - it_rb * g.n_cb;
        it_live = it_tile < n_tiles;
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  };

  // ---- fragment side ---------------------------------------------------------------------------------------------
  // lane (r, q) reads row (16-row tile base + r), logical chunk 4 grp + q of a slab: k = 16 grp + 4 q .. + 3, element s
  // feeds MFMA step s (the same (q, s) -> k map as k_gemm_wide).  Group 1 is chunk ^ 4: byte address ^ 64.
  const unsigned sw = (unsigned)((r >> 1) & 7);
  const unsigned xo0 = (unsigned)((wm * MT * 16 + r) * 128) + (((unsigned)q ^ sw) << 4);
  const unsigned wo0 = P_ABYTES + (unsigned)((wn * NT * 16 + r) * 128) + (((unsigned)q ^ sw) << 4);
  constexpr int FR = 16 * 128;   // bytes between 16-row fragments
  f32x4 acc[MT][NT], A0[MT], A1[MT], B[NT];   // (every tile's first five steps start their accumulators from zero)
#define RT_RDA(addr, a) do { a[0] = lds_read16f<0>(addr); a[1] = lds_read16f<FR>(addr); a[2] = lds_read16f<2 * FR>(addr); a[3] = lds_read16f<3 * FR>(addr); } while (0)
#define RT_RDB(addr, nt) B[nt] = lds_read16f<(nt) * FR>(addr)
  // 16 MFMAs of one weight fragment (16 output channels) against the wave's 64 pixels
  // ... starting the accumulators of that column tile from zero (first group of a tile: C = 0 in the first k step, so the
  // accumulator registers are dead between the previous tile's epilogue chunk and here)
#define RT_MF0(a, nt) do { _Pragma("unroll") for (int mt_ = 0; mt_ < MT; mt_++) \
    acc[mt_][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(B[nt][0], a[mt_][0], f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0); \
    _Pragma("unroll") for (int s_ = 1; s_ < 4; s_++) _Pragma("unroll") for (int mt_ = 0; mt_ < MT; mt_++) \
    acc[mt_][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(B[nt][s_], a[mt_][s_], acc[mt_][nt], 0, 0, 0); \
    __builtin_amdgcn_sched_barrier(0); } while (0)
#define RT_MF(a, nt) do { _Pragma("unroll") for (int s_ = 0; s_ < 4; s_++) _Pragma("unroll") for (int mt_ = 0; mt_ < MT; mt_++) \
    acc[mt_][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(B[nt][s_], a[mt_][s_], acc[mt_][nt], 0, 0, 0); \
    __builtin_amdgcn_sched_barrier(0); } while (0)

  // ---- prologue: slabs 0 and 1 requested, landed and visible; the first fragments requested ------------------------------
  dma_issue();
  dma_issue();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();   // (also publishes the bias)
  unsigned cur_b = lds_b;
  if (HALF) RT_RDA(cur_b + xo0, A1); else RT_RDA(cur_b + xo0, A0);
  RT_RDB(cur_b + wo0, 0);
  RT_RDB(cur_b + wo0, 1);
  __builtin_amdgcn_sched_barrier(0);
  bool after_epi = false;   // the next slab wait follows an epilogue whose stores may stay in flight

  // DBG & 8: s_memtime stamps of one wave (diagnostic instantiation only; sums over the kernel go to g.epi.am_max)
  constexpr bool ST = (DBG & 8) != 0;
  const bool st_on = ST && blockIdx.x == 7 && wid == (int)g.epi.am_tiles;
  unsigned long long st_t[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_sum[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define RT_ST(i) do { if (ST) { if (st_on) st_t[i] = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); } } while (0)
  // The slab hand-over (before the last two weight fragments of a slab are multiplied): this wave has all its fragments
  // of the current slab in registers and its part of the next slab has landed; after the barrier that holds for every
  // wave, so the current buffer goes to the slab after the next one and the next slab's first fragments can be read.
  // The request for the slab after the next one goes out behind the first 16 MFMAs after the barrier (all waves leave the
  // barrier together: with the requests first no wave of a SIMD has an MFMA to issue).  Measured alternatives, in shader
  // cycles per workgroup (19 tiles, K = N = 240; 2.19 M is the bare MFMA count): requests right behind the barrier 2.55 M,
  // here 2.55 M, at the start of the next slab 2.58 M -- the placement is not what the requests cost.
  // fetched / pub: wave 0, lane 0 holds the counter value an atomic issued at the start of this slab returns (tile id - G)
  auto handover = [&](unsigned fetched, bool pub) __attribute__((always_inline)) {
    RT_ST(1);
    if (after_epi) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(P_NSTORE) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    after_epi = false;
    if (pub) {   // (the atomic was issued eight steps ago: the wait above covered it)
      if (wid == 0 && lane_id() == 0) tileq[pub_slot] = tile_of(q_wgs + (int)fetched);
    }
    RT_ST(2);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    RT_ST(3);
  };
  auto st_slab_end = [&]() __attribute__((always_inline)) {
    if (ST) {
      if (st_on) {
        st_t[5] = __builtin_amdgcn_s_memtime();
        for (int i = 0; i < 5; i++) st_sum[i] += st_t[i + 1] - st_t[i];
        st_sum[7] += 1;
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  // ---- epilogue of the PREVIOUS tile, one 16-channel column tile at a time: bias / activation / LAB, 16-byte stores (a lane
  // holds 4 consecutive channels of a pixel).  The chunks ride in front of the first five MFMA steps of the next tile (each
  // frees the accumulators that step starts from zero), so a wave's epilogue VALU work overlaps the other waves' MFMAs
  // instead of all twelve waves sitting in their epilogues at once.
  const unsigned mt_step = (unsigned)(16 * g.ldc * 4);
  char* pend_cbase = nullptr;   // tile whose accumulators are waiting for their epilogue
  int pend_n0 = 0, pend_rows = 0;
  auto epi_chunk = [&](auto nttag) __attribute__((always_inline)) {
    constexpr int nt = decltype(nttag)::value;
    // (opaque per use: otherwise hipcc precomputes the 20 store offsets as 64-bit values ahead of the tile loop, spills
    //  them and reloads them -- with a vmcnt(0) each -- in every epilogue; the lane id is re-derived for the same reason)
    int ln = lane_id();
    asm volatile("" : "+v"(ln));
    const int rowb = wm * MT * 16 + (ln & 15);
    const unsigned lo = (unsigned)((rowb * g.ldc + wn * NT * 16 + (ln >> 4) * 4) * 4);
    const f32x4 bias = *reinterpret_cast<const f32x4*>(bias_l + pend_n0 + nt * 16 + (ln >> 4) * 4);
#pragma unroll
    for (int mt = 0; mt < MT; mt++) {
      f32x4 o;
      if constexpr (ACT == ACT_HSWISH && (LAB == 0 || LAB == 1) && !(DBG & 4)) {
        // the same expressions and rounding as epi_val, written on the 4-vector: packed add / mul / fma (2 values per VALU op --
        // an fp32 MFMA loop pays every VALU cycle of its epilogue) and one v_med3 per value for the clamp
        const f32x4 v = acc[mt][nt] + bias;
        f32x4 t = v + 3.0f;
#pragma unroll
        for (int j = 0; j < 4; j++) { const float tj = t[j]; t[j] = __builtin_amdgcn_fmed3f(tj, 0.0f, 6.0f); }
        o = v * t;
        o = o * 0.16666667f;
        if (LAB == 1) { const f32x4 a4 = {g.epi.lab_a, g.epi.lab_a, g.epi.lab_a, g.epi.lab_a}, c4 = {g.epi.lab_c, g.epi.lab_c, g.epi.lab_c, g.epi.lab_c}; o = __builtin_elementwise_fma(o, a4, c4); }
      } else {
#pragma unroll
        for (int j = 0; j < 4; j++) o[j] = (DBG & 4) ? acc[mt][nt][j] : epi_val<ACT, LAB>(acc[mt][nt][j] + bias[j], g.epi.act, g.epi.has_lab, g.epi.lab_a, g.epi.lab_c);
      }
      if (DBG & 4) { if (o[0] == 123.456f) *reinterpret_cast<f32x4*>(pend_cbase + (lo + mt * mt_step + nt * 64)) = o; }
      else if (!(DBG & 1) || o[0] == 123.456f) { if (rowb < pend_rows - mt * 16) *reinterpret_cast<f32x4*>(pend_cbase + (lo + mt * mt_step + nt * 64)) = o; }
      __builtin_amdgcn_sched_barrier(0);   // (one pixel fragment at a time: interleaved, the four fragments' temporaries spill accumulators)
    }
  };

  // ASC: the pixel fragments of group grp of slab kc times the squeeze-excite scales of their rows' images
  const float* sc_tab = reinterpret_cast<const float*>(smem32p + 2 * P_STAGE + P_BIAS_MAX * 4 + 16);
  int sc_j = 0, sc_b1 = 0x7fffffff, sc_b2 = 0x7fffffff;   // current tile: table parity, tile-relative first rows of images 1 and 2
  auto scale_a = [&](f32x4 (&a)[MT], int kc, int grp) __attribute__((always_inline)) {
    if (ASC) {
      int ln = lane_id();
      asm volatile("" : "+v"(ln));
      // (one table read per fragment at a per-lane address -- the image slot of the fragment's row -- instead of three vectors
      //  and selects: 4 temporaries instead of 16; lanes of a quarter mostly share the address: LDS broadcast)
      const float* t = sc_tab + (sc_j & 1) * 3 * P_SCK + kc * KC + grp * 16 + (ln >> 4) * 4;
      const int rowb = wm * MT * 16 + (ln & 15);
      if (sc_b1 >= P_BM) {   // (uniform) the whole row block lies in one image -- the majority: no per-row slot arithmetic
        const f32x4 s0 = *reinterpret_cast<const f32x4*>(t);
#pragma unroll
        for (int mt = 0; mt < MT; mt++) a[mt] *= s0;
        __builtin_amdgcn_sched_barrier(0);
      } else {
#pragma unroll
        for (int mt = 0; mt < MT; mt++) {
          const int row = rowb + mt * 16;
          const int slot = (row >= sc_b1 ? 1 : 0) + (row >= sc_b2 ? 1 : 0);
          a[mt] *= *reinterpret_cast<const f32x4*>(t + slot * P_SCK);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
  };
  // One 32-deep slab = 10 steps of 16 MFMAs (5 weight fragments x 2 groups).  Weight fragments roll through B[0..4] two
  // steps ahead of their use, the pixel fragments of group 1 are read during group 0 and those of the next slab's group 0
  // during the last two steps; every wait leaves exactly the younger reads in flight (LDS returns in order).
  // COPY: the previous slab was a half slab, whose successor's pixel fragments went to A1 (A0 was still in use).
  // EPI: first slab of a tile -- the previous tile's epilogue chunks (if there is one: pend_cbase) precede the group-0 steps.
  auto slab_full = [&](auto copy, auto epi, int kc) __attribute__((always_inline)) {
    constexpr bool COPY = decltype(copy)::value, EPI = decltype(epi)::value;
    const unsigned nxt_b = lds_b + ((cur_b - lds_b) ^ P_STAGE);
    const unsigned wa0 = cur_b + wo0, wa1 = wa0 ^ 64u, xa1 = (cur_b + xo0) ^ 64u;
    RT_ST(0);
    unsigned fetched = 0;   // (live inside this slab only: a spill right behind the asm would store the register before the atomic returns)
    bool pub = false;
    if (!EPI && fetch_now) {
      if (wid == 0 && lane_id() == 0)
        asm volatile("s_nop 4\n\tglobal_atomic_add %0, %1, %2, %3 sc0" : "=v"(fetched) : "v"((unsigned)(xq * 64)), "v"(1u), "s"(g.sched) : "memory");
      fetch_now = false; pub = true;
      __builtin_amdgcn_sched_barrier(0);
    }
    RT_RDB(wa0, 2);
    if (EPI && pend_cbase) epi_chunk(IntTag<0>{});
    lgkm_wait<2>();
    if (COPY) {
#pragma unroll
      for (int i = 0; i < MT; i++) A0[i] = A1[i];
    }
    scale_a(A0, kc, 0);
    if (EPI) RT_MF0(A0, 0); else RT_MF(A0, 0);
    if (!EPI) {
      RT_RDB(wa0, 3); RT_RDA(xa1, A1); lgkm_wait<6>(); RT_MF(A0, 1);
      RT_RDB(wa0, 4); lgkm_wait<6>(); RT_MF(A0, 2);
      RT_RDB(wa1, 0); lgkm_wait<6>(); RT_MF(A0, 3);
      RT_RDB(wa1, 1); lgkm_wait<2>(); RT_MF(A0, 4);
      RT_RDB(wa1, 2); lgkm_wait<2>(); scale_a(A1, kc, 1); RT_MF(A1, 0);
    } else {
      // (with the epilogue chunks in front of the steps the group-1 pixel fragments are read late, at step 4: A1 is then
      //  dead while the chunks' temporaries are live -- read at step 1 the slab spills accumulators)
      RT_RDB(wa0, 3); if (pend_cbase) epi_chunk(IntTag<1>{}); lgkm_wait<2>(); RT_MF0(A0, 1);
      RT_RDB(wa0, 4); if (pend_cbase) epi_chunk(IntTag<2>{}); lgkm_wait<2>(); RT_MF0(A0, 2);
      RT_RDB(wa1, 0); if (pend_cbase) epi_chunk(IntTag<3>{}); lgkm_wait<2>(); RT_MF0(A0, 3);
      RT_RDB(wa1, 1); if (pend_cbase) epi_chunk(IntTag<4>{}); RT_RDA(xa1, A1); lgkm_wait<6>(); RT_MF0(A0, 4);
      RT_RDB(wa1, 2); lgkm_wait<1>(); scale_a(A1, kc, 1); RT_MF(A1, 0);
    }
    RT_RDB(wa1, 3); lgkm_wait<2>(); RT_MF(A1, 1);
    RT_RDB(wa1, 4); lgkm_wait<2>(); RT_MF(A1, 2);
    lgkm_wait<0>();
    handover(fetched, pub);
    RT_MF(A1, 3);
    RT_RDA(nxt_b + xo0, A0); RT_RDB(nxt_b + wo0, 0);
    __builtin_amdgcn_sched_barrier(0);
    dma_issue();   // the slab after the next one, into the buffer every wave has just left
    RT_ST(4);
    RT_RDB(nxt_b + wo0, 1);
    __builtin_amdgcn_sched_barrier(0);
    RT_MF(A1, 4);
    st_slab_end();
    cur_b = nxt_b;
  };
  auto slab_half = [&](int kc) __attribute__((always_inline)) {   // group 0 only; the next slab's pixel fragments go to A1
    const unsigned nxt_b = lds_b + ((cur_b - lds_b) ^ P_STAGE);
    const unsigned wa0 = cur_b + wo0;
    RT_ST(0);
    RT_RDB(wa0, 2); lgkm_wait<2>(); scale_a(A0, kc, 0); RT_MF(A0, 0);
    RT_RDB(wa0, 3); lgkm_wait<2>(); RT_MF(A0, 1);
    RT_RDB(wa0, 4); lgkm_wait<2>(); RT_MF(A0, 2);
    lgkm_wait<0>();
    handover(0u, false);
    RT_MF(A0, 3);
    RT_RDA(nxt_b + xo0, A1); RT_RDB(nxt_b + wo0, 0);
    __builtin_amdgcn_sched_barrier(0);
    dma_issue();
    RT_ST(4);
    RT_RDB(nxt_b + wo0, 1);
    __builtin_amdgcn_sched_barrier(0);
    RT_MF(A0, 4);
    st_slab_end();
    cur_b = nxt_b;
  };
  // the slabs of one tile; FIRST: no tile precedes it (nothing to store yet)
  auto tile = [&](int j, int rb, int cb) __attribute__((always_inline)) {
    RT_ST(6);
    // The first slab carries the previous tile's 20 stores per wave (steps 0-4); at its hand-over they are the youngest
    // vector-memory operations of the wave, behind the slab request it waits for: exactly they may stay in flight.
    // (The stores of a partial row block may be skipped by whole waves: then nothing is assumed to be in flight.)
    after_epi = pend_cbase != nullptr && pend_rows == P_BM;
    if (ASC) {
      const long long m0s = (long long)rb * P_BM;
      sc_j = j;
      sc_b1 = (int)min((long long)0x7fffffff, max(0ll, (long long)g.a_tab[3 * rb + 1] - m0s));
      sc_b2 = (int)min((long long)0x7fffffff, max(0ll, (long long)g.a_tab[3 * rb + 2] - m0s));
    }
    slab_full(std::integral_constant<bool, HALF>{}, std::true_type{}, 0);
    if (ST) { if (st_on) { st_sum[5] += st_t[1] - st_t[6]; st_sum[6] += 1; } __builtin_amdgcn_sched_barrier(0); }
    // The id of tile j + 1 of this workgroup: one returning atomic of wave 0, issued at the start of the tile's second slab
    // and published at that slab's hand-over, whose vmcnt(0) covers its return.  (Inline asm: through the builtin hipcc waits
    // vmcnt(0) right behind the atomic, i.e. for the slab requests just issued; and not in the first slab, where the
    // result register would be live across the epilogue chunks.)
    fetch_now = true; pub_slot = (j + 1) & 3;
    // (nkc >= 4: at least two middle slabs.  Register allocation here is at its limit and sensitive to the loop form: the
    //  ASC instantiations spill ~100 VGPRs with the top-tested loop -- the zero-trip path's copies -- and none with the
    //  bottom-tested one; the plain K = 32 j instantiations are the other way round, by 4)
    if constexpr (ASC) {
      int kc = 1;
      do { slab_full(std::false_type{}, std::false_type{}, kc); kc++; } while (kc + 1 < nkc);
    } else {
      for (int kc = 1; kc + 1 < nkc; kc++) slab_full(std::false_type{}, std::false_type{}, kc);
    }
    if (HALF) slab_half(nkc - 1); else slab_full(std::false_type{}, std::false_type{}, nkc - 1);
    const long long m0 = (long long)rb * P_BM;
    pend_cbase = reinterpret_cast<char*>(g.C + m0 * g.ldc + g.coff + cb * P_BN);
    pend_n0 = cb * P_BN + wn * NT * 16;
    pend_rows = (int)min((long long)P_BM, g.M - m0);
  };
  {   // (every workgroup has a first tile: the grid is at most n_tiles; exit at the bottom keeps the accumulators in place)
    int j = 0, t = first_tile;
    do {
      const int rb = t / g.n_cb;
      tile(j, rb, t - rb * g.n_cb);
      j++;
      t = __builtin_amdgcn_readfirstlane(tileq[j & 3]);
    } while (t < n_tiles);
  }
  // the last tile's epilogue
  epi_chunk(IntTag<0>{}); epi_chunk(IntTag<1>{}); epi_chunk(IntTag<2>{}); epi_chunk(IntTag<3>{}); epi_chunk(IntTag<4>{});
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (the fragments read ahead for a tile that does not exist)
  // the last workgroup to finish leaves the two counters at zero for the next launch on this stream
  if (tid == 0) {
    const unsigned done = __hip_atomic_fetch_add(g.sched + 128, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (done == (unsigned)G - 1) {
      for (int qq = 0; qq < 8; qq++) __hip_atomic_store(g.sched + 16 * qq, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(g.sched + 128, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  if ((DBG & 16) && wid == 0 && lane == 0) {
    unsigned long long* o = reinterpret_cast<unsigned long long*>(g.epi.am_max);
    if (blockIdx.x == 7) { o[0] = __builtin_amdgcn_s_memtime() - ck_t0; o[1] = __builtin_amdgcn_s_memrealtime() - ck_r0; }
    o[8 + 2 * blockIdx.x] = ck_r0; o[9 + 2 * blockIdx.x] = __builtin_amdgcn_s_memrealtime();   // per block: [start, end] in 10 ns ticks
  }
  if (ST && st_on && lane == 0) {
    unsigned long long* o = reinterpret_cast<unsigned long long*>(g.epi.am_max);
    for (int i = 0; i < 8; i++) o[i] = st_sum[i];
  }
#undef RT_ST
#undef RT_RDA
#undef RT_RDB
#undef RT_MF
#undef RT_MF0
}

// ---------------------------------------------------------------------------------------------------------------------
// k_gemm32w (round 5): the 128 -> 128-channel 1x1 convolution of the recognition net's first stage (2.4 M pixels per C3 step:
// `gemm_pw/thin`, 0.93 ms at 0.55 of the MFMA peak on the register-staged 128 x 128 tile, whose four 32-deep slabs per tile cannot
// amortise a prologue, two barriers per slab and an epilogue).  Here the WHOLE weight matrix (128 x 128 fp32 = 64 KB) is resident
// in LDS for the life of a persistent workgroup and the pixel operand arrives as complete 64-row x 128-deep tiles by LDS-DMA
// (buffer form: one per-lane offset, scalar slab / tile advances, rows beyond M as zeros) into a ring of two: there is no K
// loop to pipeline at all -- tile t + 1 lands while tile t is multiplied, ONE barrier per tile.  8 waves = 4 (16 pixel rows each)
// x 2 (64 channels each).  Same fragment maps and per-accumulator K order as k_gemm<8>: bit-identical results.
// Measured against it and removed: every wave streaming its OWN 16-row tiles through a private two-buffer ring (4 waves, no
// workgroup barrier after the weights have landed, counted vmcnt): 0.97-0.99 ms against 0.80-0.82 -- one wave per SIMD does not
// keep the matrix pipe fed through its own epilogue and waits; two per SIMD need 192 KB of LDS in that form.
// ---------------------------------------------------------------------------------------------------------------------
struct GemmWArgs {
  const float* A; const float* Wp; float* C;
  long long M;
  int lda, ldc, coff, n_tiles;
  Epilogue epi;
};
constexpr int W_BM = 64, W_NTHR = 512;
constexpr unsigned W_WBYTES = 128 * 128 * 4, W_ABYTES = W_BM * 128 * 4;
constexpr size_t W_LDS = W_WBYTES + 2 * (size_t)W_ABYTES + 128 * 4;   // weights | two pixel tiles | bias

__global__ __launch_bounds__(W_NTHR, 1) void k_gemm32w(const GemmWArgs g) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem32w[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 15, q = lane >> 4;
  const int wm = wid >> 1, wn = wid & 1;
  const unsigned lds_b = __builtin_amdgcn_readfirstlane(lds_addr32(smem32w));
  float* bias_l = reinterpret_cast<float*>(smem32w + W_WBYTES + 2 * W_ABYTES);
  if (tid < 128) bias_l[tid] = g.epi.bias ? g.epi.bias[tid] : 0.f;
  const int G = gridDim.x;
  const unsigned pitch = (unsigned)(g.lda * 4);
  // per-lane request offsets: lane i of a 1-KB piece = row i / 8 of the piece, physical chunk i % 8 <- logical chunk ^ ((row >> 1) & 7);
  // every piece a wave requests starts at a row that is a multiple of 8 with the parity of the wave id: one offset per operand
  const int rsub = lane >> 3, c0 = lane & 7;
  const unsigned ch = (unsigned)((c0 ^ (((rsub >> 1) | ((wid & 1) << 2)) & 7)) << 4);
  const unsigned req_w = (unsigned)(rsub * 128) + ch;
  const unsigned req_a = (unsigned)(8 * wid + rsub) * pitch + ch;
  {   // the weights, once: 64 pieces (slab p / 16, rows 8 (p % 16) ..), wave w takes p = w + 8 i; packed [slab][n][32] in global
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.Wp), 0, 0x7fffffffu, 0x00020000);
#pragma unroll
    for (int i = 0; i < 8; i++) {
      const unsigned p = (unsigned)(wid + 8 * i);
      blds16(req_w, wrs, lds_b + p * 1024u, p * 1024u);
    }
  }
  auto issue_a = [&](int t, unsigned buf) __attribute__((always_inline)) {   // pixel tile t -> ring buffer buf: 4 slabs x rows 8 wid .. + 7
    const long long m0 = (long long)t * W_BM;
    const unsigned rows_here = (unsigned)min((long long)W_BM, g.M - m0);
    const __amdgpu_buffer_rsrc_t ars = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.A + m0 * g.lda), 0, rows_here * pitch, 0x00020000);
    const unsigned dst = lds_b + W_WBYTES + buf * W_ABYTES + (unsigned)wid * 1024u;
#pragma unroll
    for (int s = 0; s < 4; s++) blds16(req_a, ars, dst + s * 8192u, s * 128u);
  };
  int t = blockIdx.x;
  issue_a(t, 0u);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  // fragment addresses: lane (r, q) reads row (16-row tile base + r), logical chunk 4 grp + q; group 1 = byte address ^ 64
  const unsigned sw = (unsigned)((r >> 1) & 7);
  const unsigned fo = (unsigned)(r * 128) + (((unsigned)q ^ sw) << 4);
  const unsigned char* wfrag = smem32w + (wn * 64) * 128;
  const float* bl = bias_l + wn * 64 + q * 4;
  act_dispatch(g.epi.act, g.epi.has_lab, false, [&](auto at, auto lt, auto) {
    constexpr int ACT = decltype(at)::value, LAB = decltype(lt)::value;
    unsigned buf = 0;
    for (; t < g.n_tiles; t += G, buf ^= 1u) {
      const int tn = t + G;
      const bool more = tn < g.n_tiles;
      if (more) issue_a(tn, buf ^ 1u);
      __builtin_amdgcn_sched_barrier(0);
      const unsigned char* afrag = smem32w + W_WBYTES + buf * W_ABYTES + (wm * 16) * 128;
      f32x4 acc[4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
      for (int s = 0; s < 4; s++)
#pragma unroll
        for (int grp = 0; grp < 2; grp++) {
          const unsigned fg = grp ? fo ^ 64u : fo;   // (the chunk field of the address: bits 4..6)
          const f32x4 a = *reinterpret_cast<const f32x4*>(afrag + s * 8192 + fg);
#pragma unroll
          for (int nt = 0; nt < 4; nt++) {
            const f32x4 b = *reinterpret_cast<const f32x4*>(wfrag + s * 16384 + nt * 2048 + fg);
#pragma unroll
            for (int k = 0; k < 4; k++) acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[k], a[k], acc[nt], 0, 0, 0);
          }
        }
      // epilogue: bias / activation / LAB, 16-byte stores (a lane holds 4 consecutive channels of a pixel)
      const long long m0 = (long long)t * W_BM;
      const long long row = m0 + wm * 16 + r;
      const bool full = m0 + W_BM <= g.M;
      float* crow = g.C + row * g.ldc + g.coff + wn * 64 + q * 4;
#pragma unroll
      for (int nt = 0; nt < 4; nt++) {
        const f32x4 bias = *reinterpret_cast<const f32x4*>(bl + nt * 16);
        f32x4 o;
#pragma unroll
        for (int j = 0; j < 4; j++) o[j] = epi_val<ACT, LAB>(acc[nt][j] + bias[j], g.epi.act, g.epi.has_lab, g.epi.lab_a, g.epi.lab_c);
        if (full || row < g.M) *reinterpret_cast<f32x4*>(crow + nt * 16) = o;
      }
      __builtin_amdgcn_sched_barrier(0);
      // the next tile has landed (this wave's requests; the 4 stores behind them may stay in flight -- a partial tile's
      // stores may be skipped by whole waves: then nothing is assumed)
      if (more) {
        if (full) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      __syncthreads();   // ... and everybody's; every wave is done with this tile's buffer
    }
  });
}

bool gemm_w_supported(int lda, long long M, int K, int N, int Npad16, const Epilogue& epi, int ldc, int coff) {
  static const int on = getenv("RT_GEMM_W") ? atoi(getenv("RT_GEMM_W")) : 1;   // A/B: 0 = k_gemm<8>
  if (!on || epi.am_max || epi.residual || epi.a_scale) return false;
  if (K != 128 || N != 128 || Npad16 != 128) return false;
  if (lda < 128 || (lda & 3) || (long long)lda * 4 * W_BM >= (1ll << 31)) return false;
  if ((ldc & 3) || (coff & 3)) return false;   // 16-byte stores of four consecutive channels
  return M >= 65536;
}

void gemm_w(hipStream_t st, const float* A, int lda, long long M, int K, const float* Wp, int N, int Npad16, float* C, int ldc, int coff,
            const Epilogue& epi) {
  GemmWArgs g;
  g.A = A; g.Wp = Wp; g.C = C; g.M = M; g.lda = lda; g.ldc = ldc; g.coff = coff; g.epi = epi;
  g.n_tiles = (int)((M + W_BM - 1) / W_BM);
  const int grid = std::min(g.n_tiles, stream_cus(st));
  allow_big_lds((const void*)k_gemm32w, 160 * 1024);
  RT_LAUNCH(k_gemm32w, dim3((unsigned)grid), dim3(W_NTHR), W_LDS, st, g);
}

bool gemm_dma_supported(int lda, long long M, int K, int N, int Npad16, const Epilogue& epi) {
  if (epi.am_max || epi.residual) return false;
  // squeeze-excite scale: 3-int row-block table (gemm_se_tile_rows() == 256), hardswish epilogue, K within the LDS scale table
  if (epi.a_scale && (epi.a_tab_stride != 3 || !epi.a_tab || K > P_SCK || epi.act != ACT_HSWISH || epi.n_img <= 0)) return false;
  if ((N + 3) / 4 * 4 != N) return false;
  if (Npad16 != N || N % P_BN != 0 || N > P_BIAS_MAX) return false;
  // (>= 4 slabs: the request side reads the next tile's id when it has issued a tile's last slab, at the hand-over of the
  //  tile's slab nkc - 3; the id is published at the hand-over of slab 1)
  if (K % 16 != 0 || K <= 3 * KC || lda < round_up(K, KC) || (lda & 3)) return false;   // whole 16-deep groups; 32-deep slabs readable
  if ((long long)lda * 4 * P_BM >= (1ll << 31)) return false;
  return M >= P_BM;
}

void gemm_dma(hipStream_t st, const float* A, int lda, long long M, int K, const float* Wp, int N, int Npad16, float* C,
              int ldc, int coff, const Epilogue& epi) {
  int dev = 0;
  RT_HIP_CHECK(hipGetDevice(&dev));
  static std::mutex mu;
  unsigned* sched = nullptr;
  {
    static std::map<std::pair<int, hipStream_t>, unsigned*> counters;   // zero between launches (the kernel resets them)
    std::lock_guard<std::mutex> lk(mu);
    unsigned*& c = counters[{dev, st}];
    if (!c) {
      RT_HIP_CHECK(hipMalloc((void**)&c, 1024));
      RT_HIP_CHECK(hipMemset(c, 0, 1024));
    }
    sched = c;
  }
  GemmPArgs g;
  g.sched = sched;
  g.A = A; g.Wp = Wp; g.C = C; g.M = M; g.lda = lda; g.K = K; g.N = N; g.Npad = Npad16; g.ldc = ldc; g.coff = coff;
  g.n_rb = (int)((M + P_BM - 1) / P_BM); g.n_cb = N / P_BN; g.epi = epi;
  g.a_scale = epi.a_scale; g.a_tab = epi.a_tab; g.ld_scale = epi.ld_scale; g.n_img = epi.n_img;
  const int grid = std::min(g.n_rb * g.n_cb, stream_cus(st));   // (one workgroup per CU of the stream's CU partition)
  // per-XCD tile queues (RT_G32P_XCDQ=0: one queue): needs every workgroup's first tile to exist in its queue
  // Measured (round 4, same box, alternating runs): bit-identical results; in isolation 1.5 % SLOWER (615216 x 480 x 480: 2.311 vs
  // 2.274 ms -- eight queues of 32 workgroups balance worse than one queue of 256), production step 27.90 / 27.91 / 28.41 / 27.90
  // vs 27.99 / 28.53 / 28.69 / 28.15 ms (inside the noise): opt-in.
  static const int xcdq_env = getenv("RT_G32P_XCDQ") ? atoi(getenv("RT_G32P_XCDQ")) : 0;
  g.xcdq = (xcdq_env && g.n_rb >= 64 && grid % 8 == 0) ? 1 : 0;
  const bool half = K % KC != 0;   // (supported K are whole 16-deep groups)
#define RT_G32P(ACTV, LABV) do { if (half) { allow_big_lds((const void*)k_gemm32p<ACTV, LABV, true>, 160 * 1024); RT_LAUNCH((k_gemm32p<ACTV, LABV, true>), dim3((unsigned)grid), dim3(P_NTHR), P_LDS, st, g); } \
                                 else { allow_big_lds((const void*)k_gemm32p<ACTV, LABV, false>, 160 * 1024); RT_LAUNCH((k_gemm32p<ACTV, LABV, false>), dim3((unsigned)grid), dim3(P_NTHR), P_LDS, st, g); } } while (0)
  static const int dbg = getenv("RT_G32P_DBG") ? atoi(getenv("RT_G32P_DBG")) : 0;   // timing experiments only (wrong results)
  if (dbg) {
    static unsigned long long* dst = nullptr;
    if (!dst) RT_HIP_CHECK(hipMalloc((void**)&dst, 64 + 16 * 1024));
    RT_HIP_CHECK(hipMemsetAsync(dst, 0, 64 + 16 * 1024, st));
    g.epi.am_max = reinterpret_cast<float*>(dst);
    g.epi.am_tiles = getenv("RT_G32P_WAVE") ? atoi(getenv("RT_G32P_WAVE")) : 0;
#define RT_DBG_LAUNCH(D) case D: allow_big_lds((const void*)k_gemm32p<ACT_HSWISH, 1, true, D>, 160 * 1024); \
                                 RT_LAUNCH((k_gemm32p<ACT_HSWISH, 1, true, D>), dim3((unsigned)grid), dim3(P_NTHR), P_LDS, st, g); break;
    switch (dbg) {
      RT_DBG_LAUNCH(1) RT_DBG_LAUNCH(2) RT_DBG_LAUNCH(4) RT_DBG_LAUNCH(6) RT_DBG_LAUNCH(8)
      RT_DBG_LAUNCH(16) RT_DBG_LAUNCH(17) RT_DBG_LAUNCH(18) RT_DBG_LAUNCH(20) RT_DBG_LAUNCH(22) RT_DBG_LAUNCH(23)
      default: throw RtError(8, "gemm_dma: unknown RT_G32P_DBG");
    }
#undef RT_DBG_LAUNCH
    unsigned long long h[8 + 2048];
    RT_HIP_CHECK(hipMemcpyAsync(h, dst, sizeof(h), hipMemcpyDeviceToHost, st));
    RT_HIP_CHECK(hipStreamSynchronize(st));
    if (dbg & 16) {
      unsigned long long t0 = ~0ull, t1 = 0;
      for (int b = 0; b < grid; b++) { t0 = std::min(t0, h[8 + 2 * b]); t1 = std::max(t1, h[9 + 2 * b]); }
      std::string line = "g32p blocks (start / end in us after the first start, by block): ";
      char buf[64];
      for (int b = 0; b < grid; b += std::max(1, grid / 32)) { snprintf(buf, sizeof buf, "%d:%.0f/%.0f ", b, (h[8 + 2 * b] - t0) / 100.0, (h[9 + 2 * b] - t0) / 100.0); line += buf; }
      double s_end = 0, mx = 0, mn = 1e30;
      for (int b = 0; b < grid; b++) { const double e = (h[9 + 2 * b] - t0) / 100.0; s_end += e; mx = std::max(mx, e); mn = std::min(mn, e); }
      fprintf(stderr, "%s\n  span %.1f us; block end times min %.1f mean %.1f max %.1f us\n", line.c_str(), (t1 - t0) / 100.0, mn, s_end / grid, mx);
    }
    if ((dbg & 16) && h[1]) fprintf(stderr, "g32p clock: %llu shader cycles in %.1f us = %.3f GHz\n", h[0], h[1] / 100.0, h[0] / (h[1] * 10.0));
    if ((dbg & 8) && h[7]) fprintf(stderr, "g32p stamps (wave %d of block 7): %llu slabs, per slab: steps0-7 %llu | lgkm+vmcnt wait %llu | barrier %llu | step 8 + dma issue %llu | step 9 %llu ; %llu first slabs with epilogue chunks: steps 0-7 %llu cycles\n",
                                   g.epi.am_tiles, h[7], h[0] / h[7], h[1] / h[7], h[2] / h[7], h[3] / h[7], h[4] / h[7], h[6], h[6] ? h[5] / h[6] : 0);
    return;
  }
  if (epi.a_scale) {
#define RT_G32P_SE(LABV) do { if (half) { allow_big_lds((const void*)k_gemm32p<ACT_HSWISH, LABV, true, 0, true>, 160 * 1024); RT_LAUNCH((k_gemm32p<ACT_HSWISH, LABV, true, 0, true>), dim3((unsigned)grid), dim3(P_NTHR), P_LDS_ASC, st, g); } \
                              else { allow_big_lds((const void*)k_gemm32p<ACT_HSWISH, LABV, false, 0, true>, 160 * 1024); RT_LAUNCH((k_gemm32p<ACT_HSWISH, LABV, false, 0, true>), dim3((unsigned)grid), dim3(P_NTHR), P_LDS_ASC, st, g); } } while (0)
    if (epi.has_lab) RT_G32P_SE(1); else RT_G32P_SE(0);
#undef RT_G32P_SE
    return;
  }
  if (epi.act == ACT_HSWISH && epi.has_lab) RT_G32P(ACT_HSWISH, 1);
  else if (epi.act == ACT_HSWISH) RT_G32P(ACT_HSWISH, 0);
  else RT_G32P(-1, -1);
#undef RT_G32P
}

}  // namespace nn
}  // namespace rt
