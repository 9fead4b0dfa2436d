// fp16 convolution kernels staged by LDS-DMA (global_load_lds) for gfx950: k_conv16v2 (3x3 / 1x3 / 3x1 layers) and k_gemm16
// (1x1 layers over a flat pixel list).  Same fragment maps, K order and epilogue as k_conv16 (nn_f16.hip).
#include "nn_f16_dev.h"

#include <algorithm>
#include <cstdlib>
#include <mutex>
#include <type_traits>

namespace rt {
namespace nh {

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
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"   // M0 is named as clobbered on purpose: nothing else in these kernels uses it
__device__ __forceinline__ void glds16(const void* g, void* l) {
  const unsigned la = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) char*)l);
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(g), "s"(la) : "memory", "m0");
}
#pragma clang diagnostic pop
#define RT_GLDS16(gp, lp) glds16((gp), (lp))

// Fragment reads and their waits by hand (k_gemm16p): inside a loop whose body holds branches (the counted-vmcnt switch,
// the conditional DMA request) hipcc falls back to lgkmcnt(0) before each MFMA group even for plain ds_reads, which waits
// for the prefetch issued just before.  As inline asm the reads are invisible to its wait-count pass; lds_wait<N>() leaves
// the newest N reads in flight and pins the order (sched_barrier: an MFMA has no memory operand, so a "memory" clobber
// alone does not keep it behind the wait).
__device__ __forceinline__ unsigned lds_addr(const void* p) { return (unsigned)(size_t)(__attribute__((address_space(3))) const char*)p; }
__device__ __forceinline__ h8 lds_read16(unsigned byte_addr) {
  h8 v;
  asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(byte_addr));
  return v;
}
template <int OFF>
__device__ __forceinline__ h8 lds_read16_imm(unsigned byte_addr) {   // address + compile-time offset in the instruction
  static_assert(OFF >= 0 && OFF < 65536, "16-bit offset field");
  h8 v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(byte_addr), "n"(OFF));
  return v;
}
// DMA request with the LDS address already in an SGPR (uniform by construction: no readfirstlane per request)
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
__device__ __forceinline__ void glds16_s(unsigned long long gaddr, unsigned lds_sgpr) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(gaddr), "s"(lds_sgpr) : "memory", "m0");
}
#pragma clang diagnostic pop
// LDS-DMA request through a buffer resource (round 5): lane i writes 16 bytes at M0 + 16 i; source = resource base + per-lane
// byte offset + scalar offset; a lane whose offset is out of range (0x80000000) writes ZEROS -- the padding pixels / rows cost no
// select, and the per-slab / per-row advance is one scalar operand instead of a 64-bit add per lane and request.
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
__device__ __forceinline__ void blds16(unsigned voff, __amdgpu_buffer_rsrc_t rs, unsigned lds_sgpr, unsigned soff) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %3 offen lds" ::"v"(voff), "s"(rs), "s"(lds_sgpr), "s"(soff) : "memory", "m0");
}
#pragma clang diagnostic pop
template <int N>
__device__ __forceinline__ void lds_wait() {
  static_assert(N >= 0 && N <= 15, "lgkmcnt is a 4-bit counter");
  asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory");
  __builtin_amdgcn_sched_barrier(0);
}

// Halo rows un-swizzled since round 5 (make EXTRA=-DRT_V2_HSWZ=1 restores the XOR swizzle of the chunk index with bits 2-3 of the
// pixel): the swizzle made the pixel-fragment reads conflict-free at ~4 address instructions per read, and the main loop is bound
// by instruction issue, not by the LDS (24 % busy): without it the reads are 2-way conflicting and the server det network runs
// 30.4 -> 29.8 ms per 32 pages.  (The weight rows keep their swizzle: its term is a per-lane constant.)
#ifndef RT_V2_HSWZ
#define RT_V2_HSWZ 0
#endif
struct ConvArgs2 {
  ConvArgs a;
  const half_t* zeros;   // >= 16 zero bytes in device memory: DMA source of padding pixels / channels
  int hbuf_halves;       // size of one halo buffer (halves, multiple of 8)
  int hbufs;             // 2: next slab's halo prefetched; 1: single buffer (large halos)
  unsigned hw_magic, tw_magic;   // ceil(2^32 / d) for d = halo width, tile width (0: d == 1): q = umulhi(p, magic), exact for p * d < 2^32
  int wslots;            // weight ring: 3 (row r + 2 requested while row r is multiplied) or 2 (row r + 1; the 9-tap rows)
  int bdma;              // 1: the layer's byte offsets fit the buffer-resource form of the requests
};
constexpr int V2_HMAX = 8;   // DMA instructions per thread for one halo tile (8 * 512 * 16 B = 64 KB)

// KW = taps per stage (RG kernel rows of KWR taps each: KW = RG * KWR); KWR = the real kernel width
// (The cross-stage pipelining of k_gemm16p -- barrier in the middle of a stage, the next stage's first fragments read before
// the stage ends -- was built for the row-wise 3x3 form too and measured 4 % SLOWER (43.8 vs 41.9 ms per C5 step): the row
// and halo requests then have one stage less to land than with the barrier at the end of the stage.)
// NW = waves per workgroup: 8 (round 2-4: ONE 512-pixel workgroup per CU, whose eight waves move in lockstep from barrier to
// barrier) or 4 (round 5: 256-pixel tiles in <= 80 KB of LDS, so that TWO independent workgroups share a CU -- each SIMD holds one
// wave of each -- and one workgroup's DMA waits and barriers are the other's MFMA time; the weight ring shrinks to fit: c2.wslots
// = 1 is a single weight buffer refilled between two barriers, the partner workgroup being the latency cover)
// BD: requests through buffer resources (whole 32-channel slabs only; compile time, so that the pointer form's state is not live
// beside it: as a run-time choice the 128-channel row-wise instance spilled 12 VGPRs)
template <int NTN, int KW, int KWR, int DOT = 0, int NW = 8, bool BD = false>
__global__ __launch_bounds__(64 * NW, 2) void k_conv16v2(const ConvArgs2 c2) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem2[];
  const ConvArgs& a = c2.a;
  const bool kstamp = RT_STAMP_ON(a.stamps && blockIdx.y == 0 && blockIdx.x == gridDim.x / 2 && threadIdx.x == 0);
  if (kstamp) a.stamps[4000] = __builtin_amdgcn_s_memtime();
  constexpr int NTHR = 64 * NW, NTP = 2, BN = 32 * NTN, ROW = KS;  // LDS row = 32 halves (64 bytes), un-padded
  const int TH = a.TH, TW = a.TW;
  // logical block coordinates (bx: tile x channel block, by: image)
  unsigned bx = blockIdx.x, by = blockIdx.y;
  if (a.xcd) {
    const unsigned lin = xcd_remap(blockIdx.y * gridDim.x + blockIdx.x, gridDim.x * gridDim.y);
    by = lin / gridDim.x; bx = lin - by * gridDim.x;
  }
  const ImgGeom go = a.gout[by];
  const int tiles_x = (go.W + TW - 1) / TW, tiles_y = (go.H + TH - 1) / TH;
  const int zb = bx % a.nzb, tile = bx / a.nzb;
  if (tile >= tiles_x * tiles_y) return;
  const ImgGeom gi = a.gin[by];
  const int ty0 = (tile / tiles_x) * TH, tx0 = (tile % tiles_x) * TW;
  const int nb0 = zb * BN;
  constexpr int RG = KW / KWR;   // kernel rows per stage
  static_assert(RG * KWR == KW, "a stage is whole kernel rows");
  const int HH = (TH - 1) * a.SH + a.KH, HW = (TW - 1) * a.SW + KWR;
  const int SPS = a.KH / RG;    // stages per 32-channel slab
  half_t* hbuf = reinterpret_cast<half_t*>(smem2);
  half_t* wring = hbuf + (size_t)c2.hbufs * c2.hbuf_halves;
  constexpr int wbuf_halves = ((KW * BN * 4 + NTHR - 1) / NTHR * NTHR) * 8;   // whole groups of NTHR DMA slots
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int r = lane & 31, h = lane >> 5;

  const int iy0 = ty0 * a.SH - a.PT, ix0 = tx0 * a.SW - a.PL;
  const int nslab = (a.Cin + KS - 1) / KS, nrows = nslab * SPS;
  const int hchunks = HH * HW * 4; constexpr int wchunks = KW * BN * 4;
  const half_t* xtile = a.x + gi.off * a.ldx + ((long long)iy0 * gi.W + ix0) * a.ldx;
  constexpr int WI = (KW * BN * 4 + NTHR - 1) / NTHR;   // DMA instructions per thread for a kernel row (3x3, BN 128: 3)
  int wsrc[WI];
#pragma unroll
  for (int i = 0; i < WI; i++) {
    const int e = tid + i * NTHR;
    wsrc[i] = -1;
    if (e < wchunks) {
      const int row = e >> 2, cl = (e & 3) ^ ((row >> 2) & 3);
      const int dx = row / BN, n = row - dx * BN;
      if (nb0 + n < a.Npad) wsrc[i] = (dx * a.Npad + nb0 + n) * KS + cl * 8;
    }
  }
  const size_t row_halves = (size_t)KW * a.Npad * KS;
  const int wave_slot = wid * 64 * 8;   // halves: this wave's 64 consecutive 16-byte slots inside a group of NTHR
  const int wslots = c2.wslots, D = wslots - 1;   // prefetch distance in rows (0: single buffer, refilled between two barriers)
  // Buffer-resource form of the requests (whole 32-channel slabs only: a partial last slab needs per-slab channel tests):
  // PMC + ISA showed ~120 of the ~235 non-MFMA VALU instructions of a stage building 64-bit request addresses and selecting the
  // zero page; here the per-lane byte offsets are fixed for the whole tile and the row / slab advance is a scalar operand.
  constexpr bool bdma = BD;
  // (the descriptors must live in SGPRs: everything they are built from is made scalar explicitly -- the image geometry comes
  //  from a load indexed by the remapped block id, which hipcc does not prove uniform)
  auto uni_ptr = [](const half_t* p_) {
    const unsigned long long v = (unsigned long long)p_;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return (half_t*)(((unsigned long long)hi << 32) | lo);
  };
  const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(uni_ptr(a.w), 0, __builtin_amdgcn_readfirstlane((unsigned)(nrows * row_halves * 2)), 0x00020000);
  // (halo: the descriptor starts at the TILE's origin -- possibly a row above the image -- with an open range; pixels outside the
  //  image carry the out-of-range mark themselves, every valid pixel's offset from the origin is >= 0)
  const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(uni_ptr(xtile), 0, 0x7fffffffu, 0x00020000);
  const unsigned lds0 = __builtin_amdgcn_readfirstlane(lds_addr(smem2));
  const unsigned wring_b = lds0 + (unsigned)c2.hbufs * (unsigned)c2.hbuf_halves * 2u + (unsigned)__builtin_amdgcn_readfirstlane(wid) * 1024u;   // this wave's slots of ring slot 0
  const unsigned hbuf_b = lds0 + (unsigned)__builtin_amdgcn_readfirstlane(wid) * 1024u;
  auto dma_wrow = [&](int rr) {         // kernel row rr -> ring slot rr % wslots
    if (bdma) {
      const unsigned dstb = __builtin_amdgcn_readfirstlane(wring_b + (unsigned)(rr % wslots) * (unsigned)(wbuf_halves * 2));
      const unsigned soff = __builtin_amdgcn_readfirstlane((unsigned)rr * (unsigned)(row_halves * 2));
#pragma unroll
      for (int i = 0; i < WI; i++) {
        if (i * NTHR >= wchunks) break;   // (uniform)
        blds16(wsrc[i] >= 0 ? (unsigned)wsrc[i] * 2u : 0x80000000u, wrs, dstb + (unsigned)i * (NTHR * 16), soff);
      }
      return;
    }
    half_t* dst = wring + (size_t)(rr % wslots) * wbuf_halves;
    const half_t* wg = a.w + (size_t)rr * row_halves;
#pragma unroll
    for (int i = 0; i < WI; i++) {
      if (i * NTHR >= wchunks) break;   // (uniform)
      const half_t* src = wsrc[i] >= 0 ? wg + wsrc[i] : c2.zeros;
      RT_GLDS16(src, dst + (size_t)i * NTHR * 8 + wave_slot);
    }
  };
  // prologue, ordered so that the requests are in flight while the rest of the index arithmetic runs: weight rows first
  // (their sources are ready), then the halo sources and the halo, then the fragment offsets and the accumulators
  dma_wrow(0);
  if (D > 1 && nrows > 1) dma_wrow(1);
  // per-thread DMA sources, computed once: slot e = tid + 512 * i of a buffer holds (row e >> 2, physical chunk e & 3),
  // i.e. the logical chunk (e & 3) ^ ((row >> 2) & 3) of that row
  int hsrc[V2_HMAX];
#pragma unroll
  for (int i = 0; i < V2_HMAX; i++) {
    const int e = tid + i * NTHR;
    hsrc[i] = -1;
    if (e < hchunks) {
      const int p = e >> 2, cl = RT_V2_HSWZ ? ((e & 3) ^ ((p >> 2) & 3)) : (e & 3);
      const int hy = c2.hw_magic ? (int)__umulhi((unsigned)p, c2.hw_magic) : p, hx = p - hy * HW;
      const int iy = iy0 + hy, ix = ix0 + hx;
      if (iy >= 0 && iy < gi.H && ix >= 0 && ix < gi.W) hsrc[i] = ((hy * gi.W + hx) * a.ldx + cl * 8) | (cl << 28);
    }
  }
  auto dma_halo = [&](int s) {          // slab s -> halo buffer s % hbufs
    if (bdma) {
      const unsigned dstb = __builtin_amdgcn_readfirstlane(hbuf_b + (unsigned)(s % c2.hbufs) * (unsigned)(c2.hbuf_halves * 2));
      const unsigned soff = __builtin_amdgcn_readfirstlane((unsigned)s * (KS * 2));
#pragma unroll
      for (int i = 0; i < V2_HMAX; i++) {
        if (i * NTHR >= hchunks) break;   // (uniform)
        blds16(hsrc[i] >= 0 ? (unsigned)(hsrc[i] & 0x0fffffff) * 2u : 0x80000000u, xrs, dstb + (unsigned)i * (NTHR * 16), soff);
      }
      return;
    }
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
  dma_halo(0);
  int pix[NTP], oys[NTP], oxs[NTP];
#pragma unroll
  for (int j = 0; j < NTP; j++) {
    int q = (wid * NTP + j) * 32 + r;
    int ty = c2.tw_magic ? (int)__umulhi((unsigned)q, c2.tw_magic) : q, tx = q - ty * TW;
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

  if (kstamp) a.stamps[4001] = __builtin_amdgcn_s_memtime();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  if (kstamp) a.stamps[4002] = __builtin_amdgcn_s_memtime();
  const int nw = (wchunks + NTHR - 1) / NTHR, nh = (hchunks + NTHR - 1) / NTHR;   // DMA instructions per row / per halo, per thread

  for (int rr = 0; rr < nrows; rr++) {
    const int s = rr / SPS, dy = rr - s * SPS;
    const int cvalid = min(KS, a.Cin - s * KS);
    const int ksteps = (cvalid + 15) >> 4;
    const bool stamp = RT_STAMP_ON(a.stamps && blockIdx.y == 0 && blockIdx.x == gridDim.x / 2 && tid == 0);
    if (stamp) { a.stamps[rr * 5 + 0] = __builtin_amdgcn_s_memtime(); a.stamps[rr * 5 + 1] = a.stamps[rr * 5 + 0]; }
    // ---- multiply kernel row rr: the k-steps (dx, 16 channels) of the row in one software-pipelined sequence -- the
    // fragments of step i + 1 are requested from LDS before the MFMAs of step i are issued, so only the first read of a
    // stage is exposed; the DMA requests for later stages go out behind the first MFMA group ----
    const half_t* wl = wring + (size_t)(rr % wslots) * wbuf_halves;
    const half_t* halo = hbuf + (size_t)(s % c2.hbufs) * c2.hbuf_halves;
    // (a slab with fewer than 32 real channels still runs both 16-deep k-steps: its LDS rows and the packed weights are
    // zero-filled, and a fixed step count keeps the sequence below straight-line code -- with the steps behind run-time
    // tests hipcc waits lgkmcnt(0) before every MFMA group, which also waits for the prefetch just issued)
    (void)ksteps;
    constexpr int NK = KW * 2;
    const int tap0 = dy * RG * HW;
    auto frags = [&](int it, h8 (&A)[NTN], h8 (&B)[NTP]) {
      const int dx = it >> 1, ks = it & 1;
      const int cl = ks * 2 + h;
      const half_t* wrow = wl + (size_t)(dx * BN + r) * ROW + ((cl ^ aswz) << 3);
#pragma unroll
      for (int i = 0; i < NTN; i++) A[i] = *reinterpret_cast<const h8*>(wrow + i * 32 * ROW);
#pragma unroll
      for (int j = 0; j < NTP; j++) {
        const int p = pix[j] + tap0 + (RG == 1 ? dx : (dx / KWR) * HW + dx % KWR);
        B[j] = *reinterpret_cast<const h8*>(halo + p * ROW + ((RT_V2_HSWZ ? (cl ^ ((p >> 2) & 3)) : cl) << 3));
      }
    };
    auto mfmas = [&](const h8 (&A)[NTN], const h8 (&B)[NTP]) {
#pragma unroll
      for (int i = 0; i < NTN; i++)
#pragma unroll
        for (int j = 0; j < NTP; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[i], B[j], acc[i][j], 0, 0, 0);
    };
    // (sched_barrier: hipcc otherwise sinks every fragment read down to its first use and waits lgkmcnt(0) there)
    // fragment ring of PF + 1 sets: the fragments of k-step it + PF are requested before the MFMAs of step it.  PF = 1.  (Round 4
    // measured PF = 2 -- a step's reads get two MFMA groups instead of one to come back from LDS -- on every form that has the
    // registers for a third set (NTN <= 3): 75.4 / 76.1 / 75.3 ms per C5 step against 73.0 / 73.7 / 73.8: the LDS read latency
    // is not what the k-steps wait for.)
    constexpr int PF = 1, NR = PF + 1;
    h8 Af[NR][NTN], Bf[NR][NTP];
    frags(0, Af[0], Bf[0]);
    frags(1, Af[1], Bf[1]);
    if (PF == 2) frags(2, Af[2 % NR], Bf[2 % NR]);
    __builtin_amdgcn_sched_barrier(0);
    mfmas(Af[0], Bf[0]);
    __builtin_amdgcn_sched_barrier(0);
    // ---- requests for later stages (the buffers they overwrite were last read before the barrier this wave just passed) ----
    // (round 4: letting the upper four waves -- each shares a SIMD with wave w - 4 and leaves the barrier with it -- make their
    //  requests two MFMA groups later, under the lower waves' MFMAs, changed nothing: 74.3 / 74.3 / 73.7 vs 74.1 / 74.6 / 73.7 ms
    //  per C5 step; removed)
    if (D > 0 && rr + D < nrows) dma_wrow(rr + D);
    bool halo_now = false;
    if (c2.hbufs == 2) { if (dy == 0 && s + 1 < nslab) { dma_halo(s + 1); halo_now = true; } }
    (void)halo_now;
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int it = 1; it < NK; it++) {
      if (it + PF < NK) frags(it + PF, Af[(it + PF) % NR], Bf[(it + PF) % NR]);
      __builtin_amdgcn_sched_barrier(0);
      mfmas(Af[it % NR], Bf[it % NR]);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (stamp) { a.stamps[rr * 5 + 2] = __builtin_amdgcn_s_memtime(); a.stamps[rr * 5 + 3] = a.stamps[rr * 5 + 2]; }
    if (rr + 1 == nrows) break;
    // ---- the next row's data must have landed: everything except the requests made in THIS iteration ----
    const int ns = (rr + 1) / SPS, ndy = (rr + 1) - ns * SPS;
    if (D == 0 || (c2.hbufs == 1 && ndy == 0)) {
      // single halo buffer / single weight buffer: every wave is done with the old contents only after the barrier; request and
      // wait here (exposed; with NW = 4 the CU's other workgroup multiplies meanwhile)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if (D == 0) dma_wrow(rr + 1);
      if (c2.hbufs == 1 && ndy == 0) dma_halo(ns);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      continue;
    }
    // outstanding and allowed to stay in flight: row rr + 2 (nw instructions, if requested); the halo requested in this
    // iteration is only needed KH rows later, but it was issued AFTER row rr + 2, so it may stay in flight as well
    // (with one-row kernels the halo requested in this iteration is needed by the very next row: nothing may stay in flight)
    // (ring of 2: row rr + 1 itself was requested in this iteration, before the halo: only that halo may stay in flight)
    const int keep = (halo_now && ndy == 0) ? 0 : ((D > 1 && rr + 2 < nrows) ? nw : 0) + (halo_now ? nh : 0);
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

  if (DOT) {   // PFHeadLocal's phase convs: the block holds all (64) channels, nothing goes through LDS
    dot_tile16<NTN, NTP>(a, acc, lane, oys, oxs, go, (int)by);
    return;
  }
  if (kstamp) a.stamps[4003] = __builtin_amdgcn_s_memtime();
  __builtin_amdgcn_s_barrier();   // every wave is done with the last stage's LDS (no DMA is in flight any more)
  if (kstamp) a.stamps[4005] = __builtin_amdgcn_s_memtime();
  store_tile16<NTN, NTP>(a, acc, reinterpret_cast<half_t*>(smem2) + (size_t)wid * epi_scratch_halves<NTN>(), lane, nb0, oys, oxs, go);
  if (kstamp) a.stamps[4006] = __builtin_amdgcn_s_memtime();
  if (kstamp) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); a.stamps[4004] = __builtin_amdgcn_s_memtime(); }
}

static const half_t* zero_page16() {   // per device: DMA source of padding (one allocation per process and device)
  static const half_t* z[16] = {nullptr};
  static std::mutex mu;   // (the lanes of a session reach this concurrently on their first conv)
  int dev = 0;
  RT_HIP_CHECK(hipGetDevice(&dev));
  if (dev < 0 || dev >= 16) throw RtError(8, "conv16: device index out of range");
  std::lock_guard<std::mutex> lk(mu);
  if (!z[dev]) {
    void* p = nullptr;
    RT_HIP_CHECK(hipMalloc(&p, 256));
    RT_HIP_CHECK(hipMemset(p, 0, 256));
    z[dev] = (const half_t*)p;
  }
  return z[dev];
}

// ---------------------------------------------------------------------------------------------------------------------
// k_gemm16: a 1x1 convolution over a flat list of M pixels, Y[M][N] = X[M][K] . W^T, as a 256-pixel x (64 * NTN)-channel
// tile per workgroup of 8 waves (2 over the channels x 4 over the pixels; a wave owns 32 * NTN channels x 64 pixels).
// K is walked in stages of 64 channels: both operands of a stage go global -> LDS by DMA into one of two stage buffers
// ([2 slabs][rows][32 halves], 64-byte rows, XOR-swizzled like k_conv16v2), the next stage is requested right behind the
// first MFMA group of the current one and has the whole stage (32 MFMAs per wave) to land; one vmcnt(0) + barrier per
// stage.  The four 16-deep k-steps of a stage are software-pipelined (fragments of step i + 1 requested before the MFMAs
// of step i).  The packed weights and the DMA zero source pad K to whole stages.
// ---------------------------------------------------------------------------------------------------------------------
struct GemmArgs16 {
  ConvArgs a;            // x, ldx, gin/gout (one flat image), Cin = K, w, N, Npad, y, ldy, coff, nzb, epi
  const half_t* zeros;
};

template <int NTN>
__global__ __launch_bounds__(512, 1) void k_gemm16(const GemmArgs16 g) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smemg[];
  const ConvArgs& a = g.a;
  constexpr int NTHR = 512, NTP = 2, BN = 64 * NTN, BP = 256, ROW = KS;
  constexpr int XCH = 2 * BP * 4 / NTHR, WCH = 2 * BN * 4 / NTHR;          // DMA instructions per thread and stage: 4 and NTN
  constexpr int XHALVES = 2 * BP * ROW, WHALVES = 2 * BN * ROW, STAGE = XHALVES + WHALVES;
  const ImgGeom gi = a.gin[0], go = a.gout[0];
  const long long M = go.W;
  const int zb = blockIdx.x % a.nzb;
  const long long m0 = (long long)(blockIdx.x / a.nzb) * BP;
  if (m0 >= M) return;
  const int nblk = zb * BN;
  half_t* lds = reinterpret_cast<half_t*>(smemg);
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wn = wid & 1, wp = wid >> 1;
  const int r = lane & 31, h = lane >> 5;
  const int aswz = (r >> 2) & 3;
  const int nb0 = nblk + wn * 32 * NTN;      // first channel of this wave

  f32x16 acc[NTN][NTP];
#pragma unroll
  for (int i = 0; i < NTN; i++)
#pragma unroll
    for (int j = 0; j < NTP; j++)
#pragma unroll
      for (int e = 0; e < 16; e++) acc[i][j][e] = 0.f;

  const int K = a.Cin, nslab = (K + KS - 1) / KS, nst = (nslab + 1) >> 1;
  // per-thread DMA sources (slot e = tid + 512 i of an operand's stage image: slab e / (rows * 4), row, physical chunk)
  const half_t* xsrc[XCH]; int xk[XCH];
#pragma unroll
  for (int i = 0; i < XCH; i++) {
    const int e = tid + i * NTHR, slab = e / (BP * 4), rem = e - slab * (BP * 4), row = rem >> 2, cl = (rem & 3) ^ ((row >> 2) & 3);
    xk[i] = slab * KS + cl * 8;
    xsrc[i] = (m0 + row < M) ? a.x + (gi.off + m0 + row) * a.ldx + xk[i] : nullptr;
  }
  const half_t* wsrc[WCH]; int wslab[WCH];
#pragma unroll
  for (int i = 0; i < WCH; i++) {
    const int e = tid + i * NTHR, slab = e / (BN * 4), rem = e - slab * (BN * 4), row = rem >> 2, cl = (rem & 3) ^ ((row >> 2) & 3);
    wslab[i] = slab;
    wsrc[i] = (nblk + row < a.Npad) ? a.w + ((size_t)slab * a.Npad + nblk + row) * KS + cl * 8 : nullptr;
  }
  const int wave_slot = wid * 64 * 8;
  auto dma_stage = [&](int s) {
    half_t* dst = lds + (size_t)(s & 1) * STAGE + wave_slot;
    const int k0 = s * 2 * KS;
#pragma unroll
    for (int i = 0; i < XCH; i++) {
      const half_t* src = (xsrc[i] && k0 + xk[i] < K) ? xsrc[i] + k0 : g.zeros;
      RT_GLDS16(src, dst + (size_t)i * NTHR * 8);
    }
#pragma unroll
    for (int i = 0; i < WCH; i++) {
      const half_t* src = (wsrc[i] && s * 2 + wslab[i] < nslab) ? wsrc[i] + (size_t)s * 2 * a.Npad * KS : g.zeros;
      RT_GLDS16(src, dst + XHALVES + (size_t)i * NTHR * 8);
    }
  };
  dma_stage(0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  for (int s = 0; s < nst; s++) {
    const half_t* xb = lds + (size_t)(s & 1) * STAGE;
    const half_t* wb = xb + XHALVES;
    auto frags = [&](int it, h8 (&A)[NTN], h8 (&B)[NTP]) {
      const int slab = it >> 1, cl = (it & 1) * 2 + h;
      const int co = (cl ^ aswz) << 3;
      const half_t* wrow = wb + (size_t)(slab * BN + wn * 32 * NTN + r) * ROW + co;
      const half_t* xrow = xb + (size_t)(slab * BP + wp * 64 + r) * ROW + co;
#pragma unroll
      for (int j = 0; j < NTP; j++) B[j] = *reinterpret_cast<const h8*>(xrow + j * 32 * ROW);
#pragma unroll
      for (int i = 0; i < NTN; i++) A[i] = *reinterpret_cast<const h8*>(wrow + i * 32 * ROW);
    };
    auto mfmas = [&](const h8 (&A)[NTN], const h8 (&B)[NTP]) {
#pragma unroll
      for (int i = 0; i < NTN; i++)
#pragma unroll
        for (int j = 0; j < NTP; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[i], B[j], acc[i][j], 0, 0, 0);
    };
    h8 Af[2][NTN], Bf[2][NTP];
    frags(0, Af[0], Bf[0]);
    frags(1, Af[1], Bf[1]);
    __builtin_amdgcn_sched_barrier(0);
    mfmas(Af[0], Bf[0]);
    __builtin_amdgcn_sched_barrier(0);
    if (s + 1 < nst) dma_stage(s + 1);   // into the buffer every wave finished reading before the last barrier
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int it = 1; it < 4; it++) {
      if (it + 1 < 4) frags(it + 1, Af[(it + 1) & 1], Bf[(it + 1) & 1]);
      __builtin_amdgcn_sched_barrier(0);
      mfmas(Af[it & 1], Bf[it & 1]);
      __builtin_amdgcn_sched_barrier(0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }
  // (the last barrier also says that every wave is done with the stage buffers: they become the transpose scratch)
  int oys[NTP], oxs[NTP];
#pragma unroll
  for (int j = 0; j < NTP; j++) {
    const long long m = m0 + wp * 64 + j * 32 + r;
    oys[j] = m < M ? 0 : -1;
    oxs[j] = (int)m;
  }
  store_tile16<NTN, NTP>(a, acc, lds + (size_t)wid * epi_scratch_halves<NTN>(), lane, nb0, oys, oxs, go);
}

template <int NTN>
static void launch_gemm16(hipStream_t st, const GemmArgs16& g, long long mtiles) {
  constexpr size_t lds = (size_t)2 * (2 * 256 * KS + 2 * 64 * NTN * KS) * 2;   // two stage buffers (>= the epilogue scratch)
  static_assert(lds >= (size_t)8 * (32 * (32 * NTN + 8) + 128) * 2, "epilogue scratch must fit in the stage buffers");
  allow_big_lds((const void*)k_gemm16<NTN>, 160 * 1024);
  RT_LAUNCH((k_gemm16<NTN>), dim3((unsigned)(mtiles * g.a.nzb)), dim3(512), lds, st, g);
}

// ---------------------------------------------------------------------------------------------------------------------
// k_gemm16p: the same tile as k_gemm16 with the K loop pipelined ACROSS the stage boundaries.  k_gemm16 ends every
// 64-channel stage with vmcnt(0) + barrier and then starts the next one with exposed fragment reads: the MFMA stream
// drains once per 32 MFMAs.  Here a stage is one 32-channel slab in a ring of R buffers and the one barrier of a stage
// sits in its MIDDLE:
//     stage s:   reads(s, step 1) | MFMAs(s, step 0) | vmcnt: slab s + 1 landed | BARRIER | request slab s + R - 1 |
//                reads(s + 1, step 0) | MFMAs(s, step 1)
//   * after the barrier every wave's part of slab s + 1 is visible, so its first fragments are read before the stage ends
//     and the MFMAs of consecutive stages follow each other without a wait for LDS;
//   * every wave that passed the barrier has its reads of slab s - 1 back in registers (they fed MFMAs issued before it),
//     so that buffer is free for slab s + R - 1; R - 2 slabs stay in flight across the barrier (counted vmcnt).
// ---------------------------------------------------------------------------------------------------------------------
template <int NTN, int R>
__global__ __launch_bounds__(512, 1) void k_gemm16p(const GemmArgs16 g) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smemp[];
  const ConvArgs& a = g.a;
  constexpr int NTHR = 512, NTP = 2, BN = 64 * NTN, BP = 256, ROW = KS;
  constexpr int XCH = BP * 4 / NTHR, WCH = (BN * 4 + NTHR - 1) / NTHR, PER = XCH + WCH;   // DMA instructions per thread and slab
  constexpr int XHALVES = BP * ROW, WSLOTS = WCH * NTHR, STAGE = XHALVES + WSLOTS * 8;
  constexpr unsigned STAGE_B = STAGE * 2;
  const ImgGeom gi = a.gin[0], go = a.gout[0];
  const long long M = go.W;
  const unsigned bx = a.xcd ? xcd_remap(blockIdx.x, gridDim.x) : blockIdx.x;
  const int zb = bx % a.nzb;
  const long long m0 = (long long)(bx / a.nzb) * BP;
  if (m0 >= M) return;
  const int nblk = zb * BN;
  half_t* lds = reinterpret_cast<half_t*>(smemp);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);   // (in an SGPR: everything derived from it stays scalar)
  const int wn = wid & 1, wp = wid >> 1;
  const int r = lane & 31, h = lane >> 5;
  const int aswz = (r >> 2) & 3;
  const int nb0 = nblk + wn * 32 * NTN;

  const int K = a.Cin, nst = (K + KS - 1) / KS, krem = K & (KS - 1);   // krem != 0: the last slab is partly beyond K
  // ---- the stage loop is instruction-bound (stamps: ~2000 cycles of one wave's instruction stream per 512 cycles of its MFMAs),
  // so everything per-stage is kept to one add: DMA sources are running 64-bit pointers with a per-thread step (0 for rows
  // beyond M / Npad, which stay on the zero page), LDS addresses are scalar, fragment reads use the instruction's offset field
  unsigned long long xcur[XCH], wcur[WCH];
  unsigned xstep[XCH], wstep[WCH];
  int xk[XCH];
  const unsigned long long zaddr = (unsigned long long)g.zeros;
#pragma unroll
  for (int i = 0; i < XCH; i++) {
    const int e = tid + i * NTHR, row = e >> 2, cl = (e & 3) ^ ((row >> 2) & 3);
    const bool ok = m0 + row < M;
    xk[i] = cl * 8;
    xcur[i] = ok ? (unsigned long long)(a.x + (gi.off + m0 + row) * a.ldx + cl * 8) : zaddr;
    xstep[i] = ok ? KS * 2 : 0;
  }
#pragma unroll
  for (int i = 0; i < WCH; i++) {
    const int e = tid + i * NTHR, row = e >> 2, cl = (e & 3) ^ ((row >> 2) & 3);
    const bool ok = row < BN && nblk + row < a.Npad;
    wcur[i] = ok ? (unsigned long long)(a.w + ((size_t)nblk + row) * KS + cl * 8) : zaddr;
    wstep[i] = ok ? (unsigned)a.Npad * KS * 2 : 0;
  }
  const unsigned lds_b = __builtin_amdgcn_readfirstlane(lds_addr(lds));
  const unsigned slot_b = lds_b + (unsigned)wid * 1024;     // this wave's 64 16-byte slots inside a group of 512
  unsigned issue_b = slot_b;                                  // ... of the ring buffer the next slab goes into
  int issued = 0;
  auto dma_slab = [&]() {   // requests slab `issued` (slabs are requested in order)
    const bool tail = krem && issued == nst - 1;
#pragma unroll
    for (int i = 0; i < XCH; i++) {
      glds16_s((tail && xk[i] >= krem) ? zaddr : xcur[i], issue_b + i * (NTHR * 16));
      xcur[i] += xstep[i];
    }
#pragma unroll
    for (int i = 0; i < WCH; i++) {
      glds16_s(wcur[i], issue_b + XHALVES * 2 + i * (NTHR * 16));
      wcur[i] += wstep[i];
    }
    issued++;
    issue_b = (issue_b + STAGE_B == slot_b + R * STAGE_B) ? slot_b : issue_b + STAGE_B;
    __builtin_amdgcn_sched_barrier(0);
  };
  auto wait_keep = [&](int slabs) {   // leave the newest `slabs` slabs in flight
    switch (slabs * PER) {
      case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
      case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
      case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
      case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
      case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
      case 9: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
      case 12: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
      default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
  };
  // prologue: slabs 0 .. R - 2 requested, slab 0 landed and visible
#pragma unroll
  for (int s = 0; s < R - 1; s++)
    if (s < nst) dma_slab();
  f32x16 acc[NTN][NTP];
#pragma unroll
  for (int i = 0; i < NTN; i++)
#pragma unroll
    for (int j = 0; j < NTP; j++)
#pragma unroll
      for (int e = 0; e < 16; e++) acc[i][j][e] = 0.f;
  wait_keep(max(0, min(nst - 1, R - 2)));
  __builtin_amdgcn_s_barrier();

  constexpr int NR = NTN + NTP;   // ds_read_b128 per k-step
  // per-lane byte offsets inside a stage buffer, by k-step: pixel rows (B operand) and weight rows (A operand)
  const unsigned xo0 = (unsigned)((wp * 64 + r) * ROW * 2) + ((unsigned)((0 + h) ^ aswz) << 4);
  const unsigned xo1 = (unsigned)((wp * 64 + r) * ROW * 2) + ((unsigned)((2 + h) ^ aswz) << 4);
  const unsigned wo0 = (unsigned)((XHALVES + (wn * 32 * NTN + r) * ROW) * 2) + ((unsigned)((0 + h) ^ aswz) << 4);
  const unsigned wo1 = (unsigned)((XHALVES + (wn * 32 * NTN + r) * ROW) * 2) + ((unsigned)((2 + h) ^ aswz) << 4);
  constexpr int FR = 32 * ROW * 2;   // bytes between fragments (32 rows)
  auto frags = [&](unsigned buf_b, int ks, h8 (&A)[NTN], h8 (&B)[NTP]) {
    const unsigned xa = buf_b + (ks ? xo1 : xo0), wa = buf_b + (ks ? wo1 : wo0);
    B[0] = lds_read16_imm<0>(xa);
    B[1] = lds_read16_imm<FR>(xa);
    A[0] = lds_read16_imm<0>(wa);
    if (NTN > 1) A[NTN > 1 ? 1 : 0] = lds_read16_imm<FR>(wa);
    if (NTN > 2) A[NTN > 2 ? 2 : 0] = lds_read16_imm<2 * FR>(wa);
    if (NTN > 3) A[NTN > 3 ? 3 : 0] = lds_read16_imm<3 * FR>(wa);
  };
  auto mfmas = [&](const h8 (&A)[NTN], const h8 (&B)[NTP]) {
#pragma unroll
    for (int i = 0; i < NTN; i++)
#pragma unroll
      for (int j = 0; j < NTP; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[i], B[j], acc[i][j], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
  };
  h8 A0[NTN], B0[NTP], A1[NTN], B1[NTP];
  unsigned cur_b = lds_b;   // buffer of the slab being multiplied
  frags(cur_b, 0, A0, B0);
  // one stage; STEADY: slab s + R - 1 exists (a constant number of requests stays in flight across the barrier)
  auto stage = [&](int s, auto steady) {
    constexpr bool STEADY = decltype(steady)::value;
    const unsigned nxt_b = (cur_b + STAGE_B == lds_b + R * STAGE_B) ? lds_b : cur_b + STAGE_B;
    frags(cur_b, 1, A1, B1);
    lds_wait<NR>();                                          // step 0's fragments are back, step 1's stay in flight
    mfmas(A0, B0);
    if (STEADY) {
      if (PER * (R - 3) == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else if (PER * (R - 3) == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
      else if (PER * (R - 3) == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else if (PER * (R - 3) == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
      wait_keep(max(0, min(nst - 1, s + R - 2) - (s + 1)));   // slab s + 1 has landed (this wave's requests)
    }
    __builtin_amdgcn_s_barrier();                            // ... and everybody's; every wave is done with slab s - 1
    if (STEADY) dma_slab();
    __builtin_amdgcn_sched_barrier(0);
    frags(nxt_b, 0, A0, B0);
    lds_wait<NR>();
    mfmas(A1, B1);
    cur_b = nxt_b;
  };
  int s = 0;
  for (; s + R - 1 < nst; s++) stage(s, std::true_type{});
  for (; s + 1 < nst; s++) stage(s, std::false_type{});
  frags(cur_b, 1, A1, B1);
  lds_wait<NR>();
  mfmas(A0, B0);
  lds_wait<0>();
  mfmas(A1, B1);
  __builtin_amdgcn_s_barrier();   // every wave is done with the ring: it becomes the transpose scratch
  int oys[NTP], oxs[NTP];
#pragma unroll
  for (int j = 0; j < NTP; j++) {
    const long long m = m0 + wp * 64 + j * 32 + r;
    oys[j] = m < M ? 0 : -1;
    oxs[j] = (int)m;
  }
  store_tile16<NTN, NTP>(a, acc, lds + (size_t)wid * epi_scratch_halves<NTN>(), lane, nb0, oys, oxs, go);
}

template <int NTN, int R>
static void launch_gemm16p(hipStream_t st, const GemmArgs16& g, long long mtiles) {
  constexpr int WCH = (64 * NTN * 4 + 511) / 512;
  constexpr size_t lds = (size_t)R * (256 * KS + WCH * 512 * 8) * 2;
  static_assert(lds <= 160 * 1024 && lds >= (size_t)8 * (32 * (32 * NTN + 8) + 128) * 2, "ring within LDS, epilogue scratch within the ring");
  allow_big_lds((const void*)k_gemm16p<NTN, R>, 160 * 1024);
  RT_LAUNCH((k_gemm16p<NTN, R>), dim3((unsigned)(mtiles * g.a.nzb)), dim3(512), lds, st, g);
}

bool conv16_dma(hipStream_t st, const ConvArgs& a0, int n_img, int maxHo, int maxWo) {
  const int KH = a0.KH, KW = a0.KW, SH = a0.SH, SW = a0.SW, Npad = a0.Npad, Cin = a0.Cin;
  // ---- 1x1 over one flat image: k_gemm16 when the channel blocks of 256 / 128 waste little ----
  if (KH == 1 && KW == 1 && SH == 1 && SW == 1 && a0.PT == 0 && a0.PL == 0 && n_img == 1 && maxHo == 1 && Cin >= 64 && maxWo >= 4096 && !a0.epi.dot_w) {
    // channel block of 64 * NTN with the least padded work; taken when at most 10 % of the block columns are padding
    // (N = 480 -> 2 x 256, 240 -> 256, 384 -> 2 x 192; 96 or 160 stay with k_conv16's 96 / 160-wide blocks)
    int bn = 0, best = 1 << 30;
    for (int cand : {256, 192, 128, 64}) {
      const int cost = (Npad + cand - 1) / cand * cand;
      if (cost < best) { best = cost; bn = cand; }
    }
    if (best * 10 > a0.N * 11) bn = 0;
    if (bn) {
      GemmArgs16 g;
      g.a = a0; g.a.nzb = (Npad + bn - 1) / bn; g.zeros = zero_page16();
      const long long mtiles = ((long long)maxWo + 255) / 256;
      if (mtiles * g.a.nzb < (1ll << 31)) {
        static const int pipe = getenv("RT_GEMM16_PIPE") ? atoi(getenv("RT_GEMM16_PIPE")) : 4;   // 0: k_gemm16 (64-deep stages, 256 / 128 only); 4 / 5: k_gemm16p ring depth
        if (pipe == 0 && (bn == 256 || bn == 128)) { if (bn == 256) launch_gemm16<4>(st, g, mtiles); else launch_gemm16<2>(st, g, mtiles); }
        else if (pipe == 5) {
          switch (bn / 64) { case 4: launch_gemm16p<4, 5>(st, g, mtiles); break; case 3: launch_gemm16p<3, 5>(st, g, mtiles); break;
                             case 2: launch_gemm16p<2, 5>(st, g, mtiles); break; default: launch_gemm16p<1, 5>(st, g, mtiles); break; }
        } else {
          switch (bn / 64) { case 4: launch_gemm16p<4, 4>(st, g, mtiles); break; case 3: launch_gemm16p<3, 4>(st, g, mtiles); break;
                             case 2: launch_gemm16p<2, 4>(st, g, mtiles); break; default: launch_gemm16p<1, 4>(st, g, mtiles); break; }
        }
        return true;
      }
    }
    return false;
  }
  // ---- the 3x3-class layers ----
  const bool dot = a0.epi.dot_w != nullptr;
  const bool k9 = !dot && KH == 9 && KW == 9 && SH == 1 && SW == 1 && Npad == 64;   // LKPAN's 9x9 layers (256 -> 64, 64 -> 64)
  const bool k22 = dot && KH == 2 && KW == 2 && SH == 1 && SW == 1 && Npad == 64;   // PFHeadLocal's 2x2 phase convs with the dot epilogue
  if (dot && !k22) return false;
  if (!(((KH <= 3 && (KW == 1 || KW == 3) && KH * KW > 1) || k9 || k22) && Cin >= 32 && n_img <= RT_MAX_GRID_Y)) return false;
  int bn2 = 32, best = 1 << 30;
  for (int bn : {128, 96, 64, 32}) {
    const int nb = (Npad + bn - 1) / bn, cost = nb * bn + 16 * nb;
    if (cost < best) { best = cost; bn2 = bn; }
  }
  // Workgroup size (round 5): two 4-wave workgroups per CU on 256-pixel tiles (<= 80 KB of LDS each) instead of one 8-wave
  // workgroup on a 512-pixel tile.  Measured per layer on C5 (tools/layer_profile.py, ms per step, 8 -> 4 waves): the
  // recognition maps 615216 x 1728 x 192: 6.29 -> 5.46, 1230432 x 1440 x 160: 4.92 -> 4.26, 1x3 neck convs 0.61 -> 0.35; the
  // 480^2 det maps 1.35 -> 1.22; the 240^2 / 120^2 maps 3.43 -> 3.42 / 1.44 -> 1.34 (with the tile-shape search below; 1.53
  // without); only the 9x9 rows lose (4.28 -> 5.17: their 37-KB weight rows leave a 4-wave workgroup one buffer) and stay on
  // 8 waves.  RT_CONV16_NW=8 / 4 forces one form (A/B runs).
  static const int nw_env = getenv("RT_CONV16_NW") ? atoi(getenv("RT_CONV16_NW")) : 0;
  static const int group3 = getenv("RT_CONV3_GROUP") ? atoi(getenv("RT_CONV3_GROUP")) : 1;
  int nw = nw_env == 4 ? 4 : (nw_env == 8 ? 8 : (k9 ? 8 : 4));
  // (4 waves: the nine-tap form also where the 128-channel blocks of the row-wise form would be partly empty -- N = 224:
  //  2.36 -> 2.08 ms per step for the five 307608 x 2016 x 224 layers)
  const bool g3 = KH == 3 && KW == 3 && Npad >= 64 && (group3 == 2 || (group3 == 1 && (bn2 < 128 || (nw == 4 && Npad % 128 != 0))));
  if (g3) bn2 = 64;
  const int taps = g3 ? 9 : (k22 ? 4 : KW);
  auto hpix = [&](int t_h, int t_w) { return ((t_h - 1) * SH + KH) * ((t_w - 1) * SW + KW); };
  auto magic = [](int d) { return d <= 1 ? 0u : (unsigned)((1ull << 32) / (unsigned)d + 1); };
  // one plan per workgroup size: tile, halo buffers, weight ring
  struct Plan { bool ok = false; int th = 0, tw = 0, hbuf_halves = 0, hbufs = 0, wslots = 0; size_t lds = 0; };
  auto plan = [&](int nw) {
    Plan p;
    const int nthr = 64 * nw, px = 64 * nw;
    const size_t cap = nw == 8 ? (size_t)160 * 1024 : (size_t)80 * 1024;
    // full-height tiles on short maps; on taller ones the tile shape that wastes the fewest of the workgroup's pixel slots on
    // the largest image (a 120 x 120 map under 15 x 17 tiles pays for 136 columns; 12 x 20 tiles fill 94 % of their slots),
    // ties (within 3 %) going to the shape with the smaller halo; the halo tile must fit V2_HMAX DMA instructions per thread
    static const int tile_search = getenv("RT_CONV16_TILES") ? atoi(getenv("RT_CONV16_TILES")) : 1;
    int th, tw;
    if (maxHo >= 16) { const int ny = (maxHo + 15) / 16; th = maxHo >= 64 ? 16 : (maxHo + ny - 1) / ny; } else th = std::max(maxHo, 1);
    tw = std::max(1, std::min(px / th, maxWo));
    if (tile_search && nw == 4 && maxHo >= 16) {   // (8 waves: the 16 x 32 tiles the kernel was tuned with measured better)
      double best_eff = 0, best_halo = 1e30;
      int bh = th, bw = tw;
      for (int h = 8; h <= std::min(32, maxHo); h++) {
        const int ny = (maxHo + h - 1) / h, hh = (maxHo + ny - 1) / ny;          // balanced rows
        int w = std::max(1, std::min(px / hh, maxWo));
        const int nx = (maxWo + w - 1) / w; w = (maxWo + nx - 1) / nx;            // balanced columns
        if (hpix(hh, w) * 4 > V2_HMAX * nthr) continue;
        const double eff = (double)maxHo * maxWo / ((double)ny * nx * px), halo = (double)hpix(hh, w) / (hh * w);
        if (eff > best_eff * 1.03 || (eff > best_eff * 0.97 && halo < best_halo)) { best_eff = std::max(eff, best_eff); best_halo = halo; bh = hh; bw = w; }
      }
      th = bh; tw = bw;
    }
    while (hpix(th, tw) * 4 > V2_HMAX * nthr && tw > 8) tw--;
    if (hpix(th, tw) * 4 > V2_HMAX * nthr) return p;
    p.th = th; p.tw = tw;
    p.hbuf_halves = ((hpix(th, tw) * 4 + nthr - 1) / nthr * nthr) * 8;
    const size_t wslot_bytes = (size_t)((taps * bn2 * 4 + nthr - 1) / nthr * nthr) * 16, hbytes = (size_t)p.hbuf_halves * 2;
    // 8 waves: the ring the kernel was tuned with (3 rows, 2 for the nine-tap stages), two halo buffers if they fit.
    // 4 waves: the deepest buffering within 80 KB -- weights first (a stage's weights are needed at its first k-step, the halo of
    // the NEXT slab only KH / RG stages later)
    const int want_w = (k9 || g3) ? 2 : 3;
    if (nw == 8) {
      p.wslots = want_w;
      p.hbufs = (2 * hbytes + wslot_bytes * p.wslots <= cap) ? 2 : 1;
    } else {
      p.wslots = 0;
      for (int ws = want_w; ws >= 1 && !p.wslots; ws--)
        for (int hb = 2; hb >= 1 && !p.wslots; hb--)
          if (hb * hbytes + ws * wslot_bytes <= cap) { p.wslots = ws; p.hbufs = hb; }
      if (!p.wslots) return p;
    }
    if (p.hbufs * hbytes + wslot_bytes * p.wslots > cap) return p;
    p.lds = std::max(p.hbufs * hbytes + wslot_bytes * p.wslots, (size_t)nw * (32 * (bn2 + 8) + 128) * 2);   // main loop | epilogue scratch
    if (p.lds > cap) return p;
    p.ok = true;
    return p;
  };
  Plan pl = plan(nw);
  if (!pl.ok && nw == 4) { nw = 8; pl = plan(8); }
  if (!pl.ok) return false;
  const int th = pl.th, tw = pl.tw;
  ConvArgs2 c2;
  c2.a = a0; c2.a.TH = th; c2.a.TW = tw; c2.a.lp = KS; c2.a.nzb = (Npad + bn2 - 1) / bn2;
  c2.zeros = zero_page16();
  c2.hw_magic = magic((tw - 1) * SW + KW); c2.tw_magic = magic(tw);
  c2.hbuf_halves = pl.hbuf_halves; c2.hbufs = pl.hbufs; c2.wslots = pl.wslots;
  // (byte offsets inside one image / the packed weights must stay below 2^31: the out-of-range marker is bit 31)
  c2.bdma = (long long)maxHo * SH * maxWo * SW * a0.ldx * 2 < (1ll << 31) ? 1 : 0;
  const size_t lds2 = pl.lds;
  const long long tiles2 = (long long)((maxWo + tw - 1) / tw) * ((maxHo + th - 1) / th);
  dim3 grid2((unsigned)(tiles2 * c2.a.nzb), (unsigned)n_img);
  // Buffer-resource requests are the only form of the dense instances (the pointer form survives in the PFHeadLocal phase convs,
  // whose 80 input channels are not whole slabs); a layer with a partial last slab or an image beyond the 2-GB offset range takes
  // the register-staged kernel.  (A/B of the two forms on C5, same box: 444 / 444 vs 452-460 images/s.)
  if (!k22 && ((Cin % KS) != 0 || !c2.bdma)) return false;
#define RT_V2_ONE(NT, KWV, KWRV, DOTV, NWV, BDV) do { allow_big_lds((const void*)k_conv16v2<NT, KWV, KWRV, DOTV, NWV, BDV>, (NWV) == 8 ? 160 * 1024 : 80 * 1024); \
                                                      RT_LAUNCH((k_conv16v2<NT, KWV, KWRV, DOTV, NWV, BDV>), grid2, dim3(64 * (NWV)), lds2, st, c2); } while (0)
#define RT_V2_GO(NT, KWV, KWRV) do { if (nw == 4) RT_V2_ONE(NT, KWV, KWRV, 0, 4, true); else RT_V2_ONE(NT, KWV, KWRV, 0, 8, true); } while (0)
#define RT_V2_LAUNCH(NT) do { if (KW == 1) RT_V2_GO(NT, 1, 1); else RT_V2_GO(NT, 3, 3); } while (0)
  if (k9) { RT_V2_GO(2, 9, 9); return true; }
  if (g3) { RT_V2_GO(2, 9, 3); return true; }
  if (k22) { if (nw == 4) RT_V2_ONE(2, 4, 2, 1, 4, false); else RT_V2_ONE(2, 4, 2, 1, 8, false); return true; }
  switch (bn2 / 32) {
    case 1: RT_V2_LAUNCH(1); break;
    case 2: RT_V2_LAUNCH(2); break;
    case 3: RT_V2_LAUNCH(3); break;
    default: RT_V2_LAUNCH(4); break;
  }
#undef RT_V2_LAUNCH
#undef RT_V2_GO
#undef RT_V2_ONE
  return true;
}

}  // namespace nh
}  // namespace rt
