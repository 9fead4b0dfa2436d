// Micro-benchmark: cost of the "MFMA layout" global access pattern (lane = 16 q + r: pixel r, 16-byte chunk q) against the
// coalesced one (lane l: chunk l), b128 loads and stores, pixel pitch 256 B.  hipcc --offload-arch=gfx950 -O3 ta_pattern.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int MODE>   // 0: coalesced, 1: MFMA layout (16 px x 64 B per wave instr), 2: MFMA layout, quad-transposed via LDS
__global__ __launch_bounds__(256) void k_copy(const f32x4* __restrict__ x, f32x4* __restrict__ y, long long n_chunks, int do_store, int do_load, int reps) {
  __shared__ f32x4 lds[256];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 15, q = lane >> 4;
  f32x4 acc = {0, 0, 0, 0};
  // each wave instr moves 1 KB = 64 chunks: 4 pixels (coalesced) or 16 pixels x 4 chunks of a 16-chunk pixel
  const long long wave_id = (long long)blockIdx.x * 4 + wave, n_waves = (long long)gridDim.x * 4;
  for (int rep = 0; rep < reps; rep++)
  for (long long base = wave_id * 64; base + 64 <= n_chunks; base += n_waves * 64) {
    long long idx;
    if (MODE == 0) idx = base + lane;
    else {
      // block of 16 pixels x 16 chunks = 256 chunks handled by 4 consecutive wave-iterations (g = 0..3)
      const long long blk = base / 256, g = (base / 64) & 3;
      idx = blk * 256 + r * 16 + g * 4 + q;
    }
    f32x4 v = {1.f, 2.f, 3.f, 4.f};
    if (do_load) v = x[idx];
    if (MODE == 2 && do_store) {
      lds[wave * 64 + r * 4 + q] = v;               // [pixel r][chunk q]
      __builtin_amdgcn_s_waitcnt(0xc07f);           // lgkmcnt(0)
      v = lds[wave * 64 + lane];                    // lane l: pixel l >> 2, chunk l & 3
      const long long blk = base / 256, g = (base / 64) & 3;
      idx = blk * 256 + (lane >> 2) * 16 + g * 4 + (lane & 3);
    }
    if (do_store) y[idx] = v; else acc += v;
  }
  if (!do_store && acc[0] == 12345.f) y[0] = acc;
}
int main() {
  const long long n = 1ll << 26;  // 64 M chunks = 1 GiB
  f32x4 *x, *y; hipMalloc(&x, n * 16); hipMalloc(&y, n * 16); hipMemset(x, 0, n * 16); hipMemset(y, 0, n * 16);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  long long nn = n; int reps = 1;
  auto run = [&](auto kern, const char* name, int st, int ld) {
    for (int it = 0; it < 2; it++) {
      hipEventRecord(a);
      hipLaunchKernelGGL(kern, dim3(256 * 16), dim3(256), 0, 0, x, y, nn, st, ld, reps);
      hipEventRecord(b); hipEventSynchronize(b);
    }
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("%-28s load %d store %d: %7.3f ms  %6.2f TB/s\n", name, ld, st, ms, (double)nn * reps * 16 * (st + ld) / ms / 1e9);
  };
  for (int pass = 0; pass < 2; pass++) {
  if (pass == 1) { nn = 1 << 20; reps = 64; printf("-- 16 MiB working set (cache resident), 64 passes\n"); }   // 1 M chunks = 16 MiB: 4 chunks-waves per wave
  for (int st = 0; st < 2; st++) for (int ld = 0; ld < 2; ld++) {
    if (!st && !ld) continue;
    run(k_copy<0>, "coalesced", st, ld);
    run(k_copy<1>, "mfma layout", st, ld);
    if (st) run(k_copy<2>, "mfma layout, LDS transpose", st, ld);
  }
  }
  return 0;
}
