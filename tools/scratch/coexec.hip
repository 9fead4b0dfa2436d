// Does the VALU run under an executing fp32 MFMA (same wave / other wave of the SIMD)?  hipcc --offload-arch=gfx950 -O3 coexec.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
// MODE 0: 64 MFMAs per iteration; 1: 128 independent v_fma per iteration; 2: both interleaved (2 VALU after each MFMA);
// 3: both, block-wise (64 MFMAs then 128 VALU)
template <int MODE, int F16>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
  f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  float v[8];
  for (int i = 0; i < 8; i++) v[i] = threadIdx.x * 0.001f + i;
  const float a = threadIdx.x * 0.5f, b = 1.0001f;
  h8 ha, hb;
  for (int i = 0; i < 8; i++) { ha[i] = (_Float16)(threadIdx.x * 0.01f + i); hb[i] = (_Float16)1.0f; }
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int j = 0; j < 64; j++) {
      if (MODE == 0 || MODE == 2 || MODE == 3) { if (F16) acc[j & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha, hb, acc[j & 3], 0, 0, 0); else acc[j & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[j & 3], 0, 0, 0); }
      if (MODE == 1 || MODE == 2) {
        asm volatile("v_fma_f32 %0, %0, %2, %1" : "+v"(v[(2 * j) & 7]) : "v"(a), "v"(b));
        asm volatile("v_fma_f32 %0, %0, %2, %1" : "+v"(v[(2 * j + 1) & 7]) : "v"(a), "v"(b));
      }
    }
    if (MODE == 3) {
#pragma unroll
      for (int j = 0; j < 128; j++) asm volatile("v_fma_f32 %0, %0, %2, %1" : "+v"(v[j & 7]) : "v"(a), "v"(b));
    }
  }
  float s = 0;
  for (int i = 0; i < 8; i++) s += v[i];
  for (int i = 0; i < 4; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
  float* out; hipMalloc(&out, 256 * 4096 * 4);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  const int iters = 2000;
  auto run = [&](auto kern, const char* name, int blocks_per_cu) {
    for (int r = 0; r < 2; r++) { hipEventRecord(a); hipLaunchKernelGGL(kern, dim3(256 * blocks_per_cu), dim3(256), 0, 0, out, iters); hipEventRecord(b); hipEventSynchronize(b); }
    float ms; hipEventElapsedTime(&ms, a, b);
    // per SIMD: blocks_per_cu waves, each iters * 64 MFMAs (32 cycles) and / or iters * 128 VALU (4 cycles)
    printf("%-44s %d wave(s)/SIMD: %8.3f ms = %7.1f cycles per iteration and wave at 2.1 GHz\n", name, blocks_per_cu, ms, ms * 1e-3 * 2.1e9 / iters / blocks_per_cu);
  };
  for (int w : {1, 2, 3}) {
    run(k<0, 0>, "64 MFMA 16x16x4 f32 (2048 cyc)", w);
    run(k<1, 0>, "128 v_fma (512 cyc)", w);
    run(k<2, 0>, "64 MFMA f32 + 128 v_fma interleaved", w);
    run(k<3, 0>, "64 MFMA f32 then 128 v_fma", w);
    run(k<0, 1>, "64 MFMA 16x16x32 f16", w);
    run(k<2, 1>, "64 MFMA f16 + 128 v_fma interleaved", w);
    run(k<3, 1>, "64 MFMA f16 then 128 v_fma", w);
  }
  return 0;
}
