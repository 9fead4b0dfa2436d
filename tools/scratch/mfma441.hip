// layout probe: v_mfma_f32_4x4x1_16B_f32
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void probe(float* out, int mode) {
  const int lane = threadIdx.x;
  // A lane value = 100 + lane, B lane value = 1 (mode 0) -> D shows which A lane feeds each (lane, vgpr)
  // mode 1: A = 1, B = 100 + lane -> which B lane
  float a = mode == 0 ? 100.f + lane : 1.f;
  float b = mode == 0 ? 1.f : 100.f + lane;
  f32x4 c = {0, 0, 0, 0};
  f32x4 d;
  if (mode < 2) d = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0);
  else { a = 100.f + lane; b = 1.f; d = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 4, 2, 0); }  // cbsz=4, abid=2: A of block 2 to all
  for (int i = 0; i < 4; i++) out[lane * 4 + i] = d[i];
}
int main() {
  float* d; hipMalloc(&d, 64 * 4 * 4);
  float h[256];
  for (int mode = 0; mode < 3; mode++) {
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, mode);
    hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    printf("mode %d\n", mode);
    for (int l = 0; l < 64; l += 1) if (l < 10 || l % 16 == 0 || l == 63) printf(" lane %2d: %g %g %g %g\n", l, h[l*4], h[l*4+1], h[l*4+2], h[l*4+3]);
  }
  return 0;
}
