// buffer_load_dwordx4 ... lds on gfx950: lane i's 16 bytes land at M0 + 16 i?  Out-of-range lanes write zeros?  soffset is added?
//   hipcc --offload-arch=gfx950 -O2 bufdma.hip -o bufdma && ./bufdma
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(const float* x, float* y, int n_floats, unsigned so_bytes) {
  extern __shared__ float lds[];
  for (int i = threadIdx.x; i < 1024; i += blockDim.x) lds[i] = -7.f;
  __syncthreads();
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x), 0, (unsigned)n_floats * 4, 0x00020000);
  // lane i reads element block (63 - i) (reversed, to see that the LDS slot follows the LANE, not the address); lanes 5 and 9 out of range
  unsigned voff = (63 - threadIdx.x) * 16;
  if (threadIdx.x == 5 || threadIdx.x == 9) voff = 0x80000000u;
  unsigned ldsb = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds + 256;   // land at byte 256
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %3 offen lds" ::"v"(voff), "s"(rs), "s"(__builtin_amdgcn_readfirstlane(ldsb)), "s"(so_bytes) : "memory", "m0");
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = threadIdx.x; i < 1024; i += blockDim.x) y[i] = lds[i];
}
int main() {
  const int n = 4096;
  std::vector<float> h(n);
  for (int i = 0; i < n; i++) h[i] = (float)i;
  float *dx, *dy;
  hipMalloc(&dx, n * 4); hipMalloc(&dy, 1024 * 4);
  hipMemcpy(dx, h.data(), n * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 4096, 0, dx, dy, n, 1024u);   // soffset 1024 bytes = 256 floats
  std::vector<float> o(1024);
  hipMemcpy(o.data(), dy, 1024 * 4, hipMemcpyDeviceToHost);
  printf("err %s\n", hipGetErrorString(hipGetLastError()));
  int bad = 0;
  for (int lane = 0; lane < 64; lane++)
    for (int e = 0; e < 4; e++) {
      const float got = o[64 + lane * 4 + e];
      const float want = (lane == 5 || lane == 9) ? 0.f : (float)(256 + (63 - lane) * 4 + e);
      if (got != want) { if (bad < 8) printf("lane %d e %d got %g want %g\n", lane, e, got, want); bad++; }
    }
  printf("before %g %g after %g : %d mismatches\n", o[62], o[63], o[64 + 256], bad);
  return bad != 0;
}
