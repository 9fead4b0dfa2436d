// What a plain fp32 copy reaches on this part (the ceiling of the streaming kernels): hipcc --offload-arch=gfx950 -O3 copy_bw.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int U, int NT>
__global__ __launch_bounds__(256) void k_copy(const f32x4* __restrict__ x, f32x4* __restrict__ y, long long n) {
  const long long stride = (long long)gridDim.x * 256;
  long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  for (; i + (U - 1) * stride < n; i += U * stride) {
    f32x4 v[U];
#pragma unroll
    for (int u = 0; u < U; u++) v[u] = NT ? __builtin_nontemporal_load(x + i + u * stride) : x[i + u * stride];
#pragma unroll
    for (int u = 0; u < U; u++) { if (NT) __builtin_nontemporal_store(v[u], y + i + u * stride); else y[i + u * stride] = v[u]; }
  }
}
// block-contiguous: every workgroup copies one contiguous 64 KB x U chunk per iteration
template <int U>
__global__ __launch_bounds__(256) void k_copy_blk(const f32x4* __restrict__ x, f32x4* __restrict__ y, long long n) {
  for (long long base = (long long)blockIdx.x * 256 * U; base + 256 * U <= n; base += (long long)gridDim.x * 256 * U) {
    f32x4 v[U];
#pragma unroll
    for (int u = 0; u < U; u++) v[u] = x[base + u * 256 + threadIdx.x];
#pragma unroll
    for (int u = 0; u < U; u++) y[base + u * 256 + threadIdx.x] = v[u];
  }
}
int main() {
  const long long n = 1ll << 27;  // 2 GiB per buffer
  f32x4 *x, *y; hipMalloc(&x, n * 16); hipMalloc(&y, n * 16); hipMemset(x, 1, n * 16); hipMemset(y, 0, n * 16);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  auto run = [&](auto kern, const char* name, int blocks) {
    float best = 1e9;
    for (int it = 0; it < 4; it++) {
      hipEventRecord(a); hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, x, y, n); hipEventRecord(b); hipEventSynchronize(b);
      float ms; hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms;
    }
    printf("%-34s %6d blocks: %7.3f ms  %6.2f TB/s (read + write)\n", name, blocks, best, (double)n * 32 / best / 1e9);
  };
  for (int blocks : {256 * 4, 256 * 8, 256 * 16, 256 * 64}) {
    run(k_copy<1, 0>, "grid-stride, 1 x 16 B", blocks);
    run(k_copy<4, 0>, "grid-stride, 4 x 16 B in flight", blocks);
    run(k_copy<8, 0>, "grid-stride, 8 x 16 B in flight", blocks);
    run(k_copy<4, 1>, "grid-stride, 4 x 16 B, nontemporal", blocks);
    run(k_copy_blk<8>, "block-contiguous 32 KB", blocks);
  }
  hipMemcpyAsync(y, x, n * 16, hipMemcpyDeviceToDevice, 0); hipDeviceSynchronize();
  hipEventRecord(a); hipMemcpyAsync(y, x, n * 16, hipMemcpyDeviceToDevice, 0); hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b); printf("hipMemcpyAsync D2D: %7.3f ms  %6.2f TB/s\n", ms, (double)n * 32 / ms / 1e9);
  return 0;
}
