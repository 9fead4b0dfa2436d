// Calibration of rocprofv3's FETCH_SIZE on gfx950 for the access shapes of the thin LCNetV3 kernels (round 6):
// the guide's "x 2" (128-byte requests tallied at 64 bytes) was calibrated on 16-byte-per-lane streaming reads of whole lines; a
// wave of k_lc_lds reads the 64-byte (16-channel) SLICE of each 128- / 192-byte pixel.  Every kernel below reads each byte of a
// 1 GiB buffer exactly once (far beyond the 256 MiB Infinity Cache), so counter / 2^30 is the factor of that shape.
//   hipcc --offload-arch=gfx950 -O3 fetch_calib.hip -o fetch_calib
//   rocprofv3 --pmc FETCH_SIZE --output-format csv -d out/f -o f -- ./fetch_calib
//   rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_BUBBLE_sum --output-format csv -d out/r -o r -- ./fetch_calib
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
// whole lines: lane i of a wave reads 16 bytes at 16 i of the wave's 1 KB (the guide's calibration shape)
__global__ __launch_bounds__(256) void k_read_lines(const f32x4* __restrict__ x, long long n16, float* sink) {
  f32x4 s = {0, 0, 0, 0};
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n16; i += (long long)gridDim.x * 256) s += x[i];
  if (s[0] + s[1] + s[2] + s[3] == 123.456f) *sink = s[0];
}
// slices: pixels of PIX bytes, a wave reads the 64-byte slice g of 16 consecutive pixels per instruction (lane = 4 * pixel + chunk),
// slices g = 0 .. PIX / 64 - 1 by different waves (blockIdx.y)
template <int PIX>
__global__ __launch_bounds__(256) void k_read_slices(const char* __restrict__ x, long long npix, float* sink) {
  const int g = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  f32x4 s = {0, 0, 0, 0};
  for (long long p0 = ((long long)blockIdx.x * 4 + wave) * 16; p0 < npix; p0 += (long long)gridDim.x * 64) {
    const long long p = p0 + (lane >> 2);
    if (p < npix) s += *reinterpret_cast<const f32x4*>(x + p * PIX + g * 64 + (lane & 3) * 16);
  }
  if (s[0] + s[1] + s[2] + s[3] == 123.456f) *sink = s[0];
}
int main() {
  const long long bytes = 1ll << 30;
  char* x; float* sink; hipMalloc(&x, bytes); hipMalloc(&sink, 4); hipMemset(x, 1, bytes);
  hipDeviceSynchronize();
  for (int rep = 0; rep < 2; rep++) {
    hipLaunchKernelGGL(k_read_lines, dim3(256 * 8), dim3(256), 0, 0, (const f32x4*)x, bytes / 16, sink);
    hipLaunchKernelGGL(k_read_slices<64>, dim3(256 * 8, 1), dim3(256), 0, 0, x, bytes / 64, sink);
    hipLaunchKernelGGL(k_read_slices<128>, dim3(256 * 4, 2), dim3(256), 0, 0, x, bytes / 128, sink);
    hipLaunchKernelGGL(k_read_slices<192>, dim3(256 * 4, 3), dim3(256), 0, 0, x, bytes / 192, sink);
    hipLaunchKernelGGL(k_read_slices<256>, dim3(256 * 2, 4), dim3(256), 0, 0, x, bytes / 256, sink);
  }
  hipDeviceSynchronize();
  printf("fetch_calib: 5 kernels x 2, each reads 2^30 bytes once\n");
  return 0;
}
