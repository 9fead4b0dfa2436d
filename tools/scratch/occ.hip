#include <hip/hip_runtime.h>
#include <cstdio>
template <int KB> __global__ __launch_bounds__(256) void k(float* y) {
  __shared__ float s[KB * 256];
  s[threadIdx.x] = threadIdx.x; __syncthreads();
  y[threadIdx.x] = s[(threadIdx.x * 7) % (KB * 256)];
}
template <int KB> void q() { int n = 0; hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k<KB>, 256, 0); printf("LDS %3d KB -> %d blocks of 256 per CU\n", KB, n); }
int main() { q<8>(); q<16>(); q<20>(); q<24>(); q<32>(); q<33>(); q<36>(); q<40>(); q<48>(); q<52>(); q<53>(); q<54>(); q<64>();
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0); printf("sharedMemPerMultiprocessor %zu maxSharedMemoryPerMultiProcessor %zu regsPerMultiprocessor %d maxThreadsPerMultiProcessor %d\n", p.sharedMemPerBlock, p.maxSharedMemoryPerMultiProcessor, p.regsPerMultiprocessor, p.maxThreadsPerMultiProcessor); return 0; }
