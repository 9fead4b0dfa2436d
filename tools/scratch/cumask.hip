// CU-masked streams on MI355X: does hipExtStreamCreateWithCUMask work here, how do mask bits map to (XCD, CU), and what do an
// MFMA-bound and an HBM-bound kernel cost when they run side by side on disjoint CU sets instead of competing for all of them?
//   hipcc --offload-arch=gfx950 -O3 cumask.hip -o cumask && ./cumask
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <set>
#include <map>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void k_where(unsigned* out) {
  unsigned hw, xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  if (threadIdx.x == 0) { out[2 * blockIdx.x] = hw; out[2 * blockIdx.x + 1] = xcc; }
  // stay a little so that the blocks spread over everything the mask allows
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  while (__builtin_amdgcn_s_memtime() - t0 < 20000) {}
}
// MFMA-bound: 12 waves per workgroup (3 per SIMD), a persistent-style loop
__global__ __launch_bounds__(768, 1) void k_mfma(float* out, int iters) {
  f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  const float a = threadIdx.x * 0.5f, b = 1.0001f;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int j = 0; j < 64; j++) acc[j & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[j & 3], 0, 0, 0);
  }
  float s = 0;
  for (int i = 0; i < 4; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * 768 + threadIdx.x] = s;
}
// HBM-bound: every workgroup copies contiguous 64 KB pieces (grid-stride over pieces)
__global__ __launch_bounds__(256) void k_copy(const f32x4* __restrict__ src, f32x4* __restrict__ dst, long long pieces) {
  for (long long p = blockIdx.x; p < pieces; p += gridDim.x) {
    const f32x4* s = src + p * 4096;
    f32x4* d = dst + p * 4096;
    f32x4 v[16];
#pragma unroll
    for (int i = 0; i < 16; i++) v[i] = s[i * 256 + threadIdx.x];
#pragma unroll
    for (int i = 0; i < 16; i++) d[i * 256 + threadIdx.x] = v[i];
  }
}

int main() {
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  const int ncu = prop.multiProcessorCount;
  printf("device: %s, %d CUs\n", prop.name, ncu);
  const int words = (ncu + 31) / 32;
  unsigned* d_where; CK(hipMalloc(&d_where, 8192 * 8));
  std::vector<unsigned> h(8192 * 2);
  auto make_stream = [&](const std::vector<unsigned>& mask) { hipStream_t s; CK(hipExtStreamCreateWithCUMask(&s, (unsigned)mask.size(), mask.data())); return s; };
  // ---- 1. mapping (measured on the round-5 box with single-bit masks): bit i -> XCD i % 8, k = i / 8 -> shader engine k % 4,
  // CU slot k / 4; an XCD whose part of the mask is EMPTY falls back to 8 CUs (2 per SE), so every mask must keep >= 1 bit per XCD.
  auto where = [&](hipStream_t s, const char* name) {
    hipLaunchKernelGGL(k_where, dim3(4096), dim3(64), 0, s, d_where);
    CK(hipStreamSynchronize(s));
    CK(hipMemcpy(h.data(), d_where, 4096 * 8, hipMemcpyDeviceToHost));
    std::set<std::pair<unsigned, unsigned>> seen;
    std::map<unsigned, int> px;
    for (int b = 0; b < 4096; b++) seen.insert({h[2 * b + 1] & 15, (h[2 * b] >> 8) & 0xff});
    for (auto& p : seen) px[p.first]++;
    printf("%-28s %3zu distinct CUs; per XCD:", name, seen.size());
    for (auto& p : px) printf(" %d", p.second);
    printf("\n");
    return seen;
  };
  auto split = [&](int k, std::vector<unsigned>& small, std::vector<unsigned>& big) {
    small.assign(words, 0); big.assign(words, 0);
    for (int i = 0; i < ncu; i++) { if (i / 8 >= 32 - k) small[i / 32] |= 1u << (i % 32); else big[i / 32] |= 1u << (i % 32); }
  };
  {
    std::vector<unsigned> small, big; split(8, small, big);
    hipStream_t s1 = make_stream(small), s2 = make_stream(big);
    auto a = where(s1, "mask: 8 per XCD"), b = where(s2, "mask: the other 24 per XCD");
    int common = 0; for (auto& p : a) common += (int)b.count(p);
    printf("CUs in both partitions: %d\n", common);
    CK(hipStreamDestroy(s1)); CK(hipStreamDestroy(s2));
  }
  // ---- 2/3. timings
  float* d_out; CK(hipMalloc(&d_out, 768 * 4096 * 4));
  const long long pieces = 32768;   // x 64 KB = 2 GiB
  f32x4 *d_src, *d_dst; CK(hipMalloc(&d_src, pieces * 65536)); CK(hipMalloc(&d_dst, pieces * 65536));
  CK(hipMemset(d_src, 1, pieces * 65536));
  hipEvent_t e0, e1, e2, e3; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&e2)); CK(hipEventCreate(&e3));
  const int iters = 3000;   // per wave: 3000 x 64 MFMAs x 32 cycles x 3 waves/SIMD = 18.4 M cycles = ~8.5 ms
  auto t_mfma = [&](hipStream_t s, int wgs) {
    float ms = 0;
    for (int r = 0; r < 2; r++) { CK(hipEventRecord(e0, s)); hipLaunchKernelGGL(k_mfma, dim3(wgs), dim3(768), 0, s, d_out, iters); CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1)); }
    return ms;
  };
  auto t_copy = [&](hipStream_t s, int wgs, int reps) {
    float ms = 0;
    for (int r = 0; r < 2; r++) { CK(hipEventRecord(e0, s)); for (int q = 0; q < reps; q++) hipLaunchKernelGGL(k_copy, dim3(wgs), dim3(256), 0, s, d_src, d_dst, pieces); CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1)); }
    return ms / reps;
  };
  const int reps = 9;
  float mf_all = 0, cp_all = 0;
  auto both = [&](hipStream_t s_m, int wg_m, hipStream_t s_c, int wg_c, int reps, const char* name) {
    float ms_m = 0, ms_c = 0, wall = 0;
    for (int r = 0; r < 2; r++) {
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0, s_m)); CK(hipEventRecord(e2, s_c));
      hipLaunchKernelGGL(k_mfma, dim3(wg_m), dim3(768), 0, s_m, d_out, iters);
      for (int q = 0; q < reps; q++) hipLaunchKernelGGL(k_copy, dim3(wg_c), dim3(256), 0, s_c, d_src, d_dst, pieces);
      CK(hipEventRecord(e1, s_m)); CK(hipEventRecord(e3, s_c));
      CK(hipDeviceSynchronize());
      CK(hipEventElapsedTime(&ms_m, e0, e1)); CK(hipEventElapsedTime(&ms_c, e2, e3));
      float a; CK(hipEventElapsedTime(&a, e0, e3)); wall = a > ms_m ? a : ms_m;
    }
    printf("%-58s mfma %.3f ms, %d copies %.3f ms (%.2f TB/s), both done after %.3f ms; serial sum would be %.3f\n", name, ms_m, reps, ms_c,
           reps * 2.0 * pieces * 65536 / ms_c / 1e9, wall > ms_c ? wall : ms_c, mf_all + reps * cp_all);
  };
  {
    hipStream_t sa, sb; CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
    mf_all = t_mfma(sa, ncu);
    cp_all = t_copy(sb, ncu * 8, 4);
    printf("alone, all CUs: mfma %.3f ms (1 wg/CU x %d), copy 2 GiB %.3f ms = %.2f TB/s (r+w)\n", mf_all, ncu, cp_all, 2.0 * pieces * 65536 / cp_all / 1e9);
    both(sa, ncu, sb, ncu * 8, reps, "plain streams:");
    both(sa, ncu / 2, sb, ncu * 8, reps, "plain streams, mfma on 128 wgs only:");
    both(sa, 192, sb, 64 * 8, reps, "plain streams, mfma 192 wgs, copy 512 wgs:");
    hipStream_t sp0, sp1; CK(hipStreamCreateWithPriority(&sp0, hipStreamNonBlocking, 0)); CK(hipStreamCreateWithPriority(&sp1, hipStreamNonBlocking, -1));
    both(sp0, ncu, sp1, ncu * 8, reps, "priority streams (copy high):");
    both(sp1, ncu, sp0, ncu * 8, reps, "priority streams (mfma high):");
    std::vector<unsigned> full(words, 0xffffffffu);
    hipStream_t fa = make_stream(full), fb = make_stream(full);
    both(fa, ncu, fb, ncu * 8, reps, "full-mask ext streams:");
    both(fa, ncu / 2, fb, ncu * 8, reps, "full-mask ext streams, mfma on 128 wgs only:");
    both(fa, 192, fb, 64 * 8, reps, "full-mask ext streams, mfma 192 wgs, copy 512 wgs:");
    both(fa, 192, fb, 64 * 24, reps, "full-mask ext streams, mfma 192 wgs, copy 1536 wgs:");
    std::vector<unsigned> small, big; split(8, small, big);
    hipStream_t sc = make_stream(small);
    both(fa, 192, sc, 64 * 8, reps, "mfma 192 wgs unmasked | copy masked to 64 CUs:");
    both(fa, 256, sc, 64 * 8, reps, "mfma 256 wgs unmasked | copy masked to 64 CUs:");
  }
  for (int k : {8, 12, 16}) {
    std::vector<unsigned> small, big; split(k, small, big);
    hipStream_t sm = make_stream(big), sc = make_stream(small);
    const int cu_m = ncu - 8 * k, cu_c = 8 * k;
    const float mf = t_mfma(sm, cu_m);
    const float cp = t_copy(sc, cu_c * 8, 2);
    printf("k = %d CUs per XCD for the copy: alone on their partitions: mfma(%d CUs, same work per CU) %.3f ms; copy(%d CUs) %.3f ms = %.2f TB/s\n", k, cu_m, mf, cu_c, cp, 2.0 * pieces * 65536 / cp / 1e9);
    char name[96]; snprintf(name, sizeof name, "side by side, %d | %d CUs:", cu_m, cu_c);
    // same MFMA work per CU => total MFMA work scales with cu_m / ncu: report the equivalent full-work time too
    both(sm, cu_m, sc, cu_c * 8, reps, name);
    printf("   (mfma launch here carries %d/%d of the full launch's work: full-work equivalent = measured x %.3f)\n", cu_m, ncu, (double)ncu / cu_m);
    CK(hipStreamDestroy(sm)); CK(hipStreamDestroy(sc));
  }
  return 0;
}
