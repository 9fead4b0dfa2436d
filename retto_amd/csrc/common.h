// Shared host/device declarations for libretto_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <stdexcept>
#include <string>

namespace rt {

// One image of a ragged batch at some pyramid level: `off` = index of its first
// pixel in the level's concatenated NHWC activation buffer.
struct ImgGeom {
  long long off;
  int H, W;
  int pad_;
};

enum Act { ACT_NONE = 0, ACT_RELU = 1, ACT_HSWISH = 2, ACT_SWISH = 3, ACT_SIGMOID = 4 };

// Fused conv/GEMM epilogue: y = lab(act(acc + bias)) [+ residual]
struct Epilogue {
  const float* bias = nullptr;   // [Npad16] zero padded, or nullptr
  int act = ACT_NONE;     // Act
  int has_lab = 0;        // LCNetV3 LearnableAffineBlock after the activation
  float lab_a = 1.f, lab_c = 0.f;
  const float* residual = nullptr;  // optional, same row indexing as the output
  int ld_res = 0;
  // Squeeze-excite scale folded into the A operand (wide GEMM tiles only): row m of image i is
  // multiplied by a_scale[i * ld_scale + k] while its K-slab is staged.  a_tab holds, per row tile,
  // {image of the tile's first row, first row of the next image}; a tile spans at most 2 images.
  const float* a_scale = nullptr;
  int ld_scale = 0;
  const int* a_tab = nullptr;
  int a_tab_stride = 2;   // 3: {image, first row of the next image, of the one after} per 256-row block (k_gemm32p; images >= 128 rows)
  int n_img = 0;          // images behind a_scale (k_gemm32p clamps the neighbours it prefetches)
  // CTC head: instead of storing the logits tile, the wide GEMM leaves per (row, column tile) the
  // maximum, its column and sum(exp(logit - max)); nn::argmax_merge folds the tiles of a row.
  float* am_max = nullptr;
  int* am_idx = nullptr;
  float* am_sum = nullptr;
  int am_tiles = 0;
};

struct RtError : std::runtime_error {
  int code;
  RtError(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};

#define RT_HIP_CHECK(expr)                                                                       \
  do {                                                                                           \
    hipError_t e__ = (expr);                                                                     \
    if (e__ != hipSuccess)                                                                       \
      throw ::rt::RtError(4, std::string(#expr) + ": " + hipGetErrorString(e__) + " (" __FILE__ ":" + \
                                 std::to_string(__LINE__) + ")");                                \
  } while (0)

// hipLaunchKernelGGL + an immediate hipGetLastError(): a launch the runtime rejects (grid too large, too
// much LDS, no code object for the device) must surface as RT_ERR_BACKEND, not as RT_OK over garbage.
#define RT_LAUNCH(kernel, grid, block, shmem, stream, ...)                                                   \
  do {                                                                                                       \
    hipLaunchKernelGGL(kernel, grid, block, shmem, stream, __VA_ARGS__);                                     \
    hipError_t le__ = hipGetLastError();                                                                     \
    if (le__ != hipSuccess)                                                                                  \
      throw ::rt::RtError(4, std::string("kernel launch rejected: " #kernel ": ") + hipGetErrorString(le__) + \
                                 " (" __FILE__ ":" + std::to_string(__LINE__) + ")");                        \
  } while (0)
// hipFuncAttributeMaxDynamicSharedMemorySize is kept PER DEVICE and applies to the current device only: a process-wide
// "set once" flag leaves a session created later on another GPU launching > 64 KB of dynamic LDS without it (the launch is
// then rejected).  allow_big_lds(f, bytes) sets it once per (device, kernel); thread-safe (a session's lanes launch concurrently).
void allow_big_lds(const void* kernel, int bytes);
// CUs a kernel launched on `st` can occupy: the stream's CU partition (runtime.h) or the whole device
int stream_cus(hipStream_t st);
constexpr int RT_MAX_GRID_Y = 65535;  // HIP limit of gridDim.y / gridDim.z

static inline int round_up(int v, int m) { return (v + m - 1) / m * m; }
// Channel pitch of an NHWC activation: multiples of 4 (16-byte vectors); wide tensors are padded to
// 32 channels so that every pixel starts on a 128-byte line (240 -> 256: the depthwise kernels'
// 32-channel slabs and the GEMM A rows stop straddling lines).  Padding channels hold zeros.
static inline int chan_pitch(int c) { return c >= 128 ? round_up(c, 32) : round_up(c, 4); }
static inline long long round_up_ll(long long v, long long m) { return (v + m - 1) / m * m; }

}  // namespace rt
