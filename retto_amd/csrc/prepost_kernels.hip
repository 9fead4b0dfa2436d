// Pre/post-processing kernels (gfx950).  Integer / f32 / f64 arithmetic restating
// retto-core's processors and the third-party crates they call (image 0.25.6,
// imageproc 0.25.0, geo 0.30, Clipper 6.4.2 via geo-clipper).
// Compiled with -ffp-contract=off: results must be bit-identical to the CPU path the
// reference runs, so no expression may be fused or re-associated.
#include "prepost.h"

#include "geom_math.h"

namespace rt {
namespace pp {

typedef unsigned char u8;

// ===========================================================================
// thumbnail (image::imageops::thumbnail, RGB8)
// ===========================================================================
struct ThumbSrc { const u8* p; int h, w; };

__device__ __forceinline__ u8 numcast_u8(float v, int* err) {
  if (!(v > -1.0f && v < 256.0f)) { *err = 1; return 0; }
  return (u8)v;
}

// One output pixel (outx, outy) of thumbnail(src -> nw x nh); writes 3 channels to o[].
__device__ void thumb_pixel(const ThumbSrc& s, int nw, int nh, uint32_t outx, uint32_t outy, u8* o, int* err) {
  const uint32_t W = (uint32_t)s.w, H = (uint32_t)s.h;
  float x_ratio = (float)s.w / (float)nw;
  float y_ratio = (float)s.h / (float)nh;
  float bottomf = (float)outy * y_ratio;
  float topf = bottomf + y_ratio;
  uint32_t bottom = min(max(gm::f32_as_u32(ceilf(bottomf)), 0u), H - 1);
  uint32_t top = min(max(gm::f32_as_u32(ceilf(topf)), bottom), H);
  float leftf = (float)outx * x_ratio;
  float rightf = leftf + x_ratio;
  uint32_t left = min(max(gm::f32_as_u32(ceilf(leftf)), 0u), W - 1);
  uint32_t right = min(max(gm::f32_as_u32(ceilf(rightf)), left), W);
  auto px = [&](uint32_t x, uint32_t y, int c) -> uint32_t {
    if (x >= W || y >= H) { *err = 1; return 0u; }
    return (uint32_t)s.p[((size_t)y * W + x) * 3 + c];
  };
  if (bottom != top && left != right) {
    uint32_t n = (right - left) * (top - bottom);
    uint32_t rnd = n / 2;
    uint32_t s0 = 0, s1 = 0, s2 = 0;
    for (uint32_t y = bottom; y < top; y++) {
      const u8* row = s.p + ((size_t)y * W + left) * 3;
      for (uint32_t x = left; x < right; x++, row += 3) { s0 += row[0]; s1 += row[1]; s2 += row[2]; }
    }
    uint32_t v0 = (s0 + rnd) / n, v1 = (s1 + rnd) / n, v2 = (s2 + rnd) / n;
    o[0] = (u8)min(v0, 255u); o[1] = (u8)min(v1, 255u); o[2] = (u8)min(v2, 255u);
  } else if (bottom != top) {
    float fract = (gm::fract_f32(leftf) + gm::fract_f32(rightf)) / 2.0f;
    uint32_t l = right - 1;
    float fact_right = fract / (float)(top - bottom);
    float fact_left = (1.0f - fract) / (float)(top - bottom);
    for (int c = 0; c < 3; c++) {
      uint32_t sl = 0, sr = 0;
      for (uint32_t y = bottom; y < top; y++) { sl += px(l, y, c); sr += px(l + 1, y, c); }
      o[c] = numcast_u8(fact_left * (float)sl + fact_right * (float)sr, err);
    }
  } else if (left != right) {
    float fract = (gm::fract_f32(topf) + gm::fract_f32(bottomf)) / 2.0f;
    uint32_t b = top - 1;
    float fact_top = fract / (float)(right - left);
    float fact_bot = (1.0f - fract) / (float)(right - left);
    for (int c = 0; c < 3; c++) {
      uint32_t sb = 0, st = 0;
      for (uint32_t x = left; x < right; x++) { sb += px(x, b, c); st += px(x, b + 1, c); }
      o[c] = numcast_u8(fact_bot * (float)sb + fact_top * (float)st, err);
    }
  } else {
    float frac_v = (gm::fract_f32(topf) + gm::fract_f32(bottomf)) / 2.0f;
    float frac_h = (gm::fract_f32(leftf) + gm::fract_f32(rightf)) / 2.0f;
    uint32_t l = right - 1, b = top - 1;
    float fact_tr = frac_v * frac_h;
    float fact_tl = frac_v * (1.0f - frac_h);
    float fact_br = (1.0f - frac_v) * frac_h;
    float fact_bl = (1.0f - frac_v) * (1.0f - frac_h);
    for (int c = 0; c < 3; c++) {
      float k_bl = (float)px(l, b, c), k_tl = (float)px(l, b + 1, c);
      float k_br = (float)px(l + 1, b, c), k_tr = (float)px(l + 1, b + 1, c);
      o[c] = numcast_u8(fact_br * k_br + fact_tr * k_tr + fact_bl * k_bl + fact_tl * k_tl, err);
    }
  }
}

__global__ __launch_bounds__(256) void k_thumbnail(ThumbSrc s, u8* __restrict__ dst, int nh, int nw, int* err_flag) {
  long long p = (long long)blockIdx.x * 256 + threadIdx.x;
  if (p >= (long long)nh * nw) return;
  int err = 0;
  u8 o[3];
  thumb_pixel(s, nw, nh, (uint32_t)(p % nw), (uint32_t)(p / nw), o, &err);
  dst[p * 3] = o[0]; dst[p * 3 + 1] = o[1]; dst[p * 3 + 2] = o[2];
  if (err) atomicOr(err_flag, 1);
}

void thumbnail_rgb8(hipStream_t st, const uint8_t* src, int h, int w, uint8_t* dst, int nh, int nw, int* err_flag) {
  long long total = (long long)nh * nw;
  if (total <= 0) return;
  if (h == 0 || w == 0) { (void)hipMemsetAsync(dst, 0, (size_t)total * 3, st); return; }
  if (h == nh && w == nw) {  // ratio 1: every output pixel is the 1x1 block average = the pixel itself
    (void)hipMemcpyAsync(dst, src, (size_t)total * 3, hipMemcpyDeviceToDevice, st);
    return;
  }
  RT_LAUNCH(k_thumbnail, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, ThumbSrc{src, h, w}, dst, nh, nw,
                     err_flag);
}

// ===========================================================================
// det normalise
// ===========================================================================
struct Norm3 { float scale, mean[3], stdv[3]; };
__global__ __launch_bounds__(256) void k_det_normalize(const u8* __restrict__ rgb, long long npix, Norm3 nm, int layout,
                                                       float* __restrict__ out) {
  long long p = (long long)blockIdx.x * 256 + threadIdx.x;
  if (p >= npix) return;
  float v[3];
#pragma unroll
  for (int c = 0; c < 3; c++) {  // channel c of BGR
    float x = (float)rgb[p * 3 + (2 - c)];
    v[c] = (x * nm.scale - nm.mean[c]) / nm.stdv[c];
  }
  if (layout == 0) {
    float4 o = make_float4(v[0], v[1], v[2], 0.0f);
    reinterpret_cast<float4*>(out)[p] = o;
  } else {
    out[p] = v[0]; out[npix + p] = v[1]; out[2 * npix + p] = v[2];
  }
}
// all pages of a det launch group in one launch (layout 0: [pixel][4] f32, the stem's input)
__global__ __launch_bounds__(256) void k_det_normalize_batch(const NormDesc* __restrict__ descs, Norm3 nm, float* __restrict__ out) {
  const NormDesc d = descs[blockIdx.y];
  long long p = (long long)blockIdx.x * 256 + threadIdx.x;
  if (p >= d.npix) return;
  float v[3];
#pragma unroll
  for (int c = 0; c < 3; c++) {  // channel c of BGR
    float x = (float)d.rgb[p * 3 + (2 - c)];
    v[c] = (x * nm.scale - nm.mean[c]) / nm.stdv[c];
  }
  reinterpret_cast<float4*>(out)[d.out_pix + p] = make_float4(v[0], v[1], v[2], 0.0f);
}
void det_normalize_batch(hipStream_t st, const NormDesc* d_descs, int n, long long max_pix, float scale, const float* mean3,
                         const float* std3, float* out) {
  if (n <= 0 || max_pix <= 0) return;
  Norm3 nm; nm.scale = scale;
  for (int i = 0; i < 3; i++) { nm.mean[i] = mean3[i]; nm.stdv[i] = std3[i]; }
  RT_LAUNCH(k_det_normalize_batch, dim3((unsigned)((max_pix + 255) / 256), n), dim3(256), 0, st, d_descs, nm, out);
}
void det_normalize(hipStream_t st, const uint8_t* rgb, int h, int w, float scale, const float* mean3,
                   const float* std3, int layout, float* out) {
  long long npix = (long long)h * w;
  if (npix <= 0) return;
  Norm3 nm; nm.scale = scale;
  for (int i = 0; i < 3; i++) { nm.mean[i] = mean3[i]; nm.stdv[i] = std3[i]; }
  RT_LAUNCH(k_det_normalize, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, st, rgb, npix, nm, layout, out);
}


// ===========================================================================
// crops: imageproc warp_into(Bicubic, default white) + rotate270
// ===========================================================================
__device__ __forceinline__ u8 clamp_u8_trunc(float x) {
  if (x < 255.0f) { if (x > 0.0f) return (u8)x; return 0; }
  return 255;
}
__device__ __forceinline__ float cubic(float p0, float p1, float p2, float p3, float x) {
  return p1 + 0.5f * x * (p2 - p0 + x * (2.0f * p0 - 5.0f * p1 + 4.0f * p2 - p3 + x * (3.0f * (p1 - p2) + p3 - p0)));
}
__global__ __launch_bounds__(256) void k_warp_crops(const CropDesc* __restrict__ descs, u8* __restrict__ pool) {
  const CropDesc d = descs[blockIdx.y];
  int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= d.w * d.h) return;
  int y = p / d.w, x = p % d.w;
  float fx = (float)x, fy = (float)y;
  float dd = d.inv[6] * fx + d.inv[7] * fy + d.inv[8];
  float px = (d.inv[0] * fx + d.inv[1] * fy + d.inv[2]) / dd;
  float py = (d.inv[3] * fx + d.inv[4] * fy + d.inv[5]) / dd;
  u8 o[3] = {255, 255, 255};
  float left = floorf(px) - 1.0f, right = left + 4.0f;
  float top = floorf(py) - 1.0f, bottom = top + 4.0f;
  float xw = px - (left + 1.0f), yw = py - (top + 1.0f);
  if ((left >= 0.0f) && (right < (float)d.sw) && (top >= 0.0f) && (bottom < (float)d.sh)) {
    uint32_t l = gm::f32_as_u32(left), tp = gm::f32_as_u32(top);
#pragma unroll
    for (int c = 0; c < 3; c++) {
      u8 col[4];
#pragma unroll
      for (uint32_t r = 0; r < 4; r++) {
        const u8* row = d.src + ((size_t)(tp + r) * d.sw + l) * 3 + c;
        col[r] = clamp_u8_trunc(cubic((float)row[0], (float)row[3], (float)row[6], (float)row[9], xw));
      }
      o[c] = clamp_u8_trunc(cubic((float)col[0], (float)col[1], (float)col[2], (float)col[3], yw));
    }
  }
  size_t dst;
  if (!d.rot) dst = (size_t)y * d.w + x;
  else dst = (size_t)(d.w - 1 - x) * d.h + y;  // rotate270: out(y, w-1-x) = in(x, y), out width = h
  u8* q = pool + d.out_off + dst * 3;
  q[0] = o[0]; q[1] = o[1]; q[2] = o[2];
}
void warp_crops(hipStream_t st, const CropDesc* descs, int n, int max_pix, uint8_t* pool) {
  if (n <= 0 || max_pix <= 0) return;
  for (int y0 = 0; y0 < n; y0 += RT_MAX_GRID_Y)  // one grid row per crop: chunked to the gridDim.y limit
    RT_LAUNCH(k_warp_crops, dim3((max_pix + 255) / 256, std::min(n - y0, RT_MAX_GRID_Y)), dim3(256), 0, st, descs + y0, pool);
}

// cls_processor.rs:108-121 (first-max argmax) + :163-166 rotate_180_in_place
__global__ __launch_bounds__(256) void k_cls_post_rotate(const float* __restrict__ probs, const int* __restrict__ crop_of_row,
                                                         float thresh, const CropRef* __restrict__ crops,
                                                         u8* __restrict__ pool, int* __restrict__ label_idx,
                                                         float* __restrict__ score) {
  int row = blockIdx.y;
  int ci = crop_of_row[row];
  float p0 = probs[row * 2], p1 = probs[row * 2 + 1];
  int idx = p1 > p0 ? 1 : 0;
  float sc = idx ? p1 : p0;
  if (blockIdx.x == 0 && threadIdx.x == 0) { label_idx[ci] = idx; score[ci] = sc; }
  if (!(idx == 1 && sc >= thresh)) return;
  const CropRef c = crops[ci];
  long long n = (long long)c.h * c.w;
  long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n / 2) return;
  u8* a = pool + c.off + i * 3;
  u8* b = pool + c.off + (n - 1 - i) * 3;
  u8 t0 = a[0], t1 = a[1], t2 = a[2];
  a[0] = b[0]; a[1] = b[1]; a[2] = b[2];
  b[0] = t0; b[1] = t1; b[2] = t2;
}
void cls_post_rotate(hipStream_t st, const float* probs, const int* crop_of_row, int rows, float thresh,
                     const CropRef* crops, uint8_t* pool, int max_pix, int* label_idx, float* score) {
  if (rows <= 0) return;
  int bx = std::max(1, (max_pix / 2 + 255) / 256);
  for (int y0 = 0; y0 < rows; y0 += RT_MAX_GRID_Y)
    RT_LAUNCH(k_cls_post_rotate, dim3(bx, std::min(rows - y0, RT_MAX_GRID_Y)), dim3(256), 0, st, probs + 2 * (size_t)y0,
              crop_of_row + y0, thresh, crops, pool, label_idx, score);
}

// image_helper.rs:176-209
__global__ __launch_bounds__(256) void k_resize_norm(const LineDesc* __restrict__ lines, int img_h, const u8* __restrict__ pool,
                                                     int layout, float* __restrict__ out, int* err_flag) {
  const LineDesc L = lines[blockIdx.y];
  long long p = (long long)blockIdx.x * 256 + threadIdx.x;
  if (p >= (long long)img_h * L.W) return;
  int y = (int)(p / L.W), x = (int)(p % L.W);
  float v[3] = {0.0f, 0.0f, 0.0f};
  if (x < L.resized_w) {
    int err = 0;
    u8 o[3];
    thumb_pixel(ThumbSrc{pool + L.crop_off, L.h, L.w}, L.resized_w, img_h, (uint32_t)x, (uint32_t)y, o, &err);
    if (err) atomicOr(err_flag, 1);
#pragma unroll
    for (int c = 0; c < 3; c++) { float t = (float)o[c] / 255.0f; v[c] = (t - 0.5f) / 0.5f; }
  }
  if (layout == 0) {
    reinterpret_cast<float4*>(out + L.out_off)[p] = make_float4(v[0], v[1], v[2], 0.0f);
  } else {
    long long plane = (long long)img_h * L.W;
    float* o = out + L.out_off;
    o[p] = v[0]; o[plane + p] = v[1]; o[2 * plane + p] = v[2];
  }
}
void resize_norm(hipStream_t st, const LineDesc* lines, int n, int img_h, int max_W, const uint8_t* pool, int layout,
                 float* out, int* err_flag) {
  if (n <= 0 || max_W <= 0) return;
  long long total = (long long)img_h * max_W;
  for (int y0 = 0; y0 < n; y0 += RT_MAX_GRID_Y)
    RT_LAUNCH(k_resize_norm, dim3((unsigned)((total + 255) / 256), std::min(n - y0, RT_MAX_GRID_Y)), dim3(256), 0, st,
              lines + y0, img_h, pool, layout, out, err_flag);
}

// rec_processor.rs:48-97
__global__ __launch_bounds__(64) void k_ctc_decode(const int* __restrict__ idx, const float* __restrict__ prob,
                                                   const ImgGeom* __restrict__ lines, int n, int* __restrict__ tokens,
                                                   int* __restrict__ n_tokens, float* __restrict__ score) {
  int i = blockIdx.x * 64 + threadIdx.x;
  if (i >= n) return;
  const ImgGeom g = lines[i];
  int T = g.H * g.W;
  int cnt = 0; float acc = 0.0f;
  for (int t = 0; t < T; t++) {
    int id = idx[g.off + t];
    bool sel = id != 0;
    if (t >= 1) sel = sel && id != idx[g.off + t - 1];
    if (sel) { tokens[g.off + cnt] = id; acc = acc + prob[g.off + t]; cnt++; }
  }
  n_tokens[i] = cnt;
  score[i] = acc / (float)(unsigned)cnt;
}
void ctc_decode(hipStream_t st, const int* idx, const float* prob, const ImgGeom* lines, int n, int* tokens,
                int* n_tokens, float* score) {
  if (n <= 0) return;
  RT_LAUNCH(k_ctc_decode, dim3((n + 63) / 64), dim3(64), 0, st, idx, prob, lines, n, tokens, n_tokens, score);
}

// ===========================================================================
// (round 5: 8192 values per block, 16-byte loads -- was 65536 per block, scalar: one 960 x 960 map ran on 15 workgroups for 77 us,
//  6 % of the C2 call)
constexpr int SUM_PER_BLOCK = 8192;
int sum_blocks(long long n) { return (int)((n + SUM_PER_BLOCK - 1) / SUM_PER_BLOCK); }
__global__ __launch_bounds__(256) void k_sum_partial(const float* __restrict__ x, long long n, double* __restrict__ partials) {
  __shared__ double red[256];
  const long long base = (long long)blockIdx.x * SUM_PER_BLOCK;
  double s = 0.0;
  if ((((size_t)x & 15) == 0) && base + SUM_PER_BLOCK <= n) {
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    const f32x4* x4 = reinterpret_cast<const f32x4*>(x + base);
#pragma unroll
    for (int i = 0; i < SUM_PER_BLOCK / 1024; i++) { const f32x4 v = x4[threadIdx.x + 256 * i]; s += (double)v[0] + (double)v[1] + (double)v[2] + (double)v[3]; }
  } else {
    for (int i = threadIdx.x; i < SUM_PER_BLOCK; i += 256) { long long k = base + i; if (k < n) s += (double)x[k]; }
  }
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) { if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
  if (threadIdx.x == 0) partials[blockIdx.x] = red[0];
}
void sum_partial(hipStream_t st, const float* x, long long n, double* partials) {
  if (n <= 0) return;
  RT_LAUNCH(k_sum_partial, dim3(sum_blocks(n)), dim3(256), 0, st, x, n, partials);
}

}  // namespace pp
}  // namespace rt
