// Scalar geometry / size arithmetic of retto-core's pre- and post-processing, written
// once for host (scheduler) and device (kernels).  Every expression keeps the reference's
// type and operation order; translation units including this header are compiled with
// -ffp-contract=off so nothing is fused.  Citations are to /root/reference/retto-core/src.
#pragma once
#include <math.h>
#include <stdint.h>

#ifdef __HIPCC__
#include <hip/hip_runtime.h>
#define RT_HD __host__ __device__ __forceinline__
#else
#define RT_HD inline
#endif

namespace rt {
namespace gm {

// Rust `as` casts: saturating, NaN -> 0
RT_HD uint32_t f32_as_u32(float x) {
  if (!(x > 0.0f)) return 0u;
  if (x >= 4294967296.0f) return 4294967295u;
  return (uint32_t)x;
}
RT_HD int32_t f32_as_i32(float x) {
  if (x != x) return 0;
  if (x >= 2147483648.0f) return 2147483647;
  if (x <= -2147483648.0f) return (-2147483647 - 1);
  return (int32_t)x;
}
RT_HD float fract_f32(float x) { return x - truncf(x); }

// image_helper.rs:106-148 resize_both: up to two thumbnail passes; out[2i]=h, out[2i+1]=w
RT_HD int resize_both_plan(int ori_h, int ori_w, int max_side, int min_side, int* out) {
  int n = 0;
  float h = (float)ori_h, w = (float)ori_w;
  int mx = ori_h > ori_w ? ori_h : ori_w, mn = ori_h < ori_w ? ori_h : ori_w;
  if (mx > max_side) {
    float scale = (float)max_side / fmaxf(h, w);
    uint32_t rh = f32_as_u32(floorf(h * scale)) / 32u; if (rh < 1u) rh = 1u; rh *= 32u;
    uint32_t rw = f32_as_u32(floorf(w * scale)) / 32u; if (rw < 1u) rw = 1u; rw *= 32u;
    out[2 * n] = (int)rh; out[2 * n + 1] = (int)rw; n++;
  }
  if (mn < min_side) {
    float scale = (float)min_side / fminf(h, w);
    uint32_t rh = f32_as_u32(roundf(floorf(h * scale) / 32.0f)) * 32u;
    uint32_t rw = f32_as_u32(roundf(floorf(w * scale) / 32.0f)) * 32u;
    out[2 * n] = (int)rh; out[2 * n + 1] = (int)rw; n++;
  }
  return n;
}
// image_helper.rs:150-174 resize_either
RT_HD void resize_either_dims(int h, int w, int limit_type, int limit_len, int* rh, int* rw) {
  float ratio = 1.0f;
  int mx = w > h ? w : h, mn = w < h ? w : h;
  if (limit_type == 1) { if (mx > limit_len) ratio = (float)limit_len / (float)mx; }
  else { if (mn < limit_len) ratio = (float)limit_len / (float)mn; }
  *rh = (int)(f32_as_u32(roundf(floorf((float)h * ratio) / 32.0f)) * 32u);
  *rw = (int)(f32_as_u32(roundf(floorf((float)w * ratio) / 32.0f)) * 32u);
}

// points.rs:125-169: |a-b| with the difference taken in f32, the norm in f64
RT_HD float side_len(const float* a, const float* b) {
  double dx = (double)(a[0] - b[0]);
  double dy = (double)(a[1] - b[1]);
  return (float)sqrt(dx * dx + dy * dy);
}
// points.rs:179-194
RT_HD void scale_and_clip(float* box8, double bitmap_w, double bitmap_h, double ori_w, double ori_h) {
  double inv_w = ori_w / bitmap_w, inv_h = ori_h / bitmap_h;
  for (int i = 0; i < 4; i++) {
    double x1 = round((double)box8[2 * i] * inv_w);
    x1 = x1 < 0.0 ? 0.0 : (x1 > ori_w - 1.0 ? ori_w - 1.0 : x1);
    double y1 = round((double)box8[2 * i + 1] * inv_h);
    y1 = y1 < 0.0 ? 0.0 : (y1 > ori_h - 1.0 ? ori_h - 1.0 : y1);
    box8[2 * i] = (float)x1; box8[2 * i + 1] = (float)y1;
  }
}

// image_helper.rs:224-226,245: crop size (truncating casts) and the rotate270 rule.
// w/h are the PRE-rotation dims; cw/ch the untruncated f32 sizes used for the homography.
struct CropDims { int w, h, rot; float cw, ch; };
RT_HD CropDims crop_dims(const float* box8) {
  const float *tl = box8, *tr = box8 + 2, *br = box8 + 4, *bl = box8 + 6;
  float w_brc = side_len(bl, br), w_tlc = side_len(tl, tr);
  float h_brc = side_len(tr, br), h_tlc = side_len(tl, bl);
  CropDims d;
  d.cw = fmaxf(w_brc, w_tlc); d.ch = fmaxf(h_brc, h_tlc);
  d.w = (int)f32_as_u32(d.cw); d.h = (int)f32_as_u32(d.ch);
  d.rot = ((float)d.h / (float)d.w >= 1.5f) ? 1 : 0;  // x/0 = inf, 0/0 = NaN (false)
  return d;
}

// imageproc Projection::from_control_points(box -> rectangle): 8x8 DLT system solved in
// f64 (Gaussian elimination with partial pivoting), stored as f32, inverse by
// adjugate/determinant in f32 then normalised.  inv[] maps OUTPUT pixel -> SOURCE pixel.
RT_HD bool projection_inverse(const float* from8, float cw, float ch, float* inv) {
  const float to8[8] = {0.0f, 0.0f, cw, 0.0f, cw, ch, 0.0f, ch};
  double A[8][9];
  for (int i = 0; i < 4; i++) {
    double xf = (double)from8[2 * i], yf = (double)from8[2 * i + 1];
    double x = (double)to8[2 * i], y = (double)to8[2 * i + 1];
    A[2 * i][0] = 0.0; A[2 * i][1] = 0.0; A[2 * i][2] = 0.0; A[2 * i][3] = -xf; A[2 * i][4] = -yf; A[2 * i][5] = -1.0;
    A[2 * i][6] = y * xf; A[2 * i][7] = y * yf; A[2 * i][8] = -y;
    A[2 * i + 1][0] = xf; A[2 * i + 1][1] = yf; A[2 * i + 1][2] = 1.0; A[2 * i + 1][3] = 0.0; A[2 * i + 1][4] = 0.0;
    A[2 * i + 1][5] = 0.0; A[2 * i + 1][6] = -x * xf; A[2 * i + 1][7] = -x * yf; A[2 * i + 1][8] = x;
  }
  for (int c = 0; c < 8; c++) {
    int p = c; double best = fabs(A[c][c]);
    for (int r = c + 1; r < 8; r++) if (fabs(A[r][c]) > best) { best = fabs(A[r][c]); p = r; }
    if (best == 0.0) return false;
    if (p != c) for (int k = 0; k < 9; k++) { double t = A[c][k]; A[c][k] = A[p][k]; A[p][k] = t; }
    for (int r = c + 1; r < 8; r++) {
      double f = A[r][c] / A[c][c];
      for (int k = c; k < 9; k++) A[r][k] = A[r][k] - f * A[c][k];
    }
  }
  for (int r = 7; r >= 0; r--) {
    double s = A[r][8];
    for (int k = r + 1; k < 8; k++) s = s - A[r][k] * A[k][8];
    A[r][8] = s / A[r][r];
  }
  float t[9];
  for (int i = 0; i < 8; i++) t[i] = (float)A[i][8];
  t[8] = 1.0f;
  for (int i = 0; i < 9; i++) t[i] = t[i] / t[8];
  float t00 = t[0], t01 = t[1], t02 = t[2], t10 = t[3], t11 = t[4], t12 = t[5], t20 = t[6], t21 = t[7], t22 = t[8];
  float m00 = t11 * t22 - t12 * t21, m01 = t10 * t22 - t12 * t20, m02 = t10 * t21 - t11 * t20;
  float det = t00 * m00 - t01 * m01 + t02 * m02;
  if (fabsf(det) < 1e-10f) return false;
  float m10 = t01 * t22 - t02 * t21, m11 = t00 * t22 - t02 * t20, m12 = t00 * t21 - t01 * t20;
  float m20 = t01 * t12 - t02 * t11, m21 = t00 * t12 - t02 * t10, m22 = t00 * t11 - t01 * t10;
  inv[0] = m00 / det; inv[1] = -m10 / det; inv[2] = m20 / det;
  inv[3] = -m01 / det; inv[4] = m11 / det; inv[5] = -m21 / det;
  inv[6] = m02 / det; inv[7] = -m12 / det; inv[8] = m22 / det;
  float n8 = inv[8];
  for (int i = 0; i < 9; i++) inv[i] = inv[i] / n8;
  return true;
}

// image_helper.rs:176-183 resize_norm_image widths
RT_HD int resize_norm_width(int img_h, int img_w, float max_wh_ratio) {
  if (max_wh_ratio > 0.0f) {
    float v = (float)img_h * max_wh_ratio;
    if (!(v > 0.0f)) return 0;
    if (v >= 2147483647.0f) return 2147483647;
    return (int)v;
  }
  return img_w;
}
RT_HD int resize_norm_resized_w(int img_h, int W, int ori_h, int ori_w) {
  double rw = ceil((double)img_h * (double)(uint32_t)ori_w / (double)(uint32_t)ori_h);
  long long r;
  if (rw != rw) r = 0;
  else if (rw <= 0.0) r = 0;
  else if (rw > 2147483647.0) r = 2147483647LL;
  else r = (long long)rw;
  return (int)(r < (long long)W ? r : (long long)W);
}

}  // namespace gm
}  // namespace rt
