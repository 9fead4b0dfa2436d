// ============================================================================
// oracle/retto_oracle.cpp -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
//
// CPU restatement (plain C++17, scalar, single-threaded) of the pre/post
// processing arithmetic on retto-core's OCR hot path.  Only tests/,
// __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
// library.  Nothing under retto_amd/ links, imports or calls it.
//
// PARITY UNPINNED: the reference (NekoImageLand/retto @ 0.1.5) is Rust and
// cannot be compiled here (no cargo/rustc), its NN arithmetic lives in ONNX
// Runtime + model files that are absent, and its own tests hold no golden
// vectors (retto-core/src/session.rs:206-255 assert only a corner within
// 10/100 px, cls label == 180 and an exact string, all needing network).  The
// third-party crates that hold most of the integer/f32 arithmetic are NOT in
// /root/reference; they are restated below from their published algorithms at
// the versions pinned in /root/reference/Cargo.lock:
//   image 0.25.6       imageops::thumbnail, rotate270, rotate180_in_place
//   imageproc 0.25.0   contours::find_contours, geometry::min_area_rect
//                      (+convex_hull), morphology::grayscale_dilate,
//                      drawing::draw_polygon_mut (+BresenhamLineIter),
//                      geometric_transformations::{Projection, warp_into}
//   geo 0.30.0         unsigned_area, Euclidean.length
//   geo-clipper 0.9.0 / clipper-sys 0.8.0 (fork @4f102a8) = Clipper 6.4.2
//                      ClipperOffset (jtRound, etClosedPolygon)
//   ndarray-stats 0.6  argmax / max
// Where a restatement had to pick a semantics (SVD solve -> Gaussian
// elimination, Clipper's clean-up union -> identity on the hull) the choice is
// written at the function.  This file is the oracle of record for those.
//
// Build: see oracle/Makefile (g++ -O2 -ffp-contract=off: no FMA contraction,
// so every f32/f64 expression rounds exactly as written).
// ============================================================================
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <limits>
#include <vector>

#define ORC_API extern "C" __attribute__((visibility("default")))

typedef uint8_t u8;

// ---------------------------------------------------------------------------
// Rust semantics helpers
// ---------------------------------------------------------------------------
// f32::round / f64::round: half away from zero (libm roundf/round).
static inline float rs_roundf(float x) { return roundf(x); }
static inline double rs_round(double x) { return round(x); }
// `x as u32` from f32: saturating, NaN -> 0.
static inline uint32_t rs_f32_as_u32(float x) {
  if (!(x > 0.0f)) return 0u;
  if (x >= 4294967296.0f) return 4294967295u;
  return (uint32_t)x;
}
static inline int32_t rs_f32_as_i32(float x) {
  if (x != x) return 0;
  if (x >= 2147483648.0f) return INT32_MAX;
  if (x <= -2147483648.0f) return INT32_MIN;
  return (int32_t)x;
}
static inline float rs_fract(float x) { return x - truncf(x); }

// ---------------------------------------------------------------------------
// a2 / a3 size arithmetic
// ---------------------------------------------------------------------------
// retto-core/src/image_helper.rs:106-148 (resize_both).  Returns the number of
// thumbnail passes (0..2) and their target dims in out[2*i] = h, out[2*i+1] = w.
ORC_API int orc_resize_both_plan(int ori_h, int ori_w, int max_side, int min_side, int* out) {
  int n = 0;
  float h = (float)ori_h, w = (float)ori_w;
  if (std::max(ori_h, ori_w) > max_side) {
    float scale = (float)max_side / std::max(h, w);
    uint32_t rh = std::max(rs_f32_as_u32(floorf(h * scale)) / 32u, 1u) * 32u;
    uint32_t rw = std::max(rs_f32_as_u32(floorf(w * scale)) / 32u, 1u) * 32u;
    out[2 * n] = (int)rh; out[2 * n + 1] = (int)rw; n++;
  }
  if (std::min(ori_h, ori_w) < min_side) {
    // note: computed from the ORIGINAL h, w (image_helper.rs:131-136)
    float scale = (float)min_side / std::min(h, w);
    uint32_t rh = rs_f32_as_u32(rs_roundf(floorf(h * scale) / 32.0f)) * 32u;
    uint32_t rw = rs_f32_as_u32(rs_roundf(floorf(w * scale) / 32.0f)) * 32u;
    out[2 * n] = (int)rh; out[2 * n + 1] = (int)rw; n++;
  }
  return n;
}

// retto-core/src/image_helper.rs:150-174 (resize_either). limit_type 0=Min 1=Max.
ORC_API void orc_resize_either_dims(int h, int w, int limit_type, int limit_len, int* rh, int* rw) {
  float ratio = 1.0f;
  if (limit_type == 1) {
    if (std::max(w, h) > limit_len) ratio = (float)limit_len / (float)std::max(w, h);
  } else {
    if (std::min(w, h) < limit_len) ratio = (float)limit_len / (float)std::min(w, h);
  }
  *rh = (int)(rs_f32_as_u32(rs_roundf(floorf((float)h * ratio) / 32.0f)) * 32u);
  *rw = (int)(rs_f32_as_u32(rs_roundf(floorf((float)w * ratio) / 32.0f)) * 32u);
}

// ---------------------------------------------------------------------------
// image 0.25.6 imageops::thumbnail on RGB8.
// Returns 0, or -1 where the crate would index out of bounds (panic).
// ---------------------------------------------------------------------------
static inline uint32_t clampu(uint32_t v, uint32_t lo, uint32_t hi) { return v < lo ? lo : (v > hi ? hi : v); }
static inline u8 numcast_u8(float v, int* err) {
  // <u8 as NumCast>::from(f32): None unless -1 < v < 256; else truncation.
  if (!(v > -1.0f && v < 256.0f)) { *err = 1; return 0; }
  return (u8)v;
}

ORC_API int orc_thumbnail(const u8* src, int height, int width, u8* dst, int new_height, int new_width) {
  if (new_height <= 0 || new_width <= 0) return 0;
  if (height == 0 || width == 0) { memset(dst, 0, (size_t)new_height * new_width * 3); return 0; }
  const uint32_t W = (uint32_t)width, H = (uint32_t)height;
  float x_ratio = (float)width / (float)new_width;
  float y_ratio = (float)height / (float)new_height;
  int err = 0;
  auto px = [&](uint32_t x, uint32_t y, int c) -> uint32_t {
    if (x >= W || y >= H) { err = 1; return 0; }
    return src[((size_t)y * W + x) * 3 + c];
  };
  for (uint32_t outy = 0; outy < (uint32_t)new_height; outy++) {
    float bottomf = (float)outy * y_ratio;
    float topf = bottomf + y_ratio;
    uint32_t bottom = clampu(rs_f32_as_u32(ceilf(bottomf)), 0, H - 1);
    uint32_t top = clampu(rs_f32_as_u32(ceilf(topf)), bottom, H);
    for (uint32_t outx = 0; outx < (uint32_t)new_width; outx++) {
      float leftf = (float)outx * x_ratio;
      float rightf = leftf + x_ratio;
      uint32_t left = clampu(rs_f32_as_u32(ceilf(leftf)), 0, W - 1);
      uint32_t right = clampu(rs_f32_as_u32(ceilf(rightf)), left, W);
      u8* o = dst + ((size_t)outy * new_width + outx) * 3;
      if (bottom != top && left != right) {
        // thumbnail_sample_block: u32 sums, (sum + n/2) / n
        uint32_t n = (right - left) * (top - bottom);
        uint32_t rnd = n / 2;
        for (int c = 0; c < 3; c++) {
          uint32_t s = 0;
          for (uint32_t y = bottom; y < top; y++)
            for (uint32_t x = left; x < right; x++) s += px(x, y, c);
          uint32_t v = (s + rnd) / n;
          o[c] = (u8)(v > 255 ? 255 : v);
        }
      } else if (bottom != top) {
        // left == right: thumbnail_sample_fraction_horizontal(image, right-1, frac, bottom, top)
        float fract = (rs_fract(leftf) + rs_fract(rightf)) / 2.0f;
        uint32_t l = right - 1;
        float fact_right = fract / (float)(top - bottom);
        float fact_left = (1.0f - fract) / (float)(top - bottom);
        for (int c = 0; c < 3; c++) {
          uint32_t sl = 0, sr = 0;
          for (uint32_t y = bottom; y < top; y++) { sl += px(l, y, c); sr += px(l + 1, y, c); }
          o[c] = numcast_u8(fact_left * (float)sl + fact_right * (float)sr, &err);
        }
      } else if (left != right) {
        // bottom == top: thumbnail_sample_fraction_vertical(image, left, right, top-1, frac)
        float fract = (rs_fract(topf) + rs_fract(bottomf)) / 2.0f;
        uint32_t b = top - 1;
        float fact_top = fract / (float)(right - left);
        float fact_bot = (1.0f - fract) / (float)(right - left);
        for (int c = 0; c < 3; c++) {
          uint32_t sb = 0, st = 0;
          for (uint32_t x = left; x < right; x++) { sb += px(x, b, c); st += px(x, b + 1, c); }
          o[c] = numcast_u8(fact_bot * (float)sb + fact_top * (float)st, &err);
        }
      } else {
        // both empty: thumbnail_sample_fraction_both(image, right-1, frac_v, top-1, frac_h)
        float frac_v = (rs_fract(topf) + rs_fract(bottomf)) / 2.0f;
        float frac_h = (rs_fract(leftf) + rs_fract(rightf)) / 2.0f;
        uint32_t l = right - 1, b = top - 1;
        float fact_tr = frac_v * frac_h;
        float fact_tl = frac_v * (1.0f - frac_h);
        float fact_br = (1.0f - frac_v) * frac_h;
        float fact_bl = (1.0f - frac_v) * (1.0f - frac_h);
        for (int c = 0; c < 3; c++) {
          float k_bl = (float)px(l, b, c), k_tl = (float)px(l, b + 1, c);
          float k_br = (float)px(l + 1, b, c), k_tr = (float)px(l + 1, b + 1, c);
          o[c] = numcast_u8(fact_br * k_br + fact_tr * k_tr + fact_bl * k_bl + fact_tl * k_tl, &err);
        }
      }
    }
  }
  return err ? -1 : 0;
}

// ---------------------------------------------------------------------------
// a3: rgb2bgr + normalize + permute (det_processor.rs:151-160,256-274;
// image_helper.rs:211-221).  rgb is the already-resized page.
// out is [1,3,H,W] with channel order B,G,R.
// ---------------------------------------------------------------------------
ORC_API void orc_det_normalize(const u8* rgb, int h, int w, float scale, const float* mean, const float* stdv,
                               float* out) {
  size_t plane = (size_t)h * w;
  for (size_t i = 0; i < plane; i++) {
    for (int c = 0; c < 3; c++) {
      u8 v = rgb[i * 3 + (2 - c)];  // channel c of BGR
      float f = ((float)v * scale - mean[c]) / stdv[c];
      out[(size_t)c * plane + i] = f;
    }
  }
}

// ---------------------------------------------------------------------------
// points.rs
// ---------------------------------------------------------------------------
struct Pt { float x, y; };
struct IPt { int x, y; };

// points.rs:179-194 scale_and_clip for PointBox<OrderedFloat<f32>>.
ORC_API void orc_scale_and_clip(float* box8, double bitmap_w, double bitmap_h, double ori_w, double ori_h) {
  double inv_w = ori_w / bitmap_w, inv_h = ori_h / bitmap_h;
  for (int i = 0; i < 4; i++) {
    double x0 = (double)box8[2 * i], y0 = (double)box8[2 * i + 1];
    double x1 = rs_round(x0 * inv_w); x1 = x1 < 0.0 ? 0.0 : (x1 > ori_w - 1.0 ? ori_w - 1.0 : x1);
    double y1 = rs_round(y0 * inv_h); y1 = y1 < 0.0 ? 0.0 : (y1 > ori_h - 1.0 ? ori_h - 1.0 : y1);
    box8[2 * i] = (float)x1; box8[2 * i + 1] = (float)y1;
  }
}
// points.rs:125-169: side lengths, f64 sqrt then cast to f32.  The subtraction
// happens in T (= f32 here) before the cast to f64.
static float side_len(const float* a, const float* b) {
  double dx = (double)(a[0] - b[0]);
  double dy = (double)(a[1] - b[1]);
  return (float)sqrt(dx * dx + dy * dy);
}

// ---------------------------------------------------------------------------
// a5 building blocks
// ---------------------------------------------------------------------------
// det_processor.rs:286-292 + imageproc grayscale_dilate with Mask offsets
// (kx-cx, ky-cy), kernel 2x2 anchor (1,1): {(-1,-1),(0,-1),(-1,0),(0,0)}.
ORC_API void orc_threshold_dilate(const float* pred, int h, int w, float thresh, int dilate, u8* mask) {
  std::vector<u8> m((size_t)h * w);
  for (size_t i = 0; i < (size_t)h * w; i++) m[i] = pred[i] > thresh ? 255 : 0;
  if (!dilate) { memcpy(mask, m.data(), m.size()); return; }
  for (int y = 0; y < h; y++)
    for (int x = 0; x < w; x++) {
      u8 v = 0;
      for (int dy = -1; dy <= 0; dy++)
        for (int dx = -1; dx <= 0; dx++) {
          int yy = y + dy, xx = x + dx;
          if (yy >= 0 && xx >= 0) v = std::max(v, m[(size_t)yy * w + xx]);
        }
      mask[(size_t)y * w + x] = v;
    }
}

// imageproc 0.25.0 contours::find_contours::<i32> (Suzuki-Abe).
struct Contour { std::vector<IPt> pts; int border_type; /*0 outer 1 hole*/ };

static void find_contours(const u8* mask, int height, int width, std::vector<Contour>& out) {
  std::vector<int32_t> iv((size_t)height * width, 0);
  for (size_t i = 0; i < iv.size(); i++) iv[i] = mask[i] > 0 ? 1 : 0;
  auto at = [&](int x, int y) -> int32_t& { return iv[(size_t)y * width + x]; };
  // W, NW, N, NE, E, SE, S, SW
  std::deque<IPt> diffs = {{-1, 0}, {-1, -1}, {0, -1}, {1, -1}, {1, 0}, {1, 1}, {0, 1}, {-1, 1}};
  auto rotate_to = [&](IPt v) {
    size_t pos = 0;
    for (; pos < diffs.size(); pos++) if (diffs[pos].x == v.x && diffs[pos].y == v.y) break;
    std::rotate(diffs.begin(), diffs.begin() + pos, diffs.end());
  };
  auto nonzero = [&](int x, int y) -> bool {
    return x >= 0 && y >= 0 && x < width && y < height && iv[(size_t)y * width + x] != 0;
  };
  int32_t curr_border_num = 1;
  for (int y = 0; y < height; y++) {
    for (int x = 0; x < width; x++) {
      if (at(x, y) == 0) continue;
      int have = 0; IPt adj{0, 0}; int btype = 0;
      if (at(x, y) == 1 && (x == 0 || at(x - 1, y) == 0)) {
        have = 1; adj = {x - 1, y}; btype = 0;
      } else if (at(x, y) > 0 && (x + 1 == width || at(x + 1, y) == 0)) {
        have = 1; adj = {x + 1, y}; btype = 1;
      }
      if (!have) continue;
      curr_border_num += 1;
      Contour c; c.border_type = btype;
      IPt curr{x, y};
      rotate_to({adj.x - curr.x, adj.y - curr.y});
      int found = 0; IPt pos1{0, 0};
      for (size_t k = 0; k < diffs.size(); k++) {
        int nx = curr.x + diffs[k].x, ny = curr.y + diffs[k].y;
        if (nonzero(nx, ny)) { pos1 = {nx, ny}; found = 1; break; }
      }
      if (found) {
        IPt pos2 = pos1, pos3 = curr;
        for (;;) {
          c.pts.push_back(pos3);
          rotate_to({pos2.x - pos3.x, pos2.y - pos3.y});
          IPt pos4{0, 0};
          for (int k = (int)diffs.size() - 1; k >= 0; k--) {
            int nx = pos3.x + diffs[k].x, ny = pos3.y + diffs[k].y;
            if (nonzero(nx, ny)) { pos4 = {nx, ny}; break; }
          }
          bool is_right_edge = false;
          for (int k = (int)diffs.size() - 1; k >= 0; k--) {
            if (diffs[k].x == pos4.x - pos3.x && diffs[k].y == pos4.y - pos3.y) break;
            if (diffs[k].x == 1 && diffs[k].y == 0) { is_right_edge = true; break; }
          }
          if (pos3.x + 1 == width || is_right_edge) at(pos3.x, pos3.y) = -curr_border_num;
          else if (at(pos3.x, pos3.y) == 1) at(pos3.x, pos3.y) = curr_border_num;
          if (pos4.x == curr.x && pos4.y == curr.y && pos3.x == pos1.x && pos3.y == pos1.y) break;
          pos2 = pos3; pos3 = pos4;
        }
      } else {
        c.pts.push_back(curr);
        at(x, y) = -curr_border_num;
      }
      out.push_back(std::move(c));
    }
  }
}

// test hook: contour k as (x,y) pairs; returns number of contours. Call with
// k<0 to only count; sizes[k] receives the point count when sizes != null.
ORC_API int orc_find_contours(const u8* mask, int h, int w, int k, int* pts_xy, int max_pts, int* n_pts, int* btype) {
  std::vector<Contour> cs; find_contours(mask, h, w, cs);
  if (k >= 0 && k < (int)cs.size()) {
    int n = (int)cs[k].pts.size();
    *n_pts = n; *btype = cs[k].border_type;
    for (int i = 0; i < n && i < max_pts; i++) { pts_xy[2 * i] = cs[k].pts[i].x; pts_xy[2 * i + 1] = cs[k].pts[i].y; }
  }
  return (int)cs.size();
}

// imageproc 0.25.0 geometry::convex_hull + min_area_rect, generic
// over the coordinate type (i32 for contour points, f32 for offset points).
struct DPt { double x, y; };
enum { ORI_COLLINEAR = 0, ORI_CW = 1, ORI_CCW = 2 };
static int orientation(DPt p, DPt q, DPt r) {
  double val = (q.y - p.y) * (r.x - q.x) - (q.x - p.x) * (r.y - q.y);
  if (val == 0.0) return ORI_COLLINEAR;
  return val > 0.0 ? ORI_CW : ORI_CCW;
}
static double dist2(DPt a, DPt b) { return (a.x - b.x) * (a.x - b.x) + (a.y - b.y) * (a.y - b.y); }

static std::vector<DPt> convex_hull(const std::vector<DPt>& in) {
  std::vector<DPt> pts = in;
  if (pts.empty()) return pts;
  size_t sp = 0; DPt start = pts[0];
  for (size_t i = 1; i < pts.size(); i++)
    if (pts[i].y < start.y || (pts[i].y == start.y && pts[i].x < start.x)) { sp = i; start = pts[i]; }
  std::swap(pts[0], pts[sp]);
  pts.erase(pts.begin());
  // sort by polar order around start; collinear -> nearer first
  std::stable_sort(pts.begin(), pts.end(), [&](const DPt& a, const DPt& b) {
    int o = orientation(start, a, b);
    if (o == ORI_COLLINEAR) return dist2(start, a) < dist2(start, b);
    return o == ORI_CCW;
  });
  std::vector<DPt> rem;
  for (size_t i = 0; i < pts.size(); i++) {
    size_t j = i;
    while (j + 1 < pts.size() && orientation(start, pts[j], pts[j + 1]) == ORI_COLLINEAR) j++;
    rem.push_back(pts[j]);
    i = j;
  }
  std::vector<DPt> st; st.push_back(start);
  for (const DPt& p : rem) {
    while (st.size() > 1 && orientation(st[st.size() - 2], st[st.size() - 1], p) != ORI_CCW) st.pop_back();
    st.push_back(p);
  }
  return st;
}

// returns 4 corners TL,TR,BR,BL, each coordinate floor()ed (still double here;
// the caller casts to T).
static void min_area_rect(const std::vector<DPt>& points, double out[8]) {
  std::vector<DPt> hull = convex_hull(points);
  if (hull.size() == 1) { for (int i = 0; i < 4; i++) { out[2 * i] = hull[0].x; out[2 * i + 1] = hull[0].y; } return; }
  if (hull.size() == 2) {
    DPt r[4] = {hull[0], hull[1], hull[1], hull[0]};
    for (int i = 0; i < 4; i++) { out[2 * i] = r[i].x; out[2 * i + 1] = r[i].y; }
    return;
  }
  // rotating_calipers
  const double PI = 3.14159265358979323846264338327950288;
  std::vector<double> angles;
  for (size_t i = 0; i + 1 < hull.size(); i++) {  // points.windows(2): no closing edge
    double ex = hull[i + 1].x - hull[i].x, ey = hull[i + 1].y - hull[i].y;
    double a = fabs(fmod(atan2(ey, ex) + PI, PI / 2.0));
    if (angles.empty() || angles.back() != a) angles.push_back(a);  // Vec::dedup (consecutive)
  }
  double min_area = std::numeric_limits<double>::max();
  DPt res[4] = {{0, 0}, {0, 0}, {0, 0}, {0, 0}};
  for (double angle : angles) {
    double s = sin(angle), c = cos(angle);
    double min_x = std::numeric_limits<double>::max(), max_x = std::numeric_limits<double>::lowest();
    double min_y = std::numeric_limits<double>::max(), max_y = std::numeric_limits<double>::lowest();
    for (const DPt& p : hull) {
      // Point::rotate: x*cos + y*sin, y*cos - x*sin (rotation by -angle)
      double rx = p.x * c + p.y * s;
      double ry = p.y * c - p.x * s;
      min_x = std::min(min_x, rx); max_x = std::max(max_x, rx);
      min_y = std::min(min_y, ry); max_y = std::max(max_y, ry);
    }
    double area = (max_x - min_x) * (max_y - min_y);
    if (area < min_area) {
      min_area = area;
      // Point::invert_rotation: x*cos - y*sin, y*cos + x*sin
      auto inv = [&](double x, double y) -> DPt { return DPt{x * c - y * s, y * c + x * s}; };
      res[0] = inv(max_x, min_y); res[1] = inv(min_x, min_y);
      res[2] = inv(min_x, max_y); res[3] = inv(max_x, max_y);
    }
  }
  std::stable_sort(res, res + 4, [](const DPt& a, const DPt& b) { return a.x < b.x; });
  int i1 = res[1].y > res[0].y ? 0 : 1;
  int i2 = res[3].y > res[2].y ? 2 : 3;
  int i3 = res[3].y > res[2].y ? 3 : 2;
  int i4 = res[1].y > res[0].y ? 1 : 0;
  int idx[4] = {i1, i2, i3, i4};
  for (int i = 0; i < 4; i++) { out[2 * i] = floor(res[idx[i]].x); out[2 * i + 1] = floor(res[idx[i]].y); }
}

// det_processor.rs:176-186 get_mini_boxes: sside from f32 euclid_dist.
static float euclid_f32(float ax, float ay, float bx, float by) {
  float dx = ax - bx, dy = ay - by;
  return sqrtf(dx * dx + dy * dy);
}

ORC_API void orc_min_area_rect(const double* pts_xy, int n, double* out8) {
  std::vector<DPt> p(n);
  for (int i = 0; i < n; i++) p[i] = DPt{pts_xy[2 * i], pts_xy[2 * i + 1]};
  min_area_rect(p, out8);
}

// imageproc draw_polygon_mut into a bw x bh u8 canvas, colour 1.
// Returns -1 if the crate would panic (first point == last point).
static int draw_polygon(std::vector<u8>& canvas, int width, int height, const IPt poly[4]) {
  if (poly[0].x == poly[3].x && poly[0].y == poly[3].y) return -1;
  int y_min = INT32_MAX, y_max = INT32_MIN;
  for (int i = 0; i < 4; i++) { y_min = std::min(y_min, poly[i].y); y_max = std::max(y_max, poly[i].y); }
  y_min = std::max(0, std::min(y_min, height - 1));
  y_max = std::max(0, std::min(y_max, height - 1));
  IPt closed[5] = {poly[0], poly[1], poly[2], poly[3], poly[0]};
  std::vector<int> inter;
  for (int y = y_min; y <= y_max; y++) {
    inter.clear();
    for (int e = 0; e < 4; e++) {
      IPt p0 = closed[e], p1 = closed[e + 1];
      if ((p0.y <= y && p1.y >= y) || (p1.y <= y && p0.y >= y)) {
        if (p0.y == p1.y) { inter.push_back(p0.x); inter.push_back(p1.x); }
        else if (p0.y == y || p1.y == y) {
          if (p1.y > y) inter.push_back(p0.x);
          if (p0.y > y) inter.push_back(p1.x);
        } else {
          float fraction = (float)(y - p0.y) / (float)(p1.y - p0.y);
          float in = (float)p0.x + fraction * (float)(p1.x - p0.x);
          inter.push_back(rs_f32_as_i32(rs_roundf(in)));
        }
      }
    }
    std::sort(inter.begin(), inter.end());
    for (size_t k = 0; k + 1 < inter.size(); k += 2) {
      int from = std::min(inter[k], width);
      int to = std::min(inter[k + 1], width - 1);
      if (from < width && to >= 0) {
        from = std::max(0, from); to = std::max(0, to);
        for (int x = from; x <= to; x++) canvas[(size_t)y * width + x] = 1;
      }
    }
  }
  // edges with BresenhamLineIter (f32 state)
  for (int e = 0; e < 4; e++) {
    float x0 = (float)closed[e].x, y0 = (float)closed[e].y;
    float x1 = (float)closed[e + 1].x, y1 = (float)closed[e + 1].y;
    bool steep = fabsf(y1 - y0) > fabsf(x1 - x0);
    if (steep) { std::swap(x0, y0); std::swap(x1, y1); }
    if (x0 > x1) { std::swap(x0, x1); std::swap(y0, y1); }
    float dx = x1 - x0, dy = fabsf(y1 - y0);
    int x = (int)x0, y = (int)y0, end_x = (int)x1;
    float error = dx / 2.0f;
    int y_step = y0 < y1 ? 1 : -1;
    while (x <= end_x) {
      int px = steep ? y : x, py = steep ? x : y;
      if (px >= 0 && px < width && py >= 0 && py < height) canvas[(size_t)py * width + px] = 1;
      x += 1; error -= dy;
      if (error < 0.0f) { y += y_step; error += dx; }
    }
  }
  return 0;
}

// det_processor.rs:188-221 box_score_fast.  A.4: where draw_polygon_mut would
// panic the oracle defines score = 0 (box dropped).
static float box_score_fast(const float* pred, int h, int w, const IPt box[4]) {
  int x_min = INT32_MAX, x_max = INT32_MIN, y_min = INT32_MAX, y_max = INT32_MIN;
  for (int i = 0; i < 4; i++) {
    x_min = std::min(x_min, box[i].x); x_max = std::max(x_max, box[i].x);
    y_min = std::min(y_min, box[i].y); y_max = std::max(y_max, box[i].y);
  }
  auto cl = [](int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); };
  x_min = cl(x_min, 0, w - 1); x_max = cl(x_max, 0, w - 1);
  y_min = cl(y_min, 0, h - 1); y_max = cl(y_max, 0, h - 1);
  int bw = x_max - x_min + 1, bh = y_max - y_min + 1;
  IPt poly[4];
  for (int i = 0; i < 4; i++) poly[i] = IPt{box[i].x - x_min, box[i].y - y_min};
  std::vector<u8> mask((size_t)bw * bh, 0);
  if (draw_polygon(mask, bw, bh, poly) != 0) return 0.0f;
  float sum = 0.0f; size_t count = 0;
  for (int y = 0; y < bh; y++)
    for (int x = 0; x < bw; x++) {
      u8 m = mask[(size_t)y * bw + x];
      float v = pred[(size_t)(y + y_min) * w + (x + x_min)];
      sum = sum + v * (float)m;
      count += m;
    }
  return count > 0 ? sum / (float)count : 0.0f;
}

ORC_API float orc_box_score_fast(const float* pred, int h, int w, const int* box8) {
  IPt b[4]; for (int i = 0; i < 4; i++) b[i] = IPt{box8[2 * i], box8[2 * i + 1]};
  return box_score_fast(pred, h, w, b);
}

// det_processor.rs:223-252 unclip: geo area / perimeter in f32, then Clipper
// 6.4.2 ClipperOffset(miterLimit 0 -> 0.5, arcTolerance 0.5), AddPath(jtRound,
// etClosedPolygon), Execute(delta).
//
// Restatement choice: ClipperOffset::Execute finishes with a Clipper union
// (pftPositive) that only removes self-overlap of the offset path.  For the
// simple, positively oriented offset of a convex quad (the only input this
// path produces) that union returns the same vertex set minus collinear /
// duplicate vertices; the sole consumer is min_area_rect, which depends only on
// the convex hull of the vertex set.  The union is therefore restated as the
// identity and the raw offset path is returned.  Bow-tie quads (possible only
// through min_area_rect's sort-by-x corner assignment on degenerate input) are
// outside the contract.
typedef long long cInt;
static inline cInt cl_round(double v) { return v < 0 ? (cInt)(v - 0.5) : (cInt)(v + 0.5); }
struct CPt { cInt X, Y; };
struct CDPt { double X, Y; };

static double cl_area(const std::vector<CPt>& poly) {
  int size = (int)poly.size();
  if (size < 3) return 0;
  double a = 0;
  for (int i = 0, j = size - 1; i < size; ++i) { a += ((double)poly[j].X + poly[i].X) * ((double)poly[j].Y - poly[i].Y); j = i; }
  return -a * 0.5;
}
static CDPt cl_unit_normal(const CPt& p1, const CPt& p2) {
  if (p2.X == p1.X && p2.Y == p1.Y) return CDPt{0, 0};
  double Dx = (double)(p2.X - p1.X), dy = (double)(p2.Y - p1.Y);
  double f = 1 * 1.0 / std::sqrt(Dx * Dx + dy * dy);
  Dx *= f; dy *= f;
  return CDPt{dy, -Dx};
}

static void clipper_offset_round(const std::vector<CPt>& path_in, double delta, double arc_tolerance,
                                 std::vector<CPt>& dest) {
  dest.clear();
  // ClipperOffset::AddPath: strip closing / consecutive duplicates
  int highI = (int)path_in.size() - 1;
  if (highI < 0) return;
  while (highI > 0 && path_in[0].X == path_in[highI].X && path_in[0].Y == path_in[highI].Y) highI--;
  std::vector<CPt> src; src.push_back(path_in[0]);
  for (int i = 1; i <= highI; i++)
    if (src.back().X != path_in[i].X || src.back().Y != path_in[i].Y) src.push_back(path_in[i]);
  if ((int)src.size() < 3) return;
  // FixOrientations: single closed path; reverse if orientation is false
  if (!(cl_area(src) >= 0)) std::reverse(src.begin(), src.end());
  // DoOffset
  const double pi = 3.141592653589793238, two_pi = pi * 2, def_arc_tolerance = 0.25;
  if (std::fabs(delta) < 1.0E-20) { dest = src; return; }
  double y;
  if (arc_tolerance <= 0.0) y = def_arc_tolerance;
  else if (arc_tolerance > std::fabs(delta) * def_arc_tolerance) y = std::fabs(delta) * def_arc_tolerance;
  else y = arc_tolerance;
  double steps = pi / std::acos(1 - y / std::fabs(delta));
  if (steps > std::fabs(delta) * pi) steps = std::fabs(delta) * pi;
  double m_sin = std::sin(two_pi / steps), m_cos = std::cos(two_pi / steps);
  double steps_per_rad = steps / two_pi;
  if (delta < 0.0) m_sin = -m_sin;
  int len = (int)src.size();
  if (delta <= 0 && len < 3) return;
  std::vector<CDPt> normals;
  for (int j = 0; j < len - 1; ++j) normals.push_back(cl_unit_normal(src[j], src[j + 1]));
  normals.push_back(cl_unit_normal(src[len - 1], src[0]));
  int k = len - 1;
  for (int j = 0; j < len; ++j) {
    // OffsetPoint(j, k, jtRound)
    double sinA = normals[k].X * normals[j].Y - normals[j].X * normals[k].Y;
    bool done = false;
    if (std::fabs(sinA * delta) < 1.0) {
      double cosA = normals[k].X * normals[j].X + normals[j].Y * normals[k].Y;
      if (cosA > 0) {
        dest.push_back(CPt{cl_round(src[j].X + normals[k].X * delta), cl_round(src[j].Y + normals[k].Y * delta)});
        done = true;
      }
    } else if (sinA > 1.0) sinA = 1.0;
    else if (sinA < -1.0) sinA = -1.0;
    if (!done) {
      if (sinA * delta < 0) {
        dest.push_back(CPt{cl_round(src[j].X + normals[k].X * delta), cl_round(src[j].Y + normals[k].Y * delta)});
        dest.push_back(src[j]);
        dest.push_back(CPt{cl_round(src[j].X + normals[j].X * delta), cl_round(src[j].Y + normals[j].Y * delta)});
      } else {
        // DoRound
        double a = std::atan2(sinA, normals[k].X * normals[j].X + normals[k].Y * normals[j].Y);
        int nsteps = std::max((int)cl_round(steps_per_rad * std::fabs(a)), 1);
        double X = normals[k].X, Y = normals[k].Y, X2;
        for (int i = 0; i < nsteps; ++i) {
          dest.push_back(CPt{cl_round(src[j].X + X * delta), cl_round(src[j].Y + Y * delta)});
          X2 = X;
          X = X * m_cos - m_sin * Y;
          Y = X2 * m_sin + Y * m_cos;
        }
        dest.push_back(CPt{cl_round(src[j].X + normals[j].X * delta), cl_round(src[j].Y + normals[j].Y * delta)});
      }
    }
    k = j;
  }
}

// Returns the offset ring as f32 points (closed: first point repeated, as
// geo-clipper's to_geo does).  box is TL,TR,BR,BL ints.
static void unclip(const IPt box[4], float unclip_ratio, std::vector<Pt>& out) {
  out.clear();
  // geo Polygon::new closes the ring: 5 coords, f32
  float cx[5], cy[5];
  for (int i = 0; i < 4; i++) { cx[i] = (float)box[i].x; cy[i] = (float)box[i].y; }
  cx[4] = cx[0]; cy[4] = cy[0];
  bool already_closed = false;  // geo only appends when first != last; 4 distinct-or-not points
  (void)already_closed;
  // geo 0.30 unsigned_area: shoelace with coordinates shifted by the first
  // point, f32; |sum| / 2
  float shift_x = cx[0], shift_y = cy[0];
  float tmp = 0.0f;
  for (int i = 0; i < 4; i++) {
    float ax = cx[i] - shift_x, ay = cy[i] - shift_y;
    float bx = cx[i + 1] - shift_x, by = cy[i + 1] - shift_y;
    tmp += ax * by - bx * ay;  // Line::determinant
  }
  float area = fabsf(tmp / 2.0f);
  // Euclidean.length(LineString): sum of segment lengths (hypot) in f32
  float perimeter = 0.0f;
  for (int i = 0; i < 4; i++) {
    // geo Euclidean.distance = delta.x.hypot(delta.y); glibc's hypotf is the
    // correctly rounded (float)sqrt((double)x*x + (double)y*y) -- written out so
    // the restatement does not depend on the host libm.
    float dx = cx[i] - cx[i + 1], dy = cy[i] - cy[i + 1];
    perimeter += (float)sqrt((double)dx * (double)dx + (double)dy * (double)dy);
  }
  perimeter = perimeter + 0.0f;  // + sum over (no) interiors
  float distance = area * unclip_ratio / perimeter;
  std::vector<CPt> path;
  for (int i = 0; i < 5; i++) path.push_back(CPt{(cInt)(cx[i] * 1.0f), (cInt)(cy[i] * 1.0f)});
  std::vector<CPt> dest;
  clipper_offset_round(path, (double)(distance * 1.0f), 0.5, dest);
  if (dest.size() < 3) return;  // Clipper drops degenerate output: no polygons
  for (const CPt& p : dest) out.push_back(Pt{(float)((double)p.X / 1.0), (float)((double)p.Y / 1.0)});
  out.push_back(out[0]);
}

// pin kit (tests/golden/pin): the offset distance det_processor.rs:236-240 hands to Clipper, as unclip() computes it
ORC_API float orc_unclip_distance(const int* box8, float ratio) {
  float cx[5], cy[5];
  for (int i = 0; i < 4; i++) { cx[i] = (float)box8[2 * i]; cy[i] = (float)box8[2 * i + 1]; }
  cx[4] = cx[0]; cy[4] = cy[0];
  float tmp = 0.0f;
  for (int i = 0; i < 4; i++) {
    float ax = cx[i] - cx[0], ay = cy[i] - cy[0], bx = cx[i + 1] - cx[0], by = cy[i + 1] - cy[0];
    tmp += ax * by - bx * ay;
  }
  float area = fabsf(tmp / 2.0f), perimeter = 0.0f;
  for (int i = 0; i < 4; i++) {
    float dx = cx[i] - cx[i + 1], dy = cy[i] - cy[i + 1];
    perimeter += (float)sqrt((double)dx * (double)dx + (double)dy * (double)dy);
  }
  perimeter = perimeter + 0.0f;
  return area * ratio / perimeter;
}

ORC_API int orc_unclip(const int* box8, float ratio, float* out_xy, int max_pts) {
  IPt b[4]; for (int i = 0; i < 4; i++) b[i] = IPt{box8[2 * i], box8[2 * i + 1]};
  std::vector<Pt> o; unclip(b, ratio, o);
  for (size_t i = 0; i < o.size() && (int)i < max_pts; i++) { out_xy[2 * i] = o[i].x; out_xy[2 * i + 1] = o[i].y; }
  return (int)o.size();
}

// ---------------------------------------------------------------------------
// a5: DetProcessor::postprocess (det_processor.rs:279-335)
// boxes_out: n x 8 floats (TL,TR,BR,BL x,y) in ori (= after_*) coordinates.
// ---------------------------------------------------------------------------
struct DetBox { float pts[8]; float score; };

ORC_API int orc_det_postprocess(const float* pred, int h, int w, int ori_h, int ori_w, float thresh, float box_thresh,
                                float unclip_ratio, int min_mini_box_size, int dilate, float* boxes_out,
                                float* scores_out, int max_out) {
  std::vector<u8> mask((size_t)h * w);
  orc_threshold_dilate(pred, h, w, thresh, dilate, mask.data());
  std::vector<Contour> contours; find_contours(mask.data(), h, w, contours);
  std::vector<DetBox> res;
  for (const Contour& c : contours) {
    std::vector<DPt> p; p.reserve(c.pts.size());
    for (const IPt& q : c.pts) p.push_back(DPt{(double)q.x, (double)q.y});
    double r[8]; min_area_rect(p, r);
    IPt box[4];
    for (int i = 0; i < 4; i++) box[i] = IPt{(int)r[2 * i], (int)r[2 * i + 1]};
    float s1 = euclid_f32((float)box[0].x, (float)box[0].y, (float)box[1].x, (float)box[1].y);
    float s2 = euclid_f32((float)box[3].x, (float)box[3].y, (float)box[2].x, (float)box[2].y);
    float sside = std::min(s1, s2);
    if (sside < (float)min_mini_box_size) continue;
    float mean_score = box_score_fast(pred, h, w, box);
    if (mean_score < box_thresh) continue;
    std::vector<Pt> off; unclip(box, unclip_ratio, off);
    if (off.empty()) continue;  // min_area_rect would panic on no points; define: drop
    std::vector<DPt> op; for (const Pt& q : off) op.push_back(DPt{(double)q.x, (double)q.y});
    double r2[8]; min_area_rect(op, r2);
    DetBox b;
    for (int i = 0; i < 8; i++) b.pts[i] = (float)r2[i];
    float t1 = euclid_f32(b.pts[0], b.pts[1], b.pts[2], b.pts[3]);
    float t2 = euclid_f32(b.pts[6], b.pts[7], b.pts[4], b.pts[5]);
    float ss2 = std::min(t1, t2);
    if (ss2 < (float)(min_mini_box_size + 2)) continue;
    orc_scale_and_clip(b.pts, (double)w, (double)h, (double)ori_w, (double)ori_h);
    float pb_h = side_len(&b.pts[0], &b.pts[6]);  // height_tlc: TL-BL
    float pb_w = side_len(&b.pts[0], &b.pts[2]);  // width_tlc: TL-TR
    if (pb_h <= 3.0f || pb_w <= 3.0f) continue;
    b.score = mean_score;
    res.push_back(b);
  }
  // sorted_boxes (det_processor.rs:324-333): Rust's stable sort_by with a comparator on
  // the centre (TL+BR)/2 (points.rs:173-177).  The comparator is not a total order
  // (SURVEY A.5), so the outcome on inconsistent inputs is sort-algorithm specific;
  // restatement choice: a stable bottom-up merge sort (run width 1, 2, 4, ...) over the
  // contour discovery order.  Every stable sort agrees with it on strict weak orders.
  {
    auto less = [](const DetBox& a, const DetBox& b) {
      float y1 = (a.pts[1] + a.pts[5]) / 2.0f, y2 = (b.pts[1] + b.pts[5]) / 2.0f;
      if (fabsf(y1 - y2) < 10.0f) {
        float x1 = (a.pts[0] + a.pts[4]) / 2.0f, x2 = (b.pts[0] + b.pts[4]) / 2.0f;
        return x1 < x2;
      }
      return y1 < y2;
    };
    std::vector<DetBox> tmp(res.size());
    int nn = (int)res.size();
    std::vector<DetBox>*src = &res, *dst = &tmp;
    for (int width = 1; width < nn; width *= 2) {
      for (int lo = 0; lo < nn; lo += 2 * width) {
        int mid = std::min(lo + width, nn), hi = std::min(lo + 2 * width, nn);
        int i = lo, j = mid, k = lo;
        while (i < mid && j < hi) {
          if (less((*src)[j], (*src)[i])) (*dst)[k++] = (*src)[j++];
          else (*dst)[k++] = (*src)[i++];
        }
        while (i < mid) (*dst)[k++] = (*src)[i++];
        while (j < hi) (*dst)[k++] = (*src)[j++];
      }
      std::swap(src, dst);
    }
    if (src != &res) res = *src;
  }
  int n = (int)res.size();
  for (int i = 0; i < n && i < max_out; i++) {
    memcpy(boxes_out + 8 * i, res[i].pts, 8 * sizeof(float));
    scores_out[i] = res[i].score;
  }
  return n;
}

// ---------------------------------------------------------------------------
// a6: get_crop_img (image_helper.rs:223-249) + imageproc Projection /
// warp_into(Bicubic) + image::imageops::rotate270.
//
// Restatement choice: imageproc solves the 8x8 DLT system with an f64 SVD and
// stores the result as f32.  The oracle solves the same system in f64 by
// Gaussian elimination with partial pivoting, then casts to f32; the 3x3
// inverse is computed in f32 (adjugate / determinant) like imageproc's
// try_inverse.
// ---------------------------------------------------------------------------
static bool solve8(double A[8][9]) {
  for (int c = 0; c < 8; c++) {
    int p = c; double best = fabs(A[c][c]);
    for (int r = c + 1; r < 8; r++) if (fabs(A[r][c]) > best) { best = fabs(A[r][c]); p = r; }
    if (best == 0.0) return false;
    if (p != c) for (int k = 0; k < 9; k++) std::swap(A[c][k], A[p][k]);
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
  return true;
}

// from = box points, to = rectangle corners. Outputs forward transform t[9]
// and its inverse inv[9] (f32), class: 0 translation 1 affine 2 projection.
static bool projection_from_control_points(const float from[8], const float to[8], float t[9], float inv[9], int* cls) {
  double A[8][9];
  for (int i = 0; i < 4; i++) {
    double xf = (double)from[2 * i], yf = (double)from[2 * i + 1];
    double x = (double)to[2 * i], y = (double)to[2 * i + 1];
    double r0[9] = {0.0, 0.0, 0.0, -xf, -yf, -1.0, y * xf, y * yf, -y};
    double r1[9] = {xf, yf, 1.0, 0.0, 0.0, 0.0, -x * xf, -x * yf, x};
    for (int k = 0; k < 9; k++) { A[2 * i][k] = r0[k]; A[2 * i + 1][k] = r1[k]; }
  }
  if (!solve8(A)) return false;
  for (int i = 0; i < 8; i++) t[i] = (float)A[i][8];
  t[8] = 1.0f;
  // normalize(): divide by t[8] (== 1)
  for (int i = 0; i < 9; i++) t[i] = t[i] / t[8];
  // from_control_points tags its result TransformationClass::Projection
  // unconditionally, so warp_into always takes map_projective.
  *cls = 2;
  // try_inverse (f32)
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
  for (int i = 0; i < 9; i++) inv[i] = inv[i] / n8;  // normalize(inv)
  return true;
}

static inline u8 clamp_u8_trunc(float x) {  // imageproc Clamp<f32> for u8
  if (x < 255.0f) { if (x > 0.0f) return (u8)x; return 0; }
  return 255;
}
static inline float cubic(float p0, float p1, float p2, float p3, float x) {
  return p1 + 0.5f * x * (p2 - p0 + x * (2.0f * p0 - 5.0f * p1 + 4.0f * p2 - p3 + x * (3.0f * (p1 - p2) + p3 - p0)));
}

// image_helper.rs:224-226: crop dims (truncating cast) and the rotate flag.
ORC_API void orc_crop_dims(const float* box8, int* out_w, int* out_h, int* rotated, float* fw, float* fh) {
  const float* tl = box8; const float* tr = box8 + 2; const float* br = box8 + 4; const float* bl = box8 + 6;
  float w_brc = side_len(bl, br), w_tlc = side_len(tl, tr);
  float h_brc = side_len(tr, br), h_tlc = side_len(tl, bl);
  float cw = std::max(w_brc, w_tlc), ch = std::max(h_brc, h_tlc);
  uint32_t w = rs_f32_as_u32(cw), h = rs_f32_as_u32(ch);
  *fw = cw; *fh = ch;
  *rotated = (w > 0 && (float)h / (float)w >= 1.5f) ? 1 : ((w == 0 && h > 0) ? 1 : 0);  // h/0 = inf >= 1.5
  if (w == 0 && h == 0) *rotated = 0;  // 0/0 = NaN
  if (*rotated) { *out_w = (int)h; *out_h = (int)w; } else { *out_w = (int)w; *out_h = (int)h; }
}

// pin kit: the projection get_crop_img builds for a box (forward matrix and the inverse warp_into samples with), f32 row-major
ORC_API int orc_crop_projection(const float* box8, float* t9, float* inv9) {
  int ow, oh, rot; float cw, ch;
  orc_crop_dims(box8, &ow, &oh, &rot, &cw, &ch);
  float to[8] = {0.0f, 0.0f, cw, 0.0f, cw, ch, 0.0f, ch};
  int cls;
  return projection_from_control_points(box8, to, t9, inv9, &cls) ? 0 : -1;
}

// out must hold out_w*out_h*3 bytes (dims from orc_crop_dims). Returns 0, or
// -1 when the homography is singular (reference unwrap() would panic).
ORC_API int orc_get_crop_img(const u8* src, int sh, int sw, const float* box8, u8* out) {
  int ow, oh, rot; float cw, ch;
  orc_crop_dims(box8, &ow, &oh, &rot, &cw, &ch);
  int w = rot ? oh : ow, h = rot ? ow : oh;  // pre-rotation dims
  float to[8] = {0.0f, 0.0f, cw, 0.0f, cw, ch, 0.0f, ch};
  float t[9], inv[9]; int cls;
  if (!projection_from_control_points(box8, to, t, inv, &cls)) return -1;
  // warp_into uses projection.invert(): maps OUTPUT coords to SOURCE coords with `inv`
  std::vector<u8> tmp((size_t)w * h * 3);
  for (int y = 0; y < h; y++)
    for (int x = 0; x < w; x++) {
      float fx = (float)x, fy = (float)y, px, py;
      if (cls == 2) {
        float d = inv[6] * fx + inv[7] * fy + inv[8];
        px = (inv[0] * fx + inv[1] * fy + inv[2]) / d;
        py = (inv[3] * fx + inv[4] * fy + inv[5]) / d;
      } else if (cls == 1) {
        px = inv[0] * fx + inv[1] * fy + inv[2];
        py = inv[3] * fx + inv[4] * fy + inv[5];
      } else {
        px = fx + inv[2]; py = fy + inv[5];
      }
      u8* o = &tmp[((size_t)y * w + x) * 3];
      float left = floorf(px) - 1.0f, right = left + 4.0f;
      float top = floorf(py) - 1.0f, bottom = top + 4.0f;
      float xw = px - (left + 1.0f), yw = py - (top + 1.0f);
      if (!(left >= 0.0f) || !(right < (float)sw) || !(top >= 0.0f) || !(bottom < (float)sh)) {
        o[0] = o[1] = o[2] = 255;
        continue;
      }
      uint32_t l = rs_f32_as_u32(left), tp = rs_f32_as_u32(top);
      for (int c = 0; c < 3; c++) {
        u8 col[4];
        for (uint32_t r = 0; r < 4; r++) {
          const u8* row = src + ((size_t)(tp + r) * sw + l) * 3 + c;
          col[r] = clamp_u8_trunc(cubic((float)row[0], (float)row[3], (float)row[6], (float)row[9], xw));
        }
        o[c] = clamp_u8_trunc(cubic((float)col[0], (float)col[1], (float)col[2], (float)col[3], yw));
      }
    }
  if (!rot) { memcpy(out, tmp.data(), tmp.size()); return 0; }
  // image::imageops::rotate270: out(y, w-1-x) = in(x, y); out dims (h, w)
  for (int y = 0; y < h; y++)
    for (int x = 0; x < w; x++) {
      int nx = y, ny = w - 1 - x;
      memcpy(out + ((size_t)ny * h + nx) * 3, &tmp[((size_t)y * w + x) * 3], 3);
    }
  return 0;
}

// image::imageops::rotate180_in_place (image_helper.rs:268-286)
ORC_API void orc_rotate180(u8* img, int h, int w) {
  size_t n = (size_t)h * w;
  for (size_t i = 0; i < n / 2; i++) {
    u8 t[3]; memcpy(t, img + i * 3, 3);
    memcpy(img + i * 3, img + (n - 1 - i) * 3, 3);
    memcpy(img + (n - 1 - i) * 3, t, 3);
  }
}

// ---------------------------------------------------------------------------
// a8/a10: resize_norm_image (image_helper.rs:176-209).  ori_h/ori_w are the
// crop's construction-time dims; img_w_final = (img_h*mr) as usize if mr > 0
// else img_w.  out: [3, img_h, img_w_final] CHW, RGB order, zero padded.
// ---------------------------------------------------------------------------
ORC_API int orc_resize_norm_width(int img_h, int img_w, float max_wh_ratio) {
  if (max_wh_ratio > 0.0f) { float v = (float)img_h * max_wh_ratio; return v >= 0 ? (int)(size_t)v : 0; }
  return img_w;
}
ORC_API int orc_resize_norm_image(const u8* crop, int h, int w, int ori_h, int ori_w, int img_h, int img_w,
                                  float max_wh_ratio, float* out) {
  int W = orc_resize_norm_width(img_h, img_w, max_wh_ratio);
  double rw = ceil((double)img_h * (double)(uint32_t)ori_w / (double)(uint32_t)ori_h);
  size_t rwz = rw >= 0 ? (rw > 1.8e19 ? (size_t)-1 : (size_t)rw) : 0;  // f64 as usize saturates; NaN -> 0
  if (rw != rw) rwz = 0;
  int resized_w = (int)std::min<size_t>((size_t)W, rwz);
  std::vector<u8> rs((size_t)img_h * std::max(resized_w, 1) * 3);
  int rc = orc_thumbnail(crop, h, w, rs.data(), img_h, resized_w);
  size_t plane = (size_t)img_h * W;
  for (size_t i = 0; i < 3 * plane; i++) out[i] = 0.0f;
  for (int c = 0; c < 3; c++)
    for (int y = 0; y < img_h; y++)
      for (int x = 0; x < resized_w; x++) {
        float v = (float)rs[((size_t)y * resized_w + x) * 3 + c] / 255.0f;
        out[(size_t)c * plane + (size_t)y * W + x] = (v - 0.5f) / 0.5f;
      }
  return rc;
}

// ---------------------------------------------------------------------------
// a12: RecProcessor::postprocess + RecCharacter::decode
// (rec_processor.rs:190-208, :48-97); argmax = first maximum (B.7).
// probs [n,T,C]; idx_out [n,T] (argmax), prob_out [n,T]; tok_out [n,T] kept
// token ids (first tok_n[i] valid); score_out[i] = sum/count (NaN if none).
// ---------------------------------------------------------------------------
ORC_API void orc_ctc_decode(const float* probs, int n, int T, int C, int* idx_out, float* prob_out, int* tok_out,
                            int* tok_n, float* score_out) {
  for (int i = 0; i < n; i++) {
    for (int t = 0; t < T; t++) {
      const float* row = probs + ((size_t)i * T + t) * C;
      int best = 0; float bv = row[0];
      for (int c = 1; c < C; c++) if (row[c] > bv) { bv = row[c]; best = c; }
      idx_out[(size_t)i * T + t] = best; prob_out[(size_t)i * T + t] = bv;
    }
    int cnt = 0; float acc = 0.0f;
    for (int t = 0; t < T; t++) {
      int id = idx_out[(size_t)i * T + t];
      bool sel = id != 0;
      if (t >= 1) sel = sel && id != idx_out[(size_t)i * T + t - 1];
      sel = sel && id != 0;  // ignored_tokens = [0]
      if (sel) { tok_out[(size_t)i * T + cnt] = id; acc = acc + prob_out[(size_t)i * T + t]; cnt++; }
    }
    tok_n[i] = cnt;
    score_out[i] = acc / (float)(uint32_t)cnt;  // 0/0 = NaN (rec_processor.rs:94)
  }
}

// cls postprocess (cls_processor.rs:108-121): first-max argmax over 2 classes.
ORC_API void orc_cls_postprocess(const float* probs, int n, int C, int* idx_out, float* score_out) {
  for (int i = 0; i < n; i++) {
    const float* row = probs + (size_t)i * C;
    int best = 0; float bv = row[0];
    for (int c = 1; c < C; c++) if (row[c] > bv) { bv = row[c]; best = c; }
    idx_out[i] = best; score_out[i] = bv;
  }
}
