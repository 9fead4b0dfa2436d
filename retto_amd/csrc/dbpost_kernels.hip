// DB post-processing (det_processor.rs:279-335) for a batch of pages, gfx950.
// threshold + 2x2 dilate -> connected components (8-connected foreground, 4-connected
// background, as imageproc's Suzuki-Abe border following implies) -> one contour per
// foreground component (outer border) and per hole -> min-area rect -> score -> Clipper
// offset -> min-area rect -> scale/clip/filter -> reading-order sort.
// Every kernel takes the page table and uses blockIdx.y = page, so all pages of a batch
// share each launch.  Compiled with -ffp-contract=off (bit-exact f32/f64).
#include "prepost.h"

#include "geom_math.h"

namespace rt {
namespace pp {

typedef unsigned char u8;

struct Contour { int root, type, ymin, rows, base, hbase; };
struct DbWs {
  u8* mask; u8* outside;
  int *parent, *ymin, *ymax, *cidx;
  int* counters;  // 0 nContours 1 rowUsed 2 hullUsed 3 nCand 4 overflow
  Contour* contours; int contour_cap;
  int *rowmin, *rowmax; int row_cap;
  int2* hull; int hull_cap;
  DbBox* cand; int cand_cap;
  unsigned *sort_a, *sort_b;   // index runs of the reading-order sort when a page has more candidates than fit in LDS
  float *sort_cx, *sort_cy;
};
struct DbPage {
  const float* pred; int H, W, ori_h, ori_w;
  DbWs ws;
  DbBox* boxes_out; int* count_out;
};

static inline size_t al256(size_t v) { return (v + 255) & ~(size_t)255; }
static DbWs carve(void* base, int H, int W, int max_boxes, size_t* total) {
  size_t N = (size_t)H * W, o = 0;
  DbWs ws;
  char* b = (char*)base;
  auto take = [&](size_t bytes) { char* p = b ? b + o : nullptr; o += al256(bytes); return p; };
  ws.mask = (u8*)take(N); ws.outside = (u8*)take(N);
  ws.parent = (int*)take(N * 4); ws.ymin = (int*)take(N * 4); ws.ymax = (int*)take(N * 4); ws.cidx = (int*)take(N * 4);
  ws.counters = (int*)take(64);
  ws.contour_cap = (int)(N / 2 + 1024);
  ws.contours = (Contour*)take((size_t)ws.contour_cap * sizeof(Contour));
  ws.row_cap = (int)(3 * N + 1024);
  ws.rowmin = (int*)take((size_t)ws.row_cap * 4); ws.rowmax = (int*)take((size_t)ws.row_cap * 4);
  ws.hull_cap = (int)(2 * (size_t)ws.row_cap + 4 * (size_t)ws.contour_cap);
  ws.hull = (int2*)take((size_t)ws.hull_cap * sizeof(int2));
  ws.cand_cap = max_boxes;
  ws.cand = (DbBox*)take((size_t)max_boxes * sizeof(DbBox));
  ws.sort_a = (unsigned*)take((size_t)max_boxes * 4); ws.sort_b = (unsigned*)take((size_t)max_boxes * 4);
  ws.sort_cx = (float*)take((size_t)max_boxes * 4); ws.sort_cy = (float*)take((size_t)max_boxes * 4);
  *total = o;
  return ws;
}
size_t db_workspace_bytes(int H, int W, int max_boxes) { size_t t; carve(nullptr, H, W, max_boxes, &t); return t; }
size_t db_page_desc_bytes() { return sizeof(DbPage); }

__device__ __forceinline__ int uf_find(const int* parent, int x) {
  int p = parent[x];
  while (p != x) { x = p; p = parent[x]; }
  return x;
}
__device__ __forceinline__ void uf_union(int* parent, int a, int b) {
  while (true) {
    a = uf_find(parent, a); b = uf_find(parent, b);
    if (a == b) return;
    if (a > b) { int t = a; a = b; b = t; }
    int old = atomicMin(&parent[b], a);
    if (old == b) return;
    b = old;
  }
}

// det_processor.rs:286-292: mask = pred > thresh; grayscale_dilate with offsets {(-1,-1),(0,-1),(-1,0),(0,0)} -- and, in the
// same pass, connected components step 1: every pixel points at the first pixel of its horizontal run (same mask value).
// One block per image row; block-wide max-scan of run starts.  (Round 5: until then k_db_mask was its own pass that also reset
// four per-pixel work arrays -- 18 bytes written per pixel, 0.18 ms per 32 pages.  ymin / ymax / cidx / outside are only ever
// read or updated at the ROOT of a component, and a root is the smallest pixel index of its component, i.e. the first pixel
// of a run: they are initialised at run starts only, here.)
__global__ __launch_bounds__(256) void k_ccl_rows(const DbPage* __restrict__ pages, float thresh, int dilate) {
  const DbPage pg = pages[blockIdx.y];
  const int W = pg.W;
  int y = blockIdx.x;
  if (y >= pg.H) return;
  __shared__ int wave_max[4];
  __shared__ int carry_s;
  const DbWs& ws = pg.ws;
  if (y == 0 && threadIdx.x < 8) ws.counters[threadIdx.x] = 0;
  const float* prow0 = pg.pred + (size_t)y * W;
  const float* prow1 = prow0 - W;   // (read only when y > 0)
  u8* mrow = ws.mask + (size_t)y * W;
  int* prow = ws.parent + (size_t)y * W;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const bool up = dilate && y > 0;
  // t(x) = the column's own two taps; mask(x) = t(x) | t(x - 1) under dilation
  auto t = [&](int x) { return prow0[x] > thresh || (up && prow1[x] > thresh); };
  int carry = 0;
  if ((W & 3) == 0 && ((size_t)pg.pred & 15) == 0) {
    // four pixels per thread (16-byte loads of the map, one 16-byte store of the run pointers, one 4-byte store of the mask): a
    // 960-pixel row is one pass of the block instead of four
    typedef float f4 __attribute__((ext_vector_type(4)));
    typedef int i4 __attribute__((ext_vector_type(4)));
    for (int x0 = 0; x0 < W; x0 += 1024) {
      const int x = x0 + 4 * threadIdx.x;
      int v[4] = {-1, -1, -1, -1};
      bool m[4] = {false, false, false, false};
      if (x < W) {
        const f4 a0 = *reinterpret_cast<const f4*>(prow0 + x);
        f4 a1 = {0.f, 0.f, 0.f, 0.f}, l0 = {0.f, 0.f, 0.f, 0.f}, l1 = {0.f, 0.f, 0.f, 0.f};
        if (up) a1 = *reinterpret_cast<const f4*>(prow1 + x);
        if (x > 0) { l0 = *reinterpret_cast<const f4*>(prow0 + x - 4); if (up) l1 = *reinterpret_cast<const f4*>(prow1 + x - 4); }
        bool t[6];   // t(x - 2) .. t(x + 3)
        t[0] = x > 0 && (l0[2] > thresh || (up && l1[2] > thresh));
        t[1] = x > 0 && (l0[3] > thresh || (up && l1[3] > thresh));
#pragma unroll
        for (int k = 0; k < 4; k++) t[2 + k] = a0[k] > thresh || (up && a1[k] > thresh);
        bool ml = dilate ? (t[1] || t[0]) : t[1];   // mask of pixel x - 1
#pragma unroll
        for (int k = 0; k < 4; k++) {
          m[k] = dilate ? (t[2 + k] || t[1 + k]) : t[2 + k];
          if (x + k == 0 || ml != m[k]) {
            v[k] = x + k;
            const size_t i = (size_t)y * W + x + k;
            ws.ymin[i] = 0x7fffffff; ws.ymax[i] = -1; ws.cidx[i] = -1; ws.outside[i] = 0;
          }
          ml = m[k];
        }
        *reinterpret_cast<unsigned*>(mrow + x) = (m[0] ? 0xffu : 0u) | (m[1] ? 0xff00u : 0u) | (m[2] ? 0xff0000u : 0u) | (m[3] ? 0xff000000u : 0u);
      }
      // inclusive max-scan: inside the thread, then of the threads' maxima inside the wave, then over the waves
      v[1] = max(v[1], v[0]); v[2] = max(v[2], v[1]); v[3] = max(v[3], v[2]);
      int tv = v[3];
      for (int o = 1; o < 64; o <<= 1) { int tt = __shfl_up(tv, o); if (lane >= o) tv = max(tv, tt); }
      if (lane == 63) wave_max[wave] = tv;
      __syncthreads();
      int pre = carry;
      for (int k = 0; k < wave; k++) pre = max(pre, wave_max[k]);
      const int before = __shfl_up(tv, 1);             // maximum of the earlier threads of this wave
      if (lane > 0) pre = max(pre, before);
      if (x < W) {
        i4 o4;
#pragma unroll
        for (int k = 0; k < 4; k++) o4[k] = y * W + max(v[k], pre);
        *reinterpret_cast<i4*>(prow + x) = o4;
      }
      if (threadIdx.x == 255) carry_s = max(tv, pre);
      __syncthreads();
      carry = carry_s;
    }
    return;
  }
  for (int x0 = 0; x0 < W; x0 += 256) {
    int x = x0 + threadIdx.x;
    int v = -1;
    if (x < W) {
      const bool t0 = t(x), t1 = x > 0 && t(x - 1), t2 = x > 1 && t(x - 2);
      const bool m = dilate ? (t0 || t1) : t0;
      const bool ml = dilate ? (t1 || t2) : t1;    // mask of the left neighbour (x > 0)
      mrow[x] = m ? 255 : 0;
      if (x == 0 || ml != m) {
        v = x;
        const size_t i = (size_t)y * W + x;
        ws.ymin[i] = 0x7fffffff; ws.ymax[i] = -1; ws.cidx[i] = -1; ws.outside[i] = 0;
      }
    }
    // inclusive max-scan inside the wave
    for (int o = 1; o < 64; o <<= 1) { int tt = __shfl_up(v, o); if (lane >= o) v = max(v, tt); }
    if (lane == 63) wave_max[wave] = v;
    __syncthreads();
    int pre = carry;
    for (int k = 0; k < wave; k++) pre = max(pre, wave_max[k]);
    v = max(v, pre);
    if (x < W) prow[x] = y * W + v;
    if (threadIdx.x == 255) carry_s = v;
    __syncthreads();
    carry = carry_s;
  }
}
// step 2: link vertically / diagonally adjacent runs, once per pair (at the first column
// of their overlap).  Foreground is 8-connected, background 4-connected.
// (round 5: a thread takes FOUR pixels when the row length allows -- two 4-byte loads tell it whether any of them or their
//  upper neighbours starts a run; inside uniform stretches, i.e. nearly everywhere, nothing is linked and the thread is done)
__global__ __launch_bounds__(256) void k_ccl_link(const DbPage* __restrict__ pages) {
  const DbPage pg = pages[blockIdx.y];
  const int H = pg.H, W = pg.W;
  const u8* mask = pg.ws.mask;
  int* parent = pg.ws.parent;
  auto link1 = [&](int i, int x) {   // the per-pixel rule; y > 0
    const u8 m = mask[i];
    const bool my_start = (x == 0) || mask[i - 1] != m;
    if (mask[i - W] == m) {
      const bool up_start = (x == 0) || mask[i - W - 1] != m;
      if (my_start || up_start) uf_union(parent, i, i - W);
    } else if (m) {
      // N is background: diagonal neighbours are only reachable from a run end
      if (x + 1 < W && mask[i - W + 1] && !mask[i + 1]) uf_union(parent, i, i - W + 1);
      if (x > 0 && mask[i - W - 1] && my_start) uf_union(parent, i, i - W - 1);
    }
  };
  if ((W & 3) == 0) {
    const int q = blockIdx.x * 256 + threadIdx.x;   // group of four pixels
    if (q >= H * (W >> 2)) return;
    const int y = q / (W >> 2), x = (q - y * (W >> 2)) * 4;
    if (y == 0) return;
    const int i = y * W + x;
    const unsigned cur = *reinterpret_cast<const unsigned*>(mask + i), up = *reinterpret_cast<const unsigned*>(mask + i - W);
    const unsigned rep = (cur & 0xffu) * 0x01010101u;
    // uniform stretch: the four pixels, the four above them and both left neighbours hold one value -> no run starts here, and
    // with N equal to the pixel the diagonal rules do not apply
    if (cur == rep && up == rep && (x == 0 ? false : (mask[i - 1] == (u8)(cur & 0xff) && mask[i - W - 1] == (u8)(cur & 0xff)))) return;
#pragma unroll
    for (int k = 0; k < 4; k++) link1(i + k, x + k);
    return;
  }
  int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= H * W) return;
  int y = i / W, x = i % W;
  if (y == 0) return;
  link1(i, x);
}
// step 3, per RUN (round 5; until then per pixel: every pixel found its root through a segmented scan over the wave, ~270 VALU
// instructions per pixel of a 29.5 M-pixel step, 0.25 ms).  Everything this pass produces is a property of runs:
//   * after k_ccl_link only run heads carry links (uf_union works on roots, and a root is a run head), every other pixel still points
//     at its head: the head is flattened here, and a pixel's root is parent[parent[i]] from now on (k_row_extents);
//   * the row range of a foreground component is the range of its runs' rows (its top row has background or the frame above); of a
//     hole: one row above its first and one below its last run (the foreground pixels bordering it);
//   * a background component is "outside" when one of its runs touches the frame: a run that starts at x = 0 or lies on the first /
//     last row is seen at its head, a run that ENDS at x = W - 1 by the thread of that pixel.
// Only run heads (and the last column's background pixels) do any work.
// Only run heads (and the last column's background pixels) do any work; a thread takes four pixels when the row length allows and
// leaves at once when none of them starts a run (one 4-byte load of the mask).
__global__ __launch_bounds__(256) void k_ccl_stats(const DbPage* __restrict__ pages) {
  const DbPage pg = pages[blockIdx.y];
  const int H = pg.H, W = pg.W;
  const DbWs& ws = pg.ws;
  auto px = [&](int i, int y, int x) {
    const u8 m = ws.mask[i];
    const bool head = x == 0 || ws.mask[i - 1] != m;
    if (!head) {
      if (x == W - 1 && !m) ws.outside[uf_find(ws.parent, ws.parent[i])] = 1;
      return;
    }
    const int r = uf_find(ws.parent, i);
    ws.parent[i] = r;   // flatten the head (only this thread writes the slot; concurrent finds through it tolerate either value)
    if (m) {
      atomicMin(&ws.ymin[r], y);
      atomicMax(&ws.ymax[r], y);
    } else {
      atomicMin(&ws.ymin[r], y - 1);
      atomicMax(&ws.ymax[r], y + 1);
      if (x == 0 || y == 0 || y == H - 1 || (x == W - 1)) ws.outside[r] = 1;
    }
  };
  if ((W & 3) == 0) {
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (q >= H * (W >> 2)) return;
    const int y = q / (W >> 2), x = (q - y * (W >> 2)) * 4, i = y * W + x;
    const unsigned cur = *reinterpret_cast<const unsigned*>(ws.mask + i);
    const unsigned rep = (cur & 0xffu) * 0x01010101u;
    if (cur == rep && x > 0 && ws.mask[i - 1] == (u8)(cur & 0xff) && !(x + 4 == W && !(cur & 0xff))) return;   // no head, not a background run end at the frame
#pragma unroll
    for (int k = 0; k < 4; k++) px(i + k, y, x + k);
    return;
  }
  int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= H * W) return;
  const int y = i / W;
  px(i, y, i - y * W);
}
// one contour per fg component (outer border) and per hole (bg component not touching
// the frame); discovery key = raster index of the border-following start pixel
// (a root is the first pixel of a run: four pixels per thread, groups without a run start leave after one load of the mask)
__global__ __launch_bounds__(256) void k_contour_alloc(const DbPage* __restrict__ pages) {
  const DbPage pg = pages[blockIdx.y];
  const int H = pg.H, W = pg.W;
  const DbWs& ws = pg.ws;
  auto px = [&](int i) {
    if (ws.parent[i] != i) return;
    int type;
    if (ws.mask[i]) type = 0;
    else { if (ws.outside[i]) return; type = 1; }
    int y0 = ws.ymin[i], y1 = ws.ymax[i];
    if (y1 < y0) return;
    int rows = y1 - y0 + 1;
    int idx = atomicAdd(&ws.counters[0], 1);
    if (idx >= ws.contour_cap) { ws.counters[4] = 1; return; }
    int base = atomicAdd(&ws.counters[1], rows);
    int hbase = atomicAdd(&ws.counters[2], 2 * rows + 4);
    if (base + rows > ws.row_cap || hbase + 2 * rows + 4 > ws.hull_cap) { ws.counters[4] = 1; return; }
    for (int r = 0; r < rows; r++) { ws.rowmin[base + r] = 0x7fffffff; ws.rowmax[base + r] = -1; }
    Contour c; c.root = i; c.type = type; c.ymin = y0; c.rows = rows; c.base = base; c.hbase = hbase;
    ws.contours[idx] = c;
    ws.cidx[i] = idx;
  };
  if ((W & 3) == 0) {
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (q >= H * (W >> 2)) return;
    const int y = q / (W >> 2), x = (q - y * (W >> 2)) * 4, i = y * W + x;
    const unsigned cur = *reinterpret_cast<const unsigned*>(ws.mask + i);
    const unsigned rep = (cur & 0xffu) * 0x01010101u;
    if (cur == rep && x > 0 && ws.mask[i - 1] == (u8)(cur & 0xff)) return;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const u8 m = (u8)(cur >> (8 * k));
      const bool head = x + k == 0 || (k == 0 ? ws.mask[i - 1] : (u8)(cur >> (8 * (k - 1)))) != m;
      if (head) px(i + k);
    }
    return;
  }
  int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= H * W) return;
  px(i);
}
// (only pixels at the edge of a run, or background pixels next to foreground, contribute: four pixels per thread, groups
//  whose pixels and all their 4-neighbours hold one value leave after three 4-byte loads)
__global__ __launch_bounds__(256) void k_row_extents(const DbPage* __restrict__ pages) {
  const DbPage pg = pages[blockIdx.y];
  const int H = pg.H, W = pg.W;
  const DbWs& ws = pg.ws;
  auto px = [&](int i, int y, int x) {
    int r = ws.parent[ws.parent[i]];   // pixel -> head of its run -> root (k_ccl_stats flattened the heads only)
    int ci = ws.cidx[r];
    if (ci < 0) return;  // outside background (or an overflowed list)
    const Contour c = ws.contours[ci];
    if (ws.mask[i]) {
      int row = c.base + (y - c.ymin);
      if (x == 0 || !ws.mask[i - 1]) atomicMin(&ws.rowmin[row], x);
      if (x == W - 1 || !ws.mask[i + 1]) atomicMax(&ws.rowmax[row], x);
    } else {
      if (x > 0 && ws.mask[i - 1]) { int row = c.base + (y - c.ymin); atomicMin(&ws.rowmin[row], x - 1); atomicMax(&ws.rowmax[row], x - 1); }
      if (x + 1 < W && ws.mask[i + 1]) { int row = c.base + (y - c.ymin); atomicMin(&ws.rowmin[row], x + 1); atomicMax(&ws.rowmax[row], x + 1); }
      if (y > 0 && ws.mask[i - W]) { int row = c.base + (y - 1 - c.ymin); atomicMin(&ws.rowmin[row], x); atomicMax(&ws.rowmax[row], x); }
      if (y + 1 < H && ws.mask[i + W]) { int row = c.base + (y + 1 - c.ymin); atomicMin(&ws.rowmin[row], x); atomicMax(&ws.rowmax[row], x); }
    }
  };
  if ((W & 3) == 0) {
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (q >= H * (W >> 2)) return;
    const int y = q / (W >> 2), x = (q - y * (W >> 2)) * 4, i = y * W + x;
    const unsigned cur = *reinterpret_cast<const unsigned*>(ws.mask + i);
    const unsigned rep = (cur & 0xffu) * 0x01010101u;
    const u8 m = (u8)(cur & 0xff);
    // interior of a uniform region: foreground pixels that are no run ends, or background pixels without a foreground neighbour.
    // (At the frame: a foreground pixel in the first / last column IS a run end; missing rows above / below count as "same".)
    if (cur == rep && (x > 0 ? ws.mask[i - 1] == m : !m) && (x + 4 < W ? ws.mask[i + 4] == m : !m) &&
        (y == 0 || *reinterpret_cast<const unsigned*>(ws.mask + i - W) == rep || m) &&
        (y + 1 >= H || *reinterpret_cast<const unsigned*>(ws.mask + i + W) == rep || m))
      return;
#pragma unroll
    for (int k = 0; k < 4; k++) px(i + k, y, x + k);
    return;
  }
  int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= H * W) return;
  const int y = i / W;
  px(i, y, i - y * W);
}

// ---- per-contour geometry ----------------------------------------------------------
struct DP { double x, y; };

// imageproc geometry::rotating_calipers on a hull of >= 3 points.
// get(i) returns hull point i as doubles.  out: 4 corners TL,TR,BR,BL, floor()ed.
// One wavefront runs this (every lane with the same n / get): the edges are dealt over the 64 lanes -- each lane does the
// atan2 / sin / cos and the projection loop of ITS edge, which is where the time goes (serially ~8 k cycles per edge, 35
// edges on a text line's hull) -- and the sequential "first edge with the smallest area" is recovered by an argmin over
// (area, edge index).  Vec::dedup drops an angle equal to its predecessor's; an element of a run of equal angles differs
// from the last KEPT one exactly when it differs from its immediate predecessor, so each lane decides that from edge e - 1
// alone.  Same f64 operations per edge as the sequential form: bit-identical.
__device__ __forceinline__ double shfl_f64(double v, int src_lane) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __shfl(lo, src_lane); hi = __shfl(hi, src_lane);
  return __hiloint2double(hi, lo);
}
template <typename Get>
__device__ void rotating_calipers(int n, Get get, double* out8) {
  const double PI = 3.14159265358979323846264338327950288;
  const double BIG = 1.7976931348623157e308;
  const int lane = threadIdx.x & 63;
  double min_area = BIG;
  DP res[4] = {{0, 0}, {0, 0}, {0, 0}, {0, 0}};
  auto edge_angle = [&](int e) {
    DP a = get(e), b = get(e + 1);
    return fabs(fmod(atan2(b.y - a.y, b.x - a.x) + PI, PI / 2.0));
  };
  for (int e0 = 0; e0 + 1 < n; e0 += 64) {  // points.windows(2): the closing edge is not visited
    const int e = e0 + lane;
    double area = BIG;
    DP r4[4] = {{0, 0}, {0, 0}, {0, 0}, {0, 0}};
    if (e + 1 < n) {
      const double angle = edge_angle(e);
      const bool kept = e == 0 || !(angle == edge_angle(e - 1));  // Vec::dedup
      if (kept) {
        double s = sin(angle), c = cos(angle);
        double min_x = BIG, max_x = -BIG, min_y = BIG, max_y = -BIG;
        for (int i = 0; i < n; i++) {
          DP p = get(i);
          double rx = p.x * c + p.y * s;
          double ry = p.y * c - p.x * s;
          min_x = fmin(min_x, rx); max_x = fmax(max_x, rx);
          min_y = fmin(min_y, ry); max_y = fmax(max_y, ry);
        }
        area = (max_x - min_x) * (max_y - min_y);
        r4[0] = DP{max_x * c - min_y * s, min_y * c + max_x * s};
        r4[1] = DP{min_x * c - min_y * s, min_y * c + min_x * s};
        r4[2] = DP{min_x * c - max_y * s, max_y * c + min_x * s};
        r4[3] = DP{max_x * c - max_y * s, max_y * c + max_x * s};
      }
    }
    // lowest lane among those with the smallest area that is < everything seen so far (NaN areas never win: `<` is false)
    const bool cand = area < min_area;
    double barea = cand ? area : BIG;
    int blane = cand ? lane : 64;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const double oa = shfl_f64(barea, (lane ^ d));
      const int ol = __shfl(blane, lane ^ d);
      if (ol < 64 && (blane >= 64 || oa < barea || (oa == barea && ol < blane))) { barea = oa; blane = ol; }
    }
    if (blane < 64) {
      min_area = barea;
#pragma unroll
      for (int k = 0; k < 4; k++) { res[k].x = shfl_f64(r4[k].x, blane); res[k].y = shfl_f64(r4[k].y, blane); }
    }
  }
  // stable sort of 4 by x (insertion sort)
  for (int i = 1; i < 4; i++) {
    DP k = res[i]; int j = i - 1;
    while (j >= 0 && res[j].x > k.x) { res[j + 1] = res[j]; j--; }
    res[j + 1] = k;
  }
  int i1 = res[1].y > res[0].y ? 0 : 1;
  int i2 = res[3].y > res[2].y ? 2 : 3;
  int i3 = res[3].y > res[2].y ? 3 : 2;
  int i4 = res[1].y > res[0].y ? 1 : 0;
  int idx[4] = {i1, i2, i3, i4};
  for (int i = 0; i < 4; i++) { out8[2 * i] = floor(res[idx[i]].x); out8[2 * i + 1] = floor(res[idx[i]].y); }
}

template <typename Get>
__device__ void min_area_rect_hull(int n, Get get, double* out8) {
  if (n == 1) { DP p = get(0); for (int i = 0; i < 4; i++) { out8[2 * i] = p.x; out8[2 * i + 1] = p.y; } return; }
  if (n == 2) {
    DP a = get(0), b = get(1);
    out8[0] = a.x; out8[1] = a.y; out8[2] = b.x; out8[3] = b.y; out8[4] = b.x; out8[5] = b.y; out8[6] = a.x; out8[7] = a.y;
    return;
  }
  rotating_calipers(n, get, out8);
}

__device__ __forceinline__ long long orient_ll(int2 p, int2 q, int2 r) {
  return (long long)(q.y - p.y) * (long long)(r.x - q.x) - (long long)(q.x - p.x) * (long long)(r.y - q.y);
}

// Strict convex hull of a contour from its per-row extents, in imageproc's order: start
// at the top-most then left-most point, then along increasing x (screen-clockwise).
__device__ int hull_from_rows(const int* rmin, const int* rmax, int rows, int ymin, int2* hs) {
  int top = 0; while (top < rows && rmax[top] < 0) top++;
  int bot = rows - 1; while (bot >= 0 && rmax[bot] < 0) bot--;
  if (top > bot) return 0;
  int n = 0;
  // right chain
  hs[n++] = make_int2(rmin[top], ymin + top);
  for (int r = top; r <= bot; r++) {
    if (rmax[r] < 0) continue;
    int2 p = make_int2(rmax[r], ymin + r);
    if (hs[n - 1].x == p.x && hs[n - 1].y == p.y) continue;
    while (n >= 2 && orient_ll(hs[n - 2], hs[n - 1], p) >= 0) n--;
    hs[n++] = p;
  }
  // left chain (back to the start point)
  int t = n;  // keep hs[t-1] (bottom-right) as the chain base
  for (int r = bot; r >= top; r--) {
    if (rmax[r] < 0) continue;
    int2 p = make_int2(rmin[r], ymin + r);
    if (hs[n - 1].x == p.x && hs[n - 1].y == p.y) continue;
    while (n > t && orient_ll(hs[n - 2], hs[n - 1], p) >= 0) n--;
    hs[n++] = p;
  }
  if (n > 1 && hs[n - 1].x == hs[0].x && hs[n - 1].y == hs[0].y) n--;
  return n;
}

// imageproc draw_polygon_mut + BresenhamLineIter coverage of one canvas row, as a list of
// x intervals.  poly: 4 points relative to the canvas origin.
struct RowCover { int lo[8], hi[8]; int n; };
__device__ __forceinline__ long long ceil_div_ll(long long a, long long b) { return a >= 0 ? (a + b - 1) / b : -((-a) / b); }
__device__ __forceinline__ long long floor_div_ll(long long a, long long b) { return a >= 0 ? a / b : -((-a + b - 1) / b); }

__device__ void row_cover(const int2* poly, int bw, int bh, int py_min, int py_max, int y, RowCover& rc) {
  rc.n = 0;
  // scan-line fill
  if (y >= py_min && y <= py_max) {
    int inter[8]; int ni = 0;
    for (int e = 0; e < 4; e++) {
      int2 p0 = poly[e], p1 = poly[(e + 1) & 3];
      if ((p0.y <= y && p1.y >= y) || (p1.y <= y && p0.y >= y)) {
        if (p0.y == p1.y) { inter[ni++] = p0.x; inter[ni++] = p1.x; }
        else if (p0.y == y || p1.y == y) {
          if (p1.y > y) inter[ni++] = p0.x;
          if (p0.y > y) inter[ni++] = p1.x;
        } else {
          float fraction = (float)(y - p0.y) / (float)(p1.y - p0.y);
          float in = (float)p0.x + fraction * (float)(p1.x - p0.x);
          inter[ni++] = gm::f32_as_i32(roundf(in));
        }
      }
    }
    for (int i = 1; i < ni; i++) { int k = inter[i], j = i - 1; while (j >= 0 && inter[j] > k) { inter[j + 1] = inter[j]; j--; } inter[j + 1] = k; }
    for (int k = 0; k + 1 < ni; k += 2) {
      int from = min(inter[k], bw), to = min(inter[k + 1], bw - 1);
      if (from < bw && to >= 0) {
        from = max(0, from); to = max(0, to);
        if (from <= to) { rc.lo[rc.n] = from; rc.hi[rc.n] = to; rc.n++; }
      }
    }
  }
  // polygon outline: Bresenham pixels of each edge that fall in this row
  for (int e = 0; e < 4; e++) {
    int x0 = poly[e].x, y0 = poly[e].y, x1 = poly[(e + 1) & 3].x, y1 = poly[(e + 1) & 3].y;
    bool steep = abs(y1 - y0) > abs(x1 - x0);
    if (steep) { int t = x0; x0 = y0; y0 = t; t = x1; x1 = y1; y1 = t; }
    if (x0 > x1) { int t = x0; x0 = x1; x1 = t; t = y0; y0 = y1; y1 = t; }
    long long dx = x1 - x0, dy = abs(y1 - y0);
    int ystep = y0 < y1 ? 1 : -1;
    // after i steps the minor coordinate has advanced k_i = ceil((2 i dy - dx) / (2 dx)) times
    if (steep) {
      long long i = (long long)y - x0;  // major axis is the canvas y
      if (i < 0 || i > dx) continue;
      long long k = dx > 0 ? ceil_div_ll(2 * i * dy - dx, 2 * dx) : 0;
      int x = y0 + ystep * (int)k;
      if (x >= 0 && x < bw) { rc.lo[rc.n] = x; rc.hi[rc.n] = x; rc.n++; }
    } else {
      long long kk = (long long)(y - y0) * ystep;
      if (kk < 0) continue;
      long long ilo, ihi;
      if (dy == 0) { if (kk != 0) continue; ilo = 0; ihi = dx; }
      else {
        ilo = kk == 0 ? 0 : floor_div_ll(2 * dx * kk - dx, 2 * dy) + 1;
        ihi = floor_div_ll(2 * dx * kk + dx, 2 * dy);
        if (ilo < 0) ilo = 0;
        if (ihi > dx) ihi = dx;
      }
      if (ilo > ihi) continue;
      int lo = (int)(x0 + ilo), hi = (int)(x0 + ihi);
      lo = max(lo, 0); hi = min(hi, bw - 1);
      if (lo <= hi) { rc.lo[rc.n] = lo; rc.hi[rc.n] = hi; rc.n++; }
    }
  }
}

// det_processor.rs:188-221 box_score_fast (sequential f32 accumulation, row-major).
__device__ float box_score_fast(const float* pred, int H, int W, const int* box) {
  int x_min = 0x7fffffff, x_max = -0x7fffffff - 1, y_min = 0x7fffffff, y_max = -0x7fffffff - 1;
  for (int i = 0; i < 4; i++) {
    x_min = min(x_min, box[2 * i]); x_max = max(x_max, box[2 * i]);
    y_min = min(y_min, box[2 * i + 1]); y_max = max(y_max, box[2 * i + 1]);
  }
  x_min = min(max(x_min, 0), W - 1); x_max = min(max(x_max, 0), W - 1);
  y_min = min(max(y_min, 0), H - 1); y_max = min(max(y_max, 0), H - 1);
  int bw = x_max - x_min + 1, bh = y_max - y_min + 1;
  int2 poly[4];
  for (int i = 0; i < 4; i++) poly[i] = make_int2(box[2 * i] - x_min, box[2 * i + 1] - y_min);
  if (poly[0].x == poly[3].x && poly[0].y == poly[3].y) return 0.0f;  // draw_polygon_mut would panic (A.4)
  int py_min = 0x7fffffff, py_max = -0x7fffffff - 1;
  for (int i = 0; i < 4; i++) { py_min = min(py_min, poly[i].y); py_max = max(py_max, poly[i].y); }
  py_min = max(0, min(py_min, bh - 1)); py_max = max(0, min(py_max, bh - 1));
  // One wavefront per contour: every lane runs the (wave-uniform) geometry redundantly; here
  // the 64 lanes fetch and mask-test 64 consecutive pixels of a row at once, and the masked
  // values are then folded into `sum` one lane at a time in pixel order, which reproduces the
  // reference's sequential f32 accumulation bit for bit.
  const int lane = threadIdx.x & 63;
  float sum = 0.0f; unsigned long long count = 0;
  RowCover rc;
  for (int y = 0; y < bh; y++) {
    row_cover(poly, bw, bh, py_min, py_max, y, rc);
    if (rc.n == 0) continue;
    int lo = rc.lo[0], hi = rc.hi[0];
    for (int k = 1; k < rc.n; k++) { lo = min(lo, rc.lo[k]); hi = max(hi, rc.hi[k]); }
    const float* prow = pred + (size_t)(y + y_min) * W + x_min;
    for (int x0 = lo; x0 <= hi; x0 += 64) {
      const int x = x0 + lane;
      bool in = false;
      if (x <= hi)
        for (int k = 0; k < rc.n; k++) in = in || (x >= rc.lo[k] && x <= rc.hi[k]);
      const float v = in ? prow[x] : 0.0f;
      const unsigned long long m = __ballot(in);
      if (m == 0) continue;   // (wave-uniform)
      count += (unsigned long long)__popcll(m);
      // Sequential fold in pixel order, one v_readlane + one add per lane.  Masked lanes contribute +0.0f, which leaves
      // a sum that is never -0.0 (it starts at +0.0 and probabilities are >= 0) bit for bit unchanged -- so the loop needs
      // no per-pixel bit scan (the earlier ffs / shuffle / mask-update loop was ~50 cycles per pixel, most of this
      // kernel's 320 us; this form is ~8).
      const int vb = __float_as_int(v);
#pragma unroll
      for (int l = 0; l < 64; l++) sum = sum + __int_as_float(__builtin_amdgcn_readlane(vb, l));
    }
  }
  return count > 0 ? sum / (float)count : 0.0f;
}

// Clipper 6.4.2 ClipperOffset (jtRound, etClosedPolygon, arc tolerance 0.5) of the
// 4-point box, det_processor.rs:223-252.  Returns the number of
// offset vertices written to (ox, oy) (closing duplicate NOT appended: min_area_rect
// only looks at the hull).  0 = no polygon.
#define RT_MAX_OFFSET_PTS 384
__device__ __forceinline__ long long cl_round(double v) { return v < 0 ? (long long)(v - 0.5) : (long long)(v + 0.5); }

__device__ int unclip_box(const int* box, float unclip_ratio, float* ox, float* oy, int* overflow) {
  float cx[5], cy[5];
  for (int i = 0; i < 4; i++) { cx[i] = (float)box[2 * i]; cy[i] = (float)box[2 * i + 1]; }
  cx[4] = cx[0]; cy[4] = cy[0];
  float sx = cx[0], sy = cy[0], tmp = 0.0f;
  for (int i = 0; i < 4; i++) {
    float ax = cx[i] - sx, ay = cy[i] - sy, bx = cx[i + 1] - sx, by = cy[i + 1] - sy;
    tmp += ax * by - bx * ay;
  }
  float area = fabsf(tmp / 2.0f);
  float perimeter = 0.0f;
  for (int i = 0; i < 4; i++) {
    float dx = cx[i] - cx[i + 1], dy = cy[i] - cy[i + 1];
    perimeter += (float)sqrt((double)dx * (double)dx + (double)dy * (double)dy);
  }
  perimeter = perimeter + 0.0f;
  float distance = area * unclip_ratio / perimeter;
  double delta = (double)(distance * 1.0f);
  // ClipperOffset::AddPath: strip closing / consecutive duplicates
  long long sxp[4], syp[4]; int len = 0;
  {
    long long px[5], py[5];
    for (int i = 0; i < 5; i++) { px[i] = (long long)(cx[i] * 1.0f); py[i] = (long long)(cy[i] * 1.0f); }
    int highI = 4;
    while (highI > 0 && px[0] == px[highI] && py[0] == py[highI]) highI--;
    sxp[0] = px[0]; syp[0] = py[0]; len = 1;
    for (int i = 1; i <= highI; i++)
      if (sxp[len - 1] != px[i] || syp[len - 1] != py[i]) { sxp[len] = px[i]; syp[len] = py[i]; len++; }
  }
  if (len < 3) return 0;
  // FixOrientations
  {
    double a = 0;
    for (int i = 0, j = len - 1; i < len; ++i) { a += ((double)sxp[j] + (double)sxp[i]) * ((double)syp[j] - (double)syp[i]); j = i; }
    double ar = -a * 0.5;
    if (!(ar >= 0)) {
      for (int i = 0; i < len / 2; i++) {
        long long t = sxp[i]; sxp[i] = sxp[len - 1 - i]; sxp[len - 1 - i] = t;
        t = syp[i]; syp[i] = syp[len - 1 - i]; syp[len - 1 - i] = t;
      }
    }
  }
  int n = 0;
  auto push = [&](long long X, long long Y) {
    if (n < RT_MAX_OFFSET_PTS) { ox[n] = (float)((double)X / 1.0); oy[n] = (float)((double)Y / 1.0); n++; }
    else *overflow = 1;
  };
  const double pi = 3.141592653589793238, two_pi = pi * 2, def_arc_tolerance = 0.25, arc_tolerance = 0.5;
  if (fabs(delta) < 1.0E-20) { for (int i = 0; i < len; i++) push(sxp[i], syp[i]); return n; }
  double yv;
  if (arc_tolerance > fabs(delta) * def_arc_tolerance) yv = fabs(delta) * def_arc_tolerance;
  else yv = arc_tolerance;
  double steps = pi / acos(1 - yv / fabs(delta));
  if (steps > fabs(delta) * pi) steps = fabs(delta) * pi;
  double m_sin = sin(two_pi / steps), m_cos = cos(two_pi / steps);
  double steps_per_rad = steps / two_pi;
  if (delta < 0.0) m_sin = -m_sin;
  double nx[4], ny[4];
  for (int j = 0; j < len; j++) {
    int j2 = (j + 1 == len) ? 0 : j + 1;
    if (sxp[j2] == sxp[j] && syp[j2] == syp[j]) { nx[j] = 0; ny[j] = 0; continue; }
    double Dx = (double)(sxp[j2] - sxp[j]), Dy = (double)(syp[j2] - syp[j]);
    double f = 1 * 1.0 / sqrt(Dx * Dx + Dy * Dy);
    Dx *= f; Dy *= f;
    nx[j] = Dy; ny[j] = -Dx;
  }
  int k = len - 1;
  for (int j = 0; j < len; ++j) {
    double sinA = nx[k] * ny[j] - nx[j] * ny[k];
    bool done = false;
    if (fabs(sinA * delta) < 1.0) {
      double cosA = nx[k] * nx[j] + ny[j] * ny[k];
      if (cosA > 0) { push(cl_round(sxp[j] + nx[k] * delta), cl_round(syp[j] + ny[k] * delta)); done = true; }
    } else if (sinA > 1.0) sinA = 1.0;
    else if (sinA < -1.0) sinA = -1.0;
    if (!done) {
      if (sinA * delta < 0) {
        push(cl_round(sxp[j] + nx[k] * delta), cl_round(syp[j] + ny[k] * delta));
        push(sxp[j], syp[j]);
        push(cl_round(sxp[j] + nx[j] * delta), cl_round(syp[j] + ny[j] * delta));
      } else {
        double a = atan2(sinA, nx[k] * nx[j] + ny[k] * ny[j]);
        long long rs = cl_round(steps_per_rad * fabs(a));
        int nsteps = (int)(rs > 1 ? rs : 1);
        double X = nx[k], Y = ny[k], X2;
        for (int i = 0; i < nsteps; ++i) {
          push(cl_round(sxp[j] + X * delta), cl_round(syp[j] + Y * delta));
          X2 = X;
          X = X * m_cos - m_sin * Y;
          Y = X2 * m_sin + Y * m_cos;
        }
        push(cl_round(sxp[j] + nx[j] * delta), cl_round(syp[j] + ny[j] * delta));
      }
    }
    k = j;
  }
  if (n < 3) return 0;
  return n;
}

// strict convex hull (imageproc order) of <= RT_MAX_OFFSET_PTS integral-valued f32 points
// by gift wrapping; writes hull indices to hidx.
__device__ int hull_jarvis(const float* px, const float* py, int n, short* hidx) {
  int s = 0;
  for (int i = 1; i < n; i++) if (py[i] < py[s] || (py[i] == py[s] && px[i] < px[s])) s = i;
  int h = 0, p = s;
  while (true) {
    hidx[h++] = (short)p;
    int q = -1;
    for (int r = 0; r < n; r++) {
      if (px[r] == px[p] && py[r] == py[p]) continue;
      if (q < 0) { q = r; continue; }
      double val = ((double)py[q] - (double)py[p]) * ((double)px[r] - (double)px[q]) -
                   ((double)px[q] - (double)px[p]) * ((double)py[r] - (double)py[q]);
      if (val > 0.0) q = r;  // r lies on the clockwise side: it comes first
      else if (val == 0.0) {
        double dq = ((double)px[q] - px[p]) * ((double)px[q] - px[p]) + ((double)py[q] - py[p]) * ((double)py[q] - py[p]);
        double dr = ((double)px[r] - px[p]) * ((double)px[r] - px[p]) + ((double)py[r] - py[p]) * ((double)py[r] - py[p]);
        // collinear: keep the farther one if it lies in the same direction
        double dot = ((double)px[q] - px[p]) * ((double)px[r] - px[p]) + ((double)py[q] - py[p]) * ((double)py[r] - py[p]);
        if (dot > 0.0 && dr > dq) q = r;
        else if (dot < 0.0) {
          // opposite directions from p: p is interior to segment (q, r); the hull edge leaving p
          // in traversal order is the one keeping every other point on the CCW side - decided
          // by the remaining points, so leave q unchanged here.
        }
      }
    }
    if (q < 0) break;  // all points coincide
    if (px[q] == px[s] && py[q] == py[s]) break;
    if (h >= n) break;
    p = q;
  }
  return h;
}

__device__ __forceinline__ float euclid_f32(float ax, float ay, float bx, float by) {
  float dx = ax - bx, dy = ay - by;
  return sqrtf(dx * dx + dy * dy);
}


#define RT_CONTOUR_WAVES 128
__global__ __launch_bounds__(64) void k_contour_boxes(const DbPage* __restrict__ pages, DbParams prm) {
  const DbPage pg = pages[blockIdx.y];
  const DbWs& ws = pg.ws;
  const int H = pg.H, W = pg.W;
  const float* pred = pg.pred;
  const int ncont = min(ws.counters[0], ws.contour_cap);
  for (int ci = blockIdx.x; ci < ncont; ci += RT_CONTOUR_WAVES) {
  const Contour ct = ws.contours[ci];
  int2* hs = ws.hull + ct.hbase;
  int hn = hull_from_rows(ws.rowmin + ct.base, ws.rowmax + ct.base, ct.rows, ct.ymin, hs);
  if (hn == 0) continue;
  double r[8];
  min_area_rect_hull(hn, [&](int i) { return DP{(double)hs[i].x, (double)hs[i].y}; }, r);
  int box[8];
  for (int i = 0; i < 8; i++) box[i] = (int)r[i];
  float s1 = euclid_f32((float)box[0], (float)box[1], (float)box[2], (float)box[3]);
  float s2 = euclid_f32((float)box[6], (float)box[7], (float)box[4], (float)box[5]);
  float sside = fminf(s1, s2);
  if (sside < (float)prm.min_size) continue;
  float mean_score = box_score_fast(pred, H, W, box);
  if (mean_score < prm.box_thresh) continue;
  float ox[RT_MAX_OFFSET_PTS], oy[RT_MAX_OFFSET_PTS];
  short hidx[RT_MAX_OFFSET_PTS];
  int ovf = 0;
  int on = unclip_box(box, prm.unclip_ratio, ox, oy, &ovf);
  if (ovf) { ws.counters[4] = 1; continue; }
  if (on == 0) continue;
  int h2 = hull_jarvis(ox, oy, on, hidx);
  double r2[8];
  min_area_rect_hull(h2, [&](int i) { return DP{(double)ox[hidx[i]], (double)oy[hidx[i]]}; }, r2);
  DbBox b;
  for (int i = 0; i < 8; i++) b.pts[i] = (float)r2[i];
  float t1 = euclid_f32(b.pts[0], b.pts[1], b.pts[2], b.pts[3]);
  float t2 = euclid_f32(b.pts[6], b.pts[7], b.pts[4], b.pts[5]);
  if (fminf(t1, t2) < (float)(prm.min_size + 2)) continue;
  gm::scale_and_clip(b.pts, (double)W, (double)H, (double)pg.ori_w, (double)pg.ori_h);
  float pb_h = gm::side_len(&b.pts[0], &b.pts[6]);
  float pb_w = gm::side_len(&b.pts[0], &b.pts[2]);
  if (pb_h <= 3.0f || pb_w <= 3.0f) continue;
  b.score = mean_score;
  b.key = ct.type == 0 ? ct.root : ct.root - 1;
  if ((threadIdx.x & 63) == 0) {
    int slot = atomicAdd(&ws.counters[3], 1);
    if (slot >= ws.cand_cap) ws.counters[4] = 1;
    else ws.cand[slot] = b;
  }
  }
}

// det_processor.rs:324-333.  Stable bottom-up merge sort (run width 1, 2, 4, ...; take
// from the left run unless right < left) applied to the boxes in contour discovery order.
// For the strict weak orders of the contract every stable sort gives this result; for the
// comparator's non-transitive inputs this exact algorithm is the defined
// behaviour and the oracle uses the same one.  Sorting works on indices + centres in LDS.
// Parallel form: (1) the discovery order is the order of the (unique) keys -- rank by counting for pages whose
// candidates fit in LDS, through a bitmap of the key pixels + a prefix sum of its population counts otherwise (the
// O(n^2) count would take seconds at 50 k candidates); (2) every merge pass runs its pairs of runs on different threads,
// each pair merged sequentially by one thread exactly as above -- merge-path splitting of a pair is NOT used: its binary
// search assumes a transitive comparator, and on non-transitive inputs it can take another path than the sequential
// merge that defines the result.  The last passes have fewer pairs than threads (2 * width sequential steps each):
// O(n) steps in all instead of O(n log n) on one thread.  Pages of up to RT_SORT_LDS candidates sort in LDS, larger
// ones in the page's workspace (the reference's list is unbounded, det_processor.rs:279-335: no fixed cap here either,
// only the configured max_boxes_per_page).
#define RT_SORT_LDS 3072
__global__ __launch_bounds__(256) void k_sort_boxes(const DbPage* __restrict__ pages) {
  const DbPage pg = pages[blockIdx.x];
  const DbWs& ws = pg.ws;
  DbBox* out = pg.boxes_out;
  __shared__ float l_cx[RT_SORT_LDS], l_cy[RT_SORT_LDS];
  __shared__ unsigned l_a[RT_SORT_LDS], l_b[RT_SORT_LDS];
  __shared__ int part[256];
  const int n = min(ws.counters[3], ws.cand_cap);
  const bool small = n <= RT_SORT_LDS;
  float* cx = small ? l_cx : ws.sort_cx;
  float* cy = small ? l_cy : ws.sort_cy;
  unsigned* ia = small ? l_a : ws.sort_a;
  unsigned* ib = small ? l_b : ws.sort_b;
  const int tid = threadIdx.x;
  if (small) {
    int* keys = reinterpret_cast<int*>(l_b);   // (free until the first merge pass)
    for (int i = tid; i < n; i += 256) keys[i] = ws.cand[i].key;
    __syncthreads();
    for (int i = tid; i < n; i += 256) {
      const int key = keys[i];
      int rank = 0;
      for (int j = 0; j < n; j++) rank += keys[j] < key;
      ia[rank] = (unsigned)i;
    }
  } else {
    // keys are pixel indices (unique per contour): bitmap in ws.parent, per-word exclusive prefix of the set bits in ws.ymin
    const int N = pg.H * pg.W, nw = (N + 32) / 32 + 1;   // (hole contours carry root - 1 >= -1: keys are shifted by one)
    unsigned* bits = reinterpret_cast<unsigned*>(ws.parent);
    int* pref = ws.ymin;
    for (int w = tid; w < nw; w += 256) bits[w] = 0u;
    __syncthreads();
    for (int i = tid; i < n; i += 256) { const int k = ws.cand[i].key + 1; atomicOr(&bits[k >> 5], 1u << (k & 31)); }
    __syncthreads();
    const int per = (nw + 255) / 256, w0 = tid * per, w1 = min(nw, w0 + per);
    int local = 0;
    for (int w = w0; w < w1; w++) { pref[w] = local; local += __popc(bits[w]); }
    part[tid] = local;
    __syncthreads();
    if (tid == 0) { int run = 0; for (int t = 0; t < 256; t++) { const int v = part[t]; part[t] = run; run += v; } }
    __syncthreads();
    for (int w = w0; w < w1; w++) pref[w] += part[tid];
    __syncthreads();
    for (int i = tid; i < n; i += 256) {
      const int k = ws.cand[i].key + 1;
      ia[pref[k >> 5] + __popc(bits[k >> 5] & ((1u << (k & 31)) - 1u))] = (unsigned)i;
    }
  }
  for (int i = tid; i < n; i += 256) {
    cx[i] = (ws.cand[i].pts[0] + ws.cand[i].pts[4]) / 2.0f;
    cy[i] = (ws.cand[i].pts[1] + ws.cand[i].pts[5]) / 2.0f;
  }
  __syncthreads();
  auto less = [&](unsigned a, unsigned b) {
    if (fabsf(cy[a] - cy[b]) < 10.0f) return cx[a] < cx[b];
    return cy[a] < cy[b];
  };
  unsigned* src = ia; unsigned* dst = ib;
  for (int width = 1; width < n; width *= 2) {
    const int pairs = (n + 2 * width - 1) / (2 * width);
    for (int pr = tid; pr < pairs; pr += 256) {
      const int lo = pr * 2 * width, mid = min(lo + width, n), hi = min(lo + 2 * width, n);
      int i = lo, j = mid, k = lo;
      while (i < mid && j < hi) {
        if (less(src[j], src[i])) dst[k++] = src[j++];
        else dst[k++] = src[i++];
      }
      while (i < mid) dst[k++] = src[i++];
      while (j < hi) dst[k++] = src[j++];
    }
    __syncthreads();
    unsigned* t = src; src = dst; dst = t;
  }
  if (tid == 0) {
    pg.count_out[0] = n;
    pg.count_out[1] = ws.counters[4];
  }
  for (int i = tid; i < n; i += 256) out[i] = ws.cand[src[i]];
}

void db_postprocess_batch(hipStream_t st, int n, const DbPageIn* in, const DbParams& p, void* const* workspaces,
                          int max_boxes, DbBox* const* boxes_out, int* const* count_out, void* h_desc, void* d_desc) {
  if (n <= 0) return;
  DbPage* hp = (DbPage*)h_desc;
  int maxN = 0, maxH = 0, maxC = 0;
  for (int i = 0; i < n; i++) {
    size_t total;
    hp[i].pred = in[i].pred; hp[i].H = in[i].H; hp[i].W = in[i].W; hp[i].ori_h = in[i].ori_h; hp[i].ori_w = in[i].ori_w;
    hp[i].ws = carve(workspaces[i], in[i].H, in[i].W, max_boxes, &total);
    hp[i].boxes_out = boxes_out[i]; hp[i].count_out = count_out[i];
    maxN = std::max(maxN, in[i].H * in[i].W); maxH = std::max(maxH, in[i].H);
    maxC = std::max(maxC, hp[i].ws.contour_cap);
  }
  (void)hipMemcpyAsync(d_desc, h_desc, (size_t)n * sizeof(DbPage), hipMemcpyHostToDevice, st);
  const DbPage* dp = (const DbPage*)d_desc;
  // the per-pixel passes take four pixels per thread on pages whose width is a multiple of 4 (every det map is: multiples of 32)
  bool all4 = true;
  for (int i = 0; i < n; i++) all4 = all4 && (in[i].W & 3) == 0;
  dim3 grid(((all4 ? maxN / 4 : maxN) + 255) / 256, n), blk(256);
  RT_LAUNCH(k_ccl_rows, dim3(maxH, n), blk, 0, st, dp, p.thresh, p.dilate);
  RT_LAUNCH(k_ccl_link, grid, blk, 0, st, dp);
  RT_LAUNCH(k_ccl_stats, grid, blk, 0, st, dp);
  RT_LAUNCH(k_contour_alloc, grid, blk, 0, st, dp);
  RT_LAUNCH(k_row_extents, grid, blk, 0, st, dp);
  (void)maxC;
  RT_LAUNCH(k_contour_boxes, dim3(RT_CONTOUR_WAVES, n), dim3(64), 0, st, dp, p);
  RT_LAUNCH(k_sort_boxes, dim3(n), dim3(256), 0, st, dp);
}

// Sorted boxes of all pages -> one contiguous list (page order), so that the host fetches them with a single copy.
// boxes: [n][max_boxes], counts: [n][2] as written by k_sort_boxes.  One block per page.
__global__ __launch_bounds__(256) void k_pack_boxes(const DbBox* __restrict__ boxes, const int* __restrict__ counts,
                                                    int max_boxes, DbBox* __restrict__ packed) {
  const int page = blockIdx.x;
  int off = 0;
  for (int q = 0; q < page; q++) off += min(max(counts[2 * q], 0), max_boxes);
  const int cnt = min(max(counts[2 * page], 0), max_boxes);
  constexpr int WORDS = (int)(sizeof(DbBox) / 4);
  const uint32_t* src = reinterpret_cast<const uint32_t*>(boxes + (size_t)page * max_boxes);
  uint32_t* dst = reinterpret_cast<uint32_t*>(packed + off);
  for (int i = threadIdx.x; i < cnt * WORDS; i += blockDim.x) dst[i] = src[i];
}
void pack_boxes(hipStream_t st, int n, const DbBox* boxes, const int* counts, int max_boxes, DbBox* packed) {
  if (n > 0) RT_LAUNCH(k_pack_boxes, dim3(n), dim3(256), 0, st, boxes, counts, max_boxes, packed);
}

}  // namespace pp
}  // namespace rt
