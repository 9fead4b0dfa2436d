// Launchers of the pre/post-processing kernels (prepost_kernels.hip): everything in
// retto-core's DetProcessor / ClsProcessor / RecProcessor / ImageHelper that is not a
// network forward.  Bit-exact integer / f32 / f64 arithmetic (no FMA contraction).
#pragma once
#include "nn.h"
#include "common.h"

namespace rt {
namespace pp {

// image 0.25.6 imageops::thumbnail on RGB8 (used by image_helper.rs:124,139,168,184)
void thumbnail_rgb8(hipStream_t st, const uint8_t* src, int h, int w, uint8_t* dst, int nh, int nw, int* err_flag);

// det_processor.rs:151-160 + image_helper.rs:211-221: RGB8 HWC -> normalised BGR f32.
// layout 0: NHWC pitch 4 (B,G,R,0); layout 1: CHW planes
typedef nn::U8Page NormDesc;  // one page of a det launch group
void det_normalize_batch(hipStream_t st, const NormDesc* d_descs, int n, long long max_pix, float scale, const float* mean3,
                         const float* std3, float* out);
void det_normalize(hipStream_t st, const uint8_t* rgb, int h, int w, float scale, const float* mean3,
                   const float* std3, int layout, float* out);

// ---- DB post-processing (det_processor.rs:279-335) -------------------------------
struct DbParams {
  float thresh, box_thresh, unclip_ratio;
  int min_size, dilate;
};
struct DbBox { float pts[8]; float score; int key; };
// Work buffers for one page of H x W; sized by db_workspace_bytes().
size_t db_workspace_bytes(int H, int W, int max_boxes);
struct DbPageIn { const float* pred; int H, W, ori_h, ori_w; };
size_t db_page_desc_bytes();
// Runs the whole post-process for n pages with shared launches (blockIdx.y = page).
// workspaces[i]: device buffer of db_workspace_bytes(H_i, W_i, max_boxes); boxes_out[i]
// (device, max_boxes entries) receives the sorted boxes in after_* coordinates,
// count_out[i][0] the number, count_out[i][1] != 0 signals a capacity overflow.
// h_desc (pinned host) / d_desc (device): n * db_page_desc_bytes() scratch for the page table.
void db_postprocess_batch(hipStream_t st, int n, const DbPageIn* in, const DbParams& p, void* const* workspaces,
                          int max_boxes, DbBox* const* boxes_out, int* const* count_out, void* h_desc, void* d_desc);
// boxes [n][max_boxes] + counts [n][2] (both contiguous over the pages) -> packed list in page order
void pack_boxes(hipStream_t st, int n, const DbBox* boxes, const int* counts, int max_boxes, DbBox* packed);

// ---- crops -------------------------------------------------------------------------
struct CropDesc {
  const uint8_t* src; int sh, sw;   // source page (after resize_both)
  float inv[9];                     // output pixel -> source pixel
  int w, h, rot;                    // pre-rotation dims, rotate270 flag
  long long out_off;                // byte offset of this crop in the crop pool
};
void warp_crops(hipStream_t st, const CropDesc* descs, int n, int max_pix, uint8_t* pool);

// cls_processor.rs:108-121,163-166: argmax of [n,2]; rotate180 in place when label==180 && score>=thresh
struct CropRef { long long off; int h, w; int pad_; };   // final (post-rotation) dims
void cls_post_rotate(hipStream_t st, const float* probs, const int* crop_of_row, int rows, float thresh,
                     const CropRef* crops, uint8_t* pool, int max_pix, int* label_idx, float* score);

// image_helper.rs:176-209 resize_norm_image for a ragged batch of lines.
struct LineDesc {
  long long crop_off; int h, w;   // crop in the pool (current dims)
  int resized_w, W;               // thumbnail width, padded width
  long long out_off;              // element offset of this line's tensor in `out`
};
// layout 0: NHWC pitch 4 (R,G,B,0) per line [img_h][W][4]; layout 1: CHW [3][img_h][W]
void resize_norm(hipStream_t st, const LineDesc* lines, int n, int img_h, int max_W, const uint8_t* pool, int layout,
                 float* out, int* err_flag);

// rec_processor.rs:48-97 CTC greedy decode over per-token (argmax, prob) rows.
// tok_off[i] = first row of line i, T[i] rows; tokens written compacted at the same offsets.
void ctc_decode(hipStream_t st, const int* idx, const float* prob, const ImgGeom* lines, int n, int* tokens,
                int* n_tokens, float* score);

// sum of a float buffer into per-block doubles (partials has ceil(n/65536) entries)
int sum_blocks(long long n);
void sum_partial(hipStream_t st, const float* x, long long n, double* partials);

}  // namespace pp
}  // namespace rt
