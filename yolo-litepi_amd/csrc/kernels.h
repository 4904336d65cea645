// Host launchers of the non-convolution kernels (misc_kernels.hip, post_kernels.hip,
// cls_kernels.hip).  All launches are asynchronous on the given stream.
#pragma once
#include "common.h"

namespace lp {

// ---- per-image geometry (letterbox + un-letterbox), device resident ------------------
struct ImgGeom {
  long src_off;        // byte offset of this image in the source buffer (uint8 BGR HxWx3)
  int h, w;            // original size
  int new_w, new_h;    // resized (unpadded) size inside the letterboxed square
  int top, left;       // integer border offsets (e2e.py:82-83)
  float ratio;         // r (e2e.py:72)
  float pad_w, pad_h;  // float half pads dw, dh (e2e.py:76-77), subtracted un-rounded in postprocess
};

// ---- misc_kernels.hip ------------------------------------------------------------------
// letterbox (e2e.py:66-86): B images -> uint8 BGR [B,S,S,3], cv2.INTER_LINEAR fixed point, border 114
void launch_letterbox(const uint8_t* src, const ImgGeom* geom, uint8_t* dst, int B, int S, hipStream_t st, const ImgGeom* host_geoms = nullptr);
// Interp nearest x2 (model.ncnn.param:88,103)
void launch_upsample2x(int prec, const View& in, const View& out, int N, hipStream_t st);
// SPPF: three cascaded 5x5/s1/p2 max pools (model.ncnn.param:79-83) in one pass
void launch_sppf_pool(int prec, const View& in, const View& o1, const View& o2, const View& o3, int N, hipStream_t st);
// fallbacks for graphs whose concat/add could not be fused
void launch_add(int prec, const View& a, const View& b, const View& out, int N, hipStream_t st);
void launch_copy(int prec, const View& in, const View& out, int N, hipStream_t st);
// YOLO11: ConvolutionDepthWise 3x3/s1/p1 (+bias, optional SiLU), w fp32 [9][C] over physical channels
void launch_dwconv3x3_act(int prec, const View& in, const View& out, const float* w, const float* bias, int act, int N, hipStream_t st);
// YOLO11 C2PSA attention block (Reshape .. MatMul .. Softmax .. MatMul .. + depthwise positional encoding), see misc_kernels.hip
void launch_psa_attention(int prec, const View& qkv, const View& out, const float* pe_w, const float* pe_b, int heads, int dk, int dv,
                          float scale, int N, hipStream_t st);

// ---- post_kernels.hip -----------------------------------------------------------------
struct Cand {  // one candidate / kept box, 32 bytes
  float x1, y1, x2, y2, score;
  int cls, anchor, pad;
};

struct DecodeLevel {
  const void* box;  // [N,H,W,4*reg_max] view base
  const void* cls;  // [N,H,W,nc] view base
  int box_pitch, cls_pitch, H, W, anchor_off;
};

struct DecodeArgs {
  DecodeLevel lv[4];
  int nlevels, A, nc, reg_max;
  const float* anchors;  // [2][A] grid units (x row, y row)
  const float* strides;  // [A]
  const float* dfl_w;    // [reg_max]
  float* out0;           // optional [N,4+nc,A]
  const ImgGeom* geom;   // [N]
  Cand* cand;            // [N][A]
  int* cand_count;       // [N]
  float conf;
};
// Detect head decode (model.ncnn.param:184-208) + conf filter / xywh->xyxy / un-letterbox /
// clip (e2e.py:255-278) fused; out0 is only written when requested (parity hook).
void launch_decode(int prec, const DecodeArgs& a, int N, hipStream_t st);
// Same filter/transform applied to an existing out0 tensor [N,4+nc,A] (tests).
void launch_filter_out0(const float* out0, int nc, int A, const ImgGeom* geom, Cand* cand, int* cand_count,
                        float conf, int N, hipStream_t st);

// ROI bookkeeping across the batch: every image's NMS block appends its kept ROIs to one list
struct RoiTable {
  int* base;      // [N+1] (unused by the fused path; kept for the host-filled table of lp_classify)
  int* total;     // [2]  min(sum, max_rois), then the unclamped sum
  int* img;       // [max_rois] image index of ROI r
  int* slot;      // [max_rois] detection slot of ROI r within its image
  int* work;      // [2] accumulator + ticket of the NMS blocks (zero between calls)
};

struct NmsArgs {
  const Cand* cand;       // [N][A]
  int* cand_count;        // [N]   (reset to 0 by the kernel for the next call)
  Cand* sorted;           // [N][A] scratch
  lp_det* dets;           // [N][max_det]
  int* counts;            // [3N]: kept-after-filter, kept-before-filter, float bits of the mean score before the filter
  int* rects;             // [N][max_det][4] int ROI rectangles of the kept boxes
  const ImgGeom* geom;
  int A, max_det, nc;
  float iou;
  int min_area;           // < 0: no ROI filter (lp_detect semantics)
  RoiTable tab;           // tab.total == nullptr: no ROI list
  int max_rois;
  int roi_rule;           // 0: e2e.py:465-473, 1: e2e_optimize.py:480-497 (lp_config::numerics)
  int no_small;           // A/B + tests (LITEPI_NMS_NO_SMALL=1): every image takes the general path
};
// per-class greedy NMS (e2e.py:89-119,280-296) + ROI clip / area filter (e2e.py:465-473) + the batch's ROI list
void launch_nms(const NmsArgs& a, int N, hipStream_t st);
size_t nms_lds_bytes(int A);


// PIL Image.resize((S,S), BILINEAR) of every ROI + BGR->RGB (e2e.py:385-389): uint8 RGB [R,S,S,3]
struct RoiResizeArgs {
  const uint8_t* src;       // source images
  const ImgGeom* geom;      // per image (src_off/h/w)
  const int* rects;         // [N][max_det][4]
  RoiTable tab;
  uint8_t* out;             // [max_rois,S,S,3]
  int max_det, S;
  int linear;               // 0: PIL bilinear with antialias (e2e.py:366-370, 387), 1: cv2.resize INTER_LINEAR (e2e_optimize.py:388-390)
};
void launch_roi_resize(const RoiResizeArgs& a, int max_items, hipStream_t st);
size_t roi_resize_lds_bytes();

// ---- cls_kernels.hip ------------------------------------------------------------------
// conv1 3x3/s2 (3->CO) + folded BN + ReLU on (x/255 - mean)/std of the uint8 RGB crops
void launch_cls_stem(int prec, const uint8_t* rgb, const float* w /*[27][CO]*/, const float* bias, int CO,
                     const View& out, int S, const int* m_dyn, int max_items, hipStream_t st);
// ResNet18 conv1 7x7/s2 (3 -> 64) + folded BN + ReLU on the normalised crops; w fp32 [147][64] in (ky, kx, rgb) order
void launch_cls_stem7(int prec, const uint8_t* rgb, const float* w, const float* bias, const View& out, int S, const int* m_dyn,
                      int max_items, hipStream_t st);
void launch_maxpool3x3s2(int prec, const View& in, const View& out, const int* m_dyn, int max_items, hipStream_t st);
// depthwise 3x3 pad 1 stride 1|2 + folded BN (no activation); w fp32 [9][C], bias fp32 [C]
void launch_dwconv3x3(int prec, const View& in, const View& out, const float* w, const float* bias, int stride,
                      const int* m_dyn, int max_items, hipStream_t st);
// MobileNetV2 / EfficientNet-B0 (mbnet.cpp).  act: 0 none, 1 SiLU, 2 ReLU, 3 ReLU6.
// features[0] 3x3/s2 (3 -> CO) + folded BN + activation on the normalised uint8 crops; w fp32 [27][CO] in (ky, kx, rgb) order
void launch_cls_stem_act(int prec, const uint8_t* rgb, const float* w, const float* bias, int CO, int act, const View& out, int S,
                         const int* m_dyn, int max_items, hipStream_t st);
// depthwise k x k (3 | 5), pad k/2, stride 1|2, + folded BN + activation; w fp32 [k*k][C]
void launch_dwconv_act(int prec, const View& in, const View& out, const float* w, const float* bias, int k, int stride, int act,
                       const int* m_dyn, int max_items, hipStream_t st);
// in place: scale == null: x = min(x, cap); else x[r, p, c] *= sigmoid(scale[r, c]) (squeeze-excitation gate)
void launch_mb_eltwise(int prec, const View& x, const View* scale, float cap, const int* m_dyn, int max_items, hipStream_t st);
// x.mean([2,3]) : [R,H,W,C] -> [R,1,1,C]
void launch_spatial_mean(int prec, const View& in, const View& out, const int* m_dyn, int max_items, hipStream_t st);
// softmax + argmax over fp32 logits [R,pitch]; writes probs [R,nc], ids [R], conf [R] and, when dets != null,
// scatters (id, conf) into the detection records through the ROI table
void launch_softmax_argmax(const float* logits, int pitch, int nc, float* probs, int* ids, float* conf,
                           lp_det* dets, int max_det, const RoiTable* tab, const int* m_dyn, int max_items,
                           hipStream_t st);

}  // namespace lp
