// Device-side pieces of the detector post-processing shared by post_kernels.hip (stand-alone decode) and
// head_kernels.hip (decode fused into the Detect-head kernel).
#pragma once
#include "common.h"
#include "kernels.h"

namespace lp {

// ------------------------------------------------------------------------------------
// conf filter + xywh->xyxy + un-letterbox + clip for one anchor (e2e.py:255-278)
// ------------------------------------------------------------------------------------
__device__ __forceinline__ void emit_candidate(float cx, float cy, float w, float h, float score, int cls, int anchor,
                                               const ImgGeom& gm, float conf, Cand* cand, int* count) {
  if (!(score > conf)) return;
  const float hw = __fmul_rn(w, 0.5f), hh = __fmul_rn(h, 0.5f);  // w / 2 (exact either way)
  float x1 = __fsub_rn(cx, hw), y1 = __fsub_rn(cy, hh);
  float x2 = __fadd_rn(cx, hw), y2 = __fadd_rn(cy, hh);
  x1 = __fdiv_rn(__fsub_rn(x1, gm.pad_w), gm.ratio);
  x2 = __fdiv_rn(__fsub_rn(x2, gm.pad_w), gm.ratio);
  y1 = __fdiv_rn(__fsub_rn(y1, gm.pad_h), gm.ratio);
  y2 = __fdiv_rn(__fsub_rn(y2, gm.pad_h), gm.ratio);
  const float W = (float)gm.w, H = (float)gm.h;
  x1 = fminf(fmaxf(x1, 0.f), W); x2 = fminf(fmaxf(x2, 0.f), W);
  y1 = fminf(fmaxf(y1, 0.f), H); y2 = fminf(fmaxf(y2, 0.f), H);
  const int slot = atomicAdd(count, 1);
  Cand c;
  c.x1 = x1; c.y1 = y1; c.x2 = x2; c.y2 = y2; c.score = score; c.cls = cls; c.anchor = anchor; c.pad = 0;
  cand[slot] = c;
}

// ------------------------------------------------------------------------------------
// One axis of cv2.resize(INTER_LINEAR) on uint8: source index and the 11-bit coefficient pair of destination index d
// (OpenCV's published fixed-point algorithm: half-pixel centres, cvRound(f * 2048)).  Used by the letterbox kernel
// (e2e.py:80) and by the ROI resize of the e2e_optimize numerics (e2e_optimize.py:388-390).
// ------------------------------------------------------------------------------------
__device__ __forceinline__ void lin_coeff(int d, int dst, int src, int& s0, int& a0, int& a1) {
  const double inv_scale = (double)dst / (double)src;
  const double scale = 1.0 / inv_scale;
  float f = (float)((d + 0.5) * scale - 0.5);
  int s = (int)floorf(f);
  f -= (float)s;
  if (s < 0) { f = 0.f; s = 0; }
  if (s >= src - 1) { f = 0.f; s = src - 1; }
  s0 = s;
  a1 = __float2int_rn(f * 2048.f);
  a0 = __float2int_rn((1.f - f) * 2048.f);
}

}  // namespace lp
