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


}  // namespace lp
