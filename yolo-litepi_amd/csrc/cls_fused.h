// Fused ShuffleNetV2 stage kernel (cls_fused.hip): host-side declarations.
#pragma once
#include "common.h"

namespace lp {

typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));

struct FusedBlockW {
  const u32x4_t* w1;  // branch2.0 (+BN) A fragments [tile][step][lane][16 B]
  const float* b1;    // [bfp]
  const float* dw;    // branch2.3 (+BN) fp32 [9][bfp]
  const float* dwb;   // [bfp]
  const u32x4_t* w2;  // branch2.5 (+BN)
  const float* b2;    // [bfp]
};

struct FusedStageArgs {
  const void* in;     // [R, H, W, 2*bfp] fp16 (output of the stage's stride-2 block)
  void* out;          // same shape
  const int* m_dyn;   // device ROI count
  FusedBlockW blk[8];
  int nblk;
  int HW, W;          // pixels per ROI at this stage, map width
  int bf, bfp;        // logical / physical channels of one half
  int group;          // ROIs per workgroup (group * HW == 64)
  int in_pitch, out_pitch;
};

struct FusedHeadArgs {
  const void* in;        // stage-4 output [R, 2, 2, cin_p] fp16
  const int* m_dyn;
  const u32x4_t* w5;     // conv5 (+BN) A fragments [64 tiles][S5][lane]
  const float* b5;       // [1024]
  const u32x4_t* wfc;    // fc A fragments [nc_p/16 tiles][32 steps][lane]
  const float* bfc;      // [nc_p]
  float* probs;          // optional [R, nc]
  int* ids;              // optional [R]
  float* logits;         // optional [R, logits_pitch] (tests)
  lp_det* dets;          // optional: scatter (class, confidence) through the ROI table
  const int* roi_img;
  const int* roi_slot;
  int max_det, nc, nc_p, cin_p, in_pitch, logits_pitch;
};

size_t fused_head_lds_bytes(int cin_p, int nc_p);
void launch_fused_head(const FusedHeadArgs& a, int max_items, hipStream_t st);
size_t fused_stage_lds_bytes(int bfp);
void launch_fused_stage(const FusedStageArgs& a, int max_items, hipStream_t st);
std::vector<uint16_t> pack_fused_pw(const std::vector<float>& w_phys, int cout_p, int cin_p);

}  // namespace lp
