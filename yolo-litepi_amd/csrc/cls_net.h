// Whole-network ShuffleNetV2 x1.0 kernels (cls_net.hip), fp16 storage / fp32 accumulate: host-side declarations.
// Two launches replace the ~25 of the layer-at-a-time plan (reference: self.model(batch), e2e.py:393, on the
// ToTensor/Normalize'd 64x64 crops of e2e.py:366-370):
//   cls_front: one workgroup per ROI   -- conv1+BN+ReLU, maxpool, stage2 (4 blocks), stage3.0       -> [R,4,4,256]
//   cls_back : one workgroup per 4 ROIs -- stage3.1-7, stage4 (4 blocks), conv5, mean, fc, softmax, arg-max, scatter
// Activations never leave LDS inside a launch; weights stream from L2 straight into MFMA A fragments.
#pragma once
#include "cls_fused.h"
#include "common.h"

namespace lp {

// stride-2 InvertedResidual (torchvision shufflenetv2.py), BN folded, weights over the PHYSICAL
// channels of the block's input layout
struct FusedS2W {
  const float* dw1;      // branch1.0: depthwise 3x3/s2 over the cin_p input channels, fp32 [9][cin_p]
  const float* dw1b;     // [cin_p]
  const u32x4_t* pwb1;   // branch1.2 + ReLU: cin_p -> bfp, A fragments [bfp/16][ceil(cin_p/32)][64 lanes][16 B]
  const float* pwb1b;    // [bfp]
  const u32x4_t* pw1;    // branch2.0 + ReLU: cin_p -> bfp
  const float* pw1b;
  const float* dw2;      // branch2.3: depthwise 3x3/s2 over bfp channels, fp32 [9][bfp]
  const float* dw2b;
  const u32x4_t* pw2;    // branch2.5 + ReLU: bfp -> bfp, [bfp/16][ceil(bfp/32)][64][16 B]
  const float* pw2b;
};

struct ClsFrontArgs {
  const uint8_t* rgb;    // [R,64,64,3] uint8 RGB (roi_resize_pil output)
  const int* m_dyn;      // device ROI count
  void* out;             // [R,16,256] fp16: stage3.0 output, channel layout [lo 116 | pad 12 | hi 116 | pad 12]
  const u32x4_t* stem_w; // conv1 (+BN, +1/255/std folded) A fragments [2 tiles][64 lanes][16 B], K = 3 rows x 9 bytes + pad
  const float* stem_b;   // [4 border cases][32]: bias - mean/std * sum of the weights of the taps inside the image
  FusedS2W s20;          // stage2.0: cin_p 24, bfp 64
  FusedBlockW s2[3];     // stage2.1-3
  FusedS2W s30;          // stage3.0: cin_p 128 (stage-2 layout), bfp 128
};

struct ClsBackArgs {
  const void* in;        // cls_front's output
  const int* m_dyn;
  FusedBlockW s3[7];     // stage3.1-7
  FusedS2W s40;          // stage4.0: cin_p 256 (stage-3 layout), bfp 240
  FusedBlockW s4[3];     // stage4.1-3
  const u32x4_t* w5;     // conv5 (+BN) A fragments [64 tiles][15 steps][64 lanes]
  const float* b5;       // [1024]
  const u32x4_t* wfc;    // fc A fragments [nc_p/16][32][64]
  const float* bfc;      // [nc_p]
  float* probs;          // optional [R, nc]
  int* ids;              // optional [R]
  float* logits;         // optional [R, logits_pitch]
  lp_det* dets;          // optional: scatter (class, confidence) through the ROI table
  const int* roi_img;
  const int* roi_slot;
  int max_det, nc, nc_p, logits_pitch;
};

void launch_cls_front(const ClsFrontArgs& a, int max_items, hipStream_t st);
void launch_cls_back(const ClsBackArgs& a, int max_items, hipStream_t st);
// conv1 for the MFMA stem: w [24][3][3][3] (torch layout, BN folded), bias [24] -> fragments + the four bias cases
void pack_cls_stem(const std::vector<float>& w_oihw, const std::vector<float>& bias, std::vector<uint16_t>& frags, std::vector<float>& bias_cases);

}  // namespace lp
