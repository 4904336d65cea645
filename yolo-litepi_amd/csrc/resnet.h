// ResNet18 classifier plan (host side): build_classifier('resnet18') + self.model(batch) of the reference
// (src/tt100k/pipeline/e2e.py:320-323, 393) = torchvision resnet18 with fc = Linear(512, num_classes), eval mode.
// The 3x3 convolutions run on the detector's implicit-GEMM MFMA kernels (conv3x3_mfma / conv3x3s2_direct, ReLU epilogue,
// identity added BEFORE the activation), the 1x1 stride-2 downsample convs on the same stride-2 kernel with their weights
// on the centre tap, conv1 (7x7/s2 from uint8) on cls_stem7_kernel.  One launch per layer: 22 launches per call.
#pragma once
#include "classifier.h"

namespace lp {

class ResNet18Classifier : public ClassifierBase {
 public:
  ResNet18Classifier(int prec, int impl, int max_rois, int num_classes, int input_size);
  // torchvision resnet18 state_dict: conv1.weight, bn1.*, layer{1..4}.{0,1}.{conv1,bn1,conv2,bn2}.*, layer{2..4}.0.downsample.{0,1}.*, fc.*
  void load(const std::map<std::string, NamedTensor>& sd) override;
  bool loaded() const override { return loaded_; }
  int num_classes() const override { return ncls_; }
  int logits_pitch() const override { return lpitch_; }
  const float* logits() const override { return d_logits_.as<float>(); }
  bool fused_head() const override { return false; }
  void forward(const uint8_t* rgb, const int* d_R, hipStream_t st, Profiler* prof, const Post* post = nullptr) override;

 private:
  struct Block { int conv1 = -1, conv2 = -1, down = -1, cin = 0, cout = 0, stride = 1, hout = 0; std::string name; };
  int prec_, impl_, maxR_, ncls_, S_, lpitch_ = 0;
  bool loaded_ = false;
  std::vector<std::unique_ptr<ConvLayer>> convs_;
  std::vector<Block> blocks_;
  DevBuf stem_w_, stem_b_;
  int fc_ = -1;
  DevBuf a_stem_, a_x_[3], a_mean_, d_logits_;   // [R,32,32,64]; three rotating block buffers sized for [R,16,16,64]
};

}  // namespace lp
