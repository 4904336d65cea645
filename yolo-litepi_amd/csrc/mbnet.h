// MobileNetV2 and EfficientNet-B0 classifier plans (host side): build_classifier('mobilenetv2' | 'efficientnet') +
// self.model(batch) of the reference (src/tt100k/pipeline/e2e.py:324-329, 393) = torchvision mobilenet_v2 /
// efficientnet_b0 with classifier[1] = Linear(1280, num_classes), eval mode.  Both are stacks of inverted-residual blocks:
//     [1x1 expand + BN + act] -> depthwise k x k (stride) + BN + act -> [squeeze-excitation] -> 1x1 project + BN (+ x)
// (MobileNetV2: ReLU6, 3x3, no SE; EfficientNet-B0: SiLU, 3x3 / 5x5, SE with in/4 squeeze channels).  The pointwise convs
// run on the detector's MFMA 1x1 kernel (the residual in its epilogue), the depthwise convs, the SE gate and the first
// conv on the small kernels of cls_kernels.hip.  Layer-at-a-time and un-tuned, like the ResNet18 plan.
#pragma once
#include "classifier.h"

namespace lp {

class MBNetClassifier : public ClassifierBase {
 public:
  enum Arch { MOBILENET_V2 = 0, EFFICIENTNET_B0 = 1 };
  MBNetClassifier(Arch arch, int prec, int impl, int max_rois, int num_classes, int input_size);
  // torchvision state_dict: features.0.{0,1}.*, features.i[.j].{conv|block}.*, features.{18|8}.{0,1}.*, classifier.1.*
  void load(const std::map<std::string, NamedTensor>& sd) override;
  bool loaded() const override { return loaded_; }
  int num_classes() const override { return ncls_; }
  int logits_pitch() const override { return lpitch_; }
  const float* logits() const override { return d_logits_.as<float>(); }
  bool fused_head() const override { return false; }
  void forward(const uint8_t* rgb, const int* d_R, hipStream_t st, Profiler* prof, const Post* post = nullptr) override;
  // widest activation per ROI (elements) of the architecture at cls_input S: capacity checks (lp_create)
  static double widest_per_roi(int S) { return 96.0 * (S / 2.0) * (S / 2.0); }

 private:
  struct Dw { DevBuf w, b; int C = 0, k = 3, stride = 1; std::string name; };
  struct Block {
    std::string name;
    int cin = 0, cout = 0, exp = 0, k = 3, stride = 1, hin = 0, hout = 0, sqp = 0;
    bool res = false;
    int pw_expand = -1, dw = -1, se_fc1 = -1, se_fc2 = -1, pw_project = -1;
  };
  Arch arch_;
  int prec_, impl_, maxR_, ncls_, S_, lpitch_ = 0, act_pw_ = ACT_RELU, act_dw_ = 3, stem_c_ = 32;
  bool loaded_ = false;
  std::vector<std::unique_ptr<ConvLayer>> convs_;
  std::vector<Dw> dws_;
  std::vector<Block> blocks_;
  DevBuf stem_w_, stem_b_;
  int last_ = -1, fc_ = -1, last_cin_ = 0, last_h_ = 0;
  DevBuf a_x_[4], a_mean_, a_se_m_, a_se_q_, a_se_s_, d_logits_;
};

}  // namespace lp
