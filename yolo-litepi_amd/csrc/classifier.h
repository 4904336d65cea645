// ShuffleNetV2 x1.0 classifier plan (host side).  Replaces build_classifier('shufflenetv2')
// + self.model(batch) of the reference (e2e.py:331-333,393).
#pragma once
#include "common.h"
#include "conv.h"
#include "detector.h"
#include "cls_fused.h"
#include "cls_net.h"
#include "kernels.h"

namespace lp {

struct NamedTensor {
  const float* data = nullptr;
  std::vector<int64_t> shape;
  size_t numel() const { size_t n = 1; for (auto d : shape) n *= (size_t)d; return n; }
};

// where a classifier that finishes with softmax / arg-max itself (fused head) puts the results
struct ClsPost { float* probs = nullptr; int* ids = nullptr; lp_det* dets = nullptr; int max_det = 0; const int* roi_img = nullptr; const int* roi_slot = nullptr; };

// What the pipeline needs from a classifier architecture (build_classifier(arch, ..), e2e.py:320-333)
class ClassifierBase {
 public:
  typedef ClsPost Post;
  virtual ~ClassifierBase() {}
  virtual void load(const std::map<std::string, NamedTensor>& sd) = 0;   // torchvision state_dict of the architecture
  virtual bool loaded() const = 0;
  virtual int num_classes() const = 0;
  virtual int logits_pitch() const = 0;
  virtual const float* logits() const = 0;
  // rgb: device uint8 [R,S,S,3]; d_R: device ROI count.  Leaves fp32 logits [R, logits_pitch()].
  // With a fused head softmax/arg-max/scatter happen inside forward(); post describes where the results go.
  // fused_head() tells the caller whether it still has to run softmax itself.
  virtual bool fused_head() const = 0;
  virtual void forward(const uint8_t* rgb, const int* d_R, hipStream_t st, Profiler* prof, const Post* post = nullptr) = 0;
};

class Classifier : public ClassifierBase {
 public:
  Classifier(int prec, int impl, int max_rois, int num_classes, int input_size);
  // torchvision shufflenet_v2_x1_0 state_dict (fc replaced by Linear(1024, num_classes))
  void load(const std::map<std::string, NamedTensor>& sd) override;
  bool loaded() const override { return loaded_; }
  int num_classes() const override { return ncls_; }
  int logits_pitch() const override { return lpitch_; }
  const float* logits() const override { return d_logits_.as<float>(); }
  bool fused_head() const override { return use_fused_; }
  void forward(const uint8_t* rgb, const int* d_R, hipStream_t st, Profiler* prof, const Post* post = nullptr) override;

 private:
  struct DwLayer { DevBuf w, b; int C = 0, stride = 1; std::string name; };
  struct Block {
    int stride = 1, inp = 0, oup = 0;
    int b1_dw = -1, b1_pw = -1;          // stride 2 only
    int b2_pw1 = -1, b2_dw = -1, b2_pw2 = -1;
    std::string name;
  };
  struct Act { DevBuf mem; int C = 0, H = 0, W = 0; };   // [max_rois,H,W,C]
  View act_view(const Act& a, int coff = 0, int C = -1) const;
  void alloc_act(Act& a, int C, int H, int W);
  int add_pw(const std::string& name, const std::vector<float>& w_phys, const std::vector<float>& b_phys, int cin, int cout, int act, int hw);
  int add_dw(const std::string& name, const std::vector<float>& w_phys, const std::vector<float>& b_phys, int C, int stride);

  int prec_, impl_, maxR_, ncls_, S_, lpitch_ = 0;
  bool loaded_ = false;
  std::vector<std::unique_ptr<ConvLayer>> pws_;
  std::vector<DwLayer> dws_;
  std::vector<Block> blocks_;
  DevBuf stem_w_, stem_b_;
  int conv5_ = -1, fc_ = -1;
  // activations
  Act a_stem_, a_pool_, a_t1_, a_t2_, a_b1dw_, a_b1_, a_stage_[3][2], a_conv5_, a_mean_;
  DevBuf d_logits_;
  int half_c_[3], half_cp_[3];
  // fused stride-1 stage path (fp16 + MFMA only)
  struct FusedW { DevBuf w1, b1, dw, dwb, w2, b2; };
  std::vector<FusedW> fused_[3];
  // stride-2 blocks + MFMA stem of the two-launch network (cls_net.hip)
  struct FusedS2 { DevBuf dw1, dw1b, pwb1, pwb1b, pw1, pw1b, dw2, dw2b, pw2, pw2b; };
  FusedS2 fused_s2_[3];
  DevBuf stem_frag_, stem_bias4_, a_x3_;
  bool use_fused_ = false;
  DevBuf head_w5_, head_b5_, head_wfc_, head_bfc_;
  int head_cin_p_ = 0, head_nc_p_ = 0;
};

}  // namespace lp
