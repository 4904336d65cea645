// Detector execution plan: NCNN graph -> fused NHWC kernels (host side).
#pragma once
#include "common.h"
#include "c2f.h"
#include "conv.h"
#include "head.h"
#include "kernels.h"
#include "ncnn_graph.h"

namespace lp {

struct Profiler {
  // e0 / e1: events recorded around the op; k0 / k1: events attached to the op's launch itself (LP_LAUNCH); launches: kernel
  // launches inside the bracket -- exactly one: the time reported is k0 -> k1 (the kernel), otherwise e0 -> e1 (the bracket)
  struct Rec { std::string name, layer; double flops, bytes; hipEvent_t e0, e1; bool per_roi; hipEvent_t k0, k1; int launches; };
  LaunchTimer timer;
  bool enabled = false;
  std::vector<Rec> recs;
  std::vector<lp_kernel_time> results;
  hipEvent_t cur = nullptr;
  void begin(hipStream_t st);
  // per_roi: flops/bytes are per ROI and get multiplied by the ROI count at collect time
  void end(hipStream_t st, const std::string& name, const std::string& layer, double flops, double bytes, bool per_roi = false);
  void collect(int roi_count);  // after the stream is synchronised
  ~Profiler();
};

// A materialised activation (or a channel-slice view of one) after alias resolution.
struct Tensor {
  std::string name;
  int C = 0, H = 0, W = 0;       // logical shape
  std::vector<int> segs;         // logical channel segments (each padded to 8 physically)
  int Cp = 0;                    // physical channels of the view
  int buf = -1, off = 0;         // buffer index, physical channel offset inside it
  int parent = -1, parent_seg = -1;  // Slice outputs: view of parent's segment
  bool materialised = false;
  bool in_c2f = false;           // lives inside a whole-C2f launch (LDS / registers, or a concat slot the launch may never write)
  int phys(int c) const;         // logical channel -> physical channel inside the view
};

struct Buffer { int Cp = 0, H = 0, W = 0; DevBuf mem; };

struct DetOp {
  enum Kind { STEM, STEMBLOCK, CONV, BNECK, DWCONV, ATTN, UPSAMPLE, SPPF, ADD, COPY, HEAD, C2F, S2C, SPPFUSED } kind = CONV;
  int conv = -1;                 // index into convs (CONV) / bnecks (BNECK)
  int in = -1, in2 = -1, res = -1, out = -1, out2 = -1, out3 = -1;
  std::string layer;
  double flops = 0, bytes = 0;   // per image
};

class Detector {
 public:
  Detector(int prec, int impl, int max_batch, int input_size);
  void load(const std::string& param_path, const std::string& bin_path);
  bool loaded() const { return loaded_; }
  // imgs: device uint8 BGR [B,S,S,3]; geom: device ImgGeom[B]; out0: optional device fp32 [B,4+nc,A].
  // Enqueues the forward pass + decode/conf filter; candidates land in cand/cand_count.
  void forward(const uint8_t* imgs, int B, const ImgGeom* geom, float conf, float* out0, Cand* cand, int* cand_count,
               hipStream_t st, Profiler* prof);
  int num_anchors() const { return A_; }
  int num_classes() const { return nc_; }
  int reg_max() const { return reg_max_; }
  double macs_per_image() const { return macs_; }
  int input_size() const { return S_; }
  // test aid: copy a blob of the last forward to host as fp32 [B,C,H,W]
  void fetch_blob(const std::string& name, int B, std::vector<float>& out, int& C, int& H, int& W) const;

 private:
  View view(int t) const;
  int prec_, impl_, maxB_, S_;
  int fwd_calls_ = 0;   // LITEPI_SKIP_OP diagnostic
  bool loaded_ = false;
  std::vector<Tensor> tensors_;
  std::map<std::string, int> blob2tensor_;   // every blob name (aliases included) -> tensor index
  std::vector<Buffer> buffers_;
  std::vector<std::unique_ptr<ConvLayer>> convs_;
  std::vector<std::unique_ptr<BottleneckPair>> bnecks_;
  std::vector<std::unique_ptr<HeadLayer>> heads_;  // fused Detect head, one per level (fp16 MFMA plan)
  // whole-C2f launches (c2f_kernels.hip): tensor indices of each launch's views (-1: unused)
  struct C2fIO { int src0 = -1, src1 = -1, up_c = 0, cat = -1, out = -1, s2_in = -1, x = -1, cat2 = -1, out2 = -1; };
  std::vector<std::unique_ptr<C2fLayer>> c2fs_;
  std::vector<std::unique_ptr<S2ConvLayer>> s2cs_;   // stand-alone stride-2 convs on the c2f machinery
  std::vector<std::unique_ptr<SppfLayer>> sppfs_;    // SPPF modules in one launch (sppf_kernel)
  std::vector<C2fIO> c2f_io_;
  bool fused_head_ = false;                        // every level fused: the stand-alone decode launch is gone
  // YOLO11 extras: stand-alone depthwise convs and the C2PSA attention block
  struct DwLayer { DevBuf w, b; int act = ACT_NONE; };
  struct AttnLayer { DevBuf pe_w, pe_b; int heads = 0, dk = 0, dv = 0; float scale = 1.f; };
  std::vector<DwLayer> dws_;
  std::vector<AttnLayer> attns_;
  StemLayer stem_;
  std::vector<DetOp> ops_;
  // detect tail
  struct Level { int box, cls, H, W, off; };
  std::vector<Level> levels_;
  int A_ = 0, nc_ = 0, reg_max_ = 0;
  double macs_ = 0;
  DevBuf d_anchors_, d_strides_, d_dfl_;
};

}  // namespace lp
