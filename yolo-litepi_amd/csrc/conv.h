// Host-side handles of the convolution kernels (conv_kernels.hip).
#pragma once
#include "common.h"

namespace lp {

struct ConvIO {
  View in, out, res, x1;
  View up;                     // 1x1 only: half-resolution source of the leading up.C input channels (fused Interp x2 + Concat)
  int N = 0;
  const int* m_dyn = nullptr;  // device scalar item count (classifier); M = *m_dyn * out.H * out.W
  int half_c = 0, half_cp = 0; // shuffle epilogue geometry (x1.base != nullptr)
  int out_f32 = 0;
  int res_first = 0;           // residual added BEFORE the activation (ResNet) instead of after it (C2f)
  unsigned long long* stamps = nullptr;  // diagnostic (see ConvArgs::stamps)
};

// One Convolution(+bias)(+activation)(+residual) layer with its device weights, packed for
// the kernel family chosen at build time.
struct ConvLayer {
  int prec = LP_FP16, impl = IMPL_MFMA;
  int k = 1, stride = 1, Cin = 0, Cout = 0, act = ACT_NONE;
  // MFMA tiling
  int NT = 1, nsplits = 1, CK = 0, CGc = 0, nchunks = 1, steps = 0, LW = 0, PS = 0, bwh = 2, bww = 2;
  size_t lds_bytes = 0;
  unsigned rcp_cg = 0, rcp_ps = 0, rcp_pcs = 0;  // 16-bit reciprocals of CGc, PS/16 and pieces per LDS tile row (3x3 kernel)
  bool direct = false;  // 3x3 stride-2: gather B fragments from global memory (no LDS input tile)
  DevBuf d_w, d_bias;
  // fused 1x1 tail conv (second GEMM in the epilogue): 0 = none, else its 16-channel output tiles
  int T2 = 0, Cout2 = 0, act2 = ACT_NONE;
  DevBuf d_w2, d_bias2;
  std::string name;
  // host copies of the physical fp32 weights (build / attach_tail arguments): the Detect-head fusion re-packs them into
  // one weight stream (HeadLayer); a few MB per model
  std::vector<float> w_host, b_host, w2_host, b2_host;
  double macs_per_pixel() const { return (double)k * k * Cin * Cout; }

  // w_phys: fp32 [Cout][k*k][Cin] over PHYSICAL channels (zeros at padding channels);
  // bias_phys: fp32 [Cout] or empty.  hout/wout: output map size and batch_hint: images (or ROIs)
  // per call at capacity -- together they pick the tile shape and the channel split.
  void build(int prec, int impl, int k, int stride, int cin, int cout, int act,
             const std::vector<float>& w_phys, const std::vector<float>& bias_phys, int hout, int wout, int batch_hint,
             bool full_n = false, bool single_chunk = false);
  // Fuse a following 1x1 conv (weights w2_phys [cout2][Cout] over physical channels) into this 3x3 layer.
  // Requires build(..., full_n = true) (one workgroup holds every intermediate channel).  The fused layer's
  // output view then has cout2 channels.
  static bool tail_supported(int k, int stride, int cmid_phys, int cout2_phys);
  void attach_tail(int cout2_phys, int act2, const std::vector<float>& w2_phys, const std::vector<float>& bias2_phys);
  void launch(const ConvIO& io, hipStream_t st) const;
};

// C2f bottleneck x + silu(conv3x3(silu(conv3x3(x)))) as ONE launch (bottleneck_mfma_kernel): the
// intermediate lives in LDS.  The two ConvLayers only carry the packed weights (all of K in one chunk).
struct BottleneckPair {
  ConvLayer a, b;
  int prec = LP_FP16, C = 0, NT = 1, TH = 8, TW = 40, LW = 0, PS = 0, CG = 0, steps = 0;
  unsigned rcp_cg = 0, rcp_ps = 0, rcp_pcs = 0, rcp_w1 = 0, rcp_tw = 0;
  size_t lds_bytes = 0;
  std::string name;
  // optional fused C2f.cv2 (1x1 over concat[stored segments .., y_last]); y_last = this bottleneck's output
  struct Cv2 {
    int cat_global = 0;          // physical channels of the concat that precede y_last (read from the concat buffer)
    int c3 = 0, act = ACT_NONE;  // cv2 output channels (physical), activation
    const std::vector<float>* w = nullptr;  // fp32 [c3][cat_global + C] over physical concat channels
    const std::vector<float>* bias = nullptr;
    bool in_is_last_stored = false;  // the bottleneck's input is the concat's last stored segment, same buffer and pitch (C2f: y_n)
  };
  int T2 = 0, C3 = 0, act3 = ACT_NONE, kg = 0, sg = 0;
  bool cl = false;   // "concat from LDS": the tile is staged with every stored segment, cv2 gathers from it (see the kernel)
  int PSA = 0;       // its pixel pitch in LDS
  // pixel tiles (of 16) per wave in conv_a / conv_b = the kernel's P1 / P2 template arguments
  int p1() const { return (((TH + 2) * (TW + 2) + 15) / 16 + 3) / 4; }
  int p2() const { return ((TH * TW + 15) / 16 + 3) / 4; }
  DevBuf d_w3, d_b3;
  // can a C->C bottleneck on an h x w map (batch_hint images) run fused?  (LDS capacity, tile limits); cv2 = nullptr: without tail
  static bool supported(int prec, int impl, int c_phys, int h, int w, int batch_hint, const Cv2* cv2 = nullptr);
  // weights as for ConvLayer::build: fp32 [C][9][C] over physical channels, bias [C]
  void build(int prec, int c_phys, const std::vector<float>& wa, const std::vector<float>& ba, const std::vector<float>& wb,
             const std::vector<float>& bb, int h, int w, int batch_hint, const Cv2* cv2 = nullptr);
  // without cv2: out = bottleneck output.  With cv2: cat = the concat buffer view (its leading cat_global channels are read),
  // out = cv2's output view; the bottleneck output itself is not stored.
  void launch(const View& in, const View& out, int N, hipStream_t st, const View* cat = nullptr) const;
 private:
  // kg_cl > 0: try the CL variant with kg_cl staged K groups per pixel first (cl_out says whether it was taken)
  static bool plan(int prec, int c_phys, int h, int w, int batch_hint, size_t extra_lds, bool tail, int& th, int& tw, int& lw, size_t& lds,
                   int kg_cl = 0, bool* cl_out = nullptr, int* psa_out = nullptr);
  static bool cv2_shape(int prec, int c_phys, const Cv2& cv2, int& t2, int& kg, int& sg);
};

// First layer: 3x3 stride-2 conv reading the uint8 BGR image directly.
struct StemLayer {
  int prec = LP_FP16, CO = 8, act = ACT_SILU;
  int k = 3, stride = 2, pad = 1;  // YOLOv8 family: 3x3/s2/p1 (fast kernels); anything else runs the generic kernel
  DevBuf d_w, d_bias;
  DevBuf d_afrag;  // fp16, 8 channels: MFMA A fragments (stem_mfma_kernel)
  DevBuf d_afrag_blk;  // the same weights in stem_block_kernel's K-group order
  // w_bgr: fp32 [k*k*3][CO], row = (ky*k+kx)*3 + c with c in BGR order
  void build(int prec, int cout_phys, int act, const std::vector<float>& w_bgr, const std::vector<float>& bias, int k = 3, int stride = 2,
             int pad = 1);
  void launch(const uint8_t* img, int N, int Hin, int Win, const View& out, hipStream_t st) const;
  // stem + the following stride-2 3x3 conv (+ its fused 1x1 tail) in one launch (stem_block_kernel); out = the tail's output
  bool block_supported(const ConvLayer& c1) const;
  void launch_block(const uint8_t* img, int N, int Hin, int Win, const ConvLayer& c1, const View& out, hipStream_t st) const;
};

}  // namespace lp
