// Whole-C2f launches (c2f_kernels.hip): host-side declarations.
// One launch replaces a complete C2f module of the reference graph (model.ncnn.param, e.g. :88-101):
//     cv1 (1x1 + SiLU over a zero-copy concat, the first segment optionally a fused nearest-x2 upsample)
//  -> n x Bottleneck (3x3 + SiLU -> 3x3 + SiLU -> shortcut add)
//  -> cv2 (1x1 + SiLU over concat[y0, y1, .., y_{n+1}])
// and, on the 20x20 level where a whole image fits one workgroup, also the stride-2 3x3 conv in front of it
// and the SPPF module behind it (cv1 -> three cascaded 5x5 max pools -> cv2).
#pragma once
#include "common.h"

namespace lp {

enum { C2F_W_S2 = 0, C2F_W_CV1, C2F_W_A0, C2F_W_B0, C2F_W_A1, C2F_W_B1, C2F_W_CV2, C2F_W_SP1, C2F_W_SP2, C2F_NW };

struct C2fArgs {
  const void* src0;   // cv1 input, first K segment (KA channels); UP: a half-resolution tensor read at (y/2, x/2)
  const void* src1;   // cv1 input, second K segment (KB channels)
  int pitch0, pitch1;
  void* cat;          // concat buffer of the module [y0 | y1 | .. | y_{n+1}], view base at y0
  int cat_pitch;
  void* out;          // cv2 output
  int out_pitch;
  const void* w[C2F_NW];    // MFMA A fragments per phase (C2fLayer::build)
  const float* b[C2F_NW];   // fp32 bias per phase, natural channel order
  // stride-2 entry conv (whole-image configurations): s2_in [N][2H][2W][s2_pitch] -> x (global; cv1 reads it back as src1 / part of src1)
  const void* s2_in;
  int s2_pitch;
  void* x;
  int x_pitch;
  // SPPF tail: cat2 = [s | p1 | p2 | p3] (view base at s), out2 = SPPF.cv2 output
  void* cat2;
  int cat2_pitch;
  void* out2;
  int out2_pitch;
  int N, H, W, tiles_x, tiles_y;
  int debug_store;    // 1: store every y segment to the concat buffer (tools/c2f_check.py), also those cv2 reads from LDS
  unsigned long long* stamps;  // diagnostic only (LITEPI_C2F_STAMPS=<file>): 16 clock stamps per workgroup
};

// Shape key of an instantiated kernel configuration
struct C2fShape {
  int C = 0;      // hidden channels c of the module (cv1 produces 2c, every bottleneck conv is c -> c)
  int NB = 1;     // bottlenecks
  int KA = 0;     // cv1 input channels taken from src0 (0: none)
  int KB = 0;     // cv1 input channels taken from src1
  int UP = 0;     // src0 is half resolution (fused Interp nearest x2)
  int COUT = 0;   // cv2 output channels
  int MODE = 0;   // 0: C2f; 1: stride-2 3x3 entry conv + C2f (whole image per workgroup); 2: entry conv + C2f + SPPF;
                  // -1: C2f without its cv1 (y0 | y1 already in the concat buffer: the stride-2 conv in front ran cv1 as its tail)
  int KS2 = 0;    // input channels of the entry conv
  bool operator==(const C2fShape& o) const {
    return C == o.C && NB == o.NB && KA == o.KA && KB == o.KB && UP == o.UP && COUT == o.COUT && MODE == o.MODE && KS2 == o.KS2;
  }
};

struct C2fLayer {
  C2fShape sh;
  int H = 0, W = 0;
  std::string name;
  DevBuf d_w[C2F_NW], d_b[C2F_NW];
  size_t lds_bytes = 0;
  double macs_per_image = 0;

  // fp32 weights over physical channels (here physical == logical: the supported shapes have no padded segments)
  struct Src {
    const std::vector<float>*cv1 = nullptr, *cv1_b = nullptr;             // [2C][KA + KB]
    const std::vector<float>*a[2] = {nullptr, nullptr}, *a_b[2] = {nullptr, nullptr};  // [C][9][C]
    const std::vector<float>*bb[2] = {nullptr, nullptr}, *bb_b[2] = {nullptr, nullptr};
    const std::vector<float>*cv2 = nullptr, *cv2_b = nullptr;             // [COUT][(2 + NB) C]
    const std::vector<float>*s2 = nullptr, *s2_b = nullptr;               // [XC][9][KS2]
    const std::vector<float>*sp1 = nullptr, *sp1_b = nullptr;             // [C][COUT]
    const std::vector<float>*sp2 = nullptr, *sp2_b = nullptr;             // [COUT][4C]
  };
  static bool supported(const C2fShape& s, int h, int w);
  bool cv2_from_lds() const;   // cv2 takes the last y segments from the LDS planes: they never reach the concat buffer
  void build(const C2fShape& s, int h, int w, const Src& src);
  struct IO {
    View src0, src1, cat, out;   // src0.base == nullptr when KA == 0
    View s2_in, x;               // MODE >= 1
    View cat2, out2;             // MODE == 2
  };
  void launch(const IO& io, int N, hipStream_t st) const;
  std::string kernel_name() const;
};

// Stand-alone 3x3 stride-2 conv + SiLU on the c2f machinery (s2conv_kernel): weights [cout][tap][cin] over physical channels
struct S2ConvLayer {
  int Cin = 0, Cout = 0, H = 0, W = 0;   // H, W: OUTPUT map
  bool lds_staged = false;               // s2lds_kernel (input tile in LDS: v2's Cin 24 / 48 / 96) instead of the gather kernel
  int ksplit = 1;                        // its passes over the input channels
  bool has_tail = false;                 // a 1x1 conv + SiLU (C2f.cv1) on the accumulators: Cout -> Cout
  std::string name;
  DevBuf d_w, d_b, d_w2, d_b2;
  static bool supported(int cin, int cout, int hout, int wout);
  static bool tail_supported(int cin, int cout, int cout2, int hout, int wout);
  // w_tail [cout][cout] (1x1), b_tail: the fused tail, or null
  void build(int cin, int cout, int hout, int wout, const std::vector<float>& w_taps, const std::vector<float>& bias,
             const std::vector<float>* w_tail = nullptr, const std::vector<float>* b_tail = nullptr);
  void launch(const View& in, const View& out, int N, hipStream_t st) const;
};

// SPPF in one launch (sppf_kernel): cv1 (1x1 + SiLU) -> three cascaded 5x5 max pools -> cv2 (1x1 + SiLU) over concat(s, p1, p2, p3),
// for the width the whole-image C2f kernel cannot take along (v2: 192 -> 96 -> 192 @20x20)
struct SppfLayer {
  int Cin = 0, C = 0, Cout = 0, H = 0, W = 0;
  std::string name;
  DevBuf d_w1, d_b1, d_w2, d_b2;
  double macs_per_image = 0;
  static bool supported(int cin, int c, int cout, int h, int w);
  // w1 [c][cin], w2 [cout][4 c] over physical channels, fp32
  void build(int cin, int c, int cout, int h, int w, const std::vector<float>& w1, const std::vector<float>& b1, const std::vector<float>& w2,
             const std::vector<float>& b2);
  void launch(const View& in, const View& out, int N, hipStream_t st) const;
};

}  // namespace lp
