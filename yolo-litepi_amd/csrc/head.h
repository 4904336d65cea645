// Fused Detect head of one pyramid level (head_kernels.hip): host-side declarations.
// Replaces, per level, the six convolutions of Ultralytics' Detect (box tower cv2[i] = 3x3 -> 3x3 -> 1x1 (4*reg_max),
// class tower cv3[i] = 3x3 -> 3x3 -> 1x1 (nc); reference graph model.ncnn.param:151-182) AND the decode tail
// (Reshape/Permute/Softmax/DFL conv/dist2bbox/Sigmoid, :184-208) + conf filter of postprocess (e2e.py:255-278).
#pragma once
#include "common.h"
#include "kernels.h"

namespace lp {

struct HeadArgs {
  const void* in;          // neck feature map F [N][H][W][in_pitch] fp16
  const void* wstream;     // 1 KiB MFMA A fragments in consumption order (HeadLayer::build)
  const float* biasA;      // [32 * RT]  first 3x3 convs: box tower 64 | class tower 32*C3T (zero padded)
  const float* biasB;      // [32 * RT]  second 3x3 convs
  const float* biasC;      // [64 + 32]  projections: 4 x 16 DFL logits | class logits (zero padded to 32)
  const void* zeros;       // >= 16 zero bytes
  const float* anchors;    // [2][A] grid units
  const float* strides;    // [A]
  const float* dfl_w;      // [16]
  float* out0;             // optional [N][4+nc][A]
  const ImgGeom* geom;     // [N]
  Cand* cand;              // [N][A]
  int* cand_count;         // [N]
  float conf;
  int N, H, W, in_pitch, Cin;
  int TH, TW, tiles_x, ntiles;
  int KPT;                 // K steps (of 16 channels) per 3x3 tap of the first convs = Cin / 16
  int nchunks;             // weight-stream chunks
  int A, nc, anchor_off;
  int flags;               // experiment switches (LITEPI_HEAD_FLAGS): 1 = s_setprio 1 in the K loops, 2 = s_setprio 1 in the epilogues
  unsigned long long* stamps;  // diagnostic only (LITEPI_HEAD_STAMPS=<file>): 16 clock stamps per workgroup
};

// One level's weights re-packed for head_fused_kernel.
struct HeadLayer {
  int Cin = 0, H = 0, W = 0, c3 = 0, C3T = 1, nc = 1, TH = 16, TW = 16, PA = 3, PB = 2;
  int KPT = 0, KSA = 6, NPC = 2, SLOTF = 24, nchunks = 0;   // SLOTF: fragments per weight-ring slot
  int OVL = 0;                                              // MID overlays the input tile in LDS
  int A16 = 0;                                              // stage A on 16x16x32 MFMAs (16-pixel tiles)
  std::vector<unsigned short> coff;
  std::vector<unsigned char> csz, cks;
  DevBuf d_stream, d_biasA, d_biasB, d_biasC;
  size_t lds_bytes = 0;
  std::string name;
  double macs_per_image = 0;
  struct Src {  // physical fp32 weights, [out][taps][in]
    const std::vector<float>*wa, *ba;        // merged first convs: [64 + c3][9][Cin], bias [64 + c3]
    const std::vector<float>*wbb, *bbb;      // box tower second conv [64][9][64]
    const std::vector<float>*wpb, *bpb;      // box projection [64][64]
    const std::vector<float>*wbc, *bbc;      // class tower second conv [c3][9][c3]
    const std::vector<float>*wpc, *bpc;      // class projection [ncp][c3] (ncp = physical class channels >= nc)
    int ncp;
  };
  static bool supported(int cin_phys, int c2, int c3, int nc, int reg_max, int h, int w);
  void build(int cin_phys, int c3, int nc, int h, int w, int batch_hint, const Src& s);
  // anchor_off: index of the level's first anchor; out0 may be null
  void launch(const View& in, int N, int anchor_off, int A, const float* anchors, const float* strides, const float* dfl_w, float* out0,
              const ImgGeom* geom, Cand* cand, int* cand_count, float conf, hipStream_t st) const;
};

}  // namespace lp
