// Shared host-side declarations for liblitepi_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/litepi.h"

namespace lp {

// ---- errors ---------------------------------------------------------------------
struct Error : std::runtime_error {
  int code;
  Error(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};

std::string fmt(const char* f, ...) __attribute__((format(printf, 1, 2)));
void set_last_error(const std::string& s);
// raise a kernel's dynamic-LDS limit once per (device, kernel); throws when the runtime refuses
void set_max_dynamic_lds(const void* fn, int bytes);

#define LP_HIP(expr)                                                                       \
  do {                                                                                     \
    hipError_t _e = (expr);                                                                \
    if (_e != hipSuccess)                                                                  \
      throw lp::Error(LP_ERR_HIP, lp::fmt("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), \
                                          __FILE__, __LINE__));                            \
  } while (0)

// Kernel launches go through LP_LAUNCH: while a Profiler bracket is open (detector.h) the launch carries its own start / stop
// events (hipExtLaunchKernelGGL), i.e. the events time the kernel and not the dispatch around it -- between two
// hipEventRecord calls the 80x80 head measured 115 us against 105 us in rocprofv3's kernel trace.  Outside a bracket (the
// product path, graph capture) it is a plain launch.
struct LaunchTimer {
  hipEvent_t e0 = nullptr, e1 = nullptr;   // attached to the first launch inside the bracket
  int launches = 0;                        // launches seen inside the bracket
};
extern thread_local LaunchTimer* g_launch_timer;
bool print_launches();
#define LP_LAUNCH(kernel, grid, block, lds, st, ...)                                                       \
  do {                                                                                                     \
    lp::LaunchTimer* lt_ = lp::g_launch_timer;                                                             \
    if (lt_ && lp::print_launches()) {  /* LITEPI_PRINT_LAUNCH=1: grid / block / LDS of every profiled launch */ \
      const dim3 g_ = (grid), b_ = (block);                                                                \
      fprintf(stderr, "[launch] %-60.60s grid %5u block %4u lds %6zu B = %3zu granules of 1280\n", #kernel, g_.x * g_.y * g_.z, b_.x,     \
              (size_t)(lds), ((size_t)(lds) + 1279) / 1280);                                               \
    }                                                                                                      \
    if (lt_ && lt_->launches++ == 0 && lt_->e0)                                                            \
      hipExtLaunchKernelGGL(kernel, grid, block, lds, st, lt_->e0, lt_->e1, 0, __VA_ARGS__);              \
    else                                                                                                   \
      hipLaunchKernelGGL(kernel, grid, block, lds, st, __VA_ARGS__);                                       \
  } while (0)

#define LP_CHECK(cond, code, ...)                                \
  do {                                                           \
    if (!(cond)) throw lp::Error(code, lp::fmt(__VA_ARGS__));    \
  } while (0)

// ---- device memory ----------------------------------------------------------------
struct DevBuf {
  void* p = nullptr;
  size_t bytes = 0;
  DevBuf() = default;
  DevBuf(const DevBuf&) = delete;
  DevBuf& operator=(const DevBuf&) = delete;
  DevBuf(DevBuf&& o) noexcept : p(o.p), bytes(o.bytes) { o.p = nullptr; o.bytes = 0; }
  DevBuf& operator=(DevBuf&& o) noexcept {
    if (this != &o) { release(); p = o.p; bytes = o.bytes; o.p = nullptr; o.bytes = 0; }
    return *this;
  }
  ~DevBuf() { release(); }
  void alloc(size_t n, bool zero = true) {
    release();
    if (n == 0) n = 16;
    LP_HIP(hipMalloc(&p, n));
    bytes = n;
    if (zero) {
      // hipMemset runs on the null stream and may return before it has executed; the handle's
      // streams are non-blocking, so a later async copy/kernel could otherwise overtake the fill
      LP_HIP(hipMemset(p, 0, n));
      LP_HIP(hipStreamSynchronize(nullptr));
    }
  }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    bytes = 0;
  }
  template <typename T> T* as() const { return reinterpret_cast<T*>(p); }
};

inline int round_up(int x, int m) { return (x + m - 1) / m * m; }
inline int ceil_div(int a, int b) { return (a + b - 1) / b; }

// fp32 <-> fp16 on the host (round-to-nearest-even; weights are converted once at load)
uint16_t f32_to_f16(float f);
float f16_to_f32(uint16_t h);

// ---- a channel-slice view of an NHWC activation buffer ------------------------------
struct View {
  void* base = nullptr;  // address of channel 0 of pixel 0 of THIS view (offset already applied)
  int C = 0;             // physical channels of the view (multiple of 8)
  int pitch = 0;         // physical channels per pixel of the underlying buffer
  int H = 0, W = 0;
};

// ---- kernel-side argument blocks ----------------------------------------------------
struct ConvArgs {
  const void* in;
  void* out;
  const void* res;      // optional residual added after the activation (same shape as out)
  const void* x1;       // shuffle-interleave partner (classifier epilogue)
  const void* wpk;      // weights packed for the chosen kernel
  const float* bias;    // fp32 [padded Cout]
  const int* m_dyn;     // optional device scalar: number of items (ROIs); M = *m_dyn * pix_per_item
  int N, Hin, Win, Hout, Wout;
  int in_pitch, out_pitch, res_pitch, x1_pitch;
  int Cin, Cout;        // physical channel counts
  int act;              // 0 none, 1 SiLU, 2 ReLU
  int M;                // 1x1: flattened pixel count (static)
  int pix_per_item;
  // 3x3 MFMA tiling
  int CK, nchunks, steps_per_chunk, CGc, LW, PS, bwh, bww, tiles_x, tiles_y;
  // 1x1 MFMA
  int steps;            // K steps over all of Cin
  int nsplit_tiles;     // channel tiles handled per block (== NT template arg)
  // shuffle epilogue
  int half_c, half_cp;  // logical / physical channels of one half of the output
  int out_f32;          // store fp32 regardless of T (classifier logits)
  // fused 1x1 tail (second GEMM on the accumulator tile): A fragments [T2][S2][lane][16 B], bias, activation, channels
  const void* w2;
  const float* bias2;
  int act2, Cout2;
  // 1x1 kernel, fused nearest-x2 upsample: the first up_cg K groups come from a half-resolution tensor
  const void* up;
  int up_pitch, up_cg;
  int tile_major;              // 0: grid = (image, tile, split) (XCD-aware, default); 1: (tile, image, split)
  unsigned rcp_tx, rcp_cg, rcp_ps, rcp_pcs;  // 3x3 kernel: 16-bit reciprocals of tiles_x, CGc, PS/16, pieces per tile row
  int res_first;               // 1: out = act(conv + bias + res) (ResNet BasicBlock); 0: out = act(conv + bias) + res (C2f shortcut)
  unsigned magic_hw, magic_w;  // floor(2^32 / pix_per_item), floor(2^32 / Wout): flat pixel index -> (image, row, column) (fast_div)
  const void* zeros;           // >= 16 zero bytes (source of the 3x3 kernel's padding slots)
  unsigned long long* stamps;  // diagnostic only (lp_test_conv + LITEPI_STAMPS): 16 clock stamps per workgroup
};

// Fused C2f bottleneck: out = in + silu(conv3x3(silu(conv3x3(in)))) (bottleneck_mfma_kernel)
struct BneckArgs {
  const void* in;
  void* out;
  const void* w1;   // fragment-ordered weights of the two convs, all of K in one chunk: [step][tile][lane][16 B]
  const void* w2;
  const float* b1;
  const float* b2;
  const void* zeros;
  int N, H, W, C;   // C: physical channels (in == mid == out)
  int in_pitch, out_pitch;
  int TH, TW, tiles_x, LW, PS, CG, steps;
  unsigned rcp_tx, rcp_cg, rcp_ps, rcp_pcs, rcp_w1, rcp_tw;
  int tile_major;
  // fused C2f.cv2 (1x1 over the concat [y0, y1, .., y_last]): y_last comes from the accumulators, the other
  // segments are read from the concat buffer `cat` (channels [0, kg*G)); `out` is not written then
  const void* w3;   // cv2 fragments [T2][sg + sr][lane][16 B]
  const float* b3;
  const void* cat;
  void* out3;
  int cat_pitch, out3_pitch, C3, act3;
  int kg, sg;       // global K groups and their K steps (sr = register steps, fixed by NT)
  int PSA;          // CL variant: bytes per staged pixel (all kg stored groups, padded to an odd number of 16-byte slots)
  unsigned long long* stamps;  // diagnostic only (LITEPI_BNECK_STAMPS=<file>): 16 clock stamps per workgroup
};

// Fused network head: stem 3x3/s2 (uint8 -> 8 ch) + 3x3/s2 conv + its 1x1 tail (stem_block_kernel)
struct StemBlockArgs {
  const uint8_t* img;
  const void* afrag;   // stem MFMA A fragments (StemLayer::d_afrag_blk)
  const float* sbias;  // stem bias
  const void* w1;      // stride-2 conv fragments [3 steps][64 lanes][16 B] (K = 9 taps x 8 channels)
  const float* b1;
  const void* w2;      // 1x1 tail fragments (ConvLayer::d_w2)
  const float* b2;
  void* out;
  int N, Hin, Win, H1, W1, H2, W2, out_pitch, C2, act2;
};

enum ConvImpl { IMPL_MFMA = 0, IMPL_NAIVE = 1 };
enum Act { ACT_NONE = 0, ACT_SILU = 1, ACT_RELU = 2 };

}  // namespace lp
