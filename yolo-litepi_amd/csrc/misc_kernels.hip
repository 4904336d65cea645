// Byte-moving detector kernels: letterbox, nearest upsample, SPPF pooling, add/copy.
// All HBM-bound: 16-byte vector accesses over NHWC channel groups, one thread per
// (pixel, 8-channel fp16 / 4-channel fp32 group).
#include "common.h"
#include "kernels.h"
#include "post_dev.h"

#include <algorithm>
#include <cstdlib>

namespace lp {

typedef _Float16 half_t;
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <typename T> struct VecT;
template <> struct VecT<half_t> { typedef half8 type; static constexpr int G = 8; };
template <> struct VecT<float> { typedef floatx4 type; static constexpr int G = 4; };

// ------------------------------------------------------------------------------------
// letterbox (reference e2e.py:66-86).  cv2.resize(INTER_LINEAR) on uint8 is restated from
// OpenCV's published fixed-point algorithm: half-pixel centres, 11-bit coefficients
// (cvRound(f*2048)), int horizontal pass, vertical pass
//   ((b0*(S0>>4))>>16) + ((b1*(S1>>4))>>16) + 2) >> 2.
// cv2 is absent from the build container: only the identity and pad-only cases are pinned.
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void letterbox_kernel(const uint8_t* __restrict__ src, const ImgGeom* __restrict__ geom,
                                                        uint8_t* __restrict__ dst, int S) {
  const int n = blockIdx.y;
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= S * S) return;
  const int oy = idx / S, ox = idx - oy * S;
  const ImgGeom gm = geom[n];
  uint8_t* o = dst + ((long)n * S * S + idx) * 3;
  const int dy = oy - gm.top, dx = ox - gm.left;
  if (dy < 0 || dy >= gm.new_h || dx < 0 || dx >= gm.new_w) {
    o[0] = 114; o[1] = 114; o[2] = 114;
    return;
  }
  const uint8_t* im = src + gm.src_off;
  if (gm.new_w == gm.w && gm.new_h == gm.h) {
    const uint8_t* p = im + ((long)dy * gm.w + dx) * 3;
    o[0] = p[0]; o[1] = p[1]; o[2] = p[2];
    return;
  }
  int sx, ax0, ax1, sy, ay0, ay1;
  lin_coeff(dx, gm.new_w, gm.w, sx, ax0, ax1);
  lin_coeff(dy, gm.new_h, gm.h, sy, ay0, ay1);
  const int sx1 = sx + 1 < gm.w ? sx + 1 : gm.w - 1;
  const int sy1 = sy + 1 < gm.h ? sy + 1 : gm.h - 1;
  const uint8_t* r0 = im + (long)sy * gm.w * 3;
  const uint8_t* r1 = im + (long)sy1 * gm.w * 3;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const int h0 = r0[sx * 3 + c] * ax0 + r0[sx1 * 3 + c] * ax1;
    const int h1 = r1[sx * 3 + c] * ax0 + r1[sx1 * 3 + c] * ax1;
    int v = (((ay0 * (h0 >> 4)) >> 16) + ((ay1 * (h1 >> 4)) >> 16) + 2) >> 2;
    v = v < 0 ? 0 : (v > 255 ? 255 : v);
    o[c] = (uint8_t)v;
  }
}

// Tiled form (round 4; the kernel above gathers 12 single bytes per output pixel straight from global memory: one
// texture-path request per byte).  A workgroup owns an 8 x 128 tile of the letterboxed image: for each of its 8 output rows the
// TWO source rows the vertical interpolation reads (cv2's non-antialiased INTER_LINEAR touches only those: at the 3.2x
// down-scale of a 2048 x 2048 frame 5/8 of the source rows) are staged into LDS with aligned 16-byte loads over the byte
// span the tile's columns need, then every thread resamples four adjacent output pixels from LDS -- the same fixed-point
// arithmetic, bit for bit -- and stores them as three dwords.  HBM: every needed source byte once, in >= 1 KB runs.
#define LB_TH 8
#define LB_TW 128
__global__ __launch_bounds__(256) void letterbox_tiled_kernel(const uint8_t* __restrict__ src, const ImgGeom* __restrict__ geom,
                                                              uint8_t* __restrict__ dst, int S, int row_cap) {
  extern __shared__ __attribute__((aligned(16))) uint8_t band[];   // [2 * LB_TH][row_cap]
  __shared__ int s_shift[2 * LB_TH];                                // byte index of source byte b0 inside an LDS row
  const int n = blockIdx.z, tid = threadIdx.x;
  const int ox0 = blockIdx.x * LB_TW, oy0 = blockIdx.y * LB_TH;
  const ImgGeom gm = geom[n];
  const uint8_t* im = src + gm.src_off;
  const bool resize = !(gm.new_w == gm.w && gm.new_h == gm.h);
  // columns of the resized image this tile covers, and the source byte span [b0, b1) they read
  const int dxa = max(ox0 - gm.left, 0), dxb = min(min(ox0 + LB_TW, S) - gm.left, gm.new_w);   // [dxa, dxb)
  int b0 = 0, b1 = 0;
  if (dxa < dxb) {
    int sa, sb, t0, t1;
    if (resize) { lin_coeff(dxa, gm.new_w, gm.w, sa, t0, t1); lin_coeff(dxb - 1, gm.new_w, gm.w, sb, t0, t1); }
    else { sa = dxa; sb = dxb - 1; }
    const int sb1 = sb + 1 < gm.w ? sb + 1 : gm.w - 1;
    b0 = 3 * sa; b1 = 3 * (sb1 + 1);
  }
  const long img_bytes = (long)gm.h * gm.w * 3;
  // ---- stage: LDS row 2j / 2j+1 = source rows sy / sy1 of output row oy0 + j
  if (dxa < dxb) {
    for (int r = tid >> 4; r < 2 * LB_TH; r += 16) {   // 16 threads per row
      const int dy = oy0 + (r >> 1) - gm.top;
      if (dy < 0 || dy >= gm.new_h) continue;
      int sy, ay0, ay1;
      if (resize) lin_coeff(dy, gm.new_h, gm.h, sy, ay0, ay1); else sy = dy;
      const int syr = (r & 1) ? (sy + 1 < gm.h ? sy + 1 : gm.h - 1) : sy;
      const long goff = (long)syr * gm.w * 3 + b0;                      // byte offset of the span inside the image
      const long gal = ((gm.src_off + goff) & ~15L) - gm.src_off;        // 16-byte aligned start (the source buffer itself is 256-byte aligned)
      const int shift = (int)(goff - gal);
      if ((tid & 15) == 0) s_shift[r] = shift;
      const int nchunk = (shift + (b1 - b0) + 15) >> 4;
      for (int c = tid & 15; c < nchunk; c += 16) {
        const long o = gal + 16L * c;
        u32x4 v;
        if (o >= -gm.src_off && o + 16 <= img_bytes) {
          v = *reinterpret_cast<const u32x4*>(im + o);
        } else {   // the chunk straddles the end (or start) of the buffer's image: byte by byte, nothing outside is touched
          uint8_t t[16];
#pragma unroll
          for (int k = 0; k < 16; ++k) t[k] = (o + k >= 0 && o + k < img_bytes) ? im[o + k] : (uint8_t)0;
          v = *reinterpret_cast<const u32x4*>(t);
        }
        *reinterpret_cast<u32x4*>(band + (size_t)r * row_cap + 16 * c) = v;
      }
    }
  }
  __syncthreads();
  // ---- resample: thread = (row j, four adjacent columns)
  const int j = tid >> 5, ox = ox0 + 4 * (tid & 31), oy = oy0 + j;
  if (oy >= S || ox >= S) return;
  const int dy = oy - gm.top;
  const bool rowin = dy >= 0 && dy < gm.new_h;
  int ay0 = 2048, ay1 = 0;
  if (rowin && resize) { int sy; lin_coeff(dy, gm.new_h, gm.h, sy, ay0, ay1); }
  const uint8_t* r0 = band + (size_t)(2 * j) * row_cap + (rowin ? s_shift[2 * j] : 0) - b0;
  const uint8_t* r1 = band + (size_t)(2 * j + 1) * row_cap + (rowin ? s_shift[2 * j + 1] : 0) - b0;
  uint32_t out[3] = {0u, 0u, 0u};
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int dx = ox + q - gm.left;
    uint32_t px[3] = {114u, 114u, 114u};
    if (rowin && dx >= 0 && dx < gm.new_w) {
      if (resize) {
        int sx, ax0, ax1;
        lin_coeff(dx, gm.new_w, gm.w, sx, ax0, ax1);
        const int sx1 = sx + 1 < gm.w ? sx + 1 : gm.w - 1;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const int h0 = r0[sx * 3 + c] * ax0 + r0[sx1 * 3 + c] * ax1;
          const int h1 = r1[sx * 3 + c] * ax0 + r1[sx1 * 3 + c] * ax1;
          int v = (((ay0 * (h0 >> 4)) >> 16) + ((ay1 * (h1 >> 4)) >> 16) + 2) >> 2;
          px[c] = (uint32_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
        }
      } else {
#pragma unroll
        for (int c = 0; c < 3; ++c) px[c] = r0[dx * 3 + c];
      }
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const int b = 3 * q + c;
      out[b >> 2] |= px[c] << (8 * (b & 3));
    }
  }
  uint32_t* o = reinterpret_cast<uint32_t*>(dst + ((long)n * S * S + (long)oy * S + ox) * 3);   // 12-byte groups: 4-byte aligned (S % 4 == 0)
  o[0] = out[0]; o[1] = out[1]; o[2] = out[2];
}

// host_geoms (optional, the B geometries): lets the launcher size the tiled kernel's LDS rows; without them, or when a source
// row span does not fit (down-scales beyond ~10x), the per-pixel kernel runs
void launch_letterbox(const uint8_t* src, const ImgGeom* geom, uint8_t* dst, int B, int S, hipStream_t st, const ImgGeom* host_geoms) {
  static const bool no_tiled = getenv("LITEPI_LB_NAIVE") != nullptr;   // A/B switch
  if (host_geoms && !no_tiled && S % 4 == 0) {
    int cap = 0;
    for (int i = 0; i < B; ++i) {
      const ImgGeom& g = host_geoms[i];
      const double scale = (double)g.w / (double)std::max(g.new_w, 1);
      const int span = (int)(((double)LB_TW * scale + 4.0) * 3.0) + 32;   // source bytes of a tile row + alignment slack
      cap = std::max(cap, (span + 15) & ~15);
    }
    if ((size_t)cap * 2 * LB_TH <= 64 * 1024) {
      dim3 grid(ceil_div(S, LB_TW), ceil_div(S, LB_TH), B);
      LP_LAUNCH(letterbox_tiled_kernel, grid, dim3(256), (size_t)cap * 2 * LB_TH, st, src, geom, dst, S, cap);
      LP_HIP(hipGetLastError());
      return;
    }
  }
  dim3 grid(ceil_div(S * S, 256), B);
  LP_LAUNCH(letterbox_kernel, grid, dim3(256), 0, st, src, geom, dst, S);
  LP_HIP(hipGetLastError());
}

// ------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void upsample2x_kernel(const T* __restrict__ in, T* __restrict__ out, int N, int H, int W,
                                                         int CG, int in_pitch, int out_pitch) {
  constexpr int G = VecT<T>::G;
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  const long total = (long)N * (2 * H) * (2 * W) * CG;
  if (idx >= total) return;
  const int cg = (int)(idx % CG);
  const long pix = idx / CG;
  const int ox = (int)(pix % (2 * W));
  const int oy = (int)((pix / (2 * W)) % (2 * H));
  const int n = (int)(pix / ((long)4 * W * H));
  const u32x4 v = *reinterpret_cast<const u32x4*>(in + ((long)(n * H + (oy >> 1)) * W + (ox >> 1)) * in_pitch + cg * G);
  *reinterpret_cast<u32x4*>(out + pix * out_pitch + cg * G) = v;
}

void launch_upsample2x(int prec, const View& in, const View& out, int N, hipStream_t st) {
  LP_CHECK(out.H == 2 * in.H && out.W == 2 * in.W && out.C >= in.C, LP_ERR_STATE, "upsample2x: shape mismatch");
  const int G = prec == LP_FP16 ? 8 : 4;
  const int CG = in.C / G;
  const long total = (long)N * out.H * out.W * CG;
  dim3 grid((unsigned)((total + 255) / 256));
  if (prec == LP_FP16)
    LP_LAUNCH(upsample2x_kernel<half_t>, grid, dim3(256), 0, st, (const half_t*)in.base, (half_t*)out.base, N, in.H,
                       in.W, CG, in.pitch, out.pitch);
  else
    LP_LAUNCH(upsample2x_kernel<float>, grid, dim3(256), 0, st, (const float*)in.base, (float*)out.base, N, in.H,
                       in.W, CG, in.pitch, out.pitch);
  LP_HIP(hipGetLastError());
}

// ------------------------------------------------------------------------------------
// SPPF: y1 = pool5(x), y2 = pool5(y1), y3 = pool5(y2) with -inf padding == max over the
// clipped 5x5 / 9x9 / 13x13 windows of x.
// ------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void sppf_pool_kernel(const T* __restrict__ in, T* __restrict__ o1, T* __restrict__ o2,
                                                        T* __restrict__ o3, int N, int H, int W, int CG, int in_pitch,
                                                        int p1, int p2, int p3) {
  typedef typename VecT<T>::type vec;
  constexpr int G = VecT<T>::G;
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  const long total = (long)N * H * W * CG;
  if (idx >= total) return;
  const int cg = (int)(idx % CG);
  const long pix = idx / CG;
  const int x = (int)(pix % W);
  const int y = (int)((pix / W) % H);
  const int n = (int)(pix / ((long)W * H));
  vec m5, m9, m13;
#pragma unroll
  for (int i = 0; i < G; ++i) { m5[i] = (T)-INFINITY; m9[i] = (T)-INFINITY; m13[i] = (T)-INFINITY; }
  for (int dy = -6; dy <= 6; ++dy) {
    const int yy = y + dy;
    if (yy < 0 || yy >= H) continue;
    const int ady = dy < 0 ? -dy : dy;
    for (int dx = -6; dx <= 6; ++dx) {
      const int xx = x + dx;
      if (xx < 0 || xx >= W) continue;
      const int adx = dx < 0 ? -dx : dx;
      const int r = ady > adx ? ady : adx;
      const vec v = *reinterpret_cast<const vec*>(in + ((long)(n * H + yy) * W + xx) * in_pitch + cg * G);
#pragma unroll
      for (int i = 0; i < G; ++i) {
        m13[i] = v[i] > m13[i] ? v[i] : m13[i];
        if (r <= 4) m9[i] = v[i] > m9[i] ? v[i] : m9[i];
        if (r <= 2) m5[i] = v[i] > m5[i] ? v[i] : m5[i];
      }
    }
  }
  *reinterpret_cast<vec*>(o1 + pix * p1 + cg * G) = m5;
  *reinterpret_cast<vec*>(o2 + pix * p2 + cg * G) = m9;
  *reinterpret_cast<vec*>(o3 + pix * p3 + cg * G) = m13;
}

// LDS version for maps up to 40x40: one workgroup per (image, 16-byte channel group); each 5x5
// pool is separable (row max, then column max), cascaded three times in place.
#define SPPF_MAX_PIX 1600
template <typename T>
__global__ __launch_bounds__(256) void sppf_pool_lds_kernel(const T* __restrict__ in, T* __restrict__ o1, T* __restrict__ o2,
                                                            T* __restrict__ o3, int H, int W, int CG, int in_pitch, int p1,
                                                            int p2, int p3) {
  typedef typename VecT<T>::type vec;
  constexpr int G = VecT<T>::G;
  __shared__ vec A[SPPF_MAX_PIX];
  __shared__ vec B[SPPF_MAX_PIX];
  const int cg = blockIdx.x % CG, n = blockIdx.x / CG;
  const int HW = H * W, tid = threadIdx.x;
  const long base = (long)n * HW;
  for (int p = tid; p < HW; p += 256) A[p] = *reinterpret_cast<const vec*>(in + (base + p) * in_pitch + cg * G);
  __syncthreads();
  T* outs[3] = {o1, o2, o3};
  const int pitches[3] = {p1, p2, p3};
  for (int stage = 0; stage < 3; ++stage) {
    for (int p = tid; p < HW; p += 256) {
      const int y = p / W, x = p - y * W;
      vec m = A[p];
      for (int d = -2; d <= 2; ++d) {
        const int xx = x + d;
        if (d == 0 || xx < 0 || xx >= W) continue;
        const vec v = A[y * W + xx];
#pragma unroll
        for (int i = 0; i < G; ++i) m[i] = v[i] > m[i] ? v[i] : m[i];
      }
      B[p] = m;
    }
    __syncthreads();
    for (int p = tid; p < HW; p += 256) {
      const int y = p / W, x = p - y * W;
      vec m = B[p];
      for (int d = -2; d <= 2; ++d) {
        const int yy = y + d;
        if (d == 0 || yy < 0 || yy >= H) continue;
        const vec v = B[yy * W + x];
#pragma unroll
        for (int i = 0; i < G; ++i) m[i] = v[i] > m[i] ? v[i] : m[i];
      }
      A[p] = m;
      *reinterpret_cast<vec*>(outs[stage] + (base + p) * pitches[stage] + cg * G) = m;
    }
    __syncthreads();
  }
}

void launch_sppf_pool(int prec, const View& in, const View& o1, const View& o2, const View& o3, int N, hipStream_t st) {
  const int G = prec == LP_FP16 ? 8 : 4;
  const int CG = in.C / G;
  if (in.H * in.W <= SPPF_MAX_PIX) {
    dim3 grid((unsigned)(N * CG));
    if (prec == LP_FP16)
      LP_LAUNCH(sppf_pool_lds_kernel<half_t>, grid, dim3(256), 0, st, (const half_t*)in.base, (half_t*)o1.base,
                         (half_t*)o2.base, (half_t*)o3.base, in.H, in.W, CG, in.pitch, o1.pitch, o2.pitch, o3.pitch);
    else
      LP_LAUNCH(sppf_pool_lds_kernel<float>, grid, dim3(256), 0, st, (const float*)in.base, (float*)o1.base,
                         (float*)o2.base, (float*)o3.base, in.H, in.W, CG, in.pitch, o1.pitch, o2.pitch, o3.pitch);
    LP_HIP(hipGetLastError());
    return;
  }
  const long total = (long)N * in.H * in.W * CG;
  dim3 grid((unsigned)((total + 255) / 256));
  if (prec == LP_FP16)
    LP_LAUNCH(sppf_pool_kernel<half_t>, grid, dim3(256), 0, st, (const half_t*)in.base, (half_t*)o1.base,
                       (half_t*)o2.base, (half_t*)o3.base, N, in.H, in.W, CG, in.pitch, o1.pitch, o2.pitch, o3.pitch);
  else
    LP_LAUNCH(sppf_pool_kernel<float>, grid, dim3(256), 0, st, (const float*)in.base, (float*)o1.base,
                       (float*)o2.base, (float*)o3.base, N, in.H, in.W, CG, in.pitch, o1.pitch, o2.pitch, o3.pitch);
  LP_HIP(hipGetLastError());
}

// ------------------------------------------------------------------------------------
template <typename T, bool ADD>
__global__ __launch_bounds__(256) void eltwise_kernel(const T* __restrict__ a, const T* __restrict__ b, T* __restrict__ out,
                                                      long npix, int CG, int pa, int pb, int po) {
  typedef typename VecT<T>::type vec;
  constexpr int G = VecT<T>::G;
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= npix * CG) return;
  const int cg = (int)(idx % CG);
  const long pix = idx / CG;
  vec v = *reinterpret_cast<const vec*>(a + pix * pa + cg * G);
  if (ADD) {
    const vec w = *reinterpret_cast<const vec*>(b + pix * pb + cg * G);
#pragma unroll
    for (int i = 0; i < G; ++i) v[i] = (T)((float)v[i] + (float)w[i]);
  }
  *reinterpret_cast<vec*>(out + pix * po + cg * G) = v;
}

void launch_add(int prec, const View& a, const View& b, const View& out, int N, hipStream_t st) {
  const int G = prec == LP_FP16 ? 8 : 4;
  const int CG = a.C / G;
  const long npix = (long)N * a.H * a.W;
  dim3 grid((unsigned)((npix * CG + 255) / 256));
  if (prec == LP_FP16)
    LP_LAUNCH((eltwise_kernel<half_t, true>), grid, dim3(256), 0, st, (const half_t*)a.base, (const half_t*)b.base,
                       (half_t*)out.base, npix, CG, a.pitch, b.pitch, out.pitch);
  else
    LP_LAUNCH((eltwise_kernel<float, true>), grid, dim3(256), 0, st, (const float*)a.base, (const float*)b.base,
                       (float*)out.base, npix, CG, a.pitch, b.pitch, out.pitch);
  LP_HIP(hipGetLastError());
}

void launch_copy(int prec, const View& in, const View& out, int N, hipStream_t st) {
  const int G = prec == LP_FP16 ? 8 : 4;
  const int CG = in.C / G;
  const long npix = (long)N * in.H * in.W;
  dim3 grid((unsigned)((npix * CG + 255) / 256));
  if (prec == LP_FP16)
    LP_LAUNCH((eltwise_kernel<half_t, false>), grid, dim3(256), 0, st, (const half_t*)in.base, (const half_t*)nullptr,
                       (half_t*)out.base, npix, CG, in.pitch, 0, out.pitch);
  else
    LP_LAUNCH((eltwise_kernel<float, false>), grid, dim3(256), 0, st, (const float*)in.base, (const float*)nullptr,
                       (float*)out.base, npix, CG, in.pitch, 0, out.pitch);
  LP_HIP(hipGetLastError());
}

// ------------------------------------------------------------------------------------
// Depthwise 3x3 / stride 1 / pad 1 (+bias, optional SiLU) on a detector activation (NHWC).  YOLO11's DWConv layers
// (Detect class branch) -- ConvolutionDepthWise in the NCNN export.  w: fp32 [9][C] over physical channels.
// ------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void dwconv3x3_act_kernel(const T* __restrict__ in, T* __restrict__ out,
                                                            const float* __restrict__ w, const float* __restrict__ bias,
                                                            long npix, int H, int W, int C, int in_pitch, int out_pitch, int act) {
  typedef typename VecT<T>::type vec;
  constexpr int G = VecT<T>::G;
  const int CG = C / G;
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= npix * CG) return;
  const int cg = (int)(idx % CG);
  const long pix = idx / CG;
  const int ox = (int)(pix % W), oy = (int)((pix / W) % H);
  float acc[G];
#pragma unroll
  for (int i = 0; i < G; ++i) acc[i] = bias[cg * G + i];
  for (int ky = 0; ky < 3; ++ky) {
    const int iy = oy - 1 + ky;
    if (iy < 0 || iy >= H) continue;
    for (int kx = 0; kx < 3; ++kx) {
      const int ix = ox - 1 + kx;
      if (ix < 0 || ix >= W) continue;
      const vec v = *reinterpret_cast<const vec*>(in + (pix + (long)(ky - 1) * W + (kx - 1)) * in_pitch + cg * G);
      const float* wr = w + (ky * 3 + kx) * C + cg * G;
#pragma unroll
      for (int i = 0; i < G; ++i) acc[i] = fmaf((float)v[i], wr[i], acc[i]);
    }
  }
  vec o;
#pragma unroll
  for (int i = 0; i < G; ++i) o[i] = (T)(act == ACT_SILU ? acc[i] * __builtin_amdgcn_rcpf(1.f + __expf(-acc[i])) : acc[i]);
  *reinterpret_cast<vec*>(out + pix * out_pitch + cg * G) = o;
}

void launch_dwconv3x3_act(int prec, const View& in, const View& out, const float* w, const float* bias, int act, int N, hipStream_t st) {
  const int G = prec == LP_FP16 ? 8 : 4;
  LP_CHECK(in.C == out.C && in.C % G == 0 && in.H == out.H && in.W == out.W, LP_ERR_STATE, "dwconv3x3: view mismatch");
  const long npix = (long)N * in.H * in.W;
  dim3 grid((unsigned)((npix * (in.C / G) + 255) / 256));
  if (prec == LP_FP16)
    LP_LAUNCH(dwconv3x3_act_kernel<half_t>, grid, dim3(256), 0, st, (const half_t*)in.base, (half_t*)out.base, w, bias, npix,
                       in.H, in.W, in.C, in.pitch, out.pitch, act);
  else
    LP_LAUNCH(dwconv3x3_act_kernel<float>, grid, dim3(256), 0, st, (const float*)in.base, (float*)out.base, w, bias, npix,
                       in.H, in.W, in.C, in.pitch, out.pitch, act);
  LP_HIP(hipGetLastError());
}

// ------------------------------------------------------------------------------------
// YOLO11 C2PSA attention (model.ncnn.param of the reference's yolo11 export, layers reshape_166 .. add_7): the qkv
// tensor [N, HW, heads*(2*dk+dv)] holds per head dk query, dk key and dv value channels.  Per head:
//     P = softmax_m(scale * q_n . k_m),   o_n = sum_m P[n][m] v_m,   out = o + dwconv3x3(v) + pe_bias
// One workgroup = 16 queries of one (image, head); scores in LDS (fp32), K / V read through L2 (a head's K and V are
// ~75 KB).  fp32 arithmetic whatever the storage type.
// ------------------------------------------------------------------------------------
#define ATT_QB 16
template <typename T>
__global__ __launch_bounds__(256) void psa_attention_kernel(const T* __restrict__ qkv, T* __restrict__ out, const float* __restrict__ pe_w,
                                                            const float* __restrict__ pe_b, int H, int W, int heads, int dk, int dv,
                                                            float scale, int in_pitch, int out_pitch) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int HW = H * W;
  float* sq = reinterpret_cast<float*>(smem);         // [ATT_QB][dk]
  float* ss = sq + ATT_QB * dk;                        // [ATT_QB][HW]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = blockIdx.x, h = blockIdx.y, n0 = blockIdx.z * ATT_QB;
  const int hc = h * (2 * dk + dv);
  const T* base = qkv + (long)n * HW * in_pitch + hc;
  for (int i = tid; i < ATT_QB * dk; i += 256) {
    const int qi = i / dk, j = i - qi * dk;
    sq[i] = n0 + qi < HW ? (float)base[(long)(n0 + qi) * in_pitch + j] : 0.f;
  }
  __syncthreads();
  for (int m = tid; m < HW; m += 256) {
    float acc[ATT_QB];
#pragma unroll
    for (int qi = 0; qi < ATT_QB; ++qi) acc[qi] = 0.f;
    const T* kp = base + (long)m * in_pitch + dk;
    for (int j = 0; j < dk; ++j) {
      const float kv = (float)kp[j];
#pragma unroll
      for (int qi = 0; qi < ATT_QB; ++qi) acc[qi] = fmaf(sq[qi * dk + j], kv, acc[qi]);
    }
#pragma unroll
    for (int qi = 0; qi < ATT_QB; ++qi) ss[qi * HW + m] = acc[qi] * scale;
  }
  __syncthreads();
  for (int qi = wave; qi < ATT_QB; qi += 4) {  // softmax over the keys, one wave per query row
    float mx = -3.0e38f;
    for (int m = lane; m < HW; m += 64) mx = fmaxf(mx, ss[qi * HW + m]);
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    float sum = 0.f;
    for (int m = lane; m < HW; m += 64) { const float e = expf(ss[qi * HW + m] - mx); ss[qi * HW + m] = e; sum += e; }
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    const float inv = 1.f / sum;
    for (int m = lane; m < HW; m += 64) ss[qi * HW + m] *= inv;
  }
  __syncthreads();
  for (int i = tid; i < ATT_QB * dv; i += 256) {
    const int qi = i / dv, d = i - qi * dv;
    const int q = n0 + qi;
    if (q >= HW) continue;
    const T* vp = base + 2 * dk + d;
    float acc = 0.f;
    for (int m = 0; m < HW; ++m) acc = fmaf(ss[qi * HW + m], (float)vp[(long)m * in_pitch], acc);
    // positional encoding: depthwise 3x3 over the value map of this channel
    const int c = h * dv + d, oy = q / W, ox = q - oy * W;
    float pe = pe_b[c];
    for (int ky = 0; ky < 3; ++ky) {
      const int iy = oy - 1 + ky;
      if (iy < 0 || iy >= H) continue;
      for (int kx = 0; kx < 3; ++kx) {
        const int ix = ox - 1 + kx;
        if (ix < 0 || ix >= W) continue;
        pe = fmaf((float)vp[(long)(iy * W + ix) * in_pitch], pe_w[(ky * 3 + kx) * heads * dv + c], pe);
      }
    }
    out[((long)n * HW + q) * out_pitch + c] = (T)(acc + pe);
  }
}

void launch_psa_attention(int prec, const View& qkv, const View& out, const float* pe_w, const float* pe_b, int heads, int dk, int dv,
                          float scale, int N, hipStream_t st) {
  const int HW = qkv.H * qkv.W;
  LP_CHECK(qkv.C >= heads * (2 * dk + dv) && out.C >= heads * dv && out.H == qkv.H && out.W == qkv.W, LP_ERR_STATE, "attention: view mismatch");
  const size_t lds = (size_t)ATT_QB * (dk + HW) * 4;
  LP_CHECK(lds <= 150 * 1024, LP_ERR_GRAPH, "attention over %d positions does not fit LDS", HW);
  dim3 grid(N, heads, (HW + ATT_QB - 1) / ATT_QB);
  if (prec == LP_FP16) {
    set_max_dynamic_lds(reinterpret_cast<const void*>(psa_attention_kernel<half_t>), 160 * 1024);
    LP_LAUNCH(psa_attention_kernel<half_t>, grid, dim3(256), lds, st, (const half_t*)qkv.base, (half_t*)out.base, pe_w, pe_b, qkv.H,
                       qkv.W, heads, dk, dv, scale, qkv.pitch, out.pitch);
  } else {
    set_max_dynamic_lds(reinterpret_cast<const void*>(psa_attention_kernel<float>), 160 * 1024);
    LP_LAUNCH(psa_attention_kernel<float>, grid, dim3(256), lds, st, (const float*)qkv.base, (float*)out.base, pe_w, pe_b, qkv.H,
                       qkv.W, heads, dk, dv, scale, qkv.pitch, out.pitch);
  }
  LP_HIP(hipGetLastError());
}

}  // namespace lp
