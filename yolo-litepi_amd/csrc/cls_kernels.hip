// ShuffleNetV2 classifier kernels other than the pointwise convs (those are the MFMA 1x1
// kernel of conv_kernels.hip).  Replaces the torch-CPU ops behind self.model(batch) and the
// ToTensor/Normalize/softmax/argmax glue of PyTorchClassifier.predict_batch (reference
// e2e.py:366-370,391-396).  The ROI count R lives in device memory (m_dyn): every kernel is
// a grid-stride loop bounded by it, so the whole pipeline runs without a host round trip.
#include "common.h"
#include "kernels.h"

namespace lp {

typedef _Float16 half_t;
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

template <typename T> struct VecT;
template <> struct VecT<half_t> { typedef half8 type; static constexpr int G = 8; };
template <> struct VecT<float> { typedef floatx4 type; static constexpr int G = 4; };

static inline unsigned grid_for(long max_work) {
  long b = (max_work + 255) / 256;
  if (b < 1) b = 1;
  return (unsigned)(b > 2048 ? 2048 : b);
}

// ------------------------------------------------------------------------------------
// conv1: 3x3 stride 2 pad 1, 3 -> CO, on t = (x/255 - 0.18)/0.34 (ToTensor + Normalize,
// e2e.py:368-369); zero padding applies to t.  One thread per output pixel.
// ------------------------------------------------------------------------------------
template <typename T, int CO>
__global__ __launch_bounds__(256) void cls_stem_kernel(const uint8_t* __restrict__ rgb, T* __restrict__ out,
                                                       const float* __restrict__ w, const float* __restrict__ bias, int S,
                                                       int out_pitch, const int* __restrict__ m_dyn) {
  const int So = S / 2;
  const long total = (long)(*m_dyn) * So * So;
  for (long pix = (long)blockIdx.x * 256 + threadIdx.x; pix < total; pix += (long)gridDim.x * 256) {
    const int ox = (int)(pix % So);
    const int oy = (int)((pix / So) % So);
    const long r = pix / ((long)So * So);
    float acc[CO];
#pragma unroll
    for (int c = 0; c < CO; ++c) acc[c] = 0.f;
    for (int ky = 0; ky < 3; ++ky) {
      const int iy = oy * 2 - 1 + ky;
      if (iy < 0 || iy >= S) continue;
      for (int kx = 0; kx < 3; ++kx) {
        const int ix = ox * 2 - 1 + kx;
        if (ix < 0 || ix >= S) continue;
        const uint8_t* px = rgb + ((r * S + iy) * S + ix) * 3;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const float t = __fdiv_rn(__fsub_rn(__fdiv_rn((float)px[c], 255.f), 0.18f), 0.34f);
          const float* wr = w + ((ky * 3 + kx) * 3 + c) * CO;
#pragma unroll
          for (int co = 0; co < CO; ++co) acc[co] = fmaf(t, wr[co], acc[co]);
        }
      }
    }
    T* o = out + pix * out_pitch;
#pragma unroll
    for (int co = 0; co < CO; ++co) o[co] = (T)fmaxf(acc[co] + bias[co], 0.f);
  }
}

void launch_cls_stem(int prec, const uint8_t* rgb, const float* w, const float* bias, int CO, const View& out, int S,
                     const int* m_dyn, int max_items, hipStream_t st) {
  LP_CHECK(CO == 24, LP_ERR_STATE, "classifier stem expects 24 output channels");
  dim3 grid(grid_for((long)max_items * (S / 2) * (S / 2)));
  if (prec == LP_FP16)
    LP_LAUNCH((cls_stem_kernel<half_t, 24>), grid, dim3(256), 0, st, rgb, (half_t*)out.base, w, bias, S, out.pitch, m_dyn);
  else
    LP_LAUNCH((cls_stem_kernel<float, 24>), grid, dim3(256), 0, st, rgb, (float*)out.base, w, bias, S, out.pitch, m_dyn);
  LP_HIP(hipGetLastError());
}

// ------------------------------------------------------------------------------------
// ResNet18 conv1: 7x7 stride 2 pad 3, 3 -> 64, + folded BN + ReLU, on t = (x/255 - 0.18)/0.34 (torchvision resnet.py;
// e2e.py:320-323 build_classifier('resnet18'), transform e2e.py:366-370); zero padding applies to t.
// One workgroup per (ROI, output row): the 7 input rows (normalised, fp32, padded by 3 columns) and the 147 x 64 weights
// live in LDS; a thread owns one output pixel x 8 channels.
// ------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void cls_stem7_kernel(const uint8_t* __restrict__ rgb, T* __restrict__ out,
                                                        const float* __restrict__ w /*[147][64]*/, const float* __restrict__ bias,
                                                        int S, int out_pitch, const int* __restrict__ m_dyn) {
  extern __shared__ __attribute__((aligned(16))) char smem7[];
  float* lw = reinterpret_cast<float*>(smem7);          // [147][64]
  float* rows = lw + 147 * 64;                           // [7][(S + 6) * 3]
  const int So = S / 2, RW = (S + 6) * 3;
  const int R = *m_dyn;
  const int tid = threadIdx.x;
  for (int i = tid; i < 147 * 64; i += 256) lw[i] = w[i];
  const int ox = tid >> 3, cg = tid & 7;   // 32 output pixels x 8 channel groups (S = 64)
  float b8[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) b8[c] = bias[cg * 8 + c];
  for (long item = blockIdx.x; item < (long)R * So; item += gridDim.x) {
    const long r = item / So;
    const int oy = (int)(item - r * So);
    __syncthreads();
    for (int i = tid; i < 7 * RW; i += 256) {
      const int ky = i / RW, j = i - ky * RW;
      const int ix = j / 3 - 3, c = j - (j / 3) * 3;
      const int iy = oy * 2 - 3 + ky;
      float t = 0.f;
      if (iy >= 0 && iy < S && ix >= 0 && ix < S)
        t = __fdiv_rn(__fsub_rn(__fdiv_rn((float)rgb[((r * S + iy) * S + ix) * 3 + c], 255.f), 0.18f), 0.34f);
      rows[i] = t;
    }
    __syncthreads();
    if (ox < So) {
      float acc[8];
#pragma unroll
      for (int c = 0; c < 8; ++c) acc[c] = 0.f;
      for (int ky = 0; ky < 7; ++ky) {
        const float* rp = rows + ky * RW + ox * 2 * 3;   // window columns ox*2 - 3 .. ox*2 + 3 -> padded index ox*2 ..
        const float* wp = lw + (ky * 21) * 64 + cg * 8;
#pragma unroll
        for (int j = 0; j < 21; ++j) {
          const float t = rp[j];
          const floatx4 w0 = *reinterpret_cast<const floatx4*>(wp + j * 64);
          const floatx4 w1 = *reinterpret_cast<const floatx4*>(wp + j * 64 + 4);
          acc[0] = fmaf(t, w0[0], acc[0]); acc[1] = fmaf(t, w0[1], acc[1]); acc[2] = fmaf(t, w0[2], acc[2]); acc[3] = fmaf(t, w0[3], acc[3]);
          acc[4] = fmaf(t, w1[0], acc[4]); acc[5] = fmaf(t, w1[1], acc[5]); acc[6] = fmaf(t, w1[2], acc[6]); acc[7] = fmaf(t, w1[3], acc[7]);
        }
      }
      T* o = out + ((r * So + oy) * So + ox) * out_pitch + cg * 8;
#pragma unroll
      for (int c = 0; c < 8; ++c) o[c] = (T)fmaxf(acc[c] + b8[c], 0.f);
    }
  }
}

void launch_cls_stem7(int prec, const uint8_t* rgb, const float* w, const float* bias, const View& out, int S, const int* m_dyn,
                      int max_items, hipStream_t st) {
  LP_CHECK(S == 64 && out.C >= 64, LP_ERR_STATE, "ResNet stem: 64x64 input, 64 output channels");
  const size_t lds = (size_t)147 * 64 * 4 + (size_t)7 * (S + 6) * 3 * 4;
  long items = (long)max_items * (S / 2);
  dim3 grid((unsigned)(items < 1 ? 1 : (items > 4096 ? 4096 : items)));
  if (prec == LP_FP16)
    LP_LAUNCH(cls_stem7_kernel<half_t>, grid, dim3(256), lds, st, rgb, (half_t*)out.base, w, bias, S, out.pitch, m_dyn);
  else
    LP_LAUNCH(cls_stem7_kernel<float>, grid, dim3(256), lds, st, rgb, (float*)out.base, w, bias, S, out.pitch, m_dyn);
  LP_HIP(hipGetLastError());
}

// ------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void maxpool3x3s2_kernel(const T* __restrict__ in, T* __restrict__ out, int H, int W, int CG,
                                                           int in_pitch, int out_pitch, const int* __restrict__ m_dyn) {
  typedef typename VecT<T>::type vec;
  constexpr int G = VecT<T>::G;
  const int Ho = (H + 1) / 2, Wo = (W + 1) / 2;
  const long total = (long)(*m_dyn) * Ho * Wo * CG;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const int cg = (int)(idx % CG);
    const long pix = idx / CG;
    const int ox = (int)(pix % Wo);
    const int oy = (int)((pix / Wo) % Ho);
    const long r = pix / ((long)Wo * Ho);
    vec m;
#pragma unroll
    for (int i = 0; i < G; ++i) m[i] = (T)-INFINITY;
    for (int ky = 0; ky < 3; ++ky) {
      const int iy = oy * 2 - 1 + ky;
      if (iy < 0 || iy >= H) continue;
      for (int kx = 0; kx < 3; ++kx) {
        const int ix = ox * 2 - 1 + kx;
        if (ix < 0 || ix >= W) continue;
        const vec v = *reinterpret_cast<const vec*>(in + ((r * H + iy) * W + ix) * in_pitch + cg * G);
#pragma unroll
        for (int i = 0; i < G; ++i) m[i] = v[i] > m[i] ? v[i] : m[i];
      }
    }
    *reinterpret_cast<vec*>(out + pix * out_pitch + cg * G) = m;
  }
}

void launch_maxpool3x3s2(int prec, const View& in, const View& out, const int* m_dyn, int max_items, hipStream_t st) {
  const int G = prec == LP_FP16 ? 8 : 4;
  const int CG = in.C / G;
  dim3 grid(grid_for((long)max_items * out.H * out.W * CG));
  if (prec == LP_FP16)
    LP_LAUNCH(maxpool3x3s2_kernel<half_t>, grid, dim3(256), 0, st, (const half_t*)in.base, (half_t*)out.base, in.H,
                       in.W, CG, in.pitch, out.pitch, m_dyn);
  else
    LP_LAUNCH(maxpool3x3s2_kernel<float>, grid, dim3(256), 0, st, (const float*)in.base, (float*)out.base, in.H, in.W,
                       CG, in.pitch, out.pitch, m_dyn);
  LP_HIP(hipGetLastError());
}

// ------------------------------------------------------------------------------------
// depthwise 3x3, pad 1, stride 1|2, + bias (folded BN), no activation
// ------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void dwconv3x3_kernel(const T* __restrict__ in, T* __restrict__ out,
                                                        const float* __restrict__ w, const float* __restrict__ bias, int H,
                                                        int W, int Ho, int Wo, int C, int stride, int in_pitch, int out_pitch,
                                                        const int* __restrict__ m_dyn) {
  typedef typename VecT<T>::type vec;
  constexpr int G = VecT<T>::G;
  const int CG = C / G;
  const long total = (long)(*m_dyn) * Ho * Wo * CG;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const int cg = (int)(idx % CG);
    const long pix = idx / CG;
    const int ox = (int)(pix % Wo);
    const int oy = (int)((pix / Wo) % Ho);
    const long r = pix / ((long)Wo * Ho);
    float acc[G];
#pragma unroll
    for (int i = 0; i < G; ++i) acc[i] = bias[cg * G + i];
    for (int ky = 0; ky < 3; ++ky) {
      const int iy = oy * stride - 1 + ky;
      if (iy < 0 || iy >= H) continue;
      for (int kx = 0; kx < 3; ++kx) {
        const int ix = ox * stride - 1 + kx;
        if (ix < 0 || ix >= W) continue;
        const vec v = *reinterpret_cast<const vec*>(in + ((r * H + iy) * W + ix) * in_pitch + cg * G);
        const float* wr = w + (ky * 3 + kx) * C + cg * G;
#pragma unroll
        for (int i = 0; i < G; ++i) acc[i] = fmaf((float)v[i], wr[i], acc[i]);
      }
    }
    vec o;
#pragma unroll
    for (int i = 0; i < G; ++i) o[i] = (T)acc[i];
    *reinterpret_cast<vec*>(out + pix * out_pitch + cg * G) = o;
  }
}

void launch_dwconv3x3(int prec, const View& in, const View& out, const float* w, const float* bias, int stride,
                      const int* m_dyn, int max_items, hipStream_t st) {
  const int G = prec == LP_FP16 ? 8 : 4;
  dim3 grid(grid_for((long)max_items * out.H * out.W * (in.C / G)));
  if (prec == LP_FP16)
    LP_LAUNCH(dwconv3x3_kernel<half_t>, grid, dim3(256), 0, st, (const half_t*)in.base, (half_t*)out.base, w, bias,
                       in.H, in.W, out.H, out.W, in.C, stride, in.pitch, out.pitch, m_dyn);
  else
    LP_LAUNCH(dwconv3x3_kernel<float>, grid, dim3(256), 0, st, (const float*)in.base, (float*)out.base, w, bias, in.H,
                       in.W, out.H, out.W, in.C, stride, in.pitch, out.pitch, m_dyn);
  LP_HIP(hipGetLastError());
}

// ------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void spatial_mean_kernel(const T* __restrict__ in, T* __restrict__ out, int HW, int CG,
                                                           int in_pitch, int out_pitch, const int* __restrict__ m_dyn) {
  typedef typename VecT<T>::type vec;
  constexpr int G = VecT<T>::G;
  const long total = (long)(*m_dyn) * CG;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const int cg = (int)(idx % CG);
    const long r = idx / CG;
    float acc[G];
#pragma unroll
    for (int i = 0; i < G; ++i) acc[i] = 0.f;
    for (int p = 0; p < HW; ++p) {
      const vec v = *reinterpret_cast<const vec*>(in + (r * HW + p) * in_pitch + cg * G);
#pragma unroll
      for (int i = 0; i < G; ++i) acc[i] += (float)v[i];
    }
    vec o;
#pragma unroll
    for (int i = 0; i < G; ++i) o[i] = (T)(acc[i] / (float)HW);
    *reinterpret_cast<vec*>(out + r * out_pitch + cg * G) = o;
  }
}

void launch_spatial_mean(int prec, const View& in, const View& out, const int* m_dyn, int max_items, hipStream_t st) {
  const int G = prec == LP_FP16 ? 8 : 4;
  const int CG = in.C / G;
  dim3 grid(grid_for((long)max_items * CG));
  if (prec == LP_FP16)
    LP_LAUNCH(spatial_mean_kernel<half_t>, grid, dim3(256), 0, st, (const half_t*)in.base, (half_t*)out.base,
                       in.H * in.W, CG, in.pitch, out.pitch, m_dyn);
  else
    LP_LAUNCH(spatial_mean_kernel<float>, grid, dim3(256), 0, st, (const float*)in.base, (float*)out.base,
                       in.H * in.W, CG, in.pitch, out.pitch, m_dyn);
  LP_HIP(hipGetLastError());
}

// ------------------------------------------------------------------------------------
// MobileNetV2 / EfficientNet-B0 pieces (mbnet.cpp; build_classifier('mobilenetv2' | 'efficientnet'), e2e.py:324-329).
// Activation codes as common.h Act, plus 3 = ReLU6 for the kernels of this section.
// ------------------------------------------------------------------------------------
__device__ __forceinline__ float mb_act(float v, int act) {
  if (act == 1) return v * __builtin_amdgcn_rcpf(1.f + __expf(-v));   // SiLU, as Tr<T>::silu
  if (act == 2) return fmaxf(v, 0.f);
  if (act == 3) return fminf(fmaxf(v, 0.f), 6.f);
  return v;
}

// features[0]: 3x3 stride 2 pad 1, 3 -> CO (a multiple of 8), + folded BN + activation, on t = (x/255 - 0.18)/0.34.
// One thread per (output pixel, 8-channel group); w fp32 [27][CO] in (ky, kx, rgb) order.
template <typename T>
__global__ __launch_bounds__(256) void cls_stem_act_kernel(const uint8_t* __restrict__ rgb, T* __restrict__ out, const float* __restrict__ w,
                                                           const float* __restrict__ bias, int S, int CO, int act, int out_pitch,
                                                           const int* __restrict__ m_dyn) {
  const int So = S / 2, CG = CO / 8;
  const long total = (long)(*m_dyn) * So * So * CG;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const int cg = (int)(idx % CG);
    const long pix = idx / CG;
    const int ox = (int)(pix % So);
    const int oy = (int)((pix / So) % So);
    const long r = pix / ((long)So * So);
    float acc[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) acc[c] = 0.f;
    for (int ky = 0; ky < 3; ++ky) {
      const int iy = oy * 2 - 1 + ky;
      if (iy < 0 || iy >= S) continue;
      for (int kx = 0; kx < 3; ++kx) {
        const int ix = ox * 2 - 1 + kx;
        if (ix < 0 || ix >= S) continue;
        const uint8_t* px = rgb + ((r * S + iy) * S + ix) * 3;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const float t = __fdiv_rn(__fsub_rn(__fdiv_rn((float)px[c], 255.f), 0.18f), 0.34f);
          const float* wr = w + ((ky * 3 + kx) * 3 + c) * CO + cg * 8;
#pragma unroll
          for (int co = 0; co < 8; ++co) acc[co] = fmaf(t, wr[co], acc[co]);
        }
      }
    }
    T* o = out + pix * out_pitch + cg * 8;
#pragma unroll
    for (int co = 0; co < 8; ++co) o[co] = (T)mb_act(acc[co] + bias[cg * 8 + co], act);
  }
}

void launch_cls_stem_act(int prec, const uint8_t* rgb, const float* w, const float* bias, int CO, int act, const View& out, int S,
                         const int* m_dyn, int max_items, hipStream_t st) {
  LP_CHECK(CO % 8 == 0, LP_ERR_STATE, "classifier stem: output channels must be a multiple of 8");
  dim3 grid(grid_for((long)max_items * (S / 2) * (S / 2) * (CO / 8)));
  if (prec == LP_FP16)
    LP_LAUNCH(cls_stem_act_kernel<half_t>, grid, dim3(256), 0, st, rgb, (half_t*)out.base, w, bias, S, CO, act, out.pitch, m_dyn);
  else
    LP_LAUNCH(cls_stem_act_kernel<float>, grid, dim3(256), 0, st, rgb, (float*)out.base, w, bias, S, CO, act, out.pitch, m_dyn);
  LP_HIP(hipGetLastError());
}

// depthwise k x k (k = 3 | 5), pad k/2, stride 1|2, + bias (folded BN) + activation; w fp32 [k*k][C]
template <typename T>
__global__ __launch_bounds__(256) void dwconv_act_kernel(const T* __restrict__ in, T* __restrict__ out, const float* __restrict__ w,
                                                         const float* __restrict__ bias, int H, int W, int Ho, int Wo, int C, int k, int stride,
                                                         int act, int in_pitch, int out_pitch, const int* __restrict__ m_dyn) {
  typedef typename VecT<T>::type vec;
  constexpr int G = VecT<T>::G;
  const int CG = C / G, pad = k / 2;
  const long total = (long)(*m_dyn) * Ho * Wo * CG;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const int cg = (int)(idx % CG);
    const long pix = idx / CG;
    const int ox = (int)(pix % Wo);
    const int oy = (int)((pix / Wo) % Ho);
    const long r = pix / ((long)Wo * Ho);
    float acc[G];
#pragma unroll
    for (int i = 0; i < G; ++i) acc[i] = 0.f;
    for (int ky = 0; ky < k; ++ky) {
      const int iy = oy * stride - pad + ky;
      if (iy < 0 || iy >= H) continue;
      for (int kx = 0; kx < k; ++kx) {
        const int ix = ox * stride - pad + kx;
        if (ix < 0 || ix >= W) continue;
        const vec v = *reinterpret_cast<const vec*>(in + ((r * H + iy) * W + ix) * in_pitch + cg * G);
        const float* wr = w + (ky * k + kx) * C + cg * G;
#pragma unroll
        for (int i = 0; i < G; ++i) acc[i] = fmaf((float)v[i], wr[i], acc[i]);
      }
    }
    vec o;
#pragma unroll
    for (int i = 0; i < G; ++i) o[i] = (T)mb_act(acc[i] + bias[cg * G + i], act);
    *reinterpret_cast<vec*>(out + pix * out_pitch + cg * G) = o;
  }
}

void launch_dwconv_act(int prec, const View& in, const View& out, const float* w, const float* bias, int k, int stride, int act,
                       const int* m_dyn, int max_items, hipStream_t st) {
  const int G = prec == LP_FP16 ? 8 : 4;
  dim3 grid(grid_for((long)max_items * out.H * out.W * (in.C / G)));
  if (prec == LP_FP16)
    LP_LAUNCH(dwconv_act_kernel<half_t>, grid, dim3(256), 0, st, (const half_t*)in.base, (half_t*)out.base, w, bias, in.H, in.W,
                       out.H, out.W, in.C, k, stride, act, in.pitch, out.pitch, m_dyn);
  else
    LP_LAUNCH(dwconv_act_kernel<float>, grid, dim3(256), 0, st, (const float*)in.base, (float*)out.base, w, bias, in.H, in.W, out.H,
                       out.W, in.C, k, stride, act, in.pitch, out.pitch, m_dyn);
  LP_HIP(hipGetLastError());
}

// in place: x = min(x, cap) (ReLU6 behind a conv whose epilogue already applied ReLU), or, with scale != null, the
// squeeze-excitation gate x[r, p, c] *= sigmoid(scale[r, c]) (torchvision SqueezeExcitation.forward: scale * input)
template <typename T>
__global__ __launch_bounds__(256) void mb_eltwise_kernel(T* __restrict__ x, const T* __restrict__ scale, int HW, int C, int pitch, int spitch, float cap,
                                                         const int* __restrict__ m_dyn) {
  typedef typename VecT<T>::type vec;
  constexpr int G = VecT<T>::G;
  const int CG = C / G;
  const long total = (long)(*m_dyn) * HW * CG;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const int cg = (int)(idx % CG);
    const long pix = idx / CG;
    vec v = *reinterpret_cast<const vec*>(x + pix * pitch + cg * G);
    if (scale) {
      const long r = pix / HW;
      const vec sv = *reinterpret_cast<const vec*>(scale + r * spitch + cg * G);
#pragma unroll
      for (int i = 0; i < G; ++i) v[i] = (T)((float)v[i] * __builtin_amdgcn_rcpf(1.f + __expf(-(float)sv[i])));
    } else {
#pragma unroll
      for (int i = 0; i < G; ++i) v[i] = (T)fminf((float)v[i], cap);
    }
    *reinterpret_cast<vec*>(x + pix * pitch + cg * G) = v;
  }
}

void launch_mb_eltwise(int prec, const View& x, const View* scale, float cap, const int* m_dyn, int max_items, hipStream_t st) {
  const int G = prec == LP_FP16 ? 8 : 4;
  dim3 grid(grid_for((long)max_items * x.H * x.W * (x.C / G)));
  if (prec == LP_FP16)
    LP_LAUNCH(mb_eltwise_kernel<half_t>, grid, dim3(256), 0, st, (half_t*)x.base, scale ? (const half_t*)scale->base : nullptr, x.H * x.W,
                       x.C, x.pitch, scale ? scale->pitch : 0, cap, m_dyn);
  else
    LP_LAUNCH(mb_eltwise_kernel<float>, grid, dim3(256), 0, st, (float*)x.base, scale ? (const float*)scale->base : nullptr, x.H * x.W, x.C,
                       x.pitch, scale ? scale->pitch : 0, cap, m_dyn);
  LP_HIP(hipGetLastError());
}

// ------------------------------------------------------------------------------------
// softmax(dim=1) + argmax (e2e.py:394-396), one wavefront per ROI (classes strided over lanes,
// wave-level max / sum / arg-max reductions)
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void softmax_argmax_kernel(const float* __restrict__ logits, int pitch, int nc,
                                                             float* __restrict__ probs, int* __restrict__ ids,
                                                             float* __restrict__ conf, lp_det* dets, int max_det,
                                                             const int* __restrict__ roi_img, const int* __restrict__ roi_slot,
                                                             const int* __restrict__ m_dyn) {
  const int R = *m_dyn;
  const int lane = threadIdx.x & 63;
  const int wave = (blockIdx.x * 256 + threadIdx.x) >> 6, nwaves = (gridDim.x * 256) >> 6;
  for (int r = wave; r < R; r += nwaves) {
    const float* l = logits + (long)r * pitch;
    float mx = -INFINITY;
    for (int c = lane; c < nc; c += 64) mx = fmaxf(mx, l[c]);
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    float sum = 0.f;
    for (int c = lane; c < nc; c += 64) sum += expf(l[c] - mx);
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    float best = -1.f;
    int best_c = 0x7fffffff;
    for (int c = lane; c < nc; c += 64) {
      const float p = expf(l[c] - mx) / sum;
      if (probs) probs[(long)r * nc + c] = p;
      if (p > best) { best = p; best_c = c; }  // first maximum within the lane (ascending c)
    }
    for (int o = 32; o > 0; o >>= 1) {  // first maximum across lanes: larger p, then smaller class index
      const float ob = __shfl_xor(best, o);
      const int oc = __shfl_xor(best_c, o);
      if (ob > best || (ob == best && oc < best_c)) { best = ob; best_c = oc; }
    }
    if (lane == 0) {
      if (ids) ids[r] = best_c;
      if (conf) conf[r] = best;
      if (dets) {
        lp_det* d = dets + (long)roi_img[r] * max_det + roi_slot[r];
        d->cls_class = best_c;
        d->cls_conf = best;
      }
    }
  }
}

void launch_softmax_argmax(const float* logits, int pitch, int nc, float* probs, int* ids, float* conf, lp_det* dets,
                           int max_det, const RoiTable* tab, const int* m_dyn, int max_items, hipStream_t st) {
  dim3 grid(grid_for((long)max_items * 64));
  LP_LAUNCH(softmax_argmax_kernel, grid, dim3(256), 0, st, logits, pitch, nc, probs, ids, conf, dets, max_det,
                     tab ? tab->img : nullptr, tab ? tab->slot : nullptr, m_dyn);
  LP_HIP(hipGetLastError());
}

}  // namespace lp
