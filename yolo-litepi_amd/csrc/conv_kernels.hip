// Convolution kernels for gfx950 (CDNA4).  Replaces the NCNN / ONNX Runtime conv
// layers behind ex.extract("out0") (reference src/tt100k/pipeline/e2e.py:305-307;
// graph model.ncnn.param:4-182) and the ShuffleNetV2 pointwise convs behind
// self.model(batch) (e2e.py:393).
//
// Layout: activations are NHWC with a channel pitch, so C2f/SPPF/FPN concats are
// channel slices of one buffer and never materialised.  Orientation of the MFMA:
//     D[out-channel][pixel] = sum_k  W[out-channel][k] * X[k][pixel]
// i.e. A = weights (16 rows = out channels), B = 16 pixels, K = taps x in-channels
// walked in 16-byte groups (8 fp16 / 4 fp32 channels), which are contiguous in NHWC
// for a fixed tap, so every operand fragment is one 16-byte load.  The D fragment
// holds 4 consecutive rows per lane; the weight rows are permuted at pack time so
// those are 4 (x tiles) consecutive physical channels of one pixel -> vector stores.
#include "common.h"
#include "conv.h"
#include <type_traits>

namespace lp {

typedef _Float16 half_t;
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef float floatx2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

enum { EPI_PLAIN = 0, EPI_SHUFFLE = 1 };

// diagnostic phase stamps (never enabled on the product path: a.stamps is null)
#define LP_STAMP(k)                                                                                   \
  if (a.stamps && threadIdx.x == 0)                                                                   \
    a.stamps[((size_t)(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 16 + (k)] = \
        (k) == 0 ? wall_clock64() : clock64();

template <typename T> struct Tr;
template <> struct Tr<half_t> {
  static constexpr int G = 8;  // channels per 16-byte K group
  typedef half8 frag;
  typedef half4 quad;
  static __device__ __forceinline__ floatx4 mma(frag a, frag b, floatx4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
  }
  // x * sigmoid(x) with the hardware reciprocal (1 ulp): an IEEE divide costs ~10 VALU ops per element
  static __device__ __forceinline__ float silu(float v) { return v * __builtin_amdgcn_rcpf(1.f + __expf(-v)); }
};
template <> struct Tr<float> {
  static constexpr int G = 4;
  typedef floatx4 frag;
  typedef floatx4 quad;
  // exact-f32 MFMA (one K=4 instruction per element of the 16-byte group)
  static __device__ __forceinline__ floatx4 mma(frag a, frag b, floatx4 c) {
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0], b[0], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[1], b[1], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[2], b[2], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[3], b[3], c, 0, 0, 0);
    return c;
  }
  static __device__ __forceinline__ float silu(float v) { return v * __builtin_amdgcn_rcpf(1.f + __expf(-v)); }
};

template <typename T> __device__ __forceinline__ float activate(float v, int act) {
  if (act == ACT_SILU) return Tr<T>::silu(v);
  if (act == ACT_RELU) return fmaxf(v, 0.f);
  return v;
}

template <typename T> __device__ __forceinline__ typename Tr<T>::frag as_frag(u32x4 v) {
  return __builtin_bit_cast(typename Tr<T>::frag, v);
}

// One lane's 4 consecutive output channels of one pixel: bias, activation, residual,
// store.  ch0 is the first physical channel of the quad.
template <typename T, int ACT> __device__ __forceinline__ float activate_ct(float v) {
  if (ACT == ACT_SILU) return Tr<T>::silu(v);
  if (ACT == ACT_RELU) return fmaxf(v, 0.f);
  return v;
}
// act(v + b) of a lane's four accumulator values.  SiLU is written on two-wide vectors: the bias add, both multiplies and
// the +1 become v_pk_*_f32 (two elements per issue), the exponential and the reciprocal stay the hardware transcendentals --
// 4.5 VALU issues per element instead of 7.  The conv epilogues are the largest VALU consumer of the narrow layers (the
// VALU, not the matrix pipe, is what the pipelined step is bound by: profiles/README.md), so this is where it counts.
template <typename T, int ACT> __device__ __forceinline__ floatx4 act4(floatx4 v, floatx4 b) {
  v = v + b;
  if (ACT == ACT_SILU) {
    // (-log2 e in a scalar register the compiler cannot see through: as a literal it turns the multiply into four
    //  v_mul_f32 -- VOP3P has no literal operand --, from an SGPR pair it is two v_pk_mul_f32; same product, bit for bit)
    float nl2e = -1.4426950408889634f;
    asm("" : "+s"(nl2e));
    const floatx4 t = v * floatx4{nl2e, nl2e, nl2e, nl2e};
    floatx4 e;
#pragma unroll
    for (int i = 0; i < 4; ++i) e[i] = __builtin_amdgcn_exp2f(t[i]);
    e = e + 1.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) e[i] = __builtin_amdgcn_rcpf(e[i]);
    return v * e;
  }
  if (ACT == ACT_RELU) {
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = fmaxf(v[i], 0.f);
  }
  return v;
}

// Element offset of pixel `pix` in a tensor of channel pitch `pitch`: ONE 32-bit multiply.  The 64-bit product the plain
// pointer arithmetic asks for is three quarter-rate integer multiplies per address -- with one address per pixel per tensor
// that was a third of the narrow layers' epilogue issue time.  Every launch checks N*H*W*pitch < 2^32 on the host (span_ok).
__device__ __forceinline__ unsigned eoff(long pix, int pitch) { return (unsigned)pix * (unsigned)pitch; }
// x / d for x < 2^31 with m = floor(2^32 / d) (host: div_magic): the estimate is q or q - 1, one correction
__device__ __forceinline__ int fast_div(int x, int d, unsigned m) {
  int q = (int)__umulhi((unsigned)x, m);
  if (x - q * d >= d) ++q;
  return q;
}
// The activation is a template parameter: a run-time switch here costs three scalar branches
// per output element (hundreds per wave), more than the MFMAs of a small-K layer.
template <typename T, int EPI, int ACT>
__device__ __forceinline__ void store_quad(const ConvArgs& a, long pix, int ch0, floatx4 v) {
  const floatx4 b = *reinterpret_cast<const floatx4*>(a.bias + ch0);
  v = act4<T, ACT>(v, b);
  if (EPI == EPI_SHUFFLE) {
    // ShuffleNetV2 channel_shuffle(cat(x1, y), 2) fused into the store: logical output channel 2c = x1[c],
    // 2c+1 = y[c]; each half of the output is padded to half_cp.  This lane's 4 channels c..c+3 become the 8
    // consecutive logical channels 2c..2c+7: one vector load of x1, then 4-byte (x1, y) pairs -- merged into one
    // 16-byte store when the run stays inside one half and is 16-byte aligned (scalar 2-byte traffic made the
    // stride-2 blocks' pointwise convs the slowest launches of the classifier).
    const T* x1 = reinterpret_cast<const T*>(a.x1) + eoff(pix, a.x1_pitch);
    T* o = reinterpret_cast<T*>(a.out) + eoff(pix, a.out_pitch);
    if (ch0 + 4 <= a.half_c) {
      const typename Tr<T>::quad xv = *reinterpret_cast<const typename Tr<T>::quad*>(x1 + ch0);
      const int l0 = 2 * ch0;
      const int p0 = l0 < a.half_c ? l0 : a.half_cp + (l0 - a.half_c);
      const bool one_half = l0 + 8 <= a.half_c || l0 >= a.half_c;
      if (sizeof(T) == 2 && one_half && (p0 & 7) == 0) {
        half8 w;
#pragma unroll
        for (int i = 0; i < 4; ++i) { w[2 * i] = (half_t)xv[i]; w[2 * i + 1] = (half_t)v[i]; }
        *reinterpret_cast<half8*>(o + p0) = w;
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int l = l0 + 2 * i;
          const int phys = l < a.half_c ? l : a.half_cp + (l - a.half_c);  // l is even and so is half_c: a pair never straddles
          T pr[2] = {xv[i], (T)v[i]};
          if (sizeof(T) == 2) *reinterpret_cast<uint32_t*>(o + phys) = *reinterpret_cast<const uint32_t*>(pr);
          else { o[phys] = pr[0]; o[phys + 1] = pr[1]; }
        }
      }
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int c = ch0 + i;
        if (c < a.half_c) {
          const int l = 2 * c;
          const int phys = l < a.half_c ? l : a.half_cp + (l - a.half_c);
          o[phys] = x1[c];
          o[phys + 1] = (T)v[i];
        }
      }
    }
    return;
  }
  if (a.res) {
    const typename Tr<T>::quad r =
        *reinterpret_cast<const typename Tr<T>::quad*>(reinterpret_cast<const T*>(a.res) + eoff(pix, a.res_pitch) + ch0);
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] += (float)r[i];
  }
  if (a.out_f32) {
    *reinterpret_cast<floatx4*>(reinterpret_cast<float*>(a.out) + eoff(pix, a.out_pitch) + ch0) = v;
  } else {
    typename Tr<T>::quad q;
#pragma unroll
    for (int i = 0; i < 4; ++i) q[i] = (T)v[i];
    *reinterpret_cast<typename Tr<T>::quad*>(reinterpret_cast<T*>(a.out) + eoff(pix, a.out_pitch) + ch0) = q;
  }
}

// One lane's 4*NT consecutive output channels of one pixel (EPI_PLAIN, T output): bias comes from
// registers (hoisted out of the epilogue: 20 dependent global loads per lane otherwise), fp16
// results leave as 16-byte stores (two channel quads at a time).
template <typename T, int NT, int ACT>
__device__ __forceinline__ void store_lane_at(T* o, const T* r, int chbase, int Cout, const floatx4 (&v)[NT], const floatx4 (&bias)[NT],
                                              bool res_first = false, const typename Tr<T>::quad* rq = nullptr);

template <typename T, int NT, int ACT>
__device__ __forceinline__ void store_lane(const ConvArgs& a, long pix, int chbase, const floatx4 (&v)[NT],
                                           const floatx4 (&bias)[NT]) {
  T* o = reinterpret_cast<T*>(a.out) + eoff(pix, a.out_pitch) + chbase;
  const T* r = a.res ? reinterpret_cast<const T*>(a.res) + eoff(pix, a.res_pitch) + chbase : nullptr;
  store_lane_at<T, NT, ACT>(o, r, chbase, a.Cout, v, bias, a.res_first != 0);
}

// o / r already point at this lane's first channel of the pixel
template <typename T, int NT, int ACT>
__device__ __forceinline__ void store_lane_at(T* o, const T* r, int chbase, int Cout, const floatx4 (&v)[NT], const floatx4 (&bias)[NT],
                                              bool res_first, const typename Tr<T>::quad* rq) {
  // rq != nullptr: the residual quads were loaded by the caller (registers, rq[t] = channels chbase + 4t ..); r only says "there is one"
  if (res_first && r) {
    // ResNet BasicBlock: out = act(conv + bias + identity) (torchvision resnet.py BasicBlock.forward), the sum in fp32
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      if (chbase + t * 4 < Cout) {
        const typename Tr<T>::quad rr = rq ? rq[t] : *reinterpret_cast<const typename Tr<T>::quad*>(r + t * 4);
        const floatx4 y = act4<T, ACT>(v[t] + floatx4{(float)rr[0], (float)rr[1], (float)rr[2], (float)rr[3]}, bias[t]);
        typename Tr<T>::quad q;
#pragma unroll
        for (int i = 0; i < 4; ++i) q[i] = (T)y[i];
        *reinterpret_cast<typename Tr<T>::quad*>(o + t * 4) = q;
      }
    }
    return;
  }
  if constexpr (sizeof(T) == 2) {
#pragma unroll
    for (int t = 0; t + 1 < NT; t += 2) {
      if (chbase + t * 4 < Cout) {
        half8 q;
        const floatx4 y0 = act4<T, ACT>(v[t], bias[t]), y1 = act4<T, ACT>(v[t + 1], bias[t + 1]);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          q[i] = (half_t)y0[i];
          q[4 + i] = (half_t)y1[i];
        }
        if (r) {
          half8 rr;
          if (rq) {
#pragma unroll
            for (int i = 0; i < 4; ++i) { rr[i] = rq[t][i]; rr[4 + i] = rq[t + 1][i]; }
          } else {
            rr = *reinterpret_cast<const half8*>(r + t * 4);
          }
#pragma unroll
          for (int i = 0; i < 8; ++i) q[i] = (half_t)((float)q[i] + (float)rr[i]);
        }
        *reinterpret_cast<half8*>(o + t * 4) = q;
      }
    }
    if (NT & 1) {
      constexpr int t = NT - 1;
      if (chbase + t * 4 < Cout) {
        half4 q;
        const floatx4 y0 = act4<T, ACT>(v[t], bias[t]);
#pragma unroll
        for (int i = 0; i < 4; ++i) q[i] = (half_t)y0[i];
        if (r) {
          const half4 rr = rq ? rq[t] : *reinterpret_cast<const half4*>(r + t * 4);
#pragma unroll
          for (int i = 0; i < 4; ++i) q[i] = (half_t)((float)q[i] + (float)rr[i]);
        }
        *reinterpret_cast<half4*>(o + t * 4) = q;
      }
    }
  } else {
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      if (chbase + t * 4 < Cout) {
        floatx4 q = act4<T, ACT>(v[t], bias[t]);
        if (r) {
          const floatx4 rr = rq ? rq[t] : *reinterpret_cast<const floatx4*>(r + t * 4);
#pragma unroll
          for (int i = 0; i < 4; ++i) q[i] += rr[i];
        }
        *reinterpret_cast<floatx4*>(o + t * 4) = q;
      }
    }
  }
}

// Fused 1x1 "tail": out2 = act2(W2 . act1(acc + bias1) + bias2) for one pixel tile, straight from the
// accumulators.  A lane holds 4*NT consecutive intermediate channels of its pixel -- exactly the B
// operand of the next MFMA (K slice (s2, g, j) <-> channel g*4NT + G*s2 + j), so no LDS round trip; the
// intermediate is rounded to T first, as if it had been stored and re-loaded.
template <typename T> struct TailSteps;
template <> struct TailSteps<half_t> { static constexpr int per_nt(int nt) { return (nt + 1) / 2; } };  // odd NT: last K step half filled
template <> struct TailSteps<float> { static constexpr int per_nt(int nt) { return nt; } };

template <typename T, int NT, int T2, int ACT1>
__device__ __forceinline__ void tail_store(const ConvArgs& a2, long pix, int g, const floatx4 (&v)[NT], const floatx4 (&bias1)[NT],
                                           const typename Tr<T>::frag (&w2f)[T2][TailSteps<T>::per_nt(NT)],
                                           const floatx4 (&bias2)[T2]) {
  constexpr int S2 = TailSteps<T>::per_nt(NT);
  typename Tr<T>::frag bq[S2];
  if constexpr (sizeof(T) == 2) {
#pragma unroll
    for (int s = 0; s < S2; ++s) {
      const int t1 = (2 * s + 1 < NT) ? 2 * s + 1 : 0;  // clamped: the odd-NT last step has no second quad
      const floatx4 y0 = act4<T, ACT1>(v[2 * s], bias1[2 * s]);
      floatx4 y1 = floatx4{0.f, 0.f, 0.f, 0.f};
      if (2 * s + 1 < NT) y1 = act4<T, ACT1>(v[t1], bias1[t1]);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        bq[s][i] = (half_t)y0[i];
        bq[s][4 + i] = (half_t)y1[i];
      }
    }
  } else {
#pragma unroll
    for (int s = 0; s < S2; ++s) bq[s] = act4<T, ACT1>(v[s], bias1[s]);
  }
  floatx4 o[T2];
#pragma unroll
  for (int t = 0; t < T2; ++t) {
    o[t] = floatx4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < S2; ++s) o[t] = Tr<T>::mma(w2f[t][s], bq[s], o[t]);
  }
  if (a2.act == ACT_SILU) store_lane<T, T2, ACT_SILU>(a2, pix, g * 4 * T2, o, bias2);
  else if (a2.act == ACT_RELU) store_lane<T, T2, ACT_RELU>(a2, pix, g * 4 * T2, o, bias2);
  else store_lane<T, T2, ACT_NONE>(a2, pix, g * 4 * T2, o, bias2);
}

// loads of the tail's operands, once per wave
template <typename T, int NT, int T2>
__device__ __forceinline__ void tail_load(const ConvArgs& a, int lane, int g, typename Tr<T>::frag (&w2f)[T2][TailSteps<T>::per_nt(NT)],
                                          floatx4 (&bias2)[T2]) {
  constexpr int S2 = TailSteps<T>::per_nt(NT);
  const u32x4* w = reinterpret_cast<const u32x4*>(a.w2);
#pragma unroll
  for (int t = 0; t < T2; ++t) {
#pragma unroll
    for (int s = 0; s < S2; ++s) w2f[t][s] = as_frag<T>(w[(t * S2 + s) * 64 + lane]);
    bias2[t] = *reinterpret_cast<const floatx4*>(a.bias2 + g * 4 * T2 + t * 4);
  }
}

// the residual is added AFTER the activation (C2f bottleneck: x + silu(conv(..))); the fp16 variant
// above rounds the activated value to fp16 before the add, as a separate add kernel would
template <typename T, int NT, int ACT>
__device__ __forceinline__ void epilogue_tile(const ConvArgs& a, const floatx4 (&acc)[NT][5], const floatx4 (&bias)[NT], int n,
                                              int ns, int g, int oy, int oxb) {
  if (oy >= a.Hout) return;
  // the five patches of a strip are 4 pixels apart on one row: one address computation, then fixed strides
  const int chbase = ns * 16 * NT + g * 4 * NT;
  const int pix0 = (n * a.Hout + oy) * a.Wout + oxb;
  T* o = reinterpret_cast<T*>(a.out) + eoff(pix0, a.out_pitch) + chbase;
  const T* r = a.res ? reinterpret_cast<const T*>(a.res) + eoff(pix0, a.res_pitch) + chbase : nullptr;
  const int ostep = 4 * a.out_pitch, rstep = 4 * a.res_pitch;
  if (r) {
    // the residual quads of all five patches are requested before the first is used (unconditional, from clamped addresses:
    // patch 0 / quad 0 stand in for what lies outside): as `if (r) load` inside the patch loop they were five dependent
    // memory round trips in a kernel whose grids are a single round of workgroups
    typename Tr<T>::quad rq[5][NT];
#pragma unroll
    for (int p = 0; p < 5; ++p) {
      const T* rp = r - chbase + (oxb + p * 4 < a.Wout ? p * rstep : 0);   // channel 0 of the patch's pixel
#pragma unroll
      for (int t = 0; t < NT; ++t) rq[p][t] = *reinterpret_cast<const typename Tr<T>::quad*>(rp + (chbase + t * 4 < a.Cout ? chbase + t * 4 : 0));
    }
#pragma unroll
    for (int p = 0; p < 5; ++p) {
      if (oxb + p * 4 < a.Wout) {
        floatx4 v[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) v[t] = acc[t][p];
        store_lane_at<T, NT, ACT>(o + p * ostep, r, chbase, a.Cout, v, bias, a.res_first != 0, rq[p]);
      }
    }
    return;
  }
#pragma unroll
  for (int p = 0; p < 5; ++p) {
    if (oxb + p * 4 < a.Wout) {
      floatx4 v[NT];
#pragma unroll
      for (int t = 0; t < NT; ++t) v[t] = acc[t][p];
      store_lane_at<T, NT, ACT>(o + p * ostep, nullptr, chbase, a.Cout, v, bias, a.res_first != 0);
    }
  }
}

template <typename T, int NT, int NP, int EPI, int ACT>
__device__ __forceinline__ void epilogue_flat(const ConvArgs& a, const floatx4 (&acc)[NT][NP], const floatx4 (&bias)[NT],
                                              long pix0, long M, int ns, int g, int col) {
#pragma unroll
  for (int p = 0; p < NP; ++p) {
    const long pix = pix0 + p * 16 + col;
    if (pix < M) {
      if (EPI == EPI_PLAIN && !a.out_f32) {
        floatx4 v[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) v[t] = acc[t][p];
        store_lane<T, NT, ACT>(a, pix, ns * 16 * NT + g * 4 * NT, v, bias);
      } else {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          const int ch0 = ns * 16 * NT + g * 4 * NT + t * 4;
          if (ch0 < a.Cout) store_quad<T, EPI, ACT>(a, pix, ch0, acc[t][p]);
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------
// 3x3 (pad 1, stride 1|2) implicit GEMM.  One workgroup = bwh x bww waves; each wave owns
// a 4-row x 20-column strip of output pixels = five 4x4 patches (one MFMA column tile
// each) x NT 16-channel tiles.  20 divides every level of a 640 input (160/80/40/20), so
// no lane is wasted there; other sizes are masked.  Per K chunk (CK input channels) the
// block stages the halo'd input tile and the chunk's weight fragments in LDS.
// LDS image of the input: [IH][LW] pixels x PS bytes, PS and LW chosen by lds_pixel_slots /
// lds_row_width so that a B-fragment ds_read_b128 (4x4 pixels x 4 K groups) is conflict-free.
// ------------------------------------------------------------------------------------
// second launch-bound = waves per SIMD the register allocation must allow: the narrow variants run many
// short-lived workgroups and live off occupancy (4 waves/SIMD = 128 VGPRs), the wide ones off MFMA tiles
// 16-byte LDS-DMA: lane l's 16 bytes land at lds + 16*l (wave-uniform lds), no VGPR in between
#define LP_GLDS16(gptr, lptr)                                                                        \
  __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(gptr),          \
                                   (void __attribute__((address_space(3)))*)(lptr), 16, 0, 0)

template <typename T, int NT, int STRIDE, int T2 = 0>
__global__ __launch_bounds__(320, (NT <= 2 ? 4 : ((NT == 4 && (T2 > 0 || sizeof(T) == 4)) ? 2 : 3))) void conv3x3_mfma_kernel(const ConvArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int G = Tr<T>::G;
  const int tid = threadIdx.x, nthr = blockDim.x;
  const int lane = tid & 63, wave = tid >> 6, nw = nthr >> 6;
  const int g = lane >> 4, col = lane & 15;
  // prologue arithmetic is a large part of a small layer's workgroup life (hundreds of instructions per integer
  // division): every division here is a multiply by a host-checked 16-bit reciprocal (ConvLayer::launch)
  const int wy = a.bww == 2 ? wave >> 1 : wave, wx = wave - wy * a.bww;
  const int TH = 4 * a.bwh, TW = 20 * a.bww;
  // grid = (image, tile, channel split): consecutive workgroup ids round-robin over the 8 XCDs, so with the image
  // index fastest all tiles of an image (which share halo rows and columns) meet in one XCD's L2
  const int tile_id = a.tile_major ? blockIdx.x : blockIdx.y;
  const int ty = (int)((tile_id * a.rcp_tx) >> 16), tx = tile_id - ty * a.tiles_x;
  const int n = a.tile_major ? blockIdx.y : blockIdx.x, ns = blockIdx.z;
  if (a.m_dyn && n >= *a.m_dyn) return;   // classifier layers: the grid is sized for the ROI capacity (block-uniform exit)
  const int oy0 = ty * TH, ox0 = tx * TW;
  const int iy0 = oy0 * STRIDE - 1, ix0 = ox0 * STRIDE - 1;
  const int IH = (TH - 1) * STRIDE + 3, IW = (TW - 1) * STRIDE + 3;
  const int Sc = a.steps_per_chunk, CGc = a.CGc, LW = a.LW, PS = a.PS;

  // LDS: [tap-offset table 512 B] then per buffer (two when the layer has several K chunks):
  //      [weight fragments Sc*NT KB][input tile IH x LW x PS]
  int* lds_toff = reinterpret_cast<int*>(smem);
  const int wbytes = Sc * NT * 1024;
  const int RS = LW * (PS >> 4);           // 16-byte slots per tile row (pads included)
  const int bufbytes = wbytes + IH * RS * 16;
  for (int q = tid; q < Sc * 4; q += nthr) {  // byte offset of K group q inside the input tile: (tap, channel group)
    int tap = (int)(((unsigned)q * a.rcp_cg) >> 16);
    const int cg = q - tap * CGc;
    tap = tap > 8 ? 8 : tap;  // K padding slots: weights are zero, read any finite data
    const int ky = (tap * 21846) >> 16, kx = tap - 3 * ky;
    lds_toff[q] = (ky * LW + kx) * PS + cg * 16;
  }

  const int ly = wy * 4 + (col >> 2);
  int pbase[5];
#pragma unroll
  for (int p = 0; p < 5; ++p) {
    const int lx = wx * 20 + p * 4 + (col & 3);
    pbase[p] = ((ly * STRIDE) * LW + lx * STRIDE) * PS;
  }
  // Staging map.  The LDS image of a tile row is RS consecutive 16-byte slots = pixel pitch x LW, filled by
  // 1 KiB LDS-DMA pieces (64 lanes x 16 B, destination lane-linear).  The (row, piece) items of a chunk are dealt
  // round-robin to the waves; a lane's slot in a piece is a (column, channel group): slots that are pitch padding,
  // beyond the tile, or outside the image read a 16-byte zero line instead (= zero padding).  ~20 instructions per
  // item, wave-uniform parts on the SALU (VALU issue slots are the scarce resource of the narrow layers).
  const int pcs = (RS + 63) >> 6;
  const int PSs = PS >> 4;
  const int wave_s = __builtin_amdgcn_readfirstlane(wave);
  const char* zeros = reinterpret_cast<const char*>(a.zeros);
  const char* in_b = reinterpret_cast<const char*>(a.in);

  // issue every load of K chunk c into buffer b (nothing waits here)
  auto issue = [&](int c, int b) {
    char* buf = smem + 512 + b * bufbytes;
    const u32x4* wsrc = reinterpret_cast<const u32x4*>(a.wpk) + ((size_t)(ns * a.nchunks + c) * Sc * NT) * 64;
    for (int p = wave_s; p < Sc * NT; p += nw) LP_GLDS16(wsrc + p * 64 + lane, buf + p * 1024);
    char* lin = buf + wbytes;
    const int cbase = c * a.CK;
    const int nitems = IH * pcs;
    for (int it = wave_s; it < nitems; it += nw) {
      const int iy = (int)(((unsigned)it * a.rcp_pcs) >> 16), pc = it - iy * pcs;
      const int gy = iy0 + iy;
      const char* rowp = in_b + (((long)(n * a.Hin + gy) * a.Win + ix0) * a.in_pitch + cbase) * (long)sizeof(T);
      const int sl = pc * 64 + lane;
      const int ix = (int)(((unsigned)sl * a.rcp_ps) >> 16), cgs = sl - ix * PSs;
      const int gx = ix0 + ix;
      const bool ok = gy >= 0 && gy < a.Hin && gx >= 0 && gx < a.Win && ix < IW && cgs < CGc;
      const char* src = ok ? rowp + (ix * a.in_pitch + cgs * G) * (int)sizeof(T) : zeros;
      if (sl < RS) LP_GLDS16(src, lin + (iy * RS + pc * 64) * 16);
    }
  };

  floatx4 bias_r[NT];  // this lane's 4*NT output channels: bias hoisted out of the epilogue
#pragma unroll
  for (int t = 0; t < NT; ++t) bias_r[t] = *reinterpret_cast<const floatx4*>(a.bias + ns * 16 * NT + g * 4 * NT + t * 4);
  floatx4 acc[NT][5];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int p = 0; p < 5; ++p) acc[t][p] = floatx4{0.f, 0.f, 0.f, 0.f};

  LP_STAMP(0)
  LP_STAMP(1)
  issue(0, 0);
  for (int chunk = 0; chunk < a.nchunks; ++chunk) {
    if (chunk < 2) { LP_STAMP(2 + chunk * 4) }
    // chunk's loads have landed (own: vmcnt, everyone's: barrier); the barrier also says every wave is done
    // reading the other buffer, so the next chunk's loads go out now and fly during this chunk's MFMAs
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (chunk < 2) { LP_STAMP(3 + chunk * 4) }
    if (chunk + 1 < a.nchunks) issue(chunk + 1, (chunk + 1) & 1);
    if (chunk < 2) { LP_STAMP(4 + chunk * 4) }
    const u32x4* lds_w = reinterpret_cast<const u32x4*>(smem + 512 + (chunk & 1) * bufbytes);
    const char* lds_in = smem + 512 + (chunk & 1) * bufbytes + wbytes;
    for (int s = 0; s < Sc; ++s) {
      const int toff = lds_toff[4 * s + g];
      typename Tr<T>::frag af[NT], bf[5];
#pragma unroll
      for (int t = 0; t < NT; ++t) af[t] = as_frag<T>(lds_w[(s * NT + t) * 64 + lane]);
#pragma unroll
      for (int p = 0; p < 5; ++p) bf[p] = as_frag<T>(*reinterpret_cast<const u32x4*>(lds_in + pbase[p] + toff));
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int p = 0; p < 5; ++p) acc[t][p] = Tr<T>::mma(af[t], bf[p], acc[t][p]);
    }
    if (chunk < 2) { LP_STAMP(5 + chunk * 4) }
  }
  LP_STAMP(10)

  const int oy = oy0 + ly;
  const int oxb = ox0 + wx * 20 + (col & 3);
  if constexpr (T2 > 0) {
    // fused 1x1 conv on the accumulator tile (needs every intermediate channel in this workgroup: ns == 0)
    typename Tr<T>::frag w2f[T2][TailSteps<T>::per_nt(NT)];
    floatx4 bias2[T2];
    tail_load<T, NT, T2>(a, lane, g, w2f, bias2);
    ConvArgs a2 = a;
    a2.Cout = a.Cout2; a2.act = a.act2; a2.res = nullptr;
#pragma unroll
    for (int p = 0; p < 5; ++p) {
      const int ox = oxb + p * 4;
      if (oy < a.Hout && ox < a.Wout) {
        const long pix = (n * a.Hout + oy) * a.Wout + ox;
        floatx4 v[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) v[t] = acc[t][p];
        if (a.act == ACT_SILU) tail_store<T, NT, T2, ACT_SILU>(a2, pix, g, v, bias_r, w2f, bias2);
        else tail_store<T, NT, T2, ACT_NONE>(a2, pix, g, v, bias_r, w2f, bias2);
      }
    }
  } else {
    if (a.act == ACT_SILU) epilogue_tile<T, NT, ACT_SILU>(a, acc, bias_r, n, ns, g, oy, oxb);
    else if (a.act == ACT_RELU) epilogue_tile<T, NT, ACT_RELU>(a, acc, bias_r, n, ns, g, oy, oxb);
    else epilogue_tile<T, NT, ACT_NONE>(a, acc, bias_r, n, ns, g, oy, oxb);
  }
  LP_STAMP(11)
  if (a.stamps && threadIdx.x == 0) a.stamps[((size_t)(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 16 + 12] = wall_clock64();
}


// ------------------------------------------------------------------------------------
// Fused C2f bottleneck (two 3x3 convs + SiLU + shortcut) in one launch:
//     out = x + silu(conv_b(silu(conv_a(x))))          (reference graph: C2f.m[i], model.ncnn.param)
// One workgroup (4 waves) owns a TH x TW output tile.  It stages the (TH+4) x (TW+4) input tile and BOTH
// weight sets by LDS-DMA, computes conv_a on the (TH+2) x (TW+2) region conv_b needs (zero outside the image:
// that is conv_b's padding), writes it back over the input tile in LDS (rounded to T, exactly what a store +
// reload would give), runs conv_b from there and finishes with the usual bias/SiLU/residual epilogue.  The
// intermediate never leaves the CU: one launch and one HBM round trip less per bottleneck.
// Pixels are mapped linearly (tile q of a wave = 16 consecutive pixels of the region, row-major), so the
// (TH+2) x (TW+2) region costs ceil(/16) MFMA column tiles instead of whole 4x4 patches.
// ------------------------------------------------------------------------------------
// P1 / P2: pixel tiles per wave for conv_a / conv_b (compile time: the K loops are branch-free, a wave's surplus
// tile recomputes the region's last pixel and is never written)
// SEP: the intermediate gets its own LDS region instead of overwriting the input tile, so the shortcut operand x is
// taken from LDS too (narrow layers on big maps are HBM-bound: this drops one of the 2.65 input reads per output;
// used when LDS allows, i.e. NT == 1)
// T2 > 0: the C2f's closing 1x1 conv (cv2 over concat[y0, y1, .., y_last]) runs here as well, T2 = its 16-channel
// output tiles.  y_last never leaves the registers (the accumulator tile is already a B operand, as in tail_store);
// the other concat segments are gathered from the concat buffer, 16 B per lane per K step.
// CL ("concat from LDS", round 4): the input x is the last stored segment of that concat buffer, and a pixel's stored
// segments are one contiguous piece of a.kg K groups.  The tile is then staged with ALL of them (pitch a.PSA), conv_a reads
// x at its offset inside the pixel, and cv2 takes its gathered K groups from the tile instead of from global memory.  The
// halo read of x alone already pulled whole 64 / 128-byte lines, i.e. the neighbouring segments, through L2 (160x160 maps:
// 2.7 reads of the concat buffer per output pixel); now it is read once, with the halo.
#define BN_STAMP(k)                                                                                                   \
  if (a.stamps && threadIdx.x == 0)                                                                                 \
    a.stamps[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 16 + (k)] = ((k) == 0 || (k) == 15) ? wall_clock64() : clock64();
template <typename T, int NT, int P1, int P2, bool SEP, int T2 = 0, int SG = 0, bool CL = false>
__global__ __launch_bounds__(256, ((NT >= 4 || P1 >= 15 || (NT == 2 && P1 == 5)) ? 2 : 3)) void bottleneck_mfma_kernel(const BneckArgs a) {
  static_assert(!CL || (SEP && T2 > 0 && SG > 0), "CL: the staged tile must survive conv_a, and cv2's gather must be the straight-line form");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int G = Tr<T>::G;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, col = lane & 15;
  const int tile_id = a.tile_major ? blockIdx.x : blockIdx.y;
  const int ty = (int)((tile_id * a.rcp_tx) >> 16), tx = tile_id - ty * a.tiles_x;
  const int n = a.tile_major ? blockIdx.y : blockIdx.x;  // image index fastest: an image's tiles share one XCD's L2 (see conv3x3_mfma_kernel)
  BN_STAMP(0)
  BN_STAMP(1)
  const int TH = a.TH, TW = a.TW, LW = a.LW, PS = a.PS, S = a.steps, CG = a.CG;
  const int PSA = CL ? a.PSA : PS;             // bytes per staged pixel
  const int CGA = CL ? a.kg : CG;              // K groups staged per pixel
  const int c1off = CL ? (a.kg - CG) * 16 : 0; // x inside the staged pixel
  const int oy0 = ty * TH, ox0 = tx * TW;
  const int IH = TH + 4, IW = TW + 4, H1 = TH + 2, W1 = TW + 2;
  const int R1 = H1 * W1, R2 = TH * TW;
  const int n1 = (R1 + 15) >> 4, n2 = (R2 + 15) >> 4;

  // LDS: [tap table 512 B][conv_a fragments S*NT KB][conv_b fragments S*NT KB][cv2 fragments][tile: input, later the intermediate]
  int* lds_toff = reinterpret_cast<int*>(smem);
  const int wbytes = S * NT * 1024;
  const u32x4* lds_w1 = reinterpret_cast<const u32x4*>(smem + 512);
  const u32x4* lds_w2 = reinterpret_cast<const u32x4*>(smem + 512 + wbytes);
  constexpr int SR = TailSteps<T>::per_nt(NT);  // cv2 K steps fed from the accumulators
  const int w3frags = T2 > 0 ? T2 * (a.sg + SR) : 0;
  const u32x4* lds_w3 = reinterpret_cast<const u32x4*>(smem + 512 + 2 * wbytes);
  char* tile = smem + 512 + 2 * wbytes + w3frags * 1024;
  char* tile2 = SEP ? tile + (TH + 4) * LW * PSA : tile;  // where the intermediate goes (pitch PS)
  for (int q = tid; q < S * 4; q += 256) {
    int tap = (int)(((unsigned)q * a.rcp_cg) >> 16);
    const int cg = q - tap * CG;
    tap = tap > 8 ? 8 : tap;
    const int ky = (tap * 21846) >> 16, kx = tap - 3 * ky;
    lds_toff[q] = (ky * LW + kx) * PSA + c1off + cg * 16;
    if (CL) lds_toff[64 + q] = (ky * LW + kx) * PS + cg * 16;   // conv_b walks the intermediate (S <= 16)
  }

  // ---- stage: both weight sets and the halo-2 input tile (see conv3x3_mfma_kernel for the slot map)
  const int RS = LW * (PSA >> 4), PSs = PSA >> 4;
  const int pcs = (RS + 63) >> 6;
  constexpr int NPC = CL ? 4 : 3;
  const int st_pitch = CL ? a.cat_pitch : a.in_pitch;
  const int iy0 = oy0 - 2, ix0 = ox0 - 2;
  const char* zeros = reinterpret_cast<const char*>(a.zeros);
  {
    // LDS-DMA (global_load_lds).  (Staging through registers -- request every piece, then store -- was tried here as in the
    // fused head: 10 % slower for this kernel, the waves sit at the ds_writes for the whole burst of the grid's first round.)
    const int wave_s = __builtin_amdgcn_readfirstlane(wave);
    for (int p = wave_s; p < 2 * S * NT; p += 4) {
      const u32x4* src = p < S * NT ? reinterpret_cast<const u32x4*>(a.w1) + p * 64 : reinterpret_cast<const u32x4*>(a.w2) + (p - S * NT) * 64;
      LP_GLDS16(src + lane, smem + 512 + p * 1024);
    }
    for (int p = wave_s; p < w3frags; p += 4) LP_GLDS16(reinterpret_cast<const u32x4*>(a.w3) + p * 64 + lane, smem + 512 + 2 * wbytes + p * 1024);
    const char* in_b = reinterpret_cast<const char*>(CL ? a.cat : a.in);
    if (pcs <= NPC) {
      // rows to the waves, the (<= 3) 64-slot pieces of a row unrolled: what depends on the lane only -- pixel, channel group,
      // validity, byte offset inside the row -- is computed once per piece instead of once per (row, piece); an item is then a
      // scalar row pointer, one 64-bit select and the DMA (the item loop below spends ~60 instructions per item, 9-11 k issue
      // cycles per workgroup on the 80x80 maps: tools/bneck_stamps.py)
      int voff[NPC];
      bool lok[NPC], inrs[NPC];
#pragma unroll
      for (int pc = 0; pc < NPC; ++pc) {
        const int sl = pc * 64 + lane;
        const int ix = (int)(((unsigned)sl * a.rcp_ps) >> 16), cgs = sl - ix * PSs;
        const int gx = ix0 + ix;
        inrs[pc] = sl < RS;
        lok[pc] = gx >= 0 && gx < a.W && ix < IW && cgs < CGA;
        voff[pc] = (ix * st_pitch + cgs * G) * (int)sizeof(T);
      }
      for (int iy = wave_s; iy < IH; iy += 4) {
        const int gy = iy0 + iy;
        const bool rok = gy >= 0 && gy < a.H;
        const char* rowp = in_b + ((long)(n * a.H + (rok ? gy : 0)) * a.W + ix0) * st_pitch * (long)sizeof(T);
#pragma unroll
        for (int pc = 0; pc < NPC; ++pc) {
          if (pc < pcs) {   // wave-uniform
            const char* src = (rok && lok[pc]) ? rowp + voff[pc] : zeros;
            if (inrs[pc]) LP_GLDS16(src, tile + (iy * RS + pc * 64) * 16);
          }
        }
      }
    } else {
      const int nitems = IH * pcs;
      for (int it = wave_s; it < nitems; it += 4) {
        const int iy = (int)(((unsigned)it * a.rcp_pcs) >> 16), pc = it - iy * pcs;
        const int gy = iy0 + iy;
        const char* rowp = in_b + ((long)(n * a.H + gy) * a.W + ix0) * st_pitch * (long)sizeof(T);
        const int sl = pc * 64 + lane;
        const int ix = (int)(((unsigned)sl * a.rcp_ps) >> 16), cgs = sl - ix * PSs;
        const int gx = ix0 + ix;
        const bool ok = gy >= 0 && gy < a.H && gx >= 0 && gx < a.W && ix < IW && cgs < CGA;
        const char* src = ok ? rowp + (ix * st_pitch + cgs * G) * (int)sizeof(T) : zeros;
        if (sl < RS) LP_GLDS16(src, tile + (iy * RS + pc * 64) * 16);
      }
    }
  }
  BN_STAMP(2)
  floatx4 bias1[NT], bias2[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    bias1[t] = *reinterpret_cast<const floatx4*>(a.b1 + g * 4 * NT + t * 4);
    bias2[t] = *reinterpret_cast<const floatx4*>(a.b2 + g * 4 * NT + t * 4);
  }

  // ---- conv_a over the (TH+2) x (TW+2) region
  int pk1[P1], pb1[P1];  // (py << 16) | px of this lane's pixel in tile i (clamped inside the region); its LDS offset
#pragma unroll
  for (int i = 0; i < P1; ++i) {
    int p = (wave + 4 * i) * 16 + col;
    p = p < R1 ? p : R1 - 1;
    const int py = (int)(((unsigned)p * a.rcp_w1) >> 16), px = p - py * W1;
    pk1[i] = (py << 16) | px;
    pb1[i] = (py * LW + px) * PSA;
  }
  floatx4 acc[NT][P1];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int i = 0; i < P1; ++i) acc[t][i] = floatx4{0.f, 0.f, 0.f, 0.f};
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  BN_STAMP(3)
  for (int s = 0; s < S; ++s) {
    const int toff = lds_toff[4 * s + g];
    typename Tr<T>::frag af[NT], bf[P1];
#pragma unroll
    for (int t = 0; t < NT; ++t) af[t] = as_frag<T>(lds_w1[(s * NT + t) * 64 + lane]);
#pragma unroll
    for (int i = 0; i < P1; ++i) bf[i] = as_frag<T>(*reinterpret_cast<const u32x4*>(tile + pb1[i] + toff));
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int i = 0; i < P1; ++i) acc[t][i] = Tr<T>::mma(af[t], bf[i], acc[t][i]);
  }
  BN_STAMP(4)
  // ---- the intermediate replaces the input tile (every wave is done reading it after the barrier), or goes
  //      to its own region (SEP: no barrier needed before writing)
  if (!SEP) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }
  // (interior workgroups -- the conv_a region lies inside the map -- skip the per-pixel inside-the-image arithmetic:
  //  a block-uniform branch around two copies of the loop)
  auto write_mid = [&](auto interior_tag) {
    constexpr bool INTERIOR = decltype(interior_tag)::value;
#pragma unroll
    for (int i = 0; i < P1; ++i) {
      const int p = (wave + 4 * i) * 16 + col;
      if (wave + 4 * i < n1 && p < R1) {
        const int py = pk1[i] >> 16, px = pk1[i] & 0xffff;
        bool inside = true;
        if constexpr (!INTERIOR) {
          const int gy = oy0 - 1 + py, gx = ox0 - 1 + px;
          inside = gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
        }
        T* dst = reinterpret_cast<T*>(tile2 + (py * LW + px) * PS) + g * 4 * NT;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          if (g * 4 * NT + t * 4 < a.C) {
            typename Tr<T>::quad q;
            const floatx4 y = act4<T, ACT_SILU>(acc[t][i], bias1[t]);
#pragma unroll
            for (int r = 0; r < 4; ++r) q[r] = inside ? (T)y[r] : (T)0.f;
            *reinterpret_cast<typename Tr<T>::quad*>(dst + t * 4) = q;
          }
        }
      }
    }
  };
  if (oy0 >= 1 && ox0 >= 1 && oy0 + TH + 1 <= a.H && ox0 + TW + 1 <= a.W) write_mid(std::true_type{});
  else write_mid(std::false_type{});
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  BN_STAMP(5)
  // ---- conv_b over the TH x TW output tile
  int pk2[P2], pb2[P2];
#pragma unroll
  for (int i = 0; i < P2; ++i) {
    int p = (wave + 4 * i) * 16 + col;
    p = p < R2 ? p : R2 - 1;
    const int oy = (int)(((unsigned)p * a.rcp_tw) >> 16), ox = p - oy * TW;
    pk2[i] = (oy << 16) | ox;
    pb2[i] = (oy * LW + ox) * PS;
  }
  floatx4 acc2[NT][P2];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int i = 0; i < P2; ++i) acc2[t][i] = floatx4{0.f, 0.f, 0.f, 0.f};
  for (int s = 0; s < S; ++s) {
    const int toff = lds_toff[(CL ? 64 : 0) + 4 * s + g];
    typename Tr<T>::frag af[NT], bf[P2];
#pragma unroll
    for (int t = 0; t < NT; ++t) af[t] = as_frag<T>(lds_w2[(s * NT + t) * 64 + lane]);
#pragma unroll
    for (int i = 0; i < P2; ++i) bf[i] = as_frag<T>(*reinterpret_cast<const u32x4*>(tile2 + pb2[i] + toff));
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int i = 0; i < P2; ++i) acc2[t][i] = Tr<T>::mma(af[t], bf[i], acc2[t][i]);
  }
  BN_STAMP(6)
  // ---- epilogue: bias, SiLU, shortcut (x from the preserved input tile, or re-read from global memory)
  const int chbase = g * 4 * NT;
  if constexpr (T2 == 0) {
#pragma unroll
    for (int i = 0; i < P2; ++i) {
      const int p = (wave + 4 * i) * 16 + col;
      const int gy = oy0 + (pk2[i] >> 16), gx = ox0 + (pk2[i] & 0xffff);
      if (wave + 4 * i < n2 && p < R2 && gy < a.H && gx < a.W) {
        const int pix = (n * a.H + gy) * a.W + gx;
        floatx4 v[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) v[t] = acc2[t][i];
        const T* xres = SEP ? reinterpret_cast<const T*>(tile + (((pk2[i] >> 16) + 2) * LW + (pk2[i] & 0xffff) + 2) * PSA + c1off) + chbase
                            : reinterpret_cast<const T*>(a.in) + eoff(pix, a.in_pitch) + chbase;
        store_lane_at<T, NT, ACT_SILU>(reinterpret_cast<T*>(a.out) + eoff(pix, a.out_pitch) + chbase, xres, chbase, a.C, v, bias2);
      }
    }
  } else {
    // ---- ... and cv2 on concat[.., y_last]: per pixel tile, gather the stored segments, turn the accumulators into
    //      y_last (rounded to T and shortcut added exactly as the stand-alone store does), two small GEMMs, store
    constexpr int SGMAX = sizeof(T) == 2 ? 3 : 6;
    const int sg_n = SG > 0 ? SG : a.sg;
    const int ST = sg_n + SR;
    floatx4 bias3[T2];
#pragma unroll
    for (int t = 0; t < T2; ++t) bias3[t] = *reinterpret_cast<const floatx4*>(a.b3 + g * 4 * T2 + t * 4);
    // The gathered concat segments of pixel tile i + LA are requested before tile i is worked on: one tile's arithmetic
    // (~600 cycles) does not cover a global load, and with the loads issued where they are used this loop WAS the kernel
    // (17-27 k of a workgroup's 40-50 k cycles: tools/bneck_stamps.py).
    constexpr int LA = P2 < 4 ? P2 : 4;
    auto tile_pixel = [&](int i, bool& valid) {
      const int p = (wave + 4 * i) * 16 + col;
      const int gy = oy0 + (pk2[i] >> 16), gx = ox0 + (pk2[i] & 0xffff);
      valid = wave + 4 * i < n2 && p < R2 && gy < a.H && gx < a.W;
      return valid ? (n * a.H + gy) * a.W + gx : 0;
    };
    // SG > 0 (fp16): the number of gathered K steps is a template parameter and this loop is straight-line code -- every
    // load unconditional from a valid address, masked by an AND where it is USED.  (A wave-uniform `if (s < a.sg)` or a
    // per-lane `if (valid)` around a load splits the basic block, the value must be complete at the block's end, and the
    // compiler puts s_waitcnt vmcnt(0) three instructions after every request.)  SG == 0 (fp32, up to 6 steps): branches.
    constexpr bool FLAT = SG > 0;
    constexpr int SGN = FLAT ? SG : SGMAX;
    // (without SEP the shortcut operand x comes from global memory too: requested with the concat segments, 8 bytes per channel
    //  quad, clamped to quad 0 past the layer's channels and masked by the channel test where it is used)
    typename Tr<T>::quad xq[LA][NT];
    auto request = [&](int i, u32x4 (&dst)[SGN], typename Tr<T>::quad (&xdst)[NT]) {
      bool valid;
      const int pix = tile_pixel(i, valid);
      if (FLAT && !SEP) {
        const T* xp = reinterpret_cast<const T*>(a.in) + eoff(pix, a.in_pitch);
#pragma unroll
        for (int t = 0; t < NT; ++t) xdst[t] = *reinterpret_cast<const typename Tr<T>::quad*>(xp + (chbase + t * 4 < a.C ? chbase + t * 4 : 0));
      }
      const T* catpix = reinterpret_cast<const T*>(a.cat) + eoff(pix, a.cat_pitch);
      // CL: the same groups from the staged tile (the tile pixel is clamped inside the tile: always a valid address)
      const char* ldspix = tile + (((pk2[i] >> 16) + 2) * LW + (pk2[i] & 0xffff) + 2) * PSA;
#pragma unroll
      for (int s = 0; s < SGN; ++s) {
        const int grp = 4 * s + g;   // K group = 8 (4) stored concat channels of the pixel
        if constexpr (CL) {
          dst[s] = *reinterpret_cast<const u32x4*>(ldspix + (grp < a.kg ? grp * 16 : 0));
        } else if (FLAT) {
          // lanes whose group lies past the stored segments re-read group 0 of the SAME pixel (masked where used): the address
          // stays inside the pixel -- group `grp` of the last pixel of the last image would lie past the end of the buffer
          dst[s] = *reinterpret_cast<const u32x4*>(catpix + (grp < a.kg ? grp * G : 0));
        } else {
          dst[s] = u32x4{0u, 0u, 0u, 0u};
          if (s < a.sg && valid && grp < a.kg) dst[s] = *reinterpret_cast<const u32x4*>(catpix + grp * G);
        }
      }
    };
    u32x4 bgq[LA][SGN];
#pragma unroll
    for (int i = 0; i < LA; ++i) request(i, bgq[i], xq[i]);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < P2; ++i) {
      if (!FLAT && wave + 4 * i >= n2) continue;  // wave-uniform
      bool valid;
      const int pix = tile_pixel(i, valid);
      typename Tr<T>::frag bg[SGN];
#pragma unroll
      for (int s = 0; s < SGN; ++s) {
        const unsigned m = (!FLAT || (valid && 4 * s + g < a.kg)) ? 0xffffffffu : 0u;
        bg[s] = as_frag<T>(bgq[i % LA][s] & m);
      }
      typename Tr<T>::quad xqi[NT];
#pragma unroll
      for (int t = 0; t < NT; ++t) xqi[t] = xq[i % LA][t];
      if (i + LA < P2) {
        request(i + LA, bgq[i % LA], xq[i % LA]);
        __builtin_amdgcn_sched_barrier(0);   // keep the requests up here (the scheduler sinks loads to their first use)
      }
      // y_last: activation, round to T, add the shortcut, round again (= store_lane_at), kept as the register B operand
      const T* xres = SEP ? reinterpret_cast<const T*>(tile + (((pk2[i] >> 16) + 2) * LW + (pk2[i] & 0xffff) + 2) * PSA + c1off) + chbase
                          : reinterpret_cast<const T*>(a.in) + eoff(pix, a.in_pitch) + chbase;
      T yl[NT][4];
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        typename Tr<T>::quad xr;
#pragma unroll
        for (int r = 0; r < 4; ++r) xr[r] = (T)0.f;
        if (SEP) {  // x from the preserved LDS tile: always a valid address (the tile's pixel is clamped), masked by the add below
          if (chbase + t * 4 < a.C) xr = *reinterpret_cast<const typename Tr<T>::quad*>(xres + t * 4);
        } else if (FLAT) {
          if (valid && chbase + t * 4 < a.C) xr = xqi[t];   // (a select on registers)
        } else if (valid && chbase + t * 4 < a.C) xr = *reinterpret_cast<const typename Tr<T>::quad*>(xres + t * 4);
        const floatx4 y = act4<T, ACT_SILU>(acc2[t][i], bias2[t]);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const T q = (T)y[r];
          yl[t][r] = (T)((float)q + (float)xr[r]);
        }
      }
      typename Tr<T>::frag bq[SR];
      if constexpr (sizeof(T) == 2) {
#pragma unroll
        for (int s = 0; s < SR; ++s)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            bq[s][r] = yl[2 * s][r];
            bq[s][4 + r] = (2 * s + 1 < NT) ? yl[(2 * s + 1 < NT) ? 2 * s + 1 : 0][r] : (T)0.f;
          }
      } else {
#pragma unroll
        for (int s = 0; s < SR; ++s)
#pragma unroll
          for (int r = 0; r < 4; ++r) bq[s][r] = yl[s][r];
      }
      floatx4 o[T2];
#pragma unroll
      for (int t = 0; t < T2; ++t) o[t] = floatx4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < SGN; ++s) {
        if (FLAT || s < a.sg) {
#pragma unroll
          for (int t = 0; t < T2; ++t) o[t] = Tr<T>::mma(as_frag<T>(lds_w3[(t * ST + s) * 64 + lane]), bg[s], o[t]);
        }
      }
#pragma unroll
      for (int s = 0; s < SR; ++s)
#pragma unroll
        for (int t = 0; t < T2; ++t) o[t] = Tr<T>::mma(as_frag<T>(lds_w3[(t * ST + sg_n + s) * 64 + lane]), bq[s], o[t]);
      if (valid) {
        T* o3 = reinterpret_cast<T*>(a.out3) + eoff(pix, a.out3_pitch) + g * 4 * T2;
        if (a.act3 == ACT_SILU) store_lane_at<T, T2, ACT_SILU>(o3, nullptr, g * 4 * T2, a.C3, o, bias3);
        else store_lane_at<T, T2, ACT_NONE>(o3, nullptr, g * 4 * T2, a.C3, o, bias3);
      }
    }
  }
  BN_STAMP(7)
  BN_STAMP(15)
}

// ------------------------------------------------------------------------------------
// 1x1 conv = GEMM over flattened pixels.  Every pixel is read exactly once, so the pixel
// operand goes straight from global memory to registers (16 B per lane); the weights of
// this block's channel split live in LDS for the whole (grid-stride) pixel loop.  M may come
// from device memory (m_dyn): the classifier's ROI count is only known on the GPU.
// ------------------------------------------------------------------------------------
// UPS: the layer reads concat(upsample2x(u), x) (FPN top-down path: Interp + Concat + C2f.cv1 in the reference graph).
// The first up_cg K groups are then gathered from the half-resolution tensor u at (y/2, x/2) -- nearest-neighbour
// upsampling is pure addressing -- and the 4x larger copy of u is never written or re-read.
template <typename T, int NT, int NP, int EPI, bool UPS = false>
__global__ __launch_bounds__(256) void conv1x1_mfma_kernel(const ConvArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int G = Tr<T>::G;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, col = lane & 15;
  const int ns = blockIdx.y;
  const int S = a.steps;
  const int CG = a.Cin / G;

  const long M = a.m_dyn ? (long)(*a.m_dyn) * a.pix_per_item : (long)a.M;
  const long ntiles = (M + 64 * NP - 1) / (64 * NP);
  if ((long)blockIdx.x >= ntiles) return;  // grid is sized for the capacity; idle blocks must not stage weights

  u32x4* lds_w = reinterpret_cast<u32x4*>(smem);
  const u32x4* wsrc = reinterpret_cast<const u32x4*>(a.wpk) + (size_t)ns * S * NT * 64;
  for (int i = tid; i < S * NT * 64; i += 256) lds_w[i] = wsrc[i];
  __syncthreads();
  const T* in = reinterpret_cast<const T*>(a.in);
  floatx4 bias_r[NT];  // this lane's 4*NT output channels: bias hoisted out of the epilogue
#pragma unroll
  for (int t = 0; t < NT; ++t) bias_r[t] = *reinterpret_cast<const floatx4*>(a.bias + ns * 16 * NT + g * 4 * NT + t * 4);

  for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const long pix0 = (tile * 4 + wave) * 16 * NP;
    if (pix0 >= M) continue;
    const T* src[NP];
    const T* src2[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      long pix = pix0 + p * 16 + col;
      pix = pix < M ? pix : M - 1;
      src[p] = in + eoff(pix, a.in_pitch);
      src2[p] = nullptr;
      if constexpr (UPS) {
        const int hw = a.pix_per_item;
        const int n = fast_div((int)pix, hw, a.magic_hw), rem = (int)pix - n * hw;
        const int y = fast_div(rem, a.Wout, a.magic_w), x = rem - y * a.Wout;
        src2[p] = reinterpret_cast<const T*>(a.up) + eoff((n * (a.Hout >> 1) + (y >> 1)) * (a.Wout >> 1) + (x >> 1), a.up_pitch);
      }
    }
    floatx4 acc[NT][NP];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int p = 0; p < NP; ++p) acc[t][p] = floatx4{0.f, 0.f, 0.f, 0.f};

    for (int s = 0; s < S; ++s) {
      const int q = 4 * s + g;
      typename Tr<T>::frag af[NT], bf[NP];
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        u32x4 v = u32x4{0u, 0u, 0u, 0u};
        if constexpr (UPS) {
          if (q < CG) v = *reinterpret_cast<const u32x4*>(q < a.up_cg ? src2[p] + q * G : src[p] + (q - a.up_cg) * G);
        } else {
          if (q < CG) v = *reinterpret_cast<const u32x4*>(src[p] + q * G);
        }
        bf[p] = as_frag<T>(v);
      }
#pragma unroll
      for (int t = 0; t < NT; ++t) af[t] = as_frag<T>(lds_w[(s * NT + t) * 64 + lane]);
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int p = 0; p < NP; ++p) acc[t][p] = Tr<T>::mma(af[t], bf[p], acc[t][p]);
    }
    if (a.act == ACT_SILU) epilogue_flat<T, NT, NP, EPI, ACT_SILU>(a, acc, bias_r, pix0, M, ns, g, col);
    else if (a.act == ACT_RELU) epilogue_flat<T, NT, NP, EPI, ACT_RELU>(a, acc, bias_r, pix0, M, ns, g, col);
    else epilogue_flat<T, NT, NP, EPI, ACT_NONE>(a, acc, bias_r, pix0, M, ns, g, col);
  }
}

// ------------------------------------------------------------------------------------
// 3x3 stride-2 (pad 1) implicit GEMM without input staging.  A stride-2 halo tile is 4x the
// output tile, so LDS staging re-reads more than it saves: each input element feeds only
// 9/4 taps on average.  Same structure as the 1x1 kernel: this block's weight fragments (all
// of K = 9 x Cin) in LDS, B fragments gathered straight from global memory with per-tap bounds
// checks (zero padding), flattened output pixels.
// ------------------------------------------------------------------------------------
template <typename T, int NT, int NP, int T2 = 0, int U = 4>
__global__ __launch_bounds__(256) void conv3x3s2_direct_kernel(const ConvArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int G = Tr<T>::G;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, col = lane & 15;
  const int ns = blockIdx.y;
  const int S = a.steps;
  const int CG = a.Cin / G;
  const long M = a.m_dyn ? (long)(*a.m_dyn) * a.pix_per_item : (long)a.M;
  const long ntiles = (M + 64 * NP - 1) / (64 * NP);
  if ((long)blockIdx.x >= ntiles) return;

  u32x4* lds_w = reinterpret_cast<u32x4*>(smem);
  const u32x4* wsrc = reinterpret_cast<const u32x4*>(a.wpk) + (size_t)ns * S * NT * 64;
  const int nW = S * NT * 64;
  for (int r0 = 0; r0 < nW; r0 += 256 * 8) {
    u32x4 wv[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = r0 + u * 256 + tid;
      wv[u] = u32x4{0u, 0u, 0u, 0u};
      if (i < nW) wv[u] = wsrc[i];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = r0 + u * 256 + tid;
      if (i < nW) lds_w[i] = wv[u];
    }
  }
  // K group q = (tap, channel group): element offset from the window's top-left pixel and (valid, ky, kx), once
  // per workgroup instead of two integer divisions per lane per K step
  int2* lds_tab = reinterpret_cast<int2*>(smem + (size_t)nW * 16);
  for (int q = tid; q < S * 4; q += 256) {
    const int tap = q / CG, cg = q - tap * CG;
    const int ky = tap / 3, kx = tap - 3 * ky;
    lds_tab[q] = make_int2((ky * a.Win + kx) * a.in_pitch + cg * G, tap < 9 ? ((ky << 8) | kx) : -1);
  }
  __syncthreads();
  const T* in = reinterpret_cast<const T*>(a.in);
  const int HWo = a.Hout * a.Wout;
  floatx4 bias_r[NT];  // this lane's 4*NT output channels: bias hoisted out of the epilogue
#pragma unroll
  for (int t = 0; t < NT; ++t) bias_r[t] = *reinterpret_cast<const floatx4*>(a.bias + ns * 16 * NT + g * 4 * NT + t * 4);

  for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const long pix0 = (tile * 4 + wave) * 16 * NP;
    if (pix0 >= M) continue;
    const T* src[NP];
    int iy0[NP], ix0[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      long pix = pix0 + p * 16 + col;
      pix = pix < M ? pix : M - 1;
      const int n = fast_div((int)pix, HWo, a.magic_hw);
      const int rem = (int)pix - n * HWo;
      const int oy = fast_div(rem, a.Wout, a.magic_w), ox = rem - oy * a.Wout;
      iy0[p] = oy * 2 - 1;
      ix0[p] = ox * 2 - 1;
      // dereferenced only when in bounds (the signed pixel index of the padding row/column wraps like the pointer would)
      src[p] = in + (long)(int)eoff((n * a.Hin + iy0[p]) * a.Win + ix0[p], a.in_pitch);
    }
    floatx4 acc[NT][NP];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int p = 0; p < NP; ++p) acc[t][p] = floatx4{0.f, 0.f, 0.f, 0.f};

    // Every K step's gather is a memory round trip of its own, so the loads run U steps ahead of the MFMAs in a
    // register ring of U steps (deep-K layers have 9-36 steps; without the ring each step waited for its own loads;
    // shallow layers use U = 1: they are HBM-bound and want the registers for occupancy instead).
    typename Tr<T>::frag ring[U][NP];
    auto gather = [&](int s, typename Tr<T>::frag (&dst)[NP]) {
      const int2 te = s < S ? lds_tab[4 * s + g] : make_int2(0, -1);
      const int toff = te.x, ky = te.y >> 8, kx = te.y & 0xff;
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        u32x4 v = u32x4{0u, 0u, 0u, 0u};
        const int iy = iy0[p] + ky, ix = ix0[p] + kx;
        if (te.y >= 0 && iy >= 0 && iy < a.Hin && ix >= 0 && ix < a.Win) v = *reinterpret_cast<const u32x4*>(src[p] + toff);
        dst[p] = as_frag<T>(v);
      }
    };
#pragma unroll
    for (int u = 0; u < U; ++u) gather(u, ring[u]);
    for (int s0 = 0; s0 < S; s0 += U) {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int s = s0 + u;
        if (s < S) {
          typename Tr<T>::frag af[NT];
#pragma unroll
          for (int t = 0; t < NT; ++t) af[t] = as_frag<T>(lds_w[(s * NT + t) * 64 + lane]);
#pragma unroll
          for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int p = 0; p < NP; ++p) acc[t][p] = Tr<T>::mma(af[t], ring[u][p], acc[t][p]);
        }
        gather(s + U, ring[u]);
      }
    }
    if constexpr (T2 > 0) {
      typename Tr<T>::frag w2f[T2][TailSteps<T>::per_nt(NT)];
      floatx4 bias2[T2];
      tail_load<T, NT, T2>(a, lane, g, w2f, bias2);
      ConvArgs a2 = a;
      a2.Cout = a.Cout2; a2.act = a.act2; a2.res = nullptr;
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        const long pix = pix0 + p * 16 + col;
        if (pix < M) {
          floatx4 v[NT];
#pragma unroll
          for (int t = 0; t < NT; ++t) v[t] = acc[t][p];
          if (a.act == ACT_SILU) tail_store<T, NT, T2, ACT_SILU>(a2, pix, g, v, bias_r, w2f, bias2);
          else tail_store<T, NT, T2, ACT_NONE>(a2, pix, g, v, bias_r, w2f, bias2);
        }
      }
    } else {
      if (a.act == ACT_SILU) epilogue_flat<T, NT, NP, EPI_PLAIN, ACT_SILU>(a, acc, bias_r, pix0, M, ns, g, col);
      else if (a.act == ACT_RELU) epilogue_flat<T, NT, NP, EPI_PLAIN, ACT_RELU>(a, acc, bias_r, pix0, M, ns, g, col);
      else epilogue_flat<T, NT, NP, EPI_PLAIN, ACT_NONE>(a, acc, bias_r, pix0, M, ns, g, col);
    }
  }
}

// ------------------------------------------------------------------------------------
// Naive direct convolution (one thread per output element, fp32 accumulate).  GPU-side
// debugging aid selected with lp_config.conv_impl = 1; never the product path.
// Weights: [Cout][k*k][Cin] as T.
// ------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void conv_naive_kernel(const ConvArgs a, int k, int stride) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  const long M = a.m_dyn ? (long)(*a.m_dyn) * a.pix_per_item : (long)a.N * a.Hout * a.Wout;
  if (idx >= M * a.Cout) return;
  const int co = (int)(idx % a.Cout);
  const long pix = idx / a.Cout;
  const int ox = (int)(pix % a.Wout);
  const int oy = (int)((pix / a.Wout) % a.Hout);
  const int n = (int)(pix / ((long)a.Wout * a.Hout));
  const int pad = k / 2;
  const T* in = reinterpret_cast<const T*>(a.in);
  const T* w = reinterpret_cast<const T*>(a.wpk) + (size_t)co * k * k * a.Cin;
  float acc = 0.f;
  for (int ky = 0; ky < k; ++ky) {
    const int iy = oy * stride - pad + ky;
    if (iy < 0 || iy >= a.Hin) continue;
    for (int kx = 0; kx < k; ++kx) {
      const int ix = ox * stride - pad + kx;
      if (ix < 0 || ix >= a.Win) continue;
      const T* px = in + ((long)(n * a.Hin + iy) * a.Win + ix) * a.in_pitch;
      const T* wt = w + (ky * k + kx) * a.Cin;
      for (int ci = 0; ci < a.Cin; ++ci) acc = fmaf((float)px[ci], (float)wt[ci], acc);
    }
  }
  float v;
  if (a.res && a.res_first)  // ResNet BasicBlock: act(conv + bias + identity), as store_lane_at
    v = activate<T>(acc + (float)reinterpret_cast<const T*>(a.res)[pix * a.res_pitch + co] + a.bias[co], a.act);
  else
    v = activate<T>(acc + a.bias[co], a.act);
  if (a.x1) {  // shuffle epilogue (see store_quad)
    if (co < a.half_c) {
      const int l = 2 * co;
      const int phys = l < a.half_c ? l : a.half_cp + (l - a.half_c);
      T* o = reinterpret_cast<T*>(a.out) + pix * a.out_pitch;
      o[phys] = reinterpret_cast<const T*>(a.x1)[pix * a.x1_pitch + co];
      o[phys + 1] = (T)v;
    }
    return;
  }
  if (a.res && !a.res_first) v += (float)reinterpret_cast<const T*>(a.res)[pix * a.res_pitch + co];
  if (a.out_f32)
    reinterpret_cast<float*>(a.out)[pix * a.out_pitch + co] = v;
  else
    reinterpret_cast<T*>(a.out)[pix * a.out_pitch + co] = (T)v;
}

// uint8 input tile -> LDS as dwords: `rows` x `roww` dwords, row r = image row iy0 + r, dword c = row dword w0 + c, zeros outside
// the image.  Every thread requests ALL of its dwords (<= MAXI) before the first store -- unconditional loads from clamped
// addresses, masked when stored; as a loop of `if (inside) v = load; store v` each iteration was a memory round trip of its own.
template <int MAXI>
__device__ __forceinline__ void stage_u8_tile(const uint32_t* __restrict__ im, uint32_t* __restrict__ tile, int rows, int roww, int iy0, int w0,
                                              int Hin, int row_words, int tid) {
  const int total = rows * roww;
  uint32_t v[MAXI];
  unsigned keep = 0;
#pragma unroll
  for (int k = 0; k < MAXI; ++k) {
    int i = tid + 256 * k;
    i = i < total ? i : total - 1;
    const int r = i / roww, c = i - r * roww;
    const int iy = iy0 + r, wi = w0 + c;
    const int iyc = iy < 0 ? 0 : (iy < Hin ? iy : Hin - 1), wic = wi < 0 ? 0 : (wi < row_words ? wi : row_words - 1);
    v[k] = im[(long)iyc * row_words + wic];
    keep |= (iy >= 0 && iy < Hin && wi >= 0 && wi < row_words) ? 1u << k : 0u;
  }
#pragma unroll
  for (int k = 0; k < MAXI; ++k) {
    const int i = tid + 256 * k;
    if (i < total) tile[i] = ((keep >> k) & 1u) ? v[k] : 0u;
  }
}

// ------------------------------------------------------------------------------------
// Stem: 3x3 stride-2 pad-1 conv straight from the uint8 BGR image (the reference's
// BGR->RGB + x*(1/255) + HWC->CHW preprocessing, e2e.py:222-238, is folded in: the weight
// table is stored in BGR order and the scale is applied to the pixel before the FMA).
// One thread = one output pixel x all CO channels.  w: fp32 [k*k*3][CO], row = (ky*k+kx)*3 + c_bgr.  Generic in kernel
// size / stride / padding (YOLOv5's 6x6/s2/p2 stem runs here); the 3x3/s2/p1 stems of the YOLOv8 family take the
// LDS / MFMA kernels below.
// ------------------------------------------------------------------------------------
template <typename T, int CO>
__global__ __launch_bounds__(256) void stem_conv_kernel(const uint8_t* __restrict__ img, T* __restrict__ out,
                                                        const float* __restrict__ w, const float* __restrict__ bias,
                                                        int N, int Hin, int Win, int Hout, int Wout, int out_pitch,
                                                        int act, int k, int stride, int pad) {
  const long pix = (long)blockIdx.x * 256 + threadIdx.x;
  if (pix >= (long)N * Hout * Wout) return;
  const int ox = (int)(pix % Wout);
  const int oy = (int)((pix / Wout) % Hout);
  const int n = (int)(pix / ((long)Wout * Hout));
  float acc[CO];
#pragma unroll
  for (int c = 0; c < CO; ++c) acc[c] = 0.f;
  const float inv255 = 1.f / 255.f;
  for (int ky = 0; ky < k; ++ky) {
    const int iy = oy * stride - pad + ky;
    if (iy < 0 || iy >= Hin) continue;
    for (int kx = 0; kx < k; ++kx) {
      const int ix = ox * stride - pad + kx;
      if (ix < 0 || ix >= Win) continue;
      const uint8_t* px = img + ((long)(n * Hin + iy) * Win + ix) * 3;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const float v = (float)px[c] * inv255;
        const float* wr = w + ((ky * k + kx) * 3 + c) * CO;
#pragma unroll
        for (int co = 0; co < CO; ++co) acc[co] = fmaf(v, wr[co], acc[co]);
      }
    }
  }
  T* o = out + pix * out_pitch;
#pragma unroll
  for (int c0 = 0; c0 < CO; c0 += 4) {
    typename Tr<T>::quad q;
#pragma unroll
    for (int i = 0; i < 4; ++i) q[i] = (T)activate<T>(acc[c0 + i] + bias[c0 + i], act);
    *reinterpret_cast<typename Tr<T>::quad*>(o + c0) = q;
  }
}

// Stem, fast path: one workgroup = 8 x 32 output pixels.  The 17 x 65-pixel uint8 input tile is
// staged in LDS with coalesced dword loads (27 scattered byte loads per output pixel made the
// first version texture-address bound); each thread then reads 3 dwords per input row and pulls
// its 9 bytes out with v_alignbyte.  Needs 4-byte aligned image rows (Win % 4 == 0).
#define STEM_TH 8
#define STEM_TW 32
#define STEM_ROWW 52  /* dwords per staged row: 1 + 65*3 bytes rounded up, + 2 so that every thread's 3rd dword exists */
template <typename T, int CO>
__global__ __launch_bounds__(256) void stem_conv_lds_kernel(const uint8_t* __restrict__ img, T* __restrict__ out,
                                                            const float* __restrict__ w, const float* __restrict__ bias,
                                                            int N, int Hin, int Win, int Hout, int Wout, int out_pitch,
                                                            int act) {
  __shared__ uint32_t tile[(2 * STEM_TH + 1) * STEM_ROWW];
  const int tid = threadIdx.x;
  const int n = blockIdx.z, oy0 = blockIdx.y * STEM_TH, ox0 = blockIdx.x * STEM_TW;
  const int row_words = Win * 3 / 4;
  const int w0 = (6 * ox0 - 3) >> 2;  // first staged dword of a row (arithmetic shift: -1 for the left border tile)
  const uint32_t* im = reinterpret_cast<const uint32_t*>(img + (long)n * Hin * Win * 3);
  stage_u8_tile<((2 * STEM_TH + 1) * STEM_ROWW + 255) / 256>(im, tile, 2 * STEM_TH + 1, STEM_ROWW, 2 * oy0 - 1, w0, Hin, row_words, tid);
  __syncthreads();
  const int ty = tid >> 5, tx = tid & 31;
  const int oy = oy0 + ty, ox = ox0 + tx;
  if (oy >= Hout || ox >= Wout) return;
  float acc[CO];
#pragma unroll
  for (int c = 0; c < CO; ++c) acc[c] = 0.f;
  const float inv255 = 1.f / 255.f;
  const int boff = 1 + 6 * tx;        // first needed byte inside the staged row (the tile starts at byte 6*ox0-3 = 4*w0 + 1)
  const int wi = boff >> 2, sh = boff & 3;
  // bytes right of the image (ix >= Win) are zeros already; the pixel left of the image (ix = -1) lives in dword -1 -> zero
  // (ky is NOT unrolled: the weights are wave-uniform scalars, and all 27*CO of them live at once spill the SGPR file)
#pragma unroll 1
  for (int ky = 0; ky < 3; ++ky) {
    const uint32_t* rw = tile + (2 * ty + ky) * STEM_ROWW + wi;
    const uint32_t d0 = rw[0], d1 = rw[1], d2 = rw[2];
    const uint32_t wa = __builtin_amdgcn_alignbyte(d1, d0, sh);
    const uint32_t wb = __builtin_amdgcn_alignbyte(d2, d1, sh);
    const uint32_t wc = d2 >> (8 * sh);
    float px[9];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      px[j] = (float)((wa >> (8 * j)) & 0xffu) * inv255;
      px[4 + j] = (float)((wb >> (8 * j)) & 0xffu) * inv255;
    }
    px[8] = (float)(wc & 0xffu) * inv255;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      const float* wr = w + (ky * 9 + k) * CO;  // (ky*3+kx)*3 + c with k = kx*3 + c
#pragma unroll
      for (int co = 0; co < CO; ++co) acc[co] = fmaf(px[k], wr[co], acc[co]);
    }
  }
  T* o = out + ((long)(n * Hout + oy) * Wout + ox) * out_pitch;
#pragma unroll
  for (int c0 = 0; c0 < CO; c0 += 4) {
    typename Tr<T>::quad q;
#pragma unroll
    for (int i = 0; i < 4; ++i) q[i] = (T)activate<T>(acc[c0 + i] + bias[c0 + i], act);
    *reinterpret_cast<typename Tr<T>::quad*>(o + c0) = q;
  }
}


// Stem on the matrix cores (fp16, 8 output channels).  K = the 3 x 9 window bytes is only 27 deep and 8 channels
// fill half an MFMA tile, so two horizontally adjacent output pixels share one column: rows 0-7 = channels of the
// even pixel, rows 8-15 = channels of the odd one, K = 3 rows x the 15-byte union of the two windows (padded to
// 16 -> K = 48 of 64 used, two 16x16x32 steps).  A lane's 8 K values are 8 consecutive bytes of one staged row:
// 3 dword LDS reads + v_alignbyte, then bytes -> fp16 exactly via 0x6400|b (= 1024 + b) minus 1024.  The 1/255 of
// preprocess (e2e.py:222-238) is folded into the fp16 weights.  Against the scalar kernel above this drops the
// per-pixel VALU work (27 x 8 FMAs + 27 conversions) that made the stem the largest single launch.
#define STEMM_TH 8
#define STEMM_TW 64
#define STEMM_ROWW 100 /* dwords per staged row: 1 + 129*3 bytes -> 97, + the lanes' 3rd dword */
__global__ __launch_bounds__(256) void stem_mfma_kernel(const uint8_t* __restrict__ img, half_t* __restrict__ out,
                                                        const u32x4* __restrict__ afrag, const float* __restrict__ bias,
                                                        int N, int Hin, int Win, int Hout, int Wout, int out_pitch) {
  __shared__ uint32_t tile[(2 * STEMM_TH + 1) * STEMM_ROWW];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, col = lane & 15;
  const int n = blockIdx.x, oy0 = blockIdx.z * STEMM_TH, ox0 = blockIdx.y * STEMM_TW;  // image fastest: XCD-local halos
  const int row_words = Win * 3 / 4;
  const int w0 = (6 * ox0 - 3) >> 2;  // first staged dword of a row (tile byte 1 = window byte 0 of column ox0)
  const uint32_t* im = reinterpret_cast<const uint32_t*>(img + (long)n * Hin * Win * 3);
  stage_u8_tile<((2 * STEMM_TH + 1) * STEMM_ROWW + 255) / 256>(im, tile, 2 * STEMM_TH + 1, STEMM_ROWW, 2 * oy0 - 1, w0, Hin, row_words, tid);
  const half8 af0 = __builtin_bit_cast(half8, afrag[lane]);
  const half8 af1 = __builtin_bit_cast(half8, afrag[64 + lane]);
  const int c0 = (g & 1) * 4;
  const floatx4 b4 = *reinterpret_cast<const floatx4*>(bias + c0);
  __syncthreads();
  const half2v k1024 = {(half_t)1024.f, (half_t)1024.f};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int mt = wave * 4 + i;
    const int r = mt >> 1, hx = mt & 1;
    const int txe = 32 * hx + 2 * col;  // even pixel of this lane's pair, tile-relative
    half8 bf[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int q = 4 * s + g;
      const int ky = q >> 1, h = q & 1;
      const int bo = 1 + 6 * txe + 8 * h;
      const uint32_t* rw = tile + (2 * r + (ky < 3 ? ky : 2)) * STEMM_ROWW + (bo >> 2);
      const int sh = bo & 3;
      const uint32_t d0 = rw[0], d1 = rw[1], d2 = rw[2];
      uint32_t wa = __builtin_amdgcn_alignbyte(d1, d0, sh);
      uint32_t wb = __builtin_amdgcn_alignbyte(d2, d1, sh);
      if (q >= 6) { wa = 0u; wb = 0u; }  // K padding (weights are zero as well)
      // bytes -> (1024 + b) as fp16 pairs, then - 1024
      const uint32_t p0 = __builtin_amdgcn_perm(0x64646464u, wa, 0x04010400u);
      const uint32_t p1 = __builtin_amdgcn_perm(0x64646464u, wa, 0x04030402u);
      const uint32_t p2 = __builtin_amdgcn_perm(0x64646464u, wb, 0x04010400u);
      const uint32_t p3 = __builtin_amdgcn_perm(0x64646464u, wb, 0x04030402u);
      const half2v h0 = __builtin_bit_cast(half2v, p0) - k1024, h1 = __builtin_bit_cast(half2v, p1) - k1024;
      const half2v h2 = __builtin_bit_cast(half2v, p2) - k1024, h3 = __builtin_bit_cast(half2v, p3) - k1024;
      bf[s] = half8{h0[0], h0[1], h1[0], h1[1], h2[0], h2[1], h3[0], h3[1]};
    }
    floatx4 acc = {0.f, 0.f, 0.f, 0.f};
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(af0, bf[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(af1, bf[1], acc, 0, 0, 0);
    const int oy = oy0 + r, ox = ox0 + txe + (g >> 1);
    if (oy < Hout && ox < Wout) {
      half4 q4;
      const floatx4 y4 = act4<half_t, ACT_SILU>(acc, b4);
#pragma unroll
      for (int j = 0; j < 4; ++j) q4[j] = (half_t)y4[j];
      *reinterpret_cast<half4*>(out + ((long)(n * Hout + oy) * Wout + ox) * out_pitch + c0) = q4;
    }
  }
}


// 16-channel variant (v2 widths): one pixel per MFMA column, rows = the 16 output channels, K = 27 window bytes in one
// 16x16x32 step: K group g < 3 = bytes 0..7 of window row g, group 3 = byte 8 of the three rows (+ 5 zero slots).
__global__ __launch_bounds__(256) void stem_mfma16_kernel(const uint8_t* __restrict__ img, half_t* __restrict__ out,
                                                          const u32x4* __restrict__ afrag, const float* __restrict__ bias,
                                                          int N, int Hin, int Win, int Hout, int Wout, int out_pitch) {
  __shared__ uint32_t tile[(2 * STEMM_TH + 1) * STEMM_ROWW];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, col = lane & 15;
  const int n = blockIdx.x, oy0 = blockIdx.z * STEMM_TH, ox0 = blockIdx.y * STEMM_TW;
  const int row_words = Win * 3 / 4;
  const int w0 = (6 * ox0 - 3) >> 2;
  const uint32_t* im = reinterpret_cast<const uint32_t*>(img + (long)n * Hin * Win * 3);
  stage_u8_tile<((2 * STEMM_TH + 1) * STEMM_ROWW + 255) / 256>(im, tile, 2 * STEMM_TH + 1, STEMM_ROWW, 2 * oy0 - 1, w0, Hin, row_words, tid);
  const half8 af = __builtin_bit_cast(half8, afrag[lane]);
  const floatx4 b4 = *reinterpret_cast<const floatx4*>(bias + 4 * g);
  __syncthreads();
  const half2v k1024 = {(half_t)1024.f, (half_t)1024.f};
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int mt = wave * 8 + i;
    const int r = mt >> 2, tx = (mt & 3) * 16 + col;
    const int bo = 1 + 6 * tx;  // first window byte of this pixel inside a staged row
    uint32_t wa, wb;
    if (g < 3) {
      const uint32_t* rw = tile + (2 * r + g) * STEMM_ROWW + (bo >> 2);
      const int sh = bo & 3;
      const uint32_t d0 = rw[0], d1 = rw[1], d2 = rw[2];
      wa = __builtin_amdgcn_alignbyte(d1, d0, sh);
      wb = __builtin_amdgcn_alignbyte(d2, d1, sh);
    } else {
      const int b8 = bo + 8;
      const uint8_t* t8 = reinterpret_cast<const uint8_t*>(tile);
      wa = (uint32_t)t8[(2 * r + 0) * STEMM_ROWW * 4 + b8] | ((uint32_t)t8[(2 * r + 1) * STEMM_ROWW * 4 + b8] << 8) |
           ((uint32_t)t8[(2 * r + 2) * STEMM_ROWW * 4 + b8] << 16);
      wb = 0u;
    }
    const uint32_t p0 = __builtin_amdgcn_perm(0x64646464u, wa, 0x04010400u);
    const uint32_t p1 = __builtin_amdgcn_perm(0x64646464u, wa, 0x04030402u);
    const uint32_t p2 = __builtin_amdgcn_perm(0x64646464u, wb, 0x04010400u);
    const uint32_t p3 = __builtin_amdgcn_perm(0x64646464u, wb, 0x04030402u);
    const half2v h0 = __builtin_bit_cast(half2v, p0) - k1024, h1 = __builtin_bit_cast(half2v, p1) - k1024;
    const half2v h2 = __builtin_bit_cast(half2v, p2) - k1024, h3 = __builtin_bit_cast(half2v, p3) - k1024;
    const half8 bf = half8{h0[0], h0[1], h1[0], h1[1], h2[0], h2[1], h3[0], h3[1]};
    floatx4 acc = {0.f, 0.f, 0.f, 0.f};
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(af, bf, acc, 0, 0, 0);
    const int oy = oy0 + r, ox = ox0 + tx;
    if (oy < Hout && ox < Wout) {
      half4 q4;
      const floatx4 y4 = act4<half_t, ACT_SILU>(acc, b4);
#pragma unroll
      for (int j = 0; j < 4; ++j) q4[j] = (half_t)y4[j];
      *reinterpret_cast<half4*>(out + ((long)(n * Hout + oy) * Wout + ox) * out_pitch + 4 * g) = q4;
    }
  }
}

// ------------------------------------------------------------------------------------
// Network head in one launch (fp16, v1 widths): stem 3x3/s2 (uint8 BGR -> 8 ch, SiLU) -> 3x3/s2 conv (8 -> <=16 ch,
// SiLU) -> its 1x1 tail (C2f.cv1).  The 320x320x8 stem map is the largest activation of the network and was written
// and read back once per image (210 MB per 64-image step); here a workgroup computes the 17 x 65 stem pixels its
// 8 x 32 output tile needs into LDS (stem_mfma_kernel's formulation; zero outside the map = the next conv's padding),
// runs the stride-2 conv from there with static LDS offsets (the gather kernel spent its VALU on per-tap address and
// bounds arithmetic) and finishes with tail_store.  Stem halo recompute: (17 x 65) / (16 x 64) = +8 %.
// ------------------------------------------------------------------------------------
#define SB_TH 8
#define SB_TW 32
#define SB_SH (2 * SB_TH + 1)       /* stem rows  */
#define SB_SW (2 * SB_TW + 1)       /* stem cols  */
#define SB_LW 65                    /* stem tile row pitch, pixels (16 B each) */
#define SB_IR (2 * SB_SH + 1)       /* input rows */
#define SB_ROWW 102                 /* dwords per staged input row: 3 + 131*3 bytes + the lanes' 3rd dword (<= 101); with SB_LW 65 the
                                       kernel's 31,968 B of LDS are 25 granules of 1280 B: FIVE workgroups per CU (32,512 B: four) */
#define SB_PAIRS ((SB_SW + 1) / 2)  /* pixel pairs per stem row */
__global__ __launch_bounds__(256) void stem_block_kernel(const StemBlockArgs a) {
  __shared__ uint32_t in_tile[SB_IR * SB_ROWW];
  __shared__ __attribute__((aligned(16))) char st_tile[SB_SH * SB_LW * 16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, col = lane & 15;
  const int n = blockIdx.x, ox0 = blockIdx.y * SB_TW, oy0 = blockIdx.z * SB_TH;
  // ---- 1. uint8 tile: input rows 4*oy0-3 .., bytes from (4*ox0-3)*3 (= 3 mod 4: tile byte 3 is that byte)
  const int row_words = a.Win * 3 / 4;
  const int w0 = (12 * ox0 - 9) >> 2;
  const uint32_t* im = reinterpret_cast<const uint32_t*>(a.img + (long)n * a.Hin * a.Win * 3);
  {  // a thread keeps its dword column and strides over rows.  ALL of its (up to 18) loads are requested before the first store:
     // unconditional, from clamped addresses, masked when stored.  (As a loop of `if (inside) v = load; store v` this was one
     // memory round trip per row -- 18 in a row, ~15 of a workgroup's 18 us.)
    const int c = tid & 127, r0 = tid >> 7;
    const int wi = w0 + c;
    const bool colok = c < SB_ROWW && wi >= 0 && wi < row_words;
    const int wic = wi < 0 ? 0 : (wi < row_words ? wi : row_words - 1);
    constexpr int NR = (SB_IR + 1) / 2;
    uint32_t v[NR];
#pragma unroll
    for (int k = 0; k < NR; ++k) {
      const int iy = 4 * oy0 - 3 + r0 + 2 * k;
      const int iyc = iy < 0 ? 0 : (iy < a.Hin ? iy : a.Hin - 1);
      v[k] = im[(long)iyc * row_words + wic];
    }
#pragma unroll
    for (int k = 0; k < NR; ++k) {
      const int r = r0 + 2 * k;
      const int iy = 4 * oy0 - 3 + r;
      const uint32_t m = (colok && iy >= 0 && iy < a.Hin) ? 0xffffffffu : 0u;
      if (c < SB_ROWW && r < SB_IR) in_tile[r * SB_ROWW + c] = v[k] & m;
    }
  }
  const half8 af0 = __builtin_bit_cast(half8, reinterpret_cast<const u32x4*>(a.afrag)[lane]);   // StemLayer::d_afrag_blk
  const half8 af1 = __builtin_bit_cast(half8, reinterpret_cast<const u32x4*>(a.afrag)[64 + lane]);
  const int c0 = (g & 1) * 4;
  const floatx4 sb4 = *reinterpret_cast<const floatx4*>(a.sbias + c0);
  __syncthreads();
  // ---- 2. stem on the matrix cores, two pixels per column (see stem_mfma_kernel); 17 x 33 pairs, 16 per MFMA tile
  const half2v k1024 = {(half_t)1024.f, (half_t)1024.f};
  constexpr int NPAIR = SB_SH * SB_PAIRS, NTILE = (NPAIR + 15) / 16;
  // block-uniform: every stem pixel of this tile lies inside the stem map (then no per-pixel test)
  const bool sb_interior = oy0 >= 1 && ox0 >= 1 && 2 * oy0 - 1 + SB_SH <= a.H1 && 2 * ox0 - 1 + SB_SW <= a.W1;
  // K group -> (window row ky, byte half h) of the 15-byte union window (StemLayer::build packs a_blk to match): step 0 holds
  // rows 0 and 2, step 1 row 1 (+ two zero-weight groups that re-read it).  The two K groups of a 32-lane ds_read_b32 group then
  // read rows two apart = 2 x 102 = 204 dwords = 12 banks apart (row pitch SB_ROWW = 102 dwords; 16 banks at round 3's first pitch of
  // 104, for which the simulation below was run): about half the bank conflicts of the (ky, h) = (q / 2, q % 2) order
  // (852 -> 522 LDS cycles per workgroup in a simulation of the access pattern).
  const int ky0 = 2 * (g & 1), hh = g >> 1;
  // pair index of this lane's column in tile t, as (row r, pair pp): t advances by 4 tiles = 64 pairs = 1 row + 31 pairs
  // (an interior and a border form of the loop, chosen per workgroup: the per-pixel map test is 8 of an iteration's ~90 instructions
  //  and four workgroups in five lie wholly inside the stem map)
  auto stem_loop = [&](auto interior) {
  int pi = wave * 16 + col;
  int r = pi >= SB_PAIRS ? 1 : 0, pp = pi - r * SB_PAIRS;   // wave * 16 + col < 64 < 2 * SB_PAIRS
  for (int t = wave; t < NTILE; t += 4) {
    const bool live = pi < NPAIR;
    const int rc = live ? r : SB_SH - 1, ppc = live ? pp : SB_PAIRS - 1;
    half8 bf[2];
    // window byte 3 + 12 pp + 8 h of tile row 2 r + ky: dword (2 r + ky) * ROWW + 3 pp + 2 h, byte 3 of it (32-bit index arithmetic:
    // as pointer arithmetic on the __shared__ array the compiler built 64-bit products)
    const int idx0 = 2 * rc * SB_ROWW + 3 * ppc + 2 * hh;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int idx = idx0 + (s == 0 ? ky0 : 1) * SB_ROWW;
      const uint32_t d0 = in_tile[idx], d1 = in_tile[idx + 1], d2 = in_tile[idx + 2];
      const uint32_t wa = __builtin_amdgcn_alignbyte(d1, d0, 3);   // bo & 3 == 3 always
      const uint32_t wb = __builtin_amdgcn_alignbyte(d2, d1, 3);
      // (the padded K groups of step 1 carry zero weights: whatever finite bytes they read contribute nothing)
      const uint32_t p0 = __builtin_amdgcn_perm(0x64646464u, wa, 0x04010400u);
      const uint32_t p1 = __builtin_amdgcn_perm(0x64646464u, wa, 0x04030402u);
      const uint32_t p2 = __builtin_amdgcn_perm(0x64646464u, wb, 0x04010400u);
      const uint32_t p3 = __builtin_amdgcn_perm(0x64646464u, wb, 0x04030402u);
      const half2v h0 = __builtin_bit_cast(half2v, p0) - k1024, h1 = __builtin_bit_cast(half2v, p1) - k1024;
      const half2v h2 = __builtin_bit_cast(half2v, p2) - k1024, h3 = __builtin_bit_cast(half2v, p3) - k1024;
      bf[s] = half8{h0[0], h0[1], h1[0], h1[1], h2[0], h2[1], h3[0], h3[1]};
    }
    floatx4 acc = {0.f, 0.f, 0.f, 0.f};
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(af0, bf[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(af1, bf[1], acc, 0, 0, 0);
    const int c = 2 * ppc + (g >> 1);
    if (live && c < SB_SW) {
      const floatx4 y4 = act4<half_t, ACT_SILU>(acc, sb4);
      // two packed converts, then the mask on the packed words (a select per half was four converts, four selects, two packs)
      const half2v q01 = {(half_t)y4[0], (half_t)y4[1]}, q23 = {(half_t)y4[2], (half_t)y4[3]};
      uint32_t m = 0xffffffffu;
      if constexpr (!decltype(interior)::value) {
        const int sy = 2 * oy0 - 1 + rc, sx = 2 * ox0 - 1 + c;
        m = (sy >= 0 && sy < a.H1 && sx >= 0 && sx < a.W1) ? 0xffffffffu : 0u;
      }
      typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
      *reinterpret_cast<u32x2*>(st_tile + (rc * SB_LW + c) * 16 + c0 * 2) = u32x2{__builtin_bit_cast(uint32_t, q01) & m, __builtin_bit_cast(uint32_t, q23) & m};
    }
    pi += 64;
    pp += 31;
    r += 1;
    if (pp >= SB_PAIRS) { pp -= SB_PAIRS; r += 1; }
  }
  };
  if (sb_interior) stem_loop(std::true_type{});   // (verified bit-identical to the border form on every workgroup)
  else stem_loop(std::false_type{});
  // ---- 3. stride-2 3x3 conv from the stem tile (K group = tap: 9 of 12 slots), then the 1x1 tail
  half8 a1[3];
#pragma unroll
  for (int s = 0; s < 3; ++s) a1[s] = __builtin_bit_cast(half8, reinterpret_cast<const u32x4*>(a.w1)[s * 64 + lane]);
  floatx4 bias1[1], bias2[1];
  bias1[0] = *reinterpret_cast<const floatx4*>(a.b1 + g * 4);
  half8 w2f[1][1];
  ConvArgs a2;
  a2.w2 = a.w2; a2.bias2 = a.b2;
  tail_load<half_t, 1, 1>(a2, lane, g, w2f, bias2);
  a2.out = a.out; a2.out_pitch = a.out_pitch; a2.Cout = a.C2; a2.act = a.act2; a2.res = nullptr; a2.res_pitch = 0;
  int toff[3];
#pragma unroll
  for (int s = 0; s < 3; ++s) {
    int tap = 4 * s + g;
    tap = tap > 8 ? 8 : tap;  // padded K slots: zero weights, any finite data
    const int ky = (tap * 21846) >> 16, kx = tap - 3 * ky;
    toff[s] = (ky * SB_LW + kx) * 16;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int t = wave * 4 + i;
    const int oy = t >> 1, ox = (t & 1) * 16 + col;
    const char* base = st_tile + ((2 * oy) * SB_LW + 2 * ox) * 16;
    floatx4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < 3; ++s)
      acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1[s], __builtin_bit_cast(half8, *reinterpret_cast<const u32x4*>(base + toff[s])), acc, 0, 0, 0);
    const int gy = oy0 + oy, gx = ox0 + ox;
    if (gy < a.H2 && gx < a.W2) {
      const long pix = (n * a.H2 + gy) * a.W2 + gx;
      floatx4 v[1] = {acc};
      tail_store<half_t, 1, 1, ACT_SILU>(a2, pix, g, v, bias1, w2f, bias2);
    }
  }
}

// ------------------------------------------------------------------------------------
// The same launch for v2's widths (the paper's YOLO-LitePi: stem 3 -> 16, stride-2 conv 16 -> 24, C2f.cv1 24 -> 12 | 12 =
// 32 physical channels; yolo_plus_ncnn_model/model.ncnn.param:3-8).  The stem runs in stem_mfma16_kernel's formulation (one
// pixel per MFMA column, the 27 window bytes in one K step); its 17 x 65 x 16 map lives in LDS as two planes of 8 channels
// (16 B per pixel each), so that a K group of the stride-2 conv -- (tap, 8-channel half), the direct kernel's order, whose
// packed fragments and 1x1 tail this kernel takes as they are -- is one ds_read_b128 at a static offset.
// Replaces stem_conv + conv3x3s2_direct+1x1<2,2>: the 320x320x16 map (210 MB per 64-image step each way) never exists.
// ------------------------------------------------------------------------------------
template <int TH_>
struct SB16 {
  static constexpr int TH = TH_, TW = 32, SH = 2 * TH + 1, SW = 2 * TW + 1, LW = 65, IR = 2 * SH + 1, ROWW = 102;
  static constexpr int PLANE = SH * LW * 16, NPIX = SH * SW, NTILE = (NPIX + 15) / 16;
};
template <int TH_>
__global__ __launch_bounds__(256) void stem_block16_kernel(const StemBlockArgs a) {
  typedef SB16<TH_> K;
  __shared__ uint32_t in_tile[K::IR * K::ROWW];
  __shared__ __attribute__((aligned(16))) char st_tile[2 * K::PLANE];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, col = lane & 15;
  const int n = blockIdx.x, ox0 = blockIdx.y * K::TW, oy0 = blockIdx.z * K::TH;
  // ---- 1. uint8 tile (as stem_block_kernel)
  const int row_words = a.Win * 3 / 4;
  const int w0 = (12 * ox0 - 9) >> 2;
  const uint32_t* im = reinterpret_cast<const uint32_t*>(a.img + (long)n * a.Hin * a.Win * 3);
  {
    const int c = tid & 127, r0 = tid >> 7;
    const int wi = w0 + c;
    const bool colok = c < K::ROWW && wi >= 0 && wi < row_words;
    const int wic = wi < 0 ? 0 : (wi < row_words ? wi : row_words - 1);
    constexpr int NR = (K::IR + 1) / 2;
    uint32_t v[NR];
#pragma unroll
    for (int k = 0; k < NR; ++k) {
      const int iy = 4 * oy0 - 3 + r0 + 2 * k;
      const int iyc = iy < 0 ? 0 : (iy < a.Hin ? iy : a.Hin - 1);
      v[k] = im[(long)iyc * row_words + wic];
    }
#pragma unroll
    for (int k = 0; k < NR; ++k) {
      const int r = r0 + 2 * k;
      const int iy = 4 * oy0 - 3 + r;
      const uint32_t m = (colok && iy >= 0 && iy < a.Hin) ? 0xffffffffu : 0u;
      if (c < K::ROWW && r < K::IR) in_tile[r * K::ROWW + c] = v[k] & m;
    }
  }
  const half8 af = __builtin_bit_cast(half8, reinterpret_cast<const u32x4*>(a.afrag)[lane]);   // StemLayer::d_afrag (16-channel form)
  const floatx4 sb4 = *reinterpret_cast<const floatx4*>(a.sbias + 4 * g);
  // stride-2 conv: 5 K steps x 2 channel tiles, the 1x1 tail: 2 tiles x 1 step (ConvLayer's packing for the direct kernel)
  half8 a1[5][2];
#pragma unroll
  for (int s = 0; s < 5; ++s)
#pragma unroll
    for (int t = 0; t < 2; ++t) a1[s][t] = __builtin_bit_cast(half8, reinterpret_cast<const u32x4*>(a.w1)[(s * 2 + t) * 64 + lane]);
  __syncthreads();
  // ---- 2. stem: SH x 65 pixels, 16 per MFMA, two tiles per iteration (their LDS reads are requested together: with three
  //         workgroups per CU one tile's read -> unpack -> MFMA -> SiLU -> store chain per iteration left the SIMDs idle)
  const half2v k1024 = {(half_t)1024.f, (half_t)1024.f};
  const bool sb_interior = oy0 >= 1 && ox0 >= 1 && 2 * oy0 - 1 + K::SH <= a.H1 && 2 * ox0 - 1 + K::SW <= a.W1;
  // K group g < 3: bytes 0..7 of window row g (three consecutive dwords, byte-aligned); g == 3: byte 8 of the three rows (the
  // dword two further on in each row, the same byte shift) -- one address form for all lanes: base + k * kstride
  const int kstride = g < 3 ? 1 : K::ROWW;
  const int gbase = g < 3 ? g * K::ROWW : 2;
  auto stem_tile = [&](int pi, int (&d)[3], int& rc, int& cc, int& sh, bool& live) {
    live = pi < K::NPIX;
    const int pc = live ? pi : K::NPIX - 1;
    rc = (pc * 1009) >> 16;              // pc / 65 for pc < 2 * 65 * 17 (1009 / 65536 = 1 / 64.95)
    cc = pc - rc * K::SW;
    if (cc >= K::SW) { cc -= K::SW; rc += 1; }
    const int bo = 3 + 6 * cc;           // first window byte of this pixel inside a staged row
    sh = bo & 3;
    const int idx = 2 * rc * K::ROWW + (bo >> 2) + gbase;
#pragma unroll
    for (int k = 0; k < 3; ++k) d[k] = (int)in_tile[idx + k * kstride];
  };
  auto stem_finish = [&](const int (&d)[3], int rc, int cc, int sh, bool live, auto interior) {   // interior: std::true_type for tiles wholly inside the stem map
    uint32_t wa, wb;
    {
      const uint32_t wa0 = __builtin_amdgcn_alignbyte((uint32_t)d[1], (uint32_t)d[0], sh);
      const uint32_t wb0 = __builtin_amdgcn_alignbyte((uint32_t)d[2], (uint32_t)d[1], sh);
      const uint32_t x0 = __builtin_amdgcn_alignbyte(0u, (uint32_t)d[0], sh), x1 = __builtin_amdgcn_alignbyte(0u, (uint32_t)d[1], sh),
                     x2 = __builtin_amdgcn_alignbyte(0u, (uint32_t)d[2], sh);
      const uint32_t y01 = __builtin_amdgcn_perm(x1, x0, 0x0c0c0400u);   // byte 0 of x0, byte 0 of x1, 0, 0
      const uint32_t wa3 = __builtin_amdgcn_perm(x2, y01, 0x0c040100u);  // + byte 0 of x2
      wa = g < 3 ? wa0 : wa3;
      wb = g < 3 ? wb0 : 0u;
    }
    const uint32_t p0 = __builtin_amdgcn_perm(0x64646464u, wa, 0x04010400u);
    const uint32_t p1 = __builtin_amdgcn_perm(0x64646464u, wa, 0x04030402u);
    const uint32_t p2 = __builtin_amdgcn_perm(0x64646464u, wb, 0x04010400u);
    const uint32_t p3 = __builtin_amdgcn_perm(0x64646464u, wb, 0x04030402u);
    const half2v h0 = __builtin_bit_cast(half2v, p0) - k1024, h1 = __builtin_bit_cast(half2v, p1) - k1024;
    const half2v h2 = __builtin_bit_cast(half2v, p2) - k1024, h3 = __builtin_bit_cast(half2v, p3) - k1024;
    const half8 bf = half8{h0[0], h0[1], h1[0], h1[1], h2[0], h2[1], h3[0], h3[1]};
    floatx4 acc = {0.f, 0.f, 0.f, 0.f};
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(af, bf, acc, 0, 0, 0);
    if (live) {
      const floatx4 y4 = act4<half_t, ACT_SILU>(acc, sb4);
      const half2v q01 = {(half_t)y4[0], (half_t)y4[1]}, q23 = {(half_t)y4[2], (half_t)y4[3]};
      uint32_t m = 0xffffffffu;
      if constexpr (!decltype(interior)::value) {   // border tiles: stem pixels outside the map are the next conv's zero padding
        const int sy = 2 * oy0 - 1 + rc, sx = 2 * ox0 - 1 + cc;
        m = (sy >= 0 && sy < a.H1 && sx >= 0 && sx < a.W1) ? 0xffffffffu : 0u;
      }
      typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
      // channels 4g .. 4g+3: plane g >> 1, bytes 8 (g & 1) of the pixel
      *reinterpret_cast<u32x2*>(st_tile + (g >> 1) * K::PLANE + (rc * K::LW + cc) * 16 + (g & 1) * 8) =
          u32x2{__builtin_bit_cast(uint32_t, q01) & m, __builtin_bit_cast(uint32_t, q23) & m};
    }
  };
  auto stem_loop = [&](auto interior) {
    for (int t = wave; t < K::NTILE; t += 8) {
      int dA[3], dB[3], rA, cA, sA, rB, cB, sB;
      bool lA, lB;
      stem_tile(t * 16 + col, dA, rA, cA, sA, lA);
      stem_tile((t + 4) * 16 + col, dB, rB, cB, sB, lB);
      stem_finish(dA, rA, cA, sA, lA, interior);
      stem_finish(dB, rB, cB, sB, lB, interior);
    }
  };
  if (sb_interior) stem_loop(std::true_type{});   // (block-uniform: 80 % of the workgroups; the per-pixel test was 9 of a tile's ~80 instructions: 123 -> 120 us)
  else stem_loop(std::false_type{});
  // ---- 3. stride-2 3x3 conv from the stem planes (K group q = 4 s + g = (tap q / 2, channel half q % 2)), then the 1x1 tail
  floatx4 bias1[2], bias2[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) bias1[t] = *reinterpret_cast<const floatx4*>(a.b1 + g * 8 + t * 4);
  half8 w2f[2][1];
  ConvArgs a2;
  a2.w2 = a.w2; a2.bias2 = a.b2;
  tail_load<half_t, 2, 2>(a2, lane, g, w2f, bias2);
  a2.out = a.out; a2.out_pitch = a.out_pitch; a2.Cout = a.C2; a2.act = a.act2; a2.res = nullptr; a2.res_pitch = 0;
  int toff[5];
#pragma unroll
  for (int s = 0; s < 5; ++s) {
    const int q = 4 * s + g;
    int tap = q >> 1;
    tap = tap > 8 ? 8 : tap;   // padded K slots: zero weights, any finite data
    const int ky = (tap * 21846) >> 16, kx = tap - 3 * ky;
    toff[s] = (q & 1) * K::PLANE + (ky * K::LW + kx) * 16;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < K::TH / 2; ++i) {
    const int t = wave * (K::TH / 2) + i;
    const int oy = t >> 1, ox = (t & 1) * 16 + col;
    const char* base = st_tile + ((2 * oy) * K::LW + 2 * ox) * 16;
    floatx4 v[2] = {floatx4{0.f, 0.f, 0.f, 0.f}, floatx4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int s = 0; s < 5; ++s) {
      const half8 bf = __builtin_bit_cast(half8, *reinterpret_cast<const u32x4*>(base + toff[s]));
#pragma unroll
      for (int tt = 0; tt < 2; ++tt) v[tt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1[s][tt], bf, v[tt], 0, 0, 0);
    }
    const int gy = oy0 + oy, gx = ox0 + ox;
    if (gy < a.H2 && gx < a.W2) {
      const long pix = (n * a.H2 + gy) * a.W2 + gx;
      tail_store<half_t, 2, 2, ACT_SILU>(a2, pix, g, v, bias1, w2f, bias2);
    }
  }
}

// ====================================================================================
// Host side: weight packing and launch
// ====================================================================================
static size_t elem_size(int prec) { return prec == LP_FP16 ? 2 : 4; }
static unsigned div_magic(int d) { return d <= 1 ? 0xffffffffu : (unsigned)((1ull << 32) / (unsigned)d); }
static bool span_ok(int N, const View& v) { return !v.base || (double)N * v.H * v.W * v.pitch < 2147483648.0; }

// LDS geometry of the 3x3 input tile, found by enumerating the ds_read_b128 lane groups
// ({0-3,12-15,20-27}, ...; MI355X_MICROARCH.md, LDS) over every K step: with cg channel groups
// per pixel, a pixel pitch of p = (smallest value >= cg with p % 4 == 2) 16-byte slots and a row
// width = 12 (mod 16) pixels make the 4x4-pixel x 4-K-group fragment read conflict-free (4 LDS
// cycles); the former odd-pitch rule cost 8.  One group per pixel (cg = 1): pitch 1, row = 4 (mod 16).
static int lds_pixel_slots(int cg, int stride = 1) {
  if (cg <= 1) return 1;
  int p = cg;
  if (stride == 2) {  // patches of every other pixel: an odd pitch (and a row width = 4 mod 8) is the conflict-free one
    while (p % 2 != 1) ++p;
    return p;
  }
  while (p % 4 != 2) ++p;
  return p;
}
static int lds_row_width(int iw, int cg, int stride = 1) {
  int lw = iw;
  if (stride == 2 && cg > 1) {
    while (lw % 8 != 4) ++lw;
    return lw;
  }
  const int want = cg <= 1 ? 4 : 12;
  while (lw % 16 != want) ++lw;
  return lw;
}

static void put_elem(std::vector<uint8_t>& buf, size_t idx, int prec, float v) {
  if (prec == LP_FP16) {
    uint16_t h = f32_to_f16(v);
    memcpy(&buf[idx * 2], &h, 2);
  } else {
    memcpy(&buf[idx * 4], &v, 4);
  }
}

void ConvLayer::build(int prec_, int impl_, int k_, int stride_, int cin, int cout, int act_,
                      const std::vector<float>& w_phys, const std::vector<float>& bias_phys, int hout, int wout, int batch_hint,
                      bool full_n, bool single_chunk) {
  prec = prec_; impl = impl_; k = k_; stride = stride_; Cin = cin; Cout = cout; act = act_;
  LP_CHECK(k == 1 || k == 3, LP_ERR_GRAPH, "conv kernel size %d unsupported", k);
  LP_CHECK(Cin % 8 == 0 && Cout % 8 == 0, LP_ERR_GRAPH, "physical channels must be multiples of 8");
  const int taps = k * k;
  const int G = prec == LP_FP16 ? 8 : 4;
  const size_t es = elem_size(prec);
  const int tiles_total = ceil_div(Cout, 16);
  w_host = w_phys;
  b_host = bias_phys;

  // bias, padded so every quad load is in bounds
  std::vector<float> b(round_up(Cout, 64) + 64, 0.f);
  for (int c = 0; c < Cout; ++c) b[c] = bias_phys.empty() ? 0.f : bias_phys[c];
  d_bias.alloc(b.size() * 4);
  LP_HIP(hipMemcpy(d_bias.p, b.data(), b.size() * 4, hipMemcpyHostToDevice));

  if (impl == IMPL_NAIVE) {
    std::vector<uint8_t> buf((size_t)Cout * taps * Cin * es);
    for (size_t i = 0; i < (size_t)Cout * taps * Cin; ++i) put_elem(buf, i, prec, w_phys[i]);
    d_w.alloc(buf.size());
    LP_HIP(hipMemcpy(d_w.p, buf.data(), buf.size(), hipMemcpyHostToDevice));
    return;
  }

  // channel tiles per block: 4 when the layer has plenty of pixel tiles; fewer (more channel splits -> more
  // workgroups, but the input is read once per split) when the map is small.  The floor of 128 workgroups is
  // measured on the pipelined benchmark (3 batches in flight share the GPU: 512 gave the best isolated layer
  // times but 5 % less throughput; LITEPI_MIN_WGS overrides it for such sweeps)
  auto pick_nt = [&](long pixel_blocks) {
    if (full_n) return tiles_total;  // a fused tail needs every intermediate channel in one workgroup
    int nt = tiles_total >= 4 ? 4 : tiles_total;
    if (tiles_total % 4 != 0 && tiles_total > 4) nt = (tiles_total % 3 == 0) ? 3 : 4;
    static const long min_wgs = getenv("LITEPI_MIN_WGS") ? atol(getenv("LITEPI_MIN_WGS")) : 128;
    while (nt > 1 && pixel_blocks * ceil_div(tiles_total, nt) < min_wgs) nt = (nt == 4 && tiles_total % 4 == 0) ? 2 : nt - 1;
    return nt;
  };
  const int B = batch_hint > 0 ? batch_hint : 1;

  // stride-2: gather straight from global memory (conv3x3s2_direct_kernel).  LITEPI_S2_STAGED=<min Cin> routes deep-K
  // layers without a fused tail through the LDS-staged kernel instead (A/B switch; slower since the gather kernel
  // keeps its loads a few K steps ahead)
  static const int s2_staged_min_cin = getenv("LITEPI_S2_STAGED") ? atoi(getenv("LITEPI_S2_STAGED")) : 1 << 30;
  direct = (k == 3 && stride == 2) && (full_n || Cin < s2_staged_min_cin);
  if (single_chunk) {
    // weights only (BottleneckPair): every channel tile in one workgroup, all of K in one chunk
    LP_CHECK(k == 3 && stride == 1, LP_ERR_STATE, "single-chunk packing is for 3x3 stride-1 layers");
    direct = false;
    NT = tiles_total; nsplits = 1; CK = Cin; CGc = Cin / G; nchunks = 1;
    steps = ceil_div(taps * CGc, 4);
    PS = lds_pixel_slots(CGc) * 16;
    LP_CHECK(steps * 4 <= 128, LP_ERR_GRAPH, "conv3x3: too many K steps for one chunk");
  } else if (direct) {
    // stride-2: no input staging (see conv3x3s2_direct_kernel); all of K for this block's channel split in LDS
    NT = pick_nt(ceil_div((long)B * hout * wout, 256));
    CK = Cin;
    CGc = Cin / G;
    nchunks = 1;
    steps = ceil_div(taps * CGc, 4);
    while (!full_n && NT > 1 && (size_t)steps * NT * 1024 > 64 * 1024) --NT;
    nsplits = ceil_div(tiles_total, NT);
    lds_bytes = (size_t)steps * NT * 1024 + (size_t)steps * 4 * 8;  // weight fragments + K-group table
    LP_CHECK(lds_bytes <= 160 * 1024, LP_ERR_GRAPH, "conv3x3/s2 weights do not fit LDS (%zu B)", lds_bytes);
  } else if (k == 3) {
    // Tile shape (bwh x bww waves of 4x20 pixels) and K chunk CK (multiple of 8, divides Cin):
    // the halo'd input tile plus the chunk's weight fragments must fit the LDS budget (two
    // workgroups per CU).  Priority: no idle lanes on this map, >= 16-channel chunks, then the
    // largest workgroup (weights are staged once per workgroup), then the largest chunk.
    const bool small_map = (hout <= 20 && wout <= 20);
    const int cand[6][2] = {{5, 1}, {2, 2}, {2, 1}, {1, 2}, {4, 1}, {1, 1}};
    static const size_t budget = (getenv("LITEPI_LDS_KB") ? atol(getenv("LITEPI_LDS_KB")) : 80) * 1024;
    long best_score = -1;
    CK = 8; bwh = 1; bww = 1; LW = 0; NT = 1;
    for (int ci = 0; ci < 6; ++ci) {
      const int h = cand[ci][0], w = cand[ci][1];
      if (!small_map && (h == 5 || h == 4)) continue;
      const int TH = 4 * h, TW = 20 * w;
      const int IH = (TH - 1) * stride + 3, IW = (TW - 1) * stride + 3;
      const int tiles = ceil_div(hout, TH) * ceil_div(wout, TW);
      const int nt = pick_nt((long)tiles * B);
      // K chunk: everything at once when it fits (one round trip, no K padding between chunks), else the
      // largest chunk whose two buffers fit (the next chunk's LDS-DMA flies during this chunk's MFMAs)
      int ck_fit = 0, lw = 0;
      for (int ck = 8; ck <= Cin && ck <= 64; ck += 8) {
        if (Cin % ck) continue;
        const int cgc = ck / G;
        const int l = lds_row_width(IW, cgc, stride);
        const size_t one = (size_t)ceil_div(taps * cgc, 4) * nt * 1024 + (size_t)IH * l * lds_pixel_slots(cgc, stride) * 16;
        const size_t lds = 512 + (ck == Cin ? one : 2 * one);
        if (lds <= budget && ceil_div(taps * cgc, 4) * 4 <= 128 && ceil_div(l * lds_pixel_slots(cgc, stride), 64) <= 8) { ck_fit = ck; lw = l; }
      }
      if (!ck_fit) continue;
      const int util = (int)(100.0 * hout * wout / ((double)tiles * TH * TW));
      const long score = (util >= 95 ? 3 : util >= 80 ? 2 : util >= 60 ? 1 : 0) * 100000L +
                         ((ck_fit >= 16 || ck_fit == Cin) ? 10000L : 0L) + (long)h * w * 1000L + ck_fit;
      if (score > best_score) { best_score = score; CK = ck_fit; bwh = h; bww = w; LW = lw; NT = nt; }
    }
    LP_CHECK(best_score >= 0, LP_ERR_GRAPH, "conv3x3 %d->%d: no tile fits LDS", Cin, Cout);
    nsplits = ceil_div(tiles_total, NT);
    const int IH = (4 * bwh - 1) * stride + 3;
    CGc = CK / G;
    PS = lds_pixel_slots(CGc, stride) * 16;
    nchunks = Cin / CK;
    steps = ceil_div(taps * CGc, 4);
    LP_CHECK(steps * 4 <= 128, LP_ERR_GRAPH, "conv3x3: too many K steps per chunk");
    lds_bytes = 512 + (nchunks > 1 ? 2 : 1) * ((size_t)steps * NT * 1024 + (size_t)IH * LW * PS);
    LP_CHECK(ceil_div(LW * (PS / 16), 64) <= 8, LP_ERR_GRAPH, "conv3x3: LDS tile row too wide");
    auto rcp16 = [&](int d, int range) {  // x / d == (x * m) >> 16 for x < range (kernel prologue)
      const unsigned m = (65536u + d - 1) / d;
      for (int x = 0; x < range; ++x) LP_CHECK((int)((x * m) >> 16) == x / d, LP_ERR_STATE, "reciprocal of %d not exact at %d", d, x);
      return m;
    };
    rcp_cg = rcp16(CGc, 128);
    rcp_ps = rcp16(PS / 16, 512);
    rcp_pcs = rcp16(ceil_div(LW * (PS / 16), 64), 1024);
    LP_CHECK(lds_bytes <= 160 * 1024, LP_ERR_GRAPH, "conv3x3 tile does not fit LDS (%zu B)", lds_bytes);
  } else {
    NT = pick_nt(ceil_div((long)B * hout * wout, 256));
    CK = Cin;
    CGc = Cin / G;
    nchunks = 1;
    steps = ceil_div(CGc, 4);
    // all K of this block's channel split stays in LDS: fewer channel tiles per block for deep K
    while (NT > 1 && (size_t)steps * NT * 1024 > 96 * 1024) --NT;
    nsplits = ceil_div(tiles_total, NT);
    lds_bytes = (size_t)steps * NT * 1024;
    LP_CHECK(lds_bytes <= 160 * 1024, LP_ERR_GRAPH, "conv1x1 weights do not fit LDS (%zu B)", lds_bytes);
  }

  // fragment-ordered weights: [split][chunk][step][tile][lane][G]
  const size_t nfrag = (size_t)nsplits * nchunks * steps * NT * 64;
  std::vector<uint8_t> buf(nfrag * 16, 0);
  for (int ns = 0; ns < nsplits; ++ns)
    for (int ch = 0; ch < nchunks; ++ch)
      for (int s = 0; s < steps; ++s)
        for (int t = 0; t < NT; ++t)
          for (int lane = 0; lane < 64; ++lane) {
            const int g = lane >> 4, m = lane & 15;
            const int gm = m >> 2, r = m & 3;
            const int oc = ns * 16 * NT + gm * 4 * NT + t * 4 + r;  // row permutation (see header)
            const int q = 4 * s + g;
            const int tap = q / CGc, cg = q % CGc;
            if (tap >= taps || oc >= Cout) continue;
            const size_t f = ((((size_t)ns * nchunks + ch) * steps + s) * NT + t) * 64 + lane;
            for (int j = 0; j < G; ++j) {
              const int ci = ch * CK + cg * G + j;
              put_elem(buf, f * G + j, prec, w_phys[((size_t)oc * taps + tap) * Cin + ci]);
            }
          }
  d_w.alloc(buf.size());
  LP_HIP(hipMemcpy(d_w.p, buf.data(), buf.size(), hipMemcpyHostToDevice));
}

bool ConvLayer::tail_supported(int k, int stride, int cmid_phys, int cout2_phys) {
  if (k != 3 || cmid_phys % 8 != 0) return false;
  const int nt = ceil_div(cmid_phys, 16), t2 = ceil_div(cout2_phys, 16);  // a half-filled last tile has zero weights and bias
  if (stride == 1) return (nt == 4 && t2 == 4) || (nt == 2 && t2 == 1);
  if (stride == 2) return (nt == 1 && t2 == 1) || (nt == 2 && t2 == 2) || (nt == 4 && t2 == 4);
  return false;
}

void ConvLayer::attach_tail(int cout2_phys, int act2_, const std::vector<float>& w2_phys, const std::vector<float>& bias2_phys) {
  LP_CHECK(impl == IMPL_MFMA && nsplits == 1 && NT == ceil_div(Cout, 16) && tail_supported(k, stride, Cout, cout2_phys), LP_ERR_STATE,
           "conv %s: fused 1x1 tail not supported for this shape", name.c_str());
  T2 = ceil_div(cout2_phys, 16);
  Cout2 = cout2_phys;
  act2 = act2_;
  w2_host = w2_phys;
  b2_host = bias2_phys;
  const int G = prec == LP_FP16 ? 8 : 4;
  const int S2 = prec == LP_FP16 ? (NT + 1) / 2 : NT;
  std::vector<uint8_t> buf((size_t)T2 * S2 * 64 * 16, 0);
  for (int t = 0; t < T2; ++t)
    for (int s2 = 0; s2 < S2; ++s2)
      for (int lane = 0; lane < 64; ++lane) {
        const int g = lane >> 4, m = lane & 15, gm = m >> 2, r = m & 3;
        const int oc = gm * 4 * T2 + t * 4 + r;  // same row permutation as the main GEMM, for vector stores
        if (oc >= cout2_phys) continue;
        for (int j = 0; j < G; ++j) {
          if (G * s2 + j >= 4 * NT) continue;       // half-filled last K step (odd NT, fp16)
          const int mid = g * 4 * NT + G * s2 + j;  // the intermediate channel this lane's accumulators hold at (s2, j)
          if (mid >= Cout) continue;                // padding rows of a half-filled last channel tile
          put_elem(buf, ((size_t)(t * S2 + s2) * 64 + lane) * G + j, prec, w2_phys[(size_t)oc * Cout + mid]);
        }
      }
  d_w2.alloc(buf.size());
  LP_HIP(hipMemcpy(d_w2.p, buf.data(), buf.size(), hipMemcpyHostToDevice));
  std::vector<float> b(round_up(cout2_phys, 64) + 64, 0.f);
  for (int c = 0; c < cout2_phys && c < (int)bias2_phys.size(); ++c) b[c] = bias2_phys[c];
  d_bias2.alloc(b.size() * 4);
  LP_HIP(hipMemcpy(d_bias2.p, b.data(), b.size() * 4, hipMemcpyHostToDevice));
}

template <typename T, int NT>
static void launch3x3(const ConvArgs& a, int stride, dim3 grid, int threads, size_t lds, hipStream_t st) {
  if (stride == 1) {
    set_max_dynamic_lds(reinterpret_cast<const void*>(conv3x3_mfma_kernel<T, NT, 1>), 160 * 1024);
    LP_LAUNCH((conv3x3_mfma_kernel<T, NT, 1>), grid, dim3(threads), lds, st, a);
  } else {
    set_max_dynamic_lds(reinterpret_cast<const void*>(conv3x3_mfma_kernel<T, NT, 2>), 160 * 1024);
    LP_LAUNCH((conv3x3_mfma_kernel<T, NT, 2>), grid, dim3(threads), lds, st, a);
  }
}

template <typename T, int NT, int U>
static void launch_s2_u(const ConvArgs& a, dim3 grid, size_t lds, hipStream_t st) {
  set_max_dynamic_lds(reinterpret_cast<const void*>(conv3x3s2_direct_kernel<T, NT, 4, 0, U>), 160 * 1024);
  LP_LAUNCH((conv3x3s2_direct_kernel<T, NT, 4, 0, U>), grid, dim3(256), lds, st, a);
}
template <typename T, int NT>
static void launch_s2(const ConvArgs& a, dim3 grid, size_t lds, hipStream_t st) {
  if (a.steps >= 5) launch_s2_u<T, NT, 4>(a, grid, lds, st);
  else launch_s2_u<T, NT, 1>(a, grid, lds, st);
}

template <typename T, int NT, int NP, int EPI>
static void launch1x1_(const ConvArgs& a, dim3 grid, size_t lds, hipStream_t st) {
  set_max_dynamic_lds(reinterpret_cast<const void*>(conv1x1_mfma_kernel<T, NT, NP, EPI>), 160 * 1024);
  LP_LAUNCH((conv1x1_mfma_kernel<T, NT, NP, EPI>), grid, dim3(256), lds, st, a);
}

template <typename T, int NT>
static void launch1x1(const ConvArgs& a, bool shuffle, dim3 grid, size_t lds, hipStream_t st) {
  if (a.up) {
    set_max_dynamic_lds(reinterpret_cast<const void*>(conv1x1_mfma_kernel<T, NT, 4, EPI_PLAIN, true>), 160 * 1024);
    LP_LAUNCH((conv1x1_mfma_kernel<T, NT, 4, EPI_PLAIN, true>), grid, dim3(256), lds, st, a);
  } else if (shuffle)
    launch1x1_<T, NT, 4, EPI_SHUFFLE>(a, grid, lds, st);
  else
    launch1x1_<T, NT, 4, EPI_PLAIN>(a, grid, lds, st);
}

void ConvLayer::launch(const ConvIO& io, hipStream_t st) const {
  ConvArgs a;
  memset(&a, 0, sizeof(a));
  a.in = io.in.base; a.out = io.out.base; a.res = io.res.base; a.x1 = io.x1.base;
  a.wpk = d_w.p; a.bias = d_bias.as<float>(); a.m_dyn = io.m_dyn;
  a.N = io.N; a.Hin = io.in.H; a.Win = io.in.W; a.Hout = io.out.H; a.Wout = io.out.W;
  a.in_pitch = io.in.pitch; a.out_pitch = io.out.pitch; a.res_pitch = io.res.pitch; a.x1_pitch = io.x1.pitch;
  a.Cin = Cin; a.Cout = Cout; a.act = act;
  a.M = io.N * io.out.H * io.out.W; a.pix_per_item = io.out.H * io.out.W;
  a.magic_hw = div_magic(a.pix_per_item); a.magic_w = div_magic(io.out.W);
  // the kernels address with 32-bit element offsets (eoff)
  LP_CHECK(span_ok(io.N, io.in) && span_ok(io.N, io.out) && span_ok(io.N, io.res) && span_ok(io.N, io.x1) && span_ok(io.N, io.up), LP_ERR_ARG,
           "conv %s: a tensor of batch %d exceeds 2^31 elements", name.c_str(), io.N);
  a.CK = CK; a.nchunks = nchunks; a.steps_per_chunk = steps; a.CGc = CGc; a.LW = LW; a.PS = PS;
  a.bwh = bwh; a.bww = bww; a.steps = steps; a.nsplit_tiles = NT;
  a.half_c = io.half_c; a.half_cp = io.half_cp; a.out_f32 = io.out_f32; a.stamps = io.stamps; a.res_first = io.res_first;
  LP_CHECK(!io.res_first || (io.res.base && !io.x1.base && !io.out_f32 && T2 == 0), LP_ERR_STATE, "conv %s: residual-before-activation needs a plain epilogue", name.c_str());
  a.w2 = d_w2.p; a.bias2 = d_bias2.as<float>(); a.act2 = act2; a.Cout2 = Cout2;
  a.zeros = d_bias.as<float>() + round_up(Cout, 64);  // the bias buffer ends in 64 zero floats
  LP_CHECK(io.in.C + (io.up.base ? io.up.C : 0) == Cin, LP_ERR_STATE, "conv input view has %d channels, layer expects %d", io.in.C, Cin);
  if (io.up.base) {
    const int Gk = prec == LP_FP16 ? 8 : 4;
    LP_CHECK(k == 1 && impl == IMPL_MFMA && !io.x1.base && !io.m_dyn && !io.out_f32 && io.up.C % Gk == 0 && (io.up.pitch % Gk) == 0 &&
                 io.up.H * 2 == io.out.H && io.up.W * 2 == io.out.W, LP_ERR_STATE, "conv %s: fused upsample source does not fit", name.c_str());
    a.up = io.up.base; a.up_pitch = io.up.pitch; a.up_cg = io.up.C / Gk;
  }
  LP_CHECK(io.x1.base || io.out.C >= (T2 ? Cout2 : Cout) || io.out_f32, LP_ERR_STATE, "conv output view too narrow (%d < %d)", io.out.C, T2 ? Cout2 : Cout);
  LP_CHECK((io.in.pitch % 8) == 0 && (io.out.pitch % 4) == 0, LP_ERR_STATE, "unaligned channel pitch");
  const bool f16 = prec == LP_FP16;

  if (impl == IMPL_NAIVE) {
    const long total = (long)a.M * Cout;
    dim3 grid((unsigned)((total + 255) / 256));
    if (f16)
      LP_LAUNCH(conv_naive_kernel<half_t>, grid, dim3(256), 0, st, a, k, stride);
    else
      LP_LAUNCH(conv_naive_kernel<float>, grid, dim3(256), 0, st, a, k, stride);
    LP_HIP(hipGetLastError());
    return;
  }

  if (direct) {
    LP_CHECK(!io.x1.base && !io.out_f32, LP_ERR_STATE, "conv3x3/s2: unsupported epilogue");
    const long ntiles = ((long)a.M + 255) / 256;
    dim3 grid((unsigned)(ntiles < 4096 ? (ntiles > 0 ? ntiles : 1) : 4096), nsplits);
#define LP_LD(TT)                                                                                      \
  switch (NT) {                                                                                        \
    case 1: launch_s2<TT, 1>(a, grid, lds_bytes, st); break;                                           \
    case 2: launch_s2<TT, 2>(a, grid, lds_bytes, st); break;                                           \
    case 3: launch_s2<TT, 3>(a, grid, lds_bytes, st); break;                                           \
    default: launch_s2<TT, 4>(a, grid, lds_bytes, st); break;                                          \
  }
    if (T2) {
      LP_CHECK(!io.res.base, LP_ERR_STATE, "fused tail with residual unsupported");
#define LP_LDT_(TT, N_, T_, U_)                                                                                              \
  {                                                                                                                          \
    set_max_dynamic_lds(reinterpret_cast<const void*>(conv3x3s2_direct_kernel<TT, N_, 4, T_, U_>), 160 * 1024);                                                                                                              \
    LP_LAUNCH((conv3x3s2_direct_kernel<TT, N_, 4, T_, U_>), grid, dim3(256), lds_bytes, st, a);                      \
  }
#define LP_LDT(TT, N_, T_)                               \
  {                                                      \
    if (a.steps >= 5) LP_LDT_(TT, N_, T_, 4) else LP_LDT_(TT, N_, T_, 1) \
  }
      if (NT == 1 && T2 == 1) { if (f16) LP_LDT(half_t, 1, 1) else LP_LDT(float, 1, 1) }
      else if (NT == 2 && T2 == 2) { if (f16) LP_LDT(half_t, 2, 2) else LP_LDT(float, 2, 2) }
      else if (NT == 4 && T2 == 4) { if (f16) LP_LDT(half_t, 4, 4) else LP_LDT(float, 4, 4) }
      else throw Error(LP_ERR_STATE, "conv3x3/s2: unsupported fused tail shape");
#undef LP_LDT_
#undef LP_LDT
    } else if (f16) { LP_LD(half_t) } else { LP_LD(float) }
#undef LP_LD
  } else if (k == 3) {
    LP_CHECK(!io.x1.base && !io.out_f32, LP_ERR_STATE, "conv3x3: unsupported epilogue");
    const int TH = 4 * bwh, TW = 20 * bww;
    a.tiles_x = ceil_div(a.Wout, TW);
    a.tiles_y = ceil_div(a.Hout, TH);
    static const bool tile_major = getenv("LITEPI_TILE_MAJOR") != nullptr;  // A/B switch: pre-XCD-aware block order
    a.tile_major = tile_major;
    dim3 grid(io.N, a.tiles_x * a.tiles_y, nsplits);
    if (tile_major) grid = dim3(a.tiles_x * a.tiles_y, io.N, nsplits);
    const int threads = 64 * bwh * bww;
    // 16-bit reciprocals for the kernel's prologue: x / d == (x * ceil(65536 / d)) >> 16 on the ranges used
    auto rcp16 = [&](int d, int range) {
      const unsigned m = (65536u + d - 1) / d;
      for (int x = 0; x < range; ++x) LP_CHECK((int)((x * m) >> 16) == x / d, LP_ERR_STATE, "reciprocal of %d not exact at %d", d, x);
      return m;
    };
    a.rcp_tx = rcp16(a.tiles_x, a.tiles_x * a.tiles_y);
    a.rcp_cg = rcp_cg; a.rcp_ps = rcp_ps; a.rcp_pcs = rcp_pcs;  // checked once in build()
#define LP_L3(TT)                                                                  \
  switch (NT) {                                                                    \
    case 1: launch3x3<TT, 1>(a, stride, grid, threads, lds_bytes, st); break;      \
    case 2: launch3x3<TT, 2>(a, stride, grid, threads, lds_bytes, st); break;      \
    case 3: launch3x3<TT, 3>(a, stride, grid, threads, lds_bytes, st); break;      \
    default: launch3x3<TT, 4>(a, stride, grid, threads, lds_bytes, st); break;     \
  }
    if (T2) {
      LP_CHECK(!io.res.base && stride == 1, LP_ERR_STATE, "fused tail with residual unsupported");
#define LP_L3T(TT, N_, T_)                                                                                              \
  {                                                                                                                     \
    set_max_dynamic_lds(reinterpret_cast<const void*>(conv3x3_mfma_kernel<TT, N_, 1, T_>), 160 * 1024);                                                                                                         \
    LP_LAUNCH((conv3x3_mfma_kernel<TT, N_, 1, T_>), grid, dim3(threads), lds_bytes, st, a);                      \
  }
      if (NT == 4 && T2 == 4) { if (f16) LP_L3T(half_t, 4, 4) else LP_L3T(float, 4, 4) }
      else if (NT == 2 && T2 == 1) { if (f16) LP_L3T(half_t, 2, 1) else LP_L3T(float, 2, 1) }
      else throw Error(LP_ERR_STATE, "conv3x3: unsupported fused tail shape");
#undef LP_L3T
    } else if (f16) { LP_L3(half_t) } else { LP_L3(float) }
#undef LP_L3
  } else {
    LP_CHECK(stride == 1, LP_ERR_GRAPH, "strided 1x1 conv unsupported");
    const long ntiles = ((long)a.M + 255) / 256;
    dim3 grid((unsigned)(ntiles < 4096 ? (ntiles > 0 ? ntiles : 1) : 4096), nsplits);
    if (io.m_dyn) grid.x = 1024 / nsplits > 64 ? 1024 / nsplits : 64;
    const bool shuffle = io.x1.base != nullptr;
#define LP_L1(TT)                                                           \
  switch (NT) {                                                             \
    case 1: launch1x1<TT, 1>(a, shuffle, grid, lds_bytes, st); break;       \
    case 2: launch1x1<TT, 2>(a, shuffle, grid, lds_bytes, st); break;       \
    case 3: launch1x1<TT, 3>(a, shuffle, grid, lds_bytes, st); break;       \
    default: launch1x1<TT, 4>(a, shuffle, grid, lds_bytes, st); break;      \
  }
    if (f16) { LP_L1(half_t) } else { LP_L1(float) }
#undef LP_L1
  }
  LP_HIP(hipGetLastError());
}


// ---- fused bottleneck pair ---------------------------------------------------------------------
bool BottleneckPair::cv2_shape(int prec, int c, const Cv2& cv2, int& t2, int& kg, int& sg) {
  const int G = prec == LP_FP16 ? 8 : 4;
  const int nt = ceil_div(c, 16);
  if (cv2.cat_global <= 0 || cv2.cat_global % G != 0 || cv2.c3 % 8 != 0) return false;
  t2 = ceil_div(cv2.c3, 16);
  kg = cv2.cat_global / G;
  sg = ceil_div(kg, 4);
  // kernel instantiations: NT = 1 with T2 in {1, 2}, NT = 2 with T2 in {2, 4}; at most 3 (fp16) / 6 (fp32) gathered K steps
  if (!((nt == 1 && (t2 == 1 || t2 == 2)) || (nt == 2 && (t2 == 2 || t2 == 4)))) return false;
  return sg <= (prec == LP_FP16 ? 3 : 6);
}

bool BottleneckPair::plan(int prec, int c, int h, int w, int batch_hint, size_t extra_lds, bool tail, int& th, int& tw, int& lw, size_t& lds,
                          int kg_cl, bool* cl_out, int* psa_out) {
  if (cl_out) *cl_out = false;
  const int G = prec == LP_FP16 ? 8 : 4;
  if (c % 8 != 0 || c > 64) return false;
  const int nt = ceil_div(c, 16), cg = c / G;
  if (nt > 4) return false;
  const int steps = ceil_div(9 * cg, 4);
  if (steps * 4 > 128) return false;
  const int pss = lds_pixel_slots(cg);
  // 20x40 / 10x20: 8 tiles on an 80x80 / 40x40 map -- 512 workgroups for a batch of 64, exactly two per CU, instead of 640 (2.5
  // per CU: the CUs that get three decide the kernel time of these single-round grids)
  const int cand[6][2] = {{20, 40}, {16, 40}, {10, 20}, {8, 40}, {8, 20}, {4, 20}};
  const int B = batch_hint > 0 ? batch_hint : 1;
  const bool sep = nt == 1;  // separate LDS region for the intermediate (see the kernel's SEP)
  // CL pass first (the kernel's "concat from LDS" variant: fp16, one channel tile, at most one gathered K step, instantiated for
  // the 16x40 and 8x40 tiles): taken when a tile keeps two workgroups per CU.  OPT-IN, LITEPI_BNECK_CL=1 (2: the 16x40 tile
  // whatever its LDS size): it removes the re-reads of the concat buffer and is bit-equal, but the kernel is not traffic-bound --
  // the larger tile costs a workgroup per CU and 53 -> 67 us (v1) / 91 -> 96 us (v2, 8x40 tiles), DESIGN.md section 7
  const int cl_mode = getenv("LITEPI_BNECK_CL") ? atoi(getenv("LITEPI_BNECK_CL")) : 0;
  const bool try_cl = cl_out && psa_out && kg_cl > cg && kg_cl <= 4 && cl_mode > 0 && prec == LP_FP16 && nt == 1 && tail;
  for (int pass = try_cl ? 0 : 1; pass < 2; ++pass) {
  const bool clp = pass == 0;
  const int pssa = clp ? (kg_cl | 1) : pss;   // staged slots per pixel: odd = 16 consecutive pixels on 16 different 16-byte columns
  long best = -1;
  for (auto& cd : cand) {
    const int TH = cd[0], TW = cd[1];
    if (clp && !((TH == 16 || TH == 8) && TW == 40)) continue;
    if (clp && cl_mode == 2 && TH != 16) continue;
    if (clp && cl_mode == 3 && TH != 8) continue;    // (3: the 8x40 tile)
    static const bool no_big = getenv("LITEPI_BNECK_SMALL") != nullptr;  // A/B switch
    if (TH >= 16 && (nt != 1 || no_big)) continue;  // 12 + 10 (15 + 13) pixel tiles per wave: only the single-channel-tile variant has the registers
    if (TH == 10 && nt > 2) continue;               // 5 + 4 pixel tiles per wave: instantiated for one and two channel tiles
    // the balanced shapes pay only where they turn the grid into ONE full round (two workgroups on each of the 256 CUs of an
    // MI355X); on larger grids the smaller tiles' extra rounds overlap better (160x160: 50 us with 16x40, 57 us with 20x40)
    static const bool no_balanced = getenv("LITEPI_BNECK_NO_BALANCED") != nullptr;  // A/B switch
    if ((TH == 20 || TH == 10) && (no_balanced || (long)ceil_div(h, TH) * ceil_div(w, TW) * B > 512)) continue;
    if (tail && TH == 4) continue;                  // no cv2 instantiation for the smallest tile
    int l = TW + 4;
    if (cg <= 1) while (l % 16 != 2) ++l;  // one K group per pixel: 16 consecutive pixels x 4 taps conflict-free
    if (ceil_div(l * pssa, 64) > (clp ? 4 : 8)) continue;
    const size_t need = 512 + (size_t)2 * steps * nt * 1024 + extra_lds + (size_t)(TH + 4) * l * pssa * 16 + (size_t)(sep ? TH + 2 : 0) * l * pss * 16;
    if (need > 150 * 1024) continue;
    if (clp && cl_mode != 2 && need > 80 * 1024) continue;
    const long tiles = (long)ceil_div(h, TH) * ceil_div(w, TW);
    const int util = (int)(100.0 * h * w / ((double)tiles * TH * TW));
    // enough workgroups to fill the chip first, then two workgroups per CU, then no idle lanes, then the larger tile
    static const long bn_wgs = getenv("LITEPI_BNECK_WGS") ? atol(getenv("LITEPI_BNECK_WGS")) : 256;
    if (clp && tiles * B < bn_wgs) continue;   // small batches keep the small tiles of the gather plan (latency: workgroups first)
    const long score = (tiles * B >= bn_wgs ? 8 : tiles * B >= bn_wgs / 2 ? 4 : 0) * 1000L + (need <= 80 * 1024 ? 2000L : 0L) +
                       (util >= 95 ? 500L : util >= 80 ? 250L : 0L) + TH * TW / 10;
    if (score > best) { best = score; th = TH; tw = TW; lw = l; lds = need; }
  }
  if (best >= 0) {
    if (clp) { *cl_out = true; *psa_out = pssa * 16; }
    return true;
  }
  }
  return false;
}

bool BottleneckPair::supported(int prec, int impl, int c_phys, int h, int w, int batch_hint, const Cv2* cv2) {
  int th, tw, lw, t2 = 0, kg = 0, sg = 0;
  size_t lds, extra = 0;
  if (impl != IMPL_MFMA) return false;
  if (cv2) {
    if (!cv2_shape(prec, c_phys, *cv2, t2, kg, sg)) return false;
    const int nt = ceil_div(c_phys, 16);
    extra = (size_t)t2 * (sg + (prec == LP_FP16 ? (nt + 1) / 2 : nt)) * 1024;
  }
  return plan(prec, c_phys, h, w, batch_hint, extra, cv2 != nullptr, th, tw, lw, lds);
}

void BottleneckPair::build(int prec_, int c_phys, const std::vector<float>& wa, const std::vector<float>& ba,
                           const std::vector<float>& wb, const std::vector<float>& bb, int h, int w, int batch_hint, const Cv2* cv2) {
  prec = prec_; C = c_phys;
  const int G = prec == LP_FP16 ? 8 : 4;
  const int nt = ceil_div(C, 16);
  const int SR = prec == LP_FP16 ? (nt + 1) / 2 : nt;
  size_t extra = 0;
  if (cv2) {
    LP_CHECK(cv2_shape(prec, C, *cv2, T2, kg, sg), LP_ERR_GRAPH, "bottleneck %s: cv2 tail shape unsupported", name.c_str());
    extra = (size_t)T2 * (sg + SR) * 1024;
  }
  LP_CHECK(plan(prec, C, h, w, batch_hint, extra, cv2 != nullptr, TH, TW, LW, lds_bytes, (cv2 && cv2->in_is_last_stored) ? kg : 0, &cl, &PSA),
           LP_ERR_GRAPH, "bottleneck %d ch on %dx%d: no fused plan", C, h, w);
  a.name = name; b.name = name;
  a.build(prec, IMPL_MFMA, 3, 1, C, C, ACT_SILU, wa, ba, h, w, batch_hint, true, true);
  b.build(prec, IMPL_MFMA, 3, 1, C, C, ACT_SILU, wb, bb, h, w, batch_hint, true, true);
  NT = a.NT; CG = a.CGc; PS = a.PS; steps = a.steps;
  auto rcp16 = [&](int d, int range) {
    const unsigned m = (65536u + d - 1) / d;
    for (int x = 0; x < range; ++x) LP_CHECK((int)((x * m) >> 16) == x / d, LP_ERR_STATE, "reciprocal of %d not exact at %d", d, x);
    return m;
  };
  rcp_cg = rcp16(CG, 128);
  const int st_slots = (cl ? PSA : PS) / 16;   // slots per staged pixel
  rcp_ps = rcp16(st_slots, 512);
  rcp_pcs = rcp16(ceil_div(LW * st_slots, 64), 1024);
  LP_CHECK(!cl || steps <= 16, LP_ERR_STATE, "bottleneck CL: tap table too long");
  rcp_w1 = rcp16(TW + 2, 1024);
  rcp_tw = rcp16(TW, 1024);
  LP_CHECK((TH + 2) * (TW + 2) <= p1() * 4 * 16 && TH * TW <= p2() * 4 * 16, LP_ERR_STATE, "bottleneck tile exceeds the kernel's pixel-tile budget");
  if (cv2) {
    // cv2 fragments [T2][sg + SR][lane][G]: K steps 0..sg-1 walk the stored concat channels (group q = 4s+g), the
    // last SR steps the channels this lane's accumulators hold (same mapping as ConvLayer::attach_tail)
    C3 = cv2->c3; act3 = cv2->act;
    const int ccat = cv2->cat_global + C, ST = sg + SR;
    std::vector<uint8_t> buf((size_t)T2 * ST * 64 * 16, 0);
    for (int t = 0; t < T2; ++t)
      for (int s = 0; s < ST; ++s)
        for (int lane = 0; lane < 64; ++lane) {
          const int g = lane >> 4, m = lane & 15, gm = m >> 2, r = m & 3;
          const int oc = gm * 4 * T2 + t * 4 + r;
          if (oc >= C3) continue;
          for (int j = 0; j < G; ++j) {
            int ch;
            if (s < sg) {
              const int q = 4 * s + g;
              if (q >= kg) continue;
              ch = q * G + j;
            } else {
              const int s2 = s - sg;
              if (G * s2 + j >= 4 * NT) continue;
              const int mid = g * 4 * NT + G * s2 + j;
              if (mid >= C) continue;
              ch = cv2->cat_global + mid;
            }
            put_elem(buf, ((size_t)(t * ST + s) * 64 + lane) * G + j, prec, (*cv2->w)[(size_t)oc * ccat + ch]);
          }
        }
    d_w3.alloc(buf.size());
    LP_HIP(hipMemcpy(d_w3.p, buf.data(), buf.size(), hipMemcpyHostToDevice));
    std::vector<float> b3(round_up(C3, 64) + 64, 0.f);
    for (int c = 0; c < C3 && cv2->bias && c < (int)cv2->bias->size(); ++c) b3[c] = (*cv2->bias)[c];
    d_b3.alloc(b3.size() * 4);
    LP_HIP(hipMemcpy(d_b3.p, b3.data(), b3.size() * 4, hipMemcpyHostToDevice));
  }
}

template <typename T, int NT, int P1, int P2, int T2, int SG>
static void launch_bneck_sg(const BneckArgs& a, dim3 grid, size_t lds, hipStream_t st) {
  constexpr bool SEP = NT == 1;
  if constexpr (NT == 1 && T2 > 0 && SG == 1 && sizeof(T) == 2 && (P1 == 12 || P1 == 7)) {
    if (a.PSA > 0) {   // the CL variant (BottleneckPair::plan)
      set_max_dynamic_lds(reinterpret_cast<const void*>(bottleneck_mfma_kernel<T, NT, P1, P2, SEP, T2, SG, true>), 160 * 1024);
      LP_LAUNCH((bottleneck_mfma_kernel<T, NT, P1, P2, SEP, T2, SG, true>), grid, dim3(256), lds, st, a);
      return;
    }
  }
  LP_CHECK(a.PSA == 0, LP_ERR_STATE, "bottleneck: no CL kernel for this shape");
  set_max_dynamic_lds(reinterpret_cast<const void*>(bottleneck_mfma_kernel<T, NT, P1, P2, SEP, T2, SG>), 160 * 1024);
  LP_LAUNCH((bottleneck_mfma_kernel<T, NT, P1, P2, SEP, T2, SG>), grid, dim3(256), lds, st, a);
}
// fp16 cv2 tails: the gathered K steps (1..3) are a template parameter (straight-line epilogue, see the kernel)
template <typename T, int NT, int P1, int P2, int T2>
static void launch_bneck_(const BneckArgs& a, dim3 grid, size_t lds, hipStream_t st) {
  if constexpr (T2 > 0 && sizeof(T) == 2) {
    if (a.sg == 1) return launch_bneck_sg<T, NT, P1, P2, T2, 1>(a, grid, lds, st);
    if (a.sg == 2) return launch_bneck_sg<T, NT, P1, P2, T2, 2>(a, grid, lds, st);
    if (a.sg == 3) return launch_bneck_sg<T, NT, P1, P2, T2, 3>(a, grid, lds, st);
  }
  launch_bneck_sg<T, NT, P1, P2, T2, 0>(a, grid, lds, st);
}

// cv2 tails exist for NT = 1 with T2 in {1, 2} and NT = 2 with T2 in {2, 4} (BottleneckPair::cv2_shape)
template <typename T, int NT, int P1, int P2>
static void launch_bneck_t(const BneckArgs& a, int t2, dim3 grid, size_t lds, hipStream_t st) {
  if (t2 == 0) return launch_bneck_<T, NT, P1, P2, 0>(a, grid, lds, st);
  if constexpr (NT == 1 && P1 != 3) {
    if (t2 == 1) return launch_bneck_<T, NT, P1, P2, 1>(a, grid, lds, st);
    if (t2 == 2) return launch_bneck_<T, NT, P1, P2, 2>(a, grid, lds, st);
  }
  if constexpr (NT == 2 && P1 != 3) {
    if (t2 == 2) return launch_bneck_<T, NT, P1, P2, 2>(a, grid, lds, st);
    if (t2 == 4) return launch_bneck_<T, NT, P1, P2, 4>(a, grid, lds, st);
  }
  throw Error(LP_ERR_STATE, "bottleneck: no kernel for this cv2 tail");
}

// pixel tiles per wave = ceil(ceil(region / 16) / 4 waves) for the tile shapes of plan()
template <typename T, int NT>
static void launch_bneck(const BneckArgs& a, int t2, dim3 grid, size_t lds, hipStream_t st) {
  if (a.TH == 20 && a.TW == 40) {                                                        // 22x42 = 58 tiles, 20x40 = 50
    if constexpr (NT == 1) launch_bneck_t<T, 1, 15, 13>(a, t2, grid, lds, st);
    else throw Error(LP_ERR_STATE, "bottleneck: 20x40 tiles need NT == 1");
  } else if (a.TH == 16 && a.TW == 40) {                                                 // 18x42 = 48 tiles, 16x40 = 40
    if constexpr (NT == 1) launch_bneck_t<T, 1, 12, 10>(a, t2, grid, lds, st);
    else throw Error(LP_ERR_STATE, "bottleneck: 16x40 tiles need NT == 1");
  } else if (a.TH == 10 && a.TW == 20) {                                                 // 12x22 = 17 tiles, 10x20 = 13
    if constexpr (NT <= 2) launch_bneck_t<T, NT, 5, 4>(a, t2, grid, lds, st);
    else throw Error(LP_ERR_STATE, "bottleneck: 10x20 tiles need NT <= 2");
  } else if (a.TH == 8 && a.TW == 40) launch_bneck_t<T, NT, 7, 5>(a, t2, grid, lds, st);  // 10x42 = 27 tiles, 8x40 = 20
  else if (a.TH == 8 && a.TW == 20) launch_bneck_t<T, NT, 4, 3>(a, t2, grid, lds, st);    // 10x22 = 14 tiles, 8x20 = 10
  else if (a.TH == 4 && a.TW == 20) launch_bneck_t<T, NT, 3, 2>(a, t2, grid, lds, st);    //  6x22 =  9 tiles, 4x20 = 5
  else throw Error(LP_ERR_STATE, "bottleneck: no kernel for this tile shape");
}

void BottleneckPair::launch(const View& in, const View& out, int N, hipStream_t st, const View* cat) const {
  LP_CHECK(in.C == C && in.H == out.H && in.W == out.W, LP_ERR_STATE, "bottleneck: view mismatch");
  LP_CHECK((in.pitch % 8) == 0 && (out.pitch % 4) == 0, LP_ERR_STATE, "unaligned channel pitch");
  LP_CHECK(span_ok(N, in) && span_ok(N, out) && (!cat || span_ok(N, *cat)), LP_ERR_ARG, "bottleneck: a tensor of batch %d exceeds 2^31 elements", N);
  BneckArgs k;
  memset(&k, 0, sizeof(k));
  k.in = in.base; k.out = out.base; k.w1 = a.d_w.p; k.w2 = b.d_w.p; k.b1 = a.d_bias.as<float>(); k.b2 = b.d_bias.as<float>();
  k.zeros = a.d_bias.as<float>() + round_up(C, 64);
  k.N = N; k.H = in.H; k.W = in.W; k.C = C; k.in_pitch = in.pitch; k.out_pitch = out.pitch;
  k.TH = TH; k.TW = TW; k.tiles_x = ceil_div(in.W, TW); k.LW = LW; k.PS = PS; k.CG = CG; k.steps = steps;
  if (T2) {
    LP_CHECK(cat && cat->base && cat->H == in.H && cat->W == in.W && out.C >= C3, LP_ERR_STATE, "bottleneck+cv2: concat view missing or output too narrow");
    k.w3 = d_w3.p; k.b3 = d_b3.as<float>(); k.cat = cat->base; k.cat_pitch = cat->pitch;
    k.out3 = out.base; k.out3_pitch = out.pitch; k.C3 = C3; k.act3 = act3; k.kg = kg; k.sg = sg;
    k.out = nullptr;
    if (cl) {
      const size_t es = prec == LP_FP16 ? 2 : 4;
      LP_CHECK(reinterpret_cast<const char*>(in.base) == reinterpret_cast<const char*>(cat->base) + (size_t)(kg * (prec == LP_FP16 ? 8 : 4) - C) * es &&
                   in.pitch == cat->pitch,
               LP_ERR_STATE, "bottleneck %s: planned with the input as the concat's last stored segment, launched with another view", name.c_str());
      k.PSA = PSA;
    }
  } else {
    LP_CHECK(out.C >= C, LP_ERR_STATE, "bottleneck: output view too narrow");
  }
  const int tiles_y = ceil_div(in.H, TH);
  const unsigned m = (65536u + k.tiles_x - 1) / k.tiles_x;
  for (int x = 0; x < k.tiles_x * tiles_y; ++x) LP_CHECK((int)((x * m) >> 16) == x / k.tiles_x, LP_ERR_STATE, "tile reciprocal not exact");
  k.rcp_tx = m; k.rcp_cg = rcp_cg; k.rcp_ps = rcp_ps; k.rcp_pcs = rcp_pcs; k.rcp_w1 = rcp_w1; k.rcp_tw = rcp_tw;
  static const bool tile_major = getenv("LITEPI_TILE_MAJOR") != nullptr;
  k.tile_major = tile_major;
  dim3 grid(N, k.tiles_x * tiles_y);
  if (tile_major) grid = dim3(k.tiles_x * tiles_y, N);
  const bool f16 = prec == LP_FP16;
  static const char* stamp_path = getenv("LITEPI_BNECK_STAMPS");
  DevBuf d_stamps;
  if (stamp_path && *stamp_path) {
    d_stamps.alloc((size_t)grid.x * grid.y * 16 * 8);
    LP_HIP(hipMemsetAsync(d_stamps.p, 0, (size_t)grid.x * grid.y * 16 * 8, st));
    k.stamps = d_stamps.as<unsigned long long>();
  }
  switch (NT) {
    case 1: if (f16) launch_bneck<half_t, 1>(k, T2, grid, lds_bytes, st); else launch_bneck<float, 1>(k, T2, grid, lds_bytes, st); break;
    case 2: if (f16) launch_bneck<half_t, 2>(k, T2, grid, lds_bytes, st); else launch_bneck<float, 2>(k, T2, grid, lds_bytes, st); break;
    case 3: if (f16) launch_bneck<half_t, 3>(k, T2, grid, lds_bytes, st); else launch_bneck<float, 3>(k, T2, grid, lds_bytes, st); break;
    default: if (f16) launch_bneck<half_t, 4>(k, T2, grid, lds_bytes, st); else launch_bneck<float, 4>(k, T2, grid, lds_bytes, st); break;
  }
  LP_HIP(hipGetLastError());
  if (k.stamps) {  // diagnostic: dump [grid][16] stamps, one record per launch (tools/bneck_stamps.py)
    LP_HIP(hipStreamSynchronize(st));
    std::vector<unsigned long long> hs((size_t)grid.x * grid.y * 16);
    LP_HIP(hipMemcpy(hs.data(), d_stamps.p, hs.size() * 8, hipMemcpyDeviceToHost));
    if (FILE* f = fopen(stamp_path, "ab")) {
      const unsigned long long hdr[8] = {0x424e4543ull, hs.size() / 16, (unsigned long long)in.H, (unsigned long long)N, (unsigned long long)C,
                                         (unsigned long long)T2, (unsigned long long)TH, (unsigned long long)TW};
      fwrite(hdr, 8, 8, f);
      fwrite(hs.data(), 8, hs.size(), f);
      fclose(f);
    }
  }
}

void StemLayer::build(int prec_, int cout_phys, int act_, const std::vector<float>& w_bgr, const std::vector<float>& bias, int k_, int stride_, int pad_) {
  prec = prec_; CO = cout_phys; act = act_; k = k_; stride = stride_; pad = pad_;
  LP_CHECK((int)w_bgr.size() == k * k * 3 * CO, LP_ERR_GRAPH, "stem weights do not match a %dx%d kernel", k, k);
  LP_CHECK(CO == 8 || CO == 16 || CO == 32, LP_ERR_GRAPH, "stem with %d output channels unsupported", CO);
  d_w.alloc(w_bgr.size() * 4);
  LP_HIP(hipMemcpy(d_w.p, w_bgr.data(), w_bgr.size() * 4, hipMemcpyHostToDevice));
  std::vector<float> b(CO, 0.f);
  for (size_t i = 0; i < bias.size() && i < (size_t)CO; ++i) b[i] = bias[i];
  d_bias.alloc(CO * 4);
  LP_HIP(hipMemcpy(d_bias.p, b.data(), CO * 4, hipMemcpyHostToDevice));
  if (prec == LP_FP16 && CO == 8 && act == ACT_SILU && k == 3 && stride == 2 && pad == 1) {
    // stem_mfma_kernel A fragments [2 steps][64 lanes][8]: row m = (pixel parity, channel); K group q = 4s+g ->
    // (ky = q/2, byte half h = q%2) of the 15-byte union of the two windows; the even pixel uses union bytes 0..8,
    // the odd one 6..14.  1/255 folded in.
    std::vector<uint8_t> buf((size_t)2 * 64 * 16, 0);
    for (int s = 0; s < 2; ++s)
      for (int lane = 0; lane < 64; ++lane) {
        const int g = lane >> 4, m = lane & 15, par = m >> 3, ch = m & 7;
        const int q = 4 * s + g;
        if (q >= 6) continue;
        const int ky = q >> 1, h = q & 1;
        for (int j = 0; j < 8; ++j) {
          const int jj = 8 * h + j, jw = jj - 6 * par;
          if (jw < 0 || jw > 8) continue;
          put_elem(buf, ((size_t)s * 64 + lane) * 8 + j, LP_FP16, w_bgr[(size_t)(ky * 9 + jw) * CO + ch] / 255.f);
        }
      }
    d_afrag.alloc(buf.size());
    LP_HIP(hipMemcpy(d_afrag.p, buf.data(), buf.size(), hipMemcpyHostToDevice));
    // stem_block_kernel's order of the K groups: step 0 = (ky 0, h 0) (ky 2, h 0) (ky 0, h 1) (ky 2, h 1), step 1 = (ky 1, h 0)
    // zero (ky 1, h 1) zero -- the two groups of a 32-lane LDS read group are two tile rows = 16 banks apart
    std::vector<uint8_t> blk((size_t)2 * 64 * 16, 0);
    for (int s = 0; s < 2; ++s)
      for (int lane = 0; lane < 64; ++lane) {
        const int g = lane >> 4, m = lane & 15, par = m >> 3, ch = m & 7;
        if (s == 1 && (g & 1)) continue;
        const int ky = s == 0 ? 2 * (g & 1) : 1, h = g >> 1;
        for (int j = 0; j < 8; ++j) {
          const int jj = 8 * h + j, jw = jj - 6 * par;
          if (jw < 0 || jw > 8) continue;
          put_elem(blk, ((size_t)s * 64 + lane) * 8 + j, LP_FP16, w_bgr[(size_t)(ky * 9 + jw) * CO + ch] / 255.f);
        }
      }
    d_afrag_blk.alloc(blk.size());
    LP_HIP(hipMemcpy(d_afrag_blk.p, blk.data(), blk.size(), hipMemcpyHostToDevice));
  }
  if (prec == LP_FP16 && CO == 16 && act == ACT_SILU && k == 3 && stride == 2 && pad == 1) {
    // stem_mfma16_kernel A fragment [64 lanes][8]: row m = channel; K group g < 3 = window row g bytes 0..7, group 3 =
    // byte 8 of rows 0..2.  1/255 folded in.
    std::vector<uint8_t> buf((size_t)64 * 16, 0);
    for (int lane = 0; lane < 64; ++lane) {
      const int g = lane >> 4, ch = lane & 15;
      for (int j = 0; j < 8; ++j) {
        int row;
        if (g < 3) row = g * 9 + j;
        else if (j < 3) row = j * 9 + 8;
        else continue;
        put_elem(buf, (size_t)lane * 8 + j, LP_FP16, w_bgr[(size_t)row * CO + ch] / 255.f);
      }
    }
    d_afrag.alloc(buf.size());
    LP_HIP(hipMemcpy(d_afrag.p, buf.data(), buf.size(), hipMemcpyHostToDevice));
  }
}


bool StemLayer::block_supported(const ConvLayer& c1) const {
  const bool common = prec == LP_FP16 && c1.prec == LP_FP16 && c1.impl == IMPL_MFMA && c1.direct && c1.k == 3 && c1.stride == 2 && c1.nsplits == 1 &&
                      c1.act == ACT_SILU && c1.Cin == CO;
  if (!common) return false;
  if (CO == 8) return d_afrag_blk.p && c1.NT == 1 && c1.T2 == 1 && c1.steps == 3;                                     // v1: stem_block_kernel
  if (CO == 16) return d_afrag.p && c1.NT == 2 && c1.T2 == 2 && c1.steps == 5 && c1.Cout <= 32 && c1.Cout2 <= 32;     // v2: stem_block16_kernel
  return false;
}

void StemLayer::launch_block(const uint8_t* img, int N, int Hin, int Win, const ConvLayer& c1, const View& out, hipStream_t st) const {
  LP_CHECK(block_supported(c1) && Win % 4 == 0 && (reinterpret_cast<uintptr_t>(img) & 3) == 0, LP_ERR_STATE, "stem block: unsupported configuration");
  StemBlockArgs a;
  memset(&a, 0, sizeof(a));
  a.img = img; a.afrag = CO == 16 ? d_afrag.p : d_afrag_blk.p; a.sbias = d_bias.as<float>();
  a.w1 = c1.d_w.p; a.b1 = c1.d_bias.as<float>(); a.w2 = c1.d_w2.p; a.b2 = c1.d_bias2.as<float>();
  a.out = out.base; a.N = N; a.Hin = Hin; a.Win = Win; a.H1 = (Hin + 1) / 2; a.W1 = (Win + 1) / 2;
  a.H2 = out.H; a.W2 = out.W; a.out_pitch = out.pitch; a.C2 = c1.Cout2; a.act2 = c1.act2;
  LP_CHECK(a.H2 == (a.H1 + 1) / 2 && a.W2 == (a.W1 + 1) / 2 && out.C >= c1.Cout2, LP_ERR_STATE, "stem block: output view mismatch");
  dim3 grid(N, ceil_div(a.W2, SB_TW), ceil_div(a.H2, SB_TH));
  static const int th16 = getenv("LITEPI_SB16_TH") ? atoi(getenv("LITEPI_SB16_TH")) : 8;   // A/B: output rows per workgroup of the v2 kernel
  if (CO == 16 && th16 == 4) {
    dim3 grid4(N, ceil_div(a.W2, 32), ceil_div(a.H2, 4));
    LP_LAUNCH(stem_block16_kernel<4>, grid4, dim3(256), 0, st, a);
  } else if (CO == 16) {
    LP_LAUNCH(stem_block16_kernel<8>, grid, dim3(256), 0, st, a);
  } else {
    LP_LAUNCH(stem_block_kernel, grid, dim3(256), 0, st, a);
  }
  LP_HIP(hipGetLastError());
}

void StemLayer::launch(const uint8_t* img, int N, int Hin, int Win, const View& out, hipStream_t st) const {
  const bool aligned = k == 3 && stride == 2 && pad == 1 && Win % 4 == 0 && (reinterpret_cast<uintptr_t>(img) & 3) == 0 &&
                       out.H == (Hin + 1) / 2 && out.W == (Win + 1) / 2;
  static const bool no_mfma_stem = getenv("LITEPI_NO_MFMA_STEM") != nullptr;
  if (aligned && d_afrag.p && !no_mfma_stem && CO == 16) {
    dim3 g3(N, ceil_div(out.W, STEMM_TW), ceil_div(out.H, STEMM_TH));
    LP_LAUNCH(stem_mfma16_kernel, g3, dim3(256), 0, st, img, reinterpret_cast<half_t*>(out.base),
                       reinterpret_cast<const u32x4*>(d_afrag.p), d_bias.as<float>(), N, Hin, Win, out.H, out.W, out.pitch);
    LP_HIP(hipGetLastError());
    return;
  }
  if (aligned && d_afrag.p && !no_mfma_stem) {
    dim3 g3(N, ceil_div(out.W, STEMM_TW), ceil_div(out.H, STEMM_TH));
    LP_LAUNCH(stem_mfma_kernel, g3, dim3(256), 0, st, img, reinterpret_cast<half_t*>(out.base),
                       reinterpret_cast<const u32x4*>(d_afrag.p), d_bias.as<float>(), N, Hin, Win, out.H, out.W, out.pitch);
    LP_HIP(hipGetLastError());
    return;
  }
  if (aligned) {
    dim3 g2(ceil_div(out.W, STEM_TW), ceil_div(out.H, STEM_TH), N);
#define LP_STL(TT, C)                                                                                            \
  LP_LAUNCH((stem_conv_lds_kernel<TT, C>), g2, dim3(256), 0, st, img, reinterpret_cast<TT*>(out.base), \
                     d_w.as<float>(), d_bias.as<float>(), N, Hin, Win, out.H, out.W, out.pitch, act)
    if (prec == LP_FP16) {
      if (CO == 8) LP_STL(half_t, 8); else if (CO == 16) LP_STL(half_t, 16); else LP_STL(half_t, 32);
    } else {
      if (CO == 8) LP_STL(float, 8); else if (CO == 16) LP_STL(float, 16); else LP_STL(float, 32);
    }
#undef LP_STL
    LP_HIP(hipGetLastError());
    return;
  }
  const long total = (long)N * out.H * out.W;
  dim3 grid((unsigned)((total + 255) / 256));
#define LP_ST(TT, C)                                                                                     \
  LP_LAUNCH((stem_conv_kernel<TT, C>), grid, dim3(256), 0, st, img, reinterpret_cast<TT*>(out.base), \
                     d_w.as<float>(), d_bias.as<float>(), N, Hin, Win, out.H, out.W, out.pitch, act, k, stride, pad)
  if (prec == LP_FP16) {
    if (CO == 8) LP_ST(half_t, 8); else if (CO == 16) LP_ST(half_t, 16); else LP_ST(half_t, 32);
  } else {
    if (CO == 8) LP_ST(float, 8); else if (CO == 16) LP_ST(float, 16); else LP_ST(float, 32);
  }
#undef LP_ST
  LP_HIP(hipGetLastError());
}

}  // namespace lp
