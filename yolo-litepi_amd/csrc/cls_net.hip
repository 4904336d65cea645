// ShuffleNetV2 x1.0 in two launches (fp16 storage, fp32 accumulate, gfx950).  Replaces self.model(batch) +
// softmax / arg-max of PyTorchClassifier.predict_batch (reference e2e.py:393-396; network: torchvision
// shufflenet_v2_x1_0) and the ToTensor / Normalize of e2e.py:368-369.
//
// Why two kernels and not one per layer: a ROI's activations are tiny (12 KB .. 2 KB), so the layer-at-a-time plan
// was ~25 launches of a few microseconds of work each, all of them waiting on launch latency and on HBM round
// trips of tensors that fit in LDS.  Here a workgroup keeps its ROI(s) in LDS from the uint8 crop to the class
// probabilities; the only traffic is the weight stream (2.6 MB per workgroup pass, served by L2) and 8 KB per ROI
// between the two kernels.
//   cls_front: 1 ROI per workgroup (256 .. 64 pixels per map: enough MFMA columns on its own)
//   cls_back : 4 ROIs per workgroup (16 / 4 pixels per ROI: four ROIs fill one 16-column MFMA tile at stage 4)
// MFMA orientation as everywhere in this library: D[out-channel][pixel] = W . X, A = weight fragments
// (pack_fused_pw: [tile][K step][lane][8 halfs], read from L2 straight into registers: every fragment is used by
// exactly one wave, so LDS staging would only add a copy), B = 16 pixels x 8 channels per lane = one ds_read_b128
// of the NHWC LDS image.  Rows of every LDS image are padded by 16 B (odd number of 16-byte slots per row).
#include "cls_net.h"

namespace lp {

typedef _Float16 half_t;
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

#define CN_THREADS 512
#define CN_WAVES 8

// ShuffleNetV2 x1.0 geometry (checked on the host against the state_dict)
#define BF2 58
#define BFP2 64
#define BF3 116
#define BFP3 128
#define BF4 232
#define BFP4 240

// ---- small device helpers -----------------------------------------------------------------------------------------
template <int SMAX>
__device__ __forceinline__ void wload(u32x4 (&af)[SMAX], const u32x4_t* __restrict__ w, int t, int S, int lane) {
#pragma unroll
  for (int s = 0; s < SMAX; ++s) {
    af[s] = u32x4{0u, 0u, 0u, 0u};
    if (s < S) af[s] = w[((size_t)t * S + s) * 64 + lane];
  }
}

// acc[p] (+)= W(tile) . X[pixels p0 + 16p .. +15]: K steps s0 .. s0+SMAX-1 of the fragments in af; Kp = physical
// channels of the LDS rows (K groups past it are row padding / the next row: never read)
template <int SMAX, int PT>
__device__ __forceinline__ void gemm_acc(const u32x4 (&af)[SMAX], int S, int s0, const char* bsrc, int brow, int Kp, int p0, int lane,
                                         floatx4 (&acc)[PT]) {
  const int g = lane >> 4, col = lane & 15;
#pragma unroll
  for (int s = 0; s < SMAX; ++s) {
    if (s < S) {
      const half8 a = __builtin_bit_cast(half8, af[s]);
      const int kg = 4 * (s0 + s) + g;
#pragma unroll
      for (int p = 0; p < PT; ++p) {
        u32x4 v = u32x4{0u, 0u, 0u, 0u};
        if (kg * 8 < Kp) v = *reinterpret_cast<const u32x4*>(bsrc + (p0 + p * 16 + col) * brow + kg * 16);
        acc[p] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, __builtin_bit_cast(half8, v), acc[p], 0, 0, 0);
      }
    }
  }
}

template <int PT> __device__ __forceinline__ void zero_acc(floatx4 (&acc)[PT]) {
#pragma unroll
  for (int p = 0; p < PT; ++p) acc[p] = floatx4{0.f, 0.f, 0.f, 0.f};
}

// relu(acc + bias) of tile t -> fp16 LDS image dst[pixel][channel]
template <int PT>
__device__ __forceinline__ void store_relu(char* dst, int drow, int p0, int t, const float* __restrict__ bias, const floatx4 (&acc)[PT],
                                           int lane) {
  const int g = lane >> 4, col = lane & 15;
  const int ch0 = t * 16 + 4 * g;
  const floatx4 b = *reinterpret_cast<const floatx4*>(bias + ch0);
#pragma unroll
  for (int p = 0; p < PT; ++p) {
    half4 q;
#pragma unroll
    for (int i = 0; i < 4; ++i) q[i] = (half_t)fmaxf(acc[p][i] + b[i], 0.f);
    *reinterpret_cast<half4*>(dst + (p0 + p * 16 + col) * drow + ch0 * 2) = q;
  }
}

// depthwise 3x3 (pad 1, stride STRIDE) + bias over C physical channels of nroi stacked WIN x WIN maps (fp32 accumulate,
// weights fp32 [9][C] straight from L1/L2: 288 B per item, shared by every workgroup)
template <int C, int WIN, int WOUT, int STRIDE>
__device__ __forceinline__ void dwconv(const char* in, int irow, char* out, int orow, const float* __restrict__ w,
                                       const float* __restrict__ b, int nroi, int tid) {
  constexpr int CG = C / 8;
  const int items = nroi * WOUT * WOUT * CG;
  for (int i = tid; i < items; i += CN_THREADS) {
    const int cg = i % CG, px = i / CG;
    const int r = px / (WOUT * WOUT), pp = px - r * (WOUT * WOUT);
    const int oy = pp / WOUT, ox = pp - oy * WOUT;
    float acc[8];
    {
      const floatx4 b0 = *reinterpret_cast<const floatx4*>(b + cg * 8), b1 = *reinterpret_cast<const floatx4*>(b + cg * 8 + 4);
#pragma unroll
      for (int j = 0; j < 4; ++j) { acc[j] = b0[j]; acc[4 + j] = b1[j]; }
    }
    // one window row (3 taps: 36 registers of operands) in flight at a time: the pointwise weight fragments of the
    // surrounding GEMMs stay live across this phase, and nine taps of operands on top of them spill
#pragma unroll 1
    for (int ky = 0; ky < 3; ++ky) {
      const int iy = oy * STRIDE - 1 + ky;
      if (iy < 0 || iy >= WIN) continue;
      const char* rowp = in + ((r * WIN + iy) * WIN) * irow + cg * 16;
      const float* wrow = w + (ky * 3) * C + cg * 8;
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int ix = ox * STRIDE - 1 + kx;
        const bool ok = ix >= 0 && ix < WIN;
        u32x4 raw = u32x4{0u, 0u, 0u, 0u};
        if (ok) raw = *reinterpret_cast<const u32x4*>(rowp + ix * irow);
        const half8 v = __builtin_bit_cast(half8, raw);
        const floatx4 w0 = *reinterpret_cast<const floatx4*>(wrow + kx * C);
        const floatx4 w1 = *reinterpret_cast<const floatx4*>(wrow + kx * C + 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          acc[j] = fmaf((float)v[j], w0[j], acc[j]);
          acc[4 + j] = fmaf((float)v[4 + j], w1[j], acc[4 + j]);
        }
      }
    }
    half8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (half_t)acc[j];
    *reinterpret_cast<half8*>(out + px * orow + cg * 16) = o;
  }
}

// channel_shuffle(cat(a, b), 2): logical channel 2c = a[c], 2c+1 = b[c]; each half of the stage tensor is padded to bfp
__device__ __forceinline__ int shuffle_phys(int l, int bf, int bfp) { return l < bf ? l : bfp + (l - bf); }

// pointwise biases of a stage's stride-1 blocks -> LDS [NB][2][BFP] (b1 | b2), once per stage: read from global where they
// are used (right behind a GEMM) each was an exposed L2 round trip, two per block.  The caller's next barrier publishes them.
template <int BFP, int NB>
__device__ __forceinline__ void stage_biases(float* BI, const FusedBlockW* blk, int tid) {
  constexpr int Q = BFP / 4;   // float4 per vector
  for (int i = tid; i < NB * 2 * Q; i += CN_THREADS) {
    const int b = i / (2 * Q), rem = i - b * 2 * Q, which = rem / Q, j = rem - which * Q;
    const float* src = which ? blk[b].b2 : blk[b].b1;
    *reinterpret_cast<floatx4*>(BI + (size_t)(b * 2 + which) * BFP + 4 * j) = *reinterpret_cast<const floatx4*>(src + 4 * j);
  }
}

// depthwise parameters of a stride-2 block ([9][C] taps + [C] bias, fp32) -> LDS [10][C], once per workgroup
template <int C>
__device__ __forceinline__ void stage_dw(float* dst, const float* __restrict__ w, const float* __restrict__ b, int tid) {
  constexpr int Q = 10 * C / 4;
  for (int i = tid; i < Q; i += CN_THREADS)
    *reinterpret_cast<floatx4*>(dst + 4 * i) = i < 9 * C / 4 ? *reinterpret_cast<const floatx4*>(w + 4 * i) : *reinterpret_cast<const floatx4*>(b + 4 * (i - 9 * C / 4));
}

// stride-1 InvertedResidual on X[npx pixels][2*bfp] resident in LDS (in place): x_lo passes through, x_hi -> pw1+ReLU ->
// dw3x3 -> pw2+ReLU, then channel_shuffle.  TT tiles of 16 channels, S K steps, the wave handles tile rounds
// t = wave, wave + 8, .. with PT pixel tiles from p0.  f1 holds this block's pw1 fragments for the wave's FIRST round on
// entry (loaded by the caller / the previous block) and the next block's on exit (w1n; nullptr: none).
template <int BF, int BFP, int W, int PT, int ROUNDS, bool PREF = true>
__device__ __forceinline__ void s1_block(char* X, int xrow, char* T1, char* T2, int trow, const FusedBlockW& bw, const u32x4_t* w1n, int nroi,
                                         int p0, int tile0, int tstride, u32x4 (&f1)[ROUNDS][BFP / 32 + ((BFP % 32) ? 1 : 0)], int tid, float* DW, const float* BI) {
  constexpr int TT = BFP / 16, S = BFP / 32 + ((BFP % 32) ? 1 : 0);
  const int lane = tid & 63;
  const int g = lane >> 4, col = lane & 15;
  // The block's depthwise parameters ([9][BFP] taps + [BFP] bias, fp32) go global -> registers NOW and -> LDS (DW) before the
  // first barrier: the depthwise phase then reads them from LDS.  (Read from global inside the depthwise loop they were four
  // dependent L2 round trips per item -- bias, then one per window row -- with one workgroup per CU and nothing to hide them:
  // most of a block's ~6 us.)
  constexpr int NDW4 = 10 * BFP / 4, DWPT = (NDW4 + CN_THREADS - 1) / CN_THREADS;
  floatx4 dwreg[DWPT];
#pragma unroll
  for (int k = 0; k < DWPT; ++k) {
    int idx = tid + k * CN_THREADS;
    idx = idx < NDW4 ? idx : NDW4 - 1;
    dwreg[k] = idx < 9 * BFP / 4 ? *reinterpret_cast<const floatx4*>(bw.dw + 4 * idx) : *reinterpret_cast<const floatx4*>(bw.dwb + 4 * (idx - 9 * BFP / 4));
  }
  u32x4 f2[ROUNDS][S];
  floatx4 acc[ROUNDS][PT];
  // A wave whose round has no tile left (15 tiles over 8 waves) recomputes the last tile and skips the stores: loads and
  // MFMAs stay unconditional -- with the loads inside `if (tile < TT)` blocks the register allocator spilled every fragment
  // of the block straight after loading it, one L2 round trip at a time.
  int tc[ROUNDS];
  bool live[ROUNDS];
#pragma unroll
  for (int r = 0; r < ROUNDS; ++r) {
    const int t = tile0 + r * tstride;
    live[r] = t < TT;
    tc[r] = live[r] ? t : TT - 1;
  }
  // ---- t1 = relu(W1 . x_hi + b1)
  // PREF (stages 2, 3: 2-4 fragments per set): pw2's fragments are requested before pw1 runs and the next block's pw1
  // fragments before the depthwise phase, so no GEMM waits for L2.  Stage 4 (16 fragments = 64 registers per set, two
  // rounds) holds ONE set at a time: its pw1 fragments are loaded on entry (one exposed L2 round trip per block), pw2's
  // during the depthwise phase.
  if (PREF) {
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) wload<S>(f2[r], bw.w2, tc[r], S, lane);  // lands while pw1 + dw run
  } else {
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) wload<S>(f1[r], bw.w1, tc[r], S, lane);
  }
#pragma unroll
  for (int r = 0; r < ROUNDS; ++r) {
    zero_acc<PT>(acc[r]);
    gemm_acc<S, PT>(f1[r], S, 0, X + BFP * 2, xrow, BFP, p0, lane, acc[r]);
    if (live[r]) store_relu<PT>(T1, trow, p0, tc[r], BI, acc[r], lane);   // pw1's bias, staged in LDS by stage_biases()
  }
  asm volatile("" ::: "memory");
  if (PREF) {
    if (w1n) {
#pragma unroll
      for (int r = 0; r < ROUNDS; ++r) wload<S>(f1[r], w1n, tc[r], S, lane);
    }
  } else {
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) wload<S>(f2[r], bw.w2, tc[r], S, lane);
  }
#pragma unroll
  for (int k = 0; k < DWPT; ++k) {
    const int idx = tid + k * CN_THREADS;
    if (idx < NDW4) *reinterpret_cast<floatx4*>(DW + 4 * idx) = dwreg[k];
  }
  __syncthreads();
  // ---- t2 = dw3x3(t1) + bd
  dwconv<BFP, W, W, 1>(T1, trow, T2, trow, DW, DW + 9 * BFP, nroi, tid);
  __syncthreads();
  // ---- y = relu(W2 . t2 + b2); X = shuffle(cat(x_lo, y)) in place
  half4 x1v[ROUNDS][PT];
#pragma unroll
  for (int r = 0; r < ROUNDS; ++r) {
    zero_acc<PT>(acc[r]);
    gemm_acc<S, PT>(f2[r], S, 0, T2, trow, BFP, p0, lane, acc[r]);
#pragma unroll
    for (int p = 0; p < PT; ++p) x1v[r][p] = *reinterpret_cast<const half4*>(X + (p0 + p * 16 + col) * xrow + (tc[r] * 16 + 4 * g) * 2);
  }
  __syncthreads();  // every x_lo value is in registers before any interleaved pair is written
#pragma unroll
  for (int r = 0; r < ROUNDS; ++r) {
    if (live[r]) {
      const int ch0 = tc[r] * 16 + 4 * g;
      const floatx4 bias = *reinterpret_cast<const floatx4*>(BI + BFP + ch0);
#pragma unroll
      for (int p = 0; p < PT; ++p) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int c = ch0 + i;
          if (c < BF) {
            half2v pr;
            pr[0] = x1v[r][p][i];
            pr[1] = (half_t)fmaxf(acc[r][p][i] + bias[i], 0.f);
            *reinterpret_cast<half2v*>(X + (p0 + p * 16 + col) * xrow + shuffle_phys(2 * c, BF, BFP) * 2) = pr;
          }
        }
      }
    }
  }
  __syncthreads();
}

// =====================================================================================================================
// cls_front: conv1+BN+ReLU (MFMA, straight from the uint8 crop) -> maxpool -> stage2.0..3 -> stage3.0, one ROI per
// workgroup pass.
// LDS (bytes):  IN [64 rows][192] uint8 (+16 guard each side)      12352   (later: T2 of the stage-2 blocks)
//               RA: STEM [32x32][24 ch] fp16 = 49152                49152   (later: T1 / D1 / stage3.0 buffers)
//               POOL [16x16][24 ch, 64 B rows]                      16384
//               X2 [64 px][2 x 64 ch + 16 B]                        17408
// =====================================================================================================================
#define CF_IN 16
#define CF_RA 12352
#define CF_POOL (CF_RA + 49152)
#define CF_X2 (CF_POOL + 16384)
#define CF_LDS (CF_X2 + 17408)
#define CF_DW (10 * 64 * 4)      /* depthwise parameters of the running stride-1 block, fp32 [10][64], behind everything else */
#define CF_BI (3 * 2 * 64 * 4)   /* pointwise biases of the stage's stride-1 blocks, fp32 [3][2][64] */
#define CF_S2 ((10 * 24 + 10 * 64 + 10 * 128 + 10 * 128) * 4)   /* depthwise parameters of stage2.0 (24 | 64 ch) and stage3.0 (128 | 128) */
#define CF_POOLROW 64
#define CF_T1ROW 144   /* 64 ch x 2 B + 16 */
#define CF_X2ROW 272   /* 128 ch x 2 B + 16 */
#define CF_X3ROW 528   /* 256 ch x 2 B + 16 */

__global__ __launch_bounds__(CN_THREADS) void cls_front_kernel(const ClsFrontArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid0 = threadIdx.x, lane0 = tid0 & 63;
  const int R = *a.m_dyn;
  if ((int)blockIdx.x >= R) return;
  char* IN = smem + CF_IN;
  char* RA = smem + CF_RA;
  char* POOL = smem + CF_POOL;
  char* X2 = smem + CF_X2;
  char* T2 = smem;  // [64 px][144]: the IN region is dead once the stem has run
  // X2's padding channels are never written by the shuffle stores and must read as zero
  for (int i = tid0; i < 64 * CF_X2ROW / 16; i += CN_THREADS) *reinterpret_cast<u32x4*>(X2 + i * 16) = u32x4{0u, 0u, 0u, 0u};
  if (tid0 < 2) *reinterpret_cast<u32x4*>(smem + (tid0 ? CF_IN + 12288 : 0)) = u32x4{0u, 0u, 0u, 0u};  // guards
  stage_biases<BFP2, 3>(reinterpret_cast<float*>(smem + CF_LDS + CF_DW), a.s2, tid0);   // (published by the first barrier below)
  float* S2P = reinterpret_cast<float*>(smem + CF_LDS + CF_DW + CF_BI);   // stride-2 blocks' depthwise parameters, as above
  stage_dw<24>(S2P, a.s20.dw1, a.s20.dw1b, tid0);
  stage_dw<64>(S2P + 240, a.s20.dw2, a.s20.dw2b, tid0);
  stage_dw<128>(S2P + 240 + 640, a.s30.dw1, a.s30.dw1b, tid0);
  stage_dw<128>(S2P + 240 + 640 + 1280, a.s30.dw2, a.s30.dw2b, tid0);
  const half8 sa0 = __builtin_bit_cast(half8, a.stem_w[lane0]), sa1 = __builtin_bit_cast(half8, a.stem_w[64 + lane0]);

  for (int r = blockIdx.x; r < R; r += gridDim.x) {
    int tid = tid0, lane = lane0;  // see cls_back_kernel: keeps per-lane addresses from being hoisted out of the ROI loop
    asm volatile("" : "+v"(tid), "+v"(lane));
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, col = lane & 15;
    // ---- crop -> LDS (12288 B, 16 B per lane)
    {
      const u32x4* src = reinterpret_cast<const u32x4*>(a.rgb + (size_t)r * 12288);
      for (int i = tid; i < 768; i += CN_THREADS) *reinterpret_cast<u32x4*>(IN + i * 16) = src[i];
    }
    // first GEMM's weights fly during the stem
    u32x4 fpw1[1];
    wload<1>(fpw1, a.s20.pw1, wave & 3, 1, lane);
    __syncthreads();

    // ---- conv1 3x3/s2 (3 -> 24) + BN + ReLU on t = (x/255 - 0.18) / 0.34 (e2e.py:368-369), zero padding in t.
    //      K = 32: lanes g = 0..2 hold bytes 0..7 of window row ky = g (9 bytes per row: kx x RGB), lane group 3 the 9th
    //      byte of the three rows.  Bytes become fp16 exactly (0x6400 | b = 1024 + b, minus 1024); 1/255/0.34 is folded into
    //      the weights and -0.18/0.34 x (sum of the weights of the taps INSIDE the image) into four bias cases.
    for (int pt = wave * 8; pt < wave * 8 + 8; ++pt) {
      const int oy = pt >> 1, ox = (pt & 1) * 16 + col;
      const int b0 = 6 * ox - 3;
      uint32_t w0 = 0u, w1 = 0u;
      if (g < 3) {
        const int iy = 2 * oy - 1 + g;
        const int sh = b0 & 3, al = b0 - sh;
        const char* rowp = IN + (iy < 0 ? 0 : iy) * 192 + al;
        const uint32_t d0 = *reinterpret_cast<const uint32_t*>(rowp), d1 = *reinterpret_cast<const uint32_t*>(rowp + 4),
                       d2 = *reinterpret_cast<const uint32_t*>(rowp + 8);
        w0 = __builtin_amdgcn_alignbyte(d1, d0, sh);
        w1 = __builtin_amdgcn_alignbyte(d2, d1, sh);
        if (ox == 0) w0 &= 0xFF000000u;  // kx = 0 is left of the image
        if (iy < 0) { w0 = 0u; w1 = 0u; }
      } else {
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
          const int iy = 2 * oy - 1 + ky;
          const uint32_t v = iy < 0 ? 0u : (uint32_t) * reinterpret_cast<const uint8_t*>(IN + iy * 192 + b0 + 8);
          w0 |= v << (8 * ky);
        }
      }
      typedef uint32_t u32x4l __attribute__((ext_vector_type(4)));
      u32x4l hb;
      hb[0] = __builtin_amdgcn_perm(0x64646464u, w0, 0x04010400u);
      hb[1] = __builtin_amdgcn_perm(0x64646464u, w0, 0x04030402u);
      hb[2] = __builtin_amdgcn_perm(0x64646464u, w1, 0x04010400u);
      hb[3] = __builtin_amdgcn_perm(0x64646464u, w1, 0x04030402u);
      half8 bf = __builtin_bit_cast(half8, hb);
#pragma unroll
      for (int j = 0; j < 8; ++j) bf[j] = bf[j] - (half_t)1024.f;
      floatx4 c0 = floatx4{0.f, 0.f, 0.f, 0.f}, c1 = floatx4{0.f, 0.f, 0.f, 0.f};
      c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(sa0, bf, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(sa1, bf, c1, 0, 0, 0);
      const int ci = (oy == 0 ? 1 : 0) | (ox == 0 ? 2 : 0);
      const floatx4 bb0 = *reinterpret_cast<const floatx4*>(a.stem_b + ci * 32 + 4 * g);
      char* o = RA + (oy * 32 + ox) * 48;
      half4 q;
#pragma unroll
      for (int i = 0; i < 4; ++i) q[i] = (half_t)fmaxf(c0[i] + bb0[i], 0.f);
      *reinterpret_cast<half4*>(o + 8 * g) = q;
      if (g < 2) {
        const floatx4 bb1 = *reinterpret_cast<const floatx4*>(a.stem_b + ci * 32 + 16 + 4 * g);
#pragma unroll
        for (int i = 0; i < 4; ++i) q[i] = (half_t)fmaxf(c1[i] + bb1[i], 0.f);
        *reinterpret_cast<half4*>(o + 32 + 8 * g) = q;
      }
    }
    __syncthreads();
    // ---- maxpool 3x3/s2/p1: [32x32x24] -> POOL [16x16][24]
    for (int i = tid; i < 768; i += CN_THREADS) {
      const int cg = i % 3, px = i / 3;
      const int oy = px >> 4, ox = px & 15;
      half8 m;
#pragma unroll
      for (int j = 0; j < 8; ++j) m[j] = (half_t)-65504.f;
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) {
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          const int iy = 2 * oy - 1 + ky, ix = 2 * ox - 1 + kx;
          if (iy >= 0 && ix >= 0) {  // iy, ix <= 31 always
            const half8 v = *reinterpret_cast<const half8*>(RA + (iy * 32 + ix) * 48 + cg * 16);
#pragma unroll
            for (int j = 0; j < 8; ++j) m[j] = v[j] > m[j] ? v[j] : m[j];
          }
        }
      }
      *reinterpret_cast<half8*>(POOL + px * CF_POOLROW + cg * 16) = m;
    }
    __syncthreads();

    // ================= stage2.0 (stride 2): 16x16x24 -> 8x8x116 =================
    char* T1 = RA;               // [256 px][144]
    char* D1 = RA + 256 * CF_T1ROW;  // [64 px][64 B rows] branch1 depthwise output (24 ch)
    u32x4 fy[2];                 // final pointwise of this wave's branch: branch1.2 (1 step) or branch2.5 (2 steps)
    {
      const int t = wave & 3, p0 = (wave >> 2) * 128;
      floatx4 acc[8];
      zero_acc<8>(acc);
      gemm_acc<1, 8>(fpw1, 1, 0, POOL, CF_POOLROW, 24, p0, lane, acc);
      store_relu<8>(T1, CF_T1ROW, p0, t, a.s20.pw1b, acc, lane);
      if (wave < 4) wload<2>(fy, a.s20.pwb1, t, 1, lane); else wload<2>(fy, a.s20.pw2, t, 2, lane);
      dwconv<24, 16, 8, 2>(POOL, CF_POOLROW, D1, 64, S2P, S2P + 9 * 24, 1, tid);
    }
    __syncthreads();
    dwconv<64, 16, 8, 2>(T1, CF_T1ROW, T2, CF_T1ROW, S2P + 240, S2P + 240 + 9 * 64, 1, tid);
    __syncthreads();
    u32x4 f1[1][2];
    {
      const int t = wave & 3, which = wave >> 2;
      floatx4 acc[4];
      zero_acc<4>(acc);
      if (which == 0) gemm_acc<2, 4>(fy, 1, 0, D1, 64, 24, 0, lane, acc);
      else gemm_acc<2, 4>(fy, 2, 0, T2, CF_T1ROW, 64, 0, lane, acc);
      wload<2>(f1[0], a.s2[0].w1, t, 2, lane);
      const float* bias = which == 0 ? a.s20.pwb1b : a.s20.pw2b;
      const int ch0 = t * 16 + 4 * g;
      const floatx4 b = *reinterpret_cast<const floatx4*>(bias + ch0);
#pragma unroll
      for (int p = 0; p < 4; ++p) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int c = ch0 + i;
          if (c < BF2)
            *reinterpret_cast<half_t*>(X2 + (p * 16 + col) * CF_X2ROW + shuffle_phys(2 * c + which, BF2, BFP2) * 2) =
                (half_t)fmaxf(acc[p][i] + b[i], 0.f);
        }
      }
    }
    __syncthreads();
    // ================= stage2.1-3 (stride 1) on the 64 pixels =================
#pragma unroll 1
    for (int b = 0; b < 3; ++b)
      s1_block<BF2, BFP2, 8, 2, 1>(X2, CF_X2ROW, RA, T2, CF_T1ROW, a.s2[b], b + 1 < 3 ? a.s2[b + 1].w1 : nullptr, 1, (wave >> 2) * 32, wave & 3,
                                   4, f1, tid, reinterpret_cast<float*>(smem + CF_LDS), reinterpret_cast<const float*>(smem + CF_LDS + CF_DW) + b * 2 * BFP2);

    // ================= stage3.0 (stride 2): 8x8x116 -> 4x4x232 =================
    char* T1c = RA;                         // [64 px][272]
    char* D1c = RA + 64 * CF_X2ROW;         // [16 px][272] branch1 depthwise (128 physical channels)
    char* T2c = D1c + 16 * CF_X2ROW;        // [16 px][272]
    char* X3 = T2c + 16 * CF_X2ROW;         // [16 px][528]
    {
      u32x4 fa[4], fb[4], fc[4];
      wload<4>(fa, a.s30.pw1, wave, 4, lane);
      wload<4>(fb, a.s30.pwb1, wave, 4, lane);
      wload<4>(fc, a.s30.pw2, wave, 4, lane);
      for (int i = tid; i < 16 * CF_X3ROW / 16; i += CN_THREADS) *reinterpret_cast<u32x4*>(X3 + i * 16) = u32x4{0u, 0u, 0u, 0u};
      floatx4 acc[4];
      zero_acc<4>(acc);
      gemm_acc<4, 4>(fa, 4, 0, X2, CF_X2ROW, 128, 0, lane, acc);
      store_relu<4>(T1c, CF_X2ROW, 0, wave, a.s30.pw1b, acc, lane);
      dwconv<128, 8, 4, 2>(X2, CF_X2ROW, D1c, CF_X2ROW, S2P + 880, S2P + 880 + 9 * 128, 1, tid);
      __syncthreads();
      dwconv<128, 8, 4, 2>(T1c, CF_X2ROW, T2c, CF_X2ROW, S2P + 2160, S2P + 2160 + 9 * 128, 1, tid);
      __syncthreads();
      floatx4 y1[1], y2[1];
      zero_acc<1>(y1);
      zero_acc<1>(y2);
      gemm_acc<4, 1>(fb, 4, 0, D1c, CF_X2ROW, 128, 0, lane, y1);
      gemm_acc<4, 1>(fc, 4, 0, T2c, CF_X2ROW, 128, 0, lane, y2);
      const int ch0 = wave * 16 + 4 * g;
      const floatx4 b1 = *reinterpret_cast<const floatx4*>(a.s30.pwb1b + ch0), b2 = *reinterpret_cast<const floatx4*>(a.s30.pw2b + ch0);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int c = ch0 + i;
        if (c < BF3) {
          half2v pr;
          pr[0] = (half_t)fmaxf(y1[0][i] + b1[i], 0.f);
          pr[1] = (half_t)fmaxf(y2[0][i] + b2[i], 0.f);
          *reinterpret_cast<half2v*>(X3 + col * CF_X3ROW + shuffle_phys(2 * c, BF3, BFP3) * 2) = pr;
        }
      }
    }
    __syncthreads();
    // ---- 16 px x 512 B -> global
    {
      const int px = tid >> 5, ck = tid & 31;
      *reinterpret_cast<u32x4*>(reinterpret_cast<char*>(a.out) + ((size_t)r * 16 + px) * 512 + ck * 16) =
          *reinterpret_cast<const u32x4*>(X3 + px * CF_X3ROW + ck * 16);
    }
    __syncthreads();
  }
}

// =====================================================================================================================
// cls_back: stage3.1-7, stage4.0-3, conv5 + ReLU, mean, fc, softmax, arg-max, scatter; 4 ROIs per workgroup pass.
// LDS (bytes):  X3 [64 px][528]                       33792
//               RB: T1 [64][<= 496]                    31744
//               RC: T2 [64][272] / D1 + T2' / Mn + LG  17408
//               X4 [16 px][976]                        15616
// =====================================================================================================================
#define CB_RB 33792
#define CB_RC (CB_RB + 31744)
#define CB_X4 (CB_RC + 17408)
#define CB_LDS (CB_X4 + 15616)
#define CB_DW (10 * 240 * 4)     /* depthwise parameters of the running stride-1 block, fp32 [10][<= 240] */
#define CB_BI3 (7 * 2 * 128 * 4) /* pointwise biases of stage 3's stride-1 blocks, fp32 [7][2][128] */
#define CB_BI (CB_BI3 + 3 * 2 * 240 * 4)  /* + stage 4's, fp32 [3][2][240] */
#define CB_S2 ((10 * 256 + 10 * 240) * 4)  /* depthwise parameters of stage4.0: branch1 over 256 physical channels | branch2 over 240 */
#define CB_T3ROW 272   /* 128 ch x 2 + 16 */
#define CB_T4ROW 496   /* 240 ch x 2 + 16 */
#define CB_X4ROW 976   /* 480 ch x 2 + 16 */
#define CB_MROW 2064   /* 1024 ch x 2 + 16 */

__global__ __launch_bounds__(CN_THREADS) void cls_back_kernel(const ClsBackArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid0 = threadIdx.x, lane0 = tid0 & 63;
  const int R = *a.m_dyn;
  const int ngroups = (R + 3) >> 2;
  if ((int)blockIdx.x >= ngroups) return;
  char* X3 = smem;
  char* RB = smem + CB_RB;
  char* RC = smem + CB_RC;
  char* X4 = smem + CB_X4;
  // X4's padding channels are never written by the shuffle stores and must read as zero
  for (int i = tid0; i < 16 * CB_X4ROW / 16; i += CN_THREADS) *reinterpret_cast<u32x4*>(X4 + i * 16) = u32x4{0u, 0u, 0u, 0u};
  // pointwise biases of both stages' stride-1 blocks, once per workgroup (published by the first barrier of the loop)
  stage_biases<BFP3, 7>(reinterpret_cast<float*>(smem + CB_LDS + CB_DW), a.s3, tid0);
  stage_biases<BFP4, 3>(reinterpret_cast<float*>(smem + CB_LDS + CB_DW + CB_BI3), a.s4, tid0);
  float* S2P = reinterpret_cast<float*>(smem + CB_LDS + CB_DW + CB_BI);   // stage4.0's depthwise parameters
  stage_dw<256>(S2P, a.s40.dw1, a.s40.dw1b, tid0);
  stage_dw<240>(S2P + 2560, a.s40.dw2, a.s40.dw2b, tid0);

  for (int grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
    // per-lane addresses must not be hoisted out of this loop: LICM otherwise keeps ~100 loop-invariant address registers
    // live across the whole network and spills them (360 spilled registers measured)
    int tid = tid0, lane = lane0;
    asm volatile("" : "+v"(tid), "+v"(lane));
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, col = lane & 15;
    const int roi0 = grp * 4;
    const int nroi = (R - roi0) < 4 ? (R - roi0) : 4;
    u32x4 f3[1][4];
    wload<4>(f3[0], a.s3[0].w1, wave, 4, lane);
    // ---- 4 ROIs x 16 px x 512 B -> X3 (missing ROIs: zeros)
    for (int i = tid; i < 64 * 32; i += CN_THREADS) {
      const int px = i >> 5, ck = i & 31;
      u32x4 v = u32x4{0u, 0u, 0u, 0u};
      if (px < nroi * 16) v = *reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(a.in) + ((size_t)roi0 * 16 + px) * 512 + ck * 16);
      *reinterpret_cast<u32x4*>(X3 + px * CF_X3ROW + ck * 16) = v;
    }
    __syncthreads();
    // ================= stage3.1-7 on 4 x (4x4) pixels =================
#pragma unroll 1
    for (int b = 0; b < 7; ++b)
      s1_block<BF3, BFP3, 4, 4, 1>(X3, CF_X3ROW, RB, RC, CB_T3ROW, a.s3[b], b + 1 < 7 ? a.s3[b + 1].w1 : nullptr, 4, 0, wave, 8, f3, tid,
                                   reinterpret_cast<float*>(smem + CB_LDS), reinterpret_cast<const float*>(smem + CB_LDS + CB_DW) + b * 2 * BFP3);

    // ================= stage4.0 (stride 2): 4 x (4x4x232) -> 4 x (2x2x464) =================
    char* T1 = RB;                       // [64 px][496]
    char* D1 = RC;                       // [16 px][528] branch1 depthwise over the 256 physical input channels
    char* T2 = RC + 16 * CF_X3ROW;       // [16 px][496]
    u32x4 f4[2][8];
    {
      // 15 tiles over 8 waves: round 1 of wave 7 recomputes tile 14 and stores nothing (see s1_block).  No 16-fragment
      // set is kept across a depthwise phase; each is requested right before the barrier that precedes its GEMM.
      // branch1: dw -> pw (result kept in 8 registers); branch2: pw -> dw -> pw.
      const int t0 = wave, t1 = wave + 8 < 15 ? wave + 8 : 14;
      const bool live1 = wave + 8 < 15;
      floatx4 y1[2][1];
      {
        u32x4 fb[2][8];
        wload<8>(fb[0], a.s40.pwb1, t0, 8, lane);
        wload<8>(fb[1], a.s40.pwb1, t1, 8, lane);
        dwconv<256, 4, 2, 2>(X3, CF_X3ROW, D1, CF_X3ROW, S2P, S2P + 9 * 256, 4, tid);
        __syncthreads();
        zero_acc<1>(y1[0]);
        zero_acc<1>(y1[1]);
        gemm_acc<8, 1>(fb[0], 8, 0, D1, CF_X3ROW, 256, 0, lane, y1[0]);
        gemm_acc<8, 1>(fb[1], 8, 0, D1, CF_X3ROW, 256, 0, lane, y1[1]);
      }
      asm volatile("" ::: "memory");
      {
        u32x4 fa[2][8];
        wload<8>(fa[0], a.s40.pw1, t0, 8, lane);
        wload<8>(fa[1], a.s40.pw1, t1, 8, lane);
        floatx4 acc[4];
        zero_acc<4>(acc);
        gemm_acc<8, 4>(fa[0], 8, 0, X3, CF_X3ROW, 256, 0, lane, acc);
        store_relu<4>(T1, CB_T4ROW, 0, t0, a.s40.pw1b, acc, lane);
        zero_acc<4>(acc);
        gemm_acc<8, 4>(fa[1], 8, 0, X3, CF_X3ROW, 256, 0, lane, acc);
        if (live1) store_relu<4>(T1, CB_T4ROW, 0, t1, a.s40.pw1b, acc, lane);
      }
      u32x4 fc[2][8];
      wload<8>(fc[0], a.s40.pw2, t0, 8, lane);
      wload<8>(fc[1], a.s40.pw2, t1, 8, lane);
      __syncthreads();
      dwconv<240, 4, 2, 2>(T1, CB_T4ROW, T2, CB_T4ROW, S2P + 2560, S2P + 2560 + 9 * 240, 4, tid);
      __syncthreads();
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        const int t = r ? t1 : t0;
        floatx4 y2[1];
        zero_acc<1>(y2);
        gemm_acc<8, 1>(fc[r], 8, 0, T2, CB_T4ROW, 240, 0, lane, y2);
        if (r == 0 || live1) {
          const int ch0 = t * 16 + 4 * g;
          const floatx4 b1 = *reinterpret_cast<const floatx4*>(a.s40.pwb1b + ch0), b2 = *reinterpret_cast<const floatx4*>(a.s40.pw2b + ch0);
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int c = ch0 + i;
            if (c < BF4) {
              half2v pr;
              pr[0] = (half_t)fmaxf(y1[r][0][i] + b1[i], 0.f);
              pr[1] = (half_t)fmaxf(y2[0][i] + b2[i], 0.f);
              *reinterpret_cast<half2v*>(X4 + col * CB_X4ROW + shuffle_phys(2 * c, BF4, BFP4) * 2) = pr;
            }
          }
        }
      }
      asm volatile("" ::: "memory");
      wload<8>(f4[0], a.s4[0].w1, t0, 8, lane);
      wload<8>(f4[1], a.s4[0].w1, t1, 8, lane);
    }
    __syncthreads();
    // ================= stage4.1-3 on 4 x (2x2) pixels =================
#pragma unroll 1
    for (int b = 0; b < 3; ++b)
      s1_block<BF4, BFP4, 2, 1, 2, true>(X4, CB_X4ROW, RB, RB + 16 * CB_T4ROW, CB_T4ROW, a.s4[b], b + 1 < 3 ? a.s4[b + 1].w1 : nullptr, 4, 0, wave, 8, f4, tid,
                                         reinterpret_cast<float*>(smem + CB_LDS), reinterpret_cast<const float*>(smem + CB_LDS + CB_DW + CB_BI3) + b * 2 * BFP4);

    // ================= conv5 1x1 (464 -> 1024) + ReLU, mean over the 2x2 map =================
    char* Mn = RC;                                              // [4 ROIs][1024] fp16
    float* LG = reinterpret_cast<float*>(RC + 4 * CB_MROW);     // [4][nc_p] fp32 logits
    {
      // 64 output tiles, 8 per wave; K = 480 physical channels = 15 steps in two chunks; the next tile's first chunk is
      // requested before this tile's MFMAs
      u32x4 fa[8], fb[8];
      wload<8>(fa, a.w5 + (size_t)(wave * 8) * 15 * 64, 0, 8, lane);
#pragma unroll 1
      for (int tq = 0; tq < 8; ++tq) {
        const int t = wave * 8 + tq;
        const u32x4_t* wt = a.w5 + (size_t)t * 15 * 64;
#pragma unroll
        for (int s = 0; s < 8; ++s) fb[s] = s < 7 ? wt[(8 + s) * 64 + lane] : u32x4{0u, 0u, 0u, 0u};
        floatx4 acc[1];
        zero_acc<1>(acc);
        gemm_acc<8, 1>(fa, 8, 0, X4, CB_X4ROW, 480, 0, lane, acc);
        if (tq + 1 < 8) wload<8>(fa, a.w5 + (size_t)(t + 1) * 15 * 64, 0, 8, lane);
        gemm_acc<8, 1>(fb, 7, 8, X4, CB_X4ROW, 480, 0, lane, acc);
        const int ch0 = t * 16 + 4 * g;
        const floatx4 bias = *reinterpret_cast<const floatx4*>(a.b5 + ch0);
        half4 q;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          float v = fmaxf(acc[0][i] + bias[i], 0.f);
          v += __shfl_xor(v, 1);   // pixels 4r..4r+3 of ROI r sit on 4 adjacent lanes
          v += __shfl_xor(v, 2);
          q[i] = (half_t)(v * 0.25f);
        }
        if ((col & 3) == 0) *reinterpret_cast<half4*>(Mn + (col >> 2) * CB_MROW + ch0 * 2) = q;
      }
    }
    __syncthreads();
    // ================= fc: logits[roi][class] = Wfc . mean + b =================
    {
      const int Tfc = a.nc_p >> 4;
      for (int t = wave; t < Tfc; t += CN_WAVES) {
        floatx4 acc = floatx4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
        for (int s0 = 0; s0 < 32; s0 += 8) {
          u32x4 af[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) af[u] = a.wfc[((size_t)t * 32 + s0 + u) * 64 + lane];
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            u32x4 v = u32x4{0u, 0u, 0u, 0u};
            if (col < 4) v = *reinterpret_cast<const u32x4*>(Mn + col * CB_MROW + (4 * (s0 + u) + g) * 16);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8, af[u]), __builtin_bit_cast(half8, v), acc, 0, 0, 0);
          }
        }
        const int c0 = t * 16 + 4 * g;
        if (col < 4) {
#pragma unroll
          for (int i = 0; i < 4; ++i) LG[col * a.nc_p + c0 + i] = acc[i] + a.bfc[c0 + i];
        }
      }
    }
    __syncthreads();
    // ================= softmax + arg-max (e2e.py:394-396), one wave per ROI; scatter into the detection records =====
    if (wave < nroi) {
      const int r = roi0 + wave;
      const float* l = LG + wave * a.nc_p;
      float mx = -INFINITY;
      for (int c = lane; c < a.nc; c += 64) mx = fmaxf(mx, l[c]);
      for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
      float sum = 0.f;
      for (int c = lane; c < a.nc; c += 64) sum += expf(l[c] - mx);
      for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
      float best = -1.f;
      int best_c = 0x7fffffff;
      for (int c = lane; c < a.nc; c += 64) {
        const float pr = expf(l[c] - mx) / sum;
        if (a.probs) a.probs[(long)r * a.nc + c] = pr;
        if (pr > best) { best = pr; best_c = c; }
      }
      for (int o = 32; o > 0; o >>= 1) {
        const float ob = __shfl_xor(best, o);
        const int oc = __shfl_xor(best_c, o);
        if (ob > best || (ob == best && oc < best_c)) { best = ob; best_c = oc; }
      }
      if (lane == 0) {
        if (a.ids) a.ids[r] = best_c;
        if (a.dets) {
          lp_det* d = a.dets + (long)a.roi_img[r] * a.max_det + a.roi_slot[r];
          d->cls_class = best_c;
          d->cls_conf = best;
        }
      }
      if (a.logits)
        for (int c = lane; c < a.nc; c += 64) a.logits[(long)r * a.logits_pitch + c] = l[c];
    }
    __syncthreads();
  }
}

void launch_cls_front(const ClsFrontArgs& a, int max_items, hipStream_t st) {
  set_max_dynamic_lds(reinterpret_cast<const void*>(cls_front_kernel), 160 * 1024);
  int grid = max_items < 512 ? max_items : 512;
  if (grid < 1) grid = 1;
  LP_LAUNCH(cls_front_kernel, dim3(grid), dim3(CN_THREADS), CF_LDS + CF_DW + CF_BI + CF_S2, st, a);
  LP_HIP(hipGetLastError());
}

void launch_cls_back(const ClsBackArgs& a, int max_items, hipStream_t st) {
  LP_CHECK(a.nc_p % 16 == 0 && a.nc_p <= 1024 && (size_t)4 * CB_MROW + (size_t)4 * a.nc_p * 4 <= 17408, LP_ERR_STATE,
           "fused classifier: %d classes do not fit the logits buffer", a.nc);
  set_max_dynamic_lds(reinterpret_cast<const void*>(cls_back_kernel), 160 * 1024);
  int grid = (max_items + 3) / 4;
  if (grid > 256) grid = 256;
  if (grid < 1) grid = 1;
  LP_LAUNCH(cls_back_kernel, dim3(grid), dim3(CN_THREADS), CB_LDS + CB_DW + CB_BI + CB_S2, st, a);
  LP_HIP(hipGetLastError());
}

// conv1 (+BN) for the MFMA stem.  K index of lane group g, element j: g < 3: window row ky = g, byte j = kx*3 + c (j < 8);
// g = 3: j < 3: the 9th byte (kx = 2, c = 2) of window row ky = j; rest zero.
void pack_cls_stem(const std::vector<float>& w, const std::vector<float>& bias, std::vector<uint16_t>& frags, std::vector<float>& bias_cases) {
  const double scale = 1.0 / 255.0 / 0.34, shift = 0.18 / 0.34;
  frags.assign((size_t)2 * 64 * 8, 0);
  for (int t = 0; t < 2; ++t)
    for (int lane = 0; lane < 64; ++lane) {
      const int gq = lane >> 4, m = lane & 15, co = t * 16 + m;
      if (co >= 24) continue;
      for (int j = 0; j < 8; ++j) {
        int ky, kb;
        if (gq < 3) { ky = gq; kb = j; }
        else if (j < 3) { ky = j; kb = 8; }
        else continue;
        const int kx = kb / 3, c = kb % 3;
        frags[((size_t)t * 64 + lane) * 8 + j] = f32_to_f16((float)((double)w[(((size_t)co * 3 + c) * 3 + ky) * 3 + kx] * scale));
      }
    }
  bias_cases.assign((size_t)4 * 32, 0.f);
  for (int ci = 0; ci < 4; ++ci)
    for (int co = 0; co < 24; ++co) {
      double s = 0.0;
      for (int ky = 0; ky < 3; ++ky)
        for (int kx = 0; kx < 3; ++kx) {
          if ((ci & 1) && ky == 0) continue;  // top row of the window is above the image
          if ((ci & 2) && kx == 0) continue;  // left column is left of the image
          for (int c = 0; c < 3; ++c) s += (double)w[(((size_t)co * 3 + c) * 3 + ky) * 3 + kx];
        }
      bias_cases[(size_t)ci * 32 + co] = (float)((double)bias[co] - shift * s);
    }
}

}  // namespace lp
