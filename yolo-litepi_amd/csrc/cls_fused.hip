// Fused ShuffleNetV2 stage (fp16): every stride-1 InvertedResidual of a stage in ONE launch.
//
// The un-fused classifier spends ~40 launches per batch on GEMMs of a few MFLOP each, all of
// them bound by the ~6 us launch/latency floor.  Here one workgroup (one wave per 16-channel
// tile: 4/8/16 waves) owns 64 pixels (1 ROI at
// 8x8, 4 ROIs at 4x4, 16 ROIs at 2x2) and keeps the stage tensor X = [x_lo | x_hi] resident in
// LDS across all blocks; per block (reference e2e.py:393 -> torchvision InvertedResidual):
//     t1 = relu(W1 . x_hi + b1)            MFMA, A fragments straight from L2 (weights are read
//     t2 = dw3x3(t1) + bd                  VALU                      once per workgroup per block)
//     y  = relu(W2 . t2 + b2)              MFMA
//     X  = channel_shuffle(cat(x_lo, y))   in place: read x_lo, barrier, write interleaved pairs
// MFMA orientation as in conv_kernels.hip: D[out-channel][pixel]; a D fragment holds 4
// consecutive channels of one pixel per lane, which is exactly the 8-byte store the next stage's
// B fragment (8 consecutive channels of one pixel) wants in LDS.
#include "cls_fused.h"

namespace lp {

typedef _Float16 half_t;
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

#define FS_PIX 64  // pixels per workgroup = 4 MFMA column tiles

// One wave = one 16-channel output tile x 4 pixel tiles.  The S weight fragments of a GEMM are
// requested together (prefetch_w) well before they are used, so a GEMM costs one L2 round trip
// that overlaps the preceding phase instead of one per K step.
template <int SMAX>
__device__ __forceinline__ void prefetch_w(const u32x4* __restrict__ wfrag, int S, int t, int lane, bool active,
                                           u32x4 (&af)[SMAX]) {
#pragma unroll
  for (int s = 0; s < SMAX; ++s) {
    af[s] = u32x4{0u, 0u, 0u, 0u};
    if (active && s < S) af[s] = wfrag[((size_t)t * S + s) * 64 + lane];
  }
}

template <int SMAX>
__device__ __forceinline__ void pw_gemm(const u32x4 (&af)[SMAX], const char* bsrc, int brow, int S, int K, int lane,
                                        floatx4 (&acc)[4]) {
  const int g = lane >> 4, col = lane & 15;
#pragma unroll
  for (int p = 0; p < 4; ++p) acc[p] = floatx4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int s = 0; s < SMAX; ++s) {
    if (s < S) {
      const half8 a = __builtin_bit_cast(half8, af[s]);
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        u32x4 v = u32x4{0u, 0u, 0u, 0u};  // K groups past the last physical channel: LDS there is row padding
        if ((4 * s + g) * 8 < K) v = *reinterpret_cast<const u32x4*>(bsrc + (p * 16 + col) * brow + (4 * s + g) * 16);
        acc[p] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, __builtin_bit_cast(half8, v), acc[p], 0, 0, 0);
      }
    }
  }
}

template <int NW, int SMAX>
__global__ __launch_bounds__(NW * 64) void shuffle_stage_kernel(const FusedStageArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NTHR = NW * 64;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, col = lane & 15;
  const int bfp = a.bfp, bf = a.bf, HW = a.HW, W = a.W;
  const int xrow = 2 * bfp * 2 + 16;  // LDS row pitches in bytes: odd number of 16-byte slots
  const int trow = bfp * 2 + 16;
  char* X = smem;
  char* T1 = X + FS_PIX * xrow;
  char* T2 = T1 + FS_PIX * trow;
  const int CGX = 2 * bfp / 8, CGT = bfp / 8;
  const int Tt = bfp / 16, S = (bfp + 31) / 32;
  const bool active = wave < Tt;  // this wave's output tile (waves beyond the tile count only move data)
  const int ch0 = wave * 16 + 4 * g;
  const int R = *a.m_dyn;
  const int ngroups = (R + a.group - 1) / a.group;
  const half_t* in = reinterpret_cast<const half_t*>(a.in);
  half_t* out = reinterpret_cast<half_t*>(a.out);

  for (int grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
    const int roi0 = grp * a.group;
    const int nvalid = ((R - roi0) < a.group ? (R - roi0) : a.group) * HW;
    const long pix0 = (long)roi0 * HW;
    // 16-wave workgroups are capped at 128 VGPRs: they keep one fragment set in flight, not two
    constexpr bool PREF_NEXT = NW < 16;
    u32x4 w1f[SMAX], w2f[SMAX];
    if (PREF_NEXT) prefetch_w<SMAX>(a.blk[0].w1, S, wave, lane, active, w1f);
    for (int i = tid; i < FS_PIX * CGX; i += NTHR) {
      const int px = i / CGX, cg = i - px * CGX;
      u32x4 v = u32x4{0u, 0u, 0u, 0u};
      if (px < nvalid) v = *reinterpret_cast<const u32x4*>(in + (pix0 + px) * a.in_pitch + cg * 8);
      *reinterpret_cast<u32x4*>(X + px * xrow + cg * 16) = v;
    }
    __syncthreads();

    for (int b = 0; b < a.nblk; ++b) {
      const FusedBlockW bw = a.blk[b];
      floatx4 acc[4];
      // ---- t1 = relu(W1 . x_hi + b1) -------------------------------------------------------
      if (!PREF_NEXT) prefetch_w<SMAX>(bw.w1, S, wave, lane, active, w1f);
      if (PREF_NEXT) prefetch_w<SMAX>(bw.w2, S, wave, lane, active, w2f);  // lands while pw1 + dw run
      if (active) {
        pw_gemm<SMAX>(w1f, X + bfp * 2, xrow, S, bfp, lane, acc);
        const floatx4 bias = *reinterpret_cast<const floatx4*>(bw.b1 + ch0);
#pragma unroll
        for (int p = 0; p < 4; ++p) {
          half4 q;
#pragma unroll
          for (int i = 0; i < 4; ++i) q[i] = (half_t)fmaxf(acc[p][i] + bias[i], 0.f);
          *reinterpret_cast<half4*>(T1 + (p * 16 + col) * trow + ch0 * 2) = q;
        }
      }
      if (PREF_NEXT && b + 1 < a.nblk) prefetch_w<SMAX>(a.blk[b + 1].w1, S, wave, lane, active, w1f);  // for the next block
      if (!PREF_NEXT) prefetch_w<SMAX>(bw.w2, S, wave, lane, active, w1f);  // re-use the one set: lands during dw
      __syncthreads();
      // ---- t2 = dw3x3(t1) + bd (pad 1, stride 1, inside each ROI's WxW map) -------------------
      for (int i = tid; i < FS_PIX * CGT; i += NTHR) {
        const int px = i / CGT, cg = i - px * CGT;
        const int rl = px / HW, pp = px - rl * HW;
        const int y = pp / W, x = pp - y * W;
        float acc8[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc8[j] = bw.dwb[cg * 8 + j];
        for (int ky = 0; ky < 3; ++ky) {
          const int yy = y - 1 + ky;
          if (yy < 0 || yy >= W) continue;
          for (int kx = 0; kx < 3; ++kx) {
            const int xx = x - 1 + kx;
            if (xx < 0 || xx >= W) continue;
            const half8 v = *reinterpret_cast<const half8*>(T1 + (rl * HW + yy * W + xx) * trow + cg * 16);
            const float* wr = bw.dw + (ky * 3 + kx) * bfp + cg * 8;
#pragma unroll
            for (int j = 0; j < 8; ++j) acc8[j] = fmaf((float)v[j], wr[j], acc8[j]);
          }
        }
        half8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (half_t)acc8[j];
        *reinterpret_cast<half8*>(T2 + px * trow + cg * 16) = o;
      }
      __syncthreads();
      // ---- y = relu(W2 . t2 + b2); X = shuffle(cat(x_lo, y)) in place --------------------------
      half4 x1v[4];
      if (active) {
        if (PREF_NEXT) pw_gemm<SMAX>(w2f, T2, trow, S, bfp, lane, acc); else pw_gemm<SMAX>(w1f, T2, trow, S, bfp, lane, acc);
#pragma unroll
        for (int p = 0; p < 4; ++p) x1v[p] = *reinterpret_cast<const half4*>(X + (p * 16 + col) * xrow + ch0 * 2);
      }
      __syncthreads();  // every x_lo value is in registers before any interleaved pair is written
      if (active) {
        const floatx4 bias = *reinterpret_cast<const floatx4*>(bw.b2 + ch0);
#pragma unroll
        for (int p = 0; p < 4; ++p) {
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int c = ch0 + i;
            if (c < bf) {
              const int l = 2 * c;
              const int phys = l < bf ? l : bfp + (l - bf);
              half2v pr;
              pr[0] = x1v[p][i];
              pr[1] = (half_t)fmaxf(acc[p][i] + bias[i], 0.f);
              *reinterpret_cast<half2v*>(X + (p * 16 + col) * xrow + phys * 2) = pr;
            }
          }
        }
      }
      __syncthreads();
    }

    for (int i = tid; i < FS_PIX * CGX; i += NTHR) {
      const int px = i / CGX, cg = i - px * CGX;
      if (px < nvalid)
        *reinterpret_cast<u32x4*>(out + (pix0 + px) * a.out_pitch + cg * 8) = *reinterpret_cast<const u32x4*>(X + px * xrow + cg * 16);
    }
    __syncthreads();
  }
}

size_t fused_stage_lds_bytes(int bfp) { return (size_t)FS_PIX * ((2 * bfp * 2 + 16) + 2 * (bfp * 2 + 16)); }

void launch_fused_stage(const FusedStageArgs& a, int max_items, hipStream_t st) {
  const size_t lds = fused_stage_lds_bytes(a.bfp);
  LP_CHECK(lds <= 160 * 1024 && a.bfp % 16 == 0 && FS_PIX % a.HW == 0 && a.group * a.HW == FS_PIX, LP_ERR_STATE,
           "fused ShuffleNet stage: unsupported geometry (bfp %d, HW %d)", a.bfp, a.HW);
  const int Tt = a.bfp / 16, S = (a.bfp + 31) / 32;
  int groups = (max_items + a.group - 1) / a.group;
  if (groups > 1024) groups = 1024;
  if (groups < 1) groups = 1;
#define LP_FS(NW, SM)                                                                                          \
  {                                                                                                            \
    set_max_dynamic_lds(reinterpret_cast<const void*>(shuffle_stage_kernel<NW, SM>), 160 * 1024);                                                                                                \
    LP_LAUNCH((shuffle_stage_kernel<NW, SM>), dim3(groups), dim3(NW * 64), lds, st, a);                 \
  }
  if (Tt <= 4 && S <= 2) LP_FS(4, 2) else if (Tt <= 8 && S <= 4) LP_FS(8, 4) else if (Tt <= 16 && S <= 8) LP_FS(16, 8) else
    throw Error(LP_ERR_STATE, "fused ShuffleNet stage: too many channel tiles");
#undef LP_FS
  LP_HIP(hipGetLastError());
}

// ------------------------------------------------------------------------------------
// Fused classifier head (fp16): conv5 1x1 (Cin -> 1024) + ReLU, x.mean([2,3]), fc, softmax,
// arg-max and the scatter of (class, confidence) into the detection records -- reference
// e2e.py:393-396 (torchvision conv5 / mean / fc) + e2e.py:525-526.  One 16-wave workgroup owns
// 16 ROIs (64 pixels of the 2x2 stage-4 map); activations stay in LDS, the 1024-channel conv5
// output never exists in memory: each wave reduces its accumulator tiles over the 4 pixels of a ROI
// with two lane shuffles and writes the fp16 mean straight into the fc operand image.
// ------------------------------------------------------------------------------------
#define FH_ROIS 16
__global__ __launch_bounds__(1024) void cls_head_kernel(const FusedHeadArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, col = lane & 15;
  const int cin = a.cin_p;                 // physical input channels (two padded halves)
  const int xrow = cin * 2 + 16;
  const int mrow = 1024 * 2 + 16;
  char* X = smem;                          // [64 px][cin] fp16
  char* Mn = X + 64 * xrow;                // [16 rois][1024] fp16 (mean of relu(conv5))
  float* LG = reinterpret_cast<float*>(Mn + FH_ROIS * mrow);  // [16 rois][nc_p] fp32 logits
  const int S5 = (cin + 31) / 32, CGX = cin / 8;
  const int R = *a.m_dyn;
  const int ngroups = (R + FH_ROIS - 1) / FH_ROIS;
  const half_t* in = reinterpret_cast<const half_t*>(a.in);

  for (int grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
    const int roi0 = grp * FH_ROIS;
    const int nroi = (R - roi0) < FH_ROIS ? (R - roi0) : FH_ROIS;
    for (int i = tid; i < 64 * CGX; i += 1024) {
      const int px = i / CGX, cg = i - px * CGX;
      u32x4 v = u32x4{0u, 0u, 0u, 0u};
      if (px < nroi * 4) v = *reinterpret_cast<const u32x4*>(in + ((long)roi0 * 4 + px) * a.in_pitch + cg * 8);
      *reinterpret_cast<u32x4*>(X + px * xrow + cg * 16) = v;
    }
    __syncthreads();
    // ---- conv5: 64 output tiles of 16 channels, 4 per wave; mean over the 4 pixels of each ROI -----
#pragma unroll 1
    for (int tq = 0; tq < 4; ++tq) {
      const int t = wave * 4 + tq;
      floatx4 acc[4];
#pragma unroll
      for (int p = 0; p < 4; ++p) acc[p] = floatx4{0.f, 0.f, 0.f, 0.f};
      for (int s0 = 0; s0 < S5; s0 += 8) {
        u32x4 af[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          af[u] = u32x4{0u, 0u, 0u, 0u};
          if (s0 + u < S5) af[u] = a.w5[((size_t)t * S5 + s0 + u) * 64 + lane];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int s = s0 + u;
          if (s < S5) {
            const half8 av = __builtin_bit_cast(half8, af[u]);
#pragma unroll
            for (int p = 0; p < 4; ++p) {
              u32x4 v = u32x4{0u, 0u, 0u, 0u};
              if ((4 * s + g) * 8 < cin) v = *reinterpret_cast<const u32x4*>(X + (p * 16 + col) * xrow + (4 * s + g) * 16);
              acc[p] = __builtin_amdgcn_mfma_f32_16x16x32_f16(av, __builtin_bit_cast(half8, v), acc[p], 0, 0, 0);
            }
          }
        }
      }
      const int ch0 = t * 16 + 4 * g;
      const floatx4 bias = *reinterpret_cast<const floatx4*>(a.b5 + ch0);
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        half4 q;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          float v = fmaxf(acc[p][i] + bias[i], 0.f);
          v += __shfl_xor(v, 1);   // pixels 4r..4r+3 of ROI r sit on 4 adjacent lanes
          v += __shfl_xor(v, 2);
          q[i] = (half_t)(v * 0.25f);
        }
        if ((col & 3) == 0) *reinterpret_cast<half4*>(Mn + (p * 4 + (col >> 2)) * mrow + ch0 * 2) = q;
      }
    }
    __syncthreads();
    // ---- fc: logits[roi][class] = Wfc . mean + b, one 16-class tile per wave ------------------------
    const int Tfc = a.nc_p / 16;
    if (wave < Tfc) {
      floatx4 acc = floatx4{0.f, 0.f, 0.f, 0.f};
      for (int s0 = 0; s0 < 32; s0 += 8) {
        u32x4 af[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) af[u] = a.wfc[((size_t)wave * 32 + s0 + u) * 64 + lane];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const u32x4 v = *reinterpret_cast<const u32x4*>(Mn + col * mrow + (4 * (s0 + u) + g) * 16);
          acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8, af[u]), __builtin_bit_cast(half8, v), acc, 0, 0, 0);
        }
      }
      const int c0 = wave * 16 + 4 * g;
#pragma unroll
      for (int i = 0; i < 4; ++i) LG[col * a.nc_p + c0 + i] = acc[i] + a.bfc[c0 + i];
    }
    __syncthreads();
    // ---- softmax + arg-max, one wave per ROI ------------------------------------------------------------
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

size_t fused_head_lds_bytes(int cin_p, int nc_p) { return (size_t)64 * (cin_p * 2 + 16) + (size_t)FH_ROIS * (1024 * 2 + 16) + (size_t)FH_ROIS * nc_p * 4; }

void launch_fused_head(const FusedHeadArgs& a, int max_items, hipStream_t st) {
  const size_t lds = fused_head_lds_bytes(a.cin_p, a.nc_p);
  LP_CHECK(lds <= 160 * 1024 && a.nc_p % 16 == 0 && a.nc_p <= 256 && a.cin_p % 8 == 0, LP_ERR_STATE,
           "fused classifier head: unsupported geometry (cin %d, classes %d)", a.cin_p, a.nc);
  set_max_dynamic_lds(reinterpret_cast<const void*>(cls_head_kernel), 160 * 1024);
  int groups = (max_items + FH_ROIS - 1) / FH_ROIS;
  if (groups > 512) groups = 512;
  LP_LAUNCH(cls_head_kernel, dim3(groups), dim3(1024), lds, st, a);
  LP_HIP(hipGetLastError());
}

// A fragments of a pointwise conv over PHYSICAL channels: [tile][step][lane][8 halfs];
// tile t row m = output channel 16t+m, K group q = 4s+g = input channels [8q, 8q+8).
std::vector<uint16_t> pack_fused_pw(const std::vector<float>& w_phys, int cout_p, int cin_p) {
  const int Tt = cout_p / 16, S = (cin_p + 31) / 32;
  std::vector<uint16_t> buf((size_t)Tt * S * 64 * 8, 0);
  for (int t = 0; t < Tt; ++t)
    for (int s = 0; s < S; ++s)
      for (int lane = 0; lane < 64; ++lane) {
        const int g = lane >> 4, m = lane & 15;
        const int oc = t * 16 + m;
        for (int j = 0; j < 8; ++j) {
          const int ci = (4 * s + g) * 8 + j;
          if (ci < cin_p) buf[(((size_t)t * S + s) * 64 + lane) * 8 + j] = f32_to_f16(w_phys[(size_t)oc * cin_p + ci]);
        }
      }
  return buf;
}

}  // namespace lp
