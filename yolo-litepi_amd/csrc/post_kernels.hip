// Detector post-processing and ROI extraction on the GPU.  Replaces the NumPy code of
// NCNNDetector.postprocess / nms_numpy (reference e2e.py:240-296, 89-119), the ROI loop of
// HybridPipeline.run (e2e.py:465-473) and the PIL resize of PyTorchClassifier.predict_batch
// (e2e.py:385-389).  Everything that decides WHICH boxes survive is computed with explicit
// round-to-nearest fp32 operations in the reference's operation order (no FMA contraction),
// so that for the same out0 tensor the kept set is identical to the NumPy result.
#include "common.h"
#include "kernels.h"
#include "post_dev.h"
#include <cstdlib>

namespace lp {

typedef _Float16 half_t;

// ------------------------------------------------------------------------------------
// Detect head decode (model.ncnn.param:184-208): per anchor, softmax over reg_max bins of
// each box side, expectation with the DFL weights, dist2bbox around the anchor point,
// x stride; class sigmoid.  One thread per (image, anchor).
// ------------------------------------------------------------------------------------
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef float float4_t __attribute__((ext_vector_type(4)));
template <typename T> struct DecVec;
template <> struct DecVec<half_t> { typedef half8_t type; static constexpr int G = 8; };
template <> struct DecVec<float> { typedef float4_t type; static constexpr int G = 4; };
template <typename T> __device__ __forceinline__ float dec_exp(float x);
template <> __device__ __forceinline__ float dec_exp<half_t>(float x) { return __expf(x); }
template <> __device__ __forceinline__ float dec_exp<float>(float x) { return expf(x); }

// Four adjacent lanes share one anchor, one box side each: a wave reads 16 anchors x 4*reg_max
// channels = fully coalesced 16-byte loads.  The side-0 lane then gathers the four distances
// with wave shuffles and finishes the anchor.
template <typename T, int RM>
__global__ __launch_bounds__(256) void decode_kernel(const DecodeArgs a) {
  typedef typename DecVec<T>::type vec;
  constexpr int G = DecVec<T>::G;
  const int n = blockIdx.y;
  const int gid = blockIdx.x * 256 + threadIdx.x;
  int anchor = gid >> 2;
  const int side = gid & 3;
  const bool live = anchor < a.A;
  anchor = live ? anchor : a.A - 1;
  int l = 0;
  for (int i = 1; i < a.nlevels; ++i)
    if (anchor >= a.lv[i].anchor_off) l = i;
  const DecodeLevel lv = a.lv[l];
  const long pix = (long)n * lv.H * lv.W + (anchor - lv.anchor_off);
  const T* b = reinterpret_cast<const T*>(lv.box) + pix * lv.box_pitch + side * RM;
  float v[RM];
#pragma unroll
  for (int q = 0; q < RM / G; ++q) {
    const vec x = *reinterpret_cast<const vec*>(b + q * G);
#pragma unroll
    for (int i = 0; i < G; ++i) v[q * G + i] = (float)x[i];
  }
  float mx = v[0];
#pragma unroll
  for (int i = 1; i < RM; ++i) mx = fmaxf(mx, v[i]);
  float sum = 0.f, ex = 0.f;
#pragma unroll
  for (int i = 0; i < RM; ++i) {
    const float e = dec_exp<T>(v[i] - mx);
    sum += e;
    ex += e * a.dfl_w[i];
  }
  const float dist = ex / sum;
  const int lane = threadIdx.x & 63, base = lane & ~3;
  const float d0 = __shfl(dist, base), d1 = __shfl(dist, base + 1), d2 = __shfl(dist, base + 2), d3 = __shfl(dist, base + 3);
  if (side != 0 || !live) return;
  const T* cls = reinterpret_cast<const T*>(lv.cls) + pix * lv.cls_pitch;
  const float ax = a.anchors[anchor], ay = a.anchors[a.A + anchor], s = a.strides[anchor];
  const float x1 = ax - d0, y1 = ay - d1, x2 = ax + d2, y2 = ay + d3;
  const float cx = (x1 + x2) * 0.5f * s, cy = (y1 + y2) * 0.5f * s;
  const float w = (x2 - x1) * s, h = (y2 - y1) * s;
  float best = -1.f;
  int best_c = 0;
  float* o = a.out0 ? a.out0 + (long)n * (4 + a.nc) * a.A + anchor : nullptr;
  for (int c = 0; c < a.nc; ++c) {
    const float sc = 1.f / (1.f + dec_exp<T>(-(float)cls[c]));
    if (o) o[(long)(4 + c) * a.A] = sc;
    if (sc > best) { best = sc; best_c = c; }
  }
  if (o) {
    o[0] = cx; o[(long)a.A] = cy; o[2L * a.A] = w; o[3L * a.A] = h;
  }
  emit_candidate(cx, cy, w, h, best, best_c, anchor, a.geom[n], a.conf, a.cand + (long)n * a.A, a.cand_count + n);
}

void launch_decode(int prec, const DecodeArgs& a, int N, hipStream_t st) {
  dim3 grid(ceil_div(a.A * 4, 256), N);
  LP_CHECK(a.reg_max == 16 || a.reg_max == 8 || a.reg_max == 32, LP_ERR_GRAPH, "reg_max %d unsupported (8, 16, 32)", a.reg_max);
#define LP_DEC(TT)                                                                           \
  if (a.reg_max == 16) LP_LAUNCH((decode_kernel<TT, 16>), grid, dim3(256), 0, st, a); \
  else if (a.reg_max == 8) LP_LAUNCH((decode_kernel<TT, 8>), grid, dim3(256), 0, st, a); \
  else LP_LAUNCH((decode_kernel<TT, 32>), grid, dim3(256), 0, st, a);
  if (prec == LP_FP16) { LP_DEC(half_t) } else { LP_DEC(float) }
#undef LP_DEC
  LP_HIP(hipGetLastError());
}

__global__ __launch_bounds__(256) void filter_out0_kernel(const float* __restrict__ out0, int nc, int A,
                                                          const ImgGeom* __restrict__ geom, Cand* cand, int* cand_count,
                                                          float conf) {
  const int n = blockIdx.y;
  const int anchor = blockIdx.x * 256 + threadIdx.x;
  if (anchor >= A) return;
  const float* o = out0 + (long)n * (4 + nc) * A + anchor;
  float best = -INFINITY;
  int best_c = 0;
  for (int c = 0; c < nc; ++c) {
    const float sc = o[(long)(4 + c) * A];
    if (sc > best) { best = sc; best_c = c; }
  }
  emit_candidate(o[0], o[(long)A], o[2L * A], o[3L * A], best, best_c, anchor, geom[n], conf, cand + (long)n * A,
                 cand_count + n);
}

void launch_filter_out0(const float* out0, int nc, int A, const ImgGeom* geom, Cand* cand, int* cand_count, float conf,
                        int N, hipStream_t st) {
  dim3 grid(ceil_div(A, 256), N);
  LP_LAUNCH(filter_out0_kernel, grid, dim3(256), 0, st, out0, nc, A, geom, cand, cand_count, conf);
  LP_HIP(hipGetLastError());
}

// ------------------------------------------------------------------------------------
// Per-class greedy NMS, one workgroup per image.
//   1. key = (0xFFFF - class, score bits, anchor): a descending bitonic sort in LDS gives
//      class ascending, score descending, ties -> higher anchor first (the order a stable
//      ascending argsort reversed gives; e2e.py:96).
//   2. greedy sweep over the sorted boxes in chunks of 64 (one box per lane of wave 0): inside a
//      chunk the survivors are found with wave64 ballot masks -- box i, still alive, suppresses
//      every later lane of its class with IoU > thr, the ballot of those lanes clears them from
//      the alive mask, no barrier; the chunk's survivors then suppress all later chunks in
//      parallel over the workgroup (one barrier pair per 64 boxes instead of one per kept box).
//      IoU is the reference's fp32 expression inter / (area_i + area_j - inter + 1e-6)
//      (e2e.py:106-116) in its operation order, so the kept set is the NumPy one.
//   3. max_det: the reference keeps every survivor.  When more than max_det survive, the
//      max_det highest-scoring ones (over all classes) stay and are emitted in the reference's
//      class-major order; with a single class that is the head of the list, so the sweep stops early.
//   4. ROI rectangle (int truncation, clip) and area filter (e2e.py:465-473) of the kept
//      boxes, order-preserving compaction into the lp_det records, and -- when a ROI table is
//      given -- the image's slice of the batch-wide ROI list (one atomicAdd per image).
// ------------------------------------------------------------------------------------
#define NMS_THREADS 1024

size_t nms_lds_bytes(int A) {
  int npad = 1;
  while (npad < A) npad <<= 1;
  return (size_t)npad * 8 + (size_t)round_up(A, 16) + 16;
}

__device__ __forceinline__ unsigned long long nms_key(const Cand& c) {
  return ((unsigned long long)(0xFFFFu - (unsigned)c.cls) << 46) | ((unsigned long long)__float_as_uint(c.score) << 14) |
         (unsigned long long)(c.anchor & 0x3FFF);
}
// score-major key of a kept box (global top-max_det selection): unique per image
__device__ __forceinline__ unsigned long long nms_score_key(const Cand& c) {
  return ((unsigned long long)__float_as_uint(c.score) << 14) | (unsigned long long)(c.anchor & 0x3FFF);
}
// true when box j (area aj) is suppressed by kept box i (area ai): NOT (iou <= thr), e2e.py:116
__device__ __forceinline__ bool nms_suppressed(float ix1, float iy1, float ix2, float iy2, float ai, float jx1, float jy1, float jx2,
                                               float jy2, float thr) {
  const float aj = __fmul_rn(__fsub_rn(jx2, jx1), __fsub_rn(jy2, jy1));
  const float w = fmaxf(0.f, __fsub_rn(fminf(ix2, jx2), fmaxf(ix1, jx1)));
  const float h = fmaxf(0.f, __fsub_rn(fminf(iy2, jy2), fmaxf(iy1, jy1)));
  const float inter = __fmul_rn(w, h);
  const float iou = __fdiv_rn(inter, __fadd_rn(__fsub_rn(__fadd_rn(ai, aj), inter), 1e-6f));
  return !(iou <= thr);
}

// exclusive prefix sum of one int per thread over the workgroup (NMS_THREADS = 16 waves); returns the total via *total
__device__ __forceinline__ int block_exclusive_scan(int v, int* s_wave /*[17]*/, int* total) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int inc = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int t = __shfl_up(inc, o);
    if (lane >= o) inc += t;
  }
  if (lane == 63) s_wave[wave] = inc;
  __syncthreads();
  if (wave == 0) {
    int w = lane < NMS_THREADS / 64 ? s_wave[lane] : 0;
    int winc = w;
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) {
      const int t = __shfl_up(winc, o);
      if (lane >= o) winc += t;
    }
    if (lane < NMS_THREADS / 64) s_wave[lane] = winc - w;
    if (lane == NMS_THREADS / 64 - 1) s_wave[NMS_THREADS / 64] = winc;
  }
  __syncthreads();
  const int r = s_wave[wave] + inc - v;
  *total = s_wave[NMS_THREADS / 64];
  __syncthreads();  // s_wave may be reused
  return r;
}

// ROI rectangle of a kept box (int truncation + clip) and the area filter.  rule 0 = HybridPipeline.run, e2e.py:465-473:
// x1 in [0, w-1], y1 in [0, h-1], x2 in [x1+1, w], y2 in [y1+1, h]; rule 1 = HybridPipelineOptimized.run,
// e2e_optimize.py:480-497: all four clipped to [0, w] / [0, h], empty rectangles dropped.  min_area < 0: no filter.
__device__ __forceinline__ bool roi_rect(const Cand& c, const ImgGeom& gm, int rule, int min_area, int& x1, int& y1, int& x2, int& y2) {
  x1 = (int)c.x1; y1 = (int)c.y1; x2 = (int)c.x2; y2 = (int)c.y2;
  if (rule == 0) {
    x1 = min(max(x1, 0), gm.w - 1); y1 = min(max(y1, 0), gm.h - 1);
    x2 = min(max(x2, x1 + 1), gm.w); y2 = min(max(y2, y1 + 1), gm.h);
  } else {
    x1 = min(max(x1, 0), gm.w); x2 = min(max(x2, 0), gm.w);
    y1 = min(max(y1, 0), gm.h); y2 = min(max(y2, 0), gm.h);
  }
  return min_area < 0 || (((x2 - x1) * (y2 - y1) >= min_area) && x2 > x1 && y2 > y1);
}

// At most 64 candidates (the usual image: a handful of anchors pass conf 0.25), none of which max_det can cut: ONE wave does the whole image
// in registers -- lane i loads candidate i, the 64 keys are bitonic-sorted with lane shuffles (21 compare-exchange steps, the
// source lane travels with the key), the records are gathered from their source lanes, then the same __ballot sweep, ROI
// rule, compaction and ROI-list bookkeeping as the general path below: identical results (same keys, same arithmetic), none of
// its ~12 sixteen-wave barriers, LDS passes and global round trips (scratch `sorted`, binary search).  25 -> ~8 us per launch.
__device__ __forceinline__ void nms_small(const NmsArgs& a, int n, int cnt, int nimg) {
  const int lane = threadIdx.x & 63;
  const Cand* cand = a.cand + (long)n * a.A;
  Cand c;
  c.x1 = c.y1 = c.x2 = c.y2 = c.score = 0.f; c.cls = -1; c.anchor = 0; c.pad = 0;
  if (lane < cnt) c = cand[lane];
  unsigned long long key = lane < cnt ? nms_key(c) : 0ull;   // (a real key is never 0: that would take class 65535)
  int src = lane;
  // (rolled on purpose: one wave executes this once per image, from a cold instruction cache -- 21 unrolled stages were ~2.5 KB of
  //  straight-line code, each 64-byte line a fetch from L2 that nothing hides)
#pragma unroll 1
  for (int k = 2; k <= 64; k <<= 1) {
#pragma unroll 1
    for (int j = k >> 1; j > 0; j >>= 1) {
      const unsigned lo = __shfl_xor((unsigned)key, j), hi = __shfl_xor((unsigned)(key >> 32), j);
      const unsigned long long okey = ((unsigned long long)hi << 32) | lo;
      const int osrc = __shfl_xor(src, j);
      const bool lower = (lane & j) == 0, desc = (lane & k) == 0;   // as the LDS sort: the lower index of a descending pair keeps the larger key
      const bool take = (lower == desc) ? (okey > key) : (okey < key);
      key = take ? okey : key;
      src = take ? osrc : src;
    }
  }
  Cand bj;
  bj.x1 = __shfl(c.x1, src); bj.y1 = __shfl(c.y1, src); bj.x2 = __shfl(c.x2, src); bj.y2 = __shfl(c.y2, src);
  bj.score = __shfl(c.score, src); bj.cls = __shfl(c.cls, src); bj.anchor = __shfl(c.anchor, src); bj.pad = 0;
  const bool valid = lane < cnt;   // the zero keys of the empty lanes sort behind every real one
  const float aj = __fmul_rn(__fsub_rn(bj.x2, bj.x1), __fsub_rn(bj.y2, bj.y1));
  unsigned long long alive = __ballot(valid);
  unsigned long long todo = alive, kept = 0ull;
  const float thr = a.iou;
  while (todo) {  // wave-uniform
    const int i = __ffsll((long long)todo) - 1;
    todo &= todo - 1;
    kept |= 1ull << i;
    const float ix1 = __shfl(bj.x1, i), iy1 = __shfl(bj.y1, i), ix2 = __shfl(bj.x2, i), iy2 = __shfl(bj.y2, i);
    const float ai = __shfl(aj, i);
    const int ic = __shfl(bj.cls, i);
    const bool sup = lane > i && ((alive >> lane) & 1ull) && bj.cls == ic && nms_suppressed(ix1, iy1, ix2, iy2, ai, bj.x1, bj.y1, bj.x2, bj.y2, thr);
    const unsigned long long m = __ballot(sup);
    alive &= ~m;
    todo &= ~m;
  }
  const bool is_kept = (kept >> lane) & 1ull;
  const int nsel = __popcll(kept);   // <= cnt <= max_det: nothing to cut
  const ImgGeom gm = a.geom[n];
  int x1 = 0, y1 = 0, x2 = 0, y2 = 0;
  const bool ok = is_kept && roi_rect(bj, gm, a.roi_rule, a.min_area, x1, y1, x2, y2);
  const unsigned long long okm = __ballot(ok);
  const int o = __popcll(okm & ((1ull << lane) - 1ull)), run = __popcll(okm);
  double ssum = is_kept ? (double)bj.score : 0.0;
  for (int off = 32; off > 0; off >>= 1) ssum += __shfl_xor(ssum, off);
  int base = 0;
  if (lane == 0) {
    a.counts[n] = run;
    a.counts[nimg + n] = nsel;
    reinterpret_cast<float*>(a.counts)[2 * nimg + n] = nsel > 0 ? (float)(ssum / (double)nsel) : 0.f;
    a.cand_count[n] = 0;  // ready for the next call
    if (a.tab.total) {   // (the ROI list protocol of the general path)
      base = atomicAdd(a.tab.work, run);
      int one = 1;
      asm volatile("" : "+v"(one) : "v"(base));
      const int ticket = atomicAdd(a.tab.work + 1, one);
      if (ticket == nimg - 1) {
        const int raw = atomicExch(a.tab.work, 0);
        atomicExch(a.tab.work + 1, 0);
        a.tab.total[0] = raw < a.max_rois ? raw : a.max_rois;
        a.tab.total[1] = raw;
      }
    }
  }
  base = __shfl(base, 0);
  if (ok) {
    lp_det d;
    d.x1 = bj.x1; d.y1 = bj.y1; d.x2 = bj.x2; d.y2 = bj.y2; d.det_conf = bj.score; d.det_class = bj.cls;
    d.cls_class = -1; d.cls_conf = 0.f;
    a.dets[(long)n * a.max_det + o] = d;
    int* rects = a.rects + ((long)n * a.max_det + o) * 4;
    rects[0] = x1; rects[1] = y1; rects[2] = x2; rects[3] = y2;
    if (a.tab.total && base + o < a.max_rois) { a.tab.img[base + o] = n; a.tab.slot[base + o] = o; }
  }
}

__global__ __launch_bounds__(NMS_THREADS) void nms_kernel(const NmsArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __shared__ int s_wave[NMS_THREADS / 64 + 1];
  __shared__ float s_chunk[64 * 6];  // kept boxes of the current chunk: x1, y1, x2, y2, area, class bits
  __shared__ int s_nk, s_base;
  __shared__ double s_red[NMS_THREADS / 64];
  const int n = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int cnt = a.cand_count[n];
  cnt = cnt > a.A ? a.A : cnt;
  if (cnt <= 64 && cnt <= a.max_det && !a.no_small) {   // block-uniform: the other 15 waves leave before any barrier
    if (wave == 0) nms_small(a, n, cnt, (int)gridDim.x);
    return;
  }
  int npad = 1;
  while (npad < a.A) npad <<= 1;
  unsigned long long* keys = reinterpret_cast<unsigned long long*>(smem);
  unsigned char* removed = reinterpret_cast<unsigned char*>(smem + (size_t)npad * 8);
  const Cand* cand = a.cand + (long)n * a.A;
  Cand* sorted = a.sorted + (long)n * a.A;
  int nsort = 1;
  while (nsort < cnt) nsort <<= 1;

  for (int i = tid; i < nsort; i += NMS_THREADS) keys[i] = i < cnt ? nms_key(cand[i]) : 0ull;
  __syncthreads();
  for (int k = 2; k <= nsort; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = tid; i < nsort; i += NMS_THREADS) {
        const int ixj = i ^ j;
        if (ixj > i) {
          const unsigned long long x = keys[i], y = keys[ixj];
          const bool desc = (i & k) == 0;
          if (desc ? (x < y) : (x > y)) { keys[i] = y; keys[ixj] = x; }
        }
      }
      __syncthreads();
    }
  }
  // keys are sorted; rebuild the records in sorted order.  The slot of a key inside cand[]
  // is not stored in it, so each candidate finds its own rank by binary search.
  for (int i = tid; i < cnt; i += NMS_THREADS) {
    const Cand c = cand[i];
    const unsigned long long k = nms_key(c);
    int lo = 0, hi = cnt - 1;
    while (lo < hi) {  // descending order
      const int mid = (lo + hi) >> 1;
      if (keys[mid] > k) lo = mid + 1; else hi = mid;
    }
    sorted[lo] = c;
    removed[i] = 0;
  }
  __syncthreads();
  __threadfence_block();

  // ---- greedy sweep, 64 boxes per step
  int* keep = reinterpret_cast<int*>(keys);  // keys are dead from here on
  int nkeep = 0;
  const float thr = a.iou;
  const bool single_class = a.nc <= 1;
  for (int c0 = 0; c0 < cnt; c0 += 64) {
    if (wave == 0) {
      const int j = c0 + lane;
      const bool valid = j < cnt;
      Cand bj;
      bj.x1 = bj.y1 = bj.x2 = bj.y2 = 0.f; bj.cls = -1;
      if (valid) bj = sorted[j];
      const float aj = __fmul_rn(__fsub_rn(bj.x2, bj.x1), __fsub_rn(bj.y2, bj.y1));
      unsigned long long alive = __ballot(valid && !removed[valid ? j : 0]);
      unsigned long long todo = alive, kept = 0ull;
      int room = single_class ? a.max_det - nkeep : 0x7fffffff;
      while (todo && room > 0) {  // wave-uniform
        const int i = __ffsll((long long)todo) - 1;
        todo &= todo - 1;
        kept |= 1ull << i;
        --room;
        const float ix1 = __shfl(bj.x1, i), iy1 = __shfl(bj.y1, i), ix2 = __shfl(bj.x2, i), iy2 = __shfl(bj.y2, i);
        const float ai = __shfl(aj, i);
        const int ic = __shfl(bj.cls, i);
        const bool sup = lane > i && ((alive >> lane) & 1ull) && bj.cls == ic &&
                         nms_suppressed(ix1, iy1, ix2, iy2, ai, bj.x1, bj.y1, bj.x2, bj.y2, thr);
        const unsigned long long m = __ballot(sup);
        alive &= ~m;
        todo &= ~m;
      }
      if ((kept >> lane) & 1ull) {
        const int idx = __popcll(kept & ((1ull << lane) - 1ull));
        keep[nkeep + idx] = j;
        float* cb = s_chunk + idx * 6;
        cb[0] = bj.x1; cb[1] = bj.y1; cb[2] = bj.x2; cb[3] = bj.y2; cb[4] = aj; cb[5] = __int_as_float(bj.cls);
      }
      if (lane == 0) s_nk = __popcll(kept);
    }
    __syncthreads();
    const int nk = s_nk;
    nkeep += nk;
    // single class: (class, score) order is score order, so the first max_det kept boxes are the answer
    if (single_class && nkeep >= a.max_det) break;
    for (int j = c0 + 64 + tid; j < cnt; j += NMS_THREADS) {
      if (removed[j]) continue;
      const Cand bj = sorted[j];
      for (int k = 0; k < nk; ++k) {
        const float* cb = s_chunk + k * 6;
        if (__float_as_int(cb[5]) != bj.cls) continue;
        if (nms_suppressed(cb[0], cb[1], cb[2], cb[3], cb[4], bj.x1, bj.y1, bj.x2, bj.y2, thr)) { removed[j] = 1; break; }
      }
    }
    __syncthreads();
  }
  __syncthreads();

  // ---- more survivors than max_det (several classes): the max_det best scores stay.  The (score, anchor) key is
  //      unique inside an image, so the threshold key T with exactly max_det keys >= T exists; found bit by bit.
  unsigned long long T = 0ull;
  int nsel = nkeep;
  if (nkeep > a.max_det) {
    if (single_class) {
      nsel = a.max_det;  // head of the list
    } else {
      for (int bit = 45; bit >= 0; --bit) {
        const unsigned long long candT = T | (1ull << bit);
        int c = 0;
        for (int k = tid; k < nkeep; k += NMS_THREADS) c += nms_score_key(sorted[keep[k]]) >= candT ? 1 : 0;
        int tot;
        (void)block_exclusive_scan(c, s_wave, &tot);
        if (tot >= a.max_det) T = candT;
      }
      nsel = a.max_det;
    }
  }
  const bool by_key = nkeep > a.max_det && !single_class;
  const int nlist = by_key ? nkeep : nsel;

  // ---- ROI rectangle + area filter, order-preserving compaction
  const ImgGeom gm = a.geom[n];
  const int per = (nlist + NMS_THREADS - 1) / NMS_THREADS;
  const int k0 = tid * per, k1 = (k0 + per < nlist) ? k0 + per : nlist;
  int nvalid = 0;
  double ssum = 0.0;
  for (int k = k0; k < k1; ++k) {
    const Cand c = sorted[keep[k]];
    if (by_key && nms_score_key(c) < T) continue;
    ssum += (double)c.score;
    int x1, y1, x2, y2;
    nvalid += roi_rect(c, gm, a.roi_rule, a.min_area, x1, y1, x2, y2) ? 1 : 0;
  }
  int run;
  int o = block_exclusive_scan(nvalid, s_wave, &run);
  // mean detector confidence over ALL kept boxes, before the area filter (PipelineMetrics.det_confidence_avg, e2e.py:456-457)
  for (int off = 32; off > 0; off >>= 1) ssum += __shfl_xor(ssum, off);
  if (lane == 0) s_red[wave] = ssum;
  __syncthreads();
  if (tid == 0) {
    double tot = 0.0;
    for (int w = 0; w < NMS_THREADS / 64; ++w) tot += s_red[w];
    a.counts[n] = run;
    a.counts[gridDim.x + n] = nsel;
    reinterpret_cast<float*>(a.counts)[2 * gridDim.x + n] = nsel > 0 ? (float)(tot / (double)nsel) : 0.f;
    a.cand_count[n] = 0;  // ready for the next call
    // the image's slice of the batch-wide ROI list: work[0] accumulates, the block that draws the last ticket
    // (work[1]) publishes the clamped total for the classifier kernels and re-arms both words for the next call.
    // The ticket is issued only after the slice's add has RETURNED, so the last ticket implies every add is done.
    int base = 0;
    if (a.tab.total) {
      base = atomicAdd(a.tab.work, run);
      int one = 1;
      asm volatile("" : "+v"(one) : "v"(base));
      const int ticket = atomicAdd(a.tab.work + 1, one);
      if (ticket == (int)gridDim.x - 1) {
        const int raw = atomicExch(a.tab.work, 0);
        atomicExch(a.tab.work + 1, 0);
        a.tab.total[0] = raw < a.max_rois ? raw : a.max_rois;
        a.tab.total[1] = raw;  // unclamped: the host checks it against the capacity
      }
    }
    s_base = base;
  }
  __syncthreads();
  const int base = s_base;
  lp_det* dets = a.dets + (long)n * a.max_det;
  int* rects = a.rects + (long)n * a.max_det * 4;
  for (int k = k0; k < k1; ++k) {
    const Cand c = sorted[keep[k]];
    if (by_key && nms_score_key(c) < T) continue;
    int x1, y1, x2, y2;
    if (!roi_rect(c, gm, a.roi_rule, a.min_area, x1, y1, x2, y2)) continue;
    lp_det d;
    d.x1 = c.x1; d.y1 = c.y1; d.x2 = c.x2; d.y2 = c.y2; d.det_conf = c.score; d.det_class = c.cls;
    d.cls_class = -1; d.cls_conf = 0.f;
    dets[o] = d;
    rects[o * 4 + 0] = x1; rects[o * 4 + 1] = y1; rects[o * 4 + 2] = x2; rects[o * 4 + 3] = y2;
    if (a.tab.total && base + o < a.max_rois) { a.tab.img[base + o] = n; a.tab.slot[base + o] = o; }
    ++o;
  }
}

void launch_nms(const NmsArgs& a, int N, hipStream_t st) {
  set_max_dynamic_lds(reinterpret_cast<const void*>(nms_kernel), 150 * 1024);
  const size_t lds = nms_lds_bytes(a.A);
  LP_CHECK(lds <= 150 * 1024, LP_ERR_STATE, "NMS: %d anchors exceed the LDS sort capacity", a.A);
  LP_CHECK(a.A <= 16384, LP_ERR_STATE, "NMS: anchor index needs more than 14 key bits");
  const bool no_small = getenv("LITEPI_NMS_NO_SMALL") != nullptr;   // A/B switch + the path-equivalence test (read per enqueue: a captured graph keeps what it saw)
  NmsArgs b = a;
  b.no_small = no_small ? 1 : a.no_small;
  LP_LAUNCH(nms_kernel, dim3(N), dim3(NMS_THREADS), lds, st, b);
  LP_HIP(hipGetLastError());
}

// ------------------------------------------------------------------------------------
// PIL Image.resize((S,S), BILINEAR) of one ROI per workgroup (Pillow libImaging Resample.c):
// triangle filter scaled by the down-sampling factor, coefficients normalised in double and
// quantised to 22-bit fixed point, horizontal pass to uint8 then vertical pass to uint8.
// Pinned against Pillow itself through oracle/pil_resize_ref.py.
// ------------------------------------------------------------------------------------
#define RR_THREADS 1024  /* 16 waves: the byte gathers are latency-bound, TLP hides them */
#define RR_PRECISION_BITS 22
// "banded" variants (sides <= 384 px: support <= 6 -> at most 13 taps): the crop is copied into LDS in bands of
// rows with coalesced dword loads (the horizontal pass gathers 3..13 bytes per output sample: from global memory
// that is one texture-path request per byte and was the whole cost of this kernel), each band is resampled
// horizontally into the uint8 intermediate image [in_h][S][3] (Pillow's intermediate), then the vertical pass runs.
//   tiny : in_h <= 128 (the usual traffic-sign crop): 72 KB of LDS per workgroup, two ROIs per CU
//   small: in_h <= 384: 121 KB
// "large" variant: sides up to 4096 px (129 taps), one output row at a time, taps gathered from global memory.
#define RR_TINY_H 128
#define RR_SMALL_SIDE 384
#define RR_SMALL_K 13
#define RR_BAND_BYTES (40 * 1024)
#define RR_LARGE_K 129
enum { RR_MODE_TINY = 0, RR_MODE_SMALL = 1, RR_MODE_LARGE = 2 };

size_t roi_resize_lds_bytes() { return (size_t)2 * 64 * RR_LARGE_K * 4 + 4 * 64 * 4 + (size_t)RR_LARGE_K * 64 * 3 + 64; }
static size_t roi_resize_banded_lds(int rows) { return (size_t)2 * 64 * RR_SMALL_K * 4 + 4 * 64 * 4 + (size_t)rows * 64 * 3 + RR_BAND_BYTES + 64; }

// row pitch of the crop's LDS copy: the row's bytes keep their global address modulo 4 (aligned dword copies)
__device__ __forceinline__ int rr_band_pitch(int in_w) { return (in_w * 3 + 3 + 3) & ~3; }
__device__ __forceinline__ int rr_mode(int in_w, int in_h) {
  if (in_w <= RR_SMALL_SIDE && in_h <= RR_TINY_H) return RR_MODE_TINY;
  if (in_w <= RR_SMALL_SIDE && in_h <= RR_SMALL_SIDE) return RR_MODE_SMALL;
  return RR_MODE_LARGE;
}

// tap x of output index xx (Pillow's precompute_coeffs + normalize_coeffs_8bpc, one tap per thread); the x == 0 thread also
// writes the window bounds {xmin, n}
__device__ void pil_coeff_tap(int in_size, int out_size, int xx, int x, int* k, int* bounds) {
  const double scale = (double)in_size / (double)out_size;
  const double filterscale = scale < 1.0 ? 1.0 : scale;
  const double support = 1.0 * filterscale;
  const double ss = 1.0 / filterscale;
  const double center = (xx + 0.5) * scale;
  int xmin = (int)(center - support + 0.5);
  if (xmin < 0) xmin = 0;
  int xmax = (int)(center + support + 0.5);
  if (xmax > in_size) xmax = in_size;
  xmax -= xmin;
  if (x == 0) { bounds[0] = xmin; bounds[1] = xmax; }
  if (x >= xmax) return;
  double ww = 0.0;
  for (int t = 0; t < xmax; ++t) {
    double v = (t + xmin - center + 0.5) * ss;
    v = v < 0 ? -v : v;
    ww += v < 1.0 ? 1.0 - v : 0.0;
  }
  double v = (x + xmin - center + 0.5) * ss;
  v = v < 0 ? -v : v;
  double w = v < 1.0 ? 1.0 - v : 0.0;
  if (ww != 0.0) w /= ww;
  k[x] = w < 0 ? (int)(-0.5 + w * (double)(1 << RR_PRECISION_BITS)) : (int)(0.5 + w * (double)(1 << RR_PRECISION_BITS));
}

__device__ __forceinline__ uint8_t clip8(int v) {
  v >>= RR_PRECISION_BITS;
  return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

// one ROI per call, whole workgroup; MODE picks the staging strategy (block-uniform)
template <int MAXK, int MODE>
__device__ __forceinline__ void roi_resize_one(const RoiResizeArgs& a, int r, char* smem) {
  const int S = a.S;
  int* kx = reinterpret_cast<int*>(smem);
  int* ky = kx + 64 * MAXK;
  int* bx = ky + 64 * MAXK;  // [S][2]
  int* by = bx + 2 * 64;
  uint8_t* tmp = reinterpret_cast<uint8_t*>(by + 2 * 64);  // [rows][S][3]
  const int tid = threadIdx.x;
  const int half = 1 << (RR_PRECISION_BITS - 1);
  {
    const int img = a.tab.img[r], slot = a.tab.slot[r];
    const ImgGeom gm = a.geom[img];
    const int* rc = a.rects + ((long)img * a.max_det + slot) * 4;
    const int rx = rc[0], ry = rc[1];
    const int in_w = rc[2] - rx, in_h = rc[3] - ry;
    const uint8_t* src = a.src + gm.src_off;
    uint8_t* out = a.out + (long)r * S * S * 3;
    if (in_w > 4096 || in_h > 4096) {  // beyond the tap budget: defined (zero) output instead of garbage
      for (int i = tid; i < S * S * 3; i += RR_THREADS) out[i] = 0;
      return;
    }
    __syncthreads();
    // one thread per (axis, output index, tap): Pillow's double arithmetic in Pillow's order (the weight sum is
    // re-done by every tap's thread), so the table costs one division of latency instead of a serial loop
    for (int i = tid; i < 2 * 64 * MAXK; i += RR_THREADS) {
      const int axis = i / (64 * MAXK), rem = i - axis * (64 * MAXK);
      const int xx = rem / MAXK, x = rem - xx * MAXK;
      if (xx < S) pil_coeff_tap(axis ? in_h : in_w, S, xx, x, (axis ? ky : kx) + xx * MAXK, (axis ? by : bx) + 2 * xx);
    }
    __syncthreads();
    if (MODE != RR_MODE_LARGE) {
      constexpr int TMP_ROWS = MODE == RR_MODE_TINY ? RR_TINY_H : RR_SMALL_SIDE;
      uint8_t* crop = tmp + TMP_ROWS * 64 * 3;
      const int pitch = rr_band_pitch(in_w);
      const uint8_t* img_end = src + (long)gm.h * gm.w * 3;
      const int dpr = pitch >> 2;  // dwords per row
      const int rpb = RR_BAND_BYTES / pitch;  // rows per band (>= 35)
      for (int r0 = 0; r0 < in_h; r0 += rpb) {
        const int nb = in_h - r0 < rpb ? in_h - r0 : rpb;
        // 1. band -> LDS, aligned dwords.  Eight loads per thread are requested before the first store (one workgroup per CU:
        //    nothing else hides a load, and a loop of load-then-store is one memory round trip per iteration)
        for (int i0 = 0; i0 < nb * dpr; i0 += RR_THREADS * 8) {
          uint32_t v[8];
          const uint8_t* apv[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            int i = i0 + u * RR_THREADS + tid;
            i = i < nb * dpr ? i : nb * dpr - 1;
            const int row = i / dpr, j = i - row * dpr;
            const uint8_t* rp = src + ((long)(ry + r0 + row) * gm.w + rx) * 3;
            apv[u] = reinterpret_cast<const uint8_t*>(reinterpret_cast<uintptr_t>(rp) & ~(uintptr_t)3) + 4 * j;
            const bool whole = apv[u] >= src && apv[u] + 4 <= img_end;
            // (the image's first / last dword may straddle the caller's buffer: those are re-read byte-wise below)
            v[u] = *reinterpret_cast<const uint32_t*>(whole ? apv[u] : reinterpret_cast<const uint8_t*>(a.rects));   // (any valid dword)
          }
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            const int i = i0 + u * RR_THREADS + tid;
            if (i < nb * dpr) {
              const int row = i / dpr, j = i - row * dpr;
              uint32_t x = v[u];
              const uint8_t* ap = apv[u];
              if (!(ap >= src && ap + 4 <= img_end)) {  // never read outside the caller's buffer
                x = 0;
                for (int b = 0; b < 4; ++b)
                  if (ap + b >= src && ap + b < img_end) x |= (uint32_t)ap[b] << (8 * b);
              }
              *reinterpret_cast<uint32_t*>(crop + row * pitch + 4 * j) = x;
            }
          }
        }
        __syncthreads();
        // 2. horizontal pass of the band: one thread per (row, xx), three channels, BGR -> RGB
        for (int i = tid; i < nb * S; i += RR_THREADS) {
          const int row = S == 64 ? i >> 6 : i / S, xx = i - row * S;
          const int xmin = bx[2 * xx], nx = bx[2 * xx + 1];
          const uint8_t* rp = src + ((long)(ry + r0 + row) * gm.w + rx) * 3;
          const uint8_t* p = crop + row * pitch + (int)(reinterpret_cast<uintptr_t>(rp) & 3) + xmin * 3;
          uint8_t* o = tmp + ((r0 + row) * S + xx) * 3;
          if (in_w == S) {  // Pillow skips the pass when the width already matches (identity either way)
            o[0] = p[2]; o[1] = p[1]; o[2] = p[0];
          } else {
            const int* k = kx + xx * MAXK;
            int a0 = half, a1 = half, a2 = half;
            for (int x = 0; x < nx; ++x) {
              const int w = k[x];
              a0 += (int)p[x * 3 + 2] * w;
              a1 += (int)p[x * 3 + 1] * w;
              a2 += (int)p[x * 3 + 0] * w;
            }
            o[0] = clip8(a0); o[1] = clip8(a1); o[2] = clip8(a2);
          }
        }
        __syncthreads();
      }
      // 3. vertical pass: a thread keeps its column (j = byte of the 3S-byte output row) and strides over rows
      const int rowb = S * 3, lanes_rows = RR_THREADS / rowb;
      const int yo = tid / rowb, j = tid - yo * rowb;
      if (yo < lanes_rows) {
        for (int yy = yo; yy < S; yy += lanes_rows) {
          const int ymin = by[2 * yy], ny = by[2 * yy + 1];
          uint8_t v;
          if (in_h == S) {
            v = tmp[ymin * rowb + j];
          } else {
            const int* k = ky + yy * MAXK;
            int acc = half;
            for (int y = 0; y < ny; ++y) acc += (int)tmp[(ymin + y) * rowb + j] * k[y];
            v = clip8(acc);
          }
          out[yy * rowb + j] = v;
        }
      }
    } else {
      for (int yy = 0; yy < S; ++yy) {
        const int ymin = by[2 * yy], ny = by[2 * yy + 1];
        for (int i = tid; i < ny * S * 3; i += RR_THREADS) {
          const int c = i % 3, xx = (i / 3) % S, row = i / (3 * S);
          const int xmin = bx[2 * xx], nx = bx[2 * xx + 1];
          const uint8_t* p = src + ((long)(ry + ymin + row) * gm.w + rx + xmin) * 3 + (2 - c);
          uint8_t v;
          if (in_w == S) {
            v = p[0];
          } else {
            const int* k = kx + xx * MAXK;
            int acc = half;
            for (int x = 0; x < nx; ++x) acc += (int)p[x * 3] * k[x];
            v = clip8(acc);
          }
          tmp[i] = v;
        }
        __syncthreads();
        for (int i = tid; i < S * 3; i += RR_THREADS) {
          uint8_t v;
          if (in_h == S) {
            v = tmp[i];
          } else {
            const int* k = ky + yy * MAXK;
            int acc = half;
            for (int y = 0; y < ny; ++y) acc += (int)tmp[y * S * 3 + i] * k[y];
            v = clip8(acc);
          }
          out[yy * S * 3 + i] = v;
        }
        __syncthreads();
      }
    }
  }
}

// cv2.resize(roi_rgb, (S, S), interpolation=cv2.INTER_LINEAR) (e2e_optimize.py:386-390): no antialiasing, two source rows
// and columns per output pixel whatever the ROI's size; BGR -> RGB on the way.  The 2 x S coefficient triples go through
// LDS, the pixels are gathered from global memory (4 taps x 3 bytes per output pixel).
__device__ __forceinline__ void roi_resize_linear(const RoiResizeArgs& a, int r, char* smem) {
  const int S = a.S;
  int* cx = reinterpret_cast<int*>(smem);   // [S][3]: source index, a0, a1
  int* cy = cx + 3 * 64;
  const int tid = threadIdx.x;
  const int img = a.tab.img[r], slot = a.tab.slot[r];
  const ImgGeom gm = a.geom[img];
  const int* rc = a.rects + ((long)img * a.max_det + slot) * 4;
  const int rx = rc[0], ry = rc[1];
  const int in_w = rc[2] - rx, in_h = rc[3] - ry;
  const uint8_t* src = a.src + gm.src_off;
  uint8_t* out = a.out + (long)r * S * S * 3;
  __syncthreads();
  if (tid < 2 * S) {
    const int axis = tid >= S, d = tid - axis * S;
    int s0, a0, a1;
    lin_coeff(d, S, axis ? in_h : in_w, s0, a0, a1);
    int* c = (axis ? cy : cx) + 3 * d;
    c[0] = s0; c[1] = a0; c[2] = a1;
  }
  __syncthreads();
  for (int i = tid; i < S * S; i += RR_THREADS) {
    const int oy = i / S, ox = i - oy * S;
    const int sx = cx[3 * ox], ax0 = cx[3 * ox + 1], ax1 = cx[3 * ox + 2];
    const int sy = cy[3 * oy], ay0 = cy[3 * oy + 1], ay1 = cy[3 * oy + 2];
    const int sx1 = sx + 1 < in_w ? sx + 1 : in_w - 1;
    const int sy1 = sy + 1 < in_h ? sy + 1 : in_h - 1;
    const uint8_t* r0 = src + ((long)(ry + sy) * gm.w + rx) * 3;
    const uint8_t* r1 = src + ((long)(ry + sy1) * gm.w + rx) * 3;
    uint8_t* o = out + i * 3;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      int v;
      if (in_w == S && in_h == S) {
        v = r0[sx * 3 + c];   // cv2.resize returns a copy when the size already matches
      } else {
        const int h0 = r0[sx * 3 + c] * ax0 + r0[sx1 * 3 + c] * ax1;
        const int h1 = r1[sx * 3 + c] * ax0 + r1[sx1 * 3 + c] * ax1;
        v = (((ay0 * (h0 >> 4)) >> 16) + ((ay1 * (h1 >> 4)) >> 16) + 2) >> 2;
        v = v < 0 ? 0 : (v > 255 ? 255 : v);
      }
      o[2 - c] = (uint8_t)v;   // BGR -> RGB
    }
  }
}

// One launch for every ROI of the batch: a workgroup takes ROIs round-robin and picks the variant by the ROI's size
// (three launches, two of them usually empty, cost more than the resampling of a typical batch).
__global__ __launch_bounds__(RR_THREADS) void roi_resize_kernel(const RoiResizeArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int R = a.tab.total[0];
  for (int r = blockIdx.x; r < R; r += gridDim.x) {
    const int img = a.tab.img[r], slot = a.tab.slot[r];
    const int* rc = a.rects + ((long)img * a.max_det + slot) * 4;
    const int mode = rr_mode(rc[2] - rc[0], rc[3] - rc[1]);
    if (a.linear) roi_resize_linear(a, r, smem);
    else if (mode == RR_MODE_TINY) roi_resize_one<RR_SMALL_K, RR_MODE_TINY>(a, r, smem);
    else if (mode == RR_MODE_SMALL) roi_resize_one<RR_SMALL_K, RR_MODE_SMALL>(a, r, smem);
    else roi_resize_one<RR_LARGE_K, RR_MODE_LARGE>(a, r, smem);
    __syncthreads();
  }
}

void launch_roi_resize(const RoiResizeArgs& a, int max_items, hipStream_t st) {
  const size_t lds = roi_resize_banded_lds(RR_SMALL_SIDE) > roi_resize_lds_bytes() ? roi_resize_banded_lds(RR_SMALL_SIDE) : roi_resize_lds_bytes();
  set_max_dynamic_lds(reinterpret_cast<const void*>(roi_resize_kernel), 128 * 1024);
  LP_CHECK(a.S == 64, LP_ERR_ARG, "the ROI resize kernel's tap tables are sized for a 64x64 classifier input (e2e.py:367 resizes to 64x64 whatever --cls_input_size says)");
  LP_CHECK(lds <= 128 * 1024, LP_ERR_STATE, "ROI resize LDS budget");
  int grid = max_items < 512 ? max_items : 512;
  if (grid < 1) grid = 1;
  LP_LAUNCH(roi_resize_kernel, dim3(grid), dim3(RR_THREADS), lds, st, a);
  LP_HIP(hipGetLastError());
}

}  // namespace lp
