// Fused Detect head of one pyramid level on gfx950 (fp16 storage, fp32 accumulate): ONE launch runs
//     box tower  : 3x3 (Cin -> 64) + SiLU -> 3x3 (64 -> 64) + SiLU -> 1x1 (64 -> 4 x 16 DFL logits)
//     class tower: 3x3 (Cin -> c3) + SiLU -> 3x3 (c3 -> c3) + SiLU -> 1x1 (c3 -> nc)
//     decode     : softmax over the 16 bins of each box side, expectation, dist2bbox, x stride, class sigmoid,
//                  conf filter / xywh->xyxy / un-letterbox / clip -> candidates (+ out0 when requested)
// i.e. the reference graph's Detect module for one level (model.ncnn.param:151-182) plus its decode tail (:184-208)
// and the filter half of NCNNDetector.postprocess (e2e.py:255-278).  This is 51.5 % of the detector's FLOPs; the
// layer-at-a-time plan wrote and re-read the 96-channel first-conv output and the 65-channel projections through HBM
// and needed four launches per level.
//
// One workgroup (4 waves, one per SIMD) owns a TH x TW tile of anchors of one image:
//   stage A  both first 3x3 convs as one GEMM (64 + 32*C3T output channels) over the (TH+2) x (TW+2) region stage B
//            needs; result (zero outside the image = stage B's padding) -> LDS image MID, fp16
//   stage B  the second 3x3 convs from MID; accumulators stay in registers
//   stage C  the 1x1 projections straight from the stage-B accumulators: a 32x32 accumulator tile (channel rows in the
//            registers, pixel on the lane) IS the B operand of the next MFMA once its rows are rounded to fp16, with the
//            projection's K order permuted to match (MI355X guide, "an accumulator tile as the next MFMA's operand")
//   decode   in registers: the projection rows are permuted so that a lane holds the 16 bins of one box side
// MFMA: v_mfma_f32_32x32x16_f16, D[out-channel][pixel] = W . X.  A wave computes RT (or 2) row tiles x P pixel tiles per
// K step: (RT + P) KiB of LDS operands per RT*P MFMAs of 32 cycles = 0.8 - 2 ds_read_b128 per MFMA (two per MFMA and SIMD
// saturate the LDS); measured, the K loops run at 37-44 cycles per MFMA and are half of a workgroup's cycles -- the rest is
// the prologue (tile + first weight chunks) and the SiLU epilogues / decode on the VALU (tools/head_stamps.py).
// Weights: every workgroup consumes the same 150-370 fragments (1 KiB each) in the same order, so the host packs them into
// one stream and the kernel moves it through a two-slot LDS ring THROUGH REGISTERS (a global load + ds_write pair per piece
// behind the MFMAs of a K step, one chunk ahead; the first version used global_load_lds and paid 100+ issue cycles per
// piece); the input tile is staged through registers too (all rows requested before the first store).
#include "head.h"
#include "post_dev.h"

#include <algorithm>
#include <cstdlib>

namespace lp {

typedef _Float16 half_t;
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));


// diagnostic phase stamps (never enabled on the product path: a.stamps is null)
#define HD_STAMP(k)                                                                                    \
  if (a.stamps && threadIdx.x == 0) a.stamps[(size_t)blockIdx.x * 16 + (k)] = ((k) == 0 || (k) == 15) ? wall_clock64() : clock64();

typedef float floatx2 __attribute__((ext_vector_type(2)));
// x * sigmoid(x) for two values: the multiplies and the add are two-wide (v_pk_*_f32), the exponential and the reciprocal
// are the hardware transcendentals (1 ulp), as Tr<half>::silu in conv_kernels.hip
__device__ __forceinline__ floatx2 hd_silu2(floatx2 v) {
  float nl2e = -1.4426950408889634f;   // in an SGPR, or the packed multiply becomes two v_mul_f32 with a literal (act4, conv_kernels.hip)
  asm("" : "+s"(nl2e));
  const floatx2 t = v * floatx2{nl2e, nl2e};
  floatx2 e;
  e[0] = __builtin_amdgcn_exp2f(t[0]);
  e[1] = __builtin_amdgcn_exp2f(t[1]);
  e = e + floatx2{1.f, 1.f};
  floatx2 rc;
  rc[0] = __builtin_amdgcn_rcpf(e[0]);
  rc[1] = __builtin_amdgcn_rcpf(e[1]);
  return v * rc;
}
// 8 accumulator values (bias already inside) -> SiLU -> 8 halfs
__device__ __forceinline__ half8 silu_h8(const floatx16& acc, int base) {
  half8 q;
#pragma unroll
  for (int j = 0; j < 8; j += 2) {
    const floatx2 y = hd_silu2(floatx2{acc[base + j], acc[base + j + 1]});
    q[j] = (half_t)y[0];
    q[j + 1] = (half_t)y[1];
  }
  return q;
}
__device__ __forceinline__ half8 lds_h8(const char* p) { return __builtin_bit_cast(half8, *reinterpret_cast<const u32x4*>(p)); }
__device__ __forceinline__ floatx16 mfma32(half8 a, half8 b, floatx16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ floatx4 mfma16(half8 a, half8 b, floatx4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
// accumulator tile initialised with the bias of this lane's 16 channels (bias + 0 ..15)
__device__ __forceinline__ floatx16 bias16(const float* __restrict__ b) {
  floatx16 v;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const floatx4 x = *reinterpret_cast<const floatx4*>(b + 4 * q);
    v[4 * q] = x[0]; v[4 * q + 1] = x[1]; v[4 * q + 2] = x[2]; v[4 * q + 3] = x[3];
  }
  return v;
}

template <int I, int N, typename F> __device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}

// all of this wave's LDS traffic done, then the workgroup barrier -- NOT __syncthreads(): that also waits for the global
// loads in flight (vmcnt(0)), i.e. for the ring pieces requested a few hundred cycles ago
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

// K loop of weight chunk c (ring slot c & 1): acc[rt][p] += W(rt, step s) . X(pixels of p, K offset next_boff()) for s < KS.
//  * The operand fragments of step s+1 are requested BEFORE the MFMAs of step s (two register sets, static indices after
//    unrolling): with one wave per SIMD nothing else covers the LDS latency -- the compiler's own schedule of the plain loop
//    ran at 53 cycles per MFMA, this form at 37 (tools/ubench/mfma_rate.hip), against 32 for the bare instruction.
//  * The pipeline runs ACROSS chunks (FIRST / LAST mark the ends of a run of chunks that share acc): the last step of a
//    chunk first passes the ring barrier -- every wave has stored its pieces of chunk c+1 by then (they ride on steps
//    0 .. KS-2) and has its last fragments of chunk c in registers, so the barrier publishes slot (c+1)&1 and frees slot c&1 at
//    once -- and requests step 0 of chunk c+1 before its own MFMAs.  A barrier at the chunk boundary instead drained the
//    pipeline: ~500 idle matrix-pipe cycles per chunk, 3-12 chunks per stage.  (KS odd: the register sets would swap roles
//    from chunk to chunk; such chunks run stand-alone, FIRST && LAST.)
//  * side(j), j < SLOTF / 4: the weight ring's store + reload of piece j (see the kernel).
//  * SLOTF: fragments (KiB) per ring slot, 24 or 12; a wave owns SLOTF / 4 pieces of every chunk.
//  * SLOTF == 8 (three workgroups per CU): a ring slot is 64 bytes short of 8 KiB -- the LDS granule is 1280 B and 42 granules
//    per workgroup are the limit -- and lanes 60..63 of fragment 7 live in four padding slots of MID instead: f7[parity] is
//    this lane's address of fragment 7, f7_0 / f7_1 by slot parity (only the box tower's chunks have eight fragments).
template <int NA, int NB, int KS, bool FIRST, bool LAST, int SLOTF, typename F, typename G>
__device__ __forceinline__ void kchunk(const char* ring, int c, int lane16, const char* img, const int (&pix)[NB], F&& next_boff, G&& side,
                                       floatx16 (&acc)[NA][NB], half8 (&af)[2][NA], half8 (&bf)[2][NB], const char* f7_0, const char* f7_1) {
  static_assert(KS % 2 == 0 || (FIRST && LAST), "an odd chunk cannot hand its register sets to the next one");
  static_assert(KS >= 2 && NA * KS <= SLOTF, "the ring pieces ride on steps 0 .. KS-2; a chunk fits one slot");
  constexpr int SLOT = SLOTF == 8 ? 8192 - 64 : SLOTF * 1024, NPW = SLOTF / 4;
  const char* wb = ring + (c & 1) * SLOT + lane16;
  const char* wbn = ring + ((c + 1) & 1) * SLOT + lane16;
  const char* f7b = (c & 1) ? f7_1 : f7_0;
  const char* f7n = (c & 1) ? f7_0 : f7_1;
  auto ld = [&](int set, const char* w, int s) {
    const int boff = next_boff();
#pragma unroll
    for (int rt = 0; rt < NA; ++rt) {
      if (SLOTF == 8 && s * NA + rt == 7) af[set][rt] = lds_h8(w == wb ? f7b : f7n);
      else af[set][rt] = lds_h8(w + (s * NA + rt) * 1024);
    }
#pragma unroll
    for (int p = 0; p < NB; ++p) bf[set][p] = lds_h8(img + pix[p] + boff);
  };
  constexpr int PPS = (NPW + KS - 2) / (KS - 1);   // ring pieces per step
  if (FIRST) {
    lds_barrier();
    ld(0, wb, 0);
  }
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    if (s + 1 < KS) {
      ld((s + 1) & 1, wb, s + 1);
    } else if (!LAST) {
      lds_barrier();
      ld(0, wbn, 0);
    }
    // pin the order: without this the scheduler sinks the reads below the MFMAs and folds the two register sets into one
    // (reads of step s+1 issued after the last MFMA of step s: ~100 idle matrix-pipe cycles per step, 50+ cycles per MFMA)
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int rt = 0; rt < NA; ++rt)
#pragma unroll
      for (int p = 0; p < NB; ++p) acc[rt][p] = mfma32(af[s & 1][rt], bf[s & 1][p], acc[rt][p]);
#pragma unroll
    for (int j = s * PPS; j < (s + 1) * PPS && j < NPW; ++j) side(j);
  }
}

// Stage A on v_mfma_f32_16x16x32_f16 (round 4): the (TH+2) x (TW+2) region costs ceil(pixels / 16) pixel tiles instead of
// 32-pixel slots (80^2: 12 x 16 = 192 slots for 180 pixels instead of 256; 40^2: 17 x 16 for 264 instead of 384; 20^2: 144 =
// 9 x 16 instead of 256), a quarter to almost a half fewer matrix cycles in the stage that holds 54 - 81 % of a level's MFMAs, and
// as many fewer SiLUs in its epilogue.  A wave owns ALL 2*RT row tiles (16 channels each) of NB pixel tiles.  A chunk is KS
// HALF-steps: half-step s multiplies the RT row tiles of row half s & 1 with the pixel fragments of K step s >> 1 (32 channels
// of one tap), so a K step's pixel fragments are read once and used by both halves, while its 2*RT weight fragments arrive
// RT at a time (6 KiB per K step at C3T = 1: what an 8 KiB ring slot holds).  Operand pipeline, ring barrier inside the last
// half-step and the ring's side stores as in kchunk; BSET0 = register set of the chunk's first pixel fragments (the sets
// alternate per K step, so with an odd number of K steps per chunk the parity alternates per chunk: chunks are unrolled).
template <int RT, int NB, int KS, bool FIRST, bool LAST, int BSET0, int SLOTF, typename F, typename G>
__device__ __forceinline__ void kchunkA16(const char* ring, int c, int lane16, const char* img, const int (&pix)[NB], F&& next_boff, G&& side,
                                          floatx4 (&acc)[2 * RT][NB], half8 (&af)[2][RT], half8 (&bf)[2][NB]) {
  static_assert(KS % 2 == 0 && RT * KS <= (SLOTF == 8 ? 7 : SLOTF), "whole K steps; a chunk fits one slot (fragment 7 of an 8-fragment slot lives elsewhere)");
  constexpr int SLOT = SLOTF == 8 ? 8192 - 64 : SLOTF * 1024, NPW = SLOTF / 4;
  const char* wb = ring + (c & 1) * SLOT + lane16;
  const char* wbn = ring + ((c + 1) & 1) * SLOT + lane16;
  auto ldA = [&](int set, const char* w, int s) {
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) af[set][rt] = lds_h8(w + (s * RT + rt) * 1024);
  };
  auto ldB = [&](int set) {
    const int boff = next_boff();
#pragma unroll
    for (int p = 0; p < NB; ++p) bf[set][p] = lds_h8(img + pix[p] + boff);
  };
  constexpr int PPS = (NPW + KS - 2) / (KS - 1);   // ring pieces per half-step (they ride on half-steps 0 .. KS-2)
  if (FIRST) {
    lds_barrier();
    ldA(0, wb, 0);
    ldB(BSET0);
  }
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    if (s + 1 < KS) {
      ldA((s + 1) & 1, wb, s + 1);
      if ((s + 1) % 2 == 0) ldB((BSET0 + (s + 1) / 2) & 1);
    } else if (!LAST) {
      lds_barrier();
      ldA(0, wbn, 0);
      ldB((BSET0 + KS / 2) & 1);
    }
    __builtin_amdgcn_sched_barrier(0);
    const int bs = (BSET0 + s / 2) & 1, rh = s & 1;
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int p = 0; p < NB; ++p) acc[rh * RT + rt][p] = mfma16(af[s & 1][rt], bf[bs][p], acc[rh * RT + rt][p]);
#pragma unroll
    for (int j = s * PPS; j < (s + 1) * PPS && j < NPW; ++j) side(j);
  }
}

// NPC: 64-slot pieces per input-tile row, KSA: K steps per stage-A chunk (both fixed by the level's tile shape, see host)
// SLOTF: fragments per weight-ring slot (24; 12 for the two-workgroups-per-CU shape), NRW: input-tile rows per wave = ceil((TH+4)/4)
// A16: stage A on 16x16x32 MFMAs (kchunkA16): PA = 16-pixel tiles per wave, KSA = half-steps per chunk (one tap per chunk)
// NCA: stage-A chunks of the A16 path = 9 taps x ceil(Cin / 32) K steps / (KSA / 2) K steps per chunk
template <int C3T, int PA, int PB, int NPC, int KSA, int SLOTF, int NRW, bool OV = (SLOTF == 8), bool A16 = false, int NCA = 9>
__global__ __launch_bounds__(256, (SLOTF == 8 ? 3 : (SLOTF == 12 ? 2 : 1))) void head_fused_kernel(const HeadArgs a) {
  // SLOTF == 8: THREE workgroups per CU -- MID overlays the input tile (dead after stage A's K loops: one more barrier), 8 KiB
  // ring slots (the projections are then two chunks), <= 168 registers: 54,272 B of LDS.
  constexpr int RT = 2 + C3T;       // row tiles (32 channels) of the merged first convs: box 2 | class C3T
  constexpr int SLOT = SLOTF * 1024, NPW = SLOTF / 4;   // chunk stride in the weight stream; 1 KiB pieces per wave and chunk
  constexpr int SLOTB = SLOTF == 8 ? 8192 - 64 : SLOT;   // ring slot stride in LDS (see kchunk)
  constexpr int SPM = 4 * RT + 1;   // 16-byte slots per MID pixel, one of them padding (odd: conflict-free pixel stride)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  const int n = blockIdx.x % a.N, tile = blockIdx.x / a.N;   // image fastest: an image's tiles meet in one XCD's L2
  const int ty0 = tile / a.tiles_x, tx0 = tile - ty0 * a.tiles_x;
  const int oy0 = ty0 * a.TH, ox0 = tx0 * a.TW;
  const int TH = a.TH, TW = a.TW, KPT = a.KPT;
  const int RWin = TW + 4, IHin = TH + 4, SPin = 2 * KPT + 1, RSin = RWin * SPin;
  const int RW1 = TW + 2, R1 = (TH + 2) * RW1, R2 = TH * TW;
  const int nA = A16 ? (R1 + 15) >> 4 : (R1 + 31) >> 5, nB = (R2 + 31) >> 5;
  constexpr bool OVL = OV;   // MID overlays the input tile
  char* IN = smem;
  const int in_al = (IHin * RSin * 16 + 1023) & ~1023, mid_al = (R1 * SPM * 16 + 1023) & ~1023;
  char* MID = OVL ? smem : smem + in_al;
  char* RING = OVL ? smem + (IHin * RSin > R1 * SPM ? IHin * RSin : R1 * SPM) * 16 : MID + mid_al;
  // fragment 7 of ring slot `par`, this lane: inside the slot, or (SLOTF == 8, lanes 60..63) the padding slot of MID pixel
  // 172 + 4 par + lane - 60 -- beyond the input tile, never written by the MID stores
  const char* f7_0 = (SLOTF == 8 && lane >= 60) ? MID + ((R1 - 8 + lane - 60) * SPM + 4 * RT) * 16 : RING + 7 * 1024 + lane * 16;
  const char* f7_1 = (SLOTF == 8 && lane >= 60) ? MID + ((R1 - 4 + lane - 60) * SPM + 4 * RT) * 16 : RING + SLOTB + 7 * 1024 + lane * 16;
  const char* wstream = reinterpret_cast<const char*>(a.wstream) + lane * 16;
  const int nch = a.nchunks;
  HD_STAMP(0)
  HD_STAMP(1)

  // ---- weight ring, two 24 KiB slots, filled THROUGH REGISTERS: a wave owns pieces wave, wave + 4, .. (six 1 KiB
  //      fragments) of every chunk.  Behind the MFMAs of K step j of chunk c it stores piece j of chunk c+1 -- loaded one
  //      chunk earlier -- into the slot everybody left at the last barrier, and requests piece j of chunk c+2 into the same
  //      registers; plain loads, so the compiler counts the waits, and every piece has one whole chunk of MFMAs to arrive.
  //      (The first version moved the stream by LDS-DMA: each global_load_lds piece cost 100+ issue cycles that the in-order
  //      wave could not hide behind its MFMAs; a global load + ds_write pair costs ~25.)  Piece indices past a chunk's end
  //      re-read its last fragment into the slot's unused tail: no control flow around the loads (see cls_net.hip for what
  //      that does to the register allocator).
  // A16: the first convs' biases are requested in front of the tile (they initialise the accumulators right behind the prologue;
  // requested there, their round trip stood between the prologue and the first MFMA)
  const float bias_c0 = a.biasC[64];   // class 0's projection bias (the only one a one-class detector needs; used in stage C)
  floatx4 biasA16[A16 ? 2 * RT : 1];
  if constexpr (A16) {
    const int kgb = (lane >> 4) * 8;
#pragma unroll
    for (int t = 0; t < RT; ++t) {
      biasA16[2 * t] = *reinterpret_cast<const floatx4*>(a.biasA + 32 * t + kgb);
      biasA16[2 * t + 1] = *reinterpret_cast<const floatx4*>(a.biasA + 32 * t + kgb + 4);
    }
  }
  u32x4 wreg[NPW];
  // (every chunk occupies a whole 24 KiB slot image in the stream, two zero chunks follow the last one: the source of
  //  piece j of chunk c is wstream + c * 24 KiB + (wave + 4j) KiB -- no chunk table, no clamping, no scalar loads in the K loop)
  const char* wsrc = wstream;
  auto wsource = [&](int c) { wsrc = wstream + (size_t)c * SLOT + wave * 1024; };
  auto wload1 = [&](int j) { wreg[j] = *reinterpret_cast<const u32x4*>(wsrc + j * 4096); };
  auto wstore1 = [&](int c, int j) {
    char* dst = RING + (c & 1) * SLOTB + lane * 16 + (wave + 4 * j) * 1024;
    if (SLOTF == 8 && j == 1 && wave == 3) dst = const_cast<char*>((c & 1) ? f7_1 : f7_0);   // piece 7
    *reinterpret_cast<u32x4*>(dst) = wreg[j];
  };
  // ---- input tile (halo 2) -> IN through registers: rows of RSin 16-byte slots = (pixel, channel group); pad slots, pixels
  //      outside the image and the row's tail are zeros (= the conv's zero padding).  A wave takes rows wave, wave + 4, ..;
  //      what depends on the lane only (pixel, channel group, validity of a slot) is computed once per 64-slot piece.
  {
    const unsigned rcp_sp = (65536u + SPin - 1) / SPin;  // exact for slot indices < 2048 (host-checked)
    int voff[NPC];
    bool xok[NPC];
#pragma unroll
    for (int pc = 0; pc < NPC; ++pc) {
      const int sl = pc * 64 + lane;
      const int ix = (int)(((unsigned)sl * rcp_sp) >> 16), cgs = sl - ix * SPin;
      const int gx = ox0 - 2 + ix;
      xok[pc] = sl < RSin && gx >= 0 && gx < a.W && ix < RWin && cgs < 2 * KPT;
      voff[pc] = xok[pc] ? (ix * a.in_pitch + cgs * 8) * 2 : (ox0 >= 2 ? 0 : (2 - ox0) * a.in_pitch * 2);
    }
    const char* in_n = reinterpret_cast<const char*>(a.in) + ((long)n * a.H * a.W) * a.in_pitch * 2;
    u32x4 v[NRW][NPC];
#pragma unroll
    for (int j = 0; j < NRW; ++j) {
      const int iy = wave + 4 * j;
      const int gy = oy0 - 2 + iy;
      const bool rok = iy < IHin && gy >= 0 && gy < a.H;
      const char* rowp = in_n + ((long)(rok ? gy : 0) * a.W + (ox0 - 2)) * a.in_pitch * 2;
#pragma unroll
      for (int pc = 0; pc < NPC; ++pc) {
        // unconditional load from a valid address (voff of an invalid slot points at the row's first in-image pixel), masked after
        const u32x4 x = *reinterpret_cast<const u32x4*>(rowp + voff[pc]);
        v[j][pc] = (rok && xok[pc]) ? x : u32x4{0u, 0u, 0u, 0u};
      }
    }
    u32x4 w0[NPW];   // chunk 0 goes through a register set of its own so that chunk 1 can be requested in the same breath
#pragma unroll
    for (int j = 0; j < NPW; ++j) w0[j] = *reinterpret_cast<const u32x4*>(wstream + (wave + 4 * j) * 1024);
    wsource(1);
#pragma unroll
    for (int j = 0; j < NPW; ++j) wload1(j);
#pragma unroll
    for (int j = 0; j < NRW; ++j) {
      const int iy = wave + 4 * j;
#pragma unroll
      for (int pc = 0; pc < NPC; ++pc) {
        const int sl = pc * 64 + lane;
        if (iy < IHin && sl < RSin) *reinterpret_cast<u32x4*>(IN + (iy * RSin + sl) * 16) = v[j][pc];
      }
    }
#pragma unroll
    for (int j = 0; j < NPW; ++j) {
      char* dst = RING + lane * 16 + (wave + 4 * j) * 1024;
      if (SLOTF == 8 && j == 1 && wave == 3) dst = const_cast<char*>(f7_0);
      *reinterpret_cast<u32x4*>(dst) = w0[j];
    }
    // A16 with an odd KPT: the slot behind the tile's last pixel is read (against zero weights) by the last K step of a tap
    if (A16 && (KPT & 1) && tid == 0) *reinterpret_cast<u32x4*>(IN + IHin * RSin * 16) = u32x4{0u, 0u, 0u, 0u};
  }
  HD_STAMP(2)
  if (a.flags & 1) __builtin_amdgcn_s_setprio(1);

  // ---- this lane's pixels
  int pixA[PA];   // byte offset of the lane's stage-A pixel (region-1 pixel (ry, rx) -> IN pixel (ry, rx)), + its K half
  // A16: lane = (column col, K group kg); column col of a 16-pixel tile holds pixel offset sig(col) -- even offsets for the lanes
  // {0-3, 12-15}, odd for {4-11} -- and K group kg reads channel slot gam(kg) = {0, 2, 1, 3} of the step's four: with the odd
  // pixel pitch every ds_read_b128 lane group then hits 16 distinct 16-byte slots (as c2f_kernels.hip; tiles that wrap a
  // region row still collide two-way).  The weights are packed to match (HeadLayer::build).
  const int col16 = lane & 15, kg16 = lane >> 4;
  const int sig16 = col16 < 4 ? 2 * col16 : (col16 < 12 ? 2 * (col16 - 4) + 1 : 2 * (col16 - 8));
#pragma unroll
  for (int p = 0; p < PA; ++p) {
    int idx = A16 ? 16 * (wave + 4 * p) + sig16 : 32 * (wave + 4 * p) + r;
    idx = idx < R1 ? idx : R1 - 1;
    const int ry = idx / RW1, rx = idx - ry * RW1;
    pixA[p] = (ry * RWin + rx) * SPin * 16 + (A16 ? ((kg16 & 1) * 2 + (kg16 >> 1)) * 16 : h * 16);
  }
  const int lane16 = lane * 16;
  int c = 0;
  auto ring_side = [&](int j) {
    wstore1(c + 1, j);
    wload1(j);
  };

  floatx16 biasB_box[2], biasB_cls[C3T];
  // ======================= stage A: [64 + 32*C3T] x (9 * Cin) x region-1 pixels =======================
  if constexpr (A16) {
    floatx4 accA[2 * RT][PA];
#pragma unroll
    for (int t = 0; t < RT; ++t) {   // row tiles 2t, 2t+1 hold channels 32t + 8 kg + {0..3}, {4..7} of this lane's pixel
#pragma unroll
      for (int p = 0; p < PA; ++p) { accA[2 * t][p] = biasA16[2 * t]; accA[2 * t + 1][p] = biasA16[2 * t + 1]; }
    }
    {
      // K step = (tap, 32-channel block): ceil(KPT / 2) steps per tap, the byte offset kept incrementally.  Cin = 16 (mod 32), v2's
      // 48-channel P3: the last step of a tap reads one pixel's padding slot and the NEXT pixel's first slot as its channels
      // Cin .. Cin + 15 -- their weights are zero (HeadLayer::build) and what is read is finite: activations, or the zeroed guard
      // slot behind the tile's last pixel
      int cg = 0, dx = 0, boff_run = 0;
      const int SPT = (KPT + 1) >> 1;
      const int d_tap = SPin * 16 - 64 * (SPT - 1), d_row = (RWin - 2) * SPin * 16 - 64 * (SPT - 1);
      auto next_boff = [&]() {
        const int boff = boff_run;
        if (++cg == SPT) {
          cg = 0;
          if (++dx == 3) { dx = 0; boff_run += d_row; } else boff_run += d_tap;
        } else {
          boff_run += 64;
        }
        return boff;
      };
      half8 af[2][RT], bf[2][PA];
      static_for<0, NCA>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        wsource(c + 2);
        kchunkA16<RT, PA, KSA, i == 0, i == NCA - 1, (i * (KSA / 2)) & 1, SLOTF>(RING, c, lane16, IN, pixA, next_boff, ring_side, accA, af, bf);
        if (i == 0) { HD_STAMP(3) }
        ++c;
      });
    }
    HD_STAMP(4)
    if (a.flags & 1) __builtin_amdgcn_s_setprio(0);
    if (a.flags & 2) __builtin_amdgcn_s_setprio(1);
    // (stage B's biases are requested here: their round trip runs under the SiLU epilogue instead of in front of stage B's first MFMA)
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) biasB_box[rt] = bias16(a.biasB + rt * 32 + h * 16);
#pragma unroll
    for (int rt = 0; rt < C3T; ++rt) biasB_cls[rt] = bias16(a.biasB + 64 + rt * 32 + h * 16);
    // ---- SiLU, fp16, -> MID: this lane holds channels 32t + 8 kg .. + 7 of its pixel in row tiles 2t | 2t+1: one 16-byte store each
    {
      if (OVL) lds_barrier();   // MID overlays the input tile: every wave has read its last stage-A operands
      const bool interior = oy0 >= 1 && ox0 >= 1 && oy0 + TH + 1 <= a.H && ox0 + TW + 1 <= a.W;  // block-uniform
#pragma unroll
      for (int p = 0; p < PA; ++p) {
        const int pt = wave + 4 * p;
        const int idx = 16 * pt + sig16;
        if (pt < nA && idx < R1) {
          bool inside = true;
          if (!interior) {
            const int ry = idx / RW1, rx = idx - ry * RW1;
            const int gy = oy0 - 1 + ry, gx = ox0 - 1 + rx;
            inside = gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
          }
          char* dst = MID + idx * SPM * 16 + kg16 * 16;
#pragma unroll
          for (int t = 0; t < RT; ++t) {
            const floatx2 y0 = hd_silu2(floatx2{accA[2 * t][p][0], accA[2 * t][p][1]}), y1 = hd_silu2(floatx2{accA[2 * t][p][2], accA[2 * t][p][3]});
            const floatx2 y2 = hd_silu2(floatx2{accA[2 * t + 1][p][0], accA[2 * t + 1][p][1]}), y3 = hd_silu2(floatx2{accA[2 * t + 1][p][2], accA[2 * t + 1][p][3]});
            half8 q = {(half_t)y0[0], (half_t)y0[1], (half_t)y1[0], (half_t)y1[1], (half_t)y2[0], (half_t)y2[1], (half_t)y3[0], (half_t)y3[1]};
            if (!inside) {
#pragma unroll
              for (int j = 0; j < 8; ++j) q[j] = (half_t)0.f;
            }
            *reinterpret_cast<half8*>(dst + t * 64) = q;
          }
        }
      }
    }
  } else {
    floatx16 accA[RT][PA];
  #pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      const floatx16 b = bias16(a.biasA + rt * 32 + h * 16);
  #pragma unroll
      for (int p = 0; p < PA; ++p) accA[rt][p] = b;
    }
    {
      // K offset of the running step = (tap row * RWin + tap column) pixels + channel-group pair, kept incrementally
      int cg = 0, dx = 0, boff_run = 0;
      const int d_tap = SPin * 16 - 32 * (KPT - 1), d_row = (RWin - 2) * SPin * 16 - 32 * (KPT - 1);
      auto next_boff = [&]() {
        const int boff = boff_run;
        if (++cg == KPT) {
          cg = 0;
          if (++dx == 3) { dx = 0; boff_run += d_row; } else boff_run += d_tap;
        } else {
          boff_run += 32;
        }
        return boff;
      };
      // stream = A chunks | box-B chunks | class-B chunks | 1 C
      const int ncA = nch - (36 / (SLOTF / 2)) - (C3T == 2 ? 36 / (SLOTF / 2) : 18 / (SLOTF >= 18 ? 18 : 6)) - (SLOTF == 8 ? 2 : 1);
      half8 af[2][RT], bf[2][PA];
      if constexpr (KSA % 2 == 0) {   // one operand pipeline over all of stage A (ncA >= 2, host-checked)
        wsource(c + 2);
        kchunk<RT, PA, KSA, true, false, SLOTF>(RING, c, lane16, IN, pixA, next_boff, ring_side, accA, af, bf, f7_0, f7_1);
        HD_STAMP(3)
        for (++c; c < ncA - 1; ++c) {
          wsource(c + 2);
          kchunk<RT, PA, KSA, false, false, SLOTF>(RING, c, lane16, IN, pixA, next_boff, ring_side, accA, af, bf, f7_0, f7_1);
        }
        wsource(c + 2);
        kchunk<RT, PA, KSA, false, true, SLOTF>(RING, c, lane16, IN, pixA, next_boff, ring_side, accA, af, bf, f7_0, f7_1);
        ++c;
      } else {
        for (; c < ncA; ++c) {
          wsource(c + 2);
          kchunk<RT, PA, KSA, true, true, SLOTF>(RING, c, lane16, IN, pixA, next_boff, ring_side, accA, af, bf, f7_0, f7_1);
          if (c == 0) { HD_STAMP(3) }
        }
      }
    }
    HD_STAMP(4)
    if (a.flags & 1) __builtin_amdgcn_s_setprio(0);
    if (a.flags & 2) __builtin_amdgcn_s_setprio(1);
    // ---- SiLU, fp16, -> MID (zero outside the image: stage B's padding).  The weight rows are permuted at pack time
    //      so that this lane holds channels 32*rt + 16*h .. +15 of its pixel: two 16-byte stores per row tile.
    {
      if (OVL) lds_barrier();   // MID overlays the input tile: every wave has read its last stage-A operands
      const bool interior = oy0 >= 1 && ox0 >= 1 && oy0 + TH + 1 <= a.H && ox0 + TW + 1 <= a.W;  // block-uniform
  #pragma unroll
      for (int p = 0; p < PA; ++p) {
        const int pt = wave + 4 * p;
        const int idx = 32 * pt + r;
        if (pt < nA && idx < R1) {
          bool inside = true;
          if (!interior) {
            const int ry = idx / RW1, rx = idx - ry * RW1;
            const int gy = oy0 - 1 + ry, gx = ox0 - 1 + rx;
            inside = gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
          }
          char* dst = MID + idx * SPM * 16 + h * 32;
  #pragma unroll
          for (int rt = 0; rt < RT; ++rt) {
            half8 q0 = silu_h8(accA[rt][p], 0), q1 = silu_h8(accA[rt][p], 8);
            if (!inside) {
  #pragma unroll
              for (int j = 0; j < 8; ++j) { q0[j] = (half_t)0.f; q1[j] = (half_t)0.f; }
            }
            *reinterpret_cast<half8*>(dst + rt * 64) = q0;
            *reinterpret_cast<half8*>(dst + rt * 64 + 16) = q1;
          }
        }
      }
    }
  }
  HD_STAMP(5)
  if (a.flags & 1) __builtin_amdgcn_s_setprio(1);
  if (a.flags & 2) __builtin_amdgcn_s_setprio(0);

  // ---- this lane's stage-B pixels; what only the decode needs (anchors, geometry, DFL weights) is requested here, behind
  //      stage A -- live across it these 30 registers were the difference to the three-workgroup shape's 168 -- and arrives
  //      during stage B
  int pixB[PB];
  float anc_x[PB], anc_y[PB], anc_s[PB];
  int anchor_i[PB];
  bool pvalid[PB];
#pragma unroll
  for (int p = 0; p < PB; ++p) {
    const int pt = wave + 4 * p;
    const int idx0 = 32 * pt + r;
    const int idx = idx0 < R2 ? idx0 : R2 - 1;
    const int ty = idx / TW, tx = idx - ty * TW;
    pixB[p] = (ty * RW1 + tx) * SPM * 16 + h * 16;
    const int gy = oy0 + ty, gx = ox0 + tx;
    pvalid[p] = pt < nB && idx0 < R2 && gy < a.H && gx < a.W;
    anchor_i[p] = a.anchor_off + (pvalid[p] ? gy * a.W + gx : 0);
    anc_x[p] = a.anchors[anchor_i[p]];
    anc_y[p] = a.anchors[a.A + anchor_i[p]];
    anc_s[p] = a.strides[anchor_i[p]];
  }
  const ImgGeom gm = a.geom[n];
  float dflw[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) dflw[i] = a.dfl_w[i];

  // ======================= stage B, box tower: 64 x (9 * 64) x tile pixels =======================
  floatx16 accB[2][PB];
#pragma unroll
  for (int rt = 0; rt < 2; ++rt) {
    const floatx16 b = A16 ? biasB_box[rt] : bias16(a.biasB + rt * 32 + h * 16);
#pragma unroll
    for (int p = 0; p < PB; ++p) accB[rt][p] = b;
  }
  int tapB[9];   // byte offset of tap (dy, dx) in MID
#pragma unroll
  for (int t = 0; t < 9; ++t) tapB[t] = ((t / 3) * RW1 + (t % 3)) * SPM * 16;
  {
    // three chunks of 12 steps, one pipeline; tap and channel group of every step are immediates after unrolling.  (The
    // first barrier also orders the MID stores before the reads.)
    int kq = 0;
    auto next_boff = [&]() {
      const int boff = tapB[kq >> 2] + (kq & 3) * 32;
      ++kq;
      return boff;
    };
    half8 af[2][2], bf[2][PB];
    constexpr int KSB = SLOTF / 2, NCB = 36 / KSB;   // 36 K steps (9 taps x 4 channel groups), two row tiles: SLOTF / 2 steps per chunk
    static_for<0, NCB>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      wsource(c + 2);
      kchunk<2, PB, KSB, i == 0, i == NCB - 1, SLOTF>(RING, c, lane16, MID, pixB, next_boff, ring_side, accB, af, bf, f7_0, f7_1);
      ++c;
    });
  }
  HD_STAMP(6)
  // ======================= stage B, class tower: (32*C3T) x (9 * 32*C3T) x tile pixels =======================
  floatx16 accC[C3T][PB];
#pragma unroll
  for (int rt = 0; rt < C3T; ++rt) {
    const floatx16 b = A16 ? biasB_cls[rt] : bias16(a.biasB + 64 + rt * 32 + h * 16);
#pragma unroll
    for (int p = 0; p < PB; ++p) accC[rt][p] = b;
  }
  {
    int kq = 0;
    auto next_boff = [&]() {
      const int tap = C3T == 2 ? kq >> 2 : kq >> 1, cg = C3T == 2 ? kq & 3 : kq & 1;
      ++kq;
      return tapB[tap] + (8 + 2 * cg) * 16;
    };
    half8 af[2][C3T], bf[2][PB];
    // 18 (one class row tile) or 36 (two; K padded to 64 channels) K steps; steps per chunk: what fits a slot, even
    constexpr int KST = C3T == 2 ? 36 : 18;
    constexpr int KSC = C3T == 2 ? SLOTF / 2 : (SLOTF >= 18 ? 18 : 6), NCC = KST / KSC;
    static_for<0, NCC>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      wsource(c + 2);
      kchunk<C3T, PB, KSC, i == 0, i == NCC - 1, SLOTF>(RING, c, lane16, MID, pixB, next_boff, ring_side, accC, af, bf, f7_0, f7_1);
      ++c;
    });
  }
  // ======================= stage C: the 1x1 projections from the accumulators, then decode =======================
  HD_STAMP(7)
  if (a.flags & 1) __builtin_amdgcn_s_setprio(0);
  if (a.flags & 2) __builtin_amdgcn_s_setprio(1);
  half8 wcb[2][4], wcc[2 * C3T];
  if constexpr (SLOTF == 8) {
    // two projection chunks: the class projection (chunk c, stored behind the class tower's steps), then the box projection
    // (chunk c + 1: its pieces are in wreg; the slot it goes to held the class tower's last chunk, which every wave has left)
    lds_barrier();
    const char* wc = RING + (c & 1) * SLOTB + lane16;
#pragma unroll
    for (int q = 0; q < 2 * C3T; ++q) wcc[q] = lds_h8(wc + q * 1024);
#pragma unroll
    for (int j = 0; j < NPW; ++j) wstore1(c + 1, j);
    lds_barrier();   // (publishes the box projection's slot: its fragments are read only where an anchor can pass, below)
  } else {
    lds_barrier();   // the projection chunk was stored behind the class tower's steps
    const char* wb = RING + (c & 1) * SLOT + lane16;
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
      for (int q = 0; q < 4; ++q) wcb[rt][q] = lds_h8(wb + (rt * 4 + q) * 1024);
#pragma unroll
    for (int q = 0; q < 2 * C3T; ++q) wcc[q] = lds_h8(wb + (8 + q) * 1024);
  }
  HD_STAMP(8)
  // class bias: two wave-uniform 16-float rows (scalar loads, no vector-memory round trip in front of the projection) + a select
  floatx16 bCc;
  if (a.nc == 1) {   // wave-uniform: only logit 0 is ever looked at; its bias was requested at the top of the kernel
#pragma unroll
    for (int i = 0; i < 16; ++i) bCc[i] = 0.f;
    bCc[0] = bias_c0;
  } else {
    bCc = bias16(a.biasC + 64 + h * 16);
  }
  bool box_w_loaded = SLOTF != 8;
#pragma unroll
  for (int p = 0; p < PB; ++p) {
    // B operands: element j of K step (mt, s) of this lane = channel 32*mt + 16*h + 8*s + j = accumulator register 8*s + j
    // ---- class projection first: its best score decides whether the anchor can become a candidate at all
    floatx16 oc = bCc;
#pragma unroll
    for (int mt = 0; mt < C3T; ++mt)
#pragma unroll
      for (int s = 0; s < 2; ++s) oc = mfma32(wcc[mt * 2 + s], silu_h8(accC[mt][p], 8 * s), oc);
    // class scores: this lane holds the logits of classes 16*h .. 16*h + 15.  The sigmoid is monotonic: the best class is the
    // arg-max of the LOGITS (selects, no branches, no transcendentals), and one sigmoid gives its score.
    float bl = -INFINITY;
    int best_c = 0;
    if (a.nc == 1) {   // wave-uniform: the one class (the reference's detectors) needs no arg-max
      bl = h == 0 ? oc[0] : -INFINITY;
    } else {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const float v = (16 * h + i < a.nc) ? oc[i] : -INFINITY;
        const bool up = v > bl;   // strict: the first maximum in class order stays
        bl = up ? v : bl;
        best_c = up ? 16 * h + i : best_c;
      }
      const float obl = __shfl_xor(bl, 32);
      const int oc_ = __shfl_xor(best_c, 32);
      if (h == 0 && obl > bl) { bl = obl; best_c = oc_; }   // the upper half wins only if larger
    }
    const float best = __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(bl * -1.4426950408889634f));
    // ---- box projection + DFL decode only where it can matter: emit_candidate keeps an anchor iff best > conf (the same
    //      expression, so the kept set is identical), and a typical image has a handful of such anchors among 8400 -- the box
    //      tower's projection (8 of this loop's 10 MFMAs), its 32 SiLUs and the two 16-bin softmaxes per lane are most of
    //      stage C's cycles.  Wave-uniform branch; the parity hook (out0) needs every anchor and takes it always.
    const bool pass = pvalid[p] && h == 0 && best > a.conf;
    if (a.out0 == nullptr && !__any(pass)) continue;
    if (SLOTF == 8 && !box_w_loaded) {   // wave-uniform; a typical wave never gets here (no anchor of its tile can pass)
      box_w_loaded = true;
      const char* wb = RING + ((c + 1) & 1) * SLOTB + lane16;
#pragma unroll
      for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int q = 0; q < 4; ++q) wcb[rt][q] = lds_h8(rt * 4 + q == 7 ? (((c + 1) & 1) ? f7_1 : f7_0) : wb + (rt * 4 + q) * 1024);
    }
    floatx16 ob[2];
    ob[0] = bias16(a.biasC + h * 16); ob[1] = bias16(a.biasC + 32 + h * 16);
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const half8 bq = silu_h8(accB[mt][p], 8 * s);
        ob[0] = mfma32(wcb[0][mt * 2 + s], bq, ob[0]);
        ob[1] = mfma32(wcb[1][mt * 2 + s], bq, ob[1]);
      }
    // ---- decode (model.ncnn.param:184-208).  The projection rows are permuted so that this lane holds, for row tile rt,
    //      the 16 bins of box side 2*rt + h.
    float dist[2];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
      // (fp32 logits straight from the accumulators: closer to the reference than the fp16 round trip of a stored projection)
      floatx16 l = ob[rt];
      float mx = l[0];
#pragma unroll
      for (int i = 1; i < 16; ++i) mx = fmaxf(mx, l[i]);
      float sum = 0.f, ex = 0.f;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const float e = __builtin_amdgcn_exp2f((l[i] - mx) * 1.4426950408889634f);
        sum += e;
        ex += e * dflw[i];
      }
      dist[rt] = ex * __builtin_amdgcn_rcpf(sum);
    }
    const float x0 = __shfl_xor(dist[0], 32), x1 = __shfl_xor(dist[1], 32);
    const float d0 = h ? x0 : dist[0], d1 = h ? dist[0] : x0, d2 = h ? x1 : dist[1], d3 = h ? dist[1] : x1;
    if (pvalid[p]) {
      const int anchor = anchor_i[p];
      float* o = a.out0 ? a.out0 + (long)n * (4 + a.nc) * a.A + anchor : nullptr;
      if (o) {   // parity hook only (lp_detect_raw): the whole score rows
#pragma unroll
        for (int i = 0; i < 16; ++i)
          if (16 * h + i < a.nc) o[(long)(4 + 16 * h + i) * a.A] = __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(oc[i] * -1.4426950408889634f));
      }
      if (h == 0) {
        const float ax = anc_x[p], ay = anc_y[p], s = anc_s[p];
        const float bx1 = ax - d0, by1 = ay - d1, bx2 = ax + d2, by2 = ay + d3;
        const float cx = (bx1 + bx2) * 0.5f * s, cy = (by1 + by2) * 0.5f * s;
        const float w = (bx2 - bx1) * s, hh = (by2 - by1) * s;
        if (o) { o[0] = cx; o[(long)a.A] = cy; o[2L * a.A] = w; o[3L * a.A] = hh; }
        emit_candidate(cx, cy, w, hh, best, best_c, anchor, gm, a.conf, a.cand + (long)n * a.A, a.cand_count + n);
      }
    }
  }
  HD_STAMP(9)
  HD_STAMP(15)
}

// ---------------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------------
// MFMA row rho of a 32-row tile carries physical channel 16*((rho>>2)&1) + 4*(rho>>3) + (rho&3): the D fragment of lane
// half h then holds channels 16*h .. 16*h+15 in its 16 registers
static inline int row_channel(int rho) { return 16 * ((rho >> 2) & 1) + 4 * (rho >> 3) + (rho & 3); }

static size_t head_lds(int th, int tw, int kpt, int c3t, int slotf, int ovl) {
  const size_t in = ((size_t)(th + 4) * (tw + 4) * (2 * kpt + 1) * 16 + 1023) & ~(size_t)1023;
  const size_t mid = ((size_t)(th + 2) * (tw + 2) * (4 * (2 + c3t) + 1) * 16 + 1023) & ~(size_t)1023;
  if (slotf == 8) {   // MID overlays the input tile; slots 64 B short of 8 KiB (kchunk): 53,696 B = 42 LDS granules of 1280 B
    const size_t in_b = (size_t)(th + 4) * (tw + 4) * (2 * kpt + 1) * 16, mid_b = (size_t)(th + 2) * (tw + 2) * (4 * (2 + c3t) + 1) * 16;
    return (in_b > mid_b ? in_b : mid_b) + 2 * (size_t)(8192 - 64);
  }
  if (ovl) return (in > mid ? in : mid) + 2 * (size_t)slotf * 1024;
  return in + mid + 2 * (size_t)slotf * 1024;
}

// One kernel configuration per (class row tiles, K steps per tap = Cin / 16): tile shape TH x TW, pixel tiles per wave in stage
// A / B (ceil((TH+2)(TW+2)/32) <= 4*PA, ceil(TH*TW/32) <= 4*PB), 64-slot pieces per input-tile row, K steps per stage-A chunk
// (a divisor of 9*KPT with (2+C3T)*KSA <= 24 fragments).  Chosen for LDS (input tile + first-conv image + 48 KiB ring <= 160
// KiB) and for whole tiles on the 80 / 40 / 20 maps of a 640 input; other map sizes run the same shapes with masked edges.
// slotf: fragments per weight-ring slot.  The v1 P3 entries with an 8 x 16 tile and one pixel tile per wave in stage B share a
// CU: the VALU phases (SiLU epilogues, decode: transcendental-bound) and the prologue of one workgroup run under the MFMA phases
// of the others -- with one workgroup per CU the matrix pipe idles through them.  slotf 12: two per CU (80 KiB, <= 256
// registers; round 2).  slotf 8: THREE per CU (round 3: 98 us per launch against 112): MID overlays the input tile, 8-fragment
// slots 64 bytes short of 8 KiB (the LDS granule is 1280 B: 42 granules = 53,760 B per workgroup is the limit, measured with
// tools/ubench/lds_occupancy.hip; MID + two whole slots would be 53,824), the projections as two chunks, and 168 registers --
// reached by requesting what only the decode needs (anchors, geometry, DFL weights) after stage A instead of before it.
struct HeadCfg { int c3t, kpt, th, tw, pa, pb, npc, ksa, slotf, ovl, a16; };
// a16 (round 4): stage A on 16x16x32 MFMAs -- pa counts 16-pixel tiles per wave, ksa half-steps per chunk (one tap per chunk)
static const HeadCfg kHeadCfg[] = {
    {1, 2, 8, 16, 3, 1, 2, 2, 8, 1, 1},      // v1 P3: Cin 32, THREE workgroups per CU, stage A on 16-pixel tiles (12 for 180 pixels)
    {1, 4, 10, 20, 5, 2, 4, 4, 12, 1, 1},    // v1 P4: Cin 64, TWO workgroups per CU, 17 tiles of 16 for 264 pixels
    {1, 8, 10, 10, 3, 1, 4, 8, 24, 0, 1},    // v1 P5: Cin 128, 9 tiles of 16 for 144 pixels
    {2, 3, 8, 16, 3, 1, 3, 2, 12, 1, 1},     // v2 P3: Cin 48 (K padded to 64 per tap), 48-channel class tower, TWO workgroups per CU (73.5 KB)
    {2, 6, 10, 10, 3, 1, 3, 2, 12, 1, 1},    // v2 P4: Cin 96, 9 tiles of 16 for 144 pixels; MID over the input tile + 12-fragment slots: 64.8 KB, TWO per CU
    {2, 12, 10, 10, 3, 1, 6, 6, 24, 1, 1},   // v2 P5: Cin 192, 10 x 10 tiles (256 workgroups = one round; MID over the input tile: 126 KB)
    {1, 2, 8, 16, 2, 1, 2, 2, 8, 1, 0},      // v1 P3, round 3's shape: stage A on 32-pixel slots (LITEPI_HEAD_A32=1)
    {1, 2, 8, 16, 2, 1, 2, 2, 12, 0, 0},     // v1 P3: Cin 32, two workgroups per CU (LITEPI_HEAD_2WG=1)
    {1, 2, 16, 16, 3, 2, 2, 6, 24, 0, 0},    // v1 P3: Cin 32
    {1, 4, 10, 20, 3, 2, 4, 4, 12, 1, 0},    // v1 P4: Cin 64, TWO workgroups per CU (MID over the input tile: 79.5 KB)
    {1, 4, 10, 20, 3, 2, 4, 6, 24, 0, 0},    // v1 P4: Cin 64 (LITEPI_HEAD_2WG=1 / LITEPI_HEAD_1WG=1)
    {1, 8, 10, 10, 2, 1, 4, 8, 24, 0, 0},    // v1 P5: Cin 128 (256 workgroups: the overlay shape at 76 KB changed nothing end to end)
    {2, 3, 10, 20, 3, 2, 3, 3, 24, 0, 0},    // v2 P3: Cin 48
    {2, 6, 10, 10, 2, 1, 3, 6, 24, 0, 0},    // v2 P4: Cin 96
    {2, 12, 8, 8, 1, 1, 5, 6, 24, 0, 0},     // v2 P5: Cin 192
};
static const HeadCfg* find_cfg(int c3t, int kpt) {
  static const bool one_wg = getenv("LITEPI_HEAD_1WG") != nullptr;   // A/B switches: the 16 x 16 one-workgroup shape,
  static const bool two_wg = getenv("LITEPI_HEAD_2WG") != nullptr;   // the two-workgroup shape of round 2,
  static const bool a32 = getenv("LITEPI_HEAD_A32") != nullptr;      // round 3's stage A (32-pixel slots)
  for (auto& c : kHeadCfg)
    if (c.c3t == c3t && c.kpt == kpt && !(one_wg && c.slotf <= 12) && !(c.ovl && two_wg) && !(c.a16 && (a32 || one_wg || two_wg))) return &c;
  return nullptr;
}

bool HeadLayer::supported(int cin_phys, int c2, int c3, int nc, int reg_max, int h, int w) {
  if (c2 != 64 || reg_max != 16 || nc < 1 || nc > 32 || c3 < 8 || c3 > 64 || c3 % 8 != 0) return false;
  if (cin_phys % 16 != 0) return false;
  (void)h; (void)w;
  return find_cfg(c3 > 32 ? 2 : 1, cin_phys / 16) != nullptr;
}

void HeadLayer::build(int cin_phys, int c3_, int nc_, int h, int w, int batch_hint, const Src& s) {
  Cin = cin_phys; c3 = c3_; nc = nc_; H = h; W = w;
  C3T = c3 > 32 ? 2 : 1;
  KPT = Cin / 16;
  const HeadCfg* cfg = find_cfg(C3T, KPT);
  LP_CHECK(cfg, LP_ERR_STATE, "Detect head %s: no kernel configuration for Cin %d", name.c_str(), Cin);
  TH = cfg->th; TW = cfg->tw; PA = cfg->pa; PB = cfg->pb; NPC = cfg->npc; KSA = cfg->ksa; SLOTF = cfg->slotf; OVL = cfg->ovl; A16 = cfg->a16;
  lds_bytes = head_lds(TH, TW, KPT, C3T, SLOTF, OVL);
  const int RT = 2 + C3T, CM = 32 * C3T;
  // chunks hold at most 24 fragments (one ring slot): stage A KSA K steps x RT row tiles, stage B box 12 x 2, class 18 x 1
  // (12 x 2 for two class row tiles), projections 10 (12)
  LP_CHECK(KSA % 2 != 0 || 9 * KPT / KSA >= 2, LP_ERR_STATE, "Detect head %s: stage A needs two chunks", name.c_str());
  const int SPT = (KPT + 1) / 2;   // A16: K steps of 32 channels per tap (Cin = 16 mod 32: the last one half zero weights)
  if (A16)   // a chunk is KSA half-steps = KSA / 2 K steps x two row halves; fragment 7 of an 8-fragment slot is not addressable
    LP_CHECK(KSA % 2 == 0 && (9 * SPT) % (KSA / 2) == 0 && RT * KSA <= (SLOTF == 8 ? 7 : SLOTF) && (TH + 2) * (TW + 2) <= 64 * PA &&
                 (KPT % 2 == 0 || (OVL && (size_t)(TH + 4) * (TW + 4) * (2 * KPT + 1) + 1 <= (size_t)(TH + 2) * (TW + 2) * (4 * RT + 1))), LP_ERR_STATE,   // (odd KPT: the zeroed guard slot behind the tile lies inside MID's span)
             "Detect head %s: inconsistent 16-pixel-tile configuration", name.c_str());
  LP_CHECK(lds_bytes <= (SLOTF == 8 ? 53760u : (SLOTF == 12 ? 80u * 1024 : 160u * 1024)) && ((TW + 4) * (2 * KPT + 1) + 63) / 64 == NPC &&
               (A16 || ((9 * KPT) % KSA == 0 && RT * KSA <= SLOTF)) && TH + 4 <= (SLOTF <= 12 && !(OVL && SLOTF == 12) ? 12 : (OVL ? 16 : 20)) &&
               (SLOTF == 8 || SLOTF == 12 || SLOTF == 24) &&
               (SLOTF == 8 ? C3T == 1 : 8 + 2 * C3T <= SLOTF) &&
               (A16 || (TH + 2) * (TW + 2) <= 128 * PA) && TH * TW <= 128 * PB, LP_ERR_STATE, "Detect head %s: inconsistent configuration", name.c_str());
  (void)batch_hint;
  std::vector<uint16_t> stream;
  auto frag = [&](auto&& weight_of) {  // weight_of(row rho, k element e of the K step) -> float; appends one 1 KiB fragment
    const size_t base = stream.size();
    stream.resize(base + 512);
    for (int lane = 0; lane < 64; ++lane)
      for (int j = 0; j < 8; ++j) stream[base + lane * 8 + j] = f32_to_f16(weight_of(lane & 31, 8 * (lane >> 5) + j));
  };
  coff.clear(); csz.clear(); cks.clear();
  auto begin_chunk = [&]() { coff.push_back((unsigned short)(stream.size() / 512)); };
  auto end_chunk = [&](int ksteps) {
    const size_t nf = stream.size() / 512 - coff.back();
    LP_CHECK(nf >= 1 && (int)nf <= SLOTF, LP_ERR_STATE, "Detect head: chunk of %zu fragments", nf);
    csz.push_back((unsigned char)nf);
    cks.push_back((unsigned char)ksteps);
    stream.resize((size_t)(coff.back() + SLOTF) * 512, 0);   // every chunk is a whole slot image
  };
  const std::vector<float>& wa = *s.wa;  // [64 + c3][9][Cin]
  // ---- stage A on 16x16x32 MFMAs: chunk = tap; K step (tap, q) covers input channels 32 q .. 32 q + 31; per K step two half-steps
  //      of RT row tiles (16 rows each).  Fragment lane l: row l & 15 of the tile, K group kg = l >> 4 = channel slot gam(kg) =
  //      {0, 2, 1, 3} of the step's four (the kernel's pixel fragments read the same slot order: conflict-free LDS reads).  Row r
  //      of row tile R (= rh * RT + rt) is physical MID channel 32 (R >> 1) + 8 (r >> 2) + 4 (R & 1) + (r & 3): a lane's D
  //      registers of tiles 2t | 2t+1 are then 8 consecutive channels of one pixel -> one 16-byte MID store.
  auto frag16 = [&](auto&& weight_of) {  // weight_of(row r, k element e of the 32) -> float; appends one 1 KiB fragment
    const size_t base = stream.size();
    stream.resize(base + 512);
    for (int lane = 0; lane < 64; ++lane)
      for (int j = 0; j < 8; ++j) {
        const int kg = lane >> 4, slot = (kg & 1) * 2 + (kg >> 1);
        stream[base + lane * 8 + j] = f32_to_f16(weight_of(lane & 15, 8 * slot + j));
      }
  };
  if (A16) {
    for (int ks = 0; ks < 9 * SPT; ++ks) {
      const int tap = ks / SPT, q = ks % SPT;
      if (ks % (KSA / 2) == 0) begin_chunk();
      for (int rh = 0; rh < 2; ++rh)
        for (int rt = 0; rt < RT; ++rt)
          frag16([&](int r, int e) {
            const int R = rh * RT + rt;
            const int c = 32 * (R >> 1) + 8 * (r >> 2) + 4 * (R & 1) + (r & 3);   // physical MID channel
            const int src = c < 64 ? c : (c - 64 < c3 ? 64 + (c - 64) : -1);
            return (src < 0 || 32 * q + e >= Cin) ? 0.f : wa[((size_t)src * 9 + tap) * Cin + 32 * q + e];
          });
      if (ks % (KSA / 2) == KSA / 2 - 1) end_chunk(KSA);
    }
  }
  // ---- stage A: K step (tap, cg): element e = input channel 16*cg + e
  for (int ks = 0; ks < (A16 ? 0 : 9 * KPT); ++ks) {
    if (ks % KSA == 0) begin_chunk();
    const int tap = ks / KPT, cg = ks % KPT;
    for (int rt = 0; rt < RT; ++rt)
      frag([&](int rho, int e) {
        const int c = rt * 32 + row_channel(rho);          // physical MID channel
        const int src = c < 64 ? c : (c - 64 < c3 ? 64 + (c - 64) : -1);
        return src < 0 ? 0.f : wa[((size_t)src * 9 + tap) * Cin + 16 * cg + e];
      });
    if (ks % KSA == KSA - 1) end_chunk(KSA);
  }
  // ---- stage B box: K step kq = (tap, cg), 4 per tap
  const int KSB = SLOTF / 2;   // two row tiles per step
  for (int kq = 0; kq < 36; ++kq) {
    if (kq % KSB == 0) begin_chunk();
    const int tap = kq >> 2, cg = kq & 3;
    for (int rt = 0; rt < 2; ++rt)
      frag([&](int rho, int e) { return (*s.wbb)[((size_t)(rt * 32 + row_channel(rho)) * 9 + tap) * 64 + 16 * cg + e]; });
    if (kq % KSB == KSB - 1) end_chunk(KSB);
  }
  // ---- stage B class: 2*C3T K steps per tap
  {
    const int per_tap = 2 * C3T, total = 9 * per_tap, per_chunk = C3T == 2 ? SLOTF / 2 : (SLOTF >= 18 ? 18 : 6);
    for (int kq = 0; kq < total; ++kq) {
      if (kq % per_chunk == 0) begin_chunk();
      const int tap = kq / per_tap, cg = kq % per_tap;
      for (int rt = 0; rt < C3T; ++rt)
        frag([&](int rho, int e) {
          const int co = rt * 32 + row_channel(rho), ci = 16 * cg + e;
          return (co < c3 && ci < c3) ? (*s.wbc)[((size_t)co * 9 + tap) * c3 + ci] : 0.f;
        });
      if (kq % per_chunk == per_chunk - 1) end_chunk(per_chunk);
    }
  }
  // ---- stage C: K step (mt, s): element e of lane half hh = mid channel 32*mt + 16*hh + 8*s + (e & 7), hh = e >> 3
  auto box_proj = [&]() {
    for (int rt = 0; rt < 2; ++rt)
      for (int q = 0; q < 4; ++q)
        frag([&](int rho, int e) {
          const int mt = q >> 1, sq = q & 1;
          const int ci = 32 * mt + 16 * (e >> 3) + 8 * sq + (e & 7);
          return (*s.wpb)[(size_t)(rt * 32 + row_channel(rho)) * 64 + ci];
        });
  };
  auto cls_proj = [&]() {
    for (int q = 0; q < 2 * C3T; ++q)
      frag([&](int rho, int e) {
        const int mt = q >> 1, sq = q & 1;
        const int ci = 32 * mt + 16 * (e >> 3) + 8 * sq + (e & 7), co = row_channel(rho);
        return (co < nc && ci < c3) ? (*s.wpc)[(size_t)co * c3 + ci] : 0.f;
      });
  };
  if (SLOTF == 8) {   // 8-fragment slots: the class projection (needed first) and the box projection are two chunks
    begin_chunk(); cls_proj(); end_chunk(0);
    begin_chunk(); box_proj(); end_chunk(0);
  } else {
    begin_chunk(); box_proj(); cls_proj(); end_chunk(0);
  }
  nchunks = (int)coff.size();
  LP_CHECK(nchunks <= 64 && stream.size() / 512 < 65536, LP_ERR_STATE, "Detect head: weight stream too long (%d chunks)", nchunks);
  stream.resize(stream.size() + (size_t)2 * SLOTF * 512, 0);   // the ring requests two chunks past the end
  d_stream.alloc(stream.size() * 2 + 64);   // (one copy: replicating the stream per XCD changed nothing -- it is L2-resident)
  LP_HIP(hipMemcpy(d_stream.p, stream.data(), stream.size() * 2, hipMemcpyHostToDevice));
  std::vector<float> bA(32 * RT + 16, 0.f), bB(32 * RT + 16, 0.f), bC(64 + 32 + 16, 0.f);
  for (int c = 0; c < 64; ++c) { bA[c] = (*s.ba)[c]; bB[c] = s.bbb->empty() ? 0.f : (*s.bbb)[c]; bC[c] = s.bpb->empty() ? 0.f : (*s.bpb)[c]; }
  for (int c = 0; c < c3; ++c) { bA[64 + c] = (*s.ba)[64 + c]; bB[64 + c] = s.bbc->empty() ? 0.f : (*s.bbc)[c]; }
  for (int c = 0; c < nc; ++c) bC[64 + c] = s.bpc->empty() ? 0.f : (*s.bpc)[c];
  auto up = [](DevBuf& d, const std::vector<float>& v) {
    d.alloc(v.size() * 4);
    LP_HIP(hipMemcpy(d.p, v.data(), v.size() * 4, hipMemcpyHostToDevice));
  };
  up(d_biasA, bA); up(d_biasB, bB); up(d_biasC, bC);
  (void)CM;
  macs_per_image = (double)h * w * (9.0 * Cin * (64 + c3) + 9.0 * 64 * 64 + 9.0 * c3 * c3 + 64.0 * 64 + (double)c3 * nc);
}

void HeadLayer::launch(const View& in, int N, int anchor_off, int A, const float* anchors, const float* strides, const float* dfl_w,
                       float* out0, const ImgGeom* geom, Cand* cand, int* cand_count, float conf, hipStream_t st) const {
  HeadArgs a;
  memset(&a, 0, sizeof(a));
  a.in = in.base; a.in_pitch = in.pitch; a.wstream = d_stream.p;
  a.biasA = d_biasA.as<float>(); a.biasB = d_biasB.as<float>(); a.biasC = d_biasC.as<float>();
  a.zeros = d_biasC.as<float>() + 96;  // the bias buffer ends in 16 zero floats
  a.anchors = anchors; a.strides = strides; a.dfl_w = dfl_w; a.out0 = out0; a.geom = geom; a.cand = cand; a.cand_count = cand_count;
  a.conf = conf;
  a.N = N; a.H = in.H; a.W = in.W; a.Cin = Cin;
  a.TH = TH; a.TW = TW; a.tiles_x = ceil_div(in.W, TW); a.ntiles = a.tiles_x * ceil_div(in.H, TH);
  a.KPT = KPT; a.nchunks = nchunks; a.A = A; a.nc = nc; a.anchor_off = anchor_off;
  LP_CHECK(in.C == Cin && in.H == H && in.W == W && (int)coff.size() <= 64, LP_ERR_STATE,
           "Detect head %s: view does not match the plan", name.c_str());
  const dim3 grid((unsigned)(a.ntiles * N));
  static const int head_flags = getenv("LITEPI_HEAD_FLAGS") ? atoi(getenv("LITEPI_HEAD_FLAGS")) : 0;
  a.flags = head_flags;
  static const char* stamp_path = getenv("LITEPI_HEAD_STAMPS");
  DevBuf d_stamps;
  if (stamp_path && *stamp_path) {
    d_stamps.alloc((size_t)grid.x * 16 * 8);
    a.stamps = d_stamps.as<unsigned long long>();
  }
#define LP_HEAD(C3T_, PA_, PB_, NPC_, KSA_, SLOTF_, NRW_)                                                                          \
  {                                                                                                                               \
    set_max_dynamic_lds(reinterpret_cast<const void*>(head_fused_kernel<C3T_, PA_, PB_, NPC_, KSA_, SLOTF_, NRW_>), 160 * 1024); \
    LP_LAUNCH((head_fused_kernel<C3T_, PA_, PB_, NPC_, KSA_, SLOTF_, NRW_>), grid, dim3(256), lds_bytes, st, a);         \
  }
#define LP_HEAD2(C3T_, PA_, PB_, NPC_, KSA_, SLOTF_, NRW_, OV_)                                                                          \
  {                                                                                                                                    \
    set_max_dynamic_lds(reinterpret_cast<const void*>(head_fused_kernel<C3T_, PA_, PB_, NPC_, KSA_, SLOTF_, NRW_, OV_>), 160 * 1024); \
    LP_LAUNCH((head_fused_kernel<C3T_, PA_, PB_, NPC_, KSA_, SLOTF_, NRW_, OV_>), grid, dim3(256), lds_bytes, st, a);                 \
  }
#define LP_HEAD16(C3T_, PA_, PB_, NPC_, KSA_, SLOTF_, NRW_, OV_)                                                                               \
  {                                                                                                                                          \
    set_max_dynamic_lds(reinterpret_cast<const void*>(head_fused_kernel<C3T_, PA_, PB_, NPC_, KSA_, SLOTF_, NRW_, OV_, true>), 160 * 1024); \
    LP_LAUNCH((head_fused_kernel<C3T_, PA_, PB_, NPC_, KSA_, SLOTF_, NRW_, OV_, true>), grid, dim3(256), lds_bytes, st, a);                 \
  }
#define LP_HEAD16N(C3T_, PA_, PB_, NPC_, KSA_, SLOTF_, NRW_, OV_, NCA_)                                                                               \
  {                                                                                                                                                \
    set_max_dynamic_lds(reinterpret_cast<const void*>(head_fused_kernel<C3T_, PA_, PB_, NPC_, KSA_, SLOTF_, NRW_, OV_, true, NCA_>), 160 * 1024); \
    LP_LAUNCH((head_fused_kernel<C3T_, PA_, PB_, NPC_, KSA_, SLOTF_, NRW_, OV_, true, NCA_>), grid, dim3(256), lds_bytes, st, a);                 \
  }
  if (A16 && C3T == 2 && KPT == 3) LP_HEAD16N(2, 3, 1, 3, 2, 12, 3, true, 18)
  else if (A16 && C3T == 2 && KPT == 6) LP_HEAD16N(2, 3, 1, 3, 2, 12, 4, true, 27)
  else if (A16 && C3T == 2 && KPT == 12) LP_HEAD16N(2, 3, 1, 6, 6, 24, 4, true, 18)
  else if (A16 && KPT == 2) LP_HEAD16(1, 3, 1, 2, 2, 8, 3, true)
  else if (A16 && KPT == 4) LP_HEAD16(1, 5, 2, 4, 4, 12, 4, true)
  else if (A16 && KPT == 8) LP_HEAD16(1, 3, 1, 4, 8, 24, 5, false)
  else if (C3T == 1 && KPT == 2 && SLOTF == 8) LP_HEAD(1, 2, 1, 2, 2, 8, 3)
  else if (C3T == 1 && KPT == 2 && SLOTF == 12) LP_HEAD(1, 2, 1, 2, 2, 12, 3)
  else if (C3T == 1 && KPT == 2) LP_HEAD(1, 3, 2, 2, 6, 24, 5)
  else if (C3T == 1 && KPT == 4 && SLOTF == 12) LP_HEAD2(1, 3, 2, 4, 4, 12, 4, true)
  else if (C3T == 1 && KPT == 4) LP_HEAD(1, 3, 2, 4, 6, 24, 5)
  else if (C3T == 1 && KPT == 8) LP_HEAD(1, 2, 1, 4, 8, 24, 5)
  else if (C3T == 2 && KPT == 3) LP_HEAD(2, 3, 2, 3, 3, 24, 5)
  else if (C3T == 2 && KPT == 6) LP_HEAD(2, 2, 1, 3, 6, 24, 5)
  else if (C3T == 2 && KPT == 12) LP_HEAD(2, 1, 1, 5, 6, 24, 5)
  else throw Error(LP_ERR_STATE, "Detect head: no kernel configuration");
#undef LP_HEAD
#undef LP_HEAD2
#undef LP_HEAD16
#undef LP_HEAD16N
  LP_HIP(hipGetLastError());
  if (a.stamps) {  // diagnostic: dump [grid][16] stamps, one record per launch
    LP_HIP(hipStreamSynchronize(st));
    std::vector<unsigned long long> hs((size_t)grid.x * 16);
    LP_HIP(hipMemcpy(hs.data(), d_stamps.p, hs.size() * 8, hipMemcpyDeviceToHost));
    if (FILE* f = fopen(stamp_path, "ab")) {
      const unsigned long long hdr[4] = {0x48454144ull, grid.x, (unsigned long long)in.H, (unsigned long long)N};
      fwrite(hdr, 8, 4, f);
      fwrite(hs.data(), 8, hs.size(), f);
      fclose(f);
    }
  }
}

}  // namespace lp
