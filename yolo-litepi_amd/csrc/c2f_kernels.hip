// Whole-C2f launches for gfx950 (fp16 storage, fp32 accumulate).  One workgroup (8 waves) owns a 20x20 tile of one image
// -- the whole image on the 20x20 level -- and runs every layer of a C2f module of the reference graph
// (model.ncnn.param: cv1 -> Slice -> n x [3x3 -> 3x3 -> add] -> Concat -> cv2) back to back:
//
//   [s2]   3x3 stride-2 conv + SiLU in front of the module            (whole-image configurations only)
//   cv1    1x1 + SiLU over concat(src0 [nearest-x2 upsampled], src1)  -> y0 | y1
//   a_k    3x3 + SiLU on y_{k+1}                                     -> mid          (LDS -> LDS)
//   b_k    3x3 + SiLU on mid, + y_{k+1}                              -> y_{k+2}      (LDS -> LDS, in place of y_{k+1})
//   cv2    1x1 + SiLU over concat(y0 .. y_{n+1})                     -> out
//   [sppf] cv1 -> 5x5 max pool x3 -> cv2                              (whole-image configuration of the backbone's last stage)
//
// Data movement rule: ONLY what a 3x3 stride-1 conv (or a pool) reads lives in LDS -- two planes of c channels over the
// tile + halo frame.  Every 1x1 conv takes its pixel operand straight from global memory / L2 (16 B per lane = the MFMA B
// fragment of one pixel's 8 channels), including tensors this workgroup stored a phase earlier (y0 .. y_{n+1} go through the
// module's concat buffer, which exists anyway): the concat is an address, never a copy, and the LDS budget does not depend
// on the module's width.  Weights never touch LDS either: each wave streams the A fragments of its channel block from L2
// one K step ahead of the MFMAs.
//
// MFMA: v_mfma_f32_16x16x32_f16, D[out-channel][pixel] = W . X, a wave owns NT channel tiles x PT pixel tiles.  A pixel tile
// is 16 consecutive pixels of the phase's (image-clipped) region, row-major.  LDS pixel pitch = 2c + 16 bytes (an odd number
// of 16-byte slots); column `col` of the tile holds pixel offset sig(col) with sig = even offsets for the lanes
// {0-3, 12-15} and odd offsets for {4-11}, and K group g reads channel group gam(g) = {0, 2, 1, 3}: inside every
// ds_read_b128 lane group ({0-3,12-15,20-27}, ..) the 16 lanes then hit 16 distinct 16-byte slots -- conflict-free
// fragment reads without padding rows (MI355X guide, LDS).
//
// Round 4 (v2, the paper's widths c = 24 / 48 / 96; yolo_plus_ncnn_model/model.ncnn.param): general K packing for the 3x3 convs
// whose taps are not whole K steps (C2fCfg::GK), channel tiles in threes, 10-row / half-image tiles and weights from L2 where two
// planes fill the LDS, cv2 with a per-lane source table (pw_gk_phase); and, on the same machinery, s2lds_kernel (stride-2 3x3 conv
// from an LDS-staged input tile, output-channel and input-channel splits) and sppf_kernel (SPPF in one plane, pools in place).
#include "c2f.h"

#include <cstdlib>

namespace lp {

typedef _Float16 half_t;
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

namespace {

constexpr int cdiv_c(int a, int b) { return (a + b - 1) / b; }
constexpr int min_c(int a, int b) { return a < b ? a : b; }
constexpr int max_c(int a, int b) { return a > b ? a : b; }
// how many K steps of pixel fragments from global memory a 1x1 phase keeps in flight (C2fCfg::DEEP): as many as ~180
// registers of fragments + accumulators allow.  (Measured neutral: cv1 of the c = 24 module at depth 5 instead of 2, and of
// v1's c = 16 FPN module at 3 instead of 2, ran in the same 82 / 46 us -- a K step's 1.3 k cycles are not one exposed round
// trip, the fetch path is busy with the eight waves' streams whatever the depth.  Kept for v2's configurations.)
constexpr int deep_depth(int S, int NT, int PT, bool ldsw) {
  int d = 2;
  for (int t = 3; t <= 6 && t <= S; ++t)
    if ((t * (PT + (ldsw ? 0 : NT)) + (ldsw ? 2 * NT : 0) + NT * PT) * 4 <= 180) d = t;
  return d;
}

template <int C_, int NB_, int KA_, int KB_, bool UP_, int COUT_, int MODE_, int KS2_, int TH_ = 20, int NW_PERIMG = 8, bool AW_ = true, int WPS_ = 0,
          bool DEEP_ = false, int NWT_ = 0>
struct C2fCfg {
  static constexpr int C = C_, NB = NB_, KA = KA_, KB = KB_, COUT = COUT_, MODE = MODE_, KS2 = KS2_;
  static constexpr bool UP = UP_;
  static constexpr bool PERIMG = MODE_ >= 1;   // the tile is the whole image: no halo recompute, a 1-pixel zero ring
  // MODE -1: the module WITHOUT cv1 -- the launch in front (the stride-2 conv with cv1 as its 1x1 tail, s2lds_kernel) wrote y0 | y1
  // into the concat buffer; y1 (+ halo) is copied into plane 0.  For n = 2 modules cv1 on the halo-4 region was 2x its work.
  static constexpr bool XCV1 = MODE_ == -1;
  // waves per workgroup.  c = 16: four, so that two workgroups (LDS allows it) share a CU at one wave each per SIMD with
  // the whole register file -- two independent workgroups drift out of phase and cover each other's epilogues and waits;
  // eight waves under a 128-register cap spilled in every epilogue
  // whole-image configurations: NW_PERIMG = 8.  (Four waves with the whole register file each -- every wave streams its
  // channel block's weights from L2 itself, so halving the waves halves that traffic -- measured slower: 112 vs 100 us for
  // the backbone kernel; one wave per SIMD exposes every LDS and L2 latency.)
  // (NWT_: a tile configuration of c < 32 whose LDS allows only ONE workgroup per CU runs eight waves like the wide ones)
  static constexpr int NW = MODE_ >= 1 ? NW_PERIMG : (NWT_ > 0 ? NWT_ : (C_ >= 32 ? 8 : 4));
  static constexpr int TH = TH_, TW = 20;
  static constexpr int F = PERIMG ? 1 : 2 * NB;  // frame margin around the tile
  static constexpr int LW = TW + 2 * F, LH = TH + 2 * F;
  // general K packing of the 3x3 convs (v2's widths c = 24 / 48: a tap is not a whole number of 32-channel K steps): K group
  // q = 4 s + gam -> (tap q / (c/8), channel group q % (c/8)), a per-lane table of LDS offsets (c3_phase)
  static constexpr bool GK = C % 32 != 0 && C != 16;
  static constexpr bool DEEP = DEEP_ || GK || C == 96;   // (v2's configurations, A/B variants of v1's: deep_depth)
  // bytes per LDS pixel of a c-channel plane: an odd number of 16-byte slots.  c % 16 != 0 (c = 24: 48 B = 3 slots): no pad at
  // all -- the lanes that own the 8 padding rows of the second channel tile skip their stores
  static constexpr int PS = C % 16 != 0 ? 2 * C : 2 * C + 16;
  // c = 16, one bottleneck: plane 0 holds cv1's whole output y0 | y1 (2c channels per pixel), so that cv2's first K step
  // (32 channels) is one plane-0 read and the module's concat buffer is never touched: no global round trip of y0 .. y2
  static constexpr bool Y01 = C == 16 && NB == 1 && MODE_ == 0;
  static constexpr int PS0 = Y01 ? 4 * C + 16 : PS;   // plane 0 pixel pitch
  static constexpr int Y1OFF = Y01 ? 2 * C : 0;       // byte offset of y1 inside a plane-0 pixel
  static constexpr int PLANE0 = LW * LH * PS0, PLANE = LW * LH * PS;
  static constexpr int XC = 2 * C;               // output channels of the entry conv
  // how far each phase's region extends beyond the tile
  static constexpr int e_cv1 = PERIMG ? 0 : 2 * NB;
  static constexpr int e_a(int k) { return PERIMG ? 0 : 2 * (NB - k) - 1; }
  static constexpr int e_b(int k) { return PERIMG ? 0 : 2 * (NB - k) - 2; }
  static constexpr int npt(int e) { return cdiv_c((TH + 2 * e) * (TW + 2 * e), 16); }
  // block shapes (NT channel tiles x PT pixel tiles per wave, CB channel blocks): one round of blocks per phase, two
  // where one would not fit the register file (cap_pt)
  static constexpr int MAXT = (MODE_ >= 1 && NW_PERIMG == 4) ? 52 : 20;   // accumulator tiles a wave can hold
  static constexpr int cap_pt(int nt, int pt) { return nt * pt > (nt == 3 ? 21 : MAXT) ? cdiv_c(pt, 2) : pt; }
  // (whole-image configurations: cv1's fragments are staged in plane 1, which nothing uses before the first bottleneck conv,
  //  and a wave holds every output channel of two pixel tiles -- as the entry conv, see WB_S2)
  static constexpr bool W1_LDS = PERIMG;
  // (channel tiles in threes where the count allows it -- v2's 48 / 96 / 192 channels: NT = 3, a lane owns 12 consecutive channels)
  static constexpr int CT1 = 2 * C / 16, NT1 = W1_LDS ? CT1 : (CT1 % 3 == 0 ? 3 : min_c(CT1, 4)), CB1 = CT1 / NT1,
                       PT1 = W1_LDS ? 2 : cap_pt(NT1, cdiv_c(npt(e_cv1), NW / CB1));
  static constexpr int CTM = cdiv_c(C, 16), NTM = CTM % 3 == 0 ? 3 : min_c(CTM, 2), CBM = CTM / NTM;
  static constexpr int ptm(int e) { return cdiv_c(npt(e), NW / CBM); }
  static constexpr int CT2 = COUT / 16, NT2 = CT2 % 3 == 0 ? 3 : (CT2 >= 2 ? CT2 / 2 : 1), CB2 = CT2 / NT2, PT2 = cap_pt(NT2, cdiv_c(npt(0), NW / CB2));
  static constexpr int PT2W = cdiv_c(npt(0), NW / CB2);   // whole-image cv2: one round of blocks (pw_sync_phase)
  static constexpr int NTS = NTM, CBS = CBM, PTS = cdiv_c(npt(0), NW / CBS);  // SPPF.cv1 (c outputs)
  static constexpr int WPS = WPS_ > 0 ? WPS_ : (NW == 4 && MODE_ >= 1 ? 1 : 2);   // waves per SIMD the register allocation must allow
  // cv2 reads y_NB and y_{NB+1} from the LDS planes (whole K steps need c >= 32); the earlier segments from the concat buffer
  // cv2's K steps straddle the concat's segments (c = 24 / 48; c = 16 with two bottlenecks: 32 channels of y0 | y1 from the concat
  // buffer, y2 and y3 from the planes): pw_gk_phase's per-lane source table
  static constexpr bool CV2_TAB = GK || (C == 16 && NB == 2);
  static constexpr bool CV2_LDS = C >= 32 || Y01 || CV2_TAB;
  static constexpr int K2G = Y01 ? 0 : (CV2_LDS ? NB * C : (2 + NB) * C);
  // weights staged in LDS (tile configurations): fragment bytes of every phase, in execution order
  static constexpr bool AW = !PERIMG && AW_;
  static constexpr int c3_steps = GK ? cdiv_c(9 * (C / 8), 4) : (C >= 32 ? 9 * (C / 32) : 5);
  static constexpr int WB_CV1 = CT1 * cdiv_c(KA + KB, 32) * 1024, WB_M = CTM * c3_steps * 1024, WB_CV2 = CT2 * cdiv_c((2 + NB) * C, 32) * 1024;
  static constexpr int WSLOT = AW ? max_c(WB_CV1, max_c(WB_M, WB_CV2)) : 0;
  // whole-image configurations: the entry conv runs before anything lives in the planes, so ALL of its fragments are staged
  // in LDS (144 KB for 64 -> 128 channels) and a wave holds every output channel of its pixels (NT = all row tiles): each
  // pixel fragment is gathered from global memory exactly once per workgroup.  With the weights streamed from L2 by every
  // wave this phase moved 2 MB through the CU's L1 and took 88 k of the kernel's 230 k cycles.
  static constexpr int NT_S2 = XC / 16, PT_S2 = 2;
  static constexpr int WB_S2 = PERIMG ? NT_S2 * 9 * (KS2 / 32) * 1024 : 0;
  static constexpr int LDS_BYTES = max_c(PLANE0 + PLANE + 2 * WSLOT, WB_S2);
  static_assert(!W1_LDS || WB_CV1 <= PLANE, "cv1's fragments are staged in plane 1");
  static C2fShape shape() {
    C2fShape s;
    s.C = C; s.NB = NB; s.KA = KA; s.KB = KB; s.UP = UP ? 1 : 0; s.COUT = COUT; s.MODE = MODE; s.KS2 = KS2;
    return s;
  }
};

struct Ctx {
  int n, oy0, ox0, H, W;
  int lane, wave, g, sig, gam;
  int dbg;                  // diagnostic flags (C2fArgs::debug_store): bit 1 = the stride-2 gather reads one cache-hot KiB
  unsigned long long* st;   // diagnostic stamps of this workgroup (null on the product path)
};
// in-phase stamp (wave 0's first lane), pinned between the surrounding instructions
#define C2F_ISTAMP(k)                                          \
  if (cx.st && threadIdx.x == 0) {                             \
    __builtin_amdgcn_sched_barrier(0);                         \
    cx.st[(k)] = clock64();                                    \
    __builtin_amdgcn_sched_barrier(0);                         \
  }
struct Rg {
  int gy0, gx0, fy0, fx0, rh, rw, R;
  unsigned magic;
};

#define C2F_STAMP(k) \
  if (a.stamps && threadIdx.x == 0) a.stamps[(size_t)blockIdx.x * 16 + (k)] = ((k) == 0 || (k) == 15) ? wall_clock64() : clock64();

// 16-byte LDS-DMA: lane l's 16 bytes land at lds + 16*l (wave-uniform lds), no VGPR in between
#define C2F_GLDS16(gptr, lptr)                                                                        \
  __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(gptr),          \
                                   (void __attribute__((address_space(3)))*)(lptr), 16, 0, 0)

// Phase boundary: every wave's global stores, LDS-DMA requests and LDS traffic are done, then the workgroup barrier.
// (NOT __syncthreads(): its workgroup-scope fence does not wait for vmcnt on this target, so staged weights could still
// be in flight when another wave reads the slot.)
__device__ __forceinline__ void wg_sync() {
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

__device__ __forceinline__ floatx4 mma16(half8 a, half8 b, floatx4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ half8 as_h8(u32x4 v) { return __builtin_bit_cast(half8, v); }

// x * sigmoid(x) of (v + b), the arithmetic of act4<T, ACT_SILU> in conv_kernels.hip (hardware exp2 / rcp, two-wide multiplies)
__device__ __forceinline__ floatx4 silu4(floatx4 v, floatx4 b) {
  v = v + b;
  float nl2e = -1.4426950408889634f;   // in an SGPR: two v_pk_mul_f32 instead of four v_mul_f32 with a literal (act4, conv_kernels.hip)
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

template <class CFG> __device__ __forceinline__ Rg make_region(const Ctx& cx, int E) {
  Rg r;
  const int y0 = max(cx.oy0 - E, 0), y1 = min(cx.oy0 + CFG::TH + E, cx.H);
  const int x0 = max(cx.ox0 - E, 0), x1 = min(cx.ox0 + CFG::TW + E, cx.W);
  r.gy0 = y0; r.gx0 = x0;
  r.rh = y1 - y0; r.rw = x1 - x0;
  r.R = r.rh * r.rw;
  r.fy0 = y0 - (cx.oy0 - CFG::F);
  r.fx0 = x0 - (cx.ox0 - CFG::F);
  r.magic = 0xFFFFFFFFu / (unsigned)r.rw + 1u;   // rw >= 20 (host-checked): p / rw == umulhi(p, magic) for p < 2^16
  return r;
}
// linear region pixel -> (row, column); p is clamped into the region by the caller
__device__ __forceinline__ void pix_of(const Rg& r, int p, int& py, int& px) {
  py = (int)__umulhi((unsigned)p, r.magic);
  px = p - py * r.rw;
}

// 4*NT consecutive halfs (this lane's channels of one pixel) -> memory
template <int NT> __device__ __forceinline__ void store_h(char* dst, const half_t (&h)[4 * NT]) {
  if constexpr (NT == 1) {
    half4 q;
#pragma unroll
    for (int i = 0; i < 4; ++i) q[i] = h[i];
    *reinterpret_cast<half4*>(dst) = q;
  } else if constexpr (NT % 2 == 1) {   // NT = 3: 24 bytes at a 24-byte stride -- 8-byte aligned pieces
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      half4 q;
#pragma unroll
      for (int i = 0; i < 4; ++i) q[i] = h[4 * j + i];
      *reinterpret_cast<half4*>(dst + 8 * j) = q;
    }
  } else {
#pragma unroll
    for (int j = 0; j < NT / 2; ++j) {
      half8 q;
#pragma unroll
      for (int i = 0; i < 8; ++i) q[i] = h[8 * j + i];
      *reinterpret_cast<half8*>(dst + 16 * j) = q;
    }
  }
}
template <int NT> __device__ __forceinline__ void load_h(const char* src, half_t (&h)[4 * NT]) {
  if constexpr (NT == 1) {
    const half4 q = *reinterpret_cast<const half4*>(src);
#pragma unroll
    for (int i = 0; i < 4; ++i) h[i] = q[i];
  } else if constexpr (NT % 2 == 1) {
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const half4 q = *reinterpret_cast<const half4*>(src + 8 * j);
#pragma unroll
      for (int i = 0; i < 4; ++i) h[4 * j + i] = q[i];
    }
  } else {
#pragma unroll
    for (int j = 0; j < NT / 2; ++j) {
      const half8 q = *reinterpret_cast<const half8*>(src + 16 * j);
#pragma unroll
      for (int i = 0; i < 8; ++i) h[8 * j + i] = q[i];
    }
  }
}
template <int NT> __device__ __forceinline__ void to_half(const floatx4 (&v)[NT], half_t (&h)[4 * NT]) {
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) h[4 * t + r] = (half_t)v[t][r];
}

// K loop of one block: acc[t][i] += A(step s, tile t) . B(step s, pixel tile i).  DA / DB register sets for the A / B
// fragments: step s + D - 1 is requested before the MFMAs of step s (static indices after unrolling); fix() runs on a
// step's B fragments right before they are consumed (border masks of the stride-2 gather: applied at the load they would
// make the wave wait for it at once).
template <int S, int DA, int DB, int NT, int PT, class LA, class LB, class FX>
__device__ __forceinline__ void kloop(floatx4 (&acc)[NT][PT], LA&& la, LB&& lb, FX&& fix) {
  half8 af[DA][NT], bf[DB][PT];
#pragma unroll
  for (int s = 0; s < DA - 1; ++s)
    if (s < S) la(s, af[s]);
#pragma unroll
  for (int s = 0; s < DB - 1; ++s)
    if (s < S) lb(s, bf[s]);
#pragma unroll
  for (int s = 0; s < S; ++s) {
    if (s + DA - 1 < S) la(s + DA - 1, af[(s + DA - 1) % DA]);
    if (s + DB - 1 < S) lb(s + DB - 1, bf[(s + DB - 1) % DB]);
    __builtin_amdgcn_sched_barrier(0);
    fix(s, bf[s % DB]);
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int i = 0; i < PT; ++i) acc[t][i] = mma16(af[s % DA][t], bf[s % DB][i], acc[t][i]);
  }
}
struct NoFix {
  template <int PT> __device__ __forceinline__ void operator()(int, half8 (&)[PT]) const {}
};

// Where a phase's A fragments ([channel block][K step][tile][lane][16 B]) come from: an LDS slot filled by stage_weights
// (tile configurations: every wave of the workgroup needs the same fragments, so they cross the CU's L1 once instead of
// eight times) or global memory / L2 (whole-image configurations, whose weights do not fit next to the planes).
template <bool LDSW> struct ASrc {
  const char* base;   // LDSW: LDS address of the slot; else the phase's fragments in global memory
  template <int NT> __device__ __forceinline__ void load(int off_bytes, int s, half8 (&af)[NT]) const {
#pragma unroll
    for (int t = 0; t < NT; ++t) af[t] = as_h8(*reinterpret_cast<const u32x4*>(base + off_bytes + (s * NT + t) * 1024));
  }
};
// request the `bytes` (a multiple of 1 KiB) of a phase's fragments into an LDS slot; the pieces are dealt to the waves
__device__ __forceinline__ void stage_weights(const Ctx& cx, const void* src, char* slot, int bytes, int nw) {
  const char* s = reinterpret_cast<const char*>(src) + cx.lane * 16;
  for (int p = cx.wave; p < (bytes >> 10); p += nw) C2F_GLDS16(s + p * 1024, slot + p * 1024);
}

// ---- 1x1 conv + SiLU.  K = KA channels from global srcA, then KB from global srcB, then KL0 from LDS plane 0 and KL1
//      from LDS plane 1 (tile pixels: cv2 over concat(y0 .., y_NB, y_{NB+1})).  UP: srcA is a half-resolution tensor read
//      at (y/2, x/2) (Interp nearest x2 fused, as conv1x1_mfma_kernel's UPS).
//      epi(cb, ok, py, px, v): this lane's 4*NT activated channels (block cb) of region pixel (py, px); ok = a real pixel.
template <class CFG, int NT, int CB, int PT, int KA, int KB, int KL0, int KL1, bool UP, bool LDSW, class EPI, int PSL0 = CFG::PS, int PSL1 = CFG::PS>
__device__ __forceinline__ void pw_phase(const Ctx& cx, const Rg& rg, const char* __restrict__ srcA, int pitchA, const char* __restrict__ srcB,
                                         int pitchB, const char* pl0, const char* pl1, const ASrc<LDSW>& wsrc, const float* __restrict__ bias,
                                         EPI&& epi, int stamp0 = -1) {
  static_assert(KA % 32 == 0 && KB % 8 == 0 && KL0 % 32 == 0 && KL1 % 16 == 0 && (KB % 32 == 0 || KL0 + KL1 == 0),
                "K segments: whole steps, except a tail of whole 8-channel groups at the very end (global srcB, or 16 channels of plane 1)");
  constexpr int SA = KA / 32, SB = cdiv_c(KB, 32), SL0 = KL0 / 32, SL1 = cdiv_c(KL1, 32), SG = SA + SB, S = SG + SL0 + SL1;
  // a last K step that is not full (KB % 32 != 0): lanes whose channel group lies past the pixel's channels re-read group 0
  // of the same pixel (their weights are zero) -- the address stays inside the pixel
  const int tail_off = (4 * (SB - 1) + cx.gam < KB / 8) ? (SB - 1) * 64 : -cx.gam * 16;
  const int npt = (rg.R + 15) >> 4;
  const int nblk = ((npt + PT - 1) / PT) * CB;
  for (int blk = cx.wave; blk < nblk; blk += CFG::NW) {
    const int cb = blk % CB, pbk = blk / CB;
    unsigned offA[SA > 0 ? PT : 1], offB[SB > 0 ? PT : 1];
    int pb[SL0 + SL1 > 0 ? PT : 1];   // pixel slot index (x 16-byte units are added per plane)
#pragma unroll
    for (int i = 0; i < PT; ++i) {
      int p = (pbk * PT + i) * 16 + cx.sig;
      p = p < rg.R ? p : rg.R - 1;
      int py, px;
      pix_of(rg, p, py, px);
      const int gy = rg.gy0 + py, gx = rg.gx0 + px;
      if constexpr (SA > 0) {
        const int ya = UP ? (gy >> 1) : gy, xa = UP ? (gx >> 1) : gx;
        const int HA = UP ? (cx.H >> 1) : cx.H, WA = UP ? (cx.W >> 1) : cx.W;
        offA[i] = (unsigned)(((cx.n * HA + ya) * WA + xa) * pitchA) * 2u + (unsigned)cx.gam * 16u;
      }
      if constexpr (SB > 0) offB[i] = (unsigned)(((cx.n * cx.H + gy) * cx.W + gx) * pitchB) * 2u + (unsigned)cx.gam * 16u;
      if constexpr (SL0 + SL1 > 0) pb[i] = (rg.fy0 + py) * CFG::LW + rg.fx0 + px;
    }
    floatx4 acc[NT][PT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int i = 0; i < PT; ++i) acc[t][i] = floatx4{0.f, 0.f, 0.f, 0.f};
    // (laundered: the fragment addresses are loop-invariant when CB == 1, and LICM then hoists EVERY weight load of the
    //  phase to the top of the kernel, where they are spilled one by one behind s_waitcnt vmcnt(0))
    int woff = (cb * S * NT * 64 + cx.lane) * 16;
    asm volatile("" : "+v"(woff));
    floatx4 bv[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) bv[t] = *reinterpret_cast<const floatx4*>(bias + cb * 16 * NT + 4 * NT * cx.g + 4 * t);
    // (whole-image cv1: 8 waves on the CU and a handful of K steps -- the pixel fragments of up to 5 steps ahead are in flight,
    //  at depth 3 the K loop ran at the L2 round trip: 18 B/clk for the CU)
    constexpr int DB = CFG::DEEP ? deep_depth(S, NT, PT, LDSW)
                                 : ((CFG::PERIMG && LDSW) ? min_c(S, 5) : (((NT * PT <= 20 && CFG::C >= 32) || (CFG::WPS == 1 && NT * PT <= 28)) ? 3 : 2));
    constexpr int DA = LDSW ? 2 : DB;
    if (stamp0 >= 0) { C2F_ISTAMP(stamp0) }
    kloop<S, DA, DB, NT, PT>(
        acc, [&](int s, half8(&af)[NT]) { wsrc.template load<NT>(woff, s, af); },
        [&](int s, half8(&bf)[PT]) {
#pragma unroll
          for (int i = 0; i < PT; ++i) {
            if (s < SA) bf[i] = as_h8(*reinterpret_cast<const u32x4*>(srcA + (size_t)offA[SA > 0 ? i : 0] + s * 64));
            else if (s < SG && KB % 32 != 0 && s == SG - 1) bf[i] = as_h8(*reinterpret_cast<const u32x4*>(srcB + (size_t)(offB[SB > 0 ? i : 0] + (unsigned)tail_off)));
            else if (s < SG) bf[i] = as_h8(*reinterpret_cast<const u32x4*>(srcB + (size_t)offB[SB > 0 ? i : 0] + (s - SA) * 64));
            else if (s < SG + SL0) bf[i] = as_h8(*reinterpret_cast<const u32x4*>(pl0 + pb[SL0 + SL1 > 0 ? i : 0] * PSL0 + cx.gam * 16 + (s - SG) * 64));
            // (a 16-channel plane 1: the lanes of K groups 2, 3 re-read groups 0, 1 -- their weights are zero)
            else bf[i] = as_h8(*reinterpret_cast<const u32x4*>(pl1 + pb[SL0 + SL1 > 0 ? i : 0] * PSL1 + (KL1 % 32 ? (cx.gam & 1) : cx.gam) * 16 + (s - SG - SL0) * 64));
          }
        },
        NoFix{});
    if (stamp0 >= 0) { C2F_ISTAMP(stamp0 + 1) }
#pragma unroll
    for (int i = 0; i < PT; ++i) {
      const int p0 = (pbk * PT + i) * 16 + cx.sig;
      const bool ok = p0 < rg.R;
      const int p = ok ? p0 : rg.R - 1;
      int py, px;
      pix_of(rg, p, py, px);
      floatx4 v[NT];
#pragma unroll
      for (int t = 0; t < NT; ++t) v[t] = silu4(acc[t][i], bv[t]);
      epi(cb, ok, py, px, v);
    }
    if (stamp0 >= 0) { C2F_ISTAMP(stamp0 + 2) }
  }
}

// ---- cv2 of the general-K configurations (c = 24 / 48: the concat's segments are not whole K steps).  K group q = 4 s + gam of
//      concat(y0 .. y_{NB+1}) comes from the concat buffer in global memory while q < K2G / 8 (y0 .. y_{NB-1}), from plane 0 for the
//      next c / 8 groups (y_NB) and from plane 1 for the last c / 8 (y_{NB+1}): a per-lane table of source and offset per step,
//      at most one step of a phase mixes global and LDS lanes.  The weights keep the plain channel order of pw_phase.  With every
//      y segment read back from the concat buffer (the first version) the c = 24 module on the 80 x 80 map moved 298 MB per launch,
//      118 MB of it these segments.
template <class CFG, int NT, int CB, int PT, bool LDSW, class EPI>
__device__ __forceinline__ void pw_gk_phase(const Ctx& cx, const Rg& rg, const char* __restrict__ cat, int cat_pitch, const char* pl0, const char* pl1,
                                            const ASrc<LDSW>& wsrc, const float* __restrict__ bias, EPI&& epi, int stamp0 = -1) {
  constexpr int G0 = CFG::K2G / 8, GC = CFG::C / 8, GT = G0 + 2 * GC, S = cdiv_c(GT, 4), PS = CFG::PS;
  static_assert(CFG::PS0 == CFG::PS && G0 >= 1, "one pixel pitch for both planes; y0 comes from the concat buffer");
  int goff[S];            // global lanes: byte offset of the lane's K group inside the pixel
  const char* lbase[S];   // LDS lanes: plane base + offset of the lane's K group inside the pixel
  bool isg[S];
#pragma unroll
  for (int s = 0; s < S; ++s) {
    int q = 4 * s + cx.gam;
    q = q < GT ? q : GT - 1;   // (the groups past the concat carry zero weights and re-read the last real one)
    isg[s] = q < G0;
    goff[s] = (q < G0 ? q : G0 - 1) * 16;
    const int ql = q < G0 ? 0 : q - G0;
    lbase[s] = ql < GC ? pl0 + ql * 16 : pl1 + (ql - GC) * 16;
  }
  const int npt = (rg.R + 15) >> 4;
  const int nblk = ((npt + PT - 1) / PT) * CB;
  for (int blk = cx.wave; blk < nblk; blk += CFG::NW) {
    const int cb = blk % CB, pbk = blk / CB;
    unsigned offB[PT];
    int pb[PT];
#pragma unroll
    for (int i = 0; i < PT; ++i) {
      int p = (pbk * PT + i) * 16 + cx.sig;
      p = p < rg.R ? p : rg.R - 1;
      int py, px;
      pix_of(rg, p, py, px);
      offB[i] = (unsigned)(((cx.n * cx.H + rg.gy0 + py) * cx.W + rg.gx0 + px) * cat_pitch) * 2u;
      pb[i] = ((rg.fy0 + py) * CFG::LW + rg.fx0 + px) * PS;
    }
    floatx4 acc[NT][PT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int i = 0; i < PT; ++i) acc[t][i] = floatx4{0.f, 0.f, 0.f, 0.f};
    int woff = (cb * S * NT * 64 + cx.lane) * 16;
    asm volatile("" : "+v"(woff));
    floatx4 bv[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) bv[t] = *reinterpret_cast<const floatx4*>(bias + cb * 16 * NT + 4 * NT * cx.g + 4 * t);
    constexpr int DB = deep_depth(S, NT, PT, LDSW);
    constexpr int DA = LDSW ? 2 : DB;
    if (stamp0 >= 0) { C2F_ISTAMP(stamp0) }
    kloop<S, DA, DB, NT, PT>(
        acc, [&](int s, half8(&af)[NT]) { wsrc.template load<NT>(woff, s, af); },
        [&](int s, half8(&bf)[PT]) {
          const bool pure_g = 4 * s + 3 < G0, pure_l = 4 * s >= G0;   // (compile-time after unrolling)
#pragma unroll
          for (int i = 0; i < PT; ++i) {
            if (pure_g) bf[i] = as_h8(*reinterpret_cast<const u32x4*>(cat + (size_t)offB[i] + goff[s]));
            else if (pure_l) bf[i] = as_h8(*reinterpret_cast<const u32x4*>(lbase[s] + pb[i]));
            else if (isg[s]) bf[i] = as_h8(*reinterpret_cast<const u32x4*>(cat + (size_t)offB[i] + goff[s]));
            else bf[i] = as_h8(*reinterpret_cast<const u32x4*>(lbase[s] + pb[i]));
          }
        },
        NoFix{});
    if (stamp0 >= 0) { C2F_ISTAMP(stamp0 + 1) }
#pragma unroll
    for (int i = 0; i < PT; ++i) {
      const int p0 = (pbk * PT + i) * 16 + cx.sig;
      const bool ok = p0 < rg.R;
      const int p = ok ? p0 : rg.R - 1;
      int py, px;
      pix_of(rg, p, py, px);
      floatx4 v[NT];
#pragma unroll
      for (int t = 0; t < NT; ++t) v[t] = silu4(acc[t][i], bv[t]);
      epi(cb, ok, py, px, v);
    }
    if (stamp0 >= 0) { C2F_ISTAMP(stamp0 + 2) }
  }
}

// ---- 1x1 conv + SiLU of the whole-image configurations whose epilogue stores into the planes its K loop reads (cv2 -> the
//      SPPF's input, SPPF.cv1 -> the pools' input): ONE round of blocks (every accumulator tile of the phase lives in the
//      eight waves' registers), a workgroup barrier between the K loops and the epilogues.  K = KB channels from global srcB
//      (whole steps), then KL0 from plane 0 and KL1 from plane 1; weights from L2.  With the plain pw_phase these two phases ran
//      two rounds of blocks and SPPF.cv1 read cv2's output back from global memory: 46 k cycles for 1600 MFMAs per wave-set.
template <class CFG, int NT, int CB, int PT, int KB, int KL0, int KL1, int DA, class EPI>
__device__ __forceinline__ void pw_sync_phase(const Ctx& cx, const Rg& rg, const char* __restrict__ srcB, int pitchB, const char* pl0, const char* pl1,
                                              const char* __restrict__ w, const float* __restrict__ bias, EPI&& epi, int stamp0 = -1) {
  static_assert(KB % 32 == 0 && KL0 % 32 == 0 && KL1 % 32 == 0, "whole K steps");
  constexpr int SB = KB / 32, SL0 = KL0 / 32, SL1 = KL1 / 32, S = SB + SL0 + SL1, PS = CFG::PS;
  static_assert(cdiv_c(CFG::npt(0), PT) * CB <= CFG::NW && NT * PT <= 28, "one round of blocks");
  const int npt = (rg.R + 15) >> 4;
  const int nblk = ((npt + PT - 1) / PT) * CB;
  const bool has = cx.wave < nblk;   // wave-uniform
  const int blk = has ? cx.wave : 0;
  const int cb = blk % CB, pbk = blk / CB;
  floatx4 acc[NT][PT];
  if (has) {
    unsigned offB[SB > 0 ? PT : 1];
    int pb[PT];
#pragma unroll
    for (int i = 0; i < PT; ++i) {
      int p = (pbk * PT + i) * 16 + cx.sig;
      p = p < rg.R ? p : rg.R - 1;
      int py, px;
      pix_of(rg, p, py, px);
      if constexpr (SB > 0) offB[i] = (unsigned)(((cx.n * cx.H + rg.gy0 + py) * cx.W + rg.gx0 + px) * pitchB) * 2u + (unsigned)cx.gam * 16u;
      pb[i] = ((rg.fy0 + py) * CFG::LW + rg.fx0 + px) * PS + cx.gam * 16;
    }
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int i = 0; i < PT; ++i) acc[t][i] = floatx4{0.f, 0.f, 0.f, 0.f};
    int woff = (cb * S * NT * 64 + cx.lane) * 16;
    asm volatile("" : "+v"(woff));
    if (stamp0 >= 0) { C2F_ISTAMP(stamp0) }
    kloop<S, DA, 2, NT, PT>(   // DA: weight fragments (L2) requested DA - 1 steps ahead
        acc,
        [&](int s_, half8(&af)[NT]) {
#pragma unroll
          for (int t = 0; t < NT; ++t) af[t] = as_h8(*reinterpret_cast<const u32x4*>(w + woff + (s_ * NT + t) * 1024));
        },
        [&](int s_, half8(&bf)[PT]) {
#pragma unroll
          for (int i = 0; i < PT; ++i) {
            if (s_ < SB) bf[i] = as_h8(*reinterpret_cast<const u32x4*>(srcB + (size_t)offB[SB > 0 ? i : 0] + s_ * 64));
            else if (s_ < SB + SL0) bf[i] = as_h8(*reinterpret_cast<const u32x4*>(pl0 + pb[i] + (s_ - SB) * 64));
            else bf[i] = as_h8(*reinterpret_cast<const u32x4*>(pl1 + pb[i] + (s_ - SB - SL0) * 64));
          }
        },
        NoFix{});
    if (stamp0 >= 0) { C2F_ISTAMP(stamp0 + 1) }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  if (stamp0 >= 0) { C2F_ISTAMP(stamp0 + 2) }
  if (has) {
    floatx4 bv[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) bv[t] = *reinterpret_cast<const floatx4*>(bias + cb * 16 * NT + 4 * NT * cx.g + 4 * t);
#pragma unroll
    for (int i = 0; i < PT; ++i) {
      const int p0 = (pbk * PT + i) * 16 + cx.sig;
      const bool ok = p0 < rg.R;
      const int p = ok ? p0 : rg.R - 1;
      int py, px;
      pix_of(rg, p, py, px);
      floatx4 v[NT];
#pragma unroll
      for (int t = 0; t < NT; ++t) v[t] = silu4(acc[t][i], bv[t]);
      epi(cb, ok, py, px, v);
    }
    if (stamp0 >= 0) { C2F_ISTAMP(stamp0 + 3) }
  }
}

// ---- 3x3 stride-1 conv + SiLU, LDS plane -> epilogue.  K walks (tap, 32-channel block); c = 16: a K step covers two taps
//      (an even number of pixels apart: the conflict-free pairing of the header) x two channel groups.
//      One block per wave (the block shapes guarantee it).  SYNC: a workgroup barrier between the K loop and the epilogue
//      (the epilogue overwrites the plane the K loop reads).
template <class CFG, int PT, bool SYNC, bool LDSW, int PS, int CHOFF, class EPI>
__device__ __forceinline__ void c3_phase(const Ctx& cx, const Rg& rg, const char* pin, const ASrc<LDSW>& wsrc, const float* __restrict__ bias,
                                         EPI&& epi) {
  // PS: pixel pitch of the input plane; CHOFF: byte offset of the conv's input channels inside a pixel
  constexpr int C = CFG::C, NT = CFG::NTM, CB = CFG::CBM, LW = CFG::LW;
  constexpr int SPT = C >= 32 ? C / 32 : 1;
  constexpr int S = CFG::c3_steps;
  constexpr bool GK = CFG::GK;
  // c = 16: taps of step s for the lanes with (g & 1) == 0 / 1
  constexpr int TA[5] = {0, 3, 6, 1, 4}, TB[5] = {2, 5, 8, 7, 4};
  int soff[GK ? S : (C >= 32 ? 1 : 5)];
  if constexpr (GK) {
    // general packing: K group q = 4 s + gam is channel group q % G of tap q / G (G = c / 8 groups per tap); the groups past
    // the last tap carry zero weights and re-read the last real one
    constexpr int G = C / 8;
#pragma unroll
    for (int s = 0; s < S; ++s) {
      int q = 4 * s + cx.gam;
      q = q < 9 * G ? q : 9 * G - 1;
      const int tap = q / G, cg = q - tap * G;
      const int ty = (tap * 11) >> 5;   // tap / 3 for tap < 9
      soff[s] = (ty * LW + tap - 3 * ty) * PS + cg * 16;
    }
  } else if constexpr (C < 32) {
#pragma unroll
    for (int s = 0; s < 5; ++s) {
      const int ta = ((TA[s] / 3) * LW + TA[s] % 3) * PS, tb = ((TB[s] / 3) * LW + TB[s] % 3) * PS;
      soff[s] = (cx.g & 1) ? tb : ta;
    }
  }
  const int npt = (rg.R + 15) >> 4;
  const int nblk = ((npt + PT - 1) / PT) * CB;
  const bool has = cx.wave < nblk;   // wave-uniform
  const int blk = has ? cx.wave : 0;
  const int cb = blk % CB, pbk = blk / CB;
  floatx4 acc[NT][PT];
  floatx4 bv[NT];
  if (has) {
    int pb[PT];
#pragma unroll
    for (int i = 0; i < PT; ++i) {
      int p = (pbk * PT + i) * 16 + cx.sig;
      p = p < rg.R ? p : rg.R - 1;
      int py, px;
      pix_of(rg, p, py, px);
      const int slot = (rg.fy0 + py) * LW + rg.fx0 + px;
      pb[i] = (slot - LW - 1) * PS + CHOFF + (GK ? 0 : (C >= 32 ? cx.gam * 16 : (cx.g >> 1) * 16));
    }
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int i = 0; i < PT; ++i) acc[t][i] = floatx4{0.f, 0.f, 0.f, 0.f};
    int woff = (cb * S * NT * 64 + cx.lane) * 16;
    asm volatile("" : "+v"(woff));
#pragma unroll
    for (int t = 0; t < NT; ++t) bv[t] = *reinterpret_cast<const floatx4*>(bias + cb * 16 * NT + 4 * NT * cx.g + 4 * t);
    kloop<S, (LDSW ? 2 : 4), 2, NT, PT>(
        acc, [&](int s, half8(&af)[NT]) { wsrc.template load<NT>(woff, s, af); },
        [&](int s, half8(&bf)[PT]) {
          if constexpr (C >= 32 && !GK) {
            const int tap = s / SPT, cblk = s % SPT;
            const int off = ((tap / 3) * LW + tap % 3) * PS + cblk * 64;
#pragma unroll
            for (int i = 0; i < PT; ++i) bf[i] = as_h8(*reinterpret_cast<const u32x4*>(pin + pb[i] + off));
          } else {
#pragma unroll
            for (int i = 0; i < PT; ++i) bf[i] = as_h8(*reinterpret_cast<const u32x4*>(pin + pb[i] + soff[s]));
          }
        },
        NoFix{});
  }
  if constexpr (SYNC) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  }
  if (has) {
#pragma unroll
    for (int i = 0; i < PT; ++i) {
      const int p0 = (pbk * PT + i) * 16 + cx.sig;
      const bool ok = p0 < rg.R;
      const int p = ok ? p0 : rg.R - 1;
      int py, px;
      pix_of(rg, p, py, px);
      floatx4 v[NT];
#pragma unroll
      for (int t = 0; t < NT; ++t) v[t] = silu4(acc[t][i], bv[t]);
      epi(cb, ok, py, px, v);
    }
  }
}

// ---- 3x3 stride-2 pad-1 conv + SiLU, pixel operand gathered from global memory (whole-image configurations: the region is
//      the image).  Taps above / left of the image read the centre tap's address instead and are zeroed before use.
template <class CFG, int NT, int CB, int PT, bool LDSW, class EPI>
__device__ __forceinline__ void c3s2_phase(const Ctx& cx, const Rg& rg, const char* __restrict__ src, int pitch, const ASrc<LDSW>& wsrc,
                                           const float* __restrict__ bias, EPI&& epi) {
  // (diagnostic stamps 8 / 9 / 10: wave 0's first block -- K loop start, K loop end, epilogue end)
  constexpr int SPT = CFG::KS2 / 32, S = 9 * SPT;
  const int H2 = 2 * cx.H, W2 = 2 * cx.W;
  const int pixb = pitch * 2, rowb = W2 * pixb;
  const int npt = (rg.R + 15) >> 4;
  const int nblk = ((npt + PT - 1) / PT) * CB;
  for (int blk = cx.wave; blk < nblk; blk += CFG::NW) {
    const int cb = blk % CB, pbk = blk / CB;
    unsigned base[PT];
    int fl[PT];
#pragma unroll
    for (int i = 0; i < PT; ++i) {
      int p = (pbk * PT + i) * 16 + cx.sig;
      p = p < rg.R ? p : rg.R - 1;
      int py, px;
      pix_of(rg, p, py, px);
      const int gy = rg.gy0 + py, gx = rg.gx0 + px;
      base[i] = (unsigned)(((cx.n * H2 + 2 * gy) * W2 + 2 * gx) * pitch) * 2u + (unsigned)cx.gam * 16u;
      fl[i] = (gy == 0 ? 1 : 0) | (gx == 0 ? 2 : 0);
      if (cx.dbg & 2) { base[i] = (unsigned)(rowb + pixb) + (unsigned)cx.lane * 16u; fl[i] = 0; }
    }
    floatx4 acc[NT][PT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int i = 0; i < PT; ++i) acc[t][i] = floatx4{0.f, 0.f, 0.f, 0.f};
    int woff = (cb * S * NT * 64 + cx.lane) * 16;
    asm volatile("" : "+v"(woff));
    floatx4 bv[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) bv[t] = *reinterpret_cast<const floatx4*>(bias + cb * 16 * NT + 4 * NT * cx.g + 4 * t);
    constexpr int D = (LDSW && PT <= 2) ? 5 : ((NT * PT <= 20 || (CFG::WPS == 1 && NT * PT <= 28)) ? 3 : 2);
    if (blk == 0) { C2F_ISTAMP(8) }
    kloop<S, (LDSW ? 2 : D), D, NT, PT>(
        acc, [&](int s, half8(&af)[NT]) { wsrc.template load<NT>(woff, s, af); },
        [&](int s, half8(&bf)[PT]) {
          const int tap = s / SPT, cblk = s % SPT;
          const int dy = tap / 3, dx = tap % 3;
          const int toff = (dy - 1) * rowb + (dx - 1) * pixb;
#pragma unroll
          for (int i = 0; i < PT; ++i) {
            const bool inv = (dy == 0 && (fl[i] & 1)) || (dx == 0 && (fl[i] & 2));
            const unsigned o = base[i] + (unsigned)(inv ? 0 : toff);
            bf[i] = as_h8(*reinterpret_cast<const u32x4*>(src + (size_t)o + cblk * 64));
          }
        },
        [&](int s, half8(&bf)[PT]) {
          const int tap = s / SPT;
          const int dy = tap / 3, dx = tap % 3;
          if (dy == 0 || dx == 0) {
#pragma unroll
            for (int i = 0; i < PT; ++i) {
              const bool inv = (dy == 0 && (fl[i] & 1)) || (dx == 0 && (fl[i] & 2));
              const u32x4 m = __builtin_bit_cast(u32x4, bf[i]);
              bf[i] = as_h8(inv ? u32x4{0u, 0u, 0u, 0u} : m);
            }
          }
        });
    if (blk == 0) { C2F_ISTAMP(9) }
#pragma unroll
    for (int i = 0; i < PT; ++i) {
      const int p0 = (pbk * PT + i) * 16 + cx.sig;
      const bool ok = p0 < rg.R;
      const int p = ok ? p0 : rg.R - 1;
      int py, px;
      pix_of(rg, p, py, px);
      floatx4 v[NT];
#pragma unroll
      for (int t = 0; t < NT; ++t) v[t] = silu4(acc[t][i], bv[t]);
      epi(cb, ok, py, px, v);
    }
    if (blk == 0) { C2F_ISTAMP(10) }
  }
}

// LDS-only phase boundary (no global traffic of this wave has to be complete)
__device__ __forceinline__ void lds_sync() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

// ---- 5x5 stride-1 pad-2 max pool of the whole image (SPPF, model.ncnn.param:79-83), separable: a horizontal 5-max from
//      plane `pio` into plane `ptmp`, then a vertical 5-max back into `pio` (the pooled map replaces its input).  Window
//      positions outside the image are clamped onto the border pixel: it is inside the window anyway, so the maximum is the
//      same and the loops are branch-free.  gdst != null (bisect aid): the pooled map also goes to global memory.
template <class CFG>
__device__ __forceinline__ void pool_phase(const Ctx& cx, char* pio, char* ptmp, char* gdst, int gpitch) {
  constexpr int CG = CFG::C / 8, LW = CFG::LW, PS = CFG::PS, F = CFG::F, TH = CFG::TH, TW = CFG::TW;
  static_assert(TH == 20 && TW == 20, "pool strips: two strips of 10 outputs per line");
  // One thread per (line, channel group, half line): its ten 5-maxima from shared pair maxima -- p[j] = max(v[j], v[j+1]),
  // out[i] = max(p[i], p[i+2], v[i+4]) -- with the window positions clamped onto the line: 18 loads and 36 vector maxima for 10
  // outputs (a window per output: 50 loads with their index arithmetic and 40 maxima -- that form spent 5 k cycles per pass,
  // three quarters of the SPPF tail).
  auto strip = [&](const char* __restrict__ src, char* __restrict__ dst, int step) {   // position 0 of the line; bytes between positions
    const int half = (threadIdx.x / (20 * CG)) & 1;   // (only threads < 2 * 20 * CG get here)
    const int x0 = 10 * half;
    // two sub-strips of five outputs (nine loads each): the SPPF tail's accumulators stay live across the pools, and
    // fourteen loaded vectors beside them spilled
#pragma unroll
    for (int sub = 0; sub < 2; ++sub) {
      half8 v[9];
#pragma unroll
      for (int j = 0; j < 9; ++j) {
        const int x = min(max(x0 + 5 * sub - 2 + j, 0), 19);
        v[j] = *reinterpret_cast<const half8*>(src + x * step);
      }
      half8 pm[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) pm[j] = __builtin_elementwise_max(v[j], v[j + 1]);
#pragma unroll
      for (int i = 0; i < 5; ++i)
        *reinterpret_cast<half8*>(dst + (x0 + 5 * sub + i) * step) = __builtin_elementwise_max(__builtin_elementwise_max(pm[i], pm[i + 2]), v[i + 4]);
    }
  };
  const int t = threadIdx.x;
  const bool work = t < 2 * 20 * CG;
  const int line = (t % (20 * CG)) / CG, cg = t % CG;
  if (work) strip(pio + ((line + F) * LW + F) * PS + cg * 16, ptmp + ((line + F) * LW + F) * PS + cg * 16, PS);   // along x
  lds_sync();
  if (work) strip(ptmp + (F * LW + line + F) * PS + cg * 16, pio + (F * LW + line + F) * PS + cg * 16, LW * PS);   // along y
  lds_sync();
  if (gdst) {   // bisect aid: the pooled map also goes to global memory
    for (int it = threadIdx.x; it < TH * TW * CG; it += CFG::NW * 64) {
      const int pix = it / CG, g8 = it - pix * CG;
      const int y = pix / TW, x = pix - y * TW;
      *reinterpret_cast<half8*>(gdst + (size_t)(((cx.n * cx.H + y) * cx.W + x) * gpitch) * 2 + g8 * 16) =
          *reinterpret_cast<const half8*>(pio + ((y + F) * LW + x + F) * PS + g8 * 16);
    }
  }
}

// ---- SPPF.cv2 over concat(s, p1, p2, p3) without the concat: out = silu(W0 s + W1 p1 + W2 p2 + W3 p3 + b), the four
//      K segments accumulated as the pooled maps appear in plane 0 (each pool replaces its input there), so p1 .. p3 never
//      leave the CU and the conv's pixel operand comes from LDS.  One block per wave, accumulators live across the pools;
//      the next segment's weight fragments are requested before the pool that precedes it.
template <class CFG, class EPI>
__device__ __forceinline__ void sppf_tail(const Ctx& cx, const Rg& rg, char* P0, char* P1, const char* __restrict__ w, const float* __restrict__ bias,
                                          char* gcat2, int gpitch, EPI&& epi) {
  constexpr int C = CFG::C, NT = CFG::NT2, CB = CFG::CB2, PT = cdiv_c(CFG::npt(0), CFG::NW / CB), SPT = C / 32, S = 4 * SPT;
  static_assert(NT * PT <= 52 && SPT == 2, "SPPF tail: block shape");
  const int npt = (rg.R + 15) >> 4;
  const int nblk = ((npt + PT - 1) / PT) * CB;
  const bool has = cx.wave < nblk;
  const int blk = has ? cx.wave : 0;
  const int cb = blk % CB, pbk = blk / CB;
  int pb[PT];
#pragma unroll
  for (int i = 0; i < PT; ++i) {
    int p = (pbk * PT + i) * 16 + cx.sig;
    p = p < rg.R ? p : rg.R - 1;
    int py, px;
    pix_of(rg, p, py, px);
    pb[i] = ((rg.fy0 + py) * CFG::LW + rg.fx0 + px) * CFG::PS + cx.gam * 16;
  }
  floatx4 acc[NT][PT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int i = 0; i < PT; ++i) acc[t][i] = floatx4{0.f, 0.f, 0.f, 0.f};
  int woff = (cb * S * NT * 64 + cx.lane) * 16;
  asm volatile("" : "+v"(woff));
  floatx4 bv[NT];   // requested behind the last pool (live across the pools it would cost the strips their registers)
  half8 af[SPT][NT];
  auto load_a = [&](int seg) {
#pragma unroll
    for (int s = 0; s < SPT; ++s)
#pragma unroll
      for (int t = 0; t < NT; ++t) af[s][t] = as_h8(*reinterpret_cast<const u32x4*>(w + woff + ((seg * SPT + s) * NT + t) * 1024));
  };
  load_a(0);
#pragma unroll
  for (int seg = 0; seg < 4; ++seg) {
    if (seg > 0) pool_phase<CFG>(cx, P0, P1, gcat2 ? gcat2 + seg * C * 2 : nullptr, gpitch);
    if (seg == 3) {
#pragma unroll
      for (int t = 0; t < NT; ++t) bv[t] = *reinterpret_cast<const floatx4*>(bias + cb * 16 * NT + 4 * NT * cx.g + 4 * t);
    }
    if (has) {
#pragma unroll
      for (int s = 0; s < SPT; ++s) {
        half8 bf[PT];
#pragma unroll
        for (int i = 0; i < PT; ++i) bf[i] = as_h8(*reinterpret_cast<const u32x4*>(P0 + pb[i] + s * 64));
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
          for (int i = 0; i < PT; ++i) acc[t][i] = mma16(af[s][t], bf[i], acc[t][i]);
        __builtin_amdgcn_sched_barrier(0);
      }
      if (seg < 3) {
        __builtin_amdgcn_sched_barrier(0);
        load_a(seg + 1);   // flies during the pool
      }
    }
    if (seg < 3) lds_sync();   // every wave has read plane 0 before the pool overwrites it
  }
  if (has) {
#pragma unroll
    for (int i = 0; i < PT; ++i) {
      const int p0 = (pbk * PT + i) * 16 + cx.sig;
      const bool ok = p0 < rg.R;
      const int p = ok ? p0 : rg.R - 1;
      int py, px;
      pix_of(rg, p, py, px);
      floatx4 v[NT];
#pragma unroll
      for (int t = 0; t < NT; ++t) v[t] = silu4(acc[t][i], bv[t]);
      epi(cb, ok, py, px, v);
    }
  }
}

template <class CFG>
__global__ __launch_bounds__(CFG::NW * 64, CFG::WPS) void c2f_kernel(const C2fArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int C = CFG::C, NB = CFG::NB, F = CFG::F, LW = CFG::LW, PS = CFG::PS, TH = CFG::TH, TW = CFG::TW;
  constexpr bool AW = CFG::AW;
  constexpr int PS0 = CFG::PS0, Y1OFF = CFG::Y1OFF;
  char* P0 = smem;
  char* P1 = smem + CFG::PLANE0;
  char* WS = smem + CFG::PLANE0 + CFG::PLANE;   // two weight slots (AW)
  Ctx cx;
  cx.lane = threadIdx.x & 63;
  cx.wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  cx.g = cx.lane >> 4;
  {
    const int col = cx.lane & 15;
    cx.sig = col < 4 ? 2 * col : (col < 12 ? 2 * (col - 4) + 1 : 2 * (col - 8));
    cx.gam = ((cx.g & 1) << 1) | (cx.g >> 1);
  }
  cx.n = blockIdx.x % a.N;   // image fastest: the tiles of an image (which share halo pixels) meet in one XCD's L2
  const int tile = blockIdx.x / a.N;
  const int ty = tile / a.tiles_x, tx = tile - ty * a.tiles_x;
  cx.oy0 = ty * TH; cx.ox0 = tx * TW;
  cx.H = a.H; cx.W = a.W;
  cx.st = a.stamps ? a.stamps + (size_t)blockIdx.x * 16 : nullptr;
  cx.dbg = a.debug_store;
  C2F_STAMP(0)
  C2F_STAMP(1)

  // weight slots: phase q (0 = cv1, 1 = a_0, 2 = b_0, .., 2 NB + 1 = cv2) uses slot q & 1; its fragments are requested at
  // the start of phase q - 1, right behind the barrier that retired the slot's previous user, and have landed by the next
  // barrier (__syncthreads waits for the issuing wave's vmcnt)
  auto wslot = [&](int q) { return WS + (q & 1) * CFG::WSLOT; };
  auto stage = [&](int q) {
    if constexpr (AW) {
      if (q == 0) { if constexpr (!CFG::XCV1) stage_weights(cx, a.w[C2F_W_CV1], wslot(0), CFG::WB_CV1, CFG::NW); }
      else if (q == 2 * NB + 1) stage_weights(cx, a.w[C2F_W_CV2], wslot(q), CFG::WB_CV2, CFG::NW);
      else if (q <= 2 * NB) stage_weights(cx, a.w[C2F_W_A0 + q - 1], wslot(q), CFG::WB_M, CFG::NW);
    }
  };
  auto wsrc = [&](int q, int widx) {
    return ASrc<AW>{AW ? wslot(q) : reinterpret_cast<const char*>(a.w[widx])};
  };
  stage(0);
  stage(1);
  const char* src1 = reinterpret_cast<const char*>(a.src1);
  char* cat = reinterpret_cast<char*>(a.cat);
  const bool dbg = (a.debug_store & 1) != 0;

  // ---- [s2] entry conv -> x (global); its weights occupy the (still unused) planes
  if constexpr (CFG::MODE >= 1) {
    stage_weights(cx, a.w[C2F_W_S2], smem, CFG::WB_S2, CFG::NW);
    const Rg rg = make_region<CFG>(cx, 0);
    char* xo = reinterpret_cast<char*>(a.x);
    wg_sync();
    c3s2_phase<CFG, CFG::NT_S2, 1, CFG::PT_S2, true>(cx, rg, reinterpret_cast<const char*>(a.s2_in), a.s2_pitch, ASrc<true>{smem}, a.b[C2F_W_S2],
                                                     [&](int cb, bool ok, int py, int px, const floatx4(&v)[CFG::NT_S2]) {
                                                       constexpr int NT = CFG::NT_S2;
                                                       const int chb = cb * 16 * NT + 4 * NT * cx.g;
                                                       half_t h[4 * NT];
                                                       to_half<NT>(v, h);
                                                       const int gpix = (cx.n * cx.H + rg.gy0 + py) * cx.W + rg.gx0 + px;
                                                       if (ok) store_h<NT>(xo + (size_t)((unsigned)gpix * (unsigned)a.x_pitch) * 2 + chb * 2, h);
                                                     });
    wg_sync();
  }
  // pixels outside the image must read as zero (the convs' padding): they are never written, so clear the planes once.
  // Interior tiles write every pixel a later phase reads.
  {
    const bool interior = !CFG::PERIMG && cx.oy0 >= F && cx.ox0 >= F && cx.oy0 + TH + F <= a.H && cx.ox0 + TW + F <= a.W;
    if constexpr (CFG::W1_LDS) stage_weights(cx, a.w[C2F_W_CV1], P1, CFG::WB_CV1, CFG::NW);   // plane 1 is cleared behind cv1
    if (!interior) {
      constexpr int NCLR = CFG::W1_LDS ? CFG::PLANE0 : CFG::PLANE0 + CFG::PLANE;
      for (int i = threadIdx.x; i < NCLR / 16; i += CFG::NW * 64) reinterpret_cast<u32x4*>(smem)[i] = u32x4{0u, 0u, 0u, 0u};
    }
  }
  wg_sync();   // staged weights landed, planes cleared (cv1's epilogue stores into plane 0)
  C2F_STAMP(2)

  // ---- cv1 -> y0 | y1: y1 into plane 0 (whole region); into the concat buffer (tile pixels) whatever cv2 reads from there
  if constexpr (CFG::XCV1) {   // (cv1 ran in the launch in front: copy y1 of the region out of the concat buffer, every load in flight)
    const Rg rg = make_region<CFG>(cx, CFG::e_cv1);
    constexpr int CG = C / 8, NTHR = CFG::NW * 64, NPC = cdiv_c(LW * CFG::LH * CG, NTHR);
    u32x4 v[NPC];
    int dst[NPC];
#pragma unroll
    for (int j = 0; j < NPC; ++j) {
      const int it = threadIdx.x + j * NTHR;
      const int pix = it / CG, cg = it - pix * CG;
      const bool valid = pix < rg.R;
      int py, px;
      pix_of(rg, valid ? pix : rg.R - 1, py, px);
      const int gpix = (cx.n * cx.H + rg.gy0 + py) * cx.W + rg.gx0 + px;
      v[j] = *reinterpret_cast<const u32x4*>(cat + (size_t)((unsigned)gpix * (unsigned)a.cat_pitch) * 2 + C * 2 + cg * 16);
      dst[j] = valid ? ((rg.fy0 + py) * LW + rg.fx0 + px) * PS0 + Y1OFF + cg * 16 : -1;
    }
#pragma unroll
    for (int j = 0; j < NPC; ++j)
      if (dst[j] >= 0) *reinterpret_cast<u32x4*>(P0 + dst[j]) = v[j];
  } else {
    const Rg rg = make_region<CFG>(cx, CFG::e_cv1);
    pw_phase<CFG, CFG::NT1, CFG::CB1, CFG::PT1, CFG::KA, CFG::KB, 0, 0, CFG::UP, AW || CFG::W1_LDS>(
        cx, rg, reinterpret_cast<const char*>(a.src0), a.pitch0, src1, a.pitch1, nullptr, nullptr,
        ASrc < AW || CFG::W1_LDS > {CFG::W1_LDS ? P1 : (AW ? wslot(0) : reinterpret_cast<const char*>(a.w[C2F_W_CV1]))}, a.b[C2F_W_CV1],
        [&](int cb, bool ok, int py, int px, const floatx4(&v)[CFG::NT1]) {
          constexpr int NT = CFG::NT1;
          const int chb = cb * 16 * NT + 4 * NT * cx.g;
          half_t h[4 * NT];
          to_half<NT>(v, h);
          const int fy = rg.fy0 + py, fx = rg.fx0 + px;
          if (ok) {
            if (CFG::Y01) store_h<NT>(P0 + (fy * LW + fx) * PS0 + chb * 2, h);   // y0 | y1 side by side
            else if (chb >= C) store_h<NT>(P0 + (fy * LW + fx) * PS0 + (chb - C) * 2, h);
            if ((chb < CFG::K2G || dbg) && fy >= F && fy < F + TH && fx >= F && fx < F + TW) {
              const int gpix = (cx.n * cx.H + rg.gy0 + py) * cx.W + rg.gx0 + px;
              store_h<NT>(cat + (size_t)((unsigned)gpix * (unsigned)a.cat_pitch) * 2 + chb * 2, h);
            }
          }
        }, CFG::PERIMG ? -1 : 8);
  }
  wg_sync();
  if constexpr (CFG::W1_LDS) {   // plane 1 held cv1's fragments: now its zero ring
    for (int i = threadIdx.x; i < CFG::PLANE / 16; i += CFG::NW * 64) reinterpret_cast<u32x4*>(P1)[i] = u32x4{0u, 0u, 0u, 0u};
    lds_sync();
  }
  C2F_STAMP(3)

  // ---- bottlenecks: a_k: plane 0 -> plane 1; b_k: plane 1 -> y_{k+2} = y_{k+1} + ..
  //      (not the last: in place in plane 0; the last: into plane 1 behind a barrier when cv2 reads the planes)
#pragma unroll
  for (int k = 0; k < NB; ++k) {
    stage(2 * k + 2);
    {
      const Rg rg = make_region<CFG>(cx, CFG::e_a(k));
      auto epi_a = [&](int cb, bool ok, int py, int px, const floatx4(&v)[CFG::NTM]) {
        constexpr int NT = CFG::NTM;
        const int chb = cb * 16 * NT + 4 * NT * cx.g;
        half_t h[4 * NT];
        to_half<NT>(v, h);
        // (c % 16 != 0: the last lane group owns the padding rows of the second channel tile -- nothing to store)
        if (ok && (C % 16 == 0 || chb < C)) store_h<NT>(P1 + ((rg.fy0 + py) * LW + rg.fx0 + px) * PS + chb * 2, h);
      };
      const ASrc<AW> wa = wsrc(2 * k + 1, C2F_W_A0 + 2 * k);
      const float* ba = a.b[C2F_W_A0 + 2 * k];
      if (k == 0) c3_phase<CFG, CFG::ptm(CFG::e_a(0)), false, AW, PS0, Y1OFF>(cx, rg, P0, wa, ba, epi_a);
      else c3_phase<CFG, CFG::ptm(CFG::e_a(NB - 1)), false, AW, PS0, Y1OFF>(cx, rg, P0, wa, ba, epi_a);
    }
    wg_sync();
    stage(2 * k + 3);
    {
      const Rg rg = make_region<CFG>(cx, CFG::e_b(k));
      const bool last = k == NB - 1;
      auto epi_b = [&](int cb, bool ok, int py, int px, const floatx4(&v)[CFG::NTM]) {
        constexpr int NT = CFG::NTM;
        const int chb0 = cb * 16 * NT + 4 * NT * cx.g;
        const bool real = C % 16 == 0 || chb0 < C;   // (padding rows: see epi_a)
        const int chb = real ? chb0 : 0;
        const int fy = rg.fy0 + py, fx = rg.fx0 + px;
        char* yp = P0 + (fy * LW + fx) * PS0 + Y1OFF + chb * 2;
        half_t r[4 * NT], h[4 * NT];
        load_h<NT>(yp, r);
        // silu rounded to fp16, shortcut added in fp32, rounded again: what a stored conv output + an add kernel give
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
          for (int q = 0; q < 4; ++q) h[4 * t + q] = (half_t)((float)(half_t)v[t][q] + (float)r[4 * t + q]);
        if (ok && real) {
          if (!last) store_h<NT>(yp, h);
          else if (CFG::CV2_LDS) store_h<NT>(P1 + (fy * LW + fx) * PS + chb * 2, h);
          if (((2 + k) * C < CFG::K2G || dbg) && fy >= F && fy < F + TH && fx >= F && fx < F + TW) {
            const int gpix = (cx.n * cx.H + rg.gy0 + py) * cx.W + rg.gx0 + px;
            store_h<NT>(cat + (size_t)((unsigned)gpix * (unsigned)a.cat_pitch) * 2 + ((2 + k) * C + chb) * 2, h);
          }
        }
      };
      const ASrc<AW> wbk = wsrc(2 * k + 2, C2F_W_B0 + 2 * k);
      const float* bbk = a.b[C2F_W_B0 + 2 * k];
      if (k == 0 && NB > 1) c3_phase<CFG, CFG::ptm(CFG::e_b(0)), false, AW, PS, 0>(cx, rg, P1, wbk, bbk, epi_b);
      else c3_phase<CFG, CFG::ptm(CFG::e_b(NB - 1)), CFG::CV2_LDS, AW, PS, 0>(cx, rg, P1, wbk, bbk, epi_b);
    }
    wg_sync();
  }
  C2F_STAMP(4)

  // ---- cv2 over concat(y0 .. y_{NB+1}): the leading segments from the concat buffer (this workgroup's own stores,
  //      published by the barriers), y_NB and y_{NB+1} from the planes (c >= 32) -> out
  {
    const Rg rg = make_region<CFG>(cx, 0);
    char* out = reinterpret_cast<char*>(a.out);
    auto epi_cv2 = [&](int cb, bool ok, int py, int px, const floatx4(&v)[CFG::NT2]) {
          constexpr int NT = CFG::NT2;
          const int chb = cb * 16 * NT + 4 * NT * cx.g;
          half_t h[4 * NT];
          to_half<NT>(v, h);
          const int gpix = (cx.n * cx.H + rg.gy0 + py) * cx.W + rg.gx0 + px;
          if (ok) store_h<NT>(out + (size_t)((unsigned)gpix * (unsigned)a.out_pitch) * 2 + chb * 2, h);
        };
    // K segments: [concat buffer, K2G channels] [plane 0] [plane 1]; Y01: plane 0 = y0 | y1 (2c), plane 1 = y2 (c)
    if constexpr (CFG::PERIMG) {
      // one round of blocks; MODE 2: the output (2c = COUT channels) also replaces y_NB | y_{NB+1} in the planes -- SPPF.cv1's input
      static_assert(CFG::K2G % 32 == 0 && CFG::CV2_LDS && (CFG::MODE != 2 || CFG::COUT == 2 * C), "whole-image cv2");
      pw_sync_phase<CFG, CFG::NT2, CFG::CB2, CFG::PT2W, CFG::K2G, C, C, 2>(
          cx, rg, cat, a.cat_pitch, P0, P1, reinterpret_cast<const char*>(a.w[C2F_W_CV2]), a.b[C2F_W_CV2],
          [&](int cb, bool ok, int py, int px, const floatx4(&v)[CFG::NT2]) {
            constexpr int NT = CFG::NT2;
            const int chb = cb * 16 * NT + 4 * NT * cx.g;
            half_t h[4 * NT];
            to_half<NT>(v, h);
            const int gpix = (cx.n * cx.H + rg.gy0 + py) * cx.W + rg.gx0 + px;
            if (ok) {
              store_h<NT>(out + (size_t)((unsigned)gpix * (unsigned)a.out_pitch) * 2 + chb * 2, h);
              if constexpr (CFG::MODE == 2)
                store_h<NT>((chb < C ? P0 : P1) + ((rg.fy0 + py) * LW + rg.fx0 + px) * PS + (chb < C ? chb : chb - C) * 2, h);
            }
          }, 11);
    } else if constexpr (CFG::CV2_TAB) {
      pw_gk_phase<CFG, CFG::NT2, CFG::CB2, CFG::PT2, AW>(cx, rg, cat, a.cat_pitch, P0, P1, wsrc(2 * NB + 1, C2F_W_CV2), a.b[C2F_W_CV2], epi_cv2, 11);
    } else {
      pw_phase<CFG, CFG::NT2, CFG::CB2, CFG::PT2, 0, CFG::K2G, (CFG::CV2_LDS ? (CFG::Y01 ? 2 * C : C) : 0), (CFG::CV2_LDS ? C : 0), false, AW,
               decltype(epi_cv2)&, PS0, PS>(cx, rg, nullptr, 0, cat, a.cat_pitch, P0, P1, wsrc(2 * NB + 1, C2F_W_CV2), a.b[C2F_W_CV2], epi_cv2, 11);
    }
  }
  C2F_STAMP(5)

  // ---- [sppf] cv1 (cv2's output in the planes -> s, plane 0) -> cv2 accumulated over s and the three cascaded pools (sppf_tail)
  if constexpr (CFG::MODE == 2) {
    lds_sync();   // cv2's plane stores
    const Rg rg = make_region<CFG>(cx, 0);
    char* cat2 = reinterpret_cast<char*>(a.cat2);
    pw_sync_phase<CFG, CFG::NTS, CFG::CBS, CFG::PTS, 0, C, C, 2>(
        cx, rg, nullptr, 0, P0, P1, reinterpret_cast<const char*>(a.w[C2F_W_SP1]), a.b[C2F_W_SP1],
        [&](int cb, bool ok, int py, int px, const floatx4(&v)[CFG::NTS]) {
          constexpr int NT = CFG::NTS;
          const int chb = cb * 16 * NT + 4 * NT * cx.g;
          half_t h[4 * NT];
          to_half<NT>(v, h);
          if (ok) {
            store_h<NT>(P0 + ((rg.fy0 + py) * LW + rg.fx0 + px) * PS + chb * 2, h);
            if (dbg) {
              const int gpix = (cx.n * cx.H + rg.gy0 + py) * cx.W + rg.gx0 + px;
              store_h<NT>(cat2 + (size_t)((unsigned)gpix * (unsigned)a.cat2_pitch) * 2 + chb * 2, h);
            }
          }
        });
    lds_sync();
    C2F_STAMP(6)
    char* out2 = reinterpret_cast<char*>(a.out2);
    sppf_tail<CFG>(cx, rg, P0, P1, reinterpret_cast<const char*>(a.w[C2F_W_SP2]), a.b[C2F_W_SP2], dbg ? cat2 : nullptr, a.cat2_pitch,
                   [&](int cb, bool ok, int py, int px, const floatx4(&v)[CFG::NT2]) {
                     constexpr int NT = CFG::NT2;
                     const int chb = cb * 16 * NT + 4 * NT * cx.g;
                     half_t h[4 * NT];
                     to_half<NT>(v, h);
                     const int gpix = (cx.n * cx.H + rg.gy0 + py) * cx.W + rg.gx0 + px;
                     if (ok) store_h<NT>(out2 + (size_t)((unsigned)gpix * (unsigned)a.out2_pitch) * 2 + chb * 2, h);
                   });
  }
  C2F_STAMP(7)
  C2F_STAMP(15)
}

// ---- stand-alone 3x3 stride-2 conv + SiLU (the downsampling convs between the modules, model.ncnn.param:41,118): one
//      workgroup per 20 x 20 output tile, the weights staged once in LDS, the pixel operand gathered from global memory
//      (every input pixel is touched by 2.25 taps on average; the tile's neighbours share them through L1 / L2).
template <int CIN_, int COUT_, int TH_ = 20>
struct S2Cfg {
  static constexpr int KS2 = CIN_, COUT = COUT_, NW = 8, TH = TH_, TW = 20, F = 0, WPS = 2, C = 32;
  static constexpr int CT = COUT / 16, NT = min_c(CT, 4), CB = CT / NT, PT = cdiv_c(cdiv_c(TH * TW, 16), NW / CB);
  static constexpr int WBYTES = CT * 9 * (KS2 / 32) * 1024;
};
template <class CFG>
__global__ __launch_bounds__(CFG::NW * 64, 2) void s2conv_kernel(const C2fArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  Ctx cx;
  cx.lane = threadIdx.x & 63;
  cx.wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  cx.g = cx.lane >> 4;
  {
    const int col = cx.lane & 15;
    cx.sig = col < 4 ? 2 * col : (col < 12 ? 2 * (col - 4) + 1 : 2 * (col - 8));
    cx.gam = ((cx.g & 1) << 1) | (cx.g >> 1);
  }
  cx.n = blockIdx.x % a.N;
  const int tile = blockIdx.x / a.N;
  const int ty = tile / a.tiles_x, tx = tile - ty * a.tiles_x;
  cx.oy0 = ty * CFG::TH; cx.ox0 = tx * CFG::TW;
  cx.H = a.H; cx.W = a.W;
  cx.st = a.stamps ? a.stamps + (size_t)blockIdx.x * 16 : nullptr;
  cx.dbg = a.debug_store;
  C2F_STAMP(0)
  C2F_STAMP(1)
  stage_weights(cx, a.w[C2F_W_S2], smem, CFG::WBYTES, CFG::NW);
  const Rg rg = make_region<CFG>(cx, 0);
  wg_sync();
  C2F_STAMP(2)
  char* xo = reinterpret_cast<char*>(a.x);
  c3s2_phase<CFG, CFG::NT, CFG::CB, CFG::PT, true>(cx, rg, reinterpret_cast<const char*>(a.s2_in), a.s2_pitch, ASrc<true>{smem}, a.b[C2F_W_S2],
                                                  [&](int cb, bool ok, int py, int px, const floatx4(&v)[CFG::NT]) {
                                                    constexpr int NT = CFG::NT;
                                                    const int chb = cb * 16 * NT + 4 * NT * cx.g;
                                                    half_t h[4 * NT];
                                                    to_half<NT>(v, h);
                                                    const int gpix = (cx.n * cx.H + rg.gy0 + py) * cx.W + rg.gx0 + px;
                                                    if (ok) store_h<NT>(xo + (size_t)((unsigned)gpix * (unsigned)a.x_pitch) * 2 + chb * 2, h);
                                                  });
  C2F_STAMP(7)
  C2F_STAMP(15)
}
// ---- 3x3 stride-2 conv + SiLU with the INPUT tile staged in LDS (v2's downsampling convs, Cin 24 / 48; round 4).  The gather
//      kernels above fetch every pixel fragment of every tap from global memory: 16-byte pieces per lane, each input pixel 2.25
//      times, and they run at 160-220 TFLOP/s whatever the register ring's depth.  Here a workgroup copies the (2 TH + 1) x
//      (2 TW + 1) input pixels of its TH x TW output tile into an LDS plane once (coalesced 16-byte pieces, zero outside the
//      image = the conv's padding) and the K loop is c3_phase's with a pixel step of two: a plain LDS-fed MFMA loop, general K
//      packing for Cin 24 / 48 (c3_pack's order), the weights of this workgroup's output channels in a second LDS region.
//      OSPLIT workgroups share an output tile (each stages the input tile and computes COUT / OSPLIT channels) when the whole
//      weight set would not fit beside the plane.
template <int CIN_, int COUT_, int TH_, int OSPLIT_, int KSPLIT_ = 1, bool TAIL_ = false>
struct S2LCfg {
  // TAIL: a 1x1 conv + SiLU on the result (C2f.cv1 behind the stride-2 conv: COUT -> COUT channels) runs on the accumulators --
  // a lane's 4 NT consecutive channels of a pixel ARE a K group of the next MFMA's pixel operand (conv_kernels.hip tail_store)
  static constexpr bool TAIL = TAIL_;
  static constexpr int CIN = CIN_, COUT = COUT_, TH = TH_, TW = 20, NW = 8, OSPLIT = OSPLIT_, KSPLIT = KSPLIT_;
  static constexpr int COUTW = COUT / OSPLIT;   // output channels of one workgroup
  static constexpr int CINH = CIN / KSPLIT;     // input channels of one pass (Cin 96: two passes of 48 -- a 21 x 41 plane of 96 is 179 KB)
  static constexpr int CT = COUTW / 16, NT = CT % 3 == 0 ? 3 : min_c(CT, 4), CB = CT / NT;
  static constexpr bool GK = CINH % 32 != 0;
  static constexpr int G = CINH / 8, S = GK ? cdiv_c(9 * G, 4) : 9 * (CINH / 32);   // K groups per tap, K steps per pass
  static constexpr int IH = 2 * TH + 1, IW = 2 * TW + 1, LW = IW;
  static constexpr int PS = CINH % 16 != 0 ? 2 * CINH : 2 * CINH + 16;   // an odd number of 16-byte slots (C2fCfg::PS)
  static constexpr int PLANE = ((IH * LW * PS + 1023) / 1024) * 1024, WBYTES = CT * S * 1024, LDS_BYTES = PLANE + WBYTES;
  static constexpr int NPT = cdiv_c(TH * TW, 16), PT = cdiv_c(NPT, NW / CB), NBLK = cdiv_c(NPT, PT) * CB;
  static_assert(COUT % (16 * OSPLIT) == 0 && COUTW % (16 * NT) == 0 && NW % CB == 0 && CIN % (8 * KSPLIT) == 0 && LDS_BYTES <= 160 * 1024 && NBLK <= NW,
                "s2lds: shape (one block per wave: the accumulators live across the passes)");
  static_assert(CB == 1 || KSPLIT == 1, "the packed weights are [channel block][all K steps]: a pass is contiguous only for one block");
  // (NT = 2: a lane's 8 channels are one K group, one K step; NT = 3: its 12 channels are K group g of step 0 and the first half of
  //  K group g of step 1, whose second half is zero -- the tail's weights are packed to that order, S2ConvLayer::build)
  static constexpr int TSTEPS = NT == 3 ? 2 : 1;
  static_assert(!TAIL || (OSPLIT == 1 && CB == 1 && ((NT == 2 && COUT == 32) || (NT == 3 && COUT == 48))), "tail: the workgroup holds every channel of a pixel");
};
template <class CFG>
__global__ __launch_bounds__(CFG::NW * 64, 2) void s2lds_kernel(const C2fArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int TH = CFG::TH, TW = CFG::TW, NT = CFG::NT, CB = CFG::CB, PT = CFG::PT, S = CFG::S, LW = CFG::LW, PS = CFG::PS, G = CFG::G;
  char* PL = smem;
  char* WS = smem + CFG::PLANE;
  Ctx cx;
  cx.lane = threadIdx.x & 63;
  cx.wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  cx.g = cx.lane >> 4;
  {
    const int col = cx.lane & 15;
    cx.sig = col < 4 ? 2 * col : (col < 12 ? 2 * (col - 4) + 1 : 2 * (col - 8));
    cx.gam = ((cx.g & 1) << 1) | (cx.g >> 1);
  }
  cx.n = blockIdx.x % a.N;
  const int rest = blockIdx.x / a.N;
  const int os = rest % CFG::OSPLIT, tile = rest / CFG::OSPLIT;
  const int ty = tile / a.tiles_x, tx = tile - ty * a.tiles_x;
  cx.oy0 = ty * TH; cx.ox0 = tx * TW;
  cx.H = a.H; cx.W = a.W;
  cx.st = nullptr; cx.dbg = 0;
  // K step -> LDS offset of this lane's K group from the pixel's top-left tap (c3_phase's two forms)
  int soff[S];
#pragma unroll
  for (int s = 0; s < S; ++s) {
    if constexpr (CFG::GK) {
      int q = 4 * s + cx.gam;
      q = q < 9 * G ? q : 9 * G - 1;
      const int tap = q / G, cg = q - tap * G;
      const int t3 = (tap * 11) >> 5;   // tap / 3 for tap < 9
      soff[s] = (t3 * LW + tap - 3 * t3) * PS + cg * 16;
    } else {
      constexpr int SPT = CFG::CINH / 32;
      const int tap = s / SPT, cblk = s - tap * SPT;
      const int t3 = (tap * 11) >> 5;
      soff[s] = (t3 * LW + tap - 3 * t3) * PS + cblk * 64 + cx.gam * 16;
    }
  }
  // one block per wave: channel block cb, PT pixel tiles
  const bool has = cx.wave < CFG::NBLK;   // wave-uniform
  const int blk = has ? cx.wave : 0;
  const int cb = blk % CB, pbk = blk / CB;
  int pb[PT];
#pragma unroll
  for (int i = 0; i < PT; ++i) {
    int p = (pbk * PT + i) * 16 + cx.sig;
    p = p < TH * TW ? p : TH * TW - 1;
    const int py = p / TW, px = p - py * TW;
    pb[i] = (2 * py * LW + 2 * px) * PS;   // the window's top-left tap: input pixel (2 oy - 1, 2 ox - 1) = plane (2 py, 2 px)
  }
  floatx4 acc[NT][PT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int i = 0; i < PT; ++i) acc[t][i] = floatx4{0.f, 0.f, 0.f, 0.f};
  const float* bias = a.b[C2F_W_S2] + os * CFG::COUTW;
  floatx4 bv[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) bv[t] = *reinterpret_cast<const floatx4*>(bias + cb * 16 * NT + 4 * NT * cx.g + 4 * t);
#pragma unroll 1
  for (int h = 0; h < CFG::KSPLIT; ++h) {
    // ---- this pass's weights (COUTW output channels, CINH input channels) -> WS by LDS-DMA; its input channels of the tile -> PL
    stage_weights(cx, reinterpret_cast<const char*>(a.w[C2F_W_S2]) + ((size_t)os * CFG::KSPLIT + h) * CFG::WBYTES, WS, CFG::WBYTES, CFG::NW);
    {
      const int H2 = 2 * a.H, W2 = 2 * a.W, iy0 = 2 * cx.oy0 - 1, ix0 = 2 * cx.ox0 - 1;
      const char* src = reinterpret_cast<const char*>(a.s2_in) + h * CFG::CINH * 2;
      // (every piece of the tile is requested before the first is stored: 6 / 11 loads per thread in flight; in rounds of four
      //  the copy was three exposed memory round trips)
      constexpr int NPIECE = CFG::IH * CFG::IW * G, NTHR = CFG::NW * 64, U = cdiv_c(NPIECE, NTHR);
      for (int it0 = threadIdx.x; it0 < NPIECE; it0 += U * NTHR) {
        u32x4 v[U];
        int dst[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int it = it0 + u * NTHR;
          const int itc = it < NPIECE ? it : NPIECE - 1;
          const int pix = itc / G, cg = itc - pix * G;
          const int ry = pix / CFG::IW, rx = pix - ry * CFG::IW;
          const int iy = iy0 + ry, ix = ix0 + rx;
          const bool in = it < NPIECE && iy >= 0 && iy < H2 && ix >= 0 && ix < W2;
          const int iyc = in ? iy : 0, ixc = in ? ix : 0;
          const u32x4 q = *reinterpret_cast<const u32x4*>(src + (size_t)((unsigned)((cx.n * H2 + iyc) * W2 + ixc) * (unsigned)a.s2_pitch) * 2 + cg * 16);
          v[u] = in ? q : u32x4{0u, 0u, 0u, 0u};
          dst[u] = it < NPIECE ? (ry * LW + rx) * PS + cg * 16 : -1;
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
          if (dst[u] >= 0) *reinterpret_cast<u32x4*>(PL + dst[u]) = v[u];
      }
    }
    wg_sync();
    if (has) {
      int woff = (cb * S * NT * 64 + cx.lane) * 16;
      asm volatile("" : "+v"(woff));
      const ASrc<true> wsrc{WS};
      kloop<S, 2, 2, NT, PT>(
          acc, [&](int s, half8(&af)[NT]) { wsrc.template load<NT>(woff, s, af); },
          [&](int s, half8(&bf)[PT]) {
#pragma unroll
            for (int i = 0; i < PT; ++i) bf[i] = as_h8(*reinterpret_cast<const u32x4*>(PL + pb[i] + soff[s]));
          },
          NoFix{});
    }
    if (h + 1 < CFG::KSPLIT) lds_sync();   // every wave has read the plane and the weights before the next pass overwrites them
  }
  if (has) {
    char* xo = reinterpret_cast<char*>(a.x);
    half8 w2f[CFG::TSTEPS][NT];
    floatx4 b2v[NT];
    if constexpr (CFG::TAIL) {
#pragma unroll
      for (int t = 0; t < NT; ++t) {
#pragma unroll
        for (int ts = 0; ts < CFG::TSTEPS; ++ts) w2f[ts][t] = as_h8(reinterpret_cast<const u32x4*>(a.w[C2F_W_CV1])[(ts * NT + t) * 64 + cx.lane]);
        b2v[t] = *reinterpret_cast<const floatx4*>(a.b[C2F_W_CV1] + 4 * NT * cx.g + 4 * t);
      }
    }
#pragma unroll
    for (int i = 0; i < PT; ++i) {
      const int p = (pbk * PT + i) * 16 + cx.sig;
      const bool ok = p < TH * TW;
      const int pc = ok ? p : TH * TW - 1;
      const int py = pc / TW, px = pc - py * TW;
      floatx4 v[NT];
#pragma unroll
      for (int t = 0; t < NT; ++t) v[t] = silu4(acc[t][i], bv[t]);
      half_t hh[4 * NT];
      to_half<NT>(v, hh);
      if constexpr (CFG::TAIL) {   // out2 = silu(W2 . fp16(silu(conv)) + b2): K group g of step 0 = this lane's first 8 channels, of step 1 its last 4
        half8 bq[CFG::TSTEPS];
#pragma unroll
        for (int j = 0; j < 8; ++j) bq[0][j] = hh[j];
        if constexpr (CFG::TSTEPS == 2) {
#pragma unroll
          for (int j = 0; j < 8; ++j) bq[1][j] = j < 4 ? hh[8 + j] : (half_t)0.f;
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          floatx4 o = floatx4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int ts = 0; ts < CFG::TSTEPS; ++ts) o = mma16(w2f[ts][t], bq[ts], o);
          v[t] = silu4(o, b2v[t]);
        }
        to_half<NT>(v, hh);
      }
      const int chb = os * CFG::COUTW + cb * 16 * NT + 4 * NT * cx.g;
      const int gpix = (cx.n * cx.H + cx.oy0 + py) * cx.W + cx.ox0 + px;
      if (ok) store_h<NT>(xo + (size_t)((unsigned)gpix * (unsigned)a.x_pitch) * 2 + chb * 2, hh);
    }
  }
}
typedef S2LCfg<24, 48, 10, 1> S2L24x48;   // yolo_plus model.ncnn.param:10 (conv_8: 24 -> 48 @80x80): 41 KB plane + 21 KB of weights
typedef S2LCfg<48, 96, 10, 2> S2L48x96;   // :27 (conv_15: 48 -> 96 @40x40): 96 KB plane + 42 KB of weights, two workgroups per tile
typedef S2LCfg<48, 48, 10, 1> S2L48x48;   // :129 (conv_37: 48 -> 48 @40x40)
typedef S2LCfg<48, 96, 10, 2, 2> S2L48x96k2;   // the same in two passes of 24 input channels: 41 + 21 KB, two workgroups per CU
typedef S2LCfg<48, 48, 10, 1, 2> S2L48x48k2;
// (v1's 32 -> 64 convs as S2LCfg<32, 64, 10, 1, 2>, two passes of 16 channels: 22.9 / 16.2 us against 22.0 / 14.0 for the gather kernel
//  above -- at Cin 32 the tile copy costs what the gathers cost; they stay on s2conv_kernel)
typedef S2LCfg<16, 32, 10, 1, 1, true> S2L16x32t;   // v1 model.ncnn.param:19-20 (conv_6 16 -> 32 @80x80 + conv_7 = C2f.cv1 32 -> 32 as its tail)
typedef S2LCfg<24, 48, 10, 1, 1, true> S2L24x48t;   // v2 :10-11 (conv_8 24 -> 48 @80x80 + conv_9 = C2f.cv1 48 -> 48 as its tail)
typedef S2LCfg<64, 128, 10, 2, 2> S2L64x128;   // v1's 64 -> 128 convs @20x20 (model.ncnn.param:62, :134) when they do not ride in the whole-image kernels
typedef S2LCfg<96, 192, 10, 4, 2> S2L96x192;   // :44 (conv_22: 96 -> 192 @20x20): two passes of 48 input channels, four workgroups per tile
typedef S2LCfg<96, 96, 10, 2, 2> S2L96x96;     // :142 (conv_42: 96 -> 96 @20x20)

typedef S2Cfg<32, 64> S2Cfg32x64;   // model.ncnn.param:41 (conv_15: 32 -> 64 @40x40) and :118 (conv_37)
typedef S2Cfg<32, 64, 10> S2Cfg32x64h;   // half-height tiles: twice the workgroups (A/B: LITEPI_S2C_TH10=1)

// ---- SPPF in one launch for widths whose planes do not fit the whole-image C2f kernel (v2: c = 96 @20x20, yolo_plus
//      model.ncnn.param:75-87): cv1 (1x1 + SiLU) -> s in ONE LDS plane over the whole image -> cv2 accumulated over s and the three
//      cascaded 5x5 max pools as they replace each other in that plane (sppf_tail's scheme).  The pools run IN PLACE: a thread
//      owns ten outputs of one line and channel group, loads their fourteen inputs, and stores behind a workgroup barrier
//      (pool_phase needs a second plane; two planes of 96 channels are 166 KB).  OSPLIT workgroups share an image: each
//      computes s and the pools itself (cheap) and COUT / OSPLIT channels of cv2.  Weights from L2.
template <int C_, int CIN_, int COUT_, int OSPLIT_>
struct SpCfg {
  static constexpr int C = C_, CIN = CIN_, COUT = COUT_, OSPLIT = OSPLIT_, COUTW = COUT / OSPLIT, NW = 8, TH = 20, TW = 20, F = 0, LW = 20;
  static constexpr int PS = 2 * C + 16;
  static constexpr bool PERIMG = false, DEEP = true;
  static constexpr int WPS = 2;
  static constexpr int NT1 = 3, CB1 = C / 48, PT1 = cdiv_c(25, NW / CB1);
  static constexpr int NT2 = 3, CB2 = COUTW / 48, PT2 = cdiv_c(25, NW / CB2);
  static constexpr int SPT = C / 32, LDS_BYTES = TH * TW * PS;
  static_assert(C % 96 == 0 && CIN % 32 == 0 && COUTW % 48 == 0 && NW % CB1 == 0 && NW % CB2 == 0 && CB2 * cdiv_c(25, PT2) <= NW && NT2 * PT2 <= 28,
                "sppf: shape (one block per wave in cv2: the accumulators live across the pools)");
};
template <class CFG>
__device__ __forceinline__ void pool_inplace(char* P0) {
  constexpr int CG = CFG::C / 8, LW = CFG::LW, PS = CFG::PS;
  static_assert(2 * 20 * CG <= CFG::NW * 64, "one thread per (line, channel group, half line)");
  const int t = threadIdx.x;
  const bool work = t < 2 * 20 * CG;
  const int tc = work ? t : 0;
  const int half = tc / (20 * CG), line = (tc % (20 * CG)) / CG, cg = tc % CG;
  const int x0 = 10 * half;
  auto pass = [&](char* base, int step) {   // position 0 of the line; bytes between positions
    half8 o[10];
    if (work) {
      half8 v[14];
#pragma unroll
      for (int j = 0; j < 14; ++j) {
        const int x = min(max(x0 - 2 + j, 0), 19);   // clamped onto the line: the border pixel is inside the window anyway
        v[j] = *reinterpret_cast<const half8*>(base + x * step);
      }
      half8 pm[13];
#pragma unroll
      for (int j = 0; j < 13; ++j) pm[j] = __builtin_elementwise_max(v[j], v[j + 1]);
#pragma unroll
      for (int i = 0; i < 10; ++i) o[i] = __builtin_elementwise_max(__builtin_elementwise_max(pm[i], pm[i + 2]), v[i + 4]);
    }
    lds_sync();   // every thread holds its inputs
    if (work) {
#pragma unroll
      for (int i = 0; i < 10; ++i) *reinterpret_cast<half8*>(base + (x0 + i) * step) = o[i];
    }
    lds_sync();
  };
  pass(P0 + line * LW * PS + cg * 16, PS);        // along x
  pass(P0 + line * PS + cg * 16, LW * PS);        // along y
}
template <class CFG>
__global__ __launch_bounds__(CFG::NW * 64, 2) void sppf_kernel(const C2fArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int LW = CFG::LW, PS = CFG::PS, SPT = CFG::SPT;
  char* P0 = smem;
  Ctx cx;
  cx.lane = threadIdx.x & 63;
  cx.wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  cx.g = cx.lane >> 4;
  {
    const int col = cx.lane & 15;
    cx.sig = col < 4 ? 2 * col : (col < 12 ? 2 * (col - 4) + 1 : 2 * (col - 8));
    cx.gam = ((cx.g & 1) << 1) | (cx.g >> 1);
  }
  cx.n = blockIdx.x % a.N;
  const int os = blockIdx.x / a.N;
  cx.oy0 = 0; cx.ox0 = 0;
  cx.H = a.H; cx.W = a.W;
  cx.st = nullptr; cx.dbg = 0;
  const Rg rg = make_region<CFG>(cx, 0);
  // ---- cv1: s = silu(W1 . x + b1), x from global memory (the module in front stored it), s -> P0
  pw_phase<CFG, CFG::NT1, CFG::CB1, CFG::PT1, 0, CFG::CIN, 0, 0, false, false>(
      cx, rg, nullptr, 0, reinterpret_cast<const char*>(a.src1), a.pitch1, nullptr, nullptr, ASrc<false>{reinterpret_cast<const char*>(a.w[C2F_W_SP1])},
      a.b[C2F_W_SP1], [&](int cb, bool ok, int py, int px, const floatx4(&v)[CFG::NT1]) {
        constexpr int NT = CFG::NT1;
        const int chb = cb * 16 * NT + 4 * NT * cx.g;
        half_t h[4 * NT];
        to_half<NT>(v, h);
        if (ok) store_h<NT>(P0 + (py * LW + px) * PS + chb * 2, h);
      });
  lds_sync();
  // ---- cv2 over concat(s, p1, p2, p3): one block per wave, accumulators live across the pools
  constexpr int NT = CFG::NT2, CB = CFG::CB2, PT = CFG::PT2, S = 4 * SPT;
  const int npt = (rg.R + 15) >> 4;
  const int nblk = ((npt + PT - 1) / PT) * CB;
  const bool has = cx.wave < nblk;
  const int blk = has ? cx.wave : 0;
  const int cb = blk % CB, pbk = blk / CB;
  int pb[PT];
#pragma unroll
  for (int i = 0; i < PT; ++i) {
    int p = (pbk * PT + i) * 16 + cx.sig;
    p = p < rg.R ? p : rg.R - 1;
    int py, px;
    pix_of(rg, p, py, px);
    pb[i] = (py * LW + px) * PS + cx.gam * 16;
  }
  floatx4 acc[NT][PT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int i = 0; i < PT; ++i) acc[t][i] = floatx4{0.f, 0.f, 0.f, 0.f};
  int woff = (((os * CB + cb) * S) * NT * 64 + cx.lane) * 16;
  asm volatile("" : "+v"(woff));
  const char* w2 = reinterpret_cast<const char*>(a.w[C2F_W_SP2]);
#pragma unroll 1
  for (int seg = 0; seg < 4; ++seg) {
    if (seg > 0) pool_inplace<CFG>(P0);
    if (has) {
      half8 af[SPT][NT];   // (requested here: a set live across the pool would cost the strips their registers)
#pragma unroll
      for (int s = 0; s < SPT; ++s)
#pragma unroll
        for (int t = 0; t < NT; ++t) af[s][t] = as_h8(*reinterpret_cast<const u32x4*>(w2 + woff + ((seg * SPT + s) * NT + t) * 1024));
#pragma unroll
      for (int s = 0; s < SPT; ++s) {
        half8 bf[PT];
#pragma unroll
        for (int i = 0; i < PT; ++i) bf[i] = as_h8(*reinterpret_cast<const u32x4*>(P0 + pb[i] + s * 64));
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
          for (int i = 0; i < PT; ++i) acc[t][i] = mma16(af[s][t], bf[i], acc[t][i]);
      }
    }
    if (seg < 3) lds_sync();   // every wave has read the plane before the pool overwrites it
  }
  if (has) {
    char* out = reinterpret_cast<char*>(a.out2);
    const float* bias = a.b[C2F_W_SP2] + os * CFG::COUTW;
    floatx4 bv[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) bv[t] = *reinterpret_cast<const floatx4*>(bias + cb * 16 * NT + 4 * NT * cx.g + 4 * t);
#pragma unroll
    for (int i = 0; i < PT; ++i) {
      const int p0 = (pbk * PT + i) * 16 + cx.sig;
      const bool ok = p0 < rg.R;
      const int p = ok ? p0 : rg.R - 1;
      int py, px;
      pix_of(rg, p, py, px);
      floatx4 v[NT];
#pragma unroll
      for (int t = 0; t < NT; ++t) v[t] = silu4(acc[t][i], bv[t]);
      half_t h[4 * NT];
      to_half<NT>(v, h);
      const int chb = os * CFG::COUTW + cb * 16 * NT + 4 * NT * cx.g;
      const int gpix = (cx.n * cx.H + py) * cx.W + px;
      if (ok) store_h<NT>(out + (size_t)((unsigned)gpix * (unsigned)a.out2_pitch) * 2 + chb * 2, h);
    }
  }
}
typedef SpCfg<96, 192, 192, 2> SpCfgV2;   // yolo_plus model.ncnn.param:75-87 (conv_27, three pools, conv_28): 83 KB plane, two workgroups per image

// ---- instantiated configurations (YOLO-LitePi v1 widths; model.ncnn.param line of the module's cv1) ----------------------
typedef C2fCfg<32, 1, 128, 64, true, 64, 0, 0> CfgNeck40;    // :90  up(P5) | P4 -> C2f(n=1) @40x40
typedef C2fCfg<16, 1, 64, 32, true, 32, 0, 0, 16> CfgNeck80;     // :105 up(F4) | P3 -> C2f(n=1) @80x80
typedef C2fCfg<32, 1, 0, 128, false, 64, 0, 0> CfgPan40;     // :121 conv_37 | F4 -> C2f(n=1) @40x40
typedef C2fCfg<64, 1, 0, 256, false, 128, 1, 64> CfgPan20;   // :134-147 conv_42 (s2) | P5 -> C2f(n=1) @20x20
typedef C2fCfg<64, 1, 0, 128, false, 128, 2, 64> CfgBb20;    // :62-85 conv_22 (s2) -> C2f(n=1) -> SPPF @20x20
typedef C2fCfg<16, 2, 0, 32, false, 32, 0, 0, 16> CfgBb80;       // :22-38 C2f(n=2) @80x80
typedef C2fCfg<32, 2, 0, 64, false, 64, 0, 0> CfgBb40;       // :43-59 C2f(n=2) @40x40
// YOLO-LitePi v2 widths (src/tt100k/convert/model/yolo_plus/yolo_plus_ncnn_model/model.ncnn.param, line of the module's cv1).
// c = 48 / 96: the phases' weights do not fit beside the planes -- every wave streams its channel block's fragments from L2 (AW off)
typedef C2fCfg<48, 1, 192, 96, true, 96, 0, 0, 20, 8, false> CfgV2Neck40;   // :101 up(P5) | P4 -> C2f(n=1) @40x40
typedef C2fCfg<48, 1, 0, 144, false, 96, 0, 0, 20, 8, false> CfgV2Pan40;    // :132 conv_37 | F4 -> C2f(n=1) @40x40
typedef C2fCfg<24, 1, 96, 48, true, 48, 0, 0, 16> CfgV2Neck80;              // :116 up(F4) | P3 -> C2f(n=1) @80x80
typedef C2fCfg<96, 1, 0, 192, false, 192, 0, 0, 10, 8, false> CfgV2Bb20;    // :63 C2f(n=1) @20x20, two half-image tiles
typedef C2fCfg<96, 1, 0, 288, false, 192, 0, 0, 10, 8, false> CfgV2Pan20;   // :145 conv_42 | P5 -> C2f(n=1) @20x20, two half-image tiles
typedef C2fCfg<24, 2, 0, 48, false, 48, -1, 0, 20, 8, true, 0, false, 8> CfgV2Bb80x;   // v2 :13-26 C2f(n=2) @80x80 without cv1 (as CfgBb80x)
typedef C2fCfg<16, 2, 0, 32, false, 32, -1, 0, 16> CfgBb80x;     // v1 :22-38 C2f(n=2) @80x80 WITHOUT cv1 (MODE -1: cv1 is the tail of the stride-2 conv in front)
typedef C2fCfg<64, 1, 0, 256, false, 128, 0, 0, 10, 8, false> CfgPan20h;    // v1's PAN 20x20 module WITHOUT its entry conv on two half-image tiles (A/B:
typedef C2fCfg<64, 1, 0, 128, false, 128, 0, 0, 10, 8, false> CfgBb20h;     //  LITEPI_C2F_SKIP of the whole-image configurations), and the backbone's
typedef C2fCfg<24, 2, 0, 48, false, 48, 0, 0, 20, 8, true, 0, false, 8> CfgV2Bb80;   // :13 C2f(n=2) @80x80 (halo 4; 103 KB: ONE workgroup per CU, so eight waves
                                                                                     // and 20-row tiles: 130 -> 123 -> 119 us; the n = 1 module on the same shape: 91 vs 80 us for two 4-wave workgroups)
typedef C2fCfg<48, 2, 0, 96, false, 96, 0, 0, 10, 8, false> CfgV2Bb40;      // :30 C2f(n=2) @40x40 (halo 4, 10-row tiles)

template <class CFG> struct CfgName;
#define C2F_NAME(T, s) \
  template <> struct CfgName<T> { static const char* get() { return s; } };
C2F_NAME(CfgNeck40, "c2f<32,1,up128+64>")
C2F_NAME(CfgNeck80, "c2f<16,1,up64+32>")
C2F_NAME(CfgPan40, "c2f<32,1,128>")
C2F_NAME(CfgPan20, "c2f<64,1,s2+256>")
C2F_NAME(CfgBb20, "c2f<64,1,s2+128,sppf>")
C2F_NAME(CfgBb80, "c2f<16,2,32>")
C2F_NAME(CfgBb40, "c2f<32,2,64>")
C2F_NAME(CfgV2Neck40, "c2f<48,1,up192+96>")
C2F_NAME(CfgV2Pan40, "c2f<48,1,144>")
C2F_NAME(CfgV2Neck80, "c2f<24,1,up96+48>")
C2F_NAME(CfgV2Bb20, "c2f<96,1,192>")
C2F_NAME(CfgV2Pan20, "c2f<96,1,288>")
C2F_NAME(CfgV2Bb80, "c2f<24,2,48>")
C2F_NAME(CfgBb80x, "c2f<16,2,y0y1>")
C2F_NAME(CfgV2Bb80x, "c2f<24,2,y0y1>")
C2F_NAME(CfgPan20h, "c2f<64,1,256>")
C2F_NAME(CfgBb20h, "c2f<64,1,128>")
C2F_NAME(CfgV2Bb40, "c2f<48,2,96>")

template <class CFG> size_t cfg_lds() { return (size_t)CFG::LDS_BYTES; }

template <class CFG> bool try_launch(const C2fShape& s, const C2fArgs& a, hipStream_t st) {
  if (!(s == CFG::shape())) return false;
  const size_t lds = cfg_lds<CFG>();
  set_max_dynamic_lds(reinterpret_cast<const void*>(&c2f_kernel<CFG>), (int)lds);
  const int grid = a.N * a.tiles_x * a.tiles_y;
  LP_LAUNCH(c2f_kernel<CFG>, dim3(grid), dim3(CFG::NW * 64), lds, st, a);
  return true;
}
struct CfgInfo { size_t lds; const char* name; bool perimg, gk, cv2_lds; int th, tw, nt1, ntm, nt2, nt_s2; };
template <class CFG> bool try_info(const C2fShape& s, CfgInfo& ci) {
  if (!(s == CFG::shape())) return false;
  ci.lds = cfg_lds<CFG>();
  ci.name = CfgName<CFG>::get();
  ci.perimg = CFG::PERIMG;
  ci.gk = CFG::GK;
  ci.cv2_lds = CFG::CV2_LDS;
  ci.th = CFG::TH; ci.tw = CFG::TW;
  ci.nt1 = CFG::NT1; ci.ntm = CFG::NTM; ci.nt2 = CFG::NT2; ci.nt_s2 = CFG::NT_S2;
  return true;
}
// every instantiated configuration, in one place: f.template operator()<CFG>() until one returns true
template <class F> bool for_each_cfg(F&& f) {
  return f.template operator()<CfgNeck40>() || f.template operator()<CfgNeck80>() || f.template operator()<CfgPan40>() ||
         f.template operator()<CfgPan20>() || f.template operator()<CfgBb20>() || f.template operator()<CfgBb80>() ||
         f.template operator()<CfgBb40>() || f.template operator()<CfgV2Neck40>() || f.template operator()<CfgV2Pan40>() ||
         f.template operator()<CfgV2Neck80>() || f.template operator()<CfgV2Bb20>() || f.template operator()<CfgV2Pan20>() ||
         f.template operator()<CfgV2Bb80>() || f.template operator()<CfgV2Bb40>() || f.template operator()<CfgPan20h>() ||
         f.template operator()<CfgBb20h>() || f.template operator()<CfgBb80x>() || f.template operator()<CfgV2Bb80x>();
}
struct InfoFn {
  const C2fShape& s; CfgInfo& ci;
  template <class CFG> bool operator()() const { return try_info<CFG>(s, ci); }
};
struct LaunchFn {
  const C2fShape& s; const C2fArgs& a; hipStream_t st;
  template <class CFG> bool operator()() const { return try_launch<CFG>(s, a, st); }
};
// the configuration of a shape (build / launch / names: no switches -- a layer that was planned keeps its kernel)
bool cfg_info(const C2fShape& s, CfgInfo& ci) { return for_each_cfg(InfoFn{s, ci}); }
// .. and whether the PLANNER may use it (C2fLayer::supported): the A/B switches live here, read at plan time
bool cfg_enabled(const C2fShape& s, CfgInfo& ci) {
  // (CfgBb80, the n = 2 module on the 80x80 map, is opt-in: its halo-4 recompute of 16-channel layers is VALU work the
  //  two-launch bottleneck plan does not have -- 65-75 us against 59; LITEPI_C2F_BB80=1 enables it for A/B runs)
  if (!getenv("LITEPI_C2F_BB80") && s == CfgBb80::shape()) return false;
  // (v2's n = 2 backbone modules run as one launch each: 139 us against 150 for the four launches of the 80x80 one -- 145 before
  //  cv2 took y2 / y3 from the planes --, +1.5 % of the pipelined rate; 10-row tiles on the 40x40 map: 74 against 118)
  if (!cfg_info(s, ci)) return false;
  // A/B switch: LITEPI_C2F_SKIP=<configuration names separated by ';'> keeps those modules on the layer plan
  if (const char* skip = getenv("LITEPI_C2F_SKIP")) {
    const std::string list = std::string(";") + skip + ";", key = std::string(";") + ci.name + ";";
    if (list.find(key) != std::string::npos) return false;
  }
  return true;
}

int gam_of(int g) { return ((g & 1) << 1) | (g >> 1); }

// A fragments of one phase: [channel block][K step][tile][lane][8 halfs]; lane (g = lane >> 4, m = lane & 15) holds
// A[row m][K group g].  Rows are permuted so that a lane's D rows 4g .. 4g+3 of tiles 0 .. NT-1 are 4*NT consecutive
// channels (vector stores).  kidx(s, g, j) -> index into the K axis of Wm ([cout][ktot]) or -1 (zero).
template <class KIDX>
void pack_phase(DevBuf& dst, const std::vector<float>& Wm, int cout, int ktot, int NT, int S, KIDX&& kidx) {
  LP_CHECK((int)Wm.size() == cout * ktot, LP_ERR_STATE, "c2f: weight matrix %zu != %d x %d", Wm.size(), cout, ktot);
  const int CB = (cout + 16 * NT - 1) / (16 * NT);   // (rows past cout: the zero padding of the last channel tile)
  std::vector<uint16_t> buf((size_t)CB * S * NT * 64 * 8, 0);
  for (int cb = 0; cb < CB; ++cb)
    for (int s = 0; s < S; ++s)
      for (int t = 0; t < NT; ++t)
        for (int lane = 0; lane < 64; ++lane) {
          const int g = lane >> 4, m = lane & 15, gm = m >> 2, r = m & 3;
          const int oc = cb * 16 * NT + gm * 4 * NT + t * 4 + r;
          const size_t f = ((((size_t)cb * S + s) * NT + t) * 64 + lane) * 8;
          if (oc >= cout) continue;
          for (int j = 0; j < 8; ++j) {
            const int k = kidx(s, g, j);
            if (k >= 0) buf[f + j] = f32_to_f16(Wm[(size_t)oc * ktot + k]);
          }
        }
  dst.alloc(buf.size() * 2, false);
  LP_HIP(hipMemcpy(dst.p, buf.data(), buf.size() * 2, hipMemcpyHostToDevice));
}
void put_bias(DevBuf& dst, const std::vector<float>* b, int cout) {
  std::vector<float> v(cout + 64, 0.f);
  if (b) {
    LP_CHECK((int)b->size() >= cout, LP_ERR_STATE, "c2f: bias vector too short");
    for (int c = 0; c < cout; ++c) v[c] = (*b)[c];
  }
  dst.alloc(v.size() * 4, false);
  LP_HIP(hipMemcpy(dst.p, v.data(), v.size() * 4, hipMemcpyHostToDevice));
}

}  // namespace

// diagnostic stamps (LITEPI_C2F_STAMPS=<file>): 16 words per workgroup, appended to the file after the launch
static DevBuf g_stamp_buf;
static unsigned long long* stamp_buffer(size_t nwg) {
  if (g_stamp_buf.bytes < nwg * 16 * 8) g_stamp_buf.alloc(nwg * 16 * 8);
  LP_HIP(hipMemset(g_stamp_buf.p, 0, nwg * 16 * 8));
  return g_stamp_buf.as<unsigned long long>();
}
static void dump_stamps(const char* file, const std::string& name, size_t nwg, hipStream_t st) {
  LP_HIP(hipStreamSynchronize(st));
  std::vector<unsigned long long> hst(nwg * 16);
  LP_HIP(hipMemcpy(hst.data(), g_stamp_buf.p, nwg * 16 * 8, hipMemcpyDeviceToHost));
  FILE* f = fopen(file, "a");
  if (!f) return;
  fprintf(f, "# %s %zu\n", name.c_str(), nwg);
  for (size_t i = 0; i < nwg; ++i)
    for (int k = 0; k < 16; ++k) fprintf(f, "%llu%c", hst[i * 16 + k], k == 15 ? '\n' : ' ');
  fclose(f);
}

bool C2fLayer::supported(const C2fShape& s, int h, int w) {
  CfgInfo ci;
  if (!cfg_enabled(s, ci)) return false;
  if (h % ci.th != 0 || w % ci.tw != 0) return false;
  if (ci.perimg && (h != ci.th || w != ci.tw)) return false;
  if (s.UP && (h % 2 || w % 2)) return false;
  return ci.lds <= 160 * 1024;
}

void C2fLayer::build(const C2fShape& s, int h, int w, const Src& src) {
  LP_CHECK(supported(s, h, w), LP_ERR_STATE, "c2f: unsupported shape");
  sh = s; H = h; W = w;
  CfgInfo ci;
  LP_CHECK(cfg_info(s, ci), LP_ERR_STATE, "c2f: no configuration for this shape");
  lds_bytes = ci.lds;
  const int C = s.C;
  auto pw_k = [](int ktot) {
    return [ktot](int s_, int g, int j) {
      const int k = 32 * s_ + 8 * gam_of(g) + j;
      return k < ktot ? k : -1;
    };
  };
  // 3x3: weights come as [cout][tap][cin]
  auto c3_pack = [&](DevBuf& d, const std::vector<float>& wv, int cout, int cin, int NT) {
    if (cin % 32 != 0 && cin != 16) {   // general packing (C2fCfg::GK): K group q = 4 s + gam -> (tap q / G, channel group q % G)
      const int G = cin / 8, S = (9 * G + 3) / 4;
      pack_phase(d, wv, cout, 9 * cin, NT, S, [&](int s_, int g, int j) {
        const int q = 4 * s_ + gam_of(g);
        return q < 9 * G ? (q / G) * cin + 8 * (q % G) + j : -1;
      });
    } else if (cin >= 32) {
      const int spt = cin / 32;
      pack_phase(d, wv, cout, 9 * cin, NT, 9 * spt, [&](int s_, int g, int j) { return (s_ / spt) * cin + 32 * (s_ % spt) + 8 * gam_of(g) + j; });
    } else {
      static const int TA[5] = {0, 3, 6, 1, 4}, TB[5] = {2, 5, 8, 7, -1};
      pack_phase(d, wv, cout, 9 * cin, NT, 5, [&](int s_, int g, int j) {
        const int tap = (g & 1) ? TB[s_] : TA[s_];
        return tap < 0 ? -1 : tap * cin + 8 * (g >> 1) + j;
      });
    }
  };
  const int K1 = s.KA + s.KB;
  if (s.MODE != -1) {   // (MODE -1: cv1 belongs to the launch in front)
    pack_phase(d_w[C2F_W_CV1], *src.cv1, 2 * C, K1, ci.nt1, (K1 + 31) / 32, pw_k(K1));
    put_bias(d_b[C2F_W_CV1], src.cv1_b, 2 * C);
  }
  const int ntm = ci.ntm;
  for (int k = 0; k < s.NB; ++k) {
    c3_pack(d_w[C2F_W_A0 + 2 * k], *src.a[k], C, C, ntm);
    put_bias(d_b[C2F_W_A0 + 2 * k], src.a_b[k], C);
    c3_pack(d_w[C2F_W_B0 + 2 * k], *src.bb[k], C, C, ntm);
    put_bias(d_b[C2F_W_B0 + 2 * k], src.bb_b[k], C);
  }
  const int nt2 = ci.nt2;
  const int K2 = (2 + s.NB) * C;
  pack_phase(d_w[C2F_W_CV2], *src.cv2, s.COUT, K2, nt2, (K2 + 31) / 32, pw_k(K2));
  put_bias(d_b[C2F_W_CV2], src.cv2_b, s.COUT);
  macs_per_image = ((s.MODE != -1 ? (double)2 * C * K1 : 0.0) + (double)s.NB * 2 * 9 * C * C + (double)s.COUT * K2) * h * w;
  if (s.MODE >= 1) {
    c3_pack(d_w[C2F_W_S2], *src.s2, 2 * C, s.KS2, ci.nt_s2);   // NT_S2: every row tile in one wave
    put_bias(d_b[C2F_W_S2], src.s2_b, 2 * C);
    macs_per_image += 9.0 * s.KS2 * 2 * C * h * w;
  }
  if (s.MODE == 2) {
    pack_phase(d_w[C2F_W_SP1], *src.sp1, C, s.COUT, ntm, (s.COUT + 31) / 32, pw_k(s.COUT));
    put_bias(d_b[C2F_W_SP1], src.sp1_b, C);
    pack_phase(d_w[C2F_W_SP2], *src.sp2, s.COUT, 4 * C, nt2, (4 * C + 31) / 32, pw_k(4 * C));
    put_bias(d_b[C2F_W_SP2], src.sp2_b, s.COUT);
    macs_per_image += ((double)C * s.COUT + (double)s.COUT * 4 * C) * h * w;
  }
}

// the LDS-staged kernel's shapes (v2): s2lds_kernel<S2LCfg<..>>; LITEPI_NO_S2LDS=1 keeps them on conv3x3s2_direct (A/B)
static bool s2lds_shape(int cin, int cout, int hout, int wout) {
  static const bool off = getenv("LITEPI_NO_S2LDS") != nullptr;
  static const bool off96 = getenv("LITEPI_NO_S2LDS96") != nullptr;
  return !off && hout % 10 == 0 && wout % 20 == 0 &&
         ((cin == 24 && cout == 48) || (cin == 48 && (cout == 96 || cout == 48)) || (!off96 && cin == 96 && (cout == 192 || cout == 96)) ||
          (cin == 64 && cout == 128));
}
bool S2ConvLayer::tail_supported(int cin, int cout, int cout2, int hout, int wout) {
  static const bool off = getenv("LITEPI_NO_S2LDS") != nullptr || getenv("LITEPI_NO_S2TAIL") != nullptr;
  return !off && ((cin == 16 && cout == 32 && cout2 == 32) || (cin == 24 && cout == 48 && cout2 == 48)) && hout % 10 == 0 && wout % 20 == 0;
}
bool S2ConvLayer::supported(int cin, int cout, int hout, int wout) {
  return (cin == 32 && cout == 64 && hout % 20 == 0 && wout % 20 == 0) || s2lds_shape(cin, cout, hout, wout);
}

void S2ConvLayer::build(int cin, int cout, int hout, int wout, const std::vector<float>& w_taps, const std::vector<float>& bias,
                        const std::vector<float>* w_tail, const std::vector<float>* b_tail) {
  LP_CHECK(w_tail ? tail_supported(cin, cout, cout, hout, wout) : supported(cin, cout, hout, wout), LP_ERR_STATE,
           "s2conv: unsupported shape %d -> %d @%dx%d", cin, cout, hout, wout);
  Cin = cin; Cout = cout; H = hout; W = wout;
  has_tail = w_tail != nullptr;
  if (has_tail) {   // S2L16x32t / S2L24x48t: one pass, general K packing, every channel tile in one block; the tail's K groups follow
                    // the lanes' channel ownership: 4 NT consecutive channels per lane group g
    lds_staged = true; ksplit = 1;
    const int G = cin / 8, S = (9 * G + 3) / 4, nt = cout / 16;
    pack_phase(d_w, w_taps, cout, 9 * cin, nt, S, [&](int s_, int g, int j) {
      const int q = 4 * s_ + gam_of(g);
      return q < 9 * G ? (q / G) * cin + 8 * (q % G) + j : -1;
    });
    put_bias(d_b, &bias, cout);
    if (nt == 2) pack_phase(d_w2, *w_tail, cout, cout, 2, 1, [&](int, int g, int j) { return 8 * g + j; });
    else pack_phase(d_w2, *w_tail, cout, cout, 3, 2, [&](int s_, int g, int j) { return s_ == 0 ? 12 * g + j : (j < 4 ? 12 * g + 8 + j : -1); });
    put_bias(d_b2, b_tail, cout);
    return;
  }
  lds_staged = s2lds_shape(cin, cout, hout, wout);
  if (lds_staged) {   // general K packing (tap, 8-channel group), three channel tiles per block: S2LCfg / c3_phase's order;
                      // Cin 96: two passes of 48 input channels, the second pass's steps behind the first's
    ksplit = cin == 96 || cin == 64 || (cin == 48 && !getenv("LITEPI_S2LDS_K1")) ? 2 : 1;
    const int cinh = cin / ksplit;
    const int G = cinh / 8, S = (9 * G + 3) / 4;
    const int nt = cin == 64 ? 4 : 3;   // S2LCfg::NT of the configuration launch() picks (64 -> 128: 64 output channels = 4 tiles per workgroup; else 48 = 3)
    pack_phase(d_w, w_taps, cout, 9 * cin, nt, ksplit * S, [&](int s_, int g, int j) {
      const int h = s_ / S, q = 4 * (s_ % S) + gam_of(g);
      return q < 9 * G ? (q / G) * cin + h * cinh + 8 * (q % G) + j : -1;
    });
    put_bias(d_b, &bias, cout);
    return;
  }
  const int spt = cin / 32, nt = S2Cfg32x64::NT;
  pack_phase(d_w, w_taps, cout, 9 * cin, nt, 9 * spt, [&](int s_, int g, int j) { return (s_ / spt) * cin + 32 * (s_ % spt) + 8 * gam_of(g) + j; });
  put_bias(d_b, &bias, cout);
}

template <class CFG> static void launch_s2lds(const C2fArgs& a0, int N, int H, int W, hipStream_t st) {
  C2fArgs a = a0;
  a.tiles_x = W / CFG::TW; a.tiles_y = H / CFG::TH;
  set_max_dynamic_lds(reinterpret_cast<const void*>(&s2lds_kernel<CFG>), CFG::LDS_BYTES);
  LP_LAUNCH(s2lds_kernel<CFG>, dim3(N * a.tiles_x * a.tiles_y * CFG::OSPLIT), dim3(CFG::NW * 64), CFG::LDS_BYTES, st, a);
}

void S2ConvLayer::launch(const View& in, const View& out, int N, hipStream_t st) const {
  LP_CHECK(in.base && out.base && in.H == 2 * H && in.W == 2 * W && in.C >= Cin && out.H == H && out.W == W && out.C >= Cout, LP_ERR_STATE,
           "s2conv %s: bad views", name.c_str());
  LP_CHECK((double)N * in.H * in.W * in.pitch * 2.0 < 4294967296.0 && (double)N * H * W * out.pitch * 2.0 < 4294967296.0, LP_ERR_ARG,
           "s2conv: tensor too large for 32-bit byte offsets");
  C2fArgs a;
  memset(&a, 0, sizeof(a));
  a.s2_in = in.base; a.s2_pitch = in.pitch;
  a.x = out.base; a.x_pitch = out.pitch;
  a.w[C2F_W_S2] = d_w.p; a.b[C2F_W_S2] = d_b.as<float>();
  a.N = N; a.H = H; a.W = W;
  if (has_tail) {
    a.w[C2F_W_CV1] = d_w2.p; a.b[C2F_W_CV1] = d_b2.as<float>();
    if (Cin == 16) launch_s2lds<S2L16x32t>(a, N, H, W, st);
    else launch_s2lds<S2L24x48t>(a, N, H, W, st);
    LP_HIP(hipGetLastError());
    return;
  }
  if (lds_staged) {
    if (Cin == 64) launch_s2lds<S2L64x128>(a, N, H, W, st);
    else if (Cin == 24) launch_s2lds<S2L24x48>(a, N, H, W, st);
    else if (Cin == 96 && Cout == 192) launch_s2lds<S2L96x192>(a, N, H, W, st);
    else if (Cin == 96) launch_s2lds<S2L96x96>(a, N, H, W, st);
    else if (Cout == 96 && ksplit == 2) launch_s2lds<S2L48x96k2>(a, N, H, W, st);
    else if (Cout == 96) launch_s2lds<S2L48x96>(a, N, H, W, st);
    else if (ksplit == 2) launch_s2lds<S2L48x48k2>(a, N, H, W, st);
    else launch_s2lds<S2L48x48>(a, N, H, W, st);
    LP_HIP(hipGetLastError());
    return;
  }
  static const bool th10 = getenv("LITEPI_S2C_TH10") != nullptr;
  const bool half = th10 && H % 10 == 0;
  a.tiles_x = W / 20; a.tiles_y = half ? H / 10 : H / 20;
  typedef S2Cfg32x64 CFG;
  static const char* stamp_file = getenv("LITEPI_C2F_STAMPS");
  if (stamp_file) a.stamps = stamp_buffer((size_t)N * a.tiles_x * a.tiles_y);
  if (half) {
    set_max_dynamic_lds(reinterpret_cast<const void*>(&s2conv_kernel<S2Cfg32x64h>), CFG::WBYTES);
    LP_LAUNCH(s2conv_kernel<S2Cfg32x64h>, dim3(N * a.tiles_x * a.tiles_y), dim3(CFG::NW * 64), CFG::WBYTES, st, a);
  } else {
    set_max_dynamic_lds(reinterpret_cast<const void*>(&s2conv_kernel<CFG>), CFG::WBYTES);
    LP_LAUNCH(s2conv_kernel<CFG>, dim3(N * a.tiles_x * a.tiles_y), dim3(CFG::NW * 64), CFG::WBYTES, st, a);
  }
  LP_HIP(hipGetLastError());
  if (stamp_file) dump_stamps(stamp_file, name, (size_t)N * a.tiles_x * a.tiles_y, st);
}

bool SppfLayer::supported(int cin, int c, int cout, int h, int w) {
  static const bool off = getenv("LITEPI_NO_SPPF_FUSED") != nullptr;
  return !off && cin == SpCfgV2::CIN && c == SpCfgV2::C && cout == SpCfgV2::COUT && h == 20 && w == 20;
}

void SppfLayer::build(int cin, int c, int cout, int h, int w, const std::vector<float>& w1, const std::vector<float>& b1,
                      const std::vector<float>& w2, const std::vector<float>& b2) {
  LP_CHECK(supported(cin, c, cout, h, w), LP_ERR_STATE, "sppf: unsupported shape %d -> %d -> %d @%dx%d", cin, c, cout, h, w);
  Cin = cin; C = c; Cout = cout; H = h; W = w;
  auto pw_k = [](int ktot) {
    return [ktot](int s_, int g, int j) {
      const int k = 32 * s_ + 8 * gam_of(g) + j;
      return k < ktot ? k : -1;
    };
  };
  pack_phase(d_w1, w1, c, cin, SpCfgV2::NT1, cin / 32, pw_k(cin));
  put_bias(d_b1, &b1, c);
  pack_phase(d_w2, w2, cout, 4 * c, SpCfgV2::NT2, 4 * c / 32, pw_k(4 * c));
  put_bias(d_b2, &b2, cout);
  macs_per_image = ((double)c * cin + (double)cout * 4 * c) * h * w;
}

void SppfLayer::launch(const View& in, const View& out, int N, hipStream_t st) const {
  LP_CHECK(in.base && out.base && in.H == H && in.W == W && in.C >= Cin && out.H == H && out.W == W && out.C >= Cout, LP_ERR_STATE, "sppf %s: bad views", name.c_str());
  LP_CHECK((double)N * H * W * in.pitch * 2.0 < 4294967296.0 && (double)N * H * W * out.pitch * 2.0 < 4294967296.0, LP_ERR_ARG,
           "sppf: tensor too large for 32-bit byte offsets");
  C2fArgs a;
  memset(&a, 0, sizeof(a));
  a.src1 = in.base; a.pitch1 = in.pitch;
  a.out2 = out.base; a.out2_pitch = out.pitch;
  a.w[C2F_W_SP1] = d_w1.p; a.b[C2F_W_SP1] = d_b1.as<float>();
  a.w[C2F_W_SP2] = d_w2.p; a.b[C2F_W_SP2] = d_b2.as<float>();
  a.N = N; a.H = H; a.W = W; a.tiles_x = a.tiles_y = 1;
  typedef SpCfgV2 CFG;
  set_max_dynamic_lds(reinterpret_cast<const void*>(&sppf_kernel<CFG>), CFG::LDS_BYTES);
  LP_LAUNCH(sppf_kernel<CFG>, dim3(N * CFG::OSPLIT), dim3(CFG::NW * 64), CFG::LDS_BYTES, st, a);
  LP_HIP(hipGetLastError());
}

bool C2fLayer::cv2_from_lds() const {
  CfgInfo ci;
  return cfg_info(sh, ci) && ci.cv2_lds;
}

std::string C2fLayer::kernel_name() const {
  CfgInfo ci;
  return cfg_info(sh, ci) ? ci.name : "c2f";
}

void C2fLayer::launch(const IO& io, int N, hipStream_t st) const {
  C2fArgs a;
  memset(&a, 0, sizeof(a));
  auto span_ok = [&](const View& v, int mult) { return !v.base || (double)N * v.H * v.W * v.pitch * 2.0 * mult < 4294967296.0; };
  LP_CHECK(span_ok(io.src0, 1) && span_ok(io.src1, 1) && span_ok(io.cat, 1) && span_ok(io.out, 1) && span_ok(io.s2_in, 1) && span_ok(io.x, 1) &&
               span_ok(io.cat2, 1) && span_ok(io.out2, 1), LP_ERR_ARG, "c2f: tensor too large for 32-bit byte offsets");
  LP_CHECK(io.src1.base && io.cat.base && io.out.base && io.src1.H == H && io.src1.W == W && io.cat.H == H && io.out.H == H, LP_ERR_STATE,
           "c2f %s: bad views", name.c_str());
  LP_CHECK((sh.KA > 0) == (io.src0.base != nullptr), LP_ERR_STATE, "c2f %s: src0 mismatch", name.c_str());
  if (sh.KA > 0) LP_CHECK(io.src0.H == (sh.UP ? H / 2 : H) && io.src0.W == (sh.UP ? W / 2 : W) && io.src0.C >= sh.KA, LP_ERR_STATE, "c2f %s: src0 shape", name.c_str());
  LP_CHECK(io.src1.C >= sh.KB && io.cat.C >= (2 + sh.NB) * sh.C && io.out.C >= sh.COUT, LP_ERR_STATE, "c2f %s: view channels", name.c_str());
  a.src0 = io.src0.base; a.pitch0 = io.src0.pitch;
  a.src1 = io.src1.base; a.pitch1 = io.src1.pitch;
  a.cat = io.cat.base; a.cat_pitch = io.cat.pitch;
  a.out = io.out.base; a.out_pitch = io.out.pitch;
  for (int i = 0; i < C2F_NW; ++i) { a.w[i] = d_w[i].p; a.b[i] = d_b[i].as<float>(); }
  if (sh.MODE >= 1) {
    LP_CHECK(io.s2_in.base && io.x.base && io.s2_in.H == 2 * H && io.s2_in.W == 2 * W && io.s2_in.C >= sh.KS2 && io.x.C >= 2 * sh.C, LP_ERR_STATE,
             "c2f %s: entry conv views", name.c_str());
    a.s2_in = io.s2_in.base; a.s2_pitch = io.s2_in.pitch;
    a.x = io.x.base; a.x_pitch = io.x.pitch;
  }
  if (sh.MODE == 2) {
    LP_CHECK(io.cat2.base && io.out2.base && io.cat2.C >= 4 * sh.C && io.out2.C >= sh.COUT, LP_ERR_STATE, "c2f %s: SPPF views", name.c_str());
    a.cat2 = io.cat2.base; a.cat2_pitch = io.cat2.pitch;
    a.out2 = io.out2.base; a.out2_pitch = io.out2.pitch;
  }
  a.N = N; a.H = H; a.W = W;
  CfgInfo ci;
  LP_CHECK(cfg_info(sh, ci), LP_ERR_STATE, "c2f %s: no configuration for this shape", name.c_str());
  a.tiles_x = W / ci.tw; a.tiles_y = H / ci.th;
  static const bool store_all = getenv("LITEPI_C2F_STORE_ALL") != nullptr;   // bisect aid: every y segment goes to the concat buffer
  static const int dbg_flags = getenv("LITEPI_C2F_DEBUG") ? atoi(getenv("LITEPI_C2F_DEBUG")) : 0;   // see Ctx::dbg
  a.debug_store = (store_all ? 1 : 0) | (dbg_flags & ~1);
  static const char* stamp_file = getenv("LITEPI_C2F_STAMPS");
  if (stamp_file) a.stamps = stamp_buffer((size_t)N * a.tiles_x * a.tiles_y);
  const bool ok = for_each_cfg(LaunchFn{sh, a, st});
  LP_CHECK(ok, LP_ERR_STATE, "c2f %s: no kernel for this shape", name.c_str());
  LP_HIP(hipGetLastError());
  if (stamp_file && getenv("LITEPI_C2F_TWICE")) for_each_cfg(LaunchFn{sh, a, st});   // diagnostic: the stamps of an immediate second launch (warm instruction cache)
  if (stamp_file) dump_stamps(stamp_file, name, (size_t)N * a.tiles_x * a.tiles_y, st);
}

}  // namespace lp
