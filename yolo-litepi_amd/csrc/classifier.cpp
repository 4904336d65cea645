// ShuffleNetV2 x1.0 on the GPU.  BatchNorm (eval mode, eps 1e-5) is folded into the preceding
// convolution at load time.  Every stage tensor is stored as two channel halves, each padded to
// a multiple of 16 channels: [x_lo | pad | x_hi | pad]; x.chunk(2) is then two aligned channel
// views, and channel_shuffle(cat(x1, branch2(x2)), 2) is fused into the epilogue of branch2's
// last pointwise conv (conv_kernels.hip, EPI_SHUFFLE), which also copies the pass-through half.
#include "classifier.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>

namespace lp {

static const int kStageRepeats[3] = {4, 8, 4};
static const int kStageOut[3] = {116, 232, 464};

Classifier::Classifier(int prec, int impl, int max_rois, int num_classes, int input_size)
    : prec_(prec), impl_(impl), maxR_(max_rois), ncls_(num_classes), S_(input_size) {}

static View make_view(const DevBuf& buf, size_t es, int coff, int C, int pitch, int H, int W) {
  View v;
  v.base = static_cast<char*>(buf.p) + (size_t)coff * es;
  v.C = C; v.pitch = pitch; v.H = H; v.W = W;
  return v;
}

View Classifier::act_view(const Act& a, int coff, int C) const {
  return make_view(a.mem, prec_ == LP_FP16 ? 2 : 4, coff, C < 0 ? a.C : C, a.C, a.H, a.W);
}

void Classifier::alloc_act(Act& a, int C, int H, int W) {
  a.C = C; a.H = H; a.W = W;
  a.mem.alloc((size_t)maxR_ * H * W * C * (prec_ == LP_FP16 ? 2 : 4));
}

int Classifier::add_pw(const std::string& name, const std::vector<float>& w, const std::vector<float>& b, int cin,
                       int cout, int act, int hw) {
  pws_.emplace_back(new ConvLayer());
  pws_.back()->name = name;
  pws_.back()->build(prec_, impl_, 1, 1, cin, cout, act, w, b, hw, hw, std::min(maxR_, 512));
  return (int)pws_.size() - 1;
}

int Classifier::add_dw(const std::string& name, const std::vector<float>& w, const std::vector<float>& b, int C, int stride) {
  dws_.emplace_back();
  DwLayer& d = dws_.back();
  d.name = name; d.C = C; d.stride = stride;
  d.w.alloc(w.size() * 4);
  LP_HIP(hipMemcpy(d.w.p, w.data(), w.size() * 4, hipMemcpyHostToDevice));
  d.b.alloc(b.size() * 4);
  LP_HIP(hipMemcpy(d.b.p, b.data(), b.size() * 4, hipMemcpyHostToDevice));
  return (int)dws_.size() - 1;
}

namespace {
struct Folded { std::vector<float> w, b; int co = 0, ci = 0, k = 1; };  // w [co][ci][k*k]

const NamedTensor& need(const std::map<std::string, NamedTensor>& sd, const std::string& key) {
  auto it = sd.find(key);
  LP_CHECK(it != sd.end() && it->second.data, LP_ERR_ARG, "classifier state_dict lacks %s", key.c_str());
  return it->second;
}

// conv (no bias) + BatchNorm(eval) -> conv with bias
Folded fold(const std::map<std::string, NamedTensor>& sd, const std::string& conv, const std::string& bn) {
  const NamedTensor& W = need(sd, conv + ".weight");
  LP_CHECK(W.shape.size() == 4, LP_ERR_ARG, "%s.weight must be 4-D", conv.c_str());
  Folded f;
  f.co = (int)W.shape[0]; f.ci = (int)W.shape[1]; f.k = (int)W.shape[2];
  const int kk = f.k * f.k;
  const NamedTensor& g = need(sd, bn + ".weight");
  const NamedTensor& be = need(sd, bn + ".bias");
  const NamedTensor& mu = need(sd, bn + ".running_mean");
  const NamedTensor& var = need(sd, bn + ".running_var");
  LP_CHECK((int)g.numel() == f.co && (int)be.numel() == f.co && (int)mu.numel() == f.co && (int)var.numel() == f.co, LP_ERR_ARG,
           "%s: BatchNorm size mismatch", bn.c_str());
  f.w.resize((size_t)f.co * f.ci * kk);
  f.b.resize(f.co);
  for (int o = 0; o < f.co; ++o) {
    const double s = (double)g.data[o] / std::sqrt((double)var.data[o] + 1e-5);
    for (int i = 0; i < f.ci * kk; ++i) f.w[(size_t)o * f.ci * kk + i] = (float)(W.data[(size_t)o * f.ci * kk + i] * s);
    f.b[o] = (float)((double)be.data[o] - (double)mu.data[o] * s);
  }
  return f;
}

// channel layout of a stage tensor: logical c -> physical
struct Layout {
  int C = 0, half = 0, halfp = 0;  // half == 0: single contiguous segment
  int Cp() const { return half ? 2 * halfp : round_up(C, 8); }
  int phys(int c) const { return (!half || c < half) ? c : halfp + (c - half); }
};

// pointwise weights over physical channels: rows = out layout (single segment, padded), cols = in layout
void expand_pw(const Folded& f, const Layout& in, int in_first, int cout_p, std::vector<float>& w, std::vector<float>& b,
               int cin_p, int in_view_off) {
  w.assign((size_t)cout_p * cin_p, 0.f);
  b.assign(cout_p, 0.f);
  for (int o = 0; o < f.co; ++o) {
    for (int i = 0; i < f.ci; ++i) w[(size_t)o * cin_p + (in.phys(in_first + i) - in_view_off)] = f.w[(size_t)o * f.ci + i];
    b[o] = f.b[o];
  }
}

void expand_dw(const Folded& f, const Layout& in, int in_first, int cp, int in_view_off, std::vector<float>& w, std::vector<float>& b) {
  w.assign((size_t)9 * cp, 0.f);
  b.assign(cp, 0.f);
  for (int c = 0; c < f.co; ++c) {
    const int pc = in.phys(in_first + c) - in_view_off;
    for (int t = 0; t < 9; ++t) w[(size_t)t * cp + pc] = f.w[(size_t)c * 9 + t];
    b[pc] = f.b[c];
  }
}
}  // namespace

static void upload_f32(DevBuf& d, const std::vector<float>& v) {
  d.alloc((v.size() + 16) * 4);
  LP_HIP(hipMemcpy(d.p, v.data(), v.size() * 4, hipMemcpyHostToDevice));
}
static void upload_u16(DevBuf& d, const std::vector<uint16_t>& v) {
  d.alloc(v.size() * 2);
  LP_HIP(hipMemcpy(d.p, v.data(), v.size() * 2, hipMemcpyHostToDevice));
}

void Classifier::load(const std::map<std::string, NamedTensor>& sd) {
  loaded_ = false;
  pws_.clear(); dws_.clear(); blocks_.clear();
  for (auto& f : fused_) f.clear();
  // stride-1 blocks of a stage run as one fused launch (cls_fused.hip) in the fp16 MFMA configuration;
  // fp32 and the naive debug path keep one kernel per layer
  use_fused_ = prec_ == LP_FP16 && impl_ == IMPL_MFMA && S_ == 64;
  // the reference resizes every ROI to 64x64 whatever --cls_input_size says (transforms.Resize((64, 64)), e2e.py:367)
  LP_CHECK(S_ == 64, LP_ERR_ARG, "classifier input size must be 64 (the reference's transform is fixed at 64x64)");

  // conv1 + BN -> fp32 [27][24], RGB order
  {
    Folded f = fold(sd, "conv1.0", "conv1.1");
    LP_CHECK(f.co == 24 && f.ci == 3 && f.k == 3, LP_ERR_ARG, "conv1 must be 3->24 3x3");
    std::vector<float> w((size_t)27 * 24);
    for (int o = 0; o < 24; ++o)
      for (int c = 0; c < 3; ++c)
        for (int t = 0; t < 9; ++t) w[(size_t)(t * 3 + c) * 24 + o] = f.w[((size_t)o * 3 + c) * 9 + t];
    stem_w_.alloc(w.size() * 4);
    LP_HIP(hipMemcpy(stem_w_.p, w.data(), w.size() * 4, hipMemcpyHostToDevice));
    stem_b_.alloc(24 * 4);
    LP_HIP(hipMemcpy(stem_b_.p, f.b.data(), 24 * 4, hipMemcpyHostToDevice));
    if (use_fused_) {
      std::vector<uint16_t> fr;
      std::vector<float> bc;
      pack_cls_stem(f.w, f.b, fr, bc);
      upload_u16(stem_frag_, fr);
      upload_f32(stem_bias4_, bc);
    }
  }
  int H = S_ / 2;
  alloc_act(a_stem_, 24, H, H);
  H /= 2;
  alloc_act(a_pool_, 24, H, H);

  Layout lin;
  lin.C = 24;  // single segment
  size_t t1_elems = 0, t2_elems = 0, b1dw_elems = 0, b1_elems = 0;
  int inp = 24;
  for (int s = 0; s < 3; ++s) {
    const int oup = kStageOut[s], bf = oup / 2, bfp = round_up(bf, 16);
    half_c_[s] = bf; half_cp_[s] = bfp;
    Layout lout;
    lout.C = oup; lout.half = bf; lout.halfp = bfp;
    const int Hin = H, Ho = H / 2;
    for (int r = 0; r < kStageRepeats[s]; ++r) {
      Block B;
      B.name = fmt("stage%d.%d", s + 2, r);
      B.inp = inp; B.oup = oup; B.stride = r == 0 ? 2 : 1;
      const std::string p = B.name + ".";
      std::vector<float> w, b;
      if (B.stride == 2) {
        // branch1: dw(s2) on all input channels -> pw inp->bf + ReLU
        Folded d = fold(sd, p + "branch1.0", p + "branch1.1");
        LP_CHECK(d.co == inp && d.ci == 1 && d.k == 3, LP_ERR_ARG, "%sbranch1.0 shape", p.c_str());
        expand_dw(d, lin, 0, lin.Cp(), 0, w, b);
        B.b1_dw = add_dw(p + "branch1.0", w, b, lin.Cp(), 2);
        if (use_fused_) { upload_f32(fused_s2_[s].dw1, w); upload_f32(fused_s2_[s].dw1b, b); }
        Folded q = fold(sd, p + "branch1.2", p + "branch1.3");
        LP_CHECK(q.co == bf && q.ci == inp && q.k == 1, LP_ERR_ARG, "%sbranch1.2 shape", p.c_str());
        expand_pw(q, lin, 0, bfp, w, b, lin.Cp(), 0);
        B.b1_pw = add_pw(p + "branch1.2", w, b, lin.Cp(), bfp, ACT_RELU, Ho);
        if (use_fused_) { upload_u16(fused_s2_[s].pwb1, pack_fused_pw(w, bfp, lin.Cp())); upload_f32(fused_s2_[s].pwb1b, b); }
        // branch2: pw1 inp->bf + ReLU (full resolution), dw s2, pw2 + ReLU (+shuffle with branch1)
        Folded a1 = fold(sd, p + "branch2.0", p + "branch2.1");
        LP_CHECK(a1.co == bf && a1.ci == inp, LP_ERR_ARG, "%sbranch2.0 shape", p.c_str());
        expand_pw(a1, lin, 0, bfp, w, b, lin.Cp(), 0);
        B.b2_pw1 = add_pw(p + "branch2.0", w, b, lin.Cp(), bfp, ACT_RELU, Hin);
        if (use_fused_) { upload_u16(fused_s2_[s].pw1, pack_fused_pw(w, bfp, lin.Cp())); upload_f32(fused_s2_[s].pw1b, b); }
        b1dw_elems = std::max(b1dw_elems, (size_t)Ho * Ho * lin.Cp());
        b1_elems = std::max(b1_elems, (size_t)Ho * Ho * bfp);
        t1_elems = std::max(t1_elems, (size_t)Hin * Hin * bfp);
      } else {
        // x1 = first half (pass-through), x2 = second half -> branch2
        Folded a1 = fold(sd, p + "branch2.0", p + "branch2.1");
        LP_CHECK(a1.co == bf && a1.ci == bf, LP_ERR_ARG, "%sbranch2.0 shape", p.c_str());
        expand_pw(a1, lout, bf, bfp, w, b, bfp, bfp);
        B.b2_pw1 = add_pw(p + "branch2.0", w, b, bfp, bfp, ACT_RELU, Ho);
        t1_elems = std::max(t1_elems, (size_t)Ho * Ho * bfp);
        if (use_fused_) {
          fused_[s].emplace_back();
          upload_u16(fused_[s].back().w1, pack_fused_pw(w, bfp, bfp));
          upload_f32(fused_[s].back().b1, b);
        }
      }
      Layout lmid;
      lmid.C = bf;
      Folded d2 = fold(sd, p + "branch2.3", p + "branch2.4");
      LP_CHECK(d2.co == bf && d2.ci == 1 && d2.k == 3, LP_ERR_ARG, "%sbranch2.3 shape", p.c_str());
      expand_dw(d2, lmid, 0, bfp, 0, w, b);
      B.b2_dw = add_dw(p + "branch2.3", w, b, bfp, B.stride);
      if (use_fused_ && B.stride == 1) { upload_f32(fused_[s].back().dw, w); upload_f32(fused_[s].back().dwb, b); }
      if (use_fused_ && B.stride == 2) { upload_f32(fused_s2_[s].dw2, w); upload_f32(fused_s2_[s].dw2b, b); }
      Folded a2 = fold(sd, p + "branch2.5", p + "branch2.6");
      LP_CHECK(a2.co == bf && a2.ci == bf, LP_ERR_ARG, "%sbranch2.5 shape", p.c_str());
      expand_pw(a2, lmid, 0, bfp, w, b, bfp, 0);
      B.b2_pw2 = add_pw(p + "branch2.5", w, b, bfp, bfp, ACT_RELU, Ho);
      if (use_fused_ && B.stride == 1) { upload_u16(fused_[s].back().w2, pack_fused_pw(w, bfp, bfp)); upload_f32(fused_[s].back().b2, b); }
      if (use_fused_ && B.stride == 2) { upload_u16(fused_s2_[s].pw2, pack_fused_pw(w, bfp, bfp)); upload_f32(fused_s2_[s].pw2b, b); }
      t2_elems = std::max(t2_elems, (size_t)Ho * Ho * bfp);
      blocks_.push_back(B);
      inp = oup;
      lin = lout;
    }
    H = Ho;
    alloc_act(a_stage_[s][0], 2 * bfp, H, H);
    alloc_act(a_stage_[s][1], 2 * bfp, H, H);
  }
  if (use_fused_) {
    LP_CHECK(half_c_[0] == 58 && half_cp_[0] == 64 && half_c_[1] == 116 && half_cp_[1] == 128 && half_c_[2] == 232 && half_cp_[2] == 240,
             LP_ERR_STATE, "fused classifier kernels are built for ShuffleNetV2 x1.0 widths");
    a_x3_.alloc((size_t)maxR_ * 16 * 256 * 2);
  }
  const size_t es = prec_ == LP_FP16 ? 2 : 4;
  a_t1_.mem.alloc((size_t)maxR_ * t1_elems * es);
  a_t2_.mem.alloc((size_t)maxR_ * t2_elems * es);
  a_b1dw_.mem.alloc((size_t)maxR_ * b1dw_elems * es);
  a_b1_.mem.alloc((size_t)maxR_ * b1_elems * es);

  // conv5 + BN + ReLU, mean, fc
  {
    Folded f = fold(sd, "conv5.0", "conv5.1");
    LP_CHECK(f.co == 1024 && f.ci == 464 && f.k == 1, LP_ERR_ARG, "conv5 must be 464->1024 1x1");
    std::vector<float> w, b;
    expand_pw(f, lin, 0, 1024, w, b, lin.Cp(), 0);
    conv5_ = add_pw("conv5.0", w, b, lin.Cp(), 1024, ACT_RELU, H);
    if (use_fused_) {
      LP_CHECK(H == 2, LP_ERR_STATE, "fused head expects a 2x2 stage-4 map");
      head_cin_p_ = lin.Cp();
      upload_u16(head_w5_, pack_fused_pw(w, 1024, lin.Cp()));
      upload_f32(head_b5_, b);
    }
    alloc_act(a_conv5_, 1024, H, H);
    alloc_act(a_mean_, 1024, 1, 1);
    const NamedTensor& fw = need(sd, "fc.weight");
    const NamedTensor& fb = need(sd, "fc.bias");
    LP_CHECK(fw.shape.size() == 2 && fw.shape[0] == ncls_ && fw.shape[1] == 1024 && (int)fb.numel() == ncls_, LP_ERR_ARG,
             "fc must be Linear(1024, %d), got [%lld,%lld]", ncls_, (long long)(fw.shape.size() ? fw.shape[0] : 0),
             (long long)(fw.shape.size() > 1 ? fw.shape[1] : 0));
    const int cp = round_up(ncls_, 8);
    lpitch_ = round_up(ncls_, 16);
    std::vector<float> w2((size_t)cp * 1024, 0.f), b2(cp, 0.f);
    for (int o = 0; o < ncls_; ++o) {
      memcpy(&w2[(size_t)o * 1024], fw.data + (size_t)o * 1024, 1024 * 4);
      b2[o] = fb.data[o];
    }
    fc_ = add_pw("fc", w2, b2, 1024, cp, ACT_NONE, 1);
    if (use_fused_) {
      head_nc_p_ = round_up(ncls_, 16);
      std::vector<float> wf((size_t)head_nc_p_ * 1024, 0.f), bf2(head_nc_p_, 0.f);
      for (int o = 0; o < ncls_; ++o) {
        memcpy(&wf[(size_t)o * 1024], fw.data + (size_t)o * 1024, 1024 * 4);
        bf2[o] = fb.data[o];
      }
      upload_u16(head_wfc_, pack_fused_pw(wf, head_nc_p_, 1024));
      upload_f32(head_bfc_, bf2);
    }
    d_logits_.alloc((size_t)maxR_ * lpitch_ * 4);
  }
  loaded_ = true;
}

void Classifier::forward(const uint8_t* rgb, const int* d_R, hipStream_t st, Profiler* prof, const Post* post) {
  LP_CHECK(loaded_, LP_ERR_STATE, "classifier not loaded");
  const size_t es = prec_ == LP_FP16 ? 2 : 4;
  const char* sfx = prec_ == LP_FP16 ? "_f16" : "_f32";
  const double esd = (double)es;
  auto P0 = [&]() { if (prof) prof->begin(st); };
  auto P1 = [&](const char* kname, const std::string& layer, double flops, double bytes) {
    if (prof) prof->end(st, std::string(kname) + sfx, layer, flops, bytes, true);
  };
  auto run_pw = [&](int idx, const View& in, const View& out, const View* x1, int half_c, int half_cp, int out_f32) {
    const ConvLayer& c = *pws_[idx];
    ConvIO io;
    io.in = in; io.out = out; io.N = maxR_; io.m_dyn = d_R; io.out_f32 = out_f32;
    if (x1) { io.x1 = *x1; io.half_c = half_c; io.half_cp = half_cp; }
    P0();
    c.launch(io, st);
    const double px = (double)out.H * out.W;
    P1(c.impl == IMPL_NAIVE ? "conv_naive" : "conv1x1_mfma", c.name, 2.0 * c.Cin * c.Cout * px,
       px * (c.Cin + c.Cout * (x1 ? 2.0 : 1.0)) * esd);
  };
  auto run_dw = [&](int idx, const View& in, const View& out) {
    const DwLayer& d = dws_[idx];
    P0();
    launch_dwconv3x3(prec_, in, out, d.w.as<float>(), d.b.as<float>(), d.stride, d_R, maxR_, st);
    P1("dwconv3x3", d.name, 2.0 * 9 * d.C * out.H * out.W, ((double)in.H * in.W + (double)out.H * out.W) * d.C * esd);
  };

  static const bool two_launch = getenv("LITEPI_CLS_LAYERWISE") == nullptr;  // A/B switch: the layer-at-a-time plan of round 1
  if (use_fused_ && two_launch) {
    auto s2w = [](const FusedS2& f) {
      FusedS2W w;
      w.dw1 = f.dw1.as<float>(); w.dw1b = f.dw1b.as<float>(); w.pwb1 = f.pwb1.as<u32x4_t>(); w.pwb1b = f.pwb1b.as<float>();
      w.pw1 = f.pw1.as<u32x4_t>(); w.pw1b = f.pw1b.as<float>(); w.dw2 = f.dw2.as<float>(); w.dw2b = f.dw2b.as<float>();
      w.pw2 = f.pw2.as<u32x4_t>(); w.pw2b = f.pw2b.as<float>();
      return w;
    };
    auto s1w = [](const FusedW& f) {
      FusedBlockW w;
      w.w1 = f.w1.as<u32x4_t>(); w.b1 = f.b1.as<float>(); w.dw = f.dw.as<float>(); w.dwb = f.dwb.as<float>();
      w.w2 = f.w2.as<u32x4_t>(); w.b2 = f.b2.as<float>();
      return w;
    };
    ClsFrontArgs fa;
    memset(&fa, 0, sizeof(fa));
    fa.rgb = rgb; fa.m_dyn = d_R; fa.out = a_x3_.p;
    fa.stem_w = stem_frag_.as<u32x4_t>(); fa.stem_b = stem_bias4_.as<float>();
    fa.s20 = s2w(fused_s2_[0]);
    for (int q = 0; q < 3; ++q) fa.s2[q] = s1w(fused_[0][q]);
    fa.s30 = s2w(fused_s2_[1]);
    P0();
    launch_cls_front(fa, maxR_, st);
    // conv1 0.66 + stage2 2.10 + stage3.0 0.6 MMAC per ROI; bytes: the uint8 crop in, 8 KB out
    P1("cls_front", "conv1..stage3.0", 2.0 * (0.66e6 + 2.10e6 + 0.60e6), 12288.0 + 8192.0);
    ClsBackArgs ba;
    memset(&ba, 0, sizeof(ba));
    ba.in = a_x3_.p; ba.m_dyn = d_R;
    for (int q = 0; q < 7; ++q) ba.s3[q] = s1w(fused_[1][q]);
    ba.s40 = s2w(fused_s2_[2]);
    for (int q = 0; q < 3; ++q) ba.s4[q] = s1w(fused_[2][q]);
    ba.w5 = head_w5_.as<u32x4_t>(); ba.b5 = head_b5_.as<float>();
    ba.wfc = head_wfc_.as<u32x4_t>(); ba.bfc = head_bfc_.as<float>();
    ba.nc = ncls_; ba.nc_p = head_nc_p_;
    ba.logits = d_logits_.as<float>(); ba.logits_pitch = lpitch_;
    if (post) {
      ba.probs = post->probs; ba.ids = post->ids; ba.dets = post->dets; ba.max_det = post->max_det;
      ba.roi_img = post->roi_img; ba.roi_slot = post->roi_slot;
    }
    P0();
    launch_cls_back(ba, maxR_, st);
    P1("cls_back", "stage3.1..softmax", 2.0 * (3.86e6 + 2.63e6 + 1.90e6 + 1024.0 * ncls_), 8192.0 + ncls_ * 4.0);
    return;
  }
  P0();
  launch_cls_stem(prec_, rgb, stem_w_.as<float>(), stem_b_.as<float>(), 24, act_view(a_stem_), S_, d_R, maxR_, st);
  P1("cls_stem", "conv1", 2.0 * 27 * 24 * a_stem_.H * a_stem_.W, (double)S_ * S_ * 3 + (double)a_stem_.H * a_stem_.W * 24 * esd);
  P0();
  launch_maxpool3x3s2(prec_, act_view(a_stem_), act_view(a_pool_), d_R, maxR_, st);
  P1("maxpool3x3s2", "maxpool", 0.0, ((double)a_stem_.H * a_stem_.W + (double)a_pool_.H * a_pool_.W) * 24 * esd);

  View x = act_view(a_pool_);
  size_t bi = 0;
  for (int s = 0; s < 3; ++s) {
    const int bf = half_c_[s], bfp = half_cp_[s];
    for (int r = 0; r < kStageRepeats[s]; ++r, ++bi) {
      const Block& B = blocks_[bi];
      if (use_fused_ && r == 1) {
        // blocks 1..n-1 of the stage in one launch, X resident in LDS
        Act& dstf = a_stage_[s][1];
        FusedStageArgs fa;
        memset(&fa, 0, sizeof(fa));
        fa.in = x.base; fa.out = dstf.mem.p; fa.m_dyn = d_R;
        fa.nblk = kStageRepeats[s] - 1;
        for (int q = 0; q < fa.nblk; ++q) {
          const FusedW& fw = fused_[s][q];
          fa.blk[q].w1 = fw.w1.as<u32x4_t>(); fa.blk[q].b1 = fw.b1.as<float>();
          fa.blk[q].dw = fw.dw.as<float>(); fa.blk[q].dwb = fw.dwb.as<float>();
          fa.blk[q].w2 = fw.w2.as<u32x4_t>(); fa.blk[q].b2 = fw.b2.as<float>();
        }
        fa.HW = dstf.H * dstf.W; fa.W = dstf.W; fa.bf = bf; fa.bfp = bfp; fa.group = 64 / fa.HW;
        fa.in_pitch = x.pitch; fa.out_pitch = dstf.C;
        P0();
        launch_fused_stage(fa, maxR_, st);
        const double px = (double)fa.HW;
        P1("shuffle_stage_fused", fmt("stage%d.1-%d", s + 2, kStageRepeats[s] - 1),
           fa.nblk * (2.0 * 2.0 * bfp * bfp + 2.0 * 9 * bfp) * px, 2.0 * px * 2 * bfp * esd);
        x = act_view(dstf);
        bi += kStageRepeats[s] - 1;  // the loop's ++bi does not run after break
        break;
      }
      Act& dst = a_stage_[s][r & 1];
      const int Ho = dst.H;
      const View out = act_view(dst);
      const View t2 = make_view(a_t2_.mem, es, 0, bfp, bfp, Ho, Ho);
      if (B.stride == 2) {
        const View b1dw = make_view(a_b1dw_.mem, es, 0, x.C, x.C, Ho, Ho);
        const View b1 = make_view(a_b1_.mem, es, 0, bfp, bfp, Ho, Ho);
        const View t1 = make_view(a_t1_.mem, es, 0, bfp, bfp, x.H, x.W);
        run_dw(B.b1_dw, x, b1dw);
        run_pw(B.b1_pw, b1dw, b1, nullptr, 0, 0, 0);
        run_pw(B.b2_pw1, x, t1, nullptr, 0, 0, 0);
        run_dw(B.b2_dw, t1, t2);
        run_pw(B.b2_pw2, t2, out, &b1, bf, bfp, 0);
      } else {
        const View x1 = View{x.base, bfp, x.pitch, x.H, x.W};
        const View x2 = View{static_cast<char*>(x.base) + (size_t)bfp * es, bfp, x.pitch, x.H, x.W};
        const View t1 = make_view(a_t1_.mem, es, 0, bfp, bfp, Ho, Ho);
        run_pw(B.b2_pw1, x2, t1, nullptr, 0, 0, 0);
        run_dw(B.b2_dw, t1, t2);
        run_pw(B.b2_pw2, t2, out, &x1, bf, bfp, 0);
      }
      x = out;
    }
  }
  if (use_fused_) {
    FusedHeadArgs ha;
    memset(&ha, 0, sizeof(ha));
    ha.in = x.base; ha.m_dyn = d_R; ha.in_pitch = x.pitch; ha.cin_p = head_cin_p_;
    ha.w5 = head_w5_.as<u32x4_t>(); ha.b5 = head_b5_.as<float>();
    ha.wfc = head_wfc_.as<u32x4_t>(); ha.bfc = head_bfc_.as<float>();
    ha.nc = ncls_; ha.nc_p = head_nc_p_;
    ha.logits = d_logits_.as<float>(); ha.logits_pitch = lpitch_;
    if (post) {
      ha.probs = post->probs; ha.ids = post->ids; ha.dets = post->dets; ha.max_det = post->max_det;
      ha.roi_img = post->roi_img; ha.roi_slot = post->roi_slot;
    }
    P0();
    launch_fused_head(ha, maxR_, st);
    P1("cls_head_fused", "conv5+mean+fc+softmax", 4.0 * 2.0 * head_cin_p_ * 1024 + 2.0 * 1024 * head_nc_p_,
       4.0 * head_cin_p_ * esd + ncls_ * 4.0);
    return;
  }
  run_pw(conv5_, x, act_view(a_conv5_), nullptr, 0, 0, 0);
  P0();
  launch_spatial_mean(prec_, act_view(a_conv5_), act_view(a_mean_), d_R, maxR_, st);
  P1("spatial_mean", "mean", 0.0, (double)(a_conv5_.H * a_conv5_.W + 1) * 1024 * esd);
  View lg;
  lg.base = d_logits_.p; lg.C = lpitch_; lg.pitch = lpitch_; lg.H = 1; lg.W = 1;
  run_pw(fc_, act_view(a_mean_), lg, nullptr, 0, 0, 1);
}

}  // namespace lp
