// MobileNetV2 / EfficientNet-B0 on the GPU (see mbnet.h).  BatchNorm (eval, eps 1e-5) is folded into the preceding
// convolution at load time.
#include "mbnet.h"

#include <cmath>
#include <cstring>

namespace lp {

MBNetClassifier::MBNetClassifier(Arch arch, int prec, int impl, int max_rois, int num_classes, int input_size)
    : arch_(arch), prec_(prec), impl_(impl), maxR_(max_rois), ncls_(num_classes), S_(input_size) {
  if (arch_ == EFFICIENTNET_B0) { act_pw_ = ACT_SILU; act_dw_ = 1; }
}

namespace {
const char* arch_name(MBNetClassifier::Arch a) { return a == MBNetClassifier::MOBILENET_V2 ? "mobilenet_v2" : "efficientnet_b0"; }
const NamedTensor& need(const std::map<std::string, NamedTensor>& sd, const std::string& key) {
  auto it = sd.find(key);
  LP_CHECK(it != sd.end() && it->second.data, LP_ERR_ARG, "state_dict lacks %s", key.c_str());
  return it->second;
}
struct Folded { std::vector<float> w, b; int co = 0, ci = 0, k = 1; };  // w [co][ci][k][k] (torch order), BN folded
Folded fold(const std::map<std::string, NamedTensor>& sd, const std::string& conv, const std::string& bn) {
  const NamedTensor& W = need(sd, conv + ".weight");
  LP_CHECK(W.shape.size() == 4 && W.shape[2] == W.shape[3], LP_ERR_ARG, "%s.weight must be [co,ci,k,k]", conv.c_str());
  Folded f;
  f.co = (int)W.shape[0]; f.ci = (int)W.shape[1]; f.k = (int)W.shape[2];
  const int n = f.ci * f.k * f.k;
  const NamedTensor& g = need(sd, bn + ".weight");
  const NamedTensor& be = need(sd, bn + ".bias");
  const NamedTensor& mu = need(sd, bn + ".running_mean");
  const NamedTensor& var = need(sd, bn + ".running_var");
  LP_CHECK((int)g.numel() == f.co && (int)be.numel() == f.co && (int)mu.numel() == f.co && (int)var.numel() == f.co, LP_ERR_ARG,
           "%s: BatchNorm size mismatch", bn.c_str());
  f.w.resize((size_t)f.co * n);
  f.b.resize(f.co);
  for (int o = 0; o < f.co; ++o) {
    const double s = (double)g.data[o] / std::sqrt((double)var.data[o] + 1e-5);
    for (int i = 0; i < n; ++i) f.w[(size_t)o * n + i] = (float)(W.data[(size_t)o * n + i] * s);
    f.b[o] = (float)((double)be.data[o] - (double)mu.data[o] * s);
  }
  return f;
}
}  // namespace

void MBNetClassifier::load(const std::map<std::string, NamedTensor>& sd) {
  loaded_ = false;
  convs_.clear(); dws_.clear(); blocks_.clear();
  LP_CHECK(S_ == 64, LP_ERR_ARG, "classifier input size must be 64 (the reference's transform is fixed at 64x64, e2e.py:367)");
  const size_t es = prec_ == LP_FP16 ? 2 : 4;
  const int hint = std::min(maxR_, 512);
  auto add_pw = [&](const std::string& name, int cin, int cout, int act, const std::vector<float>& w, const std::vector<float>& b, int hout) {
    convs_.emplace_back(new ConvLayer());
    convs_.back()->name = name;
    convs_.back()->build(prec_, impl_, 1, 1, cin, cout, act, w, b, hout, hout, hint);
    return (int)convs_.size() - 1;
  };
  auto add_dw = [&](const std::string& name, const Folded& f, int stride) {
    LP_CHECK(f.ci == 1 && (f.k == 3 || f.k == 5) && f.co % 8 == 0, LP_ERR_ARG, "%s: depthwise conv must be [C,1,3|5,3|5] with C %% 8 == 0", name.c_str());
    dws_.emplace_back();
    Dw& d = dws_.back();
    d.C = f.co; d.k = f.k; d.stride = stride; d.name = name;
    std::vector<float> w((size_t)f.k * f.k * f.co);
    for (int c = 0; c < f.co; ++c)
      for (int t = 0; t < f.k * f.k; ++t) w[(size_t)t * f.co + c] = f.w[(size_t)c * f.k * f.k + t];
    d.w.alloc(w.size() * 4);
    LP_HIP(hipMemcpy(d.w.p, w.data(), w.size() * 4, hipMemcpyHostToDevice));
    d.b.alloc(f.b.size() * 4);
    LP_HIP(hipMemcpy(d.b.p, f.b.data(), f.b.size() * 4, hipMemcpyHostToDevice));
    return (int)dws_.size() - 1;
  };
  {  // features[0]: 3x3/s2 conv + BN -> fp32 [(ky*3+kx)*3 + c][32]
    Folded f = fold(sd, "features.0.0", "features.0.1");
    LP_CHECK(f.co == 32 && f.ci == 3 && f.k == 3, LP_ERR_ARG, "%s: features.0 must be 3->32 3x3", arch_name(arch_));
    std::vector<float> w((size_t)27 * 32);
    for (int o = 0; o < 32; ++o)
      for (int c = 0; c < 3; ++c)
        for (int t = 0; t < 9; ++t) w[((size_t)t * 3 + c) * 32 + o] = f.w[((size_t)o * 3 + c) * 9 + t];
    stem_w_.alloc(w.size() * 4);
    LP_HIP(hipMemcpy(stem_w_.p, w.data(), w.size() * 4, hipMemcpyHostToDevice));
    stem_b_.alloc(32 * 4);
    LP_HIP(hipMemcpy(stem_b_.p, f.b.data(), 32 * 4, hipMemcpyHostToDevice));
  }
  // block table: (expand ratio, kernel, stride, out channels, repeats)
  struct Stage { int t, k, s, c, n; };
  static const Stage mbv2[7] = {{1, 3, 1, 16, 1}, {6, 3, 2, 24, 2}, {6, 3, 2, 32, 3}, {6, 3, 2, 64, 4}, {6, 3, 1, 96, 3}, {6, 3, 2, 160, 3}, {6, 3, 1, 320, 1}};
  static const Stage effb0[7] = {{1, 3, 1, 16, 1}, {6, 3, 2, 24, 2}, {6, 5, 2, 40, 2}, {6, 3, 2, 80, 3}, {6, 5, 1, 112, 3}, {6, 5, 2, 192, 4}, {6, 3, 1, 320, 1}};
  const Stage* stages = arch_ == MOBILENET_V2 ? mbv2 : effb0;
  int cin = 32, H = S_ / 2, fidx = 1;
  size_t widest = (size_t)32 * H * H;
  // per-role maxima (elements per ROI): block inputs / outputs, the expanded tensor in front of the depthwise conv, behind it
  size_t w_io = (size_t)32 * H * H, w_t = 0, w_u = 0;
  for (int si = 0; si < 7; ++si) {
    for (int r = 0; r < stages[si].n; ++r) {
      Block B;
      B.cin = cin; B.cout = stages[si].c; B.k = stages[si].k; B.stride = r == 0 ? stages[si].s : 1;
      B.exp = cin * stages[si].t;
      B.hin = H; B.hout = H / B.stride;
      B.res = B.stride == 1 && B.cin == B.cout;
      const bool expand = stages[si].t != 1;
      // torchvision: mobilenet_v2 numbers its blocks features.1 .. features.17 with layers under ".conv"; efficientnet_b0 groups
      // them per stage, features.{1..7}.{r} with layers under ".block"
      const std::string p = arch_ == MOBILENET_V2 ? fmt("features.%d.conv.", fidx) : fmt("features.%d.%d.block.", si + 1, r);
      B.name = p.substr(0, p.size() - 1);
      int li = 0;
      if (expand) {
        Folded e = fold(sd, p + fmt("%d.0", li), p + fmt("%d.1", li));
        LP_CHECK(e.co == B.exp && e.ci == B.cin && e.k == 1, LP_ERR_ARG, "%s%d: expand conv shape", p.c_str(), li);
        B.pw_expand = add_pw(p + "expand", B.cin, B.exp, act_pw_, e.w, e.b, B.hin);
        ++li;
      }
      {
        Folded d = fold(sd, p + fmt("%d.0", li), p + fmt("%d.1", li));
        LP_CHECK(d.co == B.exp && d.k == B.k, LP_ERR_ARG, "%s%d: depthwise conv shape", p.c_str(), li);
        B.dw = add_dw(p + "dw", d, B.stride);
        ++li;
      }
      if (arch_ == EFFICIENTNET_B0) {
        const NamedTensor& w1 = need(sd, p + fmt("%d.fc1.weight", li));
        const NamedTensor& b1 = need(sd, p + fmt("%d.fc1.bias", li));
        const NamedTensor& w2 = need(sd, p + fmt("%d.fc2.weight", li));
        const NamedTensor& b2 = need(sd, p + fmt("%d.fc2.bias", li));
        const int sq = std::max(1, B.cin / 4);
        LP_CHECK(w1.shape.size() == 4 && w1.shape[0] == sq && w1.shape[1] == B.exp && w2.shape.size() == 4 && w2.shape[0] == B.exp && w2.shape[1] == sq &&
                     (int)b1.numel() == sq && (int)b2.numel() == B.exp, LP_ERR_ARG, "%s%d: squeeze-excitation shapes", p.c_str(), li);
        B.sqp = round_up(sq, 8);
        std::vector<float> wa((size_t)B.sqp * B.exp, 0.f), ba(B.sqp, 0.f), wb((size_t)B.exp * B.sqp, 0.f), bb(B.exp, 0.f);
        for (int o = 0; o < sq; ++o) {
          memcpy(&wa[(size_t)o * B.exp], w1.data + (size_t)o * B.exp, (size_t)B.exp * 4);
          ba[o] = b1.data[o];
        }
        for (int o = 0; o < B.exp; ++o) {
          for (int i = 0; i < sq; ++i) wb[(size_t)o * B.sqp + i] = w2.data[(size_t)o * sq + i];
          bb[o] = b2.data[o];
        }
        B.se_fc1 = add_pw(p + "se.fc1", B.exp, B.sqp, ACT_SILU, wa, ba, 1);
        B.se_fc2 = add_pw(p + "se.fc2", B.sqp, B.exp, ACT_NONE, wb, bb, 1);
        ++li;
      }
      {
        // mobilenet_v2: Conv2d at index li, BatchNorm at li + 1; efficientnet_b0: Conv2dNormActivation li.{0,1}
        Folded q = arch_ == MOBILENET_V2 ? fold(sd, p + fmt("%d", li), p + fmt("%d", li + 1)) : fold(sd, p + fmt("%d.0", li), p + fmt("%d.1", li));
        LP_CHECK(q.co == B.cout && q.ci == B.exp && q.k == 1, LP_ERR_ARG, "%s%d: project conv shape", p.c_str(), li);
        B.pw_project = add_pw(p + "project", B.exp, B.cout, ACT_NONE, q.w, q.b, B.hout);
      }
      widest = std::max(widest, std::max((size_t)B.exp * B.hin * B.hin, (size_t)B.cout * B.hout * B.hout));
      w_io = std::max(w_io, (size_t)B.cout * B.hout * B.hout);
      if (expand) w_t = std::max(w_t, (size_t)B.exp * B.hin * B.hin);
      w_u = std::max(w_u, (size_t)B.exp * B.hout * B.hout);
      blocks_.push_back(B);
      cin = B.cout;
      H = B.hout;
      ++fidx;
    }
  }
  {
    const std::string lastp = arch_ == MOBILENET_V2 ? "features.18" : "features.8";
    Folded f = fold(sd, lastp + ".0", lastp + ".1");
    LP_CHECK(f.co == 1280 && f.ci == cin && f.k == 1, LP_ERR_ARG, "%s must be %d->1280 1x1", lastp.c_str(), cin);
    last_ = add_pw(lastp, cin, 1280, act_pw_, f.w, f.b, H);
    last_cin_ = cin; last_h_ = H;
    widest = std::max(widest, (size_t)1280 * H * H);
    w_t = std::max(w_t, (size_t)1280 * H * H);   // the last 1x1 conv's output shares the expand buffer
  }
  {
    const NamedTensor& fw = need(sd, "classifier.1.weight");
    const NamedTensor& fb = need(sd, "classifier.1.bias");
    LP_CHECK(fw.shape.size() == 2 && fw.shape[0] == ncls_ && fw.shape[1] == 1280 && (int)fb.numel() == ncls_, LP_ERR_ARG,
             "classifier.1 must be Linear(1280, %d)", ncls_);
    const int cp = round_up(ncls_, 8);
    lpitch_ = round_up(ncls_, 16);
    std::vector<float> w((size_t)cp * 1280, 0.f), b(cp, 0.f);
    for (int o = 0; o < ncls_; ++o) {
      memcpy(&w[(size_t)o * 1280], fw.data + (size_t)o * 1280, 1280 * 4);
      b[o] = fb.data[o];
    }
    fc_ = add_pw("classifier.1", 1280, cp, ACT_NONE, w, b, 1);
    d_logits_.alloc((size_t)maxR_ * lpitch_ * 4);
  }
  LP_CHECK((double)widest * maxR_ < 2147483648.0, LP_ERR_ARG, "%s: max_rois = %d makes an activation exceed 2^31 elements", arch_name(arch_), maxR_);
  // Buffers by ROLE, not four of the widest (with the default max_rois = max_batch x max_det = 19,200 ROIs four 96 x 32 x 32
  // buffers were 15 GB in fp16): [0], [1] block inputs / outputs (ping-pong: a residual block reads x while it writes y),
  // [2] the expanded tensor (and the last conv's 1280 channels), [3] the depthwise conv's output.
  a_x_[0].alloc((size_t)maxR_ * w_io * es, false);
  a_x_[1].alloc((size_t)maxR_ * w_io * es, false);
  a_x_[2].alloc((size_t)maxR_ * std::max(w_t, (size_t)1) * es, false);
  a_x_[3].alloc((size_t)maxR_ * w_u * es, false);
  a_mean_.alloc((size_t)maxR_ * 1280 * es);
  a_se_m_.alloc((size_t)maxR_ * 1152 * es);
  a_se_q_.alloc((size_t)maxR_ * 64 * es);
  a_se_s_.alloc((size_t)maxR_ * 1152 * es);
  loaded_ = true;
}

void MBNetClassifier::forward(const uint8_t* rgb, const int* d_R, hipStream_t st, Profiler* prof, const Post*) {
  LP_CHECK(loaded_, LP_ERR_STATE, "classifier not loaded");
  const size_t es = prec_ == LP_FP16 ? 2 : 4;
  const double esd = (double)es;
  const char* sfx = prec_ == LP_FP16 ? "_f16" : "_f32";
  auto P0 = [&]() { if (prof) prof->begin(st); };
  auto P1 = [&](const char* kname, const std::string& layer, double flops, double bytes) {
    if (prof) prof->end(st, std::string(kname) + sfx, layer, flops, bytes, true);
  };
  auto view = [&](const DevBuf& b, int C, int H) {
    View v;
    v.base = b.p; v.C = C; v.pitch = C; v.H = H; v.W = H;
    return v;
  };
  auto pw = [&](int idx, const View& in, const View& out, const View* res, bool f32 = false) {
    const ConvLayer& c = *convs_[idx];
    ConvIO io;
    io.in = in; io.out = out; io.N = maxR_; io.m_dyn = d_R; io.out_f32 = f32 ? 1 : 0;
    if (res) io.res = *res;   // added after the (absent) activation: x + project(..)
    P0();
    c.launch(io, st);
    const double px = (double)out.H * out.W;
    P1(c.impl == IMPL_NAIVE ? "conv_naive" : "conv1x1_mfma", c.name, 2.0 * c.Cin * c.Cout * px, (px * c.Cin + px * c.Cout * (res ? 2.0 : 1.0)) * esd);
  };
  // ReLU6 behind a pointwise conv: its epilogue applied ReLU, the cap follows in place
  auto cap6 = [&](const View& x, const std::string& layer) {
    if (arch_ != MOBILENET_V2) return;
    P0();
    launch_mb_eltwise(prec_, x, nullptr, 6.f, d_R, maxR_, st);
    P1("mb_clamp6", layer, 0.0, 2.0 * x.H * x.W * x.C * esd);
  };
  const int H1 = S_ / 2;
  int cur = 0;
  View x = view(a_x_[cur], stem_c_, H1);
  P0();
  launch_cls_stem_act(prec_, rgb, stem_w_.as<float>(), stem_b_.as<float>(), stem_c_, act_dw_, x, S_, d_R, maxR_, st);
  P1("cls_stem_act", "features.0", 2.0 * 27 * stem_c_ * H1 * H1, (double)S_ * S_ * 3 + (double)H1 * H1 * stem_c_ * esd);
  for (const Block& B : blocks_) {
    const int it = 2, iu = 3, iy = cur ^ 1;   // roles: see load()
    View t = x;
    if (B.pw_expand >= 0) {
      t = view(a_x_[it], B.exp, B.hin);
      pw(B.pw_expand, x, t, nullptr);
      cap6(t, B.name + ".expand");
    }
    const View u = view(a_x_[iu], B.exp, B.hout);
    const Dw& d = dws_[B.dw];
    P0();
    launch_dwconv_act(prec_, t, u, d.w.as<float>(), d.b.as<float>(), d.k, d.stride, act_dw_, d_R, maxR_, st);
    P1("dwconv_act", d.name, 2.0 * d.k * d.k * d.C * B.hout * B.hout, ((double)B.hin * B.hin + (double)B.hout * B.hout) * d.C * esd);
    if (B.se_fc1 >= 0) {
      const View m = view(a_se_m_, B.exp, 1), q = view(a_se_q_, B.sqp, 1), s = view(a_se_s_, B.exp, 1);
      P0();
      launch_spatial_mean(prec_, u, m, d_R, maxR_, st);
      P1("spatial_mean", B.name + ".se.avgpool", 0.0, (double)(B.hout * B.hout + 1) * B.exp * esd);
      pw(B.se_fc1, m, q, nullptr);
      pw(B.se_fc2, q, s, nullptr);
      P0();
      launch_mb_eltwise(prec_, u, &s, 0.f, d_R, maxR_, st);
      P1("mb_se_scale", B.name + ".se.scale", 0.0, 2.0 * B.hout * B.hout * B.exp * esd);
    }
    const View y = view(a_x_[iy], B.cout, B.hout);
    pw(B.pw_project, u, y, B.res ? &x : nullptr);
    x = y;
    cur = iy;
  }
  const View z = view(a_x_[2], 1280, last_h_);
  pw(last_, x, z, nullptr);
  cap6(z, "features.last");
  const View mean = view(a_mean_, 1280, 1);
  P0();
  launch_spatial_mean(prec_, z, mean, d_R, maxR_, st);
  P1("spatial_mean", "avgpool", 0.0, (double)(z.H * z.W + 1) * 1280 * esd);
  View lg;
  lg.base = d_logits_.p; lg.C = lpitch_; lg.pitch = lpitch_; lg.H = 1; lg.W = 1;
  pw(fc_, mean, lg, nullptr, true);
}

}  // namespace lp
