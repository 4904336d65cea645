// ResNet18 on the GPU (see resnet.h).  BatchNorm (eval, eps 1e-5) is folded into the preceding convolution at load time.
#include "resnet.h"

#include <cmath>
#include <cstring>

namespace lp {

ResNet18Classifier::ResNet18Classifier(int prec, int impl, int max_rois, int num_classes, int input_size)
    : prec_(prec), impl_(impl), maxR_(max_rois), ncls_(num_classes), S_(input_size) {}

namespace {
const NamedTensor& need(const std::map<std::string, NamedTensor>& sd, const std::string& key) {
  auto it = sd.find(key);
  LP_CHECK(it != sd.end() && it->second.data, LP_ERR_ARG, "resnet18 state_dict lacks %s", key.c_str());
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
// torch [co][ci][3][3] -> ConvLayer's [co][tap][ci]
std::vector<float> to_taps(const Folded& f) {
  std::vector<float> w((size_t)f.co * 9 * f.ci);
  for (int o = 0; o < f.co; ++o)
    for (int i = 0; i < f.ci; ++i)
      for (int t = 0; t < 9; ++t) w[((size_t)o * 9 + t) * f.ci + i] = f.w[((size_t)o * f.ci + i) * 9 + t];
  return w;
}
// a 1x1 conv as the centre tap of a 3x3 one (stride 2, pad 1: the centre tap of output (y, x) reads input (2y, 2x))
std::vector<float> centre_tap(const Folded& f) {
  std::vector<float> w((size_t)f.co * 9 * f.ci, 0.f);
  for (int o = 0; o < f.co; ++o)
    for (int i = 0; i < f.ci; ++i) w[((size_t)o * 9 + 4) * f.ci + i] = f.w[(size_t)o * f.ci + i];
  return w;
}
}  // namespace

void ResNet18Classifier::load(const std::map<std::string, NamedTensor>& sd) {
  loaded_ = false;
  convs_.clear(); blocks_.clear();
  LP_CHECK(S_ == 64, LP_ERR_ARG, "classifier input size must be 64 (the reference's transform is fixed at 64x64, e2e.py:367)");
  const size_t es = prec_ == LP_FP16 ? 2 : 4;
  const int hint = std::min(maxR_, 512);
  {  // conv1 7x7/s2 + bn1 -> fp32 [(ky*7+kx)*3 + c][64]
    Folded f = fold(sd, "conv1", "bn1");
    LP_CHECK(f.co == 64 && f.ci == 3 && f.k == 7, LP_ERR_ARG, "conv1 must be 3->64 7x7");
    std::vector<float> w((size_t)147 * 64);
    for (int o = 0; o < 64; ++o)
      for (int c = 0; c < 3; ++c)
        for (int t = 0; t < 49; ++t) w[((size_t)t * 3 + c) * 64 + o] = f.w[((size_t)o * 3 + c) * 49 + t];
    stem_w_.alloc(w.size() * 4);
    LP_HIP(hipMemcpy(stem_w_.p, w.data(), w.size() * 4, hipMemcpyHostToDevice));
    stem_b_.alloc(64 * 4);
    LP_HIP(hipMemcpy(stem_b_.p, f.b.data(), 64 * 4, hipMemcpyHostToDevice));
  }
  auto add = [&](const std::string& name, int stride, int cin, int cout, int act, const std::vector<float>& w, const std::vector<float>& b, int hout) {
    convs_.emplace_back(new ConvLayer());
    convs_.back()->name = name;
    convs_.back()->build(prec_, impl_, 3, stride, cin, cout, act, w, b, hout, hout, hint);
    return (int)convs_.size() - 1;
  };
  int cin = 64, H = S_ / 4;   // after conv1 (/2) and the max pool (/2)
  const int widths[4] = {64, 128, 256, 512};
  for (int L = 0; L < 4; ++L) {
    for (int r = 0; r < 2; ++r) {
      Block B;
      B.name = fmt("layer%d.%d", L + 1, r);
      B.cin = cin; B.cout = widths[L]; B.stride = (L > 0 && r == 0) ? 2 : 1;
      const int Ho = H / B.stride;
      B.hout = Ho;
      Folded c1 = fold(sd, B.name + ".conv1", B.name + ".bn1");
      Folded c2 = fold(sd, B.name + ".conv2", B.name + ".bn2");
      LP_CHECK(c1.co == B.cout && c1.ci == B.cin && c1.k == 3 && c2.co == B.cout && c2.ci == B.cout && c2.k == 3, LP_ERR_ARG,
               "%s: conv shapes do not match resnet18", B.name.c_str());
      B.conv1 = add(B.name + ".conv1", B.stride, B.cin, B.cout, ACT_RELU, to_taps(c1), c1.b, Ho);
      B.conv2 = add(B.name + ".conv2", 1, B.cout, B.cout, ACT_RELU, to_taps(c2), c2.b, Ho);   // ReLU after the identity is added
      if (B.stride != 1 || B.cin != B.cout) {
        Folded d = fold(sd, B.name + ".downsample.0", B.name + ".downsample.1");
        LP_CHECK(d.co == B.cout && d.ci == B.cin && d.k == 1 && B.stride == 2, LP_ERR_ARG, "%s.downsample shape", B.name.c_str());
        B.down = add(B.name + ".downsample", 2, B.cin, B.cout, ACT_NONE, centre_tap(d), d.b, Ho);
      }
      blocks_.push_back(B);
      cin = B.cout;
      H = Ho;
    }
  }
  {
    const NamedTensor& fw = need(sd, "fc.weight");
    const NamedTensor& fb = need(sd, "fc.bias");
    LP_CHECK(fw.shape.size() == 2 && fw.shape[0] == ncls_ && fw.shape[1] == 512 && (int)fb.numel() == ncls_, LP_ERR_ARG,
             "fc must be Linear(512, %d)", ncls_);
    const int cp = round_up(ncls_, 8);
    lpitch_ = round_up(ncls_, 16);
    std::vector<float> w((size_t)cp * 512, 0.f), b(cp, 0.f);
    for (int o = 0; o < ncls_; ++o) {
      memcpy(&w[(size_t)o * 512], fw.data + (size_t)o * 512, 512 * 4);
      b[o] = fb.data[o];
    }
    convs_.emplace_back(new ConvLayer());
    convs_.back()->name = "fc";
    convs_.back()->build(prec_, impl_, 1, 1, 512, cp, ACT_NONE, w, b, 1, 1, hint);
    fc_ = (int)convs_.size() - 1;
    d_logits_.alloc((size_t)maxR_ * lpitch_ * 4);
  }
  a_stem_.alloc((size_t)maxR_ * (S_ / 2) * (S_ / 2) * 64 * es);
  for (auto& a : a_x_) a.alloc((size_t)maxR_ * (S_ / 4) * (S_ / 4) * 64 * es);   // every later map is no larger (C doubles, HW quarters)
  a_mean_.alloc((size_t)maxR_ * 512 * es);
  loaded_ = true;
}

void ResNet18Classifier::forward(const uint8_t* rgb, const int* d_R, hipStream_t st, Profiler* prof, const Post*) {
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
  auto run = [&](int idx, const View& in, const View& out, const View* res) {
    const ConvLayer& c = *convs_[idx];
    ConvIO io;
    io.in = in; io.out = out; io.N = maxR_; io.m_dyn = d_R;
    if (res) { io.res = *res; io.res_first = 1; }
    P0();
    c.launch(io, st);
    const double px = (double)out.H * out.W;
    P1(c.impl == IMPL_NAIVE ? "conv_naive" : (c.stride == 2 ? "conv3x3s2_direct" : "conv3x3_mfma"), c.name, 2.0 * 9 * c.Cin * c.Cout * px,
       ((double)in.H * in.W * c.Cin + px * c.Cout * (res ? 2.0 : 1.0)) * esd);
  };
  const int H1 = S_ / 2, H2 = S_ / 4;
  const View stem = view(a_stem_, 64, H1);
  P0();
  launch_cls_stem7(prec_, rgb, stem_w_.as<float>(), stem_b_.as<float>(), stem, S_, d_R, maxR_, st);
  P1("cls_stem7", "conv1", 2.0 * 147 * 64 * H1 * H1, (double)S_ * S_ * 3 + (double)H1 * H1 * 64 * esd);
  int cur = 0;
  View x = view(a_x_[cur], 64, H2);
  P0();
  launch_maxpool3x3s2(prec_, stem, x, d_R, maxR_, st);
  P1("maxpool3x3s2", "maxpool", 0.0, ((double)H1 * H1 + (double)H2 * H2) * 64 * esd);
  for (const Block& B : blocks_) {
    // x in a_x_[cur]; t = relu(bn1(conv1(x))) -> [cur+1]; identity (or downsample(x)) ; out = relu(bn2(conv2(t)) + identity) -> [cur+2]
    const int it = (cur + 1) % 3, io = (cur + 2) % 3;
    const View t = view(a_x_[it], B.cout, B.hout);
    run(B.conv1, x, t, nullptr);
    View idn = x;
    const View out = view(a_x_[io], B.cout, B.hout);
    if (B.down >= 0) {
      // the downsampled identity takes the output buffer's place first, conv2 then reads it as its residual and overwrites it
      // in place (each lane reads exactly the elements it writes)
      run(B.down, x, out, nullptr);
      idn = out;
    }
    run(B.conv2, t, out, &idn);
    x = out;
    cur = io;
  }
  const View mean = view(a_mean_, 512, 1);
  P0();
  launch_spatial_mean(prec_, x, mean, d_R, maxR_, st);
  P1("spatial_mean", "avgpool", 0.0, (double)(x.H * x.W + 1) * 512 * esd);
  View lg;
  lg.base = d_logits_.p; lg.C = lpitch_; lg.pitch = lpitch_; lg.H = 1; lg.W = 1;
  const ConvLayer& fc = *convs_[fc_];
  ConvIO io;
  io.in = mean; io.out = lg; io.N = maxR_; io.m_dyn = d_R; io.out_f32 = 1;
  P0();
  fc.launch(io, st);
  P1(fc.impl == IMPL_NAIVE ? "conv_naive" : "conv1x1_mfma", "fc", 2.0 * 512 * fc.Cout, 512 * esd + fc.Cout * 4.0);
}

}  // namespace lp
