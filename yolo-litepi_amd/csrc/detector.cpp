// Detector planner: turns the NCNN graph of a YOLOv8-family detector (reference
// model.ncnn.param:3-208; YOLO-LitePi v1/v2 and the YOLOv8n baseline share this topology) into
// a list of fused NHWC kernel launches:
//   * Convolution+Swish(+BinaryOp add) -> one conv kernel with bias/SiLU/residual epilogue
//   * Split -> alias, Slice -> channel-slice view, Concat -> producers write straight into
//     channel slices of one buffer (C2f, SPPF, FPN/PAN concats are never materialised)
//   * three chained 5x5 max pools -> one SPPF kernel
//   * the Reshape/Permute/Softmax/DFL/BinaryOp/Sigmoid tail -> one decode kernel
// Channel counts that are not multiples of 8 (v2: 12-channel C2f halves) are padded per
// segment; padding channels carry zero weights on both sides and stay zero.
#include "detector.h"
#include <cstring>

#include <algorithm>
#include <functional>
#include <set>

namespace lp {

// ---- profiler -------------------------------------------------------------------------
thread_local LaunchTimer* g_launch_timer = nullptr;
bool print_launches() {
  static const bool on = getenv("LITEPI_PRINT_LAUNCH") != nullptr;
  return on;
}

void Profiler::begin(hipStream_t st) {
  if (!enabled) return;
  LP_HIP(hipEventCreate(&cur));
  LP_HIP(hipEventRecord(cur, st));
  timer = LaunchTimer();
  LP_HIP(hipEventCreate(&timer.e0));
  LP_HIP(hipEventCreate(&timer.e1));
  g_launch_timer = &timer;
}
void Profiler::end(hipStream_t st, const std::string& name, const std::string& layer, double flops, double bytes, bool per_roi) {
  if (!enabled) return;
  g_launch_timer = nullptr;
  Rec r{name, layer, flops, bytes, cur, nullptr, per_roi, timer.e0, timer.e1, timer.launches};
  LP_HIP(hipEventCreate(&r.e1));
  LP_HIP(hipEventRecord(r.e1, st));
  recs.push_back(r);
  cur = nullptr;
}
void Profiler::collect(int roi_count) {
  results.clear();
  for (auto& r : recs) {
    lp_kernel_time k;
    memset(&k, 0, sizeof(k));
    snprintf(k.name, sizeof(k.name), "%s", r.name.c_str());
    snprintf(k.layer, sizeof(k.layer), "%s", r.layer.c_str());
    float ms = 0.f, kms = 0.f;
    if (hipEventElapsedTime(&ms, r.e0, r.e1) != hipSuccess) ms = -1.f;
    // one launch inside the bracket: its own start -> stop events (the kernel without the dispatch around it)
    if (r.launches == 1 && hipEventElapsedTime(&kms, r.k0, r.k1) == hipSuccess && kms > 0.f) ms = kms;
    k.ms = ms;
    k.flops = r.flops * (r.per_roi ? roi_count : 1);
    k.bytes = r.bytes * (r.per_roi ? roi_count : 1);
    results.push_back(k);
    (void)hipEventDestroy(r.e0);
    (void)hipEventDestroy(r.e1);
    (void)hipEventDestroy(r.k0);
    (void)hipEventDestroy(r.k1);
  }
  recs.clear();
}
Profiler::~Profiler() {
  for (auto& r : recs) { (void)hipEventDestroy(r.e0); (void)hipEventDestroy(r.e1); (void)hipEventDestroy(r.k0); (void)hipEventDestroy(r.k1); }
  if (g_launch_timer == &timer) g_launch_timer = nullptr;
}

// ---- tensors ---------------------------------------------------------------------------
int Tensor::phys(int c) const {
  int lo = 0, po = 0;
  for (int s : segs) {
    if (c < lo + s) return po + (c - lo);
    lo += s;
    po += round_up(s, 8);
  }
  return -1;
}

Detector::Detector(int prec, int impl, int max_batch, int input_size)
    : prec_(prec), impl_(impl), maxB_(max_batch), S_(input_size) {}

View Detector::view(int t) const {
  const Tensor* T = &tensors_[t];
  int off = 0;
  if (T->parent >= 0) {
    const Tensor& P = tensors_[T->parent];
    for (int k = 0; k < T->parent_seg; ++k) off += round_up(P.segs[k], 8);
    off += P.off;
    LP_CHECK(P.buf >= 0, LP_ERR_STATE, "blob %s: parent has no storage", T->name.c_str());
    const Buffer& b = buffers_[P.buf];
    View v;
    v.base = static_cast<char*>(b.mem.p) + (size_t)off * (prec_ == LP_FP16 ? 2 : 4);
    v.C = T->Cp; v.pitch = b.Cp; v.H = T->H; v.W = T->W;
    return v;
  }
  LP_CHECK(T->buf >= 0, LP_ERR_STATE, "blob %s has no storage", T->name.c_str());
  const Buffer& b = buffers_[T->buf];
  View v;
  v.base = static_cast<char*>(b.mem.p) + (size_t)T->off * (prec_ == LP_FP16 ? 2 : 4);
  v.C = T->Cp; v.pitch = b.Cp; v.H = T->H; v.W = T->W;
  return v;
}

void Detector::load(const std::string& param_path, const std::string& bin_path) {
  NcnnGraph g;
  g.load(param_path, bin_path);
  auto& L = g.layers;
  const int n = (int)L.size();
  const size_t es = prec_ == LP_FP16 ? 2 : 4;

  tensors_.clear(); blob2tensor_.clear(); buffers_.clear(); convs_.clear(); ops_.clear(); levels_.clear();
  bnecks_.clear(); dws_.clear(); attns_.clear(); heads_.clear(); c2fs_.clear(); c2f_io_.clear(); s2cs_.clear(); sppfs_.clear();
  fused_head_ = false;
  loaded_ = false;

  std::map<std::string, int> producer;
  std::map<std::string, std::vector<int>> consumers;
  for (int i = 0; i < n; ++i) {
    for (auto& o : L[i].outputs) producer[o] = i;
    for (auto& in : L[i].inputs) consumers[in].push_back(i);
  }
  auto prod_type = [&](const std::string& b) -> std::string {
    auto it = producer.find(b);
    return it == producer.end() ? std::string() : L[it->second].type;
  };

  // ---- find the Detect tail ---------------------------------------------------------------
  std::vector<int> head_cats;
  int first_tail = n;
  for (int i = 0; i < n; ++i) {
    if (L[i].type != "Reshape" || L[i].inputs.empty()) continue;
    auto it = producer.find(L[i].inputs[0]);
    if (it == producer.end()) continue;
    const NcnnLayer& c = L[it->second];
    if (c.type == "Concat" && c.inputs.size() == 2 && prod_type(c.inputs[0]) == "Convolution" &&
        prod_type(c.inputs[1]) == "Convolution") {
      head_cats.push_back(it->second);
      first_tail = std::min(first_tail, i);
    }
  }
  LP_CHECK(!head_cats.empty() && head_cats.size() <= 4, LP_ERR_GRAPH, "no YOLOv8-style Detect head found in %s", param_path.c_str());
  std::set<int> head_cat_set(head_cats.begin(), head_cats.end());
  auto is_tail = [&](int i) { return i >= first_tail || head_cat_set.count(i) || L[i].type == "MemoryData"; };

  // ---- YOLO11 C2PSA attention blocks: a fixed 12-layer sequence becomes one ATTN op (launch_psa_attention) --------
  struct AttnBlock { int heads, dk, dv, hw; float scale; int dw; std::string in_blob, out_blob; };
  std::map<int, AttnBlock> attn_at;
  std::vector<char> in_attn(n, 0);
  {
    const char* seq[12] = {"Reshape", "Slice", "Split", "Permute", "MatMul", "BinaryOp", "Softmax", "MatMul", "Reshape", "Reshape",
                           "ConvolutionDepthWise", "BinaryOp"};
    for (int i = 0; i + 12 <= n; ++i) {
      if (is_tail(i) || L[i].type != "Reshape" || L[i].ipar(2, 0) <= 0 || prod_type(L[i].inputs[0]) != "Convolution") continue;
      bool ok = true;
      for (int q = 0; q < 12 && ok; ++q) ok = L[i + q].type == seq[q];
      if (!ok) continue;
      AttnBlock a;
      a.heads = L[i].ipar(2); a.hw = L[i].ipar(0);
      auto it = L[i + 1].arrays.find(0);
      ok = it != L[i + 1].arrays.end() && it->second.size() == 3 && L[i + 1].ipar(1, 0) == 1 && it->second[0] == it->second[1];
      if (ok) { a.dk = (int)it->second[0]; a.dv = (int)it->second[2]; }
      ok = ok && L[i].ipar(1) == 2 * a.dk + a.dv && L[i + 3].ipar(0, 0) == 1 && L[i + 5].ipar(0, 0) == 2 && L[i + 5].ipar(1, 0) == 1 &&
           L[i + 7].ipar(0, 0) == 1 && L[i + 10].ipar(1, 1) == 3 && L[i + 10].ipar(3, 1) == 1 && L[i + 10].ipar(4, 0) == 1 &&
           L[i + 10].ipar(7, 1) == a.heads * a.dv && L[i + 10].ipar(0) == a.heads * a.dv && L[i + 11].ipar(0, 0) == 0 &&
           L[i + 11].inputs.size() == 2;
      LP_CHECK(ok, LP_ERR_GRAPH, "attention block at %s has an unexpected shape", L[i].name.c_str());
      a.scale = (float)L[i + 5].fpar(2, 1.0);
      a.dw = i + 10;
      a.in_blob = L[i].inputs[0];
      a.out_blob = L[i + 11].outputs[0];
      attn_at[i] = a;
      for (int q = 0; q < 12; ++q) in_attn[i + q] = 1;
    }
  }

  // ---- aliases (Split) and Swish fusion -----------------------------------------------------
  std::map<std::string, std::string> alias;
  std::function<std::string(const std::string&)> canon = [&](const std::string& b) {
    std::string c = b;
    while (alias.count(c)) c = alias[c];
    return c;
  };
  std::vector<int> fused_act(n, ACT_NONE);
  std::vector<std::string> conv_out(n);
  std::vector<char> skip(n, 0);
  for (int i = 0; i < n; ++i) {
    if (is_tail(i) || in_attn[i]) continue;
    if (L[i].type == "Split")
      for (auto& o : L[i].outputs) alias[o] = L[i].inputs[0];
    if (L[i].type == "Convolution" || L[i].type == "ConvolutionDepthWise") {
      const std::string& x = L[i].outputs[0];
      conv_out[i] = x;
      auto& cs = consumers[x];
      if (cs.size() == 1 && L[cs[0]].type == "Swish" && !is_tail(cs[0])) {
        fused_act[i] = ACT_SILU;
        conv_out[i] = L[cs[0]].outputs[0];
        skip[cs[0]] = 1;
      }
    }
  }
  std::map<std::string, std::vector<int>> canon_consumers;
  for (int i = 0; i < n; ++i) {
    if (L[i].type == "Split" || skip[i]) continue;
    if (in_attn[i]) {  // the block as a whole consumes its input blob
      if (attn_at.count(i)) canon_consumers[canon(attn_at[i].in_blob)].push_back(i);
      continue;
    }
    for (auto& in : L[i].inputs) canon_consumers[canon(in)].push_back(i);
  }
  // Slice sizes keyed by the canonical input blob (needed before the parent's layout is fixed)
  std::map<std::string, std::vector<int>> slice_sizes;
  for (int i = 0; i < n; ++i) {
    if (is_tail(i) || in_attn[i] || L[i].type != "Slice") continue;
    LP_CHECK(L[i].ipar(1, 0) == 0, LP_ERR_GRAPH, "Slice %s: only channel slices supported", L[i].name.c_str());
    auto it = L[i].arrays.find(0);
    LP_CHECK(it != L[i].arrays.end() && it->second.size() == L[i].outputs.size(), LP_ERR_GRAPH, "Slice %s: bad size list", L[i].name.c_str());
    std::vector<int> sz;
    for (double v : it->second) sz.push_back((int)v);
    slice_sizes[canon(L[i].inputs[0])] = sz;
  }

  // ---- pass A: shapes and tensors ---------------------------------------------------------
  auto new_tensor = [&](const std::string& name, int C, int H, int W) {
    Tensor t;
    t.name = name; t.C = C; t.H = H; t.W = W;
    auto it = slice_sizes.find(name);
    if (it != slice_sizes.end()) {
      std::vector<int> sz = it->second;
      int known = 0, autos = 0;
      for (int s : sz) { if (s == -233) ++autos; else known += s; }
      for (int& s : sz) if (s == -233) s = (C - known) / autos;
      int sum = 0;
      for (int s : sz) sum += s;
      LP_CHECK(sum == C, LP_ERR_GRAPH, "Slice sizes of blob %s do not add up to %d", name.c_str(), C);
      t.segs = sz;
    } else {
      t.segs = {C};
    }
    tensors_.push_back(t);
    blob2tensor_[name] = (int)tensors_.size() - 1;
    return (int)tensors_.size() - 1;
  };
  auto get = [&](const std::string& blob) {
    auto it = blob2tensor_.find(canon(blob));
    LP_CHECK(it != blob2tensor_.end(), LP_ERR_GRAPH, "blob %s used before it is produced", blob.c_str());
    return it->second;
  };
  struct ConvInfo { int tin = -1, tout = -1; };
  std::vector<ConvInfo> cinfo(n);
  int input_tensor = -1;
  for (int i = 0; i < n; ++i) {
    if (is_tail(i) && L[i].type != "Convolution") continue;
    if (i >= first_tail) continue;  // DFL conv etc.
    const NcnnLayer& l = L[i];
    if (skip[i]) continue;
    if (in_attn[i]) {
      if (attn_at.count(i)) {
        const AttnBlock& a = attn_at[i];
        const Tensor tin = tensors_[get(a.in_blob)];
        LP_CHECK(tin.C == a.heads * (2 * a.dk + a.dv) && tin.H * tin.W == a.hw && tin.segs.size() == 1, LP_ERR_GRAPH,
                 "attention block at %s: qkv blob is %dx%dx%d", l.name.c_str(), tin.C, tin.H, tin.W);
        new_tensor(a.out_blob, a.heads * a.dv, tin.H, tin.W);
      }
      continue;
    }
    if (l.type == "ConvolutionDepthWise") {
      const int tin = get(l.inputs[0]);
      LP_CHECK(l.ipar(1, 1) == 3 && l.ipar(3, 1) == 1 && l.ipar(4, 0) == 1 && l.ipar(2, 1) == 1 && l.ipar(7, 1) == l.ipar(0) &&
                   l.ipar(0) == tensors_[tin].C && l.ipar(9, 0) == 0 && l.ipar(8, 0) == 0, LP_ERR_GRAPH,
               "ConvolutionDepthWise %s: only depthwise 3x3/s1/p1 without fused activation supported", l.name.c_str());
      cinfo[i].tin = tin;
      cinfo[i].tout = new_tensor(conv_out[i], l.ipar(0), tensors_[tin].H, tensors_[tin].W);
      continue;
    }
    if (l.type == "Input") {
      input_tensor = new_tensor(l.outputs[0], 3, S_, S_);
    } else if (l.type == "Convolution") {
      const int tin = get(l.inputs[0]);
      const int k = l.ipar(1, 1), s = l.ipar(3, 1), pad = l.ipar(4, 0), dil = l.ipar(2, 1);
      // pad k/2 everywhere; the image conv may also be YOLOv5's 6x6/s2/p2 stem (any k, s, p: generic stem kernel)
      LP_CHECK(l.ipar(11, k) == k && l.ipar(13, s) == s && l.ipar(14, pad) == pad && dil == 1 && (pad == k / 2 || tin == input_tensor), LP_ERR_GRAPH,
               "Convolution %s: only square k with pad k/2, dilation 1 supported", l.name.c_str());
      LP_CHECK(l.in_ch == tensors_[tin].C, LP_ERR_GRAPH, "Convolution %s: weight expects %d input channels, blob has %d",
               l.name.c_str(), l.in_ch, tensors_[tin].C);
      // fused activation (9 != 0, e.g. after ncnnoptimize), int8 weights (8), asymmetric / valued padding (15, 16, 18): the
      // kernels implement none of them, and ignoring the parameter would silently compute a different network
      LP_CHECK(l.ipar(9, 0) == 0 && l.ipar(8, 0) == 0 && l.ipar(15, pad) == pad && l.ipar(16, pad) == pad && l.fpar(18, 0.0) == 0.0,
               LP_ERR_GRAPH, "Convolution %s: fused activation_type / int8 / asymmetric padding parameters are unsupported", l.name.c_str());
      const int Ho = (tensors_[tin].H + 2 * pad - k) / s + 1, Wo = (tensors_[tin].W + 2 * pad - k) / s + 1;
      cinfo[i].tin = tin;
      cinfo[i].tout = new_tensor(conv_out[i], l.ipar(0), Ho, Wo);
    } else if (l.type == "Swish") {
      throw Error(LP_ERR_GRAPH, fmt("stand-alone Swish %s unsupported", l.name.c_str()));
    } else if (l.type == "Split") {
      const int t = get(l.inputs[0]);
      for (auto& o : l.outputs) blob2tensor_[o] = t;
    } else if (l.type == "Slice") {
      const int tin = get(l.inputs[0]);
      const std::vector<int> sz = tensors_[tin].segs;
      LP_CHECK(sz.size() == l.outputs.size(), LP_ERR_GRAPH, "Slice %s: layout mismatch", l.name.c_str());
      for (size_t j = 0; j < l.outputs.size(); ++j) {
        const int t = new_tensor(l.outputs[j], sz[j], tensors_[tin].H, tensors_[tin].W);
        tensors_[t].parent = tin;
        tensors_[t].parent_seg = (int)j;
      }
    } else if (l.type == "Concat") {
      LP_CHECK(l.ipar(0, 0) == 0, LP_ERR_GRAPH, "Concat %s: only channel concat supported", l.name.c_str());
      int C = 0;
      std::vector<int> segs;
      const int t0 = get(l.inputs[0]);
      for (auto& in : l.inputs) {
        const Tensor& t = tensors_[get(in)];
        LP_CHECK(t.H == tensors_[t0].H && t.W == tensors_[t0].W, LP_ERR_GRAPH, "Concat %s: spatial mismatch", l.name.c_str());
        C += t.C;
        segs.insert(segs.end(), t.segs.begin(), t.segs.end());
      }
      const int t = new_tensor(l.outputs[0], C, tensors_[t0].H, tensors_[t0].W);
      if (tensors_[t].segs.size() == 1) tensors_[t].segs = segs;
    } else if (l.type == "BinaryOp") {
      LP_CHECK(l.ipar(0, 0) == 0 && l.inputs.size() == 2 && l.ipar(1, 0) == 0, LP_ERR_GRAPH, "BinaryOp %s: only tensor add supported", l.name.c_str());
      const Tensor a = tensors_[get(l.inputs[0])], b = tensors_[get(l.inputs[1])];
      LP_CHECK(a.C == b.C && a.H == b.H && a.W == b.W, LP_ERR_GRAPH, "BinaryOp %s: shape mismatch", l.name.c_str());
      new_tensor(l.outputs[0], a.C, a.H, a.W);
    } else if (l.type == "Pooling") {
      LP_CHECK(l.ipar(0, 0) == 0 && l.ipar(1) == 5 && l.ipar(2, 1) == 1 && l.ipar(3, 0) == 2, LP_ERR_GRAPH,
               "Pooling %s: only the SPPF 5x5/s1/p2 max pool is supported", l.name.c_str());
      const Tensor a = tensors_[get(l.inputs[0])];
      new_tensor(l.outputs[0], a.C, a.H, a.W);
    } else if (l.type == "Interp") {
      LP_CHECK(l.ipar(0, 0) == 1 && l.fpar(1, 1.0) == 2.0 && l.fpar(2, 1.0) == 2.0, LP_ERR_GRAPH, "Interp %s: only nearest x2 supported", l.name.c_str());
      const Tensor a = tensors_[get(l.inputs[0])];
      new_tensor(l.outputs[0], a.C, 2 * a.H, 2 * a.W);
    } else {
      throw Error(LP_ERR_GRAPH, fmt("unsupported NCNN layer type %s (%s)", l.type.c_str(), l.name.c_str()));
    }
  }
  LP_CHECK(input_tensor >= 0, LP_ERR_GRAPH, "graph has no Input layer");
  for (auto& t : tensors_) {
    t.Cp = 0;
    for (int s : t.segs) t.Cp += round_up(s, 8);
  }

  auto alloc_buffer = [&](int Cp, int H, int W) {
    buffers_.emplace_back();
    Buffer& b = buffers_.back();
    b.Cp = Cp; b.H = H; b.W = W;
    b.mem.alloc((size_t)maxB_ * H * W * Cp * es);
    return (int)buffers_.size() - 1;
  };

  // ---- pass B: place concat inputs inside the concat buffer -----------------------------------
  struct CopyJob { int layer, src, dst_buf, dst_off; };
  std::vector<CopyJob> copies;
  for (int i = 0; i < n; ++i) {
    if (is_tail(i) || L[i].type != "Concat") continue;
    const int tout = get(L[i].outputs[0]);
    Tensor& O = tensors_[tout];
    if (O.buf < 0) { O.buf = alloc_buffer(O.Cp, O.H, O.W); O.off = 0; }
    int o = O.off;
    size_t j = 0;
    while (j < L[i].inputs.size()) {
      const int t = get(L[i].inputs[j]);
      Tensor& T = tensors_[t];
      if (T.parent >= 0) {
        Tensor& P = tensors_[T.parent];
        const size_t m = P.segs.size();
        bool whole = T.parent_seg == 0 && j + m <= L[i].inputs.size() && P.buf < 0;
        for (size_t q = 0; whole && q < m; ++q) {
          const Tensor& Q = tensors_[get(L[i].inputs[j + q])];
          whole = Q.parent == T.parent && Q.parent_seg == (int)q;
        }
        if (whole) {
          P.buf = O.buf; P.off = o;
          o += P.Cp;
          j += m;
          continue;
        }
        copies.push_back({i, t, O.buf, o});
      } else if (T.buf < 0) {
        T.buf = O.buf; T.off = o;
      } else {
        copies.push_back({i, t, O.buf, o});
      }
      o += T.Cp;
      ++j;
    }
    LP_CHECK(o - O.off == O.Cp, LP_ERR_GRAPH, "Concat %s: layout bookkeeping error", L[i].name.c_str());
  }
  auto ensure_buffer = [&](int t) {
    Tensor& T = tensors_[t];
    LP_CHECK(T.parent < 0, LP_ERR_GRAPH, "blob %s is a slice and cannot be produced directly", T.name.c_str());
    if (T.buf < 0) { T.buf = alloc_buffer(T.Cp, T.H, T.W); T.off = 0; }
    T.materialised = true;
  };

  // ---- pass D: emit ops ------------------------------------------------------------------------
  std::vector<char> done(n, 0);
  std::map<int, int> fuse_up;  // 1x1 conv layer -> half-resolution tensor it upsamples on the fly
  macs_ = 0;
  const double esd = (double)es;

  // ---- whole-C2f launches (c2f_kernels.hip; fp16 MFMA plan, LITEPI_NO_C2F=1: off).  match_c2f recognises, from its cv1, a
  //      complete C2f module: Convolution 1x1 + Swish -> Slice (c | c) -> n x [3x3 + Swish -> 3x3 + Swish -> BinaryOp add] ->
  //      Concat(y0 .. y_{n+1}) (zero-copy, pass B) -> Convolution 1x1 + Swish.  try_c2f also folds in the stride-2 3x3 conv
  //      in front of the module and the SPPF behind it when a whole-image configuration exists for the level.
  struct C2fMatch {
    int cv1 = -1, slice = -1, a[2] = {-1, -1}, b[2] = {-1, -1}, add[2] = {-1, -1}, cat = -1, cv2 = -1, nb = 0, c = 0;
    int t_cat = -1;
    std::vector<int> ys;
  };
  // (handles built for fewer than 4 images keep the layer plan: a whole-C2f launch is one long workgroup chain per tile, and
  //  with a handful of tiles that chain is the latency -- batch-1 detect 0.475 ms against 0.43 ms; LITEPI_C2F_MIN_BATCH overrides)
  static const int c2f_min_batch = getenv("LITEPI_C2F_MIN_BATCH") ? atoi(getenv("LITEPI_C2F_MIN_BATCH")) : 4;
  const bool c2f_on = !getenv("LITEPI_NO_C2F") && prec_ == LP_FP16 && impl_ == IMPL_MFMA && maxB_ >= c2f_min_batch;
  auto is_silu_conv = [&](int j, int k, int s) {
    return j >= 0 && L[j].type == "Convolution" && !is_tail(j) && !done[j] && L[j].ipar(1, 1) == k && L[j].ipar(3, 1) == s &&
           fused_act[j] == ACT_SILU && !L[j].bias.empty() && cinfo[j].tin >= 0 && cinfo[j].tin != input_tensor;
  };
  auto sole_consumer = [&](int t) {
    auto& cs = canon_consumers[tensors_[t].name];
    return cs.size() == 1 ? cs[0] : -1;
  };
  auto zero_copy_concat = [&](int cc) {
    for (auto& cj : copies)
      if (cj.layer == cc) return false;
    return true;
  };
  auto match_c2f = [&](int i1, C2fMatch& m) -> bool {
    if (!is_silu_conv(i1, 1, 1)) return false;
    const int t1 = cinfo[i1].tout;
    const Tensor& T1 = tensors_[t1];
    if (T1.segs.size() != 2 || T1.segs[0] != T1.segs[1] || T1.parent >= 0) return false;
    const int c = T1.segs[0];
    if (c % 8 != 0 || tensors_[cinfo[i1].tin].Cp != L[i1].in_ch) return false;
    const int sl = sole_consumer(t1);
    if (sl < 0 || L[sl].type != "Slice" || L[sl].outputs.size() != 2) return false;
    const int ty0 = get(L[sl].outputs[0]), ty1 = get(L[sl].outputs[1]);
    const int cc = sole_consumer(ty0);
    if (cc < 0 || L[cc].type != "Concat" || is_tail(cc) || !zero_copy_concat(cc)) return false;
    m = C2fMatch();
    m.cv1 = i1; m.slice = sl; m.cat = cc; m.c = c;
    m.ys = {ty0, ty1};
    int cur = ty1;
    for (;;) {
      int ja = -1, jadd = -1;
      bool has_cat = false;
      for (int q : canon_consumers[tensors_[cur].name]) {
        if (q == cc) has_cat = true;
        else if (L[q].type == "Convolution" && ja == -1) ja = q;
        else if (L[q].type == "BinaryOp" && jadd == -1) jadd = q;
        else return false;
      }
      if (!has_cat) return false;
      if (ja == -1 && jadd == -1) break;
      if (ja < 0 || jadd < 0 || m.nb >= 2) return false;
      if (!is_silu_conv(ja, 3, 1) || L[ja].ipar(0) != c || L[ja].in_ch != c) return false;
      const int jb = sole_consumer(cinfo[ja].tout);
      if (!is_silu_conv(jb, 3, 1) || L[jb].ipar(0) != c || L[jb].in_ch != c) return false;
      if (sole_consumer(cinfo[jb].tout) != jadd || is_tail(jadd) || done[jadd]) return false;
      const NcnnLayer& add = L[jadd];
      if (add.ipar(0, 0) != 0 || add.inputs.size() != 2 || add.ipar(1, 0) != 0) return false;
      const int ta = get(add.inputs[0]), tb = get(add.inputs[1]);
      if (!((ta == cinfo[jb].tout && tb == cur) || (tb == cinfo[jb].tout && ta == cur))) return false;
      m.a[m.nb] = ja; m.b[m.nb] = jb; m.add[m.nb] = jadd;
      ++m.nb;
      cur = get(add.outputs[0]);
      m.ys.push_back(cur);
    }
    if (m.nb < 1 || L[cc].inputs.size() != m.ys.size()) return false;
    for (size_t q = 0; q < m.ys.size(); ++q)
      if (get(L[cc].inputs[q]) != m.ys[q]) return false;
    m.t_cat = get(L[cc].outputs[0]);
    const Tensor& TC = tensors_[m.t_cat];
    if (TC.parent >= 0 || TC.buf < 0 || TC.Cp != (2 + m.nb) * c || T1.buf != TC.buf || T1.off != TC.off) return false;
    for (size_t q = 2; q < m.ys.size(); ++q) {
      const Tensor& Y = tensors_[m.ys[q]];
      if (Y.buf != TC.buf || Y.off != TC.off + (int)q * c || Y.Cp != c) return false;
    }
    m.cv2 = sole_consumer(m.t_cat);
    if (!is_silu_conv(m.cv2, 1, 1)) return false;
    const Tensor& TO = tensors_[cinfo[m.cv2].tout];
    if (TO.Cp != L[m.cv2].ipar(0) || TO.parent >= 0) return false;
    for (int q : canon_consumers[TO.name])
      if (L[q].type == "BinaryOp") return false;
    return true;
  };
  auto c2f_shape = [&](const C2fMatch& m, int mode, int ks2) {
    C2fShape s;
    s.C = m.c; s.NB = m.nb; s.COUT = L[m.cv2].ipar(0); s.MODE = mode; s.KS2 = ks2;
    const int tin = cinfo[m.cv1].tin;
    if (fuse_up.count(m.cv1)) {
      s.UP = 1;
      s.KA = tensors_[fuse_up[m.cv1]].Cp;
      s.KB = tensors_[tin].Cp - s.KA;
    } else {
      s.KB = tensors_[tin].Cp;
    }
    return s;
  };
  auto c2f_plain_ok = [&](int i1) {   // a stand-alone C2f launch exists for the module whose cv1 is layer i1
    C2fMatch m;
    if (!c2f_on || !match_c2f(i1, m)) return false;
    const Tensor& T = tensors_[cinfo[i1].tout];
    return C2fLayer::supported(c2f_shape(m, 0, 0), T.H, T.W);
  };
  // NCNN [out][in][kh][kw] -> [out][tap][in] (3x3) / [out][in] (1x1); these modules have no padded channel segments
  auto conv_w = [&](int j) {
    const NcnnLayer& lc = L[j];
    const int k = lc.ipar(1, 1), co = lc.ipar(0), ci = lc.in_ch, taps = k * k;
    std::vector<float> w((size_t)co * taps * ci);
    for (int o = 0; o < co; ++o)
      for (int c = 0; c < ci; ++c)
        for (int t = 0; t < taps; ++t) w[((size_t)o * taps + t) * ci + c] = lc.weight[((size_t)o * ci + c) * taps + t];
    return w;
  };
  auto try_c2f = [&](int i) -> bool {
    if (!c2f_on) return false;
    C2fMatch m;
    int i0 = -1, xcat = -1, mode = 0;
    if (is_silu_conv(i, 3, 2)) {
      // stride-2 conv -> [Concat(x, other) ->] cv1: whole-image configurations only
      const int tx = cinfo[i].tout;
      if (tensors_[tx].segs.size() != 1 || tensors_[tx].Cp != L[i].ipar(0) || tensors_[cinfo[i].tin].Cp != L[i].in_ch) return false;
      int jn = sole_consumer(tx);
      if (jn >= 0 && L[jn].type == "Concat" && !is_tail(jn)) {
        const int tcat_in = get(L[jn].outputs[0]);
        if (L[jn].inputs.size() != 2 || get(L[jn].inputs[0]) != tx || !zero_copy_concat(jn) || tensors_[tx].buf != tensors_[tcat_in].buf ||
            tensors_[tx].off != tensors_[tcat_in].off || tensors_[tcat_in].parent >= 0)
          return false;
        // the launch is emitted HERE, at the stride-2 conv's position, and cv1 reads the whole Concat(x, other): `other` must
        // have been produced by then.  Layers are emitted in file order, so its producer has to precede this conv (true for
        // the reference's graphs: P5 / F4 come earlier); otherwise the module is retried at its cv1, behind the Concat.
        {
          auto po = producer.find(L[jn].inputs[1]);
          if (po != producer.end() && po->second >= i) return false;
        }
        xcat = jn;
        jn = sole_consumer(tcat_in);
      }
      if (jn < 0 || !match_c2f(jn, m) || fuse_up.count(jn) || L[i].ipar(0) != 2 * m.c) return false;
      i0 = i;
      mode = 1;
    } else if (!match_c2f(i, m)) {
      return false;
    }
    const int tout = cinfo[m.cv2].tout;
    const int Hh = tensors_[tout].H, Ww = tensors_[tout].W;
    // SPPF behind the module: cv1 (1x1) -> three chained 5x5 pools -> zero-copy Concat(s, p1, p2, p3) -> cv2 (1x1)
    int js1 = -1, js2 = -1, spcat = -1, pools[3] = {-1, -1, -1};
    if (mode == 1) {
      const int j = sole_consumer(tout);
      if (is_silu_conv(j, 1, 1) && L[j].ipar(0) == m.c && tensors_[cinfo[j].tout].segs.size() == 1) {
        const int ts = cinfo[j].tout;
        int cur = ts, cc = -1;
        bool ok = true;
        std::vector<int> chain = {ts};
        for (int q = 0; q < 3 && ok; ++q) {
          int jp = -1;
          for (int cq : canon_consumers[tensors_[cur].name]) {
            if (L[cq].type == "Pooling" && jp < 0) jp = cq;
            else if (L[cq].type == "Concat" && (cc < 0 || cc == cq)) cc = cq;
            else ok = false;
          }
          ok = ok && jp >= 0 && !done[jp];
          if (ok) { pools[q] = jp; cur = get(L[jp].outputs[0]); chain.push_back(cur); }
        }
        if (ok) {
          for (int cq : canon_consumers[tensors_[cur].name]) ok = ok && cq == cc;
          ok = ok && cc >= 0 && !is_tail(cc) && zero_copy_concat(cc) && L[cc].inputs.size() == 4;
          for (int q = 0; q < 4 && ok; ++q) ok = get(L[cc].inputs[q]) == chain[q];
        }
        if (ok) {
          const int tc2 = get(L[cc].outputs[0]);
          const Tensor& TC2 = tensors_[tc2];
          ok = TC2.parent < 0 && TC2.buf >= 0 && TC2.Cp == 4 * m.c;
          for (int q = 0; q < 4 && ok; ++q) ok = tensors_[chain[q]].buf == TC2.buf && tensors_[chain[q]].off == TC2.off + q * m.c && tensors_[chain[q]].Cp == m.c;
          const int j2 = ok ? sole_consumer(tc2) : -1;
          ok = ok && is_silu_conv(j2, 1, 1) && L[j2].ipar(0) == L[m.cv2].ipar(0) && tensors_[cinfo[j2].tout].Cp == L[j2].ipar(0) &&
               tensors_[cinfo[j2].tout].parent < 0;
          if (ok) { js1 = j; js2 = j2; spcat = cc; }
        }
      }
      if (js1 >= 0) mode = 2;
    }
    C2fShape sh = c2f_shape(m, mode, i0 >= 0 ? L[i0].in_ch : 0);
    if (!C2fLayer::supported(sh, Hh, Ww)) {
      if (mode == 2) { mode = 1; sh = c2f_shape(m, 1, L[i0].in_ch); js1 = js2 = -1; }
      if (!C2fLayer::supported(sh, Hh, Ww)) return false;   // (a stride-2 conv falls through to its own kernel; the module is tried again at its cv1)
    }
    // ---- build
    std::vector<float> w_cv1 = conv_w(m.cv1), w_cv2 = conv_w(m.cv2), w_a[2], w_b[2], w_s2, w_sp1, w_sp2;
    C2fLayer::Src src;
    src.cv1 = &w_cv1; src.cv1_b = &L[m.cv1].bias;
    src.cv2 = &w_cv2; src.cv2_b = &L[m.cv2].bias;
    for (int k = 0; k < m.nb; ++k) {
      w_a[k] = conv_w(m.a[k]); w_b[k] = conv_w(m.b[k]);
      src.a[k] = &w_a[k]; src.a_b[k] = &L[m.a[k]].bias;
      src.bb[k] = &w_b[k]; src.bb_b[k] = &L[m.b[k]].bias;
    }
    if (mode >= 1) { w_s2 = conv_w(i0); src.s2 = &w_s2; src.s2_b = &L[i0].bias; }
    if (mode == 2) {
      w_sp1 = conv_w(js1); w_sp2 = conv_w(js2);
      src.sp1 = &w_sp1; src.sp1_b = &L[js1].bias;
      src.sp2 = &w_sp2; src.sp2_b = &L[js2].bias;
    }
    c2fs_.emplace_back(new C2fLayer());
    C2fLayer& cl = *c2fs_.back();
    cl.name = (i0 >= 0 ? L[i0].name + "+" : std::string()) + L[m.cv1].name + ".." + L[m.cv2].name + (mode == 2 ? "+sppf" : "");
    cl.build(sh, Hh, Ww, src);
    C2fIO io;
    io.src1 = cinfo[m.cv1].tin;
    if (sh.UP) { io.src0 = fuse_up[m.cv1]; io.up_c = sh.KA; }
    io.cat = m.t_cat;
    io.out = tout;
    ensure_buffer(tout);
    // (the y segments cv2 takes from LDS are only stored in the bisect mode, LITEPI_C2F_STORE_ALL=1)
    for (size_t q = 2; q < m.ys.size(); ++q) tensors_[m.ys[q]].materialised = getenv("LITEPI_C2F_STORE_ALL") != nullptr || !cl.cv2_from_lds();
    if (mode >= 1) {
      io.s2_in = cinfo[i0].tin;
      io.x = cinfo[i0].tout;
      if (xcat < 0) ensure_buffer(io.x);
      else tensors_[io.x].materialised = true;
    }
    double bytes = ((double)tensors_[io.src1].C * Hh * Ww + (double)tensors_[tout].C * Hh * Ww) * esd;
    if (sh.UP) bytes -= 0.75 * sh.KA * Hh * Ww * esd;   // the upsampled segment is read at half resolution
    if (mode >= 1) bytes += ((double)sh.KS2 * 4 - (xcat < 0 ? (double)tensors_[io.src1].C : (double)tensors_[io.x].C)) * Hh * Ww * esd;
    if (mode == 2) {
      io.cat2 = get(L[spcat].outputs[0]);
      io.out2 = cinfo[js2].tout;
      ensure_buffer(io.out2);
      if (getenv("LITEPI_C2F_STORE_ALL")) {   // (s and the pooled maps stay in LDS otherwise: sppf_tail)
        for (int q = 0; q < 3; ++q) tensors_[get(L[pools[q]].outputs[0])].materialised = true;
        tensors_[cinfo[js1].tout].materialised = true;
      }
      bytes += ((double)tensors_[io.out2].C - (double)tensors_[tout].C) * Hh * Ww * esd;
    }
    if (!getenv("LITEPI_C2F_STORE_ALL")) {
      // every tensor between the module's input and its output: lp_debug_blob must not hand out their (possibly never
      // written) storage -- which of them a configuration stores is the kernel's business (c2f_kernels.hip)
      auto inside = [&](int t) { if (t >= 0 && t != tout && (mode != 2 || t != io.out2)) tensors_[t].in_c2f = true; };
      inside(cinfo[m.cv1].tout);
      inside(m.t_cat);
      for (int t : m.ys) inside(t);
      for (int k = 0; k < m.nb; ++k) {
        inside(cinfo[m.a[k]].tout); inside(cinfo[m.b[k]].tout);
        for (auto& o : L[m.add[k]].outputs) inside(get(o));
      }
      if (mode >= 1) inside(io.x);
      if (mode == 2) {
        tensors_[tout].in_c2f = true;   // the C2f's own output: SPPF.cv1 reads it from LDS
        inside(cinfo[js1].tout); inside(io.cat2);
        for (int q = 0; q < 3; ++q) inside(get(L[pools[q]].outputs[0]));
      }
    }
    c2f_io_.push_back(io);
    macs_ += cl.macs_per_image;
    DetOp op;
    op.kind = DetOp::C2F; op.conv = (int)c2fs_.size() - 1; op.layer = cl.name;
    op.in = io.src1; op.out = mode == 2 ? io.out2 : tout;
    op.flops = 2.0 * cl.macs_per_image;
    op.bytes = bytes;
    ops_.push_back(op);
    done[m.cv1] = done[m.cv2] = 1;
    for (int k = 0; k < m.nb; ++k) done[m.a[k]] = done[m.b[k]] = done[m.add[k]] = 1;
    if (i0 >= 0) done[i0] = 1;
    if (mode == 2) { done[js1] = done[js2] = 1; done[pools[0]] = done[pools[1]] = done[pools[2]] = 1; }
    return true;
  };

  // ---- SPPF in one launch (sppf_kernel; the widths the whole-image C2f kernel does not take along: v2's 192 -> 96 -> 192 @20x20):
  //      cv1 (1x1 + Swish) -> three chained 5x5 pools -> zero-copy Concat(s, p1, p2, p3) -> cv2 (1x1 + Swish); s and the pooled
  //      maps stay in LDS, the concat buffer is never written
  auto try_sppf = [&](int i) -> bool {
    if (!c2f_on || !is_silu_conv(i, 1, 1)) return false;
    const int tin = cinfo[i].tin, ts = cinfo[i].tout;
    if (tensors_[ts].segs.size() != 1 || tensors_[tin].Cp != L[i].in_ch || tensors_[ts].Cp != L[i].ipar(0)) return false;
    int cur = ts, cc = -1, pools[3] = {-1, -1, -1};
    std::vector<int> chain = {ts};
    for (int q = 0; q < 3; ++q) {
      int jp = -1;
      for (int cq : canon_consumers[tensors_[cur].name]) {
        if (L[cq].type == "Pooling" && jp < 0) jp = cq;
        else if (L[cq].type == "Concat" && (cc < 0 || cc == cq)) cc = cq;
        else return false;
      }
      if (jp < 0 || done[jp]) return false;
      pools[q] = jp; cur = get(L[jp].outputs[0]); chain.push_back(cur);
    }
    for (int cq : canon_consumers[tensors_[cur].name])
      if (cq != cc) return false;
    if (cc < 0 || is_tail(cc) || !zero_copy_concat(cc) || L[cc].inputs.size() != 4) return false;
    for (int q = 0; q < 4; ++q)
      if (get(L[cc].inputs[q]) != chain[q]) return false;
    const int tc2 = get(L[cc].outputs[0]);
    const int c = L[i].ipar(0);
    if (tensors_[tc2].parent >= 0 || tensors_[tc2].Cp != 4 * c) return false;
    const int j2 = sole_consumer(tc2);
    if (!is_silu_conv(j2, 1, 1)) return false;
    const int tout = cinfo[j2].tout;
    if (tensors_[tout].Cp != L[j2].ipar(0) || tensors_[tout].parent >= 0 || tensors_[tout].segs.size() != 1) return false;
    for (int q : canon_consumers[tensors_[tout].name])
      if (L[q].type == "BinaryOp") return false;
    const int Hh = tensors_[ts].H, Ww = tensors_[ts].W;
    if (!SppfLayer::supported(L[i].in_ch, c, L[j2].ipar(0), Hh, Ww)) return false;
    sppfs_.emplace_back(new SppfLayer());
    SppfLayer& sl = *sppfs_.back();
    sl.name = L[i].name + "+pools+" + L[j2].name;
    sl.build(L[i].in_ch, c, L[j2].ipar(0), Hh, Ww, conv_w(i), L[i].bias, conv_w(j2), L[j2].bias);
    ensure_buffer(tout);
    for (int q = 0; q < 4; ++q) tensors_[chain[q]].in_c2f = true;   // never written: lp_debug_blob must not hand them out
    tensors_[tc2].in_c2f = true;
    macs_ += sl.macs_per_image;
    DetOp op;
    op.kind = DetOp::SPPFUSED; op.conv = (int)sppfs_.size() - 1; op.layer = sl.name; op.in = tin; op.out = tout;
    op.flops = 2.0 * sl.macs_per_image;
    op.bytes = ((double)tensors_[tin].C + (double)tensors_[tout].C) * Hh * Ww * esd + (double)(L[i].weight.size() + L[j2].weight.size()) * esd;
    ops_.push_back(op);
    done[i] = done[j2] = done[pools[0]] = done[pools[1]] = done[pools[2]] = 1;
    return true;
  };

  for (int i = 0; i < first_tail; ++i) {
    if ((is_tail(i) && L[i].type != "Convolution") || skip[i] || done[i]) continue;
    const NcnnLayer& l = L[i];
    if (in_attn[i]) {
      if (!attn_at.count(i)) continue;
      const AttnBlock& a = attn_at[i];
      const int tin = get(a.in_blob), tout = get(a.out_blob);
      ensure_buffer(tout);
      const NcnnLayer& dw = L[a.dw];
      const int Cv = a.heads * a.dv;
      attns_.emplace_back();
      AttnLayer& A = attns_.back();
      A.heads = a.heads; A.dk = a.dk; A.dv = a.dv; A.scale = a.scale;
      std::vector<float> w((size_t)9 * Cv, 0.f), b(Cv, 0.f);
      for (int c = 0; c < Cv; ++c) {
        for (int t = 0; t < 9; ++t) w[(size_t)t * Cv + c] = dw.weight[(size_t)c * 9 + t];
        if (!dw.bias.empty()) b[c] = dw.bias[c];
      }
      A.pe_w.alloc(w.size() * 4); A.pe_b.alloc(b.size() * 4);
      LP_HIP(hipMemcpy(A.pe_w.p, w.data(), w.size() * 4, hipMemcpyHostToDevice));
      LP_HIP(hipMemcpy(A.pe_b.p, b.data(), b.size() * 4, hipMemcpyHostToDevice));
      const Tensor& TI = tensors_[tin];
      DetOp op;
      op.kind = DetOp::ATTN; op.layer = l.name; op.conv = (int)attns_.size() - 1; op.in = tin; op.out = tout;
      op.flops = 2.0 * a.heads * (double)a.hw * a.hw * (a.dk + a.dv);
      op.bytes = ((double)TI.C + Cv) * TI.H * TI.W * esd;
      ops_.push_back(op);
      continue;
    }
    if (l.type == "ConvolutionDepthWise") {
      const int tin = cinfo[i].tin, tout = cinfo[i].tout;
      const Tensor& TI = tensors_[tin];
      LP_CHECK(TI.segs.size() == 1 && tensors_[tout].segs.size() == 1 && TI.Cp == tensors_[tout].Cp, LP_ERR_GRAPH,
               "ConvolutionDepthWise %s: input and output must be plain tensors", l.name.c_str());
      ensure_buffer(tout);
      const int C = l.ipar(0), Cp = TI.Cp;
      dws_.emplace_back();
      DwLayer& D = dws_.back();
      D.act = fused_act[i];
      std::vector<float> w((size_t)9 * Cp, 0.f), b(Cp, 0.f);
      for (int c = 0; c < C; ++c) {
        for (int t = 0; t < 9; ++t) w[(size_t)t * Cp + TI.phys(c)] = l.weight[(size_t)c * 9 + t];
        if (!l.bias.empty()) b[TI.phys(c)] = l.bias[c];
      }
      D.w.alloc(w.size() * 4); D.b.alloc(b.size() * 4);
      LP_HIP(hipMemcpy(D.w.p, w.data(), w.size() * 4, hipMemcpyHostToDevice));
      LP_HIP(hipMemcpy(D.b.p, b.data(), b.size() * 4, hipMemcpyHostToDevice));
      DetOp op;
      op.kind = DetOp::DWCONV; op.layer = l.name; op.conv = (int)dws_.size() - 1; op.in = tin; op.out = tout;
      op.flops = 2.0 * 9.0 * C * TI.H * TI.W;
      op.bytes = 2.0 * C * TI.H * TI.W * esd;
      macs_ += 9.0 * C * TI.H * TI.W;
      ops_.push_back(op);
      continue;
    }
    if (l.type == "Convolution") {
      if (try_c2f(i)) continue;
      if (try_sppf(i)) continue;
      const int tin = cinfo[i].tin;
      int tout = cinfo[i].tout, res = -1;
      const int k = l.ipar(1, 1), s = l.ipar(3, 1);
      const int Cout = l.ipar(0), Cin = l.in_ch;
      // residual fusion: the activation output feeds exactly one BinaryOp add
      auto& cs = canon_consumers[tensors_[tout].name];
      if (cs.size() == 1 && L[cs[0]].type == "BinaryOp" && !is_tail(cs[0])) {
        const NcnnLayer& add = L[cs[0]];
        const int ta = get(add.inputs[0]), tb = get(add.inputs[1]);
        const int other = ta == tout ? tb : ta;
        if (other != tout && tensors_[other].Cp == tensors_[tout].Cp) {
          res = other;
          tout = get(add.outputs[0]);
          done[cs[0]] = 1;
        }
      }
      // sibling merge: another plain 3x3 conv reads the same input (Detect head: the box and class branches of a
      // level both start with a 3x3 conv on the neck output): one launch computes both, output channels side by
      // side in one buffer -- the input is read once and a launch disappears.  LITEPI_NO_SIBLING=1 disables it.
      static const bool no_sibling = getenv("LITEPI_NO_SIBLING") != nullptr;
      int sib = -1;
      if (!no_sibling && res < 0 && k == 3 && s == 1 && tin != input_tensor && impl_ == IMPL_MFMA && tensors_[tout].buf < 0 &&
          tensors_[tout].parent < 0 && tensors_[tout].segs.size() == 1) {
        auto plain_successor = [&](int t) {  // the output must stay a plain tensor: no fused add on it
          for (int c : canon_consumers[tensors_[t].name])
            if (L[c].type == "BinaryOp") return false;
          return true;
        };
        for (int j = i + 1; j < first_tail && sib < 0; ++j) {
          if (L[j].type != "Convolution" || done[j] || skip[j] || is_tail(j) || cinfo[j].tin != tin) continue;
          const int tj = cinfo[j].tout;
          if (L[j].ipar(1, 1) == 3 && L[j].ipar(3, 1) == 1 && fused_act[j] == fused_act[i] && tensors_[tj].buf < 0 &&
              tensors_[tj].parent < 0 && tensors_[tj].segs.size() == 1 && plain_successor(tj) && plain_successor(tout) &&
              L[j].bias.empty() == l.bias.empty())
            sib = j;
        }
      }
      if (sib >= 0) {
        const NcnnLayer& l2 = L[sib];
        const int t2 = cinfo[sib].tout;
        const int cpa = tensors_[tout].Cp, cpb = tensors_[t2].Cp, Ho = tensors_[tout].H, Wo = tensors_[tout].W;
        const int CpO = cpa + cpb;
        const int nb = alloc_buffer(CpO, Ho, Wo);
        tensors_[tout].buf = nb; tensors_[tout].off = 0; tensors_[tout].materialised = true;
        tensors_[t2].buf = nb; tensors_[t2].off = cpa; tensors_[t2].materialised = true;
        const Tensor& TI = tensors_[tin];
        std::vector<float> w((size_t)CpO * 9 * TI.Cp, 0.f), b(CpO, 0.f);
        auto put = [&](const NcnnLayer& lc, const Tensor& O, int base) {
          const int co_n = lc.ipar(0);
          for (int co = 0; co < co_n; ++co) {
            const int pc = base + O.phys(co);
            for (int ci = 0; ci < Cin; ++ci)
              for (int t = 0; t < 9; ++t) w[((size_t)pc * 9 + t) * TI.Cp + TI.phys(ci)] = lc.weight[((size_t)co * Cin + ci) * 9 + t];
            if (!lc.bias.empty()) b[pc] = lc.bias[co];
          }
        };
        put(l, tensors_[tout], 0);
        put(l2, tensors_[t2], cpa);
        convs_.emplace_back(new ConvLayer());
        convs_.back()->name = l.name + "|" + l2.name;
        convs_.back()->build(prec_, impl_, 3, 1, TI.Cp, CpO, fused_act[i], w, b, Ho, Wo, maxB_);
        const double macs = 9.0 * Cin * (Cout + l2.ipar(0)) * Ho * Wo;
        macs_ += macs;
        DetOp op;
        op.kind = DetOp::CONV; op.layer = convs_.back()->name; op.conv = (int)convs_.size() - 1;
        op.flops = 2.0 * macs;
        op.bytes = ((double)TI.C * TI.H * TI.W + (double)(Cout + l2.ipar(0)) * Ho * Wo) * esd + (double)(l.weight.size() + l2.weight.size()) * esd;
        // a tensor that stands for the merged buffer (the conv's output view); pushed last: it invalidates references
        Tensor M = tensors_[tout];
        M.name = tensors_[tout].name + "|" + tensors_[t2].name; M.C = tensors_[tout].C + tensors_[t2].C; M.Cp = CpO;
        M.segs = {cpa, cpb}; M.buf = nb; M.off = 0;
        tensors_.push_back(M);
        op.in = tin; op.out = (int)tensors_.size() - 1;
        ops_.push_back(op);
        done[sib] = 1;
        continue;
      }
      // bottleneck fusion: this 3x3 conv feeds exactly one 3x3 conv whose activation is added to THIS conv's
      // input (C2f.m[i] with shortcut): both convs, the SiLUs and the add become one launch, the intermediate
      // stays in LDS (BottleneckPair).  LITEPI_NO_BNECK=1 keeps the layer-at-a-time plan (A/B measurements).
      static const bool no_bneck = getenv("LITEPI_NO_BNECK") != nullptr;
      if (!no_bneck && res < 0 && k == 3 && s == 1 && Cin == Cout && tin != input_tensor && impl_ == IMPL_MFMA &&
          fused_act[i] == ACT_SILU && tensors_[tout].segs.size() == 1 && tensors_[tin].segs.size() == 1) {
        auto& csb = canon_consumers[tensors_[tout].name];
        if (csb.size() == 1 && L[csb[0]].type == "Convolution" && !is_tail(csb[0]) && !done[csb[0]]) {
          const int j = csb[0];
          const NcnnLayer& lb = L[j];
          const int tb = cinfo[j].tout;
          auto& csa = canon_consumers[tensors_[tb].name];
          if (lb.ipar(1, 1) == 3 && lb.ipar(3, 1) == 1 && lb.ipar(0) == Cout && lb.in_ch == Cout && fused_act[j] == ACT_SILU &&
              csa.size() == 1 && L[csa[0]].type == "BinaryOp" && !is_tail(csa[0])) {
            const NcnnLayer& add = L[csa[0]];
            const int ta = get(add.inputs[0]), tb2 = get(add.inputs[1]);
            const int other = ta == tb ? tb2 : ta;
            const int tfinal = get(add.outputs[0]);
            const Tensor& TI = tensors_[tin];
            if (other == tin && TI.Cp == tensors_[tfinal].Cp && TI.Cp == tensors_[tout].Cp &&
                BottleneckPair::supported(prec_, impl_, TI.Cp, TI.H, TI.W, maxB_)) {
              // cv2 fusion: y_last is the last segment of a zero-copy Concat whose only consumer is a 1x1 conv
              // (C2f.cv2): that conv runs in the same launch, y_last stays in registers (LITEPI_NO_CV2FUSE=1: off)
              static const bool no_cv2 = getenv("LITEPI_NO_CV2FUSE") != nullptr;
              int jc = -1, tcat = -1;
              BottleneckPair::Cv2 cv2;
              std::vector<float> w3, b3;
              {
                auto& cf = canon_consumers[tensors_[tfinal].name];
                if (!no_cv2 && cf.size() == 1 && L[cf[0]].type == "Concat" && canon(L[cf[0]].inputs.back()) == tensors_[tfinal].name) {
                  const int cc = cf[0];
                  bool copied = false;
                  for (auto& cj : copies) copied = copied || cj.layer == cc;
                  const int tO = get(L[cc].outputs[0]);
                  auto& c2 = canon_consumers[tensors_[tO].name];
                  if (!copied && c2.size() == 1 && L[c2[0]].type == "Convolution" && !is_tail(c2[0]) && !done[c2[0]] &&
                      L[c2[0]].ipar(1, 1) == 1 && L[c2[0]].ipar(3, 1) == 1 && !fuse_up.count(c2[0]) &&
                      tensors_[tfinal].buf == tensors_[tO].buf && tensors_[tO].parent < 0) {
                    const NcnnLayer& l3 = L[c2[0]];
                    const int t3 = cinfo[c2[0]].tout;
                    bool feeds_add = false;
                    for (int c : canon_consumers[tensors_[t3].name]) feeds_add = feeds_add || L[c].type == "BinaryOp";
                    const Tensor& TO = tensors_[tO];
                    const int glob = tensors_[tfinal].off - TO.off;
                    if (!feeds_add && glob > 0 && glob + TI.Cp == TO.Cp && tensors_[t3].segs.size() == 1) {
                      cv2.cat_global = glob; cv2.c3 = tensors_[t3].Cp; cv2.act = fused_act[c2[0]];
                      // (C2f: the bottleneck's input y_n sits right in front of y_last in the concat buffer)
                      {
                        int bi = TI.buf, oi = TI.off;   // storage of the input: a Slice output is a view of its parent's segment (view())
                        if (TI.parent >= 0) {
                          const Tensor& P = tensors_[TI.parent];
                          oi = P.off; bi = P.buf;
                          for (int k2 = 0; k2 < TI.parent_seg; ++k2) oi += round_up(P.segs[k2], 8);
                        }
                        cv2.in_is_last_stored = bi >= 0 && bi == TO.buf && oi == TO.off + glob - TI.Cp;
                      }
                      if (BottleneckPair::supported(prec_, impl_, TI.Cp, TI.H, TI.W, maxB_, &cv2)) {
                        jc = c2[0]; tcat = tO;
                        const int cin3 = l3.in_ch, cout3 = l3.ipar(0);
                        w3.assign((size_t)cv2.c3 * TO.Cp, 0.f);
                        b3.assign(cv2.c3, 0.f);
                        for (int co = 0; co < cout3; ++co) {
                          const int pc = tensors_[t3].phys(co);
                          for (int ci = 0; ci < cin3; ++ci) w3[(size_t)pc * TO.Cp + TO.phys(ci)] = l3.weight[(size_t)co * cin3 + ci];
                          if (!l3.bias.empty()) b3[pc] = l3.bias[co];
                        }
                        cv2.w = &w3; cv2.bias = &b3;
                      }
                    }
                  }
                }
              }
              const int tdst = jc >= 0 ? cinfo[jc].tout : tfinal;
              ensure_buffer(tdst);
              const Tensor& TM = tensors_[tout];
              const Tensor& TF = tensors_[tfinal];
              auto pack = [&](const NcnnLayer& lc, const Tensor& A, const Tensor& O, std::vector<float>& w, std::vector<float>& b) {
                w.assign((size_t)O.Cp * 9 * A.Cp, 0.f);
                b.assign(O.Cp, 0.f);
                for (int co = 0; co < Cout; ++co) {
                  const int pc = O.phys(co);
                  for (int ci = 0; ci < Cin; ++ci)
                    for (int t = 0; t < 9; ++t) w[((size_t)pc * 9 + t) * A.Cp + A.phys(ci)] = lc.weight[((size_t)co * Cin + ci) * 9 + t];
                  if (!lc.bias.empty()) b[pc] = lc.bias[co];
                }
              };
              std::vector<float> wa, ba, wb, bb;
              pack(l, tensors_[tin], TM, wa, ba);
              pack(lb, TM, TF, wb, bb);
              bnecks_.emplace_back(new BottleneckPair());
              bnecks_.back()->name = l.name + "+" + lb.name + (jc >= 0 ? "+" + L[jc].name : std::string());
              bnecks_.back()->build(prec_, tensors_[tin].Cp, wa, ba, wb, bb, tensors_[tin].H, tensors_[tin].W, maxB_, jc >= 0 ? &cv2 : nullptr);
              double macs = 2.0 * 9.0 * Cin * Cout * tensors_[tin].H * tensors_[tin].W;
              if (jc >= 0) macs += (double)L[jc].in_ch * L[jc].ipar(0) * tensors_[tin].H * tensors_[tin].W;
              macs_ += macs;
              DetOp op;
              op.kind = DetOp::BNECK; op.layer = bnecks_.back()->name; op.conv = (int)bnecks_.size() - 1;
              op.flops = 2.0 * macs;
              {
                const Tensor& T0 = tensors_[tin];
                op.bytes = 2.0 * T0.C * T0.H * T0.W * esd + (double)(l.weight.size() + lb.weight.size()) * esd;
                if (jc >= 0)
                  op.bytes = ((double)(tensors_[tcat].C - T0.C) + T0.C + tensors_[tdst].C) * T0.H * T0.W * esd +
                             (double)(l.weight.size() + lb.weight.size() + L[jc].weight.size()) * esd;
              }
              op.in = tin; op.out = tdst; op.in2 = tcat;
              ops_.push_back(op);
              if (jc >= 0) done[jc] = 1;
              done[j] = 1; done[csa[0]] = 1;
              continue;
            }
          }
        }
      }
      // 1x1 tail fusion: this 3x3 conv's activation output feeds exactly one 1x1 conv (Detect-head
      // projections, C2f cv1 after a stride-2 conv): the second GEMM runs on the accumulator tile
      int tail = -1, tmid = -1;
      if (res < 0 && k == 3 && tin != input_tensor && impl_ == IMPL_MFMA) {
        auto& cs2 = canon_consumers[tensors_[tout].name];
        if (cs2.size() == 1 && L[cs2[0]].type == "Convolution" && !is_tail(cs2[0]) && !done[cs2[0]]) {
          const NcnnLayer& lb = L[cs2[0]];
          const int tb = cinfo[cs2[0]].tout;
          const bool plain1x1 = lb.ipar(1, 1) == 1 && lb.ipar(3, 1) == 1 && tensors_[tout].segs.size() == 1;
          // B must not itself be the producer of a fused residual add
          bool b_feeds_add = false;
          for (int c : canon_consumers[tensors_[tb].name]) b_feeds_add = b_feeds_add || L[c].type == "BinaryOp";
          // (a C2f.cv1 that the whole-C2f launch computes itself is not folded into this conv)
          // (round 4: a stride-2 conv whose 1x1 consumer is the cv1 of a C2f module that can run WITHOUT its cv1 -- C2fShape::MODE -1 --
          //  keeps that cv1 as its tail on the LDS-staged kernel even though a whole-module launch exists: v2's 80x80 backbone module)
          bool s2tail = false;
          if (c2f_on && plain1x1 && s == 2 && !getenv("LITEPI_NO_C2F_XCV1") && tensors_[tout].Cp == 48 &&
              S2ConvLayer::tail_supported(Cin, tensors_[tout].Cp, tensors_[tb].Cp, tensors_[tb].H, tensors_[tb].W)) {
            C2fMatch mx;
            s2tail = match_c2f(cs2[0], mx) && C2fLayer::supported(c2f_shape(mx, -1, 0), tensors_[tb].H, tensors_[tb].W);
          }
          if (plain1x1 && !b_feeds_add && (s2tail || (!c2f_plain_ok(cs2[0]) && ConvLayer::tail_supported(k, s, tensors_[tout].Cp, tensors_[tb].Cp)))) {
            tail = cs2[0];
            tmid = tout;
            tout = tb;
            done[tail] = 1;
          }
        }
      }
      ensure_buffer(tout);
      // a stride-2 conv WITH its folded 1x1 tail on the LDS-staged kernel (v1's conv_6 + conv_7: s2lds_kernel<S2L16x32t>)
      if (c2f_on && !getenv("LITEPI_NO_S2C") && tail >= 0 && res < 0 && k == 3 && s == 2 && fused_act[i] == ACT_SILU && fused_act[tail] == ACT_SILU &&
          !l.bias.empty() && !L[tail].bias.empty() && tin != input_tensor && tensors_[tin].Cp == Cin && tensors_[tmid].Cp == Cout &&
          tensors_[tout].Cp == L[tail].ipar(0) && tensors_[tout].parent < 0 &&
          S2ConvLayer::tail_supported(Cin, Cout, L[tail].ipar(0), tensors_[tout].H, tensors_[tout].W)) {
        s2cs_.emplace_back(new S2ConvLayer());
        s2cs_.back()->name = l.name + "+" + L[tail].name;
        const std::vector<float> w2 = conv_w(tail);
        s2cs_.back()->build(Cin, Cout, tensors_[tout].H, tensors_[tout].W, conv_w(i), l.bias, &w2, &L[tail].bias);
        const Tensor& TI2 = tensors_[tin];
        const Tensor& TO2 = tensors_[tout];
        const double macs2 = (9.0 * Cin * Cout + (double)Cout * L[tail].ipar(0)) * TO2.H * TO2.W;
        macs_ += macs2;
        DetOp op;
        op.kind = DetOp::S2C; op.conv = (int)s2cs_.size() - 1; op.layer = s2cs_.back()->name; op.in = tin; op.out = tout;
        op.flops = 2.0 * macs2;
        op.bytes = ((double)TI2.C * TI2.H * TI2.W + (double)TO2.C * TO2.H * TO2.W) * esd + (double)(l.weight.size() + L[tail].weight.size()) * esd;
        ops_.push_back(op);
        // ---- A/B (LITEPI_C2F_XCV1=1): the C2f module this cv1 belongs to, WITHOUT its cv1 (C2fShape::MODE -1: y0 | y1 come from the
        //      concat buffer the launch above fills; for n = 2 modules cv1 on the halo-4 region was twice its work): both bottlenecks +
        //      cv2 in one launch.  v1's 80x80 backbone module: 56.0 us against 25.0 + 33.1 for the two bottleneck launches, 19 launches,
        //      +0.1 to +0.5 % end to end: inside the noise, so the two-launch plan stays the default.
        {
          C2fMatch m;
          done[tail] = 0;   // (match_c2f wants its cv1 unclaimed)
          // (on for v2's module, whose alternative is the whole-module launch with cv1 recomputed on the halo-4 region; opt-in for v1's)
          const bool ok = (getenv("LITEPI_C2F_XCV1") || (Cout == 48 && !getenv("LITEPI_NO_C2F_XCV1"))) && match_c2f(tail, m);
          done[tail] = 1;
          if (ok) {
            C2fShape sh = c2f_shape(m, -1, 0);
            const int t2 = cinfo[m.cv2].tout;
            const int Hh = tensors_[t2].H, Ww = tensors_[t2].W;
            if (C2fLayer::supported(sh, Hh, Ww)) {
              std::vector<float> w_cv2 = conv_w(m.cv2), w_a[2], w_b[2];
              C2fLayer::Src src;
              src.cv2 = &w_cv2; src.cv2_b = &L[m.cv2].bias;
              for (int q = 0; q < m.nb; ++q) {
                w_a[q] = conv_w(m.a[q]); w_b[q] = conv_w(m.b[q]);
                src.a[q] = &w_a[q]; src.a_b[q] = &L[m.a[q]].bias;
                src.bb[q] = &w_b[q]; src.bb_b[q] = &L[m.b[q]].bias;
              }
              c2fs_.emplace_back(new C2fLayer());
              C2fLayer& cl = *c2fs_.back();
              cl.name = L[m.a[0]].name + ".." + L[m.cv2].name;
              cl.build(sh, Hh, Ww, src);
              C2fIO io;
              io.src1 = m.t_cat;   // (unused by MODE -1: the module reads the concat buffer)
              io.cat = m.t_cat;
              io.out = t2;
              ensure_buffer(t2);
              for (size_t q = 2; q < m.ys.size(); ++q) tensors_[m.ys[q]].materialised = getenv("LITEPI_C2F_STORE_ALL") != nullptr || !cl.cv2_from_lds();
              if (!getenv("LITEPI_C2F_STORE_ALL")) {
                for (size_t q = 2; q < m.ys.size(); ++q) tensors_[m.ys[q]].in_c2f = true;
                for (int q = 0; q < m.nb; ++q) {
                  tensors_[cinfo[m.a[q]].tout].in_c2f = true; tensors_[cinfo[m.b[q]].tout].in_c2f = true;
                  for (auto& o : L[m.add[q]].outputs) tensors_[get(o)].in_c2f = true;
                }
              }
              c2f_io_.push_back(io);
              macs_ += cl.macs_per_image;
              DetOp op2;
              op2.kind = DetOp::C2F; op2.conv = (int)c2fs_.size() - 1; op2.layer = cl.name;
              op2.in = m.t_cat; op2.out = t2;
              op2.flops = 2.0 * cl.macs_per_image;
              op2.bytes = ((double)tensors_[m.t_cat].C * 0.5 + (double)tensors_[t2].C) * Hh * Ww * esd;
              ops_.push_back(op2);
              done[m.cv2] = 1;
              for (int q = 0; q < m.nb; ++q) done[m.a[q]] = done[m.b[q]] = done[m.add[q]] = 1;
            }
          }
        }
        continue;
      }
      // a stride-2 conv without a folded tail whose shape the c2f machinery covers: s2conv_kernel (LITEPI_NO_S2C=1: off)
      if (c2f_on && !getenv("LITEPI_NO_S2C") && tail < 0 && res < 0 && k == 3 && s == 2 && fused_act[i] == ACT_SILU && !l.bias.empty() && tin != input_tensor &&
          tensors_[tin].Cp == Cin && tensors_[tout].Cp == Cout && tensors_[tout].segs.size() == 1 &&
          S2ConvLayer::supported(Cin, Cout, tensors_[tout].H, tensors_[tout].W)) {
        s2cs_.emplace_back(new S2ConvLayer());
        s2cs_.back()->name = l.name;
        s2cs_.back()->build(Cin, Cout, tensors_[tout].H, tensors_[tout].W, conv_w(i), l.bias);
        const Tensor& TI2 = tensors_[tin];
        const Tensor& TO2 = tensors_[tout];
        const double macs2 = 9.0 * Cin * Cout * TO2.H * TO2.W;
        macs_ += macs2;
        DetOp op;
        op.kind = DetOp::S2C; op.conv = (int)s2cs_.size() - 1; op.layer = l.name; op.in = tin; op.out = tout;
        op.flops = 2.0 * macs2;
        op.bytes = ((double)TI2.C * TI2.H * TI2.W + (double)TO2.C * TO2.H * TO2.W) * esd + (double)l.weight.size() * esd;
        ops_.push_back(op);
        continue;
      }
      const Tensor& TI = tensors_[tin];
      const Tensor& TO = tensors_[tail >= 0 ? tmid : tout];
      double macs = (double)k * k * Cin * Cout * TO.H * TO.W;
      if (tail >= 0) macs += (double)L[tail].in_ch * L[tail].ipar(0) * TO.H * TO.W;
      macs_ += macs;
      DetOp op;
      op.layer = l.name;
      op.flops = 2.0 * macs;
      op.bytes = ((double)TI.C * TI.H * TI.W + (double)TO.C * TO.H * TO.W * (res >= 0 ? 2 : 1)) * esd + (double)l.weight.size() * esd;
      op.in = tin; op.out = tout; op.res = res;
      if (tin == input_tensor) {
        LP_CHECK(Cin == 3 && k >= 1 && k <= 7, LP_ERR_GRAPH, "first convolution must read the 3-channel image with k <= 7");
        // weights in BGR order, [k*k*3][CO]
        const int CO = TO.Cp;
        std::vector<float> w((size_t)k * k * 3 * CO, 0.f), b(CO, 0.f);
        for (int co = 0; co < Cout; ++co) {
          const int pc = TO.phys(co);
          for (int c = 0; c < 3; ++c)
            for (int ky = 0; ky < k; ++ky)
              for (int kx = 0; kx < k; ++kx)
                w[(size_t)((ky * k + kx) * 3 + (2 - c)) * CO + pc] = l.weight[(((size_t)co * 3 + c) * k + ky) * k + kx];
          if (!l.bias.empty()) b[pc] = l.bias[co];
        }
        stem_.build(prec_, CO, fused_act[i], w, b, k, s, l.ipar(4, 0));
        op.kind = DetOp::STEM;
        op.bytes = (double)3 * TI.H * TI.W + (double)TO.C * TO.H * TO.W * esd;
      } else {
        const int taps = k * k;
        std::vector<float> w((size_t)TO.Cp * taps * TI.Cp, 0.f), b(TO.Cp, 0.f);
        for (int co = 0; co < Cout; ++co) {
          const int pc = TO.phys(co);
          for (int ci = 0; ci < Cin; ++ci) {
            const int pi = TI.phys(ci);
            for (int t = 0; t < taps; ++t)
              w[((size_t)pc * taps + t) * TI.Cp + pi] = l.weight[((size_t)co * Cin + ci) * taps + t];
          }
          if (!l.bias.empty()) b[pc] = l.bias[co];
        }
        convs_.emplace_back(new ConvLayer());
        convs_.back()->name = l.name;
        convs_.back()->build(prec_, impl_, k, s, TI.Cp, TO.Cp, fused_act[i], w, b, TO.H, TO.W, maxB_, tail >= 0);
        if (tail >= 0) {
          const NcnnLayer& lb = L[tail];
          const Tensor& TB = tensors_[tout];
          const int cin2 = lb.in_ch, cout2 = lb.ipar(0);
          std::vector<float> w2((size_t)TB.Cp * TO.Cp, 0.f), b2(TB.Cp, 0.f);
          for (int co = 0; co < cout2; ++co) {
            const int pc = TB.phys(co);
            for (int ci = 0; ci < cin2; ++ci) w2[(size_t)pc * TO.Cp + TO.phys(ci)] = lb.weight[(size_t)co * cin2 + ci];
            if (!lb.bias.empty()) b2[pc] = lb.bias[co];
          }
          convs_.back()->attach_tail(TB.Cp, fused_act[tail], w2, b2);
          convs_.back()->name = l.name + "+" + lb.name;
          op.layer = convs_.back()->name;
          op.bytes = ((double)TI.C * TI.H * TI.W + (double)TB.C * TB.H * TB.W) * esd + (double)(l.weight.size() + lb.weight.size()) * esd;
        }
        op.kind = DetOp::CONV;
        op.conv = (int)convs_.size() - 1;
        if (fuse_up.count(i)) {
          op.in2 = fuse_up[i];
          op.layer = l.name + "(up)";
          op.bytes -= 0.75 * tensors_[op.in2].C * 4.0 * tensors_[op.in2].H * tensors_[op.in2].W * esd;  // u is read once, not its x4 copy
        }
      }
      ops_.push_back(op);
    } else if (l.type == "BinaryOp") {
      DetOp op;
      op.kind = DetOp::ADD; op.layer = l.name;
      op.in = get(l.inputs[0]); op.in2 = get(l.inputs[1]); op.out = get(l.outputs[0]);
      ensure_buffer(op.out);
      const Tensor& T = tensors_[op.out];
      op.bytes = 3.0 * T.C * T.H * T.W * esd;
      ops_.push_back(op);
    } else if (l.type == "Pooling") {
      // SPPF: this pool and the two that consume it in a chain
      int chain[3] = {i, -1, -1};
      for (int q = 1; q < 3; ++q) {
        auto& cs = canon_consumers[canon(L[chain[q - 1]].outputs[0])];
        for (int c : cs)
          if (L[c].type == "Pooling") chain[q] = c;
        LP_CHECK(chain[q] >= 0, LP_ERR_GRAPH, "Pooling %s is not part of an SPPF chain of three", l.name.c_str());
      }
      DetOp op;
      op.kind = DetOp::SPPF; op.layer = l.name;
      op.in = get(l.inputs[0]);
      op.out = get(L[chain[0]].outputs[0]); op.out2 = get(L[chain[1]].outputs[0]); op.out3 = get(L[chain[2]].outputs[0]);
      ensure_buffer(op.out); ensure_buffer(op.out2); ensure_buffer(op.out3);
      done[chain[1]] = done[chain[2]] = 1;
      const Tensor& T = tensors_[op.in];
      op.bytes = 4.0 * T.C * T.H * T.W * esd;
      ops_.push_back(op);
    } else if (l.type == "Interp") {
      // upsample fusion: Interp x2 -> first input of a Concat -> exactly one 1x1 conv (FPN top-down: C2f.cv1).  The conv
      // gathers those channels from the half-resolution tensor itself; the upsampled copy is never materialised.
      static const bool no_upfuse = getenv("LITEPI_NO_UPFUSE") != nullptr;
      if (!no_upfuse && impl_ == IMPL_MFMA) {
        auto& c1 = canon_consumers[canon(l.outputs[0])];
        if (c1.size() == 1 && L[c1[0]].type == "Concat" && canon(L[c1[0]].inputs[0]) == canon(l.outputs[0])) {
          auto& c2 = canon_consumers[canon(L[c1[0]].outputs[0])];
          const int tsrc = get(l.inputs[0]);
          if (c2.size() == 1 && L[c2[0]].type == "Convolution" && L[c2[0]].ipar(1, 1) == 1 && L[c2[0]].ipar(3, 1) == 1 &&
              !is_tail(c2[0]) && tensors_[get(l.outputs[0])].off == 0 && tensors_[tsrc].Cp % 8 == 0 &&
              tensors_[get(l.outputs[0])].buf >= 0 && tensors_[get(l.outputs[0])].buf == tensors_[get(L[c1[0]].outputs[0])].buf &&  // zero-copy segment 0
              tensors_[tsrc].Cp == tensors_[get(l.outputs[0])].Cp) {
            fuse_up[c2[0]] = tsrc;
            continue;
          }
        }
      }
      DetOp op;
      op.kind = DetOp::UPSAMPLE; op.layer = l.name;
      op.in = get(l.inputs[0]); op.out = get(l.outputs[0]);
      ensure_buffer(op.out);
      const Tensor& T = tensors_[op.out];
      op.bytes = 1.25 * T.C * T.H * T.W * esd;
      ops_.push_back(op);
    } else if (l.type == "Concat") {
      for (auto& cj : copies) {
        if (cj.layer != i) continue;
        // destination view = slice of the concat buffer
        Tensor d = tensors_[cj.src];
        d.name += "@cat"; d.parent = -1; d.buf = cj.dst_buf; d.off = cj.dst_off;
        tensors_.push_back(d);
        DetOp op;
        op.kind = DetOp::COPY; op.layer = l.name; op.in = cj.src; op.out = (int)tensors_.size() - 1;
        op.bytes = 2.0 * d.C * d.H * d.W * esd;
        ops_.push_back(op);
      }
    }
  }

  // ---- stem block: stem + the stride-2 conv that is its only consumer (+ that conv's fused 1x1 tail) in one launch;
  //      the 320x320 stem map is never stored (StemLayer::launch_block; LITEPI_NO_STEMBLOCK=1: off)
  if (!getenv("LITEPI_NO_STEMBLOCK") && ops_.size() >= 2 && ops_[0].kind == DetOp::STEM && ops_[1].kind == DetOp::CONV &&
      ops_[1].in == ops_[0].out && ops_[1].res < 0 && S_ % 4 == 0 &&
      canon_consumers[tensors_[ops_[0].out].name].size() == 1 && stem_.block_supported(*convs_[ops_[1].conv])) {
    const Tensor& TI = tensors_[ops_[0].in >= 0 ? ops_[0].in : input_tensor];
    const Tensor& TO = tensors_[ops_[1].out];
    ops_[0].kind = DetOp::STEMBLOCK;
    ops_[0].conv = ops_[1].conv;
    ops_[0].out = ops_[1].out;
    ops_[0].layer += "+" + ops_[1].layer;
    ops_[0].flops += ops_[1].flops;
    ops_[0].bytes = 3.0 * TI.H * TI.W + (double)TO.C * TO.H * TO.W * esd;
    ops_.erase(ops_.begin() + 1);
  }

  // ---- Detect tail ---------------------------------------------------------------------------
  reg_max_ = 0;
  std::vector<float> dfl;
  const NcnnLayer* anchors = nullptr;
  const NcnnLayer* strides = nullptr;
  for (int i = 0; i < n; ++i) {
    if (L[i].type == "Convolution" && i >= first_tail) {
      LP_CHECK(L[i].bias.empty() && L[i].ipar(0) == 1, LP_ERR_GRAPH, "unexpected convolution %s in the Detect tail", L[i].name.c_str());
      dfl = L[i].weight;
      reg_max_ = (int)dfl.size();
    }
    if (L[i].type == "MemoryData") {
      if (L[i].ipar(1, 0) == 2 && !anchors) anchors = &L[i];
      if (L[i].ipar(1, 0) == 0 && L[i].ipar(2, 0) == 0 && !strides) strides = &L[i];
    }
  }
  LP_CHECK(reg_max_ > 0 && reg_max_ <= 32, LP_ERR_GRAPH, "no DFL convolution found in the Detect tail");
  LP_CHECK(anchors && strides, LP_ERR_GRAPH, "anchor / stride constants missing from the Detect tail");
  A_ = 0;
  nc_ = -1;
  for (int hc : head_cats) {
    Level lv;
    lv.box = get(L[hc].inputs[0]);
    lv.cls = get(L[hc].inputs[1]);
    const Tensor& B = tensors_[lv.box];
    const Tensor& C = tensors_[lv.cls];
    LP_CHECK(B.C == 4 * reg_max_ && B.segs.size() == 1 && C.segs.size() == 1 && B.H == C.H && B.W == C.W, LP_ERR_GRAPH,
             "Detect head %s: box branch must have 4*reg_max channels", L[hc].name.c_str());
    LP_CHECK(nc_ < 0 || nc_ == C.C, LP_ERR_GRAPH, "Detect head: class count differs between levels");
    nc_ = C.C;
    lv.H = B.H; lv.W = B.W; lv.off = A_;
    A_ += B.H * B.W;
    levels_.push_back(lv);
  }
  // the NMS kernel packs the anchor index into 14 bits of its sort key and keeps every candidate of an image in one
  // workgroup's LDS: reject larger heads here, at load time, not on every call (a 1024x1024 input has 21504 anchors)
  LP_CHECK(A_ <= 16384, LP_ERR_GRAPH, "Detect head with %d anchors: at most 16384 are supported (input size %d is too large)", A_, S_);
  LP_CHECK((int)anchors->data.size() == 2 * A_ && (int)strides->data.size() == A_, LP_ERR_GRAPH,
           "anchor tables (%zu, %zu) do not match %d anchors", anchors->data.size(), strides->data.size(), A_);
  for (auto& lv : levels_) {
    const float st = strides->data[lv.off];
    LP_CHECK(st * lv.H == (float)S_, LP_ERR_GRAPH, "stride table does not match level %dx%d", lv.H, lv.W);
  }
  d_anchors_.alloc(anchors->data.size() * 4);
  LP_HIP(hipMemcpy(d_anchors_.p, anchors->data.data(), anchors->data.size() * 4, hipMemcpyHostToDevice));
  d_strides_.alloc(strides->data.size() * 4);
  LP_HIP(hipMemcpy(d_strides_.p, strides->data.data(), strides->data.size() * 4, hipMemcpyHostToDevice));
  d_dfl_.alloc(dfl.size() * 4);
  LP_HIP(hipMemcpy(d_dfl_.p, dfl.data(), dfl.size() * 4, hipMemcpyHostToDevice));

  // ---- Detect-head fusion (fp16 MFMA plan; LITEPI_NO_HEADFUSE=1: off).  Per level the planner has emitted three ops:
  //      A = the two first 3x3 convs merged (sibling merge: box tower 64 | class tower c3 channels in one buffer),
  //      B = box tower's second 3x3 + its 1x1 projection (fused tail), C = the same for the class tower.  When every level
  //      has exactly this shape, each (A, B, C) triple becomes one HEAD op (head_fused_kernel) that also decodes and filters,
  //      and the stand-alone decode launch disappears.
  if (!getenv("LITEPI_NO_HEADFUSE") && prec_ == LP_FP16 && impl_ == IMPL_MFMA && reg_max_ == 16) {
    struct Trip { int a, b, c, c3, proj; };   // proj: the class tower's 1x1 projection when it is a launch of its own (else -1)
    std::vector<Trip> trips;
    auto producer_of = [&](int tensor) {
      for (size_t q = 0; q < ops_.size(); ++q)
        if (ops_[q].kind == DetOp::CONV && ops_[q].out == tensor) return (int)q;
      return -1;
    };
    bool all = true;
    for (auto& lv : levels_) {
      const int ob = producer_of(lv.box);
      int oc = producer_of(lv.cls), oproj = -1;
      if (ob < 0 || oc < 0) { all = false; break; }
      // class tower: second 3x3 with the projection as its fused tail, or (48-channel towers: no tail kernel for three
      // channel tiles) the 3x3 and the 1x1 as two launches
      if (convs_[ops_[oc].conv]->k == 1) {
        const ConvLayer& cp = *convs_[ops_[oc].conv];
        oproj = oc;
        oc = producer_of(ops_[oproj].in);
        if (oc < 0 || cp.T2 != 0 || cp.act != ACT_NONE || ops_[oproj].res >= 0 || ops_[oproj].in2 >= 0 || cp.Cin != convs_[ops_[oc].conv]->Cout ||
            convs_[ops_[oc].conv]->T2 != 0 || cp.b_host.empty()) { all = false; break; }
      }
      const ConvLayer& cb = *convs_[ops_[ob].conv];
      const ConvLayer& cc = *convs_[ops_[oc].conv];
      const Tensor& tb = tensors_[ops_[ob].in];
      const Tensor& tc = tensors_[ops_[oc].in];
      bool ok = cb.k == 3 && cb.stride == 1 && cb.T2 > 0 && cb.act == ACT_SILU && cb.act2 == ACT_NONE && cb.Cin == 64 && cb.Cout == 64 &&
                cb.Cout2 == 64 && cc.k == 3 && cc.stride == 1 && (oproj >= 0 || (cc.T2 > 0 && cc.act2 == ACT_NONE)) && cc.act == ACT_SILU && cc.Cin == cc.Cout &&
                ops_[ob].res < 0 && ops_[oc].res < 0 && tb.buf >= 0 && tb.buf == tc.buf && tb.parent < 0 && tc.parent < 0 && tb.off == 0 &&
                tc.off == tb.Cp && tb.Cp == 64 && tc.Cp == cc.Cin && buffers_[tb.buf].Cp == 64 + cc.Cin;
      int oa = -1;
      for (size_t q = 0; ok && q < ops_.size(); ++q)
        if (ops_[q].kind == DetOp::CONV && ops_[q].out >= 0 && tensors_[ops_[q].out].buf == tb.buf && (int)q != ob && (int)q != oc && (int)q != oproj) oa = (int)q;
      ok = ok && oa >= 0 && oa < ob && oa < oc;
      if (ok) {
        const ConvLayer& ca = *convs_[ops_[oa].conv];
        const Tensor& ti = tensors_[ops_[oa].in];
        ok = ca.k == 3 && ca.stride == 1 && ca.T2 == 0 && ca.act == ACT_SILU && ca.Cout == 64 + cc.Cin && ops_[oa].res < 0 && ops_[oa].in2 < 0 &&
             ca.Cin == ti.Cp && !ca.b_host.empty() && tensors_[lv.cls].C == nc_ &&
             HeadLayer::supported(ca.Cin, 64, cc.Cin, nc_, reg_max_, lv.H, lv.W);
      }
      if (!ok) { all = false; break; }
      trips.push_back({oa, ob, oc, cc.Cin, oproj});
    }
    // Two class row tiles (48-channel class towers, v2).  Rounds 2-3: correct but slower end to end than the three-launch plan
    // (the kernel owned a CU's whole LDS), so opt-in.  Round 4: stage A on 16-pixel tiles, P3 and P4 at two workgroups per CU and
    // a one-round P5 shape make it the faster plan (same-box A/B 53.4 k -> 54.2 k images/s, 53 -> 43 launches): default.
    if (all && !trips.empty() && trips[0].c3 > 32) {   // (round 4: default; LITEPI_HEADFUSE=narrow restores the three-launch plan for A/B)
      const char* hf = getenv("LITEPI_HEADFUSE");
      if (hf && strcmp(hf, "narrow") == 0) all = false;
    }
    if (all && trips.size() == levels_.size()) {
      std::vector<char> dead(ops_.size(), 0);
      for (size_t q = 0; q < trips.size(); ++q) {
        const Trip& t = trips[q];
        const ConvLayer& ca = *convs_[ops_[t.a].conv];
        const ConvLayer& cb = *convs_[ops_[t.b].conv];
        const ConvLayer& cc = *convs_[ops_[t.c].conv];
        HeadLayer::Src src;
        src.wa = &ca.w_host; src.ba = &ca.b_host;
        src.wbb = &cb.w_host; src.bbb = &cb.b_host; src.wpb = &cb.w2_host; src.bpb = &cb.b2_host;
        src.wbc = &cc.w_host; src.bbc = &cc.b_host; src.wpc = &cc.w2_host; src.bpc = &cc.b2_host;
        src.ncp = cc.Cout2;
        if (t.proj >= 0) {
          const ConvLayer& cp = *convs_[ops_[t.proj].conv];
          src.wpc = &cp.w_host; src.bpc = &cp.b_host; src.ncp = cp.Cout;
        }
        heads_.emplace_back(new HeadLayer());
        heads_.back()->name = ops_[t.a].layer + "+" + ops_[t.b].layer + "+" + ops_[t.c].layer + "+decode";
        heads_.back()->build(ca.Cin, t.c3, nc_, levels_[q].H, levels_[q].W, maxB_, src);
        DetOp& op = ops_[t.a];
        op.kind = DetOp::HEAD; op.conv = (int)heads_.size() - 1; op.in2 = (int)q; op.out = -1;
        op.layer = heads_.back()->name;
        op.flops = ops_[t.a].flops + ops_[t.b].flops + ops_[t.c].flops + (t.proj >= 0 ? ops_[t.proj].flops : 0.0);
        op.bytes = (double)tensors_[op.in].C * levels_[q].H * levels_[q].W * esd;
        dead[t.b] = dead[t.c] = 1;
        if (t.proj >= 0) dead[t.proj] = 1;
      }
      std::vector<DetOp> kept;
      for (size_t q = 0; q < ops_.size(); ++q)
        if (!dead[q]) kept.push_back(ops_[q]);
      ops_.swap(kept);
      fused_head_ = true;
    }
  }
  loaded_ = true;
}

void Detector::forward(const uint8_t* imgs, int B, const ImgGeom* geom, float conf, float* out0, Cand* cand,
                       int* cand_count, hipStream_t st, Profiler* prof) {
  LP_CHECK(loaded_, LP_ERR_STATE, "detector not loaded");
  LP_CHECK(B >= 1 && B <= maxB_, LP_ERR_ARG, "batch %d outside 1..%d", B, maxB_);
  const char* sfx = prec_ == LP_FP16 ? "_f16" : "_f32";
  // Diagnostic only (tools/marginal_cost.sh): LITEPI_SKIP_OP=<i> leaves launch i out of every pass after this handle's first,
  // whose outputs stay in its (never re-used) buffers -- the marginal cost of one launch in a pipelined step.  Results are stale.
  static const int skip_op = getenv("LITEPI_SKIP_OP") ? atoi(getenv("LITEPI_SKIP_OP")) : -1;
  const bool skipping = skip_op >= 0 && fwd_calls_++ > 0 && !prof;
  int op_index = -1;
  for (const DetOp& op : ops_) {
    if (++op_index == skip_op && skipping) continue;
    if (prof) prof->begin(st);
    std::string kname;
    switch (op.kind) {
      case DetOp::STEM:
        stem_.launch(imgs, B, S_, S_, view(op.out), st);
        kname = std::string("stem_conv") + sfx;
        break;
      case DetOp::STEMBLOCK:
        stem_.launch_block(imgs, B, S_, S_, *convs_[op.conv], view(op.out), st);
        kname = std::string(stem_.CO == 16 ? "stem_block16" : "stem_block") + sfx;
        break;
      case DetOp::CONV: {
        const ConvLayer& c = *convs_[op.conv];
        ConvIO io;
        io.in = view(op.in); io.out = view(op.out); io.N = B;
        if (op.res >= 0) io.res = view(op.res);
        if (op.in2 >= 0) {  // fused upsample: leading channels from the half-resolution tensor, the rest from the concat buffer
          io.up = view(op.in2);
          io.in.base = static_cast<char*>(io.in.base) + (size_t)io.up.C * (prec_ == LP_FP16 ? 2 : 4);
          io.in.C -= io.up.C;
        }
        c.launch(io, st);
        kname = std::string(c.impl == IMPL_NAIVE ? "conv_naive" : (c.direct ? "conv3x3s2_direct" : (c.k == 3 ? "conv3x3_mfma" : "conv1x1_mfma"))) + (c.T2 ? "+1x1" : "");
        // one name per kernel instantiation (channel tiles NT, tail tiles T2, fused-upsample variant), as rocprofv3 lists them
        if (c.impl != IMPL_NAIVE) kname += c.T2 ? fmt("<%d,%d>", c.NT, c.T2) : (op.in2 >= 0 ? fmt("<%d,up>", c.NT) : fmt("<%d>", c.NT));
        kname += sfx;
        break;
      }
      case DetOp::BNECK:
        if (op.in2 >= 0) {
          const View catv = view(op.in2);
          bnecks_[op.conv]->launch(view(op.in), view(op.out), B, st, &catv);
        } else {
          bnecks_[op.conv]->launch(view(op.in), view(op.out), B, st);
        }
        {  // one name per kernel instantiation <NT, P1, P2, T2, SG> (what rocprofv3 lists as separate kernels)
          const BottleneckPair& bp = *bnecks_[op.conv];
          kname = fmt("bottleneck3x3x2<%d,%d,%d,%d,%d%s>", bp.NT, bp.p1(), bp.p2(), bp.T2, (bp.T2 > 0 && prec_ == LP_FP16) ? bp.sg : 0,
                      bp.cl ? ",cl" : "") + sfx;
        }
        break;
      case DetOp::DWCONV:
        launch_dwconv3x3_act(prec_, view(op.in), view(op.out), dws_[op.conv].w.as<float>(), dws_[op.conv].b.as<float>(), dws_[op.conv].act, B, st);
        kname = std::string("dwconv3x3_det") + sfx;
        break;
      case DetOp::ATTN: {
        const AttnLayer& A = attns_[op.conv];
        launch_psa_attention(prec_, view(op.in), view(op.out), A.pe_w.as<float>(), A.pe_b.as<float>(), A.heads, A.dk, A.dv, A.scale, B, st);
        kname = std::string("psa_attention") + sfx;
        break;
      }
      case DetOp::UPSAMPLE:
        launch_upsample2x(prec_, view(op.in), view(op.out), B, st);
        kname = std::string("upsample2x") + sfx;
        break;
      case DetOp::SPPF:
        launch_sppf_pool(prec_, view(op.in), view(op.out), view(op.out2), view(op.out3), B, st);
        kname = std::string("sppf_pool") + sfx;
        break;
      case DetOp::ADD:
        launch_add(prec_, view(op.in), view(op.in2), view(op.out), B, st);
        kname = std::string("add") + sfx;
        break;
      case DetOp::COPY:
        launch_copy(prec_, view(op.in), view(op.out), B, st);
        kname = std::string("copy") + sfx;
        break;
      case DetOp::C2F: {
        const C2fLayer& cl = *c2fs_[op.conv];
        const C2fIO& ci = c2f_io_[op.conv];
        C2fLayer::IO io;
        io.src1 = view(ci.src1);
        if (ci.src0 >= 0) {  // fused upsample: leading channels from the half-resolution tensor, the rest from the concat buffer
          io.src0 = view(ci.src0);
          io.src1.base = static_cast<char*>(io.src1.base) + (size_t)ci.up_c * 2;
          io.src1.C -= ci.up_c;
        }
        io.cat = view(ci.cat);
        io.out = view(ci.out);
        if (ci.s2_in >= 0) { io.s2_in = view(ci.s2_in); io.x = view(ci.x); }
        if (ci.cat2 >= 0) { io.cat2 = view(ci.cat2); io.out2 = view(ci.out2); }
        cl.launch(io, B, st);
        kname = cl.kernel_name() + sfx;
        break;
      }
      case DetOp::SPPFUSED:
        sppfs_[op.conv]->launch(view(op.in), view(op.out), B, st);
        kname = fmt("sppf<%d,%d,%d>", sppfs_[op.conv]->Cin, sppfs_[op.conv]->C, sppfs_[op.conv]->Cout) + sfx;
        break;
      case DetOp::S2C:
        s2cs_[op.conv]->launch(view(op.in), view(op.out), B, st);
        kname = fmt(s2cs_[op.conv]->has_tail ? "s2conv+1x1<%d,%d>" : "s2conv<%d,%d>", s2cs_[op.conv]->Cin, s2cs_[op.conv]->Cout) + sfx;
        break;
      case DetOp::HEAD:
        heads_[op.conv]->launch(view(op.in), B, levels_[op.in2].off, A_, d_anchors_.as<float>(), d_strides_.as<float>(), d_dfl_.as<float>(), out0,
                                geom, cand, cand_count, conf, st);
        kname = fmt("head_fused<%d,%d,%d,%d,%d,%d>", heads_[op.conv]->C3T, heads_[op.conv]->PA, heads_[op.conv]->PB, heads_[op.conv]->NPC,
                    heads_[op.conv]->KSA, heads_[op.conv]->SLOTF) + (heads_[op.conv]->A16 ? (heads_[op.conv]->C3T == 2 ? fmt("a16k%d", heads_[op.conv]->KPT) : std::string("a16")) : std::string()) + sfx;   // = the leading template arguments of head_fused_kernel
        break;
    }
    if (prof) prof->end(st, kname, op.layer, op.flops * B, op.bytes * B);
  }
  if (fused_head_) return;  // every level decoded and filtered inside its head kernel
  DecodeArgs a;
  memset(&a, 0, sizeof(a));
  a.nlevels = (int)levels_.size();
  for (int i = 0; i < a.nlevels; ++i) {
    const View b = view(levels_[i].box), c = view(levels_[i].cls);
    a.lv[i].box = b.base; a.lv[i].cls = c.base; a.lv[i].box_pitch = b.pitch; a.lv[i].cls_pitch = c.pitch;
    a.lv[i].H = levels_[i].H; a.lv[i].W = levels_[i].W; a.lv[i].anchor_off = levels_[i].off;
  }
  a.A = A_; a.nc = nc_; a.reg_max = reg_max_;
  a.anchors = d_anchors_.as<float>(); a.strides = d_strides_.as<float>(); a.dfl_w = d_dfl_.as<float>();
  a.out0 = out0; a.geom = geom; a.cand = cand; a.cand_count = cand_count; a.conf = conf;
  if (prof) prof->begin(st);
  launch_decode(prec_, a, B, st);
  if (prof)
    prof->end(st, std::string("decode") + sfx, "detect_decode", 0.0,
              (double)B * A_ * ((4.0 * reg_max_ + nc_) * (prec_ == LP_FP16 ? 2 : 4) + (out0 ? (4.0 + nc_) * 4 : 0.0)));
}

void Detector::fetch_blob(const std::string& name, int B, std::vector<float>& out, int& C, int& H, int& W) const {
  auto it = blob2tensor_.find(name);
  LP_CHECK(it != blob2tensor_.end(), LP_ERR_ARG, "unknown blob %s", name.c_str());
  const Tensor& T = tensors_[it->second];
  // a blob that a fused kernel keeps in registers / LDS has storage reserved (e.g. its concat slot) but is never written
  LP_CHECK(T.materialised || T.parent >= 0 || T.segs.size() > 1, LP_ERR_ARG, "blob %s is fused away (never stored)", name.c_str());
  LP_CHECK(!T.in_c2f && !(T.parent >= 0 && tensors_[T.parent].in_c2f), LP_ERR_ARG,
           "blob %s is fused away: it lives inside a whole-C2f launch (set LITEPI_C2F_STORE_ALL=1 before loading the model to have the "
           "module store its intermediates)", name.c_str());
  const View v = view(it->second);
  C = T.C; H = T.H; W = T.W;
  const size_t es = prec_ == LP_FP16 ? 2 : 4;
  const size_t npix = (size_t)B * H * W;
  // copy the pitch-wide rows of the underlying buffer and pick the view's channels out of them
  const Tensor& ST = T.parent >= 0 ? tensors_[T.parent] : T;
  const Buffer& buf = buffers_[ST.buf];
  const size_t voff = (size_t)(static_cast<const char*>(v.base) - static_cast<const char*>(buf.mem.p)) / es;
  std::vector<uint8_t> raw(npix * v.pitch * es);
  LP_HIP(hipMemcpy(raw.data(), buf.mem.p, raw.size(), hipMemcpyDeviceToHost));
  out.assign((size_t)B * C * H * W, 0.f);
  for (size_t p = 0; p < npix; ++p) {
    const size_t b = p / ((size_t)H * W), yx = p % ((size_t)H * W);
    for (int c = 0; c < C; ++c) {
      const int pc = T.phys(c);
      float f;
      if (prec_ == LP_FP16) {
        uint16_t h;
        memcpy(&h, &raw[(p * v.pitch + voff + pc) * 2], 2);
        f = f16_to_f32(h);
      } else {
        memcpy(&f, &raw[(p * v.pitch + voff + pc) * 4], 4);
      }
      out[(b * C + c) * H * W + yx] = f;
    }
  }
}

}  // namespace lp
