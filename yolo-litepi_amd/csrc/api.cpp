// C-ABI of liblitepi_hip.so (see include/litepi.h for the reference call sites each entry
// point replaces).  Owns the device, the stream, all activation/result buffers and the two
// model plans; every pipeline stage runs on the GPU -- there is no CPU fallback anywhere.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <atomic>
#include <condition_variable>
#include <dlfcn.h>
#include <cstring>
#include <mutex>
#include <set>
#include <thread>

#include "classifier.h"
#include "mbnet.h"
#include "resnet.h"
#include "common.h"
#include "copy_pool.h"
#include "detector.h"
#include "kernels.h"

namespace lp {

static thread_local std::string g_last_error;
void set_last_error(const std::string& s) { g_last_error = s; }

std::string fmt(const char* f, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, f);
  vsnprintf(buf, sizeof(buf), f, ap);
  va_end(ap);
  return std::string(buf);
}

uint16_t f32_to_f16(float f) {
  uint32_t x;
  memcpy(&x, &f, 4);
  const uint32_t sign = (x >> 16) & 0x8000u;
  const int32_t exp = (int32_t)((x >> 23) & 0xFF) - 127 + 15;
  uint32_t man = x & 0x7FFFFFu;
  if (((x >> 23) & 0xFF) == 0xFF) return (uint16_t)(sign | 0x7C00u | (man ? 0x200u : 0));
  if (exp >= 31) return (uint16_t)(sign | 0x7C00u);
  if (exp <= 0) {
    if (exp < -10) return (uint16_t)sign;
    man |= 0x800000u;
    const int shift = 14 - exp;
    uint32_t h = man >> shift;
    const uint32_t rem = man & ((1u << shift) - 1), halfway = 1u << (shift - 1);
    if (rem > halfway || (rem == halfway && (h & 1))) ++h;
    return (uint16_t)(sign | h);
  }
  uint32_t h = ((uint32_t)exp << 10) | (man >> 13);
  const uint32_t rem = man & 0x1FFFu;
  if (rem > 0x1000u || (rem == 0x1000u && (h & 1))) ++h;
  return (uint16_t)(sign | h);
}

float f16_to_f32(uint16_t h) {
  const uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
  uint32_t exp = (h >> 10) & 0x1F, man = h & 0x3FFu, x;
  if (exp == 0) {
    if (man == 0) {
      x = sign;
    } else {
      int e = -1;
      do { ++e; man <<= 1; } while (!(man & 0x400u));
      x = sign | ((uint32_t)(127 - 15 - e) << 23) | ((man & 0x3FFu) << 13);
    }
  } else if (exp == 31) {
    x = sign | 0x7F800000u | (man << 13);
  } else {
    x = sign | ((exp - 15 + 127) << 23) | (man << 13);
  }
  float f;
  memcpy(&f, &x, 4);
  return f;
}

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is a per-device attribute: set it once per (device, kernel) and
// check the result (a kernel that needs more than 64 KB of LDS fails to launch without it)
void set_max_dynamic_lds(const void* fn, int bytes) {
  static std::mutex mu;
  static std::set<std::pair<int, const void*>> done;
  int dev = 0;
  LP_HIP(hipGetDevice(&dev));
  std::lock_guard<std::mutex> lk(mu);
  if (done.count({dev, fn})) return;
  LP_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
  done.insert({dev, fn});
}

// letterbox geometry exactly as the reference computes it in Python doubles (e2e.py:72-83);
// Python's round() is round-half-to-even == nearbyint in the default rounding mode.
static ImgGeom make_geom(int h, int w, int S, long src_off) {
  ImgGeom g;
  memset(&g, 0, sizeof(g));  // padding bytes too: geometry is compared with memcmp
  const double r = std::min((double)S / h, (double)S / w);
  const int nw = (int)std::nearbyint(w * r), nh = (int)std::nearbyint(h * r);
  const double dw = (S - nw) / 2.0, dh = (S - nh) / 2.0;
  g.src_off = src_off; g.h = h; g.w = w; g.new_w = nw; g.new_h = nh;
  g.top = (int)std::nearbyint(dh - 0.1);
  g.left = (int)std::nearbyint(dw - 0.1);
  g.ratio = (float)r; g.pad_w = (float)dw; g.pad_h = (float)dh;
  return g;
}

}  // namespace lp

using namespace lp;

using lp::CopyPool;

// RCCL, bound at run time (lp_comm_* / lp_gather): the library links nothing but the HIP runtime
struct NcclId { char internal[128]; };
struct Rccl {
  void* lib = nullptr;
  int (*GetUniqueId)(NcclId*) = nullptr;
  int (*CommInitRank)(void**, int, NcclId, int) = nullptr;
  int (*Gather)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
  int (*CommDestroy)(void*) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
};
Rccl& rccl() {
  static Rccl r;
  static std::once_flag once;
  std::call_once(once, [] {
    for (const char* name : {"librccl.so.1", "librccl.so"}) {
      r.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
      if (r.lib) break;
    }
    if (!r.lib) return;
    r.GetUniqueId = reinterpret_cast<int (*)(NcclId*)>(dlsym(r.lib, "ncclGetUniqueId"));
    r.CommInitRank = reinterpret_cast<int (*)(void**, int, NcclId, int)>(dlsym(r.lib, "ncclCommInitRank"));
    r.Gather = reinterpret_cast<int (*)(const void*, void*, size_t, int, int, void*, hipStream_t)>(dlsym(r.lib, "ncclGather"));
    r.CommDestroy = reinterpret_cast<int (*)(void*)>(dlsym(r.lib, "ncclCommDestroy"));
    r.GetErrorString = reinterpret_cast<const char* (*)(int)>(dlsym(r.lib, "ncclGetErrorString"));
  });
  LP_CHECK(r.lib && r.GetUniqueId && r.CommInitRank && r.Gather && r.CommDestroy, LP_ERR_STATE,
           "RCCL is not available (dlopen librccl.so.1: %s)", r.lib ? "a symbol is missing" : dlerror());
  return r;
}
#define LP_RCCL(expr)                                                                                                           \
  do {                                                                                                                          \
    const int rc_ = (expr);                                                                                                     \
    if (rc_ != 0) throw Error(LP_ERR_HIP, fmt("%s: RCCL error %d (%s)", #expr, rc_, rccl().GetErrorString ? rccl().GetErrorString(rc_) : "?")); \
  } while (0)

struct lp_handle {
  lp_config cfg;
  void* comm = nullptr;        // ncclComm_t of lp_comm_init
  int comm_rank = 0, comm_world = 1;
  // pinned staging of the host entry points + the copy workers (created on first use)
  uint8_t* h_stage = nullptr;
  size_t h_stage_bytes = 0;
  std::unique_ptr<CopyPool> pool;
  hipStream_t own_stream = nullptr, stream = nullptr;
  std::unique_ptr<Detector> det;
  std::unique_ptr<ClassifierBase> cls;
  Profiler prof;
  bool prof_next = false;
  int max_rois = 0;
  // device buffers
  DevBuf d_src, d_lb, d_geom, d_cand, d_cand_count, d_sorted, d_dets, d_counts, d_rects, d_out0;
  DevBuf d_roi_base, d_roi_total, d_roi_img, d_roi_slot, d_roi_rgb, d_probs, d_ids, d_conf;
  std::vector<ImgGeom> geom_cache;
  hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};   // lp_run_batch: start, after the ROI resize, after detect + NMS, end
  int last_roi_count = 0;
  // chunked lp_run_batch (uniform frames, >= 32 of them): uploads on copy_stream, five events per chunk, ROI totals per chunk
  hipStream_t copy_stream = nullptr;
  std::vector<hipEvent_t> chunk_ev;
  DevBuf d_roi_hist;
  // ---- captured steps: the launch sequence of a call is a pure function of (entry point, buffers, batch, geometry,
  //      thresholds), so the second call with the same key is captured into a hipGraph and later calls replay it
  //      (one hipGraphLaunch instead of ~40 kernel launches on the host).  LITEPI_NO_GRAPH=1 keeps every call eager.
  struct GraphKey {
    int kind, B, geom_ver, min_area;
    const void* p0; void* p1; void* p2;
    float conf, iou;
    bool operator==(const GraphKey& o) const {
      return kind == o.kind && B == o.B && geom_ver == o.geom_ver && min_area == o.min_area && p0 == o.p0 && p1 == o.p1 && p2 == o.p2 &&
             conf == o.conf && iou == o.iou;
    }
  };
  struct GraphEntry { GraphKey key; hipGraph_t graph = nullptr; hipGraphExec_t exec = nullptr; bool failed = false; unsigned long stamp = 0; };
  std::vector<GraphEntry> graphs;
  int geom_ver = 0;
  unsigned long graph_clock = 0;
  void drop_graphs() {
    for (auto& g : graphs) {
      if (g.exec) (void)hipGraphExecDestroy(g.exec);
      if (g.graph) (void)hipGraphDestroy(g.graph);
    }
    graphs.clear();
  }

  RoiTable roi_table() {
    RoiTable t;
    t.base = d_roi_base.as<int>(); t.total = d_roi_total.as<int>(); t.work = d_roi_total.as<int>() + 4;
    t.img = d_roi_img.as<int>(); t.slot = d_roi_slot.as<int>();
    return t;
  }
  void ensure_src(size_t bytes) {
    if (d_src.bytes < bytes) {
      d_src.alloc(bytes + bytes / 4, false);
      ++geom_ver;  // captured steps hold the old address
    }
  }
  void alloc_post_buffers() {
    const int B = cfg.max_batch, A = det->num_anchors(), nc = det->num_classes();
    d_cand.alloc((size_t)B * A * sizeof(Cand), false);
    d_sorted.alloc((size_t)B * A * sizeof(Cand), false);
    d_cand_count.alloc((size_t)B * 4);
    d_dets.alloc((size_t)B * cfg.max_det * sizeof(lp_det));
    d_counts.alloc((size_t)3 * B * 4);
    d_rects.alloc((size_t)B * cfg.max_det * 16);
    d_out0.alloc((size_t)B * (4 + nc) * A * 4, false);
  }
  void upload_geom(const std::vector<ImgGeom>& g) {
    bool same = g.size() == geom_cache.size() && (g.empty() || memcmp(g.data(), geom_cache.data(), g.size() * sizeof(ImgGeom)) == 0);
    if (same) return;
    LP_HIP(hipMemcpyAsync(d_geom.p, g.data(), g.size() * sizeof(ImgGeom), hipMemcpyHostToDevice, stream));
    LP_HIP(hipStreamSynchronize(stream));  // g may be a temporary; uploads are rare (shape changes only)
    geom_cache = g;
    ++geom_ver;
  }
};

#define LP_API_BEGIN try {
#define LP_API_END                                   \
  }                                                  \
  catch (const lp::Error& e) {                       \
    lp::set_last_error(e.what());                    \
    return e.code;                                   \
  }                                                  \
  catch (const std::exception& e) {                  \
    lp::set_last_error(e.what());                    \
    return LP_ERR_STATE;                             \
  }                                                  \
  return LP_OK;

extern "C" {

const char* lp_last_error(void) { return lp::g_last_error.c_str(); }
int lp_version(void) { return LP_ABI_VERSION; }

void lp_default_config(lp_config* c) {
  memset(c, 0, sizeof(*c));
  c->device = 0; c->precision = LP_FP16; c->max_batch = 1; c->max_det = 300; c->num_classes = 58;
  c->det_input = 640; c->cls_input = 64; c->max_rois = 0; c->conv_impl = 0;
}

int lp_create(const lp_config* cfg, lp_handle** out) {
  LP_API_BEGIN
  LP_CHECK(cfg && out, LP_ERR_ARG, "null argument");
  LP_CHECK(cfg->max_batch >= 1 && cfg->max_batch <= 1024 && cfg->max_det >= 1 && cfg->det_input % 32 == 0 && cfg->det_input >= 64,
           LP_ERR_ARG, "bad config (max_batch %d, max_det %d, det_input %d)", cfg->max_batch, cfg->max_det, cfg->det_input);
  LP_CHECK((cfg->numerics == 0 || cfg->numerics == 1) && cfg->cls_arch >= 0 && cfg->cls_arch <= LP_CLS_EFFICIENTNET_B0, LP_ERR_ARG,
           "bad config (numerics %d, cls_arch %d)", cfg->numerics, cfg->cls_arch);
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= cfg->device)
    throw Error(LP_ERR_NODEVICE, fmt("HIP device %d not available (%d visible): liblitepi_hip has no CPU path", cfg->device, ndev));
  LP_HIP(hipSetDevice(cfg->device));
  hipDeviceProp_t prop;
  LP_HIP(hipGetDeviceProperties(&prop, cfg->device));
  if (std::string(prop.gcnArchName).find("gfx950") == std::string::npos)
    throw Error(LP_ERR_NODEVICE, fmt("device %d is %s; this library is built for gfx950 only", cfg->device, prop.gcnArchName));
  // the two measurement diagnostics make every pass after a handle's first return STALE results with LP_OK: never silently
  {
    static bool warned = false;
    const char *ss = getenv("LITEPI_SKIP_STAGE"), *so = getenv("LITEPI_SKIP_OP");
    if ((ss || so) && !warned) {
      warned = true;
      fprintf(stderr, "[litepi] WARNING: LITEPI_SKIP_STAGE=%s LITEPI_SKIP_OP=%s -- diagnostic mode (tools/marginal_cost.sh): a pipeline stage / "
                      "detector launch is LEFT OUT of every pass after a handle's first; results are stale and must not be used\n",
              ss ? ss : "", so ? so : "");
    }
  }
  std::unique_ptr<lp_handle> h(new lp_handle());
  h->cfg = *cfg;
  // every kept box is a ROI (the reference classifies all of them, e2e.py:493-497): the default capacity can not overflow
  h->max_rois = cfg->max_rois > 0 ? cfg->max_rois : cfg->max_batch * cfg->max_det;
  // the classifier's widest per-ROI tensor must stay below 2^31 elements (32-bit element offsets in the conv kernels): say so
  // here, not at the first launch.  ShuffleNetV2 / ResNet18 at cls_input S: conv1 output 24 (64) channels at (S/2)^2.
  {
    const double per_roi = cfg->cls_arch >= LP_CLS_MOBILENETV2 ? MBNetClassifier::widest_per_roi(cfg->cls_input)
                                                               : (double)(cfg->cls_arch == LP_CLS_RESNET18 ? 64 : 24) * (cfg->cls_input / 2.0) * (cfg->cls_input / 2.0);
    LP_CHECK(per_roi * h->max_rois < 2147483648.0, LP_ERR_ARG,
             "max_rois = %d (max_batch %d x max_det %d when left 0) makes the classifier's activations exceed 2^31 elements; lower "
             "max_det or set max_rois (at most %d for this classifier)", h->max_rois, cfg->max_batch, cfg->max_det, (int)(2147483647.0 / per_roi));
  }
  LP_HIP(hipStreamCreateWithFlags(&h->own_stream, hipStreamNonBlocking));
  h->stream = h->own_stream;
  for (auto& e : h->ev) LP_HIP(hipEventCreate(&e));
  h->d_geom.alloc((size_t)std::max(cfg->max_batch, h->max_rois) * sizeof(ImgGeom));
  const int S = cfg->det_input;
  h->d_lb.alloc((size_t)cfg->max_batch * S * S * 3, false);
  h->d_roi_base.alloc((size_t)(cfg->max_batch + 1) * 4);
  h->d_roi_total.alloc(32);  // total, unclamped total | accumulator, ticket (kernels.h RoiTable)
  h->d_roi_img.alloc((size_t)h->max_rois * 4);
  h->d_roi_slot.alloc((size_t)h->max_rois * 4);
  const int cs = cfg->cls_input;
  h->d_roi_rgb.alloc((size_t)h->max_rois * cs * cs * 3, false);
  h->d_probs.alloc((size_t)h->max_rois * std::max(cfg->num_classes, 1) * 4);
  h->d_ids.alloc((size_t)h->max_rois * 4);
  h->d_conf.alloc((size_t)h->max_rois * 4);
  *out = h.release();
  LP_API_END
}

void lp_destroy(lp_handle* h) {
  if (!h) return;
  (void)hipSetDevice(h->cfg.device);
  (void)hipDeviceSynchronize();
  h->drop_graphs();
  for (auto& e : h->ev)
    if (e) (void)hipEventDestroy(e);
  if (h->own_stream) (void)hipStreamDestroy(h->own_stream);
  if (h->copy_stream) (void)hipStreamDestroy(h->copy_stream);
  for (auto& e : h->chunk_ev)
    if (e) (void)hipEventDestroy(e);
  h->pool.reset();
  if (h->h_stage) (void)hipHostFree(h->h_stage);
  if (h->comm) { try { (void)rccl().CommDestroy(h->comm); } catch (...) {} }
  delete h;
}

int lp_load_detector_ncnn(lp_handle* h, const char* param_path, const char* bin_path) {
  LP_API_BEGIN
  LP_CHECK(h && param_path && bin_path, LP_ERR_ARG, "null argument");
  LP_HIP(hipSetDevice(h->cfg.device));
  std::unique_ptr<Detector> d(new Detector(h->cfg.precision, h->cfg.conv_impl, h->cfg.max_batch, h->cfg.det_input));
  d->load(param_path, bin_path);
  h->drop_graphs();
  h->det = std::move(d);
  h->alloc_post_buffers();
  LP_HIP(hipDeviceSynchronize());
  LP_API_END
}

int lp_load_classifier_tensors(lp_handle* h, int n, const char* const* names, const float* const* data,
                               const int64_t* const* shapes, const int* ndims) {
  LP_API_BEGIN
  LP_CHECK(h && names && data && shapes && ndims && n > 0, LP_ERR_ARG, "null argument");
  LP_HIP(hipSetDevice(h->cfg.device));
  std::map<std::string, NamedTensor> sd;
  for (int i = 0; i < n; ++i) {
    NamedTensor t;
    t.data = data[i];
    t.shape.assign(shapes[i], shapes[i] + ndims[i]);
    sd[names[i]] = t;
  }
  std::unique_ptr<ClassifierBase> c;
  if (h->cfg.cls_arch == LP_CLS_RESNET18) c.reset(new ResNet18Classifier(h->cfg.precision, h->cfg.conv_impl, h->max_rois, h->cfg.num_classes, h->cfg.cls_input));
  else if (h->cfg.cls_arch == LP_CLS_MOBILENETV2 || h->cfg.cls_arch == LP_CLS_EFFICIENTNET_B0)
    c.reset(new MBNetClassifier(h->cfg.cls_arch == LP_CLS_MOBILENETV2 ? MBNetClassifier::MOBILENET_V2 : MBNetClassifier::EFFICIENTNET_B0, h->cfg.precision,
                                h->cfg.conv_impl, h->max_rois, h->cfg.num_classes, h->cfg.cls_input));
  else c.reset(new Classifier(h->cfg.precision, h->cfg.conv_impl, h->max_rois, h->cfg.num_classes, h->cfg.cls_input));
  c->load(sd);
  h->drop_graphs();
  h->cls = std::move(c);
  LP_HIP(hipDeviceSynchronize());
  LP_API_END
}

int lp_set_stream(lp_handle* h, void* s) {
  LP_API_BEGIN
  LP_CHECK(h, LP_ERR_ARG, "null handle");
  h->stream = s ? reinterpret_cast<hipStream_t>(s) : h->own_stream;
  LP_API_END
}

int lp_synchronize(lp_handle* h) {
  LP_API_BEGIN
  LP_CHECK(h, LP_ERR_ARG, "null handle");
  LP_HIP(hipStreamSynchronize(h->stream));
  LP_API_END
}

int lp_profile_next(lp_handle* h, int enable) {
  LP_API_BEGIN
  LP_CHECK(h, LP_ERR_ARG, "null handle");
  h->prof_next = enable != 0;
  LP_API_END
}

int lp_profile_read(lp_handle* h, lp_kernel_time* out, int cap, int* n) {
  LP_API_BEGIN
  LP_CHECK(h && n, LP_ERR_ARG, "null argument");
  if (!h->prof.recs.empty()) {
    LP_HIP(hipStreamSynchronize(h->stream));
    int R = 0;
    LP_HIP(hipMemcpy(&R, h->d_roi_total.p, 4, hipMemcpyDeviceToHost));
    h->prof.collect(R);
  }
  const int m = std::min<int>(cap, (int)h->prof.results.size());
  for (int i = 0; i < m && out; ++i) out[i] = h->prof.results[i];
  *n = (int)h->prof.results.size();
  LP_API_END
}

int lp_detector_info(lp_handle* h, int* num_anchors, int* nc, int* reg_max, double* macs) {
  LP_API_BEGIN
  LP_CHECK(h && h->det && h->det->loaded(), LP_ERR_STATE, "detector not loaded");
  if (num_anchors) *num_anchors = h->det->num_anchors();
  if (nc) *nc = h->det->num_classes();
  if (reg_max) *reg_max = h->det->reg_max();
  if (macs) *macs = h->det->macs_per_image();
  LP_API_END
}

}  // extern "C"

// ---- pipeline pieces shared by the entry points ----------------------------------------------
namespace {

Profiler* begin_profile(lp_handle* h) {
  if (!h->prof_next) return nullptr;
  h->prof_next = false;
  h->prof.enabled = true;
  h->prof.results.clear();
  return &h->prof;
}

// detector (+ optional letterbox) on images resident at src with geometry already uploaded
void enqueue_detect(lp_handle* h, const uint8_t* src, const std::vector<ImgGeom>& geoms, int B, float conf, float* out0,
                    Profiler* prof) {
  const int S = h->cfg.det_input;
  bool identity = true;
  for (int i = 0; i < B; ++i)
    identity = identity && geoms[i].h == S && geoms[i].w == S && geoms[i].src_off == (long)i * S * S * 3;
  const uint8_t* img = src;
  if (!identity) {
    if (prof) prof->begin(h->stream);
    launch_letterbox(src, h->d_geom.as<ImgGeom>(), h->d_lb.as<uint8_t>(), B, S, h->stream, geoms.data());
    if (prof) {
      double bytes = (double)B * S * S * 3;
      for (int i = 0; i < B; ++i) bytes += (double)geoms[i].h * geoms[i].w * 3;
      prof->end(h->stream, "letterbox_u8", "letterbox", 0.0, bytes);
    }
    img = h->d_lb.as<uint8_t>();
  }
  h->det->forward(img, B, h->d_geom.as<ImgGeom>(), conf, out0, h->d_cand.as<Cand>(), h->d_cand_count.as<int>(), h->stream, prof);
}

// Diagnostic only (tools/marginal_cost.sh): LITEPI_SKIP_STAGE=nms|roi|cls leaves that stage out of every pass after the handle's
// first (its outputs stay in the handle's buffers): the marginal cost of the stage in a pipelined step.  Results are stale.
static bool skip_stage(const lp_handle* h, const char* name, const Profiler* prof) {
  static const char* s = getenv("LITEPI_SKIP_STAGE");
  return s && !prof && h->graph_clock > 1 && strcmp(s, name) == 0;
}

// NMS + ROI rectangles; with_rois: also the batch-wide ROI list the classifier stage consumes
void enqueue_nms(lp_handle* h, int B, float iou, int min_area, lp_det* dets, int* counts, bool with_rois, Profiler* prof) {
  NmsArgs a;
  memset(&a, 0, sizeof(a));
  a.cand = h->d_cand.as<Cand>(); a.cand_count = h->d_cand_count.as<int>(); a.sorted = h->d_sorted.as<Cand>();
  a.dets = dets; a.counts = counts; a.rects = h->d_rects.as<int>(); a.geom = h->d_geom.as<ImgGeom>();
  a.A = h->det->num_anchors(); a.max_det = h->cfg.max_det; a.nc = h->det->num_classes(); a.iou = iou; a.min_area = min_area;
  if (with_rois) a.tab = h->roi_table();
  a.max_rois = h->max_rois;
  a.roi_rule = h->cfg.numerics;
  if (skip_stage(h, "nms", prof)) return;
  if (prof) prof->begin(h->stream);
  launch_nms(a, B, h->stream);
  if (prof) prof->end(h->stream, "nms", "nms", 0.0, 0.0);
}

// PIL resize + ShuffleNetV2 + softmax over the ROI list; scatters (cls, conf) into dets when given
// stage: 0 = both halves, 1 = only the ROI crop + resize (the device's share of the reference's ROI loop, e2e.py:460-475),
// 2 = only the classifier (lp_run_batch times the two separately: PipelineMetrics.t_roi_extract / t_classification)
void enqueue_classify(lp_handle* h, const uint8_t* src, int B, lp_det* dets, float* probs, int* ids, float* conf, Profiler* prof, int stage = 0) {
  RoiTable tab = h->roi_table();
  if (stage != 2 && !skip_stage(h, "roi", prof)) {
    RoiResizeArgs r;
    r.src = src; r.geom = h->d_geom.as<ImgGeom>(); r.rects = h->d_rects.as<int>(); r.tab = tab;
    r.out = h->d_roi_rgb.as<uint8_t>(); r.max_det = h->cfg.max_det; r.S = h->cfg.cls_input; r.linear = h->cfg.numerics;
    if (prof) prof->begin(h->stream);
    launch_roi_resize(r, std::min(h->max_rois, B * h->cfg.max_det), h->stream);
    if (prof) prof->end(h->stream, "roi_resize_pil", "roi_resize", 0.0, (double)r.S * r.S * 3 * 2, true);
  }
  if (stage == 1 || skip_stage(h, "cls", prof)) return;
  ClsPost post;
  post.probs = probs; post.ids = ids; post.dets = dets; post.max_det = h->cfg.max_det; post.roi_img = tab.img; post.roi_slot = tab.slot;
  h->cls->forward(h->d_roi_rgb.as<uint8_t>(), tab.total, h->stream, prof, &post);
  if (!h->cls->fused_head()) {
    if (prof) prof->begin(h->stream);
    launch_softmax_argmax(h->cls->logits(), h->cls->logits_pitch(), h->cls->num_classes(), probs, ids, conf, dets, h->cfg.max_det,
                          &tab, tab.total, h->max_rois, h->stream);
    if (prof) prof->end(h->stream, "softmax_argmax", "softmax", 0.0, (double)h->cls->num_classes() * 8, true);
  }
}

// Run `enqueue` (kernel launches on h->stream only: no allocation, no synchronisation) eagerly the first time a key is
// seen -- that call also performs every one-time set-up (LDS attributes, lazy packing) --, capture it into a hipGraph the
// second time, replay the graph from then on.
template <typename F>
void run_or_capture(lp_handle* h, const lp_handle::GraphKey& key, bool allow, F&& enqueue) {
  static const bool disabled = getenv("LITEPI_NO_GRAPH") != nullptr;
  if (disabled || !allow) { enqueue(); return; }
  lp_handle::GraphEntry* e = nullptr;
  for (auto& g : h->graphs)
    if (g.key == key) { e = &g; break; }
  if (!e) {  // first sight: eager, remember the key
    if (h->graphs.size() >= 32) {  // evict the least recently used entry
      size_t lru = 0;
      for (size_t i = 1; i < h->graphs.size(); ++i)
        if (h->graphs[i].stamp < h->graphs[lru].stamp) lru = i;
      if (h->graphs[lru].exec) (void)hipGraphExecDestroy(h->graphs[lru].exec);
      if (h->graphs[lru].graph) (void)hipGraphDestroy(h->graphs[lru].graph);
      h->graphs.erase(h->graphs.begin() + lru);
    }
    lp_handle::GraphEntry ne;
    ne.key = key;
    ne.stamp = ++h->graph_clock;
    h->graphs.push_back(ne);
    enqueue();
    return;
  }
  e->stamp = ++h->graph_clock;
  if (e->failed) { enqueue(); return; }
  if (!e->exec) {
    if (hipStreamBeginCapture(h->stream, hipStreamCaptureModeThreadLocal) != hipSuccess) { e->failed = true; enqueue(); return; }
    bool ok = true;
    std::string err;
    try { enqueue(); } catch (const lp::Error& ex) { ok = false; err = ex.what(); }
    hipGraph_t graph = nullptr;
    if (hipStreamEndCapture(h->stream, &graph) != hipSuccess || !graph) ok = false;
    if (ok && hipGraphInstantiate(&e->exec, graph, nullptr, nullptr, 0) != hipSuccess) { ok = false; e->exec = nullptr; }
    if (!ok) {
      if (graph) (void)hipGraphDestroy(graph);
      (void)hipGetLastError();
      e->failed = true;
      LP_CHECK(err.empty(), LP_ERR_STATE, "%s", err.c_str());
      enqueue();
      return;
    }
    e->graph = graph;
  }
  LP_HIP(hipGraphLaunch(e->exec, h->stream));
}

// upload B host images of individual sizes into d_src; returns their geometry
std::vector<ImgGeom> upload_images(lp_handle* h, const uint8_t* const* imgs, const int* hs, const int* ws, int B) {
  std::vector<ImgGeom> g(B);
  size_t total = 0;
  for (int i = 0; i < B; ++i) {
    LP_CHECK(imgs[i] && hs[i] > 0 && ws[i] > 0, LP_ERR_ARG, "image %d is empty", i);
    g[i] = make_geom(hs[i], ws[i], h->cfg.det_input, (long)total);
    total += (size_t)hs[i] * ws[i] * 3;
    total = (total + 15) & ~(size_t)15;
  }
  h->ensure_src(total);
  // small uploads (a single frame: the batch-1 latency path) go straight from the caller's memory
  static const int n_threads = getenv("LITEPI_UPLOAD_THREADS") ? atoi(getenv("LITEPI_UPLOAD_THREADS")) : 8;
  if (n_threads <= 0 || B < 4 || total < ((size_t)4 << 20)) {
    for (int i = 0; i < B; ++i)
      LP_HIP(hipMemcpyAsync(h->d_src.as<uint8_t>() + g[i].src_off, imgs[i], (size_t)hs[i] * ws[i] * 3, hipMemcpyHostToDevice, h->stream));
    return g;
  }
  if (h->h_stage_bytes < total) {
    if (h->h_stage) { LP_HIP(hipStreamSynchronize(h->stream)); (void)hipHostFree(h->h_stage); h->h_stage = nullptr; h->h_stage_bytes = 0; }
    LP_HIP(hipHostMalloc(reinterpret_cast<void**>(&h->h_stage), total + total / 4, hipHostMallocDefault));
    h->h_stage_bytes = total + total / 4;
  }
  if (!h->pool) h->pool.reset(new CopyPool(n_threads - 1));
  // (the previous call synchronised the stream before it returned: the staging buffer is free)
  // groups of about 10 MB: the workers fill group k+1 while the DMA engine moves group k; an image larger than that is split
  // into slices so that every worker has a share
  std::vector<CopyPool::Job> jobs;
  // (the first groups are small: nothing overlaps the first group's copy, the link idles until it is staged)
  static const size_t group_mb = getenv("LITEPI_UPLOAD_GROUP_MB") ? (size_t)atol(getenv("LITEPI_UPLOAD_GROUP_MB")) : 10;
  const size_t slice = (size_t)1 << 20;
  int i0 = 0, ngroup = 0;
  while (i0 < B) {
    int i1 = i0;
    size_t gb = 0;
    const size_t group_bytes = ngroup == 0 ? (size_t)2 << 20 : (ngroup == 1 ? (size_t)5 << 20 : group_mb << 20);
    ++ngroup;
    jobs.clear();
    while (i1 < B && (i1 == i0 || gb + (size_t)hs[i1] * ws[i1] * 3 <= group_bytes)) {
      const size_t nb = (size_t)hs[i1] * ws[i1] * 3;
      for (size_t o = 0; o < nb; o += slice) jobs.push_back({imgs[i1] + o, h->h_stage + g[i1].src_off + o, std::min(slice, nb - o)});
      gb += nb;
      ++i1;
    }
    h->pool->run(jobs.data(), (int)jobs.size());
    const size_t lo = (size_t)g[i0].src_off, hi = (size_t)g[i1 - 1].src_off + (size_t)hs[i1 - 1] * ws[i1 - 1] * 3;
    LP_HIP(hipMemcpyAsync(h->d_src.as<uint8_t>() + lo, h->h_stage + lo, hi - lo, hipMemcpyHostToDevice, h->stream));
    i0 = i1;
  }
  return g;
}


// ---- lp_run_batch on B >= 32 frames of one size, LITEPI_RUN_CHUNK=<frames> (OFF by default): the batch goes through the
//      handle in chunks.  Chunk k's frames are staged and sent on a copy stream while chunk k-1's kernels run on the handle's
//      stream (an event per chunk orders the two); the frames are independent, a chunk is a complete pass (detect -> NMS ->
//      ROI resize -> classifier) over its slice of d_src with its slice of the record / count buffers, and the results are
//      bit-identical to the whole-batch pass (tests/test_gpu_device_path.py).  Measured on a 64-frame call (tools/dropin_probe.py,
//      one box): whole batch 2.76 ms, chunks of 32: 2.74, of 16: 3.18, of 8: 5.0 -- a pass through the 20-launch pipeline costs
//      0.43 ms + 7 us per frame, so k chunks add (k - 1) x 0.43 ms of kernel time while hiding at most the kernels of k - 1
//      chunks under the 1.7 ms upload: no gain at any split, hence off.
int run_chunk_frames() {   // (read per call: tests switch it inside one process)
  const char* e = getenv("LITEPI_RUN_CHUNK");
  return e ? atoi(e) : 0;
}
bool chunked_ok(const lp_handle* h, const int* hs, const int* ws, int B) {
  const int c = run_chunk_frames();
  if (c <= 0 || B < 32 || B < 2 * c || h->prof_next) return false;
  for (int i = 1; i < B; ++i)
    if (hs[i] != hs[0] || ws[i] != ws[0]) return false;
  return ((size_t)hs[0] * ws[0] * 3) % 16 == 0;
}
void run_batch_chunked(lp_handle* h, const uint8_t* const* imgs, int H, int W, int B, float conf, float iou, int min_area, lp_det* dets,
                       int* counts, int* num_det, float* det_conf_avg, lp_timing* timing) {
  const int CH = run_chunk_frames(), nch = (B + CH - 1) / CH;
  const size_t img_bytes = (size_t)H * W * 3, total = img_bytes * B;
  for (int i = 0; i < B; ++i) LP_CHECK(imgs[i], LP_ERR_ARG, "image %d is empty", i);
  h->ensure_src(total);
  if (h->h_stage_bytes < total) {
    if (h->h_stage) { LP_HIP(hipStreamSynchronize(h->stream)); (void)hipHostFree(h->h_stage); h->h_stage = nullptr; h->h_stage_bytes = 0; }
    LP_HIP(hipHostMalloc(reinterpret_cast<void**>(&h->h_stage), total + total / 4, hipHostMallocDefault));
    h->h_stage_bytes = total + total / 4;
  }
  static const int n_threads = getenv("LITEPI_UPLOAD_THREADS") ? atoi(getenv("LITEPI_UPLOAD_THREADS")) : 8;
  if (!h->pool) h->pool.reset(new CopyPool(std::max(n_threads, 1) - 1));
  if (!h->copy_stream) LP_HIP(hipStreamCreateWithFlags(&h->copy_stream, hipStreamNonBlocking));
  while ((int)h->chunk_ev.size() < 5 * nch) {
    hipEvent_t e = nullptr;
    LP_HIP(hipEventCreate(&e));
    h->chunk_ev.push_back(e);
  }
  if (h->d_roi_hist.bytes < (size_t)nch * 8) h->d_roi_hist.alloc((size_t)std::max(nch, 64) * 8);
  // every chunk sees the same geometry: frame i of the chunk at i * img_bytes behind the chunk's base
  std::vector<ImgGeom> g(CH);
  for (int i = 0; i < CH; ++i) g[i] = make_geom(H, W, h->cfg.det_input, (long)(i * img_bytes));
  h->upload_geom(g);
  std::vector<CopyPool::Job> jobs;
  const size_t slice = (size_t)1 << 20;
  for (int k = 0; k < nch; ++k) {
    const int lo = k * CH, n = std::min(CH, B - lo);
    hipEvent_t* ev = &h->chunk_ev[5 * k];
    // stage + send: groups of 4 frames (the first one of the call: a single frame, nothing overlaps its copy)
    int i0 = lo;
    while (i0 < lo + n) {
      const int i1 = std::min(lo + n, i0 + ((k == 0 && i0 == lo) ? 1 : 4));
      jobs.clear();
      for (int i = i0; i < i1; ++i)
        for (size_t o = 0; o < img_bytes; o += slice) jobs.push_back({imgs[i] + o, h->h_stage + i * img_bytes + o, std::min(slice, img_bytes - o)});
      h->pool->run(jobs.data(), (int)jobs.size());
      LP_HIP(hipMemcpyAsync(h->d_src.as<uint8_t>() + i0 * img_bytes, h->h_stage + i0 * img_bytes, (i1 - i0) * img_bytes, hipMemcpyHostToDevice,
                            h->copy_stream));
      i0 = i1;
    }
    LP_HIP(hipEventRecord(ev[4], h->copy_stream));
    LP_HIP(hipStreamWaitEvent(h->stream, ev[4], 0));
    const uint8_t* src = h->d_src.as<uint8_t>() + lo * img_bytes;
    lp_det* d = h->d_dets.as<lp_det>() + (size_t)lo * h->cfg.max_det;
    int* c = h->d_counts.as<int>() + 3 * lo;
    std::vector<ImgGeom> gk(g.begin(), g.begin() + n);
    LP_HIP(hipEventRecord(ev[0], h->stream));
    lp_handle::GraphKey k1{6, n, h->geom_ver, min_area, src, d, c, conf, iou};
    run_or_capture(h, k1, true, [&]() {
      enqueue_detect(h, src, gk, n, conf, nullptr, nullptr);
      enqueue_nms(h, n, iou, min_area, d, c, true, nullptr);
    });
    LP_HIP(hipEventRecord(ev[1], h->stream));
    lp_handle::GraphKey k2{7, n, h->geom_ver, min_area, src, d, c, conf, iou};
    run_or_capture(h, k2, true, [&]() { enqueue_classify(h, src, n, d, nullptr, nullptr, nullptr, nullptr, 1); });
    LP_HIP(hipEventRecord(ev[2], h->stream));
    lp_handle::GraphKey k3{8, n, h->geom_ver, min_area, src, d, c, conf, iou};
    run_or_capture(h, k3, true, [&]() { enqueue_classify(h, src, n, d, nullptr, nullptr, nullptr, nullptr, 2); });
    LP_HIP(hipEventRecord(ev[3], h->stream));
    LP_HIP(hipMemcpyAsync(h->d_roi_hist.as<int>() + 2 * k, h->d_roi_total.p, 8, hipMemcpyDeviceToDevice, h->stream));
  }
  LP_HIP(hipMemcpyAsync(dets, h->d_dets.p, (size_t)B * h->cfg.max_det * sizeof(lp_det), hipMemcpyDeviceToHost, h->stream));
  std::vector<int> cnt(3 * B), R(2 * nch);
  LP_HIP(hipMemcpyAsync(cnt.data(), h->d_counts.p, (size_t)3 * B * 4, hipMemcpyDeviceToHost, h->stream));
  LP_HIP(hipMemcpyAsync(R.data(), h->d_roi_hist.p, (size_t)nch * 8, hipMemcpyDeviceToHost, h->stream));
  LP_HIP(hipStreamSynchronize(h->stream));
  int kept_rois = 0, want_rois = 0;
  float t_det = 0.f, t_roi = 0.f, t_cls = 0.f, t_all = 0.f;
  for (int k = 0; k < nch; ++k) {
    const int lo = k * CH, n = std::min(CH, B - lo);
    for (int i = 0; i < n; ++i) {
      counts[lo + i] = cnt[3 * lo + i];
      if (num_det) num_det[lo + i] = cnt[3 * lo + n + i];
      if (det_conf_avg) memcpy(&det_conf_avg[lo + i], &cnt[3 * lo + 2 * n + i], 4);
    }
    kept_rois += R[2 * k];
    want_rois = std::max(want_rois, R[2 * k + 1]);
    if (timing) {
      float a = 0.f, b = 0.f, c2 = 0.f;
      (void)hipEventElapsedTime(&a, h->chunk_ev[5 * k], h->chunk_ev[5 * k + 1]);
      (void)hipEventElapsedTime(&b, h->chunk_ev[5 * k + 1], h->chunk_ev[5 * k + 2]);
      (void)hipEventElapsedTime(&c2, h->chunk_ev[5 * k + 2], h->chunk_ev[5 * k + 3]);
      t_det += a; t_roi += b; t_cls += c2;
    }
  }
  h->last_roi_count = kept_rois;
  if (timing) {
    (void)hipEventElapsedTime(&t_all, h->chunk_ev[0], h->chunk_ev[5 * (nch - 1) + 3]);
    timing->t_detection = t_det; timing->t_roi_extract = t_roi; timing->t_classification = t_cls; timing->t_total = t_all;
  }
  LP_CHECK(want_rois <= h->max_rois, LP_ERR_STATE, "%d ROIs in one chunk of this batch exceed max_rois = %d: %d detections were left unclassified",
           want_rois, h->max_rois, want_rois - h->max_rois);
}

}  // namespace

extern "C" {

int lp_detect_raw(lp_handle* h, const uint8_t* bgr, int B, float* out0) {
  LP_API_BEGIN
  LP_CHECK(h && bgr && out0, LP_ERR_ARG, "null argument");
  LP_CHECK(h->det && h->det->loaded(), LP_ERR_STATE, "detector not loaded");
  LP_CHECK(B >= 1 && B <= h->cfg.max_batch, LP_ERR_ARG, "batch %d outside 1..%d", B, h->cfg.max_batch);
  LP_HIP(hipSetDevice(h->cfg.device));
  const int S = h->cfg.det_input;
  const size_t bytes = (size_t)B * S * S * 3;
  h->ensure_src(bytes);
  LP_HIP(hipMemcpyAsync(h->d_src.p, bgr, bytes, hipMemcpyHostToDevice, h->stream));
  std::vector<ImgGeom> g(B);
  for (int i = 0; i < B; ++i) g[i] = make_geom(S, S, S, (long)i * S * S * 3);
  h->upload_geom(g);
  Profiler* prof = begin_profile(h);
  LP_HIP(hipMemsetAsync(h->d_cand_count.p, 0, (size_t)h->cfg.max_batch * 4, h->stream));
  enqueue_detect(h, h->d_src.as<uint8_t>(), g, B, 2.0f /* nothing passes: raw output only */, h->d_out0.as<float>(), prof);
  const size_t obytes = (size_t)B * (4 + h->det->num_classes()) * h->det->num_anchors() * 4;
  LP_HIP(hipMemcpyAsync(out0, h->d_out0.p, obytes, hipMemcpyDeviceToHost, h->stream));
  LP_HIP(hipStreamSynchronize(h->stream));
  if (prof) { prof->collect(0); prof->enabled = false; }
  LP_API_END
}

int lp_detect(lp_handle* h, const uint8_t* const* imgs, const int* hs, const int* ws, int B, float conf, float iou,
              lp_det* dets, int* counts) {
  LP_API_BEGIN
  LP_CHECK(h && imgs && hs && ws && dets && counts, LP_ERR_ARG, "null argument");
  LP_CHECK(h->det && h->det->loaded(), LP_ERR_STATE, "detector not loaded");
  LP_CHECK(B >= 1 && B <= h->cfg.max_batch, LP_ERR_ARG, "batch %d outside 1..%d", B, h->cfg.max_batch);
  LP_HIP(hipSetDevice(h->cfg.device));
  std::vector<ImgGeom> g = upload_images(h, imgs, hs, ws, B);
  h->upload_geom(g);
  Profiler* prof = begin_profile(h);
  lp_handle::GraphKey key{2, B, h->geom_ver, -1, h->d_src.p, h->d_dets.p, h->d_counts.p, conf, iou};
  run_or_capture(h, key, prof == nullptr, [&]() {
    enqueue_detect(h, h->d_src.as<uint8_t>(), g, B, conf, nullptr, prof);
    enqueue_nms(h, B, iou, -1, h->d_dets.as<lp_det>(), h->d_counts.as<int>(), false, prof);
  });
  LP_HIP(hipMemcpyAsync(dets, h->d_dets.p, (size_t)B * h->cfg.max_det * sizeof(lp_det), hipMemcpyDeviceToHost, h->stream));
  LP_HIP(hipMemcpyAsync(counts, h->d_counts.p, (size_t)B * 4, hipMemcpyDeviceToHost, h->stream));
  LP_HIP(hipStreamSynchronize(h->stream));
  if (prof) { prof->collect(0); prof->enabled = false; }
  LP_API_END
}

int lp_run_batch(lp_handle* h, const uint8_t* const* imgs, const int* hs, const int* ws, int B, float conf, float iou,
                 int min_area, lp_det* dets, int* counts, int* num_det, float* det_conf_avg, lp_timing* timing) {
  LP_API_BEGIN
  LP_CHECK(h && imgs && hs && ws && dets && counts, LP_ERR_ARG, "null argument");
  LP_CHECK(h->det && h->det->loaded(), LP_ERR_STATE, "detector not loaded");
  LP_CHECK(h->cls && h->cls->loaded(), LP_ERR_STATE, "classifier not loaded");
  LP_CHECK(B >= 1 && B <= h->cfg.max_batch, LP_ERR_ARG, "batch %d outside 1..%d", B, h->cfg.max_batch);
  LP_CHECK(min_area >= 0, LP_ERR_ARG, "min_area must be >= 0");
  LP_HIP(hipSetDevice(h->cfg.device));
  if (chunked_ok(h, hs, ws, B)) {
    run_batch_chunked(h, imgs, hs[0], ws[0], B, conf, iou, min_area, dets, counts, num_det, det_conf_avg, timing);
    return LP_OK;
  }
  std::vector<ImgGeom> g = upload_images(h, imgs, hs, ws, B);
  h->upload_geom(g);
  Profiler* prof = begin_profile(h);
  // three captured pieces with the stage-boundary events between them (PipelineMetrics wants detection, ROI extraction and
  // classification times separately, e2e.py:452-499)
  LP_HIP(hipEventRecord(h->ev[0], h->stream));
  lp_handle::GraphKey k1{3, B, h->geom_ver, min_area, h->d_src.p, h->d_dets.p, h->d_counts.p, conf, iou};
  run_or_capture(h, k1, prof == nullptr, [&]() {
    enqueue_detect(h, h->d_src.as<uint8_t>(), g, B, conf, nullptr, prof);
    enqueue_nms(h, B, iou, min_area, h->d_dets.as<lp_det>(), h->d_counts.as<int>(), true, prof);
  });
  LP_HIP(hipEventRecord(h->ev[2], h->stream));
  lp_handle::GraphKey k2{4, B, h->geom_ver, min_area, h->d_src.p, h->d_dets.p, h->d_counts.p, conf, iou};
  run_or_capture(h, k2, prof == nullptr, [&]() {
    enqueue_classify(h, h->d_src.as<uint8_t>(), B, h->d_dets.as<lp_det>(), nullptr, nullptr, nullptr, prof, 1);
  });
  LP_HIP(hipEventRecord(h->ev[1], h->stream));
  lp_handle::GraphKey k3{5, B, h->geom_ver, min_area, h->d_src.p, h->d_dets.p, h->d_counts.p, conf, iou};
  run_or_capture(h, k3, prof == nullptr, [&]() {
    enqueue_classify(h, h->d_src.as<uint8_t>(), B, h->d_dets.as<lp_det>(), nullptr, nullptr, nullptr, prof, 2);
  });
  LP_HIP(hipEventRecord(h->ev[3], h->stream));
  LP_HIP(hipMemcpyAsync(dets, h->d_dets.p, (size_t)B * h->cfg.max_det * sizeof(lp_det), hipMemcpyDeviceToHost, h->stream));
  std::vector<int> cnt(3 * B);
  LP_HIP(hipMemcpyAsync(cnt.data(), h->d_counts.p, (size_t)3 * B * 4, hipMemcpyDeviceToHost, h->stream));
  int R[2] = {0, 0};
  LP_HIP(hipMemcpyAsync(R, h->d_roi_total.p, 8, hipMemcpyDeviceToHost, h->stream));
  LP_HIP(hipStreamSynchronize(h->stream));
  for (int i = 0; i < B; ++i) {
    counts[i] = cnt[i];
    if (num_det) num_det[i] = cnt[B + i];
    if (det_conf_avg) memcpy(&det_conf_avg[i], &cnt[2 * B + i], 4);
  }
  h->last_roi_count = R[0];
  if (timing) {
    // the detector's decode + NMS are booked under detection like the reference's detect() (e2e.py:452-453); the ROI
    // rectangles come out of the NMS kernel, the crop + PIL resize is the device's share of the ROI loop (e2e.py:460-475)
    (void)hipEventElapsedTime(&timing->t_detection, h->ev[0], h->ev[2]);
    (void)hipEventElapsedTime(&timing->t_roi_extract, h->ev[2], h->ev[1]);
    (void)hipEventElapsedTime(&timing->t_classification, h->ev[1], h->ev[3]);
    (void)hipEventElapsedTime(&timing->t_total, h->ev[0], h->ev[3]);
  }
  if (prof) { prof->collect(R[0]); prof->enabled = false; }
  // every kept ROI must have been classified (the reference classifies all of them): a user-set max_rois that was too
  // small is an error, not a silent cls_class = -1
  LP_CHECK(R[1] <= h->max_rois, LP_ERR_STATE, "%d ROIs in this batch exceed max_rois = %d: %d detections were left unclassified", R[1],
           h->max_rois, R[1] - h->max_rois);
  LP_API_END
}

int lp_run_batch_device(lp_handle* h, const void* dev_imgs, int B, int H, int W, float conf, float iou, int min_area,
                        void* dev_dets, void* dev_counts) {
  LP_API_BEGIN
  LP_CHECK(h && dev_imgs && dev_dets && dev_counts, LP_ERR_ARG, "null argument");
  LP_CHECK(h->det && h->det->loaded(), LP_ERR_STATE, "detector not loaded");
  LP_CHECK(B >= 1 && B <= h->cfg.max_batch && H > 0 && W > 0, LP_ERR_ARG, "bad batch/shape");
  LP_HIP(hipSetDevice(h->cfg.device));
  std::vector<ImgGeom> g(B);
  for (int i = 0; i < B; ++i) g[i] = make_geom(H, W, h->cfg.det_input, (long)i * H * W * 3);
  h->upload_geom(g);
  Profiler* prof = begin_profile(h);
  const uint8_t* src = static_cast<const uint8_t*>(dev_imgs);
  const bool classify = h->cls && h->cls->loaded();
  lp_handle::GraphKey key{1, B, h->geom_ver, min_area, dev_imgs, dev_dets, dev_counts, conf, iou};
  run_or_capture(h, key, prof == nullptr, [&]() {
    enqueue_detect(h, src, g, B, conf, nullptr, prof);
    enqueue_nms(h, B, iou, classify ? min_area : -1, static_cast<lp_det*>(dev_dets), static_cast<int*>(dev_counts), classify, prof);
    if (classify) enqueue_classify(h, src, B, static_cast<lp_det*>(dev_dets), nullptr, nullptr, nullptr, prof);
  });
  if (prof) prof->enabled = false;  // records are collected by lp_profile_read after the caller synchronises
  LP_API_END
}

int lp_comm_unique_id(void* id_out) {
  LP_API_BEGIN
  LP_CHECK(id_out, LP_ERR_ARG, "null argument");
  NcclId id;
  LP_RCCL(rccl().GetUniqueId(&id));
  memcpy(id_out, &id, sizeof(id));
  LP_API_END
}

int lp_comm_init(lp_handle* h, const void* id, int rank, int world) {
  LP_API_BEGIN
  LP_CHECK(h && id && world >= 1 && rank >= 0 && rank < world, LP_ERR_ARG, "bad argument (rank %d of %d)", rank, world);
  LP_CHECK(!h->comm, LP_ERR_STATE, "the handle already has a communicator");
  LP_HIP(hipSetDevice(h->cfg.device));
  NcclId nid;
  memcpy(&nid, id, sizeof(nid));
  void* comm = nullptr;
  LP_RCCL(rccl().CommInitRank(&comm, world, nid, rank));
  h->comm = comm; h->comm_rank = rank; h->comm_world = world;
  LP_API_END
}

int lp_gather(lp_handle* h, const void* dev_send, size_t bytes, void* dev_recv, int root) {
  LP_API_BEGIN
  LP_CHECK(h && dev_send && bytes > 0, LP_ERR_ARG, "bad argument");
  LP_CHECK(h->comm, LP_ERR_STATE, "lp_comm_init has not been called on this handle");
  LP_CHECK(root >= 0 && root < h->comm_world && (h->comm_rank != root || dev_recv), LP_ERR_ARG, "bad root %d / receive buffer", root);
  LP_HIP(hipSetDevice(h->cfg.device));
  LP_RCCL(rccl().Gather(dev_send, dev_recv, bytes, 1 /* ncclUint8 */, root, h->comm, h->stream));
  LP_API_END
}

int lp_comm_destroy(lp_handle* h) {
  LP_API_BEGIN
  LP_CHECK(h, LP_ERR_ARG, "null argument");
  if (h->comm) {
    LP_HIP(hipSetDevice(h->cfg.device));
    LP_HIP(hipStreamSynchronize(h->stream));
    LP_RCCL(rccl().CommDestroy(h->comm));
    h->comm = nullptr; h->comm_world = 1; h->comm_rank = 0;
  }
  LP_API_END
}

int lp_roi_overflow(lp_handle* h, int* classified, int* kept) {
  LP_API_BEGIN
  LP_CHECK(h && classified && kept, LP_ERR_ARG, "null argument");
  LP_HIP(hipSetDevice(h->cfg.device));
  LP_HIP(hipStreamSynchronize(h->stream));
  int R[2] = {0, 0};
  LP_HIP(hipMemcpy(R, h->d_roi_total.p, sizeof(R), hipMemcpyDeviceToHost));
  *classified = R[0];
  *kept = R[1];
  LP_API_END
}

int lp_classify(lp_handle* h, const uint8_t* const* rois, const int* hs, const int* ws, int R, int* ids, float* probs) {
  LP_API_BEGIN
  LP_CHECK(h && ids && probs, LP_ERR_ARG, "null argument");
  LP_CHECK(h->cls && h->cls->loaded(), LP_ERR_STATE, "classifier not loaded");
  LP_CHECK(R >= 0 && R <= h->max_rois && R <= h->cfg.max_batch * h->cfg.max_det, LP_ERR_ARG,
           "%d ROIs exceed the capacity (%d)", R, std::min(h->max_rois, h->cfg.max_batch * h->cfg.max_det));
  if (R == 0) return LP_OK;
  LP_CHECK(rois && hs && ws, LP_ERR_ARG, "null argument");
  LP_HIP(hipSetDevice(h->cfg.device));
  // every crop is its own "image" whose single ROI is the whole crop
  std::vector<ImgGeom> g(R);
  std::vector<int> rects((size_t)R * 4, 0), img(R), slot(R, 0);
  size_t total = 0;
  for (int i = 0; i < R; ++i) {
    LP_CHECK(rois[i] && hs[i] > 0 && ws[i] > 0 && hs[i] <= 4096 && ws[i] <= 4096, LP_ERR_ARG, "ROI %d has a bad size", i);
    memset(&g[i], 0, sizeof(ImgGeom));
    g[i].src_off = (long)total; g[i].h = hs[i]; g[i].w = ws[i];
    total += ((size_t)hs[i] * ws[i] * 3 + 15) & ~(size_t)15;
    img[i] = i;
  }
  LP_CHECK(R <= (int)(h->d_geom.bytes / sizeof(ImgGeom)), LP_ERR_ARG, "too many ROIs");
  h->ensure_src(total);
  for (int i = 0; i < R; ++i)
    LP_HIP(hipMemcpyAsync(h->d_src.as<uint8_t>() + g[i].src_off, rois[i], (size_t)hs[i] * ws[i] * 3, hipMemcpyHostToDevice, h->stream));
  h->geom_cache.clear();
  LP_HIP(hipMemcpyAsync(h->d_geom.p, g.data(), (size_t)R * sizeof(ImgGeom), hipMemcpyHostToDevice, h->stream));
  DevBuf d_rects_tmp;  // [R][1][4]: one whole-crop rectangle per "image"
  d_rects_tmp.alloc((size_t)R * 16);
  for (int i = 0; i < R; ++i) {
    int* rc = &rects[(size_t)i * 4];
    rc[0] = 0; rc[1] = 0; rc[2] = ws[i]; rc[3] = hs[i];
  }
  LP_HIP(hipMemcpyAsync(d_rects_tmp.p, rects.data(), rects.size() * 4, hipMemcpyHostToDevice, h->stream));
  LP_HIP(hipMemcpyAsync(h->d_roi_img.p, img.data(), (size_t)R * 4, hipMemcpyHostToDevice, h->stream));
  LP_HIP(hipMemcpyAsync(h->d_roi_slot.p, slot.data(), (size_t)R * 4, hipMemcpyHostToDevice, h->stream));
  LP_HIP(hipMemcpyAsync(h->d_roi_total.p, &R, 4, hipMemcpyHostToDevice, h->stream));
  Profiler* prof = begin_profile(h);
  RoiTable tab = h->roi_table();
  RoiResizeArgs r;
  r.src = h->d_src.as<uint8_t>(); r.geom = h->d_geom.as<ImgGeom>(); r.rects = d_rects_tmp.as<int>(); r.tab = tab;
  r.out = h->d_roi_rgb.as<uint8_t>(); r.max_det = 1; r.S = h->cfg.cls_input; r.linear = h->cfg.numerics;
  launch_roi_resize(r, R, h->stream);
  ClsPost post;
  post.probs = h->d_probs.as<float>(); post.ids = h->d_ids.as<int>();
  h->cls->forward(h->d_roi_rgb.as<uint8_t>(), tab.total, h->stream, prof, &post);
  if (!h->cls->fused_head())
    launch_softmax_argmax(h->cls->logits(), h->cls->logits_pitch(), h->cls->num_classes(), h->d_probs.as<float>(), h->d_ids.as<int>(),
                          nullptr, nullptr, h->cfg.max_det, nullptr, tab.total, h->max_rois, h->stream);
  LP_HIP(hipMemcpyAsync(ids, h->d_ids.p, (size_t)R * 4, hipMemcpyDeviceToHost, h->stream));
  LP_HIP(hipMemcpyAsync(probs, h->d_probs.p, (size_t)R * h->cls->num_classes() * 4, hipMemcpyDeviceToHost, h->stream));
  LP_HIP(hipStreamSynchronize(h->stream));
  if (prof) { prof->collect(R); prof->enabled = false; }
  LP_API_END
}

int lp_debug_blob(lp_handle* h, const char* blob, float* out, int64_t cap, int* C, int* H, int* W) {
  LP_API_BEGIN
  LP_CHECK(h && blob && C && H && W, LP_ERR_ARG, "null argument");
  LP_CHECK(h->det && h->det->loaded(), LP_ERR_STATE, "detector not loaded");
  LP_HIP(hipSetDevice(h->cfg.device));
  LP_HIP(hipStreamSynchronize(h->stream));
  std::vector<float> v;
  h->det->fetch_blob(blob, 1, v, *C, *H, *W);
  if (out) {
    LP_CHECK((int64_t)v.size() <= cap, LP_ERR_ARG, "blob needs %zu floats, buffer has %lld", v.size(), (long long)cap);
    memcpy(out, v.data(), v.size() * 4);
  }
  LP_API_END
}

int lp_test_conv(lp_handle* h, int impl, const float* x, int N, int Cin, int H, int W, const float* w, const float* bias,
                 int Cout, int k, int stride, int act, const float* res, float* y) {
  LP_API_BEGIN
  LP_CHECK(h && x && w && y, LP_ERR_ARG, "null argument");
  LP_CHECK(Cin % 8 == 0 && Cout % 8 == 0, LP_ERR_ARG, "test conv needs channel counts that are multiples of 8");
  LP_HIP(hipSetDevice(h->cfg.device));
  const int prec = h->cfg.precision;
  const size_t es = prec == LP_FP16 ? 2 : 4;
  const int pad = k / 2, Ho = (H + 2 * pad - k) / stride + 1, Wo = (W + 2 * pad - k) / stride + 1;
  const int taps = k * k;
  std::vector<float> wp((size_t)Cout * taps * Cin), bp(Cout, 0.f);
  for (int o = 0; o < Cout; ++o) {
    for (int i = 0; i < Cin; ++i)
      for (int t = 0; t < taps; ++t) wp[((size_t)o * taps + t) * Cin + i] = w[((size_t)o * Cin + i) * taps + t];
    if (bias) bp[o] = bias[o];
  }
  ConvLayer L;
  L.build(prec, impl, k, stride, Cin, Cout, act, wp, bp, Ho, Wo, N);
  auto to_dev = [&](const float* src, int C, int HH, int WW, DevBuf& d) {
    const size_t npix = (size_t)N * HH * WW;
    std::vector<uint8_t> buf(npix * C * es);
    for (size_t p = 0; p < npix; ++p) {
      const size_t b = p / ((size_t)HH * WW), yx = p % ((size_t)HH * WW);
      for (int c = 0; c < C; ++c) {
        const float v = src[(b * C + c) * HH * WW + yx];
        if (prec == LP_FP16) { uint16_t hv = f32_to_f16(v); memcpy(&buf[(p * C + c) * 2], &hv, 2); }
        else memcpy(&buf[(p * C + c) * 4], &v, 4);
      }
    }
    d.alloc(buf.size());
    LP_HIP(hipMemcpy(d.p, buf.data(), buf.size(), hipMemcpyHostToDevice));
  };
  DevBuf dx, dy, dr;
  to_dev(x, Cin, H, W, dx);
  dy.alloc((size_t)N * Ho * Wo * Cout * es);
  ConvIO io;
  io.N = N;
  io.in = View{dx.p, Cin, Cin, H, W};
  io.out = View{dy.p, Cout, Cout, Ho, Wo};
  if (res) { to_dev(res, Cout, Ho, Wo, dr); io.res = View{dr.p, Cout, Cout, Ho, Wo}; }
  // diagnostic: LITEPI_STAMPS=<file> dumps 16 clock stamps per workgroup of a (warm) second launch
  const char* stamp_path = getenv("LITEPI_STAMPS");
  L.launch(io, h->stream);
  LP_HIP(hipStreamSynchronize(h->stream));
  if (stamp_path && *stamp_path) {
    const size_t nst = (size_t)1 << 22;
    DevBuf ds;
    ds.alloc(nst * 8);
    io.stamps = ds.as<unsigned long long>();
    L.launch(io, h->stream);
    LP_HIP(hipStreamSynchronize(h->stream));
    std::vector<unsigned long long> hs(nst);
    LP_HIP(hipMemcpy(hs.data(), ds.p, nst * 8, hipMemcpyDeviceToHost));
    size_t used = nst;
    while (used > 16 && hs[used - 16] == 0 && hs[used - 4] == 0) used -= 16;
    FILE* f = fopen(stamp_path, "wb");
    if (f) { fwrite(hs.data(), 8, used, f); fclose(f); }
    io.stamps = nullptr;
  }
  std::vector<uint8_t> raw((size_t)N * Ho * Wo * Cout * es);
  LP_HIP(hipMemcpy(raw.data(), dy.p, raw.size(), hipMemcpyDeviceToHost));
  const size_t npix = (size_t)N * Ho * Wo;
  for (size_t p = 0; p < npix; ++p) {
    const size_t b = p / ((size_t)Ho * Wo), yx = p % ((size_t)Ho * Wo);
    for (int c = 0; c < Cout; ++c) {
      float f;
      if (prec == LP_FP16) { uint16_t hv; memcpy(&hv, &raw[(p * Cout + c) * 2], 2); f = f16_to_f32(hv); }
      else memcpy(&f, &raw[(p * Cout + c) * 4], 4);
      y[(b * Cout + c) * Ho * Wo + yx] = f;
    }
  }
  LP_API_END
}

int lp_test_postprocess(lp_handle* h, const float* out0, int nc, int A, int orig_h, int orig_w, float ratio, float pad_w,
                        float pad_h, float conf, float iou, int min_area, int max_det, lp_det* dets, int* rects, int* count,
                        int* num_det) {
  LP_API_BEGIN
  LP_CHECK(h && out0 && dets && count && nc >= 1 && A >= 1 && A <= 16384, LP_ERR_ARG, "bad argument");
  LP_HIP(hipSetDevice(h->cfg.device));
  ImgGeom g;
  memset(&g, 0, sizeof(g));
  g.h = orig_h; g.w = orig_w; g.ratio = ratio; g.pad_w = pad_w; g.pad_h = pad_h;
  DevBuf d_out0, d_geom, d_cand, d_sorted, d_cnt, d_dets, d_counts, d_rects;
  d_out0.alloc((size_t)(4 + nc) * A * 4);
  LP_HIP(hipMemcpy(d_out0.p, out0, (size_t)(4 + nc) * A * 4, hipMemcpyHostToDevice));
  d_geom.alloc(sizeof(g));
  LP_HIP(hipMemcpy(d_geom.p, &g, sizeof(g), hipMemcpyHostToDevice));
  d_cand.alloc((size_t)A * sizeof(Cand)); d_sorted.alloc((size_t)A * sizeof(Cand)); d_cnt.alloc(16);
  if (max_det <= 0 || max_det > A) max_det = A;  // the reference keeps every survivor
  d_dets.alloc((size_t)max_det * sizeof(lp_det)); d_counts.alloc(16); d_rects.alloc((size_t)max_det * 16);
  launch_filter_out0(d_out0.as<float>(), nc, A, d_geom.as<ImgGeom>(), d_cand.as<Cand>(), d_cnt.as<int>(), conf, 1, h->stream);
  NmsArgs a;
  memset(&a, 0, sizeof(a));
  a.cand = d_cand.as<Cand>(); a.cand_count = d_cnt.as<int>(); a.sorted = d_sorted.as<Cand>(); a.dets = d_dets.as<lp_det>();
  a.counts = d_counts.as<int>(); a.rects = d_rects.as<int>(); a.geom = d_geom.as<ImgGeom>(); a.A = A; a.max_det = max_det; a.nc = nc;
  a.iou = iou; a.min_area = min_area; a.roi_rule = h->cfg.numerics;
  launch_nms(a, 1, h->stream);
  LP_HIP(hipStreamSynchronize(h->stream));
  int cnt[3];
  LP_HIP(hipMemcpy(cnt, d_counts.p, 12, hipMemcpyDeviceToHost));
  *count = cnt[0];
  if (num_det) *num_det = cnt[1];
  LP_HIP(hipMemcpy(dets, d_dets.p, (size_t)(*count) * sizeof(lp_det), hipMemcpyDeviceToHost));
  if (rects) LP_HIP(hipMemcpy(rects, d_rects.p, (size_t)(*count) * 16, hipMemcpyDeviceToHost));
  LP_API_END
}

int lp_test_nms_boxes(lp_handle* h, const float* boxes, const float* scores, const int* classes, int n, int orig_h, int orig_w,
                      float iou, int min_area, int max_det, lp_det* dets, int* rects, int* count, int* num_det) {
  LP_API_BEGIN
  LP_CHECK(h && boxes && scores && dets && count && n >= 0 && n <= 16384, LP_ERR_ARG, "bad argument");
  LP_HIP(hipSetDevice(h->cfg.device));
  const int A = n > 0 ? n : 1;
  ImgGeom g;
  memset(&g, 0, sizeof(g));
  g.h = orig_h; g.w = orig_w; g.ratio = 1.f;
  std::vector<Cand> cand(A);
  int nc = 1;
  for (int i = 0; i < n; ++i) {
    Cand c;
    c.x1 = boxes[4 * i]; c.y1 = boxes[4 * i + 1]; c.x2 = boxes[4 * i + 2]; c.y2 = boxes[4 * i + 3];
    c.score = scores[i]; c.cls = classes ? classes[i] : 0; c.anchor = i; c.pad = 0;
    nc = std::max(nc, c.cls + 1);
    cand[i] = c;
  }
  DevBuf d_geom, d_cand, d_sorted, d_cnt, d_dets, d_counts, d_rects;
  d_geom.alloc(sizeof(g));
  LP_HIP(hipMemcpy(d_geom.p, &g, sizeof(g), hipMemcpyHostToDevice));
  d_cand.alloc((size_t)A * sizeof(Cand)); d_sorted.alloc((size_t)A * sizeof(Cand)); d_cnt.alloc(16);
  LP_HIP(hipMemcpy(d_cand.p, cand.data(), (size_t)A * sizeof(Cand), hipMemcpyHostToDevice));
  LP_HIP(hipMemcpy(d_cnt.p, &n, 4, hipMemcpyHostToDevice));
  if (max_det <= 0 || max_det > A) max_det = A;
  d_dets.alloc((size_t)max_det * sizeof(lp_det)); d_counts.alloc(16); d_rects.alloc((size_t)max_det * 16);
  NmsArgs a;
  memset(&a, 0, sizeof(a));
  a.cand = d_cand.as<Cand>(); a.cand_count = d_cnt.as<int>(); a.sorted = d_sorted.as<Cand>(); a.dets = d_dets.as<lp_det>();
  a.counts = d_counts.as<int>(); a.rects = d_rects.as<int>(); a.geom = d_geom.as<ImgGeom>(); a.A = A; a.max_det = max_det; a.nc = nc;
  a.iou = iou; a.min_area = min_area; a.roi_rule = h->cfg.numerics;
  launch_nms(a, 1, h->stream);
  LP_HIP(hipStreamSynchronize(h->stream));
  int cnt[3];
  LP_HIP(hipMemcpy(cnt, d_counts.p, 12, hipMemcpyDeviceToHost));
  *count = cnt[0];
  if (num_det) *num_det = cnt[1];
  LP_HIP(hipMemcpy(dets, d_dets.p, (size_t)(*count) * sizeof(lp_det), hipMemcpyDeviceToHost));
  if (rects) LP_HIP(hipMemcpy(rects, d_rects.p, (size_t)(*count) * 16, hipMemcpyDeviceToHost));
  LP_API_END
}

int lp_test_roi_resize(lp_handle* h, const uint8_t* const* rois, const int* hs, const int* ws, int R, uint8_t* out_rgb) {
  LP_API_BEGIN
  LP_CHECK(h && rois && hs && ws && out_rgb && R >= 1, LP_ERR_ARG, "bad argument");
  LP_HIP(hipSetDevice(h->cfg.device));
  const int S = h->cfg.cls_input;
  std::vector<ImgGeom> g(R);
  std::vector<int> rects((size_t)R * 4), img(R), slot(R, 0);
  size_t total = 0;
  for (int i = 0; i < R; ++i) {
    LP_CHECK(hs[i] > 0 && ws[i] > 0 && hs[i] <= 4096 && ws[i] <= 4096, LP_ERR_ARG, "ROI %d has a bad size", i);
    memset(&g[i], 0, sizeof(ImgGeom));
    g[i].src_off = (long)total; g[i].h = hs[i]; g[i].w = ws[i];
    total += ((size_t)hs[i] * ws[i] * 3 + 15) & ~(size_t)15;
    img[i] = i;
    rects[i * 4 + 0] = 0; rects[i * 4 + 1] = 0; rects[i * 4 + 2] = ws[i]; rects[i * 4 + 3] = hs[i];
  }
  DevBuf d_src, d_geom, d_rects, d_img, d_slot, d_total, d_base, d_out;
  d_src.alloc(total);
  for (int i = 0; i < R; ++i) LP_HIP(hipMemcpy(d_src.as<uint8_t>() + g[i].src_off, rois[i], (size_t)hs[i] * ws[i] * 3, hipMemcpyHostToDevice));
  d_geom.alloc((size_t)R * sizeof(ImgGeom));
  LP_HIP(hipMemcpy(d_geom.p, g.data(), (size_t)R * sizeof(ImgGeom), hipMemcpyHostToDevice));
  d_rects.alloc((size_t)R * 16); LP_HIP(hipMemcpy(d_rects.p, rects.data(), (size_t)R * 16, hipMemcpyHostToDevice));
  d_img.alloc((size_t)R * 4); LP_HIP(hipMemcpy(d_img.p, img.data(), (size_t)R * 4, hipMemcpyHostToDevice));
  d_slot.alloc((size_t)R * 4);
  d_total.alloc(16); LP_HIP(hipMemcpy(d_total.p, &R, 4, hipMemcpyHostToDevice));
  d_base.alloc(16);
  d_out.alloc((size_t)R * S * S * 3);
  RoiResizeArgs r;
  r.src = d_src.as<uint8_t>(); r.geom = d_geom.as<ImgGeom>(); r.rects = d_rects.as<int>();
  r.tab.base = d_base.as<int>(); r.tab.total = d_total.as<int>(); r.tab.img = d_img.as<int>(); r.tab.slot = d_slot.as<int>();
  r.out = d_out.as<uint8_t>(); r.max_det = 1; r.S = S; r.linear = h->cfg.numerics;
  launch_roi_resize(r, R, h->stream);
  LP_HIP(hipStreamSynchronize(h->stream));
  LP_HIP(hipMemcpy(out_rgb, d_out.p, (size_t)R * S * S * 3, hipMemcpyDeviceToHost));
  LP_API_END
}

int lp_test_letterbox(lp_handle* h, const uint8_t* img, int H, int W, uint8_t* out, float* ratio, float* pad_w, float* pad_h) {
  LP_API_BEGIN
  LP_CHECK(h && img && out && H > 0 && W > 0, LP_ERR_ARG, "bad argument");
  LP_HIP(hipSetDevice(h->cfg.device));
  const int S = h->cfg.det_input;
  ImgGeom g = make_geom(H, W, S, 0);
  DevBuf d_src, d_geom, d_out;
  d_src.alloc((size_t)H * W * 3);
  LP_HIP(hipMemcpy(d_src.p, img, (size_t)H * W * 3, hipMemcpyHostToDevice));
  d_geom.alloc(sizeof(g));
  LP_HIP(hipMemcpy(d_geom.p, &g, sizeof(g), hipMemcpyHostToDevice));
  d_out.alloc((size_t)S * S * 3);
  launch_letterbox(d_src.as<uint8_t>(), d_geom.as<ImgGeom>(), d_out.as<uint8_t>(), 1, S, h->stream, &g);
  LP_HIP(hipStreamSynchronize(h->stream));
  LP_HIP(hipMemcpy(out, d_out.p, (size_t)S * S * 3, hipMemcpyDeviceToHost));
  if (ratio) *ratio = g.ratio;
  if (pad_w) *pad_w = g.pad_w;
  if (pad_h) *pad_h = g.pad_h;
  LP_API_END
}

}  // extern "C"
