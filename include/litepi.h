/*
 * litepi.h -- C-ABI of liblitepi_hip.so: the MI355X (gfx950) implementation of
 * YOLO-LitePi's two-stage inference hot path.
 *
 * The reference (vinhisreal/YOLO-LitePi) has no FFI of its own: its seam is three
 * Python classes in src/tt100k/pipeline/e2e.py whose arithmetic runs inside NCNN /
 * ONNX Runtime / torch.  Each entry point below replaces the engine call(s) and the
 * NumPy glue cited next to it; litepi/backend.py binds them with ctypes and mirrors
 * the reference classes on top (INTEGRATION.md shows the binding).
 *
 * Conventions: every function returns 0 on success or a negative lp_status;
 * lp_last_error() returns the text of the calling thread's last failure.  Plain C
 * types only.  "host" pointers are caller-owned CPU memory, "dev" pointers are
 * device (HBM) addresses on the handle's GPU.  A handle is NOT thread-safe (the
 * reference pipeline object is not re-entrant either: e2e.py:305 creates one
 * extractor per call); use one handle per GPU.
 */
#ifndef LITEPI_H
#define LITEPI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct lp_handle lp_handle;

enum lp_status {
  LP_OK = 0,
  LP_ERR_ARG = -1,         /* bad argument / shape */
  LP_ERR_IO = -2,          /* cannot read model file (-> RuntimeError, e2e.py:213-216) */
  LP_ERR_GRAPH = -3,       /* graph uses something outside the YOLOv8-family op set */
  LP_ERR_HIP = -4,         /* HIP runtime failure (-> empty result, e2e.py:309-310) */
  LP_ERR_STATE = -5,       /* model not loaded / capacity exceeded */
  LP_ERR_NODEVICE = -6     /* no usable gfx950 device: the product never falls back to CPU */
};

enum lp_precision { LP_FP32 = 0, LP_FP16 = 1 };
enum lp_numerics { LP_NUMERICS_E2E = 0, LP_NUMERICS_E2E_OPTIMIZE = 1 };
enum lp_cls_arch { LP_CLS_SHUFFLENETV2 = 0, LP_CLS_RESNET18 = 1, LP_CLS_MOBILENETV2 = 2, LP_CLS_EFFICIENTNET_B0 = 3 };   /* e2e.py:320-333 */

typedef struct lp_config {
  int device;        /* HIP device ordinal */
  int precision;     /* lp_precision: storage/MFMA input type; accumulation is always fp32 */
  int max_batch;     /* images per call (capacity of every activation buffer) */
  int max_det;       /* detections kept per image after NMS; the reference keeps all (e2e.py:280-296): when more
                        survive, the max_det highest scores over all classes stay, emitted class-ascending then
                        score-descending like the reference; set it to the anchor count for keep-all semantics */
  int num_classes;   /* classifier classes (e2e.py:353 default 58) */
  int det_input;     /* detector input size (e2e.py:1040 --det_input_size, 640) */
  int cls_input;     /* classifier input size (e2e.py:1041 --cls_input_size, 64) */
  int max_rois;      /* ROIs classified per call; 0 = max_batch * max_det = every kept box, as the reference
                        (e2e.py:493-497).  A smaller value that a batch exceeds makes lp_run_batch fail with
                        LP_ERR_STATE instead of leaving detections unclassified silently */
  int conv_impl;     /* 0 = MFMA kernels (product), 1 = naive direct kernels (GPU debug aid) */
  int numerics;      /* which of the reference's two pipelines the ROI stage follows:
                        0 = HybridPipeline (src/tt100k/pipeline/e2e.py:460-485, 366-370): clip x1<=w-1, x2>=x1+1 ..., PIL
                            antialiased bilinear resize;
                        1 = HybridPipelineOptimized (e2e_optimize.py:480-497, 386-390): clip to [0,w] x [0,h], drop empty
                            rectangles, cv2.resize INTER_LINEAR (no antialias).  PARITY UNPINNED (cv2 absent, no fixtures) */
  int cls_arch;      /* lp_cls_arch: classifier architecture (e2e.py:320-333 --clf_arch) */
  int reserved[5];
} lp_config;

/* One detection, 32 bytes.  Mirrors one result dict of HybridPipeline.run
 * (e2e.py:519-529): bbox (float box in original-image pixels; the dict stores its
 * int truncation), det_conf, det_class, cls_class, cls_conf. */
typedef struct lp_det {
  float x1, y1, x2, y2;
  float det_conf;
  int32_t det_class;
  int32_t cls_class;   /* -1 when the ROI was not classified (e2e.py:525) */
  float cls_conf;
} lp_det;

/* Per-call stage timings in ms from HIP events on the handle's stream
 * (PipelineMetrics.t_detection / t_roi_extract / t_classification / t_total,
 * e2e.py:34-62). */
typedef struct lp_timing {
  float t_detection, t_roi_extract, t_classification, t_total;
} lp_timing;

/* ABI version of this header: bumped on every incompatible change of a signature, of lp_config's meaning or of a buffer
 * contract.  300 (round 3) vs 100: lp_run_batch takes det_conf_avg, lp_test_postprocess changed, lp_config::numerics /
 * cls_arch took two reserved words, and lp_run_batch_device's dev_counts holds 3*B int32 (was 2*B: a caller that still
 * allocates 2*B is overrun).  litepi/_ffi.py refuses a library whose lp_version() differs. */
#define LP_ABI_VERSION 310

const char* lp_last_error(void);
int lp_version(void);

/* ---- lifetime ------------------------------------------------------------------ */
void lp_default_config(lp_config* cfg);
/* replaces NCNNDetector.__init__ / PyTorchClassifier.__init__ device setup (e2e.py:198-220,353-375) */
int lp_create(const lp_config* cfg, lp_handle** out);
void lp_destroy(lp_handle* h);

/* ---- model loading ------------------------------------------------------------- */
/* replaces ncnn.Net.load_param/load_model (e2e.py:213-216): parses the NCNN text graph +
 * weight blob, checks it is a YOLOv8-family detector (Conv/Swish/C2f/SPPF/Detect+DFL),
 * BN already folded, and builds the device execution plan. */
int lp_load_detector_ncnn(lp_handle* h, const char* param_path, const char* bin_path);
/* replaces build_classifier('shufflenetv2') + load_state_dict (e2e.py:331-340): takes the
 * torchvision shufflenet_v2_x1_0 state_dict as n named fp32 host tensors (PyTorch is used by
 * the caller only to read the .pth); BN is folded here. */
int lp_load_classifier_tensors(lp_handle* h, int n, const char* const* names,
                               const float* const* data, const int64_t* const* shapes,
                               const int* ndims);

/* ---- detector parity hook ------------------------------------------------------- */
/* replaces preprocess + ex.extract("out0") (e2e.py:222-238,305-307) for B images that are
 * already det_input x det_input: host uint8 BGR [B,S,S,3] -> host fp32 out0 [B,4+nc,A]. */
int lp_detect_raw(lp_handle* h, const uint8_t* bgr, int B, float* out0);

/* ---- detector + NMS ------------------------------------------------------------- */
/* replaces NCNNDetector.detect (e2e.py:298-316: letterbox, forward, postprocess, per-class
 * NMS) for B host images of individual sizes.  dets [B*max_det] receives x1..det_class
 * (cls_* untouched); counts [B] the boxes kept per image, ordered class-ascending then
 * score-descending (e2e.py:280-296). */
int lp_detect(lp_handle* h, const uint8_t* const* imgs, const int* heights, const int* widths,
              int B, float conf, float iou, lp_det* dets, int* counts);

/* ---- full pipeline --------------------------------------------------------------- */
/* replaces HybridPipeline.run (e2e.py:443-531) for a batch: detect -> ROI clip/area filter
 * (e2e.py:465-473) -> PIL-bilinear 64x64 + normalize (e2e.py:385-389) -> ShuffleNetV2 ->
 * softmax/argmax.  dets [B*max_det]: only boxes that survive the min_area filter;
 * counts [B] their number; num_det [B] (may be NULL) the pre-filter count
 * (PipelineMetrics.num_detections, e2e.py:454); det_conf_avg [B] (may be NULL) the mean
 * detector score over the pre-filter boxes (PipelineMetrics.det_confidence_avg, e2e.py:456-457). */
int lp_run_batch(lp_handle* h, const uint8_t* const* imgs, const int* heights, const int* widths,
                 int B, float conf, float iou, int min_area,
                 lp_det* dets, int* counts, int* num_det, float* det_conf_avg, lp_timing* timing);

/* Same pipeline on B equally sized images already resident in HBM (dev_imgs: uint8 BGR
 * [B,H,W,3]); results stay on the device: dev_dets [B*max_det] lp_det, dev_counts [3*B] int32
 * (kept counts, then pre-filter counts, then the float bits of the mean pre-filter score).  Asynchronous on the handle's stream; this is what
 * bench.py times and what the multi-GPU gather consumes. */
int lp_run_batch_device(lp_handle* h, const void* dev_imgs, int B, int H, int W,
                        float conf, float iou, int min_area, void* dev_dets, void* dev_counts);
/* After lp_run_batch_device (synchronises the handle's stream): *kept = ROIs that survived the area filter in the last call,
 * *classified = min(kept, max_rois).  kept > classified means a user-set lp_config::max_rois was too small and
 * kept - classified detections still carry cls_class = -1 (lp_run_batch turns the same condition into LP_ERR_STATE;
 * the asynchronous device path cannot).  With max_rois = 0 (default: max_batch * max_det) it cannot happen. */
int lp_roi_overflow(lp_handle* h, int* classified, int* kept);

/* ---- multi-GPU: the one collective of the path, owned by the library (ABI 310) ------ */
/* The reference is a single process (e2e.py:1096-1134 loops over the images); sharding the images over the GPUs of a node
 * adds exactly one exchange step: every rank's fixed-capacity payload (the lp_det records + the three count words per image
 * that lp_run_batch_device wrote) goes to one root rank ("one ncclGather to rank 0", rccl.h:745).
 * RCCL is loaded lazily (dlopen of librccl.so.1 at the first of these calls): a single-GPU host without RCCL still loads
 * liblitepi_hip.so, and these four calls then fail with LP_ERR_STATE.
 *   lp_comm_unique_id : rank 0 draws the 128-byte ncclUniqueId; the launcher hands it to every rank (any out-of-band
 *                       channel: a file, an environment variable, a torch.distributed / MPI broadcast)
 *   lp_comm_init      : ncclCommInitRank on the handle's device (collective over the ranks of the communicator)
 *   lp_gather         : ncclGather of `bytes` device bytes per rank on the handle's OWN stream, i.e. ordered behind the
 *                       lp_run_batch_device that wrote them with no event; dev_recv (root only, else may be NULL) receives
 *                       world * bytes in rank order.  Asynchronous like lp_run_batch_device.
 *   lp_comm_destroy   : ncclCommDestroy (also done by lp_destroy) */
#define LP_COMM_ID_BYTES 128
int lp_comm_unique_id(void* id_out);
int lp_comm_init(lp_handle* h, const void* id, int rank, int world);
int lp_gather(lp_handle* h, const void* dev_send, size_t bytes, void* dev_recv, int root);
int lp_comm_destroy(lp_handle* h);

/* ---- classifier alone ------------------------------------------------------------ */
/* replaces PyTorchClassifier.predict_batch (e2e.py:378-396) for R host BGR crops of
 * individual sizes: ids [R], probs [R*num_classes] (softmax). */
int lp_classify(lp_handle* h, const uint8_t* const* rois, const int* heights, const int* widths,
                int R, int* ids, float* probs);

/* ---- streams / profiling --------------------------------------------------------- */
int lp_set_stream(lp_handle* h, void* hip_stream);   /* NULL = the handle's own stream */
int lp_synchronize(lp_handle* h);
/* Per-launch device timing of the NEXT pipeline call (hipEvents around every kernel on the
 * launch stream).  After that call returns, lp_profile_read gives up to cap entries. */
typedef struct lp_kernel_time {
  char name[48];     /* kernel family, e.g. "conv3x3_mfma_f16" */
  char layer[32];    /* graph layer, e.g. "conv_48" */
  float ms;
  double flops;      /* algorithmic FLOPs of this launch (2*MAC), 0 for byte movers */
  double bytes;      /* algorithmic HBM bytes of this launch (inputs read once + outputs written once) */
} lp_kernel_time;
int lp_profile_next(lp_handle* h, int enable);
int lp_profile_read(lp_handle* h, lp_kernel_time* out, int cap, int* n);

/* ---- introspection (tests) -------------------------------------------------------- */
int lp_detector_info(lp_handle* h, int* num_anchors, int* num_det_classes, int* reg_max,
                     double* conv_macs_per_image);
/* Copy an intermediate detector blob of the last lp_detect_raw call to host as fp32
 * logical [B,C,H,W] (NCNN blob names, e.g. "44" = P3).  Test/bisect aid. */
int lp_debug_blob(lp_handle* h, const char* blob, float* out, int64_t cap, int* C, int* H, int* W);
/* Run one convolution through the chosen kernel family on host data (tests):
 * x fp32 [N,Cin,H,W], w fp32 [Cout,Cin,k,k], bias [Cout] or NULL, res [N,Cout,Ho,Wo] or NULL. */
int lp_test_conv(lp_handle* h, int impl, const float* x, int N, int Cin, int H, int W,
                 const float* w, const float* bias, int Cout, int k, int stride, int act,
                 const float* res, float* y);
/* Decode+NMS(+ROI filter) on a host out0 tensor (tests of the post-processing kernels in isolation):
 * out0 fp32 [4+nc, A], geometry of the original image -> dets/count as lp_detect.  min_area < 0: no ROI
 * filter; >= 0: the ROI clip + area filter of e2e.py:465-473, rects [count*4] (may be NULL) receives the int
 * crop rectangles and num_det (may be NULL) the pre-filter count.  max_det <= 0: keep every survivor. */
int lp_test_postprocess(lp_handle* h, const float* out0, int nc, int A, int orig_h, int orig_w,
                        float ratio, float pad_w, float pad_h, float conf, float iou, int min_area, int max_det,
                        lp_det* dets, int* rects, int* count, int* num_det);
/* NMS + ROI clip/area filter on host boxes (xyxy, original-image pixels) given directly, bypassing the decode filter:
 * replays the reference's HybridPipeline.run ROI fixtures (e2e.py:460-485) through the device kernel. */
int lp_test_nms_boxes(lp_handle* h, const float* boxes, const float* scores, const int* classes, int n,
                      int orig_h, int orig_w, float iou, int min_area, int max_det,
                      lp_det* dets, int* rects, int* count, int* num_det);
/* PIL-exact ROI resize alone: host BGR crops -> uint8 RGB [R,S,S,3]. */
int lp_test_roi_resize(lp_handle* h, const uint8_t* const* rois, const int* heights,
                       const int* widths, int R, uint8_t* out_rgb);
/* cv2-style letterbox alone: host BGR image -> uint8 BGR [S,S,3] + ratio/pad. */
int lp_test_letterbox(lp_handle* h, const uint8_t* img, int H, int W, uint8_t* out,
                      float* ratio, float* pad_w, float* pad_h);

#ifdef __cplusplus
}
#endif
#endif /* LITEPI_H */
