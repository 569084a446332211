/* mtgv.h - C ABI of libmtgv.so: the MI355X-native recognition hot path of mtg-vision
 * (detect -> crop -> embed -> cosine top-k over the card bank).
 *
 * Every entry point is what a binding for that path would call.  The reference
 * (nmichlo/mtg-vision, all paths relative to its root) is pure Python; the
 * interface each group replaces is cited.  INTEGRATION.md shows the ctypes stubs.
 *
 * Conventions
 *   - return value: 0 = ok, 1 = invalid argument (reference: AssertionError),
 *     2 = unknown key/name (KeyError), 3 = runtime failure (RuntimeError).
 *     mtgv_last_error() returns the message of the calling thread's last failure.
 *   - pointers named *_dev are device (HIP) pointers owned by the caller
 *     (torch tensors on cuda:<i>); *_host are host pointers.  The library owns
 *     only weights / bank / workspace inside its opaque handles and allocates
 *     nothing on the hot path after the first call at a given batch size.
 *   - `stream` is a hipStream_t passed as void* (0 = default stream).  A handle is
 *     bound to the HIP device current at creation and is not thread-safe; calls on
 *     one handle must be serialised by the caller (the reference is single-threaded
 *     per process: mtgvision/server.py:29-35, :280).
 *   - activations are float32; images are NHWC unless stated.
 */
#ifndef MTGV_H
#define MTGV_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#if defined(__GNUC__)
#define MTGV_API __attribute__((visibility("default")))
#else
#define MTGV_API
#endif

MTGV_API const char* mtgv_last_error(void);
MTGV_API int mtgv_version(void);
/* number of HIP devices visible; does not initialise a device context */
MTGV_API int mtgv_device_count(void);

/* Operand precision of every GEMM-shaped kernel launch of the process (convs, linears, mask and bank GEMMs).
 *   MTGV_PREC_F32   (0)  f32 operands on the f32-input matrix instruction (exact products).
 *   MTGV_PREC_F16X3 (1)  each f32 operand represented as fp16 hi + lo; three fp16 matrix instructions per
 *                        product, f32 accumulate.  Error vs fp64 at the f32 level, ~2x the GEMM rate.  Weights carry
 *                        a power-of-two scale per output row (undone in the accumulator), so their magnitude does
 *                        not matter; activations inside the handles are O(1) by construction; mtgv_op_linear, which
 *                        takes arbitrary f32 data, rescales inputs beyond 2^14 by a power of two (the other mtgv_op_*
 *                        entry points expect activations within the fp16 range, as the handles produce them).
 *                        Values that leave the fp16 range turn into inf - visible, never silently wrong.
 * The initial value comes from the environment (MTGV_GEMM_PREC=f32|f16x3, default f16x3).  Both
 * meet the path's 1e-4 contract against the reference (tests/test_gpu_precision.py).  Not thread-safe: set it
 * before the worker threads start.
 * Concurrency contract for F16X3 (measured on MI355X / ROCm 7.2): kernels that use packed-FP32
 * arithmetic (v_pk_mul_f32, v_pk_fma_f32) and run on ANOTHER STREAM of the same GPU while F16X3 launches are in
 * flight can get wrong lanes.  This library is built without those instructions (every translation unit).  Streams
 * the library itself uses besides the caller's: mtgv_detector_forward forks its prototype branch and two head branches
 * onto library-owned streams and joins them before it returns control of the caller's stream (library kernels only;
 * MTGV_DET_FORK=0 keeps the forward on one stream).  mtgv.Pipeline overlaps its stages (detect + crop / embed / match: three streams, the
 * latter two at high priority, the detector's fork-join off) only with MTGV_OVERLAP=on, and then runs nothing but library kernels on them - its output tensors are uninitialised allocations filled by the
 * kernels, the glue between the stages is mtgv_select_cards - plus, with a sharded bank, RCCL's own all-gather kernels
 * (mtgv_bank_topk_packed / mtgv_topk_merge_gathered keep every other step of the exchange inside the library).  A caller
 * that runs foreign kernels (PyTorch elementwise ops included) concurrently on a second stream must either serialise
 * them against the library's stream or select MTGV_PREC_F32. */
#define MTGV_PREC_F32 0
#define MTGV_PREC_F16X3 1
MTGV_API int mtgv_set_gemm_precision(int32_t prec);
MTGV_API int mtgv_get_gemm_precision(int32_t* prec);

/* Measurement aid: when enabled, every launch of the GEMM kernel is bracketed by HIP
 * events on its own stream.  _read waits for them and returns the summed kernel time (ms), the
 * summed algorithmic FLOPs (2*M*N*K of the unpadded problems) and the launch count since enable. */
MTGV_API int mtgv_profile_gemm(int32_t enable);
MTGV_API int mtgv_profile_gemm_read(double* total_ms, double* total_flops, int64_t* launches);
/* compulsory HBM bytes of the launches since enable: every operand element read once, every result written once */
MTGV_API int mtgv_profile_gemm_bytes(double* total_bytes);
/* write one CSV row per recorded launch (shape, tile, ms, TFLOP/s) */
MTGV_API int mtgv_profile_gemm_dump(const char* csv_path);

/* ------------------------------------------------------------------------- */
/* Encoder: ConvNeXt-V2 embedding forward.                                    */
/* Replaces CoreMlEncoder.predict (mtgvision/encoder_export.py:85-110),       */
/* ConvNeXtV2Encoder.forward (mtgvision/models/convnextv2ae.py:256-266),      */
/* ConvNeXtV2.forward (mtgvision/models/convnextv2.py:292-303) and            */
/* MtgVisionEncoder.encode (mtgvision/encoder_train.py:356-358).              */
/* ------------------------------------------------------------------------- */
typedef struct mtgv_encoder mtgv_encoder;

enum { MTGV_ENC_AE = 0, MTGV_ENC_PLAIN = 1 };
enum {
  MTGV_HEAD_CONV_LINEAR = 0, /* convnextv2ae.py:219-235 */
  MTGV_HEAD_CONV_MLP = 1,
  MTGV_HEAD_CONV_ACT_MLP = 2,
  MTGV_HEAD_POOL_LINEAR = 3, /* convnextv2ae.py:236-248 */
  MTGV_HEAD_POOL_MLP = 4,
  MTGV_HEAD_PLAIN = 5        /* GAP -> nn.LayerNorm -> Linear, convnextv2.py:280-303 */
};
enum { MTGV_IN_NCHW_F32 = 0, MTGV_IN_NHWC_F32 = 1, MTGV_IN_NHWC_U8 = 2 };

typedef struct {
  int32_t kind;       /* MTGV_ENC_* */
  int32_t image_h, image_w;
  int32_t in_chans;   /* 3 */
  int32_t z_size;
  int32_t depths[4];
  int32_t dims[4];
  int32_t head_type;  /* MTGV_HEAD_* */
  int32_t scale_io;   /* x*2-1 first (convnextv2ae.py:257-258) */
  int32_t max_batch;  /* workspace is sized for this many images */
} mtgv_encoder_cfg;

MTGV_API int mtgv_encoder_create(const mtgv_encoder_cfg* cfg, mtgv_encoder** out);
MTGV_API void mtgv_encoder_destroy(mtgv_encoder* h);
/* Upload one parameter by its reference state_dict key (layout as stored by
 * PyTorch: conv OIHW, linear [out,in]); numel must match.  Unknown key -> 2. */
MTGV_API int mtgv_encoder_set_param(mtgv_encoder* h, const char* key, const float* data_host, int64_t numel);
/* number of parameters still unset (0 = ready) */
MTGV_API int mtgv_encoder_missing_params(const mtgv_encoder* h);
/* x_dev: n images in `layout`; z_dev: (n, z_size) float32. */
MTGV_API int mtgv_encoder_forward(mtgv_encoder* h, const void* x_dev, int32_t layout, int32_t n, float* z_dev, void* stream);
/* mode 1: forwards of <= max_n images (default 16, 0 keeps it) replay a hipGraph captured per batch size;
 * mode 0 (default): every launch eager.  Measured: no latency gain on this path - its ~90 short kernels are
 * serialised by dependent-kernel boundaries, which a graph does not remove (profiles/README.md). */
MTGV_API int mtgv_encoder_set_graph(mtgv_encoder* h, int32_t mode, int32_t max_n);
/* keep a copy of every stage output of subsequent forwards (test/debug aid; off by default) */
MTGV_API int mtgv_encoder_set_capture(mtgv_encoder* h, int32_t on);
/* copy stage s (0..3) output of the last forward, NHWC (n, h, w, c), into out_dev; for tests */
MTGV_API int mtgv_encoder_stage_output(mtgv_encoder* h, int32_t stage, int32_t n, float* out_dev, void* stream);
/* algorithmic FLOPs (2*MAC) of one image's forward, split MFMA-eligible GEMM / depthwise */
MTGV_API int mtgv_encoder_flops(const mtgv_encoder* h, double* gemm_flops, double* dw_flops);

/* ------------------------------------------------------------------------- */
/* Bank: exact cosine top-k over all card vectors.                            */
/* Replaces VectorStoreQdrant.query_nearby / save_points / retrieve           */
/* (mtgvision/qdrant.py:17-111; collection "mtg", size 768, Distance.COSINE). */
/* ------------------------------------------------------------------------- */
typedef struct mtgv_bank mtgv_bank;

MTGV_API int mtgv_bank_create(int32_t dim, int64_t capacity, mtgv_bank** out);
MTGV_API void mtgv_bank_destroy(mtgv_bank* h);
MTGV_API int64_t mtgv_bank_size(const mtgv_bank* h);
/* append n vectors (device or host memory, is_device says which); stored L2-normalised */
MTGV_API int mtgv_bank_append(mtgv_bank* h, const float* vecs, int64_t n, int32_t is_device, void* stream);
/* overwrite row `row` (save_points on an existing id) */
MTGV_API int mtgv_bank_set_row(mtgv_bank* h, int64_t row, const float* vec_host, void* stream);
MTGV_API int mtgv_bank_clear(mtgv_bank* h);
/* copy stored (normalised) rows [row, row+n) to out_host */
MTGV_API int mtgv_bank_get_rows(const mtgv_bank* h, int64_t row, int64_t n, float* out_host);
/* q_dev: (b, dim) raw query vectors.  Writes ids (b,k) int64 (row index + id_base, -1 = none)
 * and scores (b,k) float32, sorted by score desc then id asc.  score_threshold is query_nearby's
 * (mtgvision/qdrant.py:83,93): hits scoring below it are dropped on the device (id -1 / score -inf, like a bank with
 * fewer than k rows); -INFINITY keeps everything, NaN is rejected. */
MTGV_API int mtgv_bank_topk(mtgv_bank* h, const float* q_dev, int32_t b, int32_t k, int64_t id_base, float score_threshold,
                            int64_t* ids_dev, float* scores_dev, void* stream);
/* Batches of >= 128 queries (dim % 64 == 0, k <= 4, F16X3 mode) are matched in two passes: approximate fp16 scores over a
 * hi-only copy of the bank, then an exact re-rank of the best candidates with a bound that proves no other row can enter
 * the top k; a query whose bound fails is scanned exactly, so the answer is always the exact top k.  count: how many
 * queries took that scan so far (banks with many near-duplicate rows); synchronises the device.  MTGV_MATCH_PREPASS=0
 * keeps to the one-pass exact kernel. */
MTGV_API int mtgv_bank_prepass_fallbacks(const mtgv_bank* h, int64_t* count);
/* merge ncand (score,id) candidates per query into the top k (multi-GPU shard merge); same threshold rule */
MTGV_API int mtgv_topk_merge(float* cand_scores_dev, const int64_t* cand_ids_dev, int32_t b, int32_t ncand, int32_t k,
                             float score_threshold, int64_t* ids_dev, float* scores_dev, void* stream);
/* The sharded match's exchange step without any arithmetic outside the library (mtgv/dist.py): the local
 * top-k of ALL ranks' queries over this rank's shard is written in the exchange format packed[b][k][2] int64 =
 * (global id or -1, float32 bit pattern of the score, zero-extended) - one buffer, so one all-gather carries ids and
 * scores; after the all-gather, gathered[n_ranks][b_total][k][2] is merged for this rank's own queries
 * [row0, row0 + b) (score desc, id asc; threshold rule as mtgv_bank_topk; n_ranks * k <= 4096).  Between the two calls
 * the caller issues only the collective (RCCL's own kernels). */
MTGV_API int mtgv_bank_topk_packed(mtgv_bank* h, const float* q_dev, int32_t b, int32_t k, int64_t id_base, int64_t* packed_dev,
                                   void* stream);
MTGV_API int mtgv_topk_merge_gathered(const int64_t* gathered_dev, int32_t n_ranks, int32_t b_total, int32_t k, int32_t row0, int32_t b,
                                      float score_threshold, int64_t* ids_dev, float* scores_dev, void* stream);

/* ------------------------------------------------------------------------- */
/* Detector: YOLOv8n-seg forward + decode + NMS + mask logits.                */
/* Replaces CardSegmenter.__call__ -> ultralytics YOLO predict                */
/* (mtgvision/od_export.py:141-160; model built in od_train.py:46-70).        */
/* ------------------------------------------------------------------------- */
typedef struct mtgv_detector mtgv_detector;

typedef struct {
  int32_t nc;        /* classes (3: od_train.py:46-50) */
  int32_t imgsz;     /* 640 */
  int32_t max_batch;
  float conf;        /* 0.25 */
  float iou;         /* 0.7 */
  int32_t max_det;   /* 300 */
  int32_t arch;      /* 0 or 8: YOLOv8n-seg; 11: YOLO11n-seg (C3k2, C2PSA, depthwise class branch; od_train.py:20) */
} mtgv_detector_cfg;

MTGV_API int mtgv_detector_create(const mtgv_detector_cfg* cfg, mtgv_detector** out);
MTGV_API void mtgv_detector_destroy(mtgv_detector* h);
/* parameter by ultralytics state_dict key ("model.0.conv.weight", "model.0.bn.running_var", ...) */
MTGV_API int mtgv_detector_set_param(mtgv_detector* h, const char* key, const float* data_host, int64_t numel);
MTGV_API int mtgv_detector_missing_params(const mtgv_detector* h);
/* fold BatchNorm into the conv weights and repack; call once after all params are set */
MTGV_API int mtgv_detector_finalize(mtgv_detector* h);
/* frames_dev: (n, imgsz, imgsz, 3) uint8, already letterboxed; flip_rgb reverses the channel
 * order first (ultralytics treats ndarray input as BGR).
 * Outputs (device): n_det (n) int32; per frame up to max_det rows of
 *   boxes (n, max_det, 4) xyxy pixels, conf (n, max_det), cls (n, max_det) int32,
 *   keep_idx (n, max_det) int32 anchor index in [0, 8400),
 *   mask_logits (n, mask_rows, 160, 160): coeffs @ protos cropped to the box (process_mask before
 *   the upsample) for the first min(n_det, mask_rows) detections of each frame; rows beyond
 *   n_det are left untouched; NULL skips the mask stage. */
MTGV_API int mtgv_detector_forward(mtgv_detector* h, const uint8_t* frames_dev, int32_t n, int32_t flip_rgb,
                                   int32_t* n_det_dev, float* boxes_dev, float* conf_dev, int32_t* cls_dev,
                                   int32_t* keep_idx_dev, float* mask_logits_dev, int32_t mask_rows, void* stream);
/* raw head outputs of the last forward: pred (n, 4+nc+32, 8400) and protos (n, 32, 160, 160) */
MTGV_API int mtgv_detector_raw(mtgv_detector* h, int32_t n, float* pred_dev, float* protos_dev, void* stream);
MTGV_API int mtgv_detector_flops(const mtgv_detector* h, double* flops_per_frame);
/* The forward's internal fork-join (the prototype branch and the P3 / P4 head branches on library-owned streams, see the
 * concurrency contract above): mode 1 on, 0 off (every launch on the caller's stream), -1 the default - on unless the
 * environment says MTGV_DET_FORK=0.  A caller that already overlaps the detector with other work on a second stream turns it
 * off (mtgv.Pipeline.run_many does: branch streams share the runtime's few hardware queues with the caller's other streams,
 * and a long event wait queued in one of them holds up whatever shares it; od_export.py:147-150 has no counterpart). */
MTGV_API int mtgv_detector_set_fork(mtgv_detector* h, int32_t mode);
/* process_mask tail: bilinear x`scale` upsample (align_corners=False) of (n, mh, mw) logits, then > 0
 * -> (n, mh*scale, mw*scale) uint8 {0,1} */
MTGV_API int mtgv_mask_binarize(const float* logits_dev, int32_t n, int32_t mh, int32_t mw, int32_t scale, uint8_t* out_dev,
                                void* stream);

/* NMS alone on decoded predictions pred (n, 4+nc+nm, na) [xywh, class scores, coeffs] */
MTGV_API int mtgv_nms(const float* pred_dev, int32_t n, int32_t nc, int32_t nm, int32_t na, float conf, float iou,
                      int32_t max_det, float max_wh, int32_t* n_det_dev, float* boxes_dev, float* conf_dev, int32_t* cls_dev,
                      int32_t* keep_idx_dev, int32_t* workspace_dev, size_t workspace_bytes, void* stream);
MTGV_API size_t mtgv_nms_workspace_bytes(int32_t n, int32_t na);

/* ------------------------------------------------------------------------- */
/* Mask -> oriented card quad.                                                */
/* Replaces the host geometry of InstanceSeg._orient                          */
/* (mtgvision/od_export.py:52-93: close the U-shaped mask, four corners,      */
/* corner 0 = the card's top-left).                                           */
/* ------------------------------------------------------------------------- */
/* masks_dev (n, h, w) uint8, non-zero = foreground (mtgv_mask_binarize output); boxes_dev (n, 4) xyxy float32 or NULL:
 * the quad reported for an empty mask.  quads_dev (n, 4, 2) float32 corners (x, y) in pixels of the mask grid, ordered
 * top-left, top-right, bottom-right, bottom-left of the card; ok_dev (n) int32: 1 = from the mask, 0 = empty mask.
 * The quad is the general quadrilateral cv2.approxPolyN(hull, 4) fits (od_export.py:72-74): the convex hull of the mask's
 * row extents, greedily contracted edge by edge (the edge whose removal adds the least area is replaced by the
 * intersection of its neighbours, first minimum wins) until four vertices remain - vertices may therefore lie outside
 * the mask, or outside the image; coordinates are truncated toward zero as the reference's astype(int) does.  "Up" is
 * mask centroid minus hull centroid: the ray from the quad's centre along it picks the card's top edge.  Fallbacks: a
 * hull of fewer than four vertices (point, line, triangle) reports the mask's bounding box; an empty mask reports
 * boxes_dev[n] (or zeros) with ok = 0. */
/* extents_dev: NULL, or (n, h, 2) int32 that receives every mask row's leftmost and rightmost foreground column (-1, -1 for
 * an empty row): the outline of the mask as the host needs it for `InstanceSeg.points` (od_export.py:152-153), 2 h
 * integers per card instead of the h x w mask. */
MTGV_API int mtgv_mask_quads(const uint8_t* masks_dev, int32_t n, int32_t h, int32_t w, const float* boxes_dev, float* quads_dev,
                             int32_t* ok_dev, int32_t* extents_dev, void* stream);
/* the same from the cropped mask logits (n, mh, mw) of the detector: a pixel of the (mh*scale, mw*scale) mask is foreground
 * where the bilinear interpolation of the logits is > 0 (what mtgv_mask_binarize writes) - the full-resolution mask is
 * never materialised.  Identical quads to mtgv_mask_binarize + mtgv_mask_quads. */
MTGV_API int mtgv_mask_quads_logits(const float* logits_dev, int32_t n, int32_t mh, int32_t mw, int32_t scale,
                                    const float* boxes_dev, float* quads_dev, int32_t* ok_dev, int32_t* extents_dev, void* stream);

/* The K cards of every frame that go on to the crop stage: the K highest-confidence detections of the padded
 * mtgv_detector_forward outputs (n_det (frames), boxes (frames, max_det, 4), score-descending) or, where a frame has
 * fewer, pad_boxes_dev[k] (k, 4).  Writes sel_boxes_dev (frames*k, 4) xyxy, frame_idx_dev (frames*k) and, unless null,
 * quads_dev (frames*k, 4, 2): the boxes' corners in the order mtgv_warp_quads expects.  The glue between
 * `results.boxes` and `extract_dewarped` (mtgvision/od_export.py:152-160, server.py:139-183), batched. */
MTGV_API int mtgv_select_cards(const int32_t* n_det_dev, const float* boxes_dev, const float* pad_boxes_dev, int32_t frames,
                               int32_t max_det, int32_t k, float* sel_boxes_dev, float* quads_dev, int32_t* frame_idx_dev,
                               void* stream);

/* ------------------------------------------------------------------------- */
/* Crop: perspective de-warp of card quads.                                   */
/* Replaces InstanceSeg.extract_dewarped (mtgvision/od_export.py:95-111).     */
/* ------------------------------------------------------------------------- */
/* frames_dev (nf, fh, fw, 3) uint8; quads_dev (nq, 4, 2) float32 source corners (x,y) in the
 * order dst corners [[0,0],[w,0],[w,h],[0,h]] are matched to; frame_idx_dev (nq) int32.
 * out_dev (nq, out_h, out_w, 3) uint8.  workspace_dev: mtgv_warp_workspace_bytes(nq) bytes. */
MTGV_API size_t mtgv_warp_workspace_bytes(int32_t nq);
MTGV_API int mtgv_warp_quads(const uint8_t* frames_dev, int32_t nf, int32_t fh, int32_t fw, const float* quads_dev,
                             const int32_t* frame_idx_dev, int32_t nq, int32_t out_h, int32_t out_w, double expand_ratio,
                             uint8_t* out_dev, void* workspace_dev, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------- */
/* Bank build: card image -> encoder input.                */
/* Replaces SyntheticBgFgMtgImages.make_cropped (mtgvision/encoder_datasets.py:733-753) */
/* as used by qdrant_populate.CardProcessor._get_card_point (qdrant_populate.py:84-90). */
/* ------------------------------------------------------------------------- */
/* images_dev: n uint8 HWC images of arbitrary sizes back to back; offsets_dev (n) byte offset of each;
 * hw_dev (n,2) int32 height,width.  Strips ceil(max(0.02 H, 0.02 W)) border pixels, area-resizes to
 * (out_h, out_w), clips: out_dev (n, out_h, out_w, 3) float32 in [0,1] (feed with MTGV_IN_NHWC_F32). */
MTGV_API int mtgv_make_cropped(const uint8_t* images_dev, const int64_t* offsets_dev, const int32_t* hw_dev, int32_t n,
                               int32_t out_h, int32_t out_w, float* out_dev, void* stream);

/* ultralytics LetterBox in front of the detector (behind CardSegmenter.__call__, mtgvision/od_export.py:147-150): the (h, w, 3)
 * uint8 frame at src_dev resampled bilinearly (align_corners = false form) to (nh, nw), placed at (top, left) of the
 * size x size x 3 uint8 image at dst_dev, the rest filled with pad_value (114 upstream).  The caller computes the geometry
 * (mtgv.detector.letterbox_geometry: r = min(size / h, size / w), nh = round(h r), ...). */
MTGV_API int mtgv_letterbox_u8(const uint8_t* src_dev, int32_t h, int32_t w, uint8_t* dst_dev, int32_t size, int32_t nh, int32_t nw,
                               int32_t top, int32_t left, int32_t pad_value, void* stream);

/* ------------------------------------------------------------------------- */
/* Single ops (unit-test and composition surface; same kernels the handles use) */
/* ------------------------------------------------------------------------- */
/* out[M,N] = act(A[M,K] W[N,K]^T + bias) (+res);  act: 0 none 1 gelu 2 mish 3 silu 4 sigmoid */
MTGV_API int mtgv_op_linear(const float* a_dev, const float* w_dev, const float* bias_dev, const float* res_dev, float* out_dev,
                            int32_t m, int32_t n, int32_t k, int32_t act, void* stream);
/* linear with the block's fused extras: rows are grouped in images of hw rows; a_scale (m/hw, k) and a_shift (k)
 * or NULL are applied to A on load (GRN apply); grn_part or NULL receives the per-row-unit sum(out^2) partials
 * (a buffer of mtgv_op_linear_ex_part_floats floats is large enough for any layout). */
MTGV_API int64_t mtgv_op_linear_ex_part_floats(int32_t m, int32_t n, int32_t k, int32_t act, int32_t hw);
/* layout of the partials the calling thread's last mtgv_op_linear_ex wrote: [ceil(m / unit_rows)][segmax][n] */
MTGV_API int mtgv_op_last_grn_layout(int32_t* unit_rows, int32_t* segmax);
MTGV_API int mtgv_op_linear_ex(const float* a_dev, const float* w_dev, const float* bias_dev, const float* res_dev, float* out_dev,
                               int32_t m, int32_t n, int32_t k, int32_t act, int32_t hw, const float* a_scale_dev,
                               const float* a_shift_dev, float* grn_part_dev, void* stream);
/* NHWC conv, weight (cout, kh, kw, cin), zero padding */
MTGV_API int mtgv_op_conv2d(const float* x_dev, const float* w_dev, const float* bias_dev, float* out_dev, int32_t n, int32_t h,
                            int32_t w, int32_t cin, int32_t cout, int32_t kh, int32_t kw, int32_t stride, int32_t pad,
                            int32_t act, void* stream);
MTGV_API int mtgv_op_layernorm(const float* x_dev, const float* w_dev, const float* b_dev, float* out_dev, int64_t rows,
                               int32_t c, float eps, void* stream);
/* depthwise 7x7 pad 3; weight (49, c) tap-major */
MTGV_API int mtgv_op_dwconv7(const float* x_dev, const float* w49_dev, const float* bias_dev, float* out_dev, int32_t n,
                             int32_t h, int32_t w, int32_t c, void* stream);
/* one ConvNeXt-V2 block on NHWC x (n,h,w,c): convnextv2.py:212-224.  params in reference layout
 * except dw weight (49,c).  ws_dev: workspace of mtgv_op_block_workspace_floats() floats. */
MTGV_API int mtgv_op_block(const float* x_dev, float* out_dev, int32_t n, int32_t h, int32_t w, int32_t c, int32_t act,
                           const float* dw_w49, const float* dw_b, const float* ln_w, const float* ln_b, const float* w1,
                           const float* b1, const float* gamma, const float* beta, const float* w2, const float* b2,
                           float* ws_dev, void* stream);
MTGV_API int64_t mtgv_op_block_workspace_floats(int32_t n, int32_t h, int32_t w, int32_t c);
MTGV_API int mtgv_op_l2norm(const float* x_dev, float* out_dev, int64_t rows, int32_t d, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MTGV_H */
