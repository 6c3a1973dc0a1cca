/*
 * mi_detectron_ops.h -- C-ABI of the MI355X (gfx950) RoI-transform / NMS hot path.
 *
 * One shared library (libmi_detectron_ops.so, built by `hipcc --offload-arch=gfx950`)
 * exports exactly the symbols declared here.  Signatures carry plain device pointers,
 * sizes and an opaque stream handle (a hipStream_t passed as void*); there are no torch,
 * THC or C++ types on the boundary.
 *
 * Conventions (they restate the reference's C boundary):
 *   - every pointer except `mi_*_host` arguments is a DEVICE pointer on the caller's
 *     current device; the library never allocates, never frees and never synchronises;
 *   - the caller owns every buffer; backward entry points ACCUMULATE into `bottom_grad`
 *     (the reference caller zero-fills it first: roi_xfrom/roi_align/functions/roi_align.py:39-40);
 *   - work is enqueued on `stream` (reference: THCState_getCurrentStream, roi_align_cuda.c:31);
 *     calls are re-entrant: the only process-wide state is the per-thread last-error string and a tuning struct that is
 *     initialised once from the environment and read-only afterwards (see mi_dbg_reload_tuning at the end);
 *   - return value: MI_OK (0) on success, a positive MI_ERR_* code otherwise.  The
 *     reference printed to stderr and called exit(-1) on a launch failure
 *     (roi_align_kernel.cu:135-139) and returned 0 for a malformed rois tensor
 *     (roi_align_cuda.c:19-22); here both become error codes and the Python shim raises.
 *   - rois are [R,5] float32 rows (batch_index, x1, y1, x2, y2) in input-image pixels.
 *     A RoI whose batch_index lies outside [0, batch) pools zeros and receives no gradient (the reference would read
 *     out of bounds there); callers that pad a RoI set to a static size mark the padding rows with batch_index -1.
 *   - `layout` selects the memory order of the 4-D feature / gradient tensor:
 *     MI_LAYOUT_NCHW (the reference's only layout) or MI_LAYOUT_NHWC (torch
 *     channels_last storage of the same logical [N,C,H,W] tensor).  Outputs of the
 *     forward ops are always dense [R,C,PH,PW].
 */
#ifndef MI_DETECTRON_OPS_H_
#define MI_DETECTRON_OPS_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2: round-2/3 entry points (top-k, segmented NMS, RPN collect, affine, result formats, FPN-fused RoIAlign), RoIs of a
 * non-existent image pool zeros.  3 (round 4): the opt-in tile-centric NCHW forward and its two entry points
 * (mi_roi_align_forward_tiles_workspace_bytes, mi_roi_align_forward_fpn_writes_records) are gone -- measured slower on
 * every shape but one; every fast forward now leaves its records (mi_roi_align_forward_writes_records).  4 (round 5):
 * mi_rpn_collect_finish_records + mi_roi_align_forward_fpn_records (the producer of the RoIs writes their records); a
 * dword-aligned top_grad is served by the generic backward instead of refused.  5 (round 5): mi_polys_to_masks_wrt_boxes;
 * the generic backward zero-fills under MI_ROI_ALIGN_OVERWRITE.  6 (round 6): mi_fpn_level_index_from_restore; the RoIAlign
 * backward over a workspace is two launches (no trailing launch: its plan counters alternate between two sets), the
 * RoIPool / RoICrop kernels are LDS-staged (same entry points, same results).  7 (round 6): mi_roi_pool_backward OVERWRITES
 * (tile kernel, the reference's addition order, no atomics); mi_roi_crop_backward_ws + _workspace_bytes (tile kernel,
 * overwrites, no atomics).  8 (round 6): no new entry point -- a forward whose workspace has no room for a backward (smaller
 * than mi_roi_align_backward_workspace_bytes), and the forward without a workspace, run the records-free NCHW kernel
 * (roi_align_fwd_slab: ONE launch, nothing written to the workspace); mi_roi_align_forward_writes_records speaks of a
 * backward-sized workspace. */
#define MI_ABI_VERSION 8

typedef void* mi_stream_t; /* hipStream_t */

enum mi_status {
  MI_OK = 0,
  MI_ERR_BAD_ARGUMENT = 1,  /* null pointer, negative size, unknown enum value          */
  MI_ERR_LAUNCH = 2,        /* hipGetLastError() != hipSuccess after the launch        */
  MI_ERR_WORKSPACE = 3,     /* workspace smaller than mi_*_workspace_bytes()            */
  MI_ERR_UNSUPPORTED = 4    /* shape outside what the kernels implement (see message)  */
};

enum mi_layout { MI_LAYOUT_NCHW = 0, MI_LAYOUT_NHWC = 1 };

/* Which RoIAlign arithmetic:
 *   CAFFE2 = lib/modeling/roi_xfrom/roi_align/src/roi_align_kernel.cu:16-121,150-270 (the one the model uses)
 *   LEGACY = lib/model/roi_align/src/roi_align_kernel.cu:15-70,94-143 (1 sample on an (aligned-1) grid;
 *            `sampling_ratio` is ignored) */
enum mi_roi_align_variant { MI_ROI_ALIGN_CAFFE2 = 0, MI_ROI_ALIGN_LEGACY = 1 };

/* NMS result conventions:
 *   GE_ORIG_ASC   = lib/utils/cython_nms.pyx:37-87: dets may be unsorted, sorted internally by
 *                   descending score (ties: higher original index first == np.argsort(kind='stable')[::-1]),
 *                   suppress when IoU >= thresh, keep[] = int64 ORIGINAL indices in ascending order.
 *   GT_SORTED_POS = lib/model/nms/src/nms_cuda_kernel.cu:41-161 + nms_gpu.py:7-12: dets must already be
 *                   sorted by descending score, suppress when IoU > thresh, keep[] = int32 positions
 *                   in the input, in kept (= ascending position) order. */
enum mi_nms_mode { MI_NMS_GE_ORIG_ASC = 0, MI_NMS_GT_SORTED_POS = 1 };

int mi_abi_version(void);
/* Message of the most recent failing call on this host thread ("" if none). */
const char* mi_last_error(void);

/* ---- RoIAlign -------------------------------------------------------------------------
 * replaces ROIAlignForwardLaucher / ROIAlignBackwardLaucher
 *   (lib/modeling/roi_xfrom/roi_align/src/roi_align_kernel.h:13-30; legacy lib/model/roi_align/src/roi_align_kernel.h:13-30)
 * and the cffi glue roi_align_forward_cuda / roi_align_backward_cuda (roi_align_cuda.c:7-76). */
int mi_roi_align_forward(const float* features, const float* rois, float* output,
                         int batch, int channels, int height, int width, int num_rois,
                         int aligned_height, int aligned_width, float spatial_scale,
                         int sampling_ratio, int variant, int layout, mi_stream_t stream);

/* Same operation with caller-provided device scratch.  What the call does with it depends on its SIZE:
 *   >= mi_roi_align_backward_workspace_bytes(...)  "a backward over these RoIs follows": a first launch condenses each RoI
 *        into a record in `workspace` (window, tap tables, LDS stages, the backward's merged weights) at its rank along a sweep
 *        of the image, a second launch consumes the records (roi_align_fwd_records on NCHW features, roi_align_fwd_nhwc on
 *        channels-last ones), and mi_roi_align_backward_ws may reuse them (MI_ROI_ALIGN_RECORDS_READY);
 *   >= mi_roi_align_forward_workspace_bytes(num_rois) only  NCHW: ONE launch, nothing written (roi_align_fwd_slab: one wave
 *        per (RoI, 8 channels), an XCD reads one 8-channel slab at a time; ABI 8); channels-last: records + roi_align_fwd_nhwc.
 * 16-byte aligned; contents are scratch (no initialisation needed, overwritten by every call; two calls that may run
 * concurrently on different streams need distinct workspaces).  workspace == NULL behaves exactly like mi_roi_align_forward
 * (NCHW: the same one-launch kernel). */
size_t mi_roi_align_forward_workspace_bytes(int num_rois);
int mi_roi_align_forward_ws(const float* features, const float* rois, float* output,
                            int batch, int channels, int height, int width, int num_rois,
                            int aligned_height, int aligned_width, float spatial_scale,
                            int sampling_ratio, int variant, int layout,
                            void* workspace, size_t workspace_bytes, mi_stream_t stream);

/* The reference's signature (ROIAlignBackwardLaucher, roi_align_kernel.cu:272-291): ACCUMULATES into a caller-zeroed bottom_grad with one global
 * fp32 atomic per tap, as ROIAlignBackward does -- correct on every shape and ~30x slower than the gather of
 * mi_roi_align_backward_ws (1.6 ms against 48 us at 512 RoIs x 256 ch x 7x7): a training loop passes a workspace. */
int mi_roi_align_backward(const float* top_grad, const float* rois, float* bottom_grad,
                          int batch, int channels, int height, int width, int num_rois,
                          int aligned_height, int aligned_width, float spatial_scale,
                          int sampling_ratio, int variant, int layout, mi_stream_t stream);

/* Backward with caller scratch (same workspace contract as mi_roi_align_forward_ws).  `flags`:
 *   MI_ROI_ALIGN_RECORDS_READY  `workspace` still holds the records written by mi_roi_align_forward_ws for the SAME
 *                               rois, feature-map size, aligned size, spatial_scale and sampling_ratio (nothing else
 *                               touched it since) AND the workspace is at least
 *                               mi_roi_align_backward_workspace_bytes() large (only then did the forward write the records'
 *                               backward block): the record launch is skipped; otherwise the flag is ignored;
 *   MI_ROI_ALIGN_OVERWRITE      bottom_grad is fully WRITTEN (no zero fill by the caller needed) instead of
 *                               accumulated into (the reference contract, functions/roi_align.py:39-44).  Honoured on
 *                               the tile path only (NCHW or channels-last bottom_grad); test
 *                               mi_roi_align_backward_overwrites() before relying on it.
 * The tile path is a gather over tiles of bottom_grad: a fixed summation order per tile; with a workspace of
 * mi_roi_align_backward_workspace_bytes() lists of more than 32 RoIs are cut into slices whose sums are added with fp32
 * atomics (see there).  The tile kernel fetches a RoI's block of gradients in 16-byte pieces: a `top_grad` that is only
 * dword-aligned takes the generic kernel instead (the reference's arithmetic and global atomics, one workgroup per (RoI, 32 channels); under MI_ROI_ALIGN_OVERWRITE the zero fill
 * is then done here). */
#define MI_ROI_ALIGN_RECORDS_READY 1
#define MI_ROI_ALIGN_OVERWRITE 2
int mi_roi_align_backward_ws(const float* top_grad, const float* rois, float* bottom_grad,
                             int batch, int channels, int height, int width, int num_rois,
                             int aligned_height, int aligned_width, float spatial_scale,
                             int sampling_ratio, int variant, int layout,
                             void* workspace, size_t workspace_bytes, int flags, mi_stream_t stream);
/* 1 when mi_roi_align_forward_ws with these arguments AND a workspace of at least mi_roi_align_backward_workspace_bytes
 * leaves the records of its rois in the workspace (so that a backward over the same rois may pass
 * MI_ROI_ALIGN_RECORDS_READY); 0 when it takes a path without records.  With a smaller (forward-sized) workspace the
 * NCHW forward writes no records (one launch, roi_align_fwd_slab); a backward over such a workspace ignores
 * MI_ROI_ALIGN_RECORDS_READY and writes its own, as it always has. */
int mi_roi_align_forward_writes_records(int channels, int height, int width, int num_rois, int aligned_height,
                                        int aligned_width, int variant, int layout);
/* 1 when mi_roi_align_backward_ws would honour MI_ROI_ALIGN_OVERWRITE for these arguments (with a workspace). */
int mi_roi_align_backward_overwrites(int channels, int height, int width, int num_rois, int aligned_height,
                                     int aligned_width, int variant, int layout);

/* ---- RoIAlign over an FPN pyramid, one call ----------------------------------------------------------------------
 * replaces the per-level loop of Generalized_RCNN.roi_feature_transform (lib/modeling/model_builder.py:266-306): one
 * RoIAlignFunction call, one H2D copy and one output tensor per level, then torch.cat and a gather by
 * `<rois>_idx_restore_int32`.  Here all levels are served by ONE pair of launches and the output is written directly in
 * the order of `rois`: every RoI carries the index of its level (`roi_levels`, device int32 [R], values
 * 0..num_levels-1 -- utils/fpn.py:11-28 computes the assignment on the host in the reference's data layer).
 * Caffe2 semantics; the maps share batch, channels and `layout` (MI_LAYOUT_NCHW, or MI_LAYOUT_NHWC for channels-last
 * storage of maps and gradient maps; output / top_grad are dense [R,C,PH,PW] either way).  Same workspace, records and flags as the single-level
 * _ws entry points (the records of a forward serve the backward over the same rois and roi_levels); the backward
 * writes (MI_ROI_ALIGN_OVERWRITE) or accumulates into every level's gradient map, also those no RoI maps to.
 * mi_roi_align_fpn_supported() == 0: shapes the fused path does not take -- loop over the levels instead. */
#define MI_FPN_MAX_LEVELS 4
typedef struct mi_fpn_levels {
  int num_levels;                             /* 1 .. MI_FPN_MAX_LEVELS */
  const float* features[MI_FPN_MAX_LEVELS];   /* forward: [N,C,height[l],width[l]]; ignored by the backward */
  float* grads[MI_FPN_MAX_LEVELS];            /* backward: gradient map of level l; ignored by the forward */
  int height[MI_FPN_MAX_LEVELS];
  int width[MI_FPN_MAX_LEVELS];
  float spatial_scale[MI_FPN_MAX_LEVELS];
} mi_fpn_levels;
int mi_roi_align_fpn_supported(const mi_fpn_levels* levels, int channels, int num_rois, int aligned_height,
                               int aligned_width, int layout);
/* Workspace with room for the PLANNED backward (>= mi_roi_align_forward_workspace_bytes(num_rois)): behind the records a
 * pre-kernel (roi_align_bwd_plan) leaves, per 16 x 32 tile of the gradient maps, the list of RoIs that touch it, and the
 * tile kernel's workgroups each take a slice of at most MI_ROI_ALIGN_BWD_SLICE (32) RoIs of one list -- the RoIs of a
 * training step cluster on the ground-truth boxes and a few tiles see hundreds of them.  The slices of one list add their
 * sums with fp32 atomics (order not fixed; lists of <= 32 RoIs are summed in a fixed order as before).  Passing only the
 * forward size to mi_roi_align_backward_ws / _fpn keeps the unplanned backward (one workgroup per tile walks the whole
 * list, no atomics).  A forward given a workspace of at least this size also writes the records' backward block (merged
 * pass weights, +2 us in its records launch); MI_ROI_ALIGN_RECORDS_READY is honoured for such a workspace only -- with a
 * smaller one the backward rewrites the records.  Only height[] / width[] / num_levels of `levels` are read (one level
 * for the single-map entries).
 * Replaces: the reference's backward is atomicAdd per tap throughout (roi_align_kernel.cu:195-270). */
size_t mi_roi_align_backward_workspace_bytes(const mi_fpn_levels* levels, int batch, int num_rois);
int mi_roi_align_forward_fpn(const mi_fpn_levels* levels, const float* rois, const int32_t* roi_levels, float* output,
                             int batch, int channels, int num_rois, int aligned_height, int aligned_width,
                             int sampling_ratio, int layout, void* workspace, size_t workspace_bytes,
                             mi_stream_t stream);
/* The same forward over records that are ALREADY in `workspace`: written for exactly these rois / roi_levels / geometry /
 * level table by mi_rpn_collect_finish_records (below) on the same stream -- the records launch is skipped.  The library
 * cannot check that the records belong to the rois; a workspace sized for a backward must have been that size when the
 * records were written (their backward block). */
int mi_roi_align_forward_fpn_records(const mi_fpn_levels* levels, const float* rois, const int32_t* roi_levels,
                                     float* output, int batch, int channels, int num_rois, int aligned_height,
                                     int aligned_width, int sampling_ratio, int layout, void* workspace,
                                     size_t workspace_bytes, mi_stream_t stream);
/* (mi_roi_align_backward_fpn has no generic fallback: `top_grad` must be 16-byte aligned -- MI_ERR_BAD_ARGUMENT otherwise;
 * a caller with a dword-aligned gradient copies it once, as the autograd mirror does, or calls the per-level entry point.) */
int mi_roi_align_backward_fpn(const mi_fpn_levels* levels, const float* top_grad, const float* rois,
                              const int32_t* roi_levels, int batch, int channels, int num_rois, int aligned_height,
                              int aligned_width, int sampling_ratio, int layout, void* workspace,
                              size_t workspace_bytes, int flags, mi_stream_t stream);

/* ---- RoIPool --------------------------------------------------------------------------
 * replaces ROIPoolForwardLaucher / ROIPoolBackwardLaucher (lib/model/roi_pooling/src/roi_pooling_kernel.h:8-20)
 * and roi_pooling_forward_cuda / roi_pooling_backward_cuda (roi_pooling_cuda.c:7,49).
 * argmax[R,C,PH,PW] int32 = flat index into the whole NCHW feature tensor, -1 for an empty bin
 * (roi_pooling_kernel.cu:75-91).  NCHW only, like the reference. */
int mi_roi_pool_forward(const float* features, const float* rois, float* output, int32_t* argmax,
                        int batch, int channels, int height, int width, int num_rois,
                        int pooled_height, int pooled_width, float spatial_scale,
                        mi_stream_t stream);

/* The backward OVERWRITES bottom_grad, every element, as ROIPoolBackward does (roi_pooling_kernel.cu:202): no zero fill by
 * the caller.  Since ABI 7 it adds a pixel's terms in the reference's order (ascending RoI, ph, pw) without atomics: the
 * result is bit-equal to the reference's and the same from run to run. */
int mi_roi_pool_backward(const float* top_grad, const float* rois, const int32_t* argmax,
                         float* bottom_grad,
                         int batch, int channels, int height, int width, int num_rois,
                         int pooled_height, int pooled_width, float spatial_scale,
                         mi_stream_t stream);

/* ---- RoICrop (bilinear grid sampler) ---------------------------------------------------
 * replaces BilinearSamplerBHWD_updateOutput_cuda_kernel / _updateGradInput_cuda_kernel
 *   (lib/model/roi_crop/src/roi_crop_cuda_kernel.h:6-37) and the glue in roi_crop_cuda.c:15,54.
 * input [N,C,H,W] dense NCHW; grid_yx [R,GH,GW,2] with (y,x) in [-1,1]; output [R,C,GH,GW].
 * RoI r samples image r / (R / N) (roi_crop_cuda_kernel.cu:64,217).  Output elements whose four
 * neighbours all fall outside the image are left untouched (the reference `continue`s, :92-93),
 * so the caller zero-fills `output`.  The backward accumulates into grad_input and, like the
 * reference (:111-194), never writes the grid gradient. */
int mi_roi_crop_forward(const float* input, const float* grid_yx, float* output,
                        int batch, int channels, int height, int width,
                        int num_rois, int grid_height, int grid_width, mi_stream_t stream);

int mi_roi_crop_backward(const float* input, const float* grid_yx, const float* grad_output,
                         float* grad_input,
                         int batch, int channels, int height, int width,
                         int num_rois, int grid_height, int grid_width, mi_stream_t stream);

/* The same sums WITHOUT global atomics and WITHOUT the caller's zero fill (ABI 7): grad_input is OVERWRITTEN, every
 * element.  Each 8 x 32-pixel tile of the image gradient is accumulated in LDS from the RoIs whose taps reach it; the
 * workspace (mi_roi_crop_backward_workspace_bytes(num_rois) bytes, 16-byte aligned, caller-owned, contents irrelevant)
 * receives every RoI's bounding box of taps in a first launch.  What the autograd mirror calls; the terms are the
 * reference's ((x weight * y weight) * gradient, roi_crop_cuda_kernel.cu:169-190), their order of addition differs (the
 * reference's atomics leave it undefined). */
size_t mi_roi_crop_backward_workspace_bytes(int num_rois);
int mi_roi_crop_backward_ws(const float* input, const float* grid_yx, const float* grad_output,
                            float* grad_input,
                            int batch, int channels, int height, int width,
                            int num_rois, int grid_height, int grid_width,
                            void* workspace, size_t workspace_bytes, mi_stream_t stream);

/* ---- RPN proposal decode ------------------------------------------------------------------------------------------
 * steps 1-3 of GenerateProposalsOp.proposals_for_one_image (lib/modeling/generate_proposals.py:105-153; helpers
 * lib/utils/boxes.py:156-196 bbox_transform, :138-153 clip_tiled_boxes, generate_proposals.py:170-182 _filter_boxes)
 * for the pre-NMS top-k anchors of every image of one level, in one launch.  The reference does this in numpy on the
 * host after copying the RPN outputs back (:58-63).
 *   bbox_pred [N,4A,H,W]; topk_scores / topk_idx [N,k]: the k best scores of each image in descending order and their
 *   flat indices into that image's [A,H,W] score map; im_info [N,3] (height, width, scale), device; base_anchors_host
 *   [A,4] float64 on the HOST (generate_anchors.py output, A <= 16); feat_stride = 1 / spatial_scale; xform_clip =
 *   cfg.BBOX_XFORM_CLIP.  Writes dets [N,k,5] (x1,y1,x2,y2,score), ready for mi_nms / mi_nms_batched, and valid [N,k]:
 *   boxes the min-size / centre filter rejects become a far-away degenerate box with valid = 0 (IoU 0 with every real
 *   box); drop them AFTER the NMS. */
int mi_rpn_decode_proposals(const float* bbox_pred, const float* topk_scores, const int64_t* topk_idx,
                            const float* im_info, const double* base_anchors_host, int num_images, int num_anchors,
                            int height, int width, int k, double feat_stride, float min_size, double xform_clip,
                            float* dets, int32_t* valid, mi_stream_t stream);

/* ---- NMS -------------------------------------------------------------------------------
 * replaces nms_cuda_compute (lib/model/nms/src/nms_cuda_kernel.h:5-6) / nms_cuda (nms_cuda.c:8-19) in
 * mode GT_SORTED_POS and reproduces utils.cython_nms.nms (cython_nms.pyx:37-87) in mode GE_ORIG_ASC.
 * dets [n,5] float32 (x1,y1,x2,y2,score), "+1" box-size convention.  Fully on-device and
 * asynchronous: keep and num_keep are device buffers (keep: n elements of int64 for GE_ORIG_ASC,
 * of int32 for GT_SORTED_POS; only the first *num_keep are written), workspace is a device
 * scratch of at least mi_nms_workspace_bytes(n) bytes, 16-byte aligned.  n == 0 writes
 * *num_keep = 0. */
size_t mi_nms_workspace_bytes(int n);
int mi_nms(const float* dets, int n, float thresh, int mode, void* keep, int32_t* num_keep,
           void* workspace, size_t workspace_bytes, mi_stream_t stream);

/* Soft-NMS: replaces utils.cython_nms.soft_nms (lib/utils/cython_nms.pyx:98-203; wrapper utils/boxes.py:327-344, call
 * site core/test.py:753-760 when TEST.SOFT_NMS.ENABLED).  dets [n,5] float32, n <= 4096.  method: 0 hard, 1 linear,
 * 2 gaussian (the values utils/boxes.py:334 passes).  Results as the reference returns them -- the re-scored rows
 * boxes[:N] into out_dets ([n,5] buffer, first *num_out rows written) and their original indices inds[:N] into
 * out_inds (int64) -- in the reference's row order (pick order, with its swap-with-last compaction).  On-device,
 * asynchronous, no workspace (state lives in LDS).  n == 0 writes *num_out = 0. */
int mi_soft_nms(const float* dets, int n, float sigma, float overlap_thresh, float score_thresh, int method,
                float* out_dets, int64_t* out_inds, int32_t* num_out, mi_stream_t stream);

/* Many Soft-NMS problems in one launch (the per-class loop of core/test.py:748-760): problem p owns rows
 * [offsets[p], offsets[p + 1]) of dets / out_dets / out_inds; `offsets` is a DEVICE array of num_segments + 1 int32
 * (so the caller needs no host copy of the segment sizes), `max_segment` an upper bound of the longest segment
 * (<= 4096; it sizes the LDS image).  Problem p writes its rows at its own offset, indices relative to its first row,
 * and its row count to num_out[p].  One workgroup per problem: the classes run side by side. */
int mi_soft_nms_segmented(const float* dets, const int32_t* offsets, int num_segments, int max_segment, float sigma,
                          float overlap_thresh, float score_thresh, int method, float* out_dets, int64_t* out_inds,
                          int32_t* num_out, mi_stream_t stream);

/* The per-class NMS of the test-time post-processing (the loop of core/test.py:748-771: `inds = np.where(scores[:, j] >
 * TEST.SCORE_THRESH)`, `keep = box_utils.nms(dets_j, TEST.NMS)`) for all classes in one call, with nothing but the final
 * flags coming back -- the class sizes exist on the device only.  Segment s (= class s + 1 at the call site) has `rows`
 * candidate rows read IN PLACE:   box of row r   = boxes  + s * box_segment_stride   + r * box_row_stride   (4 floats)
 *                                 score of row r = scores + s * score_segment_stride + r * score_row_stride
 * (strides in floats; for the reference's scores [R,C] / boxes [R,4C] blobs: boxes + 4, strides 4 and 4C; scores + 1,
 * strides 1 and C).  Rows with score <= score_thresh (or NaN) take no part.  cython_nms semantics (MI_NMS_GE_ORIG_ASC:
 * suppress at IoU >= nms_thresh, equal scores: higher row first).  kept [num_segments, rows] int32: 1 where the row
 * survives, 0 elsewhere (every element is written); num_keep [num_segments]: the survivors of each segment.
 * rows <= 4096.  The kept rows of a segment in ascending row order are `dets_j[keep, :]` of the reference.
 * workspace: mi_nms_segmented_workspace_bytes(num_segments, rows), 16-byte aligned, device. */
size_t mi_nms_segmented_workspace_bytes(int num_segments, int rows);
int mi_nms_segmented(const float* boxes, long long box_segment_stride, long long box_row_stride, const float* scores,
                     long long score_segment_stride, long long score_row_stride, int num_segments, int rows,
                     float score_thresh, float nms_thresh, int32_t* kept, int32_t* num_keep, float* masked_scores,
                     void* workspace, size_t workspace_bytes, mi_stream_t stream);
/* masked_scores (may be NULL) [num_segments, rows]: the score of every surviving row, -inf elsewhere -- the array the
 * detections_per_im cut ranks (below). */

/* The detections_per_im cut and the final gather of box_results_with_nms_and_limit (core/test.py:776-790) for hard NMS:
 * `masked_scores` from mi_nms_segmented (segments = classes 1.., rows = RoIs); top_values / top_indices = its `cap`
 * best entries in descending order (mi_topk_batched), cap <= 1024.  image_thresh = the detections_per_im-th best; every
 * surviving row at or above it is a detection (ties included).  Writes, in the reference's row order (class-major,
 * RoI-ascending inside a class): dets [cap,5] (x1,y1,x2,y2,score from boxes [rows, 4*num_classes] / scores [rows,
 * num_classes]), cls [cap] (class index, 0 for unused rows), sizes [1 + num_classes] int64 = (rows delivered, rows the
 * reference returns -- larger only when more than cap - detections_per_im scores tie at the cut --, then the detections
 * of classes 1 .. num_classes-1).  Unused rows are zero. */
int mi_detection_select(const float* scores, const float* boxes, const float* masked_scores, const float* top_values,
                        const int64_t* top_indices, int rows, int num_classes, int cap, int detections_per_im,
                        float* dets, int32_t* cls, int64_t* sizes, mi_stream_t stream);

/* Sorted top-k of float32 arrays, independent problems side by side (one workgroup each): replaces np.argsort /
 * np.argpartition of lib/modeling/generate_proposals.py:131-142 (pre-NMS top-k of a level's scores),
 * collect_and_distribute_fpn_rpn_proposals.py:85-86 (post_nms_topN of the collected levels) and np.sort of
 * core/test.py:781-784.  Problem p: values[p] [n[p]] -> out_values[p] [k[p]] descending, out_indices[p] [k[p]] int64;
 * equal values: lower index first (the reference's order of ties is undefined); NaN ranks below -inf.
 * 0 <= k <= n <= 2^24, k <= 4096.  The pointer / size arrays are HOST arrays.  A problem of more than 32768 values is
 * cut into chunks that run side by side and a merge of their winners (second launch); their candidates live in
 * `workspace` (device, 16-byte aligned, mi_topk_batched_workspace_bytes; may be NULL when no problem is that large).
 * The arrays are read with 16-byte loads from the 16-byte granules that contain them: up to 12 bytes in front of
 * values[p] and behind its last element are READ (never used) -- they must be readable memory, which any device
 * allocation and any row of a larger tensor provide. */
size_t mi_topk_batched_workspace_bytes(int num_problems, const int* n, const int* k);
int mi_topk_batched(int num_problems, const float* const* values, const int* n, const int* k, float* const* out_values,
                    int64_t* const* out_indices, void* workspace, size_t workspace_bytes, mi_stream_t stream);

/* Steps 6-8 of GenerateProposalsOp.proposals_for_one_image (lib/modeling/generate_proposals.py:155-161) for all
 * (level, image) problems and the concatenation of collect_and_distribute_fpn_rpn_proposals.py:83-90, one launch.
 * Problem p (HOST arrays of num_problems entries): dets[p] [k[p],5] / valid[p] [k[p]] from mi_rpn_decode_proposals,
 * keep[p] / num_keep[p] from mi_nms / mi_nms_batched in MI_NMS_GE_ORIG_ASC mode (keep == NULL or keep[p] == NULL: no
 * NMS), image[p] = its batch index.  A candidate is taken when it was kept, is valid, and is among the first
 * post_nms_topn (<= 0: all) such boxes of its problem.  Writes one row per candidate, problems back to back:
 * cand_scores [sum k] (score, or -inf when not taken) and cand_rois [sum k, 5] (image, x1, y1, x2, y2). k[p] <= 4096. */
int mi_rpn_collect_candidates(int num_problems, const float* const* dets, const int32_t* const* valid,
                              const int64_t* const* keep, const int32_t* const* num_keep, const int* k, const int* image,
                              int post_nms_topn, float* cand_scores, float* cand_rois, mi_stream_t stream);

/* The RoI blob of the heads from the global top-k over cand_scores (mi_topk_batched): rois [rows,5] = cand_rois[
 * top_indices], valid [rows] (uint8: the row is a proposal, i.e. its score is above -inf), levels [rows] int32 = the FPN
 * level of utils/fpn.py:11-28 (floor(canonical_level + log2(sqrt(area) / canonical_scale + 1e-6)) in fp32, clamped to
 * [k_min, k_max]).  mark_invalid != 0: rows that are no proposals get image index -1 (the RoI operators pool zeros for
 * them).  Replaces collect (:91-98) and the level half of distribute (:101-119). */
int mi_rpn_collect_finish(const float* top_scores, const int64_t* top_indices, const float* cand_rois, int rows,
                          int mark_invalid, int k_min, int k_max, float canonical_scale, float canonical_level,
                          float* rois, uint8_t* valid, int32_t* levels, mi_stream_t stream);
/* The map index of every row of a pyramid RoI blob that arrives as the reference's data layer ships it -- per-level blobs
 * `rois_fpn<l>` plus `rois_idx_restore_int32` (utils/fpn.py:31-58; consumed by modeling/model_builder.py:296-303, which
 * concatenates the pooled levels and gathers them back): row i of the blob in dataloader order lies at position
 * restore[i] of the level-major concatenation, whose spans have span_rows_host[k] rows (HOST array, num_spans <= 8,
 * level-major); out[i] = span_value_host[k] of the span that holds it.  restore: int32 or int64 device array.  One launch
 * in place of the slice fills + gather the mirror of roi_feature_transform used to issue (17 us of 84). */
int mi_fpn_level_index_from_restore(const void* restore, int restore_is_int64, int rows, int num_spans,
                                    const int* span_rows_host, const int* span_value_host, int32_t* out,
                                    mi_stream_t stream);
/* mi_rpn_collect_finish that ALSO leaves the RoIAlign records of the blob it writes (same outputs, bit for bit) -- the
 * producer of the RoIs holds every one of them in a launch of its own, so the per-RoI records of the box head's pyramid
 * call (sweep rank, window, tap tables: what mi_roi_align_forward_fpn's first launch computes) are written here and
 * mi_roi_align_forward_fpn_records starts with its gather kernel.  `levels`: the maps the head will pool from, COARSEST
 * FIRST (k_max .. k_min, modeling/FPN.py:76-78), so that RoI r pools from map k_max - roi_fpn_levels[r] -- the roi_levels
 * vector to pass on is k_max - roi_fpn_levels.  batch / channels / aligned size / sampling ratio / layout / workspace: those
 * of the forward call that will follow (mi_roi_align_fpn_supported(...) must be 1).
 * Replaces collect (:91-98), the level half of distribute (:101-119) and the record launch of the RoIAlign that follows. */
int mi_rpn_collect_finish_records(const float* top_scores, const int64_t* top_indices, const float* cand_rois, int rows,
                                  int mark_invalid, int k_min, int k_max, float canonical_scale, float canonical_level,
                                  float* rois, uint8_t* valid, int32_t* roi_fpn_levels, const mi_fpn_levels* levels,
                                  int batch, int channels, int aligned_height, int aligned_width, int sampling_ratio,
                                  int layout, void* workspace, size_t workspace_bytes, mi_stream_t stream);

/* Bounding-box voting (lib/utils/boxes.py:268-317 box_voting; call site lib/core/test.py:766-773, TEST.BBOX_VOTE): every
 * row of top_dets [num_top, 5] (x1, y1, x2, y2, score: the detections that survived NMS) is refined by the rows of its
 * class in all_dets (every candidate above the score threshold, before NMS) whose IoU with it is >= thresh (IoU =
 * utils.cython_bbox.bbox_overlaps bit for bit): box = score-weighted average of the voters' boxes; score per
 * scoring_method: 0 ID (unchanged), 1 TEMP_AVG, 2 AVG, 3 IOU_AVG, 4 GENERALIZED_AVG, 5 QUASI_SUM (`beta` as in the
 * reference).  All classes in one call: all_dets is class-major with all_offsets int32 [num_segments + 1];
 * top_segments int32 [num_top] names the segment of every top row.  out [num_top, 5] (may not alias top_dets).
 * Asynchronous, no workspace.  Averages are accumulated in fp64 and rounded once (numpy: fp32 pairwise): 1e-5 relative. */
int mi_box_voting(const float* top_dets, const int32_t* top_segments, int num_top, const float* all_dets,
                  const int32_t* all_offsets, int num_segments, float thresh, int scoring_method, float beta, float* out,
                  mi_stream_t stream);

/* Mask R-CNN training targets from polygons (lib/roi_data/mask_rcnn.py:66-76: the per-RoI host loop over
 * lib/utils/segms.py:93-119 polys_to_mask_wrt_box = pycocotools 2.0 mask_util.frPyObjects + mask_util.decode,
 * common/maskApi.c rleFrPoly / rleDecode), every RoI of a step in one launch.  Ground truth as ragged arrays:
 *   poly_xy    float32 [points, 2]        the vertices (x, y) of all polygons, image coordinates (roidb['segms'])
 *   poly_start int32  [polygons + 1]      first vertex of every polygon
 *   inst_start int32  [num_instances + 1] first polygon of every instance (an instance = a list of polygons)
 * roi_inst int32 [num_rois] names the instance of every RoI (mask_rcnn.py:62 fg_polys_inds; < 0 or >= num_instances: no
 * instance, the row is zero-filled); rois float32 [num_rois, 4] (x1, y1, x2, y2, image coordinates -- before `*= im_scale`,
 * mask_rcnn.py:99).  masks int32 [num_rois, m * m], row-major [y][x] as mask_rcnn.py:76 reshapes them: 1 where any polygon
 * of the instance, moved into the RoI's frame and scaled to m x m (float32, segms.py:104-112), covers the pixel by
 * pycocotools' rule.  m <= 64 (MI_ERR_UNSUPPORTED above).  Asynchronous, no workspace, bit-exact against the oracle's
 * sequential restatement of maskApi.c. */
int mi_polys_to_masks_wrt_boxes(const float* poly_xy, const int32_t* poly_start, const int32_t* inst_start,
                                const int32_t* roi_inst, const float* rois, int32_t* masks, int num_rois, int num_instances,
                                int m, mi_stream_t stream);

/* Independent NMS problems in one call (no reference counterpart: the reference runs one cython_nms per FPN level and
 * image on the host, modeling/generate_proposals.py:91-99,161).  `dets`, `n`, `keep`, `num_keep` are HOST arrays of
 * `num_problems` entries (device pointers / box counts); each problem follows the mi_nms contract, with at most 4096
 * boxes.  Every stage is one launch for all problems, so their greedy reduces run side by side.  Results are identical
 * to num_problems calls of mi_nms. */
size_t mi_nms_batched_workspace_bytes(int num_problems, const int* n);
int mi_nms_batched(int num_problems, const float* const* dets, const int* n, float thresh, int mode,
                   void* const* keep, int32_t* const* num_keep, void* workspace, size_t workspace_bytes,
                   mi_stream_t stream);

/* ---- IoU matrix --------------------------------------------------------------------------
 * reproduces utils.cython_bbox.bbox_overlaps (lib/utils/cython_bbox.pyx:32-73):
 * boxes [N,4], query [K,4] -> overlaps [N,K] float32, "+1" convention, 0 where iw<=0 or ih<=0. */
int mi_bbox_overlaps(const float* boxes, int num_boxes, const float* query, int num_query,
                     float* overlaps, mi_stream_t stream);

/* ---- frozen-BatchNorm chain of the ResNet bottleneck in one pass ---------------------------------------------------
 * replaces, for float32 tensors, the element-wise sequence the reference builds from torch ops:
 *   AffineChannel2d.forward (lib/nn/modules/affine.py:15-17)   x * weight[c] + bias[c]
 *   followed by ReLU (lib/modeling/ResNet.py:270-277) or by "out += residual; relu" (:284-286), or by nothing (the
 *   projection shortcut, :191-199).
 * forward:  y = relu?(x * weight[c] + bias[c] (+ residual)), operations in that order, no FMA contraction: bit-identical
 *           to the unfused chain.  residual may be NULL.  y may alias x.
 * backward: grad_x = grad_y * [y > 0]? * weight[c]; grad_residual (may be NULL) = grad_y * [y > 0]?.  `y` (the forward
 *           output) is only read when relu != 0.  weight / bias are frozen in every reference configuration
 *           (ResNet.py:76-77): no gradient is produced for them.
 * layout: MI_LAYOUT_NCHW (channel = (i / (H*W)) % C) or MI_LAYOUT_NHWC (channel = i % C, C % 4 == 0).
 * All tensors dense and 16-byte aligned. */
int mi_affine_channel_forward(const float* x, const float* weight, const float* bias, const float* residual, float* y,
                              int batch, int channels, int height, int width, int relu, int layout,
                              mi_stream_t stream);
int mi_affine_channel_backward(const float* grad_y, const float* y, const float* weight, float* grad_x,
                               float* grad_residual, int batch, int channels, int height, int width, int relu,
                               int layout, mi_stream_t stream);

/* ---- test-time result formats -------------------------------------------------------
 * mi_mask_paste_rle: one call = the per-detection body of segm_results (lib/core/test.py:807-844) for `num_masks`
 * detections: masks [num_masks, M, M] float32 (the class channel already selected, :813-816), boxes [num_masks, 4] int32
 * = `expand_boxes(ref_boxes, (M + 2) / M).astype(np.int32)` (:803-805).  Each mask is zero-padded by one pixel, resized to
 * the box with cv2.resize's INTER_LINEAR arithmetic for float32, binarised with `> thresh` (cfg.MRCNN.THRESH_BINARIZE),
 * pasted into an im_height x im_width image and run-length encoded like pycocotools.mask.encode (column-major runs, the
 * first run counts zeros).  counts [num_masks, capacity] uint32 receives the run lengths of detection d at row d,
 * num_counts [num_masks] their number -- ALWAYS the true number: when it exceeds `capacity` the row is unspecified and the
 * caller repeats the call with a larger capacity.  strings (may be NULL) [num_masks, string_capacity] receives COCO's
 * compressed ASCII form of the run lengths (maskApi.c rleToString), num_bytes [num_masks] its true length (larger than
 * string_capacity: repeat with more room; a row whose run lengths did not fit reports INT32_MAX).  M + 2 <= 64.
 * mi_keypoint_decode: heatmaps_to_keypoints (lib/utils/keypoints.py:106-157): heatmaps [num_rois, K, H, H] float32 logits,
 * rois [num_rois, 4]; every map is resized to (ceil(width), ceil(height)) of its RoI (at least min_size when > 0) with
 * cv2.resize's INTER_CUBIC arithmetic and reduced to xy_preds [num_rois, 4, K] = (x, y, logit, softmax probability over
 * the resized map) of its first maximum.  H <= 64.
 * OpenCV / pycocotools are third-party packages of the reference; their published algorithms are restated
 * (csrc/results.hip, oracle/results.py). */
int mi_mask_paste_rle(const float* masks, const int32_t* boxes, int num_masks, int mask_size, int im_height, int im_width,
                      float thresh, int capacity, uint32_t* counts, int32_t* num_counts, int string_capacity,
                      uint8_t* strings, int32_t* num_bytes, mi_stream_t stream);
int mi_keypoint_decode(const float* heatmaps, const float* rois, int num_rois, int num_keypoints, int heatmap_size,
                       int min_size, float* xy_preds, mi_stream_t stream);
/* OKS-NMS of the keypoint predictions of one image (lib/utils/keypoints.py:225-266 nms_oks / compute_oks; call site
 * lib/core/test.py:857-862, cfg.KRCNN.NMS_OKS): xy_preds [num_rois, 4, 17] as mi_keypoint_decode writes them, rois
 * [num_rois, 4]; persons are visited by descending mean keypoint logit (numpy's fp32 mean; equal means: higher index
 * first) and one is dropped when its OKS with respect to a kept one exceeds `thresh` (a double, as the reference's Python
 * float: the OKS is evaluated in fp64 exactly as numpy promotes it).  keep int64 [num_rois] receives the kept indices in
 * visiting order, *num_keep their number.  num_keypoints must be 17 (the reference's sigma table), num_rois <= 512. */
int mi_keypoint_nms_oks(const float* xy_preds, const float* rois, int num_rois, int num_keypoints, double thresh,
                        int64_t* keep, int32_t* num_keep, mi_stream_t stream);

/* ---- diagnostics (no reference counterpart) ---------------------------------------------------
 * Tuning aid used by tools/timeline_records.py / timeline_nhwc.py / timeline_bwd.py: while a non-NULL device buffer of 8 int64 per forward workgroup is set,
 * the RoIAlign forward kernels stamp the shader clock at their phase boundaries into it (layout per kernel: see the tools). */
void mi_dbg_roi_align_timeline(long long* device_buffer);
/* The MI_ROI_ALIGN_* tuning variables (csrc/common.h) are read from the environment ONCE, at the first RoIAlign call, and
 * never on the launch path.  Tests and tuning scripts that change them inside a process make the change visible with this
 * call -- the only writer of that state; call it with no RoIAlign launch in flight on any thread. */
void mi_dbg_reload_tuning(void);
/* Measurement aid of bench.py (roofline.copy_ceiling): a plain streaming copy of `bytes` (a multiple of 16; both buffers
 * 16-byte aligned) with 16 bytes per lane, non-temporal -- the box's own ceiling the RoIAlign roofline fractions are also
 * quoted against. */
int mi_dbg_copy_float4(const void* src, void* dst, size_t bytes, mi_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* MI_DETECTRON_OPS_H_ */
