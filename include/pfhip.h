/*
 * pfhip.h — C ABI of libpfhip.so: the MI355X (gfx950) device path of the background
 * forecasting hot path of nianticlabs/panoptic-forecasting.
 *
 * The reference has no FFI of its own for this path: its "native boundary" is
 * torch.ops.torch_scatter.scatter_min + ATen/cuDNN, reached from
 *   PCTransformModel.predict  panoptic_forecasting/models/pc_transform/pc_transform_model.py:26-150
 *   BGModel.predict           panoptic_forecasting/models/bg/bg_model.py:91-102
 *   hardnet.forward           panoptic_forecasting/models/bg/hardnet.py:353-387
 * Each entry point below names the reference lines it replaces.  The Python
 * shims that bind them (ctypes) are panoptic-forecasting_amd/{lib,pc_transform_model,bg_model}.py;
 * INTEGRATION.md shows the stub a reference maintainer would add.
 *
 * Conventions
 *  - every buffer is a caller-owned DEVICE pointer (tensor.data_ptr()); C-contiguous;
 *  - `stream` is a hipStream_t passed as void* (torch.cuda.current_stream().cuda_stream);
 *  - calls only enqueue work on `stream`: no allocation, no host sync, no hidden copies
 *    (plan create/destroy excepted) — they may be captured into a hipGraph;
 *  - every function returns 0 on success or a negative PF_E* code; pf_last_error() gives
 *    the thread-local message;
 *  - one in-flight call per (workspace, stream); plans are immutable after creation and
 *    may be shared between threads/streams.
 */
#ifndef PFHIP_H
#define PFHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PF_OK 0
#define PF_EINVAL (-1)    /* bad argument */
#define PF_EWORKSPACE (-2) /* workspace too small */
#define PF_EBLOB (-3)     /* malformed weight blob */
#define PF_EHIP (-4)      /* HIP runtime error (launch, alloc in plan create) */
#define PF_EUNSUPPORTED (-5)

typedef struct pf_plan pf_plan;

/* ABI version (major*1000 + minor). */
int pf_version(void);
/* Thread-local description of the last failure on this thread ("" if none). */
const char *pf_last_error(void);

/* ------------------------------------------------------------------------------------------
 * Warp + z-buffered splat  — replaces pc_transform_model.py:41-150 (unproject :41-59, cam->vehicle
 * :63, ego warp :68, vehicle->cam + project :71-78, validity :83-89, sentinel :105, 4-corner
 * bins :106-117, torch_scatter.scatter_min :118-119, winner gather :120-139).
 *
 *   depth      [B,T_total,H,W] f32        depth_mask [B,T_total,H,W] u8 (0/1)
 *   seg        [B,T_total,H,W,C] u8       C = seg_channels (1, or 3 for is_img)
 *   Kinv,K     [B,3,3] f32   E,Einv [B,4,4] f32   T_tgt [B,T_total,4,4] f32
 *     (Kinv/Einv are the caller's torch.inverse(K)/torch.inverse(E): LAPACK bits decide floor())
 *   frames t_first .. t_first+T-1 are warped (only_this_ind => t_first=ind, T=1; else 0, T_total).
 *   per_frame is a bit set (PF_SPLAT_*):
 *   PF_SPLAT_PER_FRAME clear: all T frames share ONE z-buffer (reference semantics of a single predict call)
 *                  out_seg [B,H,W,C] u8, out_depth [B,H,W] f32
 *   PF_SPLAT_PER_FRAME set:   every frame gets its own z-buffer and its own sentinel, i.e. T independent
 *                  only_this_ind=t calls in one launch: out_seg [B,T,H,W,C], out_depth [B,T,H,W]
 *   PF_SPLAT_PER_SAMPLE_SENTINEL set: the sentinel max+1 (:105) is taken per sample instead of over the whole batch
 *                  of the call (the reference's `.max()` spans the batch, so the depth written at holes won by
 *                  invalid points depends on which samples share a predict call; with this bit a sample's output is
 *                  what the reference gives for it at batch size 1 — SURVEY.md 8e opt-in)
 *   out_result2d (nullable) [B,T,H,W,2] i64: clamped (x,y) of the floor/floor corner (:147)
 *
 * Winner rule: minimum depth; ties -> lowest source element index e = r*P + t*N + n (r = corner
 * replica 0..3, P = T*N), which is pytorch_scatter's CPU rule.  Bins reached only by invalid
 * points yield seg=0, depth=max+1 (:105,:133); untouched bins seg=0, depth=-1 (:136-138).
 * Out of contract: non-finite projected coordinates, |z| >= 2^24.
 */
#define PF_SPLAT_PER_FRAME 1
#define PF_SPLAT_PER_SAMPLE_SENTINEL 2
int pf_warp_splat_workspace(int B, int T, int H, int W, int per_frame, size_t *bytes);
int pf_warp_splat(const float *depth, const uint8_t *depth_mask, const uint8_t *seg, int seg_channels,
                  const float *Kinv, const float *E, const float *T_tgt, const float *Einv,
                  const float *K, int B, int T_total, int t_first, int T, int H, int W, int per_frame,
                  uint8_t *out_seg, float *out_depth, int64_t *out_result2d,
                  void *ws, size_t ws_bytes, void *stream);

/* ------------------------------------------------------------------------------------------
 * FC-HarDNet-70 bg network — replaces bg_model.py:53-71,91-102 and hardnet.py:353-387.
 *
 * The plan is built from a weight blob (layout: panoptic-forecasting_amd/packing.py — op table +
 * BN-folded fp32 OIHW weights).  Plan creation uploads/re-tiles the weights (allocates device
 * memory once); forward calls allocate nothing.
 */
int pf_hardnet_plan_create(const void *blob_host, size_t bytes, int in_ch, int n_cls, pf_plan **out);
void pf_hardnet_plan_destroy(pf_plan *plan);
int pf_hardnet_workspace(const pf_plan *plan, int B, int H, int W, size_t *bytes);

/* Status of a forward.  The first PF_WS_STATUS_BYTES of a forward's workspace are its status block.  Word 0 is the status
 * word of the LAST FINISHED forward: written once, by the launch that ends a forward (its kernels collect their flags in a
 * private word of the block), so it is final once the stream has run the forward and stays readable while the next one runs.
 * Word 1 (PF_WS_STICKY_OFFSET) is STICKY: every forward ORs its final status word into it and only the host clears it
 * (pf_hardnet_status_sticky(clear = 1), pf_hardnet_status_reset) - the word to read after a hipGraph replay or an eager loop
 * that ran several forwards through one workspace.  The caller ZEROES THE BLOCK ONCE after allocating the workspace
 * (pf_hardnet_status_reset, or a memset): forwards do not start with a memset, the ending launch leaves the block ready for
 * the next one.
 *   PF_STATUS_RANGE      a tensor that feeds the two-term fp16 operand path (option "split_f16") held a value with
 *                        |x| > 65504 (inf included; NaN in a caller-provided input or produced by the stem): fp16 pairs cannot
 *                        represent it, so the outputs of THIS forward are not to be used.
 *   PF_STATUS_RANGE_LOW  the largest |x| a launch stored into such a tensor was non-zero and below 2^-6: the pair keeps an
 *                        absolute 2^-25 below |x| = 0.25, so a tensor of tiny values would lose relative precision that fp32
 *                        keeps.  (Unflagged: every operand within 2^-23 |x| + 2^-25 and 2^-25 <= 2^-19 of the tensor's maximum.)
 *                        Plan creation already stores every channel multiplied by a power of two that puts its expected
 *                        magnitude at 8 (an exact re-parameterisation of the folded weights, option "normalize_ranges"), so this
 *                        fires only for data far from what the weights suggest.
 * Either bit: the reference's fp32 Conv2d (hardnet.py:16-25) has no such limits - re-run the forward with plan option
 * "split_f16" = 0 (what BGModel does by default, bg_model.py `on_range_overflow`) or fail.  Never raised by fp32-only plans.
 * A caller that manages streams itself reads the words with its outputs (ordinary device memory; BGModel copies them to pinned
 * host memory behind every forward and checks them lazily); pf_hardnet_status / pf_hardnet_status_sticky are the convenience
 * forms: they copy a word to the host and SYNCHRONISE `stream` (the only calls of this library that wait for the device).
 * pf_hardnet_range_maxima: max |stored value| per op of the plan's table in the last forward (0 for ops that keep none), a
 * diagnostic for the guard (synchronises). */
#define PF_STATUS_RANGE 1u
#define PF_STATUS_RANGE_LOW 2u
#define PF_WS_STATUS_OFFSET 0
#define PF_WS_STICKY_OFFSET 4
#define PF_WS_STATUS_BYTES 2048
int pf_hardnet_status(const void *ws, unsigned *status, void *stream);
int pf_hardnet_status_sticky(void *ws, unsigned *status, int clear, void *stream);
int pf_hardnet_status_reset(void *ws, void *stream);
int pf_hardnet_range_maxima(const pf_plan *plan, const void *ws, float *maxima, int cap, int *n_ops, void *stream);

/* hop_flags: emulate the reference's on-disk hop between the two tasks on the fly */
#define PF_HOP_NONE 0
#define PF_HOP_TRAINID_LUT 1 /* seg ids -> trainIds (export_cityscapes_segmentation_results.py:34-38) */
#define PF_HOP_DEPTH_U16 2   /* depth -> round(clamp(d+1,0,255)*256) u16 (:119-124) -> x/256-1, mask=d>0,
                                d[~mask]=-1, clamp to [min_depth,max_depth] (bg_dataset.py:224-230,166-170);
                                depth_mask argument is ignored (may be NULL) */

/*
 * Fused forward: one-hot of T label maps (labels >= n_cls -> zero vector, bg_model.py:53-59) +
 * normalised masked depth (:50-51,:66-68) feed the stem conv directly (the [B,36,H,W] tensor is never
 * materialised), then hardnet (hardnet.py:353-371), bilinear align_corners upsample to (out_h,out_w)
 * (:372-384) and argmax (bg_model.py:98).
 *   seg   [B,T,H,W] u8 or i64 (seg_is_i64)      depth [B,T,H,W] f32     depth_mask [B,T,H,W] u8
 *   out_seg [B,out_h,out_w] u8 or i64 (out_seg_is_i64)
 *   out_logits (nullable) [B,n_cls,out_h,out_w] f32; out_orig_logits (nullable) [B,n_cls,H/4,W/4] f32
 */
int pf_bg_forward(const pf_plan *plan, const void *seg, int seg_is_i64, const float *depth,
                  const uint8_t *depth_mask, float depth_mean, float depth_std, int hop_flags,
                  float min_depth, float max_depth, int B, int T, int H, int W, int out_h, int out_w,
                  void *out_seg, int out_seg_is_i64, float *out_logits, float *out_orig_logits,
                  void *ws, size_t ws_bytes, void *stream);

/* Dense-input forward (convert2onehot=False configurations, bg_model.py:61-71): x [B,in_ch,H,W] f32. */
int pf_hardnet_forward_dense(const pf_plan *plan, const float *x, int B, int H, int W, int out_h,
                             int out_w, void *out_seg, int out_seg_is_i64, float *out_logits,
                             float *out_orig_logits, void *ws, size_t ws_bytes, void *stream);

/* ------------------------------------------------------------------------------------------
 * The on-disk hop between the reference's tasks, device side.  Elementwise, HBM-bound.
 *
 * pf_hop_export — what export_results applies to a prediction before the PNG encoder
 * (experiments/export_cityscapes_segmentation_results.py):
 *   seg [n] u8 or i64 (nullable) -> out_seg [n] u8; seg_mode 0 = as is (--no_convert), 1 = trainId -> label id
 *   (convert_labels :27-32, values outside 0..18 -> 0), 2 = label id -> trainId (convert_labels_to_trainid :34-38,
 *   ids outside the table -> 0);  depth [n] f32 (nullable) -> out_depth_u16 [n] = round(clamp(d+1,0,255)*256) (:119-121).
 * pf_hop_load — what BGDataset.__getitem__ applies to the u16 depth it reads (data/datasets/bg_dataset.py:224-228,
 *   166-170): d = q/256 - 1, mask = d > 0, d[~mask] = -1, masked values clamped to [min_depth, max_depth].
 * All buffers 16-byte aligned device pointers.
 */
int pf_hop_export(const void *seg, int seg_is_i64, int seg_mode, const float *depth, size_t n, uint8_t *out_seg,
                  uint16_t *out_depth_u16, void *stream);
int pf_hop_load(const uint16_t *depth_u16, size_t n, float min_depth, float max_depth, float *out_depth,
                uint8_t *out_mask, void *stream);

/* ------------------------------------------------------------------------------------------
 * fg -> panoptic merge — replaces the pasting loops of FGModel.predict_panoptic
 * (models/fg/fg_model.py:548-588; panoptic_ids=1, clear_things=1) and FGModel.predict_semantics (:455-480; panoptic_ids=0,
 * clear_things=0) together with model_utils.paste_mask (models/fg/model_utils.py:30-57).
 *
 *   background    [B,H,W] u8 / i32 / i64 (bg_kind 0/1/2) trainId canvas, or NULL (canvas of 255, :517-518)
 *   bg_depth      [B,H,W] f32 or NULL;  bg_depth_mask [B,H,W] u8 or NULL (0 -> depth 1e9, :563-564)
 *   masks         [N,MH,MW] f32 mask PROBABILITIES (after the sigmoid of :532), all images concatenated
 *   boxes         [N,4] f32 (cx,cy,w,h), or (x0,y0,x1,y1) when box_is_ulbr
 *   inst_depth    [N] f32 forecast depth per instance (required with use_depth_sorting)
 *   classes       [N] i64 thing class (0..7);  inst_offsets [B+1] i32: image b owns instances [off[b], off[b+1])
 *   out           [B,H,W] i32 or i64
 * Paste order: descending depth, stable (ATen's CPU sort) when use_depth_sorting, else index order.  A pixel takes
 * an instance's value where its bilinearly pasted mask is >= 0.5 and - when bg_depth is given - its depth is strictly
 * nearer than the current one (:580-585); value = (class+11)*1000 + running per-class id (:568-572) or class+11.
 * The mask sampling reproduces F.grid_sample(bilinear, zeros, align_corners=False) of ATen's vectorised CPU kernel
 * bit for bit.  Unlike the reference, the caller's bg_depth tensor is not modified.  W % 4 == 0.
 * Workspace: pf_panoptic_merge_workspace(n_instances) bytes.
 *
 * pf_panoptic_encode — export_cityscapes_panoptic_results.py:27-68: ids converted to Cityscapes ids when
 * convert_to_ids (255 -> 0; v > 100: id(cat)*1000 + inst; else id(v)), out_rgb [B,H,W,3] = (id%256, id/256%256,
 * id/65536), out_ids (nullable) [B,H,W] i32, out_present [B, pf_panoptic_max_ids()] u8 = 1 for every id that occurs
 * (get_segments_info's np.unique).  H*W % 4 == 0.
 */
int pf_panoptic_merge_workspace(int n_instances, size_t *bytes);
int pf_panoptic_merge(const void *background, int bg_kind, const float *bg_depth, const uint8_t *bg_depth_mask,
                      const float *masks, int MH, int MW, const float *boxes, int box_is_ulbr,
                      const float *inst_depth, const int64_t *classes, const int32_t *inst_offsets,
                      int n_instances, int B, int H, int W, int use_depth_sorting, int panoptic_ids,
                      int clear_things, void *out, int out_is_i64, void *ws, size_t ws_bytes, void *stream);
int pf_panoptic_encode(const void *seg, int seg_is_i64, int convert_to_ids, int B, int H, int W, uint8_t *out_rgb,
                       int32_t *out_ids, uint8_t *out_present, void *stream);
int pf_panoptic_max_ids(void);

/* ------------------------------------------------------------------------------------------
 * pf_bg_dense_input - the network input of BGModel.forward (models/bg/bg_model.py:61-69) for the configurations the fused stem
 * of pf_bg_forward does not cover (convert2onehot = False, or no depth channels): writes x [B, T*C (+T), H, W] f32 for
 * pf_hardnet_forward_dense.
 *   frames   kind 0: labels u8 [B,T,H,W], kind 1: labels i64 [B,T,H,W] (one-hot over `channels` classes, labels >= channels ->
 *            zero vector, :53-59), kind 2: f32 [B,T,channels,H,W] (copied: the reshape of :63-64)
 *   depth    f32 [B,T,H,W] or NULL (use_depth_inps false); depth_mask u8 [B,T,H,W]: channel T*C + t = (depth - mean) / std * mask (:66-69)
 */
int pf_bg_dense_input(const void *frames, int kind, int channels, const float *depth, const uint8_t *depth_mask, float depth_mean,
                      float depth_std, int B, int T, int H, int W, float *x, void *stream);

/* ------------------------------------------------------------------------------------------
 * Validation loss of the bg model (forward part) - replaces BGModel.loss (bg_model.py:73-89):
 * nn.CrossEntropyLoss(ignore_index) of the bilinearly upsampled (align_corners, hardnet.py:372-384) logits and the
 * accuracy counters, fused: the full-resolution logits are never materialised.
 *   logits [B,C,Hin,Win] f32 (the network's orig_size_logits, C = 11 or 19)   labels [B,out_h,out_w] i64 or u8
 *   out3 (device) = { sum over valid pixels of -log softmax(logits)[label], #valid pixels, #pixels with argmax == label }
 *   => loss = out3[0]/out3[1], accuracy = out3[2]/out3[1].  Deterministic (fixed-order fp64 reduction).
 * Workspace: pf_seg_loss_workspace(B, out_h, out_w).  (The training step - this loss with its backward pass - is
 * pf_train_forward_backward below.)
 */
int pf_seg_loss_workspace(int B, int out_h, int out_w, size_t *bytes);
int pf_seg_loss(const float *logits, int B, int C, int Hin, int Win, const void *labels, int labels_i64, int out_h,
                int out_w, int ignore_index, double *out3, void *ws, size_t ws_bytes, void *stream);

/* Process-wide execution options (not thread-safe; set before launching work):
 *   "fuse_pool"     (default 1) a 1x1 conv followed by AvgPool2d(2,2) (hardnet.py:296) pools in the conv epilogue;
 *   "fuse_upsample" (default 1) TransitionUp + 1x1 conv over cat([up(x), skip]) (hardnet.py:248-258,365-368) is
 *                   evaluated as W_skip*skip + up(W_x*x): same result up to fp32 rounding, no upsampled tensor;
 *   "use_tuned_table" (default 1) per-layer kernel shapes come from the measured table (csrc/conv_tuned.inc) where it
 *                   has the shape, else from the cost model; 0 = cost model only (tools/tune_convs.py);
 *   "split_f16"     (default 1; "split_bf16", its name while the terms were bf16, is still accepted) the table may select
 *                   the split kernels (conv_split, conv_s4) for stride-1 3x3 and 1x1 layers: every fp32 operand split into
 *                   two fp16 terms hi + mid (22 significand bits; weights pre-scaled by an exact power of two per conv),
 *                   three products on v_mfma_f32_16x16x32_f16, fp32 accumulation - single layers within 2e-5*(1+max|ref|)
 *                   of fp64 like the fp32 kernels, whole-network logits within 1e-4 either way.  Operand bound: both terms
 *                   are rounded to nearest even, |x - hi - mid| <= 2^-23 |x| + 2^-25 for |x| <= 65504 (fp32 itself: 2^-24 |x|);
 *                   beyond 65504 the forward raises PF_STATUS_RANGE, and a tensor whose largest value is below 2^-6 (where the
 *                   absolute term would dominate) raises PF_STATUS_RANGE_LOW (above) - never a silent loss in either direction;
 *                   0 = every convolution on fp32 MFMA / fp32 VALU;
 *   "normalize_ranges" (default 1; process-wide only, read when a plan is created) every activation channel is stored
 *                   multiplied by a power of two chosen from the folded weights so that its expected magnitude is 8 (producer
 *                   rows * s, consumer columns / s: bit-identical fp32 arithmetic, fp16-pair range centred on the data; a
 *                   checkpoint re-parameterised across a BatchNorm runs on the same stored values); 0 = store raw values;
 *   "range_guard"   (default 1) the PF_STATUS_RANGE checks of the split path (a compare per stored value); 0 removes them;
 *   "fuse_front"    (default 1) base.1 (3x3 s1, 16 -> 24) and base.2 (3x3 s2, 24 -> 32) as ONE kernel on a packed-pair stem output, the
 *                   tensor between them kept in LDS (csrc/conv_front.hip: workgroups march down 31-column strips): same results
 *                   (tests/test_gpu_bg_model.py), 39 % fewer front-end bytes, 0.42 ms against 0.83 ms for the two separate kernels
 *                   per 16 frames at 1024x2048; 0 = stem -> conv_split -> conv_dma stride 2;
 *   "profile_tag_ops" (default 0; process-wide only) pf_profile_* records carry one label per op of the table (tools/);
 *   "upsample_bwd_two_pass" (default 1; read by pf_train_forward_backward) the transposed bilinear interpolations of the training
 *                   step (the loss head's 4x above all) as a row pass and a column pass over a scratch tensor when the planes are
 *                   large - the one-pass kernel's own nesting and order, so the same bits; 0 = always one pass;
 *   "train_side_stream" (default 1; process-wide only, read by pf_train_create) the weight gradients of the training step - leaves
 *                   of the backward pass - run on the training plan's own lower-priority stream, forked from and joined to the
 *                   caller's stream inside every pf_train_forward_backward (a stream capture of the call stays one graph; the
 *                   results are bit-identical, the workspace grows by one dy slot); 0 = everything on the caller's stream;
 *   "train_table_batch" (default 0; process-wide only, read by pf_train_forward_backward) n > 0: the measured shape table of the
 *                   training step's convolutions (csrc/train_tuned.inc, keyed on the batch it was measured at) is consulted as if
 *                   the batch were n - the kernels of the timed configuration (n = 8) on a batch a CPU checker can afford;
 *   "fuse_pairs"    (default 1) an odd HarDBlock layer computed INSIDE its consumer where both read / write packed pairs
 *                   (csrc/conv_pair.hip: one launch per pair, the shared input staged once, the odd layer kept in LDS planes; the
 *                   (<= 12) -> (17..20)-channel pairs in the MERGED form, the odd layer's weights in the rows the consumer's second
 *                   cout tile pads with zeros): 1 = where it measured faster (launches of 65k-196k pixels), 0 = never, 2 = wherever
 *                   the kernel exists, 3 = everywhere and never merged, 4 = the merged pairs only (A/B runs, tests); results agree
 *                   with the two launches within 1e-5 (1 + max) (tests/test_gpu_conv.py);
 *   "train_blocked_sum" (default 1; process-wide only) the 3x3 convolutions of a training step add every round of 8 input channels
 *                   into a second accumulator set (blocked summation, like ATen's): forward activations 0.79-0.95 x as far from
 *                   float64 as torch-CPU fp32; 0 = one fp32 chain over all 9 Cin terms;
 *   "train_forward_s4" (default 0; process-wide only, read by pf_train_create) 1 = the forward 3x3 stride-1 and 1x1 conv +
 *                   BatchNorm layers of a training step run on the packed-pair kernels of the inference path (csrc/train_s4.hip:
 *                   weights packed on the device every step with the fixed scale 2^12 - a weight of magnitude >= 16 becomes NaN,
 *                   loudly - activation slices shadowed as fp16 pairs, fp32 outputs; the 3x3 layers on conv_s4_blocked_kernel, which
 *                   sums every round of 8 input channels on its own like the fp32 step): the step 14.45 -> 14.24 ms at batch 8 of
 *                   800x800, the forward pass 0.81-0.95 x as far from float64 as torch-CPU fp32 (fp32 step: 0.79-0.95 x; without
 *                   the blocked sums 14.0 ms and 1.0-1.2 x); 0 = conv_dma on the fp32 matrix instruction;
 *   "wgrad_taps"    (default 1; process-wide only) weight gradients of 3x3 stride-1 layers with the taps folded into the matrix rows
 *                   (csrc/wgrad_taps.hip: rows = (cout, tap) pairs, 10 outputs fill 90 of 96 rows instead of 10 of 16): 1 = per layer
 *                   where it measured faster, 0 = never (wgrad_tiled_kernel), 2 = wherever the kernel exists; same fixed-order
 *                   partial sums either way (a step stays bit-reproducible), other rounding than the tiled kernel's;
 *   "valu_remainder" (default 1) trailing cout % 16 <= 8 channels of a conv_dma layer on the vector ALU. */
int pf_set_option(const char *name, int value);
/* The same options per plan: a plan copies the process-wide values when it is created; this call changes them for
 * that plan only (read at forward time by forwards of that plan; a plan is used from one thread at a time).  One more
 * name exists only here:
 *   "table_batch"   (default 0) n > 0: the per-layer kernel table is consulted as if every forward had a batch of n —
 *                   a frame's logits then do not depend on how many frames share the call (the tuned table is keyed on
 *                   the batch size, and the split and fp32 kernels differ in the last bits). */
int pf_hardnet_plan_set_option(pf_plan *plan, const char *name, int value);

/* ------------------------------------------------------------------------------------------
 * bg training step (scope row f4) — replaces, per batch, training/train.py:185-222 for task `bg`:
 *   loss_dict = model.loss(inputs, labels)  (bg_model.py:73-89: train-mode BatchNorm, F.interpolate align_corners,
 *   CrossEntropyLoss(ignore_index=255), accuracy) ; loss.backward() ; clip_grad_value_/clip_grad_norm_ ; SGD step.
 *
 * A pf_train is built from the same blob as a plan (only the op table is used).  Parameters are ONE flat fp32 device
 * array `theta`; per conv op of the table, in table order:  W[cout][cin][k][k], then gamma, beta, running_mean,
 * running_var [cout each] for conv+BatchNorm layers (hardnet.py:16-25) or bias[cout] for a plain conv (finalConv).
 * `grad` has the same layout (the running-stat slots stay 0).  pf_train_param_layout gives the offsets of one op.
 *
 * pf_train_forward_backward: forward in training mode (batch statistics; running stats updated in theta when
 * update_running_stats), loss, backward.  Inputs either as the reference batch (seg [B,T,H,W] u8/i64 trainIds, depth,
 * depth_mask, depth_mean/std: one-hot + normalised masked depth built on the device, bg_model.py:53-69) or as a dense
 * x_dense [B,in_ch,H,W].  labels [B,out_h,out_w] u8/i64.  grad is zeroed first unless accumulate_grads (train.py's
 * accumulate_steps; loss_scale = 1/accumulate_steps multiplies the gradient).  out3 (device, 3 doubles) = {sum of the
 * per-pixel losses, number of labels != ignore_index, number of correct argmax}: loss = out3[0]/out3[1], accuracy =
 * out3[2]/out3[1].  Only enqueues on `stream`; every reduction runs in a fixed order (bit-reproducible).
 *
 * pf_sgd_step: g *= min(1, clip_norm/(||g||+1e-6)) if clip_norm > 0 (clip_grad_norm_); clamp to +-clip_value if > 0
 * (clip_grad_value_); g += weight_decay*p; buf = first_step ? g : momentum*buf + g; p -= lr*buf (torch.optim.SGD) on
 * the elements with trainable[i] != 0 (the running statistics are buffers, not parameters).  For data-parallel
 * training all-reduce `grad` (one RCCL call, ~16.5 MB) between the two calls (reference: DDP, train.py:96-103). */
typedef struct pf_train pf_train;
int pf_train_create(const void *blob, size_t blob_bytes, int in_ch, int n_cls, pf_train **out);
void pf_train_destroy(pf_train *t);
/* enable != 0: the workgroup shape of every forward / backward-data convolution is measured (all candidates, a few launches
 * each, hipEvents + a stream synchronisation) in a pass of its own in front of the first pf_train_forward_backward of a
 * configuration (B, H, W, out size) that runs outside a stream capture - that pass leaves theta, grad and the running statistics
 * alone - then kept for the life of the plan; 0 (default): the shapes of csrc/train_tuned.inc / the cost model, and the measured
 * ones are forgotten.  Call it before pf_train_workspace: the measuring pass needs n_params floats more.  The measured choice depends on
 * timing: with it, two runs need not pick the same summation order for the K-split shapes (results equal to rounding, not
 * to the bit, from run to run; within one plan they are reproducible once every geometry has been seen). */
int pf_train_autotune(pf_train *t, int enable);
/* what pf_train_autotune has measured so far: rows of 10 ints {ks, stride, Cin, Cout, Hin, Win, B, accumulate, pixel waves,
 * cout tiles} ({.., 0, 0} = the cost model's own shape won); *n_rows = rows available, at most cap_rows are written.
 * tools/tune_train.py turns them into csrc/train_tuned.inc, the table plans that do not measure consult (option "use_tuned_table"). */
int pf_train_tuned_shapes(const pf_train *t, int *rows, int cap_rows, int *n_rows);
/* which code paths the LAST pf_train_forward_backward of this plan took: stats[0] convolutions launched with a row of
 * csrc/train_tuned.inc, [1] with the cost model's shape, [2] with a shape pf_train_autotune measured, [3] odd-width convolutions
 * run on row-padded copies (gather), [4] of those, forward conv + BatchNorm layers whose output stayed in padded rows, [5]
 * odd-width layers whose input gradients came from ONE backward-data conv over all ranges, [6] launches of the generic
 * (register-staged) kernel, [7] forward convolutions run on the packed-pair kernels (option "train_forward_s4").  *n = 8 values
 * available, at most cap are written.  (Tests: the timed configuration's kernels ran.) */
int pf_train_path_stats(const pf_train *t, int *stats, int cap, int *n);
int pf_train_param_count(const pf_train *t, size_t *n_floats);
int pf_train_param_layout(const pf_train *t, int op_index, size_t *w_off, size_t *aux_off, int *has_bn);
int pf_train_workspace(const pf_train *t, int B, int H, int W, int out_h, int out_w, size_t *bytes);
int pf_train_forward_backward(const pf_train *t, float *theta, float *grad, int accumulate_grads, const void *seg, int seg_is_i64,
                              const float *depth, const uint8_t *depth_mask, float depth_mean, float depth_std, int T,
                              const float *x_dense, int B, int H, int W, const void *labels, int labels_i64, int out_h, int out_w,
                              int ignore_index, float loss_scale, float bn_momentum, float bn_eps, int update_running_stats,
                              double *out3, void *ws, size_t ws_bytes, void *stream);
/* where tensor `name` (activation, or its gradient with want_grad) lives in the training workspace (tests) */
int pf_train_tensor_view(const pf_train *t, const char *name, int want_grad, int B, int H, int W, int out_h, int out_w,
                         size_t *ws_offset, int *channels, int *h, int *w);
int pf_sgd_workspace(size_t *bytes);
int pf_sgd_step(float *theta, float *grad, float *momentum_buf, const uint8_t *trainable, size_t n, float lr, float momentum,
                float weight_decay, float clip_norm, float clip_value, int first_step, void *ws, size_t ws_bytes, void *stream);

/* Introspection for per-stage parity tests: where tensor `name` (packing.py tensor names, e.g.
 * "base.4.out") lives inside the workspace for this (B,H,W): byte offset, channels, height, width. */
int pf_hardnet_tensor_view(const pf_plan *plan, const char *name, int B, int H, int W,
                           size_t *ws_offset, int *channels, int *h, int *w);
/* Copy tensor `name` of the LAST forward of this plan out of its workspace as fp32 NCHW [B,channels,h,w].  Intermediate
 * tensors may live in the packed-pair layout of conv_s4.hip (plan option "packed_acts", on by default): two fp16 terms
 * hi = fp16(x), mid = fp16(x - hi) (round to nearest even) per element in [B][2][ceil(C/4)][H][W][4] order - this call undoes it. */
int pf_hardnet_tensor_read(const pf_plan *plan, const char *name, int B, int H, int W, const void *ws, float *dst, void *stream);
/* fp32 NCHW <-> packed-pair layout (dst of pf_s4_pack: 16 * B * ceil(C/4) * H * W bytes); tests and tensor taps.
 * status (nullable, device word): PF_STATUS_RANGE is raised when an element is not representable (|x| > 65504, NaN). */
int pf_s4_pack(const float *src, void *dst, int B, int C, int H, int W, unsigned *status, void *stream);
int pf_s4_unpack(const void *src, float *dst, int B, int C, int H, int W, void *stream);
/* Dense-equivalent FLOPs of one forward at (H,W) per sample (2*Cout*Hout*Wout*Cin*k*k summed). */
int pf_hardnet_flops(const pf_plan *plan, int H, int W, double *flops);

/* ------------------------------------------------------------------------------------------
 * Opt-in per-launch timing for the roofline report (bench.py).  While enabled, every kernel the
 * library enqueues is bracketed by hipEvents on ITS launch stream (process-wide state; do not
 * enable during graph capture).  pf_profile_collect() synchronises the recorded events and returns
 * the number of distinct kernels; pf_profile_get() returns, per kernel (label = the demangled
 * symbol rocprofv3 prints), launches, summed duration and the summed ALGORITHMIC flops / bytes of
 * those launches (SURVEY.md 8d accounting: unique input + output bytes; 2*Cout*H*W*Cin*k*k flops).
 */
int pf_profile_enable(int on);
int pf_profile_collect(void);
int pf_profile_get(int i, char *label, size_t label_cap, int *launches, double *total_ms, double *flops,
                   double *bytes);

/* Tuning/diagnostic hook (tools/tune_convs.py, tests): force the convolution kernel and tile shape of every
 * stride-1 conv the library launches from now on.  kind 0 = automatic (default); 1 = conv_dma (p0 = WM in
 * {1,2,4}, p1 = NT); 2 = conv_wave (p0 = rows per tile in {1,2,4}, p1 = NT in {1,2}, p2 = K-split waves in
 * {2,4,8,16}); 3 = conv_valu (3x3 only, p0 = rows per wave in {1,2}); 4 = conv_split (3x3 only, p0 = NT in {1,2,3},
 * p1 = 1 for 8x64-pixel tiles).  Shapes that are not built fall back to the automatic choice.  Process-wide, not
 * thread-safe. */
int pf_debug_force_conv(int kind, int p0, int p1, int p2);
/* Instrumented builds only (make libpfhip_probe.so, env PF_PROBE=1): the 64 in-kernel timestamps (shader clock)
 * written by workgroup 0 / wave 0 of the last conv_wave launch; PF_EINVAL when nothing was recorded. */
int pf_debug_probe_read(long long *host64);

#ifdef __cplusplus
}
#endif
#endif /* PFHIP_H */
