/*
 * gvl_msda.h -- C ABI of libgvl_msda.so: MI355X (gfx950) multi-scale deformable attention for GVL.
 *
 * This is the drop-in boundary for the reference's native extension `MultiScaleDeformableAttention`
 * (/root/reference/pdvc/ops/src/vision.cpp:13-16).  Each entry point names the reference interface it replaces.
 * Plain pointers and sizes only: no torch / ATen types.  All data pointers are DEVICE pointers unless the
 * parameter name ends in `_host`.  Every call is asynchronous on `stream` (a hipStream_t passed as void*;
 * NULL = the default stream), allocates nothing, never synchronises, and is safe to capture in a hipGraph.
 * Calls are stateless and re-entrant (autograd worker threads call the backward).
 *
 * Return value: 0 on success, otherwise a negative GVL_E* code (argument errors) or a positive hipError_t
 * (launch errors -- the reference only printf()s those, cuh:949-953; here they are returned).
 * gvl_last_error() returns a thread-local human readable message for the last failing call.
 *
 * Tensor layouts (row-major, contiguous, as in the reference op):
 *   value   (B, S, M, D)           S = sum_l H_l*W_l
 *   shapes  (L, 2) int64  (H_l, W_l);  for GVL's temporal features H_l = 1, W_l = T_l (ms_deform_attn.py:117)
 *   lsi     (L)    int64  first row of level l inside S
 *   loc     (B, Q, M, L, P, 2)     normalised (x, y) in [0,1]; GVL always passes y = 0.5 (ms_deform_attn.py:115)
 *   attn    (B, Q, M, L, P)
 *   out     (B, Q, M*D)
 *   sample  (B*M, D, Q, L, P)      unweighted samples, the layout of ms_deform_attn_core_pytorch(return_value=True)
 *
 * pad_mode: GVL_PAD_ZEROS  = semantics of the reference CUDA op (ms_deform_im2col_cuda.cuh:238-300, :407-511)
 *           GVL_PAD_BORDER = semantics of the reference's PyTorch fallback / captioner path
 *                            (ms_deform_attn_func.py:61-62, F.grid_sample(padding_mode='border')).
 *
 * shapes_host / lsi_host (optional, may be NULL): a HOST copy of `shapes` / `lsi`.  When given, and every level
 * has H_l == 1, fp32, D == 64 and L*P <= 16, the LDS-staged temporal kernels are used (all S rows in LDS up to
 * S = 639; beyond that -- long videos, T = 512 -> S = 960 -- level 0 is read from global memory and levels 1.. are
 * staged, which needs L*P == 16 and P == 4); otherwise the generic kernels (any D, any HxW, fp32/fp64) run.
 * Results are identical up to fp32 summation order.
 *
 * Element types: _f32 / _f64 entry points compute in the named type throughout.  _bf16 entry points (SURVEY.md
 * section 8(b) "bf16 twins"; BASELINE config 4, long videos under bf16 autocast) keep value / out / grad_out /
 * grad_value -- and, fused, proj / grad_proj -- as bfloat16 in HBM (passed as uint16_t bit patterns), while every
 * sampling location, reference point, attention weight, their gradients and ALL arithmetic stay fp32 (a bf16
 * location would quantise x*T_l to whole frames at T = 512); results are rounded to nearest-even once, on the
 * final store.  They are served by the temporal D = 64 kernels only and return GVL_EINVAL for other shapes.
 */
#ifndef GVL_MSDA_H
#define GVL_MSDA_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GVL_MSDA_ABI_VERSION 17
/* ABI history (newest first):
 * 17: + gvl_msda1d_fused_forward_shared_amax_f32 (one set of offsets / logits rows for every video: the first decoder layer in inference)
 * 16: + gvl_count_pool_f32 / gvl_count_pool_backward_f32 (the count head's pooling over the queries and its gradient, training),
 *      gvl_batch_sum_f32 (the batch-expanded query embedding's gradient), gvl_residual_dropout_layer_norm_backwardn_f32 /
 *      gvl_rdln_backward_max_grads (several output gradients summed in the load path), gvl_level_sums_f32 (the level embedding's
 *      gradient), gvl_lstm_cell_train_backward_sum_f32 (the gate gradients' running sum over the token steps),
 *      gvl_mask_rows_f32 / _backward_f32 (the padded rows of a value projection), gvl_linear_f16x3_splitk_bias_f32 (split-K with a bias, overlapping rows), gvl_wgrad_f16x3_live_f32 / gvl_wgrad_live_ints
 *      (a weight gradient that visits only the row stages with non-zero rows), gvl_mha_train_backward_amax_f32 /
 *      gvl_group_norm_rows_backward_amax_f32
 *      (row maxima of dqkv / of the pyramid's dy from the kernels that write them), gvl_index_add_rows_f32 (an embedding
 *      lookup's gradient without the additions of zero rows)
 * 15: + gvl_wgrad_group_f16x3_f32 / gvl_wgrad_group_workspace_bytes / gvl_wgrad_group_max (the weight gradients of several Linears
 *      in one launch), gvl_caption_rows (the captioner's pair rows on padded targets in one launch)
 * 14: gvl_adam_desc carries each tensor's OWN step pointer and gvl_clip_adam_step_f32 takes (n_tensors, corr) instead of one
 *      step pointer: torch.optim.Adam counts steps per parameter; max_norm is passed straight to clip_grad_norm_'s
 *      formula (0 scales the gradients to 0, as torch does)
 * 13: + gvl_cap_attend_pre_f32 / _applicable (the offsets' hidden-state product arrives precomputed; half of the sample reads
 *      from LDS)
 * 12: + gvl_reload_env (the GVL_* switches are read once per process, not on every launch), gvl_gemm_f16x3_gates_f32 /
 *      _applicable (both halves of the LSTM gate product + cell in one launch), gvl_greedy_step_partials_gemm_f32 (greedy
 *      reduction + an independent product in one launch), gvl_clip_adam_step_f32 (gradient clipping + Adam over a tensor table),
 *      gvl_group_norm_rows_backward_f32 / gvl_conv_taps_to_rows_f32 (training form of the base encoder's levels)
 * 11: + gvl_wgrad_f16x3_f32 / gvl_wgrad_workspace_bytes (weight + bias gradient of every nn.Linear of the training step),
 *      gvl_planes_refresh_f16 / gvl_planes_chunk_elems (operand planes of all weights, both orientations, two launches per
 *      step), gvl_mha_train_{forward,backward}_f32 (attention core of nn.MultiheadAttention in training),
 *      gvl_relu_dropout_rows_*
 * 10: + gvl_f16_products (one fp16 product per fp32 product: inference under autocast), operand planes K-stage-major, +
 *      gvl_residual_dropout_layer_norm_{forward,backward}_f32 / gvl_rdln_backward_blocks / gvl_advance_step /
 *      gvl_relu_dropout_{forward,backward}_f32 (training residual chains and FFN activation)
 *  9: + gvl_msda_last_kernel (diagnostic); - gvl_skinny_gemm_f16x3_f32 / gvl_skinny_pack_f16 (round 3's few-row product: slower
 *      than the tuned library kernels, removed); the temporal backward needs no workspace where the row-ownership form applies
 *  8: + gvl_gemm_f16x3_lstm_f32, gvl_cap_attend_split_levels_f32, gvl_ce_rows_*_f32
 *  7: + gvl_linear_f16x3_f32, gvl_layer_norm_rows_f32, gvl_row_absmax_f32, gvl_msda1d_fused_forward_amax_f32 (inference layers)
 *  6: + gvl_split_rows_f16, gvl_gemm_f16x3_f32, gvl_gemm_f16x3_argmax_f32, gvl_greedy_step_partials_f32,
 *      gvl_cap_attend_split_f32, gvl_lstm_cell_split_f32
 *  5: + padded (layout-independent) targets: gvl_match_cost_padded_f32, gvl_set_criterion_* take video_pair_count /
 *      num_boxes_dev, gvl_cap_attend_train_* take row_video; + gvl_proj_f32
 *  4: + gvl_col_sum_f32
 *  3: + training-time captioner step, matcher cost, set criterion
 *  2: + bf16 storage twins
 */

#define GVL_PAD_ZEROS 0
#define GVL_PAD_BORDER 1

#define GVL_EINVAL (-1)   /* bad size / null pointer / unsupported combination */
#define GVL_ENOSPC (-2)   /* workspace too small */

int gvl_msda_abi_version(void);
const char *gvl_last_error(void);

/* Force a kernel family for A/B testing: 0 = auto (default), 1 = generic only, 2 = fast where eligible.
 * (Also settable through the environment variable GVL_MSDA_IMPL=auto|generic|fast before first use.) */
void gvl_msda_set_impl(int impl);
/* Which family the most recent forward/backward call on this thread used: 1 generic, 2 fast, 3 fused. */
int gvl_msda_last_impl(void);
/* Name of the kernel form the most recent forward/backward call on this thread launched ("k_fwd_t1d_d64",
 * "k_bwd_t1d_split", "k_bwd_t1d_own", "k_bwd_t1d_d64", "k_bwd_t1d_d64<loop>", "k_fwd_generic", "k_bwd_generic"): lets a
 * test assert WHICH backward form served a shape.  Static storage; no reference equivalent. */
const char *gvl_msda_last_kernel(void);
/* The library reads its GVL_* environment switches (kernel-form overrides for A/B runs and tests: INTEGRATION.md) ONCE per process,
 * on first use; this drops what it has read, so that a change made with setenv() afterwards takes effect on the next call. */
void gvl_reload_env(void);

/* -- matcher cost matrix for ALL decoder layers in one launch: replaces the tensor-op sequence of
 *    HungarianMatcher.forward (pdvc/matcher.py:74-105 with misc/detr_utils/box_ops.py:8-47):
 *      cost[l,b,q,g] = w_bbox * L1(box, tgt_box) + w_class * (focal_pos - focal_neg)[tgt_label] + w_giou * (-GIoU_1D)
 *    pred_logits (nl,B,Q,NC), pred_boxes (nl,B,Q,2) (centre,length), tgt_labels (G) int64, tgt_boxes (G,2);
 *    cost (nl,B,Q,G).  Evaluated with one rounding per operation in the reference's order (no FMA contraction): the
 *    GIoU term is bit-identical to the PyTorch op sequence, the focal term within 1-2 ulp (device expf / logf).  *ok (device int, preset to 1 by the caller) is cleared when a box has
 *    x1 < x0 (the asserts of box_ops.py:39-40). */
int gvl_match_cost_f32(const float *pred_logits, const float *pred_boxes, const int64_t *tgt_labels,
                       const float *tgt_boxes, int n_layers, int B, int Q, int n_classes, int G, float w_class,
                       float w_bbox, float w_giou, float alpha, float gamma, float *cost, int *ok, void *stream);

/* The same cost in the PADDED target layout (what a captured hipGraph needs: no shape depends on how many events a
 * video has): every video owns `slots` target rows -- tgt_labels (B*slots), tgt_boxes (B*slots, 2) -- of which the
 * first gt_counts[b] (DEVICE int64 (B)) are real.  cost (nl,B,Q,slots) holds each video's own block only; columns at
 * or beyond the count are written as 0 and never read (the LSAP problem descriptor carries n = gt_counts[b]). */
int gvl_match_cost_padded_f32(const float *pred_logits, const float *pred_boxes, const int64_t *tgt_labels,
                              const float *tgt_boxes, const int64_t *gt_counts, int n_layers, int B, int Q,
                              int n_classes, int slots, float w_class, float w_bbox, float w_giou, float alpha,
                              float gamma, float *cost, int *ok, void *stream);

/* -- set criterion of ALL decoder layers: replaces SetCriterion.loss_labels / loss_boxes / loss_cardinality
 *    (pdvc/criterion.py:48-132) with sigmoid_focal_loss (:232-257) and cross_entropy_with_gaussian_mask (:209-229).
 *      pred_logits (nl,B,Q,NC), pred_count (nl,B,count_bins), pred_boxes (nl,B,Q,2)
 *      match_q / match_t (nl, n_pairs) int64   matched query / video-local target of every pair (pairs sorted by video)
 *      pair_video, pair_target_base (n_pairs)  video of the pair, first row of that video in the concatenated targets
 *      video_pair_start (B+1)                  first pair of every video
 *      tgt_labels (G), tgt_boxes (G,2), gt_counts (B) int64, counter_class_rate (count_bins)
 *      video_pair_count (B) int64 or NULL      padded layout: only pair slots [start[v], start[v] + count[v]) of video v
 *                                              hold a match (match_q = -1 elsewhere); NULL = every slot is a match
 *      num_boxes_dev (1) float or NULL         the normaliser of criterion.py:178-181 read from DEVICE memory (a captured
 *                                              step refreshes it per replay); NULL = the scalar argument num_boxes
 *    forward : losses (nl, 6) = [loss_ce, loss_counter, loss_bbox, loss_giou, loss_self_iou, cardinality_error]
 *    backward: grad_losses (nl, 6) -> grad_logits, grad_count, grad_boxes (same shapes as the predictions, fully
 *              written), with PyTorch's subgradient conventions. */
int gvl_set_criterion_forward_f32(const float *pred_logits, const float *pred_count, const float *pred_boxes,
                                  const int64_t *match_q, const int64_t *match_t, const int64_t *pair_video,
                                  const int64_t *pair_target_base, const int64_t *video_pair_start,
                                  const int64_t *tgt_labels, const float *tgt_boxes, const int64_t *gt_counts,
                                  const float *counter_class_rate, int n_layers, int B, int Q, int n_classes,
                                  int count_bins, int n_pairs, int G, float num_boxes, float focal_alpha,
                                  float focal_gamma, float lloss_beta, int lloss_gau_mask,
                                  const int64_t *video_pair_count, const float *num_boxes_dev, float *losses,
                                  void *stream);
int gvl_set_criterion_backward_f32(const float *pred_logits, const float *pred_count, const float *pred_boxes,
                                   const int64_t *match_q, const int64_t *match_t, const int64_t *pair_video,
                                   const int64_t *pair_target_base, const int64_t *video_pair_start,
                                   const int64_t *tgt_labels, const float *tgt_boxes, const int64_t *gt_counts,
                                   const float *counter_class_rate, int n_layers, int B, int Q, int n_classes,
                                   int count_bins, int n_pairs, int G, float num_boxes, float focal_alpha,
                                   float focal_gamma, float lloss_beta, int lloss_gau_mask,
                                   const int64_t *video_pair_count, const float *num_boxes_dev,
                                   const float *grad_losses, float *grad_logits, float *grad_count, float *grad_boxes,
                                   void *stream);

/* -- the sampling-offset / attention-weight projections of MSDeformAttn.forward (pdvc/ops/modules/ms_deform_attn.py:99-100:
 *    `self.sampling_offsets(query)`, `self.attention_weights(query)`) as ONE hand-written fp32 MFMA GEMM against the
 *    concatenated weight:  out (R, N) = x (R, K) . weight (N, K)^T + bias (N);  row-major, contiguous, fp32; bias may be
 *    NULL.  K in {256, 512, 1024} and N a multiple of 64 (512 and 256 on the path); x / weight 16-byte aligned.  Exact fp32 arithmetic
 *    (v_mfma_f32_16x16x4_f32).  Columns [0, N/2) are the raw offsets, [N/2, N) the attention logits that
 *    gvl_msda1d_fused_forward_f32 consumes as `proj`. */
int gvl_proj_f32(const float *x, const float *weight, const float *bias, int R, int K, int N, float *out, void *stream);

/* -- the dense layers around the deformable attention, for inference (gvl_layers.hip): every nn.Linear of the encoder /
 *    decoder layers -- value_proj, sampling_offsets + attention_weights, output_proj (pdvc/ops/modules/ms_deform_attn.py:
 *    95,99-100,125), linear1 / linear2 (pdvc/deformable_transformer.py:189-199,257-261), the in / out projections of
 *    nn.MultiheadAttention (:266-270), the box MLP (pdvc/pdvc.py:1166-1178) -- as ONE kernel on the fp16 matrix cores at
 *    fp32 accuracy:
 *        out_s[r][n - n_begin_s] = epilogue_s( sum_k (a[r][k] [+ a2[r % a2_rows][k]]) w[n][k] + bias[n] )
 *    a (R, K) fp32, row stride lda (a multiple of 4; lda < K = overlapping rows is allowed: the three taps of a k = 3,
 *    stride 2 conv1d over a zero-padded (frames, channels) input are ONE row of length 3 C at stride 2 C): the activation as
 *    its producer left it.  It is split into the (hi, 2^11 residual) fp16
 *    planes of gvl_split_rows_f16 INSIDE the kernel's load path; the row scale comes from amax_in (R): any upper bound of
 *    max_k |a[r][k] (+ a2)| that is at most ~2^10 above it (the producers on the path leave the exact row maximum:
 *    gvl_layer_norm_rows_f32, gvl_msda1d_fused_forward_amax_f32, this function's own amax_out, gvl_row_absmax_f32).
 *    w: planes + scales of the (N, K) weight from gvl_split_rows_f16, N % 64 == 0, K % 32 == 0; bias (N) or NULL.
 *    The N columns are cut into 1..4 SEGMENTS (ascending n_begin, multiples of 64, the first one 0), each with its own
 *    output matrix, epilogue and A variant, so that one launch serves e.g. [value_proj ; sampling_offsets ;
 *    attention_weights] where value_proj multiplies src and the other two src + pos (ms_deform_attn.py:95,99-100).
 *    Epilogue order: + bias, ReLU (GVL_LIN_RELU), resid[r][.] + (.), rows with rowmask[r] != 0 written as zeros (the
 *    masked_fill of ms_deform_attn.py:96-97); amax_out (R floats, ZERO-INITIALISED by the caller, or NULL) receives
 *    max_n |out_s[r][n]| through one atomic max per row and tile.
 *    flags: GVL_LIN_XCD_COLUMNS = 64-wide column tile c is computed on XCD c % 8 (workgroup id % 8): the head slabs of a
 *    `value` tensor are then written from the XCD whose L2 gvl_msda1d_fused_forward_* reads them from.
 *    Accuracy: |error| <= 2^-21 sum|a||w| + K 2^-33 amax_in[r] max|w| (three fp16 MFMAs, fp32 accumulation, lo.lo dropped);
 *    a non-finite a[r][k] makes row r of the outputs non-finite. */
#define GVL_LIN_ADDEND 1       /* segment flag: the A operand of this segment is a + a2 */
#define GVL_LIN_RELU 2         /* segment flag */
#define GVL_LIN_XCD_COLUMNS 1  /* launch flag */
typedef struct gvl_lin_seg {
  int n_begin;                  /* first column of the segment in the concatenated weight */
  int flags;                    /* GVL_LIN_ADDEND | GVL_LIN_RELU */
  float *out;                   /* (R, n_end - n_begin), row stride ldo */
  int64_t ldo;
  const float *amax_in;         /* (R) row maxima of the A operand this segment multiplies */
  const float *resid;           /* (R, n_end - n_begin), row stride ldr, or NULL */
  int64_t ldr;
  float *amax_out;              /* (R) or NULL */
  const unsigned char *rowmask; /* (R) or NULL */
  int width;                    /* columns [width, n_end - n_begin) of the segment are not stored (a weight block padded
                                   to a multiple of 64 rows); 0 = all */
} gvl_lin_seg;
int gvl_linear_f16x3_f32(const float *a, int64_t lda, const float *a2, int64_t lda2, int a2_rows, int R, int K,
                         const void *w_hi, const void *w_lo, const float *w_scale, const float *bias, int N,
                         const gvl_lin_seg *segs_host, int nseg, int flags, void *stream);
/*    gvl_layer_norm_rows_f32: torch.nn.LayerNorm over the last axis of x (R, C) (deformable_transformer.py:193,198,261,
 *        270,277; biased variance, eps inside the square root), C % 4 == 0, C <= 1024, one wavefront per row; also writes
 *        amax_y[r] = max |y[r][.]| and amax_ypos[r] = max |y[r][.] + pos[r % pos_rows][.]| (either may be NULL; pos (pos_rows,
 *        C) may be NULL): the row maxima gvl_linear_f16x3_f32 needs for `y` and for the attention query `y + pos`.
 *    gvl_row_absmax_f32: the same two row maxima for a tensor some other kernel produced. */
int gvl_layer_norm_rows_f32(const float *x, int R, int C, const float *gamma, const float *beta, float eps,
                            const float *pos, int pos_rows, float *y, float *amax_y, float *amax_ypos, void *stream);
int gvl_row_absmax_f32(const float *x, int64_t ldx, int R, int C, const float *pos, int64_t ldp, int pos_rows,
                       float *amax_x, float *amax_xpos, void *stream);
/*    gvl_box_refine_f32: the iterative box refinement of one decoder layer (pdvc/deformable_transformer.py:314-324, the
 *        same arithmetic again at pdvc/pdvc.py:465-474) and the reference points of the NEXT layer (:301-306) in one
 *        launch:  new_ref[r] = sigmoid(delta[r] + inverse_sigmoid(ref[r]))  (RD = 2; RD = 1: only the centre gets the prior),
 *        inverse_sigmoid as misc/detr_utils/misc.py:582-586 (eps = 1e-5);  ref_in[r][l] = new_ref[r] * valid_ratios[b(r)][l].
 *        delta (R, ldd >= 2), ref (R, RD), valid_ratios (B, L), R = B * Q -> new_ref (R, 2), ref_in (R, L, 2) (may be NULL).
 *    gvl_count_head_f32: predict_event_num (pdvc/pdvc.py:316-319): out[b] = W . max_q hs[b][q][:] + bias for hs (B, Q, C),
 *        W (n_out, C) -> (B, n_out); one workgroup per video. */
/*    gvl_mha_core_f32: the attention core of nn.MultiheadAttention in the decoder layer (deformable_transformer.py:266-270):
 *        out[b][q][h*64 ..] = softmax_k(q_h . k_h / 8, keys with key_keep[b][k] == 0 excluded) v_h, for qkv (B*Q, ld) rows
 *        [q | k | v] (each H*64 wide, the in-projection's output), head dimension 64, Q <= 320, exact fp32 arithmetic
 *        (v_mfma_f32_16x16x4_f32); key_keep (B, Q) bytes or NULL (all keys); out (B*Q, H*64); amax_out (B*Q) zero-initialised
 *        or NULL receives max |out row|. */
int gvl_mha_core_f32(const float *qkv, int64_t ld, const unsigned char *key_keep, int B, int Q, int H, float *out,
                     float *amax_out, void *stream);
/*    gvl_encoder_geometry_f32: valid ratios (pdvc/deformable_transformer.py:81-83,113: share of un-padded frames per level)
 *        and the encoder's reference points (:209-218) from the flattened padding mask in one launch:
 *        valid_ratios[b][l] = #{t < T_l : !mask[b][s_l + t]} / T_l;   ref[b][s_l + t][l'] = (t + 0.5) / (vr[b][l] T_l) * vr[b][l'].
 *        mask (B, S) bytes (non-zero = padded), level lengths / starts as host arrays of L <= 8 entries. */
/*    gvl_group_norm_rows_f32 / gvl_pyramid_geometry_f32: what surrounds the conv1d products of the base encoder's feature
 *    pyramid in inference (pdvc/base_encoder.py:55-82, pdvc/position_encoding.py:38-64, deformable_transformer.py:85-115).
 *      group norm: nn.GroupNorm(G, C) of a level's conv output given as ROWS y[(n rows_per_video + t) ldy + c] (t < T; the
 *        layout gvl_linear_f16x3_f32 leaves), written as rows dst[n dst_video_stride + t C + c] (the level's slice of the
 *        flattened (B, S, C) encoder input) and, when dst2 != NULL, to dst2[n dst2_video_stride + t C + c] (the zero-padded
 *        input of the next level's stride-2 convolution).  C / G must divide 64.
 *      geometry: mask_flat (N, S) = every level's padding mask (level 0: mask itself; level l: nearest-neighbour resampling,
 *        F.interpolate(mask.float(), size=T_l)); lvl_pos (N, S, n_sine + n_dur) = PositionEmbeddingSine of that mask (see
 *        gvl_pos_embed_sine_f32) transposed, + level_embed[l] (deformable_transformer.py:105). */
/*    TRAINING (ABI 12): gvl_group_norm_rows_backward_f32 -- the backward of gvl_group_norm_rows_f32: dy for the level's
 *    rows_per_video rows per video (the padding rows behind the T frames are zero), per-video partial sums of the affine
 *    gradients (N, C); `dout2` (may be NULL) is a second gradient of the same rows, the next level's input gradient in ITS
 *    padded layout.  gvl_conv_taps_to_rows_f32 -- the input gradient of Conv1d(k = 3, stride 2, padding 1) from the gradient of
 *    its rows of taps: dcols (N, T1, 3, C) -> dx (N, 2 T1, C), row 2 t' + k of a video = tap k of output frame t'. */
int gvl_group_norm_rows_backward_f32(const float *y, int64_t ldy, int rows_per_video, int N, int T, int C, int G, const float *gamma,
                                     float eps, const float *dout, int64_t dout_video_stride, const float *dout2,
                                     int64_t dout2_video_stride, float *dy, int64_t ld_dy, float *dgamma_part, float *dbeta_part,
                                     void *stream);
/*    ..._amax_f32 (ABI 16): amax_dy (N * rows_per_video, zero-initialised) or NULL additionally receives max |dy row| (atomic max over
 *    the row's groups): the row scale of the level's weight- and input-gradient products. */
int gvl_group_norm_rows_backward_amax_f32(const float *y, int64_t ldy, int rows_per_video, int N, int T, int C, int G,
                                          const float *gamma, float eps, const float *dout, int64_t dout_video_stride,
                                          const float *dout2, int64_t dout2_video_stride, float *dy, int64_t ld_dy,
                                          float *dgamma_part, float *dbeta_part, float *amax_dy, void *stream);
int gvl_conv_taps_to_rows_f32(const float *dcols, int N, int T1, int C, float *dx, void *stream);
int gvl_group_norm_rows_f32(const float *y, int64_t ldy, int rows_per_video, int N, int T, int C, int G, const float *gamma,
                            const float *beta, float eps, float *dst, int64_t dst_video_stride, float *dst2,
                            int64_t dst2_video_stride, void *stream);
int gvl_pyramid_geometry_f32(const unsigned char *mask, int N, int T0, int S, int L, const int64_t *lengths_host,
                             const int64_t *starts_host, const float *dim_t, const float *dur_embed,
                             const float *level_embed, int n_sine, int n_dur, float scale, unsigned char *mask_flat,
                             float *lvl_pos, void *stream);
int gvl_encoder_geometry_f32(const unsigned char *mask, int B, int S, int L, const int64_t *lengths_host,
                             const int64_t *starts_host, float *valid_ratios, float *ref, void *stream);
int gvl_box_refine_f32(const float *delta, int64_t ldd, const float *ref, int RD, const float *valid_ratios, int B, int Q,
                       int L, float *new_ref, float *ref_in, void *stream);
/*    TRAINING (ABI 15): its gradient -- grad_delta (R, 2) = g o (1 - o) with o = new_ref; grad_ref (R, RD) or NULL (only the first
 *    decoder layer's reference points carry a gradient: every later layer reads them detached, deformable_transformer.py:322) =
 *    grad_delta[c] times the derivative of inverse_sigmoid where its clamps are inactive (misc/detr_utils/misc.py:582-586). */
int gvl_box_refine_backward_f32(const float *grad_new_ref, const float *new_ref, const float *ref, int RD, int R, float *grad_delta,
                                float *grad_ref, void *stream);
int gvl_count_head_f32(const float *hs, int B, int Q, int C, const float *weight, const float *bias, int n_out, float *out,
                       void *stream);
/*    TRAINING (ABI 16): the pooling of predict_event_num alone (pdvc/pdvc.py:317 `torch.max(hs_lid, dim=1)`) with its argument,
 *    and its gradient -- pooled (B, C) = max_q hs[b][q][:], arg (B, C) int32 = the first row attaining it; grad_hs (B, Q, C) =
 *    grad_pooled[b][c] at row arg[b][c], 0 elsewhere (torch.max's gradient: the selected element alone).  The Linear on the pooled
 *    vector stays with the caller (16 rows).  grad_row (B Q) + w_row (C), both or neither: the input gradient of a ONE-output
 *    Linear on the same rows (the class head of a single-class config, pdvc/pdvc.py:455) is added, grad_row[b][q] w_row[c];
 *    grad_pooled may then be NULL. */
int gvl_count_pool_f32(const float *hs, int B, int Q, int C, float *pooled, int *arg, void *stream);
/*    gvl_batch_sum_f32 (ABI 16): the gradient of the query embedding (Q, parts * C) whose column blocks the decoder expands over
 *    the batch (pdvc/deformable_transformer.py:128-135: query_pos, tgt = chunk(query_embed), each .expand(bs, -1, -1)):
 *    out[q][h * C + c] = sum_b grads[h][b][q][c], grads: `parts` (1..4) host-side pointers to contiguous (B, Q, C) tensors (NULL =
 *    no gradient reached that block: zeros). */
int gvl_batch_sum_f32(const float *const *grads, int parts, int B, int Q, int C, float *out, void *stream);
/*    gvl_level_sums_f32 (ABI 16): the per-video half of the level embedding's gradient (pdvc/deformable_transformer.py:100,
 *    `lvl_pos_embed = pos_embed + level_embed[lvl]`): part[b][l][c] = sum of g[b][s][c] over the rows s of level l (starts / lengths:
 *    host arrays of the L <= 8 levels' row ranges in S); the sum over b is gvl_batch_sum_f32(parts = 1, Q = L). */
int gvl_level_sums_f32(const float *g, int B, int S, int C, const int *starts, const int *lengths, int L, float *part, void *stream);
/*    gvl_mask_rows_f32 / _backward_f32 (ABI 16): `value = value.masked_fill(input_padding_mask[..., None], 0)` (pdvc/ops/modules/
 *    ms_deform_attn.py:100) on the fresh (R, C) output of value_proj: rows r with mask[r] != 0 are zeroed IN PLACE (only those rows
 *    are written); backward: dx = dy with the same rows zeroed, out of place, and amax[r] = max |dx[r][.]| from the same pass. */
int gvl_mask_rows_f32(float *y, const unsigned char *mask, int R, int C, void *stream);
/*    gvl_index_add_rows_f32 (ABI 16): dst[idx[r]][:] += src[r][:] (float atomics) for the rows of src that are not all zero -- the
 *    gradient of the captioner's embedding lookup (pdvc/CaptioningHead/LSTM_DSA.py:96, `self.embed(it)`): the padded positions of a
 *    caption batch all index <pad> and carry zero gradients; skipping them removes the serialised additions on that one row.
 *    src (n, ld >= E), idx (n) int64 in [0, V) (others skipped), dst (V, ld_dst) already initialised (zeros for a fresh gradient). */
int gvl_index_add_rows_f32(const float *src, int64_t ld, const int64_t *idx, int n, int E, float *dst, int64_t ld_dst, int V,
                           void *stream);
int gvl_mask_rows_backward_f32(const float *dy, const unsigned char *mask, int R, int C, float *dx, float *amax, void *stream);
int gvl_count_pool_backward_f32(const float *grad_pooled, const int *arg, int B, int Q, int C, const float *grad_row,
                                const float *w_row, float *grad_hs, void *stream);

/* -- fp32 products of the captioner's token loop on the fp16 matrix cores at fp32 accuracy (gvl_gemm16.hip): the
 *    nn.Linear calls `self.logit(output)` (pdvc/CaptioningHead/LSTM_DSA.py:121,165), `h2att(h)` and the two halves of
 *    the LSTM's gate pre-activations (:247,267-269).
 *    gvl_split_rows_f16:  x (R, K) fp32 row-major, K % 32 == 0 -> hi, lo: R K IEEE halves each, scale (R) fp32 with
 *        x[r][k] = scale[r] (hi[r][k] + 2^-11 lo[r][k])  to 2^-22 |x|;  scale[r] = 2^floor(log2 max_k |x[r][k]|).
 *        PLANE LAYOUT (every entry point that takes or leaves planes; ABI 10): K-stage-major, element (r, k) at
 *        ((k / 32) R + r) 32 + k % 32 -- the 64 bytes one K stage takes from a row lie beside the next row's, so a staging
 *        instruction of the GEMM kernels reads ONE contiguous 1 KiB run instead of 16 half lines (-12 % on the vocabulary
 *        product).  R is the row count of the WHOLE plane: operands are passed whole, not as row slices.
 *    gvl_gemm_f16x3_f32:  out (R, ldo >= N) = A (R, K) . B (N, K)^T + bias (N, may be NULL), both operands as the
 *        planes + scales of gvl_split_rows_f16, K % 32 == 0, planes 16-byte aligned.  Three fp16 MFMAs per product
 *        (hi.hi, hi.lo, lo.hi; fp32 accumulation, cross terms in their own accumulator):
 *        |error| <= 2^-21 sum_k |a||b| + K 2^-33 max_k|a| max_k|b| per output; at K >= 256 measured BELOW the summation
 *        error of an fp32 GEMM (whose chain rounds K times). */
int gvl_split_rows_f16(const float *x, int R, int K, void *hi, void *lo, float *scale, void *stream);
/*    gvl_f16_products(n): how many fp16 products EVERY split-fp16 entry point of this library (gvl_gemm_f16x3_*,
 *        gvl_linear_f16x3_f32) spends per fp32 product from now on, for the calling thread: 3 = the exact split above
 *        (default), 1 = hi.hi only -- both operands rounded to fp16 at their row scale (11 significant bits, bf16 keeps 8;
 *        no overflow whatever the magnitude), fp32 accumulation; the lo planes are then neither fetched nor read.  This is
 *        what the host runs the Linear layers on under `torch.autocast` (the reference's autocast runs them in bf16;
 *        results agree with either to bf16 rounding).  n = 0 queries.  Returns the previous value, or GVL_EINVAL (< 0). */
int gvl_f16_products(int n);
int gvl_gemm_f16x3_f32(const void *a_hi, const void *a_lo, const float *a_scale, int R, const void *b_hi, const void *b_lo,
                       const float *b_scale, int N, int K, const float *bias, float *out, int64_t ldo, void *stream);
/*    The same product for the vocabulary layer of GREEDY decoding, with the consumer fused (`torch.max(logprobs, 1)` over
 *    `F.log_softmax(self.logit(output))`, LSTM_DSA.py:121-123,165-167): the logits (R, V) are never written.
 *    gvl_gemm_f16x3_argmax_f32 leaves, per token row and per 64 vocabulary entries, {max, sum exp(v - max), index of the
 *    first maximum, -} in partials (gvl_gemm_f16x3_argmax_chunks(V), R, 4) fp32 (16-byte aligned);
 *    gvl_greedy_step_partials_f32 reduces them to token (R) = argmax, logp (R) = log-softmax at the argmax and applies the
 *    bookkeeping of gvl_greedy_step_f32 (unfinished NULL = none).  Ties resolve to the lowest index, as torch.max. */
/*    Producers that leave their result directly in that operand form (no gvl_split_rows_f16 pass over it):
 *    gvl_cap_attend_split_f32 = gvl_cap_attend_f32 with att_res as planes + row scale (att_res itself is not written);
 *    gvl_lstm_cell_split_f32 = gvl_lstm_cell_f32 that ALSO writes h' as planes (row scale 1: |h'| < 1 by construction). */
int gvl_cap_attend_split_f32(const float *slab, const int64_t *shapes, const int64_t *lsi, const float *ref,
                             const float *off_hs, const float *h, const float *w_off_h, const float *att_h,
                             const float *alpha_w, float alpha_b, int B, int S, int C, int L, int Q, int P, int RD,
                             int att_h_ld, void *att_hi, void *att_lo, float *att_scale, void *stream);
int gvl_lstm_cell_split_f32(const float *gates_a, int lda, const float *gates_b, int ldb, const float *emb_gates,
                            const int64_t *it, const float *gates_c, int ldc, const float *c, int n, int H, float *h_out,
                            float *c_out, void *h_hi, void *h_lo, float *h_scale, void *stream);
/*    gvl_cap_attend_split_levels_f32 = gvl_cap_attend_split_f32 for a caller that also knows the level starts on the
 *        HOST (lsi_host (L), the values of `lsi`; NULL = unknown): with L = P = 4 and few enough rows in the two coarsest
 *        levels (all of levels 2, 3 plus level 3 again <= 63 rows: T <= 100 at the pyramid of the path) the kernel keeps
 *        those rows of each video's slab in LDS -- the ctx2att half of levels 2 and 3, the value half of level 3 -- and
 *        serves 3 / 8 of all sample reads from there instead of the CU's vector-memory path, which bounds the plain
 *        kernel.  Same results (the same arithmetic on the same numbers). */
int gvl_cap_attend_split_levels_f32(const float *slab, const int64_t *shapes, const int64_t *lsi, const float *ref,
                                    const float *off_hs, const float *h, const float *w_off_h, const float *att_h,
                                    const float *alpha_w, float alpha_b, int B, int S, int C, int L, int Q, int P, int RD,
                                    int att_h_ld, const int64_t *lsi_host, void *att_hi, void *att_lo, float *att_scale,
                                    void *stream);
/*    gvl_cap_attend_pre_f32 = gvl_cap_attend_split_levels_f32 whose caller has ALREADY multiplied the hidden state into the
 *        offsets (off_pre (B*Q, >= L*P) row stride off_pre_ld = h . sampling_offsets.weight[:, :C]^T,
 *        ms_deform_attn_for_caption.py:100-103 -- in the token loop these are 16 more output columns of the h2att(h) product
 *        that runs in front of this kernel).  Neither h nor the 32 KB weight is read; the LDS the weight occupied holds the
 *        value half of level 2 too, so HALF of all sample reads are served from LDS.  The sum h W^T is then rounded as the
 *        fp16x3 product rounds it (2^-21 relative of sum |h||w|) instead of as this kernel's fp32 butterfly does.
 *        gvl_cap_attend_pre_applicable: 1 when the LDS form exists for this pyramid (L = P = 4, level starts known on the
 *        host, levels 2 + 3 + 3 <= 79 rows); the entry fails with GVL_EINVAL otherwise. */
int gvl_cap_attend_pre_applicable(int S, int L, int P, const int64_t *lsi_host);
int gvl_cap_attend_pre_f32(const float *slab, const int64_t *shapes, const int64_t *lsi, const float *ref, const float *off_hs,
                           const float *off_pre, int off_pre_ld, const float *att_h, const float *alpha_w, float alpha_b, int B,
                           int S, int C, int L, int Q, int P, int RD, int att_h_ld, const int64_t *lsi_host, void *att_hi,
                           void *att_lo, float *att_scale, void *stream);
/*    gvl_gemm_f16x3_lstm_f32: the attention half of the LSTM input product WITH the cell applied to the finished tile
 *        (LSTM_DSA.py:267-269 + nn.LSTM's pointwise part, :216-217): gates = A (R, K) . W (4H, K)^T + gates_c + gates_h +
 *        emb_gates[it], (h', c') = cell(gates, c); the (R, 4H) product is never written and gvl_lstm_cell_split_f32 does not
 *        run.  Every gate operand -- the ROWS of W and the columns of gates_h (row stride ld_h), gates_c (ld_c, may be NULL)
 *        and emb_gates (V + 1, 4H) -- is in the order 4 * unit + gate (gate = i, f, g, o), i.e. row g * H + u of nn.LSTM's
 *        weight_ih at row 4 * u + g.  Outputs as gvl_lstm_cell_split_f32; same bits as that path (shared cell expression,
 *        same product).  H % 32 == 0, K % 32 == 0. */
int gvl_gemm_f16x3_lstm_f32(const void *a_hi, const void *a_lo, const float *a_scale, int R, const void *w_hi,
                            const void *w_lo, const float *w_scale, int H, int K, const float *gates_h, int64_t ld_h,
                            const float *gates_c, int64_t ld_c, const float *emb_gates, const int64_t *it, const float *c,
                            float *h_out, float *c_out, void *h_hi, void *h_lo, float *h_scale, void *stream);
/*    gvl_gemm_f16x3_gates_f32 (ABI 12): BOTH halves of the LSTM gate product of a token step in one launch, the cell applied:
 *        gates = [h | A] (n, K_h + K_a) . W (4H, K_h + K_a)^T + gates_c + emb_gates[it], (h', c') = cell(gates, c)
 *        (LSTM_DSA.py:267-269 + nn.LSTM's pointwise part) -- the first K_h contraction columns of W (nn.LSTM's weight_hh) multiply
 *        the planes (hp_*) of the step's incoming hidden state, the other K_a (weight_ih's attention columns) the planes of A;
 *        rows of W, columns of gates_c / emb_gates in the order 4 * unit + gate as for gvl_gemm_f16x3_lstm_f32.  The (n, 4H)
 *        recurrent part is neither written by a product in front nor read here.  The new state (h_out, c_out, planes) must be
 *        other buffers than the incoming one; outputs 16-byte aligned; h_out may be NULL (ABI 13: a caller whose readers of h'
 *        all take the planes saves a third of the kernel's stores).  gvl_gemm_f16x3_gates_applicable(n, H): 1 when the
 *        kernel's 256 x 160 tiles fill the chip in whole rounds (cfg A: 240 tiles), else the two-launch form is the faster one. */
int gvl_gemm_f16x3_gates_applicable(int n, int H);
int gvl_gemm_f16x3_gates_f32(const void *a_hi, const void *a_lo, const float *a_scale, const void *hp_hi, const void *hp_lo,
                             const float *hp_scale, int n, const void *w_hi, const void *w_lo, const float *w_scale, int H,
                             int K_h, int K_a, const float *gates_c, int64_t ld_c, const float *emb_gates, const int64_t *it,
                             const float *c, float *h_out, float *c_out, void *h_hi, void *h_lo, float *h_scale, void *stream);
int gvl_gemm_f16x3_argmax_chunks(int V);
int gvl_gemm_f16x3_argmax_f32(const void *x_hi, const void *x_lo, const float *x_scale, int R, const void *w_hi,
                              const void *w_lo, const float *w_scale, int V, int K, const float *bias, float *partials,
                              void *stream);
int gvl_greedy_step_partials_f32(const float *partials, int R, int V, int first_step, int64_t *token, float *logp,
                                 unsigned char *unfinished, int64_t *seq_col, float *seq_lp_col, int seq_ld, void *stream);
/*    ..._alive_f32: additionally sets *alive = 1 (device byte, zero-initialised by the caller, may be NULL) when any row is
 *    still unfinished after this step -- the loop-exit test `unfinished.sum() == 0` of LSTM_DSA.py:186-187 without a
 *    reduction kernel per step. */
int gvl_greedy_step_partials_alive_f32(const float *partials, int R, int V, int first_step, int64_t *token, float *logp,
                                       unsigned char *unfinished, int64_t *seq_col, float *seq_lp_col, int seq_ld,
                                       unsigned char *alive, void *stream);
/*    ..._gemm_f32 (ABI 12): the same AND, in the same launch, an independent plain product out (Ra, Nb) = A . B^T + bias on
 *    operand planes (the arguments of gvl_gemm_f16x3_f32) -- in the greedy loop the reduction of token t and h2att(h) for token
 *    t + 1 (LSTM_DSA.py:247) depend on different results of the step and run side by side (one launch instead of two). */
int gvl_greedy_step_partials_gemm_f32(const float *partials, int R, int V, int first_step, int64_t *token, float *logp,
                                      unsigned char *unfinished, int64_t *seq_col, float *seq_lp_col, int seq_ld,
                                      unsigned char *alive, const void *a_hi, const void *a_lo, const float *a_scale, int Ra,
                                      const void *b_hi, const void *b_lo, const float *b_scale, int Nb, int K, const float *bias,
                                      float *out, int64_t ldo, void *stream);

/* -- PositionEmbeddingSine.forward of one pyramid level (pdvc/position_encoding.py:38-64; the step in front of the
 *    path): normalised cumulative frame index -> interleaved sin / cos over `dim_t`, followed
 *    by the duration embedding broadcast over time.
 *      mask (N, T) bool/uint8, nonzero = padding;  dim_t (n_sine) = temperature ** (2 (i // 2) / n_sine);
 *      dur_embed (N, n_dur) = duration_embed_layer(step one-hot);  scale = 2 pi;  out (N, n_sine + n_dur, T). */
int gvl_pos_embed_sine_f32(const unsigned char *mask, const float *dim_t, const float *dur_embed, int N, int T,
                           int n_sine, int n_dur, float scale, float *out, void *stream);

/* -- the residual chains of the encoder / decoder layers in TRAINING (gvl_train_layers.hip; ABI 10):
 *        y = LayerNorm(x + dropout(sub))     pdvc/deformable_transformer.py:189-199 (norm1, norm2), :266-280 (norm2, norm1, norm3)
 *    one forward and one backward kernel instead of PyTorch's dropout + add + layer_norm (3 launches) and their 5 backward
 *    launches.  y, z, dy, dz, dsub: (R, C) fp32 row-major, C % 4 == 0, C <= 1024; x and sub: row r = (b, q), b = r / Q, at element
 *    offset b sb + q sq (strides multiples of 4: a transposed (Q, B, C) tensor or a batch-expanded one is read in place);
 *    gamma, beta (C).
 *    forward leaves z = x + keep sub / (1 - p), mean (R), rstd (R) (biased variance + eps, as torch.nn.LayerNorm) for the
 *    backward.  backward: dz = gradient w.r.t. x, dsub = keep dz / (1 - p) (NULL allowed when p == 0: it equals dz),
 *    dgamma_dbeta (2 C) = [sum_r dy xhat | sum_r dy]; part: workspace of gvl_rdln_backward_blocks(R) x 2 C floats.
 *    Dropout: element i kept iff hash32(i ^ hash32(seed + *step 0x9E3779B9)) >= p 2^32 -- `step` lives in DEVICE memory
 *    (gvl_advance_step: *step += 1, one tiny kernel per training forward; NULL = 0), so a step replayed from a hipGraph draws
 *    new masks; the backward regenerates the mask from the same (seed, *step).  Not torch's Philox stream; p == 0 is exactly
 *    LayerNorm(x + sub). */
int gvl_residual_dropout_layer_norm_forward_f32(const float *x, int64_t x_sb, int64_t x_sq, const float *sub, int64_t sub_sb,
                                                int64_t sub_sq, int Q, int R, int C, const float *gamma, const float *beta,
                                                float eps, float p, uint32_t seed, const int64_t *step, float *y, float *z,
                                                float *mean, float *rstd, int64_t *step_used, const float *pos, int64_t pos_sb,
                                                int64_t pos_sq, float *amax_y, float *amax_ypos, void *stream);
int gvl_residual_dropout_layer_norm_backward_f32(const float *dy, const float *z, const float *mean, const float *rstd, int R,
                                                 int C, const float *gamma, float p, uint32_t seed, const int64_t *step,
                                                 float *dz, float *dsub, float *part, float *dgamma_dbeta, float *amax_dz,
                                                 void *stream);
/*    ABI 11 additions (all optional, NULL = absent): step_used receives the value of *step this forward drew its masks for -- the
 *    backward is given THAT pointer as `step`, so a forward that ran in between (and advanced the live counter) cannot change the
 *    masks the backward regenerates; amax_y[r] = max |y[r][.]| and amax_ypos[r] = max |y[r][.] + pos[row r][.]| (pos addressed
 *    like x: b * pos_sb + q * pos_sq) -- the row maxima the next Linear product's split needs for `y` and for the attention query
 *    `y + pos` (gvl_linear_f16x3_f32); amax_dz[r] = max |dz[r][.]| / (1 - p), a bound of row r of dz AND of dsub. */
/*    ABI 16: ..._backwardn_f32 -- the same with n_dy = 1 .. gvl_rdln_backward_max_grads() gradients of the output (dys: host-side
 *    array of pointers), summed as they are loaded: the norm's result feeds the next sublayer, the next residual, heads ..., and
 *    each hands back a gradient. */
int gvl_residual_dropout_layer_norm_backwardn_f32(const float *const *dys, int n_dy, const float *z, const float *mean,
                                                  const float *rstd, int R, int C, const float *gamma, float p, uint32_t seed,
                                                  const int64_t *step, float *dz, float *dsub, float *part, float *dgamma_dbeta,
                                                  float *amax_dz, void *stream);
int gvl_rdln_backward_max_grads(void);
int gvl_rdln_backward_blocks(int R);
/*    y = dropout(relu(x)) of the FFNs (deformable_transformer.py:189-191, 257-259), n elements (n % 4 == 0, < 2^32), y == x
 *    allowed; same mask rule as above.  backward: dx = y > 0 ? dy / (1 - p) : 0 -- no mask tensor (y > 0 exactly
 *    where the element was kept and positive). */
int gvl_relu_dropout_forward_f32(const float *x, int64_t n, float p, uint32_t seed, const int64_t *step, float *y, void *stream);
int gvl_relu_dropout_backward_f32(const float *dy, const float *y, int64_t n, float p, float *dx, void *stream);
/*    the same two kernels over ROWS of C elements (C % 4 == 0, C <= 4096; one wavefront per row), leaving max |row| of the result
 *    in amax (R): the row maxima of the FFN's hidden activation / of its gradient for the Linear products around it. */
int gvl_relu_dropout_rows_forward_f32(const float *x, int R, int C, float p, uint32_t seed, const int64_t *step, float *y,
                                      float *amax, int64_t *step_used, void *stream);
int gvl_relu_dropout_rows_backward_f32(const float *dy, const float *y, int R, int C, float p, float *dx, float *amax, void *stream);
int gvl_advance_step(int64_t *step, void *stream);

/* -- column sums of a row-major fp32 matrix: out[c] = sum_r x[r * ld + c] -- the bias gradient of the nn.Linear layers on
 *    the path (autograd's AddmmBackward: grad_bias = grad_output.sum(0)).  Any C and ld >= C (16-byte loads when C, ld
 *    are multiples of 4 and x is 16-byte aligned); out is overwritten.  Summation order across row chunks is not
 *    fixed (float atomics). */
int gvl_col_sum_f32(const float *x, int ld, int R, int C, float *out, void *stream);

/* -- caption loss over the vocabulary without the (R, V) log-prob tensor (pdvc/CaptioningHead/LSTM_DSA.py:48-52 applied
 *    to the log_softmax of :121-123; R = caption rows x token steps):
 *      gvl_ce_rows_forward_f32:   out[r] = weight[r] (logits[r][target[r]] - lse[r]),  lse[r] = logsumexp_v logits[r][v]
 *                                 (= `F.log_softmax(logits, -1).gather(-1, target) * mask`); rows with weight 0 are not read
 *                                 (out = lse = 0);
 *      gvl_ce_rows_backward_f32:  logits[r][v] <- grad_out[r] weight[r] ((v == target[r]) - exp(logits[r][v] - lse[r])), IN
 *                                 PLACE: the gradient with respect to the logits, from which the caller takes the three
 *                                 gradients of the vocabulary layer.
 *    logits (R, ld >= V) fp32, target (R) int64 in [0, V), weight / out / lse / grad_out (R) fp32. */
int gvl_ce_rows_forward_f32(const float *logits, int64_t ld, int R, int V, const int64_t *target, const float *weight,
                            float *out, float *lse, void *stream);
int gvl_ce_rows_backward_f32(float *logits, int64_t ld, int R, int V, const int64_t *target, const float *weight,
                             const float *grad_out, const float *lse, float *amax, void *stream);
/*    (ABI 11: amax (R) or NULL receives |grad_out[r] weight[r]|, an upper bound of row r of the gradient -- the row scale of the
 *     split-fp16 products that consume it; the padding columns [V, ld) of every row are set to zero, ld - V <= 256) */

/* -- kernel timing inside the library (measurement only; no reference equivalent).  While enabled, every kernel the
 *    library launches is dispatched with hipExtLaunchKernel start/stop events on the launch stream, i.e. the
 *    begin/end stamps of that one dispatch (what rocprofv3 --kernel-trace reports), independent of host launch gaps.
 *    gvl_prof_collect synchronises the recorded events, writes up to `capacity` (duration in microseconds, tag,
 *    meta_a = Q (rows for the captioner kernels), meta_b = B) tuples in launch order, clears the log and returns the
 *    count.  Not for use during hipGraph capture. */
#define GVL_PROF_FWD_T1D 1
#define GVL_PROF_FWD_GENERIC 2
#define GVL_PROF_BWD_T1D 3
#define GVL_PROF_BWD_GENERIC 4
#define GVL_PROF_SAMPLE 5
#define GVL_PROF_SUM_PARTIALS 6
#define GVL_PROF_SAMPLE_BWD 7
#define GVL_PROF_CAP_ATTEND 8
#define GVL_PROF_ROW_ARGMAX 9
#define GVL_PROF_LSTM_CELL 10
#define GVL_PROF_LSAP 11
#define GVL_PROF_CAP_TRAIN_FWD 12
#define GVL_PROF_CAP_TRAIN_BWD 13
#define GVL_PROF_LSTM_TRAIN 14
#define GVL_PROF_MATCH_COST 15
#define GVL_PROF_CRITERION 16
#define GVL_PROF_POS_EMBED 17
#define GVL_PROF_COL_SUM 18
#define GVL_PROF_PROJ 19
#define GVL_PROF_SPLIT 20
#define GVL_PROF_GEMM16 21
#define GVL_PROF_LINEAR 22
#define GVL_PROF_LAYER_NORM 23
#define GVL_PROF_WGRAD 24
#define GVL_PROF_MHA_TRAIN 25
int gvl_prof_enable(int on);   /* 0 off | 1 sampling-path kernels | 2 also GVL_PROF_PROJ (stamping two consecutive launches
                                  inflates the second one's interval by 2-3 us, so level 1 leaves the projection alone) */
/* Phase stamps of the temporal kernels (diagnostics): while a DEVICE buffer of 2 x 4096 x 4 uint64 is set, every
 * workgroup of k_fwd_t1d_d64 records the 100 MHz wall clock at {start, slab staged, loop done} in the first half and
 * every workgroup of k_bwd_t1d_d64 {start, staged, phase 1 done, phase 2 done} in the second half.  NULL = off. */
void gvl_msda_debug_stamps(void *device_buffer);
/* Shader-clock probe (diagnostics, no reference equivalent): one wavefront runs a chain of `n_fma` dependent v_fma_f32 and
 * writes {s_memtime ticks (shader cycles), s_memrealtime ticks (100 MHz)} of the chain to device_out2 (2 x int64): the clock the
 * chip holds at this point of the stream = 100 MHz x out[0] / out[1].  bench.py reports it next to its timed regions (a
 * kernel-time comparison between two runs means little when one of them ran at half the clock). */
int gvl_clock_probe(long long *device_out2, int n_fma, void *stream);
int gvl_prof_collect(float *us, int *tag, int *meta_a, int *meta_b, int capacity);

/* -- forward: replaces ms_deform_attn_forward (pdvc/ops/src/ms_deform_attn.h:20-39 ->
 *    ms_deform_attn_cuda_forward, pdvc/ops/src/cuda/ms_deform_attn_cuda.cu:20-80).  `out` is fully overwritten
 *    (the reference zero-fills then accumulates, cu:54); im2col_step batching (cu:50-75) is an artefact of the
 *    reference launcher and has no equivalent here -- any B is processed in one launch. */
int gvl_msda_forward_f32(const float *value, const int64_t *shapes, const int64_t *lsi, const float *loc,
                         const float *attn, int B, int S, int M, int D, int L, int Q, int P, int pad_mode,
                         const int64_t *shapes_host, const int64_t *lsi_host, float *out, void *stream);
int gvl_msda_forward_f64(const double *value, const int64_t *shapes, const int64_t *lsi, const double *loc,
                         const double *attn, int B, int S, int M, int D, int L, int Q, int P, int pad_mode,
                         const int64_t *shapes_host, const int64_t *lsi_host, double *out, void *stream);

/* -- unweighted samples: replaces ms_deform_attn_core_pytorch(..., return_value=True)
 *    (pdvc/ops/functions/ms_deform_attn_func.py:44-68) as used by MSDeformAttnCap
 *    (pdvc/ops/modules/ms_deform_attn_for_caption.py:124-125).  sample layout (B*M, D, Q, L, P). */
int gvl_msda_sample_f32(const float *value, const int64_t *shapes, const int64_t *lsi, const float *loc, int B,
                        int S, int M, int D, int L, int Q, int P, int pad_mode, float *sample, void *stream);
int gvl_msda_sample_f64(const double *value, const int64_t *shapes, const int64_t *lsi, const double *loc, int B,
                        int S, int M, int D, int L, int Q, int P, int pad_mode, double *sample, void *stream);

/* -- backward: replaces ms_deform_attn_backward (pdvc/ops/src/ms_deform_attn.h:41-61 ->
 *    ms_deform_attn_cuda_backward, pdvc/ops/src/cuda/ms_deform_attn_cuda.cu:83-153).
 *    grad_value (B,S,M,D), grad_loc (B,Q,M,L,P,2), grad_attn (B,Q,M,L,P) are fully (over)written: the callee
 *    zero-fills what it accumulates into (the reference allocates zeros, cu:121-123), so the caller may pass
 *    uninitialised buffers.  `workspace` must hold gvl_msda_backward_workspace_bytes(...) bytes (may be NULL
 *    when that is 0; pass the same shapes_host to the query as to the call). */
size_t gvl_msda_backward_workspace_bytes(int B, int S, int M, int D, int L, int Q, int P, int elem_bytes,
                                         const int64_t *shapes_host);
int gvl_msda_backward_f32(const float *value, const int64_t *shapes, const int64_t *lsi, const float *loc,
                          const float *attn, const float *grad_out, int B, int S, int M, int D, int L, int Q, int P,
                          int pad_mode, const int64_t *shapes_host, const int64_t *lsi_host, float *grad_value,
                          float *grad_loc, float *grad_attn, void *workspace, size_t workspace_bytes, void *stream);
int gvl_msda_backward_f64(const double *value, const int64_t *shapes, const int64_t *lsi, const double *loc,
                          const double *attn, const double *grad_out, int B, int S, int M, int D, int L, int Q,
                          int P, int pad_mode, const int64_t *shapes_host, const int64_t *lsi_host,
                          double *grad_value, double *grad_loc, double *grad_attn, void *workspace,
                          size_t workspace_bytes, void *stream);

/* -- bf16 storage twins of the two calls above (same reference interfaces; arithmetic fp32, see "Element types").
 *    Workspace: gvl_msda_backward_workspace_bytes(..., elem_bytes = 2, ...) -- never 0 for bf16 (the gather writes
 *    fp32 slabs that a second kernel sums and rounds once). */
int gvl_msda_forward_bf16(const uint16_t *value, const int64_t *shapes, const int64_t *lsi, const float *loc,
                          const float *attn, int B, int S, int M, int D, int L, int Q, int P, int pad_mode,
                          const int64_t *shapes_host, const int64_t *lsi_host, uint16_t *out, void *stream);
int gvl_msda_backward_bf16(const uint16_t *value, const int64_t *shapes, const int64_t *lsi, const float *loc,
                           const float *attn, const uint16_t *grad_out, int B, int S, int M, int D, int L, int Q, int P,
                           int pad_mode, const int64_t *shapes_host, const int64_t *lsi_host, uint16_t *grad_value,
                           float *grad_loc, float *grad_attn, void *workspace, size_t workspace_bytes, void *stream);

/* -- fused module path (temporal levels, fp32, D = 64, L*P = 16, P = 4): everything MSDeformAttn.forward does between
 *    its projection GEMM and output_proj (pdvc/ops/modules/ms_deform_attn.py:99-124) in one launch, so that the
 *    sampling locations and attention weights never exist in HBM.
 *      proj  (B*Q, 2*M*L*P)  query @ [sampling_offsets.weight; attention_weights.weight]^T + bias:
 *                            columns [0, M*L*P) raw offsets (m,l,p), [M*L*P, 2*M*L*P) attention logits (m, l*p)
 *      ref   (B, Q, L, RD)   reference points, RD = 1 (centre) or 2 (centre, length)
 *    forward:  softmax over L*P (:100-101); loc = ref + off / T_l (RD=1, :103-106) or ref_c + off/P*ref_len*0.5
 *              (RD=2, :107-109); y = 0.5 (:115); then the op.  out (B, Q, M*D).
 *    backward: grad_value (B,S,M,D); grad_proj (B*Q, 2*M*L*P) (offsets and logits, softmax backward applied);
 *              grad_ref (B,Q,M,L,RD) per-head partials of d/d ref (may be NULL; the caller sums over M).
 *    host copies of shapes / lsi are REQUIRED here (the temporal kernels are the only implementation).
 *    Returns GVL_EINVAL when the shape is outside the fused kernels' domain (callers then use the unfused op). */
int gvl_msda1d_fused_forward_f32(const float *value, const int64_t *shapes, const int64_t *lsi, const float *proj,
                                 const float *ref, int B, int S, int M, int D, int L, int Q, int P, int RD,
                                 int pad_mode, const int64_t *shapes_host, const int64_t *lsi_host, float *out,
                                 void *stream);
/*    ..._amax_f32: the same forward, additionally leaving amax_out[b*Q + q] = max_c |out[b][q][c]| (B*Q floats,
 *    ZERO-INITIALISED by the caller; one atomic max per (row, head)): the row maxima from which gvl_linear_f16x3_f32
 *    derives the operand scale of `output_proj` (ms_deform_attn.py:125) behind it. */
int gvl_msda1d_fused_forward_amax_f32(const float *value, const int64_t *shapes, const int64_t *lsi, const float *proj,
                                      const float *ref, int B, int S, int M, int D, int L, int Q, int P, int RD,
                                      int pad_mode, const int64_t *shapes_host, const int64_t *lsi_host, float *out,
                                      float *amax_out, void *stream);
/*    ..._shared_amax_f32: proj_q (Q, 2*M*L*P) -- the SAME offsets / logits rows for every video (inference with the 'queries'
 *    input, deformable_transformer.py:128-135: the first decoder layer's self-attention block and the projection behind it see the
 *    batch-expanded query embedding and no video; gvl_amd/layers.py keeps their result per parameter set).  Everything else as
 *    ..._amax_f32; the launch reads Q instead of B*Q operand rows. */
int gvl_msda1d_fused_forward_shared_amax_f32(const float *value, const int64_t *shapes, const int64_t *lsi, const float *proj_q,
                                             const float *ref, int B, int S, int M, int D, int L, int Q, int P, int RD,
                                             int pad_mode, const int64_t *shapes_host, const int64_t *lsi_host, float *out,
                                             float *amax_out, void *stream);
size_t gvl_msda1d_fused_backward_workspace_bytes(int B, int S, int M, int D, int L, int Q, int P,
                                                 const int64_t *shapes_host);
int gvl_msda1d_fused_backward_f32(const float *value, const int64_t *shapes, const int64_t *lsi, const float *proj,
                                  const float *ref, const float *grad_out, int B, int S, int M, int D, int L, int Q,
                                  int P, int RD, int pad_mode, const int64_t *shapes_host, const int64_t *lsi_host,
                                  float *grad_value, float *grad_proj, float *grad_ref, void *workspace,
                                  size_t workspace_bytes, void *stream);
/*    bf16 storage: value, proj, out, grad_out, grad_value, grad_proj are bfloat16; ref / grad_ref fp32.  Workspace:
 *    gvl_msda_backward_workspace_bytes(..., elem_bytes = 2, ...). */
int gvl_msda1d_fused_forward_bf16(const uint16_t *value, const int64_t *shapes, const int64_t *lsi,
                                  const uint16_t *proj, const float *ref, int B, int S, int M, int D, int L, int Q,
                                  int P, int RD, int pad_mode, const int64_t *shapes_host, const int64_t *lsi_host,
                                  uint16_t *out, void *stream);
int gvl_msda1d_fused_backward_bf16(const uint16_t *value, const int64_t *shapes, const int64_t *lsi,
                                   const uint16_t *proj, const float *ref, const uint16_t *grad_out, int B, int S,
                                   int M, int D, int L, int Q, int P, int RD, int pad_mode,
                                   const int64_t *shapes_host, const int64_t *lsi_host, uint16_t *grad_value,
                                   uint16_t *grad_proj, float *grad_ref, void *workspace, size_t workspace_bytes,
                                   void *stream);

/* -- backward of gvl_msda_sample: autograd of ms_deform_attn_core_pytorch(return_value=True) (func.py:44-68; the
 *    reference differentiates through F.grid_sample).  grad_sample (B*M, D, Q, L, P) -> grad_value (B,S,M,D)
 *    (zero-filled by the callee) and grad_loc (B,Q,M,L,P,2).  Used by the teacher-forced captioner in training. */
int gvl_msda_sample_backward_f32(const float *value, const int64_t *shapes, const int64_t *lsi, const float *loc,
                                 const float *grad_sample, int B, int S, int M, int D, int L, int Q, int P,
                                 int pad_mode, float *grad_value, float *grad_loc, void *stream);
int gvl_msda_sample_backward_f64(const double *value, const int64_t *shapes, const int64_t *lsi, const double *loc,
                                 const double *grad_sample, int B, int S, int M, int D, int L, int Q, int P,
                                 int pad_mode, double *grad_value, double *grad_loc, void *stream);

/* -- one LSTM-DSA token step's deformable soft attention, fused (inference): replaces, per decoding step,
 *    MSDeformAttnCap.forward (pdvc/ops/modules/ms_deform_attn_for_caption.py:82-127) + the additive attention of
 *    ShowAttendTellCore.forward (pdvc/CaptioningHead/LSTM_DSA.py:247-266).
 *      slab     (B, S, 2C)  [value_proj(memory) (padded rows zeroed) | ctx2att(value_proj(memory))], C = 512
 *      ref      (B, Q, L, RD) reference points already multiplied by the valid ratios (LSTM_DSA.py:137-141)
 *      off_hs   (B*Q, L*P)  sampling_offsets bias + the part of the projection that multiplies hs
 *      h        (B*Q, C)    previous hidden state;  w_off_h (L*P, C) = sampling_offsets.weight[:, :C]
 *      att_h    (B*Q, C)    h2att(h), rows att_h_ld floats apart (>= C; lets it be a column block of a wider GEMM
 *                           output);  alpha_w (C), alpha_b: alpha_net
 *      att_res  (B*Q, C)    sum_k softmax_k(alpha_net(tanh(ctx2att(clip_k) + att_h))) * clip_k
 *    dbg_alpha / dbg_loc (B*Q, L*P) are optional outputs (may be NULL) used by the parity tests. */
int gvl_cap_attend_f32(const float *slab, const int64_t *shapes, const int64_t *lsi, const float *ref,
                       const float *off_hs, const float *h, const float *w_off_h, const float *att_h,
                       const float *alpha_w, float alpha_b, int B, int S, int C, int L, int Q, int P, int RD,
                       int att_h_ld, float *att_res, float *dbg_alpha, float *dbg_loc, void *stream);

/* -- pointwise part of the captioner's LSTM cell (single layer, bias-free nn.LSTM, LSTM_DSA.py:216-217,269):
 *    gates = (gates_c +) gates_a + gates_b + emb_gates[it] (row strides lda / ldb / ldc floats, gate order i,f,g,o,
 *    4H wide; gates_c may be NULL: the token-independent part, kept out of the per-token GEMM so that it needs no
 *    beta = 1 C operand);  c' = sigmoid(f) c + sigmoid(i) tanh(g);  h' = sigmoid(o) tanh(c').  emb_gates is the
 *    embedding table already multiplied by its slice of W_ih ((V+1, 4H)); it (n) int64 token ids. */
int gvl_lstm_cell_f32(const float *gates_a, int lda, const float *gates_b, int ldb, const float *emb_gates,
                      const int64_t *it, const float *gates_c, int ldc, const float *c, int n, int H, float *h_out,
                      float *c_out, void *stream);

/* -- TRAINING-time token step of the same captioner (teacher forcing; LSTM_DSA.py:63-117 loop, :241-271 step, and
 *    their autograd).  The matched queries only (pdvc.py:743-760), so n = B*Q rows is small and the step is
 *    launch-bound in PyTorch; here a step is  GEMM, gvl_cap_attend_train_forward, GEMM, gvl_lstm_cell_train_forward
 *    and its backward  gvl_lstm_cell_train_backward, GEMM, gvl_cap_attend_train_backward, GEMM.
 *      slab      (B, S, 2C)   [value_proj(memory) | ctx2att(value_proj(memory))], C = 512
 *      ref       (B, Q, L, RD) reference points scaled by the valid ratios; with RD = 2 a row whose length component
 *                is NEGATIVE is a centre-only point (RD = 1 arithmetic, no length gradient): one launch can then
 *                serve decoder layers with both reference forms
 *      off_hs    (B*Q, L*P)   sampling_offsets bias + hs part;  off_h (B*Q, L*P; row stride off_h_ld) its h part
 *      att_h     (B*Q, C; row stride att_h_ld)   h2att(h)
 *      alpha_w (C), alpha_b (1, DEVICE pointer)  alpha_net
 *    forward  -> att_res (B*Q, C), alpha_out (B*Q, 16) (the softmax weights, kept for the backward)
 *    backward <- grad_att_res (B*Q, C; row stride);  grad_att_h / grad_off are OVERWRITTEN (row strides given: they
 *               may be column blocks of one gradient matrix); grad_slab (B,S,2C), grad_ref (B,Q,L,RD), grad_alpha_w
 *               (C), grad_alpha_b (1) are ACCUMULATED INTO (float atomics; the caller zeroes them once per token
 *               loop, so the per-step autograd accumulation of a 12 MB slab gradient disappears).
 *    row_video (ABI 5): NULL = rows are grouped per video, row r belongs to video r / Q (as above).  Non-NULL = the
 *               COMPACT form used by the layout-independent captured train step: Q is the TOTAL number of rows, ref is
 *               (Q, L, RD), B only sizes the slab, and row_video (Q) int64 DEVICE gives each row's video; a negative
 *               entry marks an unused row of the fixed-capacity row set: its outputs (att_res, alpha_out; grad_att_h,
 *               grad_off) are written as zeros and it contributes nothing to the accumulated gradients. */
int gvl_cap_attend_train_forward_f32(const float *slab, const int64_t *shapes, const int64_t *lsi, const float *ref,
                                     const float *off_hs, const float *off_h, int off_h_ld, const float *att_h,
                                     int att_h_ld, const float *alpha_w, const float *alpha_b, int B, int S, int C,
                                     int L, int Q, int P, int RD, const int64_t *row_video, float *att_res,
                                     float *alpha_out, void *stream);
int gvl_cap_attend_train_backward_f32(const float *slab, const int64_t *shapes, const int64_t *lsi, const float *ref,
                                      const float *off_hs, const float *off_h, int off_h_ld, const float *att_h,
                                      int att_h_ld, const float *alpha_w, const float *alpha_saved,
                                      const float *grad_att_res, int grad_att_res_ld, int B, int S, int C, int L, int Q,
                                      int P, int RD, const int64_t *row_video, float *grad_slab, float *grad_att_h,
                                      int grad_att_h_ld, float *grad_off, int grad_off_ld, float *grad_ref,
                                      float *grad_alpha_w, float *grad_alpha_b, void *stream);
/*    LSTM cell (nn.LSTM single layer, bias-free; gate order i,f,g,o): gates = gates_a + gates_b + gates_c (row strides
 *    in floats).  forward keeps the ACTIVATED gates act (n, 4H); backward takes dh = grad_h_a + grad_h_b (either may
 *    be NULL), grad_c (may be NULL) and writes the pre-activation gate gradients (row stride grad_gates_ld) and
 *    grad_c_prev. */
int gvl_lstm_cell_train_forward_f32(const float *gates_a, int lda, const float *gates_b, int ldb, const float *gates_c,
                                    int ldc, const float *c_prev, int n, int H, float *act, float *h_out, float *c_out,
                                    void *stream);
int gvl_lstm_cell_train_backward_f32(const float *grad_h_a, const float *grad_h_b, const float *grad_c, const float *act,
                                     const float *c_prev, const float *c_new, int n, int H, float *grad_gates,
                                     int grad_gates_ld, float *grad_c_prev, void *stream);
/*    ..._sum_f32 (ABI 16): additionally keeps the running sum of the gate gradients over the token steps in grad_gates_sum
 *    (n, 4H contiguous; first != 0: this step starts the sum, else it is added to) -- the gradient of the token-independent gate
 *    part (W_ih's event-feature block times hs, LSTM_DSA.py:267-269), a (steps, n, 4H) reduction after the loop otherwise. */
int gvl_lstm_cell_train_backward_sum_f32(const float *grad_h_a, const float *grad_h_b, const float *grad_c, const float *act,
                                         const float *c_prev, const float *c_new, int n, int H, float *grad_gates,
                                         int grad_gates_ld, float *grad_c_prev, float *grad_gates_sum, int first, void *stream);

/* -- greedy decoding epilogue: idx[r] = argmax_v logits[r, v] (first maximal index), logp[r] = log_softmax(logits[r])
 *    at that index (LSTM_DSA.py:123 + :166-167), one read of the logits. */
int gvl_row_argmax_lse_f32(const float *logits, int R, int V, int64_t *idx, float *logp, void *stream);
/*    The same with the bookkeeping of one greedy decoding step (LSTM_DSA.py:180-190) fused in:
 *      unfinished[r] = (first_step || unfinished[r]) && token[r] > 0
 *      seq_col[r * seq_ld] = unfinished[r] ? token[r] : 0;   seq_lp_col[r * seq_ld] = logp[r]
 *    seq_col / seq_lp_col point at column t of the (R, seq_ld) output tensors; token[] is the RAW argmax (what the
 *    reference feeds to the next LSTM step). */
int gvl_greedy_step_f32(const float *logits, int R, int V, int first_step, int64_t *token, float *logp,
                        unsigned char *unfinished, int64_t *seq_col, float *seq_lp_col, int seq_ld, void *stream);

/* -- bf16-input twins of the three inference token-step kernels (BASELINE config 4, "CaptioningHead cross-attn, bf16"):
 *    under torch.autocast the GEMMs of the token loop leave the slab, h2att(h), the gate pre-activations, the
 *    pre-multiplied embedding table and the logits in bfloat16; these entry points read them as they are (uint16_t bit
 *    patterns), compute in fp32 exactly as the _f32 kernels do on the widened values, and keep the recurrent state
 *    (h, c), offsets and reference points in fp32.  att_res is written in bf16 (it is the A operand of the next bf16
 *    GEMM) and gvl_lstm_cell_bf16 can leave a bf16 copy of h' (h_bf16, may be NULL) next to the fp32 state for the same
 *    reason: one rounding either way, no separate cast kernels. */
int gvl_cap_attend_bf16(const uint16_t *slab, const int64_t *shapes, const int64_t *lsi, const float *ref,
                        const float *off_hs, const float *h, const float *w_off_h, const uint16_t *att_h,
                        const float *alpha_w, float alpha_b, int B, int S, int C, int L, int Q, int P, int RD,
                        int att_h_ld, uint16_t *att_res, float *dbg_alpha, float *dbg_loc, void *stream);
int gvl_lstm_cell_bf16(const uint16_t *gates_a, int lda, const uint16_t *gates_b, int ldb, const uint16_t *emb_gates,
                       const int64_t *it, const uint16_t *gates_c, int ldc, const float *c, int n, int H, float *h_out,
                       float *c_out, uint16_t *h_bf16, void *stream);
int gvl_row_argmax_lse_bf16(const uint16_t *logits, int R, int V, int64_t *idx, float *logp, void *stream);
int gvl_greedy_step_bf16(const uint16_t *logits, int R, int V, int first_step, int64_t *token, float *logp,
                         unsigned char *unfinished, int64_t *seq_col, float *seq_lp_col, int seq_ld, void *stream);

/* -- Hungarian matcher index path (HOST pointers, host code).  Replaces scipy.optimize.linear_sum_assignment as
 *    called at pdvc/matcher.py:124,126; results are bit-identical to scipy 1.15.3 (same augmenting-path order and
 *    tie-breaking, float32 costs promoted to float64).  row_ind / col_ind have length min(nr, nc); row_ind is
 *    ascending.  Returns 0, or GVL_EINVAL for bad arguments and for NaN / -inf costs or an infeasible matrix
 *    (scipy raises ValueError there). */
int gvl_lsap_solve_f64(const double *cost, int64_t nr, int64_t nc, int64_t *row_ind, int64_t *col_ind);
int gvl_lsap_solve_f32(const float *cost, int64_t nr, int64_t nc, int64_t *row_ind, int64_t *col_ind);

/* The whole of matcher.py:120-131 for one decoder layer: C is the HOST cost tensor (B, Q, G), G = sum(sizes);
 * video i owns columns [off_i, off_i + sizes[i]).  Writes, concatenated over videos, the one-to-one indices
 * (idx_*: sum_i min(Q, n_i) entries) and the many-to-one indices on the block tiled m2o_rate times with
 * GT id = col % n_i (rl_*: sum_i min(Q, m2o_rate*n_i) entries).  Videos are solved on num_threads host threads
 * (<= 0: hardware concurrency). */
int gvl_hungarian_batch_f32(const float *C, int B, int Q, int G, const int *sizes, int m2o_rate, int64_t *idx_rows,
                            int64_t *idx_cols, int64_t *rl_rows, int64_t *rl_cols, int num_threads);

/* The same index path ON THE DEVICE: one wavefront per assignment problem, all problems of a step in one launch,
 * no device->host copy (which is what lets a whole train / eval step be captured in a hipGraph).  Bit-identical
 * to gvl_lsap_solve_* / scipy (same float64 arithmetic and tie-breaking; tests/test_gpu_lsap.py).
 *   C         DEVICE float32 cost storage
 *   problems  DEVICE int64 [n_problems][8]: {base, ld, Q, n, tile, out_off, 0, 0}; problem p is the Q x (n*tile)
 *             matrix with element (q, k) = C[base + q*ld + (k % n)]  (tile = 1: one-to-one, tile = 4: the
 *             many-to-one variant of matcher.py:125-127 with GT id = col % n)
 *   rows_out / cols_out  DEVICE int64; problem p writes min(Q, n*tile) entries at out_off (rows ascending)
 *   status    DEVICE int, set to 1 if any problem is infeasible / has NaN or -inf costs (caller zeroes it)
 *   max_rows / max_cols: the largest min(Q, n*tile) and max(Q, n*tile) over the problems (<= 256 / <= 1024). */
int gvl_lsap_batch_device_f32(const float *C, const int64_t *problems, int n_problems, int max_rows, int max_cols,
                              int64_t *rows_out, int64_t *cols_out, int *status, void *stream);

/* -- TRAINING (ABI 15): the compact (query, caption) pair rows of the captioner on padded targets -- what pdvc.py:540-573 gathers
 *    per decoder layer (matched queries' hidden states, their captions and masks), for the fixed-capacity layout of
 *    gvl_amd/pdvc.py: caption_prediction_layers_padded.  Row r of layer k = the r-th matched pair in (video, slot) order:
 *      flat[k R + r]      = (k N + video) Nq + matched query      (row of the stacked (layers, N, Nq) hidden states)
 *      row_video[k R + r] = video, or -1 for rows beyond the batch's pair count (their caption is all <pad>, mask 0)
 *      seq / mask         = cap_tensor / cap_mask [video][matched target]  (cap_len columns)
 *      denom[0]           = N max(1, max_v pair_count[v])          (the divisor of pdvc.py:868's mean)
 *    pair_count (N) int64; q_layers / t_layers: HOST arrays of n_layers device pointers to (N G1) int64 match arrays (negative =
 *    unmatched, read as 0); cap_tensor (N, slots, cap_len) int64, cap_mask fp32.  One launch for ~45 small index operations. */
int gvl_caption_rows(const int64_t *pair_count, const int64_t *const *q_layers, const int64_t *const *t_layers, int n_layers, int N,
                     int G1, int R, int Nq, int slots, int cap_len, const int64_t *cap_tensor, const float *cap_mask, int64_t *flat,
                     int64_t *row_video, int64_t *seq, float *mask, float *denom, void *stream);

/* -- TRAINING: the weight / bias gradient of an nn.Linear -- what autograd's AddmmBackward computes as `grad_output.t().mm(input)`
 *    and `grad_output.sum(0)` for every Linear of the encoder / decoder layers (pdvc/deformable_transformer.py:189-199,257-280),
 *    of MSDeformAttn (pdvc/ops/modules/ms_deform_attn.py:95,99-100,125) and of the captioner's vocabulary layer
 *    (pdvc/CaptioningHead/LSTM_DSA.py:121) -- on the fp16 matrix cores at fp32 accuracy, one pass over dy for both:
 *        grad_w[n][k] (+)= sum_r dy[r][n] x[r][k]      grad_b[n] (+)= sum_r dy[r][n]
 *    dy (R, N) row stride ld_dy, x (R, K) row stride ld_x, fp32, N, K and both strides multiples of 4, 16-byte aligned.
 *    amax_dy / amax_x: n_amax_* >= 1 fp32 upper bounds of |dy| / |x| whose maximum bounds the whole tensor (the row maxima the
 *    producers of the path leave behind, or one number); the split scale is ONE power of two per tensor (the contraction runs
 *    over the rows).  accumulate != 0 adds to the existing grad_w / grad_b (AccumulateGrad folded in).  grad_b may be NULL.
 *    workspace: gvl_wgrad_workspace_bytes(R, N, K) bytes (split-K partial tiles, summed in a fixed order: deterministic).
 *    |error| <= 2^-22 sum_r |dy||x| + R 2^-36 max|dy| max|x|. */
size_t gvl_wgrad_workspace_bytes(int R, int N, int K);
/*    GROUPED (ABI 15): the same gradients for up to gvl_wgrad_group_max() Linears in ONE launch (+ one reduction): the 5-9 weight
 *    gradients a layer's backward owes at nearly the same time, none of them on the backward's critical path.  Together they
 *    cover the chip with 2-4 row ranges each instead of ~15 -- a quarter of the split-K partial traffic, one launch pair instead
 *    of one per Linear.  descs: HOST array (the fields of gvl_wgrad_f16x3_f32's arguments per problem); workspace:
 *    gvl_wgrad_group_workspace_bytes(descs, n) bytes, 16-byte aligned.  Results equal the single form's to summation order
 *    (other row ranges); deterministic. */
typedef struct gvl_wgrad_desc {
  const float *dy, *x, *amax_dy, *amax_x;
  float *grad_w, *grad_b;
  int64_t ld_dy, ld_x;
  int n_amax_dy, n_amax_x, R, N, K, accumulate;
} gvl_wgrad_desc;
int gvl_wgrad_group_max(void);
size_t gvl_wgrad_group_workspace_bytes(const gvl_wgrad_desc *descs, int n);
int gvl_wgrad_group_f16x3_f32(const gvl_wgrad_desc *descs, int n, void *workspace, size_t workspace_bytes, void *stream);
int gvl_wgrad_f16x3_f32(const float *dy, int64_t ld_dy, const float *amax_dy, int n_amax_dy, const float *x, int64_t ld_x,
                        const float *amax_x, int n_amax_x, int R, int N, int K, float *grad_w, float *grad_b, int accumulate,
                        void *workspace, size_t workspace_bytes, void *stream);
/*    ..._live_f32 (ABI 16): live_ws (gvl_wgrad_live_ints(R) ints of device scratch) or NULL: a list of the rows of dy with a non-zero
 *    bound is built first (amax_dy must then be per row, n_amax_dy == R) and only those are multiplied, 32 list entries per stage --
 *    rows whose bound is 0 are all zeros, e.g. the padded positions of a teacher-forced caption batch in the vocabulary layer's
 *    gradient. */
int gvl_wgrad_f16x3_live_f32(const float *dy, int64_t ld_dy, const float *amax_dy, int n_amax_dy, const float *x, int64_t ld_x,
                             const float *amax_x, int n_amax_x, int R, int N, int K, float *grad_w, float *grad_b, int accumulate,
                             void *workspace, size_t workspace_bytes, int *live_ws, void *stream);
int gvl_wgrad_live_ints(int R);

/* -- TRAINING: operand planes of all weights of the step in two launches.  The forward product of an nn.Linear needs the planes
 *    of W (N, K), its input gradient `grad_output.mm(weight)` (AddmmBackward) those of W^T; the weights change with every
 *    optimizer step (train.py:405-409), so both are rebuilt once per training forward -- for ALL matrices at once.
 *    Each descriptor (DEVICE array) names one fp32 matrix w (N, K), K % 32 == 0, contiguous (N % 32 == 0 unless it is the only or
 *    the last block of its operand: the transposed planes then carry zeros for the contraction entries N .. round_up(N, 32)), and
 *    where its planes go:
 *      hi / lo / scale      planes of the concatenated operand [.. ; W ; ..] of n_total rows: this matrix fills rows n_off ..
 *                           n_off + N (n_off % 32 == 0), K-stage-major as gvl_split_rows_f16 writes them; scale[n] per row
 *      t_hi / t_lo / t_scale  planes of the TRANSPOSED operand (K rows, contraction over the n_total rows); NULL = not wanted
 *      group_chunk_begin / group_chunks   the chunk maxima (see below) of ALL matrices of this matrix's group: one scale per group
 *    chunk_map (n_chunks x {descriptor, chunk}) / wg_map (n_workgroups x {descriptor, first 32 x 32 tile}): the launch geometry,
 *    built once by the caller: a matrix has ceil(N K / gvl_planes_chunk_elems()) chunks and ceil(N K / 4096) workgroups of four
 *    tiles.  chunk_amax: n_chunks floats of scratch. */
typedef struct gvl_plane_desc {
  const float *w;
  int N, K;
  void *hi, *lo;
  float *scale;
  int n_total, n_off;
  void *t_hi, *t_lo;
  float *t_scale;
  int group_chunk_begin, group_chunks;
  const float *bias;   /* (N) or NULL: copied to bias_dst[n_off ..] (the concatenated bias of the operand) */
  float *bias_dst;
} gvl_plane_desc;
int gvl_planes_chunk_elems(void);
int gvl_planes_refresh_f16(const gvl_plane_desc *descs_device, const int *chunk_map_device, int n_chunks, const int *wg_map_device,
                           int n_workgroups, float *chunk_amax_device, void *stream);

/* -- TRAINING: the attention core of the decoder layer's nn.MultiheadAttention (pdvc/deformable_transformer.py:266-270: self_attn
 *    over the queries with key_padding_mask and dropout on the attention weights), forward and backward, head dimension 64:
 *        out[b][q][h 64 ..] = sum_k dropout(softmax_k(q_h . k_h / 8, keys with key_keep[b][k] == 0 excluded))[k] v_h[k]
 *    qkv (B Q, ld) rows [q | k | v] (each H 64 wide: the in-projection's output, batch-major rows), out / dout (B Q, H 64),
 *    lse (B, H, Q) = log sum_k exp(score) (kept for the backward), dqkv laid out like qkv.  amax_qk / amax_v (B Q): upper bounds of
 *    |row| of the q and k columns / of the v columns (the in-projection's epilogue leaves them).  Dropout: weight (b, h, q, k) is
 *    kept iff hash32(((b H + h) Q + q) Q + k ^ hash32(seed + *step 0x9E3779B9)) >= p 2^32 (the rule of gvl_residual_dropout_*;
 *    the backward is given the step value the forward used); p == 0: plain softmax attention.  A query whose keys are all
 *    excluded yields NaN, as torch.softmax of a fully masked row.  delta_ws (B H Q) and amax_dout_ws (B Q): scratch.
 *    Products: three fp16 MFMAs per fp32 product in one fp32 accumulator (22-bit operands); scores never leave registers. */
int gvl_mha_train_forward_f32(const float *qkv, int64_t ld, const unsigned char *key_keep, const float *amax_qk, const float *amax_v,
                              int B, int Q, int H, float p, uint32_t seed, const int64_t *step, float *out, float *lse,
                              float *amax_out, void *stream);
/*    (the forward also serves INFERENCE, p = 0, lse = NULL: amax_out (B Q) zero-initialised or NULL receives max |out row| over
 *     the heads -- the row scale of the out-projection behind it; 35 us against 55 us for gvl_mha_core_f32 at B = 16, Q = 300) */
int gvl_mha_train_backward_f32(const float *qkv, int64_t ld, const unsigned char *key_keep, const float *amax_qk, const float *amax_v,
                               int B, int Q, int H, float p, uint32_t seed, const int64_t *step, const float *out, const float *lse,
                               const float *dout, float *delta_ws, float *amax_dout_ws, float *dqkv, void *stream);
/*    ..._amax_f32 (ABI 16): additionally amax_dqk / amax_dv (B Q, zero-initialised, both or neither) receive max |row| of the
 *    [dq | dk] and of the dv columns of dqkv -- the row scales the in-projection's backward products split dqkv by. */
int gvl_mha_train_backward_amax_f32(const float *qkv, int64_t ld, const unsigned char *key_keep, const float *amax_qk,
                                    const float *amax_v, int B, int Q, int H, float p, uint32_t seed, const int64_t *step,
                                    const float *out, const float *lse, const float *dout, float *delta_ws, float *amax_dout_ws,
                                    float *dqkv, float *amax_dqk, float *amax_dv, void *stream);

/* -- the same product as gvl_linear_f16x3_f32 for FEW outputs and a LONG contraction, split over the K stages: the vocabulary layer's
 *    input gradient `grad_logits.mm(logit.weight)` (AddmmBackward of pdvc/CaptioningHead/LSTM_DSA.py:121: 2208 x 512 outputs,
 *    contraction 8518 -- 72 tiles would occupy a quarter of the chip for 267 serial K stages).  out (R, N) = a (R, K) . w^T with w
 *    as planes of an (N, K) operand, N % 128 == 0, K % 32 == 0, amax_a (R) row bounds of a; no bias / epilogue.  workspace:
 *    gvl_linear_f16x3_splitk_workspace_bytes(R, N, K) bytes (partial slabs, summed in a fixed order). */
size_t gvl_linear_f16x3_splitk_workspace_bytes(int R, int N, int K);
int gvl_linear_f16x3_splitk_f32(const float *a, int64_t lda, const float *amax_a, int R, int K, const void *w_hi, const void *w_lo,
                                const float *w_scale, int N, float *out, void *workspace, size_t workspace_bytes, void *stream);
/*    ..._bias_f32 (ABI 16): the same with a bias (N, 16-byte aligned, may be NULL) added to the finished sum, and rows that may
 *    OVERLAP (0 < lda < K, as gvl_linear_f16x3_f32 allows): the k = 3, stride 2 convolutions of the feature pyramid as products over
 *    rows of taps (a few hundred rows, contraction 3 C_in: 13-52 tiles of 48 K stages otherwise). */
int gvl_linear_f16x3_splitk_bias_f32(const float *a, int64_t lda, const float *amax_a, int R, int K, const void *w_hi,
                                     const void *w_lo, const float *w_scale, const float *bias, int N, float *out, void *workspace,
                                     size_t workspace_bytes, void *stream);

/* -- the update of the training step: clip_grad_norm_(params, max_norm) + torch.optim.Adam.step() (train.py:405-409) over a table of
 *    tensors, three launches (sum of squares per chunk; total norm, clip coefficient, bias corrections; update) -- gvl_optim.hip.
 *    descs: device array of gvl_adam_desc (p, g, m, v: fp32 device tensors of n elements; vec = 1 when all four are 16-byte aligned
 *    and n % 4 == 0); chunk_map: (tensor, chunk) per workgroup, chunks of gvl_adam_chunk_elems() elements; partial: n_chunks floats;
 *    scal: 4 floats, on return {total norm, clip coefficient, and tensor 0's two bias corrections}; corr: 2 * n_tensors floats, on
 *    return {lr / (1 - beta1^t_i), sqrt(1 - beta2^t_i)} per tensor; desc.step: tensor i's step count t_i AFTER this update (torch's
 *    state['step'], a device float -- torch.optim.Adam counts per parameter, and the counts differ once a parameter sat out a step).
 *    max_norm goes into clip_grad_norm_'s formula as it is: coefficient min(1, max_norm / (total + 1e-6)), so 0 scales the
 *    gradients to 0 exactly as the reference's unconditional call (train.py:407) would.  The gradients are scaled in place by the
 *    clip coefficient, as clip_grad_norm_ leaves them. */
typedef struct gvl_adam_desc {
  float *p;
  const float *g;
  float *m, *v;
  const float *step;
  int64_t n;
  int vec, pad_;
} gvl_adam_desc;
int gvl_adam_chunk_elems(void);
/* writes the gradient addresses (host array of n_tensors device pointers) into the table's `g` fields with a launch whose ARGUMENTS
 * carry them -- recordable into a hipGraph, where the gradients of a captured step live at other addresses than the warm-up's.
 * A desc's `vec` then asserts 16-byte alignment of p, m, v only: gradients are read 16 bytes at a time when THEY are aligned too
 * (checked by the caller per call: pass vec = 0 tables otherwise). */
int gvl_adam_set_grads(gvl_adam_desc *descs_device, int n_tensors, const void *const *grads_host, void *stream);
int gvl_clip_adam_step_f32(const gvl_adam_desc *descs_device, int n_tensors, const int *chunk_map_device, int n_chunks,
                           float *partial_device, float *scal_device, float *corr_device, double max_norm, double lr, double beta1,
                           double beta2, double eps, double weight_decay, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* GVL_MSDA_H */
