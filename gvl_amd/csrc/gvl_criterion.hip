// gvl_criterion.hip -- matcher cost matrix and set criterion of the train step as single launches (MI355X, gfx950).
//
// Reference: HungarianMatcher.forward's cost (pdvc/matcher.py:74-105, misc/detr_utils/box_ops.py:8-47) and
// SetCriterion's losses (pdvc/criterion.py:48-132,209-257).  All of it is arithmetic on a few thousand scalars
// (B*Q = 4800 logits, 48 matched boxes per decoder layer): in PyTorch ~190 launch-bound kernels per layer forward and
// as many backward -- a third of the captured train step.  Here, for ALL decoder layers at once:
//   k_match_cost       C[l,b,q,g] for every (layer, video, query, target) -- contraction off, the reference's operation
//                      order, one rounding per operation (GIoU term bit-identical to the PyTorch op sequence, focal
//                      term within 1-2 ulp of it);
//   k_criterion_fwd    one workgroup per layer: loss_ce (focal), loss_counter (gaussian-masked BCE), loss_bbox (L1),
//                      loss_giou, loss_self_iou, cardinality_error;
//   k_criterion_bwd    their gradients w.r.t. pred_logits / pred_count / pred_boxes, weighted by the upstream
//                      gradient of each loss term (PyTorch's subgradient conventions: min/max ties split in half,
//                      clamp passes the gradient at the boundary, sign(0) = 0).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gvl_common.hpp"
#include "gvl_msda.h"

namespace {

using gvl::fail;

constexpr float kEps = 1e-5f;       // box_ops.py:26,47

struct CritDims {
  int nl, B, Q, NC, Wd, T1, G;     // layers, videos, queries, classes, count bins, matched pairs per layer, targets
};

__device__ inline float sigmoid_(float x) { return 1.f / (1.f + expf(-x)); }
__device__ inline float bce_logits(float x, float t) { return fmaxf(x, 0.f) - x * t + log1pf(expf(-fabsf(x))); }
__device__ inline float powg(float b, float g) { return g == 2.f ? b * b : (g == 1.f ? b : powf(b, g)); }

// sum over the workgroup, result in every thread (blockDim.x <= 1024)
__device__ inline float block_sum(float v, float *sh) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  __syncthreads();
  if (lane == 0) sh[w] = v;
  __syncthreads();
  float t = 0.f;
  for (int k = 0; k < nw; ++k) t += sh[k];
  return t;
}

// ------------------------------------------------------------------------------------------------------
// matcher cost (matcher.py:74-105).  Separate roundings per operation, as the PyTorch kernel sequence has them.
// ------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_match_cost(const float *__restrict__ logits, const float *__restrict__ boxes,
                                                    const int64_t *__restrict__ tgt_labels,
                                                    const float *__restrict__ tgt_boxes, CritDims d, float w_class,
                                                    float w_bbox, float w_giou, float alpha, float gamma,
                                                    float *__restrict__ C, int *__restrict__ ok) {
#pragma clang fp contract(off)
  const int64_t total = (int64_t)d.nl * d.B * d.Q * d.G;
  int good = 1;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int g = (int)(idx % d.G);
    const int64_t row = idx / d.G;                                   // (l, b, q)
    const float p = sigmoid_(logits[row * d.NC + (int)tgt_labels[g]]);
    const float neg = ((1.f - alpha) * powg(p, gamma)) * (-logf((1.f - p) + 1e-8f));
    const float pos = (alpha * powg(1.f - p, gamma)) * (-logf(p + 1e-8f));
    const float cost_class = pos - neg;
    const float c = boxes[row * 2], l = boxes[row * 2 + 1];
    const float tc = tgt_boxes[g * 2], tl = tgt_boxes[g * 2 + 1];
    const float sx0 = c - 0.5f * l, sx1 = c + 0.5f * l, tx0 = tc - 0.5f * tl, tx1 = tc + 0.5f * tl;
    if (!(sx1 >= sx0) || !(tx1 >= tx0)) good = 0;                    // box_ops.py:39-40
    const float inter = fmaxf(fminf(sx1, tx1) - fmaxf(sx0, tx0), 0.f);
    const float uni = ((sx1 - sx0) + (tx1 - tx0)) - inter;
    const float iou = inter / (uni + kEps);
    const float area = fmaxf(fmaxf(sx1, tx1) - fminf(sx0, tx0), 0.f);
    const float giou = iou - (area - uni) / (area + kEps);
    float cost = w_class * cost_class;
    if (w_bbox != 0.f) cost = w_bbox * (fabsf(c - tc) + fabsf(l - tl)) + cost;   // (w_bbox*cb + w_class*cc)
    cost = cost + w_giou * (-giou);
    C[idx] = cost;
  }
  if (!good) atomicAnd(ok, 0);
}

// Padded (layout-independent) form: every video owns Gp target slots, of which the first gt_counts[b] are real;
// cost (nl, B, Q, Gp) holds only the video's own block (the concatenated form above also prices every query against
// the other videos' targets, which the solver never reads).  Slots beyond the count are written as 0 and are never
// read (the solver's problem descriptor carries n = gt_counts[b]).  Same arithmetic, same roundings.
__global__ void __launch_bounds__(256) k_match_cost_padded(const float *__restrict__ logits, const float *__restrict__ boxes,
                                                           const int64_t *__restrict__ tgt_labels,
                                                           const float *__restrict__ tgt_boxes,
                                                           const int64_t *__restrict__ gt_counts, CritDims d,
                                                           float w_class, float w_bbox, float w_giou, float alpha,
                                                           float gamma, float *__restrict__ C, int *__restrict__ ok) {
#pragma clang fp contract(off)
  const int64_t total = (int64_t)d.nl * d.B * d.Q * d.G;             // d.G = Gp slots per video
  int good = 1;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int g = (int)(idx % d.G);
    const int64_t row = idx / d.G;                                   // (l, b, q)
    const int b = (int)((row / d.Q) % d.B);
    if (g >= (int)gt_counts[b]) { C[idx] = 0.f; continue; }
    const int64_t t = (int64_t)b * d.G + g;
    const float p = sigmoid_(logits[row * d.NC + (int)tgt_labels[t]]);
    const float neg = ((1.f - alpha) * powg(p, gamma)) * (-logf((1.f - p) + 1e-8f));
    const float pos = (alpha * powg(1.f - p, gamma)) * (-logf(p + 1e-8f));
    const float cost_class = pos - neg;
    const float c = boxes[row * 2], l = boxes[row * 2 + 1];
    const float tc = tgt_boxes[t * 2], tl = tgt_boxes[t * 2 + 1];
    const float sx0 = c - 0.5f * l, sx1 = c + 0.5f * l, tx0 = tc - 0.5f * tl, tx1 = tc + 0.5f * tl;
    if (!(sx1 >= sx0) || !(tx1 >= tx0)) good = 0;                    // box_ops.py:39-40
    const float inter = fmaxf(fminf(sx1, tx1) - fmaxf(sx0, tx0), 0.f);
    const float uni = ((sx1 - sx0) + (tx1 - tx0)) - inter;
    const float iou = inter / (uni + kEps);
    const float area = fmaxf(fmaxf(sx1, tx1) - fminf(sx0, tx0), 0.f);
    const float giou = iou - (area - uni) / (area + kEps);
    float cost = w_class * cost_class;
    if (w_bbox != 0.f) cost = w_bbox * (fabsf(c - tc) + fabsf(l - tl)) + cost;
    cost = cost + w_giou * (-giou);
    C[idx] = cost;
  }
  if (!good) atomicAnd(ok, 0);
}

// ------------------------------------------------------------------------------------------------------
struct CritArgs {
  const float *logits;        // (nl, B, Q, NC)
  const float *counts;        // (nl, B, Wd)
  const float *boxes;         // (nl, B, Q, 2)  (centre, length)
  const int64_t *mq;          // (nl, T1) matched query of every pair
  const int64_t *mt;          // (nl, T1) matched target (video-local)
  const int64_t *vid;         // (T1) video of every pair (pairs sorted by video)
  const int64_t *tbase;       // (T1) first row of that video's targets in the concatenated targets
  const int64_t *ent_start;   // (B+1) first pair of every video
  const int64_t *tgt_labels;  // (G)
  const float *tgt_boxes;     // (G, 2)
  const int64_t *gt_counts;   // (B)
  const float *ccr;           // (Wd) class rate of the counter
  float num_boxes, alpha, gamma, beta;
  int gau_mask;
  // layout-independent (padded) form: pair slots [ent_start[v], ent_start[v] + pair_count[v]) of video v are real, the
  // rest of the video's slots carry no match; the normaliser lives on the device.  Both null in the compact form.
  const int64_t *pair_count;  // (B) or null
  const float *num_boxes_dev; // (1) or null
};

__device__ inline float num_boxes_of(const CritArgs &a) { return a.num_boxes_dev ? a.num_boxes_dev[0] : a.num_boxes; }
__device__ inline int pairs_end(const CritArgs &a, int v) {
  return a.pair_count ? (int)a.ent_start[v] + (int)a.pair_count[v] : (int)a.ent_start[v + 1];
}
__device__ inline bool pair_valid(const CritArgs &a, int e) {
  return !a.pair_count || e < pairs_end(a, (int)a.vid[e]);
}

__device__ inline float focal_terms(float x, float t, float alpha, float gamma, float &grad) {
  const float p = sigmoid_(x);
  const float ce = bce_logits(x, t);
  const float p_t = p * t + (1.f - p) * (1.f - t);
  const float om = 1.f - p_t;
  const float mod = powg(om, gamma);
  const float a_t = alpha >= 0.f ? alpha * t + (1.f - alpha) * (1.f - t) : 1.f;
  const float dp_t = (2.f * t - 1.f) * p * (1.f - p);
  const float dmod = gamma == 2.f ? -2.f * om * dp_t : (gamma == 1.f ? -dp_t : -gamma * powf(om, gamma - 1.f) * dp_t);
  grad = a_t * ((p - t) * mod + ce * dmod);
  return a_t * ce * mod;
}

__device__ inline float counter_coef(int c, int tgt, float beta, int gau) {
  if (c == tgt || !gau) return 1.f;
  const float dc = (float)(c - tgt);
  const float mask = expf(-(dc * dc) / 8.f);                         // criterion.py:212-214, sigma = 2
  return beta == 1.f ? 1.f - mask : powf(1.f - mask, beta);
}

// 1-D IoU of two (x0, x1) segments and its gradient w.r.t. the FIRST segment (box_ops.py:19-27)
__device__ inline float iou_grad(float a0, float a1, float b0, float b1, float &g0, float &g1) {
  const float lt = fmaxf(a0, b0), rb = fminf(a1, b1);
  const float d = rb - lt;
  const float inter = fmaxf(d, 0.f);
  const float uni = (a1 - a0) + (b1 - b0) - inter;
  const float U = uni + kEps;
  const float iou = inter / U;
  const float di = (d >= 0.f ? 1.f : 0.f) * (1.f / U + inter / (U * U));   // d iou / d (rb - lt), union = .. - inter
  const float da = -inter / (U * U);                                        // d iou / d area1
  const float w_rb = a1 < b1 ? 1.f : (a1 == b1 ? 0.5f : 0.f);               // min(a1, b1) -> a1
  const float w_lt = a0 > b0 ? 1.f : (a0 == b0 ? 0.5f : 0.f);               // max(a0, b0) -> a0
  g1 = di * w_rb + da;
  g0 = -di * w_lt - da;
  return iou;
}

// per-workgroup class map in LDS: cls[b*Q + q] = label of the matched target, NC (= no object) elsewhere
__device__ inline void build_class_map(int *cls, const CritArgs &a, const CritDims &d, int l) {
  for (int i = threadIdx.x; i < d.B * d.Q; i += blockDim.x) cls[i] = d.NC;
  __syncthreads();
  for (int e = threadIdx.x; e < d.T1; e += blockDim.x) {
    if (!pair_valid(a, e)) continue;
    const int v = (int)a.vid[e];
    cls[v * d.Q + (int)a.mq[(int64_t)l * d.T1 + e]] = (int)a.tgt_labels[a.tbase[e] + a.mt[(int64_t)l * d.T1 + e]];
  }
  __syncthreads();
}

struct PairBox {
  float c, l, tc, tl;
};
__device__ inline PairBox load_pair(const CritArgs &a, const CritDims &d, int l, int e) {
  const int64_t r = ((int64_t)l * d.B + a.vid[e]) * d.Q + a.mq[(int64_t)l * d.T1 + e];
  const int64_t t = a.tbase[e] + a.mt[(int64_t)l * d.T1 + e];
  return {a.boxes[r * 2], a.boxes[r * 2 + 1], a.tgt_boxes[t * 2], a.tgt_boxes[t * 2 + 1]};
}

constexpr int kLosses = 6;   // loss_ce, loss_counter, loss_bbox, loss_giou, loss_self_iou, cardinality_error

__global__ void __launch_bounds__(1024) k_criterion_fwd(CritArgs a, CritDims d, float *__restrict__ losses) {
  extern __shared__ int cls[];
  __shared__ float red[16];
  const int l = blockIdx.x;
  build_class_map(cls, a, d, l);
  // ---- focal classification loss (criterion.py:66,232-257): sum over (b, q, class) / num_boxes ------------------
  float s = 0.f;
  const int nlog = d.B * d.Q * d.NC;
  for (int i = threadIdx.x; i < nlog; i += blockDim.x) {
    const float t = (cls[i / d.NC] == i % d.NC) ? 1.f : 0.f;
    float g;
    s += focal_terms(a.logits[(int64_t)l * nlog + i], t, a.alpha, a.gamma, g);
  }
  const float num_boxes = num_boxes_of(a);
  const float loss_ce = block_sum(s, red) / num_boxes;
  // ---- counter (criterion.py:77,209-229) --------------------------------------------------------------------------
  s = 0.f;
  for (int i = threadIdx.x; i < d.B * d.Wd; i += blockDim.x) {
    const int b = i / d.Wd, c = i % d.Wd;
    const int tgt = (int)min(a.gt_counts[b], (int64_t)(d.Wd - 1));
    const float t = c == tgt ? 1.f : 0.f;
    s += (1.f - a.ccr[c]) * bce_logits(a.counts[(int64_t)l * d.B * d.Wd + i], t) * counter_coef(c, tgt, a.beta, a.gau_mask);
  }
  const float loss_counter = block_sum(s, red) / (float)(d.B * d.Wd);
  // ---- matched boxes: L1 and GIoU (criterion.py:103-121) ----------------------------------------------------------
  float s1 = 0.f, s2 = 0.f;
  for (int e = threadIdx.x; e < d.T1; e += blockDim.x) {
    if (!pair_valid(a, e)) continue;
    const PairBox p = load_pair(a, d, l, e);
    s1 += fabsf(p.c - p.tc) + fabsf(p.l - p.tl);
    const float sx0 = p.c - 0.5f * p.l, sx1 = p.c + 0.5f * p.l, tx0 = p.tc - 0.5f * p.tl, tx1 = p.tc + 0.5f * p.tl;
    const float inter = fmaxf(fminf(sx1, tx1) - fmaxf(sx0, tx0), 0.f);
    const float uni = (sx1 - sx0) + (tx1 - tx0) - inter;
    const float area = fmaxf(fmaxf(sx1, tx1) - fminf(sx0, tx0), 0.f);
    s2 += 1.f - (inter / (uni + kEps) - (area - uni) / (area + kEps));
  }
  const float loss_bbox = block_sum(s1, red) / num_boxes;
  const float loss_giou = block_sum(s2, red) / num_boxes;
  // ---- self-IoU of the matched predictions of one video (criterion.py:123-130) ------------------------------------
  s = 0.f;
  for (int v = threadIdx.x; v < d.B; v += blockDim.x) {
    const int e0 = (int)a.ent_start[v], e1 = pairs_end(a, v);
    float acc = 0.f;
    for (int i = e0; i < e1; ++i) {
      const PairBox pi = load_pair(a, d, l, i);
      for (int j = i + 1; j < e1; ++j) {
        const PairBox pj = load_pair(a, d, l, j);
        float g0, g1;
        acc += iou_grad(pi.c - 0.5f * pi.l, pi.c + 0.5f * pi.l, pj.c - 0.5f * pj.l, pj.c + 0.5f * pj.l, g0, g1);
      }
    }
    const float cnt = (float)(e1 - e0);
    s += acc / (0.5f * cnt * (cnt - 1.f));                              // 0/0 = nan for a single match, as the reference
  }
  const float loss_self = block_sum(s, red);
  // ---- cardinality error (criterion.py:88-99; logging only) -------------------------------------------------------
  s = 0.f;
  for (int b = threadIdx.x; b < d.B; b += blockDim.x) {
    float n = 0.f;
    for (int q = 0; q < d.Q; ++q) {
      const float *x = a.logits + (((int64_t)l * d.B + b) * d.Q + q) * d.NC;
      int am = 0;
      for (int c = 1; c < d.NC; ++c) am = x[c] > x[am] ? c : am;        // first maximal index, as torch.argmax
      n += am != d.NC - 1 ? 1.f : 0.f;
    }
    s += fabsf(n - (float)a.gt_counts[b]);
  }
  const float card = block_sum(s, red) / (float)d.B;
  if (threadIdx.x == 0) {
    float *o = losses + l * kLosses;
    o[0] = loss_ce; o[1] = loss_counter; o[2] = loss_bbox; o[3] = loss_giou; o[4] = loss_self; o[5] = card;
  }
}

__global__ void __launch_bounds__(1024) k_criterion_bwd(CritArgs a, CritDims d, const float *__restrict__ gl,
                                                        float *__restrict__ g_logits, float *__restrict__ g_counts,
                                                        float *__restrict__ g_boxes) {
  extern __shared__ int cls[];
  const int l = blockIdx.x;
  const float *w = gl + l * kLosses;                                  // upstream gradient of the six loss terms
  build_class_map(cls, a, d, l);
  const int nlog = d.B * d.Q * d.NC;
  const float num_boxes = num_boxes_of(a);
  const float k_ce = w[0] / num_boxes;
  for (int i = threadIdx.x; i < nlog; i += blockDim.x) {
    const float t = (cls[i / d.NC] == i % d.NC) ? 1.f : 0.f;
    float g;
    focal_terms(a.logits[(int64_t)l * nlog + i], t, a.alpha, a.gamma, g);
    g_logits[(int64_t)l * nlog + i] = k_ce * g;
  }
  const float k_cnt = w[1] / (float)(d.B * d.Wd);
  for (int i = threadIdx.x; i < d.B * d.Wd; i += blockDim.x) {
    const int b = i / d.Wd, c = i % d.Wd;
    const int tgt = (int)min(a.gt_counts[b], (int64_t)(d.Wd - 1));
    const float t = c == tgt ? 1.f : 0.f;
    const float x = a.counts[(int64_t)l * d.B * d.Wd + i];
    g_counts[(int64_t)l * d.B * d.Wd + i] = k_cnt * (1.f - a.ccr[c]) * counter_coef(c, tgt, a.beta, a.gau_mask) * (sigmoid_(x) - t);
  }
  float *gb = g_boxes + (int64_t)l * d.B * d.Q * 2;
  for (int i = threadIdx.x; i < d.B * d.Q * 2; i += blockDim.x) gb[i] = 0.f;
  __syncthreads();
  const float k_l1 = w[2] / num_boxes, k_giou = -w[3] / num_boxes;
  for (int e = threadIdx.x; e < d.T1; e += blockDim.x) {
    if (!pair_valid(a, e)) continue;
    const PairBox p = load_pair(a, d, l, e);
    const float sgc = p.c > p.tc ? 1.f : (p.c < p.tc ? -1.f : 0.f), sgl = p.l > p.tl ? 1.f : (p.l < p.tl ? -1.f : 0.f);
    const float sx0 = p.c - 0.5f * p.l, sx1 = p.c + 0.5f * p.l, tx0 = p.tc - 0.5f * p.tl, tx1 = p.tc + 0.5f * p.tl;
    // GIoU = I/U - (A - u)/(A + eps), U = u + eps
    const float di_ = fminf(sx1, tx1) - fmaxf(sx0, tx0);
    const float inter = fmaxf(di_, 0.f);
    const float uni = (sx1 - sx0) + (tx1 - tx0) - inter;
    const float da_ = fmaxf(sx1, tx1) - fminf(sx0, tx0);
    const float area = fmaxf(da_, 0.f);
    const float U = uni + kEps, A = area + kEps;
    const float d_I = 1.f / U, d_u = -inter / (U * U) + 1.f / A, d_A = -(uni + kEps) / (A * A);
    // inter = clamp(min(sx1,tx1) - max(sx0,tx0), 0); union = len_s + len_t - inter; area = clamp(max(sx1,tx1) - min(sx0,tx0), 0)
    const float gi = (di_ >= 0.f ? 1.f : 0.f) * (d_I - d_u);            // total d giou / d (rb - lt)
    const float ga = (da_ >= 0.f ? 1.f : 0.f) * d_A;
    const float w_rb = sx1 < tx1 ? 1.f : (sx1 == tx1 ? 0.5f : 0.f), w_lt = sx0 > tx0 ? 1.f : (sx0 == tx0 ? 0.5f : 0.f);
    const float w_rb2 = sx1 > tx1 ? 1.f : (sx1 == tx1 ? 0.5f : 0.f), w_lt2 = sx0 < tx0 ? 1.f : (sx0 == tx0 ? 0.5f : 0.f);
    float g1 = gi * w_rb + ga * w_rb2 + d_u;                            // d giou / d sx1
    float g0 = -gi * w_lt - ga * w_lt2 - d_u;                           // d giou / d sx0
    g1 *= k_giou; g0 *= k_giou;
    // self-IoU with every other matched prediction of the same video
    const int v = (int)a.vid[e];
    const int e0 = (int)a.ent_start[v], e1 = pairs_end(a, v);
    const float cnt = (float)(e1 - e0);
    const float k_self = w[4] / (0.5f * cnt * (cnt - 1.f));
    for (int j = e0; j < e1; ++j) {
      if (j == e) continue;
      const PairBox pj = load_pair(a, d, l, j);
      float h0, h1;
      iou_grad(sx0, sx1, pj.c - 0.5f * pj.l, pj.c + 0.5f * pj.l, h0, h1);
      g0 += k_self * h0;
      g1 += k_self * h1;
    }
    const int64_t r = ((int64_t)a.vid[e]) * d.Q + a.mq[(int64_t)l * d.T1 + e];
    gb[r * 2] = k_l1 * sgc + (g0 + g1);                                 // x0 = c - l/2, x1 = c + l/2
    gb[r * 2 + 1] = k_l1 * sgl + 0.5f * (g1 - g0);
  }
}

int check_args(const char *what, const CritDims &d) {
  if (d.nl <= 0 || d.B <= 0 || d.Q <= 0 || d.NC <= 0 || d.Wd <= 0 || d.T1 < 0 || d.G < 0)
    return fail(GVL_EINVAL, "%s: bad sizes", what);
  if ((size_t)d.B * d.Q * sizeof(int) > 96 * 1024)
    return fail(GVL_EINVAL, "%s: B*Q = %d exceeds the on-chip class map (24576)", what, d.B * d.Q);
  return 0;
}

}  // namespace

extern "C" {

int gvl_match_cost_f32(const float *pred_logits, const float *pred_boxes, const int64_t *tgt_labels,
                       const float *tgt_boxes, int n_layers, int B, int Q, int n_classes, int G, float w_class,
                       float w_bbox, float w_giou, float alpha, float gamma, float *cost, int *ok, void *stream) {
  const CritDims d = {n_layers, B, Q, n_classes, 1, 0, G};
  if (n_layers <= 0 || B < 0 || Q < 0 || n_classes <= 0 || G < 0) return fail(GVL_EINVAL, "gvl_match_cost_f32: bad sizes");
  const int64_t total = (int64_t)n_layers * B * Q * G;
  if (total == 0) return 0;
  if (!pred_logits || !pred_boxes || !tgt_labels || !tgt_boxes || !cost || !ok)
    return fail(GVL_EINVAL, "gvl_match_cost_f32: null pointer");
  int64_t blocks = (total + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  return gvl::launch(GVL_PROF_MATCH_COST, Q, B, "k_match_cost", k_match_cost, dim3((unsigned)blocks), dim3(256), 0,
                     (hipStream_t)stream, pred_logits, pred_boxes, tgt_labels, tgt_boxes, d, w_class, w_bbox, w_giou,
                     alpha, gamma, cost, ok);
}

int gvl_match_cost_padded_f32(const float *pred_logits, const float *pred_boxes, const int64_t *tgt_labels,
                              const float *tgt_boxes, const int64_t *gt_counts, int n_layers, int B, int Q,
                              int n_classes, int slots, float w_class, float w_bbox, float w_giou, float alpha,
                              float gamma, float *cost, int *ok, void *stream) {
  const CritDims d = {n_layers, B, Q, n_classes, 1, 0, slots};
  if (n_layers <= 0 || B < 0 || Q < 0 || n_classes <= 0 || slots < 0)
    return fail(GVL_EINVAL, "gvl_match_cost_padded_f32: bad sizes");
  const int64_t total = (int64_t)n_layers * B * Q * slots;
  if (total == 0) return 0;
  if (!pred_logits || !pred_boxes || !tgt_labels || !tgt_boxes || !gt_counts || !cost || !ok)
    return fail(GVL_EINVAL, "gvl_match_cost_padded_f32: null pointer");
  int64_t blocks = (total + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  return gvl::launch(GVL_PROF_MATCH_COST, Q, B, "k_match_cost_padded", k_match_cost_padded, dim3((unsigned)blocks),
                     dim3(256), 0, (hipStream_t)stream, pred_logits, pred_boxes, tgt_labels, tgt_boxes, gt_counts, d,
                     w_class, w_bbox, w_giou, alpha, gamma, cost, ok);
}

static CritArgs make_args(const float *logits, const float *counts, const float *boxes, const int64_t *mq,
                          const int64_t *mt, const int64_t *vid, const int64_t *tbase, const int64_t *ent_start,
                          const int64_t *tgt_labels, const float *tgt_boxes, const int64_t *gt_counts,
                          const float *ccr, float num_boxes, float alpha, float gamma, float beta, int gau_mask,
                          const int64_t *pair_count, const float *num_boxes_dev) {
  return {logits, counts, boxes, mq, mt, vid, tbase, ent_start, tgt_labels, tgt_boxes, gt_counts, ccr,
          num_boxes, alpha, gamma, beta, gau_mask, pair_count, num_boxes_dev};
}

int gvl_set_criterion_forward_f32(const float *pred_logits, const float *pred_count, const float *pred_boxes,
                                  const int64_t *match_q, const int64_t *match_t, const int64_t *pair_video,
                                  const int64_t *pair_target_base, const int64_t *video_pair_start,
                                  const int64_t *tgt_labels, const float *tgt_boxes, const int64_t *gt_counts,
                                  const float *counter_class_rate, int n_layers, int B, int Q, int n_classes,
                                  int count_bins, int n_pairs, int G, float num_boxes, float focal_alpha,
                                  float focal_gamma, float lloss_beta, int lloss_gau_mask,
                                  const int64_t *video_pair_count, const float *num_boxes_dev, float *losses,
                                  void *stream) {
  const CritDims d = {n_layers, B, Q, n_classes, count_bins, n_pairs, G};
  if (int rc = check_args("gvl_set_criterion_forward_f32", d)) return rc;
  if (!pred_logits || !pred_count || !pred_boxes || !gt_counts || !counter_class_rate || !losses ||
      (n_pairs > 0 && (!match_q || !match_t || !pair_video || !pair_target_base || !tgt_labels || !tgt_boxes)) ||
      !video_pair_start)
    return fail(GVL_EINVAL, "gvl_set_criterion_forward_f32: null pointer");
  const CritArgs a = make_args(pred_logits, pred_count, pred_boxes, match_q, match_t, pair_video, pair_target_base,
                               video_pair_start, tgt_labels, tgt_boxes, gt_counts, counter_class_rate, num_boxes,
                               focal_alpha, focal_gamma, lloss_beta, lloss_gau_mask, video_pair_count, num_boxes_dev);
  const size_t lds = (size_t)B * Q * sizeof(int);
  if (int rc = gvl::ensure_lds(k_criterion_fwd, lds)) return rc;
  return gvl::launch(GVL_PROF_CRITERION, Q, B, "k_criterion_fwd", k_criterion_fwd, dim3(n_layers), dim3(1024), lds,
                     (hipStream_t)stream, a, d, losses);
}

int gvl_set_criterion_backward_f32(const float *pred_logits, const float *pred_count, const float *pred_boxes,
                                   const int64_t *match_q, const int64_t *match_t, const int64_t *pair_video,
                                   const int64_t *pair_target_base, const int64_t *video_pair_start,
                                   const int64_t *tgt_labels, const float *tgt_boxes, const int64_t *gt_counts,
                                   const float *counter_class_rate, int n_layers, int B, int Q, int n_classes,
                                   int count_bins, int n_pairs, int G, float num_boxes, float focal_alpha,
                                   float focal_gamma, float lloss_beta, int lloss_gau_mask,
                                   const int64_t *video_pair_count, const float *num_boxes_dev,
                                   const float *grad_losses, float *grad_logits, float *grad_count, float *grad_boxes,
                                   void *stream) {
  const CritDims d = {n_layers, B, Q, n_classes, count_bins, n_pairs, G};
  if (int rc = check_args("gvl_set_criterion_backward_f32", d)) return rc;
  if (!pred_logits || !pred_count || !pred_boxes || !gt_counts || !counter_class_rate || !grad_losses ||
      !grad_logits || !grad_count || !grad_boxes || !video_pair_start ||
      (n_pairs > 0 && (!match_q || !match_t || !pair_video || !pair_target_base || !tgt_labels || !tgt_boxes)))
    return fail(GVL_EINVAL, "gvl_set_criterion_backward_f32: null pointer");
  const CritArgs a = make_args(pred_logits, pred_count, pred_boxes, match_q, match_t, pair_video, pair_target_base,
                               video_pair_start, tgt_labels, tgt_boxes, gt_counts, counter_class_rate, num_boxes,
                               focal_alpha, focal_gamma, lloss_beta, lloss_gau_mask, video_pair_count, num_boxes_dev);
  const size_t lds = (size_t)B * Q * sizeof(int);
  if (int rc = gvl::ensure_lds(k_criterion_bwd, lds)) return rc;
  return gvl::launch(GVL_PROF_CRITERION, Q, B, "k_criterion_bwd", k_criterion_bwd, dim3(n_layers), dim3(1024), lds,
                     (hipStream_t)stream, a, d, grad_losses, grad_logits, grad_count, grad_boxes);
}

}  // extern "C"

// ------------------------------------------------------------------------------------------------------
// PositionEmbeddingSine.forward (pdvc/position_encoding.py:38-64) as one launch per pyramid level:
//   x_t   = cumsum_t(valid)                     valid = !mask
//   x_t   = (x_t - 0.5) / (x_last + 1e-6) * 2 pi
//   out[n, 2i,   t] = sin(x_t / dim_t[2i]),  out[n, 2i+1, t] = cos(x_t / dim_t[2i+1])      (F sine channels)
//   out[n, F + c, t] = dur[n, c]                                                              (duration embedding)
// ~20 PyTorch kernels per level otherwise.  A few workgroups per video; dim_t is passed in (computed once by torch.pow so
// that the table is the reference's bit for bit).
// ------------------------------------------------------------------------------------------------------
namespace {

__global__ void __launch_bounds__(256) k_pos_embed_sine(const unsigned char *__restrict__ mask, const float *__restrict__ dim_t,
                                                        const float *__restrict__ dur, int T, int F, int Cd, float scale,
                                                        float *__restrict__ out) {
  extern __shared__ float xs[];                     // normalised position of every frame
  __shared__ int wave_tot[4];
  __shared__ int carry;
  const int n = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (int base = 0; base < T; base += blockDim.x) {
    const int t = base + threadIdx.x;
    const int v = (t < T && !mask[(int64_t)n * T + t]) ? 1 : 0;
    int incl = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int u = __shfl_up(incl, o, 64);
      if (lane >= o) incl += u;
    }
    if (lane == 63) wave_tot[wave] = incl;
    __syncthreads();
    int pre = carry;
    for (int k = 0; k < wave; ++k) pre += wave_tot[k];
    if (t < T) xs[t] = (float)(pre + incl);
    __syncthreads();
    if (threadIdx.x == blockDim.x - 1) carry = pre + incl;
    __syncthreads();
  }
  const float last = xs[T - 1];
  const float denom = last + 1e-6f;
  float *o = out + (int64_t)n * (F + Cd) * T;
  // the table of one video is shared out over gridDim.y workgroups (each repeats the cheap scan above): 25 600 precise
  // sinf / cosf per video on ONE workgroup took 30 us per level
  const int stride = blockDim.x * gridDim.y, first = blockIdx.y * blockDim.x + threadIdx.x;
  for (int idx = first; idx < F * T; idx += stride) {
    const int c = idx / T, t = idx % T;
    const float x = (xs[t] - 0.5f) / denom * scale;
    const float p = x / dim_t[c];
    o[idx] = (c & 1) ? cosf(p) : sinf(p);
  }
  for (int idx = first; idx < Cd * T; idx += stride) o[(int64_t)F * T + idx] = dur[(int64_t)n * Cd + idx / T];
}

}  // namespace

extern "C" int gvl_pos_embed_sine_f32(const unsigned char *mask, const float *dim_t, const float *dur_embed, int N, int T,
                                      int n_sine, int n_dur, float scale, float *out, void *stream) {
  if (N < 0 || T <= 0 || n_sine <= 0 || n_dur < 0 || (n_sine & 1))
    return fail(GVL_EINVAL, "gvl_pos_embed_sine_f32: bad sizes");
  if (N == 0) return 0;
  if (!mask || !dim_t || !out || (n_dur > 0 && !dur_embed)) return fail(GVL_EINVAL, "gvl_pos_embed_sine_f32: null pointer");
  const size_t lds = (size_t)T * sizeof(float);
  if (int rc = gvl::ensure_lds(k_pos_embed_sine, lds)) return rc;
  int slices = (int)(((int64_t)(n_sine + n_dur) * T + 2047) / 2048);          // ~8 outputs per thread
  slices = slices < 1 ? 1 : (slices > 32 ? 32 : slices);
  return gvl::launch(GVL_PROF_POS_EMBED, T, N, "k_pos_embed_sine", k_pos_embed_sine, dim3(N, slices), dim3(256), lds,
                     (hipStream_t)stream, mask, dim_t, dur_embed, T, n_sine, n_dur, scale, out);
}

// ------------------------------------------------------------------------------------------------------
// Column sums of a row-major matrix: out[c] = sum_r x[r, c] -- the bias gradient of every nn.Linear on the path
// (rows = B*Q or B*S tokens).  PyTorch's generic reduction over the slow axis takes 13-33 us on these shapes
// (4800 x 512 ... 4800 x 2048), 48 times per train step.  Here: workgroup = (256-column block, 64-row chunk), each
// wave streams 16 rows of 1 KB with every load in flight, partials meet in LDS and leave as one contiguous float
// atomic per thread (2 cache lines per wave instruction) into the zero-filled output.
// ------------------------------------------------------------------------------------------------------
namespace {

constexpr int kCsRows = 64;

__global__ void __launch_bounds__(256) k_col_sum(const float *__restrict__ x, int64_t ld, int R, int C,
                                                 float *__restrict__ out) {
  __shared__ float part[4][256];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c0 = blockIdx.x * 256 + 4 * lane;
  const int r0 = blockIdx.y * kCsRows;
  const int r1 = min(R, r0 + kCsRows);
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  if (c0 < C) {                                   // C is a multiple of 4 (checked by the caller)
#pragma unroll 4
    for (int r = r0 + wave; r < r1; r += 4) {
      const float4 v = *reinterpret_cast<const float4 *>(x + (int64_t)r * ld + c0);
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
  }
  part[wave][4 * lane + 0] = acc.x; part[wave][4 * lane + 1] = acc.y;
  part[wave][4 * lane + 2] = acc.z; part[wave][4 * lane + 3] = acc.w;
  __syncthreads();
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c < C) atomicAdd(out + c, part[0][threadIdx.x] + part[1][threadIdx.x] + part[2][threadIdx.x] + part[3][threadIdx.x]);
}

// any width / row stride: one column per thread, 256 consecutive columns per workgroup
__global__ void __launch_bounds__(256) k_col_sum_scalar(const float *__restrict__ x, int64_t ld, int R, int C,
                                                        float *__restrict__ out) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  const int r0 = blockIdx.y * kCsRows, r1 = min(R, r0 + kCsRows);
  float acc = 0.f;
#pragma unroll 8
  for (int r = r0; r < r1; ++r) acc += x[(int64_t)r * ld + c];
  atomicAdd(out + c, acc);
}


// ---------------------------------------------------------------------------------------------------------------------
// Caption loss over the vocabulary (LSTM_DSA.py:48-52 applied to :121-123): per (caption row, token step) r the masked
// log-probability of the target word, out[r] = w[r] (x[r][t_r] - logsumexp_v x[r][v]) -- what
// `(F.log_softmax(logits, 2).gather(2, target) * mask)` leaves -- WITHOUT the (R, V) log-prob tensor: one read of the
// logits forward; backward the logits are overwritten by their own gradient g[r] w[r] (onehot - softmax), one read and
// one write (autograd's chain: log_softmax (read + write 150 MB at cfg A), gather, its zero-filled scatter backward,
// log_softmax backward: 0.33 ms of the train step).  Rows with w = 0 (padding, steps past the caption's end: about half
// of the padded layout) are not read at all.  One workgroup per row.
constexpr int kCeThreads = 256;

__device__ inline void ce_block_max_sum(float &m, float &s) {          // (m, s): running maximum and sum exp(x - m)
  __shared__ float sm[kCeThreads / 64], ss[kCeThreads / 64];
#pragma unroll
  for (int o = 32; o; o >>= 1) {
    const float m2 = __shfl_xor(m, o, 64), s2 = __shfl_xor(s, o, 64);
    const float mn = fmaxf(m, m2);
    s = (mn == -INFINITY) ? 0.f : s * __expf(m - mn) + s2 * __expf(m2 - mn);
    m = mn;
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) { sm[wave] = m; ss[wave] = s; }
  __syncthreads();
  float mm = sm[0], sum = ss[0];
#pragma unroll
  for (int k = 1; k < kCeThreads / 64; ++k) {
    const float mn = fmaxf(mm, sm[k]);
    sum = (mn == -INFINITY) ? 0.f : sum * __expf(mm - mn) + ss[k] * __expf(sm[k] - mn);
    mm = mn;
  }
  m = mm;
  s = sum;
}

__global__ void __launch_bounds__(kCeThreads) k_ce_rows_fwd(const float *__restrict__ logits, int64_t ld, int V,
                                                            const int64_t *__restrict__ target,
                                                            const float *__restrict__ w, float *__restrict__ out,
                                                            float *__restrict__ lse) {
  const int row = blockIdx.x;
  const float wr = w[row];
  if (wr == 0.f) {                                                     // (uniform for the workgroup)
    if (threadIdx.x == 0) { out[row] = 0.f; lse[row] = 0.f; }
    return;
  }
  const float *x = logits + (int64_t)row * ld;
  float m = -INFINITY, s = 0.f;
  for (int v = threadIdx.x; v < V; v += kCeThreads) {
    const float xv = x[v];
    if (xv > m) { s = s * __expf(m - xv) + 1.f; m = xv; }              // (exp(-inf - xv) = 0 on the first element)
    else s += __expf(xv - m);
  }
  ce_block_max_sum(m, s);
  if (threadIdx.x == 0) {
    const float l = m + __logf(s);
    lse[row] = l;
    out[row] = wr * (x[target[row]] - l);
  }
}

__global__ void __launch_bounds__(kCeThreads) k_ce_rows_bwd(float *__restrict__ logits, int64_t ld, int V,
                                                            const int64_t *__restrict__ target,
                                                            const float *__restrict__ w, const float *__restrict__ g,
                                                            const float *__restrict__ lse, float *__restrict__ amax) {
  const int row = blockIdx.x;
  const float coef = g[row] * w[row];
  float *x = logits + (int64_t)row * ld;
  // |gradient row| <= |coef| (one-hot minus a probability): the row bound the split-fp16 products behind this kernel scale by;
  // the padding columns [V, ld) become zeros (those products read whole 16-byte pieces / K stages of a row)
  if (threadIdx.x == 0 && amax) amax[row] = fabsf(coef);
  if ((int)threadIdx.x < (int)ld - V) x[V + threadIdx.x] = 0.f;
  if (coef == 0.f) {
    for (int v = threadIdx.x; v < V; v += kCeThreads) x[v] = 0.f;
    return;
  }
  const float l = lse[row];
  const int t = (int)target[row];
  for (int v = threadIdx.x; v < V; v += kCeThreads) {
    const float p = __expf(x[v] - l);
    x[v] = coef * ((v == t ? 1.f : 0.f) - p);
  }
}

}  // namespace

extern "C" int gvl_col_sum_f32(const float *x, int ld, int R, int C, float *out, void *stream) {
  if (R < 0 || C < 0 || ld < C) return fail(GVL_EINVAL, "gvl_col_sum_f32: bad sizes R=%d C=%d ld=%d", R, C, ld);
  if (C == 0) return 0;
  if (!out || (R > 0 && !x)) return fail(GVL_EINVAL, "gvl_col_sum_f32: null pointer");
  hipStream_t st = (hipStream_t)stream;
  if (int rc = gvl::zero_fill(out, (size_t)C * sizeof(float), st)) return rc;
  if (R == 0) return 0;
  const dim3 grid((C + 255) / 256, (R + kCsRows - 1) / kCsRows);
  if (!(C & 3) && !(ld & 3) && !((uintptr_t)x & 15))       // rows of float4
    return gvl::launch(GVL_PROF_COL_SUM, R, C, "k_col_sum", k_col_sum, grid, dim3(256), 0, st, x, (int64_t)ld, R, C, out);
  return gvl::launch(GVL_PROF_COL_SUM, R, C, "k_col_sum_scalar", k_col_sum_scalar, grid, dim3(256), 0, st, x,
                     (int64_t)ld, R, C, out);
}

extern "C" int gvl_ce_rows_forward_f32(const float *logits, int64_t ld, int R, int V, const int64_t *target,
                                       const float *weight, float *out, float *lse, void *stream) {
  if (R < 0 || V <= 0 || ld < V) return fail(GVL_EINVAL, "gvl_ce_rows_forward_f32: bad sizes R=%d V=%d", R, V);
  if (R == 0) return 0;
  if (!logits || !target || !weight || !out || !lse) return fail(GVL_EINVAL, "gvl_ce_rows_forward_f32: null pointer");
  return gvl::launch(GVL_PROF_CRITERION, R, V, "k_ce_rows_fwd", k_ce_rows_fwd, dim3(R), dim3(kCeThreads), 0,
                     (hipStream_t)stream, logits, ld, V, target, weight, out, lse);
}

extern "C" int gvl_ce_rows_backward_f32(float *logits, int64_t ld, int R, int V, const int64_t *target,
                                        const float *weight, const float *grad_out, const float *lse, float *amax, void *stream) {
  if (R < 0 || V <= 0 || ld < V || ld - V > kCeThreads) return fail(GVL_EINVAL, "gvl_ce_rows_backward_f32: bad sizes R=%d V=%d", R, V);
  if (R == 0) return 0;
  if (!logits || !target || !weight || !grad_out || !lse) return fail(GVL_EINVAL, "gvl_ce_rows_backward_f32: null pointer");
  return gvl::launch(GVL_PROF_CRITERION, R, V, "k_ce_rows_bwd", k_ce_rows_bwd, dim3(R), dim3(kCeThreads), 0,
                     (hipStream_t)stream, logits, ld, V, target, weight, grad_out, lse, amax);
}

// ---------------------------------------------------------------------------------------------------------------------
// TRAINING: the compact (query, caption) pair rows of the captioner on padded targets (gvl_amd/pdvc.py:
// caption_prediction_layers_padded; pdvc.py:540-573 of the reference gathers the matched queries' hidden states and captions
// per layer).  Row r of layer k is the r-th matched pair in (video, slot) order: its video from the per-video pair counts, its
// slot, the matched query and target of that layer, the caption row and its mask -- ~45 index operations of a few hundred
// elements each as PyTorch ops (cumsum, searchsorted, gathers, cats, clamps, where, repeats), one launch here.
namespace {
constexpr int kCapRowLayers = 8;
struct CapRowsParams {
  const int64_t *pair_count;             // (N) matched pairs per video
  const int64_t *q[kCapRowLayers];       // per layer: (N * G1) matched query of (video, slot)
  const int64_t *t[kCapRowLayers];       //            (N * G1) matched video-local target
  const int64_t *cap_tensor;             // (N, slots, cap_len)
  const float *cap_mask;                 // (N, slots, cap_len)
  int nl, N, G1, R, Nq, slots, cap_len;
  int64_t *flat, *row_video, *seq;       // (nl * R), (nl * R), (nl * R, cap_len)
  float *mask, *denom;                   // (nl * R, cap_len), (1): N * max(1, max pair count)
};
__global__ void __launch_bounds__(256) k_caption_rows(const CapRowsParams p) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i == 0) {
    int64_t mx = 1;
    for (int v = 0; v < p.N; ++v) mx = p.pair_count[v] > mx ? p.pair_count[v] : mx;
    *p.denom = (float)((int64_t)p.N * mx);
  }
  if (i >= p.nl * p.R) return;
  const int k = i / p.R, r = i % p.R;
  // the video of the r-th pair: the first v whose inclusive pair count exceeds r (searchsorted(right = True))
  int64_t incl = 0, before = 0;
  int v = 0;
  for (; v < p.N; ++v) {
    before = incl;
    incl += p.pair_count[v];
    if (incl > r) break;
  }
  const bool used = v < p.N;
  if (!used) { v = p.N - 1; before = incl - p.pair_count[v]; }
  int64_t slot = r - before;
  slot = slot < 0 ? 0 : (slot > p.G1 - 1 ? p.G1 - 1 : slot);
  const int64_t e = (int64_t)v * p.G1 + slot;
  int64_t q = p.q[k][e], t = p.t[k][e];
  q = q < 0 ? 0 : q;
  t = t < 0 ? 0 : t;
  p.flat[i] = ((int64_t)k * p.N + v) * p.Nq + q;
  p.row_video[i] = used ? v : -1;
  const int64_t src = ((int64_t)v * p.slots + t) * p.cap_len;
  for (int c = 0; c < p.cap_len; ++c) {
    p.seq[(int64_t)i * p.cap_len + c] = used ? p.cap_tensor[src + c] : 0;
    p.mask[(int64_t)i * p.cap_len + c] = used ? p.cap_mask[src + c] : 0.f;
  }
}
}  // namespace

extern "C" int gvl_caption_rows(const int64_t *pair_count, const int64_t *const *q_layers, const int64_t *const *t_layers, int n_layers,
                                int N, int G1, int R, int Nq, int slots, int cap_len, const int64_t *cap_tensor, const float *cap_mask,
                                int64_t *flat, int64_t *row_video, int64_t *seq, float *mask, float *denom, void *stream) {
  if (!pair_count || !q_layers || !t_layers || !cap_tensor || !cap_mask || !flat || !row_video || !seq || !mask || !denom)
    return gvl::fail(GVL_EINVAL, "gvl_caption_rows: null pointer");
  if (n_layers < 1 || n_layers > kCapRowLayers || N < 1 || G1 < 1 || R < 1 || Nq < 1 || slots < 1 || cap_len < 1)
    return gvl::fail(GVL_EINVAL, "gvl_caption_rows: 1 .. %d layers, positive sizes (got layers=%d N=%d G1=%d R=%d)", kCapRowLayers, n_layers, N, G1, R);
  CapRowsParams p;
  p.pair_count = pair_count;
  for (int k = 0; k < n_layers; ++k) {
    if (!q_layers[k] || !t_layers[k]) return gvl::fail(GVL_EINVAL, "gvl_caption_rows: null match array (layer %d)", k);
    p.q[k] = q_layers[k];
    p.t[k] = t_layers[k];
  }
  p.cap_tensor = cap_tensor; p.cap_mask = cap_mask;
  p.nl = n_layers; p.N = N; p.G1 = G1; p.R = R; p.Nq = Nq; p.slots = slots; p.cap_len = cap_len;
  p.flat = flat; p.row_video = row_video; p.seq = seq; p.mask = mask; p.denom = denom;
  hipLaunchKernelGGL(k_caption_rows, dim3((unsigned)((n_layers * R + 255) / 256)), dim3(256), 0, (hipStream_t)stream, p);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : gvl::fail((int)e, "gvl_caption_rows: launch failed: %s", hipGetErrorString(e));
}

