/*
 * oracle/msda_ref.c -- CPU restatement of the reference's multi-scale deformable attention op.
 *
 * TEST INFRASTRUCTURE ONLY.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this file's shared object; the product (gvl_amd/) never does.
 *
 * What it restates (all paths relative to /root/reference):
 *   pad_mode 0 "zeros"  : the CUDA op behind MSDeformAttnFunction
 *       forward  pdvc/ops/src/cuda/ms_deform_im2col_cuda.cuh:238-300 (+ bilinear :34-85)
 *       backward pdvc/ops/src/cuda/ms_deform_im2col_cuda.cuh:407-511 (+ bilinear :88-160)
 *       host wrapper (zero-initialised outputs, (B,Lq,M*D) layout)
 *                pdvc/ops/src/cuda/ms_deform_attn_cuda.cu:20-80, :83-153
 *   pad_mode 1 "border" : the pure-PyTorch fallback ms_deform_attn_core_pytorch
 *       pdvc/ops/functions/ms_deform_attn_func.py:44-71, i.e. per level
 *       F.grid_sample(bilinear, padding_mode='border', align_corners=False) (:61-62);
 *       coordinate un-normalisation / clipping / bilinear gradient follow the published
 *       ATen grid_sampler_2d CPU algorithm (torch 2.10; not vendored in the reference).
 *   sample ("return_value=True", func.py:67-68) used by MSDeformAttnCap
 *       pdvc/ops/modules/ms_deform_attn_for_caption.py:124-125.
 *
 * The CUDA sources themselves are unbuildable here (nvcc, THC headers, removed ATen APIs:
 * SURVEY.md section 8c), so the zeros mode is pinned by golden vectors produced from the
 * imported Python core with padding_mode='zeros' (tests/golden/make_golden.py) and the border
 * mode by vectors from the unmodified Python core.
 *
 * Layouts (row-major, contiguous):
 *   value  (B,S,M,D)   shapes (L,2) int64 = (H_l,W_l)   lsi (L) int64 start row per level
 *   loc    (B,Q,M,L,P,2) with [...,0]=x (width), [...,1]=y (height), normalised to [0,1]
 *   w      (B,Q,M,L,P)   out (B,Q,M*D)   sample (B*M, D, Q, L, P)
 * Arithmetic is carried out in the element type (float or double), like the reference.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define GVL_PAD_ZEROS 0
#define GVL_PAD_BORDER 1

#define DEFINE_MSDA(T, SUF, FLOOR)                                                              \
                                                                                                \
/* pixel coordinate of a normalised location.  zeros: cuh:286-287.  border: func.py:52 then    \
 * ATen grid_sampler_unnormalize(align_corners=False) ((g+1)*size-1)/2 followed by             \
 * clip_coordinates to [0,size-1]; *dmul receives d(pixel)/d(loc) (0 where clipped). */        \
static inline T pix_##SUF(T loc, int64_t size, int pad_mode, T *dmul) {                         \
  if (pad_mode == GVL_PAD_ZEROS) {                                                              \
    *dmul = (T)size;                                                                            \
    return loc * (T)size - (T)0.5;                                                              \
  }                                                                                             \
  T g = (T)2 * loc - (T)1;                                                                      \
  T x = ((g + (T)1) * (T)size - (T)1) / (T)2;                                                   \
  if (x <= (T)0) { *dmul = (T)0; return (T)0; }                                                 \
  T mx = (T)(size - 1);                                                                         \
  if (x >= mx) { *dmul = (T)0; return mx; }                                                     \
  *dmul = (T)size;                                                                              \
  return x;                                                                                     \
}                                                                                               \
                                                                                                \
/* One bilinear tap set.  Fills row indices (or -1) and the four weights. cuh:39-82 */         \
typedef struct { int64_t i00, i01, i10, i11; T w00, w01, w10, w11; T lh, lw; int valid; }       \
    taps_##SUF;                                                                                 \
                                                                                                \
static inline taps_##SUF taps_at_##SUF(T h, T w, int64_t H, int64_t W, int pad_mode) {          \
  taps_##SUF t;                                                                                 \
  memset(&t, 0, sizeof(t));                                                                     \
  t.i00 = t.i01 = t.i10 = t.i11 = -1;                                                           \
  /* cuh:289: zeros mode skips samples outside (-1,H)x(-1,W) entirely */                       \
  if (pad_mode == GVL_PAD_ZEROS && !(h > (T)-1 && w > (T)-1 && h < (T)H && w < (T)W)) return t; \
  t.valid = 1;                                                                                  \
  int64_t hl = (int64_t)FLOOR(h), wl = (int64_t)FLOOR(w);                                       \
  int64_t hh_ = hl + 1, wh = wl + 1;                                                            \
  T lh = h - (T)hl, lw = w - (T)wl, hh = (T)1 - lh, hw = (T)1 - lw;                             \
  t.lh = lh; t.lw = lw;                                                                         \
  t.w00 = hh * hw; t.w01 = hh * lw; t.w10 = lh * hw; t.w11 = lh * lw;                           \
  if (hl >= 0 && wl >= 0 && hl <= H - 1 && wl <= W - 1) t.i00 = hl * W + wl;                    \
  if (hl >= 0 && wh <= W - 1 && hl <= H - 1 && wh >= 0) t.i01 = hl * W + wh;                    \
  if (hh_ <= H - 1 && wl >= 0 && hh_ >= 0 && wl <= W - 1) t.i10 = hh_ * W + wl;                 \
  if (hh_ <= H - 1 && wh <= W - 1 && hh_ >= 0 && wh >= 0) t.i11 = hh_ * W + wh;                 \
  return t;                                                                                     \
}                                                                                               \
                                                                                                \
int oracle_msda_fwd_##SUF(const T *value, const int64_t *shapes, const int64_t *lsi,            \
                          const T *loc, const T *aw, int B, int S, int M, int D, int L, int Q,  \
                          int P, int pad_mode, T *out) {                                        \
  for (int b = 0; b < B; ++b)                                                                   \
    for (int q = 0; q < Q; ++q)                                                                 \
      for (int m = 0; m < M; ++m) {                                                             \
        T *o = out + (((int64_t)b * Q + q) * M + m) * D;                                        \
        for (int d = 0; d < D; ++d) o[d] = (T)0;                                                \
        int64_t wbase = (((int64_t)b * Q + q) * M + m) * L * P;                                 \
        for (int l = 0; l < L; ++l) {                                                           \
          int64_t H = shapes[2 * l], W = shapes[2 * l + 1];                                     \
          const T *vl = value + ((int64_t)b * S + lsi[l]) * M * D + (int64_t)m * D;             \
          for (int p = 0; p < P; ++p) {                                                         \
            T dm;                                                                               \
            T lx = loc[(wbase + l * P + p) * 2], ly = loc[(wbase + l * P + p) * 2 + 1];         \
            T wgt = aw[wbase + l * P + p];                                                      \
            T w_im = pix_##SUF(lx, W, pad_mode, &dm), h_im = pix_##SUF(ly, H, pad_mode, &dm);   \
            taps_##SUF t = taps_at_##SUF(h_im, w_im, H, W, pad_mode);                           \
            if (!t.valid) continue;                                                             \
            for (int d = 0; d < D; ++d) {                                                       \
              T v1 = t.i00 >= 0 ? vl[t.i00 * M * D + d] : (T)0;                                 \
              T v2 = t.i01 >= 0 ? vl[t.i01 * M * D + d] : (T)0;                                 \
              T v3 = t.i10 >= 0 ? vl[t.i10 * M * D + d] : (T)0;                                 \
              T v4 = t.i11 >= 0 ? vl[t.i11 * M * D + d] : (T)0;                                 \
              o[d] += (t.w00 * v1 + t.w01 * v2 + t.w10 * v3 + t.w11 * v4) * wgt; /* cuh:82,291 */\
            }                                                                                   \
          }                                                                                     \
        }                                                                                       \
      }                                                                                         \
  return 0;                                                                                     \
}                                                                                               \
                                                                                                \
/* return_value=True: unweighted samples, layout (B*M, D, Q, L, P) (func.py:56-68) */          \
int oracle_msda_sample_##SUF(const T *value, const int64_t *shapes, const int64_t *lsi,         \
                             const T *loc, int B, int S, int M, int D, int L, int Q, int P,     \
                             int pad_mode, T *samp) {                                           \
  for (int b = 0; b < B; ++b)                                                                   \
    for (int q = 0; q < Q; ++q)                                                                 \
      for (int m = 0; m < M; ++m) {                                                             \
        int64_t wbase = (((int64_t)b * Q + q) * M + m) * L * P;                                 \
        for (int l = 0; l < L; ++l) {                                                           \
          int64_t H = shapes[2 * l], W = shapes[2 * l + 1];                                     \
          const T *vl = value + ((int64_t)b * S + lsi[l]) * M * D + (int64_t)m * D;             \
          for (int p = 0; p < P; ++p) {                                                         \
            T dm;                                                                               \
            T lx = loc[(wbase + l * P + p) * 2], ly = loc[(wbase + l * P + p) * 2 + 1];         \
            T w_im = pix_##SUF(lx, W, pad_mode, &dm), h_im = pix_##SUF(ly, H, pad_mode, &dm);   \
            taps_##SUF t = taps_at_##SUF(h_im, w_im, H, W, pad_mode);                           \
            for (int d = 0; d < D; ++d) {                                                       \
              T r = (T)0;                                                                       \
              if (t.valid) {                                                                    \
                T v1 = t.i00 >= 0 ? vl[t.i00 * M * D + d] : (T)0;                               \
                T v2 = t.i01 >= 0 ? vl[t.i01 * M * D + d] : (T)0;                               \
                T v3 = t.i10 >= 0 ? vl[t.i10 * M * D + d] : (T)0;                               \
                T v4 = t.i11 >= 0 ? vl[t.i11 * M * D + d] : (T)0;                               \
                r = t.w00 * v1 + t.w01 * v2 + t.w10 * v3 + t.w11 * v4;                          \
              }                                                                                 \
              samp[(((((int64_t)b * M + m) * D + d) * Q + q) * L + l) * P + p] = r;             \
            }                                                                                   \
          }                                                                                     \
        }                                                                                       \
      }                                                                                         \
  return 0;                                                                                     \
}                                                                                               \
                                                                                                \
/* backward; gvalue/gloc/gaw are zero-filled here like cu:121-123 */                           \
int oracle_msda_bwd_##SUF(const T *value, const int64_t *shapes, const int64_t *lsi,            \
                          const T *loc, const T *aw, const T *gout, int B, int S, int M, int D, \
                          int L, int Q, int P, int pad_mode, T *gvalue, T *gloc, T *gaw) {      \
  memset(gvalue, 0, sizeof(T) * (size_t)B * S * M * D);                                         \
  memset(gloc, 0, sizeof(T) * (size_t)B * Q * M * L * P * 2);                                   \
  memset(gaw, 0, sizeof(T) * (size_t)B * Q * M * L * P);                                        \
  for (int b = 0; b < B; ++b)                                                                   \
    for (int q = 0; q < Q; ++q)                                                                 \
      for (int m = 0; m < M; ++m) {                                                             \
        const T *go = gout + (((int64_t)b * Q + q) * M + m) * D;                                \
        int64_t wbase = (((int64_t)b * Q + q) * M + m) * L * P;                                 \
        for (int l = 0; l < L; ++l) {                                                           \
          int64_t H = shapes[2 * l], W = shapes[2 * l + 1];                                     \
          int64_t voff = ((int64_t)b * S + lsi[l]) * M * D + (int64_t)m * D;                    \
          const T *vl = value + voff;                                                           \
          T *gvl = gvalue + voff;                                                               \
          for (int p = 0; p < P; ++p) {                                                         \
            T dmx, dmy;                                                                         \
            T lx = loc[(wbase + l * P + p) * 2], ly = loc[(wbase + l * P + p) * 2 + 1];         \
            T wgt = aw[wbase + l * P + p];                                                      \
            T w_im = pix_##SUF(lx, W, pad_mode, &dmx), h_im = pix_##SUF(ly, H, pad_mode, &dmy); \
            taps_##SUF t = taps_at_##SUF(h_im, w_im, H, W, pad_mode);                           \
            if (!t.valid) continue;                                                             \
            T hh = (T)1 - t.lh, hw = (T)1 - t.lw;                                               \
            T acc_w = (T)0, acc_x = (T)0, acc_y = (T)0;                                         \
            for (int d = 0; d < D; ++d) {                                                       \
              T tg = go[d], tgv = tg * wgt; /* cuh:111 top_grad_value */                        \
              T gh = (T)0, gwd = (T)0, v1 = 0, v2 = 0, v3 = 0, v4 = 0;                          \
              if (t.i00 >= 0) { v1 = vl[t.i00 * M * D + d]; gh -= hw * v1; gwd -= hh * v1;      \
                                gvl[t.i00 * M * D + d] += t.w00 * tgv; }                        \
              if (t.i01 >= 0) { v2 = vl[t.i01 * M * D + d]; gh -= t.lw * v2; gwd += hh * v2;    \
                                gvl[t.i01 * M * D + d] += t.w01 * tgv; }                        \
              if (t.i10 >= 0) { v3 = vl[t.i10 * M * D + d]; gh += hw * v3; gwd -= t.lh * v3;    \
                                gvl[t.i10 * M * D + d] += t.w10 * tgv; }                        \
              if (t.i11 >= 0) { v4 = vl[t.i11 * M * D + d]; gh += t.lw * v4; gwd += t.lh * v4;  \
                                gvl[t.i11 * M * D + d] += t.w11 * tgv; }                        \
              T val = t.w00 * v1 + t.w01 * v2 + t.w10 * v3 + t.w11 * v4;                        \
              acc_w += tg * val;          /* cuh:156-157 */                                     \
              acc_x += dmx * gwd * tgv;   /* cuh:158 (width * grad_w_weight * top_grad_value) */\
              acc_y += dmy * gh * tgv;    /* cuh:159 */                                         \
            }                                                                                   \
            gaw[wbase + l * P + p] = acc_w;                                                     \
            gloc[(wbase + l * P + p) * 2] = acc_x;                                              \
            gloc[(wbase + l * P + p) * 2 + 1] = acc_y;                                          \
          }                                                                                     \
        }                                                                                       \
      }                                                                                         \
  return 0;                                                                                     \
}

DEFINE_MSDA(float, f32, floorf)
DEFINE_MSDA(double, f64, floor)
