// gvl_train_layers.hip -- the residual chains of the encoder / decoder layers in TRAINING:
//
//      y = LayerNorm(x + dropout(sub))         pdvc/deformable_transformer.py:189-199 (encoder: norm1, norm2),
//                                              :266-280 (decoder: norm2 after self-attention, norm1 after the deformable
//                                              cross-attention, norm3 after the FFN)
//
// as ONE forward and ONE backward kernel (+ a small one that adds the workgroups' dgamma / dbeta partials).  PyTorch runs the chain as
// fused_dropout + add + layer_norm (3 launches, 23 us at 4800 x 512) and layer_norm_grad_input + two gamma / beta kernels +
// masked_scale + the residual's gradient add (5 launches, 39 us) -- ten such sites per train step.
//
// One wavefront per row (C <= 1024, C % 4 == 0), rows walked with a grid stride:
//   forward    z = x + keep * sub / (1 - p);  mean, rstd of z (biased variance, two passes over registers, as
//              torch.nn.LayerNorm);  y = (z - mean) rstd gamma + beta;  z, mean, rstd are what the backward keeps
//   backward   xh = (z - mean) rstd;  g = dy gamma;  dz = rstd (g - mean_c(g) - xh mean_c(g xh));  dsub = keep * dz / (1 - p);
//              per-column partial sums of dy xh and dy per WORKGROUP (registers over the wavefront's rows, LDS across the
//              workgroup's wavefronts) -> part (blocks, 2 C), summed in a fixed order by k_rdln_finish -> dgamma | dbeta
//
// Dropout mask: element i of the call is kept iff hash32(i ^ key) >= p 2^32, key = hash32(seed + step * 0x9E3779B9); `step` is
// read from DEVICE memory (a counter the host side advances once per training forward with one tiny kernel), so a step replayed
// from a hipGraph draws new masks every replay; the backward regenerates the mask from the same (seed, step) -- no mask tensor.
// The masks are NOT torch's Philox stream (no two implementations of the reference share masks either); p = 0 is exactly
// LayerNorm(x + sub).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gvl_common.hpp"
#include "gvl_msda.h"

namespace {

using gvl::fail;

constexpr int kMaxV = 4;            // float4 per lane: C <= 1024
constexpr int kWavesPerBlock = 8;   // 512 threads; up to two workgroups per CU (57 KB of LDS each in the backward)

// "lowbias32" integer hash (full avalanche, 2 multiplies)
__device__ __forceinline__ uint32_t hash32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352dU;
  x ^= x >> 15; x *= 0x846ca68bU;
  x ^= x >> 16;
  return x;
}

struct Drop {
  uint32_t key, thr;   // keep iff hash32(index ^ key) >= thr
  float scale;         // 1 / (1 - p)
};

__device__ __forceinline__ Drop make_drop(float p, uint32_t seed, const int64_t *__restrict__ step) {
  Drop d;
  const uint32_t st = step ? (uint32_t)*step : 0u;
  d.key = hash32(seed + st * 0x9E3779B9u);
  // p in [0, 1): thr = round(p 2^32) saturating below 2^32
  const double t = (double)p * 4294967296.0;
  d.thr = t >= 4294967295.0 ? 0xffffffffu : (uint32_t)t;
  d.scale = p > 0.f ? 1.f / (1.f - p) : 1.f;
  return d;
}

__device__ __forceinline__ float4 drop4(const Drop &d, float4 v, uint32_t idx) {
  v.x = hash32(idx ^ d.key) >= d.thr ? v.x * d.scale : 0.f;
  v.y = hash32((idx + 1) ^ d.key) >= d.thr ? v.y * d.scale : 0.f;
  v.z = hash32((idx + 2) ^ d.key) >= d.thr ? v.z * d.scale : 0.f;
  v.w = hash32((idx + 3) ^ d.key) >= d.thr ? v.w * d.scale : 0.f;
  return v;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// x / sub row r = (b, q) with b = r / Q: element offset b * sb + q * sq (a transposed view of nn.MultiheadAttention's
// (Q, B, C) output and the batch-expanded query embedding of the first decoder layer are read in place)
__global__ void __launch_bounds__(64 * kWavesPerBlock) k_rdln_fwd(
    const float *__restrict__ x, int64_t x_sb, int64_t x_sq, const float *__restrict__ sub, int64_t sub_sb, int64_t sub_sq,
    int Q, int R, int C, const float *__restrict__ gamma, const float *__restrict__ beta, float eps, float p, uint32_t seed,
    const int64_t *__restrict__ step, float *__restrict__ y, float *__restrict__ z, float *__restrict__ mean_out,
    float *__restrict__ rstd_out, int64_t *__restrict__ step_used, const float *__restrict__ pos, int64_t pos_sb, int64_t pos_sq,
    float *__restrict__ amax_y, float *__restrict__ amax_ypos) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n4 = C >> 2;
  const Drop d = make_drop(p, seed, step);
  // the step this forward drew its masks for, kept with the saved tensors: the backward regenerates the masks from THIS number,
  // not from the live counter, which a second forward may have advanced in between
  if (step_used && blockIdx.x == 0 && threadIdx.x == 0) *step_used = step ? *step : 0;
  const float4 *g4 = reinterpret_cast<const float4 *>(gamma), *b4 = reinterpret_cast<const float4 *>(beta);
  for (int row = blockIdx.x * kWavesPerBlock + wave; row < R; row += gridDim.x * kWavesPerBlock) {
    const int rb = row / Q, rq = row % Q;
    const float4 *xr = reinterpret_cast<const float4 *>(x + (int64_t)rb * x_sb + (int64_t)rq * x_sq);
    const float4 *sr = reinterpret_cast<const float4 *>(sub + (int64_t)rb * sub_sb + (int64_t)rq * sub_sq);
    float4 v[kMaxV];
    float sum = 0.f;
#pragma unroll
    for (int k = 0; k < kMaxV; ++k) {
      const int i = lane + 64 * k;
      v[k] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (i < n4) {
        const float4 a = xr[i], s = drop4(d, sr[i], (uint32_t)row * (uint32_t)C + 4u * (uint32_t)i);
        v[k] = make_float4(a.x + s.x, a.y + s.y, a.z + s.z, a.w + s.w);
      }
      sum += (v[k].x + v[k].y) + (v[k].z + v[k].w);
    }
    const float mean = wave_sum(sum) / (float)C;
    float sq = 0.f;
#pragma unroll
    for (int k = 0; k < kMaxV; ++k)
      if (lane + 64 * k < n4) {
        const float a = v[k].x - mean, b = v[k].y - mean, c = v[k].z - mean, e = v[k].w - mean;
        sq += (a * a + b * b) + (c * c + e * e);
      }
    const float rstd = 1.f / sqrtf(wave_sum(sq) / (float)C + eps);
    float4 *yr = reinterpret_cast<float4 *>(y + (int64_t)row * C), *zr = reinterpret_cast<float4 *>(z + (int64_t)row * C);
#pragma unroll
    for (int k = 0; k < kMaxV; ++k) {
      const int i = lane + 64 * k;
      if (i < n4) {
        const float4 g = g4[i], b = b4[i];
        zr[i] = v[k];
        const float4 o = make_float4((v[k].x - mean) * rstd * g.x + b.x, (v[k].y - mean) * rstd * g.y + b.y,
                                     (v[k].z - mean) * rstd * g.z + b.z, (v[k].w - mean) * rstd * g.w + b.w);
        yr[i] = o;
        v[k] = o;
      }
    }
    if (amax_y || amax_ypos) {
      // row maxima of y (and of y + pos, the query of the attention behind this norm): what the split of the next Linear
      // product's activation operand needs (gvl_linear_f16x3_f32), left here instead of a pass of its own
      const float4 *pr = pos ? reinterpret_cast<const float4 *>(pos + (int64_t)rb * pos_sb + (int64_t)rq * pos_sq) : nullptr;
      float m = 0.f, mp = 0.f;
#pragma unroll
      for (int k = 0; k < kMaxV; ++k) {
        const int i = lane + 64 * k;
        if (i < n4) {
          m = fmaxf(fmaxf(m, fmaxf(fabsf(v[k].x), fabsf(v[k].y))), fmaxf(fabsf(v[k].z), fabsf(v[k].w)));
          if (pr) {
            const float4 q = pr[i];
            mp = fmaxf(fmaxf(mp, fmaxf(fabsf(v[k].x + q.x), fabsf(v[k].y + q.y))), fmaxf(fabsf(v[k].z + q.z), fabsf(v[k].w + q.w)));
          }
        }
      }
#pragma unroll
      for (int o = 32; o; o >>= 1) {
        m = fmaxf(m, __shfl_xor(m, o, 64));
        mp = fmaxf(mp, __shfl_xor(mp, o, 64));
      }
      if (lane == 0) {
        if (amax_y) amax_y[row] = m;
        if (amax_ypos) amax_ypos[row] = pr ? mp : m;
      }
    }
    if (lane == 0) {
      mean_out[row] = mean;
      rstd_out[row] = rstd;
    }
  }
}

// part: (gridDim.x, 2 C): [sum_r dy xh | sum_r dy] over the rows of the workgroup
// more.p[] (NULL = absent): further gradients of the same output, summed with dy as they are loaded -- a result that feeds
// several consumers gets one gradient from each, and autograd would add them with a launch (three passes) per addend
constexpr int kRdlnMoreDy = 5;
struct RdlnMoreDy { const float *p[kRdlnMoreDy]; };
__global__ void __launch_bounds__(64 * kWavesPerBlock) k_rdln_bwd(
    const float *__restrict__ dy, const RdlnMoreDy more,
    const float *__restrict__ z, const float *__restrict__ mean_in,
    const float *__restrict__ rstd_in, int R, int C, const float *__restrict__ gamma, float p, uint32_t seed,
    const int64_t *__restrict__ step, float *__restrict__ dz, float *__restrict__ dsub, float *__restrict__ part,
    float *__restrict__ amax_dz) {
  __shared__ float4 red[kWavesPerBlock - 1][2][kMaxV * 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n4 = C >> 2;
  const Drop d = make_drop(p, seed, step);
  const float4 *g4 = reinterpret_cast<const float4 *>(gamma);
  float4 gam[kMaxV], acc_g[kMaxV], acc_b[kMaxV];
#pragma unroll
  for (int k = 0; k < kMaxV; ++k) {
    const int i = lane + 64 * k;
    gam[k] = i < n4 ? g4[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    acc_g[k] = acc_b[k] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  const float invC = 1.f / (float)C;
  for (int row = blockIdx.x * kWavesPerBlock + wave; row < R; row += gridDim.x * kWavesPerBlock) {
    const float4 *dr = reinterpret_cast<const float4 *>(dy + (int64_t)row * C);
    const float4 *zr = reinterpret_cast<const float4 *>(z + (int64_t)row * C);
    const float mean = mean_in[row], rstd = rstd_in[row];
    float4 g[kMaxV], xh[kMaxV];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int k = 0; k < kMaxV; ++k) {
      const int i = lane + 64 * k;
      g[k] = xh[k] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (i < n4) {
        float4 a = dr[i];
        const float4 zz = zr[i];
#pragma unroll
        for (int j = 0; j < kRdlnMoreDy; ++j)
          if (more.p[j]) {                                                // (wavefront-uniform; the absent ones cost a scalar test)
            const float4 b = reinterpret_cast<const float4 *>(more.p[j] + (int64_t)row * C)[i];
            a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
          }
        xh[k] = make_float4((zz.x - mean) * rstd, (zz.y - mean) * rstd, (zz.z - mean) * rstd, (zz.w - mean) * rstd);
        acc_g[k].x += a.x * xh[k].x; acc_g[k].y += a.y * xh[k].y; acc_g[k].z += a.z * xh[k].z; acc_g[k].w += a.w * xh[k].w;
        acc_b[k].x += a.x; acc_b[k].y += a.y; acc_b[k].z += a.z; acc_b[k].w += a.w;
        g[k] = make_float4(a.x * gam[k].x, a.y * gam[k].y, a.z * gam[k].z, a.w * gam[k].w);
      }
      s1 += (g[k].x + g[k].y) + (g[k].z + g[k].w);
      s2 += (g[k].x * xh[k].x + g[k].y * xh[k].y) + (g[k].z * xh[k].z + g[k].w * xh[k].w);
    }
    const float m1 = wave_sum(s1) * invC, m2 = wave_sum(s2) * invC;
    float4 *or_ = reinterpret_cast<float4 *>(dz + (int64_t)row * C);
    float4 *os = dsub ? reinterpret_cast<float4 *>(dsub + (int64_t)row * C) : nullptr;
    float am = 0.f;
#pragma unroll
    for (int k = 0; k < kMaxV; ++k) {
      const int i = lane + 64 * k;
      if (i < n4) {
        const float4 o = make_float4(rstd * (g[k].x - m1 - xh[k].x * m2), rstd * (g[k].y - m1 - xh[k].y * m2),
                                     rstd * (g[k].z - m1 - xh[k].z * m2), rstd * (g[k].w - m1 - xh[k].w * m2));
        or_[i] = o;
        if (os) os[i] = drop4(d, o, (uint32_t)row * (uint32_t)C + 4u * (uint32_t)i);
        am = fmaxf(fmaxf(am, fmaxf(fabsf(o.x), fabsf(o.y))), fmaxf(fabsf(o.z), fabsf(o.w)));
      }
    }
    if (amax_dz) {
      // max |dz row| (x 1 / (1 - p): a bound of the dropped row dsub too) for the Linear products of the sublayer's backward
#pragma unroll
      for (int o = 32; o; o >>= 1) am = fmaxf(am, __shfl_xor(am, o, 64));
      if (lane == 0) amax_dz[row] = am * d.scale;
    }
  }
  // the workgroup's column sums: wavefronts 1.. hand theirs to wavefront 0 through LDS
  if (wave > 0) {
#pragma unroll
    for (int k = 0; k < kMaxV; ++k) {
      red[wave - 1][0][lane + 64 * k] = acc_g[k];
      red[wave - 1][1][lane + 64 * k] = acc_b[k];
    }
  }
  __syncthreads();
  if (wave == 0) {
    float4 *pg = reinterpret_cast<float4 *>(part + (int64_t)blockIdx.x * 2 * C), *pb = pg + n4;
#pragma unroll
    for (int k = 0; k < kMaxV; ++k) {
      const int i = lane + 64 * k;
      if (i < n4) {
        float4 a = acc_g[k], b = acc_b[k];
#pragma unroll
        for (int w = 0; w < kWavesPerBlock - 1; ++w) {
          const float4 a2 = red[w][0][i], b2 = red[w][1][i];
          a.x += a2.x; a.y += a2.y; a.z += a2.z; a.w += a2.w;
          b.x += b2.x; b.y += b2.y; b.z += b2.z; b.w += b2.w;
        }
        pg[i] = a;
        pb[i] = b;
      }
    }
  }
}

// out[c] = sum_b part[b][c], c < C2 (= 2 C): 64 columns per workgroup (16 lanes x float4) x 64 row groups that meet in LDS --
// at most 8 independent loads per thread for the 512 partial rows of a 4800-row call (16 row groups: 32 serial loads, 10 us);
// fixed summation order (no atomics: the result is reproducible run to run)
constexpr int kFinGroups = 64;
__global__ void __launch_bounds__(16 * kFinGroups) k_rdln_finish(const float *__restrict__ part, int nb, int C2,
                                                                float *__restrict__ out) {
  __shared__ float4 red[kFinGroups][16];
  const int l16 = threadIdx.x & 15, grp = threadIdx.x >> 4, c4 = blockIdx.x * 16 + l16, n4 = C2 >> 2;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  if (c4 < n4) {
#pragma unroll 8
    for (int b = grp; b < nb; b += kFinGroups) {
      const float4 v = reinterpret_cast<const float4 *>(part + (int64_t)b * C2)[c4];
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
  }
  red[grp][l16] = acc;
  __syncthreads();
  if (grp < 8) {                                                       // 8 groups each add 8 entries, then group 0 adds those
    acc = red[grp][l16];
#pragma unroll
    for (int g = 1; g < kFinGroups / 8; ++g) {
      const float4 v = red[grp + 8 * g][l16];
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
  }
  __syncthreads();
  if (grp < 8) red[grp][l16] = acc;
  __syncthreads();
  if (grp == 0 && c4 < n4) {
#pragma unroll
    for (int g = 1; g < 8; ++g) {
      const float4 v = red[g][l16];
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    reinterpret_cast<float4 *>(out)[c4] = acc;
  }
}

// ---- y = dropout(relu(x)) of the FFNs (deformable_transformer.py:189-191, 257-259: `dropout(activation(linear1(x)))`), in place
// when y == x.  The backward needs no mask: y > 0 exactly where the element was kept AND positive, so dx = y > 0 ? dy / (1 - p) : 0
// (PyTorch: threshold + fused_dropout forward, masked_scale + threshold_backward backward, a mask tensor between them).
__global__ void __launch_bounds__(256) k_relu_dropout_fwd(const float4 *__restrict__ x, int64_t n4, float p, uint32_t seed,
                                                          const int64_t *__restrict__ step, float4 *__restrict__ y) {
  const Drop d = make_drop(p, seed, step);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    float4 v = x[i];
    v = make_float4(fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f));
    y[i] = drop4(d, v, (uint32_t)(4 * i));
  }
}

__global__ void __launch_bounds__(256) k_relu_dropout_bwd(const float4 *__restrict__ dy, const float4 *__restrict__ y,
                                                          int64_t n4, float scale, float4 *__restrict__ dx) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    const float4 g = dy[i], v = y[i];
    dx[i] = make_float4(v.x > 0.f ? g.x * scale : 0.f, v.y > 0.f ? g.y * scale : 0.f, v.z > 0.f ? g.z * scale : 0.f,
                        v.w > 0.f ? g.w * scale : 0.f);
  }
}

// row forms: one wavefront per row, the row's maximum of the result left for the next Linear product's split
__global__ void __launch_bounds__(256) k_relu_dropout_rows_fwd(const float *__restrict__ x, int R, int C, float p, uint32_t seed,
                                                               const int64_t *__restrict__ step, float *__restrict__ y,
                                                               float *__restrict__ amax, int64_t *__restrict__ step_used) {
  const Drop d = make_drop(p, seed, step);
  if (step_used && blockIdx.x == 0 && threadIdx.x == 0) *step_used = step ? *step : 0;
  const int lane = threadIdx.x & 63, n4 = C >> 2;
  for (int row = blockIdx.x * 4 + (threadIdx.x >> 6); row < R; row += gridDim.x * 4) {
    const float4 *xr = reinterpret_cast<const float4 *>(x + (int64_t)row * C);
    float4 *yr = reinterpret_cast<float4 *>(y + (int64_t)row * C);
    float m = 0.f;
    for (int i = lane; i < n4; i += 64) {
      float4 v = xr[i];
      v = make_float4(fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f));
      v = drop4(d, v, (uint32_t)row * (uint32_t)C + 4u * (uint32_t)i);       // (the element index of the flat form)
      yr[i] = v;
      m = fmaxf(fmaxf(m, fmaxf(v.x, v.y)), fmaxf(v.z, v.w));
    }
#pragma unroll
    for (int o = 32; o; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if (lane == 0) amax[row] = m;
  }
}

__global__ void __launch_bounds__(256) k_relu_dropout_rows_bwd(const float *__restrict__ dy, const float *__restrict__ y, int R, int C,
                                                               float scale, float *__restrict__ dx, float *__restrict__ amax) {
  const int lane = threadIdx.x & 63, n4 = C >> 2;
  for (int row = blockIdx.x * 4 + (threadIdx.x >> 6); row < R; row += gridDim.x * 4) {
    const float4 *gr = reinterpret_cast<const float4 *>(dy + (int64_t)row * C), *yr = reinterpret_cast<const float4 *>(y + (int64_t)row * C);
    float4 *xr = reinterpret_cast<float4 *>(dx + (int64_t)row * C);
    float m = 0.f;
    for (int i = lane; i < n4; i += 64) {
      const float4 g = gr[i], v = yr[i];
      const float4 o = make_float4(v.x > 0.f ? g.x * scale : 0.f, v.y > 0.f ? g.y * scale : 0.f, v.z > 0.f ? g.z * scale : 0.f,
                                   v.w > 0.f ? g.w * scale : 0.f);
      xr[i] = o;
      m = fmaxf(fmaxf(m, fmaxf(fabsf(o.x), fabsf(o.y))), fmaxf(fabsf(o.z), fabsf(o.w)));
    }
#pragma unroll
    for (int o = 32; o; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if (lane == 0) amax[row] = m;
  }
}

__global__ void k_advance_step(int64_t *step) { *step += 1; }

int blocks_for(int R) {
  const int need = (R + kWavesPerBlock - 1) / kWavesPerBlock;
  return need < 512 ? (need < 1 ? 1 : need) : 512;                    // two workgroups of 8 wavefronts per CU
}

int check_shape(const char *what, int R, int C, float p) {
  if (R < 0 || C <= 0 || (C & 3) || C > 256 * kMaxV)
    return fail(GVL_EINVAL, "%s: needs C %% 4 == 0 and C <= %d (got R=%d C=%d)", what, 256 * kMaxV, R, C);
  if (!(p >= 0.f && p < 1.f)) return fail(GVL_EINVAL, "%s: dropout probability must be in [0, 1) (got %g)", what, (double)p);
  if ((int64_t)R * C >= (int64_t)1 << 32) return fail(GVL_EINVAL, "%s: more than 2^32 elements", what);
  return 0;
}

}  // namespace

extern "C" int gvl_rdln_backward_blocks(int R) { return blocks_for(R); }

extern "C" int gvl_advance_step(int64_t *step, void *stream) {
  if (!step) return fail(GVL_EINVAL, "gvl_advance_step: null pointer");
  hipLaunchKernelGGL(k_advance_step, dim3(1), dim3(1), 0, (hipStream_t)stream, step);
  const hipError_t err = hipGetLastError();
  return err == hipSuccess ? 0 : fail((int)err, "gvl_advance_step: %s", hipGetErrorString(err));
}

extern "C" int gvl_residual_dropout_layer_norm_forward_f32(const float *x, int64_t x_sb, int64_t x_sq, const float *sub,
                                                           int64_t sub_sb, int64_t sub_sq, int Q, int R, int C,
                                                           const float *gamma, const float *beta, float eps, float p,
                                                           uint32_t seed, const int64_t *step, float *y, float *z,
                                                           float *mean, float *rstd, int64_t *step_used, const float *pos,
                                                           int64_t pos_sb, int64_t pos_sq, float *amax_y, float *amax_ypos,
                                                           void *stream) {
  const char *what = "gvl_residual_dropout_layer_norm_forward_f32";
  if (int rc = check_shape(what, R, C, p)) return rc;
  if (R == 0) return 0;
  if (!x || !sub || !gamma || !beta || !y || !z || !mean || !rstd) return fail(GVL_EINVAL, "%s: null pointer", what);
  if (Q <= 0 || ((sub_sb | sub_sq | x_sb | x_sq) & 3) || (((uintptr_t)x | (uintptr_t)sub | (uintptr_t)gamma | (uintptr_t)beta |
                                                    (uintptr_t)y | (uintptr_t)z) & 15))
    return fail(GVL_EINVAL, "%s: Q > 0, row strides of x / sub multiples of 4, every tensor 16-byte aligned", what);
  if (pos && (((pos_sb | pos_sq) & 3) || ((uintptr_t)pos & 15)))
    return fail(GVL_EINVAL, "%s: pos must be 16-byte aligned with row strides that are multiples of 4", what);
  return gvl::launch(GVL_PROF_LAYER_NORM, R, C, "k_rdln_fwd", k_rdln_fwd, dim3(blocks_for(R)), dim3(64 * kWavesPerBlock), 0,
                     (hipStream_t)stream, x, x_sb, x_sq, sub, sub_sb, sub_sq, Q, R, C, gamma, beta, eps, p, seed, step, y, z, mean,
                     rstd, step_used, pos, pos_sb, pos_sq, amax_y, amax_ypos);
}

extern "C" int gvl_residual_dropout_layer_norm_backward_f32(const float *dy, const float *z, const float *mean,
                                                            const float *rstd, int R, int C, const float *gamma, float p,
                                                            uint32_t seed, const int64_t *step, float *dz, float *dsub,
                                                            float *part, float *dgamma_dbeta, float *amax_dz, void *stream) {
  return gvl_residual_dropout_layer_norm_backwardn_f32(&dy, 1, z, mean, rstd, R, C, gamma, p, seed, step, dz, dsub, part, dgamma_dbeta,
                                                       amax_dz, stream);
}

extern "C" int gvl_rdln_backward_max_grads(void) { return 1 + kRdlnMoreDy; }

extern "C" int gvl_residual_dropout_layer_norm_backwardn_f32(const float *const *dys, int n_dy, const float *z, const float *mean,
                                                             const float *rstd, int R, int C, const float *gamma, float p,
                                                             uint32_t seed, const int64_t *step, float *dz, float *dsub, float *part,
                                                             float *dgamma_dbeta, float *amax_dz, void *stream) {
  const char *what = "gvl_residual_dropout_layer_norm_backwardn_f32";
  if (!dys || n_dy < 1 || n_dy > 1 + kRdlnMoreDy) return fail(GVL_EINVAL, "%s: 1..%d output gradients", what, 1 + kRdlnMoreDy);
  const float *dy = dys[0];
  RdlnMoreDy more{};
  for (int j = 1; j < n_dy; ++j) {
    if (!dys[j] || ((uintptr_t)dys[j] & 15)) return fail(GVL_EINVAL, "%s: gradient %d is null / not 16-byte aligned", what, j);
    more.p[j - 1] = dys[j];
  }
  if (int rc = check_shape(what, R, C, p)) return rc;
  if (!dgamma_dbeta) return fail(GVL_EINVAL, "%s: null pointer", what);
  if (R == 0) return gvl::zero_fill(dgamma_dbeta, sizeof(float) * 2 * C, (hipStream_t)stream);
  if (!dy || !z || !mean || !rstd || !gamma || !dz || !part) return fail(GVL_EINVAL, "%s: null pointer", what);
  if (p > 0.f && !dsub) return fail(GVL_EINVAL, "%s: dsub is needed when p > 0 (p = 0: dsub == dz)", what);
  if (((uintptr_t)dy | (uintptr_t)z | (uintptr_t)gamma | (uintptr_t)dz | (uintptr_t)dsub | (uintptr_t)part) & 15)
    return fail(GVL_EINVAL, "%s: every tensor must be 16-byte aligned", what);
  const int nb = blocks_for(R);
  if (int rc = gvl::launch(GVL_PROF_LAYER_NORM, R, C, "k_rdln_bwd", k_rdln_bwd, dim3(nb), dim3(64 * kWavesPerBlock), 0,
                           (hipStream_t)stream, dy, more, z, mean, rstd, R, C, gamma, p, seed, step, dz, p > 0.f ? dsub : nullptr,
                           part, amax_dz))
    return rc;
  return gvl::launch(GVL_PROF_LAYER_NORM, nb, 2 * C, "k_rdln_finish", k_rdln_finish, dim3((2 * C / 4 + 15) / 16), dim3(16 * kFinGroups), 0,
                     (hipStream_t)stream, part, nb, 2 * C, dgamma_dbeta);
}

extern "C" int gvl_relu_dropout_forward_f32(const float *x, int64_t n, float p, uint32_t seed, const int64_t *step, float *y,
                                            void *stream) {
  const char *what = "gvl_relu_dropout_forward_f32";
  if (n < 0 || (n & 3) || n >= ((int64_t)1 << 32)) return fail(GVL_EINVAL, "%s: needs 0 <= n < 2^32, n %% 4 == 0 (got %lld)", what, (long long)n);
  if (!(p >= 0.f && p < 1.f)) return fail(GVL_EINVAL, "%s: dropout probability must be in [0, 1) (got %g)", what, (double)p);
  if (n == 0) return 0;
  if (!x || !y || (((uintptr_t)x | (uintptr_t)y) & 15)) return fail(GVL_EINVAL, "%s: null / unaligned pointer", what);
  const int64_t n4 = n >> 2;
  const int blocks = (int)((n4 + 255) / 256 < 4096 ? (n4 + 255) / 256 : 4096);
  return gvl::launch(GVL_PROF_LAYER_NORM, (int)(n >> 10), 0, "k_relu_dropout_fwd", k_relu_dropout_fwd, dim3(blocks), dim3(256), 0,
                     (hipStream_t)stream, (const float4 *)x, n4, p, seed, step, (float4 *)y);
}

extern "C" int gvl_relu_dropout_backward_f32(const float *dy, const float *y, int64_t n, float p, float *dx, void *stream) {
  const char *what = "gvl_relu_dropout_backward_f32";
  if (n < 0 || (n & 3)) return fail(GVL_EINVAL, "%s: needs n %% 4 == 0 (got %lld)", what, (long long)n);
  if (!(p >= 0.f && p < 1.f)) return fail(GVL_EINVAL, "%s: dropout probability must be in [0, 1) (got %g)", what, (double)p);
  if (n == 0) return 0;
  if (!dy || !y || !dx || (((uintptr_t)dy | (uintptr_t)y | (uintptr_t)dx) & 15)) return fail(GVL_EINVAL, "%s: null / unaligned pointer", what);
  const int64_t n4 = n >> 2;
  const int blocks = (int)((n4 + 255) / 256 < 4096 ? (n4 + 255) / 256 : 4096);
  return gvl::launch(GVL_PROF_LAYER_NORM, (int)(n >> 10), 0, "k_relu_dropout_bwd", k_relu_dropout_bwd, dim3(blocks), dim3(256), 0,
                     (hipStream_t)stream, (const float4 *)dy, (const float4 *)y, n4, p > 0.f ? 1.f / (1.f - p) : 1.f, (float4 *)dx);
}

extern "C" int gvl_relu_dropout_rows_forward_f32(const float *x, int R, int C, float p, uint32_t seed, const int64_t *step, float *y,
                                                 float *amax, int64_t *step_used, void *stream) {
  const char *what = "gvl_relu_dropout_rows_forward_f32";
  if (R < 0 || C <= 0 || (C & 3) || (int64_t)R * C >= ((int64_t)1 << 32))
    return fail(GVL_EINVAL, "%s: needs C %% 4 == 0 and fewer than 2^32 elements (got R=%d C=%d)", what, R, C);
  if (!(p >= 0.f && p < 1.f)) return fail(GVL_EINVAL, "%s: dropout probability must be in [0, 1) (got %g)", what, (double)p);
  if (R == 0) return 0;
  if (!x || !y || !amax || (((uintptr_t)x | (uintptr_t)y) & 15)) return fail(GVL_EINVAL, "%s: null / unaligned pointer", what);
  const int blocks = (R + 3) / 4 < 2048 ? (R + 3) / 4 : 2048;
  return gvl::launch(GVL_PROF_LAYER_NORM, R, C, "k_relu_dropout_rows_fwd", k_relu_dropout_rows_fwd, dim3(blocks), dim3(256), 0,
                     (hipStream_t)stream, x, R, C, p, seed, step, y, amax, step_used);
}

extern "C" int gvl_relu_dropout_rows_backward_f32(const float *dy, const float *y, int R, int C, float p, float *dx, float *amax,
                                                  void *stream) {
  const char *what = "gvl_relu_dropout_rows_backward_f32";
  if (R < 0 || C <= 0 || (C & 3)) return fail(GVL_EINVAL, "%s: needs C %% 4 == 0 (got R=%d C=%d)", what, R, C);
  if (!(p >= 0.f && p < 1.f)) return fail(GVL_EINVAL, "%s: dropout probability must be in [0, 1) (got %g)", what, (double)p);
  if (R == 0) return 0;
  if (!dy || !y || !dx || !amax || (((uintptr_t)dy | (uintptr_t)y | (uintptr_t)dx) & 15))
    return fail(GVL_EINVAL, "%s: null / unaligned pointer", what);
  const int blocks = (R + 3) / 4 < 2048 ? (R + 3) / 4 : 2048;
  return gvl::launch(GVL_PROF_LAYER_NORM, R, C, "k_relu_dropout_rows_bwd", k_relu_dropout_rows_bwd, dim3(blocks), dim3(256), 0,
                     (hipStream_t)stream, dy, y, R, C, p > 0.f ? 1.f / (1.f - p) : 1.f, dx, amax);
}
