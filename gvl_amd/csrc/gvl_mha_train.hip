// gvl_mha_train.hip -- the attention core of the decoder layer's nn.MultiheadAttention in TRAINING
// (pdvc/deformable_transformer.py:266-270: self_attn(q, k, tgt, key_padding_mask) with dropout on the attention weights), forward
// and backward, on the fp16 matrix cores at fp32 accuracy.  PyTorch runs it as bmm + softmax + dropout + bmm and ten backward
// launches over a (B, 8, Q, Q) score tensor in HBM (93 + 160 us per layer at B = 16, Q = 300); here the scores never leave registers.
//
//   forward   out[q] = sum_k dropout(softmax_k(q . k / 8 + mask))[q][k] v[k]    per (video, head);  lse[q] = log sum_k exp(score)
//   backward  dv = Pd^T dO;  dP = keep / (1 - p) (dO V^T);  dS = P (dP - delta),  delta[q] = dO[q] . out[q];  dq = dS K / 8;  dk = dS^T Q / 8
//
// Arithmetic: every product is three fp16 MFMAs (hi.hi + hi.lo + lo.hi) in ONE fp32 accumulator -- operands are split as
// t = x c (c a power of two that puts the operand's largest element in [2^11, 2^12)), hi = fp16(t), lo = fp16(t - hi): the
// residual keeps the scale of hi (gvl_train_gemm.hip).  q, k, v, dO take c from the row maxima their producers leave behind; the
// probabilities are in [0, 1 / (1 - p)]; dS takes it from the wavefront's own tile maximum.
//
// Layout trick (cdna_hip_programming.md, "an accumulator tile as the next MFMA's operand"): a 32 x 32 result X of
// v_mfma_f32_32x32x16_f16 has its COLUMN on the lane and its rows in the 16 registers, so a following product that sums over X's
// rows takes registers 8 s .. 8 s + 7 as the B fragment of k-step s with no lane movement; the k order inside a step is permuted
// (element j of lane half h = row 16 s + 8 (j >> 2) + 4 h + (j & 3)), which the A operand -- a TRANSPOSED tile read from LDS by
// ds_read_b64_tr_b16, four consecutive rows per read -- follows.  Hence:
//   k_mha_fwd     S^T = K Q^T  (column = query)  ->  online softmax per column  ->  O^T += V^T P^T     (P^T from registers)
//   k_mha_bwd_q   the same S^T, P^T;  dP^T = V dO^T;  dS^T;  dQ^T += K^T dS^T                          (dS^T from registers)
//   k_mha_bwd_kv  S = Q K^T  (column = key);  dP = dO V^T;  dV^T += dO^T Pd;  dK^T += Q^T dS           (Pd, dS from registers)
// One wavefront owns 32 columns for the whole kernel; the other side streams through LDS in tiles of 32 rows (split while
// staged, one image serves row reads and transposed reads).  No atomics, no cross-wavefront reduction: dq comes from its own pass
// (the score tiles are recomputed: 84 instead of 60 MFMAs per tile pair, but every accumulator stays in registers).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gvl_common.hpp"
#include "gvl_gemm16_common.hpp"
#include "gvl_msda.h"

namespace {

using gvl::fail;
using namespace gvl16;

typedef __fp16 trh4 __attribute__((__vector_size__(4 * sizeof(__fp16))));
typedef _Float16 h2v __attribute__((ext_vector_type(2)));

constexpr int kD = 64;                         // head dimension
constexpr int kWaves = 5;                      // wavefronts per workgroup = 160 owned columns
constexpr int kThreads = 64 * kWaves;
constexpr int kRowB = 2 * kD + 64;             // bytes per LDS row of a plane: 128 + 64 -- four consecutive rows of a transposed
                                               // read start 64 B apart (mod 256), the 32 lanes of a half cover every bank once
constexpr int kPlaneB = 32 * kRowB;            // one plane of a 32-row tile: 6144 B
constexpr int kTileB = 2 * kPlaneB;            // hi | lo

__device__ __forceinline__ uint32_t hash32(uint32_t x) {     // (gvl_train_layers.hip: the same mask rule)
  x ^= x >> 16; x *= 0x7feb352dU;
  x ^= x >> 15; x *= 0x846ca68bU;
  x ^= x >> 16;
  return x;
}

// c = 2^11 / 2^floor(log2 amax) and 1 / c
__device__ __forceinline__ void op_scale(float amax, float &mul, float &back) {
  int e = (int)((__float_as_uint(amax) >> 23) & 0xffu);
  e = min(max(e, 13), 240);
  mul = __uint_as_float((uint32_t)(254 - e + 11) << 23);
  back = __uint_as_float((uint32_t)(e - 11) << 23);
}

__device__ __forceinline__ float wg_max(const float *__restrict__ v, int n, float *scratch) {
  float m = 0.f;
  for (int i = threadIdx.x; i < n; i += kThreads) m = fmaxf(m, v[i]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  __syncthreads();
  if ((threadIdx.x & 63) == 0) scratch[threadIdx.x >> 6] = m;
  __syncthreads();
  float r = scratch[0];
#pragma unroll
  for (int w = 1; w < kWaves; ++w) r = fmaxf(r, scratch[w]);
  return r;
}

// 8 halves of one plane from 8 floats
__device__ __forceinline__ void split8(const float (&t)[8], h8 &hi, h8 &lo) {
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    hi[c] = (_Float16)t[c];
    lo[c] = (_Float16)(t[c] - (float)hi[c]);
  }
}

// ---- a 32-row tile of a (rows, ld) fp32 matrix -> LDS planes (hi | lo), 64 columns from column `col0`; rows beyond `nrows` read
// as zeros.  Threads 0 .. 255 carry two 16-byte pieces each.
__device__ __forceinline__ void stage_load(const float *__restrict__ base, int64_t ld, int row0, int nrows, float4 (&v)[2]) {
  const int t = threadIdx.x;
  if (t < 256) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int r = row0 + (t >> 4) + 16 * i;
      v[i] = r < nrows ? *reinterpret_cast<const float4 *>(base + (int64_t)r * ld + 4 * (t & 15)) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
}
__device__ __forceinline__ void stage_store(unsigned char *tile, const float4 (&v)[2], float mul) {
  const int t = threadIdx.x;
  if (t < 256) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const float a[4] = {v[i].x * mul, v[i].y * mul, v[i].z * mul, v[i].w * mul};
      _Float16 h[4], l[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        h[c] = (_Float16)a[c];
        l[c] = (_Float16)(a[c] - (float)h[c]);
      }
      unsigned char *dst = tile + ((t >> 4) + 16 * i) * kRowB + (t & 15) * 8;
      *reinterpret_cast<uint2 *>(dst) = make_uint2(__builtin_bit_cast(uint32_t, (h2v){h[0], h[1]}), __builtin_bit_cast(uint32_t, (h2v){h[2], h[3]}));
      *reinterpret_cast<uint2 *>(dst + kPlaneB) = make_uint2(__builtin_bit_cast(uint32_t, (h2v){l[0], l[1]}), __builtin_bit_cast(uint32_t, (h2v){l[2], l[3]}));
    }
  }
}

// natural fragment of k-step s (columns 16 s .. 16 s + 15 of the tile's 64): lane l = row l & 31, 8 columns from 16 s + 8 (l >> 5)
__device__ __forceinline__ h8 frag_row(const unsigned char *plane, int s, int lane) {
  return *reinterpret_cast<const h8 *>(plane + (lane & 31) * kRowB + (16 * s + 8 * (lane >> 5)) * 2);
}
// transposed fragment: A[i = column 32 t + (l & 31) of the tile][k = rows 16 s + 8 (j >> 2) + 4 h + (j & 3)], h = l >> 5
__device__ __forceinline__ h8 frag_tr(const unsigned char *plane, int t, int s, int lane) {
  const int g = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
  const unsigned char *src = plane + (16 * s + 4 * (g >> 1) + q) * kRowB + (32 * t + 16 * (g & 1) + 4 * pp) * 2;
  const trh4 u = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) trh4 *)src);
  const trh4 v = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) trh4 *)(src + 8 * kRowB));
  return __builtin_bit_cast(h8, (__fp16 __attribute__((__vector_size__(16)))){u[0], u[1], u[2], u[3], v[0], v[1], v[2], v[3]});
}

__device__ __forceinline__ f16acc mfma3(const h8 &ah, const h8 &al, const h8 &bh, const h8 &bl, f16acc c) {
  c = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, c, 0, 0, 0);
  return c;
}

// row of register r of a 32 x 32 accumulator in lane half h
__device__ __forceinline__ int acc_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

// this wavefront's 32 owned rows of an operand (row-major fp32, 64 columns at `base`) as natural fragments in registers
struct OwnFrag { h8 hi[4], lo[4]; };
__device__ __forceinline__ void own_load(OwnFrag &f, const float *__restrict__ base, int64_t ld, int row0, int nrows, float mul, int lane) {
  const int r = row0 + (lane & 31);
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    float t[8];
    const float *src = base + (int64_t)min(r, nrows - 1) * ld + 16 * s + 8 * (lane >> 5);
    const float4 a = *reinterpret_cast<const float4 *>(src), b = *reinterpret_cast<const float4 *>(src + 4);
    const float z = r < nrows ? mul : 0.f;
    t[0] = a.x * z; t[1] = a.y * z; t[2] = a.z * z; t[3] = a.w * z; t[4] = b.x * z; t[5] = b.y * z; t[6] = b.z * z; t[7] = b.w * z;
    split8(t, f.hi[s], f.lo[s]);
  }
}

struct MhaParams {
  const float *qkv;               // (B Q, ld): [q | k | v], each H 64 wide
  int64_t ld;
  const unsigned char *keep;      // (B, Q) key mask (non-zero = attend) or NULL
  const float *amax_qk, *amax_v;  // (B Q) row bounds of the q / k columns and of the v columns
  int B, Q, H;
  float p;                        // dropout probability of the attention weights
  uint32_t seed;
  const int64_t *step;            // device step counter (gvl_advance_step) or NULL
  float *out;                     // (B Q, H 64)
  float *lse;                     // (B, H, Q)
  float *amax_out;                // (B Q) zero-initialised: max |out row| over the heads (atomic max), or NULL
  // backward
  const float *dout, *amax_dout;  // (B Q, H 64), (B Q)
  const float *delta;             // (B, H, Q): dO . out per row
  float *dqkv;                    // (B Q, ld)
  float *amax_dqk, *amax_dv;      // (B Q) zero-initialised or NULL: max |row| of the [dq | dk] and of the dv columns (atomic max)
};

struct DropKey { uint32_t key, thr; float scale; };
__device__ __forceinline__ DropKey drop_of(const MhaParams &p) {
  DropKey d;
  const uint32_t st = p.step ? (uint32_t)*p.step : 0u;
  d.key = hash32(p.seed + st * 0x9E3779B9u);
  const double t = (double)p.p * 4294967296.0;
  d.thr = p.p > 0.f ? (t >= 4294967295.0 ? 0xffffffffu : (uint32_t)t) : 0u;
  d.scale = p.p > 0.f ? 1.f / (1.f - p.p) : 1.f;
  return d;
}
// keep / (1 - p) of attention weight (video b, head h, query q, key k)
__device__ __forceinline__ float drop_factor(const DropKey &d, uint32_t bh, int Q, int q, int k) {
  const uint32_t idx = (bh * (uint32_t)Q + (uint32_t)q) * (uint32_t)Q + (uint32_t)k;
  return hash32(idx ^ d.key) >= d.thr ? d.scale : 0.f;
}

// 32-bit mask of the valid keys of tile kt (wavefront-uniform)
__device__ __forceinline__ uint32_t key_mask(const MhaParams &p, int b, int kt, int lane) {
  const int key = 32 * kt + (lane & 31);
  const bool ok = key < p.Q && (!p.keep || p.keep[(int64_t)b * p.Q + key]);
  return (uint32_t)__ballot(ok && lane < 32);
}

// =====================================================================================================================
// forward: workgroup = (video, head, block of 160 queries); wavefront w owns queries q0 + 32 w ..; K and V stream through LDS
__global__ void __launch_bounds__(kThreads) k_mha_fwd(const MhaParams p) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[2][2][kTileB];        // [buffer][K | V]
  __shared__ float red[kWaves];
  const int nqb = (p.Q + 32 * kWaves - 1) / (32 * kWaves);
  const int qb = blockIdx.x % nqb, h = (blockIdx.x / nqb) % p.H, b = blockIdx.x / (nqb * p.H);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, half = lane >> 5;
  const int q0 = qb * 32 * kWaves + 32 * wave, q = q0 + (lane & 31);
  const float *base = p.qkv + (int64_t)b * p.Q * p.ld + h * kD;
  const int C = p.H * kD, KT = (p.Q + 31) >> 5;
  float mul_qk, back_qk, mul_v, back_v;
  op_scale(wg_max(p.amax_qk + (int64_t)b * p.Q, p.Q, red), mul_qk, back_qk);
  op_scale(wg_max(p.amax_v + (int64_t)b * p.Q, p.Q, red), mul_v, back_v);
  const DropKey dk = drop_of(p);
  const uint32_t bh = (uint32_t)(b * p.H + h);

  OwnFrag fq;                                                          // Q^T as the B operand of S^T = K Q^T
  own_load(fq, base, p.ld, q0, p.Q, mul_qk, lane);
  // exp2 domain: s = (K Q^T) back_qk^2 / 8 * log2(e)
  const float sc = back_qk * back_qk * 0.125f * 1.44269504088896341f;
  f16acc o[2];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[t][r] = 0.f;
  float m = -INFINITY, l = 0.f;

  float4 sk[2], sv[2];
  stage_load(base + C, p.ld, 0, p.Q, sk);
  stage_load(base + 2 * C, p.ld, 0, p.Q, sv);
  for (int kt = 0; kt < KT; ++kt) {
    unsigned char *tk = lds[kt & 1][0], *tv = lds[kt & 1][1];
    stage_store(tk, sk, mul_qk);
    stage_store(tv, sv, mul_v);
    if (kt + 1 < KT) {
      stage_load(base + C, p.ld, 32 * (kt + 1), p.Q, sk);
      stage_load(base + 2 * C, p.ld, 32 * (kt + 1), p.Q, sv);
    }
    __syncthreads();
    f16acc s;
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
    for (int st = 0; st < 4; ++st) s = mfma3(frag_row(tk, st, lane), frag_row(tk + kPlaneB, st, lane), fq.hi[st], fq.lo[st], s);
    const uint32_t km = key_mask(p, b, kt, lane);
    float tmax = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      s[r] = (km >> acc_row(r, half)) & 1u ? s[r] * sc : -INFINITY;
      tmax = fmaxf(tmax, s[r]);
    }
    tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
    // a query with no valid key SO FAR (a leading tile of 32 padded keys): m_new = -inf, and exp2(-inf - -inf) would
    // be NaN for good -- nothing has been accumulated yet, so the rescale is 1 and this tile's weights are 0.  A row whose EVERY
    // key is masked still ends as 0 / 0 = NaN, as torch's softmax of a fully masked row does.
    const float m_new = fmaxf(m, tmax);
    const bool none = m_new == -INFINITY;
    const float corr = none ? 1.f : exp2f(m - m_new);
    float psum = 0.f;
    float pd[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float pr = none ? 0.f : exp2f(s[r] - m_new);
      psum += pr;
      pd[r] = p.p > 0.f ? pr * drop_factor(dk, bh, p.Q, min(q, p.Q - 1), 32 * kt + acc_row(r, half)) : pr;
    }
    l = l * corr + psum;
    m = m_new;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[t][r] *= corr;
    // O^T += V^T Pd^T: Pd^T (x 2^11) from the score registers
#pragma unroll
    for (int st = 0; st < 2; ++st) {
      float t8[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) t8[j] = pd[8 * st + j] * 2048.f;
      h8 ph, pl;
      split8(t8, ph, pl);
#pragma unroll
      for (int t = 0; t < 2; ++t) o[t] = mfma3(frag_tr(tv, t, st, lane), frag_tr(tv + kPlaneB, t, st, lane), ph, pl, o[t]);
    }
    // (no barrier here: the next tile goes to the OTHER buffer, and the barrier behind ITS stores is what a wavefront passes
    //  before anyone overwrites this one)
  }
  l += __shfl_xor(l, 32, 64);
  const float inv = back_v * (1.f / 2048.f) / l;
  if (q < p.Q) {
    float *op = p.out + ((int64_t)b * p.Q + q) * C + h * kD;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int g = 0; g < 4; ++g)          // registers 4 g .. 4 g + 3 = channels 32 t + 8 g + 4 half ..
        *reinterpret_cast<float4 *>(op + 32 * t + 8 * g + 4 * half) =
            make_float4(o[t][4 * g] * inv, o[t][4 * g + 1] * inv, o[t][4 * g + 2] * inv, o[t][4 * g + 3] * inv);
    if (half == 0 && p.lse) p.lse[((int64_t)b * p.H + h) * p.Q + q] = (m + log2f(l)) * 0.69314718055994531f;
  }
  if (p.amax_out) {
    // max |out| of this query's 64 channels (two lanes hold 32 each): the row scale out_proj's split needs (inference path)
    float mx = 0.f;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) mx = fmaxf(mx, fabsf(o[t][r] * inv));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    if (half == 0 && q < p.Q) atomicMax(reinterpret_cast<unsigned *>(p.amax_out) + (int64_t)b * p.Q + q, __float_as_uint(mx));
  }
}

// =====================================================================================================================
// backward, dq: the forward's structure (wavefront = 32 queries, keys stream) with dP^T = V dO^T and dQ^T += K^T dS^T
__global__ void __launch_bounds__(kThreads) k_mha_bwd_q(const MhaParams p) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[2][2][kTileB];
  __shared__ float red[kWaves];
  const int nqb = (p.Q + 32 * kWaves - 1) / (32 * kWaves);
  const int qb = blockIdx.x % nqb, h = (blockIdx.x / nqb) % p.H, b = blockIdx.x / (nqb * p.H);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, half = lane >> 5;
  const int q0 = qb * 32 * kWaves + 32 * wave, q = q0 + (lane & 31), qc = min(q, p.Q - 1);
  const float *base = p.qkv + (int64_t)b * p.Q * p.ld + h * kD;
  const int C = p.H * kD, KT = (p.Q + 31) >> 5;
  float mul_qk, back_qk, mul_v, back_v, mul_g, back_g;
  op_scale(wg_max(p.amax_qk + (int64_t)b * p.Q, p.Q, red), mul_qk, back_qk);
  op_scale(wg_max(p.amax_v + (int64_t)b * p.Q, p.Q, red), mul_v, back_v);
  op_scale(wg_max(p.amax_dout + (int64_t)b * p.Q, p.Q, red), mul_g, back_g);
  const DropKey dk = drop_of(p);
  const uint32_t bh = (uint32_t)(b * p.H + h);
  OwnFrag fq, fg;
  own_load(fq, base, p.ld, q0, p.Q, mul_qk, lane);
  own_load(fg, p.dout + (int64_t)b * p.Q * C + h * kD, C, q0, p.Q, mul_g, lane);
  const float sc = back_qk * back_qk * 0.125f * 1.44269504088896341f, sc_dp = back_v * back_g;
  const float lse2 = p.lse[((int64_t)b * p.H + h) * p.Q + qc] * 1.44269504088896341f;
  const float delta = p.delta[((int64_t)b * p.H + h) * p.Q + qc];
  f16acc dq[2];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) dq[t][r] = 0.f;
  float4 sk[2], sv[2];
  stage_load(base + C, p.ld, 0, p.Q, sk);
  stage_load(base + 2 * C, p.ld, 0, p.Q, sv);
  for (int kt = 0; kt < KT; ++kt) {
    unsigned char *tk = lds[kt & 1][0], *tv = lds[kt & 1][1];
    stage_store(tk, sk, mul_qk);
    stage_store(tv, sv, mul_v);
    if (kt + 1 < KT) {
      stage_load(base + C, p.ld, 32 * (kt + 1), p.Q, sk);
      stage_load(base + 2 * C, p.ld, 32 * (kt + 1), p.Q, sv);
    }
    __syncthreads();
    f16acc s, dp;
#pragma unroll
    for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
    for (int st = 0; st < 4; ++st) {
      s = mfma3(frag_row(tk, st, lane), frag_row(tk + kPlaneB, st, lane), fq.hi[st], fq.lo[st], s);
      dp = mfma3(frag_row(tv, st, lane), frag_row(tv + kPlaneB, st, lane), fg.hi[st], fg.lo[st], dp);
    }
    const uint32_t km = key_mask(p, b, kt, lane);
    float ds[16], amax = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const bool ok = (km >> acc_row(r, half)) & 1u;
      const float pr = ok ? exp2f(s[r] * sc - lse2) : 0.f;
      const float f = p.p > 0.f ? drop_factor(dk, bh, p.Q, qc, 32 * kt + acc_row(r, half)) : 1.f;
      ds[r] = pr * (dp[r] * sc_dp * f - delta);
      amax = fmaxf(amax, fabsf(ds[r]));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o, 64));
    float mul_s, back_s;
    op_scale(amax, mul_s, back_s);
    // dQ^T += K^T dS^T  (the tile's scale differs from tile to tile: fold it into the addend, accumulate unscaled)
    f16acc part[2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) part[t][r] = 0.f;
#pragma unroll
    for (int st = 0; st < 2; ++st) {
      float t8[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) t8[j] = ds[8 * st + j] * mul_s;
      h8 dh, dl;
      split8(t8, dh, dl);
#pragma unroll
      for (int t = 0; t < 2; ++t) part[t] = mfma3(frag_tr(tk, t, st, lane), frag_tr(tk + kPlaneB, t, st, lane), dh, dl, part[t]);
    }
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) dq[t][r] = fmaf(part[t][r], back_s, dq[t][r]);
  }
  if (q < p.Q) {
    const float f = back_qk * 0.125f;
    float *op = p.dqkv + ((int64_t)b * p.Q + q) * p.ld + h * kD;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int g = 0; g < 4; ++g)
        *reinterpret_cast<float4 *>(op + 32 * t + 8 * g + 4 * half) =
            make_float4(dq[t][4 * g] * f, dq[t][4 * g + 1] * f, dq[t][4 * g + 2] * f, dq[t][4 * g + 3] * f);
  }
  if (p.amax_dqk) {
    // the row scale the in-projection's backward products split dqkv by (as amax_out in the forward: two lanes hold a row's 64)
    float mx = 0.f;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) mx = fmaxf(mx, fabsf(dq[t][r]));
    mx *= fabsf(back_qk * 0.125f);
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    if (half == 0 && q < p.Q) atomicMax(reinterpret_cast<unsigned *>(p.amax_dqk) + (int64_t)b * p.Q + q, __float_as_uint(mx));
  }
}

// =====================================================================================================================
// backward, dk and dv: workgroup = (video, head, block of 160 keys); wavefront w owns keys k0 + 32 w ..; Q and dO stream
__global__ void __launch_bounds__(kThreads) k_mha_bwd_kv(const MhaParams p) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[2][2][kTileB];        // [buffer][Q | dO]
  __shared__ float rowv[2][2][32];                                                  // [buffer][lse log2e | delta]
  __shared__ float red[kWaves];
  const int nkb = (p.Q + 32 * kWaves - 1) / (32 * kWaves);
  const int kb = blockIdx.x % nkb, h = (blockIdx.x / nkb) % p.H, b = blockIdx.x / (nkb * p.H);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, half = lane >> 5;
  const int k0 = kb * 32 * kWaves + 32 * wave, key = k0 + (lane & 31);
  const float *base = p.qkv + (int64_t)b * p.Q * p.ld + h * kD;
  const int C = p.H * kD, QT = (p.Q + 31) >> 5;
  float mul_qk, back_qk, mul_v, back_v, mul_g, back_g;
  op_scale(wg_max(p.amax_qk + (int64_t)b * p.Q, p.Q, red), mul_qk, back_qk);
  op_scale(wg_max(p.amax_v + (int64_t)b * p.Q, p.Q, red), mul_v, back_v);
  op_scale(wg_max(p.amax_dout + (int64_t)b * p.Q, p.Q, red), mul_g, back_g);
  const DropKey dk = drop_of(p);
  const uint32_t bh = (uint32_t)(b * p.H + h);
  OwnFrag fk, fv;                                                      // K^T, V^T as B operands (column = key)
  own_load(fk, base + C, p.ld, k0, p.Q, mul_qk, lane);
  own_load(fv, base + 2 * C, p.ld, k0, p.Q, mul_v, lane);
  const bool key_ok = key < p.Q && (!p.keep || p.keep[(int64_t)b * p.Q + min(key, p.Q - 1)]);
  const float sc = back_qk * back_qk * 0.125f * 1.44269504088896341f, sc_dp = back_v * back_g;
  const float *gbase = p.dout + (int64_t)b * p.Q * C + h * kD;
  const float *lse_b = p.lse + ((int64_t)b * p.H + h) * p.Q, *del_b = p.delta + ((int64_t)b * p.H + h) * p.Q;
  f16acc dkT[2], dvT[2];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) { dkT[t][r] = 0.f; dvT[t][r] = 0.f; }
  float4 sq[2], sg[2];
  stage_load(base, p.ld, 0, p.Q, sq);
  stage_load(gbase, C, 0, p.Q, sg);
  for (int qt = 0; qt < QT; ++qt) {
    unsigned char *tq = lds[qt & 1][0], *tg = lds[qt & 1][1];
    stage_store(tq, sq, mul_qk);
    stage_store(tg, sg, mul_g);
    if (threadIdx.x < 32) {
      const int qq = min(32 * qt + (int)threadIdx.x, p.Q - 1);
      rowv[qt & 1][0][threadIdx.x] = lse_b[qq] * 1.44269504088896341f;
      rowv[qt & 1][1][threadIdx.x] = del_b[qq];
    }
    if (qt + 1 < QT) {
      stage_load(base, p.ld, 32 * (qt + 1), p.Q, sq);
      stage_load(gbase, C, 32 * (qt + 1), p.Q, sg);
    }
    __syncthreads();
    f16acc s, dp;
#pragma unroll
    for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
    for (int st = 0; st < 4; ++st) {
      s = mfma3(frag_row(tq, st, lane), frag_row(tq + kPlaneB, st, lane), fk.hi[st], fk.lo[st], s);
      dp = mfma3(frag_row(tg, st, lane), frag_row(tg + kPlaneB, st, lane), fv.hi[st], fv.lo[st], dp);
    }
    float pd[16], ds[16], amax = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = acc_row(r, half), qq = 32 * qt + row;
      const bool ok = key_ok && qq < p.Q;
      const float pr = ok ? exp2f(s[r] * sc - rowv[qt & 1][0][row]) : 0.f;
      const float f = p.p > 0.f ? drop_factor(dk, bh, p.Q, min(qq, p.Q - 1), min(key, p.Q - 1)) : 1.f;
      pd[r] = pr * f;
      ds[r] = pr * (dp[r] * sc_dp * f - rowv[qt & 1][1][row]);
      amax = fmaxf(amax, fabsf(ds[r]));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o, 64));
    float mul_s, back_s;
    op_scale(amax, mul_s, back_s);
    f16acc part[2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) part[t][r] = 0.f;
#pragma unroll
    for (int st = 0; st < 2; ++st) {
      float t8[8];
      h8 ph, pl, dh, dl;
#pragma unroll
      for (int j = 0; j < 8; ++j) t8[j] = pd[8 * st + j] * 2048.f;
      split8(t8, ph, pl);
#pragma unroll
      for (int j = 0; j < 8; ++j) t8[j] = ds[8 * st + j] * mul_s;
      split8(t8, dh, dl);
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        dvT[t] = mfma3(frag_tr(tg, t, st, lane), frag_tr(tg + kPlaneB, t, st, lane), ph, pl, dvT[t]);      // dV^T += dO^T Pd
        part[t] = mfma3(frag_tr(tq, t, st, lane), frag_tr(tq + kPlaneB, t, st, lane), dh, dl, part[t]);    // dK^T += Q^T dS
      }
    }
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) dkT[t][r] = fmaf(part[t][r], back_s, dkT[t][r]);
  }
  if (key < p.Q) {
    const float fk_ = back_qk * 0.125f, fv_ = back_g * (1.f / 2048.f);
    float *ok_ = p.dqkv + ((int64_t)b * p.Q + key) * p.ld + C + h * kD, *ov = ok_ + C;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        *reinterpret_cast<float4 *>(ok_ + 32 * t + 8 * g + 4 * half) =
            make_float4(dkT[t][4 * g] * fk_, dkT[t][4 * g + 1] * fk_, dkT[t][4 * g + 2] * fk_, dkT[t][4 * g + 3] * fk_);
        *reinterpret_cast<float4 *>(ov + 32 * t + 8 * g + 4 * half) =
            make_float4(dvT[t][4 * g] * fv_, dvT[t][4 * g + 1] * fv_, dvT[t][4 * g + 2] * fv_, dvT[t][4 * g + 3] * fv_);
      }
  }
  if (p.amax_dqk) {
    float mk = 0.f, mv = 0.f;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) { mk = fmaxf(mk, fabsf(dkT[t][r])); mv = fmaxf(mv, fabsf(dvT[t][r])); }
    mk *= fabsf(back_qk * 0.125f);
    mv *= fabsf(back_g * (1.f / 2048.f));
    mk = fmaxf(mk, __shfl_xor(mk, 32, 64));
    mv = fmaxf(mv, __shfl_xor(mv, 32, 64));
    if (half == 0 && key < p.Q) {
      atomicMax(reinterpret_cast<unsigned *>(p.amax_dqk) + (int64_t)b * p.Q + key, __float_as_uint(mk));
      atomicMax(reinterpret_cast<unsigned *>(p.amax_dv) + (int64_t)b * p.Q + key, __float_as_uint(mv));
    }
  }
}

// delta[b][h][q] = dO[q] . out[q] over the head's 64 channels, and (optionally) max |dO row|: one wavefront per row
__global__ void __launch_bounds__(256) k_mha_delta(const float *__restrict__ dout, const float *__restrict__ out, int B, int Q, int H,
                                                   float *__restrict__ delta, float *__restrict__ amax_dout) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63, C = H * kD;
  if (row >= B * Q) return;
  const int b = row / Q, q = row % Q;
  float am = 0.f;
  for (int i = lane; i < C / 4; i += 64) {                               // 16 lanes = one head
    const float4 g = reinterpret_cast<const float4 *>(dout + (int64_t)row * C)[i], o = reinterpret_cast<const float4 *>(out + (int64_t)row * C)[i];
    float d = (g.x * o.x + g.y * o.y) + (g.z * o.z + g.w * o.w);
    am = fmaxf(fmaxf(am, fmaxf(fabsf(g.x), fabsf(g.y))), fmaxf(fabsf(g.z), fabsf(g.w)));
#pragma unroll
    for (int s = 8; s > 0; s >>= 1) d += __shfl_xor(d, s, 64);
    if ((lane & 15) == 0) delta[((int64_t)b * H + (i >> 4)) * Q + q] = d;
  }
  if (amax_dout) {
#pragma unroll
    for (int s = 32; s > 0; s >>= 1) am = fmaxf(am, __shfl_xor(am, s, 64));
    if (lane == 0) amax_dout[row] = am;
  }
}

int check(const char *what, const float *qkv, int64_t ld, int B, int Q, int H, float p) {
  if (B <= 0 || Q <= 0 || H <= 0 || ld < 3 * H * kD || (ld & 3) || ((uintptr_t)qkv & 15))
    return fail(GVL_EINVAL, "%s: needs B, Q, H > 0, ld >= 3 H 64, ld %% 4 == 0, 16-byte aligned rows", what);
  if (!(p >= 0.f && p < 1.f)) return fail(GVL_EINVAL, "%s: dropout probability must be in [0, 1) (got %g)", what, (double)p);
  if ((int64_t)B * H * Q * Q >= ((int64_t)1 << 32)) return fail(GVL_EINVAL, "%s: B H Q^2 must stay below 2^32", what);
  return 0;
}

}  // namespace

extern "C" int gvl_mha_train_forward_f32(const float *qkv, int64_t ld, const unsigned char *key_keep, const float *amax_qk,
                                         const float *amax_v, int B, int Q, int H, float p, uint32_t seed, const int64_t *step,
                                         float *out, float *lse, float *amax_out, void *stream) {
  const char *what = "gvl_mha_train_forward_f32";
  if (int rc = check(what, qkv, ld, B, Q, H, p)) return rc;
  if (!qkv || !amax_qk || !amax_v || !out || ((uintptr_t)out & 15)) return fail(GVL_EINVAL, "%s: null / unaligned pointer", what);
  MhaParams a = {};
  a.qkv = qkv; a.ld = ld; a.keep = key_keep; a.amax_qk = amax_qk; a.amax_v = amax_v; a.B = B; a.Q = Q; a.H = H; a.p = p; a.seed = seed;
  a.step = step; a.out = out; a.lse = lse; a.amax_out = amax_out;
  const int nqb = (Q + 32 * kWaves - 1) / (32 * kWaves);
  return gvl::launch(GVL_PROF_MHA_TRAIN, B, Q, "k_mha_fwd", k_mha_fwd, dim3(B * H * nqb), dim3(kThreads), 0, (hipStream_t)stream, a);
}

extern "C" int gvl_mha_train_backward_f32(const float *qkv, int64_t ld, const unsigned char *key_keep, const float *amax_qk,
                                          const float *amax_v, int B, int Q, int H, float p, uint32_t seed, const int64_t *step,
                                          const float *out, const float *lse, const float *dout, float *delta_ws,
                                          float *amax_dout_ws, float *dqkv, void *stream) {
  return gvl_mha_train_backward_amax_f32(qkv, ld, key_keep, amax_qk, amax_v, B, Q, H, p, seed, step, out, lse, dout, delta_ws,
                                         amax_dout_ws, dqkv, nullptr, nullptr, stream);
}

extern "C" int gvl_mha_train_backward_amax_f32(const float *qkv, int64_t ld, const unsigned char *key_keep, const float *amax_qk,
                                               const float *amax_v, int B, int Q, int H, float p, uint32_t seed,
                                               const int64_t *step, const float *out, const float *lse, const float *dout,
                                               float *delta_ws, float *amax_dout_ws, float *dqkv, float *amax_dqk, float *amax_dv,
                                               void *stream) {
  const char *what = "gvl_mha_train_backward_amax_f32";
  if ((amax_dqk == nullptr) != (amax_dv == nullptr)) return fail(GVL_EINVAL, "%s: amax_dqk and amax_dv come together", what);
  if (int rc = check(what, qkv, ld, B, Q, H, p)) return rc;
  if (!qkv || !amax_qk || !amax_v || !out || !lse || !dout || !delta_ws || !amax_dout_ws || !dqkv ||
      (((uintptr_t)out | (uintptr_t)dout | (uintptr_t)dqkv) & 15))
    return fail(GVL_EINVAL, "%s: null / unaligned pointer", what);
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(k_mha_delta, dim3((B * Q + 3) / 4), dim3(256), 0, st, dout, out, B, Q, H, delta_ws, amax_dout_ws);
  MhaParams a = {};
  a.qkv = qkv; a.ld = ld; a.keep = key_keep; a.amax_qk = amax_qk; a.amax_v = amax_v; a.B = B; a.Q = Q; a.H = H; a.p = p; a.seed = seed;
  a.step = step; a.lse = const_cast<float *>(lse); a.dout = dout; a.amax_dout = amax_dout_ws; a.delta = delta_ws; a.dqkv = dqkv;
  a.amax_dqk = amax_dqk; a.amax_dv = amax_dv;
  const int nb = (Q + 32 * kWaves - 1) / (32 * kWaves);
  if (int rc = gvl::launch(GVL_PROF_MHA_TRAIN, B, Q, "k_mha_bwd_kv", k_mha_bwd_kv, dim3(B * H * nb), dim3(kThreads), 0, st, a)) return rc;
  return gvl::launch(GVL_PROF_MHA_TRAIN, B, Q, "k_mha_bwd_q", k_mha_bwd_q, dim3(B * H * nb), dim3(kThreads), 0, st, a);
}
