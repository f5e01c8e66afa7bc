// gvl_cap_train.hip -- TRAINING-time kernels of the LSTM-DSA captioner's token step (teacher forcing) on MI355X.
//
// Reference: ShowAttendTellCore.forward (pdvc/CaptioningHead/LSTM_DSA.py:241-271) with MSDeformAttnCap.forward
// (pdvc/ops/modules/ms_deform_attn_for_caption.py:82-127) inside it, and their autograd.  In training the captioner
// runs on the matched queries only (pdvc.py:743-760: ~3 rows per video), so every PyTorch kernel of the reference
// formulation is launch-bound: ~64 launches forward and ~150 backward per token.  Here one token step is
//     GEMM over h  ->  k_cap_train_fwd  ->  GEMM over att  ->  k_lstm_train_fwd
// and its backward
//     k_lstm_train_bwd  ->  GEMM  ->  k_cap_train_bwd  ->  GEMM,
// with all weight gradients deferred to one GEMM per weight after the time loop (gvl_amd/CaptioningHead/LSTM_DSA.py).
//
//   k_cap_train_fwd   one workgroup (4 wavefronts x 4 samples) per (video, matched query) row, lane = 8 of the 512 channels:
//                       x_k    = ref + off_k / T_l   |   ref_c + off_k / P * ref_len * 0.5        (16 samples)
//                       clip_k = border-padded linear sample of value_proj(memory) at x_k
//                       e_k    = alpha_w . tanh(ctx2att(clip_k) + h2att(h)) + alpha_b ;  alpha = softmax_k(e)
//                       att    = sum_k alpha_k clip_k
//                     (ctx2att pushed through the interpolation: the slab is [value | ctx2att(value)], see gvl_cap.hip)
//   k_cap_train_bwd   the exact gradient of the above w.r.t. the slab (scatter with hardware float atomics: rows of
//                     one video collide), h2att(h), the offsets, the reference points, alpha_w and alpha_b.  Samples
//                     are re-gathered instead of stored (32 KB per row per step would have to round-trip HBM).
//   k_lstm_train_fwd / _bwd   pointwise LSTM cell (nn.LSTM single layer, bias-free; LSTM_DSA.py:216-217,269) and
//                     its backward; the forward keeps the activated gates.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gvl_common.hpp"
#include "gvl_msda.h"

namespace {

using gvl::fail;

constexpr int kC = 512;
constexpr int kLP = 16;

__device__ inline float fast_tanh(float x) {
  const float e = __expf(2.f * x);
  return fmaf(-2.f, __builtin_amdgcn_rcpf(1.f + e), 1.f);      // v_rcp_f32 (1 ulp), not a ~10-instruction IEEE division
}
__device__ inline float sigmoidf_(float x) { return __builtin_amdgcn_rcpf(1.f + __expf(-x)); }

// reduce-scatter of v[16] over the 64 lanes: afterwards every lane holds the full sum of v[k], k = lane >> 2
__device__ inline float butterfly16(float (&v)[16], int lane) {
  float a8[8];
  const bool up5 = lane & 32;
#pragma unroll
  for (int i = 0; i < 8; ++i) a8[i] = (up5 ? v[8 + i] : v[i]) + __shfl_xor(up5 ? v[i] : v[8 + i], 32, 64);
  float a4[4];
  const bool up4 = lane & 16;
#pragma unroll
  for (int i = 0; i < 4; ++i) a4[i] = (up4 ? a8[4 + i] : a8[i]) + __shfl_xor(up4 ? a8[i] : a8[4 + i], 16, 64);
  float a2[2];
  const bool up3 = lane & 8;
#pragma unroll
  for (int i = 0; i < 2; ++i) a2[i] = (up3 ? a4[2 + i] : a4[i]) + __shfl_xor(up3 ? a4[i] : a4[2 + i], 8, 64);
  const bool up2 = lane & 4;
  float r = (up2 ? a2[1] : a2[0]) + __shfl_xor(up2 ? a2[0] : a2[1], 4, 64);
  r += __shfl_xor(r, 2, 64);
  r += __shfl_xor(r, 1, 64);
  return r;
}
__device__ inline float groups_max(float v) {
#pragma unroll
  for (int o = 4; o < 64; o <<= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ inline float groups_sum(float v) {
#pragma unroll
  for (int o = 4; o < 64; o <<= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ inline float bcast(float v, int src_lane) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), src_lane));
}

// border-mode sample geometry (grid_sampler border, align_corners=False) on the clamped row pair (r, r+1);
// dmul = d pixel / d loc (T inside the level, 0 where the coordinate is clipped -- clip_coordinates_set_grad)
struct Geo {
  int r;
  float c_lo, c_hi, dmul;
};
__device__ inline Geo border_geo(float loc, int T) {
  Geo g;
  float x = ((2.f * loc - 1.f + 1.f) * (float)T - 1.f) * 0.5f;
  const float mx = (float)(T - 1);
  g.dmul = (float)T;
  if (!(x > 0.f)) { x = 0.f; g.dmul = 0.f; }
  else if (x >= mx) { x = mx; g.dmul = 0.f; }
  const float xf = floorf(x);
  const int x0 = (int)xf;
  const float a = x - xf;
  const int rmax = T >= 2 ? T - 2 : 0;
  g.r = x0 > rmax ? rmax : x0;
  const float t0 = 1.f - a;
  const float t1 = (x0 + 1 <= T - 1) ? a : 0.f;
  g.c_lo = (x0 == g.r ? t0 : 0.f) + (x0 + 1 == g.r ? t1 : 0.f);
  g.c_hi = (x0 == g.r + 1 ? t0 : 0.f) + (x0 + 1 == g.r + 1 ? t1 : 0.f);
  return g;
}

struct RowSetup {
  bool with_len;            // reference point carries a length (RD = 2 and ref_len >= 0)
  int roff;                 // slab row of the lane group's sample (level start + r)
  float c_lo, c_hi, dmul;   // interpolation coefficients, d pixel / d loc
  float doff;               // d loc / d offset
  float off;                // total offset of the sample
  int level;
};

__device__ inline RowSetup setup_row(const int64_t *shapes, const int64_t *lsi, const float *ref, const float *off_hs,
                                     const float *off_h, int64_t row, int k_own, int L, int P, int RD) {
  RowSetup s = {false, 0, 0.f, 0.f, 0.f, 0.f, 0.f, 0};
  const int LP = L * P;
  if (k_own < LP) {
    const int l = k_own / P;
    const int T = (int)shapes[2 * l + 1];
    const float *rp = ref + (row * L + l) * RD;
    s.level = l;
    s.off = off_hs[row * LP + k_own] + off_h[k_own];
    float locx;
    // RD = 2 rows with a NEGATIVE length are centre-only reference points stored in the two-component layout (lets
    // one launch serve decoder layers with both reference forms)
    s.with_len = RD == 2 && rp[1] >= 0.f;
    if (!s.with_len) { locx = rp[0] + s.off / (float)T; s.doff = 1.f / (float)T; }           // for_caption.py:108-109
    else { locx = rp[0] + s.off / (float)P * rp[1] * 0.5f; s.doff = rp[1] * 0.5f / (float)P; }  // :110-112
    const Geo g = border_geo(locx, T);
    s.roff = (int)lsi[l] + g.r;
    s.c_lo = g.c_lo; s.c_hi = g.c_hi; s.dmul = g.dmul;
  }
  return s;
}

// ------------------------------------------------------------------------------------------------------
// One WORKGROUP of 4 wavefronts per (video, matched query) row: wavefront w owns samples 4w..4w+3 (a row's work is a
// chain of 16 dependent gather -> reduce steps, and only ~100 rows exist, so the parallelism has to come from inside
// the row); lane = 8 of the 512 channels as in gvl_cap.hip.  Per-sample scalars meet in LDS.
// ------------------------------------------------------------------------------------------------------
constexpr int kRowWaves = 4;
constexpr int kPerWave = kLP / kRowWaves;

__device__ inline float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

__global__ void __launch_bounds__(kRowWaves * 64) k_cap_train_fwd(
    const float *__restrict__ slab, const int64_t *__restrict__ shapes, const int64_t *__restrict__ lsi,
    const float *__restrict__ ref, const float *__restrict__ off_hs, const float *__restrict__ off_h, int off_h_ld,
    const float *__restrict__ att_h, int att_h_ld, const float *__restrict__ alpha_w,
    const float *__restrict__ alpha_b, int S, int L, int Q, int P, int RD, const int64_t *__restrict__ row_video,
    float *__restrict__ att_res, float *__restrict__ alpha_out) {
  __shared__ float sh_e[kLP];
  __shared__ float sh_acc[kRowWaves][64][8];
  const int64_t row = blockIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, k_own = lane >> 2, LP = L * P;
  // rows grouped per video (Q rows each), or -- compact form -- the video of every row given explicitly; a negative
  // entry marks an unused row of a fixed-capacity (padded) row set: zeros out, no work
  const int b = row_video ? (int)row_video[row] : (int)(row / Q);
  if (b < 0) {
    att_res[row * kC + threadIdx.x] = 0.f;
    att_res[row * kC + 256 + threadIdx.x] = 0.f;
    if (threadIdx.x < kLP) alpha_out[row * kLP + threadIdx.x] = 0.f;
    return;
  }
  const RowSetup rs = setup_row(shapes, lsi, ref, off_hs, off_h + row * (int64_t)off_h_ld, row, k_own, L, P, RD);
  const float4 *ah4 = reinterpret_cast<const float4 *>(att_h + row * (int64_t)att_h_ld);
  const float4 ta = ah4[lane], tb = ah4[64 + lane];
  const float4 qa = reinterpret_cast<const float4 *>(alpha_w)[lane], qb = reinterpret_cast<const float4 *>(alpha_w)[64 + lane];
  const float4 *slab4 = reinterpret_cast<const float4 *>(slab) + (int64_t)b * S * (2 * kC / 4);
#pragma unroll
  for (int i = 0; i < kPerWave; ++i) {
    const int k = wave * kPerWave + i;
    float s = 0.f;
    if (k < LP) {
      const int rr = __builtin_amdgcn_readlane(rs.roff, 4 * k);
      const float cl = bcast(rs.c_lo, 4 * k), ch = bcast(rs.c_hi, 4 * k);
      const int rr1 = min(rr + 1, S - 1);
      const float4 *r0 = slab4 + (int64_t)rr * (2 * kC / 4) + kC / 4;
      const float4 *r1 = slab4 + (int64_t)rr1 * (2 * kC / 4) + kC / 4;
      const float4 l0 = r0[lane], l1 = r0[64 + lane], u0 = r1[lane], u1 = r1[64 + lane];
      s = fmaf(qa.x, fast_tanh(fmaf(cl, l0.x, fmaf(ch, u0.x, ta.x))), s);
      s = fmaf(qa.y, fast_tanh(fmaf(cl, l0.y, fmaf(ch, u0.y, ta.y))), s);
      s = fmaf(qa.z, fast_tanh(fmaf(cl, l0.z, fmaf(ch, u0.z, ta.z))), s);
      s = fmaf(qa.w, fast_tanh(fmaf(cl, l0.w, fmaf(ch, u0.w, ta.w))), s);
      s = fmaf(qb.x, fast_tanh(fmaf(cl, l1.x, fmaf(ch, u1.x, tb.x))), s);
      s = fmaf(qb.y, fast_tanh(fmaf(cl, l1.y, fmaf(ch, u1.y, tb.y))), s);
      s = fmaf(qb.z, fast_tanh(fmaf(cl, l1.z, fmaf(ch, u1.z, tb.z))), s);
      s = fmaf(qb.w, fast_tanh(fmaf(cl, l1.w, fmaf(ch, u1.w, tb.w))), s);
    }
    s = wave_sum(s);
    if (lane == 0) sh_e[k] = s;
  }
  __syncthreads();
  float ek = k_own < LP ? sh_e[k_own] + alpha_b[0] : -INFINITY;     // lane group k holds e_k, as in the inference kernel
  const float m = groups_max(ek);
  const float pexp = (k_own < LP) ? __expf(ek - m) : 0.f;
  const float alpha = pexp / groups_sum(pexp);
  if (wave == 0 && (lane & 3) == 0) alpha_out[row * kLP + k_own] = alpha;
  const float a_lo = alpha * rs.c_lo, a_hi = alpha * rs.c_hi;
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < kPerWave; ++i) {
    const int k = wave * kPerWave + i;
    if (k < LP) {
      const int rr = __builtin_amdgcn_readlane(rs.roff, 4 * k);
      const float cl = bcast(a_lo, 4 * k), ch = bcast(a_hi, 4 * k);
      const int rr1 = min(rr + 1, S - 1);
      const float4 *r0 = slab4 + (int64_t)rr * (2 * kC / 4);
      const float4 *r1 = slab4 + (int64_t)rr1 * (2 * kC / 4);
      const float4 l0 = r0[lane], l1 = r0[64 + lane], u0 = r1[lane], u1 = r1[64 + lane];
      acc[0] = fmaf(cl, l0.x, fmaf(ch, u0.x, acc[0])); acc[1] = fmaf(cl, l0.y, fmaf(ch, u0.y, acc[1]));
      acc[2] = fmaf(cl, l0.z, fmaf(ch, u0.z, acc[2])); acc[3] = fmaf(cl, l0.w, fmaf(ch, u0.w, acc[3]));
      acc[4] = fmaf(cl, l1.x, fmaf(ch, u1.x, acc[4])); acc[5] = fmaf(cl, l1.y, fmaf(ch, u1.y, acc[5]));
      acc[6] = fmaf(cl, l1.z, fmaf(ch, u1.z, acc[6])); acc[7] = fmaf(cl, l1.w, fmaf(ch, u1.w, acc[7]));
    }
  }
#pragma unroll
  for (int c = 0; c < 8; ++c) sh_acc[wave][lane][c] = acc[c];
  __syncthreads();
  // wavefront w finishes channels pair (2w, 2w+1) of every lane: sum over the four partial wavefronts
  {
    const int c0 = 2 * wave;
    float r0 = 0.f, r1 = 0.f;
#pragma unroll
    for (int w = 0; w < kRowWaves; ++w) { r0 += sh_acc[w][lane][c0]; r1 += sh_acc[w][lane][c0 + 1]; }
    // acc[0..3] -> channels 4*lane + 0..3, acc[4..7] -> channels 256 + 4*lane + 0..3
    float *o = att_res + row * kC + (c0 < 4 ? 4 * lane + c0 : 256 + 4 * lane + (c0 - 4));
    o[0] = r0;
    o[1] = r1;
  }
}

// ------------------------------------------------------------------------------------------------------
// Channel layout of k_cap_train_bwd: lane holds channels {lane + 64 c, c = 0..7} of a 512-vector (a = c 0..3, b = c 4..7),
// NOT the float4 layout of the forward kernel: every per-channel operation is layout-blind, and with this one a wave's
// atomic instruction covers 256 contiguous bytes (2 cache lines) instead of 64 dwords 16 bytes apart (8 lines) --
// the scatter is bound by line requests at the L2 atomic units (56 -> 25 us per call, 18 us without any atomics).
__device__ inline void ld8s(const float *v, int lane, float4 &a, float4 &b) {
  a = make_float4(v[lane], v[64 + lane], v[128 + lane], v[192 + lane]);
  b = make_float4(v[256 + lane], v[320 + lane], v[384 + lane], v[448 + lane]);
}
__device__ inline void atomic_add8s(float *v, int lane, float s, const float4 &a, const float4 &b) {
  atomicAdd(v + lane, s * a.x); atomicAdd(v + 64 + lane, s * a.y); atomicAdd(v + 128 + lane, s * a.z);
  atomicAdd(v + 192 + lane, s * a.w); atomicAdd(v + 256 + lane, s * b.x); atomicAdd(v + 320 + lane, s * b.y);
  atomicAdd(v + 384 + lane, s * b.z); atomicAdd(v + 448 + lane, s * b.w);
}
__device__ inline float dot4d(const float4 &g, const float4 &u, const float4 &l) {
  return g.x * (u.x - l.x) + g.y * (u.y - l.y) + g.z * (u.z - l.z) + g.w * (u.w - l.w);
}
__device__ inline float dot4(const float4 &a, const float4 &b) { return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w; }

__global__ void __launch_bounds__(kRowWaves * 64) k_cap_train_bwd(
    const float *__restrict__ slab, const int64_t *__restrict__ shapes, const int64_t *__restrict__ lsi,
    const float *__restrict__ ref, const float *__restrict__ off_hs, const float *__restrict__ off_h, int off_h_ld,
    const float *__restrict__ att_h, int att_h_ld, const float *__restrict__ alpha_w,
    const float *__restrict__ alpha_saved, const float *__restrict__ g_att, int g_att_ld, int S, int L, int Q, int P,
    int RD, const int64_t *__restrict__ row_video, float *__restrict__ g_slab, float *__restrict__ g_att_h,
    int g_att_h_ld, float *__restrict__ g_off, int g_off_ld, float *__restrict__ g_ref,
    float *__restrict__ g_alpha_w, float *__restrict__ g_alpha_b) {
  __shared__ float sh_da[kLP], sh_dx[kLP];
  __shared__ float sh_part[kRowWaves][64][16];
  const int64_t row = blockIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, k_own = lane >> 2, LP = L * P;
  const int b = row_video ? (int)row_video[row] : (int)(row / Q);
  if (b < 0) {                                       // unused row of a padded row set: its overwritten outputs are zero
    g_att_h[row * (int64_t)g_att_h_ld + threadIdx.x] = 0.f;
    g_att_h[row * (int64_t)g_att_h_ld + 256 + threadIdx.x] = 0.f;
    if (threadIdx.x < kLP) g_off[row * (int64_t)g_off_ld + threadIdx.x] = 0.f;
    return;
  }
  const RowSetup rs = setup_row(shapes, lsi, ref, off_hs, off_h + row * (int64_t)off_h_ld, row, k_own, L, P, RD);
  const float alpha = k_own < LP ? alpha_saved[row * kLP + k_own] : 0.f;
  float4 ga, gb;
  ld8s(g_att + row * (int64_t)g_att_ld, lane, ga, gb);
  const float *slab_b = slab + (int64_t)b * S * (2 * kC);
  float *gs = g_slab + (int64_t)b * S * (2 * kC);

  // ---- value half: d alpha_k = g . clip_k ; d x_k (part 1) = alpha_k g . (V[r+1] - V[r]) ; scatter alpha_k c g -----
  float px[kPerWave];
#pragma unroll
  for (int i = 0; i < kPerWave; ++i) {
    const int k = wave * kPerWave + i;
    float pa = 0.f;
    px[i] = 0.f;
    if (k < LP) {
      const int rr = __builtin_amdgcn_readlane(rs.roff, 4 * k);
      const float cl = bcast(rs.c_lo, 4 * k), ch = bcast(rs.c_hi, 4 * k), ak = bcast(alpha, 4 * k);
      const int rr1 = min(rr + 1, S - 1);
      float4 l0, l1, u0, u1;
      ld8s(slab_b + (int64_t)rr * (2 * kC), lane, l0, l1);
      ld8s(slab_b + (int64_t)rr1 * (2 * kC), lane, u0, u1);
      pa = cl * (dot4(ga, l0) + dot4(gb, l1)) + ch * (dot4(ga, u0) + dot4(gb, u1));
      px[i] = ak * (dot4d(ga, u0, l0) + dot4d(gb, u1, l1));
      float *d0 = gs + (int64_t)rr * (2 * kC), *d1 = gs + (int64_t)rr1 * (2 * kC);
      if (cl != 0.f) atomic_add8s(d0, lane, ak * cl, ga, gb);
      if (ch != 0.f) atomic_add8s(d1, lane, ak * ch, ga, gb);
    }
    pa = wave_sum(pa);
    if (lane == 0) sh_da[k] = pa;
  }
  __syncthreads();
  const float dalpha = k_own < LP ? sh_da[k_own] : 0.f;             // lane group k: g . clip_k
  const float dsum = groups_sum(alpha * dalpha);
  const float de = alpha * (dalpha - dsum);                         // softmax backward: d e_k
  const float de_total = groups_sum(de);                            // = 0 up to rounding: softmax is shift invariant
  if (threadIdx.x == 0) atomicAdd(g_alpha_b, de_total);

  // ---- ctx2att half: t = tanh(att_ctx_k + att_h); d pre = de_k alpha_w (1 - t^2) -------------------------------
  float4 ta, tb, qa, qb;
  ld8s(att_h + row * (int64_t)att_h_ld, lane, ta, tb);
  ld8s(alpha_w, lane, qa, qb);
  float4 dha = make_float4(0.f, 0.f, 0.f, 0.f), dhb = dha, dwa = dha, dwb = dha;
#pragma unroll
  for (int i = 0; i < kPerWave; ++i) {
    const int k = wave * kPerWave + i;
    float p2 = 0.f;
    if (k < LP) {
      const int rr = __builtin_amdgcn_readlane(rs.roff, 4 * k);
      const float cl = bcast(rs.c_lo, 4 * k), ch = bcast(rs.c_hi, 4 * k), dek = bcast(de, 4 * k);
      const int rr1 = min(rr + 1, S - 1);
      float4 l0, l1, u0, u1;
      ld8s(slab_b + (int64_t)rr * (2 * kC) + kC, lane, l0, l1);
      ld8s(slab_b + (int64_t)rr1 * (2 * kC) + kC, lane, u0, u1);
      float4 da, db;
#define GVL_CH(X, Lo, Up, Tt, Qq, Dd, DH, DW)                                  \
      {                                                                        \
        const float t_ = fast_tanh(fmaf(cl, Lo.X, fmaf(ch, Up.X, Tt.X)));      \
        const float d_ = dek * Qq.X * (1.f - t_ * t_);                         \
        Dd.X = d_; DH.X += d_; DW.X = fmaf(dek, t_, DW.X);                     \
      }
      GVL_CH(x, l0, u0, ta, qa, da, dha, dwa) GVL_CH(y, l0, u0, ta, qa, da, dha, dwa)
      GVL_CH(z, l0, u0, ta, qa, da, dha, dwa) GVL_CH(w, l0, u0, ta, qa, da, dha, dwa)
      GVL_CH(x, l1, u1, tb, qb, db, dhb, dwb) GVL_CH(y, l1, u1, tb, qb, db, dhb, dwb)
      GVL_CH(z, l1, u1, tb, qb, db, dhb, dwb) GVL_CH(w, l1, u1, tb, qb, db, dhb, dwb)
#undef GVL_CH
      p2 = dot4d(da, u0, l0) + dot4d(db, u1, l1);
      float *d0 = gs + (int64_t)rr * (2 * kC) + kC, *d1 = gs + (int64_t)rr1 * (2 * kC) + kC;
      if (cl != 0.f) atomic_add8s(d0, lane, cl, da, db);
      if (ch != 0.f) atomic_add8s(d1, lane, ch, da, db);
    }
    const float dxk = wave_sum(px[i] + p2);                          // d loss / d pixel coordinate of sample k
    if (lane == 0) sh_dx[k] = dxk;
  }
  {
    float *pp = sh_part[wave][lane];
    pp[0] = dha.x; pp[1] = dha.y; pp[2] = dha.z; pp[3] = dha.w; pp[4] = dhb.x; pp[5] = dhb.y; pp[6] = dhb.z; pp[7] = dhb.w;
    pp[8] = dwa.x; pp[9] = dwa.y; pp[10] = dwa.z; pp[11] = dwa.w; pp[12] = dwb.x; pp[13] = dwb.y; pp[14] = dwb.z; pp[15] = dwb.w;
  }
  __syncthreads();
  // wavefront w finishes slots 4w..4w+3 of every lane (slots 0-7: d h2att(h) channels, 8-15: d alpha_w channels)
  {
    float r[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int w = 0; w < kRowWaves; ++w)
#pragma unroll
      for (int c = 0; c < 4; ++c) r[c] += sh_part[w][lane][4 * wave + c];
    const int ch0 = (wave & 1) * 256 + lane;                        // slot c of a lane <-> channel 64 c + lane
    if (wave < 2) {
      float *oh = g_att_h + row * (int64_t)g_att_h_ld + ch0;
      oh[0] = r[0]; oh[64] = r[1]; oh[128] = r[2]; oh[192] = r[3];
    } else {
      atomicAdd(g_alpha_w + ch0 + 0, r[0]); atomicAdd(g_alpha_w + ch0 + 64, r[1]);
      atomicAdd(g_alpha_w + ch0 + 128, r[2]); atomicAdd(g_alpha_w + ch0 + 192, r[3]);
    }
  }
  // ---- d x_k -> offsets and reference points (wavefront 0, lane group k) ---------------------------------------
  if (wave == 0 && (lane & 3) == 0) {
    const float dloc = (k_own < LP ? sh_dx[k_own] : 0.f) * rs.dmul;
    g_off[row * (int64_t)g_off_ld + k_own] = k_own < LP ? dloc * rs.doff : 0.f;
    if (k_own < LP) {
      float *gr = g_ref + (row * L + rs.level) * RD;
      atomicAdd(gr, dloc);
      if (rs.with_len) atomicAdd(gr + 1, dloc * rs.off * (0.5f / (float)P));
    }
  }
}

// ------------------------------------------------------------------------------------------------------
// LSTM cell, training: gates = ga + gb + gc (three partial pre-activations, row strides in floats), order i,f,g,o
__global__ void __launch_bounds__(256) k_lstm_train_fwd(const float *__restrict__ ga, int lda, const float *__restrict__ gb,
                                                        int ldb, const float *__restrict__ gc, int ldc,
                                                        const float *__restrict__ c_prev, int n, int H,
                                                        float *__restrict__ act, float *__restrict__ h_out,
                                                        float *__restrict__ c_out) {
  const int64_t total = (int64_t)n * H;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int row = (int)(idx / H), j = (int)(idx % H);
    float g[4];
#pragma unroll
    for (int k = 0; k < 4; ++k)
      g[k] = ga[(int64_t)row * lda + k * H + j] + gb[(int64_t)row * ldb + k * H + j] + gc[(int64_t)row * ldc + k * H + j];
    const float i_ = sigmoidf_(g[0]), f_ = sigmoidf_(g[1]), g_ = fast_tanh(g[2]), o_ = sigmoidf_(g[3]);
    const float c = f_ * c_prev[idx] + i_ * g_;
    float *a = act + (int64_t)row * 4 * H + j;
    a[0] = i_; a[H] = f_; a[2 * H] = g_; a[3 * H] = o_;
    c_out[idx] = c;
    h_out[idx] = o_ * fast_tanh(c);
  }
}

// dh = dh_a + dh_b (either may be null), dc_in may be null; writes d(pre-activation gates) (row stride ldg) and dc_prev.
// acc (n, 4H) or null: the running sum of the gate gradients over the token steps (acc_first: this launch starts it) -- the
// gradient of the token-independent gate part, which would otherwise be a reduction over (steps, n, 4H) after the loop
__global__ void __launch_bounds__(256) k_lstm_train_bwd(const float *__restrict__ dh_a, const float *__restrict__ dh_b,
                                                        const float *__restrict__ dc_in, const float *__restrict__ act,
                                                        const float *__restrict__ c_prev, const float *__restrict__ c_new,
                                                        int n, int H, float *__restrict__ dgates, int ldg,
                                                        float *__restrict__ dc_prev, float *__restrict__ acc, int acc_first) {
  const int64_t total = (int64_t)n * H;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int row = (int)(idx / H), j = (int)(idx % H);
    const float *a = act + (int64_t)row * 4 * H + j;
    const float i_ = a[0], f_ = a[H], g_ = a[2 * H], o_ = a[3 * H];
    const float dh = (dh_a ? dh_a[idx] : 0.f) + (dh_b ? dh_b[idx] : 0.f);
    const float tc = fast_tanh(c_new[idx]);
    const float dc = (dc_in ? dc_in[idx] : 0.f) + dh * o_ * (1.f - tc * tc);
    float *d = dgates + (int64_t)row * ldg + j;
    const float gi = dc * g_ * i_ * (1.f - i_), gf = dc * c_prev[idx] * f_ * (1.f - f_), gg = dc * i_ * (1.f - g_ * g_),
                go = dh * tc * o_ * (1.f - o_);
    d[0] = gi; d[H] = gf; d[2 * H] = gg; d[3 * H] = go;
    dc_prev[idx] = dc * f_;
    if (acc) {
      float *s = acc + (int64_t)row * 4 * H + j;
      if (acc_first) { s[0] = gi; s[H] = gf; s[2 * H] = gg; s[3 * H] = go; }
      else { s[0] += gi; s[H] += gf; s[2 * H] += gg; s[3 * H] += go; }
    }
  }
}

int check_cap(const char *what, int B, int S, int C, int L, int Q, int P, int RD) {
  if (C != kC || L * P > kLP || L <= 0 || P <= 0 || (RD != 1 && RD != 2) || B < 0 || Q < 0 || S <= 0)
    return fail(GVL_EINVAL, "%s: unsupported shape C=%d L=%d P=%d RD=%d (need C=512, L*P<=16)", what, C, L, P, RD);
  return 0;
}

}  // namespace

extern "C" {

int gvl_cap_attend_train_forward_f32(const float *slab, const int64_t *shapes, const int64_t *lsi, const float *ref,
                                     const float *off_hs, const float *off_h, int off_h_ld, const float *att_h,
                                     int att_h_ld, const float *alpha_w, const float *alpha_b, int B, int S, int C,
                                     int L, int Q, int P, int RD, const int64_t *row_video, float *att_res,
                                     float *alpha_out, void *stream) {
  if (int rc = check_cap("gvl_cap_attend_train_forward_f32", row_video ? 1 : B, S, C, L, Q, P, RD)) return rc;
  const int64_t n_rows = row_video ? (int64_t)Q : (int64_t)B * Q;
  if (att_h_ld < C || (att_h_ld & 3) || off_h_ld < L * P)
    return fail(GVL_EINVAL, "gvl_cap_attend_train_forward_f32: bad leading dimensions");
  if (n_rows == 0) return 0;
  if (!slab || !shapes || !lsi || !ref || !off_hs || !off_h || !att_h || !alpha_w || !alpha_b || !att_res || !alpha_out)
    return fail(GVL_EINVAL, "gvl_cap_attend_train_forward_f32: null pointer");
  if (((uintptr_t)att_h & 15) || ((uintptr_t)att_res & 15))
    return fail(GVL_EINVAL, "gvl_cap_attend_train_forward_f32: att_h / att_res must be 16-byte aligned");
  return gvl::launch(GVL_PROF_CAP_TRAIN_FWD, (int)n_rows, B, "k_cap_train_fwd", k_cap_train_fwd, dim3((unsigned)n_rows),
                     dim3(256), 0, (hipStream_t)stream, slab, shapes, lsi, ref, off_hs, off_h, off_h_ld, att_h, att_h_ld,
                     alpha_w, alpha_b, S, L, Q, P, RD, row_video, att_res, alpha_out);
}

int gvl_cap_attend_train_backward_f32(const float *slab, const int64_t *shapes, const int64_t *lsi, const float *ref,
                                      const float *off_hs, const float *off_h, int off_h_ld, const float *att_h,
                                      int att_h_ld, const float *alpha_w, const float *alpha_saved,
                                      const float *grad_att_res, int grad_att_res_ld, int B, int S, int C, int L, int Q,
                                      int P, int RD, const int64_t *row_video, float *grad_slab, float *grad_att_h,
                                      int grad_att_h_ld,
                                      float *grad_off, int grad_off_ld, float *grad_ref, float *grad_alpha_w,
                                      float *grad_alpha_b, void *stream) {
  if (int rc = check_cap("gvl_cap_attend_train_backward_f32", row_video ? 1 : B, S, C, L, Q, P, RD)) return rc;
  const int64_t n_rows = row_video ? (int64_t)Q : (int64_t)B * Q;
  if (att_h_ld < C || (att_h_ld & 3) || off_h_ld < L * P || grad_att_res_ld < C || (grad_att_res_ld & 3) ||
      grad_att_h_ld < C || (grad_att_h_ld & 3) || grad_off_ld < kLP)
    return fail(GVL_EINVAL, "gvl_cap_attend_train_backward_f32: bad leading dimensions");
  if (n_rows == 0) return 0;
  if (!slab || !shapes || !lsi || !ref || !off_hs || !off_h || !att_h || !alpha_w || !alpha_saved || !grad_att_res ||
      !grad_slab || !grad_att_h || !grad_off || !grad_ref || !grad_alpha_w || !grad_alpha_b)
    return fail(GVL_EINVAL, "gvl_cap_attend_train_backward_f32: null pointer");
  if (((uintptr_t)att_h & 15) || ((uintptr_t)grad_att_res & 15) || ((uintptr_t)grad_att_h & 15))
    return fail(GVL_EINVAL, "gvl_cap_attend_train_backward_f32: row pointers must be 16-byte aligned");
  return gvl::launch(GVL_PROF_CAP_TRAIN_BWD, (int)n_rows, B, "k_cap_train_bwd", k_cap_train_bwd, dim3((unsigned)n_rows),
                     dim3(256), 0, (hipStream_t)stream, slab, shapes, lsi, ref, off_hs, off_h, off_h_ld, att_h, att_h_ld,
                     alpha_w, alpha_saved, grad_att_res, grad_att_res_ld, S, L, Q, P, RD, row_video, grad_slab,
                     grad_att_h, grad_att_h_ld, grad_off, grad_off_ld, grad_ref, grad_alpha_w, grad_alpha_b);
}

int gvl_lstm_cell_train_forward_f32(const float *gates_a, int lda, const float *gates_b, int ldb, const float *gates_c,
                                    int ldc, const float *c_prev, int n, int H, float *act, float *h_out, float *c_out,
                                    void *stream) {
  if (n < 0 || H <= 0 || lda < 4 * H || ldb < 4 * H || ldc < 4 * H)
    return fail(GVL_EINVAL, "gvl_lstm_cell_train_forward_f32: bad sizes n=%d H=%d", n, H);
  if (n == 0) return 0;
  if (!gates_a || !gates_b || !gates_c || !c_prev || !act || !h_out || !c_out)
    return fail(GVL_EINVAL, "gvl_lstm_cell_train_forward_f32: null pointer");
  int64_t blocks = ((int64_t)n * H + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  return gvl::launch(GVL_PROF_LSTM_TRAIN, n, H, "k_lstm_train_fwd", k_lstm_train_fwd, dim3((unsigned)blocks), dim3(256),
                     0, (hipStream_t)stream, gates_a, lda, gates_b, ldb, gates_c, ldc, c_prev, n, H, act, h_out, c_out);
}

int gvl_lstm_cell_train_backward_f32(const float *grad_h_a, const float *grad_h_b, const float *grad_c, const float *act,
                                     const float *c_prev, const float *c_new, int n, int H, float *grad_gates,
                                     int grad_gates_ld, float *grad_c_prev, void *stream) {
  return gvl_lstm_cell_train_backward_sum_f32(grad_h_a, grad_h_b, grad_c, act, c_prev, c_new, n, H, grad_gates, grad_gates_ld,
                                              grad_c_prev, nullptr, 0, stream);
}

int gvl_lstm_cell_train_backward_sum_f32(const float *grad_h_a, const float *grad_h_b, const float *grad_c, const float *act,
                                         const float *c_prev, const float *c_new, int n, int H, float *grad_gates,
                                         int grad_gates_ld, float *grad_c_prev, float *grad_gates_sum, int first, void *stream) {
  if (n < 0 || H <= 0 || grad_gates_ld < 4 * H)
    return fail(GVL_EINVAL, "gvl_lstm_cell_train_backward_f32: bad sizes n=%d H=%d", n, H);
  if (n == 0) return 0;
  if (!act || !c_prev || !c_new || !grad_gates || !grad_c_prev)
    return fail(GVL_EINVAL, "gvl_lstm_cell_train_backward_f32: null pointer");
  int64_t blocks = ((int64_t)n * H + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  return gvl::launch(GVL_PROF_LSTM_TRAIN, n, H, "k_lstm_train_bwd", k_lstm_train_bwd, dim3((unsigned)blocks), dim3(256),
                     0, (hipStream_t)stream, grad_h_a, grad_h_b, grad_c, act, c_prev, c_new, n, H, grad_gates,
                     grad_gates_ld, grad_c_prev, grad_gates_sum, first);
}

}  // extern "C"
