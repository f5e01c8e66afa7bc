// gvl_cap.hip -- captioner-side kernels of the GVL hot path on MI355X (gfx950).
//
//   k_cap_attend      the deformable soft attention of one LSTM-DSA token step, fused into one launch
//                     (reference: ShowAttendTellCore.forward, pdvc/CaptioningHead/LSTM_DSA.py:241-266, with
//                     MSDeformAttnCap.forward, pdvc/ops/modules/ms_deform_attn_for_caption.py:82-127, inside it):
//                       off   = sampling_offsets([h | hs])                      (16 scalar offsets per query)
//                       x_k   = ref + off_k / T_l   |  ref_c + off_k / P * ref_len * 0.5
//                       clip_k = border-padded linear sample of value_proj(memory) at x_k          (C floats)
//                       e_k   = alpha_net(tanh(ctx2att(clip_k) + h2att(h)))      softmax over the 16 samples
//                       att   = sum_k alpha_k clip_k
//                     ctx2att is linear and border interpolation weights sum to 1, hence ctx2att(clip_k) is a
//                     sample of ctx2att(value); the kernel reads a [value | ctx2att(value)] slab built once per
//                     forward.  One wavefront owns one (video, query) row: lane holds 8 of the 512 channels, the
//                     16 offset dot-products and the 16 attention logits are reduced with a 17-step butterfly
//                     reduce-scatter (xor shuffles) that leaves sample k on lanes 4k..4k+3, row addresses travel to
//                     SGPRs with v_readlane so every gather is a coalesced 1 KiB wave load.
//   k_row_argmax_lse  greedy decoding epilogue: per row of the logits argmax and log-softmax at the argmax
//                     (LSTM_DSA.py:123 log_softmax + :166 torch.max) in one read of the logits.
//   k_sample_bwd      backward of the unweighted sampler (autograd of ms_deform_attn_core_pytorch(return_value=True),
//                     func.py:44-68, used by the teacher-forced captioner in training).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <type_traits>

#include "gvl_common.hpp"
#include "gvl_msda.h"

namespace {

using gvl::fail;

__device__ inline float fast_tanh(float x) {
  // 1 - 2/(1+e^{2x}); saturates correctly for |x| large, abs error ~1e-7
  const float e = __expf(2.f * x);
  return fmaf(-2.f, __builtin_amdgcn_rcpf(1.f + e), 1.f);      // v_rcp_f32 (1 ulp), not a ~10-instruction IEEE division
}

// reduce-scatter of v[16] over the 64 lanes: afterwards every lane holds the full sum of v[k], k = lane >> 2.
__device__ inline float butterfly16(float (&v)[16], int lane) {
  float a8[8];
  const bool up5 = lane & 32;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const float mine = up5 ? v[8 + i] : v[i];
    const float send = up5 ? v[i] : v[8 + i];
    a8[i] = mine + __shfl_xor(send, 32, 64);
  }
  float a4[4];
  const bool up4 = lane & 16;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float mine = up4 ? a8[4 + i] : a8[i];
    const float send = up4 ? a8[i] : a8[4 + i];
    a4[i] = mine + __shfl_xor(send, 16, 64);
  }
  float a2[2];
  const bool up3 = lane & 8;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const float mine = up3 ? a4[2 + i] : a4[i];
    const float send = up3 ? a4[i] : a4[2 + i];
    a2[i] = mine + __shfl_xor(send, 8, 64);
  }
  const bool up2 = lane & 4;
  float r = (up2 ? a2[1] : a2[0]) + __shfl_xor(up2 ? a2[0] : a2[1], 4, 64);
  r += __shfl_xor(r, 2, 64);
  r += __shfl_xor(r, 1, 64);
  return r;
}

// all-reduce over lane groups that differ in bits 2..5 (values are already uniform inside each group of 4 lanes)
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

__device__ inline void fma8(float a, const float4 &x0, const float4 &x1, float (&acc)[8]) {
  acc[0] = fmaf(a, x0.x, acc[0]); acc[1] = fmaf(a, x0.y, acc[1]); acc[2] = fmaf(a, x0.z, acc[2]);
  acc[3] = fmaf(a, x0.w, acc[3]); acc[4] = fmaf(a, x1.x, acc[4]); acc[5] = fmaf(a, x1.y, acc[5]);
  acc[6] = fmaf(a, x1.z, acc[6]); acc[7] = fmaf(a, x1.w, acc[7]);
}

constexpr int kC = 512;      // channels of the value half and of the ctx2att half (hidden_dim = att_hid_size = 512)
constexpr int kLP = 16;      // samples per query (cap_num_feature_levels * cap_dec_n_points)
constexpr int kWaves = 4;    // rows (wavefronts) in flight per workgroup; three workgroups per CU (152 VGPRs)

// border-mode coefficients of one temporal sample (grid_sampler border, align_corners=False), row pair (r, r+1)
__device__ inline void border_coef(float loc, int T, int &r, float &c_lo, float &c_hi) {
  const float g = 2.f * loc - 1.f;
  float x = ((g + 1.f) * (float)T - 1.f) * 0.5f;
  const float mx = (float)(T - 1);
  x = !(x > 0.f) ? 0.f : (x >= mx ? mx : x);
  const float xf = floorf(x);
  const int x0 = (int)xf;
  const float a = x - xf;
  const int rmax = T >= 2 ? T - 2 : 0;
  r = x0 > rmax ? rmax : x0;
  const float t0 = 1.f - a;
  const float t1 = (x0 + 1 <= T - 1) ? a : 0.f;
  c_lo = (x0 == r ? t0 : 0.f) + (x0 + 1 == r ? t1 : 0.f);
  c_hi = (x0 == r + 1 ? t0 : 0.f) + (x0 + 1 == r + 1 ? t1 : 0.f);
}

// ST: storage type of the slab and of att_h (fp32, or bf16 as a GEMM under autocast leaves them); arithmetic fp32
// fp32 -> two fp16 parts with x = s (hi + 2^-11 lo), the operand form of gvl_gemm_f16x3_f32 (gvl_gemm16.hip): `inv` = 1 / s
typedef _Float16 half4_t __attribute__((ext_vector_type(4)));
__device__ inline void split4_f16(const float4 v, float inv, half4_t &hi, half4_t &lo) {
  const float a[4] = {v.x * inv, v.y * inv, v.z * inv, v.w * inv};
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    hi[c] = (_Float16)a[c];
    lo[c] = (_Float16)((a[c] - (float)hi[c]) * 2048.f);
  }
}
// power-of-two scale of a row whose largest magnitude is m: s = 2^floor(log2 m) (exponent field clamped so that 1 / s exists)
__device__ inline void pow2_scale(float m, float &s, float &inv) {
  int e = (int)((__float_as_uint(m) >> 23) & 0xffu);
  e = min(max(e, 1), 253);
  s = __uint_as_float((uint32_t)e << 23);
  inv = __uint_as_float((uint32_t)(254 - e) << 23);
}

// FULL: L*P == 16 known at compile time -- the sample loops lose their guards and become straight-line code, so the
// eight loads of a sample pair really are in flight together
template <typename ST, bool FULL>
__global__ void __launch_bounds__(kWaves * 64, (FULL && sizeof(ST) == 4) ? 3 : 2) k_cap_attend(
    const ST *__restrict__ slab,         // (B, S, 2C)  [value_proj(memory) | ctx2att(value_proj(memory))]
    const int64_t *__restrict__ shapes,  // (L, 2)
    const int64_t *__restrict__ lsi,     // (L)
    const float *__restrict__ ref,       // (B, Q, L, RD) reference points scaled by the valid ratios
    const float *__restrict__ off_hs,    // (B*Q, 16)   sampling_offsets bias + hs part
    const float *__restrict__ h,         // (B*Q, C)    previous hidden state
    const float *__restrict__ w_off_h,   // (16, C)     sampling_offsets.weight[:, :C]
    const ST *__restrict__ att_h,        // (B*Q, C)    h2att(h), row stride att_h_ld elements
    const float *__restrict__ alpha_w,   // (C)
    float alpha_b, int B, int S, int L, int Q, int P, int RD, int rows_per_xcd_group, int att_h_ld,
    ST *__restrict__ att_res,            // (B*Q, C), in the storage type: it is the A operand of the next GEMM
    float *__restrict__ dbg_alpha,       // optional (B*Q, 16)
    float *__restrict__ dbg_loc,         // optional (B*Q, 16)
    _Float16 *__restrict__ o_hi,         // optional: the result as the fp16 planes + row scale of gvl_gemm_f16x3_f32
    _Float16 *__restrict__ o_lo,         //           INSTEAD of att_res (fp32 kernel only)
    float *__restrict__ o_scale) {
  __shared__ float4 wo4[kLP * kC / 4];   // 32 KiB: the h part of the offsets projection
  for (int i = threadIdx.x; i < kLP * kC / 4; i += blockDim.x) wo4[i] = reinterpret_cast<const float4 *>(w_off_h)[i];

  // XCD-aware row mapping: workgroups b and b+8 share an XCD (round-robin dispatch), so give every XCD a
  // contiguous group of videos; its L2 then holds only those videos' slabs.
  const int xcd = blockIdx.x & 7, jblk = blockIdx.x >> 3;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  // PERSISTENT: the XCD's workgroups (gridDim / 8 of them, all resident) walk its rows with a stride of one row per
  // wavefront slot -- the 32 KB above are staged once per workgroup, and 4800 rows over 3072 slots are 1.6 rounds of a
  // latency-bound row instead of the 3 rounds of 600 one-shot workgroups of 8 rows at two wavefronts per SIMD
  const int slots = (int)(gridDim.x >> 3) * kWaves;
  for (int lr = jblk * kWaves + wave; lr < rows_per_xcd_group; lr += slots) {
  const int64_t row = (int64_t)xcd * rows_per_xcd_group + lr;
  if (row >= (int64_t)B * Q) break;
  const int b = (int)(row / Q);

  // own channels: [4*lane, 4*lane+4) and [256 + 4*lane, 256 + 4*lane + 4)
  const float4 *h4 = reinterpret_cast<const float4 *>(h + row * kC);
  const float4 ha = h4[lane], hb = h4[64 + lane];
  float part[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    const float4 wa = wo4[k * (kC / 4) + lane], wb = wo4[k * (kC / 4) + 64 + lane];
    part[k] = wa.x * ha.x + wa.y * ha.y + wa.z * ha.z + wa.w * ha.w + wb.x * hb.x + wb.y * hb.y + wb.z * hb.z +
              wb.w * hb.w;
  }
  const int k_own = lane >> 2;                       // the sample this lane group owns after the butterfly
  float off = butterfly16(part, lane);
  int roff = 0;
  float c_lo = 0.f, c_hi = 0.f, locx = 0.f;
  const int LP = FULL ? kLP : L * P;
  if (k_own < LP) {
    off += off_hs[row * LP + k_own];
    const int l = k_own / P;
    const int T = (int)shapes[2 * l + 1];
    const float *rp = ref + (row * L + l) * RD;
    if (RD == 1) locx = rp[0] + off / (float)T;                              // ms_deform_attn_for_caption.py:108-109
    else locx = rp[0] + off / (float)P * rp[1] * 0.5f;                       // :110-112
    int r;
    border_coef(locx, T, r, c_lo, c_hi);
    roff = (int)lsi[l] + r;
  }
  if (dbg_loc && (lane & 3) == 0 && k_own < LP) dbg_loc[row * LP + k_own] = locx;

  // ---- pass 1: attention logits from the ctx2att half ------------------------------------------------------
  const ST *ah = att_h + row * (int64_t)att_h_ld;
  const float4 ta = ld4(ah, lane), tb = ld4(ah, 64 + lane);
  const float4 qa = reinterpret_cast<const float4 *>(alpha_w)[lane], qb = reinterpret_cast<const float4 *>(alpha_w)[64 + lane];
  const ST *slab_b = slab + (int64_t)b * S * (2 * kC);                // this video's slab; ld4 indexes groups of 4 elements
  // Samples are taken two at a time with all eight 16-byte loads requested before any arithmetic: a wavefront walks a
  // chain of 32 dependent L2 round trips otherwise (4 loads in flight), and the kernel is bound by that latency
  struct Rows { float4 l0, l1, u0, u1; };
  auto load_rows = [&](int k, int half4, float &cl, float &ch) {
    const int rr = __builtin_amdgcn_readlane(roff, 4 * k);
    cl = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, c_lo), 4 * k));
    ch = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, c_hi), 4 * k));
    const int rr1 = min(rr + 1, S - 1);             // c_hi == 0 whenever rr + 1 leaves the level (T_l == 1)
    const int64_t r0 = (int64_t)rr * (2 * kC / 4) + half4, r1 = (int64_t)rr1 * (2 * kC / 4) + half4;
    Rows r;
    r.l0 = ld4(slab_b, r0 + lane); r.l1 = ld4(slab_b, r0 + 64 + lane);
    r.u0 = ld4(slab_b, r1 + lane); r.u1 = ld4(slab_b, r1 + 64 + lane);
    return r;
  };
  auto logit_part = [&](const Rows &r, float cl, float ch) {
    float s = 0.f;
    s = fmaf(qa.x, fast_tanh(fmaf(cl, r.l0.x, fmaf(ch, r.u0.x, ta.x))), s);
    s = fmaf(qa.y, fast_tanh(fmaf(cl, r.l0.y, fmaf(ch, r.u0.y, ta.y))), s);
    s = fmaf(qa.z, fast_tanh(fmaf(cl, r.l0.z, fmaf(ch, r.u0.z, ta.z))), s);
    s = fmaf(qa.w, fast_tanh(fmaf(cl, r.l0.w, fmaf(ch, r.u0.w, ta.w))), s);
    s = fmaf(qb.x, fast_tanh(fmaf(cl, r.l1.x, fmaf(ch, r.u1.x, tb.x))), s);
    s = fmaf(qb.y, fast_tanh(fmaf(cl, r.l1.y, fmaf(ch, r.u1.y, tb.y))), s);
    s = fmaf(qb.z, fast_tanh(fmaf(cl, r.l1.z, fmaf(ch, r.u1.z, tb.z))), s);
    s = fmaf(qb.w, fast_tanh(fmaf(cl, r.l1.w, fmaf(ch, r.u1.w, tb.w))), s);
    return s;
  };
  float e[16];
  constexpr int kBatch = 2;                                          // samples whose 4 kBatch loads are in flight together (4: measured the same)
#pragma unroll
  for (int k = 0; k < 16; k += kBatch) {
    float cl[kBatch], ch[kBatch];
    Rows rr[kBatch];
#pragma unroll
    for (int t = 0; t < kBatch; ++t) {
      e[k + t] = 0.f;
      if (k + t < LP) rr[t] = load_rows(k + t, kC / 4, cl[t], ch[t]);  // ctx2att half of the slab rows
    }
    __builtin_amdgcn_sched_barrier(0);                               // keep the loads ahead of the arithmetic
#pragma unroll
    for (int t = 0; t < kBatch; ++t)
      if (k + t < LP) e[k + t] = logit_part(rr[t], cl[t], ch[t]);
  }
  float ek = butterfly16(e, lane) + alpha_b;
  if (k_own >= LP) ek = -INFINITY;
  const float m = groups_max(ek);
  const float pexp = (k_own < LP) ? __expf(ek - m) : 0.f;
  const float alpha = pexp / groups_sum(pexp);
  if (dbg_alpha && (lane & 3) == 0 && k_own < LP) dbg_alpha[row * LP + k_own] = alpha;
  const float a_lo = alpha * c_lo, a_hi = alpha * c_hi;

  // ---- pass 2: weighted sum of the value half --------------------------------------------------------------
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  auto coef = [&](int k, float &cl, float &ch) {
    cl = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, a_lo), 4 * k));
    ch = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, a_hi), 4 * k));
  };
#pragma unroll
  for (int k = 0; k < 16; k += kBatch) {
    float d0, d1;
    Rows rr[kBatch];
#pragma unroll
    for (int t = 0; t < kBatch; ++t)
      if (k + t < LP) rr[t] = load_rows(k + t, 0, d0, d1);           // value half of the slab rows
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int t = 0; t < kBatch; ++t)
      if (k + t < LP) {
        float cl, ch;
        coef(k + t, cl, ch);
        fma8(cl, rr[t].l0, rr[t].l1, acc);
        fma8(ch, rr[t].u0, rr[t].u1, acc);
      }
  }
  if (o_hi) {
    // the wavefront owns the whole row: its largest magnitude is one reduction away
    float m = 0.f;
#pragma unroll
    for (int c = 0; c < 8; ++c) m = fmaxf(m, fabsf(acc[c]));
#pragma unroll
    for (int o = 32; o; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    float sc, inv;
    pow2_scale(m, sc, inv);
    if (lane == 0) o_scale[row] = sc;
    half4_t h0, l0, h1, l1;
    split4_f16(make_float4(acc[0], acc[1], acc[2], acc[3]), inv, h0, l0);
    split4_f16(make_float4(acc[4], acc[5], acc[6], acc[7]), inv, h1, l1);
    // K-stage-major planes (gvl_gemm16_common.hpp: plane_off): channel k of row r at ((k >> 5) n + r) 32 + (k & 31)
    const int64_t at0 = ((int64_t)(lane >> 3) * ((int64_t)B * Q) + row) * 32 + 4 * (lane & 7), at1 = at0 + (int64_t)B * Q * 256;
    *reinterpret_cast<half4_t *>(o_hi + at0) = h0; *reinterpret_cast<half4_t *>(o_hi + at1) = h1;
    *reinterpret_cast<half4_t *>(o_lo + at0) = l0; *reinterpret_cast<half4_t *>(o_lo + at1) = l1;
  } else {
    ST *o = att_res + row * kC;
    st4(o, lane, make_float4(acc[0], acc[1], acc[2], acc[3]));
    st4(o, 64 + lane, make_float4(acc[4], acc[5], acc[6], acc[7]));
  }
  }
}

// k_cap_attend_lds -- the same token step with the COARSE end of the video's slab resident in LDS.  k_cap_attend is bound
// by the CU's vector-memory path: 630 MB of slab rows per token through 256 texture-address units at ~70 GB/s each are
// 35 us, whatever the rows' cache level (measured: every sample redirected to two L1-resident rows: -1.6 us; tanh removed:
// no change; twice the loads in flight: no change).  The samples of a query are spread evenly over the pyramid levels
// while the levels' rows are not (100 / 50 / 25 / 13 at cfg A): the last 32 rows of a slab receive 44 % of all reads.
// Here a workgroup serves ONE video (a contiguous share of its queries), stages the ctx2att half of the slab's last
// n_ctx rows and the value half of its last n_val rows in LDS once (as many as fit beside the offsets weight) and every
// sample whose rows lie there reads LDS (ds_read_b128, 4-8 x the rate) -- the test is wavefront-uniform (the row index
// is a scalar).  fp32, L * P == 16, result as planes (the inference token loop).
// PRE (round 5): the offsets' hidden-state part `h W_off^T` arrives precomputed (off_pre: it rides as 16 more columns in the
// h2att(h) product in front of this kernel), so neither h nor the 32 KB weight is read, and the freed LDS takes the VALUE half of
// level 2 as well (VAL_FROM = 8: samples 8 .. 15 of both halves are resident; 16 instead of 20 of a row's 32 sample reads go
// through the CU's vector-memory path, which bounds the kernel).
template <int WAVES, bool PRE = false, int VAL_FROM = 12>
__global__ void __launch_bounds__(WAVES * 64, WAVES == 12 ? 3 : 2) k_cap_attend_lds(
    const float *__restrict__ slab,         // (B, S, 2C)  [value_proj(memory) | ctx2att(value_proj(memory))]
    const int64_t *__restrict__ shapes,  // (L, 2)
    const int64_t *__restrict__ lsi,     // (L)
    const float *__restrict__ ref,       // (B, Q, L, RD) reference points scaled by the valid ratios
    const float *__restrict__ off_hs,    // (B*Q, 16)   sampling_offsets bias + hs part
    const float *__restrict__ h,         // (B*Q, C)    previous hidden state
    const float *__restrict__ w_off_h,   // (16, C)     sampling_offsets.weight[:, :C]
    const float *__restrict__ att_h,        // (B*Q, C)    h2att(h), row stride att_h_ld elements
    const float *__restrict__ alpha_w,   // (C)
    float alpha_b, int B, int S, int L, int Q, int P, int RD, int vids_per_xcd, int parts, int n_ctx, int n_val,
    int att_h_ld,
    _Float16 *__restrict__ o_hi,         // the result as the fp16 planes + row scale of gvl_gemm_f16x3_f32
    _Float16 *__restrict__ o_lo, float *__restrict__ o_scale,
    const float *__restrict__ off_pre = nullptr, int off_pre_ld = 0) {
  constexpr bool FULL = true;
  extern __shared__ float4 cap_lds[];    // [offsets weight 32 KiB (not PRE) | ctx2att half of rows S - n_ctx .. | value half of rows S - n_val ..]
  float4 *wo4 = cap_lds, *lds_ctx = cap_lds + (PRE ? 0 : kLP * kC / 4), *lds_val = lds_ctx + n_ctx * (kC / 4);
  if constexpr (!PRE)
    for (int i = threadIdx.x; i < kLP * kC / 4; i += blockDim.x) wo4[i] = reinterpret_cast<const float4 *>(w_off_h)[i];

  // workgroup -> (video, share of its queries): XCD x (= workgroup id % 8) serves videos [x vids_per_xcd, (x + 1)
  // vids_per_xcd), `parts` workgroups per video
  const int xcd = blockIdx.x & 7, jblk = blockIdx.x >> 3;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int b = xcd * vids_per_xcd + jblk / parts, part_id = jblk % parts;
  if (b >= B || jblk / parts >= vids_per_xcd) return;                  // (uniform for the workgroup)
  const float *slab_b = slab + (int64_t)b * S * (2 * kC);             // this video's slab; ld4 indexes groups of 4 elements
  const int c0 = S - n_ctx, v0 = S - n_val;                           // first slab row resident in LDS, per half
  {
    const float4 *src = reinterpret_cast<const float4 *>(slab_b);
    for (int i = threadIdx.x; i < n_ctx * (kC / 4); i += blockDim.x)
      lds_ctx[i] = src[(int64_t)(c0 + i / (kC / 4)) * (2 * kC / 4) + kC / 4 + i % (kC / 4)];
    for (int i = threadIdx.x; i < n_val * (kC / 4); i += blockDim.x)
      lds_val[i] = src[(int64_t)(v0 + i / (kC / 4)) * (2 * kC / 4) + i % (kC / 4)];
  }
  __syncthreads();
  const int per = (Q + parts - 1) / parts, q_begin = part_id * per, q_end = min(Q, q_begin + per);
  for (int q = q_begin + wave; q < q_end; q += WAVES) {
  const int64_t row = (int64_t)b * Q + q;

  const int k_own = lane >> 2;                       // the sample this lane group owns after the butterfly
  float off;
  if constexpr (PRE) {
    off = off_pre[row * off_pre_ld + min(k_own, kLP - 1)];
  } else {
    // own channels: [4*lane, 4*lane+4) and [256 + 4*lane, 256 + 4*lane + 4)
    const float4 *h4 = reinterpret_cast<const float4 *>(h + row * kC);
    const float4 ha = h4[lane], hb = h4[64 + lane];
    float part[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const float4 wa = wo4[k * (kC / 4) + lane], wb = wo4[k * (kC / 4) + 64 + lane];
      part[k] = wa.x * ha.x + wa.y * ha.y + wa.z * ha.z + wa.w * ha.w + wb.x * hb.x + wb.y * hb.y + wb.z * hb.z +
                wb.w * hb.w;
    }
    off = butterfly16(part, lane);
  }
  int roff = 0;
  float c_lo = 0.f, c_hi = 0.f, locx = 0.f;
  const int LP = FULL ? kLP : L * P;
  if (k_own < LP) {
    off += off_hs[row * LP + k_own];
    const int l = k_own / P;
    const int T = (int)shapes[2 * l + 1];
    const float *rp = ref + (row * L + l) * RD;
    if (RD == 1) locx = rp[0] + off / (float)T;                              // ms_deform_attn_for_caption.py:108-109
    else locx = rp[0] + off / (float)P * rp[1] * 0.5f;                       // :110-112
    int r;
    border_coef(locx, T, r, c_lo, c_hi);
    roff = (int)lsi[l] + r;
  }

  // ---- pass 1: attention logits from the ctx2att half ------------------------------------------------------
  const float *ah = att_h + row * (int64_t)att_h_ld;
  const float4 ta = ld4(ah, lane), tb = ld4(ah, 64 + lane);
  const float4 qa = reinterpret_cast<const float4 *>(alpha_w)[lane], qb = reinterpret_cast<const float4 *>(alpha_w)[64 + lane];
  // Samples are taken two at a time with all eight 16-byte loads requested before any arithmetic: a wavefront walks a
  // chain of 32 dependent L2 round trips otherwise (4 loads in flight), and the kernel is bound by that latency
  struct Rows { float4 l0, l1, u0, u1; };
  auto load_rows = [&](int k, int half4, float &cl, float &ch) {
    const int rr = __builtin_amdgcn_readlane(roff, 4 * k);
    cl = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, c_lo), 4 * k));
    ch = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, c_hi), 4 * k));
    const int rr1 = min(rr + 1, S - 1);             // c_hi == 0 whenever rr + 1 leaves the level (T_l == 1)
    Rows r;
    // WHERE a sample's rows live is static: the ctx2att half of levels 2, 3 (samples 8 .. 15) and the value half of level
    // 3 (samples 12 .. 15) are resident (the host launches this kernel only when they fit)
    if (half4 ? k >= 8 : k >= VAL_FROM) {
      const float4 *img = half4 ? lds_ctx : lds_val;
      const int first = half4 ? c0 : v0;
      const int i0 = (rr - first) * (kC / 4), i1 = (rr1 - first) * (kC / 4);
      r.l0 = img[i0 + lane]; r.l1 = img[i0 + 64 + lane];
      r.u0 = img[i1 + lane]; r.u1 = img[i1 + 64 + lane];
    } else {
      const int64_t r0 = (int64_t)rr * (2 * kC / 4) + half4, r1 = (int64_t)rr1 * (2 * kC / 4) + half4;
      r.l0 = ld4(slab_b, r0 + lane); r.l1 = ld4(slab_b, r0 + 64 + lane);
      r.u0 = ld4(slab_b, r1 + lane); r.u1 = ld4(slab_b, r1 + 64 + lane);
    }
    return r;
  };
  auto logit_part = [&](const Rows &r, float cl, float ch) {
    float s = 0.f;
    s = fmaf(qa.x, fast_tanh(fmaf(cl, r.l0.x, fmaf(ch, r.u0.x, ta.x))), s);
    s = fmaf(qa.y, fast_tanh(fmaf(cl, r.l0.y, fmaf(ch, r.u0.y, ta.y))), s);
    s = fmaf(qa.z, fast_tanh(fmaf(cl, r.l0.z, fmaf(ch, r.u0.z, ta.z))), s);
    s = fmaf(qa.w, fast_tanh(fmaf(cl, r.l0.w, fmaf(ch, r.u0.w, ta.w))), s);
    s = fmaf(qb.x, fast_tanh(fmaf(cl, r.l1.x, fmaf(ch, r.u1.x, tb.x))), s);
    s = fmaf(qb.y, fast_tanh(fmaf(cl, r.l1.y, fmaf(ch, r.u1.y, tb.y))), s);
    s = fmaf(qb.z, fast_tanh(fmaf(cl, r.l1.z, fmaf(ch, r.u1.z, tb.z))), s);
    s = fmaf(qb.w, fast_tanh(fmaf(cl, r.l1.w, fmaf(ch, r.u1.w, tb.w))), s);
    return s;
  };
  float e[16];
  constexpr int kBatch = 2;                                          // samples whose 4 kBatch loads are in flight together (4: measured the same)
#pragma unroll
  for (int k = 0; k < 16; k += kBatch) {
    float cl[kBatch], ch[kBatch];
    Rows rr[kBatch];
#pragma unroll
    for (int t = 0; t < kBatch; ++t) {
      e[k + t] = 0.f;
      if (k + t < LP) rr[t] = load_rows(k + t, kC / 4, cl[t], ch[t]);  // ctx2att half of the slab rows
    }
    __builtin_amdgcn_sched_barrier(0);                               // keep the loads ahead of the arithmetic
#pragma unroll
    for (int t = 0; t < kBatch; ++t)
      if (k + t < LP) e[k + t] = logit_part(rr[t], cl[t], ch[t]);
    __builtin_amdgcn_sched_barrier(0);                               // (no hoisting of the next pair's two-way loads)
  }
  float ek = butterfly16(e, lane) + alpha_b;
  if (k_own >= LP) ek = -INFINITY;
  const float m = groups_max(ek);
  const float pexp = (k_own < LP) ? __expf(ek - m) : 0.f;
  const float alpha = pexp / groups_sum(pexp);
  const float a_lo = alpha * c_lo, a_hi = alpha * c_hi;

  // ---- pass 2: weighted sum of the value half --------------------------------------------------------------
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  auto coef = [&](int k, float &cl, float &ch) {
    cl = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, a_lo), 4 * k));
    ch = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, a_hi), 4 * k));
  };
#pragma unroll
  for (int k = 0; k < 16; k += kBatch) {
    float d0, d1;
    Rows rr[kBatch];
#pragma unroll
    for (int t = 0; t < kBatch; ++t)
      if (k + t < LP) rr[t] = load_rows(k + t, 0, d0, d1);           // value half of the slab rows
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int t = 0; t < kBatch; ++t)
      if (k + t < LP) {
        float cl, ch;
        coef(k + t, cl, ch);
        fma8(cl, rr[t].l0, rr[t].l1, acc);
        fma8(ch, rr[t].u0, rr[t].u1, acc);
      }
    __builtin_amdgcn_sched_barrier(0);
  }
  {
    // the wavefront owns the whole row: its largest magnitude is one reduction away
    float m = 0.f;
#pragma unroll
    for (int c = 0; c < 8; ++c) m = fmaxf(m, fabsf(acc[c]));
#pragma unroll
    for (int o = 32; o; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    float sc, inv;
    pow2_scale(m, sc, inv);
    if (lane == 0) o_scale[row] = sc;
    half4_t h0, l0, h1, l1;
    split4_f16(make_float4(acc[0], acc[1], acc[2], acc[3]), inv, h0, l0);
    split4_f16(make_float4(acc[4], acc[5], acc[6], acc[7]), inv, h1, l1);
    // K-stage-major planes (gvl_gemm16_common.hpp: plane_off): channel k of row r at ((k >> 5) n + r) 32 + (k & 31)
    const int64_t at0 = ((int64_t)(lane >> 3) * ((int64_t)B * Q) + row) * 32 + 4 * (lane & 7), at1 = at0 + (int64_t)B * Q * 256;
    *reinterpret_cast<half4_t *>(o_hi + at0) = h0; *reinterpret_cast<half4_t *>(o_hi + at1) = h1;
    *reinterpret_cast<half4_t *>(o_lo + at0) = l0; *reinterpret_cast<half4_t *>(o_lo + at1) = l1;
  }
  }
}


// ------------------------------------------------------------------------------------------------------
// greedy-decoding epilogue: one workgroup per row: argmax (first maximal index) and log_softmax at the argmax
// ------------------------------------------------------------------------------------------------------
// exp(m - mn) for the running-max rescale; a lane that has seen no element yet carries m = -inf (and sum 0)
__device__ inline float rescale(float m, float mn) { return m == -INFINITY ? 0.f : __expf(m - mn); }

// Optional greedy-decoding bookkeeping of LSTM_DSA.py:180-190, fused (all four pointers NULL = plain argmax / LSE):
//   unfinished[row] &= (token > 0)  (step 0: = token > 0);
//   seq[row * seq_ld] = token * unfinished[row];  seq_lp[row * seq_ld] = log-prob
// ("is any row still unfinished at step t" is NOT kept here: thousands of workgroups updating one flag serialise on
//  it -- measured +10 us per step with atomicOr, +125 us with a system-scope store; it equals any(seq[:, t] != 0) and
//  is computed once after the loop)
struct GreedyBook {
  unsigned char *unfinished;
  int64_t *seq;
  float *seq_lp;
  int seq_ld, first;
};

template <typename LT>
__global__ void __launch_bounds__(256) k_row_argmax_lse(const LT *__restrict__ logits, int R, int V,
                                                        int64_t *__restrict__ idx, float *__restrict__ logp,
                                                        GreedyBook book) {
  __shared__ float s_m[4], s_s[4];
  __shared__ int s_i[4];
  const int row = blockIdx.x;
  const LT *x = logits + (int64_t)row * V;
  float m = -INFINITY, s = 0.f;
  int am = 0x7fffffff;
  auto take = [&](float v, int i) {
    if (v > m) { s = s * __expf(m - v) + 1.f; m = v; am = i; }
    else { s += __expf(v - m); }
  };
  // 16-byte loads over the 16-byte aligned body of the row (rows of V elements start at any element boundary): a
  // kernel whose only job is to stream the logits once wants few, wide loads in flight.  Thread 0 takes the leading
  // and trailing elements, in index order (ties resolve to the first maximal index).
  {
    constexpr int kPer = 16 / (int)sizeof(LT);                        // elements per 16-byte load: 4 (fp32) / 8 (bf16)
    const int head = min(V, (int)(((16 - (((uintptr_t)x) & 15)) & 15) / sizeof(LT)));
    const int nbody = (V - head) / kPer;
    const int tail0 = head + kPer * nbody;
    if (threadIdx.x == 0)
      for (int i = 0; i < head; ++i) take((float)x[i], i);
    const uint4 *x4 = reinterpret_cast<const uint4 *>(x + head);
    for (int i = threadIdx.x; i < nbody; i += blockDim.x) {
      const uint4 v = x4[i];
      const int base = head + kPer * i;
      if constexpr (sizeof(LT) == 4) {
        take(__builtin_bit_cast(float, v.x), base);
        take(__builtin_bit_cast(float, v.y), base + 1);
        take(__builtin_bit_cast(float, v.z), base + 2);
        take(__builtin_bit_cast(float, v.w), base + 3);
      } else {                                                        // bf16 -> fp32 is a 16-bit shift
        take(__builtin_bit_cast(float, v.x << 16), base);
        take(__builtin_bit_cast(float, v.x & 0xffff0000u), base + 1);
        take(__builtin_bit_cast(float, v.y << 16), base + 2);
        take(__builtin_bit_cast(float, v.y & 0xffff0000u), base + 3);
        take(__builtin_bit_cast(float, v.z << 16), base + 4);
        take(__builtin_bit_cast(float, v.z & 0xffff0000u), base + 5);
        take(__builtin_bit_cast(float, v.w << 16), base + 6);
        take(__builtin_bit_cast(float, v.w & 0xffff0000u), base + 7);
      }
    }
    if (threadIdx.x == 0)
      for (int i = tail0; i < V; ++i) take((float)x[i], i);
  }
  // wave reduction of (m, s, am)
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float m2 = __shfl_xor(m, o, 64), s2 = __shfl_xor(s, o, 64);
    const int a2 = __shfl_xor(am, o, 64);
    const float mn = fmaxf(m, m2);
    s = s * rescale(m, mn) + s2 * rescale(m2, mn);
    am = (m2 > m || (m2 == m && a2 < am)) ? a2 : am;
    m = mn;
  }
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (lane == 0) { s_m[w] = m; s_s[w] = s; s_i[w] = am; }
  __syncthreads();
  if (threadIdx.x == 0) {
    float M = s_m[0], Ssum = s_s[0];
    int A = s_i[0];
    for (int k = 1; k < 4; ++k) {
      const float mn = fmaxf(M, s_m[k]);
      Ssum = Ssum * rescale(M, mn) + s_s[k] * rescale(s_m[k], mn);
      A = (s_m[k] > M || (s_m[k] == M && s_i[k] < A)) ? s_i[k] : A;
      M = mn;
    }
    const float lp = -logf(Ssum);     // x_max - (x_max + log sum exp(x - x_max))
    idx[row] = A;
    logp[row] = lp;
    if (book.unfinished) {
      const bool unf = (book.first || book.unfinished[row]) && A > 0;
      book.unfinished[row] = unf;
      book.seq[(int64_t)row * book.seq_ld] = unf ? A : 0;
      book.seq_lp[(int64_t)row * book.seq_ld] = lp;
    }
  }
}

// ------------------------------------------------------------------------------------------------------
// backward of the unweighted sampler: gsamp (B*M, D, Q, L, P) -> grad_value (atomics), grad_loc (B,Q,M,L,P,2)
// one wavefront per (b,q,m); generic D / HxW / fp32,fp64 (the teacher-forced captioner touches few rows)
// ------------------------------------------------------------------------------------------------------
template <typename T>
__device__ inline T gfloor_(T x);
template <>
__device__ inline float gfloor_<float>(float x) { return floorf(x); }
template <>
__device__ inline double gfloor_<double>(double x) { return floor(x); }

template <typename T>
__device__ inline T pix_(T loc, int size, int pad, T &dmul) {
  if (pad == GVL_PAD_ZEROS) { dmul = (T)size; return loc * (T)size - (T)0.5; }
  T g = (T)2 * loc - (T)1;
  T x = ((g + (T)1) * (T)size - (T)1) / (T)2;
  T mx = (T)(size - 1);
  if (!(x > (T)0)) { dmul = (T)0; return (T)0; }
  if (x >= mx) { dmul = (T)0; return mx; }
  dmul = (T)size;
  return x;
}

// One workgroup per (b,q,m); its wavefronts split the D channels in slices of 64 (the teacher-forced captioner has
// few tuples -- 48 at cfg A -- but D = 1024 channels, so the parallelism must come from the channels); the partial
// grad_loc sums of the slices meet in LDS.
template <typename T>
__global__ void __launch_bounds__(1024) k_sample_bwd(const T *__restrict__ value, const int64_t *__restrict__ shapes,
                                                     const int64_t *__restrict__ lsi, const T *__restrict__ loc,
                                                     const T *__restrict__ gsamp, int B, int S, int M, int D, int L,
                                                     int Q, int P, int pad, T *__restrict__ gvalue,
                                                     T *__restrict__ gloc) {
  __shared__ T red[16][2];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const int64_t ntup = (int64_t)B * Q * M;
  const int64_t rowstride = (int64_t)M * D;
  for (int64_t tup = blockIdx.x; tup < ntup; tup += gridDim.x) {
    const int m = (int)(tup % M);
    const int64_t bq = tup / M;
    const int q = (int)(bq % Q), b = (int)(bq / Q);
    const int64_t wb = tup * L * P;
    for (int l = 0; l < L; ++l) {
      const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];
      const int64_t voff = ((int64_t)b * S + lsi[l]) * rowstride + (int64_t)m * D;
      for (int p = 0; p < P; ++p) {
        const int64_t si = wb + l * P + p;
        T dmx, dmy;
        const T w_im = pix_(loc[si * 2], W, pad, dmx), h_im = pix_(loc[si * 2 + 1], H, pad, dmy);
        const bool valid = pad == GVL_PAD_BORDER || (h_im > (T)-1 && w_im > (T)-1 && h_im < (T)H && w_im < (T)W);
        T ax = (T)0, ay = (T)0;
        if (valid) {
          const int hl = (int)gfloor_<T>(h_im), wl = (int)gfloor_<T>(w_im);
          const T lh = h_im - (T)hl, lw = w_im - (T)wl, hh = (T)1 - lh, hw = (T)1 - lw;
          const bool hl_ok = hl >= 0 && hl <= H - 1, hh_ok = hl + 1 >= 0 && hl + 1 <= H - 1;
          const bool wl_ok = wl >= 0 && wl <= W - 1, wh_ok = wl + 1 >= 0 && wl + 1 <= W - 1;
          const int64_t i00 = (int64_t)(hl * W + wl) * rowstride, i01 = (int64_t)(hl * W + wl + 1) * rowstride;
          const int64_t i10 = (int64_t)((hl + 1) * W + wl) * rowstride, i11 = (int64_t)((hl + 1) * W + wl + 1) * rowstride;
          for (int d = wave * 64 + lane; d < D; d += nw * 64) {
            const T g = gsamp[(((((int64_t)b * M + m) * D + d) * Q + q) * L + l) * P + p];
            T gh = (T)0, gw = (T)0;
            if (hl_ok && wl_ok) { const T v = value[voff + i00 + d]; gh -= hw * v; gw -= hh * v; atomicAdd(gvalue + voff + i00 + d, hh * hw * g); }
            if (hl_ok && wh_ok) { const T v = value[voff + i01 + d]; gh -= lw * v; gw += hh * v; atomicAdd(gvalue + voff + i01 + d, hh * lw * g); }
            if (hh_ok && wl_ok) { const T v = value[voff + i10 + d]; gh += hw * v; gw -= lh * v; atomicAdd(gvalue + voff + i10 + d, lh * hw * g); }
            if (hh_ok && wh_ok) { const T v = value[voff + i11 + d]; gh += lw * v; gw += lh * v; atomicAdd(gvalue + voff + i11 + d, lh * lw * g); }
            ax += dmx * gw * g;
            ay += dmy * gh * g;
          }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { ax += __shfl_xor(ax, o, 64); ay += __shfl_xor(ay, o, 64); }
        if (lane == 0) { red[wave][0] = ax; red[wave][1] = ay; }
        __syncthreads();
        if (threadIdx.x == 0) {
          T sx = (T)0, sy = (T)0;
          for (int k = 0; k < nw; ++k) { sx += red[k][0]; sy += red[k][1]; }
          gloc[si * 2] = sx;
          gloc[si * 2 + 1] = sy;
        }
        __syncthreads();
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------------
// LSTM cell pointwise part of one token step (nn.LSTM single layer, bias-free; LSTM_DSA.py:216-217,269):
//   gates = ga + gb + emb_table[it] (+ gc)   (partial pre-activations: the GEMM over the attended feature, the GEMM
//                                      over h, the pre-multiplied embedding row of the input token, and -- optional --
//                                      the token-independent hs part, so that no GEMM needs a beta = 1 C operand)
//   i,f,g,o = split(gates);  c' = sigmoid(f) c + sigmoid(i) tanh(g);  h' = sigmoid(o) tanh(c')
// one float4 of hidden units per lane
// ------------------------------------------------------------------------------------------------------
__device__ inline float sigmoidf_(float x) { return __builtin_amdgcn_rcpf(1.f + __expf(-x)); }
__device__ inline float tanhf_(float x) { return fmaf(-2.f, __builtin_amdgcn_rcpf(1.f + __expf(2.f * x)), 1.f); }

// GT: storage type of the four gate operands (fp32 / bf16 GEMM outputs); state c, h fp32
template <typename GT>
__global__ void __launch_bounds__(256) k_lstm_cell(const GT *__restrict__ ga, int lda, const GT *__restrict__ gb,
                                                   int ldb, const GT *__restrict__ emb, const int64_t *__restrict__ it,
                                                   const GT *__restrict__ gc, int ldc, const float *__restrict__ c,
                                                   int n, int H, float *__restrict__ h_out, float *__restrict__ c_out,
                                                   GT *__restrict__ h_gemm, _Float16 *__restrict__ h_hi,
                                                   _Float16 *__restrict__ h_lo, float *__restrict__ h_scale) {
  const int H4 = H >> 2;
  const int64_t total = (int64_t)n * H4;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int row = (int)(idx / H4), j = (int)(idx % H4);
    const GT *pa = ga + (int64_t)row * lda, *pb = gb + (int64_t)row * ldb, *pe = emb + it[row] * (int64_t)(4 * H);
    const GT *pc = gc ? gc + (int64_t)row * ldc : nullptr;
    float4 g4[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float4 a = ld4(pa, k * H4 + j), b = ld4(pb, k * H4 + j), e = ld4(pe, k * H4 + j);
      // (hs part + attention part) first: the same association as the reference's single W_ih GEMM over [att | hs]
      float4 x = a;
      if (pc) { const float4 cc = ld4(pc, k * H4 + j); x = make_float4(cc.x + a.x, cc.y + a.y, cc.z + a.z, cc.w + a.w); }
      g4[k] = make_float4(x.x + b.x + e.x, x.y + b.y + e.y, x.z + b.z + e.z, x.w + b.w + e.w);
    }
    const float4 cp = reinterpret_cast<const float4 *>(c)[idx];
    float4 cn, hn;
#define GVL_CELL(X) gvl_lstm_point(g4[0].X, g4[1].X, g4[2].X, g4[3].X, cp.X, cn.X, hn.X);
    GVL_CELL(x) GVL_CELL(y) GVL_CELL(z) GVL_CELL(w)
#undef GVL_CELL
    reinterpret_cast<float4 *>(c_out)[idx] = cn;
    reinterpret_cast<float4 *>(h_out)[idx] = hn;
    if (h_gemm) st4(h_gemm, idx, hn);        // h' once more in the gates' storage type: the A operand of the next GEMMs
    if (h_hi) {
      // ... or as the fp16 planes of gvl_gemm_f16x3_f32.  |h'| = |sigmoid . tanh| < 1, so the row scale is 1 for every
      // row (no overflow; an element below 2^-14 sits in fp16's subnormal range, absolute precision 2^-35 with the lo part)
      half4_t hi, lo;
      split4_f16(hn, 1.f, hi, lo);
      const int64_t at = ((int64_t)(j >> 3) * n + row) * 32 + 4 * (j & 7);       // K-stage-major planes (plane_off)
      *reinterpret_cast<half4_t *>(h_hi + at) = hi;
      *reinterpret_cast<half4_t *>(h_lo + at) = lo;
      if (j == 0) h_scale[row] = 1.f;
    }
  }
}

template <typename T>
int sample_bwd_impl(const T *value, const int64_t *shapes, const int64_t *lsi, const T *loc, const T *gsamp, int B,
                    int S, int M, int D, int L, int Q, int P, int pad, T *gvalue, T *gloc, hipStream_t st) {
  if (B < 0 || S < 0 || M <= 0 || D <= 0 || L <= 0 || Q < 0 || P <= 0 || (pad != 0 && pad != 1))
    return fail(GVL_EINVAL, "gvl_msda_sample_backward: bad arguments");
  const size_t gv_bytes = (size_t)B * S * M * D * sizeof(T);
  if (gv_bytes) {
    if (!gvalue) return fail(GVL_EINVAL, "gvl_msda_sample_backward: null pointer");
    if (int rc = gvl::zero_fill(gvalue, gv_bytes, st)) return rc;
  }
  const int64_t ntup = (int64_t)B * Q * M;
  if (ntup == 0) return 0;
  if (!value || !shapes || !lsi || !loc || !gsamp || !gloc)
    return fail(GVL_EINVAL, "gvl_msda_sample_backward: null pointer");
  int64_t blocks = ntup;
  if (blocks > 256 * 16) blocks = 256 * 16;
  int waves = (D + 63) / 64;
  if (waves > 16) waves = 16;
  return gvl::launch(GVL_PROF_SAMPLE_BWD, Q, B, "k_sample_bwd", k_sample_bwd<T>, dim3((unsigned)blocks),
                     dim3(waves * 64), 0, st, value, shapes, lsi, loc, gsamp, B, S, M, D, L, Q, P, pad, gvalue, gloc);
}


template <typename ST>
int cap_attend_impl(const char *what, const ST *slab, const int64_t *shapes, const int64_t *lsi, const float *ref,
                    const float *off_hs, const float *h, const float *w_off_h, const ST *att_h, const float *alpha_w,
                    float alpha_b, int B, int S, int C, int L, int Q, int P, int RD, int att_h_ld, ST *att_res,
                    float *dbg_alpha, float *dbg_loc, void *stream, void *o_hi = nullptr, void *o_lo = nullptr,
                    float *o_scale = nullptr, const int64_t *lsi_host = nullptr, const float *off_pre = nullptr,
                    int off_pre_ld = 0) {
  if (att_h_ld < C || (att_h_ld & 3)) return fail(GVL_EINVAL, "%s: att_h_ld must be >= C and a multiple of 4", what);
  if (C != kC || L * P > kLP || L <= 0 || P <= 0 || (RD != 1 && RD != 2) || B < 0 || Q < 0 || S <= 0)
    return fail(GVL_EINVAL, "%s: unsupported shape C=%d L=%d P=%d RD=%d (need C=512, L*P<=16)", what, C, L, P, RD);
  if ((int64_t)B * Q == 0) return 0;
  if (!slab || !shapes || !lsi || !ref || !off_hs || (!off_pre && (!h || !w_off_h)) || !att_h || !alpha_w || (!att_res && !o_hi)
      || (o_hi && (!o_lo || !o_scale)))
    return fail(GVL_EINVAL, "%s: null pointer", what);
  const int vids_per_group = (B + 7) / 8;
  if constexpr (std::is_same<ST, float>::value) {
    if (off_pre) {
      // precomputed offsets: the LDS form only (its conditions are what gvl_cap_attend_pre_applicable reports)
      const int n_ctx = lsi_host ? S - (int)lsi_host[2] : 0, n_v3 = lsi_host ? S - (int)lsi_host[3] : 0;
      if (!(o_hi && L == 4 && P == 4 && lsi_host && n_v3 >= 1 && n_ctx >= n_v3 && n_ctx + n_v3 <= 79) || off_pre_ld < kLP)
        return fail(GVL_EINVAL, "%s: the precomputed-offsets form needs L = P = 4, host level starts and a coarse end that fits the LDS", what);
      const bool val2 = 2 * n_ctx <= 79;                               // the value half of level 2 fits as well
      const int n_val = val2 ? n_ctx : n_v3;
      int parts = 32 / vids_per_group;
      parts = parts < 1 ? 1 : (parts > Q ? Q : parts);
      const size_t lds = (size_t)(n_ctx + n_val) * kC * sizeof(float);
      auto go = [&](auto kernel) -> int {
        if (int rc = gvl::ensure_lds(kernel, lds)) return rc;
        return gvl::launch(GVL_PROF_CAP_ATTEND, B * Q, B, "k_cap_attend_lds<pre>", kernel, dim3(8 * vids_per_group * parts),
                           dim3(12 * 64), lds, (hipStream_t)stream, slab, shapes, lsi, ref, off_hs, (const float *)nullptr,
                           (const float *)nullptr, att_h, alpha_w, alpha_b, B, S, L, Q, P, RD, vids_per_group, parts, n_ctx, n_val,
                           att_h_ld, (_Float16 *)o_hi, (_Float16 *)o_lo, o_scale, off_pre, off_pre_ld);
      };
      return val2 ? go(k_cap_attend_lds<12, true, 8>) : go(k_cap_attend_lds<12, true, 12>);
    }
    const char *e = gvl::env_str("GVL_CAP_LDS");
    // rows of levels 2, 3 (ctx2att half) and of level 3 (value half): known to the caller that passes host level starts
    const int n_ctx = lsi_host ? S - (int)lsi_host[2] : 0, n_val = lsi_host ? S - (int)lsi_host[3] : 0;
    if (o_hi && L == 4 && P == 4 && lsi_host && n_val >= 1 && n_ctx >= n_val && n_ctx + n_val <= 63 &&
        !(e && atoi(e) == 0)) {
      // the coarse end of each video's slab in LDS (k_cap_attend_lds): 64 half rows of 2 KB beside the 32 KB offsets weight
      int parts = 32 / vids_per_group;                                 // workgroups per video: one per CU of the video's XCD
      parts = parts < 1 ? 1 : (parts > Q ? Q : parts);
      const size_t lds = (size_t)kLP * kC * sizeof(float) + (size_t)(n_ctx + n_val) * kC * sizeof(float);
      const char *w = gvl::env_str("GVL_CAP_LDS_WAVES");
      const dim3 grid(8 * vids_per_group * parts);
      if (w && atoi(w) == 8) {
        if (int rc = gvl::ensure_lds(k_cap_attend_lds<8>, lds)) return rc;
        return gvl::launch(GVL_PROF_CAP_ATTEND, B * Q, B, "k_cap_attend_lds", k_cap_attend_lds<8>, grid, dim3(8 * 64), lds,
                           (hipStream_t)stream, slab, shapes, lsi, ref, off_hs, h, w_off_h, att_h, alpha_w, alpha_b, B, S, L,
                           Q, P, RD, vids_per_group, parts, n_ctx, n_val, att_h_ld, (_Float16 *)o_hi, (_Float16 *)o_lo, o_scale,
                           (const float *)nullptr, 0);
      }
      if (int rc = gvl::ensure_lds(k_cap_attend_lds<12>, lds)) return rc;
      return gvl::launch(GVL_PROF_CAP_ATTEND, B * Q, B, "k_cap_attend_lds", k_cap_attend_lds<12>, grid, dim3(12 * 64), lds,
                         (hipStream_t)stream, slab, shapes, lsi, ref, off_hs, h, w_off_h, att_h, alpha_w, alpha_b, B, S, L, Q,
                         P, RD, vids_per_group, parts, n_ctx, n_val, att_h_ld, (_Float16 *)o_hi, (_Float16 *)o_lo, o_scale,
                           (const float *)nullptr, 0);
    }
  }
  const int rows_per_group = vids_per_group * Q;
  int blocks_per_group = (rows_per_group + kWaves - 1) / kWaves;
  if (blocks_per_group > 96) blocks_per_group = 96;                    // 3 resident workgroups on each of an XCD's 32 CUs
  auto kern = L * P == kLP ? k_cap_attend<ST, true> : k_cap_attend<ST, false>;
  return gvl::launch(GVL_PROF_CAP_ATTEND, B * Q, B, "k_cap_attend", kern, dim3(8 * blocks_per_group),
                     dim3(kWaves * 64), 0, (hipStream_t)stream, slab, shapes, lsi, ref, off_hs, h, w_off_h, att_h,
                     alpha_w, alpha_b, B, S, L, Q, P, RD, rows_per_group, att_h_ld, att_res, dbg_alpha, dbg_loc,
                     (_Float16 *)o_hi, (_Float16 *)o_lo, o_scale);
}

template <typename GT>
int lstm_cell_impl(const char *what, const GT *gates_a, int lda, const GT *gates_b, int ldb, const GT *emb_gates,
                   const int64_t *it, const GT *gates_c, int ldc, const float *c, int n, int H, float *h_out,
                   float *c_out, GT *h_gemm, void *stream, void *h_hi = nullptr, void *h_lo = nullptr,
                   float *h_scale = nullptr) {
  if (h_hi && (!h_lo || !h_scale)) return fail(GVL_EINVAL, "%s: null pointer", what);
  if (h_hi && (H & 31)) return fail(GVL_EINVAL, "%s: operand planes need H %% 32 == 0 (got %d)", what, H);
  if (n < 0 || H <= 0 || (H & 3) || lda < 4 * H || ldb < 4 * H || (lda & 3) || (ldb & 3) ||
      (gates_c && (ldc < 4 * H || (ldc & 3))))
    return fail(GVL_EINVAL, "%s: bad sizes n=%d H=%d lda=%d ldb=%d", what, n, H, lda, ldb);
  if (n == 0) return 0;
  if (!gates_a || !gates_b || !emb_gates || !it || !c || !h_out || !c_out) return fail(GVL_EINVAL, "%s: null pointer", what);
  int64_t blocks = ((int64_t)n * (H / 4) + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  return gvl::launch(GVL_PROF_LSTM_CELL, n, H, "k_lstm_cell", k_lstm_cell<GT>, dim3((unsigned)blocks), dim3(256), 0,
                     (hipStream_t)stream, gates_a, lda, gates_b, ldb, emb_gates, it, gates_c, ldc, c, n, H, h_out, c_out,
                     h_gemm, (_Float16 *)h_hi, (_Float16 *)h_lo, h_scale);
}

template <typename LT>
int greedy_impl(const char *what, const LT *logits, int R, int V, int64_t *token, float *logp, const GreedyBook &book,
                void *stream) {
  if (R < 0 || V <= 0 || (book.unfinished && book.seq_ld <= 0)) return fail(GVL_EINVAL, "%s: bad sizes", what);
  if (R == 0) return 0;
  if (!logits || !token || !logp || (book.unfinished && (!book.seq || !book.seq_lp)))
    return fail(GVL_EINVAL, "%s: null pointer", what);
  return gvl::launch(GVL_PROF_ROW_ARGMAX, R, V, "k_row_argmax_lse", k_row_argmax_lse<LT>, dim3(R), dim3(256), 0,
                     (hipStream_t)stream, logits, R, V, token, logp, book);
}

}  // namespace

extern "C" {

int gvl_cap_attend_f32(const float *slab, const int64_t *shapes, const int64_t *lsi, const float *ref,
                       const float *off_hs, const float *h, const float *w_off_h, const float *att_h,
                       const float *alpha_w, float alpha_b, int B, int S, int C, int L, int Q, int P, int RD,
                       int att_h_ld, float *att_res, float *dbg_alpha, float *dbg_loc, void *stream) {
  return cap_attend_impl<float>("gvl_cap_attend_f32", slab, shapes, lsi, ref, off_hs, h, w_off_h, att_h, alpha_w,
                                alpha_b, B, S, C, L, Q, P, RD, att_h_ld, att_res, dbg_alpha, dbg_loc, stream);
}
int gvl_cap_attend_bf16(const uint16_t *slab, const int64_t *shapes, const int64_t *lsi, const float *ref,
                        const float *off_hs, const float *h, const float *w_off_h, const uint16_t *att_h,
                        const float *alpha_w, float alpha_b, int B, int S, int C, int L, int Q, int P, int RD,
                        int att_h_ld, uint16_t *att_res, float *dbg_alpha, float *dbg_loc, void *stream) {
  return cap_attend_impl<bf16_t>("gvl_cap_attend_bf16", (const bf16_t *)slab, shapes, lsi, ref, off_hs, h, w_off_h,
                                 (const bf16_t *)att_h, alpha_w, alpha_b, B, S, C, L, Q, P, RD, att_h_ld,
                                 (bf16_t *)att_res, dbg_alpha, dbg_loc, stream);
}

int gvl_cap_attend_split_f32(const float *slab, const int64_t *shapes, const int64_t *lsi, const float *ref,
                             const float *off_hs, const float *h, const float *w_off_h, const float *att_h,
                             const float *alpha_w, float alpha_b, int B, int S, int C, int L, int Q, int P, int RD,
                             int att_h_ld, void *att_hi, void *att_lo, float *att_scale, void *stream) {
  if (!att_hi) return fail(GVL_EINVAL, "gvl_cap_attend_split_f32: null pointer");
  return cap_attend_impl<float>("gvl_cap_attend_split_f32", slab, shapes, lsi, ref, off_hs, h, w_off_h, att_h, alpha_w,
                                alpha_b, B, S, C, L, Q, P, RD, att_h_ld, (float *)nullptr, nullptr, nullptr, stream, att_hi,
                                att_lo, att_scale);
}

int gvl_cap_attend_split_levels_f32(const float *slab, const int64_t *shapes, const int64_t *lsi, const float *ref,
                                    const float *off_hs, const float *h, const float *w_off_h, const float *att_h,
                                    const float *alpha_w, float alpha_b, int B, int S, int C, int L, int Q, int P, int RD,
                                    int att_h_ld, const int64_t *lsi_host, void *att_hi, void *att_lo, float *att_scale,
                                    void *stream) {
  if (!att_hi) return fail(GVL_EINVAL, "gvl_cap_attend_split_levels_f32: null pointer");
  return cap_attend_impl<float>("gvl_cap_attend_split_levels_f32", slab, shapes, lsi, ref, off_hs, h, w_off_h, att_h,
                                alpha_w, alpha_b, B, S, C, L, Q, P, RD, att_h_ld, (float *)nullptr, nullptr, nullptr, stream,
                                att_hi, att_lo, att_scale, lsi_host);
}

int gvl_cap_attend_pre_applicable(int S, int L, int P, const int64_t *lsi_host) {
  if (!lsi_host || L != 4 || P != 4 || S <= 0) return 0;
  const int n_ctx = S - (int)lsi_host[2], n_v3 = S - (int)lsi_host[3];
  return n_v3 >= 1 && n_ctx >= n_v3 && n_ctx + n_v3 <= 79;
}

int gvl_cap_attend_pre_f32(const float *slab, const int64_t *shapes, const int64_t *lsi, const float *ref, const float *off_hs,
                           const float *off_pre, int off_pre_ld, const float *att_h, const float *alpha_w, float alpha_b, int B,
                           int S, int C, int L, int Q, int P, int RD, int att_h_ld, const int64_t *lsi_host, void *att_hi,
                           void *att_lo, float *att_scale, void *stream) {
  if (!att_hi || !off_pre) return fail(GVL_EINVAL, "gvl_cap_attend_pre_f32: null pointer");
  return cap_attend_impl<float>("gvl_cap_attend_pre_f32", slab, shapes, lsi, ref, off_hs, nullptr, nullptr, att_h, alpha_w,
                                alpha_b, B, S, C, L, Q, P, RD, att_h_ld, (float *)nullptr, nullptr, nullptr, stream, att_hi, att_lo,
                                att_scale, lsi_host, off_pre, off_pre_ld);
}

int gvl_lstm_cell_split_f32(const float *gates_a, int lda, const float *gates_b, int ldb, const float *emb_gates,
                            const int64_t *it, const float *gates_c, int ldc, const float *c, int n, int H, float *h_out,
                            float *c_out, void *h_hi, void *h_lo, float *h_scale, void *stream) {
  if (!h_hi) return fail(GVL_EINVAL, "gvl_lstm_cell_split_f32: null pointer");
  return lstm_cell_impl<float>("gvl_lstm_cell_split_f32", gates_a, lda, gates_b, ldb, emb_gates, it, gates_c, ldc, c, n, H,
                               h_out, c_out, (float *)nullptr, stream, h_hi, h_lo, h_scale);
}

int gvl_lstm_cell_f32(const float *gates_a, int lda, const float *gates_b, int ldb, const float *emb_gates,
                      const int64_t *it, const float *gates_c, int ldc, const float *c, int n, int H, float *h_out,
                      float *c_out, void *stream) {
  return lstm_cell_impl<float>("gvl_lstm_cell_f32", gates_a, lda, gates_b, ldb, emb_gates, it, gates_c, ldc, c, n, H,
                               h_out, c_out, (float *)nullptr, stream);
}
int gvl_lstm_cell_bf16(const uint16_t *gates_a, int lda, const uint16_t *gates_b, int ldb, const uint16_t *emb_gates,
                       const int64_t *it, const uint16_t *gates_c, int ldc, const float *c, int n, int H, float *h_out,
                       float *c_out, uint16_t *h_bf16, void *stream) {
  return lstm_cell_impl<bf16_t>("gvl_lstm_cell_bf16", (const bf16_t *)gates_a, lda, (const bf16_t *)gates_b, ldb,
                                (const bf16_t *)emb_gates, it, (const bf16_t *)gates_c, ldc, c, n, H, h_out, c_out,
                                (bf16_t *)h_bf16, stream);
}

int gvl_row_argmax_lse_f32(const float *logits, int R, int V, int64_t *idx, float *logp, void *stream) {
  const GreedyBook none = {nullptr, nullptr, nullptr, 0, 0};
  return greedy_impl<float>("gvl_row_argmax_lse_f32", logits, R, V, idx, logp, none, stream);
}
int gvl_row_argmax_lse_bf16(const uint16_t *logits, int R, int V, int64_t *idx, float *logp, void *stream) {
  const GreedyBook none = {nullptr, nullptr, nullptr, 0, 0};
  return greedy_impl<bf16_t>("gvl_row_argmax_lse_bf16", (const bf16_t *)logits, R, V, idx, logp, none, stream);
}

int gvl_greedy_step_f32(const float *logits, int R, int V, int first_step, int64_t *token, float *logp,
                        unsigned char *unfinished, int64_t *seq_col, float *seq_lp_col, int seq_ld, void *stream) {
  if (!unfinished) return fail(GVL_EINVAL, "gvl_greedy_step_f32: null pointer");
  const GreedyBook book = {unfinished, seq_col, seq_lp_col, seq_ld, first_step != 0};
  return greedy_impl<float>("gvl_greedy_step_f32", logits, R, V, token, logp, book, stream);
}
int gvl_greedy_step_bf16(const uint16_t *logits, int R, int V, int first_step, int64_t *token, float *logp,
                         unsigned char *unfinished, int64_t *seq_col, float *seq_lp_col, int seq_ld, void *stream) {
  if (!unfinished) return fail(GVL_EINVAL, "gvl_greedy_step_bf16: null pointer");
  const GreedyBook book = {unfinished, seq_col, seq_lp_col, seq_ld, first_step != 0};
  return greedy_impl<bf16_t>("gvl_greedy_step_bf16", (const bf16_t *)logits, R, V, token, logp, book, stream);
}

int gvl_msda_sample_backward_f32(const float *value, const int64_t *shapes, const int64_t *lsi, const float *loc,
                                 const float *grad_sample, int B, int S, int M, int D, int L, int Q, int P,
                                 int pad_mode, float *grad_value, float *grad_loc, void *stream) {
  return sample_bwd_impl<float>(value, shapes, lsi, loc, grad_sample, B, S, M, D, L, Q, P, pad_mode, grad_value,
                                grad_loc, (hipStream_t)stream);
}
int gvl_msda_sample_backward_f64(const double *value, const int64_t *shapes, const int64_t *lsi, const double *loc,
                                 const double *grad_sample, int B, int S, int M, int D, int L, int Q, int P,
                                 int pad_mode, double *grad_value, double *grad_loc, void *stream) {
  return sample_bwd_impl<double>(value, shapes, lsi, loc, grad_sample, B, S, M, D, L, Q, P, pad_mode, grad_value,
                                 grad_loc, (hipStream_t)stream);
}

}  // extern "C"
