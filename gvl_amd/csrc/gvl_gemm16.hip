// gvl_gemm16.hip -- fp32 products of the captioner's token loop on the fp16 matrix cores, at fp32 accuracy.
//
// Reference: the three nn.Linear of one LSTM-DSA token step that dominate the eval forward -- `self.logit(output)`
// (pdvc/CaptioningHead/LSTM_DSA.py:121,165: 4800 x 512 x 8518), `h2att(h)` + the recurrent half of `nn.LSTM`
// (:247,269: 4800 x 512 x 2560 as one product) and the attention half of the LSTM input (:267-269: 4800 x 512 x 2048).
//
// Why.  gfx950 has no xf32 and its fp32 MFMA runs at the VECTOR rate (157 TFLOP/s); the fp16 / bf16 MFMA is 16 x that.
// An fp32 value is split EXACTLY (to 22+ bits) into two fp16 numbers -- x = s (hi + 2^-11 lo), s a power of two per
// row so that |hi| < 2 (no fp16 overflow, whatever the magnitude of the row), hi = fp16(x / s), lo = fp16(2^11 (x / s - hi))
// (the 2^11 keeps the residual in fp16's normal range) -- and the product of two split numbers is three fp16 MFMAs
// with fp32 accumulation:
//
//      x . w  =  sx sw [ hi.hi  +  2^-11 (hi.lo + lo.hi) ]  +  O(2^-22 |x||w|)        (lo.lo is dropped)
//
// fp16 x fp16 products are exact in the fp32 accumulator; the two cross terms go to their OWN accumulator and are
// folded in once at the end, so their low bits are not lost against the leading sum.  Measured against an fp64 product
// on the shapes above (tools/split_gemm_probe2.py, tests/test_gpu_gemm16.py): rms error 1.9e-7 against the fp32 library
// GEMM's 4.6e-7 (a k-ordered fp32 chain rounds 512 times, this one 32 + 64 times per output) -- the result is MORE
// accurate than the fp32 path it replaces, at 3/16 of the matrix-core time.  Domain: the scale is per ROW, so an element
// more than 2^-35 below its row's largest is not represented: |error| <= 2^-21 sum |a||b| + K 2^-33 max|a| max|b|.  (Three bf16 parts would need six products: bf16 carries 8 bits per part, fp16 11.)
//
// k_split_rows:  one wavefront per row -> hi / lo planes (fp16, row-major like the input) + the row scale.
// k_gemm_f16x3:  out (R, N) = A (R, K) . B (N, K)^T [+ bias], both operands as planes.
//   workgroup = 4 wavefronts (2 x 2) on a 128 x BN tile (BN = 128 | 64), wavefront tile 64 x BN/2 in 32 x 32 MFMA tiles
//   (v_mfma_f32_32x32x16_f16); K in stages of 32 through a double-buffered LDS image (64-byte rows per plane, 16-byte
//   chunks XOR-swizzled by (row >> 2) & 3: the ds_read_b128 of an MFMA operand -- lane l reads row l & 31, chunk
//   2 s + (l >> 5) -- is then conflict-free); the next stage's global loads are issued before the MFMAs of the current
//   one and written to the other buffer after them: one barrier per stage, two workgroups per CU cover each other's
//   barriers.  Per stage and wavefront: 16 ds_read_b128 feed 24 MFMAs (4 operand fragments serve 3 products).
//   Tiles are walked in groups of 8 row tiles x 8 column tiles per XCD (workgroup id -> XCD is id % 8): the 64
//   workgroups resident on one XCD share 2 MB + 2 MB of operand planes in its 4 MB L2.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "gvl_common.hpp"
#include "gvl_msda.h"

namespace {

using gvl::fail;

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef float f16acc __attribute__((ext_vector_type(16)));

constexpr float kLoScale = 2048.f, kLoInv = 1.f / 2048.f;

// ---------------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_split_rows(const float *__restrict__ x, int R, int K, _Float16 *__restrict__ hi,
                                                    _Float16 *__restrict__ lo, float *__restrict__ scale) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= R) return;
  const float4 *xr = reinterpret_cast<const float4 *>(x + (int64_t)row * K);
  const int n4 = K >> 2;
  float m = 0.f;
  for (int i = lane; i < n4; i += 64) {
    const float4 v = xr[i];
    m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
  }
#pragma unroll
  for (int o = 32; o; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  // s = 2^floor(log2 m) from the exponent field; 1 / s exactly representable for exponents 1 .. 253
  int e = (int)((__float_as_uint(m) >> 23) & 0xffu);
  e = min(max(e, 1), 253);
  const float s = __uint_as_float((uint32_t)e << 23), inv = __uint_as_float((uint32_t)(254 - e) << 23);
  if (lane == 0) scale[row] = s;
  h4 *hr = reinterpret_cast<h4 *>(hi + (int64_t)row * K), *lr = reinterpret_cast<h4 *>(lo + (int64_t)row * K);
  for (int i = lane; i < n4; i += 64) {
    const float4 v = xr[i];                                            // second read: L1 / L2
    const float a[4] = {v.x * inv, v.y * inv, v.z * inv, v.w * inv};
    h4 h, l;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      h[c] = (_Float16)a[c];
      l[c] = (_Float16)((a[c] - (float)h[c]) * kLoScale);
    }
    hr[i] = h;
    lr[i] = l;
  }
}

// ---------------------------------------------------------------------------------------------------------------------
constexpr int kBM = 128, kBK = 32, kGroupM = 8;

__device__ __forceinline__ int lds_slot(int row, int chunk) { return row * 4 + (chunk ^ ((row >> 2) & 3)); }

// What happens to the 128 x BN tile D = A . B^T once it is complete:
//   kStore       out[row][col] = D            (A = activations, B = weight; bias per column)
//   kArgmax      A = the vocabulary layer's weight, B = the hidden states: per (token row = column of D, 64 vocabulary
//                entries = the wavefront's rows of D) the maximum, its index and sum exp(v - max) -- the logits are never
//                written.  All 32 vocabulary entries of one MFMA tile column sit in ONE lane's registers (and the lane
//                32 further): the reduction is register-local plus one cross-lane step.
enum { kStore = 0, kArgmax = 2 };
// (a transposed store -- weight on the row side, one 16-byte store per 4 values of a lane -- was measured: 3 us SLOWER
//  than the 4-byte stores of 128-byte row segments on the 4800 x 2560 / 2048 outputs)

template <int BN, int EPI>
__global__ void __launch_bounds__(256, 2)
    k_gemm_f16x3(const _Float16 *__restrict__ Ah, const _Float16 *__restrict__ Al, const float *__restrict__ As,
                 const _Float16 *__restrict__ Bh, const _Float16 *__restrict__ Bl, const float *__restrict__ Bs,
                 const float *__restrict__ bias, int R, int N, int K, float *__restrict__ out, int64_t ldo, int tiles_m,
                 int tiles_n, int dbg) {
  constexpr int NJ = BN / 64;                                         // 32-column MFMA tiles per wavefront
  constexpr int NBR = BN / 64;                                        // B rows staged per thread and plane
  __shared__ uint4 sA[2][2][kBM * 4];                                 // [stage][plane hi | lo][row * 4 + swizzled chunk]
  __shared__ uint4 sB[2][2][BN * 4];

  // tile of this workgroup: XCD x walks the contiguous range [x per, (x + 1) per) of the grouped tile order
  const int total = tiles_m * tiles_n, per = (total + 7) >> 3;
  const int t = (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);
  if ((int)(blockIdx.x >> 3) >= per || t >= total) return;
  const int gsz_full = kGroupM * tiles_n, g = t / gsz_full, first_m = g * kGroupM;
  const int gm = min(tiles_m - first_m, kGroupM), in_g = t - g * gsz_full;
  const int m0 = (first_m + in_g % gm) * kBM, n0 = (in_g / gm) * BN;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = (wave & 1) * 64, wn = (wave >> 1) * (BN / 2);

  // staging map: thread -> 16-byte chunk c of row r (and r + 64) of every plane
  const int sc = tid & 3, sr = tid >> 2;
  const int64_t a_off0 = (int64_t)min(m0 + sr, R - 1) * K + sc * 8, a_off1 = (int64_t)min(m0 + sr + 64, R - 1) * K + sc * 8;
  const int64_t b_off0 = (int64_t)min(n0 + sr, N - 1) * K + sc * 8, b_off1 = (int64_t)min(n0 + sr + 64, N - 1) * K + sc * 8;

  // one K stage's share of a thread, in plain registers (named, not an array: a loop-carried array goes to scratch)
  uint4 a_h0, a_h1, a_l0, a_l1, b_h0, b_h1 = {}, b_l0, b_l1 = {};
#define GVL_FETCH(k0)                                                        \
  a_h0 = *reinterpret_cast<const uint4 *>(Ah + a_off0 + (k0));               \
  a_h1 = *reinterpret_cast<const uint4 *>(Ah + a_off1 + (k0));               \
  a_l0 = *reinterpret_cast<const uint4 *>(Al + a_off0 + (k0));               \
  a_l1 = *reinterpret_cast<const uint4 *>(Al + a_off1 + (k0));               \
  b_h0 = *reinterpret_cast<const uint4 *>(Bh + b_off0 + (k0));               \
  b_l0 = *reinterpret_cast<const uint4 *>(Bl + b_off0 + (k0));               \
  if (NBR == 2) {                                                            \
    b_h1 = *reinterpret_cast<const uint4 *>(Bh + b_off1 + (k0));             \
    b_l1 = *reinterpret_cast<const uint4 *>(Bl + b_off1 + (k0));             \
  }
#define GVL_STASH(buf)                                                       \
  sA[buf][0][slot0] = a_h0; sA[buf][0][slot1] = a_h1;                        \
  sA[buf][1][slot0] = a_l0; sA[buf][1][slot1] = a_l1;                        \
  sB[buf][0][slot0] = b_h0; sB[buf][1][slot0] = b_l0;                        \
  if (NBR == 2) { sB[buf][0][slot1] = b_h1; sB[buf][1][slot1] = b_l1; }
  const int slot0 = lds_slot(sr, sc), slot1 = lds_slot(sr + 64, sc);

  f16acc acc_m[2][NJ], acc_x[2][NJ];                                  // leading sum | cross terms
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) { acc_m[i][j][r] = 0.f; acc_x[i][j][r] = 0.f; }

  const int frow = lane & 31, fh = lane >> 5;
  // operand fragment slots of this lane: row wm + 32 i + frow (A), wn + 32 j + frow (B); chunk 2 s + fh
  int fa[2][2], fb[NJ][2];
#pragma unroll
  for (int s = 0; s < 2; ++s) {
#pragma unroll
    for (int i = 0; i < 2; ++i) fa[i][s] = lds_slot(wm + 32 * i + frow, 2 * s + fh);
#pragma unroll
    for (int j = 0; j < NJ; ++j) fb[j][s] = lds_slot(wn + 32 * j + frow, 2 * s + fh);
  }
  auto compute = [&](int buf) {
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      h8 f_ah[2], f_al[2], f_bh[NJ], f_bl[NJ];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        f_ah[i] = *reinterpret_cast<const h8 *>(&sA[buf][0][fa[i][s]]);
        f_al[i] = *reinterpret_cast<const h8 *>(&sA[buf][1][fa[i][s]]);
      }
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        f_bh[j] = *reinterpret_cast<const h8 *>(&sB[buf][0][fb[j][s]]);
        f_bl[j] = *reinterpret_cast<const h8 *>(&sB[buf][1][fb[j][s]]);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          acc_m[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f_ah[i], f_bh[j], acc_m[i][j], 0, 0, 0);
          acc_x[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f_ah[i], f_bl[j], acc_x[i][j], 0, 0, 0);
          acc_x[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f_al[i], f_bh[j], acc_x[i][j], 0, 0, 0);
        }
    }
  };

  const int KT = K / kBK;
  GVL_FETCH(0)
  GVL_STASH(0)
  __syncthreads();
  for (int kt = 0; kt + 1 < KT; ++kt) {
    const int buf = kt & 1;
    if (!(dbg & 2)) { GVL_FETCH((kt + 1) * kBK) }   // in flight under this stage's MFMAs
    compute(buf);
    GVL_STASH(buf ^ 1)                          // last read one barrier ago
    __syncthreads();
  }
  compute((KT - 1) & 1);
#undef GVL_FETCH
#undef GVL_STASH

  // C/D map of the 32 x 32 MFMA: column = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
  // branch-free optional bias: without one every read goes to As[0] (a finite power of two) and is multiplied by 0
  const float *bias_p = bias ? bias : As;
  const float bias_on = bias ? 1.f : 0.f;
  const int bias_ix = bias ? 0x7fffffff : 0;
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int col = n0 + wn + 32 * j + frow;
    const bool col_ok = col < N;
    const float cs = col_ok ? Bs[col] : 0.f;
    if constexpr (EPI == kStore) {
      const float cb = bias_on * bias_p[min(min(col, N - 1), bias_ix)];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = m0 + wm + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * fh;
          if (col_ok && row < R && (!(dbg & 1) || acc_m[i][j][r] == 12345.f))
            out[(int64_t)row * ldo + col] = (acc_m[i][j][r] + acc_x[i][j][r] * kLoInv) * (As[row] * cs) + cb;
        }
    } else {
      // two passes over the lane's 32 values of this column (recomputed, not kept: registers): maximum, then the sum
      float best = -INFINITY, sum = 0.f;
      int arg = 0x7fffffff;
#pragma unroll
      for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int row = m0 + wm + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * fh, rc = min(row, R - 1);
            const float v = (acc_m[i][j][r] + acc_x[i][j][r] * kLoInv) * (As[rc] * cs) + bias_on * bias_p[min(rc, bias_ix)];
            if (pass == 0) {
              if (row < R && v > best) { best = v; arg = row; }            // rows ascend: the first maximum is kept
            } else if (row < R) {
              sum += __expf(v - best);
            }
            if ((r & 3) == 3) __builtin_amdgcn_sched_barrier(0);           // four values at a time: more in flight spill
          }
      }
      // the lane 32 further holds the other half of this column's 64 rows
      const float b2 = __shfl_xor(best, 32), s2 = __shfl_xor(sum, 32);
      const int a2 = __shfl_xor(arg, 32);
      const float bn = fmaxf(best, b2);
      if (bn > -INFINITY) sum = sum * __expf(best - bn) + s2 * __expf(b2 - bn);        // exp(-inf) = 0 for an empty half
      arg = (b2 > best || (b2 == best && a2 < arg)) ? a2 : arg;
      if (fh == 0 && col_ok)
        reinterpret_cast<float4 *>(out)[(int64_t)((m0 + wm) >> 6) * N + col] = make_float4(bn, sum, __int_as_float(arg), 0.f);
    }
    __builtin_amdgcn_sched_barrier(0);            // one tile column at a time: interleaving them spills
  }
}

// partials (chunks, R) of {max, sum exp(v - max), index, -} -> per row argmax and log-softmax at the argmax, plus the
// bookkeeping of one greedy step (gvl_cap.hip: k_row_argmax_lse has the same tail).  Block = 32 rows x 8 chunk groups.
struct GreedyBook {
  unsigned char *unfinished;
  int64_t *seq;
  float *seq_lp;
  int seq_ld, first;
};

__global__ void __launch_bounds__(256) k_greedy_from_partials(const float4 *__restrict__ part, int R, int chunks,
                                                              int64_t *__restrict__ idx, float *__restrict__ logp,
                                                              GreedyBook book) {
  __shared__ float s_m[8][32], s_s[8][32];
  __shared__ int s_a[8][32];
  const int rr = threadIdx.x & 31, g = threadIdx.x >> 5, row = blockIdx.x * 32 + rr;
  float m = -INFINITY, s = 0.f;
  int a = 0x7fffffff;
  if (row < R)
    for (int c = g; c < chunks; c += 8) {
      const float4 p = part[(int64_t)c * R + row];
      const int pa = __float_as_int(p.z);
      const float mn = fmaxf(m, p.x);
      if (mn > -INFINITY) s = s * __expf(m - mn) + p.y * __expf(p.x - mn);
      a = (p.x > m || (p.x == m && pa < a)) ? pa : a;
      m = mn;
    }
  s_m[g][rr] = m; s_s[g][rr] = s; s_a[g][rr] = a;
  __syncthreads();
  if (g == 0 && row < R) {
#pragma unroll
    for (int k = 1; k < 8; ++k) {
      const float m2 = s_m[k][rr], s2 = s_s[k][rr];
      const int a2 = s_a[k][rr];
      const float mn = fmaxf(m, m2);
      if (mn > -INFINITY) s = s * __expf(m - mn) + s2 * __expf(m2 - mn);
      a = (m2 > m || (m2 == m && a2 < a)) ? a2 : a;
      m = mn;
    }
    const float lp = -logf(s);
    idx[row] = a;
    logp[row] = lp;
    if (book.unfinished) {
      const bool unf = (book.first || book.unfinished[row]) && a > 0;
      book.unfinished[row] = unf;
      book.seq[(int64_t)row * book.seq_ld] = unf ? a : 0;
      book.seq_lp[(int64_t)row * book.seq_ld] = lp;
    }
  }
}

int check_operands(const char *what, const void *a_hi, const void *a_lo, const float *a_scale, int R, const void *b_hi,
                   const void *b_lo, const float *b_scale, int N, int K) {
  if (R < 0 || N <= 0 || K <= 0 || (K % kBK))
    return fail(GVL_EINVAL, "%s: needs K %% 32 == 0 (got R=%d N=%d K=%d)", what, R, N, K);
  if (R == 0) return 0;
  if (!a_hi || !a_lo || !a_scale || !b_hi || !b_lo || !b_scale) return fail(GVL_EINVAL, "%s: null pointer", what);
  if (((uintptr_t)a_hi | (uintptr_t)a_lo | (uintptr_t)b_hi | (uintptr_t)b_lo) & 15)
    return fail(GVL_EINVAL, "%s: operand planes must be 16-byte aligned", what);
  return 0;
}

int debug_flags() {
  const char *e = getenv("GVL_GEMM16_DEBUG");
  return e ? atoi(e) : 0;
}

}  // namespace

extern "C" int gvl_split_rows_f16(const float *x, int R, int K, void *hi, void *lo, float *scale, void *stream) {
  if (R < 0 || K <= 0 || (K & 3)) return fail(GVL_EINVAL, "gvl_split_rows_f16: needs K %% 4 == 0 (got R=%d K=%d)", R, K);
  if (R == 0) return 0;
  if (!x || !hi || !lo || !scale) return fail(GVL_EINVAL, "gvl_split_rows_f16: null pointer");
  if (((uintptr_t)x & 15) || ((uintptr_t)hi & 7) || ((uintptr_t)lo & 7))
    return fail(GVL_EINVAL, "gvl_split_rows_f16: x must be 16-byte, hi / lo 8-byte aligned");
  return gvl::launch(GVL_PROF_SPLIT, R, K, "k_split_rows", k_split_rows, dim3((R + 3) / 4), dim3(256), 0,
                     (hipStream_t)stream, x, R, K, (_Float16 *)hi, (_Float16 *)lo, scale);
}

extern "C" int gvl_gemm_f16x3_f32(const void *a_hi, const void *a_lo, const float *a_scale, int R, const void *b_hi,
                                  const void *b_lo, const float *b_scale, int N, int K, const float *bias, float *out,
                                  int64_t ldo, void *stream) {
  if (int rc = check_operands("gvl_gemm_f16x3_f32", a_hi, a_lo, a_scale, R, b_hi, b_lo, b_scale, N, K)) return rc;
  if (ldo < N) return fail(GVL_EINVAL, "gvl_gemm_f16x3_f32: ldo < N");
  if (R == 0) return 0;
  if (!out) return fail(GVL_EINVAL, "gvl_gemm_f16x3_f32: null pointer");
  const _Float16 *ah = (const _Float16 *)a_hi, *al = (const _Float16 *)a_lo, *bh = (const _Float16 *)b_hi,
                 *bl = (const _Float16 *)b_lo;
  const int dbg = debug_flags();
  // wide 128-column tiles when they fill the chip for several rounds, narrow ones otherwise
  const int tiles_m = (R + kBM - 1) / kBM;
  const bool wide = (int64_t)tiles_m * ((N + 127) / 128) >= 1536;
  const int bn = wide ? 128 : 64, tiles_n = (N + bn - 1) / bn;
  const dim3 grid((tiles_m * tiles_n + 7) / 8 * 8);
  auto kern = wide ? k_gemm_f16x3<128, kStore> : k_gemm_f16x3<64, kStore>;
  return gvl::launch(GVL_PROF_GEMM16, R, N, "k_gemm_f16x3", kern, grid, dim3(256), 0, (hipStream_t)stream, ah, al,
                     a_scale, bh, bl, b_scale, bias, R, N, K, out, ldo, tiles_m, tiles_n, dbg);
}

extern "C" int gvl_gemm_f16x3_argmax_chunks(int V) { return V > 0 ? (V + kBM - 1) / kBM * 2 : 0; }

extern "C" int gvl_gemm_f16x3_argmax_f32(const void *x_hi, const void *x_lo, const float *x_scale, int R, const void *w_hi,
                                         const void *w_lo, const float *w_scale, int V, int K, const float *bias,
                                         float *partials, void *stream) {
  if (int rc = check_operands("gvl_gemm_f16x3_argmax_f32", x_hi, x_lo, x_scale, R, w_hi, w_lo, w_scale, V, K)) return rc;
  if (R == 0) return 0;
  if (!partials || ((uintptr_t)partials & 15)) return fail(GVL_EINVAL, "gvl_gemm_f16x3_argmax_f32: partials null / unaligned");
  const int tiles_m = (V + kBM - 1) / kBM, tiles_n = (R + 127) / 128;
  const dim3 grid((tiles_m * tiles_n + 7) / 8 * 8);
  return gvl::launch(GVL_PROF_GEMM16, R, V, "k_gemm_f16x3<argmax>", k_gemm_f16x3<128, kArgmax>, grid, dim3(256), 0,
                     (hipStream_t)stream, (const _Float16 *)w_hi, (const _Float16 *)w_lo, w_scale, (const _Float16 *)x_hi,
                     (const _Float16 *)x_lo, x_scale, bias, V, R, K, partials, (int64_t)0, tiles_m, tiles_n,
                     debug_flags());
}

extern "C" int gvl_greedy_step_partials_f32(const float *partials, int R, int V, int first_step, int64_t *token,
                                            float *logp, unsigned char *unfinished, int64_t *seq_col, float *seq_lp_col,
                                            int seq_ld, void *stream) {
  if (R < 0 || V <= 0 || (unfinished && seq_ld <= 0)) return fail(GVL_EINVAL, "gvl_greedy_step_partials_f32: bad sizes");
  if (R == 0) return 0;
  if (!partials || !token || !logp || (unfinished && (!seq_col || !seq_lp_col)))
    return fail(GVL_EINVAL, "gvl_greedy_step_partials_f32: null pointer");
  const GreedyBook book = {unfinished, seq_col, seq_lp_col, seq_ld, first_step != 0};
  return gvl::launch(GVL_PROF_ROW_ARGMAX, R, V, "k_greedy_from_partials", k_greedy_from_partials, dim3((R + 31) / 32),
                     dim3(256), 0, (hipStream_t)stream, (const float4 *)partials, R, gvl_gemm_f16x3_argmax_chunks(V), token,
                     logp, book);
}
