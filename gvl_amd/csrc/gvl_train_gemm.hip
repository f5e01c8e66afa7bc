// gvl_train_gemm.hip -- the dense layers of the TRAINING step on the fp16 matrix cores at fp32 accuracy.
//
// autograd differentiates y = x W^T + b (every nn.Linear of pdvc/deformable_transformer.py:189-199,257-280 and
// pdvc/ops/modules/ms_deform_attn.py:95,99-100,125) into  dx = dy W,  dW = dy^T x,  db = sum_r dy[r].  The first product has the
// shape of the forward one (rows x contraction, both operands contraction-major: gvl_linear_f16x3_f32 serves it on the planes of
// W^T); the second sums over the ROWS of two activations -- both operands arrive contraction-MINOR -- and is this file's kernel:
//
// k_wgrad_f16x3     dW[n][k] = sum_r dy[r][n] x[r][k]  (+ db[n] = sum_r dy[r][n] from the same pass over dy).
//                   Tiles of 128 (n) x 128 (k), the rows cut into `SK` contiguous ranges (split-K: a 512 x 512 weight is only 16
//                   tiles), 32 rows per stage.  Both fp32 tiles arrive by coalesced 16-byte loads (a row of the tile is 512
//                   contiguous bytes), are split in registers and written to LDS as [row][column] fp16 planes;
//                   ds_read_b64_tr_b16 -- gfx950's transposing LDS read -- delivers them column-major, i.e. as the MFMA's
//                   K-contiguous fragments, so no transposed copy of either activation ever exists.
//                   SPLIT: t = v 2^11 / s (s = a power of two >= the TENSOR's maximum / 2: the contraction runs over the rows, so
//                   the scale cannot be per row), hi = fp16(t), lo = fp16(t - hi) -- unlike gvl_gemm16.hip's (hi, 2^11 lo) pair
//                   the residual keeps the scale of hi, so hi.hi + hi.lo + lo.hi accumulate in ONE fp32 accumulator
//                   (32 instead of 64 registers per 32 x 64 wavefront tile) and the fragment needs no rescaling.
//                   |error| <= 2^-22 sum |dy||x| + R 2^-36 max|dy| max|x|  (lo.lo dropped; elements below 2^-25 of the tensor
//                   maximum lose relative precision, which an fp32 accumulation of the same sum loses too).
//                   Split-K partial tiles go to a workspace and k_wgrad_reduce adds them in a FIXED order (deterministic;
//                   float atomics would cap at 1.3 TB/s: 16 MB of partials = 12 us) -- optionally ON TOP of the existing
//                   gradient (AccumulateGrad's add folded in).  SK = 1 (the vocabulary layer: 268 tiles) stores directly.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gvl_common.hpp"
#include "gvl_gemm16_common.hpp"
#include "gvl_msda.h"

namespace {

using gvl::fail;
using namespace gvl16;

typedef __fp16 trh4 __attribute__((__vector_size__(4 * sizeof(__fp16))));

constexpr int kWgT = 128;                  // tile edge (both n and k)
constexpr int kWgR = 32;                   // contraction rows per stage
constexpr int kWgThreads = 512;
constexpr int kWgRowB = 2 * kWgT + 64;     // bytes per LDS row of a plane: 256 + 64 -- four consecutive rows of a transposed
                                           // read then start 64 B apart (mod 256): the 32 lanes of a half cover all 64 banks once
constexpr int kWgPlaneB = kWgR * kWgRowB;  // 10240
constexpr int kWgStageB = 4 * kWgPlaneB;   // [dy hi | dy lo | x hi | x lo]
constexpr int kWgLds = 2 * kWgStageB;      // 81920

struct WgParams {
  const float *dy, *x;
  int64_t ld_dy, ld_x;
  const float *amax_dy, *amax_x;           // (n_amax_*) row bounds (or one tensor bound)
  int n_amax_dy, n_amax_x;
  int R, N, K, tiles_n, tiles_k, SK, rows_per_split;
  float *part;                             // (SK, N, K) or, with SK == 1, the gradient itself
  float *part_b;                           // (SK, N) or NULL
  int accumulate;                          // SK == 1 only: add to what `part` / `part_b` hold
  const int *live;                         // NULL, or [0] = number of LIVE rows of dy, [1 ..] their indices ascending (k_live_rows): the
                                           // other rows are all zeros; a stage then multiplies 32 consecutive LIST entries
#ifdef GVL_WG_STAMPS
  unsigned long long *stamps;              // timing builds only: [workgroup][8] {memtime, memrealtime} x {start, loop, loop end, end}
#endif
};
#ifdef GVL_WG_STAMPS
#define GVL_WG_STAMP(i)                                                                                         \
  if (threadIdx.x == 0) {                                                                                       \
    p.stamps[blk * 8 + 2 * (i)] = __builtin_amdgcn_s_memtime();                                                 \
    p.stamps[blk * 8 + 2 * (i) + 1] = __builtin_amdgcn_s_memrealtime();                                         \
  }
#else
#define GVL_WG_STAMP(i)
#endif

// 2^11 / s with s = 2^floor(log2 amax)  (exponent clamped so that both are finite and normal), and s 2^-11
__device__ __forceinline__ void tensor_scale(float amax, float &mul, float &back) {
  int e = (int)((__float_as_uint(amax) >> 23) & 0xffu);
  e = min(max(e, 13), 240);
  mul = __uint_as_float((uint32_t)(254 - e + 11) << 23);
  back = __uint_as_float((uint32_t)(e - 11) << 23);
}

__device__ __forceinline__ void split4s(const float4 &v, float mul, uint2 &hi, uint2 &lo) {
  typedef _Float16 h2v __attribute__((ext_vector_type(2)));
  const float t[4] = {v.x * mul, v.y * mul, v.z * mul, v.w * mul};
  _Float16 h[4], l[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    h[c] = (_Float16)t[c];
    l[c] = (_Float16)(t[c] - (float)h[c]);
  }
  hi = make_uint2(__builtin_bit_cast(uint32_t, (h2v){h[0], h[1]}), __builtin_bit_cast(uint32_t, (h2v){h[2], h[3]}));
  lo = make_uint2(__builtin_bit_cast(uint32_t, (h2v){l[0], l[1]}), __builtin_bit_cast(uint32_t, (h2v){l[2], l[3]}));
}

typedef uint32_t u4v __attribute__((ext_vector_type(4)));

// buffer descriptor over [ptr, ptr + bytes): loads beyond the end return 0 -- the range checks of the tiles' edges cost no branch,
// and without branches around the loads hipcc's wait counters stay COUNTED (behind a conditional load every wait is vmcnt(0):
// the three-stage prefetch drained once per stage, memory idle during the MFMAs: 1.28 instead of 0.7 us per stage, measured)
__device__ __forceinline__ auto rsrc_of(const void *ptr, uint32_t bytes) {
  const uint64_t u = (uint64_t)(uintptr_t)ptr;
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)u), hi = __builtin_amdgcn_readfirstlane((uint32_t)(u >> 32));
  return __builtin_amdgcn_make_buffer_rsrc((void *)(uintptr_t)(((uint64_t)hi << 32) | lo), 0, __builtin_amdgcn_readfirstlane(bytes),
                                           0x00020000);
}
constexpr uint32_t kOob = 0x7ffffff0u;       // a byte offset beyond every buffer

__device__ __forceinline__ float block_max(const float *__restrict__ v, int n, float *scratch) {
  const auto rs = rsrc_of(v, (uint32_t)n * 4u);
  float m = 0.f;
  for (int base = 0; base < n; base += kWgThreads * 16) {          // 16 values per thread and round, all loads in flight together
    uint32_t q[16];
#pragma unroll
    for (int i = 0; i < 16; ++i)
      q[i] = __builtin_amdgcn_raw_buffer_load_b32(rs, (uint32_t)(base + i * kWgThreads + (int)threadIdx.x) * 4u, 0, 0);
#pragma unroll
    for (int i = 0; i < 16; ++i) m = fmaxf(m, __uint_as_float(q[i]));
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  if ((threadIdx.x & 63) == 0) scratch[threadIdx.x >> 6] = m;
  __syncthreads();
  float r = scratch[0];
#pragma unroll
  for (int w = 1; w < kWgThreads / 64; ++w) r = fmaxf(r, scratch[w]);
  __syncthreads();
  return r;
}

// X1 (gvl_f16_products(1): training under autocast): the leading fp16 product only -- no lo planes are formed, stored or read, one
// MFMA per fragment pair instead of three; operands rounded to 11 significant bits at their tensor scale, fp32 accumulation.
// one work item (row range sk, tile) of one weight gradient; `blk`: the workgroup's index for the timing stamps
template <bool X1>
__device__ __forceinline__ void wgrad_item(const WgParams &p, const int item, const int blk, unsigned char *smem) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tiles = p.tiles_n * p.tiles_k;
  const int sk = item / tiles, tile = item % tiles;
  const int tn = tile / p.tiles_k, tk = tile % p.tiles_k;
  const int n0 = tn * kWgT, k0 = tk * kWgT;
  // (live list: the row range of a split is a range of LIST entries, whole stages of 32; the descriptors then span all rows)
  const int n_live = p.live ? p.live[0] : 0;
  const int per_live = ((n_live + p.SK - 1) / p.SK + kWgR - 1) / kWgR * kWgR, first_live = sk * per_live;
  const int r_begin = p.live ? 0 : sk * p.rows_per_split, r_end = p.live ? p.R : min(p.R, r_begin + p.rows_per_split);

  GVL_WG_STAMP(0)
  float mul_a, back_a, mul_b, back_b;
  tensor_scale(block_max(p.amax_dy, p.n_amax_dy, reinterpret_cast<float *>(smem)), mul_a, back_a);
  tensor_scale(block_max(p.amax_x, p.n_amax_x, reinterpret_cast<float *>(smem)), mul_b, back_b);

  // ---- load path: thread t carries the 16-byte piece t & 31 of tile rows (t >> 5) and (t >> 5) + 16, of both operands
  const int c4 = tid & 31, lr = tid >> 5;
  // (rows beyond the range and columns beyond N / K read as zeros through the descriptor's range check)
  const int nrows = max(r_end - r_begin, 0);
  const auto a_rs = rsrc_of(p.dy + (int64_t)r_begin * p.ld_dy, (uint32_t)((int64_t)nrows * p.ld_dy * 4));
  const auto b_rs = rsrc_of(p.x + (int64_t)r_begin * p.ld_x, (uint32_t)((int64_t)nrows * p.ld_x * 4));
  const uint32_t a_col = n0 + 4 * c4 < p.N ? (uint32_t)(n0 + 4 * c4) * 4u : kOob;
  const uint32_t b_col = k0 + 4 * c4 < p.K ? (uint32_t)(k0 + 4 * c4) * 4u : kOob;
  const uint32_t a_ld = (uint32_t)p.ld_dy * 4u, b_ld = (uint32_t)p.ld_x * 4u;
  const int a_tail = p.N - (n0 + 4 * c4);            // valid columns of this thread's 16-byte piece of dy (>= 4: all)
  const uint32_t st_off = (uint32_t)(lr * kWgRowB + c4 * 8);
  struct Set { float4 a[2], b[2]; };
  auto load = [&](Set &s, int stage_i) {
    // (stage_i counts this item's stages; with a live list it indexes the list -- past its end: a stage beyond the last row, whose
    //  loads fall outside the descriptors and come back as zeros, like the pipeline's look-ahead past a plain range)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      uint32_t r = (uint32_t)(stage_i * kWgR + lr + 16 * i);
      if (p.live) {                                  // entry r of this item's part of the list -> its row (past the end: no row)
        const int e = first_live + (int)r;
        r = ((int)r < per_live && e < n_live) ? (uint32_t)p.live[1 + e] : (uint32_t)p.R;
      }
#ifdef GVL_WG_NO_LOAD
      s.a[i] = make_float4((float)r, 1.f, 2.f, 3.f); s.b[i] = make_float4(1.f, (float)r, 2.f, 3.f);
#else
      s.a[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(a_rs, min(r * a_ld + a_col, kOob), 0, 0));
      if (a_tail < 4) {                              // the piece that straddles column N (N % 4 != 0: the vocabulary layer)
        if (a_tail < 2) s.a[i].y = 0.f;
        if (a_tail < 3) s.a[i].z = 0.f;
        s.a[i].w = 0.f;
      }
      s.b[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(b_rs, min(r * b_ld + b_col, kOob), 0, 0));
#endif
    }
  };
  float4 bsum = make_float4(0.f, 0.f, 0.f, 0.f);                      // this thread's share of db (tile column 0 only)
  auto store = [&](const Set &s, int buf) {
    unsigned char *st = smem + buf * kWgStageB + st_off;
#ifdef GVL_WG_NO_STORE
    for (int i = 0; i < 2; ++i) asm volatile("" ::"v"(s.a[i].x), "v"(s.a[i].w), "v"(s.b[i].x), "v"(s.b[i].w));
    if (false)
#endif
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      uint2 hi, lo;
      split4s(s.a[i], mul_a, hi, lo);
      *reinterpret_cast<uint2 *>(st + 16 * i * kWgRowB) = hi;
      if constexpr (!X1) *reinterpret_cast<uint2 *>(st + kWgPlaneB + 16 * i * kWgRowB) = lo;
      split4s(s.b[i], mul_b, hi, lo);
      *reinterpret_cast<uint2 *>(st + 2 * kWgPlaneB + 16 * i * kWgRowB) = hi;
      if constexpr (!X1) *reinterpret_cast<uint2 *>(st + 3 * kWgPlaneB + 16 * i * kWgRowB) = lo;
      bsum.x += s.a[i].x; bsum.y += s.a[i].y; bsum.z += s.a[i].z; bsum.w += s.a[i].w;
    }
  };

  // ---- fragments.  Wavefront w: rows (n) 32 (w & 3) .. + 31 of the tile, columns (k) 64 (w >> 2) .. + 63 (two MFMA tiles).
  // v_mfma_f32_32x32x16_f16: lane l holds A[n = l & 31][c = 8 (l >> 5) + j], B[c][k = l & 31], j = 0 .. 7.  One transposed read
  // gives a lane four consecutive c of its column: the 16-lane group g = l >> 4 reads the block of rows c0 + 8 (g >> 1) ..
  // + 3 (second read: + 4), columns 16 (g & 1) .. + 15 of the wavefront's 32; lane 4 q + p of the group supplies row q, 4 p ..
  const int wn = 32 * (wave & 3), wk = 64 * (wave >> 2);
  const int g = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
  const uint32_t fr_row = (uint32_t)((8 * (g >> 1) + q) * kWgRowB);
  const uint32_t fa = fr_row + (uint32_t)((wn + 16 * (g & 1) + 4 * pp) * 2);
  const uint32_t fb = 2 * kWgPlaneB + fr_row + (uint32_t)((wk + 16 * (g & 1) + 4 * pp) * 2);
  auto tr = [&](uint32_t byte_off) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) trh4 *)(smem + byte_off));
  };
  auto frag = [&](uint32_t off) {      // 8 consecutive contraction rows of this lane's column
#ifdef GVL_WG_NO_FRAG
    return __builtin_bit_cast(h8, make_uint4(off, 1u, 2u, 3u));
#endif
    const trh4 u = tr(off), v = tr(off + 4 * kWgRowB);
    return __builtin_bit_cast(h8, (__fp16 __attribute__((__vector_size__(16)))){u[0], u[1], u[2], u[3], v[0], v[1], v[2], v[3]});
  };

  f16acc acc[2];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

  // ---- pipeline.  Stage s lives in LDS buffer s & 1; its fp32 rows sit in register set (s % 3) from three stages before it is
  // split and written (during stage s - 1), so a row's trip from memory has up to three stage times.  The two wavefronts of a
  // SIMD would run the same phases in lockstep behind the stage barrier (fragment reads, split + LDS writes, MFMAs): wavefronts
  // 4-7 therefore run their MFMAs FIRST and split afterwards, wavefronts 0-3 the other way round -- while one half occupies the
  // matrix pipe the other occupies the vector / LDS-write path (MI355X_MICROARCH.md, two waves per SIMD, item 9).
  const int nst = p.live ? (max(0, min(per_live, n_live - first_live)) + kWgR - 1) / kWgR : (r_end - r_begin + kWgR - 1) / kWgR;
  const bool mfma_first = __builtin_amdgcn_readfirstlane(wave) >= 4;
  Set set0, set1, set2;
  load(set0, 0);
  load(set1, 1);
  load(set2, 2);
  store(set0, 0);
  load(set0, 3);
  __syncthreads();
#ifdef GVL_WG_NO_MFMA
#define GVL_WG_MFMA()                                                                                   \
  _Pragma("unroll") for (int h = 0; h < 2; ++h) _Pragma("unroll") for (int j = 0; j < 2; ++j) {         \
    asm volatile("" ::"v"(ah[h]), "v"(al[h]), "v"(bh[h][j]), "v"(bl[h][j]));                            \
  }
#else
#define GVL_WG_MFMA()                                                                                   \
  _Pragma("unroll") for (int h = 0; h < 2; ++h) _Pragma("unroll") for (int j = 0; j < 2; ++j) {         \
    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[h], bh[h][j], acc[j], 0, 0, 0);                  \
    if constexpr (!X1) {                                                                                \
      acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[h], bl[h][j], acc[j], 0, 0, 0);                \
      acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[h], bh[h][j], acc[j], 0, 0, 0);                \
    }                                                                                                   \
  }
#endif
  // one stage: SET holds the rows of stage S + 1 and is re-requested for stage S + 4 as soon as they are in LDS.  FIRST / SECOND:
  // the two phases in the order of this wavefront's half -- the choice is made ONCE, outside the loop (inside, the accumulators
  // would pass through a phi per stage: hipcc copies all 32 of them twice per stage and every copy waits for the matrix pipe to drain)
#define GVL_WG_PHASE_MFMA(SET, S) GVL_WG_MFMA()
#define GVL_WG_PHASE_STORE(SET, S)                                                                      \
  store(SET, ((S) & 1) ^ 1);                                                                            \
  load(SET, (S) + 4);
#define GVL_WG_STAGE(FIRST, SECOND, SET, S)                                                             \
  {                                                                                                      \
    const uint32_t sb = (uint32_t)(((S) & 1) * kWgStageB);                                               \
    h8 ah[2], al[2], bh[2][2], bl[2][2];                                                                 \
    _Pragma("unroll") for (int h = 0; h < 2; ++h) {                                                      \
      const uint32_t ro = sb + (uint32_t)(16 * h * kWgRowB);                                             \
      ah[h] = frag(ro + fa);                                                                             \
      if constexpr (!X1) al[h] = frag(ro + kWgPlaneB + fa);                                              \
      _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                                    \
        bh[h][j] = frag(ro + fb + 64 * j);                                                               \
        if constexpr (!X1) bl[h][j] = frag(ro + kWgPlaneB + fb + 64 * j);                                \
      }                                                                                                  \
    }                                                                                                    \
    FIRST(SET, S)                                                                                        \
    SECOND(SET, S)                                                                                       \
    __syncthreads();                                                                                     \
  }
#define GVL_WG_LOOP(FIRST, SECOND)                                                                      \
  {                                                                                                      \
    int s = 0;                                                                                           \
    for (; s + 3 <= nst; s += 3) {                                                                       \
      GVL_WG_STAGE(FIRST, SECOND, set1, s)                                                               \
      GVL_WG_STAGE(FIRST, SECOND, set2, s + 1)                                                           \
      GVL_WG_STAGE(FIRST, SECOND, set0, s + 2)                                                           \
    }                                                                                                    \
    if (s < nst) GVL_WG_STAGE(FIRST, SECOND, set1, s)                                                    \
    if (s + 1 < nst) GVL_WG_STAGE(FIRST, SECOND, set2, s + 1)                                            \
  }
  GVL_WG_STAMP(1)
  if (mfma_first) GVL_WG_LOOP(GVL_WG_PHASE_MFMA, GVL_WG_PHASE_STORE)
  else GVL_WG_LOOP(GVL_WG_PHASE_STORE, GVL_WG_PHASE_MFMA)
#undef GVL_WG_LOOP
#undef GVL_WG_PHASE_MFMA
#undef GVL_WG_PHASE_STORE
#undef GVL_WG_STAGE
#undef GVL_WG_MFMA
  GVL_WG_STAMP(2)

  // ---- epilogue: C/D map column (k) = lane & 31, row (n) = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
  const float sc = back_a * back_b;
  float *out = p.part + (int64_t)sk * p.N * p.K;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int k = k0 + wk + 32 * j + (lane & 31);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int n = n0 + wn + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      if (n < p.N && k < p.K) {
        float *o = out + (int64_t)n * p.K + k;
        const float v = acc[j][r] * sc;
        *o = (p.SK == 1 && p.accumulate) ? *o + v : v;
      }
    }
  }
  if (p.part_b && tk == 0) {
    // column sums of dy: the 16 threads with the same c4 (one per row group) add up in a fixed order through LDS
    float4 *red = reinterpret_cast<float4 *>(smem);
    red[lr * 32 + c4] = bsum;
    __syncthreads();
    if (tid < 32 && n0 + 4 * tid < p.N) {
      float4 t = red[tid];
#pragma unroll
      for (int i = 1; i < 16; ++i) {
        const float4 u = red[i * 32 + tid];
        t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w;
      }
      float *o = p.part_b + (int64_t)sk * ((p.N + 3) & ~3) + n0 + 4 * tid;      // (partial bias rows: stride N rounded up to 4)
      const float tv[4] = {t.x, t.y, t.z, t.w};
      const bool add = p.SK == 1 && p.accumulate;
#pragma unroll
      for (int c = 0; c < 4; ++c)
        if (n0 + 4 * tid + c < p.N) o[c] = add ? o[c] + tv[c] : tv[c];
    }
  }
  GVL_WG_STAMP(3)
}

// live[0] = number of rows of dy with a non-zero bound, live[1 ..] = their indices, ascending (one wavefront; a ballot per 64 rows keeps
// the order).  A row whose bound is 0 is all zeros: the padded positions of a teacher-forced caption batch -- most rows of the
// vocabulary layer's (4416, 8518) gradient at cfg A.
__global__ void __launch_bounds__(64) k_live_rows(const float *__restrict__ amax, int R, int *__restrict__ live) {
  const int lane = threadIdx.x;
  int base = 0;
  for (int r0 = 0; r0 < R; r0 += 64) {
    const int r = r0 + lane;
    const bool any = r < R && amax[r] != 0.f;
    const unsigned long long m = __ballot(any);
    if (any) live[1 + base + __popcll(m & ((1ull << lane) - 1ull))] = r;
    base += __popcll(m);
  }
  if (lane == 0) live[0] = base;
}

template <bool X1>
__global__ void __launch_bounds__(kWgThreads, 2) k_wgrad_f16x3(const WgParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  // workgroup id -> (row range, tile): the work items in the order (row range, tile row, tile column) are cut into 8 contiguous
  // runs, run x on XCD x (= workgroup id % 8; speed only).  Neighbours in that order read the same rows of dy and x (each tile
  // a quarter of them at N = K = 512), so an XCD's L2 fetches them once instead of every XCD fetching everything.
  const int total = p.tiles_n * p.tiles_k * p.SK, per = (total + 7) >> 3;
  const int item = ((int)blockIdx.x & 7) * per + ((int)blockIdx.x >> 3);
  if (((int)blockIdx.x >> 3) >= per || item >= total) return;
  wgrad_item<X1>(p, item, (int)blockIdx.x, smem);
}

// GROUPED form (round 6): the weight gradients of SEVERAL Linear layers in one launch.  Alone, a 512 x 512 gradient has 16 output
// tiles; to cover the chip it was cut into ~15 row ranges whose partial tiles (15 MB written, 15 MB read back by k_wgrad_reduce)
// cost as much HBM traffic as the operands, in workgroups of ten K stages each.  A backward pass owes the gradients of 5-9
// Linears per layer at nearly the same time and none of them is on its critical path (nobody reads dW before the optimizer): taken
// together they fill the chip with 2-4 row ranges each -- a quarter of the partial traffic, workgroups of 40-75 stages, one launch
// and one reduction instead of a pair per Linear.  Work items of all problems form one list, cut into 8 runs for the XCDs as above.
constexpr int kWgGroupMax = 10;
struct WgGroup {
  int n, total;
  int first[kWgGroupMax + 1];            // first work item of problem i; first[n] = total
  WgParams p[kWgGroupMax];
};
template <bool X1>
__global__ void __launch_bounds__(kWgThreads, 2) k_wgrad_group_f16x3(const WgGroup g) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int per = (g.total + 7) >> 3;
  const int item = ((int)blockIdx.x & 7) * per + ((int)blockIdx.x >> 3);
  if (((int)blockIdx.x >> 3) >= per || item >= g.total) return;
  int i = 0;
#pragma unroll 1
  while (i + 1 < g.n && item >= g.first[i + 1]) ++i;
  wgrad_item<X1>(g.p[i], item - g.first[i], (int)blockIdx.x, smem);
}

// the grouped reduction: every problem's partial slabs summed in slab order into its gradient (and bias row); thread blocks of 256
// float4 / bias elements, `first` counts blocks
struct WgReduceGroup {
  int n;
  int first[kWgGroupMax + 1];
  struct Item { const float4 *part; float4 *grad; const float *part_b; float *grad_b; int64_t n4; int SK, N, accumulate; } r[kWgGroupMax];
};
__global__ void __launch_bounds__(256) k_wgrad_reduce_group(const WgReduceGroup g) {
  int i = 0;
#pragma unroll 1
  while (i + 1 < g.n && (int)blockIdx.x >= g.first[i + 1]) ++i;
  const WgReduceGroup::Item &r = g.r[i];
  const int64_t e = (int64_t)((int)blockIdx.x - g.first[i]) * 256 + threadIdx.x;
  if (e < r.n4) {
    float4 t = r.accumulate ? r.grad[e] : make_float4(0.f, 0.f, 0.f, 0.f);
    for (int s_ = 0; s_ < r.SK; ++s_) {
      const float4 u = r.part[(int64_t)s_ * r.n4 + e];
      t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w;
    }
    r.grad[e] = t;
  } else if (r.grad_b && e - r.n4 < r.N) {
    const int64_t b = e - r.n4;
    const int stride = (r.N + 3) & ~3;
    float t = r.accumulate ? r.grad_b[b] : 0.f;
    for (int s_ = 0; s_ < r.SK; ++s_) t += r.part_b[(int64_t)s_ * stride + b];
    r.grad_b[b] = t;
  }
}

// grad (+)= sum over the SK partial slabs, in slab order; the same for the bias row
__global__ void __launch_bounds__(256) k_wgrad_reduce(const float4 *__restrict__ part, int64_t n4, int SK, float4 *__restrict__ grad,
                                                      const float *__restrict__ part_b, int N, float *__restrict__ grad_b,
                                                      int accumulate) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n4) {
    float4 t = accumulate ? grad[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    for (int s = 0; s < SK; ++s) {
      const float4 u = part[(int64_t)s * n4 + i];
      t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w;
    }
    grad[i] = t;
  } else if (grad_b && i - n4 < N) {
    const int64_t b = i - n4;
    const int stride = (N + 3) & ~3;
    float t = accumulate ? grad_b[b] : 0.f;
    for (int s = 0; s < SK; ++s) t += part_b[(int64_t)s * stride + b];
    grad_b[b] = t;
  }
}



// ---------------------------------------------------------------------------------------------------------------------
// Operand planes of EVERY weight of the training step, refreshed by two launches per step (the weights change with every
// optimizer update; one gvl_split_rows_f16 per matrix and orientation would be ~150 launches).  For each matrix W (N, K):
//   planes of W     rows n, contraction k   -- the forward product  y = x W^T            (gvl_linear_f16x3_f32)
//   planes of W^T   rows k, contraction n   -- the input gradient   dx = dy W  = dy (W^T)^T  (the same kernel)
// both in gvl_gemm16's (hi, 2^11 lo) K-stage-major form.  ONE power-of-two scale per GROUP of matrices (a group = the blocks
// of one concatenated operand, e.g. [sampling_offsets ; attention_weights]): in W^T the contraction runs over the rows of W, so a
// per-row scale of W cannot be used there, and with one number both orientations hold the SAME fp16 values.
//   k_planes_amax     max |w| per chunk of 64 K elements (plain stores: no atomics, nothing to zero between steps)
//   k_planes_split    one wavefront per 32 x 32 tile: split, store the tile's 2 KB run of the W planes; transpose through LDS
//                     (ds_read_b64_tr_b16) and store the 2 KB run of the W^T planes
constexpr int kPlChunk = 65536;

__global__ void __launch_bounds__(256) k_planes_amax(const gvl_plane_desc *__restrict__ descs, const int2 *__restrict__ chunk_map,
                                                     float *__restrict__ chunk_amax) {
  const int2 cm = chunk_map[blockIdx.x];                     // {descriptor, chunk of that matrix}
  const gvl_plane_desc d = descs[cm.x];
  const int64_t total = (int64_t)d.N * d.K, begin = (int64_t)cm.y * kPlChunk, end = min(total, begin + kPlChunk);
  const float4 *w4 = reinterpret_cast<const float4 *>(d.w + begin);
  const int n4 = (int)((end - begin) >> 2);
  float m = 0.f;
  for (int i = threadIdx.x; i < n4; i += 1024) {             // four loads in flight per thread
    float4 v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = i + 256 * j < n4 ? w4[i + 256 * j] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int j = 0; j < 4; ++j) m = fmaxf(fmaxf(m, fmaxf(fabsf(v[j].x), fabsf(v[j].y))), fmaxf(fabsf(v[j].z), fabsf(v[j].w)));
  }
  __shared__ float red[4];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) chunk_amax[blockIdx.x] = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

__global__ void __launch_bounds__(256) k_planes_split(const gvl_plane_desc *__restrict__ descs, const int2 *__restrict__ wg_map,
                                                      const float *__restrict__ chunk_amax) {
  __shared__ __attribute__((aligned(16))) unsigned char tr_tile[4][2][32 * 64];      // [wavefront][hi | lo][n][k]: 64-byte rows
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int2 wm = wg_map[blockIdx.x];                        // {descriptor, first tile of this workgroup in the matrix}
  const gvl_plane_desc d = descs[wm.x];
  const int tiles_k = d.K >> 5, tile = wm.y + wave;
  // the group's scale: every wavefront reduces the group's chunk maxima itself (<= a few hundred numbers)
  float m = 0.f;
  for (int i = lane; i < d.group_chunks; i += 64) m = fmaxf(m, chunk_amax[d.group_chunk_begin + i]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  int e = (int)((__float_as_uint(m) >> 23) & 0xffu);
  e = min(max(e, 1), 253);
  const float s = __uint_as_float((uint32_t)e << 23), inv = __uint_as_float((uint32_t)(254 - e) << 23);
  const bool live = tile < ((d.N + 31) >> 5) * tiles_k;      // (wavefront-uniform; a matrix's tile count is padded to 4)
  const int tn = live ? tile / tiles_k : 0, tk = live ? tile % tiles_k : 0;
  const int n0 = 32 * tn, k0 = 32 * tk, r8 = lane >> 3, c4 = lane & 7;
  typedef _Float16 h2v __attribute__((ext_vector_type(2)));
  _Float16 *hi = (_Float16 *)d.hi, *lo = (_Float16 *)d.lo;
  const int64_t run = ((int64_t)tk * d.n_total + d.n_off + n0) * 32;                // this tile's 1024 halves of the W planes
  unsigned char *th = tr_tile[wave][0], *tl = tr_tile[wave][1];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int row = r8 + 8 * j;
    const bool in = n0 + row < d.N;                           // (N % 32 != 0, the vocabulary layer: rows beyond N are zeros)
    float4 v = *reinterpret_cast<const float4 *>(d.w + (int64_t)min(n0 + row, d.N - 1) * d.K + k0 + 4 * c4);
    if (!in) v = make_float4(0.f, 0.f, 0.f, 0.f);
    const float a[4] = {v.x * inv, v.y * inv, v.z * inv, v.w * inv};
    _Float16 h[4], l[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      h[c] = (_Float16)a[c];
      l[c] = (_Float16)((a[c] - (float)h[c]) * kLoScale);
    }
    const uint2 ph = make_uint2(__builtin_bit_cast(uint32_t, (h2v){h[0], h[1]}), __builtin_bit_cast(uint32_t, (h2v){h[2], h[3]}));
    const uint2 pl = make_uint2(__builtin_bit_cast(uint32_t, (h2v){l[0], l[1]}), __builtin_bit_cast(uint32_t, (h2v){l[2], l[3]}));
    if (live && in) {
      *reinterpret_cast<uint2 *>(hi + run + row * 32 + 4 * c4) = ph;
      *reinterpret_cast<uint2 *>(lo + run + row * 32 + 4 * c4) = pl;
    }
    *reinterpret_cast<uint2 *>(th + row * 64 + c4 * 8) = ph;
    *reinterpret_cast<uint2 *>(tl + row * 64 + c4 * 8) = pl;
  }
  if (live && tk == 0 && lane < 32 && n0 + lane < d.N) {
    d.scale[d.n_off + n0 + lane] = s;
    if (d.bias_dst) d.bias_dst[d.n_off + n0 + lane] = d.bias ? d.bias[n0 + lane] : 0.f;
  }
  if (!d.t_hi) return;
  if (live && tn == 0 && lane < 32) d.t_scale[k0 + lane] = s;
  // transposed 2 KB run: row k of the tile = 32 consecutive n.  One transposed read hands a lane 4 consecutive n of its k:
  // group g = lane >> 4 reads the block rows (n) 8 m + 4 u .. + 3, columns (k) 16 (g & 1) .. + 15 with m = 2 r + (g >> 1)
  // (r = 0, 1: the wavefront's two rounds), u = 0, 1: the two reads whose results are 8 consecutive n = one 16-byte store
  const int g = lane >> 4, i16 = lane & 15, q = (lane >> 2) & 3, pp = lane & 3;
  _Float16 *thi = (_Float16 *)d.t_hi, *tlo = (_Float16 *)d.t_lo;
  const int64_t trun = ((int64_t)((d.n_off + n0) >> 5) * d.K + k0) * 32;
  __builtin_amdgcn_s_waitcnt(0xc07f);                        // lgkmcnt(0): this wavefront's own LDS stores (no other wavefront reads them)
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    const int mblk = 2 * r + (g >> 1);
    const uint32_t src = (uint32_t)((8 * mblk + q) * 64 + (16 * (g & 1) + 4 * pp) * 2);
    const trh4 h0 = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) trh4 *)(th + src));
    const trh4 h1 = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) trh4 *)(th + src + 4 * 64));
    const trh4 l0 = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) trh4 *)(tl + src));
    const trh4 l1 = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) trh4 *)(tl + src + 4 * 64));
    if (live) {
      const int k = 16 * (g & 1) + i16;
      typedef __fp16 trh8 __attribute__((__vector_size__(16)));
      *reinterpret_cast<trh8 *>(thi + trun + k * 32 + 8 * mblk) = (trh8){h0[0], h0[1], h0[2], h0[3], h1[0], h1[1], h1[2], h1[3]};
      *reinterpret_cast<trh8 *>(tlo + trun + k * 32 + 8 * mblk) = (trh8){l0[0], l0[1], l0[2], l0[3], l1[0], l1[1], l1[2], l1[3]};
    }
  }
}

#ifdef GVL_WG_STAMPS
unsigned long long *g_wg_stamps = nullptr;
#endif
struct WgPlan { int tiles_n, tiles_k, SK, rows_per_split; };

WgPlan wgrad_plan(int R, int N, int K) {
  WgPlan pl;
  pl.tiles_n = (N + kWgT - 1) / kWgT;
  pl.tiles_k = (K + kWgT - 1) / kWgT;
  const int tiles = pl.tiles_n * pl.tiles_k, stages = (R + kWgR - 1) / kWgR;
  int sk = 256 / tiles;                                     // about one workgroup per compute unit ...
  sk = max(1, min(sk, stages / 4));                         // ... of at least four stages
  const int per = (stages + sk - 1) / sk;
  pl.rows_per_split = per * kWgR;
  pl.SK = (stages + per - 1) / per;
  return pl;
}

}  // namespace


extern "C" int gvl_planes_chunk_elems(void) { return kPlChunk; }

extern "C" int gvl_planes_refresh_f16(const gvl_plane_desc *descs_device, const int *chunk_map_device, int n_chunks,
                                      const int *wg_map_device, int n_workgroups, float *chunk_amax_device, void *stream) {
  if (!descs_device || !chunk_map_device || !wg_map_device || !chunk_amax_device || n_chunks <= 0 || n_workgroups <= 0)
    return fail(GVL_EINVAL, "gvl_planes_refresh_f16: null pointer / empty launch");
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(k_planes_amax, dim3(n_chunks), dim3(256), 0, st, descs_device, reinterpret_cast<const int2 *>(chunk_map_device),
                     chunk_amax_device);
  hipLaunchKernelGGL(k_planes_split, dim3(n_workgroups), dim3(256), 0, st, descs_device,
                     reinterpret_cast<const int2 *>(wg_map_device), chunk_amax_device);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : fail((int)e, "gvl_planes_refresh_f16: launch failed: %s", hipGetErrorString(e));
}

namespace {
// row ranges per problem of a group: about two workgroups per compute unit over the whole group, ranges of at least four stages
WgPlan wgrad_group_plan(int R, int N, int K, int group_tiles) {
  WgPlan pl;
  pl.tiles_n = (N + kWgT - 1) / kWgT;
  pl.tiles_k = (K + kWgT - 1) / kWgT;
  const int stages = (R + kWgR - 1) / kWgR;
  int sk = (512 + group_tiles / 2) / (group_tiles > 0 ? group_tiles : 1);
  sk = max(1, min(sk, stages / 4));
  const int per = (stages + sk - 1) / sk;
  pl.rows_per_split = per * kWgR;
  pl.SK = (stages + per - 1) / per;
  return pl;
}
int wgrad_group_tiles(const gvl_wgrad_desc *d, int n) {
  int t = 0;
  for (int i = 0; i < n; ++i) t += ((d[i].N + kWgT - 1) / kWgT) * ((d[i].K + kWgT - 1) / kWgT);
  return t;
}
size_t wgrad_part_bytes(const WgPlan &pl, int N, int K) {
  return pl.SK == 1 ? 0 : ((size_t)pl.SK * ((size_t)N * K + ((N + 3) & ~3)) * sizeof(float) + 15) & ~(size_t)15;
}
}  // namespace

extern "C" int gvl_wgrad_group_max(void) { return kWgGroupMax; }

extern "C" size_t gvl_wgrad_group_workspace_bytes(const gvl_wgrad_desc *descs, int n) {
  if (!descs || n <= 0 || n > kWgGroupMax) return 0;
  const int tiles = wgrad_group_tiles(descs, n);
  size_t total = 0;
  for (int i = 0; i < n; ++i) total += wgrad_part_bytes(wgrad_group_plan(descs[i].R, descs[i].N, descs[i].K, tiles), descs[i].N, descs[i].K);
  return total;
}

extern "C" int gvl_wgrad_group_f16x3_f32(const gvl_wgrad_desc *descs, int n, void *workspace, size_t workspace_bytes, void *stream) {
  if (!descs || n <= 0 || n > kWgGroupMax) return fail(GVL_EINVAL, "gvl_wgrad_group_f16x3_f32: 1 .. %d problems (got %d)", kWgGroupMax, n);
  const int tiles = wgrad_group_tiles(descs, n);
  WgGroup g;
  WgReduceGroup rg;
  g.n = n;
  rg.n = 0;
  int items = 0, rblocks = 0;
  size_t used = 0;
  for (int i = 0; i < n; ++i) {
    const gvl_wgrad_desc &d = descs[i];
    if (!d.dy || !d.x || !d.amax_dy || !d.amax_x || !d.grad_w) return fail(GVL_EINVAL, "gvl_wgrad_group_f16x3_f32: null pointer (problem %d)", i);
    if (d.R <= 0 || d.N <= 0 || d.K <= 0 || (d.K & 3) || (d.ld_dy & 3) || (d.ld_x & 3) || d.ld_dy < ((d.N + 3) & ~3) || d.ld_x < d.K ||
        d.n_amax_dy < 1 || d.n_amax_x < 1)
      return fail(GVL_EINVAL, "gvl_wgrad_group_f16x3_f32: problem %d: R, N, K > 0, K and both row strides multiples of 4 (R=%d N=%d K=%d)", i, d.R, d.N, d.K);
    if ((int64_t)d.R * d.ld_dy >= ((int64_t)1 << 29) || (int64_t)d.R * d.ld_x >= ((int64_t)1 << 29))
      return fail(GVL_EINVAL, "gvl_wgrad_group_f16x3_f32: operands of 2 GB or more are not addressed (32-bit buffer offsets)");
    if (((uintptr_t)d.dy | (uintptr_t)d.x | (uintptr_t)d.grad_w | (uintptr_t)d.grad_b | (uintptr_t)workspace) & 15)
      return fail(GVL_EINVAL, "gvl_wgrad_group_f16x3_f32: pointers must be 16-byte aligned");
    const WgPlan pl = wgrad_group_plan(d.R, d.N, d.K, tiles);
    WgParams &p = g.p[i];
    p.dy = d.dy; p.x = d.x; p.ld_dy = d.ld_dy; p.ld_x = d.ld_x;
    p.amax_dy = d.amax_dy; p.amax_x = d.amax_x; p.n_amax_dy = d.n_amax_dy; p.n_amax_x = d.n_amax_x;
    p.R = d.R; p.N = d.N; p.K = d.K; p.tiles_n = pl.tiles_n; p.tiles_k = pl.tiles_k; p.SK = pl.SK; p.rows_per_split = pl.rows_per_split;
    p.accumulate = d.accumulate;
    p.live = nullptr;
#ifdef GVL_WG_STAMPS
    p.stamps = g_wg_stamps;
#endif
    if (pl.SK == 1) {
      p.part = d.grad_w;
      p.part_b = d.grad_b;
    } else {
      const size_t need = wgrad_part_bytes(pl, d.N, d.K);
      if (!workspace || used + need > workspace_bytes) return fail(GVL_ENOSPC, "gvl_wgrad_group_f16x3_f32: workspace of %zu bytes needed", gvl_wgrad_group_workspace_bytes(descs, n));
      p.part = reinterpret_cast<float *>(reinterpret_cast<char *>(workspace) + used);
      p.part_b = d.grad_b ? p.part + (size_t)pl.SK * d.N * d.K : nullptr;
      used += need;
      WgReduceGroup::Item &r = rg.r[rg.n];
      r.part = reinterpret_cast<const float4 *>(p.part); r.grad = reinterpret_cast<float4 *>(d.grad_w);
      r.part_b = p.part_b; r.grad_b = d.grad_b; r.n4 = (int64_t)d.N * d.K / 4; r.SK = pl.SK; r.N = d.N; r.accumulate = d.accumulate;
      rg.first[rg.n++] = rblocks;
      rblocks += (int)((r.n4 + (d.grad_b ? d.N : 0) + 255) / 256);
    }
    g.first[i] = items;
    items += pl.tiles_n * pl.tiles_k * pl.SK;
  }
  g.first[n] = g.total = items;
  rg.first[rg.n] = rblocks;
  hipStream_t st = (hipStream_t)stream;
  const bool x1 = gvl16::g_f16_products == 1;
  auto kern = x1 ? k_wgrad_group_f16x3<true> : k_wgrad_group_f16x3<false>;
  if (int rc = gvl::ensure_lds(kern, kWgLds)) return rc;
  if (int rc = gvl::launch(GVL_PROF_WGRAD, n, items, x1 ? "k_wgrad_group_f16x1" : "k_wgrad_group_f16x3", kern, dim3(8 * ((items + 7) / 8)),
                           dim3(kWgThreads), kWgLds, st, g))
    return rc;
  if (rg.n > 0) {
    hipLaunchKernelGGL(k_wgrad_reduce_group, dim3((unsigned)rblocks), dim3(256), 0, st, rg);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail((int)e, "gvl_wgrad_group_f16x3_f32: reduce launch failed: %s", hipGetErrorString(e));
  }
  return 0;
}

extern "C" size_t gvl_wgrad_workspace_bytes(int R, int N, int K) {
  if (R <= 0 || N <= 0 || K <= 0) return 0;
  const WgPlan pl = wgrad_plan(R, N, K);
  return pl.SK == 1 ? 0 : (size_t)pl.SK * ((size_t)N * K + ((N + 3) & ~3)) * sizeof(float);
}

extern "C" int gvl_wgrad_f16x3_f32(const float *dy, int64_t ld_dy, const float *amax_dy, int n_amax_dy, const float *x, int64_t ld_x,
                                   const float *amax_x, int n_amax_x, int R, int N, int K, float *grad_w, float *grad_b,
                                   int accumulate, void *workspace, size_t workspace_bytes, void *stream) {
  return gvl_wgrad_f16x3_live_f32(dy, ld_dy, amax_dy, n_amax_dy, x, ld_x, amax_x, n_amax_x, R, N, K, grad_w, grad_b, accumulate,
                                  workspace, workspace_bytes, nullptr, stream);
}

extern "C" int gvl_wgrad_live_ints(int R) { return R > 0 ? 1 + R : 0; }

extern "C" int gvl_wgrad_f16x3_live_f32(const float *dy, int64_t ld_dy, const float *amax_dy, int n_amax_dy, const float *x, int64_t ld_x,
                                        const float *amax_x, int n_amax_x, int R, int N, int K, float *grad_w, float *grad_b,
                                        int accumulate, void *workspace, size_t workspace_bytes, int *live_ws, void *stream) {
  if (!dy || !x || !amax_dy || !amax_x || !grad_w) return fail(GVL_EINVAL, "gvl_wgrad_f16x3_f32: null pointer");
  if (live_ws && n_amax_dy != R) return fail(GVL_EINVAL, "gvl_wgrad_f16x3_live_f32: the live-stage list needs one bound per row of dy");
  if (R <= 0 || N <= 0 || K <= 0 || (K & 3) || (ld_dy & 3) || (ld_x & 3) || ld_dy < ((N + 3) & ~3) || ld_x < K || n_amax_dy < 1 ||
      n_amax_x < 1)
    return fail(GVL_EINVAL, "gvl_wgrad_f16x3_f32: R, N, K > 0, K and both row strides multiples of 4, ld_dy >= N rounded up to 4 (got R=%d N=%d K=%d)", R, N, K);
  if ((int64_t)R * ld_dy >= ((int64_t)1 << 29) || (int64_t)R * ld_x >= ((int64_t)1 << 29))
    return fail(GVL_EINVAL, "gvl_wgrad_f16x3_f32: operands of 2 GB or more are not addressed (32-bit buffer offsets)");
  if (((uintptr_t)dy | (uintptr_t)x | (uintptr_t)grad_w | (uintptr_t)grad_b | (uintptr_t)workspace) & 15)
    return fail(GVL_EINVAL, "gvl_wgrad_f16x3_f32: pointers must be 16-byte aligned");
  const WgPlan pl = wgrad_plan(R, N, K);
  const size_t need = pl.SK == 1 ? 0 : (size_t)pl.SK * ((size_t)N * K + ((N + 3) & ~3)) * sizeof(float);
  if (need > workspace_bytes || (need && !workspace)) return fail(GVL_ENOSPC, "gvl_wgrad_f16x3_f32: workspace of %zu bytes needed", need);
  hipStream_t st = (hipStream_t)stream;
  WgParams p;
  p.dy = dy; p.x = x; p.ld_dy = ld_dy; p.ld_x = ld_x;
  p.amax_dy = amax_dy; p.amax_x = amax_x; p.n_amax_dy = n_amax_dy; p.n_amax_x = n_amax_x;
  p.R = R; p.N = N; p.K = K; p.tiles_n = pl.tiles_n; p.tiles_k = pl.tiles_k; p.SK = pl.SK; p.rows_per_split = pl.rows_per_split;
  p.accumulate = accumulate;
  p.live = live_ws;
  if (live_ws) {
    hipLaunchKernelGGL(k_live_rows, dim3(1), dim3(64), 0, st, amax_dy, R, live_ws);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail((int)e, "gvl_wgrad_f16x3_live_f32: list launch failed: %s", hipGetErrorString(e));
  }
  if (pl.SK == 1) {
    p.part = grad_w;
    p.part_b = grad_b;
  } else {
    p.part = reinterpret_cast<float *>(workspace);
    p.part_b = grad_b ? p.part + (size_t)pl.SK * N * K : nullptr;
  }
#ifdef GVL_WG_STAMPS
  p.stamps = g_wg_stamps;
#endif
  auto kern = gvl16::g_f16_products == 1 ? k_wgrad_f16x3<true> : k_wgrad_f16x3<false>;
  if (int rc = gvl::ensure_lds(kern, kWgLds)) return rc;
  if (int rc = gvl::launch(GVL_PROF_WGRAD, N, K, gvl16::g_f16_products == 1 ? "k_wgrad_f16x1" : "k_wgrad_f16x3", kern,
                           dim3(8 * ((pl.tiles_n * pl.tiles_k * pl.SK + 7) / 8)), dim3(kWgThreads), kWgLds, st, p))
    return rc;
  if (pl.SK > 1) {
    const int64_t n4 = (int64_t)N * K / 4;
    const int64_t total = n4 + (grad_b ? N : 0);
    hipLaunchKernelGGL(k_wgrad_reduce, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st,
                       reinterpret_cast<const float4 *>(p.part), n4, pl.SK, reinterpret_cast<float4 *>(grad_w),
                       p.part_b, N, grad_b, accumulate);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail((int)e, "gvl_wgrad_f16x3_f32: reduce launch failed: %s", hipGetErrorString(e));
  }
  return 0;
}
