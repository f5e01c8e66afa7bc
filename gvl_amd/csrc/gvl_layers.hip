// gvl_layers.hip -- the dense layers AROUND the deformable attention, for inference: every nn.Linear of the
// encoder / decoder layers (pdvc/ops/modules/ms_deform_attn.py:95,99-100,125: value_proj, sampling_offsets +
// attention_weights, output_proj; pdvc/deformable_transformer.py:189-199,257-261: linear1 / linear2 of the FFN;
// :266-270 the in / out projections of nn.MultiheadAttention; pdvc/pdvc.py:1166-1178 the box MLP) and the LayerNorms
// between them (:193,198,270,277,261), on the fp16 matrix cores at fp32 accuracy.
//
// k_lin_f16x3       out = epilogue((A [+ A2]) . W^T + bias): A is the fp32 ACTIVATION as its producer left it; it is
//                   split into the (hi, 2^11 residual) fp16 planes of gvl_gemm16.hip IN THE LOAD PATH (global -> registers
//                   -> two v_cvt per element -> LDS), so no plane ever exists in HBM and no separate split pass runs;
//                   the weight arrives as planes (split once per parameter version) through LDS-DMA.  The row scale of the
//                   split comes from a per-row maximum `amax` that the PRODUCER of A leaves behind (LayerNorm kernel below:
//                   register-local; the sampling kernel and this kernel's own epilogue: one atomic max per row and tile).
//                   One launch serves several column SEGMENTS of a concatenated weight, each with its own output, epilogue
//                   and A variant (e.g. [value_proj | sampling_offsets ; attention_weights]: the first segment multiplies
//                   src, the second src + pos).  Epilogue per segment: bias, ReLU, residual add, zeroed rows
//                   (masked_fill of ms_deform_attn.py:96-97), row maxima of the result.
// k_ln_rows         LayerNorm of (R, C) rows, one wavefront per row, + the row maxima of the result and of result + pos
// k_row_absmax      row maxima of a tensor some other kernel produced (attention core of nn.MultiheadAttention)
//
// Accuracy: as gvl_gemm16.hip (three fp16 MFMAs per product, fp32 accumulate, lo.lo dropped): |error| <= 2^-21 sum|a||w|
// + K 2^-33 amax_row max|w|; tests/test_gpu_layers.py holds every product to the fp64 result at the fp32 library GEMM's
// own error.  A non-finite element of A makes its output row non-finite (hi = inf / NaN, lo = NaN), as in an fp32 GEMM.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "gvl_common.hpp"
#include "gvl_gemm16_common.hpp"
#include "gvl_msda.h"

namespace {

using gvl::fail;
using namespace gvl16;

constexpr int kLinBN = 64, kMaxSeg = 4;

struct LinParams {
  const float *A;
  int64_t lda;
  const float *A2;
  int64_t lda2;
  int a2_rows;
  const _Float16 *Wh, *Wl;
  const float *Ws, *bias;
  int R, N, K, tiles_m, tiles_n, nseg, xcd_cols;
  int split_stages;            // > 0: split-K -- workgroup row blockIdx.y multiplies K stages [y split_stages, (y + 1) split_stages)
  int64_t split_stride;        //      and stores its partial tile (no epilogue) at out + y split_stride (floats)
  gvl_lin_seg seg[kMaxSeg];
};

// s = 2^floor(log2 amax) (exponent clamped to the range whose reciprocal is representable), inv = 1 / s
__device__ __forceinline__ void scale_of(float amax, float &s, float &inv) {
  int e = (int)((__float_as_uint(amax) >> 23) & 0xffu);
  e = min(max(e, 1), 253);
  s = __uint_as_float((uint32_t)e << 23);
  inv = __uint_as_float((uint32_t)(254 - e) << 23);
}

__device__ __forceinline__ uint32_t pack2(_Float16 a, _Float16 b) {
  typedef _Float16 h2 __attribute__((ext_vector_type(2)));
  return __builtin_bit_cast(uint32_t, (h2){a, b});
}

// 4 consecutive k of one row -> 8 bytes per plane
__device__ __forceinline__ void split4(const float4 &x, float inv, uint2 &hi, uint2 &lo) {
  const float a[4] = {x.x * inv, x.y * inv, x.z * inv, x.w * inv};
  _Float16 hh[4], ll[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    hh[c] = (_Float16)a[c];
    ll[c] = (_Float16)((a[c] - (float)hh[c]) * kLoScale);
  }
  hi = make_uint2(pack2(hh[0], hh[1]), pack2(hh[2], hh[3]));
  lo = make_uint2(pack2(ll[0], ll[1]), pack2(ll[2], ll[3]));
}

// ---------------------------------------------------------------------------------------------------------------------
// Workgroup = 4 wavefronts on a 128 x 64 tile (wavefront tile 64 x 32: 2 MFMA tiles x {leading, cross} accumulators),
// K in stages of 32 through two LDS stage images [A hi | A lo | W hi | W lo] of 24 KB.
//   A path   4 global_load_dwordx4 per thread and stage (+ 4 of the addend), each instruction 8 whole 128-byte lines,
//            requested TWO stages ahead into one of two register sets, split and written as 8 ds_write_b64 into the
//            swizzled plane image one stage ahead;
//   W path   the weight planes travel the same way (one 16-byte chunk per plane and thread).  NOT by LDS-DMA: hipcc marks a
//            pending global_load_lds as a FLAT access of both memories and then forces EVERY later vmcnt / lgkmcnt wait to
//            zero -- the wait in front of the split arithmetic would drain the rows requested for two stages ahead once
//            per stage (seen in the ISA); with ordinary loads the waits are counted (vmcnt(n) = exactly the older set);
//   barrier  one per stage, raw s_barrier after `s_waitcnt lgkmcnt(0)`: the loads in flight are not waited for.
// 304 / 192 tiles for the 4800 / 3008 x 512 products of cfg A, up to 1216 for the FFN: two workgroups per CU.
// Two shapes of the same kernel (WM x WN wavefronts, wavefront tile 32 NI x 32 NJ):
//   <2, 2, 2, 1>  128 x 64 tile, 4 wavefronts: narrow outputs (N < 256), segment boundaries at multiples of 64;
//   <4, 2, 1, 2>  128 x 128 tile, 8 wavefronts: each A element is split once per 128 output columns instead of once per 64
//                 and a stage carries twice the MFMA work per barrier (the split arithmetic, not the matrix cores, bounds
//                 the narrow shape: tools/lin_ksweep.py).
// X1: the leading fp16 product only (gvl_f16_products(1): inference under autocast) -- no lo planes are formed, fetched or read.
template <bool HAS_A2, int WM, int WN, int NI, int NJ, bool X1 = false>
__global__ void __launch_bounds__(64 * WM * WN, (WM * WN == 4 ? 2 : 1)) k_lin_f16x3(const LinParams p) {
  constexpr int kThreads = 64 * WM * WN, kBN = 32 * NJ * WN, TBM = 32 * NI * WM;   // tile = TBM rows x kBN columns
  static_assert((TBM * 8) % kThreads == 0, "whole 16-byte A pieces per thread");
  constexpr int WCH = (4 * kBN + kThreads - 1) / kThreads;             // 16-byte weight chunks per plane and thread (the last
                                                                       // round is partial when 4 kBN % kThreads != 0: 96 x 128)
  constexpr int kASlots = TBM * 4, kBSlots = kBN * 4, kStageSlots = 2 * kASlots + 2 * kBSlots;
  constexpr int NA = TBM * 8 / kThreads;                               // A loads (16 bytes) per thread and stage
  __shared__ uint4 smem[2 * kStageSlots];

  int tm, tn;
  if (p.xcd_cols) {
    // column tile c (< 8 floor(tiles_n / 8)) on XCD c % 8 (= workgroup id % 8): with 64-wide tiles the (b, m) slabs of
    // `value` are written from the XCD whose L2 the sampling kernel reads them from (its workgroups of head m sit on XCD
    // m: ids B M apart, M = 8).  The remaining tiles_n % 8 column tiles follow behind, in any order.
    const int per = p.tiles_n >> 3, rem = p.tiles_n & 7, bid = (int)blockIdx.x, na = 8 * per * p.tiles_m;
    if (bid < na) {
      tn = (bid & 7) + 8 * ((bid >> 3) % per);
      tm = (bid >> 3) / per;
    } else {
      const int t = bid - na;
      if (t >= rem * p.tiles_m) return;
      tn = 8 * per + t % rem;
      tm = t / rem;
    }
  } else if (!tile_of((int)blockIdx.x, p.tiles_m, p.tiles_n, tm, tn)) {
    return;
  }
  const int m0 = tm * TBM, n0 = tn * kBN;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = (wave % WM) * (32 * NI), wn = (wave / WM) * (32 * NJ);
  const int R = p.R, N = p.N, K = p.K;

  // the segment of this column tile (workgroup-uniform)
  int si = 0;
#pragma unroll
  for (int s = 1; s < kMaxSeg; ++s)
    if (s < p.nseg && p.seg[s].n_begin <= n0) si = s;
  const gvl_lin_seg sg = p.seg[si];
  const bool addend = HAS_A2 && (sg.flags & GVL_LIN_ADDEND);

  // ---- A path.  One K stage of a row is exactly one 128-byte line (32 floats): wavefront w covers rows 8 NA w .. of the
  // tile with NA loads, load i = rows 8 NA w + 8 i + (lane >> 3), 16-byte piece lane & 7 -- every instruction reads 8 whole
  // lines.  Buffer loads: the descriptor (workgroup-uniform base) sits in scalar registers, the per-thread part is a 32-bit
  // offset fixed for the whole kernel, the K offset of a stage travels in the instruction's scalar offset operand -- no
  // vector register is spent on address arithmetic inside the loop (with 64-bit per-thread pointers the register allocator
  // recycled the destination of an in-flight load as the next address and serialised the prefetch).
  const int apiece = lane & 7, arow0 = wave * (8 * NA) + (lane >> 3);
  typedef uint32_t u4v __attribute__((ext_vector_type(4)));
  // (the base is passed through readfirstlane: hipcc must be able to PROVE the descriptor wave-uniform, or it wraps every
  //  buffer load in a waterfall loop)
  auto rsrc_of = [](const void *ptr) {
    const uint64_t u = (uint64_t)(uintptr_t)ptr;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)u), hi = __builtin_amdgcn_readfirstlane((uint32_t)(u >> 32));
    return __builtin_amdgcn_make_buffer_rsrc((void *)(uintptr_t)(((uint64_t)hi << 32) | lo), 0, 0x7fffffff, 0x00020000);
  };
  // split-K (the vocabulary layer's input gradient: 2208 x 512 outputs, contraction 8518): this workgroup's stage range
  const int kbeg = p.split_stages ? (int)blockIdx.y * p.split_stages : 0;
  const auto a_rs = rsrc_of(p.A + (int64_t)m0 * p.lda + (int64_t)kbeg * kBK);
  const auto a2_rs = rsrc_of((HAS_A2 ? p.A2 : p.A) + (int64_t)kbeg * kBK);
  const auto wh_rs = rsrc_of(p.Wh + (int64_t)n0 * 32 + (int64_t)kbeg * N * 32);   // planes are K-stage-major (plane_off): stage s of
  const auto wl_rs = rsrc_of(p.Wl + (int64_t)n0 * 32 + (int64_t)kbeg * N * 32);   // the tile's rows starts s N 32 halves further
  int a_off[NA], a2_off[NA];
  uint32_t a_dst[NA];
  float a_inv[NA];
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    const int row = arow0 + 8 * i, grow = min(m0 + row, R - 1);
    a_off[i] = (int)(((int64_t)(grow - m0) * p.lda + apiece * 4) * 4);
    a2_off[i] = HAS_A2 ? (int)(((int64_t)(grow % p.a2_rows) * p.lda2 + apiece * 4) * 4) : 0;
    float s_;
    scale_of(sg.amax_in[grow], s_, a_inv[i]);
    a_dst[i] = (uint32_t)lds_slot(row, apiece >> 1) * 16u + (uint32_t)(apiece & 1) * 8u;     // byte offset inside a plane image
  }

  // ---- W path: chunk id (= thread + round x threads) is piece id & 3 of weight row id >> 2 of the tile, both planes (same
  // swizzled image); ids beyond the tile's 4 kBN chunks (a partial last round) load a valid address and store nothing
  int w_off[WCH], w_dst[WCH];
#pragma unroll
  for (int c = 0; c < WCH; ++c) {
    const int id = tid + c * kThreads, wrow = (id >> 2) % kBN, wch = id & 3;
    w_off[c] = (min(wrow, N - 1 - n0) * 32 + wch * 8) * 2;
    w_dst[c] = id < 4 * kBN ? 2 * kASlots + lds_slot(wrow, wch) : -1;
  }

  struct ASet { u4v x[NA], y[NA], wh[WCH], wl[WCH]; };
  auto load_a = [&](ASet &s, int k0) {
#pragma unroll
    for (int i = 0; i < NA; ++i) s.x[i] = __builtin_amdgcn_raw_buffer_load_b128(a_rs, a_off[i], k0 * 4, 0);
    if (HAS_A2) {
#pragma unroll
      for (int i = 0; i < NA; ++i) s.y[i] = __builtin_amdgcn_raw_buffer_load_b128(a2_rs, a2_off[i], k0 * 4, 0);
    }
#pragma unroll
    for (int c = 0; c < WCH; ++c) {
      s.wh[c] = __builtin_amdgcn_raw_buffer_load_b128(wh_rs, w_off[c], k0 * N * 2, 0);
      if constexpr (!X1) s.wl[c] = __builtin_amdgcn_raw_buffer_load_b128(wl_rs, w_off[c], k0 * N * 2, 0);
    }
  };
  auto store_a = [&](ASet &s, int buf) {
    char *st = reinterpret_cast<char *>(smem + buf * kStageSlots);
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      float4 x = __builtin_bit_cast(float4, s.x[i]);
      if (HAS_A2 && addend) {
        const float4 y = __builtin_bit_cast(float4, s.y[i]);
        x.x += y.x; x.y += y.y; x.z += y.z; x.w += y.w;
      }
      uint2 hi, lo;
      if constexpr (X1) {
        hi = make_uint2(pack2((_Float16)(x.x * a_inv[i]), (_Float16)(x.y * a_inv[i])),
                        pack2((_Float16)(x.z * a_inv[i]), (_Float16)(x.w * a_inv[i])));
      } else {
        split4(x, a_inv[i], hi, lo);
        *reinterpret_cast<uint2 *>(st + kASlots * 16 + a_dst[i]) = lo;
      }
      *reinterpret_cast<uint2 *>(st + a_dst[i]) = hi;
    }
#pragma unroll
    for (int c = 0; c < WCH; ++c)
      if ((4 * kBN) % kThreads == 0 || w_dst[c] >= 0) {
        reinterpret_cast<uint4 *>(st)[w_dst[c]] = __builtin_bit_cast(uint4, s.wh[c]);
        if constexpr (!X1) reinterpret_cast<uint4 *>(st)[kBSlots + w_dst[c]] = __builtin_bit_cast(uint4, s.wl[c]);
      }
  };

  f16acc acc_m[NI][NJ], acc_x[NI][NJ];
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) { acc_m[i][j][r] = 0.f; acc_x[i][j][r] = 0.f; }

  const int frow = lane & 31, fh = lane >> 5;
  int fa[NI][2], fb[NJ][2];                                            // [tile][s]
#pragma unroll
  for (int s = 0; s < 2; ++s) {
#pragma unroll
    for (int i = 0; i < NI; ++i) fa[i][s] = lds_slot(wm + 32 * i + frow, 2 * s + fh);
#pragma unroll
    for (int j = 0; j < NJ; ++j) fb[j][s] = 2 * kASlots + lds_slot(wn + 32 * j + frow, 2 * s + fh);
  }

  const int KT = p.split_stages ? min(K / kBK - kbeg, p.split_stages) : K / kBK;
#ifdef GVL_LIN_STAMPS                                                  /* dev: 10 ns ticks of workgroup 0's phases */
  const uint64_t ts0 = __builtin_amdgcn_s_memrealtime();
#endif
  ASet set0, set1;
  load_a(set0, 0);
  load_a(set1, min(1, KT - 1) * kBK);
  store_a(set0, 0);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");

  // one K stage: `cur` holds the A rows of stage kt + 1 (requested one stage ago), `nxt` receives those of stage kt + 2
#define GVL_LIN_STAGE(cur, nxt)                                                                         \
  {                                                                                                      \
    const int buf = kt & 1;                                                                              \
    const uint4 *st = smem + buf * kStageSlots;                                                          \
    load_a(nxt, min(kt + 2, KT - 1) * kBK);                                                              \
    __builtin_amdgcn_sched_barrier(0);       /* the requests leave FIRST: left alone, the scheduler sinks them below */ \
                                             /* the split arithmetic and the rows arrive a stage late               */ \
    h8 f_ah[2][NI], f_al[2][NI], f_bh[2][NJ], f_bl[2][NJ];                                               \
    _Pragma("unroll") for (int s = 0; s < 2; ++s) {                                                      \
      _Pragma("unroll") for (int i = 0; i < NI; ++i) {                                                   \
        f_ah[s][i] = *reinterpret_cast<const h8 *>(&st[fa[i][s]]);                                       \
        if constexpr (!X1) f_al[s][i] = *reinterpret_cast<const h8 *>(&st[kASlots + fa[i][s]]);          \
      }                                                                                                  \
      _Pragma("unroll") for (int j = 0; j < NJ; ++j) {                                                   \
        f_bh[s][j] = *reinterpret_cast<const h8 *>(&st[fb[j][s]]);                                       \
        if constexpr (!X1) f_bl[s][j] = *reinterpret_cast<const h8 *>(&st[kBSlots + fb[j][s]]);          \
      }                                                                                                  \
    }                                                                                                    \
    store_a(cur, buf ^ 1);                                                                               \
    _Pragma("unroll") for (int s = 0; s < 2; ++s) _Pragma("unroll") for (int i = 0; i < NI; ++i)         \
    _Pragma("unroll") for (int j = 0; j < NJ; ++j) {                                                     \
      acc_m[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f_ah[s][i], f_bh[s][j], acc_m[i][j], 0, 0, 0); \
      if constexpr (!X1) {                                                                               \
        acc_x[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f_ah[s][i], f_bl[s][j], acc_x[i][j], 0, 0, 0); \
        acc_x[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f_al[s][i], f_bh[s][j], acc_x[i][j], 0, 0, 0); \
      }                                                                                                  \
    }                                                                                                    \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                   \
    __builtin_amdgcn_s_barrier();                                                                        \
    asm volatile("" ::: "memory");                                                                       \
    ++kt;                                                                                                \
  }
  // Two stages per iteration and NO exit between them: with a conditional exit after the first stage the loop header has
  // a predecessor on which that stage's own loads are still pending, and hipcc's wait-count pass then drains the prefetch
  // at the top of every iteration (the temporaries there reuse the registers of the set loaded later in the stage).
  int kt = 0;
#ifdef GVL_LIN_STAMPS
  const uint64_t ts1 = __builtin_amdgcn_s_memrealtime();
#endif
  for (int it = KT >> 1; it > 0; --it) {
    GVL_LIN_STAGE(set1, set0)
    GVL_LIN_STAGE(set0, set1)
  }
  if (KT & 1) GVL_LIN_STAGE(set1, set0)
#undef GVL_LIN_STAGE
#ifdef GVL_LIN_STAMPS
  const uint64_t ts2 = __builtin_amdgcn_s_memrealtime();
#endif

  // ---- epilogue.  C/D map of the 32 x 32 MFMA: column = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5).
  // Every load (row scales, residual, mask) is issued before the first store: a load between two stores makes the
  // compiler wait for all stores issued so far.
  const int row0 = m0 + wm;
  const bool relu = sg.flags & GVL_LIN_RELU;
  int ocol[NJ];
  bool st_ok[NJ];
  float cs[NJ], cb[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int col = n0 + wn + 32 * j + frow, colc = min(col, N - 1);
    ocol[j] = colc - sg.n_begin;
    st_ok[j] = col < N && (sg.width <= 0 || ocol[j] < sg.width);
    cs[j] = p.Ws[colc];
    cb[j] = p.bias ? p.bias[colc] : 0.f;
  }
  float rsc[NI][16];
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = min(row0 + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * fh, R - 1);
      float inv_;
      scale_of(sg.amax_in[row], rsc[i][r], inv_);
    }
  float v[NI][NJ][16];
  if (sg.resid) {
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
      for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = min(row0 + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * fh, R - 1);
          v[i][j][r] = st_ok[j] ? sg.resid[(int64_t)row * sg.ldr + ocol[j]] : 0.f;
        }
  }
  unsigned keep[NI];
#pragma unroll
  for (int i = 0; i < NI; ++i) keep[i] = 0xffffu;
  if (sg.rowmask) {
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = min(row0 + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * fh, R - 1);
        if (sg.rowmask[row]) keep[i] &= ~(1u << r);
      }
  }
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float o = (acc_m[i][j][r] + acc_x[i][j][r] * kLoInv) * (rsc[i][r] * cs[j]) + cb[j];
        if (relu) o = fmaxf(o, 0.f);
        if (sg.resid) o = v[i][j][r] + o;
        if (!((keep[i] >> r) & 1u)) o = 0.f;
        v[i][j][r] = o;
      }
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = row0 + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * fh;
        if (st_ok[j] && row < R) (sg.out + (int64_t)blockIdx.y * p.split_stride)[(int64_t)row * sg.ldo + ocol[j]] = v[i][j][r];
      }
  if (sg.amax_out) {
    // Row maxima of this wavefront's part: a butterfly reduce-SCATTER over the 32 lanes that hold one row's columns --
    // each exchange halves the registers a lane keeps -- leaves ONE row per lane pair: register (lane >> 1) & 15 (and, with
    // two 32-row blocks, block lane & 1).  16 exchanges per block instead of 80; then one atomic max per row (non-negative
    // floats order like their bit patterns; a NaN is larger than everything and survives).
    float one[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      float m[16], a8[8], a4[4], a2[2];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        m[r] = 0.f;
#pragma unroll
        for (int j = 0; j < NJ; ++j) m[r] = fmaxf(m[r], st_ok[j] ? fabsf(v[i][j][r]) : 0.f);
      }
      const bool b4 = lane & 16, b3 = lane & 8, b2 = lane & 4, b1 = lane & 2;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const float mine = b4 ? m[k + 8] : m[k], send = b4 ? m[k] : m[k + 8];
        a8[k] = fmaxf(mine, __shfl_xor(send, 16, 64));
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float mine = b3 ? a8[k + 4] : a8[k], send = b3 ? a8[k] : a8[k + 4];
        a4[k] = fmaxf(mine, __shfl_xor(send, 8, 64));
      }
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const float mine = b2 ? a4[k + 2] : a4[k], send = b2 ? a4[k] : a4[k + 2];
        a2[k] = fmaxf(mine, __shfl_xor(send, 4, 64));
      }
      const float mine = b1 ? a2[1] : a2[0], send = b1 ? a2[0] : a2[1];
      one[i] = fmaxf(mine, __shfl_xor(send, 2, 64));
    }
    const bool b0 = lane & 1;
    const int rr = (lane >> 1) & 15;
    if constexpr (NI == 2) {
      const float mine = b0 ? one[1] : one[0], send = b0 ? one[0] : one[1];
      const float rmax = fmaxf(mine, __shfl_xor(send, 1, 64));
      const int row = row0 + 32 * (int)b0 + (rr & 3) + 8 * (rr >> 2) + 4 * fh;
      if (row < R) atomicMax(reinterpret_cast<unsigned *>(sg.amax_out) + row, __float_as_uint(rmax));
    } else {
      const float rmax = fmaxf(one[0], __shfl_xor(one[0], 1, 64));
      const int row = row0 + (rr & 3) + 8 * (rr >> 2) + 4 * fh;
      if (!b0 && row < R) atomicMax(reinterpret_cast<unsigned *>(sg.amax_out) + row, __float_as_uint(rmax));
    }
  }
#ifdef GVL_LIN_STAMPS
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (blockIdx.x == 8 && tid == 0)
    printf("k_lin %dx%dx%d tile %dx%d: prologue %d, loop %d (%d stages), epilogue %d ticks of 10 ns\n", R, K, N, TBM, kBN,
           (int)(ts1 - ts0), (int)(ts2 - ts1), KT, (int)(__builtin_amdgcn_s_memrealtime() - ts2));
#endif
}

// ---------------------------------------------------------------------------------------------------------------------
// LayerNorm over the last axis, one wavefront per row (C <= 1024, C % 4 == 0): y = (x - mean) / sqrt(var + eps) * gamma
// + beta (biased variance, as torch.nn.LayerNorm), two passes over registers.  Also the row maxima the consumers of y
// need: amax_y[r] = max |y|, amax_yp[r] = max |y + pos[r % pos_rows]| (the query of the next attention is y + pos).
constexpr int kLnMaxV = 4;                                             // float4 per lane: C <= 1024

__global__ void __launch_bounds__(256) k_ln_rows(const float *__restrict__ x, int R, int C, const float *__restrict__ gamma,
                                                 const float *__restrict__ beta, float eps, const float *__restrict__ pos,
                                                 int pos_rows, float *__restrict__ y, float *__restrict__ amax_y,
                                                 float *__restrict__ amax_yp, int xcd_rows) {
  // xcd_rows: workgroup b serves the row group (b % 8) (groups / 8) + b / 8 -- XCD x (= b % 8) writes the x-th contiguous eighth of
  // the rows, the rows the product that reads y next stages from the SAME XCD (tile_of: XCD x walks a contiguous tile range)
  const int nb = (R + 3) >> 2, per = (nb + 7) >> 3;
  const int vb = xcd_rows ? (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
  const int row = vb * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (vb >= nb || row >= R) return;
  const int n4 = C >> 2;
  const float4 *xr = reinterpret_cast<const float4 *>(x + (int64_t)row * C);
  float4 v[kLnMaxV];
  float sum = 0.f;
#pragma unroll
  for (int k = 0; k < kLnMaxV; ++k) {
    const int i = lane + 64 * k;
    v[k] = i < n4 ? xr[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    sum += (v[k].x + v[k].y) + (v[k].z + v[k].w);
  }
#pragma unroll
  for (int o = 32; o; o >>= 1) sum += __shfl_xor(sum, o, 64);
  const float mean = sum / (float)C;
  float sq = 0.f;
#pragma unroll
  for (int k = 0; k < kLnMaxV; ++k) {
    if (lane + 64 * k < n4) {
      const float a = v[k].x - mean, b = v[k].y - mean, c = v[k].z - mean, d = v[k].w - mean;
      sq += (a * a + b * b) + (c * c + d * d);
    }
  }
#pragma unroll
  for (int o = 32; o; o >>= 1) sq += __shfl_xor(sq, o, 64);
  const float rstd = 1.f / sqrtf(sq / (float)C + eps);
  const float4 *g4 = reinterpret_cast<const float4 *>(gamma), *b4 = reinterpret_cast<const float4 *>(beta);
  const float4 *p4 = pos ? reinterpret_cast<const float4 *>(pos + (int64_t)(row % pos_rows) * C) : nullptr;
  float4 *yr = reinterpret_cast<float4 *>(y + (int64_t)row * C);
  float m0 = 0.f, m1 = 0.f;
#pragma unroll
  for (int k = 0; k < kLnMaxV; ++k) {
    const int i = lane + 64 * k;
    if (i < n4) {
      const float4 g = g4[i], b = b4[i];
      float4 o;
      o.x = (v[k].x - mean) * rstd * g.x + b.x;
      o.y = (v[k].y - mean) * rstd * g.y + b.y;
      o.z = (v[k].z - mean) * rstd * g.z + b.z;
      o.w = (v[k].w - mean) * rstd * g.w + b.w;
      yr[i] = o;
      m0 = fmaxf(fmaxf(m0, fmaxf(fabsf(o.x), fabsf(o.y))), fmaxf(fabsf(o.z), fabsf(o.w)));
      if (p4) {
        const float4 q = p4[i];
        m1 = fmaxf(fmaxf(m1, fmaxf(fabsf(o.x + q.x), fabsf(o.y + q.y))), fmaxf(fabsf(o.z + q.z), fabsf(o.w + q.w)));
      }
    }
  }
#pragma unroll
  for (int o = 32; o; o >>= 1) {
    m0 = fmaxf(m0, __shfl_xor(m0, o, 64));
    m1 = fmaxf(m1, __shfl_xor(m1, o, 64));
  }
  if (lane == 0) {
    if (amax_y) amax_y[row] = m0;
    if (amax_yp) amax_yp[row] = m1;
  }
}

// row maxima of x (R, C) [+ pos]: what k_lin_f16x3 needs for an A operand some other kernel produced
__global__ void __launch_bounds__(256) k_row_absmax(const float *__restrict__ x, int64_t ldx, int R, int C,
                                                    const float *__restrict__ pos, int64_t ldp, int pos_rows,
                                                    float *__restrict__ amax_x, float *__restrict__ amax_xp) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= R) return;
  const int n4 = C >> 2;
  const float4 *xr = reinterpret_cast<const float4 *>(x + (int64_t)row * ldx);
  const float4 *p4 = pos ? reinterpret_cast<const float4 *>(pos + (int64_t)(row % pos_rows) * ldp) : nullptr;
  float m0 = 0.f, m1 = 0.f;
  for (int i = lane; i < n4; i += 64) {
    const float4 o = xr[i];
    m0 = fmaxf(fmaxf(m0, fmaxf(fabsf(o.x), fabsf(o.y))), fmaxf(fabsf(o.z), fabsf(o.w)));
    if (p4) {
      const float4 q = p4[i];
      m1 = fmaxf(fmaxf(m1, fmaxf(fabsf(o.x + q.x), fabsf(o.y + q.y))), fmaxf(fabsf(o.z + q.z), fabsf(o.w + q.w)));
    }
  }
#pragma unroll
  for (int o = 32; o; o >>= 1) {
    m0 = fmaxf(m0, __shfl_xor(m0, o, 64));
    m1 = fmaxf(m1, __shfl_xor(m1, o, 64));
  }
  if (lane == 0) {
    if (amax_x) amax_x[row] = m0;
    if (amax_xp) amax_xp[row] = m1;
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Attention core of the decoder's nn.MultiheadAttention (deformable_transformer.py:266-270; head dimension 64, <= 320
// queries): softmax(q k^T / 8 + key mask) v for one (video, head, block of 64 queries) per workgroup, in exact fp32 on
// v_mfma_f32_16x16x4_f32.  A wavefront owns 16 queries and keeps its whole score strip (16 x Q) in registers:
//   scores   S^T = K Q^T, one 16-key tile at a time: the C/D layout leaves key 16 kt + 4 (lane >> 4) + r, query lane & 15 in
//            register r -- exactly the B-operand layout of the second product when its MFMA steps walk the keys in the order
//            (4 g + j), so P never moves between lanes or through LDS;
//   output   O^T = V^T P^T with the channel order inside an MFMA tile chosen as 4 (lane & 15) + tile, so that a lane reads
//            four consecutive channels of a V row as ONE 16-byte load and stores four consecutive output channels as one.
// K and V pass through LDS one 16-key tile at a time (4 KB, double-buffered, the next tile requested before the current
// one is multiplied), laid out so that the operand reads are conflict-free 16-byte reads.
// Also leaves max |out row| per (row, head) in amax (atomic max; zero-initialised by the caller) for out_proj's split.
typedef float f4acc __attribute__((ext_vector_type(4)));
constexpr int kMhaMaxTiles = 20;

__global__ void __launch_bounds__(256) k_mha_core(const float *__restrict__ qkv, int64_t ld, const unsigned char *__restrict__ keep,
                                                  int B, int Q, int H, float *__restrict__ out, float *__restrict__ amax) {
  // one 16-key tile of K (then of V) at a time through LDS, shared by the workgroup's four wavefronts (each reading all of
  // K and V for itself cost 394 MB of L2 reads per launch: 70 us); two buffers, one barrier per tile
  __shared__ float4 tile[2][256];
  const int nqb = (Q + 63) >> 6;
  const int qb = blockIdx.x % nqb, h = (blockIdx.x / nqb) % H, b = blockIdx.x / (nqb * H);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, qi = lane & 15, g = lane >> 4;
  const int q0 = qb * 64 + wave * 16;
  const bool active = q0 < Q;                                          // (wavefront-uniform; idle wavefronts still stage)
  const int C = H * 64, KT = (Q + 15) >> 4;
  const float *base = qkv + (int64_t)b * Q * ld + h * 64;
  // staging: thread t carries the 16-byte piece t & 15 of key row t >> 4 of the tile (a wavefront = 4 whole 256-byte rows)
  const int srow = tid >> 4, spc = tid & 15;
  const float *kstage = base + C + 4 * spc, *vstage = base + 2 * C + 4 * spc;
  const int k_slot = srow * 16 + (spc ^ srow);                         // K: 16-byte units XOR-swizzled by the key
  float4 qreg[4];
  {
    const float *qp = base + (int64_t)min(q0 + qi, Q - 1) * ld + 4 * g;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      qreg[t] = *reinterpret_cast<const float4 *>(qp + 16 * t);
      qreg[t].x *= 0.125f; qreg[t].y *= 0.125f; qreg[t].z *= 0.125f; qreg[t].w *= 0.125f;
    }
  }
  f4acc s[kMhaMaxTiles];
  float4 st = *reinterpret_cast<const float4 *>(kstage + (int64_t)min(srow, Q - 1) * ld);
#pragma unroll
  for (int kt = 0; kt < kMhaMaxTiles; ++kt) {
    if (kt < KT) {
      tile[kt & 1][k_slot] = st;
      if (kt + 1 < KT) st = *reinterpret_cast<const float4 *>(kstage + (int64_t)min(16 * (kt + 1) + srow, Q - 1) * ld);
      __syncthreads();
      f4acc acc = {0.f, 0.f, 0.f, 0.f};
      if (active) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const float4 kc = tile[kt & 1][qi * 16 + ((4 * t + g) ^ qi)];   // K[key qi of the tile][16 t + 4 g ..]
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(kc.x, qreg[t].x, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(kc.y, qreg[t].y, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(kc.z, qreg[t].z, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(kc.w, qreg[t].w, acc, 0, 0, 0);
        }
      }
      // key mask: padded keys (beyond Q) and keys the caller excludes (key_padding_mask) get -inf
      const int key0 = 16 * kt + 4 * g;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int key = key0 + r;
        const bool ok = key < Q && (!keep || keep[(int64_t)b * Q + min(key, Q - 1)]);
        acc[r] = ok ? acc[r] : -INFINITY;
      }
      s[kt] = acc;
    }
  }
  // first V tile on its way while the softmax runs
  st = *reinterpret_cast<const float4 *>(vstage + (int64_t)min(srow, Q - 1) * ld);
  // softmax over the keys of query lane & 15: registers, then the four lane groups
  float m = -INFINITY;
#pragma unroll
  for (int kt = 0; kt < kMhaMaxTiles; ++kt)
    if (kt < KT) m = fmaxf(fmaxf(m, fmaxf(s[kt][0], s[kt][1])), fmaxf(s[kt][2], s[kt][3]));
  m = fmaxf(m, __shfl_xor(m, 16, 64));
  m = fmaxf(m, __shfl_xor(m, 32, 64));
  float sum = 0.f;
#pragma unroll
  for (int kt = 0; kt < kMhaMaxTiles; ++kt)
    if (kt < KT) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float e = __expf(s[kt][r] - m);
        s[kt][r] = e;
        sum += e;
      }
    }
  sum += __shfl_xor(sum, 16, 64);
  sum += __shfl_xor(sum, 32, 64);
  // O^T = V^T P^T
  f4acc o[4];
#pragma unroll
  for (int ct = 0; ct < 4; ++ct) o[ct] = f4acc{0.f, 0.f, 0.f, 0.f};
  // (KT and KT + 1 have different parity bookkeeping: the V tiles simply continue the buffer alternation after a barrier)
  __syncthreads();
#pragma unroll
  for (int kt = 0; kt < kMhaMaxTiles; ++kt) {
    if (kt < KT) {
      tile[kt & 1][tid] = st;                                          // V: linear image, row = key of the tile
      if (kt + 1 < KT) st = *reinterpret_cast<const float4 *>(vstage + (int64_t)min(16 * (kt + 1) + srow, Q - 1) * ld);
      __syncthreads();
      if (active) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float4 vc = tile[kt & 1][(4 * g + j) * 16 + qi];       // V[key 4 g + j of the tile][4 qi ..]
          o[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(vc.x, s[kt][j], o[0], 0, 0, 0);
          o[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(vc.y, s[kt][j], o[1], 0, 0, 0);
          o[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(vc.z, s[kt][j], o[2], 0, 0, 0);
          o[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(vc.w, s[kt][j], o[3], 0, 0, 0);
        }
      }
    }
  }
  const float inv = 1.f / sum;
  float mx = 0.f;
#pragma unroll
  for (int ct = 0; ct < 4; ++ct)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      o[ct][r] *= inv;
      mx = fmaxf(mx, fabsf(o[ct][r]));
    }
  mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
  mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
  const int q = q0 + qi;
  if (active && q < Q) {
    float *op = out + ((int64_t)b * Q + q) * C + h * 64 + 16 * g;       // register r: channels 16 g + 4 r .. + 3
#pragma unroll
    for (int r = 0; r < 4; ++r) *reinterpret_cast<float4 *>(op + 4 * r) = make_float4(o[0][r], o[1][r], o[2][r], o[3][r]);
    if (amax && g == 0) atomicMax(reinterpret_cast<unsigned *>(amax) + (int64_t)b * Q + q, __float_as_uint(mx));
  }
}

// valid ratios + encoder reference points of one video per workgroup (deformable_transformer.py:81-83, 209-218)
struct LevelDims { int len[8], start[8]; };
__global__ void __launch_bounds__(256) k_encoder_geometry(const unsigned char *__restrict__ mask, int S, int L, LevelDims d,
                                                          float *__restrict__ vr_out, float *__restrict__ ref) {
  __shared__ int cnt[8];
  __shared__ float vr[8];
  const int b = blockIdx.x;
  if (threadIdx.x < 8) cnt[threadIdx.x] = 0;
  __syncthreads();
  for (int l = 0; l < L; ++l) {
    int c = 0;
    for (int t = threadIdx.x; t < d.len[l]; t += blockDim.x) c += mask[(int64_t)b * S + d.start[l] + t] ? 0 : 1;
#pragma unroll
    for (int o = 32; o; o >>= 1) c += __shfl_xor(c, o, 64);
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(&cnt[l], c);
  }
  __syncthreads();
  if (threadIdx.x < L) {
    // (count * (1 / T), not count / T: PyTorch divides a tensor by a scalar through the reciprocal -- same bits as :81-83)
    const float v = (float)cnt[threadIdx.x] * (1.0f / (float)d.len[threadIdx.x]);
    vr[threadIdx.x] = v;
    vr_out[b * L + threadIdx.x] = v;
  }
  __syncthreads();
  if (!ref) return;
  for (int l = 0; l < L; ++l) {
    const float den = vr[l] * (float)d.len[l];
    for (int t = threadIdx.x; t < d.len[l]; t += blockDim.x) {
      const float c = ((float)t + 0.5f) / den;
      float *o = ref + ((int64_t)b * S + d.start[l] + t) * L;
      for (int l2 = 0; l2 < L; ++l2) o[l2] = c * vr[l2];
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Feature pyramid of the base encoder, inference (pdvc/base_encoder.py:55-82, pdvc/position_encoding.py:38-64,
// pdvc/deformable_transformer.py:85-115): the conv1d of every level is a gvl_linear_f16x3_f32 product (k = 3, stride 2
// reads its three taps as ONE row of a strided view of the zero-padded input); what remains around it is
//   k_group_norm_rows    GroupNorm of a level's conv output, written straight into the level's rows of the flattened
//                        (B, S, C) encoder input -- and, zero-padded, into the next level's conv input;
//   k_pyramid_geometry   every level's padding mask (nearest-neighbour resampling of the frame mask), sine position
//                        embedding + duration embedding + level embedding, flattened: mask (B, S), lvl_pos (B, S, C).
// ~60 PyTorch launches per forward otherwise (transposes, cats, interpolate, cumsum, sin / cos, group-norm passes).

// one WORKGROUP (4 wavefronts) per (video, group): rows y[(n rows_per_video + t) ldy + c], t < T, c in the group's cg channels; a
// thread owns channel c = threadIdx % cg of the rows t = threadIdx / cg (mod 256 / cg), four rows' loads in flight per pass.  (Round 6:
// one wavefront per unit walked a level-0 group of a 512-frame video in 128 dependent steps per pass -- 225 us for 16 MB.)
__device__ __forceinline__ float gn_wave_sum(float v) {
#pragma unroll
  for (int o = 32; o; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
// sums of a and b over the workgroup's 256 threads (sh: 8 floats), every thread gets both
__device__ __forceinline__ void gn_block_sum2(float &a, float &b, float *sh) {
  a = gn_wave_sum(a);
  b = gn_wave_sum(b);
  const int wave = threadIdx.x >> 6;
  __syncthreads();                                                      // (the previous use of sh is over)
  if ((threadIdx.x & 63) == 0) {
    sh[wave] = a;
    sh[4 + wave] = b;
  }
  __syncthreads();
  a = (sh[0] + sh[1]) + (sh[2] + sh[3]);
  b = (sh[4] + sh[5]) + (sh[6] + sh[7]);
}
// mean and 1 / sqrt(var + eps) of the unit's T x cg values (two passes: the variance from the centred values)
__device__ __forceinline__ void gn_stats(const float *__restrict__ src, int64_t ldy, int T, int tr, int tstep, float cnt, float eps,
                                         float *sh, float &mean, float &rstd) {
  float sum = 0.f, zero = 0.f;
  int t = tr;
  for (; t + 3 * tstep < T; t += 4 * tstep) {
    const float a0 = src[(int64_t)t * ldy], a1 = src[(int64_t)(t + tstep) * ldy], a2 = src[(int64_t)(t + 2 * tstep) * ldy],
                a3 = src[(int64_t)(t + 3 * tstep) * ldy];
    sum += (a0 + a1) + (a2 + a3);
  }
  for (; t < T; t += tstep) sum += src[(int64_t)t * ldy];
  gn_block_sum2(sum, zero, sh);
  mean = sum / cnt;
  float sq = 0.f;
  t = tr;
  for (; t + 3 * tstep < T; t += 4 * tstep) {
    const float d0 = src[(int64_t)t * ldy] - mean, d1 = src[(int64_t)(t + tstep) * ldy] - mean,
                d2 = src[(int64_t)(t + 2 * tstep) * ldy] - mean, d3 = src[(int64_t)(t + 3 * tstep) * ldy] - mean;
    sq += fmaf(d0, d0, d1 * d1) + fmaf(d2, d2, d3 * d3);
  }
  for (; t < T; t += tstep) {
    const float d = src[(int64_t)t * ldy] - mean;
    sq = fmaf(d, d, sq);
  }
  zero = 0.f;
  gn_block_sum2(sq, zero, sh);
  rstd = 1.f / sqrtf(sq / cnt + eps);
}

__global__ void __launch_bounds__(256) k_group_norm_rows(const float *__restrict__ y, int64_t ldy, int rows_per_video, int T,
                                                         int C, int G, int N, const float *__restrict__ gamma,
                                                         const float *__restrict__ beta, float eps, float *__restrict__ dst,
                                                         int64_t dst_vs, float *__restrict__ dst2, int64_t dst2_vs) {
  __shared__ float sh[8];
  const int unit = blockIdx.x, n = unit / G, g = unit % G, cg = C / G;
  const int c = g * cg + (int)threadIdx.x % cg, tr = (int)threadIdx.x / cg, tstep = 256 / cg;
  const float *src = y + (int64_t)n * rows_per_video * ldy + c;
  const float cnt = (float)T * (float)cg;
  float mean, rstd;
  gn_stats(src, ldy, T, tr, tstep, cnt, eps, sh, mean, rstd);
  const float a = rstd * gamma[c], b2 = beta[c] - mean * rstd * gamma[c];
  float *d1 = dst + (int64_t)n * dst_vs + c;
  float *d2 = dst2 ? dst2 + (int64_t)n * dst2_vs + c : nullptr;
  int t = tr;
  for (; t + tstep < T; t += 2 * tstep) {
    const float v0 = fmaf(src[(int64_t)t * ldy], a, b2), v1 = fmaf(src[(int64_t)(t + tstep) * ldy], a, b2);
    d1[(int64_t)t * C] = v0;
    d1[(int64_t)(t + tstep) * C] = v1;
    if (d2) {
      d2[(int64_t)t * C] = v0;
      d2[(int64_t)(t + tstep) * C] = v1;
    }
  }
  for (; t < T; t += tstep) {
    const float v = fmaf(src[(int64_t)t * ldy], a, b2);
    d1[(int64_t)t * C] = v;
    if (d2) d2[(int64_t)t * C] = v;
  }
}

// backward of k_group_norm_rows (TRAINING: base_encoder.py:60-80's GroupNorm of a level, rows layout), one workgroup per
// (video, group): x^ = (y - mean) rstd, g = dout gamma (dout = the gradient of the level's rows of the flattened encoder input, plus
// the next level's input gradient where there is one), dy = rstd (g - (sum g + x^ sum g x^) / count); per-video partial sums of the
// affine parameters' gradients (dgamma_part, dbeta_part: (N, C), summed over the videos by the caller: fixed order).  dy covers the
// level's rows_per_video rows per video: the rows behind its T frames (the next convolution's padding rows) are zero.
__global__ void __launch_bounds__(256) k_group_norm_rows_bwd(const float *__restrict__ y, int64_t ldy, int rows_per_video, int T,
                                                             int C, int G, int N, const float *__restrict__ gamma, float eps,
                                                             const float *__restrict__ dout, int64_t dout_vs,
                                                             const float *__restrict__ dout2, int64_t dout2_vs,
                                                             float *__restrict__ dy, int64_t ld_dy, float *__restrict__ dgamma_part,
                                                             float *__restrict__ dbeta_part, float *__restrict__ amax_dy) {
  __shared__ float sh[8];
  __shared__ float sh_c[4][2][64];
  const int unit = blockIdx.x, n = unit / G, g = unit % G, cg = C / G;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = g * cg + (int)threadIdx.x % cg, tr = (int)threadIdx.x / cg, tstep = 256 / cg;
  const float *src = y + (int64_t)n * rows_per_video * ldy + c;
  const float *d1 = dout + (int64_t)n * dout_vs + c;
  const float *d2 = dout2 ? dout2 + (int64_t)n * dout2_vs + c : nullptr;
  const float cnt = (float)T * (float)cg;
  float mean, rstd;
  gn_stats(src, ldy, T, tr, tstep, cnt, eps, sh, mean, rstd);
  const float gm = gamma[c];
  float s1 = 0.f, s2 = 0.f, dg = 0.f, db = 0.f;
  int t = tr;
  for (; t + tstep < T; t += 2 * tstep) {
    const float y0 = src[(int64_t)t * ldy], y1 = src[(int64_t)(t + tstep) * ldy];
    float g0 = d1[(int64_t)t * C], g1 = d1[(int64_t)(t + tstep) * C];
    if (d2) {
      g0 += d2[(int64_t)t * C];
      g1 += d2[(int64_t)(t + tstep) * C];
    }
    const float x0 = (y0 - mean) * rstd, x1 = (y1 - mean) * rstd;
    s1 = fmaf(g0, gm, s1); s2 = fmaf(g0 * gm, x0, s2); dg = fmaf(g0, x0, dg); db += g0;
    s1 = fmaf(g1, gm, s1); s2 = fmaf(g1 * gm, x1, s2); dg = fmaf(g1, x1, dg); db += g1;
  }
  for (; t < T; t += tstep) {
    const float xh = (src[(int64_t)t * ldy] - mean) * rstd;
    const float go = d1[(int64_t)t * C] + (d2 ? d2[(int64_t)t * C] : 0.f);
    s1 = fmaf(go, gm, s1);
    s2 = fmaf(go * gm, xh, s2);
    dg = fmaf(go, xh, dg);
    db += go;
  }
  for (int o = cg; o < 64; o <<= 1) {                                  // the channel's rows sit in the lanes cg apart ...
    dg += __shfl_xor(dg, o, 64);
    db += __shfl_xor(db, o, 64);
  }
  if (lane < cg) {                                                     // ... and in the four wavefronts
    sh_c[wave][0][lane] = dg;
    sh_c[wave][1][lane] = db;
  }
  gn_block_sum2(s1, s2, sh);                                           // (its barriers also publish sh_c)
  if ((int)threadIdx.x < cg) {
    dgamma_part[(int64_t)n * C + c] = (sh_c[0][0][lane] + sh_c[1][0][lane]) + (sh_c[2][0][lane] + sh_c[3][0][lane]);
    dbeta_part[(int64_t)n * C + c] = (sh_c[0][1][lane] + sh_c[1][1][lane]) + (sh_c[2][1][lane] + sh_c[3][1][lane]);
  }
  const float k1 = s1 / cnt, k2 = s2 / cnt;
  float *dst = dy + (int64_t)n * rows_per_video * ld_dy + c;
  for (t = tr; t < rows_per_video; t += tstep) {
    float v = 0.f;
    if (t < T) {
      const float xh = (src[(int64_t)t * ldy] - mean) * rstd;
      const float go = d1[(int64_t)t * C] + (d2 ? d2[(int64_t)t * C] : 0.f);
      v = rstd * (go * gm - k1 - xh * k2);
    }
    dst[(int64_t)t * ld_dy] = v;
    if (amax_dy) {
      // max |dy row| (zero-initialised, atomic max over the row's G groups): the row scale of the convolution's weight / input
      // gradient products.  The cg lanes of a row's group are adjacent and leave the loop together: the shuffles below stay within them
      float m = fabsf(v);
      for (int o = 1; o < cg; o <<= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
      if (lane % cg == 0) atomicMax(reinterpret_cast<unsigned *>(amax_dy) + (int64_t)n * rows_per_video + t, __float_as_uint(m));
    }
  }
}

// the input gradient of a Conv1d(k = 3, stride 2, padding 1) from the gradient of its rows of taps: dcols (N, T' + 1, 3, C) --
// row (n, t') = d[x_{2t'-1} | x_{2t'} | x_{2t'+1}] -- summed into dx (N, 2 (T' + 1), C), padded frame index 2 t' + k (frame t of the
// input is row t + 1).  One thread per 4 channels of a padded row; an even row has one contributor, an odd row two.
__global__ void __launch_bounds__(256) k_taps_to_rows(const float4 *__restrict__ dcols, int N, int T1, int C4,
                                                      float4 *__restrict__ dx) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x, total = (int64_t)N * 2 * T1 * C4;
  if (i >= total) return;
  const int c = (int)(i % C4), r = (int)((i / C4) % (2 * T1)), n = (int)(i / ((int64_t)C4 * 2 * T1));
  // padded row r = 2 t' + k: (t', k) = (r / 2, r % 2) and, for even r >= 2, also (r / 2 - 1, 2)
  const float4 *row = dcols + (int64_t)n * T1 * 3 * C4;
  float4 v = row[((int64_t)(r >> 1) * 3 + (r & 1)) * C4 + c];
  if (!(r & 1) && r >= 2) {
    const float4 u = row[((int64_t)((r >> 1) - 1) * 3 + 2) * C4 + c];
    v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
  }
  dx[i] = v;
}

struct PyramidDims { int len[8], start[8]; };

// grid (videos, levels, slices): mask_flat (N, S) bytes, lvl_pos (N, S, F + Cd)
__global__ void __launch_bounds__(256) k_pyramid_geometry(const unsigned char *__restrict__ mask0, int T0, int S, int L,
                                                          PyramidDims d, const float *__restrict__ dim_t,
                                                          const float *__restrict__ dur, const float *__restrict__ level_embed,
                                                          int F, int Cd, float scale, unsigned char *__restrict__ mask_flat,
                                                          float *__restrict__ lvl_pos) {
  extern __shared__ float xs[];                     // normalised position of every frame of this level
  __shared__ int wave_tot[4];
  __shared__ int carry;
  const int n = blockIdx.x, l = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int T = d.len[l], s0 = d.start[l], C = F + Cd;
  // nearest-neighbour resampling of the frame mask (F.interpolate(mask.float(), size=T), base_encoder.py:75)
  const float rs = (float)T0 / (float)T;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (int base = 0; base < T; base += blockDim.x) {
    const int t = base + threadIdx.x;
    int v = 0;
    if (t < T) {
      const int src = l == 0 ? t : min((int)floorf((float)t * rs), T0 - 1);
      const unsigned char m = mask0[(int64_t)n * T0 + src];
      if (blockIdx.z == 0) mask_flat[(int64_t)n * S + s0 + t] = m ? 1 : 0;
      v = m ? 0 : 1;
    }
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
  const float denom = xs[T - 1] + 1e-6f;
  float *o = lvl_pos + ((int64_t)n * S + s0) * C;
  const float *le = level_embed + (int64_t)l * C;
  const int stride = blockDim.x * gridDim.z, first = blockIdx.z * blockDim.x + threadIdx.x;
  for (int idx = first; idx < T * C; idx += stride) {
    const int t = idx / C, c = idx % C;
    float v;
    if (c < F) {
      const float p = (xs[t] - 0.5f) / denom * scale / dim_t[c];
      v = (c & 1) ? cosf(p) : sinf(p);
    } else {
      v = dur[(int64_t)n * Cd + (c - F)];
    }
    o[idx] = v + le[c];
  }
}

// sigmoid(delta + inverse_sigmoid(ref)) and the next layer's scaled reference points (deformable_transformer.py:301-324)
__global__ void __launch_bounds__(256) k_box_refine(const float *__restrict__ delta, int64_t ldd, const float *__restrict__ ref,
                                                    int RD, const float *__restrict__ vr, int R, int Q, int L,
                                                    float *__restrict__ new_ref, float *__restrict__ ref_in) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= R) return;
  constexpr float eps = 1e-5f;
  float o[2];
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    float v = delta[(int64_t)r * ldd + c];
    if (c < RD) {
      const float x = fminf(fmaxf(ref[(int64_t)r * RD + c], 0.f), 1.f);
      v += logf(fmaxf(x, eps) / fmaxf(1.f - x, eps));
    }
    o[c] = 1.f / (1.f + expf(-v));
  }
  new_ref[2 * (int64_t)r] = o[0];
  new_ref[2 * (int64_t)r + 1] = o[1];
  if (ref_in) {
    const int b = r / Q;
    for (int l = 0; l < L; ++l) {
      const float s = vr[b * L + l];
      ref_in[((int64_t)r * L + l) * 2] = o[0] * s;
      ref_in[((int64_t)r * L + l) * 2 + 1] = o[1] * s;
    }
  }
}

// the gradient of k_box_refine's new_ref = sigmoid(delta + inverse_sigmoid(ref)): d delta = g o (1 - o); d ref (the first decoder
// layer's reference points come from a Linear) = d delta[c] / (clamped x) + d delta[c] / (clamped 1 - x) where the clamps pass
__global__ void __launch_bounds__(256) k_box_refine_bwd(const float *__restrict__ g, const float *__restrict__ o,
                                                        const float *__restrict__ ref, int RD, int R, float *__restrict__ gdelta,
                                                        float *__restrict__ gref) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= R) return;
  constexpr float eps = 1e-5f;
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    const float ov = o[2 * (int64_t)r + c];
    const float gd = g[2 * (int64_t)r + c] * ov * (1.f - ov);
    gdelta[2 * (int64_t)r + c] = gd;
    if (gref && c < RD) {
      // prior = log(max(x, eps)) - log(max(1 - x, eps)) with x = clamp(ref, 0, 1): each term differentiates to 1 / value where
      // its clamp is inactive (torch's clamp passes the gradient at the boundary itself: min <= x <= max)
      const float x0 = ref[(int64_t)r * RD + c];
      const float x = fminf(fmaxf(x0, 0.f), 1.f);
      float d = 0.f;
      if (x >= eps) d += 1.f / x;
      if (1.f - x >= eps) d += 1.f / (1.f - x);
      gref[(int64_t)r * RD + c] = (x0 >= 0.f && x0 <= 1.f) ? gd * d : 0.f;
    }
  }
}

// out[b] = W . max_q hs[b][q][:] + bias: one workgroup (16 wavefronts) per video.  Wavefront w pools rows w, w + 16, ...
// (four rows' loads in flight at a time: the pooling is a chain of load latencies, not of bytes), lane = float4 columns
// lane, lane + 64, ...; the wavefronts' partial maxima meet in LDS.
constexpr int kCountWaves = 16;
__global__ void __launch_bounds__(64 * kCountWaves) k_count_head(const float *__restrict__ hs, int Q, int C,
                                                                 const float *__restrict__ W, const float *__restrict__ bias,
                                                                 int n_out, float *__restrict__ out) {
  extern __shared__ float pooled[];                                    // [kCountWaves][C], then row 0 = the pooled vector
  const int b = blockIdx.x, wave = threadIdx.x >> 6, lane = threadIdx.x & 63, n4 = C >> 2;
  const float4 *base = reinterpret_cast<const float4 *>(hs + (int64_t)b * Q * C);
  for (int i = lane; i < n4; i += 64) {
    float4 m = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
    for (int q = wave; q < Q; q += 4 * kCountWaves) {
      float4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int qq = q + u * kCountWaves;
        v[u] = qq < Q ? base[(int64_t)qq * n4 + i] : m;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        m.x = fmaxf(m.x, v[u].x); m.y = fmaxf(m.y, v[u].y); m.z = fmaxf(m.z, v[u].z); m.w = fmaxf(m.w, v[u].w);
      }
    }
    reinterpret_cast<float4 *>(pooled + (int64_t)wave * C)[i] = m;
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    float m = pooled[c];
    for (int w = 1; w < kCountWaves; ++w) m = fmaxf(m, pooled[w * C + c]);
    pooled[c] = m;                                                     // (row 0: each column is touched by one thread only)
  }
  __syncthreads();
  for (int n = wave; n < n_out; n += kCountWaves) {
    float acc = 0.f;
    for (int c = lane; c < C; c += 64) acc = fmaf(pooled[c], W[(int64_t)n * C + c], acc);
#pragma unroll
    for (int o = 32; o; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if (lane == 0) out[(int64_t)b * n_out + n] = acc + (bias ? bias[n] : 0.f);
  }
}

// TRAINING form of the pooling alone: pooled[b][c] = max_q hs[b][q][c] and the FIRST row that attains it.  One workgroup per
// (video, 256 columns): wavefront w pools rows w, w + 16, ... (ascending, strict >: its first maximum), the wavefronts' pairs meet
// in LDS (equal maxima: the smaller row).
__global__ void __launch_bounds__(64 * kCountWaves) k_count_pool(const float *__restrict__ hs, int Q, int C,
                                                                 float *__restrict__ pooled, int *__restrict__ arg) {
  __shared__ float4 s_m[kCountWaves][64];
  __shared__ int4 s_a[kCountWaves][64];
  const int chunks = (C + 255) >> 8;
  const int b = blockIdx.x / chunks, c4 = (blockIdx.x % chunks) * 64 + (threadIdx.x & 63);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, n4 = C >> 2;
  float4 m = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
  int4 a = make_int4(0, 0, 0, 0);
  if (c4 < n4) {
    const float4 *base = reinterpret_cast<const float4 *>(hs + (int64_t)b * Q * C) + c4;
    for (int q = wave; q < Q; q += 4 * kCountWaves) {
      float4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int qq = q + u * kCountWaves;
        v[u] = qq < Q ? base[(int64_t)qq * n4] : make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int qq = q + u * kCountWaves;
        if (v[u].x > m.x) { m.x = v[u].x; a.x = qq; }
        if (v[u].y > m.y) { m.y = v[u].y; a.y = qq; }
        if (v[u].z > m.z) { m.z = v[u].z; a.z = qq; }
        if (v[u].w > m.w) { m.w = v[u].w; a.w = qq; }
      }
    }
  }
  s_m[wave][lane] = m; s_a[wave][lane] = a;
  __syncthreads();
  if (wave == 0 && c4 < n4) {
    for (int w = 1; w < kCountWaves; ++w) {
      const float4 v = s_m[w][lane];
      const int4 i = s_a[w][lane];
      if (v.x > m.x || (v.x == m.x && i.x < a.x)) { m.x = v.x; a.x = i.x; }
      if (v.y > m.y || (v.y == m.y && i.y < a.y)) { m.y = v.y; a.y = i.y; }
      if (v.z > m.z || (v.z == m.z && i.z < a.z)) { m.z = v.z; a.z = i.z; }
      if (v.w > m.w || (v.w == m.w && i.w < a.w)) { m.w = v.w; a.w = i.w; }
    }
    reinterpret_cast<float4 *>(pooled + (int64_t)b * C)[c4] = m;
    reinterpret_cast<int4 *>(arg + (int64_t)b * C)[c4] = a;
  }
}

// its gradient, dense: dx[b][q][c] = g[b][c] where q == arg[b][c], 0 elsewhere (one float4 per thread) -- plus, when given, the
// input gradient of a ONE-output Linear on the same rows (the class head: g_row[b][q] w_row[c]), so that the two heads' gradients
// reach hs as one tensor
__global__ void __launch_bounds__(256) k_count_pool_bwd(const float4 *__restrict__ g, const int4 *__restrict__ arg, int Q, int n4,
                                                        int64_t total4, const float *__restrict__ g_row,
                                                        const float4 *__restrict__ w_row, float4 *__restrict__ dx) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= total4) return;
  const int c4 = (int)(i % n4);
  const int64_t r = i / n4;
  const int q = (int)(r % Q);
  const int64_t bc = (r / Q) * n4 + c4;
  float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
  if (g) {
    const float4 v = g[bc];
    const int4 a = arg[bc];
    o = make_float4(a.x == q ? v.x : 0.f, a.y == q ? v.y : 0.f, a.z == q ? v.z : 0.f, a.w == q ? v.w : 0.f);
  }
  if (g_row) {
    const float s = g_row[r];
    const float4 w = w_row[c4];
    o.x = fmaf(s, w.x, o.x); o.y = fmaf(s, w.y, o.y); o.z = fmaf(s, w.z, o.z); o.w = fmaf(s, w.w, o.w);
  }
  dx[i] = o;
}

// out[q][h * C + c] = sum_b g_h[b][q][c] for up to four (B, Q, C) gradients g_h: the gradient of a (Q, parts * C) embedding whose
// column blocks were batch-expanded (one float4 of one part per thread, the B addends loaded together)
struct BatchSumParams { const float4 *g[4]; };
__global__ void __launch_bounds__(256) k_batch_sum(const BatchSumParams p, int parts, int B, int64_t qc4, int c4n,
                                                   float4 *__restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;            // over (part, q, c4)
  if (i >= qc4 * parts) return;
  const int h = (int)(i / qc4);
  const int64_t r = i % qc4;
  const float4 *g = p.g[h];
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  if (g)
    for (int b = 0; b < B; b += 8) {
      float4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = b + u < B ? g[(int64_t)(b + u) * qc4 + r] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int u = 0; u < 8; ++u) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
    }
  const int64_t q = r / c4n;
  const int c4 = (int)(r % c4n);
  out[(q * parts + h) * c4n + c4] = acc;
}

// part[b][l][c] = sum over the rows of level l of video b of g[b][s][c] (g (B, S, C), the levels consecutive row ranges of S): the
// per-video half of the level embedding's gradient (deformable_transformer.py:100: lvl_pos_embed = pos + level_embed[l]); the sum over
// the videos follows as k_batch_sum.  One workgroup per (video, level, 256 columns), 4 wavefronts interleaved over the level's rows.
constexpr int kMaxLevelsSum = 8;
struct LevelRanges { int start[kMaxLevelsSum], len[kMaxLevelsSum]; };
__global__ void __launch_bounds__(256) k_level_sums(const float *__restrict__ g, int S, int C, int L, const LevelRanges lv,
                                                    float *__restrict__ part) {
  __shared__ float4 red[3][64];
  const int chunks = (C + 255) >> 8, n4 = C >> 2;
  const int chunk = blockIdx.x % chunks, l = (blockIdx.x / chunks) % L, b = blockIdx.x / (chunks * L);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, c4 = chunk * 64 + lane;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  if (c4 < n4) {
    const float4 *base = reinterpret_cast<const float4 *>(g + ((int64_t)b * S + lv.start[l]) * C) + c4;
    for (int t = wave; t < lv.len[l]; t += 16) {
      float4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = t + 4 * u < lv.len[l] ? base[(int64_t)(t + 4 * u) * n4] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int u = 0; u < 4; ++u) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
    }
  }
  if (wave) red[wave - 1][lane] = acc;
  __syncthreads();
  if (wave == 0 && c4 < n4) {
#pragma unroll
    for (int w = 0; w < 3; ++w) { const float4 v = red[w][lane]; acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w; }
    reinterpret_cast<float4 *>(part + ((int64_t)b * L + l) * C)[c4] = acc;
  }
}

// rows with mask[r] != 0 set to zero IN PLACE (`value.masked_fill(padding_mask[..., None], 0)` of ms_deform_attn.py:100 on the fresh
// output of value_proj): one wavefront per row, only the masked rows are written -- where the out-of-place ATen op is a copy of the
// whole tensor plus a pass over it
__global__ void __launch_bounds__(256) k_mask_rows(float *__restrict__ y, const unsigned char *__restrict__ mask, int R, int n4) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= R || !mask[row]) return;
  float4 *o = reinterpret_cast<float4 *>(y) + (int64_t)row * n4;
  for (int i = lane; i < n4; i += 64) o[i] = make_float4(0.f, 0.f, 0.f, 0.f);
}

// its gradient, out of place, with the row maxima of the result in the same pass (the Linear behind it splits its operand by them)
__global__ void __launch_bounds__(256) k_mask_rows_bwd(const float *__restrict__ dy, const unsigned char *__restrict__ mask, int R,
                                                       int n4, float *__restrict__ dx, float *__restrict__ amax) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= R) return;
  const bool dead = mask[row] != 0;
  const float4 *g = reinterpret_cast<const float4 *>(dy) + (int64_t)row * n4;
  float4 *o = reinterpret_cast<float4 *>(dx) + (int64_t)row * n4;
  float am = 0.f;
  for (int i = lane; i < n4; i += 64) {
    const float4 v = dead ? make_float4(0.f, 0.f, 0.f, 0.f) : g[i];
    o[i] = v;
    am = fmaxf(fmaxf(am, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
  }
#pragma unroll
  for (int s = 32; s; s >>= 1) am = fmaxf(am, __shfl_xor(am, s, 64));
  if (lane == 0) amax[row] = am;
}

// dst[idx[r]][:] += src[r][:] for the rows of src that are not all zero: the gradient of an embedding lookup (index_select's
// backward).  One wavefront per source row; a row of zeros -- the padded positions of a teacher-forced caption batch, about half of
// them, ALL pointing at the <pad> entry -- adds nothing and is skipped, which is what takes the time in the plain atomic
// index_add: thousands of additions serialised on one destination row (50 us at 4416 x 512 -> 8518 x 512).
__global__ void __launch_bounds__(256) k_index_add_rows(const float *__restrict__ src, int64_t ld, const int64_t *__restrict__ idx, int n,
                                                        int E, int V, float *__restrict__ dst, int64_t ldd) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= n) return;
  const int64_t v = idx[row];
  if (v < 0 || v >= V) return;
  const float4 *s = reinterpret_cast<const float4 *>(src + (int64_t)row * ld);
  float *d = dst + v * ldd;
  const int n4 = E >> 2;
  for (int base = 0; base < n4; base += 256) {                         // 4 float4 per lane and pass: E <= 1024 in one pass
    float4 x[4];
    bool any = false;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = base + lane + 64 * u;
      x[u] = i < n4 ? s[i] : make_float4(0.f, 0.f, 0.f, 0.f);
      any = any || x[u].x != 0.f || x[u].y != 0.f || x[u].z != 0.f || x[u].w != 0.f;
    }
    if (!__any(any)) continue;                                         // (wavefront-uniform)
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = base + lane + 64 * u;
      if (i < n4) {
        float *o = d + 4 * i;
        if (x[u].x != 0.f) atomicAdd(o + 0, x[u].x);
        if (x[u].y != 0.f) atomicAdd(o + 1, x[u].y);
        if (x[u].z != 0.f) atomicAdd(o + 2, x[u].z);
        if (x[u].w != 0.f) atomicAdd(o + 3, x[u].w);
      }
    }
  }
}

}  // namespace

extern "C" int gvl_index_add_rows_f32(const float *src, int64_t ld, const int64_t *idx, int n, int E, float *dst, int64_t ld_dst, int V,
                                      void *stream) {
  if (n < 0 || E <= 0 || (E & 3) || V <= 0 || ld < E || ld_dst < E || (ld & 3))
    return fail(GVL_EINVAL, "gvl_index_add_rows_f32: E %% 4 == 0, ld %% 4 == 0, ld >= E, ld_dst >= E (got n=%d E=%d V=%d)", n, E, V);
  if (n == 0) return 0;
  if (!src || !idx || !dst || ((uintptr_t)src & 15)) return fail(GVL_EINVAL, "gvl_index_add_rows_f32: null / unaligned pointer");
  return gvl::launch(GVL_PROF_LAYER_NORM, n, E, "k_index_add_rows", k_index_add_rows, dim3((n + 3) / 4), dim3(256), 0, (hipStream_t)stream,
                     src, ld, idx, n, E, V, dst, ld_dst);
}

extern "C" int gvl_mask_rows_f32(float *y, const unsigned char *mask, int R, int C, void *stream) {
  if (R < 0 || C <= 0 || (C & 3)) return fail(GVL_EINVAL, "gvl_mask_rows_f32: C %% 4 == 0 (got R=%d C=%d)", R, C);
  if (R == 0) return 0;
  if (!y || !mask || ((uintptr_t)y & 15)) return fail(GVL_EINVAL, "gvl_mask_rows_f32: null / unaligned pointer");
  return gvl::launch(GVL_PROF_LAYER_NORM, R, C, "k_mask_rows", k_mask_rows, dim3((R + 3) / 4), dim3(256), 0, (hipStream_t)stream, y, mask,
                     R, C >> 2);
}

extern "C" int gvl_mask_rows_backward_f32(const float *dy, const unsigned char *mask, int R, int C, float *dx, float *amax,
                                          void *stream) {
  if (R < 0 || C <= 0 || (C & 3)) return fail(GVL_EINVAL, "gvl_mask_rows_backward_f32: C %% 4 == 0 (got R=%d C=%d)", R, C);
  if (R == 0) return 0;
  if (!dy || !mask || !dx || !amax || (((uintptr_t)dy | (uintptr_t)dx) & 15))
    return fail(GVL_EINVAL, "gvl_mask_rows_backward_f32: null / unaligned pointer");
  return gvl::launch(GVL_PROF_LAYER_NORM, R, C, "k_mask_rows_bwd", k_mask_rows_bwd, dim3((R + 3) / 4), dim3(256), 0,
                     (hipStream_t)stream, dy, mask, R, C >> 2, dx, amax);
}

extern "C" int gvl_level_sums_f32(const float *g, int B, int S, int C, const int *starts, const int *lengths, int L, float *part,
                                  void *stream) {
  if (B < 0 || S <= 0 || C <= 0 || (C & 3) || L < 1 || L > kMaxLevelsSum || !starts || !lengths)
    return fail(GVL_EINVAL, "gvl_level_sums_f32: C %% 4 == 0, 1..%d levels (got B=%d S=%d C=%d L=%d)", kMaxLevelsSum, B, S, C, L);
  if (B == 0) return 0;
  if (!g || !part || (((uintptr_t)g | (uintptr_t)part) & 15)) return fail(GVL_EINVAL, "gvl_level_sums_f32: null / unaligned pointer");
  LevelRanges lv{};
  for (int l = 0; l < L; ++l) {
    if (starts[l] < 0 || lengths[l] < 0 || starts[l] + lengths[l] > S)
      return fail(GVL_EINVAL, "gvl_level_sums_f32: level %d = rows [%d, %d) of %d", l, starts[l], starts[l] + lengths[l], S);
    lv.start[l] = starts[l]; lv.len[l] = lengths[l];
  }
  return gvl::launch(GVL_PROF_LAYER_NORM, B, S, "k_level_sums", k_level_sums, dim3(B * L * ((C + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, g, S, C, L, lv, part);
}

extern "C" int gvl_batch_sum_f32(const float *const *grads, int parts, int B, int Q, int C, float *out, void *stream) {
  if (parts < 1 || parts > 4 || B <= 0 || Q < 0 || C <= 0 || (C & 3))
    return fail(GVL_EINVAL, "gvl_batch_sum_f32: 1..4 parts, C %% 4 == 0 (got parts=%d B=%d Q=%d C=%d)", parts, B, Q, C);
  if (Q == 0) return 0;
  if (!grads || !out || ((uintptr_t)out & 15)) return fail(GVL_EINVAL, "gvl_batch_sum_f32: null / unaligned pointer");
  BatchSumParams p{};
  for (int h = 0; h < parts; ++h) {
    if ((uintptr_t)grads[h] & 15) return fail(GVL_EINVAL, "gvl_batch_sum_f32: gradient %d is not 16-byte aligned", h);
    p.g[h] = (const float4 *)grads[h];
  }
  const int64_t qc4 = (int64_t)Q * (C >> 2);
  return gvl::launch(GVL_PROF_LAYER_NORM, Q, B, "k_batch_sum", k_batch_sum, dim3((unsigned)((qc4 * parts + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, p, parts, B, qc4, C >> 2, (float4 *)out);
}

extern "C" int gvl_count_pool_f32(const float *hs, int B, int Q, int C, float *pooled, int *arg, void *stream) {
  if (B < 0 || Q <= 0 || C <= 0 || (C & 3)) return fail(GVL_EINVAL, "gvl_count_pool_f32: bad sizes (C %% 4 == 0)");
  if (B == 0) return 0;
  if (!hs || !pooled || !arg || (((uintptr_t)hs | (uintptr_t)pooled | (uintptr_t)arg) & 15))
    return fail(GVL_EINVAL, "gvl_count_pool_f32: null / unaligned pointer");
  return gvl::launch(GVL_PROF_LAYER_NORM, B, Q, "k_count_pool", k_count_pool, dim3(B * ((C + 255) / 256)), dim3(64 * kCountWaves), 0,
                     (hipStream_t)stream, hs, Q, C, pooled, arg);
}

extern "C" int gvl_count_pool_backward_f32(const float *grad_pooled, const int *arg, int B, int Q, int C, const float *grad_row,
                                           const float *w_row, float *grad_hs, void *stream) {
  if (B < 0 || Q <= 0 || C <= 0 || (C & 3)) return fail(GVL_EINVAL, "gvl_count_pool_backward_f32: bad sizes (C %% 4 == 0)");
  if (B == 0) return 0;
  if ((grad_pooled && !arg) || (grad_row && !w_row) || !grad_hs ||
      (((uintptr_t)grad_pooled | (uintptr_t)arg | (uintptr_t)grad_hs | (uintptr_t)w_row) & 15))
    return fail(GVL_EINVAL, "gvl_count_pool_backward_f32: null / unaligned pointer");
  const int64_t total4 = (int64_t)B * Q * (C >> 2);
  return gvl::launch(GVL_PROF_LAYER_NORM, B, Q, "k_count_pool_bwd", k_count_pool_bwd, dim3((unsigned)((total4 + 255) / 256)),
                     dim3(256), 0, (hipStream_t)stream, (const float4 *)grad_pooled, (const int4 *)arg, Q, C >> 2, total4, grad_row,
                     (const float4 *)w_row, (float4 *)grad_hs);
}

extern "C" int gvl_mha_core_f32(const float *qkv, int64_t ld, const unsigned char *key_keep, int B, int Q, int H, float *out,
                                float *amax_out, void *stream) {
  if (B < 0 || Q <= 0 || H <= 0 || ld < (int64_t)3 * H * 64 || (ld & 3) || Q > 16 * kMhaMaxTiles)
    return fail(GVL_EINVAL, "gvl_mha_core_f32: needs head dimension 64, ld %% 4 == 0, Q <= %d (got B=%d Q=%d H=%d)",
                16 * kMhaMaxTiles, B, Q, H);
  if (B == 0) return 0;
  if (!qkv || !out || ((uintptr_t)qkv & 15) || ((uintptr_t)out & 15)) return fail(GVL_EINVAL, "gvl_mha_core_f32: null / unaligned pointer");
  const int nqb = (Q + 63) / 64;
  return gvl::launch(GVL_PROF_LINEAR, Q, B, "k_mha_core", k_mha_core, dim3(nqb * H * B), dim3(256), 0, (hipStream_t)stream, qkv,
                     ld, key_keep, B, Q, H, out, amax_out);
}

extern "C" int gvl_group_norm_rows_f32(const float *y, int64_t ldy, int rows_per_video, int N, int T, int C, int G,
                                       const float *gamma, const float *beta, float eps, float *dst, int64_t dst_video_stride,
                                       float *dst2, int64_t dst2_video_stride, void *stream) {
  if (N < 0 || T <= 0 || C <= 0 || G <= 0 || C % G || 64 % (C / G) || ldy < C || rows_per_video < T)
    return fail(GVL_EINVAL, "gvl_group_norm_rows_f32: needs C / G in {1, 2, 4, ..., 64} (got N=%d T=%d C=%d G=%d)", N, T, C, G);
  if (N == 0) return 0;
  if (!y || !gamma || !beta || !dst) return fail(GVL_EINVAL, "gvl_group_norm_rows_f32: null pointer");
  return gvl::launch(GVL_PROF_LAYER_NORM, T, N, "k_group_norm_rows", k_group_norm_rows, dim3(N * G), dim3(256), 0,
                     (hipStream_t)stream, y, ldy, rows_per_video, T, C, G, N, gamma, beta, eps, dst, dst_video_stride, dst2,
                     dst2_video_stride);
}

extern "C" int gvl_group_norm_rows_backward_f32(const float *y, int64_t ldy, int rows_per_video, int N, int T, int C, int G,
                                                const float *gamma, float eps, const float *dout, int64_t dout_video_stride,
                                                const float *dout2, int64_t dout2_video_stride, float *dy, int64_t ld_dy,
                                                float *dgamma_part, float *dbeta_part, void *stream) {
  return gvl_group_norm_rows_backward_amax_f32(y, ldy, rows_per_video, N, T, C, G, gamma, eps, dout, dout_video_stride, dout2,
                                               dout2_video_stride, dy, ld_dy, dgamma_part, dbeta_part, nullptr, stream);
}

extern "C" int gvl_group_norm_rows_backward_amax_f32(const float *y, int64_t ldy, int rows_per_video, int N, int T, int C, int G,
                                                     const float *gamma, float eps, const float *dout, int64_t dout_video_stride,
                                                     const float *dout2, int64_t dout2_video_stride, float *dy, int64_t ld_dy,
                                                     float *dgamma_part, float *dbeta_part, float *amax_dy, void *stream) {
  if (N < 0 || T <= 0 || C <= 0 || G <= 0 || C % G || 64 % (C / G) || ldy < C || ld_dy < C || rows_per_video < T)
    return fail(GVL_EINVAL, "gvl_group_norm_rows_backward_f32: needs C / G in {1, 2, 4, ..., 64} (got N=%d T=%d C=%d G=%d)", N, T, C, G);
  if (N == 0) return 0;
  if (!y || !gamma || !dout || !dy || !dgamma_part || !dbeta_part) return fail(GVL_EINVAL, "gvl_group_norm_rows_backward_f32: null pointer");
  return gvl::launch(GVL_PROF_LAYER_NORM, T, N, "k_group_norm_rows_bwd", k_group_norm_rows_bwd, dim3(N * G), dim3(256), 0,
                     (hipStream_t)stream, y, ldy, rows_per_video, T, C, G, N, gamma, eps, dout, dout_video_stride, dout2,
                     dout2_video_stride, dy, ld_dy, dgamma_part, dbeta_part, amax_dy);
}

extern "C" int gvl_conv_taps_to_rows_f32(const float *dcols, int N, int T1, int C, float *dx, void *stream) {
  if (N < 0 || T1 <= 0 || C <= 0 || (C & 3)) return fail(GVL_EINVAL, "gvl_conv_taps_to_rows_f32: needs C %% 4 == 0 (got N=%d T1=%d C=%d)", N, T1, C);
  if (N == 0) return 0;
  if (!dcols || !dx || (((uintptr_t)dcols | (uintptr_t)dx) & 15)) return fail(GVL_EINVAL, "gvl_conv_taps_to_rows_f32: null / unaligned pointer");
  const int64_t total = (int64_t)N * 2 * T1 * (C / 4);
  return gvl::launch(GVL_PROF_LAYER_NORM, T1, N, "k_taps_to_rows", k_taps_to_rows, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, (const float4 *)dcols, N, T1, C / 4, (float4 *)dx);
}

extern "C" int gvl_pyramid_geometry_f32(const unsigned char *mask, int N, int T0, int S, int L, const int64_t *lengths_host,
                                        const int64_t *starts_host, const float *dim_t, const float *dur_embed,
                                        const float *level_embed, int n_sine, int n_dur, float scale,
                                        unsigned char *mask_flat, float *lvl_pos, void *stream) {
  if (N < 0 || T0 <= 0 || S <= 0 || L <= 0 || L > 8 || n_sine <= 0 || (n_sine & 1) || n_dur < 0 || !lengths_host || !starts_host)
    return fail(GVL_EINVAL, "gvl_pyramid_geometry_f32: bad sizes");
  if (N == 0) return 0;
  if (!mask || !dim_t || !level_embed || !mask_flat || !lvl_pos || (n_dur > 0 && !dur_embed))
    return fail(GVL_EINVAL, "gvl_pyramid_geometry_f32: null pointer");
  PyramidDims d = {};
  int tmax = 0;
  for (int l = 0; l < L; ++l) {
    d.len[l] = (int)lengths_host[l];
    d.start[l] = (int)starts_host[l];
    if (d.len[l] <= 0 || d.start[l] < 0 || d.start[l] + d.len[l] > S || (l == 0 && d.len[0] != T0))
      return fail(GVL_EINVAL, "gvl_pyramid_geometry_f32: level %d outside S (level 0 must have T0 frames)", l);
    tmax = d.len[l] > tmax ? d.len[l] : tmax;
  }
  const size_t lds = (size_t)tmax * sizeof(float);
  if (int rc = gvl::ensure_lds(k_pyramid_geometry, lds)) return rc;
  int slices = (int)(((int64_t)(n_sine + n_dur) * d.len[0] + 4095) / 4096);
  slices = slices < 1 ? 1 : (slices > 16 ? 16 : slices);
  return gvl::launch(GVL_PROF_POS_EMBED, T0, N, "k_pyramid_geometry", k_pyramid_geometry, dim3(N, L, slices), dim3(256), lds,
                     (hipStream_t)stream, mask, T0, S, L, d, dim_t, dur_embed, level_embed, n_sine, n_dur, scale, mask_flat,
                     lvl_pos);
}

extern "C" int gvl_encoder_geometry_f32(const unsigned char *mask, int B, int S, int L, const int64_t *lengths_host,
                                        const int64_t *starts_host, float *valid_ratios, float *ref, void *stream) {
  if (B < 0 || S <= 0 || L <= 0 || L > 8 || !lengths_host || !starts_host) return fail(GVL_EINVAL, "gvl_encoder_geometry_f32: bad sizes (L <= 8)");
  if (B == 0) return 0;
  if (!mask || !valid_ratios) return fail(GVL_EINVAL, "gvl_encoder_geometry_f32: null pointer");
  LevelDims d = {};
  for (int l = 0; l < L; ++l) {
    d.len[l] = (int)lengths_host[l];
    d.start[l] = (int)starts_host[l];
    if (d.len[l] <= 0 || d.start[l] < 0 || d.start[l] + d.len[l] > S) return fail(GVL_EINVAL, "gvl_encoder_geometry_f32: level %d outside S", l);
  }
  return gvl::launch(GVL_PROF_LAYER_NORM, B, S, "k_encoder_geometry", k_encoder_geometry, dim3(B), dim3(256), 0, (hipStream_t)stream,
                     mask, S, L, d, valid_ratios, ref);
}

extern "C" int gvl_box_refine_f32(const float *delta, int64_t ldd, const float *ref, int RD, const float *valid_ratios, int B,
                                  int Q, int L, float *new_ref, float *ref_in, void *stream) {
  if (B < 0 || Q < 0 || L <= 0 || ldd < 2 || (RD != 1 && RD != 2)) return fail(GVL_EINVAL, "gvl_box_refine_f32: bad sizes");
  if ((int64_t)B * Q == 0) return 0;
  if (!delta || !ref || !new_ref || (ref_in && !valid_ratios)) return fail(GVL_EINVAL, "gvl_box_refine_f32: null pointer");
  const int R = B * Q;
  return gvl::launch(GVL_PROF_LAYER_NORM, R, L, "k_box_refine", k_box_refine, dim3((R + 255) / 256), dim3(256), 0,
                     (hipStream_t)stream, delta, ldd, ref, RD, valid_ratios, R, Q, L, new_ref, ref_in);
}

extern "C" int gvl_box_refine_backward_f32(const float *grad_new_ref, const float *new_ref, const float *ref, int RD, int R,
                                           float *grad_delta, float *grad_ref, void *stream) {
  if (R < 0 || (RD != 1 && RD != 2)) return fail(GVL_EINVAL, "gvl_box_refine_backward_f32: bad sizes");
  if (R == 0) return 0;
  if (!grad_new_ref || !new_ref || !ref || !grad_delta) return fail(GVL_EINVAL, "gvl_box_refine_backward_f32: null pointer");
  return gvl::launch(GVL_PROF_LAYER_NORM, R, RD, "k_box_refine_bwd", k_box_refine_bwd, dim3((R + 255) / 256), dim3(256), 0,
                     (hipStream_t)stream, grad_new_ref, new_ref, ref, RD, R, grad_delta, grad_ref);
}

extern "C" int gvl_count_head_f32(const float *hs, int B, int Q, int C, const float *weight, const float *bias, int n_out,
                                  float *out, void *stream) {
  if (B < 0 || Q <= 0 || C <= 0 || (C & 3) || n_out <= 0 || C > 2048)
    return fail(GVL_EINVAL, "gvl_count_head_f32: bad sizes (C %% 4 == 0, C <= 2048)");
  if (B == 0) return 0;
  if (!hs || !weight || !out) return fail(GVL_EINVAL, "gvl_count_head_f32: null pointer");
  const size_t lds = (size_t)kCountWaves * C * sizeof(float);
  if (int rc = gvl::ensure_lds(k_count_head, lds)) return rc;
  return gvl::launch(GVL_PROF_LAYER_NORM, B, Q, "k_count_head", k_count_head, dim3(B), dim3(64 * kCountWaves), lds,
                     (hipStream_t)stream, hs, Q, C, weight, bias, n_out, out);
}

extern "C" int gvl_linear_f16x3_f32(const float *a, int64_t lda, const float *a2, int64_t lda2, int a2_rows, int R, int K,
                                    const void *w_hi, const void *w_lo, const float *w_scale, const float *bias, int N,
                                    const gvl_lin_seg *segs, int nseg, int flags, void *stream) {
  if (R < 0 || N <= 0 || K <= 0 || (K % kBK) || (N % kLinBN))
    return fail(GVL_EINVAL, "gvl_linear_f16x3_f32: needs K %% 32 == 0 and N %% 64 == 0 (got R=%d N=%d K=%d)", R, N, K);
  if (nseg < 1 || nseg > kMaxSeg || !segs) return fail(GVL_EINVAL, "gvl_linear_f16x3_f32: 1..%d segments", kMaxSeg);
  if ((int64_t)N * K >= (int64_t)1 << 31) return fail(GVL_EINVAL, "gvl_linear_f16x3_f32: weight plane of more than 2^31 elements");
  if (R == 0) return 0;
  if (!a || !w_hi || !w_lo || !w_scale) return fail(GVL_EINVAL, "gvl_linear_f16x3_f32: null pointer");
  // (lda < K is allowed: overlapping rows -- the tap rows of a strided convolution over a padded input)
  if (lda <= 0 || (lda & 3) || ((uintptr_t)a & 15) || ((uintptr_t)w_hi & 15) || ((uintptr_t)w_lo & 15))
    return fail(GVL_EINVAL, "gvl_linear_f16x3_f32: a (lda %% 4 == 0) and the weight planes must be 16-byte aligned");
  bool any_addend = false;
  for (int s = 0; s < nseg; ++s) {
    const gvl_lin_seg &g = segs[s];
    if (g.n_begin % kLinBN || g.n_begin < 0 || g.n_begin >= N || (s == 0 && g.n_begin != 0) ||
        (s > 0 && g.n_begin <= segs[s - 1].n_begin))
      return fail(GVL_EINVAL, "gvl_linear_f16x3_f32: segment %d starts at column %d (ascending multiples of 64 from 0)", s, g.n_begin);
    const int n_end = s + 1 < nseg ? segs[s + 1].n_begin : N;
    const int w_ = g.width > 0 ? g.width : n_end - g.n_begin;
    if (g.width < 0 || g.width > n_end - g.n_begin) return fail(GVL_EINVAL, "gvl_linear_f16x3_f32: segment %d: bad width", s);
    if (!g.out || !g.amax_in || g.ldo < w_ || (g.resid && g.ldr < w_))
      return fail(GVL_EINVAL, "gvl_linear_f16x3_f32: segment %d: null output / row maxima or a leading dimension too small", s);
    any_addend |= (g.flags & GVL_LIN_ADDEND) != 0;
  }
  if (any_addend && (!a2 || a2_rows <= 0 || lda2 < K || (lda2 & 3) || ((uintptr_t)a2 & 15)))
    return fail(GVL_EINVAL, "gvl_linear_f16x3_f32: an ADDEND segment needs a2 (16-byte aligned, lda2 >= K, a2_rows > 0)");
  LinParams p;
  p.A = a; p.lda = lda; p.A2 = any_addend ? a2 : nullptr; p.lda2 = lda2; p.a2_rows = any_addend ? a2_rows : 1;
  p.Wh = (const _Float16 *)w_hi; p.Wl = (const _Float16 *)w_lo; p.Ws = w_scale; p.bias = bias;
  p.R = R; p.N = N; p.K = K;
  p.split_stages = 0; p.split_stride = 0;
  p.nseg = nseg;
  for (int s = 0; s < kMaxSeg; ++s) p.seg[s] = segs[s < nseg ? s : nseg - 1];
  // Tile shape.  128-column tiles when every segment starts at a multiple of 128 columns; their ROW count is the largest of
  // 128 / 96 / 64 that still gives (nearly) every CU a tile: these products are bound by what ONE CU can pull through its
  // memory path and push out through its store path (tools/ubench: ~70 GB/s in, ~20 GB/s out per CU), so a grid that covers
  // 152 of the 256 CUs (4800 x 512 at 128 rows) leaves 40 % of the chip's bandwidth unused; 96 rows -> 200 tiles, 3008 x 512 at
  // 64 rows -> 188 instead of 96.  Narrow outputs and segment boundaries at odd multiples of 64 keep the 128 x 64 tile.
  bool wide = N % 128 == 0 && N >= 128 && !(flags & GVL_LIN_XCD_COLUMNS);      // (XCD placement is per 64-column head)
  for (int s = 0; s < nseg; ++s) wide = wide && segs[s].n_begin % 128 == 0;
  int bm = kBM;
  if (wide) {
    const int tn = N / 128;
    int best = ((R + 127) / 128) * tn;
    if (best < 256)
      for (int cand : {96, 64}) {
        const int t_ = ((R + cand - 1) / cand) * tn;
        if (t_ <= 256 && t_ > best) { best = t_; bm = cand; }
      }
  }
  if (gvl::env_int("GVL_LIN_TILE", 0) == 64) wide = false;
  const int bn = wide ? 128 : kLinBN;
  p.tiles_m = (R + bm - 1) / bm; p.tiles_n = N / bn;
  p.xcd_cols = (flags & GVL_LIN_XCD_COLUMNS) && p.tiles_n >= 8;
  const int grid = (p.tiles_m * p.tiles_n + 7) / 8 * 8;
  hipStream_t st = (hipStream_t)stream;
  const bool x1 = gvl16::g_f16_products == 1;
#define GVL_LIN_LAUNCH(WM, WN, NI, NJ, NAME)                                                                              \
  {                                                                                                                        \
    if (any_addend)                                                                                                        \
      return gvl::launch(GVL_PROF_LINEAR, R, N, NAME ",addend", x1 ? k_lin_f16x3<true, WM, WN, NI, NJ, true>                \
                                                                  : k_lin_f16x3<true, WM, WN, NI, NJ, false>,               \
                         dim3(grid), dim3(64 * WM * WN), 0, st, p);                                                       \
    return gvl::launch(GVL_PROF_LINEAR, R, N, NAME, x1 ? k_lin_f16x3<false, WM, WN, NI, NJ, true>                           \
                                                       : k_lin_f16x3<false, WM, WN, NI, NJ, false>,                         \
                       dim3(grid), dim3(64 * WM * WN), 0, st, p);                                                         \
  }
  if (wide && bm == 128) GVL_LIN_LAUNCH(4, 2, 1, 2, "k_lin_f16x3<128x128>")
  if (wide && bm == 96) GVL_LIN_LAUNCH(3, 2, 1, 2, "k_lin_f16x3<96x128>")
  if (wide) GVL_LIN_LAUNCH(2, 2, 1, 2, "k_lin_f16x3<64x128>")
  GVL_LIN_LAUNCH(2, 2, 2, 1, "k_lin_f16x3<128x64>")
#undef GVL_LIN_LAUNCH
}

// partial slabs of a split-K product summed in slab order (deterministic)
static __global__ void __launch_bounds__(256) k_lin_splitk_reduce(const float4 *__restrict__ part, int64_t n4, int SK, float4 *__restrict__ out,
                                                                  const float4 *__restrict__ bias4, int row4) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  float4 t = part[i];
  for (int s = 1; s < SK; ++s) {
    const float4 u = part[(int64_t)s * n4 + i];
    t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w;
  }
  if (bias4) {                                                        // (the bias joins the finished sum: row4 = N / 4)
    const float4 b = bias4[i % row4];
    t.x += b.x; t.y += b.y; t.z += b.z; t.w += b.w;
  }
  out[i] = t;
}

extern "C" size_t gvl_linear_f16x3_splitk_workspace_bytes(int R, int N, int K) {
  if (R <= 0 || N <= 0 || K <= 0) return 0;
  const int tiles = ((R + kBM - 1) / kBM) * ((N + 127) / 128), stages = K / kBK;
  int sk = max(1, min(512 / max(tiles, 1), stages / 8));
  return sk <= 1 ? 0 : (size_t)sk * R * N * sizeof(float);
}

extern "C" int gvl_linear_f16x3_splitk_f32(const float *a, int64_t lda, const float *amax_a, int R, int K, const void *w_hi,
                                           const void *w_lo, const float *w_scale, int N, float *out, void *workspace,
                                           size_t workspace_bytes, void *stream) {
  if (lda < K) return fail(GVL_EINVAL, "gvl_linear_f16x3_splitk_f32: lda >= K (overlapping rows: gvl_linear_f16x3_splitk_bias_f32)");
  return gvl_linear_f16x3_splitk_bias_f32(a, lda, amax_a, R, K, w_hi, w_lo, w_scale, nullptr, N, out, workspace, workspace_bytes, stream);
}

extern "C" int gvl_linear_f16x3_splitk_bias_f32(const float *a, int64_t lda, const float *amax_a, int R, int K, const void *w_hi,
                                                const void *w_lo, const float *w_scale, const float *bias, int N, float *out,
                                                void *workspace, size_t workspace_bytes, void *stream) {
  const char *what = "gvl_linear_f16x3_splitk_bias_f32";
  if (R <= 0 || N <= 0 || K <= 0 || (K % kBK) || (N % 128)) return fail(GVL_EINVAL, "%s: needs K %% 32 == 0 and N %% 128 == 0 (got R=%d N=%d K=%d)", what, R, N, K);
  if (!a || !amax_a || !w_hi || !w_lo || !w_scale || !out) return fail(GVL_EINVAL, "%s: null pointer", what);
  // (lda < K is allowed, as in gvl_linear_f16x3_f32: overlapping rows -- the tap rows of a strided convolution over a padded input)
  if (lda <= 0 || (lda & 3) || ((uintptr_t)a & 15) || ((uintptr_t)w_hi & 15) || ((uintptr_t)w_lo & 15) || ((uintptr_t)out & 15) ||
      ((uintptr_t)bias & 15))
    return fail(GVL_EINVAL, "%s: a (lda %% 4 == 0), the weight planes, bias and out must be 16-byte aligned", what);
  const int tiles_m = (R + kBM - 1) / kBM, tiles_n = N / 128, stages = K / kBK;
  int sk = max(1, min(512 / (tiles_m * tiles_n), stages / 8));
  const int per = (stages + sk - 1) / sk;
  sk = (stages + per - 1) / per;
  const size_t need = sk <= 1 ? 0 : (size_t)sk * R * N * sizeof(float);
  if (need > workspace_bytes || (need && (!workspace || ((uintptr_t)workspace & 15)))) return fail(GVL_ENOSPC, "%s: workspace of %zu bytes needed", what, need);
  LinParams p;
  p.A = a; p.lda = lda; p.A2 = nullptr; p.lda2 = 0; p.a2_rows = 1;
  p.Wh = (const _Float16 *)w_hi; p.Wl = (const _Float16 *)w_lo; p.Ws = w_scale; p.bias = sk > 1 ? nullptr : bias;
  p.R = R; p.N = N; p.K = K; p.nseg = 1;
  gvl_lin_seg sg = {};
  sg.n_begin = 0; sg.flags = 0; sg.out = sk > 1 ? reinterpret_cast<float *>(workspace) : out; sg.ldo = N; sg.amax_in = amax_a;
  for (int s = 0; s < kMaxSeg; ++s) p.seg[s] = sg;
  p.tiles_m = tiles_m; p.tiles_n = tiles_n; p.xcd_cols = 0;
  p.split_stages = sk > 1 ? per : 0; p.split_stride = (int64_t)R * N;
  hipStream_t st = (hipStream_t)stream;
  const int grid = (tiles_m * tiles_n + 7) / 8 * 8;
  if (int rc = gvl::launch(GVL_PROF_LINEAR, R, N, "k_lin_f16x3<128,split-K>", k_lin_f16x3<false, 4, 2, 1, 2, false>, dim3(grid, sk),
                           dim3(512), 0, st, p))
    return rc;
  if (sk > 1) {
    const int64_t n4 = (int64_t)R * N / 4;
    hipLaunchKernelGGL(k_lin_splitk_reduce, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st,
                       reinterpret_cast<const float4 *>(workspace), n4, sk, reinterpret_cast<float4 *>(out),
                       reinterpret_cast<const float4 *>(bias), N / 4);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail((int)e, "%s: reduce launch failed: %s", what, hipGetErrorString(e));
  }
  return 0;
}

extern "C" int gvl_layer_norm_rows_f32(const float *x, int R, int C, const float *gamma, const float *beta, float eps,
                                       const float *pos, int pos_rows, float *y, float *amax_y, float *amax_ypos,
                                       void *stream) {
  if (R < 0 || C <= 0 || (C & 3) || C > 256 * kLnMaxV)
    return fail(GVL_EINVAL, "gvl_layer_norm_rows_f32: needs C %% 4 == 0 and C <= %d (got R=%d C=%d)", 256 * kLnMaxV, R, C);
  if (R == 0) return 0;
  if (!x || !gamma || !beta || !y || (pos && pos_rows <= 0)) return fail(GVL_EINVAL, "gvl_layer_norm_rows_f32: null pointer");
  if (((uintptr_t)x | (uintptr_t)gamma | (uintptr_t)beta | (uintptr_t)y | (uintptr_t)pos) & 15)
    return fail(GVL_EINVAL, "gvl_layer_norm_rows_f32: operands must be 16-byte aligned");
  const int xcd_rows = gvl::env_int("GVL_ROWS_XCD", 1);   // (0: workgroup b = rows 4 b ..: A/B runs; tools/xcd_rows_probe.py: 28.8 -> 26.9 us for LN + 4800 x 512 x 512 product)
  const int nb = (R + 3) / 4, grid = xcd_rows ? 8 * ((nb + 7) / 8) : nb;
  return gvl::launch(GVL_PROF_LAYER_NORM, R, C, "k_ln_rows", k_ln_rows, dim3(grid), dim3(256), 0, (hipStream_t)stream, x,
                     R, C, gamma, beta, eps, pos, pos ? pos_rows : 1, y, amax_y, amax_ypos, xcd_rows);
}

extern "C" int gvl_row_absmax_f32(const float *x, int64_t ldx, int R, int C, const float *pos, int64_t ldp, int pos_rows,
                                  float *amax_x, float *amax_xpos, void *stream) {
  // (ldx < C is allowed: overlapping rows -- the tap rows of a strided convolution over a padded input)
  if (R < 0 || C <= 0 || (C & 3) || ldx <= 0 || (ldx & 3)) return fail(GVL_EINVAL, "gvl_row_absmax_f32: needs C %% 4 == 0, ldx %% 4 == 0");
  if (R == 0) return 0;
  if (!x || (!amax_x && !amax_xpos) || (pos && (pos_rows <= 0 || ldp < C || (ldp & 3))))
    return fail(GVL_EINVAL, "gvl_row_absmax_f32: null pointer / bad pos");
  if (((uintptr_t)x | (uintptr_t)pos) & 15) return fail(GVL_EINVAL, "gvl_row_absmax_f32: operands must be 16-byte aligned");
  return gvl::launch(GVL_PROF_LAYER_NORM, R, C, "k_row_absmax", k_row_absmax, dim3((R + 3) / 4), dim3(256), 0,
                     (hipStream_t)stream, x, ldx, R, C, pos, ldp, pos ? pos_rows : 1, amax_x, amax_xpos);
}
