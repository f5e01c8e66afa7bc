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
// more than 2^-35 below its row's largest is not represented: |error| <= 2^-21 sum |a||b| + K 2^-33 max|a| max|b|.
// (Three bf16 parts would need six products: bf16 carries 8 bits per part, fp16 11.)
//
// k_split_rows        one wavefront per row -> hi / lo planes (fp16, K-stage-major: gvl_gemm16_common.hpp) + the row scale; used
//                     for the weights (the cell / attention kernels of gvl_cap.hip write their results as planes).
// k_gemm_f16x3_w8     out (R, N) = A (R, K) . B (N, K)^T [+ bias], both operands as planes: 8 wavefronts, 256 x 128 /
//                     128 x 256 / 128 x 128 tiles, three-stage LDS ring filled by LDS-DMA, software-pipelined, persistent
//                     workgroups (comments at the kernel).  Epilogues: store | vocabulary argmax + log-sum-exp partials.
// k_gemm_f16x3        4 wavefronts, 128 x 64 tiles, two stages, three workgroups per CU: few rows / few tiles.
//   Common: v_mfma_f32_32x32x16_f16; K in stages of 32; LDS image of a stage = 64-byte rows per plane, 16-byte chunks
//   XOR-swizzled by (row >> 2) & 3 -- the ds_read_b128 of an MFMA operand (lane l reads row l & 31, chunk 2 s + (l >> 5))
//   is then conflict-free; 16 ds_read_b128 feed 24 MFMAs per wavefront and stage (4 operand fragments serve 3 products).
//   Tiles are walked in groups of 8 row tiles per XCD (workgroup id -> XCD is id % 8) so that the workgroups resident on
//   one XCD share their operand planes in its 4 MB L2.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include <type_traits>

#include "gvl_common.hpp"
#include "gvl_gemm16_common.hpp"
#include "gvl_msda.h"

namespace {

using gvl::fail;

using namespace gvl16;

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
  for (int i = lane; i < n4; i += 64) {
    const float4 v = xr[i];                                            // second read: L1 / L2
    const float a[4] = {v.x * inv, v.y * inv, v.z * inv, v.w * inv};
    h4 h, l;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      h[c] = (_Float16)a[c];
      l[c] = (_Float16)((a[c] - (float)h[c]) * kLoScale);
    }
    const int64_t at = plane_off(row, 4 * i, R);
    *reinterpret_cast<h4 *>(hi + at) = h;
    *reinterpret_cast<h4 *>(lo + at) = l;
  }
}

// quad_perm DPP move
template <int CTRL>
__device__ __forceinline__ float quad_mov(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}

// 4 x 4 transpose inside every quad: lane q of the quad ends up with a[g] = what lane g held in a[q]
__device__ __forceinline__ void quad_transpose(float (&a)[4], int q) {
  const bool odd = q & 1, hi = q & 2;
#pragma unroll
  for (int p = 0; p < 4; p += 2) {                                    // pairs (0,1) (2,3) across lanes q ^ 1
    const float recv = quad_mov<0xB1>(odd ? a[p] : a[p + 1]);
    if (odd) a[p] = recv; else a[p + 1] = recv;
  }
#pragma unroll
  for (int p = 0; p < 2; ++p) {                                       // pairs (0,2) (1,3) across lanes q ^ 2
    const float recv = quad_mov<0x4E>(hi ? a[p] : a[p + 2]);
    if (hi) a[p] = recv; else a[p + 2] = recv;
  }
}

// kLstm epilogue of one wavefront (its 64 x 32 NJ part of D at (row0, col0), columns = 4 unit + gate), in three steps per
// 32 x 32 block so that the loads can be requested early: lstm_rows (this lane's 4 rows after the transposes, their input
// tokens and row scales), lstm_loads (the other gate parts, the token's embedding row -- a dependent load -- and the
// state), lstm_finish (transposes, cell, stores).  A first version that walked the row groups one by one (load, wait,
// compute, store) spent 12 us per tile in exposed latency.
struct LstmPre {
  int rowc[4], tok[4];
  float rs[4], cp[4];
  float4 gh[4], gc[4], ge[4], cs;
};

__device__ __forceinline__ void lstm_rows(LstmPre &p, int rbase, int lane, const float *__restrict__ As, int R,
                                          const LstmEpi &le) {
  const int fh = lane >> 5, q = lane & 3;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    p.rowc[k] = min(rbase + 8 * k + 4 * fh + q, R - 1);
    p.tok[k] = (int)le.it[p.rowc[k]];
    p.rs[k] = As[p.rowc[k]];
  }
}

__device__ __forceinline__ void lstm_loads(LstmPre &p, int cbase, int lane, const float *__restrict__ Bs, int N,
                                           const LstmEpi &le) {
  const int u = (lane & 31) >> 2;
  const int gcol = cbase + 4 * u < N ? cbase + 4 * u : 0;              // first of this lane's unit's four gate columns
  const int H = le.H;
  p.cs = *reinterpret_cast<const float4 *>(Bs + gcol);
#ifdef GVL_ABLATE_CELL_LOADS                                           /* timing-only build (tools/cell_ablate.sh) */
#pragma unroll
  for (int k = 0; k < 4; ++k) { p.gh[k] = p.cs; p.gc[k] = p.cs; p.ge[k] = p.cs; p.cp[k] = p.cs.x; }
  if (true) return;
#endif
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    p.gh[k] = *reinterpret_cast<const float4 *>(le.gates_h + (int64_t)p.rowc[k] * le.ld_h + gcol);
    p.gc[k] = le.gates_c ? *reinterpret_cast<const float4 *>(le.gates_c + (int64_t)p.rowc[k] * le.ld_c + gcol)
                         : make_float4(0.f, 0.f, 0.f, 0.f);
    p.cp[k] = le.c[(int64_t)p.rowc[k] * H + (gcol >> 2)];
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) p.ge[k] = *reinterpret_cast<const float4 *>(le.emb + (int64_t)p.tok[k] * (4 * H) + gcol);
}

__device__ __forceinline__ void lstm_finish(const LstmPre &p, const f16acc &am, const f16acc &ax, int rbase, int cbase,
                                            int lane, int R, int N, const LstmEpi &le) {
  const int fh = lane >> 5, q = lane & 3, u = (lane & 31) >> 2;
  const bool unit_ok = cbase + 4 * u < N;
  const int unit = unit_ok ? (cbase + 4 * u) >> 2 : 0, H = le.H;
  const float cs[4] = {p.cs.x, p.cs.y, p.cs.z, p.cs.w};
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    float a[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) a[t] = am[4 * k + t] + ax[4 * k + t] * kLoInv;
    quad_transpose(a, q);                                              // a[gate] of row rbase + 8 k + 4 fh + q, this lane's unit
#pragma unroll
    for (int t = 0; t < 4; ++t) a[t] = a[t] * (p.rs[k] * cs[t]) + 0.f;  // (the value kStore would have written)
    if (le.gates_c) {                                                  // same association as k_lstm_cell: (c + a) + b + e
      a[0] = p.gc[k].x + a[0]; a[1] = p.gc[k].y + a[1]; a[2] = p.gc[k].z + a[2]; a[3] = p.gc[k].w + a[3];
    }
    const float gi = a[0] + p.gh[k].x + p.ge[k].x, gf = a[1] + p.gh[k].y + p.ge[k].y, gg = a[2] + p.gh[k].z + p.ge[k].z,
                go = a[3] + p.gh[k].w + p.ge[k].w;
    float cn, hn;
#ifdef GVL_ABLATE_CELL_MATH                                            /* timing-only build */
    cn = gi + gf + p.cp[k];
    hn = gg + go;
#else
    gvl_lstm_point(gi, gf, gg, go, p.cp[k], cn, hn);
#endif
    // (h' as the fp32 number it is stored as: left to itself the compiler folds the last product of the cell into the
    //  fp16 conversions below -- v_fma_mixlo_f16 of the exact product -- and the planes no longer split h' itself)
    asm volatile("" : "+v"(hn));
    const int row = rbase + 8 * k + 4 * fh + q;
    if (row < R && unit_ok) {
      const int64_t at = (int64_t)row * H + unit;
      le.c_out[at] = cn;
      le.h_out[at] = hn;
      const _Float16 hi = (_Float16)hn;                                // planes at row scale 1 (|h'| < 1), as split4_f16
      const int64_t pat = plane_off(row, unit, R);
      le.h_hi[pat] = hi;
      le.h_lo[pat] = (_Float16)((hn - (float)hi) * 2048.f);
      if (unit == 0) le.h_scale[row] = 1.f;
    }
  }
}

template <int NJ, bool SEQ>
__device__ __forceinline__ void lstm_epilogue(f16acc (&acc_m)[2][NJ], f16acc (&acc_x)[2][NJ], int row0, int col0, int lane,
                                              const float *__restrict__ As, const float *__restrict__ Bs, int R, int N,
                                              const LstmEpi &le) {
  if constexpr (NJ == 1 && !SEQ) {
    LstmPre p[2];
    lstm_rows(p[0], row0, lane, As, R, le);
    lstm_rows(p[1], row0 + 32, lane, As, R, le);
    lstm_loads(p[0], col0, lane, Bs, N, le);
    lstm_loads(p[1], col0, lane, Bs, N, le);
    lstm_finish(p[0], acc_m[0][0], acc_x[0][0], row0, col0, lane, R, N, le);
    lstm_finish(p[1], acc_m[1][0], acc_x[1][0], row0 + 32, col0, lane, R, N, le);
  } else {                                                             // 128 accumulator registers live: one block at a time
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      LstmPre p;
      lstm_rows(p, row0 + 32 * i, lane, As, R, le);
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        lstm_loads(p, col0 + 32 * j, lane, Bs, N, le);
        lstm_finish(p, acc_m[i][j], acc_x[i][j], row0 + 32 * i, col0 + 32 * j, lane, R, N, le);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
}

// epilogue of one wavefront: its 64 x 32 NJ part of D starts at (row0, col0)
template <int NJ, int EPI, bool SEQ = false>
__device__ __forceinline__ void epilogue(f16acc (&acc_m)[2][NJ], f16acc (&acc_x)[2][NJ], int row0, int col0, int lane,
                                         const float *__restrict__ As, const float *__restrict__ Bs,
                                         const float *__restrict__ bias, int R, int N, float *__restrict__ out,
                                         int64_t ldo, const LstmEpi &le) {
  // C/D map of the 32 x 32 MFMA: column = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
  const int frow = lane & 31, fh = lane >> 5;
  if constexpr (EPI == kLstm) {
    lstm_epilogue<NJ, SEQ>(acc_m, acc_x, row0, col0, lane, As, Bs, R, N, le);
    return;
  }
  // branch-free optional bias: without one every read goes to As[0] (a finite power of two) and is multiplied by 0
  const float *bias_p = bias ? bias : As;
  const float bias_on = bias ? 1.f : 0.f;
  const int bias_ix = bias ? 0x7fffffff : 0;
  if constexpr (EPI == kArgmax) {
    // scaled, biased values in place (acc_m), one 32-row block at a time: 32 registers of row scale / bias live, and the
    // cross-term accumulators are dead afterwards
    float cs[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) cs[j] = Bs[min(col0 + 32 * j + frow, N - 1)];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      float rs[16], rb[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int rc = min(row0 + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * fh, R - 1);
        rs[r] = As[rc];
        rb[r] = bias_on * bias_p[min(rc, bias_ix)];
      }
#pragma unroll
      for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          acc_m[i][j][r] = (acc_m[i][j][r] + acc_x[i][j][r] * kLoInv) * (rs[r] * cs[j]) + rb[r];
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int col = col0 + 32 * j + frow;
      float best = -INFINITY, sum = 0.f;
      int arg = 0x7fffffff;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = row0 + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * fh;
          if (row < R && acc_m[i][j][r] > best) { best = acc_m[i][j][r]; arg = row; }   // rows ascend: first maximum
        }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = row0 + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * fh;
          if (row < R) sum += __expf(acc_m[i][j][r] - best);
        }
      // the lane 32 further holds the other half of this column's 64 rows
      const float b2 = __shfl_xor(best, 32), s2 = __shfl_xor(sum, 32);
      const int a2 = __shfl_xor(arg, 32);
      const float bn = fmaxf(best, b2);
      if (bn > -INFINITY) sum = sum * __expf(best - bn) + s2 * __expf(b2 - bn);        // exp(-inf) = 0 for an empty half
      arg = (b2 > best || (b2 == best && a2 < arg)) ? a2 : arg;
      if (fh == 0 && col < N)
        reinterpret_cast<float4 *>(out)[(int64_t)(row0 >> 6) * N + col] = make_float4(bn, sum, __int_as_float(arg), 0.f);
      __builtin_amdgcn_sched_barrier(0);
    }
    return;
  }
  // kStore: the row scales of this lane's 32 rows are read ONCE, before anything is stored: a load between two stores
  // makes the compiler wait vmcnt(0) for it, i.e. for every store issued so far
  float rs[2][16];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) rs[i][r] = As[min(row0 + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * fh, R - 1)];
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int col = col0 + 32 * j + frow;
    const bool col_ok = col < N;
    const float cs = col_ok ? Bs[col] : 0.f;
    {
      const float cb = bias_on * bias_p[min(min(col, N - 1), bias_ix)];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = row0 + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * fh;
          if (col_ok && row < R)
            out[(int64_t)row * ldo + col] = (acc_m[i][j][r] + acc_x[i][j][r] * kLoInv) * (rs[i][r] * cs) + cb;
        }
    }
  }
}

// ---- four wavefronts, 128 x BN tile, two LDS stages: the form for few rows / few tiles (three workgroups per CU at BN = 64)
// DEEP: THREE stage buffers, the DMA two stages ahead and a counted wait at the stage barrier (as the eight-wavefront kernels):
// with two buffers the barrier's vmcnt(0) waits for the DMA issued at the start of the same stage, i.e. every stage costs one
// L2 round trip (0.9 us at 4800 x 512 x 512, whose MFMA work is 0.4 us per stage); 74 KB of LDS = two workgroups per CU.
template <int BN, int EPI, bool X1 = false, bool DEEP = false>
__device__ __forceinline__ void gemm4_body(int bid, const _Float16 *__restrict__ Ah, const _Float16 *__restrict__ Al,
                                           const float *__restrict__ As, const _Float16 *__restrict__ Bh,
                                           const _Float16 *__restrict__ Bl, const float *__restrict__ Bs,
                                           const float *__restrict__ bias, int R, int N, int K, float *__restrict__ out,
                                           int64_t ldo, int tiles_m, int tiles_n, const LstmEpi &le) {
  constexpr int NJ = BN / 64;                                         // 32-column MFMA tiles per wavefront
  // ONE LDS object (a second one beside an LDS-DMA target can make hipcc drain the DMA before every ds_read):
  // [stage][A hi | A lo | B hi | B lo][row * 4 + swizzled chunk], 16-byte slots
  constexpr int kASlots = kBM * 4, kBSlots = BN * 4, kStageSlots = 2 * kASlots + 2 * kBSlots;
  __shared__ uint4 smem[(DEEP ? 3 : 2) * kStageSlots];

  int tm, tn;
  if (!tile_of(bid, tiles_m, tiles_n, tm, tn)) return;
  const int m0 = tm * kBM, n0 = tn * BN;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = (wave & 1) * 64, wn = (wave >> 1) * (BN / 2);

  // Staging by LDS-DMA (global_load_lds_dwordx4): one wave-instruction writes 64 consecutive slots = 16 rows of a plane,
  // lane l the slot base + l.  The swizzle therefore sits on the SOURCE address: slot (row, position p) holds chunk
  // p ^ ((row >> 2) & 3) of the row.  Wavefront w fills rows 32 w .. 32 w + 31 of A (two instructions per plane) and the
  // same rows of B (BN = 128) or rows 16 w .. 16 w + 15 (BN = 64).
  constexpr int NBU = BN / 64;                                        // B units (16 rows) per wavefront and plane
  const int srow = lane >> 2, spos = lane & 3;
  int64_t a_src[2], b_src[NBU];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int row = (2 * wave + u) * 16 + srow;
    a_src[u] = (int64_t)min(m0 + row, R - 1) * 32 + (spos ^ ((row >> 2) & 3)) * 8;      // (plane_off: stage k0 / 32 adds k0 R)
  }
#pragma unroll
  for (int u = 0; u < NBU; ++u) {
    const int row = (NBU * wave + u) * 16 + srow;
    b_src[u] = (int64_t)min(n0 + row, N - 1) * 32 + (spos ^ ((row >> 2) & 3)) * 8;
  }
  // (DEEP: the buffer form of the DMA -- hipcc treats a pending global_load_lds as aliasing every LDS read and drains it,
  //  vmcnt(0), in front of the next ds_read: the prefetch two stages ahead would be waited for at once)
  // ... and even the buffer builtin is waited for (vmcnt(0)) in front of the stage's first ds_read here, so the instruction is
  // written out: the compiler then neither knows the LDS write nor counts the load -- its own vmcnt waits only become more
  // conservative (an uncounted younger load makes `vmcnt(n)` wait for more, never for less), the stage waits below are explicit
  typedef int v4i_ __attribute__((ext_vector_type(4)));
  auto rsrc_of = [](const void *ptr) {
    const uint64_t u = (uint64_t)(uintptr_t)ptr;
    return v4i_{(int)__builtin_amdgcn_readfirstlane((uint32_t)u), (int)(__builtin_amdgcn_readfirstlane((uint32_t)(u >> 32)) & 0xffffu),
                0x7fffffff, 0x00020000};
  };
  const v4i_ rs_ah = rsrc_of(Ah), rs_al = rsrc_of(Al), rs_bh = rsrc_of(Bh), rs_bl = rsrc_of(Bl);
  auto bdma = [&](const v4i_ &rs, int64_t elem, uint4 *dst) {
    const uint32_t lds = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) void *)dst);
    asm volatile("s_mov_b32 m0, %2\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds" ::"v"(2u * (uint32_t)elem), "s"(rs), "s"(lds) : "memory");   // (m0: hipcc writes it in front of each of its own uses and rejects it as a clobber)
  };
  auto issue = [&](int k0, int buf) {
    uint4 *st = smem + buf * kStageSlots;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      if constexpr (DEEP) {
        bdma(rs_ah, a_src[u] + (int64_t)k0 * R, st + (2 * wave + u) * 64);
        if constexpr (!X1) bdma(rs_al, a_src[u] + (int64_t)k0 * R, st + kASlots + (2 * wave + u) * 64);
      } else {
        glds16(Ah + a_src[u] + (int64_t)k0 * R, st + (2 * wave + u) * 64);
        if constexpr (!X1) glds16(Al + a_src[u] + (int64_t)k0 * R, st + kASlots + (2 * wave + u) * 64);
      }
    }
#pragma unroll
    for (int u = 0; u < NBU; ++u) {
      if constexpr (DEEP) {
        bdma(rs_bh, b_src[u] + (int64_t)k0 * N, st + 2 * kASlots + (NBU * wave + u) * 64);
        if constexpr (!X1) bdma(rs_bl, b_src[u] + (int64_t)k0 * N, st + 2 * kASlots + kBSlots + (NBU * wave + u) * 64);
      } else {
        glds16(Bh + b_src[u] + (int64_t)k0 * N, st + 2 * kASlots + (NBU * wave + u) * 64);
        if constexpr (!X1) glds16(Bl + b_src[u] + (int64_t)k0 * N, st + 2 * kASlots + kBSlots + (NBU * wave + u) * 64);
      }
    }
  };

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
    for (int j = 0; j < NJ; ++j) fb[j][s] = 2 * kASlots + lds_slot(wn + 32 * j + frow, 2 * s + fh);
  }

  const int KT = K / kBK;
  if constexpr (DEEP) {
    constexpr int kDma = (X1 ? 1 : 2) * (2 + NBU);                     // DMA instructions per wavefront and stage
    issue(0, 0);
    issue(min(1, KT - 1) * kBK, 1);                                    // (a one-stage product fetches stage 0 twice)
    wait_vmcnt<kDma>();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    int buf = 0;
    for (int kt = 0; kt < KT; ++kt) {
      const int nb2 = buf == 0 ? 2 : buf - 1;                          // (buf + 2) % 3: last read one barrier ago
      issue(min(kt + 2, KT - 1) * kBK, nb2);                           // (past the end: the last stage again, into a buffer nobody reads)
      mfma_stage<NJ, X1>(smem + buf * kStageSlots, kASlots, kBSlots, fa, fb, acc_m, acc_x);
      wait_vmcnt<kDma>();                                              // stage kt + 1 has landed, kt + 2 stays in flight
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      buf = buf == 2 ? 0 : buf + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  } else {
  issue(0, 0);
  __syncthreads();                              // (waits for the DMA: vmcnt(0))
  for (int kt = 0; kt + 1 < KT; ++kt) {
    const int buf = kt & 1;
    issue((kt + 1) * kBK, buf ^ 1);             // the other buffer was last read one barrier ago
    mfma_stage<NJ, X1>(smem + buf * kStageSlots, kASlots, kBSlots, fa, fb, acc_m, acc_x);
    __syncthreads();
  }
  mfma_stage<NJ, X1>(smem + ((KT - 1) & 1) * kStageSlots, kASlots, kBSlots, fa, fb, acc_m, acc_x);
  }
  // (kLstm: one 32-row block at a time -- 64 registers fewer than both in flight, which keeps three workgroups on a CU; the
  //  other workgroups' products cover the block's load latency)
  epilogue<NJ, EPI, true>(acc_m, acc_x, m0 + wm, n0 + wn, lane, As, Bs, bias, R, N, out, ldo, le);
}

template <int BN, int EPI, bool X1 = false, bool DEEP = false>
__global__ void __launch_bounds__(256, 2)
    k_gemm_f16x3(const _Float16 *__restrict__ Ah, const _Float16 *__restrict__ Al, const float *__restrict__ As,
                 const _Float16 *__restrict__ Bh, const _Float16 *__restrict__ Bl, const float *__restrict__ Bs,
                 const float *__restrict__ bias, int R, int N, int K, float *__restrict__ out, int64_t ldo, int tiles_m,
                 int tiles_n, const LstmEpi le) {
  gemm4_body<BN, EPI, X1, DEEP>((int)blockIdx.x, Ah, Al, As, Bh, Bl, Bs, bias, R, N, K, out, ldo, tiles_m, tiles_n, le);
}

// ---- eight wavefronts (WM x WN, 64 x 64 each), tile 64 WM x 64 WN, THREE LDS stages with the DMA two stages ahead.
// The operand traffic of 128 x 128 tiles is 1.3 GB per vocabulary product (8 TB/s out of L2 while it runs) and a stage's
// DMA, issued when the stage before it starts, does not land within that stage's 1536 MFMA cycles: measured 30 % MFMA
// utilisation, waves parked at the barrier.  Here a stage has two stages' time to arrive and the tile moves 3/4 of the
// bytes per FLOP.  One barrier per stage: counted `s_waitcnt vmcnt` (this wavefront's DMA of the stage about to be read
// is complete, the next stage's stays in flight), raw s_barrier (everybody's is, and everybody has finished reading the
// buffer that is refilled next), then the DMA of stage kt + 2, then the MFMAs of stage kt.
template <int WM, int WN, int NJ, int EPI, bool X1 = false>
__global__ void __launch_bounds__(512, 1)
    k_gemm_f16x3_w8(const _Float16 *__restrict__ Ah, const _Float16 *__restrict__ Al, const float *__restrict__ As,
                    const _Float16 *__restrict__ Bh, const _Float16 *__restrict__ Bl, const float *__restrict__ Bs,
                    const float *__restrict__ bias, int R, int N, int K, float *__restrict__ out, int64_t ldo,
                    int tiles_m, int tiles_n, const LstmEpi le) {
  static_assert(WM * WN == 8, "eight wavefronts");
  constexpr int kRowsA = 64 * WM, kRowsB = 32 * NJ * WN;             // wavefront tile 64 x 32 NJ
  constexpr int kASlots = kRowsA * 4, kBSlots = kRowsB * 4, kStageSlots = 2 * kASlots + 2 * kBSlots;
  constexpr int NGB = kRowsB / 64, NG = WM + NGB;                     // DMA instructions per wavefront and stage
  static_assert(NG == 6 || NG == 4, "the counted waits below are written for 6 or 4 DMA instructions per stage");
  __shared__ uint4 smem[3 * kStageSlots];

  // PERSISTENT workgroups: workgroup b takes the tiles b, b + gridDim, ... of the XCD-grouped order, and the stage
  // stream runs on across tile boundaries -- the DMA of the next tile's first three stages is issued under the last
  // MFMAs of the current tile and lands during its epilogue (one workgroup per CU: nobody else would fill that gap).
  int vb = (int)blockIdx.x, tm, tn;
  if (!tile_of(vb, tiles_m, tiles_n, tm, tn)) return;
  int m0 = tm * kRowsA, n0 = tn * kRowsB;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = (wave % WM) * 64, wn = (wave / WM) * (32 * NJ);

  // DMA units of 16 rows: unit q = wave + 8 i; i < WM -> A (plane q / (4 WM), block q % (4 WM)), else B likewise.  Per
  // unit everything but the lane's row inside the block is wave-uniform (plane base, LDS slot, first row); the swizzled
  // chunk depends on the lane only ((row >> 2) & 3 == (srow >> 2) & 3: blocks start at multiples of 16 rows), so a
  // stage's source addresses are rebuilt from two lane registers whenever it is issued -- nothing per-tile is kept live.
  const int srow = lane >> 2, schunk = ((lane & 3) ^ ((srow >> 2) & 3)) * 8;
  const _Float16 *base[NG];
  int dst[NG], blk16[NG];
#pragma unroll
  for (int i = 0; i < NG; ++i) {
    const int per_plane = i < WM ? 4 * WM : kRowsB / 16, q = wave + 8 * (i < WM ? i : i - WM);
    const int plane = __builtin_amdgcn_readfirstlane(q / per_plane), blk = __builtin_amdgcn_readfirstlane(q % per_plane);
    blk16[i] = blk * 16;
    base[i] = i < WM ? (plane ? Al : Ah) : (plane ? Bl : Bh);
    dst[i] = (i < WM ? plane * kASlots : 2 * kASlots + plane * kBSlots) + blk * 64;
  }
  int tm2, tn2;
  bool has_next = tile_of(vb + (int)gridDim.x, tiles_m, tiles_n, tm2, tn2);
  if (!has_next) { tm2 = tm; tn2 = tn; }
  auto issue = [&](int tm_, int tn_, int k0, int buf) {
    uint4 *st = smem + buf * kStageSlots;
#pragma unroll
    for (int i = 0; i < NG; ++i) {
      // (which plane a unit belongs to is a compile-time fact: 8 wavefronts' units of one i lie inside one plane)
      if (X1 && (i < WM ? 8 * i / (4 * WM) : 8 * (i - WM) / (kRowsB / 16))) continue;
      const int row = i < WM ? min(tm_ * kRowsA + blk16[i] + srow, R - 1) : min(tn_ * kRowsB + blk16[i] + srow, N - 1);
      glds16_at(base[i], 2u * (uint32_t)(row * 32 + schunk + k0 * (i < WM ? R : N)), st + dst[i]);   // plane_off(row, k0 + chunk, rows)
    }
  };
  constexpr int kDma = X1 ? NG / 2 : NG;                              // DMA instructions per wavefront and stage

  f16acc acc_m[2][NJ], acc_x[2][NJ];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) { acc_m[i][j][r] = 0.f; acc_x[i][j][r] = 0.f; }

  const int frow = lane & 31, fh = lane >> 5;
  // fragment slots: one register per operand and K half; the second 32-row block is 128 slots further (same swizzle
  // term: (row + 32) >> 2 has the same low bits), an immediate offset of the ds_read
  int fa[2], fb[2];
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    fa[s] = lds_slot(wm + frow, 2 * s + fh);
    fb[s] = 2 * kASlots + lds_slot(wn + frow, 2 * s + fh);
  }

  // fragments of one K half (16) of a stage: 8 ds_read_b128; the 12 MFMAs they feed
  auto frags = [&](const uint4 *st, int s, h8 (&ah)[2], h8 (&al)[2], h8 (&bh)[NJ], h8 (&bl)[NJ]) {
#ifdef GVL_ABLATE_LDS                                                  // (timing-only build: operands from registers)
#pragma unroll
    for (int i = 0; i < 2; ++i) { ah[i] = __builtin_bit_cast(h8, make_uint4(fa[s], i, 2, 3)); al[i] = ah[i]; }
#pragma unroll
    for (int j = 0; j < NJ; ++j) { bh[j] = __builtin_bit_cast(h8, make_uint4(fb[s], j, 6, 7)); bl[j] = bh[j]; }
    if (true) return;
#endif
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      ah[i] = *reinterpret_cast<const h8 *>(&st[fa[s] + 128 * i]);
      if constexpr (!X1) al[i] = *reinterpret_cast<const h8 *>(&st[kASlots + fa[s] + 128 * i]);
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      bh[j] = *reinterpret_cast<const h8 *>(&st[fb[s] + 128 * j]);
      if constexpr (!X1) bl[j] = *reinterpret_cast<const h8 *>(&st[kBSlots + fb[s] + 128 * j]);
    }
  };
  auto mfma12 = [&](const h8 (&ah)[2], const h8 (&al)[2], const h8 (&bh)[NJ], const h8 (&bl)[NJ]) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        acc_m[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh[j], acc_m[i][j], 0, 0, 0);
        if constexpr (!X1) {
          acc_x[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bl[j], acc_x[i][j], 0, 0, 0);
          acc_x[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bh[j], acc_x[i][j], 0, 0, 0);
        }
      }
  };

  // Software pipeline.  The barrier of a stage sits BETWEEN its two K halves: the fragments of a half are requested one
  // half ahead (under the 12 MFMAs before them), the DMA of the stage three ahead is issued -- into the buffer of the
  // current stage, which everybody has finished reading at that barrier -- under the second half's MFMAs.  No wavefront
  // ever waits for LDS latency, and a stage's DMA has two and a half stages' time to land.  Every wait is the same
  // counted `vmcnt(6)` (stores of an epilogue in between are younger than the stage waited for: the wait only becomes
  // conservative); past the last tile the DMA re-fetches into buffers nobody reads any more.
  const int KT = K / kBK;                                             // >= 3 (host)
  constexpr bool kPrefetch = EPI == kLstm && NJ == 1;
  LstmPre pre[2];
  const int pre_rows_at = max(KT - 5, 0), pre_loads_at = max(KT - 3, 1);        // (kt counts finished stages: 0 .. KT - 1)
  issue(tm, tn, 0, 0);
  issue(tm, tn, kBK, 1);
  issue(tm, tn, 2 * kBK, 2);
  wait_vmcnt<2 * kDma>();
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  h8 p_ah[2], p_al[2], p_bh[NJ], p_bl[NJ];                            // first K half of the stage at hand
  frags(smem, 0, p_ah, p_al, p_bh, p_bl);
  int buf = 0, kt = 0;
  for (;;) {
    const uint4 *st = smem + buf * kStageSlots;
    const int nbuf = buf == 2 ? 0 : buf + 1;
    h8 q_ah[2], q_al[2], q_bh[NJ], q_bl[NJ];                          // second K half
    frags(st, 1, q_ah, q_al, q_bh, q_bl);
    mfma12(p_ah, p_al, p_bh, p_bl);
#ifndef GVL_NO_SCHED_GROUPS
    // (4 + 2 NJ) fragment reads of the next half between the 6 NJ MFMAs of this one
    if constexpr (!X1) {
#pragma unroll
      for (int g = 0; g < 4 + 2 * NJ; ++g) {
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      }
      if constexpr (NJ == 2) __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
    } else {                                                           // (2 + NJ) reads, 2 NJ MFMAs
#pragma unroll
      for (int g = 0; g < 2 * NJ; ++g) {
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      }
      __builtin_amdgcn_sched_group_barrier(0x100, 2 - NJ, 0);
    }
#endif
    // the next stage has landed (this wavefront's part; the barrier makes it everybody's); the one after stays in flight
    wait_vmcnt<kDma>();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                // my reads of this stage are complete
#ifndef GVL_ABLATE_BARRIER
    __builtin_amdgcn_s_barrier();
#endif
    asm volatile("" ::: "memory");
#ifndef GVL_ABLATE_DMA                                                 // (tools/gemm16_ablate.sh: timing-only builds)
    {
      const bool over = kt + 3 >= KT;                                 // the stage three ahead belongs to the next tile
      issue(over ? tm2 : tm, over ? tn2 : tn, (over ? kt + 3 - KT : kt + 3) * kBK, buf);
    }
#endif
    frags(smem + nbuf * kStageSlots, 0, p_ah, p_al, p_bh, p_bl);     // (stage 0 of the next tile at a tile's end)
    mfma12(q_ah, q_al, q_bh, q_bl);
#ifndef GVL_NO_SCHED_GROUPS
    if constexpr (X1) {                                                // kDma DMA, 2 + NJ reads, 2 NJ MFMAs
#pragma unroll
      for (int g = 0; g < 2 * NJ; ++g) {
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      }
    } else if constexpr (NJ == 2) {                                    // 6 DMA, 8 reads, 12 MFMAs
#pragma unroll
      for (int g = 0; g < 6; ++g) {
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      }
      __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
    } else {                                                           // 4 DMA, 6 reads, 6 MFMAs
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      }
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      }
    }
#endif
    buf = nbuf;
    if constexpr (kPrefetch) {
      // the cell's operands are requested while the tile's last stages run: tokens and row scales four stages before
      // its end, the first 32-row block's gate parts two stages before it (the loads are older than the DMA issued after
      // them, so the counted waits above also cover them)
      if (kt == pre_rows_at) {
        lstm_rows(pre[0], m0 + wm, lane, As, R, le);
        lstm_rows(pre[1], m0 + wm + 32, lane, As, R, le);
      }
      if (kt == pre_loads_at) lstm_loads(pre[0], n0 + wn, lane, Bs, N, le);
    }
    if (++kt == KT) {
      if constexpr (kPrefetch) {
        lstm_loads(pre[1], n0 + wn, lane, Bs, N, le);
        lstm_finish(pre[0], acc_m[0][0], acc_x[0][0], m0 + wm, n0 + wn, lane, R, N, le);
        lstm_finish(pre[1], acc_m[1][0], acc_x[1][0], m0 + wm + 32, n0 + wn, lane, R, N, le);
      } else {
        epilogue<NJ, EPI>(acc_m, acc_x, m0 + wm, n0 + wn, lane, As, Bs, bias, R, N, out, ldo, le);
      }
      if (!has_next) break;
      kt = 0;
      vb += (int)gridDim.x;
      tm = tm2;
      tn = tn2;
      m0 = tm * kRowsA;
      n0 = tn * kRowsB;
      has_next = tile_of(vb + (int)gridDim.x, tiles_m, tiles_n, tm2, tn2);
      if (!has_next) { tm2 = tm; tn2 = tn; }
      // the first half of the new tile's stage 0 is read again here rather than kept across the epilogue (32 registers
      // that the epilogue needs: keeping them spilled fragments inside the MFMA loop)
      frags(smem + buf * kStageSlots, 0, p_ah, p_al, p_bh, p_bl);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) { acc_m[i][j][r] = 0.f; acc_x[i][j][r] = 0.f; }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// ---- the same eight-wavefront persistent kernel on v_mfma_f32_16x16x32_f16.  Why: on random operands the 16 x 16 x 32
// form sustains 1.2 x the FLOP rate of the 32 x 32 x 16 form (MFMA-only builds of this kernel: 84 against 103 us for the
// vocabulary product; MI355X_MICROARCH.md notes the same for bf16) -- the chip is clock / power limited under MFMA load and
// the smaller tile costs less per FLOP.
//   wavefront tile 64 x 64 = 4 x 4 MFMA tiles; a K stage (32) is ONE MFMA step; operand lane map: row = lane & 15, the 8
//   halves k = 8 (lane >> 4) ..; LDS chunks swizzled by 2 ((row >> 2) & 1) (conflict-free for this read, on the DMA's
//   source address as before); C/D: column = lane & 15, row = 4 (lane >> 4) + register.
//   The 48 MFMAs of a stage run as four quarters of 12 in the order (rows 0-1 | cols 0-1), (0-1 | 2-3), (2-3 | 2-3),
//   (2-3 | 0-1): consecutive quarters share one operand pair, the other one is read during the quarter before -- three
//   fragment sets of 16 registers rotate (a01 / a23 and two column sets that swap roles every stage: two stages per loop
//   iteration), the barrier + DMA issue sit between the second and the third quarter.
typedef float f4acc4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t pack2h(_Float16 a, _Float16 b) {
  typedef _Float16 h2v __attribute__((ext_vector_type(2)));
  return __builtin_bit_cast(uint32_t, (h2v){a, b});
}

__device__ __forceinline__ int lds_slot16(int row, int chunk) { return row * 4 + (chunk ^ (((row >> 2) & 1) << 1)); }

template <int EPI>
__device__ __forceinline__ void epilogue16(f4acc4 (&acc_m)[4][4], f4acc4 (&acc_x)[4][4], int row0, int col0, int lane,
                                           const float *__restrict__ As, const float *__restrict__ Bs,
                                           const float *__restrict__ bias, int R, int N, float *__restrict__ out,
                                           int64_t ldo) {
  const int fc = lane & 15, fq = lane >> 4;
  const float *bias_p = bias ? bias : As;                               // branch-free optional bias (see epilogue())
  const float bias_on = bias ? 1.f : 0.f;
  const int bias_ix = bias ? 0x7fffffff : 0;
  float rs[4][4];                                                       // row scales of this lane's 16 rows, read once
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) rs[i][r] = As[min(row0 + 16 * i + 4 * fq + r, R - 1)];
  static_assert(EPI == kArgmax, "the 16 x 16 x 32 form serves the argmax epilogue (its stores would be 64-byte segments)");
  // The arithmetic below is NOT hidden behind the matrix cores -- all eight wavefronts reach it together -- and cost 15 us of
  // the 120-130 us vocabulary product when written value by value as (compare + two selects, subtract + exp, guarded add;
  // timing-only build GVL_ABLATE_EPI).  Per value now: 2 FMAs (logit), 1/2 v_max3 (column maximum FIRST, across the four
  // lanes of the column too, so that no partial sum is ever rescaled), 1 FMA + v_exp (2^(v log2e - max log2e)), 1 add, and
  // compare + select for the index of the first maximum; the row < R guards only in the last row tile (wavefront-uniform).
  {
    constexpr float kLog2e = 1.4426950408889634f;
    const bool full = row0 + 64 <= R;                                   // every row of this wavefront exists
    float rb[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) rb[i][r] = bias_on * bias_p[min(min(row0 + 16 * i + 4 * fq + r, R - 1), bias_ix)];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int col = col0 + 16 * j + fc;
      const float cs = Bs[min(col, N - 1)];
      float best = -INFINITY, sum = 0.f;
      int arg = 0x7fffffff;
      float v[4][4];
#ifdef GVL_ABLATE_EPI                                                  /* timing-only build: what the epilogue's arithmetic costs */
      best = cs;                                                       /* (every accumulator stays live: 32 adds) */
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) best += acc_m[i][j][r] + acc_x[i][j][r];
      sum = rs[0][0] + rb[1][1];
      arg = row0;
#else
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          v[i][r] = (acc_m[i][j][r] + acc_x[i][j][r] * kLoInv) * (rs[i][r] * cs) + rb[i][r];
          if (!full && row0 + 16 * i + 4 * fq + r >= R) v[i][r] = -INFINITY;           // (exp2(-inf) = 0: adds nothing)
        }
#pragma unroll
      for (int i = 0; i < 4; ++i)
        best = fmaxf(best, fmaxf(fmaxf(v[i][0], v[i][1]), fmaxf(v[i][2], v[i][3])));
      // the lanes 16, 32, 48 further hold the other rows of this column
      best = fmaxf(best, __shfl_xor(best, 16));
      best = fmaxf(best, __shfl_xor(best, 32));
      if (best > -INFINITY) {                                           // (a column tile without a single existing row: sum 0)
        const float nb = -best * kLog2e;
#pragma unroll
        for (int i = 3; i >= 0; --i)
#pragma unroll
          for (int r = 3; r >= 0; --r) {                                // descending: the FIRST maximum's row is what remains
            sum += __builtin_amdgcn_exp2f(__builtin_fmaf(v[i][r], kLog2e, nb));
            arg = v[i][r] == best ? row0 + 16 * i + 4 * fq + r : arg;
          }
      }
#endif
#pragma unroll
      for (int o = 16; o <= 32; o <<= 1) {
        sum += __shfl_xor(sum, o);
        arg = min(arg, __shfl_xor(arg, o));
      }
      if (fq == 0 && col < N)
        reinterpret_cast<float4 *>(out)[(int64_t)(row0 >> 6) * N + col] = make_float4(best, sum, __int_as_float(arg), 0.f);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
}

template <int WM, int WN, int EPI, bool X1 = false>
__global__ void __launch_bounds__(512, 1)
    k_gemm_f16x3_m16(const _Float16 *__restrict__ Ah, const _Float16 *__restrict__ Al, const float *__restrict__ As,
                     const _Float16 *__restrict__ Bh, const _Float16 *__restrict__ Bl, const float *__restrict__ Bs,
                     const float *__restrict__ bias, int R, int N, int K, float *__restrict__ out, int64_t ldo,
                     int tiles_m, int tiles_n) {
  static_assert(WM * WN == 8 && WM + WN == 6, "eight wavefronts, six DMA instructions per wavefront and stage");
  constexpr int kRowsA = 64 * WM, kRowsB = 64 * WN;
  constexpr int kASlots = kRowsA * 4, kBSlots = kRowsB * 4, kStageSlots = 2 * kASlots + 2 * kBSlots;
  constexpr int NG = WM + WN;
  __shared__ uint4 smem[3 * kStageSlots];

  int vb = (int)blockIdx.x, tm, tn;
  if (!tile_of(vb, tiles_m, tiles_n, tm, tn)) return;
  int m0 = tm * kRowsA, n0 = tn * kRowsB;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = (wave % WM) * 64, wn = (wave / WM) * 64;

  // DMA units exactly as in k_gemm_f16x3_w8, with this kernel's swizzle on the source address
  const int srow = lane >> 2, schunk = ((lane & 3) ^ (((srow >> 2) & 1) << 1)) * 8;
  const _Float16 *base[NG];
  int dst[NG], blk16[NG];
#pragma unroll
  for (int i = 0; i < NG; ++i) {
    const int per_plane = 4 * (i < WM ? WM : WN), q = wave + 8 * (i < WM ? i : i - WM);
    const int plane = __builtin_amdgcn_readfirstlane(q / per_plane), blk = __builtin_amdgcn_readfirstlane(q % per_plane);
    blk16[i] = blk * 16;
    base[i] = i < WM ? (plane ? Al : Ah) : (plane ? Bl : Bh);
    dst[i] = (i < WM ? plane * kASlots : 2 * kASlots + plane * kBSlots) + blk * 64;
  }
  int tm2, tn2;
  bool has_next = tile_of(vb + (int)gridDim.x, tiles_m, tiles_n, tm2, tn2);
  if (!has_next) { tm2 = tm; tn2 = tn; }
  auto issue = [&](int tm_, int tn_, int k0, int buf) {
    uint4 *st = smem + buf * kStageSlots;
#pragma unroll
    for (int i = 0; i < NG; ++i) {
      if (X1 && (i < WM ? 8 * i / (4 * WM) : 8 * (i - WM) / (4 * WN))) continue;       // lo-plane units (see k_gemm_f16x3_w8)
      const int row = i < WM ? min(tm_ * kRowsA + blk16[i] + srow, R - 1) : min(tn_ * kRowsB + blk16[i] + srow, N - 1);
      glds16_at(base[i], 2u * (uint32_t)(row * 32 + schunk + k0 * (i < WM ? R : N)), st + dst[i]);
    }
  };
  constexpr int kDma = X1 ? NG / 2 : NG;

  f4acc4 acc_m[4][4], acc_x[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) { acc_m[i][j] = f4acc4{0.f, 0.f, 0.f, 0.f}; acc_x[i][j] = f4acc4{0.f, 0.f, 0.f, 0.f}; }

  // fragment slots: block u of 16 rows is 64 slots further (the swizzle term repeats every 8 rows)
  const int fa = lds_slot16(wm + (lane & 15), lane >> 4), fb = 2 * kASlots + lds_slot16(wn + (lane & 15), lane >> 4);
  struct Frag { h8 h[2], l[2]; };                                      // two 16-row blocks, hi | lo
  auto rdA = [&](const uint4 *st, int i0, Frag &f) {
#ifdef GVL_ABLATE_LDS                                                  /* timing-only build: operand fragments from registers */
#pragma unroll
    for (int u = 0; u < 2; ++u) { f.h[u] = __builtin_bit_cast(h8, make_uint4(fa, i0 + u, 2, 3)); f.l[u] = f.h[u]; }
    if (true) return;
#endif
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      f.h[u] = *reinterpret_cast<const h8 *>(&st[fa + 64 * (i0 + u)]);
      if constexpr (!X1) f.l[u] = *reinterpret_cast<const h8 *>(&st[kASlots + fa + 64 * (i0 + u)]);
    }
  };
  auto rdB = [&](const uint4 *st, int j0, Frag &f) {
#ifdef GVL_ABLATE_LDS
#pragma unroll
    for (int u = 0; u < 2; ++u) { f.h[u] = __builtin_bit_cast(h8, make_uint4(fb, j0 + u, 6, 7)); f.l[u] = f.h[u]; }
    if (true) return;
#endif
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      f.h[u] = *reinterpret_cast<const h8 *>(&st[fb + 64 * (j0 + u)]);
      if constexpr (!X1) f.l[u] = *reinterpret_cast<const h8 *>(&st[kBSlots + fb + 64 * (j0 + u)]);
    }
  };
#define GVL_QUARTER(A, B, I0, J0)                                                                                  \
  _Pragma("unroll") for (int u = 0; u < 2; ++u) _Pragma("unroll") for (int v = 0; v < 2; ++v) {                     \
    acc_m[I0 + u][J0 + v] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A.h[u], B.h[v], acc_m[I0 + u][J0 + v], 0, 0, 0); \
    if constexpr (!X1) {                                                                                             \
      acc_x[I0 + u][J0 + v] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A.h[u], B.l[v], acc_x[I0 + u][J0 + v], 0, 0, 0); \
      acc_x[I0 + u][J0 + v] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A.l[u], B.h[v], acc_x[I0 + u][J0 + v], 0, 0, 0); \
    }                                                                                                                \
  }
#define GVL_GROUPS_READS()  /* 4 fragment reads among 12 MFMAs (X1: 2 among 4) */                                  \
  if constexpr (X1) {                                                                                                \
    _Pragma("unroll") for (int g = 0; g < 2; ++g) {                                                                 \
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                                            \
      __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);                                                            \
    }                                                                                                                \
  } else                                                                                                             \
  _Pragma("unroll") for (int g = 0; g < 4; ++g) {                                                                   \
    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                                              \
    __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);                                                              \
  }
#define GVL_GROUPS_DMA()    /* 6 DMA + 4 fragment reads among 12 MFMAs (X1: 3 + 2 among 4) */                       \
  if constexpr (X1) {                                                                                                \
    _Pragma("unroll") for (int g = 0; g < 3; ++g) {                                                                 \
      __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                                                            \
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                            \
    }                                                                                                                \
    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);                                                              \
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                              \
  } else {                                                                                                           \
  _Pragma("unroll") for (int g = 0; g < 6; ++g) {                                                                   \
    __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                                                              \
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                              \
  }                                                                                                                  \
  _Pragma("unroll") for (int g = 0; g < 2; ++g) {                                                                   \
    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                                              \
    __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);                                                              \
  }                                                                                                                  \
  _Pragma("unroll") for (int g = 0; g < 2; ++g) {                                                                   \
    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                                              \
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                              \
  }                                                                                                                  \
  }

  const int KT = K / kBK;                                             // even and >= 4 (host)
  issue(tm, tn, 0, 0);
  issue(tm, tn, kBK, 1);
  issue(tm, tn, 2 * kBK, 2);
  wait_vmcnt<2 * kDma>();
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  Frag a01, a23, bX, bY;                                              // bX: columns 0-1 of the stage at hand
  rdA(smem, 0, a01);
  rdB(smem, 0, bX);
  int buf = 0, kt = 0;
  // (timing-only builds, tools/m16_ablate.sh: GVL_ABLATE_BARRIER / GVL_ABLATE_DMA drop the stage barrier / the operand DMA)
#ifdef GVL_ABLATE_BARRIER
#define GVL_M16_BARRIER()
#else
#define GVL_M16_BARRIER() __builtin_amdgcn_s_barrier()
#endif
#ifdef GVL_ABLATE_DMA
#define GVL_M16_ISSUE()
#else
#define GVL_M16_ISSUE()                                                                                              \
  {                                                                                                                  \
    const bool over = kt + 3 >= KT;                                                                                  \
    issue(over ? tm2 : tm, over ? tn2 : tn, (over ? kt + 3 - KT : kt + 3) * kBK, buf);                               \
  }
#endif
  // one K stage: bA holds its columns 0-1 on entry, bB receives columns 2-3 and then the NEXT stage's columns 0-1
#define GVL_STAGE(bA, bB)                                                                                           \
  {                                                                                                                  \
    const uint4 *st = smem + buf * kStageSlots;                                                                      \
    const int nbuf = buf == 2 ? 0 : buf + 1;                                                                         \
    rdB(st, 2, bB);                                                                                                  \
    GVL_QUARTER(a01, bA, 0, 0)                                                                                       \
    GVL_GROUPS_READS()                                                                                               \
    rdA(st, 2, a23);                                                                                                 \
    GVL_QUARTER(a01, bB, 0, 2)                                                                                       \
    GVL_GROUPS_READS()                                                                                               \
    wait_vmcnt<kDma>();                                                                                              \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                               \
    GVL_M16_BARRIER();                                                                                               \
    asm volatile("" ::: "memory");                                                                                   \
    GVL_M16_ISSUE();                                                                                                 \
    rdA(smem + nbuf * kStageSlots, 0, a01);                                                                          \
    GVL_QUARTER(a23, bB, 2, 2)                                                                                       \
    GVL_GROUPS_DMA()                                                                                                 \
    rdB(smem + nbuf * kStageSlots, 0, bB);                                                                           \
    GVL_QUARTER(a23, bA, 2, 0)                                                                                       \
    GVL_GROUPS_READS()                                                                                               \
    buf = nbuf;                                                                                                      \
    ++kt;                                                                                                            \
  }
  for (;;) {
    GVL_STAGE(bX, bY)
    GVL_STAGE(bY, bX)
    if (kt == KT) {
      epilogue16<EPI>(acc_m, acc_x, m0 + wm, n0 + wn, lane, As, Bs, bias, R, N, out, ldo);
      if (!has_next) break;
      kt = 0;
      vb += (int)gridDim.x;
      tm = tm2;
      tn = tn2;
      m0 = tm * kRowsA;
      n0 = tn * kRowsB;
      has_next = tile_of(vb + (int)gridDim.x, tiles_m, tiles_n, tm2, tn2);
      if (!has_next) { tm2 = tm; tn2 = tn; }
      // (the first fragments of the new tile are read again rather than kept across the epilogue)
      rdA(smem + buf * kStageSlots, 0, a01);
      rdB(smem + buf * kStageSlots, 0, bX);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) { acc_m[i][j] = f4acc4{0.f, 0.f, 0.f, 0.f}; acc_x[i][j] = f4acc4{0.f, 0.f, 0.f, 0.f}; }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#undef GVL_STAGE
#undef GVL_M16_ISSUE
#undef GVL_M16_BARRIER
#undef GVL_GROUPS_DMA
#undef GVL_GROUPS_READS
#undef GVL_QUARTER
}

// ---- the vocabulary product on 256 x 320 tiles, ONE accumulator per output (round 5).
// Why: the 128 x 256 tiles of k_gemm_f16x3_m16 keep the CU's LDS pipe 89 % occupied (83 B/clk of fragment reads + 31 B/clk of
// DMA writes against 128 B/clk) and ask the vector-memory path for 48 KB per 1536 MFMA cycles, close to the ~ 39 B/clk a CU
// sustains (DESIGN_LOG.md 4.4: the kernel ran 74 instead of 104 us without the fragment reads).  Both shrink with the tile:
//   workgroup tile   256 vocabulary rows (A) x 320 hidden rows (B): 34 x 15 = 510 tiles at (8518, 4800) = 1.99 rounds of 256 CUs;
//                    a K stage is 72 KB, TWO stage buffers (the three-stage ring of the smaller tile does not fit)
//   wavefront tile   64 (A) x 160 (B) = 4 x 10 tiles of v_mfma_f32_16x16x32_f16: 28 ds_read_b128 per 120 MFMAs (m16: 16 per 48),
//                    i.e. 58 B/clk of reads + 19 B/clk of DMA writes per CU
//   one accumulator  with hi' = 2^11 hi the product 2^22 a b = hi'a hi'b + hi'a lo_b + lo_a hi'b: the three MFMAs add into the same
//                    fp32 accumulator (the cross terms are 2^11 smaller and lose what an fp32 running sum loses anyway); the
//                    fragments arrive in the planes' (hi, 2^11 lo) format and hi is scaled after the read (four v_pk_mul_f16 per
//                    fragment) -- 160 accumulator registers, where two accumulators per output would need 320
//   schedule         the B fragments rotate through four register sets, requested two blocks (24 MFMAs) ahead; the stage's ONE
//                    barrier sits before its last two B blocks, when every read of the stage's buffer has been issued: after it the
//                    DMA of the stage after next goes into that buffer, and the next stage's first fragments (A in place, block
//                    by block as the last MFMAs release them) come from the other buffer -- whose DMA was issued a stage ago.
template <int NJ>
__device__ __forceinline__ void epilogue_v(f4acc4 (&acc)[4][NJ], int row0, int col0, int lane, const float *__restrict__ As,
                                           const float *__restrict__ Bs, const float *__restrict__ bias, int R, int N,
                                           float *__restrict__ out) {
  const int fc = lane & 15, fq = lane >> 4;
  const float *bias_p = bias ? bias : As;
  const float bias_on = bias ? 1.f : 0.f;
  const int bias_ix = bias ? 0x7fffffff : 0;
  constexpr float kLog2e = 1.4426950408889634f, kInv22 = 1.f / 4194304.f;
  const bool full = row0 + 64 <= R;
  float rs[4][4], rb[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = min(row0 + 16 * i + 4 * fq + r, R - 1);
      rs[i][r] = As[row] * kInv22;
      rb[i][r] = bias_on * bias_p[min(row, bias_ix)];
    }
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int col = col0 + 16 * j + fc;
    const float cs = Bs[min(col, N - 1)];
    float best = -INFINITY, sum = 0.f;
    int arg = 0x7fffffff;
    float v[4][4];
#ifdef GVL_V_NO_EPI                                                    /* timing-only build: every accumulator stays live, 16 adds */
    best = cs;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) best += acc[i][j][r];
    sum = rs[0][0] + rb[1][1];
    arg = row0;
    (void)v; (void)full; (void)kLog2e;
#else
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        v[i][r] = __builtin_fmaf(acc[i][j][r], rs[i][r] * cs, rb[i][r]);
        if (!full && row0 + 16 * i + 4 * fq + r >= R) v[i][r] = -INFINITY;
      }
#pragma unroll
    for (int i = 0; i < 4; ++i) best = fmaxf(best, fmaxf(fmaxf(v[i][0], v[i][1]), fmaxf(v[i][2], v[i][3])));
    best = fmaxf(best, __shfl_xor(best, 16));
    best = fmaxf(best, __shfl_xor(best, 32));
    if (best > -INFINITY) {
      const float nb = -best * kLog2e;
#pragma unroll
      for (int i = 3; i >= 0; --i)
#pragma unroll
        for (int r = 3; r >= 0; --r) {                                  // descending: the FIRST maximum's row is what remains
          sum += __builtin_amdgcn_exp2f(__builtin_fmaf(v[i][r], kLog2e, nb));
          arg = v[i][r] == best ? row0 + 16 * i + 4 * fq + r : arg;
        }
    }
#endif
#pragma unroll
    for (int o = 16; o <= 32; o <<= 1) {
      sum += __shfl_xor(sum, o);
      arg = min(arg, __shfl_xor(arg, o));
    }
    if (fq == 0 && col < N)
      reinterpret_cast<float4 *>(out)[(int64_t)(row0 >> 6) * N + col] = make_float4(best, sum, __int_as_float(arg), 0.f);
    __builtin_amdgcn_sched_barrier(0);
  }
}

// X1 (gvl_f16_products(1): inference under autocast): the leading product only -- the lo planes are neither fetched (their DMA
// units are skipped: the stage waits are vmcnt(0), not counted) nor read from LDS nor multiplied
template <bool X1>
__global__ void __launch_bounds__(512, 1)
    k_vocab_f16x3(const _Float16 *__restrict__ Ah, const _Float16 *__restrict__ Al, const float *__restrict__ As,
                  const _Float16 *__restrict__ Bh, const _Float16 *__restrict__ Bl, const float *__restrict__ Bs,
                  const float *__restrict__ bias, int R, int N, int K, float *__restrict__ out, int tiles_m, int tiles_n,
                  int chunks) {
  constexpr int kRowsA = 256, kRowsB = 320, NJ = 10;
  constexpr int kASlots = kRowsA * 4, kBSlots = kRowsB * 4, kStageSlots = 2 * kASlots + 2 * kBSlots;
  __shared__ uint4 smem[2 * kStageSlots];

  int vb = (int)blockIdx.x, tm, tn;
  if (!tile_of(vb, tiles_m, tiles_n, tm, tn)) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wa = (wave & 3) * 64, wb = (wave >> 2) * 160;

  // DMA units of 16 rows x 64 bytes (one buffer_load_dwordx4 ... lds): unit i of wavefront w is
  //   i = 0, 1: A hi block w + 8 i        i = 2, 3: A lo block w + 8 (i - 2)       i = 4, 5: B hi block w + 8 (i - 4)
  //   i = 6:    B hi block 16 + w (w < 4) | B lo block 12 + w (w >= 4)             i = 7, 8: B lo block w + 8 (i - 7)
  // The lane's part of the address (row inside the unit, swizzled 16-byte chunk) is ONE register; everything else is scalar, and
  // the planes are read through buffer descriptors: rows past the end of a plane (ragged last tiles) come back as zeros.
  const int srow = lane >> 2, schunk = ((lane & 3) ^ (((srow >> 2) & 1) << 1)) * 8;
  const uint32_t lane_off = 2u * (uint32_t)(srow * 32 + schunk);
  const __amdgpu_buffer_rsrc_t rs_ah = __builtin_amdgcn_make_buffer_rsrc((void *)Ah, 0, R * K * 2, 0x00020000),
                               rs_al = __builtin_amdgcn_make_buffer_rsrc((void *)Al, 0, R * K * 2, 0x00020000),
                               rs_bh = __builtin_amdgcn_make_buffer_rsrc((void *)Bh, 0, N * K * 2, 0x00020000),
                               rs_bl = __builtin_amdgcn_make_buffer_rsrc((void *)Bl, 0, N * K * 2, 0x00020000);
  int tm2, tn2;
  bool has_next = tile_of(vb + (int)gridDim.x, tiles_m, tiles_n, tm2, tn2);
  if (!has_next) { tm2 = tm; tn2 = tn; }
#ifdef GVL_V_SAME_SRC                                                   /* timing-only: every unit re-fetches the plane's first KiB */
#define GVL_V_SRC(X) (lane_off + 0u * (X))
#else
#define GVL_V_SRC(X) (X)
#endif
  // unit I of the K stage at k0 of tile (tm_, tn_) -> stage buffer st
#define GVL_V_DMA(I, TM, TN, K0, ST)                                                                                 \
  {                                                                                                                  \
    constexpr bool isA = (I) < 4;                                                                                    \
    const bool lo = (I) == 6 ? wave >= 4 : ((I) == 2 || (I) == 3 || (I) >= 7);                                       \
    const int blk = (I) == 6 ? (wave >= 4 ? 12 + wave : 16 + wave) : wave + (((I) == 1 || (I) == 3 || (I) == 5 || (I) == 8) ? 8 : 0);  \
    const int rows = isA ? R : N, row0 = isA ? (TM) * kRowsA : (TN) * kRowsB;                                        \
    const uint32_t soff = 2u * (uint32_t)((K0) * rows + (row0 + blk * 16) * 32);                                     \
    const int dst = (isA ? (lo ? kASlots : 0) : 2 * kASlots + (lo ? kBSlots : 0)) + blk * 64;                        \
    if (!(X1 && lo))                                                                                                 \
      __builtin_amdgcn_raw_ptr_buffer_load_lds(isA ? (lo ? rs_al : rs_ah) : (lo ? rs_bl : rs_bh),                    \
                                               (__attribute__((address_space(3))) void *)((ST) + dst), 16,           \
                                               GVL_V_SRC(lane_off + soff), 0, 0, 0);                                 \
  }

  f4acc4 acc[4][NJ];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[i][j] = f4acc4{0.f, 0.f, 0.f, 0.f};

  const int fa = lds_slot16(wa + (lane & 15), lane >> 4), fb = 2 * kASlots + lds_slot16(wb + (lane & 15), lane >> 4);
  h8 ah[4], al[4], bh[4], bl[4];
  const _Float16 k2048 = (_Float16)2048.f;
  // (timing-only builds, tools/vocab_ablate.sh: GVL_V_NO_DMA / GVL_V_NO_BARRIER / GVL_V_NO_EPI drop the operand DMA / the stage
  //  barrier / the epilogue's arithmetic)
#define GVL_V_RDA(ST, I)                                                                                             \
  {                                                                                                                  \
    ah[I] = *reinterpret_cast<const h8 *>(&(ST)[fa + 64 * (I)]);                                                     \
    if constexpr (!X1) al[I] = *reinterpret_cast<const h8 *>(&(ST)[kASlots + fa + 64 * (I)]);                        \
  }
#define GVL_V_RDB(ST, J, S)                                                                                          \
  {                                                                                                                  \
    bh[S] = *reinterpret_cast<const h8 *>(&(ST)[fb + 64 * (J)]);                                                     \
    if constexpr (!X1) bl[S] = *reinterpret_cast<const h8 *>(&(ST)[kBSlots + fb + 64 * (J)]);                        \
  }
#ifdef GVL_V_NO_VMWAIT
#define GVL_V_WAIT() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#else
#define GVL_V_WAIT() asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory")
#endif
#ifdef GVL_V_NO_BARRIER
#define GVL_V_BARRIER()
#else
#define GVL_V_BARRIER() __builtin_amdgcn_s_barrier()
#endif
  // the units of a stage's DMA are issued over seven B blocks: units 0-3 under blocks 8, 9 of the stage BEFORE the one that runs
  // while they land (AHEAD = 2: into the buffer the barrier has just freed), units 4-8 under blocks 0-2 of that stage (AHEAD = 1)
#if defined(GVL_V_NO_DMA) || defined(GVL_V_DMA_ONCE)                    /* ONCE: both buffers filled before the loop, then none */
#define GVL_V_ISSUE(P, I, AHEAD, ST)
#else
#define GVL_V_ISSUE(P, I, AHEAD, ST)                                                                                 \
  {                                                                                                                  \
    const int kn = kt + (P) + (AHEAD);                                                                               \
    const bool over = kn >= KT;                                                                                      \
    GVL_V_DMA(I, over ? tm2 : tm, over ? tn2 : tn, (over ? kn - KT : kn) * kBK, ST)                                  \
  }
#endif
#define GVL_V_MFMA3(I, J, BS, S)                                                                                     \
  {                                                                                                                  \
    acc[I][J] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[I], BS, acc[I][J], 0, 0, 0);                               \
    if constexpr (!X1) {                                                                                             \
      acc[I][J] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[I], bl[S], acc[I][J], 0, 0, 0);                          \
      acc[I][J] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[I], BS, acc[I][J], 0, 0, 0);                             \
    }                                                                                                                \
  }
  // B block J of stage parity P: its fragments sit in register set (2 P + J) & 3; the block two further is requested first.
  // HALVES: the A blocks 0-1 and 2-3 as two scheduling regions -- the stage's first block scales the A fragments that arrived
  // last (2, 3) only after the MFMAs of blocks 0, 1; its last block reloads A blocks 0, 1 from the next stage's buffer as soon
  // as their MFMAs have been issued.
#define GVL_V_BLOCK(P, J, ST, NST, PRE0, MID, POST)                                                                  \
  {                                                                                                                  \
    constexpr int cur = (2 * (P) + (J)) & 3, nx = (2 * (P) + (J) + 2) & 3;                                           \
    if constexpr ((J) + 2 < NJ) GVL_V_RDB(ST, (J) + 2, nx)                                                           \
    else GVL_V_RDB(NST, (J) + 2 - NJ, nx)                                                                            \
    PRE0                                                                                                             \
    const h8 bs = bh[cur] * k2048;                                                                                   \
    GVL_V_MFMA3(0, J, bs, cur)                                                                                       \
    GVL_V_MFMA3(1, J, bs, cur)                                                                                       \
    MID                                                                                                              \
    GVL_V_MFMA3(2, J, bs, cur)                                                                                       \
    GVL_V_MFMA3(3, J, bs, cur)                                                                                       \
    POST                                                                                                             \
    __builtin_amdgcn_sched_barrier(0);                                                                               \
  }
#define GVL_V_NONE
#define GVL_V_STAGE(P)                                                                                               \
  {                                                                                                                  \
    uint4 *st = smem + (P) * kStageSlots;                                                                            \
    uint4 *nst = smem + (1 - (P)) * kStageSlots;                                                                     \
    GVL_V_BLOCK(P, 0, st, nst,                                                                                       \
                ah[0] = ah[0] * k2048; ah[1] = ah[1] * k2048; GVL_V_ISSUE(P, 4, 1, nst) GVL_V_ISSUE(P, 5, 1, nst),   \
                __builtin_amdgcn_sched_barrier(0); ah[2] = ah[2] * k2048; ah[3] = ah[3] * k2048;, GVL_V_NONE)        \
    GVL_V_BLOCK(P, 1, st, nst, GVL_V_ISSUE(P, 6, 1, nst) GVL_V_ISSUE(P, 7, 1, nst), GVL_V_NONE, GVL_V_NONE)          \
    GVL_V_BLOCK(P, 2, st, nst, GVL_V_ISSUE(P, 8, 1, nst), GVL_V_NONE, GVL_V_NONE)                                    \
    GVL_V_BLOCK(P, 3, st, nst, GVL_V_NONE, GVL_V_NONE, GVL_V_NONE)                                                   \
    GVL_V_BLOCK(P, 4, st, nst, GVL_V_NONE, GVL_V_NONE, GVL_V_NONE)                                                   \
    GVL_V_BLOCK(P, 5, st, nst, GVL_V_NONE, GVL_V_NONE, GVL_V_NONE)                                                   \
    GVL_V_BLOCK(P, 6, st, nst, GVL_V_NONE, GVL_V_NONE, GVL_V_NONE)                                                   \
    GVL_V_BLOCK(P, 7, st, nst, GVL_V_NONE, GVL_V_NONE, GVL_V_NONE)                                                   \
    /* every read of this stage's buffer has been issued: once they are complete (and my units of the next stage     \
       have landed) the barrier makes both true for everybody */                                                     \
    GVL_V_WAIT();                                                                                                    \
    GVL_V_BARRIER();                                                                                                 \
    asm volatile("" ::: "memory");                                                                                   \
    GVL_V_BLOCK(P, 8, st, nst, GVL_V_ISSUE(P, 0, 2, st) GVL_V_ISSUE(P, 1, 2, st), GVL_V_NONE, GVL_V_NONE)            \
    GVL_V_BLOCK(P, 9, st, nst, GVL_V_ISSUE(P, 2, 2, st) GVL_V_ISSUE(P, 3, 2, st),                                    \
                __builtin_amdgcn_sched_barrier(0); GVL_V_RDA(nst, 0) GVL_V_RDA(nst, 1),                              \
                __builtin_amdgcn_sched_barrier(0); GVL_V_RDA(nst, 2) GVL_V_RDA(nst, 3))                              \
  }

  const int KT = K / kBK;                                              // even, >= 4 (host)
  int kt = 0;
#ifdef GVL_V_CLOCKS                                                    /* dev: shader clocks and 100 MHz ticks of workgroup 0 */
  const uint64_t c0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
#endif
  {                                                                    // stage 0 whole, units 0-3 of stage 1
    constexpr int P = 0;
    kt = -2;                                                           // (GVL_V_ISSUE adds P + AHEAD)
#ifdef GVL_V_DMA_ONCE
#define GVL_V_ISSUE0(I, ST) GVL_V_DMA(I, tm, tn, (kt + 2) * kBK, ST)
#else
#define GVL_V_ISSUE0(I, ST) GVL_V_ISSUE(P, I, 2, ST)
#endif
    GVL_V_ISSUE0(0, smem) GVL_V_ISSUE0(1, smem) GVL_V_ISSUE0(2, smem) GVL_V_ISSUE0(3, smem) GVL_V_ISSUE0(4, smem)
    GVL_V_ISSUE0(5, smem) GVL_V_ISSUE0(6, smem) GVL_V_ISSUE0(7, smem) GVL_V_ISSUE0(8, smem)
    kt = -1;
    GVL_V_ISSUE0(0, smem + kStageSlots) GVL_V_ISSUE0(1, smem + kStageSlots) GVL_V_ISSUE0(2, smem + kStageSlots)
    GVL_V_ISSUE0(3, smem + kStageSlots)
#ifdef GVL_V_DMA_ONCE
    GVL_V_ISSUE0(4, smem + kStageSlots) GVL_V_ISSUE0(5, smem + kStageSlots) GVL_V_ISSUE0(6, smem + kStageSlots)
    GVL_V_ISSUE0(7, smem + kStageSlots) GVL_V_ISSUE0(8, smem + kStageSlots)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
#undef GVL_V_ISSUE0
    kt = 0;
  }
  wait_vmcnt<4>();
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
#pragma unroll
  for (int i = 0; i < 4; ++i) GVL_V_RDA(smem, i)
  GVL_V_RDB(smem, 0, 0)
  GVL_V_RDB(smem, 1, 1)
  for (;;) {
    GVL_V_STAGE(0)
    GVL_V_STAGE(1)
    kt += 2;
    if (kt == KT) {
      if (((tm * kRowsA + wa) >> 6) < chunks)
        epilogue_v<NJ>(acc, tm * kRowsA + wa, tn * kRowsB + wb, lane, As, Bs, bias, R, N, out);
      if (!has_next) break;
      kt = 0;
      vb += (int)gridDim.x;
      tm = tm2;
      tn = tn2;
      has_next = tile_of(vb + (int)gridDim.x, tiles_m, tiles_n, tm2, tn2);
      if (!has_next) { tm2 = tm; tn2 = tn; }
      // (the first fragments of the new tile are read again rather than kept across the epilogue)
#pragma unroll
      for (int i = 0; i < 4; ++i) GVL_V_RDA(smem, i)
      GVL_V_RDB(smem, 0, 0)
      GVL_V_RDB(smem, 1, 1)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = f4acc4{0.f, 0.f, 0.f, 0.f};
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef GVL_V_CLOCKS
  if (blockIdx.x == 0 && tid == 0) {
    const uint64_t c1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
    printf("k_vocab clocks: %llu cycles in %llu ticks of 10 ns = %.3f GHz\n", (unsigned long long)(c1 - c0),
           (unsigned long long)(r1 - r0), (double)(c1 - c0) / (10.0 * (double)(r1 - r0)));
  }
#endif
#undef GVL_V_STAGE
#undef GVL_V_NONE
#undef GVL_V_ISSUE
#undef GVL_V_BARRIER
#undef GVL_V_WAIT
#undef GVL_V_BLOCK
#undef GVL_V_MFMA3
#undef GVL_V_RDB
#undef GVL_V_RDA
#undef GVL_V_DMA
#undef GVL_V_SRC
}

// ---- k_gates_f16x3: BOTH halves of the LSTM gate product of a token step in one launch, the cell applied (round 5).
//   gates (n, 4H) = h W_hh^T + att W_ih[att]^T + gates_c + emb_gates[it]   (LSTM_DSA.py:267-269, nn.LSTM's pointwise part :216-217)
// Round 4 ran the recurrent half inside the (n, A + 4H) product over h in front of the attention (45.9 us, 49 MB written) and the
// attention half + cell as a second product (58.8 us, reading that part back): 104.6 us per token for 68 GFLOP of fp16 MFMA.  Here
// A = the planes of [W_hh | W_ih[att]] (4H gate rows in the order 4 unit + gate, contraction K1 + K2) and B = the planes of h for
// the first K1 / 32 stages, of att for the rest; between the two the accumulators move from h's row scale to att's (powers of two:
// exact).  The k_vocab_f16x3 recipe on a 256 x 160 tile (8 x 30 = 240 tiles at (2048, 4800): one per CU, one round): eight
// wavefronts of 64 x 80 (4 x 5 tiles of v_mfma_f32_16x16x32_f16, ONE accumulator per output: 80 registers), a K stage is 52 KB and
// -- a stage being only 960 MFMA cycles per wavefront -- the ring has THREE stage buffers with the DMA two to three stages ahead
// (every wavefront issues exactly seven units per stage, so the stage barrier's wait is the counted `vmcnt(7)`).  In the
// accumulator map of the 16 x 16 x 32 MFMA the four registers of a lane are the four gates of ONE (row, unit): the cell needs no
// transposes; its operands (gate addends, embedding rows, state) are requested in two batches with every load of a batch in
// flight, and h', c' and the planes of h' leave through an LDS transpose as whole 256-byte row segments (written straight from
// the accumulator layout -- 16 rows x 16 bytes per store instruction -- the stores alone took 40 us).
struct GatesEpi {
  const float *gates_c;                // (n, >= 4H) token-independent gate part, may be null
  int64_t ld_c;
  const float *emb;                    // (V + 1, 4H)
  const int64_t *it;                   // (n)
  const float *c;                      // (n, H)
  float *h_out, *c_out;                // (n, H)
  _Float16 *h_hi, *h_lo;               // planes of h' at row scale 1
  float *h_scale;
  int H;
};

// X1 (gvl_f16_products(1): inference under autocast): the leading product only -- the lo fragments are neither read from LDS nor
// multiplied (their planes still travel with the stage: the DMA schedule of seven units per wavefront is the counted one)
template <bool X1>
__global__ void __launch_bounds__(512, 1)
    k_gates_f16x3(const _Float16 *__restrict__ Ah, const _Float16 *__restrict__ Al, const float *__restrict__ As,
                  const _Float16 *__restrict__ B1h, const _Float16 *__restrict__ B1l, const float *__restrict__ B1s,
                  const _Float16 *__restrict__ B2h, const _Float16 *__restrict__ B2l, const float *__restrict__ B2s, int R,
                  int N, int K1, int K2, int tiles_m, int tiles_n, const GatesEpi ge) {
  constexpr int kRowsA = 256, kRowsB = 160, NJ = 5;
  constexpr int kASlots = kRowsA * 4, kBSlots = kRowsB * 4, kStageSlots = 2 * kASlots + 2 * kBSlots;
  constexpr int kLdT = 68;                                             // floats per row of the epilogue's transpose image
  __shared__ uint4 smem[3 * kStageSlots];
  static_assert(2 * kRowsB * kLdT * 4 <= 3 * kStageSlots * 16, "two transpose images fit the stage ring");

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wa = (wave & 3) * 64, wb = (wave >> 2) * 80;
  const int KT1 = K1 / kBK, KT = (K1 + K2) / kBK;

  // DMA units of 16 rows x 64 bytes: unit u = wave + 8 i, i = 0 .. 6: [0, 16) A hi, [16, 32) A lo, [32, 42) B hi, [42, 52) B lo,
  // [52, 56) B lo blocks 6-9 once more (identical bytes to the same place: every wavefront issues seven units per stage)
  const int srow = lane >> 2, schunk = ((lane & 3) ^ (((srow >> 2) & 1) << 1)) * 8;
  const uint32_t lane_off = 2u * (uint32_t)(srow * 32 + schunk);
  const __amdgpu_buffer_rsrc_t rs_ah = __builtin_amdgcn_make_buffer_rsrc((void *)Ah, 0, R * (K1 + K2) * 2, 0x00020000),
                               rs_al = __builtin_amdgcn_make_buffer_rsrc((void *)Al, 0, R * (K1 + K2) * 2, 0x00020000),
                               rs_1h = __builtin_amdgcn_make_buffer_rsrc((void *)B1h, 0, N * K1 * 2, 0x00020000),
                               rs_1l = __builtin_amdgcn_make_buffer_rsrc((void *)B1l, 0, N * K1 * 2, 0x00020000),
                               rs_2h = __builtin_amdgcn_make_buffer_rsrc((void *)B2h, 0, N * K2 * 2, 0x00020000),
                               rs_2l = __builtin_amdgcn_make_buffer_rsrc((void *)B2l, 0, N * K2 * 2, 0x00020000);
  int b_lo[3], b_blk[3];                                               // the B units i = 4, 5, 6 of this wavefront
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int ub = wave + 8 * i;
    b_lo[i] = ub >= 10;
    b_blk[i] = ub < 10 ? ub : (ub < 20 ? ub - 10 : ub - 14);
  }
  // unit I of stage KN (0 .. KT - 1) of tile (TM, TN) -> stage buffer ST
#define GVL_G_DMA(I, TM, TN, KN, ST)                                                                                 \
  {                                                                                                                  \
    if constexpr ((I) < 4) {                                                                                         \
      const bool lo = (I) >= 2;                                                                                      \
      const int blk = wave + 8 * ((I) & 1);                                                                          \
      const uint32_t soff = 2u * (uint32_t)((KN) * kBK * R + ((TM) * kRowsA + blk * 16) * 32);                       \
      __builtin_amdgcn_raw_ptr_buffer_load_lds(lo ? rs_al : rs_ah,                                                   \
                                               (__attribute__((address_space(3))) void *)((ST) + (lo ? kASlots : 0) + blk * 64), \
                                               16, lane_off + soff, 0, 0, 0);                                        \
    } else {                                                                                                         \
      const bool lo = b_lo[(I) - 4], second = (KN) >= KT1;                                                           \
      const int blk = b_blk[(I) - 4], ks = second ? (KN) - KT1 : (KN);                                               \
      const uint32_t soff = 2u * (uint32_t)(ks * kBK * N + ((TN) * kRowsB + blk * 16) * 32);                         \
      __builtin_amdgcn_raw_ptr_buffer_load_lds(second ? (lo ? rs_2l : rs_2h) : (lo ? rs_1l : rs_1h),                 \
                                               (__attribute__((address_space(3))) void *)((ST) + 2 * kASlots + (lo ? kBSlots : 0) + blk * 64), \
                                               16, lane_off + soff, 0, 0, 0);                                        \
    }                                                                                                                \
  }

  const int fa0 = lds_slot16(wa + (lane & 15), lane >> 4), fb0 = 2 * kASlots + lds_slot16(wb + (lane & 15), lane >> 4);
  const _Float16 k2048 = (_Float16)2048.f;
  for (int vb = (int)blockIdx.x;; vb += (int)gridDim.x) {
    int tm, tn;
    if (!tile_of(vb, tiles_m, tiles_n, tm, tn)) break;
    f4acc4 acc[4][NJ];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < NJ; ++j) acc[i][j] = f4acc4{0.f, 0.f, 0.f, 0.f};
    h8 ah[4], al[4], bh[NJ], bl[NJ], nh[2], nl[2];
    // the three stage buffers rotate: cur is read, nxt holds the next stage, nn receives the one after
    uint4 *cur = smem, *nxt = smem + kStageSlots, *nn = smem + 2 * kStageSlots;
    int kt = 0;
    // stage KN beyond the tile's last one: the same units of stage KN - KT again (nobody reads them; the counts stay uniform)
#define GVL_G_ISSUE(I, AHEAD, ST)                                                                                    \
  {                                                                                                                  \
    const int kn = kt + (AHEAD) >= KT ? kt + (AHEAD) - KT : kt + (AHEAD);                                            \
    GVL_G_DMA(I, tm, tn, kn, ST)                                                                                     \
  }
#define GVL_G_RDA(ST, I)                                                                                             \
  {                                                                                                                  \
    ah[I] = *reinterpret_cast<const h8 *>(&(ST)[fa0 + 64 * (I)]);                                                    \
    if constexpr (!X1) al[I] = *reinterpret_cast<const h8 *>(&(ST)[kASlots + fa0 + 64 * (I)]);                       \
  }
#define GVL_G_RDB(ST, J, H_, L_)                                                                                     \
  {                                                                                                                  \
    H_ = *reinterpret_cast<const h8 *>(&(ST)[fb0 + 64 * (J)]);                                                       \
    if constexpr (!X1) L_ = *reinterpret_cast<const h8 *>(&(ST)[kBSlots + fb0 + 64 * (J)]);                          \
  }
#define GVL_G_MFMA3(I, J, BS)                                                                                        \
  {                                                                                                                  \
    acc[I][J] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[I], BS, acc[I][J], 0, 0, 0);                               \
    if constexpr (!X1) {                                                                                             \
      acc[I][J] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[I], bl[J], acc[I][J], 0, 0, 0);                          \
      acc[I][J] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[I], BS, acc[I][J], 0, 0, 0);                             \
    }                                                                                                                \
  }
#define GVL_G_BLOCK(J, PRE, MID, POST)                                                                               \
  {                                                                                                                  \
    PRE                                                                                                              \
    const h8 bs = bh[J] * k2048;                                                                                     \
    GVL_G_MFMA3(0, J, bs)                                                                                            \
    GVL_G_MFMA3(1, J, bs)                                                                                            \
    MID                                                                                                              \
    GVL_G_MFMA3(2, J, bs)                                                                                            \
    GVL_G_MFMA3(3, J, bs)                                                                                            \
    POST                                                                                                             \
    __builtin_amdgcn_sched_barrier(0);                                                                               \
  }
    // one K stage.  On entry: the A fragments (ah not yet scaled) and the B blocks 0, 1 of the stage are in registers.
    // Blocks 0-2 request B blocks 2-4 and issue units 4-6 of the stage after next (its buffer was freed by the last barrier);
    // then every read of `cur` has been issued: barrier; blocks 3, 4 read the next stage's first fragments from `nxt` (A in
    // place, pair by pair) and issue units 0-3 of the stage three ahead into `cur`.
#define GVL_G_STAGE()                                                                                                \
  {                                                                                                                  \
    GVL_G_BLOCK(0, GVL_G_RDB(cur, 2, bh[2], bl[2]) GVL_G_ISSUE(4, 2, nn) ah[0] = ah[0] * k2048; ah[1] = ah[1] * k2048;, \
                __builtin_amdgcn_sched_barrier(0); ah[2] = ah[2] * k2048; ah[3] = ah[3] * k2048;, )                  \
    GVL_G_BLOCK(1, GVL_G_RDB(cur, 3, bh[3], bl[3]) GVL_G_ISSUE(5, 2, nn), , )                                        \
    GVL_G_BLOCK(2, GVL_G_RDB(cur, 4, bh[4], bl[4]) GVL_G_ISSUE(6, 2, nn), , )                                        \
    asm volatile("s_waitcnt vmcnt(7) lgkmcnt(0)" ::: "memory");                                                      \
    __builtin_amdgcn_s_barrier();                                                                                    \
    asm volatile("" ::: "memory");                                                                                   \
    GVL_G_BLOCK(3, GVL_G_RDB(nxt, 0, nh[0], nl[0]) GVL_G_ISSUE(0, 3, cur) GVL_G_ISSUE(1, 3, cur), , )                \
    GVL_G_BLOCK(4, GVL_G_RDB(nxt, 1, nh[1], nl[1]) GVL_G_ISSUE(2, 3, cur) GVL_G_ISSUE(3, 3, cur),                    \
                __builtin_amdgcn_sched_barrier(0); GVL_G_RDA(nxt, 0) GVL_G_RDA(nxt, 1),                              \
                __builtin_amdgcn_sched_barrier(0); GVL_G_RDA(nxt, 2) GVL_G_RDA(nxt, 3))                              \
    bh[0] = nh[0]; bl[0] = nl[0]; bh[1] = nh[1]; bl[1] = nl[1];                                                      \
    {                                                                                                                \
      uint4 *t_ = cur;                                                                                               \
      cur = nxt;                                                                                                     \
      nxt = nn;                                                                                                      \
      nn = t_;                                                                                                       \
    }                                                                                                                \
    ++kt;                                                                                                            \
  }

#ifdef GVL_G_STAMPS                                                    /* dev: 10 ns ticks of workgroup 0's phases */
    const uint64_t ts0 = __builtin_amdgcn_s_memrealtime();
#define GVL_G_T(X) const uint64_t X = __builtin_amdgcn_s_memrealtime();
#else
#define GVL_G_T(X)
#endif
    // prologue: stages 0 and 1 whole, units 0-3 of stage 2 (what stage "-1" would have issued under its blocks 3, 4)
    kt = -2;
    GVL_G_ISSUE(0, 2, cur) GVL_G_ISSUE(1, 2, cur) GVL_G_ISSUE(2, 2, cur) GVL_G_ISSUE(3, 2, cur) GVL_G_ISSUE(4, 2, cur)
    GVL_G_ISSUE(5, 2, cur) GVL_G_ISSUE(6, 2, cur)
    kt = -1;
    GVL_G_ISSUE(0, 2, nxt) GVL_G_ISSUE(1, 2, nxt) GVL_G_ISSUE(2, 2, nxt) GVL_G_ISSUE(3, 2, nxt) GVL_G_ISSUE(4, 2, nxt)
    GVL_G_ISSUE(5, 2, nxt) GVL_G_ISSUE(6, 2, nxt)
    kt = 0;
    GVL_G_ISSUE(0, 2, nn) GVL_G_ISSUE(1, 2, nn) GVL_G_ISSUE(2, 2, nn) GVL_G_ISSUE(3, 2, nn)
    float ratio[NJ];                                                   // h's row scale over att's, per column of this lane
    {
      float s1[NJ], s2[NJ];
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const int col = min(tn * kRowsB + wb + 16 * j + (lane & 15), N - 1);
        s1[j] = B1s[col];
        s2[j] = B2s[col];
      }
#pragma unroll
      for (int j = 0; j < NJ; ++j) ratio[j] = s1[j] / s2[j];
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (false)
    wait_vmcnt<11>();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
#pragma unroll
    for (int i = 0; i < 4; ++i) GVL_G_RDA(cur, i)
    GVL_G_RDB(cur, 0, bh[0], bl[0])
    GVL_G_RDB(cur, 1, bh[1], bl[1])
    // the recurrent half (h at its own row scale), then the accumulators move to att's scale, then the attention half: two
    // loops, no branch inside a K loop
    GVL_G_T(ts1)
    for (; kt < KT1;) GVL_G_STAGE()
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i][j] = acc[i][j] * ratio[j];
    for (; kt < KT;) GVL_G_STAGE()
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");          // (the re-fetched stages) -- the ring becomes scratch
    GVL_G_T(ts2)
    __syncthreads();

    // ---- the cell.  Lane (fc, fq): hidden row col0 + 16 j + fc, gates 0-3 of unit (row0 >> 2) + 4 i + fq
    {
      const int fc = lane & 15, fq = lane >> 4, H = ge.H;
      const int row0 = tm * kRowsA + wa, col0 = tn * kRowsB + wb;
      constexpr float kInv22 = 1.f / 4194304.f;
      float *img_c = reinterpret_cast<float *>(smem), *img_h = img_c + kRowsB * kLdT;
      float rs[4][4];
      int g0[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        g0[i] = min(row0 + 16 * i + 4 * fq, R - 4);                    // (4H % 4 == 0: a lane's four gate rows exist or none does)
#pragma unroll
        for (int r = 0; r < 4; ++r) rs[i][r] = As[g0[i] + r] * kInv22;
      }
      int colv[NJ], tokv[NJ];
      float csv[NJ];
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        colv[j] = min(col0 + 16 * j + fc, N - 1);
        tokv[j] = (int)ge.it[colv[j]];
        csv[j] = B2s[colv[j]];
      }
      // two batches (columns j = 0-2 | 3, 4 of the wavefront tile): every load of a batch in flight at once
      float4 gcA[3][4], emA[3][4], gcB[2][4], emB[2][4];
      float cpA[3][4], cpB[2][4];
#define GVL_G_LOAD(GC, EM, CP, J0, JC)                                                                               \
  _Pragma("unroll") for (int jj = 0; jj < (JC); ++jj) _Pragma("unroll") for (int i = 0; i < 4; ++i) {               \
    GC[jj][i] = ge.gates_c ? *reinterpret_cast<const float4 *>(ge.gates_c + (int64_t)colv[(J0) + jj] * ge.ld_c + g0[i]) \
                           : make_float4(0.f, 0.f, 0.f, 0.f);                                                        \
    EM[jj][i] = *reinterpret_cast<const float4 *>(ge.emb + (int64_t)tokv[(J0) + jj] * R + g0[i]);                    \
    CP[jj][i] = ge.c[(int64_t)colv[(J0) + jj] * H + (g0[i] >> 2)];                                                   \
  }
#define GVL_G_CELL(GC, EM, CP, J0, JC)                                                                               \
  _Pragma("unroll") for (int jj = 0; jj < (JC); ++jj) _Pragma("unroll") for (int i = 0; i < 4; ++i) {               \
    const int j = (J0) + jj;                                                                                         \
    const float s0 = rs[i][0] * csv[j], s1 = rs[i][1] * csv[j], s2 = rs[i][2] * csv[j], s3 = rs[i][3] * csv[j];      \
    const float gi = (GC[jj][i].x + acc[i][j][0] * s0) + EM[jj][i].x, gf = (GC[jj][i].y + acc[i][j][1] * s1) + EM[jj][i].y, \
                gg = (GC[jj][i].z + acc[i][j][2] * s2) + EM[jj][i].z, go = (GC[jj][i].w + acc[i][j][3] * s3) + EM[jj][i].w; \
    float cn, hn;                                                                                                    \
    gvl_lstm_point(gi, gf, gg, go, CP[jj][i], cn, hn);                                                               \
    const int at = (wb + 16 * j + fc) * kLdT + (wa >> 2) + 4 * i + fq;                                               \
    img_c[at] = cn;                                                                                                  \
    img_h[at] = hn;                                                                                                  \
  }
      // whole row segments out: thread (slot, u4) takes units 4 u4 .. 4 u4 + 3 of one row per pass.  Rows of the first batch:
      // [0, 48) and [80, 128) of the tile, of the second: [48, 80) and [128, 160)
      const int u4 = tid & 15, unit = tm * (kRowsA >> 2) + 4 * u4;
      auto store_rows = [&](auto second) {
        constexpr bool kB = decltype(second)::value;
        constexpr int kPasses = kB ? 2 : 3, kHalf = kB ? 32 : 48;
        if (unit >= H) return;
#pragma unroll
        for (int it_ = 0; it_ < kPasses; ++it_) {
          const int idx = (tid >> 4) + 32 * it_;
          const int r = kB ? (idx < kHalf ? 48 + idx : 96 + idx) : (idx < kHalf ? idx : idx + 32);
          const int row = tn * kRowsB + r;
          if (row >= N) continue;
          const float4 cv = *reinterpret_cast<const float4 *>(img_c + r * kLdT + 4 * u4);
          float4 hv = *reinterpret_cast<const float4 *>(img_h + r * kLdT + 4 * u4);
          asm volatile("" : "+v"(hv.x), "+v"(hv.y), "+v"(hv.z), "+v"(hv.w));   // (h' as the fp32 numbers stored: see lstm_finish)
          *reinterpret_cast<float4 *>(ge.c_out + (int64_t)row * H + unit) = cv;
          if (ge.h_out) *reinterpret_cast<float4 *>(ge.h_out + (int64_t)row * H + unit) = hv;
          const float hx[4] = {hv.x, hv.y, hv.z, hv.w};
          _Float16 hh[4], hl[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            hh[q] = (_Float16)hx[q];
            hl[q] = (_Float16)((hx[q] - (float)hh[q]) * 2048.f);
          }
          const int64_t pat = plane_off(row, unit, N);
          *reinterpret_cast<uint2 *>(ge.h_hi + pat) = make_uint2(pack2h(hh[0], hh[1]), pack2h(hh[2], hh[3]));
          *reinterpret_cast<uint2 *>(ge.h_lo + pat) = make_uint2(pack2h(hl[0], hl[1]), pack2h(hl[2], hl[3]));
          if (unit == 0) ge.h_scale[row] = 1.f;
        }
      };
      GVL_G_LOAD(gcA, emA, cpA, 0, 3)
      GVL_G_CELL(gcA, emA, cpA, 0, 3)
      GVL_G_LOAD(gcB, emB, cpB, 3, 2)
      GVL_G_CELL(gcB, emB, cpB, 3, 2)
      GVL_G_T(ts3)
      __syncthreads();
      // (letting the first batch's rows leave under the second batch's cells was measured: 70.5-71.9 against 69.5-69.9 us, same box --
      //  the 29 MB of h', c' and planes leave at ~2.7 TB/s whenever they are issued)
      store_rows(std::false_type{});
      store_rows(std::true_type{});
#undef GVL_G_CELL
#undef GVL_G_LOAD
      __syncthreads();                                                 // the images are free before the next tile's DMA
#ifdef GVL_G_STAMPS
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (blockIdx.x == 8 && tid == 0)
        printf("k_gates: prologue %d, K loop %d, cell %d, stores %d ticks of 10 ns\n", (int)(ts1 - ts0), (int)(ts2 - ts1),
               (int)(ts3 - ts2), (int)(__builtin_amdgcn_s_memrealtime() - ts3));
#endif
    }
#undef GVL_G_T
#undef GVL_G_STAGE
#undef GVL_G_BLOCK
#undef GVL_G_MFMA3
#undef GVL_G_RDB
#undef GVL_G_RDA
#undef GVL_G_ISSUE
  }
#undef GVL_G_DMA
}

// partials (chunks, R) of {max, sum exp(v - max), index, -} -> per row argmax and log-softmax at the argmax, plus the
// bookkeeping of one greedy step (gvl_cap.hip: k_row_argmax_lse has the same tail).  Block = 16 rows x 16 chunk groups:
// 16 lanes read 256 contiguous bytes of one chunk row; 300 workgroups at R = 4800.
struct GreedyBook {
  unsigned char *unfinished;
  int64_t *seq;
  float *seq_lp;
  int seq_ld, first;
  unsigned char *alive;      // (1) or null: set to 1 when any row is still unfinished after this step (LSTM_DSA.py:186-187)
};

constexpr int kRedRows = 16, kRedGroups = 16;

__device__ __forceinline__ void greedy_body(int bid, const float4 *__restrict__ part, int R, int chunks,
                                            int64_t *__restrict__ idx, float *__restrict__ logp, const GreedyBook &book) {
  __shared__ float s_m[kRedGroups][kRedRows], s_s[kRedGroups][kRedRows];
  __shared__ int s_a[kRedGroups][kRedRows];
  const int rr = threadIdx.x % kRedRows, g = threadIdx.x / kRedRows, row = bid * kRedRows + rr;
  float m = -INFINITY, s = 0.f;
  int a = 0x7fffffff;
  if (row < R) {
    // four partials requested together (the running max / sum chain is serial, the loads need not be)
    for (int c0 = g; c0 < chunks; c0 += 4 * kRedGroups) {
      float4 p[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int c = c0 + u * kRedGroups;
        p[u] = c < chunks ? part[(int64_t)c * R + row] : make_float4(-INFINITY, 0.f, __int_as_float(0x7fffffff), 0.f);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int pa = __float_as_int(p[u].z);
        const float mn = fmaxf(m, p[u].x);
        if (mn > -INFINITY) s = s * __expf(m - mn) + p[u].y * __expf(p[u].x - mn);
        a = (p[u].x > m || (p[u].x == m && pa < a)) ? pa : a;
        m = mn;
      }
    }
  }
  s_m[g][rr] = m; s_s[g][rr] = s; s_a[g][rr] = a;
  __syncthreads();
  if (g == 0 && row < R) {
#pragma unroll
    for (int k = 1; k < kRedGroups; ++k) {
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
      if (unf && book.alive) *book.alive = 1;                           // (every writer stores the same value)
    }
  }
}

__global__ void __launch_bounds__(256) k_greedy_from_partials(const float4 *__restrict__ part, int R, int chunks,
                                                              int64_t *__restrict__ idx, float *__restrict__ logp,
                                                              GreedyBook book) {
  greedy_body((int)blockIdx.x, part, R, chunks, idx, logp, book);
}

// ONE launch for two independent pieces of a token step: the greedy bookkeeping of token t (from the vocabulary product's
// partials) and a plain product over the hidden state for token t + 1 (h2att(h): gvl_gemm_f16x3_f32's four-wavefront form).  The
// product's workgroups come first (they run 16 us, the 300 reduction workgroups 6 us beside them); saves a launch and the
// reduction's time per token.
struct GemmArgs {
  const _Float16 *Ah, *Al;
  const float *As;
  const _Float16 *Bh, *Bl;
  const float *Bs, *bias;
  int R, N, K;
  float *out;
  int64_t ldo;
  int tiles_m, tiles_n, blocks;
};

template <bool X1>
__global__ void __launch_bounds__(256, 2) k_greedy_and_gemm(const GemmArgs ga, const float4 *__restrict__ part, int R, int chunks,
                                                            int64_t *__restrict__ idx, float *__restrict__ logp,
                                                            GreedyBook book) {
  if ((int)blockIdx.x < ga.blocks) {
    const LstmEpi none = {};
    gemm4_body<64, kStore, X1, true>((int)blockIdx.x, ga.Ah, ga.Al, ga.As, ga.Bh, ga.Bl, ga.Bs, ga.bias, ga.R, ga.N, ga.K, ga.out,
                                  ga.ldo, ga.tiles_m, ga.tiles_n, none);
  } else {
    greedy_body((int)blockIdx.x - ga.blocks, part, R, chunks, idx, logp, book);
  }
}

// one persistent workgroup per CU for the eight-wavefront kernel (144 KB of LDS each)
int persistent_grid(int tiles) {
  static int cus = 0;
  if (!cus) {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0)
      n = 256;
    cus = n / 8 * 8 > 0 ? n / 8 * 8 : 8;
  }
  const char *e = gvl::env_str("GVL_GEMM16_GRID");                    // (experiments: 0 = one workgroup per tile)
  const int padded = (tiles + 7) / 8 * 8;
  if (e && atoi(e) == 0) return padded;
  return padded < cus ? padded : cus;
}

// the 16 x 16 x 32 form (two K stages per loop iteration: K % 64 == 0, at least four stages); GVL_GEMM16_MFMA=32 keeps
// the 32 x 32 x 16 kernels for A/B runs
bool use_m16(int K) {
  const char *e = gvl::env_str("GVL_GEMM16_MFMA");
  return !(e && atoi(e) == 32) && K % 64 == 0 && K >= 128;
}

// GVL_VOCAB_FORM: "m16" keeps k_gemm_f16x3_m16 for the vocabulary product, "v" takes k_vocab_f16x3 wherever it applies (A/B
// runs, tests)
int vocab_form() {
  const char *e = gvl::env_str("GVL_VOCAB_FORM");
  return !e ? 0 : (e[0] == 'm' ? 1 : (e[0] == 'v' ? 2 : 0));
}

int check_operands(const char *what, const void *a_hi, const void *a_lo, const float *a_scale, int R, const void *b_hi,
                   const void *b_lo, const float *b_scale, int N, int K) {
  if (R < 0 || N <= 0 || K <= 0 || (K % kBK))
    return fail(GVL_EINVAL, "%s: needs K %% 32 == 0 (got R=%d N=%d K=%d)", what, R, N, K);
  if ((int64_t)(R > N ? R : N) * K >= (int64_t)1 << 31)               // element offsets inside a plane are 32-bit
    return fail(GVL_EINVAL, "%s: an operand plane of more than 2^31 elements (R=%d N=%d K=%d)", what, R, N, K);
  if (R == 0) return 0;
  if (!a_hi || !a_lo || !a_scale || !b_hi || !b_lo || !b_scale) return fail(GVL_EINVAL, "%s: null pointer", what);
  if (((uintptr_t)a_hi | (uintptr_t)a_lo | (uintptr_t)b_hi | (uintptr_t)b_lo) & 15)
    return fail(GVL_EINVAL, "%s: operand planes must be 16-byte aligned", what);
  return 0;
}

}  // namespace

thread_local int gvl16::g_f16_products = 3;

extern "C" int gvl_f16_products(int n) {
  const int prev = gvl16::g_f16_products;
  if (n == 1 || n == 3) gvl16::g_f16_products = n;
  else if (n != 0) return fail(GVL_EINVAL, "gvl_f16_products: 3 (exact split), 1 (leading product only) or 0 (query), got %d", n);
  return prev;
}

extern "C" int gvl_split_rows_f16(const float *x, int R, int K, void *hi, void *lo, float *scale, void *stream) {
  if (R < 0 || K <= 0 || (K & 31)) return fail(GVL_EINVAL, "gvl_split_rows_f16: needs K %% 32 == 0 (got R=%d K=%d)", R, K);
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
  const bool x1 = gvl16::g_f16_products == 1;
  if (int rc = check_operands("gvl_gemm_f16x3_f32", a_hi, a_lo, a_scale, R, b_hi, b_lo, b_scale, N, K)) return rc;
  if (ldo < N) return fail(GVL_EINVAL, "gvl_gemm_f16x3_f32: ldo < N");
  if (R == 0) return 0;
  if (!out) return fail(GVL_EINVAL, "gvl_gemm_f16x3_f32: null pointer");
  const _Float16 *ah = (const _Float16 *)a_hi, *al = (const _Float16 *)a_lo, *bh = (const _Float16 *)b_hi,
                 *bl = (const _Float16 *)b_lo;
  // Eight-wavefront persistent kernel: 256 x 128 tiles when they fill the chip for several rounds, 128 x 128 tiles
  // (wavefront tile 64 x 32) for fewer -- 760 / 608 tiles over 256 CUs are 3 rounds of a half-size tile where the large
  // tile has 2 rounds of a full one.  128 x 64 tiles on the four-wavefront kernel (three workgroups per CU) for the
  // products with few rows.
  if (R >= 1024 && K >= 3 * kBK) {
    const int t_big = ((R + 255) / 256) * ((N + 127) / 128), t_mid = ((R + 127) / 128) * ((N + 127) / 128);
    if (t_big >= 1024) {
      const int tiles_m = (R + 255) / 256, tiles_n = (N + 127) / 128;
      // (the 16 x 16 x 32 form stores 64-byte row segments: 190 against 180 us back to back for 4800 x 512 x 8518 --
      //  it is used where nothing is stored, the argmax form)
      return gvl::launch(GVL_PROF_GEMM16, R, N, "k_gemm_f16x3_w8", (x1 ? k_gemm_f16x3_w8<4, 2, 2, kStore, true> : k_gemm_f16x3_w8<4, 2, 2, kStore, false>),
                         dim3(persistent_grid(tiles_m * tiles_n)), dim3(512), 0, (hipStream_t)stream, ah, al, a_scale, bh,
                         bl, b_scale, bias, R, N, K, out, ldo, tiles_m, tiles_n, LstmEpi{});
    }
    const char *mid_min = gvl::env_str("GVL_GEMM16_MID_MIN");          // (experiments: the fewest 128 x 128 tiles that take this form)
    if (t_mid >= (mid_min ? atoi(mid_min) : 384) && !gvl::env_str("GVL_GEMM16_NO_MID")) {
      const int tiles_m = (R + 127) / 128, tiles_n = (N + 127) / 128;
      return gvl::launch(GVL_PROF_GEMM16, R, N, "k_gemm_f16x3_w8", (x1 ? k_gemm_f16x3_w8<2, 4, 1, kStore, true> : k_gemm_f16x3_w8<2, 4, 1, kStore, false>),
                         dim3(persistent_grid(tiles_m * tiles_n)), dim3(512), 0, (hipStream_t)stream, ah, al, a_scale, bh,
                         bl, b_scale, bias, R, N, K, out, ldo, tiles_m, tiles_n, LstmEpi{});
    }
  }
  const int tiles_m = (R + kBM - 1) / kBM, tiles_n = (N + 63) / 64;
  return gvl::launch(GVL_PROF_GEMM16, R, N, "k_gemm_f16x3", (x1 ? k_gemm_f16x3<64, kStore, true, true> : k_gemm_f16x3<64, kStore, false, true>), dim3((tiles_m * tiles_n + 7) / 8 * 8),
                     dim3(256), 0, (hipStream_t)stream, ah, al, a_scale, bh, bl, b_scale, bias, R, N, K, out, ldo, tiles_m,
                     tiles_n, LstmEpi{});
}

extern "C" int gvl_gemm_f16x3_lstm_f32(const void *a_hi, const void *a_lo, const float *a_scale, int R, const void *w_hi,
                                       const void *w_lo, const float *w_scale, int H, int K, const float *gates_h,
                                       int64_t ld_h, const float *gates_c, int64_t ld_c, const float *emb_gates,
                                       const int64_t *it, const float *c, float *h_out, float *c_out, void *h_hi,
                                       void *h_lo, float *h_scale, void *stream) {
  const bool x1 = gvl16::g_f16_products == 1;
  const int N = 4 * H;
  if (H <= 0 || (H & 31)) return fail(GVL_EINVAL, "gvl_gemm_f16x3_lstm_f32: H must be a positive multiple of 32 (got %d)", H);
  if (int rc = check_operands("gvl_gemm_f16x3_lstm_f32", a_hi, a_lo, a_scale, R, w_hi, w_lo, w_scale, N, K)) return rc;
  if (ld_h < N || (ld_h & 3) || (gates_c && (ld_c < N || (ld_c & 3))))
    return fail(GVL_EINVAL, "gvl_gemm_f16x3_lstm_f32: gate operands need a row stride >= 4H, a multiple of 4");
  if (R == 0) return 0;
  if (!gates_h || !emb_gates || !it || !c || !h_out || !c_out || !h_hi || !h_lo || !h_scale)
    return fail(GVL_EINVAL, "gvl_gemm_f16x3_lstm_f32: null pointer");
  if (((uintptr_t)gates_h | (uintptr_t)gates_c | (uintptr_t)emb_gates) & 15)
    return fail(GVL_EINVAL, "gvl_gemm_f16x3_lstm_f32: gate operands must be 16-byte aligned");
  const _Float16 *ah = (const _Float16 *)a_hi, *al = (const _Float16 *)a_lo, *bh = (const _Float16 *)w_hi,
                 *bl = (const _Float16 *)w_lo;
  const LstmEpi le = {gates_h, ld_h, gates_c, ld_c, emb_gates, it, c, h_out, c_out, (_Float16 *)h_hi, (_Float16 *)h_lo,
                      h_scale, H};
  // The four-wavefront kernel by default (128 x 64 tiles, three workgroups per CU at 142 VGPRs): a tile's cell epilogue --
  // a third of its time -- runs under the other two workgroups' products, where the persistent eight-wavefront kernel
  // (one workgroup per CU, GVL_LSTM_GEMM_FORM=8) leaves the matrix cores idle for it: 60.9 against 65.6 us at 4800 x 512 x
  // 2048 although the plain product is faster on the eight-wavefront kernel
  const char *form = gvl::env_str("GVL_LSTM_GEMM_FORM");
  if (R >= 1024 && K >= 3 * kBK && form && atoi(form) == 8) {
    const int t_big = ((R + 255) / 256) * ((N + 127) / 128), t_mid = ((R + 127) / 128) * ((N + 127) / 128);
    if (t_big >= 1024) {
      const int tiles_m = (R + 255) / 256, tiles_n = (N + 127) / 128;
      return gvl::launch(GVL_PROF_GEMM16, R, N, "k_gemm_f16x3_w8<lstm>", (x1 ? k_gemm_f16x3_w8<4, 2, 2, kLstm, true> : k_gemm_f16x3_w8<4, 2, 2, kLstm, false>),
                         dim3(persistent_grid(tiles_m * tiles_n)), dim3(512), 0, (hipStream_t)stream, ah, al, a_scale, bh,
                         bl, w_scale, (const float *)nullptr, R, N, K, (float *)nullptr, (int64_t)0, tiles_m, tiles_n, le);
    }
    if (t_mid >= 384 && !gvl::env_str("GVL_GEMM16_NO_MID")) {
      const int tiles_m = (R + 127) / 128, tiles_n = (N + 127) / 128;
      return gvl::launch(GVL_PROF_GEMM16, R, N, "k_gemm_f16x3_w8<lstm>", (x1 ? k_gemm_f16x3_w8<2, 4, 1, kLstm, true> : k_gemm_f16x3_w8<2, 4, 1, kLstm, false>),
                         dim3(persistent_grid(tiles_m * tiles_n)), dim3(512), 0, (hipStream_t)stream, ah, al, a_scale, bh,
                         bl, w_scale, (const float *)nullptr, R, N, K, (float *)nullptr, (int64_t)0, tiles_m, tiles_n, le);
    }
  }
  // (128 x 128 tiles on this kernel, two workgroups per CU: 10.44 against 10.17 ms per eval step)
  const int tiles_m = (R + kBM - 1) / kBM, tiles_n = (N + 63) / 64;
  return gvl::launch(GVL_PROF_GEMM16, R, N, "k_gemm_f16x3<lstm>", (x1 ? k_gemm_f16x3<64, kLstm, true> : k_gemm_f16x3<64, kLstm, false>), dim3((tiles_m * tiles_n + 7) / 8 * 8),
                     dim3(256), 0, (hipStream_t)stream, ah, al, a_scale, bh, bl, w_scale, (const float *)nullptr, R, N, K,
                     (float *)nullptr, (int64_t)0, tiles_m, tiles_n, le);
}

// k_gates_f16x3 serves a token step when its 256 x 160 tiles fill most of the chip in whole rounds (cfg A: 8 x 30 = 240 tiles)
extern "C" int gvl_gemm_f16x3_gates_applicable(int n, int H) {
  if (n <= 0 || H <= 0 || (H & 31)) return 0;
  const int tiles = ((4 * H + 255) / 256) * ((n + 159) / 160), cus = persistent_grid(1 << 20);
  const int rounds = (tiles + cus - 1) / cus;
  return 10 * tiles >= 7 * rounds * cus;
}

extern "C" int gvl_gemm_f16x3_gates_f32(const void *a_hi, const void *a_lo, const float *a_scale, const void *hp_hi,
                                        const void *hp_lo, const float *hp_scale, int n, const void *w_hi, const void *w_lo,
                                        const float *w_scale, int H, int K_h, int K_a, const float *gates_c, int64_t ld_c,
                                        const float *emb_gates, const int64_t *it, const float *c, float *h_out, float *c_out,
                                        void *h_hi, void *h_lo, float *h_scale, void *stream) {
  const int N4 = 4 * H, K = K_h + K_a;
  if (H <= 0 || (H & 31)) return fail(GVL_EINVAL, "gvl_gemm_f16x3_gates_f32: H must be a positive multiple of 32 (got %d)", H);
  if (K_h < kBK || K_a < kBK || (K_h % kBK) || (K_a % kBK) || K < 3 * kBK)
    return fail(GVL_EINVAL, "gvl_gemm_f16x3_gates_f32: K_h and K_a must be positive multiples of 32, three stages in all (got %d, %d)", K_h, K_a);
  if (int rc = check_operands("gvl_gemm_f16x3_gates_f32", w_hi, w_lo, w_scale, N4, a_hi, a_lo, a_scale, n, K)) return rc;
  if (gates_c && (ld_c < N4 || (ld_c & 3)))
    return fail(GVL_EINVAL, "gvl_gemm_f16x3_gates_f32: gates_c needs a row stride >= 4H, a multiple of 4");
  if (n == 0) return 0;
  if (!hp_hi || !hp_lo || !hp_scale || !emb_gates || !it || !c || !c_out || !h_hi || !h_lo || !h_scale)   // (h_out: optional)
    return fail(GVL_EINVAL, "gvl_gemm_f16x3_gates_f32: null pointer");
  if (hp_hi == h_hi || hp_lo == h_lo || c == c_out)
    return fail(GVL_EINVAL, "gvl_gemm_f16x3_gates_f32: the new state must not overwrite the one the step reads");
  if ((((uintptr_t)hp_hi | (uintptr_t)hp_lo | (uintptr_t)gates_c | (uintptr_t)emb_gates | (uintptr_t)h_out | (uintptr_t)c_out
        | (uintptr_t)h_hi | (uintptr_t)h_lo) & 15))
    return fail(GVL_EINVAL, "gvl_gemm_f16x3_gates_f32: planes, gate operands and outputs must be 16-byte aligned");
  const GatesEpi ge = {gates_c, ld_c, emb_gates, it, c, h_out, c_out, (_Float16 *)h_hi, (_Float16 *)h_lo, h_scale, H};
  const int tiles_m = (N4 + 255) / 256, tiles_n = (n + 159) / 160;
  const bool x1 = gvl16::g_f16_products == 1;
  return gvl::launch(GVL_PROF_GEMM16, n, N4, x1 ? "k_gates_f16x1" : "k_gates_f16x3", x1 ? k_gates_f16x3<true> : k_gates_f16x3<false>,
                     dim3(persistent_grid(tiles_m * tiles_n)), dim3(512),
                     0, (hipStream_t)stream, (const _Float16 *)w_hi, (const _Float16 *)w_lo, w_scale, (const _Float16 *)hp_hi,
                     (const _Float16 *)hp_lo, hp_scale, (const _Float16 *)a_hi, (const _Float16 *)a_lo, a_scale, N4, n, K_h, K_a,
                     tiles_m, tiles_n, ge);
}

extern "C" int gvl_gemm_f16x3_argmax_chunks(int V) { return V > 0 ? (V + kBM - 1) / kBM * 2 : 0; }

extern "C" int gvl_gemm_f16x3_argmax_f32(const void *x_hi, const void *x_lo, const float *x_scale, int R, const void *w_hi,
                                         const void *w_lo, const float *w_scale, int V, int K, const float *bias,
                                         float *partials, void *stream) {
  const bool x1 = gvl16::g_f16_products == 1;
  if (int rc = check_operands("gvl_gemm_f16x3_argmax_f32", x_hi, x_lo, x_scale, R, w_hi, w_lo, w_scale, V, K)) return rc;
  if (R == 0) return 0;
  if (!partials || ((uintptr_t)partials & 15)) return fail(GVL_EINVAL, "gvl_gemm_f16x3_argmax_f32: partials null / unaligned");
  const _Float16 *xh = (const _Float16 *)x_hi, *xl = (const _Float16 *)x_lo, *wh = (const _Float16 *)w_hi,
                 *wl = (const _Float16 *)w_lo;
  const int tiles_m = (V + kBM - 1) / kBM;                            // 128 vocabulary entries per tile, either form
  const char *aform = gvl::env_str("GVL_ARGMAX_FORM");                // (4: the four-wavefront kernel, A/B runs)
  if (R >= 1024 && K >= 3 * kBK && !(aform && atoi(aform) == 4)) {
    const int tiles_n = (R + 255) / 256;
    if (use_m16(K)) {
      // 256 x 320 tiles with one accumulator (k_vocab_f16x3) when their rounds over the chip cost less than the 128 x 256
      // tiles': a round of the large tile takes 2.3 x a round of the small one (49 against 21 us at K = 512, one box)
      const int vm = (V + 255) / 256, vn = (R + 319) / 320, cus = persistent_grid(1 << 20);
      const int64_t cost_v = (int64_t)((vm * vn + cus - 1) / cus) * 23, cost_m = (int64_t)((tiles_m * tiles_n + cus - 1) / cus) * 10;
      if (vocab_form() == 2 || (vocab_form() == 0 && cost_v < cost_m))
        return gvl::launch(GVL_PROF_GEMM16, R, V, x1 ? "k_vocab_f16x1<argmax>" : "k_vocab_f16x3<argmax>", x1 ? k_vocab_f16x3<true> : k_vocab_f16x3<false>, dim3(persistent_grid(vm * vn)), dim3(512),
                           0, (hipStream_t)stream, wh, wl, w_scale, xh, xl, x_scale, bias, V, R, K, partials, vm, vn,
                           gvl_gemm_f16x3_argmax_chunks(V));
    }
    if (use_m16(K))
      return gvl::launch(GVL_PROF_GEMM16, R, V, "k_gemm_f16x3_m16<argmax>", (x1 ? k_gemm_f16x3_m16<2, 4, kArgmax, true> : k_gemm_f16x3_m16<2, 4, kArgmax, false>),
                         dim3(persistent_grid(tiles_m * tiles_n)), dim3(512), 0, (hipStream_t)stream, wh, wl, w_scale, xh,
                         xl, x_scale, bias, V, R, K, partials, (int64_t)0, tiles_m, tiles_n);
    return gvl::launch(GVL_PROF_GEMM16, R, V, "k_gemm_f16x3_w8<argmax>", (x1 ? k_gemm_f16x3_w8<2, 4, 2, kArgmax, true> : k_gemm_f16x3_w8<2, 4, 2, kArgmax, false>),
                       dim3(persistent_grid(tiles_m * tiles_n)), dim3(512), 0, (hipStream_t)stream, wh, wl, w_scale, xh, xl,
                       x_scale, bias, V, R, K, partials, (int64_t)0, tiles_m, tiles_n, LstmEpi{});
  }
  const int tiles_n = (R + 63) / 64;
  return gvl::launch(GVL_PROF_GEMM16, R, V, "k_gemm_f16x3<argmax>", (x1 ? k_gemm_f16x3<64, kArgmax, true> : k_gemm_f16x3<64, kArgmax, false>),
                     dim3((tiles_m * tiles_n + 7) / 8 * 8), dim3(256), 0, (hipStream_t)stream, wh, wl, w_scale, xh, xl,
                     x_scale, bias, V, R, K, partials, (int64_t)0, tiles_m, tiles_n, LstmEpi{});
}

extern "C" int gvl_greedy_step_partials_f32(const float *partials, int R, int V, int first_step, int64_t *token,
                                            float *logp, unsigned char *unfinished, int64_t *seq_col, float *seq_lp_col,
                                            int seq_ld, void *stream) {
  return gvl_greedy_step_partials_alive_f32(partials, R, V, first_step, token, logp, unfinished, seq_col, seq_lp_col, seq_ld,
                                            nullptr, stream);
}

extern "C" int gvl_greedy_step_partials_alive_f32(const float *partials, int R, int V, int first_step, int64_t *token,
                                                  float *logp, unsigned char *unfinished, int64_t *seq_col,
                                                  float *seq_lp_col, int seq_ld, unsigned char *alive, void *stream) {
  if (R < 0 || V <= 0 || (unfinished && seq_ld <= 0)) return fail(GVL_EINVAL, "gvl_greedy_step_partials_f32: bad sizes");
  if (R == 0) return 0;
  if (!partials || !token || !logp || (unfinished && (!seq_col || !seq_lp_col)))
    return fail(GVL_EINVAL, "gvl_greedy_step_partials_f32: null pointer");
  const GreedyBook book = {unfinished, seq_col, seq_lp_col, seq_ld, first_step != 0, unfinished ? alive : nullptr};
  return gvl::launch(GVL_PROF_ROW_ARGMAX, R, V, "k_greedy_from_partials", k_greedy_from_partials, dim3((R + kRedRows - 1) / kRedRows),
                     dim3(256), 0, (hipStream_t)stream, (const float4 *)partials, R, gvl_gemm_f16x3_argmax_chunks(V), token,
                     logp, book);
}

extern "C" int gvl_greedy_step_partials_gemm_f32(const float *partials, int R, int V, int first_step, int64_t *token, float *logp,
                                                 unsigned char *unfinished, int64_t *seq_col, float *seq_lp_col, int seq_ld,
                                                 unsigned char *alive, const void *a_hi, const void *a_lo, const float *a_scale,
                                                 int Ra, const void *b_hi, const void *b_lo, const float *b_scale, int Nb, int K,
                                                 const float *bias, float *out, int64_t ldo, void *stream) {
  if (R <= 0 || V <= 0 || (unfinished && seq_ld <= 0)) return fail(GVL_EINVAL, "gvl_greedy_step_partials_gemm_f32: bad sizes");
  if (!partials || !token || !logp || (unfinished && (!seq_col || !seq_lp_col)))
    return fail(GVL_EINVAL, "gvl_greedy_step_partials_gemm_f32: null pointer");
  if (int rc = check_operands("gvl_greedy_step_partials_gemm_f32", a_hi, a_lo, a_scale, Ra, b_hi, b_lo, b_scale, Nb, K)) return rc;
  if (Ra <= 0 || !out || ldo < Nb) return fail(GVL_EINVAL, "gvl_greedy_step_partials_gemm_f32: product output missing / ldo < N");
  const GreedyBook book = {unfinished, seq_col, seq_lp_col, seq_ld, first_step != 0, unfinished ? alive : nullptr};
  GemmArgs ga;
  ga.Ah = (const _Float16 *)a_hi; ga.Al = (const _Float16 *)a_lo; ga.As = a_scale;
  ga.Bh = (const _Float16 *)b_hi; ga.Bl = (const _Float16 *)b_lo; ga.Bs = b_scale; ga.bias = bias;
  ga.R = Ra; ga.N = Nb; ga.K = K; ga.out = out; ga.ldo = ldo;
  ga.tiles_m = (Ra + kBM - 1) / kBM; ga.tiles_n = (Nb + 63) / 64;
  ga.blocks = (ga.tiles_m * ga.tiles_n + 7) / 8 * 8;
  return gvl::launch(GVL_PROF_ROW_ARGMAX, R, V, "k_greedy_and_gemm", gvl16::g_f16_products == 1 ? k_greedy_and_gemm<true> : k_greedy_and_gemm<false>,
                     dim3(ga.blocks + (R + kRedRows - 1) / kRedRows), dim3(256), 0, (hipStream_t)stream, ga,
                     (const float4 *)partials, R, gvl_gemm_f16x3_argmax_chunks(V), token, logp, book);
}
