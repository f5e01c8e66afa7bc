// gvl_proj.hip -- the sampling-offset / attention-weight projection of MSDeformAttn as a hand-written fp32 MFMA GEMM.
//
// Reference: `self.sampling_offsets(query)` and `self.attention_weights(query)` (pdvc/ops/modules/ms_deform_attn.py:99-100),
// two nn.Linear(512 -> M*L*P = 128) on the same input.  gvl_amd issues them as ONE product against the concatenated
// weight [sampling_offsets.weight; attention_weights.weight] (256 x 512); its softmax / location epilogue lives in the
// sampling kernel (gvl_msda.hip, FUSED).  The product itself:
//
//     out[R, N] = X[R, K] . W[N, K]^T + bias[N]          R = B*Lq = 4800 | 3008 rows, K = 512, N = 256   (1.26 GFLOP)
//
// Arithmetic is exact fp32 (v_mfma_f32_16x16x4_f32: an fmaf chain per output, no xf32 on gfx950), so the positions the
// sampling kernel derives from it keep the 1e-4 contract.
//
// Shape of the kernel.  fp32 MFMA runs at the VECTOR rate (64 FLOP / clk / SIMD), so the product is 8 us of matrix-core
// time spread perfectly over the chip and the only question is granularity: 4800 rows over 256 CUs do not divide into
// 32- or 64-row tiles evenly (300 tiles of 32 x 128 -> two rounds on 44 CUs = 13.6 us).  Hence SMALL wave tiles:
//   workgroup = 4 wavefronts, tile 32 rows x 64 columns; wavefront w owns rows 16 (w & 1) and columns 32 (w >> 1) .. +32
//   as two 16 x 16 accumulators; 600 workgroups, 53 KB of LDS each -> three resident per CU, 2400 wave tiles of 3.4 us
//   of MFMA time over 1024 SIMDs (at most 3 per SIMD: 10.2 us).
//   K is walked in chunks of 64 through a double-buffered LDS image (plain row-major rows of 64 floats, padded): the
//   next chunk's six float4 per thread are requested before the current chunk's MFMAs and written -- one ds_write_b128
//   each -- to the other buffer after them (one barrier per chunk).
//   Operand order.  A 16x16x4 MFMA step takes A[row][k], B[k][col] for four k, lane group lk = lane >> 4 supplying one
//   of them.  WHICH four elements of the chunk form a step is free as long as A and B agree (the sum over the chunk is
//   the same set of products): step (q, c) of lane group lk uses element 16 q + 4 lk + c of the row, q, c = 0..3.  Every
//   lane then reads four float4 of its row per chunk -- four ds_read_b128 per operand instead of sixteen ds_read_b32.  (Measured on the way: with ds_read_b32 operands the LDS
//   reads alone cost as much as the MFMAs and did not overlap them: 24 us against the library's 16.5.)
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gvl_common.hpp"
#include "gvl_msda.h"

namespace {

using gvl::fail;

constexpr int kTM = 32, kTN = 64, kTK = 64;
// row pitches in floats (multiples of 4: b128-aligned rows).  B at 72 makes its ds_read_b128 conflict-free for the lane
// groups the LDS serves together (searched over pitches); A stays at 68 (2-way on one read in three) so that a workgroup
// needs 54,272 B and THREE fit the 160 KB of a CU
constexpr int kPitchA = kTK + 4, kPitchB = kTK + 8;
constexpr int kAhead = 3;                                           // chunks of global loads in flight per thread
typedef float f4acc __attribute__((ext_vector_type(4)));

struct Stage {                                                      // one chunk's share of a thread: 2 float4 of X, 4 of W
  float4 a0, a1, b0, b1, b2, b3;
};

// NCHUNK = K / 64 is a compile-time constant: the chunk loop is fully unrolled so that the three rotating register sets
// of the load pipeline are plain registers
template <int NCHUNK>
__global__ void __launch_bounds__(256) k_proj_f32(const float *__restrict__ X, const float *__restrict__ W,
                                                  const float *__restrict__ bias, int R, int N,
                                                  float *__restrict__ out) {
  constexpr int K = NCHUNK * kTK;
  __shared__ __attribute__((aligned(16))) float A_s[2][kTM][kPitchA];
  __shared__ __attribute__((aligned(16))) float B_s[2][kTN][kPitchB];
  const int r0 = blockIdx.x * kTM, n0 = blockIdx.y * kTN;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wr = (wave & 1) * 16, wc = (wave >> 1) * 32;            // this wavefront's 16 x 32 part of the tile
  const int li = lane & 15, lk = lane >> 4;                          // MFMA operand lane map: A[li][lk], B[lk][li]

  // staging map: a float4 = 4 consecutive k of one row, 16 float4 per row; thread t serves row t >> 4 of every 16-row
  // slice (2 slices of X, 4 of W).  Rows past R re-read the last row (their results are not stored).
  const int srow = threadIdx.x >> 4, sk = (threadIdx.x & 15) * 4;
  const float *xa0 = X + (int64_t)min(r0 + srow, R - 1) * K + sk, *xa1 = X + (int64_t)min(r0 + 16 + srow, R - 1) * K + sk;
  const float *wb0 = W + (int64_t)(n0 + srow) * K + sk;
  constexpr int64_t w16 = (int64_t)16 * K;
  auto fetch = [&](int kc) {
    Stage t;
    t.a0 = *reinterpret_cast<const float4 *>(xa0 + kc * kTK);
    t.a1 = *reinterpret_cast<const float4 *>(xa1 + kc * kTK);
    t.b0 = *reinterpret_cast<const float4 *>(wb0 + kc * kTK);
    t.b1 = *reinterpret_cast<const float4 *>(wb0 + w16 + kc * kTK);
    t.b2 = *reinterpret_cast<const float4 *>(wb0 + 2 * w16 + kc * kTK);
    t.b3 = *reinterpret_cast<const float4 *>(wb0 + 3 * w16 + kc * kTK);
    return t;
  };
  auto stash = [&](const Stage &t, int buf) {
    *reinterpret_cast<float4 *>(&A_s[buf][srow][sk]) = t.a0;
    *reinterpret_cast<float4 *>(&A_s[buf][16 + srow][sk]) = t.a1;
    *reinterpret_cast<float4 *>(&B_s[buf][srow][sk]) = t.b0;
    *reinterpret_cast<float4 *>(&B_s[buf][16 + srow][sk]) = t.b1;
    *reinterpret_cast<float4 *>(&B_s[buf][32 + srow][sk]) = t.b2;
    *reinterpret_cast<float4 *>(&B_s[buf][48 + srow][sk]) = t.b3;
  };

  f4acc acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
  // load pipeline: kAhead chunks requested before the first one is used; chunk c lives in set[c % kAhead]
  Stage set[kAhead];
#pragma unroll
  for (int c = 0; c < kAhead; ++c)
    if (c < NCHUNK) set[c] = fetch(c);
  stash(set[0], 0);
  __syncthreads();
#pragma unroll
  for (int kc = 0; kc < NCHUNK; ++kc) {
    const int buf = kc & 1;
    // this lane's elements of its A row and of its two B rows: float4 number 4 q + lk of the row, q = 0..3
    const float4 *a_p = reinterpret_cast<const float4 *>(&A_s[buf][wr + li][4 * lk]);
    const float4 *b0_p = reinterpret_cast<const float4 *>(&B_s[buf][wc + li][4 * lk]);
    const float4 *b1_p = reinterpret_cast<const float4 *>(&B_s[buf][wc + 16 + li][4 * lk]);
    float4 av[4], b0v[4], b1v[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) { av[q] = a_p[4 * q]; b0v[q] = b0_p[4 * q]; b1v[q] = b1_p[4 * q]; }
    __builtin_amdgcn_sched_barrier(0);          // all twelve reads issued: the later ones land under the first MFMAs
#define GVL_PROJ_STEP(Q, C)                                                          \
  acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[Q].C, b0v[Q].C, acc0, 0, 0, 0);     \
  acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[Q].C, b1v[Q].C, acc1, 0, 0, 0);
#define GVL_PROJ_QUAD(Q) GVL_PROJ_STEP(Q, x) GVL_PROJ_STEP(Q, y) GVL_PROJ_STEP(Q, z) GVL_PROJ_STEP(Q, w)
    GVL_PROJ_QUAD(0) GVL_PROJ_QUAD(1) GVL_PROJ_QUAD(2) GVL_PROJ_QUAD(3)
#undef GVL_PROJ_QUAD
#undef GVL_PROJ_STEP
    if (kc + 1 < NCHUNK) stash(set[(kc + 1) % kAhead], buf ^ 1);    // requested kAhead - 1 chunks ago
    if (kc + kAhead < NCHUNK) set[kc % kAhead] = fetch(kc + kAhead); // its set was stashed one iteration ago
    __syncthreads();
  }
  // C/D map of the 16 x 16 MFMA: column = lane & 15, row = 4 (lane >> 4) + register
  const int col = lane & 15, rbase = (lane >> 4) * 4;
  const float bb0 = bias ? bias[n0 + wc + col] : 0.f, bb1 = bias ? bias[n0 + wc + 16 + col] : 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = r0 + wr + rbase + i;
    if (row < R) {
      float *o = out + (int64_t)row * N + n0 + wc + col;
      o[0] = acc0[i] + bb0;
      o[16] = acc1[i] + bb1;
    }
  }
}

}  // namespace

extern "C" int gvl_proj_f32(const float *x, const float *weight, const float *bias, int R, int K, int N, float *out,
                            void *stream) {
  if (R < 0 || (K != 256 && K != 512 && K != 1024) || N <= 0 || (N % kTN))
    return fail(GVL_EINVAL, "gvl_proj_f32: needs K in {256, 512, 1024} and N %% 64 == 0 (got R=%d K=%d N=%d)", R, K, N);
  if (R == 0) return 0;
  if (!x || !weight || !out) return fail(GVL_EINVAL, "gvl_proj_f32: null pointer");
  if (((uintptr_t)x & 15) || ((uintptr_t)weight & 15))
    return fail(GVL_EINVAL, "gvl_proj_f32: x / weight must be 16-byte aligned");
  const dim3 grid((R + kTM - 1) / kTM, N / kTN);
  auto kern = K == 512 ? k_proj_f32<8> : (K == 256 ? k_proj_f32<4> : k_proj_f32<16>);
  return gvl::launch(GVL_PROF_PROJ, R, N, "k_proj_f32", kern, grid, dim3(256), 0, (hipStream_t)stream, x, weight, bias, R,
                     N, out);
}
