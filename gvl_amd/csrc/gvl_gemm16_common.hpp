// gvl_gemm16_common.hpp -- device helpers shared by the split-fp16 GEMM kernels (gvl_gemm16.hip: token loop of the
// captioner; gvl_layers.hip: the transformer layers' Linear products).  Not part of the C ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace gvl16 {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef float f16acc __attribute__((ext_vector_type(16)));

constexpr float kLoScale = 2048.f, kLoInv = 1.f / 2048.f;

// PLANE LAYOUT (round 4): an (R, K) operand plane is stored K-STAGE-MAJOR, [K / 32][R][32]: the 32 halves (64 bytes) that one K
// stage takes from a row lie beside those of the next ROW, so the 16 rows x 64 bytes one staging instruction moves are ONE
// contiguous 1 KiB run (8 whole 128-byte lines) -- row-major planes made it 16 half lines at a 2 K-byte stride, twice the
// requests into the L2 (vocabulary product 151 -> 133 us, single-product form 76 -> 69 us, same run; tools/kmaj_probe.py).
// Element (row, k) of a plane of `rows` rows:
__host__ __device__ __forceinline__ int64_t plane_off(int row, int k, int rows) {
  return ((int64_t)(k >> 5) * rows + row) * 32 + (k & 31);
}

extern thread_local int g_f16_products;      // 3 | 1 (gvl_f16_products; defined in gvl_gemm16.hip)

// ---------------------------------------------------------------------------------------------------------------------
#ifndef GVL_GROUPM
#define GVL_GROUPM 8
#endif
constexpr int kBM = 128, kBK = 32, kGroupM = GVL_GROUPM;

__device__ __forceinline__ int lds_slot(int row, int chunk) { return row * 4 + (chunk ^ ((row >> 2) & 3)); }

// 16 bytes per lane, global -> LDS without a register: the wavefront's 64 lanes land at lds .. lds + 1 KiB, lane-linear
__device__ __forceinline__ void glds16(const void *g, void *lds) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g,
                                   (__attribute__((address_space(3))) void *)lds, 16, 0, 0);
}

// the same with the source given as a wave-uniform base + an UNSIGNED 32-bit byte offset per lane: the instruction then takes
// the base from scalar registers and the offset from one vector register (global_load_lds_dwordx4 v_off, s[base:base+1]) -- a signed
// element index costs three more vector instructions per DMA (sign extension, 64-bit shift, 64-bit add)
__device__ __forceinline__ void glds16_at(const _Float16 *base, uint32_t byte_off, void *lds) {
  glds16(reinterpret_cast<const char *>(base) + (size_t)byte_off, lds);
}

// What happens to a tile of D = A . B^T once it is complete:
//   kStore       out[row][col] = D            (A = activations, B = weight; bias per column)
//   kArgmax      A = the vocabulary layer's weight, B = the hidden states: per (token row = column of D, 64 vocabulary
//                entries = the wavefront's rows of D) the maximum, its index and sum exp(v - max) -- the logits are never
//                written.  All 32 vocabulary entries of one MFMA tile column sit in ONE lane's registers (and the lane
//                32 further): the reduction is register-local plus one cross-lane step.
//   kLstm        D = the attention part of the LSTM gate pre-activations, never written: the weight rows arrive in the
//                order 4 unit + gate (i f g o), so the four gates of a hidden unit are four neighbouring columns = the
//                four lanes of a quad; a 4 x 4 transpose inside the quad (rows of D <-> gates) leaves every lane with
//                the four gates of ONE (row, unit), the other gate parts are added and the cell is applied in place
//                (LstmEpi below) -- the pointwise kernel and the (n, 4H) tensor between the two disappear.
enum { kStore = 0, kArgmax = 2, kLstm = 3 };

// operands of the kLstm epilogue: the other parts of the gate pre-activations in the SAME permuted column order
// (4 unit + gate), the input token's row of the pre-multiplied embedding table, the state
struct LstmEpi {
  const float *gates_h;      // (R, >= 4H) h W_hh^T part
  int64_t ld_h;
  const float *gates_c;      // (R, >= 4H) token-independent part (event features), may be null
  int64_t ld_c;
  const float *emb;          // (V + 1, 4H) embedding table x W_ih[:, :E]^T
  const int64_t *it;         // (R) input token
  const float *c;            // (R, H)
  float *h_out, *c_out;      // (R, H)
  _Float16 *h_hi, *h_lo;     // (R, H) planes of h' at row scale 1
  float *h_scale;            // (R)
  int H;
};
// (a transposed store -- weight on the row side, one 16-byte store per 4 values of a lane -- was measured: 3 us SLOWER
//  than the 4-byte stores of 128-byte row segments on the 4800 x 2560 / 2048 outputs)

// tile (row0, col0) of workgroup `bid`: XCD x (= bid % 8) walks the contiguous range [x per, (x + 1) per) of the tile order
// "groups of kGroupM row tiles, column-major inside a group"
__device__ __forceinline__ bool tile_of(int bid, int tiles_m, int tiles_n, int &tm, int &tn) {
  const int total = tiles_m * tiles_n, per = (total + 7) >> 3;
  const int t = (bid & 7) * per + (bid >> 3);
  if ((bid >> 3) >= per || t >= total) return false;
  const int gsz_full = kGroupM * tiles_n, g = t / gsz_full, first_m = g * kGroupM;
  const int gm = min(tiles_m - first_m, kGroupM), in_g = t - g * gsz_full;
  tm = first_m + in_g % gm;
  tn = in_g / gm;
  return true;
}

// Number of fp16 products per fp32 product (gvl_f16_products): 3 = the exact split (hi.hi, hi.lo, lo.hi), 1 = the leading
// product only -- operands rounded to fp16 at their row scale (11 significant bits; bf16 keeps 8), fp32 accumulation: what
// inference under torch.autocast runs on.  X1 kernels neither fetch nor read the lo planes.
// the 24 (NJ = 2) MFMAs of one wavefront on one K stage of 32: fragments at slots fa / fb of the stage image `st`
template <int NJ, bool X1 = false>
__device__ __forceinline__ void mfma_stage(const uint4 *st, int a_lo_off, int b_lo_off, const int (&fa)[2][2],
                                           const int (&fb)[NJ][2], f16acc (&acc_m)[2][NJ], f16acc (&acc_x)[2][NJ]) {
  // all fragment reads of the stage first: the second K half lands under the MFMAs of the first
  h8 f_ah[2][2], f_al[2][2], f_bh[2][NJ], f_bl[2][NJ];
#ifdef GVL_ABLATE_LDS
  st = reinterpret_cast<const uint4 *>(__builtin_assume_aligned(st, 16));
#pragma unroll
  for (int s = 0; s < 2; ++s) {
#pragma unroll
    for (int i = 0; i < 2; ++i) { f_ah[s][i] = __builtin_bit_cast(h8, make_uint4(fa[i][s], 1, 2, 3)); f_al[s][i] = f_ah[s][i]; }
#pragma unroll
    for (int j = 0; j < NJ; ++j) { f_bh[s][j] = __builtin_bit_cast(h8, make_uint4(fb[j][s], 5, 6, 7)); f_bl[s][j] = f_bh[s][j]; }
  }
  if (false)
#endif
#pragma unroll
  for (int s = 0; s < 2; ++s) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      f_ah[s][i] = *reinterpret_cast<const h8 *>(&st[fa[i][s]]);
      if constexpr (!X1) f_al[s][i] = *reinterpret_cast<const h8 *>(&st[a_lo_off + fa[i][s]]);
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      f_bh[s][j] = *reinterpret_cast<const h8 *>(&st[fb[j][s]]);
      if constexpr (!X1) f_bl[s][j] = *reinterpret_cast<const h8 *>(&st[b_lo_off + fb[j][s]]);
    }
  }
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        acc_m[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f_ah[s][i], f_bh[s][j], acc_m[i][j], 0, 0, 0);
        if constexpr (!X1) {
          acc_x[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f_ah[s][i], f_bl[s][j], acc_x[i][j], 0, 0, 0);
          acc_x[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f_al[s][i], f_bh[s][j], acc_x[i][j], 0, 0, 0);
        }
      }
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

}  // namespace gvl16
