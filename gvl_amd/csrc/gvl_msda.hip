// gvl_msda.hip -- multi-scale deformable attention for GVL on MI355X (gfx950, CDNA4).
//
// Written for wave64 / LDS / DPP directly; no CUDA compatibility layer.  Two kernel families:
//
//   generic   any D, any HxW levels, fp32 / fp64, zeros / border padding.  Forward: one lane per output
//             scalar.  Backward: one wavefront per (b,q,m), lanes stride the channels, wave reduction of
//             grad_loc / grad_attn with cross-lane shuffles, grad_value through hardware float atomics.
//
//   t1d_d64   GVL's case: temporal levels (H = 1), D = 64, L*P <= 16; fp32 arithmetic over fp32 or bf16 storage
//             (value / out / grad_out / grad_value and, fused, the projection rows).  One workgroup owns one
//             (batch, head) value slab [S][64] staged in LDS as fp32 and a chunk of the queries.  A 16-lane DPP
//             row owns one (b,q,m): lane j holds channels 4j..4j+3 (float4) AND computes the interpolation
//             coefficients of sample j, which are broadcast inside the row with DPP row_newbcast.  Rows are
//             256 B, so one ds_read_b128 per lane reads a whole row per DPP row, bank-conflict free.
//             The backward produces grad_value by a counting sort + gather in LDS (no float atomics) and
//             writes it with plain stores (partial slabs are summed by a second tiny kernel when a slab is
//             shared by several workgroups): grad_loc / grad_attn are bitwise reproducible, grad_value up to the
//             order of the entries of one slab row (integer LDS atomics fix it; ~1e-7 relative).
//
// Arithmetic follows /root/reference/pdvc/ops/src/cuda/ms_deform_im2col_cuda.cuh (cited per function) for
// pad_mode = zeros and ATen's grid_sampler(border, align_corners=False) for pad_mode = border.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <type_traits>

#include "gvl_common.hpp"
#include "gvl_msda.h"

namespace {

using gvl::fail;
using gvl::ensure_lds;

thread_local int g_last_impl = 0;
thread_local const char *g_last_kernel = "";     // kernel form of the most recent launch (gvl_msda_last_kernel)
unsigned long long *g_fwd_stamps = nullptr, *g_bwd_stamps = nullptr;   // diagnostics, see gvl_msda_debug_stamps
int g_impl = -1;  // -1 = read the environment on first use

int impl_mode() {
  if (g_impl < 0) {
    const char *e = gvl::env_str("GVL_MSDA_IMPL");
    g_impl = 0;
    if (e && !strcmp(e, "generic")) g_impl = 1;
    if (e && !strcmp(e, "fast")) g_impl = 2;
  }
  return g_impl;
}

using gvl::env_int;                                  // (cached: gvl_common.hpp)

constexpr int kPadZeros = GVL_PAD_ZEROS;
constexpr int kPadBorder = GVL_PAD_BORDER;

// ------------------------------------------------------------------------------------------------------
// shared scalar math
// ------------------------------------------------------------------------------------------------------
__device__ inline float gfloor(float x) { return floorf(x); }
__device__ inline double gfloor(double x) { return floor(x); }

// Pixel coordinate of a normalised location and d(pixel)/d(loc).
// zeros : cuh:286-287 (loc*size - 0.5).   border: grid_sampler unnormalise + clip_coordinates_set_grad.
template <typename T>
__device__ inline T pixel_coord(T loc, int size, int pad, T &dmul) {
  if (pad == kPadZeros) {
    dmul = (T)size;
    return loc * (T)size - (T)0.5;
  }
  T g = (T)2 * loc - (T)1;
  T x = ((g + (T)1) * (T)size - (T)1) / (T)2;
  T mx = (T)(size - 1);
  if (!(x > (T)0)) { dmul = (T)0; return (T)0; }   // also catches NaN
  if (x >= mx) { dmul = (T)0; return mx; }
  dmul = (T)size;
  return x;
}

// Four bilinear taps (cuh:39-82).  Row indices are -1 where the tap is outside the level.
template <typename T>
struct Taps {
  int i00, i01, i10, i11;
  T w00, w01, w10, w11, lh, lw;
  bool valid;
};

template <typename T>
__device__ inline Taps<T> make_taps(T h, T w, int H, int W, int pad) {
  Taps<T> t;
  t.i00 = t.i01 = t.i10 = t.i11 = -1;
  t.w00 = t.w01 = t.w10 = t.w11 = t.lh = t.lw = (T)0;
  t.valid = (pad == kPadBorder) || (h > (T)-1 && w > (T)-1 && h < (T)H && w < (T)W);   // cuh:289
  if (!t.valid) return t;
  int hl = (int)gfloor(h), wl = (int)gfloor(w);
  int hh_ = hl + 1, wh = wl + 1;
  T lh = h - (T)hl, lw = w - (T)wl, hh = (T)1 - lh, hw = (T)1 - lw;
  t.lh = lh; t.lw = lw;
  t.w00 = hh * hw; t.w01 = hh * lw; t.w10 = lh * hw; t.w11 = lh * lw;
  bool hl_ok = hl >= 0 && hl <= H - 1, hh_ok = hh_ >= 0 && hh_ <= H - 1;
  bool wl_ok = wl >= 0 && wl <= W - 1, wh_ok = wh >= 0 && wh <= W - 1;
  if (hl_ok && wl_ok) t.i00 = hl * W + wl;
  if (hl_ok && wh_ok) t.i01 = hl * W + wh;
  if (hh_ok && wl_ok) t.i10 = hh_ * W + wl;
  if (hh_ok && wh_ok) t.i11 = hh_ * W + wh;
  return t;
}

// ------------------------------------------------------------------------------------------------------
// generic forward: one lane per (b,q,m,d); d fastest so a wave reads contiguous channels of a value row.
// SAMPLE=false: out (B,Q,M*D) (cuh:238-300).  SAMPLE=true: unweighted samples (B*M, D, Q, L, P) (func.py:56-68).
// ------------------------------------------------------------------------------------------------------
template <typename T, bool SAMPLE>
__global__ void __launch_bounds__(256) k_fwd_generic(const T *__restrict__ value, const int64_t *__restrict__ shapes,
                                                     const int64_t *__restrict__ lsi, const T *__restrict__ loc,
                                                     const T *__restrict__ attn, int B, int S, int M, int D, int L,
                                                     int Q, int P, int pad, T *__restrict__ out) {
  const int64_t n = (int64_t)B * Q * M * D;
  const int64_t row = (int64_t)M * D;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += (int64_t)gridDim.x * blockDim.x) {
    const int d = (int)(idx % D);
    const int64_t tup = idx / D;
    const int m = (int)(tup % M);
    const int64_t bq = tup / M;
    const int q = (int)(bq % Q);
    const int b = (int)(bq / Q);
    const int64_t wb = tup * L * P;
    T acc = (T)0;
    for (int l = 0; l < L; ++l) {
      const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];
      const T *vl = value + ((int64_t)b * S + lsi[l]) * row + (int64_t)m * D + d;
      for (int p = 0; p < P; ++p) {
        const int64_t si = wb + l * P + p;
        T dm;
        const T w_im = pixel_coord(loc[si * 2], W, pad, dm);
        const T h_im = pixel_coord(loc[si * 2 + 1], H, pad, dm);
        const Taps<T> t = make_taps(h_im, w_im, H, W, pad);
        T val = (T)0;
        if (t.valid) {
          const T v1 = t.i00 >= 0 ? vl[(int64_t)t.i00 * row] : (T)0;
          const T v2 = t.i01 >= 0 ? vl[(int64_t)t.i01 * row] : (T)0;
          const T v3 = t.i10 >= 0 ? vl[(int64_t)t.i10 * row] : (T)0;
          const T v4 = t.i11 >= 0 ? vl[(int64_t)t.i11 * row] : (T)0;
          val = t.w00 * v1 + t.w01 * v2 + t.w10 * v3 + t.w11 * v4;   // cuh:82
        }
        if (SAMPLE)
          out[(((((int64_t)b * M + m) * D + d) * Q + q) * L + l) * P + p] = val;
        else
          acc += val * attn[si];                                       // cuh:291
      }
    }
    if (!SAMPLE) out[idx] = acc;
  }
}

// ------------------------------------------------------------------------------------------------------
// generic backward: one wavefront per (b,q,m) (cuh:407-511 uses a D-thread block + shared-memory tree; with
// wave64 the reduction is a cross-lane butterfly and needs no LDS or barrier).
// ------------------------------------------------------------------------------------------------------
template <typename T>
__device__ inline T wave_sum(T v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

template <typename T>
__global__ void __launch_bounds__(256) k_bwd_generic(const T *__restrict__ value, const int64_t *__restrict__ shapes,
                                                     const int64_t *__restrict__ lsi, const T *__restrict__ loc,
                                                     const T *__restrict__ attn, const T *__restrict__ gout, int B,
                                                     int S, int M, int D, int L, int Q, int P, int pad,
                                                     T *__restrict__ gvalue, T *__restrict__ gloc,
                                                     T *__restrict__ gattn) {
  const int lane = threadIdx.x & 63;
  const int wpb = blockDim.x >> 6;
  const int64_t ntup = (int64_t)B * Q * M;
  const int64_t row = (int64_t)M * D;
  for (int64_t tup = (int64_t)blockIdx.x * wpb + (threadIdx.x >> 6); tup < ntup; tup += (int64_t)gridDim.x * wpb) {
    const int m = (int)(tup % M);
    const int64_t bq = tup / M;
    const int b = (int)(bq / Q);
    const T *go = gout + tup * D;
    const int64_t wb = tup * L * P;
    for (int l = 0; l < L; ++l) {
      const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];
      const int64_t voff = ((int64_t)b * S + lsi[l]) * row + (int64_t)m * D;
      const T *vl = value + voff;
      T *gvl = gvalue + voff;
      for (int p = 0; p < P; ++p) {
        const int64_t si = wb + l * P + p;
        T dmx, dmy;
        const T w_im = pixel_coord(loc[si * 2], W, pad, dmx);
        const T h_im = pixel_coord(loc[si * 2 + 1], H, pad, dmy);
        const T wgt = attn[si];
        const Taps<T> t = make_taps(h_im, w_im, H, W, pad);
        T acc_w = (T)0, acc_x = (T)0, acc_y = (T)0;
        if (t.valid) {
          const T hh = (T)1 - t.lh, hw = (T)1 - t.lw;
          for (int d = lane; d < D; d += 64) {
            const T tg = go[d], tgv = tg * wgt;                       // cuh:111
            T gh = (T)0, gw = (T)0, v1 = (T)0, v2 = (T)0, v3 = (T)0, v4 = (T)0;
            if (t.i00 >= 0) { v1 = vl[(int64_t)t.i00 * row + d]; gh -= hw * v1; gw -= hh * v1;
                              atomicAdd(gvl + (int64_t)t.i00 * row + d, t.w00 * tgv); }
            if (t.i01 >= 0) { v2 = vl[(int64_t)t.i01 * row + d]; gh -= t.lw * v2; gw += hh * v2;
                              atomicAdd(gvl + (int64_t)t.i01 * row + d, t.w01 * tgv); }
            if (t.i10 >= 0) { v3 = vl[(int64_t)t.i10 * row + d]; gh += hw * v3; gw -= t.lh * v3;
                              atomicAdd(gvl + (int64_t)t.i10 * row + d, t.w10 * tgv); }
            if (t.i11 >= 0) { v4 = vl[(int64_t)t.i11 * row + d]; gh += t.lw * v4; gw += t.lh * v4;
                              atomicAdd(gvl + (int64_t)t.i11 * row + d, t.w11 * tgv); }
            const T val = t.w00 * v1 + t.w01 * v2 + t.w10 * v3 + t.w11 * v4;
            acc_w += tg * val;                                         // cuh:156-157
            acc_x += dmx * gw * tgv;                                   // cuh:158
            acc_y += dmy * gh * tgv;                                   // cuh:159
          }
        }
        acc_w = wave_sum(acc_w);
        acc_x = wave_sum(acc_x);
        acc_y = wave_sum(acc_y);
        if (lane == 0) {
          gattn[si] = acc_w;
          gloc[si * 2] = acc_x;
          gloc[si * 2 + 1] = acc_y;
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------------
// DPP helpers (a DPP "row" = 16 lanes = one (b,q,m) in the t1d_d64 kernels)
// ------------------------------------------------------------------------------------------------------
template <int CTRL>
__device__ inline int dpp_i(int v) {
  // bound_ctrl = true: every control used here (row_newbcast, quad_perm, row_ror) reads a valid lane, and without it
  // the compiler must materialise the `old` operand (a v_mov 0 per DPP instruction: 48 per forward pass)
  return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, true);
}
template <int CTRL>
__device__ inline float dpp_f(float v) {
  return __builtin_bit_cast(float, dpp_i<CTRL>(__builtin_bit_cast(int, v)));
}
// broadcast lane SRC of every 16-lane row to the whole row (row_newbcast, gfx90a+)
template <int SRC>
__device__ inline int row_bcast_i(int v) { return dpp_i<0x150 + SRC>(v); }
template <int SRC>
__device__ inline float row_bcast_f(float v) { return dpp_f<0x150 + SRC>(v); }
// 64-bit row broadcast (v_mov_b64_dpp, gfx90a+ DP-ALU DPP supports exactly row_newbcast): two coefficients travel
// in one instruction and arrive as an aligned register pair, whose halves v_pk_fma_f32 selects with op_sel -- no
// copies to build packed operands
typedef float f2v __attribute__((ext_vector_type(2)));
template <int SRC>
__device__ inline f2v row_bcast_f2(f2v v) {
#ifdef GVL_BCAST_2X32   // timing build: two 32-bit broadcasts instead of one 64-bit
  return (f2v){row_bcast_f<SRC>(v.x), row_bcast_f<SRC>(v.y)};
#else
  const long long r = __builtin_amdgcn_update_dpp((long long)0, __builtin_bit_cast(long long, v), 0x150 + SRC, 0xF, 0xF, true);
  return __builtin_bit_cast(f2v, r);
#endif
}
// lane SRC's v broadcast over its row and added to this lane's `addend`, in ONE VALU instruction (v_add_u32_dpp).  The
// compiler's DPP combiner does not fold the v_mov_dpp + v_add pair of the forward sample step (the broadcast is
// hoisted over the exec-mask change of the guarded store), hence the asm.  v must not be written by the two preceding
// VALU instructions (DPP read-after-write hazard; the assembler text is opaque to the hazard recogniser): `settle`
// prepends the two wait states for the first use after v was produced.
template <int SRC>
__device__ inline int row_bcast_add(int v, int addend, bool settle = false) {
  int r;
  if (settle)
    asm("s_nop 1\n\tv_add_u32_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf bound_ctrl:1"
        : "=v"(r) : "v"(v), "v"(addend), "n"(SRC));
  else
    asm("v_add_u32_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf bound_ctrl:1"
        : "=v"(r) : "v"(v), "v"(addend), "n"(SRC));
  return r;
}
typedef __attribute__((address_space(3))) const char lds_cbyte;
typedef float f4v __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) const f4v lds_cf4v;
__device__ inline float4 lds_ld4(lds_cbyte *p) {                                 // ds_read_b128 at an LDS byte address
  const f4v t = *reinterpret_cast<lds_cf4v *>(p);
  return make_float4(t.x, t.y, t.z, t.w);
}
// acc (4 channels as two pairs) += c.x * v0 + c.y * v1
__device__ inline void fma4x2(f2v c, const float4 &v0, const float4 &v1, f2v &a01, f2v &a23) {
  const f2v lo = __builtin_shufflevector(c, c, 0, 0), hi = __builtin_shufflevector(c, c, 1, 1);
  a01 = __builtin_elementwise_fma(lo, (f2v){v0.x, v0.y}, a01);
  a23 = __builtin_elementwise_fma(lo, (f2v){v0.z, v0.w}, a23);
  a01 = __builtin_elementwise_fma(hi, (f2v){v1.x, v1.y}, a01);
  a23 = __builtin_elementwise_fma(hi, (f2v){v1.z, v1.w}, a23);
}
// TIMING BUILDS ONLY (-DGVL_FWD_FMAC_DPP; measured in round 5 and not shipped): acc (4 channels) += c.x * v0 + c.y * v1 with
// the coefficient taken from lane SRC of the DPP row INSIDE the multiply-add (v_fmac_f32_dpp: the VOP2 form carries a DPP
// control on its first operand) -- 8 instructions per sample step instead of a 64-bit broadcast + 4 v_pk_fma_f32, the same
// products in the same order.  The forward's sample loop went from 3.33 to 3.94 us in situ (tools/fwd_ab.sh): a DPP operand
// costs the FMA more than the separate broadcast does.
template <int SRC>
__device__ inline void fma4x2_bcast(f2v c, const float4 &v0, const float4 &v1, f2v &a01, f2v &a23) {
  float ax = a01.x, ay = a01.y, az = a23.x, aw = a23.y;
  const float cl = c.x, ch = c.y;
#define GVL_FMAC_DPP(ACC, COEF, VAL) \
  asm("v_fmac_f32_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(ACC) : "v"(COEF), "v"(VAL), "n"(SRC));
  GVL_FMAC_DPP(ax, cl, v0.x) GVL_FMAC_DPP(ay, cl, v0.y) GVL_FMAC_DPP(az, cl, v0.z) GVL_FMAC_DPP(aw, cl, v0.w)
  GVL_FMAC_DPP(ax, ch, v1.x) GVL_FMAC_DPP(ay, ch, v1.y) GVL_FMAC_DPP(az, ch, v1.z) GVL_FMAC_DPP(aw, ch, v1.w)
#undef GVL_FMAC_DPP
  a01 = (f2v){ax, ay};
  a23 = (f2v){az, aw};
}
// all-reduce (sum) inside every 16-lane row
__device__ inline float row_allsum(float v) {
  v += dpp_f<0xB1>(v);    // quad_perm [1,0,3,2]
  v += dpp_f<0x4E>(v);    // quad_perm [2,3,0,1]
  v += dpp_f<0x124>(v);   // row_ror:4
  v += dpp_f<0x128>(v);   // row_ror:8
  return v;
}

// Reduce-scatter of v[16] over the 16 lanes of every DPP row: afterwards lane j holds sum over the row's lanes of v[j].
// Four halving stages whose partners differ in exactly the bit that selects the kept half (row_mirror: j <-> 15-j,
// row_half_mirror: j <-> 7-j inside 8, quad_perm [3,2,1,0] and [1,0,3,2]) -- 15 DPP adds + 30 selects for 16 sums,
// where 16 separate row all-reduces cost 64 DPP adds (and leave every sum on every lane, which nobody needs).
__device__ inline float row_reduce_scatter16(const float (&v)[16], int j) {
  float a8[8], a4[4], a2[2];
  const bool b1 = j & 2, b0 = j & 1;
#ifdef GVL_RS16_SELECTS    // timing build: the select form of rounds 2-5 for all four levels
  const bool b3 = j & 8, b2 = j & 4;
#pragma unroll
  for (int i = 0; i < 8; ++i) a8[i] = (b3 ? v[8 + i] : v[i]) + dpp_f<0x140>(b3 ? v[i] : v[8 + i]);
#pragma unroll
  for (int i = 0; i < 4; ++i) a4[i] = (b2 ? a8[4 + i] : a8[i]) + dpp_f<0x141>(b2 ? a8[i] : a8[4 + i]);
#else
  // Levels 1 and 2 without selects: which half of the row keeps element i and which keeps element 8 + i (4 + i) is a matter of
  // WHOLE DPP banks (lanes 0-7 | 8-15, then banks {0, 2} | {1, 3}), so each output is two v_add_f32_dpp with complementary
  // bank masks -- the partner's copy of the same element added to the own one -- instead of two selects + move + add.
  // The same sums of the same operands.  (`s_nop 1`: a VALU write needs two wait states before a DPP read of the register; the
  // assembler text is opaque to the compiler's hazard recogniser, on the way in and on the way out.)
#define GVL_RS_PAIR(D, LO, HI, CTRL, M0, M1)                                                            \
  "v_add_f32_dpp " D ", " LO ", " LO " " CTRL " row_mask:0xf bank_mask:" M0 "\n\t"                       \
  "v_add_f32_dpp " D ", " HI ", " HI " " CTRL " row_mask:0xf bank_mask:" M1 "\n\t"
  asm("s_nop 1\n\t"
      GVL_RS_PAIR("%0", "%8", "%16", "row_mirror", "0x3", "0xc") GVL_RS_PAIR("%1", "%9", "%17", "row_mirror", "0x3", "0xc")
      GVL_RS_PAIR("%2", "%10", "%18", "row_mirror", "0x3", "0xc") GVL_RS_PAIR("%3", "%11", "%19", "row_mirror", "0x3", "0xc")
      GVL_RS_PAIR("%4", "%12", "%20", "row_mirror", "0x3", "0xc") GVL_RS_PAIR("%5", "%13", "%21", "row_mirror", "0x3", "0xc")
      GVL_RS_PAIR("%6", "%14", "%22", "row_mirror", "0x3", "0xc") GVL_RS_PAIR("%7", "%15", "%23", "row_mirror", "0x3", "0xc")
      : "=&v"(a8[0]), "=&v"(a8[1]), "=&v"(a8[2]), "=&v"(a8[3]), "=&v"(a8[4]), "=&v"(a8[5]), "=&v"(a8[6]), "=&v"(a8[7])
      : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(v[4]), "v"(v[5]), "v"(v[6]), "v"(v[7]),
        "v"(v[8]), "v"(v[9]), "v"(v[10]), "v"(v[11]), "v"(v[12]), "v"(v[13]), "v"(v[14]), "v"(v[15]));
  asm("s_nop 1\n\t"
      GVL_RS_PAIR("%0", "%4", "%8", "row_half_mirror", "0x5", "0xa") GVL_RS_PAIR("%1", "%5", "%9", "row_half_mirror", "0x5", "0xa")
      GVL_RS_PAIR("%2", "%6", "%10", "row_half_mirror", "0x5", "0xa") GVL_RS_PAIR("%3", "%7", "%11", "row_half_mirror", "0x5", "0xa")
      "s_nop 1"
      : "=&v"(a4[0]), "=&v"(a4[1]), "=&v"(a4[2]), "=&v"(a4[3])
      : "v"(a8[0]), "v"(a8[1]), "v"(a8[2]), "v"(a8[3]), "v"(a8[4]), "v"(a8[5]), "v"(a8[6]), "v"(a8[7]));
#undef GVL_RS_PAIR
#endif
#pragma unroll
  for (int i = 0; i < 2; ++i) a2[i] = (b1 ? a4[2 + i] : a4[i]) + dpp_f<0x1B>(b1 ? a4[i] : a4[2 + i]);
  return (b0 ? a2[1] : a2[0]) + dpp_f<0xB1>(b0 ? a2[0] : a2[1]);
}

// Interpolation coefficients of ONE temporal sample against the LDS slab, expressed on the row pair (r, r+1):
//   sample = c_lo * V[r] + c_hi * V[r+1]           (c's already include the vertical weight wy, not attn)
//   d sample / d loc_x = dx_lo * V[r] + dx_hi * V[r+1]
//   d sample / d loc_y = dy * (horizontal interpolation)   (H = 1: cuh:124-159 collapses to +-1)
// r is clamped so that both rows are inside the slab allocation (S+1 rows) for ANY input, incl. NaN.
struct Coef1D {
  int r;        // level-local row
  float c_lo, c_hi, dx_lo, dx_hi, wy, dy;
};

template <int PAD>
__device__ inline Coef1D coef_1d(float lx, float ly, int T) {
  Coef1D c;
  c.r = 0; c.c_lo = c.c_hi = c.dx_lo = c.dx_hi = c.wy = c.dy = 0.f;
  float dmx, dmy;
  const float x = pixel_coord<float>(lx, T, PAD, dmx);
  const float y = pixel_coord<float>(ly, 1, PAD, dmy);
  bool valid = true;
  if (PAD == kPadZeros) valid = (y > -1.f && x > -1.f && y < 1.f && x < (float)T);   // cuh:289 with H = 1
  if (!valid) return c;
  // vertical: H = 1 -> only row 0 exists.  y in (-1,1): hl = -1 -> weight lh on row 0, else weight 1-lh.
  const float yf = floorf(y);
  const float lh = y - yf;
  const bool low_is_row0 = (yf == 0.f);
  c.wy = low_is_row0 ? (1.f - lh) : lh;
  c.dy = dmy * (low_is_row0 ? -1.f : 1.f);
  // horizontal
  const float xf = floorf(x);
  const int x0 = (int)xf;
  const float a = x - xf;
  // Taps x0 (weight 1 - a, d/dx = -1) and x0 + 1 (weight a, d/dx = +1), each counted only inside [0, T - 1] (cuh:57-67,
  // 125,134), laid on the row pair (r, r + 1) with r = x0 clamped to [0, max(T - 2, 0)].  x0 is in [-1, T - 1] here, so
  // there are three cases: x0 = -1 (`below`: only tap x0 + 1 = row r counts), x0 = T - 1 > r (`above`: only tap x0 = row r + 1),
  // else tap x0 = row r and tap x0 + 1 = row r + 1 (which does not exist when T = 1).  Two compares + selects: the sample loops
  // are VALU-issue bound and the tap-by-tap form (four range tests, four equality tests, eight selects) was 16 instructions
  // more per pass; same numbers (each coefficient was one term + 0).
  const int rmax = T >= 2 ? T - 2 : 0;
  const bool below = x0 < 0, above = x0 > rmax, two = T >= 2;
  c.r = below ? 0 : (above ? rmax : x0);
  const float b = 1.f - a;
  c.c_lo = below ? a : (above ? 0.f : b);
  c.c_hi = below ? 0.f : (above ? b : (two ? a : 0.f));
  c.dx_lo = below ? dmx : (above ? 0.f : -dmx);
  c.dx_hi = below ? 0.f : (above ? -dmx : (two ? dmx : 0.f));
  return c;
}

__device__ inline float4 fma4(float a, float4 v, float4 acc) {
  acc.x = fmaf(a, v.x, acc.x); acc.y = fmaf(a, v.y, acc.y);
  acc.z = fmaf(a, v.z, acc.z); acc.w = fmaf(a, v.w, acc.w);
  return acc;
}
__device__ inline float dot4(float4 a, float4 b) {
  return fmaf(a.x, b.x, fmaf(a.y, b.y, fmaf(a.z, b.z, a.w * b.w)));
}

// stage rows [row0, S) of the (b,m) value slab [S][64] into LDS as float4[(S-row0)*16]; one zero row follows.
// row0 > 0 ("L0G"): level 0 does not fit beside the other levels in the 160 KB LDS (long videos: T = 512 gives
// S*256 B = 240 KB); its rows are then read straight from global memory / L2 by the sample steps of level 0.
template <typename VT>
__device__ inline void stage_slab(float4 *slab4, const VT *value, int b, int m, int S, int M, int row0 = 0) {
  const int64_t src = ((int64_t)b * S * M + m) * 16;
  const int n = (S - row0) * 16;
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    const int s = row0 + (i >> 4), j = i & 15;
    slab4[i] = ld4(value, src + (int64_t)s * M * 16 + j);
  }
  if (threadIdx.x < 16) slab4[n + threadIdx.x] = make_float4(0.f, 0.f, 0.f, 0.f);
}

// ------------------------------------------------------------------------------------------------------
// Sampling operands of sample j of one (b,q,m).  Two sources:
//   FUSED = false   the op's own inputs: loc (B,Q,M,L,P,2) and attn (B,Q,M,L,P)           (ms_deform_attn_func.py:25)
//   FUSED = true    what MSDeformAttn.forward derives them from (ms_deform_attn.py:99-117), so that neither
//                   tensor ever exists in HBM: proj (B*Q, 2*M*L*P) = one GEMM against [sampling_offsets;
//                   attention_weights] -> columns [0, M*LP) are the raw offsets, [M*LP, 2*M*LP) the attention
//                   logits; ref (B,Q,L,RD) the reference points.  In-kernel: softmax over the DPP row (:100-101),
//                   loc = ref + off / T_l  (RD = 1, :103-106)  or  ref_c + off / P * ref_len * 0.5  (RD = 2, :107-109),
//                   y = 0.5 (:115).  Needs L*P == 16 (one sample per lane of the row).
// ------------------------------------------------------------------------------------------------------
struct RawOps {
  float a, b, c, d;   // !FUSED: (x, y, w, -)     FUSED: (offset, logit, ref0, ref1)
};
// The backward's query loops request the operands of pass k + 1 at the top of pass k and store pass k's results at its
// bottom.  gfx9 has ONE counter for vector loads and stores, and they retire out of order with respect to each other: the wait
// for the prefetched operands at the top of pass k + 1 therefore became vmcnt(0) -- a wait for the STORES issued a few
// instructions earlier, i.e. a write round trip per pass on the critical path.  This makes the prefetched registers opaque
// values in front of the stores: the compiler's wait lands here, where the loads (issued a whole pass ago) have long
// returned, and the stores drain behind the next pass's arithmetic.
#define GVL_TOUCH_PREFETCH(R, G) \
  asm volatile("" : "+v"((R).a), "+v"((R).b), "+v"((R).c), "+v"((R).d), "+v"((G).x), "+v"((G).y), "+v"((G).z), "+v"((G).w));

// p0 = loc (fp32) | proj (storage type VT);  p1 = attn | ref (always fp32: positions must not be rounded to bf16)
template <bool FUSED, typename VT>
__device__ inline RawOps fetch_ops(const void *__restrict__ p0, const float *__restrict__ p1, int64_t bq, int m, int M,
                                   int LP, int L, int RD, int j, int l) {
  RawOps r;
  if (!FUSED) {
    const int64_t i = (bq * M + m) * LP + j;
    const float2 xy = reinterpret_cast<const float2 *>(p0)[i];
    r.a = xy.x; r.b = xy.y; r.c = p1[i]; r.d = 0.f;
  } else {
    const VT *row = reinterpret_cast<const VT *>(p0) + bq * (int64_t)(2 * M * LP);
    r.a = (float)row[m * LP + j];
    r.b = (float)row[M * LP + m * LP + j];
    const float *rp = p1 + (bq * L + l) * RD;
    r.c = rp[0];
    r.d = rp[RD == 2 ? 1 : 0];                                         // (branch-free, see fetch_at; RD == 1: not used)
  }
  return r;
}

// the same loads from per-lane ELEMENT OFFSETS that the caller advances by a constant per pass (no index arithmetic
// in the loop): o0 into p0 (FUSED: the lane's offset column, its logit is MLP further; else the lane's (x, y) pair and
// weight), o1 into p1 (FUSED: the reference point of the lane's level)
template <bool FUSED, typename VT>
__device__ inline RawOps fetch_at(const void *__restrict__ p0, const float *__restrict__ p1, int64_t o0, int64_t o1,
                                  int MLP, int RD) {
  RawOps r;
  if (!FUSED) {
    const float2 xy = reinterpret_cast<const float2 *>(p0)[o0];
    r.a = xy.x; r.b = xy.y; r.c = p1[o0]; r.d = 0.f;
  } else {
    const VT *row = reinterpret_cast<const VT *>(p0) + o0;
    r.a = (float)row[0];
    r.b = (float)row[MLP];
    // (branch-free: behind a conditional load the compiler waits for EVERY outstanding load -- vmcnt(0) -- which put the
    // operand fetches of consecutive passes and the slab staging in series)
    r.c = p1[o1];
    r.d = p1[o1 + (RD == 2 ? 1 : 0)];                                 // (RD == 1: not used by resolve_ops)
  }
  return r;
}

// (v_max_f32 with the DPP control on its own operand: written as fmaxf(v, dpp(v)) the compiler emits v_mov_dpp + a
// canonicalising v_max of the moved value + the v_max -- 12 VALU instructions per reduction instead of 4 in loops that are
// VALU-issue bound.  `s_nop 1`: the two wait states between a VALU write and a DPP read of the same register.)
// fetch_ops for a query given as (bq_u: a row index that is the same for the whole wavefront, dq: this lane's offset from it):
// the 64-bit part of every address is then wavefront-uniform -- computed on the scalar unit -- and the lane part a 32-bit byte
// offset, which is an addressing mode of the load (scalar base + vector offset).  Per-lane 64-bit index arithmetic (quarter-rate
// multiplies) was a fifth of the issue time of the backward's query passes.
template <bool FUSED, typename VT>
__device__ inline RawOps fetch_ops_u(const void *__restrict__ p0, const float *__restrict__ p1, int64_t bq_u, int dq, int m, int M,
                                     int LP, int L, int RD, int j, int l) {
  RawOps r;
  if (!FUSED) {
    const unsigned e = (unsigned)((dq * M + m) * LP + j);
    const float2 xy = *reinterpret_cast<const float2 *>(
        reinterpret_cast<const char *>(reinterpret_cast<const float2 *>(p0) + bq_u * (M * LP)) + e * 8u);
    r.a = xy.x; r.b = xy.y;
    r.c = *reinterpret_cast<const float *>(reinterpret_cast<const char *>(p1 + bq_u * (M * LP)) + e * 4u);
    r.d = 0.f;
  } else {
    const char *u0 = reinterpret_cast<const char *>(reinterpret_cast<const VT *>(p0) + bq_u * (2 * M * LP));
    const unsigned e0 = (unsigned)(dq * (2 * M * LP) + m * LP + j) * (unsigned)sizeof(VT);
    r.a = (float)*reinterpret_cast<const VT *>(u0 + e0);
    r.b = (float)*reinterpret_cast<const VT *>(u0 + (e0 + (unsigned)(M * LP) * (unsigned)sizeof(VT)));
    const char *u1 = reinterpret_cast<const char *>(p1 + bq_u * (L * RD));
    const unsigned e1 = (unsigned)((dq * L + l) * RD) * 4u;
    r.c = *reinterpret_cast<const float *>(u1 + e1);
    r.d = *reinterpret_cast<const float *>(u1 + (e1 + (RD == 2 ? 4u : 0u)));
  }
  return r;
}
// four channels at quad index i4 (a 32-bit lane offset) behind a wavefront-uniform base
__device__ inline float4 ld4_u(const float *base_u, unsigned i4) {
  return *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(base_u) + i4 * 16u);
}
__device__ inline float4 ld4_u(const bf16_t *base_u, unsigned i4) {
  const bf16x4 v = *reinterpret_cast<const bf16x4 *>(reinterpret_cast<const char *>(base_u) + i4 * 8u);
  return make_float4((float)v.a, (float)v.b, (float)v.c, (float)v.d);
}

__device__ inline float row_allmax(float v) {
  asm("s_nop 1\n\tv_max_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_ror:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf bound_ctrl:1"
      : "+v"(v));
  return v;
}

// -> (x, y, attention weight) of the lane's sample; FUSED also returns d loc_x / d offset in `dloc`
// Divisions: the sample loop is VALU-issue bound and an IEEE fp32 division is ~10 instructions, so the level length
// and the point count enter as reciprocals computed ONCE per lane (invT = 1/T_l, invP = 1/P, correctly rounded) and
// the softmax normaliser through v_rcp_f32 (1 ulp): locations / weights within 1-2 ulp of the divided form.
template <bool FUSED>
__device__ inline void resolve_ops(const RawOps &r, float invT, float invP, int RD, float &x, float &y, float &w,
                                   float &dloc) {
  if (!FUSED) {
    x = r.a; y = r.b; w = r.c; dloc = 0.f;
  } else {
    const float mx = row_allmax(r.b);
    const float e = __expf(r.b - mx);
    w = e * __builtin_amdgcn_rcpf(row_allsum(e));
    if (RD == 1) { x = fmaf(r.a, invT, r.c); dloc = invT; }
    else { dloc = r.d * 0.5f * invP; x = fmaf(r.a, dloc, r.c); }
    y = 0.5f;
  }
}

// (b,m) slab of a workgroup.  Workgroups go to the 8 XCDs round-robin (id % 8), each XCD with its own L2.  The per-head
// pieces of the fused operands -- proj (B*Q, 2*M*16): 64 B of offsets and 64 B of logits per (query, head) -- are HALF a
// 128-byte line, the other half belonging to the next head: with heads on different XCDs every line is fetched by two L2s
// and half of each fetch is unused (PMC: proj traffic = 2x its bytes).  The map below puts the heads (2k, 2k + 1) of a video
// on the same XCD; both workgroups of a slab (ids B*M apart) stay together as before.  Identity when B*M % 16 != 0.
__device__ inline int slab_of_block(int idx, int BM) {
  if (BM & 15) return idx;
  const int xcd = idx & 7, slot = idx >> 3;
  return 2 * ((slot >> 1) * 8 + xcd) + (slot & 1);
}

// ------------------------------------------------------------------------------------------------------
// t1d_d64 forward.  grid = (B*M, nchunk) workgroups; workgroups of one (b,m) are B*M apart in dispatch order => same XCD L2.
// ------------------------------------------------------------------------------------------------------
// AMAX: additionally leave max |out[b, q, :]| per output ROW in amax_out (B*Q floats, zero-initialised by the caller) --
// the row maximum the split-fp16 output projection behind it (gvl_layers.hip) derives its operand scale from.  A row's
// 512 channels come from the 8 heads' workgroups: one DPP row reduction + one atomic max per (row, head).
template <int PAD, bool FULL16, bool FUSED, bool L0G, typename VT, bool AMAX = false>
__global__ void __launch_bounds__(1024) k_fwd_t1d_d64(const VT *__restrict__ value,
                                                     const int64_t *__restrict__ shapes,
                                                     const int64_t *__restrict__ lsi, const void *__restrict__ loc,
                                                     const float *__restrict__ attn, int B, int S, int M, int L, int Q,
                                                     int P, int RD, int nchunk, VT *__restrict__ out,
                                                     unsigned long long *__restrict__ stamps,
                                                     unsigned *__restrict__ amax_out, int qper, int m_shift, float invP,
                                                     int Qp) {
  // Qp (FUSED): rows of `loc` (= proj) per video -- Q, or 0 when every video reads the SAME Q rows (the first decoder layer under the
  // 'queries' input in inference: its offsets / logits depend on parameters only, gvl_amd/layers.py _first_layer_constants)
  extern __shared__ float4 slab4[];
  // grid = (B*M, nchunk): workgroups are dispatched x first, so the linear id (what the XCD round-robin sees) is
  // chunk * B*M + slab as before -- but neither a modulo nor a division by B*M is computed here, the chunk length comes from
  // the host and b, m come from a shift when M is a power of two (m_shift >= 0): four runtime integer divisions were ~80
  // instructions in front of the kernel's first memory request.
  const int BM = B * M;
  const int wg_id = (int)(blockIdx.y * gridDim.x + blockIdx.x);
  // diagnostics (gvl_msda_debug_stamps): 100 MHz wall-clock stamps per workgroup {start, slab staged, loop done}
  if (stamps && threadIdx.x == 0) stamps[wg_id * 4 + 0] = wall_clock64();
  // (nchunk < 0: diagnostic A/B switch GVL_MSDA_XCD_PAIRS=0 -- the plain id -> slab map)
  const int bm = nchunk < 0 ? (int)blockIdx.x : slab_of_block((int)blockIdx.x, BM), chunk = (int)blockIdx.y;
  int b, m;
  if (m_shift >= 0) { b = bm >> m_shift; m = bm & (M - 1); }
  else { b = bm / M; m = bm % M; }
  const int lane = threadIdx.x & 63, j = lane & 15, tq = lane >> 4;
  // (the wavefront index as a SCALAR: everything derived from it -- the wavefront's first query, the base addresses of its
  // operand rows and output rows -- is then computed on the scalar unit and the loads / stores take it as their scalar base)
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), nw = blockDim.x >> 6;
  // FULL16: L*P == 16 is a compile-time fact: no branches between the 16 sample steps, none around the operand fetches
  const int LP = FULL16 ? 16 : L * P;
  const int q0 = chunk * qper;
  const int q1 = min(Q, q0 + qper);
  // The kernel is a latency chain (launch -> slab -> LDS reads), so every global load that does not depend on LDS
  // is issued before the slab staging, and each pass prefetches the sampling operands of the next one.
  int Tl = 1, st = 0;
  // (FULL16: L * P = 16, so j / P = j * L / 16 -- a runtime division is ~20 vector instructions, and everything in front of
  // the barrier is on the critical path of a kernel whose set-up is bound by VALU issue)
  const int lvl = FULL16 ? (j * L) >> 4 : (j < LP ? j / P : 0);
  if (j < LP) {
    // the LOW dwords of the int64 entries: loaded as 64-bit values their unused high halves are registers the compiler
    // re-uses at once -- and it then waits for the load (vmcnt(0)) in front of every request that should have followed it
    Tl = *reinterpret_cast<const int *>(reinterpret_cast<const char *>(shapes) + (unsigned)(16 * lvl + 8));
    st = *reinterpret_cast<const int *>(reinterpret_cast<const char *>(lsi) + (unsigned)(8 * lvl));
  }
  int qb = q0 + wave * 4;
  // operand cursor: query index of the next fetch and the lane's element offsets for it, advanced by a constant per pass
  int qf = qb + tq;
  const int MLP = M * LP;
  int64_t o0 = FUSED ? ((int64_t)b * Qp + qf) * (2 * MLP) + m * LP + j : (((int64_t)b * Q + qf) * M + m) * LP + j;
  int64_t o1 = FUSED ? (((int64_t)b * Q + qf) * L + lvl) * RD : 0;
  const int64_t step0 = (int64_t)nw * 4 * (FUSED ? 2 * MLP : MLP), step1 = FUSED ? (int64_t)nw * 4 * L * RD : 0;
  // FULL16: no branch around the loads -- lanes past the end of the list re-read the list's last query (their results are
  // never stored).  Behind a guarded load the compiler cannot count the requests in flight and waits for ALL of them
  // (vmcnt(0)) before the first use of any operand, the slab transfer included.  Addresses = a 64-bit base that is the same
  // for the whole wavefront (scalar unit) + a 32-bit lane offset: the set-up is bound by VALU issue (four wavefronts per SIMD
  // run it), and per-lane 64-bit index arithmetic was a third of its vector instructions.
  int qf_u = qb;                                                       // (uniform) first query of the wavefront's next fetch
  auto fetch_next = [&]() {
    RawOps r = {0.f, 0.5f, 0.f, 0.f};
    if constexpr (FULL16) {
      const int qc = min(qf_u, q1 - 1);                                // (uniform) clamped into the list
      const int dq = min(tq, q1 - 1 - qc);                             // this lane's query relative to qc, clamped likewise
      const int64_t bq = (int64_t)b * Q + qc;
      // (byte offsets in 32-bit arithmetic: scalar base + zero-extended 32-bit offset is an addressing mode of the load)
      if constexpr (FUSED) {
        const char *u0 = reinterpret_cast<const char *>(reinterpret_cast<const VT *>(loc) + ((int64_t)b * Qp + qc) * (2 * MLP) + m * LP);
        const unsigned e0 = (unsigned)(dq * (2 * MLP) + j) * (unsigned)sizeof(VT);
        r.a = (float)*reinterpret_cast<const VT *>(u0 + e0);
        r.b = (float)*reinterpret_cast<const VT *>(u0 + (e0 + (unsigned)MLP * (unsigned)sizeof(VT)));
        const char *u1 = reinterpret_cast<const char *>(attn + bq * (L * RD));
        const unsigned e1 = (unsigned)((dq * L + lvl) * RD) * 4u;
        r.c = *reinterpret_cast<const float *>(u1 + e1);
        r.d = *reinterpret_cast<const float *>(u1 + (e1 + (RD == 2 ? 4u : 0u)));   // (RD == 1: not used by resolve_ops)
      } else {
        const unsigned e0 = (unsigned)(dq * MLP + j);
        const float2 xy = *reinterpret_cast<const float2 *>(
            reinterpret_cast<const char *>(reinterpret_cast<const float2 *>(loc) + (bq * M + m) * LP) + e0 * 8u);
        r.a = xy.x; r.b = xy.y;
        r.c = *reinterpret_cast<const float *>(reinterpret_cast<const char *>(attn + (bq * M + m) * LP) + e0 * 4u);
        r.d = 0.f;
      }
    } else {
      if (qf < q1 && j < LP) r = fetch_at<FUSED, VT>(loc, attn, o0, o1, MLP, RD);
    }
    qf += nw * 4;
    qf_u += nw * 4;
    o0 += step0;
    o1 += step1;
    return r;
  };
  RawOps r_n = fetch_next();
  const unsigned e_out = (unsigned)(tq * M * 16 + j);                  // this lane's output quad relative to the wavefront's query
  // L0G: level 0 (rows [0, T_0)) stays in global memory, LDS holds rows [T_0, S); needs FULL16 and P == 4 so that
  // "sample step SI belongs to level 0" is the compile-time test SI < 4
  const int row0 = L0G ? __builtin_amdgcn_readfirstlane((int)shapes[1]) : 0;
  const int64_t vg = ((int64_t)b * S * M + m) * 16 + j;               // this lane's channels of row 0 of the slab
  const float invT = 1.f / (float)Tl;                                 // (invP = 1.f / (float)P: the same division, done by the host)
  // sampling operands -> (slab row as LDS byte offset | global row index for L0G level 0, coefficient pair)
  auto prep = [&](const RawOps &r_in, int &roff, f2v &cc) {
    // (the operands as opaque values from here on: otherwise the first instructions of resolve_ops are hoisted into the
    // guarded fetch that requested them -- where they wait for the loads, in front of every later request)
    RawOps r = r_in;
    asm("" : "+v"(r.a), "+v"(r.b), "+v"(r.c), "+v"(r.d));
    float2 xy;
    float w, dloc_;
    resolve_ops<FUSED>(r, invT, invP, RD, xy.x, xy.y, w, dloc_);
    roff = 0;
    cc = (f2v){0.f, 0.f};
    if (j < LP) {
      const Coef1D c = coef_1d<PAD>(xy.x, xy.y, Tl);
      roff = (L0G && lvl == 0) ? c.r : (st - row0 + c.r) * 256;
      const float ww = w * c.wy;
      cc = (f2v){c.c_lo * ww, c.c_hi * ww};
    }
  };
  // Slab staging, software-pipelined against the coefficient arithmetic of the first kAhead passes.  The sample loop is
  // bound by VALU issue (round 5: 168 vector instructions per wavefront pass, ~100 of them prep()), while between the kernel's
  // start and the barrier behind the staging the vector ALUs have nothing to do for 2.3-2.8 us (cfg A: one cold round trip
  // for 48 KB per CU).  So the operands of the first kAhead passes are requested up front, the slab's float4 are requested, and
  // prep() of all kAhead passes runs while they travel; those passes then consist of the 16 sample steps alone.  cfg A has
  // 2.3 passes per wavefront (150 queries per workgroup, 64 per pass): every pass's coefficients are ready at the barrier.
  // Later passes (longer query lists) run as before: operands two passes ahead, coefficients one pass ahead, inside the loop.
  constexpr int kPre = 4, kAhead = 3;
  const int64_t src0 = ((int64_t)b * S * M + m) * 16;
  const int nstage = (S - row0) * 16;
  const int pstep = nw * 4;
  RawOps r_a[kAhead];
  r_a[0] = r_n;
#pragma unroll
  for (int k = 1; k < kAhead; ++k) r_a[k] = fetch_next();
  r_n = fetch_next();                                                  // operands of pass kAhead (requested BEFORE the slab:
                                                                       // requests retire in order, the operands must not queue behind it)
  // fp32 slabs go to LDS by LDS-DMA (global_load_lds_dwordx4: lane l of a wavefront writes 16 bytes at M0 + 16 l -- exactly
  // slab4[i] for i = thread + k * blockDim): no data registers, so nothing the compiler could copy or wait for between the
  // request and the barrier (staged through registers it placed a vmcnt(0) in the middle of the coefficient arithmetic that
  // should run under the transfer).  Inline assembly, because the compiler cannot count the requests of a loop whose trip
  // count it does not know and then waits for ALL outstanding loads before the first use of an operand requested earlier;
  // unseen requests only make its counted waits stricter (they retire in order, the operands first).  The wait for the
  // transfers themselves is the explicit vmcnt(0) in front of the barrier.  bf16 slabs are widened on the way and keep the
  // register path.
  constexpr bool kDma = std::is_same<VT, float>::value;
  float4 pre[kPre];
  if constexpr (kDma) {
    const char *vsrc = reinterpret_cast<const char *>(value) + (src0 + (int64_t)row0 * M * 16) * 16;     // (uniform)
    for (int i = threadIdx.x; i < nstage; i += blockDim.x) {
      const unsigned voff = (unsigned)(((i >> 4) * M * 16 + (i & 15)) * 16);
      const int dst = __builtin_amdgcn_readfirstlane((int)(uintptr_t)(lds_cbyte *)reinterpret_cast<const char *>(slab4 + (i - lane)));
      asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, %2" : : "s"(dst), "v"(voff), "s"(vsrc) : "memory");
    }
  } else {
#pragma unroll
    for (int k = 0; k < kPre; ++k) {
      const int i = min((int)(threadIdx.x + k * blockDim.x), nstage - 1);
      pre[k] = ld4(value, src0 + (int64_t)(row0 + (i >> 4)) * M * 16 + (i & 15));
    }
  }
  int roff_a[kAhead];
  f2v cc_a[kAhead];
#pragma unroll
  for (int k = 0; k < kAhead; ++k) {
    roff_a[k] = 0;
    cc_a[k] = (f2v){0.f, 0.f};
    if (qb + k * pstep < q1) prep(r_a[k], roff_a[k], cc_a[k]);         // (wave-uniform)
  }
  int roff_c = 0;
  f2v cc_c = {0.f, 0.f};
  if (qb + kAhead * pstep < q1) {                                      // a longer list: the generic loop's pipeline, primed
    prep(r_n, roff_c, cc_c);
    r_n = fetch_next();
  }
  if constexpr (!kDma) {
#pragma unroll
    for (int k = 0; k < kPre; ++k) {
      const int i = threadIdx.x + k * blockDim.x;
      if (i < nstage) slab4[i] = pre[k];
    }
    for (int i = threadIdx.x + kPre * blockDim.x; i < nstage; i += blockDim.x)
      slab4[i] = ld4(value, src0 + (int64_t)(row0 + (i >> 4)) * M * 16 + (i & 15));
  }
  if (threadIdx.x < 16) slab4[nstage + threadIdx.x] = make_float4(0.f, 0.f, 0.f, 0.f);
  if constexpr (kDma) asm volatile("s_waitcnt vmcnt(0)" : : : "memory");
  __syncthreads();
  if (stamps && threadIdx.x == 0) stamps[wg_id * 4 + 1] = wall_clock64();
  const char *slab_b = reinterpret_cast<const char *>(slab4);
  const int lane_off = j * 16;
  const int lane_lds = (int)(uintptr_t)(lds_cbyte *)slab_b + lane_off;          // this lane's 16-byte column, LDS address

  // one pass = the 16 sample steps of the wavefront's four queries + the store.  Per sample step: v_add_u32_dpp (row
  // broadcast + lane offset = LDS address), v_mov_b64_dpp (both coefficients), 2 ds_read_b128, 4 v_pk_fma_f32.
  // (Round 5, with prep() still inside the loop: requesting the reads of four steps together, two groups in flight, did not
  // help -- 4.52 -> 4.75 us in situ at 118 registers; 35 fewer VALU instructions per pass bought ~0.3 us; moving prep() under
  // the slab transfer 1.0 us.)
#ifdef GVL_FWD_ABL_NO_LDS      // timing build (tools/fwd_ablate.sh): the sample steps without their LDS reads
#define GVL_FWD_ROWS(ROW, V0, V1) { const float t_ = __builtin_bit_cast(float, (int)(uintptr_t)(ROW)); V0 = make_float4(t_, cc.x, t_, cc.y); V1 = V0; }
#else
#define GVL_FWD_ROWS(ROW, V0, V1) { V0 = lds_ld4(ROW); V1 = lds_ld4((ROW) + 256); }
#endif
  auto run_pass = [&](const int roff, const f2v cc) {
    const int q = qb + tq;
    const bool act = q < q1;
    f2v a01 = {0.f, 0.f}, a23 = {0.f, 0.f};
#ifdef GVL_FWD_FMAC_DPP         // timing build, see fma4x2_bcast
#define GVL_FWD_ACC(SI, V0, V1) fma4x2_bcast<SI>(cc, V0, V1, a01, a23);
#else
#define GVL_FWD_ACC(SI, V0, V1) fma4x2(row_bcast_f2<SI>(cc), V0, V1, a01, a23);
#endif
#define GVL_FWD_STEP(SI)                                                        \
  if (FULL16 || SI < LP) {                                                      \
    float4 v0, v1;                                                              \
    lds_cbyte *row = (lds_cbyte *)(uintptr_t)(unsigned)row_bcast_add<SI>(roff, lane_lds, SI == (L0G ? 4 : 0)); \
    GVL_FWD_ROWS(row, v0, v1)                                                   \
    GVL_FWD_ACC(SI, v0, v1)                                                     \
  }
    // The fences after every four steps bound the reads in flight (left alone the scheduler requests all 32 rows of a pass up
    // front and, with the coefficients of kAhead passes live, spills) and pin the accumulators: without AMAX their only use
    // is the guarded store, and the optimiser sinks the whole FMA chain of a pass into that guard -- behind all 32 reads.
    // Inside a group the compiler keeps two reads in flight.  Requesting a group's eight reads together was measured again
    // with prep() out of the loop (tools/fwd_ab.sh): loop 3.27 -> 3.41 us in situ.  Neither the LDS round trips nor the LDS
    // bandwidth set the pass time: with the reads REMOVED (-DGVL_FWD_ABL_NO_LDS) the loop still takes 2.99 us -- it is the
    // issue of the 64 v_pk_fma_f32 per wavefront pass plus 56 other vector instructions (two 32-bit broadcasts instead of
    // v_mov_b64_dpp: 3.21 -> 3.24 us; the broadcast folded into 128 v_fmac_f32_dpp: 3.33 -> 3.94 us).
#define GVL_FWD_FENCE asm volatile("" : "+v"(a01), "+v"(a23) : : "memory"); __builtin_amdgcn_sched_barrier(0);
#define GVL_FWD_QUAD(S0) GVL_FWD_STEP(S0) GVL_FWD_STEP(S0 + 1) GVL_FWD_STEP(S0 + 2) GVL_FWD_STEP(S0 + 3) GVL_FWD_FENCE
    if constexpr (L0G) {
      // level 0 from global memory / L2: its eight rows are requested first and consumed last, the twelve LDS steps run
      // while they travel
      float4 g0[4], g1[4];
#define GVL_FWD_G(SI)                                                           \
  {                                                                             \
    const int ro = row_bcast_i<SI>(roff);                                       \
    g0[SI] = ld4(value, vg + (int64_t)ro * (M * 16));                           \
    g1[SI] = ld4(value, vg + (int64_t)min(ro + 1, S - 1) * (M * 16));           \
  }
      GVL_FWD_G(0) GVL_FWD_G(1) GVL_FWD_G(2) GVL_FWD_G(3)
#undef GVL_FWD_G
      GVL_FWD_FENCE
      GVL_FWD_QUAD(4) GVL_FWD_QUAD(8) GVL_FWD_QUAD(12)
      GVL_FWD_ACC(0, g0[0], g1[0]) GVL_FWD_ACC(1, g0[1], g1[1]) GVL_FWD_ACC(2, g0[2], g1[2]) GVL_FWD_ACC(3, g0[3], g1[3])
    } else if constexpr (FULL16) {
      GVL_FWD_QUAD(0) GVL_FWD_QUAD(4) GVL_FWD_QUAD(8) GVL_FWD_QUAD(12)
    } else {
      GVL_FWD_STEP(0) GVL_FWD_STEP(1) GVL_FWD_STEP(2) GVL_FWD_STEP(3)
      GVL_FWD_STEP(4) GVL_FWD_STEP(5) GVL_FWD_STEP(6) GVL_FWD_STEP(7)
      GVL_FWD_STEP(8) GVL_FWD_STEP(9) GVL_FWD_STEP(10) GVL_FWD_STEP(11)
      GVL_FWD_STEP(12) GVL_FWD_STEP(13) GVL_FWD_STEP(14) GVL_FWD_STEP(15)
    }
    GVL_FWD_FENCE
#undef GVL_FWD_QUAD
#undef GVL_FWD_FENCE
#undef GVL_FWD_STEP
#undef GVL_FWD_ACC
#undef GVL_FWD_ROWS
    const float4 acc = make_float4(a01.x, a01.y, a23.x, a23.y);
    const int64_t bq_u = (int64_t)b * Q + qb;                          // (uniform) the wavefront's first query of this pass
    if (act) st4_stream(out + (bq_u * M + m) * 64, (int64_t)e_out, acc);
    if (AMAX) {
      const float mx = row_allmax(fmaxf(fmaxf(fabsf(acc.x), fabsf(acc.y)), fmaxf(fabsf(acc.z), fabsf(acc.w))));
      if (act && j == 0) atomicMax(amax_out + bq_u + tq, __float_as_uint(mx));
    }
    __builtin_amdgcn_sched_barrier(0);
  };
#pragma unroll
  for (int k = 0; k < kAhead; ++k) {
    if (qb < q1) run_pass(roff_a[k], cc_a[k]);
    qb += pstep;
  }
  for (; qb < q1; qb += pstep) {
    // operands two passes ahead are requested now; the coefficients of the NEXT pass are computed at the end of this
    // one, from operands requested one pass ago
    const RawOps r_next = r_n;
    r_n = fetch_next();
    run_pass(roff_c, cc_c);
    if (qb + pstep < q1) prep(r_next, roff_c, cc_c);
  }
  if (stamps) {
    __syncthreads();
    if (threadIdx.x == 0) stamps[wg_id * 4 + 2] = wall_clock64();
  }
}

// ------------------------------------------------------------------------------------------------------
// t1d_d64 backward.  No float atomics anywhere: LDS float atomic adds execute at ~3 cycles PER LANE on gfx950
// (tools/ubench/lds_atomics.hip: ds_add_f32 200 cycles per wave-instruction vs 35 for ds_add_u32), and global
// float atomics are bounded at ~1.3 TB/s chip-wide.  grad_value is therefore produced by a GATHER:
//   phase 1  DPP rows (16 lanes = one (b,q,m)) compute the interpolation coefficients, the two dot products per
//            sample against the LDS value slab (DPP row all-reduce) -> grad_attn / grad_loc, and record for every
//            sample an entry (slab row r, w*c_lo, w*c_hi); integer LDS atomics histogram the entries per slab row;
//   phase 2  block-wide exclusive scan of the histogram, counting-sort of the entry ids by slab row (ds_add_rtn_u32),
//            the grad_out rows of this workgroup's queries are staged into LDS over the (now dead) value slab;
//   phase 3  one wavefront per slab row, lane = channel: the row's entries (those with r == s use c_lo, those with
//            r == s-1 use c_hi) are loaded 64 at a time, (query, coefficient) are broadcast with v_readlane and the
//            query's grad_out row is read conflict-free from LDS: grad_value[s] = sum coef * grad_out[q].
// grad_value rows are written with plain coalesced stores.  When the queries of a slab do not fit one LDS carve-up they
// are cut into chunks that the slab's workgroups walk in turn, adding each chunk's rows to what the earlier chunks
// left (the row is private to the workgroup); only when a (b,m) slab is shared by several workgroups (B*M < 256) do
// partial slabs + k_sum_partials remain, one per workgroup.
// ------------------------------------------------------------------------------------------------------
// Gather steps of the backward's phase 3: list elements S0.. of a 16-entry batch `tc` = (LDS byte offset of the query's
// grad_out row, coefficient) per lane; each step broadcasts one element over the DPP row, reads the row (one conflict-free
// ds_read_b128 per lane) and accumulates.  The reads of a group are requested TOGETHER and consumed afterwards: written
// step by step the compiler waits for every read before it issues the next (one LDS round trip per step).
template <int S0>
__device__ inline void gather4(const f2v tc, const char *G_b, int lane_off, f2v &a01, f2v &a23) {
  const f2v t0 = row_bcast_f2<S0>(tc), t1 = row_bcast_f2<S0 + 1>(tc), t2 = row_bcast_f2<S0 + 2>(tc),
            t3 = row_bcast_f2<S0 + 3>(tc);
  const float4 g0 = *reinterpret_cast<const float4 *>(G_b + __builtin_bit_cast(int, t0.x) + lane_off);
  const float4 g1 = *reinterpret_cast<const float4 *>(G_b + __builtin_bit_cast(int, t1.x) + lane_off);
  const float4 g2 = *reinterpret_cast<const float4 *>(G_b + __builtin_bit_cast(int, t2.x) + lane_off);
  const float4 g3 = *reinterpret_cast<const float4 *>(G_b + __builtin_bit_cast(int, t3.x) + lane_off);
#define GVL_ACC(T, G)                                                           \
  {                                                                             \
    const f2v cf = __builtin_shufflevector(T, T, 1, 1);                         \
    a01 = __builtin_elementwise_fma(cf, (f2v){G.x, G.y}, a01);                  \
    a23 = __builtin_elementwise_fma(cf, (f2v){G.z, G.w}, a23);                  \
  }
  GVL_ACC(t0, g0) GVL_ACC(t1, g1) GVL_ACC(t2, g2) GVL_ACC(t3, g3)
}
template <int S0>
__device__ inline void gather8(const f2v tc, const char *G_b, int lane_off, f2v &a01, f2v &a23) {
  const f2v t0 = row_bcast_f2<S0>(tc), t1 = row_bcast_f2<S0 + 1>(tc), t2 = row_bcast_f2<S0 + 2>(tc),
            t3 = row_bcast_f2<S0 + 3>(tc), t4 = row_bcast_f2<S0 + 4>(tc), t5 = row_bcast_f2<S0 + 5>(tc),
            t6 = row_bcast_f2<S0 + 6>(tc), t7 = row_bcast_f2<S0 + 7>(tc);
  const float4 g0 = *reinterpret_cast<const float4 *>(G_b + __builtin_bit_cast(int, t0.x) + lane_off);
  const float4 g1 = *reinterpret_cast<const float4 *>(G_b + __builtin_bit_cast(int, t1.x) + lane_off);
  const float4 g2 = *reinterpret_cast<const float4 *>(G_b + __builtin_bit_cast(int, t2.x) + lane_off);
  const float4 g3 = *reinterpret_cast<const float4 *>(G_b + __builtin_bit_cast(int, t3.x) + lane_off);
  const float4 g4 = *reinterpret_cast<const float4 *>(G_b + __builtin_bit_cast(int, t4.x) + lane_off);
  const float4 g5 = *reinterpret_cast<const float4 *>(G_b + __builtin_bit_cast(int, t5.x) + lane_off);
  const float4 g6 = *reinterpret_cast<const float4 *>(G_b + __builtin_bit_cast(int, t6.x) + lane_off);
  const float4 g7 = *reinterpret_cast<const float4 *>(G_b + __builtin_bit_cast(int, t7.x) + lane_off);
  GVL_ACC(t0, g0) GVL_ACC(t1, g1) GVL_ACC(t2, g2) GVL_ACC(t3, g3)
  GVL_ACC(t4, g4) GVL_ACC(t5, g5) GVL_ACC(t6, g6) GVL_ACC(t7, g7)
#undef GVL_ACC
}

// LANE = SAMPLE own pass (round 6, GVL_BWD_LS; -DGVL_BWD_LS=0 is the lane = channels form of rounds 2-5).
#ifndef GVL_BWD_LS
#define GVL_BWD_LS 1
#endif
#ifndef GVL_BWD_LS_GROUP
#define GVL_BWD_LS_GROUP 2
#endif
// ------------------------------------------------------------------------------------------------------
// The two dot products of ONE sample per lane -- g[q] . V[r_j] and g[q] . V[r_j + 1] over all 64 channels (round 6; shared by
// the temporal backward kernels wherever the sample's rows are in LDS).  The query's grad_out row lives four channels per lane
// (as loaded).  At step K the lane reads the 16-byte column (j -+ K) mod 16 of ITS two rows -- the sixteen lanes of a DPP row
// then read sixteen DIFFERENT columns whatever their rows: no bank conflicts -- and multiplies with the grad_out channels of
// that column, which arrive from the lane that holds them by the same row rotation as the column's offset.  Against lane =
// channels (every lane a 4-channel slice of ALL sixteen samples' products, then two 16-value reduce-scatters over the DPP row
// to bring sample j's sums to lane j): 128 FMAs + ~90 instructions of reduce-scatter + 32 of broadcasts / addresses become 80
// row rotations + 64 packed FMAs + 2 adds (own pass of k_bwd_t1d_split 7.05 -> 6.35 us, profiles/r06_bwd_experiments.txt).
// The same products, added in another order (channel pairs in two chains, joined at the end).
// FENCED: the row requests of step group n + 1 in front of the products of group n, held in that order by scheduling fences (the
// level-split kernel: own pass 6.51 -> 6.34 us); without, the compiler's own order (the other kernels, where the fences cost
// registers the kernels do not have).
template <bool FENCED>
__device__ __forceinline__ void ls_dot_pair(const float4 *slab4, const int row, const float4 g, const int j, float &d0, float &d1) {
  f2v s0 = {0.f, 0.f}, s1 = {0.f, 0.f};
  {
      const int vaddr = (int)(uintptr_t)(lds_cbyte *)reinterpret_cast<const char *>(slab4) + row * 256;
      const int jo = j * 16;
      // (the column's offset travels by the same rotation as the channels: the direction of row_ror does not matter)
#define GVL_ROR(K, V) ((K) == 0 ? (V) : dpp_f<(K) ? 0x120 + (K) : 0x121>(V))
      // Software pipeline in groups of GVL_BWD_LS_GROUP steps: the rows of group n + 1 are requested before the products of
      // group n are taken (left to itself the compiler requests two rows, waits for both, multiplies, and only then requests the
      // next two: sixteen exposed LDS round trips per pass with four wavefronts per SIMD to hide them).  The scheduling fences keep
      // the order; the waits the compiler inserts are counted ones (the newer group stays in flight).
#define GVL_LS_ISSUE(K)                                                       \
  const int co##K = (K) == 0 ? jo : dpp_i<(K) ? 0x120 + (K) : 0x121>(jo);     \
  lds_cbyte *vc##K = (lds_cbyte *)(uintptr_t)(unsigned)(vaddr + co##K);       \
  const float4 va##K = lds_ld4(vc##K), vb##K = lds_ld4(vc##K + 256);
#define GVL_LS_USE(K)                                                         \
  {                                                                           \
    const f2v bxy = {GVL_ROR(K, g.x), GVL_ROR(K, g.y)}, bzw = {GVL_ROR(K, g.z), GVL_ROR(K, g.w)};  \
    s0 = __builtin_elementwise_fma(bxy, (f2v){va##K.x, va##K.y}, s0);         \
    s0 = __builtin_elementwise_fma(bzw, (f2v){va##K.z, va##K.w}, s0);         \
    s1 = __builtin_elementwise_fma(bxy, (f2v){vb##K.x, vb##K.y}, s1);         \
    s1 = __builtin_elementwise_fma(bzw, (f2v){vb##K.z, vb##K.w}, s1);         \
  }
#define GVL_LS_FENCE if constexpr (FENCED) __builtin_amdgcn_sched_barrier(0);
#if GVL_BWD_LS_GROUP == 4
      GVL_LS_ISSUE(0) GVL_LS_ISSUE(1) GVL_LS_ISSUE(2) GVL_LS_ISSUE(3) GVL_LS_FENCE
      GVL_LS_ISSUE(4) GVL_LS_ISSUE(5) GVL_LS_ISSUE(6) GVL_LS_ISSUE(7) GVL_LS_FENCE
      GVL_LS_USE(0) GVL_LS_USE(1) GVL_LS_USE(2) GVL_LS_USE(3) GVL_LS_FENCE
      GVL_LS_ISSUE(8) GVL_LS_ISSUE(9) GVL_LS_ISSUE(10) GVL_LS_ISSUE(11) GVL_LS_FENCE
      GVL_LS_USE(4) GVL_LS_USE(5) GVL_LS_USE(6) GVL_LS_USE(7) GVL_LS_FENCE
      GVL_LS_ISSUE(12) GVL_LS_ISSUE(13) GVL_LS_ISSUE(14) GVL_LS_ISSUE(15) GVL_LS_FENCE
      GVL_LS_USE(8) GVL_LS_USE(9) GVL_LS_USE(10) GVL_LS_USE(11) GVL_LS_FENCE
      GVL_LS_USE(12) GVL_LS_USE(13) GVL_LS_USE(14) GVL_LS_USE(15) GVL_LS_FENCE
#else
      GVL_LS_ISSUE(0) GVL_LS_ISSUE(1) GVL_LS_FENCE
      GVL_LS_ISSUE(2) GVL_LS_ISSUE(3) GVL_LS_FENCE GVL_LS_USE(0) GVL_LS_USE(1) GVL_LS_FENCE
      GVL_LS_ISSUE(4) GVL_LS_ISSUE(5) GVL_LS_FENCE GVL_LS_USE(2) GVL_LS_USE(3) GVL_LS_FENCE
      GVL_LS_ISSUE(6) GVL_LS_ISSUE(7) GVL_LS_FENCE GVL_LS_USE(4) GVL_LS_USE(5) GVL_LS_FENCE
      GVL_LS_ISSUE(8) GVL_LS_ISSUE(9) GVL_LS_FENCE GVL_LS_USE(6) GVL_LS_USE(7) GVL_LS_FENCE
      GVL_LS_ISSUE(10) GVL_LS_ISSUE(11) GVL_LS_FENCE GVL_LS_USE(8) GVL_LS_USE(9) GVL_LS_FENCE
      GVL_LS_ISSUE(12) GVL_LS_ISSUE(13) GVL_LS_FENCE GVL_LS_USE(10) GVL_LS_USE(11) GVL_LS_FENCE
      GVL_LS_ISSUE(14) GVL_LS_ISSUE(15) GVL_LS_FENCE GVL_LS_USE(12) GVL_LS_USE(13) GVL_LS_FENCE
      GVL_LS_USE(14) GVL_LS_USE(15) GVL_LS_FENCE
#endif
#undef GVL_LS_FENCE
#undef GVL_LS_USE
#undef GVL_LS_ISSUE
#undef GVL_ROR
    }
  d0 = s0.x + s0.y;
  d1 = s1.x + s1.y;
}

constexpr int kBwdThreads = 1024;
constexpr int kEntStride = 16;       // entry slot = q_local * 16 + sample

// rowsV = number of value rows staged in LDS (+1 pad): S + 1, or S - T_0 + 1 when level 0 stays in global memory
__host__ __device__ inline size_t bwd_lds_bytes(int S, int nq, int rowsV) {
  const size_t regionA = (size_t)(rowsV > nq ? rowsV : nq) * 64 * sizeof(float);
  const size_t hist = (size_t)(S + 2) * 2 * sizeof(int);
  const size_t ents = (size_t)nq * kEntStride * 4 * sizeof(int);
  return regionA + hist + ents;
}

// FUSED: loc -> proj, attn -> ref (see fetch_ops); gloc -> grad_proj (B*Q, 2*M*LP), gattn -> grad_ref partials
// (B,Q,M,L,RD) or nullptr.  The softmax / location backward of ms_deform_attn.py:99-109 is applied in the epilogue.
// Storage type VT (fp32 | bf16): value, grad_out and -- FUSED -- proj / grad_proj.  grad_value leaves this kernel in
// fp32 (`gvalue_part`): the final tensor itself when VT = float and one workgroup owns the slab, else partial slabs.
template <int PAD, bool FULL16, bool FUSED, bool L0G, bool LOOP, typename VT>
__global__ void __launch_bounds__(kBwdThreads) k_bwd_t1d_d64(const VT *__restrict__ value,
                                                             const int64_t *__restrict__ shapes,
                                                             const int64_t *__restrict__ lsi,
                                                             const void *__restrict__ loc,
                                                             const float *__restrict__ attn,
                                                             const VT *__restrict__ gout, int B, int S, int M, int L,
                                                             int Q, int P, int RD, int nchunk, int ngroup, int qper,
                                                             float *__restrict__ gvalue_part,
                                                             void *__restrict__ gloc, float *__restrict__ gattn,
                                                             unsigned long long *__restrict__ stamps) {
  extern __shared__ float4 slab4[];
  __shared__ int next_blk_s;
  int *next_blk = &next_blk_s;
  // diagnostics (gvl_msda_debug_stamps): {start, staged, phase 1 done, phase 2 done}; the kernel end closes phase 3
  if (stamps && threadIdx.x == 0) stamps[blockIdx.x * 4 + 0] = wall_clock64();
  const int row0 = L0G ? __builtin_amdgcn_readfirstlane((int)shapes[1]) : 0;    // see k_fwd_t1d_d64
  const int rowsV = S - row0 + 1;
  const int rowsA = (rowsV > qper ? rowsV : qper);
  int *cnt = reinterpret_cast<int *>(slab4 + (size_t)rowsA * 16);     // [S+2] histogram, later the fill cursor
  int *off = cnt + (S + 2);                                           // [S+2] exclusive prefix
  int *ent_r = off + (S + 2);                                         // [qper*16] slab row or -1
  float *ent_lo = reinterpret_cast<float *>(ent_r + qper * kEntStride);
  float *ent_hi = ent_lo + qper * kEntStride;
  int *sorted = reinterpret_cast<int *>(ent_hi + qper * kEntStride);

  const int BM = B * M;
  const int bm = slab_of_block(blockIdx.x % BM, BM), wgc = blockIdx.x / BM;
  const int b = bm / M, m = bm % M;
  const int lane = threadIdx.x & 63, j = lane & 15, tq = lane >> 4;
  const int wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const int LP = FULL16 ? 16 : L * P;
  int Tl = 1, st = 0;
  const int lvl = j < LP ? j / P : 0;
  if (j < LP) {
    Tl = (int)shapes[2 * lvl + 1];
    st = (int)lsi[lvl];
  }
  const float invT = 1.f / (float)Tl, invP = 1.f / (float)P;
  // The queries of a (b,m) slab are cut into nchunk chunks that fit the LDS carve-up; the slab's `ngroup` workgroups
  // take them round-robin and ACCUMULATE their grad_value rows in ONE slab per workgroup (first chunk stores, later
  // chunks add: a row is read and written by waves of this workgroup only, barriers in between), so the number of
  // partial slabs is ngroup (1 when B*M already covers the chip), not nchunk.
  // (LOOP = false: every workgroup has exactly one chunk -- the compile-time single trip keeps the scalar registers of
  // the common small-T case out of spill territory)
  int chunk = wgc;
  do {
  if (LOOP && chunk != wgc) {
    __syncthreads();                                  // the previous chunk's gather is done with the LDS
    if (stamps && threadIdx.x == 0) stamps[blockIdx.x * 4 + 0] = wall_clock64();   // stamps describe the LAST chunk
  }
  const int q0 = chunk * qper;
  const int q1 = max(q0, min(Q, q0 + qper));
  const int nq = q1 - q0;
  // operands of the first pass are requested before the slab staging; every pass prefetches the next one's
  int qb = q0 + wave * 4;
  RawOps r_n = {0.f, 0.5f, 0.f, 0.f};
  float4 g_n = make_float4(0.f, 0.f, 0.f, 0.f);
  if (qb < q1) {
    const int64_t bqn = (int64_t)b * Q + min(qb + tq, q1 - 1);
    if (j < LP) r_n = fetch_ops<FUSED, VT>(loc, attn, bqn, m, M, LP, L, RD, j, lvl);
    if (qb + tq < q1) g_n = ld4(gout, (bqn * M + m) * 16 + j);
  }
  const int64_t vg = ((int64_t)b * S * M + m) * 16 + j;
  stage_slab(slab4, value, b, m, S, M, row0);
  for (int i = threadIdx.x; i < S + 2; i += blockDim.x) cnt[i] = 0;
  for (int i = threadIdx.x; i < qper * kEntStride; i += blockDim.x) ent_r[i] = -1;
  __syncthreads();
  if (stamps && threadIdx.x == 0) stamps[blockIdx.x * 4 + 1] = wall_clock64();

  // ---- phase 1 ---------------------------------------------------------------------------------------------
  GVL_TOUCH_PREFETCH(r_n, g_n)                           // (the first pass: its operands were requested before the staging)
  for (; qb < q1; qb += nw * 4) {
    const int q = qb + tq;
    const bool act = q < q1;
    const int qq = act ? q : q1 - 1;
    const int64_t tb = (((int64_t)b * Q + qq) * M + m) * LP;
    const RawOps r = r_n;
    const float4 g = g_n;
    const int qbn = qb + nw * 4;
    if (qbn < q1) {
      const int64_t bqn = (int64_t)b * Q + min(qbn + tq, q1 - 1);
      if (j < LP) r_n = fetch_ops<FUSED, VT>(loc, attn, bqn, m, M, LP, L, RD, j, lvl);
      g_n = (qbn + tq < q1) ? ld4(gout, (bqn * M + m) * 16 + j) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    float2 xy;
    float w, dloc;
    resolve_ops<FUSED>(r, invT, invP, RD, xy.x, xy.y, w, dloc);
    int roff = 0;
    float clo = 0.f, chi = 0.f, dxlo = 0.f, dxhi = 0.f, dylo = 0.f, dyhi = 0.f;
    if (j < LP) {
      const Coef1D c = coef_1d<PAD>(xy.x, xy.y, Tl);
      roff = st + c.r;
      clo = c.c_lo * c.wy;                 // sample      = clo * V[r] + chi * V[r+1]
      chi = c.c_hi * c.wy;
      dxlo = c.dx_lo * c.wy * w;           // d out/d x   = dxlo * V[r] + dxhi * V[r+1]   (cuh:158)
      dxhi = c.dx_hi * c.wy * w;
      dylo = c.c_lo * c.dy * w;            // d out/d y   (cuh:159; H = 1)
      dyhi = c.c_hi * c.dy * w;
      const float elo = clo * w, ehi = chi * w;
      if (act && (elo != 0.f || ehi != 0.f)) {          // grad_value += elo * g on row r, ehi * g on row r+1
        const int e = (q - q0) * kEntStride + j;
        ent_r[e] = roff;
        ent_lo[e] = elo;
        ent_hi[e] = ehi;
        atomicAdd(&cnt[roff], 1);
      }
    }
    float d0, d1;
    if constexpr (GVL_BWD_LS && FUSED && FULL16 && !L0G) {
      ls_dot_pair<false>(slab4, roff, g, j, d0, d1);      // one sample per lane (every row in LDS, sixteen samples per query)
    } else {
    // per sample the two dot products g . V[r], g . V[r+1]: every lane accumulates its 4-channel part for all 16
    // samples, then two DPP-row reduce-scatters leave sample j's sums on lane j (the lane that holds its coefficients)
    float p0[16], p1[16];
#define GVL_BWD_STEP(SI)                                                      \
  if (FULL16 || SI < LP) {                                                    \
    const int rr = row_bcast_i<SI>(roff);                                     \
    float4 v0, v1;                                                            \
    if (L0G && SI < 4) {                                                      \
      v0 = ld4(value, vg + (int64_t)rr * (M * 16));                           \
      v1 = ld4(value, vg + (int64_t)min(rr + 1, S - 1) * (M * 16));           \
    } else {                                                                  \
      v0 = slab4[(rr - row0) * 16 + j];                                       \
      v1 = slab4[(rr - row0) * 16 + 16 + j];                                  \
    }                                                                         \
    p0[SI] = dot4(g, v0);                                                     \
    p1[SI] = dot4(g, v1);                                                     \
  } else {                                                                    \
    p0[SI] = 0.f;                                                             \
    p1[SI] = 0.f;                                                             \
  }
    GVL_BWD_STEP(0) GVL_BWD_STEP(1) GVL_BWD_STEP(2) GVL_BWD_STEP(3)
    GVL_BWD_STEP(4) GVL_BWD_STEP(5) GVL_BWD_STEP(6) GVL_BWD_STEP(7)
    GVL_BWD_STEP(8) GVL_BWD_STEP(9) GVL_BWD_STEP(10) GVL_BWD_STEP(11)
    GVL_BWD_STEP(12) GVL_BWD_STEP(13) GVL_BWD_STEP(14) GVL_BWD_STEP(15)
#undef GVL_BWD_STEP
    d0 = row_reduce_scatter16(p0, j);
    d1 = row_reduce_scatter16(p1, j);
    }
    const float keep_w = fmaf(clo, d0, chi * d1);
    const float keep_x = fmaf(dxlo, d0, dxhi * d1);
    const float keep_y = fmaf(dylo, d0, dyhi * d1);
    GVL_TOUCH_PREFETCH(r_n, g_n)
    if (!FUSED) {
      if (act && j < LP) {
        st_stream(gattn + tb + j, keep_w);                                     // cuh:156-157
        st_stream(reinterpret_cast<float2 *>(gloc) + tb + j, make_float2(keep_x, keep_y));   // unfused: grad_loc is fp32
      }
    } else {
      // softmax backward (ms_deform_attn.py:100-101): d logit_j = w_j (g_j - sum_k w_k g_k), g = d out / d w
      const float dsum = row_allsum(w * keep_w);
      const float glogit = w * (keep_w - dsum);
      const float goff = keep_x * dloc;                                        // d loc / d offset (:103-109)
      // d loc / d ref: the P points of a level share its reference point -> sum over the quad (P == 4)
      float gr0 = keep_x, gr1 = keep_x * r.a * (0.5f * invP);
      gr0 += dpp_f<0xB1>(gr0); gr0 += dpp_f<0x4E>(gr0);
      gr1 += dpp_f<0xB1>(gr1); gr1 += dpp_f<0x4E>(gr1);
      if (act) {
        VT *grow = reinterpret_cast<VT *>(gloc) + ((int64_t)b * Q + qq) * (int64_t)(2 * M * LP);
        grow[m * LP + j] = (VT)goff;
        grow[M * LP + m * LP + j] = (VT)glogit;
        if (gattn && (j & 3) == 0) {
          float *gr = gattn + ((((int64_t)b * Q + qq) * M + m) * L + lvl) * RD;
          gr[0] = gr0;
          if (RD == 2) gr[1] = gr1;
        }
      }
    }
  }
  __syncthreads();
  if (stamps && threadIdx.x == 0) stamps[blockIdx.x * 4 + 2] = wall_clock64();

  // ---- phase 2: exclusive scan of cnt -> off, reset cnt as the fill cursor; stage grad_out rows over the slab ----
  {
    __shared__ int wave_tot[kBwdThreads / 64];
    __shared__ int carry_s;
    if (threadIdx.x == 0) { carry_s = 0; next_blk_s = 0; }
    __syncthreads();
    for (int base = 0; base < S + 1; base += blockDim.x) {
      const int i = base + threadIdx.x;
      const int v = (i < S + 1) ? cnt[i] : 0;
      int incl = v;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const int t_ = __shfl_up(incl, o, 64);
        if (lane >= o) incl += t_;
      }
      if (lane == 63) wave_tot[wave] = incl;
      __syncthreads();
      int pre = carry_s;
      for (int k = 0; k < wave; ++k) pre += wave_tot[k];
      if (i < S + 1) { off[i] = pre + incl - v; cnt[i] = 0; }
      __syncthreads();
      if (threadIdx.x == blockDim.x - 1) carry_s = pre + incl;
      __syncthreads();
    }
    if (threadIdx.x == 0) off[S + 1] = carry_s;
  }
  float4 *G4 = slab4;                                                 // value slab is dead from here on
  {
    const int64_t src = (((int64_t)b * Q + q0) * M + m) * 16;
    for (int i = threadIdx.x; i < nq * 16; i += blockDim.x) G4[i] = ld4(gout, src + (int64_t)(i >> 4) * M * 16 + (i & 15));
  }
  __syncthreads();
  for (int e = threadIdx.x; e < nq * kEntStride; e += blockDim.x) {
    const int r = ent_r[e];
    if (r >= 0) sorted[off[r] + atomicAdd(&cnt[r], 1)] = e;
  }
  __syncthreads();
  if (stamps && threadIdx.x == 0) stamps[blockIdx.x * 4 + 3] = wall_clock64();

  // ---- phase 3: gather.  A DPP row (16 lanes x float4) owns one slab row; the four rows of a wavefront are
  // neighbours (similar list lengths).  Lane j fetches entry j of the batch, (query, coefficient) are broadcast
  // inside the row with row_newbcast and the query's grad_out row is one conflict-free ds_read_b128 per lane.
  float4 *dst4 = reinterpret_cast<float4 *>(gvalue_part) + ((int64_t)wgc * B * S * M) * 16 + ((int64_t)b * S * M + m) * 16;
  const bool accumulate = LOOP && chunk != wgc;
  const int ngroups = blockDim.x >> 4;
  const char *G_b = reinterpret_cast<const char *>(G4);
  const int lane_off = j * 16;
  // Rows are handed out dynamically, four adjacent rows per wavefront, from the LAST row down: the coarse levels at the
  // end of the slab have the longest entry lists (the same 4 points per query over 13 instead of 100 rows), and a
  // static row -> group map left most wavefronts idle while a few worked through them.
  (void)ngroups;
  for (;;) {
    int blk = 0;
    if (lane == 0) blk = atomicAdd(next_blk, 1);
    blk = __builtin_amdgcn_readfirstlane(blk);
    if (blk * 4 >= S) break;
    const int s = S - 1 - (blk * 4 + tq);
    const bool live = s >= 0;
    const int sc = live ? s : 0;
    const int a1 = off[sc], n1 = live ? off[sc + 1] - a1 : 0;           // entries with r == s     -> c_lo
    const int a0 = sc > 0 ? off[sc - 1] : 0, n0 = (live && sc > 0) ? a1 - a0 : 0;  // entries with r == s - 1 -> c_hi
    const int n_own = n1 + n0;
    // the four rows of the wavefront run in lockstep: loop to the longest of their lists
    int n = max(n_own, __shfl_xor(n_own, 16, 64));
    n = max(n, __shfl_xor(n, 32, 64));
    f2v a01 = {0.f, 0.f}, a23 = {0.f, 0.f};
    // later chunks leave the rows none of their samples touched alone (encoder self-attention: a chunk of neighbouring
    // queries reaches a window of each level, not the whole slab)
    const bool touch = live && (!accumulate || n_own > 0);
    if (accumulate && touch) {                         // the row's sum over the earlier chunks; lands during the gather
      const float4 prev = dst4[(int64_t)s * M * 16 + j];
      a01 = (f2v){prev.x, prev.y};
      a23 = (f2v){prev.z, prev.w};
    }
    for (int base = 0; base < n; base += 16) {
      const int i = base + j;
      // (byte offset of the query's grad_out row, coefficient) travel as ONE 64-bit row broadcast; past the end:
      // (row 0, coefficient 0)
      f2v tc = {0.f, 0.f};
      if (i < n_own) {
        const bool first = i < n1;
        const int e = sorted[first ? a1 + i : a0 + (i - n1)];
        tc = (f2v){__builtin_bit_cast(float, (e & ~15) << 4), first ? ent_lo[e] : ent_hi[e]};
      }
gather4<0>(tc, G_b, lane_off, a01, a23);
      if (n - base > 4) gather4<4>(tc, G_b, lane_off, a01, a23);
      if (n - base > 8) gather8<8>(tc, G_b, lane_off, a01, a23);
    }
    const float4 acc = make_float4(a01.x, a01.y, a23.x, a23.y);
    if (touch) {
      if (LOOP) dst4[(int64_t)s * M * 16 + j] = acc;                 // re-read by this workgroup's later chunks
      else st4_stream(reinterpret_cast<float *>(dst4), (int64_t)s * M * 16 + j, acc);
    }
  }
  } while (LOOP && (chunk += ngroup) < nchunk);
}

// ------------------------------------------------------------------------------------------------------
// t1d_d64 backward, LEVEL-SPLIT form -- for the case where a (b,m) slab is shared by exactly TWO workgroups (B*M = 128
// at cfg A: 256 workgroups, one per CU) and L = P = 4.  The query-split form above lets each of the two workgroups
// produce a partial grad_value slab for its half of the queries and needs k_sum_partials afterwards (12 MB written,
// 12 MB re-read, a second launch: 5 of 28 us and 1.22x the algorithmic HBM traffic, profiles/r01_pmc_traffic.json).
// Here the two workgroups split the OUTPUT instead: workgroup g owns the grad_value rows of two pyramid levels
// ({0,3} / {1,2}: 113 / 75 rows, the same number of samples) and writes them once, final, with plain stores.
// An entry list sorted by row needs EVERY sample that falls into the workgroup's levels, from all Q queries, so
//   own pass      the workgroup's half of the queries, exactly as above (coefficients, 16 sample steps against the LDS
//                 slab, reduce-scatter, grad_loc / grad_attn or the fused softmax / location epilogue) -- entries are
//                 recorded for the samples of the workgroup's levels only;
//   foreign pass  the OTHER half of the queries: operand fetch + coefficient arithmetic only (no slab reads, no dot
//                 products: ~100 of the ~430 VALU instructions of a pass), entries for the workgroup's levels, and the
//                 queries' grad_out rows go to LDS on the way.  It runs BEFORE the value slab is needed, i.e. while the
//                 slab's global loads are in flight: the staging round trip of the query-split form is hidden.
// Both workgroups read all of loc / attn (proj / ref) and grad_out of the slab's queries; they sit on the same XCD
// (workgroup ids B*M apart), so the second read is an L2 hit.  The counting sort keeps the slot each entry drew from
// the integer histogram in phase 1 (packed beside the row), so the scatter needs no second round of LDS atomics.
// ------------------------------------------------------------------------------------------------------
constexpr int kSplitStride4 = 16;                                      // float4 per staged slab row of k_bwd_t1d_split
__host__ __device__ inline int bwd_split_rows_a(int S, int qper) {     // region A in 256-byte rows: slab, later the own grad_out rows
  const int v = ((S + 1) * kSplitStride4 + 15) / 16;
  return v > qper ? v : qper;
}
__host__ __device__ inline size_t bwd_split_lds_bytes(int S, int qper) {
  // first region: the value slab (S + 1 rows), later reused for the OWN queries' grad_out rows (qper rows) -- sized for both
  const int rowsA = bwd_split_rows_a(S, qper);
  return (size_t)rowsA * 256 + (size_t)qper * 256 + (size_t)(S + 2) * 2 * sizeof(int) + (size_t)(2 * qper * 8) * 4 * sizeof(int);
}
template <int PAD, bool FUSED, typename VT>
__global__ void __launch_bounds__(kBwdThreads) k_bwd_t1d_split(const VT *__restrict__ value,
                                                               const int64_t *__restrict__ shapes,
                                                               const int64_t *__restrict__ lsi,
                                                               const void *__restrict__ loc,
                                                               const float *__restrict__ attn,
                                                               const VT *__restrict__ gout, int B, int S, int M, int Q,
                                                               int RD, int qper, VT *__restrict__ gvalue,
                                                               void *__restrict__ gloc, float *__restrict__ gattn,
                                                               unsigned long long *__restrict__ stamps) {
  constexpr int L = 4, P = 4, LP = 16;
  extern __shared__ float4 slab4[];
  __shared__ int next_blk_s;
  if (stamps && threadIdx.x == 0) stamps[blockIdx.x * 4 + 0] = wall_clock64();
  const int rowsV = bwd_split_rows_a(S, qper);                         // region A: slab, then the own grad_out rows
  float4 *G_oth = slab4 + (size_t)rowsV * 16;                          // grad_out rows of the foreign queries
  int *cnt = reinterpret_cast<int *>(G_oth + (size_t)qper * 16);       // [S+2] histogram
  int *off = cnt + (S + 2);                                            // [S+2] exclusive prefix
  const int nent = 2 * qper * 8;                                       // 8 samples of the owned levels per query
  int *ent_rp = off + (S + 2);                                         // slab row | slot in the row << 12, or -1
  float *ent_lo = reinterpret_cast<float *>(ent_rp + nent);
  float *ent_hi = ent_lo + nent;
  int *sorted = reinterpret_cast<int *>(ent_hi + nent);

  const int BM = B * M;
  const int bm = slab_of_block(blockIdx.x % BM, BM), g = blockIdx.x / BM;   // g in {0, 1}
  const int b = bm / M, m = bm % M;
  const int lane = threadIdx.x & 63, j = lane & 15, tq = lane >> 4;
  // (the wavefront index as a scalar: the first query of a pass, and with it the 64-bit part of every operand / result address,
  // is then wavefront-uniform -- see fetch_ops_u)
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), nw = blockDim.x >> 6;
  const int lvl = j >> 2;
  // (the low dwords of the int64 entries, see k_fwd_t1d_d64: as 64-bit loads the compiler waited for them at once)
  const int Tl = *reinterpret_cast<const int *>(reinterpret_cast<const char *>(shapes) + (unsigned)(16 * lvl + 8));
  const int st = *reinterpret_cast<const int *>(reinterpret_cast<const char *>(lsi) + (unsigned)(8 * lvl));
  // levels {0,3} belong to workgroup 0, {1,2} to workgroup 1; k = the sample's index among the 8 owned samples of a query
  const bool mine = g == 0 ? (lvl == 0 || lvl == 3) : (lvl == 1 || lvl == 2);
  const int k_own = g == 0 ? (lvl == 0 ? j : j - 8) : j - 4;
  const int q0 = g * qper, q1 = max(q0, min(Q, q0 + qper));            // own queries
  const int f0 = (1 - g) * qper, f1 = max(f0, min(Q, f0 + qper));      // foreign queries
  const int nq = q1 - q0;

  // ---- set-up: operands of the first foreign pass and the slab are requested first (one cold round trip for all of
  // them), then the histogram and the entry table are initialised while those loads are in flight --------------------
  // (no guards around these requests -- indices clamped into the lists instead; rows past a list's end are never stored.
  // Behind a guard the compiler cannot count the loads in flight and puts a vmcnt(0) in front of the next request.)
  RawOps rf_n, r_n;
  float4 gf_n, g_n;
  // (qc: the pass's first query clamped into its list, wavefront-uniform; dq: the lane's query relative to it, clamped likewise)
  auto fetch_pass = [&](int qfirst, int lo, int hi, RawOps &r_out, float4 &g_out) {
    const int qc = max(lo, min(qfirst, hi - 1));
    const int dq = max(0, min(tq, hi - 1 - qc));
    const int64_t bq_u = (int64_t)b * Q + qc;
    r_out = fetch_ops_u<FUSED, VT>(loc, attn, bq_u, dq, m, M, LP, L, RD, j, lvl);
    g_out = ld4_u(gout + (bq_u * M + m) * 64, (unsigned)(dq * M * 16 + j));
  };
  fetch_pass(f0 + wave * 4, f0, f1, rf_n, gf_n);
  int qb = q0 + wave * 4;
  fetch_pass(qb, q0, q1, r_n, g_n);                                    // operands of the first OWN pass
  constexpr int kPre = 3;
  const int64_t src0 = ((int64_t)b * S * M + m) * 16;
  const int nstage = S * 16;
  float4 pre[kPre];
#pragma unroll
  for (int k = 0; k < kPre; ++k) {
    const int i = min((int)(threadIdx.x + k * blockDim.x), nstage - 1);
    pre[k] = ld4(value, src0 + (int64_t)(i >> 4) * M * 16 + (i & 15));
  }
  const float invT = 1.f / (float)Tl, invP = 1.f / (float)P;
  for (int i = threadIdx.x; i < S + 2; i += blockDim.x) cnt[i] = 0;
  for (int i = threadIdx.x; i < nent; i += blockDim.x) ent_rp[i] = -1;
  __syncthreads();

  // entry of sample j of slab query `ql` (0 .. 2 qper): row, coefficients, slot in the row's list
  auto record = [&](int ql, int roff, float elo, float ehi) {
    const int e = ql * 8 + k_own;
    const int pos = atomicAdd(&cnt[roff], 1);
    ent_rp[e] = roff | (pos << 12);
    ent_lo[e] = elo;
    ent_hi[e] = ehi;
  };

  // ---- foreign pass: coefficients of the other half's samples in the owned levels + their grad_out rows -----------
  {
    int qb = f0 + wave * 4;
    RawOps r_n = rf_n;
    float4 g_n = gf_n;
    for (; qb < f1; qb += nw * 4) {
      const int q = qb + tq;
      const bool act = q < f1;
      const RawOps r = r_n;
      const float4 gq = g_n;
      const int qbn = qb + nw * 4;
      if (qbn < f1) fetch_pass(qbn, f0, f1, r_n, g_n);
      float2 xy;
      float w, dloc;
      resolve_ops<FUSED>(r, invT, invP, RD, xy.x, xy.y, w, dloc);      // (FUSED: the softmax needs all 16 lanes of the row)
      const Coef1D c = coef_1d<PAD>(xy.x, xy.y, Tl);
      const float elo = c.c_lo * c.wy * w, ehi = c.c_hi * c.wy * w;
      if (act) {
        G_oth[(q - f0) * 16 + j] = gq;
        if (mine && (elo != 0.f || ehi != 0.f)) record(qper + (q - f0), st + c.r, elo, ehi);
      }
    }
  }

  // ---- the slab goes to LDS (the own pass's first operands were requested with the set-up's other loads: requested here,
  // behind a guard, they made the wait for the slab registers a wait for themselves -- a round trip in front of the barrier)
#pragma unroll
  for (int k = 0; k < kPre; ++k) {
    const int i = threadIdx.x + k * blockDim.x;
    if (i < nstage) slab4[(i >> 4) * kSplitStride4 + (i & 15)] = pre[k];
  }
  for (int i = threadIdx.x + kPre * blockDim.x; i < nstage; i += blockDim.x)
    slab4[(i >> 4) * kSplitStride4 + (i & 15)] = ld4(value, src0 + (int64_t)(i >> 4) * M * 16 + (i & 15));
  if (threadIdx.x < 16) slab4[S * kSplitStride4 + threadIdx.x] = make_float4(0.f, 0.f, 0.f, 0.f);
  __syncthreads();
  if (stamps && threadIdx.x == 0) stamps[blockIdx.x * 4 + 1] = wall_clock64();

  // ---- own pass (phase 1 of the query-split kernel; entries for the owned levels only) -----------------------------
  GVL_TOUCH_PREFETCH(r_n, g_n)                           // (the first pass: its operands were requested before the staging)
  for (; qb < q1; qb += nw * 4) {
    const int q = qb + tq;
    const bool act = q < q1;
    const int64_t bq_u = (int64_t)b * Q + qb;                           // (uniform) the wavefront's first query of this pass
    const RawOps r = r_n;
    const float4 gq = g_n;
    const int qbn = qb + nw * 4;
    if (qbn < q1) fetch_pass(qbn, q0, q1, r_n, g_n);
    float2 xy;
    float w, dloc;
    resolve_ops<FUSED>(r, invT, invP, RD, xy.x, xy.y, w, dloc);
    const Coef1D c = coef_1d<PAD>(xy.x, xy.y, Tl);
    const int roff = st + c.r;
    const float clo = c.c_lo * c.wy, chi = c.c_hi * c.wy;
    const float dxlo = c.dx_lo * c.wy * w, dxhi = c.dx_hi * c.wy * w;
    const float dylo = c.c_lo * c.dy * w, dyhi = c.c_hi * c.dy * w;
    {
      const float elo = clo * w, ehi = chi * w;
      if (act && mine && (elo != 0.f || ehi != 0.f)) record(q - q0, roff, elo, ehi);
    }
    // (the unfused instantiations keep lane = channels: with the plain loc / attn operands hipcc hoists all sixteen steps' row
    //  reads above the products and spills ~140 registers at the kernel's 128-register cap)
    float d0, d1;
    if constexpr (GVL_BWD_LS && FUSED) {
      ls_dot_pair<true>(slab4, roff, gq, j, d0, d1);
    } else {

    float p0[16], p1[16];
#define GVL_BWD_STEP(SI)                                                      \
  {                                                                           \
    const int rr = row_bcast_i<SI>(roff);                                     \
    const float4 v0 = slab4[rr * 16 + j], v1 = slab4[rr * 16 + 16 + j];       \
    p0[SI] = dot4(gq, v0);                                                    \
    p1[SI] = dot4(gq, v1);                                                    \
  }
    GVL_BWD_STEP(0) GVL_BWD_STEP(1) GVL_BWD_STEP(2) GVL_BWD_STEP(3)
    GVL_BWD_STEP(4) GVL_BWD_STEP(5) GVL_BWD_STEP(6) GVL_BWD_STEP(7)
    GVL_BWD_STEP(8) GVL_BWD_STEP(9) GVL_BWD_STEP(10) GVL_BWD_STEP(11)
    GVL_BWD_STEP(12) GVL_BWD_STEP(13) GVL_BWD_STEP(14) GVL_BWD_STEP(15)
#undef GVL_BWD_STEP
    d0 = row_reduce_scatter16(p0, j);
    d1 = row_reduce_scatter16(p1, j);
    }
    const float keep_w = fmaf(clo, d0, chi * d1);
    const float keep_x = fmaf(dxlo, d0, dxhi * d1);
    const float keep_y = fmaf(dylo, d0, dyhi * d1);
    GVL_TOUCH_PREFETCH(r_n, g_n)
    if (!FUSED) {
      if (act) {
        const unsigned e = (unsigned)((tq * M + m) * LP + j);
        st_stream(reinterpret_cast<float *>(reinterpret_cast<char *>(gattn + bq_u * (M * LP)) + e * 4u), keep_w);   // cuh:156-157
        st_stream(reinterpret_cast<float2 *>(reinterpret_cast<char *>(reinterpret_cast<float2 *>(gloc) + bq_u * (M * LP)) + e * 8u),
                  make_float2(keep_x, keep_y));
      }
    } else {
      const float dsum = row_allsum(w * keep_w);                               // softmax backward (ms_deform_attn.py:100-101)
      const float glogit = w * (keep_w - dsum);
      const float goff = keep_x * dloc;
      float gr0 = keep_x, gr1 = keep_x * r.a * (0.5f * invP);
      gr0 += dpp_f<0xB1>(gr0); gr0 += dpp_f<0x4E>(gr0);
      gr1 += dpp_f<0xB1>(gr1); gr1 += dpp_f<0x4E>(gr1);
      if (act) {
        char *grow = reinterpret_cast<char *>(reinterpret_cast<VT *>(gloc) + bq_u * (2 * M * LP));
        const unsigned e = (unsigned)(tq * (2 * M * LP) + m * LP + j) * (unsigned)sizeof(VT);
        *reinterpret_cast<VT *>(grow + e) = (VT)goff;
        *reinterpret_cast<VT *>(grow + (e + (unsigned)(M * LP) * (unsigned)sizeof(VT))) = (VT)glogit;
        if (gattn && (j & 3) == 0) {
          char *gr = reinterpret_cast<char *>(gattn + bq_u * (M * L * RD));
          const unsigned eg = (unsigned)(((tq * M + m) * L + lvl) * RD) * 4u;
          *reinterpret_cast<float *>(gr + eg) = gr0;
          if (RD == 2) *reinterpret_cast<float *>(gr + (eg + 4u)) = gr1;
        }
      }
    }
  }
  // grad_out rows of the OWN queries for phase 3: requested now, stored over the value slab once it is dead
  constexpr int kG = 3;
  float4 gpre[kG];
  const int64_t gsrc = (((int64_t)b * Q + q0) * M + m) * 16;
#pragma unroll
  for (int k = 0; k < kG; ++k) {
    const int i = threadIdx.x + k * blockDim.x;
    if (i < nq * 16) gpre[k] = ld4(gout, gsrc + (int64_t)(i >> 4) * M * 16 + (i & 15));
  }
  __syncthreads();
  if (stamps && threadIdx.x == 0) stamps[blockIdx.x * 4 + 2] = wall_clock64();

  // ---- phase 2: exclusive scan of the histogram; own grad_out rows over the slab; entry ids into row order ----------
  float4 *G4 = slab4;
#pragma unroll
  for (int k = 0; k < kG; ++k) {
    const int i = threadIdx.x + k * blockDim.x;
    if (i < nq * 16) G4[i] = gpre[k];
  }
  for (int i = threadIdx.x + kG * blockDim.x; i < nq * 16; i += blockDim.x)
    G4[i] = ld4(gout, gsrc + (int64_t)(i >> 4) * M * 16 + (i & 15));
  {
    __shared__ int wave_tot[kBwdThreads / 64];
    __shared__ int carry_s;
    if (threadIdx.x == 0) { carry_s = 0; next_blk_s = 0; }
    __syncthreads();
    for (int base = 0; base < S + 1; base += blockDim.x) {
      const int i = base + threadIdx.x;
      const int v = (i < S + 1) ? cnt[i] : 0;
      int incl = v;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const int t_ = __shfl_up(incl, o, 64);
        if (lane >= o) incl += t_;
      }
      if (lane == 63) wave_tot[wave] = incl;
      __syncthreads();
      int pre_ = carry_s;
      for (int k = 0; k < wave; ++k) pre_ += wave_tot[k];
      if (i < S + 1) off[i] = pre_ + incl - v;
      __syncthreads();
      if (threadIdx.x == blockDim.x - 1) carry_s = pre_ + incl;
      __syncthreads();
    }
    if (threadIdx.x == 0) off[S + 1] = carry_s;
  }
  for (int e = threadIdx.x; e < nent; e += blockDim.x) {
    const int rp = ent_rp[e];
    if (rp >= 0) sorted[off[rp & 4095] + (rp >> 12)] = e;
  }
  __syncthreads();
  if (stamps && threadIdx.x == 0) stamps[blockIdx.x * 4 + 3] = wall_clock64();

  // ---- phase 3: gather over the OWNED rows, final stores ------------------------------------------------------------
  // Own rows = a FINE level (level 0 | 1: many rows, short entry lists) followed by a COARSE one (level 3 | 2: few rows,
  // the same number of entries -> lists 8x / 2x as long; with ALL queries of the slab in one workgroup they reach
  // ~250 entries).  Work units, handed out dynamically from an LDS counter, longest first:
  //   coarse rows  one row per WAVEFRONT: its four DPP rows take interleaved 16-entry batches of the row's list and
  //                their partial sums meet through two cross-row shuffles (a 250-entry list costs 4 batches, not 16);
  //   fine rows    four adjacent rows per wavefront in lockstep, as in the query-split kernel.
  const int T0 = (int)shapes[1], s3 = (int)lsi[3], s2 = (int)lsi[2];
  const int nF = g == 0 ? T0 : s2 - T0, nC = g == 0 ? S - s3 : s3 - s2;  // fine | coarse rows of this workgroup
  const int sF = g == 0 ? 0 : T0, sC = g == 0 ? s3 : s2;                 // their first slab rows
  const int n_units = nC + (nF + 3) / 4;
  const char *G_b = reinterpret_cast<const char *>(G4);
  const int lane_off = j * 16;
  const int oth_rows = rowsV - qper;          // entry query index ql >= qper lives at LDS row ql + oth_rows (G_oth)
  // (LDS byte offset of the query's grad_out row, coefficient) of list element i of slab row s; past the end: (0, 0)
  auto fetch_tc = [&](int i, int n_own, int n1, int a1, int a0) {
    f2v tc = {0.f, 0.f};
    if (i < n_own) {
      const bool first = i < n1;
      const int e = sorted[first ? a1 + i : a0 + (i - n1)];
      const int ql = e >> 3;
      tc = (f2v){__builtin_bit_cast(float, (ql < qper ? ql : ql + oth_rows) << 8), first ? ent_lo[e] : ent_hi[e]};
    }
    return tc;
  };
#define GVL_GATHER_BATCH(LEFT)                                                                         \
  gather4<0>(tc, G_b, lane_off, a01, a23);                                                             \
  if ((LEFT) > 4) gather4<4>(tc, G_b, lane_off, a01, a23);                                             \
  if ((LEFT) > 8) gather8<8>(tc, G_b, lane_off, a01, a23);
  for (;;) {
    int u = 0;
    if (lane == 0) u = atomicAdd(&next_blk_s, 1);
    u = __builtin_amdgcn_readfirstlane(u);
    if (u >= n_units) break;
    f2v a01 = {0.f, 0.f}, a23 = {0.f, 0.f};
    if (u < nC) {
      // ---- one coarse row for the whole wavefront -----------------------------------------------------------
      const int s = sC + nC - 1 - u;
      const int a1 = off[s], n1 = off[s + 1] - a1;
      const int a0 = off[s - 1], n0 = a1 - a0;                            // (s >= 1: a coarse level never starts the slab)
      const int n_own = n1 + n0;
      f2v tc = fetch_tc(tq * 16 + j, n_own, n1, a1, a0);
      for (int base = 0; base < n_own; base += 64) {
        const f2v tc_next = fetch_tc(base + 64 + tq * 16 + j, n_own, n1, a1, a0);
        const int left = n_own - base - tq * 16;                          // entries left for this DPP row's batch
        if (left > 0) { GVL_GATHER_BATCH(left) }
        tc = tc_next;
      }
      a01 += (f2v){__shfl_xor(a01.x, 16, 64), __shfl_xor(a01.y, 16, 64)};
      a23 += (f2v){__shfl_xor(a23.x, 16, 64), __shfl_xor(a23.y, 16, 64)};
      a01 += (f2v){__shfl_xor(a01.x, 32, 64), __shfl_xor(a01.y, 32, 64)};
      a23 += (f2v){__shfl_xor(a23.x, 32, 64), __shfl_xor(a23.y, 32, 64)};
      if (tq == 0)
        st4_stream(gvalue, ((int64_t)b * S + s) * M * 16 + (int64_t)m * 16 + j, make_float4(a01.x, a01.y, a23.x, a23.y));
    } else {
      // ---- four adjacent fine rows in lockstep ------------------------------------------------------------------
      const int i_row = nF - 1 - ((u - nC) * 4 + tq);
      const bool live = i_row >= 0;
      const int s = sF + (live ? i_row : 0);
      const int a1 = off[s], n1 = live ? off[s + 1] - a1 : 0;
      const int a0 = s > 0 ? off[s - 1] : 0, n0 = (live && s > 0) ? a1 - a0 : 0;
      const int n_own = n1 + n0;
      int n = max(n_own, __shfl_xor(n_own, 16, 64));
      n = max(n, __shfl_xor(n, 32, 64));
      f2v tc = fetch_tc(j, n_own, n1, a1, a0);
      for (int base = 0; base < n; base += 16) {
        const f2v tc_next = fetch_tc(base + 16 + j, n_own, n1, a1, a0);
        GVL_GATHER_BATCH(n - base)
        tc = tc_next;
      }
      if (live) st4_stream(gvalue, ((int64_t)b * S + s) * M * 16 + (int64_t)m * 16 + j, make_float4(a01.x, a01.y, a23.x, a23.y));
    }
  }
#undef GVL_GATHER_BATCH
  if (stamps) {                                                          // diagnostics only: gather done (before the store drain)
    __syncthreads();
    if (threadIdx.x == 0) stamps[(1024 + blockIdx.x) * 4] = wall_clock64();
  }
}

// ------------------------------------------------------------------------------------------------------
// t1d_d64 backward, ROW-OWNERSHIP form -- the level-split idea for slabs whose queries do NOT fit one LDS carve-up (long
// videos: T = 512 gives S = 960 rows and, in the encoder, 960 queries per slab).  The query-chunked form above keeps a
// workgroup-private fp32 partial slab in HBM and read-modify-writes it once per query chunk, then k_sum_partials adds the
// two workgroups' slabs: 4.0x (fp32) / 4.9x (bf16) the algorithmic bytes at T = 512 (profiles/r03_pmc_traffic.json).
// Here the two workgroups of a (b,m) slab split the OUTPUT as in k_bwd_t1d_split -- workgroup g owns the grad_value rows
// of two pyramid levels ({0,3} / {1,2}: the same number of samples) -- and a wavefront keeps the fp32 accumulators of ITS
// rows in registers for the whole kernel: every row is written once, final, in the storage type; no workspace, no second
// kernel.  Two phases:
//   A  the workgroup's half of the queries: coefficients, the 16 sample steps against the LDS slab (L0G: level 0 read
//      from global memory / L2, its eight row loads requested before the twelve LDS steps run), reduce-scatter,
//      grad_loc / grad_attn or the fused softmax / location epilogue.  No entries are kept.
//   B  ALL queries of the slab, in chunks of `qc`: the chunk's grad_out rows go to LDS (over the dead slab), operand
//      fetch + coefficient arithmetic only (the "foreign pass" of the level-split kernel) yield the entries of the owned
//      levels, counting sort by owned row, then the gather.  Rows are assigned STATICALLY: unit u = four adjacent owned
//      rows (one per DPP row, in lockstep) belongs to wavefront u % 16, which visits its units in a fully unrolled loop so
//      that unit i's accumulator is a fixed register quadruple across all chunks.  Fine and coarse units interleave over
//      the wavefronts, so a window of rows that one chunk of neighbouring queries reaches is spread over all of them.
// LDS: max(slab rows * 256, qc * (256 + 8 * 24) + histogram).  Eligibility: L = P = 4, two workgroups per slab, owned
// rows <= 64 * kOwnNU per workgroup; the single-chunk cfg A case stays on k_bwd_t1d_split (dynamic units, one pass).
// Semantics: /root/reference/pdvc/ops/src/cuda/ms_deform_im2col_cuda.cuh:407-511 (col2im: grad_value scatter, grad of
// sampling locations and attention weights), arithmetic shared with the kernels above (coef_1d, resolve_ops).
// ------------------------------------------------------------------------------------------------------
constexpr int kOwnNU = 10;           // units (4 rows) per wavefront: up to 640 owned rows per workgroup

__host__ __device__ inline size_t bwd_own_lds_bytes(int rowsV, int n_own_max, int qc) {
  const size_t a = (size_t)rowsV * 256;
  const size_t b = (size_t)qc * 256 + (size_t)(n_own_max + 2) * 2 * sizeof(int) + (size_t)qc * 8 * 6 * sizeof(int);
  return a > b ? a : b;
}

template <int PAD, bool FUSED, bool L0G, typename VT>
__global__ void __launch_bounds__(kBwdThreads) k_bwd_t1d_own(const VT *__restrict__ value,
                                                             const int64_t *__restrict__ shapes,
                                                             const int64_t *__restrict__ lsi,
                                                             const void *__restrict__ loc,
                                                             const float *__restrict__ attn,
                                                             const VT *__restrict__ gout, int B, int S, int M, int Q,
                                                             int RD, int qc, VT *__restrict__ gvalue,
                                                             void *__restrict__ gloc, float *__restrict__ gattn,
                                                             unsigned long long *__restrict__ stamps, int dbg) {
  constexpr int L = 4, P = 4, LP = 16;
  extern __shared__ float4 slab4[];
  if (stamps && threadIdx.x == 0) stamps[blockIdx.x * 4 + 0] = wall_clock64();
  const int BM = B * M;
  // workgroup id -> (slab, level pair g).  B*M = 128: the two workgroups of a slab are ids B*M apart, both in the first (only)
  // round of 256 and on one XCD.  B*M > 128 (dbg & 8, B*M % 8 == 0): 2 B*M workgroups run in several rounds, so the PAIR gets
  // consecutive positions on its XCD -- ids (x, k) and (x, k + 1) with x = id % 8, k = id / 8 -- and is dispatched together:
  // the second reader of every grad_out / proj row still finds it in the XCD's L2
  const int id = (int)blockIdx.x, kk = id >> 3;
  // (selects on scalars, forced back into scalar registers: as a branch the compiler treated g as a vector value and the
  // kernel went from 5 to 37 spilled vector registers -- 77 -> 84 us)
  const int g = __builtin_amdgcn_readfirstlane((dbg & 8) ? (kk & 1) : id / BM);                  // g in {0, 1}
  const int sidx = __builtin_amdgcn_readfirstlane((dbg & 8) ? (kk >> 1) * 8 + (id & 7) : id % BM);
  const int bm = (dbg & 4) ? sidx : slab_of_block(sidx, BM);
  const int b = bm / M, m = bm % M;
  const int lane = threadIdx.x & 63, j = lane & 15, tq = lane >> 4;
  const int wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const int lvl = j >> 2;
  const int Tl = (int)shapes[2 * lvl + 1], st = (int)lsi[lvl];
  const float invT = 1.f / (float)Tl, invP = 1.f / (float)P;
  const int T0 = (int)shapes[1], s1 = (int)lsi[1], s2 = (int)lsi[2], s3 = (int)lsi[3];
  const int row0 = L0G ? __builtin_amdgcn_readfirstlane(T0) : 0;
  // ownership: levels {0,3} -> workgroup 0, {1,2} -> workgroup 1; owned index o: the fine level's rows, then the coarse one's
  const int nF = g == 0 ? T0 : s2 - s1, nC = g == 0 ? S - s3 : s3 - s2;
  const int sF = g == 0 ? 0 : s1, sC = g == 0 ? s3 : s2;
  const int nOwn = nF + nC;
  const bool mine = g == 0 ? (lvl == 0 || lvl == 3) : (lvl == 1 || lvl == 2);
  const int k_own = g == 0 ? (lvl == 0 ? j : j - 8) : j - 4;
  const int o_base = (g == 0 ? lvl == 0 : lvl == 1) ? 0 : nF;
  const int qh = (Q + 1) / 2;
  const int q0 = g * qh, q1 = max(q0, min(Q, q0 + qh));                 // phase A: own half of the queries
  const int64_t vg = ((int64_t)b * S * M + m) * 16 + j;                // this lane's channels of slab row 0 (global)

  // ================================ phase A =====================================================================
  {
    int qb = q0 + wave * 4;
    RawOps r_n = {0.f, 0.5f, 0.f, 0.f};
    float4 g_n = make_float4(0.f, 0.f, 0.f, 0.f);
    if (qb < q1) {
      const int64_t bqn = (int64_t)b * Q + min(qb + tq, q1 - 1);
      r_n = fetch_ops<FUSED, VT>(loc, attn, bqn, m, M, LP, L, RD, j, lvl);
      if (qb + tq < q1) g_n = ld4(gout, (bqn * M + m) * 16 + j);
    }
    // slab rows [row0, S) -> LDS, four float4 per thread in flight
    {
      const int64_t src0 = ((int64_t)b * S * M + m) * 16;
      const int nstage = (S - row0) * 16;
      for (int base = threadIdx.x; base < nstage; base += 4 * blockDim.x) {
        float4 pre[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int i = base + k * blockDim.x;
          if (i < nstage) pre[k] = ld4(value, src0 + (int64_t)(row0 + (i >> 4)) * M * 16 + (i & 15));
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int i = base + k * blockDim.x;
          if (i < nstage) slab4[i] = pre[k];
        }
      }
      if (threadIdx.x < 16) slab4[nstage + threadIdx.x] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    __syncthreads();
    if (stamps && threadIdx.x == 0) stamps[blockIdx.x * 4 + 1] = wall_clock64();
    if (dbg & 1) qb = q1;                                                  // diagnostics (GVL_MSDA_OWN_DEBUG): traffic of one phase alone
#ifdef GVL_PHASE_TIMING   // dev build: cycles of the blocks of a pass (wave 0), summed over its passes
    long long tacc[5] = {0, 0, 0, 0, 0};
#define GVL_T(K) { const long long t_ = __builtin_amdgcn_s_memtime(); tacc[K] += t_ - tprev; tprev = t_; }
#else
#define GVL_T(K)
#endif
    GVL_TOUCH_PREFETCH(r_n, g_n)
    for (; qb < q1; qb += nw * 4) {
#ifdef GVL_PHASE_TIMING
      long long tprev = __builtin_amdgcn_s_memtime();
#endif
      const int q = qb + tq;
      const bool act = q < q1;
      const int qq = act ? q : q1 - 1;
      const int64_t tb = (((int64_t)b * Q + qq) * M + m) * LP;
      const RawOps r = r_n;
      const float4 gq = g_n;
      const int qbn = qb + nw * 4;
      if (qbn < q1) {
        const int64_t bqn = (int64_t)b * Q + min(qbn + tq, q1 - 1);
        r_n = fetch_ops<FUSED, VT>(loc, attn, bqn, m, M, LP, L, RD, j, lvl);
        g_n = (qbn + tq < q1) ? ld4(gout, (bqn * M + m) * 16 + j) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
      float2 xy;
      float w, dloc;
      resolve_ops<FUSED>(r, invT, invP, RD, xy.x, xy.y, w, dloc);
      const Coef1D c = coef_1d<PAD>(xy.x, xy.y, Tl);
      const int roff = st + c.r;
      const float clo = c.c_lo * c.wy, chi = c.c_hi * c.wy;
      const float dxlo = c.dx_lo * c.wy * w, dxhi = c.dx_hi * c.wy * w;
      const float dylo = c.c_lo * c.dy * w, dyhi = c.c_hi * c.dy * w;
      GVL_T(0)                                                               // operand hand-over + coefficients
      float d0, d1;
      if constexpr (false && GVL_BWD_LS && FUSED && !L0G) {   // (not here: the kernel sits at its register cap, +80 bytes of scratch)
        ls_dot_pair<false>(slab4, roff - row0, gq, j, d0, d1);               // one sample per lane: every row is in LDS
        GVL_T(1)
        GVL_T(2)
      } else {
      float p0[16], p1[16];
      // L0G: the eight level-0 rows of this pass are requested first and consumed last -- the twelve LDS steps run
      // while they travel
      float4 v0g[4], v1g[4];
      if (L0G) {
#define GVL_OWN_G(SI)                                                         \
  {                                                                           \
    const int rr = row_bcast_i<SI>(roff);                                     \
    v0g[SI] = ld4(value, vg + (int64_t)rr * (M * 16));                        \
    v1g[SI] = ld4(value, vg + (int64_t)min(rr + 1, S - 1) * (M * 16));        \
  }
        GVL_OWN_G(0) GVL_OWN_G(1) GVL_OWN_G(2) GVL_OWN_G(3)
#undef GVL_OWN_G
      }
      // the LDS steps in groups: the row reads of a group are requested together, then consumed (written step by step the
      // compiler keeps one or two reads in flight and every sample pays most of an LDS round trip)
#define GVL_OWN_PAIR(SI)                                                      \
  {                                                                           \
    const int ra = row_bcast_i<SI>(roff) - row0, rb = row_bcast_i<SI + 1>(roff) - row0; \
    const float4 va0 = slab4[ra * 16 + j], va1 = slab4[ra * 16 + 16 + j];     \
    const float4 vb0 = slab4[rb * 16 + j], vb1 = slab4[rb * 16 + 16 + j];     \
    p0[SI] = dot4(gq, va0);                                                   \
    p1[SI] = dot4(gq, va1);                                                   \
    p0[SI + 1] = dot4(gq, vb0);                                               \
    p1[SI + 1] = dot4(gq, vb1);                                               \
  }
#define GVL_OWN_QUAD(SI)                                                      \
  {                                                                           \
    const int ra = row_bcast_i<SI>(roff) - row0, rb = row_bcast_i<SI + 1>(roff) - row0; \
    const int rc = row_bcast_i<SI + 2>(roff) - row0, rd = row_bcast_i<SI + 3>(roff) - row0; \
    const float4 va0 = slab4[ra * 16 + j], va1 = slab4[ra * 16 + 16 + j];     \
    const float4 vb0 = slab4[rb * 16 + j], vb1 = slab4[rb * 16 + 16 + j];     \
    const float4 vc0 = slab4[rc * 16 + j], vc1 = slab4[rc * 16 + 16 + j];     \
    const float4 vd0 = slab4[rd * 16 + j], vd1 = slab4[rd * 16 + 16 + j];     \
    p0[SI] = dot4(gq, va0);                                                   \
    p1[SI] = dot4(gq, va1);                                                   \
    p0[SI + 1] = dot4(gq, vb0);                                               \
    p1[SI + 1] = dot4(gq, vb1);                                               \
    p0[SI + 2] = dot4(gq, vc0);                                               \
    p1[SI + 2] = dot4(gq, vc1);                                               \
    p0[SI + 3] = dot4(gq, vd0);                                               \
    p1[SI + 3] = dot4(gq, vd1);                                               \
  }
      if (L0G) {                                                              // (32 registers hold the level-0 rows in flight)
        GVL_OWN_PAIR(4) GVL_OWN_PAIR(6) GVL_OWN_PAIR(8) GVL_OWN_PAIR(10) GVL_OWN_PAIR(12) GVL_OWN_PAIR(14)
      } else {
        GVL_OWN_QUAD(0) GVL_OWN_QUAD(4) GVL_OWN_QUAD(8) GVL_OWN_QUAD(12)
      }
#undef GVL_OWN_PAIR
#undef GVL_OWN_QUAD
      if (L0G) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          p0[k] = dot4(gq, v0g[k]);
          p1[k] = dot4(gq, v1g[k]);
        }
      }
      GVL_T(1)                                                               // 16 sample steps: 32 row reads + 32 dot products
      d0 = row_reduce_scatter16(p0, j);
      d1 = row_reduce_scatter16(p1, j);
      GVL_T(2)                                                               // two reduce-scatters
      }
      const float keep_w = fmaf(clo, d0, chi * d1);
      const float keep_x = fmaf(dxlo, d0, dxhi * d1);
      const float keep_y = fmaf(dylo, d0, dyhi * d1);
      GVL_TOUCH_PREFETCH(r_n, g_n)
      if (!FUSED) {
        if (act) {
          st_stream(gattn + tb + j, keep_w);                                     // cuh:156-157
          st_stream(reinterpret_cast<float2 *>(gloc) + tb + j, make_float2(keep_x, keep_y));
        }
      } else {
        const float dsum = row_allsum(w * keep_w);                               // softmax backward (ms_deform_attn.py:100-101)
        const float glogit = w * (keep_w - dsum);
        const float goff = keep_x * dloc;
        float gr0 = keep_x, gr1 = keep_x * r.a * (0.5f * invP);
        gr0 += dpp_f<0xB1>(gr0); gr0 += dpp_f<0x4E>(gr0);
        gr1 += dpp_f<0xB1>(gr1); gr1 += dpp_f<0x4E>(gr1);
        if (act) {
          VT *grow = reinterpret_cast<VT *>(gloc) + ((int64_t)b * Q + qq) * (int64_t)(2 * M * LP);
          grow[m * LP + j] = (VT)goff;
          grow[M * LP + m * LP + j] = (VT)glogit;
          if (gattn && (j & 3) == 0) {
            float *gr = gattn + ((((int64_t)b * Q + qq) * M + m) * L + lvl) * RD;
            gr[0] = gr0;
            if (RD == 2) gr[1] = gr1;
          }
        }
      }
      GVL_T(3)                                                               // epilogue + stores
    }
#ifdef GVL_PHASE_TIMING
    if (stamps && threadIdx.x == 0)
      for (int k_ = 0; k_ < 4; ++k_) stamps[(2048 + blockIdx.x) * 4 + k_] = tacc[k_];
#endif
#undef GVL_T
  }
  __syncthreads();                                                       // the slab is dead from here on
  if (stamps && threadIdx.x == 0) stamps[blockIdx.x * 4 + 2] = wall_clock64();

  // ================================ phase B =====================================================================
  float4 *G4 = slab4;                                                    // [qc] grad_out rows of the chunk
  int *cnt = reinterpret_cast<int *>(G4 + (size_t)qc * 16);              // [nOwn + 2] histogram over owned rows
  int *off = cnt + (nOwn + 2);                                           // [nOwn + 2] exclusive prefix
  int *ent_rp = off + (nOwn + 2);                                        // [qc * 8] owned row | slot << 12, or -1
  float *ent_lo = reinterpret_cast<float *>(ent_rp + qc * 8);
  float *ent_hi = ent_lo + qc * 8;
  // the entries in row order, as the gather wants them: (LDS byte offset of the query's grad_out row, w c_lo, w c_hi) --
  // one LDS round trip per list element instead of the id -> entry indirection
  int *srt_q = reinterpret_cast<int *>(ent_hi + qc * 8);
  float *srt_lo = reinterpret_cast<float *>(srt_q + qc * 8);
  float *srt_hi = srt_lo + qc * 8;
  __shared__ int wave_tot[kBwdThreads / 64];
  __shared__ int carry_s;
  f2v acc01[kOwnNU], acc23[kOwnNU];
#pragma unroll
  for (int i = 0; i < kOwnNU; ++i) acc01[i] = acc23[i] = (f2v){0.f, 0.f};
  const char *G_b = reinterpret_cast<const char *>(G4);
  const int lane_off = j * 16;

  for (int c0 = (dbg & 2) ? Q : 0; c0 < Q; c0 += qc) {
    const int c1 = min(Q, c0 + qc);
    const int nq = c1 - c0;
    // operands of the first pass are requested before the barrier that frees the LDS
    int qb = c0 + wave * 4;
    RawOps r_n = {0.f, 0.5f, 0.f, 0.f};
    float4 g_n = make_float4(0.f, 0.f, 0.f, 0.f);
    if (qb < c1) {
      const int64_t bqn = (int64_t)b * Q + min(qb + tq, c1 - 1);
      r_n = fetch_ops<FUSED, VT>(loc, attn, bqn, m, M, LP, L, RD, j, lvl);
      if (qb + tq < c1) g_n = ld4(gout, (bqn * M + m) * 16 + j);
    }
    for (int i = threadIdx.x; i < nOwn + 2; i += blockDim.x) cnt[i] = 0;
    __syncthreads();
    if (stamps && threadIdx.x == 0) stamps[(1024 + blockIdx.x) * 4 + 0] = wall_clock64();   // (the LAST chunk's phases)
    // ---- coefficient pass: entries of the owned levels + the chunk's grad_out rows ---------------------------
    for (; qb < c1; qb += nw * 4) {
      const int q = qb + tq;
      const bool act = q < c1;
      const RawOps r = r_n;
      const float4 gq = g_n;
      const int qbn = qb + nw * 4;
      if (qbn < c1) {
        const int64_t bqn = (int64_t)b * Q + min(qbn + tq, c1 - 1);
        r_n = fetch_ops<FUSED, VT>(loc, attn, bqn, m, M, LP, L, RD, j, lvl);
        g_n = (qbn + tq < c1) ? ld4(gout, (bqn * M + m) * 16 + j) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
      float2 xy;
      float w, dloc;
      resolve_ops<FUSED>(r, invT, invP, RD, xy.x, xy.y, w, dloc);        // (FUSED: the softmax needs all 16 lanes of the row)
      const Coef1D c = coef_1d<PAD>(xy.x, xy.y, Tl);
      const float elo = c.c_lo * c.wy * w, ehi = c.c_hi * c.wy * w;
      if (act) {
        G4[(q - c0) * 16 + j] = gq;
        if (mine) {
          const int e = (q - c0) * 8 + k_own;
          int rp = -1;
          if (elo != 0.f || ehi != 0.f) {
            const int o = o_base + c.r;
            rp = o | (atomicAdd(&cnt[o], 1) << 12);
            ent_lo[e] = elo;
            ent_hi[e] = ehi;
          }
          ent_rp[e] = rp;
        }
      }
    }
    __syncthreads();
    if (stamps && threadIdx.x == 0) stamps[(1024 + blockIdx.x) * 4 + 1] = wall_clock64();
    // ---- exclusive scan of the histogram, entry ids into row order ---------------------------------------------
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (int base = 0; base < nOwn + 1; base += blockDim.x) {
      const int i = base + threadIdx.x;
      const int v = (i < nOwn + 1) ? cnt[i] : 0;
      int incl = v;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const int t_ = __shfl_up(incl, o, 64);
        if (lane >= o) incl += t_;
      }
      if (lane == 63) wave_tot[wave] = incl;
      __syncthreads();
      int pre_ = carry_s;
      for (int k = 0; k < wave; ++k) pre_ += wave_tot[k];
      if (i < nOwn + 1) off[i] = pre_ + incl - v;
      __syncthreads();
      if (threadIdx.x == blockDim.x - 1) carry_s = pre_ + incl;
      __syncthreads();
    }
    if (threadIdx.x == 0) off[nOwn + 1] = carry_s;
    for (int e = threadIdx.x; e < nq * 8; e += blockDim.x) {
      const int rp = ent_rp[e];
      if (rp >= 0) {
        const int pos = off[rp & 4095] + (rp >> 12);
        srt_q[pos] = (e >> 3) << 8;
        srt_lo[pos] = ent_lo[e];
        srt_hi[pos] = ent_hi[e];
      }
    }
    __syncthreads();
    if (stamps && threadIdx.x == 0) stamps[(1024 + blockIdx.x) * 4 + 2] = wall_clock64();
    // ---- gather: this wavefront's units, accumulators in registers ---------------------------------------------
    // Units are static, so the chain  list bounds -> list elements -> grad_out rows  is software-pipelined ACROSS units:
    // at the top of unit i the bounds of unit i + 2 are requested and, from the bounds requested one unit earlier, the
    // first batch of unit i + 1; unit i's batches then run on operands that are already in registers.
    struct Raw { int lo, mid, hi; };                                       // off[o - 1], off[o], off[o + 1]
    struct Unit { int a1, n1, a0, n_own, n; };
    // (all three loaders are branch-free -- clamped addresses, selects -- so that the compiler's wait counters stay exact
    // across them: behind a conditional block it waits for EVERY outstanding LDS read, prefetches included)
    auto raw_of = [&](int i) {
      Raw r = {0, 0, 0};
      if (i < kOwnNU) {                                                    // (compile-time after unrolling)
        const int o = (wave + nw * i) * 4 + tq;
        const bool live = o < nOwn;
        const int oc = live ? o : 0;
        const int lo = off[oc > 0 ? oc - 1 : 0], mid = off[oc], hi = off[oc + 1];
        r.mid = mid;
        r.hi = live ? hi : mid;
        r.lo = (live && oc > 0) ? lo : mid;
      }
      return r;
    };
    auto unit_of = [&](const Raw &r) {
      Unit u;
      u.a1 = r.mid; u.n1 = r.hi - r.mid; u.a0 = r.lo;
      u.n_own = r.hi - r.lo;                                               // entries with row == o (c_lo), then row == o - 1 (c_hi)
      // longest list of the four rows: they run in lockstep (wave-uniform trip count, in a scalar register)
      u.n = max(max(__builtin_amdgcn_readlane(u.n_own, 0), __builtin_amdgcn_readlane(u.n_own, 16)),
                max(__builtin_amdgcn_readlane(u.n_own, 32), __builtin_amdgcn_readlane(u.n_own, 48)));
      return u;
    };
    auto fetch_tc = [&](int i, const Unit &u) {
      const bool valid = i < u.n_own, first = i < u.n1;
      int pos = first ? u.a1 + i : u.a0 + (i - u.n1);
      pos = valid ? pos : 0;
      const float *cs = first ? srt_lo : srt_hi;
      const int qo = srt_q[pos];
      const float cf = cs[pos];
      return (f2v){__builtin_bit_cast(float, valid ? qo : 0), valid ? cf : 0.f};
    };
#define GVL_GATHER_BATCH(LEFT)                                                                         \
  gather4<0>(tc, G_b, lane_off, a01, a23);                                                             \
  if ((LEFT) > 4) gather4<4>(tc, G_b, lane_off, a01, a23);                                             \
  if ((LEFT) > 8) {                                                                                    \
    gather4<8>(tc, G_b, lane_off, a01, a23);                                                           \
    gather4<12>(tc, G_b, lane_off, a01, a23);                                                          \
  }
    Raw raw_n = raw_of(1);
    Unit un = unit_of(raw_of(0));
    f2v tcn = fetch_tc(j, un);
#pragma unroll
    for (int i = 0; i < kOwnNU; ++i) {
      const Unit u = un;
      f2v tc = tcn;
      const Raw raw_nn = raw_of(i + 2);
      un = unit_of(raw_n);
      raw_n = raw_nn;
      tcn = fetch_tc(j, un);
      if (u.n > 0) {
        f2v a01 = acc01[i], a23 = acc23[i];
        for (int base = 0; base < u.n; base += 16) {
          const f2v tc_next = fetch_tc(base + 16 + j, u);
          GVL_GATHER_BATCH(u.n - base)
          tc = tc_next;
        }
        acc01[i] = a01;
        acc23[i] = a23;
      }
    }
#undef GVL_GATHER_BATCH
    __syncthreads();                                                      // the next chunk rewrites the LDS
  }
  if (stamps && threadIdx.x == 0) stamps[blockIdx.x * 4 + 3] = wall_clock64();
  // ---- every owned row leaves once, final, in the storage type -----------------------------------------------------
#pragma unroll
  for (int i = 0; i < kOwnNU; ++i) {
    const int o = (wave + nw * i) * 4 + tq;
    if (o < nOwn) {
      const int s = o < nF ? sF + o : sC + (o - nF);
      st4_stream(gvalue, ((int64_t)b * S + s) * M * 16 + (int64_t)m * 16 + j,
                 make_float4(acc01[i].x, acc01[i].y, acc23[i].x, acc23[i].y));
    }
  }
}

// sum `n` fp32 partial slabs (each `count4` float4 long) into dst (storage type VT)
template <typename VT>
__global__ void __launch_bounds__(256) k_sum_partials(const float4 *__restrict__ part, int n, int64_t count4,
                                                      VT *__restrict__ dst) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count4; i += (int64_t)gridDim.x * blockDim.x) {
    float4 a = part[i];
    for (int k = 1; k < n; ++k) {
      const float4 v = part[(int64_t)k * count4 + i];
      a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
    }
    st4_stream(dst, i, a);
  }
}

// ------------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------------
constexpr size_t kLdsMax = 160 * 1024;

// How the (b,m) value slab is held on chip: all S rows in LDS, or (long videos) level 0 left in global memory
struct SlabPlan {
  bool ok, l0g;
  int rowsV;   // rows staged in LDS including the zero pad row
};
SlabPlan slab_plan(int S, int L, int P, const int64_t *shapes_host) {
  if ((size_t)(S + 1) * 64 * sizeof(float) <= kLdsMax) return {true, false, S + 1};
  if (shapes_host && L >= 2 && L * P == 16 && P == 4) {
    const int T0 = (int)shapes_host[1];
    if ((size_t)(S - T0 + 1) * 64 * sizeof(float) <= kLdsMax) return {true, true, S - T0 + 1};
  }
  return {false, false, 0};
}


int check_dims(int B, int S, int M, int D, int L, int Q, int P, int pad) {
  if (B < 0 || S < 0 || M <= 0 || D <= 0 || L <= 0 || Q < 0 || P <= 0)
    return fail(GVL_EINVAL, "gvl_msda: bad dims B=%d S=%d M=%d D=%d L=%d Q=%d P=%d", B, S, M, D, L, Q, P);
  if (pad != kPadZeros && pad != kPadBorder) return fail(GVL_EINVAL, "gvl_msda: bad pad_mode %d", pad);
  return 0;
}

bool temporal_host(const int64_t *shapes_host, const int64_t *lsi_host, int L, int S) {
  if (!shapes_host || !lsi_host) return false;
  int64_t run = 0;
  for (int l = 0; l < L; ++l) {
    if (shapes_host[2 * l] != 1 || shapes_host[2 * l + 1] < 1 || lsi_host[l] != run) return false;
    run += shapes_host[2 * l + 1];
  }
  return run == S;
}

int pick_chunks(const char *env, int BM, int Q, int target_wgs) {
  int n = env_int(env, 0);
  if (n <= 0) n = (target_wgs + BM - 1) / BM;
  const int maxn = (Q + 15) / 16;   // keep >= 16 queries per workgroup
  if (n > maxn) n = maxn;
  if (n < 1) n = 1;
  return n;
}

// ---- launchers of the t1d_d64 kernels, shared by the fp32 / bf16 and the plain / fused entry points ------------------
template <typename VT, bool FUSED>
int run_fwd_t1d(const VT *value, const int64_t *shapes, const int64_t *lsi, const void *p0, const float *p1, int B,
                int S, int M, int L, int Q, int P, int RD, int pad, const SlabPlan &plan, VT *out, hipStream_t st,
                float *amax_out = nullptr, bool shared_proj = false) {
  // one 1024-thread workgroup per CU (measured best on MI355X: the 47 KB slab is staged once per CU and 16
  // wavefronts hide the LDS latency); GVL_MSDA_FWD_{THREADS,CHUNKS} override for tuning sweeps
  const int threads = env_int("GVL_MSDA_FWD_THREADS", 1024);
  const int nchunk = pick_chunks("GVL_MSDA_FWD_CHUNKS", B * M, Q, 256);
  const size_t lds = (size_t)plan.rowsV * 64 * sizeof(float);
  const bool full = L * P == 16;
  decltype(&k_fwd_t1d_d64<kPadZeros, true, FUSED, false, VT>) kern;
  if (pad == kPadZeros)
    kern = plan.l0g ? k_fwd_t1d_d64<kPadZeros, true, FUSED, true, VT>
                    : (full || FUSED) ? k_fwd_t1d_d64<kPadZeros, true, FUSED, false, VT>
                                      : k_fwd_t1d_d64<kPadZeros, false, false, false, VT>;
  else
    kern = plan.l0g ? k_fwd_t1d_d64<kPadBorder, true, FUSED, true, VT>
                    : (full || FUSED) ? k_fwd_t1d_d64<kPadBorder, true, FUSED, false, VT>
                                      : k_fwd_t1d_d64<kPadBorder, false, false, false, VT>;
  if constexpr (FUSED && std::is_same<VT, float>::value) {
    if (amax_out)                                                     // (fused fp32 form only: what the inference layers use)
      kern = pad == kPadZeros ? (plan.l0g ? k_fwd_t1d_d64<kPadZeros, true, true, true, VT, true>
                                          : k_fwd_t1d_d64<kPadZeros, true, true, false, VT, true>)
                              : (plan.l0g ? k_fwd_t1d_d64<kPadBorder, true, true, true, VT, true>
                                          : k_fwd_t1d_d64<kPadBorder, true, true, false, VT, true>);
  } else if (amax_out) {
    return fail(GVL_EINVAL, "gvl_msda: row maxima are produced by the fused fp32 forward only");
  }
  if (int rc = ensure_lds(kern, lds)) return rc;
  g_last_impl = FUSED ? 3 : 2;
  g_last_kernel = "k_fwd_t1d_d64";
  return gvl::launch(GVL_PROF_FWD_T1D, Q, B, FUSED ? "k_fwd_t1d_d64<fused>" : "k_fwd_t1d_d64", kern,
                     dim3(B * M, nchunk), dim3(threads), lds, st, value, shapes, lsi, p0, p1, B, S, M, L, Q, P, RD,
                     env_int("GVL_MSDA_XCD_PAIRS", 1) ? nchunk : -nchunk, out, g_fwd_stamps,
                     reinterpret_cast<unsigned *>(amax_out), (Q + nchunk - 1) / nchunk,
                     (M & (M - 1)) == 0 ? __builtin_ctz((unsigned)M) : -1, 1.f / (float)P, shared_proj ? 0 : Q);
}

// number of query chunks per (b,m) slab for the backward: enough workgroups to cover the chip, and few enough
// queries per workgroup for the LDS carve-up; 0 = does not fit
// workgroups per (b,m) slab: enough to cover the 256 CUs, never more than there are chunks
int bwd_groups(int B, int M, int nchunk) {
  const int g = (256 + B * M - 1) / (B * M);
  return g < nchunk ? g : nchunk;
}
int bwd_chunks(int B, int M, int Q, int S, int rowsV) {
  int n = pick_chunks("GVL_MSDA_BWD_CHUNKS", B * M, Q, 256);
  while (n <= Q && bwd_lds_bytes(S, (Q + n - 1) / n, rowsV) > kLdsMax) ++n;
  if (n > Q) return 0;
  const int g = bwd_groups(B, M, n);                   // equal shares: a multiple of the group count
  const int up = (n + g - 1) / g * g;
  return up <= Q ? up : n;
}

// level-split form (k_bwd_t1d_split): two workgroups per slab, one chunk each, L = P = 4, everything in LDS
bool bwd_split_ok(int B, int S, int M, int L, int P, int Q, const SlabPlan &plan, int nchunk) {
  if (!env_int("GVL_MSDA_BWD_SPLIT", 1)) return false;
  if (L != 4 || P != 4 || plan.l0g || !plan.ok || nchunk != 2 || bwd_groups(B, M, nchunk) != 2 || Q < 2) return false;
  if (S + 2 > 4095) return false;                                       // row packed into 12 bits beside the slot
  return bwd_split_lds_bytes(S, (Q + 1) / 2) <= kLdsMax;
}

// row-ownership form (k_bwd_t1d_own): two workgroups per slab, L = P = 4, any number of query chunks, level 0 in LDS or
// in global memory; -> queries per phase-B chunk, 0 = not eligible
int bwd_own_chunk(int B, int S, int M, int L, int P, int Q, const SlabPlan &plan, const int64_t *shapes_host) {
  if (!env_int("GVL_MSDA_BWD_OWN", 1)) return 0;
  if (L != 4 || P != 4 || !plan.ok || !shapes_host || Q < 2 || B * M < env_int("GVL_MSDA_BWD_OWN_MINBM", 64) ||
      (B * M >= 256 && (B * M) % 8))
    return 0;
  // WHERE the pair form pays (round 4, T = 100 ... 512, B = 4 ... 64, fused entry points, back to back): it removes the
  // chunked form's read-modify-write of a partial slab per query chunk and k_sum_partials, at the price of the coefficient
  // arithmetic done in both workgroups of a slab -- a win for LONG slabs (level 0 in global memory: B = 16, T = 512: 128 -> 77
  // us encoder / 50 -> 32 us decoder shape; B = 12: 137 -> 81 / 60 -> 35; B = 8: 73 vs 77 / 36 -> 33) and wherever a workgroup
  // of the chunked form would walk two or more chunks; a loss for short slabs that fit one carve-up (B = 8, T = 100: 21 vs 35
  // us; B = 24, T = 200: 53 vs 64) -- those stay on the chunked / level-split forms.  B*M >= 256 (one workgroup per slab
  // already covers the chip, 2 B*M workgroups run in rounds): from four chunks on (B = 32, T = 512: 960 queries 230 -> 174 us,
  // 300 queries 70 vs 76).
  {
    const int n_old = bwd_chunks(B, M, Q, S, plan.rowsV);
    if (n_old <= 0) return 0;
    const int g_old = bwd_groups(B, M, n_old);
    const bool pays = B * M >= 256 ? n_old >= 4 : (plan.l0g || (n_old + g_old - 1) / g_old >= 2);
    if (!pays && !env_int("GVL_MSDA_BWD_OWN_ALWAYS", 0)) return 0;
  }
  const int n0 = (int)(shapes_host[1] + shapes_host[7]), n1 = (int)(shapes_host[3] + shapes_host[5]);   // {0,3} | {1,2}
  const int n_own = n0 > n1 ? n0 : n1;
  if (n_own > 64 * kOwnNU) return 0;
  const size_t budget = kLdsMax - 512;                                  // the kernel's static LDS (scan scratch) sits beside
  if ((size_t)plan.rowsV * 256 > budget) return 0;
  const size_t fixed = (size_t)(n_own + 2) * 2 * sizeof(int);
  const int qc_max = (int)((budget - fixed) / (256 + 8 * 24));          // grad_out row + 8 entries x (record + sorted record)
  if (qc_max < 64) return 0;
  const int forced = env_int("GVL_MSDA_BWD_OWN_QC", 0);                   // tuning sweeps
  if (forced > 0) return forced <= qc_max ? forced : qc_max;
  const int nchunk = (Q + qc_max - 1) / qc_max;
  int qc = ((Q + nchunk - 1) / nchunk + 63) / 64 * 64;                    // whole rounds of 16 wavefronts x 4 queries
  if (qc > qc_max) qc = qc_max;
  return qc;
}

// fp32 workspace the t1d_d64 backward needs: one partial slab per workgroup when a (b,m) slab is shared by several
// workgroups, and always one fp32 slab set for bf16 storage (the gather accumulates and writes fp32; k_sum_partials
// rounds once)
size_t bwd_workspace_bytes(int B, int S, int M, int nchunk, bool bf16, bool split = false) {
  if (split) return 0;                                                  // final rows written directly
  const int ngroup = bwd_groups(B, M, nchunk);
  return (ngroup > 1 || bf16) ? (size_t)ngroup * B * S * M * 64 * sizeof(float) : 0;
}

template <typename VT, bool FUSED>
int run_bwd_t1d(const VT *value, const int64_t *shapes, const int64_t *lsi, const void *p0, const float *p1,
                const VT *gout, int B, int S, int M, int L, int Q, int P, int RD, int pad, const SlabPlan &plan,
                int nchunk, const int64_t *shapes_host, VT *gvalue, void *g0, float *g1, void *ws, size_t ws_bytes,
                hipStream_t st) {
  constexpr bool kBf16 = !std::is_same<VT, float>::value;
  const int qper = (Q + nchunk - 1) / nchunk;
  if (bwd_split_ok(B, S, M, L, P, Q, plan, nchunk)) {
    auto kern = pad == kPadZeros ? k_bwd_t1d_split<kPadZeros, FUSED, VT> : k_bwd_t1d_split<kPadBorder, FUSED, VT>;
    const size_t lds = bwd_split_lds_bytes(S, qper);
    if (int rc = ensure_lds(kern, lds)) return rc;
    g_last_impl = FUSED ? 3 : 2;
    g_last_kernel = "k_bwd_t1d_split";
    return gvl::launch(GVL_PROF_BWD_T1D, Q, B, FUSED ? "k_bwd_t1d_split<fused>" : "k_bwd_t1d_split", kern,
                       dim3(2 * B * M), dim3(kBwdThreads), lds, st, value, shapes, lsi, p0, p1, gout, B, S, M, Q, RD, qper,
                       gvalue, g0, g1, g_bwd_stamps);
  }
  if (const int qc = bwd_own_chunk(B, S, M, L, P, Q, plan, shapes_host)) {
    decltype(&k_bwd_t1d_own<kPadZeros, FUSED, false, VT>) kern;
    if (pad == kPadZeros)
      kern = plan.l0g ? k_bwd_t1d_own<kPadZeros, FUSED, true, VT> : k_bwd_t1d_own<kPadZeros, FUSED, false, VT>;
    else
      kern = plan.l0g ? k_bwd_t1d_own<kPadBorder, FUSED, true, VT> : k_bwd_t1d_own<kPadBorder, FUSED, false, VT>;
    const int n0 = (int)(shapes_host[1] + shapes_host[7]), n1 = (int)(shapes_host[3] + shapes_host[5]);
    const size_t lds = bwd_own_lds_bytes(plan.rowsV, n0 > n1 ? n0 : n1, qc);
    if (int rc = ensure_lds(kern, lds)) return rc;
    g_last_impl = FUSED ? 3 : 2;
    g_last_kernel = "k_bwd_t1d_own";
    return gvl::launch(GVL_PROF_BWD_T1D, Q, B, FUSED ? "k_bwd_t1d_own<fused>" : "k_bwd_t1d_own", kern, dim3(2 * B * M),
                       dim3(kBwdThreads), lds, st, value, shapes, lsi, p0, p1, gout, B, S, M, Q, RD, qc, gvalue, g0, g1,
                       g_bwd_stamps, env_int("GVL_MSDA_OWN_DEBUG", 0) | ((B * M > 128 && (B * M) % 8 == 0) ? 8 : 0));
  }
  const int ngroup = bwd_groups(B, M, nchunk);
  const size_t lds = bwd_lds_bytes(S, qper, plan.rowsV);
  const size_t need = bwd_workspace_bytes(B, S, M, nchunk, kBf16);
  float *part = reinterpret_cast<float *>(gvalue);
  if (need) {
    if (!ws || ws_bytes < need)
      return fail(GVL_ENOSPC, "gvl_msda backward: workspace %zu < required %zu bytes", ws_bytes, need);
    part = (float *)ws;
  }
  const bool full = L * P == 16;
  decltype(&k_bwd_t1d_d64<kPadZeros, true, FUSED, false, false, VT>) kern;
  const bool loop = nchunk > ngroup;
#define GVL_BWD_PICK(PADV, LOOPV)                                                                    \
  (plan.l0g ? k_bwd_t1d_d64<PADV, true, FUSED, true, LOOPV, VT>                                      \
            : (full || FUSED) ? k_bwd_t1d_d64<PADV, true, FUSED, false, LOOPV, VT>                   \
                              : k_bwd_t1d_d64<PADV, false, false, false, LOOPV, VT>)
  if (pad == kPadZeros)
    kern = loop ? GVL_BWD_PICK(kPadZeros, true) : GVL_BWD_PICK(kPadZeros, false);
  else
    kern = loop ? GVL_BWD_PICK(kPadBorder, true) : GVL_BWD_PICK(kPadBorder, false);
#undef GVL_BWD_PICK
  if (int rc = ensure_lds(kern, lds)) return rc;
  if (int rc = gvl::launch(GVL_PROF_BWD_T1D, Q, B, FUSED ? "k_bwd_t1d_d64<fused>" : "k_bwd_t1d_d64", kern,
                           dim3(ngroup * B * M), dim3(kBwdThreads), lds, st, value, shapes, lsi, p0, p1, gout, B, S, M,
                           L, Q, P, RD, nchunk, ngroup, qper, part, g0, g1, g_bwd_stamps))
    return rc;
  if (need) {
    const int64_t count4 = (int64_t)B * S * M * 16;
    int64_t blocks = (count4 + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    if (int rc = gvl::launch(GVL_PROF_SUM_PARTIALS, Q, B, "k_sum_partials", k_sum_partials<VT>, dim3((unsigned)blocks),
                             dim3(256), 0, st, (const float4 *)part, ngroup, count4, gvalue))
      return rc;
  }
  g_last_impl = FUSED ? 3 : 2;
  g_last_kernel = loop ? "k_bwd_t1d_d64<loop>" : "k_bwd_t1d_d64";
  return 0;
}

template <typename T>
int forward_impl(const T *value, const int64_t *shapes, const int64_t *lsi, const T *loc, const T *attn, int B, int S,
                 int M, int D, int L, int Q, int P, int pad, const int64_t *shapes_host, const int64_t *lsi_host,
                 T *out, hipStream_t st) {
  if (int rc = check_dims(B, S, M, D, L, Q, P, pad)) return rc;
  const int64_t n = (int64_t)B * Q * M * D;
  if (n == 0) return 0;                                  // empty query set / batch: nothing to write
  if (!value || !shapes || !lsi || !loc || !attn || !out) return fail(GVL_EINVAL, "gvl_msda_forward: null pointer");
  const int mode = impl_mode();
  const bool temporal = S > 0 && temporal_host(shapes_host, lsi_host, L, S);
  const SlabPlan plan = temporal ? slab_plan(S, L, P, shapes_host) : SlabPlan{false, false, 0};
  const bool fast_ok = sizeof(T) == 4 && D == 64 && L * P <= 16 && temporal && plan.ok;
  if (mode == 2 && !fast_ok) return fail(GVL_EINVAL, "gvl_msda_forward: fast kernels not eligible for this call");
  if (fast_ok && mode != 1) {
    if constexpr (sizeof(T) == 4)
      return run_fwd_t1d<float, false>((const float *)value, shapes, lsi, loc, (const float *)attn, B, S, M, L, Q, P, 0,
                                       pad, plan, (float *)out, st);
  }
  int64_t blocks = (n + 255) / 256;
  if (blocks > 256 * 32) blocks = 256 * 32;
  g_last_impl = 1;
  g_last_kernel = "k_fwd_generic";
  return gvl::launch(GVL_PROF_FWD_GENERIC, Q, B, "k_fwd_generic", k_fwd_generic<T, false>, dim3((unsigned)blocks),
                     dim3(256), 0, st, value, shapes, lsi, loc, attn, B, S, M, D, L, Q, P, pad, out);
}

template <typename T>
int sample_impl(const T *value, const int64_t *shapes, const int64_t *lsi, const T *loc, int B, int S, int M, int D,
                int L, int Q, int P, int pad, T *sample, hipStream_t st) {
  if (int rc = check_dims(B, S, M, D, L, Q, P, pad)) return rc;
  const int64_t n = (int64_t)B * Q * M * D;
  if (n == 0) return 0;
  if (!value || !shapes || !lsi || !loc || !sample) return fail(GVL_EINVAL, "gvl_msda_sample: null pointer");
  int64_t blocks = (n + 255) / 256;
  if (blocks > 256 * 32) blocks = 256 * 32;
  g_last_impl = 1;
  return gvl::launch(GVL_PROF_SAMPLE, Q, B, "k_fwd_generic<sample>", k_fwd_generic<T, true>, dim3((unsigned)blocks),
                     dim3(256), 0, st, value, shapes, lsi, loc, (const T *)nullptr, B, S, M, D, L, Q, P, pad, sample);
}

template <typename T>
int backward_impl(const T *value, const int64_t *shapes, const int64_t *lsi, const T *loc, const T *attn,
                  const T *gout, int B, int S, int M, int D, int L, int Q, int P, int pad, const int64_t *shapes_host,
                  const int64_t *lsi_host, T *gvalue, T *gloc, T *gattn, void *ws, size_t ws_bytes, hipStream_t st) {
  if (int rc = check_dims(B, S, M, D, L, Q, P, pad)) return rc;
  const size_t gv_bytes = (size_t)B * S * M * D * sizeof(T);
  const int64_t ntup = (int64_t)B * Q * M;
  if (gv_bytes && !gvalue) return fail(GVL_EINVAL, "gvl_msda_backward: null pointer");
  if (ntup == 0) {
    if (gv_bytes) {
      if (int rc = gvl::zero_fill(gvalue, gv_bytes, st)) return rc;
    }
    return 0;
  }
  if (!value || !shapes || !lsi || !loc || !attn || !gout || !gvalue || !gloc || !gattn)
    return fail(GVL_EINVAL, "gvl_msda_backward: null pointer");
  const int mode = impl_mode();
  const bool temporal = S > 0 && temporal_host(shapes_host, lsi_host, L, S);
  const SlabPlan plan = temporal ? slab_plan(S, L, P, shapes_host) : SlabPlan{false, false, 0};
  const int nchunk_f = plan.ok ? bwd_chunks(B, M, Q, S, plan.rowsV) : 0;
  const int qper_f = nchunk_f > 0 ? (Q + nchunk_f - 1) / nchunk_f : Q;
  const size_t lds = nchunk_f > 0 ? bwd_lds_bytes(S, qper_f, plan.rowsV) : kLdsMax + 1;
  const bool fast_ok = sizeof(T) == 4 && D == 64 && L * P <= 16 && lds <= kLdsMax && temporal && plan.ok;
  if (mode == 2 && !fast_ok) return fail(GVL_EINVAL, "gvl_msda_backward: fast kernels not eligible for this call");
  if (fast_ok && mode != 1) {
    if constexpr (sizeof(T) == 4)
      return run_bwd_t1d<float, false>((const float *)value, shapes, lsi, loc, (const float *)attn, (const float *)gout,
                                       B, S, M, L, Q, P, 0, pad, plan, nchunk_f, shapes_host, (float *)gvalue, gloc,
                                       (float *)gattn, ws, ws_bytes, st);
  }
  if (int rc = gvl::zero_fill(gvalue, gv_bytes, st)) return rc;
  int64_t blocks = (ntup + 3) / 4;
  if (blocks > 256 * 64) blocks = 256 * 64;
  g_last_impl = 1;
  g_last_kernel = "k_bwd_generic";
  return gvl::launch(GVL_PROF_BWD_GENERIC, Q, B, "k_bwd_generic", k_bwd_generic<T>, dim3((unsigned)blocks), dim3(256),
                     0, st, value, shapes, lsi, loc, attn, gout, B, S, M, D, L, Q, P, pad, gvalue, gloc, gattn);
}

// ---- fused module path: MSDeformAttn.forward between the projection GEMM and output_proj (ms_deform_attn.py:99-124)
int fused_eligible(int B, int S, int M, int D, int L, int Q, int P, int RD, int pad, const int64_t *shapes_host,
                          const int64_t *lsi_host) {
  if (int rc = check_dims(B, S, M, D, L, Q, P, pad)) return rc;
  if (D != 64 || L * P != 16 || P != 4 || (RD != 1 && RD != 2) || S <= 0)
    return fail(GVL_EINVAL, "gvl_msda1d_fused: needs D=64, L*P=16, P=4, RD in {1,2} (got D=%d L=%d P=%d RD=%d)", D, L, P,
                RD);
  if (!temporal_host(shapes_host, lsi_host, L, S))
    return fail(GVL_EINVAL, "gvl_msda1d_fused: needs host copies of temporal (H=1) level shapes");
  return 0;
}

template <typename VT>
int fused_forward(const VT *value, const int64_t *shapes, const int64_t *lsi, const VT *proj, const float *ref,
                         int B, int S, int M, int D, int L, int Q, int P, int RD, int pad_mode,
                         const int64_t *shapes_host, const int64_t *lsi_host, VT *out, void *stream,
                         float *amax_out = nullptr, bool shared_proj = false) {
  if (int rc = fused_eligible(B, S, M, D, L, Q, P, RD, pad_mode, shapes_host, lsi_host)) return rc;
  if ((int64_t)B * Q == 0) return 0;
  if (!value || !shapes || !lsi || !proj || !ref || !out) return fail(GVL_EINVAL, "gvl_msda1d_fused_forward: null pointer");
  const SlabPlan plan = slab_plan(S, L, P, shapes_host);
  if (!plan.ok) return fail(GVL_EINVAL, "gvl_msda1d_fused_forward: slab of %d rows does not fit LDS", S);
  return run_fwd_t1d<VT, true>(value, shapes, lsi, proj, ref, B, S, M, L, Q, P, RD, pad_mode, plan, out,
                               (hipStream_t)stream, amax_out, shared_proj);
}

template <typename VT>
int fused_backward(const VT *value, const int64_t *shapes, const int64_t *lsi, const VT *proj, const float *ref,
                          const VT *grad_out, int B, int S, int M, int D, int L, int Q, int P, int RD, int pad_mode,
                          const int64_t *shapes_host, const int64_t *lsi_host, VT *grad_value, VT *grad_proj,
                          float *grad_ref, void *workspace, size_t workspace_bytes, void *stream) {
  if (int rc = fused_eligible(B, S, M, D, L, Q, P, RD, pad_mode, shapes_host, lsi_host)) return rc;
  hipStream_t st = (hipStream_t)stream;
  const size_t gv_bytes = (size_t)B * S * M * D * sizeof(VT);
  if (gv_bytes && !grad_value) return fail(GVL_EINVAL, "gvl_msda1d_fused_backward: null pointer");
  if ((int64_t)B * Q == 0) return gvl::zero_fill(grad_value, gv_bytes, st);
  if (!value || !shapes || !lsi || !proj || !ref || !grad_out || !grad_proj)
    return fail(GVL_EINVAL, "gvl_msda1d_fused_backward: null pointer");
  const SlabPlan plan = slab_plan(S, L, P, shapes_host);
  const int nchunk = plan.ok ? bwd_chunks(B, M, Q, S, plan.rowsV) : 0;
  if (nchunk <= 0) return fail(GVL_EINVAL, "gvl_msda1d_fused_backward: problem does not fit LDS");
  return run_bwd_t1d<VT, true>(value, shapes, lsi, proj, ref, grad_out, B, S, M, L, Q, P, RD, pad_mode, plan, nchunk,
                               shapes_host, grad_value, grad_proj, grad_ref, workspace, workspace_bytes, st);
}

}  // namespace

extern "C" {

int gvl_msda_abi_version(void) { return GVL_MSDA_ABI_VERSION; }
const char *gvl_last_error(void) { return gvl::g_err; }
void gvl_msda_set_impl(int impl) { g_impl = (impl >= 0 && impl <= 2) ? impl : 0; }
int gvl_msda_last_impl(void) { return g_last_impl; }
const char *gvl_msda_last_kernel(void) { return g_last_kernel; }
void gvl_reload_env(void) {
  gvl::env_reload();
  g_impl = -1;
}

void gvl_msda_debug_stamps(void *device_buffer) {
  g_fwd_stamps = (unsigned long long *)device_buffer;
  g_bwd_stamps = g_fwd_stamps ? g_fwd_stamps + 4 * 4096 : nullptr;        // backward: second half of the buffer
}

static __global__ void __launch_bounds__(64) k_clock_probe(long long *out, int n, float seed) {
  float a = seed + threadIdx.x * 1e-3f;
  const float b = 1.0001f;
  const long long r0 = __builtin_amdgcn_s_memrealtime(), t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < n; i += 16) {
#pragma unroll
    for (int u = 0; u < 16; ++u) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a) : "v"(b));
  }
  const long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0) {
    out[0] = t1 - t0;
    out[1] = r1 - r0;
  }
  if (a == 12345.678f) out[1] = 0;                                       // keep the chain alive
}

int gvl_clock_probe(long long *device_out2, int n_fma, void *stream) {
  if (!device_out2 || n_fma <= 0) return fail(GVL_EINVAL, "gvl_clock_probe: null pointer / n_fma <= 0");
  hipLaunchKernelGGL(k_clock_probe, dim3(1), dim3(64), 0, (hipStream_t)stream, device_out2, n_fma, 0.5f);
  hipError_t err = hipGetLastError();
  return err == hipSuccess ? 0 : fail((int)err, "gvl_clock_probe: launch failed: %s", hipGetErrorString(err));
}

int gvl_prof_enable(int on) {
  gvl::Profiler &p = gvl::profiler();
  std::lock_guard<std::mutex> g(p.mu);
  p.level = on < 0 ? 0 : on;
  return 0;
}

int gvl_prof_collect(float *us, int *tag, int *meta_a, int *meta_b, int capacity) {
  gvl::Profiler &p = gvl::profiler();
  std::lock_guard<std::mutex> g(p.mu);
  int n = 0;
  for (auto &e : p.entries) {
    float ms = 0.f;
    if (hipEventSynchronize(e.stop) == hipSuccess && hipEventElapsedTime(&ms, e.start, e.stop) == hipSuccess &&
        n < capacity && us) {
      us[n] = ms * 1000.f;
      if (tag) tag[n] = e.tag;
      if (meta_a) meta_a[n] = e.a;
      if (meta_b) meta_b[n] = e.b;
      ++n;
    }
    (void)hipEventDestroy(e.start);
    (void)hipEventDestroy(e.stop);
  }
  p.entries.clear();
  return n;
}

int gvl_msda_forward_f32(const float *value, const int64_t *shapes, const int64_t *lsi, const float *loc,
                         const float *attn, int B, int S, int M, int D, int L, int Q, int P, int pad_mode,
                         const int64_t *shapes_host, const int64_t *lsi_host, float *out, void *stream) {
  return forward_impl<float>(value, shapes, lsi, loc, attn, B, S, M, D, L, Q, P, pad_mode, shapes_host, lsi_host, out,
                             (hipStream_t)stream);
}
int gvl_msda_forward_f64(const double *value, const int64_t *shapes, const int64_t *lsi, const double *loc,
                         const double *attn, int B, int S, int M, int D, int L, int Q, int P, int pad_mode,
                         const int64_t *shapes_host, const int64_t *lsi_host, double *out, void *stream) {
  return forward_impl<double>(value, shapes, lsi, loc, attn, B, S, M, D, L, Q, P, pad_mode, shapes_host, lsi_host, out,
                              (hipStream_t)stream);
}
int gvl_msda_sample_f32(const float *value, const int64_t *shapes, const int64_t *lsi, const float *loc, int B, int S,
                        int M, int D, int L, int Q, int P, int pad_mode, float *sample, void *stream) {
  return sample_impl<float>(value, shapes, lsi, loc, B, S, M, D, L, Q, P, pad_mode, sample, (hipStream_t)stream);
}
int gvl_msda_sample_f64(const double *value, const int64_t *shapes, const int64_t *lsi, const double *loc, int B,
                        int S, int M, int D, int L, int Q, int P, int pad_mode, double *sample, void *stream) {
  return sample_impl<double>(value, shapes, lsi, loc, B, S, M, D, L, Q, P, pad_mode, sample, (hipStream_t)stream);
}

size_t gvl_msda_backward_workspace_bytes(int B, int S, int M, int D, int L, int Q, int P, int elem_bytes,
                                         const int64_t *shapes_host) {
  // elem_bytes = storage size of value / grad_value: 4 (fp32), 2 (bf16); 8 (fp64) never needs a workspace
  if ((elem_bytes != 4 && elem_bytes != 2) || D != 64 || L * P > 16 || S <= 0) return 0;
  const SlabPlan plan = slab_plan(S, L, P, shapes_host);
  if (!plan.ok) return 0;
  const int n = bwd_chunks(B, M, Q, S, plan.rowsV);
  if (n <= 0) return 0;
  const bool direct = bwd_split_ok(B, S, M, L, P, Q, plan, n) || bwd_own_chunk(B, S, M, L, P, Q, plan, shapes_host) > 0;
  return bwd_workspace_bytes(B, S, M, n, elem_bytes == 2, direct);
}

int gvl_msda_backward_f32(const float *value, const int64_t *shapes, const int64_t *lsi, const float *loc,
                          const float *attn, const float *grad_out, int B, int S, int M, int D, int L, int Q, int P,
                          int pad_mode, const int64_t *shapes_host, const int64_t *lsi_host, float *grad_value,
                          float *grad_loc, float *grad_attn, void *workspace, size_t workspace_bytes, void *stream) {
  return backward_impl<float>(value, shapes, lsi, loc, attn, grad_out, B, S, M, D, L, Q, P, pad_mode, shapes_host,
                              lsi_host, grad_value, grad_loc, grad_attn, workspace, workspace_bytes,
                              (hipStream_t)stream);
}
int gvl_msda_backward_f64(const double *value, const int64_t *shapes, const int64_t *lsi, const double *loc,
                          const double *attn, const double *grad_out, int B, int S, int M, int D, int L, int Q, int P,
                          int pad_mode, const int64_t *shapes_host, const int64_t *lsi_host, double *grad_value,
                          double *grad_loc, double *grad_attn, void *workspace, size_t workspace_bytes, void *stream) {
  return backward_impl<double>(value, shapes, lsi, loc, attn, grad_out, B, S, M, D, L, Q, P, pad_mode, shapes_host,
                               lsi_host, grad_value, grad_loc, grad_attn, workspace, workspace_bytes,
                               (hipStream_t)stream);
}

int gvl_msda1d_fused_forward_f32(const float *value, const int64_t *shapes, const int64_t *lsi, const float *proj,
                                 const float *ref, int B, int S, int M, int D, int L, int Q, int P, int RD,
                                 int pad_mode, const int64_t *shapes_host, const int64_t *lsi_host, float *out,
                                 void *stream) {
  return fused_forward<float>(value, shapes, lsi, proj, ref, B, S, M, D, L, Q, P, RD, pad_mode, shapes_host, lsi_host,
                              out, stream);
}
int gvl_msda1d_fused_forward_amax_f32(const float *value, const int64_t *shapes, const int64_t *lsi, const float *proj,
                                      const float *ref, int B, int S, int M, int D, int L, int Q, int P, int RD,
                                      int pad_mode, const int64_t *shapes_host, const int64_t *lsi_host, float *out,
                                      float *amax_out, void *stream) {
  if (!amax_out) return fail(GVL_EINVAL, "gvl_msda1d_fused_forward_amax_f32: null pointer");
  return fused_forward<float>(value, shapes, lsi, proj, ref, B, S, M, D, L, Q, P, RD, pad_mode, shapes_host, lsi_host,
                              out, stream, amax_out);
}
int gvl_msda1d_fused_forward_shared_amax_f32(const float *value, const int64_t *shapes, const int64_t *lsi, const float *proj_q,
                                             const float *ref, int B, int S, int M, int D, int L, int Q, int P, int RD,
                                             int pad_mode, const int64_t *shapes_host, const int64_t *lsi_host, float *out,
                                             float *amax_out, void *stream) {
  if (!amax_out) return fail(GVL_EINVAL, "gvl_msda1d_fused_forward_shared_amax_f32: null pointer");
  return fused_forward<float>(value, shapes, lsi, proj_q, ref, B, S, M, D, L, Q, P, RD, pad_mode, shapes_host, lsi_host,
                              out, stream, amax_out, true);
}
int gvl_msda1d_fused_forward_bf16(const uint16_t *value, const int64_t *shapes, const int64_t *lsi,
                                  const uint16_t *proj, const float *ref, int B, int S, int M, int D, int L, int Q,
                                  int P, int RD, int pad_mode, const int64_t *shapes_host, const int64_t *lsi_host,
                                  uint16_t *out, void *stream) {
  return fused_forward<bf16_t>((const bf16_t *)value, shapes, lsi, (const bf16_t *)proj, ref, B, S, M, D, L, Q, P, RD,
                               pad_mode, shapes_host, lsi_host, (bf16_t *)out, stream);
}

size_t gvl_msda1d_fused_backward_workspace_bytes(int B, int S, int M, int D, int L, int Q, int P,
                                                 const int64_t *shapes_host) {
  return gvl_msda_backward_workspace_bytes(B, S, M, D, L, Q, P, 4, shapes_host);
}

int gvl_msda1d_fused_backward_f32(const float *value, const int64_t *shapes, const int64_t *lsi, const float *proj,
                                  const float *ref, const float *grad_out, int B, int S, int M, int D, int L, int Q,
                                  int P, int RD, int pad_mode, const int64_t *shapes_host, const int64_t *lsi_host,
                                  float *grad_value, float *grad_proj, float *grad_ref, void *workspace,
                                  size_t workspace_bytes, void *stream) {
  return fused_backward<float>(value, shapes, lsi, proj, ref, grad_out, B, S, M, D, L, Q, P, RD, pad_mode, shapes_host,
                               lsi_host, grad_value, grad_proj, grad_ref, workspace, workspace_bytes, stream);
}
int gvl_msda1d_fused_backward_bf16(const uint16_t *value, const int64_t *shapes, const int64_t *lsi,
                                   const uint16_t *proj, const float *ref, const uint16_t *grad_out, int B, int S,
                                   int M, int D, int L, int Q, int P, int RD, int pad_mode,
                                   const int64_t *shapes_host, const int64_t *lsi_host, uint16_t *grad_value,
                                   uint16_t *grad_proj, float *grad_ref, void *workspace, size_t workspace_bytes,
                                   void *stream) {
  return fused_backward<bf16_t>((const bf16_t *)value, shapes, lsi, (const bf16_t *)proj, ref,
                                (const bf16_t *)grad_out, B, S, M, D, L, Q, P, RD, pad_mode, shapes_host, lsi_host,
                                (bf16_t *)grad_value, (bf16_t *)grad_proj, grad_ref, workspace, workspace_bytes,
                                stream);
}

// ---- bf16 storage twins of the op (fp32 arithmetic; loc / attn and their gradients stay fp32).  Served by the
// t1d_d64 kernels only: other shapes return GVL_EINVAL and the caller widens to the f32 entry points.
static int bf16_plan(const char *what, int B, int S, int M, int D, int L, int Q, int P, int pad,
                     const int64_t *shapes_host, const int64_t *lsi_host, SlabPlan &plan) {
  if (int rc = check_dims(B, S, M, D, L, Q, P, pad)) return rc;
  if (D != 64 || L * P > 16 || S <= 0 || !temporal_host(shapes_host, lsi_host, L, S))
    return fail(GVL_EINVAL, "%s: bf16 storage needs temporal (H=1) levels with host shapes, D=64, L*P<=16", what);
  plan = slab_plan(S, L, P, shapes_host);
  if (!plan.ok) return fail(GVL_EINVAL, "%s: slab of %d rows does not fit LDS", what, S);
  return 0;
}

int gvl_msda_forward_bf16(const uint16_t *value, const int64_t *shapes, const int64_t *lsi, const float *loc,
                          const float *attn, int B, int S, int M, int D, int L, int Q, int P, int pad_mode,
                          const int64_t *shapes_host, const int64_t *lsi_host, uint16_t *out, void *stream) {
  SlabPlan plan;
  if (int rc = bf16_plan("gvl_msda_forward_bf16", B, S, M, D, L, Q, P, pad_mode, shapes_host, lsi_host, plan)) return rc;
  if ((int64_t)B * Q == 0) return 0;
  if (!value || !shapes || !lsi || !loc || !attn || !out) return fail(GVL_EINVAL, "gvl_msda_forward_bf16: null pointer");
  return run_fwd_t1d<bf16_t, false>((const bf16_t *)value, shapes, lsi, loc, attn, B, S, M, L, Q, P, 0, pad_mode, plan,
                                    (bf16_t *)out, (hipStream_t)stream);
}

int gvl_msda_backward_bf16(const uint16_t *value, const int64_t *shapes, const int64_t *lsi, const float *loc,
                           const float *attn, const uint16_t *grad_out, int B, int S, int M, int D, int L, int Q, int P,
                           int pad_mode, const int64_t *shapes_host, const int64_t *lsi_host, uint16_t *grad_value,
                           float *grad_loc, float *grad_attn, void *workspace, size_t workspace_bytes, void *stream) {
  SlabPlan plan;
  if (int rc = bf16_plan("gvl_msda_backward_bf16", B, S, M, D, L, Q, P, pad_mode, shapes_host, lsi_host, plan)) return rc;
  hipStream_t st = (hipStream_t)stream;
  const size_t gv_bytes = (size_t)B * S * M * D * sizeof(uint16_t);
  if (gv_bytes && !grad_value) return fail(GVL_EINVAL, "gvl_msda_backward_bf16: null pointer");
  if ((int64_t)B * Q == 0) return gvl::zero_fill(grad_value, gv_bytes, st);
  if (!value || !shapes || !lsi || !loc || !attn || !grad_out || !grad_loc || !grad_attn)
    return fail(GVL_EINVAL, "gvl_msda_backward_bf16: null pointer");
  const int nchunk = bwd_chunks(B, M, Q, S, plan.rowsV);
  if (nchunk <= 0) return fail(GVL_EINVAL, "gvl_msda_backward_bf16: problem does not fit LDS");
  return run_bwd_t1d<bf16_t, false>((const bf16_t *)value, shapes, lsi, loc, attn, (const bf16_t *)grad_out, B, S, M, L,
                                    Q, P, 0, pad_mode, plan, nchunk, shapes_host, (bf16_t *)grad_value, grad_loc,
                                    grad_attn, workspace, workspace_bytes, st);
}

}  // extern "C"
