// micro-benchmark (round 4): issue cost of the vector instructions the temporal kernels are built from, on gfx950, at 1, 2
// and 4 wavefronts per SIMD.  Every kernel runs N iterations of a block of 16 independent-ish instructions of ONE kind;
// cycles per instruction per SIMD = s_memtime delta / (16 N waves_per_simd).
//   hipcc --offload-arch=gfx950 -O3 -o valu_cost valu_cost.hip && ./valu_cost
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f2v __attribute__((ext_vector_type(2)));
typedef float f4v __attribute__((ext_vector_type(4)));

#define REP16(...) _Pragma("unroll") for (int r_ = 0; r_ < 16; ++r_) { __VA_ARGS__ }

template <int SRC>
__device__ inline f2v row_bcast_f2(f2v v) {
  const long long r = __builtin_amdgcn_update_dpp((long long)0, __builtin_bit_cast(long long, v), 0x150 + SRC, 0xF, 0xF, true);
  return __builtin_bit_cast(f2v, r);
}
template <int S0>
__device__ inline void gather4(const f2v tc, const char *G_b, int lane_off, f2v &a01, f2v &a23) {
  const f2v t0 = row_bcast_f2<S0>(tc), t1 = row_bcast_f2<S0 + 1>(tc), t2 = row_bcast_f2<S0 + 2>(tc),
            t3 = row_bcast_f2<S0 + 3>(tc);
  const f4v g0 = *reinterpret_cast<const f4v *>(G_b + __builtin_bit_cast(int, t0.x) + lane_off);
  const f4v g1 = *reinterpret_cast<const f4v *>(G_b + __builtin_bit_cast(int, t1.x) + lane_off);
  const f4v g2 = *reinterpret_cast<const f4v *>(G_b + __builtin_bit_cast(int, t2.x) + lane_off);
  const f4v g3 = *reinterpret_cast<const f4v *>(G_b + __builtin_bit_cast(int, t3.x) + lane_off);
#define GVL_ACC(T, G)                                                           \
  {                                                                             \
    const f2v cf = __builtin_shufflevector(T, T, 1, 1);                         \
    a01 = __builtin_elementwise_fma(cf, (f2v){G.x, G.y}, a01);                  \
    a23 = __builtin_elementwise_fma(cf, (f2v){G.z, G.w}, a23);                  \
  }
  GVL_ACC(t0, g0) GVL_ACC(t1, g1) GVL_ACC(t2, g2) GVL_ACC(t3, g3)
}

template <int MODE>
__global__ void __launch_bounds__(1024) k(float *out, long long *cyc, int n) {
  __shared__ f4v lds[4096];
  for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = (f4v){1.f, 2.f, 3.f, 4.f};
  __syncthreads();
  float a0 = threadIdx.x * 1e-3f, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f;
  f2v p0 = {a0, a1}, p1 = {a2, a3}, c = {1.0001f, 0.9999f};
  f2v q0 = {a0, a1}, q1 = {a2, a3};
  int addr = (threadIdx.x & 15) * 16 + ((threadIdx.x >> 4) & 63) * 256;
  int ia = threadIdx.x;
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < n; ++it) {
    if (MODE == 0) {   // v_fma_f32, 4 independent chains
      REP16(asm volatile("v_fma_f32 %0, %0, %1, %1\n\tv_fma_f32 %2, %2, %3, %3\n\tv_fma_f32 %1, %1, %0, %0\n\tv_fma_f32 %3, %3, %2, %2"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));)
    } else if (MODE == 1) {   // v_pk_fma_f32, 4 chains
      REP16(asm volatile("v_pk_fma_f32 %0, %2, %0, %0\n\tv_pk_fma_f32 %1, %2, %1, %1\n\tv_pk_fma_f32 %3, %2, %3, %3\n\tv_pk_fma_f32 %4, %2, %4, %4"
                         : "+v"(p0), "+v"(p1), "+v"(c), "+v"(q0), "+v"(q1));)
    } else if (MODE == 2) {   // v_mov_b64_dpp row_newbcast
      REP16(asm volatile("v_mov_b64_dpp %0, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                         "v_mov_b64_dpp %1, %2 row_newbcast:5 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                         "v_mov_b64_dpp %3, %2 row_newbcast:7 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                         "v_mov_b64_dpp %4, %2 row_newbcast:9 row_mask:0xf bank_mask:0xf bound_ctrl:1"
                         : "+v"(p0), "+v"(p1), "+v"(c), "+v"(q0), "+v"(q1));)
    } else if (MODE == 3) {   // v_mov_b32_dpp row_newbcast
      REP16(asm volatile("v_mov_b32_dpp %0, %4 row_newbcast:3 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                         "v_mov_b32_dpp %1, %4 row_newbcast:5 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                         "v_mov_b32_dpp %2, %4 row_newbcast:7 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                         "v_mov_b32_dpp %3, %4 row_newbcast:9 row_mask:0xf bank_mask:0xf bound_ctrl:1"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(c.x));)
    } else if (MODE == 4) {   // v_fmac_f32_dpp: acc += bcast(coef) * g
      REP16(asm volatile("v_fmac_f32_dpp %0, %4, %5 row_newbcast:3 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                         "v_fmac_f32_dpp %1, %4, %5 row_newbcast:3 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                         "v_fmac_f32_dpp %2, %4, %5 row_newbcast:3 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                         "v_fmac_f32_dpp %3, %4, %5 row_newbcast:3 row_mask:0xf bank_mask:0xf bound_ctrl:1"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(c.x), "v"(c.y));)
    } else if (MODE == 5) {   // v_add_u32_dpp
      REP16(asm volatile("v_add_u32_dpp %0, %2, %3 row_newbcast:3 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                         "v_add_u32_dpp %1, %2, %3 row_newbcast:5 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                         "v_add_u32_dpp %0, %2, %3 row_newbcast:7 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                         "v_add_u32_dpp %1, %2, %3 row_newbcast:9 row_mask:0xf bank_mask:0xf bound_ctrl:1"
                         : "+v"(ia), "+v"(addr) : "v"(threadIdx.x), "v"(blockIdx.x));)
    } else if (MODE == 6) {   // ds_read_b128, 4 in flight, conflict-free rows
      REP16(f4v r0, r1, r2, r3;
            asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:256\n\tds_read_b128 %2, %4 offset:512\n\t"
                         "ds_read_b128 %3, %4 offset:768\n\ts_waitcnt lgkmcnt(0)"
                         : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3) : "v"(addr));
            a0 += r0.x + r1.y + r2.z + r3.w;)
    } else if (MODE == 7) {   // the gather step as shipped: bcast64 + add + read + 2 pk_fma, 4 steps grouped
      REP16(f2v tc = {__builtin_bit_cast(float, (ia & 63) << 8), c.x};
            asm volatile("" : "+v"(tc));
            gather4<0>(tc, (const char *)lds, (threadIdx.x & 15) * 16, p0, p1);)
    } else if (MODE == 8) {   // the same 4 steps with 32-bit DPP operands: add_dpp (address) + read + 4 fmac_dpp
      REP16(f4v r0, r1, r2, r3; int x0, x1, x2, x3;
            asm volatile("v_add_u32_dpp %4, %8, %9 row_newbcast:0 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                         "v_add_u32_dpp %5, %8, %9 row_newbcast:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                         "v_add_u32_dpp %6, %8, %9 row_newbcast:2 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                         "v_add_u32_dpp %7, %8, %9 row_newbcast:3 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                         "v_and_b32 %4, 0xfff0, %4\n\tv_and_b32 %5, 0xfff0, %5\n\tv_and_b32 %6, 0xfff0, %6\n\tv_and_b32 %7, 0xfff0, %7\n\t"
                         "ds_read_b128 %0, %4\n\tds_read_b128 %1, %5\n\tds_read_b128 %2, %6\n\tds_read_b128 %3, %7\n\t"
                         "s_waitcnt lgkmcnt(0)"
                         : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3), "=&v"(x0), "=&v"(x1), "=&v"(x2), "=&v"(x3) : "v"(ia), "v"(addr));
#define FM(ACC, G, S) asm volatile("v_fmac_f32_dpp %0, %1, %2 row_newbcast:" #S " row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(ACC) : "v"(c.x), "v"(G));
            FM(a0, r0.x, 0) FM(a1, r0.y, 0) FM(a2, r0.z, 0) FM(a3, r0.w, 0)
            FM(a0, r1.x, 1) FM(a1, r1.y, 1) FM(a2, r1.z, 1) FM(a3, r1.w, 1)
            FM(a0, r2.x, 2) FM(a1, r2.y, 2) FM(a2, r2.z, 2) FM(a3, r2.w, 2)
            FM(a0, r3.x, 3) FM(a1, r3.y, 3) FM(a2, r3.z, 3) FM(a3, r3.w, 3))
    } else if (MODE == 9) {   // v_cndmask + v_cmp pair (the coefficient arithmetic's staple)
      REP16(asm volatile("v_cmp_lt_f32 vcc, %0, %1\n\tv_cndmask_b32 %2, %2, %3, vcc\n\tv_cmp_lt_f32 vcc, %1, %0\n\tv_cndmask_b32 %3, %3, %2, vcc"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : : "vcc");)
    }
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + p0.x + p0.y + p1.x + p1.y + q0.x + q1.y + ia + addr;
}

int main() {
  float *o; long long *c;
  hipMalloc(&o, 256 * 1024 * 4); hipMalloc(&c, 256 * 16 * 8);
  const int n = 200;
  const char *names[] = {"v_fma_f32", "v_pk_fma_f32", "v_mov_b64_dpp row_newbcast", "v_mov_b32_dpp row_newbcast",
                         "v_fmac_f32_dpp row_newbcast", "v_add_u32_dpp row_newbcast", "ds_read_b128 (4 in flight, per read)",
                         "gather step x4: b64 bcast + and + read + 2 pk_fma (per step)",
                         "gather step x4: add_dpp + and + read + 4 fmac_dpp (per step)", "v_cmp + v_cndmask (per instr)"};
  const int per_iter[] = {64, 64, 64, 64, 64, 64, 64, 64, 64, 64};
  void (*ks[])(float *, long long *, int) = {k<0>, k<1>, k<2>, k<3>, k<4>, k<5>, k<6>, k<7>, k<8>, k<9>};
  for (int m = 0; m < 10; ++m)
    for (int threads : {256, 512, 1024}) {
      hipMemset(c, 0, 256 * 16 * 8);
      ks[m]<<<256, threads>>>(o, c, n); hipDeviceSynchronize();
      ks[m]<<<256, threads>>>(o, c, n); hipDeviceSynchronize();
      std::vector<long long> h(256 * 16);
      hipMemcpy(h.data(), c, h.size() * 8, hipMemcpyDeviceToHost);
      double s = 0; int cnt = 0;
      for (int b = 0; b < 256; ++b) for (int w = 0; w < threads / 64; ++w) { s += (double)h[b * 16 + w]; ++cnt; }
      const double per_wave = s / cnt / (double)(n * per_iter[m]);          // cycles per instruction seen by one wave
      const int wps = threads / 256;
      printf("%-66s %d wave/SIMD: %6.2f cyc per instr per wave -> %5.2f cyc per instr per SIMD\n", names[m], wps, per_wave,
             per_wave / wps);
    }
  return 0;
}
