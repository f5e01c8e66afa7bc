// micro-benchmark: LDS atomic throughput on gfx950 (u32 add, u32 add with return, f32 add), random 188-row slab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int MODE>
__global__ void __launch_bounds__(512) k(const int *idx, int n_per_thread, float *out) {
  __shared__ float fs[189 * 64];
  __shared__ unsigned us[189 * 64];
  for (int i = threadIdx.x; i < 189 * 64; i += blockDim.x) { fs[i] = 0; us[i] = 0; }
  __syncthreads();
  unsigned acc = 0;
  for (int it = 0; it < n_per_thread; ++it) {
    int a = idx[(it * blockDim.x + threadIdx.x) & 65535];
    if (MODE == 0) atomicAdd(&us[a], 1u);
    if (MODE == 1) acc += atomicAdd(&us[a], 1u);
    if (MODE == 2) atomicAdd(&fs[a], 1.0f);
    if (MODE == 3) { us[a] += 1; }                 // plain RMW (racy) for reference
    if (MODE == 4) atomicAdd(&fs[(a & ~63) | (threadIdx.x & 63)], 1.0f);   // conflict-free f32 (lane = column)
    if (MODE == 5) atomicAdd(&us[(a & ~63) | (threadIdx.x & 63)], 1u);     // conflict-free u32
  }
  __syncthreads();
  if (threadIdx.x == 0) out[blockIdx.x] = fs[5] + us[7] + acc;
}
int main() {
  std::vector<int> h(65536);
  unsigned s = 12345;
  for (auto &x : h) { s = s * 1664525u + 1013904223u; x = (s >> 8) % (188 * 64); }
  int *d; float *o;
  hipMalloc(&d, h.size() * 4); hipMalloc(&o, 4096 * 4);
  hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  const int n = 256;
  auto run = [&](auto kern, const char *name) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    kern<<<256, 512>>>(d, n, o); hipDeviceSynchronize();
    hipEventRecord(a); kern<<<256, 512>>>(d, n, o); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    double wave_instr_per_cu = 8.0 * n;  // 8 waves per WG, 1 WG per CU
    printf("%-28s %8.1f us  -> %.1f cycles per wave-instruction per CU (2.4GHz)\n", name, ms * 1e3,
           ms * 1e-3 * 2.4e9 / wave_instr_per_cu);
  };
  run(k<0>, "ds_add_u32 random");
  run(k<1>, "ds_add_rtn_u32 random");
  run(k<2>, "ds_add_f32 random");
  run(k<3>, "plain rmw random");
  run(k<4>, "ds_add_f32 conflict-free");
  run(k<5>, "ds_add_u32 conflict-free");
  return 0;
}
