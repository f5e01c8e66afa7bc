// micro-benchmark (round 4): what a dispatch of the forward kernel's GEOMETRY costs when it does (almost) nothing -- 256
// workgroups x 1024 threads, 48 KB of dynamic LDS, hipExtLaunchKernel begin/end stamps as the library takes them.
//   variant 0: empty body; 1: every thread stores one float4 (4 MB of output); 2: stages a 47 KB slab per workgroup from a
//   12.3 MB tensor into LDS + barrier (the forward's staging round trip), no compute; 3: = 2 + the 9.8 MB output store
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <vector>
#include <algorithm>
template <int MODE>
__global__ void __launch_bounds__(1024) k(const float4 *__restrict__ value, float4 *__restrict__ out, int S) {
  extern __shared__ float4 slab[];
  if (MODE >= 2) {
    const int bm = blockIdx.x % 128, b = bm / 8, m = bm % 8;
    float4 acc = make_float4(0, 0, 0, 0);
    for (int i = threadIdx.x; i < S * 16; i += blockDim.x) slab[i] = value[((size_t)b * S * 8 + m) * 16 + (size_t)(i >> 4) * 128 + (i & 15)];
    __syncthreads();
    acc = slab[(threadIdx.x * 7) % (S * 16)];
    if (MODE == 3) {
      const int q0 = (blockIdx.x / 128) * 150;
      for (int q = q0 + (threadIdx.x >> 4); q < q0 + 150; q += 64)
        __builtin_nontemporal_store(acc.x, &out[(((size_t)b * 300 + q) * 8 + m) * 16 + (threadIdx.x & 15)].x),
        __builtin_nontemporal_store(acc.y, &out[(((size_t)b * 300 + q) * 8 + m) * 16 + (threadIdx.x & 15)].y),
        __builtin_nontemporal_store(acc.z, &out[(((size_t)b * 300 + q) * 8 + m) * 16 + (threadIdx.x & 15)].z),
        __builtin_nontemporal_store(acc.w, &out[(((size_t)b * 300 + q) * 8 + m) * 16 + (threadIdx.x & 15)].w);
    } else if (acc.x == 1234.5f) out[0] = acc;
  } else if (MODE == 1) {
    out[(size_t)blockIdx.x * 1024 + threadIdx.x] = make_float4(1, 2, 3, 4);
  }
}
int main() {
  const int S = 188;
  float4 *v, *o;
  hipMalloc(&v, (size_t)16 * S * 8 * 16 * 16); hipMalloc(&o, (size_t)16 * 300 * 8 * 16 * 16);
  hipMemset(v, 0, (size_t)16 * S * 8 * 16 * 16);
  void (*ks[])(const float4 *, float4 *, int) = {k<0>, k<1>, k<2>, k<3>};
  const char *names[] = {"empty", "one float4 store per thread (4 MB)", "slab staging (12.3 MB -> LDS) + barrier", "staging + 9.8 MB streamed output"};
  for (int m = 0; m < 4; ++m) {
    hipFuncSetAttribute((const void *)ks[m], hipFuncAttributeMaxDynamicSharedMemorySize, 48 * 1024 + 256);
    std::vector<float> us;
    for (int it = 0; it < 30; ++it) {
      hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
      hipExtLaunchKernelGGL(ks[m], dim3(256), dim3(1024), 48 * 1024 + 256, 0, a, b, 0, (const float4 *)v, o, S);
      hipEventSynchronize(b);
      float ms; hipEventElapsedTime(&ms, a, b);
      if (it >= 5) us.push_back(ms * 1e3f);
    }
    std::sort(us.begin(), us.end());
    printf("%-48s dispatch median %6.2f us (min %6.2f)\n", names[m], us[us.size() / 2], us[0]);
  }
  // the empty kernel at other geometries: (workgroups, threads, dynamic LDS bytes)
  const int geo[][3] = {{256, 1024, 0}, {256, 1024, 49408}, {256, 512, 49408}, {512, 512, 49408}, {256, 256, 49408}, {1024, 256, 0},
                        {256, 64, 0}, {1, 64, 0}, {256, 1024, 150000}};
  for (auto &g_ : geo) {
    hipFuncSetAttribute((const void *)ks[0], hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    std::vector<float> us;
    for (int it = 0; it < 30; ++it) {
      hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
      hipExtLaunchKernelGGL(ks[0], dim3(g_[0]), dim3(g_[1]), g_[2], 0, a, b, 0, (const float4 *)v, o, S);
      hipEventSynchronize(b);
      float ms; hipEventElapsedTime(&ms, a, b);
      if (it >= 5) us.push_back(ms * 1e3f);
    }
    std::sort(us.begin(), us.end());
    printf("empty kernel, %4d workgroups x %4d threads, %6d B LDS: dispatch median %6.2f us (min %6.2f)\n", g_[0], g_[1], g_[2],
           us[us.size() / 2], us[0]);
  }
  return 0;
}
