// micro-benchmark (round 4): which clock does s_memtime count, and what is the shader clock in short / long kernels?
// A chain of N dependent v_fma_f32 (4 cycles each for one wave alone, MI355X_MICROARCH.md) is timed with s_memtime and
// s_memrealtime (100 MHz) around it, for chains of 0.02 .. 20 ms, one wave per SIMD on every CU.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void __launch_bounds__(256) k(long long *o, int n, float *sink) {
  float a = threadIdx.x * 1e-3f, b = 1.0001f;
  const long long r0 = __builtin_amdgcn_s_memrealtime(), t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < n; ++i) {
#pragma unroll
    for (int u = 0; u < 64; ++u) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a) : "v"(b));
  }
  const long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0) { o[blockIdx.x * 2] = t1 - t0; o[blockIdx.x * 2 + 1] = r1 - r0; }
  sink[blockIdx.x * 256 + threadIdx.x] = a;
}
int main() {
  long long *o; float *s;
  hipMalloc(&o, 256 * 16); hipMalloc(&s, 256 * 256 * 4);
  for (int n : {100, 1000, 10000, 100000}) {
    for (int rep = 0; rep < 3; ++rep) {
      k<<<256, 256>>>(o, n, s); hipDeviceSynchronize();
    }
    long long h[512];
    hipMemcpy(h, o, sizeof(h), hipMemcpyDeviceToHost);
    double t = 0, r = 0;
    for (int b = 0; b < 256; ++b) { t += h[2 * b]; r += h[2 * b + 1]; }
    t /= 256; r /= 256;
    printf("chain of %8d v_fma: s_memtime %12.0f ticks, s_memrealtime %10.0f ticks (= %8.1f us) -> s_memtime runs at %7.1f MHz; "
           "%.2f s_memtime ticks per v_fma; %.2f ns per v_fma -> %.0f MHz if a dependent v_fma takes 4 cycles\n",
           n * 64, t, r, r / 100.0, t / r * 100.0, t / (n * 64.0), r * 10.0 / (n * 64.0), 4.0 / (r * 10.0 / (n * 64.0)) * 1e3);
  }
  return 0;
}
