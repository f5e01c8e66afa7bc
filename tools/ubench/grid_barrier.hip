// Micro-benchmark: what a device-wide barrier costs on MI355X inside ONE launch (256 workgroups, one per CU, all resident):
// arrive = device-scope release fence + atomic add on a global counter by one thread, wait = spin on the counter (s_sleep between
// polls, bounded: a barrier that never completes sets a flag and returns instead of hanging the GPU), then an acquire fence.
// Between two barriers every workgroup writes a line and reads its neighbour's (the data really crosses XCDs).
//   hipcc --offload-arch=gfx950 -O3 -o tools/_bin/grid_barrier tools/ubench/grid_barrier.hip && tools/_bin/grid_barrier
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__device__ __forceinline__ bool grid_barrier(unsigned *counter, unsigned target, int *err) {
  __syncthreads();
  bool ok = true;
  if (threadIdx.x == 0) {
    __threadfence();                                       // release: this workgroup's writes reach the device scope
    atomicAdd(counter, 1u);
    unsigned spins = 0;
    while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      __builtin_amdgcn_s_sleep(1);
      if (++spins > (1u << 22)) { *err = 1; ok = false; break; }
    }
    __threadfence();                                       // acquire
  }
  __syncthreads();
  return ok;
}

__global__ void __launch_bounds__(256) k_barriers(unsigned *counter, int n_barriers, float *buf, int *err, float *out) {
  const int g = blockIdx.x, G = gridDim.x;
  float acc = 0.f;
  for (int i = 0; i < n_barriers; ++i) {
    buf[((size_t)i * G + g) * 64 + (threadIdx.x & 63)] = (float)(i + g);            // a line per workgroup and round
    if (!grid_barrier(counter, (unsigned)(i + 1) * G, err)) return;
    acc += buf[((size_t)i * G + (g + 97) % G) * 64 + (threadIdx.x & 63)];             // another XCD's line
  }
  if (threadIdx.x == 0) out[g] = acc;
}

// variant B: the exchanged data written and read with system-coherent accesses (sc0 sc1: write-through / bypass of the XCD's L2),
// the barrier itself only waits for the workgroup's stores and uses the atomic counter -- no L2 write-back / invalidate
__device__ __forceinline__ void store_sc(float *p, float v) {
  asm volatile("global_store_dword %0, %1, off sc0 sc1" : : "v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ float load_sc(const float *p) {
  float v;
  asm volatile("global_load_dword %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  return v;
}
__device__ __forceinline__ bool grid_barrier_light(unsigned *counter, unsigned target, int *err) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // this wavefront's write-through stores have completed
  __syncthreads();
  bool ok = true;
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned spins = 0;
    while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      __builtin_amdgcn_s_sleep(1);
      if (++spins > (1u << 22)) { *err = 1; ok = false; break; }
    }
  }
  __syncthreads();
  return ok;
}
__global__ void __launch_bounds__(256) k_barriers_light(unsigned *counter, int n_barriers, float *buf, int *err, float *out) {
  const int g = blockIdx.x, G = gridDim.x;
  float acc = 0.f;
  for (int i = 0; i < n_barriers; ++i) {
    store_sc(buf + ((size_t)i * G + g) * 64 + (threadIdx.x & 63), (float)(i + g));
    if (!grid_barrier_light(counter, (unsigned)(i + 1) * G, err)) return;
    acc += load_sc(buf + ((size_t)i * G + (g + 97) % G) * 64 + (threadIdx.x & 63));
  }
  if (threadIdx.x == 0) out[g] = acc;
}

__global__ void k_empty() {}

int main() {
  const int G = 256, NB = 64;
  unsigned *counter; int *err; float *buf, *out;
  (void)hipMalloc(&counter, 4); (void)hipMalloc(&err, 4); (void)hipMalloc(&buf, (size_t)NB * G * 64 * 4); (void)hipMalloc(&out, G * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int nb : {0, 1, 8, 64}) {
    float best = 1e9f;
    for (int rep = 0; rep < 10; ++rep) {
      hipMemset(counter, 0, 4); hipMemset(err, 0, 4); hipMemset(buf, 0, (size_t)NB * G * 64 * 4);
      hipDeviceSynchronize();
      hipEventRecord(e0);
      hipLaunchKernelGGL(k_barriers, dim3(G), dim3(256), 0, 0, counter, nb, buf, err, out);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      best = ms < best ? ms : best;
    }
    int herr; hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost);
    std::vector<float> h(G); hipMemcpy(h.data(), out, G * 4, hipMemcpyDeviceToHost);
    double want = 0; for (int i = 0; i < nb; ++i) want += i + (0 + 97) % G;
    printf("%2d barriers: %.2f us per launch (err %d, out[0] %.0f want %.0f)\n", nb, best * 1e3, herr, h[0], want);
  }
  for (int nb : {1, 8, 64}) {
    float best = 1e9f;
    for (int rep = 0; rep < 10; ++rep) {
      (void)hipMemset(counter, 0, 4); (void)hipMemset(err, 0, 4); (void)hipMemset(buf, 0, (size_t)NB * G * 64 * 4);
      (void)hipDeviceSynchronize();
      (void)hipEventRecord(e0);
      hipLaunchKernelGGL(k_barriers_light, dim3(G), dim3(256), 0, 0, counter, nb, buf, err, out);
      (void)hipEventRecord(e1);
      (void)hipEventSynchronize(e1);
      float ms; (void)hipEventElapsedTime(&ms, e0, e1);
      best = ms < best ? ms : best;
    }
    int herr; (void)hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost);
    std::vector<float> h(G); (void)hipMemcpy(h.data(), out, G * 4, hipMemcpyDeviceToHost);
    double want = 0; for (int i = 0; i < nb; ++i) want += i + (0 + 97) % G;
    printf("write-through data, no L2 write-back: %2d barriers: %.2f us per launch (err %d, out[0] %.0f want %.0f)\n", nb, best * 1e3, herr, h[0], want);
  }
  return 0;
}
