// gvl_common.hpp -- internals shared by the translation units of libgvl_msda.so (not part of the C ABI).
#pragma once
#include <hip/hip_ext.h>
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>

#include <stdlib.h>

#include <mutex>
#include <string>
#include <utility>
#include <vector>

#include "gvl_msda.h"

namespace gvl {

inline thread_local char g_err[512] = "";

inline int fail(int code, const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

// ---- experiment / A-B switches (GVL_* environment variables): read ONCE per process and name -- keyed by the address of the
// string literal -- so that no launch path calls getenv(); gvl_reload_env() drops the cache (tests that flip a switch between calls)
struct EnvCache {
  struct Entry { const char *name; bool set; std::string value; };
  std::mutex mu;
  std::vector<Entry *> seen;                 // (entries are never freed while a caller may hold their value)
  std::vector<Entry *> retired;
};
inline EnvCache &env_cache() {
  static EnvCache c;
  return c;
}
inline const char *env_str(const char *name) {               // nullptr when unset or empty
  EnvCache &c = env_cache();
  std::lock_guard<std::mutex> g(c.mu);
  for (EnvCache::Entry *e : c.seen)
    if (e->name == name) return e->set ? e->value.c_str() : nullptr;
  const char *v = getenv(name);
  EnvCache::Entry *e = new EnvCache::Entry{name, v && *v, v ? v : ""};
  c.seen.push_back(e);
  return e->set ? e->value.c_str() : nullptr;
}
inline int env_int(const char *name, int dflt) {
  const char *v = env_str(name);
  return v ? atoi(v) : dflt;
}
inline void env_reload() {
  EnvCache &c = env_cache();
  std::lock_guard<std::mutex> g(c.mu);
  c.retired.insert(c.retired.end(), c.seen.begin(), c.seen.end());
  c.seen.clear();
}

// ---- in-library kernel timing: exact begin/end stamps of individual dispatches (hipExtLaunchKernel events) ----
struct ProfEntry {
  hipEvent_t start, stop;
  int tag, a, b;
};
struct Profiler {
  std::mutex mu;
  int level = 0;      // 0 off | 1 the kernels of the sampling path | 2 additionally the projection GEMM in front of them
  std::vector<ProfEntry> entries;
};
inline Profiler &profiler() {
  static Profiler p;
  return p;
}

template <typename K, typename... Args>
inline int launch(int tag, int meta_a, int meta_b, const char *what, K kernel, dim3 grid, dim3 block, size_t lds,
                  hipStream_t st, Args... args) {
  Profiler &p = profiler();
  // Level 1 leaves the projection kernel (the launch directly in front of the sampling kernel) un-stamped: two
  // event-stamped launches back to back inflate the second one's interval by 2-3 us (measured: 12.2 vs 9.2-9.5 us for
  // the same dispatch, whose duration in a rocprofv3 trace is identical either way), and the bench's roofline block
  // reports exactly that second launch.
  bool stamp = p.level > 0 && (tag != GVL_PROF_PROJ || p.level > 1);
  if (stamp) {   // event-stamped launches cannot be recorded into a hipGraph: inside a capture launch plainly
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) stamp = false;
  }
  if (stamp) {
    ProfEntry e;
    e.tag = tag; e.a = meta_a; e.b = meta_b;
    if (hipEventCreate(&e.start) != hipSuccess || hipEventCreate(&e.stop) != hipSuccess)
      return fail(GVL_EINVAL, "gvl: cannot create profiling events");
    hipExtLaunchKernelGGL(kernel, grid, block, (unsigned)lds, st, e.start, e.stop, 0, args...);
    std::lock_guard<std::mutex> g(p.mu);
    p.entries.push_back(e);
  } else {
    hipLaunchKernelGGL(kernel, grid, block, lds, st, args...);
  }
  hipError_t err = hipGetLastError();
  if (err != hipSuccess) return fail((int)err, "gvl: %s launch failed: %s", what, hipGetErrorString(err));
  return 0;
}

// zero-fill as a KERNEL node.  hipMemsetAsync inside a captured stream did not re-execute reliably on hipGraph replay
// (ROCm 7.2: gradients accumulated on top of the previous replay's values), a kernel launch always does.
static __global__ void k_zero_fill(uint4 *__restrict__ p, size_t n16, unsigned char *__restrict__ tail, int ntail) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x)
    p[i] = make_uint4(0u, 0u, 0u, 0u);
  if (blockIdx.x == 0 && (int)threadIdx.x < ntail) tail[threadIdx.x] = 0;
}

inline int zero_fill(void *ptr, size_t bytes, hipStream_t st) {
  if (bytes == 0) return 0;
  if (reinterpret_cast<uintptr_t>(ptr) & 15) {               // not 16-byte aligned: fall back (never the case for torch)
    hipError_t e = hipMemsetAsync(ptr, 0, bytes, st);
    return e == hipSuccess ? 0 : fail((int)e, "gvl: memset failed: %s", hipGetErrorString(e));
  }
  const size_t n16 = bytes / 16;
  const int ntail = (int)(bytes % 16);
  size_t blocks = (n16 + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  if (blocks == 0) blocks = 1;
  hipLaunchKernelGGL(k_zero_fill, dim3((unsigned)blocks), dim3(256), 0, st, reinterpret_cast<uint4 *>(ptr), n16,
                     reinterpret_cast<unsigned char *>(ptr) + n16 * 16, ntail);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : fail((int)e, "gvl: zero-fill launch failed: %s", hipGetErrorString(e));
}

template <typename K>
inline int ensure_lds(K kernel, size_t bytes) {
  if (bytes <= 64 * 1024) return 0;
  // remember the largest size already granted per kernel: the attribute call is made once, never per launch
  // (and never again during a hipGraph capture after the warm-up launch)
  static std::mutex mu;
  static std::vector<std::pair<const void *, size_t>> granted;
  {
    std::lock_guard<std::mutex> g(mu);
    for (auto &kv : granted)
      if (kv.first == reinterpret_cast<const void *>(kernel) && kv.second >= bytes) return 0;
  }
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kernel),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  if (e != hipSuccess) return fail((int)e, "gvl: cannot raise dynamic LDS to %zu: %s", bytes, hipGetErrorString(e));
  std::lock_guard<std::mutex> g(mu);
  granted.emplace_back(reinterpret_cast<const void *>(kernel), bytes);
  return 0;
}

}  // namespace gvl

// ---- LSTM cell of one hidden unit (nn.LSTM, gate order i f g o; LSTM_DSA.py:216-217,269): shared by the pointwise kernel
// (gvl_cap.hip: k_lstm_cell) and the GEMM epilogue that absorbs it (gvl_gemm16.hip: kLstm) -- ONE expression, contraction
// written out, so that the two paths produce the same bits
__device__ __forceinline__ float gvl_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.f + __expf(-x)); }
__device__ __forceinline__ float gvl_tanh(float x) { return fmaf(-2.f, __builtin_amdgcn_rcpf(1.f + __expf(2.f * x)), 1.f); }
__device__ __forceinline__ void gvl_lstm_point(float gi, float gf, float gg, float go, float cp, float &cn, float &hn) {
  cn = fmaf(gvl_sigmoid(gf), cp, gvl_sigmoid(gi) * gvl_tanh(gg));
  hn = gvl_sigmoid(go) * gvl_tanh(cn);
}

// ---- storage types: fp32 or bf16 in HBM, fp32 in registers / LDS.  `i4` indexes groups of 4 consecutive elements.
using bf16_t = __bf16;
struct alignas(8) bf16x4 { bf16_t a, b, c, d; };

__device__ inline float4 ld4(const float *base, int64_t i4) { return reinterpret_cast<const float4 *>(base)[i4]; }
__device__ inline float4 ld4(const bf16_t *base, int64_t i4) {
  const bf16x4 v = reinterpret_cast<const bf16x4 *>(base)[i4];
  return make_float4((float)v.a, (float)v.b, (float)v.c, (float)v.d);
}
__device__ inline void st4(float *base, int64_t i4, float4 v) { reinterpret_cast<float4 *>(base)[i4] = v; }
// streaming form for outputs the kernel never reads back ("nt": written through instead of parked dirty in the XCD's L2
// until the end-of-kernel release has to flush it)
__device__ inline void st4_stream(float *base, int64_t i4, float4 v) {
  typedef float f4n __attribute__((ext_vector_type(4)));
  __builtin_nontemporal_store((f4n){v.x, v.y, v.z, v.w}, reinterpret_cast<f4n *>(base) + i4);
}
__device__ inline void st4_stream(bf16_t *base, int64_t i4, float4 v);
__device__ inline void st_stream(float *p, float v) { __builtin_nontemporal_store(v, p); }
__device__ inline void st_stream(float2 *p, float2 v) {
  typedef float f2n __attribute__((ext_vector_type(2)));
  __builtin_nontemporal_store((f2n){v.x, v.y}, reinterpret_cast<f2n *>(p));
}
__device__ inline void st4(bf16_t *base, int64_t i4, float4 v) {            // v_cvt_pk_bf16_f32: round-to-nearest-even
  bf16x4 o;
  o.a = (bf16_t)v.x; o.b = (bf16_t)v.y; o.c = (bf16_t)v.z; o.d = (bf16_t)v.w;
  reinterpret_cast<bf16x4 *>(base)[i4] = o;
}
__device__ inline void st4_stream(bf16_t *base, int64_t i4, float4 v) {
  bf16x4 o;
  o.a = (bf16_t)v.x; o.b = (bf16_t)v.y; o.c = (bf16_t)v.z; o.d = (bf16_t)v.w;
  __builtin_nontemporal_store(__builtin_bit_cast(unsigned long long, o), reinterpret_cast<unsigned long long *>(base) + i4);
}
