// gvl_optim.hip -- the update of the training step (train.py:405-409: clip_grad_norm_ + Adam.step) as three launches over a table of
// (parameter, gradient, first moment, second moment) tensors instead of torch's ~10 multi-tensor launches that read the gradients
// three times (norm, scale, update).  The optimizer's STATE stays torch.optim.Adam's (exp_avg, exp_avg_sq, step tensors: state_dict,
// checkpoint / resume unchanged); this file only computes what `clip_grad_norm_(params, max_norm); optimizer.step()` computes:
//   total = sqrt(sum g^2)                         (torch: the 2-norm of the per-tensor 2-norms)
//   coef  = min(1, max_norm / (total + 1e-6))     (torch/nn/utils/clip_grad.py)
//   g' = coef g (+ weight_decay p);  m = lerp(m, g', 1 - beta1);  v = beta2 v + (1 - beta2) g'^2
//   p -= (lr / (1 - beta1^t)) m / (sqrt(v) / sqrt(1 - beta2^t) + eps)            (torch's fused Adam, ADAM_MODE ORIGINAL)
// The clipped gradient is written back in the same pass (clip_grad_norm_ scales .grad in place: whoever reads it after the step sees
// the same tensor).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gvl_common.hpp"
#include "gvl_msda.h"

namespace {

using gvl::fail;
constexpr int kOptChunk = 4096, kOptThreads = 256;

__device__ __forceinline__ float block_sum(float v, float *sm) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) sm[w] = v;
  __syncthreads();
  return sm[0] + sm[1] + sm[2] + sm[3];                                // (fixed order: the same bits on every run)
}

__global__ void __launch_bounds__(kOptThreads) k_grad_sqnorm(const gvl_adam_desc *__restrict__ descs, const int2 *__restrict__ chunk_map,
                                                             float *__restrict__ partial) {
  __shared__ float sm[4];
  const int2 cm = chunk_map[blockIdx.x];
  const gvl_adam_desc d = descs[cm.x];
  const int64_t base = (int64_t)cm.y * kOptChunk, n = d.n - base < kOptChunk ? d.n - base : kOptChunk;
  const float *g = d.g + base;
  float s = 0.f;
  if (d.vec) {
    for (int i = threadIdx.x * 4; i < n; i += kOptThreads * 4) {
      const float4 x = *reinterpret_cast<const float4 *>(g + i);
      s += x.x * x.x + x.y * x.y + x.z * x.z + x.w * x.w;
    }
  } else {
    for (int i = threadIdx.x; i < n; i += kOptThreads) s += g[i] * g[i];
  }
  s = block_sum(s, sm);
  if (threadIdx.x == 0) partial[blockIdx.x] = s;
}

// scal: [0] total norm, [1] clip coefficient, [2] / [3] the bias corrections of tensor 0 (diagnostic);
// corr[2 i], corr[2 i + 1]: lr / (1 - beta1^t_i), sqrt(1 - beta2^t_i) from tensor i's OWN step count -- torch.optim.Adam keeps a
// step per parameter, and they differ as soon as a parameter sat out a step (no gradient on that batch) or got its state later
__global__ void __launch_bounds__(kOptThreads) k_adam_prep(const float *__restrict__ partial, int n_chunks, float max_norm,
                                                           const gvl_adam_desc *__restrict__ descs, int n_tensors, double lr,
                                                           double beta1, double beta2, float *__restrict__ scal,
                                                           float *__restrict__ corr) {
  __shared__ double sd[kOptThreads];
  for (int i = threadIdx.x; i < n_tensors; i += kOptThreads) {
    const double t = (double)*descs[i].step;
    const float a = (float)(lr / (1.0 - pow(beta1, t))), b = (float)sqrt(1.0 - pow(beta2, t));
    corr[2 * i] = a;
    corr[2 * i + 1] = b;
    if (i == 0) scal[2] = a, scal[3] = b;
  }
  double s = 0.0;
  for (int i = threadIdx.x; i < n_chunks; i += kOptThreads) s += (double)partial[i];
  sd[threadIdx.x] = s;
  __syncthreads();
  for (int o = kOptThreads / 2; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) sd[threadIdx.x] += sd[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const float total = (float)sqrt(sd[0]);
    // clip_grad_norm_'s own coefficient, max_norm passed straight through: max_norm = 0 scales every gradient to 0, as the
    // reference's unconditional call does (train.py:407) -- the fallback and this path agree for every value
    const float c = max_norm / (total + 1e-6f);
    scal[0] = total;
    scal[1] = c < 1.f ? c : 1.f;
  }
}

__device__ __forceinline__ void adam1(float &p, float &gc, float &m, float &v, float coef, float wd, float w1, float beta2, float w2,
                                      float step_size, float bc2s, float eps) {
  gc *= coef;                                                          // (what clip_grad_norm_ leaves in .grad)
  float g = gc;
  if (wd != 0.f) g += wd * p;
  m = m + w1 * (g - m);
  v = beta2 * v + w2 * g * g;
  const float denom = sqrtf(v) / bc2s + eps;
  p -= step_size * m / denom;
}

__global__ void __launch_bounds__(kOptThreads) k_adam(const gvl_adam_desc *__restrict__ descs, const int2 *__restrict__ chunk_map,
                                                      const float *__restrict__ scal, const float *__restrict__ corr, float w1,
                                                      float beta2, float w2, float eps, float wd) {
  const int2 cm = chunk_map[blockIdx.x];
  const gvl_adam_desc d = descs[cm.x];
  const int64_t base = (int64_t)cm.y * kOptChunk, n = d.n - base < kOptChunk ? d.n - base : kOptChunk;
  float *p = d.p + base, *m = d.m + base, *v = d.v + base, *g = const_cast<float *>(d.g) + base;
  const float coef = scal[1], step_size = corr[2 * cm.x], bc2s = corr[2 * cm.x + 1];
  if (d.vec) {
    for (int i = threadIdx.x * 4; i < n; i += kOptThreads * 4) {
      float4 pp = *reinterpret_cast<float4 *>(p + i), mm = *reinterpret_cast<float4 *>(m + i), vv = *reinterpret_cast<float4 *>(v + i);
      float4 gg = *reinterpret_cast<const float4 *>(g + i);
      adam1(pp.x, gg.x, mm.x, vv.x, coef, wd, w1, beta2, w2, step_size, bc2s, eps);
      adam1(pp.y, gg.y, mm.y, vv.y, coef, wd, w1, beta2, w2, step_size, bc2s, eps);
      adam1(pp.z, gg.z, mm.z, vv.z, coef, wd, w1, beta2, w2, step_size, bc2s, eps);
      adam1(pp.w, gg.w, mm.w, vv.w, coef, wd, w1, beta2, w2, step_size, bc2s, eps);
      *reinterpret_cast<float4 *>(p + i) = pp;
      if (coef != 1.f) *reinterpret_cast<float4 *>(g + i) = gg;
      *reinterpret_cast<float4 *>(m + i) = mm;
      *reinterpret_cast<float4 *>(v + i) = vv;
    }
  } else {
    for (int i = threadIdx.x; i < n; i += kOptThreads) {
      float gi = g[i];
      adam1(p[i], gi, m[i], v[i], coef, wd, w1, beta2, w2, step_size, bc2s, eps);
      if (coef != 1.f) g[i] = gi;
    }
  }
}

// the gradients' addresses travel as KERNEL ARGUMENTS: a captured step's gradients live in the graph's pool (other addresses than
// the warm-up run's), and a table upload cannot be recorded into a hipGraph -- a launch with its arguments can
constexpr int kPtrBatch = 448;
struct GradPtrs { const float *g[kPtrBatch]; };

__global__ void __launch_bounds__(kOptThreads) k_adam_set_grads(gvl_adam_desc *__restrict__ descs, int first, int count, const GradPtrs ptrs) {
  for (int i = threadIdx.x; i < count; i += kOptThreads) descs[first + i].g = ptrs.g[i];
}

}  // namespace

extern "C" int gvl_adam_chunk_elems(void) { return kOptChunk; }

extern "C" int gvl_adam_set_grads(gvl_adam_desc *descs_device, int n_tensors, const void *const *grads_host, void *stream) {
  if (!descs_device || !grads_host || n_tensors <= 0) return fail(GVL_EINVAL, "gvl_adam_set_grads: null pointer / empty table");
  for (int first = 0; first < n_tensors; first += kPtrBatch) {
    GradPtrs b;
    const int count = n_tensors - first < kPtrBatch ? n_tensors - first : kPtrBatch;
    for (int i = 0; i < count; ++i) {
      if (!grads_host[first + i] || ((uintptr_t)grads_host[first + i] & 3)) return fail(GVL_EINVAL, "gvl_adam_set_grads: gradient %d null / unaligned", first + i);
      b.g[i] = (const float *)grads_host[first + i];
    }
    hipLaunchKernelGGL(k_adam_set_grads, dim3(1), dim3(kOptThreads), 0, (hipStream_t)stream, descs_device, first, count, b);
  }
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : fail((int)e, "gvl_adam_set_grads: launch failed: %s", hipGetErrorString(e));
}

extern "C" int gvl_clip_adam_step_f32(const gvl_adam_desc *descs_device, int n_tensors, const int *chunk_map_device, int n_chunks,
                                      float *partial_device, float *scal_device, float *corr_device, double max_norm,
                                      double lr, double beta1, double beta2, double eps, double weight_decay, void *stream) {
  if (!descs_device || !chunk_map_device || !partial_device || !scal_device || !corr_device || n_chunks <= 0 || n_tensors <= 0)
    return fail(GVL_EINVAL, "gvl_clip_adam_step_f32: null pointer / empty launch");
  if (!(lr >= 0.0) || !(beta1 >= 0.0 && beta1 < 1.0) || !(beta2 >= 0.0 && beta2 < 1.0) || !(eps >= 0.0))
    return fail(GVL_EINVAL, "gvl_clip_adam_step_f32: lr, eps >= 0 and betas in [0, 1) (got %g %g %g %g)", lr, beta1, beta2, eps);
  hipStream_t st = (hipStream_t)stream;
  const int2 *cm = reinterpret_cast<const int2 *>(chunk_map_device);
  hipLaunchKernelGGL(k_grad_sqnorm, dim3(n_chunks), dim3(kOptThreads), 0, st, descs_device, cm, partial_device);
  hipLaunchKernelGGL(k_adam_prep, dim3(1), dim3(kOptThreads), 0, st, (const float *)partial_device, n_chunks, (float)max_norm,
                     descs_device, n_tensors, lr, beta1, beta2, scal_device, corr_device);
  // (1 - beta in double, as torch's kernel forms it: 1.f - 0.999f is 1.3e-5 away from 0.001)
  hipLaunchKernelGGL(k_adam, dim3(n_chunks), dim3(kOptThreads), 0, st, descs_device, cm, (const float *)scal_device,
                     (const float *)corr_device, (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), (float)eps, (float)weight_decay);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : fail((int)e, "gvl_clip_adam_step_f32: launch failed: %s", hipGetErrorString(e));
}
