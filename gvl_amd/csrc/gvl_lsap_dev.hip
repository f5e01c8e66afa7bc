// gvl_lsap_dev.hip -- the Hungarian matcher's index path ON THE DEVICE (one wavefront per assignment problem).
//
// Why: the reference moves the cost tensor to the host and calls scipy per video (pdvc/matcher.py:120-128); that
// device->host->device round trip in the middle of every train / eval step also forbids capturing the step in a
// hipGraph.  This kernel solves all (layer, video, {one-to-one, 4x-tiled}) problems of a step in one launch and
// leaves int64 indices in device memory, with results BIT-IDENTICAL to scipy.optimize.linear_sum_assignment:
// it is the same algorithm as gvl_lsap.cpp (Crouse 2016: shortest augmenting paths, rows in order, transposed when
// there are more rows than columns) with the same float64 arithmetic in the same order, and a parallel column scan
// whose selection rule reproduces the sequential scan exactly:
//     sequential:  for it in 0..remaining: if spc < lowest or (spc == lowest and column unassigned): take it
//     ==           m = min spc;  if any unassigned column attains m -> the LAST such `it`, else the FIRST `it` with m.
// Costs are float32 promoted to float64 (as scipy does); only additions / subtractions occur, so there is nothing
// for FMA contraction to change.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gvl_common.hpp"
#include "gvl_msda.h"

namespace {

using gvl::fail;

constexpr int kMaxCols = 1024;   // max(nr, nc) of one problem after orientation
constexpr int kMaxRows = 256;    // min(nr, nc)
constexpr int kMaxStaged = 7168; // distinct costs (Q * n) kept in LDS; larger problems read them from global memory

struct Problem {
  // view of one problem inside the cost tensor: element (q, k) = C[base + q * ld + (k % n)]   (k < n * tile);
  // eight int64 per problem so that the C ABI can describe it as a plain int64 array
  int64_t base, ld, Q, n, tile;
  int64_t out_off;               // offset of this problem's result inside rows_out / cols_out
  int64_t reserved0, reserved1;
};

__device__ inline double wave_min(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmin(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ inline int wave_max_i(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ inline int wave_min_i(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = min(v, __shfl_xor(v, o, 64));
  return v;
}

__global__ void __launch_bounds__(64) k_lsap(const float *__restrict__ C, const Problem *__restrict__ probs,
                                             int64_t *__restrict__ rows_out, int64_t *__restrict__ cols_out,
                                             int *__restrict__ status) {
  __shared__ double u[kMaxRows], v[kMaxCols], spc[kMaxCols];
  __shared__ int path[kMaxCols], row4col[kMaxCols], col4row[kMaxRows], remaining[kMaxCols];
  __shared__ unsigned char SR[kMaxRows], SC[kMaxCols];
  __shared__ float Cs[kMaxStaged];

  const Problem pr = probs[blockIdx.x];
  const int lane = threadIdx.x;
  const int Q = (int)pr.Q, n_ = (int)pr.n;
  const int K = n_ * (int)pr.tile;                     // columns of the (Q x K) matrix handed to the solver
  const bool transpose = K < Q;                        // scipy: solve the transposed problem when nc < nr
  const int nr = transpose ? K : Q;
  const int nc = transpose ? Q : K;
  if (nr == 0 || nc == 0) return;
  // The Q x n distinct costs of the problem are staged in LDS once (the 4x-tiled matrix repeats them): the column scan of
  // every Dijkstra step otherwise waits for global loads in the middle of a chain of ~40 dependent steps per problem.
  const bool staged = Q * n_ <= kMaxStaged;
  if (staged) {
    for (int e = lane; e < Q * n_; e += 64) Cs[e] = C[pr.base + (int64_t)(e / n_) * pr.ld + (e % n_)];
    __syncthreads();
  }
  // cost(i, j) of the oriented problem
  auto cost = [&](int i, int j) -> double {
    const int q = transpose ? j : i, k = transpose ? i : j;
    return staged ? (double)Cs[q * n_ + (k % n_)] : (double)C[pr.base + (int64_t)q * pr.ld + (k % n_)];
  };
  // validity (scipy raises on NaN / -inf)
  int bad = 0;
  for (int e = lane; e < nr * nc; e += 64) {
    const double c = cost(e / nc, e % nc);
    if (c != c || c == -INFINITY) bad = 1;
  }
  // On invalid / infeasible input scipy raises; here the status flag carries that to the host (LayerMatch.check) and
  // the outputs get a harmless in-range assignment (query i <-> target i mod n), because consumers on the device
  // (criterion, caption gather) index with them before anyone can look at the flag.
  auto fail_safe = [&]() {
    if (lane == 0) atomicExch(status, 1);
    int64_t *ro_ = rows_out + pr.out_off, *co_ = cols_out + pr.out_off;
    for (int r = lane; r < nr; r += 64) { ro_[r] = r; co_[r] = r % n_; }
  };
  if (wave_max_i(bad)) {
    fail_safe();
    return;
  }
  for (int i = lane; i < nr; i += 64) { u[i] = 0.0; col4row[i] = -1; }
  for (int j = lane; j < nc; j += 64) { v[j] = 0.0; path[j] = -1; row4col[j] = -1; }
  __syncthreads();

  for (int cur = 0; cur < nr; ++cur) {
    // ---- augmenting path from row `cur` -------------------------------------------------------------------
    double min_val = 0.0;
    int num_remaining = nc;
    for (int it = lane; it < nc; it += 64) { remaining[it] = nc - it - 1; SC[it] = 0; spc[it] = INFINITY; }
    for (int i = lane; i < nr; i += 64) SR[i] = 0;
    __syncthreads();
    int sink = -1, i = cur;
    while (sink == -1) {
      if (lane == 0) SR[i] = 1;
      const double ui = u[i];
      double local_min = INFINITY;
      for (int it = lane; it < num_remaining; it += 64) {
        const int j = remaining[it];
        const double r = min_val + cost(i, j) - ui - v[j];
        if (r < spc[j]) { path[j] = i; spc[j] = r; }
        local_min = fmin(local_min, spc[j]);
      }
      const double lowest = wave_min(local_min);
      if (lowest == INFINITY) {                          // infeasible
        fail_safe();
        return;
      }
      int last_unassigned = -1, first_any = 0x7fffffff;
      for (int it = lane; it < num_remaining; it += 64) {
        const int j = remaining[it];
        if (spc[j] == lowest) {
          first_any = min(first_any, it);
          if (row4col[j] == -1) last_unassigned = max(last_unassigned, it);
        }
      }
      last_unassigned = wave_max_i(last_unassigned);
      first_any = wave_min_i(first_any);
      const int index = last_unassigned >= 0 ? last_unassigned : first_any;
      min_val = lowest;
      const int j = remaining[index];
      if (row4col[j] == -1) sink = j; else i = row4col[j];
      __syncthreads();                                   // all lanes have read remaining[] / row4col[] for this round
      if (lane == 0) { SC[j] = 1; remaining[index] = remaining[num_remaining - 1]; }
      --num_remaining;
      __syncthreads();
    }
    // ---- dual updates (same expressions as the sequential solver) ---------------------------------------------
    for (int r = lane; r < nr; r += 64) {
      if (r == cur) u[r] += min_val;
      else if (SR[r]) u[r] += min_val - spc[col4row[r]];
    }
    for (int j = lane; j < nc; j += 64)
      if (SC[j]) v[j] -= min_val - spc[j];
    __syncthreads();
    if (lane == 0) {                                     // augment along the path (short, sequential)
      int j = sink;
      while (true) {
        const int r = path[j];
        row4col[j] = r;
        const int t_ = col4row[r];
        col4row[r] = j;
        j = t_;
        if (r == cur) break;
      }
    }
    __syncthreads();
  }

  // ---- output in scipy's order: row indices ascending ----------------------------------------------------------
  int64_t *ro = rows_out + pr.out_off, *co = cols_out + pr.out_off;
  if (!transpose) {
    for (int r = lane; r < nr; r += 64) { ro[r] = r; co[r] = col4row[r] % n_; }
  } else {
    // solved on the transpose: problem row r (a target copy) got query col4row[r]; emit sorted by query id
    for (int r = lane; r < nr; r += 64) {
      const int q = col4row[r];
      int rank = 0;
      for (int s = 0; s < nr; ++s) rank += (col4row[s] < q) ? 1 : 0;
      ro[rank] = q;
      co[rank] = r % n_;
    }
  }
}

}  // namespace

extern "C" int gvl_lsap_batch_device_f32(const float *C, const int64_t *problems, int n_problems, int max_rows,
                                         int max_cols, int64_t *rows_out, int64_t *cols_out, int *status,
                                         void *stream) {
  if (n_problems < 0) return fail(GVL_EINVAL, "gvl_lsap_batch_device_f32: bad problem count");
  if (n_problems == 0) return 0;
  if (!C || !problems || !rows_out || !cols_out || !status)
    return fail(GVL_EINVAL, "gvl_lsap_batch_device_f32: null pointer");
  if (max_rows > kMaxRows || max_cols > kMaxCols)
    return fail(GVL_EINVAL, "gvl_lsap_batch_device_f32: problem %dx%d exceeds the on-chip limit %dx%d", max_rows,
                max_cols, kMaxRows, kMaxCols);
  static_assert(sizeof(Problem) == 8 * sizeof(int64_t), "descriptor = 8 int64");
  return gvl::launch(GVL_PROF_LSAP, n_problems, 0, "k_lsap", k_lsap, dim3(n_problems), dim3(64), 0,
                     (hipStream_t)stream, C, reinterpret_cast<const Problem *>(problems), rows_out, cols_out, status);
}
