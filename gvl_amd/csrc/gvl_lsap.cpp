// gvl_lsap.cpp -- rectangular linear-sum-assignment for the Hungarian matcher's index path (host code).
//
// The reference calls scipy.optimize.linear_sum_assignment per video (pdvc/matcher.py:124,126); scipy is a
// third-party dependency that is not vendored in the reference (requirement.txt:11, unpinned; this image ships
// scipy 1.15.3).  Its published algorithm is D. F. Crouse, "On implementing 2D rectangular assignment
// algorithms", IEEE TAES 52(4), 2016 -- shortest augmenting paths with dual variables, no initialisation phase,
// rows processed in order, the transposed problem solved when there are more rows than columns, and, among
// equal shortest-path costs, preference for a column that is still unassigned.  The tie-breaking below follows
// that description step for step because the matcher's indices must be bit-identical (tests/golden/lsap_cases.npz
// holds scipy's answers for tie / near-tie matrices).
//
// The batch entry point restates matcher.py:120-131: per video the (Q x n_i) column block of C and the same
// block tiled m2o_rate times, all videos solved concurrently on host threads from ONE device->host copy.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <limits>
#include <numeric>
#include <thread>
#include <vector>

#include "gvl_msda.h"

namespace {

// one augmentation from row `i`; returns the sink column or -1 (infeasible)
int64_t augment(int64_t nc, const double *cost, std::vector<double> &u, std::vector<double> &v,
                std::vector<int64_t> &path, const std::vector<int64_t> &row4col, std::vector<double> &spc, int64_t i,
                std::vector<char> &SR, std::vector<char> &SC, std::vector<int64_t> &remaining, double *p_min) {
  double min_val = 0;
  int64_t num_remaining = nc;
  for (int64_t it = 0; it < nc; ++it) remaining[it] = nc - it - 1;   // columns are scanned in reverse order
  std::fill(SR.begin(), SR.end(), 0);
  std::fill(SC.begin(), SC.end(), 0);
  std::fill(spc.begin(), spc.end(), std::numeric_limits<double>::infinity());
  int64_t sink = -1;
  while (sink == -1) {
    int64_t index = -1;
    double lowest = std::numeric_limits<double>::infinity();
    SR[i] = 1;
    for (int64_t it = 0; it < num_remaining; ++it) {
      const int64_t j = remaining[it];
      const double r = min_val + cost[i * nc + j] - u[i] - v[j];
      if (r < spc[j]) {
        path[j] = i;
        spc[j] = r;
      }
      // ties: prefer a column that yields a new sink
      if (spc[j] < lowest || (spc[j] == lowest && row4col[j] == -1)) {
        lowest = spc[j];
        index = it;
      }
    }
    min_val = lowest;
    if (min_val == std::numeric_limits<double>::infinity()) return -1;
    const int64_t j = remaining[index];
    if (row4col[j] == -1)
      sink = j;
    else
      i = row4col[j];
    SC[j] = 1;
    remaining[index] = remaining[--num_remaining];
  }
  *p_min = min_val;
  return sink;
}

// cost: nr x nc row-major doubles.  a,b: outputs of length min(nr,nc).  returns 0, or -1 infeasible / invalid.
int solve(int64_t nr, int64_t nc, const double *cost_in, int64_t *a, int64_t *b) {
  if (nr == 0 || nc == 0) return 0;
  const bool transpose = nc < nr;
  std::vector<double> tmp;
  const double *cost = cost_in;
  if (transpose) {
    tmp.resize((size_t)nr * nc);
    for (int64_t i = 0; i < nr; ++i)
      for (int64_t j = 0; j < nc; ++j) tmp[(size_t)j * nr + i] = cost_in[(size_t)i * nc + j];
    std::swap(nr, nc);
    cost = tmp.data();
  }
  for (int64_t k = 0; k < nr * nc; ++k)
    if (std::isnan(cost[k]) || cost[k] == -std::numeric_limits<double>::infinity()) return -1;
  std::vector<double> u(nr, 0), v(nc, 0), spc(nc);
  std::vector<int64_t> path(nc, -1), col4row(nr, -1), row4col(nc, -1), remaining(nc);
  std::vector<char> SR(nr), SC(nc);
  for (int64_t cur = 0; cur < nr; ++cur) {
    double min_val;
    const int64_t sink = augment(nc, cost, u, v, path, row4col, spc, cur, SR, SC, remaining, &min_val);
    if (sink < 0) return -1;
    u[cur] += min_val;
    for (int64_t i = 0; i < nr; ++i)
      if (SR[i] && i != cur) u[i] += min_val - spc[col4row[i]];
    for (int64_t j = 0; j < nc; ++j)
      if (SC[j]) v[j] -= min_val - spc[j];
    int64_t j = sink;
    while (true) {
      const int64_t i = path[j];
      row4col[j] = i;
      std::swap(col4row[i], j);
      if (i == cur) break;
    }
  }
  if (transpose) {
    std::vector<int64_t> order(nr);
    std::iota(order.begin(), order.end(), 0);
    std::sort(order.begin(), order.end(), [&](int64_t x, int64_t y) { return col4row[x] < col4row[y]; });
    for (int64_t k = 0; k < nr; ++k) {
      a[k] = col4row[order[k]];
      b[k] = order[k];
    }
  } else {
    for (int64_t i = 0; i < nr; ++i) {
      a[i] = i;
      b[i] = col4row[i];
    }
  }
  return 0;
}

}  // namespace

extern "C" {

int gvl_lsap_solve_f64(const double *cost, int64_t nr, int64_t nc, int64_t *row_ind, int64_t *col_ind) {
  if (nr < 0 || nc < 0 || (!cost && nr * nc > 0) || ((!row_ind || !col_ind) && nr > 0 && nc > 0)) return GVL_EINVAL;
  return solve(nr, nc, cost, row_ind, col_ind);
}

int gvl_lsap_solve_f32(const float *cost, int64_t nr, int64_t nc, int64_t *row_ind, int64_t *col_ind) {
  if (nr < 0 || nc < 0 || (!cost && nr * nc > 0)) return GVL_EINVAL;
  std::vector<double> c((size_t)nr * nc);
  for (size_t k = 0; k < c.size(); ++k) c[k] = (double)cost[k];   // scipy promotes float32 input to float64
  return solve(nr, nc, c.data(), row_ind, col_ind);
}

int gvl_hungarian_batch_f32(const float *C, int B, int Q, int G, const int *sizes, int m2o_rate, int64_t *idx_rows,
                            int64_t *idx_cols, int64_t *rl_rows, int64_t *rl_cols, int num_threads) {
  if (B < 0 || Q < 0 || G < 0 || m2o_rate < 1 || (!sizes && B > 0)) return GVL_EINVAL;
  std::vector<int64_t> coff(B + 1, 0), ioff(B + 1, 0), roff(B + 1, 0);
  for (int i = 0; i < B; ++i) {
    if (sizes[i] < 0) return GVL_EINVAL;
    coff[i + 1] = coff[i] + sizes[i];
    ioff[i + 1] = ioff[i] + std::min<int64_t>(Q, sizes[i]);
    roff[i + 1] = roff[i] + std::min<int64_t>(Q, (int64_t)sizes[i] * m2o_rate);
  }
  if (coff[B] != G) return GVL_EINVAL;
  std::vector<int> status(B, 0);
  auto work = [&](int i) {
    const int n = sizes[i];
    if (n == 0 || Q == 0) return;
    std::vector<double> c((size_t)Q * n), ct((size_t)Q * n * m2o_rate);
    for (int q = 0; q < Q; ++q)
      for (int k = 0; k < n; ++k) {
        const double x = (double)C[((size_t)i * Q + q) * G + coff[i] + k];
        c[(size_t)q * n + k] = x;
        for (int r = 0; r < m2o_rate; ++r) ct[(size_t)q * n * m2o_rate + (size_t)r * n + k] = x;   // matcher.py:126
      }
    int s1 = solve(Q, n, c.data(), idx_rows + ioff[i], idx_cols + ioff[i]);
    int s2 = solve(Q, (int64_t)n * m2o_rate, ct.data(), rl_rows + roff[i], rl_cols + roff[i]);
    for (int64_t k = roff[i]; k < roff[i + 1]; ++k) rl_cols[k] %= n;                                  // matcher.py:127
    status[i] = s1 ? s1 : s2;
  };
  int nt = num_threads > 0 ? num_threads : (int)std::thread::hardware_concurrency();
  nt = std::max(1, std::min(nt, B));
  // spawning threads costs ~30 us each: only worth it when the solves are not tiny (GVL: Q = 300, n_i <= 30 is
  // ~10 us per video, so a whole batch is cheaper on the calling thread)
  if (num_threads <= 0 && (int64_t)B * Q * G * m2o_rate < (int64_t)4 << 20) nt = 1;
  if (nt <= 1) {
    for (int i = 0; i < B; ++i) work(i);
  } else {
    std::vector<std::thread> pool;
    for (int t = 0; t < nt; ++t)
      pool.emplace_back([&, t] { for (int i = t; i < B; i += nt) work(i); });
    for (auto &th : pool) th.join();
  }
  for (int i = 0; i < B; ++i)
    if (status[i]) return status[i];
  return 0;
}

}  // extern "C"
