// micro-benchmark (round 5): the training GEMM kernels of gvl_train_gemm.hip as a standalone program (no Python), so that
// timing-only ablation builds (-DGVL_WG_NO_MFMA ...) of the SAME source can be run side by side in one gpurun call.
//   usage: tgemm_bench wgrad R N K [iters]
#include "../../gvl_amd/csrc/gvl_train_gemm.hip"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

static float *dev_rand(size_t n, float scale) {
  std::vector<float> h(n);
  for (size_t i = 0; i < n; ++i) h[i] = scale * ((float)rand() / RAND_MAX * 2.f - 1.f);
  float *d;
  hipMalloc(&d, n * 4);
  hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice);
  return d;
}

int main(int argc, char **argv) {
  if (argc < 5) return 1;
  const int R = atoi(argv[2]), N = atoi(argv[3]), K = atoi(argv[4]), iters = argc > 5 ? atoi(argv[5]) : 30;
  float *dy = dev_rand((size_t)R * N, 1e-3f), *x = dev_rand((size_t)R * K, 1.f);
  std::vector<float> one(R, 1.f);
  float *am_dy, *am_x, *gw, *gb;
  hipMalloc(&am_dy, R * 4); hipMalloc(&am_x, R * 4);
  hipMemcpy(am_x, one.data(), R * 4, hipMemcpyHostToDevice);
  for (auto &v : one) v = 1e-3f;
  hipMemcpy(am_dy, one.data(), R * 4, hipMemcpyHostToDevice);
  hipMalloc(&gw, (size_t)N * K * 4); hipMalloc(&gb, N * 4);
  const size_t wsb = gvl_wgrad_workspace_bytes(R, N, K);
  void *ws; hipMalloc(&ws, wsb + 16);
  hipStream_t st; hipStreamCreate(&st);
#ifdef GVL_WG_STAMPS
  hipMalloc(&g_wg_stamps, 4096 * 8 * 8);
  hipMemset(g_wg_stamps, 0, 4096 * 8 * 8);
#endif
  std::vector<float> us;
  for (int it = 0; it < iters + 5; ++it) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipEventRecord(a, st);
    int rc = gvl_wgrad_f16x3_f32(dy, N, am_dy, R, x, K, am_x, R, R, N, K, gw, gb, 0, ws, wsb, st);
    hipEventRecord(b, st);
    hipEventSynchronize(b);
    if (rc) { printf("rc %d %s\n", rc, gvl::g_err); return 2; }
    float ms; hipEventElapsedTime(&ms, a, b);
    if (it >= 5) us.push_back(ms * 1e3f);
  }
  std::sort(us.begin(), us.end());
  const WgPlan pl = wgrad_plan(R, N, K);
  printf("wgrad R=%d N=%d K=%d  SK=%d rows/split=%d  median %7.2f us  min %7.2f us (event-to-event, incl. reduce launch)\n", R, N, K, pl.SK,
         pl.rows_per_split, us[us.size() / 2], us[0]);
#ifdef GVL_WG_STAMPS
  {
    std::vector<unsigned long long> h(4096 * 8);
    hipMemcpy(h.data(), g_wg_stamps, h.size() * 8, hipMemcpyDeviceToHost);
    std::vector<double> cyc[3], us3[3], begin;
    unsigned long long t0 = ~0ull, t1 = 0;
    for (int b = 0; b < 4096; ++b) {
      if (!h[b * 8 + 7]) continue;
      for (int i = 0; i < 3; ++i) {
        cyc[i].push_back((double)(h[b * 8 + 2 * (i + 1)] - h[b * 8 + 2 * i]));
        us3[i].push_back((double)(h[b * 8 + 2 * (i + 1) + 1] - h[b * 8 + 2 * i + 1]) * 0.01);
      }
      t0 = std::min(t0, h[b * 8 + 1]); t1 = std::max(t1, h[b * 8 + 7]);
      begin.push_back((double)h[b * 8 + 1]);
    }
    const char *nm[3] = {"prologue (amax, first loads, stage 0)", "stage loop", "epilogue (partial tile store)"};
    for (int i = 0; i < 3; ++i) {
      std::sort(cyc[i].begin(), cyc[i].end()); std::sort(us3[i].begin(), us3[i].end());
      printf("   %-40s median %8.0f cycles %6.2f us   max %6.2f us\n", nm[i], cyc[i][cyc[i].size() / 2], us3[i][us3[i].size() / 2], us3[i].back());
    }
    std::sort(begin.begin(), begin.end());
    printf("   %zu workgroups stamped; first start -> last end %.2f us; start skew (last start - first start) %.2f us; loop clock %.0f MHz\n",
           begin.size(), (double)(t1 - t0) * 0.01, (begin.back() - begin.front()) * 0.01, cyc[1][cyc[1].size() / 2] / us3[1][us3[1].size() / 2]);
  }
#endif
  return 0;
}
