// timing harness for the ablation builds of k_gemm_f16x3_w8 (tools/gemm16_ablate.sh): which part of a K stage costs what
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "../gvl_amd/csrc/gvl_gemm16.hip"

int main() {
  const int R = 4800, N = 8518, K = 512;
  std::vector<uint16_t> h((size_t)N * K);
  for (auto &v : h) v = 0x3000 + (rand() & 0x0fff) + ((rand() & 1) << 15);      // random fp16 in +-[0.125, 0.25)
  void *ah, *al, *bh, *bl; float *as, *bs, *out, *bias;
  hipMalloc(&ah, (size_t)R * K * 2); hipMalloc(&al, (size_t)R * K * 2); hipMalloc(&bh, (size_t)N * K * 2); hipMalloc(&bl, (size_t)N * K * 2);
  hipMalloc(&as, R * 4); hipMalloc(&bs, N * 4); hipMalloc(&bias, N * 4); hipMalloc(&out, (size_t)R * N * 4);
  hipMemcpy(ah, h.data(), (size_t)R * K * 2, hipMemcpyHostToDevice); hipMemcpy(al, h.data() + 1000, (size_t)R * K * 2, hipMemcpyHostToDevice);
  hipMemcpy(bh, h.data(), (size_t)N * K * 2, hipMemcpyHostToDevice); hipMemcpy(bl, h.data(), (size_t)N * K * 2, hipMemcpyHostToDevice);
  std::vector<float> ones(N, 1.f);
  hipMemcpy(as, ones.data(), R * 4, hipMemcpyHostToDevice); hipMemcpy(bs, ones.data(), N * 4, hipMemcpyHostToDevice); hipMemcpy(bias, ones.data(), N * 4, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int mode = 0; mode < 2; ++mode) {
    float ms = 0;
    for (int it = 0; it < 25; ++it) {
      if (it == 5) hipEventRecord(e0, 0);
      int rc = mode == 0 ? gvl_gemm_f16x3_f32(ah, al, as, R, bh, bl, bs, N, K, bias, out, N, 0)
                         : gvl_gemm_f16x3_argmax_f32(ah, al, as, R, bh, bl, bs, N, K, bias, out, 0);
      if (rc) { printf("rc %d %s\n", rc, gvl_last_error()); return 1; }
    }
    hipEventRecord(e1, 0); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
    printf("%s %s: %.1f us\n", VARIANT, mode ? "argmax" : "store", ms / 20 * 1e3);
  }
  return 0;
}
