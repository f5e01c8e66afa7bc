/* The drop-in boundary used from plain C: no Python, no torch -- only include/gvl_msda.h, libgvl_msda.so, the HIP runtime
 * for device memory, and the C oracle (oracle/msda_ref.c) as the checker.  Build + run: tests/test_gpu_c_abi.py.
 *   forward + backward of the op on a small temporal problem, compared with the oracle; error paths of the ABI. */
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "gvl_msda.h"

/* oracle/msda_ref.c */
int oracle_msda_fwd_f32(const float *value, const int64_t *shapes, const int64_t *lsi, const float *loc,
                        const float *attn, int B, int S, int M, int D, int L, int Q, int P, int pad, float *out);
int oracle_msda_bwd_f32(const float *value, const int64_t *shapes, const int64_t *lsi, const float *loc,
                        const float *attn, const float *gout, int B, int S, int M, int D, int L, int Q, int P, int pad,
                        float *gvalue, float *gloc, float *gattn);

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)

static float frand(unsigned *s) { *s = *s * 1664525u + 1013904223u; return (float)((*s >> 8) & 0xFFFFFF) / 16777216.0f; }
static void *dev_copy(const void *h, size_t n) { void *d = NULL; if (hipMalloc(&d, n) != hipSuccess) return NULL; hipMemcpy(d, h, n, hipMemcpyHostToDevice); return d; }
static double maxdiff(const float *a, const float *b, size_t n) { double m = 0; for (size_t i = 0; i < n; ++i) { double d = fabs((double)a[i] - b[i]); if (d > m) m = d; } return m; }

int main(void) {
  enum { B = 2, M = 8, D = 64, L = 4, P = 4, Q = 37 };
  const int64_t lens[L] = {20, 10, 5, 3};
  int64_t shapes[2 * L], lsi[L];
  int S = 0;
  for (int l = 0; l < L; ++l) { shapes[2 * l] = 1; shapes[2 * l + 1] = lens[l]; lsi[l] = S; S += (int)lens[l]; }
  const size_t nv = (size_t)B * S * M * D, nl = (size_t)B * Q * M * L * P * 2, na = nl / 2, no = (size_t)B * Q * M * D;
  float *value = malloc(4 * nv), *loc = malloc(4 * nl), *attn = malloc(4 * na), *gout = malloc(4 * no);
  unsigned seed = 7;
  for (size_t i = 0; i < nv; ++i) value[i] = 2.f * frand(&seed) - 1.f;
  for (size_t i = 0; i < nl; i += 2) { loc[i] = 1.5f * frand(&seed) - 0.25f; loc[i + 1] = 0.5f; }
  for (size_t i = 0; i < na; ++i) attn[i] = frand(&seed) / (L * P);
  for (size_t i = 0; i < no; ++i) gout[i] = 2.f * frand(&seed) - 1.f;

  if (gvl_msda_abi_version() != GVL_MSDA_ABI_VERSION) { fprintf(stderr, "ABI version mismatch\n"); return 1; }
  /* argument errors come back as codes + message, nothing is launched */
  if (gvl_msda_forward_f32(NULL, NULL, NULL, NULL, NULL, B, S, M, D, L, Q, P, 0, NULL, NULL, NULL, NULL) != GVL_EINVAL ||
      !strstr(gvl_last_error(), "null pointer")) { fprintf(stderr, "expected GVL_EINVAL\n"); return 1; }

  float *d_value = dev_copy(value, 4 * nv), *d_loc = dev_copy(loc, 4 * nl), *d_attn = dev_copy(attn, 4 * na),
        *d_gout = dev_copy(gout, 4 * no);
  int64_t *d_shapes = dev_copy(shapes, sizeof shapes), *d_lsi = dev_copy(lsi, sizeof lsi);
  float *d_out, *d_gv, *d_gl, *d_ga;
  CHECK(hipMalloc((void **)&d_out, 4 * no)); CHECK(hipMalloc((void **)&d_gv, 4 * nv));
  CHECK(hipMalloc((void **)&d_gl, 4 * nl)); CHECK(hipMalloc((void **)&d_ga, 4 * na));
  hipStream_t st; CHECK(hipStreamCreate(&st));

  for (int pad = 0; pad <= 1; ++pad) {
    int rc = gvl_msda_forward_f32(d_value, d_shapes, d_lsi, d_loc, d_attn, B, S, M, D, L, Q, P, pad, shapes, lsi, d_out, st);
    if (rc) { fprintf(stderr, "forward: %d %s\n", rc, gvl_last_error()); return 1; }
    if (gvl_msda_last_impl() != 2) { fprintf(stderr, "expected the temporal kernels (impl 2)\n"); return 1; }
    size_t ws_bytes = gvl_msda_backward_workspace_bytes(B, S, M, D, L, Q, P, 4, shapes);
    void *ws = NULL; if (ws_bytes) CHECK(hipMalloc(&ws, ws_bytes));
    rc = gvl_msda_backward_f32(d_value, d_shapes, d_lsi, d_loc, d_attn, d_gout, B, S, M, D, L, Q, P, pad, shapes, lsi, d_gv,
                               d_gl, d_ga, ws, ws_bytes, st);
    if (rc) { fprintf(stderr, "backward: %d %s\n", rc, gvl_last_error()); return 1; }
    CHECK(hipStreamSynchronize(st));
    float *out = malloc(4 * no), *gv = malloc(4 * nv), *gl = malloc(4 * nl), *ga = malloc(4 * na);
    float *r_out = malloc(4 * no), *r_gv = calloc(nv, 4), *r_gl = calloc(nl, 4), *r_ga = calloc(na, 4);
    CHECK(hipMemcpy(out, d_out, 4 * no, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(gv, d_gv, 4 * nv, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(gl, d_gl, 4 * nl, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(ga, d_ga, 4 * na, hipMemcpyDeviceToHost));
    oracle_msda_fwd_f32(value, shapes, lsi, loc, attn, B, S, M, D, L, Q, P, pad, r_out);
    oracle_msda_bwd_f32(value, shapes, lsi, loc, attn, gout, B, S, M, D, L, Q, P, pad, r_gv, r_gl, r_ga);
    const double e0 = maxdiff(out, r_out, no), e1 = maxdiff(gv, r_gv, nv), e2 = maxdiff(gl, r_gl, nl), e3 = maxdiff(ga, r_ga, na);
    printf("pad=%d max|diff| out %.2e gvalue %.2e gloc %.2e gattn %.2e\n", pad, e0, e1, e2, e3);
    if (e0 > 1e-4 || e1 > 1e-4 || e2 > 2e-3 || e3 > 1e-4) { fprintf(stderr, "mismatch vs the oracle\n"); return 1; }
    if (ws) CHECK(hipFree(ws));
    free(out); free(gv); free(gl); free(ga); free(r_out); free(r_gv); free(r_gl); free(r_ga);
  }
  /* workspace too small -> GVL_ENOSPC */
  if (gvl_msda_backward_workspace_bytes(B, S, M, D, L, Q, P, 4, shapes) > 0 &&
      gvl_msda_backward_f32(d_value, d_shapes, d_lsi, d_loc, d_attn, d_gout, B, S, M, D, L, Q, P, 0, shapes, lsi, d_gv, d_gl,
                            d_ga, NULL, 0, st) != GVL_ENOSPC) { fprintf(stderr, "expected GVL_ENOSPC\n"); return 1; }
  /* the captioner's token-loop products from plain C: split both operands, multiply, argmax + log-sum-exp without logits */
  {
    enum { GR = 70, GK = 64, GV = 301 };
    float *x = malloc(4 * GR * GK), *w = malloc(4 * GV * GK), *bias = malloc(4 * GV);
    for (int i = 0; i < GR * GK; ++i) x[i] = 2.0f * frand(&seed) - 1.0f;
    for (int i = 0; i < GV * GK; ++i) w[i] = 0.2f * frand(&seed) - 0.1f;
    for (int i = 0; i < GV; ++i) bias[i] = frand(&seed) - 0.5f;
    float *d_x = dev_copy(x, 4 * GR * GK), *d_w = dev_copy(w, 4 * GV * GK), *d_b = dev_copy(bias, 4 * GV);
    void *xh, *xl, *wh, *wl; float *xs, *ws2, *d_o, *d_part, *d_lp; int64_t *d_tok;
    CHECK(hipMalloc(&xh, 2 * GR * GK)); CHECK(hipMalloc(&xl, 2 * GR * GK)); CHECK(hipMalloc((void **)&xs, 4 * GR));
    CHECK(hipMalloc(&wh, 2 * GV * GK)); CHECK(hipMalloc(&wl, 2 * GV * GK)); CHECK(hipMalloc((void **)&ws2, 4 * GV));
    CHECK(hipMalloc((void **)&d_o, 4 * GR * GV));
    const int chunks = gvl_gemm_f16x3_argmax_chunks(GV);
    CHECK(hipMalloc((void **)&d_part, 16 * (size_t)chunks * GR)); CHECK(hipMalloc((void **)&d_lp, 4 * GR));
    CHECK(hipMalloc((void **)&d_tok, 8 * GR));
    int rc = gvl_split_rows_f16(d_x, GR, GK, xh, xl, xs, st);
    if (!rc) rc = gvl_split_rows_f16(d_w, GV, GK, wh, wl, ws2, st);
    if (!rc) rc = gvl_gemm_f16x3_f32(xh, xl, xs, GR, wh, wl, ws2, GV, GK, d_b, d_o, GV, st);
    if (!rc) rc = gvl_gemm_f16x3_argmax_f32(xh, xl, xs, GR, wh, wl, ws2, GV, GK, d_b, d_part, st);
    if (!rc) rc = gvl_greedy_step_partials_f32(d_part, GR, GV, 1, d_tok, d_lp, NULL, NULL, NULL, 0, st);
    if (rc) { fprintf(stderr, "gemm_f16x3: %d %s\n", rc, gvl_last_error()); return 1; }
    CHECK(hipStreamSynchronize(st));
    float *o = malloc(4 * GR * GV), *lp = malloc(4 * GR); int64_t *tok = malloc(8 * GR);
    CHECK(hipMemcpy(o, d_o, 4 * GR * GV, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(lp, d_lp, 4 * GR, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(tok, d_tok, 8 * GR, hipMemcpyDeviceToHost));
    double emax = 0, elp = 0; int bad = 0;
    for (int r = 0; r < GR; ++r) {
      double best = -1e300, sum = 0; int arg = -1;
      for (int n = 0; n < GV; ++n) {
        double acc = bias[n];
        for (int k = 0; k < GK; ++k) acc += (double)x[r * GK + k] * w[n * GK + k];
        const double d = fabs(acc - o[r * GV + n]); if (d > emax) emax = d;
        if (acc > best) { best = acc; arg = n; }
      }
      for (int n = 0; n < GV; ++n) {
        double acc = bias[n];
        for (int k = 0; k < GK; ++k) acc += (double)x[r * GK + k] * w[n * GK + k];
        sum += exp(acc - best);
      }
      if (tok[r] != arg) ++bad;
      const double d = fabs(-log(sum) - lp[r]); if (d > elp) elp = d;
    }
    printf("gemm_f16x3 max|diff| vs double %.2e, fused argmax: %d wrong tokens, log-prob %.2e\n", emax, bad, elp);
    if (emax > 2e-6 || bad || elp > 2e-6) { fprintf(stderr, "gemm_f16x3 mismatch\n"); return 1; }
    if (gvl_gemm_f16x3_f32(xh, xl, xs, GR, wh, wl, ws2, GV, 48, d_b, d_o, GV, st) != GVL_EINVAL) {
      fprintf(stderr, "expected GVL_EINVAL for K %% 32 != 0\n"); return 1; }
  }
  printf("C ABI OK\n");
  return 0;
}
