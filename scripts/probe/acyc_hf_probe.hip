// k_acyc_hf (two-piece f16, block-scaled) against k_acyc_bf (three-piece bf16) and a double CPU reference; timing of both at the headline grid.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/probe/acyc_hf_probe.hip -o scripts/probe/acyc_hf_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <cmath>
#include "../../dibs_amd/csrc/kernels_acyc.h"
#include "../../dibs_amd/csrc/kernels_acyc_bf16.h"
#include "../../dibs_amd/csrc/kernels_acyc_f16.h"

static void mm(const std::vector<double>& a, const std::vector<double>& b, std::vector<double>& c, int d) {
  for (int i = 0; i < d; ++i) for (int j = 0; j < d; ++j) { double s = 0; for (int k = 0; k < d; ++k) s += a[i*d+k]*b[k*d+j]; c[i*d+j] = s; }
}

static float* g_eas = nullptr;  // exp(-alpha scores), as k_edge_scores hands it to k_acyc_hf
template <typename K>
static void launch(K kern, size_t lds, int nblk, int Mloc, const float* ds, float* dp, Key2 carry, int d, int Sa, int cpb, float alpha) {
  hipLaunchKernelGGL(kern, dim3(nblk, (Mloc + 7) & ~7), dim3(256), lds, 0, ds, dp, carry, 0, Mloc, Mloc, d, Sa, cpb, alpha, 1.0f, 0, 0, nblk);
}
template <typename K>
static void launch_hf(K kern, size_t lds, int nblk, int Mloc, const float* ds, float* dp, Key2 carry, int d, int Sa, int cpb, float alpha) {
  hipLaunchKernelGGL(kern, dim3(nblk, (Mloc + 7) & ~7), dim3(256), lds, 0, ds, (const float*)g_eas, dp, carry, 0, Mloc, Mloc, d, Sa, cpb, alpha, 1.0f, 0, 0, nblk);
}

static int g_only = -1;  // >= 0: time only this variant (PMC runs)
static void run(int d, int Mloc, int Sa, float alpha, float sscale, int cpb, bool timing) {
  const int nblk = (Sa / 2 + cpb - 1) / cpb;
  std::vector<float> scores((size_t)Mloc * d * d);
  srand(7);
  for (auto& s : scores) s = sscale * 4.0f * ((float)rand() / RAND_MAX - 0.5f);
  const size_t np = (size_t)Mloc * nblk * d * d;
  float *ds, *dp[3];
  hipMalloc(&ds, scores.size() * 4);
  for (int v = 0; v < 3; ++v) { hipMalloc(&dp[v], np * 4); hipMemset(dp[v], 0, np * 4); }
  hipMemcpy(ds, scores.data(), scores.size() * 4, hipMemcpyHostToDevice);
  {
    std::vector<float> ev(scores.size());
    for (size_t i = 0; i < ev.size(); ++i) ev[i] = (float)exp(-(double)(alpha * scores[i]));
    hipMalloc(&g_eas, ev.size() * 4);
    hipMemcpy(g_eas, ev.data(), ev.size() * 4, hipMemcpyHostToDevice);
  }
  const size_t lds_bf = 2 * ABF_IMG_BYTES, lds_hf = AHF_LDS_BYTES;
  hipFuncSetAttribute((const void*)k_acyc_bf<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bf);
  hipFuncSetAttribute((const void*)k_acyc_bf<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bf);
  Key2 carry{123u, 456u};
  auto go = [&](int v) {
    if (v == 0) { if (d > 48) launch(k_acyc_bf<true>, lds_bf, nblk, Mloc, ds, dp[0], carry, d, Sa, cpb, alpha); else launch(k_acyc_bf<false>, lds_bf, nblk, Mloc, ds, dp[0], carry, d, Sa, cpb, alpha); }
    if (v == 1) { if (d > 48) launch_hf(k_acyc_hf<true, 3>, lds_hf, nblk, Mloc, ds, dp[1], carry, d, Sa, cpb, alpha); else launch_hf(k_acyc_hf<false, 3>, lds_hf, nblk, Mloc, ds, dp[1], carry, d, Sa, cpb, alpha); }
    if (v == 2) { if (d > 48) launch_hf(k_acyc_hf<true, 4>, lds_hf, nblk, Mloc, ds, dp[2], carry, d, Sa, cpb, alpha); else launch_hf(k_acyc_hf<false, 4>, lds_hf, nblk, Mloc, ds, dp[2], carry, d, Sa, cpb, alpha); }
  };
  for (int v = 0; v < 3; ++v) go(v);
  hipError_t e2 = hipDeviceSynchronize();
  std::vector<float> p[3];
  for (int v = 0; v < 3; ++v) { p[v].resize(np); hipMemcpy(p[v].data(), dp[v], np * 4, hipMemcpyDeviceToHost); }
  // CPU reference for particle 0: sum over all chains
  Key2 km = rng_split_row(carry, Mloc + 1, 1, 0);
  std::vector<double> acc(d*d, 0.0), M(d*d), P(d*d), T(d*d), G(d*d);
  for (int sa = 0; sa < Sa; ++sa) {
    for (int i = 0; i < d; ++i) for (int j = 0; j < d; ++j) {
      uint32_t bits = rng_bits_at(km, (uint64_t)Sa*d*d, (uint64_t)sa*d*d + i*d + j, 0);
      float x = rng_uniform(bits, 1.1920929e-07f, 1.0f);
      double eps = log((double)x / (1.0 - (double)x));
      double g = i == j ? 0.0 : 1.0 / (1.0 + exp(-(eps + (double)alpha * scores[i*d+j])));
      G[i*d+j] = g; M[i*d+j] = (i == j) + g / d;
    }
    P = M; int ex = d - 1; int hb = 31 - __builtin_clz(ex);
    for (int b = hb - 1; b >= 0; --b) { mm(P, P, T, d); P = T; if ((ex >> b) & 1) { mm(P, M, T, d); P = T; } }
    for (int i = 0; i < d; ++i) for (int j = 0; j < d; ++j) if (i != j) acc[i*d+j] += alpha * P[j*d+i] * G[i*d+j] * (1 - G[i*d+j]);
  }
  double maxref = 0, e[3] = {0, 0, 0}, eel[3] = {0, 0, 0}; int nan[3] = {0, 0, 0};
  for (int i = 0; i < d*d; ++i) maxref = fmax(maxref, fabs(acc[i]));
  for (int v = 0; v < 3; ++v)
    for (int i = 0; i < d*d; ++i) {
      double s = 0;
      for (int b = 0; b < nblk; ++b) s += p[v][(size_t)b*d*d + i];
      if (!std::isfinite(s)) ++nan[v];
      e[v] = fmax(e[v], fabs(acc[i] - s));
      if (fabs(acc[i]) > 1e-30 * maxref && fabs(acc[i]) > 0) eel[v] = fmax(eel[v], fabs(acc[i] - s) / fabs(acc[i]));
    }
  {  // signed statistics of the relative error on the entries that matter (|ref| > 1e-3 max): a bias shows as mean != 0
    for (int v = 0; v < 2; ++v) {
      double sum = 0, sum2 = 0; int n = 0;
      for (int i = 0; i < d*d; ++i) {
        if (fabs(acc[i]) < 1e-3 * maxref) continue;
        double s = 0;
        for (int b = 0; b < nblk; ++b) s += p[v][(size_t)b*d*d + i];
        const double r = (s - acc[i]) / acc[i];
        sum += r; sum2 += r * r; ++n;
      }
      if (n) printf("   %s signed rel err: mean %+.3e  sd %.3e  (n=%d)\n", v ? "f16x2 " : "bf16x3", sum / n, sqrt(sum2 / n - (sum / n) * (sum / n)), n);
    }
  }
  printf("d=%d Mloc=%d Sa=%d alpha=%g sscale=%g cpb=%d sync=%s max|ref|=%.3g\n   vs double, relative to max (elementwise): bf16x3 %.3g (%.3g) nan %d | f16x2 wpe3 %.3g (%.3g) nan %d | f16x2 wpe4 %.3g (%.3g) nan %d\n",
         d, Mloc, Sa, alpha, sscale, cpb, hipGetErrorName(e2), maxref, e[0] / maxref, eel[0], nan[0], e[1] / maxref, eel[1], nan[1], e[2] / maxref, eel[2], nan[2]);
  if (timing) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const char* names[3] = {"k_acyc_bf       ", "k_acyc_hf wpe 3 ", "k_acyc_hf wpe 4 "};
    for (int which = 0; which < 3; ++which) {
      if (g_only >= 0 && which != g_only) continue;
      float best = 1e9f;
      for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(a, 0);
        for (int it = 0; it < 10; ++it) go(which);
        hipEventRecord(b, 0); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b); best = fminf(best, ms / 10);
      }
      printf("   %s: %.1f us per launch\n", names[which], best * 1e3f);
    }
  }
  hipFree(ds); hipFree(g_eas); for (int v = 0; v < 3; ++v) hipFree(dp[v]);
}

int main(int argc, char** argv) {
  if (argc > 1) {  // acyc_hf_probe <variant 0..2> [d] [Mloc] [cpb]: headline grid only
    g_only = atoi(argv[1]);
    run(argc > 2 ? atoi(argv[2]) : 50, argc > 3 ? atoi(argv[3]) : 128, 32, 0.05f, 1.f, argc > 4 ? atoi(argv[4]) : 1, true);
    return 0;
  }
  hipFuncSetAttribute((const void*)k_acyc_hf<true, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)AHF_LDS_BYTES);
  run(50, 2, 4, 0.0f, 1.f, 1, false);
  run(50, 2, 4, 0.5f, 1.f, 1, false);
  run(50, 2, 4, 20.f, 1.f, 1, false);
  run(50, 2, 4, 300.f, 1.f, 1, false);
  run(50, 2, 4, 300.f, 0.02f, 1, false);
  run(50, 2, 4, 2000.f, 1.f, 1, false);
  run(64, 2, 4, 0.2f, 1.f, 1, false);
  run(33, 1, 2, 0.2f, 1.f, 1, false);
  run(48, 1, 2, 3.0f, 1.f, 1, false);
  run(49, 1, 2, 3.0f, 1.f, 1, false);
  run(60, 1, 2, 2.0f, 1.f, 1, false);
  run(50, 3, 8, 1.0f, 1.f, 2, false);
  run(50, 128, 32, 0.05f, 1.f, 1, true);
  run(50, 128, 32, 10.f, 1.f, 1, true);
  run(50, 128, 32, 0.05f, 1.f, 2, true);
  run(40, 128, 32, 0.05f, 1.f, 1, true);
  run(64, 128, 32, 0.05f, 1.f, 1, true);
  run(50, 16, 32, 0.05f, 1.f, 1, true);
  run(50, 1024, 32, 0.05f, 1.f, 1, true);
  return 0;
}
