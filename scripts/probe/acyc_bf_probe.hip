// k_acyc_bf (bf16 x 3 split MFMA) against k_acyc (f32 MFMA) and a double CPU reference; timing of both at the headline grid.
// Also prints the lane mapping of ds_read_b64_tr_b16 that kernels_acyc_bf16.h relies on.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <cmath>
#include "../../dibs_amd/csrc/kernels_acyc.h"
#include "../../dibs_amd/csrc/kernels_acyc_bf16.h"

__global__ void k_tr(short* out) {
  __shared__ __attribute__((aligned(16))) short sm[4 * 64];
  const int lane = threadIdx.x;
  for (int i = lane; i < 256; i += 64) sm[i] = (short)i;   // group g block = sm[64 g ..], [4 rows][16 cols]
  __syncthreads();
  typedef __attribute__((address_space(3))) abf_s16x4 lds_s16x4;
  const int g = lane >> 4, q = lane & 15;
  const abf_s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(sm + 64 * g + (q >> 2) * 16 + (q & 3) * 4));
  for (int j = 0; j < 4; ++j) out[lane * 4 + j] = v[j];
}

static void mm(const std::vector<double>& a, const std::vector<double>& b, std::vector<double>& c, int d) {
  for (int i = 0; i < d; ++i) for (int j = 0; j < d; ++j) { double s = 0; for (int k = 0; k < d; ++k) s += a[i*d+k]*b[k*d+j]; c[i*d+j] = s; }
}

static void run(int d, int Mloc, int Sa, float alpha, bool timing) {
  const int cpb = 1, nblk = Sa / 2;
  std::vector<float> scores((size_t)Mloc * d * d);
  srand(7);
  for (auto& s : scores) s = 4.0f * ((float)rand() / RAND_MAX - 0.5f);
  const size_t np = (size_t)Mloc * nblk * d * d;
  std::vector<float> p0(np), p1(np);
  float *ds, *dp0, *dp1;
  hipMalloc(&ds, scores.size() * 4); hipMalloc(&dp0, np * 4); hipMalloc(&dp1, np * 4);
  hipMemcpy(ds, scores.data(), scores.size() * 4, hipMemcpyHostToDevice);
  hipMemset(dp0, 0, np * 4); hipMemset(dp1, 0, np * 4);
  constexpr int DP = 64, LD = DP + 4;
  const size_t lds0 = (3 * DP + 1) * LD * 4, lds1 = 2 * ABF_IMG_BYTES;
  hipFuncSetAttribute((const void*)k_acyc<4, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds0);
  hipFuncSetAttribute((const void*)k_acyc_bf<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1);
  hipFuncSetAttribute((const void*)k_acyc_bf<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1);
  Key2 carry{123u, 456u};
  const dim3 grid(nblk, Mloc);
  hipLaunchKernelGGL((k_acyc<4, true>), grid, dim3(256), lds0, 0, ds, dp0, carry, 0, Mloc, d, Sa, cpb, alpha, 1.0f, 0, 0, nblk, LikArgs{});
  if (d > 48) hipLaunchKernelGGL(k_acyc_bf<true>, dim3(nblk, (Mloc + 7) & ~7), dim3(256), lds1, 0, ds, dp1, carry, 0, Mloc, Mloc, d, Sa, cpb, alpha, 1.0f, 0, 0, nblk, LikArgs{});
  else hipLaunchKernelGGL(k_acyc_bf<false>, dim3(nblk, (Mloc + 7) & ~7), dim3(256), lds1, 0, ds, dp1, carry, 0, Mloc, Mloc, d, Sa, cpb, alpha, 1.0f, 0, 0, nblk, LikArgs{});
  hipError_t e2 = hipDeviceSynchronize();
  hipMemcpy(p0.data(), dp0, np * 4, hipMemcpyDeviceToHost);
  hipMemcpy(p1.data(), dp1, np * 4, hipMemcpyDeviceToHost);
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
  double maxref = 0, e0 = 0, e1 = 0; int nan = 0;
  for (int i = 0; i < d*d; ++i) {
    double s0 = 0, s1 = 0;
    for (int b = 0; b < nblk; ++b) { s0 += p0[(size_t)b*d*d + i]; s1 += p1[(size_t)b*d*d + i]; }
    maxref = fmax(maxref, fabs(acc[i]));
    if (!std::isfinite(s1)) ++nan;
    e0 = fmax(e0, fabs(acc[i] - s0)); e1 = fmax(e1, fabs(acc[i] - s1));
  }
  double dmax = 0, pmax = 0;
  for (size_t i = 0; i < np; ++i) { dmax = fmax(dmax, fabs((double)p0[i] - p1[i])); pmax = fmax(pmax, fabs((double)p0[i])); }
  printf("d=%d Mloc=%d Sa=%d alpha=%g sync=%s nan=%d  relerr vs double: f32-mfma %.3g  bf16x3 %.3g   max|f32-bf16x3|/max = %.3g\n", d, Mloc, Sa, alpha,
         hipGetErrorName(e2), nan, e0 / maxref, e1 / maxref, dmax / pmax);
  if (timing) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int which = 0; which < 2; ++which) {
      float best = 1e9f;
      for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(a, 0);
        for (int it = 0; it < 10; ++it) {
          if (which == 0) hipLaunchKernelGGL((k_acyc<4, true>), grid, dim3(256), lds0, 0, ds, dp0, carry, 0, Mloc, d, Sa, cpb, alpha, 1.0f, 0, 0, nblk, LikArgs{});
          else if (d > 48) hipLaunchKernelGGL(k_acyc_bf<true>, dim3(nblk, (Mloc + 7) & ~7), dim3(256), lds1, 0, ds, dp1, carry, 0, Mloc, Mloc, d, Sa, cpb, alpha, 1.0f, 0, 0, nblk, LikArgs{});
  else hipLaunchKernelGGL(k_acyc_bf<false>, dim3(nblk, (Mloc + 7) & ~7), dim3(256), lds1, 0, ds, dp1, carry, 0, Mloc, Mloc, d, Sa, cpb, alpha, 1.0f, 0, 0, nblk, LikArgs{});
        }
        hipEventRecord(b, 0); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b); best = fminf(best, ms / 10);
      }
      printf("   %s: %.1f us per launch\n", which ? "k_acyc_bf      " : "k_acyc<4,true> ", best * 1e3f);
    }
  }
  hipFree(ds); hipFree(dp0); hipFree(dp1);
}

int main() {
  short* dout; hipMalloc(&dout, 512);
  hipLaunchKernelGGL(k_tr, dim3(1), dim3(64), 0, 0, dout);
  short h[256]; hipMemcpy(h, dout, 512, hipMemcpyDeviceToHost);
  int ok = 1;
  for (int l = 0; l < 64; ++l) for (int j = 0; j < 4; ++j) if (h[l*4+j] != 64 * (l >> 4) + j * 16 + (l & 15)) ok = 0;
  printf("ds_read_b64_tr_b16: lane l elem j == block[(l>>4)][row j][col l&15]: %s\n", ok ? "yes" : "NO");
  if (!ok) { for (int l = 0; l < 20; ++l) printf("  lane %d: %d %d %d %d\n", l, h[l*4], h[l*4+1], h[l*4+2], h[l*4+3]); }
  run(50, 2, 4, 0.0f, false);
  run(50, 2, 4, 0.5f, false);
  run(64, 2, 4, 0.2f, false);
  run(33, 1, 2, 0.2f, false);
  run(60, 1, 2, 2.0f, false);
  run(50, 128, 32, 0.05f, true);
  run(40, 128, 32, 0.05f, true);
  run(64, 128, 32, 0.05f, true);
  return 0;
}
