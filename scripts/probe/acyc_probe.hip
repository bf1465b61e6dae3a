#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include <cmath>
#include "../../dibs_amd/csrc/kernels_marginal.h"
static void mm(const std::vector<double>& a, const std::vector<double>& b, std::vector<double>& c, int d) {
  for (int i = 0; i < d; ++i) for (int j = 0; j < d; ++j) { double s = 0; for (int k = 0; k < d; ++k) s += a[i*d+k]*b[k*d+j]; c[i*d+j] = s; }
}
template <int NT> void run(int d, int grid_y) {
  constexpr int DP = 16 * NT, LD = DP + 4;
  const int Sa = 4, cpb = 2;   // paired kernel: a work unit is the chain pair (sa, sa + Sa/2); 2 units cover Sa = 4
  std::vector<float> scores((size_t)grid_y * d * d, 0.f), part((size_t)grid_y * d * d);
  float *ds, *dp; hipMalloc(&ds, scores.size()*4); hipMalloc(&dp, part.size()*4);
  hipMemcpy(ds, scores.data(), scores.size()*4, hipMemcpyHostToDevice);
  hipMemset(dp, 0, part.size()*4);
  size_t lds = (3 * DP + 1) * LD * 4;
  hipFuncSetAttribute((const void*)k_acyc<NT, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  Key2 carry{123u, 456u};
  hipLaunchKernelGGL((k_acyc<NT, true>), dim3(1, grid_y), dim3(256), lds, 0, ds, dp, carry, 0, grid_y, d, Sa, cpb, 1.0f, 1.0f, 0, 0, 1, LikArgs{});
  hipError_t e2 = hipDeviceSynchronize();
  hipMemcpy(part.data(), dp, part.size()*4, hipMemcpyDeviceToHost);
  // CPU reference for particle 0
  Key2 km = rng_split_row(carry, grid_y + 1, 1, 0);
  std::vector<double> acc(d*d, 0.0), M(d*d), P(d*d), T(d*d), G(d*d);
  for (int sa = 0; sa < Sa; ++sa) {
    for (int i = 0; i < d; ++i) for (int j = 0; j < d; ++j) {
      uint32_t bits = rng_bits_at(km, (uint64_t)Sa*d*d, (uint64_t)sa*d*d + i*d + j, 0);
      float x = rng_uniform(bits, 1.1920929e-07f, 1.0f);
      double eps = log((double)x / (1.0 - (double)x));
      double g = i == j ? 0.0 : 1.0 / (1.0 + exp(-eps));
      G[i*d+j] = g; M[i*d+j] = (i == j) + g / d;
    }
    P = M; int ex = d - 1; int hb = 31 - __builtin_clz(ex);
    for (int b = hb - 1; b >= 0; --b) { mm(P, P, T, d); P = T; if ((ex >> b) & 1) { mm(P, M, T, d); P = T; } }
    for (int i = 0; i < d; ++i) for (int j = 0; j < d; ++j) if (i != j) acc[i*d+j] += P[j*d+i] * G[i*d+j] * (1 - G[i*d+j]);
  }
  double maxref = 0, maxerr = 0; int nan = 0;
  for (int i = 0; i < d*d; ++i) { maxref = fmax(maxref, fabs(acc[i])); if (!std::isfinite(part[i])) ++nan; else maxerr = fmax(maxerr, fabs(acc[i] - part[i])); }
  printf("NT=%d d=%d grid_y=%d sync=%s nan=%d relerr=%g\n", NT, d, grid_y, hipGetErrorName(e2), nan, maxerr / maxref);
}
int main() { run<4>(50, 1); run<4>(60, 1); run<4>(64, 2); run<5>(65, 1); run<3>(40,1); return 0; }
