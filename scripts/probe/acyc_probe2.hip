#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include <cmath>
#include "../../dibs_amd/csrc/kernels_marginal.h"
template <int NT>
__global__ __launch_bounds__(256) void k_dbg(float* dbg, int d) {
  constexpr int DP = 16 * NT, LD = DP + 2;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Mb = smem; float* X = smem + (size_t)DP * LD; float* Y = X + (size_t)DP * LD;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int kp = (d + 3) & ~3;
  for (int e = tid; e < DP * LD; e += 256) { const int i = e / LD, jj = e - i * LD; float v = 0.f; if (i < d && jj < d) v = (i == jj) ? 1.0f : (0.3f + 0.001f * ((i * 7 + jj * 3) % 50)) / d; Mb[e] = v; }
  __syncthreads();
  const float* cur = Mb; int stage = 0;
  for (int e = tid; e < DP * DP; e += 256) dbg[(size_t)stage * DP * DP + e] = cur[(e / DP) * LD + e % DP];
  ++stage;
  const int ex = d - 1; int hb = 31 - __builtin_clz((unsigned)ex);
  for (int b = hb - 1; b >= 0; --b) {
    float* dst = (cur == X) ? Y : X;
    lds_matmul<NT>(dst, cur, cur, kp, lane, wave);
    __syncthreads();
    cur = dst;
    for (int e = tid; e < DP * DP; e += 256) dbg[(size_t)stage * DP * DP + e] = cur[(e / DP) * LD + e % DP];
    ++stage;
    if ((ex >> b) & 1) {
      dst = (cur == X) ? Y : X;
      lds_matmul<NT>(dst, cur, Mb, kp, lane, wave);
      __syncthreads();
      cur = dst;
      for (int e = tid; e < DP * DP; e += 256) dbg[(size_t)stage * DP * DP + e] = cur[(e / DP) * LD + e % DP];
      ++stage;
    }
  }
}
template <int NT> void run(int d) {
  constexpr int DP = 16 * NT, LD = DP + 2;
  size_t lds = 3 * DP * LD * 4; int nst = 12;
  float* dd; hipMalloc(&dd, (size_t)nst * DP * DP * 4); hipMemset(dd, 0, (size_t)nst * DP * DP * 4);
  hipFuncSetAttribute((const void*)k_dbg<NT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(k_dbg<NT>, dim3(1), dim3(256), lds, 0, dd, d);
  hipDeviceSynchronize();
  std::vector<float> h((size_t)nst * DP * DP); hipMemcpy(h.data(), dd, h.size() * 4, hipMemcpyDeviceToHost);
  std::vector<double> M(DP * DP, 0.0), P, T(DP * DP);
  for (int i = 0; i < d; ++i) for (int j = 0; j < d; ++j) M[i * DP + j] = (i == j) ? 1.0 : (double)((0.3f + 0.001f * ((i * 7 + j * 3) % 50)) / d);
  auto mm = [&](const std::vector<double>& a, const std::vector<double>& b, std::vector<double>& c) { for (int i = 0; i < DP; ++i) for (int j = 0; j < DP; ++j) { double s = 0; for (int k = 0; k < DP; ++k) s += a[i*DP+k]*b[k*DP+j]; c[i*DP+j] = s; } };
  auto cmp = [&](int st, const std::vector<double>& ref) { double me = 0, mr = 0; int nan = 0, firstbad = -1; for (int e = 0; e < DP * DP; ++e) { float v = h[(size_t)st * DP * DP + e]; if (!std::isfinite(v)) { ++nan; if (firstbad < 0) firstbad = e; continue; } double er = fabs(v - ref[e]); if (er > me) { me = er; if (er > 1e-3 * (1 + fabs(ref[e])) && firstbad < 0) firstbad = e; } mr = fmax(mr, fabs(ref[e])); } printf("  NT=%d d=%d stage %d: nan=%d relerr=%g firstbad=(%d,%d)\n", NT, d, st, nan, me / mr, firstbad / DP, firstbad % DP); };
  P = M; int st = 0; cmp(st++, P);
  int ex = d - 1, hb = 31 - __builtin_clz(ex);
  for (int b = hb - 1; b >= 0; --b) { mm(P, P, T); P = T; cmp(st++, P); if ((ex >> b) & 1) { mm(P, M, T); P = T; cmp(st++, P); } }
}
int main() { run<4>(50); run<5>(65); run<5>(50); return 0; }
