#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include <cmath>
#include "../../dibs_amd/csrc/kernels_marginal.h"
template <int NT>
__global__ __launch_bounds__(256) void kmm(const float* A, const float* B, float* C, int kp, int reps) {
  constexpr int DP = 16 * NT, LD = DP + 2;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem; float* Bs = smem + DP * LD; float* Cs = Bs + DP * LD;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int e = tid; e < DP * LD; e += 256) { int i = e / LD, j = e % LD; As[e] = j < DP ? A[i * DP + j] : 0.f; Bs[e] = j < DP ? B[i * DP + j] : 0.f; Cs[e] = -1.f; }
  __syncthreads();
  for (int r = 0; r < reps; ++r) { lds_matmul<NT>(Cs, As, Bs, kp, lane, wave); __syncthreads(); }
  for (int e = tid; e < DP * DP; e += 256) C[e] = Cs[(e / DP) * LD + e % DP];
}
template <int NT> void run() {
  constexpr int DP = 16 * NT, LD = DP + 2;
  std::vector<float> A(DP * DP), B(DP * DP), C(DP * DP);
  for (int i = 0; i < DP * DP; ++i) { A[i] = sinf(i * 0.37f); B[i] = cosf(i * 0.11f + 1); }
  float *dA, *dB, *dC; hipMalloc(&dA, DP * DP * 4); hipMalloc(&dB, DP * DP * 4); hipMalloc(&dC, DP * DP * 4);
  hipMemcpy(dA, A.data(), DP * DP * 4, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), DP * DP * 4, hipMemcpyHostToDevice);
  size_t lds = 3 * DP * LD * 4;
  hipError_t ea = hipFuncSetAttribute((const void*)kmm<NT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(kmm<NT>, dim3(1), dim3(256), lds, 0, dA, dB, dC, DP, 1);
  hipError_t e2 = hipDeviceSynchronize();
  hipMemcpy(C.data(), dC, DP * DP * 4, hipMemcpyDeviceToHost);
  double maxerr = 0; int bad = 0;
  for (int i = 0; i < DP; ++i) for (int j = 0; j < DP; ++j) { double s = 0; for (int k = 0; k < DP; ++k) s += (double)A[i * DP + k] * B[k * DP + j]; double e = fabs(s - C[i * DP + j]); if (!(e < 1e-3)) { if (bad < 5) printf("  bad (%d,%d) got %g want %g\n", i, j, C[i*DP+j], s); ++bad; } if (e > maxerr) maxerr = e; }
  printf("NT=%d lds=%zu attr=%s sync=%s maxerr=%g bad=%d\n", NT, lds, hipGetErrorName(ea), hipGetErrorName(e2), maxerr, bad);
}
int main() { run<4>(); run<5>(); run<6>(); run<8>(); return 0; }
