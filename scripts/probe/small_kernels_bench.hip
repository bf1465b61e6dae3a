// Stand-alone timing of the small per-step kernels at the headline size (d = 50, M = 128) with synthetic buffers.
// Build variants with -DPHI_EXP=n / -DKMAT_EXP=n to bisect where a kernel's time goes (see scripts/probe/README.md).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "../../dibs_amd/csrc/kernels_marginal.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <typename F>
static float time_us(F&& launch, int reps = 200) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < 20; ++i) launch();
  hipDeviceSynchronize();
  hipEventRecord(a, 0);
  for (int i = 0; i < reps; ++i) launch();
  hipEventRecord(b, 0);
  hipEventSynchronize(b);
  float ms = 0; hipEventElapsedTime(&ms, a, b);
  return ms * 1e3f / reps;
}

int main() {
  const int d = 50, k = 50, M = 128, D = d * k * 2, E = 2 * D;
  float *pack, *kz, *x, *v, *phi;
  CK(hipMalloc(&pack, (size_t)M * E * 4)); CK(hipMalloc(&kz, (size_t)M * M * 4));
  CK(hipMalloc(&x, (size_t)M * D * 4)); CK(hipMalloc(&v, (size_t)M * D * 4)); CK(hipMalloc(&phi, (size_t)M * D * 4));
  std::vector<float> h((size_t)M * E);
  for (size_t i = 0; i < h.size(); ++i) h[i] = 0.1f * (float)((i * 2654435761u >> 8) & 0xFFFF) / 65536.f;
  CK(hipMemcpy(pack, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemset(kz, 0, (size_t)M * M * 4)); CK(hipMemset(v, 0, (size_t)M * D * 4)); CK(hipMemset(x, 0, (size_t)M * D * 4));
  {
    hipFuncSetAttribute((const void*)k_kmat, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    const size_t lds = (size_t)((D + 3) & ~3) * 4;
    const float t = time_us([&] { hipLaunchKernelGGL(k_kmat, dim3(M, M / KMAT_BT), dim3(256), lds, 0, pack, (size_t)E, (size_t)0, D, kz, 0, M, 1.0f, 5.0f, 1); });
    printf("k_kmat            %7.2f us\n", t);
  }
  {
    const int ta = 16;
    const size_t lds = ((size_t)2 * ta * M + (size_t)4 * ta * 64) * 4;
    const float t = time_us([&] {
      hipLaunchKernelGGL(k_phi_update<16>, dim3(8 * (M / ta) * (((D + 63) / 64 + 7) / 8)), dim3(256), lds, 0, pack, (size_t)E, (size_t)0, (size_t)D, D, kz,
                         (const float*)nullptr, 0, x, v, phi, 0, M, M, 5.0f, 0.005f, 1, (D + 63) / 64, M / ta);
    });
    printf("k_phi_update<16>  %7.2f us\n", t);
  }
  {
    const int ta = 4;
    const size_t lds = ((size_t)2 * ta * M + (size_t)4 * ta * 64) * 4;
    const float t = time_us([&] {
      hipLaunchKernelGGL(k_phi_update<4>, dim3(8 * (M / ta) * (((D + 63) / 64 + 7) / 8)), dim3(256), lds, 0, pack, (size_t)E, (size_t)0, (size_t)D, D, kz,
                         (const float*)nullptr, 0, x, v, phi, 0, M, M, 5.0f, 0.005f, 1, (D + 63) / 64, M / ta);
    });
    printf("k_phi_update<4>   %7.2f us\n", t);
  }
  {
    const float t = time_us([&] { hipLaunchKernelGGL(k_wtotal, dim3(1, 1), dim3(64), 0, 0, x, x, x, 0, v, phi, 1, 1, 1.0f, 1.0f, 2, 0.f); });
    printf("(tiny launch)     %7.2f us\n", t);
  }
  return 0;
}
