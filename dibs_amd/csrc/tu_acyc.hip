// translation unit: acyclicity-gradient kernel instantiations and their launcher (kernels_acyc.h)
#define DIBS_TU_ACYC
#include "launch.h"
#include "kernels_acyc.h"
#include "kernels_acyc_bf16.h"
#include <stdlib.h>

template <int NT>
static void launch_nt(const AcycLaunch& a) {
  constexpr int DP = 16 * NT, LD = DP + 4;
  size_t lds = (size_t)(3 * DP + 1) * LD * 4;  // + one slack row (the k pipeline may load one step past the end)
  const bool paired = a.units != a.Sa;
  const dim3 grid(a.nblk, a.Mloc);
  if (paired) {
    dibs_allow_lds((const void*)k_acyc<NT, true>, lds);
    hipLaunchKernelGGL((k_acyc<NT, true>), grid, dim3(256), lds, a.stream, a.scores, a.part, a.carry, a.m0, a.M, a.d, a.Sa, a.cpb, a.alpha,
                       a.tau, a.layout, a.tiny, a.nblk);
  } else {
    dibs_allow_lds((const void*)k_acyc<NT, false>, lds);
    hipLaunchKernelGGL((k_acyc<NT, false>), grid, dim3(256), lds, a.stream, a.scores, a.part, a.carry, a.m0, a.M, a.d, a.Sa, a.cpb, a.alpha,
                       a.tau, a.layout, a.tiny, a.nblk);
  }
}

// 33 <= d <= 64 with paired chains: split-bf16 MFMA kernel (kernels_acyc_bf16.h); DIBS_ACYC_F32=1 keeps the f32-MFMA kernel (A/B runs)
static bool acyc_use_bf16(const AcycLaunch& a) {
  static const bool off = getenv("DIBS_ACYC_F32") != nullptr;
  return !off && a.units != a.Sa && a.d > 32 && a.d <= 64;
}

static void acyc_launch_power(const AcycLaunch& a);
void acyc_launch(const AcycLaunch& a) {
  acyc_launch_power(a);
  const int dd = a.d * a.d;
  hipLaunchKernelGGL(k_acyc_reduce, dim3(a.Mloc, (dd + 255) / 256), dim3(256), 0, a.stream, a.part, a.w_acyc, a.nblk, dd, 1.0f / (float)a.Sa);
}
static void acyc_launch_power(const AcycLaunch& a) {
  if (acyc_use_bf16(a)) {
    size_t lds = 2 * ABF_IMG_BYTES;
      const dim3 grid(a.nblk, (a.Mloc + 7) & ~7);
    if (a.d > 48) {
      dibs_allow_lds((const void*)k_acyc_bf<true>, lds);
      hipLaunchKernelGGL(k_acyc_bf<true>, grid, dim3(256), lds, a.stream, a.scores, a.part, a.carry, a.m0, a.M, a.Mloc, a.d, a.Sa, a.cpb, a.alpha,
                         a.tau, a.layout, a.tiny, a.nblk);
    } else {
      dibs_allow_lds((const void*)k_acyc_bf<false>, lds);
      hipLaunchKernelGGL(k_acyc_bf<false>, grid, dim3(256), lds, a.stream, a.scores, a.part, a.carry, a.m0, a.M, a.Mloc, a.d, a.Sa, a.cpb, a.alpha,
                         a.tau, a.layout, a.tiny, a.nblk);
    }
    return;
  }
  switch ((a.d + 15) / 16) {
    case 1: launch_nt<1>(a); break;
    case 2: launch_nt<2>(a); break;
    case 3: launch_nt<3>(a); break;
    case 4: launch_nt<4>(a); break;
    case 5: launch_nt<5>(a); break;
    case 6: launch_nt<6>(a); break;
    default: launch_nt<7>(a); break;
  }
}
