// translation unit: acyclicity-gradient kernel instantiations and their launcher (kernels_acyc.h)
#define DIBS_TU_ACYC
#include "launch.h"
#include "kernels_acyc.h"
#include "kernels_acyc_bf16.h"
#include "kernels_acyc_f16.h"
#include "kernels_acyc_big.h"
#include <stdlib.h>
#include <hip/hip_ext.h>

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

// 33 <= d <= 64 with paired chains: the split-operand kernels on the 16-bit matrix pipe (kernels_acyc_f16.h, kernels_acyc_bf16.h);
// a.pipe == DIBS_PIPE_F32 keeps the f32-MFMA kernel (tuning.h: A/B runs)
// (d <= 32 stays on k_acyc<NT>: measured at config 2 (d = 20, 32 particles) the 64-padded k_acyc_hf takes 17.4 us against 14.3 us for
//  k_acyc<2> -- its element-order draws do not make up for products that are 3 x 3 instead of 2 x 2 tiles)
static bool acyc_use_bf16(const AcycLaunch& a) { return a.pipe != DIBS_PIPE_F32 && a.units != a.Sa && a.d >= 33 && a.d <= 64; }
// 65 <= d <= 112 with paired chains: the same schemes with NT = 5 .. 7 tiles and waves (k_acyc_hfw / k_acyc_bfw)
static bool acyc_use_bfw(const AcycLaunch& a) { return a.pipe != DIBS_PIPE_F32 && a.units != a.Sa && a.d > 64 && a.d <= 112; }
// two-piece f16 operands for 65 <= d <= 112 (k_acyc_hfw, kernels_acyc_f16.h); DIBS_ACYC_BF16=1 keeps k_acyc_bfw (A/B runs)
template <int NT>
static void launch_hfw(const AcycLaunch& a) {
  const size_t lds = ahfw_lds_bytes(NT);
  const dim3 grid(a.nblk, (a.Mloc + 7) & ~7);
  dibs_allow_lds((const void*)k_acyc_hfw<NT>, lds);
  hipLaunchKernelGGL(k_acyc_hfw<NT>, grid, dim3(64 * NT), lds, a.stream, a.scores, a.eas, a.part, a.carry, a.m0, a.M, a.Mloc, a.d, a.Sa, a.cpb, a.alpha,
                     a.tau, a.layout, a.tiny, a.nblk);
}
template <int NT>
static void launch_bfw(const AcycLaunch& a) {
  const size_t lds = abfw_lds_bytes(NT);
  const dim3 grid(a.nblk, (a.Mloc + 7) & ~7);
  dibs_allow_lds((const void*)k_acyc_bfw<NT>, lds);
  hipLaunchKernelGGL(k_acyc_bfw<NT>, grid, dim3(64 * NT), lds, a.stream, a.scores, a.part, a.carry, a.m0, a.M, a.Mloc, a.d, a.Sa, a.cpb, a.alpha,
                     a.tau, a.layout, a.tiny, a.nblk);
}

// n_vars > 112: matrices in global memory (kernels_acyc_big.h); a.big = three [Mloc * Sa][dp][dp] buffers
size_t acyc_big_elems(int Mloc, int d, int Sa) {
  const size_t dp = (size_t)((d + 15) & ~15);
  return (size_t)3 * Mloc * Sa * dp * dp;
}
static void acyc_big_launch(const AcycLaunch& a) {
  const int dp = (a.d + 15) & ~15, nch = a.Mloc * a.Sa;
  const size_t msz = (size_t)nch * dp * dp;
  float* const Mb = a.big;
  float* const B1 = Mb + msz;
  float* const B2 = B1 + msz;
  hipLaunchKernelGGL(k_acycb_init, dim3((dp * dp + 255) / 256, nch), dim3(256), 0, a.stream, a.scores, Mb, a.carry, a.m0, a.M, a.d, dp, a.Sa, a.alpha,
                     a.tau, a.layout, a.tiny);
  const dim3 gg((dp + 63) / 64, (dp + 63) / 64, nch);
  const float* cur = Mb;
  const int ex = a.d - 1;
  const int hb = 31 - __builtin_clz((unsigned)ex);
  for (int bit = hb - 1; bit >= 0; --bit) {  // left-to-right binary powering (as k_acyc)
    float* dst = cur == B1 ? B2 : B1;
    hipLaunchKernelGGL(k_bgemm, gg, dim3(256), 0, a.stream, cur, cur, dst, dp);
    cur = dst;
    if ((ex >> bit) & 1) {
      dst = cur == B1 ? B2 : B1;
      hipLaunchKernelGGL(k_bgemm, gg, dim3(256), 0, a.stream, (const float*)Mb, cur, dst, dp);
      cur = dst;
    }
  }
  hipLaunchKernelGGL(k_acycb_out, dim3((a.d * a.d + 255) / 256, a.Mloc), dim3(256), 0, a.stream, a.scores, cur, a.w_acyc, a.carry, a.m0, a.M, a.d, dp,
                     a.Sa, a.alpha, a.tau, a.layout, a.tiny);
}

// mean over the chains: fixed-order sum of the blocks' partial sums (the global-memory path of n_vars > 112 sums inside k_acycb_out)
void acyc_launch_reduce(const AcycLaunch& a) {
  if (a.big) return;
  const int dd = a.d * a.d;
  hipLaunchKernelGGL(k_acyc_reduce, dim3(a.Mloc, (dd + 255) / 256), dim3(256), 0, a.stream, a.part, a.w_acyc, a.nblk, dd, 1.0f / (float)a.Sa);
}
bool acyc_power_takes_events(const AcycLaunch& a) { return !a.big && acyc_use_bf16(a); }
void acyc_launch_power(const AcycLaunch& a) {
  if (a.big) {
    acyc_big_launch(a);
    return;
  }
  if (acyc_use_bf16(a)) {
    // two-piece f16 operands (kernels_acyc_f16.h: half the matrix instructions); DIBS_PIPE_BF16: the three-piece bf16 kernel (A/B runs)
    const bool bf16 = a.pipe == DIBS_PIPE_BF16;
    const size_t lds = bf16 ? (size_t)2 * ABF_IMG_BYTES : (size_t)AHF_LDS_BYTES;
    const dim3 grid(a.nblk, (a.Mloc + 7) & ~7);
#define ACYC_BF_LAUNCH(KERNEL_)                                                                                                            \
    {                                                                                                                                        \
      dibs_allow_lds((const void*)KERNEL_, lds);                                                                                             \
      if (a.ev_start)                                                                                                                        \
        hipExtLaunchKernelGGL(KERNEL_, grid, dim3(256), (uint32_t)lds, a.stream, a.ev_start, a.ev_stop, 0u, ACYC_SRC, a.part, a.carry,       \
                              a.m0, a.M, a.Mloc, a.d, a.Sa, a.cpb, a.alpha, a.tau, a.layout, a.tiny, a.nblk);                                  \
      else                                                                                                                                   \
        hipLaunchKernelGGL(KERNEL_, grid, dim3(256), lds, a.stream, ACYC_SRC, a.part, a.carry, a.m0, a.M, a.Mloc, a.d, a.Sa, a.cpb,            \
                           a.alpha, a.tau, a.layout, a.tiny, a.nblk);                                                                          \
    }
#define ACYC_SRC a.scores
    if (bf16) {
      if (a.d > 48) ACYC_BF_LAUNCH(k_acyc_bf<true>) else ACYC_BF_LAUNCH(k_acyc_bf<false>)
    } else {
#undef ACYC_SRC
#define ACYC_SRC a.scores, a.eas
      // three waves per SIMD (no spills, M's fragments kept)
      if (a.d > 48) ACYC_BF_LAUNCH((k_acyc_hf<true, 3>)) else ACYC_BF_LAUNCH((k_acyc_hf<false, 3>))
    }
#undef ACYC_BF_LAUNCH
#undef ACYC_SRC
    return;
  }
  if (acyc_use_bfw(a)) {
    // The two-piece f16 scheme carries a truncation bias inside the MFMA's dot products (measured against the f32-MFMA kernel,
    // test_acyclicity_f16_pipe_worst_cases): k_acyc_hf (K = 64) -9e-7 on M^(d-1) at d = 64; k_acyc_hfw (K = 96 / 128) ~-1.15e-7 per
    // product level, i.e. -(d - 1) 1.15e-7 = -1.0e-5 at d = 80, which the kernel compensates to first order (residual 4e-6).  With the
    // compensation the whole range 65 .. 112 runs on it (round 4 fenced it to d <= 80): against the f64 oracle it is as close as the
    // f32-MFMA kernel at d = 96 / 100 / 112 (5.9e-6 / 7.0e-5 at alpha = 1000 / 5.9e-6 vs 5.9e-6 / 7.3e-5 / 8.8e-6); config 5 (d = 100)
    // 116.6 -> 123.2 steps/s.  a.hfw_max = 80 restores the fence, DIBS_PIPE_BF16 selects the three-piece bf16 kernel (tuning.h).
    if (a.pipe != DIBS_PIPE_BF16 && a.d <= a.hfw_max) {
      switch ((a.d + 15) / 16) {
        case 5: launch_hfw<5>(a); break;
        case 6: launch_hfw<6>(a); break;
        default: launch_hfw<7>(a); break;
      }
      return;
    }
    switch ((a.d + 15) / 16) {
      case 5: launch_bfw<5>(a); break;
      case 6: launch_bfw<6>(a); break;
      default: launch_bfw<7>(a); break;
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
