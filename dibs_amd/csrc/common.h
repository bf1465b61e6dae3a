// helpers shared by every kernel file (gfx950)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include "rng.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
// sum of a double over the wave, the same value in every lane.  Rows of 16 lanes by DPP rotations (row_ror 8, 4, 2, 1: two v_mov_b32_dpp + one
// v_add_f64 per step), the four row sums (lanes 0, 16, 32, 48) through v_readlane and added in row order -- no LDS: the butterfly over
// __shfl_xor was twelve ds_bpermute_b32 with their latency per sum, two sums per sample pair and wave in the log-probability kernels.
__device__ __forceinline__ double wave_sum_d(double v) {
  auto ror = [](double x, auto ctrl) {
    constexpr int C = decltype(ctrl)::value;
    const int lo = __double2loint(x), hi = __double2hiint(x);
    return __hiloint2double(__builtin_amdgcn_update_dpp(0, hi, C, 0xf, 0xf, false), __builtin_amdgcn_update_dpp(0, lo, C, 0xf, 0xf, false));
  };
  v += ror(v, std::integral_constant<int, 0x128>{});  // row_ror:8
  v += ror(v, std::integral_constant<int, 0x124>{});  // row_ror:4
  v += ror(v, std::integral_constant<int, 0x122>{});  // row_ror:2
  v += ror(v, std::integral_constant<int, 0x121>{});  // row_ror:1
  const int lo = __double2loint(v), hi = __double2hiint(v);
  double t = __hiloint2double(__builtin_amdgcn_readlane(hi, 0), __builtin_amdgcn_readlane(lo, 0));
  t += __hiloint2double(__builtin_amdgcn_readlane(hi, 16), __builtin_amdgcn_readlane(lo, 16));
  t += __hiloint2double(__builtin_amdgcn_readlane(hi, 32), __builtin_amdgcn_readlane(lo, 32));
  t += __hiloint2double(__builtin_amdgcn_readlane(hi, 48), __builtin_amdgcn_readlane(lo, 48));
  return t;
}
__device__ __forceinline__ double wave_max_d(double v) {  // (same scheme as wave_sum_d; a maximum does not depend on the order)
  auto ror = [](double x, auto ctrl) {
    constexpr int C = decltype(ctrl)::value;
    const int lo = __double2loint(x), hi = __double2hiint(x);
    return __hiloint2double(__builtin_amdgcn_update_dpp(0, hi, C, 0xf, 0xf, false), __builtin_amdgcn_update_dpp(0, lo, C, 0xf, 0xf, false));
  };
  auto mx2 = [](double a, double b) { return b > a ? b : a; };
  v = mx2(v, ror(v, std::integral_constant<int, 0x128>{}));
  v = mx2(v, ror(v, std::integral_constant<int, 0x124>{}));
  v = mx2(v, ror(v, std::integral_constant<int, 0x122>{}));
  v = mx2(v, ror(v, std::integral_constant<int, 0x121>{}));
  const int lo = __double2loint(v), hi = __double2hiint(v);
  double t = __hiloint2double(__builtin_amdgcn_readlane(hi, 0), __builtin_amdgcn_readlane(lo, 0));
  t = mx2(t, __hiloint2double(__builtin_amdgcn_readlane(hi, 16), __builtin_amdgcn_readlane(lo, 16)));
  t = mx2(t, __hiloint2double(__builtin_amdgcn_readlane(hi, 32), __builtin_amdgcn_readlane(lo, 32)));
  t = mx2(t, __hiloint2double(__builtin_amdgcn_readlane(hi, 48), __builtin_amdgcn_readlane(lo, 48)));
  return t;
}

__device__ __forceinline__ double sigmoid_d(double v) { return 1.0 / (1.0 + exp(-v)); }


// LDS traffic between lanes of ONE wave: the hardware executes a wave's DS operations in order, so only the
// compiler has to be kept from reordering them.
__device__ __forceinline__ void wave_lds_fence() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
