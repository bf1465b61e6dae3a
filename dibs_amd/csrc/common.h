// helpers shared by every kernel file (gfx950)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "rng.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double wave_max_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const double t = __shfl_xor(v, o, 64);
    v = t > v ? t : v;
  }
  return v;
}

__device__ __forceinline__ double sigmoid_d(double v) { return 1.0 / (1.0 + exp(-v)); }


// LDS traffic between lanes of ONE wave: the hardware executes a wave's DS operations in order, so only the
// compiler has to be kept from reordering them.
__device__ __forceinline__ void wave_lds_fence() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
