// Do MFMA and VALU instructions of DIFFERENT waves on the same SIMD overlap on gfx950?  8 waves per block: waves 0-3 (one per
// SIMD) run an MFMA loop, waves 4-7 (the second wave of each SIMD) run an integer VALU loop.  Times: MFMA only, VALU only, both.
// If they overlap, t(both) ~ max; if they share the issue / datapath, t(both) ~ sum.   f32 16x16x4 vs bf16 16x16x16.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

template <int KIND>  // 0: f32 16x16x4, 1: bf16 16x16x16
__global__ __launch_bounds__(512) void k(int iters, int do_mfma, int do_valu, float* out) {
  const int wave = threadIdx.x >> 6;
  if (wave < 4) {
    if (!do_mfma) return;
    f32x4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
    const float x = (float)threadIdx.x * 1e-3f, y = 1.0f + x;
    bf16x8 xa = {1, 2, 3, 4, 5, 6, 7, 8}, xb = {8, 7, 6, 5, 4, 3, 2, 1};
    for (int i = 0; i < iters; ++i) {
      if (KIND == 0) {
        a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a2, 0, 0, 0);
        a3 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a3, 0, 0, 0);
      } else {
        a0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xa, xb, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xa, xb, a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xa, xb, a2, 0, 0, 0);
        a3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xa, xb, a3, 0, 0, 0);
      }
    }
    if (a0[0] + a1[0] + a2[0] + a3[0] == 123.456f) out[0] = 1.f;
  } else {
    if (!do_valu) return;
    unsigned v0 = threadIdx.x, v1 = blockIdx.x;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int r = 0; r < 8; ++r) {  // 24 dependent integer VALU ops (a Threefry-like round)
        v0 += v1;
        v1 = (v1 << 13) | (v1 >> 19);
        v1 ^= v0;
      }
    }
    if (v0 == 0x12345u) out[1] = (float)v1;
  }
}

template <int KIND>
static void run(const char* name) {
  float* out; hipMalloc(&out, 16);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  const int iters = 20000, blocks = 256;
  float t[3];
  for (int cfg = 0; cfg < 3; ++cfg) {
    const int dm = cfg != 1, dv = cfg != 0;
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(512), 0, 0, 100, dm, dv, out);
    hipDeviceSynchronize();
    hipEventRecord(a, 0);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(512), 0, 0, iters, dm, dv, out);
    hipEventRecord(b, 0); hipEventSynchronize(b);
    hipEventElapsedTime(&t[cfg], a, b);
  }
  printf("%s: MFMA only %.3f ms | VALU only %.3f ms | both %.3f ms  (sum %.3f, max %.3f)\n", name, t[0], t[1], t[2], t[0] + t[1],
         t[0] > t[1] ? t[0] : t[1]);
}
int main() {
  run<0>("f32  16x16x4 ");
  run<1>("bf16 16x16x32");
  return 0;
}
