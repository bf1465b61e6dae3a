// VALU issue rate on gfx950: wave-instructions per cycle per SIMD for f32 FMA, integer add/xor/alignbit (Threefry mix), f64 FMA,
// v_log_f32 / v_rsq_f32, with 1 / 2 / 4 / 8 waves per SIMD and 1 / 2 / 4 independent chains per wave.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int KIND, int CH>
__global__ void k(int iters, float* out) {
  float f[CH]; uint32_t a[CH], b[CH]; double g[CH];
#pragma unroll
  for (int c = 0; c < CH; ++c) { f[c] = threadIdx.x * 1e-3f + c; a[c] = threadIdx.x + c; b[c] = blockIdx.x * 7 + c; g[c] = f[c]; }
  const float m = 1.0001f, ad = 1e-3f;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int r = 0; r < 8; ++r) {
#pragma unroll
      for (int c = 0; c < CH; ++c) {
        if (KIND == 0) { asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[c]) : "v"(m), "v"(ad)); }
        if (KIND == 1) {
          asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[c]) : "v"(b[c]));
          asm volatile("v_alignbit_b32 %0, %0, %0, 19" : "+v"(b[c]));
          asm volatile("v_xor_b32 %0, %0, %1" : "+v"(b[c]) : "v"(a[c]));
        }
        if (KIND == 2) { asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(g[c]) : "v"((double)m), "v"((double)ad)); }
        if (KIND == 3) { asm volatile("v_log_f32 %0, %0" : "+v"(f[c])); }
        if (KIND == 4) { asm volatile("v_rsq_f32 %0, %0" : "+v"(f[c])); }
        if (KIND == 5) { asm volatile("v_add3_u32 %0, %0, %1, %1" : "+v"(a[c]) : "v"(b[c])); }
        if (KIND == 6) { asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(g[c]) : "v"((double)m)); }
      }
    }
  }
  float s = 0; 
#pragma unroll
  for (int c = 0; c < CH; ++c) s += f[c] + (float)a[c] + (float)b[c] + (float)g[c];
  if (s == 123.456f) out[0] = s;
}

template <int KIND, int CH>
static void run(const char* name, int per_iter) {
  float* out; hipMalloc(&out, 16);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 4000;
  printf("%-28s chains %d:", name, CH);
  for (int wps : {1, 2, 4, 8}) {
    const int threads = 256, blocks = 256 * wps;  // wps blocks of 4 waves per CU -> wps waves per SIMD
    hipLaunchKernelGGL((k<KIND, CH>), dim3(blocks), dim3(threads), 0, 0, 10, out);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((k<KIND, CH>), dim3(blocks), dim3(threads), 0, 0, iters, out);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double instr_per_simd = (double)iters * 8 * CH * per_iter * wps;
    printf("  w/simd %d: %.3f ms %.2f cyc/instr@2.4GHz", wps, ms, ms * 1e-3 * 2.4e9 / instr_per_simd);
  }
  printf("\n");
}
int main() {
  run<0, 1>("v_fma_f32", 1); run<0, 2>("v_fma_f32", 1); run<0, 4>("v_fma_f32", 1); run<0, 8>("v_fma_f32", 1);
  run<1, 1>("add/alignbit/xor", 3); run<1, 2>("add/alignbit/xor", 3); run<1, 4>("add/alignbit/xor", 3);
  run<5, 1>("v_add3_u32", 1); run<5, 4>("v_add3_u32", 1);
  run<2, 1>("v_fma_f64", 1); run<2, 4>("v_fma_f64", 1);
  run<3, 1>("v_log_f32", 1); run<3, 4>("v_log_f32", 1);
  run<4, 1>("v_rsq_f32", 1); run<4, 4>("v_rsq_f32", 1);
  run<6, 1>("v_pk_fma_f32", 1); run<6, 4>("v_pk_fma_f32", 1);
  return 0;
}
