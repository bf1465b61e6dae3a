// Operand / result lane mapping of v_mfma_f32_4x4x1_16B_f32 on gfx950: 16 blocks of 4x4 outer products (K = 1).
// Hypothesis: a: lane = 4*blk + i (row i of block blk); b: lane = 4*blk + j (column j); D: lane 4*blk + j, register i.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(float* out) {
  const int lane = threadIdx.x, blk = lane >> 2, q = lane & 3;
  const float a = (float)((blk + 1) * 100 + (q + 1));      // A[blk][i = q]
  const float b = (float)((blk + 1) * 1000 + (q + 1) * 7);  // B[blk][j = q]
  f32x4 c = {0.f, 0.f, 0.f, 0.f};
  c = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0);
  for (int r = 0; r < 4; ++r) out[lane * 4 + r] = c[r];
}
int main() {
  float* d; hipMalloc(&d, 64 * 4 * 4);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  float h[256]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int lane = 0; lane < 64; ++lane)
    for (int r = 0; r < 4; ++r) {
      const int blk = lane >> 2, j = lane & 3, i = r;
      const float want = (float)((blk + 1) * 100 + (i + 1)) * (float)((blk + 1) * 1000 + (j + 1) * 7);
      if (h[lane * 4 + r] != want) { if (bad < 8) printf("lane %d reg %d: got %g want %g\n", lane, r, h[lane * 4 + r], want); ++bad; }
    }
  printf("4x4x1 layout hypothesis: %s (%d mismatches)\n", bad ? "WRONG" : "confirmed", bad);
  return 0;
}
