#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(float* out, int n) {
  extern __shared__ float s[];
  for (int i = threadIdx.x; i < n; i += blockDim.x) s[i] = (float)i;
  __syncthreads();
  if (threadIdx.x == 0) { out[0] = s[n - 1]; out[1] = s[n / 2]; }
}
int main() {
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  printf("sharedMemPerBlock %zu optin %zu perMP %zu\n", p.sharedMemPerBlock, p.sharedMemPerBlockOptin, p.maxSharedMemoryPerMultiProcessor);
  float* d; hipMalloc(&d, 8);
  for (int kb : {32, 64, 65, 96, 128, 160}) {
    int n = kb * 256; size_t bytes = (size_t)n * 4;
    hipError_t e1 = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    hipMemset(d, 0, 8);
    hipLaunchKernelGGL(k, dim3(1), dim3(256), bytes, 0, d, n);
    hipError_t e2 = hipGetLastError(); hipError_t e3 = hipDeviceSynchronize();
    float h[2]; hipMemcpy(h, d, 8, hipMemcpyDeviceToHost);
    printf("%d KB: attr=%s launch=%s sync=%s last=%g (want %d) mid=%g\n", kb, hipGetErrorName(e1), hipGetErrorName(e2), hipGetErrorName(e3), h[0], n - 1, h[1]);
  }
  return 0;
}
