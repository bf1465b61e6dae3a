// What does a cross-stream dependency cost on the critical path?  (gfx950, ROCm 7.2)   hipcc --offload-arch=gfx950 -O2 stream_hop.hip -o stream_hop
// Kernels stamp the 100 MHz wall clock at their first and last instruction; gaps are between those stamps (no profiler involved).
//   A  s1: k1, k2                                      gap k1 -> k2 (same stream, nothing in between)
//   B  s1: k1, record(ev), k2                          an event record between two kernels
//   C  s1: k1, record(ev), k2;  s2: wait(ev), k3       the fork of engine.hip::step_local (k3 = acyclicity kernel)
//   D  s1: k1 [ext launch, stopEvent = ev], k2; s2: wait(ev), k3      fork without a record packet: the event IS k1's completion signal
//   E  s2: k3, record(evj);  s1: k1 (long), wait(evj), k2             the join of step_local, the awaited kernel long finished
//   F  s2: k3 [ext launch, stopEvent = evj]; s1: k1, wait(evj), k2
//   G  s2: k3, kflag;  s1: k1, k2 (k2 polls the flag itself)           join inside the consumer kernel: no packet on s1
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void k_spin(unsigned long long* stamps, int slot, int ticks, const unsigned int* flag, unsigned int want) {
  const unsigned long long t0 = wall_clock64();
  if (flag && threadIdx.x == 0) {
    while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) __builtin_amdgcn_s_sleep(2);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  __syncthreads();
  while (wall_clock64() - t0 < (unsigned long long)ticks) {}
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    stamps[2 * slot] = t0;
    stamps[2 * slot + 1] = wall_clock64();
  }
}
__global__ void k_flag(unsigned int* flag, unsigned int v) { __hip_atomic_store(flag, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT); }

int main() {
  hipStream_t s1, s2;
  int lo, hi;
  OK(hipDeviceGetStreamPriorityRange(&lo, &hi));
  OK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
  OK(hipStreamCreateWithPriority(&s2, hipStreamNonBlocking, hi));
  hipEvent_t ev, evj;
  OK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
  OK(hipEventCreateWithFlags(&evj, hipEventDisableTiming));
  unsigned long long* st;
  unsigned int* flag;
  OK(hipMalloc(&st, 64 * 8));
  OK(hipMalloc(&flag, 4));
  OK(hipMemset(flag, 0, 4));
  const int REP = 40;
  const dim3 g(64), b(256);
  auto run = [&](char v, double& g12, double& g13) -> int {
    std::vector<double> a12, a13;
    for (int r = 0; r < REP; ++r) {
      OK(hipMemset(st, 0, 64 * 8));
      OK(hipDeviceSynchronize());
      const unsigned int want = (unsigned int)(r + 1) + 1000u * (unsigned int)v;
      switch (v) {
        case 'A':
          hipLaunchKernelGGL(k_spin, g, b, 0, s1, st, 0, 1000, nullptr, 0u);
          hipLaunchKernelGGL(k_spin, g, b, 0, s1, st, 1, 1000, nullptr, 0u);
          break;
        case 'B':
          hipLaunchKernelGGL(k_spin, g, b, 0, s1, st, 0, 1000, nullptr, 0u);
          OK(hipEventRecord(ev, s1));
          hipLaunchKernelGGL(k_spin, g, b, 0, s1, st, 1, 1000, nullptr, 0u);
          break;
        case 'C':
          hipLaunchKernelGGL(k_spin, g, b, 0, s1, st, 0, 1000, nullptr, 0u);
          OK(hipEventRecord(ev, s1));
          OK(hipStreamWaitEvent(s2, ev, 0));
          hipLaunchKernelGGL(k_spin, g, b, 0, s2, st, 2, 1000, nullptr, 0u);
          hipLaunchKernelGGL(k_spin, g, b, 0, s1, st, 1, 1000, nullptr, 0u);
          break;
        case 'D':
          hipExtLaunchKernelGGL(k_spin, g, b, 0, s1, nullptr, ev, 0, st, 0, 1000, nullptr, 0u);
          OK(hipStreamWaitEvent(s2, ev, 0));
          hipLaunchKernelGGL(k_spin, g, b, 0, s2, st, 2, 1000, nullptr, 0u);
          hipLaunchKernelGGL(k_spin, g, b, 0, s1, st, 1, 1000, nullptr, 0u);
          break;
        case 'E':
          hipLaunchKernelGGL(k_spin, g, b, 0, s2, st, 2, 500, nullptr, 0u);
          OK(hipEventRecord(evj, s2));
          hipLaunchKernelGGL(k_spin, g, b, 0, s1, st, 0, 4000, nullptr, 0u);
          OK(hipStreamWaitEvent(s1, evj, 0));
          hipLaunchKernelGGL(k_spin, g, b, 0, s1, st, 1, 1000, nullptr, 0u);
          break;
        case 'F':
          hipExtLaunchKernelGGL(k_spin, g, b, 0, s2, nullptr, evj, 0, st, 2, 500, nullptr, 0u);
          hipLaunchKernelGGL(k_spin, g, b, 0, s1, st, 0, 4000, nullptr, 0u);
          OK(hipStreamWaitEvent(s1, evj, 0));
          hipLaunchKernelGGL(k_spin, g, b, 0, s1, st, 1, 1000, nullptr, 0u);
          break;
        case 'G':
          hipLaunchKernelGGL(k_spin, g, b, 0, s2, st, 2, 500, nullptr, 0u);
          hipLaunchKernelGGL(k_flag, dim3(1), dim3(1), 0, s2, flag, want);
          hipLaunchKernelGGL(k_spin, g, b, 0, s1, st, 0, 4000, nullptr, 0u);
          hipLaunchKernelGGL(k_spin, g, b, 0, s1, st, 1, 1000, flag, want);
          break;
      }
      OK(hipDeviceSynchronize());
      unsigned long long h[6];
      OK(hipMemcpy(h, st, sizeof h, hipMemcpyDeviceToHost));
      if (r >= 5) {
        a12.push_back(((double)h[2] - (double)h[1]) * 0.01);
        if (h[4]) a13.push_back(((double)h[4] - (double)h[1]) * 0.01);
      }
    }
    std::sort(a12.begin(), a12.end());
    std::sort(a13.begin(), a13.end());
    g12 = a12[a12.size() / 2];
    g13 = a13.empty() ? 0.0 : a13[a13.size() / 2];
    return 0;
  };
  for (char v : {'A', 'B', 'C', 'D', 'E', 'F', 'G'}) {
    double g12, g13;
    if (run(v, g12, g13)) return 1;
    printf("%c: gap k1.end -> k2.start %6.2f us   k1.end -> k3.start %6.2f us (median of %d)\n", v, g12, g13, REP - 5);
  }
  return 0;
}
