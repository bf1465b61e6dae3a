// k_bge_sample at the headline size on synthetic thresholds: with / without the kernel-matrix blocks riding along, and a loop that
// only issues the Threefry calls (same count) -- how far is the kernel from its own instruction floor?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <random>
#include "../../dibs_amd/csrc/kernels_bge.h"

__global__ __launch_bounds__(256) void k_threefry_only(Key2 carry, int M, int d, int S, uint32_t* out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, m = blockIdx.y, j = blockIdx.x * 4 + wave;
  if (j >= d) return;
  const Key2 kp = rng_split_row_uniform(carry, (uint32_t)M + 1u, (uint32_t)m + 1u, 0);
  const Key2 kg = rng_split_row_uniform(kp, 2u, 1u, 0);
  const TfKeys tk = tf_keys(kg);
  const uint32_t dd = d * d, nbits = S * dd;
  uint32_t c0 = lane * dd + j, c1 = c0 + (nbits >> 1), acc = 0;
  for (int i = 0; i + 1 < d; i += 2, c0 += 2u * d, c1 += 2u * d) {
    uint32_t y0, y1, y2, y3;
    threefry2x32_uk2(tk, c0, c1, c0 + d, c1 + d, y0, y1, y2, y3);
    acc ^= y0 ^ y1 ^ y2 ^ y3;
  }
  if (acc == 0x12345678u) out[0] = acc;
}

int main() {
  const int d = 50, M = 128, S = 128, W = 1;
  std::mt19937 rng(1);
  std::vector<uint32_t> thr((size_t)M * d * d);
  for (auto& t : thr) t = rng() % 1000 < 100 ? (rng() & 0x7FFFFF) : (rng() % 3 == 0 ? 1 : 0);   // mostly empty parent sets
  uint32_t* d_thr; uint64_t* d_masks; double* d_ns; float *d_R, *d_z, *d_k; double *d_gam, *d_Nj, *d_ld; uint32_t* d_list; unsigned* d_cnt; uint32_t* d_out;
  hipMalloc(&d_thr, thr.size() * 4); hipMemcpy(d_thr, thr.data(), thr.size() * 4, hipMemcpyHostToDevice);
  hipMalloc(&d_masks, (size_t)M * d * S * 8); hipMalloc(&d_ns, (size_t)M * d * S * 8);
  std::vector<float> R(51 * 51, 0.f); for (int i = 0; i < 50; ++i) R[i * 51 + i] = 10.f;
  hipMalloc(&d_R, R.size() * 4); hipMemcpy(d_R, R.data(), R.size() * 4, hipMemcpyHostToDevice);
  std::vector<double> gam(50 * 51, 0.0), Nj(50, 100.0), ld(1, 0.0);
  hipMalloc(&d_gam, gam.size() * 8); hipMemcpy(d_gam, gam.data(), gam.size() * 8, hipMemcpyHostToDevice);
  hipMalloc(&d_Nj, 400); hipMemcpy(d_Nj, Nj.data(), 400, hipMemcpyHostToDevice);
  hipMalloc(&d_ld, 8); hipMemcpy(d_ld, ld.data(), 8, hipMemcpyHostToDevice);
  hipMalloc(&d_list, (size_t)BGE_NQ * M * d * S * 4); hipMalloc(&d_cnt, 64); hipMalloc(&d_out, 16);
  hipMalloc(&d_z, (size_t)M * 5000 * 4); hipMemset(d_z, 0, (size_t)M * 5000 * 4); hipMalloc(&d_k, M * M * 4);
  BgeParams bp{d_R, d_R, d_gam, d_Nj, d_ld, 52.0, 1};
  BgeQueues qs{d_list, d_cnt, (uint32_t)(M * d * S)};
  const size_t lds = 4 * bge_sample_wave_bytes(d, S, W), ldsk = 5000 * 4 + 64;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto time = [&](const char* name, auto&& launch) {
    float best = 1e9;
    for (int r = 0; r < 5; ++r) {
      hipMemset(d_cnt, 0, 64); hipDeviceSynchronize();
      hipEventRecord(e0, 0); launch(); hipEventRecord(e1, 0); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); best = ms < best ? ms : best;
    }
    printf("%-44s %7.1f us\n", name, best * 1e3);
  };
  hipFuncSetAttribute((const void*)k_bge_sample<4, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
  time("k_bge_sample, no kernel-matrix blocks", [&] {
    hipLaunchKernelGGL((k_bge_sample<4, true>), dim3(13, M), dim3(256), lds, 0, d_thr, d_masks, d_ns, bp, Key2{1, 2}, 0, M, d, S, W, 0, qs, KmatFuse{nullptr, nullptr, 0, 0, 0, 0.f, 0.f}); });
  time("k_bge_sample + kernel-matrix blocks", [&] {
    hipLaunchKernelGGL((k_bge_sample<4, true>), dim3(13 + 8, M), dim3(256), ldsk > lds ? ldsk : lds, 0, d_thr, d_masks, d_ns, bp, Key2{1, 2}, 0, M, d, S, W, 0, qs, KmatFuse{d_z, d_k, 5000, M, 13, 1.f, 5.f}); });
  time("Threefry calls only (same count)", [&] { hipLaunchKernelGGL(k_threefry_only, dim3(13, M), dim3(256), 0, 0, Key2{1, 2}, M, d, S, d_out); });
  return 0;
}
