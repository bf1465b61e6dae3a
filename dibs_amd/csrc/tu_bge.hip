// translation unit: BGe sampling / factorisation kernels and their launchers (kernels_bge.h)
#define DIBS_TU_BGE
#include "launch.h"
#include <stdlib.h>

template <typename K>
static void allow_lds(K kernel, size_t bytes) { dibs_allow_lds((const void*)kernel, bytes); }

size_t bge_sample_lds_bytes(int d, int S, int W) { return 4 * bge_sample_wave_bytes(d, S, W); }

void bge_launch_sample(bool sample, hipStream_t stream, const uint32_t* thr, uint64_t* masks, double* node_scores, const BgeParams& bp,
                       Key2 carry, int m0, int M, int Mloc, int d, int S, int W, int layout, const BgeQueues& qs, const KmatFuse& kf) {
  size_t lds = bge_sample_lds_bytes(d, S, W);
  const int nbx = (d + 3) / 4;
  int extra = 0;
  if (kf.z) {
    const size_t kl = (size_t)kf.len * 4 + 64;
    lds = lds > kl ? lds : kl;
    extra = (kf.M + KMAT_BT - 1) / KMAT_BT;
  }
  const dim3 grid(nbx + extra, Mloc);
  if (sample) {
    allow_lds(k_bge_sample<4, true>, lds);
    hipLaunchKernelGGL((k_bge_sample<4, true>), grid, dim3(256), lds, stream, thr, masks, node_scores, bp, carry, m0, M, d, S, W, layout, qs, kf);
  } else {
    allow_lds(k_bge_sample<4, false>, lds);
    hipLaunchKernelGGL((k_bge_sample<4, false>), grid, dim3(256), lds, stream, thr, masks, node_scores, bp, carry, m0, M, d, S, W, layout, qs, kf);
  }
}

void bge_launch_chol(hipStream_t stream, double* node_scores, const BgeParams& bp, const BgeQueues& qs, int d, int S,
                     unsigned long long* counters) {
  if (d > 128) {  // three or four mask words: one problem per wave, everything in the last tier (k_bge_chol_wide)
    const size_t lw = BGE_WIDE_WAVES * bge_wide_wave_bytes(d);
    allow_lds(k_bge_chol_wide, lw);
    hipLaunchKernelGGL(k_bge_chol_wide, dim3(2048), dim3(64 * BGE_WIDE_WAVES), lw, stream, node_scores, bp, qs, d, (d + 63) / 64);
    return;
  }
  const bool w2 = d > 64;
  bool rl = bp.n_mats == 1;
  if (rl && bge_chol_lds_bytes(d, true) > (size_t)150 * 1024) rl = false;
  const size_t lds = bge_chol_lds_bytes(d, rl);
  const dim3 grid(512), block(256);  // (512 = the resident blocks at two waves per SIMD: R / Q are staged once per slot; 1024: +2 us)
#define CHOL(RL_, W2_)                                                                                                  \
  {                                                                                                                     \
    allow_lds(k_bge_chol<RL_, W2_>, lds);                                                                               \
    hipLaunchKernelGGL((k_bge_chol<RL_, W2_>), grid, block, lds, stream, node_scores, bp, qs, d, S, counters);                    \
  }
  if (rl && !w2) CHOL(true, false)
  else if (rl) CHOL(true, true)
  else if (!w2) CHOL(false, false)
  else CHOL(false, true)
#undef CHOL
}

void bge_launch_sum_nodes(hipStream_t stream, const double* node_scores, float* out, int d, int S) {
  hipLaunchKernelGGL(k_sum_nodes, dim3((S + 127) / 128), dim3(128), 0, stream, node_scores, out, d, S);
}
