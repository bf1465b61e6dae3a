// JointDiBS likelihood kernels (LinearGaussian) -- see DESIGN.md.  (stub: filled in next milestone)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "rng.h"

struct JointWork {
  float* xtx;  // [n_mats, d, d]  x^T diag(1 - mask_j) x
  int n_mats;
};

struct JointLaunch {
  hipStream_t stream;
  const float* z;
  const float* theta;
  const float* scores;
  const uint32_t* thr;
  float* w_lik;
  float* logprobs_z;
  float* logprobs_th;
  float* baseline;
  float* pack;
  size_t pack_stride, theta_off, gtheta_off;
  int m0, M, Mloc, d, N, S;
  float alpha, tau;
  int layout, tiny, est_z;
  double sf_baseline;
  float obs_noise, mean_edge, sig_edge;
};

static inline int joint_alloc(JointWork* w, int Mloc, int d, int N, int S) { (void)w; (void)Mloc; (void)d; (void)N; (void)S; return 0; }
static inline void joint_free(JointWork* w) { (void)w; }
static inline int joint_set_data(JointWork* w, const float* x, const int32_t* mask, int N, int d) { (void)w; (void)x; (void)mask; (void)N; (void)d; return 1; }
static inline void joint_lin_theta(JointWork* w, const JointLaunch& jl, Key2 carry) { (void)w; (void)jl; (void)carry; }
static inline void joint_lin_z(JointWork* w, const JointLaunch& jl, Key2 carry) { (void)w; (void)jl; (void)carry; }
