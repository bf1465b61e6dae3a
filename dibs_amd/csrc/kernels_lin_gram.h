// JointDiBS + LinearGaussian, GRAM-MATRIX path: any number of observations.   reference: dibs/models/linearGaussian.py:278-338
// The tuned kernels of kernels_joint.h keep x [N, d] in LDS and form x (g o theta) per sample on MFMA; they stop where x no longer
// fits (about 290 observations at d = 50).  The likelihood only needs second moments of x:
//   sum_{n not intervened on j} (x_nj - (x w_j)_n)^2 = C_jj - 2 w_j^T C_:j + w_j^T C w_j,     C = C^(j) = sum_n x_n x_n^T over those n,
//   x^T r_j = (C_:j - C w_j) / obs_noise                                                    (w_j = g[:, j] o theta[:, j])
// so one d x d (interventions: d of them) double-precision Gram matrix, built once per data set on the host, replaces x.  The
// quadratic form cancels (terms ~ N var(x), result ~ N obs_noise): it is evaluated in double on the vector ALU.  Cost per sample:
// d^3 FMA for soft graphs, sum_j l_j d for hard ones -- independent of N.
// (included by kernels_joint.h inside its translation unit)
#pragma once

// v[a][j] = sum_b C^(j)[a][b] w[b][j] for the pairs this thread owns; calls f(a, j, v, c_aj)
template <typename F>
__device__ __forceinline__ void ling_cw(const double* __restrict__ Cs, const double* __restrict__ gram, int n_gram, const float* WG, int d,
                                        int tid, bool need_all, F&& f) {
  for (int e = tid; e < d * d; e += 256) {
    const int j = e / d, a = e - j * d;  // (j-major: the threads of a wave share the column of W and, with interventions, the matrix)
    if (!need_all && WG[a * d + j] == 0.f) continue;
    const double* Ca = (n_gram == 1 ? Cs : gram + (n_gram > 1 ? (size_t)j * d * d : 0)) + (size_t)a * d;  // (n_gram == -1: one matrix, read through the caches)
    double v = 0.0;
    for (int b = 0; b < d; ++b) {
      const float w = WG[b * d + j];
      if (w != 0.f) v = fma(Ca[b], (double)w, v);
    }
    f(a, j, v, Ca[j]);
  }
}

// ------------------------------------------------------------------------------------------------
// log p(theta, D | G_s), one sample per block.  grid = (S, Mloc) [mode GIVEN: (1, n graphs)], block = 256
// dynamic LDS = d*d*4 (W) + (n_gram == 1 ? d*d*8 : 0) (C) + 64 (ops_glob: 256, grid.x <= S);  n_gram: 1 = one Gram matrix, LDS-resident; d = one per node (interventions);
// -1 = one matrix that does not fit LDS beside the operands (d > 100), read through the caches
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_ling_logprobs(const double* __restrict__ gram, const double* __restrict__ ncnt, int n_gram,
                                                       const float* __restrict__ theta, const float* __restrict__ scores,
                                                       const uint32_t* __restrict__ thr, float* __restrict__ logprobs, Key2 carry, int mode,
                                                       int m0, int M_global, int d, int S, float alpha, float tau, int layout, int tiny,
                                                       float obs_noise, float mu, float sig, float* __restrict__ ops_glob) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  double* Cs = reinterpret_cast<double*>(smem_raw);
  const size_t dd = (size_t)d * d;
  // ops_glob != null (n_vars > 198): the masked weights of this block in global scratch [gridDim.y][gridDim.x][d*d] instead of LDS (the block's
  // own barriers order its writes and reads), LDS holds the reduction slots only; the block then loops over the samples s, s + gridDim.x, ...
  float* WG = ops_glob ? ops_glob + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * dd : reinterpret_cast<float*>(smem_raw + (n_gram == 1 ? dd * 8 : 0));
  double* red = ops_glob ? reinterpret_cast<double*>(smem_raw)
                         : reinterpret_cast<double*>(smem_raw + (n_gram == 1 ? dd * 8 : 0) + ((dd * 4 + 15) & ~(size_t)15));
  const int m = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* TH = theta + (size_t)m * dd;
  const Key2 key = (mode == LIN_MODE_GIVEN) ? Key2{0, 0} : lin_mode_key(mode, carry, M_global, m0 + m, layout);
  const uint64_t nbits = (uint64_t)S * dd;
  const uint32_t* thr_m = thr + (size_t)m * dd;
  const float* sc_m = scores ? scores + (size_t)m * dd : nullptr;
  if (n_gram == 1)
    for (int e = tid; e < (int)dd; e += 256) Cs[e] = gram[e];
  for (int s = blockIdx.x; s < S; s += gridDim.x) {
  double part = 0.0;
  for (int e = tid; e < (int)dd; e += 256) {
    const int a = e / d, j = e - a * d;
    const float gv = lin_sample_g(mode, key, nbits, dd, s, a, j, d, thr_m, sc_m, alpha, tau, layout, tiny);
    const float th = TH[e];
    WG[e] = gv * th;
    part += (double)(gv * lin_logn(th, mu, sig));  // prior: sum_ij g_ij logN(theta_ij)   (linearGaussian.py:278-289)
  }
  __syncthreads();
  const double inv2 = 0.5 / (double)obs_noise;
  const double lognorm_x = -0.5 * log((double)obs_noise) - 0.918938533204672742;
  // sum_j [ N_j lognorm - (C_jj - 2 w.c + w.Cw) / (2 obs_noise) ]
  for (int j = tid; j < d; j += 256) {
    const double cjj = (n_gram == 1 ? Cs : gram + (n_gram > 1 ? (size_t)j * dd : 0))[(size_t)j * d + j];
    part += ncnt[j] * lognorm_x - inv2 * cjj;
  }
  ling_cw(Cs, gram, n_gram, WG, d, tid, false, [&](int a, int j, double v, double caj) {
    part -= inv2 * (double)WG[a * d + j] * (v - 2.0 * caj);
  });
  const double tot = wave_sum_d(part);
  if (lane == 0) red[wave] = tot;
  __syncthreads();
  if (tid == 0) logprobs[(size_t)m * S + s] = (float)(red[0] + red[1] + red[2] + red[3]);
  __syncthreads();  // (the operands and the reduction slots are rewritten by the next sample of this block)
  }
}

// ------------------------------------------------------------------------------------------------
// softmax-weighted gradients, same contract (LinGradJob, blockIdx.y selects the job) as k_lin_grad, same sharing of a particle's weighted
// samples between blocks (GradSplit, kernels_joint.h).  grid = (Mloc, 2, shares), block = 256; block (x, y, z) takes share z of particle
// (x + z) mod Mloc.  Accumulation goes to a row in global memory -- the output itself while one block does everything, else the block's
// partial row -- and every element is owned by one thread.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_ling_grad(const double* __restrict__ gram, int n_gram, const float* __restrict__ theta,
                                                   const float* __restrict__ scores, const uint32_t* __restrict__ thr, LinGradJob job0,
                                                   LinGradJob job1, const float* __restrict__ baseline, int m0, int M_global, int d, int S,
                                                   float alpha, float tau, int layout, int tiny, float obs_noise, float mu, float sig,
                                                   double sf_baseline, float* __restrict__ ops_glob, GradSplit gs) {
  const LinGradJob job = blockIdx.y ? job1 : job0;
  const int mode = job.mode;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  double* Cs = reinterpret_cast<double*>(smem_raw);
  const size_t dd = (size_t)d * d;
  // (ops_glob != null, n_vars > 141: masked weights and graph of this block in global scratch [gridDim.y][gridDim.x][2][d*d], see k_ling_logprobs)
  float* WG = ops_glob ? ops_glob + (((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 2 * dd
                       : reinterpret_cast<float*>(smem_raw + (n_gram == 1 ? dd * 8 : 0));
  float* GS = WG + dd;
  double* red = ops_glob ? reinterpret_cast<double*>(smem_raw)
                         : reinterpret_cast<double*>(smem_raw + (n_gram == 1 ? dd * 8 : 0) + ((2 * dd * 4 + 15) & ~(size_t)15));
  const int m = (int)((blockIdx.x + blockIdx.z) % gridDim.x), tid = threadIdx.x;
  const float* TH = theta + (size_t)m * dd;
  float* const om_final = job.out + (size_t)m * job.out_stride;
  const Key2 key = lin_mode_key(mode, job.carry, M_global, m0 + m, layout);
  const uint64_t nbits = (uint64_t)S * dd;
  const float* lp = job.logprobs + (size_t)m * S;
  __shared__ float wch[GRAD_WCH];
  __shared__ int last_flag;
  double mx, den, sm;
  int nnz;
  grad_softmax_stats(lp, S, red, mx, den, sm, nnz);
  const int bz = blockIdx.z, NS = gridDim.z, nact = nnz < NS ? (nnz > 0 ? nnz : 1) : NS;
  if (bz >= nact) return;  // (block-uniform: no share)
  float* const om = nact > 1 ? gs.part + ((size_t)(m * 2 + (int)blockIdx.y) * NS + bz) * gs.stride : om_final;
  if (n_gram == 1)
    for (int e = tid; e < (int)dd; e += 256) Cs[e] = gram[e];
  for (int e = tid; e < (int)dd; e += 256) om[e] = 0.f;
  const double inv_on = 1.0 / (double)obs_noise;
  const float* sc_m = scores + (size_t)m * dd;
  const uint32_t* thr_m = thr + (size_t)m * dd;
  int q = 0;  // ordinal of the next weighted sample
  for (int s0 = 0; s0 < S; s0 += GRAD_WCH) {
    __syncthreads();
    if (s0 + tid < S) wch[tid] = (float)(exp((double)lp[s0 + tid] - mx) / den);
    __syncthreads();
  for (int s = s0; s < S && s < s0 + GRAD_WCH; ++s) {
    const float w = wch[s - s0];
    if (w < GRAD_W_MIN) continue;  // block-uniform
    if ((q++ % NS) != bz) continue;  // (another block's sample)
    __syncthreads();
    for (int e = tid; e < (int)dd; e += 256) {
      const int a = e / d, j = e - a * d;
      const float gv = lin_sample_g(mode, key, nbits, dd, s, a, j, d, thr_m, sc_m, alpha, tau, layout, tiny);
      GS[e] = gv;
      WG[e] = gv * TH[e];
    }
    __syncthreads();
    if (mode == LIN_MODE_Z_SCORE) {
      for (int e = tid; e < (int)dd; e += 256) om[e] += w * GS[e];
      continue;
    }
    // theta mode: prior part g (-(theta - mu) / sig^2) for every element, then the likelihood part g x^T r where g theta != 0; reparam
    // mode: d/dg needs x^T r everywhere.  (ling_cw's owner of an element is a different thread than the row-major loops': barrier.)
    if (mode == LIN_MODE_THETA) {
      for (int e = tid; e < (int)dd; e += 256) om[e] += w * GS[e] * (-(TH[e] - mu) / (sig * sig));
      __syncthreads();
    }
    const bool theta_mode = mode == LIN_MODE_THETA;
    ling_cw(Cs, gram, n_gram, WG, d, tid, !theta_mode, [&](int a, int j, double v, double caj) {
      const float xtr = (float)((caj - v) * inv_on);
      const int e = a * d + j;
      if (theta_mode) {
        om[e] += w * GS[e] * xtr;
      } else if (a != j) {
        const float th = TH[e], gv = GS[e];
        om[e] += w * (lin_logn(th, mu, sig) + th * xtr) * tau * alpha * gv * (1.0f - gv);
      }
    });
  }
  }
  __syncthreads();
  if (nact > 1) {  // (plain read-modify-writes above: release, count this block; the last one adds the rows in block order)
    __threadfence();
    if (!grad_last_block(gs.ctr + (m * 2 + (int)blockIdx.y), nact, &last_flag)) return;
    __threadfence();
    const float* const base = gs.part + (size_t)(m * 2 + (int)blockIdx.y) * NS * gs.stride;
    for (int e = tid; e < (int)dd; e += 256) om_final[e] = grad_part_sum<GRAD_NS, false>(base, gs.stride, (size_t)e, nact);
    __syncthreads();
  }
  const float bold = baseline ? baseline[m] : 0.f;
  if (mode == LIN_MODE_Z_SCORE) {
    const float scale = sf_baseline > 0.0 ? (float)exp(-(double)bold) : 1.0f;
    for (int e = tid; e < (int)dd; e += 256) {
      const int i = e / d, j = e - i * d;
      const float p = (float)sigmoid_d((double)__fmul_rn(alpha, sc_m[e]));
      om_final[e] = i == j ? 0.f : scale * alpha * (om_final[e] - p);
    }
  }
  if (job.theta_copy)
    for (int e = tid; e < (int)dd; e += 256) job.theta_copy[(size_t)m * job.out_stride + e] = TH[e];
  if (mode != LIN_MODE_THETA && job.baseline_out && tid == 0)
    job.baseline_out[m] = (mode == LIN_MODE_Z_SCORE) ? (float)(sf_baseline * (sm / S) + (1.0 - sf_baseline) * (double)bold) : bold;
}
