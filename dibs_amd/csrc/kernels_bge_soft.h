// MarginalDiBS + BGe with the reparameterised (Gumbel-soft graph) estimator  --  grad_estimator_z = "reparam".
//   reference: dibs/inference/dibs.py:395-459 (grad_z_likelihood_gumbel), :271-288 (log_joint_prob_soft),
//              dibs/models/linearGaussian.py:63-170 with a real-valued parent vector, dibs/utils/func.py:128-145.
// For a soft graph G~ and node j let p = G~[:, j] (p_j = 0), l = sum p, D = diag(p), R~ = R - I.  The masked matrices are
//   M_pa  = I + D R~ D                     (row / column j is the identity)
//   M_all = the same with p_j := 1         =>  logdet M_all = logdet M_pa + log s,   s = R_jj - b^T M_pa^-1 b,  b_k = p_k R_jk
// so ONE factorisation M_pa = L L^T per node gives both determinants, and with  y = M_pa^-1 b,
//   h_i = (M_pa^-1 o R~) p |_i = sum_b (M_pa^-1)_ib R~_ib p_b        (d/dp_i logdet M_pa = 2 h_i)
//   ds/dp_i = 2 y_i ( (R~ (p o y))_i - R_ji )
//   l_j(p)   = gamma(l) - 1/2 logdet M_pa - c2 log s,       c2 = (N + alpha_lambd - d + l + 1) / 2
//   dl_j/dp_i = gamma'(l) - 1/2 log s - h_i - (2 c2 / s) y_i ( (R~ (p o y))_i - R_ji )        (i != j)
//   gamma'(l) = 1/2 psi((N + alpha_lambd - d + l + 1)/2) - 1/2 psi((alpha_lambd - d + l + 1)/2) + log t
// (autograd of the reference formula; checked against torch.autograd in tests/test_oracle.py and on the device in
//  tests/test_gpu_parity.py).  h needs the diagonal of M_pa^-1 Q with Q_bi = R~_ib p_b: lane i solves L L^T w = Q[:, i] by
// forward / backward substitution (L is read as LDS broadcasts) and keeps w_i; lane j solves for y instead.
// One wave per node, one block per (particle, sample); d <= 64 (one matrix row per lane).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "kernels_joint.h"

struct BgeSoftParams {
  const float* R;     // [n_mats, d, d]
  const double* Nj;   // [d]
  double alpha_lambd, alpha_mu, log_t;
  int n_mats;
};

__device__ __forceinline__ double digamma_d(double x) {
  double acc = 0.0;
  while (x < 8.0) {
    acc -= 1.0 / x;
    x += 1.0;
  }
  const double f = 1.0 / (x * x);
  // asymptotic series: ln x - 1/(2x) - sum B_2n / (2n x^2n)
  const double ser = f * (1.0 / 12.0 - f * (1.0 / 120.0 - f * (1.0 / 252.0 - f * (1.0 / 240.0 - f * (1.0 / 132.0)))));
  return acc + log(x) - 0.5 / x - ser;
}

__host__ __device__ inline size_t bge_soft_wave_bytes(int d) {
  // L [d][d|1] | U [d][64] | p[64] | y[64] | dinv[64]
  return ((size_t)d * (d | 1) + (size_t)d * 64 + 3 * 64) * 4;
}
__host__ __device__ inline size_t bge_soft_shared_bytes(int d, bool r_in_lds) {
  return ((((size_t)d * d * (r_in_lds ? 2 : 1)) * 4 + 15) & ~(size_t)15) + 256;  // Gs | Rs | red[4] (+ pad)
}
__host__ __device__ inline int bge_soft_waves(int d, bool r_in_lds) {
  const size_t shared = bge_soft_shared_bytes(d, r_in_lds);
  const int nw = (int)(((size_t)160 * 1024 - 1024 - shared) / bge_soft_wave_bytes(d));
  return nw > 4 ? 4 : nw;
}
__host__ __device__ inline size_t bge_soft_lds_bytes(int d, bool r_in_lds) {
  return bge_soft_shared_bytes(d, r_in_lds) + (size_t)bge_soft_waves(d, r_in_lds) * bge_soft_wave_bytes(d);
}

// grid = (S, Mloc), block = 256
template <bool R_LDS>
__global__ __launch_bounds__(256) void k_bge_soft(const float* __restrict__ scores, BgeSoftParams bp, Key2 carry, int m0, int M_global,
                                                  int d, int S, float alpha, float tau, int layout, int tiny,
                                                  float* __restrict__ ds_out, float* __restrict__ logprobs) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int s = blockIdx.x, m = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int dd = d * d, ldl = d | 1;
  float* Gs = reinterpret_cast<float*>(smem_raw);             // soft graph of this sample, [i][j]
  float* Rs = Gs + dd;                                        // R (one matrix for all nodes)
  double* red = reinterpret_cast<double*>(smem_raw + bge_soft_shared_bytes(d, R_LDS) - 256);  // [4] partial log-probs (+ pad)
  const int nw = bge_soft_waves(d, R_LDS);
  unsigned char* wbase = smem_raw + bge_soft_shared_bytes(d, R_LDS) + (size_t)wave * bge_soft_wave_bytes(d);
  float* L = reinterpret_cast<float*>(wbase);
  float* U = L + (size_t)d * ldl;
  float* pv = U + (size_t)d * 64;
  float* yv = pv + 64;
  float* dinv = yv + 64;

  const float* sc_m = scores + (size_t)m * dd;
  const Key2 key = lin_mode_key(LIN_MODE_Z_REPARAM, carry, M_global, m0 + m, layout);  // dibs.py:430-431
  const uint64_t nbits = (uint64_t)S * dd;
  for (int e = tid; e < dd; e += 256) {
    const int i = e / d, j = e - i * d;
    Gs[e] = lin_sample_g(LIN_MODE_Z_REPARAM, key, nbits, (uint64_t)dd, s, i, j, d, nullptr, sc_m, alpha, tau, layout, tiny);
    if (R_LDS) Rs[e] = bp.R[e];
  }
  if (tid < 4) red[tid] = 0.0;
  __syncthreads();

  float* out = ds_out + ((size_t)m * S + s) * dd;
  double lp_wave = 0.0;
  if (wave < nw) {
    for (int j = wave; j < d; j += nw) {
      const float* R = R_LDS ? Rs : bp.R + (bp.n_mats > 1 ? (size_t)j * dd : 0);
      const int r = lane;
      const bool act = r < d;
      const float p_r = (act && r != j) ? Gs[r * d + j] : 0.f;
      pv[lane] = p_r;
      const double l = wave_sum_d((double)p_r);
      wave_lds_fence();
      // ---- Cholesky of M_pa (lane = row) --------------------------------------------------------
      float mypiv = 1.f;
      for (int kk = 0; kk < d; ++kk) {
        float acc = 0.f;
        if (act && r >= kk) {
          const float dlt = r == kk ? 1.f : 0.f;
          acc = dlt + p_r * pv[kk] * (R[r * d + kk] - dlt);
          const float* lr = L + (size_t)r * ldl;
          const float* lk = L + (size_t)kk * ldl;
#pragma unroll 8
          for (int q = 0; q < kk; ++q) acc = fmaf(-lr[q], lk[q], acc);  // (unrolled: the LDS reads of several terms in flight)
        }
        const float piv = __shfl(acc, kk, 64);
        const float inv = rsqrtf(piv);
        if (r == kk) {
          mypiv = piv;
          dinv[kk] = inv;
        }
        if (act && r > kk) L[(size_t)r * ldl + kk] = acc * inv;
        wave_lds_fence();
      }
      const double ld_pa = wave_sum_d(act ? log((double)mypiv) : 0.0);
      // ---- forward substitution  L u = rhs(lane):  rhs_b = p_b (R[lane][b] - [b == lane != j]) ------------------
      if (act) {
        for (int k = 0; k < d; ++k) {
          const float dlt = (k == r && r != j) ? 1.f : 0.f;
          float v = pv[k] * (R[r * d + k] - dlt);
          const float* lk = L + (size_t)k * ldl;
#pragma unroll 8
          for (int q = 0; q < k; ++q) v = fmaf(-lk[q], U[q * 64 + lane], v);
          U[k * 64 + lane] = v * dinv[k];
        }
        // ---- backward substitution  L^T w = u, down to row `stop` (lane j needs all of y, the others only w_lane) -----
        const int stop = r == j ? 0 : r;
        for (int k = d - 1; k >= stop; --k) {
          float v = U[k * 64 + lane];
#pragma unroll 8
          for (int q = k + 1; q < d; ++q) v = fmaf(-L[(size_t)q * ldl + k], U[q * 64 + lane], v);
          U[k * 64 + lane] = v * dinv[k];
        }
      }
      wave_lds_fence();
      const float h_r = act ? U[r * 64 + r] : 0.f;   // w_r of this lane's own system
      const float y_r = act ? U[r * 64 + j] : 0.f;   // y = M_pa^-1 b (lane j's system), row r
      yv[lane] = y_r;
      const double bty = wave_sum_d(act ? (double)(p_r * R[j * d + r]) * (double)y_r : 0.0);
      const double sch = (double)R[j * d + j] - bty;
      wave_lds_fence();
      float t_r = 0.f;
      if (act) {
#pragma unroll 8
        for (int b = 0; b < d; ++b) t_r = fmaf(R[r * d + b] - (b == r ? 1.f : 0.f), pv[b] * yv[b], t_r);
        t_r -= R[j * d + r];
      }
      const double Nn = bp.Nj[j], al = bp.alpha_lambd;
      double lj = 0.0, dl = 0.0;
      if (Nn > 0.0) {  // linearGaussian.py:118: a node without observations scores 0
        const double a1 = 0.5 * (Nn + al - d + l + 1.0), a2 = 0.5 * (al - d + l + 1.0), c2 = a1;
        const double gam = 0.5 * (log(bp.alpha_mu) - log(Nn + bp.alpha_mu)) + lgamma(a1) - lgamma(a2) - 0.5 * Nn * log(M_PI) +
                           0.5 * (al - d + 2.0 * l + 1.0) * bp.log_t;
        const double gprime = 0.5 * digamma_d(a1) - 0.5 * digamma_d(a2) + bp.log_t;
        const double ls = log(sch);
        lj = gam - 0.5 * ld_pa - c2 * ls;
        dl = gprime - 0.5 * ls - (double)h_r - (2.0 * c2 / sch) * (double)y_r * (double)t_r;
      }
      lp_wave += lj;
      if (act) {
        const float g = Gs[r * d + j];
        out[r * d + j] = (r == j) ? 0.f : (float)dl * tau * alpha * g * (1.0f - g);
      }
      wave_lds_fence();
    }
    if (lane == 0) red[wave] = lp_wave;
  }
  __syncthreads();
  if (tid == 0) logprobs[(size_t)m * S + s] = (float)(red[0] + red[1] + red[2] + red[3]);
}

#ifdef DIBS_TU_BGE_SOFT
// softmax over the samples and W = sum_s w_s dS_s (samples with w_s == 0 in float are skipped, in sample order)
// grid = Mloc, block = 256; dynamic LDS = S * 8
__global__ __launch_bounds__(256) void k_soft_combine(const float* __restrict__ ds, const float* __restrict__ logprobs,
                                                      float* __restrict__ w_lik, int d, int S) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  float* wt = reinterpret_cast<float*>(smem_raw);
  __shared__ double red[8];
  const int m = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* lp = logprobs + (size_t)m * S;
  double mx = -INFINITY;
  for (int s = tid; s < S; s += 256) mx = (double)lp[s] > mx ? (double)lp[s] : mx;
  mx = wave_max_d(mx);
  if (lane == 0) red[wave] = mx;
  __syncthreads();
  mx = red[0];
  for (int w = 1; w < 4; ++w) mx = red[w] > mx ? red[w] : mx;
  double den = 0.0;
  for (int s = tid; s < S; s += 256) den += exp((double)lp[s] - mx);
  den = wave_sum_d(den);
  if (lane == 0) red[4 + wave] = den;
  __syncthreads();
  den = red[4] + red[5] + red[6] + red[7];
  for (int s = tid; s < S; s += 256) wt[s] = (float)(exp((double)lp[s] - mx) / den);
  __syncthreads();
  const int dd = d * d;
  for (int e = tid; e < dd; e += 256) {
    float acc = 0.f;
    for (int s = 0; s < S; ++s) {
      const float w = wt[s];
      if (w != 0.f) acc = fmaf(w, ds[((size_t)m * S + s) * dd + e], acc);
    }
    w_lik[(size_t)m * dd + e] = acc;
  }
}

// both launches of the estimator: per-sample soft-graph scores + gradients, then the softmax-weighted combination
void bge_soft_launch(const BgeSoftParams& sp, const float* scores, Key2 carry, int m0, int M, int Mloc, int d, int S, float alpha,
                     float tau, int layout, int tiny, float* soft_ds, float* logprobs, float* w_lik, hipStream_t stream) {
  const bool rl = sp.n_mats == 1 && bge_soft_waves(d, true) >= 1;
  const size_t lds = bge_soft_lds_bytes(d, rl);
  if (rl) {
    if (lds > 48 * 1024) hipFuncSetAttribute((const void*)k_bge_soft<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(k_bge_soft<true>, dim3(S, Mloc), dim3(256), lds, stream, scores, sp, carry, m0, M, d, S, alpha, tau, layout, tiny,
                       soft_ds, logprobs);
  } else {
    if (lds > 48 * 1024) hipFuncSetAttribute((const void*)k_bge_soft<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(k_bge_soft<false>, dim3(S, Mloc), dim3(256), lds, stream, scores, sp, carry, m0, M, d, S, alpha, tau, layout, tiny,
                       soft_ds, logprobs);
  }
  hipLaunchKernelGGL(k_soft_combine, dim3(Mloc), dim3(256), (size_t)S * 4 + 16, stream, soft_ds, logprobs, w_lik, d, S);
}
#else
void bge_soft_launch(const BgeSoftParams& sp, const float* scores, Key2 carry, int m0, int M, int Mloc, int d, int S, float alpha,
                     float tau, int layout, int tiny, float* soft_ds, float* logprobs, float* w_lik, hipStream_t stream);
#endif  // DIBS_TU_BGE_SOFT
