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
//  tests/test_gpu_parity.py).
// h only needs the DIAGONAL of M_pa^-1: on the rows with p_i > 0, M_pa = D (R + Lam) D with Lam = D^-2 - I, so with B = (R + Lam)^-1
//   h_i = (1 / p_i) sum_b B_ib R~_ib = (1 / p_i) ((B R)_ii - B_ii) = (1 / p_i) (1 - B_ii (Lam_ii + 1)) = (1 - (M_pa^-1)_ii) / p_i
// (h_i = 0 where p_i = 0), and (M_pa^-1)_ii = |column i of L^-1|^2.  Lane i therefore solves L u = e_i (its column of L^-1; forward
// substitution only), lane j solves L w = b in the same pass, and y_i = u_i . w.  1 - (M_pa^-1)_ii is assembled from L_ii^2 - 1 (kept
// from the factorisation without the 1) and the off-diagonal part of the column: no cancellation for small p_i.
// One wave per node (lane = matrix row; two rows per lane for d > 64), one block per (particle, sample).  L (column-major) and the
// columns of L^-1 are packed triangles in LDS: 2 * d (d + 1) / 2 floats per wave.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "kernels_joint.h"
#include <stdlib.h>

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

__host__ __device__ inline int bge_soft_tri(int d) { return d * (d + 1) / 2; }
__host__ __device__ inline size_t bge_soft_wave_bytes(int d) {
  // L tri | U tri | p[128] | y[128] | w[128] | dinv[128]
  return (((size_t)2 * bge_soft_tri(d) + 4 * 128) * 4 + 15) & ~(size_t)15;
}
__host__ __device__ inline size_t bge_soft_shared_bytes(int d, bool r_in_lds) {
  return ((((size_t)d * d * (r_in_lds ? 1 : 0)) * 4 + 15) & ~(size_t)15) + 256;  // Rs | red[4] (+ pad)
}
__host__ __device__ inline int bge_soft_waves(int d, bool r_in_lds) {
  const size_t shared = bge_soft_shared_bytes(d, r_in_lds);
  if (shared + bge_soft_wave_bytes(d) > (size_t)160 * 1024 - 1024) return 0;
  const int nw = (int)(((size_t)160 * 1024 - 1024 - shared) / bge_soft_wave_bytes(d));
  return nw > 4 ? 4 : nw;
}
__host__ __device__ inline size_t bge_soft_lds_bytes(int d, bool r_in_lds) {
  return bge_soft_shared_bytes(d, r_in_lds) + (size_t)bge_soft_waves(d, r_in_lds) * bge_soft_wave_bytes(d);
}

// the fence between a wave's writes of the factor / inverse columns and the other lanes' reads: LDS, or (GLOB) global memory of this CU
template <bool GLOB>
__device__ __forceinline__ void bge_soft_fence() {
  if (GLOB) __threadfence_block();  // (the stores have reached the L2: write-through)
  wave_lds_fence();
}
// an entry of the packed triangles.  GLOB: read past the vector L1 -- a line may have been cached before a neighbouring entry (the next
// column, or the previous problem's factor) was written, and a store does not refresh it
template <bool GLOB>
__device__ __forceinline__ float bge_soft_ld(const float* p) {
  return GLOB ? __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : *p;
}
// GLOB (129 <= d <= 256, round 6): the two packed triangles of a wave (2 x 32 896 floats at d = 256) live in global scratch instead of LDS,
// four rows per lane, persistent blocks that walk the (sample, particle) pairs -- the same arithmetic, untuned: it exists so that
// grad_estimator_z = "reparam" has the n_vars limit of everything else.
__host__ __device__ inline size_t bge_soft_glob_lds_bytes() { return 256 + 4 * (size_t)4 * 256 * 4; }  // red | p, y, w, dinv [256] per wave
// grid = (S, Mloc) (GLOB: (blocks, 1)), block = 256.  RPL: matrix rows per lane (1: d <= 64, 2: d <= 128, 4: d <= 256)
template <bool R_LDS, int RPL, bool GLOB = false>
__global__ __launch_bounds__(256) void k_bge_soft(const float* __restrict__ scores, BgeSoftParams bp, Key2 carry, int m0, int M_global,
                                                  int d, int S, float alpha, float tau, int layout, int tiny,
                                                  float* __restrict__ ds_out, float* __restrict__ logprobs, float* __restrict__ tri_glob,
                                                  int n_prob) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int dd = d * d, ntri = bge_soft_tri(d);
  constexpr int VL = GLOB ? 256 : 128;  // length of the per-wave vectors
  float* Rs = reinterpret_cast<float*>(smem_raw);             // R (one matrix for all nodes)
  double* red = reinterpret_cast<double*>(GLOB ? smem_raw : smem_raw + bge_soft_shared_bytes(d, R_LDS) - 256);  // [4] partial log-probs (+ pad)
  const int nw = GLOB ? 4 : bge_soft_waves(d, R_LDS);
  unsigned char* wbase = GLOB ? smem_raw + 256 + (size_t)wave * 4 * VL * 4
                              : smem_raw + bge_soft_shared_bytes(d, R_LDS) + (size_t)wave * bge_soft_wave_bytes(d);
  // L, column-major packed: (r, q), r >= q, at q d - q (q - 1) / 2 + r - q;  columns of L^-1, row-major packed: (k, c), c <= k, at k (k + 1) / 2 + c
  float* Lc = GLOB ? tri_glob + ((size_t)blockIdx.x * 4 + wave) * 2 * ntri : reinterpret_cast<float*>(wbase);
  float* Ut = Lc + ntri;
  float* pv = GLOB ? reinterpret_cast<float*>(wbase) : Ut + ntri;
  float* yv = pv + VL;
  float* wv = yv + VL;
  float* dinv = wv + VL;
  if (R_LDS)
    for (int e = tid; e < dd; e += 256) Rs[e] = bp.R[e];
  for (int prob = GLOB ? (int)blockIdx.x : 0; prob < (GLOB ? n_prob : 1); prob += GLOB ? (int)gridDim.x : 1) {
  const int s = GLOB ? prob % S : (int)blockIdx.x, m = GLOB ? prob / S : (int)blockIdx.y;
  const float* sc_m = scores + (size_t)m * dd;
  const Key2 key = lin_mode_key(LIN_MODE_Z_REPARAM, carry, M_global, m0 + m, layout);  // dibs.py:430-431
  const uint64_t nbits = (uint64_t)S * dd;
  if (tid < 4) red[tid] = 0.0;
  __syncthreads();

  float* out = ds_out + ((size_t)m * S + s) * dd;
  double lp_wave = 0.0;
  if (wave < nw) {
    for (int j = wave; j < d; j += nw) {
      const float* R = R_LDS ? Rs : bp.R + (bp.n_mats > 1 ? (size_t)j * dd : 0);
      // column j of the soft graph of this sample (dibs.py:121-140), drawn by the lanes that own its rows
      float p[RPL], g[RPL];
      bool act[RPL];
      double lsum = 0.0;
#pragma unroll
      for (int h = 0; h < RPL; ++h) {
        const int r = lane + 64 * h;
        act[h] = r < d;
        g[h] = act[h] ? lin_sample_g(LIN_MODE_Z_REPARAM, key, nbits, (uint64_t)dd, s, r, j, d, nullptr, sc_m, alpha, tau, layout, tiny) : 0.f;
        p[h] = g[h];  // (0 on the diagonal)
        pv[r] = p[h];
        lsum += (double)p[h];
      }
      const double l = wave_sum_d(lsum);
      bge_soft_fence<GLOB>();
      // ---- Cholesky of M_pa, column by column (lane = row) -------------------------------------------------
      float dm1[RPL];  // L_rr^2 - 1 of this lane's rows
      double ldp = 0.0;
#pragma unroll
      for (int h = 0; h < RPL; ++h) dm1[h] = 0.f;
      for (int kk = 0; kk < d; ++kk) {
        const float pk = pv[kk];
        float acc[RPL];
#pragma unroll
        for (int h = 0; h < RPL; ++h) {
          const int r = lane + 64 * h;
          acc[h] = 0.f;
          if (act[h] && r >= kk) {
            acc[h] = p[h] * pk * (R[r * d + kk] - (r == kk ? 1.f : 0.f));   // M_pa[r][kk] - delta
            int ir = r, ik = kk;                                            // (r, q) and (kk, q) of column q = 0
#pragma unroll 8
            for (int q = 0; q < kk; ++q) {  // (unrolled: the LDS reads of several terms in flight)
              acc[h] = fmaf(-bge_soft_ld<GLOB>(Lc + ir), bge_soft_ld<GLOB>(Lc + ik), acc[h]);
              ir += d - q - 1;
              ik += d - q - 1;
            }
          }
        }
        float accp = acc[0];
#pragma unroll
        for (int h = 1; h < RPL; ++h) accp = (kk >> 6) == h ? acc[h] : accp;  // (wave-uniform select of the pivot row's block)
        const float pivm1 = __shfl(accp, kk & 63, 64);
        const float piv = 1.0f + pivm1;
        const float inv = rsqrtf(piv);
        const int ck = kk * d - kk * (kk - 1) / 2 - kk;  // (r, kk) at ck + r
#pragma unroll
        for (int h = 0; h < RPL; ++h) {
          const int r = lane + 64 * h;
          if (r == kk) {
            dm1[h] = pivm1;
            dinv[kk] = inv;
            Lc[ck + r] = piv * inv;
            ldp += log((double)piv);
          } else if (act[h] && r > kk) {
            Lc[ck + r] = acc[h] * inv;
          }
        }
        bge_soft_fence<GLOB>();
      }
      const double ld_pa = wave_sum_d(ldp);
      // ---- forward substitutions: lane c solves L u = e_c (column c of L^-1), the lane that owns row j solves L w = b instead ----
      float b_own[RPL];
#pragma unroll
      for (int h = 0; h < RPL; ++h) {
        const int r = lane + 64 * h;
        b_own[h] = act[h] ? p[h] * R[j * d + r] : 0.f;
        wv[r] = b_own[h];  // (rhs, overwritten by the solution row by row)
      }
      bge_soft_fence<GLOB>();
      for (int k = 0; k < d; ++k) {
        const int tk = k * (k + 1) / 2;
        float vs[RPL];
        float vw = wv[k];  // rhs b_k (every lane computes w_k: one broadcast chain, no lane divergence)
        int ik = k;        // (k, q) of column q = 0
#pragma unroll
        for (int h = 0; h < RPL; ++h) vs[h] = (lane + 64 * h == k) ? 1.f : 0.f;
#pragma unroll 8
        for (int q = 0; q < k; ++q) {
          const float lkq = bge_soft_ld<GLOB>(Lc + ik);
          ik += d - q - 1;
          vw = fmaf(-lkq, wv[q], vw);
#pragma unroll
          for (int h = 0; h < RPL; ++h) {
            const int c = lane + 64 * h;
            if (c <= q) vs[h] = fmaf(-lkq, bge_soft_ld<GLOB>(Ut + q * (q + 1) / 2 + c), vs[h]);
          }
        }
        const float di = dinv[k];
#pragma unroll
        for (int h = 0; h < RPL; ++h) {
          const int c = lane + 64 * h;
          if (c <= k) Ut[tk + c] = vs[h] * di;
        }
        if (lane == 0) wv[k] = vw * di;
        bge_soft_fence<GLOB>();
      }
      // ---- (M_pa^-1)_cc and y_c = (L^-T w)_c from this lane's column of L^-1 ------------------------------------
      float offd[RPL], y[RPL];
#pragma unroll
      for (int h = 0; h < RPL; ++h) {
        const int c = lane + 64 * h;
        offd[h] = 0.f;
        y[h] = 0.f;
        if (act[h]) {
          for (int k = c; k < d; ++k) {
            const float u = bge_soft_ld<GLOB>(Ut + k * (k + 1) / 2 + c);
            if (k > c) offd[h] = fmaf(u, u, offd[h]);
            y[h] = fmaf(u, wv[k], y[h]);
          }
          if (c == j) y[h] = 0.f;  // row j of M_pa is the identity and b_j = 0
        }
        yv[c] = y[h];
      }
      double btyp = 0.0;
#pragma unroll
      for (int h = 0; h < RPL; ++h) btyp += (double)b_own[h] * (double)y[h];
      const double bty = wave_sum_d(btyp);
      const double sch = (double)R[j * d + j] - bty;
      bge_soft_fence<GLOB>();
      const double Nn = bp.Nj[j], al = bp.alpha_lambd;
      double lj = 0.0, gprime = 0.0, ls = 0.0, c2 = 0.0;
      if (Nn > 0.0) {  // linearGaussian.py:118: a node without observations scores 0
        const double a1 = 0.5 * (Nn + al - d + l + 1.0), a2 = 0.5 * (al - d + l + 1.0);
        c2 = a1;
        const double gam = 0.5 * (log(bp.alpha_mu) - log(Nn + bp.alpha_mu)) + lgamma(a1) - lgamma(a2) - 0.5 * Nn * log(M_PI) +
                           0.5 * (al - d + 2.0 * l + 1.0) * bp.log_t;
        gprime = 0.5 * digamma_d(a1) - 0.5 * digamma_d(a2) + bp.log_t;
        ls = log(sch);
        lj = gam - 0.5 * ld_pa - c2 * ls;
      }
      lp_wave += lj;
#pragma unroll
      for (int h = 0; h < RPL; ++h) {
        const int r = lane + 64 * h;
        if (!act[h]) continue;
        float t_r = 0.f;
#pragma unroll 8
        for (int bb = 0; bb < d; ++bb) t_r = fmaf(R[r * d + bb] - (bb == r ? 1.f : 0.f), pv[bb] * yv[bb], t_r);
        t_r -= R[j * d + r];
        // 1 - (M_pa^-1)_rr = (L_rr^2 - 1) / L_rr^2 - |off-diagonal part of column r of L^-1|^2, each term O(p_r^2)
        const float h_r = p[h] > 0.f ? (dm1[h] / (1.0f + dm1[h]) - offd[h]) / p[h] : 0.f;
        const double dl = Nn > 0.0 ? gprime - 0.5 * ls - (double)h_r - (2.0 * c2 / sch) * (double)y[h] * (double)t_r : 0.0;
        out[r * d + j] = (r == j) ? 0.f : (float)dl * tau * alpha * g[h] * (1.0f - g[h]);
      }
      bge_soft_fence<GLOB>();
    }
    if (lane == 0) red[wave] = lp_wave;
  }
  __syncthreads();
  if (tid == 0) logprobs[(size_t)m * S + s] = (float)(red[0] + red[1] + red[2] + red[3]);
  __syncthreads();  // (GLOB: red is reset for the block's next problem)
  }
}

#ifdef DIBS_TU_BGE_SOFT
#include "kernels_bge_soft_mf.h"
// softmax over the samples and W = sum_s w_s dS_s (samples with w_s == 0 in float are skipped, in sample order)
// grid = (Mloc, ceil(d*d / 256)), block = 256: every block of a particle evaluates the S weights itself (128 exponentials) and sums one
// 256-element slice of the S x d x d gradients -- with one block per particle half of the CUs stood idle behind 164 MB of reads (126 us).
// dynamic LDS = S * 4 + 16
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
  for (int e = blockIdx.y * 256 + tid; e < dd; e += 256 * gridDim.y) {
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
                     float tau, int layout, int tiny, float* soft_ds, float* logprobs, float* w_lik, hipStream_t stream, float* tri_glob,
                     int glob_blocks) {
  if (d > 128) {  // packed triangles in global scratch (tri_glob: glob_blocks * 4 waves * 2 * tri(d) floats), persistent blocks
    hipLaunchKernelGGL((k_bge_soft<false, 4, true>), dim3(glob_blocks), dim3(256), bge_soft_glob_lds_bytes(), stream, scores, sp, carry, m0, M, d, S,
                       alpha, tau, layout, tiny, soft_ds, logprobs, tri_glob, S * Mloc);
    hipLaunchKernelGGL(k_soft_combine, dim3(Mloc, (d * d + 255) / 256), dim3(256), (size_t)S * 4 + 16, stream, soft_ds, logprobs, w_lik, d, S);
    return;
  }
  const bool rl = sp.n_mats == 1 && bge_soft_waves(d, true) >= (bge_soft_waves(d, false) < 4 ? bge_soft_waves(d, false) : 4);
  const size_t lds = bge_soft_lds_bytes(d, rl);
#define SOFT_LAUNCH(RL_, RPL_)                                                                                                          \
  {                                                                                                                                     \
    if (lds > 48 * 1024) hipFuncSetAttribute((const void*)k_bge_soft<RL_, RPL_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
    hipLaunchKernelGGL((k_bge_soft<RL_, RPL_>), dim3(S, Mloc), dim3(256), lds, stream, scores, sp, carry, m0, M, d, S, alpha, tau, layout, \
                       tiny, soft_ds, logprobs, (float*)nullptr, 1);                                                                    \
  }
  if (d <= 64) {
    // blocked factorisation on the matrix pipe (kernels_bge_soft_mf.h)
    const bool rr = sp.n_mats == 1;
    const size_t l2 = bsm_lds_bytes(d, rr);
#define SOFTM(NB_, RL_, W_)                                                                                                             \
  {                                                                                                                                     \
    if (l2 > 48 * 1024) hipFuncSetAttribute((const void*)k_bge_soft_mf<NB_, RL_, W_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)l2); \
    hipLaunchKernelGGL((k_bge_soft_mf<NB_, RL_, W_>), dim3(S, Mloc), dim3(256), l2, stream, scores, sp, carry, m0, M, d, S, alpha, tau, layout, \
                       tiny, soft_ds, logprobs);                                                                                        \
  }
    // (three waves per SIMD: 165 registers, no scratch -- 4.6 ms against 5.4 at two)
    switch ((d + 15) / 16) {
      case 1: if (rr) SOFTM(1, true, 3) else SOFTM(1, false, 3) break;
      case 2: if (rr) SOFTM(2, true, 3) else SOFTM(2, false, 3) break;
      case 3: if (rr) SOFTM(3, true, 3) else SOFTM(3, false, 3) break;
      default: if (rr) SOFTM(4, true, 3) else SOFTM(4, false, 3) break;
    }
#undef SOFTM
  } else {
    if (rl) SOFT_LAUNCH(true, 2) else SOFT_LAUNCH(false, 2)
  }
#undef SOFT_LAUNCH
  hipLaunchKernelGGL(k_soft_combine, dim3(Mloc, (d * d + 255) / 256), dim3(256), (size_t)S * 4 + 16, stream, soft_ds, logprobs, w_lik, d, S);
}
#else
void bge_soft_launch(const BgeSoftParams& sp, const float* scores, Key2 carry, int m0, int M, int Mloc, int d, int S, float alpha,
                     float tau, int layout, int tiny, float* soft_ds, float* logprobs, float* w_lik, hipStream_t stream, float* tri_glob,
                     int glob_blocks);
#endif  // DIBS_TU_BGE_SOFT
