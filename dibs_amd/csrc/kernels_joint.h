// JointDiBS + LinearGaussian likelihood kernels (gfx950).
//   log p(theta, D | G) = sum_ij g_ij logN(theta_ij; mu_e, sig_e) + sum_{n,j: not intervened} logN(x_nj; (x (g o theta))_nj, sqrt(obs_noise))
//   r = (1 - mask) o (x - x (g o theta)) / obs_noise
//   d/dg = logN(theta) + theta o (x^T r)            d/dtheta = g o (-(theta - mu_e)/sig_e^2 + x^T r)
// reference: dibs/models/linearGaussian.py:278-338; estimators dibs/inference/dibs.py:395-459 (Z, reparam),
//            :325-391 (Z, score), :488-551 (theta).  Both contractions run on v_mfma_f32_16x16x4_f32 with x, theta and
//            the per-sample operand resident in LDS.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "rng.h"
#include "common.h"
#include "kernels_acyc_bf16.h"
#include "kernels_acyc_f16.h"

enum { LIN_MODE_THETA = 0, LIN_MODE_Z_SCORE = 1, LIN_MODE_Z_REPARAM = 2, LIN_MODE_GIVEN = 3 };

#define GRAD_NS 8   // (16 measured: config 3 at step 1 500 281 us against 271, at step 5 54 against 49)
#define GRAD_NS_NN 16  // DenseNonlinearGaussian: a sample's gradient takes ~0.7 ms of a block -- finer shares balance the CUs better
#define GRAD_WCH 256  // weights are evaluated in chunks of this many samples (one double-precision exp per thread and chunk)
// A sample is differentiated when its softmax weight is at least 2^-30.  The oracle (as round 5's kernels) keeps every weight that is
// non-zero in float32, down to 1e-45; but a term w_s grad_s with w_s < 2^-30 is 64 times below the float32 resolution of the sum it is added to
// (the weights sum to 1 and the samples' gradients are of one magnitude) -- the reference's own float32 logsumexp loses it the same way -- and
// late in a run of the LinearGaussian model such samples are the majority of the "weighted" ones: config 3 at step 1 500 has 12.4 samples per
// particle with a non-zero weight in the theta estimator (worst particle 54) and 4.2 (24) with a weight of at least 2^-30 -- k_lin_grad
// 271 -> 154 us; config 5 at step 300 keeps 16.2 of 16.7 (the weights of its saturated graphs really are spread): profiles/round6_late_breakdown.txt.
#define GRAD_W_MIN 9.313225746154785e-10f
struct GradSplit {
  float* part;         // [jobs][GRAD_NS][stride] partial sums
  unsigned int* ctr;   // [jobs] arrivals (zero between launches: the last block resets it)
  size_t stride;
};

// work list of a persistent gradient kernel (k_grad_plan below): per job the softmax statistics of its samples, and the (job, share) items
// that have at least one weighted sample
struct GradPlan {
  double* stats;        // [jobs][4]: maximum, sum of exponentials, sum of the log-probabilities, number of weighted samples
  unsigned int* items;  // [<= jobs * shares]: job * 64 + share
  unsigned int* ctr;    // [0]: number of items, [1]: next item to take  (this launch's pair of the two the workspace keeps)
  unsigned int* ctr_next;  // the other pair: zeroed by this launch's plan kernel for the next launch (no memset launch between the steps)
};

struct JointWork {
  float* x;        // [N, d] device copy
  int32_t* mask;   // [N, d]
  float* wsm;      // [Mloc, S] softmax weights scratch
  float* ln_tab;   // [Mloc, d, d] DenseNN: per-particle first-layer prior table (kernels_nn.h), else null
  float* w1t;      // [Mloc, H, d, d] DenseNN fast path: first-layer weights re-laid out per hidden unit, W1T[h][a][j] = W1[j][a][h]
  size_t w1t_floats;
  int any_mask;
  double* gram;    // LinearGaussian Gram path (kernels_lin_gram.h): C^(j) [n_gram][d][d], observations not intervened on j
  double* ncnt;    // [d] their count
  int n_gram;      // 1 without interventions, else d; 0: not built
  void* nhf_w1s;   // DenseNN, f16 matrix pipe (kernels_nn_f16.h): scaled first-layer weights per column pair (float2) [Mloc][H][d][ceil(d/2)]
  void* nhf_w1p;   // ... and their packed f16 pieces {h pair, m pair} (uint2), same shape
  int* nhf_ew;     // [Mloc] exponent of the per-particle scale
  size_t nhf_pairs;  // allocated pairs (0: not allocated)
  void *nhx_w1s, *nhx_w1p;  // the same per a-QUAD and node (float4 / uint4) [Mloc][H][ceil(d/4)][d]: k_nn_logprobs_hx (kernels_nn_f16x.h)
  size_t nhx_quads;
  // which table set belongs to the CURRENT theta: both are cleared by the theta pass (first estimator of a step) and set by whichever variant
  // builds its tables, so that an estimator pass that takes the other variant than the theta pass did (the LDS size depends on soft / hard
  // graphs) builds its own instead of reading stale or uninitialised tables
  bool nhf_valid, nhx_valid;
  float* nng_scratch;         // general DenseNN path (kernels_nn_generic.h): activation records, grown on first use
  size_t nng_scratch_floats;
  // general paths beyond the LDS capacity (LinearGaussian Gram kernels: n_vars > 141 / 198, DenseNN general kernels: n_vars > 198): the
  // sampled graph (and the masked weights) of a block live here instead of in LDS; grown on first use
  float* gs_scratch;
  size_t gs_scratch_floats;
  // gradient kernels with several blocks per (particle, estimator) (GradSplit below): partial sums and arrival counters, grown on first use
  float* gpart;
  size_t gpart_floats;
  unsigned int* gctr;
  size_t gctr_n;
  GradPlan gplan;     // persistent gradient kernels: statistics + item list, grown on first use
  size_t gplan_jobs, gplan_items;
  unsigned int gplan_gen;  // launches so far: the counter pair in use alternates
};
static inline float* joint_gs_scratch(JointWork* w, size_t floats) {
  if (w->gs_scratch_floats < floats) {
    if (w->gs_scratch) hipFree(w->gs_scratch);
    w->gs_scratch = nullptr;
    w->gs_scratch_floats = 0;
    if (hipMalloc((void**)&w->gs_scratch, floats * 4) != hipSuccess) return nullptr;
    w->gs_scratch_floats = floats;
  }
  return w->gs_scratch;
}
// partial-sum area of the split gradient kernels: `jobs` (particle, estimator) pairs x GRAD_NS blocks x `stride` floats; counters zeroed once
static inline bool joint_grad_split(JointWork* w, size_t jobs, size_t stride, GradSplit* out, int ns = GRAD_NS) {
  const size_t need = jobs * (size_t)ns * stride;
  if (w->gpart_floats < need) {
    if (w->gpart) hipFree(w->gpart);
    w->gpart = nullptr;
    w->gpart_floats = 0;
    if (hipMalloc((void**)&w->gpart, need * 4) != hipSuccess) return false;
    w->gpart_floats = need;
  }
  if (w->gctr_n < jobs) {
    if (w->gctr) hipFree(w->gctr);
    w->gctr = nullptr;
    w->gctr_n = 0;
    if (hipMalloc((void**)&w->gctr, jobs * 4) != hipSuccess) return false;
    if (hipMemset(w->gctr, 0, jobs * 4) != hipSuccess) return false;
    if (hipDeviceSynchronize() != hipSuccess) return false;  // (the engine's streams do not wait for the null stream)
    w->gctr_n = jobs;
  }
  *out = GradSplit{w->gpart, w->gctr, stride};
  return true;
}

static inline bool joint_grad_plan(JointWork* w, size_t jobs, int ns, GradPlan* out) {
  if (w->gplan_jobs < jobs || w->gplan_items < jobs * (size_t)ns) {
    if (w->gplan.stats) hipFree(w->gplan.stats);
    if (w->gplan.items) hipFree(w->gplan.items);
    if (w->gplan.ctr) hipFree(w->gplan.ctr);
    w->gplan = GradPlan{nullptr, nullptr, nullptr, nullptr};
    w->gplan_jobs = w->gplan_items = 0;
    if (hipMalloc((void**)&w->gplan.stats, jobs * 4 * sizeof(double)) != hipSuccess) return false;
    if (hipMalloc((void**)&w->gplan.items, jobs * (size_t)ns * 4) != hipSuccess) return false;
    if (hipMalloc((void**)&w->gplan.ctr, 16) != hipSuccess) return false;
    if (hipMemset(w->gplan.ctr, 0, 16) != hipSuccess) return false;
    if (hipDeviceSynchronize() != hipSuccess) return false;  // (the engine's streams do not wait for the null stream)
    w->gplan_jobs = jobs;
    w->gplan_items = jobs * (size_t)ns;
    w->gplan_gen = 0;
  }
  // two counter pairs: launch k uses pair k & 1 and its plan kernel zeroes the other one for launch k + 1 (launches of a workspace are
  // ordered in one stream)
  *out = w->gplan;
  out->ctr = w->gplan.ctr + 2 * (w->gplan_gen & 1u);
  out->ctr_next = w->gplan.ctr + 2 * ((w->gplan_gen + 1u) & 1u);
  ++w->gplan_gen;
  return true;
}

struct JointLaunch {
  hipStream_t stream;
  const float* z;
  const float* theta;
  const float* scores;
  const uint32_t* thr;
  float* w_lik;
  float* logprobs_z;
  float* logprobs_th;
  const float* baseline;
  float* baseline_out;
  float* pack;
  size_t pack_stride, theta_off, gtheta_off;
  int copy_theta;  // 1: packed rows carry a copy of theta at theta_off; 0: gradient rows only
  int m0, M, Mloc, d, N, S;
  float alpha, tau;
  int layout, tiny, est_z;
  double sf_baseline;
  float obs_noise, mean_edge, sig_edge;
  int lin_f32 = 0, nn_f32 = 0;  // the engine's DibsTuning (tuning.h): keep the f32-MFMA log-probability kernels (A/B runs)
};

struct LinGeom {
  int d, N, kp, np, ldx, ldw;  // kp = ceil4(d), np = ceil16(N) (rows of x / res incl. zero padding)
};
__host__ __device__ inline LinGeom lin_geom(int d, int N, int NT) {
  LinGeom g;
  g.d = d;
  g.N = N;
  g.kp = (d + 3) & ~3;
  g.np = (N + 15) & ~15;
  const int dp = 16 * NT;
  g.ldx = dp + 2;  // x rows / res rows: (row * ld + k) bank pattern of the MFMA A/B fragment reads
  g.ldw = dp + 2;
  return g;
}
// LDS: X[np][ldx] | WG[kp][ldw] | (gradient kernel) RS[np][ldw]; theta is read from global (L2-resident, d*d floats per particle)
__host__ __device__ inline size_t lin_lds_bytes(int d, int N, int NT, bool with_res) {
  const LinGeom g = lin_geom(d, N, NT);
  size_t f = (size_t)g.np * g.ldx + (size_t)g.kp * g.ldw;
  if (with_res) f += (size_t)g.np * g.ldw;
  return ((f * 4 + 15) & ~(size_t)15) + 64 * 8;
}

__device__ __forceinline__ float lin_logn(float v, float mu, float sig) {
  const float zt = (v - mu) / sig;
  return -0.5f * zt * zt - logf(sig) - 0.918938533204672742f;
}

// element (i, j) of sample s: hard Bernoulli graph (theta / score modes) or Gumbel-soft graph (reparam)
__device__ __forceinline__ float lin_sample_g(int mode, Key2 key, uint64_t nbits, uint64_t dd, int s, int i, int j, int d,
                                              const uint32_t* __restrict__ thr_m, const float* __restrict__ sc_m, float alpha,
                                              float tau, int layout, int tiny) {
  if (mode == LIN_MODE_GIVEN) return reinterpret_cast<const int32_t*>(thr_m)[i * d + j] != 0 ? 1.0f : 0.0f;  // caller's graph
  if (i == j) return 0.f;
  const uint32_t bits = rng_bits_at(key, nbits, (uint64_t)s * dd + (uint64_t)i * d + j, layout);
  if (mode == LIN_MODE_Z_REPARAM) {
    const float eps = rng_logistic(bits, tiny);
    return 1.0f / (1.0f + expf(-tau * (eps + alpha * sc_m[i * d + j])));
  }
  return (bits >> 9) < thr_m[i * d + j] ? 1.0f : 0.0f;
}

__device__ __forceinline__ Key2 lin_mode_key(int mode, Key2 carry, int M_global, int m_global, int layout) {
  const Key2 kp = rng_split_row_uniform(carry, (uint32_t)M_global + 1u, (uint32_t)m_global + 1u, layout);
  if (mode == LIN_MODE_THETA) return kp;            // dibs.py:510: particle key itself
  return rng_split_row_uniform(kp, 2u, 1u, layout); // dibs.py:350-351 / 430-431: subk_ of split(particle key)
}

template <int NT>
__device__ __forceinline__ void lin_load_common(float* X, const float* __restrict__ x, const LinGeom g, int tid) {
  for (int e = tid; e < g.np * g.ldx; e += 256) {
    const int n = e / g.ldx, c = e - n * g.ldx;
    X[e] = (n < g.N && c < g.d) ? x[(size_t)n * g.d + c] : 0.f;
  }
}

// WG = g o theta for sample s (zero padded); returns this thread's share of sum_ij g_ij logN(theta_ij)
template <int NT>
__device__ __forceinline__ float lin_build_wg(float* WG, const float* __restrict__ TH, int mode, Key2 key, uint64_t nbits, int s,
                                              const uint32_t* thr_m, const float* sc_m, float alpha, float tau, int layout,
                                              int tiny, float mu, float sig, const LinGeom g, int tid) {
  float prior = 0.f;
  const uint64_t dd = (uint64_t)g.d * g.d;
  for (int e = tid; e < g.kp * g.ldw; e += 256) {
    const int i = e / g.ldw, j = e - i * g.ldw;
    float v = 0.f;
    if (i < g.d && j < g.d) {
      const float gv = lin_sample_g(mode, key, nbits, dd, s, i, j, g.d, thr_m, sc_m, alpha, tau, layout, tiny);
      const float th = TH[i * g.d + j];
      v = gv * th;
      prior += gv * lin_logn(th, mu, sig);
    }
    WG[e] = v;
  }
  return prior;
}

// pred = X * WG for the row tiles of this wave; calls f(n, j, pred_nj) on every valid element
template <int NT, typename F>
__device__ __forceinline__ void lin_pred_tiles(const float* X, const float* WG, const LinGeom g, int lane, int wave, F&& f) {
  const int nrt = g.np >> 4;
  for (int ti = wave; ti < nrt; ti += 4) {
    f32x4 acc[NT];
#pragma unroll
    for (int tj = 0; tj < NT; ++tj) acc[tj] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int ap = (ti * 16 + (lane & 15)) * g.ldx + (lane >> 4);
    const int bq = (lane >> 4) * g.ldw + (lane & 15);
    for (int k0 = 0; k0 < g.kp; k0 += 4) {
      const float a = X[ap + k0];
#pragma unroll
      for (int tj = 0; tj < NT; ++tj) acc[tj] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, WG[bq + k0 * g.ldw + tj * 16], acc[tj], 0, 0, 0);
    }
#pragma unroll
    for (int tj = 0; tj < NT; ++tj)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = ti * 16 + (lane >> 4) * 4 + r, j = tj * 16 + (lane & 15);
        float v = acc[tj][r];
        asm volatile("" : "+v"(v));
        if (n < g.N && j < g.d) f(n, j, v);
      }
  }
}

// ------------------------------------------------------------------------------------------------
// log p(theta, D | G_s) for all samples.  grid = (ceil(S / spb), Mloc), block = 256
// ------------------------------------------------------------------------------------------------
template <int NT>
__global__ __launch_bounds__(256) void k_lin_logprobs(const float* __restrict__ x, const int32_t* __restrict__ mask,
                                                      const float* __restrict__ theta, const float* __restrict__ scores,
                                                      const uint32_t* __restrict__ thr, float* __restrict__ logprobs, Key2 carry,
                                                      int mode, int m0, int M_global, int d, int N, int S, int spb, float alpha,
                                                      float tau, int layout, int tiny, float obs_noise, float mu, float sig,
                                                      int any_mask) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const LinGeom g = lin_geom(d, N, NT);
  float* X = smem;
  float* WG = X + (size_t)g.np * g.ldx;
  double* red = reinterpret_cast<double*>(smem + ((((size_t)g.np * g.ldx + (size_t)g.kp * g.ldw) + 3) & ~(size_t)3));
  const int m = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const size_t dd = (size_t)d * d;
  const float* __restrict__ TH = theta + (size_t)m * dd;
  lin_load_common<NT>(X, x, g, tid);
  const Key2 key = lin_mode_key(mode, carry, M_global, m0 + m, layout);
  const uint64_t nbits = (uint64_t)S * dd;
  const float inv2 = 0.5f / obs_noise;
  const float lognorm_x = -0.5f * logf(obs_noise) - 0.918938533204672742f;
  for (int c = 0; c < spb; ++c) {
    const int s = blockIdx.x * spb + c;
    if (s >= S) break;
    __syncthreads();
    float part = lin_build_wg<NT>(WG, TH, mode, key, nbits, s, thr + (size_t)m * dd, scores + (size_t)m * dd, alpha, tau, layout,
                                  tiny, mu, sig, g, tid);
    __syncthreads();
    lin_pred_tiles<NT>(X, WG, g, lane, wave, [&](int n, int j, float pred) {
      if (any_mask && mask[(size_t)n * d + j]) return;
      const float e = X[n * g.ldx + j] - pred;
      part += lognorm_x - inv2 * e * e;
    });
    const double tot = wave_sum_d((double)part);
    if (lane == 0) red[wave] = tot;
    __syncthreads();
    if (tid == 0) logprobs[(size_t)m * S + s] = (float)(red[0] + red[1] + red[2] + red[3]);
  }
}

// Same, for the legacy PRNG layout with an even number of samples and N <= 128: sample s and s + S/2 share their Threefry
// calls (element e of the [S, d, d] draw is paired with e + S d d / 2), so a block takes both and builds both operands from
// one call per element.  x does not depend on the sample: every wave keeps its MFMA A fragments (and the x values of its
// output elements) in registers, so LDS holds the two per-sample operands only (row stride == 16 mod 32: conflict-free
// B-fragment reads) and four blocks fit on a CU.
// grid = (ceil(S / 2 / ppb), Mloc), block = 256
template <int NT>
__host__ __device__ constexpr int lin_ldw2() { return (NT & 1) ? 16 * NT : 16 * NT + 16; }
__host__ __device__ inline size_t lin_lds_bytes_pair(int d, int NT) {
  const int kp = (d + 3) & ~3, ldw2 = (NT & 1) ? 16 * NT : 16 * NT + 16;
  return (((size_t)2 * kp * ldw2 * 4 + 15) & ~(size_t)15) + 64 * 8;
}
// EPQ > 0: every thread owns the elements e = tid + 256 q (q < EPQ, covers d*d <= 256 EPQ) of the d x d operand and keeps
// their sample-independent factors (theta, logN(theta), exp(-alpha s) or the Bernoulli threshold, LDS offset) in registers
// for all pairs of the block; EPQ == 0 recomputes them per pair (large d: the registers go to the x fragments instead).
template <int NT, int EPQ>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(NT <= 4 ? 3 : 1, NT <= 4 ? 3 : 2))) void k_lin_logprobs_pair(const float* __restrict__ x, const int32_t* __restrict__ mask,
                                                           const float* __restrict__ theta, const float* __restrict__ scores,
                                                           const uint32_t* __restrict__ thr, float* __restrict__ logprobs, Key2 carry,
                                                           int mode, int m0, int M_global, int d, int N, int S, int ppb, float alpha,
                                                           float tau, int layout, int tiny, float obs_noise, float mu, float sig,
                                                           int any_mask) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int LDW = lin_ldw2<NT>(), NU = 2, KSMAX = 4 * NT;
  const int kp = (d + 3) & ~3, ksteps = kp >> 2, nrt = (N + 15) >> 4;
  float* WG0 = smem;
  float* WG1 = WG0 + (size_t)kp * LDW;
  double* red = reinterpret_cast<double*>(smem + (((size_t)2 * kp * LDW + 3) & ~(size_t)3));
  const int m = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int dd = d * d;
  const float* __restrict__ TH = theta + (size_t)m * dd;
  const uint32_t* __restrict__ thr_m = thr + (size_t)m * dd;
  const float* __restrict__ sc_m = scores + (size_t)m * dd;
  // A fragments: row n = (wave + 4u) * 16 + (lane & 15), k = 4 ks + (lane >> 4); output elements (C layout):
  // n = (wave + 4u) * 16 + (lane >> 4) * 4 + r, j = tj * 16 + (lane & 15); wgt = 1 where the element counts in the likelihood
  float xa[NU][KSMAX], xe[NU][NT][4];
  uint32_t ok[NU];
  float nvalid = 0.f;
#pragma unroll
  for (int u = 0; u < NU; ++u) {
    const int na = (wave + 4 * u) * 16 + (lane & 15);
#pragma unroll
    for (int ks = 0; ks < KSMAX; ++ks) {
      const int kk = 4 * ks + (lane >> 4);
      xa[u][ks] = (na < N && kk < d) ? x[(size_t)na * d + kk] : 0.f;
    }
    ok[u] = 0u;
#pragma unroll
    for (int tj = 0; tj < NT; ++tj)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = (wave + 4 * u) * 16 + (lane >> 4) * 4 + r, j = tj * 16 + (lane & 15);
        const bool v = n < N && j < d && !(any_mask && mask[(size_t)n * d + j]);
        xe[u][tj][r] = v ? x[(size_t)n * d + j] : 0.f;
        ok[u] |= (uint32_t)v << (tj * 4 + r);
        nvalid += v ? 1.0f : 0.0f;
      }
  }
  const TfKeys tk = tf_keys(lin_mode_key(mode, carry, M_global, m0 + m, layout));
  const uint32_t half = (uint32_t)(((uint64_t)S * dd) >> 1);
  const int hS = S >> 1;
  const float inv2 = 0.5f / obs_noise;
  const float lognorm_x = -0.5f * logf(obs_noise) - 0.918938533204672742f;
  const bool soft = mode == LIN_MODE_Z_REPARAM, fast = tau == 1.0f;
  const float ulo = tiny ? 1.17549435e-38f : 1.1920929e-07f;
  const int bq = (lane >> 4) * LDW + (lane & 15);
  const float inv_d = 1.0f / (float)d;
  for (int e = tid; e < 2 * kp * LDW; e += 256) smem[e] = 0.f;  // padding and diagonal: written once
  // sample-independent factors of element e: offset in the operand, theta, logN(theta), aux = exp(-alpha s) | alpha s | thr
  // (aux carries the Bernoulli threshold's bits in the hard-graph modes)
  auto factors = [&](int e, int& off, float& th, float& ln, float& aux) {
    const int i = (int)(((float)e + 0.5f) * inv_d), j = e - i * d;  // exact for e < 2^20
    off = (i == j) ? -1 : i * LDW + j;
    th = TH[e];
    ln = lin_logn(th, mu, sig);
    if (soft) {
      const float as = alpha * sc_m[e];
      aux = fast ? expf(-as) : as;
    } else {
      aux = __uint_as_float(thr_m[e]);
    }
  };
  int offs[EPQ > 0 ? EPQ : 1];
  float ths[EPQ > 0 ? EPQ : 1], lns[EPQ > 0 ? EPQ : 1], auxs[EPQ > 0 ? EPQ : 1];
  if constexpr (EPQ > 0) {
#pragma unroll
    for (int q = 0; q < EPQ; ++q) {
      const int e = tid + 256 * q;
      offs[q] = -1;
      ths[q] = lns[q] = auxs[q] = 0.f;
      if (e < dd) factors(e, offs[q], ths[q], lns[q], auxs[q]);
    }
  }
  float part[2];
  // one element of the pair (s0, s0 + S/2): one Threefry call, both operands
  auto element = [&](int e, uint32_t cbase, int off, float th, float ln, float aux) {
    if (off < 0) return;
    uint32_t y0, y1;
    threefry2x32_uk(tk, cbase + (uint32_t)e, cbase + (uint32_t)e + half, y0, y1);
    float g0, g1;
    if (soft) {
      if (fast) {  // sigmoid(eps + a), eps = log(u / (1 - u))  ==  u / (u + (1 - u) exp(-a))
        const float u0 = rng_uniform(y0, ulo, 1.0f), u1 = rng_uniform(y1, ulo, 1.0f);
        g0 = u0 * __builtin_amdgcn_rcpf(fmaf(1.0f - u0, aux, u0));  // (v_rcp_f32: 1 ulp; an IEEE division is ten instructions)
        g1 = u1 * __builtin_amdgcn_rcpf(fmaf(1.0f - u1, aux, u1));
      } else {
        g0 = 1.0f / (1.0f + expf(-tau * (rng_logistic(y0, tiny) + aux)));
        g1 = 1.0f / (1.0f + expf(-tau * (rng_logistic(y1, tiny) + aux)));
      }
    } else {
      const uint32_t ta = __float_as_uint(aux);
      g0 = (y0 >> 9) < ta ? 1.0f : 0.0f;
      g1 = (y1 >> 9) < ta ? 1.0f : 0.0f;
    }
    WG0[off] = g0 * th;
    WG1[off] = g1 * th;
    part[0] = fmaf(g0, ln, part[0]);
    part[1] = fmaf(g1, ln, part[1]);
  };
  for (int c = 0; c < ppb; ++c) {
    const int s0 = blockIdx.x * ppb + c;
    if (s0 >= hS) break;
    __syncthreads();
    part[0] = part[1] = nvalid * lognorm_x;
    const uint32_t cbase = (uint32_t)((uint64_t)s0 * (uint64_t)dd);
    if constexpr (EPQ > 0) {
#pragma unroll
      for (int q = 0; q < EPQ; ++q) element(tid + 256 * q, cbase, offs[q], ths[q], lns[q], auxs[q]);
    } else {
      for (int e = tid; e < dd; e += 256) {
        int off;
        float th, ln, aux;
        factors(e, off, th, ln, aux);
        element(e, cbase, off, th, ln, aux);
      }
    }
    __syncthreads();
#pragma unroll
    for (int hsel = 0; hsel < 2; ++hsel) {
      const float* WG = hsel ? WG1 : WG0;
      float sq = 0.f;
#pragma unroll
      for (int u = 0; u < NU; ++u) {
        if (wave + 4 * u >= nrt) continue;
        f32x4 acc[NT];
#pragma unroll
        for (int tj = 0; tj < NT; ++tj) acc[tj] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KSMAX; ++ks) {
          if (ks >= ksteps) continue;
#pragma unroll
          for (int tj = 0; tj < NT; ++tj)
            acc[tj] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[u][ks], tj * 16 < LDW ? WG[bq + ks * 4 * LDW + tj * 16] : 0.f, acc[tj], 0, 0, 0);
        }
#pragma unroll
        for (int tj = 0; tj < NT; ++tj)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float pv = acc[tj][r];
            asm volatile("" : "+v"(pv));
            const float er = ((ok[u] >> (tj * 4 + r)) & 1u) ? xe[u][tj][r] - pv : 0.f;
            sq = fmaf(er, er, sq);
          }
      }
      part[hsel] = fmaf(-inv2, sq, part[hsel]);
    }
    const double t0 = wave_sum_d((double)part[0]), t1 = wave_sum_d((double)part[1]);
    if (lane == 0) {
      red[wave] = t0;
      red[4 + wave] = t1;
    }
    __syncthreads();
    if (tid == 0) logprobs[(size_t)m * S + s0] = (float)(red[0] + red[1] + red[2] + red[3]);
    if (tid == 1) logprobs[(size_t)m * S + s0 + hS] = (float)(red[4] + red[5] + red[6] + red[7]);
  }
}

// Same pairing, 33 <= d <= 64, on the f16 matrix pipe with TWO block-scaled pieces per operand (the arithmetic of k_acyc_hf,
// kernels_acyc_f16.h: x 2^e = h + m, a product three v_mfma_f32_16x16x32_f16 -- 48 MFMA cycles for a 16 x 16 x 64 block where the f32 MFMA
// needs 512).  x's row fragments (left operand, split once per block) stay in registers; the per-sample operand g o theta is split as it is
// built -- both samples of the pair in ONE packed split, low halves to the first image, high halves to the second -- and written with
// 2-byte stores into the transposing-read image layout ([piece][column tile][row k][16 columns], chunk swizzle (c + (k >> 2)) % 4).  The
// MFMA is issued with swapped operands, so lane (g, r) holds pred[n = 16 ti + r][j = 16 tj + 4 g + i].  Scales: x by the exponent of max |x| (block reduction, once),
// theta by the exponent of max |theta_m| (block reduction, once; |g| <= 1) -- the pieces stay below 2^14, the product is unscaled once
// per output element.  (Round 3's three-piece bf16 variant of this kernel was retired in round 6: profiles/HISTORY.md.)
// grid = (ceil(S / 2 / ppb), Mloc), block = 64 NW, dynamic LDS = 2 * AHF_IMG_BYTES + 256
template <int EPQ, bool FOUR, int NW>
__global__ __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu(NW == 8 ? 4 : 3, NW == 8 ? 4 : 3))) void k_lin_logprobs_hf(
    const float* __restrict__ x, const int32_t* __restrict__ mask, const float* __restrict__ theta, const float* __restrict__ scores,
    const uint32_t* __restrict__ thr, float* __restrict__ logprobs, Key2 carry, int mode, int m0, int M_global, int d, int N, int S, int ppb,
    float alpha, float tau, int layout, int tiny, float obs_noise, float mu, float sig, int any_mask) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  unsigned char* const sb = reinterpret_cast<unsigned char*>(smem);
  double* red = reinterpret_cast<double*>(sb + 2 * AHF_IMG_BYTES);
  constexpr int NU = 8 / NW, NTHR = 64 * NW;
  const int nrt = (N + 15) >> 4;
  const int m = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g4 = lane >> 4, r = lane & 15;
  const int dd = d * d;
  const float* __restrict__ TH = theta + (size_t)m * dd;
  const uint32_t* __restrict__ thr_m = thr + (size_t)m * dd;
  const float* __restrict__ sc_m = scores + (size_t)m * dd;
  // row n = (wave + 4 u) * 16 + r of x: the same 16 values (columns 16 tj + 4 g + i) are the lane's left-operand fragment and the x of its
  // output elements
  AhfFrag XA[NU];
  f32x4 xv[NU][ABF_NT];
  float xe[NU][ABF_NT][4];
  uint32_t ok[NU];
  float nvalid = 0.f, amax = 0.f;
#pragma unroll
  for (int u = 0; u < NU; ++u) {
    const int n = (wave + NW * u) * 16 + r;
    f32x4 (&v)[ABF_NT] = xv[u];
    ok[u] = 0u;
#pragma unroll
    for (int tj = 0; tj < ABF_NT; ++tj)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int j = 16 * tj + 4 * g4 + i;
        const bool inb = n < N && j < d;
        const float xv_ = inb ? x[(size_t)n * d + j] : 0.f;
        const bool valid = inb && !(any_mask && mask[(size_t)n * d + j]);
        v[tj][i] = xv_;
        xe[u][tj][i] = valid ? xv_ : 0.f;
        ok[u] |= (uint32_t)valid << (tj * 4 + i);
        nvalid += valid ? 1.0f : 0.0f;
        amax = fmaxf(amax, fabsf(xv_));
      }
  }
  // block-wide max |x| and max |theta_m| -> exponents of the two scales (pieces below 2^14)
  float tmax = 0.f;
  for (int e = tid; e < dd; e += NTHR) tmax = fmaxf(tmax, fabsf(TH[e]));
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    amax = fmaxf(amax, __shfl_xor(amax, o, 64));
    tmax = fmaxf(tmax, __shfl_xor(tmax, o, 64));
  }
  float* const redf = reinterpret_cast<float*>(red);
  if (lane == 0) {
    redf[wave] = amax;
    redf[NW + wave] = tmax;
  }
  __syncthreads();
  amax = tmax = 0.f;
#pragma unroll
  for (int w8 = 0; w8 < NW; ++w8) {
    amax = fmaxf(amax, redf[w8]);
    tmax = fmaxf(tmax, redf[NW + w8]);
  }
  auto scale_exp = [](float mx) {
    int e = 0;
    if (mx > 0.f && mx < 3.0e38f) e = 13 - ((int)((__float_as_uint(mx) >> 23) & 0xffu) - 127);
    return e > 60 ? 60 : (e < -60 ? -60 : e);
  };
  const int ex = __builtin_amdgcn_readfirstlane(scale_exp(amax)), et = __builtin_amdgcn_readfirstlane(scale_exp(tmax));
  const float th_scale = ahf_pow2(et), unscale = ahf_pow2(-(ex + et));
#pragma unroll
  for (int u = 0; u < NU; ++u) ahf_make_frag(xv[u], ahf_pow2(ex), XA[u]);
  __syncthreads();  // (red is reused by the sample loop)
  const TfKeys tk = tf_keys(lin_mode_key(mode, carry, M_global, m0 + m, layout));
  const uint32_t half = (uint32_t)(((uint64_t)S * dd) >> 1);
  const int hS = S >> 1;
  const float inv2 = 0.5f / obs_noise;
  const float lognorm_x = -0.5f * logf(obs_noise) - 0.918938533204672742f;
  const bool soft = mode == LIN_MODE_Z_REPARAM, fast = tau == 1.0f;
  const float ulo = tiny ? 1.17549435e-38f : 1.1920929e-07f;
  const int rd_off = (4 * g4 + (r >> 2)) * 32 + (((r & 3) + g4) & 3) * 8;
  const float inv_d = 1.0f / (float)d;
  for (int e = tid; e < 2 * AHF_IMG_BYTES / 16; e += NTHR) reinterpret_cast<float4*>(sb)[e] = make_float4(0.f, 0.f, 0.f, 0.f);  // padding, diagonal
  // sample-independent factors of element e = (i, j): byte offset of W[i][j] inside a piece, theta, logN(theta), aux (as k_lin_logprobs_pair)
  auto factors = [&](int e, int& off, float& th, float& ln, float& aux) {
    const int i = (int)(((float)e + 0.5f) * inv_d), j = e - i * d;  // exact for e < 2^20
    off = (i == j) ? -1 : (j >> 4) * ABF_TILE_BYTES + i * 32 + ((((j & 15) >> 2) + (i >> 2)) & 3) * 8 + (j & 3) * 2;
    th = TH[e];
    ln = lin_logn(th, mu, sig);
    th *= th_scale;  // (the operand carries theta 2^et)
    if (soft) {
      const float as = alpha * sc_m[e];
      aux = fast ? expf(-as) : as;
    } else {
      aux = __uint_as_float(thr_m[e]);
    }
  };
  int offs[EPQ > 0 ? EPQ : 1];
  float ths[EPQ > 0 ? EPQ : 1], lns[EPQ > 0 ? EPQ : 1], auxs[EPQ > 0 ? EPQ : 1];
  if constexpr (EPQ > 0) {
#pragma unroll
    for (int q = 0; q < EPQ; ++q) {
      const int e = tid + NTHR * q;
      offs[q] = -1;
      ths[q] = lns[q] = auxs[q] = 0.f;
      if (e < dd) factors(e, offs[q], ths[q], lns[q], auxs[q]);
    }
  }
  float part[2];
  auto element = [&](int e, uint32_t cbase, int off, float th, float ln, float aux) {
    if (off < 0) return;
    uint32_t y0, y1;
    threefry2x32_uk(tk, cbase + (uint32_t)e, cbase + (uint32_t)e + half, y0, y1);
    float g0, g1;
    if (soft) {
      if (fast) {
        const float u0 = rng_uniform(y0, ulo, 1.0f), u1 = rng_uniform(y1, ulo, 1.0f);
        g0 = u0 * __builtin_amdgcn_rcpf(fmaf(1.0f - u0, aux, u0));  // (v_rcp_f32: 1 ulp; an IEEE division is ten instructions)
        g1 = u1 * __builtin_amdgcn_rcpf(fmaf(1.0f - u1, aux, u1));
      } else {
        g0 = 1.0f / (1.0f + expf(-tau * (rng_logistic(y0, tiny) + aux)));
        g1 = 1.0f / (1.0f + expf(-tau * (rng_logistic(y1, tiny) + aux)));
      }
    } else {
      const uint32_t ta = __float_as_uint(aux);
      g0 = (y0 >> 9) < ta ? 1.0f : 0.0f;
      g1 = (y1 >> 9) < ta ? 1.0f : 0.0f;
    }
    uint32_t ph, pm;
    ahf_split(g0 * th, g1 * th, 1.0f, ph, pm);
    unsigned char* const w0 = sb + off;
    *reinterpret_cast<uint16_t*>(w0) = (uint16_t)ph;
    *reinterpret_cast<uint16_t*>(w0 + AHF_PIECE_BYTES) = (uint16_t)pm;
    *reinterpret_cast<uint16_t*>(w0 + AHF_IMG_BYTES) = (uint16_t)(ph >> 16);
    *reinterpret_cast<uint16_t*>(w0 + AHF_IMG_BYTES + AHF_PIECE_BYTES) = (uint16_t)(pm >> 16);
    part[0] = fmaf(g0, ln, part[0]);
    part[1] = fmaf(g1, ln, part[1]);
  };
  for (int c = 0; c < ppb; ++c) {
    const int s0 = blockIdx.x * ppb + c;
    if (s0 >= hS) break;
    __syncthreads();
    part[0] = part[1] = nvalid * lognorm_x;
    const uint32_t cbase = (uint32_t)((uint64_t)s0 * (uint64_t)dd);
    if constexpr (EPQ > 0) {
#pragma unroll
      for (int q = 0; q < EPQ; ++q) element(tid + NTHR * q, cbase, offs[q], ths[q], lns[q], auxs[q]);
    } else {
      for (int e = tid; e < dd; e += NTHR) {
        int off;
        float th, ln, aux;
        factors(e, off, th, ln, aux);
        element(e, cbase, off, th, ln, aux);
      }
    }
    __syncthreads();
#pragma unroll
    for (int hsel = 0; hsel < 2; ++hsel) {
      const unsigned char* img = sb + hsel * AHF_IMG_BYTES;
      float sq = 0.f;
#pragma unroll
      for (int u = 0; u < NU; ++u) {
        if (wave + NW * u >= nrt) continue;
        f32x4 acc[ABF_NT];
        ahf_matmul<FOUR>(acc, XA[u], img, rd_off);
        // (without interventions every element that does not count is padding: x = 0 there and the prediction is an exact 0 -- zero rows of
        //  the left operand, zero columns of the right one --, so the residual needs no mask: one instruction less per element)
        if (any_mask) {
#pragma unroll
          for (int tj = 0; tj < ABF_NT; ++tj)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              float pv = acc[tj][i];
              asm volatile("" : "+v"(pv));
              const float er = ((ok[u] >> (tj * 4 + i)) & 1u) ? fmaf(-pv, unscale, xe[u][tj][i]) : 0.f;
              sq = fmaf(er, er, sq);
            }
        } else {
#pragma unroll
          for (int tj = 0; tj < ABF_NT; ++tj)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              float pv = acc[tj][i];
              asm volatile("" : "+v"(pv));
              const float er = fmaf(-pv, unscale, xe[u][tj][i]);
              sq = fmaf(er, er, sq);
            }
        }
      }
      part[hsel] = fmaf(-inv2, sq, part[hsel]);
    }
    const double t0 = wave_sum_d((double)part[0]), t1 = wave_sum_d((double)part[1]);
    if (lane == 0) {
      red[wave] = t0;
      red[NW + wave] = t1;
    }
    __syncthreads();
    if (tid < 2) {
      double tot = 0.0;
      for (int w8 = 0; w8 < NW; ++w8) tot += red[tid * NW + w8];
      logprobs[(size_t)m * S + s0 + tid * hS] = (float)tot;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// softmax-weighted gradient: w = softmax(l); only samples with w_s != 0 are re-evaluated (in float the weights of
// all but a few samples underflow to exactly 0 -- the oracle skips them the same way).
//   mode THETA     : grad_theta = sum_s w_s g_s o (-(theta - mu)/sig^2 + x^T r_s)   -> pack row (+ copy of theta)
//   mode Z_REPARAM : W = sum_s w_s (logN(theta) + theta o x^T r_s) o tau alpha g~(1 - g~), off-diagonal   -> w_lik
//   mode Z_SCORE   : W = scale * alpha (sum_s w_s G_s - P), off-diagonal                                 -> w_lik
// grid = Mloc, block = 256
// ------------------------------------------------------------------------------------------------
// ---- the samples of one (particle, estimator) over several blocks ------------------------------------------------------------------
// Early in a run the softmax over the S samples is one-hot in float32 (the log-probabilities differ by hundreds) and one gradient per
// particle is evaluated; once the particles have sharpened, many samples keep a non-zero weight (config 3 at step 1 500: 13 on average,
// 53 for the worst particle; config 5 at step 300: 15 / 123) and ONE block per particle walked them one after the other -- the worst
// particle set the time of the launch (k_lin_grad 62 -> 1 070 us, k_nn_grad 3.2 -> 53 ms).  Now GRAD_NS blocks share a particle: the
// samples with non-zero weight are dealt round-robin in sample order (ordinal q -> block q mod GRAD_NS), every block accumulates its
// share, and the block that finishes LAST adds the partial sums in block order (a counter per (particle, estimator); partial sums stored
// at agent scope, as the kernel-matrix tiles do) and runs the epilogue.  The grouping is a function of the weights only -- not of the shard
// -- so the result does not depend on the rank count; with one non-zero weight block 0 does everything as before and nothing is exchanged.
// softmax statistics of a particle's S log-probabilities in double (as the oracle: dibs.py:376-382 through logsumexp): maximum, sum of
// exponentials, sum of the log-probabilities, and the number of samples whose weight is at least GRAD_W_MIN.  `red`: 3 NW doubles of LDS.
template <int NW = 4>
__device__ __forceinline__ void grad_softmax_stats(const float* __restrict__ lp, int S, double* red, double& mx, double& den, double& sm, int& nnz) {
  constexpr int NTHR = 64 * NW;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  mx = -INFINITY;
  for (int s = tid; s < S; s += NTHR) mx = (double)lp[s] > mx ? (double)lp[s] : mx;
  mx = wave_max_d(mx);
  if (lane == 0) red[wave] = mx;
  __syncthreads();
  mx = red[0];
  for (int w = 1; w < NW; ++w) mx = red[w] > mx ? red[w] : mx;
  den = 0.0;
  sm = 0.0;
  for (int s = tid; s < S; s += NTHR) {
    den += exp((double)lp[s] - mx);
    sm += (double)lp[s];
  }
  den = wave_sum_d(den);
  sm = wave_sum_d(sm);
  __syncthreads();
  if (lane == 0) {
    red[wave] = den;
    red[NW + wave] = sm;
  }
  __syncthreads();
  den = 0.0;
  sm = 0.0;
  for (int w = 0; w < NW; ++w) {
    den += red[w];
    sm += red[NW + w];
  }
  double cnt = 0.0;
  for (int s = tid; s < S; s += NTHR) cnt += ((float)(exp((double)lp[s] - mx) / den) >= GRAD_W_MIN) ? 1.0 : 0.0;
  cnt = wave_sum_d(cnt);
  if (lane == 0) red[2 * NW + wave] = cnt;
  __syncthreads();
  cnt = 0.0;
  for (int w = 0; w < NW; ++w) cnt += red[2 * NW + w];
  nnz = (int)cnt;
}
// plan of a persistent gradient kernel: one wave per job computes the statistics and appends the job's shares that have work to the item
// list (in arrival order -- nothing depends on the order: an item's partial row and its place in the sum are fixed by (job, share)).
// grid = jobs, block = 64; the counters of this launch were zeroed by the previous launch's plan kernel (joint_grad_plan).
#ifdef DIBS_TU_NN
__global__ void k_grad_plan(const float* __restrict__ logprobs, int S, int ns, GradPlan gp) {
  __shared__ double red[4];
  const int m = blockIdx.x;
  double mx, den, sm;
  int nnz;
  grad_softmax_stats<1>(logprobs + (size_t)m * S, S, red, mx, den, sm, nnz);
  if (m == 0 && threadIdx.x < 2) gp.ctr_next[threadIdx.x] = 0u;
  if (threadIdx.x == 0) {
    gp.stats[(size_t)m * 4 + 0] = mx;
    gp.stats[(size_t)m * 4 + 1] = den;
    gp.stats[(size_t)m * 4 + 2] = sm;
    gp.stats[(size_t)m * 4 + 3] = (double)nnz;
    const int nact = nnz < ns ? (nnz > 0 ? nnz : 1) : ns;
    const unsigned int base = atomicAdd(gp.ctr, (unsigned int)nact);
    for (int y = 0; y < nact; ++y) gp.items[base + y] = (unsigned int)m * 64u + (unsigned int)y;
  }
}
#endif

// the partial sums of this block are complete (stored with grad_part_store): count this block; true for the LAST of `nact` blocks, which
// then reads all of them with grad_part_load.  `flag`: one int of LDS.
__device__ __forceinline__ void grad_part_store(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float grad_part_load(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// element `idx` of the partial rows 0 .. nact-1 (row stride `stride`), added in row order; all loads are issued before the first addition
// (a rolled loop over the rows has ONE load in flight per thread: 4 480 dependent trips past the L2 for a k_nn_grad row -- the last block
// of a particle took milliseconds)
template <int NSMAX, bool ATOMIC>
__device__ __forceinline__ float grad_part_sum(const float* base, size_t stride, size_t idx, int nact) {
  if (nact == 1) return ATOMIC ? grad_part_load(base + idx) : base[idx];  // (block-uniform; 0.f + v == v)
  float v[NSMAX];
#pragma unroll
  for (int b = 0; b < NSMAX; ++b) {
    const float* p = base + (size_t)(b < nact ? b : 0) * stride + idx;
    v[b] = ATOMIC ? grad_part_load(p) : *p;
  }
  float t = 0.f;
#pragma unroll
  for (int b = 0; b < NSMAX; ++b) t += b < nact ? v[b] : 0.f;
  return t;
}
__device__ __forceinline__ bool grad_last_block(unsigned int* ctr, int nact, int* flag) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (every wave: its own partial stores are complete before the block counts itself)
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned int done = atomicAdd(ctr, 1u) + 1u;
    if (done == (unsigned int)nact) atomicExch(ctr, 0u);
    *flag = done == (unsigned int)nact;
  }
  __syncthreads();
  return *flag != 0;
}

// one estimator's inputs / outputs; the theta and the Z estimator of a step run as blockIdx.y = 0 / 1 of ONE launch (each has
// only Mloc blocks -- half the CUs -- and they are independent once both sets of log-probs exist)
struct LinGradJob {
  const float* logprobs;
  float* out;
  size_t out_stride;
  float* theta_copy;
  float* baseline_out;
  Key2 carry;
  int mode;
};
template <int NT>
__global__ __launch_bounds__(256) void k_lin_grad(const float* __restrict__ x, const int32_t* __restrict__ mask,
                                                  const float* __restrict__ theta, const float* __restrict__ scores,
                                                  const uint32_t* __restrict__ thr, LinGradJob job0, LinGradJob job1,
                                                  const float* __restrict__ baseline, int m0, int M_global, int d, int N, int S,
                                                  float alpha, float tau, int layout, int tiny, float obs_noise, float mu, float sig,
                                                  double sf_baseline, int any_mask, GradSplit gs) {
  const LinGradJob job = blockIdx.y ? job1 : job0;  // (grid = (Mloc, 2 estimators, shares); particle = (x + z) mod Mloc: see k_nn_grad)
  const float* __restrict__ logprobs = job.logprobs;
  float* __restrict__ out = job.out;
  const size_t out_stride = job.out_stride;
  float* __restrict__ theta_copy = job.theta_copy;
  float* __restrict__ baseline_out = job.baseline_out;
  const Key2 carry = job.carry;
  const int mode = job.mode;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const LinGeom g = lin_geom(d, N, NT);
  float* X = smem;
  float* WG = X + (size_t)g.np * g.ldx;
  float* RS = WG + (size_t)g.kp * g.ldw;  // residuals [np][ldw]
  double* red = reinterpret_cast<double*>(smem + ((((size_t)g.np * g.ldx + (size_t)g.kp * g.ldw + (size_t)g.np * g.ldw) + 3) & ~(size_t)3));
  const int m = (int)((blockIdx.x + blockIdx.z) % gridDim.x), tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const size_t dd = (size_t)d * d;
  const float* __restrict__ TH = theta + (size_t)m * dd;
  const Key2 key = lin_mode_key(mode, carry, M_global, m0 + m, layout);
  const uint64_t nbits = (uint64_t)S * dd;
  const float* lp = logprobs + (size_t)m * S;
  // softmax statistics (double), number of samples with a non-zero weight; this block's share of them: ordinals q = bz, bz + NS, ...
  __shared__ float wch[GRAD_WCH];
  __shared__ int last_flag;
  double mx, den, sm;
  int nnz;
  grad_softmax_stats(lp, S, red, mx, den, sm, nnz);
  const int NS = gridDim.z, bz = blockIdx.z, nact = nnz < NS ? (nnz > 0 ? nnz : 1) : NS;
  if (bz >= nact) return;  // (block-uniform: no share -- before anything is staged)
  lin_load_common<NT>(X, x, g, tid);
  for (int e = tid; e < g.np * g.ldw; e += 256) RS[e] = 0.f;

  // accumulators in the MFMA C layout: element (i = ti*16 + (lane>>4)*4 + r, j = tj*16 + (lane&15)), ti = wave + 4*u
  constexpr int NU = (NT + 3) / 4;
  f32x4 acc[NU][NT];
#pragma unroll
  for (int u = 0; u < NU; ++u)
#pragma unroll
    for (int tj = 0; tj < NT; ++tj) acc[u][tj] = f32x4{0.f, 0.f, 0.f, 0.f};
  const float inv_on = 1.0f / obs_noise;
  const float* sc_m = scores + (size_t)m * dd;
  const uint32_t* thr_m = thr + (size_t)m * dd;

  int q = 0;  // ordinal of the next sample with a non-zero weight
  for (int s0 = 0; s0 < S; s0 += GRAD_WCH) {
    __syncthreads();
    if (s0 + tid < S) wch[tid] = (float)(exp((double)lp[s0 + tid] - mx) / den);
    __syncthreads();
  for (int s = s0; s < S && s < s0 + GRAD_WCH; ++s) {
    const float w = wch[s - s0];
    if (w < GRAD_W_MIN) continue;  // block-uniform
    if ((q++ % NS) != bz) continue;  // (another block's sample)
    if (mode == LIN_MODE_Z_SCORE) {
#pragma unroll
      for (int u = 0; u < NU; ++u)
#pragma unroll
        for (int tj = 0; tj < NT; ++tj)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int i = (wave + 4 * u) * 16 + (lane >> 4) * 4 + r, j = tj * 16 + (lane & 15);
            if (i < d && j < d) acc[u][tj][r] += w * lin_sample_g(mode, key, nbits, dd, s, i, j, d, thr_m, sc_m, alpha, tau, layout, tiny);
          }
      continue;
    }
    __syncthreads();
    lin_build_wg<NT>(WG, TH, mode, key, nbits, s, thr_m, sc_m, alpha, tau, layout, tiny, mu, sig, g, tid);
    __syncthreads();
    lin_pred_tiles<NT>(X, WG, g, lane, wave, [&](int n, int j, float pred) {
      const bool mk = any_mask && mask[(size_t)n * d + j];
      RS[n * g.ldw + j] = mk ? 0.f : (X[n * g.ldx + j] - pred) * inv_on;
    });
    __syncthreads();
    // xtr = X^T * RS (K = np), then fold into the accumulators
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      const int ti = wave + 4 * u;
      if (ti >= NT) continue;  // (not `break`: keeps the trip count constant so the loop unrolls)
      f32x4 t[NT];
#pragma unroll
      for (int tj = 0; tj < NT; ++tj) t[tj] = f32x4{0.f, 0.f, 0.f, 0.f};
      const int ap = (lane >> 4) * g.ldx + ti * 16 + (lane & 15);
      const int bq = (lane >> 4) * g.ldw + (lane & 15);
      for (int k0 = 0; k0 < g.np; k0 += 4) {
        const float a = X[ap + k0 * g.ldx];
#pragma unroll
        for (int tj = 0; tj < NT; ++tj) t[tj] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, RS[bq + k0 * g.ldw + tj * 16], t[tj], 0, 0, 0);
      }
#pragma unroll
      for (int tj = 0; tj < NT; ++tj)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int i = ti * 16 + (lane >> 4) * 4 + r, j = tj * 16 + (lane & 15);
          if (i < d && j < d) {
            float xtr = t[tj][r];
            asm volatile("" : "+v"(xtr));
            const float th = TH[i * d + j];
            if (mode == LIN_MODE_THETA) {
              const float gv = lin_sample_g(mode, key, nbits, dd, s, i, j, d, thr_m, sc_m, alpha, tau, layout, tiny);
              acc[u][tj][r] += w * gv * (-(th - mu) / (sig * sig) + xtr);
            } else if (i != j) {
              const float gv = lin_sample_g(mode, key, nbits, dd, s, i, j, d, thr_m, sc_m, alpha, tau, layout, tiny);
              acc[u][tj][r] += w * (lin_logn(th, mu, sig) + th * xtr) * tau * alpha * gv * (1.0f - gv);
            }
          }
        }
    }
  }
  }
  if (nact > 1) {
    // partial sums in thread layout ([value][thread]: coalesced, no index arithmetic); the last block adds them in block order
    float* const base = gs.part + ((size_t)(m * 2 + (int)blockIdx.y) * NS) * gs.stride;
    float* const mine = base + (size_t)bz * gs.stride + tid;
#pragma unroll
    for (int u = 0; u < NU; ++u)
#pragma unroll
      for (int tj = 0; tj < NT; ++tj)
#pragma unroll
        for (int r = 0; r < 4; ++r) grad_part_store(mine + (size_t)((u * NT + tj) * 4 + r) * 256, acc[u][tj][r]);
    if (!grad_last_block(gs.ctr + (m * 2 + (int)blockIdx.y), nact, &last_flag)) return;
#pragma unroll
    for (int u = 0; u < NU; ++u)
#pragma unroll
      for (int tj = 0; tj < NT; ++tj)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[u][tj][r] = grad_part_sum<GRAD_NS, true>(base, gs.stride, (size_t)((u * NT + tj) * 4 + r) * 256 + tid, nact);
  }
  // epilogue
  const float bold = baseline ? baseline[m] : 0.f;
  const float scale = (mode == LIN_MODE_Z_SCORE && sf_baseline > 0.0) ? (float)exp(-(double)bold) : 1.0f;
  float* om = out + (size_t)m * out_stride;
#pragma unroll
  for (int u = 0; u < NU; ++u)
#pragma unroll
    for (int tj = 0; tj < NT; ++tj)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = (wave + 4 * u) * 16 + (lane >> 4) * 4 + r, j = tj * 16 + (lane & 15);
        if (i < d && j < d) {
          float v = acc[u][tj][r];
          if (mode == LIN_MODE_Z_SCORE) {
            const float p = (float)sigmoid_d((double)__fmul_rn(alpha, sc_m[i * d + j]));
            v = i == j ? 0.f : scale * alpha * (v - p);
          }
          om[i * d + j] = v;
          if (theta_copy) theta_copy[(size_t)m * out_stride + i * d + j] = TH[i * d + j];
        }
      }
  if (mode != LIN_MODE_THETA && baseline_out && tid == 0)
    baseline_out[m] = (mode == LIN_MODE_Z_SCORE) ? (float)(sf_baseline * (sm / S) + (1.0 - sf_baseline) * (double)bold) : bold;
}

// ---- host side (defined in tu_lin.hip) --------------------------------------------------------------
int joint_alloc(JointWork* w, int Mloc, int d, int N, int S);
void joint_free(JointWork* w);
int joint_set_data(JointWork* w, const float* x, const int32_t* mask, int N, int d);
// true: x fits the LDS-resident MFMA kernels; false: the Gram-matrix path of kernels_lin_gram.h runs (joint_lin_set_gram builds C)
bool joint_lin_fast_path(int d, int N, bool force_gram);
int joint_lin_set_gram(JointWork* w, const float* x, const int32_t* mask, int N, int d);
void joint_lin_all_logprobs(JointWork* w, const JointLaunch& jl, Key2 carry_theta, Key2 carry_z);
void joint_lin_all_grads(JointWork* w, const JointLaunch& jl, Key2 carry_theta, Key2 carry_z);
// log p(theta_i, D | g_i) of n given (graph, parameter) pairs (held-out scoring; dibs_score_graphs)
void joint_lin_score_given(const JointWork& jw, const float* theta, const int32_t* g, float* out, int n, int d, int N, float obs_noise,
                           float mean_edge, float sig_edge, hipStream_t stream);

#ifdef DIBS_TU_LIN
#include <stdlib.h>
#include <vector>
#include "kernels_lin_gram.h"

bool joint_lin_fast_path(int d, int N, bool force_gram) {
  // (the MFMA kernels are instantiated for up to 7 tiles of 16 variables; force_gram: DibsTuning::lin_gram)
  // (2 KiB below the capacity: the gradient kernel has a little static LDS of its own)
  return !force_gram && d <= 112 && lin_lds_bytes(d, N, (d + 15) / 16, true) <= (size_t)160 * 1024 - 2048;
}

int joint_lin_set_gram(JointWork* w, const float* x, const int32_t* mask, int N, int d) {
  bool any = false;
  if (mask)
    for (size_t i = 0; i < (size_t)N * d; ++i) any |= mask[i] != 0;
  const int ng = any ? d : 1;
  std::vector<double> C((size_t)ng * d * d, 0.0), cnt(d, 0.0);
  for (int jm = 0; jm < ng; ++jm)
    for (int n = 0; n < N; ++n) {
      if (any && mask[(size_t)n * d + jm]) continue;
      const float* xr = x + (size_t)n * d;
      double* Cj = C.data() + (size_t)jm * d * d;
      for (int a = 0; a < d; ++a) {
        const double xa = xr[a];
        for (int b = 0; b < d; ++b) Cj[(size_t)a * d + b] += xa * (double)xr[b];
      }
    }
  for (int j = 0; j < d; ++j)
    for (int n = 0; n < N; ++n) cnt[j] += (any && mask[(size_t)n * d + j]) ? 0.0 : 1.0;
  if (w->gram) hipFree(w->gram);
  if (w->ncnt) hipFree(w->ncnt);
  w->gram = nullptr;
  w->ncnt = nullptr;
  w->n_gram = 0;
  if (hipMalloc((void**)&w->gram, C.size() * 8) != hipSuccess) return 1;
  if (hipMalloc((void**)&w->ncnt, cnt.size() * 8) != hipSuccess) return 1;
  if (hipMemcpy(w->gram, C.data(), C.size() * 8, hipMemcpyHostToDevice) != hipSuccess) return 1;
  if (hipMemcpy(w->ncnt, cnt.data(), cnt.size() * 8, hipMemcpyHostToDevice) != hipSuccess) return 1;
  w->n_gram = ng;
  return 0;
}

static size_t ling_lds(int d, int n_gram, bool grad) {
  const size_t dd = (size_t)d * d;
  return (n_gram == 1 ? dd * 8 : 0) + (((grad ? 2 : 1) * dd * 4 + 15) & ~(size_t)15) + 128;
}
// the operands of a block (masked weights; for the gradient kernel the graph as well) beyond the LDS capacity: global scratch
// (n_vars > 198 for the log-probabilities, > 141 for the gradients)
static bool ling_ops_global(int d, bool grad) { return ling_lds(d, -1, grad) > (size_t)160 * 1024 - 1024; }
// the n_gram argument of the Gram kernels: a single matrix stays in LDS only while it fits beside the operands (d <= 101 for the gradient
// kernel); beyond that it is read through the caches (-1).  (Found by tests/tools/gpu_fuzz.py: d = 112 with 500 observations failed to launch.)
static int ling_ngram_arg(int d, int n_gram, bool grad) {
  return (n_gram == 1 && ling_lds(d, 1, grad) > (size_t)160 * 1024) ? -1 : n_gram;
}
int joint_alloc(JointWork* w, int Mloc, int d, int N, int S) {
  (void)N;
  w->x = nullptr;
  w->mask = nullptr;
  w->ln_tab = nullptr;
  w->w1t = nullptr;
  w->w1t_floats = 0;
  w->any_mask = 0;
  w->nng_scratch = nullptr;
  w->nng_scratch_floats = 0;
  w->gs_scratch = nullptr;
  w->gs_scratch_floats = 0;
  w->gpart = nullptr;
  w->gpart_floats = 0;
  w->gctr = nullptr;
  w->gctr_n = 0;
  w->gplan = GradPlan{nullptr, nullptr, nullptr, nullptr};
  w->gplan_jobs = w->gplan_items = 0;
  w->gplan_gen = 0;
  w->nhf_w1s = w->nhf_w1p = nullptr;
  w->nhf_ew = nullptr;
  w->nhf_pairs = 0;
  w->nhx_w1s = w->nhx_w1p = nullptr;
  w->nhx_quads = 0;
  w->nhf_valid = w->nhx_valid = false;
  w->gram = nullptr;
  w->ncnt = nullptr;
  w->n_gram = 0;
  if (hipMalloc((void**)&w->wsm, (size_t)Mloc * S * 4) != hipSuccess) return 1;
  if (hipMalloc((void**)&w->ln_tab, (size_t)Mloc * d * d * 4) != hipSuccess) return 1;
  return 0;
}
void joint_free(JointWork* w) {
  if (w->x) hipFree(w->x);
  if (w->mask) hipFree(w->mask);
  if (w->wsm) hipFree(w->wsm);
  if (w->ln_tab) hipFree(w->ln_tab);
  if (w->w1t) hipFree(w->w1t);
  w->w1t = nullptr;
  w->w1t_floats = 0;
  if (w->nng_scratch) hipFree(w->nng_scratch);
  if (w->gs_scratch) hipFree(w->gs_scratch);
  w->gs_scratch = nullptr;
  w->gs_scratch_floats = 0;
  if (w->gpart) hipFree(w->gpart);
  if (w->gctr) hipFree(w->gctr);
  w->gpart = nullptr;
  w->gctr = nullptr;
  w->gpart_floats = w->gctr_n = 0;
  if (w->gplan.stats) hipFree(w->gplan.stats);
  if (w->gplan.items) hipFree(w->gplan.items);
  if (w->gplan.ctr) hipFree(w->gplan.ctr);
  w->gplan = GradPlan{nullptr, nullptr, nullptr, nullptr};
  w->gplan_jobs = w->gplan_items = 0;
  w->gplan_gen = 0;
  if (w->nhf_w1s) hipFree(w->nhf_w1s);
  if (w->nhf_w1p) hipFree(w->nhf_w1p);
  if (w->nhf_ew) hipFree(w->nhf_ew);
  if (w->nhx_w1s) hipFree(w->nhx_w1s);
  if (w->nhx_w1p) hipFree(w->nhx_w1p);
  w->nhx_w1s = w->nhx_w1p = nullptr;
  w->nhx_quads = 0;
  w->nhf_w1s = w->nhf_w1p = nullptr;
  w->nhf_ew = nullptr;
  w->nhf_pairs = 0;
  w->nhf_valid = w->nhx_valid = false;
  if (w->gram) hipFree(w->gram);
  if (w->ncnt) hipFree(w->ncnt);
  w->gram = nullptr;
  w->ncnt = nullptr;
  w->n_gram = 0;
  w->nng_scratch = nullptr;
  w->nng_scratch_floats = 0;
  w->ln_tab = nullptr;
  w->x = nullptr;
  w->mask = nullptr;
  w->wsm = nullptr;
}
int joint_set_data(JointWork* w, const float* x, const int32_t* mask, int N, int d) {
  const size_t n = (size_t)N * d;
  if (w->x) hipFree(w->x);
  if (w->mask) hipFree(w->mask);
  if (hipMalloc((void**)&w->x, n * 4) != hipSuccess) return 1;
  if (hipMalloc((void**)&w->mask, n * 4) != hipSuccess) return 1;
  if (hipMemcpy(w->x, x, n * 4, hipMemcpyHostToDevice) != hipSuccess) return 1;
  w->any_mask = 0;
  if (mask) {
    for (size_t i = 0; i < n; ++i) w->any_mask |= mask[i] != 0;
    if (hipMemcpy(w->mask, mask, n * 4, hipMemcpyHostToDevice) != hipSuccess) return 1;
  } else if (hipMemset(w->mask, 0, n * 4) != hipSuccess) {
    return 1;
  }
  return 0;
}

template <int NT>
static void joint_lin_logprobs(JointWork* w, const JointLaunch& jl, Key2 carry, int mode) {
  const int spb = 4;
  const size_t lds1 = lin_lds_bytes(jl.d, jl.N, NT, false);
  if (lds1 > 48 * 1024) hipFuncSetAttribute((const void*)k_lin_logprobs<NT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1);
  float* lp = mode == LIN_MODE_THETA ? jl.logprobs_th : jl.logprobs_z;
  const bool paired = jl.layout == 0 && (jl.S & 1) == 0 && (uint64_t)jl.S * jl.d * jl.d < 0xFFFFFFFFull && jl.N <= 128;
  if (paired) {
    const bool use_bf = NT <= 4 && jl.d > 32 && !jl.lin_f32;  // (lin_f32: tuning.h, the f32-MFMA kernel at 33 <= d <= 64 for A/B runs)
    // pairs per block: the block's prologue (x fragments, operand factors, zeroed images) is ~a third of a pair's work; 8 pairs when that
    // still leaves two full rounds of blocks (config 3: 1 914 -> 1 964 steps/s; 16 pairs: 1 856)
    const int ppb = (use_bf && (jl.S / 2 / 8) * jl.Mloc >= 1024) ? 8 : 4;
    const size_t ldsp = lin_lds_bytes_pair(jl.d, NT);
    const dim3 grid((jl.S / 2 + ppb - 1) / ppb, jl.Mloc);
    const int epq = (jl.d * jl.d + 255) / 256;
    if (use_bf) {
      const int ldsb = 2 * AHF_IMG_BYTES + 256;
      const int epq8 = (jl.d * jl.d + 511) / 512;
#define LIN_HF_LAUNCH(EPQ_, FOUR_, NW_)                                                                                                      \
      {                                                                                                                                      \
        hipFuncSetAttribute((const void*)k_lin_logprobs_hf<EPQ_, FOUR_, NW_>, hipFuncAttributeMaxDynamicSharedMemorySize, ldsb);             \
        hipLaunchKernelGGL((k_lin_logprobs_hf<EPQ_, FOUR_, NW_>), grid, dim3(64 * NW_), ldsb, jl.stream, w->x, w->mask, jl.theta, jl.scores, \
                           jl.thr, lp, carry, mode, jl.m0, jl.M, jl.d, jl.N, jl.S, ppb, jl.alpha, jl.tau, jl.layout, jl.tiny, jl.obs_noise,  \
                           jl.mean_edge, jl.sig_edge, w->any_mask);                                                                          \
      }
      if (jl.d <= 48) LIN_HF_LAUNCH(5, false, 8)
      else if (epq8 <= 5) LIN_HF_LAUNCH(5, true, 8)
      else LIN_HF_LAUNCH(8, true, 8)
#undef LIN_HF_LAUNCH
      return;
    }
#define LIN_PAIR_LAUNCH(EPQ_)                                                                                                      \
    {                                                                                                                              \
      if (ldsp > 48 * 1024) hipFuncSetAttribute((const void*)k_lin_logprobs_pair<NT, EPQ_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsp); \
      hipLaunchKernelGGL((k_lin_logprobs_pair<NT, EPQ_>), grid, dim3(256), ldsp, jl.stream, w->x, w->mask, jl.theta, jl.scores, jl.thr, lp,   \
                         carry, mode, jl.m0, jl.M, jl.d, jl.N, jl.S, ppb, jl.alpha, jl.tau, jl.layout, jl.tiny, jl.obs_noise,         \
                         jl.mean_edge, jl.sig_edge, w->any_mask);                                                                  \
    }
    if (NT <= 4 && epq <= 4) LIN_PAIR_LAUNCH(4)
    else if (NT <= 4 && epq <= 10) LIN_PAIR_LAUNCH(10)
    else if (NT <= 4) LIN_PAIR_LAUNCH(16)
    else LIN_PAIR_LAUNCH(0)
#undef LIN_PAIR_LAUNCH
  } else {
    hipLaunchKernelGGL(k_lin_logprobs<NT>, dim3((jl.S + spb - 1) / spb, jl.Mloc), dim3(256), lds1, jl.stream, w->x, w->mask, jl.theta,
                       jl.scores, jl.thr, lp, carry, mode, jl.m0, jl.M, jl.d, jl.N, jl.S, spb, jl.alpha, jl.tau, jl.layout, jl.tiny,
                       jl.obs_noise, jl.mean_edge, jl.sig_edge, w->any_mask);
  }
}

template <int NT>
static void joint_lin_grads(JointWork* w, const JointLaunch& jl, Key2 carry_theta, Key2 carry_z) {
  const size_t lds2 = lin_lds_bytes(jl.d, jl.N, NT, true);
  if (lds2 > 48 * 1024) hipFuncSetAttribute((const void*)k_lin_grad<NT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2);
  const LinGradJob jt{jl.logprobs_th, jl.pack + (size_t)jl.m0 * jl.pack_stride + jl.gtheta_off, jl.pack_stride,
                      jl.copy_theta ? jl.pack + (size_t)jl.m0 * jl.pack_stride + jl.theta_off : nullptr, nullptr, carry_theta, LIN_MODE_THETA};
  const LinGradJob jz{jl.logprobs_z, jl.w_lik, (size_t)jl.d * jl.d, nullptr, jl.baseline_out, carry_z,
                      jl.est_z == 0 ? LIN_MODE_Z_SCORE : LIN_MODE_Z_REPARAM};
  GradSplit gs;
  if (!joint_grad_split(w, (size_t)jl.Mloc * 2, (size_t)((NT + 3) / 4) * NT * 4 * 256, &gs)) return;  // (the step's launch check reports the failed hipMalloc)
  hipLaunchKernelGGL(k_lin_grad<NT>, dim3(jl.Mloc, 2, GRAD_NS), dim3(256), lds2, jl.stream, w->x, w->mask, jl.theta, jl.scores, jl.thr, jt, jz,
                     jl.baseline, jl.m0, jl.M, jl.d, jl.N, jl.S, jl.alpha, jl.tau, jl.layout, jl.tiny, jl.obs_noise, jl.mean_edge,
                     jl.sig_edge, jl.sf_baseline, w->any_mask, gs);
}

#define LIN_NT_SWITCH(CALL_)                 \
  switch ((jl.d + 15) / 16) {                \
    case 1: CALL_(1); break;                 \
    case 2: CALL_(2); break;                 \
    case 3: CALL_(3); break;                 \
    case 4: CALL_(4); break;                 \
    case 5: CALL_(5); break;                 \
    case 6: CALL_(6); break;                 \
    default: CALL_(7); break;                \
  }
// log p(theta, D | G_s) for the samples of the theta estimator and of the Z estimator (two launches)
void joint_lin_all_logprobs(JointWork* w, const JointLaunch& jl, Key2 carry_theta, Key2 carry_z) {
  const int mz = jl.est_z == 0 ? LIN_MODE_Z_SCORE : LIN_MODE_Z_REPARAM;
  if (w->n_gram) {  // Gram-matrix path (x does not fit LDS)
    const bool glob = ling_ops_global(jl.d, false);
    const int ng = glob ? (w->n_gram == 1 ? -1 : w->n_gram) : ling_ngram_arg(jl.d, w->n_gram, false);
    const size_t lds = glob ? 256 : ling_lds(jl.d, ng, false);
    // (global operands: a bounded number of blocks per particle loop over the samples, each with its own d x d scratch)
    const int gx = glob ? (jl.S < 1024 / jl.Mloc ? jl.S : (1024 / jl.Mloc > 1 ? 1024 / jl.Mloc : 1)) : jl.S;
    float* gs = glob ? joint_gs_scratch(w, (size_t)gx * jl.Mloc * jl.d * jl.d) : nullptr;
    if (glob && !gs) return;  // (allocation failure: the launch error check of the step reports hipErrorOutOfMemory)
    if (lds > 48 * 1024) hipFuncSetAttribute((const void*)k_ling_logprobs, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(k_ling_logprobs, dim3(gx, jl.Mloc), dim3(256), lds, jl.stream, w->gram, w->ncnt, ng, jl.theta, jl.scores, jl.thr,
                       jl.logprobs_th, carry_theta, (int)LIN_MODE_THETA, jl.m0, jl.M, jl.d, jl.S, jl.alpha, jl.tau, jl.layout, jl.tiny, jl.obs_noise,
                       jl.mean_edge, jl.sig_edge, gs);
    hipLaunchKernelGGL(k_ling_logprobs, dim3(gx, jl.Mloc), dim3(256), lds, jl.stream, w->gram, w->ncnt, ng, jl.theta, jl.scores, jl.thr,
                       jl.logprobs_z, carry_z, mz, jl.m0, jl.M, jl.d, jl.S, jl.alpha, jl.tau, jl.layout, jl.tiny, jl.obs_noise, jl.mean_edge,
                       jl.sig_edge, gs);
    return;
  }
#define LIN_CALL(NT_) { joint_lin_logprobs<NT_>(w, jl, carry_theta, LIN_MODE_THETA); joint_lin_logprobs<NT_>(w, jl, carry_z, mz); }
  LIN_NT_SWITCH(LIN_CALL)
#undef LIN_CALL
}
// both softmax-weighted gradients in one launch
void joint_lin_all_grads(JointWork* w, const JointLaunch& jl, Key2 carry_theta, Key2 carry_z) {
  if (w->n_gram) {
    const bool glob = ling_ops_global(jl.d, true);
    const int ng = glob ? (w->n_gram == 1 ? -1 : w->n_gram) : ling_ngram_arg(jl.d, w->n_gram, true);
    const size_t lds = glob ? 256 : ling_lds(jl.d, ng, true);
    float* gs = glob ? joint_gs_scratch(w, (size_t)GRAD_NS * 2 * jl.Mloc * 2 * jl.d * jl.d) : nullptr;
    if (glob && !gs) return;
    GradSplit gsp;
    if (!joint_grad_split(w, (size_t)jl.Mloc * 2, (size_t)jl.d * jl.d, &gsp)) return;
    if (lds > 48 * 1024) hipFuncSetAttribute((const void*)k_ling_grad, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const LinGradJob jt{jl.logprobs_th, jl.pack + (size_t)jl.m0 * jl.pack_stride + jl.gtheta_off, jl.pack_stride,
                        jl.copy_theta ? jl.pack + (size_t)jl.m0 * jl.pack_stride + jl.theta_off : nullptr, nullptr, carry_theta, LIN_MODE_THETA};
    const LinGradJob jz{jl.logprobs_z, jl.w_lik, (size_t)jl.d * jl.d, nullptr, jl.baseline_out, carry_z,
                        jl.est_z == 0 ? LIN_MODE_Z_SCORE : LIN_MODE_Z_REPARAM};
    hipLaunchKernelGGL(k_ling_grad, dim3(jl.Mloc, 2, GRAD_NS), dim3(256), lds, jl.stream, w->gram, ng, jl.theta, jl.scores, jl.thr, jt, jz, jl.baseline,
                       jl.m0, jl.M, jl.d, jl.S, jl.alpha, jl.tau, jl.layout, jl.tiny, jl.obs_noise, jl.mean_edge, jl.sig_edge, jl.sf_baseline, gs, gsp);
    return;
  }
#define LIN_CALL(NT_) joint_lin_grads<NT_>(w, jl, carry_theta, carry_z)
  LIN_NT_SWITCH(LIN_CALL)
#undef LIN_CALL
}
#undef LIN_NT_SWITCH

template <int NT>
static void launch_lin_given(const JointWork& jw, const float* theta, const int32_t* g, float* out, int n, int d, int N, float obs_noise,
                             float mean_edge, float sig_edge, hipStream_t stream) {
  const size_t lds = lin_lds_bytes(d, N, NT, false);
  if (lds > 48 * 1024) hipFuncSetAttribute((const void*)k_lin_logprobs<NT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(k_lin_logprobs<NT>, dim3(1, n), dim3(256), lds, stream, jw.x, jw.mask, theta, (const float*)nullptr,
                     reinterpret_cast<const uint32_t*>(g), out, Key2{0, 0}, (int)LIN_MODE_GIVEN, 0, n, d, N, 1, 1, 0.f, 1.f, 0, 0,
                     obs_noise, mean_edge, sig_edge, jw.any_mask);
}
void joint_lin_score_given(const JointWork& jw, const float* theta, const int32_t* g, float* out, int n, int d, int N, float obs_noise,
                           float mean_edge, float sig_edge, hipStream_t stream) {
  if (jw.n_gram) {
    const bool glob = ling_ops_global(d, false);
    const int ng = glob ? (jw.n_gram == 1 ? -1 : jw.n_gram) : ling_ngram_arg(d, jw.n_gram, false);
    const size_t lds = glob ? 256 : ling_lds(d, ng, false);
    float* gs = glob ? joint_gs_scratch(const_cast<JointWork*>(&jw), (size_t)n * d * d) : nullptr;
    if (glob && !gs) return;
    if (lds > 48 * 1024) hipFuncSetAttribute((const void*)k_ling_logprobs, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(k_ling_logprobs, dim3(1, n), dim3(256), lds, stream, jw.gram, jw.ncnt, ng, theta, (const float*)nullptr,
                       reinterpret_cast<const uint32_t*>(g), out, Key2{0, 0}, (int)LIN_MODE_GIVEN, 0, n, d, 1, 0.f, 1.f, 0, 0, obs_noise, mean_edge,
                       sig_edge, gs);
    return;
  }
  switch ((d + 15) / 16) {
    case 1: launch_lin_given<1>(jw, theta, g, out, n, d, N, obs_noise, mean_edge, sig_edge, stream); break;
    case 2: launch_lin_given<2>(jw, theta, g, out, n, d, N, obs_noise, mean_edge, sig_edge, stream); break;
    case 3: launch_lin_given<3>(jw, theta, g, out, n, d, N, obs_noise, mean_edge, sig_edge, stream); break;
    case 4: launch_lin_given<4>(jw, theta, g, out, n, d, N, obs_noise, mean_edge, sig_edge, stream); break;
    case 5: launch_lin_given<5>(jw, theta, g, out, n, d, N, obs_noise, mean_edge, sig_edge, stream); break;
    case 6: launch_lin_given<6>(jw, theta, g, out, n, d, N, obs_noise, mean_edge, sig_edge, stream); break;
    default: launch_lin_given<7>(jw, theta, g, out, n, d, N, obs_noise, mean_edge, sig_edge, stream); break;
  }
}
#endif  // DIBS_TU_LIN
