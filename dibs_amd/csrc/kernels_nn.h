// JointDiBS + DenseNonlinearGaussian (one hidden layer of H units per node) on gfx950.
//   mean_nj = b2_j + sum_h W2_jh act( sum_a x_na g_aj W1_jah + b1_jh )
//   log p(theta, D | G) = sum_leaves logN(.; 0, sig_p) [first-layer weights masked by g^T] + sum_{n,j not intervened} logN(x_nj; mean_nj, sqrt(obs_noise))
// reference: dibs/models/nonlinearGaussian.py:35-81 (net), :248-326 (prior, likelihood); estimators as for LinearGaussian.
// For a fixed hidden unit h the first layer of ALL nodes is one GEMM  pre_h = x [N,d] * T_h [d,d],  T_h[a][j] = g[a][j] W1[j][a][h],
// so every h is a LinearGaussian-shaped pass on v_mfma_f32_16x16x4_f32; the second layer / activation live in the epilogue.
// theta row layout (= pytree leaf order of the reference): W1 [d][d][H] (node j, input a, unit h) | b1 [d][H] | W2 [d][H] | b2 [d]
// (without bias: W1 | W2).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "kernels_joint.h"

#ifndef DIBS_MAX_HIDDEN_LAYERS
#define DIBS_MAX_HIDDEN_LAYERS 8
#endif
struct NNParams {
  int H, act, bias;  // H = width of the first hidden layer (the tuned one-hidden-layer kernels below)
  float obs_noise, sig_param;
  int n_hidden, hidden[DIBS_MAX_HIDDEN_LAYERS];
};

__device__ __forceinline__ float nn_act(int a, float v) {
  switch (a) {
    case 0: return v > 0.f ? v : 0.f;
    case 1: return tanhf(v);
    case 2: return 1.0f / (1.0f + expf(-v));
    default: return v > 0.f ? v : 0.01f * v;
  }
}
__device__ __forceinline__ float nn_dact(int a, float v, float fv) {
  switch (a) {
    case 0: return v > 0.f ? 1.f : 0.f;
    case 1: return 1.f - fv * fv;
    case 2: return fv * (1.f - fv);
    default: return v > 0.f ? 1.f : 0.01f;
  }
}

struct NNOff {
  size_t w1, b1, w2, b2, P;
};
__host__ __device__ inline NNOff nn_offsets(int d, int H, int bias) {
  NNOff o;
  o.w1 = 0;
  o.b1 = (size_t)d * d * H;
  o.w2 = o.b1 + (bias ? (size_t)d * H : 0);
  o.b2 = o.w2 + (size_t)d * H;
  o.P = o.b2 + (bias ? (size_t)d : 0);
  return o;
}

// LDS (floats): X[np][ldx] | GS[d*d] | TR[max(kp, np)][ldw] | (grad kernel) CS[4][ldw] | red
// TR holds the forward operand T_h (rows < kp) and, after a barrier, the backward operand dpre_h (rows < np): the two are
// never live together, which is what lets d = 100 / N = 100 fit in 160 KiB.
__host__ __device__ inline int nn_tr_rows(const LinGeom g) { return g.kp > g.np ? g.kp : g.np; }
__host__ __device__ inline size_t nn_lds_bytes(int d, int N, int NT, bool grad) {
  const LinGeom g = lin_geom(d, N, NT);
  size_t f = (size_t)g.np * g.ldx + (size_t)d * d + (size_t)nn_tr_rows(g) * g.ldw;
  if (grad) f += (size_t)16 * g.ldw;  // two sets of per-wave column sums of k_nn_grad (up to 8 waves)
  return ((f * 4 + 15) & ~(size_t)15) + 64 * 8;
}
// k_nn_logprobs also keeps the small leaves b1 | W2 | b2 ([d][H] | [d][H] | [d]) behind `red`
__host__ __device__ inline size_t nn_lds_bytes_logprobs(int d, int N, int NT, int H) {
  return nn_lds_bytes(d, N, NT, false) + (((size_t)2 * d * H + d) * 4 + 15 & ~(size_t)15);
}

// pre = X * TW for the row tiles of this wave, results left in registers: acc[u][tj] (row tile ti = wave + 4 u)
template <int NT, int NU, int NW = 4>
__device__ __forceinline__ void nn_gemm_x_tw(const float* X, const float* TW, const LinGeom g, int lane, int wave, f32x4 (&acc)[NU][NT],
                                             int ldw_tw = 0) {
  const int nrt = g.np >> 4, ldw = ldw_tw ? ldw_tw : g.ldw;
#pragma unroll
  for (int u = 0; u < NU; ++u) {
    const int ti = wave + NW * u;
#pragma unroll
    for (int tj = 0; tj < NT; ++tj) acc[u][tj] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (ti >= nrt) continue;
    const int ap = (ti * 16 + (lane & 15)) * g.ldx + (lane >> 4);
    const int bq = (lane >> 4) * ldw + (lane & 15);
    for (int k0 = 0; k0 < g.kp; k0 += 4) {
      const float a = X[ap + k0];
#pragma unroll
      for (int tj = 0; tj < NT; ++tj) acc[u][tj] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, TW[bq + k0 * ldw + tj * 16], acc[u][tj], 0, 0, 0);
    }
#pragma unroll
    for (int tj = 0; tj < NT; ++tj)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float v = acc[u][tj][r];
        asm volatile("" : "+v"(v));  // keep MFMA results out of AGPR-sourced stores (see lds_matmul)
        acc[u][tj][r] = v;
      }
  }
}

// LN[a][j] = sum_h logN(W1[j][a][h]; 0, sig_p): the graph-dependent part of the parameter prior is sum_aj g[a][j] LN[a][j]
// (nonlinearGaussian.py:264-269).  It does not depend on the sample: one table per particle and step instead of H logN
// evaluations per element of every sampled graph.   grid = (ceil(d*d / 256), Mloc)
#ifdef DIBS_TU_NN
__global__ void k_nn_prior_table(const float* __restrict__ theta, size_t P, float* __restrict__ ln_tab, float* __restrict__ w1t, int d, int H,
                                 float sigp) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x, m = blockIdx.y;
  if (e >= d * d) return;
  const int a = e / d, j = e - a * d;
  const float* w = theta + (size_t)m * P + ((size_t)j * d + a) * H;
  float t = 0.f;
  for (int h = 0; h < H; ++h) {
    const float wv = w[h];
    t += lin_logn(wv, 0.f, sigp);
    if (w1t) w1t[((size_t)m * H + h) * d * d + e] = wv;  // W1T[h][a][j]: what k_nn_logprobs multiplies the sampled graph with, element by element
  }
  ln_tab[(size_t)m * d * d + e] = t;
}
#endif

// sample graph s into GS (row-major [a][j]); with a prior table returns this thread's share of sum g LN
__device__ __forceinline__ float nn_build_graph(float* GS, int mode, Key2 key, uint64_t nbits, int s, const uint32_t* thr_m,
                                                const float* sc_m, float alpha, float tau, int layout, int tiny, int d, int tid,
                                                const float* __restrict__ ln_m = nullptr, int nthr = 256) {
  const uint64_t dd = (uint64_t)d * d;
  float prior = 0.f;
  for (int e = tid; e < d * d; e += nthr) {
    const int a = e / d, j = e - a * d;
    const float gv = lin_sample_g(mode, key, nbits, dd, s, a, j, d, thr_m, sc_m, alpha, tau, layout, tiny);
    GS[e] = gv;
    if (ln_m) prior = fmaf(gv, ln_m[e], prior);
  }
  return prior;
}

// TW[a][j] = GS[a][j] * W1[j][a][h]  (zero padded); PRIOR: also returns this thread's share of sum g logN(W1[., ., h])
template <bool PRIOR>
__device__ __forceinline__ float nn_build_tw(float* TW, const float* GS, const float* __restrict__ th_m, int h, int H, float sigp,
                                             const LinGeom g, int tid, int nthr = 256) {
  float prior = 0.f;
  for (int e = tid; e < g.kp * g.ldw; e += nthr) {
    const int a = e / g.ldw, j = e - a * g.ldw;
    float v = 0.f;
    if (a < g.d && j < g.d) {
      const float gv = GS[a * g.d + j];
      const float w = th_m[((size_t)j * g.d + a) * H + h];
      v = gv * w;
      if (PRIOR) prior += gv * lin_logn(w, 0.f, sigp);
    }
    TW[e] = v;
  }
  return prior;
}

// the same operand from the re-laid-out weights W1T[h][a][j] (k_nn_prior_table): coalesced -- W1[j][a][h] itself is a 4 d H byte stride between
// neighbouring j, 64 sectors per wave-load; with thousands of sample gradients per launch (late in a run) those strided reads were the
// gradient kernel's time
__device__ __forceinline__ void nn_build_tw_t(float* TW, const float* GS, const float* __restrict__ w1t_h, const LinGeom g, int tid, int nthr) {
  for (int e = tid; e < g.kp * g.ldw; e += nthr) {
    const int a = e / g.ldw, j = e - a * g.ldw;
    TW[e] = (a < g.d && j < g.d) ? GS[a * g.d + j] * w1t_h[a * g.d + j] : 0.f;
  }
}

// per-wave column sums [NW][ldw] added in wave order
template <int NW>
__device__ __forceinline__ float nn_cs_sum(const float* CS, int ldw, int j) {
  float t = CS[j];
#pragma unroll
  for (int w = 1; w < NW; ++w) t += CS[w * ldw + j];
  return t;
}

// The same two steps for k_nn_logprobs with the per-particle tables (prior table LN, re-laid-out weights W1T): element index e = a d + j
// advances by the block size with a carry instead of a division, W1T is read coalesced (W1[j][a][h] itself is a 4 d H byte stride between
// neighbouring j), and only the d x d interior of the zero-padded operand is rewritten per hidden unit.
struct NNStep {
  int da, dj;  // block size = da * d + dj
};
// FAST: legacy PRNG layout with an even number of samples and S d d < 2^32 -- element e of sample s is word (s < S/2 ? 0 : 1) of the Threefry
// call on counters (c, c + S d d / 2), c = (s mod S/2) d d + e: 32-bit counters, key schedule hoisted (threefry2x32_uk), and for tau = 1
// sigmoid(eps + a) = u / (u + (1 - u) exp(-a)) on v_rcp_f32.  (The generic path -- 64-bit element index, layout switch, key schedule and an
// IEEE division per element -- is 320 vector instructions per element, 50 k of the kernel's 66 k wave-instructions per sample at config 5.)
template <bool FAST>
__device__ __forceinline__ float nn_build_graph_tab(float* GS, int mode, Key2 key, const TfKeys& tk, uint64_t nbits, int s, int S, const uint32_t* thr_m,
                                                    const float* sc_m, float alpha, float tau, int layout, int tiny, int d, int tid, int a0, int j0,
                                                    NNStep st, const float* __restrict__ ln_m, int nthr) {
  const uint64_t dd = (uint64_t)d * d;
  float prior = 0.f;
  int a = a0, j = j0;
  const int hS = S >> 1;
  const bool hi = s >= hS;
  const uint32_t cbase = (uint32_t)(hi ? s - hS : s) * (uint32_t)dd, half = (uint32_t)(nbits >> 1);
  const float ulo = tiny ? 1.17549435e-38f : 1.1920929e-07f;
  for (int e = tid; e < d * d; e += nthr) {
    float gv;
    if constexpr (FAST) {
      gv = 0.f;
      if (a != j) {
        uint32_t y0, y1;
        threefry2x32_uk(tk, cbase + (uint32_t)e, cbase + (uint32_t)e + half, y0, y1);
        const uint32_t y = hi ? y1 : y0;
        if (mode == LIN_MODE_Z_REPARAM) {
          const float as = alpha * sc_m[e];
          if (tau == 1.0f) {
            const float u = rng_uniform(y, ulo, 1.0f);
            gv = u * __builtin_amdgcn_rcpf(fmaf(1.0f - u, expf(-as), u));
          } else {
            gv = 1.0f / (1.0f + expf(-tau * (rng_logistic(y, tiny) + as)));
          }
        } else {
          gv = (y >> 9) < thr_m[e] ? 1.0f : 0.0f;
        }
      }
    } else {
      gv = lin_sample_g(mode, key, nbits, dd, s, a, j, d, thr_m, sc_m, alpha, tau, layout, tiny);
    }
    GS[e] = gv;
    prior = fmaf(gv, ln_m[e], prior);
    a += st.da;
    j += st.dj;
    if (j >= d) {
      j -= d;
      ++a;
    }
  }
  return prior;
}
// EPT >= ceil(d * d / nthr): all of the thread's table loads are issued before the first product (one trip to the L2 per hidden unit, not one
// per element)
template <int EPT>
__device__ __forceinline__ void nn_build_tw_tab(float* TW, const float* GS, const float* __restrict__ w1t_h, int d, int ldw, int tid, int nthr) {
  const int dd = d * d, pad = ldw - d;
  const float inv_d = 1.0f / (float)d;
  for (int e0 = tid; e0 < dd; e0 += EPT * nthr) {
    float wv[EPT];
#pragma unroll
    for (int q = 0; q < EPT; ++q) {
      const int e = e0 + q * nthr;
      wv[q] = w1t_h[e < dd ? e : dd - 1];
    }
#pragma unroll
    for (int q = 0; q < EPT; ++q) {
      const int e = e0 + q * nthr;
      const int a = (int)(((float)e + 0.5f) * inv_d);  // e / d, exact for e < 2^20 (a carry chain over (a, j) is 25 instructions per element)
      if (e < dd) TW[e + a * pad] = GS[e] * wv[q];
    }
  }
}

// ------------------------------------------------------------------------------------------------
// log p(theta, D | G_s) for all samples.  grid = (ceil(S / spb), Mloc), block = 256
// ------------------------------------------------------------------------------------------------
// NW waves per block (8 when the operands leave room for one block per CU only: two waves per SIMD hide the barriers and
// LDS / L2 waits of the build -> MFMA -> epilogue cycle of every hidden unit)
template <int NT, int NW, int ACT = -1>
__global__ __launch_bounds__(64 * NW) void k_nn_logprobs(const float* __restrict__ x, const int32_t* __restrict__ mask,
                                                     const float* __restrict__ theta, size_t P, const float* __restrict__ scores,
                                                     const uint32_t* __restrict__ thr, float* __restrict__ logprobs, Key2 carry,
                                                     int mode, int m0, int M_global, int d, int N, int S, int spb, float alpha,
                                                     float tau, int layout, int tiny, NNParams np_, int any_mask,
                                                     const float* __restrict__ ln_tab, const float* __restrict__ w1t) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const LinGeom g = lin_geom(d, N, NT);
  constexpr int NU = NW >= 8 ? 1 : 8 / NW, NTHR = 64 * NW;  // row tiles per wave (np / 16 <= 8, i.e. N <= 128)
  float* X = smem;
  float* GS = X + (size_t)g.np * g.ldx;
  float* TW = GS + (size_t)d * d;
  double* red = reinterpret_cast<double*>(smem + ((((size_t)g.np * g.ldx + (size_t)d * d + (size_t)nn_tr_rows(g) * g.ldw) + 3) & ~(size_t)3));
  const int m = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const size_t dd = (size_t)d * d;
  const int H = np_.H;
  const NNOff off = nn_offsets(d, H, np_.bias);
  const float* th_m = theta + (size_t)m * P;
  for (int e = tid; e < g.np * g.ldx; e += NTHR) {
    const int n = e / g.ldx, c = e - n * g.ldx;
    X[e] = (n < N && c < d) ? x[(size_t)n * d + c] : 0.f;
  }
  // graph-independent part of the prior: all leaves except the first-layer weights
  float prior_rest = 0.f;
  for (size_t e = off.b1 + tid; e < off.P; e += NTHR) prior_rest += lin_logn(th_m[e], 0.f, np_.sig_param);
  const Key2 key = (mode == LIN_MODE_GIVEN) ? Key2{0, 0} : lin_mode_key(mode, carry, M_global, m0 + m, layout);
  const uint64_t nbits = (uint64_t)S * dd;
  const float inv2 = 0.5f / np_.obs_noise;
  const float lognorm_x = -0.5f * logf(np_.obs_noise) - 0.918938533204672742f;
  const int nrt = g.np >> 4;
  const bool tab = ln_tab && w1t && mode != LIN_MODE_GIVEN;  // (block-uniform)
  const bool fastg = layout == 0 && (S & 1) == 0 && (uint64_t)S * dd < 0xFFFFFFFFull;
  const TfKeys tk = tf_keys(key);
  // small leaves in LDS (every hidden unit's epilogue reads b1 / W2 of the lane's nodes), validity of the lane's output elements as bits
  float* LV = reinterpret_cast<float*>(red + 64);
  for (int e = tid; e < 2 * d * H + d; e += NTHR) {
    float v = 0.f;
    if (e < d * H) v = np_.bias ? th_m[off.b1 + e] : 0.f;
    else if (e < 2 * d * H) v = th_m[off.w2 + (e - d * H)];
    else v = np_.bias ? th_m[off.b2 + (e - 2 * d * H)] : 0.f;
    LV[e] = v;
  }
  uint32_t okb[NU];
  float nvalid = 0.f;
#pragma unroll
  for (int u = 0; u < NU; ++u) {
    okb[u] = 0u;
#pragma unroll
    for (int tj = 0; tj < NT; ++tj)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = (wave + NW * u) * 16 + (lane >> 4) * 4 + r, j = tj * 16 + (lane & 15);
        const bool v = n < N && j < d && wave + NW * u < (g.np >> 4) && !(any_mask && mask[(size_t)n * d + j]);
        okb[u] |= (uint32_t)v << (tj * 4 + r);
        nvalid += v ? 1.0f : 0.0f;
      }
  }
  static_assert(NT * 4 <= 32, "validity bits of a row tile");
  const int a0 = tid / d, j0 = tid - a0 * d;
  const NNStep st{NTHR / d, NTHR % d};
  // table path: operand rows of 16 NT (+ 16 for even NT) floats, i.e. == 16 mod 32 -- the four k-rows of a B-fragment read then cover all
  // 32 banks (with the d + 2 stride of the shared geometry 39 % of the kernel's LDS cycles were bank conflicts)
  const int ldw_tw = (tab && g.kp * lin_ldw2<NT>() <= nn_tr_rows(g) * g.ldw) ? lin_ldw2<NT>() : g.ldw;  // (even NT: only if it fits the region)
  if (tab)
    for (int e = tid; e < g.kp * ldw_tw; e += NTHR) TW[e] = 0.f;  // padding of the operand: written once
  for (int c = 0; c < spb; ++c) {
    const int s = blockIdx.x * spb + c;
    if (s >= S) break;
    __syncthreads();
    const float pg = tab ? (fastg ? nn_build_graph_tab<true>(GS, mode, key, tk, nbits, s, S, thr + (size_t)m * dd, scores ? scores + (size_t)m * dd : nullptr,
                                                             alpha, tau, layout, tiny, d, tid, a0, j0, st, ln_tab + (size_t)m * dd, NTHR)
                                  : nn_build_graph_tab<false>(GS, mode, key, tk, nbits, s, S, thr + (size_t)m * dd, scores ? scores + (size_t)m * dd : nullptr,
                                                              alpha, tau, layout, tiny, d, tid, a0, j0, st, ln_tab + (size_t)m * dd, NTHR))
                         : nn_build_graph(GS, mode, key, nbits, s, thr + (size_t)m * dd, scores ? scores + (size_t)m * dd : nullptr, alpha,
                                          tau, layout, tiny, d, tid, ln_tab ? ln_tab + (size_t)m * dd : nullptr, NTHR);
    float part = prior_rest + pg;
    f32x4 macc[NU][NT];
#pragma unroll
    for (int u = 0; u < NU; ++u)
#pragma unroll
      for (int tj = 0; tj < NT; ++tj) macc[u][tj] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int h = 0; h < H; ++h) {
      __syncthreads();
      if (tab) nn_build_tw_tab<(NW >= 16 ? 13 : 8)>(TW, GS, w1t + ((size_t)m * H + h) * dd, d, ldw_tw, tid, NTHR);
      else if (ln_tab) nn_build_tw<false>(TW, GS, th_m, h, H, np_.sig_param, g, tid, NTHR);
      else part += nn_build_tw<true>(TW, GS, th_m, h, H, np_.sig_param, g, tid, NTHR);
      __syncthreads();
      f32x4 acc[NU][NT];
      nn_gemm_x_tw<NT, NU, NW>(X, TW, g, lane, wave, acc, ldw_tw);
      // ACT >= 0: the activation is a compile-time constant (relu, the reference's default); a run-time switch per element compiles to a
      // branch ladder of 45 instructions per element -- 44 k of the kernel's 67 k wave-instructions per sample at config 5
#pragma unroll
      for (int u = 0; u < NU; ++u)
#pragma unroll
        for (int tj = 0; tj < NT; ++tj) {
          const int j = tj * 16 + (lane & 15);
          if (j < d && wave + NW * u < nrt) {
            const float b1 = LV[j * H + h];
            const float w2 = LV[d * H + j * H + h];
#pragma unroll
            for (int r = 0; r < 4; ++r) macc[u][tj][r] += w2 * nn_act(ACT >= 0 ? ACT : np_.act, acc[u][tj][r] + b1);
          }
        }
    }
    float sq = 0.f;
#pragma unroll
    for (int u = 0; u < NU; ++u)
#pragma unroll
      for (int tj = 0; tj < NT; ++tj) {
        const int j = tj * 16 + (lane & 15);
        const float b2 = j < d ? LV[2 * d * H + j] : 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int n = (wave + NW * u) * 16 + (lane >> 4) * 4 + r;
          if ((okb[u] >> (tj * 4 + r)) & 1u) {
            const float e = X[n * g.ldx + j] - (macc[u][tj][r] + b2);
            sq = fmaf(e, e, sq);
          }
        }
      }
    part += nvalid * lognorm_x - inv2 * sq;
    const double tot = wave_sum_d((double)part);
    __syncthreads();
    if (lane == 0) red[wave] = tot;
    __syncthreads();
    if (tid == 0) {
      double t = 0.0;
      for (int w = 0; w < NW; ++w) t += red[w];
      logprobs[(size_t)m * S + s] = (float)t;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// softmax-weighted gradients (samples whose weight underflows to 0 in float are skipped, as in the oracle):
//   mode THETA     : grad_theta = sum_s w_s d/dtheta log p(theta, D | G_s)          -> pack row (+ copy of theta)
//   mode Z_REPARAM : W = sum_s w_s (d/dg) o tau alpha g~(1 - g~), off-diagonal     -> w_lik
//   mode Z_SCORE   : W = scale * alpha (sum_s w_s G_s - P), off-diagonal           -> w_lik
// grid = (Mloc, shares), block = 256.  Accumulation goes to global memory; every output element is owned by one thread.
// ------------------------------------------------------------------------------------------------
// NW waves per block: 8 when the operands leave room for ONE block per CU only (two waves per SIMD hide the barriers and the LDS / L2 waits
// of the build -> MFMA -> epilogue cycle of every hidden unit; with 4 the CU ran one wave per SIMD), 4 otherwise.
// ---- k_nn_grad's pieces ----
// accumulate into a partial row / the output without waiting for the old value: global_atomic_add_f32 without return.  One thread owns the
// element (or the adds are separated by block barriers), and same-address operations of a wave execute in issue order, so the sum is the
// sequential one; what it removes is the round trip of a read-modify-write (28 per thread and hidden unit, each past the L2 once hundreds of
// partial rows are in flight: 78 of 320 us per sample gradient at config 5 / step 300, profiles/round6_nn_grad_phases.txt)
__device__ __forceinline__ void nn_acc(float* p, float v) { (void)unsafeAtomicAdd(p, v); }
// sample graph s into GS for the gradient kernel; FAST as in nn_build_graph_tab (paired legacy layout, 32-bit counters, hoisted key schedule)
template <bool FAST>
__device__ __forceinline__ void nn_grad_build_graph(float* GS, int mode, Key2 key, const TfKeys& tk, uint64_t nbits, int s, int S,
                                                    const uint32_t* thr_m, const float* sc_m, float alpha, float tau, int layout, int tiny, int d,
                                                    int tid, int nthr) {
  if constexpr (!FAST) {
    nn_build_graph(GS, mode, key, nbits, s, thr_m, sc_m, alpha, tau, layout, tiny, d, tid, nullptr, nthr);
  } else {
    const int dd = d * d, hS = S >> 1;
    const bool hi = s >= hS;
    const uint32_t cbase = (uint32_t)(hi ? s - hS : s) * (uint32_t)dd, half = (uint32_t)(nbits >> 1);
    const float ulo = tiny ? 1.17549435e-38f : 1.1920929e-07f;
    const float inv_d = 1.0f / (float)d;
    for (int e = tid; e < dd; e += nthr) {
      const int a = (int)(((float)e + 0.5f) * inv_d), j = e - a * d;  // e / d, exact for e < 2^20
      float gv = 0.f;
      if (a != j) {
        uint32_t y0, y1;
        threefry2x32_uk(tk, cbase + (uint32_t)e, cbase + (uint32_t)e + half, y0, y1);
        const uint32_t y = hi ? y1 : y0;
        if (mode == LIN_MODE_Z_REPARAM) {
          const float as = alpha * sc_m[e];
          if (tau == 1.0f) {
            const float u = rng_uniform(y, ulo, 1.0f);
            gv = u * __builtin_amdgcn_rcpf(fmaf(1.0f - u, expf(-as), u));
          } else {
            gv = 1.0f / (1.0f + expf(-tau * (rng_logistic(y, tiny) + as)));
          }
        } else {
          gv = (y >> 9) < thr_m[e] ? 1.0f : 0.0f;
        }
      }
      GS[e] = gv;
    }
  }
}
// T_h = GS o W1T_h into the operand region: the d x d interior with all of a thread's table loads in flight at once (EPT per round), and the
// padding rows d .. kp-1 (the region also holds dpre_h, whose rows overwrite them; its padding COLUMNS only ever receive exact zeros)
template <int EPT>
__device__ __forceinline__ void nn_grad_build_tw(float* TW, const float* GS, const float* __restrict__ w1t_h, const LinGeom g, int tid, int nthr) {
  const int d = g.d, dd = d * d, pad = g.ldw - d;
  const float inv_d = 1.0f / (float)d;
  for (int e0 = tid; e0 < dd; e0 += EPT * nthr) {
    float wv[EPT];
#pragma unroll
    for (int q = 0; q < EPT; ++q) {
      const int e = e0 + q * nthr;
      wv[q] = w1t_h[e < dd ? e : dd - 1];
    }
#pragma unroll
    for (int q = 0; q < EPT; ++q) {
      const int e = e0 + q * nthr;
      const int a = (int)(((float)e + 0.5f) * inv_d);
      if (e < dd) TW[e + a * pad] = GS[e] * wv[q];
    }
  }
  for (int e = d * g.ldw + tid; e < g.kp * g.ldw; e += nthr) TW[e] = 0.f;
}

#ifdef DIBS_NN_STAMPS
static __device__ unsigned long long g_nn_stamps[64];
#define NN_ST(k)                                                        \
  do {                                                                  \
    if (tid == 0) {                                                     \
      const unsigned long long t_ = wall_clock64();                     \
      atomicAdd(&g_nn_stamps[(mode & 3) * 16 + (k)], t_ - st_prev);     \
      st_prev = t_;                                                     \
    }                                                                   \
  } while (0)
#else
#define NN_ST(k)
#endif
template <int NT, int ACT = -1, int NW = 4>   // ACT >= 0: compile-time activation (relu), as in k_nn_logprobs
__global__ __launch_bounds__(64 * NW) void k_nn_grad(const float* __restrict__ x, const int32_t* __restrict__ mask,
                                                 const float* __restrict__ theta, size_t P, const float* __restrict__ scores,
                                                 const uint32_t* __restrict__ thr, const float* __restrict__ logprobs,
                                                 float* __restrict__ out, size_t out_stride, float* __restrict__ theta_copy,
                                                 const float* __restrict__ baseline, float* __restrict__ baseline_out, Key2 carry,
                                                 int mode, int m0, int M_global, int d, int N, int S, float alpha, float tau,
                                                 int layout, int tiny, NNParams np_, double sf_baseline, int any_mask, GradSplit gs,
                                                 const float* __restrict__ w1t) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const LinGeom g = lin_geom(d, N, NT);
  constexpr int NU = 8 / NW, NUD = (NT + NW - 1) / NW, NTHR = 64 * NW;  // row tiles per wave (np / 16 <= 8), column-of-x tiles per wave
  float* X = smem;
  float* GS = X + (size_t)g.np * g.ldx;
  float* TW = GS + (size_t)d * d;                  // T_h [kp][ldw] (forward operand) ...
  float* RS = TW;                                  // ... and dpre_h [np][ldw] (backward operand), same storage
  float* CS = TW + (size_t)nn_tr_rows(g) * g.ldw;  // per-wave column sums [NW][ldw]
  double* red = reinterpret_cast<double*>(smem + ((((size_t)g.np * g.ldx + (size_t)d * d + (size_t)nn_tr_rows(g) * g.ldw + (size_t)2 * NW * g.ldw) + 3) & ~(size_t)3));
  // grid = (Mloc, shares); block (x, y) takes share y of particle (x + y) mod Mloc.  Workgroups go to the 8 XCDs round-robin by their linear
  // id x + Mloc y: with particle = x every share of particle m ran on XCD m mod 8, and the XCD that held the particles with the most weighted
  // samples set the time (150 of 256 CUs busy, 18 ms instead of 8 at config 5 / step 300); rotated by y, a particle's shares land on all
  // XCDs, and row y = 0 -- the only shares with work early in a run -- is still dispatched first.
  const int m = (int)((blockIdx.x + blockIdx.y) % gridDim.x), tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const size_t dd = (size_t)d * d;
  const int H = np_.H;
  const NNOff off = nn_offsets(d, H, np_.bias);
  const float* th_m = theta + (size_t)m * P;
  float* const om_final = out + (size_t)m * out_stride;
  const Key2 key = lin_mode_key(mode, carry, M_global, m0 + m, layout);
  const uint64_t nbits = (uint64_t)S * dd;
  const float* lp = logprobs + (size_t)m * S;
  // softmax statistics and this block's share of the samples with a non-zero weight (GradSplit, kernels_joint.h)
  __shared__ float wch[GRAD_WCH];
  __shared__ int last_flag;
  double mx, den, sm;
  int nnz;
#ifdef DIBS_NN_STAMPS
  unsigned long long st_prev = wall_clock64();
#endif
  grad_softmax_stats<NW>(lp, S, red, mx, den, sm, nnz);
  const int NS = gridDim.y, bz = blockIdx.y, nact = nnz < NS ? (nnz > 0 ? nnz : 1) : NS;
  if (bz >= nact) return;  // (block-uniform: no share -- before anything is staged)
  for (int e = tid; e < g.np * g.ldx; e += NTHR) {
    const int n = e / g.ldx, c = e - n * g.ldx;
    X[e] = (n < N && c < d) ? x[(size_t)n * d + c] : 0.f;
  }
  for (int e = tid; e < 2 * NW * g.ldw; e += NTHR) CS[e] = 0.f;  // [0]: sum_n dmean / dm h, [1]: sum_n dpre (per wave)
  float* const CS1 = CS + NW * g.ldw;
  for (int e = tid; e < nn_tr_rows(g) * g.ldw; e += NTHR) TW[e] = 0.f;  // (operand padding: see nn_grad_build_tw)
  // the accumulation row: the output itself while one block does everything, else this block's partial sums
  // outputs start at zero (theta mode: P entries; z modes: d*d)
  // Several blocks (split): the theta estimator's first-layer gradient -- d*d*H values that every sample updates -- is accumulated in THREAD
  // layout ([slot][thread]: one coalesced read-modify-write per value; in theta's own layout a wave's 64 values lie 4 d H bytes apart, and
  // with hundreds of partial rows in flight the 64-byte sectors of those updates came from HBM: 24 ms per launch at config 5, step 300);
  // the small leaves follow behind it in theta's layout.  The Z modes accumulate d*d values in W's layout (16 consecutive floats per row).
  const bool split = nact > 1;
  const size_t n_out = mode == LIN_MODE_THETA ? P : dd;
  const size_t w1sz = (size_t)H * NUD * NT * 4 * NTHR;  // thread-layout area of the first-layer gradient
  float* const prow = split ? gs.part + ((size_t)m * NS + bz) * gs.stride : nullptr;
  // `om`: where the values in theta's / W's layout accumulate (split + theta mode: only the small leaves, behind the thread-layout area)
  float* const om = !split ? om_final : (mode == LIN_MODE_THETA ? prow + w1sz - off.b1 : prow);
  if (split && mode == LIN_MODE_THETA) {
    for (size_t e = tid; e < w1sz + (P - off.b1); e += NTHR) prow[e] = 0.f;
  } else {
    for (size_t e = tid; e < n_out; e += NTHR) om[e] = 0.f;
  }
  const float inv_on = 1.0f / np_.obs_noise;
  const float inv_sp2 = 1.0f / (np_.sig_param * np_.sig_param);
  const float* sc_m = scores + (size_t)m * dd;
  const uint32_t* thr_m = thr + (size_t)m * dd;
  const int nrt = g.np >> 4;
  const bool fastg = layout == 0 && (S & 1) == 0 && (uint64_t)S * dd < 0xFFFFFFFFull;  // (block-uniform)
  const TfKeys tk = tf_keys(key);
  // which of the lane's output elements are observations that count (not padding, not intervened on): one bit each
  uint32_t okb[NU];
#pragma unroll
  for (int u = 0; u < NU; ++u) {
    okb[u] = 0u;
#pragma unroll
    for (int tj = 0; tj < NT; ++tj)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = (wave + NW * u) * 16 + (lane >> 4) * 4 + r, j = tj * 16 + (lane & 15);
        const bool v = n < N && j < d && wave + NW * u < nrt && !(any_mask && mask[(size_t)n * d + j]);
        okb[u] |= (uint32_t)v << (tj * 4 + r);
      }
  }
  static_assert(NT * 4 <= 32, "validity bits of a row tile");
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the zeroed rows are in place before another wave adds to them (barriers below)

  int q = 0;  // ordinal of the next sample with a non-zero weight
  for (int s0 = 0; s0 < S; s0 += GRAD_WCH) {
    __syncthreads();
    if (tid < GRAD_WCH && s0 + tid < S) wch[tid] = (float)(exp((double)lp[s0 + tid] - mx) / den);
    __syncthreads();
  for (int s = s0; s < S && s < s0 + GRAD_WCH; ++s) {
    const float w = wch[s - s0];
    if (w < GRAD_W_MIN) continue;  // block-uniform
    if ((q++ % NS) != bz) continue;  // (another block's sample)
    __syncthreads();
    NN_ST(0);
    if (fastg) nn_grad_build_graph<true>(GS, mode, key, tk, nbits, s, S, thr_m, sc_m, alpha, tau, layout, tiny, d, tid, NTHR);
    else nn_grad_build_graph<false>(GS, mode, key, tk, nbits, s, S, thr_m, sc_m, alpha, tau, layout, tiny, d, tid, NTHR);
    __syncthreads();
    NN_ST(1);
#ifdef DIBS_NN_STAMPS
    if (tid == 0) atomicAdd(&g_nn_stamps[(mode & 3) * 16 + 12], 1ull);
#endif
    if (mode == LIN_MODE_Z_SCORE) {
      for (int e = tid; e < (int)dd; e += NTHR) om[e] += w * GS[e];
      continue;
    }
    // ---- forward: mean ----
    f32x4 macc[NU][NT];
#pragma unroll
    for (int u = 0; u < NU; ++u)
#pragma unroll
      for (int tj = 0; tj < NT; ++tj) macc[u][tj] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int h = 0; h < H; ++h) {
      __syncthreads();
      if (w1t) nn_grad_build_tw<8>(TW, GS, w1t + ((size_t)m * H + h) * dd, g, tid, NTHR);
      else nn_build_tw<false>(TW, GS, th_m, h, H, np_.sig_param, g, tid, NTHR);
      __syncthreads();
      NN_ST(2);
      f32x4 acc[NU][NT];
      nn_gemm_x_tw<NT, NU, NW>(X, TW, g, lane, wave, acc);
      NN_ST(3);
#pragma unroll
      for (int u = 0; u < NU; ++u)
#pragma unroll
        for (int tj = 0; tj < NT; ++tj) {
          const int j = tj * 16 + (lane & 15);
          if (j < d && wave + NW * u < nrt) {
            const float b1 = np_.bias ? th_m[off.b1 + (size_t)j * H + h] : 0.f;
            const float w2 = th_m[off.w2 + (size_t)j * H + h];
#pragma unroll
            for (int r = 0; r < 4; ++r) macc[u][tj][r] += w2 * nn_act(ACT >= 0 ? ACT : np_.act, acc[u][tj][r] + b1);
          }
        }
    }
    NN_ST(4);
    // dmean = (1 - mask) (x - mean) / obs_noise  (kept in registers, C layout; zero on padding rows / columns)
#pragma unroll
    for (int tj = 0; tj < NT; ++tj) {
      float t = 0.f;
      const float b2 = (np_.bias && tj * 16 + (lane & 15) < d) ? th_m[off.b2 + tj * 16 + (lane & 15)] : 0.f;
#pragma unroll
      for (int u = 0; u < NU; ++u)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int n = (wave + NW * u) * 16 + (lane >> 4) * 4 + r, j = tj * 16 + (lane & 15);
          float dm = 0.f;
          if ((okb[u] >> (tj * 4 + r)) & 1u) dm = (X[n * g.ldx + j] - macc[u][tj][r] - b2) * inv_on;
          macc[u][tj][r] = dm;
          t += dm;
        }
      t += __shfl_xor(t, 16);
      t += __shfl_xor(t, 32);
      if (lane < 16) CS[wave * g.ldw + tj * 16 + lane] = t;
    }
    __syncthreads();
    if (mode == LIN_MODE_THETA && np_.bias)
      for (int j = tid; j < d; j += NTHR)  // d/db2_j = sum_n dmean_nj
        nn_acc(om + off.b2 + j, w * nn_cs_sum<NW>(CS, g.ldw, j));
    // ---- backward, one hidden unit at a time ----
    for (int h = 0; h < H; ++h) {
      __syncthreads();
      NN_ST(5);
      if (w1t) nn_grad_build_tw<8>(TW, GS, w1t + ((size_t)m * H + h) * dd, g, tid, NTHR);
      else nn_build_tw<false>(TW, GS, th_m, h, H, np_.sig_param, g, tid, NTHR);
      __syncthreads();
      NN_ST(6);
      f32x4 acc[NU][NT];
      nn_gemm_x_tw<NT, NU, NW>(X, TW, g, lane, wave, acc);
      __syncthreads();  // every wave is done reading T_h: its storage now takes dpre_h
      NN_ST(7);
#pragma unroll
      for (int tj = 0; tj < NT; ++tj) {
        const int j = tj * 16 + (lane & 15);
        const float b1 = (np_.bias && j < d) ? th_m[off.b1 + (size_t)j * H + h] : 0.f;
        const float w2 = j < d ? th_m[off.w2 + (size_t)j * H + h] : 0.f;
        float t2 = 0.f, t1 = 0.f;
#pragma unroll
        for (int u = 0; u < NU; ++u)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int n = (wave + NW * u) * 16 + (lane >> 4) * 4 + r;
            if (n < g.np && wave + NW * u < nrt) {
              const float pre = acc[u][tj][r] + b1;
              const float hv = nn_act(ACT >= 0 ? ACT : np_.act, pre);
              const float dm = macc[u][tj][r];
              const float dp = dm * w2 * nn_dact(ACT >= 0 ? ACT : np_.act, pre, hv);  // dpre
              RS[n * g.ldw + j] = dp;
              t1 += dp;                                                   // for d/db1
              t2 += dm * hv;                                              // for d/dW2
            }
          }
        t2 += __shfl_xor(t2, 16);
        t2 += __shfl_xor(t2, 32);
        t1 += __shfl_xor(t1, 16);
        t1 += __shfl_xor(t1, 32);
        if (lane < 16) {
          CS[wave * g.ldw + j] = t2;
          CS1[wave * g.ldw + j] = t1;
        }
      }
      __syncthreads();
      NN_ST(8);
      if (mode == LIN_MODE_THETA)
        for (int j = tid; j < d; j += NTHR) {  // (column sums of dpre / dm h by wave, added in wave order)
          if (np_.bias) nn_acc(om + off.b1 + (size_t)j * H + h, w * nn_cs_sum<NW>(CS1, g.ldw, j));
          nn_acc(om + off.w2 + (size_t)j * H + h, w * nn_cs_sum<NW>(CS, g.ldw, j));
        }
      NN_ST(9);
      // xtr[a][j] = sum_n x[n][a] dpre[n][j]  (d/dT_h)
#pragma unroll
      for (int u = 0; u < NUD; ++u) {
        const int ti = wave + NW * u;
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
        NN_ST(10);
#pragma unroll
        for (int tj = 0; tj < NT; ++tj)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int a = ti * 16 + (lane >> 4) * 4 + r, j = tj * 16 + (lane & 15);
            if (a < d && j < d) {
              float xtr = t[tj][r];
              asm volatile("" : "+v"(xtr));
              const float gv = GS[a * d + j];
              if (gv == 0.f) continue;  // no term (hard graphs late in a run: a few per cent of the entries are edges)
              const float w1 = w1t ? w1t[((size_t)m * H + h) * dd + (size_t)a * d + j] : th_m[((size_t)j * d + a) * H + h];
              if (mode == LIN_MODE_THETA) {
                const float v = w * gv * (xtr - w1 * inv_sp2);
                if (split) nn_acc(prow + (size_t)(((h * NUD + u) * NT + tj) * 4 + r) * NTHR + tid, v);
                else nn_acc(om + ((size_t)j * d + a) * H + h, v);
              } else if (a != j) {
                nn_acc(om + a * d + j, w * (lin_logn(w1, 0.f, np_.sig_param) + w1 * xtr) * tau * alpha * gv * (1.0f - gv));
              }
            }
          }
      }
      NN_ST(11);
    }
  }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's adds have been performed
  __syncthreads();
  NN_ST(0);
  if (nact > 1) {
    // release the partial sums, count this block, and the LAST block of the particle
    // adds the rows in block order into the output
    __threadfence();
    if (!grad_last_block(gs.ctr + m, nact, &last_flag)) return;
    __threadfence();
    const float* const base = gs.part + (size_t)m * NS * gs.stride;
    if (mode == LIN_MODE_THETA) {
      for (int h = 0; h < H; ++h)
#pragma unroll
        for (int u = 0; u < NUD; ++u) {
          const int ti = wave + NW * u;
          if (ti >= NT) continue;
#pragma unroll
          for (int tj = 0; tj < NT; ++tj)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int a = ti * 16 + (lane >> 4) * 4 + r, j = tj * 16 + (lane & 15);
              if (a < d && j < d) {
                const size_t slot = (size_t)(((h * NUD + u) * NT + tj) * 4 + r) * NTHR + tid;
                om_final[((size_t)j * d + a) * H + h] = grad_part_sum<GRAD_NS_NN, false>(base, gs.stride, slot, nact);
              }
            }
        }
      for (size_t e = off.b1 + tid; e < P; e += NTHR)  // the small leaves
        om_final[e] = grad_part_sum<GRAD_NS_NN, false>(base, gs.stride, w1sz + (e - off.b1), nact);
    } else {
      for (size_t e = tid; e < n_out; e += NTHR) om_final[e] = grad_part_sum<GRAD_NS_NN, false>(base, gs.stride, e, nact);
    }
    __syncthreads();
    NN_ST(13);
  }
  // epilogue
  const float bold = baseline ? baseline[m] : 0.f;
  if (mode == LIN_MODE_THETA) {
    // graph-independent prior gradient of the remaining leaves: -theta / sig_p^2 (the softmax weights sum to 1)
    for (size_t e = off.b1 + tid; e < off.P; e += NTHR) om_final[e] += -th_m[e] * inv_sp2;
    if (theta_copy)
      for (size_t e = tid; e < P; e += NTHR) theta_copy[(size_t)m * out_stride + e] = th_m[e];
  } else if (mode == LIN_MODE_Z_SCORE) {
    const float scale = sf_baseline > 0.0 ? (float)exp(-(double)bold) : 1.0f;
    for (int e = tid; e < (int)dd; e += NTHR) {
      const int i = e / d, j = e - i * d;
      const float p = (float)sigmoid_d((double)__fmul_rn(alpha, sc_m[e]));
      om_final[e] = i == j ? 0.f : scale * alpha * (om_final[e] - p);
    }
  }
  if (mode != LIN_MODE_THETA && baseline_out && tid == 0)
    baseline_out[m] = (mode == LIN_MODE_Z_SCORE) ? (float)(sf_baseline * (sm / S) + (1.0 - sf_baseline) * (double)bold) : bold;
}

// theta init with the stax key discipline (nonlinearGaussian.py:155-186; stax.serial / Dense of jax.example_libraries):
// subkey(m, j) = row m*d+j of split(key, M*d); per stax layer: rng, layer_rng = split(rng) (the activation layer consumes
// one too); Dense: k1, k2 = split(layer_rng); W = normal(k1, (in, out)) * sig; b = normal(k2, (out,)) * sig.
// one thread per (local particle, node)
#ifdef DIBS_TU_NN
__global__ void k_init_theta_nn(float* __restrict__ theta, size_t P, Key2 key, int m0, int Mloc, int M_global, int d, int H, int bias,
                                float sig, int layout) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= Mloc * d) return;
  const int m = t / d, j = t - m * d;
  const NNOff off = nn_offsets(d, H, bias);
  float* th = theta + (size_t)m * P;
  Key2 rng = rng_split_row(key, (uint32_t)(M_global * d), (uint32_t)((m0 + m) * d + j), layout);
  for (int layer = 0; layer < 3; ++layer) {
    const Key2 lr = rng_split_row(rng, 2u, 1u, layout);
    rng = rng_split_row(rng, 2u, 0u, layout);
    if (layer == 1) continue;  // activation: no parameters
    const int in = layer == 0 ? d : H, outn = layer == 0 ? H : 1;
    const uint64_t nw = (uint64_t)in * outn;
    float* W = layer == 0 ? th + off.w1 + (size_t)j * d * H : th + off.w2 + (size_t)j * H;
    if (bias) {
      const Key2 k1 = rng_split_row(lr, 2u, 0u, layout), k2 = rng_split_row(lr, 2u, 1u, layout);
      for (uint64_t i = 0; i < nw; ++i) W[i] = rng_normal(rng_bits_at(k1, nw, i, layout)) * sig;
      float* B = layer == 0 ? th + off.b1 + (size_t)j * H : th + off.b2 + j;
      for (uint64_t i = 0; i < (uint64_t)outn; ++i) B[i] = rng_normal(rng_bits_at(k2, (uint64_t)outn, i, layout)) * sig;
    } else {
      for (uint64_t i = 0; i < nw; ++i) W[i] = rng_normal(rng_bits_at(lr, nw, i, layout)) * sig;
    }
  }
}
#endif

// ---- host side (defined in tu_nn.hip) ---------------------------------------------------------------
// true: the tuned one-hidden-layer kernels of this file apply; false: the general path of kernels_nn_generic.h runs
bool joint_nn_fast_path(int d, int N, const NNParams& np_);
// (both return non-zero when the scratch area of the general path cannot be allocated)
int joint_nn_dispatch(JointWork* w, const JointLaunch& jl, Key2 carry, int mode, const NNParams& np_, size_t P);
int joint_nn_score_given(const JointWork& jw, const float* theta, const int32_t* g, float* out, int n, int d, int N, const NNParams& np_,
                          size_t P, hipStream_t stream);
// theta = stax initialisation stream of sample_parameters (nonlinearGaussian.py:155-186)
void joint_nn_init_theta(float* theta, size_t P, Key2 key, int m0, int Mloc, int M, int d, const NNParams& np_, int layout, hipStream_t stream);

#ifdef DIBS_TU_NN
#include "kernels_nn_generic.h"
#include "kernels_nn_f16.h"
#include "kernels_nn_f16x.h"
void dibs_allow_lds(const void* kernel, size_t bytes);  // (engine.hip)

// first layer on the f16 matrix pipe (kernels_nn_f16.h): 33 <= d <= 112, Threefry-paired samples, tables allocated; DIBS_NN_F32=1 keeps
// the f32-MFMA kernel (A/B runs).  Returns false when the f32 kernel has to run.
template <int NT>
static bool joint_nn_logprobs_hf(JointWork* w, const JointLaunch& jl, Key2 carry, int mode, const NNParams& np_, size_t P, float* lp) {
  if constexpr (NT < 3) {
    return false;
  } else {
    const bool off = jl.nn_f32 != 0;  // (tuning.h)
    const bool paired = jl.layout == 0 && (jl.S & 1) == 0 && (uint64_t)jl.S * jl.d * jl.d < 0xFFFFFFFFull;
    const bool soft = mode == LIN_MODE_Z_REPARAM;
    const size_t lds = nhf_lds_bytes(jl.d, NT, np_.H, soft);
    if (mode == LIN_MODE_THETA) w->nhf_valid = w->nhx_valid = false;  // (theta moved since the last step)
    if (off || !paired || jl.N > 128 || !w->ln_tab) return false;
    const int hS = jl.S / 2, ppb = (hS / 4) * jl.Mloc >= 1024 ? 4 : (hS >= 2 ? 2 : 1);
    if (!w->nhf_ew && hipMalloc((void**)&w->nhf_ew, (size_t)jl.Mloc * 4) != hipSuccess) {
      (void)hipGetLastError();
      w->nhf_ew = nullptr;
      return false;
    }
    // d >= 65: the per-sample operand in REGISTERS against an x^T image (k_nn_logprobs_hx: no block barrier per hidden unit); below
    // the image variant (k_nn_logprobs_hf: smaller blocks, several per CU)
    if constexpr (NT >= 5) {
      const size_t ldsx = nhx_lds_bytes(jl.d, NT, jl.N, np_.H, soft);
      if (ldsx <= (size_t)160 * 1024 - 512) {
        const size_t quads = (size_t)jl.Mloc * np_.H * ((jl.d + 3) / 4) * jl.d;
        if (w->nhx_quads < quads) {
          if (w->nhx_w1s) hipFree(w->nhx_w1s);
          if (w->nhx_w1p) hipFree(w->nhx_w1p);
          w->nhx_w1s = w->nhx_w1p = nullptr;
          w->nhx_quads = 0;
          w->nhx_valid = false;
          if (hipMalloc(&w->nhx_w1s, quads * 16) != hipSuccess || hipMalloc(&w->nhx_w1p, quads * 16) != hipSuccess) {
            (void)hipGetLastError();
            return false;
          }
          w->nhx_quads = quads;
        }
        if (!w->nhx_valid) {  // theta is the same for both estimators of a step: the tables are built once per step and variant
          hipLaunchKernelGGL(k_nn_w1_exp, dim3(jl.Mloc), dim3(256), 0, jl.stream, jl.theta, P, w->nhf_ew, jl.d, np_.H);
          const int nq = ((jl.d + 3) / 4) * jl.d;
          hipLaunchKernelGGL(k_nn_tables_hx, dim3((nq + 255) / 256, np_.H, jl.Mloc), dim3(256), 0, jl.stream, jl.theta, P, w->nhf_ew,
                             (float4*)w->nhx_w1s, (uint4*)w->nhx_w1p, jl.d, np_.H);
          w->nhx_valid = true;
        }
#define NHX_LAUNCH(NTN_, ACT_, SOFT_)                                                                                                        \
        {                                                                                                                                    \
          dibs_allow_lds((const void*)k_nn_logprobs_hx<NT, NTN_, ACT_, SOFT_>, ldsx);                                                        \
          hipLaunchKernelGGL((k_nn_logprobs_hx<NT, NTN_, ACT_, SOFT_>), dim3((hS + ppb - 1) / ppb, (jl.Mloc + 7) & ~7), dim3(64 * NT), ldsx,  \
                             jl.stream, w->x, w->mask, jl.theta, P, jl.scores, jl.thr, lp, carry, mode, jl.m0, jl.M, jl.Mloc, jl.d, jl.N,    \
                             jl.S, ppb, jl.alpha, jl.tau, jl.layout, jl.tiny, np_, w->any_mask, w->ln_tab, (const float4*)w->nhx_w1s,        \
                             (const uint4*)w->nhx_w1p, w->nhf_ew);                                                                           \
        }
#define NHX_PICK(NTN_)                                                                                                                       \
        if (soft) {                                                                                                                          \
          if (np_.act == 0) NHX_LAUNCH(NTN_, 0, true) else NHX_LAUNCH(NTN_, -1, true)                                                         \
        } else {                                                                                                                             \
          if (np_.act == 0) NHX_LAUNCH(NTN_, 0, false) else NHX_LAUNCH(NTN_, -1, false)                                                       \
        }
        if (jl.N <= 112) { NHX_PICK(7) } else { NHX_PICK(8) }
#undef NHX_PICK
#undef NHX_LAUNCH
        return true;
      }
    }
    if (lds > (size_t)160 * 1024 - 512) return false;
    const size_t pairs = (size_t)jl.Mloc * np_.H * jl.d * (nhf_dp2(jl.d) / 2);
    if (w->nhf_pairs < pairs) {
      if (w->nhf_w1s) hipFree(w->nhf_w1s);
      if (w->nhf_w1p) hipFree(w->nhf_w1p);
      w->nhf_w1s = w->nhf_w1p = nullptr;
      w->nhf_pairs = 0;
      w->nhf_valid = false;
      if (hipMalloc(&w->nhf_w1s, pairs * 8) != hipSuccess || hipMalloc(&w->nhf_w1p, pairs * 8) != hipSuccess) {
        (void)hipGetLastError();
        return false;
      }
      w->nhf_pairs = pairs;
    }
    if (!w->nhf_valid) {  // (once per step and variant, see JointWork)
      hipLaunchKernelGGL(k_nn_w1_exp, dim3(jl.Mloc), dim3(256), 0, jl.stream, jl.theta, P, w->nhf_ew, jl.d, np_.H);
      const int npr = jl.d * (nhf_dp2(jl.d) / 2);
      hipLaunchKernelGGL(k_nn_tables_hf, dim3((npr + 255) / 256, np_.H, jl.Mloc), dim3(256), 0, jl.stream, jl.theta, P, w->nhf_ew,
                         (float2*)w->nhf_w1s, (uint2*)w->nhf_w1p, jl.d, np_.H);
      w->nhf_valid = true;
    }
#define NHF_LAUNCH(ACT_, SOFT_)                                                                                                               \
    {                                                                                                                                         \
      dibs_allow_lds((const void*)k_nn_logprobs_hf<NT, ACT_, SOFT_>, lds);                                                                    \
      hipLaunchKernelGGL((k_nn_logprobs_hf<NT, ACT_, SOFT_>), dim3((hS + ppb - 1) / ppb, (jl.Mloc + 7) & ~7), dim3(NHF_NTHR), lds, jl.stream, \
                         w->x, w->mask, jl.theta, P, jl.scores, jl.thr, lp, carry, mode, jl.m0, jl.M, jl.Mloc, jl.d, jl.N, jl.S, ppb, jl.alpha, jl.tau, \
                         jl.layout, jl.tiny, np_, w->any_mask, w->ln_tab, (const float2*)w->nhf_w1s, (const uint2*)w->nhf_w1p, w->nhf_ew);    \
    }
    if (soft) {
      if (np_.act == 0) NHF_LAUNCH(0, true) else NHF_LAUNCH(-1, true)
    } else {
      if (np_.act == 0) NHF_LAUNCH(0, false) else NHF_LAUNCH(-1, false)
    }
#undef NHF_LAUNCH
    return true;
  }
}

template <int NT>
static void joint_nn_launch(JointWork* w, const JointLaunch& jl, Key2 carry, int mode, const NNParams& np_, size_t P) {
  // samples per block: the block's prologue (x and the small leaves into LDS, validity bits) is shared by them; 4 while that leaves at least
  // four rounds of blocks (config 5: spb 2 / 4 / 8 -> 51.2 / 53.0 / 53.1 steps/s)
  const int spb = (jl.S / 4) * jl.Mloc >= 1024 ? 4 : 2;
  const size_t lds1 = nn_lds_bytes_logprobs(jl.d, jl.N, NT, np_.H), lds2 = nn_lds_bytes(jl.d, jl.N, NT, true);
  float* lp = mode == LIN_MODE_THETA ? jl.logprobs_th : jl.logprobs_z;
  const size_t w1t_need = (size_t)jl.Mloc * np_.H * jl.d * jl.d;
  if (w->w1t_floats < w1t_need) {  // (first launch; optional: without it the kernel reads W1 in place)
    if (w->w1t) hipFree(w->w1t);
    w->w1t = nullptr;
    w->w1t_floats = hipMalloc((void**)&w->w1t, w1t_need * 4) == hipSuccess ? w1t_need : 0;
    if (!w->w1t_floats) w->w1t = nullptr;
  }
  if (mode == LIN_MODE_THETA && w->ln_tab)  // theta is the same for both estimators of a step: the tables are built once (theta runs first)
    hipLaunchKernelGGL(k_nn_prior_table, dim3((jl.d * jl.d + 255) / 256, jl.Mloc), dim3(256), 0, jl.stream, jl.theta, P, w->ln_tab, w->w1t, jl.d,
                       np_.H, np_.sig_param);
#define NN_LP_LAUNCH(NW_, ACT_)                                                                                                             \
  {                                                                                                                                         \
    if (lds1 > 48 * 1024) hipFuncSetAttribute((const void*)k_nn_logprobs<NT, NW_, ACT_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1); \
    hipLaunchKernelGGL((k_nn_logprobs<NT, NW_, ACT_>), dim3((jl.S + spb - 1) / spb, jl.Mloc), dim3(64 * NW_), lds1, jl.stream, w->x, w->mask,   \
                       jl.theta, P, jl.scores, jl.thr, lp, carry, mode, jl.m0, jl.M, jl.d, jl.N, jl.S, spb, jl.alpha, jl.tau, jl.layout,    \
                       jl.tiny, np_, w->any_mask, w->ln_tab, w->w1t);                                                                       \
  }
  if (joint_nn_logprobs_hf<NT>(w, jl, carry, mode, np_, P, lp)) {
    // (log-probs done on the f16 matrix pipe)
  } else if (lds1 > 80 * 1024) {  // one block per CU: run it with 16 waves
    if (np_.act == 0) NN_LP_LAUNCH(16, 0) else NN_LP_LAUNCH(16, -1)
  } else {
    if (np_.act == 0) NN_LP_LAUNCH(4, 0) else NN_LP_LAUNCH(4, -1)
  }
#undef NN_LP_LAUNCH
  float* out = mode == LIN_MODE_THETA ? jl.pack + (size_t)jl.m0 * jl.pack_stride + jl.gtheta_off : jl.w_lik;
  const size_t ostride = mode == LIN_MODE_THETA ? jl.pack_stride : (size_t)jl.d * jl.d;
  float* tcopy = (mode == LIN_MODE_THETA && jl.copy_theta) ? jl.pack + (size_t)jl.m0 * jl.pack_stride + jl.theta_off : nullptr;
  GradSplit gs;  // (several blocks per particle once many samples keep a non-zero weight: kernels_joint.h; the row is P floats for theta, d*d for Z)
  // (partial row of a block: the first-layer gradient in thread layout + the small leaves for theta, d*d for Z; the shares per particle are
  //  cut back when GRAD_NS_NN rows per particle would exceed 4 GiB -- hidden widths in the dozens)
  const bool wide = lds2 > 80 * 1024;  // one block per CU: 8 waves (see k_nn_grad)
  const int nthr = wide ? 512 : 256, nud = wide ? (NT + 7) / 8 : (NT + 3) / 4;
  const size_t row_theta = (size_t)np_.H * nud * NT * 4 * nthr + (P - (size_t)jl.d * jl.d * np_.H);
  const size_t row = row_theta > (size_t)jl.d * jl.d ? row_theta : (size_t)jl.d * jl.d;
  int ns_nn = GRAD_NS_NN;
  while (ns_nn > 1 && (size_t)jl.Mloc * ns_nn * row * 4 > ((size_t)4 << 30)) ns_nn >>= 1;
  if (!joint_grad_split(w, (size_t)jl.Mloc, row, &gs, ns_nn)) return;  // (the step's launch check reports the failed hipMalloc)
#define NN_GRAD_LAUNCH(ACT_, NW_)                                                                                                              \
  {                                                                                                                                            \
    if (lds2 > 48 * 1024) hipFuncSetAttribute((const void*)k_nn_grad<NT, ACT_, NW_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2);   \
    hipLaunchKernelGGL((k_nn_grad<NT, ACT_, NW_>), dim3(jl.Mloc, ns_nn), dim3(64 * NW_), lds2, jl.stream, w->x, w->mask, jl.theta, P, jl.scores,  \
                       jl.thr, lp, out, ostride, tcopy, jl.baseline, mode == LIN_MODE_THETA ? (float*)nullptr : jl.baseline_out, carry, mode,    \
                       jl.m0, jl.M, jl.d, jl.N, jl.S, jl.alpha, jl.tau, jl.layout, jl.tiny, np_, jl.sf_baseline, w->any_mask, gs,                \
                       w->ln_tab ? w->w1t : nullptr);                                                                                         \
  }
  if (wide) {
    if (np_.act == 0) NN_GRAD_LAUNCH(0, 8) else NN_GRAD_LAUNCH(-1, 8)
  } else {
    if (np_.act == 0) NN_GRAD_LAUNCH(0, 4) else NN_GRAD_LAUNCH(-1, 4)
  }
#undef NN_GRAD_LAUNCH
}

#ifdef DIBS_NN_STAMPS
extern "C" void dibs_debug_nn_stamps(unsigned long long* out, int reset) {
  hipDeviceSynchronize();
  hipMemcpyFromSymbol(out, HIP_SYMBOL(g_nn_stamps), sizeof(unsigned long long) * 64);
  if (reset) {
    unsigned long long z[64] = {0};
    hipMemcpyToSymbol(HIP_SYMBOL(g_nn_stamps), z, sizeof(z));
  }
}
#endif
bool joint_nn_fast_path(int d, int N, const NNParams& np_) {
  return np_.n_hidden == 1 && np_.H >= 1 && np_.H <= 64 && N <= 128 && d <= 112 && nn_lds_bytes(d, N, (d + 15) / 16, true) <= (size_t)160 * 1024 - 2048;  // (2 KiB for the static LDS of k_nn_grad)
}

// scratch of the general path: grown on first use (activation records of the work items; see kernels_nn_generic.h)
static float* nng_scratch(JointWork* w, size_t floats) {
  if (w->nng_scratch_floats < floats) {
    if (w->nng_scratch) hipFree(w->nng_scratch);
    w->nng_scratch = nullptr;
    w->nng_scratch_floats = 0;
    if (hipMalloc((void**)&w->nng_scratch, floats * 4) != hipSuccess) return nullptr;
    w->nng_scratch_floats = floats;
  }
  return w->nng_scratch;
}

// blocks of the persistent log-prob kernel (each owns 256 * hsum floats of activation records)
static int nng_blocks(long work) { return (int)(work < 2048 ? work : 2048); }

static int joint_nng_launch(JointWork* w, const JointLaunch& jl, Key2 carry, int mode, const NNParams& np_) {
  const NNNet net = nn_net(jl.d, np_);
  size_t lds = (((size_t)jl.d * jl.d + 3) & ~(size_t)3) * 4 + 128;
  const int nb = nng_blocks((long)jl.S * jl.Mloc);
  // shares per particle of the gradient kernel (GradSplit): as many as keep its per-block activation records within 2 GiB and its partial
  // rows within 2 GiB
  int ns_g = GRAD_NS;
  const size_t rec = (size_t)2 * net.hsum * jl.d * jl.N, row = (size_t)net.P > (size_t)jl.d * jl.d ? (size_t)net.P : (size_t)jl.d * jl.d;
  while (ns_g > 1 && ((size_t)jl.Mloc * ns_g * rec * 4 > ((size_t)2 << 30) || (size_t)jl.Mloc * ns_g * row * 4 > ((size_t)2 << 30))) ns_g >>= 1;
  const size_t need1 = (size_t)nb * 256 * net.hsum, need2 = (size_t)jl.Mloc * ns_g * rec;
  float* scr = nng_scratch(w, need1 > need2 ? need1 : need2);
  if (!scr) return 1;
  float* gs = nullptr;  // (n_vars > 198: the sampled graph of a block does not fit LDS -- global scratch)
  if (lds > (size_t)160 * 1024 - 1024) {
    const size_t nblk = (size_t)jl.Mloc * ns_g;
    gs = joint_gs_scratch(w, ((size_t)nb > nblk ? (size_t)nb : nblk) * jl.d * jl.d);
    if (!gs) return 1;
    lds = 256;
  }
  GradSplit gsp;
  if (!joint_grad_split(w, (size_t)jl.Mloc, row, &gsp, ns_g)) return 1;
  float* lp = mode == LIN_MODE_THETA ? jl.logprobs_th : jl.logprobs_z;
  if (lds > 48 * 1024) {
    hipFuncSetAttribute((const void*)k_nng_logprobs, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipFuncSetAttribute((const void*)k_nng_grad, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  }
  hipLaunchKernelGGL(k_nng_logprobs, dim3(nb), dim3(256), lds, jl.stream, w->x, w->mask, jl.theta, jl.scores, jl.thr, lp, carry, mode,
                     jl.m0, jl.M, jl.d, jl.N, jl.S, jl.alpha, jl.tau, jl.layout, jl.tiny, np_, w->any_mask, scr, jl.Mloc, gs);
  float* out = mode == LIN_MODE_THETA ? jl.pack + (size_t)jl.m0 * jl.pack_stride + jl.gtheta_off : jl.w_lik;
  const size_t ostride = mode == LIN_MODE_THETA ? jl.pack_stride : (size_t)jl.d * jl.d;
  float* tcopy = (mode == LIN_MODE_THETA && jl.copy_theta) ? jl.pack + (size_t)jl.m0 * jl.pack_stride + jl.theta_off : nullptr;
  hipLaunchKernelGGL(k_nng_grad, dim3(jl.Mloc, ns_g), dim3(256), lds, jl.stream, w->x, w->mask, jl.theta, jl.scores, jl.thr, lp, out, ostride, tcopy,
                     jl.baseline, mode == LIN_MODE_THETA ? (float*)nullptr : jl.baseline_out, carry, mode, jl.m0, jl.M, jl.d, jl.N, jl.S,
                     jl.alpha, jl.tau, jl.layout, jl.tiny, np_, jl.sf_baseline, w->any_mask, scr, gs, gsp);
  return 0;
}

int joint_nn_dispatch(JointWork* w, const JointLaunch& jl, Key2 carry, int mode, const NNParams& np_, size_t P) {
  if (!joint_nn_fast_path(jl.d, jl.N, np_)) {
    return joint_nng_launch(w, jl, carry, mode, np_);
  }
  switch ((jl.d + 15) / 16) {
    case 1: joint_nn_launch<1>(w, jl, carry, mode, np_, P); break;
    case 2: joint_nn_launch<2>(w, jl, carry, mode, np_, P); break;
    case 3: joint_nn_launch<3>(w, jl, carry, mode, np_, P); break;
    case 4: joint_nn_launch<4>(w, jl, carry, mode, np_, P); break;
    case 5: joint_nn_launch<5>(w, jl, carry, mode, np_, P); break;
    case 6: joint_nn_launch<6>(w, jl, carry, mode, np_, P); break;
    default: joint_nn_launch<7>(w, jl, carry, mode, np_, P); break;
  }
  return 0;
}

template <int NT>
static void launch_nn_given(const JointWork& jw, const float* theta, const int32_t* g, float* out, int n, int d, int N,
                            const NNParams& np_, size_t P, hipStream_t stream) {
  const size_t lds = nn_lds_bytes_logprobs(d, N, NT, np_.H);
  if (lds > 48 * 1024) hipFuncSetAttribute((const void*)k_nn_logprobs<NT, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL((k_nn_logprobs<NT, 4>), dim3(1, n), dim3(256), lds, stream, jw.x, jw.mask, theta, P, (const float*)nullptr,
                     reinterpret_cast<const uint32_t*>(g), out, Key2{0, 0}, (int)LIN_MODE_GIVEN, 0, n, d, N, 1, 1, 0.f, 1.f, 0, 0, np_,
                     jw.any_mask, (const float*)nullptr, (const float*)nullptr);
}
int joint_nn_score_given(const JointWork& jw, const float* theta, const int32_t* g, float* out, int n, int d, int N, const NNParams& np_,
                         size_t P, hipStream_t stream) {
  if (!joint_nn_fast_path(d, N, np_)) {
    const NNNet net = nn_net(d, np_);
    size_t lds = (((size_t)d * d + 3) & ~(size_t)3) * 4 + 128;
    const int nb = nng_blocks(n);
    float* scr = nng_scratch(const_cast<JointWork*>(&jw), (size_t)nb * 256 * net.hsum);
    if (!scr) return 1;
    float* gs = nullptr;
    if (lds > (size_t)160 * 1024 - 1024) {
      gs = joint_gs_scratch(const_cast<JointWork*>(&jw), (size_t)nb * d * d);
      if (!gs) return 1;
      lds = 256;
    }
    if (lds > 48 * 1024) hipFuncSetAttribute((const void*)k_nng_logprobs, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(k_nng_logprobs, dim3(nb), dim3(256), lds, stream, jw.x, jw.mask, theta, (const float*)nullptr,
                       reinterpret_cast<const uint32_t*>(g), out, Key2{0, 0}, (int)LIN_MODE_GIVEN, 0, n, d, N, 1, 0.f, 1.f, 0, 0, np_, jw.any_mask, scr, n, gs);
    return 0;
  }
  switch ((d + 15) / 16) {
    case 1: launch_nn_given<1>(jw, theta, g, out, n, d, N, np_, P, stream); break;
    case 2: launch_nn_given<2>(jw, theta, g, out, n, d, N, np_, P, stream); break;
    case 3: launch_nn_given<3>(jw, theta, g, out, n, d, N, np_, P, stream); break;
    case 4: launch_nn_given<4>(jw, theta, g, out, n, d, N, np_, P, stream); break;
    case 5: launch_nn_given<5>(jw, theta, g, out, n, d, N, np_, P, stream); break;
    case 6: launch_nn_given<6>(jw, theta, g, out, n, d, N, np_, P, stream); break;
    default: launch_nn_given<7>(jw, theta, g, out, n, d, N, np_, P, stream); break;
  }
  return 0;
}
void joint_nn_init_theta(float* theta, size_t P, Key2 key, int m0, int Mloc, int M, int d, const NNParams& np_, int layout, hipStream_t stream) {
  const int nt = Mloc * d;
  (void)P;
  hipLaunchKernelGGL(k_nng_init_theta, dim3((nt + 63) / 64), dim3(64), 0, stream, theta, key, m0, Mloc, M, d, np_, layout);
}
#endif  // DIBS_TU_NN
