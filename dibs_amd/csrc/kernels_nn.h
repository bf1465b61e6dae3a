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
#include <type_traits>
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

// k_nn_logprobs: LDS (floats) X[np][ldx] | GS[d*d] | TR[max(kp, np)][ldw] (the operand T_h) | red | the small leaves b1 | W2 | b2
// ([d][H] | [d][H] | [d]).  (k_nn_grad has its own layout: nn_grad_lds_floats.)
__host__ __device__ inline int nn_tr_rows(const LinGeom g) { return g.kp > g.np ? g.kp : g.np; }
__host__ __device__ inline size_t nn_lds_bytes(int d, int N, int NT) {
  const LinGeom g = lin_geom(d, N, NT);
  const size_t f = (size_t)g.np * g.ldx + (size_t)d * d + (size_t)nn_tr_rows(g) * g.ldw;
  return ((f * 4 + 15) & ~(size_t)15) + 64 * 8;
}
__host__ __device__ inline size_t nn_lds_bytes_logprobs(int d, int N, int NT, int H) {
  return nn_lds_bytes(d, N, NT) + (((size_t)2 * d * H + d) * 4 + 15 & ~(size_t)15);
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
// evaluations per element of every sampled graph.  Also W1T[h][a][j] = W1[j][a][h] (what the kernels multiply the sampled graph with, element
// by element).  A block takes 16 nodes x 16 inputs with all hidden units through LDS (reads in runs of 16 H floats per node, writes in runs
// of 16 nodes).   grid = (ceil(d / 16) nodes, ceil(d / 16) inputs, Mloc), block = 256, dynamic LDS = 256 H floats
#ifdef DIBS_TU_NN
__global__ __launch_bounds__(256) void k_nn_prior_table(const float* __restrict__ theta, size_t P, float* __restrict__ ln_tab, float* __restrict__ w1t,
                                                        int d, int H, float sigp) {
  extern __shared__ __attribute__((aligned(16))) float tl[];  // [16 nodes][16 inputs][H]
  const int j0 = blockIdx.x * 16, a0 = blockIdx.y * 16, m = blockIdx.z, tid = threadIdx.x;
  const float* w = theta + (size_t)m * P;
  const int run = 16 * H;
  for (int i = tid; i < 16 * run; i += 256) {
    const int jj = i / run, r = i - jj * run, a = a0 + r / H;
    tl[i] = (j0 + jj < d && a < d) ? w[((size_t)(j0 + jj) * d + a0) * H + r] : 0.f;
  }
  __syncthreads();
  const int jj = tid & 15, al = tid >> 4, a = a0 + al, j = j0 + jj;
  if (a >= d || j >= d) return;
  const int e = a * d + j;
  float t = 0.f;
  for (int h = 0; h < H; ++h) {  // (sum in h order, as before)
    const float wv = tl[jj * run + al * H + h];
    t += lin_logn(wv, 0.f, sigp);
    if (w1t) w1t[((size_t)m * H + h) * d * d + e] = wv;
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
// softmax-weighted gradients (samples whose weight is below GRAD_W_MIN are skipped; the oracle skips those that underflow to 0):
//   mode THETA     : grad_theta = sum_s w_s d/dtheta log p(theta, D | G_s)          -> pack row (+ copy of theta)
//   mode Z_REPARAM : W = sum_s w_s (d/dg) o tau alpha g~(1 - g~), off-diagonal     -> w_lik
//   mode Z_SCORE   : W = scale * alpha (sum_s w_s G_s - P), off-diagonal           -> w_lik
// grid = (Mloc, shares), block = 64 NW.
//
// One sample gradient, ONE COLUMN TILE (16 nodes j) AT A TIME -- the nodes' networks are independent (mean_nj depends on column j of every
// T_h only), so for a column tile the whole chain runs with the activations of all hidden units in registers:
//   slices T_h[:, tile] of up to NN_HC hidden units -> LDS;  pre_h = x T_h (f32 MFMA) -> h_h = act(pre_h + b1) kept in registers;
//   dmean;  dpre_h = dmean w2 act'(.) -> LDS slices (same storage);  xtr_h = x^T dpre_h (f32 MFMA);  gradient terms.
// Round 5's kernel walked hidden units outermost over ALL columns: h_h could not stay anywhere (d = 100: 140 registers per thread, LDS full),
// so the backward pass rebuilt T_h and repeated the forward GEMM -- 15 GEMMs and 10 operand builds per sample where 10 and 5 are needed, at
// 256 VGPRs with spills.  (More hidden units than NN_HC: processed in groups, and only then the forward product is repeated.)
// The first-layer gradient -- d*d*H values that every sample updates -- accumulates in a partial row in THREAD layout ([slot][thread]) with
// no-return atomic adds, for the entries that are edges only; the block that finishes a particle last (or its only block) adds the partial
// rows in block order and writes theta's layout through LDS (theta's own layout puts a wave's 64 values 4 d H bytes apart).
// ------------------------------------------------------------------------------------------------
#define NN_HC 8
#define NN_SLR 128  // rows of a slice (kp <= 112, np <= 128): a compile-time stride keeps the hidden units' LDS offsets in the instructions
// LDS floats: X[np][ldx] | GS[d*d] | SL[hcs][NN_SLR][16] | CS[8][16] | CS2[hcs][2][8][16] | red     (hcs = hidden units per group)
__host__ __device__ inline size_t nn_grad_lds_floats(int d, int N, int NT, int hcs) {
  const LinGeom g = lin_geom(d, N, NT);
  return (size_t)g.np * g.ldx + (size_t)d * d + (size_t)hcs * NN_SLR * 16 + (size_t)8 * 16 + (size_t)hcs * 2 * 8 * 16;
}
__host__ __device__ inline size_t nn_grad_lds_bytes(int d, int N, int NT, int hcs) {
  return ((nn_grad_lds_floats(d, N, NT, hcs) * 4 + 15) & ~(size_t)15) + 64 * 8;
}
// hidden units per group: as many as registers (NN_HC) and LDS hold (2 KiB stay free for the kernel's static LDS); 0: does not fit at all
inline int nn_grad_hcs(int d, int N, int NT, int H) {
  int hcs = H < NN_HC ? H : NN_HC;
  while (hcs > 0 && nn_grad_lds_bytes(d, N, NT, hcs) > (size_t)160 * 1024 - 2048) --hcs;
  return hcs;
}

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
// operand slices of column tile tj for hidden units h0 .. h0+hn-1: SL[hc][a][jl] = GS[a][j] W1[j][a][h0+hc], j = 16 tj + jl; zero for a >= d,
// j >= d.  All of a thread's weight loads of a round (EPT) are in flight together.  w1t_m: the particle's re-laid-out weights W1T[h][a][j]
// (k_nn_prior_table; coalesced).
template <int EPT>
__device__ __forceinline__ void nn_grad_build_slices(float* SL, const float* GS, const float* __restrict__ w1t_m, int h0, int hn, int tj,
                                                     const LinGeom g, int tid, int nthr) {
  const int d = g.d, kp = g.kp, tot = hn * kp * 16;
  const float inv_kp = 1.0f / (float)kp;
  for (int i0 = tid; i0 < tot; i0 += EPT * nthr) {
    float wv[EPT];
#pragma unroll
    for (int q = 0; q < EPT; ++q) {
      const int i = i0 + q * nthr, row = i >> 4, j = tj * 16 + (i & 15);
      const int hc = (int)(((float)row + 0.5f) * inv_kp), a = row - hc * kp;
      const bool ok = i < tot && a < d && j < d;
      wv[q] = !ok ? 0.f : w1t_m[((size_t)(h0 + hc) * d + a) * d + j];
    }
#pragma unroll
    for (int q = 0; q < EPT; ++q) {
      const int i = i0 + q * nthr, row = i >> 4, j = tj * 16 + (i & 15);
      const int hc = (int)(((float)row + 0.5f) * inv_kp), a = row - hc * kp;
      if (i < tot) SL[(hc * NN_SLR + a) * 16 + (i & 15)] = (a < d && j < d) ? GS[a * d + j] * wv[q] : 0.f;
    }
  }
}
// pre_hc = x[row tiles of this wave] * SL[hc] for hc < HN (A fragment loaded once per k-step for all hidden units; two k-steps per trip so the
// second step's LDS reads are in flight during the first step's MFMAs)
template <int HN, int NU, int NW>
__device__ __forceinline__ void nn_grad_gemm_fwd(const float* X, const float* SL, const LinGeom g, int lane, int wave, f32x4 (&acc)[NN_HC][NU]) {
  const int nrt = g.np >> 4;
#pragma unroll
  for (int u = 0; u < NU; ++u) {
#pragma unroll
    for (int hc = 0; hc < HN; ++hc) acc[hc][u] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int ti = wave + NW * u;
    if (ti >= nrt) continue;
    const float* ap = X + (ti * 16 + (lane & 15)) * g.ldx + (lane >> 4);
    const float* bq = SL + (lane >> 4) * 16 + (lane & 15);
    int k0 = 0;
    for (; k0 + 4 < g.kp; k0 += 8) {
      const float a0 = ap[0], a1 = ap[4];
      float b0[HN], b1[HN];
#pragma unroll
      for (int hc = 0; hc < HN; ++hc) {
        b0[hc] = bq[hc * NN_SLR * 16];
        b1[hc] = bq[hc * NN_SLR * 16 + 64];
      }
#pragma unroll
      for (int hc = 0; hc < HN; ++hc) acc[hc][u] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b0[hc], acc[hc][u], 0, 0, 0);
#pragma unroll
      for (int hc = 0; hc < HN; ++hc) acc[hc][u] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b1[hc], acc[hc][u], 0, 0, 0);
      ap += 8;
      bq += 128;
    }
    if (k0 < g.kp) {
      const float a0 = ap[0];
#pragma unroll
      for (int hc = 0; hc < HN; ++hc) acc[hc][u] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, bq[hc * NN_SLR * 16], acc[hc][u], 0, 0, 0);
    }
#pragma unroll
    for (int hc = 0; hc < HN; ++hc)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float v = acc[hc][u][r];
        asm volatile("" : "+v"(v));  // keep MFMA results out of AGPR-sourced stores (see lds_matmul)
        acc[hc][u][r] = v;
      }
  }
}
// xtr_hc[a][jl] = sum_n x[n][a] dpre_hc[n][jl] for the a-tile ti, hc < HN  (np is a multiple of 16: two k-steps per trip)
template <int HN>
__device__ __forceinline__ void nn_grad_gemm_xtr(const float* X, const float* SL, const LinGeom g, int lane, int ti, f32x4 (&t)[NN_HC]) {
#pragma unroll
  for (int hc = 0; hc < HN; ++hc) t[hc] = f32x4{0.f, 0.f, 0.f, 0.f};
  const float* ap = X + (lane >> 4) * g.ldx + ti * 16 + (lane & 15);
  const float* bq = SL + (lane >> 4) * 16 + (lane & 15);
  const int st = 4 * g.ldx;
  for (int k0 = 0; k0 < g.np; k0 += 8) {
    const float a0 = ap[0], a1 = ap[st];
    float b0[HN], b1[HN];
#pragma unroll
    for (int hc = 0; hc < HN; ++hc) {
      b0[hc] = bq[hc * NN_SLR * 16];
      b1[hc] = bq[hc * NN_SLR * 16 + 64];
    }
#pragma unroll
    for (int hc = 0; hc < HN; ++hc) t[hc] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b0[hc], t[hc], 0, 0, 0);
#pragma unroll
    for (int hc = 0; hc < HN; ++hc) t[hc] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b1[hc], t[hc], 0, 0, 0);
    ap += 2 * st;
    bq += 128;
  }
#pragma unroll
  for (int hc = 0; hc < HN; ++hc)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float v = t[hc][r];
      asm volatile("" : "+v"(v));
      t[hc][r] = v;
    }
}
#define NN_HN_SWITCH(hn_, CALL)      \
  switch (hn_) {                     \
    case 1: { constexpr int HN = 1; CALL; } break; \
    case 2: { constexpr int HN = 2; CALL; } break; \
    case 3: { constexpr int HN = 3; CALL; } break; \
    case 4: { constexpr int HN = 4; CALL; } break; \
    case 5: { constexpr int HN = 5; CALL; } break; \
    case 6: { constexpr int HN = 6; CALL; } break; \
    case 7: { constexpr int HN = 7; CALL; } break; \
    default: { constexpr int HN = 8; CALL; } break; \
  }

#ifdef DIBS_NN_STAMPS
static __device__ unsigned long long g_nn_stamps[128];
#define NN_ST(k)                                                        \
  do {                                                                  \
    if (tid == 0) {                                                     \
      const unsigned long long t_ = wall_clock64();                     \
      atomicAdd(&g_nn_stamps[(mode & 3) * 32 + (k)], t_ - st_prev);     \
      st_prev = t_;                                                     \
    }                                                                   \
  } while (0)
#else
#define NN_ST(k)
#endif
template <int ACT = -1, int NW = 4>   // ACT >= 0: compile-time activation (relu), as in k_nn_logprobs; NW waves (8 when one block fills a CU's LDS)
__global__ __launch_bounds__(64 * NW) void k_nn_grad(const float* __restrict__ x, const int32_t* __restrict__ mask,
                                                 const float* __restrict__ theta, size_t P, const float* __restrict__ scores,
                                                 const uint32_t* __restrict__ thr, const float* __restrict__ logprobs,
                                                 float* __restrict__ out, size_t out_stride, float* __restrict__ theta_copy,
                                                 const float* __restrict__ baseline, float* __restrict__ baseline_out, Key2 carry,
                                                 int mode, int m0, int M_global, int d, int N, int S, float alpha, float tau,
                                                 int layout, int tiny, NNParams np_, double sf_baseline, int any_mask, GradSplit gs,
                                                 const float* __restrict__ w1t, const float* __restrict__ ln_tab, int NT, int hcs, GradPlan gp,
                                                 int NS) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const LinGeom g = lin_geom(d, N, NT);
  constexpr int NU = 8 / NW, NUDM = (7 + NW - 1) / NW, NTHR = 64 * NW, HC = NN_HC;  // row tiles per wave (np / 16 <= 8), a-tiles per wave (NT <= 7)
  const int H = np_.H;  // hcs <= HC: hidden units per group (launcher: nn_grad_hcs)
  float* X = smem;
  float* GS = X + (size_t)g.np * g.ldx;
  float* SL = GS + (size_t)d * d;         // T slices [hc][a < kp][16] and, after a barrier, dpre slices [hc][n < np][16]; NN_SLR rows apart
  float* CS = SL + (size_t)hcs * NN_SLR * 16;  // per-wave column sums of dmean [NW][16]
  float* CS2 = CS + 8 * 16;                 // per-wave column sums of dm h / dpre [hc][2][NW][16]
  const size_t lds_floats = nn_grad_lds_floats(d, N, NT, hcs);
  // PERSISTENT blocks (one or two per CU) take work items (particle, share) off the plan's list (k_grad_plan, kernels_joint.h) until it is
  // empty.  With one block per (particle, share) -- 4 096 at config 5, each needing the CU's whole LDS -- the 3 840 that have nothing to do
  // early in a run queued behind the 256 that do, and every block paid the prologue (x into LDS, validity bits): 512 us per launch where the
  // work of a block was 280.  An item's partial row and its place in the sum depend on (particle, share) only, never on which block ran it.
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, jl = lane & 15;
  const size_t dd = (size_t)d * d;
  const NNOff off = nn_offsets(d, H, np_.bias);
  const uint64_t nbits = (uint64_t)S * dd;
  __shared__ float wch[GRAD_WCH];
  __shared__ unsigned short lst[GRAD_WCH];
  __shared__ int lst_n[2];
  __shared__ int last_flag;
  __shared__ unsigned int cur_item;
#ifdef DIBS_NN_STAMPS
  unsigned long long st_prev = wall_clock64();
#endif
  const float inv_on = 1.0f / np_.obs_noise;
  const float inv_sp2 = 1.0f / (np_.sig_param * np_.sig_param);
  const int nrt = g.np >> 4;
  const bool fastg = layout == 0 && (S & 1) == 0 && (uint64_t)S * dd < 0xFFFFFFFFull;  // (block-uniform)
  // which of the lane's output elements are observations that count (not padding, not intervened on): bit tj*4 + r of okb[u]
  uint32_t okb[NU];
#pragma unroll
  for (int u = 0; u < NU; ++u) {
    okb[u] = 0u;
#pragma unroll
    for (int tj = 0; tj < 7; ++tj)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = (wave + NW * u) * 16 + (lane >> 4) * 4 + r, j = tj * 16 + (lane & 15);
        const bool v = tj < NT && n < N && j < d && wave + NW * u < nrt && !(any_mask && mask[(size_t)n * d + j]);
        okb[u] |= (uint32_t)v << (tj * 4 + r);
      }
  }
  const unsigned int n_items = gp.ctr[0];
  bool x_ok = false;  // (block-uniform: x is in LDS -- the transposition at the end of a theta item borrows its storage)
  for (;;) {
  __syncthreads();
  if (tid == 0) cur_item = atomicAdd(gp.ctr + 1, 1u);
  __syncthreads();
  const unsigned int item_i = cur_item;
  if (item_i >= n_items) break;
  const unsigned int item = gp.items[item_i];
  const int m = (int)(item >> 6), bz = (int)(item & 63u);
  const double mx = gp.stats[(size_t)m * 4 + 0], den = gp.stats[(size_t)m * 4 + 1], sm = gp.stats[(size_t)m * 4 + 2];
  const int nnz = (int)gp.stats[(size_t)m * 4 + 3];
  const int nact = nnz < NS ? (nnz > 0 ? nnz : 1) : NS;
  const float* th_m = theta + (size_t)m * P;
  float* const om_final = out + (size_t)m * out_stride;
  const Key2 key = lin_mode_key(mode, carry, M_global, m0 + m, layout);
  const float* lp = logprobs + (size_t)m * S;
  NN_ST(14);
  if (!x_ok) {
    for (int e = tid; e < g.np * g.ldx; e += NTHR) {
      const int n = e / g.ldx, c = e - n * g.ldx;
      X[e] = (n < N && c < d) ? x[(size_t)n * d + c] : 0.f;
    }
    x_ok = true;
  }
  NN_ST(15);
  // Where things accumulate.  theta mode: the first-layer gradient always in this block's partial row (thread layout, w1sz floats); the small
  // leaves b1 | W2 | b2 behind it when several blocks share the particle, else in the output itself.  Z modes: d*d values in W's layout, in
  // the partial row (shared) or the output.
  const bool split = nact > 1;
  const size_t n_out = mode == LIN_MODE_THETA ? P : dd;
  const size_t w1sz = (size_t)NT * NUDM * H * 4 * NTHR;  // slots ((tj NUDM + u) H + h) 4 + r, see the x^T dpre epilogue
  float* const prow = gs.part + ((size_t)m * NS + bz) * gs.stride;
  float* const om = mode == LIN_MODE_THETA ? (split ? prow + w1sz - off.b1 : om_final) : (split ? prow : om_final);
  if (mode == LIN_MODE_THETA) {
    for (size_t e = tid; e < w1sz; e += NTHR) prow[e] = 0.f;
    for (size_t e = off.b1 + tid; e < P; e += NTHR) om[e] = 0.f;
  } else {
    for (size_t e = tid; e < dd; e += NTHR) om[e] = 0.f;
  }
  NN_ST(6);
  const float* sc_m = scores + (size_t)m * dd;
  const uint32_t* thr_m = thr + (size_t)m * dd;
  const float* w1t_m = w1t + (size_t)m * H * dd;  // (W1T[h][a][j] and the prior table exist whenever this kernel runs: joint_nn_launch)
  const TfKeys tk = tf_keys(key);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the zeroed rows are in place before another wave adds to them (barriers below)
  NN_ST(5);

  // This block's samples of a chunk are listed first (wave 0, ballots) and the heavy loop runs over the list: a scan over all S samples
  // with the per-sample state live across it paid the loop's register spills on every skipped sample (1.8 us each: 230 us per block at
  // config 5, more than the gradient of a sample -- profiles/round6_nn_grad_phases.txt).
  int q = 0;  // weighted samples before this chunk (block-uniform)
  for (int s0 = 0; s0 < S; s0 += GRAD_WCH) {
    __syncthreads();
    if (tid < GRAD_WCH && s0 + tid < S) wch[tid] = (float)(exp((double)lp[s0 + tid] - mx) / den);
    __syncthreads();
    NN_ST(18);
    if (wave == 0) {
      int cnt = 0, qq = q;
      for (int i0 = 0; i0 < GRAD_WCH; i0 += 64) {
        const int i = i0 + lane;
        const float wv = s0 + i < S ? wch[i] : 0.f;
        const bool has = wv >= GRAD_W_MIN;
        const unsigned long long hb = __ballot(has), lt = (1ull << lane) - 1ull;
        const int ord = qq + __popcll(hb & lt);
        const bool mine = has && (ord % NS) == bz;
        const unsigned long long mb = __ballot(mine);
        if (mine) {  // (in place: position <= i, and the wave has read its 64 entries before it writes)
          const int p = cnt + __popcll(mb & lt);
          wch[p] = wv;
          lst[p] = (unsigned short)i;
        }
        cnt += __popcll(mb);
        qq += __popcll(hb);
      }
      if (lane == 0) {
        lst_n[0] = cnt;
        lst_n[1] = qq;
      }
    }
    __syncthreads();
    const int nmine = lst_n[0];
    q = lst_n[1];
    NN_ST(19);
  for (int k = 0; k < nmine; ++k) {
    const int s = s0 + lst[k];
    const float w = wch[k];
    __syncthreads();
    NN_ST(0);
    if (fastg) nn_grad_build_graph<true>(GS, mode, key, tk, nbits, s, S, thr_m, sc_m, alpha, tau, layout, tiny, d, tid, NTHR);
    else nn_grad_build_graph<false>(GS, mode, key, tk, nbits, s, S, thr_m, sc_m, alpha, tau, layout, tiny, d, tid, NTHR);
    __syncthreads();
    NN_ST(1);
#ifdef DIBS_NN_STAMPS
    if (tid == 0) atomicAdd(&g_nn_stamps[(mode & 3) * 32 + 12], 1ull);
#endif
    if (mode == LIN_MODE_Z_SCORE) {
      for (int e = tid; e < (int)dd; e += NTHR) om[e] += w * GS[e];
      continue;
    }
    for (int tj = 0; tj < NT; ++tj) {
      const int j = tj * 16 + jl;
      const bool jok = j < d;
      f32x4 hv[HC][NU];  // act(pre_h + b1) of the lane's elements, all hidden units of a group
      f32x4 macc[NU];
#pragma unroll
      for (int u = 0; u < NU; ++u) macc[u] = f32x4{0.f, 0.f, 0.f, 0.f};
      float w2v[HC];  // second-layer weights of the lane's node for the current group (requested ahead of the products)
      const float b2 = (np_.bias && jok) ? th_m[off.b2 + j] : 0.f;
      f32x4 zs[NUDM];  // Z estimator: sum_h W1 xtr_h of the lane's elements
#pragma unroll
      for (int u = 0; u < NUDM; ++u) zs[u] = f32x4{0.f, 0.f, 0.f, 0.f};
      // The bodies are instantiated per group size HN (generic lambdas; NN_HN_SWITCH picks one): every hidden-unit loop is unrolled over
      // exactly the units that exist, so no register is held for an absent one.
      // ---- forward for the group h0 .. h0+HN-1: slices, products, activations (into hv), mean ----
      auto fwd = [&](auto hnc, const int h0) {
        constexpr int HN = decltype(hnc)::value;
        __syncthreads();  // the slices' last readers (previous tile's x^T dpre, previous group) are done
        nn_grad_build_slices<8>(SL, GS, w1t_m, h0, HN, tj, g, tid, NTHR);
        float b1v[HN];
#pragma unroll
        for (int hc = 0; hc < HN; ++hc) {
          b1v[hc] = (np_.bias && jok) ? th_m[off.b1 + (size_t)j * H + h0 + hc] : 0.f;
          w2v[hc] = jok ? th_m[off.w2 + (size_t)j * H + h0 + hc] : 0.f;
        }
        __syncthreads();
        NN_ST(2);
        nn_grad_gemm_fwd<HN, NU, NW>(X, SL, g, lane, wave, hv);
#pragma unroll
        for (int hc = 0; hc < HN; ++hc)
#pragma unroll
          for (int u = 0; u < NU; ++u)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const float a_ = nn_act(ACT >= 0 ? ACT : np_.act, hv[hc][u][r] + b1v[hc]);
              hv[hc][u][r] = a_;
              macc[u][r] = fmaf(w2v[hc], a_, macc[u][r]);
            }
        NN_ST(3);
      };
      // ---- backward for the group: dpre slices, column sums, small leaves, x^T dpre, gradient terms ----
      auto bwd = [&](auto hnc, const int h0, const bool again) {
        constexpr int HN = decltype(hnc)::value;
        if (again) {  // (more hidden units than a group holds: this group's activations again)
          __syncthreads();
          nn_grad_build_slices<8>(SL, GS, w1t_m, h0, HN, tj, g, tid, NTHR);
          float b1v[HN];
#pragma unroll
          for (int hc = 0; hc < HN; ++hc) {
            b1v[hc] = (np_.bias && jok) ? th_m[off.b1 + (size_t)j * H + h0 + hc] : 0.f;
            w2v[hc] = jok ? th_m[off.w2 + (size_t)j * H + h0 + hc] : 0.f;
          }
          __syncthreads();
          nn_grad_gemm_fwd<HN, NU, NW>(X, SL, g, lane, wave, hv);
#pragma unroll
          for (int hc = 0; hc < HN; ++hc)
#pragma unroll
            for (int u = 0; u < NU; ++u)
#pragma unroll
              for (int r = 0; r < 4; ++r) hv[hc][u][r] = nn_act(ACT >= 0 ? ACT : np_.act, hv[hc][u][r] + b1v[hc]);
        }
        __syncthreads();  // every wave is done reading the T slices: their storage now takes dpre
        NN_ST(5);
#pragma unroll
        for (int hc = 0; hc < HN; ++hc) {
          float t1 = 0.f, t2 = 0.f;
#pragma unroll
          for (int u = 0; u < NU; ++u)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int n = (wave + NW * u) * 16 + (lane >> 4) * 4 + r;
              if (wave + NW * u < nrt) {
                const float a_ = hv[hc][u][r], dm = macc[u][r];
                // act' from the activation value alone (relu / leaky relu: sign(h) = sign(pre); tanh: 1 - h^2; sigmoid: h (1 - h))
                const float dp = dm * w2v[hc] * nn_dact(ACT >= 0 ? ACT : np_.act, a_, a_);
                SL[(hc * NN_SLR + n) * 16 + jl] = dp;
                t1 += dp;        // for d/db1
                t2 += dm * a_;   // for d/dW2
              }
            }
          t1 += __shfl_xor(t1, 16);
          t1 += __shfl_xor(t1, 32);
          t2 += __shfl_xor(t2, 16);
          t2 += __shfl_xor(t2, 32);
          if (lane < 16) {
            CS2[((hc * 2 + 0) * 8 + wave) * 16 + lane] = t2;
            CS2[((hc * 2 + 1) * 8 + wave) * 16 + lane] = t1;
          }
        }
        __syncthreads();
        NN_ST(7);
        if (mode == LIN_MODE_THETA) {  // small leaves: column sums by wave, added in wave order
          for (int i = tid; i < HN * 16; i += NTHR) {
            const int hc = i >> 4, j2 = tj * 16 + (i & 15);
            if (j2 < d) {
              float s2 = 0.f, s1 = 0.f;
              for (int wv_ = 0; wv_ < NW; ++wv_) {
                s2 += CS2[((hc * 2 + 0) * 8 + wv_) * 16 + (i & 15)];
                s1 += CS2[((hc * 2 + 1) * 8 + wv_) * 16 + (i & 15)];
              }
              nn_acc(om + off.w2 + (size_t)j2 * H + h0 + hc, w * s2);
              if (np_.bias) nn_acc(om + off.b1 + (size_t)j2 * H + h0 + hc, w * s1);
            }
          }
          if (h0 == 0 && np_.bias && tid >= NTHR - 16) {  // d/db2_j = sum_n dmean_nj
            const int c = tid - (NTHR - 16), j2 = tj * 16 + c;
            if (j2 < d) {
              float sb = 0.f;
              for (int wv_ = 0; wv_ < NW; ++wv_) sb += CS[wv_ * 16 + c];
              nn_acc(om + off.b2 + j2, w * sb);
            }
          }
        }
        NN_ST(8);
        // xtr_h[a][j] = sum_n x[n][a] dpre_h[n][j]  (d/dT_h), a-tiles ti = wave + NW u
#pragma unroll
        for (int u = 0; u < NUDM; ++u) {
          const int ti = wave + NW * u;
          if (ti >= NT) continue;
          // the first-layer weights of the lane's elements: requested before the product, used after it (one trip past the L2 per tile
          // instead of one per element).  32-bit element indices off the particle's base pointer.
          float gvr[4];
          f32x4 w1v[HN];
          const int a0 = ti * 16 + (lane >> 4) * 4;
          const uint32_t wi = (uint32_t)h0 * (uint32_t)dd + (uint32_t)a0 * (uint32_t)d + (uint32_t)j;                       // W1T[h0][a0][j]
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            gvr[r] = (a0 + r < d && jok) ? GS[(a0 + r) * d + j] : 0.f;
#pragma unroll
            for (int hc = 0; hc < HN; ++hc)
              w1v[hc][r] = gvr[r] == 0.f ? 0.f : w1t_m[wi + (uint32_t)hc * (uint32_t)dd + (uint32_t)(r * d)];
          }
          f32x4 t[HC];
          nn_grad_gemm_xtr<HN>(X, SL, g, lane, ti, t);
          NN_ST(9);
          // partial-row slot of element (tj, u, h, r): ((tj NUDM + u) H + h) 4 + r  (thread layout: slot * NTHR + tid)
          const uint32_t s0_ = ((uint32_t)(tj * NUDM + u) * (uint32_t)H + (uint32_t)h0) * 4u * NTHR + (uint32_t)tid;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            if (gvr[r] == 0.f) continue;  // no term (hard graphs late in a run: a few per cent of the entries are edges)
#pragma unroll
            for (int hc = 0; hc < HN; ++hc) {
              if (mode == LIN_MODE_THETA) nn_acc(prow + (s0_ + (uint32_t)((hc * 4 + r) * NTHR)), w * gvr[r] * (t[hc][r] - w1v[hc][r] * inv_sp2));
              else zs[u][r] = fmaf(w1v[hc][r], t[hc][r], zs[u][r]);
            }
          }
          NN_ST(10);
        }
      };
      for (int h0 = 0; h0 < H; h0 += hcs) {
        const int hn = H - h0 < hcs ? H - h0 : hcs;
        NN_HN_SWITCH(hn, (fwd(std::integral_constant<int, HN>{}, h0)));
      }
      // ---- dmean = (1 - mask) (x - mean) / obs_noise (registers, C layout; zero on padding rows / columns) ----
      {
        float t = 0.f;
#pragma unroll
        for (int u = 0; u < NU; ++u)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int n = (wave + NW * u) * 16 + (lane >> 4) * 4 + r;
            float dm = 0.f;
            if ((okb[u] >> (tj * 4 + r)) & 1u) dm = (X[n * g.ldx + j] - macc[u][r] - b2) * inv_on;
            macc[u][r] = dm;
            t += dm;
          }
        t += __shfl_xor(t, 16);
        t += __shfl_xor(t, 32);
        if (lane < 16) CS[wave * 16 + lane] = t;
      }
      NN_ST(4);
      for (int h0 = 0; h0 < H; h0 += hcs) {
        const int hn = H - h0 < hcs ? H - h0 : hcs;
        NN_HN_SWITCH(hn, (bwd(std::integral_constant<int, HN>{}, h0, H > hcs)));
      }
      if (mode == LIN_MODE_Z_REPARAM) {
#pragma unroll
        for (int u = 0; u < NUDM; ++u) {
          const int ti = wave + NW * u;
          if (ti >= NT) continue;
          float lnv[4];  // sum_h logN(W1[j][a][h]; 0, sig_p): the particle's prior table
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int a = ti * 16 + (lane >> 4) * 4 + r;
            const bool ok = a < d && jok && a != j;
            lnv[r] = ok ? ln_tab[(size_t)m * dd + (size_t)a * d + j] : 0.f;
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int a = ti * 16 + (lane >> 4) * 4 + r;
            if (a >= d || !jok || a == j) continue;
            const float gv = GS[a * d + j];
            nn_acc(om + a * d + j, w * (lnv[r] + zs[u][r]) * tau * alpha * gv * (1.0f - gv));
          }
        }
        NN_ST(11);
      }
    }
  }
  }
  NN_ST(20);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's adds have been performed
  NN_ST(16);
  __syncthreads();
  NN_ST(17);
  if (split) {
    // release the partial sums, count this block, and the LAST block of the particle adds the rows in block order into the output
    __threadfence();
    if (!grad_last_block(gs.ctr + m, nact, &last_flag)) continue;  // (next item)
    __threadfence();
  }
  const float* const base = gs.part + (size_t)m * NS * gs.stride;
  if (mode == LIN_MODE_THETA) {
    // first-layer gradient: thread layout -> theta's layout [j][a][h], as many column tiles at a time as LDS holds (X, GS, the slices are dead)
    const int tpc = (int)(lds_floats / ((size_t)16 * d * H));
    if (tpc >= 1) {
      x_ok = false;
      for (int tj0 = 0; tj0 < NT; tj0 += tpc) {
        const int tj1 = tj0 + tpc < NT ? tj0 + tpc : NT;
        for (int h = 0; h < H; ++h)
#pragma unroll
          for (int u = 0; u < NUDM; ++u) {
            const int ti = wave + NW * u;
            if (ti >= NT) continue;
            for (int tj = tj0; tj < tj1; ++tj)
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                const int a = ti * 16 + (lane >> 4) * 4 + r, j = tj * 16 + jl;
                if (a < d && j < d) {
                  const size_t slot = (size_t)(((tj * NUDM + u) * H + h) * 4 + r) * NTHR + tid;
                  smem[((size_t)(j - 16 * tj0) * d + a) * H + h] = grad_part_sum<GRAD_NS_NN, false>(base, gs.stride, slot, nact);
                }
              }
          }
        __syncthreads();
        const int nj = (d - 16 * tj0) < 16 * (tj1 - tj0) ? (d - 16 * tj0) : 16 * (tj1 - tj0);
        const size_t cnt = (size_t)nj * d * H, o0 = (size_t)16 * tj0 * d * H;
        for (size_t e = tid; e < cnt; e += NTHR) om_final[o0 + e] = smem[e];
        __syncthreads();
      }
    } else {
      for (int h = 0; h < H; ++h)
#pragma unroll
        for (int u = 0; u < NUDM; ++u) {
          const int ti = wave + NW * u;
          if (ti >= NT) continue;
          for (int tj = 0; tj < NT; ++tj)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int a = ti * 16 + (lane >> 4) * 4 + r, j = tj * 16 + jl;
              if (a < d && j < d) {
                const size_t slot = (size_t)(((tj * NUDM + u) * H + h) * 4 + r) * NTHR + tid;
                om_final[((size_t)j * d + a) * H + h] = grad_part_sum<GRAD_NS_NN, false>(base, gs.stride, slot, nact);
              }
            }
        }
    }
    if (split)
      for (size_t e = off.b1 + tid; e < P; e += NTHR)  // the small leaves
        om_final[e] = grad_part_sum<GRAD_NS_NN, false>(base, gs.stride, w1sz + (e - off.b1), nact);
  } else if (split) {
    for (size_t e = tid; e < n_out; e += NTHR) om_final[e] = grad_part_sum<GRAD_NS_NN, false>(base, gs.stride, e, nact);
  }
  __syncthreads();
  NN_ST(13);
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
  }  // (next item)
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
int dibs_cu_count();                                     // (engine.hip: compute units of the current device)

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
          hipLaunchKernelGGL(k_nn_w1_exp, dim3(jl.Mloc), dim3(1024), 0, jl.stream, jl.theta, P, w->nhf_ew, jl.d, np_.H);
          dibs_allow_lds((const void*)k_nn_tables_hx, (size_t)256 * np_.H * 4);
          hipLaunchKernelGGL(k_nn_tables_hx, dim3((jl.d + 15) / 16, (jl.d + 15) / 16, jl.Mloc), dim3(256), (size_t)256 * np_.H * 4, jl.stream, jl.theta, P,
                             w->nhf_ew, (float4*)w->nhx_w1s, (uint4*)w->nhx_w1p, jl.d, np_.H);
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
      hipLaunchKernelGGL(k_nn_w1_exp, dim3(jl.Mloc), dim3(1024), 0, jl.stream, jl.theta, P, w->nhf_ew, jl.d, np_.H);
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
  const size_t lds1 = nn_lds_bytes_logprobs(jl.d, jl.N, NT, np_.H);
  float* lp = mode == LIN_MODE_THETA ? jl.logprobs_th : jl.logprobs_z;
  const size_t w1t_need = (size_t)jl.Mloc * np_.H * jl.d * jl.d;
  if (w->w1t_floats < w1t_need) {  // (first launch)
    if (w->w1t) hipFree(w->w1t);
    w->w1t = nullptr;
    w->w1t_floats = hipMalloc((void**)&w->w1t, w1t_need * 4) == hipSuccess ? w1t_need : 0;
    if (!w->w1t_floats) w->w1t = nullptr;
  }
  if (!w->w1t || !w->ln_tab) return;  // (k_nn_grad reads both; the step's launch check reports the failed hipMalloc)
  if (mode == LIN_MODE_THETA && w->ln_tab)  // theta is the same for both estimators of a step: the tables are built once (theta runs first)
    dibs_allow_lds((const void*)k_nn_prior_table, (size_t)256 * np_.H * 4);
    hipLaunchKernelGGL(k_nn_prior_table, dim3((jl.d + 15) / 16, (jl.d + 15) / 16, jl.Mloc), dim3(256), (size_t)256 * np_.H * 4, jl.stream, jl.theta, P,
                       w->ln_tab, w->w1t, jl.d, np_.H, np_.sig_param);
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
  GradSplit gs;  // (several blocks per particle once many samples keep a non-zero weight: kernels_joint.h)
  // (partial row of a block: the first-layer gradient in thread layout + the small leaves for theta, d*d for Z; the shares per particle are
  //  cut back when GRAD_NS_NN rows per particle would exceed 4 GiB -- hidden widths in the dozens)
  const int hcs = nn_grad_hcs(jl.d, jl.N, NT, np_.H);  // (> 0: joint_nn_fast_path)
  const size_t lds2 = nn_grad_lds_bytes(jl.d, jl.N, NT, hcs);
  const bool wide = lds2 > 80 * 1024;  // one block per CU: 8 waves (see k_nn_grad)
  const int nthr = wide ? 512 : 256, nudm = wide ? 1 : 2;  // (k_nn_grad: NTHR, NUDM)
  const size_t row_theta = (size_t)NT * nudm * np_.H * 4 * nthr + (P - (size_t)jl.d * jl.d * np_.H);
  const size_t row = row_theta > (size_t)jl.d * jl.d ? row_theta : (size_t)jl.d * jl.d;
  int ns_nn = GRAD_NS_NN;
  while (ns_nn > 1 && (size_t)jl.Mloc * ns_nn * row * 4 > ((size_t)4 << 30)) ns_nn >>= 1;
  if (!joint_grad_split(w, (size_t)jl.Mloc, row, &gs, ns_nn)) return;  // (the step's launch check reports the failed hipMalloc)
  GradPlan gp;
  if (!joint_grad_plan(w, (size_t)jl.Mloc, ns_nn, &gp)) return;
  hipLaunchKernelGGL(k_grad_plan, dim3(jl.Mloc), dim3(64), 0, jl.stream, lp, jl.S, ns_nn, gp);
  // persistent blocks: as many as are resident at once (one per CU when a block fills the LDS, else two), at most one per item
  const long max_items = (long)jl.Mloc * ns_nn;
  const int resident = dibs_cu_count() * (wide ? 1 : 2);
  const int nblk = (int)(max_items < resident ? max_items : resident);
#define NN_GRAD_LAUNCH(ACT_, NW_)                                                                                                              \
  {                                                                                                                                            \
    if (lds2 > 48 * 1024) hipFuncSetAttribute((const void*)k_nn_grad<ACT_, NW_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2);       \
    hipLaunchKernelGGL((k_nn_grad<ACT_, NW_>), dim3(nblk), dim3(64 * NW_), lds2, jl.stream, w->x, w->mask, jl.theta, P, jl.scores,               \
                       jl.thr, lp, out, ostride, tcopy, jl.baseline, mode == LIN_MODE_THETA ? (float*)nullptr : jl.baseline_out, carry, mode,    \
                       jl.m0, jl.M, jl.d, jl.N, jl.S, jl.alpha, jl.tau, jl.layout, jl.tiny, np_, jl.sf_baseline, w->any_mask, gs,                \
                       w->w1t, w->ln_tab, NT, hcs, gp, ns_nn);                                                                               \
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
  hipMemcpyFromSymbol(out, HIP_SYMBOL(g_nn_stamps), sizeof(unsigned long long) * 128);
  if (reset) {
    unsigned long long z[128] = {0};
    hipMemcpyToSymbol(HIP_SYMBOL(g_nn_stamps), z, sizeof(z));
  }
}
#endif
bool joint_nn_fast_path(int d, int N, const NNParams& np_) {
  return np_.n_hidden == 1 && np_.H >= 1 && np_.H <= 64 && N <= 128 && d <= 112 && nn_grad_hcs(d, N, (d + 15) / 16, np_.H) > 0 &&
         nn_lds_bytes_logprobs(d, N, (d + 15) / 16, np_.H) <= (size_t)160 * 1024 - 512;
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
