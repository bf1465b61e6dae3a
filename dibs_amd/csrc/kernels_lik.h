// score-function estimator: softmax weights over the sampled graphs (device function shared by k_lik_weights_score and the
// ride-along blocks of k_acyc)
#pragma once
#include "common.h"

// ------------------------------------------------------------------------------------------------
// K4  likelihood weights of the score-function estimator: l_s = sum_j node score, w = softmax(l),
//     W_lik = scale * alpha * (sum_s w_s G_s - P) off-diagonal; baseline EMA.
//     reference: dibs.py:359-389 (closed form of the signed-logsumexp ratio)
// grid = Mloc, block = 256; dynamic LDS = S*d*W*8 + S*8 + S*4
// ------------------------------------------------------------------------------------------------
struct LikArgs {
  const double* node_scores;
  const uint64_t* masks;
  const float* probs;
  float* logprobs;
  float* w_lik;
  const float* baseline;
  float* baseline_out;
  float alpha;
  double sf_baseline;
  int d, S, W, masks_in_lds, ny;
  unsigned int* queue_counts;
};
// body of one block (m, y of ny).  Also called from the k_acyc launch when that launch leaves block slots free (a rank with
// few particles): these latency-bound blocks then hide behind the acyclicity blocks -- the k_bge_big launch they depend
// on precedes both in stream order.
__device__ __forceinline__ void lik_weights_block(unsigned char* smem_raw, const LikArgs& A, int m, int y) {
  const double* __restrict__ node_scores = A.node_scores;
  const uint64_t* __restrict__ masks = A.masks;
  const float* __restrict__ probs = A.probs;
  float* __restrict__ logprobs = A.logprobs;
  float* __restrict__ w_lik = A.w_lik;
  const float* __restrict__ baseline = A.baseline;
  float* __restrict__ baseline_out = A.baseline_out;
  const float alpha = A.alpha;
  const double sf_baseline = A.sf_baseline;
  const int d = A.d, S = A.S, W = A.W, masks_in_lds = A.masks_in_lds, ny = A.ny;
  // the BGe queues of this step have been consumed (stream order): reset their counters for the next step
  if (A.queue_counts && m == 0 && y == 0 && threadIdx.x < 16) A.queue_counts[threadIdx.x] = 0u;  // (16 counters are allocated; BGE_NQ used)
  // block (m, y) handles the columns j = y, y + ny, ... of particle m; every block recomputes l_s / softmax
  double* lp = reinterpret_cast<double*>(smem_raw);
  double* lp2 = lp + S;  // [2][S] partial sums
  float* wt = reinterpret_cast<float*>(lp2 + 2 * S);
  float* nzw = wt + S;                          // non-zero softmax weights, in sample order ...
  int* nzi = reinterpret_cast<int*>(nzw + S);   // ... and their sample indices
  uint64_t* mkl = reinterpret_cast<uint64_t*>(smem_raw + (((size_t)S * 36 + 15) & ~(size_t)15));
  __shared__ double red[8];
  __shared__ int nnz_s;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ncol = (d - y + ny - 1) / ny;  // columns of this block
  const uint64_t* mg = masks + (size_t)m * d * S * W;  // [j][s][w]
  if (masks_in_lds)
    for (int e = tid; e < ncol * S * W; e += 256) {
      const int c = e / (S * W), rest = e - c * (S * W);
      mkl[e] = mg[(size_t)(y + c * ny) * S * W + rest];
    }
  {
    // l_s = sum_j node score: two threads per sample when they fit, loads batched eight deep; the partial sums are
    // combined in a fixed order (deterministic)
    const int nsplit = (2 * S <= 256) ? 2 : 1;
    const int jw = (d + nsplit - 1) / nsplit;
    const double* nsm = node_scores + (size_t)m * d * S;
    for (int idx = tid; idx < nsplit * S; idx += 256) {
      const int part = idx / S, s = idx - part * S;
      const int j0 = part * jw, j1 = (j0 + jw < d) ? j0 + jw : d;
      double t = 0.0;
      int j = j0;
      for (; j + 32 <= j1; j += 32) {  // (far-cache latency: as many loads in flight as registers allow; additions in j order)
        double v[32];
#pragma unroll
        for (int u = 0; u < 32; ++u) v[u] = nsm[(size_t)(j + u) * S + s];
#pragma unroll
        for (int u = 0; u < 32; ++u) t += v[u];
      }
      {
        double v[32];
#pragma unroll
        for (int u = 0; u < 32; ++u) v[u] = (j + u < j1) ? nsm[(size_t)(j + u) * S + s] : 0.0;
#pragma unroll
        for (int u = 0; u < 32; ++u) t += (j + u < j1) ? v[u] : 0.0;
      }
      lp2[part * S + s] = t;
    }
    __syncthreads();
    for (int s = tid; s < S; s += 256) {
      const double t = nsplit == 2 ? lp2[s] + lp2[S + s] : lp2[s];
      lp[s] = t;
      if (y == 0) logprobs[(size_t)m * S + s] = (float)t;
    }
  }
  __syncthreads();
  double mx = -INFINITY, sm = 0.0;
  for (int s = tid; s < S; s += 256) mx = lp[s] > mx ? lp[s] : mx;
  mx = wave_max_d(mx);
  if (lane == 0) red[wave] = mx;
  __syncthreads();
  mx = red[0];
  for (int w = 1; w < 4; ++w) mx = red[w] > mx ? red[w] : mx;
  double den = 0.0;
  for (int s = tid; s < S; s += 256) {
    den += exp(lp[s] - mx);
    sm += lp[s];
  }
  den = wave_sum_d(den);
  sm = wave_sum_d(sm);
  if (lane == 0) red[4 + wave] = den;
  __syncthreads();
  den = red[4] + red[5] + red[6] + red[7];
  __syncthreads();
  if (lane == 0) red[wave] = sm;
  for (int s = tid; s < S; s += 256) wt[s] = (float)(exp(lp[s] - mx) / den);
  __syncthreads();
  sm = red[0] + red[1] + red[2] + red[3];
  // in float most softmax weights are exactly 0 while the particles still differ (one-hot in the limit): only samples with
  // w_s != 0 are visited, in sample order, so the sum is bit-identical to the full loop
  if (wave == 0) {
    int base = 0;
    for (int s0 = 0; s0 < S; s0 += 64) {
      const int s = s0 + lane;
      const float w = s < S ? wt[s] : 0.f;
      const unsigned long long bal = __ballot(w != 0.f);
      if (w != 0.f) {
        const int pos = base + __popcll(bal & ((1ull << lane) - 1ull));
        nzi[pos] = s;
        nzw[pos] = w;
      }
      base += __popcll(bal);
    }
    if (lane == 0) nnz_s = base;
  }
  __syncthreads();
  const int nnz = nnz_s;
  const float bold = baseline[m];
  const float scale = sf_baseline > 0.0 ? (float)exp(-(double)bold) : 1.0f;
  for (int e = tid; e < ncol * d; e += 256) {
    const int c = e / d, i = e - c * d, j = y + c * ny;
    float out = 0.f;
    if (i != j) {
      float acc = 0.f;
      const int w = i >> 6;
      const uint64_t bit = 1ull << (i & 63);
      const uint64_t* col = masks_in_lds ? mkl + (size_t)c * S * W : mg + (size_t)j * S * W;
      int q = 0;
      for (; q + 8 <= nnz; q += 8) {  // eight mask words in flight; additions stay in sample order
        uint64_t mw[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) mw[u] = col[(size_t)nzi[q + u] * W + w];
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += (mw[u] & bit) ? nzw[q + u] : 0.f;
      }
      for (; q < nnz; ++q) acc += (col[(size_t)nzi[q] * W + w] & bit) ? nzw[q] : 0.f;
      out = scale * alpha * (acc - probs[(size_t)m * d * d + i * d + j]);
    }
    w_lik[(size_t)m * d * d + i * d + j] = out;
  }
  if (tid == 0 && y == 0) baseline_out[m] = (float)(sf_baseline * (sm / S) + (1.0 - sf_baseline) * (double)bold);
}

#ifdef DIBS_TU_ENGINE
__global__ __launch_bounds__(256) void k_lik_weights_score(LikArgs A) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  lik_weights_block(smem_raw, A, blockIdx.x, blockIdx.y);
}
#endif
