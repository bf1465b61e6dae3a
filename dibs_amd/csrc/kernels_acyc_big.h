// acyclicity-constraint gradient for n_vars > 112: the matrices no longer fit in LDS, the matrix powers go through global memory (gfx950)
#pragma once
#include "common.h"

// ------------------------------------------------------------------------------------------------
// K5c  same computation as k_acyc (kernels_acyc.h; reference: graph_utils.py:8-28, dibs.py:121-140, 557-601) for any size:
//   k_acycb_init   M_c = I + G~_c / d for every chain c = (particle m, Monte-Carlo sample sa), [n_chains][dp][dp] f32, zero padded to dp =
//                  ceil16(d); the Gumbel-soft graph is drawn as in k_acyc (dibs.py:595: the particle key used directly)
//   k_bgemm        batched C_c = A_c B_c on v_mfma_f32_16x16x4_f32, 64 x 64 tiles through LDS -- left-to-right binary powering of d - 1 as
//                  jnp.linalg.matrix_power does, driven from the host (acyc_big_launch)
//   k_acycb_out    w_acyc[m][a][b] = 1/Sa sum_sa (M_c^{d-1})[b][a] * tau alpha g (1 - g), chains in ascending order; g is drawn again
//                  (same stream, same formula) rather than stored
// The fallback behind the constructor's full range, not tuned: three [n_chains][dp][dp] buffers live in HBM (2 GB at d = 200 with 128
// particles and 32 chains) and every product is a round trip through the caches.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float acycb_soft_graph(Key2 km, uint64_t nbits, uint64_t idx, float as, float tau, int layout, int tiny) {
  const uint32_t y = rng_bits_at(km, nbits, idx, layout);
  if (tau == 1.0f) {  // sigmoid(eps + a) with eps = log(u / (1 - u))  ==  u / (u + (1 - u) exp(-a)); saturated edges give exactly 1 (k_acyc)
    const float ulo = tiny ? 1.17549435e-38f : 1.1920929e-07f;
    const float u = rng_uniform(y, ulo, 1.0f), ea = expf(-as);
    const float den = fmaf(1.0f - u, ea, u);
    return den == u ? 1.0f : u * __builtin_amdgcn_rcpf(den);
  }
  return 1.0f / (1.0f + expf(-tau * (rng_logistic(y, tiny) + as)));
}

// grid = (ceil(dp * dp / 256), n_chains_local), block = 256
__global__ __launch_bounds__(256) void k_acycb_init(const float* __restrict__ scores, float* __restrict__ Mbuf, Key2 carry, int m0, int M_global,
                                                    int d, int dp, int Sa, float alpha, float tau, int layout, int tiny) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= dp * dp) return;
  const int c = blockIdx.y, m = c / Sa, sa = c - m * Sa;
  const int a = e / dp, b = e - a * dp;
  float v = 0.f;
  if (a < d && b < d) {
    v = 1.0f;
    if (a != b) {
      const Key2 km = rng_split_row(carry, (uint32_t)M_global + 1u, (uint32_t)(m0 + m) + 1u, layout);
      const uint64_t dd = (uint64_t)d * d;
      const float g = acycb_soft_graph(km, (uint64_t)Sa * dd, (uint64_t)sa * dd + (uint64_t)(a * d + b), alpha * scores[(size_t)m * dd + a * d + b],
                                       tau, layout, tiny);
      v = g * (1.0f / (float)d);
    }
  }
  Mbuf[(size_t)c * dp * dp + e] = v;
}

// C_c = A_c B_c; grid = (ceil(dp / 64), ceil(dp / 64), n_chains), block = 256: wave w owns rows 16 w .. 16 w + 15 of the 64 x 64 tile
// MFMA operands: A[row = lane % 16][k = lane / 16], B[k = lane / 16][col = lane % 16], D[row = 4 (lane / 16) + r][col = lane % 16]
#define BGEMM_LDA 20
#define BGEMM_LDB 68
__global__ __launch_bounds__(256) void k_bgemm(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C, int dp) {
  __shared__ __attribute__((aligned(16))) float As[64 * BGEMM_LDA];
  __shared__ __attribute__((aligned(16))) float Bs[16 * BGEMM_LDB];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, r = lane & 15;
  const size_t mat = (size_t)blockIdx.z * dp * dp;
  const int row0 = blockIdx.y * 64, col0 = blockIdx.x * 64;
  const float* Ac = A + mat;
  const float* Bc = B + mat;
  f32x4 acc[4];
#pragma unroll
  for (int ct = 0; ct < 4; ++ct) acc[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int ar = tid >> 2, ak = (tid & 3) * 4;    // A tile: row ar, k ak .. ak + 3
  const int bk = tid >> 4, bc = (tid & 15) * 4;   // B tile: k bk, columns bc .. bc + 3
  for (int k0 = 0; k0 < dp; k0 += 16) {
    float4 av = make_float4(0.f, 0.f, 0.f, 0.f), bv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (row0 + ar < dp) av = *reinterpret_cast<const float4*>(Ac + (size_t)(row0 + ar) * dp + k0 + ak);   // (dp, k0, ak multiples of 4: aligned)
    if (col0 + bc < dp) bv = *reinterpret_cast<const float4*>(Bc + (size_t)(k0 + bk) * dp + col0 + bc);
    __syncthreads();
    *reinterpret_cast<float4*>(As + ar * BGEMM_LDA + ak) = av;
    *reinterpret_cast<float4*>(Bs + bk * BGEMM_LDB + bc) = bv;
    __syncthreads();
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const float a = As[(16 * wave + r) * BGEMM_LDA + 4 * ks + g];
#pragma unroll
      for (int ct = 0; ct < 4; ++ct) acc[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, Bs[(4 * ks + g) * BGEMM_LDB + 16 * ct + r], acc[ct], 0, 0, 0);
    }
  }
  float* Cc = C + mat;
#pragma unroll
  for (int ct = 0; ct < 4; ++ct)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = row0 + 16 * wave + 4 * g + i, col = col0 + 16 * ct + r;
      if (row < dp && col < dp) Cc[(size_t)row * dp + col] = acc[ct][i];
    }
}

// grid = (ceil(d * d / 256), Mloc), block = 256; thread <-> element (a fastest: the transposed reads of the powers are coalesced)
__global__ __launch_bounds__(256) void k_acycb_out(const float* __restrict__ scores, const float* __restrict__ Pbuf, float* __restrict__ w_acyc,
                                                   Key2 carry, int m0, int M_global, int d, int dp, int Sa, float alpha, float tau, int layout,
                                                   int tiny) {
  const int e = blockIdx.x * 256 + threadIdx.x, m = blockIdx.y;
  if (e >= d * d) return;
  const int b = e / d, a = e - b * d;
  float acc = 0.f;
  if (a != b) {
    const Key2 km = rng_split_row(carry, (uint32_t)M_global + 1u, (uint32_t)(m0 + m) + 1u, layout);
    const uint64_t dd = (uint64_t)d * d;
    const float as = alpha * scores[(size_t)m * dd + a * d + b], ta = tau * alpha;
    for (int sa = 0; sa < Sa; ++sa) {
      const float g = acycb_soft_graph(km, (uint64_t)Sa * dd, (uint64_t)sa * dd + (uint64_t)(a * d + b), as, tau, layout, tiny);
      acc += Pbuf[((size_t)m * Sa + sa) * dp * dp + (size_t)b * dp + a] * (ta * g * (1.0f - g));
    }
  }
  w_acyc[(size_t)m * d * d + (size_t)a * d + b] = acc * (1.0f / (float)Sa);
}
