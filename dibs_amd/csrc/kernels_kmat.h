// latent / parameter kernel matrix slab (device function shared by k_kmat and the ride-along blocks of k_bge_sample)
#pragma once
#include "common.h"

// The latent kernel matrix only needs z, which is final when a step starts.  On a single rank its blocks ride along in the
// k_bge_sample launch (extra blockIdx.x range): that kernel is bound by VALU issue, k_kmat by the latency of the far cache
// levels, so the two overlap almost for free.  (Several ranks: the rows of the other ranks arrive with the all-gather, the
// kernel matrix stays in phase B.)
#define KMAT_BT 16
struct KmatFuse {
  const float* z;   // [M, len] (null: nothing fused)
  float* kout;      // [M, M]
  int len, M, nbx;  // nbx: first blockIdx.x of the kernel-matrix range
  float scale, h;
};

// ------------------------------------------------------------------------------------------------
// K8a kernel matrix slab: kz[a, b] = scale * exp(-||z_a - z_b||^2 / h) for local a, all b (direct differences:
//     the entries are ~e^-40 at d = 50 and must not be flushed or computed by cancellation).
//     reference: kernel.py:20-30 / 52-71, svgd.py:165-176 / 537-551
// grid = Mloc, block = 256; dynamic LDS = len * 4
// ------------------------------------------------------------------------------------------------
#define KMAT_CH 32768  // floats of z_a staged in LDS at a time (128 KiB); longer vectors (DenseNN theta at d = 100) go in chunks
__device__ __forceinline__ void kmat_block(float* __restrict__ smem, const float* __restrict__ pack, size_t pack_stride,
                                           size_t seg_off, int len, float* __restrict__ kout, int m0, int M, float scale, float h,
                                           int symmetric, int a, int bt, const float* __restrict__ kadd = nullptr,
                                           float* __restrict__ ksum = nullptr) {
  // (ksum != null: also writes kadd[a][b] + k[a][b] -- the joint models' weight matrix kz + kt, formed once here instead of per use in the
  //  SVGD transform, where it can then be a scalar operand; kadd is the matrix of an EARLIER launch)
  // block (a, bt): particle a (local) against b = bt * KMAT_BT .. +KMAT_BT-1; wave w takes b = b0 + w, b0 + w + 4, ...
  // symmetric (one rank holds all particles): tiles below the diagonal are skipped and k[a][b] is mirrored into k[b][a]
  // -- the sum of squared differences is the same number either way, so the slab is bit-identical to the full computation.
  const int b0 = bt * KMAT_BT, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (symmetric && b0 + KMAT_BT - 1 < a) return;
  const float* za = pack + (size_t)(m0 + a) * pack_stride + seg_off;
  double acc[KMAT_BT / 4];
#pragma unroll
  for (int q = 0; q < KMAT_BT / 4; ++q) acc[q] = 0.0;
  for (int c0 = 0; c0 < len; c0 += KMAT_CH) {
    const int clen = len - c0 < KMAT_CH ? len - c0 : KMAT_CH;
    const int len4 = clen >> 2;
    if (c0) __syncthreads();
    for (int e0 = 0; e0 < len4; e0 += 8 * 256) {  // (loads requested together: a rolled load -> LDS store loop makes one trip per element)
      float4 t[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int e = e0 + u * 256 + tid;
        t[u] = reinterpret_cast<const float4*>(za + c0)[e < len4 ? e : len4 - 1];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int e = e0 + u * 256 + tid;
        if (e < len4) reinterpret_cast<float4*>(smem)[e] = t[u];
      }
    }
    for (int e = (len4 << 2) + tid; e < clen; e += 256) smem[e] = za[c0 + e];
    __syncthreads();
    // the wave's four b rows advance together: 8 independent 16-byte loads in flight per lane and pass (the kernel is
    // bound by the latency of the far cache levels, not by their bandwidth)
    const float4* zb4[KMAT_BT / 4];
    const float* zbs[KMAT_BT / 4];
#pragma unroll
    for (int q = 0; q < KMAT_BT / 4; ++q) {
      const int b = b0 + wave + 4 * q;
      zbs[q] = pack + (size_t)(b < M ? b : M - 1) * pack_stride + seg_off + c0;  // rows past the end repeat the last one
      zb4[q] = reinterpret_cast<const float4*>(zbs[q]);
    }
    const float4* za4 = reinterpret_cast<const float4*>(smem);
    float s0[KMAT_BT / 4], s1[KMAT_BT / 4];
#pragma unroll
    for (int q = 0; q < KMAT_BT / 4; ++q) s0[q] = s1[q] = 0.f;
    int e = lane;
    for (; e + 64 < len4; e += 128) {
      float4 qa[KMAT_BT / 4], qb[KMAT_BT / 4];
#pragma unroll
      for (int q = 0; q < KMAT_BT / 4; ++q) {
        qa[q] = zb4[q][e];
        qb[q] = zb4[q][e + 64];
      }
      const float4 pa = za4[e], pb = za4[e + 64];
#pragma unroll
      for (int q = 0; q < KMAT_BT / 4; ++q) {
        float t;
        t = pa.x - qa[q].x; s0[q] = fmaf(t, t, s0[q]); t = pa.y - qa[q].y; s0[q] = fmaf(t, t, s0[q]);
        t = pa.z - qa[q].z; s0[q] = fmaf(t, t, s0[q]); t = pa.w - qa[q].w; s0[q] = fmaf(t, t, s0[q]);
        t = pb.x - qb[q].x; s1[q] = fmaf(t, t, s1[q]); t = pb.y - qb[q].y; s1[q] = fmaf(t, t, s1[q]);
        t = pb.z - qb[q].z; s1[q] = fmaf(t, t, s1[q]); t = pb.w - qb[q].w; s1[q] = fmaf(t, t, s1[q]);
      }
    }
    for (; e < len4; e += 64) {
      const float4 pa = za4[e];
#pragma unroll
      for (int q = 0; q < KMAT_BT / 4; ++q) {
        const float4 qa = zb4[q][e];
        float t;
        t = pa.x - qa.x; s0[q] = fmaf(t, t, s0[q]); t = pa.y - qa.y; s0[q] = fmaf(t, t, s0[q]);
        t = pa.z - qa.z; s0[q] = fmaf(t, t, s0[q]); t = pa.w - qa.w; s0[q] = fmaf(t, t, s0[q]);
      }
    }
    for (int e1 = (len4 << 2) + lane; e1 < clen; e1 += 64) {
#pragma unroll
      for (int q = 0; q < KMAT_BT / 4; ++q) {
        const float t = smem[e1] - zbs[q][e1];
        s1[q] = fmaf(t, t, s1[q]);
      }
    }
#pragma unroll
    for (int q = 0; q < KMAT_BT / 4; ++q) acc[q] += (double)s0[q] + (double)s1[q];
  }
#pragma unroll
  for (int q = 0; q < KMAT_BT / 4; ++q) {
    const int b = b0 + wave + 4 * q;
    if (b >= M) continue;
    const double tot = wave_sum_d(acc[q]);
    if (lane == 0) {
      const float kv = (float)((double)scale * exp(-tot / (double)h));
      kout[(size_t)a * M + b] = kv;
      if (symmetric && b > a) kout[(size_t)b * M + a] = kv;
      if (ksum) {
        const float sv = kadd[(size_t)a * M + b] + kv;
        ksum[(size_t)a * M + b] = sv;
        if (symmetric && b > a) ksum[(size_t)b * M + a] = sv;
      }
    }
  }
}

