// HIP kernels (gfx950) for the MarginalDiBS + BGe SVGD step.  One step = the launches listed in
// engine.hip::step_local / step_update; DESIGN.md has the per-kernel roofline and byte counts.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "rng.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double wave_max_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const double t = __shfl_xor(v, o, 64);
    v = t > v ? t : v;
  }
  return v;
}

__device__ __forceinline__ double sigmoid_d(double v) { return 1.0 / (1.0 + exp(-v)); }

// ------------------------------------------------------------------------------------------------
// particle init: z = normal(subk, (M, d, k, 2)) * std          svgd.py:145-146 / 509-510
// ------------------------------------------------------------------------------------------------
__global__ void k_init_z(float* __restrict__ z, Key2 key, uint64_t n_total, uint64_t offset, uint64_t n_local, float stdv,
                         int layout) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_local) return;
  z[i] = rng_normal(rng_bits_at(key, n_total, offset + i, layout)) * stdv;
}

// theta = mean + sig * normal ; theta += sign(theta) * min_edge      linearGaussian.py:212-227
__global__ void k_init_theta_lin(float* __restrict__ th, Key2 key, uint64_t n_total, uint64_t offset, uint64_t n_local,
                                 float mean_edge, float sig_edge, float min_edge, int layout) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_local) return;
  const float v = mean_edge + sig_edge * rng_normal(rng_bits_at(key, n_total, offset + i, layout));
  const float sg = v > 0.f ? 1.f : (v < 0.f ? -1.f : 0.f);
  th[i] = v + sg * min_edge;
}

// ------------------------------------------------------------------------------------------------
// K1  edge scores: scores[m] = U V^T via v_mfma_f32_16x16x4_f32 (k-ordered exact-f32 fma chain),
//     thresholds thr = ceil(sigmoid(alpha * s) * 2^23) for Bernoulli(p) == ((bits >> 9) < thr)
//     reference: dibs.py:168-184 (edge_probs), dibs.py:115 (bernoulli)
// grid = Mloc, block = 256; dynamic LDS = 2 * dpad * ldk * 4
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_edge_scores(const float* __restrict__ z, float* __restrict__ scores,
                                                     uint32_t* __restrict__ thr, float* __restrict__ probs, float alpha,
                                                     int d, int k, int dpad, int ldk) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Us = smem;
  float* Vs = smem + (size_t)dpad * ldk;
  const int m = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float2* zm = reinterpret_cast<const float2*>(z + (size_t)m * d * k * 2);
  for (int e = tid; e < dpad * ldk; e += 256) {
    const int i = e / ldk, q = e - i * ldk;
    float2 uv = make_float2(0.f, 0.f);
    if (i < d && q < k) uv = zm[(size_t)i * k + q];
    Us[e] = uv.x;
    Vs[e] = uv.y;
  }
  __syncthreads();
  const int nt = dpad >> 4;
  const int kp = (k + 3) & ~3;
  // tiles are dealt out over the waves of the gridDim.y blocks of this particle (the epilogue's double-precision sigmoid
  // is most of the work: more blocks, shorter critical path)
  for (int t = blockIdx.y * 4 + wave; t < nt * nt; t += 4 * gridDim.y) {
    const int ti = t / nt, tj = t - ti * nt;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const float* ua = Us + (size_t)(ti * 16 + (lane & 15)) * ldk + (lane >> 4);
    const float* vb = Vs + (size_t)(tj * 16 + (lane & 15)) * ldk + (lane >> 4);
    for (int k0 = 0; k0 < kp; k0 += 4) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ua[k0], vb[k0], acc, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = ti * 16 + (lane >> 4) * 4 + r, col = tj * 16 + (lane & 15);
      if (row < d && col < d) {
        const float s = acc[r];
        const size_t o = ((size_t)m * d + row) * d + col;
        scores[o] = s;
        const float pf = (float)sigmoid_d((double)__fmul_rn(alpha, s));
        thr[o] = row == col ? 0u : (uint32_t)ceilf(pf * 8388608.0f);
        probs[o] = row == col ? 0.f : pf;  // edge_probs (dibs.py:168-184), reused by the prior / estimator kernels
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// K2+K3  BGe node scores.  Wave w of block (jb, m) owns node j = jb * WAVES + w of particle m:
//   1. samples column j of all S graphs (Threefry; sample s and s + S/2 share one counter pair in the legacy
//      layout, so both outputs of every call are used) and stores the parent sets (bit masks);
//   2. per sample factors A = R[pa u {j}] with j ordered last (Cholesky): logdet R[pa,pa] = sum_{r<l} log d_r,
//      Schur complement of j = d_l, which give the two masked slogdets of the reference.  Three tiers by n = l+1:
//        n <= 8   one problem per lane, 8x8 Cholesky entirely in registers;
//        n <= 32  two problems per wave (32-lane groups), lane = row, own row of L in registers, pivot rows
//                 broadcast from LDS with ds_read_b128;
//        else     one problem per wave, L in LDS (generic).
//   reference: dibs.py:102-119 (sample_g), linearGaussian.py:63-118, func.py:128-145
// grid = (ceil(d / WAVES), Mloc), block = 64 * WAVES; WAVES = 4 shares one R in LDS (no interventions),
// WAVES = 1 loads R_j per block.  dynamic LDS: bge_lds_bytes()
// layouts: masks [Mloc][d][S][W] u64, node_scores [Mloc][d][S] f64
// ------------------------------------------------------------------------------------------------
struct BgeParams {
  const float* R;       // [n_mats, d, d]
  const double* gam;    // [d, d+1]  log_gamma_term(j, l)
  const double* Nj;     // [d]
  double alpha_lambd;
  int n_mats;
};

__host__ __device__ inline int bge_lbuf_floats(int d) {
  const int generic = d > 32 ? d * (d | 1) : 0;
  const int g32 = 2 * 32 * 36;  // >= 4 * 16 * 20
  return ((generic > g32 ? generic : g32) + 3) & ~3;
}
__host__ __device__ inline int bge_idx_len(int d) { return d + 4 > 36 ? d + 4 : 36; }  // >= 32: the 32-lane tier zero-fills 32 slots
__host__ __device__ inline size_t bge_wave_bytes(int d, int S, int W) {
  // masks[S*W] u64 | thr[d] | lim[d] | loc[S]
  size_t b = (size_t)S * W * 8 + (size_t)2 * d * 4 + (size_t)S * 4;
  return (b + 15) & ~(size_t)15;
}
__host__ __device__ inline size_t bge_lds_bytes(int d, int S, int W, int waves) {
  const size_t r = (((size_t)d * d * 4) + 15) & ~(size_t)15;
  return r + (size_t)waves * bge_wave_bytes(d, S, W);
}

// LDS traffic between lanes of ONE wave: the hardware executes a wave's DS operations in order, so only the
// compiler has to be kept from reordering them.
__device__ __forceinline__ void wave_lds_fence() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ double bge_assemble(const BgeParams& bp, int j, int l, int d, double Nn, double ld_pa, double schur) {
  if (!(Nn > 0.0)) return 0.0;
  const double al = bp.alpha_lambd;
  const double ld_all = ld_pa + log(schur);
  return bp.gam[(size_t)j * (d + 1) + l] + 0.5 * (Nn + al - d + l) * ld_pa - 0.5 * (Nn + al - d + l + 1) * ld_all;
}

// Work queues for the problems that do not fit the per-lane tier: code = (m * d + j) * S + s.
struct BgeQueues {
  uint32_t* list12;      // 9 <= n <= 12   one problem per lane, registers only (k_bge_big, tier lane<12>)
  uint32_t* list16;      // 13 <= n <= 16  one problem per lane (tier lane<16>)
  uint32_t* list32;      // 17 <= n <= 32  32-lane groups, L in LDS
  uint32_t* listg;       // n > 32         one problem per wave
  unsigned int* counts;  // [4]: zero at creation, reset by k_lik_weights_score / the scoring path after every use
};

// The latent kernel matrix only needs z, which is final when a step starts.  On a single rank its blocks ride along in the
// k_bge_nodes launch (extra blockIdx.x range): that kernel is bound by VALU issue, k_kmat by the latency of the far cache
// levels, so the two overlap almost for free.  (Several ranks: the rows of the other ranks arrive with the all-gather, the
// kernel matrix stays in phase B.)
#define KMAT_BT 16
struct KmatFuse {
  const float* z;   // [M, len] (null: nothing fused)
  float* kout;      // [M, M]
  int len, M, nbx;  // nbx: first blockIdx.x of the kernel-matrix range
  float scale, h;
};
__device__ __forceinline__ void kmat_block(float* __restrict__ smem, const float* __restrict__ pack, size_t pack_stride,
                                           size_t seg_off, int len, float* __restrict__ kout, int m0, int M, float scale, float h,
                                           int symmetric, int a, int bt);

template <int WAVES, bool SAMPLE>
__global__ __launch_bounds__(64 * WAVES) void k_bge_nodes(const uint32_t* __restrict__ thr, uint64_t* __restrict__ masks,
                                                          double* __restrict__ node_scores, BgeParams bp, Key2 carry,
                                                          int m0, int M_global, int d, int S, int W, int layout,
                                                          unsigned long long* __restrict__ counters, BgeQueues qs, KmatFuse kf) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  if (WAVES == 4 && kf.z && (int)blockIdx.x >= kf.nbx) {  // kernel-matrix role (block-uniform)
    kmat_block(reinterpret_cast<float*>(smem_raw), kf.z, (size_t)kf.len, (size_t)0, kf.len, kf.kout, 0, kf.M, kf.scale, kf.h, 1,
               (int)blockIdx.y, (int)blockIdx.x - kf.nbx);
    return;
  }
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m = blockIdx.y;
  const int j = blockIdx.x * WAVES + wave;
  const bool active = j < d;
  float* Rs = reinterpret_cast<float*>(smem_raw);
  const size_t r_bytes = (((size_t)d * d * 4) + 15) & ~(size_t)15;
  unsigned char* wbase = smem_raw + r_bytes + (size_t)wave * bge_wave_bytes(d, S, W);
  uint64_t* mk = reinterpret_cast<uint64_t*>(wbase);
  uint32_t* thrs = reinterpret_cast<uint32_t*>(mk + (size_t)S * W);

  {  // R: shared by the block (WAVES > 1 requires n_mats == 1)
    const float* Rg = bp.R + (bp.n_mats > 1 ? (size_t)(blockIdx.x * WAVES) * d * d : 0);
    for (int e = tid; e < d * d; e += 64 * WAVES) Rs[e] = Rg[e];
  }
  // lim = 512 * thr:  y < lim  <=>  (y >> 9) < thr.  thr == 2^23 (p == 1.0f) has no 32-bit lim; those rows are forced on.
  uint32_t* lims = thrs + d;
  uint32_t* loc = lims + d;  // per sample: queue tier << 28 | index inside this block's reservation, or ~0
  __shared__ unsigned int blk_cnt[4], blk_base[4];
  if (tid < 4) blk_cnt[tid] = 0u;
  uint64_t force0 = 0, force1 = 0;
  if (active && SAMPLE) {
    for (int i0 = 0; i0 < d; i0 += 64) {
      const int i = i0 + lane;
      const uint32_t t = i < d ? thr[((size_t)m * d + i) * d + j] : 0u;
      if (i < d) {
        thrs[i] = t;
        lims[i] = t >= 0x800000u ? 0xFFFFFFFFu : t << 9;
      }
      const uint64_t f = __ballot(t >= 0x800000u);
      if (i0 == 0) force0 = f; else force1 = f;
    }
  }
  __syncthreads();
  if (active) {
  if (!SAMPLE) {  // scoring of given graphs (dibs_score_graphs): the parent sets come from the caller
    const uint64_t* mg = masks + ((size_t)m * d + j) * S * W;
    for (int e = lane; e < S * W; e += 64) mk[e] = mg[e];
  }

  // ---- 1. sample column j of the S graphs -------------------------------------------------------
  // particle key = row (1 + m_global) of split(carry, M+1); subk_ = row 1 of split(particle key)   dibs.py:350-351
  const Key2 kp = rng_split_row_uniform(carry, (uint32_t)M_global + 1u, (uint32_t)(m0 + m) + 1u, layout);
  const Key2 kg = rng_split_row_uniform(kp, 2u, 1u, layout);
  const uint64_t dd = (uint64_t)d * d, nbits = (uint64_t)S * dd;
  if (!SAMPLE) {
  } else if ((S & 1) == 0) {
    const int hS = S >> 1;
    for (int p = lane; p < hS; p += 64) {
      uint64_t a0 = 0, a1 = 0, b0 = 0, b1 = 0;
      const uint64_t cbase = (uint64_t)p * dd + j;
      if (layout == 0 && nbits < 0xFFFFFFFFull) {
        // legacy layout, 32-bit counters: element c pairs with c + n/2 in one Threefry call.  The kernel is bound by VALU
        // issue and this loop is most of it: per output bit one compare and one add-with-carry (word = 2 * word + bit,
        // i.e. rows arrive MSB first and the word is bit-reversed once at the end).
        const TfKeys tk = tf_keys(kg);
        uint32_t c0 = (uint32_t)cbase, c1 = (uint32_t)cbase + (uint32_t)(nbits >> 1);
        uint32_t wa[4] = {0u, 0u, 0u, 0u}, wb[4] = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int w32 = 0; w32 < 4; ++w32) {
          const int i0 = w32 * 32;
          if (i0 < d) {
            const int i1 = d < i0 + 32 ? d : i0 + 32;
            uint32_t A = 0u, B = 0u;
            int i = i0;
            for (; i + 1 < i1; i += 2, c0 += 2u * (uint32_t)d, c1 += 2u * (uint32_t)d) {
              uint32_t y0, y1, y2, y3;
              threefry2x32_uk2(tk, c0, c1, c0 + (uint32_t)d, c1 + (uint32_t)d, y0, y1, y2, y3);
              const uint32_t L = lims[i], L2 = lims[i + 1];
              asm volatile("v_cmp_lt_u32 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(A) : "v"(y0), "v"(L) : "vcc");
              asm volatile("v_cmp_lt_u32 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(B) : "v"(y1), "v"(L) : "vcc");
              asm volatile("v_cmp_lt_u32 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(A) : "v"(y2), "v"(L2) : "vcc");
              asm volatile("v_cmp_lt_u32 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(B) : "v"(y3), "v"(L2) : "vcc");
            }
            if (i < i1) {
              uint32_t y0, y1;
              threefry2x32_uk(tk, c0, c1, y0, y1);
              const uint32_t L = lims[i];
              asm volatile("v_cmp_lt_u32 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(A) : "v"(y0), "v"(L) : "vcc");
              asm volatile("v_cmp_lt_u32 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(B) : "v"(y1), "v"(L) : "vcc");
              c0 += (uint32_t)d;
              c1 += (uint32_t)d;
            }
            wa[w32] = __brev(A) >> (32 - (i1 - i0));
            wb[w32] = __brev(B) >> (32 - (i1 - i0));
          }
        }
        a0 = (((uint64_t)wa[1] << 32) | wa[0]) | force0;
        b0 = (((uint64_t)wb[1] << 32) | wb[0]) | force0;
        a1 = (((uint64_t)wa[3] << 32) | wa[2]) | force1;
        b1 = (((uint64_t)wb[3] << 32) | wb[2]) | force1;
      } else {
        for (int i = 0; i < d; ++i) {
          uint32_t y0, y1;
          rng_bits_pair(kg, nbits, cbase + (uint64_t)i * d, layout, y0, y1);
          const uint32_t t = thrs[i];
          const uint64_t ba = (uint64_t)((y0 >> 9) < t) << (i & 63), bb = (uint64_t)((y1 >> 9) < t) << (i & 63);
          if (i < 64) { a0 |= ba; b0 |= bb; } else { a1 |= ba; b1 |= bb; }
        }
      }
      mk[p * W] = a0;
      mk[(p + hS) * W] = b0;
      if (W > 1) { mk[p * W + 1] = a1; mk[(p + hS) * W + 1] = b1; }
    }
  } else {
    for (int s = lane; s < S; s += 64) {
      uint64_t a0 = 0, a1 = 0;
      for (int i = 0; i < d; ++i) {
        const uint32_t y = rng_bits_at(kg, nbits, (uint64_t)s * dd + (uint64_t)i * d + j, layout);
        const uint64_t ba = (uint64_t)((y >> 9) < thrs[i]) << (i & 63);
        if (i < 64) a0 |= ba; else a1 |= ba;
      }
      mk[s * W] = a0;
      if (W > 1) mk[s * W + 1] = a1;
    }
  }
  wave_lds_fence();
  if (SAMPLE) {
    uint64_t* mg = masks + ((size_t)m * d + j) * S * W;
    for (int e = lane; e < S * W; e += 64) mg[e] = mk[e];
  }

  const double Nn = bp.Nj[j];
  double* ns_out = node_scores + ((size_t)m * d + j) * S;
  const double score_l0 = bge_assemble(bp, j, 0, d, Nn, 0.0, (double)Rs[j * d + j]);
  double flops = 0.0;

  // ---- 2. n <= 8: one problem per lane, registers only; larger problems are queued for k_bge_big -----------
  for (int s0 = 0; s0 < S; s0 += 64) {
    const int s = s0 + lane;
    const bool valid = s < S;
    uint64_t w0 = valid ? mk[s * W] : 0ull, w1 = (valid && W > 1) ? mk[s * W + 1] : 0ull;
    const int l = __popcll(w0) + __popcll(w1);
    const bool small = valid && l <= 7;
    if (valid && l == 0) {
      // no parents: logdet R[pa,pa] = 0, Schur complement = R_jj
      ns_out[s] = score_l0;
    } else if (small) {
      int idx[8];
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        int b = 0;
        if (w0) { b = __ffsll((long long)w0) - 1; w0 &= w0 - 1; }
        else if (w1) { b = 64 + __ffsll((long long)w1) - 1; w1 &= w1 - 1; }
        idx[t] = (t == l) ? j : b;
      }
      const int n = l + 1;
      float A[8][8];
#pragma unroll
      for (int r = 0; r < 8; ++r)
#pragma unroll
        for (int c = 0; c <= r; ++c) A[r][c] = (r < n) ? Rs[idx[r] * d + idx[c]] : (r == c ? 1.0f : 0.0f);
      float piv[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        float dk = A[k][k];
#pragma unroll
        for (int p = 0; p < k; ++p) dk = fmaf(-A[k][p], A[k][p], dk);
        piv[k] = dk;
        const float inv = rsqrtf(dk);
#pragma unroll
        for (int r = k + 1; r < 8; ++r) {
          float v = A[r][k];
#pragma unroll
          for (int p = 0; p < k; ++p) v = fmaf(-A[r][p], A[k][p], v);
          A[r][k] = v * inv;
        }
      }
      double prod = 1.0, schur = 1.0;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        if (k < l) prod *= (double)piv[k];
        if (k == l) schur = (double)piv[k];
      }
      ns_out[s] = bge_assemble(bp, j, l, d, Nn, log(prod), schur);
      flops += (double)n * n * n / 3.0;
    }
    // queue the rest by tier.  Slots are reserved per BLOCK (LDS counters here, one global atomicAdd per tier and block below):
    // one atomic per wave pass made the four global counters the bottleneck while most problems are still queued.
    const bool big = valid && !small && l > 0;
    const int tier = !big ? -1 : (l <= 11 ? 0 : (l <= 15 ? 1 : (l <= 31 ? 2 : 3)));
    const unsigned long long lt = (1ull << lane) - 1ull;
    uint32_t myloc = 0xFFFFFFFFu;
#pragma unroll
    for (int tq = 0; tq < 4; ++tq) {
      const unsigned long long bal = __ballot(tier == tq);
      if (bal) {
        const int leader = __ffsll((long long)bal) - 1;
        unsigned int base = 0;
        if (lane == leader) base = atomicAdd(&blk_cnt[tq], (unsigned int)__popcll(bal));
        base = __shfl(base, leader, 64);
        if (tier == tq) myloc = ((uint32_t)tq << 28) | (base + (unsigned int)__popcll(bal & lt));
      }
    }
    if (valid) loc[s] = myloc;
  }
  if (counters) {
    const double tot = wave_sum_d(flops);
    if (lane == 0 && tot > 0.0) atomicAdd(counters, (unsigned long long)tot);  // (same-address atomics serialise: skip the zeros)
  }
  }  // active
  __syncthreads();
  if (tid < 4) {
    const unsigned int c = blk_cnt[tid];
    blk_base[tid] = c ? atomicAdd(&qs.counts[tid], c) : 0u;
  }
  __syncthreads();
  if (active)
    for (int s = lane; s < S; s += 64) {
      const uint32_t v = loc[s];
      if (v != 0xFFFFFFFFu) {
        const uint32_t tq = v >> 28;
        uint32_t* lst = tq == 0 ? qs.list12 : (tq == 1 ? qs.list16 : (tq == 2 ? qs.list32 : qs.listg));
        lst[blk_base[tq] + (v & 0x0FFFFFFFu)] = (uint32_t)(((size_t)m * d + j) * S + s);
      }
    }
}

// ------------------------------------------------------------------------------------------------
// K3b  queued BGe problems, spread evenly over the GPU (a hub node makes ALL samples of one (m, j) large; handled inside
//      k_bge_nodes they serialise in one wave and the kernel time is that tail).
//      G = 16 / 32: G lanes per problem (lane = row), 64 / G problems per wave.  Row r of L lives in the owning lane's
//      registers; the pivot row is re-read from LDS (one ds_read_b128 per four columns, broadcast inside the group); every
//      lane recomputes the pivot d_k from that row, so a column step depends on LDS only through the previous column's store.
//      G = 64: one problem per wave, lane owns rows r and r + 64, L in LDS (any n <= 128).
// grid = any (grid-stride over the queue), block = 256; dynamic LDS: bge_big_lds_bytes()
// ------------------------------------------------------------------------------------------------
__host__ __device__ inline size_t bge_big_wave_bytes(int d, int G) {
  const int ilen = bge_idx_len(d);
  const size_t lb = G == 64 ? (size_t)d * (d | 1) : (size_t)(64 / G) * G * (G + 4);
  return (((lb + 3) & ~(size_t)3) * 4 + (size_t)(G == 64 ? 1 : 64 / G) * ilen * 4 + 15) & ~(size_t)15;
}
// waves of a block that work in the one-problem-per-wave tier (each needs a d x d factor in LDS): as many as fit next to R
__host__ __device__ inline int bge_generic_waves(int d, bool r_in_lds) {
  const size_t r = r_in_lds ? ((((size_t)d * d * 4) + 15) & ~(size_t)15) : 0;
  const size_t room = (size_t)160 * 1024 - 1024 - r, per = bge_big_wave_bytes(d, 64);
  const int nw = (int)(room / per);
  return nw > 4 ? 4 : (nw < 1 ? 1 : nw);
}
__host__ __device__ inline size_t bge_big_lds_bytes(int d, int G, bool r_in_lds) {
  const size_t r = r_in_lds ? ((((size_t)d * d * 4) + 15) & ~(size_t)15) : 0;
  return r + (G == 64 ? bge_generic_waves(d, r_in_lds) : 4) * bge_big_wave_bytes(d, G);
}

template <int G, bool R_LDS>
__device__ __forceinline__ void bge_big_body(unsigned char* smem_raw, const uint64_t* __restrict__ masks, double* __restrict__ node_scores,
                                             const BgeParams& bp, const uint32_t* __restrict__ list, unsigned int cnt, int d, int S, int W,
                                             unsigned long long* __restrict__ counters, int vblock, int vgrid) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if ((unsigned int)(vblock * (G < 64 ? 4 * (64 / G) : 1)) >= cnt) return;  // nothing queued for this block (block-uniform)
  float* Rs = reinterpret_cast<float*>(smem_raw);
  const size_t r_bytes = R_LDS ? ((((size_t)d * d * 4) + 15) & ~(size_t)15) : 0;
  if (R_LDS) {
    for (int e = tid; e < d * d; e += 256) Rs[e] = bp.R[e];
    __syncthreads();
  }
  unsigned char* wbase = smem_raw + r_bytes + (size_t)wave * bge_big_wave_bytes(d, G);
  const int ilen = bge_idx_len(d);
  double flops = 0.0;
  if constexpr (G < 64) {
    constexpr int NP = 64 / G, LD = G + 4;
    float* Lb = reinterpret_cast<float*>(wbase);
    int* idxs = reinterpret_cast<int*>(Lb + ((NP * G * LD + 3) & ~3));
    const int grp = lane / G, r = lane % G;
    int* myidx = idxs + grp * ilen;
    float* Lh = Lb + grp * G * LD;
    for (unsigned int q = (vblock * 4 + wave) * NP; q < cnt; q += vgrid * 4 * NP) {
      const bool has = q + grp < cnt;
      const uint32_t code = has ? list[q + grp] : 0u;
      const int s = code % S, mj = code / S, j = mj % d;
      const float* R = R_LDS ? Rs : bp.R + (bp.n_mats > 1 ? (size_t)j * d * d : 0);
      int l = 0;
      myidx[r] = 0;  // slots >= n must hold a valid index
      wave_lds_fence();
      if (has) {
        for (int w = 0; w < W; ++w) {
          const uint64_t word = masks[(size_t)code * W + w];
          for (int bb = r; bb < 64; bb += G)
            if ((word >> bb) & 1ull) myidx[l + __popcll(word & ((1ull << bb) - 1ull))] = w * 64 + bb;
          l += __popcll(word);
        }
        if (r == 0) myidx[l] = j;
      }
      const int n = has ? l + 1 : 0;
      int nmax = 0;
#pragma unroll
      for (int g = 0; g < NP; ++g) {
        const int t = __shfl(n, g * G, 64);
        nmax = t > nmax ? t : nmax;
      }
      wave_lds_fence();
      const int ir = (r < n) ? myidx[r] : 0;
      float Ar[G], Dk[G], Lr[G];
#pragma unroll
      for (int k = 0; k < G; ++k) {
        const int ik = myidx[k];
        Ar[k] = R[ir * d + ik];
        Dk[k] = R[ik * d + ik];
        Lr[k] = 0.f;
      }
      float mypiv = 1.f;
#pragma unroll
      for (int k = 0; k < G; ++k) {
        // constant trip count (a `break` here keeps hipcc from unrolling and L would be indexed through GPR-index mode);
        // columns beyond the largest problem of this wave are skipped by a wave-uniform branch
        if (k < nmax) {
          float acc = Ar[k], pv = Dk[k];
#pragma unroll
          for (int p4 = 0; p4 < (k + 3) / 4; ++p4) {
            const float4 v = *reinterpret_cast<const float4*>(Lh + k * LD + p4 * 4);
            if (p4 * 4 + 0 < k) { acc = fmaf(-Lr[p4 * 4 + 0], v.x, acc); pv = fmaf(-v.x, v.x, pv); }
            if (p4 * 4 + 1 < k) { acc = fmaf(-Lr[p4 * 4 + 1], v.y, acc); pv = fmaf(-v.y, v.y, pv); }
            if (p4 * 4 + 2 < k) { acc = fmaf(-Lr[p4 * 4 + 2], v.z, acc); pv = fmaf(-v.z, v.z, pv); }
            if (p4 * 4 + 3 < k) { acc = fmaf(-Lr[p4 * 4 + 3], v.w, acc); pv = fmaf(-v.w, v.w, pv); }
          }
          if (r == k) mypiv = pv;
          const float lv = acc * rsqrtf(pv);
          Lr[k] = lv;
          if (r > k) Lh[r * LD + k] = lv;
          wave_lds_fence();
        }
      }
      double lg = (r < l) ? log((double)mypiv) : 0.0;
#pragma unroll
      for (int o = G / 2; o > 0; o >>= 1) lg += __shfl_xor(lg, o, 64);
      const double schur = (double)__shfl(mypiv, grp * G + (l < G ? l : 0), 64);
      if (r == 0 && has) {
        node_scores[(size_t)mj * S + s] = bge_assemble(bp, j, l, d, bp.Nj[j], lg, schur);
        flops += (double)n * n * n / 3.0;
      }
    }
  } else {
    const int nw = bge_generic_waves(d, R_LDS);
    if (wave >= nw) return;  // (no block-wide barrier below this point)
    float* Lb = reinterpret_cast<float*>(wbase);
    int* myidx = reinterpret_cast<int*>(Lb + (((size_t)d * (d | 1) + 3) & ~(size_t)3));
    const int ldl = d | 1;
    for (unsigned int q = vblock * nw + wave; q < cnt; q += vgrid * nw) {
      const uint32_t code = list[q];
      const int s = code % S, mj = code / S, j = mj % d;
      const float* R = R_LDS ? Rs : bp.R + (bp.n_mats > 1 ? (size_t)j * d * d : 0);
      int l = 0;
      for (int w = 0; w < W; ++w) {
        const uint64_t word = masks[(size_t)code * W + w];
        if ((word >> lane) & 1ull) myidx[l + __popcll(word & ((1ull << lane) - 1ull))] = w * 64 + lane;
        l += __popcll(word);
      }
      if (lane == 0) myidx[l] = j;
      const int n = l + 1;
      wave_lds_fence();
      float mypiv[2] = {1.f, 1.f};
      for (int kk = 0; kk < n; ++kk) {
        const int ik = myidx[kk];
        float accs[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int r = lane + h * 64;
          float acc = 0.f;
          if (r >= kk && r < n) {
            acc = R[myidx[r] * d + ik];
            const float* lr = Lb + (size_t)r * ldl;
            const float* lk = Lb + (size_t)kk * ldl;
#pragma unroll 8
            for (int p = 0; p < kk; ++p) acc = fmaf(-lr[p], lk[p], acc);  // (unrolled: several LDS reads in flight)
          }
          accs[h] = acc;
          if (r == kk) mypiv[h] = acc;
        }
        const float piv = __shfl(kk < 64 ? accs[0] : accs[1], kk & 63, 64);
        const float inv = rsqrtf(piv);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int r = lane + h * 64;
          if (r > kk && r < n) Lb[(size_t)r * ldl + kk] = accs[h] * inv;
        }
        wave_lds_fence();
      }
      double lg = 0.0;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int r = lane + h * 64;
        if (r < l) lg += log((double)mypiv[h]);
      }
      const double ld_pa = wave_sum_d(lg);
      const double schur = (double)__shfl(l < 64 ? mypiv[0] : mypiv[1], l & 63, 64);
      if (lane == 0) {
        node_scores[(size_t)mj * S + s] = bge_assemble(bp, j, l, d, bp.Nj[j], ld_pa, schur);
        flops += (double)n * n * n / 3.0;
      }
      wave_lds_fence();
    }
  }
  if (counters) {
    const double tot = wave_sum_d(flops);
    if (lane == 0 && tot > 0.0) atomicAdd(counters, (unsigned long long)tot);
  }
}

// Tier lane<NMAX>: one queued problem per LANE, the whole factorisation in registers (no LDS traffic, no barriers, every lane
// busy because the queue is dense).  Rows n..NMAX-1 are identity rows, so the unrolled code is the same for every lane.
template <int NMAX, bool R_LDS>
__device__ __forceinline__ void bge_lane_body(unsigned char* smem_raw, const uint64_t* __restrict__ masks, double* __restrict__ node_scores,
                                              const BgeParams& bp, const uint32_t* __restrict__ list, unsigned int cnt, int d, int S, int W,
                                              unsigned long long* __restrict__ counters, int vblock, int vgrid) {
  const int tid = threadIdx.x, lane = tid & 63;
  if ((unsigned int)(vblock * 256) >= cnt) return;  // nothing queued for this block (block-uniform)
  float* Rs = reinterpret_cast<float*>(smem_raw);
  if (R_LDS) {
    for (int e = tid; e < d * d; e += 256) Rs[e] = bp.R[e];
    __syncthreads();
  }
  double flops = 0.0;
  for (unsigned int q = vblock * 256 + tid; q < cnt; q += vgrid * 256) {
    const uint32_t code = list[q];
    const int s = code % S, mj = code / S, j = mj % d;
    const float* R = R_LDS ? Rs : bp.R + (bp.n_mats > 1 ? (size_t)j * d * d : 0);
    uint64_t w0 = masks[(size_t)code * W], w1 = W > 1 ? masks[(size_t)code * W + 1] : 0ull;
    const int l = __popcll(w0) + __popcll(w1), n = l + 1;
    int idx[NMAX];
#pragma unroll
    for (int t = 0; t < NMAX; ++t) {
      int b = 0;
      if (w0) { b = __ffsll((long long)w0) - 1; w0 &= w0 - 1; }
      else if (w1) { b = 64 + __ffsll((long long)w1) - 1; w1 &= w1 - 1; }
      idx[t] = (t == l) ? j : b;
    }
    float A[NMAX][NMAX];
#pragma unroll
    for (int r = 0; r < NMAX; ++r)
#pragma unroll
      for (int c = 0; c <= r; ++c) A[r][c] = (r < n) ? R[idx[r] * d + idx[c]] : (r == c ? 1.0f : 0.0f);
    double prod = 1.0, schur = 1.0;
#pragma unroll
    for (int k = 0; k < NMAX; ++k) {
      float dk = A[k][k];
#pragma unroll
      for (int p = 0; p < k; ++p) dk = fmaf(-A[k][p], A[k][p], dk);
      if (k < l) prod *= (double)dk;
      if (k == l) schur = (double)dk;
      const float inv = rsqrtf(dk);
#pragma unroll
      for (int r = k + 1; r < NMAX; ++r) {
        float v = A[r][k];
#pragma unroll
        for (int p = 0; p < k; ++p) v = fmaf(-A[r][p], A[k][p], v);
        A[r][k] = v * inv;
      }
    }
    node_scores[(size_t)mj * S + s] = bge_assemble(bp, j, l, d, bp.Nj[j], log(prod), schur);
    flops += (double)n * n * n / 3.0;
  }
  if (counters) {
    const double tot = wave_sum_d(flops);
    if (lane == 0 && tot > 0.0) atomicAdd(counters, (unsigned long long)tot);
  }
}

// one launch for the four queues: consecutive block ranges work on list12 / list16 (one problem per lane), list32 (32-lane
// groups) and the generic one-problem-per-wave tier.  In the steady state of a run all queues are empty and every block
// returns at once; separate launches would cost their launch latencies for nothing.
template <bool R_LDS>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) void k_bge_big(const uint64_t* __restrict__ masks, double* __restrict__ node_scores, BgeParams bp,
                                                 BgeQueues qs, int n12, int n16, int n32, int d, int S, int W,
                                                 unsigned long long* __restrict__ counters) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int b = blockIdx.x, ng = gridDim.x - n12 - n16 - n32;
  if (b < n12) bge_lane_body<12, R_LDS>(smem_raw, masks, node_scores, bp, qs.list12, qs.counts[0], d, S, W, counters, b, n12);
  else if (b < n12 + n16) bge_lane_body<16, R_LDS>(smem_raw, masks, node_scores, bp, qs.list16, qs.counts[1], d, S, W, counters, b - n12, n16);
  else if (b < n12 + n16 + n32)
    bge_big_body<32, R_LDS>(smem_raw, masks, node_scores, bp, qs.list32, qs.counts[2], d, S, W, counters, b - n12 - n16, n32);
  else bge_big_body<64, R_LDS>(smem_raw, masks, node_scores, bp, qs.listg, qs.counts[3], d, S, W, counters, b - n12 - n16 - n32, ng);
}

// out[s] = sum_j node_scores[j][s]   (scoring of given graphs)
__global__ void k_sum_nodes(const double* __restrict__ node_scores, float* __restrict__ out, int d, int S) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= S) return;
  double t = 0.0;
  for (int j = 0; j < d; ++j) t += node_scores[(size_t)j * S + s];
  out[s] = (float)t;
}

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
  if (A.queue_counts && m == 0 && y == 0 && threadIdx.x < 4) A.queue_counts[threadIdx.x] = 0u;
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

__global__ __launch_bounds__(256) void k_lik_weights_score(LikArgs A) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  lik_weights_block(smem_raw, A, blockIdx.x, blockIdx.y);
}

// ------------------------------------------------------------------------------------------------
// K5  acyclicity gradient: for Gumbel-soft graphs G~ = sigmoid(tau (eps + alpha s)), M = I + G~/d,
//     dh/dG~ = (M^{d-1})^T (h = tr(M^d) - d), chained through G~.  Matrix powers on f32 MFMA, all operands
//     resident in LDS.  Each block handles CPB chains of one particle and writes their SUM.
//     reference: graph_utils.py:8-28, dibs.py:121-140, 557-601
// grid = (ceil(Sa / CPB), Mloc), block = 256; dynamic LDS = 3 * DP * LD * 4, DP = 16 NT, LD = DP + 2
// ------------------------------------------------------------------------------------------------
// Matrices live in LDS as [DP rows][LD] with the COLUMNS PERMUTED: logical column c sits at pc(c) = (c & 15) * NT + (c >> 4),
// so the NT values {c, c+16, c+32, ...} that one lane needs for the B fragments of a k-step (and produces in the C tile)
// are contiguous: one ds_read_b128 / ds_write_b128 for NT = 4.  LD = 16 NT + 4.
template <int NT>
__device__ __forceinline__ int acyc_pc(int c) { return (c & 15) * NT + (c >> 4); }

// C = A * B.  Operands are OFFSETS (in floats) into the kernel's LDS array so that every access is a ds_* instruction
// (a runtime-selected generic pointer would turn them into flat accesses).  The next k-step's fragments are loaded
// while the current MFMAs issue.
// ODD: the number of k-steps (kp / 4) is odd.  A template parameter, not a runtime `if` around the last MFMA: the accumulators
// must not meet a control-flow join between an MFMA and the s_nop that covers its latency -- hipcc places register copies
// for the join right behind the (opaque) asm MFMA and reads the accumulator too early.
// ZC (needs >= 2 k-steps): the first MFMA of every tile takes C = 0 as an inline constant instead of a zeroed accumulator.
template <int NT, bool ODD, bool ZC>
__device__ __forceinline__ void lds_matmul(float* __restrict__ lds, int c_off, int a_off, int b_off, int kp, int lane,
                                           int wave) {
  constexpr int DP = 16 * NT, LD = DP + 4;
  for (int ti = wave; ti < NT; ti += 4) {
    f32x4 acc[NT];
    if constexpr (!ZC) {
#pragma unroll
      for (int tj = 0; tj < NT; ++tj) acc[tj] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    // A[row][k]: k = k0 + kk -> physical ((k0 & 15) + kk) * NT + (k0 >> 4)
    const int ap = a_off + (ti * 16 + (lane & 15)) * LD + (lane >> 4) * NT;
    // B[k][tj * 16 + col], tj = 0..NT-1 -> physical col * NT + tj (contiguous)
    const int bq = b_off + (lane >> 4) * LD + (lane & 15) * NT;
    // two-stage register pipeline over the kp / 4 k-steps (step s: k0 = 4 s); the loads of the following step are
    // issued before the MFMAs of the current one.  A step index == nsteps is loaded but never used (addresses stay
    // inside the LDS allocation: one slack row is allocated behind the last buffer).
    const int ksteps = kp >> 2, nsteps = ksteps & ~1;  // the pipelined loop takes the steps in pairs; an odd last step follows it
    float a0, a1, b0[NT], b1[NT];
#define ACYC_LOAD(A_, B_, S_)                                                     \
    {                                                                             \
      const int kk0 = (S_) << 2;                                                  \
      A_ = lds[ap + (kk0 & 15) * NT + (kk0 >> 4)];                                \
      if constexpr (NT == 4) {                                                    \
        const float4 t4 = *reinterpret_cast<const float4*>(lds + bq + kk0 * LD);  \
        B_[0] = t4.x; B_[1] = t4.y; B_[2] = t4.z; B_[3] = t4.w;                   \
      } else {                                                                    \
        _Pragma("unroll") for (int tj = 0; tj < NT; ++tj) B_[tj] = lds[bq + kk0 * LD + tj]; \
      }                                                                           \
    }
    // MFMA as inline asm with the accumulator tied in place ("+a"): with the builtin, hipcc renamed the accumulators
    // across the pipelined loop (v_accvgpr_read / _mov / _write + s_nop at the loop head), serialising every iteration.
    // Hazards hipcc cannot see around asm: accumulator init -> first MFMA (s_nop below) and last MFMA -> accumulator
    // read (s_nop after the loop); back-to-back MFMAs on the same accumulator need none.
#define ACYC_MFMA(A_, B_) \
    _Pragma("unroll") for (int tj = 0; tj < NT; ++tj) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[tj]) : "v"(A_), "v"(B_[tj]));
#define ACYC_MFMA_Z(A_, B_) \
    _Pragma("unroll") for (int tj = 0; tj < NT; ++tj) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, 0" : "=&v"(acc[tj]) : "v"(A_), "v"(B_[tj]));
    ACYC_LOAD(a0, b0, 0)
    int st0 = 0;
    if constexpr (ZC) {
      ACYC_LOAD(a1, b1, 1)
      ACYC_MFMA_Z(a0, b0)
      ACYC_LOAD(a0, b0, 2)
      ACYC_MFMA(a1, b1)
      st0 = 2;
    } else {
      asm volatile("s_nop 4" ::: "memory");
    }
#undef ACYC_MFMA_Z
#pragma unroll 1
    for (int st = st0; st < nsteps; st += 2) {
      ACYC_LOAD(a1, b1, st + 1)
      ACYC_MFMA(a0, b0)
      ACYC_LOAD(a0, b0, st + 2)
      ACYC_MFMA(a1, b1)
    }
    if constexpr (ODD) { ACYC_MFMA(a0, b0) }  // (its fragments were loaded by the last pass, or by the prologue when ksteps == 1)
    // last MFMA -> accumulator read: one wait for the whole group (volatile asm statements keep their order, so every
    // accumulator's first read sits behind the s_nop)
    asm volatile("s_nop 15\n\ts_nop 7" : "+v"(acc[NT - 1]));
#pragma unroll
    for (int tj = 0; tj < NT - 1; ++tj) asm volatile("" : "+v"(acc[tj]));
#undef ACYC_LOAD
#undef ACYC_MFMA
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int o = c_off + (ti * 16 + (lane >> 4) * 4 + r) * LD + (lane & 15) * NT;
      float tmp[NT];
#pragma unroll
      for (int tj = 0; tj < NT; ++tj) {
        // Force the MFMA result through a VGPR: hipcc (ROCm 7.2) otherwise emits `ds_write_b32 vaddr, aN` (AGPR data
        // operand) for part of the tile, which stored wrong values on gfx950 (scripts/probe/acyc_probe.hip).
        tmp[tj] = acc[tj][r];
        asm volatile("" : "+v"(tmp[tj]));
      }
      if constexpr (NT == 4) {
        *reinterpret_cast<float4*>(lds + o) = make_float4(tmp[0], tmp[1], tmp[2], tmp[3]);
      } else {
#pragma unroll
        for (int tj = 0; tj < NT; ++tj) lds[o + tj] = tmp[tj];
      }
    }
  }
}

template <int NT, bool PAIRED>
__global__ __launch_bounds__(256) void k_acyc(const float* __restrict__ scores, float* __restrict__ part, Key2 carry, int m0,
                                              int M_global, int d, int Sa, int cpb, float alpha, float tau, int layout,
                                              int tiny, int n_acyc_blk, LikArgs lik) {
  constexpr int DP = 16 * NT, LD = DP + 4, BUF = DP * LD;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  if ((int)blockIdx.x >= n_acyc_blk) {  // score-estimator role (block-uniform): see lik_weights_block
    lik_weights_block(reinterpret_cast<unsigned char*>(smem), lik, (int)blockIdx.y, (int)blockIdx.x - n_acyc_blk);
    return;
  }
  const int blk = blockIdx.x, m = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const Key2 km = rng_split_row_uniform(carry, (uint32_t)M_global + 1u, (uint32_t)(m0 + m) + 1u, layout);  // dibs.py:595: key used directly
  const uint64_t dd = (uint64_t)d * d, nbits = (uint64_t)Sa * dd;
  const int kp = (d + 3) & ~3;
  const bool kodd = (kp >> 2) & 1;
  const float inv_d = 1.0f / (float)d;
  const float* sm = scores + (size_t)m * dd;
  // thread t owns column pj = t % DP and rows pi0 + q * R of the d x d matrix: the same elements in every chain, and all
  // LDS offsets are compile-time functions of q (a d-dependent mapping would keep more lanes busy at d = 50 but its
  // offsets end up as loop-invariant VGPRs and cost an occupancy step).  Registers that stay live across the matmuls
  // decide the occupancy, so only `out` and the second chain's soft graph are kept; exp(-alpha s) is recomputed when
  // noise is drawn and g for the epilogue is read back from buffer 0.
  constexpr int R = 256 / DP, EPT = (DP + R - 1) / R;
  const int pj = tid % DP, pi0 = tid / DP;
  const bool pact = pi0 < R && pj < d;
  const int pcj = acyc_pc<NT>(pj);
  const bool fast = tau == 1.0f;   // sigmoid(eps + a) with eps = log(u / (1 - u))  ==  u / (u + (1 - u) exp(-a)): no log / exp per draw
  const float ulo = tiny ? 1.17549435e-38f : 1.1920929e-07f;
  const float fd = (float)d;
  // legacy PRNG layout: element e of the [Sa, d, d] noise tensor shares its Threefry call with element e + Sa*d*d/2, i.e.
  // chain sa with chain sa + Sa/2 at the same (i, j).  A block therefore takes both chains of a pair (`paired`; the host
  // sizes the grid in pairs) and draws the noise of both with one call per element -- the noise is most of this kernel's
  // VALU work, and VALU work does not overlap with the f32 MFMAs.
  constexpr bool paired = PAIRED;  // host: layout == legacy && Sa even && Sa * d * d < 2^32 (one code path per instantiation: SGPR pressure)
  const int n_units = paired ? (Sa >> 1) : Sa;
  const TfKeys tk = tf_keys(km);
  float out[EPT], gnext[EPT];
#pragma unroll
  for (int q = 0; q < EPT; ++q) {
    out[q] = 0.f;
    gnext[q] = 0.f;
  }
  for (int e = tid; e < BUF; e += 256) smem[e] = 0.f;  // padding of buffer 0: zeroed once, never written afterwards

  for (int c = 0; c < cpb; ++c) {
    const int unit = blk * cpb + c;
    if (unit >= n_units) break;
    for (int hf = 0; hf < (paired ? 2 : 1); ++hf) {
      const int sa = paired ? unit + hf * (Sa >> 1) : unit;
      const float* sml = sm;
      asm volatile("" : "+s"(sml));  // opaque per pass: otherwise exp(-alpha s) is hoisted out of the loops into EPT live VGPRs
      __syncthreads();
      // buffer 0: M = I + G~/d  (permuted columns)
#pragma unroll
      for (int q = 0; q < EPT; ++q) {
        const int i = pi0 + q * R;
        if (pact && i < d) {
          float v = 1.0f;
          if (i != pj) {
            float g;
            if (paired && hf == 1) {
              g = gnext[q];
            } else {
              const float as = alpha * sml[i * d + pj];
              const float ea = fast ? expf(-as) : as;
              uint32_t y0, y1 = 0u;
              if (paired) {
                const uint32_t c0 = (uint32_t)((uint64_t)sa * dd) + (uint32_t)(i * d + pj);
                threefry2x32_uk(tk, c0, c0 + (uint32_t)(nbits >> 1), y0, y1);
              } else {
                y0 = rng_bits_at(km, nbits, (uint64_t)sa * dd + (uint64_t)(i * d + pj), layout);
              }
              if (fast) {
                const float u0 = rng_uniform(y0, ulo, 1.0f), u1 = rng_uniform(y1, ulo, 1.0f);
                g = u0 / (u0 + (1.0f - u0) * ea);
                gnext[q] = u1 / (u1 + (1.0f - u1) * ea);
              } else {
                g = 1.0f / (1.0f + expf(-tau * (rng_logistic(y0, tiny) + ea)));
                gnext[q] = 1.0f / (1.0f + expf(-tau * (rng_logistic(y1, tiny) + ea)));
              }
            }
            v = g * inv_d;
          }
          smem[i * LD + pcj] = v;
        }
      }
      __syncthreads();
      // left-to-right binary powering of e = d - 1; the running power ping-pongs between buffers 1 and 2
      const int ex = d - 1;
      int cur = 0;
      if (ex >= 1) {
        const int hb = 31 - __builtin_clz((unsigned)ex);
        for (int b = hb - 1; b >= 0; --b) {
          int dst = (cur == BUF) ? 2 * BUF : BUF;
          if (kp < 8) lds_matmul<NT, true, false>(smem, dst, cur, cur, kp, lane, wave);
          else if (kodd) lds_matmul<NT, true, true>(smem, dst, cur, cur, kp, lane, wave);
          else lds_matmul<NT, false, true>(smem, dst, cur, cur, kp, lane, wave);
          __syncthreads();
          cur = dst;
          if ((ex >> b) & 1) {
            dst = (cur == BUF) ? 2 * BUF : BUF;
            if (kp < 8) lds_matmul<NT, true, false>(smem, dst, cur, 0, kp, lane, wave);
            else if (kodd) lds_matmul<NT, true, true>(smem, dst, cur, 0, kp, lane, wave);
            else lds_matmul<NT, false, true>(smem, dst, cur, 0, kp, lane, wave);
            __syncthreads();
            cur = dst;
          }
        }
      }
      // out[i][j] += (M^{d-1})[j][i] * tau * alpha * g (1 - g)   (i != j);  g = d * M[i][j] from buffer 0
#pragma unroll
      for (int q = 0; q < EPT; ++q) {
        const int i = pi0 + q * R;
        if (pact && i < d && i != pj) {
          const float g = smem[i * LD + pcj] * fd;
          out[q] += smem[cur + pj * LD + acyc_pc<NT>(i)] * tau * alpha * g * (1.0f - g);
        }
      }
    }
  }
  if (pact) {
    float* po = part + ((size_t)m * n_acyc_blk + blk) * dd;
#pragma unroll
    for (int q = 0; q < EPT; ++q) {
      const int i = pi0 + q * R;
      if (i < d) po[i * d + pj] = out[q];
    }
  }
}

// ------------------------------------------------------------------------------------------------
// K7a total score-space gradient  W = W_lik - beta * mean_s(W_acyc) + W_prior   (elementwise; wide grid)
//     reference: dibs.py:604-658 (latent prior), graph.py:93-108 / 182-196 (prior on edge probabilities)
// grid = (Mloc, 4), block = 256
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_wtotal(const float* __restrict__ probs, const float* __restrict__ w_lik,
                                                const float* __restrict__ acyc_part, int n_part, float* __restrict__ w_acyc,
                                                float* __restrict__ w_tot, int d, int Sa, float alpha, float beta,
                                                int prior_kind, float er_c) {
  const int m = blockIdx.x, tid = threadIdx.x;
  const size_t dd = (size_t)d * d;
  const float inv_sa = 1.0f / (float)Sa;
  const float* pm = probs + m * dd;
  for (int e = blockIdx.y * 256 + tid; e < (int)dd; e += 256 * gridDim.y) {
    const int i = e / d, j = e - i * d;
    float ac = 0.f;
    int q = 0;
    for (; q + 4 <= n_part; q += 4) {  // loads in flight together, additions in fixed order
      float v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = acyc_part[((size_t)m * n_part + q + u) * dd + e];
#pragma unroll
      for (int u = 0; u < 4; ++u) ac += v[u];
    }
    for (; q < n_part; ++q) ac += acyc_part[((size_t)m * n_part + q) * dd + e];
    ac *= inv_sa;
    w_acyc[m * dd + e] = ac;
    float pr = 0.f;
    if (i != j && prior_kind != 2) {
      const float p = pm[e];
      const float dp = alpha * p * (1.0f - p);
      if (prior_kind == 0) pr = er_c * dp;
      else {
        float cs = 0.f;  // soft in-degree of node j (graph.py:182-196)
        for (int r = 0; r < d; ++r) cs += pm[r * d + j];
        pr = (-3.0f / (1.0f + cs)) * dp;
      }
    }
    w_tot[m * dd + e] = w_lik[m * dd + e] - beta * ac + pr;
  }
}

// ------------------------------------------------------------------------------------------------
// K7b back-projection: grad = [W V, W^T U] - z / sigma^2, written next to a copy of z into the packed all-gather row
//     [z | grad_z | theta | grad_theta].   (autodiff of dibs.py:179-180 in closed form)
// grid = (Mloc, ZS), block = 256; block y handles rows i = y, y + ZS, ...; dynamic LDS = (d*d + 2*d*k) * 4
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_zgrad(const float* __restrict__ z, const float* __restrict__ w_tot,
                                               float* __restrict__ pack, size_t pack_stride, int m0, int d, int k,
                                               float inv_sig2) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Wm = smem;
  float2* Zs = reinterpret_cast<float2*>(smem + (size_t)d * d);
  const int m = blockIdx.x, tid = threadIdx.x;
  const size_t dd = (size_t)d * d;
  for (int e = tid; e < (int)dd; e += 256) Wm[e] = w_tot[m * dd + e];
  const float2* zm = reinterpret_cast<const float2*>(z + (size_t)m * d * k * 2);
  for (int e = tid; e < d * k; e += 256) Zs[e] = zm[e];
  __syncthreads();
  float* prow = pack + (size_t)(m0 + m) * pack_stride;
  float2* pz = reinterpret_cast<float2*>(prow);
  float2* pg = reinterpret_cast<float2*>(prow + (size_t)d * k * 2);
  const int nrow = (d - blockIdx.y + gridDim.y - 1) / gridDim.y;
  for (int e = tid; e < nrow * k; e += 256) {
    const int i = blockIdx.y + (e / k) * gridDim.y, q = e % k;
    float su = 0.f, sv = 0.f;
    for (int j = 0; j < d; ++j) {
      const float2 zj = Zs[j * k + q];
      su = fmaf(Wm[i * d + j], zj.y, su);  // dU[i,q] = sum_j W[i,j] V[j,q]
      sv = fmaf(Wm[j * d + i], zj.x, sv);  // dV[i,q] = sum_j W[j,i] U[j,q]
    }
    const float2 zi = Zs[i * k + q];
    pz[i * k + q] = zi;
    pg[i * k + q] = make_float2(su - zi.x * inv_sig2, sv - zi.y * inv_sig2);
  }
}

// ------------------------------------------------------------------------------------------------
// K8a kernel matrix slab: kz[a, b] = scale * exp(-||z_a - z_b||^2 / h) for local a, all b (direct differences:
//     the entries are ~e^-40 at d = 50 and must not be flushed or computed by cancellation).
//     reference: kernel.py:20-30 / 52-71, svgd.py:165-176 / 537-551
// grid = Mloc, block = 256; dynamic LDS = len * 4
// ------------------------------------------------------------------------------------------------
#define KMAT_CH 32768  // floats of z_a staged in LDS at a time (128 KiB); longer vectors (DenseNN theta at d = 100) go in chunks
__device__ __forceinline__ void kmat_block(float* __restrict__ smem, const float* __restrict__ pack, size_t pack_stride,
                                           size_t seg_off, int len, float* __restrict__ kout, int m0, int M, float scale, float h,
                                           int symmetric, int a, int bt) {
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
    for (int e = tid; e < len4; e += 256) reinterpret_cast<float4*>(smem)[e] = reinterpret_cast<const float4*>(za + c0)[e];
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
    }
  }
}

__global__ __launch_bounds__(256) void k_kmat(const float* __restrict__ pack, size_t pack_stride, size_t seg_off, int len,
                                              float* __restrict__ kout, int m0, int M, float scale, float h, int symmetric) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  kmat_block(smem, pack, pack_stride, seg_off, len, kout, m0, M, scale, h, symmetric, blockIdx.x, blockIdx.y);
}

// ------------------------------------------------------------------------------------------------
// K8b+K9 SVGD transform + optimizer step on one segment (z or theta) of the packed rows:
//     phi_a = -(1/M) sum_b [ (kz+kt)[a,b] grad_b - (2/h) kseg[a,b] (x_b - x_a) ]   (column kxx[:,a], symmetric kernel)
//     rmsprop: v = 0.9 v + 0.1 phi^2 ; x -= step * phi / sqrt(v + 1e-8)      |  gd: x -= step * phi
//     reference: svgd.py:194-224, 591-670, 265, 718-719; jax.example_libraries.optimizers.rmsprop
// grid = (ceil(len / 256), ceil(Mloc / TA)), block = 256
// ------------------------------------------------------------------------------------------------
// block = 64 consecutive elements x TA local particles; wave w sums over the quarter b in [w Mq, (w+1) Mq) of the particles and
// the four partial sums are added in wave order (the order is a function of M only: results do not depend on TA or the rank
// count).  Every [z_b | grad_b] element read from L2 serves TA particles; the kernel tables sit in LDS as [b][TA].
template <int TA>
__global__ __launch_bounds__(256) void k_phi_update(const float* __restrict__ pack, size_t pack_stride, size_t val_off,
                                                    size_t grad_off, int len, const float* __restrict__ kz,
                                                    const float* __restrict__ kt, int seg_is_theta, float* __restrict__ x,
                                                    float* __restrict__ v, float* __restrict__ phi_out, int m0, int Mloc,
                                                    int M, float h, float stepsize, int rmsprop) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* ksum = smem;                   // [M][TA]  kz + kt
  float* krep = smem + (size_t)TA * M;  // [M][TA]  (2 / h) * kernel whose gradient gives the repulsion
  float* part = krep + (size_t)TA * M;  // [4][TA][64] partial sums
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int a0 = blockIdx.y * TA;
  const float c2h = 2.0f / h;
  for (int e = tid; e < TA * M; e += 256) {
    const int b = e / TA, q = e - b * TA, a = a0 + q;
    float s = 0.f, r = 0.f;
    if (a < Mloc) {
      const float z1 = kz[(size_t)a * M + b];
      const float t1 = kt ? kt[(size_t)a * M + b] : 0.f;
      s = z1 + t1;
      r = seg_is_theta ? t1 : z1;
    }
    ksum[e] = s;
    krep[e] = c2h * r;
  }
  __syncthreads();
  const int i = blockIdx.x * 64 + lane;
  const bool ok = i < len;
  float xa[TA], acc[TA];
#pragma unroll
  for (int q = 0; q < TA; ++q) {
    const int a = a0 + q;
    xa[q] = (ok && a < Mloc) ? pack[(size_t)(m0 + a) * pack_stride + val_off + i] : 0.f;
    acc[q] = 0.f;
  }
  const int Mq = (M + 3) >> 2;
  const int b_lo = wave * Mq, b_hi = (b_lo + Mq < M) ? b_lo + Mq : M;
  if (ok) {
    int b = b_lo;
    // 16 rows (32 loads) are issued together; the FMA loop runs on packed f32 (v_pk_fma_f32: two particles per instruction --
    // this loop is bound by VALU issue)
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    f32x2 acc2[TA / 2], xa2[TA / 2];
#pragma unroll
    for (int q2 = 0; q2 < TA / 2; ++q2) {
      acc2[q2] = f32x2{acc[2 * q2], acc[2 * q2 + 1]};
      xa2[q2] = f32x2{xa[2 * q2], xa[2 * q2 + 1]};
    }
    for (; b + 16 <= b_hi; b += 16) {
      float g[16], xb[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        g[u] = pack[(size_t)(b + u) * pack_stride + grad_off + i];
        xb[u] = pack[(size_t)(b + u) * pack_stride + val_off + i];
      }
      asm volatile("" ::: "memory");  // keep the 32 loads ahead of the arithmetic (hipcc would interleave them to save VGPRs)
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const f32x2 g2 = f32x2{g[u], g[u]}, xb2 = f32x2{xb[u], xb[u]};
#pragma unroll
        for (int q4 = 0; q4 < TA / 4; ++q4) {
          const float4 ks = *reinterpret_cast<const float4*>(ksum + (size_t)(b + u) * TA + q4 * 4);
          const float4 kr = *reinterpret_cast<const float4*>(krep + (size_t)(b + u) * TA + q4 * 4);
          const f32x2 ks_lo = f32x2{ks.x, ks.y}, ks_hi = f32x2{ks.z, ks.w}, kr_lo = f32x2{kr.x, kr.y}, kr_hi = f32x2{kr.z, kr.w};
          acc2[2 * q4] = __builtin_elementwise_fma(-kr_lo, xb2 - xa2[2 * q4], __builtin_elementwise_fma(ks_lo, g2, acc2[2 * q4]));
          acc2[2 * q4 + 1] = __builtin_elementwise_fma(-kr_hi, xb2 - xa2[2 * q4 + 1], __builtin_elementwise_fma(ks_hi, g2, acc2[2 * q4 + 1]));
        }
      }
    }
#pragma unroll
    for (int q2 = 0; q2 < TA / 2; ++q2) {
      acc[2 * q2] = acc2[q2].x;
      acc[2 * q2 + 1] = acc2[q2].y;
    }
    for (; b < b_hi; ++b) {
      const float g = pack[(size_t)b * pack_stride + grad_off + i];
      const float xb = pack[(size_t)b * pack_stride + val_off + i];
#pragma unroll
      for (int q = 0; q < TA; ++q) acc[q] = fmaf(-krep[(size_t)b * TA + q], xb - xa[q], fmaf(ksum[(size_t)b * TA + q], g, acc[q]));
    }
  }
#pragma unroll
  for (int q = 0; q < TA; ++q) part[(wave * TA + q) * 64 + lane] = acc[q];
  __syncthreads();
  if (!ok) return;
#pragma unroll
  for (int qq = 0; qq < TA / 4; ++qq) {
    const int q = wave + 4 * qq, a = a0 + q;
    if (a >= Mloc) continue;
    const float tot = ((part[(0 * TA + q) * 64 + lane] + part[(1 * TA + q) * 64 + lane]) + part[(2 * TA + q) * 64 + lane]) +
                      part[(3 * TA + q) * 64 + lane];
    const float phi = -tot / (float)M;
    const float xv = pack[(size_t)(m0 + a) * pack_stride + val_off + i];
    const size_t o = (size_t)a * len + i;
    if (phi_out) phi_out[o] = phi;
    if (rmsprop) {
      const float vv = v[o] * 0.9f + phi * phi * 0.1f;
      v[o] = vv;
      x[o] = xv - stepsize * phi / sqrtf(vv + 1e-8f);
    } else {
      x[o] = xv - stepsize * phi;
    }
  }
}
