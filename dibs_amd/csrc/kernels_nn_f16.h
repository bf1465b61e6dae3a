// JointDiBS + DenseNonlinearGaussian: log p(theta, D | G_s) of all samples with the first layer on the f16 matrix pipe (gfx950)
#pragma once
#include "kernels_nn.h"
#include "kernels_acyc_f16.h"
#include <type_traits>

// ------------------------------------------------------------------------------------------------
// K-NN-hf  k_nn_logprobs (kernels_nn.h; reference: nonlinearGaussian.py:35-81, 248-326) with the per-hidden-unit product
//          pre_h = x [N, d] * (G_s o W1T_h) [d, d]  evaluated as float products on v_mfma_f32_16x16x32_f16: both operands are carried as
//          two f16 pieces of a block-scaled value (the arithmetic of k_acyc_hf, kernels_acyc_f16.h: x 2^e = h + m, product =
//          Ah Bh + Ah Bm + Am Bh, three 16-cycle instructions per 16 x 16 x 32 block where the f32 MFMA needs eight of 32 cycles).
//
//  * x never changes: every wave keeps the fragments of ITS row tile of x (left operand; rows 16 w .. 16 w + 15, split once per block with
//    the exponent of max |x|) and the x of its output elements in registers -- x is not in LDS at all.  The MFMA is issued with swapped
//    operands (kernels_acyc_bf16.h), so lane (g, r) holds pre[n = 16 w + r][j = 16 tj + 4 g + i].
//  * the per-unit operand is an image [piece][column tile][row a][16 columns] f16 read with ds_read_b64_tr_b16 (chunk rotation as in
//    k_acyc_bf).  W1 arrives pre-scaled per particle (2^ew from max |W1|, k_nn_w1_exp) from two tables built once per step
//    (k_nn_tables_hf): for HARD graphs (theta estimator, score estimator) the packed f16 pieces of every column pair -- the operand is a
//    bit-select with the sampled graph's pair mask, no arithmetic --, for SOFT graphs (reparam estimator) the scaled float, multiplied
//    with g~ and split with four instructions per pair (ahf_split).
//  * samples are taken in Threefry pairs (s, s + S/2): one call per element serves both graphs (the f32 kernel drew every call twice).
// One block = 8 waves (row tile w of x each; N <= 128), one block per CU (image 56 KiB + two graphs 40 / 80 KiB at d = 100).
// grid = (ceil(S / 2 / ppb), Mloc), block = 512, dynamic LDS = nhf_lds_bytes()
// ------------------------------------------------------------------------------------------------
template <int NT>
struct Nhf {
  // (a column tile is KROWS rows of 32 bytes + 32 bytes of padding: the operand is written in element order -- the lanes of one store
  //  instruction cover a whole row a, i.e. all NT tiles -- and tiles 4 096 bytes apart would put them on the same 8 banks: 7-way conflicts)
  static constexpr int NKS = (NT + 1) / 2, KROWS = 32 * NKS, TILE_BYTES = KROWS * 32 + 32, PIECE_BYTES = NT * TILE_BYTES, IMG_BYTES = 2 * PIECE_BYTES;
};
constexpr int NHF_NW = 8, NHF_NTHR = 64 * NHF_NW;
__host__ __device__ inline int nhf_dp2(int d) { return (d + 1) & ~1; }          // row pitch of the pair tables (pairs of columns)
__host__ __device__ inline int nhf_dp4(int d) { return (d + 3) & ~3; }
__host__ __device__ inline size_t nhf_lds_bytes(int d, int NT, int H, bool soft) {
  const size_t nks = (size_t)(NT + 1) / 2, img = 2 * (size_t)NT * ((32 * nks) * 32 + 32);
  const size_t graphs = 2 * ((((size_t)d * (nhf_dp2(d) / 2) * (soft ? 8 : 4)) + 15) & ~(size_t)15);
  const size_t lvt = ((size_t)2 * H + 1) * nhf_dp4(d) * 4;
  return img + graphs + lvt + 64 * 8 + 64;
}

#ifdef DIBS_TU_NN
// ew[m]: exponent that brings max |W1[m]| into [2^13, 2^14) (the f16 pieces of g o W1 then stay below 2^14).  grid = Mloc, block = 1024
// (eight loads in flight per thread: a rolled loop with one load per trip took 97 us for the 51 MB of config 5)
__global__ __launch_bounds__(1024) void k_nn_w1_exp(const float* __restrict__ theta, size_t P, int* __restrict__ ew, int d, int H) {
  __shared__ float red[16];
  const int m = blockIdx.x, tid = threadIdx.x;
  const float* w = theta + (size_t)m * P;
  const int n = d * d * H;
  float mx = 0.f;
  for (int e0 = tid; e0 < n; e0 += 8 * 1024) {
    float v[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) v[q] = e0 + q * 1024 < n ? fabsf(w[e0 + q * 1024]) : 0.f;
#pragma unroll
    for (int q = 0; q < 8; ++q) mx = fmaxf(mx, v[q]);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
  if ((tid & 63) == 0) red[tid >> 6] = mx;
  __syncthreads();
  if (tid == 0) {
    for (int q = 1; q < 16; ++q) mx = fmaxf(mx, red[q]);
    int e = 0;
    if (mx > 0.f && mx < 3.0e38f) e = 13 - ((int)((__float_as_uint(mx) >> 23) & 0xffu) - 127);
    ew[m] = e < -60 ? -60 : (e > 60 ? 60 : e);
  }
}
// w1s[m][h][a][jp] = (W1[2 jp][a][h], W1[2 jp + 1][a][h]) 2^ew  (float2; column d of an odd d: 0)
// w1p[m][h][a][jp] = their packed f16 pieces {h pair, m pair}
// grid = (ceil(d * dp2 / 2 / 256), H, Mloc), block = 256
__global__ __launch_bounds__(256) void k_nn_tables_hf(const float* __restrict__ theta, size_t P, const int* __restrict__ ew, float2* __restrict__ w1s,
                                                      uint2* __restrict__ w1p, int d, int H) {
  const int npr = nhf_dp2(d) >> 1, q = blockIdx.x * 256 + threadIdx.x, h = blockIdx.y, m = blockIdx.z;
  if (q >= d * npr) return;
  const int a = q / npr, j0 = 2 * (q - a * npr);
  const float s = ahf_pow2(ew[m]);
  const float* w = theta + (size_t)m * P;
  const float v0 = w[((size_t)j0 * d + a) * H + h] * s, v1 = j0 + 1 < d ? w[((size_t)(j0 + 1) * d + a) * H + h] * s : 0.f;
  const size_t o = ((size_t)m * H + h) * d * npr + q;
  w1s[o] = make_float2(v0, v1);
  uint32_t ph, pm;
  ahf_split(v0, v1, 1.0f, ph, pm);
  w1p[o] = make_uint2(ph, pm);
}
#endif

template <int NT>
struct NhfFrag {
  abf_u32x4 a[(NT + 1) / 2][2];  // [k-step][piece]
};
template <int NT>
__device__ __forceinline__ void nhf_make_frag(const f32x4 (&v)[NT], float s, NhfFrag<NT>& f) {
#pragma unroll
  for (int tj = 0; tj < NT; ++tj) {
    uint32_t h0, m0, h1, m1;
    ahf_split(v[tj][0], v[tj][1], s, h0, m0);
    ahf_split(v[tj][2], v[tj][3], s, h1, m1);
    const int ks = tj >> 1;
    if (tj & 1) {
      f.a[ks][0].z = h0; f.a[ks][0].w = h1;
      f.a[ks][1].z = m0; f.a[ks][1].w = m1;
    } else {
      f.a[ks][0].x = h0; f.a[ks][0].y = h1;
      f.a[ks][1].x = m0; f.a[ks][1].y = m1;
    }
  }
  if (NT & 1) {  // (the partner tile of the last one does not exist: zero columns of the left operand)
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      f.a[NT >> 1][p].z = 0u;
      f.a[NT >> 1][p].w = 0u;
    }
  }
}
// k-step KS of the column tiles tj .. tj + NU - 1: 4 fragment reads and 3 MFMAs per tile, small terms first, tiles alternating
template <int NT, int KS, int NU>
__device__ __forceinline__ void nhf_tile_step(f32x4 (&acc)[NT], const NhfFrag<NT>& A, const unsigned char* img, int rd_off, int tj) {
  ahf_f16x8 b[NU][2];
#pragma unroll
  for (int u = 0; u < NU; ++u)
#pragma unroll
    for (int p = 0; p < 2; ++p) b[u][p] = ahf_tr_pair(img + rd_off + p * Nhf<NT>::PIECE_BYTES + (tj + u) * Nhf<NT>::TILE_BYTES + KS * 32 * 32);
  const ahf_f16x8 ah = __builtin_bit_cast(ahf_f16x8, A.a[KS][0]), am = __builtin_bit_cast(ahf_f16x8, A.a[KS][1]);
#pragma unroll
  for (int u = 0; u < NU; ++u) {
    if (KS == 0) acc[tj + u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b[u][1], ah, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
    else acc[tj + u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b[u][1], ah, acc[tj + u], 0, 0, 0);
  }
#pragma unroll
  for (int u = 0; u < NU; ++u) acc[tj + u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b[u][0], am, acc[tj + u], 0, 0, 0);
#pragma unroll
  for (int u = 0; u < NU; ++u) acc[tj + u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b[u][0], ah, acc[tj + u], 0, 0, 0);
}
template <int NT, int KS>
struct NhfSteps {
  static __device__ __forceinline__ void run(f32x4 (&acc)[NT], const NhfFrag<NT>& A, const unsigned char* img, int rd_off) {
#pragma unroll
    for (int tj = 0; tj + 1 < NT; tj += 2) nhf_tile_step<NT, KS, 2>(acc, A, img, rd_off, tj);
    if (NT & 1) nhf_tile_step<NT, KS, 1>(acc, A, img, rd_off, NT - 1);
    NhfSteps<NT, KS + 1>::run(acc, A, img, rd_off);
  }
};
template <int NT>
struct NhfSteps<NT, (NT + 1) / 2> {
  static __device__ __forceinline__ void run(f32x4 (&)[NT], const NhfFrag<NT>&, const unsigned char*, int) {}
};

template <int NT, int ACT, bool SOFT>
__global__ __launch_bounds__(NHF_NTHR) void k_nn_logprobs_hf(const float* __restrict__ x, const int32_t* __restrict__ mask, const float* __restrict__ theta,
                                                            size_t P, const float* __restrict__ scores, const uint32_t* __restrict__ thr,
                                                            float* __restrict__ logprobs, Key2 carry, int mode, int m0, int M_global, int Mloc, int d, int N,
                                                            int S, int ppb, float alpha, float tau, int layout, int tiny, NNParams np_, int any_mask,
                                                            const float* __restrict__ ln_tab, const float2* __restrict__ w1s,
                                                            const uint2* __restrict__ w1p, const int* __restrict__ ew) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  typedef Nhf<NT> G;
  unsigned char* const sb = reinterpret_cast<unsigned char*>(smem);
  const int npr = nhf_dp2(d) >> 1, dp4 = nhf_dp4(d), H = np_.H;
  // LDS: image | graphs of the pair (SOFT: float2 per column pair; hard: one 32-bit pair mask) | b1^T, W2^T [H][dp4], b2 [dp4] | red
  unsigned char* const gs0 = sb + G::IMG_BYTES;
  const size_t gbytes = (((size_t)d * npr * (SOFT ? 8 : 4)) + 15) & ~(size_t)15;
  unsigned char* const gs1 = gs0 + gbytes;
  float* const LVT = reinterpret_cast<float*>(gs1 + gbytes);
  double* const red = reinterpret_cast<double*>(LVT + ((size_t)2 * H + 1) * dp4);
  // XCD-aware block order (grid.y = particles rounded up to 8; see k_acyc_bf): all blocks of a particle run on one XCD, whose L2 then
  // holds that particle's tables (200 KB at config 5, re-read by each of its 64 sample pairs) instead of every XCD holding everyone's
  const int Lb = blockIdx.x + gridDim.x * blockIdx.y, p_lo = Lb & 7, tq = Lb >> 3;
  const int bx = tq % (int)gridDim.x, m = (tq / (int)gridDim.x) * 8 + p_lo;
  if (m >= Mloc) return;  // (block-uniform)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g4 = lane >> 4, r = lane & 15;
  const size_t dd = (size_t)d * d;
  const NNOff off = nn_offsets(d, H, np_.bias);
  const float* th_m = theta + (size_t)m * P;
  const int nrt = (N + 15) >> 4;
  // x: row n = 16 wave + r, columns 16 tj + 4 g4 + i -- left-operand fragment and the x of the lane's output elements
  f32x4 xv[NT];
  uint32_t okb = 0u;
  float nvalid = 0.f, xmax = 0.f;
  {
    const int n = 16 * wave + r;
#pragma unroll
    for (int tj = 0; tj < NT; ++tj)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int j = 16 * tj + 4 * g4 + i;
        const bool inb = n < N && j < d;
        const float v = inb ? x[(size_t)n * d + j] : 0.f;
        const bool valid = inb && !(any_mask && mask[(size_t)n * d + j]);
        xv[tj][i] = v;
        okb |= (uint32_t)valid << (tj * 4 + i);
        nvalid += valid ? 1.0f : 0.0f;
        xmax = fmaxf(xmax, fabsf(v));
      }
  }
  static_assert(NT * 4 <= 32, "validity bits of a row tile");
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) xmax = fmaxf(xmax, __shfl_xor(xmax, o, 64));
  float* const redf = reinterpret_cast<float*>(red);
  if (lane == 0) redf[wave] = xmax;
  // padding rows / columns of the image are zero for the whole block; small leaves transposed: b1T[h][j], W2T[h][j], b2[j]
  for (int e = tid; e < G::IMG_BYTES / 16; e += NHF_NTHR) reinterpret_cast<float4*>(sb)[e] = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int e = tid; e < (2 * H + 1) * dp4; e += NHF_NTHR) {
    const int row = e / dp4, j = e - row * dp4;
    float v = 0.f;
    if (j < d) {
      if (row < H) v = np_.bias ? th_m[off.b1 + (size_t)j * H + row] : 0.f;
      else if (row < 2 * H) v = th_m[off.w2 + (size_t)j * H + (row - H)];
      else v = np_.bias ? th_m[off.b2 + j] : 0.f;
    }
    LVT[e] = v;
  }
  float prior_rest = 0.f;  // graph-independent part of the prior: all leaves except the first-layer weights
  for (size_t e = off.b1 + tid; e < off.P; e += NHF_NTHR) prior_rest += lin_logn(th_m[e], 0.f, np_.sig_param);
  __syncthreads();
  {
    float t = 0.f;
#pragma unroll
    for (int w8 = 0; w8 < NHF_NW; ++w8) t = fmaxf(t, redf[w8]);
    xmax = t;
  }
  int ex = 0;  // x 2^ex in [2^13, 2^14)
  if (xmax > 0.f) ex = 13 - ((int)((__float_as_uint(xmax) >> 23) & 0xffu) - 127);
  ex = ex > 60 ? 60 : (ex < -60 ? -60 : ex);
  ex = __builtin_amdgcn_readfirstlane(ex);
  NhfFrag<NT> XA;
  nhf_make_frag<NT>(xv, ahf_pow2(ex), XA);
  const int ewm = __builtin_amdgcn_readfirstlane(ew[m]);
  const float unscale = ahf_pow2(-(ex + ewm));
  const TfKeys tk = tf_keys(lin_mode_key(mode, carry, M_global, m0 + m, layout));
  const uint32_t half = (uint32_t)(((uint64_t)S * dd) >> 1);
  const int hS = S >> 1;
  const float inv2 = 0.5f / np_.obs_noise;
  const float lognorm_x = -0.5f * logf(np_.obs_noise) - 0.918938533204672742f;
  const bool fast = tau == 1.0f;
  const float ulo = tiny ? 1.17549435e-38f : 1.1920929e-07f;
  const int rd_off = (4 * g4 + (r >> 2)) * 32 + (((r & 3) + g4) & 3) * 8;
  const float inv_npr = 1.0f / (float)npr;
  const uint32_t* thr_m = thr + (size_t)m * dd;
  const float* sc_m = scores ? scores + (size_t)m * dd : nullptr;
  const float* ln_m = ln_tab + (size_t)m * dd;
  const int npairs = d * npr;
  // the thread's column pairs q = tid + 512 k (k < NPQ): image byte offset of the pair, computed once per block
  constexpr int NPQ = (16 * NT * 8 * NT + NHF_NTHR - 1) / NHF_NTHR;
  int woff[NPQ];
#pragma unroll
  for (int k = 0; k < NPQ; ++k) {
    const int q = tid + NHF_NTHR * k;
    const int a = (int)(((float)q + 0.5f) * inv_npr), j0 = 2 * (q - a * npr);
    woff[k] = q < npairs ? (j0 >> 4) * G::TILE_BYTES + a * 32 + ((((j0 & 15) >> 2) + (a >> 2)) & 3) * 8 + (j0 & 3) * 2 : -1;
  }
  typedef typename std::conditional<SOFT, float2, uint2>::type TabT;
  const TabT* const tab_m = (SOFT ? reinterpret_cast<const TabT*>(w1s) : reinterpret_cast<const TabT*>(w1p)) + (size_t)m * H * npairs;
  TabT wnext[NPQ];  // table entries of the NEXT hidden unit's operand, requested before the current unit's products
  auto fetch = [&](int h) {
    const TabT* t = tab_m + (size_t)h * npairs;
#pragma unroll
    for (int k = 0; k < NPQ; ++k) {
      const int q = tid + NHF_NTHR * k;
      wnext[k] = t[q < npairs ? q : 0];
    }
  };
  fetch(0);
  __syncthreads();

  for (int c = 0; c < ppb; ++c) {
    const int s0 = bx * ppb + c;
    if (s0 >= hS) break;
    // ---- graphs of the pair (s0, s0 + S/2): one Threefry call per element ----
    float pg[2] = {0.f, 0.f};
    const uint32_t cbase = (uint32_t)((uint64_t)s0 * (uint64_t)dd);
    for (int q = tid; q < npairs; q += NHF_NTHR) {
      const int a = (int)(((float)q + 0.5f) * inv_npr), j0 = 2 * (q - a * npr);  // exact for q < 2^20
      float g0[2] = {0.f, 0.f}, g1[2] = {0.f, 0.f};  // [element of the pair]: sample s0 / s0 + S/2
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int j = j0 + k;
        if (j < d && j != a) {
          const int e = a * d + j;
          uint32_t y0, y1;
          threefry2x32_uk(tk, cbase + (uint32_t)e, cbase + (uint32_t)e + half, y0, y1);
          if constexpr (SOFT) {
            const float as = alpha * sc_m[e];
            if (fast) {  // sigmoid(eps + a), eps = log(u / (1 - u))  ==  u / (u + (1 - u) exp(-a))
              const float ea = expf(-as);
              const float u0 = rng_uniform(y0, ulo, 1.0f), u1 = rng_uniform(y1, ulo, 1.0f);
              g0[k] = u0 * __builtin_amdgcn_rcpf(fmaf(1.0f - u0, ea, u0));
              g1[k] = u1 * __builtin_amdgcn_rcpf(fmaf(1.0f - u1, ea, u1));
            } else {
              g0[k] = 1.0f / (1.0f + expf(-tau * (rng_logistic(y0, tiny) + as)));
              g1[k] = 1.0f / (1.0f + expf(-tau * (rng_logistic(y1, tiny) + as)));
            }
          } else {
            const uint32_t t = thr_m[e];
            g0[k] = (y0 >> 9) < t ? 1.0f : 0.0f;
            g1[k] = (y1 >> 9) < t ? 1.0f : 0.0f;
          }
          const float ln = ln_m[e];
          pg[0] = fmaf(g0[k], ln, pg[0]);
          pg[1] = fmaf(g1[k], ln, pg[1]);
        }
      }
      if constexpr (SOFT) {
        reinterpret_cast<float2*>(gs0)[q] = make_float2(g0[0], g0[1]);
        reinterpret_cast<float2*>(gs1)[q] = make_float2(g1[0], g1[1]);
      } else {
        reinterpret_cast<uint32_t*>(gs0)[q] = (g0[0] != 0.f ? 0xffffu : 0u) | (g0[1] != 0.f ? 0xffff0000u : 0u);
        reinterpret_cast<uint32_t*>(gs1)[q] = (g1[0] != 0.f ? 0xffffu : 0u) | (g1[1] != 0.f ? 0xffff0000u : 0u);
      }
    }
#pragma unroll
    for (int hsel = 0; hsel < 2; ++hsel) {
      const unsigned char* gsel = hsel ? gs1 : gs0;
      f32x4 macc[NT];
#pragma unroll
      for (int tj = 0; tj < NT; ++tj) macc[tj] = f32x4{0.f, 0.f, 0.f, 0.f};
      for (int h = 0; h < H; ++h) {
        __syncthreads();  // graphs written (h = 0) / the image's last reader is done
        // ---- operand image of hidden unit h: (g o W1T_h) 2^ew as two f16 pieces, one 4-byte store per column pair and piece ----
#pragma unroll
        for (int k = 0; k < NPQ; ++k) {
          const int q = tid + NHF_NTHR * k;
          if (woff[k] >= 0) {
            uint32_t ph, pm;
            if constexpr (SOFT) {
              const float2 gg = reinterpret_cast<const float2*>(gsel)[q];
              ahf_split(gg.x * wnext[k].x, gg.y * wnext[k].y, 1.0f, ph, pm);
            } else {
              const uint32_t msk = reinterpret_cast<const uint32_t*>(gsel)[q];
              ph = wnext[k].x & msk;
              pm = wnext[k].y & msk;
            }
            unsigned char* const w0 = sb + woff[k];
            *reinterpret_cast<uint32_t*>(w0) = ph;
            *reinterpret_cast<uint32_t*>(w0 + G::PIECE_BYTES) = pm;
          }
        }
        fetch(h + 1 < H ? h + 1 : 0);  // (the next unit's -- or the next sample's first unit's -- entries travel during the products below)
        __syncthreads();
        if (wave < nrt) {
          f32x4 acc[NT];
          NhfSteps<NT, 0>::run(acc, XA, sb, rd_off);
          const float* b1t = LVT + (size_t)h * dp4;
          const float* w2t = LVT + (size_t)(H + h) * dp4;
#pragma unroll
          for (int tj = 0; tj < NT; ++tj) {
            const int j = 16 * tj + 4 * g4;
            if (j < dp4) {
              const f32x4 b1 = *reinterpret_cast<const f32x4*>(b1t + j), w2 = *reinterpret_cast<const f32x4*>(w2t + j);
#pragma unroll
              for (int i = 0; i < 4; ++i) macc[tj][i] += w2[i] * nn_act(ACT >= 0 ? ACT : np_.act, fmaf(acc[tj][i], unscale, b1[i]));
            }
          }
        }
      }
      float sq = 0.f;
      if (wave < nrt) {
        const float* b2t = LVT + (size_t)2 * H * dp4;
#pragma unroll
        for (int tj = 0; tj < NT; ++tj) {
          const int j = 16 * tj + 4 * g4;
          if (j < dp4) {
            const f32x4 b2 = *reinterpret_cast<const f32x4*>(b2t + j);
#pragma unroll
            for (int i = 0; i < 4; ++i)
              if ((okb >> (tj * 4 + i)) & 1u) {
                const float e = xv[tj][i] - (macc[tj][i] + b2[i]);
                sq = fmaf(e, e, sq);
              }
          }
        }
      }
      const float part = prior_rest + pg[hsel] + nvalid * lognorm_x - inv2 * sq;
      const double tot = wave_sum_d((double)part);
      if (lane == 0) red[hsel * NHF_NW + wave] = tot;
    }
    __syncthreads();
    if (tid < 2) {
      double t = 0.0;
      for (int w8 = 0; w8 < NHF_NW; ++w8) t += red[tid * NHF_NW + w8];
      logprobs[(size_t)m * S + s0 + tid * hS] = (float)t;
    }
  }
}
