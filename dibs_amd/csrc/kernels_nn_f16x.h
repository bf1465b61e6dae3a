// JointDiBS + DenseNonlinearGaussian log-probs on the f16 matrix pipe, per-sample operand in REGISTERS (gfx950)
#pragma once
#include "kernels_nn_f16.h"

// ------------------------------------------------------------------------------------------------
// K-NN-hx  k_nn_logprobs_hf (kernels_nn_f16.h) with the two operands' roles swapped.  There, x's row tiles are the register operand and the
//          per-(sample, hidden unit) operand G_s o W1T_h is an LDS image that all waves rebuild and re-read behind two block barriers per
//          unit -- measured latency-bound: 49 % of the wave cycles parked (profiles/round4_cfg5_nn_hf_pmc.txt).  Here the transposed product
//              pre_h^T [d, N] = (G_s o W1T_h)^T [d, d] * x^T [d, N]
//          is evaluated: x^T is the LDS image -- built ONCE per block, read-only afterwards -- and every wave owns a row tile of NODES j
//          and builds its own left-operand fragments of (G_s o W1T_h)^T in registers from per-lane loads: no image build, no LDS writes
//          and NO block barrier per hidden unit (one per sample pair, for the graphs and the log-prob sum); the waves drift apart and one
//          wave's operand build runs beside another's products.
//  * lane (g, r) of wave w: node j = 16 w + r; its operand values are a = 16 tj + 4 g + i (tj < NT, i < 4): for HARD graphs one 16-byte
//    load per tile of the packed f16 pieces of W1[j][a .. a+3][h] 2^ew (table w1q_p [h][a / 4][j]), AND-ed with pair masks made once per
//    sample from the node's row of the bit-packed graph; for SOFT graphs one 16-byte load of the scaled floats (w1q_s), multiplied with the
//    lane's g~ values (registers, read once per sample) and split (ahf_split).  The next unit's entries are requested before the products.
//  * the result lane (g, r) holds pre[j = 16 w + r][n = 16 tn + 4 g + i]: b1 / W2 of the epilogue are two scalars per lane and unit, the
//    x of the residual 4 NTN registers per lane.  (The soft-graph instantiation sits at 256 registers with 27 spilled dwords; giving up the
//    x registers or the table prefetch to make room was measured: 3.48 -> 4.73 ms per launch -- the spills are the cheaper evil.)
// Same arithmetic as k_nn_logprobs_hf up to the summation order inside the MFMA (k runs over a in both).
// grid = (ceil(S / 2 / ppb), Mloc rounded up to 8; re-indexed XCD-aware inside), block = 64 NT (one wave per row tile of nodes),
// dynamic LDS = nhx_lds_bytes()
// ------------------------------------------------------------------------------------------------
// column tiles of x^T (observations) of the INSTANTIATION that runs: 7 up to 112 observations, else 8 (the template parameter NTN -- the
// image always has that many tiles, the ones beyond the data stay zero)
__host__ __device__ inline int nhx_ntn(int N) { return N <= 112 ? 7 : 8; }
__host__ __device__ inline size_t nhx_img_bytes(int NT, int N) { return 2 * (size_t)nhx_ntn(N) * ((size_t)32 * ((NT + 1) / 2) * 32 + 32); }
__host__ __device__ inline size_t nhx_graph_bytes(int d, bool soft) {
  // soft: two samples of g~ [a][dp4(j)] floats; hard: two samples of [j][4 words] bits (a < 128)
  return soft ? 2 * (size_t)d * nhf_dp4(d) * 4 : 2 * (size_t)nhf_dp4(d) * 16;
}
__host__ __device__ inline size_t nhx_lds_bytes(int d, int NT, int N, int H, bool soft) {
  return nhx_img_bytes(NT, N) + nhx_graph_bytes(d, soft) + ((size_t)2 * H + 1) * nhf_dp4(d) * 4 + 64 * 8 + 64;
}

#ifdef DIBS_TU_NN
// w1q_s[m][h][aq][j] = W1[j][4 aq .. 4 aq + 3][h] 2^ew  (float4; inputs beyond d: 0)
// w1q_p[m][h][aq][j] = their packed f16 pieces {h(a0,a1), h(a2,a3), m(a0,a1), m(a2,a3)}
// A block takes 16 nodes x 16 inputs (4 quads) of one particle with ALL hidden units through LDS: theta's layout W1[j][a][h] is read in runs of
// 16 H floats per node, the tables are written in runs of 16 nodes (one thread per (quad, node) reading theta directly touched 64 lines per
// load: 110 us per step at config 5).   grid = (ceil(d / 16) nodes, ceil(d / 16) inputs, Mloc), block = 256, dynamic LDS = 256 H floats
__global__ __launch_bounds__(256) void k_nn_tables_hx(const float* __restrict__ theta, size_t P, const int* __restrict__ ew, float4* __restrict__ w1q_s,
                                                      uint4* __restrict__ w1q_p, int d, int H) {
  extern __shared__ __attribute__((aligned(16))) float tl[];  // [16 nodes][16 inputs][H]
  const int naq = (d + 3) >> 2, j0 = blockIdx.x * 16, a0 = blockIdx.y * 16, m = blockIdx.z, tid = threadIdx.x;
  const float s = ahf_pow2(ew[m]);
  const float* w = theta + (size_t)m * P;
  const int run = 16 * H;
  for (int i = tid; i < 16 * run; i += 256) {
    const int jj = i / run, r = i - jj * run, a = a0 + r / H;
    tl[i] = (j0 + jj < d && a < d) ? w[((size_t)(j0 + jj) * d + a0) * H + r] * s : 0.f;
  }
  __syncthreads();
  for (int o = tid; o < H * 4 * 16; o += 256) {  // (h, quad, node): node fastest
    const int jj = o & 15, ql = (o >> 4) & 3, h = o >> 6, aq = (a0 >> 2) + ql, j = j0 + jj;
    if (aq >= naq || j >= d) continue;
    float v[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = tl[jj * run + (4 * ql + i) * H + h];
    const size_t q = ((size_t)m * H + h) * naq * d + (size_t)aq * d + j;
    w1q_s[q] = make_float4(v[0], v[1], v[2], v[3]);
    uint32_t h0, m0, h1, m1;
    ahf_split(v[0], v[1], 1.0f, h0, m0);
    ahf_split(v[2], v[3], 1.0f, h1, m1);
    w1q_p[q] = make_uint4(h0, h1, m0, m1);
  }
}
#endif

template <int NT, int NTN, int ACT, bool SOFT>
__global__ __launch_bounds__(64 * NT) void k_nn_logprobs_hx(const float* __restrict__ x, const int32_t* __restrict__ mask, const float* __restrict__ theta,
                                                          size_t P, const float* __restrict__ scores, const uint32_t* __restrict__ thr,
                                                          float* __restrict__ logprobs, Key2 carry, int mode, int m0, int M_global, int Mloc, int d,
                                                          int N, int S, int ppb, float alpha, float tau, int layout, int tiny, NNParams np_,
                                                          int any_mask, const float* __restrict__ ln_tab, const float4* __restrict__ w1q_s,
                                                          const uint4* __restrict__ w1q_p, const int* __restrict__ ew) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int NKS = (NT + 1) / 2, KROWS = 32 * NKS, TILE_BYTES = KROWS * 32 + 32, PIECE_BYTES = NTN * TILE_BYTES, IMG_BYTES = 2 * PIECE_BYTES,
                NTHR = 64 * NT;
  unsigned char* const sb = reinterpret_cast<unsigned char*>(smem);
  const int dp4 = nhf_dp4(d), H = np_.H, naq = (d + 3) >> 2;
  unsigned char* const gbase = sb + IMG_BYTES;
  const size_t gbytes = nhx_graph_bytes(d, SOFT) / 2;
  float* const LVT = reinterpret_cast<float*>(gbase + 2 * gbytes);  // b1^T, W2^T [H][dp4], b2 [dp4]
  double* const red = reinterpret_cast<double*>(LVT + ((size_t)2 * H + 1) * dp4);
  // XCD-aware block order: all blocks of a particle on one XCD (its tables stay in that L2)
  const int Lb = blockIdx.x + gridDim.x * blockIdx.y, p_lo = Lb & 7, tq = Lb >> 3;
  const int bx = tq % (int)gridDim.x, m = (tq / (int)gridDim.x) * 8 + p_lo;
  if (m >= Mloc) return;  // (block-uniform)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g4 = lane >> 4, r = lane & 15;
  const int j = 16 * wave + r;  // this lane's node
  const bool jok = j < d;
  const size_t dd = (size_t)d * d;
  const NNOff off = nn_offsets(d, H, np_.bias);
  const float* th_m = theta + (size_t)m * P;
  // ---- x^T image (once per block): element (a, n) = x[n][a] 2^ex as two f16 pieces; max |x| first ----
  float xmax = 0.f;
  for (int e = tid; e < N * d; e += NTHR) xmax = fmaxf(xmax, fabsf(x[e]));
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) xmax = fmaxf(xmax, __shfl_xor(xmax, o, 64));
  float* const redf = reinterpret_cast<float*>(red);
  if (lane == 0) redf[wave] = xmax;
  for (int e = tid; e < IMG_BYTES / 16; e += NTHR) reinterpret_cast<float4*>(sb)[e] = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int e = tid; e < (2 * H + 1) * dp4; e += NTHR) {
    const int row = e / dp4, jj = e - row * dp4;
    float v = 0.f;
    if (jj < d) {
      if (row < H) v = np_.bias ? th_m[off.b1 + (size_t)jj * H + row] : 0.f;
      else if (row < 2 * H) v = th_m[off.w2 + (size_t)jj * H + (row - H)];
      else v = np_.bias ? th_m[off.b2 + jj] : 0.f;
    }
    LVT[e] = v;
  }
  float prior_rest = 0.f;  // graph-independent part of the prior: all leaves except the first-layer weights
  for (size_t e = off.b1 + tid; e < off.P; e += NTHR) prior_rest += lin_logn(th_m[e], 0.f, np_.sig_param);
  __syncthreads();
  {
    float t = 0.f;
#pragma unroll
    for (int w8 = 0; w8 < NT; ++w8) t = fmaxf(t, redf[w8]);
    xmax = t;
  }
  int ex = 0;
  if (xmax > 0.f) ex = 13 - ((int)((__float_as_uint(xmax) >> 23) & 0xffu) - 127);
  ex = ex > 60 ? 60 : (ex < -60 ? -60 : ex);
  ex = __builtin_amdgcn_readfirstlane(ex);
  {
    const float sx = ahf_pow2(ex);
    const int npn = (N + 1) >> 1;  // observation pairs (n, n + 1): adjacent 2-byte slots of an image row
    for (int q = tid; q < d * npn; q += NTHR) {
      const int a = q / npn, n0 = 2 * (q - a * npn);
      const float v0 = x[(size_t)n0 * d + a], v1 = n0 + 1 < N ? x[(size_t)(n0 + 1) * d + a] : 0.f;
      uint32_t ph, pm;
      ahf_split(v0, v1, sx, ph, pm);
      unsigned char* const w0 = sb + (n0 >> 4) * TILE_BYTES + a * 32 + ((((n0 & 15) >> 2) + (a >> 2)) & 3) * 8 + (n0 & 3) * 2;
      *reinterpret_cast<uint32_t*>(w0) = ph;
      *reinterpret_cast<uint32_t*>(w0 + PIECE_BYTES) = pm;
    }
  }
  // x of the lane's output elements (node j, observations n = 16 tn + 4 g4 + i) and their validity
  f32x4 xe[NTN];
  uint32_t okb = 0u;
  float nvalid = 0.f;
#pragma unroll
  for (int tn = 0; tn < NTN; ++tn)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int n = 16 * tn + 4 * g4 + i;
      const bool inb = jok && n < N;
      const bool valid = inb && !(any_mask && mask[(size_t)n * d + j]);
      xe[tn][i] = inb ? x[(size_t)n * d + j] : 0.f;
      okb |= (uint32_t)valid << (tn * 4 + i);
      nvalid += valid ? 1.0f : 0.0f;
    }
  static_assert(NTN * 4 <= 32, "validity bits of a node's observations");
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
  const float inv_d = 1.0f / (float)d;
  const uint32_t* thr_m = thr + (size_t)m * dd;
  const float* sc_m = scores ? scores + (size_t)m * dd : nullptr;
  const float* ln_m = ln_tab + (size_t)m * dd;
  typedef typename std::conditional<SOFT, float4, uint4>::type TabT;
  const TabT* const tab_m = (SOFT ? reinterpret_cast<const TabT*>(w1q_s) : reinterpret_cast<const TabT*>(w1q_p)) + (size_t)m * H * naq * d;
  const int jc = jok ? j : 0;
  TabT wnext[NT];  // the lane's table entries of the NEXT hidden unit (a-quads 4 tj + g4)
  auto fetch = [&](int h) {
    const TabT* t = tab_m + (size_t)h * naq * d;
#pragma unroll
    for (int tj = 0; tj < NT; ++tj) {
      const int aq = 4 * tj + g4;
      wnext[tj] = t[(size_t)(aq < naq ? aq : 0) * d + jc];
    }
  };
  fetch(0);
  __syncthreads();

  for (int c = 0; c < ppb; ++c) {
    const int s0 = bx * ppb + c;
    if (s0 >= hS) break;
    // ---- graphs of the pair (s0, s0 + S/2): one Threefry call per element, elements in (a, j) order over all threads ----
    float pg[2] = {0.f, 0.f};
    if (!SOFT) {
      for (int e = tid; e < 2 * dp4 * 4; e += NTHR) reinterpret_cast<uint32_t*>(gbase)[e] = 0u;
      __syncthreads();
    }
    const uint32_t cbase = (uint32_t)((uint64_t)s0 * (uint64_t)dd);
    for (int e = tid; e < (int)dd; e += NTHR) {
      const int a = (int)(((float)e + 0.5f) * inv_d), jj = e - a * d;  // exact for e < 2^20
      float g0 = 0.f, g1 = 0.f;
      if (a != jj) {
        uint32_t y0, y1;
        threefry2x32_uk(tk, cbase + (uint32_t)e, cbase + (uint32_t)e + half, y0, y1);
        if constexpr (SOFT) {
          const float as = alpha * sc_m[e];
          if (fast) {  // sigmoid(eps + a), eps = log(u / (1 - u))  ==  u / (u + (1 - u) exp(-a))
            const float ea = expf(-as);
            const float u0 = rng_uniform(y0, ulo, 1.0f), u1 = rng_uniform(y1, ulo, 1.0f);
            g0 = u0 * __builtin_amdgcn_rcpf(fmaf(1.0f - u0, ea, u0));
            g1 = u1 * __builtin_amdgcn_rcpf(fmaf(1.0f - u1, ea, u1));
          } else {
            g0 = 1.0f / (1.0f + expf(-tau * (rng_logistic(y0, tiny) + as)));
            g1 = 1.0f / (1.0f + expf(-tau * (rng_logistic(y1, tiny) + as)));
          }
        } else {
          const uint32_t t = thr_m[e];
          g0 = (y0 >> 9) < t ? 1.0f : 0.0f;
          g1 = (y1 >> 9) < t ? 1.0f : 0.0f;
        }
        const float ln = ln_m[e];
        pg[0] = fmaf(g0, ln, pg[0]);
        pg[1] = fmaf(g1, ln, pg[1]);
      }
      if constexpr (SOFT) {
        reinterpret_cast<float*>(gbase)[a * dp4 + jj] = g0;
        reinterpret_cast<float*>(gbase + gbytes)[a * dp4 + jj] = g1;
      } else {  // bit a of node jj's row
        if (g0 != 0.f) atomicOr(reinterpret_cast<uint32_t*>(gbase) + jj * 4 + (a >> 5), 1u << (a & 31));
        if (g1 != 0.f) atomicOr(reinterpret_cast<uint32_t*>(gbase + gbytes) + jj * 4 + (a >> 5), 1u << (a & 31));
      }
    }
    __syncthreads();
#pragma unroll
    for (int hsel = 0; hsel < 2; ++hsel) {
      const unsigned char* gsel = gbase + hsel * gbytes;
      // the lane's graph values for node j, inputs a = 16 tj + 4 g4 + i: g~ (soft) or the two pair masks of the quad (hard)
      f32x4 gv[SOFT ? NT : 1];
      uint32_t gm[SOFT ? 1 : NT][2];
      if constexpr (SOFT) {
#pragma unroll
        for (int tj = 0; tj < NT; ++tj)
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int a = 16 * tj + 4 * g4 + i;
            gv[tj][i] = (jok && a < d) ? reinterpret_cast<const float*>(gsel)[a * dp4 + j] : 0.f;
          }
      } else {
        const abf_u32x4 row = *reinterpret_cast<const abf_u32x4*>(gsel + (size_t)jc * 16);
#pragma unroll
        for (int tj = 0; tj < NT; ++tj) {
          const uint32_t word = (tj >> 1) == 0 ? row.x : ((tj >> 1) == 1 ? row.y : ((tj >> 1) == 2 ? row.z : row.w));
          const uint32_t nib = jok ? (word >> (16 * (tj & 1) + 4 * g4)) & 0xfu : 0u;
          gm[tj][0] = ((nib & 1u) ? 0xffffu : 0u) | ((nib & 2u) ? 0xffff0000u : 0u);
          gm[tj][1] = ((nib & 4u) ? 0xffffu : 0u) | ((nib & 8u) ? 0xffff0000u : 0u);
        }
      }
      f32x4 macc[NTN];
#pragma unroll
      for (int tn = 0; tn < NTN; ++tn) macc[tn] = f32x4{0.f, 0.f, 0.f, 0.f};
      for (int h = 0; h < H; ++h) {
        // ---- left operand of unit h in registers: rows of (G o W1T_h)^T for the lane's node ----
        NhfFrag<NT> A;
#pragma unroll
        for (int tj = 0; tj < NT; ++tj) {
          uint32_t h0, m0_, h1, m1;
          if constexpr (SOFT) {
            ahf_split(gv[tj][0] * wnext[tj].x, gv[tj][1] * wnext[tj].y, 1.0f, h0, m0_);
            ahf_split(gv[tj][2] * wnext[tj].z, gv[tj][3] * wnext[tj].w, 1.0f, h1, m1);
          } else {
            h0 = wnext[tj].x & gm[tj][0];
            h1 = wnext[tj].y & gm[tj][1];
            m0_ = wnext[tj].z & gm[tj][0];
            m1 = wnext[tj].w & gm[tj][1];
          }
          const int ks = tj >> 1;
          if (tj & 1) {
            A.a[ks][0].z = h0; A.a[ks][0].w = h1;
            A.a[ks][1].z = m0_; A.a[ks][1].w = m1;
          } else {
            A.a[ks][0].x = h0; A.a[ks][0].y = h1;
            A.a[ks][1].x = m0_; A.a[ks][1].y = m1;
          }
        }
        if (NT & 1) {
#pragma unroll
          for (int p = 0; p < 2; ++p) {
            A.a[NT >> 1][p].z = 0u;
            A.a[NT >> 1][p].w = 0u;
          }
        }
        const float b1 = LVT[(size_t)h * dp4 + jc], w2 = LVT[(size_t)(H + h) * dp4 + jc];
        fetch(h + 1 < H ? h + 1 : 0);  // (the next unit's -- or the next sample's first unit's -- entries travel during the products)
        // ---- products against the x^T image: acc[tn] = pre_h[j][n = 16 tn + 4 g4 + i] 2^(ex + ew) ----
        f32x4 acc[NTN];
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
          const ahf_f16x8 ah = __builtin_bit_cast(ahf_f16x8, A.a[ks][0]), am = __builtin_bit_cast(ahf_f16x8, A.a[ks][1]);
#pragma unroll
          for (int tn = 0; tn < NTN; tn += 2) {
            constexpr int dummy = 0;
            (void)dummy;
            const int nu = tn + 1 < NTN ? 2 : 1;
            ahf_f16x8 b[2][2];
#pragma unroll
            for (int u = 0; u < 2; ++u)
              if (u < nu) {
                b[u][0] = ahf_tr_pair(sb + rd_off + (tn + u) * TILE_BYTES + ks * 32 * 32);
                b[u][1] = ahf_tr_pair(sb + rd_off + PIECE_BYTES + (tn + u) * TILE_BYTES + ks * 32 * 32);
              }
#pragma unroll
            for (int u = 0; u < 2; ++u)
              if (u < nu) {
                if (ks == 0) acc[tn + u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b[u][1], ah, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                else acc[tn + u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b[u][1], ah, acc[tn + u], 0, 0, 0);
              }
#pragma unroll
            for (int u = 0; u < 2; ++u)
              if (u < nu) acc[tn + u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b[u][0], am, acc[tn + u], 0, 0, 0);
#pragma unroll
            for (int u = 0; u < 2; ++u)
              if (u < nu) acc[tn + u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b[u][0], ah, acc[tn + u], 0, 0, 0);
          }
        }
#pragma unroll
        for (int tn = 0; tn < NTN; ++tn)
#pragma unroll
          for (int i = 0; i < 4; ++i) macc[tn][i] += w2 * nn_act(ACT >= 0 ? ACT : np_.act, fmaf(acc[tn][i], unscale, b1));
      }
      float sq = 0.f;
      const float b2 = LVT[(size_t)2 * H * dp4 + jc];
#pragma unroll
      for (int tn = 0; tn < NTN; ++tn)
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if ((okb >> (tn * 4 + i)) & 1u) {
            const float e = xe[tn][i] - (macc[tn][i] + b2);
            sq = fmaf(e, e, sq);
          }
      const float part = prior_rest + pg[hsel] + nvalid * lognorm_x - inv2 * sq;
      const double tot = wave_sum_d((double)part);
      if (lane == 0) red[hsel * 8 + wave] = tot;
    }
    __syncthreads();
    if (tid < 2) {
      double t = 0.0;
      for (int w8 = 0; w8 < NT; ++w8) t += red[tid * 8 + w8];
      logprobs[(size_t)m * S + s0 + tid * hS] = (float)t;
    }
  }
}
