// JointDiBS + DenseNonlinearGaussian, GENERAL path: any tuple of hidden layers (stax.serial of Dense / activation pairs), any
// width, any number of observations.   reference: dibs/models/nonlinearGaussian.py:35-81 (net), :155-186 (init), :248-326
// The tuned kernels of kernels_nn.h cover one hidden layer of <= 64 units with x resident in LDS (BASELINE.json's config 5); this
// file is the slower path behind the rest of the constructor's range.  One work item = (node j, observation n): the per-node MLP
// is evaluated on the row x[n] o g[:, j] on the vector ALU, hidden activations live in a global scratch area (L2-resident),
// laid out [k][item] so that the threads of a wave touch consecutive addresses.  No atomics: every output element has one owner
// thread that sums over the observations in order (results are run-to-run reproducible).
// theta row layout (pytree leaf order of the reference): for every Dense layer l: W_l [d][in_l][out_l] | b_l [d][out_l] (with bias).
#pragma once
// (included by kernels_nn.h inside its translation unit, after NNParams / nn_act / the lin_* helpers)

struct NNNet {
  int nl;                                   // Dense layers = hidden layers + 1
  int sizes[DIBS_MAX_HIDDEN_LAYERS + 2];    // d, hidden..., 1
  long woff[DIBS_MAX_HIDDEN_LAYERS + 1], boff[DIBS_MAX_HIDDEN_LAYERS + 1];
  int hoff[DIBS_MAX_HIDDEN_LAYERS + 2];     // offset of layer l's OUTPUT vector inside an item's activation record (l = 0 .. nl-1)
  int hsum, maxh;                           // record length = sum of the output widths, widest output
  long P;
};
__host__ __device__ inline NNNet nn_net(int d, const NNParams& p) {
  NNNet t;
  t.nl = p.n_hidden + 1;
  t.sizes[0] = d;
  long off = 0;
  t.hsum = 0;
  t.maxh = 1;
  for (int l = 0; l < t.nl; ++l) {
    t.sizes[l + 1] = l < p.n_hidden ? p.hidden[l] : 1;
    t.woff[l] = off;
    off += (long)d * t.sizes[l] * t.sizes[l + 1];
    t.boff[l] = off;
    if (p.bias) off += (long)d * t.sizes[l + 1];
    t.hoff[l] = t.hsum;
    t.hsum += t.sizes[l + 1];
    if (t.sizes[l + 1] > t.maxh) t.maxh = t.sizes[l + 1];
  }
  t.P = off;
  return t;
}

// derivative of the activation from its VALUE (relu / leaky relu: fv > 0 <=> pre > 0)
__device__ __forceinline__ float nn_dact_from_value(int a, float fv) {
  switch (a) {
    case 0: return fv > 0.f ? 1.f : 0.f;
    case 1: return 1.f - fv * fv;
    case 2: return fv * (1.f - fv);
    default: return fv > 0.f ? 1.f : 0.01f;
  }
}

// forward pass of node j on observation n.  `rec`: activation record of this item, element k at rec[k * stride]; returns the mean.
__device__ __forceinline__ float nng_forward(const NNNet& net, const NNParams& np_, const float* __restrict__ th_m, const float* __restrict__ x,
                                             const float* GS, int d, int j, int n, float* rec, size_t stride) {
  for (int l = 0; l < net.nl; ++l) {
    const int in = net.sizes[l], out = net.sizes[l + 1];
    const float* W = th_m + net.woff[l] + (size_t)j * in * out;
    const float* B = th_m + net.boff[l] + (size_t)j * out;
    const float* src = l ? rec + (size_t)net.hoff[l - 1] * stride : nullptr;
    float* dst = rec + (size_t)net.hoff[l] * stride;
    for (int o0 = 0; o0 < out; o0 += 8) {
      float acc[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) acc[u] = (np_.bias && o0 + u < out) ? B[o0 + u] : 0.f;
      for (int a = 0; a < in; ++a) {
        const float v = l ? src[(size_t)a * stride] : x[(size_t)n * d + a] * GS[a * d + j];
        if (v != 0.f) {  // (hard graphs: most inputs of the first layer are masked out)
          const float* wr = W + (size_t)a * out + o0;
#pragma unroll
          for (int u = 0; u < 8; ++u)
            if (o0 + u < out) acc[u] = fmaf(v, wr[u], acc[u]);
        }
      }
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (o0 + u < out) dst[(size_t)(o0 + u) * stride] = l < net.nl - 1 ? nn_act(np_.act, acc[u]) : acc[u];
    }
  }
  return rec[(size_t)net.hoff[net.nl - 1] * stride];
}

// ------------------------------------------------------------------------------------------------
// log p(theta, D | G_s) of one sample per block.  grid = (S, Mloc) [mode GIVEN: (1, n graphs)], block = 256
// dynamic LDS = d * d * 4 + 64;  scratch: gridDim.x * gridDim.y * 256 * hsum floats
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_nng_logprobs(const float* __restrict__ x, const int32_t* __restrict__ mask,
                                                      const float* __restrict__ theta, const float* __restrict__ scores,
                                                      const uint32_t* __restrict__ thr, float* __restrict__ logprobs, Key2 carry, int mode,
                                                      int m0, int M_global, int d, int N, int S, float alpha, float tau, int layout,
                                                      int tiny, NNParams np_, int any_mask, float* __restrict__ scratch, int n_m,
                                                      float* __restrict__ gs_glob) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  // (gs_glob != null, n_vars > 198: the sampled graph of this block in global scratch [gridDim.x][d*d] instead of LDS; the block's own
  //  barriers order its writes and reads)
  float* GS = gs_glob ? gs_glob + (size_t)blockIdx.x * d * d : smem;
  double* red = reinterpret_cast<double*>(gs_glob ? smem : smem + (((size_t)d * d + 3) & ~(size_t)3));
  const NNNet net = nn_net(d, np_);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const size_t dd = (size_t)d * d;
  const uint64_t nbits = (uint64_t)S * dd;
  // persistent blocks: the activation records (256 * hsum floats per block) are sized by the grid, not by S * n_m
  float* rec = scratch + (size_t)blockIdx.x * 256 * (size_t)net.hsum + tid;
  for (long wi = blockIdx.x; wi < (long)S * n_m; wi += gridDim.x) {
    const int m = (int)(wi / S), s = (int)(wi - (long)m * S);
    const float* th_m = theta + (size_t)m * net.P;
    const Key2 key = (mode == LIN_MODE_GIVEN) ? Key2{0, 0} : lin_mode_key(mode, carry, M_global, m0 + m, layout);
    const uint32_t* thr_m = thr + (size_t)m * dd;
    const float* sc_m = scores ? scores + (size_t)m * dd : nullptr;
    for (int e = tid; e < (int)dd; e += 256) {
      const int a = e / d, j = e - a * d;
      GS[e] = lin_sample_g(mode, key, nbits, dd, s, a, j, d, thr_m, sc_m, alpha, tau, layout, tiny);
    }
    __syncthreads();
    // prior: every leaf N(0, sig_param); first-layer weights weighted by g[a][j]   (nonlinearGaussian.py:260-272)
    float part = 0.f;
    for (int l = 0; l < net.nl; ++l) {
      const int in = net.sizes[l], out = net.sizes[l + 1];
      const long nw = (long)d * in * out;
      for (long e = tid; e < nw; e += 256) {
        const float lw = lin_logn(th_m[net.woff[l] + e], 0.f, np_.sig_param);
        if (l == 0) {
          const int a = (int)((e / out) % in), j = (int)(e / ((long)in * out));
          part = fmaf(GS[a * d + j], lw, part);
        } else {
          part += lw;
        }
      }
      if (np_.bias)
        for (long e = tid; e < (long)d * out; e += 256) part += lin_logn(th_m[net.boff[l] + e], 0.f, np_.sig_param);
    }
    // likelihood
    const float inv2 = 0.5f / np_.obs_noise;
    const float lognorm_x = -0.5f * logf(np_.obs_noise) - 0.918938533204672742f;
    for (int it = tid; it < d * N; it += 256) {
      const int j = it / N, n = it - j * N;
      if (any_mask && mask[(size_t)n * d + j]) continue;
      const float mean = nng_forward(net, np_, th_m, x, GS, d, j, n, rec, 256);
      const float e = x[(size_t)n * d + j] - mean;
      part += lognorm_x - inv2 * e * e;
    }
    const double tot = wave_sum_d((double)part);
    if (lane == 0) red[wave] = tot;
    __syncthreads();
    if (tid == 0) logprobs[(size_t)m * S + s] = (float)(red[0] + red[1] + red[2] + red[3]);
    __syncthreads();  // GS / red are rewritten by the next work item
  }
}

// ------------------------------------------------------------------------------------------------
// softmax-weighted gradients, same contract as k_nn_grad (kernels_nn.h) and the same sharing of a particle's weighted samples between blocks
// (GradSplit, kernels_joint.h).  grid = (Mloc, shares), block = 256; block (x, y) takes share y of particle (x + y) mod Mloc
// dynamic LDS = d * d * 4 + 128;  scratch: Mloc * shares * 2 * hsum * d * N floats (activations | pre-activation gradients, per block)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_nng_grad(const float* __restrict__ x, const int32_t* __restrict__ mask,
                                                  const float* __restrict__ theta, const float* __restrict__ scores,
                                                  const uint32_t* __restrict__ thr, const float* __restrict__ logprobs,
                                                  float* __restrict__ out, size_t out_stride, float* __restrict__ theta_copy,
                                                  const float* __restrict__ baseline, float* __restrict__ baseline_out, Key2 carry, int mode,
                                                  int m0, int M_global, int d, int N, int S, float alpha, float tau, int layout, int tiny,
                                                  NNParams np_, double sf_baseline, int any_mask, float* __restrict__ scratch,
                                                  float* __restrict__ gs_glob, GradSplit gs) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const size_t blk = (size_t)blockIdx.y * gridDim.x + blockIdx.x;
  float* GS = gs_glob ? gs_glob + blk * d * d : smem;  // (see k_nng_logprobs)
  double* red = reinterpret_cast<double*>(gs_glob ? smem : smem + (((size_t)d * d + 3) & ~(size_t)3));
  const NNNet net = nn_net(d, np_);
  const int m = (int)((blockIdx.x + blockIdx.y) % gridDim.x), tid = threadIdx.x;
  const size_t dd = (size_t)d * d, NI = (size_t)d * N;
  const float* th_m = theta + (size_t)m * net.P;
  float* const om_final = out + (size_t)m * out_stride;
  float* ACT = scratch + blk * 2 * net.hsum * NI;        // [hsum][NI] layer outputs (this block's)
  float* DPR = ACT + (size_t)net.hsum * NI;              // [hsum][NI] d log p / d pre-activation
  const size_t n_out = mode == LIN_MODE_THETA ? (size_t)net.P : dd;
  const Key2 key = lin_mode_key(mode, carry, M_global, m0 + m, layout);
  const uint64_t nbits = (uint64_t)S * dd;
  const float* lp = logprobs + (size_t)m * S;
  __shared__ float wch[GRAD_WCH];
  __shared__ int last_flag;
  double mx, den, sm;
  int nnz;
  grad_softmax_stats(lp, S, red, mx, den, sm, nnz);
  const int bz = blockIdx.y, NS = gridDim.y, nact = nnz < NS ? (nnz > 0 ? nnz : 1) : NS;
  if (bz >= nact) return;  // (block-uniform: no share)
  float* const om = nact > 1 ? gs.part + ((size_t)m * NS + bz) * gs.stride : om_final;
  for (size_t e = tid; e < n_out; e += 256) om[e] = 0.f;
  const float inv_on = 1.0f / np_.obs_noise, inv_sp2 = 1.0f / (np_.sig_param * np_.sig_param);
  const float* sc_m = scores + (size_t)m * dd;
  const uint32_t* thr_m = thr + (size_t)m * dd;

  int q = 0;  // ordinal of the next weighted sample
  for (int s0 = 0; s0 < S; s0 += GRAD_WCH) {
    __syncthreads();
    if (s0 + tid < S) wch[tid] = (float)(exp((double)lp[s0 + tid] - mx) / den);
    __syncthreads();
  for (int s = s0; s < S && s < s0 + GRAD_WCH; ++s) {
    const float w = wch[s - s0];
    if (w < GRAD_W_MIN) continue;  // block-uniform
    if ((q++ % NS) != bz) continue;  // (another block's sample)
    __syncthreads();
    for (int e = tid; e < (int)dd; e += 256) {
      const int a = e / d, j = e - a * d;
      GS[e] = lin_sample_g(mode, key, nbits, dd, s, a, j, d, thr_m, sc_m, alpha, tau, layout, tiny);
    }
    __syncthreads();
    if (mode == LIN_MODE_Z_SCORE) {
      for (int e = tid; e < (int)dd; e += 256) om[e] += w * GS[e];
      continue;
    }
    // ---- phase 1: forward + backward of every (node, observation) item; records go to the scratch area ----
    for (size_t it = tid; it < NI; it += 256) {
      const int j = (int)(it / N), n = (int)(it - (size_t)j * N);
      float* arec = ACT + it;
      float* drec = DPR + it;
      const float mean = nng_forward(net, np_, th_m, x, GS, d, j, n, arec, NI);
      const bool live = !(any_mask && mask[(size_t)n * d + j]);
      drec[(size_t)net.hoff[net.nl - 1] * NI] = live ? (x[(size_t)n * d + j] - mean) * inv_on : 0.f;  // d / d mean
      for (int l = net.nl - 1; l >= 1; --l) {  // d / d pre of layer l-1 from layer l
        const int in = net.sizes[l], outw = net.sizes[l + 1];
        const float* W = th_m + net.woff[l] + (size_t)j * in * outw;
        const float* dn = drec + (size_t)net.hoff[l] * NI;
        for (int a = 0; a < in; ++a) {
          float t = 0.f;
          for (int o = 0; o < outw; ++o) t = fmaf(dn[(size_t)o * NI], W[(size_t)a * outw + o], t);
          const float fv = arec[(size_t)(net.hoff[l - 1] + a) * NI];
          drec[(size_t)(net.hoff[l - 1] + a) * NI] = t * nn_dact_from_value(np_.act, fv);
        }
      }
    }
    __syncthreads();  // (one block per particle: block-scope visibility of the records is enough)
    // ---- phase 2: one owner thread per output element, sums over the observations in order ----
    if (mode == LIN_MODE_THETA) {
      for (int l = 0; l < net.nl; ++l) {
        const int in = net.sizes[l], outw = net.sizes[l + 1];
        const long nw = (long)d * in * outw;
        for (long e = tid; e < nw; e += 256) {
          const int o = (int)(e % outw), a = (int)((e / outw) % in), j = (int)(e / ((long)in * outw));
          const float* dp = DPR + (size_t)(net.hoff[l] + o) * NI + (size_t)j * N;
          float t = 0.f;
          if (l == 0) {
            const float gv = GS[a * d + j];
            if (gv != 0.f) {
              for (int n = 0; n < N; ++n) t = fmaf(dp[n], x[(size_t)n * d + a], t);
              t = gv * (t - th_m[net.woff[0] + e] * inv_sp2);   // likelihood part and masked prior of the first-layer weight
            }
          } else {
            const float* ap = ACT + (size_t)(net.hoff[l - 1] + a) * NI + (size_t)j * N;
            for (int n = 0; n < N; ++n) t = fmaf(dp[n], ap[n], t);
          }
          om[net.woff[l] + e] += w * t;
        }
        if (np_.bias)
          for (long e = tid; e < (long)d * outw; e += 256) {
            const int o = (int)(e % outw), j = (int)(e / outw);
            const float* dp = DPR + (size_t)(net.hoff[l] + o) * NI + (size_t)j * N;
            float t = 0.f;
            for (int n = 0; n < N; ++n) t += dp[n];
            om[net.boff[l] + e] += w * t;
          }
      }
    } else {  // Z_REPARAM: d / d g[a][j], chained through the soft graph
      const int h1 = net.sizes[1];
      for (int e = tid; e < (int)dd; e += 256) {
        const int a = e / d, j = e - a * d;
        if (a == j) continue;
        const float* W = th_m + net.woff[0] + ((size_t)j * d + a) * h1;
        float lw = 0.f;
        for (int o = 0; o < h1; ++o) lw += lin_logn(W[o], 0.f, np_.sig_param);
        float t = 0.f;
        for (int n = 0; n < N; ++n) {
          float u = 0.f;
          for (int o = 0; o < h1; ++o) u = fmaf(DPR[(size_t)o * NI + (size_t)j * N + n], W[o], u);
          t = fmaf(u, x[(size_t)n * d + a], t);
        }
        const float gv = GS[e];
        om[e] += w * (lw + t) * tau * alpha * gv * (1.0f - gv);
      }
    }
  }
  }
  __syncthreads();
  if (nact > 1) {  // (plain read-modify-writes above: release, count this block; the last one adds the rows in block order)
    __threadfence();
    if (!grad_last_block(gs.ctr + m, nact, &last_flag)) return;
    __threadfence();
    const float* const base = gs.part + (size_t)m * NS * gs.stride;
    for (size_t e = tid; e < n_out; e += 256) om_final[e] = grad_part_sum<GRAD_NS, false>(base, gs.stride, e, nact);
    __syncthreads();
  }
  const float bold = baseline ? baseline[m] : 0.f;
  if (mode == LIN_MODE_THETA) {
    // graph-independent prior gradient of the leaves behind the first-layer weights: -theta / sig_p^2 (the weights sum to 1)
    for (long e = net.boff[0] + tid; e < net.P; e += 256) om_final[e] += -th_m[e] * inv_sp2;  // (boff[0] = end of the first-layer weights)
    if (theta_copy)
      for (long e = tid; e < net.P; e += 256) theta_copy[(size_t)m * out_stride + e] = th_m[e];
  } else if (mode == LIN_MODE_Z_SCORE) {
    const float scale = sf_baseline > 0.0 ? (float)exp(-(double)bold) : 1.0f;
    for (int e = tid; e < (int)dd; e += 256) {
      const int i = e / d, j = e - i * d;
      const float p = (float)sigmoid_d((double)__fmul_rn(alpha, sc_m[e]));
      om_final[e] = i == j ? 0.f : scale * alpha * (om_final[e] - p);
    }
  }
  if (mode != LIN_MODE_THETA && baseline_out && tid == 0)
    baseline_out[m] = (mode == LIN_MODE_Z_SCORE) ? (float)(sf_baseline * (sm / S) + (1.0 - sf_baseline) * (double)bold) : bold;
}

// theta init with the stax key discipline for ANY stack (nonlinearGaussian.py:155-186): subkey(m, j) = row m*d+j of split(key, M*d);
// per stax layer (Dense AND activation): rng, layer_rng = split(rng); Dense: k1, k2 = split(layer_rng); W = normal(k1, (in, out)) * sig;
// b = normal(k2, (out,)) * sig (without bias the layer key draws W directly).   one thread per (local particle, node)
__global__ void k_nng_init_theta(float* __restrict__ theta, Key2 key, int m0, int Mloc, int M_global, int d, NNParams np_, int layout) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= Mloc * d) return;
  const int m = t / d, j = t - m * d;
  const NNNet net = nn_net(d, np_);
  float* th = theta + (size_t)m * net.P;
  Key2 rng = rng_split_row(key, (uint32_t)(M_global * d), (uint32_t)((m0 + m) * d + j), layout);
  for (int si = 0; si < 2 * net.nl - 1; ++si) {
    const Key2 lr = rng_split_row(rng, 2u, 1u, layout);
    rng = rng_split_row(rng, 2u, 0u, layout);
    if (si & 1) continue;  // activation: no parameters
    const int l = si >> 1, in = net.sizes[l], outn = net.sizes[l + 1];
    const uint64_t nw = (uint64_t)in * outn;
    float* W = th + net.woff[l] + (size_t)j * in * outn;
    if (np_.bias) {
      const Key2 k1 = rng_split_row(lr, 2u, 0u, layout), k2 = rng_split_row(lr, 2u, 1u, layout);
      for (uint64_t i = 0; i < nw; ++i) W[i] = rng_normal(rng_bits_at(k1, nw, i, layout)) * np_.sig_param;
      float* B = th + net.boff[l] + (size_t)j * outn;
      for (uint64_t i = 0; i < (uint64_t)outn; ++i) B[i] = rng_normal(rng_bits_at(k2, (uint64_t)outn, i, layout)) * np_.sig_param;
    } else {
      for (uint64_t i = 0; i < nw; ++i) W[i] = rng_normal(rng_bits_at(lr, nw, i, layout)) * np_.sig_param;
    }
  }
}
