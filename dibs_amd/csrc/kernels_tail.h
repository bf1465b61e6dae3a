// per-particle tail of phase A (gfx950): score-estimator weights -> W_lik, total score-space gradient, back-projection onto Z.
// One launch replaces k_lik_weights_score + k_wtotal + k_zgrad: each of those consumed only the particle's own data and
// was latency-bound (three dependent launches of 9-15 us for ~1 GFLOP between them).
#pragma once
#include "common.h"
#include "kernels_kmat.h"

// ------------------------------------------------------------------------------------------------
// K4+K7  one block of 1024 threads per particle, everything staged in LDS:
//   A  (BGe score estimator only)  l_s = sum_j node score, w = softmax(l) in double, baseline EMA        dibs.py:359-389
//   B  W = W_lik - beta * mean_sa(W_acyc) + W_prior, W_lik = scale * alpha * (sum_s w_s G_s - P) offdiag   dibs.py:604-658,
//      graph.py:93-108 / 182-196
//   C  grad = [W V, W^T U] - z / sigma^2 on v_mfma_f32_16x16x4_f32 (k-ordered chains: bit-reproducible), written with a copy of z into
//      the packed row [z | grad_z | ...]                                                                  (autodiff of dibs.py:179-180)
// Other estimators (joint models, soft-graph BGe) hand W_lik over in global memory (node_scores == nullptr) and use B + C.
// mean_sa(W_acyc) arrives reduced (k_acyc_reduce).  Sums run in a fixed order (samples ascending, j ascending): results do not depend
// on the grid.  With one CU per particle the phases are latency chains, so every global input that does not depend on phase A is
// requested before it.
// grid = Mloc, block = 1024; dynamic LDS = tail_lds_bytes()
// ------------------------------------------------------------------------------------------------
#define TAIL_NT 1024
#define TAIL_EPT 3   // elements of W per thread and pass whose inputs are prefetched
#define TAIL_SU 4    // samples per lane of the barrier-free softmax (S <= 256; more: the block-wide reductions)
struct TailArgs {
  // A: score estimator
  const double* node_scores;  // [Mloc][d][S]  (null: w_lik is an input)
  const uint64_t* masks;      // [Mloc][d][S][W]
  float* logprobs;            // [Mloc][S]
  const float* baseline;
  float* baseline_out;
  double sf_baseline;
  unsigned int* queue_counts;  // BGe queue counters of this step: consumed (stream order), reset here for the next step
  int S, W, stage_cap;         // stage_cap: samples with non-zero weight whose parent sets fit in LDS (more: read through the caches)
  // B
  const float* probs;   // [Mloc][d][d]
  float* w_lik;         // [Mloc][d][d]  output (A) or input
  const float* w_acyc;  // [Mloc][d][d]  mean over the chains
  float alpha, beta;
  int prior_kind;
  float er_c;
  // C
  const float* z;  // [Mloc][d][k][2]
  float* pack;          // row of particle m0 + m: [z | grad_z | ...] (copy_z) or [grad_z | ...] (gradient rows only)
  size_t pack_stride;
  int copy_z;           // 1: packed rows [z | grad_z | ...]; 0: the row starts with grad_z and z is not copied
  int m0, d, k, ldz;
  float inv_sig2;
  unsigned long long* dbg;  // profiling: phase time stamps of block 0 (100 MHz ticks, accumulated in dbg[1..5]); null in production
  float* w_tot;             // large n_vars (W, U, V do not fit in LDS): phases A and B only, W goes to w_tot [Mloc][d][d] and
                            // k_backproject_big does phase C; ldz must be 0 then.  null: everything in this kernel
  // join with the engine's second stream INSIDE the kernel (see tail_join_wait): the acyclicity chain (edge probabilities in the split
  // pipeline, matrix powers, reduction) ends with a one-thread kernel that stores `join_seq` to *join_flag.  null: stream order covers it.
  const unsigned int* join_flag;
  unsigned int join_seq;
  unsigned int* join_err;   // set to 1 when the flag did not arrive within the time-out (the host checks it after the chunk)
  // kt.x != null: blocks blockIdx.x >= n_part are units of the tiled latent kernel matrix (kmat_tile_block, last unit of a tile writes the
  // entries) riding along: the matrix needs only Z, the next kernel (the SVGD transform) is its first reader, and this launch leaves half
  // of the machine idle for 17 us at 128 particles
  int n_part;
  KmatTile kt;
};

// The main stream does not wait for the second stream with an event (a barrier packet in front of this kernel: +2.5 .. 7 us on the
// critical path even when the awaited kernels finished long ago, scripts/probe/stream_hop.hip); the kernel itself polls a flag that the
// second stream's last kernel stores after w_acyc is complete.  That kernel's end has released its predecessors' stores to memory; the
// consumer reads w_acyc with agent-scope loads (tail_ld_joined: past this XCD's L2, whose lines may be a step old) AFTER the block has met
// the polling thread at a barrier -- no acquire fence: `buffer_inv sc1` drops the XCD's whole L2 under the other blocks' feet (measured:
// the join gap it saves is lost again).  Normally the flag is 14+ us old when the poll happens: one uncached load.  Bounded: a flag that
// never arrives (a launch that failed) ends the wait after ~0.2 s and raises join_err instead of hanging the GPU.
__device__ __forceinline__ void tail_join_wait(const TailArgs& A) {
  if (!A.join_flag) return;
  const unsigned long long t0 = wall_clock64();
  // (sequence numbers wrap: compare as a signed difference)
  while ((int)(__hip_atomic_load(A.join_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - A.join_seq) < 0) {
    __builtin_amdgcn_s_sleep(4);
    if (wall_clock64() - t0 > 20000000ull) {  // 100 MHz ticks
      if (A.join_err) *A.join_err = 1u;
      break;
    }
  }
}
__device__ __forceinline__ float tail_ld_joined(const TailArgs& A, const float* p) {
  return A.join_flag ? __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : *p;
}
// Do kernels of two streams really run side by side here?  Counter collection (rocprofv3 --pmc), AMD_SERIALIZE_KERNEL and debuggers run ONE
// dispatch at a time in submission order: a consumer that polls for a flag of a kernel queued behind it would then spin until its time-out.
// The engine asks once per process: k_probe_wait on the main stream (spins up to 2 ms), k_probe_set on the second stream behind it.
__global__ void k_probe_wait(const unsigned int* flag, unsigned int* seen) {
  const unsigned long long t0 = wall_clock64();
  unsigned int v = 0;
  while ((v = __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == 0u && wall_clock64() - t0 < 200000ull) __builtin_amdgcn_s_sleep(8);
  *seen = v;
}
__global__ void k_probe_set(unsigned int* flag) { __hip_atomic_store(flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// one wave, FIRST on the second stream of a step (fork without an event): waits until k_edge_scores_p's last block has published `seq`,
// then `delay` more ticks (100 MHz; DIBS_FORK_DELAY, default 0: measured, no gain once the stream has the lowest priority).  One polling
// wave leaves the machine to the kernels it waits for; the kernels behind it start with the usual acquire.  Bounded like tail_join_wait.
__global__ void k_wait_flag(const unsigned int* flag, unsigned int seq, unsigned int delay, unsigned int* err) {
  if (threadIdx.x != 0) return;
  const unsigned long long t0 = wall_clock64();
  while ((int)(__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - seq) < 0) {
    __builtin_amdgcn_s_sleep(2);
    if (wall_clock64() - t0 > 20000000ull) {
      if (err) *err = 2u;
      break;
    }
  }
  const unsigned long long t1 = wall_clock64();
  while (wall_clock64() - t1 < delay) __builtin_amdgcn_s_sleep(1);
}
// one thread, last on the second stream
// (RELAXED: the end of the kernel in front of this one has already released its stores; a release here would write the L2 back once more --
//  the one-thread kernel took 4.4 us with it)
__global__ void k_join_flag(unsigned int* flag, unsigned int seq) { __hip_atomic_store(flag, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__host__ __device__ inline int tail_nsplit(int S, int d) {
  int n = TAIL_NT / (S > 0 ? S : 1);
  n = n > 8 ? 8 : n;
  n = n > d ? d : n;
  return n < 1 ? 1 : n;
}
// LDS: [ W dp16 x (dp16 + 2) f32  UNION  phase-A scratch: lp S f64, lp2 nsplit*S f64, wt S f32 ] [U, V: kp4 x ldz f32 each] [colsum d f32]
//      [nzw S f32] [nzi S i32] [staged parent sets cap*d*W u64]
__host__ __device__ inline int tail_ldw(int d) { return ((d + 15) & ~15) + 2; }  // == 2 (mod 4): conflict-free MFMA A-operand reads
// (ldz == 0: the large-n_vars mode -- no W, U, V in LDS)
__host__ __device__ inline size_t tail_union_bytes(int d, int S, bool lik, bool big = false) {
  const size_t w = big ? 0 : (size_t)((d + 15) & ~15) * tail_ldw(d) * 4, a = lik ? (size_t)S * 8 * (1 + tail_nsplit(S, d)) + (size_t)S * 4 : 0;
  return ((w > a ? w : a) + 15) & ~(size_t)15;
}
__host__ __device__ inline size_t tail_fixed_bytes(int d, int ldz, int S, bool lik) {
  return (tail_union_bytes(d, S, lik, ldz == 0) + ((size_t)2 * ((d + 3) & ~3) * ldz + d) * 4 + (lik ? (size_t)S * 8 : 0) + 15) & ~(size_t)15;
}
// row stride of the U / V images: >= k rounded up to the 16-column tiles; == 16 (mod 32) keeps the B-operand reads conflict-free
__host__ inline int tail_ldz(int d, int k, int S, bool lik, size_t lds_limit) {
  const int kq = (k + 15) & ~15, want = (kq & 16) ? kq : kq + 16;
  return tail_fixed_bytes(d, want, S, lik) <= lds_limit ? want : kq;
}
__host__ inline int tail_stage_cap(int d, int ldz, int S, int W, size_t lds_limit) {
  const size_t fixed = tail_fixed_bytes(d, ldz, S, true);
  if (fixed >= lds_limit) return 0;
  const size_t c = (lds_limit - fixed) / ((size_t)d * W * 8);
  return (int)(c > (size_t)S ? (size_t)S : c);
}
__host__ inline size_t tail_lds_bytes(int d, int ldz, int S, int W, bool lik, int stage_cap) {
  return tail_fixed_bytes(d, ldz, S, lik) + (lik ? (size_t)stage_cap * d * W * 8 : 0);
}

__global__ __launch_bounds__(TAIL_NT) void k_particle_grad(TailArgs A) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  if (A.kt.x != nullptr && (int)blockIdx.x >= A.n_part) {  // (block-uniform)
    kmat_tile_block(reinterpret_cast<float*>(smem_raw), A.kt, (int)blockIdx.x - A.n_part, (int)gridDim.x - A.n_part, (int)threadIdx.x);
    return;
  }
  const int m = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int d = A.d, k = A.k, S = A.S, W = A.W, ldz = A.ldz;
  const int dd = d * d, dp16 = (d + 15) & ~15, kp4 = (d + 3) & ~3, ldw = tail_ldw(d);
  const bool lik = A.node_scores != nullptr;
  const int nsplit = tail_nsplit(S, d);
  float* Wm = reinterpret_cast<float*>(smem_raw);    // phase B / C ...
  double* lp = reinterpret_cast<double*>(smem_raw);  // ... phase A scratch in the same bytes
  double* lp2 = lp + S;
  float* wt = reinterpret_cast<float*>(lp2 + (size_t)nsplit * S);
  const bool big = A.w_tot != nullptr;  // (ldz == 0)
  float* Us = reinterpret_cast<float*>(smem_raw + tail_union_bytes(d, S, lik, big));
  float* Vs = Us + (size_t)kp4 * ldz;
  float* cs = Vs + (size_t)kp4 * ldz;
  float* nzw = cs + d;
  int* nzi = reinterpret_cast<int*>(nzw + S);
  uint64_t* mk = reinterpret_cast<uint64_t*>(smem_raw + tail_fixed_bytes(d, ldz, S, true));
  __shared__ double red[2 * (TAIL_NT / 64)];
  __shared__ int nnz_s;
  if (A.join_flag && tid == 0) tail_join_wait(A);  // (the block meets this thread at its first barrier; w_acyc is read behind it)

  // ---- requests that do not depend on phase A: Z (-> U / V images), this thread's first elements of mean(W_acyc), the edge
  // probabilities (and W_lik of the other estimators), the soft in-degrees of the scale-free prior
  const float* pm = A.probs + (size_t)m * dd;
  const float* wag = A.w_acyc + (size_t)m * dd;
  float* wlg = A.w_lik + (size_t)m * dd;
  float pa[TAIL_EPT], pp[TAIL_EPT], pw[TAIL_EPT];
#pragma unroll
  for (int u = 0; u < TAIL_EPT; ++u) {
    const int e = u * TAIL_NT + tid;
    pa[u] = (e < dd && !A.join_flag) ? wag[e] : 0.f;
    pp[u] = e < dd ? pm[e] : 0.f;
    pw[u] = (e < dd && !lik) ? wlg[e] : 0.f;
  }
  auto load_pa_joined = [&]() {  // (join_flag: called right behind the first block barrier)
    if (A.join_flag) {
#pragma unroll
      for (int u = 0; u < TAIL_EPT; ++u) {
        const int e = u * TAIL_NT + tid;
        pa[u] = e < dd ? tail_ld_joined(A, wag + e) : 0.f;
      }
    }
  };
  // Z and (score estimator) the first batch of this thread's node scores are requested TOGETHER, before anything is stored to LDS: a rolled
  // `load -> LDS store` loop waits for every load on its own, and phase A's loads used to be issued only behind it (two to four dependent
  // trips to the far cache levels at the start of the step's most latency-bound kernel)
  const float2* zm = reinterpret_cast<const float2*>(A.z + (size_t)m * d * k * 2);
  constexpr int ZB = 4;
  const int nz = big ? 0 : d * k;
  float2 zpre[ZB];
#pragma unroll
  for (int u = 0; u < ZB; ++u) {
    const int e = u * TAIL_NT + tid;
    zpre[u] = zm[e < nz ? e : 0];
  }
  const int nsplit_ = tail_nsplit(S, d), jw_ = (d + nsplit_ - 1) / nsplit_;
  double nspre[8];
  {
    const bool has = lik && tid < nsplit_ * S;
    const int part = has ? tid / S : 0, s_ = tid - part * S, j0 = part * jw_, j1 = (j0 + jw_ < d) ? j0 + jw_ : d;
    const double* nsm = A.node_scores + (size_t)m * d * S;
#pragma unroll
    for (int u = 0; u < 8; ++u) nspre[u] = (has && j0 + u < j1) ? nsm[(size_t)(j0 + u) * S + s_] : 0.0;
  }
#pragma unroll
  for (int u = 0; u < ZB; ++u) {
    const int e = u * TAIL_NT + tid;
    if (e < nz) {
      const int j = e / k, q = e - j * k;
      Us[j * ldz + q] = zpre[u].x;
      Vs[j * ldz + q] = zpre[u].y;
    }
  }
  for (int e0 = ZB * TAIL_NT; e0 < nz; e0 += ZB * TAIL_NT) {  // (d k > 4096)
    float2 zz[ZB];
#pragma unroll
    for (int u = 0; u < ZB; ++u) {
      const int e = e0 + u * TAIL_NT + tid;
      zz[u] = zm[e < nz ? e : 0];
    }
#pragma unroll
    for (int u = 0; u < ZB; ++u) {
      const int e = e0 + u * TAIL_NT + tid;
      if (e < nz) {
        const int j = e / k, q = e - j * k;
        Us[j * ldz + q] = zz[u].x;
        Vs[j * ldz + q] = zz[u].y;
      }
    }
  }
  for (int e = tid; e < (big ? 0 : (kp4 - d) * ldz); e += TAIL_NT) {  // k-step padding rows of the MFMA operands
    Us[d * ldz + e] = 0.f;
    Vs[d * ldz + e] = 0.f;
  }
  if (A.prior_kind == 1)
    for (int j = tid; j < d; j += TAIL_NT) {
      float c = 0.f;
      for (int r = 0; r < d; ++r) c += pm[r * d + j];  // graph.py:182-196
      cs[j] = c;
    }

  unsigned long long ts[6];
  ts[0] = ts[1] = wall_clock64();
  int nnz = 0;
  float scale = 1.0f;
  bool staged = false;
  if (lik) {
    if (A.queue_counts && m == 0 && tid < 16) A.queue_counts[tid] = 0u;  // (16 counters are allocated; BGE_NQ used)
    // ---- A. l_s = sum_j node score: nsplit threads per sample, each a contiguous range of j, loads batched; the partial sums
    // are combined in part order
    const double* nsm = A.node_scores + (size_t)m * d * S;
    const int jw = (d + nsplit - 1) / nsplit;
    for (int idx = tid; idx < nsplit * S; idx += TAIL_NT) {
      const int part = idx / S, s = idx - part * S;
      const int j0 = part * jw, j1 = (j0 + jw < d) ? j0 + jw : d;
      double t = 0.0;
      int j = j0;
      if (idx == tid) {  // the first eight (requested in the prologue), in the same ascending order as the loops below
#pragma unroll
        for (int u = 0; u < 8; ++u) t += (j0 + u < j1) ? nspre[u] : 0.0;
        j = j0 + 8 < j1 ? j0 + 8 : j1;
      }
      for (; j + 8 <= j1; j += 8) {
        double v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = nsm[(size_t)(j + u) * S + s];
#pragma unroll
        for (int u = 0; u < 8; ++u) t += v[u];
      }
      {
        double v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = (j + u < j1) ? nsm[(size_t)(j + u) * S + s] : 0.0;
#pragma unroll
        for (int u = 0; u < 8; ++u) t += (j + u < j1) ? v[u] : 0.0;
      }
      lp2[(size_t)part * S + s] = t;
    }
    __syncthreads();
    load_pa_joined();
    double sm = 0.0;
    if (S <= 64 * TAIL_SU) {
      // softmax over the samples WITHOUT block barriers: every wave does all of it for itself (lane takes samples lane, lane + 64, ...), with
      // wave-level reductions; wave 0 publishes the list of samples with non-zero weight.  (Rounds 2-4: one block-wide reduction per
      // quantity -- seven barriers of sixteen waves and two double-precision exp per sample in a row, 5.2 us of the kernel's 18.7.)
      double l[TAIL_SU], ex[TAIL_SU];
      double mx = -INFINITY;
#pragma unroll
      for (int u = 0; u < TAIL_SU; ++u) {
        const int s_ = lane + 64 * u;
        l[u] = -INFINITY;
        if (s_ < S) {
          double t = lp2[s_];
          for (int p = 1; p < nsplit; ++p) t += lp2[(size_t)p * S + s_];
          l[u] = t;
          if (wave == 0) A.logprobs[(size_t)m * S + s_] = (float)t;
        }
        mx = l[u] > mx ? l[u] : mx;
      }
      mx = wave_max_d(mx);
      double den = 0.0;
#pragma unroll
      for (int u = 0; u < TAIL_SU; ++u) {
        const bool in = lane + 64 * u < S;
        ex[u] = in ? exp(l[u] - mx) : 0.0;
        den += ex[u];
        sm += in ? l[u] : 0.0;
      }
      den = wave_sum_d(den);
      sm = wave_sum_d(sm);
      if (wave == 0) {
        int base = 0;
#pragma unroll
        for (int u = 0; u < TAIL_SU; ++u) {
          if (64 * u < S) {  // (wave-uniform)
            const float w = (float)(ex[u] / den);
            const unsigned long long bal = __ballot(w != 0.f);
            if (w != 0.f) {
              const int pos = base + __popcll(bal & ((1ull << lane) - 1ull));
              nzi[pos] = lane + 64 * u;
              nzw[pos] = w;
            }
            base += __popcll(bal);
          }
        }
        if (lane == 0) nnz_s = base;
      }
      __syncthreads();
    } else {
    for (int s = tid; s < S; s += TAIL_NT) {
      double t = lp2[s];
      for (int p = 1; p < nsplit; ++p) t += lp2[(size_t)p * S + s];
      lp[s] = t;
      A.logprobs[(size_t)m * S + s] = (float)t;
    }
    __syncthreads();
    constexpr int NW = TAIL_NT / 64;
    double mx = -INFINITY;
    for (int s = tid; s < S; s += TAIL_NT) mx = lp[s] > mx ? lp[s] : mx;
    mx = wave_max_d(mx);
    if (lane == 0) red[wave] = mx;
    __syncthreads();
    mx = red[0];
    for (int w = 1; w < NW; ++w) mx = red[w] > mx ? red[w] : mx;
    double den = 0.0;
    for (int s = tid; s < S; s += TAIL_NT) {
      den += exp(lp[s] - mx);
      sm += lp[s];
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
    for (int s = tid; s < S; s += TAIL_NT) wt[s] = (float)(exp(lp[s] - mx) / den);
    __syncthreads();
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
    }  // (S > 64 TAIL_SU)
    ts[1] = wall_clock64();
    nnz = nnz_s;
    const float bold = A.baseline[m];
    scale = A.sf_baseline > 0.0 ? (float)exp(-(double)bold) : 1.0f;
    if (tid == 0) A.baseline_out[m] = (float)(A.sf_baseline * (sm / S) + (1.0 - A.sf_baseline) * (double)bold);
    // parent sets of the visited samples: [q][j][w]
    staged = nnz <= A.stage_cap;
    const uint64_t* mg = A.masks + (size_t)m * d * S * W;  // [j][s][w]
    if (staged) {
      for (int e = tid; e < nnz * d * W; e += TAIL_NT) {
        const int q = e / (d * W), r = e - q * (d * W), j = r / W, w = r - j * W;
        mk[e] = mg[((size_t)j * S + nzi[q]) * W + w];
      }
    }
  }
  __syncthreads();  // (phase A's scratch is dead: W takes its place)
  if (!lik) load_pa_joined();
  ts[2] = wall_clock64();
  for (int e = tid; e < (big ? 0 : dp16 * ldw); e += TAIL_NT) Wm[e] = 0.f;  // padding rows / columns of the MFMA operand
  __syncthreads();

  // ---- B. total score-space gradient of this particle -> LDS
  {
    const uint64_t* mg = A.masks + (size_t)m * d * S * W;
    for (int e0 = 0; e0 < dd; e0 += TAIL_EPT * TAIL_NT) {
      if (e0) {
#pragma unroll
        for (int u = 0; u < TAIL_EPT; ++u) {
          const int e = e0 + u * TAIL_NT + tid;
          pa[u] = e < dd ? tail_ld_joined(A, wag + e) : 0.f;
          pp[u] = e < dd ? pm[e] : 0.f;
          pw[u] = (e < dd && !lik) ? wlg[e] : 0.f;
        }
      }
#pragma unroll
      for (int u = 0; u < TAIL_EPT; ++u) {
        const int e = e0 + u * TAIL_NT + tid;
        if (e >= dd) continue;
        const int i = e / d, j = e - i * d;
        const float p = pp[u];
        float wl = pw[u];
        if (lik) {
          if (i != j) {
            float acc = 0.f;
            const int w = i >> 6;
            const uint64_t bit = 1ull << (i & 63);
            if (staged) {
              for (int qq = 0; qq < nnz; ++qq) acc += (mk[((size_t)qq * d + j) * W + w] & bit) ? nzw[qq] : 0.f;
            } else {
              for (int qq = 0; qq < nnz; ++qq) acc += (mg[((size_t)j * S + nzi[qq]) * W + w] & bit) ? nzw[qq] : 0.f;
            }
            wl = scale * A.alpha * (acc - p);
          }
          wlg[e] = wl;
        }
        float pr = 0.f;
        if (i != j && A.prior_kind != 2) {
          const float dp = A.alpha * p * (1.0f - p);
          pr = A.prior_kind == 0 ? A.er_c * dp : (-3.0f / (1.0f + cs[j])) * dp;
        }
        const float wt_ = wl - A.beta * pa[u] + pr;
        if (big) A.w_tot[(size_t)m * dd + e] = wt_;
        else Wm[i * ldw + j] = wt_;
      }
    }
  }
  __syncthreads();

  ts[3] = wall_clock64();
  if (big) return;  // (phase C: k_backproject_big)
  // ---- C. back-projection: dU = W V, dV = W^T U, one 16 x 16 tile of both per wave and pass
  // MFMA operands: A[row = lane % 16][k = lane / 16], B[k = lane / 16][col = lane % 16], D[row = 4 (lane / 16) + r][col = lane % 16]
  float* prow = A.pack + (size_t)(A.m0 + m) * A.pack_stride;
  float2* pz = reinterpret_cast<float2*>(prow);
  float2* pg = reinterpret_cast<float2*>(prow + (A.copy_z ? (size_t)d * k * 2 : (size_t)0));
  const int ntj = (k + 15) >> 4, ntiles = (dp16 >> 4) * ntj, g = lane >> 4, r = lane & 15;
  for (int t = wave; t < ntiles; t += TAIL_NT / 64) {
    const int ti = t / ntj, tj = t - ti * ntj;
    f32x4 du = {0.f, 0.f, 0.f, 0.f}, dv = {0.f, 0.f, 0.f, 0.f};
    const float* wa = Wm + (ti * 16 + r) * ldw + g;   // W[i = ti*16 + r][j = k0 + g]
    const float* wtr = Wm + g * ldw + ti * 16 + r;    // W[j = k0 + g][i = ti*16 + r]
    const float* vb = Vs + g * ldz + tj * 16 + r;     // V[j = k0 + g][q = tj*16 + r]
    const float* ub = Us + g * ldz + tj * 16 + r;
    for (int k0 = 0; k0 < kp4; k0 += 4) {
      du = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[k0], vb[k0 * ldz], du, 0, 0, 0);
      dv = __builtin_amdgcn_mfma_f32_16x16x4f32(wtr[k0 * ldw], ub[k0 * ldz], dv, 0, 0, 0);
    }
    const int q = tj * 16 + r;
    if (q < k) {
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const int i = ti * 16 + 4 * g + rr;
        if (i < d) {
          const float2 zi = make_float2(Us[i * ldz + q], Vs[i * ldz + q]);
          if (A.copy_z) pz[i * k + q] = zi;
          pg[i * k + q] = make_float2(du[rr] - zi.x * A.inv_sig2, dv[rr] - zi.y * A.inv_sig2);
        }
      }
    }
  }
  ts[4] = wall_clock64();
  if (A.dbg && m == 0 && tid == 0)
    for (int u = 1; u < 5; ++u) atomicAdd(A.dbg + u, ts[u] - ts[u - 1]);
}

// ------------------------------------------------------------------------------------------------
// K7c  phase C of k_particle_grad for sizes whose W, U, V do not fit in one block's LDS: grad = [W V, W^T U] - z / sigma^2 from W in
//      global memory (k_particle_grad with w_tot), one block per (particle, 16 rows of the result, 32 latent columns):
//      LDS = the 16 rows and the 16 columns of W the tile needs + the 32-column slices of U and V; waves 0, 1 take the two column tiles
//      of dU, waves 2, 3 those of dV; the same k-ordered v_mfma_f32_16x16x4_f32 chains as in k_particle_grad.
// grid = (Mloc, ceil(d / 16), ceil(k / 32)), block = 256; dynamic LDS = backproject_big_lds(d)
// ------------------------------------------------------------------------------------------------
#define BPB_LDQ 48  // row stride of the U / V slices: 32 columns + 16 (== 16 mod 32: conflict-free B-operand reads)
__host__ __device__ inline size_t backproject_big_lds(int d) {
  const size_t kp4 = (size_t)((d + 3) & ~3);
  return (16 * (kp4 + 2) + kp4 * 18 + 2 * kp4 * BPB_LDQ) * 4;
}
__global__ __launch_bounds__(256) void k_backproject_big(const float* __restrict__ w_tot, const float* __restrict__ z, float* __restrict__ pack,
                                                         size_t pack_stride, int copy_z, int m0, int d, int k, float inv_sig2) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int m = blockIdx.x, ti = blockIdx.y, q0 = blockIdx.z * 32;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, r = lane & 15;
  const int kp4 = (d + 3) & ~3, ldr = kp4 + 2;
  float* Wr = smem;                      // [16][ldr]   W[ti*16 + r][j]
  float* Wc = Wr + 16 * ldr;             // [kp4][18]   W[j][ti*16 + r]
  float* Us = Wc + (size_t)kp4 * 18;     // [kp4][BPB_LDQ]
  float* Vs = Us + (size_t)kp4 * BPB_LDQ;
  const float* wm = w_tot + (size_t)m * d * d;
  for (int e = tid; e < 16 * kp4; e += 256) {
    const int rr = e / kp4, j = e - rr * kp4, i = ti * 16 + rr;
    Wr[rr * ldr + j] = (i < d && j < d) ? wm[(size_t)i * d + j] : 0.f;
  }
  for (int e = tid; e < kp4 * 16; e += 256) {
    const int j = e >> 4, rr = e & 15, i = ti * 16 + rr;
    Wc[j * 18 + rr] = (i < d && j < d) ? wm[(size_t)j * d + i] : 0.f;
  }
  const float2* zm = reinterpret_cast<const float2*>(z + (size_t)m * d * k * 2);
  for (int e = tid; e < kp4 * 32; e += 256) {
    const int j = e >> 5, q = e & 31;
    float2 uv = make_float2(0.f, 0.f);
    if (j < d && q0 + q < k) uv = zm[(size_t)j * k + q0 + q];
    Us[j * BPB_LDQ + q] = uv.x;
    Vs[j * BPB_LDQ + q] = uv.y;
  }
  __syncthreads();
  const int tj = wave & 1, kind = wave >> 1;  // kind 0: dU = W V, 1: dV = W^T U
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  if (kind == 0) {
    const float* wa = Wr + r * ldr + g;
    const float* vb = Vs + g * BPB_LDQ + tj * 16 + r;
    for (int k0 = 0; k0 < kp4; k0 += 4) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[k0], vb[k0 * BPB_LDQ], acc, 0, 0, 0);
  } else {
    const float* wtr = Wc + g * 18 + r;
    const float* ub = Us + g * BPB_LDQ + tj * 16 + r;
    for (int k0 = 0; k0 < kp4; k0 += 4) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wtr[k0 * 18], ub[k0 * BPB_LDQ], acc, 0, 0, 0);
  }
  float* prow = pack + (size_t)(m0 + m) * pack_stride;
  float* pz = prow;
  float* pg = prow + (copy_z ? (size_t)d * k * 2 : (size_t)0);
  const int q = q0 + tj * 16 + r;
  if (q < k) {
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      const int i = ti * 16 + 4 * g + rr;
      if (i < d) {
        const float zi = (kind == 0 ? Us : Vs)[i * BPB_LDQ + tj * 16 + r];  // (row i lies inside the slice: all d rows are staged)
        const size_t o = ((size_t)i * k + q) * 2 + kind;                    // [d][k][2]: (U, V) interleaved
        if (copy_z) pz[o] = zi;
        pg[o] = acc[rr] - zi * inv_sig2;
      }
    }
  }
}
