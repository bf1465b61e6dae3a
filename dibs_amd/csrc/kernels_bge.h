// BGe marginal likelihood of sampled graphs (gfx950): Bernoulli sampling of the parent sets, then one small Cholesky
// factorisation per (particle m, sample s, node j).
//   reference: dibs/inference/dibs.py:102-119 (sample_g), dibs/models/linearGaussian.py:63-118 (BGe node score),
//              dibs/utils/func.py:128-145 (masked slogdet)
//
// For node j with parent set pa (l = |pa|) the reference needs logdet R[pa,pa] and logdet R[pa+j,pa+j] (as slogdet of the
// masked d x d matrix; SURVEY.md 8(a) E1 shows the identity).  With j ordered LAST in a Cholesky factorisation of
// A = R[pa+j, pa+j] the leading l pivots give the first determinant and the last pivot is the Schur complement
//   score = gam(j, l) - 1/2 logdet R[pa,pa] - 1/2 (N + alpha_lambd - d + l + 1) log(schur).
// When the parent set is larger than its complement the SAME two determinants come from the complementary minors of
// Q = R^-1 (Jacobi: det R[A,A] = det R * det Q[V\A, V\A]): factorise Q[C+j, C+j], C = V \ (pa+j), j last, leading pivots
// -> logdet Q[C,C] =: ld, last pivot pi:
//   score = gam(j, l) - 1/2 (logdet R + ld) + 1/2 (N + alpha_lambd - d + l) log(pi).
// So every problem has n = min(l + 1, d - l) <= (d + 1) / 2 rows.
//
// Two launches per step:
//   k_bge_sample  one wave per (m, j): Threefry -> parent-set bit masks of all S samples (bit-exact with the reference's
//                 stream), problems appended to one queue per size tier (n <= 4, 8, ..., 32, larger)
//   k_bge_chol    persistent blocks work the queues off, largest tier first:
//                   n <= 16   one problem per LANE, the whole factor in registers;
//                   n <= 32   one problem per QUAD (4 lanes): lane q owns rows q, q+4, ...; the pivot row is broadcast with
//                             DPP quad_perm inside v_fmac_f32_dpp, so a column step is one instruction per (row block, p);
//                   larger    one problem per wave, factor in LDS (only reached for d > 64).
#pragma once
#include "common.h"
#include "kernels_kmat.h"

// problems interleaved per lane in the register tiers n <= 4 / 8 / 12 (see bge_chol_lane)
#define BGE_NPL0 4
#define BGE_NPL1 2
#define BGE_NPL2 2
#define BGE_NQ 9  // queue tiers: q = (n + 3) / 4 - 1 for n <= 32, 8 for larger problems

struct BgeParams {
  const float* Rp;      // [n_mats, d+1, d+1]  R with a zero row / column d (index d = "no variable": padding rows of a tier)
  const float* Qp;      // [n_mats, d+1, d+1]  R^-1, same layout
  const double* gam;    // [d, d+1]  log_gamma_term(j, l)
  const double* Nj;     // [d]
  const double* ldR;    // [n_mats]  logdet R
  double alpha_lambd;
  int n_mats;
};

// A queue entry carries everything the factorisation needs -- {code = (m * d + j) * S + s, j, parent-set words} -- so that the consumer
// issues ONE independent 16-byte load per problem (d > 64: two) instead of the chain list -> code -> masks[code] -> (code / S) % d.
struct BgeQueues {
  uint4* list;           // [BGE_NQ][cap][bge_entry_u4(W)]  {code, j, w0.lo, w0.hi} (, {w1.lo, w1.hi, w2.lo, w2.hi} (, {w3.lo, w3.hi, 0, 0}))
  unsigned int* counts;  // [BGE_NQ]: zero at creation, reset by the consumer of the node scores after every use
  uint32_t cap;
};

// 16-byte pieces of a queue entry {code, j, w[0 .. W-1]}: 1, 2, 2, 3 for W = 1 .. 4
__host__ __device__ inline int bge_entry_u4(int W) { return (2 + 2 * W + 3) / 4; }
__host__ __device__ inline int bge_rows(int l, int d) { return l + 1 <= d - l ? l + 1 : d - l; }
__host__ __device__ inline int bge_tier(int n) { return n <= 32 ? (n + 3) / 4 - 1 : BGE_NQ - 1; }

// log(x) of a positive float with ~1e-7 ABSOLUTE error: x = 2^e m, m in [sqrt(1/2), sqrt(2)), log m = 2 atanh((m-1)/(m+1)).
// (The Schur complement enters the score with a factor ~ (N + l) / 2: the relative error of logf on a value like log(400)
//  would be amplified to 1e-5 .. 1e-4; a double-precision log costs ~100 instructions per problem.)
__device__ __forceinline__ double bge_log(float x) {
  int e;
  float m = frexpf(x, &e);  // m in [0.5, 1)
  if (m < 0.70710678f) {
    m *= 2.0f;
    e -= 1;
  }
  const float r = (m - 1.0f) / (m + 1.0f), r2 = r * r;
  const float p = fmaf(r2, fmaf(r2, fmaf(r2, fmaf(r2, fmaf(r2, 2.0f / 11.0f, 2.0f / 9.0f), 2.0f / 7.0f), 2.0f / 5.0f), 2.0f / 3.0f), 2.0f);
  return (double)e * 0.6931471805599453 + (double)(r * p);
}

// node score from the factorisation: ld2 = sum of log2 of the leading pivots, last = last pivot (see the file header)
__device__ __forceinline__ double bge_score(const BgeParams& bp, int j, int l, int d, bool comp, float ld2, float last) {
  const double Nn = bp.Nj[j];
  if (!(Nn > 0.0)) return 0.0;  // linearGaussian.py:118
  const double c = Nn + bp.alpha_lambd - d + l;
  const double g = bp.gam[(size_t)j * (d + 1) + l], ld = 0.6931471805599453 * (double)ld2, ll = bge_log(last);
  return comp ? g - 0.5 * (bp.ldR[bp.n_mats > 1 ? j : 0] + ld) + 0.5 * c * ll : g - 0.5 * ld - 0.5 * (c + 1.0) * ll;
}

// the same with the table entries already in registers (k_bge_chol requests them before the factorisation)
__device__ __forceinline__ double bge_score_pre(double Nn, double g, double ldRv, double alpha_lambd, int l, int d, bool comp, float ld2, float last) {
  if (!(Nn > 0.0)) return 0.0;  // linearGaussian.py:118
  const double c = Nn + alpha_lambd - d + l;
  const double ld = 0.6931471805599453 * (double)ld2, ll = bge_log(last);
  return comp ? g - 0.5 * (ldRv + ld) + 0.5 * c * ll : g - 0.5 * ld - 0.5 * (c + 1.0) * ll;
}

// index set of a problem: the parents (or, complement form, the non-parents other than j) as a bit mask over the d variables
__device__ __forceinline__ void bge_index_mask(uint64_t& w0, uint64_t& w1, int j, int d, bool comp) {
  if (!comp) return;
  const uint64_t v0 = d >= 64 ? ~0ull : (1ull << d) - 1ull, v1 = d > 64 ? (d >= 128 ? ~0ull : (1ull << (d - 64)) - 1ull) : 0ull;
  w0 = ~w0 & v0;
  w1 = ~w1 & v1;
  if (j < 64) w0 &= ~(1ull << j);
  else w1 &= ~(1ull << (j - 64));
}

// ------------------------------------------------------------------------------------------------
// K2  sampling + queueing.  grid = (ceil(d / WAVES) [+ kernel-matrix blocks], Mloc), block = 64 * WAVES
//     layouts: masks [Mloc][d][S][W] u64, node_scores [Mloc][d][S] f64
//     SAMPLE = false: the parent sets are given (dibs_score_graphs), only the queueing runs
// ------------------------------------------------------------------------------------------------
__host__ __device__ inline size_t bge_sample_wave_bytes(int d, int S, int W) {
  // masks[S*W] u64 | thr[d] | lim[d] | loc[S]
  return ((size_t)S * W * 8 + (size_t)2 * d * 4 + (size_t)S * 4 + 15) & ~(size_t)15;
}

template <int WAVES, bool SAMPLE>
__global__ __launch_bounds__(64 * WAVES) void k_bge_sample(const uint32_t* __restrict__ thr, uint64_t* __restrict__ masks,
                                                           double* __restrict__ node_scores, BgeParams bp, Key2 carry, int m0,
                                                           int M_global, int d, int S, int W, int layout, BgeQueues qs, KmatFuse kf) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  // (s_setprio 3 for this kernel and k_bge_chol -- the critical path of a step, beside an acyclicity kernel with slack -- was measured:
  //  4 200 vs 4 207 steps/s, no effect)
  if (kf.z && (int)blockIdx.x >= kf.nbx) {  // kernel-matrix role (block-uniform; WAVES == 4): see KmatFuse
    kmat_block(reinterpret_cast<float*>(smem_raw), kf.z, (size_t)kf.len, (size_t)0, kf.len, kf.kout, 0, kf.M, kf.scale, kf.h, 1,
               (int)blockIdx.y, (int)blockIdx.x - kf.nbx);
    return;
  }
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (kf.pub_flag && blockIdx.x == 0 && blockIdx.y == 0 && tid == 0)  // (see KmatFuse: the fork flag of the step)
    __hip_atomic_store(kf.pub_flag, kf.pub_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const int m = blockIdx.y;
  const int j = blockIdx.x * WAVES + wave;
  const bool active = j < d;
  unsigned char* wbase = smem_raw + (size_t)wave * bge_sample_wave_bytes(d, S, W);
  uint64_t* mk = reinterpret_cast<uint64_t*>(wbase);
  uint32_t* thrs = reinterpret_cast<uint32_t*>(mk + (size_t)S * W);
  // lim = 512 * thr:  y < lim  <=>  (y >> 9) < thr.  thr == 2^23 (p == 1.0f) has no 32-bit lim; those rows are forced on.
  uint32_t* lims = thrs + d;
  uint32_t* loc = lims + d;  // per sample: queue tier << 28 | index inside this block's reservation, or ~0
  __shared__ unsigned int blk_cnt[BGE_NQ], blk_base[BGE_NQ];
  if (tid < BGE_NQ) blk_cnt[tid] = 0u;
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
    if (!SAMPLE) {
      const uint64_t* mg = masks + ((size_t)m * d + j) * S * W;
      for (int e = lane; e < S * W; e += 64) mk[e] = mg[e];
    }
    // ---- 1. sample column j of the S graphs -----------------------------------------------------
    // particle key = row (1 + m_global) of split(carry, M+1); subk_ = row 1 of split(particle key)   dibs.py:350-351
    const Key2 kp = rng_split_row_uniform(carry, (uint32_t)M_global + 1u, (uint32_t)(m0 + m) + 1u, layout);
    const Key2 kg = rng_split_row_uniform(kp, 2u, 1u, layout);
    const uint64_t dd = (uint64_t)d * d, nbits = (uint64_t)S * dd;
    if (!SAMPLE) {
    } else if (W > 2) {
      // n_vars > 128: plain word loop, one Threefry call per sampled bit whatever the layout (the fallback behind the constructor's
      // full range, not tuned)
      for (int s = lane; s < S; s += 64)
        for (int w = 0; w < W; ++w) {
          uint64_t a = 0;
          const int i1 = d < 64 * w + 64 ? d : 64 * w + 64;
          for (int i = 64 * w; i < i1; ++i) {
            const uint32_t y = rng_bits_at(kg, nbits, (uint64_t)s * dd + (uint64_t)i * d + j, layout);
            a |= (uint64_t)((y >> 9) < thrs[i]) << (i & 63);
          }
          mk[s * W + w] = a;
        }
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
                // (Skipping the call of a row whose threshold is 0 or 2^23 -- both outputs of a call belong to one row -- was measured and
                //  dropped: the wave-uniform test needs the limits before the call instead of after it, +2 .. 4 % on this loop, and neither
                //  the first steps nor a sparse posterior (p ~ 1e-10: thr = 1, which must still be drawn) have such rows.)
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

    // ---- 2. no parents: closed form (logdet of the empty minor is 0, Schur complement R_jj); the rest is queued by size.
    // Slots are reserved per BLOCK (LDS counters here, one global atomicAdd per tier and block below): same-address global
    // atomics from every wave were the bottleneck while most problems are still queued.
    const float rjj = bp.Rp[(bp.n_mats > 1 ? (size_t)j * (d + 1) * (d + 1) : 0) + (size_t)j * (d + 1) + j];
    const double score_l0 = bge_score(bp, j, 0, d, false, 0.f, rjj);
    double* ns_out = node_scores + ((size_t)m * d + j) * S;
    for (int s0 = 0; s0 < S; s0 += 64) {
      const int s = s0 + lane;
      const bool valid = s < S;
      int l = 0;
      for (int w = 0; w < W; ++w) l += valid ? __popcll(mk[s * W + w]) : 0;
      if (valid && l == 0) ns_out[s] = score_l0;
      // (n_vars > 128: every problem goes to the last tier -- k_bge_chol_wide works only that one)
      const int tier = (valid && l > 0) ? (W > 2 ? BGE_NQ - 1 : bge_tier(bge_rows(l, d))) : -1;
      const unsigned long long lt = (1ull << lane) - 1ull;
      uint32_t myloc = 0xFFFFFFFFu;
      for (int tq = 0; tq < BGE_NQ; ++tq) {
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
  }  // active
  __syncthreads();
  if (tid < BGE_NQ) {
    const unsigned int c = blk_cnt[tid];
    blk_base[tid] = c ? atomicAdd(&qs.counts[tid], c) : 0u;
  }
  __syncthreads();
  if (active)
    for (int s = lane; s < S; s += 64) {
      const uint32_t v = loc[s];
      if (v != 0xFFFFFFFFu) {
        const uint32_t tq = v >> 28;
        uint4* dst = qs.list + ((size_t)tq * qs.cap + blk_base[tq] + (v & 0x0FFFFFFFu)) * bge_entry_u4(W);
        const uint64_t w0 = mk[s * W];
        dst[0] = make_uint4((uint32_t)(((size_t)m * d + j) * S + s), (uint32_t)j, (uint32_t)w0, (uint32_t)(w0 >> 32));
        if (W > 1) {
          const uint64_t w1 = mk[s * W + 1], w2 = W > 2 ? mk[s * W + 2] : 0ull;
          dst[1] = make_uint4((uint32_t)w1, (uint32_t)(w1 >> 32), (uint32_t)w2, (uint32_t)(w2 >> 32));
        }
        if (W > 3) {
          const uint64_t w3 = mk[s * W + 3];
          dst[2] = make_uint4((uint32_t)w3, (uint32_t)(w3 >> 32), 0u, 0u);
        }
      }
    }
}

// ------------------------------------------------------------------------------------------------
// K3  queued factorisations
// ------------------------------------------------------------------------------------------------
// NPL problems per LANE, n <= NMAX: indices and the lower triangle in registers, fully unrolled (right-looking: the updates of
// a column step are independent).  Rows n .. NMAX-1 are padding: their index is d (the zero row / column of Rp) and their
// diagonal is set to 1, so the unrolled code is the same for every lane and the padding pivots are exactly 1 (log2 = 0).
// Small problems are latency-bound at the two waves per SIMD this kernel runs with (index extraction and the pivot chain are
// serial): NPL independent problems per lane are interleaved by the compiler.
// `mat` = offset (floats) of a problem's matrix (R or Q, of node j) from `R`.
template <int NMAX, bool W2, int NPL>
__device__ __forceinline__ void bge_chol_lane(const float* __restrict__ R, const int (&mat)[NPL], int ldr, int d, uint64_t (&w0)[NPL],
                                              uint64_t (&w1)[NPL], const int (&j)[NPL], const int (&li)[NPL] /* index-set size = n - 1 */,
                                              float (&ld2)[NPL], float (&last)[NPL]) {
  int idx[NPL][NMAX];
#pragma unroll
  for (int t = 0; t < NMAX; ++t)
#pragma unroll
    for (int u = 0; u < NPL; ++u) {
      int b = d;
      if (w0[u]) { b = __ffsll((long long)w0[u]) - 1; w0[u] &= w0[u] - 1; }
      else if (W2 && w1[u]) { b = 64 + __ffsll((long long)w1[u]) - 1; w1[u] &= w1[u] - 1; }
      idx[u][t] = (t == li[u]) ? j[u] : b;
    }
  float A[NPL][NMAX][NMAX];
#pragma unroll
  for (int r = 0; r < NMAX; ++r)
#pragma unroll
    for (int u = 0; u < NPL; ++u) {
      const int ro = mat[u] + idx[u][r] * ldr;
#pragma unroll
      for (int c = 0; c <= r; ++c) A[u][r][c] = R[ro + idx[u][c]];
      if (r > li[u]) A[u][r][r] = 1.0f;
    }
  float lsum[NPL], lst[NPL], inv[NPL];
#pragma unroll
  for (int u = 0; u < NPL; ++u) { lsum[u] = 0.f; lst[u] = 1.f; }
#pragma unroll
  for (int k = 0; k < NMAX; ++k) {
#pragma unroll
    for (int u = 0; u < NPL; ++u) {
      const float dk = A[u][k][k];
      lsum[u] += __log2f(dk);
      lst[u] = (k == li[u]) ? dk : lst[u];
      inv[u] = __builtin_amdgcn_rsqf(dk);
    }
#pragma unroll
    for (int r = k + 1; r < NMAX; ++r)
#pragma unroll
      for (int u = 0; u < NPL; ++u) A[u][r][k] *= inv[u];
#pragma unroll
    for (int c = k + 1; c < NMAX; ++c)
#pragma unroll
      for (int r = c; r < NMAX; ++r)
#pragma unroll
        for (int u = 0; u < NPL; ++u) A[u][r][c] = fmaf(-A[u][r][k], A[u][c][k], A[u][r][c]);
  }
#pragma unroll
  for (int u = 0; u < NPL; ++u) {
    ld2[u] = lsum[u] - __log2f(lst[u]);  // leading pivots only (the padding pivots are 1)
    last[u] = lst[u];
  }
}

// acc -= bcast_quad(b, lane C) * own   -- one VALU instruction (DPP quad_perm broadcast of the first source)
__device__ __forceinline__ void fmac_quad(float& acc, const float& b, const float& own, int c) {
  switch (c) {
    case 0: asm volatile("v_fmac_f32_dpp %0, -%1, %2 quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(b), "v"(own)); break;
    case 1: asm volatile("v_fmac_f32_dpp %0, -%1, %2 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(b), "v"(own)); break;
    case 2: asm volatile("v_fmac_f32_dpp %0, -%1, %2 quad_perm:[2,2,2,2] row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(b), "v"(own)); break;
    default: asm volatile("v_fmac_f32_dpp %0, -%1, %2 quad_perm:[3,3,3,3] row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(b), "v"(own)); break;
  }
}
// (s_nop 1: a DPP read needs two wait states after the VALU write of its source; the writer here is the last v_fmac above it)
__device__ __forceinline__ float bcast_quad(const float& v, int c) {
  float o;
  switch (c) {
    case 0: asm volatile("s_nop 1\n\tv_mov_b32_dpp %0, %1 quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf" : "=v"(o) : "v"(v)); break;
    case 1: asm volatile("s_nop 1\n\tv_mov_b32_dpp %0, %1 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf" : "=v"(o) : "v"(v)); break;
    case 2: asm volatile("s_nop 1\n\tv_mov_b32_dpp %0, %1 quad_perm:[2,2,2,2] row_mask:0xf bank_mask:0xf" : "=v"(o) : "v"(v)); break;
    default: asm volatile("s_nop 1\n\tv_mov_b32_dpp %0, %1 quad_perm:[3,3,3,3] row_mask:0xf bank_mask:0xf" : "=v"(o) : "v"(v)); break;
  }
  return o;
}

// column step K of the quad factorisation as a template recursion: every register index is a compile-time constant.
// Right-looking: the pivot column is scaled, then every later column c receives its rank-1 update
//   L[a][c] -= L[a][K] * bcast(L[c/4][K], lane c%4)        (a >= c/4)
// -- all updates of a step are independent (ILP at two waves per SIMD), and only the first group of a step reads registers
// written by the asm statement in front of it.
template <int NB, int K>
struct BgeQuadCol {
  static __device__ __forceinline__ void run(float (&L)[NB][4 * NB], int li, float& lsum, float& lst) {
    constexpr int kb = K >> 2, kq = K & 3, N = 4 * NB;
    const float piv = bcast_quad(L[kb][K], kq);
    lsum += __log2f(piv);
    lst = (K == li) ? piv : lst;
    float inv = __builtin_amdgcn_rsqf(piv);
    // gfx950: a VALU instruction must not read the result of a transcendental op in the next issue slot; hipcc inserts that
    // wait state for its own instructions but not in front of an asm statement
    asm volatile("s_nop 1" : "+v"(inv));
#pragma unroll
    for (int a = kb; a < NB; ++a) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(L[a][K]) : "v"(inv));
    if (K + 1 < N) asm volatile("s_nop 1" ::: "memory");  // the scaled column is read through DPP next
#pragma unroll
    for (int c = K + 1; c < N; ++c)
#pragma unroll
      for (int a = c >> 2; a < NB; ++a) fmac_quad(L[a][c], L[c >> 2][K], L[a][K], c & 3);
    BgeQuadCol<NB, K + 1>::run(L, li, lsum, lst);
  }
};
template <int NB>
struct BgeQuadCol<NB, 4 * NB> {
  static __device__ __forceinline__ void run(float (&)[NB][4 * NB], int, float&, float&) {}
};

// One problem per QUAD, n <= 4 NB.  Lane q of the quad owns rows q, q + 4, ...: L[a][c] is row 4a + q, column c <= 4a + 3
// (entries right of the diagonal are zero-filled scratch).  Column step k (row k lives in lane k % 4, block k / 4):
//   pivot = bcast(L[k/4][k]);  L[a][k] *= rsqrt(pivot)                            (a >= k/4)
//   L[a][c] -= L[a][k] * bcast(L[c/4][k], lane c%4)   for c > k, a >= c/4        (v_fmac_f32_dpp)
// Rows above c inside block c/4 compute garbage that only ever feeds themselves.  `qidx`: this quad's index list in LDS
// (QS ints, 16-byte aligned).
#define BGE_QS 36
template <int NB, bool W2>
__device__ __forceinline__ void bge_chol_quad(const float* __restrict__ R, int mat, int ldr, int d, int* __restrict__ qidx,
                                              uint64_t w0, uint64_t w1, int j, int li, float& ld2, float& last) {
  constexpr int N = 4 * NB;
  const int q = threadIdx.x & 3;
  // ---- index list: padding = d, lane q extracts the set bits of its quarter of the mask, j goes last
#pragma unroll
  for (int t = 0; t < NB; ++t) qidx[4 * t + q] = d;
  wave_lds_fence();
  {
    uint32_t chunk;
    int base_bit, t;
    if (W2) {
      const uint64_t w = q < 2 ? w0 : w1;
      chunk = (uint32_t)(w >> (32 * (q & 1)));
      base_bit = 32 * q;
      t = (q >= 2 ? __popcll(w0) : 0) + ((q & 1) ? __popc((uint32_t)w) : 0);
    } else {
      chunk = (uint32_t)(w0 >> (16 * q)) & 0xFFFFu;
      base_bit = 16 * q;
      t = __popcll(w0 & ((1ull << (16 * q)) - 1ull));
    }
    while (__any(chunk != 0u)) {
      if (chunk) {
        const int b = __ffs((int)chunk) - 1;
        chunk &= chunk - 1u;
        if (t < N) qidx[t] = base_bit + b;
        ++t;
      }
    }
  }
  wave_lds_fence();
  if (q == 0 && li < N) qidx[li] = j;
  wave_lds_fence();
  int cidx[N], ro[NB];
#pragma unroll
  for (int t = 0; t < NB; ++t) {
    const int4 v = *reinterpret_cast<const int4*>(qidx + 4 * t);
    cidx[4 * t] = v.x; cidx[4 * t + 1] = v.y; cidx[4 * t + 2] = v.z; cidx[4 * t + 3] = v.w;
  }
#pragma unroll
  for (int a = 0; a < NB; ++a) ro[a] = mat + qidx[4 * a + q] * ldr;
  // ---- gather
  float L[NB][N];
#pragma unroll
  for (int a = 0; a < NB; ++a) {
#pragma unroll
    for (int c = 0; c < 4 * a + 4; ++c) {
      float v = R[ro[a] + cidx[c]];
      if (c > 4 * a) v = (c - 4 * a <= q) ? v : 0.f;                   // right of the diagonal of this lane's row
      if (c >= 4 * a) v = (c - 4 * a == q && 4 * a + q > li) ? 1.f : v;  // padding row: unit diagonal
      L[a][c] = v;
    }
  }
  // every entry is pinned into its register HERE: hipcc would otherwise sink the selects above to their first use, i.e. directly in
  // front of an asm statement that reads them through DPP (two wait states needed, and it cannot see inside the asm)
#pragma unroll
  for (int a = 0; a < NB; ++a)
#pragma unroll
    for (int c = 0; c < 4 * a + 4; ++c) asm volatile("" : "+v"(L[a][c]));
  asm volatile("s_nop 1" ::: "memory");
  float lsum = 0.f, lst = 1.f;
  BgeQuadCol<NB, 0>::run(L, li, lsum, lst);
  ld2 = lsum - __log2f(lst);
  last = lst;
}

// ---- one problem per PAIR of lanes (round 5), 17 <= n <= 24 ----------------------------------------------------------------------------
// The quad layout repeats everything that is not a multiply-add of the update -- pivot broadcast, log2, rsqrt, the scaling of the pivot
// column, index list, gather addresses, the double-precision score -- on four lanes per problem, and that is 55 % of its instructions
// (N = 24: 686 update FMAs of ~1 520 instructions per lane, 6 080 lane-instructions per problem for n^3 / 6 <= 2 304 useful ones).  With
// TWO lanes per problem -- lane p owns rows p, p + 2, ... -- the update costs 1 222 FMAs per lane (2 444 per problem: the row-cyclic
// triangle wastes 6 % instead of 16 %) and the rest is paid twice instead of four times: ~3 950 lane-instructions per problem.  The pair
// broadcast is the same DPP quad_perm inside v_fmac_f32_dpp ([0,0,2,2] / [1,1,3,3]).  156 registers of matrix at N = 24: the tiers beyond
// (n = 25 .. 32) stay on the quad layout.
__device__ __forceinline__ void fmac_pair(float& acc, const float& b, const float& own, int c) {
  if (c & 1) asm volatile("v_fmac_f32_dpp %0, -%1, %2 quad_perm:[1,1,3,3] row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(b), "v"(own));
  else asm volatile("v_fmac_f32_dpp %0, -%1, %2 quad_perm:[0,0,2,2] row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(b), "v"(own));
}
__device__ __forceinline__ float bcast_pair(const float& v, int c) {
  float o;
  if (c & 1) asm volatile("s_nop 1\n\tv_mov_b32_dpp %0, %1 quad_perm:[1,1,3,3] row_mask:0xf bank_mask:0xf" : "=v"(o) : "v"(v));
  else asm volatile("s_nop 1\n\tv_mov_b32_dpp %0, %1 quad_perm:[0,0,2,2] row_mask:0xf bank_mask:0xf" : "=v"(o) : "v"(v));
  return o;
}
template <int NR, int K>
struct BgePairCol {
  static __device__ __forceinline__ void run(float (&L)[NR][2 * NR], int li, float& lsum, float& lst) {
    constexpr int kb = K >> 1, N = 2 * NR;
    const float piv = bcast_pair(L[kb][K], K);
    lsum += __log2f(piv);
    lst = (K == li) ? piv : lst;
    float inv = __builtin_amdgcn_rsqf(piv);
    asm volatile("s_nop 1" : "+v"(inv));  // (transcendental result -> VALU read inside an asm statement: see BgeQuadCol)
#pragma unroll
    for (int a = kb; a < NR; ++a) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(L[a][K]) : "v"(inv));
    if (K + 1 < N) asm volatile("s_nop 1" ::: "memory");  // the scaled column is read through DPP next
#pragma unroll
    for (int c = K + 1; c < N; ++c)
#pragma unroll
      for (int a = c >> 1; a < NR; ++a) fmac_pair(L[a][c], L[c >> 1][K], L[a][K], c);
    BgePairCol<NR, K + 1>::run(L, li, lsum, lst);
  }
};
template <int NR>
struct BgePairCol<NR, 2 * NR> {
  static __device__ __forceinline__ void run(float (&)[NR][2 * NR], int, float&, float&) {}
};
#define BGE_PS 28  // ints of LDS per pair: index list of up to 24 entries, 16-byte aligned
template <int NR, bool W2>
__device__ __forceinline__ void bge_chol_pair(const float* __restrict__ R, int mat, int ldr, int d, int* __restrict__ pidx, uint64_t w0, uint64_t w1,
                                              int j, int li, float& ld2, float& last) {
  constexpr int N = 2 * NR;
  const int p = threadIdx.x & 1;
  // ---- index list: padding = d, each lane extracts the set bits of its half of the mask, j goes last
#pragma unroll
  for (int t = 0; t < NR; ++t) pidx[2 * t + p] = d;
  wave_lds_fence();
  {
    uint64_t chunk;
    int base_bit, t;
    if (W2) {
      chunk = p ? w1 : w0;
      base_bit = 64 * p;
      t = p ? __popcll(w0) : 0;
    } else {
      chunk = (w0 >> (32 * p)) & 0xFFFFFFFFull;
      base_bit = 32 * p;
      t = p ? __popc((uint32_t)w0) : 0;
    }
    while (__any(chunk != 0ull)) {
      if (chunk) {
        const int b = __ffsll((long long)chunk) - 1;
        chunk &= chunk - 1ull;
        if (t < N) pidx[t] = base_bit + b;
        ++t;
      }
    }
  }
  wave_lds_fence();
  if (p == 0 && li < N) pidx[li] = j;
  wave_lds_fence();
  int ro[NR];
#pragma unroll
  for (int a = 0; a < NR; ++a) ro[a] = mat + pidx[2 * a + p] * ldr;
  // ---- gather, column by column (the column's index is read when it is needed: 24 registers less than holding the list):
  // L[a][c] = row 2 a + p, column c <= 2 a + 1 (the entry right of the diagonal of lane 0's row is zero-filled scratch)
  float L[NR][N];
#pragma unroll
  for (int c4 = 0; c4 < N; c4 += 4) {
    const int4 ci4 = *reinterpret_cast<const int4*>(pidx + c4);
    const int ci[4] = {ci4.x, ci4.y, ci4.z, ci4.w};
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int c = c4 + u;
#pragma unroll
      for (int a = c >> 1; a < NR; ++a) {
        float v = R[ro[a] + ci[u]];
        if (c > 2 * a) v = p ? v : 0.f;                                   // column 2 a + 1: only the odd lane's row reaches it
        if (c >= 2 * a) v = (c - 2 * a == p && 2 * a + p > li) ? 1.f : v;  // padding row: unit diagonal
        L[a][c] = v;
      }
    }
  }
  // (pinned into registers before the first DPP read: see bge_chol_quad)
#pragma unroll
  for (int a = 0; a < NR; ++a)
#pragma unroll
    for (int c = 0; c < 2 * a + 2; ++c) asm volatile("" : "+v"(L[a][c]));
  asm volatile("s_nop 1" ::: "memory");
  float lsum = 0.f, lst = 1.f;
  BgePairCol<NR, 0>::run(L, li, lsum, lst);
  ld2 = lsum - __log2f(lst);
  last = lst;
}

// One problem per WAVE with 33 .. 64 rows, RIGHT-looking with the matrix in registers: lane r holds row r of the lower triangle (64
// VGPRs); column step K scales the pivot column, publishes it in LDS (64 floats) and every lane updates its own row with broadcast reads
// of that column -- a quarter of an LDS read per multiply-add where the left-looking bge_chol_wave needs two, and half a KiB of LDS per
// wave instead of a d x d factor (which limited a block to two waves at d = 128).  Rows n .. 63 are padding (index d = the zero row /
// column of Rp, unit diagonal), so every lane runs the same straight-line code.
// `col`: 64 floats of LDS of this wave (16-byte aligned); `idx`: the problem's index list in LDS (>= 64 ints, padding entries = d).
template <int K>
struct BgeWaveCol {
  static __device__ __forceinline__ void run(float (&A)[64], float* __restrict__ col, int lane, int li, float& lsum, float& lst) {
    // (no branch on K < n: the padding steps are exact no-ops -- pivot 1, column 0 -- and cost (64 - n)^2 / 2 multiply-adds, while a
    //  conditional per step made hipcc spill the 64 row registers)
    const float piv = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(A[K]), K));  // (the builtin is int -> int)
    lsum += K < li ? __log2f(piv) : 0.f;
    lst = (K == li) ? piv : lst;
    const float l = A[K] * __builtin_amdgcn_rsqf(piv);  // L[r][K] for r > K (lanes r <= K: unused)
    col[lane] = l;
    wave_lds_fence();
#pragma unroll
    for (int c4 = (K + 1) & ~3; c4 < 64; c4 += 4) {
      const float4 lc = *reinterpret_cast<const float4*>(col + c4);
      if (c4 + 0 > K) A[c4 + 0] = fmaf(-l, lc.x, A[c4 + 0]);
      if (c4 + 1 > K) A[c4 + 1] = fmaf(-l, lc.y, A[c4 + 1]);
      if (c4 + 2 > K) A[c4 + 2] = fmaf(-l, lc.z, A[c4 + 2]);
      if (c4 + 3 > K) A[c4 + 3] = fmaf(-l, lc.w, A[c4 + 3]);
    }
    wave_lds_fence();  // (the column buffer is rewritten by the next step)
    BgeWaveCol<K + 1>::run(A, col, lane, li, lsum, lst);
  }
};
template <>
struct BgeWaveCol<64> {
  static __device__ __forceinline__ void run(float (&)[64], float*, int, int, float&, float&) {}
};
__device__ __forceinline__ void bge_chol_wave_reg(const float* __restrict__ R, size_t mat, int ldr, int d, float* __restrict__ col,
                                                  const int* __restrict__ idx, int n, int li, float& ld2, float& last) {
  const int lane = threadIdx.x & 63;
  float A[64];
  const size_t ro = mat + (size_t)idx[lane] * ldr;  // (padding lanes: row d, all zeros)
#pragma unroll
  for (int c = 0; c < 64; ++c) A[c] = R[ro + idx[c]];  // (entries right of the diagonal are never used as results)
  if (lane >= n) A[0] = 0.f;
#pragma unroll
  for (int c = 0; c < 64; ++c) A[c] = (c == lane && lane >= n) ? 1.0f : A[c];
  float lsum = 0.f, lst = 1.f;
  BgeWaveCol<0>::run(A, col, lane, li, lsum, lst);
  ld2 = lsum;
  last = lst;
}

// (the left-looking variant with the factor in LDS, for up to 128 rows, lives in k_bge_chol_wide)
__host__ __device__ inline size_t bge_generic_wave_bytes(int d) {
  (void)d;
  return (64 + 64 + 8) * 4;  // k_bge_chol (d <= 128, n <= 64): column buffer of bge_chol_wave_reg + index list
}
// LDS of k_bge_chol:  [R | Q (optional)] [quad-tier index lists: 4 waves x 16 quads x BGE_QS ints] [one-problem-per-wave tier: nwg factors].
// The two scratch regions are disjoint on purpose: the waves of a block walk the work units without block barriers, so one wave may
// already be in the per-wave tier while its neighbour still builds quad index lists.  (Until the randomised test of tests/tools/gpu_fuzz.py
// ran d = 96 / 112 with thousands of queued problems the regions were one array indexed by wave * max(sizes) but sized for nwg < 4 waves:
// the quad lists of waves >= nwg lay beyond the allocation -- NaN node scores at d > 80.)
__host__ __device__ inline size_t bge_quad_bytes() { return (size_t)32 * BGE_PS * 4; }  // (>= 16 * BGE_QS * 4: quad and pair lists share the region)
// waves of a block that can work in the one-problem-per-wave tier (each needs a d x d factor in LDS)
__host__ __device__ inline int bge_generic_waves(int d, bool r_in_lds) {
  const size_t r = r_in_lds ? (((size_t)2 * (d + 1) * (d + 1) * 4 + 15) & ~(size_t)15) : 0;
  const size_t room = (size_t)160 * 1024 - 2048 - r - 4 * bge_quad_bytes(), per = bge_generic_wave_bytes(d);
  const int nw = (int)(room / per);
  return nw > 4 ? 4 : (nw < 1 ? 1 : nw);
}
// d <= 64: every problem has n <= 32 rows (complement form), the per-wave tier and its LDS are not needed
__host__ __device__ inline size_t bge_chol_lds_bytes(int d, bool r_in_lds) {
  const size_t r = r_in_lds ? (((size_t)2 * (d + 1) * (d + 1) * 4 + 15) & ~(size_t)15) : 0;
  const bool generic = d > 64;
  return r + 4 * bge_quad_bytes() + (generic ? (size_t)bge_generic_waves(d, r_in_lds) * bge_generic_wave_bytes(d) : 0);
}

// grid = any (persistent: work units are dealt round-robin, largest tier first), block = 256; dynamic LDS = bge_chol_lds_bytes()
// R_LDS: one matrix pair (no interventions) resident in LDS; otherwise R_j / Q_j are read through the caches.
// The walk over the work units is software-pipelined: the queue entries of a block's NEXT unit are requested before the current unit is
// factorised, and a problem's table entries (log-gamma term, N_j, logdet R) as soon as its parent count is known -- at two waves per SIMD
// (215 registers) nothing else hides those round trips (39 % of the wave cycles were s_waitcnt before: profiles/round2_pmc_lds_window.txt).
template <bool R_LDS, bool W2>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(R_LDS ? 2 : 1))) void k_bge_chol(double* __restrict__ node_scores, BgeParams bp, BgeQueues qs, int d, int S,
                                                  unsigned long long* __restrict__ counters) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  __shared__ unsigned int cnt_s[BGE_NQ];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // R / Q -> LDS: the loads of a thread are requested together, in front of everything else, and stored once the queue entries of the
  // block's first unit have been requested as well.  (Until round 4 this was `for (e = tid; ...) { Rs[e] = Rp[e]; Rs[msz + e] = Qp[e]; }`
  // behind the counter load and the entry fetch: hipcc keeps such a loop rolled with a full s_waitcnt per iteration -- 2 + 10 dependent
  // trips to the L2 at d = 50 before a block factorised anything, in each of the two rounds of blocks.)
  constexpr int RB = 12;
  const int ldr_ = d + 1, msz_ = ldr_ * ldr_;
  float rpre[R_LDS ? RB : 1], qpre[R_LDS ? RB : 1];
  if (R_LDS) {
#pragma unroll
    for (int u = 0; u < RB; ++u) {
      const int e = u * 256 + tid, ec = e < msz_ ? e : msz_ - 1;
      rpre[u] = bp.Rp[ec];
      qpre[u] = bp.Qp[ec];
    }
  }
  if (tid < BGE_NQ) cnt_s[tid] = qs.counts[tid];
  __syncthreads();
  const int nwg = W2 ? bge_generic_waves(d, R_LDS) : 4;
  unsigned int cnt[BGE_NQ], units[BGE_NQ], total = 0;
#pragma unroll
  for (int qi = 0; qi < BGE_NQ; ++qi) {
    cnt[qi] = cnt_s[qi];
    const unsigned int per = qi == 0 ? 256u * BGE_NPL0 : (qi == 1 ? 256u * BGE_NPL1 : (qi == 2 ? 256u * BGE_NPL2 : (qi == 3 ? 256u : (qi < 6 ? 128u : (qi < 8 ? 64u : (unsigned int)nwg)))));  // (tiers 4, 5: one problem per pair of lanes)
    units[qi] = (cnt[qi] + per - 1u) / per;
    total += units[qi];
  }
  if (blockIdx.x >= total) return;  // (block-uniform)
  const int W = W2 ? 2 : 1;
  // work unit u -> (tier, index inside the tier), largest tier first
  auto unit_of = [&](unsigned int u, int& qi, unsigned int& local) {
    qi = BGE_NQ - 1;
    unsigned int base = 0;
#pragma unroll
    for (int t = BGE_NQ - 1; t > 0; --t)
      if (qi == t && u >= base + units[t]) { base += units[t]; qi = t - 1; }
    local = u - base;
  };
  // queue entries of this thread's problems of a unit (up to BGE_NPL0 per thread; a quad / a wave shares one entry); j = ~0: no problem
  auto fetch = [&](int qi, unsigned int local, uint4 (&ea)[BGE_NPL0], uint4 (&eb)[BGE_NPL0]) {
    const unsigned int n_q = cnt_s[qi];
    const int npl = qi == 0 ? BGE_NPL0 : (qi == 1 ? BGE_NPL1 : (qi == 2 ? BGE_NPL2 : 1));
    const unsigned int usz = qi < 4 ? 256u * (unsigned int)npl : (qi < 6 ? 128u : (qi < 8 ? 64u : (unsigned int)nwg));
    const unsigned int idx = qi < 4 ? (unsigned int)tid : (qi < 6 ? (unsigned int)(tid >> 1) : (qi < 8 ? (unsigned int)(tid >> 2) : (unsigned int)wave));
    const bool lane_ok = qi < 8 || wave < nwg;
    const uint4* lst = qs.list + (size_t)qi * qs.cap * W;  // (bge_entry_u4(W) == W for W <= 2)
#pragma unroll
    for (int v = 0; v < BGE_NPL0; ++v) {
      const unsigned int pi = local * usz + 256u * (unsigned int)v + idx;
      const bool has = v < npl && pi < n_q && lane_ok;
      ea[v] = make_uint4(0u, 0xFFFFFFFFu, 0u, 0u);
      eb[v] = make_uint4(0u, 0u, 0u, 0u);
      if (has) {
        ea[v] = lst[(size_t)pi * W];
        if (W2) eb[v] = lst[(size_t)pi * W + 1];
      }
    }
  };
  // (d > 64 with R resident: the second mask word of the entries in flight would cost the second wave per SIMD -- fetched at use there)
  constexpr bool PRE = !(R_LDS && W2);
  uint4 na[BGE_NPL0], nb[BGE_NPL0];
  int nqi;
  unsigned int nlocal;
  unit_of(blockIdx.x, nqi, nlocal);
  if (PRE) fetch(nqi, nlocal, na, nb);  // (in flight while R / Q are staged)

  const int ldr = d + 1, msz = ldr * ldr;
  const float* Rg = bp.Rp;
  float* Rs = reinterpret_cast<float*>(smem_raw);
  const size_t r_bytes = R_LDS ? (((size_t)2 * msz * 4 + 15) & ~(size_t)15) : 0;
  if (R_LDS) {
#pragma unroll
    for (int u = 0; u < RB; ++u) {
      const int e = u * 256 + tid;
      if (e < msz) {
        Rs[e] = rpre[u];
        Rs[msz + e] = qpre[u];
      }
    }
    for (int e0 = RB * 256; e0 < msz; e0 += RB * 256) {  // (n_vars > 54)
      float rr[RB], qq[RB];
#pragma unroll
      for (int u = 0; u < RB; ++u) {
        const int e = e0 + u * 256 + tid, ec = e < msz ? e : msz - 1;
        rr[u] = bp.Rp[ec];
        qq[u] = bp.Qp[ec];
      }
#pragma unroll
      for (int u = 0; u < RB; ++u) {
        const int e = e0 + u * 256 + tid;
        if (e < msz) {
          Rs[e] = rr[u];
          Rs[msz + e] = qq[u];
        }
      }
    }
    __syncthreads();
  }
  int* const qbase = reinterpret_cast<int*>(smem_raw + r_bytes + (size_t)wave * bge_quad_bytes());
  unsigned char* const gbase = smem_raw + r_bytes + 4 * bge_quad_bytes() + (size_t)(wave < nwg ? wave : 0) * bge_generic_wave_bytes(d);
  // matrix offset of a problem relative to `Rm`: LDS holds [R | Q]; global memory holds Rp and Qp as separate arrays
  const float* Rm = R_LDS ? Rs : Rg;
  const long qoff = R_LDS ? (long)msz : (long)(bp.Qp - bp.Rp);

  float flops = 0.f;
  // one problem: decode its entry, index mask (parents or complement), matrix offset; table entries requested here
  struct Tab {
    double Nn, g, ldRv;
  };
  auto load = [&](const uint4& ea, const uint4& eb, bool& has, uint32_t& code, int& jj, int& l, int& li, int& mat, bool& comp, uint64_t& w0,
                  uint64_t& w1, Tab& tb) {
    has = ea.y != 0xFFFFFFFFu;
    code = ea.x;
    jj = has ? (int)ea.y : d;  // (no problem: every index is the padding index)
    w0 = ((uint64_t)ea.w << 32) | ea.z;
    w1 = W2 ? (((uint64_t)eb.y << 32) | eb.x) : 0ull;
    l = __popcll(w0) + __popcll(w1);
    if (PRE) {
      tb.Nn = has ? bp.Nj[jj] : 0.0;
      tb.g = has ? bp.gam[(size_t)jj * (d + 1) + l] : 0.0;
      tb.ldRv = has ? bp.ldR[bp.n_mats > 1 ? jj : 0] : 0.0;
    }
    comp = has && (l + 1 > d - l);
    bge_index_mask(w0, w1, jj, d, comp);
    li = has ? (comp ? d - 1 - l : l) : 0;  // rows before j
    mat = (int)((comp ? qoff : 0) + ((has && bp.n_mats > 1) ? (long)jj * msz : 0));
  };
  auto store = [&](bool writer, uint32_t code, int jj, const Tab& tb, int l, int li, bool comp, float ld2, float last) {
    if (writer) {
      if (PRE) node_scores[code] = bge_score_pre(tb.Nn, tb.g, tb.ldRv, bp.alpha_lambd, l, d, comp, ld2, last);
      else node_scores[code] = bge_score(bp, jj, l, d, comp, ld2, last);
    }
    if (counters) {  // profiling: executed Cholesky flops, n^3 / 3 per problem
      const float n = (float)(li + 1);
      flops += writer ? n * n * n * (1.0f / 3.0f) : 0.f;
    }
  };
#define BGE_LANE_TIER(NMAX_, NPL_)                                                                             \
  {                                                                                                            \
    int jj[NPL_], l[NPL_], li[NPL_], mat[NPL_];                                                                \
    bool comp[NPL_], has[NPL_];                                                                                \
    uint32_t code[NPL_];                                                                                       \
    uint64_t w0[NPL_], w1[NPL_];                                                                               \
    float ld2[NPL_], last[NPL_];                                                                               \
    Tab tb[NPL_];                                                                                              \
    _Pragma("unroll") for (int v = 0; v < NPL_; ++v) load(ca[v], cb[v], has[v], code[v], jj[v], l[v], li[v], mat[v], comp[v], w0[v], w1[v], tb[v]); \
    bge_chol_lane<NMAX_, W2, NPL_>(Rm, mat, ldr, d, w0, w1, jj, li, ld2, last);                                \
    _Pragma("unroll") for (int v = 0; v < NPL_; ++v) store(has[v], code[v], jj[v], tb[v], l[v], li[v], comp[v], ld2[v], last[v]); \
  }
#define BGE_QUAD_TIER(NB_)                                                                                     \
  {                                                                                                            \
    bool has, comp;                                                                                            \
    uint32_t code;                                                                                             \
    int jj, l, li, mat;                                                                                        \
    uint64_t w0, w1;                                                                                           \
    float ld2, last;                                                                                           \
    Tab tb;                                                                                                    \
    load(ca[0], cb[0], has, code, jj, l, li, mat, comp, w0, w1, tb);                                           \
    bge_chol_quad<NB_, W2>(Rm, mat, ldr, d, qbase + (lane >> 2) * BGE_QS, w0, w1, jj, li, ld2, last);          \
    store(has && (tid & 3) == 0, code, jj, tb, l, li, comp, ld2, last);                                            \
  }
#define BGE_PAIR_TIER(NR_)                                                                                     \
  {                                                                                                            \
    bool has, comp;                                                                                            \
    uint32_t code;                                                                                             \
    int jj, l, li, mat;                                                                                        \
    uint64_t w0, w1;                                                                                           \
    float ld2, last;                                                                                           \
    Tab tb;                                                                                                    \
    load(ca[0], cb[0], has, code, jj, l, li, mat, comp, w0, w1, tb);                                           \
    bge_chol_pair<NR_, W2>(Rm, mat, ldr, d, qbase + (lane >> 1) * BGE_PS, w0, w1, jj, li, ld2, last);          \
    store(has && (tid & 1) == 0, code, jj, tb, l, li, comp, ld2, last);                                            \
  }
  for (unsigned int u = blockIdx.x; u < total; u += gridDim.x) {
    const int qi = nqi;
    uint4 ca[BGE_NPL0], cb[BGE_NPL0];
    if (PRE) {
#pragma unroll
      for (int v = 0; v < BGE_NPL0; ++v) {
        ca[v] = na[v];
        cb[v] = nb[v];
      }
    } else {
      fetch(nqi, nlocal, ca, cb);
    }
    if (u + gridDim.x < total) {  // (block-uniform)
      unit_of(u + gridDim.x, nqi, nlocal);
      if (PRE) fetch(nqi, nlocal, na, nb);
    }
    switch (qi) {
      case 0: BGE_LANE_TIER(4, BGE_NPL0) break;
      case 1: BGE_LANE_TIER(8, BGE_NPL1) break;
      case 2: BGE_LANE_TIER(12, BGE_NPL2) break;
      case 3: BGE_LANE_TIER(16, 1) break;
      case 4: BGE_PAIR_TIER(10) break;
      case 5: BGE_PAIR_TIER(12) break;
      case 6: BGE_QUAD_TIER(7) break;
      case 7: BGE_QUAD_TIER(8) break;
      default:
        if (W2) {
          bool has, comp;
          uint32_t code;
          int jj, l, li, mat;
          uint64_t w0, w1;
          float ld2 = 0.f, last = 1.f;
          Tab tb;
          load(ca[0], cb[0], has, code, jj, l, li, mat, comp, w0, w1, tb);
          if (has) {  // (d <= 128 here: every such problem has 33 .. 64 rows)
            float* colb = reinterpret_cast<float*>(gbase);
            int* myidx = reinterpret_cast<int*>(colb + 64);
            myidx[lane] = d;
            wave_lds_fence();
            if ((w0 >> lane) & 1ull) myidx[__popcll(w0 & ((1ull << lane) - 1ull))] = lane;
            if ((w1 >> lane) & 1ull) myidx[__popcll(w0) + __popcll(w1 & ((1ull << lane) - 1ull))] = 64 + lane;
            wave_lds_fence();
            if (lane == 0) myidx[li] = jj;
            wave_lds_fence();
            bge_chol_wave_reg(Rm, (size_t)mat, ldr, d, colb, myidx, li + 1, li, ld2, last);
            wave_lds_fence();
          }
          store(has && lane == 0, code, jj, tb, l, li, comp, ld2, last);
        }
        break;
    }
  }
#undef BGE_LANE_TIER
#undef BGE_QUAD_TIER
#undef BGE_PAIR_TIER
  if (counters) {  // one atomic per block (same-address atomics from every wave cost tens of microseconds)
    __shared__ float fl_s[4];
    const float tot = wave_sum(flops);
    if (lane == 0) fl_s[wave] = tot;
    __syncthreads();
    const float bt = fl_s[0] + fl_s[1] + fl_s[2] + fl_s[3];
    if (tid == 0 && bt > 0.f) atomicAdd(counters, (unsigned long long)bt);
  }
}

// ------------------------------------------------------------------------------------------------
// K3b  n_vars > 128 (up to 256: parent sets of three or four mask words): every queued problem (last tier only, see k_bge_sample) is
//      factorised by ONE wave, the factor in LDS -- the algorithm of bge_chol_wave with the index set taken from up to four words.
//      n = min(l + 1, d - l) <= (d + 1) / 2 <= 128 rows: lane owns rows r and r + 64.  R / Q are read through the caches.
//      The fallback behind the constructor's full range, not tuned.
// grid = any (waves stride over the queue), block = 64 * BGE_WIDE_WAVES; dynamic LDS = BGE_WIDE_WAVES * bge_wide_wave_bytes(d)
// ------------------------------------------------------------------------------------------------
#define BGE_WIDE_WAVES 2
__host__ __device__ inline int bge_wide_nmax(int d) { return (d + 1) / 2; }
__host__ __device__ inline size_t bge_wide_wave_bytes(int d) {
  const size_t n = (size_t)bge_wide_nmax(d);
  return (((n * (n | 1) + 3) & ~(size_t)3) * 4 + (n + 4) * 4 + 15) & ~(size_t)15;
}
#ifdef DIBS_TU_BGE
__global__ __launch_bounds__(64 * BGE_WIDE_WAVES) void k_bge_chol_wide(double* __restrict__ node_scores, BgeParams bp, BgeQueues qs, int d, int W) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const unsigned int n_q = qs.counts[BGE_NQ - 1];
  const int nmax = bge_wide_nmax(d), ldl = nmax | 1, EW = bge_entry_u4(W);
  float* Lb = reinterpret_cast<float*>(smem_raw + (size_t)wave * bge_wide_wave_bytes(d));
  int* myidx = reinterpret_cast<int*>(Lb + (((size_t)nmax * ldl + 3) & ~(size_t)3));
  const uint4* lst = qs.list + (size_t)(BGE_NQ - 1) * qs.cap * EW;
  const int ldr = d + 1, msz = ldr * ldr;
  for (unsigned int pi = blockIdx.x * BGE_WIDE_WAVES + wave; pi < n_q; pi += gridDim.x * BGE_WIDE_WAVES) {
    const uint4 e0 = lst[(size_t)pi * EW];
    const uint4 e1 = W > 1 ? lst[(size_t)pi * EW + 1] : make_uint4(0u, 0u, 0u, 0u);
    const uint4 e2 = W > 3 ? lst[(size_t)pi * EW + 2] : make_uint4(0u, 0u, 0u, 0u);
    const uint32_t code = e0.x;
    const int j = (int)e0.y;
    uint64_t w[4] = {((uint64_t)e0.w << 32) | e0.z, ((uint64_t)e1.y << 32) | e1.x, ((uint64_t)e1.w << 32) | e1.z, ((uint64_t)e2.y << 32) | e2.x};
    int l = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) l += __popcll(w[q]);
    const bool comp = l + 1 > d - l;
    if (comp) {  // complement form: the non-parents other than j
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int lo = 64 * q, nb = d - lo;
        const uint64_t valid = nb >= 64 ? ~0ull : (nb > 0 ? (1ull << nb) - 1ull : 0ull);
        w[q] = ~w[q] & valid;
        if (j >= lo && j < lo + 64) w[q] &= ~(1ull << (j - lo));
      }
    }
    const int li = comp ? d - 1 - l : l, n = li + 1;  // rows before j; j goes last
    const float* R = comp ? bp.Qp : bp.Rp;
    const size_t mat = bp.n_mats > 1 ? (size_t)j * msz : 0;
    {
      int base = 0;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if ((w[q] >> lane) & 1ull) myidx[base + __popcll(w[q] & ((1ull << lane) - 1ull))] = 64 * q + lane;
        base += __popcll(w[q]);
      }
      if (lane == 0) myidx[li] = j;
    }
    wave_lds_fence();
    if (n <= 64) {  // (wave-uniform) the register-resident variant: most problems once the particles have sharpened
      wave_lds_fence();
      int* idx64 = myidx;
      // index list padded to 64 entries with d (the scatter above wrote n of them)
      const int mine = lane < n ? idx64[lane] : d;
      wave_lds_fence();
      idx64[lane] = mine;
      wave_lds_fence();
      float ld2r, lastr;
      bge_chol_wave_reg(R, mat, ldr, d, Lb, idx64, n, li, ld2r, lastr);
      wave_lds_fence();
      if (lane == 0) node_scores[code] = bge_score(bp, j, l, d, comp, ld2r, lastr);
      continue;
    }
    float mypiv[2] = {1.f, 1.f};
    for (int kk = 0; kk < n; ++kk) {
      const int ik = myidx[kk];
      float accs[2];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int r = lane + h * 64;
        float acc = 0.f;
        if (r >= kk && r < n) {
          acc = R[mat + (size_t)myidx[r] * ldr + ik];
          const float* lr = Lb + (size_t)r * ldl;
          const float* lk = Lb + (size_t)kk * ldl;
#pragma unroll 8
          for (int p_ = 0; p_ < kk; ++p_) acc = fmaf(-lr[p_], lk[p_], acc);
        }
        accs[h] = acc;
        if (r == kk) mypiv[h] = acc;
      }
      const float piv = __shfl(kk < 64 ? accs[0] : accs[1], kk & 63, 64);
      const float inv = __builtin_amdgcn_rsqf(piv);
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int r = lane + h * 64;
        if (r > kk && r < n) Lb[(size_t)r * ldl + kk] = accs[h] * inv;
      }
      wave_lds_fence();
    }
    float lg = 0.f;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int r = lane + h * 64;
      if (r < li) lg += __log2f(mypiv[h]);
    }
    const float ld2 = wave_sum(lg);
    const float last = __shfl(li < 64 ? mypiv[0] : mypiv[1], li & 63, 64);
    wave_lds_fence();
    if (lane == 0) node_scores[code] = bge_score(bp, j, l, d, comp, ld2, last);
  }
}

// out[s] = sum_j node_scores[j][s]   (scoring of given graphs)
__global__ void k_sum_nodes(const double* __restrict__ node_scores, float* __restrict__ out, int d, int S) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= S) return;
  double t = 0.0;
  for (int j = 0; j < d; ++j) t += node_scores[(size_t)j * S + s];
  out[s] = (float)t;
}
#endif
