// HIP kernels (gfx950) for the MarginalDiBS + BGe SVGD step.  One step = the launches listed in
// engine.hip::step_local / step_update; DESIGN.md has the per-kernel roofline and byte counts.
#pragma once
#include "common.h"

#include "kernels_kmat.h"

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
// grid = (Mloc, NB), block = 256; dynamic LDS = 2 * dpad * ldk * 4, ldk = row stride of a latent-dimension chunk of `kc` columns.
// The latent dimension is walked in chunks of kc columns (one chunk when U, V fit in LDS -- n_vars <= 112 with k = d); a wave keeps the
// accumulators of its tiles (t = first, first + stride, ...; at most MAXT <= EDGE_MAXT) across the chunks, so the fma chain of every score runs
// over k in ascending order whatever the chunking: scores do not depend on it.
// ------------------------------------------------------------------------------------------------
#define EDGE_MAXT 12
template <int MAXT>   // accumulator sets per wave: 1 (up to 16 tiles per particle: n_vars <= 64 with four blocks), 4, EDGE_MAXT
__global__ __launch_bounds__(256) void k_edge_scores(const float* __restrict__ z, float* __restrict__ scores,
                                                     uint32_t* __restrict__ thr, float* __restrict__ probs, float* __restrict__ eas,
                                                     float alpha, int d, int k, int dpad, int ldk, int kc, unsigned long long* dbg) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  unsigned long long ts[5] = {0, 0, 0, 0, 0};  // (profiling only: dbg != null)
  if (dbg) ts[0] = ts[1] = ts[2] = wall_clock64();
  float* Us = smem;
  float* Vs = smem + (size_t)dpad * ldk;
  const int m = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float2* zm = reinterpret_cast<const float2*>(z + (size_t)m * d * k * 2);
  const int nt = dpad >> 4;
  // tiles are dealt out over the waves of the gridDim.y blocks of this particle (the epilogue's double-precision sigmoid
  // is most of the work: more blocks, shorter critical path)
  const int t0 = blockIdx.y * 4 + wave, tstride = 4 * gridDim.y;
  f32x4 acc[MAXT];
#pragma unroll
  for (int u = 0; u < MAXT; ++u) acc[u] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int q0 = 0; q0 < k; q0 += kc) {
    const int kn = k - q0 < kc ? k - q0 : kc, kp = (kn + 3) & ~3;
    if (q0) __syncthreads();
    // Z -> LDS.  The loads of a thread are requested TOGETHER (16 per batch: one batch at d = k = 50) and the padding is zero-filled while
    // they are in flight.  (Rounds 1-4 had `for (e = tid; ...) { load; store to LDS; }`: hipcc keeps such a loop rolled with a full
    // s_waitcnt in front of every LDS store -- 17 dependent trips to the L2 per block, 8 of the kernel's 10 us.)
    constexpr int ZB = 16;
    const int nval = d * kn;
    for (int e0 = 0; e0 < nval; e0 += ZB * 256) {
      float2 uv[ZB];
#pragma unroll
      for (int u = 0; u < ZB; ++u) {
        const int e = e0 + u * 256 + tid, ec = e < nval ? e : nval - 1, i = ec / kn, q = ec - i * kn;
        uv[u] = zm[(size_t)i * k + q0 + q];  // (past the end: the last element again, not stored)
      }
      if (e0 == 0) {
        for (int e = tid; e < dpad * ldk; e += 256) {
          const int i = e / ldk, q = e - i * ldk;
          if (i >= d || q >= kn) {
            Us[e] = 0.f;
            Vs[e] = 0.f;
          }
        }
      }
#pragma unroll
      for (int u = 0; u < ZB; ++u) {
        const int e = e0 + u * 256 + tid, i = e / kn, q = e - i * kn;
        if (e < nval) {
          Us[i * ldk + q] = uv[u].x;
          Vs[i * ldk + q] = uv[u].y;
        }
      }
    }
    if (dbg) ts[1] = wall_clock64();
    __syncthreads();
    if (dbg) ts[2] = wall_clock64();
#pragma unroll
    for (int u = 0; u < MAXT; ++u) {
      const int t = t0 + u * tstride;
      if (t < nt * nt) {
        const int ti = t / nt, tj = t - ti * nt;
        const float* ua = Us + (size_t)(ti * 16 + (lane & 15)) * ldk + (lane >> 4);
        const float* vb = Vs + (size_t)(tj * 16 + (lane & 15)) * ldk + (lane >> 4);
        f32x4 a = acc[u];
        for (int k0 = 0; k0 < kp; k0 += 4) a = __builtin_amdgcn_mfma_f32_16x16x4f32(ua[k0], vb[k0], a, 0, 0, 0);
        acc[u] = a;
      }
    }
  }
  if (dbg) ts[3] = wall_clock64();
#pragma unroll
  for (int u = 0; u < MAXT; ++u) {
    const int t = t0 + u * tstride;
    if (t >= nt * nt) continue;
    const int ti = t / nt, tj = t - ti * nt;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = ti * 16 + (lane >> 4) * 4 + r, col = tj * 16 + (lane & 15);
      if (row < d && col < d) {
        const float s = acc[u][r];
        const size_t o = ((size_t)m * d + row) * d + col;
        scores[o] = s;
        // (eas: exp(-alpha s) as float, the factor of the Gumbel-soft graphs u / (u + (1 - u) exp(-alpha s)) that is common to all
        //  acyclicity chains of the particle -- k_acyc_hf would evaluate it once per pair of chains)
        const double ex = exp(-(double)__fmul_rn(alpha, s));
        const float pf = (float)(1.0 / (1.0 + ex));
        if (eas) eas[o] = (float)ex;
        thr[o] = row == col ? 0u : (uint32_t)ceilf(pf * 8388608.0f);
        probs[o] = row == col ? 0.f : pf;  // edge_probs (dibs.py:168-184), reused by the prior / estimator kernels
      }
    }
  }
  if (dbg && blockIdx.x == 0 && blockIdx.y == 0 && tid == 0) {  // profiling: loads+stores | barrier | MFMA | epilogue (100 MHz ticks)
    if (dbg) ts[4] = wall_clock64();
    for (int u = 1; u < 5; ++u) atomicAdd(dbg + u, ts[u] - ts[u - 1]);
  }
}

// K1p (round 5)  the same for n_vars <= 64, k <= 64: ONE block of 16 waves per particle.  k_edge_scores runs four blocks per particle and
//     each of them loads the whole particle (the score tiles of a block need 16 rows of U and all of V, but U and V are interleaved): 4 x 40
//     KB at the headline size, and its load phase is what the kernel spends its time in (4.1 of 10 us, measured with clock stamps).  Here Z is
//     loaded once -- wave w takes rows w, w + 16, ..., one lane per latent column, all loads of a lane in flight together, no index
//     arithmetic beyond an add -- and wave t computes tile t.  Same MFMA sequence, same epilogue: results are bit-identical.
// grid = Mloc, block = 1024; dynamic LDS = 2 * dpad * ldk * 4
__global__ __launch_bounds__(1024) void k_edge_scores_p(const float* __restrict__ z, float* __restrict__ scores, uint32_t* __restrict__ thr,
                                                        float* __restrict__ probs, float* __restrict__ eas, float alpha, int d, int k,
                                                        int dpad, int ldk, unsigned int* __restrict__ done_ctr,
                                                        unsigned int* __restrict__ done_flag, unsigned int done_seq) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Us = smem;
  float* Vs = smem + (size_t)dpad * ldk;
  const int m = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float2* zm = reinterpret_cast<const float2*>(z + (size_t)m * d * k * 2);
  const int nt = dpad >> 4;
  float2 uv[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int i = wave + 16 * r;
    const bool in = i < d && lane < k;
    uv[r] = zm[in ? i * k + lane : 0];
    if (!in) uv[r] = make_float2(0.f, 0.f);
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int i = wave + 16 * r;
    if (i < dpad) {
      if (lane < ldk) {  // (ldk < 64 for small k: the row ends before the wave does)
        Us[i * ldk + lane] = uv[r].x;
        Vs[i * ldk + lane] = uv[r].y;
      }
      if (64 + lane < ldk) {  // (row padding beyond the 64 latent columns a wave covers: ldk = kp + (2 - kp) mod 32 <= 98)
        Us[i * ldk + 64 + lane] = 0.f;
        Vs[i * ldk + 64 + lane] = 0.f;
      }
    }
  }
  __syncthreads();
  const int t = wave;
  if (t < nt * nt) {  // (wave-uniform)
    const int ti = t / nt, tj = t - ti * nt, kp = (k + 3) & ~3;
    const float* ua = Us + (size_t)(ti * 16 + (lane & 15)) * ldk + (lane >> 4);
    const float* vb = Vs + (size_t)(tj * 16 + (lane & 15)) * ldk + (lane >> 4);
    f32x4 a = {0.f, 0.f, 0.f, 0.f};
    for (int k0 = 0; k0 < kp; k0 += 4) a = __builtin_amdgcn_mfma_f32_16x16x4f32(ua[k0], vb[k0], a, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = ti * 16 + (lane >> 4) * 4 + r, col = tj * 16 + (lane & 15);
      if (row < d && col < d) {
        const float s = a[r];
        const size_t o = ((size_t)m * d + row) * d + col;
        const double ex = exp(-(double)__fmul_rn(alpha, s));  // (the epilogue of k_edge_scores, operation for operation)
        const float pf = (float)(1.0 / (1.0 + ex));
        if (done_ctr) {  // (what the second stream reads goes out at agent scope: see below)
          __hip_atomic_store(scores + o, s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (eas) __hip_atomic_store(eas + o, (float)ex, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
          scores[o] = s;
          if (eas) eas[o] = (float)ex;
        }
        if (thr) thr[o] = row == col ? 0u : (uint32_t)ceilf(pf * 8388608.0f);  // (null: the copy of the second stream, scores / eas only)
        if (probs) probs[o] = row == col ? 0.f : pf;
      }
    }
  }
  // done_ctr != null: the fork to the engine's second stream without an event (k_wait_flag there polls done_flag): scores / eas were stored at
  // agent scope (complete once the storing wave has waited for vmcnt(0): the barrier alone does not), every block counts itself, the last one
  // publishes the step's sequence number
  if (done_ctr) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0 && atomicAdd(done_ctr, 1u) == gridDim.x - 1u) {
      atomicExch(done_ctr, 0u);  // (the next launch of this kernel is behind this one in its stream)
      __hip_atomic_store(done_flag, done_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// the tiled kernel matrix as launches of its own (second stream of the joint / many-particle configurations): grid <= tiles * chunks, block = 1024;
// dynamic LDS = kmat_tile_lds_bytes().  k_kmat_finish: grid = Mloc, block = 256
__global__ __launch_bounds__(KT_NT) void k_kmat_tile(KmatTile kt) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  kmat_tile_block(smem, kt, (int)blockIdx.x, (int)gridDim.x, (int)threadIdx.x);
}
__global__ __launch_bounds__(256) void k_kmat_finish(const double* __restrict__ part, int nchunk, int Mloc, int M, int symmetric, float scale, float h,
                                                     float* __restrict__ kout, const float* __restrict__ kadd, float* __restrict__ ksum, int tile) {
  kmat_finish_row(part, nchunk, Mloc, M, symmetric, scale, h, kout, kadd, ksum, (int)blockIdx.x, (int)threadIdx.x, 256, tile);
}
// 64 x 64 tiles (kernels_kmat.h, K8a''): grid <= tiles * pieces, block = 512; dynamic LDS = kmat_tile64_lds_bytes()
__global__ __launch_bounds__(KT2_NT) void k_kmat_tile64(KmatTile kt) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  kmat_tile64_block(smem, kt, (int)blockIdx.x, (int)gridDim.x, (int)threadIdx.x);
}

__global__ __launch_bounds__(256) void k_kmat(const float* __restrict__ pack, size_t pack_stride, size_t seg_off, int len,
                                              float* __restrict__ kout, int m0, int M, float scale, float h, int symmetric,
                                              const float* __restrict__ kadd, float* __restrict__ ksum) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  kmat_block(smem, pack, pack_stride, seg_off, len, kout, m0, M, scale, h, symmetric, blockIdx.x, blockIdx.y, kadd, ksum);
}

// rmsprop of jax.example_libraries.optimizers (svgd.py:117-120, 265): v <- 0.9 v + (1 - 0.9) phi^2, x <- x - step phi / sqrt(v + 1e-8).
// Every operation rounded on its own (no contraction into FMAs): the instantiations of k_phi_update / k_phi_gemm are chosen from the SHARD
// size, and the compiler contracted `v * 0.9f + phi * phi * 0.1f` differently in them -- a 128-particle run on 4 ranks (TA = 4) differed
// from the single-rank run (TA = 8) in the last bit of z from the third step on (found by tests/test_gpu_ipc.py).
// (`#pragma clang fp contract(off)`: HIP's __fmul_rn / __fadd_rn are plain operators to the compiler and contract like them.)
__device__ __forceinline__ float rmsprop_moment(float v, float phi) {
#pragma clang fp contract(off)
  const float a = v * 0.9f, p2 = phi * phi, b = p2 * 0.1f;
  return a + b;
}
__device__ __forceinline__ float rmsprop_step(float x, float vv, float phi, float stepsize) {
#pragma clang fp contract(off)
  const float num = stepsize * phi, den = sqrtf(vv + 1e-8f), q = num / den;
  return x - q;
}
__device__ __forceinline__ float gd_step(float x, float phi, float stepsize) {
#pragma clang fp contract(off)
  const float q = stepsize * phi;
  return x - q;
}

// ------------------------------------------------------------------------------------------------
// K8b+K9 SVGD transform + optimizer step on one segment (z or theta) of the packed rows:
//     phi_a = -(1/M) sum_b [ (kz+kt)[a,b] grad_b - (2/h) kseg[a,b] (x_b - x_a) ]   (column kxx[:,a], symmetric kernel)
//     rmsprop: v = 0.9 v + 0.1 phi^2 ; x -= step * phi / sqrt(v + 1e-8)      |  gd: x -= step * phi
//     reference: svgd.py:194-224, 591-670, 265, 718-719; jax.example_libraries.optimizers.rmsprop
// grid = (ceil(len / 256), ceil(Mloc / TA)), block = 256;  vout != null: the updated segment is also written to vout[a][vout_off + i]
// ------------------------------------------------------------------------------------------------
// block = 64 consecutive elements x TA local particles; wave w sums over the quarter b in [w Mq, (w+1) Mq) of the particles and
// the four partial sums are added in wave order.  Round 5: inside a quarter the rows are taken in PAIRS (b_lo + 2p, b_lo + 2p + 1) whose
// two running sums are the halves of one packed register (v_pk_fma_f32 over rows instead of over particles: the loaded values pair up as
// they arrive -- packing over particles made hipcc duplicate every loaded value into a register pair, 128 VGPRs for 64 values) and are
// added at the end, even rows + odd rows.  The order is a function of M only: results do not depend on TA or the rank count.
// Every [z_b | grad_b] element read from L2 serves TA particles; the kernel entries are scalar operands (s_load from kz / kt).
// dynamic LDS = phi_update_lds_bytes(TA, M)
__device__ __forceinline__ float buf_ld(__amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}
__host__ __device__ inline size_t phi_update_lds_bytes(int TA, int M) {
  (void)M;
  return (size_t)4 * TA * 64 * 4;  // the four waves' partial sums
}
// FULL: M is a multiple of 64, i.e. every wave's quarter is a whole number of 8-pair batches, whole particle groups, buffer < 4 GiB -- no
// clamps, no per-pair tests, buffer loads (the headline size, configs 3 and 4); otherwise rows past a quarter repeat its last row and meet zero
// kernel entries.  JOINT: a second kernel matrix (theta), weights ks = kz + kt and the repulsion of the segment's own kernel.
template <int TA, bool FULL, bool JOINT>
__global__ __launch_bounds__(256) void k_phi_update(const float* __restrict__ pack, size_t pack_stride, size_t val_off,
                                                    size_t grad_off, int len, const float* __restrict__ kz,
                                                    const float* __restrict__ kt, int seg_is_theta, float* __restrict__ x,
                                                    float* __restrict__ v, float* __restrict__ phi_out, int m0, int Mloc,
                                                    int M, float h, float stepsize, int rmsprop, int ncols, int ngroups,
                                                    float* __restrict__ vout, size_t vout_stride, size_t vout_off) {
  // 1-D grid, XCD-aware: workgroups go round-robin to the 8 XCDs, each with its own L2.  Linear id L = 8 (ngroups c_hi + g) + c_lo runs
  // the `ngroups` particle groups of column slab c = 8 c_hi + c_lo on XCD c_lo, back to back: the slab's [z_b | grad_b] rows (64 KiB) are
  // fetched into that L2 once instead of once per group (a (cols, groups) grid spread every slab over 4 XCDs and re-read the 5 MB of
  // packed rows 16 times per launch: 82 MB of L2 misses).
  const int L = blockIdx.x, c_lo = L & 7, tq = L >> 3, grp = tq % ngroups, bx = (tq / ngroups) * 8 + c_lo;
  if (bx >= ncols) return;  // (block-uniform)
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* part = smem;  // [4][TA][64] partial sums of the four waves
  const int Mq = (M + 3) >> 2;  // rows per wave
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // (uniform: row offsets live in SGPRs)
  // What bounds this kernel is not latency -- all its blocks are resident at once (latest block start 0.7 us after the first) -- but what a
  // wave ISSUES.  Rounds 2-4: ~2 700 instructions per wave for 512 of arithmetic (a 64-bit multiply chain per load on the ONE scalar unit
  // the four SIMDs of a CU share, integer divisions in the table index, clamps, per-row tests) = 14 M wave-instructions per launch, 19 us.
  // A first lean version still took 21 us: its 128 broadcast ds_read_b128 of the kernel tables per wave are 8 LDS cycles each = 6.8 us per
  // CU.  Now the kernel entries are what they are -- wave-uniform SCALARS: read with s_load straight from kz (k[a][b], k[a][b + 1] are
  // neighbours: one 64-bit scalar operand of the packed FMA), no LDS table, no staging, no barrier in front of the arithmetic; rows arrive
  // through buffer loads (descriptor + scalar row offset + one vector column offset: one instruction per load).
  const int a0 = grp * TA;
  const float c2h = 2.0f / h;
  const int i = bx * 64 + lane;
  const bool ok = i < len;
  const uint32_t il = (uint32_t)(ok ? i : len - 1);  // (loads of lanes past the end are clamped, not predicated)
  const int b_lo = wave * Mq, b_hi = (b_lo + Mq < M) ? b_lo + Mq : M;
  const int nrow = FULL ? Mq : (b_hi > b_lo ? b_hi - b_lo : 0);  // (wave-uniform)
  const int npair = (nrow + 1) >> 1;
  const int rlast = nrow > 0 ? b_lo + nrow - 1 : 0;
  constexpr int PFP = 8;  // row PAIRS per batch
  const __amdgpu_buffer_rsrc_t rs_g = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(pack + grad_off), 0, 0xFFFFFFFF, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(pack + val_off), 0, 0xFFFFFFFF, 0x00020000);
  const uint32_t stride_b = (uint32_t)pack_stride * 4u;
  // rows of batch `pb` -> registers.  FULL: consecutive rows, scalar offsets; else clamped row numbers through plain pointers
  auto fetch = [&](int pb, f32x2 (&g)[PFP], f32x2 (&xv)[PFP]) {
    if constexpr (FULL) {
      // (two descriptors -- gradient and value segment -- share the row's scalar offset, which advances between the loads: one SGPR in
      //  flight instead of one per load; the sched_barrier keeps hipcc from computing all 32 offsets first and spilling them to VGPR lanes)
      uint32_t so = (uint32_t)(b_lo + 2 * PFP * pb) * stride_b;
#pragma unroll
      for (int p = 0; p < PFP; ++p) {
        const float g_lo = buf_ld(rs_g, il * 4u, so), x_lo = buf_ld(rs_x, il * 4u, so);
        so += stride_b;
        __builtin_amdgcn_sched_barrier(0);
        const float g_hi = buf_ld(rs_g, il * 4u, so), x_hi = buf_ld(rs_x, il * 4u, so);
        so += stride_b;
        __builtin_amdgcn_sched_barrier(0);
        g[p] = f32x2{g_lo, g_hi};
        xv[p] = f32x2{x_lo, x_hi};
      }
    } else {
#pragma unroll
      for (int p = 0; p < PFP; ++p) {
        const int q0 = b_lo + 2 * (PFP * pb + p), r0 = q0 < rlast ? q0 : rlast, r1 = q0 + 1 < rlast ? q0 + 1 : rlast;
        const float* p0 = pack + (size_t)r0 * pack_stride;
        const float* p1 = pack + (size_t)r1 * pack_stride;
        g[p] = f32x2{(p0 + grad_off)[il], (p1 + grad_off)[il]};
        xv[p] = f32x2{(p0 + val_off)[il], (p1 + val_off)[il]};
      }
    }
  };
  f32x2 ga[PFP], xa_[PFP], gb[PFP], xb_[PFP];
  constexpr int KF = (TA >= 8 ? 2 : 4) / (JOINT ? 2 : 1);  // row pairs per group of scalar kernel-entry loads (TA * 2 KF SGPRs per group and matrix)
  fetch(0, ga, xa_);
  f32x2 xa2[TA];  // the block's own values, each in both halves of a register pair (the packed subtraction below wants them so); JOINT: times -(2 / h)
  float xown[TA / 4];  // ... and those of the particles this thread finishes below (a0 + wave + 4 qq), as they are
  const f32x2 mc2h = f32x2{-c2h, -c2h};
#pragma unroll
  for (int qq = 0; qq < TA / 4; ++qq) xown[qq] = 0.f;
#pragma unroll
  for (int q = 0; q < TA; ++q) {
    const int a = (FULL || a0 + q < Mloc) ? a0 + q : Mloc - 1;
    const float t = FULL ? buf_ld(rs_x, il * 4u, (uint32_t)(m0 + a) * stride_b) : (pack + (size_t)(m0 + a) * pack_stride + val_off)[il];
    xa2[q] = JOINT ? f32x2{t, t} * mc2h : f32x2{t, t};
#pragma unroll
    for (int qq = 0; qq < TA / 4; ++qq) xown[qq] = (q == wave + 4 * qq) ? t : xown[qq];  // (wave-uniform select)
  }
  // optimizer state of the elements this thread finishes below (particle a0 + wave + 4 qq, element i)
  float ve[TA / 4];
#pragma unroll
  for (int qq = 0; qq < TA / 4; ++qq) {
    const int a = (FULL || a0 + wave + 4 * qq < Mloc) ? a0 + wave + 4 * qq : Mloc - 1;
    ve[qq] = (v + (size_t)a * len)[il];
  }
  f32x2 acc2[TA];
#pragma unroll
  for (int q = 0; q < TA; ++q) acc2[q] = f32x2{0.f, 0.f};
  // rows b = b_lo + 2 pr, b + 1 against the TA particles of the block; the two halves of acc2 = even / odd rows of the quarter
  auto pair_step = [&](int pr, f32x2 gp, f32x2 xp) {
    const int b = b_lo + 2 * pr;
    const f32x2 xpc = JOINT ? xp * mc2h : xp;
    const bool two = FULL || 2 * pr + 1 < nrow;  // (an odd quarter's last pair has one row)
#pragma unroll
    for (int q = 0; q < TA; ++q) {
      const int a = (FULL || a0 + q < Mloc) ? a0 + q : Mloc - 1;  // (groups past the end: finite values, never stored)
      const float* kp = kz + (size_t)a * M + b;                    // wave-uniform address: s_load
      f32x2 k1 = f32x2{kp[0], two ? kp[1] : 0.f};
      if constexpr (!JOINT) {
        // ks g - kr (x_b - x_a) with ks = kz, kr = (2 / h) kz:   kz (g - (2 / h) (x_b - x_a))
        acc2[q] = __builtin_elementwise_fma(k1, __builtin_elementwise_fma(mc2h, xp - xa2[q], gp), acc2[q]);
      } else {
        // (kz + kt) g - (2 / h) k_seg (x_b - x_a) as three multiply-adds whose kernel entries stay scalar operands: forming kz + kt or
        // (2 / h) k_seg first would be vector instructions on wave-uniform values (and, hoisted by hipcc, 230 VGPRs)
        // JOINT: kz points at the weight matrix kz + kt (k_kmat forms it), kt at the matrix of THIS segment (repulsion); both stay scalar
        // operands.  xpc / xac carry the factor -(2 / h).
        const float* tp = kt + (size_t)a * M + b;
        const f32x2 k2 = f32x2{tp[0], two ? tp[1] : 0.f};
        acc2[q] = __builtin_elementwise_fma(k2, xpc - xa2[q], __builtin_elementwise_fma(k1, gp, acc2[q]));
      }
    }
  };
  // batches of 8 row pairs, double-buffered: the next batch is requested before the current one is multiplied.  Order of the sums: pairs
  // ascending -- a function of M only.
  const int nbat = (npair + PFP - 1) / PFP;
  for (int pb = 0; pb < nbat; pb += 2) {
    if (pb + 1 < nbat) fetch(pb + 1, gb, xb_);
    // (a compiler fence per KF pairs: hipcc otherwise hoists the scalar loads of the whole quarter -- 256 SGPRs -- to the top and parks
    //  them in VGPR lanes: 235 v_writelane / v_readlane per wave)
#pragma unroll
    for (int p = 0; p < PFP; ++p) {
      if ((p & (KF - 1)) == 0) asm volatile("" ::: "memory");
      if (FULL || PFP * pb + p < npair) pair_step(PFP * pb + p, ga[p], xa_[p]);
    }
    if (pb + 1 < nbat) {
      if (pb + 2 < nbat) fetch(pb + 2, ga, xa_);
#pragma unroll
      for (int p = 0; p < PFP; ++p) {
        if ((p & (KF - 1)) == 0) asm volatile("" ::: "memory");
        if (FULL || PFP * (pb + 1) + p < npair) pair_step(PFP * (pb + 1) + p, gb[p], xb_[p]);
      }
    }
  }
#pragma unroll
  for (int q = 0; q < TA; ++q) part[(wave * TA + q) * 64 + lane] = acc2[q].x + acc2[q].y;  // even rows + odd rows of the quarter
  __syncthreads();
  if (ok) {
#pragma unroll
    for (int qq = 0; qq < TA / 4; ++qq) {
      const int q = wave + 4 * qq, a = a0 + q;
      if (!FULL && a >= Mloc) continue;
      const float tot = ((part[(0 * TA + q) * 64 + lane] + part[(1 * TA + q) * 64 + lane]) + part[(2 * TA + q) * 64 + lane]) +
                        part[(3 * TA + q) * 64 + lane];
      const float phi = -tot / (float)M;
      const float xv = xown[qq];  // (own value: loaded in the prologue)
      const size_t o = (size_t)a * len + i;
      if (phi_out) phi_out[o] = phi;
      float xn;
      if (rmsprop) {
        const float vv = rmsprop_moment(ve[qq], phi);
        v[o] = vv;
        xn = rmsprop_step(xv, vv, phi, stepsize);
      } else {
        xn = gd_step(xv, phi, stepsize);
      }
      x[o] = xn;
      // overlapped exchange: the new value also goes straight into this rank's send rows [Mloc][Ev] (no separate export pass)
      if (vout) vout[(size_t)a * vout_stride + vout_off + i] = xn;
    }
  }
}


// ------------------------------------------------------------------------------------------------
// K8c  the same transform + optimizer step for MANY particles (M >= 256: config 4's 1 024, config 5's 256) as one GEMM on the f32 MFMA:
//     -M phi_a(i) = sum_b ks[a,b] grad_b(i)  +  sum_b (-kr[a,b]) x_b(i)  +  x_a(i) sum_b kr[a,b],      ks = kz + kt,  kr = (2/h) kseg
//   i.e. [ks | -kr] (Mloc x 2M) times [grad ; x] (2M x len) plus a rank-one term whose row sums are accumulated from the same LDS tiles.
//   (k_phi_update evaluates kr (x_b - x_a) per pair on the vector ALU -- three instructions per packed FMA, 847 us at M = 1 024.  The
//   regrouped sum differs by the rounding of c |x_a| per particle, ~1e-7 of the gradient's own magnitude; PHI within 1e-5 in the tests.)
// Block = 64 particles x 64 elements, 4 waves (wave w: rows 16 w .. 16 w + 15, 1 x 4 MFMA tiles; 128-particle blocks leave 2.5 blocks per
// CU at config 4 -- CUs with three set the time: 310 us against [see below] with five 64-particle blocks per CU), k-chunks of 32 particles, the next
// chunk's tiles prefetched into registers while the current one is multiplied.  The accumulation order of every output element is a
// function of M only (chunks in ascending b, first the gradient half, then the value half): results do not depend on the rank count.
// grid = 8 * ceil(Mloc / 64) * ceil(ceil(len / 64) / 8) (one-dimensional, re-indexed XCD-aware inside), block = 256
// ------------------------------------------------------------------------------------------------
#define PG_RT 1            // row tiles (of 16 particles) per wave
#define PG_BM (64 * PG_RT)
#define PG_BN 64
#define PG_BK 32
#define PG_LDA 36  // A tile [a][k]: fragment reads (row = lane % 16, k = lane / 16) two-way (the 64-lane minimum)
#define PG_LDB 80  // B tile [k][i]: row stride == 16 mod 32
template <bool JOINT>
__global__ __launch_bounds__(256) void k_phi_gemm(const float* __restrict__ pack, size_t pack_stride, size_t val_off, size_t grad_off, int len,
                                                  const float* __restrict__ kz, const float* __restrict__ kt, int seg_is_theta,
                                                  float* __restrict__ x, float* __restrict__ v, float* __restrict__ phi_out, int m0, int Mloc,
                                                  int M, float h, float stepsize, int rmsprop, int ncols, int nrb, float* __restrict__ vout,
                                                  size_t vout_stride, size_t vout_off) {
  __shared__ __attribute__((aligned(16))) float As[PG_BM * PG_LDA];
  __shared__ __attribute__((aligned(16))) float Bs[PG_BK * PG_LDB];
  __shared__ float rs_s[PG_BM];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, r = lane & 15;
  // 1-D grid, XCD-aware (as k_phi_update): linear id L = 8 (nrb c_hi + rb) + c_lo runs the nrb row blocks of column slab c = 8 c_hi + c_lo
  // on XCD c_lo at the same time -- the slab's [grad ; x] columns (2 M x 256 bytes) come out of that L2 for all but the first reader
  const int L = blockIdx.x, c_lo = L & 7, tq = L >> 3, rb = tq % nrb, cb = (tq / nrb) * 8 + c_lo;
  if (cb >= ncols) return;  // (block-uniform)
  const int i0 = cb * PG_BN, a0 = rb * PG_BM;
  const float c2h = 2.0f / h;
  const int nch = (M + PG_BK - 1) / PG_BK;  // chunks per half
  // loader roles: A tile -- k = tid % 32, rows tid / 32 + 8 j (32 lanes read 128 contiguous bytes of a kernel-matrix row);
  //               B tile -- rows tid / 16 and tid / 16 + 16, columns 4 (tid % 16) .. + 3
  // The next chunk's tiles wait in registers while the current one is multiplied out of LDS (a chunk is 64 MFMAs per wave: the loads of
  // a slab nobody has touched yet come from beyond the L2).
  const int lk = tid & 31, la = tid >> 5;
  const int lb = tid >> 4, lc = (tid & 15) * 4;
  const bool vec_ok = ((pack_stride | val_off | grad_off) & 3) == 0;
  // (loads are unconditional -- the kernel matrices are allocated with a tile of slack, the rows of `pack` are clamped -- and the raw values
  //  wait in registers: they are masked / combined only when they are stored to LDS, since any arithmetic on them here would put the wait
  //  for the loads in front of the MFMAs.  One 32-bit lane offset serves all rows of the A tile: their bases are uniform.)
  float pz[8 * PG_RT], pt[JOINT ? 8 * PG_RT : 1];
  float4 pb[2];
  const bool full_cols = vec_ok && i0 + PG_BN <= len;  // block-uniform
  const uint32_t aoff = (uint32_t)((a0 + la) * M + lk) * 4u;
  auto fetch = [&](int ch) {
    const bool xhalf = ch >= nch;
    const int b0 = (xhalf ? ch - nch : ch) * PG_BK;
    const uint32_t vo = aoff + (uint32_t)b0 * 4u;
#pragma unroll
    for (int j = 0; j < 8 * PG_RT; ++j) {
      pz[j] = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(kz + (size_t)j * 8 * M) + vo);
      if constexpr (JOINT) pt[j] = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(kt + (size_t)j * 8 * M) + vo);
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int bb = b0 + lb + 16 * u, bbc = bb < M ? bb : M - 1, i = i0 + lc;
      const float* src = pack + (size_t)bbc * pack_stride + (xhalf ? val_off : grad_off);
      if (full_cols) {
        pb[u] = *reinterpret_cast<const float4*>(src + i);
      } else {
        const int last = len - 1;
        pb[u].x = src[i < last ? i : last];
        pb[u].y = src[i + 1 < last ? i + 1 : last];
        pb[u].z = src[i + 2 < last ? i + 2 : last];
        pb[u].w = src[i + 3 < last ? i + 3 : last];
      }
    }
  };
  auto stash = [&](int ch) {
    const bool xhalf = ch >= nch;
    const int b0 = (xhalf ? ch - nch : ch) * PG_BK;
    const float sz = xhalf ? (seg_is_theta ? 0.f : -c2h) : 1.0f, st = xhalf ? (seg_is_theta ? -c2h : 0.f) : 1.0f;
    // (rows beyond Mloc hold garbage: their outputs are never written; columns beyond M must be exact zeros)
    const bool bok = b0 + lk < M;
#pragma unroll
    for (int j = 0; j < 8 * PG_RT; ++j) {
      const float val = JOINT ? fmaf(st, pt[j], sz * pz[j]) : sz * pz[j];
      As[(la + 8 * j) * PG_LDA + lk] = bok ? val : 0.f;
    }
    if (full_cols && b0 + PG_BK <= M) {  // (block-uniform)
#pragma unroll
      for (int u = 0; u < 2; ++u) *reinterpret_cast<float4*>(&Bs[(lb + 16 * u) * PG_LDB + lc]) = pb[u];
    } else {
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int i = i0 + lc;
        float4 q = pb[u];
        const bool rok = b0 + lb + 16 * u < M;
        q.x = (rok && i < len) ? q.x : 0.f;
        q.y = (rok && i + 1 < len) ? q.y : 0.f;
        q.z = (rok && i + 2 < len) ? q.z : 0.f;
        q.w = (rok && i + 3 < len) ? q.w : 0.f;
        *reinterpret_cast<float4*>(&Bs[(lb + 16 * u) * PG_LDB + lc]) = q;
      }
    }
  };
  f32x4 acc[PG_RT][4];
#pragma unroll
  for (int ti = 0; ti < PG_RT; ++ti)
#pragma unroll
    for (int tj = 0; tj < 4; ++tj) acc[ti][tj] = f32x4{0.f, 0.f, 0.f, 0.f};
  float rs[PG_RT];  // partial row sums of kr (a function of M only: chunks in ascending order, fixed quarters)
#pragma unroll
  for (int ti = 0; ti < PG_RT; ++ti) rs[ti] = 0.f;
  fetch(0);
  for (int ch = 0; ch < 2 * nch; ++ch) {
    __syncthreads();  // (the previous chunk has been read)
    stash(ch);
    __syncthreads();
    if (ch + 1 < 2 * nch) fetch(ch + 1);
    const float* A = As + (16 * PG_RT * wave + r) * PG_LDA + g;
    const float* B = Bs + g * PG_LDB + r;
    // fragments of k-step ks + 1 are read while the MFMAs of k-step ks run
    float fa[2][PG_RT], fb[2][4];
#pragma unroll
    for (int ti = 0; ti < PG_RT; ++ti) fa[0][ti] = A[16 * ti * PG_LDA];
#pragma unroll
    for (int tj = 0; tj < 4; ++tj) fb[0][tj] = B[16 * tj];
#pragma unroll
    for (int ks = 0; ks < PG_BK / 4; ++ks) {
      const int cur = ks & 1, nxt = cur ^ 1;
      if (ks + 1 < PG_BK / 4) {
#pragma unroll
        for (int ti = 0; ti < PG_RT; ++ti) fa[nxt][ti] = A[16 * ti * PG_LDA + 4 * (ks + 1)];
#pragma unroll
        for (int tj = 0; tj < 4; ++tj) fb[nxt][tj] = B[4 * (ks + 1) * PG_LDB + 16 * tj];
      }
      __builtin_amdgcn_sched_barrier(0);  // (hipcc otherwise sinks each read to its MFMA and waits for it there)
#pragma unroll
      for (int tj = 0; tj < 4; ++tj)
#pragma unroll
        for (int ti = 0; ti < PG_RT; ++ti)
          acc[ti][tj] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[cur][ti], fb[cur][tj], acc[ti][tj], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (ch >= nch) {  // row sums of kr: thread t adds the quarter t % 4 (8 columns) of row t / 4 (+ 64) of this chunk
#pragma unroll
      for (int ti = 0; ti < PG_RT; ++ti) {
        const float4 q0 = *reinterpret_cast<const float4*>(&As[((tid >> 2) + 64 * ti) * PG_LDA + 8 * (tid & 3)]);
        const float4 q1 = *reinterpret_cast<const float4*>(&As[((tid >> 2) + 64 * ti) * PG_LDA + 8 * (tid & 3) + 4]);
        rs[ti] -= ((q0.x + q0.y) + (q0.z + q0.w)) + ((q1.x + q1.y) + (q1.z + q1.w));
      }
    }
  }
  // rs_s[row] = sum of the four quarter sums, in quarter order
#pragma unroll
  for (int ti = 0; ti < PG_RT; ++ti) {
    float t = rs[ti];
    const float t1 = __shfl_xor(t, 1);
    t = (tid & 1) ? t1 + t : t + t1;        // (same operands, same order on both lanes)
    const float t2 = __shfl_xor(t, 2);
    t = (tid & 2) ? t2 + t : t + t2;
    if ((tid & 3) == 0) rs_s[(tid >> 2) + 64 * ti] = t;
  }
  __syncthreads();
  const float inv_m = 1.0f / (float)M;
#pragma unroll
  for (int ti = 0; ti < PG_RT; ++ti)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int al = 16 * PG_RT * wave + 16 * ti + 4 * g + q, a = a0 + al;
      if (a >= Mloc) continue;
      const float rsa = rs_s[al];
#pragma unroll
      for (int tj = 0; tj < 4; ++tj) {
        const int i = i0 + 16 * tj + r;
        if (i >= len) continue;
        const float xv = pack[(size_t)(m0 + a) * pack_stride + val_off + i];
        const float tot = fmaf(rsa, xv, acc[ti][tj][q]);
        const float phi = -tot * inv_m;
        const size_t o = (size_t)a * len + i;
        if (phi_out) phi_out[o] = phi;
        float xn;
        if (rmsprop) {
          const float vv = rmsprop_moment(v[o], phi);
          v[o] = vv;
          xn = rmsprop_step(xv, vv, phi, stepsize);
        } else {
          xn = gd_step(xv, phi, stepsize);
        }
        x[o] = xn;
        if (vout) vout[(size_t)a * vout_stride + vout_off + i] = xn;
      }
    }
}
