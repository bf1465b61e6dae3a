// acyclicity-constraint gradient on f32 MFMA (gfx950)
#pragma once
#include "common.h"
#include "kernels_lik.h"

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

