// acyclicity-constraint gradient on f32 MFMA (gfx950)
#pragma once
#include "common.h"

// ------------------------------------------------------------------------------------------------
// K5  acyclicity gradient: for Gumbel-soft graphs G~ = sigmoid(tau (eps + alpha s)), M = I + G~/d,
//     dh/dG~ = (M^{d-1})^T (h = tr(M^d) - d), chained through G~.  Matrix powers on f32 MFMA, all operands
//     resident in LDS.  Each block handles CPB chains of one particle and writes their SUM; k_acyc_reduce adds the partial sums.
//     reference: graph_utils.py:8-28, dibs.py:121-140, 557-601
// grid = (ceil(Sa / CPB), Mloc), block = 256; dynamic LDS = 3 * DP * LD * 4, DP = 16 NT, LD = DP + 2
//
// Serves d <= 32 and 65 <= d <= 112 (and unpaired PRNG layouts); 33 <= d <= 64 runs on kernels_acyc_bf16.h.  At the headline size
// (d = 50, measured while this was the production kernel): 5.96 M MFMAs = 186 k MFMA cycles per SIMD + 20.1 M other vector
// instructions = 78 k cycles of the ~350 k the launch takes (141.5 us, 32 % of the FP32 peak; f32 MFMAs do not overlap with the
// vector work of co-resident waves).  Measured on the box and reverted:
//  * border strips on v_mfma_f32_4x4x1_16b_f32 (3 x 3 core tiles on three waves, rows / columns 48-51 as 16 independent 4 x 4 blocks
//    per instruction on the fourth wave): MFMA cycles per product 6 656 -> 4 576, correct, but 148 us -- the border wave has 8 issue
//    cycles per k-step to hide four LDS operand loads behind; with deeper prefetch 198-222 VGPRs, two blocks per CU, 163-173 us;
//  * unpadded 52-row buffers with an XOR swizzle (39 KiB per block, four blocks per CU, the 2 048 pair-units in exactly two rounds):
//    148 us -- the launch is bound by issue per SIMD, not by the tail round;
//  * A fragments as one ds_read_b128 per four k-steps (conflict-free): 148 us -- LDS is not the limiter either;
//  * a second stream beside the BGe kernels: -1 % (it pays with the bf16 kernel, whose MFMAs do overlap with vector work).
// ------------------------------------------------------------------------------------------------
// Matrices live in LDS as [DP rows][LD] with the COLUMNS PERMUTED: logical column c sits at pc(c) = (c & 15) * NT + (c >> 4),
// so the NT values {c, c+16, c+32, ...} that one lane needs for the B fragments of a k-step (and produces in the C tile)
// are contiguous: one ds_read_b128 / ds_write_b128 for NT = 4.  LD = 16 NT + 4.
template <int NT>
__device__ __forceinline__ int acyc_pc(int c) { return (c & 15) * NT + (c >> 4); }

// Reduction of the per-block partial sums part[m][blk][d*d] over the blocks of a particle, in block order (fixed: bit-reproducible,
// independent of the grid), scaled by 1 / Sa.  A launch of its own behind the acyclicity kernel ON ITS STREAM, i.e. beside the BGe kernels
// and off the critical path; the consumer (k_particle_grad) then reads one value per element instead of one per block.
// Measured alternatives inside the acyclicity launch (last block of a particle to draw a ticket reduces): with agent-scope release /
// acquire fences around the ticket 98 -> 427 us per launch (every fence writes back / invalidates the XCD's whole L2: buffer_wbl2 /
// buffer_inv sc1); with sc1 (agent-scope) stores and loads of the partial sums and no fence 98 -> 121 us.  The kernel boundary is cheaper.
// grid = (Mloc, ceil(d*d / 256)), block = 256
#ifdef DIBS_TU_ACYC
__global__ __launch_bounds__(256) void k_acyc_reduce(const float* __restrict__ part, float* __restrict__ w_acyc, int nblk, int dd, float inv_sa) {
  const int m = blockIdx.x, e = blockIdx.y * 256 + threadIdx.x;
  if (e >= dd) return;
  const float* pm = part + (size_t)m * nblk * dd + e;
  float ac = 0.f;
  int q = 0;
  for (; q + 16 <= nblk; q += 16) {  // loads in flight together, additions in block order
    float v[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) v[u] = pm[(size_t)(q + u) * dd];
#pragma unroll
    for (int u = 0; u < 16; ++u) ac += v[u];
  }
  {
    float v[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) v[u] = q + u < nblk ? pm[(size_t)(q + u) * dd] : 0.f;
#pragma unroll
    for (int u = 0; u < 16; ++u) ac += q + u < nblk ? v[u] : 0.f;
  }
  w_acyc[(size_t)m * dd + e] = ac * inv_sa;
}
#endif

// k-steps S .. KS-1 of one tile row as a template recursion (see lds_matmul)
template <int NT, int KS, int S>
__device__ __forceinline__ void acyc_steps(f32x4 (&acc)[NT], float (&af)[2], float (&bf)[2][NT], const float* ap, const float* bq) {
  constexpr int DP = 16 * NT, LD = DP + 4;
  if constexpr (S == 0) {  // prologue: fragments of step 0
    af[0] = ap[0];
    if constexpr (NT == 4) {
      const float4 t4 = *reinterpret_cast<const float4*>(bq);
      bf[0][0] = t4.x; bf[0][1] = t4.y; bf[0][2] = t4.z; bf[0][3] = t4.w;
    } else {
#pragma unroll
      for (int tj = 0; tj < NT; ++tj) bf[0][tj] = bq[tj];
    }
  }
  if constexpr (S + 1 < KS) {  // fragments of step S + 1
    constexpr int kk0 = (S + 1) << 2, n = (S + 1) & 1;
    af[n] = ap[(kk0 & 15) * NT + (kk0 >> 4)];
    if constexpr (NT == 4) {
      const float4 t4 = *reinterpret_cast<const float4*>(bq + kk0 * LD);
      bf[n][0] = t4.x; bf[n][1] = t4.y; bf[n][2] = t4.z; bf[n][3] = t4.w;
    } else {
#pragma unroll
      for (int tj = 0; tj < NT; ++tj) bf[n][tj] = bq[kk0 * LD + tj];
    }
  }
  constexpr int c = S & 1;
#pragma unroll
  for (int tj = 0; tj < NT; ++tj) {
    if constexpr (S == 0) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, 0" : "=&v"(acc[tj]) : "v"(af[c]), "v"(bf[c][tj]));
    else asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[tj]) : "v"(af[c]), "v"(bf[c][tj]));
  }
  if constexpr (S + 1 < KS) acyc_steps<NT, KS, S + 1>(acc, af, bf, ap, bq);
}

// C = A * B.  Operands are OFFSETS (in floats) into the kernel's LDS array so that every access is a ds_* instruction
// (a runtime-selected generic pointer would turn them into flat accesses).
// KS = number of k-steps (ceil(d / 4)), a template parameter: the whole tile row is straight-line code -- every LDS offset is
// an immediate, no loop counter, no address arithmetic (this kernel's time is the SUM of its MFMA and VALU issue cycles), and the
// accumulators never meet a control-flow join between an MFMA and the s_nop that covers its latency (hipcc places register
// copies for a join right behind the opaque asm MFMA and would read the accumulator too early).
// The fragments of step s + 1 are loaded before the MFMAs of step s issue; the first MFMA of every tile takes C = 0 as an inline
// constant instead of a zeroed accumulator.
template <int NT, int KS>
__device__ __forceinline__ void lds_matmul(float* __restrict__ lds, int c_off, int a_off, int b_off, int lane, int wave) {
  constexpr int DP = 16 * NT, LD = DP + 4;
  for (int ti = wave; ti < NT; ti += 4) {
    f32x4 acc[NT];
    // A[row][k]: k = k0 + kk -> physical ((k0 & 15) + kk) * NT + (k0 >> 4)
    const float* ap = lds + a_off + (ti * 16 + (lane & 15)) * LD + (lane >> 4) * NT;
    // B[k][tj * 16 + col], tj = 0..NT-1 -> physical col * NT + tj (contiguous)
    const float* bq = lds + b_off + (lane >> 4) * LD + (lane & 15) * NT;
    float af[2], bf[2][NT];
    // MFMA as inline asm with the accumulator tied in place: with the builtin, hipcc renamed the accumulators across the
    // pipelined steps (v_accvgpr_read / _mov / _write + s_nop).  Hazards hipcc cannot see around asm: last MFMA -> accumulator
    // read (s_nop below); back-to-back MFMAs on the same accumulator need none.
    acyc_steps<NT, KS, 0>(acc, af, bf, ap, bq);
    // last MFMA -> accumulator read: one wait for the whole group (volatile asm statements keep their order, so every
    // accumulator's first read sits behind the s_nop)
    asm volatile("s_nop 15\n\ts_nop 7" : "+v"(acc[NT - 1]));
#pragma unroll
    for (int tj = 0; tj < NT - 1; ++tj) asm volatile("" : "+v"(acc[tj]));
    // C tile -> LDS straight from the accumulator registers (ds_write2_b32 takes two arbitrary data VGPRs: no gather moves into a
    // register quad for ds_write_b128).  The stores are asm: hipcc (ROCm 7.2) emitted `ds_write_b32 vaddr, aN` (AGPR data operand)
    // for part of such a tile, which stored wrong values on gfx950 (scripts/probe/acyc_probe.hip); "v" constraints rule that out.
    // LDS byte address of row r = 0 (the dynamic array does not start at 0: static __shared__ variables precede it)
    typedef __attribute__((address_space(3))) float lds_float;
    const unsigned o4 = (unsigned)(uintptr_t)(lds_float*)(lds + c_off + (ti * 16 + (lane >> 4) * 4) * LD + (lane & 15) * NT);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      // (the offset fields hold 8 bits of dwords: rows beyond that get their own address register)
      constexpr bool imm = 3 * LD + NT <= 256;
      const unsigned orow = imm ? o4 : o4 + r * LD * 4;
      const int ro = imm ? r * LD : 0;
#pragma unroll
      for (int tj = 0; tj + 1 < NT; tj += 2)
        asm volatile("ds_write2_b32 %0, %1, %2 offset0:%3 offset1:%4" ::"v"(orow), "v"(acc[tj][r]), "v"(acc[tj + 1][r]), "n"(imm ? r * LD + tj : tj),
                     "n"(imm ? r * LD + tj + 1 : tj + 1)
                     : "memory");
      if constexpr (NT & 1) asm volatile("ds_write_b32 %0, %1 offset:%2" ::"v"(orow), "v"(acc[NT - 1][r]), "n"(((imm ? r * LD : 0) + NT - 1) * 4) : "memory");
      (void)ro;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (hipcc does not count asm LDS operations)
  }
}

// a matrix of this tile count has 4 NT - 3 .. 4 NT k-steps: one straight-line instantiation each (block-uniform dispatch)
template <int NT>
__device__ __forceinline__ void acyc_matmul(float* __restrict__ lds, int c_off, int a_off, int b_off, int ksteps, int lane, int wave) {
  if constexpr (NT == 1) {
    switch (ksteps) {
      case 1: lds_matmul<NT, 1>(lds, c_off, a_off, b_off, lane, wave); break;
      case 2: lds_matmul<NT, 2>(lds, c_off, a_off, b_off, lane, wave); break;
      case 3: lds_matmul<NT, 3>(lds, c_off, a_off, b_off, lane, wave); break;
      default: lds_matmul<NT, 4>(lds, c_off, a_off, b_off, lane, wave); break;
    }
  } else {
    switch (ksteps - (4 * NT - 3)) {
      case 0: lds_matmul<NT, 4 * NT - 3>(lds, c_off, a_off, b_off, lane, wave); break;
      case 1: lds_matmul<NT, 4 * NT - 2>(lds, c_off, a_off, b_off, lane, wave); break;
      case 2: lds_matmul<NT, 4 * NT - 1>(lds, c_off, a_off, b_off, lane, wave); break;
      default: lds_matmul<NT, 4 * NT>(lds, c_off, a_off, b_off, lane, wave); break;
    }
  }
}

template <int NT, bool PAIRED>
__global__ __launch_bounds__(256) void k_acyc(const float* __restrict__ scores, float* __restrict__ part, Key2 carry, int m0,
                                              int M_global, int d, int Sa, int cpb, float alpha, float tau, int layout,
                                              int tiny, int n_acyc_blk) {
  constexpr int DP = 16 * NT, LD = DP + 4, BUF = DP * LD;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int blk = blockIdx.x, m = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const Key2 km = rng_split_row_uniform(carry, (uint32_t)M_global + 1u, (uint32_t)(m0 + m) + 1u, layout);  // dibs.py:595: key used directly
  const uint64_t dd = (uint64_t)d * d, nbits = (uint64_t)Sa * dd;
  const int kp = (d + 3) & ~3;
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
                // (v_rcp_f32, 1 ulp: the IEEE division sequence is ten instructions per draw, and this kernel's time is the SUM of its
                //  MFMA and VALU issue cycles)
                // saturated edges (exp(-alpha s) below the rounding of the denominator) give exactly 1, as the reference's sigmoid does
                const float den0 = fmaf(1.0f - u0, ea, u0), den1 = fmaf(1.0f - u1, ea, u1);
                g = den0 == u0 ? 1.0f : u0 * __builtin_amdgcn_rcpf(den0);
                gnext[q] = den1 == u1 ? 1.0f : u1 * __builtin_amdgcn_rcpf(den1);
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
          acyc_matmul<NT>(smem, dst, cur, cur, kp >> 2, lane, wave);
          __syncthreads();
          cur = dst;
          if ((ex >> b) & 1) {
            dst = (cur == BUF) ? 2 * BUF : BUF;
            acyc_matmul<NT>(smem, dst, cur, 0, kp >> 2, lane, wave);
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
          float g = smem[i * LD + pcj] * fd;
          g = g > 0.99999988f ? 1.0f : g;  // (g / d) * d of a saturated edge may round to 1 - 2^-24 or 1 + 2^-23: g (1 - g) has to be 0 there
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

