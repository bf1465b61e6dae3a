// Soft-graph BGe estimator (grad_estimator_z = "reparam" of MarginalDiBS) for n_vars <= 64 as a BLOCKED factorisation on the matrix pipe.
//   reference: dibs/inference/dibs.py:395-459, dibs/models/linearGaussian.py:63-170 with a real-valued parent vector, dibs/utils/func.py:128-145;
//   closed forms: header of kernels_bge_soft.h (the same quantities: logdet M_pa, the Schur complement s, diag(M_pa^-1), y = M_pa^-1 b).
// Round 3's k_bge_soft_reg (one matrix row per lane, 64-step column loop; retired) issued one LDS broadcast per multiply-add and ran two waves per SIMD:
// 17 ms per step at the headline size (819 200 factorisations of 50 x 50 matrices).  Here one wave owns one (sample, node) problem as
// 16 x 16 blocks in the accumulator layout of v_mfma_f32_16x16x4_f32 (lane (g, c) = (lane / 16, lane % 16), register r: element
// (4 g + r, c) of the block), and everything of order n^3 is an MFMA whose operands are those registers:
//   * node j is ordered LAST (positions 0 .. d-2: the other variables, position d-1: j with p_j := 1, then identity padding to 16 NB).
//     The factorisation of that matrix M_all gives logdet M_pa = sum of the leading log-pivots, s = the last pivot, w = L^-1 b = the last
//     row of L; the last row of T = L^-1 is -y^T / sqrt(s); diag(M_pa^-1)_c = sum_{k >= c, k != d-1} T[k][c]^2.
//   * upper-block right-looking factorisation A = U^T U.  Panel k:  the diagonal block goes through 1.3 KB of LDS into rows (lane c of
//     every 16-lane row holds row c; the four rows of lanes work redundantly), is factorised and inverted there with DPP row broadcasts
//     (row_newbcast), T_kk = L_kk^-1 returns through LDS;   U_ki = T_kk A_ki   (A operand: T_kk rows from LDS, B operand: the block's
//     registers);   A_ij -= U_ki^T U_kj   (both operands are registers: register r of a block X in accumulator layout IS the A operand of
//     X^T for the k-slab {r, 4 + r, 8 + r, 12 + r}, and of Y the B operand for the same slab);
//     block row k of T:   T_ki = -T_kk sum_{l = i .. k-1} U_lk^T T_li.
//   * the matrix is kept as A - I (diagonal entries p^2 (R_vv - 1)): pivots minus one stay exact for small p, as in k_bge_soft_reg.
// LDS per wave 3.3 KB (k_bge_soft_reg: 13.4 KB): occupancy is set by registers.  Block algebra checked in tests/tools/bge_soft_blocked_emulation.py.
#pragma once
#include "kernels_bge_soft.h"

#define BSM_LDP 20  // row stride (floats) of the 16 x 16 LDS blocks: 16-byte rows, lanes of a 16-lane group on distinct banks
__host__ __device__ inline size_t bsm_wave_bytes() { return ((size_t)5 * 16 * BSM_LDP + 4 * 64) * 4; }  // A_kk | T_kk, k < 4 | p by position | p y by variable | variable of a position (x d, x 1)
__host__ __device__ inline size_t bsm_lds_bytes(int d, bool r_in_lds) { return bge_soft_shared_bytes(d, r_in_lds) + 64 * 8 + 4 * bsm_wave_bytes(); }  // R, red | c_j | waves

// log(x) of a positive float with ~1e-7 ABSOLUTE error in float arithmetic (the scheme of bge_log, kernels_bge.h): x = 2^e m,
// m in [sqrt(1/2), sqrt(2)), log m = 2 atanh((m - 1) / (m + 1))
__device__ __forceinline__ double bsm_log(float x) {
  int e;
  float m = frexpf(x, &e);
  if (m < 0.70710678f) {
    m *= 2.0f;
    e -= 1;
  }
  const float r = (m - 1.0f) / (m + 1.0f), r2 = r * r;
  const float p = fmaf(r2, fmaf(r2, fmaf(r2, fmaf(r2, fmaf(r2, 2.0f / 11.0f, 2.0f / 9.0f), 2.0f / 7.0f), 2.0f / 5.0f), 2.0f / 3.0f), 2.0f);
  return (double)e * 0.6931471805599453 + (double)(r * p);
}
// Stirling tails at z >= 8: lgamma(z) = (z - 1/2) log z - z + 1/2 log 2 pi + bsm_ser_lgamma(z), digamma(z) = log z + bsm_ser_digamma(z)
__device__ __forceinline__ double bsm_ser_lgamma(double z) {
  const double r = 1.0 / z, f = r * r;
  return r * (1.0 / 12.0 - f * (1.0 / 360.0 - f * (1.0 / 1260.0 - f * (1.0 / 1680.0 - f * (1.0 / 1188.0)))));
}
__device__ __forceinline__ double bsm_ser_digamma(double z) {
  const double r = 1.0 / z, f = r * r;
  return -0.5 * r - f * (1.0 / 12.0 - f * (1.0 / 120.0 - f * (1.0 / 252.0 - f * (1.0 / 240.0 - f * (1.0 / 132.0)))));
}

// DPP row broadcasts as hand-placed instructions: written with the builtin (v_mov_b32_dpp + v_fma_f32) hipcc hoists the 120 broadcasts of a
// 16 x 16 substitution in front of their uses -- 256 VGPRs + 215 AGPRs, one wave per SIMD.  Volatile asm statements keep their order and
// fuse the broadcast into the multiply-add; the price is that hipcc does not see the hazards of what is inside them: a DPP read needs two
// wait states after the VALU write of its source (s_nop 1 where a statement reads what the statement before it wrote).
// acc -= (value of `b` in lane n of this lane's row of 16) * own
__device__ __forceinline__ void bsm_fmac_bc(float& acc, const float& b, const float& own, int n) {
#define BSM_FB(N_) case N_: asm volatile("v_fmac_f32_dpp %0, -%1, %2 row_newbcast:" #N_ " row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(b), "v"(own)); break;
  switch (n) {
    BSM_FB(0) BSM_FB(1) BSM_FB(2) BSM_FB(3) BSM_FB(4) BSM_FB(5) BSM_FB(6) BSM_FB(7)
    BSM_FB(8) BSM_FB(9) BSM_FB(10) BSM_FB(11) BSM_FB(12) BSM_FB(13) BSM_FB(14)
    default: asm volatile("v_fmac_f32_dpp %0, -%1, %2 row_newbcast:15 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(b), "v"(own)); break;
  }
#undef BSM_FB
}
// value of `v` in lane n of this lane's row of 16.  NOP: two wait states in front (`v` may have been written by the statement before)
template <bool NOP>
__device__ __forceinline__ float bsm_bcast(const float& v, int n) {
  float o;
#define BSM_BC(N_)                                                                                                                    \
  case N_:                                                                                                                            \
    if (NOP) asm volatile("s_nop 1\n\tv_mov_b32_dpp %0, %1 row_newbcast:" #N_ " row_mask:0xf bank_mask:0xf" : "=v"(o) : "v"(v));      \
    else asm volatile("v_mov_b32_dpp %0, %1 row_newbcast:" #N_ " row_mask:0xf bank_mask:0xf" : "=v"(o) : "v"(v));                     \
    break;
  switch (n) {
    BSM_BC(0) BSM_BC(1) BSM_BC(2) BSM_BC(3) BSM_BC(4) BSM_BC(5) BSM_BC(6) BSM_BC(7)
    BSM_BC(8) BSM_BC(9) BSM_BC(10) BSM_BC(11) BSM_BC(12) BSM_BC(13) BSM_BC(14)
    default:
      if (NOP) asm volatile("s_nop 1\n\tv_mov_b32_dpp %0, %1 row_newbcast:15 row_mask:0xf bank_mask:0xf" : "=v"(o) : "v"(v));
      else asm volatile("v_mov_b32_dpp %0, %1 row_newbcast:15 row_mask:0xf bank_mask:0xf" : "=v"(o) : "v"(v));
      break;
  }
#undef BSM_BC
  return o;
}
// x = lane in `mask` ? y : x for a CONSTANT lane set: the mask is moved into vcc by two scalar instructions in front of the v_cndmask -- one
// vector instruction where a test on the lane index costs v_cmp + v_cndmask.  (As "s" operands the 80 masks of a problem were hoisted out
// of the node loop into 160 SGPRs, i.e. parked in VGPR lanes: a v_readlane pair per use.)
__device__ __forceinline__ void bsm_sel(float& x, const float& y, uint64_t mask) {
  asm("s_mov_b32 vcc_lo, %2\n\ts_mov_b32 vcc_hi, %3\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(x) : "v"(y), "i"((uint32_t)mask), "i"((uint32_t)(mask >> 32)) : "vcc");
}
// x = lane in `mask` ? 1 : 0
__device__ __forceinline__ float bsm_ind(uint64_t mask) {
  float x;
  asm("s_mov_b32 vcc_lo, %1\n\ts_mov_b32 vcc_hi, %2\n\tv_cndmask_b32 %0, 0, 1.0, vcc" : "=v"(x) : "i"((uint32_t)mask), "i"((uint32_t)(mask >> 32)) : "vcc");
  return x;
}
__device__ __forceinline__ void bsm_nops(int n) {  // n wait states
  if (n == 1) asm volatile("s_nop 0");
  if (n == 2) asm volatile("s_nop 1");
}

// grid = (S, Mloc), block = 256 (one node per wave and pass); dynamic LDS = bsm_lds_bytes()
template <int NB, bool R_LDS, int WPE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void k_bge_soft_mf(const float* __restrict__ scores, BgeSoftParams bp, Key2 carry, int m0, int M_global,
                                                     int d, int S, float alpha, float tau, int layout, int tiny,
                                                     float* __restrict__ ds_out, float* __restrict__ logprobs) {
  constexpr int LDP = BSM_LDP;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int s = blockIdx.x, m = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, c = lane & 15;
  const int dd = d * d, dl1 = d - 1;
  float* Rs = reinterpret_cast<float*>(smem_raw);
  double* red = reinterpret_cast<double*>(smem_raw + bge_soft_shared_bytes(d, R_LDS) - 256);
  double* cj = reinterpret_cast<double*>(smem_raw + bge_soft_shared_bytes(d, R_LDS));  // per node: the terms of the score that depend on j only
  float* Ab = reinterpret_cast<float*>(smem_raw + bge_soft_shared_bytes(d, R_LDS) + 64 * 8 + (size_t)wave * bsm_wave_bytes());
  float* Tb = Ab + 16 * LDP;
  float* pvs = Tb + 4 * 16 * LDP;
  float* pyv = pvs + 64;
  int* vmd = reinterpret_cast<int*>(pyv + 64);  // variable at a position, times d (row offset into R); padding positions: j
  int* vmc = vmd + 64;
  const float* sc_m = scores + (size_t)m * dd;
  const Key2 key = lin_mode_key(LIN_MODE_Z_REPARAM, carry, M_global, m0 + m, layout);  // dibs.py:430-431
  const uint64_t nbits = (uint64_t)S * dd;
  if (R_LDS)
    for (int e = tid; e < dd; e += 256) Rs[e] = bp.R[e];
  if (tid < 4) red[tid] = 0.0;
  if (tid < d) cj[tid] = 0.5 * (log(bp.alpha_mu) - log(bp.Nj[tid] + bp.alpha_mu)) - 0.5 * bp.Nj[tid] * log(M_PI);
  __syncthreads();
  float* out = ds_out + ((size_t)m * S + s) * dd;
  double lp_wave = 0.0;
  const bool act = lane < d;
  for (int j = wave; j < d; j += 4) {
    const float* R = R_LDS ? Rs : bp.R + (bp.n_mats > 1 ? (size_t)j * dd : 0);
    // position -> variable: j and d-1 swap places
    auto var_of = [&](int ps) { return ps == j ? dl1 : (ps == dl1 ? j : ps); };
    const int v = act ? var_of(lane) : 0;
    const float gs = act ? lin_sample_g(LIN_MODE_Z_REPARAM, key, nbits, (uint64_t)dd, s, v, j, d, nullptr, sc_m, alpha, tau, layout, tiny) : 0.f;
    const float p = gs;  // column j of the soft graph (dibs.py:121-140); 0 for v == j and on the padding positions
    const double l = wave_sum_d((double)p);
    pvs[lane] = act ? (lane == dl1 ? 1.f : p) : 0.f;
    vmc[lane] = act ? v : j;
    vmd[lane] = (act ? v : j) * d;
    wave_lds_fence();
    // ---- A - I in upper blocks, accumulator layout ----------------------------------------------------------------------------------
    f32x4 acc[NB][NB];
    {
      f32x4 prow[NB];
      int rrow[NB][4];
#pragma unroll
      for (int i = 0; i < NB; ++i) {
        prow[i] = *reinterpret_cast<const f32x4*>(pvs + 16 * i + 4 * g);
        const int4 r4 = *reinterpret_cast<const int4*>(vmd + 16 * i + 4 * g);
        rrow[i][0] = r4.x; rrow[i][1] = r4.y; rrow[i][2] = r4.z; rrow[i][3] = r4.w;
      }
#pragma unroll
      for (int jb = 0; jb < NB; ++jb) {
        const float pc = pvs[16 * jb + c];
        const int vc = vmc[16 * jb + c];
#pragma unroll
        for (int i = 0; i <= jb; ++i)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float rv = R[rrow[i][r] + vc];
            if (i == jb) rv -= 4 * g + r == c ? 1.f : 0.f;
            acc[i][jb][r] = prow[i][r] * pc * rv;
          }
      }
    }
    float dm1 = 0.f;  // pivot - 1 of this lane's position
#pragma unroll
    for (int k = 0; k < NB; ++k) {
      // ---- diagonal block: registers -> rows (lane c of every 16-lane row: row c) ---------------------------------------------------
#pragma unroll
      for (int r = 0; r < 4; ++r) Ab[(4 * g + r) * LDP + c] = acc[k][k][r];
      wave_lds_fence();
      float Sr[16], invs[16], Tr[16];
#pragma unroll
      for (int q = 0; q < 16; q += 4) {
        const f32x4 t4 = *reinterpret_cast<const f32x4*>(Ab + c * LDP + q);
        Sr[q] = t4[0]; Sr[q + 1] = t4[1]; Sr[q + 2] = t4[2]; Sr[q + 3] = t4[3];
      }
      // rows of this diagonal block that are not identity padding (wave-uniform; only the last block has any): a padding step has pivot 1 and
      // a zero column -- skipped, exactly
      const int nreal = k == NB - 1 ? d - 16 * (NB - 1) : 16;
      if (k < NB - 1) {
        // Full block, software-pipelined: column t + 1 is final after the FIRST multiply-add of step t, so its pivot chain (broadcast, rsq,
        // scaling) is issued between the remaining multiply-adds of step t instead of stalling the wave four dependent instructions long.
        // Wait states: two between a write and a DPP read of the register, one after rsq; multiply-adds in between count.
        float lv;
        {
          const float pm = bsm_bcast<true>(Sr[0], 0), inv = __builtin_amdgcn_rsqf(1.0f + pm);
          invs[0] = inv;
          bsm_sel(dm1, pm, 1ull << (16 * k));
          asm volatile("s_nop 0\n\tv_mul_f32 %0, %1, %2\n\ts_nop 1" : "=v"(lv) : "v"(Sr[0]), "v"(inv));
          Sr[0] = lv;
        }
#pragma unroll
        for (int t = 0; t < 15; ++t) {
          // multiply-adds of step t: columns t + 1 | b0 .. | c0 .. | d0 .. 15 around the pivot chain of column t + 1 (bounds are constants
          // of the unrolled step)
          const int eb = 14 - t < 2 ? 14 - t : 2, b0 = t + 2, c0 = b0 + eb, ec = 14 - t - eb < 6 ? 14 - t - eb : 6, d0 = c0 + ec, ed = 16 - d0;
          bsm_fmac_bc(Sr[t + 1], Sr[t], Sr[t], t + 1);
#pragma unroll
          for (int c2 = b0; c2 < c0; ++c2) bsm_fmac_bc(Sr[c2], Sr[t], Sr[t], c2);
          bsm_nops(2 - eb);
          const float pm = bsm_bcast<false>(Sr[t + 1], t + 1), inv = __builtin_amdgcn_rsqf(1.0f + pm);
          invs[t + 1] = inv;
          bsm_sel(dm1, pm, 1ull << (16 * k + t + 1));
#pragma unroll
          for (int c2 = c0; c2 < d0; ++c2) bsm_fmac_bc(Sr[c2], Sr[t], Sr[t], c2);
          asm volatile("s_nop 0\n\tv_mul_f32 %0, %1, %2" : "=v"(lv) : "v"(Sr[t + 1]), "v"(inv));
#pragma unroll
          for (int c2 = d0; c2 < 16; ++c2) bsm_fmac_bc(Sr[c2], Sr[t], Sr[t], c2);
          bsm_nops(ed >= 2 ? 0 : 2 - ed);
          Sr[t + 1] = lv;
        }
      } else {
#pragma unroll
        for (int t = 0; t < 16; ++t) {
          invs[t] = 1.0f;
          if (t < nreal) {
            const float pivm1 = bsm_bcast<true>(Sr[t], t), inv = __builtin_amdgcn_rsqf(1.0f + pivm1);
            invs[t] = inv;
            bsm_sel(dm1, pivm1, 1ull << (16 * k + t));
            // column t of L below the diagonal (the diagonal entry itself and the rows above it are never read again); s_nop: a VALU read of
            // a transcendental result needs a wait state, and the statements below read the product through DPP (two)
            float lv;
            asm volatile("s_nop 0\n\tv_mul_f32 %0, %1, %2\n\ts_nop 1" : "=v"(lv) : "v"(Sr[t]), "v"(inv));
            Sr[t] = lv;
#pragma unroll
            for (int c2 = t + 1; c2 < 16; ++c2) bsm_fmac_bc(Sr[c2], Sr[t], Sr[t], c2);
          }
        }
      }
      // T_kk = L_kk^-1, row c per lane, right-looking: T L = I; once column m of T is final (m = 15 .. 0) every earlier column receives its
      // term -L[m][q] T[c][m] -- independent multiply-adds (the column-by-column form is one dependent chain per column).  Padding columns
      // are the identity's.
#pragma unroll
      for (int q = 0; q < 16; ++q) Tr[q] = bsm_ind(0x0001000100010001ull << q);
#pragma unroll
      for (int mm = 15; mm >= 0; --mm) {
        if (mm < nreal) {
          Tr[mm] *= invs[mm];
#pragma unroll
          for (int q = mm - 1; q >= 0; --q) bsm_fmac_bc(Tr[q], Sr[q], Tr[mm], mm);  // T[c][q] -= L[mm][q] T[c][mm]
        }
      }
      float* Tk = Tb + k * 16 * LDP;
      if (g == 0) {
#pragma unroll
        for (int q = 0; q < 16; q += 4) *reinterpret_cast<f32x4*>(Tk + c * LDP + q) = f32x4{Tr[q], Tr[q + 1], Tr[q + 2], Tr[q + 3]};
      }
      wave_lds_fence();
      if (k + 1 < NB) {
        const f32x4 top = *reinterpret_cast<const f32x4*>(Tk + c * LDP + 4 * g);  // A operand of T_kk: [row c][4 g + s]
        // ---- U_ki = T_kk A_ki ----------------------------------------------------------------------------------------------------------
#pragma unroll
        for (int i = k + 1; i < NB; ++i) {
          f32x4 u = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int q = 0; q < 4; ++q) u = __builtin_amdgcn_mfma_f32_16x16x4f32(top[q], acc[k][i][q], u, 0, 0, 0);
          acc[k][i] = u;
        }
        // ---- A_ij -= U_ki^T U_kj -------------------------------------------------------------------------------------------------------
#pragma unroll
        for (int i = k + 1; i < NB; ++i) {
          const f32x4 nu = -acc[k][i];
#pragma unroll
          for (int jb = i; jb < NB; ++jb)
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[i][jb] = __builtin_amdgcn_mfma_f32_16x16x4f32(nu[q], acc[k][jb][q], acc[i][jb], 0, 0, 0);
        }
      }
    }
    // ---- T = L^-1 block row by block row (after the factorisation: its blocks and the trailing matrix are not live together, which is what
    // ---- lets three waves per SIMD hold their registers): T_ki = -T_kk sum_{l = i .. k-1} U_lk^T T_li ------------------------------------
    f32x4 T[NB][NB];  // blocks (row block, column block <= row block)
#pragma unroll
    for (int k = 0; k < NB; ++k) {
      const float* Tk = Tb + k * 16 * LDP;
#pragma unroll
      for (int r = 0; r < 4; ++r) T[k][k][r] = Tk[(4 * g + r) * LDP + c];
      if (k > 0) {
        const f32x4 top = *reinterpret_cast<const f32x4*>(Tk + c * LDP + 4 * g);
#pragma unroll
        for (int i = 0; i < k; ++i) {
          f32x4 sa = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int l2 = i; l2 < k; ++l2)
#pragma unroll
            for (int q = 0; q < 4; ++q) sa = __builtin_amdgcn_mfma_f32_16x16x4f32(acc[l2][k][q], T[l2][i][q], sa, 0, 0, 0);
          f32x4 tn = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int q = 0; q < 4; ++q) tn = __builtin_amdgcn_mfma_f32_16x16x4f32(-top[q], sa[q], tn, 0, 0, 0);
          T[k][i] = tn;
        }
      }
    }
    wave_lds_fence();  // (Ab / Tb are rewritten by the next node)
    // ---- per position: log-pivots, off-diagonal column norms of T without the row of j, the row of j ---------------------------------
    const float sf = 1.0f + __shfl(dm1, dl1, 64);  // s = R_jj - b^T M_pa^-1 b: the last pivot
    const double sch = (double)sf;
    float offd = 0.f, yrow = 0.f;
    {
      // row d-1 (node j) lives in block row jb, in the lanes / register with 4 g + r == (d-1) % 16: taken out as the y row and left out of
      // the column norms; only that block row pays for the test (wave-uniform branch)
      const int jb = dl1 >> 4, rl = dl1 & 15;
      const bool mr[4] = {4 * g == rl, 4 * g + 1 == rl, 4 * g + 2 == rl, 4 * g + 3 == rl};
      const bool lo[4] = {4 * g > c, 4 * g + 1 > c, 4 * g + 2 > c, 4 * g + 3 > c};  // below the diagonal inside a diagonal block
#pragma unroll
      for (int i = 0; i < NB; ++i) {
        float o = 0.f, yv = 0.f;
#pragma unroll
        for (int jr = i; jr < NB; ++jr) {
          f32x4 tv = T[jr][i];
          if (jr == jb) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              yv = mr[r] ? tv[r] : yv;
              tv[r] = mr[r] ? 0.f : tv[r];
            }
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) o = fmaf((jr > i || lo[r]) ? tv[r] : 0.f, tv[r], o);
        }
        o += __shfl_xor(o, 16, 64);
        o += __shfl_xor(o, 32, 64);
        yv += __shfl_xor(yv, 16, 64);
        yv += __shfl_xor(yv, 32, 64);
        offd = g == i ? o : offd;
        yrow = g == i ? yv : yrow;
      }
    }
    const float y = lane < dl1 ? -yrow * sqrtf(sf) : 0.f;  // y = M_pa^-1 b by position (0 for j itself: row j of M_pa is the identity, b_j = 0)
    if (act) pyv[v] = p * y;
    wave_lds_fence();
    // The scalars of the score -- lgamma(a1) - lgamma(a2), digamma(a1) - digamma(a2), log s, the log-pivots -- with ONE double logarithm and one
    // reciprocal per lane instead of every lane evaluating lgamma / digamma (1 800 of the 8 000 instructions of a problem before):
    // lgamma(a) = lgamma(a + 8) - sum_{i < 8} log(a + i), digamma(a) = digamma(a + 8) - sum_{i < 8} 1 / (a + i), Stirling series at a + 8;
    // lanes 0 .. 7: a1 + i, 8 .. 15: a2 + i, 16 / 17: a1 + 8 / a2 + 8, 18: s.
    const double Nn = bp.Nj[j], al = bp.alpha_lambd;
    const double a1 = 0.5 * (Nn + al - d + l + 1.0), a2 = 0.5 * (al - d + l + 1.0), c2 = a1;
    const double arg = lane < 8 ? a1 + lane : lane < 16 ? a2 + (lane - 8) : lane == 16 ? a1 + 8.0 : lane == 17 ? a2 + 8.0 : lane == 18 ? sch : 1.0;
    // (float logarithm / reciprocal of the double argument: 1e-7 relative on terms of size <= 300 -- the factorisation's rounding is 1e3 times that)
    const double X = bsm_log((float)arg), Rc = (double)(1.0f / (float)arg);
    const double logpiv = lane < dl1 ? bsm_log(1.0f + dm1) : 0.0;  // (1e-7 absolute per pivot: the factorisation's own rounding is 1e3 times that)
    const double cg = lane < 8 ? -1.0 : lane < 16 ? 1.0 : lane == 16 ? a1 + 7.5 : lane == 17 ? -(a2 + 7.5) : 0.0;
    const double s1 = wave_sum_d(cg * X - 0.5 * logpiv);  // lgamma(a1) - lgamma(a2) - 1/2 logdet M_pa without the series and the linear term
    const double s2 = wave_sum_d(lane < 8 ? -Rc : lane < 16 ? Rc : lane == 16 ? X : lane == 17 ? -X : 0.0);
    double lj = 0.0, gprime = 0.0, ls = 0.0;
    if (Nn > 0.0) {  // linearGaussian.py:118: a node without observations scores 0
      ls = __shfl(X, 18, 64);
      const double lg_ld = s1 + bsm_ser_lgamma(a1 + 8.0) - bsm_ser_lgamma(a2 + 8.0) - (a1 - a2);
      gprime = 0.5 * (s2 + bsm_ser_digamma(a1 + 8.0) - bsm_ser_digamma(a2 + 8.0)) + bp.log_t;
      lj = cj[j] + lg_ld + 0.5 * (al - d + 2.0 * l + 1.0) * bp.log_t - c2 * ls;
    }
    lp_wave += lj;
    if (act) {
      const float* Rrow = R + v * d;
      float t_r = -pyv[v] - R[j * d + v];  // (R - I) (p o y) - R_j
      if ((d & 1) == 0) {  // (rows of R are 8-byte aligned)
        float t_2 = 0.f;
#pragma unroll 4
        for (int bb = 0; bb < d; bb += 2) {
          const float2 r2 = *reinterpret_cast<const float2*>(Rrow + bb), y2 = *reinterpret_cast<const float2*>(pyv + bb);
          t_r = fmaf(r2.x, y2.x, t_r);
          t_2 = fmaf(r2.y, y2.y, t_2);
        }
        t_r += t_2;
      } else {
#pragma unroll 8
        for (int bb = 0; bb < d; ++bb) t_r = fmaf(Rrow[bb], pyv[bb], t_r);
      }
      // 1 - (M_pa^-1)_vv = (L_vv^2 - 1) / L_vv^2 - |off-diagonal part of column v of L^-1|^2, each term O(p_v^2)
      const float h_r = p > 0.f ? (dm1 / (1.0f + dm1) - offd) / p : 0.f;
      const double dl = Nn > 0.0 ? gprime - 0.5 * ls - (double)h_r - (2.0 * c2 / sch) * (double)y * (double)t_r : 0.0;
      out[v * d + j] = v == j ? 0.f : (float)dl * tau * alpha * gs * (1.0f - gs);
    }
    wave_lds_fence();
  }
  if (lane == 0) red[wave] = lp_wave;
  __syncthreads();
  if (tid == 0) logprobs[(size_t)m * S + s] = (float)(red[0] + red[1] + red[2] + red[3]);
}
