// acyclicity-constraint gradient, 33 <= d <= 64: matrix powers on the f16 MFMA with two-way split, block-scaled operands (gfx950)
#pragma once
#include "common.h"
#include "kernels_acyc.h"
#include "kernels_acyc_bf16.h"

// ------------------------------------------------------------------------------------------------
// K5c  same computation as k_acyc / k_acyc_bf (reference: graph_utils.py:8-28, dibs.py:121-140, 557-601), half the matrix
//      instructions of k_acyc_bf.  A float x is carried as TWO f16 pieces of a block-scaled value, x 2^e = h + m (h = rne_f16(x 2^e),
//      m = rne_f16(x 2^e - h): 22+ mantissa bits), and a product is evaluated as
//          Ah Bh + (Ah Bm + Am Bh)                                   (the dropped Am Bm is <= 2^-24 relative, of either sign)
//      with v_mfma_f32_16x16x32_f16 accumulating in float: 3 instructions of 16 cycles per 16 x 16 x 32 block (k_acyc_bf: 6, the
//      f32 MFMA: 8 of 32 cycles).  Products of f16 values are exact in float.  Emulated against double (numpy, tests/tools/
//      acyc_split_emulation.py): 0.8 - 2e-6 of max |M^(d-1)| over alpha = 0 .. 1000, the same as a float matmul (0.8 - 5e-6) and
//      the three-piece bf16 scheme (0.6 - 1.3e-6).
//
// The f16 range (6e-5 .. 65504 normal) needs a scale per power.  All entries of a power of M = I + G~/d are >= 0 and P >= I, and
//      max (P P) <= 64 max(P)^2,    max (M P) <= rowsum(M) max(P) < 2 max(P),
// so the shift of the NEXT power is chosen from the exact maximum of the CURRENT one -- which every wave knows one barrier late for
// free: a wave's row maximum travels through LDS with the image it stores (8 v_max3 + 6 DPP steps per product).  The stored pieces
// always stay below 2^15; the bound is loose by at most 2^6, i.e. entries down to 2^-12 of the largest keep full two-piece
// precision and smaller ones an absolute error of 2^-34 of the largest (irrelevant: every entry is added to O(1) neighbours by
// the back-projection).  The scales are powers of two (exact) and are undone once, in the epilogue factor.
//
// Layout, lane mapping, swapped-operand MFMA, transposing reads and the chunk rotation of the image are those of k_acyc_bf
// (kernels_acyc_bf16.h); an image has two pieces (16 KiB), two images + the float staging rows of the epilogue 34 KiB per block:
// four blocks per CU at 128 registers.
// Measured and dropped: skipping the seven products of a chain whose soft graph is saturated everywhere (g (1 - g) = 0 on all edges, the
// contribution is exactly zero) -- a flag per chain from the draw loop costs 2-3 us per launch and never fires: at t = 100 ... 1 000 of the
// headline trajectory every chain keeps live edges (74.9 us per launch throughout, scripts/gpu_steady_bench.py).
// grid = (ceil(Sa / 2 / cpb), Mloc rounded up to 8; re-indexed XCD-aware inside), block = 256, dynamic LDS = AHF_LDS_BYTES
// ------------------------------------------------------------------------------------------------
typedef _Float16 ahf_f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 ahf_f16x2 __attribute__((ext_vector_type(2)));

constexpr int AHF_NT = 4, AHF_TILE_BYTES = 64 * 32, AHF_PIECE_BYTES = AHF_NT * AHF_TILE_BYTES, AHF_IMG_BYTES = 2 * AHF_PIECE_BYTES,
              AHF_LDT = 68, AHF_IMG_STRIDE = 64 * AHF_LDT * 4 /* the float staging area of the epilogue overlays one image */,
              AHF_LDS_BYTES = 2 * AHF_IMG_STRIDE + 64;
constexpr int AHF_E0 = 14;  // M = I + G~/d is stored as M 2^14 (largest entry exactly 1)

// (x0, x1) s -> packed f16 pairs h, m with x s = h + m up to 2^-23 relative (s: a power of two, wave-uniform).  Four instructions per
// pair: v_pk_mul_f32 + v_cvt_pk_f16_f32 for h, one v_fma_mix{lo,hi}_f16 per value for m = rne_f16(x s - h) (the f16 operand is converted
// inside the fma, x s - h is exact in float; the plain form -- two conversions back, a packed subtraction, a packed conversion -- takes six)
__device__ __forceinline__ void ahf_split(float x0, float x1, float s, uint32_t& h, uint32_t& m) {
  const abf_f32x2 xs = abf_f32x2{x0, x1} * s;
  const ahf_f16x2 hh = __builtin_convertvector(xs, ahf_f16x2);  // v_cvt_pk_f16_f32 (round to nearest even)
  h = __builtin_bit_cast(uint32_t, hh);
  asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(m) : "v"(x0), "s"(s), "v"(h));
  asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(m) : "v"(x1), "s"(s), "v"(h));
}

struct AhfFrag {
  abf_u32x4 a[2][2];  // [k-step][piece]
};
__device__ __forceinline__ void ahf_make_frag(const f32x4 (&v)[AHF_NT], float s, AhfFrag& f) {
#pragma unroll
  for (int tj = 0; tj < AHF_NT; ++tj) {
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
}
__device__ __forceinline__ void ahf_store_image(unsigned char* img, int wr_off, const AhfFrag& f) {
#pragma unroll
  for (int tj = 0; tj < AHF_NT; ++tj)
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const abf_u32x4 q = f.a[tj >> 1][p];
      const abf_u32x2 w = (tj & 1) ? abf_u32x2{q.z, q.w} : abf_u32x2{q.x, q.y};
      *reinterpret_cast<abf_u32x2*>(img + wr_off + p * AHF_PIECE_BYTES + tj * AHF_TILE_BYTES) = w;
    }
}
__device__ __forceinline__ ahf_f16x8 ahf_tr_pair(const unsigned char* p) {
  typedef __attribute__((address_space(3))) abf_s16x4 lds_s16x4;
  const abf_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p));
  const abf_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p + 16 * 32));
  return __builtin_bit_cast(ahf_f16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
}
// k-step ks of the tile pair (tp, tp + 1): 4 fragment reads and 3 MFMAs per tile, small terms first, the two tiles alternating so
// that consecutive MFMAs never share an accumulator
template <int NU>
__device__ __forceinline__ void ahf_group(f32x4 (&acc)[AHF_NT], const AhfFrag& A, const unsigned char* img, int rd_off, int ks, int tp) {
  ahf_f16x8 b[NU][2];
#pragma unroll
  for (int u = 0; u < NU; ++u)
#pragma unroll
    for (int p = 0; p < 2; ++p) b[u][p] = ahf_tr_pair(img + rd_off + p * AHF_PIECE_BYTES + (tp + u) * AHF_TILE_BYTES + ks * 32 * 32);
  const ahf_f16x8 ah = __builtin_bit_cast(ahf_f16x8, A.a[ks][0]), am = __builtin_bit_cast(ahf_f16x8, A.a[ks][1]);
#pragma unroll
  for (int u = 0; u < NU; ++u) {
    if (ks == 0) acc[tp + u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b[u][1], ah, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
    else acc[tp + u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b[u][1], ah, acc[tp + u], 0, 0, 0);
  }
#pragma unroll
  for (int u = 0; u < NU; ++u) acc[tp + u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b[u][0], am, acc[tp + u], 0, 0, 0);
#pragma unroll
  for (int u = 0; u < NU; ++u) acc[tp + u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b[u][0], ah, acc[tp + u], 0, 0, 0);
}
// AHF_ORDER (experiment switch, default 0): 0 = per k-step small terms then the large one, all in one accumulator; 1 = the small terms of BOTH
// k-steps first, then the large ones (fragments of the large term read again); 2 = small terms in an accumulator of their own, added at the end
#ifndef AHF_ORDER
#define AHF_ORDER 0
#endif
template <int NU, bool SMALL, bool BIG>
__device__ __forceinline__ void ahf_group_part(f32x4 (&acc)[AHF_NT], const AhfFrag& A, const unsigned char* img, int rd_off, int ks, int tp, bool first) {
  ahf_f16x8 b[NU][2];
#pragma unroll
  for (int u = 0; u < NU; ++u) {
    b[u][0] = ahf_tr_pair(img + rd_off + (tp + u) * AHF_TILE_BYTES + ks * 32 * 32);
    if (SMALL) b[u][1] = ahf_tr_pair(img + rd_off + AHF_PIECE_BYTES + (tp + u) * AHF_TILE_BYTES + ks * 32 * 32);
  }
  const ahf_f16x8 ah = __builtin_bit_cast(ahf_f16x8, A.a[ks][0]), am = __builtin_bit_cast(ahf_f16x8, A.a[ks][1]);
  if (SMALL) {
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      if (first) acc[tp + u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b[u][1], ah, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
      else acc[tp + u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b[u][1], ah, acc[tp + u], 0, 0, 0);
    }
#pragma unroll
    for (int u = 0; u < NU; ++u) acc[tp + u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b[u][0], am, acc[tp + u], 0, 0, 0);
  }
  if (BIG) {
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      if (first && !SMALL) acc[tp + u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b[u][0], ah, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
      else acc[tp + u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b[u][0], ah, acc[tp + u], 0, 0, 0);
    }
  }
}
template <bool FOUR>
__device__ __forceinline__ void ahf_matmul(f32x4 (&acc)[AHF_NT], const AhfFrag& A, const unsigned char* img, int rd_off) {
#if AHF_ORDER == 0
  ahf_group<2>(acc, A, img, rd_off, 0, 0);
  if constexpr (FOUR) {
    ahf_group<2>(acc, A, img, rd_off, 0, 2);
    ahf_group<2>(acc, A, img, rd_off, 1, 0);
    ahf_group<2>(acc, A, img, rd_off, 1, 2);
  } else {
    ahf_group<1>(acc, A, img, rd_off, 0, 2);
    ahf_group<2>(acc, A, img, rd_off, 1, 0);
    ahf_group<1>(acc, A, img, rd_off, 1, 2);
    acc[3] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
#elif AHF_ORDER == 1
  constexpr int N2 = FOUR ? 2 : 1;
  ahf_group_part<2, true, false>(acc, A, img, rd_off, 0, 0, true);
  ahf_group_part<N2, true, false>(acc, A, img, rd_off, 0, 2, true);
  ahf_group_part<2, true, false>(acc, A, img, rd_off, 1, 0, false);
  ahf_group_part<N2, true, false>(acc, A, img, rd_off, 1, 2, false);
  ahf_group_part<2, false, true>(acc, A, img, rd_off, 0, 0, false);
  ahf_group_part<N2, false, true>(acc, A, img, rd_off, 0, 2, false);
  ahf_group_part<2, false, true>(acc, A, img, rd_off, 1, 0, false);
  ahf_group_part<N2, false, true>(acc, A, img, rd_off, 1, 2, false);
  if (!FOUR) acc[3] = f32x4{0.f, 0.f, 0.f, 0.f};
#else
  constexpr int N2 = FOUR ? 2 : 1;
  f32x4 sm_[AHF_NT];
  ahf_group_part<2, true, false>(sm_, A, img, rd_off, 0, 0, true);
  ahf_group_part<2, false, true>(acc, A, img, rd_off, 0, 0, true);
  ahf_group_part<N2, true, false>(sm_, A, img, rd_off, 0, 2, true);
  ahf_group_part<N2, false, true>(acc, A, img, rd_off, 0, 2, true);
  ahf_group_part<2, true, false>(sm_, A, img, rd_off, 1, 0, false);
  ahf_group_part<2, false, true>(acc, A, img, rd_off, 1, 0, false);
  ahf_group_part<N2, true, false>(sm_, A, img, rd_off, 1, 2, false);
  ahf_group_part<N2, false, true>(acc, A, img, rd_off, 1, 2, false);
#pragma unroll
  for (int tj = 0; tj < (FOUR ? 4 : 3); ++tj) acc[tj] += sm_[tj];
  if (!FOUR) acc[3] = f32x4{0.f, 0.f, 0.f, 0.f};
#endif
}

// largest of the wave's 16 x 64 accumulator values (all >= 0: bit patterns compare as integers), in every lane's SGPR copy
template <int NT>
__device__ __forceinline__ uint32_t ahf_wave_max(const f32x4 (&acc)[NT]) {
  int v = 0;  // (integer maxima: fmaxf would canonicalise every input -- 24 extra instructions per product)
#pragma unroll
  for (int tj = 0; tj < NT; ++tj) {
    const int x0 = (int)__float_as_uint(acc[tj][0]), x1 = (int)__float_as_uint(acc[tj][1]), x2 = (int)__float_as_uint(acc[tj][2]),
              x3 = (int)__float_as_uint(acc[tj][3]);
    const int t = x0 > x1 ? x0 : x1;
    v = t > v ? t : v;
    const int u = x2 > x3 ? x2 : x3;
    v = u > v ? u : v;
  }
#define AHF_DPP_MAX(ctrl, rmask)                                                     \
  {                                                                                  \
    const int t = __builtin_amdgcn_update_dpp(0, v, ctrl, rmask, 0xf, false);        \
    v = t > v ? t : v;                                                               \
  }
  AHF_DPP_MAX(0xB1, 0xf)   // quad_perm [1,0,3,2]
  AHF_DPP_MAX(0x4E, 0xf)   // quad_perm [2,3,0,1]
  AHF_DPP_MAX(0x141, 0xf)  // row_half_mirror
  AHF_DPP_MAX(0x140, 0xf)  // row_mirror: every lane of a row of 16 holds the row's maximum
  AHF_DPP_MAX(0x142, 0xa)  // row_bcast:15 into rows 1, 3
  AHF_DPP_MAX(0x143, 0xc)  // row_bcast:31 into rows 2, 3: lane 63 holds the wave's maximum
#undef AHF_DPP_MAX
  return (uint32_t)__builtin_amdgcn_readlane(v, 63);
}
__device__ __forceinline__ float ahf_pow2(int e) { return __uint_as_float((uint32_t)(e + 127) << 23); }

// FOUR: d > 48 (all four row / column tiles in use); the d <= 48 instantiation skips the fourth wave's products and the fourth column tile
template <bool FOUR, int WPE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void k_acyc_hf(const float* __restrict__ scores, const float* __restrict__ eas, float* __restrict__ part, Key2 carry, int m0,
                                                 int M_global, int Mloc, int d, int Sa, int cpb, float alpha, float tau, int layout,
                                                 int tiny, int n_acyc_blk) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  unsigned char* const sb = reinterpret_cast<unsigned char*>(smem);
  uint32_t* const slots = reinterpret_cast<uint32_t*>(sb + 2 * AHF_IMG_STRIDE);  // [2][4] row maxima of the waves, ping-pong
  // XCD-aware block order: all blocks of a particle on one XCD (see k_acyc_bf)
  const int L = blockIdx.x + gridDim.x * blockIdx.y, p_lo = L & 7, tq = L >> 3;
  const int bx = tq % (int)gridDim.x, m = (tq / (int)gridDim.x) * 8 + p_lo;
  if (m >= Mloc) return;  // (block-uniform)
  const int blk = bx, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g4 = lane >> 4, r = lane & 15;
  const int a = 16 * wave + r, b0 = 4 * g4;  // row; first column within a tile
  const Key2 km = rng_split_row_uniform(carry, (uint32_t)M_global + 1u, (uint32_t)(m0 + m) + 1u, layout);  // dibs.py:595: key used directly
  const uint64_t dd = (uint64_t)d * d, nbits = (uint64_t)Sa * dd;
  const float inv_d = 1.0f / (float)d;
  const bool fast = tau == 1.0f && eas != nullptr;  // sigmoid(eps + a) with eps = log(u / (1 - u))  ==  u / (u + (1 - u) exp(-a))
  const float* sm = (fast ? eas : scores) + (size_t)m * dd;  // fast: exp(-alpha s) from k_edge_scores (the same for every chain of the particle)
  const float ulo = tiny ? 1.17549435e-38f : 1.1920929e-07f;
  const int n_units = Sa >> 1;  // host: legacy PRNG layout, Sa even, Sa d d < 2^32 (chains sa and sa + Sa/2 share their Threefry calls)
  const TfKeys tk = tf_keys(km);
  const bool row_active = FOUR || 16 * wave < d;
  const int wr_off = a * 32 + ((g4 + (r >> 2)) & 3) * 8;
  const int rd_off = (4 * g4 + (r >> 2)) * 32 + (((r & 3) + g4) & 3) * 8;
  float* const po = part + ((size_t)m * n_acyc_blk + blk) * dd + (size_t)a * d;
  f32x4 g[AHF_NT], gnext[AHF_NT], out[AHF_NT];
#pragma unroll
  for (int tj = 0; tj < AHF_NT; ++tj) {
    gnext[tj] = f32x4{0.f, 0.f, 0.f, 0.f};
    out[tj] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const float s0 = ahf_pow2(AHF_E0);
  int par = 0;  // slot set the next product publishes its maxima in

  for (int c = 0; c < cpb; ++c) {
    const int unit = blk * cpb + c;
    if (unit >= n_units) break;
    // Soft graphs of BOTH chains of the pair, drawn in ELEMENT order -- thread tid takes elements e = tid, tid + 256, ... of the d x d
    // matrix: every lane busy (10 wave-draws per wave at d = 50 where the owner-lane order needs 14, the fourth wave's and the fourth
    // column tile's mostly for idle lanes), the Threefry counter is base + e, the score loads are coalesced and requested together --
    // and handed to the owning lanes through LDS: G0 / G1 [64][AHF_LDT] float over the (free) images.
    {
      const int sa = unit;
      float* const G0 = reinterpret_cast<float*>(sb);
      float* const G1 = reinterpret_cast<float*>(sb + AHF_IMG_STRIDE);
      const int ndd = (int)dd, ndraw = (ndd + 255) >> 8;
      // (a, b) of element tid, advanced by 256 = qa d + qb per draw
      const int qa = 256 / d, qb = 256 - qa * d;
      int ea_ = tid / d, eb_ = tid - ea_ * d;
      float s_next = tid < ndd ? sm[tid] : 0.f;  // (the next draw's score is requested one draw ahead)
      const uint32_t cbase = (uint32_t)((uint64_t)sa * dd), chalf = (uint32_t)(nbits >> 1);
      for (int k = 0; k < ndraw; ++k) {
        const int e = tid + 256 * k;
        const float s_cur = s_next;
        s_next = e + 256 < ndd ? sm[e + 256] : 0.f;
        if (e < ndd) {
          float gv0 = 0.f, gv1 = 0.f;
          if (ea_ != eb_) {
            const float ea = fast ? s_cur : alpha * s_cur;
            uint32_t y0, y1;
            const uint32_t c0 = cbase + (uint32_t)e;
            threefry2x32_uk(tk, c0, c0 + chalf, y0, y1);
            if (fast) {
              const float u0 = rng_uniform(y0, ulo, 1.0f), u1 = rng_uniform(y1, ulo, 1.0f);
              // (saturated edges must give exactly 1 as the reference's sigmoid does: see k_acyc_bf)
              const float den0 = fmaf(1.0f - u0, ea, u0), den1 = fmaf(1.0f - u1, ea, u1);
              gv0 = den0 == u0 ? 1.0f : u0 * __builtin_amdgcn_rcpf(den0);
              gv1 = den1 == u1 ? 1.0f : u1 * __builtin_amdgcn_rcpf(den1);
            } else {
              gv0 = 1.0f / (1.0f + expf(-tau * (rng_logistic(y0, tiny) + ea)));
              gv1 = 1.0f / (1.0f + expf(-tau * (rng_logistic(y1, tiny) + ea)));
            }
          }
          G0[ea_ * AHF_LDT + eb_] = gv0;
          G1[ea_ * AHF_LDT + eb_] = gv1;
        }
        ea_ += qa;
        eb_ += qb;
        if (eb_ >= d) {
          eb_ -= d;
          ++ea_;
        }
      }
      __syncthreads();
#pragma unroll
      for (int tj = 0; tj < AHF_NT; ++tj) {
        f32x4 v0 = f32x4{0.f, 0.f, 0.f, 0.f}, v1 = v0;
        if (a < d && (FOUR || tj < 3)) {
          v0 = *reinterpret_cast<const f32x4*>(G0 + a * AHF_LDT + 16 * tj + b0);
          v1 = *reinterpret_cast<const f32x4*>(G1 + a * AHF_LDT + 16 * tj + b0);
          if (16 * tj + 16 > d) {  // (wave-uniform: the tile that holds column d - 1; columns beyond hold stale LDS)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const bool in = b0 + 16 * tj + i < d;
              v0[i] = in ? v0[i] : 0.f;
              v1[i] = in ? v1[i] : 0.f;
            }
          }
        }
        g[tj] = v0;
        gnext[tj] = v1;
      }
      __syncthreads();  // the images are written next
    }
    for (int hf = 0; hf < 2; ++hf) {
      if (hf == 1) {
#pragma unroll
        for (int tj = 0; tj < AHF_NT; ++tj) g[tj] = gnext[tj];
      }
      AhfFrag A;
      f32x4 acc[AHF_NT];
      abf_m0(g, acc, a, b0, d, inv_d);
      ahf_make_frag(acc, s0, A);
      AhfFrag AM;               // M's own fragments for the "times M" steps (kept when the register budget allows: WPE == 3)
      if constexpr (WPE <= 3) AM = A;
      int cur = 0;              // image holding the running power
      int E = AHF_E0;           // ... as P 2^E
      int exP = AHF_E0;         // exponent of its largest stored entry (the unit diagonal of M: exactly 2^14)
      ahf_store_image(sb, wr_off, A);
      if (!FOUR && !row_active) ahf_store_image(sb + AHF_IMG_STRIDE, wr_off, A);  // (all zero; the previous chain's float staging may have covered these rows)
      __syncthreads();
      // left-to-right binary powering of e = d - 1 (>= 32 here)
      const int ex = d - 1;
      const int hb = 31 - __builtin_clz((unsigned)ex);
      int Eacc = 0;
      for (int bit = hb - 1; bit >= 0; --bit) {
        const bool mult = (ex >> bit) & 1, last_sq = bit == 0 && !mult;
        if (row_active) ahf_matmul<FOUR>(acc, A, sb + cur, rd_off);  // P <- P P : acc = P^2 2^(2E) <= 64 max^2 < 2^(2 exP + 8)
        cur ^= AHF_IMG_STRIDE;
        Eacc = 2 * E;
        if (!last_sq) {
          {
            const int sh = 7 - 2 * exP;  // stored pieces < 2^15
            const uint32_t wm = ahf_wave_max<AHF_NT>(acc);
            if (lane == 0) slots[par * 4 + wave] = wm;
            if (row_active) {
              ahf_make_frag(acc, ahf_pow2(sh), A);
              ahf_store_image(sb + cur, wr_off, A);
            }
            __syncthreads();
            const abf_u32x4 q = *reinterpret_cast<const abf_u32x4*>(slots + par * 4);
            uint32_t mx = q.x > q.y ? q.x : q.y;
            mx = mx > q.z ? mx : q.z;
            mx = mx > q.w ? mx : q.w;
            mx = __builtin_amdgcn_readfirstlane(mx);
            E = Eacc + sh;
            exP = (int)(mx >> 23) - 127 + sh;
            par ^= 1;
          }
          if (mult) {  // P <- M P : acc = (M 2^14)(P 2^E) < 2^14 2 max < 2^(exP + 16)
            if constexpr (WPE <= 3) {
              if (row_active) ahf_matmul<FOUR>(acc, AM, sb + cur, rd_off);
            } else {
              f32x4 mv[AHF_NT];
              AhfFrag A0;
              abf_m0(g, mv, a, b0, d, inv_d);
              ahf_make_frag(mv, s0, A0);
              if (row_active) ahf_matmul<FOUR>(acc, A0, sb + cur, rd_off);
            }
            cur ^= AHF_IMG_STRIDE;
            Eacc = AHF_E0 + E;
            if (bit != 0) {
              const int sh = -1 - exP;
              const uint32_t wm = ahf_wave_max<AHF_NT>(acc);
              if (lane == 0) slots[par * 4 + wave] = wm;
              if (row_active) {
                ahf_make_frag(acc, ahf_pow2(sh), A);
                ahf_store_image(sb + cur, wr_off, A);
              }
              __syncthreads();
              const abf_u32x4 q = *reinterpret_cast<const abf_u32x4*>(slots + par * 4);
              uint32_t mx = q.x > q.y ? q.x : q.y;
              mx = mx > q.z ? mx : q.z;
              mx = mx > q.w ? mx : q.w;
              mx = __builtin_amdgcn_readfirstlane(mx);
              E = Eacc + sh;
              exP = (int)(mx >> 23) - 127 + sh;
              par ^= 1;
            }
          }
        }
      }
      // acc = rows of M^{d-1} 2^Eacc; through the free image as float [row][AHF_LDT] to read it transposed
      float* T = reinterpret_cast<float*>(sb + cur);
#pragma unroll
      for (int tj = 0; tj < AHF_NT; ++tj) *reinterpret_cast<f32x4*>(T + a * AHF_LDT + 16 * tj + b0) = acc[tj];
      __syncthreads();
      const float ta = ldexpf(tau * alpha, -Eacc);
#pragma unroll
      for (int tj = 0; tj < AHF_NT; ++tj)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int b = b0 + 16 * tj + i;
          const float gv = g[tj][i];  // 0 outside the matrix and on the diagonal
          const float v = T[b * AHF_LDT + a] * (ta * gv * (1.0f - gv));
          out[tj][i] += v;
        }
      __syncthreads();  // T is the image the next chain's first product writes
    }
  }
  if (a < d) {
#pragma unroll
    for (int tj = 0; tj < AHF_NT; ++tj)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int b = b0 + 16 * tj + i;
        if (b < d) po[b] = out[tj][i];
      }
  }
}


// ------------------------------------------------------------------------------------------------
// K5c'  the same scheme for 65 <= d <= 112: NT = ceil(d / 16) in {5, 6, 7} tiles and NT waves per block (one block = one pair of chains).
//       Differences to k_acyc_hf: a k-step covers the tile pair (2 ks, 2 ks + 1) of the left operand and the image of a power has
//       32 ceil(NT / 2) rows per column tile -- for odd NT the last 16 are zero rows (written once), so every fragment read is a plain one
//       (k_acyc_bfw needs a zero page there); the squaring bound is max (P P) <= 128 max(P)^2.  Images of 2 x NT x 32 ceil(NT/2) x 32 bytes:
//       30 / 37 / 56 KiB each (k_acyc_bfw: 38 / 55 / 75), i.e. two blocks per CU up to NT = 6.  Replaces k_acyc_bfw (BASELINE config 5 is
//       d = 100); DIBS_ACYC_BF16=1 keeps that one for A/B runs.
// grid = (ceil(Sa / 2 / cpb), Mloc rounded up to 8; re-indexed XCD-aware inside), block = 64 NT, dynamic LDS = ahfw_lds_bytes(NT)
// ------------------------------------------------------------------------------------------------
template <int NT>
struct Ahfw {
  static constexpr int NKS = (NT + 1) / 2, KROWS = 32 * NKS, TILE_BYTES = KROWS * 32, PIECE_BYTES = NT * TILE_BYTES, IMG_BYTES = 2 * PIECE_BYTES,
                       LDT = 16 * NT + 4, T_BYTES = 16 * NT * LDT * 4, IMG_STRIDE = IMG_BYTES > T_BYTES ? IMG_BYTES : T_BYTES, NTHR = 64 * NT;
};
__host__ __device__ inline size_t ahfw_lds_bytes(int nt) {
  const size_t nks = (size_t)(nt + 1) / 2, img = 2 * (size_t)nt * 32 * nks * 32, t = (size_t)16 * nt * (16 * nt + 4) * 4;
  return 2 * (img > t ? img : t) + 128;
}
template <int NT>
struct AhfwFrag {
  abf_u32x4 a[(NT + 1) / 2][2];
};
template <int NT>
__device__ __forceinline__ void ahfw_make_frag(const f32x4 (&v)[NT], float s, AhfwFrag<NT>& f) {
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
  if (NT & 1) {  // (the partner tile of the last one does not exist)
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      f.a[NT >> 1][p].z = 0u;
      f.a[NT >> 1][p].w = 0u;
    }
  }
}
template <int NT>
__device__ __forceinline__ void ahfw_store_image(unsigned char* img, int wr_off, const AhfwFrag<NT>& f) {
#pragma unroll
  for (int tj = 0; tj < NT; ++tj)
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const abf_u32x4 q = f.a[tj >> 1][p];
      const abf_u32x2 w = (tj & 1) ? abf_u32x2{q.z, q.w} : abf_u32x2{q.x, q.y};
      *reinterpret_cast<abf_u32x2*>(img + wr_off + p * Ahfw<NT>::PIECE_BYTES + tj * Ahfw<NT>::TILE_BYTES) = w;
    }
}
template <int NT, int KS, int NU>
__device__ __forceinline__ void ahfw_tile_step(f32x4 (&acc)[NT], const AhfwFrag<NT>& A, const unsigned char* img, int rd_off, int tj) {
  ahf_f16x8 b[NU][2];
#pragma unroll
  for (int u = 0; u < NU; ++u)
#pragma unroll
    for (int p = 0; p < 2; ++p) b[u][p] = ahf_tr_pair(img + rd_off + p * Ahfw<NT>::PIECE_BYTES + (tj + u) * Ahfw<NT>::TILE_BYTES + KS * 32 * 32);
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
struct AhfwSteps {
  static __device__ __forceinline__ void run(f32x4 (&acc)[NT], const AhfwFrag<NT>& A, const unsigned char* img, int rd_off) {
#pragma unroll
    for (int tj = 0; tj + 1 < NT; tj += 2) ahfw_tile_step<NT, KS, 2>(acc, A, img, rd_off, tj);
    if (NT & 1) ahfw_tile_step<NT, KS, 1>(acc, A, img, rd_off, NT - 1);
    AhfwSteps<NT, KS + 1>::run(acc, A, img, rd_off);
  }
};
template <int NT>
struct AhfwSteps<NT, (NT + 1) / 2> {
  static __device__ __forceinline__ void run(f32x4 (&)[NT], const AhfwFrag<NT>&, const unsigned char*, int) {}
};

template <int NT>
__global__ __launch_bounds__(64 * NT) void k_acyc_hfw(const float* __restrict__ scores, const float* __restrict__ eas, float* __restrict__ part, Key2 carry,
                                                      int m0, int M_global, int Mloc, int d, int Sa, int cpb, float alpha, float tau, int layout, int tiny,
                                                      int n_acyc_blk) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  typedef Ahfw<NT> G;
  unsigned char* const sb = reinterpret_cast<unsigned char*>(smem);
  uint32_t* const slots = reinterpret_cast<uint32_t*>(sb + 2 * G::IMG_STRIDE);  // [2][8] row maxima of the waves, ping-pong
  const int L = blockIdx.x + gridDim.x * blockIdx.y, p_lo = L & 7, tq = L >> 3;
  const int bx = tq % (int)gridDim.x, m = (tq / (int)gridDim.x) * 8 + p_lo;
  if (m >= Mloc) return;  // (block-uniform)
  const int blk = bx, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g4 = lane >> 4, r = lane & 15;
  const int a = 16 * wave + r, b0 = 4 * g4;
  const Key2 km = rng_split_row_uniform(carry, (uint32_t)M_global + 1u, (uint32_t)(m0 + m) + 1u, layout);  // dibs.py:595: key used directly
  const uint64_t dd = (uint64_t)d * d, nbits = (uint64_t)Sa * dd;
  const float inv_d = 1.0f / (float)d;
  const bool fast = tau == 1.0f && eas != nullptr;
  const float* sm = (fast ? eas : scores) + (size_t)m * dd;
  const float ulo = tiny ? 1.17549435e-38f : 1.1920929e-07f;
  const int n_units = Sa >> 1;
  const TfKeys tk = tf_keys(km);
  const int wr_off = a * 32 + ((g4 + (r >> 2)) & 3) * 8;
  const int rd_off = (4 * g4 + (r >> 2)) * 32 + (((r & 3) + g4) & 3) * 8;
  float* const po = part + ((size_t)m * n_acyc_blk + blk) * dd + (size_t)a * d;
  f32x4 g[NT], gnext[NT], out[NT];
#pragma unroll
  for (int tj = 0; tj < NT; ++tj) {
    gnext[tj] = f32x4{0.f, 0.f, 0.f, 0.f};
    out[tj] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const float s0 = ahf_pow2(AHF_E0);
  int par = 0;

  for (int c = 0; c < cpb; ++c) {
    const int unit = blk * cpb + c;
    if (unit >= n_units) break;
    // soft graphs of both chains of the pair in element order, handed to the owning lanes through LDS (see k_acyc_hf)
    {
      const int sa = unit;
      float* const G0 = reinterpret_cast<float*>(sb);
      float* const G1 = reinterpret_cast<float*>(sb + G::IMG_STRIDE);
      const int ndd = (int)dd, ndraw = (ndd + G::NTHR - 1) / G::NTHR;
      const int qa = G::NTHR / d, qb = G::NTHR - qa * d;
      int ea_ = tid / d, eb_ = tid - ea_ * d;
      float s_next = tid < ndd ? sm[tid] : 0.f;
      const uint32_t cbase = (uint32_t)((uint64_t)sa * dd), chalf = (uint32_t)(nbits >> 1);
      for (int k = 0; k < ndraw; ++k) {
        const int e = tid + G::NTHR * k;
        const float s_cur = s_next;
        s_next = e + G::NTHR < ndd ? sm[e + G::NTHR] : 0.f;
        if (e < ndd) {
          float gv0 = 0.f, gv1 = 0.f;
          if (ea_ != eb_) {
            const float ea = fast ? s_cur : alpha * s_cur;
            uint32_t y0, y1;
            const uint32_t c0 = cbase + (uint32_t)e;
            threefry2x32_uk(tk, c0, c0 + chalf, y0, y1);
            if (fast) {
              const float u0 = rng_uniform(y0, ulo, 1.0f), u1 = rng_uniform(y1, ulo, 1.0f);
              const float den0 = fmaf(1.0f - u0, ea, u0), den1 = fmaf(1.0f - u1, ea, u1);
              gv0 = den0 == u0 ? 1.0f : u0 * __builtin_amdgcn_rcpf(den0);   // (saturated edges give exactly 1: see k_acyc_bf)
              gv1 = den1 == u1 ? 1.0f : u1 * __builtin_amdgcn_rcpf(den1);
            } else {
              gv0 = 1.0f / (1.0f + expf(-tau * (rng_logistic(y0, tiny) + ea)));
              gv1 = 1.0f / (1.0f + expf(-tau * (rng_logistic(y1, tiny) + ea)));
            }
          }
          G0[ea_ * G::LDT + eb_] = gv0;
          G1[ea_ * G::LDT + eb_] = gv1;
        }
        ea_ += qa;
        eb_ += qb;
        if (eb_ >= d) {
          eb_ -= d;
          ++ea_;
        }
      }
      __syncthreads();
#pragma unroll
      for (int tj = 0; tj < NT; ++tj) {
        f32x4 v0 = f32x4{0.f, 0.f, 0.f, 0.f}, v1 = v0;
        if (a < d && 16 * tj < d) {
          v0 = *reinterpret_cast<const f32x4*>(G0 + a * G::LDT + 16 * tj + b0);
          v1 = *reinterpret_cast<const f32x4*>(G1 + a * G::LDT + 16 * tj + b0);
          if (16 * tj + 16 > d) {  // (wave-uniform: the tile that holds column d - 1; columns beyond hold stale LDS)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const bool in = b0 + 16 * tj + i < d;
              v0[i] = in ? v0[i] : 0.f;
              v1[i] = in ? v1[i] : 0.f;
            }
          }
        }
        g[tj] = v0;
        gnext[tj] = v1;
      }
      __syncthreads();
      // rows 16 NT .. KROWS - 1 of every column tile (odd NT: the second half of the last k-step) are zero in both images: the float
      // staging areas above / below covered them
      if (NT & 1) {
        for (int e = tid; e < 2 * 2 * NT * 64; e += G::NTHR) {  // [image][piece][tile]: 16 rows x 32 bytes = 64 x 8 bytes
          const int q8 = e & 63, t3 = e >> 6, tile = t3 % NT, pi = (t3 / NT) & 1, im = t3 / (2 * NT);
          *reinterpret_cast<abf_u32x2*>(sb + im * G::IMG_STRIDE + pi * G::PIECE_BYTES + tile * G::TILE_BYTES + 16 * NT * 32 + q8 * 8) = abf_u32x2{0u, 0u};
        }
      }
    }
    for (int hf = 0; hf < 2; ++hf) {
      if (hf == 1) {
#pragma unroll
        for (int tj = 0; tj < NT; ++tj) g[tj] = gnext[tj];
      }
      AhfwFrag<NT> A;
      f32x4 acc[NT];
      abfw_m0<NT>(g, acc, a, b0, d, inv_d);
      ahfw_make_frag<NT>(acc, s0, A);
      int cur = 0, E = AHF_E0, exP = AHF_E0, Eacc = 0;
      ahfw_store_image<NT>(sb, wr_off, A);
      __syncthreads();
      const int ex = d - 1;
      const int hb = 31 - __builtin_clz((unsigned)ex);
      for (int bit = hb - 1; bit >= 0; --bit) {
        const bool mult = (ex >> bit) & 1, last_sq = bit == 0 && !mult;
        AhfwSteps<NT, 0>::run(acc, A, sb + cur, rd_off);  // P <- P P : acc = P^2 2^(2E) <= 128 max^2 < 2^(2 exP + 9)
        cur ^= G::IMG_STRIDE;
        Eacc = 2 * E;
        if (!last_sq) {
          {
            const int sh = 6 - 2 * exP;  // stored pieces < 2^15
            const uint32_t wm = ahf_wave_max<NT>(acc);
            if (lane == 0) slots[par * 8 + wave] = wm;
            ahfw_make_frag<NT>(acc, ahf_pow2(sh), A);
            ahfw_store_image<NT>(sb + cur, wr_off, A);
            __syncthreads();
            uint32_t mx = 0u;
#pragma unroll
            for (int w8 = 0; w8 < NT; ++w8) {
              const uint32_t q = slots[par * 8 + w8];
              mx = q > mx ? q : mx;
            }
            mx = __builtin_amdgcn_readfirstlane(mx);
            E = Eacc + sh;
            exP = (int)(mx >> 23) - 127 + sh;
            par ^= 1;
          }
          if (mult) {  // P <- M P : acc = (M 2^14)(P 2^E) < 2^14 2 max < 2^(exP + 16)
            f32x4 mv[NT];
            AhfwFrag<NT> A0;
            abfw_m0<NT>(g, mv, a, b0, d, inv_d);
            ahfw_make_frag<NT>(mv, s0, A0);
            AhfwSteps<NT, 0>::run(acc, A0, sb + cur, rd_off);
            cur ^= G::IMG_STRIDE;
            Eacc = AHF_E0 + E;
            if (bit != 0) {
              const int sh = -1 - exP;
              const uint32_t wm = ahf_wave_max<NT>(acc);
              if (lane == 0) slots[par * 8 + wave] = wm;
              ahfw_make_frag<NT>(acc, ahf_pow2(sh), A);
              ahfw_store_image<NT>(sb + cur, wr_off, A);
              __syncthreads();
              uint32_t mx = 0u;
#pragma unroll
              for (int w8 = 0; w8 < NT; ++w8) {
                const uint32_t q = slots[par * 8 + w8];
                mx = q > mx ? q : mx;
              }
              mx = __builtin_amdgcn_readfirstlane(mx);
              E = Eacc + sh;
              exP = (int)(mx >> 23) - 127 + sh;
              par ^= 1;
            }
          }
        }
      }
      float* T = reinterpret_cast<float*>(sb + cur);
#pragma unroll
      for (int tj = 0; tj < NT; ++tj) *reinterpret_cast<f32x4*>(T + a * G::LDT + 16 * tj + b0) = acc[tj];
      __syncthreads();
      // (first-order compensation of the pipe's truncation bias: every product level of the 3 x 32-term f16 dot products loses ~1.15e-7
      //  relative, always downwards, which repeated squaring turns into -(d - 1) 1.15e-7 on M^(d-1) -- measured against the f32-MFMA
      //  kernel at d = 80: mean -1.0e-5 / -8.2e-6 over the entries at alpha = 1000 / 20, spread 1e-6; with the factor the difference of
      //  the two kernels is their rounding noise.  tests: test_acyclicity_f16_pipe_worst_cases)
      const float ta = ldexpf(tau * alpha * (1.0f + 1.15e-7f * (float)(d - 1)), -Eacc);
#pragma unroll
      for (int tj = 0; tj < NT; ++tj)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int b = b0 + 16 * tj + i;
          const float gv = g[tj][i];
          out[tj][i] += T[b * G::LDT + a] * (ta * gv * (1.0f - gv));
        }
      __syncthreads();
      if ((NT & 1) && hf == 0) {  // (the staging rows covered the zero rows of that image: restore them before the next chain's products)
        const int im = cur ? 1 : 0;
        for (int e = tid; e < 2 * NT * 64; e += G::NTHR) {
          const int q8 = e & 63, t3 = e >> 6, tile = t3 % NT, pi = t3 / NT;
          *reinterpret_cast<abf_u32x2*>(sb + im * G::IMG_STRIDE + pi * G::PIECE_BYTES + tile * G::TILE_BYTES + 16 * NT * 32 + q8 * 8) = abf_u32x2{0u, 0u};
        }
        // (ordered before the first read of those rows by the barrier after the next chain's first image store)
      }
    }
  }
  if (a < d) {
#pragma unroll
    for (int tj = 0; tj < NT; ++tj)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int b = b0 + 16 * tj + i;
        if (b < d) po[b] = out[tj][i];
      }
  }
}
