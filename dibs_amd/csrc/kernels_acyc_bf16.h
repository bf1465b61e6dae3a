// acyclicity-constraint gradient, 33 <= d <= 64: matrix powers on the bf16 MFMA with three-way split operands (gfx950)
#pragma once
#include "common.h"
#include "kernels_acyc.h"

// ------------------------------------------------------------------------------------------------
// K5b  same computation as k_acyc (kernels_acyc.h; reference: graph_utils.py:8-28, dibs.py:121-140, 557-601), other arithmetic
//      unit.  A float x is carried as three bf16 pieces x = h + m + l (each the round-to-nearest bf16 of the running residual,
//      24+ mantissa bits in total) and a product A B is evaluated as
//          Ah Bh + (Ah Bm + Am Bh) + (Am Bm + Ah Bl + Al Bh)          (the dropped terms are <= 2^-24 relative)
//      with v_mfma_f32_16x16x32_bf16 accumulating in float: 6 instructions of 16 cycles for a 16 x 16 x 32 block where the f32
//      MFMA needs 8 of 32 cycles.  Products of bf16 values are exact in float, so the result differs from a float matmul only
//      in summation order and the 2^-24 tail.
//
// One block = one pair of Monte-Carlo chains of one particle (as k_acyc), 4 waves; wave w owns rows 16 w .. 16 w + 15 of every
// power.  The MFMA is issued with its operands SWAPPED (srcA = fragment of the right-hand matrix, srcB = fragment of the wave's
// own rows), which leaves the transposed tile in the accumulators: lane (g, r) = (lane / 16, lane % 16) holds row r of the
// wave's rows and columns 16 tj + 4 g + i (tile tj, register i) -- exactly the layout of a LEFT-operand fragment of the next
// product (row = lane % 16, eight k-values per lane; the order of k within a k-step is free as long as both operands agree).
// The running power therefore never leaves the registers as a left operand; only its right-operand image goes through LDS:
//   image[piece p][column tile tj][row k][16 columns] bf16, 32 bytes per row,
// written with one ds_write_b64 per (tile, piece) (4 consecutive columns of the lane's row) and read with
// ds_read_b64_tr_b16, whose 4 x 16 block transpose hands lane (g, c) rows k0 + 4 g .. + 3 of column c: the k order
// 16 (2 ks + e / 4) + 4 g + e % 4 (e = 0..7) of the register operand.  Both access patterns are 512 contiguous bytes per
// wave-instruction (conflict-free).  Two images (ping-pong) = 48 KiB per block: three blocks per CU.
// Powers of one matrix commute, so "times M" steps of the binary powering are evaluated as M * P with M's fragments rebuilt
// from the soft graph the thread drew itself -- M needs no image after the first squaring.
// The last power is written as float [row][68] into the free image and read back transposed for
//   out[a][b] += (M^{d-1})[b][a] * tau * alpha * g (1 - g).
// grid = (ceil(Sa / 2 / cpb), Mloc rounded up to 8; re-indexed XCD-aware inside), block = 256,
// dynamic LDS = 2 * 24576
//
// Where the time goes at the headline size (2 048 blocks, 94 us; counters of scripts/probe/acyc_bf_probe.hip): 5.5 M MFMAs = 86 k cycles
// per SIMD, 26 M other vector instructions (a third each: Threefry draws, operand splits, the rest) = ~102 k cycles; SQ_ACTIVE_INST_ANY
// is 94 % of the launch, i.e. the kernel runs at the sum of the two.  Tried and dropped, each measured on the box:
//  * fragments of the next tile pair loaded ahead of the current MFMAs: +-0 (three waves per SIMD hide the LDS latency);
//  * two blocks per CU with 256 registers: 120 us;
//  * the block's sum kept in `part` (read-modify-write per chain) instead of 16 registers: +10 us;
//  * M's fragments kept for the whole chain instead of rebuilt for the two "times M" steps: spills, 114 us;
//  * soft graphs drawn in element order (every lane busy, 40 instead of 56 wave-draws per pair) and handed to the owners through LDS:
//    93-95 us, within noise of the plain version.
// ------------------------------------------------------------------------------------------------
typedef short abf_s16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 abf_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 abf_bf16x2 __attribute__((ext_vector_type(2)));
typedef float abf_f32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t abf_u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t abf_u32x4 __attribute__((ext_vector_type(4)));

constexpr int ABF_NT = 4, ABF_KROWS = 64, ABF_TILE_BYTES = ABF_KROWS * 32, ABF_PIECE_BYTES = ABF_NT * ABF_TILE_BYTES,
              ABF_IMG_BYTES = 3 * ABF_PIECE_BYTES, ABF_LDT = 68;

__device__ __forceinline__ uint32_t abf_cvt_pk(float x0, float x1) {
  const abf_bf16x2 v = __builtin_convertvector(abf_f32x2{x0, x1}, abf_bf16x2);  // v_cvt_pk_bf16_f32 (round to nearest even)
  return __builtin_bit_cast(uint32_t, v);
}
// (x0, x1) -> packed bf16 pairs h, m, l with x = h + m + l up to 2^-25 relative
__device__ __forceinline__ void abf_split(float x0, float x1, uint32_t& h, uint32_t& m, uint32_t& l) {
  h = abf_cvt_pk(x0, x1);
  float r0 = x0 - __uint_as_float(h << 16), r1 = x1 - __uint_as_float(h & 0xffff0000u);  // exact
  m = abf_cvt_pk(r0, r1);
  r0 -= __uint_as_float(m << 16);
  r1 -= __uint_as_float(m & 0xffff0000u);
  l = abf_cvt_pk(r0, r1);
}

// left-operand fragments A[ks][piece] of the 16 values v[tj][i] a lane holds; the same packed pairs go to the image
struct AbfFrag {
  abf_u32x4 a[2][3];
};
__device__ __forceinline__ void abf_make_frag(const f32x4 (&v)[ABF_NT], AbfFrag& f) {
#pragma unroll
  for (int tj = 0; tj < ABF_NT; ++tj) {
    uint32_t h0, m0, l0, h1, m1, l1;
    abf_split(v[tj][0], v[tj][1], h0, m0, l0);
    abf_split(v[tj][2], v[tj][3], h1, m1, l1);
    const int ks = tj >> 1;
    if (tj & 1) {
      f.a[ks][0].z = h0; f.a[ks][0].w = h1;
      f.a[ks][1].z = m0; f.a[ks][1].w = m1;
      f.a[ks][2].z = l0; f.a[ks][2].w = l1;
    } else {
      f.a[ks][0].x = h0; f.a[ks][0].y = h1;
      f.a[ks][1].x = m0; f.a[ks][1].y = m1;
      f.a[ks][2].x = l0; f.a[ks][2].y = l1;
    }
  }
}
// image rows of this lane (row 16 wave + r, columns 4 g .. 4 g + 3 of every tile)
__device__ __forceinline__ void abf_store_image(unsigned char* img, int wr_off, const AbfFrag& f) {
#pragma unroll
  for (int tj = 0; tj < ABF_NT; ++tj)
#pragma unroll
    for (int p = 0; p < 3; ++p) {
      const abf_u32x4 q = f.a[tj >> 1][p];
      const abf_u32x2 w = (tj & 1) ? abf_u32x2{q.z, q.w} : abf_u32x2{q.x, q.y};
      *reinterpret_cast<abf_u32x2*>(img + wr_off + p * ABF_PIECE_BYTES + tj * ABF_TILE_BYTES) = w;
    }
}

__device__ __forceinline__ abf_bf16x8 abf_tr_pair(const unsigned char* p) {
  typedef __attribute__((address_space(3))) abf_s16x4 lds_s16x4;
  const abf_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p));
  const abf_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p + 16 * 32));
  return __builtin_bit_cast(abf_bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
}

// One group = k-step ks of the tile pair (tp, tp + 1): 6 fragment reads and 6 MFMAs per tile.  NU = 1: only tile tp (the other one lies
// beyond column d - 1).  Small terms first; the two tiles alternate so that consecutive MFMAs never share an accumulator.
// Accuracy against a double reference at d = 50 (scripts/probe/acyc_bf_probe.hip): 2.5e-6 of max |out|, the f32-MFMA kernel 3e-7.
// The difference is not the split (x - h - m - l <= 2^-27 x) nor the accumulation order (all ten small products of a tile first,
// Ah Bh last: same 2.5e-6) -- the 32-term dot product inside one bf16 MFMA truncates, and with all entries >= 0 that is a bias
// of ~2^-24 per product level which the powering multiplies by the exponent.
template <int NU>
__device__ __forceinline__ void abf_group(f32x4 (&acc)[ABF_NT], const AbfFrag& A, const unsigned char* img, int rd_off, int ks, int tp) {
  abf_bf16x8 b[NU][3];
#pragma unroll
  for (int u = 0; u < NU; ++u)
#pragma unroll
    for (int p = 0; p < 3; ++p) b[u][p] = abf_tr_pair(img + rd_off + p * ABF_PIECE_BYTES + (tp + u) * ABF_TILE_BYTES + ks * 32 * 32);
  const abf_bf16x8 ah = __builtin_bit_cast(abf_bf16x8, A.a[ks][0]), am = __builtin_bit_cast(abf_bf16x8, A.a[ks][1]),
                   al = __builtin_bit_cast(abf_bf16x8, A.a[ks][2]);
#pragma unroll
  for (int u = 0; u < NU; ++u) {
    if (ks == 0) acc[tp + u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[u][2], ah, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
    else acc[tp + u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[u][2], ah, acc[tp + u], 0, 0, 0);
  }
#pragma unroll
  for (int u = 0; u < NU; ++u) acc[tp + u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[u][0], al, acc[tp + u], 0, 0, 0);
#pragma unroll
  for (int u = 0; u < NU; ++u) acc[tp + u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[u][1], am, acc[tp + u], 0, 0, 0);
#pragma unroll
  for (int u = 0; u < NU; ++u) acc[tp + u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[u][1], ah, acc[tp + u], 0, 0, 0);
#pragma unroll
  for (int u = 0; u < NU; ++u) acc[tp + u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[u][0], am, acc[tp + u], 0, 0, 0);
#pragma unroll
  for (int u = 0; u < NU; ++u) acc[tp + u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[u][0], ah, acc[tp + u], 0, 0, 0);
}
// acc[tj] (row r of the wave's rows, columns 16 tj + 4 g + i) = (rows of A) * (matrix behind the image); four_tiles: d > 48, else the
// fourth column tile is all padding and stays zero.  (Loading the next group's fragments ahead of the current MFMAs changed nothing at
// three waves per SIMD -- the other waves cover the LDS latency -- and costs 24 registers.)
template <bool FOUR>
__device__ __forceinline__ void abf_matmul(f32x4 (&acc)[ABF_NT], const AbfFrag& A, const unsigned char* img, int rd_off) {
  abf_group<2>(acc, A, img, rd_off, 0, 0);
  if constexpr (FOUR) {
    abf_group<2>(acc, A, img, rd_off, 0, 2);
    abf_group<2>(acc, A, img, rd_off, 1, 0);
    abf_group<2>(acc, A, img, rd_off, 1, 2);
  } else {
    abf_group<1>(acc, A, img, rd_off, 0, 2);
    abf_group<2>(acc, A, img, rd_off, 1, 0);
    abf_group<1>(acc, A, img, rd_off, 1, 2);
    acc[3] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
}

// M = I + G~/d in the lane's layout
__device__ __forceinline__ void abf_m0(const f32x4 (&g)[ABF_NT], f32x4 (&v)[ABF_NT], int a, int b0, int d, float inv_d) {
#pragma unroll
  for (int tj = 0; tj < ABF_NT; ++tj)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int b = b0 + 16 * tj + i;
      v[tj][i] = (a == b && a < d) ? 1.0f : g[tj][i] * inv_d;
    }
}

// FOUR: d > 48 (all four row / column tiles in use); the d <= 48 instantiation skips the fourth wave's products and the fourth column tile
template <bool FOUR>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) void k_acyc_bf(const float* __restrict__ scores, float* __restrict__ part, Key2 carry, int m0,
                                                 int M_global, int Mloc, int d, int Sa, int cpb, float alpha, float tau, int layout,
                                                 int tiny, int n_acyc_blk) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  unsigned char* const sb = reinterpret_cast<unsigned char*>(smem);
  // XCD-aware block order (grid = (blocks per particle, particles rounded up to 8)): workgroups go round-robin to the 8 XCDs in the
  // order of their linear id L; L = 8 (gridDim.x p_hi + u) + p_lo puts all blocks u of particle 8 p_hi + p_lo on XCD p_lo, so a
  // particle's score matrix is fetched into one L2 instead of all eight (17 MB of L2 misses per launch otherwise).
  const int L = blockIdx.x + gridDim.x * blockIdx.y, p_lo = L & 7, tq = L >> 3;
  const int bx = tq % (int)gridDim.x, m = (tq / (int)gridDim.x) * 8 + p_lo;
  if (m >= Mloc) return;  // (block-uniform)
  const int blk = bx, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g4 = lane >> 4, r = lane & 15;
  const int a = 16 * wave + r, b0 = 4 * g4;  // row; first column within a tile
  const Key2 km = rng_split_row_uniform(carry, (uint32_t)M_global + 1u, (uint32_t)(m0 + m) + 1u, layout);  // dibs.py:595: key used directly
  const uint64_t dd = (uint64_t)d * d, nbits = (uint64_t)Sa * dd;
  const float inv_d = 1.0f / (float)d;
  const float* sm = scores + (size_t)m * dd;
  const bool fast = tau == 1.0f;  // sigmoid(eps + a) with eps = log(u / (1 - u))  ==  u / (u + (1 - u) exp(-a))
  const float ulo = tiny ? 1.17549435e-38f : 1.1920929e-07f;
  const int n_units = Sa >> 1;  // host: legacy PRNG layout, Sa even, Sa d d < 2^32 (chains sa and sa + Sa/2 share their Threefry calls)
  const TfKeys tk = tf_keys(km);
  // byte offsets of this lane inside an image: writes (row a, columns b0..b0+3), transposing reads (16-lane group g: rows 4 g + r / 4,
  // 8-byte chunk r % 4)
  // The 8-byte chunk c of image row k sits at position (c + (k >> 2)) % 4 of the row: the 16 lanes of one ds_write_b64 group (16 rows,
  // one chunk each) then cover all 32 banks instead of 8 (rows are 8 banks long), and a reading group still sees its 4 x 16 block as
  // 128 contiguous bytes.
  // d <= 48: the fourth wave's rows (48..63) are all padding -- it skips products and splits (its image rows are zeroed once below and
  // nothing overwrites them) -- and the fourth column tile is skipped by every wave: 9 of 16 tile products instead of 16.
  const bool row_active = FOUR || 16 * wave < d;
  const int wr_off = a * 32 + ((g4 + (r >> 2)) & 3) * 8;
  const int rd_off = (4 * g4 + (r >> 2)) * 32 + (((r & 3) + g4) & 3) * 8;
  float* const po = part + ((size_t)m * n_acyc_blk + blk) * dd + (size_t)a * d;
  f32x4 g[ABF_NT], gnext[ABF_NT];
#pragma unroll
  for (int tj = 0; tj < ABF_NT; ++tj) gnext[tj] = f32x4{0.f, 0.f, 0.f, 0.f};
  // (the sum over the block's chains stays in registers: a read-modify-write of `part` per chain instead cost 10 us per launch)
  f32x4 out[ABF_NT];
#pragma unroll
  for (int tj = 0; tj < ABF_NT; ++tj) out[tj] = f32x4{0.f, 0.f, 0.f, 0.f};

  for (int c = 0; c < cpb; ++c) {
    const int unit = blk * cpb + c;
    if (unit >= n_units) break;
    for (int hf = 0; hf < 2; ++hf) {
      const int sa = unit + hf * (Sa >> 1);
      const float* sml = sm;
      asm volatile("" : "+s"(sml));  // opaque per pass: keeps exp(-alpha s) out of loop-invariant registers
      // soft graph of this chain (hf = 0 draws both chains of the pair)
#pragma unroll
      for (int tj = 0; tj < ABF_NT; ++tj)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int b = b0 + 16 * tj + i;
          float gv = 0.f;
          if (a < d && b < d && a != b) {
            if (hf == 1) {
              gv = gnext[tj][i];
            } else {
              const float as = alpha * sml[a * d + b];
              const float ea = fast ? expf(-as) : as;
              uint32_t y0, y1;
              const uint32_t c0 = (uint32_t)((uint64_t)sa * dd) + (uint32_t)(a * d + b);
              threefry2x32_uk(tk, c0, c0 + (uint32_t)(nbits >> 1), y0, y1);
              if (fast) {
                const float u0 = rng_uniform(y0, ulo, 1.0f), u1 = rng_uniform(y1, ulo, 1.0f);
                // (saturated edges -- exp(-alpha s) below the rounding of the denominator, every step once alpha s > ~17 -- must give exactly
                //  1 as the reference's sigmoid does: g (1 - g) = 0 multiplies matrix-power entries of any size; u * rcp(u) is 1 +- 1 ulp)
                const float den0 = fmaf(1.0f - u0, ea, u0), den1 = fmaf(1.0f - u1, ea, u1);
                gv = den0 == u0 ? 1.0f : u0 * __builtin_amdgcn_rcpf(den0);
                gnext[tj][i] = den1 == u1 ? 1.0f : u1 * __builtin_amdgcn_rcpf(den1);
              } else {
                gv = 1.0f / (1.0f + expf(-tau * (rng_logistic(y0, tiny) + ea)));
                gnext[tj][i] = 1.0f / (1.0f + expf(-tau * (rng_logistic(y1, tiny) + ea)));
              }
            }
          }
          g[tj][i] = gv;
        }
      AbfFrag A;
      f32x4 acc[ABF_NT];
      abf_m0(g, acc, a, b0, d, inv_d);
      abf_make_frag(acc, A);
      int cur = 0;  // image holding the running power
      abf_store_image(sb, wr_off, A);
      if (!FOUR && !row_active) abf_store_image(sb + ABF_IMG_BYTES, wr_off, A);  // (all zero; the previous chain's float copy may have covered these rows)
      __syncthreads();
      // left-to-right binary powering of e = d - 1 (>= 32 here)
      const int ex = d - 1;
      const int hb = 31 - __builtin_clz((unsigned)ex);
      for (int bit = hb - 1; bit >= 0; --bit) {
        const bool mult = (ex >> bit) & 1, last_sq = bit == 0 && !mult;
        if (row_active) abf_matmul<FOUR>(acc, A, sb + cur, rd_off);  // P <- P P
        cur ^= ABF_IMG_BYTES;
        if (!last_sq) {
          if (row_active) {
            abf_make_frag(acc, A);
            abf_store_image(sb + cur, wr_off, A);
          }
          __syncthreads();
          if (mult) {  // P <- M P
            f32x4 mv[ABF_NT];
            AbfFrag A0;
            abf_m0(g, mv, a, b0, d, inv_d);
            abf_make_frag(mv, A0);
            if (row_active) abf_matmul<FOUR>(acc, A0, sb + cur, rd_off);
            cur ^= ABF_IMG_BYTES;
            if (bit != 0) {
              if (row_active) {
                abf_make_frag(acc, A);
                abf_store_image(sb + cur, wr_off, A);
              }
              __syncthreads();
            }
          }
        }
      }
      // acc = rows of M^{d-1}; through the free image as float [row][ABF_LDT] to read it transposed
      float* T = reinterpret_cast<float*>(sb + cur);
#pragma unroll
      for (int tj = 0; tj < ABF_NT; ++tj) *reinterpret_cast<f32x4*>(T + a * ABF_LDT + 16 * tj + b0) = acc[tj];
      __syncthreads();
      const float ta = tau * alpha;
#pragma unroll
      for (int tj = 0; tj < ABF_NT; ++tj)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int b = b0 + 16 * tj + i;
          const float gv = g[tj][i];  // 0 outside the matrix and on the diagonal
          const float v = T[b * ABF_LDT + a] * (ta * gv * (1.0f - gv));
          out[tj][i] += v;
        }
      __syncthreads();  // T is the image the next chain's first product writes
    }
  }
  if (a < d) {
#pragma unroll
    for (int tj = 0; tj < ABF_NT; ++tj)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int b = b0 + 16 * tj + i;
        if (b < d) po[b] = out[tj][i];
      }
  }
}

// ------------------------------------------------------------------------------------------------
// K5b'  the same scheme for 65 <= d <= 112: NT = ceil(d / 16) in {5, 6, 7} tiles and NT waves per block (one block = one pair of chains,
//       as above).  Differences to k_acyc_bf: the image of a power has 16 NT rows per column tile (2 x 3 x NT x 16 NT x 32 bytes: 77 / 111 /
//       151 KB -- one block per CU from NT = 6 on), a k-step covers the tile pair (2 ks, 2 ks + 1) of the left operand, and for odd NT the
//       last k-step's second half has no rows in the image: those fragment reads go to a zeroed 512-byte page (the left operand's pieces
//       are zero there anyway, but stale LDS bits could be NaN patterns).  Replaces the f32-MFMA k_acyc<NT> at these sizes (BASELINE
//       config 5 is d = 100): six 16-cycle bf16 MFMAs per 16 x 16 x 32 block instead of eight 32-cycle f32 ones.
// grid = (ceil(Sa / 2 / cpb), Mloc rounded up to 8; re-indexed XCD-aware inside), block = 64 NT, dynamic LDS = abfw_lds_bytes(NT)
// ------------------------------------------------------------------------------------------------
template <int NT>
struct Abfw {
  static constexpr int NKS = (NT + 1) / 2, KROWS = 16 * NT, TILE_BYTES = KROWS * 32, PIECE_BYTES = NT * TILE_BYTES, IMG_BYTES = 3 * PIECE_BYTES,
                       LDT = 16 * NT + 4;
};
__host__ __device__ inline size_t abfw_lds_bytes(int nt) { return (size_t)2 * 3 * nt * (16 * nt) * 32 + 512; }

template <int NT>
struct AbfwFrag {
  abf_u32x4 a[(NT + 1) / 2][3];
};
template <int NT>
__device__ __forceinline__ void abfw_make_frag(const f32x4 (&v)[NT], AbfwFrag<NT>& f) {
#pragma unroll
  for (int tj = 0; tj < NT; ++tj) {
    uint32_t h0, m0, l0, h1, m1, l1;
    abf_split(v[tj][0], v[tj][1], h0, m0, l0);
    abf_split(v[tj][2], v[tj][3], h1, m1, l1);
    const int ks = tj >> 1;
    if (tj & 1) {
      f.a[ks][0].z = h0; f.a[ks][0].w = h1;
      f.a[ks][1].z = m0; f.a[ks][1].w = m1;
      f.a[ks][2].z = l0; f.a[ks][2].w = l1;
    } else {
      f.a[ks][0].x = h0; f.a[ks][0].y = h1;
      f.a[ks][1].x = m0; f.a[ks][1].y = m1;
      f.a[ks][2].x = l0; f.a[ks][2].y = l1;
    }
  }
  if (NT & 1) {  // (the partner tile of the last one does not exist)
#pragma unroll
    for (int p = 0; p < 3; ++p) {
      f.a[NT >> 1][p].z = 0u;
      f.a[NT >> 1][p].w = 0u;
    }
  }
}
template <int NT>
__device__ __forceinline__ void abfw_store_image(unsigned char* img, int wr_off, const AbfwFrag<NT>& f) {
#pragma unroll
  for (int tj = 0; tj < NT; ++tj)
#pragma unroll
    for (int p = 0; p < 3; ++p) {
      const abf_u32x4 q = f.a[tj >> 1][p];
      const abf_u32x2 w = (tj & 1) ? abf_u32x2{q.z, q.w} : abf_u32x2{q.x, q.y};
      *reinterpret_cast<abf_u32x2*>(img + wr_off + p * Abfw<NT>::PIECE_BYTES + tj * Abfw<NT>::TILE_BYTES) = w;
    }
}
// fragment of rows 32 ks .. 32 ks + 31 of one column tile; HALF: only rows 32 ks .. + 15 exist, the other half comes from the zero page
template <bool HALF>
__device__ __forceinline__ abf_bf16x8 abfw_tr_pair(const unsigned char* p, const unsigned char* zero) {
  typedef __attribute__((address_space(3))) abf_s16x4 lds_s16x4;
  const abf_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p));
  const abf_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(HALF ? zero : p + 16 * 32));
  return __builtin_bit_cast(abf_bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
}
// k-step KS of the column tiles tj .. tj + NU - 1: 6 fragment reads and 6 MFMAs per tile (small terms first, as in abf_group); the two
// tiles alternate so that consecutive MFMAs never share an accumulator
template <int NT, int KS, int NU>
__device__ __forceinline__ void abfw_tile_step(f32x4 (&acc)[NT], const AbfwFrag<NT>& A, const unsigned char* img, int rd_off, const unsigned char* zero,
                                               int tj) {
  constexpr bool HALF = (NT & 1) && KS == (NT >> 1);
  abf_bf16x8 b[NU][3];
#pragma unroll
  for (int u = 0; u < NU; ++u)
#pragma unroll
    for (int p = 0; p < 3; ++p)
      b[u][p] = abfw_tr_pair<HALF>(img + rd_off + p * Abfw<NT>::PIECE_BYTES + (tj + u) * Abfw<NT>::TILE_BYTES + KS * 32 * 32, zero);
  const abf_bf16x8 ah = __builtin_bit_cast(abf_bf16x8, A.a[KS][0]), am = __builtin_bit_cast(abf_bf16x8, A.a[KS][1]),
                   al = __builtin_bit_cast(abf_bf16x8, A.a[KS][2]);
#pragma unroll
  for (int u = 0; u < NU; ++u) {
    if (KS == 0) acc[tj + u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[u][2], ah, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
    else acc[tj + u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[u][2], ah, acc[tj + u], 0, 0, 0);
  }
#pragma unroll
  for (int u = 0; u < NU; ++u) acc[tj + u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[u][0], al, acc[tj + u], 0, 0, 0);
#pragma unroll
  for (int u = 0; u < NU; ++u) acc[tj + u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[u][1], am, acc[tj + u], 0, 0, 0);
#pragma unroll
  for (int u = 0; u < NU; ++u) acc[tj + u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[u][1], ah, acc[tj + u], 0, 0, 0);
#pragma unroll
  for (int u = 0; u < NU; ++u) acc[tj + u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[u][0], am, acc[tj + u], 0, 0, 0);
#pragma unroll
  for (int u = 0; u < NU; ++u) acc[tj + u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[u][0], ah, acc[tj + u], 0, 0, 0);
}
template <int NT, int KS>
struct AbfwSteps {
  static __device__ __forceinline__ void run(f32x4 (&acc)[NT], const AbfwFrag<NT>& A, const unsigned char* img, int rd_off, const unsigned char* zero) {
#pragma unroll
    for (int tj = 0; tj + 1 < NT; tj += 2) abfw_tile_step<NT, KS, 2>(acc, A, img, rd_off, zero, tj);
    if (NT & 1) abfw_tile_step<NT, KS, 1>(acc, A, img, rd_off, zero, NT - 1);
    AbfwSteps<NT, KS + 1>::run(acc, A, img, rd_off, zero);
  }
};
template <int NT>
struct AbfwSteps<NT, (NT + 1) / 2> {
  static __device__ __forceinline__ void run(f32x4 (&)[NT], const AbfwFrag<NT>&, const unsigned char*, int, const unsigned char*) {}
};
template <int NT>
__device__ __forceinline__ void abfw_matmul(f32x4 (&acc)[NT], const AbfwFrag<NT>& A, const unsigned char* img, int rd_off, const unsigned char* zero) {
  AbfwSteps<NT, 0>::run(acc, A, img, rd_off, zero);
}
template <int NT>
__device__ __forceinline__ void abfw_m0(const f32x4 (&g)[NT], f32x4 (&v)[NT], int a, int b0, int d, float inv_d) {
#pragma unroll
  for (int tj = 0; tj < NT; ++tj)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int b = b0 + 16 * tj + i;
      v[tj][i] = (a == b && a < d) ? 1.0f : g[tj][i] * inv_d;
    }
}

template <int NT>
__global__ __launch_bounds__(64 * NT) void k_acyc_bfw(const float* __restrict__ scores, float* __restrict__ part, Key2 carry, int m0, int M_global,
                                                      int Mloc, int d, int Sa, int cpb, float alpha, float tau, int layout, int tiny, int n_acyc_blk) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  unsigned char* const sb = reinterpret_cast<unsigned char*>(smem);
  constexpr int IMG = Abfw<NT>::IMG_BYTES, LDT = Abfw<NT>::LDT;
  unsigned char* const zero = sb + 2 * IMG;  // 512 zero bytes
  const int L = blockIdx.x + gridDim.x * blockIdx.y, p_lo = L & 7, tq = L >> 3;
  const int bx = tq % (int)gridDim.x, m = (tq / (int)gridDim.x) * 8 + p_lo;
  if (m >= Mloc) return;  // (block-uniform)
  const int blk = bx, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g4 = lane >> 4, r = lane & 15;
  const int a = 16 * wave + r, b0 = 4 * g4;
  const Key2 km = rng_split_row_uniform(carry, (uint32_t)M_global + 1u, (uint32_t)(m0 + m) + 1u, layout);  // dibs.py:595: key used directly
  const uint64_t dd = (uint64_t)d * d, nbits = (uint64_t)Sa * dd;
  const float inv_d = 1.0f / (float)d;
  const float* sm = scores + (size_t)m * dd;
  const bool fast = tau == 1.0f;
  const float ulo = tiny ? 1.17549435e-38f : 1.1920929e-07f;
  const int n_units = Sa >> 1;
  const TfKeys tk = tf_keys(km);
  const int wr_off = a * 32 + ((g4 + (r >> 2)) & 3) * 8;
  const int rd_off = (4 * g4 + (r >> 2)) * 32 + (((r & 3) + g4) & 3) * 8;
  float* const po = part + ((size_t)m * n_acyc_blk + blk) * dd + (size_t)a * d;
  if (tid < 128) reinterpret_cast<uint32_t*>(zero)[tid] = 0u;
  f32x4 g[NT], gnext[NT], out[NT];
#pragma unroll
  for (int tj = 0; tj < NT; ++tj) {
    gnext[tj] = f32x4{0.f, 0.f, 0.f, 0.f};
    out[tj] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  for (int c = 0; c < cpb; ++c) {
    const int unit = blk * cpb + c;
    if (unit >= n_units) break;
    for (int hf = 0; hf < 2; ++hf) {
      const int sa = unit + hf * (Sa >> 1);
      const float* sml = sm;
      asm volatile("" : "+s"(sml));  // opaque per pass: keeps exp(-alpha s) out of loop-invariant registers
#pragma unroll
      for (int tj = 0; tj < NT; ++tj)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int b = b0 + 16 * tj + i;
          float gv = 0.f;
          if (a < d && b < d && a != b) {
            if (hf == 1) {
              gv = gnext[tj][i];
            } else {
              const float as = alpha * sml[a * d + b];
              const float ea = fast ? expf(-as) : as;
              uint32_t y0, y1;
              const uint32_t c0 = (uint32_t)((uint64_t)sa * dd) + (uint32_t)(a * d + b);
              threefry2x32_uk(tk, c0, c0 + (uint32_t)(nbits >> 1), y0, y1);
              if (fast) {
                const float u0 = rng_uniform(y0, ulo, 1.0f), u1 = rng_uniform(y1, ulo, 1.0f);
                const float den0 = fmaf(1.0f - u0, ea, u0), den1 = fmaf(1.0f - u1, ea, u1);
                gv = den0 == u0 ? 1.0f : u0 * __builtin_amdgcn_rcpf(den0);   // (saturated edges give exactly 1: see k_acyc_bf)
                gnext[tj][i] = den1 == u1 ? 1.0f : u1 * __builtin_amdgcn_rcpf(den1);
              } else {
                gv = 1.0f / (1.0f + expf(-tau * (rng_logistic(y0, tiny) + ea)));
                gnext[tj][i] = 1.0f / (1.0f + expf(-tau * (rng_logistic(y1, tiny) + ea)));
              }
            }
          }
          g[tj][i] = gv;
        }
      AbfwFrag<NT> A;
      f32x4 acc[NT];
      abfw_m0<NT>(g, acc, a, b0, d, inv_d);
      abfw_make_frag<NT>(acc, A);
      int cur = 0;
      abfw_store_image<NT>(sb, wr_off, A);
      __syncthreads();
      const int ex = d - 1;
      const int hb = 31 - __builtin_clz((unsigned)ex);
      for (int bit = hb - 1; bit >= 0; --bit) {
        const bool mult = (ex >> bit) & 1, last_sq = bit == 0 && !mult;
        abfw_matmul<NT>(acc, A, sb + cur, rd_off, zero);  // P <- P P
        cur ^= IMG;
        if (!last_sq) {
          abfw_make_frag<NT>(acc, A);
          abfw_store_image<NT>(sb + cur, wr_off, A);
          __syncthreads();
          if (mult) {  // P <- M P
            f32x4 mv[NT];
            AbfwFrag<NT> A0;
            abfw_m0<NT>(g, mv, a, b0, d, inv_d);
            abfw_make_frag<NT>(mv, A0);
            abfw_matmul<NT>(acc, A0, sb + cur, rd_off, zero);
            cur ^= IMG;
            if (bit != 0) {
              abfw_make_frag<NT>(acc, A);
              abfw_store_image<NT>(sb + cur, wr_off, A);
              __syncthreads();
            }
          }
        }
      }
      float* T = reinterpret_cast<float*>(sb + cur);
#pragma unroll
      for (int tj = 0; tj < NT; ++tj) *reinterpret_cast<f32x4*>(T + a * LDT + 16 * tj + b0) = acc[tj];
      __syncthreads();
      const float ta = tau * alpha;
#pragma unroll
      for (int tj = 0; tj < NT; ++tj)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int b = b0 + 16 * tj + i;
          const float gv = g[tj][i];
          out[tj][i] += T[b * LDT + a] * (ta * gv * (1.0f - gv));
        }
      __syncthreads();
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
