// Counter-based PRNG for the engine: Threefry-2x32-20 and the JAX layering the reference relies on
// (jax.random.split / bits / uniform / normal / bernoulli / logistic; call sites:
//  dibs/inference/svgd.py:145-146,245,251,294,509-513,695-703 and dibs/inference/dibs.py:115,350,431,595).
// Host+device so the host can advance the loop-carry key while kernels derive per-particle keys.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define DIBS_HD __host__ __device__ __forceinline__

struct Key2 {
  uint32_t a, b;
};

DIBS_HD uint32_t rotl32(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }

DIBS_HD void threefry2x32(uint32_t k0, uint32_t k1, uint32_t c0, uint32_t c1, uint32_t& o0, uint32_t& o1) {
  const uint32_t k2 = k0 ^ k1 ^ 0x1BD11BDAu;
  uint32_t x0 = c0 + k0, x1 = c1 + k1;
#define TF_R(r) x0 += x1; x1 = rotl32(x1, r); x1 ^= x0;
  TF_R(13) TF_R(15) TF_R(26) TF_R(6)
  x0 += k1; x1 += k2 + 1u;
  TF_R(17) TF_R(29) TF_R(16) TF_R(24)
  x0 += k2; x1 += k0 + 2u;
  TF_R(13) TF_R(15) TF_R(26) TF_R(6)
  x0 += k0; x1 += k1 + 3u;
  TF_R(17) TF_R(29) TF_R(16) TF_R(24)
  x0 += k1; x1 += k2 + 4u;
  TF_R(13) TF_R(15) TF_R(26) TF_R(6)
  x0 += k2; x1 += k0 + 5u;
#undef TF_R
  o0 = x0;
  o1 = x1;
}

// split row for wave-uniform arguments (key, row from kernel arguments / blockIdx): marking them uniform lets the compiler run
// the whole derivation on the scalar unit instead of spending ~70 VALU issue slots per Threefry call in every wave
__device__ __forceinline__ Key2 rng_split_row_uniform(Key2 key, uint32_t num, uint32_t r, int layout);

// Threefry for wave-uniform keys, as one fixed instruction sequence (67 VALU ops): the key schedule lives in SGPRs,
// each injection is folded into the following round's add (v_add3_u32), and the optimiser cannot re-derive the first rounds
// as extra induction variables (it did, at +50 % instructions, when this ran inside a counter loop).  The sampling kernels are
// bound by VALU issue, so the instruction count of this routine is their speed.
struct TfKeys {
  uint32_t k0, k1, k2, k2p1, k0p2, k1p3, k2p4, k0p5;
};
__device__ __forceinline__ TfKeys tf_keys(Key2 key) {
  TfKeys K;
#if defined(__HIP_DEVICE_COMPILE__)
  K.k0 = __builtin_amdgcn_readfirstlane(key.a);
  K.k1 = __builtin_amdgcn_readfirstlane(key.b);
#else
  K.k0 = key.a;
  K.k1 = key.b;
#endif
  K.k2 = K.k0 ^ K.k1 ^ 0x1BD11BDAu;
  K.k2p1 = K.k2 + 1u;
  K.k0p2 = K.k0 + 2u;
  K.k1p3 = K.k1 + 3u;
  K.k2p4 = K.k2 + 4u;
  K.k0p5 = K.k0 + 5u;
  return K;
}
#define TF_RA(r) "v_add_u32 %0, %0, %1\n\tv_alignbit_b32 %1, %1, %1, " #r "\n\tv_xor_b32 %1, %1, %0\n\t"
#define TF_RI(ka, kb, r) "v_add_u32 %1, " kb ", %1\n\tv_add3_u32 %0, %0, " ka ", %1\n\tv_alignbit_b32 %1, %1, %1, " #r "\n\tv_xor_b32 %1, %1, %0\n\t"
__device__ __forceinline__ void threefry2x32_uk(const TfKeys& K, uint32_t c0, uint32_t c1, uint32_t& o0, uint32_t& o1) {
#if defined(__HIP_DEVICE_COMPILE__)
  uint32_t x0 = c0, x1 = c1;
  // alignbit(x, x, 32 - r) == rotl(x, r)
  asm volatile(
      TF_RI("%2", "%3", 19) TF_RA(17) TF_RA(6) TF_RA(26)          /* x = c + (k0, k1); rounds 13 15 26 6 */
      TF_RI("%3", "%5", 15) TF_RA(3) TF_RA(16) TF_RA(8)           /* += (k1, k2 + 1);  rounds 17 29 16 24 */
      TF_RI("%4", "%6", 19) TF_RA(17) TF_RA(6) TF_RA(26)          /* += (k2, k0 + 2) */
      TF_RI("%2", "%7", 15) TF_RA(3) TF_RA(16) TF_RA(8)           /* += (k0, k1 + 3) */
      TF_RI("%3", "%8", 19) TF_RA(17) TF_RA(6) TF_RA(26)          /* += (k1, k2 + 4) */
      "v_add_u32 %0, %4, %0\n\tv_add_u32 %1, %9, %1"            /* += (k2, k0 + 5) */
      : "+v"(x0), "+v"(x1)
      : "s"(K.k0), "s"(K.k1), "s"(K.k2), "s"(K.k2p1), "s"(K.k0p2), "s"(K.k1p3), "s"(K.k2p4), "s"(K.k0p5));
  o0 = x0;
  o1 = x1;
#else  // host pass of the single-source compile: never executed
  threefry2x32(K.k0, K.k1, c0, c1, o0, o1);
#endif
}
#undef TF_RA
#undef TF_RI
// two independent calls interleaved instruction by instruction (dependent VALU ops of one chain do not issue back to back
// at full rate; two chains per wave fill the gaps)
#define TF2_RA(r) "v_add_u32 %0, %0, %1\n\tv_add_u32 %2, %2, %3\n\tv_alignbit_b32 %1, %1, %1, " #r "\n\tv_alignbit_b32 %3, %3, %3, " #r \
                  "\n\tv_xor_b32 %1, %1, %0\n\tv_xor_b32 %3, %3, %2\n\t"
#define TF2_RI(ka, kb, r) "v_add_u32 %1, " kb ", %1\n\tv_add_u32 %3, " kb ", %3\n\tv_add3_u32 %0, %0, " ka ", %1\n\tv_add3_u32 %2, %2, " ka ", %3\n\t" \
                          "v_alignbit_b32 %1, %1, %1, " #r "\n\tv_alignbit_b32 %3, %3, %3, " #r "\n\tv_xor_b32 %1, %1, %0\n\tv_xor_b32 %3, %3, %2\n\t"
__device__ __forceinline__ void threefry2x32_uk2(const TfKeys& K, uint32_t ca0, uint32_t ca1, uint32_t cb0, uint32_t cb1, uint32_t& oa0,
                                                 uint32_t& oa1, uint32_t& ob0, uint32_t& ob1) {
#if defined(__HIP_DEVICE_COMPILE__)
  uint32_t x0 = ca0, x1 = ca1, z0 = cb0, z1 = cb1;
  asm volatile(
      TF2_RI("%4", "%5", 19) TF2_RA(17) TF2_RA(6) TF2_RA(26)
      TF2_RI("%5", "%7", 15) TF2_RA(3) TF2_RA(16) TF2_RA(8)
      TF2_RI("%6", "%8", 19) TF2_RA(17) TF2_RA(6) TF2_RA(26)
      TF2_RI("%4", "%9", 15) TF2_RA(3) TF2_RA(16) TF2_RA(8)
      TF2_RI("%5", "%10", 19) TF2_RA(17) TF2_RA(6) TF2_RA(26)
      "v_add_u32 %0, %6, %0\n\tv_add_u32 %2, %6, %2\n\tv_add_u32 %1, %11, %1\n\tv_add_u32 %3, %11, %3"
      : "+v"(x0), "+v"(x1), "+v"(z0), "+v"(z1)
      : "s"(K.k0), "s"(K.k1), "s"(K.k2), "s"(K.k2p1), "s"(K.k0p2), "s"(K.k1p3), "s"(K.k2p4), "s"(K.k0p5));
  oa0 = x0; oa1 = x1; ob0 = z0; ob1 = z1;
#else
  threefry2x32(K.k0, K.k1, ca0, ca1, oa0, oa1);
  threefry2x32(K.k0, K.k1, cb0, cb1, ob0, ob1);
#endif
}
#undef TF2_RA
#undef TF2_RI

// element i of random_bits(key, n).  layout 0 = legacy (counts split in halves, concat(y0, y1)),
// 1 = partitionable (counter (hi, lo) = i, y0 ^ y1).
DIBS_HD uint32_t rng_bits_at(Key2 key, uint64_t n, uint64_t i, int layout) {
  uint32_t y0, y1;
  if (layout == 1) {
    threefry2x32(key.a, key.b, (uint32_t)(i >> 32), (uint32_t)i, y0, y1);
    return y0 ^ y1;
  }
  const uint64_t half = (n + 1) / 2;
  const uint64_t c = i < half ? i : i - half;
  uint32_t c1 = (uint32_t)(half + c);
  if ((n & 1) && c == half - 1) c1 = 0;
  threefry2x32(key.a, key.b, (uint32_t)c, c1, y0, y1);
  return i < half ? y0 : y1;
}

// the two elements c and c + n/2 of random_bits(key, n) for EVEN n (one Threefry call in the legacy layout)
DIBS_HD void rng_bits_pair(Key2 key, uint64_t n, uint64_t c, int layout, uint32_t& lo, uint32_t& hi) {
  const uint64_t half = n >> 1;
  if (layout == 1) {
    uint32_t y0, y1;
    threefry2x32(key.a, key.b, (uint32_t)(c >> 32), (uint32_t)c, y0, y1);
    lo = y0 ^ y1;
    const uint64_t c2 = c + half;
    threefry2x32(key.a, key.b, (uint32_t)(c2 >> 32), (uint32_t)c2, y0, y1);
    hi = y0 ^ y1;
  } else {
    threefry2x32(key.a, key.b, (uint32_t)c, (uint32_t)(half + c), lo, hi);
  }
}

// Explicit per-particle keys (dibs_engine_eval_gradients: the reference's eltwise_grad_* methods take `subkeys` [n_particles, 2] instead of
// deriving them from a loop-carry key).  Every kernel derives a particle's key as row 1 + m of split(carry, M + 1); with M = -1 (num == 0,
// never a real split) the carry slot holds the DEVICE ADDRESS of a Key2 array indexed by the global particle id, and "row r" is entry r - 1.
__device__ __forceinline__ Key2 rng_explicit_row(Key2 key, uint32_t r) {
  const Key2* p = reinterpret_cast<const Key2*>(((uint64_t)key.b << 32) | (uint64_t)key.a);
  return p[r - 1u];
}

// row r of jax.random.split(key, num)
DIBS_HD Key2 rng_split_row(Key2 key, uint32_t num, uint32_t r, int layout) {
  Key2 o;
#if defined(__HIP_DEVICE_COMPILE__)
  if (num == 0u) return rng_explicit_row(key, r);
#endif
  if (layout == 1) {
    threefry2x32(key.a, key.b, 0u, r, o.a, o.b);
    return o;
  }
  o.a = rng_bits_at(key, 2ull * num, 2ull * r, 0);
  o.b = rng_bits_at(key, 2ull * num, 2ull * r + 1, 0);
  return o;
}

DIBS_HD float rng_unit_float(uint32_t bits) {
  union {
    uint32_t u;
    float f;
  } v;
  v.u = (bits >> 9) | 0x3F800000u;
  return v.f - 1.0f;
}

// correctly-rounded single ops that the compiler may not contract into an fma
DIBS_HD float dibs_fmul(float a, float b) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __fmul_rn(a, b);
#else
  volatile float r = a * b;
  return r;
#endif
}
DIBS_HD float dibs_fadd(float a, float b) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __fadd_rn(a, b);
#else
  volatile float r = a + b;
  return r;
#endif
}
#define DIBS_FMUL(a, b) dibs_fmul(a, b)
#define DIBS_FADD(a, b) dibs_fadd(a, b)

// jax.random.uniform(minval=lo, maxval=hi): max(lo, floats * (hi - lo) + lo), two roundings (no fma)
DIBS_HD float rng_uniform(uint32_t bits, float lo, float hi) {
  const float v = DIBS_FADD(DIBS_FMUL(rng_unit_float(bits), hi - lo), lo);
  return v > lo ? v : lo;
}

// jax.random.logistic: x = uniform(eps|tiny, 1); log(x / (1 - x))
__device__ __forceinline__ float rng_logistic(uint32_t bits, int tiny) {
  const float lo = tiny ? 1.17549435e-38f : 1.1920929e-07f;
  const float x = rng_uniform(bits, lo, 1.0f);
  return logf(x / (1.0f - x));
}

// jax.random.normal: sqrt(2) * erfinv(uniform(nextafter(-1, 0), 1)); erfinv = Giles' single-precision
// polynomial (the form XLA lowers lax.erf_inv(f32) to)
__device__ __forceinline__ float rng_normal(uint32_t bits) {
  const float x = rng_uniform(bits, -0.99999994f, 1.0f);
  float w = (float)(-log1p((double)DIBS_FMUL(-x, x)));
  const bool lt = w < 5.0f;
  w = lt ? w - 2.5f : sqrtf(w) - 3.0f;
  const float A[9] = {2.81022636e-08f, 3.43273939e-07f, -3.5233877e-06f, -4.39150654e-06f, 0.00021858087f,
                      -0.00125372503f, -0.00417768164f, 0.246640727f, 1.50140941f};
  const float B[9] = {-0.000200214257f, 0.000100950558f, 0.00134934322f, -0.00367342844f, 0.00573950773f,
                      -0.0076224613f, 0.00943887047f, 1.00167406f, 2.83297682f};
  float p = lt ? A[0] : B[0];
#pragma unroll
  for (int i = 1; i < 9; ++i) p = DIBS_FADD(lt ? A[i] : B[i], DIBS_FMUL(p, w));
  return DIBS_FMUL(1.41421354f, DIBS_FMUL(p, x));
}

// Threefry on the scalar unit (SALU has no rotate: shift, shift, or).  hipcc selects v_alignbit_b32 for a rotate even when
// the operands are uniform, so the rounds are spelled out.
#define TFS_R(r, rr) "s_add_u32 %0, %0, %1\n\ts_lshl_b32 %2, %1, " #r "\n\ts_lshr_b32 %1, %1, " #rr "\n\ts_or_b32 %1, %1, %2\n\ts_xor_b32 %1, %1, %0\n\t"
__device__ __forceinline__ void threefry2x32_scalar(uint32_t k0, uint32_t k1, uint32_t c0, uint32_t c1, uint32_t& o0, uint32_t& o1) {
#if defined(__HIP_DEVICE_COMPILE__)
  const uint32_t k2 = k0 ^ k1 ^ 0x1BD11BDAu;
  uint32_t x0 = c0 + k0, x1 = c1 + k1, t;
  asm volatile(TFS_R(13, 19) TFS_R(15, 17) TFS_R(26, 6) TFS_R(6, 26) : "+s"(x0), "+s"(x1), "=&s"(t) : : "scc");
  x0 += k1; x1 += k2 + 1u;
  asm volatile(TFS_R(17, 15) TFS_R(29, 3) TFS_R(16, 16) TFS_R(24, 8) : "+s"(x0), "+s"(x1), "=&s"(t) : : "scc");
  x0 += k2; x1 += k0 + 2u;
  asm volatile(TFS_R(13, 19) TFS_R(15, 17) TFS_R(26, 6) TFS_R(6, 26) : "+s"(x0), "+s"(x1), "=&s"(t) : : "scc");
  x0 += k0; x1 += k1 + 3u;
  asm volatile(TFS_R(17, 15) TFS_R(29, 3) TFS_R(16, 16) TFS_R(24, 8) : "+s"(x0), "+s"(x1), "=&s"(t) : : "scc");
  x0 += k1; x1 += k2 + 4u;
  asm volatile(TFS_R(13, 19) TFS_R(15, 17) TFS_R(26, 6) TFS_R(6, 26) : "+s"(x0), "+s"(x1), "=&s"(t) : : "scc");
  o0 = x0 + k2;
  o1 = x1 + k0 + 5u;
#else
  threefry2x32(k0, k1, c0, c1, o0, o1);
#endif
}
#undef TFS_R

__device__ __forceinline__ Key2 rng_split_row_uniform(Key2 key, uint32_t num, uint32_t r, int layout) {
#if defined(__HIP_DEVICE_COMPILE__)
  const uint32_t k0 = __builtin_amdgcn_readfirstlane(key.a), k1 = __builtin_amdgcn_readfirstlane(key.b);
  num = __builtin_amdgcn_readfirstlane(num);
  r = __builtin_amdgcn_readfirstlane(r);
  Key2 o;
  if (num == 0u) {  // explicit per-particle keys (see rng_explicit_row)
    const Key2 x = rng_explicit_row(Key2{k0, k1}, r);
    o.a = __builtin_amdgcn_readfirstlane(x.a);
    o.b = __builtin_amdgcn_readfirstlane(x.b);
    return o;
  }
  if (layout == 1) {
    threefry2x32_scalar(k0, k1, 0u, r, o.a, o.b);
    return o;
  }
  // legacy: element i of the 2 num draws comes from the call with counters (c, num + c), c = i mod num; first / second output
  uint32_t y0, y1;
  const uint32_t i0 = 2u * r, i1 = 2u * r + 1u;
  const uint32_t ca = i0 < num ? i0 : i0 - num, cb = i1 < num ? i1 : i1 - num;
  threefry2x32_scalar(k0, k1, ca, num + ca, y0, y1);
  o.a = i0 < num ? y0 : y1;
  threefry2x32_scalar(k0, k1, cb, num + cb, y0, y1);
  o.b = i1 < num ? y0 : y1;
  return o;
#else
  return rng_split_row(key, num, r, layout);
#endif
}
