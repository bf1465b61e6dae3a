/*
 * ORACLE (test infrastructure only -- never linked or called by the product path).
 *
 * Plain-C port of the DiBS SVGD step of larslorch/dibs with closed-form gradients.  It restates the
 * same reference functions as oracle/dibs_oracle.py (which uses autograd where the reference uses
 * jax.grad); tests/test_oracle_c.py pins this port against that file, and tests/test_prng.py pins the
 * PRNG layer against public known-answer values.  Used (a) as the checker for the HIP kernels at sizes
 * the Python oracle is too slow for, (b) as bench.py's cpu_baseline ("port").
 *
 * Build:  make -C oracle      ->  oracle/_build/liboracle_f64.so (real = double), liboracle_f32.so (float)
 *
 * Reference citations (paths relative to /root/reference):
 *   step orchestration         dibs/inference/svgd.py:226-267 (marginal), 673-721 (joint)
 *   edge probs / soft graphs   dibs/inference/dibs.py:102-140, 168-184
 *   score-function estimator   dibs/inference/dibs.py:325-391   (closed form: W = alpha (sum_s w_s G_s - P))
 *   reparam estimator          dibs/inference/dibs.py:395-459
 *   theta estimator            dibs/inference/dibs.py:488-551
 *   acyclicity                 dibs/graph_utils.py:8-28, dibs/inference/dibs.py:557-601
 *   latent prior               dibs/inference/dibs.py:604-658, dibs/models/graph.py:93-108, 182-196, 263-276
 *   BGe                        dibs/models/linearGaussian.py:63-170, dibs/utils/func.py:128-145
 *   LinearGaussian             dibs/models/linearGaussian.py:212-227, 278-338
 *   DenseNonlinearGaussian     dibs/models/nonlinearGaussian.py:155-186, 248-326
 *   kernel / phi               dibs/kernel.py:20-30, 52-71; dibs/inference/svgd.py:165-224, 537-670
 *   optimizer, PRNG            third-party jax (see oracle/prng.py header)
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#ifdef _OPENMP
#include <omp.h>
#endif
#include "../include/dibs_hip.h"

#ifdef ORACLE_F32
typedef float real;
#define R_EXP expf
#define R_LOG logf
#define R_SQRT sqrtf
#define R_FMA fmaf
#else
typedef double real;
#define R_EXP exp
#define R_LOG log
#define R_SQRT sqrt
#define R_FMA(a, b, c) ((a) * (b) + (c))
#endif

#define ORC_EXPORT __attribute__((visibility("default")))

/* ------------------------------------------------------------------------------------------------
 * PRNG: Threefry-2x32-20 + the JAX layering (see oracle/prng.py for the derivation and citations)
 * ---------------------------------------------------------------------------------------------- */
static inline uint32_t rotl32(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }

ORC_EXPORT void orc_threefry2x32(uint32_t k0, uint32_t k1, uint32_t c0, uint32_t c1, uint32_t* o0, uint32_t* o1) {
  static const int R[2][4] = {{13, 15, 26, 6}, {17, 29, 16, 24}};
  uint32_t ks[3] = {k0, k1, k0 ^ k1 ^ 0x1BD11BDAu};
  uint32_t x0 = c0 + ks[0], x1 = c1 + ks[1];
  for (int b = 0; b < 5; ++b) {
    for (int r = 0; r < 4; ++r) {
      x0 += x1;
      x1 = rotl32(x1, R[b & 1][r]);
      x1 ^= x0;
    }
    x0 += ks[(b + 1) % 3];
    x1 += ks[(b + 2) % 3] + (uint32_t)(b + 1);
  }
  *o0 = x0;
  *o1 = x1;
}

/* element i of random_bits(key, n) */
static inline uint32_t bits_at(const uint32_t key[2], int64_t n, int64_t i, int layout) {
  uint32_t y0, y1;
  if (layout == DIBS_RNG_PARTITIONABLE) {
    orc_threefry2x32(key[0], key[1], (uint32_t)((uint64_t)i >> 32), (uint32_t)i, &y0, &y1);
    return y0 ^ y1;
  }
  int64_t half = (n + 1) / 2;
  int64_t c = i < half ? i : i - half;
  uint32_t c1 = (uint32_t)(half + c);
  if ((n & 1) && c == half - 1) c1 = 0; /* odd tail padded with a zero count */
  orc_threefry2x32(key[0], key[1], (uint32_t)c, c1, &y0, &y1);
  return i < half ? y0 : y1;
}

ORC_EXPORT void orc_random_bits(const uint32_t key[2], int64_t n, int layout, uint32_t* out) {
  for (int64_t i = 0; i < n; ++i) out[i] = bits_at(key, n, i, layout);
}

/* row r of split(key, num) */
static inline void split_row(const uint32_t key[2], int num, int r, int layout, uint32_t out[2]) {
  if (layout == DIBS_RNG_PARTITIONABLE) {
    orc_threefry2x32(key[0], key[1], 0u, (uint32_t)r, &out[0], &out[1]);
    return;
  }
  out[0] = bits_at(key, 2 * (int64_t)num, 2 * (int64_t)r, layout);
  out[1] = bits_at(key, 2 * (int64_t)num, 2 * (int64_t)r + 1, layout);
}

ORC_EXPORT void orc_split(const uint32_t key[2], int num, int layout, uint32_t* out) {
  for (int r = 0; r < num; ++r) split_row(key, num, r, layout, out + 2 * r);
}

static inline float unit_float(uint32_t bits) {
  union { uint32_t u; float f; } v;
  v.u = (bits >> 9) | 0x3F800000u;
  return v.f - 1.0f;
}

static inline float uniform_from_bits(uint32_t bits, float lo, float hi) {
  volatile float span = hi - lo; /* volatile: keep the two roundings separate (no fma contraction) */
  volatile float prod = unit_float(bits) * span;
  float v = prod + lo;
  return v > lo ? v : lo;
}

static float erfinv_f32(float x) {
  static const float A[9] = {2.81022636e-08f, 3.43273939e-07f, -3.5233877e-06f, -4.39150654e-06f, 0.00021858087f,
                             -0.00125372503f, -0.00417768164f, 0.246640727f, 1.50140941f};
  static const float B[9] = {-0.000200214257f, 0.000100950558f, 0.00134934322f, -0.00367342844f, 0.00573950773f,
                             -0.0076224613f, 0.00943887047f, 1.00167406f, 2.83297682f};
  volatile float xx = -x * x;
  float w = (float)(-log1p((double)xx));
  const float* c = w < 5.0f ? A : B;
  w = w < 5.0f ? w - 2.5f : sqrtf(w) - 3.0f;
  float p = c[0];
  for (int i = 1; i < 9; ++i) {
    volatile float pw = p * w;
    p = c[i] + pw;
  }
  return p * x;
}

static inline float normal_from_bits(uint32_t bits) {
  const float lo = -0.99999994f; /* nextafter(-1, 0) */
  return 1.41421354f * erfinv_f32(uniform_from_bits(bits, lo, 1.0f));
}

static inline float logistic_from_bits(uint32_t bits, int tiny) {
  float lo = tiny ? 1.17549435e-38f : 1.1920929e-07f;
  float x = uniform_from_bits(bits, lo, 1.0f);
  return logf(x / (1.0f - x));
}

ORC_EXPORT void orc_normal(const uint32_t key[2], int64_t n, int layout, float* out) {
  for (int64_t i = 0; i < n; ++i) out[i] = normal_from_bits(bits_at(key, n, i, layout));
}
ORC_EXPORT void orc_logistic(const uint32_t key[2], int64_t n, int layout, int tiny, float* out) {
  for (int64_t i = 0; i < n; ++i) out[i] = logistic_from_bits(bits_at(key, n, i, layout), tiny);
}
ORC_EXPORT void orc_uniform(const uint32_t key[2], int64_t n, int layout, float* out) {
  for (int64_t i = 0; i < n; ++i) out[i] = uniform_from_bits(bits_at(key, n, i, layout), 0.0f, 1.0f);
}

/* ------------------------------------------------------------------------------------------------
 * sizes
 * ---------------------------------------------------------------------------------------------- */
ORC_EXPORT int64_t orc_theta_size(const dibs_config* c) {
  int d = c->n_vars;
  if (!c->joint) return 0;
  if (c->likelihood == DIBS_LIK_LINGAUSS) return (int64_t)d * d;
  if (c->likelihood == DIBS_LIK_DENSENN) {
    int64_t p = 0;
    int in = d;
    for (int l = 0; l <= c->nn_n_hidden; ++l) {
      int out = l < c->nn_n_hidden ? c->nn_hidden[l] : 1;
      p += (int64_t)d * in * out + (c->nn_bias ? (int64_t)d * out : 0);
      in = out;
    }
    return p;
  }
  return 0;
}
ORC_EXPORT int orc_real_bytes(void) { return (int)sizeof(real); }

static double latent_std(const dibs_config* c) {
  if (c->latent_prior_std > 0) return c->latent_prior_std;
  return (double)(1.0f / sqrtf((float)c->n_dim)); /* 1/jnp.sqrt(k) in float32  svgd.py:142,302 */
}

/* ------------------------------------------------------------------------------------------------
 * particle initialisation    svgd.py:293-295, 125-148, 489-515; linearGaussian.py:212-227
 * ---------------------------------------------------------------------------------------------- */
ORC_EXPORT int orc_init_particles(const dibs_config* c, const uint32_t key_in[2], real* z, real* theta,
                                  uint32_t key_out[2]) {
  const int L = c->rng_layout;
  const int d = c->n_vars, k = c->n_dim, M = c->n_particles;
  uint32_t ks[4], is[4];
  orc_split(key_in, 2, L, ks); /* key, subk = split(key) */
  key_out[0] = ks[0];
  key_out[1] = ks[1];
  orc_split(ks + 2, 2, L, is); /* inside _sample_initial_random_particles: key, subk = split(key) */
  const float stdf = (float)latent_std(c);
  int64_t n = (int64_t)M * d * k * 2;
  for (int64_t i = 0; i < n; ++i) z[i] = (real)(normal_from_bits(bits_at(is + 2, n, i, L)) * stdf);
  if (c->joint) {
    uint32_t ts[4];
    orc_split(is, 2, L, ts); /* key, subk = split(key) ; theta = sample_parameters(key=subk) */
    if (c->likelihood == DIBS_LIK_LINGAUSS) {
      int64_t nt = (int64_t)M * d * d;
      for (int64_t i = 0; i < nt; ++i) {
        float v = (float)c->lin_mean_edge + (float)c->lin_sig_edge * normal_from_bits(bits_at(ts + 2, nt, i, L));
        float sg = v > 0 ? 1.0f : (v < 0 ? -1.0f : 0.0f);
        theta[i] = (real)(v + sg * (float)c->lin_min_edge);
      }
    } else if (c->likelihood == DIBS_LIK_DENSENN) {
      /* nonlinearGaussian.py:168-178: subkeys = split(key, M*d); per (m, j): stax.serial init
       * (rng, layer_rng = split(rng) per stax layer incl. activations; Dense: k1, k2 = split(layer_rng)) */
      const int64_t P = orc_theta_size(c);
      const int nl = c->nn_n_hidden + 1;
      for (int m = 0; m < M; ++m)
        for (int j = 0; j < d; ++j) {
          uint32_t rng[2];
          split_row(ts + 2, M * d, m * d + j, L, rng);
          int in = d;
          int64_t off = 0; /* offset of leaf group within the particle's theta */
          for (int si = 0; si < 2 * nl - 1; ++si) {
            uint32_t sp[4];
            orc_split(rng, 2, L, sp);
            rng[0] = sp[0];
            rng[1] = sp[1];
            if (si & 1) continue;
            int li = si / 2;
            int out = li < c->nn_n_hidden ? c->nn_hidden[li] : 1;
            real* W = theta + (int64_t)m * P + off + (int64_t)j * in * out;
            int64_t nw = (int64_t)in * out;
            if (c->nn_bias) {
              uint32_t kk[4];
              orc_split(sp + 2, 2, L, kk);
              for (int64_t i = 0; i < nw; ++i) W[i] = (real)(normal_from_bits(bits_at(kk, nw, i, L)) * (float)c->nn_sig_param);
              real* B = theta + (int64_t)m * P + off + (int64_t)d * in * out + (int64_t)j * out;
              for (int i = 0; i < out; ++i) B[i] = (real)(normal_from_bits(bits_at(kk + 2, out, i, L)) * (float)c->nn_sig_param);
              off += (int64_t)d * in * out + (int64_t)d * out;
            } else {
              for (int64_t i = 0; i < nw; ++i) W[i] = (real)(normal_from_bits(bits_at(sp + 2, nw, i, L)) * (float)c->nn_sig_param);
              off += (int64_t)d * in * out;
            }
            in = out;
          }
        }
    } else {
      return 1;
    }
  }
  return 0;
}

/* ------------------------------------------------------------------------------------------------
 * BGe precomputation: R_j and N_j per node (they do not depend on the graph)   linearGaussian.py:78-94
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
  int n_mats;     /* 1 (no interventions) or d */
  real* R;        /* [n_mats, d, d] */
  double* Nj;     /* [d] */
  double* gam;    /* [d, d+1] log_gamma_term(j, l) for integer l */
  double alpha_lambd, small_t;
} bge_pre;

static void bge_prepare(const dibs_config* c, const real* x, const int32_t* mask, const real* mean_obs, bge_pre* b) {
  const int d = c->n_vars, N = c->n_observations;
  b->alpha_lambd = c->bge_alpha_lambd > 0 ? c->bge_alpha_lambd : d + 2.0;
  const double amu = c->bge_alpha_mu;
  b->small_t = amu * (b->alpha_lambd - d - 1) / (amu + 1);
  int any = 0;
  if (mask)
    for (int64_t i = 0; i < (int64_t)N * d; ++i) any |= mask[i] != 0;
  b->n_mats = any ? d : 1;
  b->R = (real*)malloc(sizeof(real) * (size_t)b->n_mats * d * d);
  b->Nj = (double*)malloc(sizeof(double) * d);
  b->gam = (double*)malloc(sizeof(double) * d * (d + 1));
  double* xb = (double*)malloc(sizeof(double) * d);
  double* Rm = (double*)malloc(sizeof(double) * d * d);
  for (int jm = 0; jm < (any ? d : 1); ++jm) {
    double Nn = 0;
    for (int n = 0; n < N; ++n) Nn += (any && mask[(int64_t)n * d + jm]) ? 0.0 : 1.0;
    for (int a = 0; a < d; ++a) {
      double s = 0;
      for (int n = 0; n < N; ++n)
        if (!(any && mask[(int64_t)n * d + jm])) s += (double)x[(int64_t)n * d + a];
      xb[a] = Nn > 0 ? s / Nn : 0.0;
    }
    for (int a = 0; a < d; ++a)
      for (int bb = 0; bb < d; ++bb) {
        double s = 0;
        for (int n = 0; n < N; ++n)
          if (!(any && mask[(int64_t)n * d + jm]))
            s += ((double)x[(int64_t)n * d + a] - xb[a]) * ((double)x[(int64_t)n * d + bb] - xb[bb]);
        double ma = mean_obs ? (double)mean_obs[a] : 0.0, mb = mean_obs ? (double)mean_obs[bb] : 0.0;
        Rm[a * d + bb] = (a == bb ? b->small_t : 0.0) + s + (Nn * amu / (Nn + amu)) * (xb[a] - ma) * (xb[bb] - mb);
      }
    for (int i = 0; i < d * d; ++i) b->R[(size_t)jm * d * d + i] = (real)Rm[i];
    if (any) b->Nj[jm] = Nn; else for (int j = 0; j < d; ++j) b->Nj[j] = Nn;
  }
  for (int j = 0; j < d; ++j)
    for (int l = 0; l <= d; ++l) {
      double Nn = b->Nj[j], al = b->alpha_lambd;
      b->gam[j * (d + 1) + l] = 0.5 * (log(amu) - log(Nn + amu)) + lgamma(0.5 * (Nn + al - d + l + 1)) -
                                lgamma(0.5 * (al - d + l + 1)) - 0.5 * Nn * log(M_PI) +
                                0.5 * (al - d + 2 * l + 1) * log(b->small_t);
    }
  free(xb);
  free(Rm);
}
static void bge_free(bge_pre* b) { free(b->R); free(b->Nj); free(b->gam); }

/* log|det| of a dense n x n matrix by LU with partial pivoting (what jnp.linalg.slogdet does) -- destroys a */
static real lu_logabsdet(real* a, int n) {
  real ld = 0;
  for (int k = 0; k < n; ++k) {
    int p = k;
    real best = fabs((double)a[k * n + k]);
    for (int i = k + 1; i < n; ++i)
      if (fabs((double)a[i * n + k]) > best) { best = (real)fabs((double)a[i * n + k]); p = i; }
    if (p != k)
      for (int j = 0; j < n; ++j) { real t = a[k * n + j]; a[k * n + j] = a[p * n + j]; a[p * n + j] = t; }
    real piv = a[k * n + k];
    ld += R_LOG((real)fabs((double)piv));
    real inv = (real)1 / piv;
    for (int i = k + 1; i < n; ++i) {
      real f = a[i * n + k] * inv;
      if (f != 0)
        for (int j = k + 1; j < n; ++j) a[i * n + j] -= f * a[k * n + j];
    }
  }
  return ld;
}

/* BGe node score for a HARD parent set.  mode 0: the reference's masked d x d slogdet (func.py:128-145),
 * mode 1: Cholesky of the gathered principal sub-matrix with j ordered last (identical value). */
static real bge_node_score_hard(const dibs_config* c, const bge_pre* b, int j, const uint8_t* g /* [d,d] */,
                                int mode, real* work /* >= 2*d*d */) {
  const int d = c->n_vars;
  const real* R = b->R + (b->n_mats > 1 ? (size_t)j * d * d : 0);
  const double Nn = b->Nj[j];
  if (Nn == 0) return 0;
  int idx[256];
  int l = 0;
  for (int i = 0; i < d; ++i)
    if (g[i * d + j]) idx[l++] = i;
  real ld_pa = 0, ld_all = 0;
  if (mode == 0) {
    real* a = work;
    for (int pass = 0; pass < 2; ++pass) {
      for (int r = 0; r < d; ++r)
        for (int q = 0; q < d; ++q) {
          int mr = g[r * d + j] || (pass && r == j), mq = g[q * d + j] || (pass && q == j);
          a[r * d + q] = (mr && mq) ? R[r * d + q] : (r == q ? (real)1 : (real)0);
        }
      real v = lu_logabsdet(a, d);
      if (pass == 0) ld_pa = v; else ld_all = v;
    }
  } else {
    idx[l] = j;
    const int n = l + 1;
    real* a = work;
    for (int r = 0; r < n; ++r)
      for (int q = 0; q <= r; ++q) a[r * n + q] = R[idx[r] * d + idx[q]];
    real ld = 0;
    ld_pa = 0;
    for (int kk = 0; kk < n; ++kk) {
      real s = a[kk * n + kk];
      for (int p = 0; p < kk; ++p) s -= a[kk * n + p] * a[kk * n + p];
      real lkk = R_SQRT(s);
      a[kk * n + kk] = lkk;
      if (kk == n - 1) ld_pa = ld;
      ld += 2 * R_LOG(lkk);
      for (int r = kk + 1; r < n; ++r) {
        real t = a[r * n + kk];
        for (int p = 0; p < kk; ++p) t -= a[r * n + p] * a[kk * n + p];
        a[r * n + kk] = t / lkk;
      }
    }
    ld_all = ld;
  }
  const double al = b->alpha_lambd;
  return (real)(b->gam[j * (d + 1) + l] + 0.5 * (Nn + al - d + l) * (double)ld_pa -
                0.5 * (Nn + al - d + l + 1) * (double)ld_all);
}

/* ------------------------------------------------------------------------------------------------
 * LinearGaussian log joint and its gradients (hard or soft g)   linearGaussian.py:278-338
 *   r = (1-mask) o (x - x (g o theta)) / obs_noise
 *   d/dg = logN(theta; mu_e, sig_e) + theta o (x^T r),  d/dtheta = -g o (theta - mu_e)/sig_e^2 + g o (x^T r)
 * ---------------------------------------------------------------------------------------------- */
static real lingauss_eval(const dibs_config* c, const real* x, const int32_t* mask, const real* g, const real* theta,
                          real* dg, real* dth, real* work /* >= N*d + d*d */) {
  const int d = c->n_vars, N = c->n_observations;
  const double on = c->lin_obs_noise, mu = c->lin_mean_edge, sg = c->lin_sig_edge;
  real* res = work;          /* [N, d] */
  real* xtr = work + (size_t)N * d; /* [d, d] */
  double lp = 0;
  const double lognorm_th = -log(sg) - 0.5 * log(2 * M_PI);
  const double lognorm_x = -0.5 * log(on) - 0.5 * log(2 * M_PI);
  for (int i = 0; i < d * d; ++i) {
    double zt = ((double)theta[i] - mu) / sg;
    lp += (double)g[i] * (-0.5 * zt * zt + lognorm_th);
  }
  for (int n = 0; n < N; ++n)
    for (int j = 0; j < d; ++j) {
      double m = 0;
      for (int i = 0; i < d; ++i) m += (double)x[(size_t)n * d + i] * (double)g[i * d + j] * (double)theta[i * d + j];
      double e = (double)x[(size_t)n * d + j] - m;
      int mk = mask && mask[(size_t)n * d + j];
      if (!mk) lp += -0.5 * e * e / on + lognorm_x;
      res[(size_t)n * d + j] = mk ? (real)0 : (real)(e / on);
    }
  if (dg || dth) {
    for (int i = 0; i < d; ++i)
      for (int j = 0; j < d; ++j) {
        double s = 0;
        for (int n = 0; n < N; ++n) s += (double)x[(size_t)n * d + i] * (double)res[(size_t)n * d + j];
        xtr[i * d + j] = (real)s;
      }
    for (int i = 0; i < d * d; ++i) {
      double zt = ((double)theta[i] - mu) / sg;
      if (dg) dg[i] = (real)((-0.5 * zt * zt + lognorm_th) + (double)theta[i] * (double)xtr[i]);
      if (dth) dth[i] = (real)((double)g[i] * (-zt / sg) + (double)g[i] * (double)xtr[i]);
    }
  }
  return (real)lp;
}

/* ------------------------------------------------------------------------------------------------
 * DenseNonlinearGaussian log joint + gradients   nonlinearGaussian.py:248-326
 * theta layout per particle: for each layer l: W_l [d, in_l, out_l] then (bias) b_l [d, out_l]
 * ---------------------------------------------------------------------------------------------- */
static inline double act_f(int a, double v) {
  switch (a) {
    case DIBS_ACT_RELU: return v > 0 ? v : 0;
    case DIBS_ACT_TANH: return tanh(v);
    case DIBS_ACT_SIGMOID: return 1.0 / (1.0 + exp(-v));
    default: return v > 0 ? v : 0.01 * v;
  }
}
static inline double act_df(int a, double v, double fv) {
  switch (a) {
    case DIBS_ACT_RELU: return v > 0 ? 1 : 0;
    case DIBS_ACT_TANH: return 1 - fv * fv;
    case DIBS_ACT_SIGMOID: return fv * (1 - fv);
    default: return v > 0 ? 1 : 0.01;
  }
}

static real densenn_eval(const dibs_config* c, const real* x, const int32_t* mask, const real* g, const real* theta,
                         real* dg, real* dth) {
  const int d = c->n_vars, N = c->n_observations, nl = c->nn_n_hidden + 1;
  const double on = c->nn_obs_noise, sp = c->nn_sig_param;
  const double lognorm_p = -log(sp) - 0.5 * log(2 * M_PI);
  const double lognorm_x = -0.5 * log(on) - 0.5 * log(2 * M_PI);
  int sizes[DIBS_MAX_HIDDEN_LAYERS + 2];
  int64_t woff[DIBS_MAX_HIDDEN_LAYERS + 1], boff[DIBS_MAX_HIDDEN_LAYERS + 1];
  sizes[0] = d;
  int64_t off = 0;
  int maxw = d;
  for (int l = 0; l < nl; ++l) {
    sizes[l + 1] = l < c->nn_n_hidden ? c->nn_hidden[l] : 1;
    if (sizes[l + 1] > maxw) maxw = sizes[l + 1];
    woff[l] = off;
    off += (int64_t)d * sizes[l] * sizes[l + 1];
    boff[l] = off;
    if (c->nn_bias) off += (int64_t)d * sizes[l + 1];
  }
  const int64_t P = off;
  double lp = 0;
  /* prior: all leaves N(0, sig_param); first-layer weights masked by g.T[:, :, None]  (:260-272) */
  for (int l = 0; l < nl; ++l) {
    for (int j = 0; j < d; ++j)
      for (int a = 0; a < sizes[l]; ++a)
        for (int o = 0; o < sizes[l + 1]; ++o) {
          int64_t ix = woff[l] + ((int64_t)j * sizes[l] + a) * sizes[l + 1] + o;
          double w = (double)theta[ix], lw = -0.5 * (w / sp) * (w / sp) + lognorm_p;
          double gm = l == 0 ? (double)g[a * d + j] : 1.0;
          lp += gm * lw;
          if (dth) dth[ix] = (real)(gm * (-w / (sp * sp)));
          if (dg && l == 0) dg[a * d + j] += (real)lw;
        }
    if (c->nn_bias)
      for (int64_t i = 0; i < (int64_t)d * sizes[l + 1]; ++i) {
        double w = (double)theta[boff[l] + i];
        lp += -0.5 * (w / sp) * (w / sp) + lognorm_p;
        if (dth) dth[boff[l] + i] = (real)(-w / (sp * sp));
      }
  }
  (void)P;
  double* pre = (double*)malloc(sizeof(double) * (size_t)(nl + 1) * maxw * 2);
  double* actv = pre + (size_t)(nl + 1) * maxw;
  double* delta = (double*)malloc(sizeof(double) * (size_t)maxw * 2);
  for (int j = 0; j < d; ++j)
    for (int n = 0; n < N; ++n) {
      if (mask && mask[(size_t)n * d + j]) continue;
      /* forward for node j on row n: input x[n, :] o g[:, j]   (:291) */
      for (int a = 0; a < d; ++a) actv[a] = (double)x[(size_t)n * d + a] * (double)g[a * d + j];
      for (int l = 0; l < nl; ++l) {
        const int in = sizes[l], out = sizes[l + 1];
        for (int o = 0; o < out; ++o) {
          double s = c->nn_bias ? (double)theta[boff[l] + (int64_t)j * out + o] : 0.0;
          for (int a = 0; a < in; ++a)
            s += actv[(size_t)l * maxw + a] * (double)theta[woff[l] + ((int64_t)j * in + a) * out + o];
          pre[(size_t)(l + 1) * maxw + o] = s;
          actv[(size_t)(l + 1) * maxw + o] = l < nl - 1 ? act_f(c->nn_activation, s) : s;
        }
      }
      double mean = actv[(size_t)nl * maxw];
      double e = (double)x[(size_t)n * d + j] - mean;
      lp += -0.5 * e * e / on + lognorm_x;
      if (!(dg || dth)) continue;
      /* backward */
      double* dl = delta;
      double* dl2 = delta + maxw;
      dl[0] = e / on; /* d lp / d mean */
      for (int l = nl - 1; l >= 0; --l) {
        const int in = sizes[l], out = sizes[l + 1];
        for (int a = 0; a < in; ++a) dl2[a] = 0;
        for (int o = 0; o < out; ++o) {
          double dpre = dl[o];
          if (l < nl - 1) dpre *= act_df(c->nn_activation, pre[(size_t)(l + 1) * maxw + o], actv[(size_t)(l + 1) * maxw + o]);
          if (dth && c->nn_bias) dth[boff[l] + (int64_t)j * out + o] += (real)dpre;
          for (int a = 0; a < in; ++a) {
            int64_t ix = woff[l] + ((int64_t)j * in + a) * out + o;
            if (dth) dth[ix] += (real)(dpre * actv[(size_t)l * maxw + a]);
            dl2[a] += dpre * (double)theta[ix];
          }
        }
        double* t = dl; dl = dl2; dl2 = t;
      }
      if (dg)
        for (int a = 0; a < d; ++a) dg[a * d + j] += (real)(dl[a] * (double)x[(size_t)n * d + a]);
    }
  free(pre);
  free(delta);
  return (real)lp;
}

/* ------------------------------------------------------------------------------------------------
 * the step
 * ---------------------------------------------------------------------------------------------- */
typedef struct orc_debug {
  real* scores;     /* [M, d, d] */
  real* logprobs_z; /* [M, S] */
  real* logprobs_th;/* [M, S] */
  real* w_lik;      /* [M, d, d] */
  real* w_acyc;     /* [M, d, d] */
  real* grad_z;     /* [M, D] */
  real* grad_theta; /* [M, P] */
  real* kxx;        /* [M, M] */
  real* phi_z;      /* [M, D] */
  real* phi_theta;  /* [M, P] */
  real* node_scores;/* [M, S, d]  (BGe) */
  uint8_t* g_samples;/* [M, S, d, d] hard graphs of the Z (score) estimator */
} orc_debug;

static void matmul_dd(const real* a, const real* b, real* c, int d) {
  for (int i = 0; i < d; ++i)
    for (int j = 0; j < d; ++j) c[i * d + j] = 0;
  for (int i = 0; i < d; ++i)
    for (int k = 0; k < d; ++k) {
      real aik = a[i * d + k];
      for (int j = 0; j < d; ++j) c[i * d + j] += aik * b[k * d + j];
    }
}

/* result = m^n by binary powering (jnp.linalg.matrix_power order)  graph_utils.py:26 */
static void matpow(const real* m, int n, real* result, real* zb, real* tmp, int d) {
  int have_z = 0, have_r = 0;
  while (n > 0) {
    if (!have_z) { memcpy(zb, m, sizeof(real) * d * d); have_z = 1; }
    else { matmul_dd(zb, zb, tmp, d); memcpy(zb, tmp, sizeof(real) * d * d); }
    int bit = n & 1;
    n >>= 1;
    if (bit) {
      if (!have_r) { memcpy(result, zb, sizeof(real) * d * d); have_r = 1; }
      else { matmul_dd(result, zb, tmp, d); memcpy(result, tmp, sizeof(real) * d * d); }
    }
  }
  if (!have_r) for (int i = 0; i < d; ++i) for (int j = 0; j < d; ++j) result[i * d + j] = i == j;
}

static inline double sigmoid_d(double v) { return 1.0 / (1.0 + exp(-v)); }

/* One SVGD step in two phases so that particles can be sharded over ranks:
 *   phase A (local)  : estimators for the particles [m0, m0 + Mloc) of this rank -> packed rows
 *                      [z | grad_z | theta | grad_theta] (row stride E = ceil4(2D + 2P)), key / baseline advance
 *   phase B (update) : kernel rows, phi and optimizer step for the local particles from ALL packed rows.
 * z, vz, theta, vtheta, baseline are the LOCAL shards; PRNG rows are indexed by the GLOBAL particle id. */
static int64_t pack_stride(const dibs_config* c) {
  const int64_t D = (int64_t)c->n_vars * c->n_dim * 2, P = orc_theta_size(c);
  return (2 * D + 2 * P + 3) & ~(int64_t)3;
}
ORC_EXPORT int64_t orc_pack_stride(const dibs_config* c) { return pack_stride(c); }

ORC_EXPORT int orc_step_local(const dibs_config* c, const real* x, const int32_t* mask, const real* mean_obs, real* z,
                              real* theta, uint32_t key[2], real* baseline, int t, real* pack_local, orc_debug* dbg,
                              int bge_mode, int n_threads) {
  const int d = c->n_vars, k = c->n_dim, Mg = c->n_particles, S = c->n_grad_mc_samples, Sa = c->n_acyclicity_mc_samples;
  const int M = Mg / c->n_ranks, m0 = c->rank * M; /* M = local particle count from here on */
  const int L = c->rng_layout;
  const int64_t D = (int64_t)d * k * 2, dd = (int64_t)d * d, P = orc_theta_size(c), E = pack_stride(c);
  const real alpha = (real)(c->alpha_linear * t), beta = (real)(c->beta_linear * t), tau = (real)c->tau;
  const double sigz = latent_std(c);
  if (d > 256) return 2;
#ifdef _OPENMP
  if (n_threads > 0) omp_set_num_threads(n_threads);
#endif
  bge_pre bp;
  memset(&bp, 0, sizeof bp);
  if (c->likelihood == DIBS_LIK_BGE) bge_prepare(c, x, mask, mean_obs, &bp);

  real* scores = (real*)malloc(sizeof(real) * M * dd);
  real* wlik = (real*)calloc((size_t)M * dd, sizeof(real));
  real* wacyc = (real*)calloc((size_t)M * dd, sizeof(real));
  real* gradz = (real*)malloc(sizeof(real) * M * D);
  real* gradth = P ? (real*)calloc((size_t)M * P, sizeof(real)) : NULL;
  real* lpz = (real*)malloc(sizeof(real) * (size_t)M * S);
  real* lpth = (real*)malloc(sizeof(real) * (size_t)M * S);
  real* newb = (real*)malloc(sizeof(real) * M);

  /* scores = U V^T  (dibs.py:179-180); f32 build: k-ordered fmaf chain (what an f32 MFMA computes) */
  for (int m = 0; m < M; ++m)
    for (int i = 0; i < d; ++i)
      for (int j = 0; j < d; ++j) {
        real acc = 0;
        for (int q = 0; q < k; ++q)
          acc = R_FMA(z[(m * (int64_t)d + i) * k * 2 + q * 2], z[(m * (int64_t)d + j) * k * 2 + q * 2 + 1], acc);
        scores[m * dd + i * d + j] = acc;
      }

  uint32_t carry[2] = {key[0], key[1]};
  uint32_t* pk = (uint32_t*)malloc(sizeof(uint32_t) * 2 * (Mg + 1));

  /* ---- theta estimator (joint only; FIRST key batch, svgd.py:695-696) ---- */
  if (c->joint) {
    orc_split(carry, Mg + 1, L, pk);
    carry[0] = pk[0];
    carry[1] = pk[1];
    int err = 0;
#pragma omp parallel for schedule(dynamic)
    for (int m = 0; m < M; ++m) {
      const uint32_t* km = pk + 2 * (1 + m0 + m);
      real* g = (real*)malloc(sizeof(real) * dd);
      real* dth = (real*)malloc(sizeof(real) * P);
      real* acc = (real*)calloc(P, sizeof(real));
      real* work = (real*)malloc(sizeof(real) * ((size_t)c->n_observations * d + dd));
      float* pf = (float*)malloc(sizeof(float) * dd);
      real* lps = (real*)malloc(sizeof(real) * S);
      for (int i = 0; i < dd; ++i) pf[i] = (float)sigmoid_d((double)(alpha * scores[m * dd + i]));
      /* pass 1: log probs; pass 2: weighted gradient with w = softmax (== the signed-LSE ratio of dibs.py:531-549) */
      const int64_t n = (int64_t)S * dd;
      for (int pass = 0; pass < 2; ++pass) {
        double mx = -INFINITY, den = 0;
        if (pass == 1) {
          for (int s = 0; s < S; ++s) mx = lps[s] > mx ? lps[s] : mx;
          for (int s = 0; s < S; ++s) den += exp((double)lps[s] - mx);
        }
        for (int s = 0; s < S; ++s) {
          for (int i = 0; i < d; ++i)
            for (int j = 0; j < d; ++j) {
              float u = uniform_from_bits(bits_at(km, n, s * dd + i * d + j, L), 0.0f, 1.0f); /* particle key itself: dibs.py:510 */
              g[i * d + j] = (i != j && u < pf[i * d + j]) ? (real)1 : (real)0;
            }
          if (pass == 0) {
            if (c->likelihood == DIBS_LIK_LINGAUSS) lps[s] = lingauss_eval(c, x, mask, g, theta + m * P, NULL, NULL, work);
            else if (c->likelihood == DIBS_LIK_DENSENN) lps[s] = densenn_eval(c, x, mask, g, theta + m * P, NULL, NULL);
            else err = 1;
          } else {
            double w = exp((double)lps[s] - mx) / den;
            if (w == 0) continue;
            if (c->likelihood == DIBS_LIK_LINGAUSS) lingauss_eval(c, x, mask, g, theta + m * P, NULL, dth, work);
            else { memset(dth, 0, sizeof(real) * P); densenn_eval(c, x, mask, g, theta + m * P, NULL, dth); }
            for (int64_t i = 0; i < P; ++i) acc[i] += (real)(w * (double)dth[i]);
          }
        }
      }
      memcpy(gradth + m * P, acc, sizeof(real) * P);
      memcpy(lpth + (size_t)m * S, lps, sizeof(real) * S);
      free(g); free(dth); free(acc); free(work); free(pf); free(lps);
    }
    if (err) return 3;
  }

  /* ---- Z likelihood estimator ---- */
  orc_split(carry, Mg + 1, L, pk);
  carry[0] = pk[0];
  carry[1] = pk[1];
  if (c->grad_estimator_z != DIBS_EST_SCORE && c->grad_estimator_z != DIBS_EST_REPARAM) return 4;
  if (c->grad_estimator_z == DIBS_EST_REPARAM && c->likelihood == DIBS_LIK_BGE) return 5; /* TODO soft BGe */
#pragma omp parallel for schedule(dynamic)
  for (int m = 0; m < M; ++m) {
    uint32_t sp[4];
    orc_split(pk + 2 * (1 + m0 + m), 2, L, sp); /* subk, subk_ = split(subk)  dibs.py:350 / :430 */
    const uint32_t* kg = sp + 2;
    real* g = (real*)malloc(sizeof(real) * dd);
    uint8_t* gh = (uint8_t*)malloc(dd);
    real* dg = (real*)malloc(sizeof(real) * dd);
    real* acc = (real*)calloc(dd, sizeof(real));
    real* work = (real*)malloc(sizeof(real) * ((size_t)c->n_observations * d + 2 * dd + 16));
    float* pf = (float*)malloc(sizeof(float) * dd);
    real* lps = (real*)malloc(sizeof(real) * S);
    real* P_ = (real*)malloc(sizeof(real) * dd);
    for (int i = 0; i < dd; ++i) {
      double pv = sigmoid_d((double)(alpha * scores[m * dd + i]));
      pf[i] = (float)pv;
      P_[i] = (real)pv;
    }
    const int64_t n = (int64_t)S * dd;
    for (int pass = 0; pass < 2; ++pass) {
      double mx = -INFINITY, den = 0;
      if (pass == 1) {
        for (int s = 0; s < S; ++s) mx = lps[s] > mx ? lps[s] : mx;
        for (int s = 0; s < S; ++s) den += exp((double)lps[s] - mx);
      }
      for (int s = 0; s < S; ++s) {
        if (c->grad_estimator_z == DIBS_EST_SCORE) {
          for (int i = 0; i < d; ++i)
            for (int j = 0; j < d; ++j) {
              float u = uniform_from_bits(bits_at(kg, n, s * dd + i * d + j, L), 0.0f, 1.0f);
              gh[i * d + j] = (i != j && u < pf[i * d + j]) ? 1 : 0;
              g[i * d + j] = gh[i * d + j];
            }
          if (pass == 0) {
            if (dbg && dbg->g_samples) memcpy(dbg->g_samples + ((size_t)m * S + s) * dd, gh, dd);
            if (c->likelihood == DIBS_LIK_BGE) {
              double tot = 0;
              for (int j = 0; j < d; ++j) {
                real ns = bge_node_score_hard(c, &bp, j, gh, bge_mode, work);
                if (dbg && dbg->node_scores) dbg->node_scores[((size_t)m * S + s) * d + j] = ns;
                tot += (double)ns;
              }
              lps[s] = (real)tot;
            } else if (c->likelihood == DIBS_LIK_LINGAUSS) lps[s] = lingauss_eval(c, x, mask, g, theta + m * P, NULL, NULL, work);
            else lps[s] = densenn_eval(c, x, mask, g, theta + m * P, NULL, NULL);
          } else {
            double w = exp((double)lps[s] - mx) / den;
            for (int i = 0; i < dd; ++i) acc[i] += (real)(w * (double)g[i]);
          }
        } else { /* reparam: soft graph  dibs.py:121-140, 271-288 */
          for (int i = 0; i < d; ++i)
            for (int j = 0; j < d; ++j) {
              float e = logistic_from_bits(bits_at(kg, n, s * dd + i * d + j, L), c->logistic_minval_tiny);
              g[i * d + j] = i == j ? (real)0 : (real)sigmoid_d((double)(tau * ((real)e + alpha * scores[m * dd + i * d + j])));
            }
          if (pass == 0) {
            if (c->likelihood == DIBS_LIK_LINGAUSS) lps[s] = lingauss_eval(c, x, mask, g, theta + m * P, NULL, NULL, work);
            else lps[s] = densenn_eval(c, x, mask, g, theta + m * P, NULL, NULL);
          } else {
            double w = exp((double)lps[s] - mx) / den;
            if (w == 0) continue;
            if (c->likelihood == DIBS_LIK_LINGAUSS) lingauss_eval(c, x, mask, g, theta + m * P, dg, NULL, work);
            else { memset(dg, 0, sizeof(real) * dd); densenn_eval(c, x, mask, g, theta + m * P, dg, NULL); }
            for (int i = 0; i < d; ++i)
              for (int j = 0; j < d; ++j)
                if (i != j) acc[i * d + j] += (real)(w * (double)dg[i * d + j] * (double)(tau * alpha) * (double)g[i * d + j] * (1.0 - (double)g[i * d + j]));
          }
        }
      }
    }
    double bsum = 0;
    for (int s = 0; s < S; ++s) bsum += (double)lps[s];
    if (c->grad_estimator_z == DIBS_EST_SCORE) {
      /* W = alpha (sum_s w_s G_s - P) offdiag; baseline c>0 multiplies by exp(-b)  dibs.py:363-367, 376-382 */
      double scale = c->score_function_baseline > 0 ? exp(-(double)baseline[m]) : 1.0;
      for (int i = 0; i < d; ++i)
        for (int j = 0; j < d; ++j)
          wlik[m * dd + i * d + j] = i == j ? (real)0 : (real)(scale * (double)alpha * ((double)acc[i * d + j] - (double)P_[i * d + j]));
      newb[m] = (real)(c->score_function_baseline * (bsum / S) + (1 - c->score_function_baseline) * (double)baseline[m]);
    } else {
      memcpy(wlik + m * dd, acc, sizeof(real) * dd);
      newb[m] = baseline[m];
    }
    memcpy(lpz + (size_t)m * S, lps, sizeof(real) * S);
    free(g); free(gh); free(dg); free(acc); free(work); free(pf); free(lps); free(P_);
  }

  /* ---- latent prior: acyclicity + graph prior  (dibs.py:557-658) ---- */
  orc_split(carry, Mg + 1, L, pk);
  carry[0] = pk[0];
  carry[1] = pk[1];
  double er_c = 0;
  if (c->graph_prior == DIBS_PRIOR_ER) {
    double p = c->graph_prior_edges_per_node * d / ((d * (d - 1)) / 2.0);
    er_c = log(p) - log(1 - p);
  }
#pragma omp parallel for schedule(dynamic)
  for (int m = 0; m < M; ++m) {
    const uint32_t* km = pk + 2 * (1 + m0 + m); /* particle key itself: dibs.py:595 */
    real* gs = (real*)malloc(sizeof(real) * dd);
    real* mm = (real*)malloc(sizeof(real) * dd);
    real* pw = (real*)malloc(sizeof(real) * dd);
    real* zb = (real*)malloc(sizeof(real) * dd);
    real* tmp = (real*)malloc(sizeof(real) * dd);
    real* acc = (real*)calloc(dd, sizeof(real));
    const int64_t n = (int64_t)Sa * dd;
    for (int s = 0; s < Sa; ++s) {
      for (int i = 0; i < d; ++i)
        for (int j = 0; j < d; ++j) {
          float e = logistic_from_bits(bits_at(km, n, s * dd + i * d + j, L), c->logistic_minval_tiny);
          real gv = i == j ? (real)0 : (real)sigmoid_d((double)(tau * ((real)e + alpha * scores[m * dd + i * d + j])));
          gs[i * d + j] = gv;
          mm[i * d + j] = (i == j ? (real)1 : (real)0) + gv / (real)d;
        }
      matpow(mm, d - 1, pw, zb, tmp, d);
      /* dh/dG = (M^{d-1})^T ; chain through G~ = sigmoid(tau (eps + alpha s)) */
      for (int i = 0; i < d; ++i)
        for (int j = 0; j < d; ++j)
          if (i != j) acc[i * d + j] += pw[j * d + i] * tau * alpha * gs[i * d + j] * ((real)1 - gs[i * d + j]);
    }
    for (int i = 0; i < dd; ++i) wacyc[m * dd + i] = acc[i] / (real)Sa;
    free(gs); free(mm); free(pw); free(zb); free(tmp); free(acc);
  }

  /* ---- grad_z = [W V, W^T U] - z/sigma^2 with W = W_lik - beta W_acyc + W_prior ---- */
#pragma omp parallel for schedule(dynamic)
  for (int m = 0; m < M; ++m) {
    real* W = (real*)malloc(sizeof(real) * dd);
    real* Pm = (real*)malloc(sizeof(real) * dd);
    real* colsum = (real*)calloc(d, sizeof(real));
    for (int i = 0; i < d; ++i)
      for (int j = 0; j < d; ++j) {
        Pm[i * d + j] = i == j ? (real)0 : (real)sigmoid_d((double)(alpha * scores[m * dd + i * d + j]));
        colsum[j] += Pm[i * d + j];
      }
    for (int i = 0; i < d; ++i)
      for (int j = 0; j < d; ++j) {
        real pr = 0;
        if (i != j) {
          real dp = alpha * Pm[i * d + j] * ((real)1 - Pm[i * d + j]);
          if (c->graph_prior == DIBS_PRIOR_ER) pr = (real)er_c * dp;
          else if (c->graph_prior == DIBS_PRIOR_SF) pr = (real)-3 / ((real)1 + colsum[j]) * dp;
        }
        W[i * d + j] = wlik[m * dd + i * d + j] - beta * wacyc[m * dd + i * d + j] + pr;
      }
    const real* zm = z + m * D;
    real* gm = gradz + m * D;
    const real inv = (real)(1.0 / (sigz * sigz));
    for (int i = 0; i < d; ++i)
      for (int q = 0; q < k; ++q) {
        real su = 0, sv = 0;
        for (int j = 0; j < d; ++j) {
          su += W[i * d + j] * zm[(j * k + q) * 2 + 1]; /* dU[i,q] = sum_j W[i,j] V[j,q] */
          sv += W[j * d + i] * zm[(j * k + q) * 2];     /* dV[i,q] = sum_j W[j,i] U[j,q] */
        }
        gm[(i * k + q) * 2] = su - zm[(i * k + q) * 2] * inv;
        gm[(i * k + q) * 2 + 1] = sv - zm[(i * k + q) * 2 + 1] * inv;
      }
    free(W); free(Pm); free(colsum);
  }

  /* ---- pack rows [z | grad_z | theta | grad_theta] ---- */
  for (int m = 0; m < M; ++m) {
    real* row = pack_local + (int64_t)m * E;
    memcpy(row, z + m * D, sizeof(real) * D);
    memcpy(row + D, gradz + m * D, sizeof(real) * D);
    if (P) {
      memcpy(row + 2 * D, theta + m * P, sizeof(real) * P);
      memcpy(row + 2 * D + P, gradth + m * P, sizeof(real) * P);
    }
    for (int64_t i = 2 * D + 2 * P; i < E; ++i) row[i] = 0;
  }
  if (dbg) {
    if (dbg->scores) memcpy(dbg->scores, scores, sizeof(real) * M * dd);
    if (dbg->logprobs_z) memcpy(dbg->logprobs_z, lpz, sizeof(real) * (size_t)M * S);
    if (dbg->logprobs_th && c->joint) memcpy(dbg->logprobs_th, lpth, sizeof(real) * (size_t)M * S);
    if (dbg->w_lik) memcpy(dbg->w_lik, wlik, sizeof(real) * M * dd);
    if (dbg->w_acyc) memcpy(dbg->w_acyc, wacyc, sizeof(real) * M * dd);
    if (dbg->grad_z) memcpy(dbg->grad_z, gradz, sizeof(real) * M * D);
    if (dbg->grad_theta && P) memcpy(dbg->grad_theta, gradth, sizeof(real) * M * P);
  }
  for (int m = 0; m < M; ++m) baseline[m] = newb[m];
  key[0] = carry[0];
  key[1] = carry[1];
  free(scores); free(wlik); free(wacyc); free(gradz); free(gradth); free(lpz); free(lpth); free(newb); free(pk);
  if (c->likelihood == DIBS_LIK_BGE) bge_free(&bp);
  return 0;
}

ORC_EXPORT int orc_step_update(const dibs_config* c, const real* pack_all, real* z, real* vz, real* theta, real* vtheta,
                               orc_debug* dbg, int n_threads) {
  const int d = c->n_vars, k = c->n_dim, Mg = c->n_particles;
  const int M = Mg / c->n_ranks, m0 = c->rank * M;
  const int64_t D = (int64_t)d * k * 2, P = orc_theta_size(c), E = pack_stride(c);
#ifdef _OPENMP
  if (n_threads > 0) omp_set_num_threads(n_threads);
#endif
  /* ---- kernel rows (kernel.py:20-30 / 52-71): local a against all b ---- */
  real* kz = (real*)malloc(sizeof(real) * (size_t)M * Mg);
  real* kt = (real*)calloc((size_t)M * Mg, sizeof(real));
#pragma omp parallel for schedule(dynamic)
  for (int a = 0; a < M; ++a)
    for (int b = 0; b < Mg; ++b) {
      const real* ra = pack_all + (int64_t)(m0 + a) * E;
      const real* rb = pack_all + (int64_t)b * E;
      double s = 0;
      for (int64_t i = 0; i < D; ++i) { double df = (double)ra[i] - (double)rb[i]; s += df * df; }
      kz[(size_t)a * Mg + b] = (real)(c->scale_latent * exp(-s / c->h_latent));
      if (c->joint) {
        double st = 0;
        for (int64_t i = 0; i < P; ++i) { double df = (double)ra[2 * D + i] - (double)rb[2 * D + i]; st += df * df; }
        kt[(size_t)a * Mg + b] = (real)(c->scale_theta * exp(-st / c->h_theta));
      }
    }
  /* ---- phi (svgd.py:194-224, 591-670): phi_a = -(1/M) sum_b [ k[b,a] grad_b - (2/h) k_seg[b,a] (x_b - x_a) ]
   *      (column kxx[:, a]; the kernel is symmetric so row a of the local slab is that column) ---- */
  real* phiz = (real*)malloc(sizeof(real) * M * D);
  real* phith = P ? (real*)malloc(sizeof(real) * M * P) : NULL;
#pragma omp parallel for schedule(dynamic)
  for (int a = 0; a < M; ++a) {
    const real* ra = pack_all + (int64_t)(m0 + a) * E;
    for (int64_t i = 0; i < D; ++i) {
      double s = 0;
      for (int b = 0; b < Mg; ++b) {
        const real* rb = pack_all + (int64_t)b * E;
        double kk = (double)kz[(size_t)a * Mg + b] + (double)kt[(size_t)a * Mg + b];
        s += kk * (double)rb[D + i] - (2.0 / c->h_latent) * (double)kz[(size_t)a * Mg + b] * ((double)rb[i] - (double)ra[i]);
      }
      phiz[a * D + i] = (real)(-s / Mg);
    }
    for (int64_t i = 0; i < P; ++i) {
      double s = 0;
      for (int b = 0; b < Mg; ++b) {
        const real* rb = pack_all + (int64_t)b * E;
        double kk = (double)kz[(size_t)a * Mg + b] + (double)kt[(size_t)a * Mg + b];
        s += kk * (double)rb[2 * D + P + i] - (2.0 / c->h_theta) * (double)kt[(size_t)a * Mg + b] * ((double)rb[2 * D + i] - (double)ra[2 * D + i]);
      }
      phith[a * P + i] = (real)(-s / Mg);
    }
  }
  if (dbg) {
    if (dbg->kxx) for (size_t i = 0; i < (size_t)M * Mg; ++i) dbg->kxx[i] = kz[i] + kt[i];
    if (dbg->phi_z) memcpy(dbg->phi_z, phiz, sizeof(real) * M * D);
    if (dbg->phi_theta && P) memcpy(dbg->phi_theta, phith, sizeof(real) * M * P);
  }
  /* ---- optimizer (jax.example_libraries.optimizers.rmsprop: gamma 0.9, eps 1e-8 inside the sqrt) ---- */
  for (int pass = 0; pass < 2; ++pass) {
    real* xx = pass ? theta : z;
    real* vv = pass ? vtheta : vz;
    const real* gg = pass ? phith : phiz;
    int64_t n = pass ? M * P : M * D;
    for (int64_t i = 0; i < n; ++i) {
      if (c->optimizer == DIBS_OPT_RMSPROP) {
        vv[i] = vv[i] * (real)0.9 + gg[i] * gg[i] * (real)(1.0 - 0.9);
        xx[i] = xx[i] - (real)c->stepsize * gg[i] / R_SQRT(vv[i] + (real)1e-8);
      } else {
        xx[i] = xx[i] - (real)c->stepsize * gg[i];
      }
    }
  }
  free(kz); free(kt); free(phiz); free(phith);
  return 0;
}

/* single-rank step = phase A + phase B on the same rows */
ORC_EXPORT int orc_step(const dibs_config* c, const real* x, const int32_t* mask, const real* mean_obs, real* z,
                        real* vz, real* theta, real* vtheta, uint32_t key[2], real* baseline, int t, orc_debug* dbg,
                        int bge_mode, int n_threads) {
  if (c->n_ranks != 1) return 9;
  real* pack = (real*)malloc(sizeof(real) * (size_t)c->n_particles * pack_stride(c));
  int rc = orc_step_local(c, x, mask, mean_obs, z, theta, key, baseline, t, pack, dbg, bge_mode, n_threads);
  if (!rc) rc = orc_step_update(c, pack, z, vz, theta, vtheta, dbg, n_threads);
  free(pack);
  return rc;
}

ORC_EXPORT int orc_run(const dibs_config* c, const real* x, const int32_t* mask, const real* mean_obs, real* z, real* vz,
                       real* theta, real* vtheta, uint32_t key[2], real* baseline, int t_start, int n_steps, int bge_mode,
                       int n_threads) {
  for (int t = t_start; t < t_start + n_steps; ++t) {
    int rc = orc_step(c, x, mask, mean_obs, z, vz, theta, vtheta, key, baseline, t, NULL, bge_mode, n_threads);
    if (rc) return rc;
  }
  return 0;
}

/* log p(D | G) / log p(theta, D | G) for a batch of hard graphs  (svgd.py:110-113, 370-372, 475-478, 838-841) */
ORC_EXPORT int orc_score_graphs(const dibs_config* c, const real* x, const int32_t* mask, const real* mean_obs,
                                const int32_t* g, const real* theta, int n, real* out, int bge_mode) {
  const int d = c->n_vars;
  const int64_t dd = (int64_t)d * d, P = orc_theta_size(c);
  bge_pre bp;
  if (c->likelihood == DIBS_LIK_BGE) bge_prepare(c, x, mask, mean_obs, &bp);
  real* work = (real*)malloc(sizeof(real) * ((size_t)c->n_observations * d + 2 * dd + 16));
  uint8_t* gh = (uint8_t*)malloc(dd);
  real* gr = (real*)malloc(sizeof(real) * dd);
  for (int q = 0; q < n; ++q) {
    for (int i = 0; i < dd; ++i) { gh[i] = g[q * dd + i] != 0; gr[i] = gh[i]; }
    if (c->likelihood == DIBS_LIK_BGE) {
      double tot = 0;
      for (int j = 0; j < d; ++j) tot += (double)bge_node_score_hard(c, &bp, j, gh, bge_mode, work);
      out[q] = (real)tot;
    } else if (c->likelihood == DIBS_LIK_LINGAUSS) out[q] = lingauss_eval(c, x, mask, gr, theta + q * P, NULL, NULL, work);
    else out[q] = densenn_eval(c, x, mask, gr, theta + q * P, NULL, NULL);
  }
  free(work); free(gh); free(gr);
  if (c->likelihood == DIBS_LIK_BGE) bge_free(&bp);
  return 0;
}
