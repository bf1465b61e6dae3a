"""ORACLE (test infrastructure only -- never imported by the product path).

Counter-based PRNG restatement: Threefry-2x32 (20 rounds) and the JAX-style
``PRNGKey / split / bits / uniform / normal / bernoulli / logistic`` layer the
reference calls at

  * dibs/inference/svgd.py:145-146, 245, 251, 294, 509-513, 695-703, 751   (split / normal)
  * dibs/inference/dibs.py:115 (bernoulli), 350, 358, 430-431, 436, 517 (split), 431, 595 (logistic)
  * dibs/models/linearGaussian.py:225 (normal)

The arithmetic lives in a third-party dependency that is NOT under /root/reference
(``jax`` / ``jaxlib``, pinned only as ``jax>=0.3.17, jaxlib>=0.3.14`` in setup.py:16-17), so this file
restates the published algorithm:

  * Threefry-2x32: Salmon et al., "Parallel random numbers: as easy as 1, 2, 3" (SC'11), 20 rounds,
    rotation constants (13,15,26,6 | 17,29,16,24), key-schedule parity 0x1BD11BDA.
  * JAX layering (jax/_src/prng.py, jax/_src/random.py): documented below per function.

Pinning: the Threefry block function is pinned by the Random123 known-answer vectors; the layering is
pinned by the values printed in JAX's public documentation for ``PRNGKey(0)`` (see tests/test_prng.py).
``legacy`` layout = ``jax_threefry_partitionable=False`` (JAX default until 0.5.0, i.e. for the whole
version range the reference was released against); ``partitionable`` is the newer default.
"""
import numpy as np

U32 = np.uint32
_ROT = ((13, 15, 26, 6), (17, 29, 16, 24))


def _rotl(x, r):
    return (x << U32(r)) | (x >> U32(32 - r))


def threefry2x32(k0, k1, c0, c1):
    """Threefry-2x32-20 block function on uint32 arrays (vectorised over counters)."""
    with np.errstate(over="ignore"):
        k0 = U32(k0)
        k1 = U32(k1)
        x0 = np.asarray(c0, dtype=U32).copy()
        x1 = np.asarray(c1, dtype=U32).copy()
        ks = (k0, k1, U32(k0 ^ k1 ^ U32(0x1BD11BDA)))
        x0 = x0 + ks[0]
        x1 = x1 + ks[1]
        for blk in range(5):
            for r in _ROT[blk % 2]:
                x0 = x0 + x1
                x1 = _rotl(x1, r)
                x1 = x1 ^ x0
            x0 = x0 + ks[(blk + 1) % 3]
            x1 = x1 + ks[(blk + 2) % 3] + U32(blk + 1)
    return x0, x1


def PRNGKey(seed):
    """jax.random.PRNGKey: [seed >> 32, seed & 0xffffffff] as uint32[2]."""
    seed = int(seed)
    return np.array([(seed >> 32) & 0xFFFFFFFF, seed & 0xFFFFFFFF], dtype=U32)


def _bits_legacy(key, n):
    """jax._src.prng.threefry_random_bits (non-partitionable): counts = iota(n) padded to even,
    split in halves (x0 = first half, x1 = second half), output = concat(y0, y1)[:n]."""
    half = (n + 1) // 2
    cnt = np.arange(2 * half, dtype=U32)
    if n % 2:
        cnt[-1] = 0  # jax pads the odd tail with a zero count
    y0, y1 = threefry2x32(key[0], key[1], cnt[:half], cnt[half:])
    return np.concatenate([y0, y1])[:n]


def _bits_partitionable(key, n):
    """threefry_random_bits_partitionable for 32-bit output: element i uses counter (hi(i), lo(i))
    and returns y0 ^ y1."""
    idx = np.arange(n, dtype=np.uint64)
    y0, y1 = threefry2x32(key[0], key[1], (idx >> np.uint64(32)).astype(U32), idx.astype(U32))
    return y0 ^ y1


def random_bits(key, n, layout="legacy"):
    return _bits_legacy(key, n) if layout == "legacy" else _bits_partitionable(key, n)


def split(key, num=2, layout="legacy"):
    """jax.random.split -> uint32[num, 2]."""
    if layout == "legacy":
        return _bits_legacy(key, 2 * num).reshape(num, 2)
    idx = np.arange(num, dtype=np.uint64)
    y0, y1 = threefry2x32(key[0], key[1], (idx >> np.uint64(32)).astype(U32), idx.astype(U32))
    return np.stack([y0, y1], axis=1)


def _unit_floats(bits):
    """(bits >> 9) | 0x3f800000 bit-cast to f32, minus 1 -> [0, 1) exactly m * 2^-23."""
    fb = (bits >> U32(9)) | U32(0x3F800000)
    return fb.view(np.float32) - np.float32(1.0)


def uniform(key, shape, minval=0.0, maxval=1.0, layout="legacy"):
    """jax.random.uniform (float32): max(minval, floats * (maxval - minval) + minval)."""
    n = int(np.prod(shape)) if len(shape) else 1
    f = _unit_floats(random_bits(key, n, layout))
    lo = np.float32(minval)
    hi = np.float32(maxval)
    out = f * np.float32(hi - lo) + lo
    return np.maximum(lo, out).astype(np.float32).reshape(shape)


def bernoulli(key, p, shape, layout="legacy"):
    """jax.random.bernoulli: uniform(key, shape) < p  (p broadcast, float32)."""
    return uniform(key, shape, layout=layout) < np.asarray(p, dtype=np.float32)


# Giles (2010) single-precision erfinv, the form XLA lowers lax.erf_inv(f32) to
_ERFINV_LT5 = [2.81022636e-08, 3.43273939e-07, -3.5233877e-06, -4.39150654e-06, 0.00021858087,
               -0.00125372503, -0.00417768164, 0.246640727, 1.50140941]
_ERFINV_GE5 = [-0.000200214257, 0.000100950558, 0.00134934322, -0.00367342844, 0.00573950773,
               -0.0076224613, 0.00943887047, 1.00167406, 2.83297682]


def erfinv_f32(x):
    x = np.asarray(x, dtype=np.float32)
    w = (-np.log1p((-x * x).astype(np.float64))).astype(np.float32)
    lt = w < np.float32(5.0)
    with np.errstate(invalid="ignore"):
        w2 = np.where(lt, w - np.float32(2.5), np.sqrt(w) - np.float32(3.0)).astype(np.float32)
    p = np.where(lt, np.float32(_ERFINV_LT5[0]), np.float32(_ERFINV_GE5[0])).astype(np.float32)
    for a, b in zip(_ERFINV_LT5[1:], _ERFINV_GE5[1:]):
        p = (np.where(lt, np.float32(a), np.float32(b)) + p * w2).astype(np.float32)
    return (p * x).astype(np.float32)


def normal(key, shape, layout="legacy"):
    """jax.random.normal (float32): sqrt(2) * erfinv(uniform(key, shape, nextafter(-1, 0), 1))."""
    lo = np.nextafter(np.float32(-1.0), np.float32(0.0), dtype=np.float32)
    u = uniform(key, shape, lo, 1.0, layout)
    return (np.float32(np.sqrt(2.0)) * erfinv_f32(u)).astype(np.float32)


def logistic(key, shape, layout="legacy", minval=None):
    """jax.random.logistic (float32): x = uniform(key, shape, finfo.eps, 1); log(x / (1 - x))."""
    lo = np.finfo(np.float32).eps if minval is None else minval
    x = uniform(key, shape, lo, 1.0, layout)
    with np.errstate(divide="ignore"):
        return np.log(x / (np.float32(1.0) - x)).astype(np.float32)
