"""Host-side counter-based PRNG with the key/stream semantics of ``jax.random`` (Threefry-2x32), so a
script written against the reference keeps its seeds: ``PRNGKey``, ``split``, ``normal``, ``uniform``,
``bernoulli``, ``logistic``, ``permutation`` (the last one is NOT stream-compatible with jax).

Only small host-side draws go through here (data factories, key bookkeeping); the per-step sampling of the
SVGD loop happens inside the HIP kernels (dibs_amd/csrc/rng.h implements the same streams on the device).
Call sites in the reference: dibs/target.py:78-104, dibs/models/graph.py:44-52, svgd.py:294."""
import numpy as np

_U32 = np.uint32
_ROT = ((13, 15, 26, 6), (17, 29, 16, 24))
LAYOUT = "legacy"  # jax_threefry_partitionable=False (the default for the jax versions the reference pins)


def _threefry(k0, k1, c0, c1):
    with np.errstate(over="ignore"):
        k0, k1 = _U32(k0), _U32(k1)
        ks = (k0, k1, _U32(k0 ^ k1 ^ _U32(0x1BD11BDA)))
        x0 = np.asarray(c0, _U32) + ks[0]
        x1 = np.asarray(c1, _U32) + ks[1]
        for blk in range(5):
            for r in _ROT[blk & 1]:
                x0 = x0 + x1
                x1 = (x1 << _U32(r)) | (x1 >> _U32(32 - r))
                x1 = x1 ^ x0
            x0 = x0 + ks[(blk + 1) % 3]
            x1 = x1 + ks[(blk + 2) % 3] + _U32(blk + 1)
    return x0, x1


def PRNGKey(seed):
    seed = int(seed)
    return np.array([(seed >> 32) & 0xFFFFFFFF, seed & 0xFFFFFFFF], dtype=_U32)


def as_key(key):
    """Accept an int seed or anything array-like holding the two uint32 words of a jax PRNGKey."""
    if isinstance(key, (int, np.integer)):
        return PRNGKey(key)
    k = np.asarray(key)
    if k.shape != (2,):
        raise ValueError("key must be an int seed or a uint32[2] array (jax.random.PRNGKey)")
    return k.astype(_U32)


def bits(key, n):
    key = as_key(key)
    if LAYOUT == "legacy":
        half = (n + 1) // 2
        cnt = np.arange(2 * half, dtype=_U32)
        if n % 2:
            cnt[-1] = 0
        y0, y1 = _threefry(key[0], key[1], cnt[:half], cnt[half:])
        return np.concatenate([y0, y1])[:n]
    idx = np.arange(n, dtype=np.uint64)
    y0, y1 = _threefry(key[0], key[1], (idx >> np.uint64(32)).astype(_U32), idx.astype(_U32))
    return y0 ^ y1


def split(key, num=2):
    key = as_key(key)
    if LAYOUT == "legacy":
        return bits(key, 2 * num).reshape(num, 2)
    idx = np.arange(num, dtype=_U32)
    y0, y1 = _threefry(key[0], key[1], np.zeros(num, _U32), idx)
    return np.stack([y0, y1], 1)


def uniform(key, shape=(), minval=0.0, maxval=1.0):
    n = int(np.prod(shape)) if len(shape) else 1
    fb = (bits(key, n) >> _U32(9)) | _U32(0x3F800000)
    f = fb.view(np.float32) - np.float32(1.0)
    lo, hi = np.float32(minval), np.float32(maxval)
    return np.maximum(lo, f * np.float32(hi - lo) + lo).astype(np.float32).reshape(shape)


def bernoulli(key, p=0.5, shape=()):
    return uniform(key, shape) < np.asarray(p, np.float32)


_A = [2.81022636e-08, 3.43273939e-07, -3.5233877e-06, -4.39150654e-06, 0.00021858087, -0.00125372503, -0.00417768164,
      0.246640727, 1.50140941]
_B = [-0.000200214257, 0.000100950558, 0.00134934322, -0.00367342844, 0.00573950773, -0.0076224613, 0.00943887047,
      1.00167406, 2.83297682]


def normal(key, shape=()):
    lo = np.nextafter(np.float32(-1.0), np.float32(0.0), dtype=np.float32)
    x = uniform(key, shape, lo, 1.0)
    w = (-np.log1p((-x * x).astype(np.float64))).astype(np.float32)
    lt = w < np.float32(5.0)
    with np.errstate(invalid="ignore"):
        w = np.where(lt, w - np.float32(2.5), np.sqrt(w) - np.float32(3.0)).astype(np.float32)
    p = np.where(lt, np.float32(_A[0]), np.float32(_B[0])).astype(np.float32)
    for a, b in zip(_A[1:], _B[1:]):
        p = (np.where(lt, np.float32(a), np.float32(b)) + p * w).astype(np.float32)
    return (np.float32(np.sqrt(2.0)) * (p * x)).astype(np.float32)


def logistic(key, shape=()):
    x = uniform(key, shape, np.finfo(np.float32).eps, 1.0)
    return np.log(x / (np.float32(1.0) - x)).astype(np.float32)


def permutation(key, n):
    """Random permutation of range(n): argsort of uniform draws (NOT jax's sort-based shuffle stream)."""
    return np.argsort(uniform(key, (int(n),)), kind="stable")


def choice(key, n, shape, replace=False):
    if replace:
        return (uniform(key, shape) * n).astype(np.int64)
    return permutation(key, n)[: int(np.prod(shape))].reshape(shape)
