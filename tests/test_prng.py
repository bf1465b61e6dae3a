"""Pins the PRNG layer of the oracle (and of the product's host-side twin) with PUBLIC known answers:
Random123's Threefry-2x32 KAT vectors and the values printed in JAX's documentation / test-suite for the
default (non-partitionable) key layout."""
import numpy as np

from oracle import prng
from dibs_amd import random as prandom


def test_threefry_random123_kat():
    # Random123 known-answer vectors (also in jax tests/random_test.py::testThreefry2x32)
    assert [hex(int(v[0])) for v in prng.threefry2x32(0, 0, [0], [0])] == ["0x6b200159", "0x99ba4efe"]
    assert [hex(int(v[0])) for v in prng.threefry2x32(0xFFFFFFFF, 0xFFFFFFFF, [0xFFFFFFFF], [0xFFFFFFFF])] == \
        ["0x1cb996fc", "0xbb002be7"]
    assert [hex(int(v[0])) for v in prng.threefry2x32(0x13198A2E, 0x03707344, [0x243F6A88], [0x85A308D3])] == \
        ["0xc4923a9c", "0x483df7a0"]


def test_jax_documented_values_legacy_layout():
    k0 = prng.PRNGKey(0)
    assert k0.tolist() == [0, 0]
    # "JAX - The Sharp Bits" / jax.random docs: split(PRNGKey(0))
    assert prng.split(k0).tolist() == [[4146024105, 967050713], [2718843009, 1272950319]]
    assert np.allclose(prng.normal(k0, (1,)), [-0.20584226], rtol=0, atol=1e-8)
    assert np.allclose(prng.normal(prng.split(k0)[1], (1,)), [-1.2515389], rtol=0, atol=2e-7)
    assert np.allclose(prng.uniform(k0, (1,)), [0.41845703], rtol=0, atol=1e-8)
    # jax docs (PRNGKey(42)) and jax tests/random_test.py::testRngRandomBits (PRNGKey(1701), odd length -> padding)
    assert prng.split(prng.PRNGKey(42)).tolist() == [[2465931498, 3679230171], [255383827, 267815257]]
    assert np.allclose(prng.normal(prng.PRNGKey(42), (3,)), [0.18693547, -1.2806505, -1.5593132], rtol=0, atol=2e-7)
    assert prng.random_bits(prng.PRNGKey(1701), 3).tolist() == [56197195, 4200222568, 961309823]


def test_c_port_matches_numpy_prng(c_oracle64):
    co = c_oracle64
    for seed in (0, 7, 2**33 + 5):
        k = prng.PRNGKey(seed)
        for n in (1, 2, 5, 8):
            assert (co.split(k, n, 0) == prng.split(k, n, "legacy")).all()
            assert (co.split(k, n, 1) == prng.split(k, n, "partitionable")).all()
        for n in (1, 7, 64, 1001):
            assert (co.random_bits(k, n, 0) == prng.random_bits(k, n, "legacy")).all()
            assert (co.random_bits(k, n, 1) == prng.random_bits(k, n, "partitionable")).all()
        assert np.array_equal(co.normal(k, 1001), prng.normal(k, (1001,)))
        assert np.abs(co.logistic(k, 1000) - prng.logistic(k, (1000,))).max() < 2e-6


def test_product_host_random_matches_oracle():
    for seed in (0, 3, 12345):
        k = prng.PRNGKey(seed)
        assert (prandom.PRNGKey(seed) == k).all()
        assert (prandom.split(k, 5) == prng.split(k, 5)).all()
        assert np.array_equal(prandom.normal(k, (4, 9)), prng.normal(k, (4, 9)))
        assert np.array_equal(prandom.uniform(k, (33,)), prng.uniform(k, (33,)))
        assert np.array_equal(prandom.logistic(k, (10,)), prng.logistic(k, (10,)))
        assert np.array_equal(prandom.bernoulli(k, 0.3, (50,)), prng.bernoulli(k, 0.3, (50,)))


def test_uniform_range_and_bernoulli_threshold_identity():
    k = prng.PRNGKey(11)
    bits = prng.random_bits(k, 4096)
    u = prng.uniform(k, (4096,))
    assert u.min() >= 0 and u.max() < 1
    # the device samples Bernoulli(p) as (bits >> 9) < ceil(p * 2^23): identical to uniform < p
    for p in (0.0, 1e-7, 0.25, 0.5, 0.7310586, 0.99999994, 1.0):
        pf = np.float32(p)
        thr = np.uint32(np.ceil(np.float64(pf) * 8388608.0))
        assert np.array_equal((bits >> np.uint32(9)) < thr, u < pf)
