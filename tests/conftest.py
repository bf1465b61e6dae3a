import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `pytest -m gpu` on the GPU box)")


@pytest.fixture(scope="session")
def c_oracle64():
    from oracle.c_oracle import COracle
    return COracle("f64")


@pytest.fixture(scope="session")
def c_oracle32():
    from oracle.c_oracle import COracle
    return COracle("f32")


def make_data(d, n_obs=100, seed=0, prior="er", edges_per_node=None, joint=False):
    """Synthetic ER / SF linear-Gaussian data set through the product's own host-side factory."""
    from dibs_amd import random
    from dibs_amd.target import make_linear_gaussian_equivalent_model, make_linear_gaussian_model
    epn = edges_per_node if edges_per_node is not None else (1 if d <= 5 else 2)
    f = make_linear_gaussian_model if joint else make_linear_gaussian_equivalent_model
    data, gm, lm = f(key=random.PRNGKey(seed), n_vars=d, graph_prior_str=prior, edges_per_node=epn,
                     n_observations=n_obs)
    return data, gm, lm


def rel_err(a, b):
    a = np.asarray(a, np.float64).reshape(-1)
    b = np.asarray(b, np.float64).reshape(-1)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300))
