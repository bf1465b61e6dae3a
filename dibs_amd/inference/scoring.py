"""Device scoring of batches of hard graphs (post-hoc evaluators and mixture weights of the reference:
svgd.py:110-113, 370-372, 475-478, 838-841) through ``dibs_score_graphs``."""
import ctypes as C
import numpy as np

from .. import _lib
from .._abi import make_config
from ..engine import Engine


def score_graphs(likelihood_model, g, theta, x, interv_mask=None):
    g = np.ascontiguousarray(g, np.int32)
    x = np.ascontiguousarray(x, np.float32)
    n, d = g.shape[0], g.shape[-1]
    mask = None if interv_mask is None else np.ascontiguousarray(np.asarray(interv_mask) != 0, np.int32)
    joint = likelihood_model._dibs_likelihood != "bge"
    cfg = make_config(n_vars=d, n_particles=1, n_observations=x.shape[0], joint=joint, graph_prior="uniform",
                      has_interventions=mask is not None and bool(mask.any()), **likelihood_model._config_kwargs())
    eng = Engine(cfg)
    try:
        out = np.empty(n, np.float32)
        th = None
        if joint:
            th = np.ascontiguousarray(np.asarray(theta, np.float32).reshape(n, -1))
        mo = getattr(likelihood_model, "mean_obs", None)
        mo = None if mo is None else np.ascontiguousarray(mo, np.float32)
        p = lambda a: None if a is None else a.ctypes.data_as(C.c_void_p)
        eng._keep = (mo,)
        if mo is not None:  # BGe prior mean travels with the data
            eng.set_data(x, mask, mo)
        _lib.check(eng.lib.dibs_score_graphs(eng._h, p(g), p(th), n, p(x), p(mask), x.shape[0], p(out)))
        return out
    finally:
        eng.close()
