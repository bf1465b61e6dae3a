"""Device scoring of batches of hard graphs (post-hoc evaluators and mixture weights of the reference:
svgd.py:110-113, 370-372, 475-478, 838-841) through ``dibs_score_graphs``.

The scoring engine (device buffers, stream) is kept per (model hyper-parameters, n_vars, BGe prior mean) and reused by later calls;
the engine itself keeps the statistics of the last data set it scored against (``get_mixture`` / ``neg_ave_log_*`` call this repeatedly
with the same x)."""
import atexit
import ctypes as C
import numpy as np

from .. import _lib
from .._abi import make_config
from ..engine import Engine

_ENGINES = {}     # key -> Engine (at most _MAX, least recently used first out)
_MAX = 4


def _close_all():
    for eng in _ENGINES.values():
        eng.close()
    _ENGINES.clear()


atexit.register(_close_all)


def _engine_for(likelihood_model, d, has_interv, mo):
    joint = likelihood_model._dibs_likelihood != "bge"
    kw = dict(likelihood_model._config_kwargs())
    key = (type(likelihood_model).__name__, d, bool(has_interv), tuple(sorted((k, repr(v)) for k, v in kw.items())),
           None if mo is None else mo.tobytes())
    eng = _ENGINES.pop(key, None)
    if eng is None:
        # (n_observations of the config is not used by dibs_score_graphs: the held-out set brings its own row count)
        cfg = make_config(n_vars=d, n_particles=1, n_observations=1, joint=joint, graph_prior="uniform", has_interventions=has_interv, **kw)
        eng = Engine(cfg)
        if mo is not None:   # BGe prior mean travels with the data
            eng._keep = (mo,)
            eng.set_data(np.zeros((1, d), np.float32), None, mo)
        while len(_ENGINES) >= _MAX:
            _ENGINES.pop(next(iter(_ENGINES))).close()
    _ENGINES[key] = eng      # most recently used last
    return eng


def score_graphs(likelihood_model, g, theta, x, interv_mask=None):
    g = np.ascontiguousarray(g, np.int32)
    x = np.ascontiguousarray(x, np.float32)
    n, d = g.shape[0], g.shape[-1]
    mask = None if interv_mask is None else np.ascontiguousarray(np.asarray(interv_mask) != 0, np.int32)
    joint = likelihood_model._dibs_likelihood != "bge"
    mo = getattr(likelihood_model, "mean_obs", None)
    mo = None if mo is None else np.ascontiguousarray(mo, np.float32)
    eng = _engine_for(likelihood_model, d, mask is not None and bool(mask.any()), mo)
    out = np.empty(n, np.float32)
    th = None
    if joint:
        th = np.ascontiguousarray(np.asarray(theta, np.float32).reshape(n, -1))
    p = lambda a: None if a is None else a.ctypes.data_as(C.c_void_p)
    _lib.check(eng.lib.dibs_score_graphs(eng._h, p(g), p(th), n, p(x), p(mask), x.shape[0], p(out)))
    return out
