"""Posterior metrics on particle distributions (host side, O(M d^2)).  Same functions as dibs/metrics.py
(reference lines 12-270); the DAG filter uses the float32 matrix-power test of the reference on purpose."""
from typing import Any, NamedTuple

import numpy as np

from .graph_utils import elwise_acyclic_constr_nograd
from .utils.tree import tree_mul, tree_select


class ParticleDistribution(NamedTuple):
    logp: Any
    g: Any
    theta: Any = None


def _lse(a, b=None, axis=0):
    a = np.asarray(a, np.float64)
    m = np.max(a, axis=axis, keepdims=True)
    m = np.where(np.isfinite(m), m, 0.0)
    e = np.exp(a - m)
    s = np.sum(e if b is None else e * np.asarray(b, np.float64), axis=axis)
    with np.errstate(divide="ignore"):
        return np.log(np.abs(s)) + np.squeeze(m, axis), np.sign(s)


def pairwise_structural_hamming_distance(*, x, y):
    x, y = np.asarray(x), np.asarray(y)
    assert x.ndim == 3 and y.ndim == 3
    diff = np.abs(x[:, None] - y[None])
    diff = diff + diff.transpose((0, 1, 3, 2))
    diff = np.where(diff > 1, 1, diff)
    return diff.sum(axis=(2, 3)) / 2


def _dag_filter(dist, n_vars):
    return elwise_acyclic_constr_nograd(dist.g, n_vars) == 0


def expected_shd(*, dist, g):
    g = np.asarray(g)
    n_vars = g.shape[0]
    is_dag = _dag_filter(dist, n_vars)
    if is_dag.sum() == 0:
        return n_vars * (n_vars - 1) / 2
    particles = np.asarray(dist.g)[is_dag]
    lw = np.asarray(dist.logp, np.float64)[is_dag]
    lw = lw - _lse(lw)[0]
    shds = pairwise_structural_hamming_distance(x=particles, y=g[None]).squeeze(1)
    le, sg = _lse(lw, shds)
    return float(sg * np.exp(le))


def expected_edges(*, dist):
    n_vars = np.asarray(dist.g).shape[-1]
    is_dag = _dag_filter(dist, n_vars)
    if is_dag.sum() == 0:
        le, sg = _lse(dist.logp, np.asarray(dist.g).sum(axis=(-1, -2)))
        return float(sg * np.exp(le))
    particles = np.asarray(dist.g)[is_dag]
    lw = np.asarray(dist.logp, np.float64)[is_dag]
    lw = lw - _lse(lw)[0]
    le, sg = _lse(lw, particles.sum(axis=(-1, -2)))
    return float(sg * np.exp(le))


def threshold_metrics(*, dist, g):
    from sklearn import metrics as skm
    g = np.asarray(g)
    n_vars = g.shape[0]
    g_flat = g.reshape(-1)
    is_dag = _dag_filter(dist, n_vars)
    if is_dag.sum() == 0:
        base = float(g.sum() / (n_vars * (n_vars - 1)))
        return {"roc_auc": 0.5, "prc_auc": base, "ave_prec": base}
    particles = np.asarray(dist.g)[is_dag]
    lw = np.asarray(dist.logp, np.float64)[is_dag]
    lw = lw - _lse(lw)[0]
    le, sg = _lse(lw[:, None, None], particles)
    p_edge = (sg * np.exp(le)).reshape(-1)
    fpr, tpr, _ = skm.roc_curve(g_flat, p_edge)
    prec, rec, _ = skm.precision_recall_curve(g_flat, p_edge)
    return {"fpr": fpr.tolist(), "tpr": tpr.tolist(), "roc_auc": skm.auc(fpr, tpr), "precision": prec.tolist(),
            "recall": rec.tolist(), "prc_auc": skm.auc(rec, prec), "ave_prec": skm.average_precision_score(g_flat, p_edge)}


def neg_ave_log_marginal_likelihood(*, dist, eltwise_log_marginal_likelihood, x):
    n_vars = np.asarray(x).shape[1]
    is_dag = _dag_filter(dist, n_vars)
    if is_dag.sum() == 0:
        g = np.zeros((1, n_vars, n_vars), dtype=np.asarray(dist.g).dtype)
        lw = np.array([0.0])
    else:
        g = np.asarray(dist.g)[is_dag]
        lw = np.asarray(dist.logp, np.float64)[is_dag]
        lw = lw - _lse(lw)[0]
    ll = eltwise_log_marginal_likelihood(g, x)
    le, sg = _lse(lw, ll)
    return float(-sg * np.exp(le))


def neg_ave_log_likelihood(*, dist, eltwise_log_likelihood, x):
    assert dist.theta is not None
    n_vars = np.asarray(x).shape[1]
    is_dag = _dag_filter(dist, n_vars)
    if is_dag.sum() == 0:
        g = np.asarray(dist.g) * 0
        theta = tree_mul(dist.theta, 0.0)
        lw = np.asarray(dist.logp, np.float64) * 0.0
    else:
        g = np.asarray(dist.g)[is_dag]
        theta = tree_select(dist.theta, is_dag)
        lw = np.asarray(dist.logp, np.float64)[is_dag]
        lw = lw - _lse(lw)[0]
    ll = eltwise_log_likelihood(g, theta, x)
    le, sg = _lse(lw, ll)
    return float(-sg * np.exp(le))
