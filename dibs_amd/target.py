"""Synthetic ground-truth Bayes nets and data (host side).  Same factory functions as dibs/target.py
(reference lines 12-321); random streams follow dibs_amd.random (jax-compatible for split/normal/bernoulli,
own stream for permutations), so data sets are deterministic per key but not bit-identical to jax's."""
from typing import Any, NamedTuple

import numpy as np

from . import random
from .models import (BGe, DenseNonlinearGaussian, ErdosReniDAGDistribution, LinearGaussian, ScaleFreeDAGDistribution,
                     UniformDAGDistributionRejection)


class Data(NamedTuple):
    passed_key: Any
    n_vars: int
    n_observations: int
    n_ho_observations: int
    g: Any
    theta: Any
    x: Any
    x_ho: Any
    x_interv: Any


def make_synthetic_bayes_net(*, key, n_vars, graph_model, generative_model, n_observations=100, n_ho_observations=100,
                             n_intervention_sets=10, perc_intervened=0.1):
    key = random.as_key(key)
    passed_key = key.copy()
    key, subk = random.split(key)
    g = np.asarray(graph_model.sample_G(subk))
    key, subk = random.split(key)
    theta = generative_model.sample_parameters(key=subk, n_vars=n_vars)
    key, subk = random.split(key)
    x = generative_model.sample_obs(key=subk, n_samples=n_observations, g=g, theta=theta)
    key, subk = random.split(key)
    x_ho = generative_model.sample_obs(key=subk, n_samples=n_ho_observations, g=g, theta=theta)
    x_interv = []
    for _ in range(n_intervention_sets):
        key, subk = random.split(key)
        n_interv = int(np.ceil(n_vars * perc_intervened))
        targets = random.choice(subk, n_vars, (n_interv,), replace=False)
        interv = {int(k): 0.0 for k in targets}
        key, subk = random.split(key)
        x_interv.append((interv, generative_model.sample_obs(key=subk, n_samples=n_observations, g=g, theta=theta,
                                                             interv=interv)))
    return Data(passed_key=passed_key, n_vars=n_vars, n_observations=n_observations, n_ho_observations=n_ho_observations,
                g=g, theta=theta, x=x, x_ho=x_ho, x_interv=x_interv)


def make_graph_model(*, n_vars, graph_prior_str, edges_per_node=2):
    if graph_prior_str == "er":
        return ErdosReniDAGDistribution(n_vars=n_vars, n_edges_per_node=edges_per_node)
    if graph_prior_str == "sf":
        return ScaleFreeDAGDistribution(n_vars=n_vars, n_edges_per_node=edges_per_node)
    assert n_vars <= 5, "Naive uniform DAG sampling only possible up to 5 nodes"
    return UniformDAGDistributionRejection(n_vars=n_vars)


def make_linear_gaussian_equivalent_model(*, key, n_vars=20, graph_prior_str="sf", bge_mean_obs=None, bge_alpha_mu=None,
                                          bge_alpha_lambd=None, obs_noise=0.1, mean_edge=0.0, sig_edge=1.0, min_edge=0.5,
                                          n_observations=100, n_ho_observations=100, edges_per_node=2):
    graph_model = make_graph_model(n_vars=n_vars, graph_prior_str=graph_prior_str, edges_per_node=edges_per_node)
    gen = LinearGaussian(n_vars=n_vars, obs_noise=obs_noise, mean_edge=mean_edge, sig_edge=sig_edge, min_edge=min_edge)
    lik = BGe(n_vars=n_vars, mean_obs=bge_mean_obs, alpha_mu=bge_alpha_mu, alpha_lambd=bge_alpha_lambd)
    key, subk = random.split(random.as_key(key))
    data = make_synthetic_bayes_net(key=subk, n_vars=n_vars, graph_model=graph_model, generative_model=gen,
                                    n_observations=n_observations, n_ho_observations=n_ho_observations)
    return data, graph_model, lik


def make_linear_gaussian_model(*, key, n_vars=20, graph_prior_str="sf", obs_noise=0.1, mean_edge=0.0, sig_edge=1.0,
                               min_edge=0.5, n_observations=100, n_ho_observations=100, edges_per_node=2):
    graph_model = make_graph_model(n_vars=n_vars, graph_prior_str=graph_prior_str, edges_per_node=edges_per_node)
    kw = dict(n_vars=n_vars, obs_noise=obs_noise, mean_edge=mean_edge, sig_edge=sig_edge, min_edge=min_edge)
    key, subk = random.split(random.as_key(key))
    data = make_synthetic_bayes_net(key=subk, n_vars=n_vars, graph_model=graph_model, generative_model=LinearGaussian(**kw),
                                    n_observations=n_observations, n_ho_observations=n_ho_observations)
    return data, graph_model, LinearGaussian(**kw)


def make_nonlinear_gaussian_model(*, key, n_vars=20, graph_prior_str="sf", obs_noise=0.1, sig_param=1.0,
                                  hidden_layers=(5,), n_observations=100, n_ho_observations=100, edges_per_node=2):
    graph_model = make_graph_model(n_vars=n_vars, graph_prior_str=graph_prior_str, edges_per_node=edges_per_node)
    kw = dict(n_vars=n_vars, hidden_layers=hidden_layers, obs_noise=obs_noise, sig_param=sig_param)
    key, subk = random.split(random.as_key(key))
    data = make_synthetic_bayes_net(key=subk, n_vars=n_vars, graph_model=graph_model,
                                    generative_model=DenseNonlinearGaussian(**kw), n_observations=n_observations,
                                    n_ho_observations=n_ho_observations)
    return data, graph_model, DenseNonlinearGaussian(**kw)
