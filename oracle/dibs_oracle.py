"""ORACLE (test infrastructure only -- never imported by the product path).

CPU restatement of the DiBS SVGD hot path of larslorch/dibs (reference @ /root/reference), written
with torch (CPU, float64 by default) so that every place the reference calls ``jax.grad`` is an actual
reverse-mode derivative here (``torch.autograd``) rather than a hand-derived closed form.  The closed
forms used by the C port (oracle/dibs_oracle.c) and by the HIP kernels are validated against this file.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import it.

Pinning status: the reference has no tests / golden vectors and cannot be imported here (needs jax),
so the floating-point part is "parity unpinned" at the JAX boundary; the PRNG bit-streams are pinned
by public known-answer values (oracle/prng.py, tests/test_prng.py), and the model formulas restated
here are pinned against independent ground truth in tests/test_known_answers.py (BGe: normal-Wishart
evidence in closed form via scipy, score equivalence; LinearGaussian: scipy.stats; acyclicity
constraint, graph priors, kernels, RMSprop: hand-computed values).

Every function cites the reference file:line it follows (paths relative to /root/reference).
"""
from dataclasses import dataclass, field
import math
import numpy as np
import torch

from . import prng

DT = torch.float64


# ----------------------------------------------------------------------------------------------
# model hyper-parameters (mirrors the constructor kwargs of the reference model classes)
# ----------------------------------------------------------------------------------------------
@dataclass
class BGeParams:  # dibs/models/linearGaussian.py:35-48
    mean_obs: object = None
    alpha_mu: float = 1.0
    alpha_lambd: float = None  # default n_vars + 2


@dataclass
class LinGaussParams:  # dibs/models/linearGaussian.py:190-195
    obs_noise: float = 0.1
    mean_edge: float = 0.0
    sig_edge: float = 1.0
    min_edge: float = 0.5


@dataclass
class DenseNNParams:  # dibs/models/nonlinearGaussian.py:105-111
    hidden_layers: tuple = (5,)
    obs_noise: float = 0.1
    sig_param: float = 1.0
    activation: str = "relu"
    bias: bool = True


@dataclass
class GraphPrior:  # dibs/models/graph.py:27-30, 126-129, 210-211
    kind: str = "er"  # 'er' | 'sf' | 'uniform'
    n_edges_per_node: int = 2


@dataclass
class Config:
    """kwargs of MarginalDiBS / JointDiBS (svgd.py:60-77, 425-442) + sample() sizes."""
    joint: bool = False
    likelihood: str = "bge"  # 'bge' | 'lingauss' | 'densenn'
    bge: BGeParams = field(default_factory=BGeParams)
    lin: LinGaussParams = field(default_factory=LinGaussParams)
    nn: DenseNNParams = field(default_factory=DenseNNParams)
    prior: GraphPrior = field(default_factory=GraphPrior)
    h_latent: float = 5.0
    h_theta: float = 500.0
    scale_latent: float = 1.0
    scale_theta: float = 1.0
    optimizer: str = "rmsprop"
    stepsize: float = 0.005
    alpha_linear: float = 1.0
    beta_linear: float = 1.0
    tau: float = 1.0
    n_grad_mc_samples: int = 128
    n_acyclicity_mc_samples: int = 32
    grad_estimator_z: str = "score"
    score_function_baseline: float = 0.0
    latent_prior_std: float = None
    rng_layout: str = "legacy"


def zero_diagonal(g):  # dibs/utils/func.py:117-125
    d = g.shape[-1]
    return g * (1.0 - torch.eye(d, dtype=g.dtype))


def _t(a):
    return torch.as_tensor(np.asarray(a), dtype=DT)


# ----------------------------------------------------------------------------------------------
# p(G | Z)                                                                  dibs/inference/dibs.py
# ----------------------------------------------------------------------------------------------
def scores_uv(z):  # einsum('...ik,...jk->...ij', u, v)   dibs.py:179-180
    return z[..., 0] @ z[..., 1].transpose(-1, -2)


def edge_probs(z, alpha):  # dibs.py:168-184
    return zero_diagonal(torch.sigmoid(alpha * scores_uv(z)))


def particle_to_g_lim(z):  # dibs.py:84-99
    return zero_diagonal((scores_uv(z) > 0).to(DT)).to(torch.int32)


def particle_to_soft_graph(z, eps, alpha, tau):  # dibs.py:121-140
    return zero_diagonal(torch.sigmoid(tau * (eps + alpha * scores_uv(z))))


def latent_log_prob(g, z, alpha):  # dibs.py:187-229 (edge_log_probs + latent_log_prob)
    s = scores_uv(z)
    log_p = zero_diagonal(torch.nn.functional.logsigmoid(alpha * s))
    log_1p = zero_diagonal(torch.nn.functional.logsigmoid(alpha * -s))
    return torch.sum(g * log_p + (1 - g) * log_1p)


def sample_g(p, key, n_samples, layout):  # dibs.py:102-119
    d = p.shape[-1]
    b = prng.bernoulli(key, p.detach().numpy().astype(np.float32), (n_samples, d, d), layout)
    return zero_diagonal(_t(b.astype(np.float64)))


# ----------------------------------------------------------------------------------------------
# graph priors                                                             dibs/models/graph.py
# ----------------------------------------------------------------------------------------------
def log_graph_prior_soft(soft_g, prior, d):
    if prior.kind == "er":  # graph.py:27-30, 93-108
        n_edges = prior.n_edges_per_node * d
        p = n_edges / ((d * (d - 1)) / 2)
        n_pairs = d * (d - 1) / 2.0
        e = soft_g.sum()
        return e * math.log(p) + (n_pairs - e) * math.log(1 - p)
    if prior.kind == "sf":  # graph.py:182-196
        return torch.sum(-3 * torch.log(1 + soft_g.sum(0)))
    if prior.kind == "uniform":  # graph.py:263-276
        return soft_g.sum() * 0.0
    raise ValueError(prior.kind)


# ----------------------------------------------------------------------------------------------
# likelihood models
# ----------------------------------------------------------------------------------------------
def _slogdet_masked(m, parents):  # dibs/utils/func.py:128-145
    d = parents.shape[0]
    mask = torch.outer(parents, parents)
    submat = mask * m + (1 - mask) * torch.eye(d, dtype=DT)
    return torch.linalg.slogdet(submat)[1]


def bge_node_score(j, n_parents, g, x, interv, hp):  # dibs/models/linearGaussian.py:63-118
    d = x.shape[-1]
    alpha_lambd = hp.alpha_lambd if hp.alpha_lambd is not None else d + 2
    mean_obs = _t(np.zeros(d)) if hp.mean_obs is None else _t(hp.mean_obs)
    small_t = (hp.alpha_mu * (alpha_lambd - d - 1)) / (hp.alpha_mu + 1)
    T = small_t * torch.eye(d, dtype=DT)
    keep = 1 - interv[:, j]
    xk = x * keep[:, None]
    N = keep.sum()
    if float(N) == 0.0:  # jnp.isclose(N, 0)
        return torch.zeros((), dtype=DT)
    x_bar = xk.sum(0, keepdim=True) / N
    x_center = (xk - x_bar) * keep[:, None]
    s_N = x_center.T @ x_center
    R = T + s_N + ((N * hp.alpha_mu) / (N + hp.alpha_mu)) * ((x_bar - mean_obs).T @ (x_bar - mean_obs))
    parents = g[:, j]
    parents_and_j = (g + torch.eye(d, dtype=DT))[:, j]
    log_gamma_term = (
        0.5 * (math.log(hp.alpha_mu) - torch.log(N + hp.alpha_mu))
        + torch.lgamma(0.5 * (N + alpha_lambd - d + n_parents + 1))
        - torch.lgamma(0.5 * (alpha_lambd - d + n_parents + 1))
        - 0.5 * N * math.log(math.pi)
        + 0.5 * (alpha_lambd - d + 2 * n_parents + 1) * math.log(small_t)
    )
    log_term_r = (
        0.5 * (N + alpha_lambd - d + n_parents) * _slogdet_masked(R, parents)
        - 0.5 * (N + alpha_lambd - d + n_parents + 1) * _slogdet_masked(R, parents_and_j)
    )
    return log_gamma_term + log_term_r


def bge_log_marginal(g, x, interv, hp):  # linearGaussian.py:120-170
    d = x.shape[-1]
    n_parents_all = g.sum(0)
    return sum(bge_node_score(j, n_parents_all[j], g, x, interv, hp) for j in range(d))


def _normal_logpdf(x, loc, scale):
    return -0.5 * ((x - loc) / scale) ** 2 - math.log(scale) - 0.5 * math.log(2 * math.pi)


def lingauss_log_joint(g, theta, x, interv, hp):  # linearGaussian.py:278-338
    log_prob_theta = torch.sum(g * _normal_logpdf(theta, hp.mean_edge, hp.sig_edge))
    ll = _normal_logpdf(x, x @ (g * theta), math.sqrt(hp.obs_noise))
    log_lik = torch.sum(torch.where(interv != 0, torch.zeros_like(ll), ll))
    return log_prob_theta + log_lik


_ACT = {"relu": torch.relu, "tanh": torch.tanh, "sigmoid": torch.sigmoid,
        "leakyrelu": lambda v: torch.nn.functional.leaky_relu(v, 0.01)}


def densenn_forward(theta, xin, hp):
    """theta: flat list [W_0[d,in,h0], b_0[d,h0], W_1[d,h0,h1], b_1, ..., W_L[d,hL,1], b_L[d,1]] (bias=True)
    or [W_0, W_1, ...] (bias=False); xin [d, N, d] -> means [N, d]   (nonlinearGaussian.py:35-81, 131-135)."""
    act = _ACT[hp.activation]
    h = xin
    stride = 2 if hp.bias else 1
    n_layers = len(theta) // stride
    for li in range(n_layers):
        w = theta[li * stride]
        h = torch.einsum("jnd,jdh->jnh", h, w)
        if hp.bias:
            h = h + theta[li * stride + 1][:, None, :]
        if li < n_layers - 1:
            h = act(h)
    return h.squeeze(-1).T


def densenn_log_joint(g, theta, x, interv, hp):  # nonlinearGaussian.py:248-326
    lp = 0.0
    stride = 2 if hp.bias else 1
    for i, leaf in enumerate(theta):
        leaf_lp = _normal_logpdf(leaf, 0.0, hp.sig_param)
        if i == 0:  # first-layer weights masked by g.T[:, :, None]   (:264-269)
            leaf_lp = leaf_lp * g.T[:, :, None]
        lp = lp + leaf_lp.sum()
    all_x_msk = x[None] * g.T[:, None]  # [d, N, d]     (:291)
    means = densenn_forward(theta, all_x_msk, hp)
    ll = _normal_logpdf(x, means, math.sqrt(hp.obs_noise))
    return lp + torch.sum(torch.where(interv != 0, torch.zeros_like(ll), ll))


def log_joint_prob(cfg, g, theta, x, interv):
    """the callable bound at svgd.py:94 / :459"""
    if cfg.likelihood == "bge":
        return bge_log_marginal(g, x, interv, cfg.bge)
    if cfg.likelihood == "lingauss":
        return lingauss_log_joint(g, theta[0], x, interv, cfg.lin)
    if cfg.likelihood == "densenn":
        return densenn_log_joint(g, theta, x, interv, cfg.nn)
    raise ValueError(cfg.likelihood)


# ----------------------------------------------------------------------------------------------
# theta containers: a particle's theta is a list of tensors (pytree leaves in jax flatten order)
# ----------------------------------------------------------------------------------------------
def sample_parameters(cfg, key, n_particles, d):
    """likelihood_model.sample_parameters(key=, n_particles=, n_vars=)   (svgd.py:512-513)"""
    if cfg.likelihood == "lingauss":  # linearGaussian.py:212-227
        hp = cfg.lin
        th = hp.mean_edge + hp.sig_edge * prng.normal(key, (n_particles, d, d), cfg.rng_layout).astype(np.float32)
        th = th + np.sign(th) * np.float32(hp.min_edge)
        return [[_t(th[m])] for m in range(n_particles)]
    if cfg.likelihood == "densenn":  # nonlinearGaussian.py:155-186 (+ stax.serial/Dense init key order)
        hp = cfg.nn
        subkeys = prng.split(key, n_particles * d, cfg.rng_layout).reshape(n_particles, d, 2)
        sizes = [d] + list(hp.hidden_layers) + [1]
        n_layers = len(sizes) - 1
        n_stax = 2 * n_layers - 1  # Dense, act, Dense, act, ..., Dense
        out = []
        for m in range(n_particles):
            leaves = [[] for _ in range(n_layers * (2 if hp.bias else 1))]
            for j in range(d):
                # stax.serial: rng, layer_rng = split(rng) per layer (activations consume one too)
                rng = subkeys[m, j]
                li = 0
                for si in range(n_stax):
                    rng, layer_rng = prng.split(rng, 2, cfg.rng_layout)
                    if si % 2 == 1:
                        continue  # activation layer: no params
                    fan_in, fan_out = sizes[li], sizes[li + 1]
                    if hp.bias:  # stax.Dense: k1, k2 = split(rng); W = W_init(k1), b = b_init(k2)
                        k1, k2 = prng.split(layer_rng, 2, cfg.rng_layout)
                        w = prng.normal(k1, (fan_in, fan_out), cfg.rng_layout) * np.float32(hp.sig_param)
                        b = prng.normal(k2, (fan_out,), cfg.rng_layout) * np.float32(hp.sig_param)
                        leaves[2 * li].append(w)
                        leaves[2 * li + 1].append(b)
                    else:  # dense_no_bias (nonlinearGaussian.py:17-32): w_init(rng, ...)
                        w = prng.normal(layer_rng, (fan_in, fan_out), cfg.rng_layout) * np.float32(hp.sig_param)
                        leaves[li].append(w)
                    li += 1
            out.append([_t(np.stack(leaf)) for leaf in leaves])
        return out
    raise NotImplementedError("BGe has no parameters (linearGaussian.py:53-54)")


# ----------------------------------------------------------------------------------------------
# signed logsumexp (jax.scipy.special.logsumexp(a, b=, return_sign=True))
# ----------------------------------------------------------------------------------------------
def _lse_signed(a, b, axis):
    """log|sum_s b_s exp(a_s)|, sign    (max-shifted like jax)"""
    amax = torch.max(a, dim=axis, keepdim=True)[0]
    amax = torch.where(torch.isfinite(amax), amax, torch.zeros_like(amax))
    s = torch.sum(b * torch.exp(a - amax), dim=axis)
    return torch.log(torch.abs(s)) + amax.squeeze(axis), torch.sign(s)


def _lse(a):
    return torch.logsumexp(a, dim=0)


# ----------------------------------------------------------------------------------------------
# estimators (per particle)                                                dibs/inference/dibs.py
# ----------------------------------------------------------------------------------------------
def grad_z_likelihood_score_function(cfg, z, theta, baseline, t, subk, x, interv):  # dibs.py:325-391
    S = cfg.n_grad_mc_samples
    alpha = cfg.alpha_linear * t
    p = edge_probs(z, alpha)
    subk, subk_ = prng.split(subk, 2, cfg.rng_layout)
    g_samples = sample_g(p, subk_, S, cfg.rng_layout)
    subk, subk_ = prng.split(subk, 2, cfg.rng_layout)  # unused rng for minibatching
    with torch.no_grad():
        logprobs = torch.stack([log_joint_prob(cfg, g_samples[s], theta, x, interv) for s in range(S)])
    logprobs_adj = logprobs if cfg.score_function_baseline <= 0.0 else logprobs - baseline
    grads = []
    for s in range(S):  # eltwise_grad_latent_log_prob   dibs.py:232-247
        zz = z.detach().clone().requires_grad_(True)
        grads.append(torch.autograd.grad(latent_log_prob(g_samples[s], zz, alpha), zz)[0].reshape(-1))
    grad_z = torch.stack(grads).T  # [D, S]
    log_num, sign = _lse_signed(logprobs_adj[None, :], grad_z, 1)
    log_den = _lse(logprobs)
    sf_grad = sign * torch.exp(log_num - math.log(S) - log_den + math.log(S))
    new_baseline = cfg.score_function_baseline * logprobs.mean() + (1 - cfg.score_function_baseline) * baseline
    aux = dict(logprobs=logprobs, g_samples=g_samples)
    return sf_grad.reshape(z.shape), new_baseline, aux


def grad_z_likelihood_gumbel(cfg, z, theta, baseline, t, subk, x, interv):  # dibs.py:395-459
    S = cfg.n_grad_mc_samples
    d = z.shape[0]
    alpha = cfg.alpha_linear * t
    subk, subk_ = prng.split(subk, 2, cfg.rng_layout)
    eps = _t(prng.logistic(subk_, (S, d, d), cfg.rng_layout))
    subk, subk_ = prng.split(subk, 2, cfg.rng_layout)
    logprobs, grads = [], []
    for s in range(S):  # log_joint_prob_soft  dibs.py:271-288 and its grad wrt z
        zz = z.detach().clone().requires_grad_(True)
        lp = log_joint_prob(cfg, particle_to_soft_graph(zz, eps[s], alpha, cfg.tau), theta, x, interv)
        grads.append(torch.autograd.grad(lp, zz)[0])
        logprobs.append(lp.detach())
    logprobs = torch.stack(logprobs)
    grad_z = torch.stack(grads)  # [S, d, k, 2]
    log_num, sign = _lse_signed(logprobs[:, None, None, None], grad_z, 0)
    log_den = _lse(logprobs)
    stable = sign * torch.exp(log_num - math.log(S) - log_den + math.log(S))
    return stable, baseline, dict(logprobs=logprobs, eps=eps)


def grad_theta_likelihood(cfg, z, theta, t, subk, x, interv):  # dibs.py:488-551
    S = cfg.n_grad_mc_samples
    alpha = cfg.alpha_linear * t
    p = edge_probs(z, alpha)
    g_samples = sample_g(p, subk, S, cfg.rng_layout)  # NOTE: particle key itself (dibs.py:510)
    subk, subk_ = prng.split(subk, 2, cfg.rng_layout)
    logprobs, grads = [], []
    for s in range(S):
        th = [leaf.detach().clone().requires_grad_(True) for leaf in theta]
        lp = log_joint_prob(cfg, g_samples[s], th, x, interv)
        grads.append(torch.autograd.grad(lp, th, allow_unused=True))
        logprobs.append(lp.detach())
    logprobs = torch.stack(logprobs)
    log_den = _lse(logprobs)
    out = []
    for li in range(len(theta)):
        gl = torch.stack([torch.zeros_like(theta[li]) if g[li] is None else g[li] for g in grads])
        a = logprobs.reshape((S,) + (1,) * (gl.ndim - 1))  # expand_by  func.py:8-18
        log_num, sign = _lse_signed(a, gl, 0)
        out.append(sign * torch.exp(log_num - math.log(S) - log_den + math.log(S)))
    return out, dict(logprobs=logprobs, g_samples=g_samples)


def acyclic_constr(mat, d):  # dibs/graph_utils.py:8-28
    M = torch.eye(d, dtype=DT) + mat / d
    return torch.trace(torch.linalg.matrix_power(M, d)) - d


def grad_constraint_gumbel(cfg, z, key, t):  # dibs.py:557-601
    d = z.shape[0]
    Sa = cfg.n_acyclicity_mc_samples
    alpha = cfg.alpha_linear * t
    eps = _t(prng.logistic(key, (Sa, d, d), cfg.rng_layout))  # NOTE: particle key itself (dibs.py:595)
    grads = []
    for s in range(Sa):
        zz = z.detach().clone().requires_grad_(True)
        h = acyclic_constr(particle_to_soft_graph(zz, eps[s], alpha, cfg.tau), d)
        grads.append(torch.autograd.grad(h, zz)[0])
    return torch.stack(grads).mean(0)


def grad_latent_prior(cfg, z, key, t, latent_prior_std):  # dibs.py:604-658
    d = z.shape[0]
    alpha = cfg.alpha_linear * t
    zz = z.detach().clone().requires_grad_(True)
    lp = log_graph_prior_soft(edge_probs(zz, alpha), cfg.prior, d)
    grad_prior_z = torch.autograd.grad(lp, zz)[0]
    grad_constraint = grad_constraint_gumbel(cfg, z, key, t)
    return -(cfg.beta_linear * t) * grad_constraint - z / (latent_prior_std ** 2.0) + grad_prior_z


# ----------------------------------------------------------------------------------------------
# kernel + SVGD transform                                   dibs/kernel.py, dibs/inference/svgd.py
# ----------------------------------------------------------------------------------------------
def _sqnorm_theta(a, b):  # func.py:100-114
    return sum(((la - lb) ** 2).sum() for la, lb in zip(a, b))


def f_kernel(cfg, za, tha, zb, thb):  # kernel.py:20-30 / :52-71
    k = cfg.scale_latent * torch.exp(-torch.sum((za - zb) ** 2.0) / cfg.h_latent)
    if cfg.joint:
        k = k + cfg.scale_theta * torch.exp(-_sqnorm_theta(tha, thb) / cfg.h_theta)
    return k


def kernel_mat(cfg, z, theta):  # svgd.py:165-176 / :537-551
    M = z.shape[0]
    with torch.no_grad():
        return torch.stack([torch.stack([f_kernel(cfg, z[a], theta[a] if cfg.joint else None,
                                                   z[b], theta[b] if cfg.joint else None)
                                         for b in range(M)]) for a in range(M)])


def svgd_phi(cfg, z, theta, kxx, dz_log_prob, dtheta_log_prob):
    """_parallel_update_z / _parallel_update_theta   svgd.py:194-224, 591-670.
    particle a uses COLUMN kxx[:, a] (vmap in_axes (0, 1, None, None))."""
    M = z.shape[0]
    phi_z = []
    phi_theta = []
    for a in range(M):
        rep_z = []
        rep_th = []
        for b in range(M):  # grad wrt FIRST argument (z_b, theta_b) of k((z_b,th_b),(z_a,th_a))
            zb = z[b].detach().clone().requires_grad_(True)
            if cfg.joint:
                thb = [leaf.detach().clone().requires_grad_(True) for leaf in theta[b]]
                kv = f_kernel(cfg, zb, thb, z[a], theta[a])
                gs = torch.autograd.grad(kv, [zb] + thb)
                rep_z.append(gs[0])
                rep_th.append(gs[1:])
            else:
                kv = f_kernel(cfg, zb, None, z[a], None)
                rep_z.append(torch.autograd.grad(kv, zb)[0])
        rep_z = torch.stack(rep_z)
        wga = kxx[:, a][:, None, None, None] * dz_log_prob
        phi_z.append(-(wga + rep_z).mean(0))
        if cfg.joint:
            leaves = []
            for li in range(len(theta[0])):
                gl = torch.stack([dtheta_log_prob[b][li] for b in range(M)])
                kk = kxx[:, a].reshape((M,) + (1,) * (gl.ndim - 1))
                rl = torch.stack([rep_th[b][li] for b in range(M)])
                leaves.append(-(kk * gl + rl).mean(0))
            phi_theta.append(leaves)
    return torch.stack(phi_z), phi_theta


def opt_update(cfg, g, x, v):
    """jax.example_libraries.optimizers.rmsprop(step, gamma=0.9, eps=1e-8) / sgd(step)   (svgd.py:117-120)
    third-party; restated from the published source: v = v*gamma + g^2*(1-gamma); x -= step*g/sqrt(v+eps)."""
    if cfg.optimizer == "rmsprop":
        v = v * 0.9 + g * g * (1.0 - 0.9)
        return x - cfg.stepsize * g / torch.sqrt(v + 1e-8), v
    if cfg.optimizer == "gd":
        return x - cfg.stepsize * g, v
    raise ValueError()


# ----------------------------------------------------------------------------------------------
# state + step + sample
# ----------------------------------------------------------------------------------------------
@dataclass
class State:
    z: torch.Tensor
    v_z: torch.Tensor
    theta: list = None  # per particle list of leaves
    v_theta: list = None
    key: np.ndarray = None
    sf_baseline: torch.Tensor = None
    latent_prior_std: float = None
    t: int = 0


def init_state(cfg, key, n_particles, d, n_dim=None):
    """sample(): svgd.py:293-307 / 750-766 and _sample_initial_random_particles :125-148 / :489-515"""
    key = np.asarray(key, dtype=np.uint32)
    key, subk = prng.split(key, 2, cfg.rng_layout)
    k = n_dim or d
    std = cfg.latent_prior_std or np.float32(1.0 / np.sqrt(np.float32(k)))
    ikey, isubk = prng.split(subk, 2, cfg.rng_layout)
    z = prng.normal(isubk, (n_particles, d, k, 2), cfg.rng_layout) * np.float32(std)
    theta = None
    if cfg.joint:
        ikey, isubk = prng.split(ikey, 2, cfg.rng_layout)
        theta = sample_parameters(cfg, isubk, n_particles, d)
    lps = cfg.latent_prior_std if cfg.latent_prior_std is not None else float(1.0 / np.sqrt(np.float32(k)))
    st = State(z=_t(z), v_z=torch.zeros(z.shape, dtype=DT), theta=theta, key=key,
               sf_baseline=torch.zeros(n_particles, dtype=DT), latent_prior_std=float(lps), t=0)
    if theta is not None:
        st.v_theta = [[torch.zeros_like(leaf) for leaf in th] for th in theta]
    return st


def svgd_step(cfg, st, x, interv, t, return_aux=False):
    """_svgd_step: svgd.py:226-267 (marginal) / :673-721 (joint)"""
    M = st.z.shape[0]
    z = st.z
    key = st.key
    aux = {}
    dtheta = None
    if cfg.joint:
        ks = prng.split(key, M + 1, cfg.rng_layout)
        key = ks[0]
        dtheta = []
        for m in range(M):
            gth, a = grad_theta_likelihood(cfg, z[m], st.theta[m], t, ks[1 + m], x, interv)
            dtheta.append(gth)
        aux["dtheta"] = dtheta
    ks = prng.split(key, M + 1, cfg.rng_layout)
    key = ks[0]
    if cfg.grad_estimator_z == "score":
        est = grad_z_likelihood_score_function
    elif cfg.grad_estimator_z == "reparam":
        est = grad_z_likelihood_gumbel
    else:
        raise ValueError(f"Unknown gradient estimator `{cfg.grad_estimator_z}`")
    dz_lik, new_b, lik_aux = [], [], []
    for m in range(M):
        g_, b_, a_ = est(cfg, z[m], st.theta[m] if cfg.joint else None, st.sf_baseline[m], t, ks[1 + m], x, interv)
        dz_lik.append(g_)
        new_b.append(b_)
        lik_aux.append(a_)
    dz_lik = torch.stack(dz_lik)
    ks = prng.split(key, M + 1, cfg.rng_layout)
    key = ks[0]
    dz_prior = torch.stack([grad_latent_prior(cfg, z[m], ks[1 + m], t, st.latent_prior_std) for m in range(M)])
    dz = dz_prior + dz_lik
    kxx = kernel_mat(cfg, z, st.theta)
    phi_z, phi_theta = svgd_phi(cfg, z, st.theta, kxx, dz, dtheta)
    new_z, new_vz = opt_update(cfg, phi_z, z, st.v_z)
    out = State(z=new_z, v_z=new_vz, theta=st.theta, v_theta=st.v_theta, key=key,
                sf_baseline=torch.stack([torch.as_tensor(b, dtype=DT) for b in new_b]),
                latent_prior_std=st.latent_prior_std, t=t + 1)
    if cfg.joint:
        nth, nvt = [], []
        for m in range(M):
            lt, lv = [], []
            for li in range(len(st.theta[m])):
                a, b = opt_update(cfg, phi_theta[m][li], st.theta[m][li], st.v_theta[m][li])
                lt.append(a)
                lv.append(b)
            nth.append(lt)
            nvt.append(lv)
        out.theta, out.v_theta = nth, nvt
    if return_aux:
        aux.update(dz_lik=dz_lik, dz_prior=dz_prior, kxx=kxx, phi_z=phi_z, phi_theta=phi_theta, lik_aux=lik_aux)
        return out, aux
    return out


def sample(cfg, x, interv, key, n_particles, steps, n_dim=None, callback=None, callback_every=None):
    """MarginalDiBS.sample / JointDiBS.sample   svgd.py:274-331 / 730-795 (incl. the step overshoot when
    callback_every does not divide steps)."""
    x = _t(x)
    interv = torch.zeros_like(x) if interv is None else _t(interv)
    st = init_state(cfg, key, n_particles, x.shape[-1], n_dim)
    callback_every = callback_every or steps
    for t0 in (range(0, steps, callback_every) if steps else range(0)):
        for t in range(t0, t0 + callback_every):
            st = svgd_step(cfg, st, x, interv, t)
        if callback:
            callback(t=t0 + callback_every, zs=st.z, thetas=st.theta)
    g = particle_to_g_lim(st.z)
    return (g, st.theta, st) if cfg.joint else (g, st)
