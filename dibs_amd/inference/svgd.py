"""``MarginalDiBS`` / ``JointDiBS``: drop-in for the classes of dibs/inference/svgd.py (reference lines
17-375 and 380-844): same keyword-only constructors, same ``sample`` protocol (chunks of ``callback_every``
steps, callback after every chunk, step overshoot when ``callback_every`` does not divide ``steps``), same
return types.  The SVGD loop itself runs in the HIP engine (include/dibs_hip.h); model / kernel / prior
objects are recognised by type and lowered to ``dibs_config`` -- arbitrary callables are rejected."""
import numpy as np

from .. import random
from .._abi import make_config
from ..engine import Engine
from ..kernel import AdditiveFrobeniusSEKernel, JointAdditiveFrobeniusSEKernel
from ..metrics import ParticleDistribution
from .dibs import DiBS


def _tree_add_axis(theta):
    """a single particle's parameters with a leading particle axis of length one (array or the DenseNN pytree)"""
    from ..utils.tree import tree_map
    if isinstance(theta, np.ndarray) or not isinstance(theta, (list, tuple)):
        return np.asarray(theta)[None]
    return tree_map(lambda a: np.asarray(a)[None], theta)


def _logsumexp(a):
    a = np.asarray(a, np.float64)
    m = a.max()
    return m + np.log(np.exp(a - m).sum())


class _SVGDBase(DiBS):
    _joint = False

    def _init_common(self, *, x, graph_model, likelihood_model, interv_mask, kernel, kernel_param, optimizer,
                     optimizer_param, alpha_linear, beta_linear, tau, n_grad_mc_samples, n_acyclicity_mc_samples,
                     grad_estimator_z, score_function_baseline, latent_prior_std, verbose):
        x = np.asarray(x, np.float32)
        if interv_mask is None:
            interv_mask = np.zeros_like(x, dtype=np.int32)
        interv_mask = (np.asarray(interv_mask) != 0).astype(np.int32)
        if interv_mask.shape != x.shape:
            raise AssertionError("interv_mask must have the shape of x")
        super().__init__(x=x, interv_mask=interv_mask, alpha_linear=alpha_linear, beta_linear=beta_linear, tau=tau,
                         n_grad_mc_samples=n_grad_mc_samples, n_acyclicity_mc_samples=n_acyclicity_mc_samples,
                         grad_estimator_z=grad_estimator_z, score_function_baseline=score_function_baseline,
                         latent_prior_std=latent_prior_std, verbose=verbose)
        self.likelihood_model = likelihood_model
        self.graph_model = graph_model
        if not hasattr(graph_model, "_dibs_prior"):
            raise NotImplementedError("graph_model must be one of ErdosReniDAGDistribution, ScaleFreeDAGDistribution, "
                                      "UniformDAGDistributionRejection (arbitrary callables cannot be lowered to the device)")
        lik = getattr(likelihood_model, "_dibs_likelihood", None)
        allowed = ("lingauss", "densenn") if self._joint else ("bge",)
        if lik not in allowed:
            raise NotImplementedError(f"likelihood_model must be one of {allowed} for {type(self).__name__}; "
                                      "arbitrary callables cannot be lowered to the device")
        self.kernel = kernel(**kernel_param)
        if self._joint:
            if not isinstance(self.kernel, JointAdditiveFrobeniusSEKernel):
                raise NotImplementedError("kernel must be JointAdditiveFrobeniusSEKernel")
        elif not isinstance(self.kernel, AdditiveFrobeniusSEKernel):
            raise NotImplementedError("kernel must be AdditiveFrobeniusSEKernel")
        if optimizer not in ("gd", "rmsprop"):
            raise ValueError()
        self.optimizer = optimizer
        self.optimizer_param = optimizer_param
        self._engine = None

    # ---- lowering to the C ABI -------------------------------------------------------------------
    def _make_config(self, n_particles, n_dim, rank=0, n_ranks=1, device_id=0):
        k = self.kernel
        kw = dict(self.likelihood_model._config_kwargs())
        if self._joint:
            kw.update(h_latent=k.h_latent, h_theta=k.h_theta, scale_latent=k.scale_latent, scale_theta=k.scale_theta)
        else:
            kw.update(h_latent=k.h, scale_latent=k.scale)
        if self.grad_estimator_z not in ("score", "reparam"):
            raise ValueError(f"Unknown gradient estimator `{self.grad_estimator_z}`")
        return make_config(
            n_vars=self.n_vars, n_particles=n_particles, n_observations=self.x.shape[0], n_dim=n_dim, joint=self._joint,
            graph_prior=self.graph_model._dibs_prior, edges_per_node=getattr(self.graph_model, "n_edges_per_node", 2),
            grad_estimator_z=self.grad_estimator_z, optimizer=self.optimizer, stepsize=self.optimizer_param["stepsize"],
            alpha_linear=self.alpha_linear, beta_linear=self.beta_linear, tau=self.tau,
            n_grad_mc_samples=self.n_grad_mc_samples, n_acyclicity_mc_samples=self.n_acyclicity_mc_samples,
            score_function_baseline=self.score_function_baseline, latent_prior_std=self.latent_prior_std,
            rng_layout=random.LAYOUT, has_interventions=bool(self.interv_mask.any()), rank=rank, n_ranks=n_ranks,
            device_id=device_id, **kw)

    def _new_engine(self, n_particles, n_dim, stream=None, **kw):
        eng = Engine(self._make_config(n_particles, n_dim, **kw), stream=stream)
        eng.set_data(self.x, self.interv_mask if self.interv_mask.any() else None,
                     getattr(self.likelihood_model, "mean_obs", None))
        return eng

    # ---- estimators of the reference's DiBS base, on the device (dibs.py:255-269, 295-321, 467-485, 626-658) ----
    def _flat_thetas(self, thetas):
        if not self._joint or thetas is None:
            return None
        lm = self.likelihood_model
        if lm._dibs_likelihood == "densenn":
            return np.ascontiguousarray(lm.tree_to_flat(thetas), np.float32)
        th = np.asarray(thetas, np.float32)
        return np.ascontiguousarray(th.reshape(th.shape[0], -1))

    def _eval(self, zs, thetas, baselines, t, **keys):
        zs = np.ascontiguousarray(zs, np.float32)
        n_particles, d, k = zs.shape[0], zs.shape[1], zs.shape[2]
        eng = self._new_engine(n_particles, k)
        try:
            eng.set_state(z=zs, theta=self._flat_thetas(thetas),
                          baseline=None if baselines is None else np.ascontiguousarray(baselines, np.float32))
            return eng.eval_gradients(int(t), **keys)
        finally:
            eng.close()

    def eltwise_log_joint_prob(self, gs, single_theta, rng=None):
        """log p(theta, D | G) (marginal model: log p(D | G)) for a batch of hard graphs ``[n, d, d]`` on the training data (dibs.py:255-269)."""
        from .scoring import score_graphs
        gs = np.asarray(gs)
        th = None
        if self._joint:
            flat = self._flat_thetas(_tree_add_axis(single_theta))
            th = np.repeat(flat, gs.shape[0], axis=0)
        return score_graphs(self.likelihood_model, gs, th, self.x, self.interv_mask if self.interv_mask.any() else None)

    def eltwise_grad_z_likelihood(self, zs, thetas, baselines, t, subkeys):
        """Estimator of grad_Z log p(theta, D | Z) for every particle -> (``[n_particles, d, k, 2]``, baselines ``[n_particles]``)
        (dibs.py:295-321): the score-function or the Gumbel-softmax estimator, as ``grad_estimator_z`` says, with particle m's graphs drawn
        from ``subkeys[m]`` exactly as in one SVGD step (svgd.py:245-249, 699-701)."""
        if self.grad_estimator_z not in ("score", "reparam"):
            raise ValueError(f"Unknown gradient estimator `{self.grad_estimator_z}`")
        subkeys = np.asarray(subkeys, np.uint32)
        r = self._eval(zs, thetas, baselines, t, keys_lik=subkeys, keys_theta=subkeys if self._joint else None)
        return r["grad_z_lik"], r["baseline"]

    def eltwise_grad_theta_likelihood(self, zs, thetas, t, subkeys):
        """Estimator of grad_theta log p(theta, D | Z) for every particle, theta-shaped (dibs.py:467-551; joint models only)."""
        if not self._joint:
            raise NotImplementedError("the marginal model has no parameters")
        subkeys = np.asarray(subkeys, np.uint32)
        r = self._eval(zs, thetas, None, t, keys_lik=subkeys, keys_theta=subkeys)
        return self._theta_out(r["grad_theta"])

    def eltwise_grad_latent_prior(self, zs, subkeys, t):
        """grad_Z log p(Z) = -beta(t) E[grad h(G~)] - Z / sigma_z^2 + grad log p(G_alpha(Z)) for every particle (dibs.py:626-658); the
        acyclicity noise of particle m is drawn from ``subkeys[m]`` (used directly, dibs.py:595)."""
        zs = np.ascontiguousarray(zs, np.float32)
        eng = self._new_engine(zs.shape[0], zs.shape[2])
        try:
            # (the prior terms do not read theta; a joint engine's state still needs a well-formed one)
            eng.set_state(z=zs, theta=np.zeros((zs.shape[0], eng.P), np.float32) if eng.P else None)
            return eng.eval_gradients(int(t), keys_prior=np.asarray(subkeys, np.uint32))["grad_z_prior"]
        finally:
            eng.close()

    def _theta_out(self, flat):
        lm = self.likelihood_model
        if lm._dibs_likelihood == "lingauss":
            return flat.reshape(flat.shape[0], self.n_vars, self.n_vars)
        return lm.flat_to_tree(flat, self.n_vars)

    def _run_sample(self, key, n_particles, steps, n_dim_particles, callback, callback_every):
        key = random.as_key(key)
        n_dim = n_dim_particles or self.n_vars
        eng = self._new_engine(n_particles, n_dim)
        self._engine = eng
        try:
            eng.init_particles(key)
            if self.latent_prior_std is None:
                self.latent_prior_std = float(np.float32(1.0) / np.sqrt(np.float32(n_dim)))
            callback_every = callback_every or steps
            for t in (range(0, steps, callback_every) if steps else range(0)):
                eng.run(t, callback_every)
                if callback:
                    st = eng.get_state()
                    kw = dict(dibs=self, t=t + callback_every, zs=st["z"])
                    if self._joint:
                        kw["thetas"] = self._theta_out(st["theta"])
                    callback(**kw)
            st = eng.get_state()
        finally:
            eng.close()
            self._engine = None
        self.last_state = st
        return st


class MarginalDiBS(_SVGDBase):
    """SVGD inference of p(G | D) with a closed-form marginal likelihood (BGe)."""
    _joint = False

    def __init__(self, *, x, graph_model, likelihood_model, interv_mask=None, kernel=AdditiveFrobeniusSEKernel,
                 kernel_param=None, optimizer="rmsprop", optimizer_param=None, alpha_linear=1.0, beta_linear=1.0, tau=1.0,
                 n_grad_mc_samples=128, n_acyclicity_mc_samples=32, grad_estimator_z="score",
                 score_function_baseline=0.0, latent_prior_std=None, verbose=False):
        if kernel_param is None:
            kernel_param = {"h": 5.0}
        if optimizer_param is None:
            optimizer_param = {"stepsize": 0.005}
        self._init_common(x=x, graph_model=graph_model, likelihood_model=likelihood_model, interv_mask=interv_mask,
                          kernel=kernel, kernel_param=kernel_param, optimizer=optimizer, optimizer_param=optimizer_param,
                          alpha_linear=alpha_linear, beta_linear=beta_linear, tau=tau, n_grad_mc_samples=n_grad_mc_samples,
                          n_acyclicity_mc_samples=n_acyclicity_mc_samples, grad_estimator_z=grad_estimator_z,
                          score_function_baseline=score_function_baseline, latent_prior_std=latent_prior_std,
                          verbose=verbose)
        from .scoring import score_graphs
        lm = likelihood_model
        self.eltwise_log_marginal_likelihood_observ = lambda g, x_ho: score_graphs(lm, g, None, x_ho, None)
        self.eltwise_log_marginal_likelihood_interv = lambda g, x_ho, m_ho: score_graphs(lm, g, None, x_ho, m_ho)

    def sample(self, *, key, n_particles, steps, n_dim_particles=None, callback=None, callback_every=None):
        st = self._run_sample(key, n_particles, steps, n_dim_particles, callback, callback_every)
        return self.particle_to_g_lim(st["z"])

    def get_empirical(self, g):
        g = np.asarray(g)
        n = g.shape[0]
        unique, counts = np.unique(g, axis=0, return_counts=True)
        return ParticleDistribution(logp=np.log(counts) - np.log(n), g=unique)

    def get_mixture(self, g):
        from .scoring import score_graphs
        g = np.asarray(g)
        logp = score_graphs(self.likelihood_model, g, None, self.x,
                            self.interv_mask if self.interv_mask.any() else None).astype(np.float64)
        return ParticleDistribution(logp=logp - _logsumexp(logp), g=g)


class JointDiBS(_SVGDBase):
    """SVGD inference of p(G, Theta | D)."""
    _joint = True

    def __init__(self, *, x, graph_model, likelihood_model, interv_mask=None, kernel=JointAdditiveFrobeniusSEKernel,
                 kernel_param=None, optimizer="rmsprop", optimizer_param=None, alpha_linear=0.05, beta_linear=1.0, tau=1.0,
                 n_grad_mc_samples=128, n_acyclicity_mc_samples=32, grad_estimator_z="reparam",
                 score_function_baseline=0.0, latent_prior_std=None, verbose=False):
        if kernel_param is None:
            kernel_param = {"h_latent": 5.0, "h_theta": 500.0}
        if optimizer_param is None:
            optimizer_param = {"stepsize": 0.005}
        self._init_common(x=x, graph_model=graph_model, likelihood_model=likelihood_model, interv_mask=interv_mask,
                          kernel=kernel, kernel_param=kernel_param, optimizer=optimizer, optimizer_param=optimizer_param,
                          alpha_linear=alpha_linear, beta_linear=beta_linear, tau=tau, n_grad_mc_samples=n_grad_mc_samples,
                          n_acyclicity_mc_samples=n_acyclicity_mc_samples, grad_estimator_z=grad_estimator_z,
                          score_function_baseline=score_function_baseline, latent_prior_std=latent_prior_std,
                          verbose=verbose)
        from .scoring import score_graphs
        lm = likelihood_model
        flat = (lambda th: lm.tree_to_flat(th)) if lm._dibs_likelihood == "densenn" else (lambda th: np.asarray(th))
        self._flat_theta = flat
        self.eltwise_log_likelihood_observ = lambda g, th, x_ho: score_graphs(lm, g, flat(th), x_ho, None)
        self.eltwise_log_likelihood_interv = lambda g, th, x_ho, m_ho: score_graphs(lm, g, flat(th), x_ho, m_ho)

    def sample(self, *, key, n_particles, steps, n_dim_particles=None, callback=None, callback_every=None):
        st = self._run_sample(key, n_particles, steps, n_dim_particles, callback, callback_every)
        return self.particle_to_g_lim(st["z"]), self._theta_out(st["theta"])

    def get_empirical(self, g, theta):
        n = np.asarray(g).shape[0]
        return ParticleDistribution(logp=-np.log(n) * np.ones(n), g=np.asarray(g), theta=theta)

    def get_mixture(self, g, theta):
        from .scoring import score_graphs
        logp = score_graphs(self.likelihood_model, np.asarray(g), self._flat_theta(theta), self.x,
                            self.interv_mask if self.interv_mask.any() else None).astype(np.float64)
        return ParticleDistribution(logp=logp - _logsumexp(logp), g=np.asarray(g), theta=theta)
