"""ctypes mirror of include/dibs_hip.h (struct layout + enums).  Pure definitions: importing this
module does not load the HIP library."""
import ctypes as C

ABI_VERSION = 2
MAX_HIDDEN_LAYERS = 8

LIK = {"bge": 0, "lingauss": 1, "densenn": 2}
PRIOR = {"er": 0, "sf": 1, "uniform": 2}
EST = {"score": 0, "reparam": 1}
OPT = {"gd": 0, "rmsprop": 1}
RNG = {"legacy": 0, "partitionable": 1}
ACT = {"relu": 0, "tanh": 1, "sigmoid": 2, "leakyrelu": 3}

BUF = dict(Z=0, V_Z=1, THETA=2, V_THETA=3, SCORES=4, LOGPROBS_Z=5, W_LIK=6, W_ACYC=7, GRAD_Z=8,
           GRAD_THETA=9, KXX=10, PHI_Z=11, BASELINE=12, NODE_SCORES=13, PARENT_MASKS=14,
           LOGPROBS_THETA=15, PHI_THETA=16, GATHER=17)
KERNELS = ["edge", "bge_nodes", "lik_weights", "acyc", "zgrad", "kmat", "phi_update", "lin_logprobs",
           "lin_grad", "nn_theta", "nn_z", "pack", "bge_big", "acyc_reduce", "particle_grad", "k15"]
K_COUNT = 16


class DibsConfig(C.Structure):
    _fields_ = [
        ("abi_version", C.c_int32),
        ("n_vars", C.c_int32),
        ("n_dim", C.c_int32),
        ("n_particles", C.c_int32),
        ("n_observations", C.c_int32),
        ("n_grad_mc_samples", C.c_int32),
        ("n_acyclicity_mc_samples", C.c_int32),
        ("joint", C.c_int32),
        ("likelihood", C.c_int32),
        ("graph_prior", C.c_int32),
        ("grad_estimator_z", C.c_int32),
        ("optimizer", C.c_int32),
        ("rng_layout", C.c_int32),
        ("logistic_minval_tiny", C.c_int32),
        ("has_interventions", C.c_int32),
        ("nn_n_hidden", C.c_int32),
        ("nn_hidden", C.c_int32 * MAX_HIDDEN_LAYERS),
        ("nn_activation", C.c_int32),
        ("nn_bias", C.c_int32),
        ("rank", C.c_int32),
        ("n_ranks", C.c_int32),
        ("device_id", C.c_int32),
        ("reserved_i", C.c_int32 * 5),
        ("alpha_linear", C.c_double),
        ("beta_linear", C.c_double),
        ("tau", C.c_double),
        ("h_latent", C.c_double),
        ("h_theta", C.c_double),
        ("scale_latent", C.c_double),
        ("scale_theta", C.c_double),
        ("stepsize", C.c_double),
        ("score_function_baseline", C.c_double),
        ("latent_prior_std", C.c_double),
        ("graph_prior_edges_per_node", C.c_double),
        ("bge_alpha_mu", C.c_double),
        ("bge_alpha_lambd", C.c_double),
        ("lin_obs_noise", C.c_double),
        ("lin_mean_edge", C.c_double),
        ("lin_sig_edge", C.c_double),
        ("lin_min_edge", C.c_double),
        ("nn_obs_noise", C.c_double),
        ("nn_sig_param", C.c_double),
        ("reserved_d", C.c_double * 6),
    ]


def make_config(*, n_vars, n_particles, n_observations, n_dim=None, joint=False, likelihood="bge",
                graph_prior="er", edges_per_node=2, grad_estimator_z=None, optimizer="rmsprop",
                stepsize=0.005, alpha_linear=None, beta_linear=1.0, tau=1.0, n_grad_mc_samples=128,
                n_acyclicity_mc_samples=32, score_function_baseline=0.0, latent_prior_std=None,
                h_latent=5.0, h_theta=500.0, scale_latent=1.0, scale_theta=1.0, rng_layout="legacy",
                logistic_minval_tiny=False, has_interventions=False, bge_alpha_mu=1.0, bge_alpha_lambd=None,
                lin_obs_noise=0.1, lin_mean_edge=0.0, lin_sig_edge=1.0, lin_min_edge=0.5,
                nn_hidden=(5,), nn_activation="relu", nn_bias=True, nn_obs_noise=0.1, nn_sig_param=1.0,
                rank=0, n_ranks=1, device_id=0):
    """Build a dibs_config with the reference's defaults (svgd.py:60-83 marginal, :425-448 joint)."""
    c = DibsConfig()
    c.abi_version = ABI_VERSION
    c.n_vars = int(n_vars)
    c.n_dim = int(n_dim or n_vars)
    c.n_particles = int(n_particles)
    c.n_observations = int(n_observations)
    c.n_grad_mc_samples = int(n_grad_mc_samples)
    c.n_acyclicity_mc_samples = int(n_acyclicity_mc_samples)
    c.joint = int(bool(joint))
    c.likelihood = LIK[likelihood]
    c.graph_prior = PRIOR[graph_prior]
    if grad_estimator_z is None:
        grad_estimator_z = "reparam" if joint else "score"
    c.grad_estimator_z = EST[grad_estimator_z]
    c.optimizer = OPT[optimizer]
    c.rng_layout = RNG[rng_layout]
    c.logistic_minval_tiny = int(bool(logistic_minval_tiny))
    c.has_interventions = int(bool(has_interventions))
    hl = tuple(nn_hidden)
    if len(hl) > MAX_HIDDEN_LAYERS:
        raise ValueError(f"at most {MAX_HIDDEN_LAYERS} hidden layers supported")
    c.nn_n_hidden = len(hl)
    for i, h in enumerate(hl):
        c.nn_hidden[i] = int(h)
    c.nn_activation = ACT[nn_activation]
    c.nn_bias = int(bool(nn_bias))
    c.rank, c.n_ranks, c.device_id = int(rank), int(n_ranks), int(device_id)
    c.alpha_linear = float((0.05 if joint else 1.0) if alpha_linear is None else alpha_linear)
    c.beta_linear = float(beta_linear)
    c.tau = float(tau)
    c.h_latent, c.h_theta = float(h_latent), float(h_theta)
    c.scale_latent, c.scale_theta = float(scale_latent), float(scale_theta)
    c.stepsize = float(stepsize)
    c.score_function_baseline = float(score_function_baseline)
    c.latent_prior_std = float(latent_prior_std) if latent_prior_std else 0.0
    c.graph_prior_edges_per_node = float(edges_per_node)
    c.bge_alpha_mu = float(bge_alpha_mu)
    c.bge_alpha_lambd = float(bge_alpha_lambd) if bge_alpha_lambd else 0.0
    c.lin_obs_noise, c.lin_mean_edge = float(lin_obs_noise), float(lin_mean_edge)
    c.lin_sig_edge, c.lin_min_edge = float(lin_sig_edge), float(lin_min_edge)
    c.nn_obs_noise, c.nn_sig_param = float(nn_obs_noise), float(nn_sig_param)
    return c
