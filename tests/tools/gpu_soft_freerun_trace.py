"""Per-step deviation of the free-running MarginalDiBS + reparam-BGe trajectory of tests/test_gpu_parity.py::test_marginal_bge_reparam_free_running_30_steps
from the float64 torch oracle: relative Z error after every step, and -- when a step's error jumps -- the coordinates that carry it together
with the oracle's phi there (a phi below the float32 noise of the largest one may take RMSprop's +-stepsize / sqrt(0.1) step with the other sign)."""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from conftest import make_data, rel_err
from dibs_amd._abi import make_config
from dibs_amd.engine import Engine
from oracle import prng, dibs_oracle as O
d, M, S, Sa, steps = 12, 4, 8, 4, 30
data, _, _ = make_data(d, seed=4)
x = data.x.astype(np.float32)
cfg = make_config(n_vars=d, n_particles=M, n_observations=x.shape[0], edges_per_node=2, grad_estimator_z="reparam", n_grad_mc_samples=S, n_acyclicity_mc_samples=Sa)
ocfg = O.Config(likelihood="bge", grad_estimator_z="reparam", n_grad_mc_samples=S, n_acyclicity_mc_samples=Sa, prior=O.GraphPrior("er", 2))
st = O.init_state(ocfg, prng.PRNGKey(9), M, d)
eng = Engine(cfg); eng.set_data(x); eng.init_particles(prng.PRNGKey(9))
xt = torch.as_tensor(x.astype(np.float64)); it = torch.zeros_like(xt)
prev = 0.0
for t in range(steps):
    dbg = {}
    st2 = O.svgd_step(ocfg, st, xt, it, t, debug=dbg) if "debug" in O.svgd_step.__code__.co_varnames else O.svgd_step(ocfg, st, xt, it, t)
    eng.run(t, 1)
    z = eng.get_state()["z"]; zo = st2.z.numpy()
    err = rel_err(z, zo)
    dz = np.abs(z - zo); worst = np.unravel_index(np.argmax(dz), dz.shape)
    step_o = (st2.z.numpy() - st.z.numpy())[worst]; step_d = (z - zprev)[worst] if t else float("nan")
    print(f"t={t:2d} rel err {err:.2e} (x{err / prev if prev else 0:.1f})  worst coord {worst}: |dz| {dz[worst]:.2e}, oracle step {step_o:+.3e}, device step {step_d:+.3e}")
    prev = err; st = st2; zprev = z.copy()
eng.close()
zf = np.abs(z - zo).ravel() / np.abs(zo).max()
print("final: max %.2e, p99.9 %.2e, p99 %.2e, p90 %.2e, median %.2e; coordinates above 1e-4: %d of %d" % (zf.max(), np.percentile(zf, 99.9), np.percentile(zf, 99), np.percentile(zf, 90), np.median(zf), int((zf > 1e-4).sum()), zf.size))
