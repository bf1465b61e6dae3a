import sys, numpy as np, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from conftest import make_data, rel_err
from dibs_amd._abi import make_config
from dibs_amd.engine import Engine
from oracle import prng, dibs_oracle as O
for (d, M, S, Sa) in [(12, 1, 4, 2), (20, 1, 4, 2), (40, 1, 2, 2), (50, 2, 4, 2)]:
    data, _, _ = make_data(d, seed=4); x = data.x.astype(np.float32)
    cfg = make_config(n_vars=d, n_particles=M, n_observations=x.shape[0], edges_per_node=2, grad_estimator_z="reparam", n_grad_mc_samples=S, n_acyclicity_mc_samples=Sa)
    ocfg = O.Config(likelihood="bge", grad_estimator_z="reparam", n_grad_mc_samples=S, n_acyclicity_mc_samples=Sa, prior=O.GraphPrior("er", 2))
    st = O.init_state(ocfg, prng.PRNGKey(9), M, d)
    eng = Engine(cfg); eng.set_data(x); eng.init_particles(prng.PRNGKey(9))
    xt = torch.as_tensor(x.astype(np.float64)); it = torch.zeros_like(xt)
    for t in (1, 3):
        st.z = torch.as_tensor(st.z.numpy().astype(np.float32).astype(np.float64)); st.v_z = torch.as_tensor(st.v_z.numpy().astype(np.float32).astype(np.float64))
        eng.set_state(z=st.z.numpy(), v_z=st.v_z.numpy(), key=st.key, baseline=np.zeros(M))
        st2, aux = O.svgd_step(ocfg, st, xt, it, t, return_aux=True)
        eng.run(t, 1)
        lp_o = np.stack([a["logprobs"].numpy() for a in aux["lik_aux"]])
        dz = (aux["dz_lik"] + aux["dz_prior"]).numpy()
        print(d, t, "logprobs rel", rel_err(eng.read("LOGPROBS_Z"), lp_o), "abs", np.abs(eng.read("LOGPROBS_Z").reshape(lp_o.shape) - lp_o).max(), "grad_z rel", rel_err(eng.read("GRAD_Z"), dz))
        st = st2
    eng.close()
