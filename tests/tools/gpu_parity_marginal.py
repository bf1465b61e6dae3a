"""GPU bring-up check: one SVGD step of the HIP engine vs the C oracle (f64), stage by stage."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dibs_amd._abi import make_config
from dibs_amd.engine import Engine
from oracle.c_oracle import COracle
from oracle import prng


def synth(d, N, seed=0, epn=2):
    rng = np.random.default_rng(seed)
    p = min(0.9, epn * d / (d * (d - 1) / 2))
    A = np.tril((rng.random((d, d)) < p), -1)
    perm = rng.permutation(d)
    A = A[np.ix_(perm, perm)]
    Wt = (rng.normal(size=(d, d)) + np.sign(rng.normal(size=(d, d))) * 0.5) * A
    order = np.argsort(np.argsort(perm))  # not needed: do generic ancestral sampling by topological sort
    x = np.zeros((N, d))
    # topological order of A (A[i,j]=1 means i->j): node perm order
    import networkx as nx
    G = nx.DiGraph(A)
    for j in nx.topological_sort(G):
        x[:, j] = x @ Wt[:, j] + np.sqrt(0.1) * rng.normal(size=N)
    return x.astype(np.float32), A.astype(np.int32)


def rel(a, b):
    a = np.asarray(a, np.float64).reshape(-1); b = np.asarray(b, np.float64).reshape(-1)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300))


def main(d=5, M=4, S=128, Sa=32, steps=(0, 1, 5), epn=1):
    x, _ = synth(d, 100, 0, epn)
    cfg = make_config(n_vars=d, n_particles=M, n_observations=100, edges_per_node=epn, n_grad_mc_samples=S,
                      n_acyclicity_mc_samples=Sa)
    co = COracle("f64")
    st = co.new_state(cfg, prng.PRNGKey(1))
    eng = Engine(cfg)
    eng.set_data(x)
    eng.init_particles(prng.PRNGKey(1))
    g = eng.get_state()
    print(f"d={d} M={M}: init z rel {rel(g['z'], st['z']):.2e}  key {g['key']} vs {st['key']}")
    for t in steps:
        # start both from the oracle's state (f32-rounded) so stages can be compared one by one
        z32 = st["z"].astype(np.float32)
        st["z"] = z32.astype(np.float64); st["v_z"] = st["v_z"].astype(np.float32).astype(np.float64)
        eng.set_state(z=z32, v_z=st["v_z"], key=st["key"], baseline=st["baseline"])
        dbg = co.step(cfg, x.astype(np.float64), None, st, t, debug=True)
        eng.run(t, 1)
        g = eng.get_state()
        gm = eng.read("PARENT_MASKS").reshape(M, d, S, -1)
        # compare sampled graphs
        og = dbg["g_samples"]  # [M,S,d,d] (i,j)
        gg = np.zeros_like(og)
        for i in range(d):
            gg[:, :, i, :] = ((gm[:, :, :, i // 64] >> np.uint64(i % 64)) & np.uint64(1)).astype(np.uint8).transpose(0, 2, 1)
        nflip = int((gg != og).sum())
        ns = eng.read("NODE_SCORES").reshape(M, d, S).transpose(0, 2, 1)
        print(f" t={t}: graphs flipped {nflip}/{og.size}  scores {rel(eng.read('SCORES'), dbg['scores']):.2e}  "
              f"node {rel(ns, dbg['node_scores']):.2e}  lp {rel(eng.read('LOGPROBS_Z'), dbg['logprobs_z']):.2e}  "
              f"w_lik {rel(eng.read('W_LIK'), dbg['w_lik']):.2e}  w_acyc {rel(eng.read('W_ACYC'), dbg['w_acyc']):.2e}  "
              f"grad {rel(eng.read('GRAD_Z'), dbg['grad_z']):.2e}  kxx {rel(eng.read('KXX'), dbg['kxx']):.2e}  "
              f"phi {rel(eng.read('PHI_Z'), dbg['phi_z']):.2e}  z {rel(g['z'], st['z']):.2e} v {rel(g['v_z'], st['v_z']):.2e} "
              f"key {bool((g['key'] == st['key']).all())}")
    eng.close()


if __name__ == "__main__":
    main(5, 4, 128, 32, (0, 1, 5), 1)
    main(20, 8, 128, 32, (0, 3), 2)
    main(50, 4, 128, 32, (0, 2), 2)
    # timing at the headline size
    d, M = 50, 128
    x, _ = synth(d, 100, 0, 2)
    cfg = make_config(n_vars=d, n_particles=M, n_observations=100)
    eng = Engine(cfg); eng.set_data(x); eng.init_particles(prng.PRNGKey(1))
    eng.run(0, 3)
    eng.set_profiling(True); eng.reset_timers()
    t0 = time.time(); eng.run(3, 50); dt = time.time() - t0
    print(f"headline d=50 M=128: {50/dt:.1f} steps/s (profiling on)")
    for k_, (ms, n) in eng.timers().items():
        print(f"   {k_:12s} {ms/n*1e3:9.1f} us/launch  x{n}")
    eng.set_profiling(False)
    t0 = time.time(); eng.run(53, 100); dt = time.time() - t0
    print(f"headline d=50 M=128: {100/dt:.1f} steps/s")
