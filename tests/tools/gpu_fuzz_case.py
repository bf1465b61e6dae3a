"""Re-run single fuzz trials and print stage-by-stage deviations:  python gpu_fuzz_case.py <seed> <trial> [<trial> ...]  (same stream as gpu_fuzz.py)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dibs_amd._abi import make_config
from dibs_amd.engine import Engine
from oracle import prng
from oracle.c_oracle import COracle
from gpu_fuzz import draw, rel

seed = int(sys.argv[1]); want = [int(a) for a in sys.argv[2:]]
rng = np.random.default_rng(seed)
co = COracle("f64")
for trial in range(max(want) + 1):
    fam, kw, interv, t = draw(rng)
    d, M, S, N = kw["n_vars"], kw["n_particles"], kw["n_grad_mc_samples"], kw["n_observations"]
    x = (rng.normal(size=(N, d)) @ (np.eye(d) + 0.3 * np.triu(rng.normal(size=(d, d)), 1))).astype(np.float32)
    if os.environ.get("FUZZ_SCALE"):
        x = (x * np.float32(rng.choice([1.0, 0.05, 8.0])) + np.float32(rng.choice([0.0, 0.0, 3.0]))).astype(np.float32)
    mask = (rng.random((N, d)) < 0.15).astype(np.int32) if interv else None
    try:
        cfg = make_config(**kw); eng = Engine(cfg)
    except Exception as e:
        continue
    key = prng.PRNGKey(int(rng.integers(1 << 30)))
    if trial not in want:
        eng.close(); continue
    eng.set_data(x, mask)
    st = co.new_state(cfg, key)
    st["v_z"] = np.ones_like(st["v_z"])
    if st.get("v_theta") is not None: st["v_theta"] = np.ones_like(st["v_theta"])
    print(f"== trial {trial} {fam} t={t}", {k: v for k, v in kw.items()})
    for step in (t, t + 1):
        for name in ("z", "v_z", "baseline", "theta", "v_theta"):
            if st.get(name) is not None: st[name] = st[name].astype(np.float32).astype(np.float64)
        sk = dict(z=st["z"], v_z=st["v_z"], key=st["key"], baseline=st["baseline"])
        if st.get("theta") is not None: sk.update(theta=st["theta"], v_theta=st["v_theta"])
        eng.set_state(**sk)
        dbg = co.step(cfg, x, mask, st, step, debug=True)
        eng.run(step, 1)
        g = eng.get_state()
        if st.get("theta") is not None: print(f"   theta rel {rel(g['theta'], st['theta']):.3e} max|theta| {np.abs(st['theta']).max():.3e}")
        print(f" step {step}: z rel {rel(g['z'], st['z']):.3e} v_z rel {rel(g['v_z'], st['v_z']):.3e}  oracle z finite {np.isfinite(st['z']).all()}  device z finite {np.isfinite(g['z']).all()}  baseline oracle {st['baseline'][:3]} device {g['baseline'][:3]}")
        names = {"SCORES": "scores", "NODE_SCORES": "node_scores", "LOGPROBS_Z": "logprobs_z", "W_LIK": "w_lik", "W_ACYC": "w_acyc", "GRAD_Z": "grad_z", "KXX": "kxx", "PHI_Z": "phi_z", "LOGPROBS_THETA": "logprobs_theta", "GRAD_THETA": "grad_theta", "PHI_THETA": "phi_theta"}
        for bn, on in names.items():
            if on not in dbg: continue
            try:
                a = eng.read(bn)
            except Exception as e:
                continue
            b = np.asarray(dbg[on])
            if bn == "NODE_SCORES" and a.size: a = a.reshape(M, d, S).transpose(0, 2, 1)
            if a.size != b.size: print("   ", bn, "size mismatch", a.shape, b.shape); continue
            print(f"    {bn:12s} rel {rel(a, b):.3e}  max|oracle| {np.abs(b).max():.3e} finite dev {np.isfinite(a).all()} orc {np.isfinite(b).all()}")
    eng.close()
