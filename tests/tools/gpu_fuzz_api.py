"""Randomised tests of the rest of the boundary (GPU box; companion of gpu_fuzz.py):
  A  dibs_score_graphs (hard-graph scorers of the three models, interventions, dense / empty / random graphs) against the C oracle
  B  particle sharding: R rank engines + device concat bit-identical to one engine (marginal BGe and the joint models)
  C  chunking / checkpointing: run(0, n) == run(0, a); get_state; set_state; run(a, n - a) bit for bit
  D  MarginalDiBS + BGe with the reparameterised estimator against the torch-autograd oracle (small d)

    python tests/tools/gpu_fuzz_api.py [n_trials_per_section] [seed]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dibs_amd._abi import make_config          # noqa: E402
from dibs_amd.engine import Engine             # noqa: E402
from oracle import prng                        # noqa: E402
from oracle.c_oracle import COracle            # noqa: E402


def rel(a, b):
    a, b = np.asarray(a, np.float64).ravel(), np.asarray(b, np.float64).ravel()
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300))


def data(rng, n, d):
    return (rng.normal(size=(n, d)) @ (np.eye(d) + 0.3 * np.triu(rng.normal(size=(d, d)), 1))).astype(np.float32)


def model_kw(rng, fam):
    if fam == "bge":
        return dict(bge_alpha_mu=float(rng.choice([1.0, 0.5])))
    if fam == "lingauss":
        return dict(joint=True, likelihood="lingauss", lin_obs_noise=float(rng.choice([0.1, 0.5])), lin_sig_edge=float(rng.choice([1.0, 2.0])))
    hidden = [(5,), (8,), (4, 3), (70,)][int(rng.integers(4))]
    return dict(joint=True, likelihood="densenn", nn_hidden=hidden, nn_activation=str(rng.choice(["relu", "tanh", "sigmoid", "leakyrelu"])),
                nn_bias=bool(rng.random() < 0.7))


def section_a(rng, co, n):
    from dibs_amd.models import BGe, DenseNonlinearGaussian, LinearGaussian
    bad = 0
    for trial in range(n):
        fam = str(rng.choice(["bge", "lingauss", "densenn"]))
        d = int(rng.choice([2, 3, 6, 17, 33, 50, 64, 65, 80])) if fam != "densenn" else int(rng.choice([2, 5, 12, 20]))
        N = int(rng.choice([1, 7, 100, 150]))
        if os.environ.get("FUZZ_BIG"):   # larger graphs, observation counts beyond the LDS-resident kernels
            d = int(rng.choice([64, 80, 96, 104, 112])) if fam != "densenn" else int(rng.choice([20, 50, 100]))
            N = int(rng.choice([20, 130, 300, 500]))
        kw = model_kw(rng, fam)
        x = data(rng, N, d)
        mask = (rng.random((N, d)) < 0.2).astype(np.int32) if rng.random() < 0.4 and N > 1 else None
        G = 7
        g = (rng.random((G, d, d)) < rng.choice([0.05, 0.3, 0.9])).astype(np.int32)
        g[:, np.arange(d), np.arange(d)] = 0
        g[0] = 0
        g[1] = np.triu(np.ones((d, d), np.int32), 1)
        cfg = make_config(n_vars=d, n_particles=1, n_observations=N, edges_per_node=0.4 if d <= 3 else 1, has_interventions=mask is not None, **kw)
        # go through the public scorer (the facade builds theta layouts)
        from dibs_amd.inference.scoring import score_graphs
        if fam == "bge":
            lm = BGe(n_vars=d, alpha_mu=kw["bge_alpha_mu"])
            got = score_graphs(lm, g, None, x, mask)
            ref = co.score_graphs(cfg, x, mask, g)
        elif fam == "lingauss":
            lm = LinearGaussian(n_vars=d, obs_noise=kw["lin_obs_noise"], sig_edge=kw["lin_sig_edge"])
            th = rng.normal(size=(G, d, d)).astype(np.float32)
            got = score_graphs(lm, g, th, x, mask)
            ref = co.score_graphs(cfg, x, mask, g, th.reshape(G, -1).astype(np.float64))
        else:
            lm = DenseNonlinearGaussian(n_vars=d, hidden_layers=kw["nn_hidden"], activation=kw["nn_activation"], bias=kw["nn_bias"])
            sizes, P = [d] + list(kw["nn_hidden"]) + [1], 0
            for a_, b_ in zip(sizes[:-1], sizes[1:]):
                P += d * a_ * b_ + (d * b_ if kw["nn_bias"] else 0)
            flat = rng.normal(size=(G, P)).astype(np.float32)   # leaves concatenated in pytree order, as the boundary takes them
            got = score_graphs(lm, g, flat, x, mask)
            ref = co.score_graphs(cfg, x, mask, g, flat.astype(np.float64))
        e = rel(got, ref)
        ok = e < 5e-5
        bad += not ok
        if not ok or trial % 10 == 0:
            print(f"A[{trial}] {'ok  ' if ok else 'FAIL'} rel {e:.2e} {fam} d={d} N={N} interv={mask is not None} {kw}", flush=True)
    return bad


def _leaves(theta):
    if isinstance(theta, (list, tuple)):
        out = []
        for t in theta:
            out += _leaves(t)
        return out
    return [theta]


def run_sharded(cfg_kw, x, mask, key, R, steps, overlapped=False):
    """R rank engines in one process; the all-gathers replaced by device concats.  overlapped: the protocol of
    dibs_amd.distributed.run_sharded_overlapped (values on a side stream behind the optimizer step, kernel-matrix slab behind that gather,
    gradient rows between the phases)."""
    import torch
    tstream = torch.cuda.Stream()
    engs = []
    for r in range(R):
        e = Engine(make_config(rank=r, n_ranks=R, **cfg_kw), stream=tstream.cuda_stream)
        e.set_data(x, mask)
        e.init_particles(key)
        engs.append(e)
    if overlapped:
        n = engs[0].plane_elems_per_rank()
        side, exported, ready = torch.cuda.Stream(), torch.cuda.Event(), torch.cuda.Event()
        with torch.cuda.stream(tstream):
            vs = [torch.zeros(n, dtype=torch.float32, device="cuda") for _ in range(R)]
            gs = [torch.zeros(n, dtype=torch.float32, device="cuda") for _ in range(R)]
            planes = torch.zeros(2 * n * R, dtype=torch.float32, device="cuda")
            vals, grads = planes[:n * R], planes[n * R:]

            def exchange(done):
                for r in range(R):
                    if not done:
                        engs[r].export_values(vs[r].data_ptr())
                exported.record(tstream)
                with torch.cuda.stream(side):
                    side.wait_event(exported)
                    torch.cat(vs, out=vals)
                    for r in range(R):
                        engs[r].kmat_values(vals.data_ptr(), side.cuda_stream)
                    ready.record(side)

            exchange(False)
            for t in range(steps):
                for r in range(R):
                    engs[r].step_local_grads(t, gs[r].data_ptr())
                torch.cat(gs, out=grads)
                tstream.wait_event(ready)
                for r in range(R):
                    engs[r].step_update_planes(t, planes.data_ptr(), vs[r].data_ptr())
                exchange(True)
    else:
        n = engs[0].gather_elems_per_rank()
        with torch.cuda.stream(tstream):
            sends = [torch.zeros(n, dtype=torch.float32, device="cuda") for _ in range(R)]
            recv = torch.zeros(n * R, dtype=torch.float32, device="cuda")
            for t in range(steps):
                for r in range(R):
                    engs[r].step_local(t, sends[r].data_ptr())
                torch.cat(sends, out=recv)
                for r in range(R):
                    engs[r].step_update(t, recv.data_ptr())
    torch.cuda.synchronize()
    st = [e.get_state() for e in engs]
    for e in engs:
        e.close()
    out = {"z": np.concatenate([s["z"] for s in st]), "key": st[0]["key"]}
    if st[0].get("theta") is not None:
        out["theta"] = np.concatenate([s["theta"] for s in st])
    return out


def section_bc(rng, n):
    bad = 0
    for trial in range(n):
        fam = str(rng.choice(["bge", "bge", "lingauss", "densenn"]))
        d = int(rng.choice([3, 8, 20, 33, 40, 50, 64, 70])) if fam != "densenn" else int(rng.choice([3, 8, 12]))
        R = int(rng.choice([2, 4, 8]))
        M = R * int(rng.choice([1, 2, 3, 8]))
        N = int(rng.choice([5, 100, 140]))
        S = int(rng.choice([2, 8, 32, 128])) if d <= 33 else int(rng.choice([2, 8, 32]))
        Sa = int(rng.choice([1, 2, 4, 8, 32]))
        if os.environ.get("FUZZ_BIG"):
            d = int(rng.choice([50, 64, 80, 96, 112])) if fam != "densenn" else int(rng.choice([12, 30, 65]))
            M = R * int(rng.choice([2, 8, 16]))
            N = int(rng.choice([20, 130, 300]))
            S = int(rng.choice([8, 32, 64]))
        kw = dict(n_vars=d, n_particles=M, n_observations=N, n_grad_mc_samples=S, n_acyclicity_mc_samples=Sa,
                  edges_per_node=0.4 if d <= 3 else (1 if d <= 9 else 2), graph_prior=str(rng.choice(["er", "sf"])),
                  rng_layout=str(rng.choice(["legacy", "legacy", "partitionable"])), **model_kw(rng, fam))
        if fam != "bge":
            kw["grad_estimator_z"] = str(rng.choice(["reparam", "score"]))
        x = data(rng, N, d)
        mask = (rng.random((N, d)) < 0.15).astype(np.int32) if rng.random() < 0.3 else None
        kw["has_interventions"] = mask is not None
        key = prng.PRNGKey(int(rng.integers(1 << 30)))
        steps = int(rng.integers(2, 25))
        ref = Engine(make_config(**kw))
        ref.set_data(x, mask)
        ref.init_particles(key)
        ref.run(0, steps)
        sref = ref.get_state()
        ref.close()
        # C: chunked with a checkpoint in the middle, on a fresh engine
        a = int(rng.integers(1, steps))
        e1 = Engine(make_config(**kw))
        e1.set_data(x, mask)
        e1.init_particles(key)
        e1.run(0, a)
        snap = {k_: v for k_, v in e1.get_state().items() if v is not None}
        e1.close()
        e2 = Engine(make_config(**kw))
        e2.set_data(x, mask)
        e2.set_state(**snap)
        b = int(rng.integers(a, steps + 1))     # second chunk boundary on the same engine (b == a / b == steps: empty chunks)
        e2.run(a, b - a)
        e2.run(b, steps - b)
        s2 = e2.get_state()
        e2.close()
        okc = np.array_equal(s2["z"], sref["z"]) and (s2["key"] == sref["key"]).all() and \
            (sref.get("theta") is None or np.array_equal(s2["theta"], sref["theta"]))
        # B: sharded
        sh = run_sharded(kw, x, mask, key, R, steps, overlapped=bool(trial % 2))
        okb = np.array_equal(sh["z"], sref["z"]) and (sh["key"] == sref["key"]).all() and \
            (sref.get("theta") is None or np.array_equal(sh["theta"], sref["theta"]))
        fin = np.isfinite(sref["z"]).all()
        ok = okb and okc and fin
        bad += not ok
        if not ok or trial % 10 == 0:
            print(f"BC[{trial}] {'ok  ' if ok else 'FAIL'} sharded={okb} chunked={okc} finite={fin} {fam} d={d} M={M} R={R} S={S} Sa={Sa} N={N} "
                  f"interv={mask is not None} {kw.get('grad_estimator_z', 'score')} {kw['rng_layout']}", flush=True)
    return bad


def section_d(rng, n):
    import torch
    from oracle import dibs_oracle as O
    bad = 0
    for trial in range(n):
        d = int(rng.choice([2, 3, 5, 8, 12]))
        M, S, Sa = int(rng.choice([1, 2])), int(rng.choice([1, 2, 4])), int(rng.choice([1, 2]))
        N = int(rng.choice([5, 40]))
        if os.environ.get("FUZZ_SOFT_BIG"):   # the upper end of the soft-graph BGe kernel's range (torch-autograd oracle: slow)
            d, M, S, Sa = int(rng.choice([16, 17, 20, 31, 32, 33, 40, 47, 48, 49, 50, 57, 63, 64])), 1, int(rng.choice([1, 2])), 1   # (block boundaries of k_bge_soft_mf)
        x = data(rng, N, d)
        interv = rng.random() < 0.3
        mask = (rng.random((N, d)) < 0.15).astype(np.int32) if interv else None
        epn = 0.4 if d <= 3 else 1
        cfg = make_config(n_vars=d, n_particles=M, n_observations=N, edges_per_node=epn, grad_estimator_z="reparam",
                          n_grad_mc_samples=S, n_acyclicity_mc_samples=Sa, has_interventions=interv)
        ocfg = O.Config(likelihood="bge", grad_estimator_z="reparam", n_grad_mc_samples=S, n_acyclicity_mc_samples=Sa,
                        prior=O.GraphPrior("er", epn))
        key = prng.PRNGKey(int(rng.integers(1 << 30)))
        st = O.init_state(ocfg, key, M, d)
        eng = Engine(cfg)
        eng.set_data(x, mask)
        eng.init_particles(key)
        xt = torch.as_tensor(x.astype(np.float64))
        it = torch.as_tensor((mask if mask is not None else np.zeros_like(x)).astype(np.float64))
        worst = 0.0
        for t in (1, 4):
            st.z = torch.as_tensor(st.z.numpy().astype(np.float32).astype(np.float64))
            st.v_z = torch.ones_like(st.v_z)
            eng.set_state(z=st.z.numpy(), v_z=st.v_z.numpy(), key=st.key, baseline=np.zeros(M))
            st2, aux = O.svgd_step(ocfg, st, xt, it, t, return_aux=True)
            eng.run(t, 1)
            g = eng.get_state()
            worst = max(worst, rel(g["z"], st2.z.numpy()), rel(eng.read("PHI_Z"), aux["phi_z"].numpy()) / 20.0)
            st = st2
        eng.close()
        ok = worst < 1e-4
        bad += not ok
        if not ok or trial % 5 == 0:
            print(f"D[{trial}] {'ok  ' if ok else 'FAIL'} rel {worst:.2e} d={d} M={M} S={S} Sa={Sa} N={N} interv={interv}", flush=True)
    return bad


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    rng = np.random.default_rng(seed)
    co = COracle("f64")
    if os.environ.get("FUZZ_SOFT_BIG"):
        bad_a = bad_bc = 0
        bad_d = section_d(rng, n)
    else:
        bad_a = section_a(rng, co, n)
        bad_bc = section_bc(rng, n)
        bad_d = section_d(rng, max(n // 3, 5))
    print(f"A (scorers) {bad_a} bad, B/C (sharding, chunking) {bad_bc} bad, D (BGe reparam) {bad_d} bad")
    return 1 if bad_a + bad_bc + bad_d else 0


if __name__ == "__main__":
    sys.exit(main())
