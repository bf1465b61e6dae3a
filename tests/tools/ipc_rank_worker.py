"""One rank of a particle-sharded run whose exchange goes through mapped peer memory (dibs_engine_comm_init_ipc): R of these processes
share ONE GPU and execute the N > 1 loop of dibs_engine_run_sharded -- rank != 0, the chunk boundaries, both protocols,
dibs_engine_gather_particles.  Started by tests/test_gpu_ipc.py; the rendez-vous is a directory (every rank writes its 128-byte blob to
<dir>/blob_<rank>.bin and reads the others'): neither torch.distributed nor RCCL is involved.

    python tests/tools/ipc_rank_worker.py <dir> <rank> <n_ranks> <case-json>
case: {"d", "M", "S", "Sa", "joint", "model", "chunks": [[t0, n], ...], "overlapped", "seed"}
Writes <dir>/out_<rank>.npz: z_all / theta_all as gathered on THIS rank after every chunk, and this rank's own carry state at the end."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def case_config(case, rank=0, n_ranks=1):
    from dibs_amd._abi import make_config
    kw = {}
    if case.get("joint"):
        kw = dict(joint=True, likelihood=case.get("model", "lingauss"))
        if kw["likelihood"] == "densenn":
            kw.update(nn_hidden=(5,), graph_prior="sf")
    return make_config(n_vars=case["d"], n_particles=case["M"], n_observations=case.get("N", 100), n_grad_mc_samples=case["S"],
                       n_acyclicity_mc_samples=case["Sa"], rank=rank, n_ranks=n_ranks, **kw)


def case_data(case):
    from conftest import make_data
    data, _, _ = make_data(case["d"], n_obs=case.get("N", 100), seed=case.get("data_seed", 3), joint=bool(case.get("joint")))
    return data.x


def exchange_blobs(rdv, rank, n_ranks, blob, timeout=120.0):
    tmp = os.path.join(rdv, f"blob_{rank}.tmp")
    with open(tmp, "wb") as f:
        f.write(blob)
    os.replace(tmp, os.path.join(rdv, f"blob_{rank}.bin"))   # (atomic: a reader never sees half a blob)
    out, t0 = [], time.time()
    for r in range(n_ranks):
        p = os.path.join(rdv, f"blob_{r}.bin")
        while not os.path.exists(p):
            if time.time() - t0 > timeout:
                raise TimeoutError(f"rank {rank}: blob of rank {r} did not appear")
            time.sleep(0.01)
        out.append(open(p, "rb").read())
    return out


def main():
    rdv, rank, n_ranks, case = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), json.loads(sys.argv[4])
    from dibs_amd import random as prng
    from dibs_amd.engine import Engine
    eng = Engine(case_config(case, rank, n_ranks))
    eng.set_data(case_data(case))
    eng.init_particles(prng.PRNGKey(case.get("seed", 8)))
    eng.comm_init_ipc(exchange_blobs(rdv, rank, n_ranks, eng.ipc_export()))
    out = {}
    for i, (t0, n) in enumerate(case["chunks"]):
        ov = case["overlapped"] if not isinstance(case["overlapped"], list) else case["overlapped"][i]
        if case.get("drop_flag") == [rank, i]:   # fault injection: this rank loses a completion flag in this chunk
            eng.debug_drop_next_flag()
        eng.run_sharded(t0, n, bool(ov))
        z, th = eng.gather_particles()
        out[f"z_{i}"] = z
        if th is not None:
            out[f"theta_{i}"] = th
    st = eng.get_state()
    out["own_z"], out["key"] = st["z"], st["key"]
    out["flag_fallbacks"] = np.int64(eng.flag_fallbacks())
    np.savez(os.path.join(rdv, f"out_{rank}.npz"), **out)
    # nobody unmaps while a peer may still be inside its last exchange: leave together
    open(os.path.join(rdv, f"done_{rank}"), "w").close()
    t0 = time.time()
    while not all(os.path.exists(os.path.join(rdv, f"done_{r}")) for r in range(n_ranks)) and time.time() - t0 < 60:
        time.sleep(0.01)
    eng.close()


if __name__ == "__main__":
    main()
