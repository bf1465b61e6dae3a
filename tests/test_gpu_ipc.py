"""The N > 1 loop of dibs_engine_run_sharded EXECUTED with N > 1 on one GPU: R processes (one rank each) share the device and exchange their
rows through mapped peer memory (dibs_engine_comm_init_ipc, dibs_amd/csrc/exchange_ipc.h) -- RCCL refuses a communicator whose ranks share a
GPU.  Everything a real multi-GPU run executes on rank != 0 runs here: the bring-up, both exchange protocols, chunk boundaries, the switch
between protocols, dibs_engine_gather_particles.  Must be BIT-identical to dibs_engine_run of a single-rank engine (PRNG rows are indexed by the
global particle id, phi sums over b in global order; svgd.py:226-267 / 673-721 of the reference is the arithmetic of a step)."""
import json
import os
import subprocess
import sys
import tempfile

import numpy as np
import pytest

from dibs_amd import random as prng

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = os.path.join(ROOT, "tests", "tools", "ipc_rank_worker.py")
sys.path.insert(0, os.path.join(ROOT, "tests", "tools"))

pytestmark = pytest.mark.gpu


def _reference(case):
    """the single-rank engine through the same chunks; states after every chunk"""
    from dibs_amd.engine import Engine
    from ipc_rank_worker import case_config, case_data
    eng = Engine(case_config(case))
    eng.set_data(case_data(case))
    eng.init_particles(prng.PRNGKey(case.get("seed", 8)))
    outs = []
    for t0, n in case["chunks"]:
        eng.run(t0, n)
        outs.append(eng.get_state())
    eng.close()
    return outs


def _run_ranks(case, R, extra_env=None, timeout=600):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"), DIBS_IPC_TIMEOUT_MS="20000")
    env.update(extra_env or {})
    with tempfile.TemporaryDirectory() as rdv:
        procs = [subprocess.Popen([sys.executable, WORKER, rdv, str(r), str(R), json.dumps(case)], env=env, stdout=subprocess.PIPE,
                                  stderr=subprocess.STDOUT, text=True) for r in range(R)]
        logs = []
        try:
            for p in procs:
                logs.append(p.communicate(timeout=timeout)[0])
        finally:
            for p in procs:   # (exactly the processes started here)
                if p.poll() is None:
                    p.kill()
        for r, p in enumerate(procs):
            assert p.returncode == 0, f"rank {r} failed:\n{logs[r][-3000:]}"
        return [dict(np.load(os.path.join(rdv, f"out_{r}.npz"))) for r in range(R)]


CASES = [
    # marginal BGe (score estimator), packed protocol then overlapped, across chunk boundaries
    (2, dict(d=20, M=16, S=32, Sa=8, chunks=[[0, 3], [3, 2], [5, 3]], overlapped=[0, 0, 1])),
    (4, dict(d=20, M=16, S=32, Sa=8, chunks=[[0, 3], [3, 4]], overlapped=[1, 1])),
    (4, dict(d=50, M=16, S=32, Sa=8, chunks=[[0, 2], [2, 3]], overlapped=[0, 1])),
    # joint LinearGaussian (theta rows travel as well)
    (2, dict(d=20, M=16, S=32, Sa=8, joint=True, chunks=[[0, 3], [3, 3]], overlapped=[1, 0])),
    (4, dict(d=20, M=16, S=32, Sa=8, joint=True, chunks=[[0, 2], [2, 2], [4, 2]], overlapped=[0, 1, 1])),
    # >= 256 particles: the SVGD transform as a GEMM (k_phi_gemm), 68 particles per rank
    (4, dict(d=6, M=272, S=16, Sa=4, chunks=[[0, 2], [2, 2]], overlapped=[0, 1])),
    # late step indices: the edge probabilities are saturated, every sample of a particle keeps a softmax weight and the gradient kernels share
    # them between blocks (GradSplit) -- the shares must not depend on the shard
    (2, dict(d=20, M=16, S=32, Sa=8, joint=True, chunks=[[400, 2], [402, 2]], overlapped=[0, 1])),
    (4, dict(d=12, M=8, S=24, Sa=4, joint=True, model="densenn", chunks=[[300, 2], [302, 1]], overlapped=[1, 0])),
    # eight ranks (the driver's largest layout), two particles each
    (8, dict(d=20, M=16, S=32, Sa=8, chunks=[[0, 3], [3, 3]], overlapped=[0, 1])),
    (8, dict(d=12, M=16, S=16, Sa=4, joint=True, chunks=[[0, 2], [2, 2]], overlapped=[1, 0])),
]


@pytest.mark.parametrize("R,case", CASES, ids=[f"R{r}-d{c['d']}-M{c['M']}-{'joint' if c.get('joint') else 'marg'}-t{c['chunks'][0][0]}-{''.join(map(str, c['overlapped']))}"
                                               for r, c in CASES])
def test_sharded_loop_over_mapped_memory_is_bit_identical(R, case):
    ref = _reference(case)
    outs = _run_ranks(case, R)
    Mloc = case["M"] // R
    for i, st in enumerate(ref):
        for r in range(R):   # every rank gathered ALL particles after every chunk
            assert np.array_equal(outs[r][f"z_{i}"], st["z"]), f"chunk {i}: z gathered on rank {r} differs from the single-rank run"
            if case.get("joint"):
                assert np.array_equal(outs[r][f"theta_{i}"], st["theta"]), f"chunk {i}: theta gathered on rank {r} differs"
    for r in range(R):
        assert np.array_equal(outs[r]["own_z"], ref[-1]["z"][r * Mloc:(r + 1) * Mloc])
        assert (outs[r]["key"] == ref[-1]["key"]).all(), "the loop-carry key advances identically on every rank"


@pytest.mark.parametrize("R", [4, 8])
def test_headline_size_ranks_on_one_gpu(R):
    """the headline workload (d = 50, 128 particles, S = 128, Sa = 32) as 4 ranks of 32 and as 8 ranks of 16 particles (the layout of
    `bench.py --gpus 8`) on one GPU, 6 steps in two chunks, packed then overlapped"""
    case = dict(d=50, M=128, S=128, Sa=32, chunks=[[0, 3], [3, 3]], overlapped=[0, 1], seed=1, data_seed=0)
    ref = _reference(case)
    outs = _run_ranks(case, R)
    for i, st in enumerate(ref):
        for r in range(R):
            assert np.array_equal(outs[r][f"z_{i}"], st["z"])


def test_missing_rank_times_out_with_an_error():
    """a rank whose peers never show up fails the run with a message instead of hanging: rank 1 of 2 exits right after the bring-up"""
    from dibs_amd._lib import DibsHipError
    from dibs_amd.engine import Engine
    from ipc_rank_worker import case_config, case_data, exchange_blobs
    case = dict(d=8, M=4, S=8, Sa=4)
    code = ("import sys, json; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "from ipc_rank_worker import *\n"
            "from dibs_amd.engine import Engine\n"
            "case = json.loads(sys.argv[2]); e = Engine(case_config(case, 1, 2)); e.set_data(case_data(case))\n"
            "exchange_blobs(sys.argv[1], 1, 2, e.ipc_export())\n"
            "import time; time.sleep(float(sys.argv[3]))\n") % (ROOT, os.path.join(ROOT, "tests", "tools"))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    os.environ["DIBS_IPC_TIMEOUT_MS"] = "300"
    try:
        with tempfile.TemporaryDirectory() as rdv:
            peer = subprocess.Popen([sys.executable, "-c", code, rdv, json.dumps(case), "8"], env=env)
            try:
                eng = Engine(case_config(case, 0, 2))
                eng.set_data(case_data(case))
                eng.init_particles(prng.PRNGKey(1))
                eng.comm_init_ipc(exchange_blobs(rdv, 0, 2, eng.ipc_export()))
                with pytest.raises(DibsHipError, match="did not arrive"):
                    eng.run_sharded(0, 1, False)
                eng.close()
            finally:
                peer.kill()
                peer.wait()
    finally:
        del os.environ["DIBS_IPC_TIMEOUT_MS"]
