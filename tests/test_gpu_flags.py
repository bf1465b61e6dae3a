"""The in-kernel flags between the engine's two streams (fork: k_wait_flag, join: tail_join_wait in k_particle_grad) must DEGRADE, not
fail: a chunk that saw a bounded wait run out is repeated on events from the chunk-start copy of the loop carry, and the engine stays on
events (include/dibs_hip.h: dibs_engine_flag_fallbacks).  Deterministic through fault injection (dibs_engine_debug_drop_next_flag), and on
a starved device: HSA_CU_MASK down to 32 CUs, and beside a second process that fills the GPU.  The arithmetic of a step
(svgd.py:226-267 of the reference) does not depend on how the streams synchronise: every result must be BIT-identical to the undisturbed run."""
import json
import os
import subprocess
import sys
import tempfile

import numpy as np
import pytest

from dibs_amd import random as prng

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOOLS = os.path.join(ROOT, "tests", "tools")
sys.path.insert(0, TOOLS)

pytestmark = pytest.mark.gpu

CASE = dict(d=50, M=128, S=64, Sa=16, chunks=[[0, 3], [3, 3], [6, 2]], seed=5, data_seed=1)


@pytest.fixture(autouse=True)
def _flags_even_with_other_engines_alive(monkeypatch):
    # (an engine uses the flags only while it is alone in its process; earlier tests of a session may have left engines behind -- the
    #  facade keeps its scoring engines.  DIBS_FLAGS_MULTI is latched at creation: tuning.h)
    monkeypatch.setenv("DIBS_FLAGS_MULTI", "1")


def _run_here(case, drop_in_chunk=None):
    from dibs_amd.engine import Engine
    from ipc_rank_worker import case_config, case_data
    eng = Engine(case_config(case))
    eng.set_data(case_data(case))
    eng.init_particles(prng.PRNGKey(case.get("seed", 8)))
    for i, (t0, n) in enumerate(case["chunks"]):
        if drop_in_chunk == i:
            eng.debug_drop_next_flag()
        eng.run(t0, n)
    st, fb = eng.get_state(), eng.flag_fallbacks()
    eng.close()
    return st, fb


def test_lost_flag_is_repaired_by_a_rerun_on_events():
    ref, fb0 = _run_here(CASE)
    assert fb0 == 0
    got, fb = _run_here(CASE, drop_in_chunk=1)
    assert fb == 1, "the chunk with the lost flag must have been repeated exactly once (is the engine using the flags at all?)"
    assert np.array_equal(got["z"], ref["z"]) and np.array_equal(got["v_z"], ref["v_z"]) and (got["key"] == ref["key"]).all()
    assert np.array_equal(got["baseline"], ref["baseline"])


def test_lost_flag_joint_model():
    case = dict(d=20, M=32, S=32, Sa=8, joint=True, chunks=[[0, 2], [2, 3], [5, 2]], seed=2)
    ref, _ = _run_here(case)
    got, fb = _run_here(case, drop_in_chunk=1)
    assert fb == 1
    assert np.array_equal(got["z"], ref["z"]) and np.array_equal(got["theta"], ref["theta"]) and np.array_equal(got["v_theta"], ref["v_theta"])


def test_lost_flag_on_one_rank_of_a_sharded_run():
    """rank 1 of 2 loses a flag in the second chunk: BOTH ranks repeat it (one agreement all-gather per chunk) and end bit-identical"""
    from test_gpu_ipc import _reference, _run_ranks
    case = dict(d=50, M=32, S=32, Sa=8, chunks=[[0, 2], [2, 3], [5, 2]], overlapped=[0, 1, 0], drop_flag=[1, 1])
    ref = _reference({k: v for k, v in case.items() if k != "drop_flag"})
    outs = _run_ranks(case, 2)
    for r in range(2):
        assert int(outs[r]["flag_fallbacks"]) == 1, f"rank {r}"
        for i, st in enumerate(ref):
            assert np.array_equal(outs[r][f"z_{i}"], st["z"]), f"rank {r} chunk {i}"


def _worker(case, env_extra):
    env = dict(os.environ)
    env.update(env_extra)
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "o.npz")
        r = subprocess.run([sys.executable, os.path.join(TOOLS, "single_run_worker.py"), out, json.dumps(case)], env=env, capture_output=True, text=True,
                           timeout=600)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        return dict(np.load(out))


def test_masked_down_device():
    """32 of the 256 CUs (HSA_CU_MASK): the polling blocks of k_particle_grad can occupy every CU the second stream would need"""
    ref, _ = _run_here(CASE)
    got = _worker(CASE, {"HSA_CU_MASK": "0:0-31"})
    print("HSA_CU_MASK 0:0-31: chunks repeated on events:", int(got["flag_fallbacks"]))
    assert np.array_equal(got["z"], ref["z"]) and (got["key"] == ref["key"]).all()


def test_persistent_gradient_kernel_on_a_masked_down_device():
    """JointDiBS + DenseNonlinearGaussian late in a run (several weighted samples per particle) on 32 of the 256 CUs: k_nn_grad launches one or
    two persistent blocks per CU of the WHOLE device (kernels_nn.h); the ones that are not resident must simply find the item list empty when
    their turn comes -- no block waits for another one.  Bit-identical to the run on the full device."""
    case = dict(d=20, M=16, S=32, Sa=8, joint=True, model="densenn", chunks=[[300, 2], [302, 2]], seed=3)
    ref, _ = _run_here(case)
    got = _worker(case, {"HSA_CU_MASK": "0:0-31"})
    assert np.array_equal(got["z"], ref["z"]) and np.array_equal(got["theta"], ref["theta"]) and (got["key"] == ref["key"]).all()


def test_beside_a_process_that_fills_the_gpu():
    ref, _ = _run_here(CASE)
    burner = subprocess.Popen([sys.executable, os.path.join(TOOLS, "single_run_worker.py"), "--burn", "25"])
    try:
        import time
        time.sleep(6)   # (torch import + the first products)
        got = _worker(CASE, {})
    finally:
        burner.kill()
        burner.wait()
    print("beside a saturating process: chunks repeated on events:", int(got["flag_fallbacks"]))
    assert np.array_equal(got["z"], ref["z"]) and (got["key"] == ref["key"]).all()
