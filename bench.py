#!/usr/bin/env python
"""SVGD steps/sec of the DiBS hot path on MI355X (BASELINE.json metric: d=50, n_particles=128, BGe).

    python bench.py --gpus N --steps K --warmup W [--config headline|2|3|4|5]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W
    (N > 1 without a launcher: spawns the N ranks itself)

One "step" = one SVGD step (svgd.py:226-267 / 673-721 of the reference) over all particles on synthetic data that is resident in HBM
before the timed region.  W untimed steps t = 0..W-1 of the trajectory from PRNGKey(1), then EXACTLY K steps t = W..W+K-1 timed
between barrier + synchronize fences (max over ranks).  The cost of a step depends on t (sampled parent sets shrink as the particles
sharpen), so the timed window is restored from a snapshot and measured repeatedly -- at least `--reps` times and until the timed
regions add up to `--min-seconds` of GPU time: `value` comes from the MEDIAN repetition, `rep_ms_per_step` summarises all of them.
N > 1 shards the particles over the ranks (strong scaling: total work fixed) and runs the step loop inside the engine
(dibs_engine_run_sharded): one all-gather of the packed rows per step below 512 particles, from there on the gradient rows between the
phases and the new values beside the next phase A.  RCCL by default; DIBS_COMM=ipc exchanges through mapped peer memory, so that the
ranks may share devices (`--gpus 4` on a one-GPU box runs the whole N > 1 path; its value measures processes time-sharing a GPU).
`steady_state` repeats the timed window from step --steady-t (300) of the same trajectory.
Rank 0 prints ONE JSON line (DESIGN.md "Measurement" explains every field).

--config selects the workload: `headline` (default) is BASELINE.json's metric config; 2 .. 5 are BASELINE.json configs[1..4] at their
stated sizes on the GPUs given (configs 4 / 5 are quoted on 8 GPUs; with --gpus 1 all particles sit on one device)."""
import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

S_MC, SA_MC, N_OBS = 128, 32, 100
PEAK_F32_TFLOPS = 157.3   # MI355X_MICROARCH.md: FP32 vector (packed) == FP32 MFMA peak, 64 FLOP/clk/SIMD
PEAK_BF16_TFLOPS = 2500.0  # dense bf16 MFMA peak (MI355X_MICROARCH.md)
PEAK_HBM_GBPS = 8000.0
N_SIMD, CLK_GHZ = 1024, 2.4
VALU_CYC_PER_INSTR = 4.0  # measured issue rate of plain (unpacked) VALU wave-instructions, scripts/probe/valu_rate.hip

CONFIGS = {
    "headline": dict(d=50, M=128, model="bge", label="MarginalDiBS+BGe (score-function estimator), Erdos-Renyi-2 linear-Gaussian data"),
    "2": dict(d=20, M=32, model="bge", label="BASELINE configs[1]: MarginalDiBS+BGe, d=20, 32 particles"),
    "3": dict(d=50, M=128, model="lingauss", label="BASELINE configs[2]: JointDiBS+LinearGaussian (reparam estimator), d=50, 128 particles"),
    "4": dict(d=50, M=1024, model="bge", label="BASELINE configs[3]: BGe, d=50, 1024 particles (MarginalDiBS: JointDiBS+BGe is not constructible)"),
    "5": dict(d=100, M=256, model="densenn", label="BASELINE configs[4]: JointDiBS+DenseNonlinearGaussian (5,), d=100, 256 particles, interv_mask, scale-free prior"),
}


def binary_powering_matmuls(n):
    return (n.bit_length() - 1) + (bin(n).count("1") - 1)


def self_spawn(args):
    """`python bench.py --gpus N` without a launcher: re-exec under torch.distributed.run, one rank per GPU."""
    port = 29500 + os.getpid() % 2000
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    raise SystemExit(subprocess.call(cmd, env=env))


def make_workload(name, M, rank=0, n_ranks=1, device_id=0):
    """(cfg, x, mask) of a named config: synthetic data from the product's own factory (dibs_amd/target.py), PRNGKey(0)."""
    from dibs_amd import random
    from dibs_amd._abi import make_config
    from dibs_amd.target import make_linear_gaussian_equivalent_model, make_linear_gaussian_model, make_nonlinear_gaussian_model
    c = CONFIGS[name]
    d = c["d"]
    common = dict(n_vars=d, n_particles=M, n_observations=N_OBS, n_grad_mc_samples=S_MC, n_acyclicity_mc_samples=SA_MC, rank=rank,
                  n_ranks=n_ranks, device_id=device_id)
    if c["model"] == "bge":
        data, _, _ = make_linear_gaussian_equivalent_model(key=random.PRNGKey(0), n_vars=d, graph_prior_str="er", n_observations=N_OBS)
        return make_config(**common), data.x, None
    if c["model"] == "lingauss":
        data, _, _ = make_linear_gaussian_model(key=random.PRNGKey(0), n_vars=d, graph_prior_str="er", n_observations=N_OBS)
        return make_config(joint=True, likelihood="lingauss", **common), data.x, None
    data, _, _ = make_nonlinear_gaussian_model(key=random.PRNGKey(0), n_vars=d, graph_prior_str="sf", n_observations=N_OBS)
    rng = np.random.default_rng(0)
    mask = np.zeros((N_OBS, d), np.int32)
    for r in range(0, N_OBS, 10):   # 10 intervention sets of ceil(0.1 d) nodes clamped to 0 (target.py:97-105 geometry)
        mask[r:r + 10, rng.choice(d, int(np.ceil(0.1 * d)), replace=False)] = 1
    x = np.where(mask == 1, 0.0, data.x).astype(np.float32)
    return make_config(joint=True, likelihood="densenn", nn_hidden=(5,), graph_prior="sf", has_interventions=True, **common), x, mask


def roofline_of(dom, avg_s, cfgname, M, K, eng_counters):
    """Algorithmic work of the dominant kernel per launch, priced against the FP32 peak."""
    c = CONFIGS[cfgname]
    d, H = c["d"], 5
    roof = {"kernel": dom, "avg_launch_us": avg_s * 1e6, "unit": "TFLOP/s", "peak": PEAK_F32_TFLOPS, "bound": "mfma", "traffic": None}
    if dom == "acyc":
        n_mm = binary_powering_matmuls(d - 1)
        flops = M * SA_MC * n_mm * 2 * d ** 3
        roof.update(flops_per_launch=flops, achieved=flops / avg_s / 1e12,
                    flops_model=f"M*Sa*c(d-1)*2*d^3, c({d - 1})={n_mm} matmuls of binary powering (SURVEY 8(d) F_acyc)")
        if 32 < d <= 64 and os.environ.get("DIBS_ACYC_BF16") and not os.environ.get("DIBS_ACYC_F32"):
            # A/B run: round 3's three-piece bf16 kernel (24 mantissa bits per operand)
            bf16_flops = M * SA_MC * n_mm * 6 * 2 * 64 ** 3
            roof.update(rocprof_kernel=f"k_acyc_bf<{'true' if d > 48 else 'false'}>", pipe="mfma_bf16 (3-way split operands: 6 bf16 products per f32 product)",
                        pipe_mantissa_bits=24, executed_bf16_tflops=bf16_flops / avg_s / 1e12, peak_bf16_tflops=PEAK_BF16_TFLOPS,
                        frac_of_bf16_peak=bf16_flops / avg_s / 1e12 / PEAK_BF16_TFLOPS)
        elif 32 < d <= int(os.environ.get("DIBS_ACYC_HFW_MAX", "112")) and not os.environ.get("DIBS_ACYC_F32") and not (d > 64 and os.environ.get("DIBS_ACYC_BF16")):
            # float products evaluated on the f16 matrix pipe with two block-scaled pieces per operand (3 f16 MFMAs of 16 cycles per
            # 16 x 16 x 32 block, kernels_acyc_f16.h).  `achieved` / `frac` price the ALGORITHMIC float flops against the FP32 peak; the f16
            # flops the kernel actually issues (16-padded tiles, 32-padded contraction, 3 products) against the dense 16-bit peak next to it.
            nt = 4 if d <= 64 else (d + 15) // 16
            rows = (64 if d > 48 else 48) if d <= 64 else 16 * nt
            kdim = 64 if d <= 64 else 32 * ((nt + 1) // 2)
            f16_flops = M * SA_MC * n_mm * 3 * 2 * rows * rows * kdim
            roof.update(rocprof_kernel=(f"k_acyc_hf<{'true' if d > 48 else 'false'}, 3>" if d <= 64 else f"k_acyc_hfw<{nt}>"),
                        pipe="mfma_f16 (2 block-scaled pieces per operand: 3 f16 products per f32 product)", pipe_mantissa_bits=22,
                        executed_f16_tflops=f16_flops / avg_s / 1e12, peak_f16_tflops=PEAK_BF16_TFLOPS,
                        frac_of_f16_peak=f16_flops / avg_s / 1e12 / PEAK_BF16_TFLOPS)
        elif 80 < d <= 112 and not os.environ.get("DIBS_ACYC_F32"):
            nt = (d + 15) // 16   # three-piece bf16 kernel (DIBS_ACYC_HFW_MAX=80 / DIBS_ACYC_BF16=1; the default there is the f16 kernel above since round 5)
            bf16_flops = M * SA_MC * n_mm * 6 * 2 * (16 * nt) ** 3
            roof.update(rocprof_kernel=f"k_acyc_bfw<{nt}>", pipe="mfma_bf16 (3-way split operands: 6 bf16 products per f32 product)", pipe_mantissa_bits=24,
                        executed_bf16_tflops=bf16_flops / avg_s / 1e12, peak_bf16_tflops=PEAK_BF16_TFLOPS,
                        frac_of_bf16_peak=bf16_flops / avg_s / 1e12 / PEAK_BF16_TFLOPS)
        else:
            roof.update(rocprof_kernel=f"k_acyc<{(d + 15) // 16}, true>", pipe="mfma_f32", pipe_mantissa_bits=24)
    elif dom == "bge_big":
        # VALU kernel: the work actually executed (the reference's dense 2 d^3/3-per-determinant count is not what runs:
        # only R[pa + j] is factorised).  Priced against the f32 vector peak.
        flops = float(eng_counters[0]) / K
        roof.update(rocprof_kernel="k_bge_chol", pipe="valu_f32", flops_per_launch=flops, achieved=flops / avg_s / 1e12,
                    flops_model="executed Cholesky flops sum (l+1)^3/3, counted on the device; f32 VECTOR peak (bound 'mfma' = FP32 compute, "
                                "the pipe says which unit)", dense_equivalent_flops=M * S_MC * d * 2 * d ** 3 / 3.0)
    elif dom == "bge_nodes":
        # Threefry sampling: integer VALU work, no flops.  Bound = VALU issue: 67 instructions per Threefry-2x32 call.
        calls = M * (S_MC // 2) * d * d
        floor_s = calls * 67 / 64 / N_SIMD * VALU_CYC_PER_INSTR / (CLK_GHZ * 1e9)
        roof.update(rocprof_kernel="k_bge_sample<4, true>", pipe="valu_int", unit="fraction of VALU issue", peak=1.0, achieved=floor_s / avg_s,
                    flops_model="Threefry calls * 67 VALU instr / (1024 SIMD * 1 instr per 4 clk at 2.4 GHz) / measured time")
    elif dom in ("lin_logprobs", "lin_grad"):
        flops = 2 * M * S_MC * 2 * N_OBS * d * d   # both estimators' forward products x (G o theta): half of SURVEY 8(d) F_lik(LinG)
        roof.update(rocprof_kernel="k_lin_logprobs_pair" if dom == "lin_logprobs" else "k_lin_grad", pipe="mfma_f32", flops_per_launch=flops,
                    achieved=flops / avg_s / 1e12, flops_model="2 estimators * M*S*2*N*d^2 (forward products of SURVEY 8(d) F_lik(LinG))")
        if dom == "lin_logprobs" and 32 < d <= 64:
            # k_lin_logprobs_hf: the float products run on the f16 matrix pipe with two block-scaled pieces per operand (as k_acyc_hf above):
            # `achieved` / `frac` price the algorithmic float flops against the FP32 peak, the f16 flops actually issued (row tiles of 16
            # observations, 64-padded contraction, 16-wide column tiles, 3 products) against the dense f16 peak next to it.
            f16_flops = 2 * M * S_MC * 3 * 2 * (16 * ((N_OBS + 15) // 16)) * 64 * (64 if d > 48 else 48)
            roof.update(rocprof_kernel="k_lin_logprobs_hf", pipe="mfma_f16 (2 block-scaled pieces per operand: 3 f16 products per f32 product)",
                        executed_f16_tflops=f16_flops / avg_s / 1e12, peak_f16_tflops=PEAK_BF16_TFLOPS,
                        frac_of_f16_peak=f16_flops / avg_s / 1e12 / PEAK_BF16_TFLOPS)
    elif dom in ("nn_theta", "nn_z"):
        flops = M * S_MC * d * (2 * N_OBS * d * H + 2 * N_OBS * H)
        roof.update(rocprof_kernel=("k_nn_logprobs_hx" if d > 64 else "k_nn_logprobs_hf") + " + k_nn_grad", pipe="mfma_f16 (log-probs: 2 block-scaled pieces per operand) + mfma_f32 (gradients of the samples with non-zero weight)",
                    flops_per_launch=flops, achieved=flops / avg_s / 1e12,
                    flops_model="M*S*d*(2NdH + 2NH): forward pass of one estimator (a third of SURVEY 8(d) F_lik(NN) per estimator)")
    elif dom in ("phi_update", "kmat"):
        # particle-coupling kernels (dominant only with many particles on one GPU): vector-ALU kernels on packed f32
        Dp = 2 * d * d + (d * d if c["model"] == "lingauss" else (d * (d * H + H + H + 1) if c["model"] == "densenn" else 0))
        flops = (6.0 if dom == "phi_update" else 1.5) * M * M * Dp
        roof.update(rocprof_kernel="k_phi_update<TA>" if dom == "phi_update" else "k_kmat", pipe="valu_f32 (v_pk_fma_f32)", flops_per_launch=flops,
                    achieved=flops / avg_s / 1e12,
                    flops_model=("3*M*M*(D+P) FMA: phi_a = -(1/M) sum_b [k(b,a) grad_b - (2/h) k(b,a) (x_b - x_a)] (SURVEY 8(d) F_kern; z and theta segments "
                                 "are separate launches: avg_launch_us is their mean, flops their sum / launches)") if dom == "phi_update" else
                                "3*M*M*(D+P)/2 flop: squared distances by direct differences, upper triangle (z and theta segments are separate launches)")
        if c["model"] != "bge":
            roof["flops_per_launch"] = flops / 2.0
            roof["achieved"] = roof["flops_per_launch"] / avg_s / 1e12
        if dom == "phi_update" and M >= 256:
            # many particles: the transform runs as one GEMM on the f32 MFMA ([ks | -kr] x [grad ; x], k_phi_gemm): 2 * M * 2M * len flop per
            # launch, i.e. 4 (not 6) flop per (a, b, i) -- the (x_b - x_a) subtraction became a rank-one row-sum term
            roof.update(rocprof_kernel="k_phi_gemm", pipe="mfma_f32", flops_per_launch=roof["flops_per_launch"] * 4.0 / 6.0,
                        achieved=roof["flops_per_launch"] * 4.0 / 6.0 / avg_s / 1e12,
                        flops_model="4*M*M*len: [kz + kt | -(2/h) kseg] (M x 2M) times [grad ; x] (2M x len) on v_mfma_f32_16x16x4_f32, plus the rank-one "
                                    "row-sum term (SURVEY 8(d) F_kern regrouped)")
    else:
        roof.update(pipe="valu_f32", achieved=None)
    roof["frac"] = roof["achieved"] / roof["peak"] if roof.get("achieved") else None
    return roof


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="headline", choices=list(CONFIGS))
    ap.add_argument("--reps", type=int, default=7, help="minimum repetitions of the timed window (median reported)")
    ap.add_argument("--min-seconds", type=float, default=8.0, help="repeat the timed window until the timed regions add up to this much")
    ap.add_argument("--n-particles", type=int, default=None, help="override the particle count of the config")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--steady-t", type=int, default=300,
                    help="also time the same K-step window from this step of the trajectory (sample() runs thousands of steps: the sampled parent "
                         "sets shrink as the particles sharpen and a step gets cheaper); 0: skip.  Reported as `steady_state`, never as `value`")
    ap.add_argument("--dist-smoke", action="store_true",
                    help="with --gpus 1: run the sharded code path (process group of ONE rank, real RCCL calls, overlapped exchange) instead of "
                         "dibs_engine_run -- exercises on one GPU what --gpus N > 1 executes")
    args = ap.parse_args()
    K, W, N = args.steps, args.warmup, args.gpus
    M = args.n_particles or CONFIGS[args.config]["M"]
    if args.config == "headline" and M == 1024:
        args.config = "4"
    if N > 1 and "WORLD_SIZE" not in os.environ:
        self_spawn(args)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != N:
        raise SystemExit(f"--gpus {N} but WORLD_SIZE={world}")

    import torch
    from dibs_amd import random
    from dibs_amd.engine import Engine

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
    # DIBS_COMM=ipc: the exchange through mapped peer memory (dibs_engine_comm_init_ipc) instead of RCCL -- the ranks may then SHARE devices
    # (rank r on device r mod #devices): `--gpus 4` on a one-GPU box executes the whole N > 1 path, rank != 0 included.  The harness's own
    # barrier / max-over-ranks go through gloo then (RCCL refuses a group with a duplicate GPU).
    comm_ipc = os.environ.get("DIBS_COMM", "rccl").lower() == "ipc"
    dev_id = local_rank % torch.cuda.device_count() if comm_ipc else local_rank
    torch.cuda.set_device(dev_id)
    sharded = N > 1 or args.dist_smoke
    dist = None
    if sharded:
        os.environ.setdefault("MASTER_PORT", str(29500 + os.getpid() % 2000))
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if comm_ipc:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    ctl_dev = "cpu" if comm_ipc else "cuda"   # where the harness's own small reductions live

    def fence():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    loop_note = {}

    def measure(cfgname, M_, reps_min, min_seconds):
        """Engine + runner for one workload; returns (engine, run, snapshot at t = W, per-repetition seconds)."""
        cfg, x, mask = make_workload(cfgname, M_, rank, N, dev_id)
        # N > 1: the step loop runs INSIDE the engine (dibs_engine_run_sharded): RCCL communicators created by libdibs_hip.so from unique
        # ids (torch.distributed only carries those bytes and provides the barrier / max-over-ranks of this harness), ncclAllGather issued
        # on the engine's own streams.  DIBS_BENCH_TORCH_LOOP=1: the Python-driven loop over torch collectives (the test harness).
        torch_loop = sharded and bool(os.environ.get("DIBS_BENCH_TORCH_LOOP"))
        # exchange protocol: ONE all-gather of the packed rows per step while the payload is small (128 particles: 0.6 - 2.5 MB per rank, the
        # collective is latency-bound and the fused kernel matrix of the packed protocol is cheaper than a side stream); the overlapped
        # exchange (values beside phase A, only gradients between the phases) from 512 particles on, where it takes 20 MB and the
        # 128 x 1024 kernel slab off the critical path.  DIBS_BENCH_EXCHANGE=packed|overlapped overrides.
        ex = os.environ.get("DIBS_BENCH_EXCHANGE", "overlapped" if M_ >= 512 else "packed")
        overlapped = ex == "overlapped"
        tstream = torch.cuda.Stream() if sharded else None

        def new_engine(on_torch_stream):
            e_ = Engine(cfg, stream=tstream.cuda_stream if on_torch_stream else None)
            e_.set_data(x, mask)
            e_.init_particles(random.PRNGKey(1))
            return e_

        eng = new_engine(torch_loop)
        if sharded and not torch_loop:
            # the in-engine communicator has to come up on EVERY rank; if it does not on any (an RCCL build that refuses the ids, a missing
            # symbol), all ranks agree on the Python-driven loop over torch.distributed -- the same HIP step, the collective issued by torch
            from dibs_amd.distributed import init_native_comm
            # Two agreements: (1) a LOCAL probe -- librccl loads and exports every symbol the engine binds (drawing a unique id touches all of
            # that and no other rank) -- agreed on BEFORE anybody enters the blocking ncclCommInitRank, so that a rank whose library is
            # unusable cannot leave the others stuck inside the bring-up; (2) the bring-up itself.  A failure INSIDE ncclCommInitRank on a
            # subset of the ranks is fatal by RCCL's own rules (the other ranks block until its timeout aborts the job) -- not recoverable here.
            err = ""
            if comm_ipc:
                from dibs_amd.distributed import init_ipc_comm
                init_ipc_comm(eng)   # (no fallback: the torch loop needs RCCL, which cannot serve ranks that share a device)
                flag = torch.tensor([1], dtype=torch.int32)
            else:
                try:
                    if os.environ.get("DIBS_BENCH_FAIL_NATIVE"):   # (tests: exercise the agreement + fallback below)
                        raise RuntimeError("DIBS_BENCH_FAIL_NATIVE is set")
                    eng.comm_unique_ids(1)
                except Exception as ex_:   # noqa: BLE001
                    err = f"{type(ex_).__name__}: {ex_}"
                flag = torch.tensor([0 if err else 1], dtype=torch.int32, device="cuda")
                dist.all_reduce(flag, op=dist.ReduceOp.MIN)
                if int(flag.item()) == 1:
                    try:
                        init_native_comm(eng, None, 2 if overlapped else 1)
                    except Exception as ex_:   # noqa: BLE001
                        err = f"{type(ex_).__name__}: {ex_}"
                    flag = torch.tensor([0 if err else 1], dtype=torch.int32, device="cuda")
                    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if int(flag.item()) == 0:
                if rank == 0:
                    print(f"bench: in-engine RCCL communicator failed ({err or 'on another rank'}); using the torch.distributed loop", file=sys.stderr)
                eng.close()
                torch_loop = True
                loop_note["loop"] = "torch.distributed (in-engine communicator failed: " + (err or "on another rank") + ")"
                eng = new_engine(True)
        if sharded and not torch_loop:
            n_el = eng.plane_elems_per_rank() if overlapped else eng.gather_elems_per_rank()
            send = torch.zeros(n_el, device="cuda")            # (only for the stand-alone timing of the collective below)
            recv = torch.zeros(n_el * N, device="cuda")

            def run(t0, n):
                eng.run_sharded(t0, n, overlapped)

            def restore(snap_):
                eng.set_state(**snap_)
                eng.run_sharded(W, 0, overlapped)   # untimed: gathers the values of the restored state (in a run they travel during the step before)
        elif sharded:
            from dibs_amd.distributed import OverlapBuffers, refresh_values, run_sharded_overlapped
            with torch.cuda.stream(tstream):
                buf = OverlapBuffers(eng, N, torch.device("cuda", local_rank), torch.float32)
            send, recv = buf.gsend, buf.grads

            def run(t0, n):
                with torch.cuda.stream(tstream):
                    # per step: phase A (+ kernel-matrix slab from the values gathered on the side stream) -> all-gather of the gradient
                    # rows (RCCL) -> phase B -> export + all-gather of the new values on the side stream, beside the next phase A
                    run_sharded_overlapped(eng, t0, n, buf, always_collective=True)

            def restore(snap_):
                eng.set_state(**snap_)
                with torch.cuda.stream(tstream):
                    refresh_values(eng, buf, always_collective=True)   # untimed: the values of the restored state (in a run they were gathered during the step before)
        else:
            send = recv = None

            def run(t0, n):
                eng.run(t0, n)

            def restore(snap_):
                eng.set_state(**snap_)
        run(0, W)                       # untimed warm-up: steps 0 .. W-1 of the trajectory
        fence()
        snap = {k_: v for k_, v in eng.get_state().items() if v is not None}   # state at t = W (this rank's particles)
        rep_s, total = [], 0.0
        while len(rep_s) < max(reps_min, 1) or (total < min_seconds and len(rep_s) < 2000):
            if rep_s:
                restore(snap)           # untimed: back to t = W
            fence()
            t_begin = time.perf_counter()
            run(W, K)                   # timed: steps W .. W+K-1
            fence()
            el = time.perf_counter() - t_begin
            if dist is not None:
                tt = torch.tensor([el], dtype=torch.float64, device=ctl_dev)
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                el = float(tt.item())
            rep_s.append(el)
            total += el
        return eng, run, snap, rep_s, (cfg, x, mask), (tstream, send, recv, restore)

    eng, run, snap, rep_s, (cfg, x, mask), (tstream, send, recv, restore) = measure(args.config, M, args.reps, args.min_seconds)
    elapsed = float(np.median(rep_s))
    steps_per_s = K / elapsed
    c = CONFIGS[args.config]
    d = c["d"]
    P = eng.P
    out = {
        "metric": f"SVGD steps/sec (d={d}, n_particles={M}, {'BGe' if c['model'] == 'bge' else c['model']})", "value": steps_per_s,
        "unit": "steps/s", "n_gpus": N, "steps": K, "warmup": W, "ms_per_step": 1e3 * elapsed / K, "higher_is_better": True,
        "scaling": "strong" if N > 1 else "n/a", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": c["label"], "name": args.config, "n_vars": d, "n_particles": M, "n_observations": N_OBS,
                   "n_grad_mc_samples": S_MC, "n_acyclicity_mc_samples": SA_MC,
                   "timed_steps": f"t={W}..{W + K - 1} of one trajectory from PRNGKey(1)",
                   "parallelism": (f"particles sharded over {N} rank(s), step loop and {'mapped-memory' if comm_ipc else 'RCCL'} all-gathers inside the engine (dibs_engine_run_sharded); "
                                   + ("ONE all-gather of the packed rows [z | grad_z | theta | grad_theta] per step" if M < 512 and os.environ.get("DIBS_BENCH_EXCHANGE") != "overlapped" else
                                      "gradients all-gathered between the phases, values beside phase A")) if sharded else "single GPU"},
        "reps": len(rep_s), "timed_seconds_total": float(np.sum(rep_s)),
        "rep_ms_per_step": {"median": 1e3 * elapsed / K, "min": 1e3 * min(rep_s) / K, "max": 1e3 * max(rep_s) / K,
                            "first5": [1e3 * r / K for r in rep_s[:5]]},
        "rep_spread": (float(np.percentile(rep_s, 90)) - float(np.percentile(rep_s, 10))) / elapsed,
    }
    if comm_ipc and sharded:
        out["n_devices"] = min(N, torch.cuda.device_count())   # (ranks share devices: n_gpus is the number of RANKS here)
    if loop_note:
        out["config"]["parallelism"] = f"particles sharded over {N} rank(s), step loop driven from Python, collectives by " + loop_note["loop"]

    if sharded:
        # ---- diagnosis of a sharded run: per-rank kernel timers, the collective alone, config 4 beside the strong-scaling headline ----
        restore(snap)
        eng.set_profiling(True)
        eng.reset_timers()
        run(W, K)
        fence()
        mine = {k_: ms / K * 1e3 for k_, (ms, n_) in eng.timers().items()}
        eng.set_profiling(False)
        allk = [None] * N
        dist.all_gather_object(allk, mine)
        ag_us = None
        if not comm_ipc:   # (the collective alone through torch.distributed: RCCL only)
            with torch.cuda.stream(tstream):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                for _ in range(5):
                    dist.all_gather_into_tensor(recv, send)
                e0.record(tstream)
                for _ in range(50):
                    dist.all_gather_into_tensor(recv, send)
                e1.record(tstream)
            fence()
            ag_us = e0.elapsed_time(e1) / 50 * 1e3
        out["sharded"] = {
            "comm": ("mapped peer memory (dibs_engine_comm_init_ipc): %d rank(s) on %d device(s)" % (N, min(N, torch.cuda.device_count()))) if comm_ipc
                    else "RCCL (dibs_engine_comm_init)",
            "kernel_us_per_step_by_rank": allk, "allgather_us": ag_us, "allgather_bytes_per_rank": int(send.numel() * 4),
            "exchange": "allgather_us / allgather_bytes_per_rank: the collective on the critical path of a step, timed alone through torch.distributed "
                        "(packed protocol: the rows [z | grad_z | theta | grad_theta]; overlapped protocol: the gradient rows -- the values travel on a "
                        "side stream beside phase A)",
            "strong_scaling_bound": "128 particles: a rank's step is a chain of dependent launches of 7-20 us that shrink little with the shard "
                                    "(profiles/round4_shard_scaling.txt: 194 / 151 / 108 / 96 us per rank-step at 1/2/4/8 ranks on one GPU in the "
                                    "in-engine loop, before the collective) => <= 2.0x at 8 GPUs; the >= 6x of north_star needs per-rank work >> "
                                    "launch latency, i.e. config 4 (1024 particles, 128 per rank)"}
        if args.config == "headline":
            eng.close()
            e4, _, _, rep4, _, _ = measure("4", 1024, 3, 0.5)
            el4 = float(np.median(rep4))
            out["config4"] = {"metric": "SVGD steps/sec (d=50, n_particles=1024, BGe)", "value": K / el4, "ms_per_step": 1e3 * el4 / K,
                              "n_gpus": N, "particles_per_rank": 1024 // N, "scaling": "weak vs the 1-GPU headline (128 particles per rank at 8 GPUs)"}
            eng = e4

    if not sharded and args.steady_t > W + K:
        # ---- the same K-step window late in the trajectory (untimed run-up from the snapshot at t = W, then the window repeated from a snapshot)
        eng.set_state(**snap)
        eng.run(W, args.steady_t - W)
        snap_late = {k_: v for k_, v in eng.get_state().items() if v is not None}
        late = []
        while len(late) < 5 or (sum(late) < 1.0 and len(late) < 500):
            eng.set_state(**snap_late)
            fence()
            t_begin = time.perf_counter()
            eng.run(args.steady_t, K)
            fence()
            late.append(time.perf_counter() - t_begin)
        el_late = float(np.median(late))
        out["steady_state"] = {"value": K / el_late, "unit": "steps/s", "ms_per_step": 1e3 * el_late / K, "reps": len(late),
                               "timed_steps": f"t={args.steady_t}..{args.steady_t + K - 1} of the same trajectory"}

    if rank == 0 and not sharded:
        # ---- roofline of the dominant kernel: the same K steps replayed with per-kernel HIP events on the engine's stream ----
        eng.set_state(**snap)
        eng.set_profiling(True)
        eng.reset_timers()
        eng.run(W, K)
        timers = eng.timers()
        counters = eng.counters()   # [0]: executed Cholesky flops (sum n^3/3 over the queued problems) of the profiled steps
        eng.set_profiling(False)
        total_ms = sum(ms for ms, _ in timers.values())
        dom = max(timers, key=lambda k_: timers[k_][0])
        dom_ms, dom_n = timers[dom]
        roof = roofline_of(dom, dom_ms / dom_n * 1e-3, args.config, M, K, counters)
        roof.update(launches=dom_n, share_of_step=dom_ms / total_ms,
                    duration="kernel alone on the GPU (the step is serialised while timing); in the timed region the acyclicity kernel runs on "
                             "its second stream beside the likelihood kernels, see kernel_us_per_step_concurrent / frac_concurrent")
        for tag in ("round6", "round5", "round4", "round3", "round2"):   # separate rocprofv3 --pmc passes of this command (scripts/collect_profiles.sh)
            suffix = "" if args.config == "headline" else f"_cfg{args.config}"
            pmc = os.path.join(ROOT, "profiles", f"{tag}{suffix}_pmc_hbm.json")
            if os.path.exists(pmc):
                rec = json.load(open(pmc)).get(dom)
                if rec:
                    roof["traffic"] = rec["hbm_bytes_per_launch"]
                    roof["traffic_source"] = f"profiles/{tag}{suffix}_pmc_hbm.json (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE, KiB -> bytes)"
                break
        out["kernel_us_per_step"] = {k_: ms / K * 1e3 for k_, (ms, n_) in timers.items()}
        # the same replay with the production schedule (acyclicity kernel on the second stream, timed there): overlapping kernels share the GPU
        eng.set_state(**snap)
        eng.set_profiling(2)
        eng.reset_timers()
        eng.run(W, K)
        conc = {k_: (ms, n_) for k_, (ms, n_) in eng.timers().items()}
        out["kernel_us_per_step_concurrent"] = {k_: ms / K * 1e3 for k_, (ms, n_) in conc.items()}
        eng.set_profiling(False)
        if dom in conc and roof.get("achieved"):
            roof["avg_launch_us_concurrent"] = conc[dom][0] / conc[dom][1] * 1e3
            roof["frac_concurrent"] = roof["frac"] * roof["avg_launch_us"] / roof["avg_launch_us_concurrent"]
        out["roofline"] = roof
        bytes_step = 16.0 * M * (2 * d * d + P)
        out["hbm_algorithmic"] = {"bytes_per_step": bytes_step, "achieved_GBps": bytes_step * steps_per_s / 1e9,
                                  "frac_of_8TBps": bytes_step * steps_per_s / 1e9 / PEAK_HBM_GBPS}
        if c["model"] == "bge":
            out["bge_executed_gflop_per_step"] = float(counters[0]) / K / 1e9

        if not args.no_cpu_baseline:
            # ---- CPU baseline: the oracle's C port on this box's host cores, bounded sample of the SAME window (from t = W) ----
            from oracle.c_oracle import COracle
            from dibs_amd._abi import DibsConfig
            cores = min(os.cpu_count() or 1, M)   # OpenMP THREADS used (os.cpu_count() counts hardware threads, not physical cores); over particles: more threads than particles are idle
            cfg1 = DibsConfig.from_buffer_copy(cfg)

            def cpu(prec, n_steps, mode):
                co = COracle(prec)
                st = {k_: (v.astype(co.real) if v.dtype.kind == "f" else v.copy()) for k_, v in snap.items()}
                st.setdefault("theta", None), st.setdefault("v_theta", None)
                t0 = time.perf_counter()
                co.run(cfg1, x, mask, st, W, n_steps, bge_mode=mode, n_threads=cores)
                return n_steps / (time.perf_counter() - t0)

            probe = cpu("f32", 1, 0)                       # one step to size the sample: about 20 s of CPU work, 2 .. 20 steps
            n_cpu = int(min(max(20.0 * probe, 2), 20))
            faithful = cpu("f32", n_cpu, 0)   # the reference's algorithm (BGe: masked d x d slogdet (LU) per node), float32
            out["cpu_baseline"] = {"value": faithful, "unit": "steps/s", "cores": cores, "kind": "port",
                                   "sample": f"{n_cpu} steps (t={W}..{W + n_cpu - 1}) of the same workload from the same state, f32 C port of the "
                                             "reference algorithm" + (" (masked d x d LU slogdet per node, func.py:128-145)" if c["model"] == "bge" else "")
                                             + f", OpenMP over particles on {cores} threads (hardware threads of the host, not physical cores)", "threads": cores}
            if c["model"] == "bge":
                out["cpu_baseline_compact_f32"] = {"value": cpu("f32", n_cpu, 1), "unit": "steps/s", "cores": cores, "kind": "port",
                                                   "sample": f"{n_cpu} steps, same port with the GPU's compact parent-set Cholesky"}
            out["cpu_baseline_f64"] = {"value": cpu("f64", 2, 0), "unit": "steps/s", "cores": cores, "kind": "port",
                                       "sample": "2 steps, f64 build of the reference algorithm"}
    eng.close()
    if rank == 0:
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
