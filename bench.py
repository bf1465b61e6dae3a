#!/usr/bin/env python
"""SVGD steps/sec of the DiBS hot path on MI355X (BASELINE.json metric: d=50, n_particles=128, BGe).

    python bench.py --gpus N --steps K --warmup W        (N > 1 without a launcher: spawns the N ranks itself)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One "step" = one SVGD step (svgd.py:226-267 of the reference) over all 128 particles on synthetic ER-2 linear-Gaussian
data that is resident in HBM before the timed region.  W untimed steps t = 0..W-1 of the trajectory from PRNGKey(1), then
EXACTLY K steps t = W..W+K-1 timed between barrier + synchronize fences (max over ranks).  The cost of a step depends on t
(sampled parent sets shrink as the particles sharpen), so the timed window is restored from a snapshot and measured
`--reps` times: `value` comes from the MEDIAN repetition, all repetitions are listed in `rep_ms_per_step`.
N > 1 shards the particles over the ranks (strong scaling: total work fixed) with one RCCL all-gather of [z | grad_z] per
step.  Rank 0 prints ONE JSON line (DESIGN.md "Measurement" explains every field)."""
import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

D_VARS, N_PARTICLES, N_OBS, S_MC, SA_MC = 50, 128, 100, 128, 32
PEAK_F32_TFLOPS = 157.3   # MI355X_MICROARCH.md: FP32 vector (packed) == FP32 MFMA peak, 64 FLOP/clk/SIMD
PEAK_BF16_TFLOPS = 2500.0  # dense bf16 MFMA peak (MI355X_MICROARCH.md)
PEAK_HBM_GBPS = 8000.0
N_SIMD, CLK_GHZ = 1024, 2.4
VALU_CYC_PER_INSTR = 4.0  # measured issue rate of plain (unpacked) VALU wave-instructions, scripts/probe/valu_rate.hip


def binary_powering_matmuls(n):
    return (n.bit_length() - 1) + (bin(n).count("1") - 1)


def self_spawn(args):
    """`python bench.py --gpus N` without a launcher: re-exec under torch.distributed.run, one rank per GPU."""
    port = 29500 + os.getpid() % 2000
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    raise SystemExit(subprocess.call(cmd, env=env))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--reps", type=int, default=7, help="repetitions of the timed window (median reported)")
    ap.add_argument("--n-particles", type=int, default=N_PARTICLES,
                    help="128 = BASELINE.json's metric; 1024 = config 4 (particles sharded over the ranks)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()
    K, W, N, M = args.steps, args.warmup, args.gpus, args.n_particles
    if N > 1 and "WORLD_SIZE" not in os.environ:
        self_spawn(args)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != N:
        raise SystemExit(f"--gpus {N} but WORLD_SIZE={world}")

    import torch
    from dibs_amd import random
    from dibs_amd._abi import make_config
    from dibs_amd.engine import Engine
    from dibs_amd.target import make_linear_gaussian_equivalent_model

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dist = None
    if N > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    data, _, _ = make_linear_gaussian_equivalent_model(key=random.PRNGKey(0), n_vars=D_VARS, graph_prior_str="er",
                                                       n_observations=N_OBS)
    cfg = make_config(n_vars=D_VARS, n_particles=M, n_observations=N_OBS, n_grad_mc_samples=S_MC,
                      n_acyclicity_mc_samples=SA_MC, rank=rank, n_ranks=N, device_id=local_rank)
    # N > 1: engine kernels and the RCCL all-gather share one dedicated (non-default) torch stream
    tstream = torch.cuda.Stream() if N > 1 else None
    eng = Engine(cfg, stream=tstream.cuda_stream if tstream is not None else None)
    eng.set_data(data.x)
    eng.init_particles(random.PRNGKey(1))

    if N > 1:
        from dibs_amd.distributed import make_buffers, run_sharded
        with torch.cuda.stream(tstream):
            send, recv = make_buffers(eng, N, torch.device("cuda", local_rank), torch.float32)

        def run(t0, n):
            with torch.cuda.stream(tstream):
                run_sharded(eng, t0, n, send, recv)   # phase A -> one all-gather (RCCL) -> phase B, per step
    else:
        def run(t0, n):
            eng.run(t0, n)

    def fence():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    run(0, W)                       # untimed warm-up: steps 0 .. W-1 of the trajectory
    fence()
    snap = {k_: v for k_, v in eng.get_state().items() if v is not None}   # state at t = W (this rank's particles)
    rep_s = []
    for rep in range(max(args.reps, 1)):
        if rep:
            eng.set_state(**snap)   # untimed: back to t = W
        fence()
        t_begin = time.perf_counter()
        run(W, K)                   # timed: steps W .. W+K-1
        fence()
        el = time.perf_counter() - t_begin
        if dist is not None:
            tt = torch.tensor([el], dtype=torch.float64, device="cuda")
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            el = float(tt.item())
        rep_s.append(el)
    elapsed = float(np.median(rep_s))
    steps_per_s = K / elapsed

    out = {
        "metric": f"SVGD steps/sec (d={D_VARS}, n_particles={M}, BGe)", "value": steps_per_s, "unit": "steps/s", "n_gpus": N,
        "steps": K, "warmup": W, "ms_per_step": 1e3 * elapsed / K, "higher_is_better": True,
        "scaling": "strong" if N > 1 else "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "MarginalDiBS+BGe (score-function estimator), Erdos-Renyi-2 linear-Gaussian data",
                   "n_vars": D_VARS, "n_particles": M, "n_observations": N_OBS, "n_grad_mc_samples": S_MC,
                   "n_acyclicity_mc_samples": SA_MC, "timed_steps": f"t={W}..{W + K - 1} of one trajectory from PRNGKey(1)",
                   "parallelism": f"particles sharded over {N} rank(s), 1 all-gather/step" if N > 1 else "single GPU"},
        "reps": len(rep_s), "rep_ms_per_step": [1e3 * r / K for r in rep_s],
        "rep_spread": (max(rep_s) - min(rep_s)) / elapsed,
    }

    if rank == 0 and N == 1:
        # ---- roofline of the dominant kernel: the same K steps replayed with per-kernel HIP events on the engine's stream ----
        eng.set_state(**snap)
        eng.set_profiling(True)
        eng.reset_timers()
        eng.run(W, K)
        timers = eng.timers()
        bge_flops = float(eng.counters()[0]) / K   # executed Cholesky flops per step (sum n^3/3 over the queued problems)
        eng.set_profiling(False)
        total_ms = sum(ms for ms, _ in timers.values())
        dom = max(timers, key=lambda k_: timers[k_][0])
        dom_ms, dom_n = timers[dom]
        avg_s = dom_ms / dom_n * 1e-3
        # SURVEY.md 8(d) algorithmic flop counts per step (= per launch: each kernel is launched once per step)
        acyc_flops = M * SA_MC * binary_powering_matmuls(D_VARS - 1) * 2 * D_VARS ** 3   # F_acyc
        dense_bge = M * S_MC * D_VARS * 2 * D_VARS ** 3 / 3.0                            # F_lik(BGe), reference's dense count
        names = {"acyc": "k_acyc_bf<true>", "bge_nodes": "k_bge_sample<4, true>", "bge_big": "k_bge_chol<true, false>"}
        roof = {"kernel": dom, "rocprof_kernel": names.get(dom, "k_" + dom), "avg_launch_us": avg_s * 1e6, "launches": dom_n,
                "share_of_step": dom_ms / total_ms, "unit": "TFLOP/s", "peak": PEAK_F32_TFLOPS, "traffic": None}
        if dom == "acyc":
            # float products evaluated on the bf16 matrix pipe with three-way split operands (6 bf16 MFMAs per float product block,
            # kernels_acyc_bf16.h).  `achieved` / `frac` price the ALGORITHMIC float flops against the FP32 peak; the bf16 flops the
            # kernel actually issues (64-padded tiles, 6 products) are priced against the dense bf16 peak next to it.
            n_mm = binary_powering_matmuls(D_VARS - 1)
            bf16_flops = M * SA_MC * n_mm * 6 * 2 * 64 ** 3
            roof.update(bound="mfma", pipe="mfma_bf16 (3-way split operands: 6 bf16 products per f32 product)", flops_per_launch=acyc_flops,
                        achieved=acyc_flops / avg_s / 1e12,
                        flops_model="M*Sa*c(d-1)*2*d^3, c(49)=7 matmuls of binary powering (SURVEY 8(d) F_acyc)",
                        executed_bf16_tflops=bf16_flops / avg_s / 1e12, peak_bf16_tflops=PEAK_BF16_TFLOPS,
                        frac_of_bf16_peak=bf16_flops / avg_s / 1e12 / PEAK_BF16_TFLOPS,
                        duration="kernel alone on the GPU (the step is serialised while timing); in the timed region it runs on its second "
                                 "stream beside the BGe kernels, see kernel_us_per_step_concurrent")
        elif dom == "bge_big":
            # VALU kernel: the work actually executed (the reference's dense 2 d^3/3-per-determinant count is not what runs:
            # only R[pa + j] is factorised).  Priced against the f32 vector peak.
            roof.update(bound="mfma", pipe="valu_f32", flops_per_launch=bge_flops, achieved=bge_flops / avg_s / 1e12,
                        flops_model="executed Cholesky flops sum (l+1)^3/3, counted on the device; f32 VECTOR peak (bound 'mfma' = "
                                    "FP32 compute, the pipe says which unit)", dense_equivalent_flops=dense_bge)
        elif dom == "bge_nodes":
            # Threefry sampling: integer VALU work, no flops.  Bound = VALU issue: 67 instructions per Threefry-2x32 call.
            calls = M * (S_MC // 2) * D_VARS * D_VARS
            floor_s = calls * 67 / 64 / N_SIMD * VALU_CYC_PER_INSTR / (CLK_GHZ * 1e9)
            roof.update(bound="mfma", pipe="valu_int", unit="fraction of VALU issue", peak=1.0, achieved=floor_s / avg_s,
                        flops_model="Threefry calls * 67 VALU instr / (1024 SIMD * 1 instr per 4 clk at 2.4 GHz) / measured time")
        else:
            roof.update(bound="mfma", pipe="valu_f32", achieved=None)
        roof["frac"] = roof["achieved"] / roof["peak"] if roof.get("achieved") else None
        for tag in ("round2", "round1"):   # separate rocprofv3 --pmc passes of this command (scripts/collect_profiles.sh)
            pmc = os.path.join(ROOT, "profiles", f"{tag}_pmc_hbm.json")
            if os.path.exists(pmc):
                rec = json.load(open(pmc)).get(dom)
                if rec:
                    roof["traffic"] = rec["hbm_bytes_per_launch"]
                    roof["traffic_source"] = f"profiles/{tag}_pmc_hbm.json (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE, KiB -> bytes)"
                break
        out["roofline"] = roof
        bytes_step = 16.0 * M * (2 * D_VARS * D_VARS)
        out["hbm_algorithmic"] = {"bytes_per_step": bytes_step, "achieved_GBps": bytes_step * steps_per_s / 1e9,
                                  "frac_of_8TBps": bytes_step * steps_per_s / 1e9 / PEAK_HBM_GBPS}
        out["kernel_us_per_step"] = {k_: ms / K * 1e3 for k_, (ms, n_) in timers.items()}
        # the same replay with the production schedule (acyclicity kernel on the second stream, timed there): overlapping kernels share the GPU
        eng.set_state(**snap)
        eng.set_profiling(2)
        eng.reset_timers()
        eng.run(W, K)
        out["kernel_us_per_step_concurrent"] = {k_: ms / K * 1e3 for k_, (ms, n_) in eng.timers().items()}
        eng.set_profiling(False)
        out["bge_executed_gflop_per_step"] = bge_flops / 1e9

        if not args.no_cpu_baseline:
            # ---- CPU baseline: the oracle's C port on this box's host cores, bounded sample of the SAME window (from t = W) ----
            from oracle.c_oracle import COracle
            cores = min(os.cpu_count() or 1, M)   # OpenMP over particles: more threads than particles are idle
            cfg1 = make_config(n_vars=D_VARS, n_particles=M, n_observations=N_OBS, n_grad_mc_samples=S_MC,
                               n_acyclicity_mc_samples=SA_MC)

            def cpu(prec, n_steps, mode):
                co = COracle(prec)
                st = {k_: (v.astype(co.real) if v.dtype.kind == "f" else v.copy()) for k_, v in snap.items()}
                st.setdefault("theta", None), st.setdefault("v_theta", None)
                t0 = time.perf_counter()
                co.run(cfg1, data.x, None, st, W, n_steps, bge_mode=mode, n_threads=cores)
                return n_steps / (time.perf_counter() - t0)

            n_cpu = min(max(K, 10), 20)
            faithful = cpu("f32", n_cpu, 0)   # the reference's algorithm: masked d x d slogdet (LU) per node, float32
            out["cpu_baseline"] = {"value": faithful, "unit": "steps/s", "cores": cores, "kind": "port",
                                   "sample": f"{n_cpu} steps (t={W}..{W + n_cpu - 1}) of the same workload from the same state, f32 C port of the "
                                             "reference algorithm (masked d x d LU slogdet per node, func.py:128-145), OpenMP over particles"}
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
