#!/usr/bin/env python
"""SVGD steps/sec of the DiBS hot path on MI355X (BASELINE.json metric: d=50, n_particles=128, BGe).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One "step" = one SVGD step (svgd.py:226-267 of the reference) over all 128 particles on synthetic ER-2
linear-Gaussian data that is resident in HBM before the timed region.  N > 1 shards the particles over the
ranks (strong scaling: total work fixed) with one RCCL all-gather of [z | grad_z] per step.
Rank 0 prints ONE JSON line (see DESIGN.md "Measurement" for every field)."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

D_VARS, N_PARTICLES, N_OBS, S_MC, SA_MC = 50, 128, 100, 128, 32
PEAK_F32_TFLOPS = 157.3   # MI355X_MICROARCH.md: FP32 vector == FP32 MFMA peak
PEAK_HBM_GBPS = 8000.0


def binary_powering_matmuls(n):
    return (n.bit_length() - 1) + (bin(n).count("1") - 1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()
    K, W, N = args.steps, args.warmup, args.gpus
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
    cfg = make_config(n_vars=D_VARS, n_particles=N_PARTICLES, n_observations=N_OBS, n_grad_mc_samples=S_MC,
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
    t_begin = time.perf_counter()
    run(W, K)                       # timed: steps W .. W+K-1
    fence()
    elapsed = time.perf_counter() - t_begin
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    steps_per_s = K / elapsed

    out = {
        "metric": "SVGD steps/sec (d=50, n_particles=128, BGe)", "value": steps_per_s, "unit": "steps/s", "n_gpus": N,
        "steps": K, "warmup": W, "ms_per_step": 1e3 * elapsed / K, "higher_is_better": True,
        "scaling": "strong" if N > 1 else "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "MarginalDiBS+BGe (score-function estimator), Erdos-Renyi-2 linear-Gaussian data",
                   "n_vars": D_VARS, "n_particles": N_PARTICLES, "n_observations": N_OBS, "n_grad_mc_samples": S_MC,
                   "n_acyclicity_mc_samples": SA_MC, "timed_steps": f"t={W}..{W + K - 1} of one trajectory from PRNGKey(1)",
                   "parallelism": f"particles sharded over {N} rank(s), 1 all-gather/step" if N > 1 else "single GPU"},
    }

    if rank == 0 and N == 1:
        # ---- roofline of the dominant kernel: same K steps replayed with per-kernel HIP events ----
        eng.init_particles(random.PRNGKey(1))
        eng.run(0, W)
        eng.set_profiling(True)
        eng.reset_timers()
        eng.run(W, K)
        timers = eng.timers()
        bge_flops = float(eng.counters()[0])  # executed Cholesky flops (sum n^3/3 over all sampled parent sets)
        eng.set_profiling(False)
        total_ms = sum(ms for ms, _ in timers.values())
        dom = max(timers, key=lambda k_: timers[k_][0])
        dom_ms, dom_n = timers[dom]
        avg_s = dom_ms / dom_n * 1e-3
        # SURVEY.md 8(d) algorithmic flop counts per step (= per launch: each kernel is launched once per step)
        dense_bge = N_PARTICLES * S_MC * D_VARS * 2 * D_VARS ** 3 / 3.0          # F_lik(BGe), dense-Cholesky count
        acyc_flops = N_PARTICLES * SA_MC * binary_powering_matmuls(D_VARS - 1) * 2 * D_VARS ** 3   # F_acyc
        rocprof_names = {"acyc": "k_acyc<4, true>", "bge_nodes": "k_bge_nodes<4, true>", "bge_big": "k_bge_big<16|32|64, true> (3 launches)"}
        roof = {"kernel": dom, "rocprof_kernel": rocprof_names.get(dom, "k_" + dom), "bound": "mfma",
                "pipe": "mfma_f32" if dom == "acyc" else "valu_f32", "avg_launch_us": avg_s * 1e6, "launches": dom_n,
                "share_of_step": dom_ms / total_ms, "unit": "TFLOP/s", "peak": PEAK_F32_TFLOPS, "traffic": None}
        if dom == "acyc":
            roof["flops_per_launch"] = acyc_flops
            roof["flops_model"] = "M*Sa*c(d-1)*2*d^3, c(49)=7 matmuls of binary powering (SURVEY 8(d) F_acyc)"
            roof["achieved"] = acyc_flops / avg_s / 1e12
        elif dom in ("bge_nodes", "bge_big"):
            # the BGe kernels skip the dense count by factorising only R[pa U j]; both figures are given
            roof["flops_per_launch"] = dense_bge
            roof["flops_model"] = ("M*S*d*2*d^3/3 dense count (SURVEY 8(d) F_lik) over the whole BGe group; the kernels execute "
                                   "sum (l+1)^3/3 instead, see executed_tflops")
            grp_s = (timers["bge_nodes"][0] + timers["bge_big"][0]) / dom_n * 1e-3
            roof["achieved"] = dense_bge / grp_s / 1e12
            roof["executed_tflops"] = bge_flops / max(dom_n, 1) / grp_s / 1e12
        else:
            roof["achieved"] = None
        roof["frac"] = roof["achieved"] / PEAK_F32_TFLOPS if roof["achieved"] else None
        pmc = os.path.join(ROOT, "profiles", "round1_pmc_hbm.json")   # separate rocprofv3 --pmc passes of this command
        if os.path.exists(pmc):
            rec = json.load(open(pmc)).get(dom)
            if rec:
                roof["traffic"] = rec["hbm_bytes_per_launch"]
                roof["traffic_source"] = "profiles/round1_pmc_hbm.json (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE, KiB -> bytes)"
        out["roofline"] = roof
        bytes_step = 16.0 * N_PARTICLES * (2 * D_VARS * D_VARS)
        out["hbm_algorithmic"] = {"bytes_per_step": bytes_step, "achieved_GBps": bytes_step * steps_per_s / 1e9,
                                  "frac_of_8TBps": bytes_step * steps_per_s / 1e9 / PEAK_HBM_GBPS}
        out["kernel_us_per_step"] = {k_: ms / K * 1e3 for k_, (ms, n_) in timers.items()}

        if not args.no_cpu_baseline:
            # ---- CPU baseline: the oracle's C port on this box's host cores, bounded sample ----
            from oracle.c_oracle import COracle
            cores = min(os.cpu_count() or 1, N_PARTICLES)   # OpenMP over particles: more threads than particles are idle
            co = COracle("f64")
            cfg1 = make_config(n_vars=D_VARS, n_particles=N_PARTICLES, n_observations=N_OBS, n_grad_mc_samples=S_MC,
                               n_acyclicity_mc_samples=SA_MC)
            st = co.new_state(cfg1, random.PRNGKey(1))
            t0 = time.perf_counter()
            co.run(cfg1, data.x, None, st, 0, 2, bge_mode=0, n_threads=cores)   # the reference's masked d x d slogdet
            faithful = 2 / (time.perf_counter() - t0)
            st = co.new_state(cfg1, random.PRNGKey(1))
            t0 = time.perf_counter()
            co.run(cfg1, data.x, None, st, 0, 8, bge_mode=1, n_threads=cores)   # same compact Cholesky as the GPU
            compact = 8 / (time.perf_counter() - t0)
            out["cpu_baseline"] = {"value": faithful, "unit": "steps/s", "cores": cores, "kind": "port",
                                   "sample": "2 steps (t=0,1) of the same workload, f64 C port of the reference algorithm "
                                             "(masked d x d LU slogdet per node, as func.py:128-145), OpenMP over particles"}
            out["cpu_baseline_compact"] = {"value": compact, "unit": "steps/s", "cores": cores, "kind": "port",
                                           "sample": "8 steps (t=0..7), same C port with the GPU's compact parent-set Cholesky"}
    eng.close()
    if rank == 0:
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
