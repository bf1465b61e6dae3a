#!/bin/bash
# per-kernel durations of config 5 (one repetition of the driver's window) -> gpurun_out/<tag>_cfg5_kernel_stats.csv
TAG=${1:-r4}
DIBS_NO_ACYC_STREAM2=1 bash scripts/rocprof_bench.sh ${TAG}_cfg5 --config 5 --reps 1 --min-seconds 0 --steps 20 --warmup 5 | head -25
