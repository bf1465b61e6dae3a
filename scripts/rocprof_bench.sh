#!/bin/bash
# usage (on the GPU box, from the repo root): scripts/rocprof_bench.sh <tag> [bench args]
# kernel-trace + stats of the default bench workload; summary copied to gpurun_out/<tag>_kernel_stats.csv
set -e
TAG=${1:-prof}; shift || true
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o $TAG -- python bench.py --no-cpu-baseline "$@" > $OUT/bench.log 2>&1 || { tail -20 $OUT/bench.log; exit 1; }
grep "\"metric\"" $OUT/bench.log | tail -1 > gpurun_out/${TAG}_bench.json
f=$(find $OUT -name "*kernel_stats.csv" | head -1)
cp "$f" gpurun_out/${TAG}_kernel_stats.csv
head -20 gpurun_out/${TAG}_kernel_stats.csv
