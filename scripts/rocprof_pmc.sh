#!/bin/bash
# usage: scripts/rocprof_pmc.sh <tag> "<counters>" -- <cmd...>   (one PMC pass; summary csv -> gpurun_out/<tag>_pmc.csv)
set -e
TAG=$1; CNT=$2; shift 3
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/pmc_$TAG
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --pmc $CNT --kernel-trace --output-format csv -d $OUT -o $TAG -- "$@" > $OUT/run.log 2>&1 || { tail -20 $OUT/run.log; exit 1; }
f=$(find $OUT -name "*counter_collection.csv" | head -1)
python - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in rows:
    k = r["Kernel_Name"].split("(")[0][:40]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    cnt[(k, r["Counter_Name"])] += 1
for k in agg:
    print(k, {c: f"{v / cnt[(k, c)]:.4g}" for c, v in agg[k].items()})
PY
