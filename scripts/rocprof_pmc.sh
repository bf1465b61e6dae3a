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
import os
rows = list(csv.DictReader(open(sys.argv[1])))
lastn = int(os.environ.get("LASTN", "0"))   # LASTN=n: average only the last n dispatches of every kernel (steady state)
series = collections.defaultdict(list)
for r in rows:
    series[(r["Kernel_Name"].split("(")[0][:40], r["Counter_Name"])].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for (k, c), v in series.items():
    v.sort()
    v = v[-lastn:] if lastn else v
    agg[k][c] = sum(x for _, x in v)
    cnt[(k, c)] = len(v)
for k in agg:
    print(k, {c: f"{v / cnt[(k, c)]:.4g}" for c, v in agg[k].items()})
PY
