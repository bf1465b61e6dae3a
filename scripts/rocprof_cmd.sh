#!/bin/bash
# usage: scripts/rocprof_cmd.sh <tag> -- <cmd...>   kernel-trace + stats of an arbitrary command -> gpurun_out/<tag>_kernel_stats.csv
set -e
TAG=$1; shift 2
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o $TAG -- "$@" > $OUT/run.log 2>&1 || { tail -20 $OUT/run.log; exit 1; }
f=$(find $OUT -name "*kernel_stats.csv" | head -1)
cp "$f" gpurun_out/${TAG}_kernel_stats.csv
python - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    print(f"{r['Name'].split('(')[0][:50]:50s} calls {int(r['Calls']):6d}  avg {float(r['AverageNs'])/1e3:9.1f} us  {float(r['Percentage']):6.2f} %")
PY
tail -3 $OUT/run.log
