#!/bin/bash
# rocprofv3 kernel stats of one short bench run: tools/kernel_stats.sh <workload> [extra bench args]   (run on the GPU box from the repo root)
set -e
R=$PWD; W=${1:-cfg4}; shift || true
O=$R/gpurun_out/kstats_$W; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 $R/bench.py --workload $W --steps 1 --warmup 1 --no-cpu-baseline "$@" > $O/bench.json 2> $O/err.log
python3 - <<PY
import csv, glob
f = glob.glob("$O/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:14]:
    print("%-44s calls %5s  avg %10.3f ms  total %9.1f ms  %5s %%" % (r["Name"].split("(")[0].replace("void ", "")[:44], r["Calls"], float(r["AverageNs"]) / 1e6, float(r["TotalDurationNs"]) / 1e6, r["Percentage"]))
PY
