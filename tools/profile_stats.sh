#!/bin/bash
# rocprofv3 kernel-trace summary of one bench workload (run on the GPU box from the repo root):  tools/profile_stats.sh cfg5 131072
set -e
W=${1:-cfg5}; U=${2:-131072}
R=$PWD; O=$R/gpurun_out/prof_$W; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --workload $W --steps 2 --warmup 1 --no-cpu-baseline --no-other-workloads --reads-per-step $U > $O/bench_stats.json 2> $O/stats.err
python3 - <<PY
import csv, glob, os
f = sorted(glob.glob("$O/stats/**/*kernel_stats.csv", recursive=True), key=os.path.getmtime)[-1]
import shutil; shutil.copy(f, "$O/kernel_stats.csv")
for r in csv.DictReader(open(f)):
    print(r["Name"].split("(")[0][-60:], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"])
PY
rm -rf $O/stats
