#!/bin/bash
# per-kernel totals of one workload's bench steps: tools/kstats_cfg.sh cfg4 tag [ENV=VALUE ...] -> gpurun_out/kstats_<tag>/ (rocprofv3 --kernel-trace --stats)
set -e
W=$1; T=$2; shift 2
for kv in "$@"; do export "$kv"; done
R=$PWD; O=$R/gpurun_out/kstats_$T; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O -o run -- python3 $R/bench.py --workload $W --steps 2 --warmup 1 --no-cpu-baseline > $O/bench.json 2> $O/err.log
python3 - <<PY
import csv, glob
f = glob.glob("$O/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:12]:
    print(r["Name"].replace("void ", "")[:44].ljust(46), r["Calls"].rjust(5), "%9.2f ms" % (int(r["TotalDurationNs"]) / 1e6), "%8.3f avg" % (float(r["AverageNs"]) / 1e6), r["Percentage"])
PY
cut -c1-260 $O/bench.json
