#!/bin/bash
# kernel timeline of one bench step (rocprofv3 --kernel-trace): which kernels of the two streams really share the chip
set -e
R=$PWD; O=$R/gpurun_out/trace; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --reads-per-step 524288 > $O/bench.json 2> $O/err.log
python3 - <<PY
import csv, glob
f = glob.glob("$O/**/*kernel_trace.csv", recursive=True)[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "")[:28], r.get("Queue_Id", r.get("Stream_Id", "?"))) for r in csv.DictReader(open(f))]
rows.sort()
t0 = rows[0][0]
sel = [r for r in rows if any(k in r[2] for k in ("k_lookup", "k_pass1", "k_pass2", "k_prune", "k_anchors", "k_select"))]
last = sel[-90:]
for s, e, n, q in last: print("%10.3f %10.3f %8.3f ms  q%-3s %s" % ((s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e6, q, n))
PY
