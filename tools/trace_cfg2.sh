#!/bin/bash
# kernel timeline of the steady sub-batches of a cfg2 step (rocprofv3 --kernel-trace around tools/timeline_cfg.py): where the device idles between its kernels
set -e
R=$PWD; O=$R/gpurun_out/trace2; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O -- python3 $R/tools/timeline_cfg.py ${1:-cfg2} > $O/out.log 2> $O/err.log
python3 - <<PY
import csv, glob
f = glob.glob("$O/**/*kernel_trace.csv", recursive=True)[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "")[:26], "K") for r in csv.DictReader(open(f))]
g = glob.glob("$O/**/*memory_copy_trace.csv", recursive=True)
if g:
    for r in csv.DictReader(open(g[0])): rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "copy " + r.get("Direction", "")[:20], "C"))
rows.sort()
t0 = rows[0][0]
last = rows[-150:-60]
prev_end = None
for s, e, n, k in last:
    print("%10.3f %10.3f %8.3f ms  %s %s" % ((s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e6, k, n))
PY
