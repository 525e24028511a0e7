#!/bin/bash
# K1 v4 configuration probe on the 3 Gbp workload: prints ms_lookup per 1 M reads for a few settings
set -e
mkdir -p gpurun_out
run() { echo "== $*"; env "$@" timeout -k 10 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline $BENCH_EXTRA 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.readline())
st = d['stages_ms_per_step']; print('reads/s %.0f  step %.0f ms  lookup %.1f  frac %.3f  anchors %.1f  pass1 %.1f  pass2 %.1f  host %.1f' % (d['value'], d['ms_per_step'], st['ms_lookup'], d['roofline']['frac'], st['ms_anchors'], st['ms_pass1'], st['ms_pass2'], st.get('ms_host', 0)))"; }
for cfg in "$@"; do run $cfg; done
