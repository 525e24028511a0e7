#!/bin/bash
# K1 v4 configuration probe on the 3 Gbp workload: prints ms_lookup per 1 M reads for a few settings
set -e
mkdir -p gpurun_out
run() { echo "== $*"; env "$@" timeout -k 10 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.readline())
print('reads/s %.0f  lookup %.1f ms  frac %.3f  surv/read %.2f  anchors %.1f' % (d['value'], d['stages_ms_per_step']['ms_lookup'], d['roofline']['frac'], d['per_read']['survivors'], d['stages_ms_per_step']['ms_anchors']))"; }
for cfg in "$@"; do run $cfg; done
