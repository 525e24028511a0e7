#!/bin/bash
# pass-2 configuration probe on the 3 Gbp workload: prints ms_pass2 per 1 M reads
run() { echo "== $*"; env "$@" timeout -k 10 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.readline())
s = d['stages_ms_per_step']
print('reads/s %.0f  pass2 %.1f ms  pass1 %.1f  anchors %.1f  lookup %.1f  host %.1f' % (d['value'], s['ms_pass2'], s['ms_pass1'], s['ms_anchors'], s['ms_lookup'], s['ms_host']))"; }
for cfg in "$@"; do run $cfg; done
