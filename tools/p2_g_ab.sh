#!/bin/bash
# pass 2 with four against eight windows a wave on every workload (tuning build): tools/p2_g_ab.sh
run() { w=$1; tag=$2; shift 2; env "$@" timeout -k 10 150 python bench.py --workload $w --steps 4 --warmup 1 --no-cpu-baseline --no-other-workloads > gpurun_out/ab_$tag.json 2> gpurun_out/ab_$tag.err; python3 -c "
import json
d=json.loads(open('gpurun_out/ab_$tag.json').read().strip().splitlines()[-1]); print('$tag', round(d['value']), round(d['ms_per_step'],1), {k: round(v,1) for k,v in d.get('stages_ms_per_step',{}).items()})"; }
for w in cfg3 cfg2 cfg5; do run $w ${w}_g16 GM_P2_G=16; run $w ${w}_g8 GM_RAMP_MIN=8192; done
