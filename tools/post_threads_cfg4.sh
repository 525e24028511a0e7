#!/bin/bash
# threads of k_post_sw_cs (one thread per pass-2 result, a column scratch each) on the colour-space workload, tuning build: tools/post_threads_cfg4.sh
run() { tag=$1; shift; env "$@" timeout -k 10 120 python bench.py --workload cfg4 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/pt_$tag.json 2> gpurun_out/pt_$tag.err; python3 -c "
import json
d=json.loads(open('gpurun_out/pt_$tag.json').read().strip().splitlines()[-1]); print('$tag', round(d['value']), round(d['ms_per_step'],1), {k: round(v,1) for k,v in d.get('stages_ms_per_step',{}).items()})"; }
for t in 32768 65536 131072 262144; do run t$t GM_POST_THREADS=$t; done
for t in 32768 131072; do run s$t GM_POST_THREADS=$t GM_OVERLAP=0; done
