run() { tag=$1; shift; env "$@" timeout -k 10 100 python bench.py --workload cfg2 --steps 4 --warmup 1 --no-cpu-baseline --no-other-workloads > gpurun_out/bk_$tag.json 2> gpurun_out/bk_$tag.err; python3 -c "
import json
d=json.loads(open('gpurun_out/bk_$tag.json').read().strip().splitlines()[-1]); print('$tag', round(d['value']), round(d['ms_per_step'],1), {k: round(v,1) for k,v in d.get('stages_ms_per_step',{}).items()})"; grep -m1 "\[bkt\]" gpurun_out/bk_$tag.err; }
run max GM_RAMP_MIN=8192
for n in 1 2 3 4 6; do run pc$n GM_BKT_PER_CU=$n GM_TIMELINE=1; done
