run() { w=$1; tag=$2; shift 2; env "$@" timeout -k 10 150 python bench.py --workload $w --steps 2 --warmup 1 --no-cpu-baseline --no-other-workloads > gpurun_out/ab_$tag.json 2> gpurun_out/ab_$tag.err; python3 -c "
import json
d=json.loads(open('gpurun_out/ab_$tag.json').read().strip().splitlines()[-1]); print('$tag', round(d['value']), round(d['ms_per_step'],1), {k: round(v,1) for k,v in d.get('stages_ms_per_step',{}).items()})"; }
for w in cfg2 cfg3; do run $w ${w}_s16 GM_P2_G=16 GM_OVERLAP=0; run $w ${w}_s8 GM_OVERLAP=0; run $w ${w}_s1 GM_OVERLAP=0 GM_P2_G4=0; done
