run() { tag=$1; shift; env "$@" timeout -k 10 150 python bench.py --workload cfg3 --steps 3 --warmup 1 --no-cpu-baseline --no-other-workloads > gpurun_out/l_$tag.json 2> gpurun_out/l_$tag.err; python3 -c "
import json
d=json.loads(open('gpurun_out/l_$tag.json').read().strip().splitlines()[-1]); print('$tag', round(d['value']), round(d['ms_per_step'],1), d['roofline']['kernel'], round(d['roofline']['avg_launch_ms'],2), round(d['per_unit']['list_entries']))"; }
run l14g512 GM_K5_LSW=14 GM_K1_THREADS=512 GM_K5_GRID=512
run l14g256 GM_K5_LSW=14 GM_K1_THREADS=512 GM_K5_GRID=256
run l14t1024 GM_K5_LSW=14 GM_K1_THREADS=1024 GM_K5_GRID=512
