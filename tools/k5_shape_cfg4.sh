#!/bin/bash
# K1 shapes for the colour-space workload (tuning build): tools/k5_shape_cfg4.sh  -> one line per shape: reads/s, ms per step, K1 kernel, K1 ms per launch
run() { tag=$1; shift; env "$@" timeout -k 10 120 python bench.py --workload cfg4 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/s_$tag.json 2> gpurun_out/s_$tag.err; python3 -c "
import json,sys
d=json.loads(open('gpurun_out/s_$tag.json').read().strip().splitlines()[-1]); print('$tag', round(d['value']), round(d['ms_per_step'],1), d['roofline']['kernel'], round(d['roofline']['avg_launch_ms'],2))"; }
run g320 GM_K5_LSW=14 GM_K1_THREADS=512 GM_K5_GRID=320
run g384 GM_K5_LSW=14 GM_K1_THREADS=512 GM_K5_GRID=384
run g448 GM_K5_LSW=14 GM_K1_THREADS=512 GM_K5_GRID=448
run g512 GM_K5_LSW=14 GM_K1_THREADS=512 GM_K5_GRID=512
run g512_p16 GM_K5_LSW=14 GM_K1_THREADS=512 GM_K5_GRID=512 GM_P2_G=16
run g512_serial GM_K5_LSW=14 GM_K1_THREADS=512 GM_K5_GRID=512 GM_OVERLAP=0
run l13_g1024 GM_K5_LSW=13 GM_K1_THREADS=256 GM_K5_GRID=1024
run l13t512_g1024 GM_K5_LSW=13 GM_K1_THREADS=512 GM_K5_GRID=768
