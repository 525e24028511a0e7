#!/bin/bash
# usage: tools/bench_stages.sh "<ENV=VAL ...>" [bench args]  -> one line: env, reads/s, stage ms
envs="$1"; shift
out=$(env $envs python bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" 2>gpurun_out/bench_stages.err | tail -1)
python3 - "$envs" "$out" <<'PY'
import json, sys
d = json.loads(sys.argv[2])
st = {k[3:]: round(v, 1) for k, v in d["stages_ms_per_step"].items()}
print("%-28s %9.0f reads/s  %s  roof %.4f  exact %.2e retries? pruned %.0f" % (sys.argv[1] or "(default)", d["value"], st, d["roofline"]["frac"], d["per_read"]["exact_order_frac"], d["per_read"].get("survivors_pruned", 0)))
PY
