#!/bin/bash
# rocprofv3 passes for the 3 Gbp workload (run on the GPU box from the repo root): kernel stats, then PMC passes.
# Small step (262 144 reads = 2 sub-batches) keeps every pass short.
set -e
R=$PWD; O=$R/gpurun_out/prof_cfg3; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
ARGS="$R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-other-workloads --reads-per-step 262144"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $ARGS > $O/bench_stats.json 2> $O/stats.err
PASSES=${PASSES:-"FETCH_SIZE|WRITE_SIZE|SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_LDS|SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES|TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"}
IFS="|"; for pass in $PASSES; do unset IFS
  tag=$(echo $pass | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $O/pmc_$tag -- python3 $ARGS > $O/bench_$tag.json 2> $O/pmc_$tag.err || echo "pass $tag failed"
  python3 $R/tools/pmc_summary.py $O/pmc_$tag $O/pmc_$tag.summary.csv || true
  rm -rf $O/pmc_$tag
  echo "pass $tag done"
done
python3 - <<PY
import csv, glob
f = glob.glob("$O/stats/**/*kernel_stats.csv", recursive=True)
for r in csv.DictReader(open(f[0])):
    print(r["Name"].split("(")[0][:40], r["Calls"], r["AverageNs"], r["Percentage"])
PY
