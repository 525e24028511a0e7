#!/bin/bash
# lane use of the pass-2 kernels: SQ_THREAD_CYCLES_VALU / (SQ_ACTIVE_INST_VALU x 64), one rocprofv3 --pmc pass per workload (--kernel-trace only beside it)
set -e
R=$PWD; O=$R/gpurun_out/pmc_p2; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for W in cfg3 cfg4; do
  rocprofv3 --kernel-trace --pmc SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES --output-format csv -d $O/$W -- python3 $R/bench.py --workload $W --steps 1 --warmup 1 --no-cpu-baseline --no-other-workloads --reads-per-step 262144 > $O/bench_$W.json 2> $O/$W.err || echo "pass $W failed"
  python3 $R/tools/pmc_summary.py $O/$W $O/$W.summary.csv || true
  rm -rf $O/$W
  grep -E "pass2|pass1|post_sw" $O/$W.summary.csv || true
done
