#!/bin/bash
# fabric read requests of k_lookup_v5 for two builds of the library on ONE box (GPU box; libraries named on the command line, built by tools/build_k5_stamps.sh with K5_OUT)
set -e
R=$PWD; O=$R/gpurun_out/rdreq_ab; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for lib in "$@"; do
  export GM_LIB_PATH=$R/shrimp_amd/$lib
  rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum --output-format csv -d $O/$lib -- python3 $R/tools/k5_stamps.py 262144 > $O/$lib.log 2> $O/$lib.err || echo "pass $lib failed"
  python3 $R/tools/pmc_summary.py $O/$lib $O/$lib.summary.csv || true
  rm -rf $O/$lib
  echo "$lib: $(grep 'k_lookup_v5' $O/$lib.summary.csv | tr '\n' ' ')"
done
