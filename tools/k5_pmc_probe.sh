#!/bin/bash
# round 4: fabric read requests / L2 hits of k_lookup_v5 with pass B in forward and in reverse record order (GPU box; libraries from tools/build_k5_stamps.sh)
set -e
R=$PWD; tag=${1:-r04d}; O=$R/gpurun_out/${tag}_pmc; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for v in fwd rev; do
  lib=libgm_k5stamps.so; [ $v = rev ] && lib=libgm_k5stamps_rev.so
  export GM_LIB_PATH=$R/shrimp_amd/$lib
  for pass in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
    t=$(echo $pass | cut -d' ' -f1)
    rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $O/${v}_$t -- python3 $R/tools/k5_stamps.py 262144 > $O/${v}_$t.log 2> $O/${v}_$t.err || echo "pass $v $t failed"
    python3 $R/tools/pmc_summary.py $O/${v}_$t $O/${v}_$t.summary.csv || true
    rm -rf $O/${v}_$t
    grep "k_lookup_v5" $O/${v}_$t.summary.csv || true
  done
done
