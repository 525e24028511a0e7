#!/bin/bash
# round 4: the half-size shape of k_lookup_v5 (two workgroups per CU) against the full shape -- phase stamps and kernel time per 262 144 reads (GPU box).
# usage: tools/k5_half_probe.sh <tag>     (libraries built beforehand: make -C shrimp_amd/csrc && tools/build_k5_stamps.sh)
set -e -o pipefail
tag=${1:-r04a}
out=gpurun_out/${tag}_half_probe.txt
mkdir -p gpurun_out
: > $out
run() {  # label, then env assignments
  local label=$1; shift
  echo "=== $label: $*" | tee -a $out
  env "$@" GM_LIB_PATH=shrimp_amd/libgm_k5stamps.so timeout -k 10 240 python tools/k5_stamps.py 262144 >> $out 2>&1
  tail -n 13 $out
}
run "full shape (k_lookup_v5<15>)" GM_K5_HALF=0
run "half shape, Bloom, rounds by estimate" GM_K5_HALF=1
run "half shape, Bloom, 2 rounds" GM_K5_HALF=1 GM_K5_ROUNDS=2
run "half shape, plain seen[], rounds by estimate" GM_K5_HALF=2
run "half shape, Bloom, ONE workgroup per CU" GM_K5_HALF=1 GM_K5_GRID=256
echo "=== 150-base reads (the mates of cfg5), k_lookup_v5_rounds, 131 072 reads" | tee -a $out
GM_STAMPS_READ_LEN=150 GM_LIB_PATH=shrimp_amd/libgm_k5stamps.so timeout -k 10 240 python tools/k5_stamps.py 131072 >> $out 2>&1
tail -n 13 $out
