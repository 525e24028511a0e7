#!/bin/bash
# round 4: pass B of k_lookup_v5 in reverse record order (L2 reuse) against the forward order -- stamps per 262 144 reads, full and half shape (GPU box)
set -e -o pipefail
tag=${1:-r04c}
out=gpurun_out/${tag}_rev_probe.txt
mkdir -p gpurun_out; : > $out
run() { local lib=$1 label=$2; shift 2; echo "=== $label: $lib $*" | tee -a $out; env "$@" GM_LIB_PATH=shrimp_amd/$lib timeout -k 10 240 python tools/k5_stamps.py 262144 >> $out 2>&1; tail -n 13 $out | cut -c1-300; }
run libgm_k5stamps.so "forward, full shape" GM_K5_HALF=0
run libgm_k5stamps_rev.so "reverse, full shape" GM_K5_HALF=0
run libgm_k5stamps_rev.so "reverse, half shape, 2 rounds" GM_K5_HALF=1 GM_K5_ROUNDS=2
