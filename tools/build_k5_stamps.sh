#!/bin/bash
# diagnostic build of the library with k_lookup_v5's phase stamps (see tools/k5_stamps.py); the normal objects must be built first
set -e
cd "$(dirname "$0")/../shrimp_amd/csrc"
mkdir -p /tmp/k5st
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -mllvm -amdgpu-atomic-optimizer-strategy=None -DGM_TUNING -DK5_STAMPS $K5_EXTRA -c gm_lookup5.hip -o /tmp/k5st/gm_lookup5_st.o
hipcc --offload-arch=gfx950 -shared -o ../${K5_OUT:-libgm_k5stamps.so} build/gm_host.o build/gm_index.o build/gm_lookup.o /tmp/k5st/gm_lookup5_st.o build/gm_anchors.o build/gm_sw.o build/gm_post.o build/gm_pair.o build/gm_prune.o build/gm_cxx_shims.o build/gm_merge.o -lz
