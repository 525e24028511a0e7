#!/bin/bash
# diagnostic build with k_anchors' phase stamps (tools/k2_stamps.py); the normal objects must be built first
set -e
cd "$(dirname "$0")/../shrimp_amd/csrc"
mkdir -p /tmp/k2st
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DGM_TUNING -DK2_STAMPS -c gm_anchors.hip -o /tmp/k2st/gm_anchors_st.o
hipcc --offload-arch=gfx950 -shared -o ../libgm_k2stamps.so build/gm_host.o build/gm_index.o build/gm_lookup.o build/gm_lookup5.o /tmp/k2st/gm_anchors_st.o build/gm_sw.o build/gm_post.o build/gm_pair.o build/gm_prune.o build/gm_cxx_shims.o build/gm_merge.o -lz
