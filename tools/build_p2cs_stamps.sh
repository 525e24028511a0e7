#!/bin/bash
# diagnostic build with k_pass2_cs_g4's phase stamps (tools/p2cs_stamps.py); the normal objects must be built first
set -e
cd "$(dirname "$0")/../shrimp_amd/csrc"
mkdir -p /tmp/p2st
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DGM_TUNING -DP2CS_STAMPS $P2_EXTRA -c gm_sw.hip -o /tmp/p2st/gm_sw_st.o
hipcc --offload-arch=gfx950 -shared -o ../libgm_p2csstamps.so build/gm_host.o build/gm_index.o build/gm_lookup.o build/gm_lookup5.o build/gm_anchors.o /tmp/p2st/gm_sw_st.o build/gm_post.o build/gm_pair.o build/gm_prune.o build/gm_cxx_shims.o build/gm_merge.o -lz
