#!/bin/bash
# diagnostic build of the library with the finalisation's stage counters (-DGM_HOST_PROFILE; tools/host_profile.py); the normal objects must be built first
set -e
cd "$(dirname "$0")/../shrimp_amd/csrc"
mkdir -p /tmp/hp
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -w -DGM_TUNING -DGM_HOST_PROFILE -c gm_host.hip -o /tmp/hp/gm_host_hp.o
hipcc --offload-arch=gfx950 -shared -o ../libgm_hostprof.so /tmp/hp/gm_host_hp.o build/gm_index.o build/gm_lookup.o build/gm_lookup5.o build/gm_anchors.o build/gm_sw.o build/gm_post.o build/gm_pair.o build/gm_prune.o build/gm_cxx_shims.o build/gm_merge.o -lz
