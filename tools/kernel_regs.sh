#!/bin/bash
# register / spill / occupancy report of one .hip file's kernels: tools/kernel_regs.sh gm_lookup5 [extra hipcc flags]
cd "$(dirname "$0")/../shrimp_amd/csrc"
F=$1; shift
X=""; [ "$F" = gm_lookup5 ] && X="-mllvm -amdgpu-atomic-optimizer-strategy=None"
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -w -DGM_TUNING $X "$@" -Rpass-analysis=kernel-resource-usage -c $F.hip -o /tmp/kr_$F.o 2>&1 | python3 -c '
import sys, re
cur = {}
for line in sys.stdin:
    if "error" in line: print(line.rstrip())
    m = re.search(r"remark: (?:[^:]+:\d+:\d+: )?\s*(Function Name|VGPRs|AGPRs|SGPRs Spill|VGPRs Spill|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\S+)", line)
    if not m: continue
    k, v = m.groups()
    if k == "Function Name": cur = {"name": v}
    cur[k] = v
    if k.startswith("LDS"):
        if "rocprim" not in cur["name"]:
            print("%-60s V %4s A %3s occ %s sgprSpill %4s vgprSpill %4s scratch %s" % (cur["name"][:60], cur.get("VGPRs"), cur.get("AGPRs"), cur.get("Occupancy [waves/SIMD]"), cur.get("SGPRs Spill"), cur.get("VGPRs Spill"), cur.get("ScratchSize [bytes/lane]")))
'
