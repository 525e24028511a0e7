"""Phase shares of k_pass2_cs_g4 on the colour-space 3 Gbp workload (diagnostic build -DP2CS_STAMPS, loaded through GM_LIB_PATH).
usage (GPU box): GM_LIB_PATH=shrimp_amd/libgm_p2csstamps.so python tools/p2cs_stamps.py [reads]"""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from shrimp_amd import gmapper as gm, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
gname, gseed, _, _, _ = synth.CONFIGS["cfg3"]
contigs = synth.make_genome(synth.contig_lengths(gname, 1.0), gseed)
reads, _ = synth.make_cs_reads(contigs, n, 50, 4)
p = gm.default_params_cs()
ix = gm.Index(contigs, params=p); s = gm.Session(ix, params=p, max_batch_reads=131072)
s.map_reads_cs(reads[:8192])
lib = gm.lib(); out = (C.c_ulonglong * 8)()
lib.gm_debug_p2cs_stamps(out)
s.map_reads_cs(reads)
print({k: v for k, v in s.stats.items() if k.startswith("ms_") or k == "full_calls"})
lib.gm_debug_p2cs_stamps(out)
v = [int(x) for x in out]; np_ = v[3] or 1
for nm, x in zip(["set-up (unpack, translations)", "cells", "traceback"], v): print("%-32s %10.0f ticks per pass" % (nm, x / np_))
print("steps per pass %.1f, a group's own steps %.1f per window slot, band cells per pass %.0f, ticks per step %.0f" % (v[4] / np_, v[5] / np_, v[6] / np_, v[1] / max(v[4], 1)))
print("passes", v[3])
