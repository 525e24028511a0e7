"""Phase shares of k_anchors on the 3 Gbp workload (diagnostic build -DK2_STAMPS, loaded through GM_LIB_PATH).
usage (GPU box): GM_OVERLAP=0 GM_LIB_PATH=shrimp_amd/libgm_k2stamps.so python tools/k2_stamps.py"""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from shrimp_amd import gmapper as gm, synth
n = 262144
gname, gseed, _, L, rseed = synth.CONFIGS["cfg3"]
L = int(os.environ.get("GM_STAMPS_READ_LEN", L))            # 150: the reads of the paired workload, mapped unpaired here
contigs = synth.make_genome(synth.contig_lengths(gname, 1.0), gseed)
reads, _ = synth.make_reads(contigs, n, L, rseed)
ix = gm.Index(contigs); s = gm.Session(ix, max_batch_reads=131072)
os.environ["GM_RAMP_MIN"] = "131072"
s.map_reads(reads[:8192])
lib = gm.lib(); out = (C.c_ulonglong * 8)()
lib.gm_debug_k2_stamps(out)
s.map_reads(reads)
print({k: v for k, v in s.stats.items() if k.startswith("ms_")})
lib.gm_debug_k2_stamps(out)
v = [int(x) for x in out]; tot = sum(v[:5]) or 1
for nm, x in zip(["load + key sort", "contig look-up", "collapse (class sort)", "windows", "window sort"], v): print("%-24s %6.2f %%  %8.0f ticks per read-strand" % (nm, 100.0 * x / tot, x / (2.0 * n)))
