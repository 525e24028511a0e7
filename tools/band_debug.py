"""How many anti-diagonal steps of sw_full_cs the band leaves out (diagnostic build of gm_sw.hip with -DGM_BAND_DEBUG, linked like tools/build_k5_stamps.sh
links its object, loaded through GM_LIB_PATH).  Result at the end of round 2: none -- the reference's band keeps a corner block above and below the anchor box
(anchor_get_x_range, anchors.c:64-95), so the first row reaches column 0 and the last one the window's end; in letter space (two stripes of a 100 bp read) ~150 of 203.
usage (GPU box): GM_LIB_PATH=shrimp_amd/libgm_banddbg.so python tools/band_debug.py"""
import os, sys, ctypes as C
sys.path.insert(0, os.getcwd())
import numpy as np
from shrimp_amd import gmapper as gm, synth
contigs = synth.make_genome(synth.contig_lengths("cfg2", 1.0), 2)
reads, _ = synth.make_cs_reads(contigs, 20000, 50, 4)
p = gm.default_params_cs(); ix = gm.Index(contigs, params=p); s = gm.Session(ix, params=p, max_batch_reads=131072)
lib = gm.lib(); out = (C.c_ulonglong * 8)(); lib.gm_debug_band(out)
s.map_reads_cs(reads); lib.gm_debug_band(out); v = [int(x) for x in out]
print("stripes", v[2], "full steps/stripe %.1f band steps/stripe %.1f rw %.1f rl %.1f" % (v[0] / v[2], v[1] / v[2], v[3] / v[2], v[4] / v[2]))
