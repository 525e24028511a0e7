"""Where the host finalisation's cycles go (library built with -DGM_HOST_PROFILE, see tools/build_host_profile.sh): python tools/host_profile.py [cfg2|cfg3]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from shrimp_amd import gmapper as gm, synth
w = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
gname, gseed, _, L, rseed = synth.CONFIGS[w]
contigs = synth.make_genome(synth.contig_lengths(gname, 1.0), gseed)
reads, _ = synth.make_reads(contigs, 1_000_000, L, rseed)
ix = gm.Index(contigs); s = gm.Session(ix, max_batch_reads=131072)
packed = np.ascontiguousarray(synth.pack_reads(reads))
d = torch.from_numpy(packed.view(np.int32)).cuda()
s.map_device(d.data_ptr(), len(reads), L, return_bytes=False)
lib = gm.lib(); lib.gm_host_profile_dump()
t = time.time(); s.map_device(d.data_ptr(), len(reads), L, return_bytes=False); dt = time.time() - t
print("step %.1f ms" % (dt * 1e3), {k: round(v, 1) for k, v in s.stats.items() if k.startswith("ms_")}, flush=True)
lib.gm_host_profile_dump()
