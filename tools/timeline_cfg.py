"""Where the calling thread of gm_map_reads_device spends a step (GM_TIMELINE=1 prints one line per sub-batch to stderr): python tools/timeline_cfg.py [cfg2|cfg3]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from shrimp_amd import gmapper as gm, synth
w = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
gname, gseed, _, L, rseed = synth.CONFIGS["cfg3" if w == "cfg4" else w]
if w == "cfg4": L, rseed = 50, 4                          # (bench.py WORKLOADS["cfg4"]: the 3 Gbp genome, 50 colours)
contigs = synth.make_genome(synth.contig_lengths(gname, 1.0), gseed)
if w == "cfg4":                                            # colour space: packed colours + primer bytes from host memory, as bench.py's step does
    reads, _ = synth.make_cs_reads(contigs, 1_000_000, L, rseed)
    p = gm.default_params_cs(); ix = gm.Index(contigs, params=p); s = gm.Session(ix, params=p, max_batch_reads=131072)
    packed = np.ascontiguousarray(synth.pack_reads(np.ascontiguousarray(reads[:, 1:]))); ibp = np.ascontiguousarray(reads[:, 0])
    s.map_cs_packed(packed, ibp, len(reads), L, return_bytes=False)
    os.environ["GM_TIMELINE"] = "1"
    for _ in range(2):
        t = time.time(); s.map_cs_packed(packed, ibp, len(reads), L, return_bytes=False); dt = time.time() - t
        print("step %.1f ms" % (dt * 1e3), {k: round(v, 1) for k, v in s.stats.items() if k.startswith("ms_")}, file=sys.stderr)
    sys.exit(0)
reads, _ = synth.make_reads(contigs, 1_000_000, L, rseed)
ix = gm.Index(contigs); s = gm.Session(ix, max_batch_reads=131072)
packed = np.ascontiguousarray(synth.pack_reads(reads))
d = torch.from_numpy(packed.view(np.int32)).cuda()
s.map_device(d.data_ptr(), len(reads), L, return_bytes=False)
os.environ["GM_TIMELINE"] = "1"
t = time.time(); s.map_device(d.data_ptr(), len(reads), L, return_bytes=False); dt = time.time() - t
print("step %.1f ms" % (dt * 1e3), {k: round(v, 1) for k, v in s.stats.items() if k.startswith("ms_")})
