"""As tools/k5_stamps.py for the 50-colour workload (cfg4): phase shares of k_lookup_v5 and how many read-strands fall back.
usage (GPU box): GM_LIB_PATH=shrimp_amd/libgm_k5stamps.so python tools/k5_stamps_cs.py [reads]"""
import os, sys, ctypes as C, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from shrimp_amd import gmapper as gm, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
contigs = synth.make_genome(synth.contig_lengths("cfg3", 1.0), 3)
reads, _ = synth.make_cs_reads(contigs, n, 50, 4)
p = gm.default_params_cs()
ix = gm.Index(contigs, params=p); s = gm.Session(ix, params=p, max_batch_reads=131072)
os.environ["GM_RAMP_MIN"] = "131072"
s.map_reads_cs(reads[:8192])
lib = gm.lib(); out = (C.c_ulonglong * 16)()
has = hasattr(lib, "gm_debug_k5_stamps")
if has: lib.gm_debug_k5_stamps(out)
t = time.time(); s.map_reads_cs(reads); dt = time.time() - t
print("kernel", lib.gm_last_lookup_kernel().decode(), "map %.3fs" % dt, {k: v for k, v in s.stats.items() if k.startswith("ms_") or k in ("survivors", "survivors_pruned", "list_entries", "lookups")})
if has:
    lib.gm_debug_k5_stamps(out); v = [int(x) for x in out]; tot = sum(v[:6]) or 1
    for nm, x in zip(["setup(+clear wait)", "pass A", "pass B", "region table", "rules+output", "bookkeeping+clear"], v): print("%-20s %6.2f %%  %8.0f ticks per read-strand" % (nm, 100.0 * x / tot, x / (2.0 * n)))
    print("candidates per read-strand %.1f, fallbacks %d of %d" % (v[6] / (2.0 * n), v[7], 2 * n))
