"""Debug helper: SAM of one golden through a forced K1 variant vs the golden; prints the differing records."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import oracle_api as oa
from shrimp_amd import gmapper as gm
name = sys.argv[1] if len(sys.argv) > 1 else "cfg2s_100bp_2Mbp"
contigs, reads, sam = oa.load_golden(name)
ix = gm.Index(contigs); s = gm.Session(ix, max_batch_reads=4096)
got = oa.sam_header(contigs) + s.map_reads(reads)
print("kernel", gm.lib().gm_last_lookup_kernel().decode(), "stats", s.stats)
a = [l for l in got.split(b"\n") if not l.startswith(b"@")]; b = [l for l in sam.split(b"\n") if not l.startswith(b"@")]
sa, sb = set(a), set(b)
print("records got %d want %d, only-got %d only-want %d" % (len(a), len(b), len(sa - sb), len(sb - sa)))
for l in sorted(sa - sb)[:8]: print("GOT ", l[:160].decode())
for l in sorted(sb - sa)[:8]: print("WANT", l[:160].decode())
