"""How far the device post_sw (gm_post.hip) is from the host routine on a colour-space golden: differing SAM lines and fields.
usage (GPU box): python tools/cs_post_diff.py [golden name]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from shrimp_amd import gmapper as gm
from tests import oracle_api as oa
name = sys.argv[1] if len(sys.argv) > 1 else "cfg4s_50col_2Mbp"
contigs, reads, sam = oa.load_golden(name)
p = gm.default_params_cs(); p.sam_unaligned = 1 if name.endswith("_unal") else 0
ix = gm.Index(contigs, params=p); s = gm.Session(ix, params=p, max_batch_reads=1024)
got = oa.sam_header(contigs) + s.map_reads_cs(reads)
a, b = got.split(b"\n"), sam.split(b"\n")
nd = 0
for x, y in zip(a, b):
    if x != y:
        nd += 1
        fx, fy = x.split(b"\t"), y.split(b"\t")
        if nd <= 8: print([ (i, u, v) for i, (u, v) in enumerate(zip(fx, fy)) if u != v ])
print("lines", len(a), len(b), "differing", nd)
