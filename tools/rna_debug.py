import os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from shrimp_amd import gmapper as gm
from tests import oracle_api as oa
tag = sys.argv[1]
G = "/root/repo/tests/golden"
g = oa.load_rna_case(tag)
p = gm.default_params_cs() if g["colour"] else gm.default_params()
p.sam_unaligned = 1
if g["opts"]:
    for k, v in dict(local_alignment=1, ungapped=1, anchor_width=0, a_gap_open_score=-255, b_gap_open_score=-255, hash_filter_calls=0).items(): setattr(p, k, v)
ix = gm.Index(g["contigs"], names=g["contig_names"], params=p)
s = gm.Session(ix, params=p, max_batch_reads=256)
head = oa.sam_header(g["contigs"], g["contig_names"])
paths = [os.path.join(G, f) for f in g["files"]]
if g["pairing"]:
    mode, lo, hi = g["pairing"]
    got = head + s.map_pairs_file(paths[0], paths[1], mode=mode, min_insert=lo, max_insert=hi)
else:
    got = head + s.map_reads_file(paths[0], qual_delta=33 if paths[0].endswith(".fq.gz") else None)
a, b = got.split(b"\n"), g["sam"].split(b"\n")
nd = 0
for x, y in zip(a, b):
    if x != y:
        nd += 1
        fx, fy = x.split(b"\t"), y.split(b"\t")
        if nd <= 6: print([(i, u, v) for i, (u, v) in enumerate(zip(fx, fy)) if u != v], fx[0], fx[9] if len(fx) > 9 else b"")
print("lines", len(a), len(b), "differing", nd)
