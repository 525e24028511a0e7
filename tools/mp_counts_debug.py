"""Stage counts of the paired front (anchors need GM_NO_PRUNE=1 to be comparable with the oracle's) with and without the mate-pair region counts."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from shrimp_amd import gmapper as gm
from tests import oracle_api as oa
base = sys.argv[1] if len(sys.argv) > 1 else "cfg5s_2x150_1Mbp"
g = oa.load_golden_pairs(base)
ix = gm.Index(g["contigs"], names=g["contig_names"]); s = gm.Session(ix, max_batch_reads=4096)
for hp in (1, 0):
    o = gm.PairOpts.default(g["mode"], g["ins"][0], g["ins"][1]); o.half_paired = hp
    s.map_pairs(g["m1"], g["m2"], g["names1"], g["names2"], opts=o)
    st = s.stats
    print("half_paired", hp, "kernel", gm.lib().gm_last_lookup_kernel().decode(), {k: st[k] for k in ("survivors", "anchors", "windows")})
for opts in ("", "half-paired=0"):
    q = oa.Session(g["contigs"], g["contig_names"], opts=opts or None); q.set_pairing(g["mode"], *g["ins"])
    q.map_pairs_sam(g["m1"], g["m2"], g["names1"], g["names2"], nthreads=4)
    print("oracle", repr(opts), q.last_pair_counts())
