"""ad-hoc GPU bring-up script (not a test): prints where the HIP path and the oracle/golden diverge"""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np
from shrimp_amd import gmapper as gm, synth
from tests import oracle_api as oa

name = sys.argv[1] if len(sys.argv) > 1 else "cfg2s_100bp_2Mbp"
contigs, reads, sam = oa.load_golden(name)
p = gm.default_params(); p.sam_unaligned = 1 if name.endswith("_unal") else 0
t = time.time(); ix = gm.Index(contigs, params=p); print("index build %.2fs, %d MB, slabs %d, cutoff %d" % (time.time() - t, ix.nbytes >> 20, ix.n_slabs, ix.list_cutoff))
s = gm.Session(ix, params=p, max_batch_reads=16384)
t = time.time(); got = oa.sam_header(contigs) + s.map_reads(reads); print("map %.2fs" % (time.time() - t)); print(s.stats)
la, lb = got.split(b"\n"), sam.split(b"\n")
print("lines got/want", len(la), len(lb))
nd = 0
for i, (x, y) in enumerate(zip(la, lb)):
    if x != y:
        nd += 1
        if nd <= 5: print("DIFF line", i, "\n  got ", x[:260], "\n  want", y[:260])
print("differing lines:", nd)
o = oa.Session(contigs); want = o.tophits(reads); o.close()
gt = s.tophits(reads)
print("tophits rows got/want", gt.shape, want.shape)
if gt.shape == want.shape:
    bad = np.nonzero((gt != want).any(axis=1))[0]
    print("tophit rows differing:", len(bad))
    for b in bad[:5]: print("  got ", gt[b], "\n  want", want[b])
else:
    # per-read counts
    import collections
    cg = collections.Counter(gt[:, 0]); cw = collections.Counter(want[:, 0])
    badr = [r for r in set(cg) | set(cw) if cg.get(r, 0) != cw.get(r, 0)]
    print("reads with different tophit counts:", len(badr), sorted(badr)[:10])
    for r in sorted(badr)[:3]:
        print(" read", r, "\n got\n", gt[gt[:, 0] == r], "\n want\n", want[want[:, 0] == r])
