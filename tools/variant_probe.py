"""Run one golden through several kernel-variant environments and report pass/fail (GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import oracle_api as oa
from shrimp_amd import gmapper as gm
name = sys.argv[1]
contigs, reads, sam = oa.load_golden(name)
for spec in sys.argv[2:]:
    env = dict(kv.split("=") for kv in spec.split(",")) if spec != "-" else {}
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        ix = gm.Index(contigs); s = gm.Session(ix, max_batch_reads=4096)
        res = []
        for rep in range(3):
            got = oa.sam_header(contigs) + s.map_reads(reads)
            res.append(got == sam)
        print(spec, res, {k: s.stats[k] for k in ("survivors", "survivors_pruned", "anchors", "windows", "retries")}, flush=True)
        s.close(); ix.close()
    finally:
        for k, v in old.items():
            if v is None: os.environ.pop(k, None)
            else: os.environ[k] = v
