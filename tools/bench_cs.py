"""Colour-space throughput on the 3 Gbp genome (BASELINE configs[3]: 50-colour SOLiD reads, 4 % colour errors + one indel)
+ parity of a sample against the CPU oracle.  Not the headline metric (bench.py is); numbers go to DESIGN.md.
usage: python tools/bench_cs.py [n_reads] [sample_reads] [scale]"""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from shrimp_amd import gmapper as gm, synth
n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
n_sample = int(sys.argv[2]) if len(sys.argv) > 2 else 20_000
scale = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
gname, gseed, _, _, _ = synth.CONFIGS["cfg3"]
contigs = synth.make_genome(synth.contig_lengths(gname, scale), gseed)
reads, _ = synth.make_cs_reads(contigs, n_reads, 50, 4)
p = gm.default_params_cs()
t0 = time.time(); ix = gm.Index(contigs, params=p); t_ix = time.time() - t0
s = gm.Session(ix, params=p, max_batch_reads=int(os.environ.get("GM_SUBBATCH", "131072")))
s.map_reads_cs(reads[:8192])      # warm-up (buffers, LDS attributes)
t0 = time.perf_counter(); sam = s.map_reads_cs(reads); dt = time.perf_counter() - t0
st = s.stats
out = {"workload": "50-colour CS reads (1 indel, 4%% colour errors) vs %d bp genome" % sum(len(c) for c in contigs), "reads": n_reads,
       "reads_per_s": n_reads / dt, "sam_bytes": len(sam), "mapped_frac": st["reads_matched"] / n_reads, "retries": st["retries"], "index_build_s": t_ix,
       "stages_ms": {k: v for k, v in st.items() if k.startswith("ms_")},
       "per_read": {k: st[k] / n_reads for k in ("lookups", "list_entries", "survivors", "survivors_pruned", "windows", "vec_calls", "full_calls", "exact_order_reads")}}
if n_sample:
    from tests import oracle_api as oa
    oa.load().gmo_set_threads(16)
    t0 = time.time(); o = oa.Session(contigs, opts="colour=1"); t_idx = time.time() - t0
    t0 = time.perf_counter(); want = o.map_sam(reads[:n_sample], nthreads=16); cdt = time.perf_counter() - t0
    got = s.map_reads_cs(reads[:n_sample])
    out.update({"oracle_reads_per_s_16thr": n_sample / cdt, "sample_reads": n_sample, "sample_sam_identical": bool(got == want), "oracle_index_s": t_idx})
print(json.dumps(out))
