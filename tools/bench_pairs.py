"""Paired-mode throughput on the 3 Gbp genome (BASELINE configs[4]: 2 x 150 bp, -p opp-in -I 100,600) + parity of a sample
against the CPU oracle.  Not the headline metric (bench.py is); numbers go to DESIGN.md.
usage: python tools/bench_pairs.py [n_pairs] [sample_pairs] [scale]"""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from shrimp_amd import gmapper as gm, synth
n_pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
n_sample = int(sys.argv[2]) if len(sys.argv) > 2 else 5_000
scale = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
gname, gseed, _, _, _ = synth.CONFIGS["cfg3"]
contigs = synth.make_genome(synth.contig_lengths(gname, scale), gseed)
reads, _ = synth.make_pairs(contigs, n_pairs, 150, 5)
m1, m2 = reads[0::2], reads[1::2]
ix = gm.Index(contigs); s = gm.Session(ix, max_batch_reads=int(os.environ.get("GM_SUBBATCH", "65536")))
# GM_PAIR_MATCH_MODE=3|2, GM_PAIR_HALF=0: the paired match modes (-n 3 / -n 2) and --no-half-paired at this size
mm = int(os.environ.get("GM_PAIR_MATCH_MODE", "4")); hp = int(os.environ.get("GM_PAIR_HALF", "1"))
po = gm.PairOpts.default("opp-in", 100, 600); po.match_mode = mm; po.half_paired = hp
s.map_pairs(m1[:4096], m2[:4096], opts=po)      # warm-up (buffers, LDS attributes)
t0 = time.perf_counter(); sam = s.map_pairs(m1, m2, opts=po); dt = time.perf_counter() - t0
st = s.stats
out = {"workload": "2x150bp opp-in pairs vs %d bp genome, -I 100,600, match_mode %d, half_paired %d" % (sum(len(c) for c in contigs), mm, hp), "pairs": n_pairs, "pairs_per_s": n_pairs / dt,
       "lookup_kernel": gm.lib().gm_last_lookup_kernel().decode(), "mp_unfiltered": st["mp_unfiltered"], "anchors": st["anchors"], "windows": st["windows"],
       "reads_per_s": 2 * n_pairs / dt, "sam_bytes": len(sam), "pairs_mapped_frac": st["reads_matched"] / n_pairs, "retries": st["retries"]}
if n_sample:
    from tests import oracle_api as oa
    oa.load().gmo_set_threads(16)
    t0 = time.time(); o = oa.Session(contigs, opts="mp-match-mode=%d;half-paired=%d" % (mm, hp)); o.set_pairing("opp-in", 100, 600); t_idx = time.time() - t0
    t0 = time.perf_counter(); want = o.map_pairs_sam(m1[:n_sample], m2[:n_sample], nthreads=16); cdt = time.perf_counter() - t0
    oc = o.last_pair_counts()
    got = s.map_pairs(m1[:n_sample], m2[:n_sample], opts=po); st2 = s.stats
    out.update({"oracle_pairs_per_s_16thr": n_sample / cdt, "sample_pairs": n_sample, "sample_sam_identical": bool(got == want), "oracle_index_s": t_idx,
                "sample_anchors_windows": [st2["anchors"], st2["windows"]], "oracle_anchors_windows": list(oc)})
print(json.dumps(out))
