"""Phase shares of k_lookup_v5 on the 3 Gbp workload (diagnostic build with -DK5_STAMPS, loaded through GM_LIB_PATH).
usage (GPU box): GM_LIB_PATH=shrimp_amd/libgm_k5stamps.so python tools/k5_stamps.py [reads]"""
import os, sys, ctypes as C, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from shrimp_amd import gmapper as gm, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
gname, gseed, _, L, rseed = synth.CONFIGS["cfg3"]
L = int(os.environ.get("GM_STAMPS_READ_LEN", L))            # 150: the reads of the paired workload (cfg5), mapped unpaired here -- K1 is the same
contigs = synth.make_genome(synth.contig_lengths(gname, 1.0), gseed)
reads, _ = synth.make_reads(contigs, n, L, rseed)
if os.environ.get("GM_STAMPS_RANDOM_READS"):                # reads that map nowhere (no cluster of ~250 colinear hits in any read-strand): what the exact stages cost without one
    reads = np.random.Generator(np.random.PCG64(99)).integers(0, 4, size=reads.shape, dtype=np.uint8)
ix = gm.Index(contigs); s = gm.Session(ix, max_batch_reads=131072)
os.environ["GM_RAMP_MIN"] = "131072"
s.map_reads(reads[:8192])
lib = gm.lib()
out = (C.c_ulonglong * 16)()
has = hasattr(lib, "gm_debug_k5_stamps")
if has: lib.gm_debug_k5_stamps(out)
for nk in (os.environ.get("GM_STAMPS_NK_SWEEP", "").split(",") if os.environ.get("GM_STAMPS_NK_SWEEP") else [None]):
  if nk is not None:
    os.environ["GM_K5_NK"] = nk; print("---- GM_K5_NK=%s" % nk)
    if has: lib.gm_debug_k5_stamps(out)
  t = time.time(); s.map_reads(reads); dt = time.time() - t
  print("kernel", lib.gm_last_lookup_kernel().decode(), "map %.3fs" % dt, {k: v for k, v in s.stats.items() if k.startswith("ms_") or k in ("survivors", "survivors_pruned", "list_entries")})
  if has:
      lib.gm_debug_k5_stamps(out)
      v = [int(x) for x in out]; tot = sum(v) or 1
      # (profiles/r04h, r04i were printed by the sorted-exact-stage experiment of commit 60c0567: there stamp 10 = counting sort to bucket order, 3 = position order,
      # 4 = decisions + output)
      names = ["setup(+clear wait)", "pass A", "pass B", "region table", "rules+output", "bookkeeping+clear"]
      tot = sum(v[:6]) or 1
      for nm, x in zip(names, v): print("%-20s %6.2f %%  %8.0f ticks per read-strand" % (nm, 100.0 * x / tot, x / (2.0 * n)))
      for nm, x in zip(["  set-up: to the first barrier", "  set-up: k-mers ahead", "  region table: main loop", "  rules: main loop (2a)"], v[8:12]): print("%-32s %8.0f ticks per read-strand (not in the phase above)" % (nm, x / (2.0 * n)))
      print("candidates per read-strand %.1f, fallbacks %d of %d" % (v[6] / (2.0 * n), v[7], 2 * n))
      if v[12]: print("pass A, first of two runs (K5_ABL_A_TWICE: lists from HBM; the line 'pass A' above is then the second run, lists from the caches) %8.0f ticks per read-strand" % (v[12] / (2.0 * n)))
