"""The bench's FASTA-file leg alone: N x 1 M cfg3 reads through gm_map_reads_file_cb, with one session (GM_FILE_ONE_SESSION=1) and with the twin.
usage (GPU box): python tools/file_leg.py [millions of reads, default 2]"""
import os, sys, time, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from shrimp_amd import gmapper as gm, synth
M = int(sys.argv[1]) if len(sys.argv) > 1 else 2
gname, gseed, _, L, rseed = synth.CONFIGS["cfg3"]
contigs = synth.make_genome(synth.contig_lengths(gname, 1.0), gseed)
reads, _ = synth.make_reads(contigs, 1_000_000, L, rseed)
ix = gm.Index(contigs); s = gm.Session(ix, max_batch_reads=131072)
lut = np.frombuffer(b"ACGTNNNNNNNNNNNN", dtype=np.uint8)
lines = np.empty((len(reads), L + 1), dtype=np.uint8); lines[:, :L] = lut[reads]; lines[:, L] = 10
body = lines.tobytes()[:-1].split(b"\n")
nm = np.char.add(np.char.add(">r", np.arange(len(reads)).astype("U8")), "\n").astype("S")
with tempfile.TemporaryDirectory() as td:
    fpath = os.path.join(td, "reads.fa")
    with open(fpath, "wb") as f:
        for rep in range(M): f.write(b"".join(a + b + b"\n" for a, b in zip(nm.tolist(), body)))
    for mode in ("1", "", "1", ""):
        if mode: os.environ["GM_FILE_ONE_SESSION"] = mode
        else: os.environ.pop("GM_FILE_ONE_SESSION", None)
        s.map_reads_file_chunks(fpath, collect=False) if mode == "1" and "warm" not in globals() else None
        warm = True
        t0 = time.perf_counter(); parts = s.map_reads_file_chunks(fpath, collect=False); dt = time.perf_counter() - t0
        print("one session" if mode else "twin sessions", "%d reads in %.1f ms = %.3f M reads/s, %d chunks, %d bytes of SAM" % (M * len(reads), dt * 1e3, M * len(reads) / dt / 1e6, len(parts), sum(parts)), flush=True)
