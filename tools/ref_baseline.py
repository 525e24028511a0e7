#!/usr/bin/env python3
"""Reference gmapper vs the CPU restatement (oracle), timed in the BUILD container on BASELINE configs[0] and configs[1]
(SURVEY.md 8(d) item 1).  Needs oracle/_ref/gmapper-ls (make -f oracle/Makefile.ref) and /root/reference is
not read.  Index: gmapper-ls -S once, then -L; "Read Mapping Time" is taken from the reference's own statistics (ref: gmapper.c:800-804).
Writes profiles/r02_ref_baseline.json.

    python tools/ref_baseline.py [--threads 8] [--cfg2-reads 200000] [--skip-cfg2]
"""
import argparse, json, os, re, subprocess, sys, tempfile, time
import numpy as np
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
from shrimp_amd import synth
from tests import oracle_api as oa
from tools.make_golden import write_fa_codes
REF = os.path.join(ROOT, "oracle", "_ref", "gmapper-ls")


def run_ref(tmp, contigs, reads, threads):
    g = os.path.join(tmp, "g.fa"); r = os.path.join(tmp, "r.fa"); idx = os.path.join(tmp, "idx")
    write_fa_codes(g, [b"contig%d" % (i + 1) for i in range(len(contigs))], contigs)
    write_fa_codes(r, [b"r%d" % i for i in range(len(reads))], list(reads))
    t0 = time.time()
    subprocess.run([REF, "-S", idx, g], capture_output=True, check=True)          # builds and saves the index (no reads)
    t_build = time.time() - t0
    t0 = time.time()
    p = subprocess.run([REF, "-N", str(threads), "-L", idx, r], capture_output=True, check=True)
    t_total = time.time() - t0
    log = p.stderr.decode(errors="replace")
    m = re.search(r"Read Mapping Time:\s+([0-9.]+)", log) or re.search(r"Mapping Time[^0-9]*([0-9.]+)", log)
    t_map = float(m.group(1)) if m else None
    body = b"".join(l + b"\n" for l in p.stdout.split(b"\n") if l and not l.startswith(b"@"))
    return {"index_build_s": round(t_build, 2), "load_plus_map_s": round(t_total, 2), "read_mapping_time_s": t_map, "sam_records": body.count(b"\n")}, body, log


def run_oracle(contigs, reads, threads):
    t0 = time.time(); o = oa.Session(contigs); t_build = time.time() - t0
    t0 = time.time(); sam = o.map_sam(reads, nthreads=threads); t_map = time.time() - t0
    o.close()
    body = b"".join(l + b"\n" for l in sam.split(b"\n") if l and not l.startswith(b"@"))
    return {"index_build_s": round(t_build, 2), "map_s": round(t_map, 3), "sam_records": body.count(b"\n")}, body


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--threads", type=int, default=8)
    ap.add_argument("--cfg2-reads", type=int, default=200000, help="reads of configs[1] to time (the full 1 M take the reference ~1 min of mapping; the index build dominates)")
    ap.add_argument("--skip-cfg2", action="store_true")
    a = ap.parse_args()
    out = {"host": "build container: %d vCPU, %s" % (os.cpu_count(), open("/proc/cpuinfo").read().split("model name")[1].split("\n")[0].strip(": \t")), "threads": a.threads, "cases": {}}
    cases = [("cfg1", None)] + ([] if a.skip_cfg2 else [("cfg2", a.cfg2_reads)])
    for name, n in cases:
        gname, gseed, nr, L, rseed = synth.CONFIGS[name]
        contigs = synth.make_genome(synth.contig_lengths(gname, 1.0), gseed)
        reads, _ = synth.make_reads(contigs, n or nr, L, rseed)
        with tempfile.TemporaryDirectory(dir="/tmp") as tmp:
            ref, rbody, log = run_ref(tmp, contigs, reads, a.threads)
        orc, obody = run_oracle(contigs, reads, a.threads)
        t_ref = ref["read_mapping_time_s"] or ref["load_plus_map_s"]
        case = {"reads": int(len(reads)), "read_len": L, "genome_bp": int(sum(len(c) for c in contigs)), "reference": ref, "oracle": orc,
                "reference_reads_per_s": round(len(reads) / t_ref, 1), "oracle_reads_per_s": round(len(reads) / orc["map_s"], 1),
                "ref_over_oracle": round((len(reads) / t_ref) / (len(reads) / orc["map_s"]), 3), "sam_identical": rbody == obody}
        out["cases"][name] = case
        print(name, json.dumps(case), flush=True)
        if not ref["read_mapping_time_s"]: print(log[-1500:])
    json.dump(out, open(os.path.join(ROOT, "profiles", "r02_ref_baseline.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
