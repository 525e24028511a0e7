#!/usr/bin/env python3
"""Reference gmapper vs the CPU restatement (oracle), timed in the BUILD container on BASELINE configs[0] and configs[1]
(SURVEY.md 8(d) item 1).  Needs oracle/_ref/gmapper-ls (make -f oracle/Makefile.ref) and /root/reference is
not read.  Index: gmapper-ls -S once, then -L; "Read Mapping Time" is taken from the reference's own statistics (ref: gmapper.c:800-804).
Writes profiles/r03_ref_baseline.json.

    python tools/ref_baseline.py [--threads 8] [--cfg2-reads 200000] [--skip-cfg2] [--cfg3-group 4 --cfg3-reads 100000]

--cfg3-group G: the 3 Gbp configuration the way BASELINE.md section 3 item 4 prescribes (the reference's own practice for genomes beyond RAM,
SPLITTING_AND_MERGING:21-98): ONE contig group of the 24-contig genome (its first G contigs, 125 Mbp each), a fixed subsample of the cfg3 reads
(drawn from the WHOLE genome, so most do not belong to the group -- as in a real split run) mapped against it by the reference and by the port.
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
    ap.add_argument("--skip-cfg1", action="store_true")
    ap.add_argument("--cfg3-group", type=int, default=0, help="contigs of the 3 Gbp genome in the one group that is measured (0 = skip)")
    ap.add_argument("--cfg3-reads", type=int, default=100000)
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r03_ref_baseline.json"))
    a = ap.parse_args()
    out = {"host": "build container: %d vCPU, %s" % (os.cpu_count(), open("/proc/cpuinfo").read().split("model name")[1].split("\n")[0].strip(": \t")), "threads": a.threads, "cases": {}}
    if os.path.exists(a.out):
        try: out["cases"] = json.load(open(a.out))["cases"]
        except Exception: pass
    cases = ([] if a.skip_cfg1 else [("cfg1", None)]) + ([] if a.skip_cfg2 else [("cfg2", a.cfg2_reads)]) + ([("cfg3_group", a.cfg3_reads)] if a.cfg3_group else [])
    for name, n in cases:
        gname, gseed, nr, L, rseed = synth.CONFIGS["cfg3" if name == "cfg3_group" else name]
        contigs = synth.make_genome(synth.contig_lengths(gname, 1.0), gseed)
        reads, _ = synth.make_reads(contigs, n or nr, L, rseed)
        if name == "cfg3_group":
            whole = int(sum(len(c) for c in contigs)); contigs = contigs[:a.cfg3_group]
        with tempfile.TemporaryDirectory(dir="/tmp") as tmp:
            ref, rbody, log = run_ref(tmp, contigs, reads, a.threads)
        orc, obody = run_oracle(contigs, reads, a.threads)
        t_ref = ref["read_mapping_time_s"] or ref["load_plus_map_s"]
        case = {"reads": int(len(reads)), "read_len": L, "genome_bp": int(sum(len(c) for c in contigs)), "reference": ref, "oracle": orc,
                "reference_reads_per_s": round(len(reads) / t_ref, 1), "oracle_reads_per_s": round(len(reads) / orc["map_s"], 1),
                "ref_over_oracle": round((len(reads) / t_ref) / (len(reads) / orc["map_s"]), 3), "sam_identical": rbody == obody}
        if name == "cfg3_group":
            case["what"] = ("reference gmapper-ls -N %d vs the port: ONE contig group (%d of 24 contigs, %d of %d bp) of the 3 Gbp genome, %d cfg3 reads drawn from the whole genome "
                            "(BASELINE.md section 3 item 4); a whole-genome figure would sum %d such groups" % (a.threads, a.cfg3_group, case["genome_bp"], whole, len(reads), (24 + a.cfg3_group - 1) // a.cfg3_group))
        else:
            case["what"] = "reference gmapper-ls -N %d vs the port on the same reads" % a.threads
        out["cases"][name] = case
        print(name, json.dumps(case), flush=True)
        if not ref["read_mapping_time_s"]: print(log[-1500:])
        json.dump(out, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
