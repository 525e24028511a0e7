#!/usr/bin/env python3
"""Golden fixtures for the merge step, from the REFERENCE programs built by oracle/Makefile.ref:

  inputs   oracle/_ref/gmapper-{ls,cs} on shards (contig groups / read halves) of committed golden inputs
  outputs  oracle/_ref/mergesam on those SAM files, one run per option set

tests/golden/merge/<case>.in<k>.sam.gz    a shard's SAM as the reference's gmapper wrote it (with its @PG line: mergesam renumbers them)
tests/golden/merge/<case>.reads.gz        the reads file given to mergesam
tests/golden/merge/<case>@<set>.out.gz    mergesam's output for the option set (SAM incl. header, or the --un / --al text)
tests/golden/merge/cases.json             case -> reads, inputs, option sets (argument lists and the command line that went into @PG)

Only data is stored: program outputs and inputs.  Run from the repo root after `make -f oracle/Makefile.ref`."""
import gzip, json, os, subprocess, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import make_golden as mg
from shrimp_amd import synth
REF = os.path.join(ROOT, "oracle", "_ref"); OUT = os.path.join(ROOT, "tests", "golden", "merge")
G = os.path.join(ROOT, "tests", "golden")

OPTION_SETS = {
    "default": [], "all_contigs": ["--all-contigs"], "single_best_all": ["--single-best-mapping", "--all-contigs"], "single_best": ["--single-best-mapping"],
    "strata": ["--strata"], "o3": ["-o", "3"], "o3_max5": ["-o", "3", "--max-alignments", "5"], "max4": ["--max-alignments", "4"],
    "unal": ["--sam-unaligned"], "unal_single_best_all": ["--sam-unaligned", "--single-best-mapping", "--all-contigs"],
    "no_mapq": ["--no-mapping-qualities"], "leave_mapq": ["--no-mapping-qualities", "--leave-mapq-untouched"],
    "all_min20": ["--all-contigs", "--min-mapq", "20"], "single_best_all_min30": ["--single-best-mapping", "--all-contigs", "--min-mapq", "30"],
    "no_half_paired": ["--no-half-paired"], "no_half_paired_unal": ["--no-half-paired", "--sam-unaligned"],
    "no_improper": ["--single-best-mapping", "--all-contigs", "--no-improper-mappings"], "strata_o2_unal": ["--strata", "-o", "2", "--sam-unaligned"],
    "un": ["--un"], "al": ["--al"], "un_single_best_all": ["--un", "--single-best-mapping", "--all-contigs"],
}
ALL = list(OPTION_SETS)
FEW = ["default", "single_best_all", "unal", "un", "al"]


def gz(path, data):
    with gzip.open(path, "wb", compresslevel=9) as f:
        f.write(data)


def run(args, cwd):
    p = subprocess.run(args, cwd=cwd, capture_output=True)
    if p.returncode != 0:
        print(p.stderr.decode()[-2000:]); raise SystemExit("failed: " + " ".join(args))
    return p.stdout


def merge_case(cases, name, d, reads_file, sam_files, sets):
    """sam_files / reads_file: names inside the scratch directory d (mergesam runs there, so that @PG holds no scratch path)"""
    gz(os.path.join(OUT, name + ".reads.gz"), open(os.path.join(d, reads_file), "rb").read())
    for k, s in enumerate(sam_files):
        gz(os.path.join(OUT, "%s.in%d.sam.gz" % (name, k)), open(os.path.join(d, s), "rb").read())
    entry = {"reads": reads_file, "inputs": sam_files, "sets": {}}
    for st in sets:
        a = list(OPTION_SETS[st]); fx = a and a[0] in ("--un", "--al")
        argv = ["mergesam"] + ([a[0], "out.fx"] + a[1:] if fx else ["--sam"] + a) + [reads_file] + sam_files
        out = run([os.path.join(REF, "mergesam")] + argv[1:], d)
        if fx: out = open(os.path.join(d, "out.fx"), "rb").read()
        else:   # the program name in @PG is the path it was started with: keep the fixture free of it
            out = out.replace(b"CL:" + os.path.join(REF, "mergesam").encode() + b" ", b"CL:mergesam ")
        gz(os.path.join(OUT, "%s@%s.out.gz" % (name, st)), out)
        entry["sets"][st] = {"args": a, "command_line": " ".join(argv) + " "}
    cases[name] = entry
    print(name, len(sets), "option sets")


def main():
    os.makedirs(OUT, exist_ok=True)
    cases = {}
    ls, cs = os.path.join(REF, "gmapper-ls"), os.path.join(REF, "gmapper-cs")
    with tempfile.TemporaryDirectory() as d:
        w = lambda n, data: open(os.path.join(d, n), "wb").write(data)
        # ---- (1) unpaired letter space, genome in two contig groups, with and without unaligned records ----
        z = np.load(os.path.join(G, "stress_60bp.npz"))
        contigs = [z["contig%d" % i] for i in range(4)]; names = [b"contig%d" % (i + 1) for i in range(4)]
        mg.write_fa_codes(os.path.join(d, "db1.fa"), names[:1], contigs[:1]); mg.write_fa_codes(os.path.join(d, "db2.fa"), names[1:], contigs[1:])
        mg.write_fa_codes(os.path.join(d, "db.fa"), names, contigs)
        reads = z["reads"][:1400]
        mg.write_fa_codes(os.path.join(d, "qr.fa"), [b"r%d" % i for i in range(len(reads))], list(reads))
        for k in (1, 2):
            w("map-db%d.sam" % k, run([ls, "-N", "4", "qr.fa", "db%d.fa" % k], d))
            w("mapu-db%d.sam" % k, run([ls, "-N", "4", "--sam-unaligned", "qr.fa", "db%d.fa" % k], d))
        merge_case(cases, "ls_db2", d, "qr.fa", ["map-db1.sam", "map-db2.sam"], ALL)
        merge_case(cases, "ls_db2_unal", d, "qr.fa", ["mapu-db1.sam", "mapu-db2.sam"], ["default", "unal", "unal_single_best_all", "strata_o2_unal", "un", "al", "un_single_best_all"])
        # ---- (2) read halves x contig groups in one step, and read halves of one genome with the MAPQ left alone (SPLITTING_AND_MERGING:60-148) ----
        h = len(reads) // 2
        mg.write_fa_codes(os.path.join(d, "qr-1of2.fa"), [b"r%d" % i for i in range(h)], list(reads[:h]))
        mg.write_fa_codes(os.path.join(d, "qr-2of2.fa"), [b"r%d" % i for i in range(h, len(reads))], list(reads[h:]))
        four = []
        for q in (1, 2):
            for k in (1, 2):
                n = "map-qr%dof2-db%dof2.sam" % (q, k); w(n, run([ls, "-N", "4", "qr-%dof2.fa" % q, "db%d.fa" % k], d)); four.append(n)
            w("map-qr%dof2.sam" % q, run([ls, "-N", "4", "qr-%dof2.fa" % q, "db.fa"], d))
        merge_case(cases, "ls_qr2_db2", d, "qr.fa", four, ["default", "single_best_all", "all_contigs"])
        merge_case(cases, "ls_qr2", d, "qr.fa", ["map-qr1of2.sam", "map-qr2of2.sam"], ["leave_mapq", "no_mapq", "default"])
        # ---- (3) pairs ----
        zp = np.load(os.path.join(G, "stress_pairs_2x100.npz"))
        NP = 500
        nm = [n for pair in zip(zp["names1"][:NP], zp["names2"][:NP]) for n in pair]; sq = [q for pair in zip(list(zp["mates1"][:NP]), list(zp["mates2"][:NP])) for q in pair]
        mg.write_fa_codes(os.path.join(d, "pr.fa"), nm, sq)
        for k in (1, 2):
            w("pmap-db%d.sam" % k, run([ls, "-N", "4", "-p", "opp-in", "-I", "100,600", "pr.fa", "db%d.fa" % k], d))
            w("pmapu-db%d.sam" % k, run([ls, "-N", "4", "--sam-unaligned", "-p", "opp-in", "-I", "100,600", "pr.fa", "db%d.fa" % k], d))
        merge_case(cases, "pairs_db2", d, "pr.fa", ["pmap-db1.sam", "pmap-db2.sam"], ALL)
        merge_case(cases, "pairs_db2_unal", d, "pr.fa", ["pmapu-db1.sam", "pmapu-db2.sam"], ["default", "unal", "unal_single_best_all", "no_half_paired_unal", "un", "al"])
        # ---- (4) colour-space FASTQ reads (CS / CQ tags, the FASTQ name parser) ----
        zc = np.load(os.path.join(G, "cfg4s_50col_2Mbp.npz")); zq = np.load(os.path.join(G, "cfg4s_50col_fq.npz"))
        cc = [zc["contig%d" % i] for i in range(sum(1 for f in zc.files if f.startswith("contig") and f[6:].isdigit()))]
        cr = zc["reads"][:600]; q = zq["quals"][:600]
        half = len(cc) // 2
        mg.write_fa_codes(os.path.join(d, "cdb1.fa"), [b"contig%d" % (i + 1) for i in range(half)], cc[:half])
        mg.write_fa_codes(os.path.join(d, "cdb2.fa"), [b"contig%d" % (i + 1) for i in range(half, len(cc))], cc[half:])
        tab = np.full(16, ord("."), dtype=np.uint8); tab[:4] = np.frombuffer(b"0123", dtype=np.uint8)
        with open(os.path.join(d, "cr.csfastq"), "wb") as f:
            for i in range(len(cr)):
                f.write(b"@r%d extra words\n" % i + b"ACGT"[cr[i, 0]:cr[i, 0] + 1] + tab[cr[i, 1:]].tobytes() + b"\n+\n" + q[i].tobytes() + b"\n")
        for k in (1, 2):
            w("cmap-db%d.sam" % k, run([cs, "-N", "4", "--sam-unaligned", "cr.csfastq", "cdb%d.fa" % k], d))
        merge_case(cases, "cs_fq_db2", d, "cr.csfastq", ["cmap-db1.sam", "cmap-db2.sam"], FEW + ["single_best", "strata"])
    with open(os.path.join(OUT, "cases.json"), "w") as f:
        json.dump(cases, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
