#!/usr/bin/env python3
"""Generate tests/golden/* from the REFERENCE binary built by oracle/Makefile.ref.

Runs only in the build container (needs /root/reference).  Fixtures are data only:
inputs (2-bit/4-bit packed codes in .npz) and the reference's SAM output (gz, @PG line dropped
because it embeds the command line, gmapper/gmapper.c:3007), plus known-answer records from the
reference's own sw_vector()/sw_full_ls() (oracle/ref_kat.cpp).

    make -f oracle/Makefile.ref && python tools/make_golden.py
"""
import gzip, os, subprocess, sys, tempfile
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from shrimp_amd import synth

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
REF = os.path.join(ROOT, "oracle", "_ref", "gmapper-ls")
OUT = os.path.join(ROOT, "tests", "golden")


def stress_genome(seed=77):
    """Small repeat-rich genome: tandem repeats (lists beyond the cutoff), dispersed copies of a
    family with a few mutations, homopolymers, N runs, IUPAC codes, a contig shorter than a window."""
    rng = np.random.Generator(np.random.PCG64(seed))
    def rnd(n): return rng.integers(0, 4, size=n, dtype=np.uint8)
    fam = rnd(400)
    c1 = [rnd(30000)]
    for k in range(25):
        f = fam.copy()
        m = rng.integers(0, 400, size=4); f[m] = (f[m] + 1) & 3
        c1 += [f, rnd(int(rng.integers(50, 3000)))]
    c1 += [np.tile(np.array([0, 1, 2, 3, 3, 2, 1, 0, 2], dtype=np.uint8), 1500), rnd(20000)]   # 9-mer tandem x1500
    c1 += [np.zeros(300, dtype=np.uint8), rnd(5000), np.full(120, 15, dtype=np.uint8), rnd(40000)]
    c1 = np.concatenate(c1)
    c2 = rnd(120000)
    c2[5000:5400] = fam                      # family copy on another contig
    c2[60000:60010] = 15
    iu = rng.integers(0, 120000, size=40); c2[iu] = rng.integers(4, 15, size=40).astype(np.uint8)
    c3 = rnd(70)                             # shorter than a 100bp read's window
    c4 = np.concatenate([rnd(50000), c1[2000:2600], rnd(3000), c1[2000:2600][::-1].copy(), rnd(20000)])
    return [c1, c2, c3, c4]


def stress_reads(contigs, n, L, seed=78):
    reads, _ = synth.make_reads([c for c in contigs if len(c) > 2 * L + 40], n, L, seed, p_sub=0.03, p_ins=0.004, p_del=0.004)
    rng = np.random.Generator(np.random.PCG64(seed + 5))
    # the source contigs may contain codes > 3; clamp sampled IUPAC to N for reads and sprinkle extra Ns
    reads = np.where(reads > 3, 15, reads).astype(np.uint8)
    k = rng.integers(0, n, size=n // 20)
    reads[k, rng.integers(0, L, size=k.size)] = 15
    return reads


def write_fa_codes(path, names, seqs):
    T = np.frombuffer(b"ACGTUMRWSYKVHDBN", dtype=np.uint8)
    with open(path, "wb") as f:
        for nm, s in zip(names, seqs):
            f.write(b">" + nm + b"\n")
            t = T[s]
            for k in range(0, len(t), 70):
                f.write(t[k:k + 70].tobytes() + b"\n")


def run_case(name, contigs, reads, extra=()):
    with tempfile.TemporaryDirectory() as d:
        g = os.path.join(d, "g.fa"); r = os.path.join(d, "r.fa")
        write_fa_codes(g, [b"contig%d" % (i + 1) for i in range(len(contigs))], contigs)
        write_fa_codes(r, [b"r%d" % i for i in range(len(reads))], list(reads))
        p = subprocess.run([REF, "-N", "4", *extra, r, g], capture_output=True, check=True)
        body = b"".join(l + b"\n" for l in p.stdout.split(b"\n") if l and not l.startswith(b"@PG"))
    np.savez_compressed(os.path.join(OUT, name + ".npz"),
                        **{"contig%d" % i: c for i, c in enumerate(contigs)}, reads=reads)
    with gzip.open(os.path.join(OUT, name + ".sam.gz"), "wb", compresslevel=9) as f:
        f.write(body)
    n_map = sum(1 for l in body.split(b"\n") if l and not l.startswith(b"@"))
    print(f"{name}: {len(reads)} reads -> {n_map} SAM records")


def n1_noisy_case():
    """-n 1 on noisy reads (9 % substitutions): reads whose only evidence is ONE k-mer match still map -- no region counts, a window per anchor (gmapper.c:2610-2625)"""
    contigs = synth.make_genome([400000, 250000], 77)
    reads, _ = synth.make_reads(contigs, 4000, 70, 78, p_sub=0.09, p_ins=0.004, p_del=0.004)
    run_case("n1_noisy_70bp", contigs, reads, extra=("-n", "1"))
    # reads built to have exactly ONE list entry: a 14-base block (one placement of the span-14 default seed) left intact, a substitution next to it on either side
    # and every 4th base from there on; -h 30% lets their alignments through.  390 of 400 map with -n 1, none without it.
    rng = np.random.default_rng(5); L = 60
    one = np.empty((400, L), dtype=np.uint8)
    for i in range(len(one)):
        c = contigs[rng.integers(0, len(contigs))]
        p = int(rng.integers(0, len(c) - L)); r = c[p:p + L].copy()
        b = int(rng.integers(0, L - 14 + 1))
        for j in list(range(b - 1, -1, -4)) + list(range(b + 14, L, 4)): r[j] = (r[j] + 1 + rng.integers(0, 3)) & 3
        if rng.random() < 0.5: r = (3 - r[::-1]).astype(np.uint8)
        one[i] = r
    run_case("n1_onehit_60bp", contigs, one, extra=("-n", "1", "-h", "30%"))


def mirna_case():
    """-M mirna: the mode's option bundle (gmapper.c:1497-1517: -H, -U, anchor width 0, gap opens -255, no f1 cache, -n 1, window 100 %, --local, no mapping
    qualities) with its five default seeds of span 20 / weight 14 (leading and trailing zeros, gmapper-defaults.h:230-238) on 22-base reads"""
    contigs = synth.make_genome([300000, 200000], 91)
    reads, _ = synth.make_reads(contigs, 3000, 22, 92, p_sub=0.04, p_ins=0.0, p_del=0.0)
    run_case("mirna_22bp", contigs, reads, extra=("-M", "mirna"))


def read_fa(path):
    names, seqs = [], []
    for line in open(path, "rb"):
        line = line.strip()
        if not line: continue
        if line.startswith(b">"): names.append(line[1:].split()[0]); seqs.append(b"")
        else: seqs[-1] += line
    return names, seqs


CODE = {c: i for i, c in enumerate(b"ACGTUMRWSYKVHDBN")}


def to_codes(seq):
    return np.array([CODE[c] for c in seq.upper()], dtype=np.uint8)


def run_pair_case(name, contigs, contig_names, m1, m2, names1, names2, mode, ins):
    """mates adjacent in one file, as gmapper expects in paired mode (ref: gmapper.c:2319-2322)"""
    with tempfile.TemporaryDirectory() as d:
        g = os.path.join(d, "g.fa"); r = os.path.join(d, "r.fa")
        write_fa_codes(g, contig_names, contigs)
        names = [n for pair in zip(names1, names2) for n in pair]
        seqs = [q for pair in zip(list(m1), list(m2)) for q in pair]
        write_fa_codes(r, names, seqs)
        p = subprocess.run([REF, "-N", "4", "-p", mode, "-I", "%d,%d" % ins, r, g], capture_output=True, check=True)
        body = b"".join(l + b"\n" for l in p.stdout.split(b"\n") if l and not l.startswith(b"@PG"))
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **{"contig%d" % i: c for i, c in enumerate(contigs)},
                        mates1=np.asarray(m1), mates2=np.asarray(m2),
                        names1=np.array(names1), names2=np.array(names2), contig_names=np.array(contig_names),
                        mode=np.array(mode), ins=np.array(ins))
    with gzip.open(os.path.join(OUT, name + ".sam.gz"), "wb", compresslevel=9) as f:
        f.write(body)
    n_rec = sum(1 for l in body.split(b"\n") if l and not l.startswith(b"@"))
    print("%s: %d pairs -> %d SAM records" % (name, len(m1), n_rec))


def paired_cases():
    # (1) the reference's own pairing fixture (not_in_dist/test_pairing): 24 pairs, mates of 30 and 50 bp, under all four pair modes
    fx = "/root/reference/not_in_dist/test_pairing"
    gn, gs = read_fa(os.path.join(fx, "reference-pairing.fa"))
    rn, rs = read_fa(os.path.join(fx, "reads-pairing.fa"))
    contigs = [to_codes(q) for q in gs]
    m1 = np.stack([to_codes(q) for q in rs[0::2]]); m2 = np.stack([to_codes(q) for q in rs[1::2]])
    for mode in ("opp-in", "opp-out", "col-fw", "col-bw"):
        run_pair_case("pairfix_" + mode, contigs, gn, m1, m2, rn[0::2], rn[1::2], mode, (0, 500))
    # (2) cfg5-like: 2 x 150 bp opp-in pairs, insert ~ N(300, 30), 1 % substitutions, uniform genome
    contigs = synth.make_genome([600_000, 400_000], 55)
    reads, _ = synth.make_pairs(contigs, 1500, 150, 5)
    n = len(reads) // 2
    run_pair_case("cfg5s_2x150_1Mbp", contigs, [b"contig1", b"contig2"], reads[0::2], reads[1::2],
                  [b"p%d/1" % i for i in range(n)], [b"p%d/2" % i for i in range(n)], "opp-in", (100, 600))
    # (3) pairs on the repeat-rich stress genome (half-paired rescue, duplicate pruning, unpaired mates)
    sg = stress_genome()
    r2, _ = synth.make_pairs([c for c in sg if len(c) > 2000], 1200, 100, 6, ins_mean=250, ins_sd=40, ins_min=120, p_sub=0.02)
    r2 = np.where(r2 > 3, 15, r2).astype(np.uint8)
    n = len(r2) // 2
    run_pair_case("stress_pairs_2x100", sg, [b"contig%d" % (i + 1) for i in range(len(sg))], r2[0::2], r2[1::2],
                  [b"q%d:1" % i for i in range(n)], [b"q%d:2" % i for i in range(n)], "opp-in", (100, 600))


def chimeric_pairs_case():
    """pairs whose mates come from different places (every third pair of the cfg5-like set takes the second mate of another pair): both mates map on their own, never
    as a pair -- what --single-best-mapping --all-contigs turns into IMPROPER pairs (output.c:1178-1226); the default run is the base golden"""
    z = np.load(os.path.join(OUT, "cfg5s_2x150_1Mbp.npz"))
    contigs = [z["contig%d" % i] for i in range(sum(1 for f in z.files if f.startswith("contig") and f[6:].isdigit()))]
    N = 600
    m1 = z["mates1"][:N].copy(); m2 = z["mates2"][:N].copy()
    src = np.arange(N); src[0::3] = (src[0::3] + 301) % N
    m2 = m2[src]
    run_pair_case("chimeric_pairs_2x150", contigs, [bytes(x) for x in z["contig_names"]], m1, m2, [b"c%d/1" % i for i in range(N)], [b"c%d/2" % i for i in range(N)], "opp-in", (100, 600))


OPTION_CASES = {
    # tag: (base golden whose inputs are reused, reference command-line options)
    "chim_single_best_all": ("chimeric_pairs_2x150", ["--single-best-mapping", "--all-contigs"]),
    "chim_single_best_all_noimp": ("chimeric_pairs_2x150", ["--single-best-mapping", "--all-contigs", "--no-improper-mappings"]),
    "chim_single_best": ("chimeric_pairs_2x150", ["--single-best-mapping"]),
    "strata":    ("stress_60bp", ["--strata"]),
    "max3_o5":   ("stress_60bp", ["--max-alignments", "3", "-o", "5"]),
    "scores":    ("stress_100bp_unal", ["-m", "8", "-i", "-12", "-g", "-30", "-q", "-28", "-e", "-5", "-f", "-4", "-h", "60%", "-w", "150%",
                                        "-l", "80%", "-r", "50%", "-o", "6", "-a", "10", "--sam-unaligned"]),
    "seeds":     ("cfg2s_100bp_2Mbp", ["-s", "1111101111,110110110110111,1110100111010111", "-z", "40"]),
    "pairs_strata": ("stress_pairs_2x100", ["--strata", "-o", "4"]),
    "local":     ("stress_100bp_unal", ["--local", "--sam-unaligned"]),
    "local60":   ("stress_60bp", ["--local", "-h", "40%"]),
    "local_cfg2": ("cfg2s_100bp_2Mbp", ["--local"]),
    "ungapped":  ("stress_100bp_unal", ["--local", "-U", "--sam-unaligned"]),
    "ungapped60_n1": ("stress_60bp", ["--local", "-U", "-n", "1", "-h", "45%"]),
    # output policy of read_output / readpair_output (output.c:955-1008,1070-1291): the best mapping only, per class or over all classes (with an improper pair of two
    # unpaired mappings when both are good), no Z tags with --all-contigs, no mapping qualities at all
    "single_best": ("stress_60bp", ["--single-best-mapping"]),
    "all_contigs": ("stress_60bp", ["--all-contigs"]),
    "no_mapq": ("stress_100bp_unal", ["--no-mapping-qualities", "--sam-unaligned"]),
    "pairs_single_best": ("stress_pairs_2x100", ["--single-best-mapping"]),
    "pairs_single_best_all": ("stress_pairs_2x100", ["--single-best-mapping", "--all-contigs"]),
    "pairs_single_best_all_noimp": ("stress_pairs_2x100", ["--single-best-mapping", "--all-contigs", "--no-improper-mappings"]),
    "pairs_no_mapq": ("stress_pairs_2x100", ["--no-mapping-qualities"]),
    # the optional tail of the SAM records (output.c:452-465,729-756): --extra-sam-fields (ZM / ZR / ZV / ZH / ZE), --read-group, --sam-r2 (paired mode)
    # (inputs without N: the reference's reverse_alignment_edit_string never returns on a letter it does not know -- its assert(0) is compiled out)
    "extra_fields": ("cfg2s_100bp_2Mbp", ["--extra-sam-fields"]),
    "extra_fields_rg_unal": ("n1_noisy_70bp", ["--extra-sam-fields", "--read-group", "grp1,sampleA", "--sam-unaligned"]),
    "rg_unal": ("stress_100bp_unal", ["--read-group", "grp1,sampleA", "--sam-unaligned"]),
    "pairs_r2_rg_extra": ("cfg5s_2x150_1Mbp", ["--sam-r2", "--read-group", "grp1,sampleA", "--extra-sam-fields"]),
    "pairs_r2_rg": ("stress_pairs_2x100", ["--sam-r2", "--read-group", "grp1,sampleA"]),
    # region geometry of the k-mer hit counts (--region-bits, --region-overlap)
    "regions_10_30": ("stress_60bp", ["--region-bits", "10", "--region-overlap", "30"]),
    "regions_12_200": ("cfg2s_100bp_2Mbp", ["--region-bits", "12", "--region-overlap", "200"]),
    # -t: the full-SW tie-breaks are not reversed for hits on the negative strand (Tflag off; mapping.c:378,393)
    "tiebreak_off": ("stress_100bp_unal", ["-t", "--sam-unaligned"]),
    "pairs_tiebreak_off": ("stress_pairs_2x100", ["-t"]),
    # -F / -C: only the read as given / only its reverse complement
    "positive": ("stress_60bp", ["-F"]),
    "negative": ("stress_60bp", ["-C"]),
    # -n 1 on its own: no region counts at all, every list entry becomes an anchor and every anchor a window (gmapper.c:2610-2624)
    "n1":        ("stress_60bp", ["-n", "1"]),
    "hashed":    ("stress_60bp", ["-H"]),
    "hashed_w16": ("cfg2s_100bp_2Mbp", ["-H", "-s", "11111111101111111,1111110111011101111,111101110010000101111011"]),
    "pairs_hashed": ("stress_pairs_2x100", ["-H", "-o", "3"]),
    "pairs_local": ("stress_pairs_2x100", ["--local"]),
    "pairs_ungapped": ("stress_pairs_2x100", ["--local", "-U"]),
    # --no-half-paired: mate-pair region counts in the anchor lists (use_mp_region_counts = 1), no unpaired rescue
    "no_half_paired": ("stress_pairs_2x100", ["--no-half-paired"]),
    "cfg5_no_half_paired": ("cfg5s_2x150_1Mbp", ["--no-half-paired"]),
    # -n 3 in paired mode: a region marked once counts when the mate has hits within reach (use_mp_region_counts 2, or 3 with --no-half-paired; hit list mode 3,
    # gmapper.c:2657-2673); -n 2 in paired mode: no region counts at all, a window per anchor
    "pairs_n3": ("stress_pairs_2x100", ["-n", "3"]),
    "pairs_n3_nhp": ("stress_pairs_2x100", ["-n", "3", "--no-half-paired"]),
    "cfg5_n3": ("cfg5s_2x150_1Mbp", ["-n", "3"]),
    "cfg5_n3_nhp": ("cfg5s_2x150_1Mbp", ["-n", "3", "--no-half-paired"]),
    "pairs_n2": ("stress_pairs_2x100", ["-n", "2"]),
    "pairs_hashed_n3": ("stress_pairs_2x100", ["-H", "-n", "3"]),                       # hashed seeds under the mate-pair modes of the generic lookup kernel
    "pairs_local_n3_nhp": ("stress_pairs_2x100", ["--local", "-n", "3", "--no-half-paired"]),
    "cfg5_n2": ("cfg5s_2x150_1Mbp", ["-n", "2"]),
}


# colour-space option sets (gmapper-cs): inputs of a committed colour-space golden, the reference's SAM body as <base>@<tag>.sam.gz
CS_OPTION_CASES = {
    "cs_local":        ("cfg4s_50col_2Mbp", ["--local"]),
    "cs_local_unal":   ("stress_cs_60col_unal", ["--local", "--sam-unaligned"]),
    "cs_ungapped":     ("cfg4s_50col_2Mbp", ["--local", "-U"]),
    "cs_ungapped_unal": ("stress_cs_60col_unal", ["--local", "-U", "--sam-unaligned", "-h", "40%"]),
    "cs_tiebreak_off": ("stress_cs_60col_unal", ["-t", "--sam-unaligned"]),
    "cs_xover_taboo":  ("stress_cs_60col_unal", ["-x", "-25", "--indel-taboo-len", "3", "--pr-xover", "0.05", "--sam-unaligned"]),
    "cs_no_mapq":      ("cfg4s_50col_2Mbp", ["--no-mapping-qualities"]),         # global sw_full_cs, no post_sw: sw_full_cs's own strings and counts in the output
    "cs_single_best":  ("stress_cs_60col_unal", ["--single-best-mapping", "--sam-unaligned"]),
    "cs_extra_rg":     ("cfg4s_50col_2Mbp", ["--extra-sam-fields", "--read-group", "grp1,sampleA", "--sam-unaligned"]),
}


def cs_option_cases(only=None):
    for tag, (base, extra) in CS_OPTION_CASES.items():
        if only and tag not in only: continue
        z = np.load(os.path.join(OUT, base + ".npz"))
        contigs = [z["contig%d" % i] for i in range(sum(1 for f in z.files if f.startswith("contig") and f[6:].isdigit()))]
        with tempfile.TemporaryDirectory() as d:
            g = os.path.join(d, "g.fa"); r = os.path.join(d, "r.csfasta")
            write_fa_codes(g, [b"contig%d" % (i + 1) for i in range(len(contigs))], contigs)
            synth.write_csfasta_reads(r, z["reads"])
            p = subprocess.run([os.path.join(ROOT, "oracle", "_ref", "gmapper-cs"), "-N", "4", *extra, r, g], capture_output=True, check=True)
            body = b"".join(l + b"\n" for l in p.stdout.split(b"\n") if l and not l.startswith(b"@PG") and not l.startswith(b"@RG"))
        with gzip.open(os.path.join(OUT, "%s@%s.sam.gz" % (base, tag)), "wb", compresslevel=9) as f:
            f.write(body)
        print("%s@%s: %d SAM records" % (base, tag, sum(1 for l in body.split(b"\n") if l and not l.startswith(b"@"))))


CS_PAIR_OPTION_CASES = {        # tag -> (base colour-space pair golden, extra gmapper-cs options)
    "cs_pairs_local": ("cs_pairs_50col_opp-in", ["--local"]),
    "cs_pairs_local_colbw": ("cs_pairs_50col_col-bw", ["--local"]),
    "cs_pairs_n3": ("cs_pairs_50col_opp-in", ["-n", "3"]),                                # paired match mode 3 in colour space; with --no-half-paired on a mode that reverses a mate
    "cs_pairs_n3_colbw_nhp": ("cs_pairs_50col_col-bw", ["-n", "3", "--no-half-paired"]),
    "cs_pairs_n2": ("cs_pairs_50col_opp-in", ["-n", "2"]),
    "cs_pairs_r2_extra": ("cs_pairs_50col_col-bw", ["--sam-r2", "--extra-sam-fields"]),
}


def cs_pair_option_cases(only=None):
    """non-default options on the committed colour-space pairs (mates adjacent in one csfasta file, as cs_pair_case writes them): only the reference's SAM body is stored"""
    for tag, (base, extra) in CS_PAIR_OPTION_CASES.items():
        if only and tag not in only: continue
        z = np.load(os.path.join(OUT, base + ".npz"))
        contigs = [z["contig%d" % i] for i in range(sum(1 for f in z.files if f.startswith("contig") and f[6:].isdigit()))]
        m1, m2 = z["mates1"], z["mates2"]; n = len(m1)
        with tempfile.TemporaryDirectory() as d:
            g = os.path.join(d, "g.fa"); r = os.path.join(d, "r.csfasta")
            write_fa_codes(g, [bytes(x) for x in z["contig_names"]], contigs)
            with open(r, "wb") as f:
                for i in range(n):
                    for nm, row in ((bytes(z["names1"][i]), m1[i]), (bytes(z["names2"][i]), m2[i])):
                        f.write(b">" + nm + b"\n" + b"ACGT"[row[0]:row[0] + 1] + bytes(b"0123"[c] if c < 4 else ord(".") for c in row[1:]) + b"\n")
            ins = tuple(int(x) for x in z["ins"])
            p = subprocess.run([os.path.join(ROOT, "oracle", "_ref", "gmapper-cs"), "-N", "4", "-p", str(z["mode"]), "-I", "%d,%d" % ins, "--sam-unaligned", *extra, r, g], capture_output=True, check=True)
            body = b"".join(l + b"\n" for l in p.stdout.split(b"\n") if l and not l.startswith(b"@PG") and not l.startswith(b"@RG"))
        with gzip.open(os.path.join(OUT, "%s@%s.sam.gz" % (base, tag)), "wb", compresslevel=9) as f:
            f.write(body)
        print("%s@%s: %d SAM records" % (base, tag, sum(1 for l in body.split(b"\n") if l and not l.startswith(b"@"))))


def option_cases(only=None):
    """non-default options on inputs that are already committed: only the reference's SAM body is stored (<base>@<tag>.sam.gz)"""
    for tag, (base, extra) in OPTION_CASES.items():
        if only and tag not in only: continue
        z = np.load(os.path.join(OUT, base + ".npz"))
        contigs = [z["contig%d" % i] for i in range(sum(1 for f in z.files if f.startswith("contig") and f[6:].isdigit()))]
        with tempfile.TemporaryDirectory() as d:
            g = os.path.join(d, "g.fa"); r = os.path.join(d, "r.fa")
            if "mates1" in z.files:
                cn = [bytes(x) for x in z["contig_names"]]
                write_fa_codes(g, cn, contigs)
                names = [bytes(n) for pair in zip(z["names1"], z["names2"]) for n in pair]
                seqs = [q for pair in zip(list(z["mates1"]), list(z["mates2"])) for q in pair]
                write_fa_codes(r, names, seqs)
                extra = ["-p", str(z["mode"]), "-I", "%d,%d" % tuple(int(x) for x in z["ins"]), *extra]
            else:
                write_fa_codes(g, [b"contig%d" % (i + 1) for i in range(len(contigs))], contigs)
                write_fa_codes(r, [b"r%d" % i for i in range(len(z["reads"]))], list(z["reads"]))
            p = subprocess.run([REF, "-N", "4", *extra, r, g], capture_output=True, check=True)
            body = b"".join(l + b"\n" for l in p.stdout.split(b"\n") if l and not l.startswith(b"@PG") and not l.startswith(b"@RG"))     # (--read-group adds an @RG header line)
        with gzip.open(os.path.join(OUT, "%s@%s.sam.gz" % (base, tag)), "wb", compresslevel=9) as f:
            f.write(body)
        print("%s@%s: %d SAM records" % (base, tag, sum(1 for l in body.split(b"\n") if l and not l.startswith(b"@"))))


def index_cases():
    """the reference's own index files (-S) for a tiny genome + the SAM it produces from them (-L); also with hashed seeds (-H)"""
    _index_case("idxfix", "11110111,1101011011", [])
    _index_case("idxfix_h", "1111011101111011,11011101101110111011", ["-H"])


def _index_case(dirname, seeds, extra):
    import shutil
    rng = np.random.default_rng(5)
    c1 = rng.integers(0, 4, 12000, dtype=np.uint8); c1[3000:3040] = 15; c1[7000:7003] = [5, 6, 14]
    c2 = rng.integers(0, 4, 7003, dtype=np.uint8)
    contigs = [c1, c2]; names = [b"chrA", b"chrB"]
    reads, _ = synth.make_reads(contigs, 400, 50, 91)
    d = os.path.join(OUT, dirname); shutil.rmtree(d, ignore_errors=True); os.makedirs(d)
    with tempfile.TemporaryDirectory() as t:
        g = os.path.join(t, "g.fa"); r = os.path.join(t, "r.fa")
        write_fa_codes(g, names, contigs)
        write_fa_codes(r, [b"r%d" % i for i in range(len(reads))], list(reads))
        subprocess.run([REF, *extra, "-s", seeds, "-S", os.path.join(d, "idx"), g], capture_output=True, check=True)
        p = subprocess.run([REF, "-N", "2", "-L", os.path.join(d, "idx"), r], capture_output=True, check=True)
        body = b"".join(l + b"\n" for l in p.stdout.split(b"\n") if l and not l.startswith(b"@PG"))
    np.savez_compressed(os.path.join(d, "inputs.npz"), contig0=c1, contig1=c2, reads=reads, seeds=np.array(seeds), contig_names=np.array(names))
    with gzip.open(os.path.join(d, "from_index.sam.gz"), "wb", compresslevel=9) as f:
        f.write(body)
    print(dirname + ": %d SAM records from the reference's -L run; files:" % sum(1 for l in body.split(b"\n") if l and not l.startswith(b"@")), sorted(os.listdir(d)))


def local_kat_cases():
    kat = subprocess.run([os.path.join(ROOT, "oracle", "_ref", "ref_kat"), "600", "local"], capture_output=True, check=True).stdout
    with gzip.open(os.path.join(OUT, "sw_kat_local.txt.gz"), "wb", compresslevel=9) as f:
        f.write(kat)
    print("sw_kat_local:", kat.count(b"\nL ") + 1, "sw_full_ls local-mode answers")


def cs_kat_cases():
    kat = subprocess.run([os.path.join(ROOT, "oracle", "_ref", "ref_kat_cs"), "700"], capture_output=True, check=True).stdout
    with gzip.open(os.path.join(OUT, "sw_kat_cs.txt.gz"), "wb", compresslevel=9) as f:
        f.write(kat)
    print("sw_kat_cs:", kat.count(b"\nC ") + 1, "colour-space vector,", kat.count(b"\nS "), "sw_full_cs")
    katl = subprocess.run([os.path.join(ROOT, "oracle", "_ref", "ref_kat_cs"), "400", "local"], capture_output=True, check=True).stdout
    with gzip.open(os.path.join(OUT, "sw_kat_cs_local.txt.gz"), "wb", compresslevel=9) as f:
        f.write(katl)
    print("sw_kat_cs_local:", katl.count(b"\nL ") + 1, "sw_full_cs in local mode")
    katx = subprocess.run([os.path.join(ROOT, "oracle", "_ref", "ref_kat_cs"), "500", "xover"], capture_output=True, check=True).stdout
    with gzip.open(os.path.join(OUT, "sw_kat_cs_xover.txt.gz"), "wb", compresslevel=9) as f:
        f.write(katx)
    print("sw_kat_cs_xover:", katx.count(b"\nX ") + 1, "sw_full_cs with per-position crossover scores,", katx.count(b"\nY "), "of them in local mode")
    katr = subprocess.run([os.path.join(ROOT, "oracle", "_ref", "ref_kat_cs"), "300", "rna"], capture_output=True, check=True).stdout
    with gzip.open(os.path.join(OUT, "sw_kat_cs_rna.txt.gz"), "wb", compresslevel=9) as f:
        f.write(katr)
    print("sw_kat_cs_rna:", katr.count(b"\nC ") + 1, "colour-space vector,", katr.count(b"\nS "), "+", katr.count(b"\nL "), "sw_full_cs (global + local), all on an RNA genome with is_rna = true")


def post_kat_cases():
    """the reference's own post_sw() on its own sw_full_cs results of sw_kat_cs.txt.gz (oracle/ref_kat_post.cpp), and its own sw_gapless() (oracle/ref_kat_gapless.cpp)"""
    src = gzip.open(os.path.join(OUT, "sw_kat_cs.txt.gz"), "rb").read()
    kat = subprocess.run([os.path.join(ROOT, "oracle", "_ref", "ref_kat_post")], input=src, capture_output=True, check=True).stdout
    with gzip.open(os.path.join(OUT, "sw_kat_post.txt.gz"), "wb", compresslevel=9) as f:
        f.write(kat)
    print("sw_kat_post:", kat.count(b"\nP "), "post_sw answers")
    kat = subprocess.run([os.path.join(ROOT, "oracle", "_ref", "ref_kat_gapless"), "1200"], capture_output=True, check=True).stdout
    with gzip.open(os.path.join(OUT, "sw_kat_gapless.txt.gz"), "wb", compresslevel=9) as f:
        f.write(kat)
    print("sw_kat_gapless:", kat.count(b"\nG ") + 1, "sw_gapless answers")
    katr = subprocess.run([os.path.join(ROOT, "oracle", "_ref", "ref_kat_gapless"), "600", "rna"], capture_output=True, check=True).stdout
    with gzip.open(os.path.join(OUT, "sw_kat_gapless_rna.txt.gz"), "wb", compresslevel=9) as f:
        f.write(katr)
    print("sw_kat_gapless_rna:", katr.count(b"\nG ") + 1, "colour-space sw_gapless answers on RNA genomes with is_rna = true")


def run_cs_case(name, contigs, reads, extra=()):
    """colour-space reads (codes[n, 1 + colours]) through the reference's gmapper-cs"""
    with tempfile.TemporaryDirectory() as d:
        g = os.path.join(d, "g.fa"); r = os.path.join(d, "r.csfasta")
        write_fa_codes(g, [b"contig%d" % (i + 1) for i in range(len(contigs))], contigs)
        synth.write_csfasta_reads(r, reads)
        p = subprocess.run([os.path.join(ROOT, "oracle", "_ref", "gmapper-cs"), "-N", "4", *extra, r, g], capture_output=True, check=True)
        body = b"".join(l + b"\n" for l in p.stdout.split(b"\n") if l and not l.startswith(b"@PG"))
    np.savez_compressed(os.path.join(OUT, name + ".npz"),
                        **{"contig%d" % i: c for i, c in enumerate(contigs)}, reads=reads)
    with gzip.open(os.path.join(OUT, name + ".sam.gz"), "wb", compresslevel=9) as f:
        f.write(body)
    n_map = sum(1 for l in body.split(b"\n") if l and not l.startswith(b"@"))
    print(f"{name}: {len(reads)} colour-space reads -> {n_map} SAM records")


def cs_cases():
    contigs = synth.make_genome(synth.contig_lengths("cfg2", 0.02), 4)
    reads, _ = synth.make_cs_reads(contigs, 3000, 50, 4)
    run_cs_case("cfg4s_50col_2Mbp", contigs, reads)
    sg = stress_genome()
    reads, _ = synth.make_cs_reads([c for c in sg if len(c) > 200], 2000, 60, 5, p_col=0.03, p_dot=0.004)
    run_cs_case("stress_cs_60col_unal", sg, reads, extra=("--sam-unaligned",))
    # the reference's own colour-space index files (-S) for a tiny genome + the SAM gmapper-cs produces from them (-L)
    import shutil
    rng = np.random.default_rng(6)
    c1 = rng.integers(0, 4, 12000, dtype=np.uint8); c1[3000:3040] = 15; c1[7000:7003] = [5, 6, 14]
    c2 = rng.integers(0, 4, 7003, dtype=np.uint8)
    contigs = [c1, c2]; names = [b"chrA", b"chrB"]
    reads, _ = synth.make_cs_reads(contigs, 400, 40, 92)
    seeds = "11110111,1101011011"
    refcs = os.path.join(ROOT, "oracle", "_ref", "gmapper-cs")
    d = os.path.join(OUT, "idxfix_cs"); shutil.rmtree(d, ignore_errors=True); os.makedirs(d)
    with tempfile.TemporaryDirectory() as t:
        g = os.path.join(t, "g.fa"); r = os.path.join(t, "r.csfasta")
        write_fa_codes(g, names, contigs)
        synth.write_csfasta_reads(r, reads)
        subprocess.run([refcs, "-s", seeds, "-S", os.path.join(d, "idx"), g], capture_output=True, check=True)
        p = subprocess.run([refcs, "-N", "2", "-L", os.path.join(d, "idx"), r], capture_output=True, check=True)
        body = b"".join(l + b"\n" for l in p.stdout.split(b"\n") if l and not l.startswith(b"@PG"))
    np.savez_compressed(os.path.join(d, "inputs.npz"), contig0=c1, contig1=c2, reads=reads, seeds=np.array(seeds), contig_names=np.array(names))
    with gzip.open(os.path.join(d, "from_index.sam.gz"), "wb", compresslevel=9) as f:
        f.write(body)
    print("idxfix_cs: %d SAM records from gmapper-cs -L; files:" % sum(1 for l in body.split(b"\n") if l and not l.startswith(b"@")), sorted(os.listdir(d)))


def cs_pair_case():
    """colour-space pairs through the reference's gmapper-cs -p <mode> (mates adjacent in one csfasta file): opp-in with unmappable mates and skipped
    cycles; the three modes that reverse a mate (read_reverse swaps the strands of a colour read) on 300 pairs each"""
    contigs = synth.make_genome([600_000, 400_000], 55)
    cn = [b"contig1", b"contig2"]; ins = (100, 600)
    rcl = lambda x: synth.COMPLEMENT[x[:, ::-1]]
    for mode, npairs, seed in (("opp-in", 800, 15), ("opp-out", 300, 21), ("col-fw", 300, 22), ("col-bw", 300, 23)):
        reads, _ = synth.make_pairs(contigs, npairs, 50, seed, ins_mean=250, ins_sd=30)
        a, b = reads[0::2].copy(), reads[1::2].copy()               # make_pairs yields opp-in mates: the other orientations are derived from them
        if mode == "opp-out": a, b = rcl(a), rcl(b)
        elif mode == "col-fw": b = rcl(b)
        elif mode == "col-bw": a = rcl(a)
        m1 = synth.cs_from_letters(a, 1); m2 = synth.cs_from_letters(b, 2)
        rng = np.random.default_rng(17)                               # some mates that map nowhere: half-paired records, unaligned pairs
        for i in range(len(m1)):
            if i % 20 == 7: m2[i, 1:] = rng.integers(0, 4, m2.shape[1] - 1)
            if i % 50 == 3: m1[i, 1:] = rng.integers(0, 4, m1.shape[1] - 1); m2[i, 1:] = rng.integers(0, 4, m2.shape[1] - 1)
            if i % 33 == 5: m1[i, 10] = 15                              # a skipped cycle ('.')
        n = len(m1); names1 = [b"p%d/1" % i for i in range(n)]; names2 = [b"p%d/2" % i for i in range(n)]
        name = "cs_pairs_50col_" + mode
        with tempfile.TemporaryDirectory() as d:
            g = os.path.join(d, "g.fa"); r = os.path.join(d, "r.csfasta")
            write_fa_codes(g, cn, contigs)
            with open(r, "wb") as f:
                for i in range(n):
                    for nm, row in ((names1[i], m1[i]), (names2[i], m2[i])):
                        f.write(b">" + nm + b"\n" + b"ACGT"[row[0]:row[0] + 1] + bytes(b"0123"[c] if c < 4 else ord(".") for c in row[1:]) + b"\n")
            p = subprocess.run([os.path.join(ROOT, "oracle", "_ref", "gmapper-cs"), "-N", "4", "-p", mode, "-I", "%d,%d" % ins, "--sam-unaligned", r, g], capture_output=True, check=True)
            body = b"".join(l + b"\n" for l in p.stdout.split(b"\n") if l and not l.startswith(b"@PG"))
        np.savez_compressed(os.path.join(OUT, name + ".npz"), **{"contig%d" % i: c for i, c in enumerate(contigs)}, mates1=m1, mates2=m2,
                            names1=np.array(names1), names2=np.array(names2), contig_names=np.array(cn), mode=np.array(mode), ins=np.array(ins))
        with gzip.open(os.path.join(OUT, name + ".sam.gz"), "wb", compresslevel=9) as f:
            f.write(body)
        print("%s: %d colour-space pairs -> %d SAM records" % (name, n, sum(1 for l in body.split(b"\n") if l and not l.startswith(b"@"))))


def fastq_cases():
    """FASTQ input (-Q autodetected): the QUAL strings in the SAM, once PHRED+33 (--qv-offset 33), once the default PHRED+64"""
    z = np.load(os.path.join(OUT, "stress_100bp_unal.npz"))
    contigs = [z["contig%d" % i] for i in range(sum(1 for f in z.files if f.startswith("contig") and f[6:].isdigit()))]
    reads = z["reads"][:600]
    rng = np.random.default_rng(17)
    T = np.frombuffer(b"ACGTUMRWSYKVHDBN", dtype=np.uint8)
    for tag, delta, extra in (("fq33", 33, ["--qv-offset", "33"]), ("fq64", 64, [])):
        q = (rng.integers(2, 41, size=reads.shape) + delta).astype(np.uint8)
        with tempfile.TemporaryDirectory() as d:
            g = os.path.join(d, "g.fa"); r = os.path.join(d, "r.fq")
            write_fa_codes(g, [b"contig%d" % (i + 1) for i in range(len(contigs))], contigs)
            with open(r, "wb") as f:
                for i in range(len(reads)):
                    f.write(b"@r%d\n" % i + T[reads[i]].tobytes() + b"\n+\n" + q[i].tobytes() + b"\n")
            p = subprocess.run([REF, "-N", "4", "--sam-unaligned", *extra, r, g], capture_output=True, check=True)
            body = b"".join(l + b"\n" for l in p.stdout.split(b"\n") if l and not l.startswith(b"@PG"))
        np.savez_compressed(os.path.join(OUT, "stress_100bp_%s.npz" % tag), quals=q, n_reads=np.array(len(reads)), qual_delta=np.array(delta))
        with gzip.open(os.path.join(OUT, "stress_100bp_%s.sam.gz" % tag), "wb", compresslevel=9) as f:
            f.write(body)
        print("stress_100bp_%s: %d SAM records" % (tag, sum(1 for l in body.split(b"\n") if l and not l.startswith(b"@"))))


def cs_fastq_cases():
    """colour-space FASTQ (csfastq, PHRED+33): per-position crossover scores, post_sw with read QVs, QUAL from post_sw, CQ:Z"""
    z = np.load(os.path.join(OUT, "cfg4s_50col_2Mbp.npz"))
    contigs = [z["contig%d" % i] for i in range(sum(1 for f in z.files if f.startswith("contig") and f[6:].isdigit()))]
    reads = z["reads"][:1500]
    rng = np.random.default_rng(19)
    q = (rng.integers(2, 36, size=(reads.shape[0], reads.shape[1] - 1)) + 33).astype(np.uint8)
    tab = np.full(16, ord("."), dtype=np.uint8); tab[:4] = np.frombuffer(b"0123", dtype=np.uint8)
    with tempfile.TemporaryDirectory() as d:
        g = os.path.join(d, "g.fa"); r = os.path.join(d, "r.csfastq")
        write_fa_codes(g, [b"contig%d" % (i + 1) for i in range(len(contigs))], contigs)
        with open(r, "wb") as f:
            for i in range(len(reads)):
                f.write(b"@r%d\n" % i + b"ACGT"[reads[i, 0]:reads[i, 0] + 1] + tab[reads[i, 1:]].tobytes() + b"\n+\n" + q[i].tobytes() + b"\n")
        p = subprocess.run([os.path.join(ROOT, "oracle", "_ref", "gmapper-cs"), "-N", "4", "--sam-unaligned", r, g], capture_output=True)
        if p.returncode != 0:
            print(p.stderr.decode()[-2000:]); raise SystemExit(1)
        body = b"".join(l + b"\n" for l in p.stdout.split(b"\n") if l and not l.startswith(b"@PG"))
        pl = subprocess.run([os.path.join(ROOT, "oracle", "_ref", "gmapper-cs"), "-N", "4", "--sam-unaligned", "--local", r, g], capture_output=True, check=True)      # the same reads with --local
        body_l = b"".join(l + b"\n" for l in pl.stdout.split(b"\n") if l and not l.startswith(b"@PG"))
    np.savez_compressed(os.path.join(OUT, "cfg4s_50col_fq.npz"), quals=q, n_reads=np.array(len(reads)), qual_delta=np.array(33))
    with gzip.open(os.path.join(OUT, "cfg4s_50col_fq.sam.gz"), "wb", compresslevel=9) as f:
        f.write(body)
    with gzip.open(os.path.join(OUT, "cfg4s_50col_fq@cs_fq_local.sam.gz"), "wb", compresslevel=9) as f:
        f.write(body_l)
    print("cfg4s_50col_fq: %d SAM records" % sum(1 for l in body.split(b"\n") if l and not l.startswith(b"@")))


def paired_fastq_case():
    """paired FASTQ (mates adjacent in one file, PHRED+33): QUAL strings in the paired and half-paired records"""
    z = np.load(os.path.join(OUT, "stress_pairs_2x100.npz"))
    contigs = [z["contig%d" % i] for i in range(sum(1 for f in z.files if f.startswith("contig") and f[6:].isdigit()))]
    cn = [bytes(x) for x in z["contig_names"]]
    m1, m2 = z["mates1"][:400], z["mates2"][:400]
    n1 = [bytes(x) for x in z["names1"]][:400]; n2 = [bytes(x) for x in z["names2"]][:400]
    rng = np.random.default_rng(23)
    q1 = (rng.integers(2, 41, size=m1.shape) + 33).astype(np.uint8); q2 = (rng.integers(2, 41, size=m2.shape) + 33).astype(np.uint8)
    T = np.frombuffer(b"ACGTUMRWSYKVHDBN", dtype=np.uint8)
    with tempfile.TemporaryDirectory() as d:
        g = os.path.join(d, "g.fa"); r = os.path.join(d, "r.fq")
        write_fa_codes(g, cn, contigs)
        with open(r, "wb") as f:
            for i in range(len(m1)):
                f.write(b"@" + n1[i] + b"\n" + T[m1[i]].tobytes() + b"\n+\n" + q1[i].tobytes() + b"\n")
                f.write(b"@" + n2[i] + b"\n" + T[m2[i]].tobytes() + b"\n+\n" + q2[i].tobytes() + b"\n")
        p = subprocess.run([REF, "-N", "4", "--qv-offset", "33", "--sam-unaligned", "-p", str(z["mode"]), "-I", "%d,%d" % tuple(int(x) for x in z["ins"]), r, g],
                           capture_output=True, check=True)
        body = b"".join(l + b"\n" for l in p.stdout.split(b"\n") if l and not l.startswith(b"@PG"))
    np.savez_compressed(os.path.join(OUT, "stress_pairs_fq33.npz"), quals1=q1, quals2=q2, n_pairs=np.array(len(m1)), qual_delta=np.array(33))
    with gzip.open(os.path.join(OUT, "stress_pairs_fq33.sam.gz"), "wb", compresslevel=9) as f:
        f.write(body)
    print("stress_pairs_fq33: %d SAM records" % sum(1 for l in body.split(b"\n") if l and not l.startswith(b"@")))


def main():
    os.makedirs(OUT, exist_ok=True)
    if "--fastq-only" in sys.argv:
        fastq_cases(); cs_fastq_cases(); paired_fastq_case(); return
    if "--cs-only" in sys.argv:
        cs_cases(); return
    if "--cs-kat-only" in sys.argv:
        cs_kat_cases(); return
    if "--local-kat-only" in sys.argv:
        local_kat_cases(); return
    if "--post-kat-only" in sys.argv:
        post_kat_cases(); return
    if "--cs-kat-only" in sys.argv:
        cs_kat_cases(); return
    if "--cs-pair-option-tags" in sys.argv:
        cs_pair_option_cases(only=sys.argv[sys.argv.index("--cs-pair-option-tags") + 1].split(",")); return
    if "--cs-pair-options-only" in sys.argv:
        cs_pair_option_cases(); return
    if "--cs-option-tags" in sys.argv:
        cs_option_cases(only=sys.argv[sys.argv.index("--cs-option-tags") + 1].split(",")); return
    if "--cs-options-only" in sys.argv:
        cs_option_cases(); return
    if "--index-only" in sys.argv:
        index_cases(); return
    if "--options-only" in sys.argv:
        option_cases(); return
    if "--option-tags" in sys.argv:                                   # --option-tags pairs_n3,cfg5_n3: just these
        option_cases(only=sys.argv[sys.argv.index("--option-tags") + 1].split(",")); return
    if "--chimeric-only" in sys.argv:
        chimeric_pairs_case(); option_cases(only=["chim_single_best_all", "chim_single_best_all_noimp", "chim_single_best"]); return
    if "--mirna-only" in sys.argv:
        mirna_case(); return
    if "--n1-only" in sys.argv:
        n1_noisy_case(); return
    if "--paired-only" in sys.argv:
        paired_cases(); return
    contigs, reads, _ = synth.make_config("cfg1")
    run_case("cfg1_36bp_1Mbp", contigs, reads)
    contigs, reads, _ = synth.make_config("cfg2", scale=0.02, n_reads=5000)
    run_case("cfg2s_100bp_2Mbp", contigs, reads)
    sg = stress_genome()
    run_case("stress_60bp", sg, stress_reads(sg, 3000, 60))
    run_case("stress_100bp_unal", sg, stress_reads(sg, 1500, 100, seed=90), extra=("--sam-unaligned",))
    kat = subprocess.run([os.path.join(ROOT, "oracle", "_ref", "ref_kat"), "1500"], capture_output=True, check=True).stdout
    with gzip.open(os.path.join(OUT, "sw_kat.txt.gz"), "wb", compresslevel=9) as f:
        f.write(kat)
    print("sw_kat:", kat.count(b"\nV ") + 1, "vector,", kat.count(b"\nF "), "full")
    paired_cases()
    chimeric_pairs_case()
    option_cases()
    n1_noisy_case()
    mirna_case()
    cs_option_cases()
    index_cases()
    cs_kat_cases()
    post_kat_cases()
    local_kat_cases()
    cs_cases()
    fastq_cases()
    cs_fastq_cases()
    paired_fastq_case()


def text_cases():
    """Text input: reads as the file's characters -- lower case, X / x / . for N, U, ambiguity codes; csfasta with '.', '4', 'N' for a skipped
    cycle and a lower-case primer -- through the reference with --sam-unaligned: what it prints verbatim from the file's text (SEQ of unaligned reads,
    CS:Z) and what it normalises.  Fixtures: the read lines themselves + the reference's SAM.  (With --local the reference exits on a reverse-strand
    clip that holds X or U -- "error in getting reverse complement", gmapper/output.c:213-216 -- so there is no golden for that.)"""
    rng = np.random.Generator(np.random.PCG64(2024))
    def run(exe, extra, reads_path, genome_path):
        p = subprocess.run([os.path.join(ROOT, "oracle", "_ref", exe), "-N", "2", *extra, reads_path, genome_path], capture_output=True, check=True)
        return b"".join(l + b"\n" for l in p.stdout.split(b"\n") if l and not l.startswith(b"@PG"))
    z = np.load(os.path.join(OUT, "stress_100bp_unal.npz")); contigs = [z["contig%d" % i] for i in range(len(z.files) - 1)]; reads = z["reads"][:400]
    LET = b"ACGTUMRWSYKVHDBN"; lines = []
    for r in reads:
        t = bytearray(LET[c] for c in r)
        for i, c in enumerate(r):
            if c == 15: t[i] = b"NnXx."[rng.integers(0, 5)]
            elif rng.random() < 0.15: t[i] = t[i] + 32
        lines.append(bytes(t))
    for k in range(0, 400, 9):                        # reads that cannot map: SEQ shows their text (X . U kept, ambiguity codes -> N)
        t = bytearray(lines[k])
        for q in rng.integers(0, 100, 25): t[q] = b"XUx.uRYK"[rng.integers(0, 8)]
        lines[k] = bytes(t)
    with tempfile.TemporaryDirectory() as d:
        g = os.path.join(d, "g.fa"); r = os.path.join(d, "r.fa")
        write_fa_codes(g, [b"contig%d" % (i + 1) for i in range(len(contigs))], contigs)
        with open(r, "wb") as f:
            for i, t in enumerate(lines): f.write(b">r%d\n" % i + t + b"\n")
        sam = run("gmapper-ls", ["--sam-unaligned"], r, g)
    with gzip.open(os.path.join(OUT, "text_ls_reads.txt.gz"), "wb", compresslevel=9) as f: f.write(b"\n".join(lines) + b"\n")
    with gzip.open(os.path.join(OUT, "text_ls_unal.sam.gz"), "wb", compresslevel=9) as f: f.write(sam)
    z = np.load(os.path.join(OUT, "stress_cs_60col_unal.npz")); contigs = [z["contig%d" % i] for i in range(len(z.files) - 1)]; reads = z["reads"][:300]
    lines = []
    for r in reads:
        t = bytearray(b"ACGT"[r[0]:r[0] + 1])
        for c in r[1:]: t += (b"0123"[c:c + 1] if c < 4 else b".4Nn"[rng.integers(0, 4):][:1])
        if rng.random() < 0.2: t[0] = t[0] + 32
        lines.append(bytes(t))
    with tempfile.TemporaryDirectory() as d:
        g = os.path.join(d, "g.fa"); r = os.path.join(d, "r.csfasta")
        write_fa_codes(g, [b"contig%d" % (i + 1) for i in range(len(contigs))], contigs)
        with open(r, "wb") as f:
            for i, t in enumerate(lines): f.write(b">r%d\n" % i + t + b"\n")
        sam = run("gmapper-cs", ["--sam-unaligned"], r, g)
    with gzip.open(os.path.join(OUT, "text_cs_reads.txt.gz"), "wb", compresslevel=9) as f: f.write(b"\n".join(lines) + b"\n")
    with gzip.open(os.path.join(OUT, "text_cs_unal.sam.gz"), "wb", compresslevel=9) as f: f.write(sam)


def file_cases():
    """File input: reads files as users have them -- gzip, '#' comments, sequences folded over lines, descriptions after the name, reads of
    several lengths mixed, FASTQ with folded sequences, a read longer than --longest-read -- through the reference with --sam-unaligned.  Fixtures:
    the files themselves (data the reference read) + the reference's SAM."""
    rng = np.random.Generator(np.random.PCG64(4242))
    sg = stress_genome()
    def run(exe, extra, reads_path, genome_path):
        p = subprocess.run([os.path.join(ROOT, "oracle", "_ref", exe), "-N", "2", "--sam-unaligned", *extra, reads_path, genome_path], capture_output=True, check=True)
        return b"".join(l + b"\n" for l in p.stdout.split(b"\n") if l and not l.startswith(b"@PG"))
    LET = b"ACGTUMRWSYKVHDBN"
    sets = [stress_reads(sg, 150, L, seed=300 + L) for L in (36, 50, 75, 100, 130)]
    recs = [(L, r) for rs in sets for L, r in ((rs.shape[1], x) for x in rs)]
    order = rng.permutation(len(recs))
    with tempfile.TemporaryDirectory() as d:
        g = os.path.join(d, "g.fa"); write_fa_codes(g, [b"contig%d" % (i + 1) for i in range(len(sg))], sg)
        # (1) FASTA, gzip, folded lines, comments, descriptions
        fa = bytearray(b"# reads of five lengths\n; second comment line\n")
        for n, k in enumerate(order):
            L, r = recs[k]; t = bytes(LET[c] for c in r)
            fa += b">read_%d  len=%d some description\twith a tab\n" % (n, L)
            w = int(rng.integers(20, 80))
            for o in range(0, L, w): fa += t[o:o + w] + b"\n"
            if n % 37 == 0: fa += b"# a comment between reads\n"
        fa += b">too_long\n" + bytes(LET[c] for c in rng.integers(0, 4, 1100)) + b"\n>after_long\n" + bytes(LET[c] for c in recs[0][1]) + b"\n"
        with gzip.open(os.path.join(d, "r.fa.gz"), "wb") as f: f.write(bytes(fa))
        sam = run("gmapper-ls", [], os.path.join(d, "r.fa.gz"), g)
        with gzip.open(os.path.join(OUT, "file_ls_mixed.fa.gz"), "wb", compresslevel=9) as f: f.write(bytes(fa))
        with gzip.open(os.path.join(OUT, "file_ls_mixed.sam.gz"), "wb", compresslevel=9) as f: f.write(sam)
        print("file_ls_mixed:", sum(1 for l in sam.split(b"\n") if l and not l.startswith(b"@")), "records")
        # (2) FASTQ (PHRED+33), plain, sequence and qualities folded, three lengths
        fq = bytearray()
        for n, k in enumerate(order[:300]):
            L, r = recs[k]; t = bytes(LET[c] for c in r); q = bytes((rng.integers(2, 40, L) + 33).astype(np.uint8))
            fq += b"@fq%d extra\n" % n
            if n % 3 == 0: fq += t[:L // 2] + b"\n" + t[L // 2:] + b"\n+fq%d\n" % n + q[:L // 3] + b"\n" + q[L // 3:] + b"\n"
            else: fq += t + b"\n+\n" + q + b"\n"
        open(os.path.join(d, "r.fq"), "wb").write(bytes(fq))
        sam = run("gmapper-ls", ["--qv-offset", "33"], os.path.join(d, "r.fq"), g)
        with gzip.open(os.path.join(OUT, "file_ls_mixed.fq.gz"), "wb", compresslevel=9) as f: f.write(bytes(fq))
        with gzip.open(os.path.join(OUT, "file_ls_mixed_fq.sam.gz"), "wb", compresslevel=9) as f: f.write(sam)
        print("file_ls_mixed_fq:", sum(1 for l in sam.split(b"\n") if l and not l.startswith(b"@")), "records")
        # (3) csfasta, two lengths (reads of the committed colour-space cases, cut)
        z = np.load(os.path.join(OUT, "stress_cs_60col_unal.npz")); cc = [z["contig%d" % i] for i in range(len(z.files) - 1)]; cr = z["reads"][:400]
        gc = os.path.join(d, "gc.fa"); write_fa_codes(gc, [b"contig%d" % (i + 1) for i in range(len(cc))], cc)
        cf = bytearray(b"# csfasta\n")
        for n, r in enumerate(cr):
            L = 60 if n % 2 else 45
            cf += b">c%d_F3\n" % n + b"ACGT"[r[0]:r[0] + 1] + bytes(b"0123"[c] if c < 4 else ord(".") for c in r[1:1 + L]) + b"\n"
        with gzip.open(os.path.join(d, "r.csfasta.gz"), "wb") as f: f.write(bytes(cf))
        sam = run("gmapper-cs", [], os.path.join(d, "r.csfasta.gz"), gc)
        with gzip.open(os.path.join(OUT, "file_cs_mixed.csfasta.gz"), "wb", compresslevel=9) as f: f.write(bytes(cf))
        with gzip.open(os.path.join(OUT, "file_cs_mixed.sam.gz"), "wb", compresslevel=9) as f: f.write(sam)
        print("file_cs_mixed:", sum(1 for l in sam.split(b"\n") if l and not l.startswith(b"@")), "records")


def cs_paired_fastq_cases():
    """csfastq pairs (PHRED+33) through gmapper-cs -p <mode>: QV-dependent crossover scores and post_sw error rates for both mates, QUAL / CQ:Z in paired, half-paired
    and unaligned records; opp-in and a mode that reverses a mate (col-bw)"""
    for mode in ("opp-in", "col-bw"):
        z = np.load(os.path.join(OUT, "cs_pairs_50col_%s.npz" % mode))
        contigs = [z["contig%d" % i] for i in range(sum(1 for f in z.files if f.startswith("contig") and f[6:].isdigit()))]
        cn = [bytes(x) for x in z["contig_names"]]
        N = 300
        m1, m2 = z["mates1"][:N], z["mates2"][:N]; n1 = [bytes(x) for x in z["names1"]][:N]; n2 = [bytes(x) for x in z["names2"]][:N]
        rng = np.random.default_rng(31)
        q1 = (rng.integers(2, 36, size=(N, m1.shape[1] - 1)) + 33).astype(np.uint8); q2 = (rng.integers(2, 36, size=(N, m2.shape[1] - 1)) + 33).astype(np.uint8)
        tab = np.full(16, ord("."), dtype=np.uint8); tab[:4] = np.frombuffer(b"0123", dtype=np.uint8)
        with tempfile.TemporaryDirectory() as d:
            g = os.path.join(d, "g.fa"); r = os.path.join(d, "r.csfastq")
            write_fa_codes(g, cn, contigs)
            with open(r, "wb") as f:
                for i in range(N):
                    for nm, row, q in ((n1[i], m1[i], q1[i]), (n2[i], m2[i], q2[i])):
                        f.write(b"@" + nm + b"\n" + b"ACGT"[row[0]:row[0] + 1] + tab[row[1:]].tobytes() + b"\n+\n" + q.tobytes() + b"\n")
            p = subprocess.run([os.path.join(ROOT, "oracle", "_ref", "gmapper-cs"), "-N", "4", "--sam-unaligned", "-p", mode, "-I", "%d,%d" % tuple(int(x) for x in z["ins"]), r, g],
                               capture_output=True)
            if p.returncode != 0:
                print(p.stderr.decode()[-1500:]); raise SystemExit(1)
            body = b"".join(l + b"\n" for l in p.stdout.split(b"\n") if l and not l.startswith(b"@PG"))
            pl = subprocess.run([os.path.join(ROOT, "oracle", "_ref", "gmapper-cs"), "-N", "4", "--sam-unaligned", "--local", "-p", mode, "-I", "%d,%d" % tuple(int(x) for x in z["ins"]), r, g],
                                capture_output=True)
            if pl.returncode != 0:
                print(pl.stderr.decode()[-1500:]); raise SystemExit(1)
            with gzip.open(os.path.join(OUT, "cs_pairs_fq_%s@cs_pairs_fq_local.sam.gz" % mode), "wb", compresslevel=9) as f:      # the same pairs with --local
                f.write(b"".join(l + b"\n" for l in pl.stdout.split(b"\n") if l and not l.startswith(b"@PG")))
        np.savez_compressed(os.path.join(OUT, "cs_pairs_fq_%s.npz" % mode), quals1=q1, quals2=q2, n_pairs=np.array(N), qual_delta=np.array(33))
        with gzip.open(os.path.join(OUT, "cs_pairs_fq_%s.sam.gz" % mode), "wb", compresslevel=9) as f:
            f.write(body)
        print("cs_pairs_fq_%s: %d SAM records" % (mode, sum(1 for l in body.split(b"\n") if l and not l.startswith(b"@"))))


def pair_file_cases():
    """paired file input: (1) gmapper -1 a.fq.gz -2 b.fq.gz, PHRED+33, mates cut to a mix of lengths; (2) one FASTA file with the mates adjacent (folded lines,
    comments).  Fixtures: the files themselves + the reference's SAM."""
    z = np.load(os.path.join(OUT, "stress_pairs_2x100.npz"))
    contigs = [z["contig%d" % i] for i in range(sum(1 for f in z.files if f.startswith("contig") and f[6:].isdigit()))]
    cn = [bytes(x) for x in z["contig_names"]]
    N = 360; m1, m2 = z["mates1"][:N], z["mates2"][:N]
    rng = np.random.default_rng(77); T = np.frombuffer(b"ACGTUMRWSYKVHDBN", dtype=np.uint8)
    l1 = rng.choice([100, 80, 64], size=N); l2 = rng.choice([100, 60], size=N)
    ins = tuple(int(x) for x in z["ins"])
    with tempfile.TemporaryDirectory() as d:
        g = os.path.join(d, "g.fa"); write_fa_codes(g, cn, contigs)
        a = bytearray(); b = bytearray(); il = bytearray(b"# pairs, mates adjacent\n")
        for i in range(N):
            s1 = T[m1[i][:l1[i]]].tobytes(); s2 = T[m2[i][100 - l2[i]:]].tobytes()     # mate 2 keeps its 3' part, so that the pair still spans the insert
            q1 = bytes((rng.integers(2, 40, l1[i]) + 33).astype(np.uint8)); q2 = bytes((rng.integers(2, 40, l2[i]) + 33).astype(np.uint8))
            a += b"@q%d/1 first mate\n" % i + s1 + b"\n+\n" + q1 + b"\n"; b += b"@q%d/2\n" % i + s2 + b"\n+\n" + q2 + b"\n"
            il += b">q%d/1\n" % i + s1[:50] + b"\n" + s1[50:] + b"\n>q%d/2 desc\n" % i + s2 + b"\n"
        with gzip.open(os.path.join(d, "a.fq.gz"), "wb") as f: f.write(bytes(a))
        with gzip.open(os.path.join(d, "b.fq.gz"), "wb") as f: f.write(bytes(b))
        open(os.path.join(d, "il.fa"), "wb").write(bytes(il))
        run = lambda args: b"".join(l + b"\n" for l in subprocess.run([REF, "-N", "4", "--sam-unaligned", "-p", str(z["mode"]), "-I", "%d,%d" % ins, *args, g],
                                                                        capture_output=True, check=True).stdout.split(b"\n") if l and not l.startswith(b"@PG"))
        sam12 = run(["--qv-offset", "33", "-1", os.path.join(d, "a.fq.gz"), "-2", os.path.join(d, "b.fq.gz")])
        samil = run([os.path.join(d, "il.fa")])
    for nm, data in (("file_pairs_1.fq.gz", bytes(a)), ("file_pairs_2.fq.gz", bytes(b)), ("file_pairs_il.fa.gz", bytes(il)), ("file_pairs_12.sam.gz", sam12), ("file_pairs_il.sam.gz", samil)):
        with gzip.open(os.path.join(OUT, nm), "wb", compresslevel=9) as f: f.write(data)
    print("file_pairs_12:", sum(1 for l in sam12.split(b"\n") if l and not l.startswith(b"@")), "records; file_pairs_il:", sum(1 for l in samil.split(b"\n") if l and not l.startswith(b"@")))


PREPROCESS_CASES = {
    # tag: (input fixture, options of the reference, gm_params_t fields): the read loop's preprocessing (ref: gmapper.c:262-284,427-472,495-521)
    "pre_trim": ("pre_reads.fa.gz", ["--trim-front", "3", "--trim-end", "5"], {"trim_front": 3, "trim_end": 5}),
    "pre_q_default": ("pre_reads.fq.gz", ["--qv-offset", "64"], {}),                                    # --min-avg-qv is 10 unless told otherwise: low-quality reads get no record
    "pre_q_min20": ("pre_reads.fq.gz", ["--qv-offset", "64", "--min-avg-qv", "20"], {"min_avg_qv": 20}),
    "pre_q_none": ("pre_reads.fq.gz", ["--qv-offset", "64", "--min-avg-qv", "-1"], {"min_avg_qv": -1}),
    "pre_q_illumina": ("pre_reads.fq.gz", ["--qv-offset", "64", "--trim-illumina"], {"trim_illumina": 1}),
    "pre_q_ignore": ("pre_reads.fq.gz", ["--qv-offset", "64", "--ignore-qvs"], {"ignore_qvs": 1}),
    "pre_q_trim": ("pre_reads.fq.gz", ["--qv-offset", "64", "--trim-front", "2", "--trim-end", "3", "--trim-illumina"], {"trim_front": 2, "trim_end": 3, "trim_illumina": 1}),
}


def preprocess_cases():
    """The read loop's preprocessing through the reference: --trim-front / --trim-end, --trim-illumina, --min-avg-qv (default 10), --ignore-qvs on a FASTA and a FASTQ
    (PHRED+64) file of mixed lengths, and --trim-second on the pair files of pair_file_cases.  Fixtures: the files + the reference's SAM per option set."""
    rng = np.random.Generator(np.random.PCG64(5151))
    sg = stress_genome()
    LET = b"ACGTUMRWSYKVHDBN"
    sets = [stress_reads(sg, 120, L, seed=500 + L) for L in (50, 75, 100)]
    recs = [r for rs in sets for r in rs]
    order = rng.permutation(len(recs))
    fa = bytearray(); fq = bytearray()
    for n, k in enumerate(order):
        r = recs[k]; L = len(r); t = bytes(LET[c] for c in r)
        fa += b">t%d\n" % n + t + b"\n"
        kind = n % 6
        if kind == 0: q = rng.integers(2, 9, L)                                    # poor read: mean below 10
        elif kind == 1: q = rng.integers(8, 24, L)                                 # mean around 15: kept by default, dropped at 20
        else: q = rng.integers(15, 41, L)
        if kind in (2, 3): q[L - int(rng.integers(1, 12)):] = 2                    # an Illumina "B" tail (PHRED+64: 'B' = quality 2)
        if n % 41 == 0: q[:] = 2                                                   # all B: --trim-illumina leaves nothing
        fq += b"@t%d\n" % n + t + b"\n+\n" + bytes((q + 64).astype(np.uint8)) + b"\n"
    with gzip.open(os.path.join(OUT, "pre_reads.fa.gz"), "wb", compresslevel=9) as f: f.write(bytes(fa))
    with gzip.open(os.path.join(OUT, "pre_reads.fq.gz"), "wb", compresslevel=9) as f: f.write(bytes(fq))
    with tempfile.TemporaryDirectory() as d:
        g = os.path.join(d, "g.fa"); write_fa_codes(g, [b"contig%d" % (i + 1) for i in range(len(sg))], sg)
        open(os.path.join(d, "pre_reads.fa.gz"), "wb").write(gzip.compress(bytes(fa))); open(os.path.join(d, "pre_reads.fq.gz"), "wb").write(gzip.compress(bytes(fq)))
        for tag, (src, extra, _) in PREPROCESS_CASES.items():
            p = subprocess.run([os.path.join(ROOT, "oracle", "_ref", "gmapper-ls"), "-N", "2", "--sam-unaligned", *extra, os.path.join(d, src), g], capture_output=True)
            if p.returncode != 0: print(p.stderr.decode()[-1500:]); raise SystemExit(1)
            sam = b"".join(l + b"\n" for l in p.stdout.split(b"\n") if l and not l.startswith(b"@PG"))
            with gzip.open(os.path.join(OUT, tag + ".sam.gz"), "wb", compresslevel=9) as f: f.write(sam)
            print(tag + ":", sum(1 for l in sam.split(b"\n") if l and not l.startswith(b"@")), "records")
    # pairs: --trim-second (the second mate only; the reference trims a first mate after packing it, see gm_check_pair_trim)
    z = np.load(os.path.join(OUT, "stress_pairs_2x100.npz"))
    contigs = [z["contig%d" % i] for i in range(sum(1 for f in z.files if f.startswith("contig") and f[6:].isdigit()))]
    cn = [bytes(x) for x in z["contig_names"]]; ins = tuple(int(x) for x in z["ins"])
    with tempfile.TemporaryDirectory() as d:
        g = os.path.join(d, "g.fa"); write_fa_codes(g, cn, contigs)
        for nm in ("file_pairs_1.fq.gz", "file_pairs_2.fq.gz"): open(os.path.join(d, nm), "wb").write(open(os.path.join(OUT, nm), "rb").read())
        p = subprocess.run([REF, "-N", "4", "--sam-unaligned", "-p", str(z["mode"]), "-I", "%d,%d" % ins, "--qv-offset", "33", "--trim-second", "--trim-front", "2", "--trim-end", "4",
                            "-1", os.path.join(d, "file_pairs_1.fq.gz"), "-2", os.path.join(d, "file_pairs_2.fq.gz"), g], capture_output=True)
        if p.returncode != 0: print(p.stderr.decode()[-1500:]); raise SystemExit(1)
        sam = b"".join(l + b"\n" for l in p.stdout.split(b"\n") if l and not l.startswith(b"@PG"))
    with gzip.open(os.path.join(OUT, "pre_pairs_trim_second.sam.gz"), "wb", compresslevel=9) as f: f.write(sam)
    print("pre_pairs_trim_second:", sum(1 for l in sam.split(b"\n") if l and not l.startswith(b"@")), "records")


FORMAT_CASES = {
    # tag: (base golden, program, options): the reference's SHRiMP-format / pretty output, whole (its #FORMAT line included)
    "fmt_shrimp": ("stress_60bp", "gmapper-ls", ["--shrimp-format"]),
    "fmt_pretty_R": ("stress_60bp", "gmapper-ls", ["-P", "-R"]),
    "fmt_local_pretty": ("stress_100bp_unal", "gmapper-ls", ["--local", "-P"]),
    "fmt_cs_shrimp_R": ("stress_cs_60col_unal", "gmapper-cs", ["--shrimp-format", "-R"]),
    "fmt_cs_pretty": ("stress_cs_60col_unal", "gmapper-cs", ["-P"]),
    # pairs: one line per mate under the mate's own name, ">name" for the unmapped mate of a half-paired mapping
    "fmt_pairs_shrimp_R": ("stress_pairs_2x100", "gmapper-ls", ["--shrimp-format", "-R"]),
    "fmt_pairs_pretty": ("pairfix_opp-out", "gmapper-ls", ["-P"]),
    "fmt_pairs_colbw": ("pairfix_col-bw", "gmapper-ls", ["--shrimp-format"]),
    "fmt_cs_pairs_pretty_R": ("cs_pairs_50col_col-bw", "gmapper-cs", ["-P", "-R"]),
    "fmt_cs_pairs_shrimp": ("cs_pairs_50col_opp-in", "gmapper-cs", ["--shrimp-format"]),
}


def format_cases():
    for tag, (base, exe, extra) in FORMAT_CASES.items():
        z = np.load(os.path.join(OUT, base + ".npz"))
        contigs = [z["contig%d" % i] for i in range(sum(1 for f in z.files if f.startswith("contig") and f[6:].isdigit()))]
        with tempfile.TemporaryDirectory() as d:
            g = os.path.join(d, "g.fa"); r = os.path.join(d, "r.fa")
            write_fa_codes(g, [b"contig%d" % (i + 1) for i in range(len(contigs))], contigs)
            if "mates1" in z.files:
                write_fa_codes(g, [bytes(x) for x in z["contig_names"]], contigs)
                names = [bytes(n) for pair in zip(z["names1"], z["names2"]) for n in pair]
                seqs = [q for pair in zip(list(z["mates1"]), list(z["mates2"])) for q in pair]
                if exe == "gmapper-cs":
                    tab = np.full(16, ord("."), dtype=np.uint8); tab[:4] = np.frombuffer(b"0123", dtype=np.uint8)
                    with open(r, "wb") as f:
                        for nm, q in zip(names, seqs): f.write(b">" + nm + b"\n" + b"ACGT"[q[0]:q[0] + 1] + tab[q[1:]].tobytes() + b"\n")
                else: write_fa_codes(r, names, seqs)
                extra = ["-p", str(z["mode"]), "-I", "%d,%d" % tuple(int(x) for x in z["ins"]), *extra]
            elif exe == "gmapper-cs": synth.write_csfasta_reads(r, z["reads"])
            else: write_fa_codes(r, [b"r%d" % i for i in range(len(z["reads"]))], list(z["reads"]))
            p = subprocess.run([os.path.join(ROOT, "oracle", "_ref", exe), "-N", "4", *extra, r, g], capture_output=True, check=True)
        with gzip.open(os.path.join(OUT, "%s@%s.txt.gz" % (base, tag)), "wb", compresslevel=9) as f:
            f.write(p.stdout)
        print("%s@%s: %d mappings" % (base, tag, p.stdout.count(b"\n>")))


RNA_CASES = {
    # tag: (binary, genome fixture, read fixtures, options of the reference): RNA sequences -- uracil and no thymine (ref: fasta.c:528-542).  A contig's flag decides its
    # reverse complement (A -> U) and its colour translation (U read as T; genome.c:1107-1118); the LAST contig's flag is genome_is_rna, which the SW calls get
    # (genome.c:1063-1064; mapping.c:375-388,1318-1327); a letter-space read's own flag decides its reverse complement (gmapper.c:487).
    "rna_ls": ("gmapper-ls", "rna_genome.fa.gz", ["rna_reads_ls.fa.gz"], []),
    "rna_ls_pairs": ("gmapper-ls", "rna_genome.fa.gz", ["rna_pairs_1.fa.gz", "rna_pairs_2.fa.gz"], ["-p", "opp-in", "-I", "100,500"]),
    "rna_ls_last_dna": ("gmapper-ls", "rna_genome_last_dna.fa.gz", ["rna_reads_ls.fa.gz"], []),
    "rna_cs": ("gmapper-cs", "rna_genome.fa.gz", ["rna_reads_cs.fa.gz"], []),
    "rna_cs_fq": ("gmapper-cs", "rna_genome.fa.gz", ["rna_reads_cs.fq.gz"], ["--qv-offset", "33"]),
    "rna_cs_last_dna": ("gmapper-cs", "rna_genome_last_dna.fa.gz", ["rna_reads_cs.fa.gz"], []),
    "rna_cs_last_rna": ("gmapper-cs", "rna_genome_last_rna.fa.gz", ["rna_reads_cs.fa.gz"], []),
    "rna_cs_ungapped": ("gmapper-cs", "rna_genome.fa.gz", ["rna_reads_cs.fa.gz"], ["--local", "-U"]),
}


def rna_cases():
    """RNA genomes and RNA reads through the reference.  Fixtures: three genome files (two RNA contigs; the same + a DNA contig last; the DNA contig first), letter-space
    reads spelled in RNA, in DNA and in both, mate files, colour reads with and without QVs, and the reference's SAM for each combination of RNA_CASES."""
    rng = np.random.Generator(np.random.PCG64(9090))
    def contig(n, p_last):                                                          # codes 0..2 + `last` (3 = T, 4 = U), the last letter rare so that colour reads survive
        return rng.choice(np.array([0, 1, 2, 9], dtype=np.uint8), size=n, p=[0.32, 0.30, 0.30, 0.08])
    rna1 = contig(30000, 0); rna2 = contig(20000, 0); dna3 = contig(20000, 0)
    rna1[rna1 == 9] = 4; rna2[rna2 == 9] = 4; dna3[dna3 == 9] = 3
    rna2[5000:5040] = 15                                                            # a run of N
    LET = b"ACGTUMRWSYKVHDBN"
    def fa(names, seqs):
        out = bytearray()
        for nm, sq in zip(names, seqs):
            out += b">" + nm + b"\n"; t = bytes(LET[c] for c in sq)
            for k in range(0, len(t), 60): out += t[k:k + 60] + b"\n"
        return bytes(out)
    genomes = {"rna_genome.fa.gz": fa([b"rna1", b"rna2"], [rna1, rna2]),
               "rna_genome_last_dna.fa.gz": fa([b"rna1", b"rna2", b"dna3"], [rna1, rna2, dna3]),
               "rna_genome_last_rna.fa.gz": fa([b"dna3", b"rna1", b"rna2"], [dna3, rna1, rna2])}
    contigs = [rna1, rna2, dna3]
    def draw(L, nsub, indel):
        c = contigs[int(rng.integers(0, 3))]
        p = int(rng.integers(0, len(c) - L - 8)); s = c[p:p + L + 4].copy()
        if indel == 1: s = np.delete(s, int(rng.integers(10, L - 10)))
        elif indel == 2: s = np.insert(s, int(rng.integers(10, L - 10)), int(rng.integers(0, 3)))
        s = s[:L].copy()
        for _ in range(nsub): s[int(rng.integers(0, L))] = int(rng.integers(0, 3))
        return s
    def revcomp(s, rna):                                                            # a molecule's other strand, spelled like the first
        cm = np.array([4 if rna else 3, 2, 1, 0, 0, 10, 9, 7, 8, 6, 5, 14, 13, 12, 11, 15], dtype=np.uint8)
        return cm[s[::-1]]
    # letter space: 360 reads of 60 letters
    ls = bytearray()
    for n in range(360):
        s = draw(60, int(rng.integers(0, 4)), int(rng.integers(0, 6)) if n % 3 == 0 else 0)
        rna = bool((s == 4).any())
        if n % 2: s = revcomp(s, rna)
        kind = n % 12
        if kind == 5: s = np.where(s == 4, 3, s).astype(np.uint8)                  # an RNA molecule spelled with T
        elif kind == 7:                                                            # both U and T: not RNA to the reference
            s = s.copy(); idx = np.flatnonzero((s == 4) | (s == 3))
            if len(idx) >= 2: s[idx[0]] = 3; s[idx[1]] = 4
        elif kind == 9: s = s.copy(); s[int(rng.integers(0, 60))] = 15
        ls += b">r%d\n" % n + bytes(LET[c] for c in s) + b"\n"
    # mates: opp-in, inserts of 150-400, the molecule spelled in its contig's alphabet
    m1 = bytearray(); m2 = bytearray()
    for n in range(160):
        c = contigs[int(rng.integers(0, 2))]; ins = int(rng.integers(150, 400)); p = int(rng.integers(0, len(c) - ins))
        frag = c[p:p + ins].copy()
        for _ in range(int(rng.integers(0, 4))): frag[int(rng.integers(0, ins))] = int(rng.integers(0, 3))
        a = frag[:50]; b = revcomp(frag[-50:], True)
        if n % 2: a, b = b, a
        if n % 10 == 3: b = draw(50, 0, 0)                                          # a mate from somewhere else
        m1 += b">p%d/1\n" % n + bytes(LET[x] for x in a) + b"\n"; m2 += b">p%d/2\n" % n + bytes(LET[x] for x in b) + b"\n"
    # colour space: 400 reads of 40 colours (primer T), with QVs in the FASTQ twin
    cm4 = np.array([[0, 1, 2, 3], [1, 0, 3, 2], [2, 3, 0, 1], [3, 2, 1, 0]])
    cs = bytearray(); cq = bytearray()
    for n in range(400):
        s = draw(40, 0, int(rng.integers(0, 6)) if n % 4 == 0 else 0)
        if n % 2: s = revcomp(s, False)
        s = np.where(s == 4, 3, s); s = np.where(s > 3, 0, s)
        cols = [int(cm4[3][s[0]])] + [int(cm4[s[k]][s[k + 1]]) for k in range(39)]
        for _ in range(int(rng.integers(0, 3))): cols[int(rng.integers(0, 40))] = int(rng.integers(0, 4))
        t = b"T" + bytes(48 + c for c in cols)
        if n % 37 == 0: t = t[:20] + b"." + t[21:]
        q = rng.integers(8, 38, 40)
        cs += b">c%d\n" % n + t + b"\n"; cq += b"@c%d\n" % n + t + b"\n+\n" + bytes((q + 33).astype(np.uint8)) + b"\n"
    reads = {"rna_reads_ls.fa.gz": bytes(ls), "rna_pairs_1.fa.gz": bytes(m1), "rna_pairs_2.fa.gz": bytes(m2), "rna_reads_cs.fa.gz": bytes(cs), "rna_reads_cs.fq.gz": bytes(cq)}
    for nm, data in {**genomes, **reads}.items():
        with gzip.open(os.path.join(OUT, nm), "wb", compresslevel=9) as f: f.write(data)
    with tempfile.TemporaryDirectory() as d:
        for nm, data in genomes.items(): open(os.path.join(d, nm[:-3]), "wb").write(data)
        for nm, data in reads.items(): open(os.path.join(d, nm), "wb").write(gzip.compress(data))
        for tag, (binary, g, rds, extra) in RNA_CASES.items():
            files = [os.path.join(d, r) for r in rds]
            if len(files) == 2: files = ["-1", files[0], "-2", files[1]]
            p = subprocess.run([os.path.join(ROOT, "oracle", "_ref", binary), "-N", "2", "--sam-unaligned", *extra, *files, os.path.join(d, g[:-3])], capture_output=True)
            if p.returncode != 0: print(p.stderr.decode()[-1500:]); raise SystemExit(1)
            sam = b"".join(l + b"\n" for l in p.stdout.split(b"\n") if l and not l.startswith(b"@PG"))
            with gzip.open(os.path.join(OUT, tag + ".sam.gz"), "wb", compresslevel=9) as f: f.write(sam)
            recs = [l for l in sam.split(b"\n") if l and not l.startswith(b"@")]
            print(tag + ":", len(recs), "records,", sum(1 for l in recs if not int(l.split(b"\t")[1]) & 4), "mapped")


if __name__ == "__main__":
    if "--pair-file-only" in sys.argv:
        os.makedirs(OUT, exist_ok=True); pair_file_cases()
    elif "--cs-pairs-fq-only" in sys.argv:
        os.makedirs(OUT, exist_ok=True); cs_paired_fastq_cases()
    elif "--format-only" in sys.argv:
        os.makedirs(OUT, exist_ok=True); format_cases()
    elif "--file-only" in sys.argv:
        os.makedirs(OUT, exist_ok=True); file_cases()
    elif "--rna-only" in sys.argv:
        os.makedirs(OUT, exist_ok=True); rna_cases()
    elif "--preprocess-only" in sys.argv:
        os.makedirs(OUT, exist_ok=True); preprocess_cases()
    else:
        main()
