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


def main():
    os.makedirs(OUT, exist_ok=True)
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


if __name__ == "__main__":
    main()
