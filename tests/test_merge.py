"""gm_merge_sam (shrimp_amd/csrc/gm_merge.cpp) against the reference's mergesam (ref: mergesam/mergesam.c).

Host-only text processing inside the C-ABI library, so these run without a GPU: every fixture under tests/golden/merge is the output of the
reference's mergesam on SAM files the reference's gmapper wrote (tools/make_merge_golden.py).  When oracle/_ref/mergesam is present (build
container) the same comparison is repeated on fresh shards cut a different way."""
import gzip, json, os, subprocess
import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
M = os.path.join(ROOT, "tests", "golden", "merge")
CASES = json.load(open(os.path.join(M, "cases.json")))
FLAGS = {"--strata": ("strata", 1), "--single-best-mapping": ("single_best", 1), "--all-contigs": ("all_contigs", 1), "--sam-unaligned": ("sam_unaligned", 1),
         "--no-half-paired": ("half_paired", 0), "--no-mapping-qualities": ("no_mapping_qualities", 1), "--leave-mapq-untouched": ("leave_mapq", 1),
         "--no-improper-mappings": ("no_improper_mappings", 1)}
VALUES = {"-o": "max_outputs", "--max-alignments": "max_alignments", "--min-mapq": "min_mapq"}


def gz(name):
    with gzip.open(os.path.join(M, name), "rb") as f:
        return f.read()


def options_of(args):
    kw, out, it = {}, "sam", iter(args)
    for a in it:
        if a in ("--un", "--al"): out = a[2:]
        elif a in FLAGS: kw[FLAGS[a][0]] = FLAGS[a][1]
        else: kw[VALUES[a]] = int(next(it))
    return kw, out


@pytest.mark.parametrize("case", sorted(CASES))
def test_merge_matches_reference_mergesam(case):
    from shrimp_amd import gmapper as gm
    c = CASES[case]
    reads = gz(case + ".reads.gz"); sams = [gz("%s.in%d.sam.gz" % (case, k)) for k in range(len(c["inputs"]))]
    for st, d in sorted(c["sets"].items()):
        kw, out = options_of(d["args"])
        want = gz("%s@%s.out.gz" % (case, st))
        for threads in (1, 5):
            # (the fixture's own @PG line reads "CL:mergesam --sam ...": the program's path was cut away when it was made)
            got = gm.merge_sam(reads, sams, command_line=d["command_line"] if out == "sam" else None, output=out, threads=threads, **kw)
            assert got == want, (case, st, threads)


def test_merge_of_one_whole_run_keeps_every_record():
    """merging a single whole-genome SAM: every record comes back with its placement, CIGAR, sequence and tags (size-independent property)"""
    from shrimp_amd import gmapper as gm
    with gzip.open(os.path.join(ROOT, "tests", "golden", "stress_60bp.sam.gz"), "rb") as f:
        sam = f.read()
    n = np.load(os.path.join(ROOT, "tests", "golden", "stress_60bp.npz"))["reads"].shape[0]
    reads = b"".join(b">r%d\nA\n" % i for i in range(n))
    got = gm.merge_sam(reads, [sam], threads=2)
    body = lambda t: [l for l in t.split(b"\n") if l and not l.startswith(b"@")]
    a, b = body(got), body(sam)
    assert len(a) == len(b)
    # (within a read the records come back in the bounded heap's array order, mergesam_heap.c; a Z field may lose one unit on the way through
    # exp / log -- tnlog(inv_tnlog(x)) == x - 1 for some x -- as with the reference's program)
    key = lambda l: (l.split(b"\t")[0], l.split(b"\t")[1:4], l.split(b"\t")[5:12])
    a.sort(key=key); b.sort(key=key)
    for x, y in zip(a, b):
        fx, fy = x.split(b"\t"), y.split(b"\t")
        assert fx[:4] == fy[:4] and fx[5:12] == fy[5:12]
        assert fx[-1] == fy[-1]
        # (MAPQ is NOT preserved: it is recomputed from Z fields rounded to 1/1000 nat, so 1 - z0/z1 of a confident mapping collapses -- 72 becomes 250 --
        # in the reference's program as well; the byte-level behaviour is pinned by the fixtures above)


def test_merge_rejects_bad_input():
    from shrimp_amd import gmapper as gm
    with pytest.raises(gm.GmError):
        gm.merge_sam(b">r0\nA\n", [b"r0\t0\tc\t1\t250\t3M\t*\t0\t0\tAAA\t*\tAS:i:30\tNM:i:0\n"])            # no Z fields: MAPQ cannot be recomputed
    with pytest.raises(gm.GmError):
        gm.merge_sam(b">r0\nA\n", [b"r0\t0\tc\t1\n"])                                                      # not a SAM record
    with pytest.raises(gm.GmError):
        gm.merge_sam(b">r0\nA\n", [b""], single_best=1, no_mapping_qualities=1)                             # ref: mergesam.c:555-558
    assert gm.merge_sam(b">r0\nA\n", [b"r0\t0\tc\t1\t250\t3M\t*\t0\t0\tAAA\t*\tAS:i:30\tNM:i:0\n"], no_mapping_qualities=1, leave_mapq=1) == \
        b"r0\t0\tc\t1\t250\t3M\t*\t0\t0\tAAA\t*\tAS:i:30\tNM:i:0\n"
    assert gm.merge_sam(b"", [b""]) == b""


@pytest.mark.skipif(not os.path.exists(os.path.join(ROOT, "oracle", "_ref", "mergesam")), reason="reference mergesam not built (oracle/Makefile.ref)")
def test_merge_against_reference_program_on_other_shards(tmp_path):
    """three-way split of the reads of a committed case by record (the reference's program run here, on inputs no fixture holds)"""
    from shrimp_amd import gmapper as gm
    case = "pairs_db2"
    reads = gz(case + ".reads.gz"); sams = [gz("%s.in%d.sam.gz" % (case, k)) for k in range(2)]
    # shard every input in two by read: records of the first 201 pairs / the rest
    def cut(t):
        hdr = [l for l in t.split(b"\n") if l.startswith(b"@")]; rec = [l for l in t.split(b"\n") if l and not l.startswith(b"@")]
        k = next(i for i, l in enumerate(rec) if int(l.split(b"\t")[0][1:]) > 200)
        return b"\n".join(hdr + rec[:k]) + b"\n", b"\n".join(hdr + rec[k:]) + b"\n"
    parts = [p for t in sams for p in cut(t)]
    (tmp_path / "r.fa").write_bytes(reads)
    names = []
    for i, p in enumerate(parts):
        (tmp_path / ("s%d.sam" % i)).write_bytes(p); names.append("s%d.sam" % i)
    exe = os.path.join(ROOT, "oracle", "_ref", "mergesam")
    for args in ([], ["--single-best-mapping", "--all-contigs"], ["--strata", "-o", "4"], ["--sam-unaligned", "--no-half-paired"]):
        ref = subprocess.run([exe, "--sam", *args, "r.fa", *names], cwd=tmp_path, capture_output=True, check=True).stdout
        kw, _ = options_of(args)
        got = gm.merge_sam(reads, parts, command_line=" ".join([exe, "--sam", *args, "r.fa", *names]) + " ", threads=3, **kw)
        assert got == ref, args
