import pytest
"""Host-side pieces that need no GPU: bitfield packing, synthetic inputs, read sharding."""
import numpy as np
from shrimp_amd import synth, parallel


def unpack(words, n):
    return np.array([(int(words[i // 8]) >> (4 * (i % 8))) & 15 for i in range(n)], dtype=np.uint8)


def test_pack_nibbles_matches_reference_bitfield_layout():
    rng = np.random.default_rng(1)
    for n in (1, 7, 8, 9, 36, 100, 101):
        c = rng.integers(0, 16, size=n, dtype=np.uint8)
        assert (unpack(synth.pack_nibbles(c), n) == c).all()
    r = rng.integers(0, 16, size=(6, 37), dtype=np.uint8)
    w = synth.pack_reads(r)
    assert w.shape == (6, 5)
    for k in range(6):
        assert (unpack(w[k], 37) == r[k]).all()
        assert unpack(w[k], 40)[37:].sum() == 0          # unused nibbles are zero, as in fasta_sequence_to_bitfield


def test_synth_is_deterministic_and_shaped():
    a, ra, ta = synth.make_config("cfg1", n_reads=500)
    b, rb, tb = synth.make_config("cfg1", n_reads=500)
    assert len(a) == 1 and len(a[0]) == 1_000_000 and (a[0] == b[0]).all() and (ra == rb).all()
    assert ra.shape == (500, 36) and ra.max() <= 3
    lens = synth.contig_lengths("cfg3")
    assert len(lens) == 24 and max(lens) < 2**31 and 2.9e9 < sum(lens) < 2**32


def test_shard_bounds_cover_input_in_whole_chunks():
    for n, w in ((10_000, 2), (10_001, 8), (999, 4), (0, 2), (1_000_000, 8)):
        b = parallel.shard_bounds(n, w)
        assert b[0][0] == 0 and b[-1][1] == n
        assert all(b[i][1] == b[i + 1][0] for i in range(w - 1))
        assert all(lo % 1000 == 0 for lo, hi in b if lo < n)
    b = parallel.shard_bounds(10_000, 3, chunk=1001, paired=True)
    assert all(lo % 2 == 0 for lo, _ in b)                # mates stay together


def test_sequence_to_bitfield_follows_the_reference_tables():
    """gm_sequence_to_bitfield == fasta_sequence_to_bitfield (ref: common/fasta.c:609-673) with the tables of fasta.c:151-200 / fasta.h:26-42; host code,
    works without a GPU."""
    import numpy as np
    from shrimp_amd import gmapper as gm, synth
    text = b"ACGTUMRWSYKVHDBNacgtumrwsykvhdbnXx." + b"ACGTACGTAC"
    want = np.array(list(range(16)) * 2 + [15, 15, 15] + [0, 1, 2, 3, 0, 1, 2, 3, 0, 1], dtype=np.uint8)
    words, ib = gm.sequence_to_bitfield(text)
    assert ib is None and (words == synth.pack_reads(want[None, :])[0]).all()
    words, ib = gm.sequence_to_bitfield(b"t0123.4NnXx3210", colour_space=True)
    assert ib == 3 and (words == synth.pack_reads(np.array([[0, 1, 2, 3, 15, 15, 15, 15, 15, 15, 3, 2, 1, 0]], dtype=np.uint8))[0]).all()
    import pytest
    with pytest.raises(gm.GmError): gm.sequence_to_bitfield(b"ACGT0")                       # a colour in a letter-space read (the reference exits, fasta.c:639-650)
    with pytest.raises(gm.GmError): gm.sequence_to_bitfield(b"N0123", colour_space=True)    # no primer letter (the reference drops the read, :626-634)
    with pytest.raises(gm.GmError): gm.sequence_to_bitfield(b"TACGT", colour_space=True)    # letters in a colour-space read


def _simple_reads(path):
    """(name, seq, qual or None) of an unfolded FASTA / FASTQ fixture"""
    import gzip
    L = gzip.open(path, "rb").read().split(b"\n")
    if L[0][:1] == b"@": return [(L[i][1:].split()[0], L[i + 1], L[i + 3]) for i in range(0, len(L) - 3, 4)]
    return [(L[i][1:].split()[0], L[i + 1], None) for i in range(0, len(L) - 1, 2)]


def test_read_preprocessing_matches_reference_goldens():
    """A22's read-loop preprocessing (ref: gmapper.c:262-284 trim_read, :427-472 trimming / Illumina B tails / quality checks, :495-521 dropped reads) through
    gm_preprocess_read_text, host code only: which reads the reference gave a record at all, and the SEQ / QUAL it printed for the forward-strand and the unaligned
    ones, under --trim-front/--trim-end, --trim-illumina, --min-avg-qv (10 by default), --ignore-qvs"""
    import gzip, os
    from shrimp_amd import gmapper as gm
    from tools.make_golden import PREPROCESS_CASES
    G = os.path.join(os.path.dirname(__file__), "golden")
    tr = bytes.maketrans(b"RYSWKMBDHVacgtn", b"NNNNNNNNNNACGTN")
    ndrop = 0
    for tag, (src, _, fields) in PREPROCESS_CASES.items():
        p = gm.default_params()
        for k, v in fields.items(): setattr(p, k, v)
        kept = []
        for name, seq, qual in _simple_reads(os.path.join(G, src)):
            sq, ql, drop = gm.preprocess_read(p, seq, qual, 64)
            if drop: ndrop += 1
            else: kept.append((name, sq, ql))
        recs = {}
        order = []
        for l in gzip.open(os.path.join(G, tag + ".sam.gz"), "rb").read().split(b"\n"):
            if not l or l[:1] == b"@": continue
            t = l.split(b"\t")
            if t[0] not in recs: recs[t[0]] = t; order.append(t[0])
        assert order == [k[0] for k in kept], (tag, len(order), len(kept))
        for name, sq, ql in kept:
            t = recs[name]
            if int(t[1]) & 16: continue
            assert t[9] == sq.translate(tr), (tag, name, t[9], sq)
            if ql is not None: assert t[10] == (ql if int(t[1]) & 4 else bytes(c - 31 for c in ql)), (tag, name)     # (unaligned: verbatim, ref: output.c:419-421; mapped: PHRED+33, :539-570)
    assert ndrop > 300
    # what is refused: the first mate of a pair (the reference trims it after packing it), a quality value out of range
    p = gm.default_params(); p.trim_front = 2
    with pytest.raises(RuntimeError): gm.preprocess_read(p, b"ACGTACGTAC", None, 64, mate=1)
    p = gm.default_params()
    with pytest.raises(RuntimeError): gm.preprocess_read(p, b"ACGT", b"!!!!", 64)
    p.no_qv_check = 1; p.min_avg_qv = -1
    assert gm.preprocess_read(p, b"ACGT", b"!!!!", 64) == (b"ACGT", b"!!!!", False)
    # an RNA read (uracil, no thymine: the reference complements its A to U, ref: fasta.c:528-542, util.h:125-151) passes like any other -- the device tells it from its letters
    assert gm.preprocess_read(gm.default_params(), b"ACGUACGUACGUACGU", None, 64) == (b"ACGUACGUACGUACGU", None, False)
    assert gm.preprocess_read(gm.default_params(), b"ACGUACGTACGUACGT", None, 64)[2] is False
