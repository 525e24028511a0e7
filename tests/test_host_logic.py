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
