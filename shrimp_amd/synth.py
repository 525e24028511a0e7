"""Deterministic synthetic genomes and reads.

Pure numpy so the very same inputs can be produced in the build container (for the
reference binary / oracle) and on the GPU box (for bench.py and the parity tests).
Nothing here is part of the mapping path.

Base codes follow the reference's 4-bit alphabet (common/fasta.h:26-42): A0 C1 G2 T3.
"""
from __future__ import annotations

import numpy as np

LETTERS = np.frombuffer(b"ACGT", dtype=np.uint8)
# complement_base table over all 16 codes (common/util.h:125-151)
COMPLEMENT = np.array([3, 2, 1, 0, 0, 10, 9, 7, 8, 6, 5, 14, 13, 12, 11, 15], dtype=np.uint8)

# hg18 chr1..22,X,Y lengths (Mbp, rounded) used only as *proportions* for the 24-contig cfg3 genome
_HG18_MBP = [247, 243, 199, 191, 181, 171, 159, 146, 140, 135, 134, 132, 114, 106, 100, 89, 79, 76,
             64, 62, 47, 50, 155, 58]


def contig_lengths(cfg: str, scale: float = 1.0) -> list[int]:
    """Contig lengths for the named configuration (`scale` shrinks it for CPU-sized tests)."""
    if cfg == "cfg1":
        base = [1_000_000]
    elif cfg == "cfg2":
        base = [25_000_000] * 4
    elif cfg == "cfg3":
        tot = float(sum(_HG18_MBP))
        base = [int(3.0e9 * m / tot) for m in _HG18_MBP]
    else:
        raise ValueError(cfg)
    return [max(1000, int(b * scale)) for b in base]


def make_genome(lengths: list[int], seed: int) -> list[np.ndarray]:
    """i.i.d. uniform ACGT contigs as uint8 code arrays (0..3)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    return [rng.integers(0, 4, size=n, dtype=np.uint8) for n in lengths]


def make_reads(contigs: list[np.ndarray], n_reads: int, read_len: int, seed: int,
               p_sub: float = 0.02, p_ins: float = 0.002, p_del: float = 0.002,
               p_rc: float = 0.5):
    """Sample reads: uniform start over the concatenated genome, `p_rc` reverse-complemented,
    then per-base substitution / insertion / deletion.  Returns (codes[n_reads, read_len] uint8,
    truth dict of arrays cn, pos, strand)."""
    rng = np.random.Generator(np.random.PCG64(seed + 1_000_003))
    lens = np.array([len(c) for c in contigs], dtype=np.int64)
    margin = read_len + 16
    usable = np.maximum(lens - margin, 1)
    cum = np.concatenate([[0], np.cumsum(usable)])
    u = rng.integers(0, cum[-1], size=n_reads, dtype=np.int64)
    cn = np.searchsorted(cum, u, side="right") - 1
    pos = u - cum[cn]
    strand = (rng.random(n_reads) < p_rc).astype(np.uint8)

    # source segments of length `margin`, oriented like the read
    src = np.empty((n_reads, margin), dtype=np.uint8)
    ar = np.arange(margin, dtype=np.int64)
    for c in range(len(contigs)):
        m = np.nonzero(cn == c)[0]
        if m.size == 0:
            continue
        src[m] = contigs[c][pos[m, None] + ar[None, :]]
    rc = strand == 1
    src[rc] = COMPLEMENT[src[rc][:, ::-1]]

    out = np.empty((n_reads, read_len), dtype=np.uint8)
    ptr = np.zeros(n_reads, dtype=np.int64)
    rows = np.arange(n_reads)
    for j in range(read_len):
        r = rng.random(n_reads)
        ins = r < p_ins
        dele = (~ins) & (r < p_ins + p_del)
        ptr = np.minimum(ptr + dele, margin - 1)
        b = src[rows, ptr]
        sub = (~ins) & (rng.random(n_reads) < p_sub)
        b = np.where(sub & (b < 4), (b + rng.integers(1, 4, size=n_reads, dtype=np.uint8)) & 3, b)
        b = np.where(ins, rng.integers(0, 4, size=n_reads, dtype=np.uint8), b)
        out[:, j] = b
        ptr = np.minimum(ptr + (~ins), margin - 1)
    return out, {"cn": cn.astype(np.int32), "pos": pos, "strand": strand}


def make_pairs(contigs: list[np.ndarray], n_pairs: int, read_len: int, seed: int, ins_mean: float = 300.0, ins_sd: float = 30.0,
               ins_min: int = 160, p_sub: float = 0.01):
    """opp-in pairs: a fragment of length ~N(ins_mean, ins_sd) (clipped to >= ins_min and >= read_len),
    mate 1 = its first read_len bases, mate 2 = reverse complement of its last read_len bases; the fragment is taken from either
    strand with equal probability.  Returns codes[2*n_pairs, read_len] with mates adjacent."""
    rng = np.random.Generator(np.random.PCG64(seed + 2_000_003))
    lens = np.array([len(c) for c in contigs], dtype=np.int64)
    ins = np.maximum(np.maximum(np.rint(rng.normal(ins_mean, ins_sd, n_pairs)).astype(np.int64), ins_min), read_len)
    maxins = int(ins.max())
    usable = np.maximum(lens - maxins - 1, 1)
    cum = np.concatenate([[0], np.cumsum(usable)])
    u = rng.integers(0, cum[-1], size=n_pairs, dtype=np.int64)
    cn = np.searchsorted(cum, u, side="right") - 1
    pos = u - cum[cn]
    strand = (rng.random(n_pairs) < 0.5)
    out = np.empty((2 * n_pairs, read_len), dtype=np.uint8)
    ar = np.arange(read_len, dtype=np.int64)
    for c in range(len(contigs)):
        m = np.nonzero(cn == c)[0]
        if m.size == 0:
            continue
        left = contigs[c][pos[m, None] + ar[None, :]]                                  # fragment start, + strand
        right = contigs[c][(pos[m] + ins[m] - read_len)[:, None] + ar[None, :]]         # fragment end, + strand
        fwd = ~strand[m]
        m1 = np.where(fwd[:, None], left, COMPLEMENT[right[:, ::-1]])
        m2 = np.where(fwd[:, None], COMPLEMENT[right[:, ::-1]], left)
        out[2 * m] = m1
        out[2 * m + 1] = m2
    sub = (rng.random(out.shape) < p_sub) & (out < 4)
    out = np.where(sub, (out + rng.integers(1, 4, size=out.shape, dtype=np.uint8)) & 3, out).astype(np.uint8)
    return out, {"cn": cn.astype(np.int32), "pos": pos, "ins": ins, "strand": strand}


def make_cs_reads(contigs: list[np.ndarray], n_reads: int, n_colours: int, seed: int, p_col: float = 0.04, p_rc: float = 0.5,
                  p_dot: float = 0.0):
    """Colour-space (SOLiD) reads as SURVEY.md 8(d) cfg4 describes them: a letter-space fragment of `n_colours` bases
    with exactly one indel of 1-3 bases, translated to colours behind a 'T' primer (colour = XOR of adjacent 2-bit
    codes), then per-colour substitution with probability `p_col` (and, optionally, skipped cycles '.', code 15).
    Returns codes[n_reads, n_colours + 1] uint8 -- column 0 is the primer letter code (3 = T) -- and the truth dict."""
    rng = np.random.Generator(np.random.PCG64(seed + 3_000_003))
    L = n_colours
    lens = np.array([len(c) for c in contigs], dtype=np.int64)
    margin = L + 8
    usable = np.maximum(lens - margin, 1)
    cum = np.concatenate([[0], np.cumsum(usable)])
    u = rng.integers(0, cum[-1], size=n_reads, dtype=np.int64)
    cn = np.searchsorted(cum, u, side="right") - 1
    pos = u - cum[cn]
    strand = (rng.random(n_reads) < p_rc).astype(np.uint8)
    src = np.empty((n_reads, margin), dtype=np.uint8)
    ar = np.arange(margin, dtype=np.int64)
    for c in range(len(contigs)):
        m = np.nonzero(cn == c)[0]
        if m.size:
            src[m] = contigs[c][pos[m, None] + ar[None, :]]
    rc = strand == 1
    src[rc] = COMPLEMENT[src[rc][:, ::-1]]
    src = np.where(src > 3, rng.integers(0, 4, size=src.shape, dtype=np.uint8), src)   # the sequenced molecule has real bases
    at = rng.integers(5, L - 5, size=n_reads)
    k = rng.integers(1, 4, size=n_reads)
    is_ins = rng.random(n_reads) < 0.5
    j = np.arange(L, dtype=np.int64)[None, :]
    idx = np.where(j < at[:, None], j, np.where(is_ins[:, None], np.maximum(j - k[:, None], 0), j + k[:, None]))
    bases = np.take_along_axis(src, idx, axis=1)
    inserted = is_ins[:, None] & (j >= at[:, None]) & (j < (at + k)[:, None])
    bases = np.where(inserted, rng.integers(0, 4, size=bases.shape, dtype=np.uint8), bases).astype(np.uint8)
    prev = np.concatenate([np.full((n_reads, 1), 3, dtype=np.uint8), bases[:, :-1]], axis=1)
    col = prev ^ bases
    err = rng.random(col.shape) < p_col
    col = np.where(err, (col + rng.integers(1, 4, size=col.shape, dtype=np.uint8)) & 3, col).astype(np.uint8)
    if p_dot > 0:
        col = np.where(rng.random(col.shape) < p_dot, 15, col).astype(np.uint8)
    out = np.concatenate([np.full((n_reads, 1), 3, dtype=np.uint8), col], axis=1)
    return out, {"cn": cn.astype(np.int32), "pos": pos, "strand": strand}


def cs_from_letters(codes: np.ndarray, seed: int, p_col: float = 0.03) -> np.ndarray:
    """Letter-space reads as sequenced -> colour-space reads behind a 'T' primer (colour = XOR of adjacent 2-bit codes) with per-colour
    substitutions: codes[n, 1 + L], column 0 the primer letter code (3 = T).  Used for colour-space pairs (mates from make_pairs)."""
    rng = np.random.default_rng(seed + 4_000_004)
    b = np.asarray(codes, dtype=np.uint8) & 3
    prev = np.concatenate([np.full((b.shape[0], 1), 3, np.uint8), b[:, :-1]], axis=1)
    col = (prev ^ b) & 3
    err = rng.random(col.shape) < p_col
    col = np.where(err, (col + rng.integers(1, 4, col.shape)) & 3, col).astype(np.uint8)
    return np.concatenate([np.full((b.shape[0], 1), 3, np.uint8), col], axis=1)


def write_csfasta_reads(path: str, reads: np.ndarray) -> None:
    """codes[n, 1 + colours] -> csfasta (primer letter, then colours 0-3 or '.')"""
    n, L = reads.shape
    tab = np.full(16, ord("."), dtype=np.uint8); tab[:4] = np.frombuffer(b"0123", dtype=np.uint8)
    body = tab[reads[:, 1:]]
    with open(path, "wb") as f:
        for i in range(n):
            f.write(b">r%d\n" % i)
            f.write(LETTERS[reads[i, :1]].tobytes() + body[i].tobytes())
            f.write(b"\n")


def pack_nibbles(codes: np.ndarray) -> np.ndarray:
    """Pack a 1-D code array 8 bases per uint32, base i in nibble i%8 of word i/8
    (the reference's bitfield layout, common/util.h:41 EXTRACT).  Little-endian bytes: byte k of
    the stream holds bases 2k (low nibble) and 2k+1 (high nibble)."""
    codes = np.asarray(codes, dtype=np.uint8)
    n = codes.shape[0]
    nw = (n + 7) // 8
    b = np.zeros(nw * 4, dtype=np.uint8)
    ne = (n + 1) // 2
    b[:ne] = codes[0::2] & 0xF
    b[:n // 2] |= (codes[1::2] & 0xF) << 4
    return b.view("<u4")


def pack_reads(codes: np.ndarray) -> np.ndarray:
    """Pack reads[n, L] row-wise to uint32[n, ceil(L/8)]."""
    codes = np.asarray(codes, dtype=np.uint8)
    n, L = codes.shape
    nw = (L + 7) // 8
    b = np.zeros((n, nw * 4), dtype=np.uint8)
    ne = (L + 1) // 2
    b[:, :ne] = codes[:, 0::2] & 0xF
    b[:, :L // 2] |= (codes[:, 1::2] & 0xF) << 4
    return np.ascontiguousarray(b).view("<u4")


def write_fasta_genome(path: str, contigs: list[np.ndarray], width: int = 70) -> None:
    with open(path, "wb") as f:
        for i, c in enumerate(contigs):
            f.write(b">contig%d\n" % (i + 1))
            s = LETTERS[c]
            for k in range(0, len(s), width * 10000):
                blk = s[k:k + width * 10000]
                full = (len(blk) // width) * width
                if full:
                    lines = np.concatenate(
                        [blk[:full].reshape(-1, width),
                         np.full((full // width, 1), 10, dtype=np.uint8)], axis=1)
                    f.write(lines.tobytes())
                if full < len(blk):
                    f.write(blk[full:].tobytes() + b"\n")


def read_names(n_reads: int) -> list[bytes]:
    return [b"r%d" % i for i in range(n_reads)]


def write_fasta_reads(path: str, reads: np.ndarray) -> None:
    n, L = reads.shape
    s = LETTERS[reads]
    with open(path, "wb") as f:
        for i in range(n):
            f.write(b">r%d\n" % i)
            f.write(s[i].tobytes())
            f.write(b"\n")


CONFIGS = {
    # name: (genome cfg, genome seed, n_reads, read_len, read seed)
    "cfg1": ("cfg1", 12345, 10_000, 36, 12345),
    "cfg2": ("cfg2", 2, 1_000_000, 100, 2),
    "cfg3": ("cfg3", 3, 10_000_000, 100, 3),
}


def make_config(name: str, scale: float = 1.0, n_reads: int | None = None):
    g, gseed, nr, L, rseed = CONFIGS[name]
    contigs = make_genome(contig_lengths(g, scale), gseed)
    reads, truth = make_reads(contigs, n_reads if n_reads is not None else nr, L, rseed)
    return contigs, reads, truth
