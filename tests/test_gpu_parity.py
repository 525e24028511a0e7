"""GPU parity tests (run with -m gpu on an MI355X): the HIP path, through the C ABI, against
(a) the reference's own golden outputs, (b) the CPU oracle on seeded inputs."""
import os
import numpy as np
import pytest
from tests import oracle_api as oa

pytestmark = pytest.mark.gpu

GOLDEN = ["cfg1_36bp_1Mbp", "cfg2s_100bp_2Mbp", "stress_60bp", "stress_100bp_unal"]


@pytest.fixture(scope="module")
def gm():
    # torch's wheel carries its own HIP runtime: it must initialise before libgmapper_hip.so's (the order bench.py uses),
    # or torch finds no device later in the same process (test_index_replication_path_of_the_multi_gpu_start_up)
    try:
        import torch
        torch.cuda.init()
    except Exception:
        pass
    from shrimp_amd import gmapper
    if gmapper.lib().gm_device_count() < 1:
        pytest.fail("no HIP device: the product path has no CPU fallback")
    return gmapper


def _first_diff(a: bytes, b: bytes):
    la, lb = a.split(b"\n"), b.split(b"\n")
    for i, (x, y) in enumerate(zip(la, lb)):
        if x != y:
            return i, x[:300], y[:300]
    return min(len(la), len(lb)), b"<eof>", b"<eof>"


def test_sw_vector_known_answers(gm):
    """S1: every vector-SW known answer produced by the reference's own sw_vector()."""
    gm.sw_vector_setup(1400, 1000, -33, -7, -33, -3, 10, -15, 0, True)
    recs = [r for r in oa.load_kat() if r[0] == "V"]
    assert len(recs) >= 1500
    # every record, in ONE launch of the batch entry: the records' genome bitfields are laid end to end (each starts on a word boundary)
    words, goffs, glens, rlens, reads, want = [], [], [], [], [], []
    base = 0; rw_max = max(len(r[5]) for r in recs)
    for _, goff, glen, rlen, g, r, score in recs:
        words.append(g); goffs.append(base * 8 + goff); glens.append(glen); rlens.append(rlen); want.append(score)
        reads.append(np.concatenate([r, np.zeros(rw_max - len(r), dtype=np.uint32)]))
        base += len(g)
    got = gm.sw_vector_batch(np.concatenate(words), goffs, glens, np.stack(reads), rlens)
    bad = np.nonzero(np.asarray(got) != np.asarray(want))[0]
    assert bad.size == 0, (bad[:10], [got[i] for i in bad[:10]], [want[i] for i in bad[:10]])
    # and the single-call form with the reference's parameter list
    _, goff, glen, rlen, g, r, score = recs[0]
    assert gm.sw_vector(g, goff, glen, r, rlen) == score


def test_sw_vector_batch_random_vs_oracle(gm, oracle_lib):
    """>= 10^4 random + adversarial windows in one batch against the CPU restatement."""
    import ctypes as C
    rng = np.random.default_rng(5)
    n, L = 4000, 100
    G = rng.integers(0, 4, size=200_000, dtype=np.uint8)
    G[rng.integers(0, G.size, 300)] = 15
    starts = rng.integers(0, G.size - 400, size=n)
    reads = np.stack([G[s + 20:s + 20 + L].copy() for s in starts])
    mut = rng.random(reads.shape) < 0.06
    reads = np.where(mut, rng.integers(0, 4, size=reads.shape), reads).astype(np.uint8)
    reads[:50] = 0; G[:2000] = 0                      # homopolymer block: all ties
    from shrimp_amd import synth
    gw = synth.pack_nibbles(G); rw = synth.pack_reads(reads)
    glen = np.full(n, 140, dtype=np.int32); glen[::7] = 90   # window shorter than the read
    gm.sw_vector_setup(1400, 1000, -33, -7, -33, -3, 10, -15, 0, True)
    got = gm.sw_vector_batch(gw, starts, glen, rw, np.full(n, L, dtype=np.int32))
    u32p = C.POINTER(C.c_uint32)
    want = np.array([oracle_lib.gmo_sw_vector(gw.ctypes.data_as(u32p), int(starts[i]), int(glen[i]),
                                              np.ascontiguousarray(rw[i]).ctypes.data_as(u32p), L) for i in range(n)])
    assert (got == want).all(), np.nonzero(got != want)[0][:10]


def test_sw_vector_early_stop_is_exact_about_the_threshold(gm):
    """pass 1's early stop (unpaired reads): a window stops only when no alignment can reach the threshold any more -- the full score of every stopped
    window is below the threshold, its returned value is a lower bound of it, and a window that did not stop returns the full score"""
    from shrimp_amd import synth
    rng = np.random.default_rng(17)
    n, L = 6000, 100
    G = rng.integers(0, 4, size=300_000, dtype=np.uint8)
    G[rng.integers(0, G.size, 300)] = 15
    starts = rng.integers(0, G.size - 400, size=n)
    reads = np.stack([G[s + 20:s + 20 + L].copy() for s in starts])
    # a third true hits (6 % substitutions), a third half-reads (the other half random: scores around the threshold), a third chance windows with two seed-like runs
    reads = np.where(rng.random(reads.shape) < 0.06, rng.integers(0, 4, size=reads.shape), reads).astype(np.uint8)
    third = n // 3
    for i in range(third, 2 * third):
        cut = int(rng.integers(30, 70)); side = int(rng.integers(0, 2))
        if side: reads[i, cut:] = rng.integers(0, 4, size=L - cut)
        else: reads[i, :cut] = rng.integers(0, 4, size=cut)
    for i in range(2 * third, n):
        keep = np.zeros(L, dtype=bool)
        for _ in range(2):
            a = int(rng.integers(0, L - 14)); keep[a:a + 14] = True
        reads[i] = np.where(keep, reads[i], rng.integers(0, 4, size=L))
    gw = synth.pack_nibbles(G); rw = synth.pack_reads(reads)
    glen = np.full(n, 140, dtype=np.int32); glen[::7] = 90; glen[3::11] = 250
    rlen = np.full(n, L, dtype=np.int32)
    gm.sw_vector_setup(1400, 1000, -33, -7, -33, -3, 10, -15, 0, True)
    full = gm.sw_vector_batch(gw, starts, glen, rw, rlen)
    n_stopped = 0
    for thr in (470, 300, 700):
        got, stopped = gm.sw_vector_batch_bounded(gw, starts, glen, rw, rlen, thr)
        st = stopped.astype(bool)
        assert (got[~st] == full[~st]).all()
        assert (full[st] < thr).all(), np.nonzero(st & (full >= thr))[0][:10]
        assert (got[st] <= full[st]).all() and (got[st] >= 0).all()
        assert ((full >= thr) <= ~st).all()
        n_stopped += int(st.sum())
        if thr == 470: assert st.sum() > n // 4 and (full >= thr).sum() > n // 4      # both outcomes are well represented
    assert n_stopped > 0


@pytest.mark.parametrize("L", [150, 200])
def test_sw_vector_early_stop_two_stripes_is_exact_about_the_threshold(gm, L):
    """the same for reads of more than 128 bases (two stripes of 64 lanes x 2 rows; 2 x 150 bp pairs' unpaired pass): a window may stop in the drain of the first stripe
    (the rows of the second stripe count as rows left, and the alignments that already left through the stripe's last row are bounded from the carry row), between the
    stripes, or in the drain of the second -- every stopped window's full score is below the threshold, every other window returns the full score"""
    from shrimp_amd import synth
    rng = np.random.default_rng(29 + L)
    n = 4500
    G = rng.integers(0, 4, size=400_000, dtype=np.uint8)
    G[rng.integers(0, G.size, 300)] = 15
    starts = rng.integers(0, G.size - 600, size=n)
    reads = np.stack([G[s + 25:s + 25 + L].copy() for s in starts])
    reads = np.where(rng.random(reads.shape) < 0.06, rng.integers(0, 4, size=reads.shape), reads).astype(np.uint8)
    third = n // 3
    for i in range(third, 2 * third):          # part of the read random: the true part lies in the first stripe, in the second, or across the seam
        a = int(rng.integers(0, L - 40)); b = int(rng.integers(a + 30, L))
        keep = np.zeros(L, dtype=bool); keep[a:b] = True
        reads[i] = np.where(keep, reads[i], rng.integers(0, 4, size=L))
    for i in range(2 * third, n):              # chance windows with two seed-like runs, one of them below row 128
        keep = np.zeros(L, dtype=bool)
        a = int(rng.integers(0, 100)); keep[a:a + 14] = True
        a = int(rng.integers(128, L - 14)); keep[a:a + 14] = True
        reads[i] = np.where(keep, reads[i], rng.integers(0, 4, size=L))
    gw = synth.pack_nibbles(G); rw = synth.pack_reads(reads)
    glen = np.full(n, int(L * 1.4), dtype=np.int32); glen[::7] = L - 10; glen[3::11] = 2 * L
    rlen = np.full(n, L, dtype=np.int32)
    gm.sw_vector_setup(1400, 1000, -33, -7, -33, -3, 10, -15, 0, True)
    full = gm.sw_vector_batch(gw, starts, glen, rw, rlen)
    n_stopped = 0
    for thr in (int(4.7 * L), 3 * L, 7 * L, 9 * L):
        got, stopped = gm.sw_vector_batch_bounded(gw, starts, glen, rw, rlen, thr)
        st = stopped.astype(bool)
        assert (got[~st] == full[~st]).all()
        assert (full[st] < thr).all(), np.nonzero(st & (full >= thr))[0][:10]
        assert (got[st] <= full[st]).all() and (got[st] >= 0).all()
        assert ((full >= thr) <= ~st).all()
        n_stopped += int(st.sum())
        if thr == int(4.7 * L): assert st.sum() > n // 5 and (full >= thr).sum() > n // 5
    assert n_stopped > 0


def test_sw_vector_long_reads_multi_stripe(gm, oracle_lib):
    """reads longer than 128 rows exercise the stripe carry"""
    import ctypes as C
    from shrimp_amd import synth
    rng = np.random.default_rng(9)
    n, L, W = 64, 300, 420
    G = rng.integers(0, 4, size=50_000, dtype=np.uint8)
    starts = rng.integers(0, G.size - 600, size=n)
    reads = np.stack([G[s + 40:s + 40 + L].copy() for s in starts])
    reads = np.where(rng.random(reads.shape) < 0.04, rng.integers(0, 4, size=reads.shape), reads).astype(np.uint8)
    gw = synth.pack_nibbles(G); rw = synth.pack_reads(reads)
    gm.sw_vector_setup(1400, 1000, -33, -7, -33, -3, 10, -15, 0, True)
    got = gm.sw_vector_batch(gw, starts, np.full(n, W, dtype=np.int32), rw, np.full(n, L, dtype=np.int32))
    u32p = C.POINTER(C.c_uint32)
    want = np.array([oracle_lib.gmo_sw_vector(gw.ctypes.data_as(u32p), int(starts[i]), W,
                                              np.ascontiguousarray(rw[i]).ctypes.data_as(u32p), L) for i in range(n)])
    assert (got == want).all()


def test_sw_full_ls_known_answers(gm):
    """S2: full SW + traceback against the reference's own sw_full_ls() answers."""
    gm.sw_full_ls_setup(1400, 1000, -33, -7, -33, -3, 10, -15, True, 8)
    n = 0
    for rec in oa.load_kat():
        if rec[0] != "F": continue
        _, goff, glen, rlen, ax, ay, alen, aw, rv, g, r, exp, edb, eqr = rec
        f, db, qr = gm.sw_full_ls(g, goff, glen, r, rlen, (ax, ay, alen, aw), revcmpl=bool(rv))
        got = [f[k] for k in ("score", "read_start", "rmapped", "genome_start", "gmapped", "matches", "mismatches", "insertions", "deletions")]
        assert got == exp, (n, goff, glen, rlen, ax, ay, alen, aw, rv, got, exp)
        assert db == edb and qr == eqr
        n += 1
    assert n >= 2990                                   # every record of the reference's own sw_full_ls (global mode)
    inv, cells, secs = gm.seam_stats("sw_full_ls")      # ref: sw_full_ls_stats, gmapper.c:745
    assert inv == n and cells > 0 and secs > 0


@pytest.mark.parametrize("name", GOLDEN)
def test_sam_matches_reference_golden(gm, name):
    """S4/S5: index build + full pipeline -> SAM, byte-identical to the reference binary's output."""
    contigs, reads, sam = oa.load_golden(name)
    p = gm.default_params()
    p.sam_unaligned = 1 if name.endswith("_unal") else 0
    ix = gm.Index(contigs, params=p)
    s = gm.Session(ix, params=p, max_batch_reads=4096)
    got = oa.sam_header(contigs) + s.map_reads(reads)
    st = s.stats
    s.close(); ix.close()
    assert got == sam, (_first_diff(got, sam), st)


KERNEL_VARIANTS = [
    {"GM_NO_BUCKETS": "1"},                                  # generic lookup kernels on a one-slab index
    {"GM_SLAB_BITS": "18"},                                  # several slabs, few list entries per read-strand: the slab-sweep lane-group kernel (k_lookup_v3)
    {"GM_SLAB_BITS": "18", "GM_K1_V4": "1"},                 # k_lookup_v4 forced (hashed pre-count, exact count per bin on the candidates, bin borders)
    {"GM_SLAB_BITS": "17", "GM_K1_V4": "1", "GM_K1_THREADS": "128"},   # more lists than lane groups
    {"GM_SLAB_BITS": "13", "GM_K1_V4": "1", "GM_K4_TABBITS": "12"},    # v4 with a folded table far smaller than the genome: many false candidates, 2^2-region bins
    {"GM_SLAB_BITS": "18", "GM_K1_V4": "1", "GM_K4_BINCAP": "16"},     # v4 candidate bins overflow: read-strands redone by the slab-sweep kernel in list mode
    {"GM_SLAB_BITS": "18", "GM_K1_V4": "1", "GM_K4_WCAP": "8"},        # v4 window -> list by binary search
    {"GM_SLAB_BITS": "17", "GM_K1_THREADS": "128"},          # v3 with more lists than lane groups x register windows
    {"GM_SLAB_BITS": "18", "GM_K1_V4": "1", "GM_SCAP": "256", "GM_SCAP2": "64"},   # v4 survivors beyond the LDS tiers: heavy tier (re-emission by the lane-per-list kernel)
    {"GM_SLAB_BITS": "18", "GM_K1_V2": "1"},                 # the lane-per-list kernel (also the heavy tier's re-emission)
    {"GM_NO_PRUNE": "1"},                                    # K2 on the unpruned survivors
    {"GM_PRUNE_V1": "1"},                                    # the first prune kernel (256-base bins, compare-and-swap updates) for every read-strand
    {"GM_PRUNE_HBITS": "6"},                                 # k_prune_v2 with a 64-slot table: most read-strands overflow it and go to k_prune in list mode
    {"GM_SCAP": "256", "GM_SCAP2": "64"},                    # small LDS tiers: most read-strands take the heavy tier
    {"GM_SLAB_BITS": "18", "GM_K1_V5": "1"},                 # k_lookup_v5 forced (wave-per-list streaming, strip lists, region table + fused prune in LDS)
    {"GM_NO_BUCKETS": "1", "GM_K1_V5": "1"},                 # v5 on a one-slab index
    {"GM_SLAB_BITS": "18", "GM_K1_V5": "1", "GM_K5_LSW": "12"},    # v5 with 2^17 folded counters, a 1 024-slot region table, 16 x 40 candidate records
    {"GM_SLAB_BITS": "18", "GM_K1_V5": "1", "GM_K5_LSW": "9"},     # v5 with tiny tables (64 candidate records): most read-strands fall back to the slab-sweep kernel + K1b in list mode
    {"GM_SLAB_BITS": "18", "GM_K1_V5": "1", "GM_K5_LSW": "11", "GM_K5_CANDLIMIT": "40"},   # v5 mixed: some read-strands in LDS, some fall back
    {"GM_SLAB_BITS": "18", "GM_K1_V5": "1", "GM_NO_PRUNE": "1"},   # v5 without the prune rules (all survivors to K2)
    {"GM_SLAB_BITS": "18", "GM_K1_V5": "1", "GM_SCAP": "256", "GM_SCAP2": "64"},   # v5 survivors beyond K2's LDS tier: heavy tier
    {"GM_SLAB_BITS": "17", "GM_K1_V5": "1", "GM_K1_THREADS": "128"},               # v5 with two waves per workgroup
    {"GM_SLAB_BITS": "18", "GM_K1_V5": "1", "GM_K5_ROUNDS": "3"},   # v5 with pass B + exact stages per third of the genome (k_lookup_v5_rounds: what 2 x 150 bp reads on 3 Gbp take)
    {"GM_NO_BUCKETS": "1", "GM_K1_V5": "1", "GM_K5_ROUNDS": "2"},   # ... per half, one-slab index
    {"GM_SLAB_BITS": "18", "GM_K1_V5": "1", "GM_K5_ROUNDS": "4", "GM_NO_PRUNE": "1"},   # ... four parts, without the prune rules
    {"GM_SLAB_BITS": "18", "GM_K1_V5": "1", "GM_K5_ROUNDS": "3", "GM_SCAP": "256", "GM_SCAP2": "64"},   # ... more kept than K2's LDS tier takes: the whole read-strand falls back
    {"GM_SLAB_BITS": "18", "GM_K1_V5": "1", "GM_K5_HALF": "1", "GM_K5_ROUNDS": "3"},   # the half-size shape (k_lookup_v5_half: two 512-thread workgroups per CU, Bloom bits in seen[], rounds)
    {"GM_NO_BUCKETS": "1", "GM_K1_V5": "1", "GM_K5_HALF": "1", "GM_K5_ROUNDS": "2"},   # ... one-slab index, two parts
    {"GM_SLAB_BITS": "18", "GM_K1_V5": "1", "GM_K5_HALF": "1", "GM_K5_ROUNDS": "1", "GM_NO_PRUNE": "1"},   # ... one part, without the prune rules
    {"GM_SLAB_BITS": "18", "GM_K1_V5": "1", "GM_K5_HALF": "2", "GM_K5_ROUNDS": "4"},   # ... the variant without the Bloom bits, four parts
    {"GM_SLAB_BITS": "18", "GM_K1_V5": "1", "GM_K5_HALF": "1", "GM_K5_ROUNDS": "2", "GM_K5_CANDLIMIT": "40"},   # ... most read-strands fall back
    {"GM_P1_EARLY": "0"},                                    # pass 1 without the early stop of windows that cannot reach the threshold
    {"GM_P2_G": "8"},                                        # pass 2 with eight windows a wave and int16_t carry rows (the default in letter space: four, int)
]


@pytest.mark.parametrize("env", KERNEL_VARIANTS, ids=lambda e: ",".join("%s=%s" % kv for kv in e.items()))
@pytest.mark.parametrize("name", ["cfg2s_100bp_2Mbp", "stress_60bp"])
def test_kernel_variants_match_reference_golden(gm, name, env):
    """every lookup / prune / anchor code path (bucket, lane-per-list, lane-group, multi-slab, heavy tier) gives the reference's SAM"""
    import os
    contigs, reads, sam = oa.load_golden(name)
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        ix = gm.Index(contigs)
        s = gm.Session(ix, max_batch_reads=4096)
        got = oa.sam_header(contigs) + s.map_reads(reads)
        st = s.stats
        kern = gm.lib().gm_last_lookup_kernel().decode()
        s.close(); ix.close()
    finally:
        for k, v in old.items():
            if v is None: os.environ.pop(k, None)
            else: os.environ[k] = v
    assert got == sam, (_first_diff(got, sam), st)
    want_kern = ("k_lookup_v5_half" if "GM_K5_HALF" in env else "k_lookup_v5_rounds" if "GM_K5_ROUNDS" in env else "k_lookup_v5") if "GM_K1_V5" in env else ("k_lookup_v4" if "GM_K1_V4" in env else None)
    assert want_kern is None or kern == want_kern, kern


@pytest.mark.parametrize("tag", sorted(oa.OPTION_CASES))
def test_option_sets_match_reference_golden(gm, tag):
    """non-default options through gm_params_t / custom seeds: --strata, --max-alignments, -o, scores, thresholds, window geometry,
    anchor width, cutoff; paired --strata -- byte-identical to the reference binary run with the same options"""
    base, _, fields, seeds = oa.OPTION_CASES[tag]
    want = oa.load_option_sam(base, tag)
    p = gm.default_params()
    for k, v in fields.items(): setattr(p, k, v)
    if base.startswith("stress_pairs") or base.startswith("chimeric_pairs"):
        g = oa.load_golden_pairs(base)
        ix = gm.Index(g["contigs"], names=g["contig_names"], seeds=seeds, params=p)
        s = gm.Session(ix, params=p, max_batch_reads=4096)
        got = oa.sam_header(g["contigs"], g["contig_names"]) + s.map_pairs(g["m1"], g["m2"], g["names1"], g["names2"], mode=g["mode"],
                                                                            min_insert=g["ins"][0], max_insert=g["ins"][1])
    else:
        contigs, reads, _ = oa.load_golden(base)
        ix = gm.Index(contigs, seeds=seeds, params=p)
        s = gm.Session(ix, params=p, max_batch_reads=4096)
        got = oa.sam_header(contigs) + s.map_reads(reads)
    st = s.stats
    s.close(); ix.close()
    assert got == want, (_first_diff(got, want), st)


def test_mirna_mode_matches_reference_golden(gm):
    """gmapper -M mirna (gmapper.c:1497-1517): the bundle -H, -U, anchor width 0, gap opens -255, no f1 cache, -n 1, window 100 %, --local with the mode's five
    seeds (span 20, weight 14, zeros at both ends) on 22-base reads -- byte-identical SAM"""
    contigs, reads, want = oa.load_golden("mirna_22bp")
    _, fields, seeds = oa.MIRNA_MODE
    p = gm.default_params()
    for k, v in fields.items(): setattr(p, k, v)
    ix = gm.Index(contigs, seeds=seeds, params=p); s = gm.Session(ix, params=p, max_batch_reads=4096)
    got = oa.sam_header(contigs) + s.map_reads(reads)
    st = s.stats
    s.close(); ix.close()
    assert got == want, (_first_diff(got, want), st)


def test_one_session_through_every_mode_in_turn(gm):
    """one session, its buffer sets and capacities re-chosen as the calls change mode: unpaired, pairs -n 3, unpaired, pairs (default), pairs -n 2, pairs --no-half-paired,
    pairs -n 3 --no-half-paired, unpaired -- every output equals its reference golden"""
    g = oa.load_golden_pairs("stress_pairs_2x100")
    names = g["contig_names"]
    reads = np.concatenate([g["m1"][:400], g["m2"][:400]])
    ix = gm.Index(g["contigs"], names=names); s = gm.Session(ix, max_batch_reads=512)
    o = oa.Session(g["contigs"], names); want_u = o.map_sam(reads, nthreads=8); o.close()
    def pairs(**f):
        opts = gm.PairOpts.default(g["mode"], g["ins"][0], g["ins"][1])
        for k, v in f.items(): setattr(opts, k, v)
        return oa.sam_header(g["contigs"], names) + s.map_pairs(g["m1"], g["m2"], g["names1"], g["names2"], opts=opts)
    assert s.map_reads(reads) == want_u
    assert pairs(match_mode=3) == oa.load_option_sam("stress_pairs_2x100", "pairs_n3")
    assert s.map_reads(reads) == want_u
    assert pairs() == g["sam"]
    assert pairs(match_mode=2) == oa.load_option_sam("stress_pairs_2x100", "pairs_n2")
    assert pairs(half_paired=0) == oa.load_option_sam("stress_pairs_2x100", "no_half_paired")
    assert pairs(match_mode=3, half_paired=0) == oa.load_option_sam("stress_pairs_2x100", "pairs_n3_nhp")
    assert pairs() == g["sam"]
    assert s.map_reads(reads) == want_u
    s.close(); ix.close()


def test_two_sessions_on_one_device_from_two_threads(gm):
    """two sessions mapping at the same time from two host threads (ctypes drops the GIL during the calls): the lookup kernels' per-device scratch exists twice and
    the library hands a set to each call in flight -- both outputs equal the golden, repeatedly"""
    import threading
    contigs, reads, sam = oa.load_golden("cfg2s_100bp_2Mbp")
    g = oa.load_golden_pairs("cfg5s_2x150_1Mbp")
    ix = gm.Index(contigs); ixp = gm.Index(g["contigs"], names=g["contig_names"])
    s1 = gm.Session(ix, max_batch_reads=2048); s2 = gm.Session(ixp, max_batch_reads=1024)
    out = {"a": [], "b": []}
    def run_a():
        for _ in range(4): out["a"].append(oa.sam_header(contigs) + s1.map_reads(reads))
    def run_b():
        for _ in range(4): out["b"].append(oa.sam_header(g["contigs"], g["contig_names"]) + s2.map_pairs(g["m1"], g["m2"], g["names1"], g["names2"], mode=g["mode"], min_insert=g["ins"][0], max_insert=g["ins"][1]))
    ta = threading.Thread(target=run_a); tb = threading.Thread(target=run_b)
    ta.start(); tb.start(); ta.join(); tb.join()
    s1.close(); s2.close(); ix.close(); ixp.close()
    assert len(out["a"]) == 4 and all(o == sam for o in out["a"])
    assert len(out["b"]) == 4 and all(o == g["sam"] for o in out["b"])


def test_three_threads_share_the_two_scratch_sets_of_a_device(gm):
    """three sessions mapping at once through k_lookup_v5 with tables so small that read-strands fall back (the fall-back lists are the per-device scratch): two calls are
    in flight, the third waits for a set -- every output equals the golden, repeatedly"""
    import threading
    contigs, reads, sam = oa.load_golden("cfg2s_100bp_2Mbp")
    env = {"GM_SLAB_BITS": "18", "GM_K1_V5": "1", "GM_K5_LSW": "11", "GM_K5_CANDLIMIT": "40"}
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        ix = gm.Index(contigs)
        ss = [gm.Session(ix, max_batch_reads=1024) for _ in range(3)]
        outs = [[] for _ in ss]
        def run(i):
            for _ in range(3): outs[i].append(oa.sam_header(contigs) + ss[i].map_reads(reads))
        th = [threading.Thread(target=run, args=(i,)) for i in range(3)]
        for x in th: x.start()
        for x in th: x.join()
        kern = gm.lib().gm_last_lookup_kernel().decode()
        for s in ss: s.close()
        ix.close()
    finally:
        for k, v in old.items():
            if v is None: os.environ.pop(k, None)
            else: os.environ[k] = v
    for o in outs: assert len(o) == 3 and all(x == sam for x in o)


def test_n1_on_noisy_reads_matches_reference_golden(gm):
    """match_mode 1 (-n 1) where it matters: 70-base reads with 9 % substitutions, 57 of which map only because ONE k-mer match is enough -- the lookup kernel keeps
    every list entry (no region counts, gmapper.c:2610-2616), a window per anchor, pass 1 with min_matches 1"""
    contigs, reads, want = oa.load_golden("n1_noisy_70bp")
    p = gm.default_params(); p.match_mode = 1
    ix = gm.Index(contigs, params=p); s = gm.Session(ix, params=p, max_batch_reads=4096)
    got = oa.sam_header(contigs) + s.map_reads(reads)
    st = s.stats
    s.close(); ix.close()
    assert got == want, (_first_diff(got, want), st)
    # reads built to have ONE list entry each: a region marked once must still give an anchor (390 of 400 map in the reference with -n 1 -h 30%, none without -n 1)
    contigs, reads, want = oa.load_golden("n1_onehit_60bp")
    p = gm.default_params(); p.match_mode = 1; p.sw_full_threshold = 30.0; p.sw_vect_threshold = 30.0
    ix = gm.Index(contigs, params=p); s = gm.Session(ix, params=p, max_batch_reads=4096)
    got = oa.sam_header(contigs) + s.map_reads(reads)
    st = s.stats
    s.close(); ix.close()
    assert got == want, (_first_diff(got, want), st)


@pytest.mark.parametrize("tag", sorted(oa.SAM_TAIL_CASES))
def test_sam_record_tail_options_match_reference_golden(gm, tag):
    """--extra-sam-fields (ZM / ZR / ZV / ZH / ZE: the window's match count and scores, the edit string, reversed on the reverse strand), --read-group (RG:Z) and
    --sam-r2 (R2:Z / X2:Z: the mate's sequence) on mapped, half-mapped and unaligned records, letter and colour space, unpaired and paired"""
    base, fields, cs, pairs = oa.SAM_TAIL_CASES[tag]
    want = oa.load_option_sam(base, tag)
    p = gm.default_params_cs() if cs else gm.default_params()
    for k, v in fields.items(): setattr(p, k, v)
    if pairs:
        g = oa.load_golden_pairs(base)
        ix = gm.Index(g["contigs"], names=g["contig_names"], params=p); s = gm.Session(ix, params=p, max_batch_reads=4096)
        fn = s.map_pairs_cs if cs else s.map_pairs
        got = oa.sam_header(g["contigs"], g["contig_names"]) + fn(g["m1"], g["m2"], g["names1"], g["names2"], mode=g["mode"], min_insert=g["ins"][0], max_insert=g["ins"][1])
    else:
        contigs, reads, _ = oa.load_golden(base)
        ix = gm.Index(contigs, params=p); s = gm.Session(ix, params=p, max_batch_reads=4096)
        got = oa.sam_header(contigs) + (s.map_reads_cs(reads) if cs else s.map_reads(reads))
    st = s.stats
    s.close(); ix.close()
    assert got == want, (_first_diff(got, want), st)


def _idxfix(which="idxfix"):
    import gzip, os
    d = os.path.join(oa.ROOT, "tests", "golden", which)
    z = np.load(os.path.join(d, "inputs.npz"))
    with gzip.open(os.path.join(d, "from_index.sam.gz"), "rb") as f:
        sam = f.read()
    return d, [z["contig0"], z["contig1"]], [bytes(x) for x in z["contig_names"]], z["reads"], str(z["seeds"]).split(","), sam


def test_index_files_written_by_the_reference_load_and_map(gm):
    """gm_index_load on the reference's own -S files (.genome + .seed.N): the SAM equals the reference's -L run, and the
    index equals the one built here from the same contigs"""
    d, contigs, names, reads, seeds, sam = _idxfix()
    ix = gm.Index.load(os.path.join(d, "idx"))
    s = gm.Session(ix, max_batch_reads=1024)
    got = oa.sam_header(contigs, names) + s.map_reads(reads)
    s.close()
    assert got == sam, _first_diff(got, sam)
    ix2 = gm.Index(contigs, names=names, seeds=seeds)
    for sn in range(2):
        for k in (0, 1, 77, 4095, 4 ** 7 - 1, 12345 % 4 ** 7):
            assert ix.get_list(sn, k).tolist() == ix2.get_list(sn, k).tolist(), (sn, k)
    ix.close(); ix2.close()


def test_index_files_saved_here_are_the_reference_files(gm, tmp_path):
    """gm_index_save output, decompressed, is byte-identical to what stock gmapper -S wrote for the same genome and seeds"""
    import gzip
    d, contigs, names, reads, seeds, sam = _idxfix()
    ix = gm.Index(contigs, names=names, seeds=seeds)
    ix.save(str(tmp_path / "mine"))
    ix.close()
    for suffix in (".genome", ".seed.0", ".seed.1"):
        with gzip.open(os.path.join(d, "idx" + suffix), "rb") as f: want = f.read()
        with gzip.open(str(tmp_path / ("mine" + suffix)), "rb") as f: got = f.read()
        assert got == want, (suffix, len(got), len(want))
    ix = gm.Index.load(str(tmp_path / "mine"))            # and round-trips
    s = gm.Session(ix, max_batch_reads=1024)
    got = oa.sam_header(contigs, names) + s.map_reads(reads)
    s.close(); ix.close()
    assert got == sam


PAIRED = ["pairfix_opp-in", "pairfix_opp-out", "pairfix_col-fw", "pairfix_col-bw", "cfg5s_2x150_1Mbp", "stress_pairs_2x100"]


@pytest.mark.parametrize("name", PAIRED)
def test_paired_sam_matches_reference_golden(gm, name):
    """paired mode (-p, -I): pair-up, paired pass 1/2, half-paired fall-back, paired MAPQ and mate fields --
    byte-identical to the reference binary, incl. its own pairing fixture (mates of 30 and 50 bp) in all four modes."""
    g = oa.load_golden_pairs(name)
    ix = gm.Index(g["contigs"], names=g["contig_names"])
    s = gm.Session(ix, max_batch_reads=4096)
    got = oa.sam_header(g["contigs"], g["contig_names"]) + s.map_pairs(g["m1"], g["m2"], g["names1"], g["names2"], mode=g["mode"],
                                                                        min_insert=g["ins"][0], max_insert=g["ins"][1])
    st = s.stats
    s.close(); ix.close()
    assert got == g["sam"], (_first_diff(got, g["sam"]), st)


def test_paired_small_subbatches_and_unpaired_after(gm, oracle_lib):
    """sub-batch boundaries must not show: 300 pairs in sub-batches of 64, then the same session maps unpaired reads"""
    g = oa.load_golden_pairs("stress_pairs_2x100")
    ix = gm.Index(g["contigs"], names=g["contig_names"])
    s = gm.Session(ix, max_batch_reads=64)
    n = 300
    got = s.map_pairs(g["m1"][:n], g["m2"][:n], mode=g["mode"], min_insert=g["ins"][0], max_insert=g["ins"][1])
    o = oa.Session(g["contigs"], g["contig_names"]); o.set_pairing(g["mode"], *g["ins"])
    want = o.map_pairs_sam(g["m1"][:n], g["m2"][:n], nthreads=4)
    assert got == want, _first_diff(got, want)
    got_u = s.map_reads(g["m1"][:n])
    o.set_pairing(0, 0, 1000)
    want_u = o.map_sam(g["m1"][:n], nthreads=4)
    o.close(); s.close(); ix.close()
    assert got_u == want_u, _first_diff(got_u, want_u)


def test_two_stream_pipeline_equals_stage_order(gm):
    """the overlapped sub-batch pipeline (front of sub-batch i+1 beside the back of sub-batch i, ramped sub-batch sizes, SAM text appended
    by the host jobs) gives the same bytes as the strict stage order, unpaired and paired, and both equal the reference golden"""
    import os
    contigs, reads, sam = oa.load_golden("stress_60bp")
    g = oa.load_golden_pairs("stress_pairs_2x100")
    outs = {}
    for mode in ("1", "0"):
        old = os.environ.get("GM_OVERLAP"); os.environ["GM_OVERLAP"] = mode
        os.environ["GM_RAMP_MIN"] = "64"
        try:
            ix = gm.Index(contigs); s = gm.Session(ix, max_batch_reads=256)
            u = oa.sam_header(contigs) + s.map_reads(reads)
            s.close(); ix.close()
            ix = gm.Index(g["contigs"], names=g["contig_names"]); s = gm.Session(ix, max_batch_reads=64)
            p = oa.sam_header(g["contigs"], g["contig_names"]) + s.map_pairs(g["m1"], g["m2"], g["names1"], g["names2"], mode=g["mode"],
                                                                             min_insert=g["ins"][0], max_insert=g["ins"][1])
            s.close(); ix.close()
        finally:
            os.environ.pop("GM_RAMP_MIN", None)
            if old is None: os.environ.pop("GM_OVERLAP", None)
            else: os.environ["GM_OVERLAP"] = old
        outs[mode] = (u, p)
    assert outs["1"][0] == sam, _first_diff(outs["1"][0], sam)
    assert outs["1"] == outs["0"]
    assert outs["1"][1] == g["sam"], _first_diff(outs["1"][1], g["sam"])


def test_tophits_match_oracle_on_stress(gm, oracle_lib):
    """stage parity: the pass-1 survivors (ext-heap array order, scores, anchor boxes)"""
    contigs, reads, _ = oa.load_golden("stress_60bp")
    ix = gm.Index(contigs); s = gm.Session(ix, max_batch_reads=8192)
    got = s.tophits(reads)
    o = oa.Session(contigs); want = o.tophits(reads); o.close()
    s.close(); ix.close()
    assert got.shape == want.shape and (got == want).all(), (got.shape, want.shape)


def test_index_lists_match_oracle_semantics(gm):
    """S5: lists are ascending, skip N-containing spans and never cross contigs (ref: genome.c:1139-1163)."""
    rng = np.random.default_rng(3)
    c1 = rng.integers(0, 4, size=5000, dtype=np.uint8); c1[100:105] = 15
    c2 = rng.integers(0, 4, size=3000, dtype=np.uint8)
    ix = gm.Index([c1, c2])
    mask = "11110111101111"; span = len(mask)
    allc = np.concatenate([c1, c2])
    tot = 0
    for q in list(range(80, 130)) + list(range(4980, 5010)):
        if q + span > len(allc): continue
        win = allc[q:q + span]
        idx = 0
        for t in range(span):                      # mask bit t <-> base q+span-1-t
            if mask[span - 1 - t] == "1": idx = (idx << 2) | int(win[span - 1 - t] & 3)
        lst = ix.get_list(0, idx)
        valid = (15 not in win) and not (q < 5000 < q + span)
        assert (q in lst) == valid, (q, valid)
        assert (np.diff(lst.astype(np.int64)) > 0).all()
        tot += 1
    ix.close()
    assert tot > 50


def test_edge_cases_empty_tiny_long_and_degenerate_reads(gm, oracle_lib):
    """empty input, reads shorter than every seed, reads of exactly one k-mer, all-N and homopolymer reads, 600 and 1000 bp reads
    (multi-stripe vector SW, global back-pointer scratch), over-long reads: HIP == oracle, errors as documented"""
    rng = np.random.default_rng(11)
    contigs, _, _ = oa.load_golden("stress_60bp")
    big = max(contigs, key=len)
    ix = gm.Index(contigs); s = gm.Session(ix, max_batch_reads=256)
    o = oa.Session(contigs)
    assert s.map_reads(np.zeros((0, 50), dtype=np.uint8)) == b""                       # empty batch
    assert s.map_pairs(np.zeros((0, 50), dtype=np.uint8), np.zeros((0, 40), dtype=np.uint8)) == b""
    for L in (8, 13, 14, 15, 19, 20, 33):                                               # around the seed spans 14 / 19 / 19
        st0 = rng.integers(0, len(big) - L - 1, size=40)
        reads = np.stack([big[a:a + L] for a in st0]).astype(np.uint8)
        assert s.map_reads(reads) == o.map_sam(reads, nthreads=2), L
    L = 70
    reads = np.stack([big[a:a + L] for a in rng.integers(0, len(big) - L - 1, size=24)]).astype(np.uint8)
    reads[0] = 15; reads[1] = 0; reads[2] = 3; reads[3, ::2] = 15; reads[4, :40] = 15   # all N, poly-A, poly-T, every other base N, N prefix
    assert s.map_reads(reads) == o.map_sam(reads, nthreads=2)
    for L, n in ((600, 12), (1000, 6)):
        st0 = rng.integers(0, len(big) - L - 1, size=n)
        reads = np.stack([big[a:a + L].copy() for a in st0]).astype(np.uint8)
        mut = rng.random(reads.shape) < 0.03
        reads = np.where(mut, rng.integers(0, 4, size=reads.shape), reads).astype(np.uint8)
        reads[1] = reads[1][::-1].copy()                                                # a read that should not map as is
        assert s.map_reads(reads) == o.map_sam(reads, nthreads=2), L
    with pytest.raises(gm.GmError):                                                     # longer than --longest-read (ref: gmapper.c:497-507 skips it)
        s.map_reads(np.zeros((1, 1001), dtype=np.uint8))
    o.close(); s.close(); ix.close()


def test_index_replication_path_of_the_multi_gpu_start_up(gm):
    """what a non-root rank does at start-up (shrimp_amd/parallel.py): allocate from the metadata blob, receive every resident
    array through a zero-copy torch view of the raw device pointer -- here with a 1-rank RCCL group (the broadcast call itself)
    and, for the data movement, a device-to-device copy between the two indexes' views; the replica must map like the original"""
    import torch
    import torch.distributed as dist
    from shrimp_amd import parallel
    dev = torch.device("cuda", 0)
    for name, env in (("cfg2s_100bp_2Mbp", {}), ("stress_60bp", {"GM_SLAB_BITS": "18"}), ("cfg4s_50col_2Mbp", {})):      # bucket layout / multi-slab layout / colour space
        old = {k: os.environ.get(k) for k in env}; os.environ.update(env)
        try:
            contigs, reads, sam = oa.load_golden(name)
            cs = "col" in name
            par = gm.default_params_cs() if cs else gm.default_params()
            src = gm.Index(contigs, params=par)
            rep = gm.Index.alloc_like(src.meta(), device=0)
            a, b = src.device_arrays(), rep.device_arrays()
            assert [n for _, n in a] == [n for _, n in b] and len(a) == 2 + 3 * 3 and (a[-1][1] > 0) == cs      # last = colour translation of the genome
            for (pa, na), (pb, nb) in zip(a, b):
                if not na: continue
                ta = torch.as_tensor(parallel._DevArray(pa, na), device=dev); tb = torch.as_tensor(parallel._DevArray(pb, nb), device=dev)
                assert ta.data_ptr() == pa and tb.data_ptr() == pb          # views, not copies
                tb.copy_(ta)
            torch.cuda.synchronize()
            s = gm.Session(rep, params=par, max_batch_reads=4096)
            got = oa.sam_header(contigs) + (s.map_reads_cs(reads) if cs else s.map_reads(reads))
            s.close(); rep.close()
            assert got == sam, _first_diff(got, sam)
            if not dist.is_initialized():
                os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29577")
                dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
            same = parallel.broadcast_index(src, 0, dev, src=0)            # 1-rank group: exercises meta + dist.broadcast on the views
            assert same is src
            src.close()
        finally:
            for k, v in old.items():
                if v is None: os.environ.pop(k, None)
                else: os.environ[k] = v
    if dist.is_initialized(): dist.destroy_process_group()


def test_colour_space_kernels_known_answers(gm):
    """S1/S2 in colour space on the GPU: sw_vector with use_colours (first-colour row) and sw_full_cs (4 layers, crossovers,
    traceback, alignment strings) against the reference's own functions: 700 vector + 1400 full-SW known answers"""
    recs = oa.load_kat_cs()
    vec = [r for r in recs if r[0] == "C"]
    gm.sw_vector_setup(1400, 1000, -33, -7, -33, -3, 10, 10 - 20, 1, True)          # mismatch = match + crossover (ref: gmapper.c:2935)
    for _, goff, glen, rlen, initbp, gcs, gls, rd, score in vec:
        got = gm.sw_vector_batch_cs(gcs, gls, [goff], [glen], rd[None, :], [rlen], [initbp])[0]
        assert got == score, (goff, glen, rlen, initbp, got, score)
    _, goff, glen, rlen, initbp, gcs, gls, rd, score = vec[0]                       # the reference's own parameter list
    assert gm.sw_vector(gcs, goff, glen, rd, rlen, genome_ls=gls, initbp=initbp) == score
    gm.sw_full_cs_setup(1400, 1000, -33, -7, -33, -3, 10, -24, -20, True, 8, 0)
    n = 0
    for r in recs:
        if r[0] != "S": continue
        _, (goff, glen, rlen, initbp, ax, ay, alen, awidth, rv, thresh), gls, rd, want, db, qr = r
        f, gdb, gqr = gm.sw_full_cs(gls, goff, glen, rd, rlen, initbp, thresh, (ax, ay, alen, awidth), revcmpl=bool(rv))
        if want[0] == 0:
            assert f["score"] == 0, (f, want)
        else:
            got = [f[k] for k in ("score", "read_start", "rmapped", "genome_start", "gmapped", "matches", "mismatches", "insertions", "deletions", "crossovers")]
            assert got == want and gdb.encode() == db and gqr.encode() == qr, (got, want, gdb, db, gqr, qr)
        n += 1
    assert n >= 1400
    nl = 0
    for r in oa.load_kat_cs("sw_kat_cs_local.txt.gz"):          # local_alignment = true (ref: sw-full-cs.c:199-203,315,439-552)
        _, (goff, glen, rlen, initbp, ax, ay, alen, awidth, rv, thresh), gls, rd, want, db, qr = r
        f, gdb, gqr = gm.sw_full_cs(gls, goff, glen, rd, rlen, initbp, thresh, (ax, ay, alen, awidth), revcmpl=bool(rv), local=True)
        if want[0] == 0:
            assert f["score"] == 0, (f, want)
        else:
            got = [f[k] for k in ("score", "read_start", "rmapped", "genome_start", "gmapped", "matches", "mismatches", "insertions", "deletions", "crossovers")]
            assert got == want and gdb.encode() == db and gqr.encode() == qr, (got, want, gdb, db, gqr, qr)
        nl += 1
    assert nl >= 800
    # is_rna = true on RNA genomes (what gmapper passes when the last contig is RNA, ref: genome.c:1063-1064): U reads as T in the first-colour row, the letter translations hold U for T
    nr = 0
    for r in oa.load_kat_cs("sw_kat_cs_rna.txt.gz"):
        if r[0] == "C":
            _, goff, glen, rlen, initbp, gcs, gls, rd, score = r
            assert gm.sw_vector(gcs, goff, glen, rd, rlen, genome_ls=gls, initbp=initbp, is_rna=True) == score, (goff, glen, rlen, initbp, score)
        else:
            kind, (goff, glen, rlen, initbp, ax, ay, alen, awidth, rv, thresh), gls, rd, want, db, qr = r
            f, gdb, gqr = gm.sw_full_cs(gls, goff, glen, rd, rlen, initbp, thresh, (ax, ay, alen, awidth), revcmpl=bool(rv), local=kind == "L", is_rna=True)
            if want[0] == 0:
                assert f["score"] == 0, (f, want)
            else:
                got = [f[k] for k in ("score", "read_start", "rmapped", "genome_start", "gmapped", "matches", "mismatches", "insertions", "deletions", "crossovers")]
                assert got == want and gdb.encode() == db and gqr.encode() == qr, (got, want, gdb, db, gqr, qr)
        nr += 1
    assert nr >= 1500
    # per-position crossover scores (crossover_score[]: what gmapper passes for every read with quality values, ref: mapping.c:375-379, sw-full-cs.c:312-322), global and local mode
    nx = ny = 0
    for r in oa.load_kat_cs("sw_kat_cs_xover.txt.gz"):
        kind, (goff, glen, rlen, initbp, ax, ay, alen, awidth, rv, thresh), gls, rd, want, db, qr, xs = r
        f, gdb, gqr = gm.sw_full_cs(gls, goff, glen, rd, rlen, initbp, thresh, (ax, ay, alen, awidth), revcmpl=bool(rv), local=(kind == "Y"), xover=xs)
        if want[0] == 0:
            assert f["score"] == 0, (f, want)
        else:
            got = [f[k] for k in ("score", "read_start", "rmapped", "genome_start", "gmapped", "matches", "mismatches", "insertions", "deletions", "crossovers")]
            assert got == want and gdb.encode() == db and gqr.encode() == qr, (kind, got, want, gdb, db, gqr, qr)
        nx += kind == "X"; ny += kind == "Y"
    assert nx >= 1000 and ny >= 500
    # a score the device cannot hold (8 bits a position) is refused loudly, not answered with "no alignment": the reason is in gm_last_error()
    kind, (goff, glen, rlen, initbp, ax, ay, alen, awidth, rv, thresh), gls, rd, want, db, qr, xs = oa.load_kat_cs("sw_kat_cs_xover.txt.gz")[0]
    bad = xs.copy(); bad[0] = -1000
    f, _, _ = gm.sw_full_cs(gls, goff, glen, rd, rlen, initbp, thresh, (ax, ay, alen, awidth), xover=bad)
    assert f["score"] == 0 and b"crossover score outside" in gm.lib().gm_last_error()


CS_GOLDEN = ["cfg4s_50col_2Mbp", "stress_cs_60col_unal"]


@pytest.mark.parametrize("name", CS_GOLDEN)
def test_colour_space_sam_matches_reference_golden(gm, name):
    """the colour-space read path (colour index, first-colour skip, CS filter on the input strand, sw_full_cs, post_sw,
    CS SAM fields) -> SAM byte-identical to the reference's gmapper-cs"""
    contigs, reads, sam = oa.load_golden(name)
    p = gm.default_params_cs()
    p.sam_unaligned = 1 if name.endswith("_unal") else 0
    ix = gm.Index(contigs, params=p)
    s = gm.Session(ix, params=p, max_batch_reads=1024)
    got = oa.sam_header(contigs) + s.map_reads_cs(reads)
    st = s.stats
    s.close(); ix.close()
    assert got == sam, (_first_diff(got, sam), st)


@pytest.mark.parametrize("env", [{"GM_SLAB_BITS": "18"}, {"GM_SLAB_BITS": "18", "GM_K1_V4": "1"}, {"GM_NO_BUCKETS": "1"}, {"GM_SLAB_BITS": "18", "GM_K1_V2": "1"},
                                 {"GM_SCAP": "256", "GM_SCAP2": "64"}, {"GM_SLAB_BITS": "18", "GM_K1_V5": "1"}, {"GM_SLAB_BITS": "18", "GM_K1_V5": "1", "GM_K5_LSW": "12"},
                                 {"GM_POST_SW_HOST": "1"},       # post_sw by the host routine instead of k_post_sw_cs
                                 {"GM_P1_EARLY": "0"},           # pass 1 without the early stop
                                 {"GM_P2_G": "16"},              # pass 2 with four windows a wave and int carry rows (the default here: eight, int16_t)
                                 {"GM_P2_G4": "0"}],             # pass 2 with one window a wave
                         ids=lambda e: ",".join("%s=%s" % kv for kv in e.items()))
def test_colour_space_kernel_variants(gm, env):
    """every lookup kernel skips the first colour and reads strand 1 the colour-space way"""
    contigs, reads, sam = oa.load_golden("cfg4s_50col_2Mbp")
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        p = gm.default_params_cs()
        ix = gm.Index(contigs, params=p)
        s = gm.Session(ix, params=p, max_batch_reads=4096)
        got = oa.sam_header(contigs) + s.map_reads_cs(reads)
        st = s.stats
        s.close(); ix.close()
    finally:
        for k, v in old.items():
            if v is None: os.environ.pop(k, None)
            else: os.environ[k] = v
    assert got == sam, (_first_diff(got, sam), st)


def test_colour_space_longer_reads_vs_oracle(gm, oracle_lib):
    """100-colour reads (two row stripes in sw_full_cs, windows of 140) with skipped cycles, against the CPU restatement"""
    from shrimp_amd import synth
    contigs = synth.make_genome([300_000, 150_000], 21)
    reads, _ = synth.make_cs_reads(contigs, 1500, 100, 22, p_col=0.05, p_dot=0.003)
    o = oa.Session(contigs, opts="colour=1"); o.set(True, True)
    want = o.map_sam(reads, nthreads=4); o.close()
    p = gm.default_params_cs(); p.sam_unaligned = 1
    ix = gm.Index(contigs, params=p)
    s = gm.Session(ix, params=p, max_batch_reads=512)
    got = s.map_reads_cs(reads)
    st = s.stats
    s.close(); ix.close()
    assert got == want, (_first_diff(got, want), st)
    assert st["reads_matched"] > 1300


def test_colour_space_api_misuse_is_refused(gm):
    from shrimp_amd import synth
    contigs = synth.make_genome([50_000], 3)
    ix = gm.Index(contigs); s = gm.Session(ix)
    with pytest.raises(gm.GmError):
        s.map_reads_cs(np.zeros((4, 51), dtype=np.uint8))        # letter-space session
    s.close(); ix.close()
    p = gm.default_params_cs()
    ix = gm.Index(contigs, params=p); s = gm.Session(ix, params=p)
    with pytest.raises(gm.GmError):
        s.map_reads(np.zeros((4, 50), dtype=np.uint8))           # colour-space session needs primer letters
    bad = np.zeros((4, 51), dtype=np.uint8); bad[2, 0] = 15
    with pytest.raises(gm.GmError):
        s.map_reads_cs(bad)                                      # primer letter must be A/C/G/T
    with pytest.raises(gm.GmError):
        gm.Session(ix, params=gm.default_params())               # session / index mode mismatch
    s.close(); ix.close()


def test_colour_space_index_files_of_the_reference(gm, tmp_path):
    """gmapper-cs -S files: loaded here they give the SAM of the reference's -L run; saved here they are the same bytes"""
    import gzip
    d, contigs, names, reads, seeds, sam = _idxfix("idxfix_cs")
    p = gm.default_params_cs()
    ix = gm.Index.load(os.path.join(d, "idx"), params=p)
    s = gm.Session(ix, params=p, max_batch_reads=1024)
    got = oa.sam_header(contigs, names) + s.map_reads_cs(reads)
    s.close(); ix.close()
    assert got == sam, _first_diff(got, sam)
    ix = gm.Index(contigs, names=names, seeds=seeds, params=p)
    ix.save(str(tmp_path / "mine"))
    ix.close()
    for suffix in (".genome", ".seed.0", ".seed.1"):
        with gzip.open(os.path.join(d, "idx" + suffix), "rb") as f: want = f.read()
        with gzip.open(str(tmp_path / ("mine" + suffix)), "rb") as f: got = f.read()
        assert got == want, (suffix, len(got), len(want))
    with pytest.raises(gm.GmError):
        gm.Index.load(os.path.join(d, "idx"))               # a colour-space index is refused by a letter-space load


def test_local_mode_long_reads_and_band_escape_vs_oracle(gm, oracle_lib):
    """--local on 250 bp reads (four row stripes in the full SW) with indel-rich ends, so that some best local alignments leave
    the anchor band and take the threshold-band second run (ref: sw-full-ls.c:395-398); against the CPU restatement"""
    from shrimp_amd import synth
    contigs = synth.make_genome([400_000, 200_000], 31)
    reads, _ = synth.make_reads(contigs, 1200, 250, 32, p_sub=0.04, p_ins=0.01, p_del=0.01)
    rng = np.random.default_rng(33)
    reads[:, :40] = np.where(rng.random((reads.shape[0], 40)) < 0.5, rng.integers(0, 4, (reads.shape[0], 40)), reads[:, :40]).astype(np.uint8)   # ragged heads: clipped
    o = oa.Session(contigs, opts="local=1"); o.set(True, True)
    want = o.map_sam(reads, nthreads=4); o.close()
    p = gm.default_params(); p.local_alignment = 1; p.sam_unaligned = 1
    ix = gm.Index(contigs, params=p)
    s = gm.Session(ix, params=p, max_batch_reads=512)
    got = s.map_reads(reads)
    st = s.stats
    s.close(); ix.close()
    assert got == want, (_first_diff(got, want), st)
    assert b"S" in got.split(b"\t")[5] or got.count(b"S\t") > 100      # soft clips are there


def test_hashed_seed_index_files_of_the_reference(gm, tmp_path):
    """-H index files (4^12 lists per seed, weight-13 and weight-16 seeds): loaded here they give the SAM of the reference's -L run;
    saved here they are the same bytes; and the mode flags are checked on load"""
    import gzip
    d, contigs, names, reads, seeds, sam = _idxfix("idxfix_h")
    p = gm.default_params(); p.hash_seeds = 1
    ix = gm.Index.load(os.path.join(d, "idx"), params=p)
    s = gm.Session(ix, params=p, max_batch_reads=1024)
    got = oa.sam_header(contigs, names) + s.map_reads(reads)
    s.close(); ix.close()
    assert got == sam, _first_diff(got, sam)
    ix = gm.Index(contigs, names=names, seeds=seeds, params=p)
    ix.save(str(tmp_path / "mine"))
    ix.close()
    for suffix in (".genome", ".seed.0", ".seed.1"):
        with gzip.open(os.path.join(d, "idx" + suffix), "rb") as f: want = f.read()
        with gzip.open(str(tmp_path / ("mine" + suffix)), "rb") as f: got = f.read()
        assert got == want, (suffix, len(got), len(want))
    with pytest.raises(gm.GmError):
        gm.Index.load(os.path.join(d, "idx"))                       # a hashed index is refused by a plain load
    with pytest.raises(gm.GmError):
        gm.Index(contigs, names=names, seeds=seeds)                 # a weight-16 seed needs -H


@pytest.mark.parametrize("tag", ["fq33", "fq64"])
def test_fastq_quals_match_reference_golden(gm, tag):
    """gm_map_reads_fastq: QUAL column as the reference prints it for FASTQ input (reversed / re-based for mapped reads, verbatim for unmapped)"""
    from tests.test_oracle import _fastq_case
    contigs, reads, quals, delta, sam = _fastq_case(tag)
    p = gm.default_params(); p.sam_unaligned = 1
    ix = gm.Index(contigs, params=p); s = gm.Session(ix, params=p, max_batch_reads=256)
    got = oa.sam_header(contigs) + s.map_reads_fastq(reads, quals, delta)
    with pytest.raises(gm.GmError):
        s.map_reads_fastq(reads[:2], [quals[0], quals[1][:-1]], delta)          # QUAL length must equal the read length
    s.close(); ix.close()
    assert got == sam, _first_diff(got, sam)


def test_colour_space_fastq_local_matches_reference_golden(gm):
    """csfastq with --local: per-position crossover scores in the local mode of sw_full_cs (out-of-band cells carry their ROW's crossover score, ref: sw-full-cs.c:312-322),
    no post_sw, QUAL '*', CQ:Z -- byte-identical to gmapper-cs --local on the csfastq file"""
    import gzip
    from tests.test_oracle import _cs_fastq_case
    contigs, reads, quals, delta, _ = _cs_fastq_case()
    with gzip.open(os.path.join(oa.ROOT, "tests", "golden", "cfg4s_50col_fq@cs_fq_local.sam.gz"), "rb") as f: want = f.read()
    p = gm.default_params_cs(); p.sam_unaligned = 1; p.local_alignment = 1
    ix = gm.Index(contigs, params=p); s = gm.Session(ix, params=p, max_batch_reads=512)
    got = oa.sam_header(contigs) + s.map_reads_cs_fastq(reads, quals, delta)
    st = s.stats
    s.close(); ix.close()
    assert got == want, (_first_diff(got, want), st)


def test_colour_space_fastq_matches_reference_golden(gm, monkeypatch):
    """gm_map_reads_cs_fastq: per-position crossover scores on the device, post_sw with per-colour error rates, QUAL from post_sw,
    CQ:Z -- byte-identical to gmapper-cs on a csfastq file.  post_sw runs on the device for these reads too (round 3: error rates from the host's table,
    base qualities back): a huge guard tolerance sends its results through the host redo (the counter shows the device path was the one that ran)."""
    from tests.test_oracle import _cs_fastq_case
    contigs, reads, quals, delta, sam = _cs_fastq_case()
    p = gm.default_params_cs(); p.sam_unaligned = 1
    ix = gm.Index(contigs, params=p); s = gm.Session(ix, params=p, max_batch_reads=512)
    monkeypatch.setenv("GM_POST_GUARD_TOL", "0.49")
    got_redo = oa.sam_header(contigs) + s.map_reads_cs_fastq(reads, quals, delta)
    st_redo = s.stats
    monkeypatch.delenv("GM_POST_GUARD_TOL")
    assert got_redo == sam, (_first_diff(got_redo, sam), st_redo)
    assert st_redo["post_sw_host_redo"] > 0.5 * st_redo["full_calls"] > 0, st_redo
    got = oa.sam_header(contigs) + s.map_reads_cs_fastq(reads, quals, delta)
    st = s.stats
    assert st["post_sw_host_redo"] < 0.01 * st["full_calls"], st
    plain = s.map_reads_cs(reads[:200])                    # the same session without QVs afterwards: global crossover score again
    s.close(); ix.close()
    assert got == sam, (_first_diff(got, sam), st)
    _, _, want_plain = oa.load_golden("cfg4s_50col_2Mbp")
    body = b"".join(l + b"\n" for l in want_plain.split(b"\n") if l and not l.startswith(b"@"))
    plain = b"".join(l + b"\n" for l in plain.split(b"\n") if l and l.split(b"\t")[1] != b"4")      # that golden was made without --sam-unaligned
    assert plain == b"".join(l + b"\n" for l in body.split(b"\n") if l and int(l.split(b"\t")[0][1:]) < 200)


def test_sw_full_ls_local_mode_known_answers(gm):
    """S2 seam with local_alignment = 1: anchor box (and the threshold-band second run when the filter's alignment leaves it) and
    anchors == NULL, against the reference's own sw_full_ls"""
    gm.sw_full_ls_setup(1400, 1000, -33, -7, -33, -3, 10, -15, True, 8)
    n = 0
    for (goff, glen, rlen, ax, ay, alen, aw, rv, no_anchor, thresh, sv), g, r, want, edb, eqr in oa.load_kat_local():
        f, db, qr = gm.sw_full_ls(g, goff, glen, r, rlen, None if no_anchor else (ax, ay, alen, aw), revcmpl=bool(rv), threshscore=thresh, maxscore=sv,
                                  local_alignment=True)
        got = [f[k] for k in ("score", "read_start", "rmapped", "genome_start", "gmapped", "matches", "mismatches", "insertions", "deletions")]
        assert got == want and db == edb and qr == eqr, (goff, glen, rlen, no_anchor, got, want)
        n += 1
    assert n >= 500


def test_paired_fastq_matches_reference_golden(gm):
    """gm_map_pairs_fastq: QUAL strings in paired and half-paired records as the reference prints them"""
    from tests.test_oracle import _paired_fastq_case
    g, q1, q2, delta, sam = _paired_fastq_case()
    p = gm.default_params(); p.sam_unaligned = 1
    ix = gm.Index(g["contigs"], names=g["contig_names"], params=p)
    s = gm.Session(ix, params=p, max_batch_reads=256)
    got = oa.sam_header(g["contigs"], g["contig_names"]) + s.map_pairs_fastq(g["m1"], g["m2"], q1, q2, delta, g["names1"], g["names2"], mode=g["mode"],
                                                                             min_insert=g["ins"][0], max_insert=g["ins"][1])
    s.close(); ix.close()
    assert got == sam, _first_diff(got, sam)


def test_long_reads_on_many_slabs_vs_oracle(gm, oracle_lib):
    """700 bp reads on an index cut into 23 slabs: the lane-group kernel's per-slab window maps do not fit the LDS, the lane-per-list kernel
    takes over; three row stripes in the vector filter, eleven in the full SW"""
    from shrimp_amd import synth
    contigs = synth.make_genome([2_000_000, 900_000], 41)
    reads, _ = synth.make_reads(contigs, 300, 700, 42, p_sub=0.03, p_ins=0.003, p_del=0.003)
    o = oa.Session(contigs); o.set(True, True)
    want = o.map_sam(reads, nthreads=4); o.close()
    old = os.environ.get("GM_SLAB_BITS"); os.environ["GM_SLAB_BITS"] = "17"
    try:
        p = gm.default_params(); p.sam_unaligned = 1
        ix = gm.Index(contigs, params=p)
        s = gm.Session(ix, params=p, max_batch_reads=128)
        got = s.map_reads(reads)
        kern = gm.lib().gm_last_lookup_kernel().decode()
        s.close(); ix.close()
    finally:
        if old is None: os.environ.pop("GM_SLAB_BITS", None)
        else: os.environ["GM_SLAB_BITS"] = old
    assert got == want, _first_diff(got, want)
    assert kern == "k_lookup", kern


def test_longest_default_read_length_vs_oracle(gm, oracle_lib):
    """1000 bp reads (the reference's default --longest-read): eight row stripes in the vector filter, sixteen in the full SW, 1400 bp windows"""
    from shrimp_amd import synth
    contigs = synth.make_genome([1_500_000], 51)
    reads, _ = synth.make_reads(contigs, 60, 1000, 52, p_sub=0.03, p_ins=0.003, p_del=0.003)
    o = oa.Session(contigs); o.set(True, True)
    want = o.map_sam(reads, nthreads=4); o.close()
    p = gm.default_params(); p.sam_unaligned = 1
    ix = gm.Index(contigs, params=p)
    s = gm.Session(ix, params=p, max_batch_reads=64)
    got = s.map_reads(reads)
    with pytest.raises(gm.GmError):
        s.map_reads(np.zeros((2, 1001), dtype=np.uint8))          # beyond longest_read_len: refused, as the reference skips such reads
    s.close(); ix.close()
    assert got == want, _first_diff(got, want)


def test_device_resident_path_matches_the_host_buffer_path(gm):
    """gm_map_reads_device (reads already in HBM, the path bench.py times) against gm_map_reads on the same reads: SAM bytes with emit_sam = 1,
    the same alignment statistics with emit_sam = 0."""
    import torch
    from shrimp_amd import synth
    contigs, reads, sam = oa.load_golden("cfg2s_100bp_2Mbp")
    ix = gm.Index(contigs); s = gm.Session(ix, max_batch_reads=2048)
    want = s.map_reads(reads); st_host = dict(s.stats)
    dev = torch.from_numpy(synth.pack_reads(reads).view(np.int32)).to("cuda:0")
    got = s.map_device(dev.data_ptr(), reads.shape[0], reads.shape[1], emit_sam=True)
    st_dev = dict(s.stats)
    n0 = s.map_device(dev.data_ptr(), reads.shape[0], reads.shape[1], emit_sam=False, return_bytes=False)
    st_nosam = dict(s.stats)
    s.close(); ix.close()
    assert got == want and oa.sam_header(contigs) + got == sam
    keys = ("reads", "reads_matched", "lookups", "list_entries", "survivors", "windows", "vec_calls", "full_calls")
    assert [st_dev[k] for k in keys] == [st_host[k] for k in keys] == [st_nosam[k] for k in keys]
    assert n0 == 0 and st_nosam["sam_records"] == st_host["sam_records"]


def _write_fasta(path, names, seqs):
    T = np.frombuffer(b"ACGTUMRWSYKVHDBN", dtype=np.uint8)
    with open(path, "wb") as f:
        for nm, sq in zip(names, seqs):
            f.write(b">" + nm + b"\n")
            t = T[sq]
            for k in range(0, len(t), 70):
                f.write(t[k:k + 70].tobytes() + b"\n")


def _seam_driver(tmp_path):
    """tests/seam_driver.cpp compiled here (g++) against libgmapper_hip.so: it calls the seams by the reference's C++-linkage (mangled) names"""
    import subprocess
    from shrimp_amd import gmapper
    exe = str(tmp_path / "seam_driver")
    libdir = os.path.dirname(gmapper.LIB_PATH)
    r = subprocess.run(["g++", "-O1", "-std=c++17", "-o", exe, os.path.join(oa.ROOT, "tests", "seam_driver.cpp"), "-L" + libdir, "-lgmapper_hip", "-Wl,-rpath," + libdir,
                        "-Wl,-rpath-link,/opt/rocm/lib"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    return exe


def _hexw(a): return ",".join("%x" % int(x) for x in a)


def test_mangled_seams_on_every_known_answer(gm, tmp_path):
    """Drop-in at S1-S3 through the names the reference's objects reference (_Z9sw_vectorPjiiS_iS_ib, _Z10sw_gaplessPjiS_iiiS_ib, _Z10sw_full_lsPji...,
    _Z10sw_full_csPji..., _Z7post_swPjiPcP15sw_full_results; ref: sw-vector.h:3-6, sw-gapless.h:11-14, sw-full-ls.h:9-13, sw-full-cs.h:7-11, sw-post.h:8-12):
    a C++ program of our own (tests/seam_driver.cpp), linked against the library, replays EVERY known-answer record the reference's functions produced --
    1 500 sw_vector + 700 colour-space sw_vector, 2 400 sw_gapless (letter and colour space), 2 990 sw_full_ls, 1 400 sw_full_cs + 800 in local mode and 1 833 post_sw
    (with and without quality values; posterior compared to the last bit, %a) -- on the GPU."""
    import gzip, subprocess
    exe = _seam_driver(tmp_path)
    req, want = [], []
    req += ["setup_v 0 -15", "setup_f_ls"]
    nV = nF = 0
    for rec in oa.load_kat():
        if rec[0] == "V":
            _, goff, glen, rlen, g, r, score = rec
            req.append("V %d %d %d %s %s" % (goff, glen, rlen, _hexw(g), _hexw(r))); want.append("V %d" % score); nV += 1
        else:
            _, goff, glen, rlen, ax, ay, alen, aw, rv, g, r, exp, edb, eqr = rec
            req.append("F %d %d %d %d %d %d %d %d %s %s" % (goff, glen, rlen, ax, ay, alen, aw, rv, _hexw(g), _hexw(r)))
            want.append("F " + " ".join(str(x) for x in exp) + " %s %s" % (edb, eqr)); nF += 1
    req += ["setup_v 1 -10", "setup_f_cs"]
    cs = oa.load_kat_cs(); srecs = [r for r in cs if r[0] == "S"]
    for r in cs:
        if r[0] == "C":
            _, goff, glen, rlen, initbp, gcs, gls, rd, score = r
            req.append("C %d %d %d %d %s %s %s" % (goff, glen, rlen, initbp, _hexw(gcs), _hexw(gls), _hexw(rd))); want.append("C %d" % score)
    def s_args(r):
        _, (goff, glen, rlen, initbp, ax, ay, alen, awidth, rv, thresh), gls, rd, w, db, qr = r
        return "%d %d %d %d %d %d %d %d %d %d %s %s" % (goff, glen, rlen, initbp, ax, ay, alen, awidth, rv, thresh, _hexw(gls), _hexw(rd))
    for r in srecs:
        w, db, qr = r[4], r[5], r[6]
        req.append("S " + s_args(r))
        want.append("S " + " ".join(str(x) for x in w) + " %s %s" % ((db or b"-").decode(), (qr or b"-").decode()) if w[0] != 0 else None)      # score 0: nothing else is defined
    lrecs = oa.load_kat_cs("sw_kat_cs_local.txt.gz")                # sw_full_cs with local_alignment = true
    for r in lrecs:
        w, db, qr = r[4], r[5], r[6]
        req.append("L " + s_args(r))
        want.append("L " + " ".join(str(x) for x in w) + " %s %s" % ((db or b"-").decode(), (qr or b"-").decode()) if w[0] != 0 else None)
    xrecs = oa.load_kat_cs("sw_kat_cs_xover.txt.gz")                # sw_full_cs with crossover_score[] (X: global, Y: local)
    for r in xrecs:
        w, db, qr = r[4], r[5], r[6]
        req.append(r[0] + " " + s_args(r[:7]) + " " + ",".join(str(int(x)) for x in r[7]))
        want.append(r[0] + " " + " ".join(str(x) for x in w) + " %s %s" % ((db or b"-").decode(), (qr or b"-").decode()) if w[0] != 0 else None)
    req.append("rna 1")                                             # is_rna = true on RNA genomes (sw_kat_cs_rna: the reference's own answers, global and local)
    for r in oa.load_kat_cs("sw_kat_cs_rna.txt.gz"):
        if r[0] == "C":
            _, goff, glen, rlen, initbp, gcs, gls, rd, score = r
            req.append("C %d %d %d %d %s %s %s" % (goff, glen, rlen, initbp, _hexw(gcs), _hexw(gls), _hexw(rd))); want.append("C %d" % score)
        else:
            w, db, qr = r[4], r[5], r[6]
            req.append(r[0] + " " + s_args(r))
            want.append(r[0] + " " + " ".join(str(x) for x in w) + " %s %s" % ((db or b"-").decode(), (qr or b"-").decode()) if w[0] != 0 else None)
    req.append("rna 0")
    with gzip.open(os.path.join(oa.ROOT, "tests", "golden", "sw_kat_post.txt.gz"), "rt") as f: post = [l.split() for l in f if l.strip()]
    K = [t for t in post if t[0] == "K"][0][1:]
    nP = 0
    for useq in (0, 1):
        req.append("setup_p %d %s" % (useq, " ".join(K)))
        for t in post:
            if t[0] != "P" or int(t[2]) != useq: continue
            req.append("P %s %s" % (t[3], s_args(srecs[int(t[1])]))); want.append("P " + " ".join(t[4:])); nP += 1
    nG = 0
    with gzip.open(os.path.join(oa.ROOT, "tests", "golden", "sw_kat_gapless.txt.gz"), "rt") as f: gl = [l.split() for l in f if l.startswith("G ")]
    for half, mm in ((0, -15), (1, -24)):
        req.append("setup_g 10 %d" % mm)
        for t in gl:
            if (t[5] != "-1") != bool(half): continue
            req.append(" ".join(t[:-1])); want.append("G " + t[-1]); nG += 1
    with gzip.open(os.path.join(oa.ROOT, "tests", "golden", "sw_kat_gapless_rna.txt.gz"), "rt") as f: glr = [l.split() for l in f if l.startswith("G ")]
    req.append("rna 1")                                             # sw_gapless with is_rna = true on RNA genomes (the forced first colour, ref: sw-gapless.c:84)
    for t in glr: req.append(" ".join(t[:-1])); want.append("G " + t[-1]); nG += 1
    req.append("rna 0")
    req.append("stats")
    assert nV >= 1500 and nF >= 2990 and len(srecs) >= 1400 and nP >= 1800 and nG >= 3000
    p = subprocess.run([exe], input=("\n".join(req) + "\n").encode(), capture_output=True, timeout=1500)
    assert p.returncode == 0, p.stderr[-2000:]
    got = p.stdout.decode().split("\n")
    assert got[-1] == "" and len(got) == len(want) + 2, (len(got), len(want))
    bad = []
    for i, w in enumerate(want):
        if w is None:
            if not got[i][:4] in ("S 0 ", "L 0 ", "X 0 ", "Y 0 "): bad.append((i, got[i][:200], "S 0 ... / L 0 ... / X 0 ... / Y 0 ..."))
        elif got[i] != w: bad.append((i, got[i][:300], w[:300]))
    assert not bad, (len(bad), bad[:5])
    st = [int(x) for x in got[len(want)].split()[1:]]
    # the *_stats entries (ref: gmapper.c:734-745 reads them): every set-up above resets its counters, so each shows the calls since its last set-up
    nq = sum(1 for t in post if t[0] == "P" and t[2] == "1")
    nrc = sum(1 for r in oa.load_kat_cs("sw_kat_cs_rna.txt.gz") if r[0] == "C"); nrs = sum(1 for r in oa.load_kat_cs("sw_kat_cs_rna.txt.gz") if r[0] != "C")
    assert st == [700 + nrc, (nG - len(glr)) // 2 + len(glr), nF, len(srecs) + len(lrecs) + len(xrecs) + nrs + nP, nq], st


def test_full_size_genome_vs_oracle(gm, oracle_lib):
    """BASELINE configs[2..4] at full genome size: the 24-contig 3.0 Gbp genome (global 32-bit coordinates beyond 2^31, six slabs, the lookup kernel
    the benchmark really runs), 20 000 letter-space reads, 2 000 2x150 bp pairs and 5 000 50-colour reads against the CPU oracle."""
    from shrimp_amd import synth
    contigs = synth.make_genome(synth.contig_lengths("cfg3", 1.0), 3)
    offs = np.cumsum([0] + [len(c) for c in contigs])
    assert offs[-1] > 2**31 and offs[-1] < 2**32
    high = {b"contig%d" % (i + 1) for i in range(len(contigs)) if offs[i] >= 2**31}
    reads, _ = synth.make_reads(contigs, 20000, 100, 31)
    pr, _ = synth.make_pairs(contigs, 2000, 150, 35)
    m1, m2 = pr[0::2], pr[1::2]
    o = oa.Session(contigs)
    want = o.map_sam(reads, nthreads=16)
    o.set_pairing("opp-in", 100, 600)
    want_p = o.map_pairs_sam(m1, m2, nthreads=16)
    o.set_half_paired(False)
    want_n = o.map_pairs_sam(m1[:500], m2[:500], nthreads=16)                               # --no-half-paired: mate-pair region counts at full size
    o.close()
    ix = gm.Index(contigs); s = gm.Session(ix)
    assert ix.n_slabs == 6
    got = s.map_reads(reads)
    kern = gm.lib().gm_last_lookup_kernel().decode()
    got_p = s.map_pairs(m1, m2, mode="opp-in", min_insert=100, max_insert=600)
    onh = gm.PairOpts.default("opp-in", 100, 600); onh.half_paired = 0
    got_n = s.map_pairs(m1[:500], m2[:500], opts=onh)
    s.close(); ix.close()
    assert kern == "k_lookup_v5", kern
    assert got == want, _first_diff(got, want)
    assert sum(1 for l in got.split(b"\n") if l and l.split(b"\t")[2] in high) > 1000       # hits at global positions >= 2^31
    assert got_p == want_p, _first_diff(got_p, want_p)
    assert got_n == want_n, _first_diff(got_n, want_n)
    # a region geometry other than the default at full size (--region-bits 10 --region-overlap 30: twice the regions, the fast lookup kernel's tables twice as loaded)
    o = oa.Session(contigs, opts="region-bits=10;region-overlap=30")
    want_r = o.map_sam(reads[:4000], nthreads=16); o.close()
    pr_ = gm.default_params(); pr_.region_bits = 10; pr_.region_overlap = 30
    ix = gm.Index(contigs, params=pr_); s = gm.Session(ix, params=pr_)
    got_r = s.map_reads(reads[:4000]); kern_r = gm.lib().gm_last_lookup_kernel().decode()
    s.close(); ix.close()
    assert got_r == want_r, (kern_r, _first_diff(got_r, want_r))
    cs, _ = synth.make_cs_reads(contigs, 5000, 50, 37)
    o = oa.Session(contigs, opts="colour=1")
    want_c = o.map_sam(cs, nthreads=16); o.close()
    p = gm.default_params_cs()
    ix = gm.Index(contigs, params=p); s = gm.Session(ix, params=p)
    got_c = s.map_reads_cs(cs)
    s.close(); ix.close()
    assert got_c == want_c, _first_diff(got_c, want_c)
    assert sum(1 for l in got_c.split(b"\n") if l and l.split(b"\t")[2] in high) > 200


@pytest.mark.parametrize("mode", ["ls", "cs"])
def test_text_input_matches_reference_golden(gm, mode):
    """A22: reads handed over as the file's characters (gm_map_reads_text packs them with gm_sequence_to_bitfield): lower case, X / . / U, ambiguity codes;
    csfasta with '.', '4', 'N' and a lower-case primer -- against the reference's SAM with --sam-unaligned, which prints SEQ of unaligned reads and CS:Z
    from the file's text (ref: gmapper/output.c:326-351,451,727)."""
    import gzip
    d = os.path.join(oa.ROOT, "tests", "golden")
    with gzip.open(os.path.join(d, "text_%s_reads.txt.gz" % mode), "rb") as f: lines = [l for l in f.read().split(b"\n") if l]
    with gzip.open(os.path.join(d, "text_%s_unal.sam.gz" % mode), "rb") as f: sam = f.read()
    contigs, _, _ = oa.load_golden("stress_100bp_unal" if mode == "ls" else "stress_cs_60col_unal")
    p = gm.default_params() if mode == "ls" else gm.default_params_cs()
    p.sam_unaligned = 1
    ix = gm.Index(contigs, params=p); s = gm.Session(ix, params=p, max_batch_reads=512)
    got = oa.sam_header(contigs) + s.map_reads_text(lines)
    s.close(); ix.close()
    assert got == sam, _first_diff(got, sam)
    assert any(ch in l.split(b"\t")[9] for l in sam.split(b"\n") if l and not l.startswith(b"@") for ch in (b"X", b"U", b".")) or mode == "cs"


@pytest.mark.parametrize("tag", sorted(oa.PAIR_MODE_CASES))
def test_paired_match_modes_match_reference_golden(gm, oracle_lib, tag):
    """gm_pair_opts_t.match_mode 3 and 2 (gmapper -p <mode> -n 3 / -n 2), with and without half-paired.  Mode 3: the lookup lists each read-strand's regions marked
    twice, then keeps the entries of such regions and of regions the mate reaches (GmMpDev; rule 3 in k_mp_filter without half-paired), the window kernel runs hit-list
    mode 3 (heavy_mp, mapping.c:1080-1093,1153-1157), pass 1 takes one match.  Mode 2: every list entry, a window per anchor.  SAM == the reference's; the stage
    check is the collapsed-anchor and window counts against the oracle's."""
    base, oopts, fields = oa.PAIR_MODE_CASES[tag]
    g = oa.load_golden_pairs(base); want = oa.load_option_sam(base, tag)
    o = oa.Session(g["contigs"], g["contig_names"], opts=oopts); o.set_pairing(g["mode"], *g["ins"])
    o.map_pairs_sam(g["m1"], g["m2"], g["names1"], g["names2"], nthreads=8)
    want_anchors, want_windows = o.last_pair_counts(); o.close()
    p = gm.default_params()
    opts = gm.PairOpts.default(g["mode"], g["ins"][0], g["ins"][1])
    for k, v in fields.items():
        if k.startswith("param_"): setattr(p, k[6:], v)
        else: setattr(opts, k, v)
    ix = gm.Index(g["contigs"], names=g["contig_names"], params=p); s = gm.Session(ix, params=p, max_batch_reads=4096)
    got = oa.sam_header(g["contigs"], g["contig_names"]) + s.map_pairs(g["m1"], g["m2"], g["names1"], g["names2"], opts=opts)
    st = s.stats
    s.close(); ix.close()
    assert got == want, (_first_diff(got, want), st)
    assert (st["anchors"], st["windows"]) == (want_anchors, want_windows), (st["anchors"], st["windows"], want_anchors, want_windows)


@pytest.mark.parametrize("tag,fields", [("pairs_n3", dict(match_mode=3)), ("pairs_n3_nhp", dict(match_mode=3, half_paired=0)), ("pairs_n2", dict(match_mode=2)),
                                         ("no_half_paired", dict(half_paired=0))])
def test_paired_match_modes_over_many_sub_batches(gm, tag, fields):
    """the same reference goldens with 256-pair sub-batches: the front of sub-batch i + 1 (its own mate-pair rows and views, per buffer pair) is queued beside the back
    of sub-batch i; heavy-tier read-strands and the exact redo of --no-half-paired fall inside the pipeline"""
    g = oa.load_golden_pairs("stress_pairs_2x100"); want = oa.load_option_sam("stress_pairs_2x100", tag)
    ix = gm.Index(g["contigs"], names=g["contig_names"]); s = gm.Session(ix, max_batch_reads=256)
    opts = gm.PairOpts.default(g["mode"], g["ins"][0], g["ins"][1])
    for k, v in fields.items(): setattr(opts, k, v)
    got = oa.sam_header(g["contigs"], g["contig_names"]) + s.map_pairs(g["m1"], g["m2"], g["names1"], g["names2"], opts=opts)
    st = s.stats
    s.close(); ix.close()
    assert got == want, (_first_diff(got, want), st)
    assert st["mp_unfiltered"] == 0


@pytest.mark.parametrize("base,tag", [("stress_pairs_2x100", "no_half_paired"), ("cfg5s_2x150_1Mbp", "cfg5_no_half_paired")])
def test_no_half_paired_mate_pair_region_counts(gm, oracle_lib, base, tag):
    """A7: gm_pair_opts_t.half_paired = 0 -- k_mp_filter applies the other mate's region counts to each mate's list entries (mapping.c:545-608,733-742)
    and the unpaired rescue is off.  SAM == the reference's `--no-half-paired` run; the stage check is the window count, which the filter changes
    (and K1b does not): it equals the oracle's with the filter, and differs from the default paired run's."""
    g = oa.load_golden_pairs(base)
    want = oa.load_option_sam(base, tag)
    o = oa.Session(g["contigs"], g["contig_names"], opts="half-paired=0")
    o.set_pairing(g["mode"], *g["ins"])
    o.map_pairs_sam(g["m1"], g["m2"], g["names1"], g["names2"], nthreads=4)
    _, want_windows = o.last_pair_counts(); o.close()
    ix = gm.Index(g["contigs"], names=g["contig_names"])
    s = gm.Session(ix, max_batch_reads=4096)
    opts = gm.PairOpts.default(g["mode"], g["ins"][0], g["ins"][1]); opts.half_paired = 0
    got = oa.sam_header(g["contigs"], g["contig_names"]) + s.map_pairs(g["m1"], g["m2"], g["names1"], g["names2"], opts=opts)
    st = s.stats
    s.map_pairs(g["m1"], g["m2"], g["names1"], g["names2"], mode=g["mode"], min_insert=g["ins"][0], max_insert=g["ins"][1])
    st_default = s.stats
    s.close(); ix.close()
    assert got == want, (_first_diff(got, want), st)
    # gm_map_stats_t.mp_unfiltered: (pair, strand) items beyond the filter's LDS tiers.  None may be left in a result: a sub-batch that has some (the repeat-rich
    # stress genome: heavy-tier read-strands) is redone through the generic kernel's mate-pair modes (rows of regions marked twice, rule 1 in the sweep), which is exact
    # for rows of any length -- the window count equals the oracle's on both genomes.
    assert st["mp_unfiltered"] == 0, st["mp_unfiltered"]
    if not base.startswith("cfg5s"): assert st["retries"] >= 1, st
    assert st["windows"] < st_default["windows"], (st["windows"], st_default["windows"])
    assert st["windows"] == want_windows, (st["windows"], want_windows)


@pytest.mark.parametrize("mode", ["opp-in", "opp-out", "col-fw", "col-bw"])
def test_colour_space_pairs_match_reference_golden(gm, mode):
    """paired colour space (gm_map_pairs_cs, pair mode opp-in): K1 / pair-up / CS filter / sw_full_cs at half the threshold / post_sw / readpair_pass2 / the unpaired
    rescue / CS SAM fields -- byte-identical to gmapper-cs -p <mode> -I 100,600 --sam-unaligned (paired, half-paired and unaligned records); in opp-out, col-fw and
    col-bw a mate is reversed: it keeps its colours and swaps its strand labels (GmIndexDev::cs_flip)"""
    g = oa.load_golden_pairs("cs_pairs_50col_" + mode)
    want = g["sam"]
    p = gm.default_params_cs(); p.sam_unaligned = 1
    ix = gm.Index(g["contigs"], names=g["contig_names"], params=p)
    s = gm.Session(ix, params=p, max_batch_reads=4096)
    got = oa.sam_header(g["contigs"], g["contig_names"]) + s.map_pairs_cs(g["m1"], g["m2"], g["names1"], g["names2"], mode=g["mode"],
                                                                           min_insert=g["ins"][0], max_insert=g["ins"][1])
    st = s.stats
    s.close(); ix.close()
    assert got == want, (_first_diff(got, want), st)


@pytest.mark.parametrize("tag", sorted(oa.CS_PAIR_OPTION_CASES))
def test_colour_space_pairs_local_match_reference_golden(gm, tag):
    """gmapper-cs -p <mode> -I 100,600 --sam-unaligned --local: the paired pipeline with sw_full_cs in local mode at half the threshold, sw_full_cs's own strings and
    counts in the output (no post_sw), no mapping qualities -- byte-identical to the reference (opp-in, and col-bw where a mate is reversed)"""
    base, _, fields = oa.CS_PAIR_OPTION_CASES[tag]
    g = oa.load_golden_pairs(base)
    want = oa.load_option_sam(base, tag)
    p = gm.default_params_cs(); p.sam_unaligned = 1
    opts = gm.PairOpts.default(g["mode"], g["ins"][0], g["ins"][1])
    for k, v in fields.items():
        if k.startswith("pair_"): setattr(opts, k[5:], v)         # gm_pair_opts_t: match_mode 3 / 2, half_paired (the -n 3 / -n 2 cases)
        else: setattr(p, k, v)
    ix = gm.Index(g["contigs"], names=g["contig_names"], params=p)
    s = gm.Session(ix, params=p, max_batch_reads=4096)
    got = oa.sam_header(g["contigs"], g["contig_names"]) + s.map_pairs_cs(g["m1"], g["m2"], g["names1"], g["names2"], opts=opts)
    st = s.stats
    s.close(); ix.close()
    assert got == want, (_first_diff(got, want), st)


@pytest.mark.parametrize("pair_mode", ["opp-in", "opp-out", "col-fw", "col-bw"])
@pytest.mark.parametrize("mm,hp", [(3, 1), (3, 0), (2, 1), (4, 0)])
def test_paired_match_modes_in_every_pair_mode_vs_oracle(gm, oracle_lib, pair_mode, mm, hp):
    """match_mode x half_paired x pair mode on the colour-space pairs of each pair mode, against the oracle (pinned to the reference on the opp-in and col-bw cases
    above): SAM and the anchor / window counts.  In col-fw / col-bw the two mates' region deltas are not each other's negation (mapping.c:2381-2384,2407-2410), which
    the flag pass of rule 3 has to respect."""
    g = oa.load_golden_pairs("cs_pairs_50col_" + pair_mode)
    o = oa.Session(g["contigs"], g["contig_names"], opts="colour=1;mp-match-mode=%d;half-paired=%d" % (mm, hp)); o.set(True, True); o.set_pairing(g["mode"], *g["ins"])
    want = oa.sam_header(g["contigs"], g["contig_names"]) + o.map_pairs_sam(g["m1"], g["m2"], g["names1"], g["names2"], nthreads=8)
    want_counts = o.last_pair_counts(); o.close()
    p = gm.default_params_cs(); p.sam_unaligned = 1
    opts = gm.PairOpts.default(g["mode"], g["ins"][0], g["ins"][1]); opts.match_mode = mm; opts.half_paired = hp
    ix = gm.Index(g["contigs"], names=g["contig_names"], params=p); s = gm.Session(ix, params=p, max_batch_reads=4096)
    got = oa.sam_header(g["contigs"], g["contig_names"]) + s.map_pairs_cs(g["m1"], g["m2"], g["names1"], g["names2"], opts=opts)
    st = s.stats
    s.close(); ix.close()
    assert got == want, (_first_diff(got, want), st)
    assert st["mp_unfiltered"] == 0 and (st["anchors"], st["windows"]) == want_counts, (st["anchors"], st["windows"], want_counts)


@pytest.mark.parametrize("paired", [False, True])
def test_genome_shards_mapped_here_then_merged_match_mergesam(gm, paired):
    """SURVEY 8(f)3 end to end: the contig groups of tests/golden/merge mapped by the library (byte-identical to the reference's per-shard SAM files),
    the two texts merged by gm_merge_sam: the reference's mergesam output on the reference's shard files."""
    import gzip, json
    M = os.path.join(os.path.dirname(__file__), "golden", "merge")
    case = "pairs_db2" if paired else "ls_db2"
    c = json.load(open(os.path.join(M, "cases.json")))[case]
    rd = lambda n: gzip.open(os.path.join(M, n), "rb").read()
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "stress_60bp.npz"))
    contigs = [z["contig%d" % i] for i in range(4)]; names = [b"contig%d" % (i + 1) for i in range(4)]
    if paired:
        g = oa.load_golden_pairs("stress_pairs_2x100"); NP = 500
    else:
        reads = z["reads"][:1400]
    texts = []
    for k, (lo, hi) in enumerate(((0, 1), (1, 4))):
        ix = gm.Index(contigs[lo:hi], names=names[lo:hi]); s = gm.Session(ix, max_batch_reads=4096)
        if paired: body = s.map_pairs(g["m1"][:NP], g["m2"][:NP], list(g["names1"][:NP]), list(g["names2"][:NP]), mode="opp-in", min_insert=100, max_insert=600)
        else: body = s.map_reads(reads)
        s.close(); ix.close()
        want = rd("%s.in%d.sam.gz" % (case, k))
        pg = [l for l in want.split(b"\n") if l.startswith(b"@PG")]
        assert body == b"".join(l + b"\n" for l in want.split(b"\n") if l and not l.startswith(b"@")), _first_diff(body, want)
        texts.append(oa.sam_header(contigs[lo:hi], names[lo:hi]) + pg[0] + b"\n" + body)      # the shard's @PG line as the reference's run wrote it
    reads_text = rd(case + ".reads.gz")
    for st in ("default", "single_best_all", "strata", "all_contigs"):
        d = c["sets"][st]
        kw = {"--strata": {"strata": 1}, "--single-best-mapping": {"single_best": 1}, "--all-contigs": {"all_contigs": 1}}
        opts = {}
        for a in d["args"]: opts.update(kw[a])
        got = gm.merge_sam(reads_text, texts, command_line=d["command_line"], threads=4, **opts)
        want = rd("%s@%s.out.gz" % (case, st))
        assert got == want, (st, _first_diff(got, want))


@pytest.mark.parametrize("case", ["ls_fa_gz", "ls_fq", "cs_fa_gz"])
def test_reads_files_match_reference_golden(gm, case, tmp_path):
    """gm_map_reads_file on the very files the reference read -- gzip, '#' comments, folded sequences, descriptions, five read lengths mixed,
    folded FASTQ, a read beyond --longest-read (dropped), csfasta -- records in the file's order, byte-identical to the reference's SAM."""
    import gzip, shutil
    G = os.path.join(os.path.dirname(__file__), "golden")
    src, want_name, base = {"ls_fa_gz": ("file_ls_mixed.fa.gz", "file_ls_mixed.sam.gz", "stress_60bp"), "ls_fq": ("file_ls_mixed.fq.gz", "file_ls_mixed_fq.sam.gz", "stress_60bp"),
                            "cs_fa_gz": ("file_cs_mixed.csfasta.gz", "file_cs_mixed.sam.gz", "stress_cs_60col_unal")}[case]
    z = np.load(os.path.join(G, base + ".npz")); contigs = [z["contig%d" % i] for i in range(sum(1 for f in z.files if f.startswith("contig")))]
    want = gzip.open(os.path.join(G, want_name), "rb").read()
    path = os.path.join(G, src)
    if case == "ls_fq":                                       # the reference read this one as plain text
        path = str(tmp_path / "r.fq"); open(path, "wb").write(gzip.open(os.path.join(G, src), "rb").read())
    p = gm.default_params_cs() if case.startswith("cs") else gm.default_params()
    p.sam_unaligned = 1
    ix = gm.Index(contigs, params=p); s = gm.Session(ix, params=p, max_batch_reads=4096)
    got = oa.sam_header(contigs) + s.map_reads_file(path, qual_delta=33 if case == "ls_fq" else None)
    st = s.stats
    s.close(); ix.close()
    assert got == want, (_first_diff(got, want), st)


FORMATS = {"fmt_shrimp": ("stress_60bp", dict(output_format=1)), "fmt_pretty_R": ("stress_60bp", dict(output_format=2, print_read_seq=1)),
           "fmt_local_pretty": ("stress_100bp_unal", dict(output_format=2, local_alignment=1)),
           "fmt_cs_shrimp_R": ("stress_cs_60col_unal", dict(output_format=1, print_read_seq=1)), "fmt_cs_pretty": ("stress_cs_60col_unal", dict(output_format=2)),
           "fmt_pairs_shrimp_R": ("stress_pairs_2x100", dict(output_format=1, print_read_seq=1)), "fmt_pairs_pretty": ("pairfix_opp-out", dict(output_format=2)),
           "fmt_pairs_colbw": ("pairfix_col-bw", dict(output_format=1)), "fmt_cs_pairs_pretty_R": ("cs_pairs_50col_col-bw", dict(output_format=2, print_read_seq=1)),
           "fmt_cs_pairs_shrimp": ("cs_pairs_50col_opp-in", dict(output_format=1))}


@pytest.mark.parametrize("tag", sorted(FORMATS))
def test_output_formats_match_reference_golden(gm, tag):
    """--shrimp-format and -P/--pretty (with -R), letter and colour space, global and local alignments: the reference's whole output"""
    import gzip
    base, fields = FORMATS[tag]
    G = os.path.join(os.path.dirname(__file__), "golden")
    z = np.load(os.path.join(G, base + ".npz")); contigs = [z["contig%d" % i] for i in range(sum(1 for f in z.files if f.startswith("contig") and f[6:].isdigit()))]
    want = gzip.open(os.path.join(G, "%s@%s.txt.gz" % (base, tag)), "rb").read()
    cs = "cs" in base
    p = gm.default_params_cs() if cs else gm.default_params()
    for k, v in fields.items(): setattr(p, k, v)
    if "mates1" in z.files:
        g = oa.load_golden_pairs(base)
        ix = gm.Index(g["contigs"], names=g["contig_names"], params=p); s = gm.Session(ix, params=p, max_batch_reads=4096)
        body = (s.map_pairs_cs if cs else s.map_pairs)(g["m1"], g["m2"], g["names1"], g["names2"], mode=g["mode"], min_insert=g["ins"][0], max_insert=g["ins"][1])
    else:
        ix = gm.Index(contigs, params=p); s = gm.Session(ix, params=p, max_batch_reads=4096)
        body = s.map_reads_cs(z["reads"]) if cs else s.map_reads(z["reads"])
    s.close(); ix.close()
    head = b"#FORMAT: readname contigname strand contigstart contigend readstart readend readlength score editstring" + (b" readsequence" if fields.get("print_read_seq") else b"") + b"\n"
    got = head + body
    assert got == want, _first_diff(got, want)


@pytest.mark.parametrize("mode", ["opp-in", "col-bw"])
def test_colour_space_fastq_pairs_match_reference_golden(gm, mode):
    """csfastq pairs (gm_map_pairs_cs_fastq): QV-dependent crossover scores and post_sw error rates for both mates, QUAL = post_sw's base qualities, CQ:Z in every
    record kind -- the reference's gmapper-cs -p <mode> on a csfastq file, opp-in and a mode that reverses a mate"""
    import gzip
    G = os.path.join(os.path.dirname(__file__), "golden")
    g = oa.load_golden_pairs("cs_pairs_50col_" + mode); zq = np.load(os.path.join(G, "cs_pairs_fq_%s.npz" % mode)); N = int(zq["n_pairs"])
    want = gzip.open(os.path.join(G, "cs_pairs_fq_%s.sam.gz" % mode), "rb").read()
    p = gm.default_params_cs(); p.sam_unaligned = 1
    ix = gm.Index(g["contigs"], names=g["contig_names"], params=p); s = gm.Session(ix, params=p, max_batch_reads=4096)
    got = oa.sam_header(g["contigs"], g["contig_names"]) + s.map_pairs_cs(g["m1"][:N], g["m2"][:N], list(g["names1"][:N]), list(g["names2"][:N]), mode=mode,
                                                                          min_insert=g["ins"][0], max_insert=g["ins"][1], quals1=zq["quals1"], quals2=zq["quals2"],
                                                                          qual_delta=int(zq["qual_delta"]))
    st = s.stats
    s.close(); ix.close()
    assert got == want, (_first_diff(got, want), st)


@pytest.mark.parametrize("mode", ["opp-in", "col-bw"])
def test_colour_space_fastq_pairs_local_match_reference_golden(gm, mode):
    """csfastq pairs with --local: per-position crossover scores in sw_full_cs's local mode for both mates, no post_sw (QUAL '*'), CQ:Z -- gmapper-cs -p <mode> --local on the csfastq
    file of the test above"""
    import gzip
    G = os.path.join(os.path.dirname(__file__), "golden")
    g = oa.load_golden_pairs("cs_pairs_50col_" + mode); zq = np.load(os.path.join(G, "cs_pairs_fq_%s.npz" % mode)); N = int(zq["n_pairs"])
    want = gzip.open(os.path.join(G, "cs_pairs_fq_%s@cs_pairs_fq_local.sam.gz" % mode), "rb").read()
    p = gm.default_params_cs(); p.sam_unaligned = 1; p.local_alignment = 1
    ix = gm.Index(g["contigs"], names=g["contig_names"], params=p); s = gm.Session(ix, params=p, max_batch_reads=4096)
    got = oa.sam_header(g["contigs"], g["contig_names"]) + s.map_pairs_cs(g["m1"][:N], g["m2"][:N], list(g["names1"][:N]), list(g["names2"][:N]), mode=mode,
                                                                          min_insert=g["ins"][0], max_insert=g["ins"][1], quals1=zq["quals1"], quals2=zq["quals2"],
                                                                          qual_delta=int(zq["qual_delta"]))
    st = s.stats
    s.close(); ix.close()
    assert got == want, (_first_diff(got, want), st)


def test_read_loop_preprocessing_matches_reference_golden(gm):
    """the file entries with the read loop's preprocessing (gm_params_t trim_front / trim_end / trim_illumina / min_avg_qv / ignore_qvs; ref: gmapper.c:262-284,427-472,
    495-521): the reference's SAM under each option set, whole and through the streaming form in chunks of 100 reads; paired: --trim-second"""
    import gzip
    from tools.make_golden import PREPROCESS_CASES
    G = os.path.join(os.path.dirname(__file__), "golden")
    z = np.load(os.path.join(G, "stress_60bp.npz")); contigs = [z["contig%d" % i] for i in range(sum(1 for f in z.files if f.startswith("contig")))]
    for tag, (src, _, fields) in PREPROCESS_CASES.items():
        p = gm.default_params(); p.sam_unaligned = 1
        for k, v in fields.items(): setattr(p, k, v)
        ix = gm.Index(contigs, params=p); s = gm.Session(ix, params=p, max_batch_reads=4096)
        want = gzip.open(os.path.join(G, tag + ".sam.gz"), "rb").read()
        got = oa.sam_header(contigs) + s.map_reads_file(os.path.join(G, src), qual_delta=64)
        assert got == want, (tag, _first_diff(got, want), s.stats)
        parts = s.map_reads_file_chunks(os.path.join(G, src), chunk_reads=100, qual_delta=64)
        assert len(parts) >= 3 and oa.sam_header(contigs) + b"".join(parts) == want, (tag, len(parts))
        s.close(); ix.close()
    g = oa.load_golden_pairs("stress_pairs_2x100")
    p = gm.default_params(); p.sam_unaligned = 1; p.trim_front = 2; p.trim_end = 4; p.trim_first = 0; p.trim_second = 1
    ix = gm.Index(g["contigs"], names=g["contig_names"], params=p); s = gm.Session(ix, params=p, max_batch_reads=4096)
    want = gzip.open(os.path.join(G, "pre_pairs_trim_second.sam.gz"), "rb").read()
    body = s.map_pairs_file(os.path.join(G, "file_pairs_1.fq.gz"), os.path.join(G, "file_pairs_2.fq.gz"), qual_delta=33, mode=g["mode"], min_insert=g["ins"][0], max_insert=g["ins"][1])
    got = oa.sam_header(g["contigs"], g["contig_names"]) + body
    assert got == want, (_first_diff(got, want), s.stats)
    parts = s.map_pairs_file_chunks(os.path.join(G, "file_pairs_1.fq.gz"), os.path.join(G, "file_pairs_2.fq.gz"), chunk_pairs=64, qual_delta=33, mode=g["mode"], min_insert=g["ins"][0], max_insert=g["ins"][1])
    assert len(parts) >= 5 and b"".join(parts) == body
    p.trim_first = 1                                           # the first mate: refused (the reference trims it after packing it)
    s2 = gm.Session(ix, params=p, max_batch_reads=4096)
    with pytest.raises(RuntimeError): s2.map_pairs_file(os.path.join(G, "file_pairs_1.fq.gz"), os.path.join(G, "file_pairs_2.fq.gz"), qual_delta=33, mode=g["mode"])
    s2.close(); s.close(); ix.close()


def test_file_entries_edge_cases(gm, tmp_path):
    """the streaming file entries at their edges: an empty file, a file whose reads are all dropped (mean quality below --min-avg-qv), a chunk size that equals the number of
    reads, a chunk of one read, an odd read at the end of an interleaved pair file (it has no mate and is ignored), a write function that asks to stop"""
    import ctypes as C
    contigs, reads, _ = oa.load_golden("stress_60bp")
    p = gm.default_params(); p.sam_unaligned = 1
    ix = gm.Index(contigs, params=p); s = gm.Session(ix, params=p, max_batch_reads=4096)
    T = np.frombuffer(b"ACGTUMRWSYKVHDBN", dtype=np.uint8)
    empty = str(tmp_path / "empty.fa"); open(empty, "wb").close()
    assert s.map_reads_file(empty) == b"" and s.map_reads_file_chunks(empty, chunk_reads=7) == []
    fq = str(tmp_path / "low.fq")
    with open(fq, "wb") as f:
        for i, r in enumerate(reads[:40]): f.write(b"@q%d\n" % i + T[r].tobytes() + b"\n+\n" + b"#" * len(r) + b"\n")       # PHRED+33 '#': quality 2
    assert s.map_reads_file(fq, qual_delta=33) == b""
    fa = str(tmp_path / "r.fa")
    with open(fa, "wb") as f:
        for i, r in enumerate(reads[:50]): f.write(b">r%d\n" % i + T[r].tobytes() + b"\n")
    whole = s.map_reads_file(fa)
    assert whole.count(b"\n") >= 50
    for chunk in (50, 1, 49, 51):
        parts = s.map_reads_file_chunks(fa, chunk_reads=chunk)
        assert b"".join(parts) == whole and len(parts) == (50 + chunk - 1) // chunk, (chunk, len(parts))
    il = str(tmp_path / "il.fa")
    with open(il, "wb") as f:
        for i, r in enumerate(reads[:21]): f.write(b">p%d/%d\n" % (i // 2, 1 + i % 2) + T[r].tobytes() + b"\n")
    body = s.map_pairs_file(il, mode="opp-in", min_insert=0, max_insert=1000)
    names = {l.split(b"\t")[0] for l in body.split(b"\n") if l}                 # (a pair prints under the common prefix of its mates' names)
    assert b"p9" in names and not any(n.startswith(b"p10") for n in names)
    # a write function that returns non-zero stops the call with an error
    WRITE = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t)
    cb = WRITE(lambda ctx, pp, n: 1)
    st = gm.MapStats()
    rc = gm.lib().gm_map_reads_file_cb(s.h, fa.encode(), -1, 64, C.c_size_t(10), cb, None, C.byref(st))
    assert rc != 0 and b"asked to stop" in gm.lib().gm_last_error()
    s.close(); ix.close()


@pytest.mark.parametrize("case", ["two_fastq_gz", "interleaved_fasta"])
def test_pair_files_match_reference_golden(gm, case, tmp_path):
    """gm_map_pairs_file: `gmapper -1 a.fq.gz -2 b.fq.gz` (PHRED+33, mates cut to a mix of lengths) and one FASTA file with the mates adjacent -- the files the
    reference read, records in file order"""
    import gzip
    G = os.path.join(os.path.dirname(__file__), "golden")
    g = oa.load_golden_pairs("stress_pairs_2x100")
    p = gm.default_params(); p.sam_unaligned = 1
    ix = gm.Index(g["contigs"], names=g["contig_names"], params=p); s = gm.Session(ix, params=p, max_batch_reads=4096)
    if case == "two_fastq_gz":
        want = gzip.open(os.path.join(G, "file_pairs_12.sam.gz"), "rb").read()
        body = s.map_pairs_file(os.path.join(G, "file_pairs_1.fq.gz"), os.path.join(G, "file_pairs_2.fq.gz"), qual_delta=33, mode=g["mode"], min_insert=g["ins"][0], max_insert=g["ins"][1])
    else:
        want = gzip.open(os.path.join(G, "file_pairs_il.sam.gz"), "rb").read()
        path = str(tmp_path / "il.fa"); open(path, "wb").write(gzip.open(os.path.join(G, "file_pairs_il.fa.gz"), "rb").read())
        body = s.map_pairs_file(path, mode=g["mode"], min_insert=g["ins"][0], max_insert=g["ins"][1])
    st = s.stats
    s.close(); ix.close()
    got = oa.sam_header(g["contigs"], g["contig_names"]) + body
    assert got == want, (_first_diff(got, want), st)


def test_csfastq_files_match_reference_golden(gm, tmp_path):
    """file input in colour space with quality values: the csfastq file of the cfg4s_50col_fq golden through gm_map_reads_file, and the csfastq pairs of
    cs_pairs_fq_opp-in (mates adjacent in one file) through gm_map_pairs_file -- the goldens of the packed-array entry points"""
    import gzip
    G = os.path.join(os.path.dirname(__file__), "golden")
    tab = np.full(16, ord("."), dtype=np.uint8); tab[:4] = np.frombuffer(b"0123", dtype=np.uint8)
    rec = lambda nm, row, q: b"@" + nm + b"\n" + b"ACGT"[row[0]:row[0] + 1] + tab[row[1:]].tobytes() + b"\n+\n" + bytes(q) + b"\n"
    # unpaired
    z = np.load(os.path.join(G, "cfg4s_50col_2Mbp.npz")); zq = np.load(os.path.join(G, "cfg4s_50col_fq.npz")); n = int(zq["n_reads"])
    contigs = [z["contig%d" % i] for i in range(sum(1 for f in z.files if f.startswith("contig") and f[6:].isdigit()))]
    path = str(tmp_path / "r.csfastq"); open(path, "wb").write(b"".join(rec(b"r%d" % i, z["reads"][i], zq["quals"][i]) for i in range(n)))
    want = gzip.open(os.path.join(G, "cfg4s_50col_fq.sam.gz"), "rb").read()
    p = gm.default_params_cs(); p.sam_unaligned = 1
    ix = gm.Index(contigs, params=p); s = gm.Session(ix, params=p, max_batch_reads=4096)
    got = oa.sam_header(contigs) + s.map_reads_file(path, qual_delta=33)
    s.close(); ix.close()
    assert got == want, _first_diff(got, want)
    # pairs
    g = oa.load_golden_pairs("cs_pairs_50col_opp-in"); zq = np.load(os.path.join(G, "cs_pairs_fq_opp-in.npz")); N = int(zq["n_pairs"])
    path = str(tmp_path / "p.csfastq")
    open(path, "wb").write(b"".join(rec(bytes(g["names1"][i]), g["m1"][i], zq["quals1"][i]) + rec(bytes(g["names2"][i]), g["m2"][i], zq["quals2"][i]) for i in range(N)))
    want = gzip.open(os.path.join(G, "cs_pairs_fq_opp-in.sam.gz"), "rb").read()
    ix = gm.Index(g["contigs"], names=g["contig_names"], params=p); s = gm.Session(ix, params=p, max_batch_reads=4096)
    got = oa.sam_header(g["contigs"], g["contig_names"]) + s.map_pairs_file(path, qual_delta=33, mode="opp-in", min_insert=g["ins"][0], max_insert=g["ins"][1])
    s.close(); ix.close()
    assert got == want, _first_diff(got, want)


def test_100mbp_genome_every_list_entry_an_anchor_vs_oracle(gm, oracle_lib):
    """-n 1 (unpaired) and -n 2 (paired) on the 100 Mbp genome of BASELINE configs[1]: ~1 600 list entries per read-strand all become anchors and windows -- rows beyond
    K2's LDS tier (the heavy tier's re-emission keeps every entry too), window lists that grow past their first capacity -- against the CPU oracle, stage counts included."""
    from shrimp_amd import synth
    contigs = synth.make_genome(synth.contig_lengths("cfg2", 1.0), 2)
    reads, _ = synth.make_reads(contigs, 3000, 100, 43)
    pr, _ = synth.make_pairs(contigs, 1500, 100, 44); m1, m2 = pr[0::2], pr[1::2]
    o = oa.Session(contigs, opts="cmw-mode=1;mp-match-mode=2")
    want = o.map_sam(reads, nthreads=16)
    o.set_pairing("opp-in", 100, 600)
    want_p = o.map_pairs_sam(m1, m2, nthreads=16); want_counts = o.last_pair_counts(); o.close()
    p = gm.default_params(); p.match_mode = 1
    ix = gm.Index(contigs, params=p); s = gm.Session(ix, params=p, max_batch_reads=4096)
    got = s.map_reads(reads); st = s.stats
    kern = gm.lib().gm_last_lookup_kernel().decode()
    opts = gm.PairOpts.default("opp-in", 100, 600); opts.match_mode = 2
    got_p = s.map_pairs(m1, m2, opts=opts); st_p = s.stats
    s.close(); ix.close()
    assert kern == "k_lookup", kern                                           # the generic slab-sweep kernel: the others all apply the two-marks rule
    assert got == want, (_first_diff(got, want), st)
    assert st["anchors"] > 1000 * len(reads), st["anchors"]
    assert got_p == want_p, (_first_diff(got_p, want_p), st_p)
    assert (st_p["anchors"], st_p["windows"]) == want_counts, (st_p["anchors"], st_p["windows"], want_counts)


def test_100mbp_genome_bucket_lookup_vs_oracle(gm, oracle_lib):
    """BASELINE configs[1] at full genome size: 4 x 25 Mbp (one slab, Poisson(6) lists -> k_lookup_bkt, the 64-byte bucket kernel: length + the first 15 positions
    per k-mer, longer lists continue in pos[]).  20 000 reads of the workload plus reads placed on lists LONGER than a bucket holds, against the CPU oracle."""
    from shrimp_amd import synth
    contigs = synth.make_genome(synth.contig_lengths("cfg2", 1.0), 2)
    assert sum(len(c) for c in contigs) == 100_000_000
    reads, _ = synth.make_reads(contigs, 20000, 100, 41)
    ix = gm.Index(contigs)
    assert ix.n_slabs == 1 and ix.has_buckets
    # lists beyond the bucket: scan list lengths of seed 0 until 40 lists of more than 15 entries are found, and cut a read out of the genome at a position of each
    offs = np.cumsum([0] + [len(c) for c in contigs]); extra = []; longest = 0
    for mapidx in range(0, 200000):
        lst = ix.get_list(0, mapidx)
        if len(lst) > 15:
            longest = max(longest, len(lst))
            g = int(lst[len(lst) // 2]); cn = int(np.searchsorted(offs, g, side="right") - 1); o = g - int(offs[cn])
            if o + 100 <= len(contigs[cn]): extra.append(contigs[cn][o:o + 100].copy())
            if len(extra) >= 40: break
    assert len(extra) >= 40 and longest > 15, (len(extra), longest)
    reads = np.concatenate([reads, np.stack(extra)])
    o = oa.Session(contigs)
    want = o.map_sam(reads, nthreads=16); o.close()
    s = gm.Session(ix)
    got = s.map_reads(reads)
    kern = gm.lib().gm_last_lookup_kernel().decode(); st = s.stats
    s.close(); ix.close()
    assert kern == "k_lookup_bkt", kern
    assert 4.0 < st["list_entries"] / st["lookups"] < 9.0, st                # Poisson(6) lists: the bucket kernel's real load
    assert got == want, _first_diff(got, want)
    # the reads cut at list positions map (each has at least one lookup that ran past its bucket)
    mapped = {l.split(b"\t")[0] for l in got.split(b"\n") if l and not l.startswith(b"@")}
    assert all(b"r%d" % (20000 + i) in mapped for i in range(len(extra)))


def test_post_sw_rounding_guard_redoes_on_the_host(gm, monkeypatch):
    """k_post_sw_cs computes the colour-space posterior with ocml's exp / log; the host redoes every result whose AS / MAPQ / Z0 / Z1 would be rounded within the
    guard's tolerance of a boundary (Finalizer::post_sw / finalize_read).  Forced here: with a tolerance of 0.49 nearly every device result is sent back through
    the host routine -- the non-destructive re-call record lets it start from sw_full_cs's own strings -- and the SAM stays the reference's; with the production
    tolerance the count is reported and the SAM is the reference's too."""
    contigs, reads, sam = oa.load_golden("cfg4s_50col_2Mbp")
    p = gm.default_params_cs()
    ix = gm.Index(contigs, params=p)
    s = gm.Session(ix, params=p)
    got0 = oa.sam_header(contigs) + s.map_reads_cs(reads); st0 = s.stats
    monkeypatch.setenv("GM_POST_GUARD_TOL", "0.49")
    got1 = oa.sam_header(contigs) + s.map_reads_cs(reads); st1 = s.stats
    monkeypatch.setenv("GM_POST_GUARD_TOL", "0.01")               # 4 % of the AS values alone
    got2 = oa.sam_header(contigs) + s.map_reads_cs(reads); st2 = s.stats
    s.close(); ix.close()
    assert got0 == sam, _first_diff(got0, sam)
    assert got1 == sam, _first_diff(got1, sam)
    assert got2 == sam, _first_diff(got2, sam)
    assert st1["post_sw_host_redo"] > 0.5 * st1["full_calls"] > 0, st1
    assert 0 < st2["post_sw_host_redo"] < st1["post_sw_host_redo"], (st2["post_sw_host_redo"], st1["post_sw_host_redo"])
    assert st0["post_sw_host_redo"] < 0.001 * st0["full_calls"], st0


def test_release_build_reproduces_the_reference_goldens():
    """`make release` (TUNING=0: no tuning / ablation knobs compiled in; the build bench.py measures) against the reference goldens: the golden tests of this
    file in a child interpreter with GM_LIB_PATH on libgmapper_hip_release.so."""
    import subprocess, sys
    rel = os.path.join(oa.ROOT, "shrimp_amd", "libgmapper_hip_release.so")
    assert os.path.exists(rel), "make -C shrimp_amd/csrc release (or __graft_entry__.build()) has not run"
    env = dict(os.environ, GM_LIB_PATH=rel)
    sel = "test_sam_matches_reference_golden or test_paired_sam_matches_reference_golden or test_colour_space_sam_matches_reference_golden or test_sw_vector_known_answers or test_colour_space_pairs_match_reference_golden"
    p = subprocess.run([sys.executable, "-m", "pytest", os.path.join(oa.ROOT, "tests", "test_gpu_parity.py"), "-q", "-x", "-m", "gpu", "-k", sel, "-p", "no:cacheprovider"],
                       capture_output=True, text=True, env=env, cwd=oa.ROOT, timeout=1500)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-2000:]
    import re
    m = re.search(r"(\d+) passed", p.stdout)
    assert m and int(m.group(1)) >= 17, p.stdout[-500:]


@pytest.mark.parametrize("tag", sorted(oa.CS_OPTION_CASES))
def test_colour_space_local_and_ungapped_match_reference_golden(gm, tag):
    """gmapper-cs --local and -U on the GPU: the colour-space cell of k_pass2_cs_g4 in local mode (floors at 0 / the crossover score with null back pointers, best cell of
    the whole band; no post_sw, MAPQ 255, no Z tags) and sw_gapless on the contig's colour translation in pass 1 (forced first colour; reverse-strand hits on the
    reverse-complement contig's colours) -- byte-identical to the reference's output"""
    base, _, fields, _ = oa.CS_OPTION_CASES[tag]
    contigs, reads, _ = oa.load_golden(base)
    want = oa.load_option_sam(base, tag)
    p = gm.default_params_cs()
    for k, v in fields.items(): setattr(p, k, v)
    ix = gm.Index(contigs, params=p); s = gm.Session(ix, params=p, max_batch_reads=1024)
    got = oa.sam_header(contigs) + s.map_reads_cs(reads)
    s.close(); ix.close()
    assert got == want, _first_diff(got, want)


@pytest.mark.parametrize("tag", sorted(oa.RNA_CASES))
def test_rna_sequences_match_reference_golden(gm, tag):
    """RNA contigs and RNA reads (uracil and no thymine, ref: fasta.c:528-542) on the GPU: a contig's own flag in its reverse complement (A <-> U) and its colour
    translation (U read as T; genome.c:1107-1118), the LAST contig's flag as genome_is_rna in the first-colour row of pass 1, in sw_gapless and in sw_full_cs's letter
    translations (genome.c:1063-1064; mapping.c:375-388,1318-1327), a letter-space read's own flag in its reverse complement (gmapper.c:487) -- byte-identical to
    gmapper-ls / gmapper-cs on the rna_* fixtures: through the file entries (the files the reference read) and through the packed-code entries"""
    G = os.path.join(os.path.dirname(__file__), "golden")
    g = oa.load_rna_case(tag)
    p = gm.default_params_cs() if g["colour"] else gm.default_params()
    p.sam_unaligned = 1
    if g["opts"] == "local=1;ungapped=1":                     # --local -U with the companions the binary sets (ref: gmapper.c:2057-2062)
        for k, v in dict(local_alignment=1, ungapped=1, anchor_width=0, a_gap_open_score=-255, b_gap_open_score=-255, hash_filter_calls=0).items(): setattr(p, k, v)
    else: assert g["opts"] is None
    ix = gm.Index(g["contigs"], names=g["contig_names"], params=p)
    s = gm.Session(ix, params=p, max_batch_reads=256)         # (several sub-batches: the per-read flags of each)
    head = oa.sam_header(g["contigs"], g["contig_names"])
    paths = [os.path.join(G, f) for f in g["files"]]
    if g["pairing"]:
        mode, lo, hi = g["pairing"]
        got = head + s.map_pairs_file(paths[0], paths[1], mode=mode, min_insert=lo, max_insert=hi)
        assert got == g["sam"], _first_diff(got, g["sam"])
        (n1, m1, _), (n2, m2, _) = g["reads"]
        got = head + s.map_pairs(m1, m2, n1, n2, mode=mode, min_insert=lo, max_insert=hi)
        assert got == g["sam"], _first_diff(got, g["sam"])
    else:
        got = head + s.map_reads_file(paths[0], qual_delta=33 if paths[0].endswith(".fq.gz") else None)
        assert got == g["sam"], (_first_diff(got, g["sam"]), s.stats)
        names, codes, quals = g["reads"][0]
        if not g["colour"]:
            got = head + s.map_reads(codes, names)
            assert got == g["sam"], _first_diff(got, g["sam"])
        elif quals is None:
            got = head + s.map_reads_cs(codes, names)
            assert got == g["sam"], _first_diff(got, g["sam"])
    s.close(); ix.close()
