"""ctypes wrapper over oracle/libgm_oracle.so (the CPU restatement).

TEST INFRASTRUCTURE: imported only by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg -- never by the product package shrimp_amd.
"""
import ctypes as C
import gzip, os, subprocess
import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
_LIB = None


def build():
    src = [os.path.join(ROOT, "oracle", f) for f in ("gm_oracle_main.cpp", "gm_oracle.hpp")]
    so = os.path.join(ROOT, "oracle", "libgm_oracle.so")
    if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in src):
        subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "libgm_oracle.so"], check=True, capture_output=True)
    return so


def load():
    global _LIB
    if _LIB is not None:
        return _LIB
    L = C.CDLL(build())
    u32p, u8p = C.POINTER(C.c_uint32), C.POINTER(C.c_uint8)
    L.gmo_sw_vector.argtypes = [u32p, C.c_int, C.c_int, u32p, C.c_int]; L.gmo_sw_vector.restype = C.c_int
    L.gmo_sw_full_ls.argtypes = [u32p, C.c_int, C.c_int, u32p, C.c_int, C.c_longlong, C.c_longlong, C.c_int, C.c_int, C.c_int,
                                 C.POINTER(C.c_int), C.c_char_p, C.c_char_p, C.c_int]
    L.gmo_sw_full_ls.restype = C.c_int
    L.gmo_sw_full_ls_local.argtypes = [u32p, C.c_int, C.c_int, u32p, C.c_int, C.c_int, C.c_int, C.c_longlong, C.c_longlong, C.c_int, C.c_int, C.c_int, C.c_int,
                                       C.POINTER(C.c_int), C.c_char_p, C.c_char_p, C.c_int]
    L.gmo_sw_full_ls_local.restype = C.c_int
    L.gmo_session_create.argtypes = [C.c_int, C.POINTER(u8p), C.POINTER(C.c_uint64), C.c_void_p]; L.gmo_session_create.restype = C.c_void_p
    L.gmo_session_create_opts.argtypes = [C.c_int, C.POINTER(u8p), C.POINTER(C.c_uint64), C.c_void_p, C.c_char_p]; L.gmo_session_create_opts.restype = C.c_void_p
    L.gmo_session_destroy.argtypes = [C.c_void_p]
    L.gmo_session_cutoff.argtypes = [C.c_void_p]; L.gmo_session_cutoff.restype = C.c_uint
    L.gmo_session_set.argtypes = [C.c_void_p, C.c_int, C.c_int]
    L.gmo_session_set_half_paired.argtypes = [C.c_void_p, C.c_int]
    L.gmo_last_pair_counts.argtypes = [C.POINTER(C.c_uint64)]
    L.gmo_map_sam.argtypes = [C.c_void_p, C.c_int, C.c_int, u8p, C.c_char_p, C.c_int, C.POINTER(C.c_uint64)]; L.gmo_map_sam.restype = C.c_void_p
    L.gmo_session_set_pairing.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
    L.gmo_map_pairs_sam.argtypes = [C.c_void_p, C.c_int, C.c_int, u8p, C.c_int, u8p, C.c_char_p, C.c_char_p, C.c_int]; L.gmo_map_pairs_sam.restype = C.c_void_p
    L.gmo_free.argtypes = [C.c_void_p]
    L.gmo_sw_vector_cs.argtypes = [u32p, C.c_int, C.c_int, u32p, C.c_int, u32p, C.c_int]; L.gmo_sw_vector_cs.restype = C.c_int
    L.gmo_sw_full_cs.argtypes = [u32p, C.c_int, C.c_int, u32p, C.c_int, C.c_int, C.c_int, C.c_longlong, C.c_longlong, C.c_int, C.c_int, C.c_int,
                                 C.POINTER(C.c_int), C.c_char_p, C.c_char_p, C.c_int]; L.gmo_sw_full_cs.restype = C.c_int
    L.gmo_index_selfcheck.argtypes = [C.c_void_p, C.c_int]; L.gmo_index_selfcheck.restype = C.c_int
    L.gmo_set_threads.argtypes = [C.c_int]
    L.gmo_map_tophits.argtypes = [C.c_void_p, C.c_int, C.c_int, u8p, C.c_int, C.POINTER(C.c_longlong), C.c_long]; L.gmo_map_tophits.restype = C.c_long
    _LIB = L
    return L


PAIR_MODES = {"none": 0, "opp-in": 1, "opp-out": 2, "col-fw": 3, "col-bw": 4}


class Session:
    def __init__(self, contigs, contig_names=None, opts=None):
        self.L = load()
        self.contigs = [np.ascontiguousarray(c, dtype=np.uint8) for c in contigs]
        n = len(self.contigs)
        ptrs = (C.POINTER(C.c_uint8) * n)(*[c.ctypes.data_as(C.POINTER(C.c_uint8)) for c in self.contigs])
        lens = (C.c_uint64 * n)(*[len(c) for c in self.contigs])
        names = None
        if contig_names is not None:
            self._names = (C.c_char_p * n)(*[bytes(x) for x in contig_names]); names = C.cast(self._names, C.c_void_p)
        self.h = self.L.gmo_session_create_opts(n, ptrs, lens, names, opts.encode() if opts else None)

    def index_selfcheck(self, nthreads):
        """chunk-parallel index builder == sequential restatement of load_genome (genome.c:1012-1182)"""
        return bool(self.L.gmo_index_selfcheck(self.h, int(nthreads)))

    def set_pairing(self, mode, min_insert, max_insert):
        self.L.gmo_session_set_pairing(self.h, PAIR_MODES[mode] if isinstance(mode, str) else int(mode), int(min_insert), int(max_insert))

    def map_pairs_sam(self, m1, m2, names1=None, names2=None, nthreads=4):
        m1 = np.ascontiguousarray(m1, dtype=np.uint8); m2 = np.ascontiguousarray(m2, dtype=np.uint8)
        n1 = b"\n".join(bytes(x) for x in names1) if names1 is not None else None
        n2 = b"\n".join(bytes(x) for x in names2) if names2 is not None else None
        p = self.L.gmo_map_pairs_sam(self.h, m1.shape[0], m1.shape[1], m1.ctypes.data_as(C.POINTER(C.c_uint8)),
                                     m2.shape[1], m2.ctypes.data_as(C.POINTER(C.c_uint8)), n1, n2, nthreads)
        s = C.string_at(p); self.L.gmo_free(p)
        return s

    def map_pairs_sam_q(self, m1, m2, quals1, quals2, qual_delta=64, names1=None, names2=None, nthreads=4):
        m1 = np.ascontiguousarray(m1, dtype=np.uint8); m2 = np.ascontiguousarray(m2, dtype=np.uint8)
        n1 = b"\n".join(bytes(x) for x in names1) if names1 is not None else None
        n2 = b"\n".join(bytes(x) for x in names2) if names2 is not None else None
        u8p = C.POINTER(C.c_uint8)
        self.L.gmo_map_pairs_sam_q.restype = C.c_void_p
        self.L.gmo_map_pairs_sam_q.argtypes = [C.c_void_p, C.c_int, C.c_int, u8p, C.c_int, u8p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_int, C.c_int]
        p = self.L.gmo_map_pairs_sam_q(self.h, m1.shape[0], m1.shape[1], m1.ctypes.data_as(u8p), m2.shape[1], m2.ctypes.data_as(u8p), n1, n2,
                                       b"\n".join(quals1), b"\n".join(quals2), qual_delta, nthreads)
        s = C.string_at(p); self.L.gmo_free(p)
        return s

    def set_half_paired(self, on):
        self.L.gmo_session_set_half_paired(self.h, int(bool(on)))

    def last_pair_counts(self):
        """(collapsed anchors, windows) over both mates and strands of the last paired call"""
        o = (C.c_uint64 * 2)(); self.L.gmo_last_pair_counts(o); return int(o[0]), int(o[1])

    def set(self, hash_filter_calls=True, sam_unaligned=False):
        self.L.gmo_session_set(self.h, int(hash_filter_calls), int(sam_unaligned))

    @property
    def cutoff(self):
        return self.L.gmo_session_cutoff(self.h)

    def map_sam(self, reads, nthreads=4):
        reads = np.ascontiguousarray(reads, dtype=np.uint8)
        n, Lr = reads.shape
        st = (C.c_uint64 * 7)()
        p = self.L.gmo_map_sam(self.h, n, Lr, reads.ctypes.data_as(C.POINTER(C.c_uint8)), None, nthreads, st)
        s = C.string_at(p)
        self.L.gmo_free(p)
        self.stats = dict(vec_calls=st[0], vec_cells=st[1], vec_bypassed=st[2], full_calls=st[3], reads_matched=st[4], dup_pruned=st[5], local_retries=st[6])
        return s

    def map_sam_q(self, reads, quals, qual_delta=64, nthreads=4):
        """FASTQ reads: quals = list of QUAL strings (bytes) as in the file"""
        reads = np.ascontiguousarray(reads, dtype=np.uint8)
        n, Lr = reads.shape
        self.L.gmo_map_sam_q.restype = C.c_void_p
        self.L.gmo_map_sam_q.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_uint8), C.c_char_p, C.c_char_p, C.c_int, C.c_int]
        p = self.L.gmo_map_sam_q(self.h, n, Lr, reads.ctypes.data_as(C.POINTER(C.c_uint8)), None, b"\n".join(quals), qual_delta, nthreads)
        s = C.string_at(p)
        self.L.gmo_free(p)
        return s

    def tophits(self, reads, nthreads=4):
        reads = np.ascontiguousarray(reads, dtype=np.uint8)
        n, Lr = reads.shape
        cap = n * 30
        rows = np.zeros((cap, 12), dtype=np.int64)
        w = self.L.gmo_map_tophits(self.h, n, Lr, reads.ctypes.data_as(C.POINTER(C.c_uint8)), nthreads,
                                   rows.ctypes.data_as(C.POINTER(C.c_longlong)), cap)
        return rows[:w]

    def close(self):
        if self.h:
            self.L.gmo_session_destroy(self.h); self.h = None

    def __del__(self):
        try: self.close()
        except Exception: pass


def sam_header(contigs, names=None):
    names = names if names is not None else [b"contig%d" % (i + 1) for i in range(len(contigs))]
    return b"@HD\tVN:1.0\tSO:unsorted\n" + b"".join(b"@SQ\tSN:%s\tLN:%d\n" % (bytes(nm), len(c)) for nm, c in zip(names, contigs))


def load_golden_pairs(name):
    d = os.path.join(ROOT, "tests", "golden")
    z = np.load(os.path.join(d, name + ".npz"))
    contigs = [z["contig%d" % i] for i in range(sum(1 for f in z.files if f.startswith("contig") and f[6:].isdigit()))]
    with gzip.open(os.path.join(d, name + ".sam.gz"), "rb") as f:
        sam = f.read()
    return dict(contigs=contigs, contig_names=[bytes(x) for x in z["contig_names"]], m1=z["mates1"], m2=z["mates2"],
                names1=[bytes(x) for x in z["names1"]], names2=[bytes(x) for x in z["names2"]],
                mode=str(z["mode"]), ins=tuple(int(x) for x in z["ins"]), sam=sam)


def load_golden(name):
    d = os.path.join(ROOT, "tests", "golden")
    z = np.load(os.path.join(d, name + ".npz"))
    contigs = [z["contig%d" % i] for i in range(len(z.files) - 1)]
    with gzip.open(os.path.join(d, name + ".sam.gz"), "rb") as f:
        sam = f.read()
    return contigs, z["reads"], sam


def parse_words(s):
    return np.array([int(x, 16) for x in s.split(",")], dtype=np.uint32)


def load_kat():
    with gzip.open(os.path.join(ROOT, "tests", "golden", "sw_kat.txt.gz"), "rt") as f:
        for line in f:
            t = line.split()
            if t[0] == "V":
                yield ("V", int(t[1]), int(t[2]), int(t[3]), parse_words(t[4]), parse_words(t[5]), int(t[6]))
            else:
                yield ("F", int(t[1]), int(t[2]), int(t[3]), int(t[4]), int(t[5]), int(t[6]), int(t[7]), int(t[8]),
                       parse_words(t[9]), parse_words(t[10]), [int(x) for x in t[11:20]], t[20], t[21])


# Non-default option sets whose reference output is committed as <base>@<tag>.sam.gz (tools/make_golden.py OPTION_CASES):
# tag -> (base golden, oracle option string (the reference's long option names), product gm_params_t fields, seeds, sam_unaligned)
OPTION_CASES = {
    "strata": ("stress_60bp", "strata=1", dict(strata=1), None),
    "max3_o5": ("stress_60bp", "max-alignments=3;report=5", dict(max_alignments=3, num_outputs=5), None),
    "scores": ("stress_100bp_unal", "match=8;mismatch=-12;open-r=-30;open-q=-28;ext-r=-5;ext-q=-4;full-threshold=60;vec-threshold=60;"
               "match-window=150;cmw-overlap=80;cmw-threshold=50;report=6;anchor-width=10",
               dict(match_score=8, mismatch_score=-12, a_gap_open_score=-30, b_gap_open_score=-28, a_gap_extend_score=-5, b_gap_extend_score=-4,
                    sw_full_threshold=60.0, sw_vect_threshold=60.0, window_len=150.0, window_overlap=80.0, window_gen_threshold=50.0,
                    num_outputs=6, anchor_width=10, sam_unaligned=1), None),
    "seeds": ("cfg2s_100bp_2Mbp", "seeds=1111101111,110110110110111,1110100111010111;cutoff=40", dict(list_cutoff=40),
              ["1111101111", "110110110110111", "1110100111010111"]),
    "pairs_strata": ("stress_pairs_2x100", "strata=1;report=4", dict(strata=1, num_outputs=4), None),
    # --local: sw_full_ls with local_alignment = 1 (soft clips), no mapping qualities (MAPQ 255, no Z0/Z1)
    "local": ("stress_100bp_unal", "local=1", dict(local_alignment=1, sam_unaligned=1), None),
    "local60": ("stress_60bp", "local=1;full-threshold=40;vec-threshold=40", dict(local_alignment=1, sw_full_threshold=40.0, sw_vect_threshold=40.0), None),
    "local_cfg2": ("cfg2s_100bp_2Mbp", "local=1", dict(local_alignment=1), None),
    # -U: ungapped filter (sw_gapless) in pass 1, one window per anchor, anchor_width 0, gap opens -255, no f1 cache; needs --local
    "ungapped": ("stress_100bp_unal", "local=1;ungapped=1", dict(local_alignment=1, ungapped=1, anchor_width=0, a_gap_open_score=-255, b_gap_open_score=-255,
                                                                    hash_filter_calls=0, sam_unaligned=1), None),
    "ungapped60_n1": ("stress_60bp", "local=1;ungapped=1;cmw-mode=1;full-threshold=45;vec-threshold=45",
                      dict(local_alignment=1, ungapped=1, anchor_width=0, a_gap_open_score=-255, b_gap_open_score=-255, hash_filter_calls=0, match_mode=1,
                           sw_full_threshold=45.0, sw_vect_threshold=45.0), None),
    # output policy (output.c:955-1008,1070-1291)
    "single_best": ("stress_60bp", "single-best-mapping=1", dict(single_best_mapping=1), None),
    "all_contigs": ("stress_60bp", "all-contigs=1", dict(all_contigs=1), None),
    "no_mapq": ("stress_100bp_unal", "no-mapping-qualities=1", dict(no_mapping_qualities=1, sam_unaligned=1), None),
    "pairs_single_best": ("stress_pairs_2x100", "single-best-mapping=1", dict(single_best_mapping=1), None),
    "pairs_single_best_all": ("stress_pairs_2x100", "single-best-mapping=1;all-contigs=1", dict(single_best_mapping=1, all_contigs=1), None),
    "pairs_single_best_all_noimp": ("stress_pairs_2x100", "single-best-mapping=1;all-contigs=1;no-improper-mappings=1",
                                    dict(single_best_mapping=1, all_contigs=1, no_improper_mappings=1), None),
    "pairs_no_mapq": ("stress_pairs_2x100", "no-mapping-qualities=1", dict(no_mapping_qualities=1), None),
    # chimeric pairs (mates from different places): improper pairs of two unpaired mappings with --single-best-mapping --all-contigs (output.c:1178-1226)
    "chim_single_best_all": ("chimeric_pairs_2x150", "single-best-mapping=1;all-contigs=1", dict(single_best_mapping=1, all_contigs=1), None),
    "chim_single_best_all_noimp": ("chimeric_pairs_2x150", "single-best-mapping=1;all-contigs=1;no-improper-mappings=1",
                                   dict(single_best_mapping=1, all_contigs=1, no_improper_mappings=1), None),
    "chim_single_best": ("chimeric_pairs_2x150", "single-best-mapping=1", dict(single_best_mapping=1), None),
    # --region-bits / --region-overlap
    "regions_10_30": ("stress_60bp", "region-bits=10;region-overlap=30", dict(region_bits=10, region_overlap=30), None),
    "regions_12_200": ("cfg2s_100bp_2Mbp", "region-bits=12;region-overlap=200", dict(region_bits=12, region_overlap=200), None),
    # -t: Tflag off
    "tiebreak_off": ("stress_100bp_unal", "tiebreak-off=1", dict(tiebreak_rev=0, sam_unaligned=1), None),
    "pairs_tiebreak_off": ("stress_pairs_2x100", "tiebreak-off=1", dict(tiebreak_rev=0), None),
    # -F / -C: one strand only (mapping.c:879-880)
    "positive": ("stress_60bp", "positive=1", dict(strand_only=1), None),
    "negative": ("stress_60bp", "negative=1", dict(strand_only=2), None),
    # -n 1: use_region_counts off -- all list entries are anchors, a window per anchor, one k-mer match is enough (gmapper.c:2610-2625)
    "n1": ("stress_60bp", "cmw-mode=1", dict(match_mode=1), None),
    # -H: hashed seeds (4^12 lists per seed, any weight)
    "hashed": ("stress_60bp", "hash-spaced-kmers=1", dict(hash_seeds=1), None),
    "hashed_w16": ("cfg2s_100bp_2Mbp", "hash-spaced-kmers=1;seeds=11111111101111111,1111110111011101111,111101110010000101111011", dict(hash_seeds=1),
                   ["11111111101111111", "1111110111011101111", "111101110010000101111011"]),
    "pairs_hashed": ("stress_pairs_2x100", "hash-spaced-kmers=1;report=3", dict(hash_seeds=1, num_outputs=3), None),
    "pairs_local": ("stress_pairs_2x100", "local=1", dict(local_alignment=1), None),
    "pairs_ungapped": ("stress_pairs_2x100", "local=1;ungapped=1", dict(local_alignment=1, ungapped=1, anchor_width=0, a_gap_open_score=-255, b_gap_open_score=-255,
                                                                        hash_filter_calls=0), None),
}


# paired match modes (tools/make_golden.py OPTION_CASES pairs_n3 ... cfg5_n2): tag -> (base pair golden, oracle option string, gm_pair_opts_t fields)
#   -n 3: a region marked once counts when the mate has hits within reach (use_mp_region_counts 2; 3 with --no-half-paired), hit list mode 3 (mapping.c:733-742,1080-1093,1153-1157)
#   -n 2: no region counts at all, a window per anchor (gmapper.c:2652-2673)
PAIR_MODE_CASES = {
    "pairs_n3": ("stress_pairs_2x100", "mp-match-mode=3", dict(match_mode=3)),
    "pairs_n3_nhp": ("stress_pairs_2x100", "mp-match-mode=3;half-paired=0", dict(match_mode=3, half_paired=0)),
    "cfg5_n3": ("cfg5s_2x150_1Mbp", "mp-match-mode=3", dict(match_mode=3)),
    "cfg5_n3_nhp": ("cfg5s_2x150_1Mbp", "mp-match-mode=3;half-paired=0", dict(match_mode=3, half_paired=0)),
    "pairs_n2": ("stress_pairs_2x100", "mp-match-mode=2", dict(match_mode=2)),
    # ("param_" fields go to gm_params_t)
    "pairs_hashed_n3": ("stress_pairs_2x100", "hash-spaced-kmers=1;mp-match-mode=3", dict(match_mode=3, param_hash_seeds=1)),
    "pairs_local_n3_nhp": ("stress_pairs_2x100", "local=1;mp-match-mode=3;half-paired=0", dict(match_mode=3, half_paired=0, param_local_alignment=1)),
    "cfg5_n2": ("cfg5s_2x150_1Mbp", "mp-match-mode=2", dict(match_mode=2)),
}

# colour-space option sets (tools/make_golden.py CS_OPTION_CASES: gmapper-cs with these options on a committed colour-space golden's inputs):
# tag -> (base golden, oracle option string, product gm_params_t fields, sam_unaligned)
CS_OPTION_CASES = {
    "cs_local": ("cfg4s_50col_2Mbp", "colour=1;local=1", dict(local_alignment=1), False),
    "cs_local_unal": ("stress_cs_60col_unal", "colour=1;local=1", dict(local_alignment=1, sam_unaligned=1), True),
    "cs_ungapped": ("cfg4s_50col_2Mbp", "colour=1;local=1;ungapped=1", dict(local_alignment=1, ungapped=1, anchor_width=0, a_gap_open_score=-255, b_gap_open_score=-255, hash_filter_calls=0), False),
    "cs_ungapped_unal": ("stress_cs_60col_unal", "colour=1;local=1;ungapped=1;full-threshold=40",
                         dict(local_alignment=1, ungapped=1, anchor_width=0, a_gap_open_score=-255, b_gap_open_score=-255, hash_filter_calls=0, sw_full_threshold=40.0, sam_unaligned=1), True),
    "cs_tiebreak_off": ("stress_cs_60col_unal", "colour=1;tiebreak-off=1", dict(tiebreak_rev=0, sam_unaligned=1), True),
    "cs_xover_taboo": ("stress_cs_60col_unal", "colour=1;crossover=-25;indel-taboo-len=3;pr-xover=0.05",
                       dict(crossover_score=-25, indel_taboo_len=3, pr_xover=0.05, sam_unaligned=1), True),
    "cs_no_mapq": ("cfg4s_50col_2Mbp", "colour=1;no-mapping-qualities=1", dict(no_mapping_qualities=1), False),
    "cs_single_best": ("stress_cs_60col_unal", "colour=1;single-best-mapping=1", dict(single_best_mapping=1, sam_unaligned=1), True),
}


# colour-space pairs with options (tools/make_golden.py CS_PAIR_OPTION_CASES): tag -> (base pair golden, oracle option string, product gm_params_t fields)
CS_PAIR_OPTION_CASES = {
    "cs_pairs_local": ("cs_pairs_50col_opp-in", "colour=1;local=1", dict(local_alignment=1)),
    "cs_pairs_local_colbw": ("cs_pairs_50col_col-bw", "colour=1;local=1", dict(local_alignment=1)),
    # paired match modes in colour space ("pair_" fields go to gm_pair_opts_t)
    "cs_pairs_n3": ("cs_pairs_50col_opp-in", "colour=1;mp-match-mode=3", dict(pair_match_mode=3)),
    "cs_pairs_n3_colbw_nhp": ("cs_pairs_50col_col-bw", "colour=1;mp-match-mode=3;half-paired=0", dict(pair_match_mode=3, pair_half_paired=0)),
    "cs_pairs_n2": ("cs_pairs_50col_opp-in", "colour=1;mp-match-mode=2", dict(pair_match_mode=2)),
}


# -M mirna (gmapper.c:1497-1517 + gmapper-defaults.h:230-238): (oracle option string, gm_params_t fields, seeds)
MIRNA_SEEDS = ["00111111001111111100", "00111111110011111100", "00111111111100111100", "00111111111111001100", "00111111111111110000"]
MIRNA_MODE = ("hash-spaced-kmers=1;seeds=%s;local=1;ungapped=1;cmw-mode=1;match-window=100" % ",".join(MIRNA_SEEDS),
              dict(hash_seeds=1, ungapped=1, local_alignment=1, anchor_width=0, a_gap_open_score=-255, b_gap_open_score=-255, hash_filter_calls=0, match_mode=1,
                   window_len=100.0), MIRNA_SEEDS)


# the optional tail of the SAM records (--extra-sam-fields, --read-group, --sam-r2; output.c:452-465,729-756): host formatting pinned on the product directly against the
# reference's output -- the oracle does not restate it.  tag -> (base golden, gm_params_t fields, colour space, pairs)
SAM_TAIL_CASES = {
    "extra_fields": ("cfg2s_100bp_2Mbp", dict(extra_sam_fields=1), False, False),
    "extra_fields_rg_unal": ("n1_noisy_70bp", dict(extra_sam_fields=1, read_group=b"grp1", sam_unaligned=1), False, False),
    "rg_unal": ("stress_100bp_unal", dict(read_group=b"grp1", sam_unaligned=1), False, False),
    "pairs_r2_rg_extra": ("cfg5s_2x150_1Mbp", dict(sam_r2=1, read_group=b"grp1", extra_sam_fields=1), False, True),
    "pairs_r2_rg": ("stress_pairs_2x100", dict(sam_r2=1, read_group=b"grp1"), False, True),
    "cs_extra_rg": ("cfg4s_50col_2Mbp", dict(extra_sam_fields=1, read_group=b"grp1", sam_unaligned=1), True, False),
    "cs_pairs_r2_extra": ("cs_pairs_50col_col-bw", dict(sam_r2=1, extra_sam_fields=1, sam_unaligned=1), True, True),
}


def load_option_sam(base, tag):
    with gzip.open(os.path.join(ROOT, "tests", "golden", "%s@%s.sam.gz" % (base, tag)), "rb") as f:
        return f.read()


def load_kat_local():
    """sw_full_ls in local mode (Gflag off), with an anchor box and without, from the reference's own function (oracle/ref_kat.cpp local)"""
    with gzip.open(os.path.join(ROOT, "tests", "golden", "sw_kat_local.txt.gz"), "rt") as f:
        for line in f:
            t = line.split()
            if t[0] != "L": continue
            # goff glen rlen ax ay alen awidth rv no_anchor thresh sv | genome read | 9 ints | dbalign qralign
            yield (tuple(int(x) for x in t[1:12]), parse_words(t[12]), parse_words(t[13]), [int(x) for x in t[14:23]], t[23], t[24])


def load_kat_cs(name="sw_kat_cs.txt.gz"):
    """colour-space known answers produced by the reference's own sw_vector(use_colours) / sw_full_cs (oracle/ref_kat_cs.cpp); sw_kat_cs_local.txt.gz: "L" records, sw_full_cs in local mode"""
    recs = []
    words = lambda t: np.array([int(x, 16) for x in t.split(b",")], dtype=np.uint32)
    with gzip.open(os.path.join(ROOT, "tests", "golden", name), "rb") as f:
        for line in f:
            t = line.split()
            if t[0] == b"C":
                recs.append(("C", int(t[1]), int(t[2]), int(t[3]), int(t[4]), words(t[5]), words(t[6]), words(t[7]), int(t[8])))
            elif t[0] in (b"X", b"Y"):      # sw_kat_cs_xover.txt.gz: per-position crossover scores behind the read words (X: global, Y: local mode)
                recs.append((t[0].decode(), [int(x) for x in t[1:11]], words(t[11]), words(t[12]), [int(x) for x in t[14:24]],
                             b"" if t[24] == b"-" else t[24], b"" if t[25] == b"-" else t[25], np.array([int(x) for x in t[13].split(b",")], dtype=np.int32)))
            else:
                recs.append((t[0].decode(), [int(x) for x in t[1:11]], words(t[11]), words(t[12]), [int(x) for x in t[13:23]],
                             b"" if t[23] == b"-" else t[23], b"" if t[24] == b"-" else t[24]))
    return recs


# RNA fixtures (tools/make_golden.py rna_cases): tag -> (colour space?, genome file, read files, oracle options, pairing)
RNA_CASES = {
    "rna_ls": (False, "rna_genome.fa.gz", ["rna_reads_ls.fa.gz"], None, None),
    "rna_ls_pairs": (False, "rna_genome.fa.gz", ["rna_pairs_1.fa.gz", "rna_pairs_2.fa.gz"], None, ("opp-in", 100, 500)),
    "rna_ls_last_dna": (False, "rna_genome_last_dna.fa.gz", ["rna_reads_ls.fa.gz"], None, None),
    "rna_cs": (True, "rna_genome.fa.gz", ["rna_reads_cs.fa.gz"], None, None),
    "rna_cs_fq": (True, "rna_genome.fa.gz", ["rna_reads_cs.fq.gz"], None, None),
    "rna_cs_last_dna": (True, "rna_genome_last_dna.fa.gz", ["rna_reads_cs.fa.gz"], None, None),
    "rna_cs_last_rna": (True, "rna_genome_last_rna.fa.gz", ["rna_reads_cs.fa.gz"], None, None),
    "rna_cs_ungapped": (True, "rna_genome.fa.gz", ["rna_reads_cs.fa.gz"], "local=1;ungapped=1", None),
}
_LS_CODE = {c: i for i, c in enumerate(b"ACGTUMRWSYKVHDBN")}


def read_fastx(name, colour=False):
    """names, n x L code matrix (colour space: primer letter code first, then colours, 15 for '.'), QUAL strings or None -- of a fixed-length FASTA/FASTQ fixture"""
    with gzip.open(os.path.join(ROOT, "tests", "golden", name), "rb") as f:
        lines = [l for l in f.read().split(b"\n") if l]
    fq = lines[0][:1] == b"@"
    step = 4 if fq else 2
    names = [lines[i][1:] for i in range(0, len(lines), step)]
    seqs = [lines[i + 1] for i in range(0, len(lines), step)]
    quals = [lines[i + 3] for i in range(0, len(lines), step)] if fq else None
    if colour:
        codes = np.array([[_LS_CODE[s[0]]] + [c - 48 if 48 <= c <= 51 else 15 for c in s[1:]] for s in seqs], dtype=np.uint8)
    else:
        codes = np.array([[_LS_CODE[c] for c in s.upper()] for s in seqs], dtype=np.uint8)
    return names, codes, quals


def read_genome_fixture(name):
    """contig names and code arrays of a FASTA genome fixture"""
    with gzip.open(os.path.join(ROOT, "tests", "golden", name), "rb") as f:
        text = f.read()
    names, contigs = [], []
    for block in text.split(b">")[1:]:
        head, _, body = block.partition(b"\n")
        names.append(head.strip())
        contigs.append(np.array([_LS_CODE[c] for c in body.replace(b"\n", b"").upper()], dtype=np.uint8))
    return names, contigs


def load_rna_case(tag):
    colour, gname, rfiles, opts, pairing = RNA_CASES[tag]
    cn, contigs = read_genome_fixture(gname)
    with gzip.open(os.path.join(ROOT, "tests", "golden", tag + ".sam.gz"), "rb") as f:
        sam = f.read()
    return dict(colour=colour, contig_names=cn, contigs=contigs, reads=[read_fastx(r, colour) for r in rfiles], files=rfiles, opts=opts, pairing=pairing, sam=sam)
