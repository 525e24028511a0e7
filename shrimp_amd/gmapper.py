"""Python host mirror of the gmapper hot path on MI355X (ctypes over libgmapper_hip.so).

Mirrors the reference's call surface for this path (names, argument meaning, error behaviour):
    load_genome / genomemap globals   -> Index            (ref: gmapper/genome.c:1012-1182)
    sw_vector_setup / sw_vector       -> sw_vector_*      (ref: common/sw-vector.c:388-515)
    sw_full_ls_setup / sw_full_ls     -> sw_full_ls       (ref: common/sw-full-ls.c:568-683)
    handle_read + SAM emission        -> Session.map_reads (ref: gmapper/mapping.c:1773-1868, gmapper/output.c:227-774)
All compute happens in the HIP library; there is no Python/CPU fallback: importing works without
a GPU (so that symbol checks can run), every compute call fails loudly without one.
"""
from __future__ import annotations

import ctypes as C
import os
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GM_LIB_PATH") or os.path.join(_HERE, "libgmapper_hip.so")   # GM_LIB_PATH: an experimental build of the same ABI


class GmError(RuntimeError):
    pass


class Params(C.Structure):   # gm_params_t (include/gmapper_hip.h)
    _fields_ = [("match_score", C.c_int), ("mismatch_score", C.c_int),
                ("a_gap_open_score", C.c_int), ("a_gap_extend_score", C.c_int),
                ("b_gap_open_score", C.c_int), ("b_gap_extend_score", C.c_int),
                ("window_len", C.c_double), ("window_overlap", C.c_double), ("window_gen_threshold", C.c_double),
                ("sw_vect_threshold", C.c_double), ("sw_full_threshold", C.c_double),
                ("match_mode", C.c_int), ("num_outputs", C.c_int), ("num_tmp_outputs", C.c_int), ("anchor_width", C.c_int),
                ("region_bits", C.c_int), ("region_overlap", C.c_int), ("list_cutoff", C.c_uint32),
                ("hash_filter_calls", C.c_int), ("tiebreak_rev", C.c_int), ("sam_unaligned", C.c_int), ("longest_read_len", C.c_int),
                ("strata", C.c_int), ("max_alignments", C.c_int),
                ("colour_space", C.c_int), ("crossover_score", C.c_int), ("indel_taboo_len", C.c_int), ("pr_xover", C.c_double),
                ("local_alignment", C.c_int), ("ungapped", C.c_int), ("hash_seeds", C.c_int), ("output_format", C.c_int), ("print_read_seq", C.c_int), ("strand_only", C.c_int),
                ("single_best_mapping", C.c_int), ("all_contigs", C.c_int), ("no_mapping_qualities", C.c_int), ("no_improper_mappings", C.c_int),
                ("extra_sam_fields", C.c_int), ("sam_r2", C.c_int), ("read_group", C.c_char * 64),
                ("trim_front", C.c_int), ("trim_end", C.c_int), ("trim_first", C.c_int), ("trim_second", C.c_int), ("trim_illumina", C.c_int),
                ("min_avg_qv", C.c_int), ("ignore_qvs", C.c_int), ("no_qv_check", C.c_int)]


class MapStats(C.Structure):   # gm_map_stats_t
    _fields_ = [(n, C.c_uint64) for n in ("reads", "reads_matched", "sam_records", "lookups", "list_entries", "list_bytes",
                                          "survivors", "anchors", "windows", "vec_calls", "vec_cells", "vec_bypassed",
                                          "full_calls", "full_cells", "exact_order_reads", "retries", "survivors_pruned", "mp_unfiltered", "post_sw_host_redo")] + \
               [(n, C.c_double) for n in ("ms_lookup", "ms_anchors", "ms_pass1", "ms_select", "ms_pass2", "ms_host")]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


PAIR_MODES = {"opp-in": 1, "opp-out": 2, "col-fw": 3, "col-bw": 4}   # ref: gmapper-definitions.h:42-46, -p option gmapper.c:1583-1600


class PairOpts(C.Structure):    # gm_pair_opts_t
    _fields_ = [("pair_mode", C.c_int), ("min_insert_size", C.c_int), ("max_insert_size", C.c_int),
                ("insert_size_mean", C.c_double), ("insert_size_stddev", C.c_double), ("half_paired", C.c_int), ("match_mode", C.c_int)]

    @staticmethod
    def default(mode="opp-in", min_insert=0, max_insert=1000):
        o = PairOpts(); lib().gm_pair_opts_default(C.byref(o))
        o.pair_mode = PAIR_MODES[mode] if isinstance(mode, str) else int(mode)
        o.min_insert_size = int(min_insert); o.max_insert_size = int(max_insert)
        return o


class MergeOptions(C.Structure):   # gm_merge_options_t (mergesam's options of the same names, ref: mergesam/mergesam.c:216-243)
    _fields_ = [(n, C.c_int) for n in ("max_outputs", "max_alignments", "strata", "half_paired", "sam_unaligned", "single_best", "all_contigs",
                                       "no_mapping_qualities", "leave_mapq", "no_improper_mappings", "min_mapq", "fastq", "threads", "output",
                                       "header_given")] + [("command_line", C.c_char_p)]


MERGE_OUT = {"sam": 0, "un": 1, "al": 2}


def merge_sam(reads_text: bytes, sam_texts, command_line=None, output="sam", **options) -> bytes:
    """mergesam (ref: mergesam/mergesam.c): SAM texts of several runs -> one, in the order of the reads text, mapping qualities recomputed from
    the Z fields.  options: the fields of gm_merge_options_t (max_outputs, strata, single_best, all_contigs, sam_unaligned, ...)."""
    L = lib(); o = MergeOptions(); L.gm_merge_options_default(C.byref(o))
    for k, v in options.items():
        if not hasattr(o, k):
            raise TypeError("merge_sam: unknown option %r" % k)
        setattr(o, k, int(v))
    o.output = MERGE_OUT[output] if isinstance(output, str) else int(output)
    if command_line is not None:
        o.command_line = command_line if isinstance(command_line, bytes) else command_line.encode()
    texts = [bytes(t) for t in sam_texts]
    arr = (C.c_char_p * len(texts))(*texts); lens = (C.c_size_t * len(texts))(*[len(t) for t in texts])
    out = C.c_void_p(); ol = C.c_size_t()
    _check(L.gm_merge_sam(C.byref(o), reads_text, len(reads_text), len(texts), arr, lens, C.byref(out), C.byref(ol)), "gm_merge_sam")
    res = C.string_at(out, ol.value) if out.value else b""
    if out.value: L.gm_free(out)
    return res


class Anchor(C.Structure):      # struct gm_anchor == the reference's struct anchor (gmapper-definitions.h:66-74)
    _fields_ = [("x", C.c_longlong), ("y", C.c_longlong), ("length", C.c_int), ("width", C.c_int),
                ("weight", C.c_int), ("cn", C.c_int), ("score", C.c_int)]


class SwFullResults(C.Structure):   # struct gm_sw_full_results == the reference's struct sw_full_results (sw-full-common.h:13-48)
    _fields_ = [(n, C.c_int) for n in ("read_start", "rmapped", "genome_start", "gmapped", "matches", "mismatches",
                                       "insertions", "deletions", "score", "posterior_score", "pct_posterior_score")] + \
               [("dbalign", C.c_void_p), ("qralign", C.c_void_p), ("qual", C.c_void_p), ("posterior", C.c_double), ("mqv", C.c_int)] + \
               [(n, C.c_double) for n in ("z0", "z1", "z2", "z3", "pr_top_random_at_location", "pr_missed_mp", "insert_size_denom")] + \
               [("crossovers", C.c_int), ("dup", C.c_bool), ("in_use", C.c_bool)]


# every entry point include/gmapper_hip.h declares
EXPORTS = ["gm_map_pairs_file", "gm_map_reads_file_cb", "gm_map_pairs_file_cb", "gm_preprocess_read_text", "gm_map_pairs_cs_fastq", "gm_map_reads_file", "gm_merge_options_default", "gm_merge_sam", "gm_release_cache", "gm_map_pairs_cs", "gm_last_error", "gm_device_count", "gm_params_default", "gm_params_default_cs", "gm_index_build", "gm_index_free", "gm_index_list_cutoff",
           "gm_index_save", "gm_index_load", "gm_index_bytes", "gm_index_n_slabs", "gm_index_has_buckets", "gm_index_get_list", "gm_index_device_array", "gm_index_meta", "gm_index_alloc_like",
           "sw_vector_setup", "sw_vector", "sw_vector_stats", "sw_vector_cleanup", "gm_sw_vector_batch", "gm_sw_vector_batch_bounded",
           "sw_gapless_setup", "sw_gapless", "sw_gapless_stats", "gm_sw_gapless_batch",
           "sw_full_ls_setup", "sw_full_ls", "sw_full_ls_cleanup", "sw_full_ls_stats",
           "sw_full_cs_setup", "sw_full_cs", "sw_full_cs_cleanup", "sw_full_cs_stats", "gm_sw_vector_batch_cs",
           "post_sw_setup", "post_sw", "post_sw_cleanup", "post_sw_stats",
           "gm_session_create", "gm_session_free", "gm_sequence_to_bitfield", "gm_map_reads_text", "gm_map_reads", "gm_map_reads_fastq", "gm_map_reads_cs", "gm_map_reads_cs_fastq", "gm_map_reads_device", "gm_free", "gm_debug_tophits",
           "gm_pair_opts_default", "gm_map_pairs", "gm_map_pairs_fastq",
           "gm_last_lookup_timing", "gm_last_lookup_kernel", "gm_abi_sizeof"]

_lib = None


def lib():
    """The HIP library.  Raises (never falls back) when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise GmError(f"{LIB_PATH} is missing: build it with `make -C shrimp_amd/csrc` (or __graft_entry__.build())")
    L = C.CDLL(LIB_PATH)
    u32p, vp = C.POINTER(C.c_uint32), C.c_void_p
    L.gm_last_error.restype = C.c_char_p
    L.gm_last_lookup_kernel.restype = C.c_char_p
    L.gm_device_count.restype = C.c_int
    L.gm_params_default.argtypes = [C.POINTER(Params)]
    L.gm_params_default_cs.argtypes = [C.POINTER(Params)]
    L.gm_index_build.argtypes = [C.POINTER(vp), C.c_int, C.c_int, C.POINTER(u32p), u32p, C.POINTER(C.c_char_p), C.c_int, C.POINTER(C.c_char_p), C.POINTER(Params)]
    L.gm_index_free.argtypes = [vp]
    L.gm_index_save.argtypes = [vp, C.c_char_p]
    L.gm_index_load.argtypes = [C.POINTER(vp), C.c_int, C.c_char_p, C.POINTER(Params)]
    L.gm_index_list_cutoff.argtypes = [vp]; L.gm_index_list_cutoff.restype = C.c_uint32
    L.gm_index_bytes.argtypes = [vp]; L.gm_index_bytes.restype = C.c_uint64
    L.gm_index_n_slabs.argtypes = [vp]; L.gm_index_n_slabs.restype = C.c_int
    L.gm_index_has_buckets.argtypes = [vp]; L.gm_index_has_buckets.restype = C.c_int
    L.gm_index_get_list.argtypes = [vp, C.c_int, C.c_uint32, u32p, u32p, C.c_uint32]
    L.gm_index_device_array.argtypes = [vp, C.c_int, C.POINTER(vp), C.POINTER(C.c_uint64)]
    L.gm_index_meta.argtypes = [vp, vp, C.POINTER(C.c_uint64)]
    L.gm_index_alloc_like.argtypes = [C.POINTER(vp), C.c_int, vp, C.c_uint64]
    L.sw_vector_setup.argtypes = [C.c_int] * 9 + [C.c_bool]
    L.sw_vector.argtypes = [u32p, C.c_int, C.c_int, u32p, C.c_int, u32p, C.c_int, C.c_bool]
    L.gm_sw_vector_batch.argtypes = [C.c_int, u32p, C.c_uint64, C.POINTER(C.c_int64), C.POINTER(C.c_int), u32p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.gm_sw_vector_batch_bounded.argtypes = [C.c_int, u32p, C.c_uint64, C.POINTER(C.c_int64), C.POINTER(C.c_int), u32p, C.c_int, C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_int),
                                             C.POINTER(C.c_uint8)]
    L.sw_gapless_setup.argtypes = [C.c_int, C.c_int, C.c_bool]
    L.sw_gapless.argtypes = [u32p, C.c_int, u32p, C.c_int, C.c_int, C.c_int, u32p, C.c_int, C.c_bool]
    L.gm_sw_gapless_batch.argtypes = [C.c_int, u32p, u32p, C.c_uint64, C.POINTER(C.c_int64), C.POINTER(C.c_int), u32p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int),
                                      C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.sw_full_ls_setup.argtypes = [C.c_int] * 8 + [C.c_bool, C.c_int]
    L.sw_full_ls.argtypes = [u32p, C.c_int, C.c_int, u32p, C.c_int, C.c_int, C.c_int, C.POINTER(SwFullResults), C.c_bool, C.POINTER(Anchor), C.c_int, C.c_int]
    L.sw_full_ls.restype = None
    L.sw_full_cs_setup.argtypes = [C.c_int] * 9 + [C.c_bool, C.c_int, C.c_int]
    L.sw_full_cs.argtypes = [u32p, C.c_int, C.c_int, u32p, C.c_int, C.c_int, C.c_int, C.POINTER(SwFullResults), C.c_bool, C.c_bool, C.POINTER(Anchor), C.c_int, C.c_int, vp]
    L.sw_full_cs.restype = None
    L.gm_sw_vector_batch_cs.argtypes = [C.c_int, u32p, u32p, C.c_uint64, C.POINTER(C.c_int64), C.POINTER(C.c_int), u32p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.gm_session_create.argtypes = [C.POINTER(vp), vp, C.POINTER(Params), C.c_int]
    L.gm_session_free.argtypes = [vp]
    L.gm_map_reads.argtypes = [vp, C.c_int, C.c_int, u32p, C.c_char_p, C.POINTER(vp), C.POINTER(C.c_size_t), C.POINTER(MapStats)]
    L.gm_map_reads_fastq.argtypes = [vp, C.c_int, C.c_int, u32p, C.c_char_p, C.c_char_p, C.c_int, C.POINTER(vp), C.POINTER(C.c_size_t), C.POINTER(MapStats)]
    L.gm_map_reads_cs.argtypes = [vp, C.c_int, C.c_int, u32p, C.POINTER(C.c_uint8), C.c_char_p, C.POINTER(vp), C.POINTER(C.c_size_t), C.POINTER(MapStats)]
    L.gm_map_reads_cs_fastq.argtypes = [vp, C.c_int, C.c_int, u32p, C.POINTER(C.c_uint8), C.c_char_p, C.c_char_p, C.c_int, C.POINTER(vp), C.POINTER(C.c_size_t), C.POINTER(MapStats)]
    L.gm_map_reads_device.argtypes = [vp, C.c_int, C.c_int, vp, C.c_int, C.POINTER(vp), C.POINTER(C.c_size_t), C.POINTER(MapStats)]
    L.gm_free.argtypes = [vp]
    L.gm_pair_opts_default.argtypes = [C.POINTER(PairOpts)]
    L.gm_map_pairs_fastq.argtypes = [vp, C.c_int, C.c_int, u32p, C.c_int, u32p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_int, C.POINTER(PairOpts),
                                     C.POINTER(vp), C.POINTER(C.c_size_t), C.POINTER(MapStats)]
    L.gm_map_pairs.argtypes = [vp, C.c_int, C.c_int, u32p, C.c_int, u32p, C.c_char_p, C.c_char_p, C.POINTER(PairOpts),
                               C.POINTER(vp), C.POINTER(C.c_size_t), C.POINTER(MapStats)]
    L.gm_map_pairs_cs_fastq.argtypes = [vp, C.c_int, C.c_int, u32p, C.POINTER(C.c_uint8), C.c_int, u32p, C.POINTER(C.c_uint8), C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_int,
                                        C.POINTER(PairOpts), C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.POINTER(MapStats)]
    L.gm_map_pairs_file.argtypes = [vp, C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.POINTER(PairOpts), C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.POINTER(MapStats)]
    L.gm_map_reads_file.argtypes = [vp, C.c_char_p, C.c_int, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.POINTER(MapStats)]
    L.gm_merge_options_default.argtypes = [C.POINTER(MergeOptions)]
    L.gm_merge_sam.argtypes = [C.POINTER(MergeOptions), C.c_char_p, C.c_size_t, C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_size_t), C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
    L.gm_map_pairs_cs.argtypes = [vp, C.c_int, C.c_int, u32p, C.POINTER(C.c_uint8), C.c_int, u32p, C.POINTER(C.c_uint8), C.c_char_p, C.c_char_p, C.POINTER(PairOpts),
                                  C.POINTER(vp), C.POINTER(C.c_size_t), C.POINTER(MapStats)]
    L.gm_debug_tophits.argtypes = [vp, C.c_int, C.c_int, u32p, C.POINTER(C.c_longlong), C.c_long, C.POINTER(C.c_long)]
    L.gm_last_lookup_timing.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_uint64), C.POINTER(C.c_int)]
    _lib = L
    return L


def _check(rc, what):
    if rc != 0:
        raise GmError(f"{what} failed ({rc}): {lib().gm_last_error().decode()}")


def default_params() -> Params:
    p = Params()
    lib().gm_params_default(C.byref(p))
    return p


def default_params_cs() -> Params:
    """the gmapper-cs binary's defaults (colour space; ref: gmapper.c:1748-1755)"""
    p = Params()
    lib().gm_params_default_cs(C.byref(p))
    return p


def pack_codes(codes: np.ndarray) -> np.ndarray:
    """4-bit codes -> the reference's bitfield (8 bases per uint32; ref: common/util.h:41)."""
    from .synth import pack_nibbles
    return pack_nibbles(np.asarray(codes, dtype=np.uint8))


class Index:
    """Device-resident seed index + genome (the reference's genomemap/genome_contigs globals)."""

    def __init__(self, contigs, names=None, seeds=None, params: Params | None = None, device: int = 0):
        L = lib()
        self.params = params or default_params()
        self._packed = [np.ascontiguousarray(pack_codes(c)) for c in contigs]
        n = len(contigs)
        self.contig_len = [int(len(c)) for c in contigs]
        ptrs = (C.POINTER(C.c_uint32) * n)(*[p.ctypes.data_as(C.POINTER(C.c_uint32)) for p in self._packed])
        lens = (C.c_uint32 * n)(*self.contig_len)
        cn = None
        if names is not None:
            cn = (C.c_char_p * n)(*[s.encode() if isinstance(s, str) else s for s in names])
        sd = None
        ns = 0
        if seeds:
            ns = len(seeds); sd = (C.c_char_p * ns)(*[s.encode() for s in seeds])
        self.h = C.c_void_p()
        _check(L.gm_index_build(C.byref(self.h), device, n, ptrs, lens, cn, ns, sd, C.byref(self.params)), "gm_index_build")
        self.device = device

    @classmethod
    def load(cls, prefix: str, params: "Params | None" = None, device: int = 0) -> "Index":
        """the reference's -L: <prefix>.genome + <prefix>.seed.<n> as written by stock gmapper -S (or by save())"""
        self = cls.__new__(cls)
        self.params = params or default_params()
        self.h = C.c_void_p(); self.device = device; self._packed = []; self.contig_len = []
        _check(lib().gm_index_load(C.byref(self.h), device, prefix.encode(), C.byref(self.params)), "gm_index_load")
        return self

    def save(self, prefix: str) -> None:
        """the reference's -S: files stock gmapper can load with -L"""
        _check(lib().gm_index_save(self.h, prefix.encode()), "gm_index_save")

    @property
    def list_cutoff(self): return lib().gm_index_list_cutoff(self.h)
    @property
    def nbytes(self): return lib().gm_index_bytes(self.h)
    @property
    def n_slabs(self): return lib().gm_index_n_slabs(self.h)
    @property
    def has_buckets(self): return bool(lib().gm_index_has_buckets(self.h))

    def get_list(self, sn: int, mapidx: int) -> np.ndarray:
        L = lib(); n = C.c_uint32()
        _check(L.gm_index_get_list(self.h, sn, mapidx, C.byref(n), None, 0), "gm_index_get_list")
        out = np.zeros(n.value, dtype=np.uint32)
        if n.value:
            _check(L.gm_index_get_list(self.h, sn, mapidx, C.byref(n), out.ctypes.data_as(C.POINTER(C.c_uint32)), n.value), "gm_index_get_list")
        return out

    def device_arrays(self):
        """(device pointer, nbytes) of every resident array, for the start-up broadcast."""
        L = lib(); out = []
        kind = 0
        while True:
            p = C.c_void_p(); b = C.c_uint64()
            if L.gm_index_device_array(self.h, kind, C.byref(p), C.byref(b)) != 0:
                break
            out.append((p.value, b.value)); kind += 1
        return out

    def meta(self) -> bytes:
        L = lib(); nb = C.c_uint64(0)
        L.gm_index_meta(self.h, None, C.byref(nb))
        buf = C.create_string_buffer(nb.value)
        L.gm_index_meta(self.h, buf, C.byref(nb))
        return buf.raw

    @classmethod
    def alloc_like(cls, meta: bytes, device: int = 0) -> "Index":
        self = cls.__new__(cls)
        self.h = C.c_void_p(); self.device = device; self.params = default_params()
        _check(lib().gm_index_alloc_like(C.byref(self.h), device, meta, len(meta)), "gm_index_alloc_like")
        return self

    def close(self):
        if getattr(self, "h", None):
            lib().gm_index_free(self.h); self.h = None

    def __del__(self):
        try: self.close()
        except Exception: pass


class Session:
    """One mapping stream on one GPU (the reference's per-thread state)."""

    def __init__(self, index: Index, params: Params | None = None, max_batch_reads: int = 131072):
        self.index = index
        self.h = C.c_void_p()
        self.params = params or index.params
        _check(lib().gm_session_create(C.byref(self.h), index.h, C.byref(self.params), max_batch_reads), "gm_session_create")
        self.stats = None

    def map_reads(self, reads_codes: np.ndarray, names=None) -> bytes:
        """reads_codes: [n, L] uint8 4-bit codes -> SAM records (bytes), input order."""
        from .synth import pack_reads
        reads_codes = np.ascontiguousarray(reads_codes, dtype=np.uint8)
        n, Lr = reads_codes.shape
        packed = np.ascontiguousarray(pack_reads(reads_codes))
        return self.map_packed(packed, n, Lr, names)

    def map_reads_fastq(self, reads_codes: np.ndarray, quals, qual_delta: int = 64, names=None) -> bytes:
        """FASTQ reads: quals = one QUAL string (bytes) per read, as in the file; qual_delta = the file's offset (--qv-offset)."""
        from .synth import pack_reads
        reads_codes = np.ascontiguousarray(reads_codes, dtype=np.uint8)
        n, Lr = reads_codes.shape
        packed = np.ascontiguousarray(pack_reads(reads_codes))
        L = lib(); sam = C.c_void_p(); sl = C.c_size_t(); st = MapStats()
        nm = None if names is None else b"\n".join(x if isinstance(x, bytes) else x.encode() for x in names)
        _check(L.gm_map_reads_fastq(self.h, n, Lr, packed.ctypes.data_as(C.POINTER(C.c_uint32)), nm, b"\n".join(quals), int(qual_delta),
                                    C.byref(sam), C.byref(sl), C.byref(st)), "gm_map_reads_fastq")
        out = C.string_at(sam, sl.value) if sam.value else b""
        L.gm_free(sam)
        self.stats = st.as_dict()
        return out

    def map_reads_cs_fastq(self, reads_codes: np.ndarray, quals, qual_delta: int = 33, names=None) -> bytes:
        """csfastq reads: reads_codes as for map_reads_cs; quals = one QV string (bytes, one character per colour) per read."""
        from .synth import pack_reads
        reads_codes = np.ascontiguousarray(reads_codes, dtype=np.uint8)
        n, L1 = reads_codes.shape
        initbp = np.ascontiguousarray(reads_codes[:, 0])
        packed = np.ascontiguousarray(pack_reads(np.ascontiguousarray(reads_codes[:, 1:])))
        L = lib(); sam = C.c_void_p(); sl = C.c_size_t(); st = MapStats()
        nm = None if names is None else b"\n".join(x if isinstance(x, bytes) else x.encode() for x in names)
        _check(L.gm_map_reads_cs_fastq(self.h, n, L1 - 1, packed.ctypes.data_as(C.POINTER(C.c_uint32)), initbp.ctypes.data_as(C.POINTER(C.c_uint8)), nm,
                                       b"\n".join(quals), int(qual_delta), C.byref(sam), C.byref(sl), C.byref(st)), "gm_map_reads_cs_fastq")
        out = C.string_at(sam, sl.value) if sam.value else b""
        L.gm_free(sam)
        self.stats = st.as_dict()
        return out

    def map_reads_cs(self, reads_codes: np.ndarray, names=None) -> bytes:
        """Colour-space reads: reads_codes [n, 1 + colours] uint8, column 0 the primer letter code (A0 C1 G2 T3), then colours
        (0-3, 15 = skipped cycle) -- csfasta without the text.  Needs an index / session made with default_params_cs()."""
        from .synth import pack_reads
        reads_codes = np.ascontiguousarray(reads_codes, dtype=np.uint8)
        n, L1 = reads_codes.shape
        initbp = np.ascontiguousarray(reads_codes[:, 0])
        packed = np.ascontiguousarray(pack_reads(np.ascontiguousarray(reads_codes[:, 1:])))
        L = lib(); sam = C.c_void_p(); sl = C.c_size_t(); st = MapStats()
        nm = None
        if names is not None:
            nm = b"\n".join(x if isinstance(x, bytes) else x.encode() for x in names)
        _check(L.gm_map_reads_cs(self.h, n, L1 - 1, packed.ctypes.data_as(C.POINTER(C.c_uint32)), initbp.ctypes.data_as(C.POINTER(C.c_uint8)), nm,
                                 C.byref(sam), C.byref(sl), C.byref(st)), "gm_map_reads_cs")
        out = C.string_at(sam, sl.value) if sam.value else b""
        L.gm_free(sam)
        self.stats = st.as_dict()
        return out

    def map_reads_file(self, path, fastq=-1, qual_delta=None) -> bytes:
        """A reads file as the reference's reader takes it: FASTA / FASTQ, plain or gzip, any mix of read lengths (gm_map_reads_file)."""
        L = lib(); sam = C.c_void_p(); sl = C.c_size_t(); st = MapStats()
        qd = qual_delta if qual_delta is not None else (33 if self.params.colour_space else 64)      # gmapper-defaults.h:41-42
        _check(L.gm_map_reads_file(self.h, os.fsencode(path), int(fastq), int(qd), C.byref(sam), C.byref(sl), C.byref(st)), "gm_map_reads_file")
        out = C.string_at(sam, sl.value) if sam.value else b""
        if sam.value: L.gm_free(sam)
        self.stats = st.as_dict()
        return out

    def map_reads_file_chunks(self, path, chunk_reads=0, fastq=-1, qual_delta=None, collect=True):
        """gm_map_reads_file_cb: the file in chunks of at most chunk_reads reads; returns the list of texts the write function received (one per chunk; with
        collect=False their lengths only: the form a timed loop uses)."""
        L = lib(); st = MapStats(); parts = []
        qd = qual_delta if qual_delta is not None else (33 if self.params.colour_space else 64)
        WRITE = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t)
        cb = WRITE((lambda ctx, p, n: (parts.append(C.string_at(p, n)), 0)[1]) if collect else (lambda ctx, p, n: (parts.append(n), 0)[1]))
        _check(L.gm_map_reads_file_cb(self.h, os.fsencode(path), int(fastq), int(qd), C.c_size_t(int(chunk_reads)), cb, None, C.byref(st)), "gm_map_reads_file_cb")
        self.stats = st.as_dict()
        return parts

    def map_pairs_file_chunks(self, path1, path2=None, chunk_pairs=0, fastq=-1, qual_delta=None, mode="opp-in", min_insert=0, max_insert=1000, opts: "PairOpts | None" = None):
        """gm_map_pairs_file_cb: at most chunk_pairs pairs at a time; returns the texts the write function received."""
        L = lib(); st = MapStats(); parts = []
        qd = qual_delta if qual_delta is not None else (33 if self.params.colour_space else 64)
        o = opts if opts is not None else PairOpts.default(mode, min_insert, max_insert)
        WRITE = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t)
        cb = WRITE(lambda ctx, p, n: (parts.append(C.string_at(p, n)), 0)[1])
        _check(L.gm_map_pairs_file_cb(self.h, os.fsencode(path1), None if path2 is None else os.fsencode(path2), int(fastq), int(qd), C.byref(o), C.c_size_t(int(chunk_pairs)),
                                      cb, None, C.byref(st)), "gm_map_pairs_file_cb")
        self.stats = st.as_dict()
        return parts

    def map_pairs_file(self, path1, path2=None, fastq=-1, qual_delta=None, mode="opp-in", min_insert=0, max_insert=1000, opts: "PairOpts | None" = None) -> bytes:
        """Pairs from one file (mates adjacent) or two (-1 / -2): FASTA / FASTQ, plain or gzip, any mix of lengths (gm_map_pairs_file)."""
        L = lib(); sam = C.c_void_p(); sl = C.c_size_t(); st = MapStats()
        qd = qual_delta if qual_delta is not None else (33 if self.params.colour_space else 64)
        o = opts if opts is not None else PairOpts.default(mode, min_insert, max_insert)
        _check(L.gm_map_pairs_file(self.h, os.fsencode(path1), None if path2 is None else os.fsencode(path2), int(fastq), int(qd), C.byref(o),
                                   C.byref(sam), C.byref(sl), C.byref(st)), "gm_map_pairs_file")
        out = C.string_at(sam, sl.value) if sam.value else b""
        if sam.value: L.gm_free(sam)
        self.stats = st.as_dict()
        return out

    def map_pairs_cs(self, mates1: np.ndarray, mates2: np.ndarray, names1=None, names2=None, mode="opp-in", min_insert=0, max_insert=1000,
                     opts: "PairOpts | None" = None, quals1=None, quals2=None, qual_delta=33) -> bytes:
        """Colour-space pairs (gmapper-cs -p opp-in): mates as for map_reads_cs ([n, 1 + colours], column 0 the primer letter code)."""
        from .synth import pack_reads
        L = lib(); sam = C.c_void_p(); sl = C.c_size_t(); st = MapStats()
        m1 = np.ascontiguousarray(mates1, dtype=np.uint8); m2 = np.ascontiguousarray(mates2, dtype=np.uint8)
        if m1.shape[0] != m2.shape[0]:
            raise ValueError("mates1 and mates2 must hold the same number of reads")
        ib1 = np.ascontiguousarray(m1[:, 0]); ib2 = np.ascontiguousarray(m2[:, 0])
        p1 = np.ascontiguousarray(pack_reads(np.ascontiguousarray(m1[:, 1:]))); p2 = np.ascontiguousarray(pack_reads(np.ascontiguousarray(m2[:, 1:])))
        join = lambda names: None if names is None else b"\n".join(x if isinstance(x, bytes) else x.encode() for x in names)
        o = opts if opts is not None else PairOpts.default(mode, min_insert, max_insert)
        u32p, u8p = C.POINTER(C.c_uint32), C.POINTER(C.c_uint8)
        if quals1 is not None:
            jq = lambda q: b"\n".join(bytes(np.ascontiguousarray(r, dtype=np.uint8)) for r in q)
            _check(L.gm_map_pairs_cs_fastq(self.h, m1.shape[0], m1.shape[1] - 1, p1.ctypes.data_as(u32p), ib1.ctypes.data_as(u8p), m2.shape[1] - 1, p2.ctypes.data_as(u32p),
                                           ib2.ctypes.data_as(u8p), join(names1), join(names2), jq(quals1), jq(quals2), int(qual_delta), C.byref(o), C.byref(sam), C.byref(sl),
                                           C.byref(st)), "gm_map_pairs_cs_fastq")
        else:
            _check(L.gm_map_pairs_cs(self.h, m1.shape[0], m1.shape[1] - 1, p1.ctypes.data_as(u32p), ib1.ctypes.data_as(u8p), m2.shape[1] - 1, p2.ctypes.data_as(u32p),
                                     ib2.ctypes.data_as(u8p), join(names1), join(names2), C.byref(o), C.byref(sam), C.byref(sl), C.byref(st)), "gm_map_pairs_cs")
        out = C.string_at(sam, sl.value) if sam.value else b""
        if sam.value: L.gm_free(sam)
        self.stats = st.as_dict()
        return out

    def map_cs_packed(self, packed: np.ndarray, initbp: np.ndarray, n: int, n_colours: int, return_bytes: bool = True):
        """Colour-space reads already packed (pack_reads of the colours, one primer byte per read): the form a timed loop uses.
        With return_bytes=False the SAM text stays in the library's buffer and its length is returned."""
        L = lib(); sam = C.c_void_p(); sl = C.c_size_t(); st = MapStats()
        _check(L.gm_map_reads_cs(self.h, n, n_colours, packed.ctypes.data_as(C.POINTER(C.c_uint32)), initbp.ctypes.data_as(C.POINTER(C.c_uint8)), None,
                                 C.byref(sam), C.byref(sl), C.byref(st)), "gm_map_reads_cs")
        out = sl.value
        if return_bytes: out = C.string_at(sam, sl.value) if sam.value else b""
        if sam.value: L.gm_free(sam)
        self.stats = st.as_dict()
        return out

    def map_pairs_packed(self, p1: np.ndarray, p2: np.ndarray, n: int, len1: int, len2: int, opts: "PairOpts", return_bytes: bool = True):
        """Paired mode on packed mates (pack_reads of each mate set)."""
        L = lib(); sam = C.c_void_p(); sl = C.c_size_t(); st = MapStats()
        _check(L.gm_map_pairs(self.h, n, len1, p1.ctypes.data_as(C.POINTER(C.c_uint32)), len2, p2.ctypes.data_as(C.POINTER(C.c_uint32)),
                              None, None, C.byref(opts), C.byref(sam), C.byref(sl), C.byref(st)), "gm_map_pairs")
        out = sl.value
        if return_bytes: out = C.string_at(sam, sl.value) if sam.value else b""
        if sam.value: L.gm_free(sam)
        self.stats = st.as_dict()
        return out

    def map_reads_text(self, seqs, names=None, quals=None, qual_delta: int = 64) -> bytes:
        """Reads as text lines of one length (letters; or primer + colours in a colour-space session), packed inside the library."""
        L = lib(); sam = C.c_void_p(); sl = C.c_size_t(); st = MapStats()
        seqs = [x if isinstance(x, bytes) else x.encode() for x in seqs]
        n = len(seqs); read_len = len(seqs[0]) - (1 if self.params.colour_space else 0) if n else 1
        join = lambda v: None if v is None else b"\n".join(x if isinstance(x, bytes) else x.encode() for x in v)
        _check(L.gm_map_reads_text(self.h, n, read_len, b"\n".join(seqs), join(names), join(quals), int(qual_delta), C.byref(sam), C.byref(sl), C.byref(st)), "gm_map_reads_text")
        out = C.string_at(sam, sl.value) if sam.value else b""
        if sam.value: L.gm_free(sam)
        self.stats = st.as_dict()
        return out

    def map_text_buffer(self, seq_buf: bytes, n: int, read_len: int, names_buf: "bytes | None" = None, return_bytes: bool = True):
        """gm_map_reads_text on a ready-made buffer (n lines of read_len letters, '\\n' between them; names likewise): the form a timed loop uses -- parsing, packing and
        the upload all happen inside the call.  With return_bytes=False the SAM text stays in the library's buffer and its length is returned."""
        L = lib(); sam = C.c_void_p(); sl = C.c_size_t(); st = MapStats()
        _check(L.gm_map_reads_text(self.h, n, read_len, seq_buf, names_buf, None, 64, C.byref(sam), C.byref(sl), C.byref(st)), "gm_map_reads_text")
        out = sl.value
        if return_bytes: out = C.string_at(sam, sl.value) if sam.value else b""
        if sam.value: L.gm_free(sam)
        self.stats = st.as_dict()
        return out

    def map_packed(self, packed: np.ndarray, n: int, read_len: int, names=None) -> bytes:
        L = lib(); sam = C.c_void_p(); sl = C.c_size_t(); st = MapStats()
        nm = None
        if names is not None:
            nm = b"\n".join(x if isinstance(x, bytes) else x.encode() for x in names)
        _check(L.gm_map_reads(self.h, n, read_len, packed.ctypes.data_as(C.POINTER(C.c_uint32)), nm, C.byref(sam), C.byref(sl), C.byref(st)), "gm_map_reads")
        out = C.string_at(sam, sl.value) if sam.value else b""
        L.gm_free(sam)
        self.stats = st.as_dict()
        return out

    def map_pairs(self, mates1: np.ndarray, mates2: np.ndarray, names1=None, names2=None, mode="opp-in", min_insert=0, max_insert=1000,
                  opts: "PairOpts | None" = None) -> bytes:
        """Paired mode (-p mode -I min,max): mates1 [n, L1], mates2 [n, L2] uint8 codes -> SAM records of every pair, input order."""
        from .synth import pack_reads
        L = lib(); sam = C.c_void_p(); sl = C.c_size_t(); st = MapStats()
        m1 = np.ascontiguousarray(mates1, dtype=np.uint8); m2 = np.ascontiguousarray(mates2, dtype=np.uint8)
        if m1.shape[0] != m2.shape[0]:
            raise ValueError("mates1 and mates2 must hold the same number of reads")
        p1 = np.ascontiguousarray(pack_reads(m1)); p2 = np.ascontiguousarray(pack_reads(m2))
        join = lambda names: None if names is None else b"\n".join(x if isinstance(x, bytes) else x.encode() for x in names)
        o = opts if opts is not None else PairOpts.default(mode, min_insert, max_insert)
        _check(L.gm_map_pairs(self.h, m1.shape[0], m1.shape[1], p1.ctypes.data_as(C.POINTER(C.c_uint32)), m2.shape[1], p2.ctypes.data_as(C.POINTER(C.c_uint32)),
                              join(names1), join(names2), C.byref(o), C.byref(sam), C.byref(sl), C.byref(st)), "gm_map_pairs")
        out = C.string_at(sam, sl.value) if sam.value else b""
        L.gm_free(sam)
        self.stats = st.as_dict()
        return out

    def map_pairs_fastq(self, mates1: np.ndarray, mates2: np.ndarray, quals1, quals2, qual_delta: int = 64, names1=None, names2=None,
                        mode="opp-in", min_insert=0, max_insert=1000, opts: "PairOpts | None" = None) -> bytes:
        """Paired FASTQ: as map_pairs plus one QUAL string (bytes) per mate and the file's quality offset."""
        from .synth import pack_reads
        L = lib(); sam = C.c_void_p(); sl = C.c_size_t(); st = MapStats()
        m1 = np.ascontiguousarray(mates1, dtype=np.uint8); m2 = np.ascontiguousarray(mates2, dtype=np.uint8)
        p1 = np.ascontiguousarray(pack_reads(m1)); p2 = np.ascontiguousarray(pack_reads(m2))
        join = lambda names: None if names is None else b"\n".join(x if isinstance(x, bytes) else x.encode() for x in names)
        o = opts if opts is not None else PairOpts.default(mode, min_insert, max_insert)
        _check(L.gm_map_pairs_fastq(self.h, m1.shape[0], m1.shape[1], p1.ctypes.data_as(C.POINTER(C.c_uint32)), m2.shape[1], p2.ctypes.data_as(C.POINTER(C.c_uint32)),
                                    join(names1), join(names2), b"\n".join(quals1), b"\n".join(quals2), int(qual_delta), C.byref(o),
                                    C.byref(sam), C.byref(sl), C.byref(st)), "gm_map_pairs_fastq")
        out = C.string_at(sam, sl.value) if sam.value else b""
        L.gm_free(sam)
        self.stats = st.as_dict()
        return out

    def map_device(self, dev_ptr: int, n: int, read_len: int, emit_sam: bool = True, return_bytes: bool = True):
        """Reads already resident in HBM (packed layout).  With return_bytes=False the SAM text is produced
        in the library's buffer and released without a Python-side copy (returns its length)."""
        L = lib(); sam = C.c_void_p(); sl = C.c_size_t(); st = MapStats()
        _check(L.gm_map_reads_device(self.h, n, read_len, C.c_void_p(dev_ptr), int(emit_sam), C.byref(sam), C.byref(sl), C.byref(st)), "gm_map_reads_device")
        out = sl.value
        if return_bytes:
            out = C.string_at(sam, sl.value) if sam.value else b""
        if sam.value: L.gm_free(sam)
        self.stats = st.as_dict()
        return out

    def tophits(self, reads_codes: np.ndarray) -> np.ndarray:
        from .synth import pack_reads
        reads_codes = np.ascontiguousarray(reads_codes, dtype=np.uint8)
        n, Lr = reads_codes.shape
        packed = np.ascontiguousarray(pack_reads(reads_codes))
        rows = np.zeros((n * 30, 12), dtype=np.int64); nr = C.c_long()
        _check(lib().gm_debug_tophits(self.h, n, Lr, packed.ctypes.data_as(C.POINTER(C.c_uint32)),
                                      rows.ctypes.data_as(C.POINTER(C.c_longlong)), n * 30, C.byref(nr)), "gm_debug_tophits")
        return rows[:nr.value]

    def lookup_timing(self):
        ms = C.c_double(); b = C.c_uint64(); n = C.c_int()
        lib().gm_last_lookup_timing(self.h, C.byref(ms), C.byref(b), C.byref(n))
        return ms.value, b.value, n.value

    def close(self):
        if getattr(self, "h", None):
            lib().gm_session_free(self.h); self.h = None

    def __del__(self):
        try: self.close()
        except Exception: pass


# ---- S1 / S2 seams ---------------------------------------------------------------------------------
def sw_vector_setup(dblen, qrlen, a_gap_open, a_gap_ext, b_gap_open, b_gap_ext, match, mismatch, use_colours=0, reset_stats=True):
    _check(lib().sw_vector_setup(dblen, qrlen, a_gap_open, a_gap_ext, b_gap_open, b_gap_ext, match, mismatch, use_colours, reset_stats), "sw_vector_setup")


def sw_vector_batch(genome_words: np.ndarray, g_off, glen, reads_words: np.ndarray, rlen) -> np.ndarray:
    L = lib()
    genome_words = np.ascontiguousarray(genome_words, dtype=np.uint32)
    reads_words = np.ascontiguousarray(reads_words, dtype=np.uint32)
    n = reads_words.shape[0]
    g_off = np.ascontiguousarray(g_off, dtype=np.int64); glen = np.ascontiguousarray(glen, dtype=np.int32); rlen = np.ascontiguousarray(rlen, dtype=np.int32)
    out = np.zeros(n, dtype=np.int32)
    _check(L.gm_sw_vector_batch(n, genome_words.ctypes.data_as(C.POINTER(C.c_uint32)), genome_words.size,
                                g_off.ctypes.data_as(C.POINTER(C.c_int64)), glen.ctypes.data_as(C.POINTER(C.c_int)),
                                reads_words.ctypes.data_as(C.POINTER(C.c_uint32)), reads_words.shape[1],
                                rlen.ctypes.data_as(C.POINTER(C.c_int)), out.ctypes.data_as(C.POINTER(C.c_int))), "gm_sw_vector_batch")
    return out


def sw_vector_batch_bounded(genome_words: np.ndarray, g_off, glen, reads_words: np.ndarray, rlen, threshold: int):
    """gm_sw_vector_batch with pass 1's early stop against `threshold`: (scores, stopped); where stopped, the score is a lower bound below the threshold"""
    L = lib()
    genome_words = np.ascontiguousarray(genome_words, dtype=np.uint32)
    reads_words = np.ascontiguousarray(reads_words, dtype=np.uint32)
    n = reads_words.shape[0]
    g_off = np.ascontiguousarray(g_off, dtype=np.int64); glen = np.ascontiguousarray(glen, dtype=np.int32); rlen = np.ascontiguousarray(rlen, dtype=np.int32)
    out = np.zeros(n, dtype=np.int32); stopped = np.zeros(n, dtype=np.uint8)
    _check(L.gm_sw_vector_batch_bounded(n, genome_words.ctypes.data_as(C.POINTER(C.c_uint32)), genome_words.size,
                                        g_off.ctypes.data_as(C.POINTER(C.c_int64)), glen.ctypes.data_as(C.POINTER(C.c_int)),
                                        reads_words.ctypes.data_as(C.POINTER(C.c_uint32)), reads_words.shape[1],
                                        rlen.ctypes.data_as(C.POINTER(C.c_int)), int(threshold), out.ctypes.data_as(C.POINTER(C.c_int)),
                                        stopped.ctypes.data_as(C.POINTER(C.c_uint8))), "gm_sw_vector_batch_bounded")
    return out, stopped


def sw_vector(genome_words, goff, glen, read_words, rlen, genome_ls=None, initbp=-1, is_rna=False) -> int:
    g = np.ascontiguousarray(genome_words, dtype=np.uint32); r = np.ascontiguousarray(read_words, dtype=np.uint32)
    gl = None
    if genome_ls is not None:
        gla = np.ascontiguousarray(genome_ls, dtype=np.uint32); gl = gla.ctypes.data_as(C.POINTER(C.c_uint32))
    return lib().sw_vector(g.ctypes.data_as(C.POINTER(C.c_uint32)), goff, glen, r.ctypes.data_as(C.POINTER(C.c_uint32)), rlen, gl, initbp, bool(is_rna))


def sw_gapless_setup(match, mismatch, reset_stats=True):
    _check(lib().sw_gapless_setup(match, mismatch, reset_stats), "sw_gapless_setup")


def sw_gapless_batch(genome_words, genome_woff, glen, reads_words, rlen, g_idx, r_idx, genome_ls=None, initbp=None) -> np.ndarray:
    """n calls of the ungapped filter (ref: sw-gapless.c:57-117): call i's bitfield starts at word genome_woff[i]; colour space: genome_ls + initbp"""
    L = lib()
    g = np.ascontiguousarray(genome_words, dtype=np.uint32); r = np.ascontiguousarray(reads_words, dtype=np.uint32)
    n = r.shape[0]
    u32 = lambda a: a.ctypes.data_as(C.POINTER(C.c_uint32)); ip = lambda a: a.ctypes.data_as(C.POINTER(C.c_int))
    wo = np.ascontiguousarray(genome_woff, dtype=np.int64); out = np.zeros(n, dtype=np.int32)
    gn, rl, gi, ri = (np.ascontiguousarray(a, dtype=np.int32) for a in (glen, rlen, g_idx, r_idx))
    gl = np.ascontiguousarray(genome_ls, dtype=np.uint32) if genome_ls is not None else None
    ib = np.ascontiguousarray(initbp, dtype=np.int32) if initbp is not None else None
    _check(L.gm_sw_gapless_batch(n, u32(g), u32(gl) if gl is not None else None, g.size if gl is None else min(g.size, gl.size), wo.ctypes.data_as(C.POINTER(C.c_int64)), ip(gn),
                                 u32(r), r.shape[1], ip(rl), ip(gi), ip(ri), ip(ib) if ib is not None else None, ip(out)), "gm_sw_gapless_batch")
    return out


def sw_gapless(genome_words, glen, read_words, rlen, g_idx, r_idx, genome_ls=None, init_bp=-1) -> int:
    g = np.ascontiguousarray(genome_words, dtype=np.uint32); r = np.ascontiguousarray(read_words, dtype=np.uint32)
    gl = None
    if genome_ls is not None:
        gla = np.ascontiguousarray(genome_ls, dtype=np.uint32); gl = gla.ctypes.data_as(C.POINTER(C.c_uint32))
    return lib().sw_gapless(g.ctypes.data_as(C.POINTER(C.c_uint32)), glen, r.ctypes.data_as(C.POINTER(C.c_uint32)), rlen, g_idx, r_idx, gl, init_bp, False)


def sw_full_ls_setup(dblen, qrlen, a_gap_open, a_gap_ext, b_gap_open, b_gap_ext, match, mismatch, reset_stats=True, anchor_width=8):
    _check(lib().sw_full_ls_setup(dblen, qrlen, a_gap_open, a_gap_ext, b_gap_open, b_gap_ext, match, mismatch, reset_stats, anchor_width), "sw_full_ls_setup")


def sw_full_ls(genome_words, goff, glen, read_words, rlen, anchor, revcmpl=False, threshscore=0, maxscore=0, local_alignment=False):
    """anchor = (x, y, length, width) or None (the threshold band); returns (fields dict, dbalign, qralign)."""
    L = lib()
    g = np.ascontiguousarray(genome_words, dtype=np.uint32); r = np.ascontiguousarray(read_words, dtype=np.uint32)
    a = Anchor(anchor[0], anchor[1], anchor[2], anchor[3], 1, 0, 0) if anchor is not None else None
    s = SwFullResults()
    L.sw_full_ls(g.ctypes.data_as(C.POINTER(C.c_uint32)), goff, glen, r.ctypes.data_as(C.POINTER(C.c_uint32)), rlen, int(threshscore), int(maxscore),
                 C.byref(s), bool(revcmpl), C.byref(a) if a is not None else None, 1 if a is not None else 0, 1 if local_alignment else 0)
    db = C.string_at(s.dbalign).decode() if s.dbalign else ""
    qr = C.string_at(s.qralign).decode() if s.qralign else ""
    L.gm_free(s.dbalign); L.gm_free(s.qralign)
    return {n: getattr(s, n) for n, _ in SwFullResults._fields_[:9]}, db, qr


# ---- colour space S1/S2/S3 at the kernel seams (the read pipeline is Session.map_reads_cs*) ----
def sw_vector_batch_cs(genome_cs, genome_ls, g_off, glen, reads_words, rlen, initbp):
    """colour-space vector filter: genome_cs / genome_ls = colour and letter bitfields of the same contig; one initial base per read"""
    L = lib()
    gc = np.ascontiguousarray(genome_cs, dtype=np.uint32); gl = np.ascontiguousarray(genome_ls, dtype=np.uint32)
    r = np.ascontiguousarray(reads_words, dtype=np.uint32)
    n = r.shape[0]
    go = np.ascontiguousarray(g_off, dtype=np.int64); gn = np.ascontiguousarray(glen, dtype=np.int32); rl = np.ascontiguousarray(rlen, dtype=np.int32)
    ib = np.ascontiguousarray(initbp, dtype=np.int32); out = np.zeros(n, dtype=np.int32)
    ip = lambda a: a.ctypes.data_as(C.POINTER(C.c_int))
    _check(L.gm_sw_vector_batch_cs(n, gc.ctypes.data_as(C.POINTER(C.c_uint32)), gl.ctypes.data_as(C.POINTER(C.c_uint32)), min(gc.size, gl.size),
                                   go.ctypes.data_as(C.POINTER(C.c_int64)), ip(gn), r.ctypes.data_as(C.POINTER(C.c_uint32)), r.shape[1], ip(rl), ip(ib), ip(out)),
           "gm_sw_vector_batch_cs")
    return out


def preprocess_read(params, seq: bytes, qual: "bytes | None" = None, qual_delta: int = 64, mate: int = 0):
    """gm_preprocess_read_text (host only): what the file entries do to one read before mapping it -> (seq, qual, dropped)"""
    sb = C.create_string_buffer(seq, len(seq) + 1); qb = None if qual is None else C.create_string_buffer(qual, len(qual) + 1); d = C.c_int(0)
    _check(lib().gm_preprocess_read_text(C.byref(params), int(mate), sb, qb, int(qual_delta), C.byref(d)), "gm_preprocess_read_text")
    return sb.value, (None if qb is None else qb.value), bool(d.value)


def sw_full_cs_setup(dblen, qrlen, a_gap_open, a_gap_ext, b_gap_open, b_gap_ext, match, mismatch, xover, reset_stats=True, anchor_width=8, indel_taboo_len=0):
    _check(lib().sw_full_cs_setup(dblen, qrlen, a_gap_open, a_gap_ext, b_gap_open, b_gap_ext, match, mismatch, xover, reset_stats, anchor_width, indel_taboo_len),
           "sw_full_cs_setup")


def sw_full_cs(genome_ls, goff, glen, read_words, rlen, initbp, thresh, anchor, revcmpl=False, local=False, xover=None, is_rna=False):
    """anchor = (x, y, length, width); local: the reference's local_alignment argument; xover: the reference's crossover_score argument (one int per read position, what
    gmapper hands over for a read with quality values, ref: mapping.c:375-379) or None; returns (fields dict incl. crossovers, dbalign, qralign)."""
    L = lib()
    g = np.ascontiguousarray(genome_ls, dtype=np.uint32); r = np.ascontiguousarray(read_words, dtype=np.uint32)
    a = Anchor(anchor[0], anchor[1], anchor[2], anchor[3], 1, 0, 0)
    s = SwFullResults()
    xs = None if xover is None else np.ascontiguousarray(xover, dtype=np.int32)
    L.sw_full_cs(g.ctypes.data_as(C.POINTER(C.c_uint32)), goff, glen, r.ctypes.data_as(C.POINTER(C.c_uint32)), rlen, initbp, thresh,
                 C.byref(s), bool(revcmpl), bool(is_rna), C.byref(a), 1, 1 if local else 0, None if xs is None else xs.ctypes.data_as(C.POINTER(C.c_int)))
    db = C.string_at(s.dbalign).decode() if s.dbalign else ""
    qr = C.string_at(s.qralign).decode() if s.qralign else ""
    if s.dbalign: L.gm_free(s.dbalign)
    if s.qralign: L.gm_free(s.qralign)
    f = {n: getattr(s, n) for n in ("score", "read_start", "rmapped", "genome_start", "gmapped", "matches", "mismatches", "insertions", "deletions", "crossovers")}
    return f, db, qr


def seam_stats(which):
    """(invocations, cells, seconds) of this thread's sw_vector / sw_full_ls / sw_full_cs / post_sw calls (ref: the *_stats functions gmapper.c:734-745 reads)."""
    L = lib()
    if which == "sw_gapless":                                 # (invocations, cells, ns): three integers, as sw-gapless.h:12 declares them
        inv, cells, ticks = C.c_uint64(0), C.c_uint64(0), C.c_uint64(0)
        L.sw_gapless_stats.restype = None
        L.sw_gapless_stats(C.byref(inv), C.byref(cells), C.byref(ticks))
        return inv.value, cells.value, ticks.value
    inv, cells, secs = C.c_uint64(0), C.c_uint64(0), C.c_double(0)
    fn = getattr(L, {"sw_vector": "sw_vector_stats", "sw_full_ls": "sw_full_ls_stats", "sw_full_cs": "sw_full_cs_stats", "post_sw": "post_sw_stats"}[which])
    fn.restype = None if which != "post_sw" else C.c_int
    fn(C.byref(inv), C.byref(cells), C.byref(secs))
    return inv.value, cells.value, secs.value


def sequence_to_bitfield(text, colour_space: bool = False):
    """fasta_sequence_to_bitfield (ref: common/fasta.c:609-673) through the library: returns (uint32 words, primer letter code or None)."""
    t = text if isinstance(text, bytes) else text.encode()
    n = len(t) - (1 if colour_space else 0)
    words = np.zeros(max(1, (n + 7) // 8), dtype=np.uint32); b = C.c_int(0)
    _check(lib().gm_sequence_to_bitfield(int(colour_space), t, len(t), words.ctypes.data_as(C.POINTER(C.c_uint32)), C.byref(b)), "gm_sequence_to_bitfield")
    return words, (b.value if colour_space else None)
