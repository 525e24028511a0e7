// gm_common.h -- shared device/host definitions for libgmapper_hip.so (gfx950 only).
// "ref:" citations are file:line in SHRiMP 2.2.3.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <vector>
#include "../../include/gmapper_hip.h"

#define GM_MAX_SEEDS 16
#define GM_WAVE 64

// ---- error plumbing -------------------------------------------------------------------------
void gm_set_error(const char* fmt, ...);
#define GM_HIP(call)                                                                          \
  do {                                                                                        \
    hipError_t e_ = (call);                                                                   \
    if (e_ != hipSuccess) {                                                                   \
      gm_set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
      return GM_E_NODEVICE;                                                                   \
    }                                                                                         \
  } while (0)

// ---- device-side view of the resident index --------------------------------------------------
// Layout in HBM (DESIGN.md "Index layout"):
//   genome : 4-bit codes, all contigs concatenated in *global* coordinates (position p is nibble p%8
//            of word p/8); the reference's per-contig bitfields are re-packed once on the host.
//   pos[sn]: all k-mer start positions of seed sn, sorted by (mapidx, position): the reference's
//            genomemap[sn][mapidx][] lists back to back, each ascending (ref: genome.c:1156-1163).
//   dir[sn]: K*S+1 offsets into pos[sn], K = 4^weight, S = n_slabs: dir[k*S+s] = first entry of
//            list k whose position is >= s << slab_bits.  List k = [dir[k*S], dir[(k+1)*S]);
//            its slice inside slab s = [dir[k*S+s], dir[k*S+s+1]).  S = 1 is plain CSR.
struct GmSeedDev {
  uint64_t mask;       // ref: seed_type.mask[0], LSB = most recent base (gmapper-definitions.h:59-63)
  int span, weight;
  const uint32_t* dir;
  const uint32_t* pos;
  const uint32_t* bkt;   // optional [K][16]: list length + first 15 positions (small genomes; see gm_index.hip)
  const uint32_t* sdir;  // optional [K+1]: strip list of k = spos[sdir[k] .. sdir[k+1]): list k's entries in the first region_overlap bases
  const uint32_t* spos;  //   of a region > 0, in list order (derived on the device, gm_lookup5.hip)
  uint32_t n_pos;
};

// Paired -n 3 (mate-pair region counts 2 / 3, ref: mapping.c:545-608,733-742,1080-1093): what the lookup and the window kernels need to know about the MATE.
// A row holds, sorted, the regions one read-strand marked twice or more (its RG_GET_HAS_2 set); a region R of the other mate's opposite strand then has
// RG_GET_MP_CNT(R) >= 2 exactly when some row entry lies in [R + dmin, R + dmax].  cnt[rs] > cap: the row did not fit (the consumers count the item, the host refuses).
#define GM_MP_CAP 8192              // (150-base reads on 3 Gbp: 72 k list entries per read-strand over 1.46 M regions mark ~1 800 regions twice by chance alone)
#define GM_MP_FLAG 0x80000000u
struct GmMpDev {
  int mode;                           // 0: off; else the lookup's generic kernel runs in one of these modes (the window kernel reads rows / cnt / dmin / dmax whatever the mode):
                                      // 1: it only LISTS the regions each read-strand marked twice (out_rows / out_cnt);
                                      // 2: it keeps, besides the entries of regions marked twice, those of regions the mate reaches -- count_main >= 2 || count_mp >= 2,
                                      //    use_mp_region_counts == 2 (-n 3);
                                      // 3: it only FLAGS (GM_MP_FLAG in the row word) the mate's row entries that some region this read-strand marked reaches: for a region
                                      //    the mate marked twice that is count_mp >= 1;
                                      // 4: it keeps the entries of regions with count_mp >= 1 && count_main + count_mp >= 3, use_mp_region_counts == 3 (-n 3 without half-paired):
                                      //    the mate reaches the region (count_mp == 2), or the region is marked twice and its own row entry carries the flag
                                      // 5: it keeps the entries of regions marked twice that the mate reaches -- count_main >= 2 && count_mp >= 2, use_mp_region_counts == 1
                                      //    (-n 4 without half-paired): the exact path for sub-batches in which k_mp_filter's LDS tiers did not hold every pair
  int dmin[2], dmax[2];               // this mate's region deltas per strand (delta_region_min / _max, ref: mapping.c:2422-2430)
  int mate_dmin[2], mate_dmax[2];     // the mate's, per ITS strand (mode 3 asks the mate's question: in the col-fw / col-bw pair modes the two are not each other's negation)
  uint32_t* rows; const uint32_t* cnt;           // the mate's rows: read-strand rs of this mate looks at row rs ^ 1 (same pair, other strand)
  uint32_t* out_rows; uint32_t* out_cnt;         // this mate's own rows (written in mode 1, read in mode 4)
};
// some row entry in [lo, hi]?  (row sorted ascending; the flag bit is not part of the value).  *first: index of the first such entry
__device__ __forceinline__ bool gm_mp_reach(const uint32_t* row, uint32_t n, long long lo, long long hi, uint32_t* first = nullptr) {
  if (hi < 0 || n == 0) return false;
  if (lo < 0) lo = 0;
  uint32_t a = 0, z = n;
  while (a < z) { const uint32_t m = (a + z) >> 1; if ((long long)(row[m] & ~GM_MP_FLAG) < lo) a = m + 1; else z = m; }
  if (first) *first = a;
  return a < n && (long long)(row[a] & ~GM_MP_FLAG) <= hi;
}

struct GmIndexDev {
  const uint32_t* genome;
  const uint32_t* genome_cs;          // colour space only: colour p = lstocs(letter p-1, letter p), 'T' before a contig's first letter (ref: fasta.c:586-606)
  int colour;                         // 1 in colour space (== min_kmer_pos, ref: gmapper.c:477-480)
  int hflag;                          // -H: lists are keyed by kmer_to_mapidx_hash, 4^12 of them per seed (ref: gmapper.h:323-336)
  int cs_flip;                        // colour space, a mate the pair mode reverses (read_reverse, ref: gmapper.c:174-185): the read keeps its colours, its strand LABELS
                                      // swap -- strand label st stands for strand st ^ cs_flip of the read as sequenced, which is then the read's input strand
  int no_region_counts;               // set per call: the lookup keeps EVERY list entry -- unpaired -n 1, paired -n 2 (use_region_counts off, ref: gmapper.c:2610-2616,2652-2657)
  GmMpDev mp;                         // set per call (paired -n 3)
  // RNA sequences (uracil and no thymine, ref: fasta.c:528-542).  A contig's own flag decides its reverse complement (A <-> U) and its colour translation (U read as T),
  // ref: genome.c:1107-1118; the LAST contig's flag is genome_is_rna, which sw_vector / sw_gapless / sw_full_cs get (genome.c:1063-1064; mapping.c:375-388,1318-1327);
  // a letter-space read's own flag decides its reverse complement (gmapper.c:487).  Null pointers: no contig / no read of this call is RNA.
  const uint8_t* contig_rna;          // [n_contigs] or null
  int genome_is_rna;
  const uint8_t* read_rna;            // set per call: [reads of the sub-batch] or null (letter space only)
  uint64_t total_len;                 // sum of contig lengths (< 2^32)
  int n_contigs;
  const uint32_t* contig_off;         // [n_contigs+1] global offsets (ref: contig_offsets[])
  int n_seeds, min_seed_span, max_seed_span;
  int slab_bits, n_slabs;
  int region_bits, region_overlap;
  uint32_t list_cutoff;
  GmSeedDev seed[GM_MAX_SEEDS];
};

// ---- per-batch scoring constants (host computes every double-derived threshold) --------------
struct GmScoreDev {
  int match, mismatch;
  int a_go, a_ge, b_go, b_ge;         // positive penalties (= -score), as sw_vector_setup stores them
  int anchor_width;
  int match_mode, min_matches;
  int skip_strands;                   // bit st set: strand st gets no anchor list (-C / -F, ref: mapping.c:879-880)
  int num_tmp_outputs;
  int tiebreak_rev;
  int hash_filter_calls;
  int gapless;                        // -U: ungapped pass-1 filter, one window per anchor (ref: gmapper.c:2057-2062)
  int local;                          // Gflag off: sw_full_ls in local mode (ref: gmapper.c:2303-2305)
  double wgen_thr_frac;               // window_gen_threshold/100.0 (or <0: absolute = -value)
  double vect_thr_frac, full_thr_frac;
  int wgen_abs, vect_abs, full_abs;   // absolute thresholds when the fractions are negative
};

// candidate window ("read_hit", ref: gmapper-definitions.h:131-160) as it lives in HBM
struct GmHit {
  uint32_t g_off;        // contig-relative window start on the + strand (g_off_pos_strand)
  int32_t  ax, ay;       // anchor box relative to the window (ref: struct anchor x,y)
  int32_t  alen, awidth; // anchor length / width
  int32_t  score_window_gen;
  int32_t  score_vector; // -1 = not scored
  int32_t  pct_score_vector;
  uint16_t cn, w_len;
  uint16_t matches, flags;
};

// result of pass 2 for one selected hit (what the host needs to finish A18/A19/A21)
#define GM_MAX_OPS 2048
struct GmFullRes {
  int32_t read_idx;
  int16_t st, gen_st;            // after reverse_hit (ref: mapping.c:254-263)
  uint32_t cn; uint32_t g_off;   // g_off on the gen_st strand
  int32_t w_len;
  int32_t score_vector;          // re-scored in pass 2 (ref: mapping.c:386-388)
  int32_t score_max, matches, score_window_gen;
  int32_t score;                 // sw_full_ls score (0 = below threshold / not run)
  int32_t read_start, rmapped, genome_start, gmapped;
  int32_t n_match, n_mismatch, n_ins, n_del;
  int32_t n_ops; uint32_t ops_off;   // ops bytes ('M','I','D') in the batch's op pool
  int32_t sort_idx;              // paired mode: position in the read's (strand 0, strand 1) window list (ref: mapping.c:2545-2552)
  uint32_t hit_slot;             // index of the window in the batch's GmHit array
  int32_t n_xover;               // colour space: crossovers on the sw_full_cs path (ref: sw-full-cs.c:929-932)
};

// complement_base as sixteen nibbles (ref: util.h:125-151): entry c = the complement of code c; in an RNA sequence the complement of A is U (code 4), not T
__host__ __device__ __forceinline__ uint64_t gm_cmpl_tab(bool rna) { return 0xFBCDE56879A00123ull + (rna ? 1ull : 0ull); }

#ifdef GM_NO_READ_RNA
#define GM_READ_RNA(ix, rd) false
#else
#define GM_READ_RNA(ix, rd) ((ix).read_rna && (ix).read_rna[rd])      // the read's own RNA flag (letter space), see GmIndexDev
#endif
#define GM_SEAM_RNA 0x100             // single-call seams (sw_vector / sw_gapless / sw_full_cs): is_rna rides in bit 8 of the primer-letter word handed to the kernel

// 4-bit code i of strand st of a packed read.  Letter space: strand 1 is the reverse complement (ref: util.c:540-596).
// Colour space: strand 1 holds the colours in reverse order, rc[i] = read[len - i] for i >= 1 (ref: util.c:600-617); rc[0]
// involves the primer letter and is never used: k-mers start at colour 1 and the alignment kernels only see strand 0.
__device__ __forceinline__ uint32_t gm_read_code(const uint32_t* __restrict__ rw, int read_len, int st, int colour, int i, bool rna = false) {
  if (colour) {
    if (st && i == 0) return 15u;
    const int src = st ? (read_len - i) : i;
    return (rw[src >> 3] >> ((src & 7) * 4)) & 0xf;
  }
  const int src = st ? (read_len - 1 - i) : i;
  uint32_t c = (rw[src >> 3] >> ((src & 7) * 4)) & 0xf;
  if (st) { const uint64_t cm = gm_cmpl_tab(rna); c = (uint32_t)(cm >> (c * 4)) & 0xf; }   // complement_base, ref: util.h:125-151
  return c;
}

// Map index of the k-mer whose bases are kmer[0 .. span) (one 4-bit code per byte).  Default: kmer_to_mapidx_orig (ref: gmapper.h:349-368):
// walk the mask from its LSB (= the most recent base), every 1-bit appends the base's low 2 bits.  -H: kmer_to_mapidx_hash (ref: :323-336):
// the 4-bit window words (newest base in nibble 0 of word 0), masked to the seed's 1-positions (seed_hash_mask, seeds.c:83-102), folded through
// hash() over BPTO32BW(max_seed_span) words, low 24 bits.
#define GM_HASH_TABLE_POWER 12
__host__ __device__ __forceinline__ uint32_t gm_hash32(uint32_t a) {   // ref: gmapper.h:309-319
  a = (a + 0x7ed55d16u) + (a << 12); a = (a ^ 0xc761c23cu) ^ (a >> 19); a = (a + 0x165667b1u) + (a << 5);
  a = (a + 0xd3a2646cu) ^ (a << 9); a = (a + 0xfd7046c5u) + (a << 3); a = (a ^ 0xb55a4f09u) ^ (a >> 16);
  return a;
}
__device__ __forceinline__ uint32_t gm_mapidx(const GmIndexDev& ix, uint64_t mask, int span, const uint8_t* kmer) {
  uint32_t mapidx = 0;
  if (!ix.hflag) {
    for (int t = 0; t < span; t++) if ((mask >> t) & 1) mapidx = (mapidx << 2) | (kmer[span - 1 - t] & 3u);
    return mapidx;
  }
  const int nw = (ix.max_seed_span + 7) >> 3;
  for (int w = 0; w < nw; w++) {
    uint32_t word = 0;
    for (int t = 0; t < 8; t++) { const int age = 8 * w + t; if (age < span && ((mask >> age) & 1)) word |= (uint32_t)(kmer[span - 1 - age] & 0xfu) << (4 * t); }
    mapidx = gm_hash32(word ^ mapidx);
  }
  return mapidx & ((1u << (2 * GM_HASH_TABLE_POWER)) - 1u);
}

static inline int gm_ceil_div(long long a, long long b) { return (int)((a + b - 1) / b); }
