// gm_oracle.hpp -- CPU restatement of SHRiMP2 gmapper's letter-space hot path.
//
// TEST INFRASTRUCTURE ONLY.  Only tests/, __graft_entry__.smoke() and bench.py's
// cpu_baseline leg may build, link or call this.  The product path (shrimp_amd/csrc,
// libgmapper_hip.so) never includes or links anything under oracle/.
//
// Parity pinning: this restatement is checked byte-for-byte against the reference binary
// built from /root/reference by oracle/Makefile.ref (tests/golden/*.sam.gz were generated
// by tools/make_golden.py from that binary) and against known-answer triples produced by the
// reference's own sw_vector()/sw_full_ls() (oracle/_ref/libref_sw.so).
//
// It is plain scalar C++ (no SSE); every function cites the reference file:line it follows.
// All "file:line" citations are relative to /root/reference.
#pragma once
#include <algorithm>
#include <cassert>
#include <climits>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <vector>
#include <omp.h>
#include <omp.h>

namespace gmo {

typedef long long llint;

// ---------------------------------------------------------------------------------------------
// 4-bit alphabet and bitfields (common/fasta.h:26-42, common/util.h:41-42, common/fasta.c:28-57)
// ---------------------------------------------------------------------------------------------
static inline int EXTRACT(const uint32_t* g, llint i) { return (g[i / 8] >> (4 * (i % 8))) & 0xf; }
static inline int BPTO32BW(int x) { return (x + 7) / 8; }

static inline int char_to_code_ls(unsigned char c) {  // fasta_open translate table, common/fasta.c:164-198
  switch (c) {
    case 'A': case 'a': return 0;  case 'C': case 'c': return 1;
    case 'G': case 'g': return 2;  case 'T': case 't': return 3;
    case 'U': case 'u': return 4;  case 'M': case 'm': return 5;
    case 'R': case 'r': return 6;  case 'W': case 'w': return 7;
    case 'S': case 's': return 8;  case 'Y': case 'y': return 9;
    case 'K': case 'k': return 10; case 'V': case 'v': return 11;
    case 'H': case 'h': return 12; case 'D': case 'd': return 13;
    case 'B': case 'b': return 14; case 'N': case 'n': case '.': case 'X': case 'x': return 15;
    default: return -1;
  }
}
static const char LSTRANS[17] = "ACGTUMRWSYKVHDBN";  // base_translate, common/fasta.c:689-690

static inline int complement_base(int b, bool is_rna = false) {  // common/util.h:125-151: in an RNA sequence the complement of A is U
  static const int cmpl[16] = {3, 2, 1, 0, 0, 10, 9, 7, 8, 6, 5, 14, 13, 12, 11, 15};
  return (is_rna && cmpl[b] == 3) ? 4 : cmpl[b];
}
// "is RNA": uracil and no thymine (common/fasta.c:528-542, on the text; the same test on the 4-bit codes)
static inline bool codes_are_rna(const uint8_t* codes, size_t n) {
  bool u = false, t = false;
  for (size_t i = 0; i < n; i++) { u |= codes[i] == 4; t |= codes[i] == 3; }
  return u && !t;
}

static inline std::vector<uint32_t> pack_codes(const uint8_t* codes, size_t n) {  // fasta_sequence_to_bitfield, fasta.c:609-673
  std::vector<uint32_t> bf(BPTO32BW((int)std::min<size_t>(n, INT_MAX - 8)) > 0 ? (n + 7) / 8 : 0, 0);
  for (size_t i = 0; i < n; i++) bf[i / 8] |= (uint32_t)(codes[i] & 0xf) << (4 * (i % 8));
  return bf;
}

// reverse_complement_read_ls (common/util.c:540-596): rc[i] = cmpl(read[len-1-i]), unused nibbles 0.
static inline std::vector<uint32_t> revcomp_ls(const uint32_t* read, size_t len, bool is_rna = false) {
  std::vector<uint32_t> rc((len + 7) / 8, 0);
  for (size_t i = 0; i < len; i++) {
    int b = complement_base(EXTRACT(read, (llint)(len - 1 - i)), is_rna);
    rc[i / 8] |= (uint32_t)b << (4 * (i % 8));
  }
  return rc;
}

// ---------------------------------------------------------------------------------------------
// Parameters = the reference's globals at their letter-space defaults
// (gmapper/gmapper-defaults.h:21-68, gmapper/gmapper.h:47-127)
// ---------------------------------------------------------------------------------------------
struct Seed { uint64_t mask; int span; int weight; };

struct Params {
  int match_score = 10, mismatch_score = -15;
  int a_gap_open_score = -33, a_gap_extend_score = -7;
  int b_gap_open_score = -33, b_gap_extend_score = -3;
  double window_len = 140.0, window_overlap = 90.0;
  double window_gen_threshold = 55.0, sw_vect_threshold = 50.0, sw_full_threshold = 50.0;
  int match_mode = 2, num_outputs = 10, num_tmp_outputs = 30, anchor_width = 8;
  int region_bits = 11, region_overlap = 50;
  uint32_t list_cutoff = 4294967295u;
  bool hash_filter_calls = true;  // -Z turns this off
  bool use_sanger_qvs = true; int qual_vector_offset = 0;      // gmapper.h:78-79
  bool Qflag = false; int qual_delta = 64;   // FASTQ input: QUAL strings travel to the SAM output (output.c:539-570; gmapper-defaults.h:41)
  bool Hflag = false;             // -H: hashed seeds (kmer_to_mapidx_hash, 4^12 lists per seed whatever its weight; gmapper.h:323-336)
  bool gapless = false;           // -U: ungapped filter (gapless_sw; gmapper.c:2057-2062 also sets anchor_width 0, gap opens -255, no f1 cache)
  bool Tflag = true, Gflag = true, compute_mapping_qualities = true;
  bool strata = false;
  bool single_best_mapping = false, all_contigs = false, improper_mappings = true;   // --single-best-mapping, --all-contigs, --no-improper-mappings (gmapper.c:2252-2268)
  bool Fflag = true, Cflag = true;  // -F / -C: positive / negative strand only (gmapper.c:1979-1992,2444-2451); both in paired mode
  int max_alignments = 0;
  bool sam_unaligned = false;
  int longest_read_len = 1000;
  double score_alpha = 0, score_beta = 0;
  std::vector<Seed> seeds;
  int max_seed_span = 0, min_seed_span = 64;
  // pairing (gmapper/gmapper.h:138-151, gmapper-defaults.h:25-29)
  int pair_mode = 0;               // 0 none, 1 opp-in, 2 opp-out, 3 col-fw, 4 col-bw
  int min_insert_size = 0, max_insert_size = 1000;
  double insert_size_mean = 200, insert_size_stddev = 100;
  bool half_paired = true;
  int mp_match_mode = 4;            // the paired option set's match mode (gmapper-defaults.h: DEF_MATCH_MODE_PAIRED); with half_paired it picks use_mp_region_counts
  // colour space (gmapper/gmapper.h:55-57,119; gmapper-defaults.h:52-58): set by set_colour_space()
  bool colour = false;
  int crossover_score = -20, indel_taboo_len = 0;
  double pr_xover = 0.03, pr_mismatch = .01, pr_del_open = 0, pr_del_extend = 0, pr_ins_open = 0, pr_ins_extend = 0;
};

#define GMO_IS_ABSOLUTE(x) ((x) < 0)
// common/util.h:53 -- keep the macro's expression order (base is an int expression at every call site)
#define GMO_ABS_OR_PCT(x, base) (GMO_IS_ABSOLUTE(x) ? -(x) : (base) * ((x) / 100.0))

// add_spaced_seed (gmapper/seeds.c:9-43): first char -> highest bit, last char -> bit 0
static inline void add_spaced_seed(Params& P, const char* s) {
  Seed sd; sd.mask = 0; sd.span = (int)strlen(s); sd.weight = 0;
  for (int i = 0; i < sd.span; i++) { sd.mask = (sd.mask << 1) | (s[i] == '1'); sd.weight += (s[i] == '1'); }
  P.seeds.push_back(sd);
  P.max_seed_span = std::max(P.max_seed_span, sd.span);
  P.min_seed_span = std::min(P.min_seed_span, sd.span);
}
// load_default_seeds(0) in letter space (gmapper/seeds.c:53-80; gmapper-defaults.h:212-227): 3 seeds of weight 12
static inline void load_default_seeds(Params& P) {
  add_spaced_seed(P, "11110111101111");
  add_spaced_seed(P, "1111011100100001111");
  add_spaced_seed(P, "1111000011001101111");
}
// score -> probability derivation (gmapper/gmapper.c:2557-2572)
static inline void derive_score_probs(Params& P) {
  if (P.colour) {   // CS: pr_xover => alpha => pr_mismatch => rest
    P.score_alpha = (double)P.crossover_score / (log(P.pr_xover / 3) / log(2.0));
    P.pr_mismatch = 1.0 / (1.0 + 1.0 / 3.0 * pow(2.0, ((double)P.match_score - (double)P.mismatch_score) / P.score_alpha));
  } else {          // LS: pr_mismatch => alpha => rest
    P.pr_mismatch = .01;
    P.score_alpha = ((double)P.match_score - (double)P.mismatch_score) / (log((1 - P.pr_mismatch) / (P.pr_mismatch / 3.0)) / log(2.0));
  }
  P.score_beta = (double)P.match_score - 2 * P.score_alpha - P.score_alpha * log(1 - P.pr_mismatch) / log(2.0);
  P.pr_del_open = pow(2.0, (double)P.a_gap_open_score / P.score_alpha);
  P.pr_ins_open = pow(2.0, (double)P.b_gap_open_score / P.score_alpha);
  P.pr_del_extend = pow(2.0, (double)P.a_gap_extend_score / P.score_alpha);
  P.pr_ins_extend = pow(2.0, ((double)P.b_gap_extend_score - P.score_beta) / P.score_alpha);
}
// the gmapper-cs binary's defaults (gmapper.c:1748-1755; gmapper-defaults.h:52-58,64-66): same seeds, same gap scores
static inline void set_colour_space(Params& P) {
  P.colour = true;
  P.match_score = 10; P.mismatch_score = -24; P.crossover_score = -20;
  P.a_gap_open_score = -33; P.a_gap_extend_score = -7; P.b_gap_open_score = -33; P.b_gap_extend_score = -3;
  P.sw_vect_threshold = 47.0; P.sw_full_threshold = 50.0;
  derive_score_probs(P);
}

// lstocs / cstols (common/util.h:157-205).  With is_rna a U counts as T going in, and cstols hands back U where it would hand back T.
static inline int lstocs(int first_letter, int second_letter, bool is_rna = false) {
  static const int colourmat[4][4] = {{0, 1, 2, 3}, {1, 0, 3, 2}, {2, 3, 0, 1}, {3, 2, 1, 0}};
  if (is_rna) { if (first_letter == 4) first_letter = 3; if (second_letter == 4) second_letter = 3; }
  if (first_letter > 3 || second_letter > 3) return 15;     // anything non-{A,C,G,T} -> N
  return colourmat[first_letter][second_letter];
}
static inline int cstols(int first_letter, int colour, bool is_rna = false) {
  if (first_letter == 15 || !(colour >= 0 && colour <= 3)) return 15;
  if (is_rna && first_letter == 4) first_letter = 3;
  const int ret = (first_letter % 2 == 0) ? ((4 + first_letter + colour) % 4) : ((4 + first_letter - colour) % 4);
  return (is_rna && ret == 3) ? 4 : ret;
}

// ---------------------------------------------------------------------------------------------
// Genome + index (gmapper/genome.c:1012-1182; gmapper/gmapper.h:349-368)
// ---------------------------------------------------------------------------------------------
struct Genome {
  std::vector<std::string> names;
  std::vector<uint32_t> len, offsets;            // genome_len[], contig_offsets[]
  std::vector<std::vector<uint32_t>> fwd, rc;    // genome_contigs[], genome_contigs_rc[]
  std::vector<std::vector<uint32_t>> cs_fwd, cs_rc;   // genome_cs_contigs[], genome_cs_contigs_rc[] (colour space only)
  bool colour = false;
  bool is_rna = false;                           // genome_is_rna: the flag of the LAST contig read (genome.c:1063-1064); the SW calls get this one
  std::vector<uint8_t> contig_rna;               // each contig's own flag: its reverse complement and colour translation use that (genome.c:1107-1118)
  int num_contigs() const { return (int)len.size(); }
  // bitfield_to_colourspace (common/fasta.c:586-606): colour i = lstocs(letter i-1, letter i), a T before the first letter
  static std::vector<uint32_t> to_colourspace(const uint32_t* src, size_t n, bool rna = false) {
    std::vector<uint32_t> dst((n + 7) / 8, 0u);
    int lastbp = 3;
    for (size_t i = 0; i < n; i++) { int a = EXTRACT(src, (llint)i); dst[i / 8] |= (uint32_t)lstocs(lastbp, a, rna) << (4 * (i % 8)); lastbp = a; }
    return dst;
  }
  void add_contig(const std::string& name, const uint8_t* codes, size_t n) {
    uint32_t off = offsets.empty() ? 0u : offsets.back() + len.back();
    names.push_back(name); offsets.push_back(off); len.push_back((uint32_t)n);
    const bool rna = codes_are_rna(codes, n);
    contig_rna.push_back(rna ? 1 : 0); is_rna = rna;
    fwd.push_back(pack_codes(codes, n));
    rc.push_back(revcomp_ls(fwd.back().data(), n, rna));
    if (colour) {   // genome.c:1108-1119
      cs_fwd.push_back(to_colourspace(fwd.back().data(), n, rna));
      cs_rc.push_back(to_colourspace(rc.back().data(), n, rna));
    }
  }
  // the sequence the seed index is built over (genome.c:1126-1136: the colour translation in colour space)
  const uint32_t* index_seq(int cn) const { return colour ? cs_fwd[cn].data() : fwd[cn].data(); }
};

// kmer_to_mapidx_orig (gmapper/gmapper.h:349-368) for the k-mer whose most recent base is seq[end]
template <class GetBase>
static inline uint32_t kmer_to_mapidx(const Seed& sd, GetBase base_at, llint end) {
  uint64_t a = sd.mask; uint32_t mapidx = 0; int i = 0;
  do {
    if (a & 1) { mapidx <<= 2; mapidx |= (uint32_t)(base_at(end - i) & 0x3); }
    a >>= 1; i++;
  } while (a != 0);
  return mapidx;
}

// -H: hash() and kmer_to_mapidx_hash (gmapper/gmapper.h:309-336) with seed_hash_mask (seeds.c:83-102).  The reference hashes the
// 4-bit window words (newest base in nibble 0 of word 0) masked to the seed's 1-positions, over BPTO32BW(max_seed_span) words.
static const int HASH_TABLE_POWER = 12;
static inline uint32_t hash32(uint32_t a) {
  a = (a + 0x7ed55d16) + (a << 12); a = (a ^ 0xc761c23c) ^ (a >> 19); a = (a + 0x165667b1) + (a << 5);
  a = (a + 0xd3a2646c) ^ (a << 9); a = (a + 0xfd7046c5) + (a << 3); a = (a ^ 0xb55a4f09) ^ (a >> 16);
  return a;
}
template <class GetBase>
static inline uint32_t kmer_to_mapidx_hash(const Seed& sd, int max_seed_span, GetBase base_at, llint end) {
  uint32_t mapidx = 0;
  for (int w = 0; w < BPTO32BW(max_seed_span); w++) {
    uint32_t word = 0;
    for (int t = 0; t < 8; t++) { const int age = 8 * w + t; if (age < sd.span && ((sd.mask >> age) & 1)) word |= (uint32_t)(base_at(end - age) & 0xf) << (4 * t); }
    mapidx = hash32(word ^ mapidx);
  }
  return mapidx & ((1u << (2 * HASH_TABLE_POWER)) - 1u);
}
static inline int index_key_bits(const Params& P, const Seed& sd) { return P.Hflag ? 2 * HASH_TABLE_POWER : 2 * sd.weight; }   // genome.c:1034

struct Index {                    // genomemap / genomemap_len in CSR form (lists ascending, genome.c:1156-1163)
  std::vector<std::vector<uint32_t>> start;   // [sn][4^W + 1]  (total entries per seed < 2^32: positions are uint32)
  std::vector<std::vector<uint32_t>> pos;     // [sn][total]
  uint32_t list_len(int sn, uint32_t idx) const { return (uint32_t)(start[sn][idx + 1] - start[sn][idx]); }
  const uint32_t* list(int sn, uint32_t idx) const { return pos[sn].data() + start[sn][idx]; }
};

// map index of the k-mer ending at the newest base of a rolling 2-bit window `w` (newest base in
// bits 63:62, older bases below): identical to kmer_to_mapidx_orig because walking the mask from
// its LSB appends the newest base first (=> top of the index) -- see kmer_to_mapidx above.
static inline uint32_t mapidx_from_window(uint64_t w, const int* sel, int weight) {
  uint32_t m = 0;
  for (int k = 0; k < weight; k++) m = (m << 2) | (uint32_t)((w >> (62 - 2 * sel[k])) & 3u);
  return m;
}

static inline void build_index_seq(const Params& P, const Genome& G, Index& I) {
  int ns = (int)P.seeds.size();
  I.start.assign(ns, {}); I.pos.assign(ns, {});
#pragma omp parallel for schedule(dynamic, 1)
  for (int sn = 0; sn < ns; sn++) {
    const Seed& sd = P.seeds[sn];
    size_t cap = (size_t)1 << index_key_bits(P, sd);
    std::vector<uint32_t>& st = I.start[sn];
    st.assign(cap + 1, 0);
    int sel[64], nsel = 0;                       // ages (0 = newest base) of the mask's 1-bits, LSB first
    for (int t = 0; t < sd.span; t++) if ((sd.mask >> t) & 1) sel[nsel++] = t;
    for (int pass = 0; pass < 2; pass++) {
      std::vector<uint32_t> fill;
      if (pass == 1) {
        uint32_t acc = 0;
        for (size_t k = 0; k <= cap; k++) { uint32_t c = st[k]; st[k] = acc; acc += c; }
        I.pos[sn].resize(acc);
        fill.assign(st.begin(), st.end() - 1);
      }
      for (int cn = 0; cn < G.num_contigs(); cn++) {
        const uint32_t* g = G.index_seq(cn);
        int load = 0;  // genome.c:1139-1154: N/X resets the run; k-mers never span contigs
        uint64_t w = 0;
        for (uint32_t p = 0; p < G.len[cn]; p++) {
          int base = EXTRACT(g, p);
          w = (w >> 2) | ((uint64_t)(base & 3) << 62);
          if (base == 15) load = 0; else if (load < P.max_seed_span) load++;
          if (load < sd.span) continue;
          uint32_t mi = P.Hflag ? kmer_to_mapidx_hash(sd, P.max_seed_span, [&](llint q) { return EXTRACT(g, q); }, (llint)p) : mapidx_from_window(w, sel, nsel);
          if (pass == 0) st[mi]++;
          else I.pos[sn][fill[mi]++] = G.offsets[cn] + p - sd.span + 1;
        }
      }
    }
  }
}

// Same index, built by T threads over consecutive genome pieces (test-infrastructure speed only: the
// 3 Gbp bench baseline needs it in tens of seconds).  Piece t counts into its own histogram, the
// histograms are turned into per-piece write cursors in genome order, so every list comes out
// ascending exactly as the sequential builder (and genome.c:1156-1163) produce it.
static inline void build_index(const Params& P, const Genome& G, Index& I, int nthreads = 0) {
  const int ns = (int)P.seeds.size();
  uint64_t total = 0; for (auto l : G.len) total += l;
  int T = nthreads > 0 ? nthreads : omp_get_max_threads();
  if (nthreads <= 0 && total < (1u << 22)) T = 1;
  struct Piece { int cn; uint32_t a, b; };
  std::vector<std::vector<Piece>> slots(T);
  {  // cut the genome into T runs of (nearly) equal length, in genome order
    const uint64_t per = (total + T - 1) / T; uint64_t acc = 0;
    for (int cn = 0; cn < G.num_contigs(); cn++) {
      uint32_t a = 0;
      while (a < G.len[cn]) {
        const int t = (int)std::min<uint64_t>(T - 1, acc / std::max<uint64_t>(1, per));
        const uint64_t room = (uint64_t)(t + 1) * per - acc;
        const uint32_t b = (uint32_t)std::min<uint64_t>(G.len[cn], (uint64_t)a + std::max<uint64_t>(1, room));
        slots[t].push_back({cn, a, b}); acc += b - a; a = b;
      }
    }
  }
  std::vector<std::vector<int>> sel(ns); std::vector<size_t> cap(ns);
  for (int sn = 0; sn < ns; sn++) {
    cap[sn] = (size_t)1 << index_key_bits(P, P.seeds[sn]);
    for (int t = 0; t < P.seeds[sn].span; t++) if ((P.seeds[sn].mask >> t) & 1) sel[sn].push_back(t);
  }
  I.start.assign(ns, {}); I.pos.assign(ns, {});
  // k-mers of seed sn that end inside slot t's pieces, in genome order
  auto scan = [&](int t, int sn, auto&& emit) {
    const Seed& sd = P.seeds[sn];
    for (const Piece& pc : slots[t]) {
      const uint32_t* g = G.index_seq(pc.cn);
      int load = 0; uint64_t w = 0;
      const uint32_t p0 = pc.a >= (uint32_t)(P.max_seed_span - 1) ? pc.a - (uint32_t)(P.max_seed_span - 1) : 0;   // warm-up for the k-mers ending in [a, b)
      for (uint32_t p = p0; p < pc.b; p++) {
        const int base = EXTRACT(g, p);
        w = (w >> 2) | ((uint64_t)(base & 3) << 62);
        if (base == 15) load = 0; else if (load < P.max_seed_span) load++;      // genome.c:1139-1154
        if (p < pc.a || load < sd.span) continue;
        emit(P.Hflag ? kmer_to_mapidx_hash(sd, P.max_seed_span, [&](llint q) { return EXTRACT(g, q); }, (llint)p) : mapidx_from_window(w, sel[sn].data(), (int)sel[sn].size()),
             G.offsets[pc.cn] + p - sd.span + 1);
      }
    }
  };
  // Two-level counting sort so that every scatter stays cache-resident: first by the high key bits
  // into <= 4096 buckets (each slot writes its own sub-range of a bucket, slots in genome order),
  // then inside each bucket by the low key bits.  Both levels are stable, so lists stay ascending.
  std::unique_ptr<uint64_t[]> tmp; uint64_t tmp_cap = 0;
  for (int sn = 0; sn < ns; sn++) {
    const bool vb = getenv("GMO_VERBOSE") != nullptr; double tv = omp_get_wtime();
    auto lap = [&](const char* what) { if (vb) { const double n = omp_get_wtime(); fprintf(stderr, "  seed %d %s %.2f s\n", sn, what, n - tv); tv = n; } };
    const int kbits = index_key_bits(P, P.seeds[sn]), lo_bits = std::min(kbits, 12), nb = 1 << (kbits - lo_bits);
    std::vector<std::vector<uint64_t>> cur(T, std::vector<uint64_t>(nb, 0));
#pragma omp parallel for schedule(static, 1) num_threads(T)
    for (int t = 0; t < T; t++) { uint64_t* c = cur[t].data(); scan(t, sn, [&](uint32_t mi, uint32_t) { c[mi >> lo_bits]++; }); }
    lap("count");
    std::vector<uint64_t> bbase(nb + 1, 0);
    { uint64_t acc = 0;
      for (int b = 0; b < nb; b++) { bbase[b] = acc; for (int t = 0; t < T; t++) { const uint64_t c = cur[t][b]; cur[t][b] = acc; acc += c; } }
      bbase[nb] = acc; }
    const uint64_t n_ent = bbase[nb];
    if (n_ent > tmp_cap) { tmp.reset(new uint64_t[n_ent]); tmp_cap = n_ent; }   // uninitialised on purpose: every slot is written by the scatter
    lap("alloc");
#pragma omp parallel for schedule(static, 1) num_threads(T)
    for (int t = 0; t < T; t++) {
      uint64_t* c = cur[t].data(); uint64_t* out = tmp.get();
      scan(t, sn, [&](uint32_t mi, uint32_t pos) { out[c[mi >> lo_bits]++] = ((uint64_t)(mi & ((1u << lo_bits) - 1)) << 32) | pos; });
    }
    lap("scatter");
    I.start[sn].assign(cap[sn] + 1, 0); I.pos[sn].resize(n_ent);
    lap("alloc2");
    uint32_t* st = I.start[sn].data(); uint32_t* ps = I.pos[sn].data();
#pragma omp parallel for schedule(dynamic, 16) num_threads(T)
    for (int b = 0; b < nb; b++) {
      std::vector<uint32_t> c((size_t)1 << lo_bits, 0);
      for (uint64_t i = bbase[b]; i < bbase[b + 1]; i++) c[tmp[i] >> 32]++;
      uint64_t acc = bbase[b];
      for (uint32_t k = 0; k < (1u << lo_bits); k++) { st[((size_t)b << lo_bits) + k] = (uint32_t)acc; const uint32_t n = c[k]; c[k] = (uint32_t)(acc - bbase[b]); acc += n; }
      for (uint64_t i = bbase[b]; i < bbase[b + 1]; i++) ps[bbase[b] + c[tmp[i] >> 32]++] = (uint32_t)tmp[i];
    }
    st[cap[sn]] = (uint32_t)n_ent;
    lap("buckets");
  }
}

// automatic list cutoff (gmapper/gmapper.c:2811-2837)
static inline uint32_t auto_list_cutoff(const Params& P, const Genome& G) {
  unsigned long long tot = 0; for (auto l : G.len) tot += l;
  int maxw = 0; for (auto& s : P.seeds) maxw = std::max(maxw, s.weight);
  if (P.Hflag) maxw = HASH_TABLE_POWER;                      // gmapper.c:2820-2822
  uint32_t cutoff = 1000;
  unsigned long long p4 = 1ull << (2 * maxw);
  if ((uint32_t)((100ull * tot) / p4) > cutoff) cutoff = (uint32_t)((100ull * tot) / p4);
  return cutoff;
}

// get_contig_num (gmapper/gmapper.h:373-405) == upper_bound(offsets, idx) - 1
static inline int get_contig_num(const Genome& G, uint32_t idx) {
  int cn = 0;
  while (cn < G.num_contigs() - 1 && idx >= G.offsets[cn + 1]) cn++;
  return cn;
}

// ---------------------------------------------------------------------------------------------
// Anchors (common/anchors.c, common/anchors.h; struct anchor gmapper-definitions.h:66-74)
// ---------------------------------------------------------------------------------------------
struct Anchor { llint x = 0, y = 0; int length = 0, width = 0, weight = 0, cn = 0; };

static inline void anchor_join(const Anchor* a, int n, Anchor* dest) {  // anchors.c:9-52
  llint nw_min = INT_MAX, sw_min = INT_MAX, ne_max = INT_MIN, se_max = INT_MIN;
  dest->weight = 0; dest->cn = a[0].cn;
  for (int i = 0; i < n; i++) {
    llint nw = a[i].x + a[i].y, sw = a[i].x - a[i].y;
    llint ne = sw + 2 * (a[i].width - 1), se = nw + 2 * (a[i].length - 1);
    nw_min = std::min(nw_min, nw); sw_min = std::min(sw_min, sw);
    ne_max = std::max(ne_max, ne); se_max = std::max(se_max, se);
    dest->weight += a[i].weight;
  }
  if ((nw_min + sw_min) % 2 != 0) nw_min--;
  dest->x = (nw_min + sw_min) / 2;
  dest->y = nw_min - dest->x;
  if ((ne_max - sw_min) % 2 != 0) ne_max++;
  dest->width = (int)((ne_max - sw_min) / 2 + 1);
  if ((se_max - nw_min) % 2 != 0) se_max++;
  dest->length = (int)((se_max - nw_min) / 2 + 1);
}
static inline void anchor_widen(Anchor* a, int w) { a->x -= w / 2; a->y += w / 2; a->width += w; }  // anchors.c:55-61
static inline void anchor_get_x_range(const Anchor* a, int x_len, int y_len, int y, int* x_min, int* x_max) {  // anchors.c:64-95
  (void)y_len;
  if (y < a->y) *x_min = 0;
  else if (y <= a->y + (a->length - 1)) *x_min = (int)(a->x + (y - a->y));
  else *x_min = (int)(a->x + a->length);
  if (*x_min < 0) *x_min = 0;
  if (*x_min >= x_len) *x_min = x_len - 1;
  if (y < a->y - (a->width - 1)) *x_max = (int)(a->x + (a->width - 1) - 1);
  else if (y <= a->y - (a->width - 1) + (a->length - 1)) *x_max = (int)(a->x + (a->width - 1) + (y - (a->y - (a->width - 1))));
  else *x_max = x_len - 1;
  if (*x_max < 0) *x_max = 0;
  if (*x_max >= x_len) *x_max = x_len - 1;
}
static inline void anchor_uw_join(Anchor* dest, const Anchor* src) {  // anchors.c:98-119
  if (src->x < dest->x) {
    llint tmp = dest->x;
    dest->x = src->x; dest->y = src->y;
    if (src->x + src->length > tmp + dest->length) dest->length = src->length;
    else dest->length += (int)(tmp - dest->x);
  } else {
    if (src->x + src->length > dest->x + dest->length) dest->length = (int)(src->x - dest->x + src->length);
  }
  dest->weight += src->weight;
}
static inline bool anchor_uw_colinear(const Anchor* a, const Anchor* b) { return a->x - a->y == b->x - b->y; }  // anchors.h:17-20
static inline void anchor_reverse(Anchor* a, int x_len, int y_len) {  // anchors.h:30-34
  a->x = -a->x + (x_len - 1) - (a->length - 1) - (a->width - 1);
  a->y = -a->y + (y_len - 1) - (a->length - 1) + (a->width - 1);
}

// ---------------------------------------------------------------------------------------------
// Vector SW filter: score-only affine local SW (common/sw-vector.c:228-377,453-515).
// The SSE2 anti-diagonal sweep with -1/-2 sentinels computes exactly this recurrence over
// the glen x rlen matrix (see DESIGN.md "sw_vector semantics"); int16 range is guaranteed by
// sw_vector_setup's match*qrlen < 32768 check (sw-vector.c:393-398).
//   H(i,j) = max(0, H(i-1,j-1)+s, A(i,j), B(i,j))
//   A(i,j) = max(A(i,j-1) - a_ext, H(i,j-1) - a_open - a_ext)       gap along the genome
//   B(i,j) = max(B(i-1,j) - b_ext, H(i-1,j) - b_open - b_ext)       gap along the read
// Equality is on the raw 4-bit code (N matches N).
// ---------------------------------------------------------------------------------------------
static inline int sw_vector(const Params& P, const uint32_t* genome, llint goff, int glen,
                            const uint32_t* read, int rlen) {
  const int a_go = -P.a_gap_open_score, a_ge = -P.a_gap_extend_score;
  const int b_go = -P.b_gap_open_score, b_ge = -P.b_gap_extend_score;
  std::vector<int> H(glen + 1, 0), B(glen + 1, -b_go);   // nogap[] = 0, b_gap[] = -b_gap_open (sw-vector.c:261-264)
  std::vector<int8_t> db(glen);
  for (int j = 0; j < glen; j++) db[j] = (int8_t)EXTRACT(genome, goff + j);
  int score = 0;
  for (int i = 0; i < rlen; i++) {
    int q = EXTRACT(read, i);
    int hdiag = 0;          // H(i-1, -1)
    int hleft = 0;          // H(i, -1)
    int a = -a_go;          // v_a_gap initial = a_gap_ext - a_gap_open_ext (sw-vector.c:312-313)
    for (int j = 0; j < glen; j++) {
      a = std::max(a - a_ge, hleft - a_go - a_ge);
      int b = std::max(B[j + 1] - b_ge, H[j + 1] - b_go - b_ge);
      int h = hdiag + (db[j] == q ? P.match_score : P.mismatch_score);
      h = std::max(h, 0); h = std::max(h, a); h = std::max(h, b);
      hdiag = H[j + 1];
      H[j + 1] = h; B[j + 1] = b; hleft = h;
      score = std::max(score, h);
    }
  }
  return score;
}

// sw_gapless (common/sw-gapless.c:57-117): best ungapped segment on the diagonal through (g_idx, r_idx) of the contig.  Colour space (genome_ls != null:
// genome holds colours): a diagonal that starts at the read's first colour compares it with lstocs(letter, primer) first (:84-94); `mismatch` is what
// sw_gapless_setup got (f1_setup hands over match + crossover in colour space, gmapper.c:2935, f1-wrapper.h:66-68).
static inline int sw_gapless(const Params& P, const uint32_t* genome, int glen, const uint32_t* read, int rlen, int g_idx, int r_idx,
                             const uint32_t* genome_ls = nullptr, int init_bp = -1, bool is_rna = false) {
  const int mismatch = genome_ls ? P.match_score + P.crossover_score : P.mismatch_score;
  int g_left, r_left;
  if (g_idx < r_idx) { g_left = 0; r_left = r_idx - g_idx; } else { g_left = g_idx - r_idx; r_left = 0; }
  int g_right = g_left, r_right = r_left, score = 0, max_score = 0;
  if (genome_ls != nullptr && r_left == 0) {           // forcefully match the first colour of the read
    const int real_colour = lstocs((int)EXTRACT(genome_ls, g_right), init_bp, is_rna);
    if (real_colour == (int)EXTRACT(read, 0)) score = P.match_score; else { r_left++; g_left++; }
    r_right++; g_right++;
    max_score = score;
  }
  while (g_right < glen && r_right < rlen) {
    score += (EXTRACT(genome, g_right) == EXTRACT(read, r_right)) ? P.match_score : mismatch;
    if (score > max_score) max_score = score;
    g_right++; r_right++;
    if (score < 0) score = 0;
  }
  return max_score;
}

// Colour-space vector SW (common/sw-vector.c:112-146 first row, then the same recurrence on colours; sw_vector :453-515).
// genome_cs / read hold colours, genome_ls the letters of the same window; the first read colour is compared with the
// colour between the read's initial base and the genome letter.  `mismatch` is what sw_vector_setup got: match + crossover
// in colour space (gmapper.c:2935).
static inline int sw_vector_cs(const Params& P, int mismatch, const uint32_t* genome_cs, llint goff, int glen,
                               const uint32_t* read, int rlen, const uint32_t* genome_ls, int initbp, bool is_rna = false) {
  const int a_go = -P.a_gap_open_score, a_ge = -P.a_gap_extend_score;
  const int b_go = -P.b_gap_open_score, b_ge = -P.b_gap_extend_score;
  std::vector<int> H(glen + 1, 0), B(glen + 1, -b_go);
  int score = 0;
  for (int i = 0; i < rlen; i++) {
    const int q = EXTRACT(read, i);
    int hdiag = 0, hleft = 0, a = -a_go;
    for (int j = 0; j < glen; j++) {
      const int gcode = (i == 0) ? lstocs((int)EXTRACT(genome_ls, goff + j), initbp, is_rna) : (int)EXTRACT(genome_cs, goff + j);
      a = std::max(a - a_ge, hleft - a_go - a_ge);
      const int b = std::max(B[j + 1] - b_ge, H[j + 1] - b_go - b_ge);
      int h = hdiag + (gcode == q ? P.match_score : mismatch);
      h = std::max(h, 0); h = std::max(h, a); h = std::max(h, b);
      hdiag = H[j + 1];
      H[j + 1] = h; B[j + 1] = b; hleft = h;
      score = std::max(score, h);
    }
  }
  return score;
}

// hash_genome_window (common/util.h:224-245) with common/hash.h:70-95
static inline uint32_t hash_genome_window(const uint32_t* genome, uint32_t goff, uint32_t glen) {
  uint32_t key = 0;
  for (uint32_t i = 0; i < (glen + 15) / 16; i++) {
    uint32_t buffer = 0;
    for (uint32_t j = 0; j < 16 && i * 16 + j < glen; j++) {
      buffer <<= 2; buffer |= (uint32_t)(EXTRACT(genome, (llint)goff + i * 16 + j) & 0x3);
    }
    key += (buffer >> 16);
    uint32_t tmp = ((buffer & 0xFFFF) << 11) ^ key;
    key = (key << 16) ^ tmp;
    key += key >> 11;
  }
  key ^= key << 3; key += key >> 5; key ^= key << 4; key += key >> 17; key ^= key << 25; key += key >> 6;
  return key;
}

// ---------------------------------------------------------------------------------------------
// Full SW, letter space (common/sw-full-ls.c:154-516,637-683; struct common/sw-full-common.h:13-48)
// ---------------------------------------------------------------------------------------------
struct SwFullResults {
  int read_start = 0, rmapped = 0, genome_start = 0, gmapped = 0;
  int matches = 0, mismatches = 0, insertions = 0, deletions = 0, score = 0;
  int posterior_score = 0, pct_posterior_score = 0;
  std::string dbalign, qralign;
  double posterior = 0;
  int mqv = 255; double z0 = 0, z1 = 0, z2 = 0, z3 = 0, pr_top_random_at_location = 0, pr_missed_mp = 0, insert_size_denom = 0;
  int crossovers = 0; std::string qual;   // colour space: sw_full_cs / post_sw (sw-full-common.h:29-31)
  std::string ops;   // backtrace ops in alignment order: 'M' match/mismatch, 'I' BACK_INSERTION (gap in read), 'D' BACK_DELETION (gap in genome)
};

enum { FROM_NORTH_NORTH = 1, FROM_NORTH_NORTHWEST = 2, FROM_WEST_NORTHWEST = 3, FROM_WEST_WEST = 4,
       FROM_NORTHWEST_NORTH = 5, FROM_NORTHWEST_NORTHWEST = 6, FROM_NORTHWEST_WEST = 7 };

struct SwFullWorkspace {
  uint64_t local_retries = 0;     // local mode: alignments that took the second, threshold-band run
  struct Cell { int n, w, nw; int8_t bn, bw, bnw; };
  std::vector<Cell> m;
  std::vector<int8_t> db, qr;
  uint64_t cells = 0;
};

static inline void sw_full_ls(const Params& P, SwFullWorkspace& W, const uint32_t* genome, llint goff, int glen,
                              const uint32_t* read, int rlen, int threshscore, int maxscore, SwFullResults* sfr,
                              bool revcmpl, const Anchor* anchors, int anchors_cnt, int local_alignment) {
  const int lena = glen, lenb = rlen;
  const int a_go = -P.a_gap_open_score, a_ge = -P.a_gap_extend_score;
  const int b_go = -P.b_gap_open_score, b_ge = -P.b_gap_extend_score;
  const int match = P.match_score, mismatch = P.mismatch_score;
  W.db.resize(lena); W.qr.resize(lenb);
  for (int i = 0; i < lena; i++) W.db[i] = (int8_t)EXTRACT(genome, goff + i);
  for (int i = 0; i < lenb; i++) W.qr[i] = (int8_t)EXTRACT(read, i);
  auto init_cell = [&](size_t idx, int local) {  // sw-full-ls.c:66-80
    auto& c = W.m[idx];
    if (local) { c.nw = 0; c.n = -b_go; c.w = -a_go; }
    else { c.nw = -INT_MAX / 2; c.n = -INT_MAX / 2; c.w = -INT_MAX / 2; }
    c.bnw = c.bn = c.bw = 0;
  };
  int score = 0, max_i = 0, max_j = 0;
  // full_sw (sw-full-ls.c:154-403) over one band; local mode (Gflag off): states floor at 0 with a null back pointer, the best
  // cell is the first in row-major order to reach the filter's score, where the scan stops (:293-374)
  auto full_sw = [&](const Anchor& rectangle) {
    // The reference never clears swmatrix between calls; every cell it reads is written first
    // (band geometry is monotone), so a fresh poison fill is equivalent and catches mistakes.
    W.m.assign((size_t)(lena + 1) * (lenb + 1), SwFullWorkspace::Cell{INT_MIN / 4, INT_MIN / 4, INT_MIN / 4, 0, 0, 0});
    for (int j = 0; j < lena + 1; j++) init_cell(j, 1);  // sw-full-ls.c:194-196
    score = 0; max_i = 0; max_j = 0;
    for (int i = 0; i < lenb; i++) {
      int x_min, x_max;
      anchor_get_x_range(&rectangle, lena, lenb, i, &x_min, &x_max);
      init_cell((size_t)(i + 1) * (lena + 1) + (x_min - 1) + 1, local_alignment ? 1 : 0);   // :228-233
      W.cells += x_max - x_min + 1;
      int j;
      for (j = x_min; j <= x_max; j++) {
        auto* cnw = &W.m[(size_t)i * (lena + 1) + j];
        auto* cn = cnw + 1; auto* cw = cnw + (lena + 1); auto* cur = cw + 1;
        int ms = (W.db[j] == W.qr[i]) ? match : mismatch;
        int tmp; int8_t tmp2;
        if (!revcmpl) {                                        // :264-278
          tmp = cnw->nw + ms; tmp2 = FROM_NORTHWEST_NORTHWEST;
          if (cnw->n + ms > tmp) { tmp = cnw->n + ms; tmp2 = FROM_NORTHWEST_NORTH; }
          if (cnw->w + ms > tmp) { tmp = cnw->w + ms; tmp2 = FROM_NORTHWEST_WEST; }
        } else {                                               // :279-292
          tmp = cnw->w + ms; tmp2 = FROM_NORTHWEST_WEST;
          if (cnw->n + ms > tmp) { tmp = cnw->n + ms; tmp2 = FROM_NORTHWEST_NORTH; }
          if (cnw->nw + ms > tmp) { tmp = cnw->nw + ms; tmp2 = FROM_NORTHWEST_NORTHWEST; }
        }
        if (tmp <= 0 && local_alignment) { tmp = 0; tmp2 = 0; }
        cur->nw = tmp; cur->bnw = tmp2;
        if (!revcmpl) {                                        // north :303-320
          tmp = cn->nw - b_go - b_ge; tmp2 = FROM_NORTH_NORTHWEST;
          if (cn->n - b_ge > tmp) { tmp = cn->n - b_ge; tmp2 = FROM_NORTH_NORTH; }
        } else {
          tmp = cn->n - b_ge; tmp2 = FROM_NORTH_NORTH;
          if (cn->nw - b_go - b_ge > tmp) { tmp = cn->nw - b_go - b_ge; tmp2 = FROM_NORTH_NORTHWEST; }
        }
        if (tmp <= 0 && local_alignment) { tmp = 0; tmp2 = 0; }
        cur->n = tmp; cur->bn = tmp2;
        if (!revcmpl) {                                        // west :330-347
          tmp = cw->nw - a_go - a_ge; tmp2 = FROM_WEST_NORTHWEST;
          if (cw->w - a_ge > tmp) { tmp = cw->w - a_ge; tmp2 = FROM_WEST_WEST; }
        } else {
          tmp = cw->w - a_ge; tmp2 = FROM_WEST_WEST;
          if (cw->nw - a_go - a_ge > tmp) { tmp = cw->nw - a_go - a_ge; tmp2 = FROM_WEST_NORTHWEST; }
        }
        if (tmp <= 0 && local_alignment) { tmp = 0; tmp2 = 0; }
        cur->w = tmp; cur->bw = tmp2;
        if (local_alignment || i == lenb - 1) {                // :359-368 (global: last read row only)
          int t = std::max(cur->n, cur->nw); t = std::max(t, cur->w);
          if (t > score) { score = t; max_i = i; max_j = j; }
        }
        if (score == maxscore && local_alignment) break;      // :370-371
      }
      if (score == maxscore && local_alignment) break;        // :374-375
      if (i + 1 < lenb) {                                      // :378-385
        int nx_min, nx_max;
        anchor_get_x_range(&rectangle, lena, lenb, i + 1, &nx_min, &nx_max);
        for (int j2 = x_max + 1; j2 <= nx_max; j2++) init_cell((size_t)(i + 1) * (lena + 1) + (j2 + 1), local_alignment);
      }
    }
  };
  Anchor rectangle;
  auto threshold_band = [&]() {                       // sw-full-ls.c:179-192
    Anchor t[2];
    t[0].x = 0; t[0].y = (lenb * match - threshscore) / match; t[0].length = 1; t[0].width = 1;
    t[1].x = lena - 1; t[1].y = lenb - 1 - t[0].y; t[1].length = 1; t[1].width = 1;
    anchor_join(t, 2, &rectangle);
  };
  if (anchors != nullptr && P.anchor_width >= 0) {    // sw-full-ls.c:176-178
    anchor_join(anchors, anchors_cnt, &rectangle);
    anchor_widen(&rectangle, P.anchor_width);
  } else threshold_band();
  full_sw(rectangle);
  if (local_alignment && score != maxscore && anchors != nullptr) {         // :395-398: the filter's alignment left the band: once more over the band the threshold allows
    W.local_retries++;
    threshold_band();
    full_sw(rectangle);
    assert(score == maxscore);
  }
  sfr->score = score;
  sfr->ops.clear();
  if (score <= 0) {
    // The reference backtraces from cell (0,0) here, reading whatever a previous call left in its
    // scratch matrix; the result is discarded by every caller (score 0 fails the threshold).
    sfr->rmapped = 1; sfr->gmapped = 1; sfr->genome_start = (int)goff;
    return;
  }
  // do_backtrace (sw-full-ls.c:413-516)
  int i = max_i, j = max_j;
  auto* cell = &W.m[(size_t)(i + 1) * (lena + 1) + j + 1];
  int from = cell->bnw; int fromscore = cell->nw;
  if (cell->w > fromscore) { from = cell->bw; fromscore = cell->w; }
  if (cell->n > fromscore) from = cell->bn;
  std::string rev;
  while (i >= 0 && j >= 0) {
    switch (from) {
      case FROM_NORTH_NORTH: case FROM_NORTH_NORTHWEST:
        rev.push_back('D'); sfr->deletions++; sfr->read_start = i--; break;
      case FROM_WEST_WEST: case FROM_WEST_NORTHWEST:
        rev.push_back('I'); sfr->insertions++; sfr->genome_start = j--; break;
      case FROM_NORTHWEST_NORTH: case FROM_NORTHWEST_NORTHWEST: case FROM_NORTHWEST_WEST:
        rev.push_back('M');
        if (W.db[j] == W.qr[i]) sfr->matches++; else sfr->mismatches++;
        sfr->read_start = i--; sfr->genome_start = j--; break;
      default: assert(0);
    }
    cell = &W.m[(size_t)(i + 1) * (lena + 1) + j + 1];
    switch (from) {
      case FROM_NORTH_NORTH: from = cell->bn; break;
      case FROM_NORTH_NORTHWEST: from = cell->bnw; break;
      case FROM_WEST_WEST: from = cell->bw; break;
      case FROM_WEST_NORTHWEST: from = cell->bnw; break;
      case FROM_NORTHWEST_NORTH: from = cell->bn; break;
      case FROM_NORTHWEST_NORTHWEST: from = cell->bnw; break;
      case FROM_NORTHWEST_WEST: from = cell->bw; break;
    }
    if (from == 0) break;
  }
  sfr->ops.assign(rev.rbegin(), rev.rend());
  // pretty_print (sw-full-ls.c:524-560)
  {
    int pi = sfr->read_start, pj = sfr->genome_start;
    for (char op : sfr->ops) {
      if (op == 'D') { sfr->dbalign.push_back('-'); sfr->qralign.push_back(LSTRANS[W.qr[pi++]]); }
      else if (op == 'I') { sfr->dbalign.push_back(LSTRANS[W.db[pj++]]); sfr->qralign.push_back('-'); }
      else { sfr->dbalign.push_back(LSTRANS[W.db[pj++]]); sfr->qralign.push_back(LSTRANS[W.qr[pi++]]); }
    }
  }
  sfr->gmapped = max_j - sfr->genome_start + 1;     // sw-full-ls.c:672-675
  sfr->genome_start += (int)goff;
  sfr->rmapped = max_i - sfr->read_start + 1;
}

// ---------------------------------------------------------------------------------------------
// Full SW, colour space (common/sw-full-cs.c:249-623 full_sw, :633-937 do_backtrace, :945-1060 pretty_print,
// :1146-1236 sw_full_cs).  Four letter-space translations of the colour read (start letter (k + initbp) % 4), a
// 3-state affine DP in four layers; the NW and N transitions may come from another layer at +xover_penalty, the W
// transition may not; N-vs-anything scores 0.  Global mode and (round 3) local mode.
// ---------------------------------------------------------------------------------------------
struct CsParams { int match = 10, mismatch = -24, xover = -20, a_go = 33, a_ge = 7, b_go = 33, b_ge = 3, anchor_width = 8, indel_taboo_len = 0; };
typedef SwFullResults SwFullCsResults;

static inline void sw_full_cs(const CsParams& C, const uint32_t* genome_ls, llint goff, int glen, const uint32_t* read, int rlen, int initbp,
                              int threshscore, SwFullCsResults* sfr, bool revcmpl, const Anchor* anchors, int anchors_cnt,
                              const int* crossover_score = nullptr,     // per-position crossover scores from the read's QVs (gmapper.c:532-544), or null
                              int local_alignment = 0,                  // Gflag off (--local): states floored at 0 / the crossover score with a null back pointer, best cell of the whole band (:199-203,315,439-552)
                              bool is_rna = false) {                    // genome_is_rna: the four letter translations of the read hold U where DNA has T (:1191)
  const int lena = glen, lenb = rlen;
  struct Lay { int n, w, nw; int8_t bn, bw, bnw; };
  struct Cell { Lay from[4]; };
  std::vector<Cell> m((size_t)(lena + 1) * (lenb + 1));
  for (auto& c : m) for (int k = 0; k < 4; k++) c.from[k] = Lay{INT_MIN / 4, INT_MIN / 4, INT_MIN / 4, 0, 0, 0};   // poison: every cell read is written first
  std::vector<int8_t> db(lena); std::vector<int8_t> qr[4];
  for (int i = 0; i < lena; i++) db[i] = (int8_t)EXTRACT(genome_ls, goff + i);
  for (int k = 0; k < 4; k++) {                                          // :1182-1197
    qr[k].resize(lenb);
    int letter = (k + initbp) % 4;
    for (int j = 0; j < lenb; j++) {
      const int base = EXTRACT(read, j);
      if (base == 15) { qr[k][j] = 15; letter = (k + initbp) % 4; }
      else { qr[k][j] = (int8_t)cstols(letter, base, is_rna); letter = qr[k][j]; }
    }
  }
  int xo = C.xover;                                                      // global_xover_penalty; per row below (:312)
  auto init_cell = [&](size_t idx, int local) {                          // :201-247
    for (int k = 0; k < 4; k++) {
      Lay& l = m[idx].from[k];
      if (local) { const int x = (k == 0) ? 0 : xo; l.nw = x; l.n = -C.b_go + x; l.w = -C.a_go + x; }
      else { l.nw = l.n = l.w = -INT_MAX / 2; }
      l.bn = l.bw = l.bnw = 0;
    }
  };
  auto FROM_x = [](int mat, int dir) { return (int8_t)((dir << 2) | mat); };
  Anchor rectangle;
  anchor_join(anchors, anchors_cnt, &rectangle); anchor_widen(&rectangle, C.anchor_width);      // :284-287
  for (int j = 0; j < lena + 1; j++) init_cell(j, 1);                     // :266-268
  int score = 0, max_i = 0, max_j = 0, max_k = 0;
  for (int i = 0; i < lenb; i++) {
    int x_min, x_max;
    anchor_get_x_range(&rectangle, lena, lenb, i, &x_min, &x_max);
    xo = crossover_score ? crossover_score[i] : C.xover;                  // :312
    init_cell((size_t)(i + 1) * (lena + 1) + (x_min - 1) + 1, local_alignment ? 1 : 0);   // :315-322
    const bool notaboo = i < lenb - C.indel_taboo_len;
    for (int j = x_min; j <= x_max; j++) {
      Cell* cnw = &m[(size_t)i * (lena + 1) + j]; Cell* cn = cnw + 1; Cell* cw = cnw + (lena + 1); Cell* cur = cw + 1;
      for (int k = 0; k < 4; k++) {
        int ms;
        if (db[j] == 15 || qr[k][i] == 15) ms = 0; else ms = (db[j] == qr[k][i]) ? C.match : C.mismatch;
        int tmp; int8_t tmp2;
        // ---- northwest :356-438
        if (!revcmpl) {
          tmp = cnw->from[k].nw + ms; tmp2 = FROM_x(k, FROM_NORTHWEST_NORTHWEST);
          if (notaboo && cnw->from[k].n + ms > tmp) { tmp = cnw->from[k].n + ms; tmp2 = FROM_x(k, FROM_NORTHWEST_NORTH); }
          if (cnw->from[k].w + ms > tmp) { tmp = cnw->from[k].w + ms; tmp2 = FROM_x(k, FROM_NORTHWEST_WEST); }
        } else {
          tmp = cnw->from[k].w + ms; tmp2 = FROM_x(k, FROM_NORTHWEST_WEST);
          if (notaboo && cnw->from[k].n + ms > tmp) { tmp = cnw->from[k].n + ms; tmp2 = FROM_x(k, FROM_NORTHWEST_NORTH); }
          if (cnw->from[k].nw + ms > tmp) { tmp = cnw->from[k].nw + ms; tmp2 = FROM_x(k, FROM_NORTHWEST_NORTHWEST); }
        }
        for (int l = 0; l < 4; l++) {
          if (l == k) continue;
          if (!revcmpl) {
            if (cnw->from[l].nw + ms + xo > tmp) { tmp = cnw->from[l].nw + ms + xo; tmp2 = FROM_x(l, FROM_NORTHWEST_NORTHWEST); }
            if (notaboo && cnw->from[l].n + ms + xo > tmp) { tmp = cnw->from[l].n + ms + xo; tmp2 = FROM_x(l, FROM_NORTHWEST_NORTH); }
            if (cnw->from[l].w + ms + xo > tmp) { tmp = cnw->from[l].w + ms + xo; tmp2 = FROM_x(l, FROM_NORTHWEST_WEST); }
          } else {
            if (cnw->from[l].w + ms + xo > tmp) { tmp = cnw->from[l].w + ms + xo; tmp2 = FROM_x(l, FROM_NORTHWEST_WEST); }
            if (notaboo && cnw->from[l].n + ms + xo > tmp) { tmp = cnw->from[l].n + ms + xo; tmp2 = FROM_x(l, FROM_NORTHWEST_NORTH); }
            if (cnw->from[l].nw + ms + xo > tmp) { tmp = cnw->from[l].nw + ms + xo; tmp2 = FROM_x(l, FROM_NORTHWEST_NORTHWEST); }
          }
        }
        const int resetval = k ? xo : 0;                                    // :350-353
        if (tmp <= resetval && local_alignment) { tmp = resetval; tmp2 = 0; }   // :439-442
        cur->from[k].nw = tmp; cur->from[k].bnw = tmp2;
        // ---- north :447-503
        if (!revcmpl) {
          tmp = cn->from[k].nw - C.b_go - C.b_ge; tmp2 = FROM_x(k, FROM_NORTH_NORTHWEST);
          if (!notaboo || cn->from[k].n - C.b_ge > tmp) { tmp = cn->from[k].n - C.b_ge; tmp2 = FROM_x(k, FROM_NORTH_NORTH); }
        } else {
          tmp = cn->from[k].n - C.b_ge; tmp2 = FROM_x(k, FROM_NORTH_NORTH);
          if (notaboo && cn->from[k].nw - C.b_go - C.b_ge > tmp) { tmp = cn->from[k].nw - C.b_go - C.b_ge; tmp2 = FROM_x(k, FROM_NORTH_NORTHWEST); }
        }
        for (int l = 0; l < 4; l++) {
          if (l == k) continue;
          if (!revcmpl) {
            if (notaboo && cn->from[l].nw - C.b_go - C.b_ge + xo > tmp) { tmp = cn->from[l].nw - C.b_go - C.b_ge + xo; tmp2 = FROM_x(l, FROM_NORTH_NORTHWEST); }
            if (cn->from[l].n - C.b_ge + xo > tmp) { tmp = cn->from[l].n - C.b_ge + xo; tmp2 = FROM_x(l, FROM_NORTH_NORTH); }
          } else {
            if (cn->from[l].n - C.b_ge + xo > tmp) { tmp = cn->from[l].n - C.b_ge + xo; tmp2 = FROM_x(l, FROM_NORTH_NORTH); }
            if (notaboo && cn->from[l].nw - C.b_go - C.b_ge + xo > tmp) { tmp = cn->from[l].nw - C.b_go - C.b_ge + xo; tmp2 = FROM_x(l, FROM_NORTH_NORTHWEST); }
          }
        }
        if (tmp <= resetval && local_alignment) { tmp = resetval; tmp2 = 0; }   // :503-506
        cur->from[k].n = tmp; cur->from[k].bn = tmp2;
        // ---- west :512-541 (no crossover on a genomic gap)
        if (!revcmpl) {
          tmp = cw->from[k].nw - C.a_go - C.a_ge; tmp2 = FROM_x(k, FROM_WEST_NORTHWEST);
          if (!notaboo || cw->from[k].w - C.a_ge > tmp) { tmp = cw->from[k].w - C.a_ge; tmp2 = FROM_x(k, FROM_WEST_WEST); }
        } else {
          tmp = cw->from[k].w - C.a_ge; tmp2 = FROM_x(k, FROM_WEST_WEST);
          if (notaboo && cw->from[k].nw - C.a_go - C.a_ge > tmp) { tmp = cw->from[k].nw - C.a_go - C.a_ge; tmp2 = FROM_x(k, FROM_WEST_NORTHWEST); }
        }
        if (tmp <= resetval && local_alignment) { tmp = resetval; tmp2 = 0; }   // :540-543
        cur->from[k].w = tmp; cur->from[k].bw = tmp2;
        // ---- max on the last read row (local: of every row) :547-575
        if (local_alignment || i == lenb - 1) {
          const Lay& c = cur->from[k];
          if (!revcmpl) {
            if (c.nw > score) { score = c.nw; max_i = i; max_j = j; max_k = k; }
            if (c.n > score) { score = c.n; max_i = i; max_j = j; max_k = k; }
            if (c.w > score) { score = c.w; max_i = i; max_j = j; max_k = k; }
          } else {
            if (c.w > score) { score = c.w; max_i = i; max_j = j; max_k = k; }
            if (c.n > score) { score = c.n; max_i = i; max_j = j; max_k = k; }
            if (c.nw > score) { score = c.nw; max_i = i; max_j = j; max_k = k; }
          }
        }
      }
    }
    if (i + 1 < lenb) {                                                   // :598-606
      int nx_min, nx_max;
      anchor_get_x_range(&rectangle, lena, lenb, i + 1, &nx_min, &nx_max);
      for (int j = x_max + 1; j <= nx_max; j++) init_cell((size_t)(i + 1) * (lena + 1) + (j + 1), local_alignment);
    }
  }
  *sfr = SwFullCsResults();
  sfr->score = score;
  if (!(score >= 0 && score >= threshscore)) { sfr->score = 0; return; }  // :1216-1226
  // do_backtrace :633-937: bt entries = type | 0x80 when the step crosses over
  enum { BACK_INSERTION = 1, BACK_A_DELETION, BACK_B_DELETION, BACK_C_DELETION, BACK_D_DELETION, BACK_A_MM, BACK_B_MM, BACK_C_MM, BACK_D_MM };
  int i = max_i, j = max_j, k = max_k;
  const Cell* cell = &m[(size_t)(i + 1) * (lena + 1) + j + 1];
  int from = cell->from[k].bnw, fromscore = cell->from[k].nw;
  if (cell->from[k].w > fromscore) { from = cell->from[k].bw; fromscore = cell->from[k].w; }
  if (cell->from[k].n > fromscore) from = cell->from[k].bn;
  assert(from != 0);
  std::vector<uint8_t> rev;
  while (i >= 0 && j >= 0) {
    const int dir = from >> 2, lay = from & 3;
    uint8_t bt;
    if (dir == FROM_NORTH_NORTH || dir == FROM_NORTH_NORTHWEST) { sfr->deletions++; sfr->read_start = i--; bt = (uint8_t)(BACK_A_DELETION + k); }
    else if (dir == FROM_WEST_WEST || dir == FROM_WEST_NORTHWEST) { sfr->insertions++; sfr->genome_start = j--; bt = BACK_INSERTION; }
    else {
      if (db[j] == qr[k][i] || db[j] == 15 || qr[k][i] == 15) sfr->matches++; else sfr->mismatches++;
      sfr->read_start = i--; sfr->genome_start = j--; bt = (uint8_t)(BACK_A_MM + k);
    }
    if (k != lay) { bt |= 0x80; sfr->crossovers++; k = lay; }
    rev.push_back(bt);
    cell = &m[(size_t)(i + 1) * (lena + 1) + j + 1];
    switch (dir) {
      case FROM_NORTH_NORTH: from = cell->from[k].bn; break;
      case FROM_NORTH_NORTHWEST: from = cell->from[k].bnw; break;
      case FROM_WEST_WEST: from = cell->from[k].bw; break;
      case FROM_WEST_NORTHWEST: from = cell->from[k].bnw; break;
      case FROM_NORTHWEST_NORTH: from = cell->from[k].bn; break;
      case FROM_NORTHWEST_NORTHWEST: from = cell->from[k].bnw; break;
      case FROM_NORTHWEST_WEST: from = cell->from[k].bw; break;
      default: assert(0);
    }
    if (from == 0) break;
  }
  if (k != 0) { rev.back() |= 0x80; sfr->crossovers++; }                   // :929-932 (first step of the alignment)
  // pretty_print :945-1060
  {
    int pi = sfr->read_start, pj = sfr->genome_start;
    for (size_t t = rev.size(); t-- > 0;) {
      const uint8_t bt = rev[t]; const int type = bt & 0x0f; const bool xov = (bt & 0x80) != 0;
      if (type == BACK_INSERTION) { sfr->dbalign.push_back(LSTRANS[db[pj++]]); sfr->qralign.push_back('-'); sfr->ops.push_back('I'); continue; }
      const bool del = type >= BACK_A_DELETION && type <= BACK_D_DELETION;
      const int lay = del ? type - BACK_A_DELETION : type - BACK_A_MM;
      char q = LSTRANS[qr[lay][pi++]];
      if (xov) q = (char)tolower((int)q);
      if (del) { sfr->dbalign.push_back('-'); sfr->qralign.push_back(q); sfr->ops.push_back('D'); }
      else {
        const char d = LSTRANS[db[pj++]];
        if (q == 'n' || q == 'N') q = xov ? (char)tolower((int)d) : d;
        sfr->dbalign.push_back(d); sfr->qralign.push_back(q); sfr->ops.push_back('M');
      }
    }
  }
  sfr->gmapped = max_j - sfr->genome_start + 1;                          // :1219-1223
  sfr->genome_start += (int)goff;
  sfr->rmapped = max_i - sfr->read_start + 1;
}


// ---------------------------------------------------------------------------------------------
// post_sw (common/sw-post.c:639-758), reads without quality values: a 16-state forward-backward
// over the aligned columns of a colour-space alignment.  State j = (previous letter << 2 | letter);
// all sums run in the reference's order, in doubles, through the same libm calls, so that the
// truncations downstream (QV characters, Z0/Z1, MAPQ, posterior_score) see the same bits.
// ---------------------------------------------------------------------------------------------
struct PostSwColumn {              // struct column, sw-post.c:61-78 (one letter and one colour emission at most)
  double forwards[16], backwards[16], forwscale, backscale;
  int nlets, ncols, let, col; double leterr, colerr;
  double posterior[4]; int max_posterior, base_call;
};
static inline int qv_from_pr_err(double pr_err) {  // util.h:267-276
  if (pr_err > .99999999) return 0; else if (pr_err < 1E-25) return 250;
  return (int)(-10.0 * log(pr_err) / log(10.0));
}
static inline double post_node_prior(const PostSwColumn& c, int j) {   // nodePrior, sw-post.c:111-138
  double val = 0;
  const int l = (j >> 2) & 3, r = j & 3;
  if (c.nlets) { if (r == c.let) val = val - log(1 - c.leterr); else val = val - log(c.leterr / 3.0); }
  if (c.ncols) { if ((l ^ r) == c.col) val = val - log(1 - c.colerr); else val = val - log(c.colerr / 3.0); }
  return val;
}
static inline void post_sw(const Params& P, const uint32_t* read, int init_bp, SwFullResults* sfr, const char* qual = nullptr) {   // qual: the read's QV string, or null
  std::vector<PostSwColumn> cols;
  {  // load_local_vectors (sw-post.c:448-528)
    int start_run = 0, j, min_qv = 10000;
    for (j = 0; j < sfr->read_start; j++) {
      int col = EXTRACT(read, j);
      if (col == 15) { start_run = 15; min_qv = 0; j = sfr->read_start; break; }
      start_run ^= col;
      if (qual) min_qv = std::min(min_qv, (int)qual[P.qual_vector_offset + j]);
    }
    for (size_t i = 0; i < sfr->dbalign.size(); i++) {
      if (sfr->qralign[i] == '-') continue;       // deletion: nothing to emit
      PostSwColumn c; memset(&c, 0, sizeof c);
      if (sfr->dbalign[i] != '-') {
        c.nlets = 1; c.leterr = P.pr_mismatch;
        switch (sfr->dbalign[i]) {                // fasta_get_initial_base (fasta.c:556-577): A/C/G/T, anything else -1
          case 'A': case 'a': c.let = 0; break; case 'C': case 'c': c.let = 1; break;
          case 'G': case 'g': c.let = 2; break; case 'T': case 't': c.let = 3; break; default: c.let = -1;
        }
      }
      c.ncols = 1;
      const int col = EXTRACT(read, j); const bool first = cols.empty();
      if ((first && start_run == 15) || col == 15) { c.col = 0; c.colerr = .75; }
      else {
        c.col = col ^ (first ? start_run : 0);
        if (qual) {                                  // sw-post.c:486-491
          const int q = (int)qual[P.qual_vector_offset + j];
          const int qv = (first ? std::min(min_qv, q) : q) - P.qual_delta;
          c.colerr = (qv <= 0) ? .99999999 : (qv >= 250 ? 1E-25 : pow(10.0, -(double)qv / 10.0));
          if (!P.use_sanger_qvs) c.colerr /= (1 + c.colerr);
          if (c.colerr > .75) c.colerr = .75;
        } else c.colerr = P.pr_xover;
      }
      int bc = char_to_code_ls((unsigned char)sfr->qralign[i]);   // char_to_base (fasta.c:28-42): case-insensitive
      c.base_call = bc;
      cols.push_back(c); j++;
    }
  }
  const int len = (int)cols.size();
  if (len == 0) { sfr->posterior = 0; return; }
  PostSwColumn* a = cols.data();
  double total;
  {  // do_forwards (sw-post.c:317-360)
    a[0].forwscale = 999999999;
    for (int j = 0; j < 16; j++) {
      if (((j >> 2) & 3) == init_bp) { a[0].forwards[j] = post_node_prior(a[0], j); a[0].forwscale = std::min(a[0].forwscale, a[0].forwards[j]); }
      else a[0].forwards[j] = HUGE_VAL;
    }
    for (int j = 0; j < 16; j++) a[0].forwards[j] -= a[0].forwscale;
    for (int i = 1; i < len; i++) {
      a[i].forwscale = 999999999;
      for (int j = 0; j < 16; j++) a[i].forwards[j] = 0;
      for (int j = 0; j < 16; j++) {
        const double val = post_node_prior(a[i], j);
        for (int k = 0; k < 16; k++) if (((j >> 2) & 3) == (k & 3)) a[i].forwards[j] += exp(-1 * (a[i - 1].forwards[k]));
        a[i].forwards[j] = val - log(a[i].forwards[j]);
        a[i].forwscale = (a[i].forwscale < a[i].forwards[j]) ? a[i].forwscale : a[i].forwards[j];   // MIN2
      }
      for (int j = 0; j < 16; j++) a[i].forwards[j] -= a[i].forwscale;
      a[i].forwscale += a[i - 1].forwscale;
    }
    double val = 0;
    for (int j = 0; j < 16; j++) val += exp(-1 * (a[len - 1].forwards[j]));
    total = -log(val) + a[len - 1].forwscale;
  }
  {  // do_backwards (sw-post.c:269-315); its own total is only a sanity value in the reference
    int i = len - 1;
    a[i].backscale = 999999999;
    for (int j = 0; j < 16; j++) { a[i].backwards[j] = 0; a[i].backscale = (a[i].backscale < a[i].backwards[j]) ? a[i].backscale : a[i].backwards[j]; }
    for (int j = 0; j < 16; j++) a[i].backwards[j] -= a[i].backscale;
    for (i = len - 2; i >= 0; i--) {
      a[i].backscale = 999999999;
      for (int j = 0; j < 16; j++) a[i].backwards[j] = 0;
      for (int j = 0; j < 16; j++) {
        for (int k = 0; k < 16; k++)
          if ((j & 3) == ((k >> 2) & 3)) { const double val = post_node_prior(a[i + 1], k); a[i].backwards[j] += exp(-1 * (val + a[i + 1].backwards[k])); }
        a[i].backwards[j] = -log(a[i].backwards[j]);
        a[i].backscale = (a[i].backscale < a[i].backwards[j]) ? a[i].backscale : a[i].backwards[j];
      }
      for (int j = 0; j < 16; j++) a[i].backwards[j] -= a[i].backscale;
      a[i].backscale += a[i + 1].backscale;
    }
  }
  for (int i = 0; i < len; i++) {   // post_traceback (sw-post.c:183-212)
    for (int j = 0; j < 4; j++) a[i].posterior[j] = 0;
    for (int j = 0; j < 16; j++)
      a[i].posterior[j & 3] += exp(-1 * (a[i].forwards[j] + a[i].backwards[j] + a[i].forwscale + a[i].backscale - total));
    int mx = 0;
    for (int j = 1; j < 4; j++) if (a[i].posterior[j] > a[i].posterior[mx]) mx = j;
    a[i].max_posterior = mx;
  }
  {  // fix_base_calls (sw-post.c:531-565)
    int j = 0, prev_base = init_bp;
    sfr->matches = 0; sfr->mismatches = 0; sfr->crossovers = 0;
    for (size_t i = 0; i < sfr->qralign.size(); i++) {
      if (sfr->qralign[i] == '-') continue;
      const int crt = a[j].max_posterior;
      if ((prev_base ^ crt) == a[j].col) sfr->qralign[i] = "ACGT"[crt];
      else { sfr->qralign[i] = "acgt"[crt]; ++sfr->crossovers; }
      if (sfr->dbalign[i] != '-') { if (toupper(sfr->dbalign[i]) == toupper(sfr->qralign[i])) ++sfr->matches; else ++sfr->mismatches; }
      prev_base = crt; ++j;
    }
  }
  {  // get_base_qualities (sw-post.c:568-586): the posterior of the base sw_full_cs had called, capped at 40
    sfr->qual.assign(len, '!');
    for (int k = 0; k < len; k++) {
      // (the reference indexes posterior[] with any 4-bit code; only A/C/G/T and N are defined behaviour)
      int tmp = (a[k].base_call >= 0 && a[k].base_call <= 3) ? qv_from_pr_err(1 - a[k].posterior[a[k].base_call]) : 0;
      if (tmp > 40) tmp = 40;
      sfr->qual[k] = (char)(33 + tmp);
    }
  }
  {  // get_posterior (sw-post.c:589-612)
    double res = exp(-total);
    for (size_t i = 0; i < sfr->dbalign.size(); i++) {
      if (sfr->dbalign[i] == '-') { res *= P.pr_ins_extend; if (i == 0 || sfr->dbalign[i - 1] != '-') res *= P.pr_ins_open; }
      else if (sfr->qralign[i] == '-') { res *= P.pr_del_extend; if (i == 0 || sfr->qralign[i - 1] != '-') res *= P.pr_del_open; }
    }
    sfr->posterior = res;
  }
}


// ---------------------------------------------------------------------------------------------
// Per-read pipeline (gmapper/mapping.c)
// ---------------------------------------------------------------------------------------------
struct Hit {                       // struct read_hit, gmapper-definitions.h:131-160
  Anchor anchor;
  llint g_off = 0, g_off_pos_strand = 0;
  int score_window_gen = 0, score_vector = -1, pct_score_vector = 0, score_full = -1;
  double pct_score_full = 0;
  int pass1_key = 0, pass2_key = 0, score_max = 0, matches = 0, cn = 0, w_len = 0, st = 0, gen_st = 0;
  int saved = 0, sort_idx = 0;
  int pair_min = -1, pair_max = -1;
  std::vector<int> paired_hit_idx;
  bool has_sfr = false;
  SwFullResults sfr;
};

struct Read {
  std::string name, seq, qual;
  std::vector<uint32_t> bits[2];   // read[0] forward, read[1] reverse complement
  int read_len = 0, window_len = 0, max_n_kmers = 0, min_kmer_pos = 0, input_strand = 0;
  int initbp[2] = {0, 0};          // colour space: the primer letter (gmapper.c:481-482)
  std::vector<int> crossover_score; // colour space with QVs: per colour (gmapper.c:532-544)
  std::vector<uint32_t> mapidx[2];
  std::vector<Anchor> anchors[2];
  std::vector<Hit> hits[2];
  bool paired = false, first_in_pair = false; Read* mate_pair = nullptr;
  int delta_g_off_min[2] = {0, 0}, delta_g_off_max[2] = {0, 0};
  int delta_region_min[2] = {0, 0}, delta_region_max[2] = {0, 0};   // mate-pair region counts (mapping.c:2422-2430)
  std::vector<Hit> final_unpaired_hits; bool mapped = false;
};

struct Stats { uint64_t vec_calls = 0, vec_cells = 0, vec_bypassed = 0, full_calls = 0, full_cells = 0, reads_matched = 0, dup_pruned = 0;
               uint64_t pair_anchors = 0, pair_windows = 0; };   // paired mode: collapsed anchors / windows of both mates (stage check of the mate-pair region counts)

struct ThreadState {               // the reference's threadprivate state
  std::vector<uint16_t> region_map[2][2];       // region_map[number_in_pair][st]
  int region_map_id = 0;
  std::vector<uint32_t> f1_tag, f1_score;       // f1_window_cache (common/f1-wrapper.h:27-37)
  uint32_t f1_hash_tag = 0;
  SwFullWorkspace sww;
  Stats stats;
};

struct Mapper {
  Params P; const Genome* G = nullptr; const Index* I = nullptr;
  static const int region_map_id_bits = 13;
  static const int f1_window_cache_size = 1048576;

  void init_thread(ThreadState& T) const {
    int n_regions = 1 << (32 - P.region_bits);
    for (int nip = 0; nip < 2; nip++) for (int st = 0; st < 2; st++) T.region_map[nip][st].assign(n_regions, 0);
    T.region_map_id = 0;
    T.f1_tag.assign(f1_window_cache_size, 0); T.f1_score.assign(f1_window_cache_size, 0);
    T.f1_hash_tag = 0;
  }

  // launch_scan_threads body, LS unpaired part (gmapper/gmapper.c:436-531)
  void prepare_read(Read& re) const {
    if (P.colour) { prepare_read_cs(re); return; }
    re.read_len = (int)re.seq.size();
    re.max_n_kmers = re.read_len - P.min_seed_span + 1;
    std::vector<uint8_t> codes(re.read_len);
    for (int i = 0; i < re.read_len; i++) codes[i] = (uint8_t)char_to_code_ls((unsigned char)re.seq[i]);
    re.bits[0] = pack_codes(codes.data(), codes.size());
    re.bits[1] = revcomp_ls(re.bits[0].data(), re.read_len, codes_are_rna(codes.data(), codes.size()));   // re->is_rna (fasta.c:528-542), gmapper.c:487
    if (re.max_n_kmers < 0) re.max_n_kmers = 0;
    re.min_kmer_pos = 0; re.input_strand = 0;
    re.window_len = (uint16_t)GMO_ABS_OR_PCT(P.window_len, re.read_len);  // gmapper.c:530
  }
  // colour space: seq = primer letter + colours (gmapper.c:475-487; fasta.c:609-673, colour table :154-163)
  static int char_to_code_cs(unsigned char c) {
    switch (c) { case '0': return 0; case '1': return 1; case '2': return 2; case '3': return 3;
                 case '4': case 'N': case 'n': case '.': case 'X': case 'x': return 15; default: return -1; }
  }
  void prepare_read_cs(Read& re) const {
    re.read_len = (int)re.seq.size();
    re.max_n_kmers = re.read_len - P.min_seed_span + 1;
    std::vector<uint8_t> codes(re.read_len > 0 ? re.read_len - 1 : 0);
    for (int i = 1; i < re.read_len; i++) codes[i - 1] = (uint8_t)char_to_code_cs((unsigned char)re.seq[i]);
    re.bits[0] = pack_codes(codes.data(), codes.size());
    re.read_len--;
    re.max_n_kmers -= 2;                           // 1st colour never enters a k-mer
    re.min_kmer_pos = 1;
    re.initbp[0] = re.initbp[1] = char_to_code_ls((unsigned char)re.seq[0]);
    {  // reverse_complement_read_cs (util.c:600-617)
      const uint32_t* r = re.bits[0].data(); const int len = re.read_len;
      std::vector<uint8_t> rc(len, 0);
      int base = cstols(re.initbp[0], EXTRACT(r, 0));
      for (int i = 1; i < len; i++) { base = cstols(base, EXTRACT(r, i)); rc[len - i] = (uint8_t)EXTRACT(r, i); }
      rc[0] = (uint8_t)lstocs(base, complement_base(re.initbp[1]));
      re.bits[1] = pack_codes(rc.data(), rc.size());
    }
    if (re.max_n_kmers < 0) re.max_n_kmers = 0;
    re.input_strand = 0;
    re.window_len = (uint16_t)GMO_ABS_OR_PCT(P.window_len, re.read_len);
    re.crossover_score.clear();
    if (P.Qflag) {                                 // gmapper.c:532-544
      re.crossover_score.resize(re.read_len);
      for (int j = 0; j < re.read_len; j++) {
        int c = (int)(P.score_alpha * log(pr_err_from_qv((int)re.qual[j] - P.qual_delta) / 3.0) / log(2.0));
        if (c > -1) c = -1; else if (c < 2 * P.crossover_score) c = 2 * P.crossover_score;
        re.crossover_score[j] = c;
      }
    }
  }
  static double pr_err_from_qv(int qv) {           // util.h:285-293
    if (qv <= 0) return .99999999; else if (qv >= 250) return 1E-25;
    return pow(10.0, -(double)qv / 10.0);
  }

  // read_get_mapidxs_per_strand (mapping.c:37-70)
  void read_get_mapidxs(Read& re) const {
    int ns = (int)P.seeds.size();
    for (int st = 0; st < 2; st++) {
      re.mapidx[st].assign((size_t)ns * re.max_n_kmers, 0);
      const uint32_t* r = re.bits[st].data();
      for (int i = 0; i < re.read_len; i++)
        for (int sn = 0; sn < ns; sn++) {
          if (i < re.min_kmer_pos + P.seeds[sn].span - 1) continue;
          int r_idx = i - P.seeds[sn].span + 1;
          re.mapidx[st][sn * re.max_n_kmers + (r_idx - re.min_kmer_pos)] = P.Hflag ?
              kmer_to_mapidx_hash(P.seeds[sn], P.max_seed_span, [&](llint q) { return EXTRACT(r, q); }, i) :
              kmer_to_mapidx(P.seeds[sn], [&](llint q) { return EXTRACT(r, q); }, i);
        }
    }
  }

  // read_get_region_counts (mapping.c:459-542); bit layout RG_* (mapping.c:25-31)
  void read_get_region_counts(ThreadState& T, Read& re, int st) const {
    if (T.region_map_id == 0) {   // mapping.c:471-487: id wrapped -> fresh maps
      T.region_map_id = 1;
      for (int nip = 0; nip < 2; nip++) for (int s = 0; s < 2; s++) std::fill(T.region_map[nip][s].begin(), T.region_map[nip][s].end(), 0);
    }
    uint16_t* rm = T.region_map[re.first_in_pair || !re.paired ? 0 : 1][st].data();
    auto mark = [&](int region) {
      if ((rm[region] >> 3) == T.region_map_id) rm[region] |= 0x1;
      else rm[region] = (uint16_t)((T.region_map_id << 3) + 0x6);
    };
    int ns = (int)P.seeds.size();
    for (int sn = 0; sn < ns; sn++)
      for (int i = 0; re.min_kmer_pos + i + P.seeds[sn].span - 1 < re.read_len; i++) {
        uint32_t mi = re.mapidx[st][sn * re.max_n_kmers + i];
        uint32_t len = I->list_len(sn, mi);
        if (len > P.list_cutoff) continue;
        const uint32_t* l = I->list(sn, mi);
        for (uint32_t j = 0; j < len; j++) {
          int region = (int)(l[j] >> P.region_bits);
          mark(region);
          if ((l[j] & ((1u << P.region_bits) - 1)) < (uint32_t)P.region_overlap && region > 0) mark(region - 1);
        }
      }
  }

  // heap_uu (common/heap.h:44-139, DEF_HEAP(uint32_t, uint, uu)) -- exact sift rules matter for tie order
  struct HeapUU {
    struct E { uint32_t key; uint32_t rest; };
    std::vector<E> a; uint32_t load = 0;
    void percolate_up(uint32_t node) {
      uint32_t parent = node / 2;
      while (node > 1 && a[node - 1].key < a[parent - 1].key) { std::swap(a[parent - 1], a[node - 1]); node = parent; parent = node / 2; }
    }
    void percolate_down(uint32_t node) {
      for (;;) {
        uint32_t left = node * 2, right = left + 1, mn = node;
        if (left <= load && a[left - 1].key < a[node - 1].key) mn = left;
        if (right <= load && a[right - 1].key < a[mn - 1].key) mn = right;
        if (mn == node) break;
        std::swap(a[mn - 1], a[node - 1]); node = mn;
      }
    }
    void insert(E e) { a[load] = e; load++; percolate_up(load); }
    void replace_min(E e) { a[0] = e; percolate_down(1); }
    void extract_min() { load--; if (load > 0) { a[0] = a[load]; percolate_down(1); } }
  };

  // advance_index_in_genomemap (mapping.c:646-805), unpaired branch (use_mp_region_counts == 0)
  void advance_index(const ThreadState& T, int nip, int st, uint32_t* idx, uint32_t max_idx, const uint32_t* map, int mp_mode = 0) const {
    const uint16_t* rm = T.region_map[nip][st].data();
    auto mp_ok = [&](int region) {                                           // mapping.c:733-742
      const int count_main = (rm[region] & 0x1) ? 2 : 1, count_mp = (rm[region] >> 1) & 0x3;
      return (mp_mode == 1 && count_main >= 2 && count_mp >= 2) || (mp_mode == 2 && (count_main >= 2 || count_mp >= 2)) ||
             (mp_mode == 3 && count_mp >= 1 && count_main + count_mp >= 3);
    };
    while (mp_mode && *idx < max_idx) {
      int region = (int)(map[*idx] >> P.region_bits);
      if (mp_ok(region)) break;
      if (region > 0 && (map[*idx] & ((1u << P.region_bits) - 1)) < (uint32_t)P.region_overlap && mp_ok(region - 1)) break;
      (*idx)++;
    }
    if (mp_mode) return;
    while (*idx < max_idx) {
      int region = (int)(map[*idx] >> P.region_bits);
      if (rm[region] & 0x1) break;
      if (region > 0 && (map[*idx] & ((1u << P.region_bits) - 1)) < (uint32_t)P.region_overlap) {
        region--;
        if (rm[region] & 0x1) break;
      }
      (*idx)++;
    }
  }

  // read_get_anchor_list_per_strand (mapping.c:861-1006), collapse = true, use_region_counts = (match_mode == 2)
  void read_get_anchor_list(const ThreadState& T, Read& re, int st) const {
    int ns = (int)P.seeds.size();
    // unpaired: use_region_counts = (match_mode == 2) (gmapper.c:2615); paired default (mode 4, half-paired): true, no mp counts (:2652-2660)
    bool use_region_counts = re.paired ? (P.mp_match_mode != 2) : (P.match_mode == 2);    // paired -n 2: no region counts at all (gmapper.c:2652-2657)
    const int nip = (re.first_in_pair || !re.paired) ? 0 : 1;
    const int mp_mode = re.paired ? mp_region_mode() : 0;
    re.anchors[st].clear();
    if (re.mapidx[st].empty()) return;
    if (!re.paired && ((st == 0 && !P.Fflag) || (st == 1 && !P.Cflag))) return;                    // mapping.c:879-880
    HeapUU h; h.a.resize((size_t)ns * re.max_n_kmers + 1);
    std::vector<uint32_t> idx((size_t)ns * re.max_n_kmers, 0);
    std::vector<int> anchor_cache(re.read_len, -1);
    for (int sn = 0; sn < ns; sn++)
      for (int i = 0; re.min_kmer_pos + i + P.seeds[sn].span - 1 < re.read_len; i++) {
        uint32_t off = sn * re.max_n_kmers + i;
        uint32_t mi = re.mapidx[st][off];
        uint32_t len = I->list_len(sn, mi); const uint32_t* l = I->list(sn, mi);
        if (len > P.list_cutoff) idx[off] = len;
        if (use_region_counts) advance_index(T, nip, st, &idx[off], len, l, mp_mode);
        if (idx[off] < len) { h.insert({l[idx[off]], off}); idx[off]++; }
      }
    std::vector<Anchor>& A = re.anchors[st];
    while (h.load > 0) {
      HeapUU::E tmp = h.a[0];
      uint32_t off = tmp.rest; int sn = off / re.max_n_kmers; int i = off % re.max_n_kmers;
      Anchor an; an.x = tmp.key; an.y = re.min_kmer_pos + i; an.length = P.seeds[sn].span; an.width = 1; an.weight = 1;
      an.cn = get_contig_num(*G, (uint32_t)an.x);
      A.push_back(an);
      {  // collapse (mapping.c:957-971)
        int n = (int)A.size();
        uint32_t diag = (uint32_t)((A[n - 1].x + re.read_len - A[n - 1].y) % re.read_len);
        int j = anchor_cache[diag];
        if (j >= 0 && A[j].cn == A[n - 1].cn && anchor_uw_colinear(&A[j], &A[n - 1])) { anchor_uw_join(&A[j], &A[n - 1]); A.pop_back(); }
        else anchor_cache[diag] = n - 1;
      }
      uint32_t mi = re.mapidx[st][off];
      uint32_t len = I->list_len(sn, mi); const uint32_t* l = I->list(sn, mi);
      if (use_region_counts) advance_index(T, nip, st, &idx[off], len, l, mp_mode);
      if (idx[off] < len) { h.replace_min({l[idx[off]], off}); idx[off]++; }
      else h.extract_min();
    }
  }

  // read_get_hit_list_per_strand (mapping.c:1025-1229), gapless = false, match_mode 1 or 2
  // hit_list.match_mode: unpaired 1 / 2 (gmapper.c:2619); paired: -n 4 -> 2, -n 3 -> 3, -n 2 -> 1 (gmapper.c:2666-2668)
  int hit_match_mode(const Read& re) const { return !re.paired ? P.match_mode : (P.mp_match_mode == 3 ? 3 : (P.mp_match_mode == 2 ? 1 : 2)); }
  // pass1.min_matches of the paired option set (gmapper.c:2673): 2 for -n 4, else 1; the unpaired sets (also the half-paired rescue, :2708) keep match_mode / 2
  int pair_min_matches() const { return (P.mp_match_mode == 3 || P.mp_match_mode == 2) ? 1 : 2; }
  void read_get_hit_list(Read& re, int st, const ThreadState* T = nullptr) const {
    const int hmode = hit_match_mode(re);
    std::vector<Anchor>& A = re.anchors[st];
    std::vector<Hit>& H = re.hits[st];
    H.clear();
    int n = (int)A.size();
    for (int i = 0; i < n; i++) {
      int cn = A[i].cn;
      int w_len = re.window_len;
      if ((uint32_t)w_len > G->len[cn]) w_len = (int)G->len[cn];
      llint gend = (A[i].x - G->offsets[cn]) + re.read_len - 1 - A[i].y;
      if (gend > (llint)(uint32_t)(G->len[cn] - 1)) gend = (llint)(uint32_t)(G->len[cn] - 1);
      llint gstart = (gend >= re.window_len) ? gend - re.window_len : 0;
      int max_idx = i;
      int max_score = A[i].length * P.match_score;
      bool heavy_mp = false;
      if (hmode == 3) {                                            // mapping.c:1080-1093: the mate has a region marked twice within reach of this anchor's region
        const uint16_t* rm = T->region_map[re.first_in_pair ? 0 : 1][st].data();
        int region = (int)(A[i].x >> P.region_bits);
        heavy_mp = ((rm[region] >> 1) & 0x3) >= 2;
        if (!heavy_mp && region > 0 && (A[i].x & ((1u << P.region_bits) - 1)) < (uint32_t)P.region_overlap) heavy_mp = ((rm[region - 1] >> 1) & 0x3) >= 2;
      }
      if (!P.gapless && (hmode == 2 || (hmode == 3 && !heavy_mp)) && A[i].weight == 1) max_score = -1;
      for (int j = i - 1; !P.gapless && j >= 0 && A[j].x >= (llint)G->offsets[cn] + gstart; j--) {   // gapless: only the anchor itself (mapping.c:1095)
        if (A[j].y >= A[i].y) continue;
        int short_len, long_len;
        if (A[i].x - (llint)G->offsets[cn] - A[i].y > A[j].x - (llint)G->offsets[cn] - A[j].y) {
          short_len = (int)(A[i].y - A[j].y) + A[i].length; long_len = (int)(A[i].x - A[j].x) + A[i].length;
        } else {
          short_len = (int)(A[i].x - A[j].x) + A[i].length; long_len = (int)(A[i].y - A[j].y) + A[i].length;
        }
        int tmp_score;
        if (long_len > short_len) tmp_score = short_len * P.match_score + P.b_gap_open_score + (long_len - short_len) * P.b_gap_extend_score;
        else tmp_score = short_len * P.match_score;
        if (tmp_score > max_score) { max_idx = j; max_score = tmp_score; }
      }
      if (P.gapless || hmode == 1 || (hmode == 3 && heavy_mp) ||                                  // mapping.c:1153-1157
          max_score >= (int)GMO_ABS_OR_PCT(P.window_gen_threshold, (re.read_len < w_len ? re.read_len : w_len) * P.match_score)) {
        int x_len = (int)(A[i].x - A[max_idx].x) + A[i].length;
        llint goff;
        if ((re.window_len - x_len) / 2 < A[max_idx].x - G->offsets[cn]) goff = (A[max_idx].x - G->offsets[cn]) - (re.window_len - x_len) / 2;
        else goff = 0;
        if (goff + w_len > (llint)G->len[cn]) goff = (llint)(uint32_t)(G->len[cn] - (uint32_t)w_len);
        Anchor a[3];
        if (max_idx < i) {
          a[0] = A[i]; a[0].x -= (llint)G->offsets[cn] + goff;
          a[1] = A[max_idx]; a[1].x -= (llint)G->offsets[cn] + goff;
          anchor_join(a, 2, &a[2]);
        } else { a[2] = A[i]; a[2].x -= (llint)G->offsets[cn] + goff; }
        Hit h;
        h.g_off = goff; h.g_off_pos_strand = goff; h.w_len = w_len; h.cn = cn; h.st = st; h.gen_st = 0;
        h.anchor = a[2]; h.score_window_gen = max_score;
        h.matches = (max_idx == i ? A[i].weight : A[i].weight + A[max_idx].weight);
        h.score_vector = -1; h.score_full = -1;
        h.score_max = (re.read_len < w_len ? re.read_len : w_len) * P.match_score;
        H.push_back(h);
      }
    }
    // insertion sort by g_off within contig (mapping.c:1210-1223)
    for (int i = 1; i < (int)H.size(); i++) {
      int j = i;
      while (j >= 1 && H[j - 1].cn == H[i].cn && H[j - 1].g_off > H[i].g_off) j--;
      if (j < i) { Hit tmp = H[i]; for (int k = i - 1; k >= j; k--) H[k + 1] = H[k]; H[j] = tmp; }
    }
  }

  // f1_run (common/f1-wrapper.h:97-134), gapped branch; genome_ls != nullptr selects the colour-space filter
  int f1_run(ThreadState& T, const uint32_t* genome, llint goff, int wlen, const uint32_t* read, int rlen, uint32_t tag,
             const uint32_t* genome_ls = nullptr, int initbp = -1, bool gapless_call = false, int gapless_glen = 0, int gapless_g_idx = 0,
             int gapless_r_idx = 0) const {
    uint32_t hv = 0;
    if (P.hash_filter_calls && tag != 0) {
      hv = hash_genome_window(genome, (uint32_t)goff, (uint32_t)wlen) % f1_window_cache_size;
      if (T.f1_tag[hv] == tag) { T.stats.vec_bypassed++; return (int)T.f1_score[hv]; }
    }
    int score = gapless_call ? sw_gapless(P, genome, gapless_glen, read, rlen, gapless_g_idx, gapless_r_idx, genome_ls, initbp, G->is_rna) :
                genome_ls ? sw_vector_cs(P, P.match_score + P.crossover_score, genome, goff, wlen, read, rlen, genome_ls, initbp, G->is_rna)   // gmapper.c:2935; genome_is_rna: mapping.c:1318
                          : sw_vector(P, genome, goff, wlen, read, rlen);
    T.stats.vec_calls++; T.stats.vec_cells += (uint64_t)wlen * rlen;
    if (P.hash_filter_calls && tag != 0) { T.f1_tag[hv] = tag; T.f1_score[hv] = (uint32_t)score; }
    return score;
  }

  // read_pass1_per_strand (mapping.c:1261-1339), letter space, only_paired = false
  void read_pass1(ThreadState& T, Read& re, int st, bool only_paired = false, int min_matches = -1) const {
    if (min_matches < 0) min_matches = P.match_mode;
    int last_good_cn = -1; unsigned int last_good_g_off = 0;
    T.f1_hash_tag++;
    for (auto& h : re.hits[st]) {
      if (only_paired && h.pair_min < 0) continue;                 // mapping.c:1271-1273
      if (h.matches < min_matches) continue;   // pass1.min_matches = match_mode (gmapper.c:2625); paired: gmapper.c:2673
      if (h.saved == 1) { last_good_cn = h.cn; last_good_g_off = (unsigned int)h.g_off_pos_strand; continue; }
      if (last_good_cn >= 0 && h.cn == last_good_cn &&
          h.g_off_pos_strand + (unsigned int)GMO_ABS_OR_PCT(P.window_overlap, re.window_len) <= (llint)(unsigned int)(last_good_g_off + re.window_len)) {
        h.score_vector = 0; h.pct_score_vector = 0; continue;
      }
      if (h.score_vector <= 0) {
        if (P.colour) {   // mapping.c:1297-1319: the hit is first turned onto the read's input strand
          if (h.st != re.input_strand) reverse_hit(re, h);
          const uint32_t* gen_cs = (h.gen_st == 0 ? G->cs_fwd[h.cn].data() : G->cs_rc[h.cn].data());
          const uint32_t* gen_ls = (h.gen_st == 0 ? G->fwd[h.cn].data() : G->rc[h.cn].data());
          h.score_vector = f1_run(T, gen_cs, h.g_off, h.w_len, re.bits[h.st].data(), re.read_len, T.f1_hash_tag, gen_ls, re.initbp[st],
                                  P.gapless, (int)G->len[h.cn], (int)(h.g_off + h.anchor.x), (int)h.anchor.y);
        } else
        h.score_vector = f1_run(T, G->fwd[h.cn].data(), h.g_off, h.w_len, re.bits[st].data(), re.read_len, T.f1_hash_tag, nullptr, -1,
                                P.gapless, (int)G->len[h.cn], (int)(h.g_off + h.anchor.x), (int)h.anchor.y);     // mapping.c:1321-1328
        h.pct_score_vector = (1000 * 100 * h.score_vector) / h.score_max;
        if (h.score_vector >= (int)GMO_ABS_OR_PCT(P.sw_vect_threshold, h.score_max)) { last_good_cn = h.cn; last_good_g_off = (unsigned int)h.g_off_pos_strand; }
      }
    }
  }

  // extheap_unpaired_pass1 (common/heap.h:226-327; CMP mapping.c:1369) over Hit pointers
  static void xh_up(std::vector<Hit*>& a, int node) {
    int parent = node / 2;
    while (node > 1 && a[node - 1]->pass1_key < a[parent - 1]->pass1_key) { std::swap(a[parent - 1], a[node - 1]); node = parent; parent = node / 2; }
  }
  static void xh_down(std::vector<Hit*>& a, int load, int node) {
    for (;;) {
      int left = node * 2, right = left + 1, mn = node;
      if (left <= load && a[left - 1]->pass1_key < a[node - 1]->pass1_key) mn = left;
      if (right <= load && a[right - 1]->pass1_key < a[mn - 1]->pass1_key) mn = right;
      if (mn == node) break;
      std::swap(a[mn - 1], a[node - 1]); node = mn;
    }
  }

  // read_get_vector_hits (mapping.c:1376-1411)
  void read_get_vector_hits(Read& re, std::vector<Hit*>& a, int& load) const {
    a.assign(P.num_tmp_outputs, nullptr); load = 0;
    bool absthr = GMO_IS_ABSOLUTE(P.sw_vect_threshold);
    for (int st = 0; st < 2; st++)
      for (auto& h : re.hits[st]) {
        if (h.saved == 1) continue;
        if (h.score_vector >= (int)GMO_ABS_OR_PCT(P.sw_vect_threshold, h.score_max) &&
            (load < P.num_tmp_outputs || (absthr ? h.score_vector > a[0]->pass1_key : h.pct_score_vector > a[0]->pass1_key))) {
          h.pass1_key = absthr ? h.score_vector : h.pct_score_vector;
          if (load < P.num_tmp_outputs) { a[load] = &h; load++; xh_up(a, load); }
          else { a[0] = &h; xh_down(a, load, 1); }
        }
      }
  }

  // reverse_hit (mapping.c:254-263)
  void reverse_hit(const Read& re, Hit& h) const {
    h.g_off = (llint)G->len[h.cn] - h.g_off - h.w_len;
    anchor_reverse(&h.anchor, h.w_len, re.read_len);
    h.gen_st = 1 - h.gen_st; h.st = 1 - h.st;
  }

  // hit_run_full_sw (mapping.c:331-402), letter space
  void hit_run_full_sw(ThreadState& T, const Read& re, Hit& h, int thresh) const {
    if (h.st != re.input_strand) reverse_hit(re, h);
    const uint32_t* gen = (h.gen_st == 0 ? G->fwd[h.cn].data() : G->rc[h.cn].data());
    h.has_sfr = true; h.sfr = SwFullResults();
    if (P.colour) {   // mapping.c:375-379
      CsParams C; C.match = P.match_score; C.mismatch = P.mismatch_score; C.xover = P.crossover_score;
      C.a_go = -P.a_gap_open_score; C.a_ge = -P.a_gap_extend_score; C.b_go = -P.b_gap_open_score; C.b_ge = -P.b_gap_extend_score;
      C.anchor_width = P.anchor_width; C.indel_taboo_len = P.indel_taboo_len;
      T.stats.full_calls++;
      sw_full_cs(C, gen, h.g_off, h.w_len, re.bits[h.st].data(), re.read_len, re.initbp[h.st], thresh, &h.sfr, h.gen_st && P.Tflag, &h.anchor, 1,
                 re.crossover_score.empty() ? nullptr : re.crossover_score.data(), P.Gflag ? 0 : 1, G->is_rna);   // mapping.c:375-379
      h.score_full = h.sfr.score;
      h.pct_score_full = (1000 * 100 * h.score_full) / h.score_max;
      return;
    }
    h.score_vector = sw_vector(P, gen, h.g_off, h.w_len, re.bits[h.st].data(), re.read_len);
    T.stats.vec_calls++; T.stats.vec_cells += (uint64_t)h.w_len * re.read_len;
    if (h.score_vector >= thresh) {
      T.stats.full_calls++;
      sw_full_ls(P, T.sww, gen, h.g_off, h.w_len, re.bits[h.st].data(), re.read_len, thresh, h.score_vector, &h.sfr,
                 h.gen_st && P.Tflag, &h.anchor, 1, P.Gflag ? 0 : 1);
    } else h.sfr.score = 0;
    h.score_full = h.sfr.score;
    h.pct_score_full = (1000 * 100 * h.score_full) / h.score_max;
  }

  // hit_run_post_sw (mapping.c:1609-1625), letter space
  void hit_run_post_sw(const Read& re, Hit& h) const {
    SwFullResults& s = h.sfr;
    if (P.colour) post_sw(P, re.bits[h.st].data(), re.initbp[h.st], &s, P.Qflag ? re.qual.c_str() : nullptr);
    else s.posterior = pow(2.0, ((double)s.score - (double)s.rmapped * (2.0 * P.score_alpha + P.score_beta)) / P.score_alpha);
    s.posterior_score = (int)rint(P.score_alpha * log(s.posterior) / log(2.0) + (double)s.rmapped * (2.0 * P.score_alpha + P.score_beta));
    if (s.posterior_score < 0) s.posterior_score = 0;
    s.pct_posterior_score = (1000 * 100 * s.posterior_score) / h.score_max;
    h.score_full = s.posterior_score; h.pct_score_full = s.pct_posterior_score;
  }

  static int cmp_gen_start(const Hit* a, const Hit* b) {  // mapping.c:1485-1494
    if (a->cn != b->cn) return a->cn - b->cn;
    if (a->gen_st != b->gen_st) return a->gen_st - b->gen_st;
    return a->sfr.genome_start - b->sfr.genome_start;
  }
  static int cmp_gen_end(const Hit* a, const Hit* b) {    // mapping.c:1496-1506
    if (a->cn != b->cn) return a->cn - b->cn;
    if (a->gen_st != b->gen_st) return a->gen_st - b->gen_st;
    return (-a->sfr.genome_start - a->sfr.rmapped + a->sfr.deletions - a->sfr.insertions) -
           (-b->sfr.genome_start - b->sfr.rmapped + b->sfr.deletions - b->sfr.insertions);
  }
  // read_remove_duplicate_hits (mapping.c:1520-1606); glibc qsort == stable merge sort on these sizes
  template <class Cmp>
  static void dedup_pass(std::vector<Hit*>& v, Cmp cmp, uint64_t& pruned) {
    std::stable_sort(v.begin(), v.end(), [&](const Hit* a, const Hit* b) { return cmp(a, b) < 0; });
    size_t i = 0, k = 0, n = v.size();
    while (i < n) {
      int mx = v[i]->pass2_key; size_t mx_idx = i, j = i + 1;
      while (j < n && !cmp(v[i], v[j])) { if (v[j]->pass2_key > mx) { mx = v[j]->pass2_key; mx_idx = j; } j++; }
      if (mx_idx != k) v[k] = v[mx_idx];
      k++; i = j;
    }
    pruned += n - k; v.resize(k);
  }

  // read_pass2 (mapping.c:1631-1750)
  void read_pass2(ThreadState& T, Read& re, std::vector<Hit*>& p1, int n1, std::vector<Hit*>& p2) const {
    p2.clear();
    for (int i = 0; i < n1; i++) {
      Hit* rh = p1[i];
      if (rh->score_full < 0 || !rh->has_sfr) {
        hit_run_full_sw(T, re, *rh, (int)GMO_ABS_OR_PCT(P.sw_full_threshold, rh->score_max));
        if (P.compute_mapping_qualities && rh->score_full > 0) hit_run_post_sw(re, *rh);
        rh->pass2_key = GMO_IS_ABSOLUTE(P.sw_full_threshold) ? rh->score_full : (int)rh->pct_score_full;
      }
      if (rh->score_full >= GMO_ABS_OR_PCT(P.sw_full_threshold, rh->score_max)) p2.push_back(rh);
    }
    dedup_pass(p2, cmp_gen_start, T.stats.dup_pruned);
    dedup_pass(p2, cmp_gen_end, T.stats.dup_pruned);
    std::stable_sort(p2.begin(), p2.end(), [](const Hit* a, const Hit* b) { return (b->pass2_key - a->pass2_key) < 0; });  // mapping.c:1479-1482
    if ((int)p2.size() > P.num_outputs) p2.resize(P.num_outputs);
    if (P.strata && !p2.empty()) { size_t i = 1; while (i < p2.size() && p2[0]->score_full == p2[i]->score_full) i++; p2.resize(i); }
    if (!p2.empty()) {
      if (P.max_alignments == 0 || (int)p2.size() <= P.max_alignments) T.stats.reads_matched++;
      else p2.clear();
    }
    for (auto* h : p2) h->saved = 1;
  }

  // ---- SAM emission (gmapper/output.c:15-64,164-220,227-774,777-793,955-1008) ----
  static std::string make_cigar(int read_start, int read_end, int read_length, const std::string& qr, const std::string& db,
                                std::vector<std::pair<int, char>>* out) {
    std::vector<std::pair<int, char>>& c = *out; c.clear();
    if (read_start > 1) c.push_back({read_start - 1, 'S'});
    int i = 0, n = (int)qr.size();
    while (i < n) {
      int length; char op;
      if (qr[i] == '-') { for (length = 0; i + length < n && qr[i + length] == '-'; length++); op = 'D'; }
      else if (db[i] == '-') { for (length = 0; i + length < n && db[i + length] == '-'; length++); op = 'I'; }
      else { for (length = 0; i + length < n && db[i + length] != '-' && qr[i + length] != '-'; length++); op = 'M'; }
      c.push_back({length, op}); i += length;
    }
    if (read_end != read_length) c.push_back({read_length - read_end, 'S'});
    return std::string();
  }
  static char rc_char(char ch) {  // reverse(), output.c:164-220
    switch (ch) {
      case 'A': return 'T'; case 'a': return 't'; case 'T': return 'A'; case 't': return 'a';
      case 'C': return 'G'; case 'c': return 'g'; case 'G': return 'C'; case 'g': return 'c';
      case '-': return '-'; case 'N': return 'N'; case 'n': return 'n'; case '.': return '.';
      case 'R': return 'Y'; case 'r': return 'y'; case 'Y': return 'R'; case 'y': return 'r';
      case 'S': return 'S'; case 's': return 's'; case 'W': return 'W'; case 'w': return 'w';
      case 'K': return 'M'; case 'k': return 'm'; case 'M': return 'K'; case 'm': return 'k';
      case 'B': return 'V'; case 'b': return 'v'; case 'V': return 'B'; case 'v': return 'b';
      case 'D': return 'H'; case 'd': return 'h'; case 'H': return 'D'; case 'h': return 'd';
      default: return '?';
    }
  }
  static int qv_from_pr_corr(double pr_corr) {  // util.h:267-283
    double pr_err = 1 - pr_corr;
    if (pr_err > .99999999) return 0; else if (pr_err < 1E-25) return 250;
    return (int)(-10.0 * log(pr_err) / log(10.0));
  }
  static int double_to_neglog(double x) { return (int)((double)1000 * -log(x)); }  // util.h:297-301

  void hit_output(const Read& re, const Hit* rh, std::string& out) const {
    char buf[256];
    std::string seq(re.read_len, 'N');
    if (!P.colour) {
      for (int i = 0; i < re.read_len; i++) {   // output.c:320-352
        char c = re.seq[i];
        switch (c) {
          case 'R': case 'Y': case 'S': case 'W': case 'K': case 'M': case 'B': case 'D': case 'H': case 'V': seq[i] = 'N'; break;
          default: if (c >= 'a') c -= 32; seq[i] = c; break;
        }
      }
    } else seq = "*";                           // output.c:353-355
    if (rh == nullptr) {                        // unmapped (output.c:411-466), unpaired
      out += re.name; out += "\t4\t*\t0\t0\t*\t*\t0\t0\t"; out += seq; out += "\t";
      out += (P.Qflag && !P.colour) ? re.qual : std::string("*");         // output.c:419-421: as read, no offset conversion
      if (P.colour) { out += "\tCQ:Z:"; out += P.Qflag ? re.qual : std::string("*"); out += "\tCS:Z:"; out += re.seq; }   // output.c:441-451
      out += "\n";
      return;
    }
    const SwFullResults& s = rh->sfr;
    bool reverse_strand = (rh->gen_st == 1);
    int read_start = s.read_start + 1, read_end = read_start + s.rmapped - 1;
    int genome_length = (int)G->len[rh->cn];
    std::vector<std::pair<int, char>> cigar;
    make_cigar(read_start, read_end, re.read_len, s.qralign, s.dbalign, &cigar);
    int j = P.colour ? 0 : read_start - 1;      // output.c:485-493: colour space prints the aligned part only
    if (P.colour) seq.assign(read_end - read_start + 1, 'N');
    for (size_t i = 0; i < s.qralign.size(); i++) {   // output.c:494-533
      char c = s.qralign[i];
      if (c != '-') {
        if (c >= 'a') c -= 32;
        if (c != 'A' && c != 'G' && c != 'C' && c != 'T' && c != 'N') c = 'N';
        seq[j++] = c;
      }
    }
    std::string qual = "*";
    if (!P.colour && P.Qflag) {                 // output.c:539-570: reversed with the read, re-based to PHRED+33
      qual = re.qual;
      if (reverse_strand) std::reverse(qual.begin(), qual.end());
      if (P.qual_delta != 33) for (auto& c : qual) c = (char)(c - P.qual_delta + 33);
    }
    if (!P.colour) seq.resize(j + (re.read_len - read_end));
    else {                                      // output.c:572-580: hard clips.  QUAL stays "*" unless the reads came with QVs:
      for (auto& c : cigar) if (c.second == 'S') c.second = 'H';   // then it is post_sw's base qualities (:581-621, Qflag)
      if (P.Qflag && P.compute_mapping_qualities) {
        qual = s.qual;
        if (reverse_strand) for (int i = 0; i < s.rmapped / 2; i++) std::swap(qual[i], qual[s.rmapped - i - 1]);
      }
    }
    int genome_start;
    if (!reverse_strand) genome_start = s.genome_start + 1;
    else {
      int right = genome_length - s.genome_start;
      genome_start = right - (read_end - read_start - s.deletions + s.insertions);
      std::string t(seq.size(), ' ');
      for (size_t i = 0; i < seq.size(); i++) t[seq.size() - 1 - i] = rc_char(seq[i]);
      seq = t;
      std::reverse(cigar.begin(), cigar.end());
    }
    int flag = reverse_strand ? 0x10 : 0;
    out += re.name;
    snprintf(buf, sizeof buf, "\t%i\t", flag); out += buf;
    out += G->names[rh->cn];
    snprintf(buf, sizeof buf, "\t%u\t%i\t", (unsigned)genome_start, s.mqv); out += buf;
    for (auto& c : cigar) { snprintf(buf, sizeof buf, "%d%c", c.first, c.second); out += buf; }
    out += "\t*\t0\t0\t"; out += seq; out += "\t"; out += qual;
    snprintf(buf, sizeof buf, "\tAS:i:%d", rh->score_full); out += buf;
    if (P.compute_mapping_qualities && !P.all_contigs) { snprintf(buf, sizeof buf, "\tZ0:i:%d\tZ1:i:%d", double_to_neglog(s.z0), double_to_neglog(s.z1)); out += buf; }   // output.c:691
    snprintf(buf, sizeof buf, "\tNM:i:%d", s.mismatches + s.deletions + s.insertions); out += buf;
    if (P.colour) {                             // output.c:717-730
      if (P.Qflag) { out += "\tCQ:Z:"; out += re.qual; }
      out += "\tCS:Z:"; out += re.seq;
      snprintf(buf, sizeof buf, "\tCM:i:%d\tXX:Z:", s.crossovers); out += buf; out += s.qralign;
    }
    out += "\n";
  }

  // read_output (output.c:955-1008) + compute_unpaired_mqv (output.c:777-793)
  void read_output(const Read& re, std::vector<Hit*>& p2, std::string& out) const {
    if (p2.empty()) return;
    if (P.compute_mapping_qualities) {
      double z1 = 0.0;
      for (auto* h : p2) z1 += h->sfr.posterior;
      for (auto* h : p2) {
        h->sfr.z0 = h->sfr.posterior; h->sfr.z1 = z1;
        h->sfr.mqv = qv_from_pr_corr(h->sfr.posterior / z1);
        if (h->sfr.mqv < 4) h->sfr.mqv = 0;
      }
      if (P.single_best_mapping) {                                 // output.c:977-984: the first mapping with the highest quality
        size_t mx = 0;
        for (size_t i = 1; i < p2.size(); i++) if (p2[i]->sfr.mqv > p2[mx]->sfr.mqv) mx = i;
        hit_output(re, p2[mx], out);
        return;
      }
    }
    for (auto* h : p2) hit_output(re, h, out);
  }

  // =============================================================================================
  // Paired mode (default option sets: match_mode 4, half-paired, mapping qualities on;
  // gmapper.c:2636-2720).  All of it follows gmapper/mapping.c:266-325,405-456,1871-2636 and
  // gmapper/output.c:795-942,1070-1291.
  // =============================================================================================
  struct HitPair {                  // struct read_hit_pair, gmapper-definitions.h:162-173
    Hit* rh[2] = {nullptr, nullptr}; int rh_idx[2] = {-1, -1};
    int score_max = 0, score = 0, pct_score = 0, key = 0, insert_size = 0; bool improper_mapping = false;
  };
  struct PairEntry {                // pair_entry, gmapper-definitions.h:186-193
    Read* re[2]; std::vector<Hit> pool[2]; std::vector<HitPair> final_paired_hits; bool mapped = false;
  };

  // readpair_compute_mp_ranges (mapping.c:2317-2442); only the delta_g_off part is used without mp region counts
  void readpair_compute_mp_ranges(Read& re1, Read& re2) const {
    const int mn = P.min_insert_size, mx = P.max_insert_size;
    int a = mn - re2.window_len, b = mx + (re1.window_len - re1.read_len) - re2.read_len;
    int c = -mx + re1.read_len + (re2.read_len - re2.window_len), d = -mn + re1.window_len;
    switch (P.pair_mode) {
      case 1: break;
      case 2: a += re1.read_len + re2.read_len; b += re1.read_len + re2.read_len; c -= re1.read_len + re2.read_len; d -= re1.read_len + re2.read_len; break;
      case 3: a += re2.read_len; b += re2.read_len; c -= re2.read_len; d -= re2.read_len; break;
      case 4: a += re1.read_len; b += re1.read_len; c -= re1.read_len; d -= re1.read_len; break;
      default: assert(0);
    }
    re1.delta_g_off_min[0] = a; re1.delta_g_off_max[0] = b; re1.delta_g_off_min[1] = c; re1.delta_g_off_max[1] = d;
    if (P.pair_mode == 1 || P.pair_mode == 2) {
      re2.delta_g_off_min[0] = -d; re2.delta_g_off_max[0] = -c; re2.delta_g_off_min[1] = -b; re2.delta_g_off_max[1] = -a;
    } else {
      re2.delta_g_off_min[0] = -b; re2.delta_g_off_max[0] = -a; re2.delta_g_off_min[1] = -d; re2.delta_g_off_max[1] = -c;
    }
    const int R = 1 << P.region_bits;                                        // mapping.c:2422-2430
    for (Read* re : {&re1, &re2}) for (int st = 0; st < 2; st++) {
      const int mn = re->delta_g_off_min[st], mx = re->delta_g_off_max[st];
      re->delta_region_min[st] = mn >= 0 ? mn / R : -1 - (-mn - 1) / R;
      re->delta_region_max[st] = mx > 0 ? 1 + (mx - 1) / R : -(-mx / R);
    }
  }

  // use_mp_region_counts of the paired option set (gmapper.c:2657-2662): 1 = match mode 4 without half-paired, 2 / 3 = match mode 3 with / without
  int mp_region_mode() const { return P.mp_match_mode == 4 ? (P.half_paired ? 0 : 1) : (P.mp_match_mode == 3 ? (P.half_paired ? 2 : 3) : 0); }

  // read_get_mp_region_counts (mapping.c:545-608): for every region the read's list entries mark, the best count of the mate's regions (other
  // strand) within the insert-size range: 0 none, 1 marked once, 2 marked twice or more
  void read_get_mp_region_counts(ThreadState& T, Read& re, int st) const {
    const int nip = re.first_in_pair ? 0 : 1;
    uint16_t* rm = T.region_map[nip][st].data();
    const uint16_t* mp = T.region_map[1 - nip][1 - st].data();
    const int n_regions = (int)T.region_map[nip][st].size();
    auto set_cnt = [&](int region) {
      if ((rm[region] & 0x6) != 0x6) return;                                 // RG_VALID_MP_CNT
      int first = std::max(0, region + re.delta_region_min[st]), last = std::min(n_regions - 1, region + re.delta_region_max[st]), mx = 0;
      for (int k = first; k <= last && mx < 2; k++) if ((mp[k] >> 3) == T.region_map_id) mx = (mp[k] & 0x1) ? 2 : 1;
      rm[region] = (uint16_t)((rm[region] & ~0x6) | (mx << 1));             // RG_SET_MP_CNT
    };
    int ns = (int)P.seeds.size();
    for (int sn = 0; sn < ns; sn++)
      for (int i = 0; re.min_kmer_pos + i + P.seeds[sn].span - 1 < re.read_len; i++) {
        uint32_t mi = re.mapidx[st][sn * re.max_n_kmers + i];
        uint32_t len = I->list_len(sn, mi);
        if (len > P.list_cutoff) continue;
        const uint32_t* l = I->list(sn, mi);
        for (uint32_t j = 0; j < len; j++) {
          int region = (int)(l[j] >> P.region_bits);
          set_cnt(region);
          if (region > 0 && (l[j] & ((1u << P.region_bits) - 1)) < (uint32_t)P.region_overlap) set_cnt(region - 1);
        }
      }
  }

  // readpair_pair_up_hits (mapping.c:266-325)
  void readpair_pair_up_hits(Read& re1, Read& re2) const {
    for (int st1 = 0; st1 < 2; st1++) {
      int st2 = 1 - st1;
      int j = 0;
      auto& H1 = re1.hits[st1]; auto& H2 = re2.hits[st2];
      for (int i = 0; i < (int)H1.size(); i++) {
        while (j < (int)H2.size() && (H2[j].cn < H1[i].cn ||
               (H2[j].cn == H1[i].cn && (int64_t)H2[j].g_off < (int64_t)H1[i].g_off + (int64_t)re1.delta_g_off_min[st1]))) j++;
        int k = j;
        while (k < (int)H2.size() && H2[k].cn == H1[i].cn && (int64_t)H2[k].g_off <= (int64_t)H1[i].g_off + (int64_t)re1.delta_g_off_max[st1]) k++;
        if (j == k) continue;
        H1[i].pair_min = j; H1[i].pair_max = k - 1;
        for (int l = j; l < k; l++) { if (H2[l].pair_min < 0) H2[l].pair_min = i; H2[l].pair_max = i; }
      }
    }
  }

  static void xhp_up(std::vector<HitPair>& a, int node) {
    int parent = node / 2;
    while (node > 1 && a[node - 1].key < a[parent - 1].key) { std::swap(a[parent - 1], a[node - 1]); node = parent; parent = node / 2; }
  }
  static void xhp_down(std::vector<HitPair>& a, int load, int node) {
    for (;;) {
      int left = node * 2, right = left + 1, mn = node;
      if (left <= load && a[left - 1].key < a[node - 1].key) mn = left;
      if (right <= load && a[right - 1].key < a[mn - 1].key) mn = right;
      if (mn == node) break;
      std::swap(a[mn - 1], a[node - 1]); node = mn;
    }
  }

  // readpair_get_vector_hits (mapping.c:1877-1932)
  void readpair_get_vector_hits(Read& re1, Read& re2, std::vector<HitPair>& a, int& load) const {
    a.assign(P.num_tmp_outputs, HitPair()); load = 0;
    const bool absthr = GMO_IS_ABSOLUTE(P.sw_vect_threshold);
    for (int st1 = 0; st1 < 2; st1++) {
      int st2 = 1 - st1;
      for (auto& h1 : re1.hits[st1]) {
        if (h1.saved == 1) continue;
        if (h1.pair_min < 0) continue;
        for (int j = h1.pair_min; j <= h1.pair_max; j++) {
          Hit& h2 = re2.hits[st2][j];
          if (h2.saved == 1) continue;
          HitPair tmp;
          tmp.score = h1.score_vector + h2.score_vector;
          tmp.score_max = h1.score_max + h2.score_max;
          tmp.pct_score = (1000 * 100 * tmp.score) / tmp.score_max;
          tmp.key = absthr ? tmp.score : tmp.pct_score;
          if (tmp.score >= (int)GMO_ABS_OR_PCT(P.sw_vect_threshold, tmp.score_max) && (load < P.num_tmp_outputs || tmp.key > a[0].key)) {
            tmp.rh[0] = &h1; tmp.rh[1] = &h2;
            tmp.insert_size = (int)(st1 == 0 ? h2.g_off - (h1.g_off + h1.w_len) : h1.g_off - (h2.g_off + h2.w_len));
            if (load < P.num_tmp_outputs) { a[load] = tmp; load++; xhp_up(a, load); }
            else { a[0] = tmp; xhp_down(a, load, 1); }
          }
        }
      }
    }
  }

  // get_insert_size (mapping.c:405-456)
  int get_insert_size(const Hit* rh, const Hit* rh_mp) const {
    if (rh_mp == nullptr || rh == nullptr || rh->cn != rh_mp->cn) return 0;
    auto ends = [&](const Hit* h, int* gstart, int* gend) {
      int read_start = h->sfr.read_start + 1, read_end = read_start + h->sfr.rmapped - 1;
      int glen = (int)G->len[h->cn];
      if (h->gen_st != 1) *gstart = h->sfr.genome_start + 1;
      else *gstart = (glen - h->sfr.genome_start) - (read_end - read_start - h->sfr.deletions + h->sfr.insertions);
      *gend = *gstart + h->sfr.gmapped - 1;
    };
    int gs_mp, ge_mp, gs, ge; ends(rh_mp, &gs_mp, &ge_mp); ends(rh, &gs, &ge);
    int fivep = (rh->gen_st == 1) ? ge : gs - 1;
    int fivep_mp = (rh_mp->gen_st == 1) ? ge_mp : gs_mp - 1;
    return fivep_mp - fivep;
  }

  // readpair_compute_paired_hit (mapping.c:2053-2080)
  void readpair_compute_paired_hit(Hit* rh1, Hit* rh2, bool absthr, HitPair* dest) const {
    dest->rh[0] = rh1; dest->rh[1] = rh2;
    dest->score_max = rh1->score_max + rh2->score_max;
    dest->score = rh1->score_full + rh2->score_full;
    dest->pct_score = (1000 * 100 * dest->score) / dest->score_max;
    dest->key = absthr ? dest->score : dest->pct_score;
    int ins_sz = get_insert_size(rh1, rh2);
    int sign;
    if (P.pair_mode == 1 || P.pair_mode == 3) sign = (rh1->gen_st == 0) ? +1 : -1;
    else sign = (rh1->gen_st == 1) ? +1 : -1;
    dest->insert_size = sign * ins_sz;
    dest->improper_mapping = false;
  }

  // readpair_push_dominant_single_hits (mapping.c:2083-2110)
  template <class Cmp>
  void push_dominant(std::vector<HitPair>& v, bool absthr, int nip, Cmp cmp) const {
    std::stable_sort(v.begin(), v.end(), [&](const HitPair& a, const HitPair& b) { return cmp(a, b) < 0; });
    size_t i = 0, n = v.size();
    while (i < n) {
      int mx = v[i].rh[nip]->score_full; size_t mi = i, j = i + 1;
      while (j < n && !cmp(v[i], v[j])) { if (v[j].rh[nip]->score_full > mx) { mx = v[j].rh[nip]->score_full; mi = j; } j++; }
      for (size_t k = i; k < j; k++)
        if (k != mi) { v[k].rh[nip] = v[mi].rh[nip]; readpair_compute_paired_hit(v[k].rh[0], v[k].rh[1], absthr, &v[k]); }
      i = j;
    }
  }
  static int pair_pointer_cmp(const HitPair& a, const HitPair& b) {   // pass2_readpair_pointer_cmp (mapping.c:2004-2050), non-NULL case
    if (a.rh[0]->sort_idx != b.rh[0]->sort_idx) return a.rh[0]->sort_idx - b.rh[0]->sort_idx;
    return a.rh[1]->sort_idx - b.rh[1]->sort_idx;
  }

  // readpair_pass2 (mapping.c:2181-2314)
  void readpair_pass2(ThreadState& T, Read& re1, Read& re2, std::vector<HitPair>& p1, int n1, std::vector<HitPair>& p2) const {
    p2.clear();
    const bool absthr = GMO_IS_ABSOLUTE(P.sw_full_threshold);
    const double mate_thres = P.sw_full_threshold * 0.5;                     // gmapper.c:2677
    for (int i = 0; i < n1; i++) {
      for (int j = 0; j < 2; j++) {
        Hit* rh = p1[i].rh[j]; Read& re = (j == 0 ? re1 : re2);
        if (rh->score_full < 0 || !rh->has_sfr) {
          hit_run_full_sw(T, re, *rh, (int)GMO_ABS_OR_PCT(mate_thres, rh->score_max));
          if (P.compute_mapping_qualities && rh->score_full > 0) hit_run_post_sw(re, *rh);
        }
      }
      if (p1[i].rh[0]->score_full == 0 || p1[i].rh[1]->score_full == 0) continue;
      if (p1[i].rh[0]->score_full + p1[i].rh[1]->score_full >= (int)GMO_ABS_OR_PCT(P.sw_full_threshold, p1[i].score_max)) {
        HitPair hp; readpair_compute_paired_hit(p1[i].rh[0], p1[i].rh[1], absthr, &hp); p2.push_back(hp);
      }
    }
    // readpair_remove_duplicate_hits (mapping.c:2113-2175)
    push_dominant(p2, absthr, 0, [](const HitPair& a, const HitPair& b) { return cmp_gen_start(a.rh[0], b.rh[0]); });
    push_dominant(p2, absthr, 0, [](const HitPair& a, const HitPair& b) { return cmp_gen_end(a.rh[0], b.rh[0]); });
    push_dominant(p2, absthr, 1, [](const HitPair& a, const HitPair& b) { return cmp_gen_start(a.rh[1], b.rh[1]); });
    push_dominant(p2, absthr, 1, [](const HitPair& a, const HitPair& b) { return cmp_gen_end(a.rh[1], b.rh[1]); });
    std::stable_sort(p2.begin(), p2.end(), [](const HitPair& a, const HitPair& b) { return pair_pointer_cmp(a, b) < 0; });
    { size_t m = 0, i = 0, n = p2.size();                                  // removedups (common/util.c:1240-1255)
      while (i < n) { size_t j = i + 1; while (j < n && !pair_pointer_cmp(p2[i], p2[j])) j++; if (m < i) p2[m] = p2[i]; m++; i = j; }
      p2.resize(m); }
    std::stable_sort(p2.begin(), p2.end(), [](const HitPair& a, const HitPair& b) { return (b.key - a.key) < 0; });
    if ((int)p2.size() > P.num_outputs) p2.resize(P.num_outputs);
    if (P.strata && !p2.empty()) { size_t i = 1; while (i < p2.size() && p2[0].score == p2[i].score) i++; p2.resize(i); }
    if (!p2.empty() && !(P.max_alignments == 0 || (int)p2.size() <= P.max_alignments)) p2.clear();
    for (auto& hp : p2) { hp.rh[0]->saved = 1; hp.rh[1]->saved = 1; }
  }

  // readpair_save_final_hits (mapping.c:2446-2499): pool order = first appearance; paired_hit_idx in increasing pair index
  void readpair_save_final_hits(PairEntry& pe, std::vector<HitPair>& p2) const {
    size_t base = pe.final_paired_hits.size();
    pe.final_paired_hits.insert(pe.final_paired_hits.end(), p2.begin(), p2.end());
    for (size_t i = 0; i < p2.size(); i++)
      for (int nip = 0; nip < 2; nip++) {
        if (pe.final_paired_hits[base + i].rh[nip] != nullptr) {
          pe.pool[nip].push_back(*pe.final_paired_hits[base + i].rh[nip]);
          int pidx = (int)pe.pool[nip].size() - 1;
          for (size_t j = i; j < p2.size(); j++) {
            HitPair& hp = pe.final_paired_hits[base + j];
            if (hp.rh[nip] == p2[i].rh[nip]) { hp.rh[nip] = nullptr; hp.rh_idx[nip] = pidx; pe.pool[nip][pidx].paired_hit_idx.push_back((int)(base + j)); }
          }
        }
      }
  }

  // handle_read as the half-paired fall-back (unpaired_mapping_options[nip][0], gmapper.c:2700-2714):
  // regions / anchors / windows are reused, pass 1 re-runs over all windows, results are saved, not printed
  void handle_read_half(ThreadState& T, Read& re) const {
    read_pass1(T, re, 0, false, 2); read_pass1(T, re, 1, false, 2);                  // unpaired_mapping_options[..][0].pass1.min_matches = 2 (gmapper.c:2708)
    std::vector<Hit*> p1, p2; int n1 = 0;
    read_get_vector_hits(re, p1, n1);
    read_pass2(T, re, p1, n1, p2);
    if (!p2.empty()) { for (auto* h : p2) re.final_unpaired_hits.push_back(*h); re.mapped = true; }   // read_save_final_hits (mapping.c:1753-1770)
  }

  static double normal_cdf(double x, double mean, double stddev) {            // common/util.h:311-326
    double y = (x - mean) / stddev; if (y < 0) y = -y;
    double b0 = 0.2316419, b1 = 0.319381530, b2 = -0.356563782, b3 = 1.781477937, b4 = -1.821255978, b5 = 1.330274429, pi = 3.141592653589;
    double t = 1.0 / (1.0 + b0 * y);
    double res = (exp(-y * y / 2) / sqrt(2.0 * pi)) * ((((b5 * t + b4) * t + b3) * t + b2) * t + b1) * t;
    if (x > mean) res = 1 - res;
    return res;
  }
  double get_pr_insert_size(double ins) const {                                // output.c:795-808
    double res = normal_cdf(ins + 10, P.insert_size_mean, P.insert_size_stddev) - normal_cdf(ins - 10, P.insert_size_mean, P.insert_size_stddev);
    if (res < 1e-200) res = 1e-200;
    return res;
  }
  static double get_pr_missed(const Read& re) { return re.read_len < 40 ? 1e-10 : (re.read_len < 60 ? 1e-14 : 1e-16); }   // mapping.h:28-37
  double pr_random_mapping_given_score(const Read& re, int score) const {      // mapping.h:39-61
    int read_len = re.read_len;
    if (score > read_len * P.match_score) return 1e-200;
    unsigned a = (unsigned)(read_len * P.match_score - score), b = (unsigned)abs(P.colour ? P.crossover_score : P.mismatch_score - P.match_score);   // colour space: crossovers
    int n_mismatches = (a == 0) ? 0 : (int)((a - 1) / b + 1);                   // ceil_div (util.h:213-219)
    double lnck = 0.0; for (int i = 0; i < n_mismatches; i++) lnck += log(read_len - i) - log(i + 1);   // log_nchoosek (util.c:1306-1313)
    double tmp = -lnck - n_mismatches * log(3) + read_len * log(4);
    return exp(-tmp);
  }

  // compute_paired_mqv (output.c:811-942)
  void compute_paired_mqv(PairEntry& pe) const {
    double z1[2], z3, pr_top_random[3] = {1.0, 1.0, 1.0}, pr_missed_mp[2], class_select_denom;
    long long total_genome_size = 0; for (auto l : G->len) total_genome_size += l;
    for (int nip = 0; nip < 2; nip++) {
      z1[nip] = 0;
      for (auto& h : pe.re[nip]->final_unpaired_hits) z1[nip] += h.sfr.posterior;
      for (auto& h : pe.re[nip]->final_unpaired_hits) { h.sfr.z0 = h.sfr.posterior; h.sfr.z1 = z1[nip]; }
    }
    double insert_size_denom = 0.0;
    for (auto& hp : pe.final_paired_hits) insert_size_denom += get_pr_insert_size(hp.insert_size);
    for (int nip = 0; nip < 2; nip++) for (auto& h : pe.pool[nip]) h.sfr.insert_size_denom = insert_size_denom;
    z3 = 0.0;
    for (int nip = 0; nip < 2; nip++)
      for (auto& rh : pe.pool[nip]) {
        double tmp = 0.0;
        for (int idx : rh.paired_hit_idx) {
          HitPair& hp = pe.final_paired_hits[idx];
          Hit& mp = pe.pool[1 - nip][hp.rh_idx[1 - nip]];
          tmp += get_pr_insert_size(hp.insert_size) * mp.sfr.posterior;
        }
        tmp *= rh.sfr.posterior;
        if (tmp < 1e-200) tmp = 1e-200;
        rh.sfr.z2 = tmp;
        if (nip == 0) z3 += tmp;
      }
    for (int nip = 0; nip < 2; nip++) for (auto& h : pe.pool[nip]) h.sfr.z3 = z3;
    for (int nip = 0; nip < 2; nip++) {
      auto& U = pe.re[nip]->final_unpaired_hits;
      if (U.empty()) continue;
      size_t mx = 0;
      for (size_t i = 1; i < U.size(); i++) if (U[i].sfr.z0 > U[mx].sfr.z0) mx = i;
      pr_top_random[nip] = pr_random_mapping_given_score(*pe.re[nip], U[mx].sfr.posterior_score);
      for (auto& h : U) h.sfr.pr_top_random_at_location = pr_top_random[nip];
      pr_top_random[nip] *= (double)total_genome_size;
      if (pr_top_random[nip] > 1) pr_top_random[nip] = 1.0;
    }
    for (auto& hp : pe.final_paired_hits) {
      double tmp = pr_random_mapping_given_score(*pe.re[0], pe.pool[0][hp.rh_idx[0]].sfr.posterior_score);
      tmp *= pr_random_mapping_given_score(*pe.re[1], pe.pool[1][hp.rh_idx[1]].sfr.posterior_score);
      tmp *= 1000;
      if (tmp < pr_top_random[2]) pr_top_random[2] = tmp;
    }
    for (auto& hp : pe.final_paired_hits) {
      pe.pool[0][hp.rh_idx[0]].sfr.pr_top_random_at_location = pr_top_random[2];
      pe.pool[1][hp.rh_idx[1]].sfr.pr_top_random_at_location = pr_top_random[2];
    }
    pr_top_random[2] *= (double)total_genome_size;
    if (pr_top_random[2] > 1) pr_top_random[2] = 1.0;
    for (int nip = 0; nip < 2; nip++) {
      pr_missed_mp[nip] = get_pr_missed(*pe.re[1 - nip]);
      for (auto& h : pe.re[nip]->final_unpaired_hits) h.sfr.pr_missed_mp = pr_missed_mp[nip];
    }
    class_select_denom = 0.0;
    if (!pe.re[0]->final_unpaired_hits.empty()) class_select_denom += pr_top_random[1] * pr_top_random[2] * pr_missed_mp[0];
    if (!pe.re[1]->final_unpaired_hits.empty()) class_select_denom += pr_top_random[0] * pr_top_random[2] * pr_missed_mp[1];
    if (!pe.final_paired_hits.empty()) class_select_denom += pr_top_random[0] * pr_top_random[1];
    for (int nip = 0; nip < 2; nip++)
      for (auto& rh : pe.re[nip]->final_unpaired_hits) {
        double p_corr = (pr_top_random[1 - nip] * pr_top_random[2] * pr_missed_mp[nip] / class_select_denom) * (rh.sfr.z0 / rh.sfr.z1);
        rh.sfr.mqv = qv_from_pr_corr(p_corr); if (rh.sfr.mqv < 4) rh.sfr.mqv = 0;
      }
    for (auto& hp : pe.final_paired_hits)
      for (int nip = 0; nip < 2; nip++) {
        Hit& rh = pe.pool[nip][hp.rh_idx[nip]];
        double p_corr = (pr_top_random[0] * pr_top_random[1] / class_select_denom) * (rh.sfr.z2 / rh.sfr.z3);
        rh.sfr.mqv = qv_from_pr_corr(p_corr); if (rh.sfr.mqv < 4) rh.sfr.mqv = 0;
      }
  }

  // hit_output in paired mode (output.c:227-774): rh may be null (unmapped mate line), rh_mp may be null
  void hit_output_paired(const Read& re, const Hit* rh, const Hit* rh_mp, bool first_in_pair, bool improper, std::string& out) const {
    char buf[256];
    const Read& re_mp = *re.mate_pair;
    std::string qname = re.name;                                             // common prefix of the two names (output.c:365-378)
    { size_t n = std::min(re.name.size(), re_mp.name.size()), i = 0;
      while (i < n && re.name[i] == re_mp.name[i]) i++;
      if (i > 0 && (re.name[i - 1] == ':' || re.name[i - 1] == '/')) i--;
      qname = re.name.substr(0, i); }
    std::string seq(re.read_len, 'N');
    if (!P.colour) {
      for (int i = 0; i < re.read_len; i++) {
        char c = re.seq[i];
        switch (c) { case 'R': case 'Y': case 'S': case 'W': case 'K': case 'M': case 'B': case 'D': case 'H': case 'V': seq[i] = 'N'; break;
                     default: if (c >= 'a') c -= 32; seq[i] = c; break; }
      }
    } else seq = "*";                                                          // output.c:353-355
    const bool paired_alignment = (rh != nullptr && rh_mp != nullptr && !improper);
    const bool query_unmapped = (rh == nullptr), mate_unmapped = (rh_mp == nullptr);
    bool reverse_strand = false, reverse_strand_mp = false;
    int genome_start_mp = 0, genome_end_mp = 0, mpos = 0; const char* mrnm = "*";
    if (!mate_unmapped) {
      int rs = rh_mp->sfr.read_start + 1, rend = rs + rh_mp->sfr.rmapped - 1, glen = (int)G->len[rh_mp->cn];
      reverse_strand_mp = (rh_mp->gen_st == 1);
      if (!reverse_strand_mp) genome_start_mp = rh_mp->sfr.genome_start + 1;
      else genome_start_mp = (glen - rh_mp->sfr.genome_start) - (rend - rs - rh_mp->sfr.deletions + rh_mp->sfr.insertions);
      genome_end_mp = genome_start_mp + rh_mp->sfr.gmapped - 1;
      mpos = genome_start_mp; mrnm = G->names[rh_mp->cn].c_str();
    }
    const bool second_in_pair = !first_in_pair;
    auto flags = [&]() {
      return 0x1 | (paired_alignment ? 0x2 : 0) | (query_unmapped ? 0x4 : 0) | (mate_unmapped ? 0x8 : 0) | (reverse_strand ? 0x10 : 0) |
             (reverse_strand_mp ? 0x20 : 0) | (first_in_pair ? 0x40 : 0) | (second_in_pair ? 0x80 : 0);
    };
    if (query_unmapped) {                                                     // output.c:411-466 (half_paired => only this case)
      out += qname; snprintf(buf, sizeof buf, "\t%i\t*\t0\t0\t*\t%s\t%u\t0\t", flags(), mrnm, (unsigned)mpos); out += buf;
      out += seq; out += "\t"; out += (P.Qflag && !P.colour) ? re.qual : std::string("*");    // output.c:419-421
      if (P.colour) { out += "\tCQ:Z:"; out += P.Qflag ? re.qual : std::string("*"); out += "\tCS:Z:"; out += re.seq; }   // output.c:441-451
      out += "\n";
      return;
    }
    const SwFullResults& s = rh->sfr;
    reverse_strand = (rh->gen_st == 1);
    int read_start = s.read_start + 1, read_end = read_start + s.rmapped - 1, genome_length = (int)G->len[rh->cn];
    std::vector<std::pair<int, char>> cigar;
    make_cigar(read_start, read_end, re.read_len, s.qralign, s.dbalign, &cigar);
    int j = P.colour ? 0 : read_start - 1;                                       // output.c:485-493: colour space prints the aligned part only
    if (P.colour) seq.assign(read_end - read_start + 1, 'N');
    for (size_t i = 0; i < s.qralign.size(); i++) {
      char c = s.qralign[i];
      if (c != '-') { if (c >= 'a') c -= 32; if (c != 'A' && c != 'G' && c != 'C' && c != 'T' && c != 'N') c = 'N'; seq[j++] = c; }
    }
    std::string cs_qual = "*";
    if (!P.colour) seq.resize(j + (re.read_len - read_end));
    else {                                                                      // output.c:572-621: hard clips, post_sw's base qualities with QVs
      for (auto& c : cigar) if (c.second == 'S') c.second = 'H';
      if (P.Qflag && P.compute_mapping_qualities) { cs_qual = s.qual; if (rh->gen_st == 1) for (int i = 0; i < s.rmapped / 2; i++) std::swap(cs_qual[i], cs_qual[s.rmapped - i - 1]); }
    }
    int genome_start;
    if (!reverse_strand) genome_start = s.genome_start + 1;
    else {
      genome_start = (genome_length - s.genome_start) - (read_end - read_start - s.deletions + s.insertions);
      std::string t(seq.size(), ' ');
      for (size_t i = 0; i < seq.size(); i++) t[seq.size() - 1 - i] = rc_char(seq[i]);
      seq = t; std::reverse(cigar.begin(), cigar.end());
    }
    int genome_end = genome_start + s.gmapped - 1;
    int isize = 0;
    if (!mate_unmapped) {
      if (G->names[rh->cn] == mrnm) {
        mrnm = "=";
        int fivep = reverse_strand ? genome_end : genome_start - 1;
        int fivep_mp = reverse_strand_mp ? genome_end_mp : genome_start_mp - 1;
        isize = fivep_mp - fivep;
      } else isize = 0;
    }
    out += qname;
    snprintf(buf, sizeof buf, "\t%i\t", flags()); out += buf;
    out += G->names[rh->cn];
    snprintf(buf, sizeof buf, "\t%u\t%i\t", (unsigned)genome_start, s.mqv); out += buf;
    for (auto& c : cigar) { snprintf(buf, sizeof buf, "%d%c", c.first, c.second); out += buf; }
    snprintf(buf, sizeof buf, "\t%s\t%u\t%i\t", mrnm, (unsigned)mpos, isize); out += buf;
    out += seq; out += "\t";
    if (P.colour) out += cs_qual;
    else if (P.Qflag) {                          // output.c:539-570
      std::string qual = re.qual;
      if (reverse_strand) std::reverse(qual.begin(), qual.end());
      if (P.qual_delta != 33) for (auto& c : qual) c = (char)(c - P.qual_delta + 33);
      out += qual;
    } else out += "*";
    snprintf(buf, sizeof buf, "\tAS:i:%d", rh->score_full); out += buf;
    if (P.compute_mapping_qualities && !P.all_contigs) {
      if (rh != nullptr && rh_mp != nullptr && !improper)
        snprintf(buf, sizeof buf, "\tZ2:i:%d\tZ3:i:%d\tZ4:i:%d\tZ6:i:%d", double_to_neglog(s.z2), double_to_neglog(s.z3),
                 double_to_neglog(s.pr_top_random_at_location), double_to_neglog(s.insert_size_denom));
      else
        snprintf(buf, sizeof buf, "\tZ0:i:%d\tZ1:i:%d\tZ4:i:%d\tZ5:i:%d", double_to_neglog(s.z0), double_to_neglog(s.z1),
                 double_to_neglog(s.pr_top_random_at_location), double_to_neglog(s.pr_missed_mp));
      out += buf;
    }
    snprintf(buf, sizeof buf, "\tNM:i:%d", s.mismatches + s.deletions + s.insertions); out += buf;
    if (P.colour) {                             // output.c:717-730
      if (P.Qflag) { out += "\tCQ:Z:"; out += re.qual; }
      out += "\tCS:Z:"; out += re.seq;
      snprintf(buf, sizeof buf, "\tCM:i:%d\tXX:Z:", s.crossovers); out += buf; out += s.qralign;
    }
    out += "\n";
  }

  // readpair_output (output.c:1070-1291), default flags (no single-best-mapping)
  // get_idx_mp_max_mqv (output.c:1049-1067): the pair that holds pool hit idx of mate nip and, of those, the other mate's best mapping quality (first one)
  int get_idx_mp_max_mqv(const PairEntry& pe, int nip, int idx) const {
    int best_other = -1, idx_pair = -1;
    for (int pi : pe.pool[nip][idx].paired_hit_idx) {
      const Hit& o = pe.pool[1 - nip][pe.final_paired_hits[pi].rh_idx[1 - nip]];
      if (o.sfr.mqv > best_other) { best_other = o.sfr.mqv; idx_pair = pi; }
    }
    return idx_pair;
  }
  void readpair_output(PairEntry& pe, std::string& out) const {
    int first[3] = {0, 0, 0}, last[3] = {(int)pe.re[0]->final_unpaired_hits.size(), (int)pe.re[1]->final_unpaired_hits.size(), (int)pe.final_paired_hits.size()};
    if (P.compute_mapping_qualities) {
      compute_paired_mqv(pe);
      if (P.single_best_mapping && (last[2] > 0 || last[0] > 0 || last[1] > 0)) {                          // output.c:1094-1235
        int max_idx_unpaired[2] = {-1, -1}, max_idx_paired[2] = {-1, -1}, max_mqv_unpaired[2] = {-1, -1}, max_mqv_paired[2] = {-1, -1};
        for (int nip = 0; nip < 2; nip++) {
          const auto& U = pe.re[nip]->final_unpaired_hits;
          for (int i = 0; i < (int)U.size(); i++) if (U[i].sfr.mqv > max_mqv_unpaired[nip]) { max_mqv_unpaired[nip] = U[i].sfr.mqv; max_idx_unpaired[nip] = i; }
          for (int i = 0; i < (int)pe.pool[nip].size(); i++) if (pe.pool[nip][i].sfr.mqv > max_mqv_paired[nip]) { max_mqv_paired[nip] = pe.pool[nip][i].sfr.mqv; max_idx_paired[nip] = i; }
        }
        if (!P.all_contigs) {                                       // the top mapping of each class
          for (int nip = 0; nip < 2; nip++) if (max_idx_unpaired[nip] >= 0) { first[nip] = max_idx_unpaired[nip]; last[nip] = first[nip] + 1; }
          const int best_nip = max_mqv_paired[0] > max_mqv_paired[1] ? 0 : 1;
          if (max_mqv_paired[best_nip] >= 0) { first[2] = get_idx_mp_max_mqv(pe, best_nip, max_idx_paired[best_nip]); last[2] = first[2] + 1; }
        } else {                                                    // the top mapping over all classes
          int max_mqv[2], max_is_paired[2], max_idx[2];
          for (int nip = 0; nip < 2; nip++) {
            if (max_mqv_unpaired[nip] > max_mqv_paired[nip]) { max_mqv[nip] = max_mqv_unpaired[nip]; max_is_paired[nip] = 0; max_idx[nip] = max_idx_unpaired[nip]; }
            else { max_mqv[nip] = max_mqv_paired[nip]; max_is_paired[nip] = 1; max_idx[nip] = max_idx_paired[nip]; }
          }
          const int best_nip = max_mqv[0] >= max_mqv[1] ? 0 : 1;
          if (max_is_paired[best_nip] == 1) {
            last[0] = 0; last[1] = 0; first[2] = get_idx_mp_max_mqv(pe, best_nip, max_idx[best_nip]); last[2] = first[2] + 1;
          } else {                                                  // an unpaired mapping wins: can it be paired with the other mate's best one, across contigs?
            auto& OU = pe.re[1 - best_nip]->final_unpaired_hits;
            int idx_best_other = -1; double max_other_z0 = 0.0;
            for (int i = 0; i < (int)OU.size(); i++) if (OU[i].sfr.z0 > max_other_z0) { max_other_z0 = OU[i].sfr.z0; idx_best_other = i; }
            int best_other_mqv = -1;
            if (idx_best_other >= 0) best_other_mqv = qv_from_pr_corr(max_other_z0 / OU[idx_best_other].sfr.z1);
            if (!P.improper_mappings || max_mqv_unpaired[best_nip] < 10 || best_other_mqv < 10) {
              last[2] = 0; last[1 - best_nip] = 0; first[best_nip] = max_idx[best_nip]; last[best_nip] = first[best_nip] + 1;
            } else {
              HitPair hp; hp.rh[best_nip] = &pe.re[best_nip]->final_unpaired_hits[max_idx[best_nip]]; hp.rh[1 - best_nip] = &OU[idx_best_other];
              hp.score_max = hp.rh[0]->score_max + hp.rh[1]->score_max; hp.insert_size = get_insert_size(hp.rh[best_nip], hp.rh[1 - best_nip]); hp.improper_mapping = true;
              pe.final_paired_hits.push_back(hp);
              last[0] = 0; last[1] = 0; first[2] = (int)pe.final_paired_hits.size() - 1; last[2] = first[2] + 1;
            }
          }
        }
      }
    }
    for (int i = first[2]; i < last[2]; i++) {
      HitPair& hp = pe.final_paired_hits[i];
      Hit* rh1 = hp.rh[0] ? hp.rh[0] : &pe.pool[0][hp.rh_idx[0]]; Hit* rh2 = hp.rh[1] ? hp.rh[1] : &pe.pool[1][hp.rh_idx[1]];
      hit_output_paired(*pe.re[0], rh1, rh2, true, hp.improper_mapping, out);
      hit_output_paired(*pe.re[1], rh2, rh1, false, hp.improper_mapping, out);
    }
    for (int nip = 0; nip < 2; nip++)
      for (int ui = first[nip]; ui < last[nip]; ui++) {
        Hit& rh = pe.re[nip]->final_unpaired_hits[ui];
        Read& rep = *pe.re[nip];
        if (rep.first_in_pair) { hit_output_paired(rep, &rh, nullptr, true, false, out); hit_output_paired(*rep.mate_pair, nullptr, &rh, false, false, out); }
        else { hit_output_paired(*rep.mate_pair, nullptr, &rh, true, false, out); hit_output_paired(rep, &rh, nullptr, false, false, out); }
      }
  }

  // read_reverse (gmapper.c:174-185)
  static void read_reverse(Read& re) { std::swap(re.bits[0], re.bits[1]); re.input_strand = 1 - re.input_strand; }

  // handle_readpair (mapping.c:2502-2636) + the pair set-up of the read loop (gmapper.c:561-577)
  void handle_readpair(ThreadState& T, Read& re1, Read& re2, std::string& out) const {
    static const bool pair_reverse[5][2] = {{0, 0}, {0, 0}, {1, 1}, {0, 1}, {1, 0}};   // gmapper-defaults.h:184-191
    if (pair_reverse[P.pair_mode][0]) read_reverse(re1);
    if (pair_reverse[P.pair_mode][1]) read_reverse(re2);
    re1.paired = true; re1.first_in_pair = true; re1.mate_pair = &re2;
    re2.paired = true; re2.first_in_pair = false; re2.mate_pair = &re1;
    PairEntry pe; pe.re[0] = &re1; pe.re[1] = &re2;
    read_get_mapidxs(re1); read_get_mapidxs(re2);
    readpair_compute_mp_ranges(re1, re2);
    if (P.mp_match_mode != 2) {                                              // regions.recompute = use_regions && match_mode != 2 (gmapper.c:2652)
    T.region_map_id++; T.region_map_id &= ((1 << region_map_id_bits) - 1);
    read_get_region_counts(T, re1, 0); read_get_region_counts(T, re1, 1);
    read_get_region_counts(T, re2, 0); read_get_region_counts(T, re2, 1);
    }
    if (mp_region_mode()) { read_get_mp_region_counts(T, re1, 0); read_get_mp_region_counts(T, re1, 1); read_get_mp_region_counts(T, re2, 0); read_get_mp_region_counts(T, re2, 1); }   // mapping.c:2531-2538
    read_get_anchor_list(T, re1, 0); read_get_anchor_list(T, re1, 1);
    read_get_anchor_list(T, re2, 0); read_get_anchor_list(T, re2, 1);
    for (Read* re : {&re1, &re2}) {
      read_get_hit_list(*re, 0, &T); read_get_hit_list(*re, 1, &T);
      for (size_t i = 0; i < re->hits[0].size(); i++) re->hits[0][i].sort_idx = (int)i;
      for (size_t i = 0; i < re->hits[1].size(); i++) re->hits[1][i].sort_idx = (int)(re->hits[0].size() + i);
    }
    for (Read* re : {&re1, &re2}) for (int st = 0; st < 2; st++) { T.stats.pair_anchors += re->anchors[st].size(); T.stats.pair_windows += re->hits[st].size(); }
    readpair_pair_up_hits(re1, re2);
    read_pass1(T, re1, 0, true, pair_min_matches()); read_pass1(T, re1, 1, true, pair_min_matches());
    read_pass1(T, re2, 0, true, pair_min_matches()); read_pass1(T, re2, 1, true, pair_min_matches());
    std::vector<HitPair> p1, p2; int n1 = 0;
    readpair_get_vector_hits(re1, re2, p1, n1);
    readpair_pass2(T, re1, re2, p1, n1, p2);
    if (!p2.empty()) { readpair_save_final_hits(pe, p2); pe.mapped = true; }
    // hits of pass 1 that were not saved lose their alignment (free_sfrp, mapping.c:2590-2595) and are re-aligned if selected again
    for (int i = 0; i < n1; i++) for (int j = 0; j < 2; j++) if (p1[i].rh[j]->has_sfr && p1[i].rh[j]->saved != 1) { p1[i].rh[j]->has_sfr = false; }
    if (P.half_paired) { handle_read_half(T, re1); handle_read_half(T, re2); }   // stop_threshold 101 % is never met (gmapper.c:2686-2687)
    readpair_output(pe, out);
    if (P.sam_unaligned && !(pe.mapped || re1.mapped || re2.mapped)) {
      hit_output_paired(re1, nullptr, nullptr, true, false, out); hit_output_paired(re2, nullptr, nullptr, false, false, out);
    }
  }

  // handle_read (mapping.c:1773-1868) with the single default unpaired option set (gmapper.c:2601-2634)
  void handle_read(ThreadState& T, Read& re, std::string& out, std::vector<Hit*>* top_out = nullptr) const {
    read_get_mapidxs(re);
    bool regions = (P.match_mode == 2);
    if (regions) {
      T.region_map_id++; T.region_map_id &= ((1 << region_map_id_bits) - 1);
      read_get_region_counts(T, re, 0); read_get_region_counts(T, re, 1);
    }
    read_get_anchor_list(T, re, 0); read_get_anchor_list(T, re, 1);
    read_get_hit_list(re, 0); read_get_hit_list(re, 1);
    for (size_t i = 0; i < re.hits[0].size(); i++) re.hits[0][i].sort_idx = (int)i;
    for (size_t i = 0; i < re.hits[1].size(); i++) re.hits[1][i].sort_idx = (int)(re.hits[0].size() + i);
    read_pass1(T, re, 0); read_pass1(T, re, 1);
    std::vector<Hit*> p1, p2; int n1 = 0;
    read_get_vector_hits(re, p1, n1);
    if (top_out) { top_out->assign(p1.begin(), p1.begin() + n1); }
    read_pass2(T, re, p1, n1, p2);
    if (!p2.empty()) read_output(re, p2, out);
    else if (P.sam_unaligned) hit_output(re, nullptr, out);
  }
};

}  // namespace gmo
