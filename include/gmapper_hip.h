/*
 * gmapper_hip.h -- C ABI of libgmapper_hip.so: an MI355X (gfx950) implementation of the
 * SHRiMP2 gmapper hot path (spaced-seed lookup + Smith-Waterman extension).
 *
 * Plain C types only (pointers + sizes); no torch / HIP types cross this boundary.
 * All "ref:" citations are file:line in compbio-UofT/shrimp (SHRiMP 2.2.3).
 *
 * Seams replaced:
 *   S1  sw_vector_setup / sw_vector / sw_vector_stats      ref: common/sw-vector.h:1-6, common/sw-vector.c:388-515
 *       sw_gapless_setup / sw_gapless / sw_gapless_stats   ref: common/sw-gapless.h:11-14, common/sw-gapless.c:29-117
 *   S2  sw_full_ls_setup / sw_full_ls                       ref: common/sw-full-ls.h, common/sw-full-ls.c:568-683
 *   S4  handle_read (per-read pipeline -> SAM text)        ref: gmapper/mapping.h:23, gmapper/mapping.c:1773-1868
 *   S5  load_genome (index build) + its globals            ref: gmapper/genome.h:23-32, gmapper/genome.c:1012-1182
 * The reference calls S1/S2/S4 once per window / per read from an OpenMP thread; a GPU needs
 * batches, so every seam has a batch form; the single-call forms keep the reference's exact
 * parameter lists (for drop-in linking) and are thin wrappers over a batch of one.
 *
 * Errors: integer return codes (0 = ok, <0 = GM_E_*); nothing throws across the ABI.
 * Threading: one host thread per gm_session; a session owns its HIP streams on one device.  TWO mapping calls (of two sessions) may be in flight on the SAME device --
 * the lookup kernels' per-device scratch exists twice -- a third waits for one of them to return; sessions on different devices run side by side.  The unpaired file entry
 * uses this itself: a file of more than one chunk is mapped by the session and a twin of it (same index and parameters, made on first use, freed with the session) on two
 * threads, so that one chunk's tail runs under the next chunk's lookups (GM_FILE_ONE_SESSION=1 in the environment keeps it to the one session).
 */
#ifndef GMAPPER_HIP_H
#define GMAPPER_HIP_H

#include <stddef.h>
#include <stdint.h>
#include <stdbool.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GM_OK            0
#define GM_E_NODEVICE   -1   /* no HIP device / HIP runtime error (message via gm_last_error) */
#define GM_E_ARG        -2   /* invalid argument */
#define GM_E_NOTSETUP   -3   /* call before *_setup (the reference abort()s here, sw-vector.c:463) */
#define GM_E_RANGE      -4   /* match*qrlen >= 32768 (ref: sw-vector.c:393-398, exit(1) there) */
#define GM_E_OVERFLOW   -5   /* a per-read candidate list exceeded every configured capacity */
#define GM_E_NOMEM      -6

const char *gm_last_error(void);
/* sizeof of the structs that cross the ABI, for bindings to check their mirrors against: which = 0 gm_params_t, 1 gm_pair_opts_t, 2 gm_map_stats_t, 3 gm_merge_options_t
 * (-1 for any other value) */
int gm_abi_sizeof(int which);
/* number of visible HIP devices (0 when none); never initialises more than the runtime */
int gm_device_count(void);

/* ---- scoring / pipeline parameters = the reference's globals (ref: gmapper/gmapper.h:47-127) ---- */
typedef struct gm_params {
  int match_score, mismatch_score;             /* ref: gmapper-defaults.h:45-46   (10, -15) */
  int a_gap_open_score, a_gap_extend_score;    /* ref: gmapper-defaults.h:47,49   (-33, -7)  gap along the genome */
  int b_gap_open_score, b_gap_extend_score;    /* ref: gmapper-defaults.h:48,50   (-33, -3)  gap along the read   */
  double window_len;                           /* ref: gmapper-defaults.h:31  140.0 (%; negative = absolute) */
  double window_overlap;                       /* ref: gmapper-defaults.h:32   90.0 */
  double window_gen_threshold;                 /* ref: gmapper-defaults.h:62   55.0 */
  double sw_vect_threshold, sw_full_threshold; /* ref: gmapper-defaults.h:67-68; LS: vect := full (gmapper.c:2456-2458) */
  int match_mode;                              /* ref: gmapper-defaults.h:34    2 */
  int num_outputs, num_tmp_outputs;            /* ref: gmapper.h:53-55         10, 30 */
  int anchor_width;                            /* ref: gmapper-defaults.h:36    8 */
  int region_bits, region_overlap;             /* ref: gmapper-defaults.h:22-23 11, 50 */
  uint32_t list_cutoff;                        /* 0 = automatic (ref: gmapper.c:2811-2837) */
  int hash_filter_calls;                       /* ref: gmapper-defaults.h:17 true; 0 == -Z */
  int tiebreak_rev;                            /* Tflag, ref: gmapper.h:87 true */
  int sam_unaligned;                           /* ref: gmapper.h:185 false */
  int longest_read_len;                        /* ref: gmapper-defaults.h:72 1000 */
  int strata;                                  /* --strata: only the best-scoring hits (ref: gmapper.h:85, mapping.c:1706-1712,2268-2274) false */
  int max_alignments;                          /* --max-alignments: drop reads with more final hits (ref: gmapper.h:54, mapping.c:1713-1722) 0 = all */
  int colour_space;                            /* shrimp_mode == MODE_COLOUR_SPACE, i.e. the gmapper-cs binary (ref: util.c:28-38) 0 */
  int crossover_score;                         /* ref: gmapper-defaults.h:54  -20; the vector filter's mismatch is match + crossover (gmapper.c:2935) */
  int indel_taboo_len;                         /* ref: gmapper.h:57  0 */
  double pr_xover;                             /* ref: gmapper.h:119  0.03: fixes score_alpha in colour space (gmapper.c:2557-2563) */
  int local_alignment;                         /* --local, i.e. Gflag off (ref: gmapper.c:2303-2305): sw_full_ls in local mode (soft clips); mapping qualities
                                                  are then unavailable (gmapper.c:2325-2328): MAPQ 255, no Z0-Z6 tags.  Letter space and colour space, unpaired and paired (sw_full_cs with
                                                  local_alignment, ref: sw-full-cs.c:199-203,315,439-552; no post_sw then, mapping.c:1648; with csfastq quality values too, unpaired and paired).  0 */
  int ungapped;                                /* -U (gapless_sw): pass 1 scores windows with sw_gapless (ref: sw-gapless.c:57-117), every anchor opens a window
                                                  (mapping.c:1095,1154).  As the reference's -U does, also set anchor_width 0, both gap opens -255 and
                                                  hash_filter_calls 0; requires local_alignment (gmapper.c:2330-2333).  In colour space sw_gapless runs on the contig's colour translation with the
                                                  read's first colour forced against the primer (sw-gapless.c:84-94).  0 */
  int hash_seeds;                              /* -H (Hflag): lists are keyed by kmer_to_mapidx_hash -- 4^12 lists per seed whatever its weight, so seeds
                                                  heavier than 14 are allowed (ref: gmapper.h:309-336, seeds.c:83-102,132-136).  An index property.  0 */
  int output_format;                           /* 0: SAM (-E, the binary's default); 1: --shrimp-format, one line per mapping -- readname contigname strand
                                                  contigstart contigend readstart readend readlength score editstring (ref: common/output.c:280-352 output_normal,
                                                  :36-115 alignment_edit_string); 2: -P/--pretty, each line followed by the aligned G: / R: (/ T:) rows
                                                  (ref: common/output.c:118-262 output_pretty).  Unaligned reads print nothing in 1 / 2; in paired mode every mate has its
                                                  own line under its own name, and the unmapped mate of a half-paired mapping prints ">name" (gmapper/output.c:292-294).  0 */
  int print_read_seq;                          /* -R: the read's sequence as a last column of the SHRiMP-format line (ref: gmapper.h Rflag, output.c:310-313)  0 */
  int strand_only;                             /* 1: -F / --positive (only the read as given), 2: -C / --negative (only its reverse complement): the other strand gets no
                                                  anchor list (ref: mapping.c:879-880, gmapper.c:1979-1992).  Unpaired only -- paired mode maps both strands, as the
                                                  reference does after its warning (gmapper.c:2448-2451).  0 */
  /* output policy of read_output / readpair_output (ref: output.c:955-1008,1070-1291; gmapper.c:2252-2268) */
  int single_best_mapping;                     /* --single-best-mapping: only the mapping with the highest quality -- unpaired: the first such; paired: the best unpaired
                                                  mapping of each mate and the pair that holds the best paired one (with all_contigs: ONE record pair over all classes;
                                                  an unpaired winner is joined with the other mate's best unpaired mapping into an IMPROPER pair when both qualities
                                                  reach 10, unless no_improper_mappings).  Without mapping qualities (--local, no_mapping_qualities) it does nothing, as in the reference.  0 */
  int all_contigs;                             /* --all-contigs: no Z0-Z6 tags (ref: output.c:691); changes what single_best_mapping selects.  0 */
  int no_mapping_qualities;                    /* --no-mapping-qualities: MAPQ 255, no Z tags, no post_sw (colour space prints sw_full_cs's own strings), as --local implies.  0 */
  int no_improper_mappings;                    /* --no-improper-mappings  0 */
  /* optional tail of every SAM record (ref: output.c:452-465,729-756) */
  int extra_sam_fields;                        /* --extra-sam-fields: ZM:i matches, ZR:i window-generation score, ZV:i vector score, ZH:i full SW score, ZE:Z edit string
                                                  (alignment_edit_string, reversed for mappings on the reverse strand) on mapped records.  0.
                                                  UNTESTED corner: a read or window letter outside A/C/G/T on a reverse-strand mapping -- the reference's
                                                  reverse_alignment_edit_string never returns on such a letter (its loop index stops advancing, output.c:164-220), so no golden can
                                                  exist; this library passes the letter through unchanged */
  int sam_r2;                                  /* --sam-r2 (paired mode only): R2:Z (colour space: X2:Z) = the mate's sequence as given.  0 */
  char read_group[64];                         /* --read-group: RG:Z:<name> on every record ("" = none; the @RG header line is the caller's, as the @SQ lines are) */
  /* the read loop's preprocessing, applied by the file entry points to every read before it is mapped (ref: gmapper.c:262-284 trim_read, :427-472, :495-521) */
  int trim_front, trim_end;                    /* --trim-front / --trim-end: that many characters off either end of the sequence (and its quality string).  0, 0.
                                                  trim_front is refused in colour space, as the binary refuses it (gmapper.c:2134-2137) */
  int trim_first, trim_second;                 /* paired mode: which mates are trimmed (--trim-first: 1, 0; --trim-second: 0, 1).  1, 1 -- but trimming the FIRST mate is
                                                  refused: the reference trims it after packing it (gmapper.c:427-438,474-475) and prints records that contradict themselves */
  int trim_illumina;                           /* --trim-illumina (letter space, FASTQ): a tail of 'B' quality values is cut off with its bases.  0 */
  int min_avg_qv;                              /* --min-avg-qv: a read with quality values whose integer mean (sum / length) is below this gets no record at all (ref: gmapper.h:81,
                                                  gmapper.c:456-462,491-498).  10; < 0: none */
  int ignore_qvs;                              /* --ignore-qvs: the quality values are printed but not used -- no min_avg_qv, no range check; colour space: the global crossover
                                                  score and error rates (ref: gmapper.c:532,2962).  0 */
  int no_qv_check;                             /* --no-qv-check: a quality value outside [-10, 50] is NOT an error (ref: gmapper.c:463-472: the binary exits there).  0 */
} gm_params_t;

void gm_params_default(gm_params_t *p);        /* letter-space defaults of the reference binary (gmapper-ls) */
void gm_params_default_cs(gm_params_t *p);     /* colour-space defaults (gmapper-cs): mismatch -24, vector threshold 47%, ref: gmapper.c:1748-1755 */

/* ---------------------------------------------------------------------------------------------
 * S5: index.  Replaces load_genome() and the globals genomemap / genomemap_len / genome_contigs /
 * contig_offsets / genome_len (ref: gmapper/genome.c:1012-1182, gmapper/gmapper.h:262-275).
 * contigs[c] is the reference's own 4-bit bitfield (8 bases per uint32, base i in nibble i%8 of
 * word i/8; ref: common/util.h:41, common/fasta.c:609-673).  seeds are "0/1" strings
 * (ref: gmapper/seeds.c:9-43); n_seeds == 0 selects the binary's default 3 seeds of weight 12
 * (ref: gmapper-defaults.h:212-227).  The index is built on the device (histogram-free radix
 * sort of (k-mer, position) pairs) and stays resident in HBM.
 * ------------------------------------------------------------------------------------------- */
typedef struct gm_index gm_index_t;
int  gm_index_build(gm_index_t **out, int device, int n_contigs, const uint32_t *const *contigs,
                    const uint32_t *contig_len, const char *const *contig_names,
                    int n_seeds, const char *const *seeds, const gm_params_t *params);
void gm_index_free(gm_index_t *ix);
/* The reference's on-disk index ("-S prefix" / "-L prefix": <prefix>.genome + <prefix>.seed.<n>, gzip; ref: gmapper/genome.c:15-270,
 * 670-831).  gm_index_save writes files stock gmapper can load; gm_index_load reads files stock gmapper wrote (letter or colour space, with or without -H: the mode and Hflag words of the files must match params)
 * and uploads them -- the lists are taken as they are, only the per-slab directory is derived on the device. */
int  gm_index_save(const gm_index_t *ix, const char *prefix);
int  gm_index_load(gm_index_t **out, int device, const char *prefix, const gm_params_t *params);
/* introspection (mirrors the reference's globals) */
uint32_t gm_index_list_cutoff(const gm_index_t *ix);
uint64_t gm_index_bytes(const gm_index_t *ix);
int      gm_index_n_slabs(const gm_index_t *ix);
int      gm_index_has_buckets(const gm_index_t *ix);   /* 1 when the 64-byte bucket layout is resident (small genomes) */
/* genomemap_len[sn][mapidx] / genomemap[sn][mapidx][0..len) copied back to the host (tests) */
int gm_index_get_list(const gm_index_t *ix, int sn, uint32_t mapidx, uint32_t *len, uint32_t *positions, uint32_t cap);
/* raw device pointers + sizes of the resident arrays, for the single RCCL broadcast at start-up
 * (SURVEY.md section 8(e)); kind: 0 genome, 1+3*sn directory of seed sn, 2+3*sn positions of seed sn,
 * 3+3*sn the 64-byte buckets of seed sn (bytes == 0 when that layout is not resident), 1+3*n_seeds the colour translation
 * of the genome (bytes == 0 in letter space); larger kinds return GM_E_ARG */
int gm_index_device_array(const gm_index_t *ix, int kind, void **dev_ptr, uint64_t *bytes);
/* allocate an index with the same shape (from the metadata blob of a built index) so that a
 * non-root rank can receive the arrays; meta is host memory */
int gm_index_meta(const gm_index_t *ix, void *meta, uint64_t *meta_bytes);
int gm_index_alloc_like(gm_index_t **out, int device, const void *meta, uint64_t meta_bytes);

/* ---------------------------------------------------------------------------------------------
 * S1: vector Smith-Waterman filter (score only).  ref: common/sw-vector.c:388-515
 * Same parameter lists as the reference (penalties passed negative).  State is per calling
 * thread, as in the reference (threadprivate).  With use_colours the read's first colour is compared with lstocs(genome_ls[j], initbp)
 * (see the colour-space block below), with is_rna a U in the genome letters counting as T there (ref: sw-vector.c:129,289; util.h:182-205).
 * ------------------------------------------------------------------------------------------- */
int sw_vector_setup(int dblen, int qrlen, int a_gap_open, int a_gap_ext, int b_gap_open, int b_gap_ext,
                    int match, int mismatch, int use_colours, bool reset_stats);
int sw_vector(uint32_t *genome, int goff, int glen, uint32_t *read, int rlen,
              uint32_t *genome_ls, int initbp, bool is_rna);
void sw_vector_stats(uint64_t *invocs, uint64_t *cells, double *secs);
int sw_vector_cleanup(void);
/* batch form: n independent windows; genome/read are host bitfields, words are uploaded once.
 * g_off[i] indexes into `genome` (one shared bitfield), reads[i*read_words .. ) holds read i. */
int gm_sw_vector_batch(int n, const uint32_t *genome, uint64_t genome_words, const int64_t *g_off, const int *glen,
                       const uint32_t *reads, int read_words, const int *rlen, int *scores);
/* the same with pass 1's early stop (unpaired reads, ref: mapping.c:1332-1335 only compares the score with the threshold): a window whose remaining
 * cells can no longer lift any alignment to `threshold` stops there; stopped[i] = 1 and scores[i] = the best score up to that point, which like the
 * final one is below `threshold`.  stopped[i] = 0: scores[i] is the value gm_sw_vector_batch returns.  Letter space, reads up to 128 bases stop early. */
int gm_sw_vector_batch_bounded(int n, const uint32_t *genome, uint64_t genome_words, const int64_t *g_off, const int *glen,
                               const uint32_t *reads, int read_words, const int *rlen, int threshold, int *scores, uint8_t *stopped);

/* S1, ungapped form (-U / gapless_sw): the best ungapped segment on the diagonal of `genome` (glen positions from its first word) through
 * (g_idx, r_idx).  ref: common/sw-gapless.h:11-14, sw-gapless.c:29-117; f1_setup / f1_run call these instead of sw_vector when gapless_sw is set
 * (f1-wrapper.h:66-68,122-125).  Colour space: genome = colours, genome_ls = the letters, and a diagonal that starts at the read's first colour
 * compares it with lstocs(genome_ls[g], init_bp) (:84-94).  sw_gapless_stats: invocations, cells (+= rlen per call, :111), time inside (here ns). */
int  sw_gapless_setup(int match, int mismatch, bool reset_stats);
int  sw_gapless(uint32_t *genome, int glen, uint32_t *read, int rlen, int g_idx, int r_idx, uint32_t *genome_ls, int init_bp, bool is_rna);
void sw_gapless_stats(uint64_t *invocs, uint64_t *cells, uint64_t *ticks);
/* batch form: call i's bitfield starts at word genome_woff[i] of `genome` (and of `genome_ls`, which is NULL in letter space; initbp may then be NULL) */
int gm_sw_gapless_batch(int n, const uint32_t *genome, const uint32_t *genome_ls, uint64_t genome_words, const int64_t *genome_woff, const int *glen,
                        const uint32_t *reads, int read_words, const int *rlen, const int *g_idx, const int *r_idx, const int *initbp, int *scores);

/* ---------------------------------------------------------------------------------------------
 * S2: full Smith-Waterman with traceback, letter space.  ref: common/sw-full-ls.c:568-683
 * struct layouts follow the reference (ref: common/sw-full-common.h:13-48, gmapper-definitions.h:66-74)
 * for the fields this path fills; dbalign/qralign are malloc()ed and owned by the caller,
 * as with the reference's xstrdup (ref: sw-full-ls.c:676-677).  Both alignment modes: local_alignment = 0 (global in the read, gmapper's
 * default) and 1 (--local: floored states, and when the best score in the anchor band is not maxscore a second run over the band
 * the threshold allows, :395-398); anchors = one box as gmapper passes it, or NULL (that threshold band, :179-192).
 * ------------------------------------------------------------------------------------------- */
struct gm_anchor { long long x, y; int length, width, weight, cn, score; };
struct gm_sw_full_results {                    /* field for field struct sw_full_results (ref: common/sw-full-common.h:13-48) */
  int read_start, rmapped, genome_start, gmapped, matches, mismatches, insertions, deletions, score;
  int posterior_score, pct_posterior_score;   /* filled by hit_run_post_sw in the reference, untouched here */
  char *dbalign, *qralign, *qual;
  double posterior;
  int mqv; double z0, z1, z2, z3, pr_top_random_at_location, pr_missed_mp, insert_size_denom;
  int crossovers;                             /* colour space only */
  bool dup, in_use;
};
int sw_full_ls_setup(int dblen, int qrlen, int a_gap_open, int a_gap_ext, int b_gap_open, int b_gap_ext,
                     int match, int mismatch, bool reset_stats, int anchor_width);
void sw_full_ls(uint32_t *genome, int goff, int glen, uint32_t *read, int rlen, int threshscore, int maxscore,
                struct gm_sw_full_results *sfr, bool revcmpl, struct gm_anchor *anchors, int anchors_cnt,
                int local_alignment);
int sw_full_ls_cleanup(void);
void sw_full_ls_stats(uint64_t *invocs, uint64_t *cells, double *secs);   /* ref: common/sw-full-ls.h:11, called from gmapper.c:745 */

/* ---------------------------------------------------------------------------------------------
 * S1/S2 in colour space.  sw_vector_setup(..., use_colours = 1, ...) makes sw_vector() compare the read's first colour with
 * lstocs(genome_ls[j], initbp) (ref: common/sw-vector.c:112-146); `mismatch` is then match + crossover (ref: gmapper.c:2935).
 * sw_full_cs: four letter-space translations of the colour read, 3-state affine DP in four layers with crossovers between
 * layers on the NW and N transitions, traceback with crossover marks (lower-case in qralign), ref: common/sw-full-cs.c:249-1236.
 * Both modes (local_alignment 0 / 1, ref: sw-full-cs.c:199-203,315,439-552) and one anchor box, as gmapper calls it (ref: mapping.c:375-379).  crossover_score: NULL (the
 * global penalty everywhere) or rlen per-position penalties, what gmapper passes for every read with quality values (ref: mapping.c:375-379, gmapper.c:532-544, used per row
 * at sw-full-cs.c:312-322); the device keeps them in 8 bits (gmapper clamps them to [2 * global, -1]).  An argument combination that is not implemented (several anchors, a
 * score outside [-128, 127]) is refused LOUDLY -- the reason on stderr and in gm_last_error(), sfr->score = 0 -- never answered as if the window held no alignment.
 * is_rna (all three seams; gmapper passes genome_is_rna, the flag of the LAST contig it read: genome.c:1063-1064): lstocs reads a U as T, cstols takes a U as T and hands back U
 * where it would hand back T (util.h:157-205) -- in sw_vector's and sw_gapless's first-colour comparison and in the four letter translations of sw_full_cs.  In letter space the
 * argument changes nothing (sw-vector.c uses it in the colour-space row only).
 * RNA in the batch entries (gm_map_reads* / gm_map_pairs*): a contig with uracil and no thymine is an RNA contig (fasta.c:528-542) -- its reverse complement holds U for A and its
 * colour translation reads U as T (genome.c:1107-1118); the last contig's flag is what the SW stages get, as in the reference; a letter-space read with U and no T is
 * reverse-complemented with U for A (gmapper.c:487).  The read's flag is taken from the letters the device gets, i.e. AFTER the file entries' trimming (the reference takes it
 * from the file's sequence before trimming: the two differ only for a read that holds both U and T and loses every T to the trim).
 * ------------------------------------------------------------------------------------------- */
int sw_full_cs_setup(int dblen, int qrlen, int a_gap_open, int a_gap_ext, int b_gap_open, int b_gap_ext,
                     int match, int mismatch, int global_xover_penalty, bool reset_stats, int anchor_width, int indel_taboo_len);
void sw_full_cs(uint32_t *genome_ls, int goff, int glen, uint32_t *read, int rlen, int initbp, int threshscore,
                struct gm_sw_full_results *sfr, bool revcmpl, bool is_rna, struct gm_anchor *anchors, int anchors_cnt,
                int local_alignment, int *crossover_score);
int sw_full_cs_cleanup(void);
void sw_full_cs_stats(uint64_t *invocs, uint64_t *cells, double *secs);   /* ref: common/sw-full-cs.h:9, called from gmapper.c:742 */

/* ---------------------------------------------------------------------------------------------
 * S3: post_sw, the colour-space posterior of one alignment (16-state forward-backward over the aligned columns, doubles through libm in
 * the reference's operation order -- a host routine here as in the reference).  ref: common/sw-post.h:6-9, sw-post.c:364-758; called by
 * hit_run_post_sw (gmapper/mapping.c:1609-1625).  post_sw re-calls sfr->qralign in place, recounts matches / mismatches / crossovers,
 * mallocs sfr->qual (base qualities, PHRED+33) and fills sfr->posterior.  State is per calling thread.  Returns follow the reference (1).
 * ------------------------------------------------------------------------------------------- */
int  post_sw_setup(int max_len, double pr_snp, double pr_xover, double pr_del_open, double pr_del_extend, double pr_ins_open, double pr_ins_extend,
                   bool use_read_qvs, bool use_sanger_qvs, int qual_vector_offset, int qual_delta, bool reset_stats);
void post_sw(uint32_t *read, int initbp, char *qual, struct gm_sw_full_results *sfr);
int  post_sw_cleanup(void);
int  post_sw_stats(uint64_t *invocs, uint64_t *cells, double *secs);
/* batch form of the colour-space vector filter: as gm_sw_vector_batch plus the letter-space genome and one initial base per read */
int gm_sw_vector_batch_cs(int n, const uint32_t *genome_cs, const uint32_t *genome_ls, uint64_t genome_words, const int64_t *g_off,
                          const int *glen, const uint32_t *reads, int read_words, const int *rlen, const int *initbp, int *scores);

/* ---------------------------------------------------------------------------------------------
 * S4: the per-read pipeline.  Replaces handle_read() for unpaired letter-space reads
 * (ref: gmapper/mapping.c:1773-1868) and the read loop body around it (ref: gmapper/gmapper.c:436-560).
 * Input: n reads of one length, as 4-bit bitfields (read_words = (read_len+7)/8 words each), plus
 * names ('\n' separated) and original sequence text for SAM SEQ.  Output: the SAM records the
 * reference would append to its thread output buffer, in input order (ref: gmapper/output.c:227-774).
 * ------------------------------------------------------------------------------------------- */
typedef struct gm_session gm_session_t;
int  gm_session_create(gm_session_t **out, const gm_index_t *ix, const gm_params_t *params, int max_batch_reads);
void gm_session_free(gm_session_t *s);

typedef struct gm_map_stats {
  uint64_t reads, reads_matched, sam_records;
  uint64_t lookups, list_entries, list_bytes;      /* seed-lookup algorithmic work (SURVEY.md 8(d) B_seed) */
  uint64_t survivors, anchors, windows;            /* after region filter / collapse / window generation */
  uint64_t vec_calls, vec_cells, vec_bypassed;     /* pass-1 vector SW (ref: sw-vector.c:509 swcells) */
  uint64_t full_calls, full_cells;                 /* pass-2 (vector re-score + banded full SW) */
  uint64_t exact_order_reads;                      /* read-strands that needed heap-order emulation */
  uint64_t retries;                                /* capacity-overflow re-runs */
  uint64_t survivors_pruned;                       /* survivors with no neighbour within window_len + read_len, removed before K2 (exact) */
  uint64_t mp_unfiltered;                          /* mate-pair region counts: always 0 in a result -- a sub-batch with (pair, strand) items beyond the filter's LDS tiers is
                                                      redone through the exact row-based path (counted in retries); rows beyond their capacity fail the call (GM_E_OVERFLOW) */
  uint64_t post_sw_host_redo;                      /* colour space: device post_sw results the host routine redid because a value to be rounded (AS, MAPQ, Z0 / Z1) lay at a boundary */
  double   ms_lookup, ms_anchors, ms_pass1, ms_select, ms_pass2, ms_host;   /* device time per stage (events) */
} gm_map_stats_t;

/* A22: the read loop's text -> bitfield step inside the library.  gm_sequence_to_bitfield == fasta_sequence_to_bitfield with the reference's code
 * tables (ref: common/fasta.c:609-673, :151-200; fasta.h:26-42): letters A C G T U M R W S Y K V H D B N -> 0..15 (X and '.' -> 15, either case);
 * in colour space the first character is the primer letter (returned in *initbp), then colours 0-3 ('4', 'N', '.', 'X' -> 15).  Host code, no device.
 * gm_map_reads_text takes the reads as lines of one length and keeps the characters for the fields the reference prints from the file's text
 * (SEQ of unaligned reads and clipped ends, ref: gmapper/output.c:326-351; CS:Z, :451,727). */
int gm_sequence_to_bitfield(int colour_space, const char *seq, int seq_len, uint32_t *words, int *initbp);
int gm_map_reads_text(gm_session_t *s, int n_reads, int read_len, const char *seqs, const char *names, const char *quals, int qual_delta,
                      char **sam, size_t *sam_len, gm_map_stats_t *stats);
/* File input: the reads file as the reference's reader takes it -- FASTA or FASTQ (fastq: 0 / 1, -1 detects from the first character),
 * plain or gzip, '#' comment lines, sequences over several lines, names cut at the first blank, any mix of read lengths, primer + colours in a
 * colour-space session (ref: common/fasta.c:61-150 fasta_open, :315-552 fasta_get_next_read_with_range).  Reads longer than longest_read_len are
 * dropped without a record and reading stops at a malformed entry, as in the reference (gmapper.c:495-521, fasta.c:362-372).  Records come back in
 * the file's order. */
int gm_map_reads_file(gm_session_t *s, const char *path, int fastq, int qual_delta, char **sam, size_t *sam_len, gm_map_stats_t *stats);
/* The same, streaming (ref: gmapper.c:322-398,588-607: the reference reads chunks of reads and prints as it goes): the file is read at most chunk_reads reads at a time
 * (0 = 2^20; the next chunk is read and preprocessed by a second thread while this one is mapped), every read goes through the read loop's preprocessing (gm_params_t:
 * trim_*, min_avg_qv, ignore_qvs, no_qv_check) and each chunk's records are handed to `write` in the file's order.  Neither the file nor the output is held whole: host
 * memory stays at a few hundred bytes per read of one chunk.  `write` returns 0 to go on; anything else stops the call with an error.
 * gm_map_reads_file is this function with a write function that collects the text. */
typedef int (*gm_write_fn)(void *ctx, const char *text, size_t len);
int gm_map_reads_file_cb(gm_session_t *s, const char *path, int fastq, int qual_delta, size_t chunk_reads, gm_write_fn write, void *ctx, gm_map_stats_t *stats);
/* The preprocessing by itself (host code, no device): seq (primer letter first in colour space) and qual (NULL for FASTA input) are NUL-terminated and edited in place;
 * *drop = 1 when the read gets no record.  mate: 0 unpaired, 1 / 2 the mates of a pair. */
int gm_preprocess_read_text(const gm_params_t *params, int mate, char *seq, char *qual, int qual_delta, int *drop);
/* host-buffer form: reads are uploaded, SAM text is returned in a malloc()ed buffer (*sam, *sam_len) */
int gm_map_reads(gm_session_t *s, int n_reads, int read_len, const uint32_t *reads_packed,
                 const char *names, char **sam, size_t *sam_len, gm_map_stats_t *stats);
/* device-resident form for measurement: reads_dev is a device pointer to the same packed layout;
 * runs the whole device pipeline and the host finalisation (pass-2 selection, MAPQ, SAM text)
 * unless emit_sam == 0, in which case only the alignment records are produced and counted. */
int gm_map_reads_device(gm_session_t *s, int n_reads, int read_len, const void *reads_dev,
                        int emit_sam, char **sam, size_t *sam_len, gm_map_stats_t *stats);
/* FASTQ input (the reference's -Q, letter space): as gm_map_reads, plus the reads' QUAL strings ('\n' separated, as in the file) and the
 * file's quality offset (--qv-offset; the binary's letter-space default is 64, gmapper-defaults.h:41).  QVs do not influence letter-space
 * alignment; they travel to the SAM QUAL column: reversed with the read and re-based to PHRED+33 for mapped reads, verbatim for unmapped
 * ones (ref: output.c:419-421,539-570). */
int gm_map_reads_fastq(gm_session_t *s, int n_reads, int read_len, const uint32_t *reads_packed, const char *names,
                       const char *quals, int qual_delta, char **sam, size_t *sam_len, gm_map_stats_t *stats);
/* colour-space reads (SOLiD): colours_packed holds n_colours 4-bit colour codes per read (0-3, 15 for a skipped cycle) in the
 * same bitfield layout, initbp[i] the primer letter code (A0 C1 G2 T3) -- what fasta_sequence_to_bitfield / fasta_get_initial_base
 * produce (ref: gmapper.c:475-487).  The session's index must have been built with colour_space = 1.  Replaces handle_read() for
 * the gmapper-cs binary: colour k-mers from colour 1 on, the colour vector filter on the read's input strand, sw_full_cs, the
 * post_sw forward-backward pass and the colour-space SAM fields (ref: mapping.c:1297-1319,375-379,1613-1614; sw-post.c:639-758;
 * output.c:441-451,485-493,572-580,717-730). */
int gm_map_reads_cs(gm_session_t *s, int n_reads, int n_colours, const uint32_t *colours_packed, const uint8_t *initbp,
                    const char *names, char **sam, size_t *sam_len, gm_map_stats_t *stats);
/* csfastq reads: as gm_map_reads_cs, plus one QV character per colour ('\n' separated strings, offset qual_delta; 33 is the binary's colour-space
 * default, gmapper-defaults.h:42).  The QVs give per-position crossover scores in sw_full_cs (ref: gmapper.c:532-544, sw-full-cs.c:312), per-colour
 * error rates in post_sw (sw-post.c:486-491), the SAM QUAL column (post_sw's base qualities) and the CQ:Z tag (output.c:613-621,724-727). */
int gm_map_reads_cs_fastq(gm_session_t *s, int n_reads, int n_colours, const uint32_t *colours_packed, const uint8_t *initbp,
                          const char *names, const char *quals, int qual_delta, char **sam, size_t *sam_len, gm_map_stats_t *stats);
void gm_free(void *p);          /* every *sam the library hands out; a large text buffer is parked for the next call instead of being unmapped */
void gm_release_cache(void);    /* drops the parked buffer */

/* ---------------------------------------------------------------------------------------------
 * Paired mode.  Replaces handle_readpair() (ref: gmapper/mapping.c:2502-2636) with the binary's
 * default paired option sets (ref: gmapper/gmapper.c:2636-2720: two matches per mate, half-paired
 * fall-back, mapping qualities) and readpair_output() (ref: gmapper/output.c:1070-1291).
 * pair_mode / insert sizes are the reference's -p and -I options (ref: gmapper.c:1583-1623,
 * gmapper-defaults.h:28-31,184-191); mean/stddev feed the insert-size term of the paired MAPQ
 * (ref: output.c:795-808).  mates1[i] and mates2[i] are the two reads of pair i; each side has
 * ONE length per call.  Output: for every pair the paired records, then the half-paired records
 * of mate 1, then of mate 2 -- the text the reference appends to its output buffer.
 * ------------------------------------------------------------------------------------------- */
enum { GM_PAIR_OPP_IN = 1, GM_PAIR_OPP_OUT = 2, GM_PAIR_COL_FW = 3, GM_PAIR_COL_BW = 4 };   /* ref: gmapper-definitions.h:42-46 */
typedef struct gm_pair_opts {
  int pair_mode;                               /* ref: gmapper.h:140 */
  int min_insert_size, max_insert_size;        /* ref: gmapper-defaults.h:28-29  0 / 1000 */
  double insert_size_mean, insert_size_stddev; /* ref: gmapper-defaults.h:30-31  200 / 100 */
  int half_paired;                             /* ref: gmapper.h:181 true; 0 = --no-half-paired: each mate's list entries are filtered by the other mate's region
                                                  counts (mapping.c:545-608,733-742, use_mp_region_counts = 1) and no unpaired rescue runs (gmapper.c:2657-2683) */
  int match_mode;                              /* the reference's -n in paired mode (ref: gmapper-defaults.h:35 DEF_MATCH_MODE_PAIRED 4; gmapper.c:2652-2673):
                                                  4: two k-mer matches per mate (region counts; with half_paired = 0 also the mate's);
                                                  3: one match is enough where the mate has hits within reach (use_mp_region_counts 2, or 3 with half_paired = 0;
                                                     hit list mode 3, mapping.c:733-742,1080-1093,1153-1157);
                                                  2: no region counts at all, a window per anchor.   0 is read as 4.
                                                  PERFORMANCE LIMIT of modes 2 and 3: a read-strand's list entries all reach the window kernel, whose LDS tier holds 4 096 of them;
                                                  on genomes beyond a few hundred Mbp (17 k - 72 k entries per read-strand at 3 Gbp) nearly every read-strand then takes the
                                                  heavy tier (a segmented sort and a stream synchronisation per sub-batch): the output stays exact, the throughput drops by
                                                  more than an order of magnitude.  Mode 4 (the reference's default) has no such limit */
} gm_pair_opts_t;
void gm_pair_opts_default(gm_pair_opts_t *o);
int gm_map_pairs(gm_session_t *s, int n_pairs, int len1, const uint32_t *mates1_packed, int len2, const uint32_t *mates2_packed,
                 const char *names1, const char *names2, const gm_pair_opts_t *opts,
                 char **sam, size_t *sam_len, gm_map_stats_t *stats);
/* Colour-space pairs (gmapper-cs -p <mode>; ref: handle_readpair mapping.c:2502-2636 with the colour-space branches of read_pass1_per_strand :1297-1319,
 * hit_run_full_sw :375-379 and hit_output output.c:353-355,441-451,485-537,572-580,717-730): mates as packed colours and one primer-letter byte per read, as
 * gm_map_reads_cs takes them, in all four pair modes (a mate the mode reverses keeps its colours and swaps its strand labels). */
int gm_map_pairs_cs(gm_session_t *s, int n_pairs, int len1, const uint32_t *mates1_packed, const uint8_t *initbp1, int len2, const uint32_t *mates2_packed,
                    const uint8_t *initbp2, const char *names1, const char *names2, const gm_pair_opts_t *opts,
                    char **sam, size_t *sam_len, gm_map_stats_t *stats);
/* Paired reads from files: `path2` NULL = one file with the mates adjacent, else mate 1 from path1 and mate 2 from path2 (gmapper's -1 / -2; ref: gmapper.c:363-620).
 * Formats and rules as gm_map_reads_file; pairs of any mix of lengths; a pair with a mate beyond longest_read_len is dropped whole.  Letter or colour space by the session. */
int gm_map_pairs_file(gm_session_t *s, const char *path1, const char *path2, int fastq, int qual_delta, const gm_pair_opts_t *opts,
                      char **sam, size_t *sam_len, gm_map_stats_t *stats);
/* streaming form (see gm_map_reads_file_cb): at most chunk_pairs pairs at a time (0 = 2^19); a pair is never split across chunks (ref: gmapper.c:2319-2322) */
int gm_map_pairs_file_cb(gm_session_t *s, const char *path1, const char *path2, int fastq, int qual_delta, const gm_pair_opts_t *opts, size_t chunk_pairs,
                         gm_write_fn write, void *ctx, gm_map_stats_t *stats);
/* csfastq pairs: as gm_map_pairs_cs, plus one QV character per colour and mate ('\n' separated strings, offset qual_delta), used as gm_map_reads_cs_fastq uses them
 * (per-position crossover scores in sw_full_cs, per-colour error rates in post_sw, QUAL = post_sw's base qualities, CQ:Z). */
int gm_map_pairs_cs_fastq(gm_session_t *s, int n_pairs, int len1, const uint32_t *mates1_packed, const uint8_t *initbp1, int len2, const uint32_t *mates2_packed,
                          const uint8_t *initbp2, const char *names1, const char *names2, const char *quals1, const char *quals2, int qual_delta,
                          const gm_pair_opts_t *opts, char **sam, size_t *sam_len, gm_map_stats_t *stats);
/* FASTQ pairs: as gm_map_pairs, plus the mates' QUAL strings ('\n' separated) and the file's quality offset; the QUAL column is printed as
 * gm_map_reads_fastq prints it (mates keep their input orientation: read_reverse() leaves seq and qual alone, ref: gmapper.c:174-185). */
int gm_map_pairs_fastq(gm_session_t *s, int n_pairs, int len1, const uint32_t *mates1_packed, int len2, const uint32_t *mates2_packed,
                       const char *names1, const char *names2, const char *quals1, const char *quals2, int qual_delta,
                       const gm_pair_opts_t *opts, char **sam, size_t *sam_len, gm_map_stats_t *stats);

/* ---------------------------------------------------------------------------------------------
 * Merging.  Replaces the mergesam program (ref: mergesam/mergesam.c main :308-816, sam_reader.c pp_ll_combine_and_check :417-716,
 * render.c :210-277; SPLITTING_AND_MERGING:100-148): SAM texts of several gmapper runs -- the same reads against different contig groups,
 * different reads against the same genome, or both -- become one SAM text in the order of the reads text, with the mapping qualities
 * recomputed from the Z0-Z6 fields gmapper writes unless --all-contigs was given.  Host code (text in, text out).
 *
 * reads_text is the reads file (FASTA or FASTQ; for pairs the file gmapper read, mates adjacent): only the names are used, and a record
 * belongs to the first read, from the last matched one on, whose name starts with its QNAME (ref: sam_reader.c:947).  Every SAM text
 * must list its records in that order (gmapper's output does).  The fields mirror mergesam's options of the same names; the
 * reference's --un / --al files are `output` 1 / 2 (FASTA / FASTQ text of the unaligned / aligned reads instead of SAM).
 * command_line, when given, is what follows "CL:" in the @PG line mergesam adds for itself (NULL: no such line). */
#define GM_MERGE_OUT_SAM              0
#define GM_MERGE_OUT_UNALIGNED_READS  1
#define GM_MERGE_OUT_ALIGNED_READS    2
typedef struct gm_merge_options {
  int max_outputs;            /* -o/--report (10) */
  int max_alignments;         /* --max-alignments (0 = all) */
  int strata;                 /* --strata */
  int half_paired;            /* --half-paired (1) / --no-half-paired (0) */
  int sam_unaligned;          /* --sam-unaligned */
  int single_best;            /* --single-best-mapping (forces max_outputs = 1) */
  int all_contigs;            /* --all-contigs: last merge of these reads, the Z fields are dropped */
  int no_mapping_qualities;   /* --no-mapping-qualities (MAPQ 255 unless leave_mapq) */
  int leave_mapq;             /* --leave-mapq-untouched */
  int no_improper_mappings;   /* --no-improper-mappings */
  int min_mapq;               /* --min-mapq (with all_contigs) */
  int fastq;                  /* reads text: 0 FASTA, 1 FASTQ, -1 detect from the first character (mergesam's default) */
  int threads;                /* -N */
  int output;                 /* GM_MERGE_OUT_* */
  int header_given;           /* --sam-header: the caller supplies the header; only the first sorted line and the @PG line are written */
  const char *command_line;
} gm_merge_options_t;
void gm_merge_options_default(gm_merge_options_t *o);
int gm_merge_sam(const gm_merge_options_t *opts, const char *reads_text, size_t reads_len, int n_files, const char *const *sam_text,
                 const size_t *sam_len, char **out, size_t *out_len);     /* *out: gm_free */

/* per-read top-K candidate rows after pass 1 (heap array order), for stage parity tests:
 * 12 x int64 per row: read st cn g_off w_len score_vector pct_score_vector matches ax ay alen awidth */
int gm_debug_tophits(gm_session_t *s, int n_reads, int read_len, const uint32_t *reads_packed,
                     long long *rows, long cap, long *n_rows);

/* time of the dominant kernel (seed lookup) during the last gm_map_* call, from HIP events on the
 * session's own stream, and the algorithmic bytes it moved */
int gm_last_lookup_timing(gm_session_t *s, double *ms, uint64_t *alg_bytes, int *launches);
/* name of the seed-lookup kernel the last mapping call launched (k_lookup_bkt / k_lookup_v4 / k_lookup_v3 / k_lookup): what the roofline figures refer to */
const char *gm_last_lookup_kernel(void);

#ifdef __cplusplus
}
#endif
#endif
