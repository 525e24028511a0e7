// gm_internal.h -- host-side objects behind the opaque handles of include/gmapper_hip.h
#pragma once
#include "gm_common.h"

struct GmSeedHost {
  uint64_t mask = 0; int span = 0, weight = 0;
  int kbits = 0;                        // log2(number of lists): 2 * weight, or 24 with -H (ref: genome.c:1034)
  uint32_t* d_dir = nullptr; uint32_t* d_pos = nullptr; uint32_t* d_bkt = nullptr;
  uint32_t* d_sdir = nullptr; uint32_t* d_spos = nullptr; uint32_t n_spos = 0;   // strip lists (gm_index_derive_strips)
  uint32_t n_pos = 0; uint64_t dir_words = 0;
  std::string text;
};

struct GmIndexHost {
  int device = 0;
  uint64_t total_len = 0;
  int n_contigs = 0;
  std::vector<uint32_t> contig_off;     // n_contigs + 1
  std::vector<std::string> names;
  uint32_t* d_genome = nullptr; uint64_t genome_words = 0;
  uint32_t* d_genome_cs = nullptr;      // colour space: colour translation of d_genome (same coordinates)
  uint32_t* d_contig_off = nullptr;
  // RNA contigs (uracil and no thymine, ref: fasta.c:528-542), derived on the device from the resident genome (gm_index_derive_rna: after a build, a load or a
  // broadcast alike): each contig's own flag, and the LAST contig's as genome_is_rna (genome.c:1063-1064)
  bool rna_ready = false; std::vector<uint8_t> contig_rna; uint8_t* d_contig_rna = nullptr; int genome_is_rna = 0;
  int n_seeds = 0, min_seed_span = 64, max_seed_span = 0;
  GmSeedHost seeds[GM_MAX_SEEDS];
  int slab_bits = 29, n_slabs = 1;
  gm_params_t params;
  uint32_t list_cutoff = 0;
  bool strips_ready = false;
  GmIndexDev dev_view() const;
};
struct gm_index : GmIndexHost {};

// stats slots written by the kernels (uint64 each)
enum { GS_LOOKUPS = 0, GS_ENTRIES, GS_SURVIVORS, GS_ANCHORS, GS_WINDOWS, GS_VEC_CALLS, GS_VEC_CELLS, GS_VEC_BYPASSED,
       GS_FULL_CALLS, GS_FULL_CELLS, GS_EXACT_ORDER, GS_OVERFLOW_SURV, GS_OVERFLOW_HITS, GS_PRUNED, GS_MP_UNFILTERED, GS_N };
// Counters are striped: same-address device atomics retire at ~12 ns each (MI355X_MICROARCH 'fanin'),
// which at one atomic per wave would cost more than the kernels themselves.  Stripe = block & 1023,
// one 128-byte line per stripe; the host sums the stripes.
#define GS_STRIPES 1024
#define GS_STRIDE 16
#define GS_ADD(stats, slot, val) atomicAdd(&(stats)[(size_t)(blockIdx.x & (GS_STRIPES - 1)) * GS_STRIDE + (slot)], (unsigned long long)(val))

// Tuning / path-forcing knobs (environment variables GM_K1_*, GM_K4_*, GM_K5_*, GM_SCAP, ... -- the kernel-variant tests force code paths with them) and
// the ablation bits of the lookup kernels exist only in builds with -DGM_TUNING (the Makefile's default; `make TUNING=0` is the release build: the knobs
// read as unset and every ablation branch folds away).  GM_HOST_THREADS, GM_OVERLAP and GM_POST_SW_HOST are run-time options of every build.
#include <cstdlib>
#ifdef GM_TUNING
static inline const char* gm_tune(const char* name) { return getenv(name); }
#define GM_ABL(bits) (ablate & (bits))
#else
static inline const char* gm_tune(const char*) { return nullptr; }
#define GM_ABL(bits) (false)
#endif

// ---- kernel launchers (one per .hip file) -----------------------------------------------------
int gm_index_build_device(GmIndexHost* ix, hipStream_t stream);
int gm_index_colour_genome_device(GmIndexHost* ix, hipStream_t stream);   // derives d_genome_cs from d_genome
int gm_index_from_lists_device(GmIndexHost* ix, int sn, const uint32_t* lens, const uint32_t* pos, uint32_t total);

// The dynamic-LDS limit raised for a kernel so far, per device (hipFuncSetAttribute acts on the current device's code object: a process that maps on a
// second device must raise it there too).  One static instance per kernel at its launch site: `static GmLdsLimit lim; size_t& configured = lim.cur();`
struct GmLdsLimit { size_t v[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; size_t& cur() { int d = 0; (void)hipGetDevice(&d); return v[(d >= 0 && d < 16) ? d : 0]; } };


// K1 seed lookup + region filter: one workgroup per read-strand; read-strands with more than scap
// survivors are listed in d_heavy_list (count in d_surv_cnt) and re-run by gm_launch_lookup_redo
void gm_lookup_set_scratch_slot(int slot);   // which of the device's two lookup-scratch sets the calling thread's launches use (0 / 1)
void gm_lookup_set_start_flags(uint32_t* flags, int cap, uint32_t epoch);   // pinned words the persistent K1 grid raises when its workgroups are resident
int gm_lookup_start_flag_grid(void);
// Optional fusion of K1b into K1 (k_lookup_v5): when `fuse` is given and the chosen kernel can apply the prune rules itself, the kept
// keys go straight to fuse->d_surv2 / d_surv_cnt2 and *fuse->fused is set to 1 (the caller then skips gm_launch_prune).
struct GmFusePrune { uint64_t* d_surv2; uint32_t* d_surv_cnt2; int scap2; int window_len; int e_max; int* fused; };
int gm_launch_lookup(const GmIndexDev& ix, const uint32_t* d_reads, int n_reads, int read_len, int read_words,
                     uint64_t* d_surv, uint32_t* d_surv_cnt, int scap, uint32_t* d_heavy_list, uint32_t* d_heavy_cnt, int heavy_cap,
                     unsigned long long* d_stats, hipStream_t stream, uint32_t* d_surv_seg = nullptr,   // d_surv_seg[rs][S + 1]: survivors after each slab
                     const GmFusePrune* fuse = nullptr);
int gm_index_derive_rna(GmIndexHost* ix, hipStream_t stream);        // per-contig RNA flags (gm_index.hip); idempotent
int gm_index_derive_strips(GmIndexHost* ix, hipStream_t stream);     // strip lists for k_lookup_v5 (gm_lookup5.hip); idempotent
void gm_lookup5_set_start_flags(uint32_t* flags, int cap, uint32_t epoch);
int gm_lookup5_start_flag_grid(void);
int gm_lookup5_last_rounds(void);
int gm_lookup5_last_half(void);
int gm_lookup5_launch(const GmIndexDev& ix, const uint32_t* d_reads, int n_reads, int read_len, int read_words, int max_n_kmers, int NL,
                      uint64_t* d_out, uint32_t* d_out_cnt, int out_cap, uint32_t* d_surv_cnt, int prune, uint32_t D, int e_max,
                      uint32_t* d_heavy_list, uint32_t* d_heavy_cnt, int heavy_cap, unsigned long long* d_stats, hipStream_t stream,
                      uint32_t** fb_list, uint32_t** fb_cnt, int* fb_cap_out,
                      uint64_t* d_raw = nullptr, int raw_cap = 0, uint32_t* d_surv_seg = nullptr, uint32_t** pl_list = nullptr, uint32_t** pl_cnt = nullptr);
int gm_launch_lookup_redo(const GmIndexDev& ix, const uint32_t* d_reads, int n_reads, int read_len, int read_words,
                          int n_heavy, const uint32_t* d_redo_list, const uint64_t* d_redo_off, uint64_t* d_out,
                          unsigned long long* d_stats, hipStream_t stream);
size_t gm_lookup_lds_bytes(const GmIndexDev& ix, int read_len);

// K1b: exact removal of survivors with no other survivor within window_len + read_len (gm_prune.hip); the kept
// ones go to d_surv2 (stride scap2), d_surv_cnt2 = their number, or 0xFFFFFFFF for read-strands of the heavy tier
int gm_launch_prune(int n_reads, int read_len, int window_len, int e_max, int n_slabs, int slab_bits, const uint64_t* d_surv, const uint32_t* d_surv_cnt,
                    const uint32_t* d_surv_seg, int scap,
                    uint64_t* d_surv2, uint32_t* d_surv_cnt2, int scap2, uint32_t* d_heavy_list, uint32_t* d_heavy_cnt, int heavy_cap,
                    unsigned long long* d_stats, hipStream_t stream,
                    const uint32_t* d_rs_list = nullptr, const uint32_t* d_rs_cnt = nullptr, int rs_cap = 0);   // list mode: only the listed read-strands

// K2 anchors + candidate windows: one wave per read-strand (LDS tier) + heavy tier on global arrays
int gm_launch_anchors(const GmIndexDev& ix, const GmScoreDev& sc, int n_reads, int read_len, int window_len,
                      const uint64_t* d_surv, const uint32_t* d_surv_cnt, int scap,
                      GmHit* d_hits, uint16_t* d_perm, uint32_t* d_hit_cnt, int hcap, unsigned long long* d_stats, hipStream_t stream);
int gm_launch_anchors_heavy(const GmIndexDev& ix, const GmScoreDev& sc, int n_reads, int read_len, int window_len,
                            int n_heavy, const uint32_t* d_heavy_list, const uint64_t* d_seg_off, const uint32_t* d_seg_n,
                            const uint32_t* d_seg_begin32, const uint32_t* d_seg_end32, uint64_t total_keys,
                            uint64_t* d_keys_in, uint64_t* d_keys_sorted, uint32_t* d_aux, uint32_t* d_nxt, uint32_t* d_ord,
                            GmHit* d_hits, uint16_t* d_perm, uint32_t* d_hit_cnt, int hcap, unsigned long long* d_stats, hipStream_t stream);

// K3 pass 1 (vector SW + overlap rule), one wave per read-strand
int gm_launch_pass1(const GmIndexDev& ix, const GmScoreDev& sc, const uint32_t* d_reads, int n_reads, int read_len, int read_words,
                    int window_len, int window_overlap_abs, GmHit* d_hits, const uint16_t* d_perm, const uint32_t* d_hit_cnt, int hcap,
                    unsigned long long* d_slots, unsigned long long* d_stats, hipStream_t stream,
                    const int32_t* d_pair_min = nullptr, const uint8_t* d_saved = nullptr,   // paired mode: only_paired / saved windows
                    const uint8_t* d_initbp = nullptr,                                         // colour space: primer letter per read
                    bool early_stop = false);       // unpaired reads: a window may stop once it cannot reach the threshold (its score is then a lower bound below it)

// K4a top-K selection (ref: read_get_vector_hits), one thread per read; K4b pass 2, one wave per selected hit
#define GM_SEL_MAX 64
int gm_launch_select(const GmScoreDev& sc, int n_reads, int read_len, const GmHit* d_hits, const uint16_t* d_perm,
                     const uint32_t* d_hit_cnt, int hcap, int32_t* d_sel, uint32_t* d_sel_cnt, uint32_t* d_sel_off,
                     uint32_t* d_work, uint32_t* d_n_work, hipStream_t stream, const uint8_t* d_saved = nullptr);
int gm_launch_build_work(int n_reads, const uint32_t* d_sel_cnt, uint32_t* d_sel_off, uint32_t* d_work, uint32_t* d_n_work, hipStream_t stream);
int gm_launch_pass2(const GmIndexDev& ix, const GmScoreDev& sc, const uint32_t* d_reads, int n_reads, int read_len, int read_words,
                    int window_len, GmHit* d_hits, const uint16_t* d_perm, int hcap, const int32_t* d_sel, const uint32_t* d_sel_cnt,
                    const uint32_t* d_work, const uint32_t* d_n_work, GmFullRes* d_res, uint8_t* d_ops, int ops_stride,
                    uint8_t* d_back, size_t back_stride, int grid, unsigned long long* d_stats, hipStream_t stream,
                    const int32_t* d_sel_sidx = nullptr, int input_strand = 0, int write_back = 0,
                    uint32_t* d_order = nullptr, uint32_t* d_cls_cnt = nullptr);        // the work items listed by kind (as many words as d_work), four counter words; without them: the one-window kernel

// colour-space pass 2 (sw_full_cs per selected window); ops_stride bytes per result: backtrace bytes, then (genome << 4 | read) codes
// colour-space post_sw on the device (gm_post.hip): constants = CsPostConsts of gm_host.hip (logs taken on the host), one record per pass-2 result
struct GmCsPostDev { double let_m, let_x, col_m[2], col_x[2], pr_del_open, pr_del_extend, pr_ins_open, pr_ins_extend;
                     // reads with quality values (csfastq): qv[read][colour] = the colour's QV clamped to 0..250, qtab[2 q] / [2 q + 1] = log(1 - e(q)) / log(e(q) / 3) as the HOST's libm
                     // computes them (ref: sw-post.c:486-491), bq[result][read position] receives the base qualities (PHRED + 33, ref: :568-586); all null without QVs
                     const uint8_t* qv; const double* qtab; uint8_t* bq; };
struct GmPostRes { double posterior; int32_t cs_match, cs_mismatch, cs_xover, valid; };
#define GM_POST_THREADS 131072      // k_post_sw_cs: one thread per pass-2 result, a column scratch each (6.9 KB at 50 colours); 32 768 threads were half a wave per SIMD for a kernel that waits on its scratch: cfg4 5.62 -> 6.00 M reads/s (tools/post_threads_cfg4.sh)
int gm_launch_post_sw_cs(const GmCsPostDev& K, const uint32_t* d_reads, const uint8_t* d_initbp, int read_len, int read_words, const GmFullRes* d_res, uint8_t* d_ops,
                         int ops_stride, const uint32_t* d_n_work, uint32_t res_cap, GmPostRes* d_post, double* d_fw, uint32_t* d_info, int threads, hipStream_t stream);
int gm_launch_pass2_cs(const GmIndexDev& ix, const GmScoreDev& sc, const int* cs_params9, const uint32_t* d_reads, const uint8_t* d_initbp, int n_reads,
                       int read_len, int read_words, int window_len, const GmHit* d_hits, int hcap, const int32_t* d_sel, const uint32_t* d_work,
                       const uint32_t* d_n_work, GmFullRes* d_res, uint8_t* d_ops, int ops_stride, uint32_t* d_back, size_t back_words, int grid,
                       unsigned long long* d_stats, hipStream_t stream, const int8_t* d_xover = nullptr,   // d_xover[n_reads][read_len]: per-position crossover scores (reads with QVs)
                       const int32_t* d_sel_sidx = nullptr,                                                // paired mode: sort index of every selected window
                       uint32_t* d_order = nullptr, uint32_t* d_cls_cnt = nullptr);                        // the work items listed by kind (as many words as d_work), four counter words

// paired mode (gm_pair.hip): mate ranges per window, pair top-K, saved marks, mate reversal
int gm_launch_revcomp_reads(uint32_t* d_reads, int n_reads, int read_len, int read_words, hipStream_t stream, const uint8_t* d_read_rna = nullptr);
int gm_launch_read_rna_flags(const uint32_t* d_reads, int n_reads, int read_len, int read_words, uint8_t* d_flags, hipStream_t stream);   // re->is_rna per packed letter-space read
struct MpDelta { int amin[2], amax[2], bmin[2], bmax[2]; };   // region deltas of mate 1 / mate 2 per strand (ref: mapping.c:2422-2430)
MpDelta gm_mp_region_deltas(int region_bits, const int* dmin1, const int* dmax1, const int* dmin2, const int* dmax2);
int gm_launch_mp_filter(int n_pairs, int region_bits, int region_overlap, uint64_t* d_surv1, uint32_t* d_cnt1, int scap1, uint64_t* d_surv2, uint32_t* d_cnt2, int scap2,
                        const int* dmin1, const int* dmax1, const int* dmin2, const int* dmax2, uint32_t* d_seg1, uint32_t* d_seg2, int n_slabs,
                        unsigned long long* d_unfiltered, hipStream_t stream);
int gm_launch_pair_up(int n_pairs, const GmHit* hits1, const uint16_t* perm1, const uint32_t* cnt1, int hcap1,
                      const GmHit* hits2, const uint16_t* perm2, const uint32_t* cnt2, int hcap2,
                      int32_t* pmin1, int32_t* pmax1, int32_t* pmin2, int32_t* pmax2, const int* delta_min, const int* delta_max, hipStream_t stream);
int gm_launch_pair_select(const GmScoreDev& sc, int n_pairs, int len1, int len2,
                          const GmHit* hits1, const uint16_t* perm1, const uint32_t* cnt1, int hcap1, const int32_t* pmin1, const int32_t* pmax1,
                          const GmHit* hits2, const uint16_t* perm2, const uint32_t* cnt2, int hcap2,
                          int32_t* sel1, int32_t* sidx1, uint32_t* selcnt1, int32_t* sel2, int32_t* sidx2, uint32_t* selcnt2,
                          uint32_t* pairs, uint32_t* pair_cnt, hipStream_t stream);
int gm_launch_mark_saved(uint8_t* d_saved, const uint32_t* d_list, int n, hipStream_t stream);

// S1 batch kernel on caller-provided bitfields
int gm_launch_sw_vector_batch(const GmScoreDev& sc, int n, const uint32_t* d_genome, const long long* d_goff, const int* d_glen,
                              const uint32_t* d_reads, int read_words, const int* d_rlen, int max_g, int max_r, int* d_scores, hipStream_t stream,
                              int early_thr = 0, uint8_t* d_stopped = nullptr);   // > 0: pass 1's early stop against this threshold, stopped[i] = 1 where it fired

// S2 single alignment on caller-provided bitfields
int gm_launch_sw_full_single(const GmScoreDev& sc, const uint32_t* d_genome, long long goff, int glen, const uint32_t* d_read, int rlen,
                             long long ax, long long ay, int alen, int awidth, int revcmpl, uint8_t* d_back, int* d_out, uint8_t* d_ops, int ops_cap,
                             hipStream_t stream, int has_anchor = 1, int thresh = 0, int maxscore = 0, int local = 0);   // no anchor: the threshold band; local: Gflag off

// colour space S1/S2 (gm_sw.hip): cs_params9 = match mismatch xover a_go a_ge b_go b_ge anchor_width indel_taboo_len (penalties positive)
int gm_launch_sw_gapless_batch(int n, int match, int mismatch, const uint32_t* d_genome, const uint32_t* d_genome_ls, const long long* d_woff, const int* d_glen,
                               const uint32_t* d_reads, int read_words, const int* d_rlen, const int* d_gidx, const int* d_ridx, const int* d_initbp, int max_r,
                               int* d_scores, hipStream_t stream);
int gm_launch_sw_vector_batch_cs(const GmScoreDev& sc, int n, const uint32_t* d_genome_cs, const uint32_t* d_genome_ls, const long long* d_goff,
                                 const int* d_glen, const uint32_t* d_reads, int read_words, const int* d_rlen, const int* d_initbp, int max_g, int max_r,
                                 int* d_scores, hipStream_t stream);
int gm_launch_sw_full_cs_single(const int* cs_params9, const uint32_t* d_genome_ls, long long goff, int glen, const uint32_t* d_read, int rlen, int initbp,
                                int thresh, long long ax, long long ay, int alen, int awidth, int revcmpl, uint32_t* d_back, int* d_out, uint8_t* d_ops,
                                int ops_cap, hipStream_t stream, int local = 0, const int8_t* d_xrow = nullptr);
