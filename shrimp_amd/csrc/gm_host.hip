// gm_host.hip -- host side of libgmapper_hip.so: handles, batching, and the part of the per-read
// pipeline the reference keeps on the CPU next to its output code:
//   hit_run_post_sw (LS)            ref: gmapper/mapping.c:1609-1625
//   read_pass2 selection + dedup    ref: gmapper/mapping.c:1520-1606,1661-1750
//   compute_unpaired_mqv            ref: gmapper/output.c:777-793
//   hit_output (SAM record)         ref: gmapper/output.c:227-774, make_cigar :15-64
// No CPU fallback exists for the device stages: without a HIP device every entry point fails.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <thread>
#include <mutex>
#include <condition_variable>
#include <deque>
#include <functional>
#include <future>
#include <unordered_map>
#include <zlib.h>
#include "gm_common.h"
#include "gm_internal.h"

// ---- errors -----------------------------------------------------------------------------------
static thread_local char g_err[512] = "";
void gm_set_error(const char* fmt, ...) {
  va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof g_err, fmt, ap); va_end(ap);
}
extern "C" const char* gm_last_error(void) { return g_err; }
extern "C" int gm_device_count(void) { int n = 0; if (hipGetDeviceCount(&n) != hipSuccess) return 0; return n; }
// The SAM text of a large call is hundreds of megabytes: allocating it afresh for every call (page faults on first touch) and unmapping it in gm_free cost 20+ ms per
// 1 M reads that nothing overlaps.  One freed text buffer of >= 16 MB is therefore parked here and handed to the next call (gm_release_cache() drops it).
static struct { std::mutex m; char* p = nullptr; size_t cap = 0; char* live = nullptr; size_t live_cap = 0; } g_outcache;
static char* outcache_take(size_t want, size_t* cap) {
  std::lock_guard<std::mutex> g(g_outcache.m);
  if (g_outcache.p && g_outcache.cap >= want) { char* r = g_outcache.p; *cap = g_outcache.cap; g_outcache.p = nullptr; g_outcache.cap = 0; return r; }
  return nullptr;
}
static void outcache_track(char* p, size_t cap) { std::lock_guard<std::mutex> g(g_outcache.m); g_outcache.live = p; g_outcache.live_cap = cap; }
static void gm_text_pool_drop();
extern "C" void gm_release_cache(void) { { std::lock_guard<std::mutex> g(g_outcache.m); free(g_outcache.p); g_outcache.p = nullptr; g_outcache.cap = 0; } gm_text_pool_drop(); }
extern "C" void gm_free(void* p) {
  if (!p) return;
  { std::lock_guard<std::mutex> g(g_outcache.m);
    if (p == g_outcache.live && g_outcache.live_cap >= ((size_t)16 << 20)) {
      g_outcache.live = nullptr;
      if (g_outcache.p && g_outcache.cap >= g_outcache.live_cap) { /* a larger one is parked already */ }
      else { free(g_outcache.p); g_outcache.p = (char*)p; g_outcache.cap = g_outcache.live_cap; return; }
    } else if (p == g_outcache.live) g_outcache.live = nullptr;
  }
  free(p);
}

extern "C" void gm_params_default(gm_params_t* p) {
  memset(p, 0, sizeof *p);
  p->match_score = 10; p->mismatch_score = -15;
  p->a_gap_open_score = -33; p->a_gap_extend_score = -7; p->b_gap_open_score = -33; p->b_gap_extend_score = -3;
  p->window_len = 140.0; p->window_overlap = 90.0; p->window_gen_threshold = 55.0;
  p->sw_vect_threshold = 50.0; p->sw_full_threshold = 50.0;     // LS: vect := full (ref: gmapper.c:2456-2458)
  p->match_mode = 2; p->num_outputs = 10; p->num_tmp_outputs = 30; p->anchor_width = 8;
  p->region_bits = 11; p->region_overlap = 50; p->list_cutoff = 0; p->hash_filter_calls = 1; p->tiebreak_rev = 1;
  p->sam_unaligned = 0; p->longest_read_len = 1000; p->strata = 0; p->max_alignments = 0;
  p->colour_space = 0; p->crossover_score = -20; p->indel_taboo_len = 0; p->pr_xover = 0.03; p->local_alignment = 0; p->ungapped = 0; p->hash_seeds = 0; p->output_format = 0; p->print_read_seq = 0; p->strand_only = 0;
  p->single_best_mapping = 0; p->all_contigs = 0; p->no_mapping_qualities = 0; p->no_improper_mappings = 0;
  p->extra_sam_fields = 0; p->sam_r2 = 0; memset(p->read_group, 0, sizeof p->read_group);
  p->trim_front = 0; p->trim_end = 0; p->trim_first = 1; p->trim_second = 1; p->trim_illumina = 0; p->min_avg_qv = 10; p->ignore_qvs = 0; p->no_qv_check = 0;      // ref: gmapper.h:63-67,75,81,104
}
extern "C" int gm_abi_sizeof(int which) {
  switch (which) { case 0: return (int)sizeof(gm_params_t); case 1: return (int)sizeof(gm_pair_opts_t); case 2: return (int)sizeof(gm_map_stats_t); case 3: return (int)sizeof(gm_merge_options_t); }
  return -1;
}
// compute_mapping_qualities (ref: gmapper.c:2258,2325-2328): off with --no-mapping-qualities and in local mode -- then MAPQ 255, no Z tags, no post_sw (mapping.c:1648)
static inline bool gm_mqv_on(const gm_params_t& P) { return !P.local_alignment && !P.no_mapping_qualities; }
// the gmapper-cs binary's defaults (ref: gmapper.c:1748-1755, gmapper-defaults.h:52-58,64-66)
extern "C" void gm_params_default_cs(gm_params_t* p) {
  gm_params_default(p);
  p->colour_space = 1; p->mismatch_score = -24; p->sw_vect_threshold = 47.0; p->sw_full_threshold = 50.0;
}

static GmScoreDev make_score(const gm_params_t& P) {
  GmScoreDev s; memset(&s, 0, sizeof s);
  s.match = P.match_score;
  s.mismatch = P.colour_space ? P.match_score + P.crossover_score : P.mismatch_score;   // what f1_setup hands the vector filter (ref: gmapper.c:2933-2936)
  s.a_go = -P.a_gap_open_score; s.a_ge = -P.a_gap_extend_score; s.b_go = -P.b_gap_open_score; s.b_ge = -P.b_gap_extend_score;
  s.anchor_width = P.anchor_width; s.match_mode = P.match_mode; s.min_matches = P.match_mode;   // ref: gmapper.c:2625
  s.skip_strands = P.strand_only == 1 ? 2 : (P.strand_only == 2 ? 1 : 0);                       // -F: no strand 1; -C: no strand 0
  s.num_tmp_outputs = P.num_tmp_outputs; s.tiebreak_rev = P.tiebreak_rev; s.hash_filter_calls = P.hash_filter_calls; s.local = P.local_alignment ? 1 : 0; s.gapless = P.ungapped ? 1 : 0;
  auto frac = [](double thr, double* f, int* a) { if (thr < 0) { *f = -1.0; *a = (int)(-thr); } else { *f = thr / 100.0; *a = 0; } };
  frac(P.window_gen_threshold, &s.wgen_thr_frac, &s.wgen_abs);
  frac(P.sw_vect_threshold, &s.vect_thr_frac, &s.vect_abs);
  frac(P.sw_full_threshold, &s.full_thr_frac, &s.full_abs);
  return s;
}

// ---- index ------------------------------------------------------------------------------------
GmIndexDev GmIndexHost::dev_view() const {
  GmIndexDev d; memset(&d, 0, sizeof d);
  d.genome = d_genome; d.genome_cs = d_genome_cs; d.colour = params.colour_space ? 1 : 0; d.hflag = params.hash_seeds ? 1 : 0; d.total_len = total_len; d.n_contigs = n_contigs; d.contig_off = d_contig_off;
  d.contig_rna = d_contig_rna; d.genome_is_rna = genome_is_rna; d.read_rna = nullptr;
  d.n_seeds = n_seeds; d.min_seed_span = min_seed_span; d.max_seed_span = max_seed_span;
  d.slab_bits = slab_bits; d.n_slabs = n_slabs; d.region_bits = params.region_bits; d.region_overlap = params.region_overlap;
  d.list_cutoff = list_cutoff;
  for (int i = 0; i < n_seeds; i++) {
    d.seed[i].mask = seeds[i].mask; d.seed[i].span = seeds[i].span; d.seed[i].weight = seeds[i].weight;
    d.seed[i].dir = seeds[i].d_dir; d.seed[i].pos = seeds[i].d_pos; d.seed[i].bkt = seeds[i].d_bkt; d.seed[i].n_pos = seeds[i].n_pos;
    d.seed[i].sdir = strips_ready ? seeds[i].d_sdir : nullptr; d.seed[i].spos = strips_ready ? seeds[i].d_spos : nullptr;
  }
  return d;
}

static int add_seed(GmIndexHost* ix, const char* s) {   // add_spaced_seed, ref: gmapper/seeds.c:9-43
  if (ix->n_seeds >= GM_MAX_SEEDS) return GM_E_ARG;
  GmSeedHost& sd = ix->seeds[ix->n_seeds];
  sd.mask = 0; sd.span = (int)strlen(s); sd.weight = 0; sd.text = s;
  if (sd.span < 1 || sd.span > 32) return GM_E_ARG;
  for (int i = 0; i < sd.span; i++) {
    if (s[i] != '0' && s[i] != '1') return GM_E_ARG;
    sd.mask = (sd.mask << 1) | (uint64_t)(s[i] == '1'); sd.weight += (s[i] == '1');
  }
  if (sd.weight < 1 || (!ix->params.hash_seeds && sd.weight > 14)) return GM_E_ARG;   // MAX_SEED_WEIGHT, ref: gmapper-definitions.h:50, seeds.c:132-136
  sd.kbits = ix->params.hash_seeds ? 2 * GM_HASH_TABLE_POWER : 2 * sd.weight;
  ix->max_seed_span = std::max(ix->max_seed_span, sd.span);
  ix->min_seed_span = std::min(ix->min_seed_span, sd.span);
  ix->n_seeds++;
  return GM_OK;
}

static void choose_slabs(GmIndexHost* ix) {
  int bits = 12;
  while ((1ull << bits) < ix->total_len) bits++;
  int sb = std::min(bits, 29);
  if (const char* e = gm_tune("GM_SLAB_BITS")) sb = std::max(ix->params.region_bits + 2, std::min(31, atoi(e)));
  ix->slab_bits = sb;
  ix->n_slabs = (int)((ix->total_len + (1ull << sb) - 1) >> sb);
  if (ix->n_slabs < 1) ix->n_slabs = 1;
}

// shared by gm_index_build and gm_index_load: contig table, cutoff, slabs, genome re-packed into global coordinates and uploaded
static int index_prepare(gm_index* ix, int n_contigs, const uint32_t* const* contigs, const uint32_t* contig_len, const char* const* contig_names) {
  ix->n_contigs = n_contigs;
  ix->contig_off.resize(n_contigs + 1);
  uint64_t tot = 0;
  for (int c = 0; c < n_contigs; c++) {
    ix->contig_off[c] = (uint32_t)tot; tot += contig_len[c];
    char nm[64]; snprintf(nm, sizeof nm, "contig%d", c + 1);
    ix->names.push_back(contig_names && contig_names[c] ? contig_names[c] : nm);
  }
  if (tot >= (1ull << 32)) { gm_set_error("genome of %llu bp exceeds the reference's 32-bit global coordinates", (unsigned long long)tot); return GM_E_ARG; }
  ix->contig_off[n_contigs] = (uint32_t)tot;
  ix->total_len = tot;
  // automatic list cutoff (ref: gmapper.c:2811-2837): max(1000, 100*total/4^maxW)
  if (ix->params.list_cutoff == 0) {
    int maxw = 0; for (int i = 0; i < ix->n_seeds; i++) maxw = std::max(maxw, ix->seeds[i].weight);
    if (ix->params.hash_seeds) maxw = GM_HASH_TABLE_POWER;            // ref: gmapper.c:2820-2822
    uint32_t cutoff = 1000; unsigned long long p4 = 1ull << (2 * maxw);
    if ((uint32_t)((100ull * tot) / p4) > cutoff) cutoff = (uint32_t)((100ull * tot) / p4);
    ix->list_cutoff = cutoff;
  } else ix->list_cutoff = ix->params.list_cutoff;
  choose_slabs(ix);
  // re-pack the per-contig bitfields into one bitfield in global coordinates
  ix->genome_words = (tot + 7) / 8 + 64;
  std::vector<uint32_t> g(ix->genome_words, 0);
  for (int c = 0; c < n_contigs; c++) {
    const uint64_t off = ix->contig_off[c]; const uint32_t* src = contigs[c]; const uint64_t len = contig_len[c];
    const int sh = (int)(off & 7) * 4; uint64_t w0 = off >> 3; const uint64_t nw = (len + 7) / 8;
    for (uint64_t k = 0; k < nw; k++) {
      uint32_t w = src[k];
      if (k == nw - 1 && (len & 7)) w &= (1u << ((len & 7) * 4)) - 1u;
      g[w0 + k] |= w << sh;
      if (sh) g[w0 + k + 1] |= w >> (32 - sh);
    }
  }
  GM_HIP(hipMalloc(&ix->d_genome, ix->genome_words * 4));
  GM_HIP(hipMemcpy(ix->d_genome, g.data(), ix->genome_words * 4, hipMemcpyHostToDevice));
  GM_HIP(hipMalloc(&ix->d_contig_off, (size_t)(n_contigs + 1) * 4));
  GM_HIP(hipMemcpy(ix->d_contig_off, ix->contig_off.data(), (size_t)(n_contigs + 1) * 4, hipMemcpyHostToDevice));
  return GM_OK;
}

extern "C" void gm_index_free(gm_index_t* ix);
extern "C" int gm_index_build(gm_index_t** out, int device, int n_contigs, const uint32_t* const* contigs,
                              const uint32_t* contig_len, const char* const* contig_names,
                              int n_seeds, const char* const* seeds, const gm_params_t* params) {
  if (!out || n_contigs < 1 || !contigs || !contig_len) { gm_set_error("gm_index_build: bad arguments"); return GM_E_ARG; }
  if (gm_device_count() <= device) { gm_set_error("no HIP device %d (the seed index lives in HBM; there is no CPU path)", device); return GM_E_NODEVICE; }
  GM_HIP(hipSetDevice(device));
  gm_index* ix = new gm_index();
  ix->device = device;
  if (params) ix->params = *params; else gm_params_default(&ix->params);
  int rc = GM_OK;
  if (n_seeds == 0) {   // load_default_seeds(0), letter space: ref gmapper-defaults.h:212-227
    rc |= add_seed(ix, "11110111101111"); rc |= add_seed(ix, "1111011100100001111"); rc |= add_seed(ix, "1111000011001101111");
  } else for (int i = 0; i < n_seeds; i++) rc |= add_seed(ix, seeds[i]);
  if (rc != GM_OK) { delete ix; gm_set_error("invalid spaced seed"); return GM_E_ARG; }
  rc = index_prepare(ix, n_contigs, contigs, contig_len, contig_names);
  if (rc != GM_OK) { gm_index_free(ix); return rc; }
  hipStream_t stream; GM_HIP(hipStreamCreate(&stream));
  rc = gm_index_build_device(ix, stream);
  (void)hipStreamDestroy(stream);
  if (rc != GM_OK) { gm_index_free(ix); return rc; }
  *out = ix;
  return GM_OK;
}

extern "C" void gm_index_free(gm_index_t* ix) {
  if (!ix) return;
  (void)hipSetDevice(ix->device);
  (void)hipFree(ix->d_genome); (void)hipFree(ix->d_genome_cs); (void)hipFree(ix->d_contig_off); (void)hipFree(ix->d_contig_rna);
  for (int i = 0; i < ix->n_seeds; i++) { (void)hipFree(ix->seeds[i].d_dir); (void)hipFree(ix->seeds[i].d_pos); (void)hipFree(ix->seeds[i].d_bkt);
                                           (void)hipFree(ix->seeds[i].d_sdir); (void)hipFree(ix->seeds[i].d_spos); }
  delete ix;
}
extern "C" uint32_t gm_index_list_cutoff(const gm_index_t* ix) { return ix->list_cutoff; }
extern "C" int gm_index_n_slabs(const gm_index_t* ix) { return ix->n_slabs; }
extern "C" int gm_index_has_buckets(const gm_index_t* ix) { return ix->seeds[0].d_bkt != nullptr; }
extern "C" uint64_t gm_index_bytes(const gm_index_t* ix) {
  uint64_t b = ix->genome_words * 4 * (ix->d_genome_cs ? 2 : 1);
  for (int i = 0; i < ix->n_seeds; i++) b += (ix->seeds[i].dir_words + (uint64_t)ix->seeds[i].n_pos + (ix->seeds[i].d_bkt ? (16ull << ix->seeds[i].kbits) : 0ull) +
                                            (ix->seeds[i].d_sdir ? (1ull << ix->seeds[i].kbits) + 1 + ix->seeds[i].n_spos : 0ull)) * 4;
  return b;
}
extern "C" int gm_index_get_list(const gm_index_t* ix, int sn, uint32_t mapidx, uint32_t* len, uint32_t* positions, uint32_t cap) {
  if (sn < 0 || sn >= ix->n_seeds || mapidx >= (1u << ix->seeds[sn].kbits)) return GM_E_ARG;
  GM_HIP(hipSetDevice(ix->device));
  uint32_t be[2];
  GM_HIP(hipMemcpy(&be[0], ix->seeds[sn].d_dir + (size_t)mapidx * ix->n_slabs, 4, hipMemcpyDeviceToHost));
  GM_HIP(hipMemcpy(&be[1], ix->seeds[sn].d_dir + (size_t)(mapidx + 1) * ix->n_slabs, 4, hipMemcpyDeviceToHost));
  *len = be[1] - be[0];
  uint32_t n = std::min(*len, cap);
  if (n && positions) GM_HIP(hipMemcpy(positions, ix->seeds[sn].d_pos + be[0], (size_t)n * 4, hipMemcpyDeviceToHost));
  return GM_OK;
}
extern "C" int gm_index_device_array(const gm_index_t* ix, int kind, void** dev_ptr, uint64_t* bytes) {
  if (kind == 0) { *dev_ptr = ix->d_genome; *bytes = ix->genome_words * 4; return GM_OK; }
  if (kind == 1 + 3 * ix->n_seeds) { *dev_ptr = ix->d_genome_cs; *bytes = ix->d_genome_cs ? ix->genome_words * 4 : 0; return GM_OK; }   // colour translation of the genome
  const int sn = (kind - 1) / 3, what = (kind - 1) % 3; if (kind < 0 || sn < 0 || sn >= ix->n_seeds) return GM_E_ARG;
  if (what == 0) { *dev_ptr = ix->seeds[sn].d_dir; *bytes = (ix->seeds[sn].dir_words + 16) * 4; }
  else if (what == 1) { *dev_ptr = ix->seeds[sn].d_pos; *bytes = ((uint64_t)ix->seeds[sn].n_pos + 64) * 4; }
  else { *dev_ptr = ix->seeds[sn].d_bkt; *bytes = ix->seeds[sn].d_bkt ? (16ull << ix->seeds[sn].kbits) * 4 : 0; }
  return GM_OK;
}

// metadata blob: everything but the device arrays (fixed header + contig offsets + names + seed strings)
struct MetaHdr { uint64_t magic, total_len, genome_words; int32_t n_contigs, n_seeds, slab_bits, n_slabs; uint32_t list_cutoff, pad; gm_params_t params;
                 uint32_t n_pos[GM_MAX_SEEDS]; uint64_t dir_words[GM_MAX_SEEDS]; uint32_t has_bkt[GM_MAX_SEEDS]; };
extern "C" int gm_index_meta(const gm_index_t* ix, void* meta, uint64_t* meta_bytes) {
  std::string blob;
  MetaHdr h; memset(&h, 0, sizeof h);
  h.magic = 0x474D4958ull; h.total_len = ix->total_len; h.genome_words = ix->genome_words; h.n_contigs = ix->n_contigs; h.n_seeds = ix->n_seeds;
  h.slab_bits = ix->slab_bits; h.n_slabs = ix->n_slabs; h.list_cutoff = ix->list_cutoff; h.params = ix->params;
  for (int i = 0; i < ix->n_seeds; i++) { h.n_pos[i] = ix->seeds[i].n_pos; h.dir_words[i] = ix->seeds[i].dir_words; h.has_bkt[i] = ix->seeds[i].d_bkt != nullptr; }
  blob.append((const char*)&h, sizeof h);
  blob.append((const char*)ix->contig_off.data(), (size_t)(ix->n_contigs + 1) * 4);
  for (auto& n : ix->names) { blob += n; blob.push_back('\0'); }
  for (int i = 0; i < ix->n_seeds; i++) { blob += ix->seeds[i].text; blob.push_back('\0'); }
  if (meta && *meta_bytes >= blob.size()) memcpy(meta, blob.data(), blob.size());
  *meta_bytes = blob.size();
  return GM_OK;
}
extern "C" int gm_index_alloc_like(gm_index_t** out, int device, const void* meta, uint64_t meta_bytes) {
  if (meta_bytes < sizeof(MetaHdr)) return GM_E_ARG;
  MetaHdr h; memcpy(&h, meta, sizeof h);
  if (h.magic != 0x474D4958ull) return GM_E_ARG;
  if (gm_device_count() <= device) { gm_set_error("no HIP device %d", device); return GM_E_NODEVICE; }
  GM_HIP(hipSetDevice(device));
  gm_index* ix = new gm_index();
  ix->device = device; ix->params = h.params; ix->total_len = h.total_len; ix->genome_words = h.genome_words; ix->n_contigs = h.n_contigs;
  ix->slab_bits = h.slab_bits; ix->n_slabs = h.n_slabs; ix->list_cutoff = h.list_cutoff;
  const char* p = (const char*)meta + sizeof h;
  ix->contig_off.assign((const uint32_t*)p, (const uint32_t*)p + h.n_contigs + 1); p += (size_t)(h.n_contigs + 1) * 4;
  for (int c = 0; c < h.n_contigs; c++) { ix->names.push_back(p); p += strlen(p) + 1; }
  for (int i = 0; i < h.n_seeds; i++) { if (add_seed(ix, p) != GM_OK) { delete ix; return GM_E_ARG; } p += strlen(p) + 1; }
  GM_HIP(hipMalloc(&ix->d_genome, ix->genome_words * 4));
  if (ix->params.colour_space) GM_HIP(hipMalloc(&ix->d_genome_cs, ix->genome_words * 4));
  GM_HIP(hipMalloc(&ix->d_contig_off, (size_t)(h.n_contigs + 1) * 4));
  GM_HIP(hipMemcpy(ix->d_contig_off, ix->contig_off.data(), (size_t)(h.n_contigs + 1) * 4, hipMemcpyHostToDevice));
  for (int i = 0; i < h.n_seeds; i++) {
    ix->seeds[i].n_pos = h.n_pos[i]; ix->seeds[i].dir_words = h.dir_words[i];
    GM_HIP(hipMalloc(&ix->seeds[i].d_dir, (h.dir_words[i] + 16) * 4));
    GM_HIP(hipMalloc(&ix->seeds[i].d_pos, ((uint64_t)h.n_pos[i] + 64) * 4));
    if (h.has_bkt[i]) GM_HIP(hipMalloc(&ix->seeds[i].d_bkt, (16ull << ix->seeds[i].kbits) * 4));
  }
  *out = ix;
  return GM_OK;
}

// ---- session ----------------------------------------------------------------------------------
// One read set of a sub-batch: every device array K1..K4 need for reads of ONE length.  Unpaired
// mapping uses set[0]; paired mapping uses set[0] for the first mates and set[1] for the second.
struct DevSet {
  // capacities (grown on overflow)
  int cur_len = -1, scap = 0, scap2 = 0, hcap = 0, rcap_per_read = 8, ops_stride = 0, eff_batch = 0;
  int post_threads = 0;                                      // threads of k_post_sw_cs (each owns a column scratch)
  int p2_grid = 0;                                           // pass-2 grid for this read length: the session's, capped so that the back-pointer scratch stays within 2 GB
  int8_t* d_xover = nullptr; bool xover_on = false;        // colour space with QVs: per-position crossover scores [B][read_len]
  uint8_t* d_qv = nullptr; uint8_t* d_post_bq = nullptr;   // ... and the QVs themselves (clamped to 0..250) for post_sw on the device, which leaves the base qualities in d_post_bq [rcap][read_len]
  uint8_t* d_read_rna = nullptr;                             // letter space: 1 for a read with uracil and no thymine (re->is_rna, ref: fasta.c:528-542), set per sub-batch by k_read_rna_flags
  uint32_t* d_reads = nullptr; uint8_t* d_initbp = nullptr; uint64_t* d_surv = nullptr; uint32_t* d_surv_cnt = nullptr;   // d_initbp: colour space primer letters
  uint32_t* d_surv_seg = nullptr;                                          // [2B][S + 1] survivors after each slab (K1 emits slab by slab)
  uint64_t* d_surv2 = nullptr; uint32_t* d_surv_cnt2 = nullptr;            // survivors after the exact isolation prune (K1b) = input of K2
  GmHit* d_hits = nullptr; uint16_t* d_perm = nullptr; uint32_t* d_hit_cnt = nullptr; unsigned long long* d_slots = nullptr;
  uint32_t* d_heavy_list = nullptr; uint32_t* d_heavy_cnt = nullptr;   // read-strands beyond the LDS tier of K2
  int32_t* d_sel = nullptr; int32_t* d_sel_sidx = nullptr; uint32_t* d_sel_cnt = nullptr; uint32_t* d_sel_off = nullptr; uint32_t* d_work = nullptr; uint32_t* d_n_work = nullptr; uint32_t* d_p2_order = nullptr; uint32_t* d_p2_cls = nullptr;   // (pass 2's work items by kind, see k_p2cs_classify)
  GmFullRes* d_res = nullptr; uint8_t* d_ops = nullptr; uint8_t* d_back = nullptr; size_t back_stride = 0;
  GmPostRes* d_post = nullptr; double* d_post_fw = nullptr; uint32_t* d_post_info = nullptr;   // colour space: post_sw on the device (gm_post.hip), its per-thread scratch
  // paired mode only: mate range of every window (by sorted position) and the "saved" mark (by hit slot)
  int32_t* d_pmin = nullptr; int32_t* d_pmax = nullptr; uint8_t* d_saved = nullptr; uint32_t* d_saved_list = nullptr;
  int caps_pair_mode = 0;      // the paired match mode the capacities were chosen for
  uint32_t* d_mp_rows = nullptr; uint32_t* d_mp_cnt = nullptr; int mp_rows_for = 0;     // paired -n 3: the regions each read-strand marked twice (GmMpDev), for mp_rows_for read-strands
};

// Pinned host staging for one sub-batch's results: device-to-host copies run at link rate and the buffers are reused
// (three slots: one being filled while the host threads still work on the two sub-batches before it).
struct HostSlot {
  GmFullRes* res = nullptr; uint8_t* ops = nullptr; uint32_t* sel_cnt = nullptr; uint32_t* sel_off = nullptr; uint32_t* reads = nullptr;
  uint8_t* post_bq = nullptr; size_t post_bq_cap = 0; bool post_bq_on = false;   // base qualities of the device's post_sw (reads with QVs), read_len bytes per result
  GmPostRes* post = nullptr; size_t post_cap = 0; bool post_on = false;      // colour space: the device's post_sw results of this sub-batch (post_on: they are there)
  size_t res_cap = 0, ops_cap = 0, n_cap = 0, reads_cap = 0; uint32_t n_work = 0;
  bool want_reads = false;                                   // set by the caller: copy the sub-batch's packed reads back as well (reads handed over in device memory)
  unsigned long long* stats = nullptr; size_t stats_cap = 0;   // the sub-batch's stage counters as the device left them (pinned: their copy must not block the calling thread)
};
static int slot_reserve(void** p, size_t* cap, size_t bytes) {
  if (bytes <= *cap) return GM_OK;
  if (*p) (void)hipHostFree(*p);
  *p = nullptr; *cap = 0;
  const size_t want = bytes + bytes / 4 + 4096;
  GM_HIP(hipHostMalloc(p, want, hipHostMallocDefault));
  *cap = want;
  return GM_OK;
}
static void slot_free(HostSlot& h) {
  void* ptrs[] = {h.res, h.ops, h.sel_cnt, h.sel_off, h.reads, h.post, h.post_bq, h.stats};
  for (void* p : ptrs) if (p) (void)hipHostFree(p);
  h = HostSlot();
}

// A grow-only pinned host buffer that stands in for a std::vector as the target of a device-to-host copy (a copy into pageable memory is staged and blocks the caller)
template <class T> struct GmPinVec {
  T* p = nullptr; size_t cap = 0, n = 0;
  int resize(size_t k) { n = k; void* q = p; const int rc = slot_reserve(&q, &cap, k * sizeof(T)); p = (T*)q; return rc; }
  T* data() { return p; } const T* data() const { return p; } T& operator[](size_t i) { return p[i]; } const T& operator[](size_t i) const { return p[i]; } size_t size() const { return n; }
  void release() { if (p) (void)hipHostFree(p); p = nullptr; cap = 0; n = 0; }
};
struct gm_session {
  GmPinVec<GmFullRes> pin_res[8]; GmPinVec<uint8_t> pin_ops[8];   // paired path: results of the pairs' pass 2 (0, 1: the mates) and of the half-paired rescue (2, 3); 4-7: the same for
                                                                  // odd sub-batches (the output stage of a sub-batch reads them on its own thread while the next one's copies arrive)
  gm_session* twin = nullptr;                     // a second session on the same index with the same parameters, made by the file entry the first time a file has more than one
                                                  // chunk: it maps every other chunk, so that a chunk's tail runs under the next chunk's lookups (freed with this one)
  const gm_index* ix = nullptr;
  gm_params_t P; GmScoreDev sc;
  double score_alpha = 0, score_beta = 0;
  double pr_mismatch = .01, pr_del_open = 0, pr_del_extend = 0, pr_ins_open = 0, pr_ins_extend = 0;   // post_sw_setup's arguments (ref: gmapper.c:2568-2571,2959-2963)
  double* d_qtab = nullptr;                       // colour space: log(1 - e(q)), log(e(q) / 3) for q = 0..250 as the host's libm computes them, for post_sw on the device with QVs
  int max_batch = 0, p2_grid = 16384;             // pass-2 waves in flight (GM_P2_GRID); each owns a back-pointer scratch, see DevSet::p2_grid
  hipStream_t stream = nullptr;                   // front of the pipeline (reads in, K1, K1b, K2); the only stream of the paired path
  hipStream_t stream_b = nullptr;                 // back of the pipeline (pass 1, selection, pass 2, results out): runs beside the next sub-batch's front
  hipStream_t stream_c = nullptr;                 // host -> device copies of the next sub-batch (never queued behind kernels)
  hipEvent_t ev[12] = {};                          // (null until created: gm_session_free also takes a session whose set-up stopped half-way)
  hipEvent_t pev[2][10] = {};                      // per buffer set: stage boundaries, front done, copies done
  hipEvent_t pkev[2][4] = {};                      // paired mode: begin / end of K1 for mate set 0 and 1 of buffer pair k
  unsigned long long* d_pstats[2] = {nullptr, nullptr};   // per buffer set: counters of one sub-batch
  uint32_t* h_pin = nullptr;                      // pinned words the front writes (heavy count per set); from word 16 on: K1's start flags
  uint32_t flag_epoch = 0, front_epoch[2] = {0, 0}; int front_flag_grid[2] = {0, 0};
  DevSet set[2];
  DevSet set2[2];                                 // paired mode: the second pair of buffer sets of the two-stream pipeline
  HostSlot slot[3];
  uint32_t* d_pairs = nullptr; uint32_t* d_pair_cnt = nullptr; int pairs_cap = 0;   // paired mode: selected (mate 1, mate 2) window pairs
  unsigned long long* d_stats = nullptr;
  std::vector<uint32_t> h_genome;                 // host copy of the packed genome: only for the SHRiMP / pretty output formats, which print genome letters (fetched on first use)
  // last lookup timing
  double last_lookup_ms = 0; uint64_t last_lookup_bytes = 0; int last_lookup_launches = 0;
};

static void free_buffers(DevSet& D) {
  void* ptrs[] = {D.d_read_rna, D.d_xover, D.d_reads, D.d_initbp, D.d_surv, D.d_surv_cnt, D.d_surv_seg, D.d_surv2, D.d_surv_cnt2, D.d_hits, D.d_perm, D.d_hit_cnt, D.d_slots, D.d_heavy_list, D.d_heavy_cnt,
                  D.d_sel, D.d_sel_sidx, D.d_sel_cnt, D.d_sel_off, D.d_work, D.d_n_work, D.d_p2_order, D.d_p2_cls, D.d_res, D.d_ops, D.d_back,
                  D.d_pmin, D.d_pmax, D.d_saved, D.d_saved_list, D.d_post, D.d_post_fw, D.d_post_info, D.d_qv, D.d_post_bq, D.d_mp_rows, D.d_mp_cnt};
  for (void* p : ptrs) if (p) (void)hipFree(p);
  D.d_post = nullptr; D.d_post_fw = nullptr; D.d_post_info = nullptr; D.d_qv = nullptr; D.d_post_bq = nullptr;
  D.d_read_rna = nullptr; D.d_xover = nullptr; D.d_reads = nullptr; D.d_initbp = nullptr; D.d_surv = nullptr; D.d_surv_cnt = nullptr; D.d_surv_seg = nullptr; D.d_surv2 = nullptr; D.d_surv_cnt2 = nullptr; D.d_hits = nullptr; D.d_perm = nullptr; D.d_hit_cnt = nullptr; D.d_slots = nullptr;
  D.d_heavy_list = nullptr; D.d_heavy_cnt = nullptr; D.d_sel = nullptr; D.d_sel_sidx = nullptr; D.d_sel_cnt = nullptr; D.d_sel_off = nullptr;
  D.d_work = nullptr; D.d_n_work = nullptr; D.d_p2_order = nullptr; D.d_p2_cls = nullptr; D.d_res = nullptr; D.d_ops = nullptr; D.d_back = nullptr;
  D.d_pmin = nullptr; D.d_pmax = nullptr; D.d_saved = nullptr; D.d_saved_list = nullptr;
  D.d_mp_rows = nullptr; D.d_mp_cnt = nullptr; D.mp_rows_for = 0;
  D.cur_len = -1;
}

static int pow2ceil(long long v) { int p = 1; while (p < v) p <<= 1; return p; }
// host threads of this process's share (a multi-rank job divides the cores itself: GM_HOST_THREADS)
// The hardware's processor count, and the container's CPU quota where there is one (cgroup v2 cpu.max / v1 cfs quota).  A one-GPU box of this pool shows 256 processors
// and grants 16 CPUs' worth of time per period.  The worker count does NOT follow the quota: the finalisation comes in bursts (a ~10 ms job per sub-batch), the quota is
// accounted per 100 ms, and 32 threads that finish a burst early beat 16 that never exceed it -- measured on such a box: 100 Mbp workload 9.2-9.5 M reads/s with 32 threads,
// 8.5 M with 16; colour space 4.27 against 4.00 M; the 3 Gbp workload, whose host work averages 7 cores, is the same.  gm_usable_cores() is for reporting.
static int gm_usable_cores() {
  static const int cached = [] {
    int n = (int)std::max(1u, std::thread::hardware_concurrency());
    long long quota = -1, period = -1;
    if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) { char q[64] = ""; if (fscanf(f, "%63s %lld", q, &period) == 2 && strcmp(q, "max") != 0) quota = atoll(q); fclose(f); }
    else {
      if (FILE* g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) { if (fscanf(g, "%lld", &quota) != 1) quota = -1; fclose(g); }
      if (FILE* g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (fscanf(g, "%lld", &period) != 1) period = -1; fclose(g); }
    }
    if (quota > 0 && period > 0) n = std::min<long long>(n, std::max<long long>(1, (quota + period - 1) / period));
    return n;
  }();
  return cached;
}
static int gm_host_threads() {
  int n = (int)std::min<unsigned>(32, std::max(1u, std::thread::hardware_concurrency()));
  if (const char* e = getenv("GM_HOST_THREADS")) n = std::max(1, atoi(e));
  (void)&gm_usable_cores;
  return n;
}
// The host side's worker threads, kept for the life of the process (round 4: every sub-batch's finalisation started -- and joined -- up to 32 threads of its own, eight times
// per million reads).  gm_run_on_threads(n, f) runs f on n threads at once (n - 1 pool threads + the caller) and returns when all are done; calls from different threads
// (two finalisation jobs overlap) share the pool, which grows to the largest n asked for.
namespace {
struct GmThreadPool {
  // one call of run(): shared by the caller and the queue entries, so that whoever finishes last still finds it alive
  struct Call { const std::function<void()>* fn = nullptr; int left = 0; std::mutex m; std::condition_variable cv; };
  std::mutex m; std::condition_variable cv; std::deque<std::shared_ptr<Call>> q; std::vector<std::thread> workers; bool stop = false;
  ~GmThreadPool() { { std::lock_guard<std::mutex> lk(m); stop = true; } cv.notify_all(); for (auto& t : workers) if (t.joinable()) t.join(); }
  void loop() {
    for (;;) {
      std::shared_ptr<Call> c;
      { std::unique_lock<std::mutex> lk(m); cv.wait(lk, [&] { return stop || !q.empty(); }); if (q.empty()) return; c = std::move(q.front()); q.pop_front(); }
      (*c->fn)();
      { std::lock_guard<std::mutex> lk(c->m); if (--c->left == 0) c->cv.notify_all(); }
    }
  }
  void run(int n, const std::function<void()>& fn) {
    if (n <= 1) { fn(); return; }
    auto c = std::make_shared<Call>(); c->fn = &fn; c->left = n - 1;
    { std::lock_guard<std::mutex> lk(m);
      while ((int)workers.size() < n - 1) workers.emplace_back([this] { loop(); });
      for (int t = 1; t < n; t++) q.push_back(c); }
    cv.notify_all();
    fn();
    std::unique_lock<std::mutex> lk(c->m); c->cv.wait(lk, [&] { return c->left == 0; });      // (fn stays valid until here: every worker has returned from it)
  }
};
GmThreadPool& gm_pool() { static GmThreadPool* p = new GmThreadPool(); return *p; }      // (never destroyed: no join at process exit while a caller may still hold the library)
}
static void gm_run_on_threads(int n, const std::function<void()>& fn) { gm_pool().run(n, fn); }
// Text buffers of the finalisation chunks, recycled across sub-batches and calls: a fresh std::string per chunk and job meant ~250 MB of first-touch page faults per million reads.
namespace {
struct GmTextPool {
  std::mutex m; std::vector<std::string> free_;
  std::string take() { std::lock_guard<std::mutex> lk(m); if (free_.empty()) return std::string(); std::string t = std::move(free_.back()); free_.pop_back(); t.clear(); return t; }
  void give(std::string&& t) { if (t.capacity() == 0 || t.capacity() > (16u << 20)) return; std::lock_guard<std::mutex> lk(m); if (free_.size() < 512) free_.push_back(std::move(t)); }
  void drop() { std::lock_guard<std::mutex> lk(m); free_.clear(); free_.shrink_to_fit(); }
};
GmTextPool& gm_text_pool() { static GmTextPool* p = new GmTextPool(); return *p; }
}
static void gm_text_pool_drop() { gm_text_pool().drop(); }
// fn(begin, end) over [0, n) in pieces of `grain`, on the host threads
template <class F> static void gm_parallel_for(size_t n, size_t grain, F fn) {
  const size_t pieces = (n + grain - 1) / std::max<size_t>(1, grain);
  const int nt = (int)std::min<size_t>((size_t)gm_host_threads(), pieces);
  if (nt <= 1) { if (n) fn((size_t)0, n); return; }
  std::atomic<size_t> next(0);
  gm_run_on_threads(nt, [&]() { for (;;) { const size_t c = next.fetch_add(1); if (c >= pieces) break; fn(c * grain, std::min(n, (c + 1) * grain)); } });
}
static bool gm_fixed_lines(const char* text, size_t n, size_t want) {
  if (!n) return true;
  const size_t len = strlen(text), stride = want + 1;
  if (len != n * stride - 1 && len != n * stride) return false;
  std::atomic<bool> ok(true);
  gm_parallel_for(n, 16384, [&](size_t b, size_t e) {
    for (size_t i = b; i < e; i++) {
      const char* p = text + i * stride;
      if (memchr(p, '\n', want) != nullptr || (i + 1 < n && p[want] != '\n')) { ok = false; return; }
    }
  });
  return ok;
}

static int window_len_of(const gm_params_t& P, int read_len) {   // ref: gmapper.c:530
  double w = P.window_len < 0 ? -P.window_len : read_len * (P.window_len / 100.0);
  return (int)(uint16_t)w;
}

static int alloc_buffers(gm_session* s, DevSet& D, int read_len, bool paired = false) {
  free_buffers(D);
  const gm_index* ix = s->ix;
  const int read_words = (read_len + 7) / 8;
  const int W = window_len_of(s->P, read_len);
  // sub-batch size under a device-memory budget (candidate windows dominate when hcap has grown)
  const double budget = 16e9;
  const double per_read = 2.0 * ((double)(D.scap + D.scap2) * 8 + (double)D.hcap * (sizeof(GmHit) + 2 + 8)) + (double)D.rcap_per_read * (sizeof(GmFullRes) + read_len + W + 16);
  D.eff_batch = (int)std::max(64.0, std::min((double)s->max_batch, budget / per_read));
  const int B = D.eff_batch, rs = 2 * B;
  (void)ix;
  D.ops_stride = ((read_len + W + 15) / 16) * 16;
  D.back_stride = (((size_t)read_len * W + 255) / 256) * 256;
  if (s->P.colour_space) { D.ops_stride *= 2; D.back_stride *= 12; }   // backtrace byte + letter codes per column; three words of back pointers per cell
  GM_HIP(hipMalloc(&D.d_reads, (size_t)B * read_words * 4 + 64));
  if (!s->P.colour_space) GM_HIP(hipMalloc(&D.d_read_rna, (size_t)B + 64));
  if (s->P.colour_space) { GM_HIP(hipMalloc(&D.d_initbp, (size_t)B + 64)); GM_HIP(hipMalloc(&D.d_xover, (size_t)B * read_len + 64)); GM_HIP(hipMalloc(&D.d_qv, (size_t)B * read_len + 64)); }
  GM_HIP(hipMalloc(&D.d_surv, (size_t)rs * D.scap * 8));
  GM_HIP(hipMalloc(&D.d_surv_cnt, (size_t)rs * 4));
  if (D.scap2 > 0) {
    GM_HIP(hipMalloc(&D.d_surv2, (size_t)rs * D.scap2 * 8)); GM_HIP(hipMalloc(&D.d_surv_cnt2, (size_t)rs * 4));
    GM_HIP(hipMalloc(&D.d_surv_seg, (size_t)rs * (ix->n_slabs + 1) * 4));
  }
  GM_HIP(hipMalloc(&D.d_hits, (size_t)rs * D.hcap * sizeof(GmHit)));
  GM_HIP(hipMalloc(&D.d_perm, (size_t)rs * D.hcap * 2));
  GM_HIP(hipMalloc(&D.d_hit_cnt, (size_t)rs * 4));
  GM_HIP(hipMalloc(&D.d_slots, (size_t)rs * D.hcap * 8));
  GM_HIP(hipMalloc(&D.d_heavy_list, (size_t)rs * 4));
  GM_HIP(hipMalloc(&D.d_heavy_cnt, 4));
  GM_HIP(hipMalloc(&D.d_sel, (size_t)B * GM_SEL_MAX * 4));
  GM_HIP(hipMalloc(&D.d_sel_cnt, (size_t)B * 4));
  GM_HIP(hipMalloc(&D.d_sel_off, (size_t)B * 4));
  const size_t rcap = (size_t)B * D.rcap_per_read;
  GM_HIP(hipMalloc(&D.d_work, (size_t)B * GM_SEL_MAX * 4));
  GM_HIP(hipMalloc(&D.d_n_work, 4));
  GM_HIP(hipMalloc(&D.d_p2_order, (size_t)B * GM_SEL_MAX * 4)); GM_HIP(hipMalloc(&D.d_p2_cls, 16));
  GM_HIP(hipMalloc(&D.d_res, rcap * sizeof(GmFullRes)));
  GM_HIP(hipMalloc(&D.d_ops, rcap * D.ops_stride));
  if (s->P.colour_space && !getenv("GM_POST_SW_HOST")) {     // post_sw on the device (gm_post.hip): one record per result, forward values + column descriptors per thread
    GM_HIP(hipMalloc(&D.d_post, rcap * sizeof(GmPostRes)));
    D.post_threads = GM_POST_THREADS;
    while (D.post_threads > 16384 && (size_t)D.post_threads * (size_t)(read_len + 1) * 140 > ((size_t)2 << 30)) D.post_threads /= 2;      // (long reads: at most 2 GB of scratch a buffer set)
    if (const char* e = gm_tune("GM_POST_THREADS")) D.post_threads = std::max(4096, std::min(1 << 20, atoi(e) & ~63));
    GM_HIP(hipMalloc(&D.d_post_fw, (size_t)D.post_threads * (size_t)(read_len + 1) * 17 * 8));
    GM_HIP(hipMalloc(&D.d_post_info, (size_t)D.post_threads * (size_t)(read_len + 1) * 4));
    GM_HIP(hipMalloc(&D.d_post_bq, rcap * (size_t)read_len + 64));
  }
  D.p2_grid = (int)std::max<size_t>(256, std::min<size_t>((size_t)s->p2_grid, ((size_t)2 << 30) / D.back_stride));
  GM_HIP(hipMalloc(&D.d_back, (size_t)D.p2_grid * D.back_stride));
  if (paired) {
    GM_HIP(hipMalloc(&D.d_sel_sidx, (size_t)B * GM_SEL_MAX * 4));
    GM_HIP(hipMalloc(&D.d_pmin, (size_t)rs * D.hcap * 4)); GM_HIP(hipMalloc(&D.d_pmax, (size_t)rs * D.hcap * 4));
    GM_HIP(hipMalloc(&D.d_saved, (size_t)rs * D.hcap)); GM_HIP(hipMalloc(&D.d_saved_list, (size_t)B * GM_SEL_MAX * 4));
  }
  D.cur_len = read_len;
  return GM_OK;
}

// pair_mode: the paired match mode when called for pairs (0: unpaired).  Modes 3 and 2 have no prune rules (a single k-mer match can open a window) and K2 takes
// the lookup's rows directly: its LDS tier holds 4096 keys; mode 2 keeps every list entry
static void choose_caps(gm_session* s, DevSet& D, int read_len, int pair_mode = 0) {
  const gm_index* ix = s->ix;
  const int max_n_kmers = std::max(0, read_len - ix->min_seed_span + 1);
  double lists = 0, avg_len = 0;
  for (int i = 0; i < ix->n_seeds; i++) {
    lists += std::max(0, read_len - ix->seeds[i].span + 1);
    avg_len += (double)ix->seeds[i].n_pos / (double)(1ull << ix->seeds[i].kbits) / ix->n_seeds;
  }
  const double entries = lists * avg_len;
  const double region = (double)(1 << ix->params.region_bits) + ix->params.region_overlap;
  double expected = entries * std::min(1.0, entries * region / std::max(1.0, (double)ix->total_len)) + lists;
  if (pair_mode ? pair_mode == 2 : s->P.match_mode == 1) expected = entries + lists;                        // -n 1 (paired: -n 2): every list entry is kept
  // scap = capacity of the LDS tier of K2 (16 B of LDS per entry); read-strands beyond it take the heavy tier
  // chance partial matches echo on neighbouring offsets / other seeds, so the survivors come out ~1.6x the independence estimate
  D.scap = std::min(16384, std::max(256, pow2ceil((long long)(2.2 * expected) + 128)));
  D.hcap = 64;
  // K1b (exact isolation prune) shrinks K2's input; its LDS tier is sized for what typically remains
  // K1b's bounds assume the window-generation threshold: not in -U mode
  // (a read that maps keeps one survivor per list at its true place -- 251 at 100 bases -- before any chance survivor: a tier of 256 sent 2 % of the read-strands of the
  // 100 Mbp workload through the heavy tier, whose host round trips then cost a quarter of the step; the tier holds one and a half times the lists)
  D.scap2 = (s->P.match_mode == 2 && !s->P.ungapped && !gm_tune("GM_NO_PRUNE")) ? std::max(std::min(D.scap, std::max(256, pow2ceil((long long)(1.5 * lists)))), D.scap / 8) : 0;
  if (pair_mode == 2 || pair_mode == 3) { D.scap = std::min(D.scap, 4096); D.scap2 = 0; }
  if (const char* e = gm_tune("GM_SCAP")) D.scap = std::min(16384, std::max(64, pow2ceil(atoi(e))));
  if (const char* e = gm_tune("GM_SCAP2")) { if (D.scap2) D.scap2 = std::min(D.scap, std::max(64, pow2ceil(atoi(e)))); }
  if (const char* e = gm_tune("GM_HCAP")) D.hcap = std::min(32768, std::max(4, pow2ceil(atoi(e))));
  (void)max_n_kmers;
}

extern "C" void gm_session_free(gm_session_t* s);
extern "C" int gm_session_create(gm_session_t** out, const gm_index_t* ix, const gm_params_t* params, int max_batch_reads) {
  if (!out || !ix) return GM_E_ARG;
  GM_HIP(hipSetDevice(ix->device));
  gm_session* s = new gm_session();
  s->ix = ix; s->P = params ? *params : ix->params;
  // (the parameters are checked before anything is allocated; every failure further down goes through gm_session_free, which releases what exists by then)
  if (s->P.strand_only < 0 || s->P.strand_only > 2) { delete s; gm_set_error("strand_only %d: 0 (both), 1 (-F) or 2 (-C)", params ? params->strand_only : 0); return GM_E_ARG; }
  if (s->P.match_mode != 1 && s->P.match_mode != 2) { delete s; gm_set_error("match_mode %d: 1 or 2 (ref: gmapper.c:2624; 3 and 4 are paired-mode settings with mate-pair region counts)", s->P.match_mode); return GM_E_ARG; }
  if (s->P.ungapped && !s->P.local_alignment) { delete s; gm_set_error("ungapped mode needs local alignment (ref: gmapper.c:2330-2333)"); return GM_E_ARG; }
  if ((s->P.colour_space != 0) != (ix->params.colour_space != 0)) { delete s; gm_set_error("session and index disagree on colour space"); return GM_E_ARG; }
  s->sc = make_score(s->P);
#define GM_HIP_S(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { gm_set_error("%s: %s", #call, hipGetErrorString(e_)); gm_session_free(s); return GM_E_NODEVICE; } } while (0)
  // score -> probability derivation (ref: gmapper.c:2557-2572)
  if (s->P.colour_space) {   // pr_xover => alpha => pr_mismatch
    s->score_alpha = (double)s->P.crossover_score / (log(s->P.pr_xover / 3) / log(2.0));
    s->pr_mismatch = 1.0 / (1.0 + 1.0 / 3.0 * pow(2.0, ((double)s->P.match_score - (double)s->P.mismatch_score) / s->score_alpha));
  } else {                   // pr_mismatch => alpha
    s->pr_mismatch = .01;
    s->score_alpha = ((double)s->P.match_score - (double)s->P.mismatch_score) / (log((1 - s->pr_mismatch) / (s->pr_mismatch / 3.0)) / log(2.0));
  }
  s->score_beta = (double)s->P.match_score - 2 * s->score_alpha - s->score_alpha * log(1 - s->pr_mismatch) / log(2.0);
  s->pr_del_open = pow(2.0, (double)s->P.a_gap_open_score / s->score_alpha);
  s->pr_ins_open = pow(2.0, (double)s->P.b_gap_open_score / s->score_alpha);
  s->pr_del_extend = pow(2.0, (double)s->P.a_gap_extend_score / s->score_alpha);
  s->pr_ins_extend = pow(2.0, ((double)s->P.b_gap_extend_score - s->score_beta) / s->score_alpha);
  if (s->P.colour_space && !getenv("GM_POST_SW_HOST")) {
    // The per-colour error rates post_sw derives from quality values (ref: sw-post.c:486-491, pr_err_from_qv util.h:285-293; Sanger QVs, the binary's default),
    // tabulated here with the host's libm for every QV the formula distinguishes: the device kernel reads the very doubles the host routine computes.
    std::vector<double> qt(2 * 251);
    for (int q = 0; q <= 250; q++) {
      double e = q <= 0 ? .99999999 : (q >= 250 ? 1E-25 : pow(10.0, -(double)q / 10.0));
      if (e > .75) e = .75;
      qt[2 * q] = log(1 - e); qt[2 * q + 1] = log(e / 3.0);
    }
    GM_HIP_S(hipMalloc(&s->d_qtab, qt.size() * 8));
    GM_HIP_S(hipMemcpy(s->d_qtab, qt.data(), qt.size() * 8, hipMemcpyHostToDevice));
  }
  s->max_batch = std::max(64, std::min(max_batch_reads > 0 ? max_batch_reads : 131072, 1 << 20));
  if (const char* e = gm_tune("GM_P2_GRID")) s->p2_grid = std::max(64, std::min(65536, atoi(e)));
  // The front stream gets the higher queue priority: K1's fall-back kernels (a few workgroups that each want most of a CU's LDS) otherwise wait for milliseconds
  // behind the back stream's thousands of small pass-1 workgroups, which refill every CU the moment the persistent K1 grid has left it.
  { int lo = 0, hi = 0; (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
    if (getenv("GM_STREAM_PRIO_OFF") || hi == lo) { GM_HIP_S(hipStreamCreate(&s->stream)); }
    else GM_HIP_S(hipStreamCreateWithPriority(&s->stream, hipStreamDefault, hi)); }
  GM_HIP_S(hipStreamCreateWithFlags(&s->stream_b, hipStreamNonBlocking));
  GM_HIP_S(hipStreamCreateWithFlags(&s->stream_c, hipStreamNonBlocking));
  for (auto& e : s->ev) GM_HIP_S(hipEventCreate(&e));
  for (auto& row : s->pev) for (auto& e : row) GM_HIP_S(hipEventCreate(&e));
  for (auto& row : s->pkev) for (auto& e : row) GM_HIP_S(hipEventCreate(&e));
  GM_HIP_S(hipMalloc(&s->d_stats, (size_t)GS_STRIPES * GS_STRIDE * 8));
  for (auto& d : s->d_pstats) GM_HIP_S(hipMalloc(&d, (size_t)GS_STRIPES * GS_STRIDE * 8));
  GM_HIP_S(hipHostMalloc((void**)&s->h_pin, (16 + 1024) * 4, hipHostMallocDefault));
  memset(s->h_pin, 0, (16 + 1024) * 4);
  { const int rc = gm_index_derive_rna(const_cast<gm_index*>(ix), s->stream); if (rc) { gm_session_free(s); return rc; } }      // which contigs are RNA (once per index)
  // k_lookup_v5 streams large indexes with the help of per-list strip lists, derived once per index (not stored in the index files)
  if ((ix->n_slabs > 1 || gm_tune("GM_K1_V5")) && !gm_tune("GM_NO_V5") && ix->params.region_bits >= 9 && ix->params.region_bits <= 16) {
    const int rc = gm_index_derive_strips(const_cast<gm_index*>(ix), s->stream);
    if (rc) { gm_session_free(s); return rc; }
  }
  *out = s;
  return GM_OK;
#undef GM_HIP_S
}
extern "C" void gm_session_free(gm_session_t* s) {
  if (!s) return;
  if (s->twin) { gm_session_free(s->twin); s->twin = nullptr; }
  (void)hipSetDevice(s->ix->device);
  for (auto& v : s->pin_res) v.release();
  for (auto& v : s->pin_ops) v.release();
  free_buffers(s->set[0]); free_buffers(s->set[1]); free_buffers(s->set2[0]); free_buffers(s->set2[1]);
  for (auto& h : s->slot) slot_free(h);
  if (s->d_qtab) (void)hipFree(s->d_qtab);
  if (s->d_pairs) (void)hipFree(s->d_pairs);
  if (s->d_pair_cnt) (void)hipFree(s->d_pair_cnt);
  if (s->d_stats) (void)hipFree(s->d_stats);
  for (auto& d : s->d_pstats) if (d) (void)hipFree(d);
  if (s->h_pin) (void)hipHostFree(s->h_pin);
  for (auto& e : s->ev) if (e) (void)hipEventDestroy(e);
  for (auto& row : s->pev) for (auto& e : row) if (e) (void)hipEventDestroy(e);
  for (auto& row : s->pkev) for (auto& e : row) if (e) (void)hipEventDestroy(e);
  if (s->stream) (void)hipStreamDestroy(s->stream);
  if (s->stream_b) (void)hipStreamDestroy(s->stream_b);
  if (s->stream_c) (void)hipStreamDestroy(s->stream_c);
  delete s;
}

// ---- host finalisation --------------------------------------------------------------------------
struct FHit {            // one pass-2 candidate on the host (read_hit + sw_full_results subset)
  const GmFullRes* r; const uint8_t* ops;
  int score_full, pass2_key; double pct_score_full; double posterior; int mqv; double z0, z1;
  bool dev_post = false;          // colour space: posterior and re-called letters came from k_post_sw_cs (the host routine can still redo them)
  double z2, z3, pr_top_random, insert_size_denom, pr_missed_mp;   // paired mode (ref: sw-full-common.h:30-44)
  std::string db, qr, qual; int cs_match = 0, cs_mismatch = 0, cs_xover = 0;   // colour space: dbalign / qralign and the counts post_sw leaves (ref: sw-post.c:531-565)
};

static inline char* put_uint(char* p, unsigned long long v) {      // two digits per division (a 64-bit division is ~25 cycles; a record prints half a dozen numbers)
  static const char D2[201] = "00010203040506070809101112131415161718192021222324252627282930313233343536373839404142434445464748495051525354555657585960616263646566676869707172737475767778798081828384858687888990919293949596979899";
  if (v < 10) { *p++ = (char)('0' + v); return p; }
  char tmp[24]; int n = 0;
  if (v <= 0xFFFFFFFFull) { uint32_t w = (uint32_t)v; while (w >= 100) { const uint32_t r = w % 100; w /= 100; tmp[n++] = D2[2 * r + 1]; tmp[n++] = D2[2 * r]; } if (w >= 10) { tmp[n++] = D2[2 * w + 1]; tmp[n++] = D2[2 * w]; } else tmp[n++] = (char)('0' + w); }
  else { while (v >= 100) { const unsigned r = (unsigned)(v % 100); v /= 100; tmp[n++] = D2[2 * r + 1]; tmp[n++] = D2[2 * r]; } if (v >= 10) { tmp[n++] = D2[2 * v + 1]; tmp[n++] = D2[2 * v]; } else tmp[n++] = (char)('0' + v); }
  while (n) *p++ = tmp[--n];
  return p;
}
static inline char* put_int(char* p, long long v) { if (v < 0) { *p++ = '-'; return put_uint(p, (unsigned long long)(-v)); } return put_uint(p, (unsigned long long)v); }
static inline char* put_str(char* p, const char* s, size_t n) { memcpy(p, s, n); return p + n; }

static int qv_from_pr_corr(double pr_corr) {   // ref: common/util.h:267-283
  double pr_err = 1 - pr_corr;
  if (pr_err > .99999999) return 0; else if (pr_err < 1E-25) return 250;
  return (int)(-10.0 * log(pr_err) / log(10.0));
}
static int double_to_neglog(double x) { return (int)((double)1000 * -log(x)); }   // ref: common/util.h:297-301

static int cmp_gen_start(const FHit* a, const FHit* b) {   // ref: mapping.c:1485-1494
  if (a->r->cn != b->r->cn) return (int)a->r->cn - (int)b->r->cn;
  if (a->r->gen_st != b->r->gen_st) return a->r->gen_st - b->r->gen_st;
  return a->r->genome_start - b->r->genome_start;
}
static int cmp_gen_end(const FHit* a, const FHit* b) {     // ref: mapping.c:1496-1506
  if (a->r->cn != b->r->cn) return (int)a->r->cn - (int)b->r->cn;
  if (a->r->gen_st != b->r->gen_st) return a->r->gen_st - b->r->gen_st;
  return (-a->r->genome_start - a->r->rmapped + a->r->n_del - a->r->n_ins) - (-b->r->genome_start - b->r->rmapped + b->r->n_del - b->r->n_ins);
}
// A stable sort of the handful of mappings a read has: insertion sort below 24 elements (the same order as any stable sort; std::stable_sort asks the allocator for a merge
// buffer at every call -- three calls a read were a third of the finalisation's cycles), the library's merge sort above.
template <class It, class Less> static inline void gm_small_stable_sort(It b, It e, Less less) {
  const auto n = e - b;
  if (n < 2) return;
  if (n > 24) { std::stable_sort(b, e, less); return; }
  for (It i = b + 1; i != e; ++i) {
    auto x = *i; It j = i;
    while (j != b && less(x, *(j - 1))) { *j = *(j - 1); --j; }
    *j = x;
  }
}
template <class Cmp>
static void dedup_pass(std::vector<FHit*>& v, Cmp cmp) {     // ref: mapping.c:1552-1600 (glibc qsort == stable merge sort here)
  if (v.size() < 2) return;
  gm_small_stable_sort(v.begin(), v.end(), [&](const FHit* a, const FHit* b) { return cmp(a, b) < 0; });
  size_t i = 0, k = 0, n = v.size();
  while (i < n) {
    int mx = v[i]->pass2_key; size_t mi = i, j = i + 1;
    while (j < n && !cmp(v[i], v[j])) { if (v[j]->pass2_key > mx) { mx = v[j]->pass2_key; mi = j; } j++; }
    if (mi != k) v[k] = v[mi];
    k++; i = j;
  }
  v.resize(k);
}

static const char CODE2SEQ[17] = "ACGTNNNNNNNNNNNN";   // SEQ letter of an aligned read base (ref: output.c:485-533: non-ACGTN -> N)
static inline char seq_from_text(char c) {          // ref: gmapper/output.c:326-351
  switch (c) { case 'R': case 'Y': case 'S': case 'W': case 'K': case 'M': case 'B': case 'D': case 'H': case 'V': return 'N'; default: return c >= 'a' ? (char)(c - 32) : c; }
}
static inline char rc_char(char c) { switch (c) { case 'A': return 'T'; case 'C': return 'G'; case 'G': return 'C'; case 'T': return 'A'; default: return 'N'; } }

// ---- colour space: post_sw (ref: common/sw-post.c:639-758) for reads without quality values -----------------------------------
// A 16-state forward-backward over the aligned columns; state j = previous letter << 2 | letter.  The sums run in the reference's
// order in doubles through the same libm calls, so posterior (-> Z0/Z1, MAPQ, AS) carries the same bits; what is hoisted out of
// the reference's inner loops (node priors, exp() of the previous column, log() of a sum shared by four states) is computed from
// identical operands, once instead of four or sixteen times.
struct CsPostConsts {
  double let_m, let_x, col_m[2], col_x[2];            // log(1 - e), log(e / 3) for the letter and the two colour error rates
  double pr_del_open, pr_del_extend, pr_ins_open, pr_ins_extend;
  bool sanger = true; int qoff = 0;                   // read QVs: PHRED (Sanger) or Solexa-style odds (ref: sw-post.c:489-491); offset of the first colour's QV in the string
};
static CsPostConsts cs_post_consts_from(double pr_snp, double pr_xover, double pr_del_open, double pr_del_extend, double pr_ins_open, double pr_ins_extend,
                                        bool sanger, int qoff) {
  CsPostConsts c; const double ce[2] = {pr_xover, .75};
  c.let_m = log(1 - pr_snp); c.let_x = log(pr_snp / 3.0);
  for (int k = 0; k < 2; k++) { c.col_m[k] = log(1 - ce[k]); c.col_x[k] = log(ce[k] / 3.0); }
  c.pr_del_open = pr_del_open; c.pr_del_extend = pr_del_extend; c.pr_ins_open = pr_ins_open; c.pr_ins_extend = pr_ins_extend; c.sanger = sanger; c.qoff = qoff;
  return c;
}
static CsPostConsts cs_post_consts(const gm_session* s) {
  return cs_post_consts_from(s->pr_mismatch, s->P.pr_xover, s->pr_del_open, s->pr_del_extend, s->pr_ins_open, s->pr_ins_extend, true, 0);
}
// exp() of 16 state values of one column.  The 16 arguments take few distinct values (a state's prior has four possible values, the
// transition terms depend on one letter only), and exp is a pure function: computing it once per distinct bit pattern gives the very same
// doubles as 16 calls do.
static inline void exp_neg16(const double* in, double* out) {
  uint64_t seen[16]; double val[16]; int ns = 0;
  for (int k = 0; k < 16; k++) {
    uint64_t b; memcpy(&b, &in[k], 8);
    int j = 0; while (j < ns && seen[j] != b) j++;
    if (j == ns) { seen[ns] = b; val[ns] = exp(-1 * in[k]); ns++; }
    out[k] = val[j];
  }
}

static void cs_post_sw(const CsPostConsts& K, const uint32_t* rw, int init_bp, int read_start, FHit& h,
                       const char* qual = nullptr, int qual_delta = 33, bool base_quals = false) {   // qual: the read's QV string (csfastq) or null; base_quals: h.qual even without it
  struct Col { double prior[16], fw[16], bw[16], fs, bs; int col, base_call; };
  std::string& db = h.db; std::string& qr = h.qr;
  static thread_local std::vector<Col> colbuf;
  if (colbuf.size() < db.size() + 1) colbuf.resize(db.size() + 1);
  Col* cols = colbuf.data(); int len = 0;
  auto colour = [&](int j) { return (int)((rw[j >> 3] >> ((j & 7) * 4)) & 0xf); };
  {  // load_local_vectors, ref: sw-post.c:448-528
    int start_run = 0, j, min_qv = 10000;
    for (j = 0; j < read_start; j++) {
      const int c = colour(j);
      if (c == 15) { start_run = 15; min_qv = 0; j = read_start; break; }
      start_run ^= c;
      if (qual) min_qv = std::min(min_qv, (int)qual[K.qoff + j]);
    }
    for (size_t i = 0; i < db.size(); i++) {
      if (qr[i] == '-') continue;
      Col& c = cols[len];
      int let = -2;                                     // -2: no letter emission (insertion); -1: a letter no state matches
      if (db[i] != '-') switch (db[i]) { case 'A': case 'a': let = 0; break; case 'C': case 'c': let = 1; break; case 'G': case 'g': let = 2; break;
                                         case 'T': case 't': let = 3; break; default: let = -1; }
      const int cc = colour(j); int which;
      if ((len == 0 && start_run == 15) || cc == 15) { c.col = 0; which = 1; } else { c.col = cc ^ (len == 0 ? start_run : 0); which = 0; }
      double col_m = K.col_m[which], col_x = K.col_x[which];
      if (qual && which == 0) {                         // the colour's own error rate, ref: sw-post.c:486-491 (use_sanger_qvs = true)
        const int qv = (len == 0 ? std::min(min_qv, (int)qual[K.qoff + j]) : (int)qual[K.qoff + j]) - qual_delta;
        double e = qv <= 0 ? .99999999 : (qv >= 250 ? 1E-25 : pow(10.0, -(double)qv / 10.0));
        if (!K.sanger) e /= (1 + e);
        if (e > .75) e = .75;
        col_m = log(1 - e); col_x = log(e / 3.0);
      }
      c.base_call = -1;
      switch (qr[i]) { case 'A': case 'a': c.base_call = 0; break; case 'C': case 'c': c.base_call = 1; break; case 'G': case 'g': c.base_call = 2; break;
                       case 'T': case 't': c.base_call = 3; break; default: break; }
      for (int st = 0; st < 16; st++) {                 // nodePrior, ref: sw-post.c:111-138
        const int l = (st >> 2) & 3, r = st & 3; double val = 0;
        if (let != -2) val = val - ((r == let) ? K.let_m : K.let_x);
        val = val - (((l ^ r) == c.col) ? col_m : col_x);
        c.prior[st] = val;
      }
      len++; j++;
    }
  }
  if (len == 0) { h.posterior = 0; return; }
  double total;
  {  // do_forwards, ref: sw-post.c:317-360
    Col& a = cols[0]; a.fs = 999999999;
    for (int j = 0; j < 16; j++) { if (((j >> 2) & 3) == init_bp) { a.fw[j] = a.prior[j]; a.fs = (a.fs < a.fw[j]) ? a.fs : a.fw[j]; } else a.fw[j] = HUGE_VAL; }
    for (int j = 0; j < 16; j++) a.fw[j] -= a.fs;
    for (int i = 1; i < len; i++) {
      Col& c = cols[i]; const Col& p = cols[i - 1];
      double e[16], lg[4];
      exp_neg16(p.fw, e);
      for (int l = 0; l < 4; l++) { double sum = 0; for (int k = l; k < 16; k += 4) sum += e[k]; lg[l] = log(sum); }   // states k with right(k) == l, ascending
      c.fs = 999999999;
      for (int j = 0; j < 16; j++) { c.fw[j] = c.prior[j] - lg[(j >> 2) & 3]; c.fs = (c.fs < c.fw[j]) ? c.fs : c.fw[j]; }
      for (int j = 0; j < 16; j++) c.fw[j] -= c.fs;
      c.fs += p.fs;
    }
    double val = 0;
    for (int j = 0; j < 16; j++) val += exp(-1 * (cols[len - 1].fw[j]));
    total = -log(val) + cols[len - 1].fs;
  }
  {  // do_backwards, ref: sw-post.c:269-315
    Col& z = cols[len - 1]; z.bs = 999999999;
    for (int j = 0; j < 16; j++) { z.bw[j] = 0; z.bs = (z.bs < z.bw[j]) ? z.bs : z.bw[j]; }
    for (int j = 0; j < 16; j++) z.bw[j] -= z.bs;
    for (int i = len - 2; i >= 0; i--) {
      Col& c = cols[i]; const Col& n = cols[i + 1];
      double e[16], nl[4];
      { double a[16]; for (int k = 0; k < 16; k++) a[k] = n.prior[k] + n.bw[k]; exp_neg16(a, e); }
      for (int r = 0; r < 4; r++) { double sum = 0; for (int k = 4 * r; k < 4 * r + 4; k++) sum += e[k]; nl[r] = -log(sum); }     // states k with left(k) == r
      c.bs = 999999999;
      for (int j = 0; j < 16; j++) { c.bw[j] = nl[j & 3]; c.bs = (c.bs < c.bw[j]) ? c.bs : c.bw[j]; }
      for (int j = 0; j < 16; j++) c.bw[j] -= c.bs;
      c.bs += n.bs;
    }
  }
  {  // post_traceback + fix_base_calls, ref: sw-post.c:183-212,531-565
    int j = 0, prev_base = init_bp; h.cs_match = h.cs_mismatch = h.cs_xover = 0; h.qual.clear();
    for (size_t i = 0; i < qr.size(); i++) {
      if (qr[i] == '-') continue;
      const Col& c = cols[j];
      double post[4] = {0, 0, 0, 0};
      { double a[16], ev[16]; for (int st = 0; st < 16; st++) a[st] = c.fw[st] + c.bw[st] + c.fs + c.bs - total; exp_neg16(a, ev);
        for (int st = 0; st < 16; st++) post[st & 3] += ev[st]; }
      int crt = 0; for (int b = 1; b < 4; b++) if (post[b] > post[crt]) crt = b;
      if (qual || base_quals) {                         // get_base_qualities, ref: sw-post.c:568-586: of the letter sw_full_cs had called
        int t = c.base_call >= 0 ? qv_from_pr_corr(post[c.base_call]) : 0;
        if (t > 40) t = 40;
        h.qual.push_back((char)(33 + t));
      }
      if ((prev_base ^ crt) == c.col) qr[i] = "ACGT"[crt]; else { qr[i] = "acgt"[crt]; h.cs_xover++; }
      if (db[i] != '-') { if (toupper((unsigned char)db[i]) == toupper((unsigned char)qr[i])) h.cs_match++; else h.cs_mismatch++; }
      prev_base = crt; j++;
    }
  }
  {  // get_posterior, ref: sw-post.c:589-612
    double res = exp(-total);
    for (size_t i = 0; i < db.size(); i++) {
      if (db[i] == '-') { res *= K.pr_ins_extend; if (i == 0 || db[i - 1] != '-') res *= K.pr_ins_open; }
      else if (qr[i] == '-') { res *= K.pr_del_extend; if (i == 0 || qr[i - 1] != '-') res *= K.pr_del_open; }
    }
    h.posterior = res;
  }
}
// dbalign / qralign of a colour-space alignment from the backtrace bytes and letter codes k_pass2_cs wrote (pretty_print, ref: sw-full-cs.c:945-1060)
// recalled: the read letters and lower-case marks k_post_sw_cs left in the spare bits of the backtrace bytes (gm_post.hip) instead of sw_full_cs's own
static void cs_alignment_strings(const uint8_t* bt, const uint8_t* codes, int n, std::string& db, std::string& qr, bool recalled = false) {
  static const char L[17] = "ACGTUMRWSYKVHDBN";
  db.clear(); qr.clear();
  for (int t = 0; t < n; t++) {
    const int type = bt[t] & 0x0f; const bool xov = recalled ? (bt[t] & 0x40) != 0 : (bt[t] & 0x80) != 0;
    if (type == 1) { db.push_back(L[codes[t] >> 4]); qr.push_back('-'); continue; }
    char q = recalled ? L[(bt[t] >> 4) & 3] : L[codes[t] & 15]; if (xov) q = (char)tolower((unsigned char)q);
    if (type >= 2 && type <= 5) { db.push_back('-'); qr.push_back(q); continue; }
    const char d = L[codes[t] >> 4];
    if (q == 'n' || q == 'N') q = xov ? (char)tolower((unsigned char)d) : d;      // an unknown read letter is shown as the genome's
    db.push_back(d); qr.push_back(q);
  }
}

// -DGM_HOST_PROFILE (diagnostic builds): where the finalisation's cycles go -- rdtsc sums per stage over all worker threads, printed by gm_host_profile_dump()
#ifdef GM_HOST_PROFILE
#include <x86intrin.h>
struct GmHpSlots { unsigned long long v[8] = {0, 0, 0, 0, 0, 0, 0, 0}; };
static std::mutex g_hp_m; static std::vector<GmHpSlots*> g_hp_all;
static GmHpSlots& gm_hp_mine() {      // per thread (the pool's threads live as long as the process): no shared counter is touched per read
  static thread_local GmHpSlots* mine = nullptr;
  if (!mine) { mine = new GmHpSlots(); std::lock_guard<std::mutex> lk(g_hp_m); g_hp_all.push_back(mine); }
  return *mine;
}
struct GmHp { unsigned long long t; GmHpSlots& S; GmHp() : t(__rdtsc()), S(gm_hp_mine()) {} void lap(int k) { const unsigned long long n = __rdtsc(); S.v[k] += n - t; t = n; } };
extern "C" void gm_host_profile_dump(void) {
  static const char* nm[8] = {"post_sw / FHit build", "dedup + sort", "mapping qualities", "record text", "unaligned record", "reads (count)", "", ""};
  unsigned long long g[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  { std::lock_guard<std::mutex> lk(g_hp_m); for (GmHpSlots* s : g_hp_all) for (int k = 0; k < 8; k++) { g[k] += s->v[k]; s->v[k] = 0; } }
  unsigned long long tot = 0; for (int k = 0; k < 5; k++) tot += g[k];
  for (int k = 0; k < 6; k++) { fprintf(stderr, "[host profile] %-22s %14llu%s %5.1f %%\n", nm[k], g[k], k < 5 ? " cycles" : "       ", k < 5 ? 100.0 * g[k] / (tot ? tot : 1) : 0.0); }
  if (g[5]) fprintf(stderr, "[host profile] %.0f cycles per read\n", (double)tot / (double)g[5]);
}
#define GM_HP_DECL GmHp hp_
#define GM_HP(k) hp_.lap(k)
#else
#define GM_HP_DECL do { } while (0)
#define GM_HP(k) do { } while (0)
#endif
struct Finalizer {
  const gm_session* s; int read_len, read_words; const uint32_t* reads;   // host copy of the packed reads of this sub-batch
  const char* const* name_ptr; const int* name_len; long name_base;
  const uint8_t* initbp = nullptr; int ops_half = 0; CsPostConsts csk = CsPostConsts();
  const GmFullRes* res_base = nullptr; const GmPostRes* post_base = nullptr;           // colour space: post_sw results of the device, parallel to the sub-batch's result records
  const uint8_t* bq_base = nullptr;                                                    // ... and, for reads with QVs, the base qualities it computed (read_len bytes per result)
  const char* const* seq_ptr = nullptr;                                                // text input: the read as it stood in the file (fields the reference prints from re->seq)
  const char* const* qual_ptr = nullptr; int qual_delta = 33;                          // FASTQ input: QUAL string of every read of this sub-batch   // colour space: primer letters of this sub-batch, ops_stride / 2
  const uint32_t* hgen = nullptr;                                                      // SHRiMP / pretty output: the packed genome on the host

  // dbalign / qralign of a letter-space result from its op record ('M' both, 'I' genome only, 'D' read only), the genome and the read: what sw_full_ls
  // hands to the output routines (ref: sw-full-ls.c pretty_print).  A reverse-strand result aligned the read to the reverse complement of the contig.
  void ls_alignment_strings(const GmFullRes& r, const uint8_t* ops, int nops, const uint32_t* rw, std::string& db, std::string& qr) const {
    static const char L[17] = "ACGTUMRWSYKVHDBN";
    const gm_index* ix = s->ix; const uint64_t base = ix->contig_off[r.cn]; const int glen = (int)(ix->contig_off[r.cn + 1] - ix->contig_off[r.cn]);
    auto gcode = [&](int j) -> int {
      const uint64_t p = base + (uint64_t)(r.gen_st ? glen - 1 - j : j); int c = (int)((hgen[p >> 3] >> ((p & 7) * 4)) & 0xf);
      if (r.gen_st) c = (int)((0xFBCDE56879A00123ull >> (c * 4)) & 0xf);                    // complement_base, ref: util.h:125-151
      return c; };
    db.clear(); qr.clear();
    int gi = r.genome_start, ri = r.read_start;
    for (int k = 0; k < nops; k++) {
      const char op = (char)ops[k];
      if (op == 'M') { db.push_back(L[gcode(gi++)]); qr.push_back(L[(rw[ri >> 3] >> ((ri & 7) * 4)) & 0xf]); ri++; }
      else if (op == 'I') { db.push_back(L[gcode(gi++)]); qr.push_back('-'); }
      else { db.push_back('-'); qr.push_back(L[(rw[ri >> 3] >> ((ri & 7) * 4)) & 0xf]); ri++; }
    }
  }
  // ref: common/output.c:36-115
  static void edit_string(const std::string& db, const std::string& qr, std::string& o) {
    const int len = (int)db.size(); int consec = 0; bool refgap = false; char nb[16];
    for (int i = 0; i <= len; i++) {
      if (i != len && db[i] == qr[i] && db[i] != '-') { consec++; continue; }
      if (refgap && (consec != 0 || (i == len ? '\0' : db[i]) != '-')) { o += ')'; refgap = false; }
      if (consec != 0) { o.append(nb, snprintf(nb, sizeof nb, "%d", consec)); consec = 0; }
      if (i == len) break;
      if (db[i] == '-') { if (islower((unsigned char)qr[i])) o += 'x'; if (!refgap) o += '('; o += (char)toupper((unsigned char)qr[i]); refgap = true; continue; }
      if (qr[i] == '-') o += '-';
      else if (db[i] == toupper((unsigned char)qr[i])) { o += 'x'; consec++; }
      else if (islower((unsigned char)qr[i])) { o += 'x'; o += (char)toupper((unsigned char)qr[i]); }
      else o += qr[i];
    }
  }
  // reverse_alignment_edit_string, ref: gmapper/output.c:83-122
  static void reverse_edit_string(std::string& e) {
    const int n = (int)e.size(); std::string r(e.size(), ' ');
    for (int i = 0; i < n;) {
      const char c = e[n - 1 - i];
      if (isdigit((unsigned char)c)) { int j = i + 1; while (j < n && isdigit((unsigned char)e[n - 1 - j])) j++; j--; memcpy(&r[i], &e[n - 1 - j], (size_t)(j - i + 1)); i = j + 1; }
      else { r[i] = c == ')' ? '(' : c == '(' ? ')' : c == 'A' ? 'T' : c == 'C' ? 'G' : c == 'G' ? 'C' : c == 'T' ? 'A' : c; i++; }
    }
    e.swap(r);
  }
  // the optional tail of a SAM record and its newline (ref: output.c:452-465,729-761): R2:Z / X2:Z (--sam-r2), RG:Z (--read-group), --extra-sam-fields of a mapping
  static void sam_tail(const gm_params_t& P, std::string& out, const std::string* mate_seq, const FHit* h, const std::string* db, const std::string* qr) {
    if (mate_seq) { out += P.colour_space ? "\tX2:Z:" : "\tR2:Z:"; out += *mate_seq; }
    if (P.read_group[0]) { out += "\tRG:Z:"; out.append(P.read_group, strnlen(P.read_group, sizeof P.read_group)); }
    if (h && P.extra_sam_fields) {
      char b[96]; const GmFullRes& r = *h->r;
      out.append(b, (size_t)snprintf(b, sizeof b, "\tZM:i:%d\tZR:i:%d\tZV:i:%d\tZH:i:%d\tZE:Z:", r.matches, r.score_window_gen, r.score_vector, r.score));
      std::string e; edit_string(*db, *qr, e);
      if (r.gen_st == 1) reverse_edit_string(e);
      out += e;
    }
    out += '\n';
  }
  // one mapping in the SHRiMP format, with the pretty rows when asked (ref: gmapper/output.c:270-296; common/output.c:280-352 output_normal, :118-262 output_pretty)
  void emit_shrimp(const FHit& h, const char* nm, size_t nl, int rd, const uint32_t* rw, const std::string& db, const std::string& qr, std::string& out) const {
    const gm_params_t& P = s->P; const gm_index* ix = s->ix; const GmFullRes& r = *h.r;
    const bool rev = r.gen_st == 1, cs = P.colour_space != 0;
    const uint32_t glen = (uint32_t)(ix->contig_off[r.cn + 1] - ix->contig_off[r.cn]);
    const uint32_t gs = (uint32_t)r.genome_start, ge = gs + (uint32_t)r.gmapped - 1;
    const uint32_t igs = rev ? glen - ge - 1 : gs, ige = rev ? glen - gs - 1 : ge;
    char b[160];
    out += '>'; out.append(nm, nl); out += '\t'; out += ix->names[r.cn]; out += '\t'; out += rev ? '-' : '+';
    out.append(b, snprintf(b, sizeof b, "\t%u\t%u\t%d\t%d\t%d\t%d\t", igs + 1, ige + 1, r.read_start + 1, r.read_start + r.rmapped, read_len, h.score_full));
    edit_string(db, qr, out); out += '\t';
    std::string rstr;                                                                   // readtostr, ref: common/output.c:17-34
    auto read_text = [&]() { static const char LS[17] = "ACGTUMRWSYKVHDBN", CS[17] = "0123!@#$%^&*?~;."; rstr.clear(); if (cs) rstr += LS[initbp[rd] & 15];
      for (int i = 0; i < read_len; i++) { const int c = (rw[i >> 3] >> ((i & 7) * 4)) & 0xf; rstr += cs ? CS[c] : LS[c]; } };
    if (P.print_read_seq) { read_text(); out += rstr; }
    out += '\n';
    if (P.output_format != 2) return;
    static const char LS[17] = "ACGTUMRWSYKVHDBN";
    const uint64_t base = ix->contig_off[r.cn];
    auto fwd_letter = [&](uint32_t j) { const uint64_t p = base + j; return LS[(hgen[p >> 3] >> ((p & 7) * 4)) & 0xf]; };   // (the reference reads the forward contig here on either strand)
    const uint32_t read_start = (uint32_t)r.read_start, read_end = (uint32_t)(r.read_start + r.rmapped - 1);
    std::string gpre, gpost, lspre, lspost, mpre;
    for (uint32_t j = 0; j < read_start; j++) { gpre += (gs + j > read_start) ? fwd_letter(gs - read_start + j) : '-'; lspre += '-'; mpre += ' '; }
    if (read_end < (uint32_t)read_len - 1) for (uint32_t j = 0; j < (uint32_t)read_len - read_end - 1; j++) { gpost += (ge + 1 + j < glen) ? fwd_letter(ge + 1 + j) : '-'; lspost += '-'; }
    out.append(b, snprintf(b, sizeof b, "G: %10lld    ", (long long)(rev ? ige + 1 : igs + 1))); out += gpre; out += db; out += gpost;
    out.append(b, snprintf(b, sizeof b, "    %-10lld\n", (long long)(rev ? igs + 1 : ige + 1)));
    out.append(b, snprintf(b, sizeof b, "%16s ", "")); out += mpre;
    for (size_t j = 0; j < db.size(); j++) {
      if (db[j] == qr[j] && db[j] != '-') out += '|';
      else if (db[j] == toupper((unsigned char)qr[j])) out += 'X';
      else if (islower((unsigned char)qr[j])) out += 'x';
      else out += ' ';
    }
    out += '\n';
    if (cs) { out.append(b, snprintf(b, sizeof b, "T: %10s    ", "")); out += lspre; out += qr; out += lspost; out += '\n'; }
    else { out.append(b, snprintf(b, sizeof b, "R: %10u    ", read_start + 1)); out += lspre; out += qr; out += lspost; out.append(b, snprintf(b, sizeof b, "    %-10u\n", read_end + 1)); }
    if (cs) {
      out.append(b, snprintf(b, sizeof b, "R: %10u   ", read_start + 1));
      read_text(); size_t k = 0; out += rstr[k++];
      for (uint32_t j = 0; j < read_start; j++) out += rstr[k++];
      for (size_t j = 0; k < rstr.size();) { if (j < qr.size() && qr[j] == '-') out += '-'; else out += rstr[k++]; if (j < qr.size()) j++; }
      out.append(b, snprintf(b, sizeof b, "    %-10u\n", read_end + 1));
    }
    out += '\n';
  }

  // Rounding guard for results of the device's post_sw (gm_post.hip): its posterior differs from the host routine's in the last bits (ocml exp / log instead of
  // glibc's), which matters only where a value is about to be rounded right at a boundary.  Every such conversion below -- rint() for AS, truncation for MAPQ
  // (with its two cut-offs) and for the Z tags -- checks the distance to the boundary; within guard_tol (1e-7, five orders above the difference) the read's device
  // results are redone by the host routine (same libm as the reference), so the bytes never hang on the device's last bits.
  double guard_tol = 1e-7; std::atomic<uint64_t>* redo_ctr = nullptr;
  bool near_rint(double x) const { const double f = x - floor(x); return fabs(f - 0.5) < guard_tol; }              // rint(): boundary at the half integers
  bool near_trunc(double x) const { return fabs(x - rint(x)) < guard_tol; }                                          // (int): boundary at the integers
  bool near_qv(double pr_corr) const {                                                                               // qv_from_pr_corr, ref: common/util.h:267-283
    const double pr_err = 1 - pr_corr;
    if (fabs(pr_err - .99999999) < guard_tol * 1e-2 || (pr_err > 0 && fabs(pr_err / 1E-25 - 1.0) < guard_tol)) return true;
    if (pr_err > .99999999 || pr_err < 1E-25) return false;
    return near_trunc(-10.0 * log(pr_err) / log(10.0));
  }
  void redo_on_host(FHit& h) const {                    // sw_full_cs's own strings, then the host's post_sw (ref: sw-post.c:636-758), then the posterior score
    const GmFullRes* r = h.r;
    cs_alignment_strings(h.ops, h.ops + ops_half, std::min(r->n_ops, ops_half), h.db, h.qr);
    cs_post_sw(csk, reads + (size_t)r->read_idx * read_words, (int)initbp[r->read_idx], r->read_start, h, qual_ptr ? qual_ptr[r->read_idx] : nullptr, qual_delta);
    h.dev_post = false;
    const double a = s->score_alpha, b = s->score_beta;
    int ps = (int)rint(a * log(h.posterior) / log(2.0) + (double)r->rmapped * (2.0 * a + b));
    if (ps < 0) ps = 0;
    h.score_full = ps; h.pct_score_full = (1000 * 100 * ps) / r->score_max;
    h.pass2_key = s->P.sw_full_threshold < 0 ? h.score_full : (int)h.pct_score_full;
    if (redo_ctr) redo_ctr->fetch_add(1, std::memory_order_relaxed);
  }
  // hit_run_post_sw for one pass-2 result (ref: mapping.c:1609-1625)
  void post_sw(FHit& h, const GmFullRes* r, const uint8_t* ops) const {
    const gm_params_t& P = s->P;
    h.r = r; h.ops = ops + (size_t)r->ops_off; h.mqv = 255; h.z0 = h.z1 = 0; h.posterior = 0; h.dev_post = false;
    h.z2 = h.z3 = h.pr_top_random = h.insert_size_denom = h.pr_missed_mp = 0;
    h.score_full = r->score;
    h.pct_score_full = (1000 * 100 * h.score_full) / r->score_max;                 // ref: mapping.c:400-401
    if (h.score_full > 0 && gm_mqv_on(P)) {                        // local mode / --no-mapping-qualities: no post_sw (ref: gmapper.c:2325-2328, mapping.c:1648)
      const double a = s->score_alpha, b = s->score_beta;
      if (P.colour_space) {
        if (post_base && (!qual_ptr || bq_base) && post_base[r - res_base].valid == 1) {   // k_post_sw_cs ran: the op record carries the re-called letters in its spare bits
          const GmPostRes& pr = post_base[r - res_base];
          cs_alignment_strings(h.ops, h.ops + ops_half, std::min(r->n_ops, ops_half), h.db, h.qr, true);
          h.posterior = pr.posterior; h.cs_match = pr.cs_match; h.cs_mismatch = pr.cs_mismatch; h.cs_xover = pr.cs_xover; h.qual.clear(); h.dev_post = true;
          if (qual_ptr) {                                                // base qualities of the read positions the alignment covers, in alignment order
            size_t nq = 0; for (char c : h.qr) nq += c != '-';
            h.qual.assign((const char*)bq_base + (size_t)(r - res_base) * read_len, std::min(nq, (size_t)read_len));
          }
          if (near_rint(a * log(h.posterior) / log(2.0) + (double)r->rmapped * (2.0 * a + b))) { redo_on_host(h); return; }      // AS at a rounding boundary
        } else {
        cs_alignment_strings(h.ops, h.ops + ops_half, std::min(r->n_ops, ops_half), h.db, h.qr);
        cs_post_sw(csk, reads + (size_t)r->read_idx * read_words, (int)initbp[r->read_idx], r->read_start, h,
                   qual_ptr ? qual_ptr[r->read_idx] : nullptr, qual_delta);
        }
      }
      int ps;
      if (!P.colour_space) {
        // letter space: posterior and score are functions of (score, mapped read length) alone -- a small per-thread table of the values libm gave for the pairs seen
        // last (reads of one length that map with 0, 1, 2 mismatches share a handful of pairs) instead of a pow and a log per mapping; the same doubles, so the same bytes
        struct Memo { int score, rmapped; double a, b, post; int ps; };
        static thread_local Memo memo[256];                        // (zero-initialised: a == 0 matches no session)
        Memo& m = memo[((unsigned)r->score * 31u + (unsigned)r->rmapped) & 255u];
        if (m.score != r->score || m.rmapped != r->rmapped || m.a != a || m.b != b) {
          m.score = r->score; m.rmapped = r->rmapped; m.a = a; m.b = b;
          m.post = pow(2.0, ((double)r->score - (double)r->rmapped * (2.0 * a + b)) / a);
          m.ps = (int)rint(a * log(m.post) / log(2.0) + (double)r->rmapped * (2.0 * a + b));
        }
        h.posterior = m.post; ps = m.ps;
      } else
      ps = (int)rint(a * log(h.posterior) / log(2.0) + (double)r->rmapped * (2.0 * a + b));
      if (ps < 0) ps = 0;
      h.score_full = ps; h.pct_score_full = (1000 * 100 * ps) / r->score_max;
    }
    else if (h.score_full > 0 && P.colour_space) {                 // colour space, local mode: no post_sw (mapping.c:1648): sw_full_cs's own strings and counts go out
      cs_alignment_strings(h.ops, h.ops + ops_half, std::min(r->n_ops, ops_half), h.db, h.qr);
      h.cs_match = r->n_match; h.cs_mismatch = r->n_mismatch; h.cs_xover = r->n_xover; h.qual.clear();
    }
    h.pass2_key = P.sw_full_threshold < 0 ? h.score_full : (int)h.pct_score_full;
  }
  // read_pass2's selection over the n pass-2 results of one read (ref: mapping.c:1628-1750): p2 = final hits in output order
  void select_hits(const GmFullRes* res, const uint8_t* ops, int n, std::vector<FHit>& fh, std::vector<FHit*>& p2) const {
    const gm_params_t& P = s->P;
    GM_HP_DECL;
    fh.clear(); p2.clear();
    fh.resize(n);
    for (int i = 0; i < n; i++) {
      FHit& h = fh[i]; post_sw(h, &res[i], ops);
      const double thr = P.sw_full_threshold < 0 ? -P.sw_full_threshold : h.r->score_max * (P.sw_full_threshold / 100.0);
      if (h.score_full >= thr) p2.push_back(&h);                                   // ref: mapping.c:1661 (double compare)
    }
    GM_HP(0);
    dedup_pass(p2, cmp_gen_start);
    dedup_pass(p2, cmp_gen_end);
    gm_small_stable_sort(p2.begin(), p2.end(), [](const FHit* a, const FHit* b) { return (b->pass2_key - a->pass2_key) < 0; });   // ref :1479-1482,1678
    if ((int)p2.size() > P.num_outputs) p2.resize(P.num_outputs);
    if (P.strata && !p2.empty()) { size_t i = 1; while (i < p2.size() && p2[0]->score_full == p2[i]->score_full) i++; p2.resize(i); }   // ref :1706-1712
    if (P.max_alignments != 0 && (int)p2.size() > P.max_alignments) p2.clear();                                                        // ref :1713-1722
    GM_HP(1);
  }

  // emits the SAM records of read `rd` (local index) into out; returns number of records
  int finalize_read(int rd, const GmFullRes* res, const uint8_t* ops, int ops_stride, int n, std::string& out, std::vector<FHit>& fh, std::vector<FHit*>& p2) const {
    const gm_params_t& P = s->P; const gm_index* ix = s->ix;
    select_hits(res, ops, n, fh, p2);
    GM_HP_DECL;
#ifdef GM_HOST_PROFILE
    hp_.S.v[5]++;
#endif
    const uint32_t* rw = reads + (size_t)rd * read_words;
    char nbuf[32]; const char* nm; size_t nl;
    if (name_ptr) { nm = name_ptr[rd]; nl = (size_t)name_len[rd]; }
    else { nbuf[0] = 'r'; nl = (size_t)(put_int(nbuf + 1, (long long)(name_base + rd)) - nbuf); nm = nbuf; }
    // room reserved per record (std::string::resize zero-fills it: a letter-space record writes SEQ + QUAL + fixed fields, a colour-space one also CQ / CS / XX)
    const size_t need = 64 + nl + (P.colour_space ? 8 : 3) * (size_t)read_len + 320;
    // colour space: the read as csfasta text, primer letter + colours ('.' for a skipped cycle), for the CS:Z tag (ref: output.c:451,730)
    auto put_csfasta = [&](char* p) { if (seq_ptr) return put_str(p, seq_ptr[rd], (size_t)read_len + 1);     // verbatim, as the reference prints re->seq
      *p++ = "ACGT"[initbp[rd] & 3]; for (int i = 0; i < read_len; i++) { int c = (rw[i >> 3] >> ((i & 7) * 4)) & 0xf; *p++ = (c < 4) ? (char)('0' + c) : '.'; } return p; };
    if (p2.empty()) {
      if (P.sam_unaligned) {                                                       // ref: output.c:411-466
        size_t o = out.size(); out.resize(o + need); char* p = &out[o];
        p = put_str(p, nm, nl); p = put_str(p, "\t4\t*\t0\t0\t*\t*\t0\t0\t", 17);
        if (P.colour_space) {                                                        // ref: output.c:353-355,441-451
          p = put_str(p, "*\t*\tCQ:Z:", 9);
          if (qual_ptr) p = put_str(p, qual_ptr[rd], (size_t)read_len); else *p++ = '*';
          p = put_str(p, "\tCS:Z:", 6); p = put_csfasta(p); out.resize(p - out.data()); sam_tail(P, out, nullptr, nullptr, nullptr, nullptr); return 1;
        }
        if (seq_ptr) for (int i = 0; i < read_len; i++) *p++ = seq_from_text(seq_ptr[rd][i]);
        else for (int i = 0; i < read_len; i++) { int c = (rw[i >> 3] >> ((i & 7) * 4)) & 0xf; *p++ = "ACGTUMRWSYKVHDBN"[c]; }      // (the reference prints the file's own letters here: a U stays a U)
        if (qual_ptr) { *p++ = '\t'; p = put_str(p, qual_ptr[rd], (size_t)read_len); }      // ref: output.c:419-421 (verbatim)
        else p = put_str(p, "\t*", 2);
        out.resize(p - out.data()); sam_tail(P, out, nullptr, nullptr, nullptr, nullptr);
        GM_HP(4);
        return 1;
      }
      return 0;
    }
    if (gm_mqv_on(P)) {                                                            // compute_unpaired_mqv, ref: output.c:777-793,975
      for (int pass = 0; pass < 2; pass++) {
        double z1 = 0.0;
        for (auto* h : p2) z1 += h->posterior;
        for (auto* h : p2) { h->z0 = h->posterior; h->z1 = z1; h->mqv = qv_from_pr_corr(h->posterior / z1); if (h->mqv < 4) h->mqv = 0; }
        // rounding guard (see post_sw above): MAPQ and the Z0 / Z1 tags of this read's records, when any posterior came from the device
        bool dev = false, near = false;
        for (auto* h : p2) dev = dev || h->dev_post;
        if (pass || !dev) break;
        near = near_trunc(1000.0 * -log(z1));
        for (auto* h : p2) near = near || near_qv(h->posterior / z1) || near_trunc(1000.0 * -log(h->z0));
        if (!near) break;
        for (auto* h : p2) if (h->dev_post) redo_on_host(*h);
      }
      if (P.single_best_mapping) {                                                 // the first mapping with the highest quality, ref: output.c:977-984
        size_t mx = 0;
        for (size_t i = 1; i < p2.size(); i++) if (p2[i]->mqv > p2[mx]->mqv) mx = i;
        FHit* best = p2[mx]; p2.assign(1, best);
      }
    }
    GM_HP(2);
    if (P.output_format) {                                                         // --shrimp-format / --pretty, ref: gmapper/output.c:270-296
      std::string db, qr;
      for (auto* h : p2) {
        if (P.colour_space) emit_shrimp(*h, nm, nl, rd, rw, h->db, h->qr, out);
        else { ls_alignment_strings(*h->r, h->ops, std::min(h->r->n_ops, ops_stride), rw, db, qr); emit_shrimp(*h, nm, nl, rd, rw, db, qr, out); }
      }
      return (int)p2.size();
    }
    for (auto* h : p2) {
      const GmFullRes& r = *h->r;
      // The record is assembled in a buffer on the stack and appended (its bound -- every CIGAR operation a run of its own -- is ~2 KB for a 250-byte record, and
      // std::string::resize zero-fills what it adds); a record beyond the buffer grows the string in place as before.
      const size_t bound = need + 12 * (size_t)r.n_ops + ix->names[r.cn].size();
      char stage[4096]; const bool staged = bound <= sizeof stage;
      if (!staged) { const size_t o = out.size(); out.resize(o + bound); }
      char* p = staged ? stage : &out[out.size() - bound];
      auto commit = [&](char* e) { if (staged) out.append(stage, (size_t)(e - stage)); else out.resize((size_t)(e - out.data())); };
      const bool rev = r.gen_st == 1;
      const int read_start = r.read_start + 1, read_end = read_start + r.rmapped - 1;
      const int glen = (int)(ix->contig_off[r.cn + 1] - ix->contig_off[r.cn]);
      p = put_str(p, nm, nl); *p++ = '\t';
      p = put_int(p, rev ? 16 : 0); *p++ = '\t';
      p = put_str(p, ix->names[r.cn].data(), ix->names[r.cn].size()); *p++ = '\t';
      int genome_start;
      if (!rev) genome_start = r.genome_start + 1;
      else genome_start = (glen - r.genome_start) - (read_end - read_start - r.n_del + r.n_ins);   // ref: output.c:626-634
      p = put_uint(p, (unsigned)genome_start); *p++ = '\t';
      p = put_int(p, h->mqv); *p++ = '\t';
      // CIGAR (make_cigar, ref: output.c:15-64): 'I' op = gap in the read -> D; 'D' op = gap in the genome -> I
      struct Run { int len; char op; }; Run runs[GM_MAX_OPS]; int nr = 0;
      const char clip = P.colour_space ? 'H' : 'S';                                 // ref: output.c:575-579
      if (read_start > 1) runs[nr++] = {read_start - 1, clip};
      if (P.colour_space) {
        const int na = (int)h->qr.size();
        for (int i = 0; i < na;) {
          const char op = h->qr[i] == '-' ? 'D' : (h->db[i] == '-' ? 'I' : 'M'); int j = i;
          while (j < na && (h->qr[j] == '-' ? 'D' : (h->db[j] == '-' ? 'I' : 'M')) == op) j++;
          if (nr < GM_MAX_OPS - 1) runs[nr++] = {j - i, op};
          i = j;
        }
      } else {
      const int nops = std::min(r.n_ops, ops_stride);
      for (int i = 0; i < nops;) {
        const char op = (char)h->ops[i]; int j = i; while (j < nops && h->ops[j] == (uint8_t)op) j++;
        if (nr < GM_MAX_OPS - 1) runs[nr++] = {j - i, op == 'M' ? 'M' : (op == 'I' ? 'D' : 'I')};
        i = j;
      }
      }
      if (read_end != read_len) runs[nr++] = {read_len - read_end, clip};
      if (!rev) for (int i = 0; i < nr; i++) { p = put_uint(p, (unsigned)runs[i].len); *p++ = runs[i].op; }
      else for (int i = nr - 1; i >= 0; i--) { p = put_uint(p, (unsigned)runs[i].len); *p++ = runs[i].op; }
      p = put_str(p, "\t*\t0\t0\t", 7);
      if (P.colour_space) {   // SEQ = the aligned letters post_sw called (ref: output.c:485-537), then the colour-space tags (:717-730)
        auto up = [](char c) { if (c >= 'a') c -= 32; return (c == 'A' || c == 'C' || c == 'G' || c == 'T' || c == 'N') ? c : 'N'; };
        const int na = (int)h->qr.size();
        if (!rev) { for (int i = 0; i < na; i++) if (h->qr[i] != '-') *p++ = up(h->qr[i]); }
        else for (int i = na - 1; i >= 0; i--) if (h->qr[i] != '-') *p++ = rc_char(up(h->qr[i]));
        *p++ = '\t';
        if (qual_ptr && !h->qual.empty()) {                                         // post_sw's base qualities, ref: output.c:613-621 (none in local mode: no post_sw there)
          const int nq = (int)h->qual.size();
          if (!rev) p = put_str(p, h->qual.data(), (size_t)nq); else for (int i = nq - 1; i >= 0; i--) *p++ = h->qual[i];
        } else *p++ = '*';
        p = put_str(p, "\tAS:i:", 6); p = put_int(p, h->score_full);
        if (gm_mqv_on(P) && !P.all_contigs) {                                        // ref: output.c:691-696
          p = put_str(p, "\tZ0:i:", 6); p = put_int(p, double_to_neglog(h->z0));
          p = put_str(p, "\tZ1:i:", 6); p = put_int(p, double_to_neglog(h->z1));
        }
        p = put_str(p, "\tNM:i:", 6); p = put_int(p, h->cs_mismatch + r.n_del + r.n_ins);
        if (qual_ptr) { p = put_str(p, "\tCQ:Z:", 6); p = put_str(p, qual_ptr[rd], (size_t)read_len); }   // ref: output.c:724-727
        p = put_str(p, "\tCS:Z:", 6); p = put_csfasta(p);
        p = put_str(p, "\tCM:i:", 6); p = put_int(p, h->cs_xover);
        p = put_str(p, "\tXX:Z:", 6); p = put_str(p, h->qr.data(), h->qr.size());
        commit(p); sam_tail(P, out, nullptr, h, &h->db, &h->qr);
        continue;
      }
      // SEQ: read bases in input orientation (aligned part from qralign == the read's own letters), revcomp on '-'
      // (text input: the clipped ends -- local mode only -- keep the file's letters, ref: output.c:326-351,482-533; on the reverse strand the reference's
      // reverse() knows no 'X' / 'U' and exits there, those print as N here)
      auto seq_at = [&](int i) -> char { const int c = (rw[i >> 3] >> ((i & 7) * 4)) & 0xf; return (seq_ptr && (i < read_start - 1 || i >= read_end)) ? seq_from_text(seq_ptr[rd][i]) : CODE2SEQ[c]; };
      if (!seq_ptr || (read_start == 1 && read_end == read_len)) {                // every base from its code: eight to a word, the reverse strand through the complement's letters
        static const char RC2SEQ[17] = "TGCANNNNNNNNNNNN";                         // rc_char(CODE2SEQ[c])
        if (!rev) { for (int i = 0; i < read_len; i += 8) { uint32_t x = rw[i >> 3]; const int m = std::min(8, read_len - i); for (int k = 0; k < m; k++) { *p++ = CODE2SEQ[x & 15u]; x >>= 4; } } }
        else { for (int i = read_len - 1; i >= 0;) { const uint32_t x = rw[i >> 3]; for (int k = i & 7; k >= 0; k--, i--) *p++ = RC2SEQ[(x >> (4 * k)) & 15u]; } }
      } else {
      if (!rev) for (int i = 0; i < read_len; i++) *p++ = seq_at(i);
      else for (int i = read_len - 1; i >= 0; i--) { const char c = seq_at(i); *p++ = c == '.' ? '.' : rc_char(c); }
      }
      if (qual_ptr) {                                                              // ref: output.c:539-570
        *p++ = '\t';
        const char* q = qual_ptr[rd]; const int dq = 33 - qual_delta;
        if (!rev) for (int i = 0; i < read_len; i++) *p++ = (char)(q[i] + dq);
        else for (int i = read_len - 1; i >= 0; i--) *p++ = (char)(q[i] + dq);
        p = put_str(p, "\tAS:i:", 6);
      } else p = put_str(p, "\t*\tAS:i:", 8);
      p = put_int(p, h->score_full);
      if (gm_mqv_on(P) && !P.all_contigs) {                                        // ref: output.c:691-696
        p = put_str(p, "\tZ0:i:", 6); p = put_int(p, double_to_neglog(h->z0));
        p = put_str(p, "\tZ1:i:", 6); p = put_int(p, double_to_neglog(h->z1));
      }
      p = put_str(p, "\tNM:i:", 6); p = put_int(p, r.n_mismatch + r.n_del + r.n_ins);
      commit(p);
      if (P.extra_sam_fields) { std::string db, qr; ls_alignment_strings(r, h->ops, std::min(r.n_ops, ops_stride), rw, db, qr); sam_tail(P, out, nullptr, h, &db, &qr); }
      else sam_tail(P, out, nullptr, nullptr, nullptr, nullptr);
    }
    GM_HP(3);
    return (int)p2.size();
  }
};

// Heavy tier of K2: the few read-strands whose survivors exceed the LDS tier (low-complexity reads,
// repeats).  Sizes are known now, so every array is allocated exactly, the keys are re-emitted by K1
// and sorted by one segmented radix sort; then K2 runs on global arrays.  Rare by construction.
// the view a kernel that reads the reads of set D gets: with that set's RNA flags (letter space; null in colour space)
static inline GmIndexDev gm_view_of(const GmIndexDev& dv, const DevSet& D) { GmIndexDev v = dv; v.read_rna = D.d_read_rna; return v; }

static int run_heavy_tier(gm_session* s, DevSet& D, const GmIndexDev& dv, int n, int read_len, int read_words, int W, int n_heavy,
                          unsigned long long* d_stats = nullptr, const GmScoreDev* sc = nullptr) {
  if (!sc) sc = &s->sc;
  hipStream_t q = s->stream;
  if (!d_stats) d_stats = s->d_stats;
  std::vector<uint32_t> list(n_heavy), cnt_all((size_t)n * 2);
  GM_HIP(hipMemcpyAsync(list.data(), D.d_heavy_list, (size_t)n_heavy * 4, hipMemcpyDeviceToHost, q));
  GM_HIP(hipMemcpyAsync(cnt_all.data(), D.d_surv_cnt, cnt_all.size() * 4, hipMemcpyDeviceToHost, q));
  GM_HIP(hipStreamSynchronize(q));
  std::sort(list.begin(), list.end());
  std::vector<uint64_t> off(n_heavy + 1); std::vector<uint32_t> segn(n_heavy), b32(n_heavy), e32(n_heavy);
  uint64_t tot = 0;
  for (int i = 0; i < n_heavy; i++) {
    const uint32_t c = cnt_all[list[i]];
    off[i] = tot; segn[i] = c; b32[i] = (uint32_t)tot; e32[i] = (uint32_t)(tot + c);
    tot += (uint64_t)std::max(64, pow2ceil(c));       // room for the padded window sort
  }
  off[n_heavy] = tot;
  if (tot >= (1ull << 32)) { gm_set_error("heavy tier: %llu keys in one sub-batch", (unsigned long long)tot); return GM_E_OVERFLOW; }
  uint32_t *d_list = nullptr, *d_segn = nullptr, *d_b32 = nullptr, *d_e32 = nullptr, *d_aux = nullptr, *d_nxt = nullptr, *d_ord = nullptr;
  uint64_t *d_off = nullptr, *d_kin = nullptr, *d_ks = nullptr;
  GM_HIP(hipMalloc(&d_list, (size_t)n_heavy * 4)); GM_HIP(hipMalloc(&d_segn, (size_t)n_heavy * 4)); GM_HIP(hipMalloc(&d_b32, (size_t)n_heavy * 4)); GM_HIP(hipMalloc(&d_e32, (size_t)n_heavy * 4));
  GM_HIP(hipMalloc(&d_off, (size_t)(n_heavy + 1) * 8));
  GM_HIP(hipMalloc(&d_kin, tot * 8)); GM_HIP(hipMalloc(&d_ks, tot * 8)); GM_HIP(hipMalloc(&d_aux, tot * 4)); GM_HIP(hipMalloc(&d_nxt, tot * 4)); GM_HIP(hipMalloc(&d_ord, tot * 4));
  GM_HIP(hipMemcpyAsync(d_list, list.data(), (size_t)n_heavy * 4, hipMemcpyHostToDevice, q));
  GM_HIP(hipMemcpyAsync(d_segn, segn.data(), (size_t)n_heavy * 4, hipMemcpyHostToDevice, q));
  GM_HIP(hipMemcpyAsync(d_b32, b32.data(), (size_t)n_heavy * 4, hipMemcpyHostToDevice, q));
  GM_HIP(hipMemcpyAsync(d_e32, e32.data(), (size_t)n_heavy * 4, hipMemcpyHostToDevice, q));
  GM_HIP(hipMemcpyAsync(d_off, off.data(), (size_t)(n_heavy + 1) * 8, hipMemcpyHostToDevice, q));
  GM_HIP(hipMemsetAsync(d_ks, 0xff, tot * 8, q));
  int rc = gm_launch_lookup_redo(gm_view_of(dv, D), D.d_reads, n, read_len, read_words, n_heavy, d_list, d_off, d_kin, d_stats, q);
  if (rc == GM_OK)
    rc = gm_launch_anchors_heavy(dv, *sc, n, read_len, W, n_heavy, d_list, d_off, d_segn, d_b32, d_e32, tot, d_kin, d_ks, d_aux, d_nxt, d_ord,
                                 D.d_hits, D.d_perm, D.d_hit_cnt, D.hcap, d_stats, q);
  GM_HIP(hipStreamSynchronize(q));
  (void)hipFree(d_list); (void)hipFree(d_segn); (void)hipFree(d_b32); (void)hipFree(d_e32); (void)hipFree(d_off);
  (void)hipFree(d_kin); (void)hipFree(d_ks); (void)hipFree(d_aux); (void)hipFree(d_nxt); (void)hipFree(d_ord);
  return rc;
}

// tight-cluster bound of the prune rules (gm_prune.hip): smallest window-generation threshold over all contigs, in survivors' x extent
static int prune_e_max(const gm_session* s, int read_len, int W) {
  int e_max = -1;
  if (gm_tune("GM_PRUNE_NO_RULE2")) return e_max;
  if (s->sc.match > 0 && s->sc.b_go >= 0 && s->sc.b_ge >= 0) {
    long long min_clen = 1ll << 40;
    for (int c = 0; c < s->ix->n_contigs; c++) min_clen = std::min<long long>(min_clen, (long long)s->ix->contig_off[c + 1] - s->ix->contig_off[c]);
    const int w_len = (int)std::min<long long>(W, min_clen);
    const int base = std::min(read_len, w_len) * s->sc.match;
    const int thr = s->sc.wgen_thr_frac < 0 ? s->sc.wgen_abs : (int)((double)base * s->sc.wgen_thr_frac);
    e_max = (thr + s->sc.match - 1) / s->sc.match - s->ix->max_seed_span - 1;
  }
  return e_max;
}

// K1 with K1b's prune fused where the chosen kernel can (k_lookup_v5); *fused tells launch_prune_anchors to skip K1b
static int launch_lookup(gm_session* s, DevSet& D, const GmIndexDev& dv, int n, int read_len, int read_words, int W, unsigned long long* d_stats, int* fused) {
  GmFusePrune f; f.d_surv2 = D.d_surv2; f.d_surv_cnt2 = D.d_surv_cnt2; f.scap2 = D.scap2; f.window_len = W; f.e_max = D.scap2 > 0 ? prune_e_max(s, read_len, W) : -1; f.fused = fused;
  *fused = 0;
  return gm_launch_lookup(gm_view_of(dv, D), D.d_reads, n, read_len, read_words, D.d_surv, D.d_surv_cnt, D.scap, D.d_heavy_list, D.d_heavy_cnt, 2 * D.eff_batch, d_stats, s->stream, D.d_surv_seg, &f);
}

// K1b + K2 on the survivors of K1
static int launch_prune_anchors(gm_session* s, DevSet& D, const GmIndexDev& dv, int n, int read_len, int W, unsigned long long* d_stats = nullptr, int fused = 0) {
  hipStream_t q = s->stream;
  if (!d_stats) d_stats = s->d_stats;
  if (D.scap2 > 0) {
    if (!fused) {
      const int e_max = prune_e_max(s, read_len, W);
      int rc = gm_launch_prune(n, read_len, W, e_max, s->ix->n_slabs, s->ix->slab_bits, D.d_surv, D.d_surv_cnt, D.d_surv_seg, D.scap, D.d_surv2, D.d_surv_cnt2, D.scap2, D.d_heavy_list, D.d_heavy_cnt, 2 * D.eff_batch, d_stats, q);
      if (rc) return rc;
    }
    return gm_launch_anchors(dv, s->sc, n, read_len, W, D.d_surv2, D.d_surv_cnt2, D.scap2, D.d_hits, D.d_perm, D.d_hit_cnt, D.hcap, d_stats, q);
  }
  return gm_launch_anchors(dv, s->sc, n, read_len, W, D.d_surv, D.d_surv_cnt, D.scap, D.d_hits, D.d_perm, D.d_hit_cnt, D.hcap, d_stats, q);
}

// One sub-batch on the device, in two halves that run on different streams so that the back of sub-batch i (pass 1 and 2: VALU-bound,
// 56-64 VGPRs, ~1 KB of LDS per wave) shares the CUs with the front of sub-batch i + 1 (K1: one 1 024-thread workgroup per CU that waits on
// memory most of the time and leaves 96 VGPRs per SIMD and 26 KB of LDS free).  Each half uses the buffer set k of its sub-batch only.
//
// the index as the kernels see it, with the session's per-call switches: -n 1 keeps every list entry (use_region_counts off, ref: gmapper.c:2610-2616)
static GmIndexDev session_view(const gm_session* s, const DevSet* D = nullptr) {
  GmIndexDev dv = s->ix->dev_view(); dv.no_region_counts = s->P.match_mode == 1 ? 1 : 0;
  if (D) dv.read_rna = D->d_read_rna;                        // (letter space: the flags k_read_rna_flags left for this sub-batch's reads)
  return dv;
}

// Front, stream A: K1, K1b, K2; nothing here waits on the host (the heavy count lands in pinned memory, event pev[k][6] marks the end).
static int pipeline_front(gm_session* s, int k, int n, int read_len) {
  DevSet& D = s->set[k];
  const GmIndexDev dv = session_view(s);
  const int read_words = (read_len + 7) / 8;
  const int W = window_len_of(s->P, read_len);
  hipStream_t q = s->stream;
  unsigned long long* d_stats = s->d_pstats[k];
  GM_HIP(hipMemsetAsync(d_stats, 0, (size_t)GS_STRIPES * GS_STRIDE * 8, q));
  GM_HIP(hipEventRecord(s->pev[k][0], q));
  if (D.d_read_rna) { const int rc0 = gm_launch_read_rna_flags(D.d_reads, n, read_len, read_words, D.d_read_rna, q); if (rc0) return rc0; }      // which reads are RNA (ref: fasta.c:528-542)
  gm_lookup_set_start_flags(s->h_pin + 16, 1024, ++s->flag_epoch);
  int fused = 0;
  int rc = launch_lookup(s, D, dv, n, read_len, read_words, W, d_stats, &fused);
  s->front_epoch[k] = s->flag_epoch; s->front_flag_grid[k] = rc ? 0 : gm_lookup_start_flag_grid();
  gm_lookup_set_start_flags(nullptr, 0, 0);
  if (rc) return rc;
  GM_HIP(hipEventRecord(s->pev[k][1], q));
  rc = launch_prune_anchors(s, D, dv, n, read_len, W, d_stats, fused);
  if (rc) return rc;
  GM_HIP(hipMemcpyAsync(&s->h_pin[k], D.d_heavy_cnt, 4, hipMemcpyDeviceToHost, q));
  GM_HIP(hipEventRecord(s->pev[k][2], q));
  GM_HIP(hipEventRecord(s->pev[k][6], q));
  return GM_OK;
}

// Back, stream B: heavy tier if any (stream A, rare), pass 1, selection, pass 2, results to the host slot.  Returns 1 when a capacity grew
// (the buffers of set k were re-allocated: the caller re-submits the sub-batch from the front).
static int pipeline_back(gm_session* s, int k, HostSlot& H, int n, int read_len, gm_map_stats_t* st, float* lookup_ms, bool next_front_queued = false,
                         const std::function<int()>* after_pass2 = nullptr) {
  DevSet& D = s->set[k];
  const GmIndexDev dv = session_view(s);
  const int read_words = (read_len + 7) / 8;
  const int W = window_len_of(s->P, read_len);
  const int overlap_abs = (int)(unsigned int)(s->P.window_overlap < 0 ? -s->P.window_overlap : W * (s->P.window_overlap / 100.0));   // ref: mapping.c:1289
  hipStream_t q = s->stream_b;
  unsigned long long* d_stats = s->d_pstats[k];
  {
    GM_HIP(hipEventSynchronize(s->pev[k][6]));
    const uint32_t n_heavy = s->h_pin[k];
    int rc;
    float ms[5];                                                     // (the front's times now: its events are recorded again when this set takes the sub-batch after next)
    GM_HIP(hipEventElapsedTime(&ms[0], s->pev[k][0], s->pev[k][1]));
    GM_HIP(hipEventElapsedTime(&ms[1], s->pev[k][1], s->pev[k][2]));
    if (n_heavy) {
      rc = run_heavy_tier(s, D, dv, n, read_len, read_words, W, (int)n_heavy, d_stats); if (rc) return rc;
      GM_HIP(hipEventRecord(s->pev[k][6], s->stream));
    }
    GM_HIP(hipStreamWaitEvent(q, s->pev[k][6], 0));
    if (next_front_queued && s->front_flag_grid[k ^ 1] > 0) {
      // K1 of the next sub-batch becomes runnable at the same moment as this pass 1.  Its persistent workgroups need a whole CU's LDS each; pass-1 waves
      // that get there first keep them off the CUs until pass 1 is over.  So: wait (bounded) until all of them have raised their flag.
      volatile uint32_t* f = s->h_pin + 16; const uint32_t want = s->front_epoch[k ^ 1]; const int g = s->front_flag_grid[k ^ 1];
      const auto t0 = std::chrono::steady_clock::now();
      for (;;) {
        int up = 0; for (int i = 0; i < g; i++) up += f[i] == want;
        if (up == g || std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() > 5.0) break;
      }
    }
    GM_HIP(hipEventRecord(s->pev[k][3], q));
    rc = gm_launch_pass1(gm_view_of(dv, D), s->sc, D.d_reads, n, read_len, read_words, W, overlap_abs, D.d_hits, D.d_perm, D.d_hit_cnt, D.hcap, D.d_slots, d_stats, q,
                         nullptr, nullptr, D.d_initbp, true);
    if (rc) return rc;
    GM_HIP(hipEventRecord(s->pev[k][4], q));
    rc = gm_launch_select(s->sc, n, read_len, D.d_hits, D.d_perm, D.d_hit_cnt, D.hcap, D.d_sel, D.d_sel_cnt, D.d_sel_off, D.d_work, D.d_n_work, q);
    if (rc) return rc;
    GM_HIP(hipEventRecord(s->pev[k][5], q));
    unsigned long long hs[GS_N]; uint32_t n_work = 0;
    std::vector<unsigned long long> hraw((size_t)GS_STRIPES * GS_STRIDE);
    auto fold = [&]() { for (int c = 0; c < GS_N; c++) { hs[c] = 0; for (int t = 0; t < GS_STRIPES; t++) hs[c] += hraw[(size_t)t * GS_STRIDE + c]; } };
    GM_HIP(hipMemcpyAsync(&n_work, D.d_n_work, 4, hipMemcpyDeviceToHost, q));
    GM_HIP(hipMemcpyAsync(hraw.data(), d_stats, hraw.size() * 8, hipMemcpyDeviceToHost, q));
    GM_HIP(hipStreamSynchronize(q));
    fold();
    const size_t rcap = (size_t)D.eff_batch * D.rcap_per_read;
    bool retry = false;
    if (hs[GS_OVERFLOW_SURV]) { gm_set_error("heavy list overflow"); return GM_E_OVERFLOW; }
    if (hs[GS_OVERFLOW_HITS]) { if (D.hcap >= 32768) { gm_set_error("window list overflow at capacity %d", D.hcap); return GM_E_OVERFLOW; } D.hcap *= 4; retry = true; }
    if (n_work > rcap) { if (D.rcap_per_read >= 32) { gm_set_error("pass-2 work overflow"); return GM_E_OVERFLOW; } D.rcap_per_read = std::min(32, D.rcap_per_read * 2); retry = true; }
    if (retry) {   // capacities grew: re-allocate and let the caller re-submit (the sub-batch size may have shrunk)
      if (st) st->retries++;
      GM_HIP(hipStreamSynchronize(s->stream));           // the other set's front may be in flight; nothing may touch freed buffers
      rc = alloc_buffers(s, D, read_len); if (rc) return rc;
      return 1;
    }
    if (s->P.colour_space) {
      const int cs9[9] = {s->P.match_score, s->P.mismatch_score, s->P.crossover_score, -s->P.a_gap_open_score, -s->P.a_gap_extend_score,
                          -s->P.b_gap_open_score, -s->P.b_gap_extend_score, s->P.anchor_width, s->P.indel_taboo_len};   // sw_full_cs_setup's arguments (ref: gmapper.c:2944-2947)
      rc = gm_launch_pass2_cs(dv, s->sc, cs9, D.d_reads, D.d_initbp, n, read_len, read_words, W, D.d_hits, D.hcap, D.d_sel, D.d_work, D.d_n_work,
                              D.d_res, D.d_ops, D.ops_stride, (uint32_t*)D.d_back, D.back_stride / 4, D.p2_grid, d_stats, q, D.xover_on ? D.d_xover : nullptr, nullptr, D.d_p2_order, D.d_p2_cls);
      // post_sw of every result on the device (with the reads' quality values where they have them: per-colour error rates from the host's table, base qualities back), unless
      // the alignment is local (no mapping qualities at all, ref: gmapper.c:2325-2328)
      H.post_on = rc == GM_OK && D.d_post && gm_mqv_on(s->P) && (!D.xover_on || s->d_qtab);
      H.post_bq_on = H.post_on && D.xover_on;
      if (H.post_on) {
        const CsPostConsts c = cs_post_consts(s);
        GmCsPostDev K; K.let_m = c.let_m; K.let_x = c.let_x; K.col_m[0] = c.col_m[0]; K.col_m[1] = c.col_m[1]; K.col_x[0] = c.col_x[0]; K.col_x[1] = c.col_x[1];
        K.pr_del_open = c.pr_del_open; K.pr_del_extend = c.pr_del_extend; K.pr_ins_open = c.pr_ins_open; K.pr_ins_extend = c.pr_ins_extend;
        K.qv = H.post_bq_on ? D.d_qv : nullptr; K.qtab = H.post_bq_on ? s->d_qtab : nullptr; K.bq = H.post_bq_on ? D.d_post_bq : nullptr;
        rc = gm_launch_post_sw_cs(K, D.d_reads, D.d_initbp, read_len, read_words, D.d_res, D.d_ops, D.ops_stride, D.d_n_work, (uint32_t)rcap, D.d_post, D.d_post_fw, D.d_post_info,
                                  D.post_threads, q);
      }
    } else
    rc = gm_launch_pass2(gm_view_of(dv, D), s->sc, D.d_reads, n, read_len, read_words, W, D.d_hits, D.d_perm, D.hcap, D.d_sel, D.d_sel_cnt, D.d_work, D.d_n_work,
                         D.d_res, D.d_ops, D.ops_stride, D.d_back, D.back_stride, D.p2_grid, d_stats, q, nullptr, 0, 0, D.d_p2_order, D.d_p2_cls);
    if (rc) return rc;
    GM_HIP(hipEventRecord(s->pev[k][7], q));
    { size_t cap;
      cap = H.res_cap; rc = slot_reserve((void**)&H.res, &cap, (size_t)n_work * sizeof(GmFullRes)); H.res_cap = cap; if (rc) return rc;
      cap = H.ops_cap; rc = slot_reserve((void**)&H.ops, &cap, (size_t)n_work * D.ops_stride); H.ops_cap = cap; if (rc) return rc;
      cap = H.n_cap; rc = slot_reserve((void**)&H.sel_cnt, &cap, (size_t)n * 4); if (rc) return rc;
      cap = H.n_cap; rc = slot_reserve((void**)&H.sel_off, &cap, (size_t)n * 4); H.n_cap = cap; if (rc) return rc; }
    H.n_work = n_work;
    if (!s->P.colour_space) H.post_on = false;
    if (!H.post_on) H.post_bq_on = false;
    if (H.post_on) { size_t cap = H.post_cap; rc = slot_reserve((void**)&H.post, &cap, (size_t)n_work * sizeof(GmPostRes)); H.post_cap = cap; if (rc) return rc; }
    if (H.post_bq_on) { size_t cap = H.post_bq_cap; rc = slot_reserve((void**)&H.post_bq, &cap, (size_t)n_work * read_len + 64); H.post_bq_cap = cap; if (rc) return rc; }
    // The stage counters first, then an event: from there on nothing of this set that the FRONT writes is read any more (the copies below take d_res / d_ops / d_sel_* /
    // d_post only), so the front of the sub-batch after next may be queued onto this set before the host waits for the results (after_pass2, see map_impl).
    { size_t cap = H.stats_cap; rc = slot_reserve((void**)&H.stats, &cap, hraw.size() * 8); H.stats_cap = cap; if (rc) return rc; }
    GM_HIP(hipMemcpyAsync(H.stats, d_stats, hraw.size() * 8, hipMemcpyDeviceToHost, q));
    if (H.want_reads) {                                              // reads handed over in device memory: the host's finalisation needs this sub-batch's letters too
      size_t cap = H.reads_cap; rc = slot_reserve((void**)&H.reads, &cap, (size_t)n * read_words * 4); H.reads_cap = cap; if (rc) return rc;
      GM_HIP(hipMemcpyAsync(H.reads, D.d_reads, (size_t)n * read_words * 4, hipMemcpyDeviceToHost, q));
    }
    GM_HIP(hipEventRecord(s->pev[k][9], q));
    if (n_work) {
      if (H.post_on) GM_HIP(hipMemcpyAsync(H.post, D.d_post, (size_t)n_work * sizeof(GmPostRes), hipMemcpyDeviceToHost, q));
      if (H.post_bq_on) GM_HIP(hipMemcpyAsync(H.post_bq, D.d_post_bq, (size_t)n_work * read_len, hipMemcpyDeviceToHost, q));
      GM_HIP(hipMemcpyAsync(H.res, D.d_res, (size_t)n_work * sizeof(GmFullRes), hipMemcpyDeviceToHost, q));
      GM_HIP(hipMemcpyAsync(H.ops, D.d_ops, (size_t)n_work * D.ops_stride, hipMemcpyDeviceToHost, q));
    }
    GM_HIP(hipMemcpyAsync(H.sel_cnt, D.d_sel_cnt, (size_t)n * 4, hipMemcpyDeviceToHost, q));
    GM_HIP(hipMemcpyAsync(H.sel_off, D.d_sel_off, (size_t)n * 4, hipMemcpyDeviceToHost, q));
    if (after_pass2) { rc = (*after_pass2)(); if (rc) return rc; }
    GM_HIP(hipStreamSynchronize(q));
    memcpy(hraw.data(), H.stats, hraw.size() * 8);
    fold();
    GM_HIP(hipEventElapsedTime(&ms[2], s->pev[k][3], s->pev[k][4]));
    GM_HIP(hipEventElapsedTime(&ms[3], s->pev[k][4], s->pev[k][5]));
    GM_HIP(hipEventElapsedTime(&ms[4], s->pev[k][5], s->pev[k][7]));
    *lookup_ms = ms[0];
    if (st) {
      st->lookups += hs[GS_LOOKUPS]; st->list_entries += hs[GS_ENTRIES]; st->list_bytes += 12ull * hs[GS_LOOKUPS] + 4ull * hs[GS_ENTRIES];
      st->survivors += hs[GS_SURVIVORS]; st->anchors += hs[GS_ANCHORS]; st->windows += hs[GS_WINDOWS];
      st->vec_calls += hs[GS_VEC_CALLS]; st->vec_cells += hs[GS_VEC_CELLS]; st->vec_bypassed += hs[GS_VEC_BYPASSED];
      st->full_calls += hs[GS_FULL_CALLS]; st->full_cells += hs[GS_FULL_CELLS]; st->exact_order_reads += hs[GS_EXACT_ORDER]; st->survivors_pruned += hs[GS_PRUNED];
      st->ms_lookup += ms[0]; st->ms_anchors += ms[1]; st->ms_pass1 += ms[2]; st->ms_select += ms[3]; st->ms_pass2 += ms[4];
    }
    s->last_lookup_bytes += 12ull * hs[GS_LOOKUPS] + 4ull * hs[GS_ENTRIES];
    return GM_OK;
  }
}

// both halves back to back on set 0 (stage dumps)
static int run_device_pipeline(gm_session* s, DevSet& D, HostSlot& H, int n, int read_len, gm_map_stats_t* st, float* lookup_ms) {
  (void)D;
  int rc = pipeline_front(s, 0, n, read_len);
  if (rc) return rc;
  return pipeline_back(s, 0, H, n, read_len, st, lookup_ms);
}

// The lookup kernels keep per-DEVICE scratch (fall-back lists, start flags, the rounds kernel's rows: gm_lookup.hip, gm_lookup5.hip), shared by every session on that
// device: mapping calls of different sessions on one device take turns (a single call already fills the GPU).  Sessions on different devices run side by side.
// (round 4: that scratch exists twice per device -- gm_lookup_set_scratch_slot -- so TWO mapping calls may be in flight on a device, e.g. the two sessions the file entry
// alternates its chunks between: the tail of one call, its last back half and host work, then runs under the other's lookups.  A third call waits for a free set.)
struct GmDevTurn {
  static std::mutex& m(int d) { static std::mutex a[16]; return a[d]; }
  static std::condition_variable& cv(int d) { static std::condition_variable a[16]; return a[d]; }
  static bool& busy(int d, int k) { static bool a[16][2] = {}; return a[d][k]; }
  int dev, slot;
  explicit GmDevTurn(const gm_session* s) : dev((int)((unsigned)s->ix->device & 15u)), slot(0) {
    std::unique_lock<std::mutex> lk(m(dev));
    cv(dev).wait(lk, [&] { return !busy(dev, 0) || !busy(dev, 1); });
    slot = busy(dev, 0) ? 1 : 0; busy(dev, slot) = true;
    gm_lookup_set_scratch_slot(slot);
  }
  ~GmDevTurn() { { std::lock_guard<std::mutex> lk(m(dev)); busy(dev, slot) = false; } cv(dev).notify_one(); }
  GmDevTurn(const GmDevTurn&) = delete; GmDevTurn& operator=(const GmDevTurn&) = delete;
};

static int map_impl(gm_session* s, int n_reads, int read_len, const uint32_t* reads_host, const void* reads_dev,
                    const char* names, int emit_sam, char** sam, size_t* sam_len, gm_map_stats_t* stats, const uint8_t* initbp_host = nullptr,
                    const char* quals = nullptr, int qual_delta = 33, const char* seq_text = nullptr, uint32_t* per_read_bytes = nullptr) {
  if (!s || n_reads < 0 || read_len < 1) { gm_set_error("gm_map_reads: bad arguments"); return GM_E_ARG; }
  if ((s->P.colour_space != 0) != (initbp_host != nullptr)) {
    gm_set_error(s->P.colour_space ? "colour-space session: use gm_map_reads_cs (colours + primer letters)" : "gm_map_reads_cs needs a colour-space session"); return GM_E_ARG; }
  if (read_len > s->P.longest_read_len || read_len >= 32768 / std::max(1, s->P.match_score)) { gm_set_error("read length %d out of range (ref: sw-vector.c:393-398)", read_len); return GM_E_RANGE; }
  const double tl_in = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
  GmDevTurn dev_turn(s);
  const double tl_turn = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
  GM_HIP(hipSetDevice(s->ix->device));
  DevSet& D = s->set[0];
  if (stats) memset(stats, 0, sizeof *stats);
  if (D.cur_len != read_len || (D.caps_pair_mode != 0 && D.caps_pair_mode != 4)) { choose_caps(s, D, read_len); D.caps_pair_mode = 0; int rc = alloc_buffers(s, D, read_len); if (rc) return rc; }
  const int read_words = (read_len + 7) / 8;
  if ((s->P.output_format || s->P.extra_sam_fields) && s->h_genome.empty()) {    // the SHRiMP / pretty formats and the ZE:Z edit string print genome letters: one download per session
    s->h_genome.resize(s->ix->genome_words);
    GM_HIP(hipMemcpy(s->h_genome.data(), s->ix->d_genome, s->ix->genome_words * 4, hipMemcpyDeviceToHost));
  }
  s->last_lookup_ms = 0; s->last_lookup_bytes = 0; s->last_lookup_launches = 0;
  // names
  std::vector<const char*> nptr; std::vector<int> nlen;
  std::vector<const char*> qptr; std::vector<int8_t> xbuf; std::vector<uint8_t> qvbuf; std::vector<const char*> sptr;
  if (seq_text && gm_fixed_lines(seq_text, (size_t)n_reads, (size_t)(read_len + (s->P.colour_space ? 1 : 0)))) {     // (the usual case: no search for the line ends)
    const size_t stride = (size_t)(read_len + (s->P.colour_space ? 1 : 0)) + 1; sptr.resize(n_reads);
    for (int i = 0; i < n_reads; i++) sptr[i] = seq_text + (size_t)i * stride;
  } else
  if (seq_text) {                                            // one line per read: read_len letters, or primer + read_len colours
    const char* p = seq_text; const int want = read_len + (s->P.colour_space ? 1 : 0);
    for (int i = 0; i < n_reads; i++) {
      const char* e = strchr(p, '\n'); if (!e) e = p + strlen(p);
      if ((int)(e - p) != want) { gm_set_error("read %d: %d characters, expected %d", i, (int)(e - p), want); return GM_E_ARG; }
      sptr.push_back(p); p = *e ? e + 1 : e;
    }
  }
  if (quals && gm_fixed_lines(quals, (size_t)n_reads, (size_t)read_len)) {
    qptr.resize(n_reads); for (int i = 0; i < n_reads; i++) qptr[i] = quals + (size_t)i * ((size_t)read_len + 1);
  } else
  if (quals) {
    const char* p = quals;
    for (int i = 0; i < n_reads; i++) {
      const char* e = strchr(p, '\n'); if (!e) e = p + strlen(p);
      if ((int)(e - p) != read_len) { gm_set_error("read %d: QUAL string of %d characters for %d bases", i, (int)(e - p), read_len); return GM_E_ARG; }
      qptr.push_back(p); p = *e ? e + 1 : e;
    }
  }
  if (names) { const char* p = names; for (int i = 0; i < n_reads; i++) { const char* e = strchr(p, '\n'); if (!e) e = p + strlen(p); nptr.push_back(p); nlen.push_back((int)(e - p)); p = *e ? e + 1 : e; } }
  // Sub-batches are pipelined: while the GPU works on sub-batch i+1, host threads finish sub-batch i
  // (pass-2 selection, MAPQ, SAM text).  Output stays in input order.
  struct Job {
    int idx = 0;
    HostSlot* hs = nullptr;
    const uint32_t* hreads = nullptr; int base = 0, n = 0;
    std::vector<std::string> outs; std::vector<uint64_t> cm, cr; double ms = 0;
    std::thread th;
  };
  std::vector<std::unique_ptr<Job>> jobs;
  int nthreads = (int)std::min<unsigned>(32, std::max(1u, std::thread::hardware_concurrency()));   // a multi-rank job divides the cores itself: GM_HOST_THREADS (bench.py: cores / ranks of this node)
  if (const char* e = getenv("GM_HOST_THREADS")) nthreads = std::max(1, atoi(e));
  const int ops_stride = D.ops_stride;
  // The SAM text is assembled as the jobs finish, in input order (each job appends after its predecessor), so that the copy -- and the
  // first touch of the output pages -- overlaps the device work instead of following it.
  struct OutBuf { char* p = nullptr; size_t len = 0, cap = 0; int turn = 0; bool failed = false; std::mutex m; std::condition_variable cv; ~OutBuf() { free(p); } } ob;
  std::atomic<uint64_t> post_redo(0);
  auto run_job = [&, nthreads, ops_stride](Job* J) {
    auto t0 = std::chrono::steady_clock::now();
    const int n = J->n; const int chunk = std::max(256, std::min(4096, n / (2 * nthreads)));   // small sub-batches (the ramp) still use every thread
    const int nchunks = (n + chunk - 1) / chunk;
    J->outs.assign(nchunks, std::string()); J->cm.assign(nchunks, 0); J->cr.assign(nchunks, 0);
    std::atomic<int> next(0);
    Finalizer F{s, read_len, read_words, J->hreads, names ? nptr.data() + J->base : nullptr, names ? nlen.data() + J->base : nullptr, (long)J->base};
    if (s->P.colour_space) { F.initbp = initbp_host + J->base; F.ops_half = ops_stride / 2; F.csk = cs_post_consts(s); if (J->hs->post_on) { F.res_base = J->hs->res; F.post_base = J->hs->post; if (J->hs->post_bq_on) F.bq_base = J->hs->post_bq; } }
    F.redo_ctr = &post_redo;
    if (const char* e = gm_tune("GM_POST_GUARD_TOL")) F.guard_tol = atof(e);          // (tests: a huge tolerance sends every result through the redo path)
    if (quals) { F.qual_ptr = qptr.data() + J->base; F.qual_delta = qual_delta; }
    if (seq_text) F.seq_ptr = sptr.data() + J->base;
    if (s->P.output_format || s->P.extra_sam_fields) F.hgen = s->h_genome.data();
    auto worker = [&]() {
      std::vector<FHit> fh; std::vector<FHit*> p2;
      for (;;) {
        int c = next.fetch_add(1); if (c >= nchunks) break;
        std::string& o = J->outs[c]; o = gm_text_pool().take(); if (emit_sam) o.reserve((size_t)chunk * (read_len + 120));
        for (int rd = c * chunk; rd < std::min(n, (c + 1) * chunk); rd++) {
          const uint32_t cnt = J->hs->sel_cnt[rd], off = J->hs->sel_off[rd];
          const size_t before = o.size();
          int k = F.finalize_read(rd, cnt ? &J->hs->res[off] : nullptr, J->hs->ops, ops_stride, (int)cnt, o, fh, p2);
          if (per_read_bytes) per_read_bytes[J->base + rd] = (uint32_t)(o.size() - before);
          if (!p2.empty()) J->cm[c]++;
          J->cr[c] += k;
          if (!emit_sam) o.clear();
        }
      }
    };
    gm_run_on_threads(nthreads, worker);
    J->ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if (emit_sam) {
      size_t sz = 0; for (auto& o : J->outs) sz += o.size();
      std::unique_lock<std::mutex> lk(ob.m);
      ob.cv.wait(lk, [&] { return ob.turn == J->idx; });
      if (!ob.failed && ob.len + sz + 1 > ob.cap) {
        // first guess: this job's bytes per read for all reads, then geometric growth (large blocks move by remapping)
        size_t want = std::max(ob.len + sz + 1, ob.cap + ob.cap / 2);
        if (!ob.cap) want = std::max(want, (size_t)((double)sz / std::max(1, n) * 1.02 * n_reads) + 4096);
        // the buffer of the last call, if it was given back and is about large enough: the first guess is an estimate (+- 1 % from one call to the next), a parked buffer just
        // below it would be thrown away for a fresh one (300 MB of page faults, ~25 ms a call on the calling thread's path), and a buffer that turns out short grows below anyway
        size_t ccap = 0; char* np = ob.p ? nullptr : outcache_take(std::max(ob.len + sz + 1, want - want / 8), &ccap);
        if (np) want = ccap; else { if (!ob.p) want += want / 16; np = (char*)realloc(ob.p, want); }
        if (!np) ob.failed = true; else { ob.p = np; ob.cap = want; }
      }
      if (!ob.failed) {
        char* dst = ob.p + ob.len;
        std::vector<size_t> off(J->outs.size()); size_t a = 0; for (size_t c = 0; c < J->outs.size(); c++) { off[c] = a; a += J->outs[c].size(); }
        std::atomic<size_t> nextc(0);
        auto copier = [&]() { for (;;) { const size_t c = nextc.fetch_add(1); if (c >= J->outs.size()) break; memcpy(dst + off[c], J->outs[c].data(), J->outs[c].size()); gm_text_pool().give(std::move(J->outs[c])); J->outs[c] = std::string(); } };
        gm_run_on_threads(std::min(nthreads, 4), copier);
        ob.len += sz;
      }
      ob.turn = J->idx + 1;
      lk.unlock(); ob.cv.notify_all();
    }
    for (auto& o : J->outs) gm_text_pool().give(std::move(o));          // (what the copy above did not hand back already)
  };
  size_t joined = 0;
  // Two buffer sets: the front of sub-batch i + 1 (stream A) is queued before the host turns to the back of sub-batch i (stream B),
  // so K1 of the next sub-batch and pass 1 / pass 2 of this one share the CUs (see pipeline_front).  GM_OVERLAP=0: one set, in order.
  bool overlap = n_reads > D.eff_batch;
  if (const char* e = getenv("GM_OVERLAP")) overlap = overlap && atoi(e) != 0;
  auto same_caps = [&](const DevSet& a, const DevSet& b) { return a.cur_len == b.cur_len && a.scap == b.scap && a.scap2 == b.scap2 && a.hcap == b.hcap &&
                                                                  a.rcap_per_read == b.rcap_per_read && a.eff_batch == b.eff_batch && b.d_pmin == nullptr; };
  auto match_sets = [&](int from) -> int {                       // give the other set the capacities of set `from`
    DevSet& a = s->set[from]; DevSet& b = s->set[from ^ 1];
    if (same_caps(a, b)) return GM_OK;
    b.scap = a.scap; b.scap2 = a.scap2; b.hcap = a.hcap; b.rcap_per_read = a.rcap_per_read;
    return alloc_buffers(s, b, read_len);
  };
  if (overlap) { int rc = match_sets(0); if (rc) return rc; if (s->set[1].eff_batch != D.eff_batch) overlap = false; }
  auto join_all = [&]() { for (auto& j : jobs) if (j->th.joinable()) j->th.join(); };
  // copies of one sub-batch into set k; host memory goes through stream C so that the call never waits behind queued kernels
  auto queue_inputs = [&](int k, int base, int n) -> int {
    DevSet& S = s->set[k];
    if (!reads_host) {
      GM_HIP(hipMemcpyAsync(S.d_reads, (const uint32_t*)reads_dev + (size_t)base * read_words, (size_t)n * read_words * 4, hipMemcpyDeviceToDevice, s->stream));
      S.xover_on = false;
      return GM_OK;
    }
    hipStream_t c = s->stream_c;
    GM_HIP(hipMemcpyAsync(S.d_reads, reads_host + (size_t)base * read_words, (size_t)n * read_words * 4, hipMemcpyHostToDevice, c));
    if (initbp_host) GM_HIP(hipMemcpyAsync(S.d_initbp, initbp_host + base, (size_t)n, hipMemcpyHostToDevice, c));
    S.xover_on = false;
    if (initbp_host && quals) {                              // per-position crossover scores from the QVs, ref: gmapper.c:532-544
      xbuf.resize((size_t)n * read_len); qvbuf.resize((size_t)n * read_len);
      for (int i = 0; i < n; i++) {
        const char* q = qptr[base + i];
        for (int j = 0; j < read_len; j++) {
          const int qv = (int)q[j] - qual_delta;
          qvbuf[(size_t)i * read_len + j] = (uint8_t)std::min(250, std::max(0, qv));   // what post_sw's error-rate formula distinguishes (qv <= 0, qv >= 250, ref: util.h:285-293)
          const double pe = qv <= 0 ? .99999999 : (qv >= 250 ? 1E-25 : pow(10.0, -(double)qv / 10.0));   // pr_err_from_qv, ref: util.h:285-293
          int cx = (int)(s->score_alpha * log(pe / 3.0) / log(2.0));
          if (cx > -1) cx = -1; else if (cx < 2 * s->P.crossover_score) cx = 2 * s->P.crossover_score;
          xbuf[(size_t)i * read_len + j] = (int8_t)cx;
        }
      }
      GM_HIP(hipMemcpyAsync(S.d_xover, xbuf.data(), xbuf.size(), hipMemcpyHostToDevice, c));
      GM_HIP(hipMemcpyAsync(S.d_qv, qvbuf.data(), qvbuf.size(), hipMemcpyHostToDevice, c));
      GM_HIP(hipStreamSynchronize(c));                      // xbuf is reused by the next sub-batch
      S.xover_on = true;
    }
    GM_HIP(hipEventRecord(s->pev[k][8], c));
    GM_HIP(hipStreamWaitEvent(s->stream, s->pev[k][8], 0));
    return GM_OK;
  };
  // Sub-batch sizes: while the two halves overlap, the first sub-batch's front and the last one's back (and its host work) have nothing to
  // run beside, so the sizes ramp up from 8 192 at the start and halve towards the end (results do not depend on the split).
  int ramp_min = 8192; if (const char* e = gm_tune("GM_RAMP_MIN")) ramp_min = std::max(64, atoi(e));
  auto size_at = [&](int base, int eff) {
    const int R = n_reads - base;
    int n = std::min(eff, R);
    if (overlap && ramp_min < eff) {
      n = std::min(n, std::max(ramp_min, base + ramp_min));
      n = std::min(n, std::max(ramp_min, R / 2));
      if (R - n < ramp_min / 2) n = std::min(eff, R);
    }
    return n;
  };
  // GM_TIMELINE=1: where the calling thread's time goes, per sub-batch (stderr)
  const bool timeline = getenv("GM_TIMELINE") != nullptr;
  auto now_ms = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  const double tl0 = now_ms();
  if (timeline) fprintf(stderr, "[timeline] before the first sub-batch: %.2f ms waiting for the device's turn, %.2f ms set-up\n", tl_turn - tl_in, tl0 - tl_turn);
  // Sub-batch i lives in set i & 1.  Its front is queued as early as the set allows: the first two at once, the front of i + 2 from inside the back of i -- behind pass 2
  // and the copy of the counters, in front of the wait for the result copies (pipeline_back's after_pass2; the device orders it behind event pev[.][9]).  Short fronts
  // (the bucket kernel on a small genome: 6 ms a sub-batch) then follow each other without the gap of a host round trip.
  int q_base = 0, q_idx = 0;                                  // next sub-batch whose front is not queued yet: its first read, its number
  int fr_n[2] = {0, 0};                                       // reads of the sub-batch whose front is queued on set k
  auto queue_front = [&](bool behind_back) -> int {           // queues the front of sub-batch q_idx, if there is one
    if (q_base >= n_reads) return GM_OK;
    const int k = overlap ? (q_idx & 1) : 0;
    const int n = size_at(q_base, s->set[k].eff_batch);
    if (behind_back) { GM_HIP(hipStreamWaitEvent(s->stream, s->pev[k][9], 0)); GM_HIP(hipStreamWaitEvent(s->stream_c, s->pev[k][9], 0)); }
    int rc = queue_inputs(k, q_base, n); if (!rc) rc = pipeline_front(s, k, n, read_len);
    if (rc) return rc;
    fr_n[k] = n; q_base += n; q_idx++;
    return GM_OK;
  };
  const std::function<int()> ahead = [&]() -> int { return queue_front(true); };
  int base = 0, idx = 0;
  while (base < n_reads) {
    const int cur = overlap ? (idx & 1) : 0;
    int rc = GM_OK; float lk = 0;
    const double tl_a = now_ms();
    while (!rc && q_idx <= idx + (overlap ? 1 : 0) && q_base < n_reads) rc = queue_front(false);      // (the start of the call, and after a re-allocation)
    const int n = fr_n[cur];
    const bool have_next = q_idx > idx + 1;
    HostSlot& HS = s->slot[jobs.size() % 3];
    HS.want_reads = reads_host == nullptr;
    const double tl_b = now_ms();
    // (the front of i + 2 from inside this back only when the capacities are settled -- a re-allocation frees the set's buffers -- i.e. not for the first sub-batches)
    const bool go_ahead = overlap && idx >= 2;
    if (!rc) rc = pipeline_back(s, cur, HS, n, read_len, stats, &lk, have_next, go_ahead ? &ahead : nullptr);
    const double tl_c = now_ms();
    if (rc == 1) {                                            // capacities of set `cur` grew: same for the other set, then again from the front of this sub-batch
      GM_HIP(hipStreamSynchronize(s->stream));
      if (overlap) { rc = match_sets(cur); if (rc) { join_all(); return rc; } }
      q_base = base; q_idx = idx;
      continue;
    }
    if (rc) { (void)hipStreamSynchronize(s->stream); (void)hipStreamSynchronize(s->stream_b); join_all(); return rc; }
    s->last_lookup_ms += lk; s->last_lookup_launches++;
    std::unique_ptr<Job> J(new Job());
    J->base = base; J->n = n; J->idx = (int)jobs.size();
    J->hs = &HS;
    J->hreads = reads_host ? reads_host + (size_t)base * read_words : HS.reads;
    // at most two host jobs outstanding
    const double tl_d = now_ms();
    while (jobs.size() - joined >= 2) { jobs[joined]->th.join(); joined++; }
    const double tl_e = now_ms();
    Job* jp = J.get();
    J->th = std::thread(run_job, jp);
    jobs.push_back(std::move(J));
    if (timeline) fprintf(stderr, "[timeline] t %8.2f  n %7d  fronts %6.2f  back %6.2f  join-wait %6.2f  spawn %5.2f  (K1 %.2f ms, %u read-strands through the heavy tier)\n",
                          tl_a - tl0, n, tl_b - tl_a, tl_c - tl_b, tl_e - tl_d, now_ms() - tl_e, lk, s->h_pin[cur]);
    base += n; idx++;
  }
  const double tl_f = now_ms();
  for (auto& j : jobs) if (j->th.joinable()) j->th.join();
  if (timeline) { fprintf(stderr, "[timeline] t %8.2f  final join %6.2f; host jobs:", tl_f - tl0, now_ms() - tl_f); for (auto& j : jobs) fprintf(stderr, " %.1f", j->ms); fprintf(stderr, "\n"); }
  uint64_t matched = 0, records = 0;
  for (auto& j : jobs) {
    for (size_t c = 0; c < j->cm.size(); c++) { matched += j->cm[c]; records += j->cr[c]; }
    if (stats) stats->ms_host += j->ms;
  }
  if (stats) { stats->reads = n_reads; stats->reads_matched = matched; stats->sam_records = records; stats->post_sw_host_redo = post_redo.load(); }
  if (emit_sam && sam) {
    if (ob.failed) return GM_E_NOMEM;
    char* r = ob.p;                                            // (not shrunk: gm_free parks a large buffer for the next call)
    if (!r || ob.cap < ob.len + 1) { r = (char*)realloc(ob.p, ob.len + 1); if (!r) return GM_E_NOMEM; ob.cap = ob.len + 1; }
    ob.p = nullptr;
    outcache_track(r, ob.cap);
    r[ob.len] = 0; *sam = r; if (sam_len) *sam_len = ob.len;
  } else { if (sam) *sam = nullptr; if (sam_len) *sam_len = 0; }
  return GM_OK;
}

extern "C" int gm_map_reads(gm_session_t* s, int n_reads, int read_len, const uint32_t* reads_packed, const char* names,
                            char** sam, size_t* sam_len, gm_map_stats_t* stats) {
  return map_impl(s, n_reads, read_len, reads_packed, nullptr, names, 1, sam, sam_len, stats);
}
extern "C" int gm_map_reads_fastq(gm_session_t* s, int n_reads, int read_len, const uint32_t* reads_packed, const char* names, const char* quals, int qual_delta,
                                  char** sam, size_t* sam_len, gm_map_stats_t* stats) {
  if (!quals) { gm_set_error("gm_map_reads_fastq: no QUAL strings"); return GM_E_ARG; }
  if (s && s->P.colour_space) { gm_set_error("gm_map_reads_fastq: letter space only"); return GM_E_ARG; }
  return map_impl(s, n_reads, read_len, reads_packed, nullptr, names, 1, sam, sam_len, stats, nullptr, quals, qual_delta);
}
extern "C" int gm_map_reads_cs_fastq(gm_session_t* s, int n_reads, int n_colours, const uint32_t* colours_packed, const uint8_t* initbp, const char* names,
                                     const char* quals, int qual_delta, char** sam, size_t* sam_len, gm_map_stats_t* stats) {
  if (!colours_packed || !initbp || !quals) { gm_set_error("gm_map_reads_cs_fastq: bad arguments"); return GM_E_ARG; }
  if (s && 2 * s->P.crossover_score < -128) { gm_set_error("crossover score %d: per-position scores are kept in 8 bits", s->P.crossover_score); return GM_E_RANGE; }
  for (int i = 0; i < n_reads; i++) if (initbp[i] > 3) { gm_set_error("read %d: primer letter code %d", i, (int)initbp[i]); return GM_E_ARG; }
  return map_impl(s, n_reads, n_colours, colours_packed, nullptr, names, 1, sam, sam_len, stats, initbp, quals, qual_delta);
}
extern "C" int gm_map_reads_cs(gm_session_t* s, int n_reads, int n_colours, const uint32_t* colours_packed, const uint8_t* initbp, const char* names,
                               char** sam, size_t* sam_len, gm_map_stats_t* stats) {
  if (!colours_packed || !initbp) { gm_set_error("gm_map_reads_cs: bad arguments"); return GM_E_ARG; }
  for (int i = 0; i < n_reads; i++) if (initbp[i] > 3) { gm_set_error("read %d: primer letter code %d (the reference rejects such reads, fasta.c:636-645)", i, (int)initbp[i]); return GM_E_ARG; }
  return map_impl(s, n_reads, n_colours, colours_packed, nullptr, names, 1, sam, sam_len, stats, initbp);
}
extern "C" int gm_map_reads_device(gm_session_t* s, int n_reads, int read_len, const void* reads_dev, int emit_sam, char** sam, size_t* sam_len, gm_map_stats_t* stats) {
  return map_impl(s, n_reads, read_len, nullptr, reads_dev, nullptr, emit_sam, sam, sam_len, stats);
}
// ---- A22: text -> 4-bit codes (ref: common/fasta.c:609-673; tables :151-200, fasta.h:26-42) --------------------------------------------------
// Host code (no device needed).  Letter space: A C G T U M R W S Y K V H D B N = 0..15, X and '.' = 15, either case.  Colour space: the first
// character is the primer letter (A/C/G/T, either case; anything else: the reference drops the read), then colours 0-3, and 4 / N / n / . / X / x = 15.
extern "C" int gm_sequence_to_bitfield(int colour_space, const char* seq, int seq_len, uint32_t* words, int* initbp) {
  if (!seq || seq_len < 1 || !words) { gm_set_error("gm_sequence_to_bitfield: bad arguments"); return GM_E_ARG; }
  static const signed char LS[128] = {
    -1,-1,-1,-1,-1,-1,-1,-1,-1,-1,-1,-1,-1,-1,-1,-1,-1,-1,-1,-1,-1,-1,-1,-1,-1,-1,-1,-1,-1,-1,-1,-1, -1,-1,-1,-1,-1,-1,-1,-1,-1,-1,-1,-1,-1,-1,15,-1, -1,-1,-1,-1,-1,-1,-1,-1,-1,-1,-1,-1,-1,-1,-1,-1,
    -1, 0,14, 1,13,-1,-1, 2,12,-1,-1,10,-1, 5,15,-1, -1,-1, 6, 8, 3, 4,11, 7,15, 9,-1,-1,-1,-1,-1,-1,  -1, 0,14, 1,13,-1,-1, 2,12,-1,-1,10,-1, 5,15,-1, -1,-1, 6, 8, 3, 4,11, 7,15, 9,-1,-1,-1,-1,-1,-1};
  int i = 0, idx = 0;
  if (colour_space) {
    int b;
    switch (seq[0]) { case 'A': case 'a': b = 0; break; case 'C': case 'c': b = 1; break; case 'G': case 'g': b = 2; break; case 'T': case 't': b = 3; break;
                      default: gm_set_error("colour-space read does not start with a primer letter (ref: fasta.c:626-634)"); return GM_E_ARG; }
    if (initbp) *initbp = b;
    i = 1;
  }
  const int n = seq_len - i;
  memset(words, 0, (size_t)((n + 7) / 8) * 4);
  for (; i < seq_len; i++, idx++) {
    const unsigned char ch = (unsigned char)seq[i]; int a = -1;
    if (colour_space) { if (ch >= '0' && ch <= '3') a = ch - '0'; else if (ch == '4' || ch == 'N' || ch == 'n' || ch == '.' || ch == 'X' || ch == 'x') a = 15; }
    else if (ch < 128) a = LS[ch];
    if (a < 0) { gm_set_error("invalid character 0x%x in a read (did you mix up letter space and colour space? ref: fasta.c:639-650)", ch); return GM_E_ARG; }
    words[idx >> 3] |= (uint32_t)a << ((idx & 7) * 4);
  }
  return GM_OK;
}

// Text entry point: n reads of one length as lines ('\n' separated): letters, or in a colour-space session primer + colours.  The library packs them
// (gm_sequence_to_bitfield) and keeps the characters for what the reference prints from the file's text: SEQ of unaligned reads and of clipped ends,
// the CS:Z tag.  quals (optional): FASTQ QUAL lines and their offset.
extern "C" int gm_map_reads_text(gm_session_t* s, int n_reads, int read_len, const char* seqs, const char* names, const char* quals, int qual_delta,
                                 char** sam, size_t* sam_len, gm_map_stats_t* stats) {
  if (!s || !seqs || n_reads < 0 || read_len < 1) { gm_set_error("gm_map_reads_text: bad arguments"); return GM_E_ARG; }
  const int cs = s->P.colour_space ? 1 : 0, line = read_len + cs, rwords = (read_len + 7) / 8;
  std::vector<uint32_t> packed((size_t)n_reads * rwords); std::vector<uint8_t> ibp(cs ? n_reads : 0);
  if (gm_fixed_lines(seqs, (size_t)n_reads, (size_t)line)) {      // packing on the host threads (one thread took 90 ms per million 100-base reads)
    std::atomic<int> bad_rc(GM_OK); std::mutex em; std::string emsg;
    gm_parallel_for((size_t)n_reads, 8192, [&](size_t b0, size_t e0) {
      for (size_t i = b0; i < e0 && bad_rc.load(std::memory_order_relaxed) == GM_OK; i++) {
        int b = 0; int rc = gm_sequence_to_bitfield(cs, seqs + i * ((size_t)line + 1), line, packed.data() + i * rwords, &b);
        if (rc) { std::lock_guard<std::mutex> lk(em); if (bad_rc == GM_OK) { bad_rc = rc; emsg = gm_last_error(); } return; }
        if (cs) ibp[i] = (uint8_t)b;
      }
    });
    if (bad_rc != GM_OK) { gm_set_error("%s", emsg.c_str()); return bad_rc; }
  } else {
  const char* p = seqs;
  for (int i = 0; i < n_reads; i++) {
    const char* e = strchr(p, '\n'); if (!e) e = p + strlen(p);
    if ((int)(e - p) != line) { gm_set_error("read %d: %d characters, expected %d", i, (int)(e - p), line); return GM_E_ARG; }
    int b = 0; const int rc = gm_sequence_to_bitfield(cs, p, line, packed.data() + (size_t)i * rwords, &b); if (rc) return rc;
    if (cs) ibp[i] = (uint8_t)b;
    p = *e ? e + 1 : e;
  }
  }
  return map_impl(s, n_reads, read_len, packed.data(), nullptr, names, 1, sam, sam_len, stats, cs ? ibp.data() : nullptr, quals, qual_delta, seqs);
}

// (the file entry points -- the reads reader, the per-read preprocessing, chunked streaming -- live in gm_host_files.inc, included below)

extern "C" int gm_last_lookup_timing(gm_session_t* s, double* ms, uint64_t* alg_bytes, int* launches) {
  if (!s) return GM_E_ARG;
  if (ms) *ms = s->last_lookup_ms; if (alg_bytes) *alg_bytes = s->last_lookup_bytes; if (launches) *launches = s->last_lookup_launches;
  return GM_OK;
}

#include "gm_host_pairs.inc"
#include "gm_host_files.inc"
#include "gm_index_io.inc"

// stage dump for parity tests: hits selected by pass 1, in ext-heap array order (before pass 2 / reverse_hit)
extern "C" int gm_debug_tophits(gm_session_t* s, int n_reads, int read_len, const uint32_t* reads_packed, long long* rows, long cap, long* n_rows) {
  if (!s) return GM_E_ARG;
  GmDevTurn dev_turn(s);     // (the device's lookup scratch is shared by its sessions: one call at a time, like the mapping entries)
  GM_HIP(hipSetDevice(s->ix->device));
  DevSet& D = s->set[0];
  if (D.cur_len != read_len || (D.caps_pair_mode != 0 && D.caps_pair_mode != 4)) { choose_caps(s, D, read_len); D.caps_pair_mode = 0; int rc = alloc_buffers(s, D, read_len); if (rc) return rc; }
  if (n_reads > D.eff_batch) { gm_set_error("gm_debug_tophits: at most %d reads per call", D.eff_batch); return GM_E_ARG; }
  const int read_words = (read_len + 7) / 8;
  float lk; int rc;
  do {
    if (n_reads > D.eff_batch) { gm_set_error("gm_debug_tophits: at most %d reads per call", D.eff_batch); return GM_E_ARG; }
    GM_HIP(hipMemcpyAsync(D.d_reads, reads_packed, (size_t)n_reads * read_words * 4, hipMemcpyHostToDevice, s->stream));
    rc = run_device_pipeline(s, D, s->slot[0], n_reads, read_len, nullptr, &lk);
  } while (rc == 1);
  if (rc) return rc;
  std::vector<int32_t> sel((size_t)n_reads * GM_SEL_MAX); std::vector<GmHit> hits((size_t)n_reads * 2 * D.hcap);
  GM_HIP(hipMemcpy(sel.data(), D.d_sel, sel.size() * 4, hipMemcpyDeviceToHost));
  GM_HIP(hipMemcpy(hits.data(), D.d_hits, hits.size() * sizeof(GmHit), hipMemcpyDeviceToHost));
  long w = 0;
  for (int rd = 0; rd < n_reads; rd++)
    for (uint32_t k = 0; k < s->slot[0].sel_cnt[rd]; k++) {
      if (w >= cap) { *n_rows = w; return GM_OK; }
      const int id = sel[(size_t)rd * GM_SEL_MAX + k]; const int st = id >> 16, hi = id & 0xFFFF;
      const GmHit& h = hits[((size_t)rd * 2 + st) * D.hcap + hi];
      long long r[12] = {rd, st, h.cn, h.g_off, h.w_len, h.score_vector, h.pct_score_vector, h.matches, h.ax, h.ay, h.alen, h.awidth};
      memcpy(rows + w * 12, r, sizeof r); w++;
    }
  *n_rows = w;
  return GM_OK;
}

// ---- S1: vector SW on caller bitfields ----------------------------------------------------------
struct SwVecState { bool init = false, colours = false; GmScoreDev sc; int dblen = 0, qrlen = 0; uint64_t invocs = 0, cells = 0; double secs = 0; };
static thread_local SwVecState g_sv;

extern "C" int sw_vector_setup(int dblen, int qrlen, int a_gap_open, int a_gap_ext, int b_gap_open, int b_gap_ext,
                               int match, int mismatch, int use_colours, bool reset_stats) {
  if (match * qrlen >= 32768) { gm_set_error("match * qrlen >= 32768 (ref: sw-vector.c:393-398)"); return GM_E_RANGE; }
  if (gm_device_count() < 1) { gm_set_error("no HIP device"); return GM_E_NODEVICE; }
  gm_params_t P; gm_params_default(&P);
  P.match_score = match; P.mismatch_score = mismatch; P.a_gap_open_score = a_gap_open; P.a_gap_extend_score = a_gap_ext;
  P.b_gap_open_score = b_gap_open; P.b_gap_extend_score = b_gap_ext;
  g_sv.sc = make_score(P); g_sv.dblen = dblen; g_sv.qrlen = qrlen; g_sv.init = true; g_sv.colours = use_colours != 0;
  if (reset_stats) { g_sv.invocs = g_sv.cells = 0; g_sv.secs = 0; }
  return 0;
}
extern "C" int sw_vector_cleanup(void) { g_sv.init = false; return 0; }
extern "C" void sw_vector_stats(uint64_t* invocs, uint64_t* cells, double* secs) {
  if (invocs) *invocs = g_sv.invocs; if (cells) *cells = g_sv.cells; if (secs) *secs = g_sv.secs;
}

static int sw_vector_batch_impl(int n, const uint32_t* genome, uint64_t genome_words, const int64_t* g_off, const int* glen,
                                const uint32_t* reads, int read_words, const int* rlen, int* scores, int early_thr, uint8_t* stopped) {
  if (!g_sv.init) { gm_set_error("sw_vector called before sw_vector_setup"); return GM_E_NOTSETUP; }
  if (n <= 0) return GM_OK;
  int max_g = 0, max_r = 0;
  for (int i = 0; i < n; i++) { max_g = std::max(max_g, glen[i]); max_r = std::max(max_r, rlen[i]); if (glen[i] < 1 || rlen[i] < 1) return GM_E_ARG; }
  if (max_g > g_sv.dblen || max_r > g_sv.qrlen) { gm_set_error("window/read longer than sw_vector_setup sizes"); return GM_E_ARG; }
  uint32_t *dg = nullptr, *dr = nullptr; long long* dgo = nullptr; int *dgl = nullptr, *drl = nullptr, *ds = nullptr;
  GM_HIP(hipMalloc(&dg, (genome_words + 8) * 4)); GM_HIP(hipMalloc(&dr, (size_t)n * read_words * 4 + 32));
  GM_HIP(hipMalloc(&dgo, (size_t)n * 8)); GM_HIP(hipMalloc(&dgl, (size_t)n * 4)); GM_HIP(hipMalloc(&drl, (size_t)n * 4)); GM_HIP(hipMalloc(&ds, (size_t)n * 4));
  GM_HIP(hipMemset(dg, 0, (genome_words + 8) * 4));
  GM_HIP(hipMemcpy(dg, genome, genome_words * 4, hipMemcpyHostToDevice));
  GM_HIP(hipMemcpy(dr, reads, (size_t)n * read_words * 4, hipMemcpyHostToDevice));
  GM_HIP(hipMemcpy(dgo, g_off, (size_t)n * 8, hipMemcpyHostToDevice));
  GM_HIP(hipMemcpy(dgl, glen, (size_t)n * 4, hipMemcpyHostToDevice));
  GM_HIP(hipMemcpy(drl, rlen, (size_t)n * 4, hipMemcpyHostToDevice));
  uint8_t* dst = nullptr;
  if (stopped) GM_HIP(hipMalloc(&dst, (size_t)n));
  // (the early stop needs a scheme in which a cell gains at most `match` and gaps cost: otherwise every window runs to its end)
  const GmScoreDev& sv = g_sv.sc;
  if (!(sv.match > 0 && sv.mismatch <= sv.match && sv.a_go >= 0 && sv.a_ge >= 0 && sv.b_go >= 0 && sv.b_ge >= 0 && sv.match * (2 * 128 + 2) < 32000)) early_thr = 0;
  int rc = gm_launch_sw_vector_batch(g_sv.sc, n, dg, dgo, dgl, dr, read_words, drl, max_g, max_r, ds, 0, early_thr, dst);
  if (rc == GM_OK) {
    GM_HIP(hipDeviceSynchronize()); GM_HIP(hipMemcpy(scores, ds, (size_t)n * 4, hipMemcpyDeviceToHost));
    if (stopped) GM_HIP(hipMemcpy(stopped, dst, (size_t)n, hipMemcpyDeviceToHost));
  }
  (void)hipFree(dg); (void)hipFree(dr); (void)hipFree(dgo); (void)hipFree(dgl); (void)hipFree(drl); (void)hipFree(ds); if (dst) (void)hipFree(dst);
  for (int i = 0; i < n; i++) { g_sv.invocs++; g_sv.cells += (uint64_t)glen[i] * rlen[i]; }
  return rc;
}
extern "C" int gm_sw_vector_batch(int n, const uint32_t* genome, uint64_t genome_words, const int64_t* g_off, const int* glen,
                                  const uint32_t* reads, int read_words, const int* rlen, int* scores) {
  return sw_vector_batch_impl(n, genome, genome_words, g_off, glen, reads, read_words, rlen, scores, 0, nullptr);
}
extern "C" int gm_sw_vector_batch_bounded(int n, const uint32_t* genome, uint64_t genome_words, const int64_t* g_off, const int* glen,
                                          const uint32_t* reads, int read_words, const int* rlen, int threshold, int* scores, uint8_t* stopped) {
  if (threshold <= 0 || !stopped) { gm_set_error("gm_sw_vector_batch_bounded: threshold > 0 and a stopped[] array are required"); return GM_E_ARG; }
  return sw_vector_batch_impl(n, genome, genome_words, g_off, glen, reads, read_words, rlen, scores, threshold, stopped);
}

extern "C" int sw_vector(uint32_t* genome, int goff, int glen, uint32_t* read, int rlen, uint32_t* genome_ls, int initbp, bool is_rna) {
  if (!g_sv.init) abort();   // ref: sw-vector.c:462-463
  // is_rna only matters to the colour-space first-colour row (lstocs(genome_ls[j], initbp, is_rna), ref: sw-vector.c:129,289): it rides in bit 8 of the primer word
  if (is_rna && genome_ls) initbp |= GM_SEAM_RNA;
  int64_t go = goff; int score = 0;
  const uint64_t gw = ((uint64_t)goff + glen + 7) / 8;
  if (g_sv.colours) {        // colour space: genome = colours, genome_ls = letters of the same contig (ref: sw-vector.c:476-479)
    if (!genome_ls) { gm_set_error("sw_vector: colour space needs genome_ls"); return GM_E_ARG; }
    int rc = gm_sw_vector_batch_cs(1, genome, genome_ls, gw, &go, &glen, read, (rlen + 7) / 8, &rlen, &initbp, &score);
    return rc == GM_OK ? score : rc;
  }
  int rc = gm_sw_vector_batch(1, genome, gw, &go, &glen, read, (rlen + 7) / 8, &rlen, &score);
  return rc == GM_OK ? score : rc;
}

// ---- S1, ungapped: sw_gapless on caller bitfields (ref: common/sw-gapless.h:11-14, sw-gapless.c:29-117) -----------------------------
// What f1_setup / f1_run call when gapless_sw is set (ref: f1-wrapper.h:66-68,122-125).  State per calling thread, like the reference's threadprivate statics.
struct SwGaplessState { bool init = false; int match = 0, mismatch = 0; uint64_t invocs = 0, cells = 0, ticks = 0; };
static thread_local SwGaplessState g_sg;
extern "C" int sw_gapless_setup(int match, int mismatch, bool reset_stats) {
  if (gm_device_count() < 1) { gm_set_error("no HIP device"); return GM_E_NODEVICE; }
  g_sg.match = match; g_sg.mismatch = mismatch; g_sg.init = true;
  if (reset_stats) g_sg.invocs = g_sg.cells = g_sg.ticks = 0;
  return 0;
}
extern "C" void sw_gapless_stats(uint64_t* invocs, uint64_t* cells, uint64_t* ticks) {      // (ticks: nanoseconds spent inside sw_gapless on this thread; the reference counts rdtsc ticks)
  if (invocs) *invocs = g_sg.invocs; if (cells) *cells = g_sg.cells; if (ticks) *ticks = g_sg.ticks;
}
// n independent calls; call i's genome bitfield starts at word genome_woff[i] of `genome` (and of `genome_ls`, colour space only: then `genome` holds colours,
// `genome_ls` the letters of the same contig and initbp[i] the read's primer letter) and holds glen[i] positions
// device buffers released when the holder goes out of scope
struct GmDevBufs { std::vector<void*> p; ~GmDevBufs() { for (void* q : p) if (q) (void)hipFree(q); }
                   template <class T> hipError_t get(T** out, size_t bytes) { void* q = nullptr; const hipError_t e = hipMalloc(&q, bytes); if (e == hipSuccess) { p.push_back(q); *out = (T*)q; } return e; } };
extern "C" int gm_sw_gapless_batch(int n, const uint32_t* genome, const uint32_t* genome_ls, uint64_t genome_words, const int64_t* genome_woff, const int* glen,
                                   const uint32_t* reads, int read_words, const int* rlen, const int* g_idx, const int* r_idx, const int* initbp, int* scores) {
  if (!g_sg.init) { gm_set_error("sw_gapless called before sw_gapless_setup"); return GM_E_NOTSETUP; }
  if (n <= 0) return GM_OK;
  if (genome_ls && !initbp) { gm_set_error("gm_sw_gapless_batch: colour space needs initbp"); return GM_E_ARG; }
  int max_r = 0;
  for (int i = 0; i < n; i++) {
    if (glen[i] < 1 || rlen[i] < 1 || g_idx[i] < 0 || r_idx[i] < 0 || g_idx[i] >= glen[i] || r_idx[i] >= rlen[i] || genome_woff[i] < 0 ||
        (uint64_t)genome_woff[i] + ((uint64_t)glen[i] + 7) / 8 > genome_words || (rlen[i] + 7) / 8 > read_words) { gm_set_error("gm_sw_gapless_batch: call %d out of range", i); return GM_E_ARG; }
    max_r = std::max(max_r, rlen[i]);
  }
  const auto t0 = std::chrono::steady_clock::now();
  GmDevBufs bufs;      // (owns every device buffer below: a failing call in the middle returns without leaking the ones allocated before it)
  uint32_t *dg = nullptr, *dgl = nullptr, *dr = nullptr; long long* dwo = nullptr; int *dn = nullptr, *drl = nullptr, *dgi = nullptr, *dri = nullptr, *dib = nullptr, *ds = nullptr;
  GM_HIP(bufs.get(&dg, (genome_words + 8) * 4)); GM_HIP(hipMemset(dg, 0, (genome_words + 8) * 4)); GM_HIP(hipMemcpy(dg, genome, genome_words * 4, hipMemcpyHostToDevice));
  if (genome_ls) { GM_HIP(bufs.get(&dgl, (genome_words + 8) * 4)); GM_HIP(hipMemset(dgl, 0, (genome_words + 8) * 4)); GM_HIP(hipMemcpy(dgl, genome_ls, genome_words * 4, hipMemcpyHostToDevice)); }
  GM_HIP(bufs.get(&dr, (size_t)n * read_words * 4 + 32)); GM_HIP(hipMemcpy(dr, reads, (size_t)n * read_words * 4, hipMemcpyHostToDevice));
  GM_HIP(bufs.get(&dwo, (size_t)n * 8)); GM_HIP(hipMemcpy(dwo, genome_woff, (size_t)n * 8, hipMemcpyHostToDevice));
  int** const dst[5] = {&dn, &drl, &dgi, &dri, &dib}; const int* const src[5] = {glen, rlen, g_idx, r_idx, initbp};
  for (int k = 0; k < 5; k++) { if (!src[k]) continue; GM_HIP(bufs.get(dst[k], (size_t)n * 4)); GM_HIP(hipMemcpy(*dst[k], src[k], (size_t)n * 4, hipMemcpyHostToDevice)); }
  GM_HIP(bufs.get(&ds, (size_t)n * 4));
  int rc = gm_launch_sw_gapless_batch(n, g_sg.match, g_sg.mismatch, dg, dgl, dwo, dn, dr, read_words, drl, dgi, dri, dib, max_r, ds, 0);
  if (rc == GM_OK) { GM_HIP(hipDeviceSynchronize()); GM_HIP(hipMemcpy(scores, ds, (size_t)n * 4, hipMemcpyDeviceToHost)); }
  if (rc == GM_OK) {                                                                             // (a failed launch scored nothing: it does not count)
    for (int i = 0; i < n; i++) { g_sg.invocs++; g_sg.cells += (uint64_t)rlen[i]; }              // ref: sw-gapless.c:111 (cells += rlen)
    g_sg.ticks += (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
  }
  return rc;
}
extern "C" int sw_gapless(uint32_t* genome, int glen, uint32_t* read, int rlen, int g_idx, int r_idx, uint32_t* genome_ls, int init_bp, bool is_rna) {
  if (!g_sg.init) abort();   // ref: sw-gapless.c:66-67
  if (is_rna && genome_ls) init_bp |= GM_SEAM_RNA;      // lstocs(letter, init_bp, is_rna), ref: sw-gapless.c:84
  int64_t wo = 0; int score = 0;
  int rc = gm_sw_gapless_batch(1, genome, genome_ls, ((uint64_t)glen + 7) / 8, &wo, &glen, read, (rlen + 7) / 8, &rlen, &g_idx, &r_idx, genome_ls ? &init_bp : nullptr, &score);
  return rc == GM_OK ? score : rc;
}

// ---- S2: full SW on caller bitfields ------------------------------------------------------------
struct SwFullState { bool init = false; GmScoreDev sc; int dblen = 0, qrlen = 0; uint64_t invocs = 0, cells = 0; double secs = 0; };
static thread_local SwFullState g_sf;
static const char LSTRANS[17] = "ACGTUMRWSYKVHDBN";   // base_translate, ref: common/fasta.c:689-690

extern "C" int sw_full_ls_setup(int dblen, int qrlen, int a_gap_open, int a_gap_ext, int b_gap_open, int b_gap_ext,
                                int match, int mismatch, bool reset_stats, int anchor_width) {
  if (gm_device_count() < 1) { gm_set_error("no HIP device"); return GM_E_NODEVICE; }
  gm_params_t P; gm_params_default(&P);
  P.match_score = match; P.mismatch_score = mismatch; P.a_gap_open_score = a_gap_open; P.a_gap_extend_score = a_gap_ext;
  P.b_gap_open_score = b_gap_open; P.b_gap_extend_score = b_gap_ext; P.anchor_width = anchor_width;
  g_sf.sc = make_score(P); g_sf.dblen = dblen; g_sf.qrlen = qrlen; g_sf.init = true;
  if (reset_stats) { g_sf.invocs = g_sf.cells = 0; g_sf.secs = 0; }
  return 0;
}
extern "C" int sw_full_ls_cleanup(void) { g_sf.init = false; return 0; }
// invocations, cells (window x read, the upper bound of the band the reference counts, ref: sw-full-ls.c:237) and seconds spent inside sw_full_ls on this thread
extern "C" void sw_full_ls_stats(uint64_t* invocs, uint64_t* cells, double* secs) {
  if (invocs) *invocs = g_sf.invocs; if (cells) *cells = g_sf.cells; if (secs) *secs = g_sf.secs;
}
struct SeamTimer { double* acc; std::chrono::steady_clock::time_point t0; explicit SeamTimer(double* a) : acc(a), t0(std::chrono::steady_clock::now()) {}
                   ~SeamTimer() { *acc += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); } };

extern "C" void sw_full_ls(uint32_t* genome, int goff, int glen, uint32_t* read, int rlen, int threshscore, int maxscore,
                           struct gm_sw_full_results* sfr, bool revcmpl, struct gm_anchor* anchors, int anchors_cnt, int local_alignment) {
  if (!g_sf.init) abort();   // ref: sw-full-ls.c:649-650
  SeamTimer tm(&g_sf.secs); g_sf.invocs++; g_sf.cells += (uint64_t)std::max(glen, 0) * (uint64_t)std::max(rlen, 0);
  if ((anchors != nullptr && anchors_cnt != 1) || glen > g_sf.dblen || rlen > g_sf.qrlen || glen < 1 || rlen < 1) {
    gm_set_error("sw_full_ls: one anchor box (gmapper's call, ref: mapping.c:391-394) or none (the threshold band) is implemented");
    sfr->score = 0; sfr->dbalign = strdup(""); sfr->qralign = strdup(""); return;
  }
  const uint64_t gw = ((uint64_t)goff + glen + 7) / 8 + 8; const int rwords = (rlen + 7) / 8 + 1;
  const int ops_cap = glen + rlen + 8;
  uint32_t *dg = nullptr, *dr = nullptr; uint8_t *dback = nullptr, *dops = nullptr; int* dout = nullptr;
  bool ok = hipMalloc(&dg, gw * 4) == hipSuccess && hipMalloc(&dr, (size_t)rwords * 4) == hipSuccess &&
            hipMalloc(&dback, (size_t)glen * rlen + 256) == hipSuccess && hipMalloc(&dops, ops_cap) == hipSuccess && hipMalloc(&dout, 16 * 4) == hipSuccess;
  int out[16] = {0}; std::vector<uint8_t> ops(ops_cap);
  if (ok) {
    ok = hipMemset(dg, 0, gw * 4) == hipSuccess && hipMemcpy(dg, genome, (gw - 8) * 4, hipMemcpyHostToDevice) == hipSuccess &&
         hipMemcpy(dr, read, (size_t)(rwords - 1) * 4, hipMemcpyHostToDevice) == hipSuccess &&
         gm_launch_sw_full_single(g_sf.sc, dg, goff, glen, dr, rlen, anchors ? anchors[0].x : 0, anchors ? anchors[0].y : 0, anchors ? anchors[0].length : 1,
                                  anchors ? anchors[0].width : 1, revcmpl ? 1 : 0, dback, dout, dops, ops_cap, 0, anchors ? 1 : 0, threshscore, maxscore,
                                  local_alignment ? 1 : 0) == GM_OK &&
         hipDeviceSynchronize() == hipSuccess && hipMemcpy(out, dout, sizeof out, hipMemcpyDeviceToHost) == hipSuccess &&
         hipMemcpy(ops.data(), dops, ops_cap, hipMemcpyDeviceToHost) == hipSuccess;
  }
  (void)hipFree(dg); (void)hipFree(dr); (void)hipFree(dback); (void)hipFree(dops); (void)hipFree(dout);
  if (!ok) { gm_set_error("sw_full_ls: HIP failure"); sfr->score = 0; sfr->dbalign = strdup(""); sfr->qralign = strdup(""); return; }
  sfr->score = out[0];
  std::string db, qr;
  if (out[0] > 0) {
    sfr->read_start = out[1]; sfr->rmapped = out[2]; sfr->genome_start = out[3]; sfr->gmapped = out[4];
    sfr->matches += out[5]; sfr->mismatches += out[6]; sfr->insertions += out[7]; sfr->deletions += out[8];
    int pi = out[1], pj = out[3];   // pretty_print, ref: sw-full-ls.c:524-560 (genome_start already includes goff)
    auto nib = [](const uint32_t* b, long long i) { return (int)((b[i / 8] >> (4 * (i % 8))) & 0xf); };
    for (int k = 0; k < out[9]; k++) {
      if (ops[k] == 'D') { db.push_back('-'); qr.push_back(LSTRANS[nib(read, pi++)]); }
      else if (ops[k] == 'I') { db.push_back(LSTRANS[nib(genome, pj++)]); qr.push_back('-'); }
      else { db.push_back(LSTRANS[nib(genome, pj++)]); qr.push_back(LSTRANS[nib(read, pi++)]); }
    }
  } else {   // the reference backtraces stale scratch here; every caller discards it (score 0 < threshold)
    sfr->rmapped = 1; sfr->gmapped = 1; sfr->genome_start = goff;
  }
  sfr->dbalign = strdup(db.c_str()); sfr->qralign = strdup(qr.c_str());
}

extern "C" int gm_sw_vector_batch_cs(int n, const uint32_t* genome_cs, const uint32_t* genome_ls, uint64_t genome_words, const int64_t* g_off,
                                     const int* glen, const uint32_t* reads, int read_words, const int* rlen, const int* initbp, int* scores) {
  if (!g_sv.init) { gm_set_error("sw_vector called before sw_vector_setup"); return GM_E_NOTSETUP; }
  if (n <= 0) return GM_OK;
  int max_g = 0, max_r = 0;
  for (int i = 0; i < n; i++) { max_g = std::max(max_g, glen[i]); max_r = std::max(max_r, rlen[i]); if (glen[i] < 1 || rlen[i] < 1 || initbp[i] < 0 || (initbp[i] & ~GM_SEAM_RNA) > 3) return GM_E_ARG; }
  if (max_g > g_sv.dblen || max_r > g_sv.qrlen) { gm_set_error("window/read longer than sw_vector_setup sizes"); return GM_E_ARG; }
  uint32_t *dgc = nullptr, *dgl = nullptr, *dr = nullptr; long long* dgo = nullptr; int *dgn = nullptr, *drl = nullptr, *dib = nullptr, *ds = nullptr;
  GM_HIP(hipMalloc(&dgc, (genome_words + 8) * 4)); GM_HIP(hipMalloc(&dgl, (genome_words + 8) * 4)); GM_HIP(hipMalloc(&dr, (size_t)n * read_words * 4 + 32));
  GM_HIP(hipMalloc(&dgo, (size_t)n * 8)); GM_HIP(hipMalloc(&dgn, (size_t)n * 4)); GM_HIP(hipMalloc(&drl, (size_t)n * 4)); GM_HIP(hipMalloc(&dib, (size_t)n * 4)); GM_HIP(hipMalloc(&ds, (size_t)n * 4));
  GM_HIP(hipMemset(dgc, 0, (genome_words + 8) * 4)); GM_HIP(hipMemset(dgl, 0, (genome_words + 8) * 4));
  GM_HIP(hipMemcpy(dgc, genome_cs, genome_words * 4, hipMemcpyHostToDevice)); GM_HIP(hipMemcpy(dgl, genome_ls, genome_words * 4, hipMemcpyHostToDevice));
  GM_HIP(hipMemcpy(dr, reads, (size_t)n * read_words * 4, hipMemcpyHostToDevice));
  GM_HIP(hipMemcpy(dgo, g_off, (size_t)n * 8, hipMemcpyHostToDevice)); GM_HIP(hipMemcpy(dgn, glen, (size_t)n * 4, hipMemcpyHostToDevice));
  GM_HIP(hipMemcpy(drl, rlen, (size_t)n * 4, hipMemcpyHostToDevice)); GM_HIP(hipMemcpy(dib, initbp, (size_t)n * 4, hipMemcpyHostToDevice));
  int rc = gm_launch_sw_vector_batch_cs(g_sv.sc, n, dgc, dgl, dgo, dgn, dr, read_words, drl, dib, max_g, max_r, ds, 0);
  if (rc == GM_OK) { GM_HIP(hipDeviceSynchronize()); GM_HIP(hipMemcpy(scores, ds, (size_t)n * 4, hipMemcpyDeviceToHost)); }
  (void)hipFree(dgc); (void)hipFree(dgl); (void)hipFree(dr); (void)hipFree(dgo); (void)hipFree(dgn); (void)hipFree(drl); (void)hipFree(dib); (void)hipFree(ds);
  for (int i = 0; i < n; i++) { g_sv.invocs++; g_sv.cells += (uint64_t)glen[i] * rlen[i]; }
  return rc;
}

// ---- S2 in colour space: sw_full_cs on caller bitfields (ref: common/sw-full-cs.c:1084-1236) --------------
struct SwFullCsState { bool init = false; int p[9]; int dblen = 0, qrlen = 0; uint64_t invocs = 0, cells = 0; double secs = 0; };
static thread_local SwFullCsState g_sc;
extern "C" int sw_full_cs_setup(int dblen, int qrlen, int a_gap_open, int a_gap_ext, int b_gap_open, int b_gap_ext,
                                int match, int mismatch, int global_xover_penalty, bool reset_stats, int anchor_width, int indel_taboo_len) {
  if (reset_stats) { g_sc.invocs = g_sc.cells = 0; g_sc.secs = 0; }
  if (gm_device_count() < 1) { gm_set_error("no HIP device"); return GM_E_NODEVICE; }
  const int p[9] = {match, mismatch, global_xover_penalty, -a_gap_open, -a_gap_ext, -b_gap_open, -b_gap_ext, anchor_width, indel_taboo_len};
  memcpy(g_sc.p, p, sizeof p); g_sc.dblen = dblen; g_sc.qrlen = qrlen; g_sc.init = true;
  return 0;
}
extern "C" int sw_full_cs_cleanup(void) { g_sc.init = false; return 0; }
extern "C" void sw_full_cs_stats(uint64_t* invocs, uint64_t* cells, double* secs) {   // ref: sw-full-cs.c:1127-1140 (cells: window x read x 4 layers here)
  if (invocs) *invocs = g_sc.invocs; if (cells) *cells = g_sc.cells; if (secs) *secs = g_sc.secs;
}

extern "C" void sw_full_cs(uint32_t* genome_ls, int goff, int glen, uint32_t* read, int rlen, int initbp, int threshscore,
                           struct gm_sw_full_results* sfr, bool revcmpl, bool is_rna, struct gm_anchor* anchors, int anchors_cnt,
                           int local_alignment, int* crossover_score) {
  if (!g_sc.init) abort();   // ref: sw-full-cs.c:1155-1156
  SeamTimer tm(&g_sc.secs); g_sc.invocs++; g_sc.cells += 4ull * (uint64_t)std::max(glen, 0) * (uint64_t)std::max(rlen, 0);
  // A refusal must not read as "no alignment" (score 0 is what a window below the threshold returns): the reason goes to stderr as well as to gm_last_error().
  auto fail = [&](const char* why) { gm_set_error("sw_full_cs: %s", why); fprintf(stderr, "gmapper_hip: sw_full_cs refused: %s\n", why); sfr->score = 0; sfr->dbalign = nullptr; sfr->qralign = nullptr; };
  if (anchors == nullptr || anchors_cnt != 1 || glen > g_sc.dblen || rlen > g_sc.qrlen || glen < 1 || rlen < 1 ||
      initbp < 0 || initbp > 3 || g_sc.p[7] < 0) {
    fail("only one anchor box (gmapper's call, ref: mapping.c:375-379) is implemented"); return;
  }
  // crossover_score: one score per read position (from the QVs, ref: gmapper.c:532-544 -- clamped there to [2 * global, -1]); the kernels keep a row of them in 8 bits
  std::vector<int8_t> xrow;
  if (crossover_score) {
    xrow.resize((size_t)rlen + 16, 0);
    for (int i = 0; i < rlen; i++) {
      if (crossover_score[i] < -128 || crossover_score[i] > 127) { fail("a per-position crossover score outside [-128, 127] (the device keeps them in 8 bits)"); return; }
      xrow[i] = (int8_t)crossover_score[i];
    }
  }
  const uint64_t gw = ((uint64_t)goff + glen + 7) / 8 + 8; const int rwords = (rlen + 7) / 8 + 1;
  const int ops_cap = glen + rlen + 8;
  const size_t back_bytes = ((size_t)glen * rlen * 3 + 64) * 4;
  uint32_t *dg = nullptr, *dr = nullptr, *dback = nullptr; uint8_t* dops = nullptr; int* dout = nullptr; int8_t* dx = nullptr;
  bool ok = hipMalloc(&dg, gw * 4) == hipSuccess && hipMalloc(&dr, (size_t)rwords * 4) == hipSuccess && hipMalloc(&dback, back_bytes) == hipSuccess &&
            hipMalloc(&dops, ops_cap) == hipSuccess && hipMalloc(&dout, 16 * 4) == hipSuccess && (xrow.empty() || hipMalloc(&dx, xrow.size()) == hipSuccess);
  int out[16] = {0}; std::vector<uint8_t> ops(ops_cap);
  if (ok) {
    ok = hipMemset(dg, 0, gw * 4) == hipSuccess && hipMemcpy(dg, genome_ls, (gw - 8) * 4, hipMemcpyHostToDevice) == hipSuccess &&
         hipMemset(dr, 0, (size_t)rwords * 4) == hipSuccess && hipMemcpy(dr, read, (size_t)(rwords - 1) * 4, hipMemcpyHostToDevice) == hipSuccess &&
         (!dx || hipMemcpy(dx, xrow.data(), xrow.size(), hipMemcpyHostToDevice) == hipSuccess) &&
         hipMemset(dback, 0, back_bytes) == hipSuccess &&                                     // out-of-band cells: back == 0 (ref: init_cell)
         gm_launch_sw_full_cs_single(g_sc.p, dg, goff, glen, dr, rlen, initbp | (is_rna ? GM_SEAM_RNA : 0), threshscore, anchors[0].x, anchors[0].y, anchors[0].length, anchors[0].width,
                                     revcmpl ? 1 : 0, dback, dout, dops, ops_cap, 0, local_alignment ? 1 : 0, dx) == GM_OK &&
         hipDeviceSynchronize() == hipSuccess && hipMemcpy(out, dout, 12 * 4, hipMemcpyDeviceToHost) == hipSuccess &&
         hipMemcpy(ops.data(), dops, ops_cap, hipMemcpyDeviceToHost) == hipSuccess;
  }
  (void)hipFree(dg); (void)hipFree(dr); (void)hipFree(dback); (void)hipFree(dops); (void)hipFree(dout); (void)hipFree(dx);
  if (!ok) { fail("HIP failure"); return; }
  sfr->score = out[0];
  if (out[0] <= 0) { sfr->score = 0; sfr->dbalign = nullptr; sfr->qralign = nullptr; return; }          // below threshold: no strings (ref :1224-1226)
  sfr->read_start = out[1]; sfr->rmapped = out[2]; sfr->genome_start = out[3]; sfr->gmapped = out[4];
  sfr->matches += out[5]; sfr->mismatches += out[6]; sfr->insertions += out[7]; sfr->deletions += out[8]; sfr->crossovers += out[9];
  // pretty_print, ref :945-1060: the four translations of the colour read, lower case on crossovers, N in the read shows the genome letter
  auto nib = [](const uint32_t* b, long long i) { return (int)((b[i / 8] >> (4 * (i % 8))) & 0xf); };
  std::vector<uint8_t> qr[4];
  for (int k = 0; k < 4; k++) {
    qr[k].resize(rlen); int letter = (k + initbp) % 4;
    for (int j = 0; j < rlen; j++) {
      const int base = nib(read, j);
      if (base == 15) { qr[k][j] = 15; letter = (k + initbp) % 4; }
      else {                                                   // cstols(letter, base, is_rna), ref: util.h:157-180
        const int lt = (is_rna && letter == 4) ? 3 : letter;
        int l2 = (lt % 2 == 0) ? ((4 + lt + base) % 4) : ((4 + lt - base) % 4); if (is_rna && l2 == 3) l2 = 4;
        qr[k][j] = (uint8_t)((letter == 15 || base > 3) ? 15 : l2); letter = qr[k][j];
      }
    }
  }
  std::string db, q;
  int pi = out[1], pj = out[3];
  for (int t = 0; t < std::min(out[10], ops_cap); t++) {
    const int type = ops[t] & 0x0f; const bool xov = (ops[t] & 0x80) != 0;
    if (type == 1) { db.push_back(LSTRANS[nib(genome_ls, pj++)]); q.push_back('-'); continue; }
    const bool del = type >= 2 && type <= 5;
    const int lay = del ? type - 2 : type - 6;
    char c = LSTRANS[qr[lay][pi++]];
    if (xov) c = (char)tolower((int)c);
    if (del) { db.push_back('-'); q.push_back(c); }
    else { const char d = LSTRANS[nib(genome_ls, pj++)]; if (c == 'n' || c == 'N') c = xov ? (char)tolower((int)d) : d; db.push_back(d); q.push_back(c); }
  }
  sfr->dbalign = strdup(db.c_str()); sfr->qralign = strdup(q.c_str());
}


// ---- S3: post_sw on a caller's sw_full_results (ref: common/sw-post.c:364-758, sw-post.h:6-9) ------------------------------------------
// The colour-space posterior of one alignment, host doubles through libm in the reference's operation order (cs_post_sw above, the routine the
// read pipeline uses).  State per calling thread, like the reference's threadprivate statics.
struct PostSwState { bool init = false; CsPostConsts K; bool use_read_qvs = false; int qual_delta = 33, max_len = 0; uint64_t invocs = 0, cells = 0; double secs = 0; };
static thread_local PostSwState g_ps;
extern "C" int post_sw_setup(int max_len, double pr_snp, double pr_xover, double pr_del_open, double pr_del_extend, double pr_ins_open, double pr_ins_extend,
                             bool use_read_qvs, bool use_sanger_qvs, int qual_vector_offset, int qual_delta, bool reset_stats) {
  g_ps.K = cs_post_consts_from(pr_snp, pr_xover, pr_del_open, pr_del_extend, pr_ins_open, pr_ins_extend, use_sanger_qvs, use_read_qvs ? qual_vector_offset : 0);
  g_ps.use_read_qvs = use_read_qvs; g_ps.qual_delta = qual_delta; g_ps.max_len = max_len; g_ps.init = true;
  if (reset_stats) { g_ps.invocs = g_ps.cells = 0; g_ps.secs = 0; }
  return 1;                                                  // the reference returns 1 (sw-post.c:441)
}
extern "C" int post_sw_cleanup(void) { g_ps.init = false; return 1; }
extern "C" int post_sw_stats(uint64_t* invocs, uint64_t* cells, double* secs) {
  if (invocs) *invocs = g_ps.invocs; if (cells) *cells = g_ps.cells; if (secs) *secs = g_ps.secs;
  return 1;
}
// read: the colour read as the reference's 4-bit bitfield; qual: its QV string (used when post_sw_setup got use_read_qvs); sfr: the result of
// sw_full_cs -- qralign is re-called in place, matches / mismatches / crossovers are recounted, qual (malloc) and posterior are filled.
extern "C" void post_sw(uint32_t* read, int initbp, char* qual, struct gm_sw_full_results* sfr) {
  if (!g_ps.init) abort();                                   // ref: sw-post.c:657-658
  if (!sfr || !sfr->dbalign || !sfr->qralign) abort();
  SeamTimer tm(&g_ps.secs); g_ps.invocs++;
  FHit h; h.db = sfr->dbalign; h.qr = sfr->qralign;
  cs_post_sw(g_ps.K, read, initbp, sfr->read_start, h, g_ps.use_read_qvs ? qual : nullptr, g_ps.qual_delta, true);
  size_t len = 0; for (char c : h.qr) len += c != '-';
  g_ps.cells += 16 * (uint64_t)len;
  memcpy(sfr->qralign, h.qr.data(), h.qr.size());
  sfr->matches = h.cs_match; sfr->mismatches = h.cs_mismatch; sfr->crossovers = h.cs_xover;
  sfr->qual = (char*)malloc(h.qr.size() + 1);
  if (sfr->qual) { memcpy(sfr->qual, h.qual.data(), h.qual.size()); sfr->qual[h.qual.size()] = 0; }
  sfr->posterior = h.posterior;
}
