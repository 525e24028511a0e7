// gm_oracle_main.cpp -- CLI + C entry points around gm_oracle.hpp.
// TEST INFRASTRUCTURE ONLY (see gm_oracle.hpp).  Build: make -C oracle
//
//   gm_oracle [-N threads] [-Z] [--sam-unaligned] reads.fa genome.fa   > out.sam
//
// mirrors `gmapper-ls reads.fa genome.fa` for FASTA input (gmapper/gmapper.c:1720-3109).
#include "gm_oracle.hpp"
#include <omp.h>
#include <chrono>
#include <fstream>
#include <iostream>

using namespace gmo;

// fasta_get_next_read_with_range / extract_name (common/fasta.c:242-277,303-545), FASTA only
static bool read_fasta(const char* path, std::vector<std::string>& names, std::vector<std::string>& seqs) {
  std::ifstream f(path);
  if (!f) return false;
  std::string line;
  while (std::getline(f, line)) {
    if (!line.empty() && line.back() == '\r') line.pop_back();
    if (line.empty() || line[0] == '#') continue;
    if (line[0] == '>') {
      std::string nm = line.substr(1);
      size_t tab = nm.find('\t'); if (tab != std::string::npos) nm = nm.substr(0, tab);
      size_t b = 0; while (b < nm.size() && isspace((unsigned char)nm[b])) b++;   // strtrim
      nm = nm.substr(b);
      size_t e = 0; while (e < nm.size() && nm[e] != ' ' && nm[e] != '\t') e++;
      names.push_back(nm.substr(0, e)); seqs.emplace_back();
    } else if (!seqs.empty()) seqs.back() += line;
  }
  return true;
}

struct Session {
  Mapper M; Genome G; Index I;
};

static uint64_t g_last_pair_counts[2] = {0, 0};   // anchors, windows of the last paired call
static void map_all(const Session& S, std::vector<Read>& reads, int nthreads, std::string& out, Stats* stats_out) {
  const int chunk = 64;   // finer than the reference's 1000-read chunks so that a bounded sample still fills every core
  int nchunks = (int)((reads.size() + chunk - 1) / chunk);
  std::vector<std::string> outs(nchunks);
  std::vector<Stats> st(nthreads);
#pragma omp parallel num_threads(nthreads)
  {
    ThreadState T; S.M.init_thread(T);
#pragma omp for schedule(dynamic, 1)
    for (int c = 0; c < nchunks; c++) {
      size_t lo = (size_t)c * chunk, hi = std::min(reads.size(), lo + chunk);
      if (S.M.P.pair_mode != 0) {           // mates are adjacent; chunk is even (gmapper.c:2319-2322)
        for (size_t r = lo; r + 1 < hi; r += 2) {
          S.M.prepare_read(reads[r]); S.M.prepare_read(reads[r + 1]);
          if (reads[r].read_len > S.M.P.longest_read_len || reads[r + 1].read_len > S.M.P.longest_read_len) continue;
          S.M.handle_readpair(T, reads[r], reads[r + 1], outs[c]);
        }
      } else
      for (size_t r = lo; r < hi; r++) {
        S.M.prepare_read(reads[r]);
        if (reads[r].read_len > S.M.P.longest_read_len) continue;
        S.M.handle_read(T, reads[r], outs[c]);
      }
    }
    st[omp_get_thread_num()] = T.stats;
    st[omp_get_thread_num()].full_cells = T.sww.local_retries;      // reported in the stats' 7th slot
  }
  for (auto& o : outs) out += o;
  g_last_pair_counts[0] = g_last_pair_counts[1] = 0;
  for (auto& s : st) { g_last_pair_counts[0] += s.pair_anchors; g_last_pair_counts[1] += s.pair_windows; }
  if (stats_out) {
    for (auto& s : st) {
      stats_out->vec_calls += s.vec_calls; stats_out->vec_cells += s.vec_cells; stats_out->vec_bypassed += s.vec_bypassed;
      stats_out->full_calls += s.full_calls; stats_out->full_cells += s.full_cells; stats_out->reads_matched += s.reads_matched;
      stats_out->dup_pruned += s.dup_pruned;
    }
  }
}

#ifdef GM_ORACLE_MAIN
int main(int argc, char** argv) {
  int nthreads = 1; bool noz = false, unal = false, colour = strstr(argv[0], "-cs") != nullptr; int pair_mode = 0, ins_min = 0, ins_max = 1000;
  std::vector<const char*> pos;
  std::string cl;
  for (int i = 0; i < argc; i++) { if (i) cl += ' '; cl += argv[i]; }
  for (int i = 1; i < argc; i++) {
    if (!strcmp(argv[i], "-N") && i + 1 < argc) nthreads = atoi(argv[++i]);
    else if (!strcmp(argv[i], "-Z")) noz = true;
    else if (!strcmp(argv[i], "--sam-unaligned")) unal = true;
    else if (!strcmp(argv[i], "--cs")) colour = true;       // the reference picks the mode from the binary's name (util.c:28-38)
    else if (!strcmp(argv[i], "-p") && i + 1 < argc) { const char* m = argv[++i]; pair_mode = !strcmp(m, "opp-in") ? 1 : !strcmp(m, "opp-out") ? 2 : !strcmp(m, "col-fw") ? 3 : !strcmp(m, "col-bw") ? 4 : 0; }
    else if (!strcmp(argv[i], "-I") && i + 1 < argc) { sscanf(argv[++i], "%d,%d", &ins_min, &ins_max); }
    else pos.push_back(argv[i]);
  }
  if (pos.size() != 2) { fprintf(stderr, "usage: gm_oracle [-N n] [-Z] [--sam-unaligned] reads.fa genome.fa\n"); return 1; }
  Session S;
  load_default_seeds(S.M.P); derive_score_probs(S.M.P);
  if (colour) { set_colour_space(S.M.P); S.G.colour = true; }
  S.M.P.hash_filter_calls = !noz; S.M.P.sam_unaligned = unal;
  S.M.P.pair_mode = pair_mode; S.M.P.min_insert_size = ins_min; S.M.P.max_insert_size = ins_max;
  std::vector<std::string> gn, gs;
  if (!read_fasta(pos[1], gn, gs)) { fprintf(stderr, "cannot read genome\n"); return 1; }
  for (size_t c = 0; c < gn.size(); c++) {
    std::vector<uint8_t> codes(gs[c].size());
    for (size_t i = 0; i < codes.size(); i++) codes[i] = (uint8_t)char_to_code_ls((unsigned char)gs[c][i]);
    S.G.add_contig(gn[c], codes.data(), codes.size());
  }
  auto t0 = std::chrono::steady_clock::now();
  build_index(S.M.P, S.G, S.I);
  S.M.P.list_cutoff = auto_list_cutoff(S.M.P, S.G);
  S.M.G = &S.G; S.M.I = &S.I;
  auto t1 = std::chrono::steady_clock::now();
  std::vector<std::string> rn, rs;
  if (!read_fasta(pos[0], rn, rs)) { fprintf(stderr, "cannot read reads\n"); return 1; }
  std::vector<Read> reads(rn.size());
  for (size_t i = 0; i < rn.size(); i++) { reads[i].name = rn[i]; reads[i].seq = rs[i]; }
  printf("@HD\tVN:1.0\tSO:unsorted\n");
  for (int c = 0; c < S.G.num_contigs(); c++) printf("@SQ\tSN:%s\tLN:%u\n", S.G.names[c].c_str(), S.G.len[c]);
  printf("@PG\tID:gmapper\tVN:2.2.3\tCL:%s\n", cl.c_str());
  std::string out; Stats st;
  auto t2 = std::chrono::steady_clock::now();
  map_all(S, reads, nthreads, out, &st);
  auto t3 = std::chrono::steady_clock::now();
  fwrite(out.data(), 1, out.size(), stdout);
  fprintf(stderr, "oracle: index %.2fs, mapping %.3fs (%zu reads, %d threads, %.0f reads/s); vec calls %llu bypassed %llu full calls %llu matched %llu dup %llu cutoff %u\n",
          std::chrono::duration<double>(t1 - t0).count(), std::chrono::duration<double>(t3 - t2).count(), reads.size(), nthreads,
          reads.size() / std::chrono::duration<double>(t3 - t2).count(),
          (unsigned long long)st.vec_calls, (unsigned long long)st.vec_bypassed, (unsigned long long)st.full_calls,
          (unsigned long long)st.reads_matched, (unsigned long long)st.dup_pruned, S.M.P.list_cutoff);
  return 0;
}
#endif

// ---------------------------------------------------------------------------------------------
// C entry points for tests/ (ctypes).  Default letter-space parameters throughout.
// ---------------------------------------------------------------------------------------------
static Params default_params() { Params P; load_default_seeds(P); derive_score_probs(P); return P; }

extern "C" {

int gmo_sw_vector(const uint32_t* genome, int goff, int glen, const uint32_t* read, int rlen) {
  static const Params P = default_params();
  return sw_vector(P, genome, goff, glen, read, rlen);
}

// out[9] = score read_start rmapped genome_start gmapped matches mismatches insertions deletions
int gmo_sw_full_ls(const uint32_t* genome, int goff, int glen, const uint32_t* read, int rlen,
                   long long ax, long long ay, int alen, int awidth, int revcmpl,
                   int* out, char* dbalign, char* qralign, int cap) {
  static const Params P = default_params();
  SwFullWorkspace W; SwFullResults s;
  Anchor a; a.x = ax; a.y = ay; a.length = alen; a.width = awidth; a.weight = 1;
  sw_full_ls(P, W, genome, goff, glen, read, rlen, 0, 0, &s, revcmpl != 0, &a, 1, 0);
  int v[9] = {s.score, s.read_start, s.rmapped, s.genome_start, s.gmapped, s.matches, s.mismatches, s.insertions, s.deletions};
  memcpy(out, v, sizeof v);
  if ((int)s.dbalign.size() + 1 > cap) return -1;
  strcpy(dbalign, s.dbalign.c_str()); strcpy(qralign, s.qralign.c_str());
  return 0;
}

// local mode (Gflag off): anchors optional (has_anchor = 0: the threshold band), thresh / maxscore as gmapper passes them
int gmo_sw_full_ls_local(const uint32_t* genome, int goff, int glen, const uint32_t* read, int rlen, int thresh, int maxscore,
                         long long ax, long long ay, int alen, int awidth, int has_anchor, int revcmpl,
                         int* out, char* dbalign, char* qralign, int cap) {
  static const Params P = default_params();
  SwFullWorkspace W; SwFullResults s;
  Anchor a; a.x = ax; a.y = ay; a.length = alen; a.width = awidth; a.weight = 1;
  if (has_anchor) sw_full_ls(P, W, genome, goff, glen, read, rlen, thresh, maxscore, &s, revcmpl != 0, &a, 1, 1);
  else sw_full_ls(P, W, genome, goff, glen, read, rlen, thresh, maxscore, &s, revcmpl != 0, nullptr, 0, 1);
  int v[9] = {s.score, s.read_start, s.rmapped, s.genome_start, s.gmapped, s.matches, s.mismatches, s.insertions, s.deletions};
  memcpy(out, v, sizeof v);
  if ((int)s.dbalign.size() + 1 > cap) return -1;
  strcpy(dbalign, s.dbalign.c_str()); strcpy(qralign, s.qralign.c_str());
  return 0;
}

// colour space kernels with the binary's CS defaults (gmapper-defaults.h:52-58); out[10] = ... deletions crossovers
int gmo_sw_vector_cs(const uint32_t* genome_cs, int goff, int glen, const uint32_t* read, int rlen, const uint32_t* genome_ls, int initbp) {
  Params P = default_params();
  return sw_vector_cs(P, 10 + (-20), genome_cs, goff, glen, read, rlen, genome_ls, initbp);
}
int gmo_sw_full_cs_mode(const uint32_t* genome_ls, int goff, int glen, const uint32_t* read, int rlen, int initbp, int thresh,
                        long long ax, long long ay, int alen, int awidth, int revcmpl, int local, int* out, char* dbalign, char* qralign, int cap);
int gmo_sw_full_cs(const uint32_t* genome_ls, int goff, int glen, const uint32_t* read, int rlen, int initbp, int thresh,
                   long long ax, long long ay, int alen, int awidth, int revcmpl, int* out, char* dbalign, char* qralign, int cap) {
  return gmo_sw_full_cs_mode(genome_ls, goff, glen, read, rlen, initbp, thresh, ax, ay, alen, awidth, revcmpl, 0, out, dbalign, qralign, cap);
}
int gmo_sw_full_cs_mode(const uint32_t* genome_ls, int goff, int glen, const uint32_t* read, int rlen, int initbp, int thresh,
                        long long ax, long long ay, int alen, int awidth, int revcmpl, int local, int* out, char* dbalign, char* qralign, int cap) {
  CsParams C; SwFullCsResults s;
  Anchor a; a.x = ax; a.y = ay; a.length = alen; a.width = awidth; a.weight = 1;
  sw_full_cs(C, genome_ls, goff, glen, read, rlen, initbp, thresh, &s, revcmpl != 0, &a, 1, nullptr, local);
  int v[10] = {s.score, s.read_start, s.rmapped, s.genome_start, s.gmapped, s.matches, s.mismatches, s.insertions, s.deletions, s.crossovers};
  memcpy(out, v, sizeof v);
  if ((int)s.dbalign.size() + 1 > cap) return -1;
  strcpy(dbalign, s.dbalign.c_str()); strcpy(qralign, s.qralign.c_str());
  return 0;
}

// sw_gapless with the scores sw_gapless_setup got (colour space: `mismatch` is what f1_setup hands over, match + crossover)
int gmo_sw_gapless(const uint32_t* genome, int glen, const uint32_t* read, int rlen, int g_idx, int r_idx, const uint32_t* genome_ls, int init_bp, int is_rna, int match, int mismatch) {
  Params P = default_params();
  P.match_score = match;
  if (genome_ls) P.crossover_score = mismatch - match; else P.mismatch_score = mismatch;
  return sw_gapless(P, genome, glen, read, rlen, g_idx, r_idx, genome_ls, init_bp, is_rna != 0);
}

// the same two functions with is_rna set (what gmapper passes for a genome whose last contig is RNA, ref: genome.c:1063-1064): U reads as T in lstocs, cstols hands back U for T
int gmo_sw_vector_cs_rna(const uint32_t* genome_cs, int goff, int glen, const uint32_t* read, int rlen, const uint32_t* genome_ls, int initbp) {
  Params P = default_params();
  return sw_vector_cs(P, 10 + (-20), genome_cs, goff, glen, read, rlen, genome_ls, initbp, true);
}
int gmo_sw_full_cs_rna(const uint32_t* genome_ls, int goff, int glen, const uint32_t* read, int rlen, int initbp, int thresh,
                       long long ax, long long ay, int alen, int awidth, int revcmpl, int local, int* out, char* dbalign, char* qralign, int cap) {
  CsParams C; SwFullCsResults s;
  Anchor a; a.x = ax; a.y = ay; a.length = alen; a.width = awidth; a.weight = 1;
  sw_full_cs(C, genome_ls, goff, glen, read, rlen, initbp, thresh, &s, revcmpl != 0, &a, 1, nullptr, local, true);
  int v[10] = {s.score, s.read_start, s.rmapped, s.genome_start, s.gmapped, s.matches, s.mismatches, s.insertions, s.deletions, s.crossovers};
  memcpy(out, v, sizeof v);
  if ((int)s.dbalign.size() + 1 > cap) return -1;
  strcpy(dbalign, s.dbalign.c_str()); strcpy(qralign, s.qralign.c_str());
  return 0;
}

// sw_full_cs with a per-position crossover_score[] (ref: sw-full-cs.c:312; gmapper.c:532-544 builds it from the read's QVs), either mode
int gmo_sw_full_cs_xover(const uint32_t* genome_ls, int goff, int glen, const uint32_t* read, int rlen, int initbp, int thresh,
                         long long ax, long long ay, int alen, int awidth, int revcmpl, int local, const int* xover, int* out, char* dbalign, char* qralign, int cap) {
  CsParams C; SwFullCsResults s;
  Anchor a; a.x = ax; a.y = ay; a.length = alen; a.width = awidth; a.weight = 1;
  sw_full_cs(C, genome_ls, goff, glen, read, rlen, initbp, thresh, &s, revcmpl != 0, &a, 1, xover, local);
  int v[10] = {s.score, s.read_start, s.rmapped, s.genome_start, s.gmapped, s.matches, s.mismatches, s.insertions, s.deletions, s.crossovers};
  memcpy(out, v, sizeof v);
  if ((int)s.dbalign.size() + 1 > cap) return -1;
  strcpy(dbalign, s.dbalign.c_str()); strcpy(qralign, s.qralign.c_str());
  return 0;
}

// opts = "key=value;key=value": the reference's command-line options by their long names (gmapper.c:1040-1140):
// match mismatch open-r ext-r open-q ext-q match-window cmw-overlap cmw-threshold vec-threshold full-threshold
// cmw-mode report anchor-width cutoff strata max-alignments seeds (comma separated 0/1 strings)
static void apply_opts(Params& P, const char* opts) {
  if (!opts) return;
  std::string s(opts); size_t i = 0;
  if (s.find("colour=1") != std::string::npos) set_colour_space(P);   // first: it changes the defaults the other keys override
  while (i < s.size()) {
    size_t e = s.find(';', i); if (e == std::string::npos) e = s.size();
    std::string kv = s.substr(i, e - i); i = e + 1;
    size_t q = kv.find('='); if (q == std::string::npos) continue;
    const std::string k = kv.substr(0, q), v = kv.substr(q + 1); const double d = atof(v.c_str());
    if (k == "match") P.match_score = (int)d; else if (k == "mismatch") P.mismatch_score = (int)d;
    else if (k == "open-r") P.a_gap_open_score = (int)d; else if (k == "ext-r") P.a_gap_extend_score = (int)d;
    else if (k == "open-q") P.b_gap_open_score = (int)d; else if (k == "ext-q") P.b_gap_extend_score = (int)d;
    else if (k == "match-window") P.window_len = d; else if (k == "cmw-overlap") P.window_overlap = d;
    else if (k == "cmw-threshold") P.window_gen_threshold = d; else if (k == "vec-threshold") P.sw_vect_threshold = d;
    else if (k == "full-threshold") P.sw_full_threshold = d; else if (k == "cmw-mode") P.match_mode = (int)d;
    else if (k == "report") P.num_outputs = (int)d; else if (k == "anchor-width") P.anchor_width = (int)d;
    else if (k == "cutoff") P.list_cutoff = (uint32_t)d; else if (k == "strata") P.strata = d != 0;
    else if (k == "max-alignments") P.max_alignments = (int)d;
    else if (k == "region-bits") P.region_bits = (int)d; else if (k == "region-overlap") P.region_overlap = (int)d;
    else if (k == "pr-xover") P.pr_xover = d;
    else if (k == "tiebreak-off") { if (d != 0) P.Tflag = false; }                       // -t: no reversed tie-breaks on the negative strand (mapping.c:378,393)
    else if (k == "single-best-mapping") P.single_best_mapping = d != 0;
    else if (k == "all-contigs") P.all_contigs = d != 0;
    else if (k == "no-mapping-qualities") { if (d != 0) P.compute_mapping_qualities = false; }
    else if (k == "no-improper-mappings") P.improper_mappings = d == 0;
    else if (k == "positive") { if (d != 0) { P.Fflag = true; P.Cflag = false; } }      // -F
    else if (k == "negative") { if (d != 0) { P.Cflag = true; P.Fflag = false; } }      // -C
    else if (k == "mp-match-mode") P.mp_match_mode = (int)d;     // 4 (default) / 3: the paired option set's match mode; 0 switches the mate-pair region counts off (sensitivity checks)
    else if (k == "half-paired") P.half_paired = d != 0;        // --no-half-paired: mate-pair region counts, no unpaired rescue (gmapper.c:2657-2683)
    else if (k == "local") { P.Gflag = d == 0; if (!P.Gflag) P.compute_mapping_qualities = false; }   // --local (gmapper.c:2303-2305,2325-2328)
    else if (k == "ungapped") { if (d != 0) { P.gapless = true; P.anchor_width = 0; P.a_gap_open_score = -255; P.b_gap_open_score = -255; P.hash_filter_calls = false; } }   // -U (gmapper.c:2057-2062)
    else if (k == "hash-spaced-kmers") P.Hflag = d != 0;     // -H
    else if (k == "crossover") P.crossover_score = (int)d; else if (k == "indel-taboo-len") P.indel_taboo_len = (int)d;
    else if (k == "seeds") {
      P.seeds.clear(); P.max_seed_span = 0; P.min_seed_span = 64;
      size_t a = 0; while (a < v.size()) { size_t b = v.find(',', a); if (b == std::string::npos) b = v.size(); add_spaced_seed(P, v.substr(a, b - a).c_str()); a = b + 1; }
    }
  }
  derive_score_probs(P);
}

void* gmo_session_create_opts(int n_contigs, const uint8_t* const* codes, const uint64_t* lens, const char* const* names, const char* opts);
void* gmo_session_create(int n_contigs, const uint8_t* const* codes, const uint64_t* lens, const char* const* names) {
  return gmo_session_create_opts(n_contigs, codes, lens, names, nullptr);
}
void* gmo_session_create_opts(int n_contigs, const uint8_t* const* codes, const uint64_t* lens, const char* const* names, const char* opts) {
  Session* S = new Session();
  S->M.P = default_params();
  S->M.P.list_cutoff = 4294967295u;
  apply_opts(S->M.P, opts);
  S->G.colour = S->M.P.colour;
  for (int c = 0; c < n_contigs; c++) {
    char nm[64]; snprintf(nm, sizeof nm, "contig%d", c + 1);
    S->G.add_contig(names && names[c] ? names[c] : nm, codes[c], (size_t)lens[c]);
  }
  const double t0 = omp_get_wtime();
  build_index(S->M.P, S->G, S->I);
  if (getenv("GMO_VERBOSE")) fprintf(stderr, "gm_oracle: index built in %.1f s\n", omp_get_wtime() - t0);
  if (S->M.P.list_cutoff == 4294967295u) S->M.P.list_cutoff = auto_list_cutoff(S->M.P, S->G);     // DEF_LIST_CUTOFF = automatic (gmapper.c:2811-2837)
  S->M.G = &S->G; S->M.I = &S->I;
  return S;
}
void gmo_session_destroy(void* s) { delete (Session*)s; }
void gmo_session_set_half_paired(void* s, int on) { ((Session*)s)->M.P.half_paired = on != 0; }   // --no-half-paired on an existing session (no index change)
unsigned gmo_session_cutoff(void* s) { return ((Session*)s)->M.P.list_cutoff; }
// the chunk-parallel index builder against the sequential restatement of load_genome (genome.c:1012-1182)
int gmo_index_selfcheck(void* s, int nthreads) {
  Session* S = (Session*)s; Index a, b;
  build_index_seq(S->M.P, S->G, a); build_index(S->M.P, S->G, b, nthreads);
  return a.start == b.start && a.pos == b.pos && b.start == S->I.start && b.pos == S->I.pos;
}
void gmo_set_threads(int n) { if (n > 0) omp_set_num_threads(n); }
void gmo_session_set_pairing(void* s, int pair_mode, int min_insert, int max_insert) {
  Session* S = (Session*)s; S->M.P.pair_mode = pair_mode; S->M.P.min_insert_size = min_insert; S->M.P.max_insert_size = max_insert;
}
void gmo_session_set(void* s, int hash_filter_calls, int sam_unaligned) {
  Session* S = (Session*)s; S->M.P.hash_filter_calls = hash_filter_calls != 0; S->M.P.sam_unaligned = sam_unaligned != 0;
}

static std::string code_seq(const uint8_t* c, int L) { std::string s(L, 'N'); for (int i = 0; i < L; i++) s[i] = LSTRANS[c[i] & 15]; return s; }
// colour-space read: code 0 is the primer letter, the rest are colours (0-3, anything else '.')
static std::string code_seq_cs(const uint8_t* c, int L) { std::string s(L, '.'); if (L) s[0] = LSTRANS[c[0] & 3]; for (int i = 1; i < L; i++) if (c[i] < 4) s[i] = (char)('0' + c[i]); return s; }

// reads: n x L code matrix (row-major); names: '\n'-separated or NULL (-> r<i>); returns malloc'd SAM body (no header)
char* gmo_map_sam(void* s, int n, int L, const uint8_t* codes, const char* names, int nthreads, uint64_t* stats7) {
  Session* S = (Session*)s;
  std::vector<Read> reads(n);
  const char* p = names;
  for (int i = 0; i < n; i++) {
    if (p) { const char* e = strchr(p, '\n'); if (!e) e = p + strlen(p); reads[i].name.assign(p, e); p = *e ? e + 1 : e; }
    else { char nm[32]; snprintf(nm, sizeof nm, "r%d", i); reads[i].name = nm; }
    reads[i].seq = S->M.P.colour ? code_seq_cs(codes + (size_t)i * L, L) : code_seq(codes + (size_t)i * L, L);
  }
  std::string out; Stats st;
  map_all(*S, reads, nthreads > 0 ? nthreads : 1, out, &st);
  if (stats7) { stats7[0] = st.vec_calls; stats7[1] = st.vec_cells; stats7[2] = st.vec_bypassed; stats7[3] = st.full_calls; stats7[4] = st.reads_matched; stats7[5] = st.dup_pruned; stats7[6] = st.full_cells; }
  char* r = (char*)malloc(out.size() + 1);
  memcpy(r, out.data(), out.size()); r[out.size()] = 0;
  return r;
}
// FASTQ reads: quals = '\n'-separated QUAL strings as read from the file, qual_delta = --qv-offset (64 by default in letter space)
char* gmo_map_sam_q(void* s, int n, int L, const uint8_t* codes, const char* names, const char* quals, int qual_delta, int nthreads) {
  Session* S = (Session*)s;
  std::vector<Read> reads(n);
  const char* p = names; const char* q = quals;
  for (int i = 0; i < n; i++) {
    if (p) { const char* e = strchr(p, '\n'); if (!e) e = p + strlen(p); reads[i].name.assign(p, e); p = *e ? e + 1 : e; }
    else { char nm[32]; snprintf(nm, sizeof nm, "r%d", i); reads[i].name = nm; }
    { const char* e = strchr(q, '\n'); if (!e) e = q + strlen(q); reads[i].qual.assign(q, e); q = *e ? e + 1 : e; }
    reads[i].seq = S->M.P.colour ? code_seq_cs(codes + (size_t)i * L, L) : code_seq(codes + (size_t)i * L, L);
  }
  const bool oq = S->M.P.Qflag; const int od = S->M.P.qual_delta;
  S->M.P.Qflag = true; S->M.P.qual_delta = qual_delta;
  std::string out;
  map_all(*S, reads, nthreads > 0 ? nthreads : 1, out, nullptr);
  S->M.P.Qflag = oq; S->M.P.qual_delta = od;
  char* r = (char*)malloc(out.size() + 1);
  memcpy(r, out.data(), out.size()); r[out.size()] = 0;
  return r;
}
// pairs: mates 1 are n x L1 codes, mates 2 are n x L2 codes; names '\n'-separated (or NULL -> p<i>/1, p<i>/2)
char* gmo_map_pairs_sam(void* s, int n, int L1, const uint8_t* codes1, int L2, const uint8_t* codes2,
                        const char* names1, const char* names2, int nthreads) {
  Session* S = (Session*)s;
  std::vector<Read> reads((size_t)2 * n);
  const char* p1 = names1; const char* p2 = names2;
  auto next_name = [](const char*& p, std::string& dst) { const char* e = strchr(p, '\n'); if (!e) e = p + strlen(p); dst.assign(p, e); p = *e ? e + 1 : e; };
  for (int i = 0; i < n; i++) {
    char nm[40];
    if (p1) next_name(p1, reads[2 * i].name); else { snprintf(nm, sizeof nm, "p%d/1", i); reads[2 * i].name = nm; }
    if (p2) next_name(p2, reads[2 * i + 1].name); else { snprintf(nm, sizeof nm, "p%d/2", i); reads[2 * i + 1].name = nm; }
    reads[2 * i].seq = S->M.P.colour ? code_seq_cs(codes1 + (size_t)i * L1, L1) : code_seq(codes1 + (size_t)i * L1, L1);
    reads[2 * i + 1].seq = S->M.P.colour ? code_seq_cs(codes2 + (size_t)i * L2, L2) : code_seq(codes2 + (size_t)i * L2, L2);
  }
  std::string out;
  map_all(*S, reads, nthreads > 0 ? nthreads : 1, out, nullptr);
  char* r = (char*)malloc(out.size() + 1);
  memcpy(r, out.data(), out.size()); r[out.size()] = 0;
  return r;
}
// FASTQ pairs: as gmo_map_pairs_sam plus the mates' QUAL strings ('\n' separated) and the file's quality offset
char* gmo_map_pairs_sam_q(void* s, int n, int L1, const uint8_t* codes1, int L2, const uint8_t* codes2, const char* names1, const char* names2,
                          const char* quals1, const char* quals2, int qual_delta, int nthreads) {
  Session* S = (Session*)s;
  std::vector<Read> reads((size_t)2 * n);
  const char* p1 = names1; const char* p2 = names2; const char* q1 = quals1; const char* q2 = quals2;
  auto next = [](const char*& p, std::string& dst) { const char* e = strchr(p, '\n'); if (!e) e = p + strlen(p); dst.assign(p, e); p = *e ? e + 1 : e; };
  for (int i = 0; i < n; i++) {
    char nm[40];
    if (p1) next(p1, reads[2 * i].name); else { snprintf(nm, sizeof nm, "p%d/1", i); reads[2 * i].name = nm; }
    if (p2) next(p2, reads[2 * i + 1].name); else { snprintf(nm, sizeof nm, "p%d/2", i); reads[2 * i + 1].name = nm; }
    next(q1, reads[2 * i].qual); next(q2, reads[2 * i + 1].qual);
    reads[2 * i].seq = code_seq(codes1 + (size_t)i * L1, L1);
    reads[2 * i + 1].seq = code_seq(codes2 + (size_t)i * L2, L2);
  }
  const bool oq = S->M.P.Qflag; const int od = S->M.P.qual_delta;
  S->M.P.Qflag = true; S->M.P.qual_delta = qual_delta;
  std::string out;
  map_all(*S, reads, nthreads > 0 ? nthreads : 1, out, nullptr);
  S->M.P.Qflag = oq; S->M.P.qual_delta = od;
  char* r = (char*)malloc(out.size() + 1);
  memcpy(r, out.data(), out.size()); r[out.size()] = 0;
  return r;
}
void gmo_free(void* p) { free(p); }
void gmo_last_pair_counts(uint64_t* out2) { out2[0] = g_last_pair_counts[0]; out2[1] = g_last_pair_counts[1]; }

// Stage dump for one batch, for GPU-vs-oracle stage parity: for every read the pass-1 survivors
// (top-K heap array order) as rows of 12 ints:
//   read st cn g_off w_len score_vector pct_score_vector matches anchor.x anchor.y anchor.length anchor.width
// (hit state *before* pass 2 / reverse_hit).  Returns number of rows written (<= cap).
long gmo_map_tophits(void* s, int n, int L, const uint8_t* codes, int nthreads, long long* rows, long cap) {
  Session* S = (Session*)s;
  std::vector<std::vector<long long>> per(n);
#pragma omp parallel num_threads(nthreads > 0 ? nthreads : 1)
  {
    ThreadState T; S->M.init_thread(T);
#pragma omp for schedule(dynamic, 64)
    for (int i = 0; i < n; i++) {
      Read re; re.seq = code_seq(codes + (size_t)i * L, L); re.name = "r";
      S->M.prepare_read(re);
      S->M.read_get_mapidxs(re);
      T.region_map_id++; T.region_map_id &= ((1 << Mapper::region_map_id_bits) - 1);
      S->M.read_get_region_counts(T, re, 0); S->M.read_get_region_counts(T, re, 1);
      S->M.read_get_anchor_list(T, re, 0); S->M.read_get_anchor_list(T, re, 1);
      S->M.read_get_hit_list(re, 0); S->M.read_get_hit_list(re, 1);
      S->M.read_pass1(T, re, 0); S->M.read_pass1(T, re, 1);
      std::vector<Hit*> p1; int n1 = 0;
      S->M.read_get_vector_hits(re, p1, n1);
      for (int k = 0; k < n1; k++) {
        Hit* h = p1[k];
        long long r[12] = {i, h->st, h->cn, h->g_off, h->w_len, h->score_vector, h->pct_score_vector, h->matches,
                           h->anchor.x, h->anchor.y, h->anchor.length, h->anchor.width};
        per[i].insert(per[i].end(), r, r + 12);
      }
    }
  }
  long w = 0;
  for (int i = 0; i < n; i++) for (size_t k = 0; k + 12 <= per[i].size(); k += 12) { if (w >= cap) return w; memcpy(rows + w * 12, &per[i][k], 12 * sizeof(long long)); w++; }
  return w;
}

}  // extern "C"
