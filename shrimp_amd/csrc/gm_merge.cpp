// gm_merge.cpp -- SURVEY 8(f)3: merging the SAM outputs of several gmapper runs (genome shards and / or read shards) with the mapping qualities
// recomputed from the Z0-Z6 fields, as the reference's mergesam does (ref: mergesam/mergesam.c, mergesam/sam_reader.c, mergesam/render.c,
// mergesam/mergesam_heap.c, SPLITTING_AND_MERGING:100-148).  Host code: the work is text in, text out; on MI355X the whole index of a 3 Gbp genome is
// resident on every GPU, so this step is only needed for genomes split over ranks or for shards mapped elsewhere.
//
// Layout differs from the reference (which streams files through ring buffers and linked lists of `pretty` records, 40 000 reads at a time): the inputs
// are whole texts in memory; one sequential sweep per file assigns every record line to its read (the reference's prefix match against the reads file,
// sam_reader.c:893-1031), then blocks of reads are combined and rendered by a pool of threads into per-block strings.  What is kept to the letter is
// everything that decides a byte of the output: the bounded heap and its array order (mergesam_heap.c), the class lists and their order, the Z-field
// arithmetic (inv_tnlog / tnlog through glibc's exp / log, as the reference), the header sort, and the rendering rules.
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <string_view>
#include <thread>
#include <vector>
#include "../../include/gmapper_hip.h"

void gm_set_error(const char* fmt, ...);

namespace {

using sv = std::string_view;

enum { PAIRED = 0, UNPAIRED = 2, FIRST_LEG = 3, SECOND_LEG = 4, UNMAPPED = 5, N_CLASSES = 6 };   // ref: mergesam.h:15-21
enum { Z_PAIRED = (1 << 2) | (1 << 3) | (1 << 4) | (1 << 6), Z_HALF = (1 << 0) | (1 << 1) | (1 << 4) | (1 << 5), Z_UNPAIRED = (1 << 0) | (1 << 1) };   // ref: sam2pretty_lib.h:30-32

inline double inv_tnlog(int x) { return exp(-(x / 1000.0)); }                 // ref: sam2pretty_lib.c:12-18
inline int tnlog(double x) { return (int)(1000.0 * -log(x)); }
inline int qv_from_pr_corr(double pr_corr) {                                  // ref: common/util.h:266-282
  const double pr_err = 1 - pr_corr;
  if (pr_err > .99999999) return 0;
  if (pr_err < 1E-25) return 250;
  return (int)(-10.0 * log(pr_err) / log(10.0));
}
inline int sv_atoi(sv s) {                                                    // atoi on a field that ends at a tab
  size_t i = 0; while (i < s.size() && (s[i] == ' ' || (s[i] >= '\t' && s[i] <= '\r'))) i++;
  bool neg = false; if (i < s.size() && (s[i] == '-' || s[i] == '+')) { neg = s[i] == '-'; i++; }
  long long v = 0; for (; i < s.size() && s[i] >= '0' && s[i] <= '9'; i++) v = v * 10 + (s[i] - '0');
  return (int)(neg ? -v : v);
}

struct Rec {                                                                  // one alignment record (the reference's `pretty`, the fields mergesam touches)
  sv name, ref, cigar, mref, seq, qual, aux;
  bool has_aux = false;
  int flags = 0, pos = 0, mapq = 0, mpos = 0, isize = 0, score = 0;
  bool has_score = false;
  unsigned has_zs = 0; double z[7] = {0, 0, 0, 0, 0, 0, 0};
  int fileno = 0;
  bool paired_seq = false, proper = false, mapped = false, mp_mapped = false, rev = false, mp_rev = false, first = false, second = false,
       not_primary = false, qc_fail = false, dup = false;
  Rec* mate = nullptr;
  // filled by aux_fields() for the unaligned / FASTA / FASTQ renderings
  bool colour_space = false, has_csq = false, has_rg = false, has_r2 = false;
  sv cs_string, cs_qual, read_group, r2;
  std::string text;                                                           // the rendered line
};

// ref: sam2pretty_lib.c:1360-1514 (pretty_from_string_inplace)
bool parse_record(sv line, Rec& R) {
  size_t p = 0;
  auto field = [&](sv& out) -> bool { const size_t t = line.find('\t', p); if (t == sv::npos) return false; out = line.substr(p, t - p); p = t + 1; return true; };
  sv f_flags, f_pos, f_mapq, f_mpos, f_isize;
  if (!field(R.name) || !field(f_flags) || !field(R.ref) || !field(f_pos) || !field(f_mapq) || !field(R.cigar) || !field(R.mref) || !field(f_mpos) ||
      !field(f_isize) || !field(R.seq)) return false;
  R.flags = sv_atoi(f_flags); R.pos = sv_atoi(f_pos); R.mapq = sv_atoi(f_mapq); R.mpos = sv_atoi(f_mpos); R.isize = sv_atoi(f_isize);
  size_t t = line.find('\t', p);
  R.qual = line.substr(p, t == sv::npos ? sv::npos : t - p);
  if (t != sv::npos && t + 6 < line.size()) {                                 // optional fields shorter than this are dropped, as in the reference
    size_t s = t + 1;
    size_t nt = line.find('\t', s);
    if (line[s] == 'A' && line[s + 1] == 'S') {
      R.score = sv_atoi(line.substr(s + 5, nt == sv::npos ? sv::npos : nt - (s + 5))); R.has_score = true;
      if (nt != sv::npos && nt + 6 < line.size()) {
        for (int i = 0; nt != sv::npos && i < 7; i++) {
          if (line[nt + 1] != 'Z') break;
          const int zi = line[nt + 2] - 48;
          if (zi < 0 || zi > 6 || (R.has_zs & (1u << zi))) return false;
          R.has_zs |= 1u << zi;
          s = nt + 1;
          if (line.size() - s < 6) return false;
          nt = line.find('\t', s);
          R.z[zi] = inv_tnlog(sv_atoi(line.substr(s + 5, nt == sv::npos ? sv::npos : nt - (s + 5))));
        }
        if (nt != sv::npos) { R.aux = line.substr(nt + 1); R.has_aux = true; }
      } else if (nt != sv::npos) { R.aux = line.substr(nt); R.has_aux = true; }
    } else { R.aux = line.substr(s); R.has_aux = true; }
  }
  R.paired_seq = R.flags & 0x1; R.proper = R.flags & 0x2; R.mapped = !(R.flags & 0x4); R.mp_mapped = !(R.flags & 0x8); R.rev = R.flags & 0x10;
  R.mp_rev = R.flags & 0x20; R.first = R.flags & 0x40; R.second = R.flags & 0x80; R.not_primary = R.flags & 0x100; R.qc_fail = R.flags & 0x200;
  R.dup = R.flags & 0x400;
  return true;
}

int flag_of(const Rec& R) {                                                   // ref: sam2pretty_lib.c:62-87
  return (R.paired_seq ? 0x1 : 0) | (R.proper ? 0x2 : 0) | (R.mapped ? 0 : 0x4) | (R.mp_mapped ? 0 : 0x8) | (R.rev ? 0x10 : 0) | (R.mp_rev ? 0x20 : 0) |
         (R.first ? 0x40 : 0) | (R.second ? 0x80 : 0) | (R.not_primary ? 0x100 : 0) | (R.qc_fail ? 0x200 : 0) | (R.dup ? 0x400 : 0);
}

void aux_fields(Rec& R) {                                                     // ref: sam2pretty_lib.c:1251-1358 (the tags the unaligned / FASTA renderings use)
  if (!R.has_aux) return;
  sv a = R.aux;
  while (true) {
    const size_t t = a.find('\t'); const sv tok = a.substr(0, t);
    if (tok.size() >= 2) {
      const sv data = tok.size() >= 5 ? tok.substr(5) : sv();
      if (tok[0] == 'C' && tok[1] == 'S') { R.colour_space = true; R.cs_string = data; }
      else if (tok[0] == 'C' && tok[1] == 'Q') { R.has_csq = true; R.cs_qual = data; }
      else if (tok[0] == 'R' && tok[1] == 'G') { R.has_rg = true; R.read_group = data; }
      else if (tok[0] == 'R' && tok[1] == '2') { R.has_r2 = true; R.r2 = data; }
    }
    if (t == sv::npos) break;
    a = a.substr(t + 1);
  }
  R.has_aux = false;
}

void put_int(std::string& o, long long v) { char b[24]; const int n = snprintf(b, sizeof b, "%lld", v); o.append(b, n); }

void render_unaligned(Rec& R) {                                               // ref: render.c:27-60
  std::string& o = R.text; o.clear();
  o.append(R.name); o += '\t'; put_int(o, R.flags | 0x4 | 0x8); o += "\t*\t0\t0\t*\t*\t0\t0\t";
  if (R.colour_space) o += "*\t*"; else { o.append(R.seq); o += '\t'; o.append(R.qual); }
  if (R.colour_space) { if (R.has_csq) { o += "\tCQ:Z:"; o.append(R.cs_qual); } o += "\tCS:Z:"; o.append(R.cs_string); }
  if (R.has_rg) { o += "\tRG:Z:"; o.append(R.read_group); }
  if (R.has_r2) { o += "\tR2:Z:"; o.append(R.r2); }
  if (R.has_aux) { o += '\t'; o.append(R.aux); }
}

void render_sam(Rec& R) {                                                     // ref: render.c:210-277
  if (!R.mapped) { render_unaligned(R); return; }
  R.flags = flag_of(R);
  std::string& o = R.text; o.clear();
  o.append(R.name); o += '\t'; put_int(o, R.flags); o += '\t'; o.append(R.ref); o += '\t'; put_int(o, R.pos); o += '\t'; put_int(o, R.mapq >= 4 ? R.mapq : 0);
  o += '\t'; o.append(R.cigar); o += '\t'; if (R.ref == R.mref) o += '='; else o.append(R.mref);
  o += '\t'; put_int(o, R.mpos); o += '\t'; put_int(o, R.isize); o += '\t'; o.append(R.seq); o += '\t'; o.append(R.qual);
  if (R.has_score) { o += "\tAS:i:"; put_int(o, R.score); }
  for (int i = 0; i < 7; i++) if (R.has_zs & (1u << i)) { o += "\tZ"; o += (char)('0' + i); o += ":i:"; put_int(o, tnlog(R.z[i])); }
  if (R.colour_space) { if (R.has_csq) { o += "\tCQ:Z:"; o.append(R.cs_qual); } o += "\tCS:Z:"; o.append(R.cs_string); }
  if (R.has_r2) { o += "\tR2:Z:"; o.append(R.r2); }
  if (R.has_rg) { o += "\tRG:Z:"; o.append(R.read_group); }
  if (R.has_aux) { o += '\t'; o.append(R.aux); }
}

void render_fastx(Rec& R) {                                                   // ref: render.c:114-144
  const sv read = R.colour_space ? R.cs_string : R.seq;
  sv q; bool has_q = false;
  if (R.colour_space) { if (R.has_csq) { q = R.cs_qual; has_q = true; } } else if (!R.qual.empty() && R.qual[0] != '*') { q = R.qual; has_q = true; }
  std::string& o = R.text; o.clear();
  if (!read.empty() && !(read[0] == '*' && read.size() == 1)) {
    if (!has_q) { o += '>'; o.append(R.name); o += '\n'; o.append(read); o += '\n'; }
    else { o += '@'; o.append(R.name); o += '\n'; o.append(read); o += "\n+\n"; o.append(q); }
  }
}

// ref: mergesam_heap.c -- a min-heap on `score`, bounded; its ARRAY order is the output order
struct HeapElem { int score; Rec* rest; };
struct Heap {
  std::vector<HeapElem> a; int load = 0, capacity = 0;
  static bool less(const HeapElem& x, const HeapElem& y) { return x.score < y.score; }    // e_compare: the second score never decides (mergesam_heap.c:7-18)
  void up(int node) { int parent = node / 2; while (node > 1 && less(a[node - 1], a[parent - 1])) { std::swap(a[parent - 1], a[node - 1]); node = parent; parent = node / 2; } }
  void down(int node) {
    while (true) {
      const int left = node * 2, right = left + 1; int m = node;
      if (left <= load && less(a[left - 1], a[node - 1])) m = left;
      if (right <= load && less(a[right - 1], a[m - 1])) m = right;
      if (m == node) break;
      std::swap(a[m - 1], a[node - 1]); node = m;
    }
  }
  void insert_bounded(const HeapElem& e) {
    if (load < capacity) { if (load == 0) { a[0] = e; load = 1; } else { a[load] = e; load++; up(load); } }
    else if (less(a[0], e)) { a[0] = e; down(1); }
  }
  void insert_bounded_strata(const HeapElem& e) {
    if (load == 0) { a[0] = e; load = 1; }
    else if (less(e, a[0])) return;
    else if (less(a[0], e)) { load = 1; a[0] = e; }
    else if (load < capacity) a[load++] = e;
  }
};

struct Opts {
  gm_merge_options_t o; int n_files; int64_t genome_length;
  bool paired = false, unpaired = false;
};

typedef std::vector<Rec*> List;
struct ReadLists { std::vector<List> l; };                                    // [file * N_CLASSES + class]

struct MergeError { std::string msg; };

// ref: sam_reader.c:117-294
void consolidate_paired(const Opts& O, ReadLists& L, Rec** unaligned, Heap& h, List& result) {
  h.load = 0;
  const int F = O.n_files;
  std::vector<Rec*> best(F, nullptr); std::vector<char> summed(F, 0);
  double z3_sum = 0, global_ins_denom = 0, z4_min = 1.0;
  for (int i = 0; i < F; i++)
    for (Rec* pa : L.l[i * N_CLASSES + PAIRED]) {
      Rec* mp = pa->mate;
      if (!O.o.no_mapping_qualities) {
        if (pa->has_zs != (unsigned)Z_PAIRED) throw MergeError{"paired record of read " + std::string(pa->name) + " lacks the Z2 Z3 Z4 Z6 fields: mapping qualities cannot be recomputed"};
        const int mapq_score = pa->mapq + mp->mapq; const int fn = pa->fileno;
        if (!best[fn] || best[fn]->mapq + best[fn]->mate->mapq < mapq_score)
          best[fn] = (pa->mapq > mp->mapq || (pa->mapq == mp->mapq && pa->score > mp->score)) ? pa : mp;
      }
      if (!O.o.single_best) { const HeapElem e{pa->mapq + mp->mapq, pa}; if (O.o.strata) h.insert_bounded_strata(e); else h.insert_bounded(e); }
      if (pa->has_zs == (unsigned)Z_PAIRED) {
        if (!summed[pa->fileno]) { z3_sum += pa->z[3]; summed[pa->fileno] = 1; global_ins_denom += pa->z[6]; }
        z4_min = std::min(z4_min, pa->z[4]);
      }
    }
  if (O.o.single_best) {
    for (int i = 0; i < F; i++) if (Rec* pa = best[i]) { pa->z[3] = pa->mate->z[3] = z3_sum; pa->z[4] = pa->mate->z[4] = z4_min; }
    int bi = -1; double best_z2 = 0;
    for (int i = 0; i < F; i++) if (Rec* pa = best[i]) { const double nz = std::max(pa->z[2], pa->mate->z[2]); if (bi == -1 || best_z2 < nz) { best_z2 = nz; bi = i; } }
    if (bi != -1) { Rec* pa = best[bi]; pa->z[6] = global_ins_denom; pa->mate->z[6] = global_ins_denom; h.load = 0; h.insert_bounded(HeapElem{0, pa}); }
  } else {
    for (int i = 0; i < h.load; i++) { Rec* pa = h.a[i].rest; if (pa->has_zs == (unsigned)Z_PAIRED) { pa->z[3] = pa->mate->z[3] = z3_sum; pa->z[4] = pa->mate->z[4] = z4_min; pa->z[6] = pa->mate->z[6] = global_ins_denom; } }
  }
  result.clear();
  if (h.load > 0 && (O.o.max_alignments == 0 || h.load <= O.o.max_alignments))
    for (int i = h.load > O.o.max_outputs ? 1 : 0; i < h.load; i++) result.push_back(h.a[i].rest);
  if ((O.o.sam_unaligned || O.o.output == GM_MERGE_OUT_UNALIGNED_READS) && h.load > 0) *unaligned = h.a[0].rest;
}

// ref: sam_reader.c:296-399
void consolidate_single(const Opts& O, ReadLists& L, int cls, Rec** unaligned, Heap& h, List& result, bool& result_set) {
  h.load = 0;
  const int F = O.n_files;
  std::vector<char> summed(F, 0); double z1_sum = 0; Rec* max_pa = nullptr;
  for (int i = 0; i < F; i++)
    for (Rec* pa : L.l[i * N_CLASSES + cls]) {
      if (!O.o.no_mapping_qualities) {
        if ((pa->has_zs & Z_UNPAIRED) != (unsigned)Z_UNPAIRED) throw MergeError{"record of read " + std::string(pa->name) + " lacks the Z0 Z1 fields: mapping qualities cannot be recomputed"};
        if (!summed[pa->fileno]) { z1_sum += pa->z[1]; summed[pa->fileno] = 1; }
        if (!max_pa || max_pa->z[0] < pa->z[0]) max_pa = pa;
      }
      if (!O.o.single_best) { const HeapElem e{pa->score, pa}; if (O.o.strata) h.insert_bounded_strata(e); else h.insert_bounded(e); }
    }
  if (!O.o.no_mapping_qualities) {
    if (!max_pa) return;                                                      // nothing in this class: file 0's (empty) list stays
    if (O.o.single_best) { h.load = 1; h.a[0].rest = max_pa; }
    for (int i = 0; i < h.load; i++) { Rec* pa = h.a[i].rest; pa->z[1] = z1_sum; if (cls != UNPAIRED) pa->z[4] = max_pa->z[4]; }
  }
  result.clear(); result_set = true;
  if (h.load > 0 && (O.o.max_alignments == 0 || h.load <= O.o.max_alignments))
    for (int i = h.load > O.o.max_outputs ? 1 : 0; i < h.load; i++) result.push_back(h.a[i].rest);
  if ((O.o.sam_unaligned || O.o.output == GM_MERGE_OUT_UNALIGNED_READS) && h.load > 0) *unaligned = h.a[0].rest;
}

struct Cigar { std::vector<char> op; std::vector<int> len; };
Cigar cigar_of(sv c) {                                                        // ref: sam2pretty_lib.c:498-520
  Cigar C; int last = -1;
  for (int i = 0; i < (int)c.size(); i++) if (c[i] > 57) { C.op.push_back(c[i]); C.len.push_back(sv_atoi(c.substr(last + 1, i - last - 1))); last = i; }
  return C;
}
int genome_end_unpadded(const Rec& R) {                                       // ref: sam2pretty_lib.c:523-563
  const Cigar C = cigar_of(R.cigar);
  if (C.op.empty()) throw MergeError{"read " + std::string(R.name) + " has no CIGAR string"};
  int e = R.pos;
  for (size_t i = 0; i < C.op.size(); i++) switch (C.op[i]) {
    case 'N': case 'D': case 'M': e += C.len[i]; break;
    case 'S': case 'H': case 'P': case 'I': break;
    default: throw MergeError{"cannot walk the CIGAR string " + std::string(R.cigar)};
  }
  return e - 1;
}

// one read: ref sam_reader.c:417-716 (pp_ll_combine_and_check) and the print loop of mergesam.c:743-764
void combine_read(const Opts& O, ReadLists& L, Heap& h, std::string& out) {
  Rec* unaligned = nullptr;
  List paired, first_leg, second_leg, unpaired;
  bool have_first = false, have_second = false, set = false;
  if (O.paired) {
    consolidate_paired(O, L, &unaligned, h, paired);
    if (O.o.half_paired) {
      // (when a class is empty everywhere the reference leaves file 0's list as it is -- empty too)
      consolidate_single(O, L, FIRST_LEG, &unaligned, h, first_leg, set); have_first = true;
      consolidate_single(O, L, SECOND_LEG, &unaligned, h, second_leg, set); have_second = true;
      if (O.o.no_mapping_qualities == 0) { /* lists were cleared or left empty */ }
    }
  } else if (O.unpaired) {
    consolidate_single(O, L, UNPAIRED, &unaligned, h, unpaired, set);
  }
  (void)have_first; (void)have_second;
  const bool first_empty = first_leg.empty(), second_empty = second_leg.empty(), paired_empty = paired.empty();
  Rec* best_alignment = nullptr;
  if (!O.o.no_mapping_qualities) {
    const double G = (double)O.genome_length;
    const double paired_scale = (!first_empty ? std::min(first_leg[0]->z[4] * G, 1.0) : 1.0) * (!second_empty ? std::min(second_leg[0]->z[4] * G, 1.0) : 1.0);
    double first_leg_scale = 0.0, second_leg_scale = 0.0;
    if (!first_empty) first_leg_scale = (!paired_empty ? std::min(paired[0]->z[4] * G, 1.0) : 1.0) * (!second_empty ? std::min(second_leg[0]->z[4] * G, 1.0) : 1.0) * first_leg[0]->z[5];
    if (!second_empty) second_leg_scale = (!paired_empty ? std::min(paired[0]->z[4] * G, 1.0) : 1.0) * (!first_empty ? std::min(first_leg[0]->z[4] * G, 1.0) : 1.0) * second_leg[0]->z[5];
    const double class_denom = (!paired_empty ? paired_scale : 0.0) + (!first_empty ? first_leg_scale : 0.0) + (!second_empty ? second_leg_scale : 0.0);
    if (O.unpaired || class_denom > 0) {
      for (Rec* pa : paired) {
        pa->mapq = qv_from_pr_corr((pa->z[2] * paired_scale) / (pa->z[3] * class_denom));
        pa->mate->mapq = qv_from_pr_corr((pa->mate->z[2] * paired_scale) / (pa->mate->z[3] * class_denom));
        Rec* mx = pa->mapq > pa->mate->mapq ? pa : pa->mate;
        if (!best_alignment || mx->mapq > best_alignment->mapq) best_alignment = mx;
      }
      for (Rec* pa : first_leg) { pa->mapq = qv_from_pr_corr((pa->z[0] * first_leg_scale) / (pa->z[1] * class_denom)); if (!best_alignment || pa->mapq > best_alignment->mapq) best_alignment = pa; }
      for (Rec* pa : second_leg) { pa->mapq = qv_from_pr_corr((pa->z[0] * second_leg_scale) / (pa->z[1] * class_denom)); if (!best_alignment || pa->mapq > best_alignment->mapq) best_alignment = pa; }
      for (Rec* pa : unpaired) { pa->mapq = qv_from_pr_corr(pa->z[0] / pa->z[1]); if (!best_alignment || pa->mapq > best_alignment->mapq) best_alignment = pa; }
    }
  }
  List m;
  if (O.o.all_contigs && O.o.single_best && !O.o.no_improper_mappings) {
    if (best_alignment) {
      Rec* ba = best_alignment;
      if (ba->paired_seq && !ba->mp_mapped && ba->mapq >= 10) {               // a half-paired best mapping: pair it up with the best mapping of the other mate
        const List& other = ba->first ? second_leg : first_leg;
        Rec* bp = nullptr;
        for (Rec* pa : other) if (!bp || pa->mapq > bp->mapq) bp = pa;
        if (bp && qv_from_pr_corr(bp->z[0] / bp->z[1]) >= 10) {
          ba->mate = bp; ba->mp_mapped = true; ba->mp_rev = bp->rev; ba->mref = bp->ref; ba->mpos = bp->pos;
          bp->mate = ba; bp->mp_mapped = true; bp->mp_rev = ba->rev; bp->mref = ba->ref; bp->mpos = ba->pos;
          const int e1 = genome_end_unpadded(*ba), e2 = genome_end_unpadded(*bp);
          int isize = 0;                                                      // ref: sam2pretty_lib.c:565-594
          if (ba->ref == bp->ref) { ba->mref = "="; bp->mref = "="; const int f1 = ba->rev ? e1 : ba->pos - 1, f2 = bp->rev ? e2 : bp->pos - 1; isize = f2 - f1; }
          ba->isize = isize; bp->isize = -isize;
        }
      }
      m.push_back(ba);
    }
  } else {
    m.insert(m.end(), paired.begin(), paired.end()); m.insert(m.end(), first_leg.begin(), first_leg.end());
    m.insert(m.end(), second_leg.begin(), second_leg.end()); m.insert(m.end(), unpaired.begin(), unpaired.end());
  }
  if (!m.empty() && O.o.all_contigs && O.o.min_mapq > 0) {
    List keep;
    for (Rec* c : m) {
      const int mq = std::max(c->mapq, c->paired_seq ? c->mate->mapq : 0);
      if (mq < O.o.min_mapq) continue;
      if (c->paired_seq) { if (c->mapq < O.o.min_mapq) c->mapped = false; else if (c->mate->mapq < O.o.min_mapq) c->mate->mapped = false; }
      keep.push_back(c);
    }
    m.swap(keep);
  }
  if (O.o.all_contigs) for (Rec* pa : m) { pa->has_zs = 0; if (pa->mate) pa->mate->has_zs = 0; }
  if (O.o.no_mapping_qualities && !O.o.leave_mapq) for (Rec* pa : m) { pa->mapq = 255; if (pa->mate) pa->mate->mapq = 255; }

  const bool un_file = O.o.output == GM_MERGE_OUT_UNALIGNED_READS, al_file = O.o.output == GM_MERGE_OUT_ALIGNED_READS;
  if (m.empty() && (O.o.sam_unaligned || un_file)) {
    if (!unaligned) {
      if (!O.o.half_paired && O.paired) { List tmp; bool s2; consolidate_single(O, L, FIRST_LEG, &unaligned, h, tmp, s2); consolidate_single(O, L, SECOND_LEG, &unaligned, h, tmp, s2); }
      for (int i = 0; i < O.n_files; i++) { const List& u = L.l[i * N_CLASSES + UNMAPPED]; if (!u.empty()) { unaligned = u[0]; break; } }
    }
    if (unaligned) {
      m.assign(1, unaligned);
      aux_fields(*unaligned); if (un_file) render_fastx(*unaligned); else render_unaligned(*unaligned);
      if (unaligned->paired_seq && unaligned->mate) { aux_fields(*unaligned->mate); if (un_file) render_fastx(*unaligned->mate); else render_unaligned(*unaligned->mate); }
    }
  } else if (!un_file) {
    if (al_file && m.size() > 1) m.resize(1);
    for (Rec* pa : m) {
      if (al_file) { aux_fields(*pa); render_fastx(*pa); } else render_sam(*pa);
      if (pa->mate) { if (al_file) { aux_fields(*pa->mate); render_fastx(*pa->mate); } else render_sam(*pa->mate); }
    }
  } else m.clear();
  for (Rec* pa : m) {
    if (pa->paired_seq) {
      if (pa->first) { out += pa->text; out += '\n'; if (pa->mate) { out += pa->mate->text; out += '\n'; } }
      else { if (pa->mate) { out += pa->mate->text; out += '\n'; } out += pa->text; out += '\n'; }
    } else { out += pa->text; out += '\n'; }
  }
}

// ref: sam_reader.c:719-758 (pp_ll_append_and_check): the class of a record (pair) of one file
void classify(const Opts& O, List* cls, Rec* pa) {
  const bool want_un = O.o.sam_unaligned || O.o.output == GM_MERGE_OUT_UNALIGNED_READS;
  if (pa->paired_seq) {
    if (pa->proper) cls[PAIRED].push_back(pa->first ? pa : pa->mate);
    else if ((O.o.half_paired || want_un) && (pa->mapped || pa->mp_mapped)) {
      if (pa->mapped) cls[pa->first ? FIRST_LEG : SECOND_LEG].push_back(pa);
      else cls[pa->first ? SECOND_LEG : FIRST_LEG].push_back(pa->mate);
    } else if (want_un && !pa->mapped && !pa->mp_mapped) cls[UNMAPPED].push_back(pa);
  } else {
    if (pa->mapped) cls[UNPAIRED].push_back(pa);
    else if (want_un) cls[UNMAPPED].push_back(pa);
  }
}

// read names of a FASTA / FASTQ text, ref: fastx_readnames.c:20-113
void read_names_of(sv text, int fastq, std::vector<std::string>& names) {
  if (fastq < 0) {                                                            // ref: file_buffer.c:25-62
    size_t i = 0;
    while (i < text.size() && (text[i] == '#' || text[i] == ';')) { const size_t nl = text.find('\n', i); if (nl == sv::npos) { i = text.size(); break; } i = nl + 1; }
    fastq = 0;
    if (i < text.size()) { if (text[i] == '@') fastq = 1; else if (text[i] == '>') fastq = 0; else throw MergeError{"reads text: neither FASTA nor FASTQ (first character '" + std::string(1, text[i]) + "')"}; }
  }
  auto add = [&](sv line) {
    // (the reference copies min(254, strlen) characters starting after the marker, then cuts at the first blank or tab)
    std::string n(line.substr(1, std::min<size_t>(254, line.size())));
    const size_t c = n.find_first_of(" \t"); if (c != std::string::npos) n.resize(c);
    names.push_back(std::move(n));
  };
  bool seen_name = false, seen_plus = false, mode_set = false, colour = false; long seq = 0, qual = 0;
  size_t p = 0;
  while (p < text.size()) {
    size_t nl = text.find('\n', p); if (nl == sv::npos) break;               // a last line without a newline is never seen by the reference either
    const sv line = text.substr(p, nl - p); p = nl + 1;
    if (line.empty()) continue;
    if (!fastq) { if (line[0] == '>') add(line); continue; }
    if (!seen_name) { if (line[0] == '@') { add(line); seen_name = true; seq = qual = 0; seen_plus = false; } }
    else if (!seen_plus) {
      if (line[0] == '+') seen_plus = true;
      else { if (!mode_set && line.size() >= 2) { colour = line[1] < 63; mode_set = true; } seq += (long)line.size() - 1; }
    } else { qual += (long)line.size() - 1; if (qual == seq + (colour ? -1 : 0)) seen_name = false; }
  }
}

struct LineRef { uint32_t read; size_t off, len; };

// header comparison, ref: sam_reader.c:807-849
int header_field_cmp(const std::string& a, const std::string& b, const char* f) {
  const bool ah = a[1] == f[0] && a[2] == f[1], bh = b[1] == f[0] && b[2] == f[1];
  if (ah && bh) return a.compare(b) < 0 ? -1 : (a.compare(b) > 0 ? 1 : 0);
  if (ah) return -1; if (bh) return 1; return 0;
}
int header_cmp(const std::string& a, const std::string& b) {
  for (const char* f : {"HD", "SQ", "RG", "PG", "CO"}) { const int r = header_field_cmp(a, b, f); if (r) return r; }
  return 0;
}

}  // namespace

extern "C" void gm_merge_options_default(gm_merge_options_t* o) {
  if (!o) return;
  memset(o, 0, sizeof *o);
  o->max_outputs = 10; o->half_paired = 1; o->fastq = -1; o->threads = 1; o->output = GM_MERGE_OUT_SAM;
}

extern "C" int gm_merge_sam(const gm_merge_options_t* opts, const char* reads_text, size_t reads_len, int n_files, const char* const* sam_text,
                            const size_t* sam_len, char** out, size_t* out_len) {
  if (!opts || !reads_text || n_files < 1 || !sam_text || !sam_len || !out || !out_len) { gm_set_error("gm_merge_sam: bad arguments"); return GM_E_ARG; }
  *out = nullptr; *out_len = 0;
  Opts O; O.o = *opts; O.n_files = n_files; O.genome_length = 0;
  if (O.o.max_outputs <= 0 || O.o.max_alignments < 0) { gm_set_error("gm_merge_sam: max_outputs must be positive, max_alignments non-negative"); return GM_E_ARG; }
  if (O.o.single_best && O.o.no_mapping_qualities) { gm_set_error("gm_merge_sam: single_best cannot be combined with no_mapping_qualities (ref: mergesam.c:555-558)"); return GM_E_ARG; }
  if (O.o.single_best) O.o.max_outputs = 1;                                   // ref: mergesam.c:560-563
  try {
    std::vector<std::string> names;
    read_names_of(sv(reads_text, reads_len), O.o.fastq, names);
    const uint32_t NR = (uint32_t)names.size();
    // ---- sweep 1: every record line to its read (ref: sam_reader.c:893-1031); header lines aside ----
    std::vector<std::vector<LineRef>> lines(n_files);
    std::vector<std::vector<std::string>> headers(n_files);
    std::vector<std::string> sweep_err(n_files);
    auto sweep = [&](int f) {
      const sv T(sam_text[f], sam_len[f]); size_t p = 0; uint32_t cur = 0;
      while (p < T.size()) {
        size_t nl = T.find('\n', p); if (nl == sv::npos) nl = T.size();
        const sv line = T.substr(p, nl - p); const size_t off = p; p = nl + 1;
        if (line.empty()) continue;
        if (line[0] == '@') { headers[f].emplace_back(line); continue; }
        const size_t tab = line.find('\t');
        if (tab == sv::npos || tab == 0) { sweep_err[f] = "SAM text " + std::to_string(f) + ": a record without fields"; return; }
        // the record belongs to the first read, from the current one on, whose name starts with QNAME
        while (cur < NR && !(names[cur].size() >= tab ? memcmp(names[cur].data(), line.data(), tab) == 0 : false)) cur++;
        if (cur == NR) return;                                                // a record of no remaining read: the reference never gets past it, the rest of this text is left out
        lines[f].push_back(LineRef{cur, off, line.size()});
      }
    };
    { std::vector<std::thread> th; std::atomic<int> next{0}; const int nt = std::max(1, std::min(O.o.threads, n_files));
      for (int t = 0; t < nt; t++) th.emplace_back([&] { for (int f; (f = next++) < n_files;) sweep(f); });
      for (auto& t : th) t.join(); }
    for (auto& e : sweep_err) if (!e.empty()) throw MergeError{e};
    // paired or not: the first record of any file decides (ref: sam_reader.c:964-979; mixing is an error, mergesam.c:727-730)
    for (int f = 0; f < n_files; f++) for (const LineRef& lr : lines[f]) {
      const sv line(sam_text[f] + lr.off, lr.len); const size_t t = line.find('\t'); const int fl = sv_atoi(line.substr(t + 1, 8));
      if (fl & 1) O.paired = true; else O.unpaired = true;
    }
    if (O.paired && O.unpaired) throw MergeError{"paired and unpaired records in the same merge"};
    // ---- headers (ref: mergesam.c:84-151) ----
    std::string header;
    { std::vector<std::string> hl; int pg_id = 0;
      for (int f = 0; f < n_files; f++) for (std::string& s : headers[f]) {
        if (s.compare(0, 7, "@PG\tID:") == 0) s = "@PG\tID:" + std::to_string(pg_id++) + "-" + s.substr(7);
        hl.push_back(s);
      }
      if (!hl.empty()) {
        for (const std::string& s : hl) {
          if (s.size() < 4) throw MergeError{"SAM header line too short: " + s};
          if (s.compare(0, 3, "@SQ") == 0) {                                  // every @SQ line of every file counts (ref: mergesam.c:48-82)
            size_t x = 4; for (; x < s.size(); x++) if (s[x] == '\t' && x + 2 < s.size() && s[x + 1] == 'L' && s[x + 2] == 'N') { x++; break; }
            if (x >= s.size() || s[x] != 'L') throw MergeError{"@SQ line without LN: " + s};
            O.genome_length += sv_atoi(sv(s).substr(x + 3));
          }
        }
        std::stable_sort(hl.begin(), hl.end(), [](const std::string& a, const std::string& b) { return header_cmp(a, b) < 0; });
        const std::string self = O.o.command_line ? std::string("@PG\tID:mergesam\tVN:2.2.0\tCL:") + O.o.command_line : std::string();
        bool printed_self = self.empty();
        header += hl[0]; header += '\n';
        if (!O.o.header_given) for (size_t i = 1; i < hl.size(); i++) {
          if (!printed_self && hl[i].compare(0, 3, "@PG") == 0) { header += self; header += '\n'; printed_self = true; }
          if (hl[i] != hl[i - 1]) { header += hl[i]; header += '\n'; }
        }
        if (!printed_self) { header += self; header += '\n'; }
      }
    }
    // ---- sweep 2: blocks of reads, combined and rendered by the thread pool ----
    const uint32_t BLOCK = 2048; const uint32_t nblocks = (NR + BLOCK - 1) / BLOCK;
    std::vector<std::string> block_out(nblocks); std::vector<std::string> block_err(nblocks);
    const int cutoff = O.o.max_alignments == 0 ? O.o.max_outputs : std::min(O.o.max_alignments, O.o.max_outputs);
    auto work = [&](uint32_t b) {
      const uint32_t r0 = b * BLOCK, r1 = std::min(NR, r0 + BLOCK);
      Heap h; h.capacity = cutoff + (O.o.single_best ? 0 : 1); h.a.resize(h.capacity);
      std::vector<std::vector<Rec>> recs(n_files);
      std::vector<ReadLists> RL(r1 - r0); for (auto& x : RL) x.l.resize((size_t)n_files * N_CLASSES);
      try {
        for (int f = 0; f < n_files; f++) {
          const auto& LL = lines[f];
          auto lo = std::lower_bound(LL.begin(), LL.end(), r0, [](const LineRef& a, uint32_t r) { return a.read < r; });
          auto hi = std::lower_bound(LL.begin(), LL.end(), r1, [](const LineRef& a, uint32_t r) { return a.read < r; });
          recs[f].resize(hi - lo);
          Rec* prev = nullptr;                                               // a paired record waiting for its mate (the next record of the file)
          size_t k = 0;
          for (auto it = lo; it != hi; ++it, ++k) {
            Rec& R = recs[f][k];
            if (!parse_record(sv(sam_text[f] + it->off, it->len), R)) throw MergeError{"SAM text " + std::to_string(f) + ": malformed record of read '" + names[it->read] + "'"};
            R.fileno = f;
            if (R.paired_seq) {
              if (prev) { prev->mate = &R; R.mate = prev; prev = nullptr; classify(O, &RL[it->read - r0].l[(size_t)f * N_CLASSES], &R); }
              else prev = &R;
            } else classify(O, &RL[it->read - r0].l[(size_t)f * N_CLASSES], &R);
          }
        }
        std::string& o = block_out[b];
        for (uint32_t r = r0; r < r1; r++) combine_read(O, RL[r - r0], h, o);
      } catch (const MergeError& e) { block_err[b] = e.msg; }
    };
    { std::vector<std::thread> th; std::atomic<uint32_t> next{0}; const int nt = std::max(1, (int)std::min<uint32_t>((uint32_t)std::max(1, O.o.threads), std::max(1u, nblocks)));
      for (int t = 0; t < nt; t++) th.emplace_back([&] { for (uint32_t b; (b = next++) < nblocks;) work(b); });
      for (auto& t : th) t.join(); }
    for (auto& e : block_err) if (!e.empty()) throw MergeError{e};
    size_t total = (O.o.output == GM_MERGE_OUT_SAM ? header.size() : 0); for (auto& s : block_out) total += s.size();
    char* buf = (char*)malloc(total + 1); if (!buf) { gm_set_error("gm_merge_sam: out of memory"); return GM_E_NOMEM; }
    size_t w = 0;
    if (O.o.output == GM_MERGE_OUT_SAM) { memcpy(buf, header.data(), header.size()); w = header.size(); }
    for (auto& s : block_out) { memcpy(buf + w, s.data(), s.size()); w += s.size(); }
    buf[w] = 0; *out = buf; *out_len = w;
    return GM_OK;
  } catch (const MergeError& e) { gm_set_error("gm_merge_sam: %s", e.msg.c_str()); return GM_E_ARG; }
  catch (const std::bad_alloc&) { gm_set_error("gm_merge_sam: out of memory"); return GM_E_NOMEM; }
}
