// seam_driver.cpp -- a C++ caller of libgmapper_hip.so through the reference's own (C++-linkage, Itanium-mangled) seam names: what SHRiMP's objects
// reference once sw-vector.o / sw-gapless.o / sw-full-ls.o / sw-full-cs.o / sw-post.o are dropped from its link line (INTEGRATION.md section A;
// ref: common/sw-vector.h:3-6, sw-gapless.h:11-14, sw-full-ls.h:9-13, sw-full-cs.h:7-11, sw-post.h:8-12 -- `extern "C"` is commented out in
// common/util.h:8-10, so the names are _Z9sw_vectorPjiiS_iS_ib, _Z10sw_full_lsPjiiS_iiiP15sw_full_resultsbP6anchorii, _Z7post_swPjiPcP15sw_full_results ...).
// TEST PROGRAM (tests/test_gpu_parity.py compiles it with g++ on the GPU box and feeds it the known-answer records); own code, nothing of the reference.
// The two structs are declared under the reference's names so that the mangled names come out right; their layouts are the public header's
// gm_sw_full_results / gm_anchor (include/gmapper_hip.h), which follow common/sw-full-common.h:13-48 and gmapper-definitions.h:66-74 field for field.
//
// stdin: one request per line (hex words comma separated); stdout: one answer per line.
//   setup_v <colours 0|1> <mismatch>          sw_vector_setup(1400, 1000, -33, -7, -33, -3, 10, mismatch, colours, true)
//   setup_f_ls | setup_f_cs                   sw_full_ls_setup(..., 10, -15, true, 8) | sw_full_cs_setup(..., 10, -24, -20, true, 8, 0)
//   setup_g <match> <mismatch>                sw_gapless_setup
//   setup_p <use_qvs> <6 doubles, %a>         post_sw_setup(2400, ..., use_qvs, true, 0, 33, true)
//   V goff glen rlen <genome> <read>                                   -> V score
//   C goff glen rlen initbp <genome_cs> <genome_ls> <read>             -> C score
//   G glen rlen g_idx r_idx init_bp <genome> <read> <genome_ls | ->   -> G score
//   F goff glen rlen ax ay alen awidth rv <genome> <read>              -> F score read_start rmapped genome_start gmapped matches mismatches insertions deletions dbalign qralign
//   S goff glen rlen initbp ax ay alen awidth rv thresh <genome_ls> <read> -> S (the same ten + crossovers) dbalign qralign   ("-" when empty)
//   L ... (the S request's fields)                                        -> L ...: sw_full_cs with local_alignment = true
//   P <qual | -> goff glen rlen initbp ax ay alen awidth rv thresh <genome_ls> <read> -> P posterior(%a) matches mismatches crossovers qralign qual   (sw_full_cs, then post_sw)
//   stats                                                              -> stats <invocations of sw_vector, sw_gapless, sw_full_ls, sw_full_cs, post_sw>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cstdint>
#include <string>
#include <vector>
#include <sstream>
#include <iostream>

struct anchor { long long x, y; int length, width, weight, cn, score; };
struct sw_full_results {
  int read_start, rmapped, genome_start, gmapped, matches, mismatches, insertions, deletions, score;
  int posterior_score, pct_posterior_score;
  char *dbalign, *qralign, *qual;
  double posterior;
  int mqv; double z0, z1, z2, z3, pr_top_random_at_location, pr_missed_mp, insert_size_denom;
  int crossovers;
  bool dup, in_use;
};

// C++ linkage on purpose: these resolve to the mangled exports of the library (shrimp_amd/csrc/gm_cxx_shims.cpp)
int  sw_vector_setup(int, int, int, int, int, int, int, int, int, bool);
int  sw_vector(uint32_t*, int, int, uint32_t*, int, uint32_t*, int, bool);
void sw_vector_stats(uint64_t*, uint64_t*, double*);
int  sw_gapless_setup(int, int, bool);
int  sw_gapless(uint32_t*, int, uint32_t*, int, int, int, uint32_t*, int, bool);
void sw_gapless_stats(uint64_t*, uint64_t*, uint64_t*);
int  sw_full_ls_setup(int, int, int, int, int, int, int, int, bool, int);
void sw_full_ls(uint32_t*, int, int, uint32_t*, int, int, int, struct sw_full_results*, bool, struct anchor*, int, int);
void sw_full_ls_stats(uint64_t*, uint64_t*, double*);
int  sw_full_cs_setup(int, int, int, int, int, int, int, int, int, bool, int, int);
void sw_full_cs(uint32_t*, int, int, uint32_t*, int, int, int, struct sw_full_results*, bool, bool, struct anchor*, int, int, int*);
void sw_full_cs_stats(uint64_t*, uint64_t*, double*);
int  post_sw_setup(int, double, double, double, double, double, double, bool, bool, int, int, bool);
void post_sw(uint32_t*, int, char*, struct sw_full_results*);
int  post_sw_stats(uint64_t*, uint64_t*, double*);

static std::vector<uint32_t> words(const std::string& t) {
  std::vector<uint32_t> v; const char* p = t.c_str();
  while (*p) { char* e; v.push_back((uint32_t)strtoul(p, &e, 16)); p = e; if (*p == ',') p++; }
  v.push_back(0); v.push_back(0);
  return v;
}
static const char* str(const char* s) { return (s && s[0]) ? s : "-"; }

static bool g_is_rna = false;
int main() {
  std::string line;
  while (std::getline(std::cin, line)) {
    std::istringstream in(line); std::string op; in >> op;
    if (op == "rna") { int v; in >> v; g_is_rna = v != 0; }      // the is_rna argument of the calls that follow (sw_vector / sw_gapless / sw_full_cs)
    else if (op == "setup_v") { int col, mm; in >> col >> mm; if (sw_vector_setup(1400, 1000, -33, -7, -33, -3, 10, mm, col, true)) return 3; }
    else if (op == "setup_f_ls") { if (sw_full_ls_setup(1400, 1000, -33, -7, -33, -3, 10, -15, true, 8)) return 3; }
    else if (op == "setup_f_cs") { if (sw_full_cs_setup(1400, 1000, -33, -7, -33, -3, 10, -24, -20, true, 8, 0)) return 3; }
    else if (op == "setup_g") { int m, mm; in >> m >> mm; if (sw_gapless_setup(m, mm, true)) return 3; }
    else if (op == "setup_p") {
      int q; std::string k[6]; in >> q; for (auto& s : k) in >> s;
      double d[6]; for (int i = 0; i < 6; i++) d[i] = strtod(k[i].c_str(), nullptr);
      post_sw_setup(1400 + 1000, d[0], d[1], d[2], d[3], d[4], d[5], q != 0, true, 0, 33, true);
    } else if (op == "V") {
      int goff, glen, rlen; std::string g, r; in >> goff >> glen >> rlen >> g >> r;
      auto gw = words(g), rw = words(r);
      printf("V %d\n", sw_vector(gw.data(), goff, glen, rw.data(), rlen, nullptr, -1, false));
    } else if (op == "C") {
      int goff, glen, rlen, ib; std::string gc, gl, r; in >> goff >> glen >> rlen >> ib >> gc >> gl >> r;
      auto gcw = words(gc), glw = words(gl), rw = words(r);
      printf("C %d\n", sw_vector(gcw.data(), goff, glen, rw.data(), rlen, glw.data(), ib, g_is_rna));
    } else if (op == "G") {
      int glen, rlen, gi, ri, ib; std::string g, r, gl; in >> glen >> rlen >> gi >> ri >> ib >> g >> r >> gl;
      auto gw = words(g), rw = words(r); std::vector<uint32_t> glw; if (gl != "-") glw = words(gl);
      printf("G %d\n", sw_gapless(gw.data(), glen, rw.data(), rlen, gi, ri, gl != "-" ? glw.data() : nullptr, ib, g_is_rna));
    } else if (op == "F") {
      int goff, glen, rlen, rv; struct anchor a; memset(&a, 0, sizeof a); std::string g, r;
      in >> goff >> glen >> rlen >> a.x >> a.y >> a.length >> a.width >> rv >> g >> r; a.weight = 2;
      auto gw = words(g), rw = words(r);
      const int sv = sw_vector(gw.data(), goff, glen, rw.data(), rlen, nullptr, -1, false);       // maxscore, as hit_run_full_sw hands it over (mapping.c:390-398)
      struct sw_full_results f; memset(&f, 0, sizeof f);
      sw_full_ls(gw.data(), goff, glen, rw.data(), rlen, 0, sv, &f, rv != 0, &a, 1, 0);
      printf("F %d %d %d %d %d %d %d %d %d %s %s\n", f.score, f.read_start, f.rmapped, f.genome_start, f.gmapped, f.matches, f.mismatches, f.insertions, f.deletions,
             str(f.dbalign), str(f.qralign));
      free(f.dbalign); free(f.qralign);
    } else if (op == "S" || op == "P" || op == "L" || op == "X" || op == "Y") {      // X / Y: crossover_score[] behind the read (global / local mode)
      std::string q; if (op == "P") in >> q;
      int goff, glen, rlen, ib, rv, thresh; struct anchor a; memset(&a, 0, sizeof a); std::string gl, r, xs;
      in >> goff >> glen >> rlen >> ib >> a.x >> a.y >> a.length >> a.width >> rv >> thresh >> gl >> r; a.weight = 2;
      std::vector<int> xv;
      if (op == "X" || op == "Y") { in >> xs; const char* p = xs.c_str(); while (*p) { char* e; xv.push_back((int)strtol(p, &e, 10)); p = e; if (*p == ',') p++; } }
      auto glw = words(gl), rw = words(r);
      struct sw_full_results f; memset(&f, 0, sizeof f);
      sw_full_cs(glw.data(), goff, glen, rw.data(), rlen, ib, thresh, &f, rv != 0, g_is_rna, &a, 1, (op == "L" || op == "Y") ? 1 : 0, xv.empty() ? nullptr : xv.data());
      if (op != "P") {
        printf("%s %d %d %d %d %d %d %d %d %d %d %s %s\n", op.c_str(), f.score, f.read_start, f.rmapped, f.genome_start, f.gmapped, f.matches, f.mismatches, f.insertions, f.deletions,
               f.crossovers, str(f.dbalign), str(f.qralign));
      } else {
        if (f.score <= 0 || !f.dbalign || !f.dbalign[0]) { printf("P none\n"); }
        else {
          std::vector<char> qb(q.begin(), q.end()); qb.push_back(0);
          post_sw(rw.data(), ib, q == "-" ? nullptr : qb.data(), &f);
          printf("P %a %d %d %d %s %s\n", f.posterior, f.matches, f.mismatches, f.crossovers, str(f.qralign), str(f.qual));
        }
      }
      free(f.dbalign); free(f.qralign); free(f.qual);
    } else if (op == "stats") {
      uint64_t n[5] = {0, 0, 0, 0, 0}, c = 0, t = 0; double s = 0;
      sw_vector_stats(&n[0], &c, &s); sw_gapless_stats(&n[1], &c, &t); sw_full_ls_stats(&n[2], &c, &s); sw_full_cs_stats(&n[3], &c, &s); post_sw_stats(&n[4], &c, &s);
      printf("stats %llu %llu %llu %llu %llu\n", (unsigned long long)n[0], (unsigned long long)n[1], (unsigned long long)n[2], (unsigned long long)n[3], (unsigned long long)n[4]);
    } else if (!op.empty()) { fprintf(stderr, "seam_driver: unknown request %s\n", op.c_str()); return 2; }
  }
  return 0;
}
