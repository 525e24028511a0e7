// ref_kat_post.cpp -- known-answer generator for the reference's own post_sw() (S3: the colour-space posterior of one alignment,
// ref: common/sw-post.c:636-758).  TEST INFRASTRUCTURE ONLY.  Compiled (by oracle/Makefile.ref, only where /root/reference exists) against the
// reference headers where they lie and linked with oracle/_ref/libref_sw.so.
// Input (stdin): the "S" records of tests/golden/sw_kat_cs.txt.gz (ref_kat_cs.cpp's format).  Each is run through the reference's sw_full_cs again
// (same scores as ref_kat_cs.cpp) and, when it produced an alignment, through post_sw -- once without quality values (gmapper-cs on csfasta) and,
// for every second record, once with a seeded QV string (csfastq, PHRED+33: use_read_qvs, ref: gmapper.c:2960-2962).
// Output records (consumed by tools/make_golden.py -> tests/golden/sw_kat_post.txt.gz):
//   K pr_mismatch pr_xover pr_del_open pr_del_extend pr_ins_open pr_ins_extend   (the post_sw_setup arguments, %a)
//   P <S-record ordinal> <0|1 use_qvs> <qual string or -> <posterior as %a> matches mismatches crossovers <qralign after> <qual out>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cstdint>
#include <vector>
#include <string>
#include <random>
#include <cmath>
#include "common/util.h"
#include "common/sw-full-common.h"
#include "common/sw-full-cs.h"
#include "common/sw-post.h"
#include "common/anchors.h"

static std::vector<uint32_t> words(const char* s) {
  std::vector<uint32_t> v; const char* p = s;
  while (*p) { v.push_back((uint32_t)strtoul(p, (char**)&p, 16)); if (*p == ',') p++; }
  v.push_back(0); v.push_back(0);
  return v;
}

struct Rec { int goff, glen, rlen, initbp; long long ax, ay; int alen, awidth, rv, thresh; std::vector<uint32_t> gl, rb; };

int main() {
  std::vector<Rec> recs;
  static char line[1 << 20];
  while (fgets(line, sizeof line, stdin)) {
    if (line[0] != 'S') continue;
    Rec r; char g[1 << 16], b[1 << 12];
    if (sscanf(line, "S %d %d %d %d %lld %lld %d %d %d %d %65535s %4095s", &r.goff, &r.glen, &r.rlen, &r.initbp, &r.ax, &r.ay, &r.alen, &r.awidth, &r.rv, &r.thresh, g, b) != 12) return 2;
    r.gl = words(g); r.rb = words(b);
    recs.push_back(r);
  }
  sw_full_cs_setup(1400, 1000, -33, -7, -33, -3, 10, -24, -20, true, 8, 0);
  std::mt19937_64 rng(20261004);
  for (int pass = 0; pass < 2; pass++) {
    // the probabilities gmapper-cs derives from its default scores (ref: gmapper.c:2557-2572; match 10, mismatch -24, crossover -20, gaps -33/-7, -33/-3,
    // pr_xover 0.03), Sanger QVs, offset 0, delta 33 (gmapper.h:78-80); printed as %a so that the test hands the seam the very same doubles
    const double pr_xover = 0.03, alpha = -20.0 / (log(pr_xover / 3) / log(2.0));
    const double pr_mismatch = 1.0 / (1.0 + 1.0 / 3.0 * pow(2.0, (10.0 - (-24.0)) / alpha));
    const double beta = 10.0 - 2 * alpha - alpha * log(1 - pr_mismatch) / log(2.0);
    const double pdo = pow(2.0, -33.0 / alpha), pio = pow(2.0, -33.0 / alpha), pde = pow(2.0, -7.0 / alpha), pie = pow(2.0, (-3.0 - beta) / alpha);
    if (pass == 0) printf("K %a %a %a %a %a %a\n", pr_mismatch, pr_xover, pdo, pde, pio, pie);
    post_sw_setup(1400 + 1000, pr_mismatch, pr_xover, pdo, pde, pio, pie, pass == 1, true, 0, 33, true);
    for (size_t i = pass; i < recs.size(); i += 1 + pass) {
      Rec& r = recs[i];
      struct anchor a; memset(&a, 0, sizeof a);
      a.x = r.ax; a.y = r.ay; a.length = r.alen; a.width = r.awidth; a.weight = 2;
      struct sw_full_results sfr; memset(&sfr, 0, sizeof sfr);
      sw_full_cs(r.gl.data(), r.goff, r.glen, r.rb.data(), r.rlen, r.initbp, r.thresh, &sfr, r.rv != 0, false, &a, 1, 0, NULL);
      if (sfr.score <= 0 || !sfr.dbalign || !sfr.dbalign[0]) { free(sfr.dbalign); free(sfr.qralign); continue; }
      std::string q;
      if (pass == 1) for (int k = 0; k < r.rlen; k++) q.push_back((char)(33 + 2 + rng() % 39));
      post_sw(r.rb.data(), r.initbp, pass == 1 ? (char*)q.c_str() : NULL, &sfr);
      printf("P %zu %d %s %a %d %d %d %s %s\n", i, pass, pass == 1 ? q.c_str() : "-", sfr.posterior, sfr.matches, sfr.mismatches, sfr.crossovers, sfr.qralign, sfr.qual ? sfr.qual : "-");
      free(sfr.dbalign); free(sfr.qralign); free(sfr.qual);
    }
  }
  return 0;
}
