// ref_kat.cpp -- known-answer generator for the reference's own sw_vector() / sw_full_ls().
// TEST INFRASTRUCTURE ONLY.  Compiled (by oracle/Makefile.ref, only where /root/reference
// exists) against the reference headers where they lie and linked with oracle/_ref/libref_sw.so.
// Writes text records consumed by tools/make_golden.py -> tests/golden/sw_kat.txt.gz
//   V goff glen rlen <genome words hex,...> <read words hex,...> score
//   F goff glen rlen ax ay alen awidth revcmpl <genome words> <read words> score read_start rmapped genome_start gmapped matches mismatches insertions deletions dbalign qralign
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cstdint>
#include <vector>
#include <random>
#include "common/sw-vector.h"
#include "common/sw-full-common.h"
#include "common/sw-full-ls.h"
#include "common/anchors.h"

static void put(std::vector<uint32_t>& bf, int i, int v) { bf[i / 8] |= (uint32_t)(v & 0xf) << (4 * (i % 8)); }
static void dump(const std::vector<uint32_t>& bf) { for (size_t i = 0; i < bf.size(); i++) printf("%s%x", i ? "," : "", bf[i]); }

int main(int argc, char** argv) {
  int n = argc > 1 ? atoi(argv[1]) : 2000;
  const bool local_mode = argc > 2 && !strcmp(argv[2], "local");   // "L" records: sw_full_ls with local_alignment = 1, with and without an anchor
  std::mt19937_64 rng(20260101);
  sw_vector_setup(1400, 1000, -33, -7, -33, -3, 10, -15, 0, true);
  sw_full_ls_setup(1400, 1000, -33, -7, -33, -3, 10, -15, true, 8);
  for (int t = 0; t < n; t++) {
    int rlen = 20 + rng() % 131;                       // 20..150
    int kind = rng() % 8;
    int glen = (kind == 0) ? (int)(rlen - rng() % 10)  // window shorter than the read
                           : (int)(rlen * 1.4);
    if (glen < 8) glen = 8;
    int goff = rng() % 23;
    std::vector<int> g(goff + glen + 9), r(rlen);
    for (auto& b : g) b = rng() % 4;
    // read = mutated copy of a window diagonal, so that scores are meaningful
    int start = goff + (glen > rlen ? rng() % (glen - rlen + 1) : 0);
    int gi = start;
    double psub = (kind == 1) ? 0.0 : (kind == 2 ? 0.15 : 0.03), pind = (kind == 3) ? 0.05 : 0.005;
    for (int i = 0; i < rlen; i++) {
      double u = (rng() % 100000) / 100000.0;
      if (u < pind) { r[i] = rng() % 4; continue; }          // insertion in read
      if (u < 2 * pind) gi += 1 + rng() % 3;                  // deletion from read
      int b = g[gi < (int)g.size() ? gi : (int)g.size() - 1]; gi++;
      if ((rng() % 100000) / 100000.0 < psub) b = (b + 1 + rng() % 3) & 3;
      r[i] = b;
    }
    if (kind == 4) { for (int k = 0; k < 3; k++) { r[rng() % rlen] = 15; g[goff + rng() % glen] = 15; } }   // N codes (N==N matches)
    if (kind == 5) { for (int k = 0; k < 4; k++) g[goff + rng() % glen] = 4 + rng() % 11; }                 // IUPAC codes
    if (kind == 6) { for (auto& b : g) b = 0; for (auto& b : r) b = 0; }                                    // homopolymer: every tie rule fires
    std::vector<uint32_t> gb((goff + glen + 16) / 8 + 1, 0), rb(rlen / 8 + 1, 0);
    for (size_t i = 0; i < g.size(); i++) put(gb, (int)i, g[i]);
    for (int i = 0; i < rlen; i++) put(rb, i, r[i]);
    int sv = sw_vector(gb.data(), goff, glen, rb.data(), rlen, NULL, -1, false);
    if (!local_mode) { printf("V %d %d %d ", goff, glen, rlen); dump(gb); printf(" "); dump(rb); printf(" %d\n", sv); }
    // full SW around a plausible anchor box
    struct anchor a; memset(&a, 0, sizeof a);
    a.x = (start - goff) + (int)(rng() % 7) - 3; a.y = 0; a.length = 14 + rng() % (rlen > 20 ? rlen - 14 : 6); a.width = 1 + rng() % 4; a.weight = 2;
    if (rng() % 4 == 0) { a.y = rng() % 10; a.x += a.y; }
    if (local_mode) {
      const int thresh = (rlen * 10) / 2;
      if (sv < thresh) continue;                      // gmapper calls the full SW only then (mapping.c:390)
      if (t % 3 == 0) { a.width = 1; a.length = 12; }  // thin anchors: the best local alignment leaves the band more often (the second run)
      for (int rv = 0; rv < 2; rv++)
        for (int na = 0; na < 2; na++) {              // with the anchor box, and without (threshold band)
          struct sw_full_results sfr; memset(&sfr, 0, sizeof sfr);
          sw_full_ls(gb.data(), goff, glen, rb.data(), rlen, thresh, sv, &sfr, rv, na ? NULL : &a, na ? 0 : 1, 1);
          if (sfr.score <= 0) continue;
          printf("L %d %d %d %lld %lld %d %d %d %d %d %d ", goff, glen, rlen, a.x, a.y, a.length, a.width, rv, na, thresh, sv); dump(gb); printf(" "); dump(rb);
          printf(" %d %d %d %d %d %d %d %d %d %s %s\n", sfr.score, sfr.read_start, sfr.rmapped, sfr.genome_start, sfr.gmapped,
                 sfr.matches, sfr.mismatches, sfr.insertions, sfr.deletions, sfr.dbalign, sfr.qralign);
          free(sfr.dbalign); free(sfr.qralign);
        }
      continue;
    }
    for (int rv = 0; rv < 2; rv++) {
      struct sw_full_results sfr; memset(&sfr, 0, sizeof sfr);
      sw_full_ls(gb.data(), goff, glen, rb.data(), rlen, 0, sv, &sfr, rv, &a, 1, 0);
      if (sfr.score <= 0) continue;   // reference backtraces from stale scratch in this case; not a defined answer
      printf("F %d %d %d %lld %lld %d %d %d ", goff, glen, rlen, a.x, a.y, a.length, a.width, rv); dump(gb); printf(" "); dump(rb);
      printf(" %d %d %d %d %d %d %d %d %d %s %s\n", sfr.score, sfr.read_start, sfr.rmapped, sfr.genome_start, sfr.gmapped,
             sfr.matches, sfr.mismatches, sfr.insertions, sfr.deletions, sfr.dbalign, sfr.qralign);
      free(sfr.dbalign); free(sfr.qralign);
    }
  }
  return 0;
}
