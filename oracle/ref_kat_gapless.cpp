// ref_kat_gapless.cpp -- known-answer generator for the reference's own sw_gapless() (S1's ungapped filter, ref: common/sw-gapless.c:57-117,
// sw-gapless.h:11-14), in letter space and in colour space (first colour forced against lstocs(genome_ls, init_bp), :84-94).
// TEST INFRASTRUCTURE ONLY.  Compiled (by oracle/Makefile.ref, only where /root/reference exists) against the reference headers where they lie
// and linked with oracle/_ref/libref_sw.so.  Records (tools/make_golden.py -> tests/golden/sw_kat_gapless.txt.gz); bitfields are hex words:
//   G glen rlen g_idx r_idx init_bp <genome words> <read words> <genome_ls words or -> score        (init_bp -1: letter space, genome_ls NULL)
//   second argument "rna": colour-space records only, the genome holds U for every T and sw_gapless gets is_rna = true (lstocs reads U as T, ref: sw-gapless.c:84, util.h:182-205)
//   -> tests/golden/sw_kat_gapless_rna.txt.gz
// Scores: match 10, mismatch -15 in letter space; match 10, mismatch -24 in colour space (what f1_setup hands over, ref: f1-wrapper.h:66-68).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cstdint>
#include <vector>
#include <random>
#include "common/util.h"
#include "common/sw-gapless.h"

static void put(std::vector<uint32_t>& bf, int i, int v) { bf[i / 8] |= (uint32_t)(v & 0xf) << (4 * (i % 8)); }
static void dump(const std::vector<uint32_t>& bf) { for (size_t i = 0; i < bf.size(); i++) printf("%s%x", i ? "," : "", bf[i]); }

int main(int argc, char** argv) {
  int n = argc > 1 ? atoi(argv[1]) : 1200;
  const bool rna = argc > 2 && !strcmp(argv[2], "rna");
  std::mt19937_64 rng(20261005);
  for (int cs = rna ? 1 : 0; cs < 2; cs++) {
    sw_gapless_setup(10, cs ? -24 : -15, true);
    for (int t = 0; t < n; t++) {
      const int rlen = 12 + rng() % 140, glen = 30 + rng() % 400, kind = rng() % 8;
      std::vector<int> g(glen + 1), r(rlen);
      for (auto& b : g) b = rng() % 4;
      if (kind == 5) for (auto& b : g) b = rna ? 3 : 0;                        // homopolymer
      if (rna) for (auto& b : g) if (b == 3) b = BASE_U;                       // an RNA contig
      if (kind == 6) for (int k = 0; k < 4; k++) g[rng() % glen] = 15;         // N
      // the diagonal through (g_idx, r_idx): inside the contig, hanging over its start, or over its end
      int r_idx = rng() % rlen, g_idx;
      if (kind == 0) g_idx = rng() % (r_idx + 1);                              // g_idx <= r_idx: the read starts before the contig
      else if (kind == 1) g_idx = glen - 1 - (int)(rng() % 10);                // runs off the contig's end
      else g_idx = r_idx + (int)(rng() % (glen > rlen ? glen - rlen + 1 : 1));
      if (g_idx >= glen) g_idx = glen - 1;
      // the read's letters follow the genome along that diagonal, with substitutions
      const double psub = kind == 2 ? 0.0 : (kind == 3 ? 0.25 : 0.05);
      for (int i = 0; i < rlen; i++) {
        const int gi = g_idx - r_idx + i;
        int b = (gi >= 0 && gi < glen) ? ((g[gi] == BASE_U ? 3 : g[gi]) & 3) : (int)(rng() % 4);
        if ((rng() % 100000) / 100000.0 < psub) b = (b + 1 + rng() % 3) & 3;
        r[i] = b;
      }
      std::vector<uint32_t> gl(glen / 8 + 2, 0), gc(glen / 8 + 2, 0), rb(rlen / 8 + 2, 0);
      for (int i = 0; i < glen; i++) { put(gl, i, g[i]); put(gc, i, lstocs(i ? g[i - 1] : BASE_T, g[i], rna)); }
      int init_bp = -1;
      if (!cs) { for (int i = 0; i < rlen; i++) put(rb, i, r[i]); if (kind == 6) put(rb, (int)(rng() % rlen), 15); }
      else {
        // colours of the read; its first colour hangs on the primer letter.  Half of the diagonals start at read position 0 (the forced first colour)
        init_bp = rng() % 4;
        if (t % 2 == 0) { g_idx -= r_idx; r_idx = 0; if (g_idx < 0) g_idx = 0; }
        const int gi0 = g_idx - r_idx;
        const int before0 = (gi0 > 0 && gi0 - 1 < glen) ? g[gi0 - 1] : BASE_T; const int before = before0 == BASE_U ? 3 : before0;
        std::vector<int> rc(rlen);
        for (int i = 0, last = (t % 4 == 0) ? init_bp : before; i < rlen; i++) { rc[i] = lstocs(last, r[i], false); last = r[i]; }
        if (t % 4 == 0 && r_idx == 0 && (t & 8)) { const int gl0 = g[g_idx > 0 ? g_idx : 0]; init_bp = ((gl0 == BASE_U ? 3 : gl0) ^ rc[0]) & 3; }     // some first colours that do match lstocs(letter, primer)
        for (int i = 0; i < rlen; i++) if ((rng() % 100) < 4) rc[i] = (rc[i] + 1 + rng() % 3) & 3;
        if (kind == 7) rc[rng() % rlen] = 15;
        for (int i = 0; i < rlen; i++) put(rb, i, rc[i]);
      }
      const int sc = sw_gapless(cs ? gc.data() : gl.data(), glen, rb.data(), rlen, g_idx, r_idx, cs ? gl.data() : NULL, init_bp, rna);
      printf("G %d %d %d %d %d ", glen, rlen, g_idx, r_idx, init_bp); dump(cs ? gc : gl); printf(" "); dump(rb); printf(" ");
      if (cs) dump(gl); else printf("-");
      printf(" %d\n", sc);
    }
  }
  uint64_t inv = 0, cells = 0, ticks = 0;
  sw_gapless_stats(&inv, &cells, &ticks);
  fprintf(stderr, "sw_gapless_stats (colour-space half): %llu invocations, %llu cells\n", (unsigned long long)inv, (unsigned long long)cells);
  return 0;
}
