// ref_kat_cs.cpp -- known-answer generator for the reference's own colour-space kernels: sw_vector() with use_colours
// and sw_full_cs().  TEST INFRASTRUCTURE ONLY.  Compiled (by oracle/Makefile.ref, only where /root/reference exists)
// against the reference headers where they lie and linked with oracle/_ref/libref_sw.so.
// Records (consumed by tools/make_golden.py -> tests/golden/sw_kat_cs.txt.gz); bitfields are hex words, comma separated:
//   C goff glen rlen initbp <genome_cs words> <genome_ls words> <read colour words> score
//   S goff glen rlen initbp ax ay alen awidth revcmpl thresh <genome_ls words> <read colour words>
//     score read_start rmapped genome_start gmapped matches mismatches insertions deletions crossovers dbalign qralign   ("-" when empty)
//   L ...  (second argument "local"): the S record's fields for sw_full_cs in local mode -> tests/golden/sw_kat_cs_local.txt.gz
//   X / Y ...  (second argument "xover"): sw_full_cs with a per-position crossover_score[] (what every csfastq read hands it, ref: mapping.c:375-379, gmapper.c:532-544),
//     global (X) and local (Y) mode; the S record's fields with the rlen scores (comma separated, in [2 * global, -1] as gmapper.c:538-542 clamps them) behind the read words
//     -> tests/golden/sw_kat_cs_xover.txt.gz
//   C / S / L with second argument "rna": the genome holds U for every T (an RNA contig) and both functions get is_rna = true (what gmapper passes for a genome whose last
//     contig is RNA, ref: genome.c:1063-1064, mapping.c:375-388,1318-1327) -- lstocs reads U as T, cstols hands back U for T (util.h:157-205); global and local mode
//     -> tests/golden/sw_kat_cs_rna.txt.gz
// Scores are the binary's colour-space defaults (ref: gmapper-defaults.h:52-58): match 10, mismatch -24, crossover -20,
// gaps -33/-7 (reference) -33/-3 (query); the vector filter's mismatch is match + crossover (ref: gmapper.c:2935).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cstdint>
#include <vector>
#include <random>
#include "common/util.h"
#include "common/sw-vector.h"
#include "common/sw-full-common.h"
#include "common/sw-full-cs.h"
#include "common/anchors.h"

static void put(std::vector<uint32_t>& bf, int i, int v) { bf[i / 8] |= (uint32_t)(v & 0xf) << (4 * (i % 8)); }
static void dump(const std::vector<uint32_t>& bf) { for (size_t i = 0; i < bf.size(); i++) printf("%s%x", i ? "," : "", bf[i]); }

int main(int argc, char** argv) {
  int n = argc > 1 ? atoi(argv[1]) : 1500;
  const bool local = argc > 2 && !strcmp(argv[2], "local");     // "L" records: the same cases through sw_full_cs(.., local_alignment = true) (ref: sw-full-cs.c:199-203,315,439-552); no C / S records
  const bool xover = argc > 2 && !strcmp(argv[2], "xover");     // "X" / "Y" records only
  const bool rna = argc > 2 && !strcmp(argv[2], "rna");         // C, S and L records on an RNA genome with is_rna = true
  std::mt19937_64 rng(20260202), xrng(20261005);
  sw_vector_setup(1400, 1000, -33, -7, -33, -3, 10, 10 + (-20), 1, true);
  sw_full_cs_setup(1400, 1000, -33, -7, -33, -3, 10, -24, -20, true, 8, 0);
  for (int t = 0; t < n; t++) {
    int rlen = 20 + rng() % 56;                        // 20..75 colours
    int kind = rng() % 8;
    int glen = (kind == 0) ? (int)(rlen - rng() % 6) : (int)(rlen * 1.4);
    if (glen < 8) glen = 8;
    int goff = 1 + rng() % 23;
    std::vector<int> g(goff + glen + 9);
    for (auto& b : g) b = rng() % 4;
    if (kind == 4) for (int k = 0; k < 3; k++) g[goff + rng() % glen] = 15;                 // N in the genome
    if (kind == 6) for (auto& b : g) b = rna ? 3 : 0;                                        // homopolymer: every tie rule fires
    if (rna) for (auto& b : g) if (b == 3) b = BASE_U;                                       // an RNA contig: uracil, no thymine
    // letter-space read = mutated copy of a window diagonal
    int start = goff + (glen > rlen ? rng() % (glen - rlen + 1) : 0);
    std::vector<int> rl(rlen);
    int gi = start;
    double psub = (kind == 1) ? 0.0 : (kind == 2 ? 0.10 : 0.02), pind = (kind == 3) ? 0.05 : 0.01;
    for (int i = 0; i < rlen; i++) {
      double u = (rng() % 100000) / 100000.0;
      if (u < pind) { rl[i] = rng() % 4; continue; }
      if (u < 2 * pind) gi += 1 + rng() % 3;
      int b = g[gi < (int)g.size() ? gi : (int)g.size() - 1]; if (b == BASE_U) b = 3; b &= 3; gi++;
      if ((rng() % 100000) / 100000.0 < psub) b = (b + 1 + rng() % 3) & 3;
      rl[i] = b;
    }
    int initbp = rng() % 4;
    std::vector<int> rc(rlen);
    for (int i = 0, last = initbp; i < rlen; i++) { rc[i] = lstocs(last, rl[i], false); last = rl[i]; }
    double pcol = (kind == 1) ? 0.0 : (kind == 5 ? 0.12 : 0.04);                             // colour (sequencing) errors -> crossovers
    for (int i = 0; i < rlen; i++) if ((rng() % 100000) / 100000.0 < pcol) rc[i] = (rc[i] + 1 + rng() % 3) & 3;
    if (kind == 7) for (int k = 0; k < 2; k++) rc[rng() % rlen] = 15;                        // '.' colours
    std::vector<uint32_t> gl((g.size() + 15) / 8 + 1, 0), gc((g.size() + 15) / 8 + 1, 0), rb(rlen / 8 + 1, 0);
    for (size_t i = 0; i < g.size(); i++) { put(gl, (int)i, g[i]); put(gc, (int)i, lstocs(i ? g[i - 1] : BASE_T, g[i], rna)); }   // ref: fasta.c:586-607
    for (int i = 0; i < rlen; i++) put(rb, i, rc[i]);
    int sv = sw_vector(gc.data(), goff, glen, rb.data(), rlen, gl.data(), initbp, rna);
    if (!local && !xover) { printf("C %d %d %d %d ", goff, glen, rlen, initbp); dump(gc); printf(" "); dump(gl); printf(" "); dump(rb); printf(" %d\n", sv); }
    struct anchor a; memset(&a, 0, sizeof a);
    a.x = (start - goff) + (int)(rng() % 7) - 3; a.y = 0; a.length = 10 + rng() % (rlen > 16 ? rlen - 10 : 6); a.width = 1 + rng() % 4; a.weight = 2;
    if (rng() % 4 == 0) { a.y = rng() % 10; a.x += a.y; }
    int thresh = (rng() % 3 == 0) ? (int)(0.6 * rlen * 10) : (int)(0.3 * rlen * 10);
    if (xover) {
      // per-position scores as the read loop derives them from quality values: most positions good (near the global -20 .. -40), some poor (-1 .. -8)
      std::vector<int> xs(rlen);
      for (int i = 0; i < rlen; i++) { const int u = xrng() % 10; xs[i] = u < 6 ? -(int)(14 + xrng() % 27) : (u < 9 ? -(int)(1 + xrng() % 12) : -40); }
      for (int md = 0; md < 2; md++) for (int rv = 0; rv < 2; rv++) {
        if (md == 1 && (t & 1)) continue;                                    // local mode on every other case
        struct sw_full_results sfr; memset(&sfr, 0, sizeof sfr);
        sw_full_cs(gl.data(), goff, glen, rb.data(), rlen, initbp, thresh, &sfr, rv != 0, false, &a, 1, md, xs.data());
        printf("%s %d %d %d %d %lld %lld %d %d %d %d ", md ? "Y" : "X", goff, glen, rlen, initbp, (long long)a.x, (long long)a.y, a.length, a.width, rv, thresh);
        dump(gl); printf(" "); dump(rb); printf(" ");
        for (int i = 0; i < rlen; i++) printf("%s%d", i ? "," : "", xs[i]);
        printf(" %d %d %d %d %d %d %d %d %d %d %s %s\n", sfr.score, sfr.read_start, sfr.rmapped, sfr.genome_start, sfr.gmapped,
               sfr.matches, sfr.mismatches, sfr.insertions, sfr.deletions, sfr.crossovers,
               (sfr.dbalign && sfr.dbalign[0]) ? sfr.dbalign : "-", (sfr.qralign && sfr.qralign[0]) ? sfr.qralign : "-");
        free(sfr.dbalign); free(sfr.qralign);
      }
      continue;
    }
    for (int md = 0; md < (rna ? 2 : 1); md++) for (int rv = 0; rv < 2; rv++) {
      const bool loc = rna ? md == 1 : local;
      struct sw_full_results sfr; memset(&sfr, 0, sizeof sfr);
      sw_full_cs(gl.data(), goff, glen, rb.data(), rlen, initbp, thresh, &sfr, rv != 0, rna, &a, 1, loc ? 1 : 0, NULL);
      printf("%s %d %d %d %d %lld %lld %d %d %d %d ", loc ? "L" : "S", goff, glen, rlen, initbp, (long long)a.x, (long long)a.y, a.length, a.width, rv, thresh);
      dump(gl); printf(" "); dump(rb);
      printf(" %d %d %d %d %d %d %d %d %d %d %s %s\n", sfr.score, sfr.read_start, sfr.rmapped, sfr.genome_start, sfr.gmapped,
             sfr.matches, sfr.mismatches, sfr.insertions, sfr.deletions, sfr.crossovers,
             (sfr.dbalign && sfr.dbalign[0]) ? sfr.dbalign : "-", (sfr.qralign && sfr.qralign[0]) ? sfr.qralign : "-");
      free(sfr.dbalign); free(sfr.qralign);
    }
  }
  return 0;
}
