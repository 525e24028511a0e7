// gm_lookup.hip -- K1: spaced-seed lookup + exact region filter.  HBM-bound; this is the kernel
// priced against the HBM roofline (B_seed = sum over lookups of 12 + 4*len bytes).
//
// Replaces, for one read-strand per workgroup:
//   read_get_mapidxs_per_strand   ref: gmapper/mapping.c:37-70     k-mer -> map index
//   read_get_region_counts        ref: gmapper/mapping.c:459-542   ">= 2 k-mer hits in a 2048(+50) bp region"
//   advance_index_in_genomemap    ref: gmapper/mapping.c:646-805   per-entry survival test (unpaired branch :731-743)
// The reference walks every inverted list twice through a 4 MB per-thread region map in DRAM
// (latency bound).  Here the genome is cut into slabs of 2^slab_bits positions; per slab the
// region counters (2 bits each) live in LDS, the list slices of that slab are streamed from HBM
// once (coalesced: consecutive lanes read consecutive entries of the flattened slice set) and
// re-read from L2 for the survival test.  Output: the surviving (position, list id) pairs.
//
// Exactness: a region's count is the number of list entries e with region(e) == r, plus those in
// the first `region_overlap` bases of region r+1 (ref :521-533); an entry survives iff its region,
// or region-1 when it lies in that strip, has count >= 2 (ref :733-742).  Counts do not depend on
// the visiting order, so the slab sweep reproduces them exactly; slab borders are handled by
// reading the (rare) entries of the neighbouring slices that can mark a border region.
#include <algorithm>
#include "gm_common.h"
#include "gm_internal.h"

#define K1_THREADS 256
#define K1_SHORT 64u      // slices up to this many entries are streamed by their own lane
#ifndef K1_UNROLL
#define K1_UNROLL 8       // positions in flight per lane (two dwordx4 per 8)
#endif

struct K1Smem {
  uint32_t n_surv;
  uint32_t total;
};

__device__ __forceinline__ void k1_mark(uint32_t* bm, uint32_t rloc) {
  const uint32_t w = rloc >> 4, sh = (rloc & 15u) * 2u;
  const uint32_t old = atomicOr(&bm[w], 1u << sh);
  if (old & (1u << sh)) atomicOr(&bm[w], 2u << sh);
}
// 8 consecutive positions with 4-byte alignment: two global_load_dwordx4 (gfx950 runs in unaligned-access mode)
typedef uint32_t k1_u32x4 __attribute__((ext_vector_type(4), aligned(4)));
__device__ __forceinline__ void k1_load8(const uint32_t* __restrict__ src, uint32_t* p) {
  k1_u32x4 v[K1_UNROLL / 4];
#pragma unroll
  for (int q = 0; q < K1_UNROLL / 4; q++) v[q] = *(const k1_u32x4*)(src + 4 * q);
#pragma unroll
  for (int q = 0; q < K1_UNROLL / 4; q++) { p[4 * q] = v[q].x; p[4 * q + 1] = v[q].y; p[4 * q + 2] = v[q].z; p[4 * q + 3] = v[q].w; }
}
__device__ __forceinline__ bool k1_has2(const uint32_t* bm, uint32_t rloc) {
  return (bm[rloc >> 4] >> ((rloc & 15u) * 2u + 1u)) & 1u;
}

// dynamic LDS layout: codes[read_len pad 4] | kS[NL] | lbeg[NL] | lend[NL] | lo[NL] | hi[NL] | pre[NL+1] | bitmap[bm_words]
// MP (paired -n 3): the modes of GmMpDev -- 1 lists the regions this read-strand marks twice, 2 / 4 / 5 widen or narrow the survival rule by the mate's rows, 3 flags the
// mate's rows; 6: GmIndexDev.no_region_counts, every list entry survives.  (A template parameter, not a run-time switch: one more live scalar in the default
// instantiation tips its SGPR spills into scratch memory, and a kernel with scratch takes 0.95 ms to dispatch instead of 0.07 -- measured on the fall-back launch.)
template <bool BKT, int MP = 0>
__device__ __forceinline__ void
k_lookup_body(const int item, GmIndexDev ix, const uint32_t* __restrict__ reads, int n_reads, int read_len, int read_words, int max_n_kmers,
         int NL, int bm_words, uint64_t* __restrict__ surv, uint32_t* __restrict__ surv_cnt, int scap_all,
         uint32_t* __restrict__ heavy_list, uint32_t* __restrict__ heavy_cnt, int heavy_cap,
         const uint32_t* __restrict__ redo_list, const uint64_t* __restrict__ redo_off,
         const uint32_t* __restrict__ fb_list, const uint32_t* __restrict__ fb_cnt,     // list mode: block b runs read-strand fb_list[b] (b < *fb_cnt) as in normal mode
         unsigned long long* __restrict__ stats, int ablate, uint32_t* __restrict__ surv_seg) {
  extern __shared__ __align__(16) uint32_t smem[];
  __shared__ K1Smem sh;
  const int tid = threadIdx.x;
  const int nthr = blockDim.x;       // 256, or NL rounded up to a wave in bucket mode (one list per thread)
  // normal mode: block b = read-strand b, output slot surv[b*scap_all .. +scap_all).
  // redo mode (redo_list != 0): block b re-runs heavy read-strand redo_list[b] into surv[redo_off[b] .. redo_off[b+1])
  const bool redo = (redo_list != nullptr);
  const int rs = redo ? (int)redo_list[item] : (fb_cnt ? (int)fb_list[item] : item);
  const int rd = rs >> 1, st = rs & 1;
  uint64_t* out = redo ? (surv + redo_off[item]) : (surv + (size_t)rs * scap_all);
  const uint32_t scap = redo ? (uint32_t)(redo_off[item + 1] - redo_off[item]) : (uint32_t)scap_all;
  uint8_t* codes = (uint8_t*)smem;
  const int code_words = (read_len + 3) / 4;
  uint32_t* kS = smem + code_words;
  uint32_t* lbeg = kS + NL;
  uint32_t* lend = lbeg + NL;
  uint32_t* lo = lend + NL;
  uint32_t* hi = lo + NL;
  uint32_t* pre = hi + NL;              // NL + 1
  uint32_t* bm = smem + ((code_words + 6 * NL + 1 + 3) & ~3);     // 16-byte aligned for the b128 clears
  const int S = ix.n_slabs, rb = ix.region_bits;
  const uint32_t rmask = (1u << rb) - 1u, ovl = (uint32_t)ix.region_overlap;
  uint32_t* const sh_mp_row = bm + bm_words;             // (dynamic LDS behind the region counters; MP launches only) MP 1: the row being collected; 2, 3, 4: the mate's row
  uint32_t* const sh_own_row = sh_mp_row + GM_MP_CAP;     // MP 4: this read-strand's own row, with the flags the mate's pass left
  __shared__ uint32_t sh_mp_n, sh_own_n;

  // ---- 0. read codes of this strand (strand 1 = reverse complement, ref: util.c:540-596) ----
  const uint32_t* rw = reads + (size_t)rd * read_words;
  for (int i = tid; i < read_len; i += nthr) {
    codes[i] = (uint8_t)gm_read_code(rw, read_len, st ^ ix.cs_flip, ix.colour, i, ix.read_rna && ix.read_rna[rd]);   // strand 1 = reverse complement (ref: util.c:540-617)
  }
  if (tid == 0) sh.n_surv = 0;
  if (MP == 1 && tid == 0) sh_mp_n = 0;
  if (MP >= 2 && MP <= 5) {   // the mate's row (other strand of the same pair), held in LDS for the reach tests
    const uint32_t mn = ix.mp.cnt[rs ^ 1];
    if (tid == 0) { sh_mp_n = mn <= (uint32_t)GM_MP_CAP ? mn : 0u; if (mn > (uint32_t)GM_MP_CAP && !redo && MP != 3) GS_ADD(stats, GS_MP_UNFILTERED, 1ull); }
    if (mn <= (uint32_t)GM_MP_CAP) for (uint32_t i = tid; i < mn; i += nthr) sh_mp_row[i] = ix.mp.rows[(size_t)(rs ^ 1) * GM_MP_CAP + i] & ~GM_MP_FLAG;
  }
  if (MP == 4) {
    const uint32_t on = ix.mp.out_cnt[rs];
    if (tid == 0) { sh_own_n = on <= (uint32_t)GM_MP_CAP ? on : 0u; if (on > (uint32_t)GM_MP_CAP && !redo) GS_ADD(stats, GS_MP_UNFILTERED, 1ull); }
    if (on <= (uint32_t)GM_MP_CAP) for (uint32_t i = tid; i < on; i += nthr) sh_own_row[i] = ix.mp.out_rows[(size_t)rs * GM_MP_CAP + i];
  }
  __syncthreads();
  const int mp_dmin = (MP >= 2 && MP <= 5) ? ix.mp.dmin[st] : 0, mp_dmax = (MP >= 2 && MP <= 5) ? ix.mp.dmax[st] : 0;
  // count_mp >= 2 for region reg: the mate marked a region twice within [reg + dmin, reg + dmax] (ref: mapping.c:573-582)
  auto mp_reach = [&](uint32_t reg) __attribute__((always_inline)) -> bool { return (MP == 2 || MP == 4 || MP == 5) && gm_mp_reach(sh_mp_row, sh_mp_n, (long long)reg + mp_dmin, (long long)reg + mp_dmax); };
  // MP 3: this read-strand marks reg -- every region X the mate (strand 1 - st) marked twice that reaches it, X + mate_dmin <= reg <= X + mate_dmax, has count_mp >= 1
  const int mt_dmin = MP == 3 ? ix.mp.mate_dmin[1 - st] : 0, mt_dmax = MP == 3 ? ix.mp.mate_dmax[1 - st] : 0;
  auto mp_flag = [&](uint32_t reg) __attribute__((always_inline)) {
    uint32_t a;
    if (gm_mp_reach(sh_mp_row, sh_mp_n, (long long)reg - mt_dmax, (long long)reg - mt_dmin, &a))
      for (; a < sh_mp_n && (long long)sh_mp_row[a] <= (long long)reg - mt_dmin; a++) atomicOr(&ix.mp.rows[(size_t)(rs ^ 1) * GM_MP_CAP + a], GM_MP_FLAG);
  };
  // MP 4: count_mp >= 1 && count_main + count_mp >= 3 (ref: mapping.c:733-742)
  auto mp_ok3 = [&](uint32_t reg, uint32_t rloc) __attribute__((always_inline)) -> bool {
    if (mp_reach(reg)) return true;
    if (!k1_has2(bm, rloc)) return false;
    uint32_t a;
    return gm_mp_reach(sh_own_row, sh_own_n, (long long)reg, (long long)reg, &a) && (sh_own_row[a] & GM_MP_FLAG) != 0u;
  };
  // does the list entry survive?  Default: its region, or the one before it when the entry lies in the overlap strip, was marked twice (ref: mapping.c:733-777)
  auto keep = [&](uint32_t reg, uint32_t rloc, bool strip) __attribute__((always_inline)) -> bool {   // (inlined: a call would put the captures in scratch memory, and a kernel
                                                                                                       // with scratch takes 0.9 ms to dispatch instead of 0.07)
    if (MP == 3) { mp_flag(reg); if (strip) mp_flag(reg - 1u); return false; }
    if (MP == 4) return mp_ok3(reg, rloc) || (strip && mp_ok3(reg - 1u, rloc - 1u));
    if (MP == 5) return (k1_has2(bm, rloc) && mp_reach(reg)) || (strip && k1_has2(bm, rloc - 1u) && mp_reach(reg - 1u));
    if (MP == 6) return true;                      // no region counts (-n 1; paired -n 2): every list entry survives (the marks are still made, nothing reads them)
    return k1_has2(bm, rloc) || mp_reach(reg) || (strip && (k1_has2(bm, rloc - 1u) || mp_reach(reg - 1u)));
  };

  // ---- 1. map indexes + whole-list bounds (ref: mapping.c:53-66, KMER_TO_MAPIDX gmapper.h:349-368) ----
  // BKT (small genomes, one slab, one list per thread): the probe is ONE 64-byte bucket per k-mer holding the
  // list length and its first 15 positions, so a lookup costs one HBM sector instead of directory + list.
  unsigned long long my_lookups = 0, my_entries = 0;
  uint32_t bp[16];                        // BKT: bp[0] = list length, bp[1..15] = first positions (registers)
  uint32_t blen = 0;                      // BKT: entries held in registers (0 when the list is long or skipped)
#pragma unroll
  for (int q = 0; q < 16; q++) bp[q] = 0;
  for (int off = tid; off < NL; off += (BKT ? NL : nthr)) {     // bucket mode: exactly one list per thread
    const int sn = off / max_n_kmers, i = off - sn * max_n_kmers;
    uint32_t k = 0, b = 0, e = 0;
    const int span = ix.seed[sn].span;
    if (i >= ix.colour && i + span <= read_len) {           // colour space: min_kmer_pos = 1 (ref: gmapper.c:477-480)
      const uint64_t mask = ix.seed[sn].mask;
      const uint32_t mapidx = gm_mapidx(ix, mask, span, codes + i);
      k = mapidx * (uint32_t)S;
      my_lookups++;
      if (BKT) {
        const uint4* bk = (const uint4*)(ix.seed[sn].bkt + (size_t)mapidx * 16);
        const uint4 q0 = bk[0], q1 = bk[1], q2 = bk[2], q3 = bk[3];
        bp[0] = q0.x; bp[1] = q0.y; bp[2] = q0.z; bp[3] = q0.w; bp[4] = q1.x; bp[5] = q1.y; bp[6] = q1.z; bp[7] = q1.w;
        bp[8] = q2.x; bp[9] = q2.y; bp[10] = q2.z; bp[11] = q2.w; bp[12] = q3.x; bp[13] = q3.y; bp[14] = q3.z; bp[15] = q3.w;
        const uint32_t len = bp[0];
        if (len <= ix.list_cutoff) {        // ref: mapping.c:497 (longer lists are skipped, not deleted)
          my_entries += len;
          if (len <= 15u) blen = len;
          else { const uint32_t* dir = ix.seed[sn].dir; b = dir[k]; e = dir[k + 1]; }   // rare: stream the list itself
        }
      } else {
        const uint32_t* dir = ix.seed[sn].dir;
        b = dir[k]; e = dir[k + S];
        if (e - b > ix.list_cutoff) { b = 0; e = 0; }    // ref: mapping.c:497 (skipped, not deleted)
        my_entries += (e - b);
      }
    }
    kS[off] = k; lbeg[off] = b; lend[off] = e;
  }
  __syncthreads();

  // ---- 2. slab sweep ----
  // Short slices (<= K1_SHORT entries; all of them on uniform genomes) are streamed by the lane that
  // owns the list: its bounds stay in registers and K1_UNROLL loads are in flight per lane.  Long
  // slices (repeats) are flattened over the whole workgroup so that one lane never walks a long list.
  auto emit = [&](uint32_t p, int off) {
    const uint32_t slot = atomicAdd(&sh.n_surv, 1u);
    if (slot < scap) {   // sort key of K2: position, then read offset y, then seed
      const uint32_t sn = (uint32_t)off / (uint32_t)max_n_kmers, y = (uint32_t)off - sn * (uint32_t)max_n_kmers;
      out[slot] = ((uint64_t)p << 32) | ((uint64_t)y << 16) | sn;
    }
  };
  for (int s = 0; s < S; s++) {
    const uint64_t B = (uint64_t)s << ix.slab_bits;
    const uint64_t E = B + (1ull << ix.slab_bits);
    const uint32_t rbase = (uint32_t)(B >> rb);           // local region index = region - rbase + 1
    const uint32_t rend = (uint32_t)(E >> rb);
    {   // clear the region counters (16-byte stores)
      uint4* bm4 = (uint4*)bm; const int n4 = GM_ABL(8) ? 0 : (bm_words >> 2);
      for (int w = tid; w < n4; w += nthr) bm4[w] = make_uint4(0, 0, 0, 0);
      for (int w = (n4 << 2) + tid; w < bm_words; w += nthr) bm[w] = 0;
    }
    if (tid == 0) sh.total = 0;
    __syncthreads();
    // -- phase 0: mark --
    bool any_long = false;
    if (BKT) {
#pragma unroll
      for (int u = 0; u < 15; u++)
        if ((uint32_t)u < blen) {
          const uint32_t pv = bp[u + 1]; const uint32_t reg = pv >> rb, rloc = reg - rbase + 1u;
          k1_mark(bm, rloc);
          if (((pv & rmask) < ovl) && reg > 0) k1_mark(bm, rloc - 1u);
        }
    }
    for (int off = tid; off < NL; off += nthr) {
      uint32_t l = 0, h = 0, c = 0;
      const uint32_t lb = lbeg[off], le = lend[off];
      if (le > lb) {
        const uint32_t* plist = ix.seed[off / max_n_kmers].pos;
        if (S == 1) { l = lb; h = le; }
        else { const uint32_t* dir = ix.seed[off / max_n_kmers].dir; l = dir[kS[off] + s]; h = dir[kS[off] + s + 1]; }
        if (h - l > K1_SHORT) {
          c = (h - l) + (l > lb ? 1u : 0u) + (h < le ? 1u : 0u);     // + one candidate border entry on each side
          any_long = true;
        } else {
          for (uint32_t e = l; e < h; e += K1_UNROLL) {
            // unconditional wide loads (two dwordx4 per lane, both in flight); lanes past the slice end read the
            // next list / the 0xffffffff tail pad and are masked below -- predicated loads would serialise
            uint32_t p[K1_UNROLL];
            k1_load8(plist + e, p);
#pragma unroll
            for (int u = 0; u < K1_UNROLL; u++)
              if (e + u < h) {
                const uint32_t reg = p[u] >> rb, rloc = reg - rbase + 1u;
                if GM_ABL(4) { if (p[u] == 0x12345u) bm[0] = 1; continue; }
                k1_mark(bm, rloc);
                if (((p[u] & rmask) < ovl) && reg > 0) k1_mark(bm, rloc - 1u);
              }
          }
          if (S > 1 && !GM_ABL(2)) {
            // entries of the previous slab inside the last region before B count for local region 0
            for (uint32_t q = l; q > lb;) { --q; if ((plist[q] >> rb) + 1u != rbase) break; k1_mark(bm, 0u); }
            // entries of the next slab inside the overlap strip count for this slab's last region
            for (uint32_t q = h; q < le; q++) { const uint32_t pq = plist[q]; if ((pq >> rb) != rend || (pq & rmask) >= ovl) break; k1_mark(bm, rend - rbase); }
          }
        }
      }
      lo[off] = l; hi[off] = h; pre[off] = c;
    }
    if (any_long) sh.total = 1;            // benign race: every writer stores 1
    __syncthreads();
    const bool have_long = sh.total != 0;
    uint32_t total = 0;
    if (have_long) {
      __syncthreads();
      // exclusive scan of pre[0..NL) by wave 0
      if (tid < GM_WAVE) {
        const int per = (NL + GM_WAVE - 1) / GM_WAVE;
        const int a0 = tid * per, a1 = min(NL, a0 + per);
        uint32_t sum = 0;
        for (int a = a0; a < a1; a++) sum += pre[a];
        uint32_t incl = sum;
        for (int d = 1; d < GM_WAVE; d <<= 1) { uint32_t o = __shfl_up(incl, d); if (tid >= d) incl += o; }
        uint32_t run = incl - sum;
        for (int a = a0; a < a1; a++) { uint32_t c = pre[a]; pre[a] = run; run += c; }
        if (tid == GM_WAVE - 1) { pre[NL] = incl; sh.total = incl; }
      }
      __syncthreads();
      total = sh.total;
    }
    for (int phase = 0; phase < (MP == 1 ? 1 : 2); phase++) {
      if (BKT && phase == 1) {
        uint32_t alive = 0;                 // bit u: register entry u survives
#pragma unroll
        for (int u = 0; u < 15; u++)
          if ((uint32_t)u < blen) {
            const uint32_t pv = bp[u + 1]; const uint32_t reg = pv >> rb, rloc = reg - rbase + 1u;
            const bool strip = ((pv & rmask) < ovl) && reg > 0;
            if (keep(reg, rloc, strip)) alive |= 1u << u;
          }
        if (alive) {
          const uint32_t cnt = __popc(alive);
          uint32_t slot = atomicAdd(&sh.n_surv, cnt);
          const uint32_t sn = (uint32_t)tid / (uint32_t)max_n_kmers, y = (uint32_t)tid - sn * (uint32_t)max_n_kmers;
#pragma unroll
          for (int u = 0; u < 15; u++)
            if (alive & (1u << u)) { if (slot < scap) out[slot] = ((uint64_t)bp[u + 1] << 32) | ((uint64_t)y << 16) | sn; slot++; }
        }
      }
      if (phase == 1 && !GM_ABL(1)) {
        // -- phase 1 (short slices): survival test, re-reading the slice from L1/L2 --
        for (int off = tid; off < NL; off += nthr) {
          const uint32_t l = lo[off], h = hi[off];
          if (h <= l || h - l > K1_SHORT) continue;
          const uint32_t* plist = ix.seed[off / max_n_kmers].pos;
          for (uint32_t e = l; e < h; e += K1_UNROLL) {
            uint32_t p[K1_UNROLL];
            k1_load8(plist + e, p);
#pragma unroll
            for (int u = 0; u < K1_UNROLL; u++)
              if (e + u < h) {
                const uint32_t reg = p[u] >> rb, rloc = reg - rbase + 1u;
                const bool strip = ((p[u] & rmask) < ovl) && reg > 0;
                if (keep(reg, rloc, strip)) emit(p[u], off);
              }
          }
        }
      }
      // -- long slices, flattened over the workgroup (both phases) --
      for (uint32_t e0 = 0; e0 < total; e0 += nthr) {
        const uint32_t e = e0 + tid;
        if (e < total) {
          // list of entry e: largest l with pre[l] <= e (pre non-decreasing; empty lists skipped by <=)
          int a = 0, z = NL;
          while (z - a > 1) { int m = (a + z) >> 1; if (pre[m] <= e) a = m; else z = m; }
          const int off = a;
          const uint32_t l = lo[off], h = hi[off];
          const uint32_t ext_lo = l - (l > lbeg[off] ? 1u : 0u);
          const uint32_t idx = ext_lo + (e - pre[off]);
          const uint32_t* plist = ix.seed[off / max_n_kmers].pos;
          const uint32_t p = plist[idx];
          const uint32_t reg = p >> rb;
          if (idx >= l && idx < h) {
            const uint32_t rloc = reg - rbase + 1u;
            const bool strip = ((p & rmask) < ovl) && reg > 0;
            if (phase == 0) {
              k1_mark(bm, rloc);
              if (strip) k1_mark(bm, rloc - 1u);
            } else if (keep(reg, rloc, strip)) emit(p, off);
          } else if (phase == 0) {
            if (idx < l) {
              if (reg + 1u == rbase) {
                k1_mark(bm, 0u);
                for (uint32_t q = idx; q > lbeg[off];) { --q; if ((plist[q] >> rb) + 1u != rbase) break; k1_mark(bm, 0u); }
              }
            } else {
              if (reg == rend && (p & rmask) < ovl) {
                k1_mark(bm, rend - rbase);
                for (uint32_t q = idx + 1; q < lend[off]; q++) {
                  const uint32_t pq = plist[q];
                  if ((pq >> rb) != rend || (pq & rmask) >= ovl) break;
                  k1_mark(bm, rend - rbase);
                }
              }
            }
          }
        }
      }
      __syncthreads();
    }
    if (MP == 1) {   // the regions of THIS slab marked twice: local indexes 1 .. rend - rbase (their counts are complete: the border entries of both neighbours are in)
      const uint32_t last = (uint32_t)min((uint64_t)rend, (ix.total_len >> rb) + 1u) - rbase;
      for (int w = tid; w < bm_words; w += nthr) {
        uint32_t m = bm[w] & 0xAAAAAAAAu;
        while (m) {
          const int b = __ffs(m) - 1; m &= m - 1u;
          const uint32_t rloc = (uint32_t)w * 16u + (uint32_t)(b >> 1);
          if (rloc >= 1u && rloc <= last) { const uint32_t slot = atomicAdd(&sh_mp_n, 1u); if (slot < (uint32_t)GM_MP_CAP) sh_mp_row[slot] = rloc - 1u + rbase; }
        }
      }
      __syncthreads();
    }
    if (surv_seg && tid == 0) surv_seg[(size_t)rs * (S + 1) + s + 1] = sh.n_surv;      // survivors come out slab by slab: K1b prunes per slab
  }
  if (MP == 1) {   // sort the row (the slabs come in order, the lanes within one do not), write it out; nothing else leaves the kernel in this mode
    const uint32_t n = sh_mp_n;
    if (tid == 0) ix.mp.out_cnt[rs] = n;
    if (n <= (uint32_t)GM_MP_CAP) {
      uint32_t np = 1; while (np < n) np <<= 1;
      for (uint32_t i = n + tid; i < np; i += nthr) sh_mp_row[i] = 0xFFFFFFFFu;
      __syncthreads();
      for (uint32_t k = 2; k <= np; k <<= 1)
        for (uint32_t j = k >> 1; j > 0; j >>= 1) {
          for (uint32_t i = tid; i < np; i += nthr) {
            const uint32_t l = i ^ j;
            if (l > i) { const uint32_t a = sh_mp_row[i], b = sh_mp_row[l]; if (((i & k) == 0) ? (a > b) : (a < b)) { sh_mp_row[i] = b; sh_mp_row[l] = a; } }
          }
          __syncthreads();
        }
      for (uint32_t i = tid; i < n; i += nthr) ix.mp.out_rows[(size_t)rs * GM_MP_CAP + i] = sh_mp_row[i];
    }
    return;
  }
  if (MP == 3) return;                   // (only the flags leave the kernel in this mode)
  if (redo) return;                      // counters were taken by the first run
  if (tid == 0) {
    surv_cnt[rs] = sh.n_surv;
    GS_ADD(stats, GS_SURVIVORS, (unsigned long long)sh.n_surv);
    if (sh.n_surv > scap) {              // too many for the LDS tier of K2: handled by the heavy tier
      const uint32_t hs = atomicAdd(heavy_cnt, 1u);
      if (hs < (uint32_t)heavy_cap) heavy_list[hs] = (uint32_t)rs; else GS_ADD(stats, GS_OVERFLOW_SURV, 1ull);
    }
  }
  // per-wave reduction of the work counters
  if (fb_cnt) return;                    // the work counters were taken by the first run
  for (int d = GM_WAVE / 2; d > 0; d >>= 1) { my_lookups += __shfl_down(my_lookups, d); my_entries += __shfl_down(my_entries, d); }
  if ((tid & (GM_WAVE - 1)) == 0) { GS_ADD(stats, GS_LOOKUPS, my_lookups); GS_ADD(stats, GS_ENTRIES, my_entries); }
}

// One block per read-strand; in list mode (fb_list) a bounded grid walks the list -- one block per possible entry meant hundreds of thousands of empty
// workgroups per launch, each of which still had to be given its LDS before it could return (1 ms per launch with an empty list).
template <bool BKT, int MP = 0>
__global__ void __launch_bounds__(1024)
k_lookup(GmIndexDev ix, const uint32_t* __restrict__ reads, int n_reads, int read_len, int read_words, int max_n_kmers,
         int NL, int bm_words, uint64_t* __restrict__ surv, uint32_t* __restrict__ surv_cnt, int scap_all,
         uint32_t* __restrict__ heavy_list, uint32_t* __restrict__ heavy_cnt, int heavy_cap,
         const uint32_t* __restrict__ redo_list, const uint64_t* __restrict__ redo_off,
         const uint32_t* __restrict__ fb_list, const uint32_t* __restrict__ fb_cnt,
         unsigned long long* __restrict__ stats, int ablate, uint32_t* __restrict__ surv_seg) {
  if (fb_cnt) {
    const int n_items = (int)*fb_cnt;
    for (int item = blockIdx.x; item < n_items; item += gridDim.x) {
      k_lookup_body<BKT, MP>(item, ix, reads, n_reads, read_len, read_words, max_n_kmers, NL, bm_words, surv, surv_cnt, scap_all, heavy_list, heavy_cnt, heavy_cap,
                         redo_list, redo_off, fb_list, fb_cnt, stats, ablate, surv_seg);
      __syncthreads();
    }
  } else k_lookup_body<BKT, MP>((int)blockIdx.x, ix, reads, n_reads, read_len, read_words, max_n_kmers, NL, bm_words, surv, surv_cnt, scap_all, heavy_list, heavy_cnt, heavy_cap,
                            redo_list, redo_off, fb_list, fb_cnt, stats, ablate, surv_seg);
}

// ---------------------------------------------------------------------------------------------
// Bucket mode (small genomes: one slab, mean list length <= 12): one thread per k-mer of the read-strand.
// The probe is ONE 64-byte bucket per k-mer = {list length, first 15 positions}, so a lookup costs one HBM
// sector instead of directory + list, and both sweeps run out of registers.  Lists longer than 15 entries
// (and <= the cutoff) are streamed from pos[] by their wave, 64 entries at a time.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(512)
k_lookup_bkt(GmIndexDev ix, const uint32_t* __restrict__ reads, int n_reads, int read_len, int read_words, int max_n_kmers,
             int NL, int bm_words, uint64_t* __restrict__ surv, uint32_t* __restrict__ surv_cnt, int scap_all,
             uint32_t* __restrict__ heavy_list, uint32_t* __restrict__ heavy_cnt, int heavy_cap,
             unsigned long long* __restrict__ stats, uint32_t* __restrict__ surv_seg) {
  extern __shared__ __align__(16) uint32_t smem[];
  __shared__ uint32_t n_surv;
  const int tid = threadIdx.x, lane = tid & 63;
  const int rs = blockIdx.x, rd = rs >> 1, st = rs & 1;
  uint8_t* codes = (uint8_t*)smem;
  uint32_t* bm = smem + ((((read_len + 3) / 4) + 3) & ~3);
  const int rb = ix.region_bits;
  const uint32_t rmask = (1u << rb) - 1u, ovl = (uint32_t)ix.region_overlap;
  uint64_t* out = surv + (size_t)rs * scap_all;
  const uint32_t scap = (uint32_t)scap_all;
  const uint32_t* rw = reads + (size_t)rd * read_words;
  for (int i = tid; i < read_len; i += blockDim.x) {
    codes[i] = (uint8_t)gm_read_code(rw, read_len, st ^ ix.cs_flip, ix.colour, i, ix.read_rna && ix.read_rna[rd]);
  }
  { uint4* bm4 = (uint4*)bm; for (int w = tid; w < (bm_words >> 2); w += blockDim.x) bm4[w] = make_uint4(0, 0, 0, 0); }
  if (tid == 0) n_surv = 0;
  __syncthreads();
  // my k-mer
  const int sn = tid / max_n_kmers, i = tid - sn * max_n_kmers;
  uint32_t len = 0, lb = 0; bool longl = false; uint32_t lookups = 0, y = (uint32_t)i;
  uint4 q0 = make_uint4(0, 0, 0, 0), q1 = q0, q2 = q0, q3 = q0;
  const uint32_t* plist = nullptr;
  if (tid < NL && i >= ix.colour && i + ix.seed[sn].span <= read_len) {
    const int span = ix.seed[sn].span; const uint64_t mask = ix.seed[sn].mask;
    const uint32_t mapidx = gm_mapidx(ix, mask, span, codes + i);
    const uint4* bk = (const uint4*)(ix.seed[sn].bkt + (size_t)mapidx * 16);
    q0 = bk[0]; q1 = bk[1]; q2 = bk[2]; q3 = bk[3];
    lookups = 1;
    len = q0.x;
    if (len > ix.list_cutoff) len = 0;                 // ref: mapping.c:497 (skipped, not deleted)
    if (len > 15u) { longl = true; lb = ix.seed[sn].dir[mapidx]; plist = ix.seed[sn].pos; }
  }
  const uint32_t p[15] = {q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, q3.x, q3.y, q3.z, q3.w};
  const uint32_t nreg = longl ? 0u : len;
  // -- mark --
#pragma unroll
  for (int u = 0; u < 15; u++)
    if ((uint32_t)u < nreg) {
      const uint32_t reg = p[u] >> rb;
      k1_mark(bm, reg + 1u);
      if (((p[u] & rmask) < ovl) && reg > 0) k1_mark(bm, reg);
    }
  unsigned long long lm = __ballot(longl);
  while (lm) {                                         // long lists of this wave, one at a time, 64 entries per step
    const int l = __builtin_ctzll(lm); lm &= lm - 1;
    const uint32_t b = (uint32_t)__builtin_amdgcn_readlane((int)lb, l), n = (uint32_t)__builtin_amdgcn_readlane((int)len, l);
    const uint32_t* pl = ix.seed[__builtin_amdgcn_readlane(sn, l)].pos;
    for (uint32_t q = lane; q < n; q += 64) {
      const uint32_t pv = pl[b + q]; const uint32_t reg = pv >> rb;
      k1_mark(bm, reg + 1u);
      if (((pv & rmask) < ovl) && reg > 0) k1_mark(bm, reg);
    }
  }
  __syncthreads();
  // -- test + emit --
  uint32_t alive = 0;
#pragma unroll
  for (int u = 0; u < 15; u++)
    if ((uint32_t)u < nreg) {
      const uint32_t reg = p[u] >> rb;
      const bool strip = ((p[u] & rmask) < ovl) && reg > 0;
      if (k1_has2(bm, reg + 1u) || (strip && k1_has2(bm, reg))) alive |= 1u << u;
    }
  if (alive) {
    uint32_t slot = atomicAdd(&n_surv, (uint32_t)__popc(alive));
#pragma unroll
    for (int u = 0; u < 15; u++)
      if (alive & (1u << u)) { if (slot < scap) out[slot] = ((uint64_t)p[u] << 32) | ((uint64_t)y << 16) | (uint32_t)sn; slot++; }
  }
  lm = __ballot(longl);
  while (lm) {
    const int l = __builtin_ctzll(lm); lm &= lm - 1;
    const uint32_t b = (uint32_t)__builtin_amdgcn_readlane((int)lb, l), n = (uint32_t)__builtin_amdgcn_readlane((int)len, l);
    const int lsn = __builtin_amdgcn_readlane(sn, l); const uint32_t ly = (uint32_t)__builtin_amdgcn_readlane((int)y, l);
    const uint32_t* pl = ix.seed[lsn].pos;
    for (uint32_t q = lane; q < n; q += 64) {
      const uint32_t pv = pl[b + q]; const uint32_t reg = pv >> rb;
      const bool strip = ((pv & rmask) < ovl) && reg > 0;
      if (k1_has2(bm, reg + 1u) || (strip && k1_has2(bm, reg))) {
        const uint32_t slot = atomicAdd(&n_surv, 1u);
        if (slot < scap) out[slot] = ((uint64_t)pv << 32) | ((uint64_t)ly << 16) | (uint32_t)lsn;
      }
    }
  }
  __syncthreads();
  if (tid == 0) {
    surv_cnt[rs] = n_surv;
    if (surv_seg) surv_seg[(size_t)rs * 2 + 1] = n_surv;      // one slab
    GS_ADD(stats, GS_SURVIVORS, n_surv);
    if (n_surv > scap) {
      const uint32_t hs = atomicAdd(heavy_cnt, 1u);
      if (hs < (uint32_t)heavy_cap) heavy_list[hs] = (uint32_t)rs; else GS_ADD(stats, GS_OVERFLOW_SURV, 1ull);
    }
  }
  unsigned long long lk = lookups, en = len;
  for (int d = GM_WAVE / 2; d > 0; d >>= 1) { lk += __shfl_down(lk, d); en += __shfl_down(en, d); }
  if (lane == 0) { GS_ADD(stats, GS_LOOKUPS, lk); GS_ADD(stats, GS_ENTRIES, en); }
}

// ---------------------------------------------------------------------------------------------
// Bucket mode, persistent form (round 4): the same lookup as k_lookup_bkt with the workgroups resident -- each strides through read-strands -- and the probes issued ONE
// READ-STRAND AHEAD: a thread keeps its k-mer slot (seed, offset: span, mask and bucket base in registers), works out the map index of the NEXT read-strand from a second
// code buffer and has its 64-byte bucket on the way while the marks and tests of the current one run out of the registers loaded an iteration earlier.  (One workgroup per
// read-strand paid the whole chain -- codes, barrier, map index, bucket, marks, barrier, tests -- per launch slot: ~25 us each at eight workgroups a CU.)
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(512)
k_lookup_bkt_p(GmIndexDev ix, const uint32_t* __restrict__ reads, int n_reads, int read_len, int read_words, int max_n_kmers,
               int NL, int bm_words, uint64_t* __restrict__ surv, uint32_t* __restrict__ surv_cnt, int scap_all,
               uint32_t* __restrict__ heavy_list, uint32_t* __restrict__ heavy_cnt, int heavy_cap,
               unsigned long long* __restrict__ stats, uint32_t* __restrict__ surv_seg) {
  extern __shared__ __align__(16) uint32_t smem[];
  __shared__ uint32_t n_surv;
  const int tid = threadIdx.x, lane = tid & 63;
  const int cwords = (((read_len + 3) / 4) + 3) & ~3;                 // words of one code buffer
  uint8_t* const codes0 = (uint8_t*)smem; uint8_t* const codes1 = (uint8_t*)(smem + cwords);
  uint32_t* const bm = smem + 2 * cwords;
  const int rb = ix.region_bits;
  const uint32_t rmask = (1u << rb) - 1u, ovl = (uint32_t)ix.region_overlap;
  const uint32_t scap = (uint32_t)scap_all;
  const int n_rs = 2 * n_reads;
  // this thread's k-mer slot, the same for every read-strand
  const int sn = tid / max_n_kmers, i = tid - sn * max_n_kmers;
  int span = 0; uint64_t mask = 0; const uint32_t* bkt = nullptr; const uint32_t* dirp = nullptr; const uint32_t* posp = nullptr;
  bool slot = false;
  if (tid < NL && sn < ix.n_seeds) {
    span = ix.seed[sn].span; mask = ix.seed[sn].mask; bkt = ix.seed[sn].bkt; dirp = ix.seed[sn].dir; posp = ix.seed[sn].pos;
    slot = i >= ix.colour && i + span <= read_len;
  }
  const uint32_t y = (uint32_t)i;
  auto fill_codes = [&](const int rs, uint8_t* cb) {
    if (rs >= n_rs) return;
    const int rd = rs >> 1, st = rs & 1;
    const uint32_t* rw = reads + (size_t)rd * read_words;
    const bool rna = ix.read_rna && ix.read_rna[rd];
    for (int k = tid; k < read_len; k += blockDim.x) cb[k] = (uint8_t)gm_read_code(rw, read_len, st ^ ix.cs_flip, ix.colour, k, rna);
  };
  struct Probe { uint4 q0, q1, q2, q3; uint32_t mapidx; };
  auto fetch = [&](const int rs, const uint8_t* cb, Probe& P) {      // the bucket of this thread's k-mer of read-strand rs: loads issued, used an iteration later
    P.q0 = make_uint4(0, 0, 0, 0); P.q1 = P.q0; P.q2 = P.q0; P.q3 = P.q0; P.mapidx = 0;
    if (slot && rs < n_rs) {
      if (!ix.hflag) {
        // KMER_TO_MAPIDX (ref: gmapper.h:349-368) without a branch per base: eight code bytes per LDS round trip, the mask bit selects (as k_lookup_v5's set-up)
        uint32_t m = 0;
        for (int t0 = 0; t0 < ix.max_seed_span; t0 += 8) {
          uint32_t c[8];
#pragma unroll
          for (int u = 0; u < 8; u++) { const int x = i + span - 1 - t0 - u; c[u] = cb[min(max(x, 0), read_len - 1)]; }
#pragma unroll
          for (int u = 0; u < 8; u++) m = ((mask >> (t0 + u)) & 1ull) ? ((m << 2) | (c[u] & 3u)) : m;
        }
        P.mapidx = m;
      } else P.mapidx = gm_mapidx(ix, mask, span, cb + i);
      const uint4* bk = (const uint4*)(bkt + (size_t)P.mapidx * 16);
      P.q0 = bk[0]; P.q1 = bk[1]; P.q2 = bk[2]; P.q3 = bk[3];
    }
  };
  unsigned long long lk = 0, en = 0;
  { uint4* bm4 = (uint4*)bm; for (int w = tid; w < (bm_words >> 2); w += blockDim.x) bm4[w] = make_uint4(0, 0, 0, 0); }
  if (tid == 0) n_surv = 0;
  fill_codes((int)blockIdx.x, codes0);
  __syncthreads();
  Probe cur; fetch((int)blockIdx.x, codes0, cur);
  int it = 0;
  for (int rs = blockIdx.x; rs < n_rs; rs += gridDim.x, it++) {
    uint8_t* const cb_next = (it & 1) ? codes0 : codes1;
    const int rs_next = rs + (int)gridDim.x;
    fill_codes(rs_next, cb_next);
    __syncthreads();                                     // the next read-strand's codes are in; the bitmap is clear, n_surv is 0
    Probe nxt; fetch(rs_next, cb_next, nxt);
    // ---- the current read-strand, from the registers loaded an iteration ago ----
    uint64_t* out = surv + (size_t)rs * scap_all;
    uint32_t len = 0, lb = 0; bool longl = false;
    if (slot) {
      lk++;
      len = cur.q0.x;
      if (len > ix.list_cutoff) len = 0;                 // ref: mapping.c:497 (skipped, not deleted)
      if (len > 15u) { longl = true; lb = dirp[cur.mapidx]; }
      en += len;
    }
    const uint32_t p[15] = {cur.q0.y, cur.q0.z, cur.q0.w, cur.q1.x, cur.q1.y, cur.q1.z, cur.q1.w, cur.q2.x, cur.q2.y, cur.q2.z, cur.q2.w, cur.q3.x, cur.q3.y, cur.q3.z, cur.q3.w};
    const uint32_t nreg = longl ? 0u : len;
    // the marks of the bucket's entries, five at a time: the first marks of five entries go out together, then the second marks of those that found theirs set (a mark
    // is two dependent LDS atomics; one entry after the other made a chain of up to thirty round trips); the few strip marks (2.4 % of the entries) afterwards
#pragma unroll
    for (int u0 = 0; u0 < 15; u0 += 5) {
      uint32_t old[5];
#pragma unroll
      for (int k = 0; k < 5; k++) { old[k] = 0; if ((uint32_t)(u0 + k) < nreg) { const uint32_t rl = (p[u0 + k] >> rb) + 1u; old[k] = atomicOr(&bm[rl >> 4], 1u << ((rl & 15u) * 2u)); } }
#pragma unroll
      for (int k = 0; k < 5; k++) if ((uint32_t)(u0 + k) < nreg) { const uint32_t rl = (p[u0 + k] >> rb) + 1u, sh = (rl & 15u) * 2u; if (old[k] & (1u << sh)) atomicOr(&bm[rl >> 4], 2u << sh); }
    }
#pragma unroll
    for (int u = 0; u < 15; u++)
      if ((uint32_t)u < nreg && ((p[u] & rmask) < ovl) && (p[u] >> rb) > 0) k1_mark(bm, p[u] >> rb);
    unsigned long long lm = __ballot(longl);
    while (lm) {                                         // long lists of this wave, one at a time, 64 entries per step
      const int l = __builtin_ctzll(lm); lm &= lm - 1;
      const uint32_t b = (uint32_t)__builtin_amdgcn_readlane((int)lb, l), n = (uint32_t)__builtin_amdgcn_readlane((int)len, l);
      const uint32_t* pl = ix.seed[__builtin_amdgcn_readlane(sn, l)].pos;
      for (uint32_t q = lane; q < n; q += 64) {
        const uint32_t pv = pl[b + q]; const uint32_t reg = pv >> rb;
        k1_mark(bm, reg + 1u);
        if (((pv & rmask) < ovl) && reg > 0) k1_mark(bm, reg);
      }
    }
    __syncthreads();
    uint32_t alive = 0;
#pragma unroll
    for (int u = 0; u < 15; u++)
      if ((uint32_t)u < nreg) {
        const uint32_t reg = p[u] >> rb;
        const bool strip = ((p[u] & rmask) < ovl) && reg > 0;
        if (k1_has2(bm, reg + 1u) || (strip && k1_has2(bm, reg))) alive |= 1u << u;
      }
    if (alive) {
      uint32_t sl = atomicAdd(&n_surv, (uint32_t)__popc(alive));
#pragma unroll
      for (int u = 0; u < 15; u++)
        if (alive & (1u << u)) { if (sl < scap) out[sl] = ((uint64_t)p[u] << 32) | ((uint64_t)y << 16) | (uint32_t)sn; sl++; }
    }
    lm = __ballot(longl);
    while (lm) {
      const int l = __builtin_ctzll(lm); lm &= lm - 1;
      const uint32_t b = (uint32_t)__builtin_amdgcn_readlane((int)lb, l), n = (uint32_t)__builtin_amdgcn_readlane((int)len, l);
      const int lsn = __builtin_amdgcn_readlane(sn, l); const uint32_t ly = (uint32_t)__builtin_amdgcn_readlane((int)y, l);
      const uint32_t* pl = ix.seed[lsn].pos;
      for (uint32_t q = lane; q < n; q += 64) {
        const uint32_t pv = pl[b + q]; const uint32_t reg = pv >> rb;
        const bool strip = ((pv & rmask) < ovl) && reg > 0;
        if (k1_has2(bm, reg + 1u) || (strip && k1_has2(bm, reg))) {
          const uint32_t sl = atomicAdd(&n_surv, 1u);
          if (sl < scap) out[sl] = ((uint64_t)pv << 32) | ((uint64_t)ly << 16) | (uint32_t)lsn;
        }
      }
    }
    __syncthreads();                                     // every test has read the bitmap, every survivor is counted
    if (tid == 0) {
      const uint32_t ns = n_surv;
      surv_cnt[rs] = ns;
      if (surv_seg) surv_seg[(size_t)rs * 2 + 1] = ns;    // one slab
      GS_ADD(stats, GS_SURVIVORS, ns);
      if (ns > scap) {
        const uint32_t hs = atomicAdd(heavy_cnt, 1u);
        if (hs < (uint32_t)heavy_cap) heavy_list[hs] = (uint32_t)rs; else GS_ADD(stats, GS_OVERFLOW_SURV, 1ull);
      }
      n_surv = 0;
    }
    { uint4* bm4 = (uint4*)bm; for (int w = tid; w < (bm_words >> 2); w += blockDim.x) bm4[w] = make_uint4(0, 0, 0, 0); }      // (visible behind the barrier at the next top)
    cur = nxt;
  }
  (void)posp;
  for (int d = GM_WAVE / 2; d > 0; d >>= 1) { lk += __shfl_down(lk, d); en += __shfl_down(en, d); }
  if (lane == 0) { GS_ADD(stats, GS_LOOKUPS, lk); GS_ADD(stats, GS_ENTRIES, en); }
}

// ---------------------------------------------------------------------------------------------
// K1 v3 (large genomes: several slabs, list slices of tens of entries).  Same slab sweep and the same
// exact region counters as k_lookup; what differs is the lane mapping and the amount of code per entry.
// rocprofv3 on the 3 Gbp workload showed the lane-per-list kernel neither HBM- nor L2-bound but
// instruction-issue bound (profiles/r01c_*): ~125 VALU instructions per list entry.  Here
//   * K1G = 8 lanes share one list slice and read it with ONE dwordx4 each (a 32-position window per
//     list and wave-instruction), so the per-window bookkeeping is shared by up to 32 entries;
//   * everything a window needs sits in a 12-byte LDS record per list (pointer to the list, read offset
//     and seed) plus 16-bit per-slab offsets, fetched from the directory ONCE per list, not per slab;
//   * one small loop body serves both phases (no unrolling: the first version of this kernel was 27 k
//     instructions long and paid for it in instruction fetch);
//   * out-of-slice lanes are steered to a spare counter instead of being branched around.
// ---------------------------------------------------------------------------------------------
#define K1G 8
#define K1W (K1G * 4)    // positions per window
#define K1Q 4            // windows per lane group whose loads are in flight together

__global__ void __launch_bounds__(768)
k_lookup_v3(GmIndexDev ix, const uint32_t* __restrict__ reads, int n_reads, int read_len, int read_words, int max_n_kmers,
            int NL, int bm_words, uint64_t* __restrict__ surv, uint32_t* __restrict__ surv_cnt, int scap_all,
            uint32_t* __restrict__ heavy_list, uint32_t* __restrict__ heavy_cnt, int heavy_cap, unsigned long long* __restrict__ stats, int ablate,
            uint32_t* __restrict__ surv_seg) {
  extern __shared__ __align__(16) uint32_t smem[];
  __shared__ uint32_t n_surv, n_lists, any_long;
  const int tid = threadIdx.x, nthr = blockDim.x;
  const int rs = blockIdx.x, rd = rs >> 1, st = rs & 1;
  uint64_t* out = surv + (size_t)rs * scap_all;
  const uint32_t scap = (uint32_t)scap_all;
  const int S = ix.n_slabs, rb = ix.region_bits;
  const uint32_t rmask = (1u << rb) - 1u, ovl = (uint32_t)ix.region_overlap;
  // LDS: codes | rec[NL] (3 words: list offset lo/hi, y << 16 | seed) | wmap[S][2 NL] (u16) | nwin[S] (u16) | doff[NL * (S + 1)] (u16) | bitmap
  uint8_t* codes = (uint8_t*)smem;
  const int code_words = (read_len + 3) / 4;
  uint32_t* rec = smem + code_words;
  uint16_t* wmap_all = (uint16_t*)(rec + 3 * NL);      // [S][2 * NL] window -> list | window number << 15
  uint16_t* nwin_s = wmap_all + (size_t)S * 2 * NL;    // [S] windows in the flat array of slab s (padded to 2 * ((S + 1) / 2))
  uint16_t* doff = nwin_s + ((S + 1) & ~1);
  uint32_t* bm = smem + ((code_words + 3 * NL + S * NL + (S + 1) / 2 + (NL * (S + 1) + 1) / 2 + 3) & ~3);

  const uint32_t* rw = reads + (size_t)rd * read_words;
  for (int i = tid; i < read_len; i += nthr) {
    codes[i] = (uint8_t)gm_read_code(rw, read_len, st ^ ix.cs_flip, ix.colour, i, ix.read_rna && ix.read_rna[rd]);
  }
  if (tid == 0) { n_surv = 0; n_lists = 0; any_long = 0; }
  __syncthreads();

  const uint32_t* __restrict__ pos0 = ix.seed[0].pos;
  // ---- map indexes, list bounds, per-slab offsets (ref: mapping.c:53-66, KMER_TO_MAPIDX gmapper.h:349-368) ----
  // only non-empty lists get a record (compact, any order: the survivors are sorted later)
  unsigned long long my_lookups = 0, my_entries = 0;
  for (int off = tid; off < NL; off += nthr) {
    const int sn = off / max_n_kmers, i = off - sn * max_n_kmers;
    const int span = ix.seed[sn].span;
    if (i < ix.colour || i + span > read_len) continue;
    const uint64_t mask = ix.seed[sn].mask;
    const uint32_t mapidx = gm_mapidx(ix, mask, span, codes + i);
    const uint32_t* dir = ix.seed[sn].dir + (size_t)mapidx * (uint32_t)S;
    my_lookups++;
    const uint32_t b = dir[0], e = dir[S];
    if (e == b || e - b > ix.list_cutoff) continue;        // ref: mapping.c:497 (longer lists are skipped, not deleted)
    my_entries += (e - b);
    const uint32_t j = atomicAdd(&n_lists, 1u);
    const uint64_t ptr = (uint64_t)((ix.seed[sn].pos + b) - pos0);        // element offset from seed 0's array (both hipMalloc'ed: 256-byte aligned)
    rec[3 * j] = (uint32_t)ptr; rec[3 * j + 1] = (uint32_t)(ptr >> 32); rec[3 * j + 2] = ((uint32_t)i << 16) | (uint32_t)sn;
    uint16_t* d = doff + (size_t)j * (S + 1);
    d[0] = 0; d[S] = (uint16_t)(e - b);
    for (int s = 1; s < S; s++) d[s] = (uint16_t)(dir[s] - b);
  }
  __syncthreads();
  const int nl = (int)n_lists;
  // Windows per slab: list j contributes ceil((slice + neighbours) / 32) windows; the first two of every list are laid
  // out in one flat array per slab (wave w builds the array of slab w, w + nwaves, ...), the rest (repeats) is handled apart.
  for (int sl = tid / GM_WAVE; sl < S; sl += nthr / GM_WAVE) {
    const int ln = tid & (GM_WAVE - 1);
    const int per = (nl + GM_WAVE - 1) / GM_WAVE;
    const int a0 = ln * per, a1 = min(nl, a0 + per);
    auto nw_of = [&](int a) -> uint32_t {
      const uint16_t* d = doff + (size_t)a * (S + 1);
      const uint32_t d0 = d[sl], d1 = d[sl + 1], dn = d[S];
      const uint32_t lo = d0 - ((S > 1 && d0 > 0) ? 1u : 0u), hi = d1 + ((S > 1 && d1 < dn) ? 1u : 0u);
      return (hi - lo + K1W - 1) / K1W;
    };
    uint32_t sum = 0, lng = 0;
    for (int a = a0; a < a1; a++) { const uint32_t nw = nw_of(a); sum += min(nw, 2u); lng |= (nw > 2) ? 1u : 0u; }
    uint32_t incl = sum;
    for (int dd = 1; dd < GM_WAVE; dd <<= 1) { const uint32_t o = __shfl_up(incl, dd); if (ln >= dd) incl += o; }
    uint32_t run = incl - sum;
    uint16_t* wm = wmap_all + (size_t)sl * 2 * NL;
    for (int a = a0; a < a1; a++) { const uint32_t c = min(nw_of(a), 2u); for (uint32_t k = 0; k < c; k++) wm[run + k] = (uint16_t)((uint32_t)a | (k << 15)); run += c; }
    if (__any(lng != 0) && ln == 0) any_long = 1u;
    if (ln == GM_WAVE - 1) nwin_s[sl] = (uint16_t)incl;
  }

  const int ng = nthr / K1G, g = tid / K1G, gl4 = (tid % K1G) * 4;
  const int lane = tid & (GM_WAVE - 1);
  const uint32_t spare = (uint32_t)bm_words * 16u - 1u;     // counter slot no region of a slab maps to
  for (int s = 0; s < S; s++) {
    const uint64_t B = (uint64_t)s << ix.slab_bits;
    const uint32_t rbase = (uint32_t)(B >> rb);             // local region index = region - rbase + 1
    const uint32_t rend = (uint32_t)((B + (1ull << ix.slab_bits)) >> rb);
    {
      uint4* bm4 = (uint4*)bm; const int n4 = bm_words >> 2;
      for (int w = tid; w < n4; w += nthr) bm4[w] = make_uint4(0, 0, 0, 0);
    }
    __syncthreads();
    const int nwin = (int)nwin_s[s]; const bool have_long = any_long != 0;
    const uint16_t* wmap = wmap_all + (size_t)s * 2 * NL;
    // One 32-position window: the lane owns positions [w, w + 4) of list j (relative to the list's first entry).
    // rl[u] = local region counter of element u, or `spare` when the element is outside the slice (neighbour, padding).
    auto region_slots = [&](const bool act, const int j, const uint32_t w, const uint32_t pv[4], uint32_t rl[4], uint32_t& r0, uint32_t& nv, uint32_t& d1, uint32_t& dn) {
      uint32_t d0 = 0; d1 = 0; dn = 0;
      if (act) { const uint16_t* d = doff + (size_t)j * (S + 1); d0 = d[s]; d1 = d[s + 1]; dn = d[S]; }
      nv = d1 - d0;                                                      // 0 for inactive lanes
      r0 = w - d0;                                                       // wraps to 0xFFFFFFFF for the previous slab's neighbour
#pragma unroll
      for (int u = 0; u < 4; u++) rl[u] = (r0 + (uint32_t)u < nv) ? ((pv[u] >> rb) - rbase + 1u) : spare;
    };
    auto mark_window = [&](const bool act, const int j, const uint32_t w, const uint32_t pv[4], uint32_t rl[4]) {
      uint32_t r0, nv, d1, dn;
      region_slots(act, j, w, pv, rl, r0, nv, d1, dn);
      if (!act) return;
      if GM_ABL(4) { if ((pv[0] ^ pv[1] ^ pv[2] ^ pv[3]) == 0x12345u) bm[0] = 1; return; }
      uint32_t old[4];
#pragma unroll
      for (int u = 0; u < 4; u++) old[u] = atomicOr(&bm[rl[u] >> 4], 1u << ((rl[u] & 15u) * 2u));
#pragma unroll
      for (int u = 0; u < 4; u++) if ((old[u] >> ((rl[u] & 15u) * 2u)) & 1u) atomicOr(&bm[rl[u] >> 4], 2u << ((rl[u] & 15u) * 2u));
      uint32_t strip = 0;                                                // overlap strip (ref: mapping.c:521-533): rare
#pragma unroll
      for (int u = 0; u < 4; u++) if ((pv[u] & rmask) < ovl && rl[u] != spare) strip |= 1u << u;
      if (strip) {
#pragma unroll
        for (int u = 0; u < 4; u++) if ((strip >> u) & 1u) k1_mark(bm, rl[u] - 1u);          // region 0's lands on the unused slot 0
      }
      if (S > 1) {
        const uint32_t un = nv - r0;                                     // lane element that is the next slab's first entry
        if (r0 == 0xFFFFFFFFu || (d1 < dn && un < 4u)) {
          const uint32_t* plist = pos0 + (long long)(((uint64_t)rec[3 * j + 1] << 32) | rec[3 * j]);
          if (r0 == 0xFFFFFFFFu) {
            // entries of the previous slab inside the last region before B count for local region 0
            if ((pv[0] >> rb) + 1u == rbase) {
              k1_mark(bm, 0u);
              for (uint32_t q = w; q > 0;) { --q; if ((plist[q] >> rb) + 1u != rbase) break; k1_mark(bm, 0u); }
            }
          }
          if (d1 < dn && un < 4u) {
            const uint32_t p = un == 0 ? pv[0] : (un == 1 ? pv[1] : (un == 2 ? pv[2] : pv[3]));
            // entries of the next slab inside the overlap strip count for this slab's last region
            if ((p >> rb) == rend && (p & rmask) < ovl) {
              k1_mark(bm, rend - rbase);
              for (uint32_t q = w + un + 1u; q < dn; q++) { const uint32_t pq = plist[q]; if ((pq >> rb) != rend || (pq & rmask) >= ovl) break; k1_mark(bm, rend - rbase); }
            }
          }
        }
      }
    };
    // survival test of a window whose region slots are known; every lane of the wave must call it (ballots)
    auto test_window = [&](const int j, const uint32_t pv[4], const uint32_t rl[4]) {
      uint32_t wv[4];
#pragma unroll
      for (int u = 0; u < 4; u++) wv[u] = bm[rl[u] >> 4];
      uint32_t hit = 0, strip = 0;
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const bool valid = rl[u] != spare;
        if (valid && ((wv[u] >> ((rl[u] & 15u) * 2u + 1u)) & 1u)) hit |= 1u << u;
        else if (valid && (pv[u] & rmask) < ovl && (pv[u] >> rb) > 0) strip |= 1u << u;
      }
      if (strip) {
#pragma unroll
        for (int u = 0; u < 4; u++) if (((strip >> u) & 1u) && k1_has2(bm, rl[u] - 1u)) hit |= 1u << u;
      }
      // one counter update per wave and window; slots are handed out element-major (any order will do: K2 sorts)
      const unsigned long long lt = (1ull << lane) - 1ull;
      uint32_t tot = 0, pre[4];
#pragma unroll
      for (int u = 0; u < 4; u++) { const unsigned long long bal = __ballot((hit >> u) & 1u); pre[u] = tot + (uint32_t)__popcll(bal & lt); tot += (uint32_t)__popcll(bal); }
      if (tot) {
        uint32_t basev = 0;
        if (lane == 0) basev = atomicAdd(&n_surv, tot);
        basev = __shfl(basev, 0);
        if (hit) {
          const uint32_t ysn = rec[3 * j + 2];
#pragma unroll
          for (int u = 0; u < 4; u++)
            if ((hit >> u) & 1u) { const uint32_t slot = basev + pre[u]; if (slot < scap) out[slot] = ((uint64_t)pv[u] << 32) | ysn; }   // sort key of K2: position, read offset y, seed
        }
      }
    };
    // window t of the flat array -> (list, start); false past the slice (no load issued); with_next: include the next slab's neighbour
    auto locate = [&](const int t, const bool with_next, int& j, uint32_t& w) -> bool {
      if (t >= nwin) return false;
      const uint32_t m = wmap[t];
      j = (int)(m & 0x7FFFu);
      const uint16_t* d = doff + (size_t)j * (S + 1);
      const uint32_t d0 = d[s], d1 = d[s + 1], dn = d[S];
      w = d0 - ((S > 1 && d0 > 0) ? 1u : 0u) + (m >> 15) * K1W + (uint32_t)gl4;
      return w < d1 + ((with_next && S > 1 && d1 < dn) ? 1u : 0u);
    };
    auto load_window = [&](const int j, const uint32_t w) -> k1_u32x4 {
      return *(const k1_u32x4*)(pos0 + (long long)(((uint64_t)rec[3 * j + 1] << 32) | rec[3 * j]) + w);
    };
    for (int phase = 0; phase < (GM_ABL(1) ? 1 : 2); phase++) {   // mark, then test
      for (int t0 = 0; t0 < nwin; t0 += K1Q * ng) {        // K1Q windows per lane group in flight
        int jq[K1Q]; uint32_t wq[K1Q], pq[K1Q][4]; bool aq[K1Q];
#pragma unroll
        for (int q = 0; q < K1Q; q++) {
          jq[q] = 0; wq[q] = 0;
          k1_u32x4 v = {0, 0, 0, 0};
          aq[q] = locate(t0 + q * ng + g, phase == 0, jq[q], wq[q]);
          if (aq[q] && !GM_ABL(16)) v = load_window(jq[q], wq[q]);
          pq[q][0] = v.x; pq[q][1] = v.y; pq[q][2] = v.z; pq[q][3] = v.w;
        }
#pragma unroll
        for (int q = 0; q < K1Q; q++) {
          uint32_t rl[4];
          if (phase == 0) mark_window(aq[q], jq[q], wq[q], pq[q], rl);
          else { uint32_t r0, nv, d1, dn; region_slots(aq[q], jq[q], wq[q], pq[q], rl, r0, nv, d1, dn); test_window(jq[q], pq[q], rl); }
        }
      }
      if (have_long) {                                      // third and later windows of long slices (repeats)
        for (int j0 = 0; j0 < nl; j0 += ng) {
          const int j = j0 + g;
          uint32_t w = 1u, w_hi = 0u;
          if (j < nl) {
            const uint16_t* d = doff + (size_t)j * (S + 1);
            const uint32_t d0 = d[s], d1 = d[s + 1], dn = d[S];
            w = d0 - ((S > 1 && d0 > 0) ? 1u : 0u) + 2u * K1W + (uint32_t)gl4;
            w_hi = d1 + ((phase == 0 && S > 1 && d1 < dn) ? 1u : 0u);
          }
          for (; __any(w < w_hi); w += K1W) {
            const bool act = w < w_hi;
            uint32_t pv[4] = {0, 0, 0, 0}, rl[4];
            if (act) { const k1_u32x4 v = load_window(j, w); pv[0] = v.x; pv[1] = v.y; pv[2] = v.z; pv[3] = v.w; }
            if (phase == 0) mark_window(act, j, w, pv, rl);
            else { uint32_t r0, nv, d1, dn; region_slots(act, j, w, pv, rl, r0, nv, d1, dn); test_window(j, pv, rl); }
          }
        }
      }
      __syncthreads();
    }
    if (surv_seg && tid == 0) surv_seg[(size_t)rs * (S + 1) + s + 1] = n_surv;         // survivors come out slab by slab: K1b prunes per slab
  }
  if (tid == 0) {
    surv_cnt[rs] = n_surv;
    GS_ADD(stats, GS_SURVIVORS, (unsigned long long)n_surv);
    if (n_surv > scap) {
      const uint32_t hs = atomicAdd(heavy_cnt, 1u);
      if (hs < (uint32_t)heavy_cap) heavy_list[hs] = (uint32_t)rs; else GS_ADD(stats, GS_OVERFLOW_SURV, 1ull);
    }
  }
  for (int d = GM_WAVE / 2; d > 0; d >>= 1) { my_lookups += __shfl_down(my_lookups, d); my_entries += __shfl_down(my_entries, d); }
  if ((tid & (GM_WAVE - 1)) == 0) { GS_ADD(stats, GS_LOOKUPS, my_lookups); GS_ADD(stats, GS_ENTRIES, my_entries); }
}

// ---------------------------------------------------------------------------------------------
// K1 v4 (large genomes, several slabs): hashed pre-count, exact count on the few candidates.
//
// The slab sweep above visits every list once per slab, so each visit handles a ~30-entry slice and most of its
// instructions are bookkeeping (rocprofv3: ~1.4 VALU wave-instructions per list entry and pass; VALU-issue bound).
// Here the region counters of the WHOLE genome are folded into one LDS table of 2^tab_bits 2-bit counters
// (counter = region mod 2^tab_bits).  Pass A streams every list once, whole, 32 entries per 8-lane group and
// instruction, and counts into the folded table; pass B streams them again and keeps the entries whose folded
// counter (or that of region-1 for an entry in the overlap strip) reached 2: the *candidates*.  A folded counter is
// >= the true counter of each region mapped to it, so every entry that marks a region with true count >= 2 is a
// candidate; hence for a candidate's regions the exact count taken over candidates alone equals the true count
// whenever that is >= 2, and stays < 2 otherwise.  Pass C therefore redoes the exact count per slab on the
// candidates only (a flat array, one entry per lane: ~10 % of the entries), with the reference's rule
// (ref: mapping.c:521-533,733-742), and emits the survivors slab by slab as the kernels above do.
// Slab borders: an entry of a slab's last region is also fed (mark only) to the next slab's list, whose bitmap has
// one extra counter below its first region; an overlap-strip entry of a slab's first region is fed (mark only) to
// the previous slab.  Workgroups are persistent (one per CU, looping over read-strands) so that each owns one
// candidate scratch in global memory (L2-resident).  A read-strand whose candidates do not fit its scratch bins is
// handed to the slab-sweep kernel in list mode (rare: repeats).
// ---------------------------------------------------------------------------------------------
#define K4Q 4                // windows per lane group and pipeline stage
#define K4_MO_SELF 0x8000u   // candidate copy in the NEXT slab's bin: marks its own region only (counter 0 there)
#define K4_MO_PREV 0x4000u   // candidate copy in the PREVIOUS slab's bin: marks region - 1 only (that slab's last region)

__global__ void __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(5, 5)))
k_lookup_v4(GmIndexDev ix, const uint32_t* __restrict__ reads, int n_reads, int read_len, int read_words, int max_n_kmers,
            int NL, int tab_bits, int cbits, int SC, int wcap, uint64_t* __restrict__ surv, uint32_t* __restrict__ surv_cnt, int scap_all,
            uint32_t* __restrict__ heavy_list, uint32_t* __restrict__ heavy_cnt, int heavy_cap, unsigned long long* __restrict__ stats,
            uint32_t* __restrict__ surv_seg, uint64_t* __restrict__ scratch, int bin_cap, uint32_t* __restrict__ fb_list, uint32_t* __restrict__ fb_cnt,
            int fb_cap, int ablate, uint32_t* __restrict__ start_flags, uint32_t start_epoch) {
  // resident: tell the host (pinned memory), which holds the other stream's pass-1 launch back until every workgroup of this grid has a CU
  if (start_flags && threadIdx.x == 0) __hip_atomic_store(&start_flags[blockIdx.x], start_epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  __builtin_amdgcn_s_setprio(3);                            // memory-latency-bound: issue first when ready; VALU-bound kernels of the other stream fill the gaps

  extern __shared__ __align__(16) uint32_t smem[];
  __shared__ uint32_t n_surv, n_lists, n_win, overflow, pair_cnt[2];
  const int tid = threadIdx.x, nthr = blockDim.x, lane = tid & (GM_WAVE - 1);
  const int S = ix.n_slabs, rb = ix.region_bits;
  const uint32_t rmask = (1u << rb) - 1u, ovl = (uint32_t)ix.region_overlap;
  const uint32_t R = 1u << (cbits - rb);                     // regions per exact-count bin (a bin = 2^cbits positions, cbits <= slab_bits)
  const uint32_t tmask = (1u << tab_bits) - 1u;
  // LDS: codes | rec[NL] (4 words: list offset lo/hi from seed 0's array, length, y << 16 | seed) | wbase[NL + 1] | bin_cnt[SC] | wmap[wcap] (u16) | table
  uint8_t* codes = (uint8_t*)smem;
  const int code_words = (read_len + 3) / 4;
  uint32_t* rec = smem + ((code_words + 3) & ~3);
  uint32_t* wbase = rec + 4 * NL;
  uint32_t* bin_cnt = wbase + NL + 1;
  uint16_t* wmap = (uint16_t*)(bin_cnt + SC);
  uint32_t* tab = smem + ((((code_words + 3) & ~3) + 4 * NL + NL + 1 + SC + (wcap + 1) / 2 + 3) & ~3);
  const int tab_words = (1 << (tab_bits - 4)) + 4;            // + the spare counters of the empty lanes
  const uint32_t* __restrict__ pos0 = ix.seed[0].pos;
  uint64_t* my_scratch = scratch + (size_t)blockIdx.x * SC * bin_cap;
  const int bin_shift = 31 - __builtin_clz((unsigned)bin_cap);        // bin_cap is a power of two (k4_launch)
  const int ng = nthr / 8, g = tid / 8, gl4 = (tid & 7) * 4;
  unsigned long long my_lookups = 0, my_entries = 0;
  bool tab_clean = false;                                   // uniform over the workgroup

  for (int rs = blockIdx.x; rs < 2 * n_reads; rs += gridDim.x) {
    const int rd = rs >> 1, st = rs & 1;
    uint64_t* out = surv + (size_t)rs * scap_all;
    const uint32_t scap = (uint32_t)scap_all;
    __syncthreads();                                         // the previous read-strand is done with the LDS
    const uint32_t* rw = reads + (size_t)rd * read_words;
    for (int i = tid; i < read_len; i += nthr) codes[i] = (uint8_t)gm_read_code(rw, read_len, st ^ ix.cs_flip, ix.colour, i, ix.read_rna && ix.read_rna[rd]);
    // pass C un-marks every word it touched, so after a read-strand that went through it the table is already clear
    if (!tab_clean) { uint4* t4 = (uint4*)tab; for (int w = tid; w < (tab_words >> 2); w += nthr) t4[w] = make_uint4(0, 0, 0, 0); }
    tab_clean = false;
    for (int c = tid; c < SC; c += nthr) bin_cnt[c] = 0;
    if (tid == 0) { n_surv = 0; n_lists = 0; overflow = 0; pair_cnt[0] = 0; pair_cnt[1] = 0; }
    __syncthreads();
    // ---- map indexes and whole-list bounds (ref: mapping.c:53-66, KMER_TO_MAPIDX gmapper.h:349-368) ----
    for (int off = tid; off < NL; off += nthr) {
      const int sn = off / max_n_kmers, i = off - sn * max_n_kmers;
      const int span = ix.seed[sn].span;
      if (i < ix.colour || i + span > read_len) continue;
      const uint64_t mask = ix.seed[sn].mask;
      const uint32_t mapidx = gm_mapidx(ix, mask, span, codes + i);
      const uint32_t* dir = ix.seed[sn].dir + (size_t)mapidx * (uint32_t)S;
      my_lookups++;
      const uint32_t b = dir[0], e = dir[S];
      if (e == b || e - b > ix.list_cutoff) continue;        // ref: mapping.c:497 (longer lists are skipped, not deleted)
      my_entries += (e - b);
      const uint32_t j = atomicAdd(&n_lists, 1u);
      const uint64_t ptr = (uint64_t)((ix.seed[sn].pos + b) - pos0);
      rec[4 * j] = (uint32_t)ptr; rec[4 * j + 1] = (uint32_t)(ptr >> 32); rec[4 * j + 2] = e - b; rec[4 * j + 3] = ((uint32_t)i << 16) | (uint32_t)sn;
    }
    __syncthreads();
    const int nl = (int)n_lists;
    // ---- windows of 32 entries: prefix over the lists, and the window -> list map of the first wcap windows (wave 0) ----
    if (tid < GM_WAVE) {
      const int per = (nl + GM_WAVE - 1) / GM_WAVE;
      const int a0 = min(nl, lane * per), a1 = min(nl, a0 + per);
      uint32_t sum = 0;
      for (int a = a0; a < a1; a++) sum += (rec[4 * a + 2] + 31u) >> 5;
      uint32_t incl = sum;
      for (int dd = 1; dd < GM_WAVE; dd <<= 1) { const uint32_t o = __shfl_up(incl, dd); if (lane >= dd) incl += o; }
      uint32_t run = incl - sum;
      for (int a = a0; a < a1; a++) {
        wbase[a] = run;
        const uint32_t c = (rec[4 * a + 2] + 31u) >> 5;
        for (uint32_t k = 0; k < c && run + k < (uint32_t)wcap; k++) wmap[run + k] = (uint16_t)a;
        run += c;
      }
      if (lane == GM_WAVE - 1) { wbase[nl] = incl; n_win = incl; }
    }
    __syncthreads();
    const int nwin = (int)n_win;
    // window t -> list j, first entry e0 of this lane, number of this lane's valid entries (0..4)
    auto locate = [&](const int t, int& j, uint32_t& e0) -> int {
      if (t >= nwin) return 0;
      if (t < wcap) j = (int)wmap[t];
      else { int lo = 0, hi = nl; while (hi - lo > 1) { const int m = (lo + hi) >> 1; if (wbase[m] <= (uint32_t)t) lo = m; else hi = m; } j = lo; }
      e0 = ((uint32_t)t - wbase[j]) * 32u + (uint32_t)gl4;
      const uint32_t len = rec[4 * j + 2];
      return e0 >= len ? 0 : (int)min(4u, len - e0);
    };
    auto load_window = [&](const int j, const uint32_t e0) -> k1_u32x4 {
      return *(const k1_u32x4*)(pos0 + (long long)(((uint64_t)rec[4 * j + 1] << 32) | rec[4 * j]) + e0);
    };
    // Both passes are software pipelines: the loads of the next K4Q windows are in flight while this group's current ones are
    // processed (one workgroup per CU: without it the memory system idles during every processing stretch).
    struct Win { int j, nv; uint32_t p[4]; };
    auto fetch = [&](const int t, Win& w) {
      uint32_t e0 = 0; w.j = 0;
      w.nv = locate(t, w.j, e0);
      k1_u32x4 v = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
      if (w.nv) v = load_window(w.j, e0);
      // 0xFFFFFFFF = no entry (never a k-mer start); the passes below are straight-line code over all four slots
      w.p[0] = v.x; w.p[1] = w.nv > 1 ? v.y : 0xFFFFFFFFu; w.p[2] = w.nv > 2 ? v.z : 0xFFFFFFFFu; w.p[3] = w.nv > 3 ? v.w : 0xFFFFFFFFu;
    };
    // ---- pass A: folded counts ----
    if (!GM_ABL(8)) {
      Win cur[K4Q], nxt[K4Q];
#pragma unroll
      for (int q = 0; q < K4Q; q++) fetch(q * ng + g, cur[q]);
      for (int t0 = 0; t0 < nwin; t0 += K4Q * ng) {
#pragma unroll
        for (int q = 0; q < K4Q; q++) fetch(t0 + (K4Q + q) * ng + g, nxt[q]);
#pragma unroll
        for (int q = 0; q < K4Q; q++) {
          const Win& w = cur[q];
          if GM_ABL(2) { if ((w.p[0] ^ w.p[1] ^ w.p[2] ^ w.p[3]) == 0x12345u) tab[0] = 1; continue; }   // loads only
          // Folded counts only have to be >= the true ones (the exact rule is applied in pass C), so this pass takes two short cuts: an empty lane
          // (0xFFFFFFFF) counts into the folded counter of its pseudo-region instead of a spare word, and a strip entry of region 0 marks
          // counter (0 - 1) & tmask.  Both can only add candidates.
          uint32_t old[4], h[4];
#pragma unroll
          for (int u = 0; u < 4; u++) { h[u] = (w.p[u] >> rb) & tmask; old[u] = atomicOr(&tab[h[u] >> 4], 1u << ((h[u] & 15u) * 2u)); }
#pragma unroll
          for (int u = 0; u < 4; u++) if (((old[u] >> ((h[u] & 15u) * 2u)) & 3u) == 1u) atomicOr(&tab[h[u] >> 4], 2u << ((h[u] & 15u) * 2u));
          // overlap strip (ref: mapping.c:521-533): 2.4 % of the entries (0xFFFFFFFF is never in it) -- one rarely taken loop for the four
          uint32_t sm = 0;
#pragma unroll
          for (int u = 0; u < 4; u++) sm |= ((w.p[u] & rmask) < ovl) ? (1u << u) : 0u;
          while (sm) {
            const int u = __builtin_ctz(sm); sm &= sm - 1u;
            const uint32_t pu = u == 0 ? w.p[0] : (u == 1 ? w.p[1] : (u == 2 ? w.p[2] : w.p[3]));
            k1_mark(tab, ((pu >> rb) - 1u) & tmask);
          }
        }
#pragma unroll
        for (int q = 0; q < K4Q; q++) cur[q] = nxt[q];
      }
    }
    __syncthreads();
    // ---- pass B: candidates = entries whose folded counter reached 2, binned by position ----
    if (!GM_ABL(1)) {
      Win cur[K4Q], nxt[K4Q];
#pragma unroll
      for (int q = 0; q < K4Q; q++) fetch(q * ng + g, cur[q]);
      for (int t0 = 0; t0 < nwin; t0 += K4Q * ng) {
#pragma unroll
        for (int q = 0; q < K4Q; q++) fetch(t0 + (K4Q + q) * ng + g, nxt[q]);
#pragma unroll
        for (int q = 0; q < K4Q; q++) {
          const Win& w = cur[q];
          uint32_t hit = 0, sm = 0;
#pragma unroll
          for (int u = 0; u < 4; u++) {
            const uint32_t r = w.p[u] >> rb;
            hit |= k1_has2(tab, r & tmask) ? (1u << u) : 0u;
            sm |= ((w.p[u] & rmask) < ovl) ? (1u << u) : 0u;
          }
          sm &= ~hit;
          while (sm) {                                             // strip entries whose own counter is below 2: the counter of region - 1 decides
            const int u = __builtin_ctz(sm); sm &= sm - 1u;
            const uint32_t pu = u == 0 ? w.p[0] : (u == 1 ? w.p[1] : (u == 2 ? w.p[2] : w.p[3]));
            if (k1_has2(tab, ((pu >> rb) - 1u) & tmask)) hit |= 1u << u;
          }
          hit &= (1u << w.nv) - 1u;                                // empty lanes (their pseudo-region's counter may have reached 2) are no entries
          if (hit && GM_ABL(16)) { if (hit == 0x55u) tab[1] = 1; hit = 0; }
          if (hit) {
            const uint32_t ysn = rec[4 * w.j + 3];
            while (hit) {
              const int u = __builtin_ctz(hit); hit &= hit - 1u;
              const uint32_t p = u == 0 ? w.p[0] : (u == 1 ? w.p[1] : (u == 2 ? w.p[2] : w.p[3]));
              const uint32_t r = p >> rb, sl = p >> cbits, rloc = r & (R - 1u);
              const uint64_t key = ((uint64_t)p << 32) | ysn;
              uint32_t idx = atomicAdd(&bin_cnt[sl], 1u);
              if (idx < (uint32_t)bin_cap) my_scratch[(sl << bin_shift) + idx] = key; else overflow = 1u;
              if (((rloc + 1u) & (R - 1u)) > 1u) continue;         // not in a bin's first or last region: no border copy
              if (rloc == R - 1u && (int)sl + 1 < SC) {          // also counts for the next bin's counter 0
                idx = atomicAdd(&bin_cnt[sl + 1], 1u);
                if (idx < (uint32_t)bin_cap) my_scratch[((sl + 1u) << bin_shift) + idx] = key | K4_MO_SELF; else overflow = 1u;
              }
              if (rloc == 0u && sl > 0 && (p & rmask) < ovl) {   // its strip mark belongs to the previous bin's last region
                idx = atomicAdd(&bin_cnt[sl - 1], 1u);
                if (idx < (uint32_t)bin_cap) my_scratch[((sl - 1u) << bin_shift) + idx] = key | K4_MO_PREV; else overflow = 1u;
              }
            }
          }
        }
#pragma unroll
        for (int q = 0; q < K4Q; q++) cur[q] = nxt[q];
      }
    }
    __syncthreads();
    if (overflow) {                                            // bins too small for this read-strand: the slab-sweep kernel redoes it
      if (tid == 0) {
        const uint32_t f = atomicAdd(fb_cnt, 1u);
        if (f < (uint32_t)fb_cap) fb_list[f] = (uint32_t)rs; else GS_ADD(stats, GS_OVERFLOW_SURV, 1ull);
      }
      continue;
    }
    // ---- pass C: exact counts per slab over the candidates (local counter = region - first region of the slab + 1) ----
    if GM_ABL(4) continue;
    __threadfence_block();
    { uint4* t4 = (uint4*)tab; for (int w = tid; w < (tab_words >> 2); w += nthr) t4[w] = make_uint4(0, 0, 0, 0); }
    __syncthreads();
    const int per_slab = 1 << (ix.slab_bits - cbits);        // bins per index slab
    const uint32_t half = (R + 2u + 15u) & ~15u;             // counters of one bin, in whole table words: two bins fit the table
    for (int s = 0; s < SC; s++) {
      // Two bins per round when each has at most one candidate per thread (the usual case): their bitmaps sit side by side in the table,
      // the candidates are read once and stay in registers, and the survivors' slots come from per-bin counts -- three barriers for two bins.
      if (s + 1 < SC && bin_cnt[s] <= (uint32_t)nthr && bin_cnt[s + 1] <= (uint32_t)nthr) {
        const uint32_t ncA = bin_cnt[s], ncB = bin_cnt[s + 1];
        const uint64_t keyA = (uint32_t)tid < ncA ? my_scratch[((uint32_t)s << bin_shift) + (uint32_t)tid] : 0ull;
        const uint64_t keyB = (uint32_t)tid < ncB ? my_scratch[(((uint32_t)s + 1u) << bin_shift) + (uint32_t)tid] : 0ull;
        const uint32_t base0 = n_surv;                           // stable since the last barrier
        bool hitv[2] = {false, false}; uint32_t rank[2] = {0, 0}, locv[2] = {0, 0};
#pragma unroll
        for (int b = 0; b < 2; b++) {
          const uint64_t key = b ? keyB : keyA;
          if ((uint32_t)tid < (b ? ncB : ncA)) {
            const uint32_t p = (uint32_t)(key >> 32), fl = (uint32_t)key;
            const uint32_t loc = (p >> rb) - (uint32_t)(s + b) * R + 1u + (uint32_t)b * half;   // MO_SELF copy: 0; MO_PREV copy: R + 1
            locv[b] = loc;
            if (fl & K4_MO_PREV) k1_mark(tab, loc - 1u);
            else {
              k1_mark(tab, loc);
              if (!(fl & K4_MO_SELF) && (p & rmask) < ovl && (p >> rb) > 0) k1_mark(tab, loc - 1u);
            }
          }
        }
        __syncthreads();
#pragma unroll
        for (int b = 0; b < 2; b++) {
          const uint64_t key = b ? keyB : keyA;
          bool hit = false;
          if ((uint32_t)tid < (b ? ncB : ncA)) {
            const uint32_t p = (uint32_t)(key >> 32), fl = (uint32_t)key;
            if (!(fl & (K4_MO_SELF | K4_MO_PREV))) hit = k1_has2(tab, locv[b]) || ((p & rmask) < ovl && (p >> rb) > 0 && k1_has2(tab, locv[b] - 1u));
          }
          const unsigned long long bal = __ballot(hit);
          uint32_t basev = 0;
          if (bal) {
            if (lane == 0) basev = atomicAdd(&pair_cnt[b], (uint32_t)__popcll(bal));
            basev = __shfl(basev, 0);
          }
          hitv[b] = hit; rank[b] = basev + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull));
        }
        __syncthreads();
        const uint32_t cA = pair_cnt[0], cB = pair_cnt[1];
        if (hitv[0]) { const uint32_t slot = base0 + rank[0]; if (slot < scap) out[slot] = keyA; }
        if (hitv[1]) { const uint32_t slot = base0 + cA + rank[1]; if (slot < scap) out[slot] = keyB; }
        if ((uint32_t)tid < ncA) { tab[locv[0] >> 4] = 0; if (locv[0]) tab[(locv[0] - 1u) >> 4] = 0; }
        if ((uint32_t)tid < ncB) { tab[locv[1] >> 4] = 0; tab[(locv[1] - 1u) >> 4] = 0; }
        __syncthreads();
        if (tid == 0) {
          n_surv = base0 + cA + cB; pair_cnt[0] = 0; pair_cnt[1] = 0;
          if (surv_seg) {
            if (((s + 1) % per_slab) == 0) surv_seg[(size_t)rs * (S + 1) + s / per_slab + 1] = base0 + cA;
            if (((s + 2) % per_slab) == 0 || s + 2 == SC) surv_seg[(size_t)rs * (S + 1) + (s + 1) / per_slab + 1] = base0 + cA + cB;
          }
        }
        __syncthreads();
        s++;
        continue;
      }
      const int nc = (int)bin_cnt[s];
      const uint64_t* cand = my_scratch + (size_t)s * bin_cap;
      const uint32_t rbase = (uint32_t)s * R;
      for (int i = tid; i < nc; i += nthr) {
        const uint64_t key = cand[i]; const uint32_t p = (uint32_t)(key >> 32), fl = (uint32_t)key;
        const uint32_t loc = (p >> rb) - rbase + 1u;             // MO_SELF copy: 0; MO_PREV copy: R + 1
        if (fl & K4_MO_PREV) k1_mark(tab, loc - 1u);
        else {
          k1_mark(tab, loc);
          if (!(fl & K4_MO_SELF) && (p & rmask) < ovl && (p >> rb) > 0) k1_mark(tab, loc - 1u);
        }
      }
      __syncthreads();
      for (int i0 = 0; i0 < nc; i0 += nthr) {
        const int i = i0 + tid;
        bool hit = false; uint64_t key = 0;
        if (i < nc) {
          key = cand[i]; const uint32_t p = (uint32_t)(key >> 32), fl = (uint32_t)key;
          if (!(fl & (K4_MO_SELF | K4_MO_PREV))) {
            const uint32_t loc = (p >> rb) - rbase + 1u;
            hit = k1_has2(tab, loc) || ((p & rmask) < ovl && (p >> rb) > 0 && k1_has2(tab, loc - 1u));
          }
        }
        const unsigned long long bal = __ballot(hit);
        if (bal) {
          uint32_t basev = 0;
          if (lane == 0) basev = atomicAdd(&n_surv, (uint32_t)__popcll(bal));
          basev = __shfl(basev, 0);
          if (hit) { const uint32_t slot = basev + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull)); if (slot < scap) out[slot] = key; }
        }
      }
      __syncthreads();
      for (int i = tid; i < nc; i += nthr) {                   // un-mark: zero the words this slab touched
        const uint32_t loc = ((uint32_t)(cand[i] >> 32) >> rb) - rbase + 1u;
        tab[loc >> 4] = 0; if (loc) tab[(loc - 1u) >> 4] = 0;
      }
      if (surv_seg && tid == 0 && (((s + 1) % per_slab) == 0 || s + 1 == SC)) surv_seg[(size_t)rs * (S + 1) + s / per_slab + 1] = n_surv;   // survivors come out slab by slab
      __syncthreads();
    }
    tab_clean = !GM_ABL(~0);
    if (tid == 0) {
      if (surv_seg) surv_seg[(size_t)rs * (S + 1)] = 0;
      surv_cnt[rs] = n_surv;
      GS_ADD(stats, GS_SURVIVORS, (unsigned long long)n_surv);
      if (n_surv > scap) {
        const uint32_t hs = atomicAdd(heavy_cnt, 1u);
        if (hs < (uint32_t)heavy_cap) heavy_list[hs] = (uint32_t)rs; else GS_ADD(stats, GS_OVERFLOW_SURV, 1ull);
      }
    }
  }
  for (int d = GM_WAVE / 2; d > 0; d >>= 1) { my_lookups += __shfl_down(my_lookups, d); my_entries += __shfl_down(my_entries, d); }
  if ((tid & (GM_WAVE - 1)) == 0) { GS_ADD(stats, GS_LOOKUPS, my_lookups); GS_ADD(stats, GS_ENTRIES, my_entries); }
}

// start flags of the persistent K1 grid (see gm_host.hip, pipeline_back): set before a launch, consumed by it
// Launch state of the CALLING THREAD (thread_local: a session's front pipeline sets the flags, launches and reads the grid back on one host thread; two threads that map
// on two devices side by side each keep their own -- round 3 advisor: as process-wide statics one session could launch with the other's pinned flag pointer and epoch).
static thread_local uint32_t* g_k4_flags = nullptr; static thread_local uint32_t g_k4_epoch = 0; static thread_local int g_k4_flag_cap = 0, g_k4_flag_grid = 0;
void gm_lookup_set_start_flags(uint32_t* flags, int cap, uint32_t epoch) { g_k4_flags = flags; g_k4_flag_cap = cap; g_k4_epoch = epoch; g_k4_flag_grid = 0; }
int gm_lookup_start_flag_grid(void) { return g_k4_flag_grid; }   // workgroups that will raise a flag for the last launch (0: none)

static void k1_geometry(const GmIndexDev& ix, int read_len, int* max_n_kmers, int* NL, int* bm_words, size_t* lds) {
  *max_n_kmers = read_len - ix.min_seed_span + 1;
  if (*max_n_kmers < 0) *max_n_kmers = 0;
  *NL = ix.n_seeds * (*max_n_kmers);
  uint64_t slab_len = (ix.n_slabs == 1) ? ix.total_len : (1ull << ix.slab_bits);
  uint64_t regions = (slab_len >> ix.region_bits) + 3;      // +1 region before, +1 partial, +1 slack
  *bm_words = (int)((regions + 15) / 16);
  *bm_words = (*bm_words + 3) & ~3;
  *lds = (size_t)((((read_len + 3) / 4) + 6 * (*NL) + 1 + 3) & ~3) * 4 + (size_t)(*bm_words) * 4;
}

// v4 launch: returns false when the geometry does not fit (the caller falls back to the slab-sweep kernels)
struct K4Scratch { uint64_t* scratch = nullptr; size_t words = 0; uint32_t* fb = nullptr; int cus = 0; };
static K4Scratch g_k4[16][2];                                   // (two sets per device, see gm_lookup_set_scratch_slot)
static thread_local int g_k4_slot = 0;
void gm_lookup5_set_scratch_slot(int slot);
// The lookup kernels' scratch (fall-back lists, the rounds kernel's rows, v4's candidate bins) exists twice per device: a mapping call runs all its launches with the set
// its thread was given, so that two calls -- two sessions -- can be in flight on one device (gm_host.hip hands the sets out).
void gm_lookup_set_scratch_slot(int slot) { g_k4_slot = slot & 1; gm_lookup5_set_scratch_slot(slot & 1); }
static bool k4_launch(const GmIndexDev& ix, const uint32_t* d_reads, int n_reads, int read_len, int read_words, int max_n_kmers, int NL,
                      uint64_t* d_surv, uint32_t* d_surv_cnt, int scap, uint32_t* d_heavy_list, uint32_t* d_heavy_cnt, int heavy_cap,
                      unsigned long long* d_stats, hipStream_t stream, uint32_t* d_surv_seg, size_t lds_generic, int bm_words) {
  int dev = 0; if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return false;
  K4Scratch& K = g_k4[dev][g_k4_slot];
  const int S = ix.n_slabs;
  {  // v4's fixed cost per read-strand (two 128 KB table clears, per-bin passes) pays off from ~30 k list entries per read-strand
     // (measured: 45 k at 100 bp / 3 Gbp 7 % faster than the slab sweep, 17 k at 50 colours 40 % slower)
    double entries = 0;
    for (int sn = 0; sn < ix.n_seeds; sn++) {
      const double lists = std::max(0, read_len - ix.seed[sn].span + 1 - ix.colour);
      entries += lists * (double)ix.seed[sn].n_pos / (double)(1ull << (ix.hflag ? 2 * GM_HASH_TABLE_POWER : 2 * ix.seed[sn].weight));
    }
    if (entries < 30000.0 && !gm_tune("GM_K1_V4")) return false;
  }
  int tab_bits = 19; if (const char* e = gm_tune("GM_K4_TABBITS")) tab_bits = std::max(6, std::min(19, atoi(e)));
  // pass C keeps the exact counters of one bin of 2^cbits positions (+2) in the table's LDS; bins nest inside the index slabs
  const int cbits = std::min(ix.slab_bits, tab_bits + ix.region_bits - 1);
  if (cbits < ix.region_bits + 1) return false;
  const int SC = (int)((ix.total_len + (1ull << cbits) - 1) >> cbits);
  if (SC < 1 || SC > 4096) return false;
  int wcap = 4096; if (const char* e = gm_tune("GM_K4_WCAP")) wcap = std::max(2, std::min(4096, atoi(e) & ~1));   // windows with a direct window -> list entry (the rest: binary search)
  const int fb_cap = 4096;
  int bin_cap = 4096; if (const char* e = gm_tune("GM_K4_BINCAP")) { const int v = std::max(16, std::min(1 << 20, atoi(e))); bin_cap = 16; while (bin_cap < v) bin_cap <<= 1; }
  const int code_words = (read_len + 3) / 4;
  size_t lds = (size_t)(((((code_words + 3) & ~3) + 4 * NL + NL + 1 + SC + (wcap + 1) / 2 + 3) & ~3) + (1 << (tab_bits - 4)) + 4) * 4;
  if (const char* e = gm_tune("GM_K4_LDS_PAD")) lds += (size_t)std::max(0, atoi(e));   // experiment: what the LDS footprint does to the co-residency with pass 1 / pass 2
  if (lds > 160 * 1024 - 64) return false;
  int wgs_per_cu = 1; if (const char* e = gm_tune("GM_K4_WGS")) wgs_per_cu = std::max(1, std::min(8, atoi(e)));
  if (!K.cus) { if (hipDeviceGetAttribute(&K.cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || K.cus < 1) K.cus = 256; }
  int grid = std::min(2 * n_reads, K.cus * wgs_per_cu);
  if (const char* e = gm_tune("GM_K4_GRID")) grid = std::max(1, std::min(2 * n_reads, atoi(e)));
  const size_t need = (size_t)std::max(grid, K.cus) * SC * bin_cap;
  if (need > K.words) {
    if (K.scratch) (void)hipFree(K.scratch);
    K.scratch = nullptr; K.words = 0;
    if (hipMalloc(&K.scratch, need * 8) != hipSuccess) return false;
    K.words = need;
  }
  if (!K.fb) { if (hipMalloc(&K.fb, (size_t)(fb_cap + 4) * 4) != hipSuccess) return false; }
  if (hipMemsetAsync(K.fb + fb_cap, 0, 4, stream) != hipSuccess) return false;
  static GmLdsLimit lim_configured4, lim_configured_g; size_t &configured4 = lim_configured4.cur(), &configured_g = lim_configured_g.cur();
  if (lds > 48 * 1024 && lds > configured4) { if (hipFuncSetAttribute((const void*)k_lookup_v4, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return false; configured4 = lds; }
  if (lds_generic > 48 * 1024 && lds_generic > configured_g) {
    if (hipFuncSetAttribute((const void*)k_lookup<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_generic) != hipSuccess) return false; configured_g = lds_generic; }
  int k4_threads = 1024; if (const char* e = gm_tune("GM_K1_THREADS")) k4_threads = std::max(64, std::min(1024, atoi(e) & ~63));
  const bool use_flags = g_k4_flags && grid <= g_k4_flag_cap;
  g_k4_flag_grid = use_flags ? grid : 0;
  hipLaunchKernelGGL(k_lookup_v4, dim3(grid), dim3(k4_threads), lds, stream, ix, d_reads, n_reads, read_len, read_words, max_n_kmers, NL, tab_bits, cbits, SC, wcap,
                     d_surv, d_surv_cnt, scap, d_heavy_list, d_heavy_cnt, heavy_cap, d_stats, d_surv_seg, K.scratch, bin_cap, K.fb, K.fb + fb_cap, fb_cap,
                     gm_tune("GM_K1_ABLATE") ? atoi(gm_tune("GM_K1_ABLATE")) : 0, use_flags ? g_k4_flags : nullptr, g_k4_epoch);
  // read-strands whose candidates overflowed their bins: the slab-sweep kernel in list mode (blocks beyond the list's end return at once)
  hipLaunchKernelGGL(k_lookup<false>, dim3(std::min(fb_cap, 1024)), dim3(K1_THREADS), lds_generic, stream, ix, d_reads, n_reads, read_len, read_words,
                     max_n_kmers, NL, bm_words, d_surv, d_surv_cnt, scap, d_heavy_list, d_heavy_cnt, heavy_cap,
                     (const uint32_t*)nullptr, (const uint64_t*)nullptr, (const uint32_t*)K.fb, (const uint32_t*)(K.fb + fb_cap), d_stats, 0, d_surv_seg);
  return true;
}

size_t gm_lookup_lds_bytes(const GmIndexDev& ix, int read_len) {
  int a, b, c; size_t l; k1_geometry(ix, read_len, &a, &b, &c, &l); return l;
}

// the mate-pair modes of the generic kernel: their dynamic LDS limit (set once per device for both instantiations)
static int k1_mp_lds(size_t& lds) {
  lds += (size_t)2 * GM_MP_CAP * 4;                                   // the mate's row and this read-strand's own
  if (lds + 64 > 160 * 1024) { gm_set_error("lookup kernel (mate-pair modes) needs %zu bytes of LDS", lds); return GM_E_ARG; }
  static GmLdsLimit lim_mp; size_t& configured = lim_mp.cur();
  if (lds > 48 * 1024 && lds > configured) {
    GM_HIP(hipFuncSetAttribute((const void*)k_lookup<false, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    GM_HIP(hipFuncSetAttribute((const void*)k_lookup<false, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    GM_HIP(hipFuncSetAttribute((const void*)k_lookup<false, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    GM_HIP(hipFuncSetAttribute((const void*)k_lookup<false, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    GM_HIP(hipFuncSetAttribute((const void*)k_lookup<false, 5>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    configured = lds;
  }
  return GM_OK;
}

static thread_local const char* g_k1_name = "";
extern "C" const char* gm_last_lookup_kernel(void) { return g_k1_name; }   // which K1 variant the calling thread's last gm_launch_lookup chose (for bench.py / profiles)

int gm_launch_lookup(const GmIndexDev& ix, const uint32_t* d_reads, int n_reads, int read_len, int read_words,
                     uint64_t* d_surv, uint32_t* d_surv_cnt, int scap, uint32_t* d_heavy_list, uint32_t* d_heavy_cnt, int heavy_cap,
                     unsigned long long* d_stats, hipStream_t stream, uint32_t* d_surv_seg, const GmFusePrune* fuse) {
  int max_n_kmers, NL, bm_words; size_t lds;
  k1_geometry(ix, read_len, &max_n_kmers, &NL, &bm_words, &lds);
  if (lds > 160 * 1024) { gm_set_error("lookup kernel needs %zu bytes of LDS (read_len %d, slab_bits %d)", lds, read_len, ix.slab_bits); return GM_E_ARG; }
  GM_HIP(hipMemsetAsync(d_heavy_cnt, 0, 4, stream));
  if (fuse && fuse->fused) *fuse->fused = 0;
  if (NL == 0 || n_reads == 0) { GM_HIP(hipMemsetAsync(d_surv_cnt, 0, (size_t)n_reads * 2 * 4, stream)); return GM_OK; }
  static GmLdsLimit lim_configured; size_t& configured = lim_configured.cur();
  if (lds > 48 * 1024 && lds > configured) {
    GM_HIP(hipFuncSetAttribute((const void*)k_lookup<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    configured = lds;
  }
  if (ix.mp.mode) {   // paired -n 3: the generic slab-sweep kernel in one of its mate-pair modes (GmMpDev)
    int rc = k1_mp_lds(lds); if (rc) return rc;
    const int k1_threads = std::min(1024, std::max(256, (NL + 63) & ~63));
    g_k1_name = "k_lookup";
    if (ix.mp.mode == 1)
      hipLaunchKernelGGL((k_lookup<false, 1>), dim3(n_reads * 2), dim3(k1_threads), lds, stream, ix, d_reads, n_reads, read_len, read_words,
                         max_n_kmers, NL, bm_words, d_surv, d_surv_cnt, scap, d_heavy_list, d_heavy_cnt, heavy_cap,
                         (const uint32_t*)nullptr, (const uint64_t*)nullptr, (const uint32_t*)nullptr, (const uint32_t*)nullptr, d_stats, 0, (uint32_t*)nullptr);
    else if (ix.mp.mode == 3)
      hipLaunchKernelGGL((k_lookup<false, 3>), dim3(n_reads * 2), dim3(k1_threads), lds, stream, ix, d_reads, n_reads, read_len, read_words,
                         max_n_kmers, NL, bm_words, d_surv, d_surv_cnt, scap, d_heavy_list, d_heavy_cnt, heavy_cap,
                         (const uint32_t*)nullptr, (const uint64_t*)nullptr, (const uint32_t*)nullptr, (const uint32_t*)nullptr, d_stats, 0, (uint32_t*)nullptr);
    else if (ix.mp.mode == 5)
      hipLaunchKernelGGL((k_lookup<false, 5>), dim3(n_reads * 2), dim3(k1_threads), lds, stream, ix, d_reads, n_reads, read_len, read_words,
                         max_n_kmers, NL, bm_words, d_surv, d_surv_cnt, scap, d_heavy_list, d_heavy_cnt, heavy_cap,
                         (const uint32_t*)nullptr, (const uint64_t*)nullptr, (const uint32_t*)nullptr, (const uint32_t*)nullptr, d_stats, 0, d_surv_seg);
    else if (ix.mp.mode == 4)
      hipLaunchKernelGGL((k_lookup<false, 4>), dim3(n_reads * 2), dim3(k1_threads), lds, stream, ix, d_reads, n_reads, read_len, read_words,
                         max_n_kmers, NL, bm_words, d_surv, d_surv_cnt, scap, d_heavy_list, d_heavy_cnt, heavy_cap,
                         (const uint32_t*)nullptr, (const uint64_t*)nullptr, (const uint32_t*)nullptr, (const uint32_t*)nullptr, d_stats, 0, d_surv_seg);
    else
      hipLaunchKernelGGL((k_lookup<false, 2>), dim3(n_reads * 2), dim3(k1_threads), lds, stream, ix, d_reads, n_reads, read_len, read_words,
                         max_n_kmers, NL, bm_words, d_surv, d_surv_cnt, scap, d_heavy_list, d_heavy_cnt, heavy_cap,
                         (const uint32_t*)nullptr, (const uint64_t*)nullptr, (const uint32_t*)nullptr, (const uint32_t*)nullptr, d_stats, 0, d_surv_seg);
    GM_HIP(hipGetLastError());
    return GM_OK;
  }
  const bool all = ix.no_region_counts != 0;      // every list entry survives: the generic slab-sweep kernel carries that switch, the filtering kernels do not apply
  const bool bkt = !all && ix.seed[0].bkt != nullptr && ix.n_slabs == 1 && NL <= 512 && !gm_tune("GM_NO_BUCKETS");
  if (!all && !bkt && NL < 65536 && !gm_tune("GM_K1_V2") && !gm_tune("GM_K1_V3") && !gm_tune("GM_K1_V4") && !gm_tune("GM_NO_V5")) {
    // k_lookup_v5 (gm_lookup5.hip): wave-per-list streaming, candidates and the exact rule in LDS, K1b's prune rules fused when the caller allows.
    // It takes the read-strands with many list entries (gm_lookup5_launch declines the others: 50-colour reads, small genomes) -- round 2: 24 %
    // fewer VALU instructions than v4 + K1b, 2.09 M reads/s against 1.90 M on the 3 Gbp workload.
    const bool want_fuse = fuse && fuse->scap2 > 0;
    const uint32_t D = (uint32_t)std::max(want_fuse ? fuse->window_len : 0, read_len);
    const int e_max = want_fuse ? std::min(fuse->e_max, read_len) : -1;
    uint32_t *fbl = nullptr, *fbc = nullptr, *pll = nullptr, *plc = nullptr; int fbcap = 0;
    gm_lookup5_set_start_flags(g_k4_flags, g_k4_flag_cap, g_k4_epoch);
    bool fused = want_fuse;
    int r = gm_lookup5_launch(ix, d_reads, n_reads, read_len, read_words, max_n_kmers, NL, fused ? fuse->d_surv2 : d_surv, fused ? fuse->d_surv_cnt2 : d_surv_cnt,
                              fused ? fuse->scap2 : scap, d_surv_cnt, fused ? 1 : 0, D, e_max, d_heavy_list, d_heavy_cnt, heavy_cap, d_stats, stream, &fbl, &fbc, &fbcap,
                              d_surv, scap, d_surv_seg, &pll, &plc);
    if (r == 0 && fused) {   // the prune rules do not fit region-sized bins (very long reads): v5 without them, K1b afterwards
      fused = false;
      r = gm_lookup5_launch(ix, d_reads, n_reads, read_len, read_words, max_n_kmers, NL, d_surv, d_surv_cnt, scap, d_surv_cnt, 0, D, -1,
                            d_heavy_list, d_heavy_cnt, heavy_cap, d_stats, stream, &fbl, &fbc, &fbcap);
    }
    g_k4_flag_grid = gm_lookup5_start_flag_grid();
    gm_lookup5_set_start_flags(nullptr, 0, 0);
    if (r < 0) return r;
    if (r == 1) {
      g_k1_name = gm_lookup5_last_half() ? "k_lookup_v5_half" : gm_lookup5_last_rounds() > 1 ? "k_lookup_v5_rounds" : "k_lookup_v5";
      // read-strands whose candidates did not fit the LDS tiers (none on the benchmark genome): the slab-sweep kernel in list mode (blocks beyond the list's end
      // return at once), then K1b for those.  (k_lookup_v4 in list mode was tried for them: no faster, and its 134 KB workgroups wait longer for a CU.)
      hipLaunchKernelGGL(k_lookup<false>, dim3(std::min(fbcap, 1024)), dim3(K1_THREADS), lds, stream, ix, d_reads, n_reads, read_len, read_words,
                         max_n_kmers, NL, bm_words, d_surv, d_surv_cnt, scap, d_heavy_list, d_heavy_cnt, heavy_cap,
                         (const uint32_t*)nullptr, (const uint64_t*)nullptr, (const uint32_t*)fbl, (const uint32_t*)fbc, d_stats, 0, d_surv_seg);
      GM_HIP(hipGetLastError());
      if (fused) {
        const int rc = gm_launch_prune(n_reads, read_len, fuse->window_len, fuse->e_max, ix.n_slabs, ix.slab_bits, d_surv, d_surv_cnt, d_surv_seg, scap,
                                       fuse->d_surv2, fuse->d_surv_cnt2, fuse->scap2, d_heavy_list, d_heavy_cnt, heavy_cap, d_stats, stream, fbl, fbc, fbcap);
        if (rc) return rc;
        if (pll) {   // read-strands whose members v5 left in their raw rows (more kept than K2's tier under region-sized bins): K1b only
          const int rc2 = gm_launch_prune(n_reads, read_len, fuse->window_len, fuse->e_max, ix.n_slabs, ix.slab_bits, d_surv, d_surv_cnt, d_surv_seg, scap,
                                          fuse->d_surv2, fuse->d_surv_cnt2, fuse->scap2, d_heavy_list, d_heavy_cnt, heavy_cap, d_stats, stream, pll, plc, fbcap);
          if (rc2) return rc2;
        }
        if (fuse->fused) *fuse->fused = 1;
      }
      return GM_OK;
    }
  }
  if (bkt) {
    g_k1_name = "k_lookup_bkt";
    const size_t lds_b = (size_t)((((read_len + 3) / 4) + 3) & ~3) * 4 + (size_t)bm_words * 4;
    // the persistent form (probes one read-strand ahead) unless GM_BKT_V1 asks for one workgroup per read-strand; its grid: what fits a CU by waves and LDS, times the CUs
    const int threads_b = (NL + 63) & ~63;
    const size_t lds_p = lds_b + (size_t)((((read_len + 3) / 4) + 3) & ~3) * 4;
    int dev_b = 0, cus_b = 256; (void)hipGetDevice(&dev_b); if (hipDeviceGetAttribute(&cus_b, hipDeviceAttributeMultiprocessorCount, dev_b) != hipSuccess || cus_b < 1) cus_b = 256;
    { static GmLdsLimit lim_b, lim_p; size_t &cb = lim_b.cur(), &cp = lim_p.cur();      // (a small region size on a bucket-sized genome: the bitmap can pass 48 KB)
      if (lds_b > 48 * 1024 && lds_b > cb) { GM_HIP(hipFuncSetAttribute((const void*)k_lookup_bkt, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_b)); cb = lds_b; }
      if (lds_p > 48 * 1024 && lds_p <= 64 * 1024 && lds_p > cp) { GM_HIP(hipFuncSetAttribute((const void*)k_lookup_bkt_p, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_p)); cp = lds_p; } }
    int per_cu = 0;                                              // resident workgroups per CU (registers, waves, LDS): the persistent grid is exactly that
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)k_lookup_bkt_p, threads_b, lds_p) != hipSuccess || per_cu < 1) per_cu = 4;
    if (const char* e = gm_tune("GM_BKT_PER_CU")) { const int v = atoi(e); if (v >= 1 && v < per_cu) { if (getenv("GM_TIMELINE")) fprintf(stderr, "[bkt] %d workgroups a CU fit, %d taken (%d threads, %zu B of LDS)\n", per_cu, v, threads_b, lds_p); per_cu = v; } }
    if (gm_tune("GM_BKT_V1") || threads_b > 512 || lds_p > 64 * 1024)
    hipLaunchKernelGGL(k_lookup_bkt, dim3(n_reads * 2), dim3(threads_b), lds_b, stream, ix, d_reads, n_reads, read_len, read_words,
                       max_n_kmers, NL, bm_words, d_surv, d_surv_cnt, scap, d_heavy_list, d_heavy_cnt, heavy_cap, d_stats, d_surv_seg);
    else
    hipLaunchKernelGGL(k_lookup_bkt_p, dim3(std::min(n_reads * 2, cus_b * per_cu)), dim3(threads_b), lds_p, stream, ix, d_reads, n_reads, read_len, read_words,
                       max_n_kmers, NL, bm_words, d_surv, d_surv_cnt, scap, d_heavy_list, d_heavy_cnt, heavy_cap, d_stats, d_surv_seg);
  } else if (!all && ix.n_slabs > 1 && NL < 65536 && !gm_tune("GM_K1_V2") && !gm_tune("GM_K1_V3") && k4_launch(ix, d_reads, n_reads, read_len, read_words, max_n_kmers, NL, d_surv, d_surv_cnt, scap,
                                                                                                   d_heavy_list, d_heavy_cnt, heavy_cap, d_stats, stream, d_surv_seg, lds, bm_words)) {
    // k_lookup_v4 ran (hashed pre-count + exact count on the candidates); read-strands it could not hold were redone in list mode
    g_k1_name = "k_lookup_v4";
  } else if (!all && ix.list_cutoff < 65536u && !gm_tune("GM_K1_V2") &&
             (size_t)((((read_len + 3) / 4) + 3 * NL + ix.n_slabs * NL + (ix.n_slabs + 1) / 2 + (NL * (ix.n_slabs + 1) + 1) / 2 + 3) & ~3) * 4 + (size_t)bm_words * 4 <= 160 * 1024) {
    // (long reads on many slabs: the per-slab window maps outgrow the LDS and the lane-per-list kernel below takes over)
    g_k1_name = "k_lookup_v3";
    const size_t lds3 = (size_t)((((read_len + 3) / 4) + 3 * NL + ix.n_slabs * NL + (ix.n_slabs + 1) / 2 + (NL * (ix.n_slabs + 1) + 1) / 2 + 3) & ~3) * 4 + (size_t)bm_words * 4;
    static GmLdsLimit lim_configured3; size_t& configured3 = lim_configured3.cur();
    if (lds3 > 48 * 1024 && lds3 > configured3) { GM_HIP(hipFuncSetAttribute((const void*)k_lookup_v3, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds3)); configured3 = lds3; }
    int k1_threads = 768;
    if (const char* e = gm_tune("GM_K1_THREADS")) k1_threads = std::max(64, std::min(768, atoi(e) & ~63));
    hipLaunchKernelGGL(k_lookup_v3, dim3(n_reads * 2), dim3(k1_threads), lds3, stream, ix, d_reads, n_reads, read_len, read_words,
                       max_n_kmers, NL, bm_words, d_surv, d_surv_cnt, scap, d_heavy_list, d_heavy_cnt, heavy_cap, d_stats,
                       gm_tune("GM_K1_ABLATE") ? atoi(gm_tune("GM_K1_ABLATE")) : 0, d_surv_seg);
  } else {
    g_k1_name = "k_lookup";
    // one list per lane when the read-strand's lists fit a workgroup: every list slice is in flight at once
    int k1_threads = std::min(1024, (NL + 63) & ~63);
    if (const char* e = gm_tune("GM_K1_THREADS")) k1_threads = std::max(64, std::min(1024, atoi(e) & ~63));
    if (all) {
      static GmLdsLimit lim6; size_t& c6 = lim6.cur();
      if (lds > 48 * 1024 && lds > c6) { GM_HIP(hipFuncSetAttribute((const void*)k_lookup<false, 6>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); c6 = lds; }
      hipLaunchKernelGGL((k_lookup<false, 6>), dim3(n_reads * 2), dim3(k1_threads), lds, stream, ix, d_reads, n_reads, read_len, read_words,
                         max_n_kmers, NL, bm_words, d_surv, d_surv_cnt, scap, d_heavy_list, d_heavy_cnt, heavy_cap,
                         (const uint32_t*)nullptr, (const uint64_t*)nullptr, (const uint32_t*)nullptr, (const uint32_t*)nullptr, d_stats, 0, d_surv_seg);
    } else
    hipLaunchKernelGGL(k_lookup<false>, dim3(n_reads * 2), dim3(k1_threads), lds, stream, ix, d_reads, n_reads, read_len, read_words,
                       max_n_kmers, NL, bm_words, d_surv, d_surv_cnt, scap, d_heavy_list, d_heavy_cnt, heavy_cap,
                       (const uint32_t*)nullptr, (const uint64_t*)nullptr, (const uint32_t*)nullptr, (const uint32_t*)nullptr, d_stats,
                       gm_tune("GM_K1_ABLATE") ? atoi(gm_tune("GM_K1_ABLATE")) : 0, d_surv_seg);
  }
  GM_HIP(hipGetLastError());
  return GM_OK;
}

// heavy tier: re-run the listed read-strands, each into its exactly sized slice of d_out
int gm_launch_lookup_redo(const GmIndexDev& ix, const uint32_t* d_reads, int n_reads, int read_len, int read_words,
                          int n_heavy, const uint32_t* d_redo_list, const uint64_t* d_redo_off, uint64_t* d_out,
                          unsigned long long* d_stats, hipStream_t stream) {
  int max_n_kmers, NL, bm_words; size_t lds;
  k1_geometry(ix, read_len, &max_n_kmers, &NL, &bm_words, &lds);
  if (n_heavy == 0) return GM_OK;
  if (ix.no_region_counts) {
    static GmLdsLimit lim6r; size_t& c6 = lim6r.cur();
    if (lds > 48 * 1024 && lds > c6) { GM_HIP(hipFuncSetAttribute((const void*)k_lookup<false, 6>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); c6 = lds; }
    hipLaunchKernelGGL((k_lookup<false, 6>), dim3(n_heavy), dim3(K1_THREADS), lds, stream, ix, d_reads, n_reads, read_len, read_words,
                       max_n_kmers, NL, bm_words, d_out, (uint32_t*)nullptr, 0, (uint32_t*)nullptr, (uint32_t*)nullptr, 0,
                       d_redo_list, d_redo_off, (const uint32_t*)nullptr, (const uint32_t*)nullptr, d_stats, 0, (uint32_t*)nullptr);
  } else
  if (ix.mp.mode == 5) {
    int rc = k1_mp_lds(lds); if (rc) return rc;
    hipLaunchKernelGGL((k_lookup<false, 5>), dim3(n_heavy), dim3(K1_THREADS), lds, stream, ix, d_reads, n_reads, read_len, read_words,
                       max_n_kmers, NL, bm_words, d_out, (uint32_t*)nullptr, 0, (uint32_t*)nullptr, (uint32_t*)nullptr, 0,
                       d_redo_list, d_redo_off, (const uint32_t*)nullptr, (const uint32_t*)nullptr, d_stats, 0, (uint32_t*)nullptr);
  } else
  if (ix.mp.mode == 4) {
    int rc = k1_mp_lds(lds); if (rc) return rc;
    hipLaunchKernelGGL((k_lookup<false, 4>), dim3(n_heavy), dim3(K1_THREADS), lds, stream, ix, d_reads, n_reads, read_len, read_words,
                       max_n_kmers, NL, bm_words, d_out, (uint32_t*)nullptr, 0, (uint32_t*)nullptr, (uint32_t*)nullptr, 0,
                       d_redo_list, d_redo_off, (const uint32_t*)nullptr, (const uint32_t*)nullptr, d_stats, 0, (uint32_t*)nullptr);
  } else
  if (ix.mp.mode == 2) {
    int rc = k1_mp_lds(lds); if (rc) return rc;
    hipLaunchKernelGGL((k_lookup<false, 2>), dim3(n_heavy), dim3(K1_THREADS), lds, stream, ix, d_reads, n_reads, read_len, read_words,
                       max_n_kmers, NL, bm_words, d_out, (uint32_t*)nullptr, 0, (uint32_t*)nullptr, (uint32_t*)nullptr, 0,
                       d_redo_list, d_redo_off, (const uint32_t*)nullptr, (const uint32_t*)nullptr, d_stats, 0, (uint32_t*)nullptr);
  } else
  hipLaunchKernelGGL(k_lookup<false>, dim3(n_heavy), dim3(K1_THREADS), lds, stream, ix, d_reads, n_reads, read_len, read_words,
                     max_n_kmers, NL, bm_words, d_out, (uint32_t*)nullptr, 0, (uint32_t*)nullptr, (uint32_t*)nullptr, 0,
                     d_redo_list, d_redo_off, (const uint32_t*)nullptr, (const uint32_t*)nullptr, d_stats, 0, (uint32_t*)nullptr);
  GM_HIP(hipGetLastError());
  return GM_OK;
}
