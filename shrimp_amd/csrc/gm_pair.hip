// gm_pair.hip -- paired-mode glue kernels between K2 (candidate windows) and K3/K4 (gfx950 only).
//   readpair_pair_up_hits        ref: gmapper/mapping.c:266-325
//   readpair_get_vector_hits     ref: gmapper/mapping.c:1877-1932   (ext-heap CMP :1871-1873, common/heap.h:226-327)
//   read_reverse                 ref: gmapper/gmapper.c:174-185
// These are thin per-pair loops over a handful of windows; they are latency-, not bandwidth-bound, and
// exist so that the window lists never leave HBM between K2 and pass 2.
#include "gm_internal.h"

// In-place reverse complement of every packed read: the stored orientation of a read_reverse'd mate is
// read[0] = revcomp(input) (ref: gmapper.c:174-185 swaps read[0]/read[1] and flips input_strand).
__global__ void __launch_bounds__(256) k_revcomp_reads(uint32_t* __restrict__ reads, int n_reads, int read_len, int read_words, const uint8_t* __restrict__ read_rna) {
  const int rd = blockIdx.x * blockDim.x + threadIdx.x;
  if (rd >= n_reads) return;
  uint32_t* rw = reads + (size_t)rd * read_words;
  const uint64_t cm = gm_cmpl_tab(read_rna && read_rna[rd]);   // complement_base as nibbles (ref: util.h:125-151); an RNA read's complement of A is U
  auto get = [&](int i) { return (rw[i >> 3] >> ((i & 7) * 4)) & 0xfu; };
  auto put = [&](int i, uint32_t c) { const int sh = (i & 7) * 4; rw[i >> 3] = (rw[i >> 3] & ~(0xfu << sh)) | (c << sh); };
  for (int i = 0, j = read_len - 1; i <= j; i++, j--) {
    const uint32_t a = get(i), b = get(j);
    const uint32_t ca = (uint32_t)(cm >> (a * 4)) & 0xf, cb = (uint32_t)(cm >> (b * 4)) & 0xf;
    put(i, cb);
    if (i != j) put(j, ca);
  }
}

struct PairDelta { int dmin[2], dmax[2]; };

// One thread per (pair, st1): two-pointer sweep of mate 1's strand-st1 windows against mate 2's
// strand-(1-st1) windows, both in (contig, g_off) order.  pair_min/pair_max are positions in the
// mate's sorted list; the arrays were preset to -1.
__global__ void __launch_bounds__(256)
k_pair_up(int n_pairs, const GmHit* __restrict__ hits1, const uint16_t* __restrict__ perm1, const uint32_t* __restrict__ cnt1, int hcap1,
          const GmHit* __restrict__ hits2, const uint16_t* __restrict__ perm2, const uint32_t* __restrict__ cnt2, int hcap2,
          int32_t* __restrict__ pmin1, int32_t* __restrict__ pmax1, int32_t* __restrict__ pmin2, int32_t* __restrict__ pmax2, PairDelta dl) {
  const int id = blockIdx.x * blockDim.x + threadIdx.x;
  if (id >= n_pairs * 2) return;
  const int pr = id >> 1, st1 = id & 1, st2 = 1 - st1;
  const size_t rs1 = (size_t)pr * 2 + st1, rs2 = (size_t)pr * 2 + st2;
  const int n1 = (int)min(cnt1[rs1], (uint32_t)hcap1), n2 = (int)min(cnt2[rs2], (uint32_t)hcap2);
  if (n1 == 0 || n2 == 0) return;
  const GmHit* H1 = hits1 + rs1 * hcap1; const uint16_t* P1 = perm1 + rs1 * hcap1;
  const GmHit* H2 = hits2 + rs2 * hcap2; const uint16_t* P2 = perm2 + rs2 * hcap2;
  int32_t* mn1 = pmin1 + rs1 * hcap1; int32_t* mx1 = pmax1 + rs1 * hcap1;
  int32_t* mn2 = pmin2 + rs2 * hcap2; int32_t* mx2 = pmax2 + rs2 * hcap2;
  const long long dmin = dl.dmin[st1], dmax = dl.dmax[st1];
  int j = 0;
  for (int i = 0; i < n1; i++) {
    const GmHit& h1 = H1[P1[i]];
    const int cn = h1.cn; const long long g1 = (long long)h1.g_off;
    while (j < n2) {
      const GmHit& h2 = H2[P2[j]];
      if ((int)h2.cn < cn || ((int)h2.cn == cn && (long long)h2.g_off < g1 + dmin)) j++; else break;
    }
    int k = j;
    while (k < n2) {
      const GmHit& h2 = H2[P2[k]];
      if ((int)h2.cn == cn && (long long)h2.g_off <= g1 + dmax) k++; else break;
    }
    if (j == k) continue;
    mn1[i] = j; mx1[i] = k - 1;
    for (int l = j; l < k; l++) { if (mn2[l] < 0) mn2[l] = i; mx2[l] = i; }
  }
}

// One thread per pair: top-K window pairs by summed vector score with the reference's ext-heap
// (array order is kept: pass 2 walks it in that order), then the unique windows of each mate
// become that mate's pass-2 selection list.
//   sel{1,2}[pr][a]  = (st << 16) | hit index        sidx{1,2}[pr][a] = sort_idx of that window
//   pairs[pr][k]     = (a << 8) | b   (indices into the two selection lists)
__global__ void __launch_bounds__(64)
k_pair_select(GmScoreDev sc, int n_pairs, int len1, int len2,
              const GmHit* __restrict__ hits1, const uint16_t* __restrict__ perm1, const uint32_t* __restrict__ cnt1, int hcap1,
              const int32_t* __restrict__ pmin1, const int32_t* __restrict__ pmax1,
              const GmHit* __restrict__ hits2, const uint16_t* __restrict__ perm2, const uint32_t* __restrict__ cnt2, int hcap2,
              int32_t* __restrict__ sel1, int32_t* __restrict__ sidx1, uint32_t* __restrict__ selcnt1,
              int32_t* __restrict__ sel2, int32_t* __restrict__ sidx2, uint32_t* __restrict__ selcnt2,
              uint32_t* __restrict__ pairs, uint32_t* __restrict__ pair_cnt) {
  const int pr = blockIdx.x * blockDim.x + threadIdx.x;
  if (pr >= n_pairs) return;
  int key[GM_SEL_MAX]; uint32_t id[GM_SEL_MAX];     // id = st1 << 30 | i << 15 | j  (sorted positions)
  int load = 0;
  const int K = min(sc.num_tmp_outputs, GM_SEL_MAX);
  const bool absthr = sc.vect_thr_frac < 0;
  for (int st1 = 0; st1 < 2; st1++) {
    const size_t rs1 = (size_t)pr * 2 + st1, rs2 = (size_t)pr * 2 + (1 - st1);
    const int n1 = (int)min(cnt1[rs1], (uint32_t)hcap1);
    const GmHit* H1 = hits1 + rs1 * hcap1; const uint16_t* P1 = perm1 + rs1 * hcap1;
    const GmHit* H2 = hits2 + rs2 * hcap2; const uint16_t* P2 = perm2 + rs2 * hcap2;
    for (int i = 0; i < n1; i++) {
      const int lo = pmin1[rs1 * hcap1 + i];
      if (lo < 0) continue;
      const int hi = pmax1[rs1 * hcap1 + i];
      const GmHit& h1 = H1[P1[i]];
      const int smax1 = (len1 < (int)h1.w_len ? len1 : (int)h1.w_len) * sc.match;
      for (int j = lo; j <= hi; j++) {
        const GmHit& h2 = H2[P2[j]];
        const int smax2 = (len2 < (int)h2.w_len ? len2 : (int)h2.w_len) * sc.match;
        const int score = h1.score_vector + h2.score_vector, score_max = smax1 + smax2;
        const int pct = (1000 * 100 * score) / score_max;
        const int k = absthr ? score : pct;
        const int thr = sc.vect_thr_frac < 0 ? sc.vect_abs : (int)((double)score_max * sc.vect_thr_frac);
        if (score >= thr && (load < K || k > key[0])) {                                   // ref: mapping.c:1911-1918
          const uint32_t me = ((uint32_t)st1 << 30) | ((uint32_t)i << 15) | (uint32_t)j;
          if (load < K) {
            key[load] = k; id[load] = me; load++;
            int node = load, parent = node / 2;
            while (node > 1 && key[node - 1] < key[parent - 1]) {
              int tk = key[parent - 1]; key[parent - 1] = key[node - 1]; key[node - 1] = tk;
              uint32_t ti = id[parent - 1]; id[parent - 1] = id[node - 1]; id[node - 1] = ti;
              node = parent; parent = node / 2;
            }
          } else {
            key[0] = k; id[0] = me;
            int node = 1;
            for (;;) {
              int left = node * 2, right = left + 1, mn = node;
              if (left <= load && key[left - 1] < key[node - 1]) mn = left;
              if (right <= load && key[right - 1] < key[mn - 1]) mn = right;
              if (mn == node) break;
              int tk = key[mn - 1]; key[mn - 1] = key[node - 1]; key[node - 1] = tk;
              uint32_t ti = id[mn - 1]; id[mn - 1] = id[node - 1]; id[node - 1] = ti;
              node = mn;
            }
          }
        }
      }
    }
  }
  // unique windows per mate, in order of first appearance
  const int n1_st0 = (int)min(cnt1[(size_t)pr * 2], (uint32_t)hcap1), n2_st0 = (int)min(cnt2[(size_t)pr * 2], (uint32_t)hcap2);
  int32_t* S1 = sel1 + (size_t)pr * GM_SEL_MAX; int32_t* X1 = sidx1 + (size_t)pr * GM_SEL_MAX;
  int32_t* S2 = sel2 + (size_t)pr * GM_SEL_MAX; int32_t* X2 = sidx2 + (size_t)pr * GM_SEL_MAX;
  int c1 = 0, c2 = 0;
  for (int k = 0; k < load; k++) {
    const int st1 = (int)(id[k] >> 30), i = (int)((id[k] >> 15) & 0x7FFF), j = (int)(id[k] & 0x7FFF), st2 = 1 - st1;
    const int v1 = (st1 << 16) | (int)perm1[((size_t)pr * 2 + st1) * hcap1 + i];
    const int v2 = (st2 << 16) | (int)perm2[((size_t)pr * 2 + st2) * hcap2 + j];
    int a = 0; while (a < c1 && S1[a] != v1) a++;
    if (a == c1) { S1[c1] = v1; X1[c1] = (st1 == 0 ? i : n1_st0 + i); c1++; }
    int b = 0; while (b < c2 && S2[b] != v2) b++;
    if (b == c2) { S2[c2] = v2; X2[c2] = (st2 == 0 ? j : n2_st0 + j); c2++; }
    pairs[(size_t)pr * GM_SEL_MAX + k] = ((uint32_t)a << 8) | (uint32_t)b;
  }
  selcnt1[pr] = (uint32_t)c1; selcnt2[pr] = (uint32_t)c2; pair_cnt[pr] = (uint32_t)load;
}

__global__ void __launch_bounds__(256) k_mark_saved(uint8_t* __restrict__ saved, const uint32_t* __restrict__ list, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) saved[list[i]] = 1;
}

// re->is_rna of every packed letter-space read: uracil and no thymine among its letters (ref: common/fasta.c:528-542).  Taken from the reads as they were handed in --
// before a read_reverse'd mate is turned (its complement of U is A).
__global__ void __launch_bounds__(256) k_read_rna_flags(const uint32_t* __restrict__ reads, int n_reads, int read_len, int read_words, uint8_t* __restrict__ flags) {
  const int rd = blockIdx.x * blockDim.x + threadIdx.x;
  if (rd >= n_reads) return;
  const uint32_t* rw = reads + (size_t)rd * read_words;
  bool u = false, t = false;
  for (int w = 0; w < read_words; w++) {
    const uint32_t x = rw[w]; const int nn = min(8, read_len - 8 * w);
    uint32_t valid = nn >= 8 ? 0x88888888u : ((0x88888888u >> (4 * (8 - nn))));      // the top bit of every nibble that holds a letter
    const uint32_t xu = x ^ 0x44444444u, xt = x ^ 0x33333333u;                          // a zero nibble = a U / a T
    u |= (((xu - 0x11111111u) & ~xu) & valid) != 0u;
    t |= (((xt - 0x11111111u) & ~xt) & valid) != 0u;
  }
  flags[rd] = (u && !t) ? 1 : 0;
}
int gm_launch_read_rna_flags(const uint32_t* d_reads, int n_reads, int read_len, int read_words, uint8_t* d_flags, hipStream_t stream) {
  if (n_reads <= 0) return GM_OK;
  hipLaunchKernelGGL(k_read_rna_flags, dim3((n_reads + 255) / 256), dim3(256), 0, stream, d_reads, n_reads, read_len, read_words, d_flags);
  GM_HIP(hipGetLastError());
  return GM_OK;
}
int gm_launch_revcomp_reads(uint32_t* d_reads, int n_reads, int read_len, int read_words, hipStream_t stream, const uint8_t* d_read_rna) {
  if (n_reads == 0) return GM_OK;
  hipLaunchKernelGGL(k_revcomp_reads, dim3((n_reads + 255) / 256), dim3(256), 0, stream, d_reads, n_reads, read_len, read_words, d_read_rna);
  GM_HIP(hipGetLastError());
  return GM_OK;
}

int gm_launch_pair_up(int n_pairs, const GmHit* hits1, const uint16_t* perm1, const uint32_t* cnt1, int hcap1,
                      const GmHit* hits2, const uint16_t* perm2, const uint32_t* cnt2, int hcap2,
                      int32_t* pmin1, int32_t* pmax1, int32_t* pmin2, int32_t* pmax2, const int* delta_min, const int* delta_max, hipStream_t stream) {
  if (n_pairs == 0) return GM_OK;
  PairDelta dl; dl.dmin[0] = delta_min[0]; dl.dmin[1] = delta_min[1]; dl.dmax[0] = delta_max[0]; dl.dmax[1] = delta_max[1];
  GM_HIP(hipMemsetAsync(pmin1, 0xff, (size_t)n_pairs * 2 * hcap1 * 4, stream)); GM_HIP(hipMemsetAsync(pmax1, 0xff, (size_t)n_pairs * 2 * hcap1 * 4, stream));
  GM_HIP(hipMemsetAsync(pmin2, 0xff, (size_t)n_pairs * 2 * hcap2 * 4, stream)); GM_HIP(hipMemsetAsync(pmax2, 0xff, (size_t)n_pairs * 2 * hcap2 * 4, stream));
  hipLaunchKernelGGL(k_pair_up, dim3((n_pairs * 2 + 255) / 256), dim3(256), 0, stream, n_pairs, hits1, perm1, cnt1, hcap1, hits2, perm2, cnt2, hcap2,
                     pmin1, pmax1, pmin2, pmax2, dl);
  GM_HIP(hipGetLastError());
  return GM_OK;
}

int gm_launch_pair_select(const GmScoreDev& sc, int n_pairs, int len1, int len2,
                          const GmHit* hits1, const uint16_t* perm1, const uint32_t* cnt1, int hcap1, const int32_t* pmin1, const int32_t* pmax1,
                          const GmHit* hits2, const uint16_t* perm2, const uint32_t* cnt2, int hcap2,
                          int32_t* sel1, int32_t* sidx1, uint32_t* selcnt1, int32_t* sel2, int32_t* sidx2, uint32_t* selcnt2,
                          uint32_t* pairs, uint32_t* pair_cnt, hipStream_t stream) {
  if (n_pairs == 0) return GM_OK;
  hipLaunchKernelGGL(k_pair_select, dim3((n_pairs + 63) / 64), dim3(64), 0, stream, sc, n_pairs, len1, len2, hits1, perm1, cnt1, hcap1, pmin1, pmax1,
                     hits2, perm2, cnt2, hcap2, sel1, sidx1, selcnt1, sel2, sidx2, selcnt2, pairs, pair_cnt);
  GM_HIP(hipGetLastError());
  return GM_OK;
}

int gm_launch_mark_saved(uint8_t* d_saved, const uint32_t* d_list, int n, hipStream_t stream) {
  if (n == 0) return GM_OK;
  hipLaunchKernelGGL(k_mark_saved, dim3((n + 255) / 256), dim3(256), 0, stream, d_saved, d_list, n);
  GM_HIP(hipGetLastError());
  return GM_OK;
}

// ---------------------------------------------------------------------------------------------
// Mate-pair region counts (ref: mapping.c:545-608 read_get_mp_region_counts, :733-742 the test in advance_index_in_genomemap with
// use_mp_region_counts == 1: match mode 4 without half-paired, gmapper.c:2657-2662).  A list entry of mate A, strand st, whose region X
// (or X - 1 from the overlap strip) was marked twice stays only if mate B, strand 1 - st, marked some region in [X + dmin, X + dmax] twice.
// K1 has already applied the "marked twice" part, and the regions a read-strand marked twice are exactly the regions its SURVIVORS mark twice
// (every entry that marks such a region is a survivor), so both maps are rebuilt from the survivor lists: one workgroup per (pair, st) puts the
// regions of A[st] and of B[1 - st] into two LDS region tables (gm_region_table.h, flags A / B = once / twice), then drops from both lists the
// entries that fail the test -- the relation is symmetric (A[st] looks at B[1 - st] and the other way round) -- and compacts them in place.
// A read-strand beyond the LDS tier (count > scap: heavy tier) or a table that runs full leaves both lists as they are: a superset of the
// reference's anchors (the filter only removes entries that cannot pair up); such items are counted in GS_MP_UNFILTERED.
// ---------------------------------------------------------------------------------------------
#include "gm_region_table.h"
#define MPF_THREADS 1024
#define MPF_HBITS 14
#define MPF_PER_THREAD 16          // survivors per thread held in registers during the compaction: scap <= 16 384

__global__ void __launch_bounds__(MPF_THREADS)
k_mp_filter(int n_pairs, int rb, uint32_t ovl, uint64_t* __restrict__ surv1, uint32_t* __restrict__ cnt1, int scap1,
            uint64_t* __restrict__ surv2, uint32_t* __restrict__ cnt2, int scap2, MpDelta dl, uint32_t* __restrict__ seg1, uint32_t* __restrict__ seg2, int n_slabs,
            unsigned long long* __restrict__ unfiltered) {
  extern __shared__ __align__(16) uint32_t mpf_lds[];
  uint32_t* tagA = mpf_lds; uint32_t* tagB = tagA + (1u << MPF_HBITS);
  __shared__ uint32_t sh_fail, sh_keep[2], sh_wave[MPF_THREADS / 64];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const uint32_t hmask = (1u << MPF_HBITS) - 1u; const int hshift = 32 - MPF_HBITS;
  const uint32_t rmask = (1u << rb) - 1u;
  for (int w = blockIdx.x; w < 2 * n_pairs; w += gridDim.x) {
    const int p = w >> 1, st = w & 1;
    const int rsA = 2 * p + st, rsB = 2 * p + (1 - st);
    const uint32_t nA = cnt1[rsA], nB = cnt2[rsB];
    __syncthreads();                                           // (the tables and the counters of the previous item are done with)
    if (nA > (uint32_t)scap1 || nB > (uint32_t)scap2 || nA > MPF_THREADS * MPF_PER_THREAD || nB > MPF_THREADS * MPF_PER_THREAD) {
      if (tid == 0 && (nA || nB)) GS_ADD(unfiltered, GS_MP_UNFILTERED, 1ull);
      continue;
    }
    if (nA == 0 && nB == 0) continue;
    uint64_t* sA = surv1 + (size_t)rsA * scap1; uint64_t* sB = surv2 + (size_t)rsB * scap2;
    for (uint32_t i = tid; i < (2u << MPF_HBITS); i += MPF_THREADS) mpf_lds[i] = 0;
    if (tid == 0) { sh_fail = 0; sh_keep[0] = sh_keep[1] = 0; }
    __syncthreads();
    // the two region maps
    for (int side = 0; side < 2; side++) {
      const uint64_t* s = side ? sB : sA; const uint32_t n = side ? nB : nA; uint32_t* tag = side ? tagB : tagA;
      for (uint32_t i = tid; i < n; i += MPF_THREADS) {
        const uint32_t pos = (uint32_t)(s[i] >> 32), r = pos >> rb;
        bool ok = k5_insert(tag, hmask, hshift, r, false) != 0xFFFFFFFFu;
        if ((pos & rmask) < ovl && r > 0) ok = (k5_insert(tag, hmask, hshift, r - 1u, false) != 0xFFFFFFFFu) && ok;   // ref: mapping.c:521-533
        if (!ok) sh_fail = 1u;
      }
    }
    __syncthreads();
    if (sh_fail) { if (tid == 0) GS_ADD(unfiltered, GS_MP_UNFILTERED, 1ull); continue; }
    // the test, then the compaction of each list in place (its survivors wait in registers across the barrier)
    for (int side = 0; side < 2; side++) {
      uint64_t* s = side ? sB : sA; const uint32_t n = side ? nB : nA;
      const uint32_t* own = side ? tagB : tagA; const uint32_t* mate = side ? tagA : tagB;
      const int dmin = side ? dl.bmin[1 - st] : dl.amin[st], dmax = side ? dl.bmax[1 - st] : dl.amax[st];
      auto has2 = [&](const uint32_t* tag, long long r) -> bool {
        if (r < 0 || r > 0xFFFFFFFFll >> rb) return false;
        uint32_t t; k5_find(tag, hmask, hshift, (uint32_t)r, t);
        return (t & K5_FB) != 0u;
      };
      auto mp_ok = [&](long long X) -> bool {                  // count_main >= 2 && count_mp >= 2 (ref: mapping.c:733-742, use_mp_region_counts == 1)
        if (!has2(own, X)) return false;
        for (long long k = X + dmin; k <= X + dmax; k++) if (has2(mate, k)) return true;
        return false;
      };
      uint64_t keep_v[MPF_PER_THREAD]; uint32_t nk = 0;
#pragma unroll
      for (int u = 0; u < MPF_PER_THREAD; u++) {
        const uint32_t i = (uint32_t)tid * MPF_PER_THREAD + (uint32_t)u;          // a contiguous piece per thread: the compaction keeps the order
        keep_v[u] = ~0ull;
        if (i < n) {
          const uint64_t key = s[i]; const uint32_t pos = (uint32_t)(key >> 32), r = pos >> rb;
          if (mp_ok((long long)r) || ((pos & rmask) < ovl && r > 0 && mp_ok((long long)r - 1))) { keep_v[u] = key; nk++; }
        }
      }
      // exclusive prefix of nk over the workgroup
      uint32_t incl = nk;
      for (int d = 1; d < 64; d <<= 1) { const uint32_t o = __shfl_up(incl, d); if (lane >= d) incl += o; }
      if (lane == 63) sh_wave[wv] = incl;
      __syncthreads();                                         // (also: every thread has read its part of the list)
      uint32_t base = 0, total = 0;
      for (int k = 0; k < MPF_THREADS / 64; k++) { const uint32_t c = sh_wave[k]; if (k < wv) base += c; total += c; }
      uint32_t o = base + incl - nk;
#pragma unroll
      for (int u = 0; u < MPF_PER_THREAD; u++) if (keep_v[u] != ~0ull) s[o++] = keep_v[u];
      if (tid == 0) { if (side) cnt2[rsB] = total; else cnt1[rsA] = total; }
      // K1's per-slab segment ends (K1b prunes slab by slab when a read-strand does not fit its table) no longer hold: one segment with everything --
      // K1b then keeps what lies beyond the first slab as it is (its border guard), a superset of what its rules keep, still exact
      { uint32_t* sg = side ? seg2 : seg1; const int rs = side ? rsB : rsA;
        if (sg && tid <= n_slabs) sg[(size_t)rs * (n_slabs + 1) + tid] = tid == 0 ? 0u : total; }
      __syncthreads();                                         // sh_wave is reused by the other side
    }
  }
}

MpDelta gm_mp_region_deltas(int region_bits, const int* dmin1, const int* dmax1, const int* dmin2, const int* dmax2) {
  MpDelta dl;
  const int R = 1 << region_bits;
  auto rmin = [&](int v) { return v >= 0 ? v / R : -1 - (-v - 1) / R; };          // ref: mapping.c:2422-2430
  auto rmax = [&](int v) { return v > 0 ? 1 + (v - 1) / R : -(-v / R); };
  for (int st = 0; st < 2; st++) { dl.amin[st] = rmin(dmin1[st]); dl.amax[st] = rmax(dmax1[st]); dl.bmin[st] = rmin(dmin2[st]); dl.bmax[st] = rmax(dmax2[st]); }
  return dl;
}

int gm_launch_mp_filter(int n_pairs, int region_bits, int region_overlap, uint64_t* d_surv1, uint32_t* d_cnt1, int scap1, uint64_t* d_surv2, uint32_t* d_cnt2, int scap2,
                        const int* dmin1, const int* dmax1, const int* dmin2, const int* dmax2, uint32_t* d_seg1, uint32_t* d_seg2, int n_slabs,
                        unsigned long long* d_unfiltered, hipStream_t stream) {
  if (n_pairs == 0) return GM_OK;
  const MpDelta dl = gm_mp_region_deltas(region_bits, dmin1, dmax1, dmin2, dmax2);
  const size_t lds = (size_t)(2u << MPF_HBITS) * 4;
  static GmLdsLimit lim_mp; size_t& configured = lim_mp.cur();
  if (lds > configured) { GM_HIP(hipFuncSetAttribute((const void*)k_mp_filter, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); configured = lds; }
  const int grid = std::min(2 * n_pairs, 1024);
  hipLaunchKernelGGL(k_mp_filter, dim3(grid), dim3(MPF_THREADS), lds, stream, n_pairs, region_bits, (uint32_t)region_overlap, d_surv1, d_cnt1, scap1, d_surv2, d_cnt2, scap2,
                     dl, d_seg1, d_seg2, n_slabs, d_unfiltered);
  GM_HIP(hipGetLastError());
  return GM_OK;
}
