// gm_pair.hip -- paired-mode glue kernels between K2 (candidate windows) and K3/K4 (gfx950 only).
//   readpair_pair_up_hits        ref: gmapper/mapping.c:266-325
//   readpair_get_vector_hits     ref: gmapper/mapping.c:1877-1932   (ext-heap CMP :1871-1873, common/heap.h:226-327)
//   read_reverse                 ref: gmapper/gmapper.c:174-185
// These are thin per-pair loops over a handful of windows; they are latency-, not bandwidth-bound, and
// exist so that the window lists never leave HBM between K2 and pass 2.
#include "gm_internal.h"

// In-place reverse complement of every packed read: the stored orientation of a read_reverse'd mate is
// read[0] = revcomp(input) (ref: gmapper.c:174-185 swaps read[0]/read[1] and flips input_strand).
__global__ void __launch_bounds__(256) k_revcomp_reads(uint32_t* __restrict__ reads, int n_reads, int read_len, int read_words) {
  const int rd = blockIdx.x * blockDim.x + threadIdx.x;
  if (rd >= n_reads) return;
  uint32_t* rw = reads + (size_t)rd * read_words;
  const uint64_t cm = 0xFBCDE56879A00123ull;   // complement_base as nibbles (ref: util.h:125-151)
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

int gm_launch_revcomp_reads(uint32_t* d_reads, int n_reads, int read_len, int read_words, hipStream_t stream) {
  if (n_reads == 0) return GM_OK;
  hipLaunchKernelGGL(k_revcomp_reads, dim3((n_reads + 255) / 256), dim3(256), 0, stream, d_reads, n_reads, read_len, read_words);
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
