// gm_prune.hip -- K1b: drop survivors that cannot take part in any anchor chain or window (gfx950 only).
//
// After the reference's region filter (K1) a read-strand on a 3 Gbp genome keeps ~2 600 list entries,
// almost all of them chance pairs that merely share a 2 048 bp region.  The reference turns every one
// into an anchor, merges them through its heap and then finds that nearly all are inert:
//   * the collapse cache only joins, or is only disturbed between, anchors of one diagonal whose
//     start positions differ by less than read_len            (ref: gmapper/mapping.c:957-971);
//   * the window look-back of anchor i only visits anchors j with x_i - x_j <= window_len, and a lone
//     weight-1 anchor never opens a window in match mode 2    (ref: gmapper/mapping.c:1084-1100,1102-1151).
// Hence a survivor with no other survivor within D = max(window_len, read_len) positions changes
// neither the anchor list seen by any other anchor nor the window list; removing it is exact.  (The
// look-back stops at the first anchor below its bound; an isolated anchor between i and that bound
// would itself be within window_len of i.  An isolated anchor that overwrites a cache slot could only
// block a join between two anchors less than read_len apart that it lies between.)  K2 then sorts
// 3-4x fewer keys on a 3 Gbp genome.  Only the `anchors` statistic changes.
//
//
// Second rule (tight clusters).  In match mode 2 a window needs max_score >= thr = (int)abs_or_pct(window_gen_threshold,
// min(read_len, w_len) * match) (ref: mapping.c:1153-1155) where max_score is at most short_len * match and
// short_len = min(dx, dy) + length_i <= (x extent of the anchors involved) + max seed span (a collapsed anchor never
// extends past its last survivor + span, ref: common/anchors.c:98-119; gap penalties only lower the score).  A group of
// survivors that is more than D away from every other survivor and spans at most e_max = ceil(thr / match) - max_span - 1
// positions can therefore open no window and touches nothing outside itself: chance partial matches (one seed hit
// plus its shifted / other-seed echoes on the same diagonal) are exactly that.  Dropping any subset of such a group
// is exact as well (the bound holds for the rest), so the test is made per survivor on what its three bins show.
//
// One workgroup per read-strand: survivors are binned by position >> bin_bits (bin size >= D) in an LDS
// hash table holding, per occupied bin, a saturating count and the min/max offset; a survivor is kept
// if its bin holds two entries or the adjacent bin's nearest entry lies within D.  Conservative where
// it is not exact (bin size > D), never the other way.
#include "gm_common.h"
#include <algorithm>
#include "gm_internal.h"

__global__ void __launch_bounds__(1024)
k_prune(int n_rs, const uint64_t* __restrict__ surv, const uint32_t* __restrict__ surv_cnt, const uint32_t* __restrict__ surv_seg, int scap,
        uint64_t* __restrict__ surv2, uint32_t* __restrict__ surv_cnt2, int scap2, uint32_t D, int e_max, int bin_bits, int hbits,
        int n_slabs, int slab_bits,
        uint32_t* __restrict__ heavy_list, uint32_t* __restrict__ heavy_cnt, int heavy_cap, unsigned long long* __restrict__ stats,
        const uint32_t* __restrict__ rs_list, const uint32_t* __restrict__ rs_cnt) {   // list mode: block b prunes read-strand rs_list[b] (b < *rs_cnt)
  extern __shared__ __align__(16) uint32_t sm[];
  __shared__ uint32_t n_keep, too_many;
  if (rs_cnt && blockIdx.x >= *rs_cnt) return;
  const int rs = rs_cnt ? (int)rs_list[blockIdx.x] : (int)blockIdx.x, tid = threadIdx.x;
  const uint32_t n = surv_cnt[rs];
  if (n > (uint32_t)scap) { if (tid == 0) surv_cnt2[rs] = 0xFFFFFFFFu; return; }      // already on the heavy list (K1)
  if (n == 0) { if (tid == 0) surv_cnt2[rs] = 0; return; }
  const uint32_t H = 1u << hbits;
  uint32_t* keys = sm; uint32_t* info = sm + H;        // info = cnt(2, saturating) << 24 | min_off << 12 | max_off
  if (tid == 0) { n_keep = 0; too_many = 0; }
  const uint64_t* in = surv + (size_t)rs * scap;
  uint64_t* out = surv2 + (size_t)rs * scap2;
  const uint32_t omask = (1u << bin_bits) - 1u;
  auto slot_of = [&](uint32_t key) { return (key * 2654435761u) >> (32 - hbits); };
  auto find = [&](uint32_t key) -> uint32_t {          // info of the bin, or 0 when the bin is empty
    uint32_t h = slot_of(key);
    for (;;) {
      const uint32_t k = keys[h];
      if (k == key) return info[h];
      if (k == 0u) return 0u;
      h = (h + 1u) & (H - 1u);
    }
  };
  // K1 emits the survivors slab by slab (surv_seg[s + 1] = how many after slab s), so each slab's segment is pruned on its own
  // with a small table; survivors closer than D + e_max to a slab border are kept as they are (their neighbours may sit in the
  // other segment) -- a superset of what the rules keep, hence still exact.
  const uint32_t guard = D + (uint32_t)max(0, e_max);
  // All survivors fit the table at half load (the usual case): one round over the whole read-strand, no border guards.
  const bool whole = n <= H / 2;
  const int rounds = whole ? 1 : n_slabs;
  for (int s = 0; s < rounds; s++) {
    const uint32_t b = (whole || s == 0 || !surv_seg) ? 0u : min(n, surv_seg[(size_t)rs * (n_slabs + 1) + s]);
    const uint32_t e = (whole || !surv_seg || s == n_slabs - 1) ? n : min(n, surv_seg[(size_t)rs * (n_slabs + 1) + s + 1]);
    if (e <= b) continue;
    if (e - b > H / 2) { if (tid == 0) too_many = 1; continue; }       // segment larger than the table allows: heavy tier
    { uint4* k4 = (uint4*)keys; uint4* i4 = (uint4*)info;     // H >= 256: whole 16-byte stores
      for (uint32_t i = tid; i < H / 4; i += blockDim.x) { k4[i] = make_uint4(0, 0, 0, 0); i4[i] = make_uint4(0x00FFF000u, 0x00FFF000u, 0x00FFF000u, 0x00FFF000u); } }
    __syncthreads();
    for (uint32_t i = b + tid; i < e; i += blockDim.x) {
      const uint32_t x = (uint32_t)(in[i] >> 32), key = (x >> bin_bits) + 1u, o = x & omask;
      uint32_t h = slot_of(key);
      for (;;) {
        const uint32_t k = atomicCAS(&keys[h], 0u, key);
        if (k == 0u || k == key) break;
        h = (h + 1u) & (H - 1u);
      }
      uint32_t old = info[h];
      for (;;) {
        const uint32_t c = min(3u, (old >> 24) + 1u), mn = min((old >> 12) & 0xFFFu, o), mx = max(old & 0xFFFu, o);
        const uint32_t prev = atomicCAS(&info[h], old, (c << 24) | (mn << 12) | mx);
        if (prev == old) break;
        old = prev;
      }
    }
    __syncthreads();
    const uint64_t B = (uint64_t)s << slab_bits, E = B + (1ull << slab_bits);
    for (uint32_t i = b + tid; i < e; i += blockDim.x) {
      const uint64_t ent = in[i];
      const uint32_t x = (uint32_t)(ent >> 32), bin = x >> bin_bits;
      bool keep;
      if (!whole && n_slabs > 1 && ((uint64_t)x < B + guard || (uint64_t)x + guard >= E)) keep = true;
      else {
        const uint32_t own = find(bin + 1u), lf = bin > 0 ? find(bin) : 0u, rt = find(bin + 2u);
        // (1) isolation: two entries in the bin: the other one is max - min away; three or more: keep
        keep = (own >> 24) >= 3u || ((own >> 24) == 2u && (own & 0xFFFu) - ((own >> 12) & 0xFFFu) <= D);
        if (!keep && lf) keep = x - (((bin - 1u) << bin_bits) + (lf & 0xFFFu)) <= D;
        if (!keep && rt) keep = (((bin + 1u) << bin_bits) + ((rt >> 12) & 0xFFFu)) - x <= D;
        // (2) tight cluster: everything in the three bins (which cover x -+ (D + e_max)) spans at most e_max positions
        if (keep && e_max >= 0) {
          uint32_t gmin = (bin << bin_bits) + ((own >> 12) & 0xFFFu), gmax = (bin << bin_bits) + (own & 0xFFFu);
          if (lf) gmin = ((bin - 1u) << bin_bits) + ((lf >> 12) & 0xFFFu);
          if (rt) gmax = ((bin + 1u) << bin_bits) + (rt & 0xFFFu);
          if (gmax - gmin <= (uint32_t)e_max) keep = false;
        }
      }
      if (keep) { const uint32_t sl = atomicAdd(&n_keep, 1u); if (sl < (uint32_t)scap2) out[sl] = ent; }
    }
    __syncthreads();
  }
  __syncthreads();
  if (tid == 0) {
    const uint32_t k = n_keep;
    if (k > (uint32_t)scap2 || too_many) {  // still too many for the LDS tier of K2: heavy tier (re-emits all survivors)
      surv_cnt2[rs] = 0xFFFFFFFFu;
      const uint32_t hs = atomicAdd(heavy_cnt, 1u);
      if (hs < (uint32_t)heavy_cap) heavy_list[hs] = (uint32_t)rs; else GS_ADD(stats, GS_OVERFLOW_SURV, 1ull);
    } else surv_cnt2[rs] = k;
    GS_ADD(stats, GS_PRUNED, (unsigned long long)(n - min(k, n)));
  }
}

int gm_launch_prune(int n_reads, int read_len, int window_len, int e_max, int n_slabs, int slab_bits, const uint64_t* d_surv, const uint32_t* d_surv_cnt,
                    const uint32_t* d_surv_seg, int scap,
                    uint64_t* d_surv2, uint32_t* d_surv_cnt2, int scap2, uint32_t* d_heavy_list, uint32_t* d_heavy_cnt, int heavy_cap,
                    unsigned long long* d_stats, hipStream_t stream, const uint32_t* d_rs_list, const uint32_t* d_rs_cnt, int rs_cap) {
  if (n_reads == 0) return GM_OK;
  const uint32_t D = (uint32_t)std::max(window_len, read_len);
  if (e_max > read_len) e_max = read_len;
  int bin_bits = 1; while ((1u << bin_bits) < D + (uint32_t)std::max(0, e_max)) bin_bits++;
  if (bin_bits > 12) { gm_set_error("prune: D = %u does not fit the 12-bit bin offsets", D); return GM_E_ARG; }
  // table for one slab's segment: twice the expected share of the survivor capacity (segments beyond half the table go to the heavy tier)
  const int segs = d_surv_seg ? std::max(1, n_slabs) : 1;
  int hbits = 8; while ((1 << hbits) < 2 * std::max(128, (segs > 1 ? (2 * scap) / segs : scap))) hbits++;
  const size_t lds = (size_t)8 << hbits;
  static size_t configured = 0;
  if (lds > 48 * 1024 && lds > configured) { GM_HIP(hipFuncSetAttribute((const void*)k_prune, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); configured = lds; }
  // latency-bound (hash probes): as many lanes per read-strand as a segment has work for
  const int pthreads = gm_tune("GM_PRUNE_THREADS") ? atoi(gm_tune("GM_PRUNE_THREADS")) : std::min(1024, std::max(128, (1 << hbits) / 8));
  hipLaunchKernelGGL(k_prune, dim3(d_rs_cnt ? rs_cap : n_reads * 2), dim3(pthreads), lds, stream, n_reads * 2, d_surv, d_surv_cnt, d_surv_seg, scap, d_surv2, d_surv_cnt2, scap2,
                     D, e_max, bin_bits, hbits, segs, slab_bits, d_heavy_list, d_heavy_cnt, heavy_cap, d_stats, d_rs_list, d_rs_cnt);
  GM_HIP(hipGetLastError());
  return GM_OK;
}
