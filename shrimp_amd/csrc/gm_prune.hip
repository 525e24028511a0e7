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
#include "gm_region_table.h"

__device__ void prune_one(uint32_t* sm, uint32_t* n_keep_p, uint32_t* too_many_p, int rs, const uint64_t* __restrict__ surv, const uint32_t* __restrict__ surv_cnt,
                          const uint32_t* __restrict__ surv_seg, int scap, uint64_t* __restrict__ surv2, uint32_t* __restrict__ surv_cnt2, int scap2, uint32_t D, int e_max,
                          int bin_bits, int hbits, int n_slabs, int slab_bits, uint32_t* __restrict__ heavy_list, uint32_t* __restrict__ heavy_cnt, int heavy_cap,
                          unsigned long long* __restrict__ stats);

__global__ void __launch_bounds__(1024)
k_prune(int n_rs, const uint64_t* __restrict__ surv, const uint32_t* __restrict__ surv_cnt, const uint32_t* __restrict__ surv_seg, int scap,
        uint64_t* __restrict__ surv2, uint32_t* __restrict__ surv_cnt2, int scap2, uint32_t D, int e_max, int bin_bits, int hbits,
        int n_slabs, int slab_bits,
        uint32_t* __restrict__ heavy_list, uint32_t* __restrict__ heavy_cnt, int heavy_cap, unsigned long long* __restrict__ stats,
        const uint32_t* __restrict__ rs_list, const uint32_t* __restrict__ rs_cnt) {   // list mode: block b prunes read-strand rs_list[b] (b < *rs_cnt)
  extern __shared__ __align__(16) uint32_t sm[];
  __shared__ uint32_t n_keep, too_many;
  // list mode: a bounded grid walks the list (a grid as long as the list's capacity -- hundreds of thousands of workgroups that exit at once -- kept
  // the persistent K1 grid of the next sub-batch and pass 1 of this one from sharing the CUs: the two-stream pipeline ran in stage order)
  if (rs_cnt) {
    const uint32_t cnt = *rs_cnt;
    for (uint32_t b = blockIdx.x; b < cnt; b += gridDim.x) { prune_one(sm, &n_keep, &too_many, (int)rs_list[b], surv, surv_cnt, surv_seg, scap, surv2, surv_cnt2, scap2, D, e_max, bin_bits, hbits,
                                                                       n_slabs, slab_bits, heavy_list, heavy_cnt, heavy_cap, stats); __syncthreads(); }
    return;
  }
  prune_one(sm, &n_keep, &too_many, (int)blockIdx.x, surv, surv_cnt, surv_seg, scap, surv2, surv_cnt2, scap2, D, e_max, bin_bits, hbits, n_slabs, slab_bits, heavy_list, heavy_cnt, heavy_cap, stats);
}

__device__ void prune_one(uint32_t* sm, uint32_t* n_keep_p, uint32_t* too_many_p, int rs, const uint64_t* __restrict__ surv, const uint32_t* __restrict__ surv_cnt,
                          const uint32_t* __restrict__ surv_seg, int scap, uint64_t* __restrict__ surv2, uint32_t* __restrict__ surv_cnt2, int scap2, uint32_t D, int e_max,
                          int bin_bits, int hbits, int n_slabs, int slab_bits, uint32_t* __restrict__ heavy_list, uint32_t* __restrict__ heavy_cnt, int heavy_cap,
                          unsigned long long* __restrict__ stats) {
  uint32_t& n_keep = *n_keep_p; uint32_t& too_many = *too_many_p;
  const int tid = threadIdx.x;
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


// ---------------------------------------------------------------------------------------------
// k_prune_v2: the same two rules with bins of 2^bb >= D + e_max positions (the 2 048-base regions by default) in the region table of
// gm_region_table.h.  What changes against k_prune:
//  * only single-shot LDS atomics (slot claim by one CAS, counts as flag bits, min / max by atomicMax): k_prune's compare-and-swap loop on the
//    packed (count, min, max) word serialises the ~250 survivors of a read that really maps, which all fall into one or two bins;
//  * survivors of one region come in twos by construction (that is why they survived), so the table holds about half as many keys as survivors:
//    4 096 slots x 12 B = 48 KB, three workgroups per CU instead of two;
//  * a survivor at least D + e_max away from both ends of its bin never needs the neighbour bins (no neighbour-bin survivor is within D of it or
//    of anything within e_max of it): 83 % of them at 2 048-base bins skip two look-ups of mostly absent keys.
// Bigger bins only make the rules more conservative where they are not exact (three or more survivors in a bin: kept; rule 2 sees more
// bystanders).  Read-strands with more survivors than the table takes go to k_prune (list mode).
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(512)
k_prune_v2(int n_rs, const uint64_t* __restrict__ surv, const uint32_t* __restrict__ surv_cnt, int scap, uint64_t* __restrict__ surv2, uint32_t* __restrict__ surv_cnt2,
           int scap2, uint32_t D, int e_max, int bb, int hbits, uint32_t n_max, uint32_t* __restrict__ heavy_list, uint32_t* __restrict__ heavy_cnt, int heavy_cap,
           unsigned long long* __restrict__ stats, uint32_t* __restrict__ ov_list, uint32_t* __restrict__ ov_cnt, int ov_cap) {
  extern __shared__ __align__(16) uint32_t sm[];
  __shared__ uint32_t n_keep, full;
  const int rs = blockIdx.x, tid = threadIdx.x, nthr = blockDim.x, lane = tid & 63;
  const uint32_t n = surv_cnt[rs];
  if (n > (uint32_t)scap) { if (tid == 0) surv_cnt2[rs] = 0xFFFFFFFFu; return; }      // already on the heavy list (K1)
  if (n == 0) { if (tid == 0) surv_cnt2[rs] = 0; return; }
  if (n > n_max) {                                                                   // more than the table takes: k_prune does this one
    if (tid == 0) { const uint32_t f = atomicAdd(ov_cnt, 1u); if (f < (uint32_t)ov_cap) ov_list[f] = (uint32_t)rs; else GS_ADD(stats, GS_OVERFLOW_SURV, 1ull); }
    return;
  }
  const uint32_t H = 1u << hbits, hmask = H - 1u; const int hshift = 32 - hbits;
  uint32_t* htag = sm; uint32_t* hmin = sm + H; uint32_t* hmax = hmin + H;
  { uint4* t4 = (uint4*)sm; for (uint32_t w = tid; w < (3u * H) >> 2; w += nthr) t4[w] = make_uint4(0, 0, 0, 0); }
  if (tid == 0) { n_keep = 0; full = 0; }
  __syncthreads();
  const uint64_t* in = surv + (size_t)rs * scap;
  uint64_t* out = surv2 + (size_t)rs * scap2;
  const uint32_t bmask = (1u << bb) - 1u;
  for (uint32_t i = tid; i < n; i += nthr) {
    const uint32_t x = (uint32_t)(in[i] >> 32), r = x >> bb, off = x & bmask;
    const uint32_t h = k5_insert(htag, hmask, hshift, r, true);
    if (h == 0xFFFFFFFFu) full = 1u; else { atomicMax(&hmin[h], 0x10000u - off); atomicMax(&hmax[h], off + 1u); }
  }
  __syncthreads();
  const uint32_t edge = D + (uint32_t)max(e_max, 0);
  if (!full)
    for (uint32_t i0 = 0; i0 < n; i0 += nthr) {
      const uint32_t i = i0 + tid;
      bool keep = false; uint64_t ent = 0;
      if (i < n) {
        ent = in[i];
        const uint32_t x = (uint32_t)(ent >> 32), r = x >> bb, off = x & bmask;
        uint32_t town, tlf = 0, trt = 0, hlf = 0xFFFFFFFFu, hrt = 0xFFFFFFFFu;
        const uint32_t hown = k5_find(htag, hmask, hshift, r, town);
        if (off < edge && r > 0) hlf = k5_find(htag, hmask, hshift, r - 1u, tlf);
        if (off + edge >= (1u << bb)) hrt = k5_find(htag, hmask, hshift, r + 1u, trt);
        const uint32_t co = (town & K5_FE) ? 3u : ((town & K5_FD) ? 2u : 1u);
        const uint32_t omin = 0x10000u - hmin[hown], omax = hmax[hown] - 1u, rbase = r << bb;
        const bool cl = (tlf & K5_FC) != 0u, cr = (trt & K5_FC) != 0u;
        // (1) isolation: two entries in the bin: the other one is max - min away; three or more: keep
        keep = co >= 3u || (co == 2u && omax - omin <= D);
        uint32_t lmin = 0, lmax = 0, rmin = 0, rmax = 0;
        if (cl) { lmin = 0x10000u - hmin[hlf]; lmax = hmax[hlf] - 1u; }
        if (cr) { rmin = 0x10000u - hmin[hrt]; rmax = hmax[hrt] - 1u; }
        if (!keep && cl) keep = x - (rbase - (1u << bb) + lmax) <= D;
        if (!keep && cr) keep = (rbase + (1u << bb) + rmin) - x <= D;
        // (2) tight cluster: everything in the three bins (which cover x -+ (D + e_max)) spans at most e_max positions
        if (keep && e_max >= 0) {
          uint32_t gmin = rbase + omin, gmax = rbase + omax;
          if (cl) gmin = rbase - (1u << bb) + lmin;
          if (cr) gmax = rbase + (1u << bb) + rmax;
          if (gmax - gmin <= (uint32_t)e_max) keep = false;
        }
      }
      const unsigned long long bk = __ballot(keep);
      if (bk) {
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(&n_keep, (uint32_t)__popcll(bk));
        base = __builtin_amdgcn_readfirstlane(base);
        if (keep) { const uint32_t sl = base + (uint32_t)__popcll(bk & ((1ull << lane) - 1ull)); if (sl < (uint32_t)scap2) out[sl] = ent; }
      }
    }
  __syncthreads();
  if (tid == 0) {
    const uint32_t k = n_keep;
    if (full || k > (uint32_t)scap2) {
      // table full (cannot happen with n <= n_max < slots), or more kept than K2's LDS tier takes: region-sized bins keep a few more than k_prune's
      // 256-base bins do (three survivors in one region are all kept), so k_prune gets the last word before the heavy tier -- a read-strand on the
      // heavy list stalls the two-stream pipeline for its whole sub-batch
      const uint32_t f = atomicAdd(ov_cnt, 1u); if (f < (uint32_t)ov_cap) ov_list[f] = (uint32_t)rs; else GS_ADD(stats, GS_OVERFLOW_SURV, 1ull);
    } else {
      surv_cnt2[rs] = k;
      GS_ADD(stats, GS_PRUNED, (unsigned long long)(n - k));
    }
  }
}

int gm_launch_prune(int n_reads, int read_len, int window_len, int e_max, int n_slabs, int slab_bits, const uint64_t* d_surv, const uint32_t* d_surv_cnt,
                    const uint32_t* d_surv_seg, int scap,
                    uint64_t* d_surv2, uint32_t* d_surv_cnt2, int scap2, uint32_t* d_heavy_list, uint32_t* d_heavy_cnt, int heavy_cap,
                    unsigned long long* d_stats, hipStream_t stream, const uint32_t* d_rs_list, const uint32_t* d_rs_cnt, int rs_cap) {
  if (n_reads == 0) return GM_OK;
  const uint32_t D = (uint32_t)std::max(window_len, read_len);
  if (e_max > read_len) e_max = read_len;
  if (!d_rs_cnt && !gm_tune("GM_PRUNE_V1") && D + (uint32_t)std::max(0, e_max) <= 65535u) {
    // k_prune_v2 for every read-strand; the few with more survivors than its table takes are listed and done by k_prune below (list mode)
    int dev = 0; GM_HIP(hipGetDevice(&dev));
    static uint32_t* ov[16] = {nullptr}; static int ov_cap[16] = {0};
    if (dev >= 0 && dev < 16) {
      const int need = std::max(4096, 2 * n_reads);
      if (need > ov_cap[dev]) { if (ov[dev]) { GM_HIP(hipDeviceSynchronize()); (void)hipFree(ov[dev]); } GM_HIP(hipMalloc(&ov[dev], (size_t)(need + 4) * 4)); ov_cap[dev] = need; }
      uint32_t* ovc = ov[dev] + ov_cap[dev];
      GM_HIP(hipMemsetAsync(ovc, 0, 4, stream));
      int bb = 11; while ((1u << bb) < D + (uint32_t)std::max(0, e_max)) bb++;
      int hb = 12; if (const char* e = gm_tune("GM_PRUNE_HBITS")) hb = std::max(6, std::min(13, atoi(e)));
      const uint32_t n_max = (3u << hb) / 4u;
      const size_t lds2 = (size_t)12 << hb;
      static GmLdsLimit lim_configured2; size_t& configured2 = lim_configured2.cur();
      if (lds2 > 48 * 1024 && lds2 > configured2) { GM_HIP(hipFuncSetAttribute((const void*)k_prune_v2, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2)); configured2 = lds2; }
      hipLaunchKernelGGL(k_prune_v2, dim3(n_reads * 2), dim3(512), lds2, stream, n_reads * 2, d_surv, d_surv_cnt, scap, d_surv2, d_surv_cnt2, scap2, D, e_max, bb, hb, n_max,
                         d_heavy_list, d_heavy_cnt, heavy_cap, d_stats, ov[dev], ovc, ov_cap[dev]);
      GM_HIP(hipGetLastError());
      return gm_launch_prune(n_reads, read_len, window_len, e_max, n_slabs, slab_bits, d_surv, d_surv_cnt, d_surv_seg, scap, d_surv2, d_surv_cnt2, scap2,
                             d_heavy_list, d_heavy_cnt, heavy_cap, d_stats, stream, ov[dev], ovc, ov_cap[dev]);
    }
  }
  int bin_bits = 1; while ((1u << bin_bits) < D + (uint32_t)std::max(0, e_max)) bin_bits++;
  if (bin_bits > 12) { gm_set_error("prune: D = %u does not fit the 12-bit bin offsets", D); return GM_E_ARG; }
  // table for one slab's segment: twice the expected share of the survivor capacity (segments beyond half the table go to the heavy tier)
  const int segs = d_surv_seg ? std::max(1, n_slabs) : 1;
  int hbits = 8; while ((1 << hbits) < 2 * std::max(128, (segs > 1 ? (2 * scap) / segs : scap))) hbits++;
  const size_t lds = (size_t)8 << hbits;
  static GmLdsLimit lim_configured; size_t& configured = lim_configured.cur();
  if (lds > 48 * 1024 && lds > configured) { GM_HIP(hipFuncSetAttribute((const void*)k_prune, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); configured = lds; }
  // latency-bound (hash probes): as many lanes per read-strand as a segment has work for
  const int pthreads = gm_tune("GM_PRUNE_THREADS") ? atoi(gm_tune("GM_PRUNE_THREADS")) : std::min(1024, std::max(128, (1 << hbits) / 8));
  hipLaunchKernelGGL(k_prune, dim3(d_rs_cnt ? std::min(rs_cap, 1024) : n_reads * 2), dim3(pthreads), lds, stream, n_reads * 2, d_surv, d_surv_cnt, d_surv_seg, scap, d_surv2, d_surv_cnt2, scap2,
                     D, e_max, bin_bits, hbits, segs, slab_bits, d_heavy_list, d_heavy_cnt, heavy_cap, d_stats, d_rs_list, d_rs_cnt);
  GM_HIP(hipGetLastError());
  return GM_OK;
}
