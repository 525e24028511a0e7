// gm_anchors.hip -- K2: survivors -> ordered, collapsed anchors -> candidate windows.
// One wave per read-strand.  Replaces
//   read_get_anchor_list_per_strand  ref: gmapper/mapping.c:861-1006  (k-way heap merge + colinear collapse)
//   read_get_hit_list_per_strand     ref: gmapper/mapping.c:1025-1229 (window generation + insertion sort)
//
// Order.  The reference merges the lists with a binary min-heap keyed by position only
// (ref: common/heap.h:44-139): equal positions pop in a history-dependent order.  What that order can
// influence is narrow (DESIGN.md "tie order"):
//   * the collapse is order-free: survivors are collapsed per diagonal class (x + L - y) % L, and two
//     survivors of one class at one position have the same y, hence merge whatever their order;
//   * anchors are created in pop order, so only the relative order of anchors with EQUAL x is open;
//     it matters only if (a) the best partner of an anchor in the look-back scan is, or ties in score
//     with, an equal-x sibling, or (b) two windows with equal (contig, g_off) come from equal-x anchors.
// Fast path: canonical order (x, y, seed), data-parallel collapse, and detection of (a)/(b).
// Exact path (taken when detected, and always in the heavy tier): lane 0 replays the reference's heap
// (same insert order, same strict-< sift rules) and the sequential collapse.  Both paths are exact.
//
// Two tiers.  LDS tier (BIG = false): survivors (<= scap) sorted by a bitonic network in LDS.
// Heavy tier (BIG = true): the few read-strands with more survivors (low-complexity reads, repeats);
// their keys were re-emitted by K1 into exactly sized global segments and sorted by one segmented
// radix sort; the exact path then runs on global arrays with agent-scope fences.
#include <cstring>
#include <algorithm>
#include "gm_common.h"
#include "gm_internal.h"
#include <rocprim/device/device_segmented_radix_sort.hpp>

template <bool BIG> struct K2Idx { typedef uint16_t type; static constexpr uint32_t none = 0xFFFFu; };
template <> struct K2Idx<true> { typedef uint32_t type; static constexpr uint32_t none = 0xFFFFFFFFu; };

template <bool BIG>
struct K2Ws {            // per-wave workspace
  typedef typename K2Idx<BIG>::type idx_t;
  uint64_t* key;         // [cap]  sort keys pos<<32 | y<<16 | seed; later anchors: x<<32 | len<<16 | weight
  uint32_t* aux;         // [cap]  y | cn<<16   (bit 15 = dead marker during the parallel collapse)
  idx_t* nxt;            // [cap]  exact path: next survivor of the same list   } LDS tier: the two u16 arrays are
  idx_t* ord;            // [cap]  exact path: pop order                        } contiguous = one u32[cap] scratch
  idx_t* first;          // [NL]   exact path: cursor per list
  uint32_t* hk;          // [NL]   exact path: heap keys / temp
  uint16_t* hr;          // [NL]   exact path: heap payload (list id)
  int16_t*  cache;       // [read_len] anchor_cache (ref: mapping.c:871,909-910)
};

// LDS tier: a plain barrier.  Heavy tier: the arrays live in global memory and are written by one
// lane and read by the others, so the barrier also releases/acquires at agent scope (L1 invalidate).
#ifdef K2_STAMPS
__device__ unsigned long long k2_stamps[8];
#define K2_STAMP(i) do { if (lane == 0) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); atomicAdd(&k2_stamps[i], t_ - t_prev); t_prev = t_; } } while (0)
extern "C" int gm_debug_k2_stamps(unsigned long long* out) {
  unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(k2_stamps), sizeof z) != hipSuccess) return GM_E_NODEVICE;
  if (hipMemcpyToSymbol(HIP_SYMBOL(k2_stamps), z, sizeof z) != hipSuccess) return GM_E_NODEVICE;
  return GM_OK;
}
#else
#define K2_STAMP(i) do { } while (0)
#endif
template <bool BIG> __device__ __forceinline__ void k2_sync() { if (BIG) __threadfence(); __syncthreads(); }

template <bool BIG, typename T> __device__ void k2_bitonic(T* key, int npad, int lane) {
  for (int k = 2; k <= npad; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int t = lane; t < (npad >> 1); t += GM_WAVE) {
        const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
        const int l = i | j;
        const bool up = ((i & k) == 0);
        const T a = key[i], b = key[l];
        if ((a > b) == up) { key[i] = b; key[l] = a; }
      }
      k2_sync<BIG>();
    }
}

// The same network on up to 1024 keys held in registers (R = 1, 2, 4, 8, 16 per lane, key i = lane * R + j): partners less than R apart sit in the same lane,
// the others come by a lane exchange -- one round trip per stage instead of an LDS read, a write and a barrier.  Equal keys are identical values
// (a survivor key is unique, pads are all ~0), so the result is the sorted array whatever the network does with them.
template <typename T> __device__ __forceinline__ T k2_xor_lane(T v, int dl) { return (T)__shfl_xor(v, dl); }
template <> __device__ __forceinline__ uint64_t k2_xor_lane<uint64_t>(uint64_t v, int dl) {
  const uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)v, dl), hi = (uint32_t)__shfl_xor((int)(uint32_t)(v >> 32), dl);
  return ((uint64_t)hi << 32) | lo;
}
template <typename T, int R> __device__ __forceinline__ void k2_bitonic_regs(T* key, int lane) {
  T v[R];
#pragma unroll
  for (int j = 0; j < R; j++) v[j] = key[lane * R + j];
#pragma unroll
  for (int k = 2; k <= 64 * R; k <<= 1) {
#pragma unroll
    for (int d = k >> 1; d > 0; d >>= 1) {
      if (d < R) {
#pragma unroll
        for (int j = 0; j < R; j++) if ((j & d) == 0) {
          const bool up = (((lane * R + j) & k) == 0);
          const T a = v[j], b = v[j | d];
          const bool sw = (a > b) == up;
          v[j] = sw ? b : a; v[j | d] = sw ? a : b;
        }
      } else {
        const int dl = d / R;
        const bool lower = (lane & dl) == 0;
#pragma unroll
        for (int j = 0; j < R; j++) {
          const bool up = (((lane * R + j) & k) == 0);
          const T o = k2_xor_lane<T>(v[j], dl);
          const bool less = v[j] < o;
          v[j] = ((up == lower) == less) ? v[j] : o;         // the lower index keeps the smaller key where the run ascends
        }
      }
    }
  }
#pragma unroll
  for (int j = 0; j < R; j++) key[lane * R + j] = v[j];
}
// keys in LDS, written and fenced by the caller; sorted and fenced on return
// RMAX: keys per lane the register networks go up to (16: 1 024 keys, 114 registers; 32: 2 048 keys, 248 registers -- a kernel of its own, k_anchors<false, 32>, taken only where
// the LDS tier holds 2 048 survivors: 150-base reads on a 3 Gbp genome keep ~800 per read-strand, a fifth of the read-strands more than 1 024)
template <bool BIG, typename T, int RMAX = 16> __device__ void k2_sort(T* key, int npad, int lane) {
  if (!BIG && npad <= 64 * RMAX) {                              // (the read-strand that carries the true hit keeps ~300 survivors at 100 bp on a 3 Gbp genome, ~700 at 150 bp)
    if (npad == 64) k2_bitonic_regs<T, 1>(key, lane); else if (npad == 128) k2_bitonic_regs<T, 2>(key, lane);
    else if (npad == 256) k2_bitonic_regs<T, 4>(key, lane); else if (npad == 512) k2_bitonic_regs<T, 8>(key, lane);
    else if (npad == 1024 || RMAX < 32) k2_bitonic_regs<T, 16>(key, lane);
    else k2_bitonic_regs<T, RMAX>(key, lane);
    __syncthreads();
  } else k2_bitonic<BIG, T>(key, npad, lane);
}

// anchor_join of two anchors (ref: common/anchors.c:9-52) in window-relative coordinates
struct K2Box { long long x, y; int length, width; };
__device__ __forceinline__ K2Box k2_join2(long long x0, long long y0, int l0, int w0, long long x1, long long y1, int l1, int w1) {
  long long nw0 = x0 + y0, sw0 = x0 - y0, ne0 = sw0 + 2 * (w0 - 1), se0 = nw0 + 2 * (l0 - 1);
  long long nw1 = x1 + y1, sw1 = x1 - y1, ne1 = sw1 + 2 * (w1 - 1), se1 = nw1 + 2 * (l1 - 1);
  long long nw_min = min(nw0, nw1), sw_min = min(sw0, sw1), ne_max = max(ne0, ne1), se_max = max(se0, se1);
  K2Box r;
  if ((nw_min + sw_min) % 2 != 0) nw_min--;
  r.x = (nw_min + sw_min) / 2;
  r.y = nw_min - r.x;
  if ((ne_max - sw_min) % 2 != 0) ne_max++;
  r.width = (int)((ne_max - sw_min) / 2 + 1);
  if ((se_max - nw_min) % 2 != 0) se_max++;
  r.length = (int)((se_max - nw_min) / 2 + 1);
  return r;
}

__device__ __forceinline__ int k2_threshold(double frac, int absval, int base) {
  // (int)abs_or_pct(thr, base), ref: common/util.h:53 -- frac = thr/100.0 computed on the host
  return frac < 0 ? absval : (int)((double)base * frac);
}

// ---- exact pop order: replay of heap_uu over the survivors (lane 0) -----------------------------
template <bool BIG>
__device__ void k2_exact_order(K2Ws<BIG>& ws, int n, int NL, int max_n_kmers) {
  typedef typename K2Idx<BIG>::type idx_t;
  const idx_t NONE = (idx_t)K2Idx<BIG>::none;
  for (int o = 0; o < NL; o++) ws.first[o] = NONE;
  for (int t = n - 1; t >= 0; t--) {
    const uint64_t k = ws.key[t];
    const int off = (int)(k & 0xFFFF) * max_n_kmers + (int)((k >> 16) & 0xFFFF);
    ws.nxt[t] = ws.first[off]; ws.first[off] = (idx_t)t;
  }
  // heap_uu (ref: common/heap.h:44-139): 1-based nodes over hk/hr[0..load)
  int load = 0;
  auto pos_of = [&](int t) -> uint32_t { return (uint32_t)(ws.key[t] >> 32); };
  auto down = [&](int node) {
    for (;;) {
      int left = node * 2, right = left + 1, mn = node;
      if (left <= load && ws.hk[left - 1] < ws.hk[node - 1]) mn = left;
      if (right <= load && ws.hk[right - 1] < ws.hk[mn - 1]) mn = right;
      if (mn == node) break;
      uint32_t tk = ws.hk[mn - 1]; ws.hk[mn - 1] = ws.hk[node - 1]; ws.hk[node - 1] = tk;
      uint16_t tr = ws.hr[mn - 1]; ws.hr[mn - 1] = ws.hr[node - 1]; ws.hr[node - 1] = tr;
      node = mn;
    }
  };
  for (int o = 0; o < NL; o++) {          // initial inserts in (seed, read position) order, ref: mapping.c:913-935
    if (ws.first[o] == NONE) continue;
    ws.hk[load] = pos_of((int)ws.first[o]); ws.hr[load] = (uint16_t)o; load++;
    int node = load, parent = node / 2;
    while (node > 1 && ws.hk[node - 1] < ws.hk[parent - 1]) {
      uint32_t tk = ws.hk[parent - 1]; ws.hk[parent - 1] = ws.hk[node - 1]; ws.hk[node - 1] = tk;
      uint16_t tr = ws.hr[parent - 1]; ws.hr[parent - 1] = ws.hr[node - 1]; ws.hr[node - 1] = tr;
      node = parent; parent = node / 2;
    }
  }
  int m = 0;
  while (load > 0) {                      // ref: mapping.c:937-989
    const int o = ws.hr[0];
    const int t = (int)ws.first[o];
    ws.ord[m++] = (idx_t)t;
    const idx_t nx = ws.nxt[t];
    if (nx != NONE) { ws.first[o] = nx; ws.hk[0] = pos_of((int)nx); ws.hr[0] = (uint16_t)o; down(1); }
    else { load--; if (load > 0) { ws.hk[0] = ws.hk[load]; ws.hr[0] = ws.hr[load]; down(1); } }
  }
  // apply the pop order: it only permutes entries inside groups of equal position, so only the
  // low words (y, seed) move; hk is free now and serves as the per-group temporary.
  int g0 = 0;
  while (g0 < n) {
    int g1 = g0 + 1;
    while (g1 < n && (ws.key[g1] >> 32) == (ws.key[g0] >> 32)) g1++;
    if (g1 - g0 > 1) {
      for (int g = g0; g < g1; g++) ws.hk[g - g0] = (uint32_t)ws.key[ws.ord[g]];
      for (int g = g0; g < g1; g++) ws.key[g] = (ws.key[g] & 0xFFFFFFFF00000000ull) | ws.hk[g - g0];
    }
    g0 = g1;
  }
}

// ---- sequential collapse in pop order (lane 0), ref: mapping.c:937-971 ---------------------------
template <bool BIG>
__device__ int k2_collapse_seq(const GmIndexDev& ix, K2Ws<BIG>& ws, int n, int read_len) {
  int na = 0;
  for (int t = 0; t < n; t++) {
    const uint64_t k = ws.key[t];
    const uint32_t x = (uint32_t)(k >> 32);
    const uint32_t au = ws.aux[t];
    const int y = (int)((k >> 16) & 0xFFFF), cn = (int)(au >> 16);     // y from the key: the pop order permuted the keys
    const int len = ix.seed[(int)(k & 0xFFFF)].span;
    const int diag = (int)(((long long)x + read_len - y) % read_len);
    const int j = ws.cache[diag];
    bool joined = false;
    if (j >= 0) {
      const uint64_t aj = ws.key[j];
      const uint32_t auj = ws.aux[j];
      const uint32_t xj = (uint32_t)(aj >> 32);
      if ((int)(auj >> 16) == cn && ((long long)xj - (long long)(auj & 0x7FFF) == (long long)x - y)) {
        // anchor_uw_join(dest = A[j], src), ref: common/anchors.c:98-119 (src.x >= dest.x here)
        uint32_t lj = (uint32_t)(aj >> 16) & 0xFFFF, wj = (uint32_t)aj & 0xFFFF;
        if ((long long)x + len > (long long)xj + lj) lj = (uint32_t)(x - xj + len);
        wj = min(wj + 1u, 0xFFFFu);
        ws.key[j] = ((uint64_t)xj << 32) | ((uint64_t)lj << 16) | wj;
        joined = true;
      }
    }
    if (!joined) {
      ws.cache[diag] = (int16_t)na;
      ws.key[na] = ((uint64_t)x << 32) | ((uint64_t)len << 16) | 1u;
      ws.aux[na] = (uint32_t)y | ((uint32_t)cn << 16);
      na++;
    }
  }
  return na;
}

// ---- data-parallel collapse (LDS tier).  Entries in canonical order; ck = u32 scratch [npad]. ------
// Within a diagonal class the survivors are visited in position order; an entry opens a new anchor iff
// its (true diagonal, contig) differs from its class predecessor's (all members of a run are colinear
// with the run's first entry, so comparing with the predecessor == comparing with the cached anchor).
template <int RMAX>
__device__ int k2_collapse_par(const GmIndexDev& ix, K2Ws<false>& ws, uint32_t* ck, int n, int npad, int read_len, int lane) {
  for (int t = lane; t < npad; t += GM_WAVE) {
    uint32_t c = 0xFFFFFFFFu;
    if (t < n) {
      const uint32_t x = (uint32_t)(ws.key[t] >> 32); const uint32_t y = ws.aux[t] & 0x7FFFu;
      const uint32_t cls = ((x % (uint32_t)read_len) + (uint32_t)read_len - y) % (uint32_t)read_len;
      c = (cls << 16) | (uint32_t)t;
    }
    ck[t] = c;
  }
  k2_sync<false>();
  k2_sort<false, uint32_t, RMAX>(ck, npad, lane);
  // Runs of one (class, contig, diagonal) are contiguous now.  A run's head takes the extent of all its members and their number -- a read that really maps
  // puts ~250 colinear k-mer hits into ONE run, and walking it from its head (one lane, two dependent LDS reads per member) was two thirds of this kernel's time.
  // So: every member finds its head by a running maximum over the head positions (wave scan, chunk by chunk) and adds itself with two LDS atomics: extent by
  // atomicMax into the head's ck word (length << 16 | t, the class bits are dead by then), count by atomicAdd into the low word of the head's key (which only
  // the head itself had read).  Maximum and count do not depend on the order, so the result is the serial loop's.
  int carry_head = -1; uint32_t carry_c = 0xFFFFFFFFu;
  for (int c0 = 0; c0 < n; c0 += GM_WAVE) {
    const int r = c0 + lane; const bool valid = r < n;
    uint32_t c = 0xFFFFFFFFu; int t = 0; long long x = 0; int y = 0, cn = 0, span = 0; bool head = false;
    if (valid) {
      c = ck[r]; t = (int)(c & 0xFFFF);
      const uint64_t k = ws.key[t]; const uint32_t au = ws.aux[t];
      x = (long long)(k >> 32); y = (int)(au & 0x7FFF); cn = (int)(au >> 16); span = ix.seed[(int)(k & 0xFFFF)].span;
    }
    uint32_t cp = (uint32_t)__shfl_up((int)c, 1); if (lane == 0) cp = carry_c;       // the class word of the entry before (its ck slot may already hold a head's extent)
    if (valid) {
      head = true;
      if (r > 0 && (cp >> 16) == (c >> 16)) {
        const int tp = (int)(cp & 0xFFFF);
        const long long xp = (long long)(ws.key[tp] >> 32); const uint32_t aup = ws.aux[tp];
        head = !((int)(aup >> 16) == cn && (xp - (long long)(aup & 0x7FFF)) == (x - y));
      }
    }
    carry_c = (uint32_t)__shfl((int)c, GM_WAVE - 1);
    int hidx = (valid && head) ? r : -1;
    for (int d = 1; d < GM_WAVE; d <<= 1) { const int o = __shfl_up(hidx, d); if (lane >= d) hidx = max(hidx, o); }
    hidx = max(hidx, carry_head);
    carry_head = __shfl(hidx, GM_WAVE - 1);
    k2_sync<false>();                                            // every lane has read the words the heads now take over
    if (valid && head) { ck[r] = ((uint32_t)span << 16) | (uint32_t)t; ((uint32_t*)&ws.key[t])[0] = 1u; }
    k2_sync<false>();
    if (valid && !head) {
      const int th = (int)(ck[hidx] & 0xFFFF);
      const long long xh = (long long)(ws.key[th] >> 32);
      atomicMax(&ck[hidx], ((uint32_t)(x + span - xh) << 16) | (uint32_t)th);
      atomicAdd(&((uint32_t*)&ws.key[th])[0], 1u);
      atomicOr(&ws.aux[t], 0x8000u);
    }
  }
  k2_sync<false>();
  for (int r = lane; r < n; r += GM_WAVE) {                       // heads: x | extent | members (min(w, 0xFFFF), ref: anchors.h uw_join)
    const uint32_t c = ck[r]; const int t = (int)(c & 0xFFFF);
    if (ws.aux[t] & 0x8000u) continue;
    const uint64_t k = ws.key[t];
    ws.key[t] = (k & 0xFFFFFFFF00000000ull) | ((uint64_t)(c >> 16) << 16) | (uint64_t)min((uint32_t)k, 0xFFFFu);
  }
  k2_sync<false>();
  // stable compaction of the heads, in canonical (= creation) order
  int na = 0;
  for (int c0 = 0; c0 < n; c0 += GM_WAVE) {
    const int t = c0 + lane;
    uint64_t k = 0; uint32_t au = 0x8000u;
    if (t < n) { k = ws.key[t]; au = ws.aux[t]; }
    const bool alive = !(au & 0x8000u);
    const unsigned long long bal = __ballot(alive);
    k2_sync<false>();
    if (alive) { const int d = na + __popcll(bal & ((1ull << lane) - 1ull)); ws.key[d] = k; ws.aux[d] = au; }
    na += __popcll(bal);
    k2_sync<false>();
  }
  return na;
}

// ---- window generation (ref: mapping.c:1048-1207), one anchor per lane.  Returns the number of windows;
// DETECT additionally reports whether the open order of equal-x anchors could matter. ----------------
template <bool BIG, bool DETECT>
__device__ int k2_windows(const GmIndexDev& ix, const GmScoreDev& sc, K2Ws<BIG>& ws, int na, int read_len, int window_len,
                          GmHit* H, uint32_t* hitx, int hcap, int lane, bool* sensitive, int rs) {
  const int match = sc.match;
  // match_mode 3 (paired -n 3): an anchor whose region the mate reaches with a region it marked twice ("heavy_mp", ref: mapping.c:1080-1093) opens a window on its
  // own weight and skips the threshold (:1100-1103,1153-1157).  The mate's row: GmMpDev.
  const uint32_t mp_n = sc.match_mode == 3 ? ix.mp.cnt[rs ^ 1] : 0u;
  const uint32_t* mp_row = sc.match_mode == 3 ? ix.mp.rows + (size_t)(rs ^ 1) * GM_MP_CAP : nullptr;
  const int mp_dmin = ix.mp.dmin[rs & 1], mp_dmax = ix.mp.dmax[rs & 1];
  int nh = 0; bool sens = false;
  for (int c0 = 0; c0 < na; c0 += GM_WAVE) {
    const int i = c0 + lane;
    bool pass = false; GmHit h; uint32_t myx = 0;
    if (i < na) {
      const uint64_t ai = ws.key[i]; const uint32_t aui = ws.aux[i];
      const long long xi = (long long)(ai >> 32); const int yi = (int)(aui & 0x7FFF);
      const int leni = (int)((ai >> 16) & 0xFFFF), wi = (int)(ai & 0xFFFF);
      const int cn = (int)(aui >> 16);
      myx = (uint32_t)xi;
      const long long coff = ix.contig_off[cn];
      const long long clen = (long long)ix.contig_off[cn + 1] - coff;
      int w_len = window_len;
      if ((long long)w_len > clen) w_len = (int)clen;
      long long gend = (xi - coff) + read_len - 1 - yi;
      if (gend > clen - 1) gend = clen - 1;
      const long long gstart = (gend >= window_len) ? gend - window_len : 0;
      int max_idx = i;
      int max_score = leni * match;
      bool heavy_mp = false;
      if (sc.match_mode == 3 && mp_n <= (uint32_t)GM_MP_CAP) {
        const uint32_t reg = (uint32_t)xi >> ix.region_bits;
        heavy_mp = gm_mp_reach(mp_row, mp_n, (long long)reg + mp_dmin, (long long)reg + mp_dmax);
        if (!heavy_mp && reg > 0 && ((uint32_t)xi & ((1u << ix.region_bits) - 1u)) < (uint32_t)ix.region_overlap)
          heavy_mp = gm_mp_reach(mp_row, mp_n, (long long)reg - 1 + mp_dmin, (long long)reg - 1 + mp_dmax);
      }
      if (!sc.gapless && (sc.match_mode == 2 || (sc.match_mode == 3 && !heavy_mp)) && wi == 1) max_score = -1;
      bool tie_sibling = false;        // another candidate with the argmax's x reached the same score
      for (int j = i - 1; !sc.gapless && j >= 0; j--) {           // -U: only the anchor itself, no threshold (ref: mapping.c:1070,1095,1154)
        const uint64_t aj = ws.key[j];
        const long long xj = (long long)(aj >> 32);
        if (xj < coff + gstart) break;
        const int yj = (int)(ws.aux[j] & 0x7FFF);
        if (yj >= yi) continue;
        int short_len, long_len;
        if (xi - yi > xj - yj) { short_len = (yi - yj) + leni; long_len = (int)(xi - xj) + leni; }
        else { short_len = (int)(xi - xj) + leni; long_len = (yi - yj) + leni; }
        int tmp;
        if (long_len > short_len) tmp = short_len * match - sc.b_go - (long_len - short_len) * sc.b_ge;   // ref :1133-1135 (b_ penalties both ways)
        else tmp = short_len * match;
        if (tmp > max_score) { max_idx = j; max_score = tmp; tie_sibling = false; }
        else if (DETECT && tmp == max_score && max_idx != i && xj == (long long)(ws.key[max_idx] >> 32)) tie_sibling = true;
      }
      const int base = (read_len < w_len ? read_len : w_len) * match;
      if (sc.gapless || sc.match_mode == 1 || (sc.match_mode == 3 && heavy_mp) || max_score >= k2_threshold(sc.wgen_thr_frac, sc.wgen_abs, base)) {
        const uint64_t am = ws.key[max_idx]; const uint32_t aum = ws.aux[max_idx];
        const long long xm = (long long)(am >> 32);
        if (DETECT && max_idx != i && (tie_sibling || xm == xi)) sens = true;          // case (a)
        const int x_len = (int)(xi - xm) + leni;
        long long goff;
        if ((window_len - x_len) / 2 < xm - coff) goff = (xm - coff) - (window_len - x_len) / 2; else goff = 0;
        if (goff + w_len > clen) goff = clen - w_len;
        K2Box b;
        if (max_idx < i) {
          b = k2_join2(xi - coff - goff, yi, leni, 1, xm - coff - goff, (int)(aum & 0x7FFF), (int)((am >> 16) & 0xFFFF), 1);
        } else { b.x = xi - coff - goff; b.y = yi; b.length = leni; b.width = 1; }
        h.g_off = (uint32_t)goff; h.ax = (int32_t)b.x; h.ay = (int32_t)b.y; h.alen = b.length; h.awidth = b.width;
        h.score_window_gen = max_score; h.score_vector = -1; h.pct_score_vector = 0;
        h.cn = (uint16_t)cn; h.w_len = (uint16_t)w_len;
        const int mt = (max_idx == i) ? wi : wi + (int)(am & 0xFFFF);
        h.matches = (uint16_t)min(mt, 0xFFFF); h.flags = 0;
        pass = true;
      }
    }
    const unsigned long long bal = __ballot(pass);
    if (pass) {
      const int slot = nh + __popcll(bal & ((1ull << lane) - 1ull));
      if (slot < hcap) { H[slot] = h; if (DETECT) hitx[slot] = myx; }
    }
    nh += __popcll(bal);
  }
  if (DETECT) *sensitive = __any(sens);
  return nh;
}

template <bool BIG, int RMAX = 16>
__global__ void __launch_bounds__(GM_WAVE)
k_anchors(GmIndexDev ix, GmScoreDev sc, int n_reads, int read_len, int window_len, int max_n_kmers, int NL,
          const uint64_t* __restrict__ surv, const uint32_t* __restrict__ surv_cnt, int scap,
          // heavy tier only: list of read-strands, their segments in big_keys/big_aux/big_nxt/big_ord
          int n_heavy, const uint32_t* __restrict__ heavy_list, const uint64_t* __restrict__ seg_off, const uint32_t* __restrict__ seg_n,
          uint64_t* __restrict__ big_keys, uint32_t* __restrict__ big_aux, uint32_t* __restrict__ big_nxt, uint32_t* __restrict__ big_ord,
          GmHit* __restrict__ hits, uint16_t* __restrict__ perm, uint32_t* __restrict__ hit_cnt, int hcap,
          unsigned long long* __restrict__ stats,
          // LDS tier only: this launch takes the read-strands with lmin < survivors <= lcap (its LDS arrays hold lcap); the launch with lmin == 0 also writes the empty and the heavy ones off
          int lcap, int lmin) {
  typedef typename K2Idx<BIG>::type idx_t;
  extern __shared__ __align__(16) uint8_t smem_raw[];
  __shared__ int sh_na;
  const int lane = threadIdx.x;
#ifdef K2_STAMPS
  unsigned long long t_prev = __builtin_amdgcn_s_memtime();
#endif
  int rs, n;
  K2Ws<BIG> ws;
  uint8_t* base = smem_raw;
  if (BIG) {
    if ((int)blockIdx.x >= n_heavy) return;
    rs = (int)heavy_list[blockIdx.x];
    if ((sc.skip_strands >> (rs & 1)) & 1) { if (lane == 0) hit_cnt[rs] = 0; return; }                   // -C / -F
    n = (int)seg_n[blockIdx.x];
    const uint64_t o = seg_off[blockIdx.x];
    ws.key = big_keys + o; ws.aux = big_aux + o; ws.nxt = (idx_t*)(big_nxt + o); ws.ord = (idx_t*)(big_ord + o);
  } else {
    rs = blockIdx.x;
    const uint32_t n_all = ((sc.skip_strands >> (rs & 1)) & 1) ? 0u : surv_cnt[rs];                     // -C / -F: this strand has no anchor list (ref: mapping.c:879-880)
    if (n_all == 0 || n_all > (uint32_t)scap) { if (lane == 0 && lmin == 0) hit_cnt[rs] = 0; return; }   // > scap: heavy tier
    if (n_all <= (uint32_t)lmin || n_all > (uint32_t)lcap) return;                       // another launch's
    n = (int)n_all;
    ws.key = (uint64_t*)base;                    base += (size_t)lcap * 8;
    ws.aux = (uint32_t*)base;                    base += (size_t)lcap * 4;
    ws.nxt = (idx_t*)base;                       base += (size_t)lcap * sizeof(idx_t);
    ws.ord = (idx_t*)base;                       base += (size_t)lcap * sizeof(idx_t);
  }
  ws.hk = (uint32_t*)base;                       base += (size_t)NL * 4;
  ws.first = (idx_t*)base;                       base += (size_t)((NL + 1) & ~1) * sizeof(idx_t);
  ws.hr = (uint16_t*)base;                       base += (size_t)((NL + 1) & ~1) * 2;
  ws.cache = (int16_t*)base;
  uint32_t* scratch32 = (uint32_t*)ws.nxt;       // LDS tier: nxt+ord = u32[scap]; heavy tier: nxt = u32[cap]
  GmHit* H = hits + (size_t)rs * hcap;
  int npad = 64; while (npad < n) npad <<= 1;

  // loads the survivors in canonical order and fills aux = y | contig<<16 (get_contig_num, ref: gmapper.h:373-405)
  auto load_sorted = [&]() {
    if (!BIG) {
      const uint64_t* sv = surv + (size_t)rs * scap;
      for (int t = lane; t < npad; t += GM_WAVE) ws.key[t] = (t < n) ? sv[t] : ~0ull;
      k2_sync<BIG>();
      k2_sort<BIG, uint64_t, RMAX>(ws.key, npad, lane);
    }
    K2_STAMP(0);
    if (!BIG && ix.n_contigs <= GM_WAVE) {
      // up to 64 contigs: lane l keeps contig_off[l], the binary search gathers from the lanes (six uniform steps, no memory round trip per step;
      // every lane takes part in every gather, the ones past n search for position 0)
      const uint32_t my_off = lane < ix.n_contigs ? ix.contig_off[lane] : 0xFFFFFFFFu;
      for (int t0 = 0; t0 < n; t0 += GM_WAVE) {
        const int t = t0 + lane;
        const uint64_t k = t < n ? ws.key[t] : 0ull;
        const uint32_t x = (uint32_t)(k >> 32);
        int lo = 0, hi = ix.n_contigs;
        for (int it = 0; it < 6; it++) {
          const int m = (lo + hi) >> 1;
          const uint32_t v = (uint32_t)__shfl((int)my_off, m);
          const bool open = hi - lo > 1, le = v <= x;
          lo = (open && le) ? m : lo; hi = (open && !le) ? m : hi;
        }
        if (t < n) ws.aux[t] = (uint32_t)((k >> 16) & 0x7FFF) | ((uint32_t)lo << 16);
      }
    } else
    for (int t = lane; t < n; t += GM_WAVE) {
      const uint64_t k = ws.key[t];
      const uint32_t x = (uint32_t)(k >> 32);
      int lo = 0, hi = ix.n_contigs;
      while (hi - lo > 1) { int m = (lo + hi) >> 1; if (ix.contig_off[m] <= x) lo = m; else hi = m; }
      ws.aux[t] = (uint32_t)((k >> 16) & 0x7FFF) | ((uint32_t)lo << 16);
    }
    k2_sync<BIG>();
  };

  // orders the windows by (contig, g_off), stable == the reference's insertion sort (ref: mapping.c:1210-1223);
  // detect: case (b), equal (contig, g_off) windows made from equal-x anchors (hitx = scratch32)
  auto sort_windows = [&](int nhc, bool detect) -> bool {
    bool sens = false;
    uint16_t* P = perm + (size_t)rs * hcap;
    if (nhc > 0) {
      int hp = 64; while (hp < nhc) hp <<= 1;
      // nhc <= na <= n, and every key array holds pow2ceil(n) >= hp entries.
      // The records were written by other lanes of this wave: read them back past the L1 (sc1 loads).
      for (int t = lane; t < hp; t += GM_WAVE) {
        uint64_t k = ~0ull;
        if (t < nhc) {
          const uint32_t go = __hip_atomic_load(&H[t].g_off, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          const uint32_t cw = __hip_atomic_load((const uint32_t*)&H[t].cn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          k = ((uint64_t)(cw & 0xFFFF) << 48) | ((uint64_t)go << 16) | (uint64_t)t;
        }
        ws.key[t] = k;
      }
      k2_sync<BIG>();
      k2_sort<BIG, uint64_t, RMAX>(ws.key, hp, lane);
      for (int t = lane; t < nhc; t += GM_WAVE) {
        P[t] = (uint16_t)(ws.key[t] & 0xFFFF);
        if (detect && t > 0 && (ws.key[t] >> 16) == (ws.key[t - 1] >> 16) &&
            scratch32[ws.key[t] & 0xFFFF] == scratch32[ws.key[t - 1] & 0xFFFF]) sens = true;
      }
    }
    return __any(sens);
  };

  int na = 0, nh = 0;
  bool need_exact = BIG;
  load_sorted();
  K2_STAMP(1);
  if (!BIG) {
    // ---- fast path: order-free collapse + windows in canonical order, with sensitivity detection ----
    na = k2_collapse_par<RMAX>(ix, *(K2Ws<false>*)&ws, scratch32, n, npad, read_len, lane);
    K2_STAMP(2);
    bool sens = false;
    nh = k2_windows<BIG, true>(ix, sc, ws, na, read_len, window_len, H, scratch32, hcap, lane, &sens, rs);
    k2_sync<BIG>();
    K2_STAMP(3);
    need_exact = sens;
    if (!need_exact) need_exact = sort_windows(min(nh, hcap), true);
    K2_STAMP(4);
  }
  if (need_exact) {
    // ---- exact path: reference pop order (heap replay when a position carries two read offsets) ----
    if (!BIG) { k2_sync<BIG>(); load_sorted(); }
    bool danger = false;
    for (int t = lane + 1; t < n; t += GM_WAVE) {
      const uint64_t a = ws.key[t - 1], b = ws.key[t];
      danger |= ((a >> 32) == (b >> 32)) && (((a >> 16) & 0xFFFF) != ((b >> 16) & 0xFFFF));
    }
    danger = __any(danger);
    for (int d = lane; d < read_len; d += GM_WAVE) ws.cache[d] = -1;
    k2_sync<BIG>();
    if (lane == 0) {
      GS_ADD(stats, GS_EXACT_ORDER, 1ull);
      if (danger) k2_exact_order<BIG>(ws, n, NL, max_n_kmers);
      sh_na = k2_collapse_seq<BIG>(ix, ws, n, read_len);
    }
    k2_sync<BIG>();
    na = sh_na;
    bool dummy;
    nh = k2_windows<BIG, false>(ix, sc, ws, na, read_len, window_len, H, scratch32, hcap, lane, &dummy, rs);
    k2_sync<BIG>();
    sort_windows(min(nh, hcap), false);
  }
  if (lane == 0) {
    hit_cnt[rs] = (uint32_t)nh;
    GS_ADD(stats, GS_ANCHORS, na);
    GS_ADD(stats, GS_WINDOWS, min(nh, hcap));
    if (nh > hcap) GS_ADD(stats, GS_OVERFLOW_HITS, 1ull);
  }
}

static size_t k2_lds_bytes(bool big, int scap, int NL, int read_len) {
  size_t b = (size_t)NL * 4 + (size_t)((NL + 1) & ~1) * (big ? 4 : 2) + (size_t)((NL + 1) & ~1) * 2 + (size_t)read_len * 2;
  if (!big) b += (size_t)scap * (8 + 4 + 2 + 2);
  return (b + 15) & ~(size_t)15;
}

int gm_launch_anchors(const GmIndexDev& ix, const GmScoreDev& sc, int n_reads, int read_len, int window_len,
                      const uint64_t* d_surv, const uint32_t* d_surv_cnt, int scap,
                      GmHit* d_hits, uint16_t* d_perm, uint32_t* d_hit_cnt, int hcap, unsigned long long* d_stats, hipStream_t stream) {
  const int max_n_kmers = std::max(0, read_len - ix.min_seed_span + 1);
  const int NL = ix.n_seeds * max_n_kmers;
  if (n_reads == 0) return GM_OK;
  // Where the LDS tier holds 2 048 survivors or more (150-base reads on 3 Gbp keep ~800 per read-strand, a fifth of the read-strands more than 1 024), two launches:
  // the read-strands of up to 1 024 survivors with LDS arrays of 1 024 (16 KB a wave: nine waves per CU instead of four) and the 1 024-key register sort, then the others
  // with the full arrays and the 2 048-key register sort (a kernel of its own: 248 registers).
  const bool split = scap >= 2048 && !(gm_tune("GM_K2_WIDE") && atoi(gm_tune("GM_K2_WIDE")) == 0);
  const size_t lds = k2_lds_bytes(false, scap, NL, read_len), lds_a = k2_lds_bytes(false, split ? 1024 : scap, NL, read_len);
  if (lds > 64 * 1024) {
    static GmLdsLimit lim_configured; size_t& configured = lim_configured.cur();
    if (lds > 160 * 1024) { gm_set_error("anchor kernel LDS %zu too large", lds); return GM_E_ARG; }
    if (lds > configured) {
      GM_HIP(hipFuncSetAttribute((const void*)k_anchors<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      GM_HIP(hipFuncSetAttribute((const void*)k_anchors<false, 32>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); configured = lds; }
  }
  hipLaunchKernelGGL(k_anchors<false>, dim3(n_reads * 2), dim3(GM_WAVE), lds_a, stream, ix, sc, n_reads, read_len, window_len, max_n_kmers, NL,
                     d_surv, d_surv_cnt, scap, 0, (const uint32_t*)nullptr, (const uint64_t*)nullptr, (const uint32_t*)nullptr,
                     (uint64_t*)nullptr, (uint32_t*)nullptr, (uint32_t*)nullptr, (uint32_t*)nullptr,
                     d_hits, d_perm, d_hit_cnt, hcap, d_stats, split ? 1024 : scap, 0);
  if (split)
  hipLaunchKernelGGL((k_anchors<false, 32>), dim3(n_reads * 2), dim3(GM_WAVE), lds, stream, ix, sc, n_reads, read_len, window_len, max_n_kmers, NL,
                     d_surv, d_surv_cnt, scap, 0, (const uint32_t*)nullptr, (const uint64_t*)nullptr, (const uint32_t*)nullptr,
                     (uint64_t*)nullptr, (uint32_t*)nullptr, (uint32_t*)nullptr, (uint32_t*)nullptr,
                     d_hits, d_perm, d_hit_cnt, hcap, d_stats, scap, 1024);
  GM_HIP(hipGetLastError());
  return GM_OK;
}

// heavy tier: segmented radix sort of the re-emitted keys (64-bit, all bits), then the exact path on global arrays
int gm_launch_anchors_heavy(const GmIndexDev& ix, const GmScoreDev& sc, int n_reads, int read_len, int window_len,
                            int n_heavy, const uint32_t* d_heavy_list, const uint64_t* d_seg_off, const uint32_t* d_seg_n,
                            const uint32_t* d_seg_begin32, const uint32_t* d_seg_end32, uint64_t total_keys,
                            uint64_t* d_keys_in, uint64_t* d_keys_sorted, uint32_t* d_aux, uint32_t* d_nxt, uint32_t* d_ord,
                            GmHit* d_hits, uint16_t* d_perm, uint32_t* d_hit_cnt, int hcap, unsigned long long* d_stats, hipStream_t stream) {
  if (n_heavy == 0) return GM_OK;
  const int max_n_kmers = std::max(0, read_len - ix.min_seed_span + 1);
  const int NL = ix.n_seeds * max_n_kmers;
  size_t tmp_bytes = 0; void* tmp = nullptr;
  hipError_t e = rocprim::segmented_radix_sort_keys(nullptr, tmp_bytes, d_keys_in, d_keys_sorted, (unsigned int)total_keys, (unsigned int)n_heavy,
                                                    d_seg_begin32, d_seg_end32, 0, 64, stream);
  if (e != hipSuccess) { gm_set_error("segmented_radix_sort_keys size query: %s", hipGetErrorString(e)); return GM_E_NODEVICE; }
  GM_HIP(hipMalloc(&tmp, tmp_bytes ? tmp_bytes : 16));
  e = rocprim::segmented_radix_sort_keys(tmp, tmp_bytes, d_keys_in, d_keys_sorted, (unsigned int)total_keys, (unsigned int)n_heavy,
                                         d_seg_begin32, d_seg_end32, 0, 64, stream);
  if (e != hipSuccess) { (void)hipFree(tmp); gm_set_error("segmented_radix_sort_keys: %s", hipGetErrorString(e)); return GM_E_NODEVICE; }
  const size_t lds = k2_lds_bytes(true, 0, NL, read_len);
  hipLaunchKernelGGL(k_anchors<true>, dim3(n_heavy), dim3(GM_WAVE), lds, stream, ix, sc, n_reads, read_len, window_len, max_n_kmers, NL,
                     (const uint64_t*)nullptr, (const uint32_t*)nullptr, 0, n_heavy, d_heavy_list, d_seg_off, d_seg_n,
                     d_keys_sorted, d_aux, d_nxt, d_ord, d_hits, d_perm, d_hit_cnt, hcap, d_stats, 0, 0);
  GM_HIP(hipGetLastError());
  GM_HIP(hipStreamSynchronize(stream));
  (void)hipFree(tmp);
  return GM_OK;
}
