// gm_lookup5.hip -- K1 v5: spaced-seed lookup + exact region filter + exact prune in ONE kernel (gfx950 only).
//
// Same contract as gm_lookup.hip (which keeps the bucket / slab-sweep / lane-per-list kernels and documents the
// reference lines replaced: read_get_mapidxs, read_get_region_counts, advance_index_in_genomemap,
// ref: gmapper/mapping.c:37-70,459-542,646-805) plus the exact prune of gm_prune.hip, for read-strands with many
// list entries (100 bp reads on a 3 Gbp genome: 251 lists of ~179 entries).  What is different from k_lookup_v4:
//
//  * One WAVE streams one list at a time: the list descriptor (pointer, length, y/seed) is wave-uniform (SGPRs),
//    lane l takes W = ceil(n / 64) consecutive entries of a chunk of n <= 256, read with one dwordx4.  No window
//    map, no prefix over the lists, no per-lane descriptor look-up.
//  * Pre-count in two 1-bit tables: seen[2^(lsw+5)] (region folded onto its low bits) and, an eighth of its size,
//    twice[]: the first mark of a counter sets its seen bit, every later one sets the twice bit.  Pass A marks,
//    pass B streams the lists again and keeps the entries whose twice bit is set -- a superset of the entries
//    whose region reaches the reference's count of 2 (a region's second mark always finds the seen bit set).
//  * The overlap strip (an entry in the first region_overlap bases of a region also counts for the region before,
//    ref: mapping.c:521-533,733-742) is not tested per entry: the index carries, per list, the few strip entries
//    again as a second short list (sdir / spos, derived on the device from dir / pos, 2.4 % of the entries), and
//    two small dense loops mark / test region - 1 for exactly those.
//  * The candidates (~13 % of the entries) stay in LDS -- pass B needs only twice[], so they are written over the
//    dead seen[] table -- and are resolved there: an open-addressing table keyed by region holds the exact mark
//    count (the reference's rule; exact because every entry of a region with count >= 2 is a candidate) and, per
//    region, the number and the min / max offset of the candidates inside it, from which the prune rules of
//    gm_prune.hip are evaluated with region-sized bins.  Judging the prune on candidates instead of survivors
//    only keeps more (both rules are monotone: more neighbours => keep), which is still exact.
//  * 6 workgroup barriers per read-strand (v4: ~25), no candidate scratch in global memory.
//
// Read-strands whose candidates do not fit (repeats) are redone by k_lookup<false> in list mode + k_prune in
// list mode (gm_lookup.hip / gm_prune.hip).
#include <algorithm>
#include <mutex>
#include "gm_common.h"
#include "gm_internal.h"
#include <rocprim/device/device_scan.hpp>

typedef uint32_t k5_u32x4 __attribute__((ext_vector_type(4), aligned(4)));
typedef __attribute__((address_space(3))) uint32_t k5_lds_u32;
// LDS word at an absolute LDS byte address (the tables of k_lookup_v5 start at the workgroup's LDS base, which the kernel checks to be 0:
// table offsets are then OR-ed, not added, into the address)
#define K5_LDS(off) ((k5_lds_u32*)(uintptr_t)(uint32_t)(off))
typedef __attribute__((address_space(3))) uint16_t k5_lds_u16;
#define K5_LDS16(off) ((k5_lds_u16*)(uintptr_t)(uint32_t)(off))
#define K5_LDS_OR(off, m) __hip_atomic_fetch_or(K5_LDS(off), (m), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)

#ifndef K5_Q
#define K5_Q 6            // list chunks in flight per wave (8 and 10 measured: no gain)
#endif
#define K5_XREC 64                                      // chunk records beyond one per list (a list of more than 256 positions takes one record per 256)
enum { C_NCAND = 0, C_OVERFLOW, C_NMEMB, C_NKEEP, C_NSTRIP, C_NEDGE, C_NLISTS /* two words: read-strands alternate */, C_SINK = 8, C_NRAW = 9, C_NDEFER = 10, C_WORDS = 12 };

// Diagnostic build (-DK5_STAMPS): thread 0 of every workgroup adds the cycles between the phase boundaries of each read-strand to k5_stamps[]
// (0-5: setup, pass A, pass B, region table, rules + output, clears; 6, 7: candidates, fallbacks; 8-11: parts of set-up / the exact stages, taken out of
// the phase they sit in); gm_debug_k5_stamps() reads and resets all 16.  No stamp executes in the normal build.
#ifdef K5_STAMPS
__device__ unsigned long long k5_stamps[16];
#define K5_STAMP(i) do { if (tid == 0) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); atomicAdd(&k5_stamps[i], t_ - t_prev); t_prev = t_; } } while (0)
extern "C" int gm_debug_k5_stamps(unsigned long long* out) {
  unsigned long long z[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(k5_stamps), sizeof z) != hipSuccess) return GM_E_NODEVICE;
  if (hipMemcpyToSymbol(HIP_SYMBOL(k5_stamps), z, sizeof z) != hipSuccess) return GM_E_NODEVICE;
  return GM_OK;
}
#else
#define K5_STAMP(i) do { } while (0)
#endif

struct K5Args {
  const uint32_t* reads; int n_reads, read_len, read_words, max_n_kmers, NL;
  int lsw, ltw;             // log2(words) of seen[] and twice[]
  int cand_cap, hbits;      // candidate records and log2(slots of the region table), both inside the seen[] area
  int cand_limit;
  uint64_t* out; uint32_t* out_cnt; int out_cap; uint32_t* surv_cnt;
  int prune; uint32_t D; int e_max;
  uint32_t* heavy_list; uint32_t* heavy_cnt; int heavy_cap;
  unsigned long long* stats;
  uint32_t* fb_list; uint32_t* fb_cnt; int fb_cap;
  uint32_t* start_flags; uint32_t start_epoch;
  // fused prune only: a read-strand that keeps more than out_cap survivors under the region-sized bins writes ALL its members to its raw row and goes to pl_list --
  // k_prune (256-base bins) prunes it from there; without this the lane-per-list kernel had to produce the members again
  uint64_t* raw_out; int raw_cap; uint32_t* surv_seg; int n_slabs; uint32_t* pl_list; uint32_t* pl_cnt; int pl_cap;
};

// lane * W for a wave-uniform W in 1..4 (the 24-bit multiply is a full-rate instruction)
__device__ __forceinline__ uint32_t k5_lane_times(int lane, uint32_t W) {
  uint32_t r; asm("v_mul_u32_u24 %0, %1, %2" : "=v"(r) : "v"(lane), "s"(W)); return r;     // (the compiler turns __umul24 by a uniform factor into the quarter-rate v_mul_lo_u32)
}

#include "gm_region_table.h"

// LSWC: log2(words) of seen[] as a compile-time constant (the production size, 15), or 0 for the size in K5Args (small-table test variants, long reads):
// with constants the table masks are immediates and the base of seen[] folds into the LDS instructions' offset field.
template <int LSWC>
__global__ void __launch_bounds__(1024)
k_lookup_v5(GmIndexDev ix, K5Args a) {
  if (a.start_flags && threadIdx.x == 0) __hip_atomic_store(&a.start_flags[blockIdx.x], a.start_epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  __builtin_amdgcn_s_setprio(3);
  extern __shared__ __align__(16) uint32_t smem[];
  const int tid = threadIdx.x, nthr = blockDim.x, lane = tid & (GM_WAVE - 1);
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6), nwv = nthr >> 6;
  const int S = ix.n_slabs, rb = ix.region_bits, lsw = LSWC ? LSWC : a.lsw, ltw = lsw - 3, hbits = lsw - 2;
  const int cand_cap = LSWC ? (int)(((4u << LSWC) / 4u / 6u) & ~15u) : a.cand_cap;
  const uint32_t rmask = (1u << rb) - 1u, ovl = (uint32_t)ix.region_overlap;
  const uint32_t wmask = (1u << lsw) - 1u, tmask = (1u << ltw) - 1u;
  // LDS: twice | seen (later: candidates | region table) | rec[NL] (4 words) | srec[NL] (4 words) | codes | ctrl
  // Byte offsets into the tables come straight from the position: twice word of p at (p >> (rb - 2)) & tmask4 (twice[] starts at 0: no base to add), its
  // seen word at swb + ((p >> (rb - 2)) & smask4) (swb = the size of twice[]: an instruction offset when the sizes are compile-time), bit (p >> (rb + lsw)) & 31.
  if ((uint32_t)(uintptr_t)(k5_lds_u32*)smem != 0u) __builtin_trap();      // no static LDS in this kernel: the dynamic segment starts at 0
  uint32_t* twice = smem;
  uint32_t* seen = smem + (1u << ltw);
  // after pass A the seen[] area is re-used: candidate positions (cand_cap words) | candidate y / seed (u16) -- cand_cap = the first quarter / 6 bytes; with the
  // production sizes both arrays lie below 64 KB, so that pass B's stores carry their base in the LDS instruction's offset field -- then htag | hmin | hmax
  // (2^hbits = 2^(lsw - 2) words each: the other three quarters)
  uint32_t* candp = seen;
  uint16_t* candy = (uint16_t*)(candp + cand_cap);
  uint32_t* htag = seen + (1u << hbits);
  uint32_t* hmin = htag + (1u << hbits);
  uint32_t* hmax = hmin + (1u << hbits);
  uint32_t* rec = seen + (1u << lsw);
  const int RC = a.NL + K5_XREC;                                 // records: one per chunk of at most 256 positions
  uint32_t* srec = rec + 4 * RC;
  uint8_t* codes = (uint8_t*)(srec + 4 * RC);
  uint32_t* ctrl = (uint32_t*)(codes + 2 * ((a.read_len + 15) & ~15));           // two code buffers (this read-strand's and the next one's)
  const uint32_t smask4 = wmask << 2, tmask4 = tmask << 2, swb = 4u << ltw;
  const int sh_a = rb - 2, sh_b = rb + lsw;
  const uint32_t hmask = (1u << hbits) - 1u; const int hshift = 32 - hbits;
  const int hclr_q = (int)(3u << hbits) >> 2;                         // uint4 words of the region table (cleared during pass B)
  const int tab_q = (int)(((1u << ltw) + (1u << lsw)) >> 2);             // uint4 words of twice + seen
  const uint32_t* __restrict__ pos0 = ix.seed[0].pos;
  const uint32_t* __restrict__ spos0 = ix.seed[0].spos;
  unsigned long long my_lookups = 0, my_entries = 0;
  // the prune rules' arguments as opaque scalars, fixed here in uniform control flow (see stage 2a)
  uint32_t k_prune = __builtin_amdgcn_readfirstlane(a.prune != 0 ? 1u : 0u), k_D = __builtin_amdgcn_readfirstlane(a.D);
  uint32_t k_emax_on = __builtin_amdgcn_readfirstlane(a.e_max >= 0 ? 1u : 0u), k_emax = __builtin_amdgcn_readfirstlane((uint32_t)max(a.e_max, 0));
  asm volatile("" : "+s"(k_prune), "+s"(k_D), "+s"(k_emax_on), "+s"(k_emax));

  { uint4* t4 = (uint4*)smem; for (int w = tid; w < tab_q; w += nthr) t4[w] = make_uint4(0, 0, 0, 0); }
  if (tid < C_WORDS) ctrl[tid] = 0;                            // thread 0 re-zeroes them at the end of every read-strand

  auto mark = [&](const uint32_t p) {                          // one mark of region p >> rb (strip loop; the main loop has its own batched form)
    const uint32_t av = p >> sh_a, m = 1u << ((p >> sh_b) & 31u);
    const uint32_t old = K5_LDS_OR(swb + (av & smask4), m);
    if (old & m) K5_LDS_OR(av & tmask4, m);
  };
  auto has2 = [&](const uint32_t p) -> bool { return (*K5_LDS((p >> sh_a) & tmask4) >> ((p >> sh_b) & 31u)) & 1u; };

#ifdef K5_STAMPS
  unsigned long long t_prev = __builtin_amdgcn_s_memtime();
#endif
  // Set-up of a read-strand (k-mers -> map indexes -> directory entries) one read-strand ahead: while the passes of read-strand n run, the
  // directory words of read-strand n + 1 are already on their way (two dependent HBM round trips and a 19-step loop per k-mer off the critical
  // path).  One k-mer per thread (NL <= threads); longer reads take the plain path.
  const bool ahead = a.NL <= nthr;
  const uint32_t cpad = (uint32_t)((a.read_len + 15) & ~15);     // (code buffer b = codes + b * cpad: pointer arithmetic keeps the LDS address space, a pointer array does not)
  uint32_t pf_b = 0, pf_e = 0, pf_sb = 0, pf_se = 0; bool pf_ok = false;
  auto fill_codes = [&](const int r, uint8_t* cb) {
    if (r >= 2 * a.n_reads) return;
    const uint32_t* rw = a.reads + (size_t)(r >> 1) * a.read_words;
    for (int i = tid; i < a.read_len; i += nthr) cb[i] = (uint8_t)gm_read_code(rw, a.read_len, (r & 1) ^ ix.cs_flip, ix.colour, i);
  };
  // A thread keeps the same k-mer slot (seed, offset in the read) for every read-strand: its seed's span and mask stay in registers, the seed's
  // pointers in a small LDS table.  (ix.seed[sn] with a run-time sn is a chain of loads from the kernel-argument segment -- a memory round trip
  // per field and use: 4 k cycles per read-strand before.)
  uint32_t* stab = ctrl + C_WORDS;                             // per seed: dir, sdir (pointers), pos - pos0, spos - spos0 (elements)
  if (tid < ix.n_seeds) {
    const GmSeedDev& sd = ix.seed[tid];
    const uint64_t d = (uint64_t)(uintptr_t)sd.dir, sdp = (uint64_t)(uintptr_t)sd.sdir, po = (uint64_t)(sd.pos - pos0), so = (uint64_t)(sd.spos - spos0);
    *(uint4*)&stab[8 * tid] = make_uint4((uint32_t)d, (uint32_t)(d >> 32), (uint32_t)sdp, (uint32_t)(sdp >> 32));
    *(uint4*)&stab[8 * tid + 4] = make_uint4((uint32_t)po, (uint32_t)(po >> 32), (uint32_t)so, (uint32_t)(so >> 32));
  }
  int k_sn = 0, k_i = 0, k_span = 0; uint64_t k_mask = 0; bool k_ok = false;
  // The k-mer slots sit on the first threads: ceil(NL / 64) waves ("set-up waves", 4 of 16 at 100 bp) work out the map indexes of the next read-strand, the
  // other waves skip that arithmetic (their directory loads are dummies) and go straight to pass A, where they take one list each ahead of the rota
  // (see the step generator) -- the set-up costs a quarter of the vector instructions it took when every stride-th thread of all the waves held a slot.
  const int k_stride = 1, k_slot = tid;
  const int nsw = ahead ? min(nwv, (a.NL + GM_WAVE - 1) / GM_WAVE) : nwv;      // set-up waves
  if (ahead && k_slot < a.NL) {
    k_sn = k_slot / a.max_n_kmers; k_i = k_slot - k_sn * a.max_n_kmers;
    k_span = ix.seed[k_sn].span; k_mask = ix.seed[k_sn].mask;
    k_ok = k_i >= ix.colour && k_i + k_span <= a.read_len;
    if (!k_ok) { k_span = 0; k_mask = 0; }
  }
  // One record per chunk of at most 256 positions (the list's address + the chunk's, its length, y / seed), so that the step generator of the two passes is
  // "next record": a list longer than 256 takes several; its strip list rides on the first one.  Records beyond the array are dropped, the read-strand then
  // falls back (rec_over below).  `cnl` is this read-strand's counter word.
  uint32_t* cnl_w = nullptr;
  auto put_records = [&](const uint64_t la, const uint32_t len, const uint32_t ysn, const uint64_t sptr, const uint32_t slen) {
    const uint32_t nch = (len + 255u) >> 8;
    const uint32_t j = atomicAdd(cnl_w, nch);
    for (uint32_t c = 0; c < nch && j + c < (uint32_t)RC; c++) {
      const uint64_t ca = la + ((uint64_t)c << 10);
      *(uint4*)&rec[4 * (j + c)] = make_uint4((uint32_t)ca, (uint32_t)(ca >> 32), min(256u, len - (c << 8)), ((ysn >> 12) & 0xFFF0u) | (ysn & 0xFu));   // y << 4 | seed: as the candidates carry it
      *(uint4*)&srec[4 * (j + c)] = c ? make_uint4(0u, 0u, 0u, 0u) : make_uint4((uint32_t)sptr, (uint32_t)(sptr >> 32), slen, 0u);
    }
  };
  auto kmer_ahead = [&](const int r, const uint8_t* cb) {      // the four directory words of this thread's k-mer of read-strand r: loads issued, used at the next top
    pf_ok = k_ok && r < 2 * a.n_reads;
    uint32_t mapidx = 0;
    if (wv >= nsw) {}                                        // (no slot in this wave: wave-uniform)
    else if (!ix.hflag) {
      // KMER_TO_MAPIDX (ref: gmapper.h:349-368) without a branch per base: eight code bytes per LDS round trip, the mask bit selects
      for (int t0 = 0; t0 < ix.max_seed_span; t0 += 8) {
        uint32_t c[8];
#pragma unroll
        for (int u = 0; u < 8; u++) { const int x = k_i + k_span - 1 - t0 - u; c[u] = cb[min(max(x, 0), a.read_len - 1)]; }
#pragma unroll
        for (int u = 0; u < 8; u++) mapidx = ((k_mask >> (t0 + u)) & 1ull) ? ((mapidx << 2) | (c[u] & 3u)) : mapidx;
      }
    } else if (pf_ok) mapidx = gm_mapidx(ix, k_mask, k_span, cb + k_i);
    if (!pf_ok) mapidx = 0;
    // (unconditional loads and, at the next top, an unconditional use: conditional ones leave the compiler's counter model with "maybe pending" loads
    // and it then waits for vmcnt(0) inside the streaming loops)
    const uint4 sp = *(const uint4*)&stab[8 * k_sn];
    // (pointers rebuilt from integers must name the global address space: a flat load also counts as an LDS operation, and every LDS wait
    // behind it would wait for the memory round trip)
    typedef const uint32_t __attribute__((address_space(1)))* k5_gptr;
    const k5_gptr dir = (k5_gptr)(((uint64_t)sp.y << 32) | sp.x) + (size_t)mapidx * (uint32_t)S;
    const k5_gptr sdir = (k5_gptr)(((uint64_t)sp.w << 32) | sp.z) + mapidx;
    pf_b = dir[0]; pf_e = dir[S]; pf_sb = sdir[0]; pf_se = sdir[1];
  };
  __syncthreads();
  if (ahead) { fill_codes(blockIdx.x, codes); __syncthreads(); kmer_ahead(blockIdx.x, codes); }
  int it = 0;
  for (int rs = blockIdx.x; rs < 2 * a.n_reads; rs += gridDim.x, it++) {
    // (the list counter alternates between two words: with the set-up ahead there is no barrier between thread 0's reset at the end of a
    // read-strand and the first additions of the next one; the other word was reset a whole read-strand earlier)
    uint32_t* const cnl = &ctrl[C_NLISTS + (it & 1)];
    cnl_w = cnl;
    if (ahead) {
      fill_codes(rs + (int)gridDim.x, codes + (uint32_t)((it + 1) & 1) * cpad);
      const uint32_t b = pf_b, e = pf_e, sb = pf_sb, se = pf_se;
      if ((b ^ e ^ sb ^ se) == 0x9E3779B9u && e - b == 0x7F4A7C15u) ctrl[C_SINK] = 1u;        // never true for list bounds: pins the use of all four words here
      if (pf_ok) {
        my_lookups++;
        if (e != b && e - b <= ix.list_cutoff) {               // ref: mapping.c:497 (longer lists are skipped, not deleted)
          const uint4 so = *(const uint4*)&stab[8 * k_sn + 4];
          my_entries += (e - b);
          const uint64_t ptr = (((uint64_t)so.y << 32) | so.x) + b, sptr = (((uint64_t)so.w << 32) | so.z) + sb;
          put_records((uint64_t)(uintptr_t)(pos0 + ptr), e - b, ((uint32_t)k_i << 16) | (uint32_t)k_sn, sptr, se - sb);
        }
      }
      __syncthreads();                                         // rec[] of this read-strand and the codes of the next one are in; the tables are clear
      K5_STAMP(8);
      kmer_ahead(rs + (int)gridDim.x, codes + (uint32_t)((it + 1) & 1) * cpad);
      K5_STAMP(9);
    } else {
    const int rd = rs >> 1, st = rs & 1;
    const uint32_t* rw = a.reads + (size_t)rd * a.read_words;
    for (int i = tid; i < a.read_len; i += nthr) codes[i] = (uint8_t)gm_read_code(rw, a.read_len, st ^ ix.cs_flip, ix.colour, i);
    __syncthreads();                                           // also: the tables and the control words are clear
    // ---- map indexes, list bounds, strip-list bounds (ref: mapping.c:53-66, KMER_TO_MAPIDX gmapper.h:349-368) ----
    for (int off = tid; off < a.NL; off += nthr) {
      const int sn = off / a.max_n_kmers, i = off - sn * a.max_n_kmers;
      const int span = ix.seed[sn].span;
      if (i < ix.colour || i + span > a.read_len) continue;
      const uint32_t mapidx = gm_mapidx(ix, ix.seed[sn].mask, span, codes + i);
      const uint32_t* dir = ix.seed[sn].dir + (size_t)mapidx * (uint32_t)S;
      my_lookups++;
      const uint32_t b = dir[0], e = dir[S];
      if (e == b || e - b > ix.list_cutoff) continue;          // ref: mapping.c:497 (longer lists are skipped, not deleted)
      const uint32_t sb = ix.seed[sn].sdir[mapidx], se = ix.seed[sn].sdir[mapidx + 1];
      my_entries += (e - b);
      const uint64_t sptr = (uint64_t)((ix.seed[sn].spos + sb) - spos0);
      put_records((uint64_t)(uintptr_t)(ix.seed[sn].pos + b), e - b, ((uint32_t)i << 16) | (uint32_t)sn, sptr, se - sb);
    }
    __syncthreads();
    }
    // (no barrier behind the k-mers ahead: they read the next read-strand's codes and the seed table only, and their scattered loads -- 64 cache
    // lines per instruction -- queue up in the address unit; the waves go on to pass A as they get through)
    K5_STAMP(0);
    const int nl_raw = (int)*cnl, nl = min(nl_raw, RC);
    const bool rec_over = nl_raw > RC;                          // more chunks than records: the read-strand is redone by the fall-back kernels

    // One chunk of a list: n <= 256 entries from src, W = ceil(n / 64) per lane.  Wave-uniform (SGPRs).
    struct Step { uint32_t n, ysn; const uint32_t* src; };
    // generator: the next record of this wave.  The waves without k-mer slots reach pass A earlier than the set-up waves: each of them takes one record of
    // the first `xh` ahead of the rota (record wv - nsw), then every wave strides through the rest (xh + wv, xh + wv + nwv, ...).
    const int xh = min(nl, nwv - nsw);
    int gj = wv, gnext = wv;
    uint4 gr = make_uint4(0, 0, 0, 0);                         // that record, read one step ahead of its use (an LDS round trip off the critical path)
    auto gen_reset = [&]() { gnext = xh + wv; gj = (wv >= nsw && wv - nsw < xh) ? wv - nsw : gnext; if (gj == gnext) gnext += nwv; if (gj < nl) gr = *(const uint4*)&rec[4 * gj]; };
    auto gen = [&](Step& s) {
      s.n = 0; s.ysn = 0; s.src = pos0;
      if (gj < nl) {
        const uint64_t o = ((uint64_t)__builtin_amdgcn_readfirstlane(gr.y) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane(gr.x);
        s.n = __builtin_amdgcn_readfirstlane(gr.z); s.ysn = __builtin_amdgcn_readfirstlane(gr.w);
        s.src = (const uint32_t*)(const uint32_t __attribute__((address_space(1)))*)(uintptr_t)o;   // (rebuilt from integers: it has to name the global address space, else the loads become flat loads)
        gj = gnext; gnext += nwv; if (gj < nl) gr = *(const uint4*)&rec[4 * gj];
      }
    };
    // EVERY step issues exactly one dwordx4 per lane -- lanes past the chunk re-read its last entry, an exhausted generator reads the first
    // words of pos[] -- so that the loads form one straight pipeline of depth K5_Q.  The last lane may read up to 3 words past the chunk
    // (next list / tail pad).  (Hand-issued inline-asm loads with a manual s_waitcnt vmcnt(K5_Q - 1) returned stale registers now and then on
    // gfx950 -- tools/probes/vmcnt_order.hip -- so the loads are left to the compiler, which places the counted waits itself.)
    auto issue = [&](const Step& s, k5_u32x4& v) {
      const uint32_t W = (s.n + 63u) >> 6, x = k5_lane_times(lane, W), lim = max(s.n, 1u) - 1u;      // (lim on the scalar side)
      const uint32_t e = (x < lim ? x : lim) << 2;
#ifdef K5_ASM_LOADS
      asm volatile("global_load_dwordx4 %0, %1, %2" : "=&v"(v) : "v"(e), "s"(s.src) : "memory");
#else
      v = *(const k5_u32x4*)((const char*)s.src + e);
#endif
    };
#ifndef K5_ASM_LOADS
// (compiler-tracked loads: the empty statement only pins the four words to one register tuple at the point of use, which keeps the register
// allocator from copying them out of a shared temporary right behind the load -- and waiting for vmcnt(0) there)
#define K5_WAIT_OLDEST(v) asm volatile("" : "+v"(v))
#elif defined(K5_DEBUG_WAIT0)
#define K5_WAIT_OLDEST(v) asm volatile("s_waitcnt vmcnt(0)" : "+v"(v) :: "memory")
#else
#define K5_WAIT_OLDEST(v) asm volatile("s_waitcnt vmcnt(%1)" : "+v"(v) : "n"(K5_Q - 1) : "memory")
#endif
#ifdef K5_TWICE_BRANCH
#define K5_TWICE(av, t) do { if (t) K5_LDS_OR((av) & tmask4, (t)); } while (0)
#else
#define K5_TWICE(av, t) K5_LDS_OR((av) & tmask4, (t))
#endif

    // ================= pass A: marks =================
    {
      // strip entries mark the region before theirs (region 0's strip entries are not in the strip lists)
      for (int j0 = tid >> 2; j0 < nl; j0 += nthr >> 2) {
        const uint4 sr = *(const uint4*)&srec[4 * j0];
        const uint32_t* sp = spos0 + (((uint64_t)sr.y << 32) | sr.x);
        for (uint32_t e = (uint32_t)(tid & 3) * 4u; e < sr.z; e += 16u) {
          const k5_u32x4 v = *(const k5_u32x4*)(sp + e);
          const uint32_t nv = min(4u, sr.z - e);
          mark(v.x - (1u << rb));
          if (nv > 1) mark(v.y - (1u << rb));
          if (nv > 2) mark(v.z - (1u << rb));
          if (nv > 3) mark(v.w - (1u << rb));
        }
      }
      // (the first K5_Q loads go out behind the strip loop, so that the compiler's counter model sees the same six pending loads on both
      // ways into the main loop and waits with vmcnt(K5_Q - 1), not vmcnt(0))
      gen_reset();
      Step sd[K5_Q]; k5_u32x4 sv[K5_Q];
#pragma unroll
      for (int q = 0; q < K5_Q; q++) { gen(sd[q]); issue(sd[q], sv[q]); }
      // No exit from the middle of a round: a step with n == 0 (generator exhausted) is a no-op, so the loop tests sd[0] once per round.  (An
      // exit per step merges, in the structurized loop, the counter states behind each of the K5_Q loads at the loop header, and the compiler
      // then waits for vmcnt(0) at every step: one memory round trip per step instead of a K5_Q-deep pipeline.)
      while (sd[0].n) {
#pragma unroll
        for (int q = 0; q < K5_Q; q++) {
          K5_WAIT_OLDEST(sv[q]);
          // Marks only have to cover every real entry (pass B and the region table apply the exact rule): the last active lane also
          // marks the up-to-3 words it read past the chunk.
          const uint32_t W = (sd[q].n + 63u) >> 6;
          if (k5_lane_times(lane, W) < sd[q].n) {
            // all seen[] updates of the step first, then the twice[] updates (old & m is 0 or m: the update is unconditional, a no-op for
            // first marks -- cheaper than a branch per entry): one LDS round trip per step, not per entry
            const uint32_t a0 = sv[q].x >> sh_a, a1 = sv[q].y >> sh_a, a2 = sv[q].z >> sh_a, a3 = sv[q].w >> sh_a;
            // slots 1 and 2 without a branch: a slot beyond W ORs a zero mask (a no-op that returns the word) -- two scalar selects and two ANDs instead of
            // a compare, a select and a branch per slot and table (the scalar unit, one per CU, is the busier one in this loop); the fourth slot is rare
            const uint32_t k1 = W > 1u ? 0xFFFFFFFFu : 0u, k2 = W > 2u ? 0xFFFFFFFFu : 0u;
            const uint32_t m0 = 1u << ((sv[q].x >> sh_b) & 31u), m1 = (1u << ((sv[q].y >> sh_b) & 31u)) & k1, m2 = (1u << ((sv[q].z >> sh_b) & 31u)) & k2, m3 = 1u << ((sv[q].w >> sh_b) & 31u);
            uint32_t o3 = 0;
            const uint32_t o0 = K5_LDS_OR(swb + (a0 & smask4), m0);
            const uint32_t o1 = K5_LDS_OR(swb + (a1 & smask4), m1);
            const uint32_t o2 = K5_LDS_OR(swb + (a2 & smask4), m2);
            if (W > 3) o3 = K5_LDS_OR(swb + (a3 & smask4), m3);
            K5_TWICE(a0, o0 & m0);
            K5_TWICE(a1, o1 & m1);
            K5_TWICE(a2, o2 & m2);
            if (W > 3) K5_TWICE(a3, o3 & m3);
          }
          gen(sd[q]); issue(sd[q], sv[q]);
        }
      }
      // (the dummy loads of the exhausted generator stay in flight: nothing reads their registers)
    }
    // Pass B streams the same lists again: its first K5_Q loads go out here, before the barrier, so that the wait for the slowest wave, the clear of
    // the region table and the strip loop hide the fill of its load pipeline (~2 us per pass otherwise).
    Step sdB[K5_Q]; k5_u32x4 svB[K5_Q];
    gen_reset();
#pragma unroll
    for (int q = 0; q < K5_Q; q++) { gen(sdB[q]); issue(sdB[q], svB[q]); }
    __syncthreads();
    K5_STAMP(1);
    // ================= pass B: candidates (into the dead seen[] area) =================
    { uint4* h4 = (uint4*)htag; for (int w = tid; w < hclr_q; w += nthr) h4[w] = make_uint4(0, 0, 0, 0); }
    {
      // Candidates are appended to one dense array (positions + y/seed) through a counter in LDS: one atomic per wave and step.  (Per-wave segments
      // with a wave-uniform counter -- no atomic at all -- were measured and dropped: pass B gained 2 %, the two exact stages lost 20 % to the
      // segment -> slot mapping and to half-empty iterations.)
      // a strip entry whose own region stays below 2 is a candidate when the region before reaches 2
      for (int j0 = tid >> 2; j0 < nl; j0 += nthr >> 2) {
        const uint4 sr = *(const uint4*)&srec[4 * j0];
        const uint32_t* sp = spos0 + (((uint64_t)sr.y << 32) | sr.x);
        const uint32_t ysn = rec[4 * j0 + 3];
        for (uint32_t e = (uint32_t)(tid & 3) * 4u; e < sr.z; e += 16u) {
          const k5_u32x4 v = *(const k5_u32x4*)(sp + e);
          const uint32_t nv = min(4u, sr.z - e);               // >= 1
          // (entry 0 is tested unconditionally: a load whose result is only used under a condition keeps the compiler's counter model
          // "pending" into the main loop below, which then waits for vmcnt(0) at every step)
          const bool c0 = !has2(v.x) && has2(v.x - (1u << rb));
          const bool c1 = nv > 1 && !has2(v.y) && has2(v.y - (1u << rb));
          const bool c2 = nv > 2 && !has2(v.z) && has2(v.z - (1u << rb));
          const bool c3 = nv > 3 && !has2(v.w) && has2(v.w - (1u << rb));
          if (c0 | c1 | c2 | c3) {
            uint32_t idx = atomicAdd(&ctrl[C_NCAND], (uint32_t)c0 + (uint32_t)c1 + (uint32_t)c2 + (uint32_t)c3);
            if (idx + 4u <= (uint32_t)cand_cap) {
              const uint16_t y16 = (uint16_t)ysn;
              if (c0) { candp[idx] = v.x; candy[idx++] = y16; }
              if (c1) { candp[idx] = v.y; candy[idx++] = y16; }
              if (c2) { candp[idx] = v.z; candy[idx++] = y16; }
              if (c3) { candp[idx] = v.w; candy[idx++] = y16; }
            }
          }
        }
      }
      Step (&sd)[K5_Q] = sdB; k5_u32x4 (&sv)[K5_Q] = svB;
      while (sd[0].n) {                                                 // (no exit inside a round, as in pass A)
#pragma unroll
        for (int q = 0; q < K5_Q; q++) {
          const Step& s = sd[q];
          K5_WAIT_OLDEST(sv[q]);
          const uint32_t W = (s.n + 63u) >> 6;
          const uint32_t e0 = k5_lane_times(lane, W);
          // the four twice[] words unconditionally (no branch per entry slot; a lane's words past its own entries -- min(W, n - e0) of them --
          // belong to the next lane / list and are masked), the four answers as one bit mask
          const uint32_t mine = min(W, __builtin_elementwise_sub_sat(s.n, e0));
#ifdef K5_ABL_B
          const uint32_t t0 = 0, t1 = 0, t2 = 0, b3 = ((sv[q].x ^ sv[q].y ^ sv[q].z ^ sv[q].w) == 0x12345678u) << 3;
#else
          const uint32_t t0 = *K5_LDS((sv[q].x >> sh_a) & tmask4), t1 = *K5_LDS((sv[q].y >> sh_a) & tmask4);
          const uint32_t t2 = *K5_LDS((sv[q].z >> sh_a) & tmask4);
          // (the fourth slot only in steps of more than 192 entries -- one wave-uniform branch; lists of the default seeds on 3 Gbp hold ~180)
          uint32_t b3 = 0;
          if (W > 3u) b3 = __builtin_amdgcn_ubfe(*K5_LDS((sv[q].w >> sh_a) & tmask4), __builtin_amdgcn_ubfe(sv[q].w, sh_b, 5), 1) << 3;
#endif
          const uint32_t hits = (__builtin_amdgcn_ubfe(t0, __builtin_amdgcn_ubfe(sv[q].x, sh_b, 5), 1) | (__builtin_amdgcn_ubfe(t1, __builtin_amdgcn_ubfe(sv[q].y, sh_b, 5), 1) << 1) |
                                 (__builtin_amdgcn_ubfe(t2, __builtin_amdgcn_ubfe(sv[q].z, sh_b, 5), 1) << 2) | b3) &
                                ((1u << mine) - 1u);
          // one reservation per lane with hits (8 % of the entries are candidates: a dozen lanes per step add to the same LDS word, which the
          // LDS serialises in as many cycles -- cheaper than four ballots, their counts and a prefix per entry)
          const uint32_t cnt = (uint32_t)__popc(hits);
          if (cnt) {
            const uint32_t ci = atomicAdd(&ctrl[C_NCAND], cnt);
            if (ci + cnt <= (uint32_t)cand_cap) {
              // slot k of a lane goes to ci + (hits below k): no running index (and no register copy) between the four predicated store pairs,
              // the list's y / seed word in one register for all of them
              uint32_t y16 = s.ysn;
              asm volatile("" : "+v"(y16));
              const uint32_t c1 = ci + (hits & 1u), c2 = ci + (uint32_t)__popc(hits & 3u), c3 = ci + (uint32_t)__popc(hits & 7u);
              // (absolute LDS addresses: candp[] starts at swb, candy[] behind it -- with compile-time sizes both bases are instruction offsets, a store pair costs two shifts)
              const uint32_t cyb = swb + 4u * (uint32_t)cand_cap;
              if (hits & 1u) { *K5_LDS(swb + (ci << 2)) = sv[q].x; *K5_LDS16(cyb + (ci << 1)) = (uint16_t)y16; }
              if (hits & 2u) { *K5_LDS(swb + (c1 << 2)) = sv[q].y; *K5_LDS16(cyb + (c1 << 1)) = (uint16_t)y16; }
              if (hits & 4u) { *K5_LDS(swb + (c2 << 2)) = sv[q].z; *K5_LDS16(cyb + (c2 << 1)) = (uint16_t)y16; }
              if (hits & 8u) { *K5_LDS(swb + (c3 << 2)) = sv[q].w; *K5_LDS16(cyb + (c3 << 1)) = (uint16_t)y16; }
            }
          }
          gen(sd[q]); issue(sd[q], sv[q]);
        }
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    K5_STAMP(2);
    // Every count the device produced is clamped to the capacity of what it indexes before it bounds a loop: the reservation counter runs past cand_cap when a
    // read-strand has more candidates than slots (the stores are skipped, the read-strand falls back), so it is never used as it stands.
    const uint32_t nc_raw = ctrl[C_NCAND];
    bool fallback = nc_raw > (uint32_t)a.cand_limit;
    const uint32_t nc = min(nc_raw, (uint32_t)cand_cap);
    // ================= exact stage 1: region table over the candidates =================
    // A wave pays the longest chain of LDS round trips among its lanes, so the rare cases are taken out of the main loops and worked off
    // densely afterwards: the overlap-strip marks here (2 % of the candidates, but some lane of most waves), the neighbour-region look-ups of
    // stage 2 below.  Their lists (region numbers, then candidate indexes) live in rec[] / srec[], dead since pass B.
    uint32_t* const elist = rec; const uint32_t ecap = 8u * (uint32_t)RC;
    // Stage 1 in two steps.  (a) every candidate tries the FIRST slot of its region's probe sequence, in straight-line code: one compare-and-swap, and when the slot holds the
    // region already, the flag update -- the same three LDS round trips for every lane.  Two thirds of the candidates are done with that (the table is a third full).
    // (b) the others -- another region sits in their first slot -- are listed and probed on from the second slot, densely, together with the strip marks: a probing loop
    // costs a wave the longest sequence among its lanes (five at this load), and it now pays that once per 64 LISTED candidates, not once per 64 candidates.
    uint32_t* const dlist = elist + 2u * (uint32_t)RC; const uint32_t scap = 2u * (uint32_t)RC, dcap = ecap - scap;
    auto settle = [&](const uint32_t i, const uint32_t h, const uint32_t off) {      // candidate i sits in slot h
      atomicMax(&hmin[h], 0x10000u - off); atomicMax(&hmax[h], off + 1u);
      candp[i] = (h << 16) | off;                              // slot and offset: stage 2 reads the region back from the tag (no second probe sequence)
    };
    auto probe_on = [&](const uint32_t i) {                        // step (b) for candidate i, its strip mark included
      const uint32_t p = candp[i], r = p >> rb, off = p & rmask;
      const uint32_t h = k5_insert(htag, hmask, hshift, r, true, true);
      if (h != 0xFFFFFFFFu) {
        settle(i, h, off);
        if (off < ovl && r > 0 && k5_insert(htag, hmask, hshift, r - 1u, false) == 0xFFFFFFFFu) ctrl[C_OVERFLOW] = 1u;
      } else ctrl[C_OVERFLOW] = 1u;
    };
    if (!fallback) {
      for (uint32_t i0 = 0; i0 < nc; i0 += nthr) {
        const uint32_t i = i0 + tid;
        const bool live = i < nc;
        const uint32_t p = candp[live ? i : 0u], r = p >> rb, off = p & rmask, r1 = r + 1u;
        const uint32_t h0 = k5_hash(r1, hshift);
        uint32_t prev = 0xFFFFFFFFu;
        if (live) prev = atomicCAS(&htag[h0], 0u, (r1 << 8) | K5_FA | K5_FC);
        const bool hit = live && (prev == 0u || (prev >> 8) == r1);
        if (hit) {
          if (prev != 0u) k5_again(htag, h0, prev, true);
          settle(i, h0, off);
        }
        const bool strip = hit && off < ovl && r > 0, later = live && !hit;      // the overlap strip also counts for the region before (ref: mapping.c:521-533)
        const unsigned long long bs = __ballot(strip), bd = __ballot(later);
        if (bs | bd) {
          uint32_t sb = 0, db = 0;
          if (lane == 0) { if (bs) sb = atomicAdd(&ctrl[C_NSTRIP], (uint32_t)__popcll(bs)); if (bd) db = atomicAdd(&ctrl[C_NDEFER], (uint32_t)__popcll(bd)); }
          sb = __builtin_amdgcn_readfirstlane(sb); db = __builtin_amdgcn_readfirstlane(db);
          const unsigned long long below = (1ull << lane) - 1ull;
          if (strip) { const uint32_t j = sb + (uint32_t)__popcll(bs & below); if (j < scap) elist[j] = r - 1u; else ctrl[C_OVERFLOW] = 1u; }
          if (later) { const uint32_t j = db + (uint32_t)__popcll(bd & below); if (j < dcap) dlist[j] = i; else probe_on(i); }      // (a full list: probed on the spot)
        }
      }
    }
    __syncthreads();
    K5_STAMP(10);
    if (!fallback) {
      const uint32_t ns = min(ctrl[C_NSTRIP], scap), nd = min(ctrl[C_NDEFER], dcap);
      for (uint32_t j = tid; j < nd; j += nthr) probe_on(dlist[j]);
      for (uint32_t j = tid; j < ns; j += nthr)
        if (k5_insert(htag, hmask, hshift, elist[j], false) == 0xFFFFFFFFu) ctrl[C_OVERFLOW] = 1u;
    }
    __syncthreads();
    K5_STAMP(3);
    fallback = fallback || rec_over || ctrl[C_OVERFLOW] != 0u;
    // ================= exact stage 2: the reference's rule (ref: mapping.c:733-742), prune rules (gm_prune.hip), output =================
    if (!fallback) {
      unsigned long long* out = (unsigned long long*)a.out + (size_t)rs * a.out_cap;
      // The neighbour regions only matter near the region's ends: a candidate at least D + e_max away from both has no neighbour-region
      // candidate within D of it or of anything within e_max of it, so both rules see the same with the neighbours left out (83 % of the
      // members at 2 048-base regions; each look-up of an absent region is a full probe sequence).
      const uint32_t edge = a.D + (uint32_t)max(a.e_max, 0);
      auto emit = [&](const bool memb, const bool keep, const uint32_t p, const uint32_t i) {      // (called by whole waves)
        const unsigned long long bm = __ballot(memb), bk = __ballot(keep);
        if (bm) {
          uint32_t base = 0;
          if (lane == 0) { atomicAdd(&ctrl[C_NMEMB], (uint32_t)__popcll(bm)); if (bk) base = atomicAdd(&ctrl[C_NKEEP], (uint32_t)__popcll(bk)); }
          base = __builtin_amdgcn_readfirstlane(base);
          if (keep) {
            const uint32_t slot = base + (uint32_t)__popcll(bk & ((1ull << lane) - 1ull));
            const uint32_t y16 = candy[i];
            if (slot < (uint32_t)a.out_cap) out[slot] = ((unsigned long long)p << 32) | ((unsigned long long)(y16 >> 4) << 16) | (y16 & 15u);
          }
        }
      };
      // (2a) every candidate: its region's tag; members away from the region's ends are decided here, the rest goes to the list.  Straight-line code: the three
      // table words of a candidate are read together, the cases are predicates, the list is appended to once per wave (the branchy form of this loop spent more
      // scalar instructions on its control flow than vector instructions on the work, and reloaded kernel arguments inside it).
#ifndef K5_2A_BRANCHY
      {
        // (the kernel arguments of the rules as opaque scalars: left to itself the compiler re-loads them from the argument segment inside the loop, a memory
        // round trip per iteration; predicates combined with & and |, not && and ||: no branch per clause)
        const uint32_t do_prune = k_prune, Dv = k_D, emax_on = k_emax_on, emax = k_emax;
        const uint32_t rsz = 1u << rb;
        for (uint32_t i0 = 0; i0 < nc; i0 += nthr) {
          const uint32_t i = i0 + tid;
          const uint32_t valid = i < nc ? 1u : 0u;
          const uint32_t w = candp[valid ? i : 0u], h = w >> 16, off = w & 0xFFFFu;
          const uint32_t town = htag[h], mn = hmin[h], mx = hmax[h];
          const uint32_t r = (town >> 8) - 1u, p = (r << rb) | off;
          const uint32_t is_m = (town >> 1) & 1u /* K5_FB */, r_pos = r > 0u ? 1u : 0u;
          const uint32_t nb_m = do_prune & (((off < edge ? 1u : 0u) & r_pos) | (off + edge >= rsz ? 1u : 0u)), nb_o = (off < ovl ? 1u : 0u) & r_pos;
          const uint32_t nbv = valid & (is_m ? nb_m : nb_o);
          const bool nb = nbv != 0u;
          const unsigned long long bn = __ballot(nb);
          if (bn) {
            uint32_t base = 0;
            if (lane == 0) base = atomicAdd(&ctrl[C_NEDGE], (uint32_t)__popcll(bn));
            base = __builtin_amdgcn_readfirstlane(base);
            if (nb) {
              const uint32_t j = base + (uint32_t)__popcll(bn & ((1ull << lane) - 1ull));
              if (j < ecap) elist[j] = i; else ctrl[C_OVERFLOW] = 1u;
            }
          }
          const uint32_t fe = (town >> 4) & 1u /* K5_FE */, fd = (town >> 3) & 1u /* K5_FD */;
          const uint32_t span = (mx - 1u) - (0x10000u - mn);
          const uint32_t iso = fe | (fd & (span <= Dv ? 1u : 0u));                          // (1) isolation: three or more inside, or two at most D apart
          const uint32_t tight = emax_on & (span <= emax ? 1u : 0u);                        // (2) tight cluster
          const uint32_t membv = valid & is_m & (nbv ^ 1u), keepv = membv & ((iso & (tight ^ 1u)) | (do_prune ^ 1u));
          emit(membv != 0u, keepv != 0u, p, i);
        }
      }
#else
      // (2a) every candidate: its region's tag; members away from the region's ends are decided here, the rest goes to the list
      for (uint32_t i0 = 0; i0 < nc; i0 += nthr) {
        const uint32_t i = i0 + tid;
        bool memb = false, keep = false; uint32_t p = 0;
        if (i < nc) {
          const uint32_t w = candp[i], h = w >> 16, off = w & 0xFFFFu, town = htag[h], r = (town >> 8) - 1u;
          p = (r << rb) | off;
          const bool is_m = (town & K5_FB) != 0u;
          const bool nb = is_m ? (a.prune && ((off < edge && r > 0) || off + edge >= (1u << rb))) : (off < ovl && r > 0);
          if (nb) {
            const uint32_t j = atomicAdd(&ctrl[C_NEDGE], 1u);
            if (j < ecap) elist[j] = i; else ctrl[C_OVERFLOW] = 1u;
          } else if (is_m) {
            memb = true; keep = true;
            if (a.prune) {
              const uint32_t co = (town & K5_FE) ? 3u : ((town & K5_FD) ? 2u : 1u);
              const uint32_t span = (hmax[h] - 1u) - (0x10000u - hmin[h]);
              keep = co >= 3u || (co == 2u && span <= a.D);                         // (1) isolation
              if (keep && a.e_max >= 0 && span <= (uint32_t)a.e_max) keep = false;    // (2) tight cluster
            }
          }
        }
        emit(memb, keep, p, i);
      }
#endif
      __syncthreads();
      K5_STAMP(11);
      // (2b) the listed candidates, with the regions before / behind theirs
      const uint32_t ne = min(ctrl[C_NEDGE], ecap);
      for (uint32_t j0 = 0; j0 < ne; j0 += nthr) {
        const uint32_t j = j0 + tid;
        bool memb = false, keep = false; uint32_t p = 0, i = 0;
        if (j < ne) {
          i = elist[j];
          const uint32_t w = candp[i], hown = w >> 16, off = w & 0xFFFFu, town = htag[hown], r = (town >> 8) - 1u;
          p = (r << rb) | off;
          uint32_t tlf = 0, trt = 0;
          memb = (town & K5_FB) != 0u;
          uint32_t hlf = 0xFFFFFFFFu; bool have_lf = false;
          if (!memb && off < ovl && r > 0) { hlf = k5_find(htag, hmask, hshift, r - 1u, tlf); have_lf = true; memb = (tlf & K5_FB) != 0u; }
          if (memb) {
            keep = true;
            if (a.prune) {
              uint32_t hrt = 0xFFFFFFFFu; trt = 0;
              if (off < edge) { if (!have_lf && r > 0) hlf = k5_find(htag, hmask, hshift, r - 1u, tlf); } else tlf = 0;
              if (off + edge >= (1u << rb)) hrt = k5_find(htag, hmask, hshift, r + 1u, trt);
              const uint32_t co = (town & K5_FE) ? 3u : ((town & K5_FD) ? 2u : 1u);
              const uint32_t omin = 0x10000u - hmin[hown], omax = hmax[hown] - 1u;
              const bool cl = (tlf & K5_FC) != 0u, cr = (trt & K5_FC) != 0u;
              const uint32_t rbase = r << rb;
              // (1) isolation: another candidate within D (two in the region: the other one is max - min away; three or more: keep)
              keep = co >= 3u || (co == 2u && omax - omin <= a.D);
              uint32_t lmin = 0, lmax = 0, rmin = 0, rmax = 0;
              if (cl) { lmin = 0x10000u - hmin[hlf]; lmax = hmax[hlf] - 1u; }
              if (cr) { rmin = 0x10000u - hmin[hrt]; rmax = hmax[hrt] - 1u; }
              if (!keep && cl) keep = p - (rbase - (1u << rb) + lmax) <= a.D;
              if (!keep && cr) keep = (rbase + (1u << rb) + rmin) - p <= a.D;
              // (2) tight cluster: everything in the three regions (which cover p -+ (D + e_max)) spans at most e_max positions
              if (keep && a.e_max >= 0) {
                uint32_t gmin = rbase + omin, gmax = rbase + omax;
                if (cl) gmin = rbase - (1u << rb) + lmin;
                if (cr) gmax = rbase + (1u << rb) + rmax;
                if (gmax - gmin <= (uint32_t)a.e_max) keep = false;
              }
            }
          }
        }
        emit(memb, keep, p, i);
      }
    }
    __syncthreads();
    if (ctrl[C_OVERFLOW] != 0u) fallback = true;                    // (the list of stage 2 ran over)
    // More kept than K2's LDS tier takes (1 read-strand in 8 000 on the benchmark genome): k_prune's finer bins get the last word before the heavy tier.  The members are
    // all here, so they go to the read-strand's raw row as K1 without the fused rules would have left them, and only k_prune runs for it (list pl_list).  (Sending these
    // read-strands back through the lane-per-list kernel cost 1-6 ms per launch: its few large-LDS workgroups queue behind the other stream's pass-1 workgroups.)
    bool prune_only = false;
    if (!fallback && a.prune && ctrl[C_NKEEP] > (uint32_t)a.out_cap) {
      if (a.raw_out && ctrl[C_NMEMB] <= (uint32_t)a.raw_cap) {
        prune_only = true;
        unsigned long long* raw = (unsigned long long*)a.raw_out + (size_t)rs * a.raw_cap;
        // slab by slab, as the sweep kernels leave them (K1b prunes a long row per slab segment: surv_seg)
        for (int sl = 0; sl < a.n_slabs; sl++) {
        for (uint32_t i0 = 0; i0 < nc; i0 += nthr) {
          const uint32_t i = i0 + tid;
          bool memb = false; uint32_t p = 0;
          if (i < nc) {
            const uint32_t w = candp[i], h = w >> 16, off = w & 0xFFFFu, town = htag[h], r = (town >> 8) - 1u;
            p = (r << rb) | off;
            if ((int)(p >> ix.slab_bits) == sl || (sl == a.n_slabs - 1 && (int)(p >> ix.slab_bits) >= a.n_slabs)) {
              memb = (town & K5_FB) != 0u;
              if (!memb && off < ovl && r > 0) { uint32_t tlf; k5_find(htag, hmask, hshift, r - 1u, tlf); memb = (tlf & K5_FB) != 0u; }
            }
          }
          const unsigned long long bm = __ballot(memb);
          if (bm) {
            uint32_t base = 0;
            if (lane == 0) base = atomicAdd(&ctrl[C_NRAW], (uint32_t)__popcll(bm));
            base = __builtin_amdgcn_readfirstlane(base);
            if (memb) {
              const uint32_t slot = base + (uint32_t)__popcll(bm & ((1ull << lane) - 1ull)), y16 = candy[i];
              if (slot < (uint32_t)a.raw_cap) raw[slot] = ((unsigned long long)p << 32) | ((unsigned long long)(y16 >> 4) << 16) | (y16 & 15u);
            }
          }
        }
        __syncthreads();
        if (tid == 0 && a.surv_seg) { if (sl == 0) a.surv_seg[(size_t)rs * (a.n_slabs + 1)] = 0u; a.surv_seg[(size_t)rs * (a.n_slabs + 1) + sl + 1] = ctrl[C_NRAW]; }
        }
      } else fallback = true;
    }
    K5_STAMP(4);
#ifdef K5_STAMPS
    if (tid == 0) { atomicAdd(&k5_stamps[6], (unsigned long long)nc); if (fallback) atomicAdd(&k5_stamps[7], 1ull); }
#endif
    if (tid == 0) {
      if (fallback) {                                          // candidates beyond the LDS tiers: the slab-sweep kernel redoes this read-strand
        const uint32_t f = atomicAdd(a.fb_cnt, 1u);
        if (f < (uint32_t)a.fb_cap) a.fb_list[f] = (uint32_t)rs; else GS_ADD(a.stats, GS_OVERFLOW_SURV, 1ull);
      } else if (prune_only) {                                 // members in the raw row, segment ends written above
        const uint32_t nm = ctrl[C_NRAW];
        a.surv_cnt[rs] = nm;
        GS_ADD(a.stats, GS_SURVIVORS, (unsigned long long)nm);
        const uint32_t f = atomicAdd(a.pl_cnt, 1u);
        if (f < (uint32_t)a.pl_cap) a.pl_list[f] = (uint32_t)rs; else GS_ADD(a.stats, GS_OVERFLOW_SURV, 1ull);
      } else {
        const uint32_t nm = ctrl[C_NMEMB], nk = ctrl[C_NKEEP];
        a.surv_cnt[rs] = nm;
        GS_ADD(a.stats, GS_SURVIVORS, (unsigned long long)nm);
        if (a.prune) GS_ADD(a.stats, GS_PRUNED, (unsigned long long)(nm - nk));
        const bool heavy = nk > (uint32_t)a.out_cap;
        if (a.prune) a.out_cnt[rs] = heavy ? 0xFFFFFFFFu : nk;
        if (heavy) {
          const uint32_t hs = atomicAdd(a.heavy_cnt, 1u);
          if (hs < (uint32_t)a.heavy_cap) a.heavy_list[hs] = (uint32_t)rs; else GS_ADD(a.stats, GS_OVERFLOW_SURV, 1ull);
        }
      }
      for (int c = 0; c < C_NLISTS; c++) ctrl[c] = 0;
      *cnl = 0; ctrl[C_NRAW] = 0; ctrl[C_NDEFER] = 0;
    }
    { uint4* t4 = (uint4*)smem; for (int w = tid; w < tab_q; w += nthr) t4[w] = make_uint4(0, 0, 0, 0); }
    K5_STAMP(5);
  }
  for (int d = GM_WAVE / 2; d > 0; d >>= 1) { my_lookups += __shfl_down(my_lookups, d); my_entries += __shfl_down(my_entries, d); }
  if (lane == 0) { GS_ADD(a.stats, GS_LOOKUPS, my_lookups); GS_ADD(a.stats, GS_ENTRIES, my_entries); }
}

// ---------------------------------------------------------------------------------------------
// Strip lists: for every list k of a seed, its entries in the first region_overlap bases of a region > 0, in list
// order: spos[sdir[k] .. sdir[k + 1]).  Derived from dir / pos on the device (one flag pass, a scan over 64-entry
// groups, one scatter pass); not part of the index files.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ bool k5_strip(uint32_t p, int rb, uint32_t ovl) { return (p & ((1u << rb) - 1u)) < ovl && (p >> rb) > 0u; }

__global__ void __launch_bounds__(256) k_strip_count(const uint32_t* __restrict__ pos, uint32_t n_pos, int rb, uint32_t ovl, uint32_t* __restrict__ grp_cnt, uint32_t n_groups) {
  const uint32_t lane = threadIdx.x & 63u;
  uint64_t g = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const uint64_t stride = ((uint64_t)gridDim.x * blockDim.x) >> 6;
  for (; g < n_groups; g += stride) {
    const uint64_t i = g * 64 + lane;
    const bool f = i < n_pos && k5_strip(pos[i], rb, ovl);
    const unsigned long long b = __ballot(f);
    if (lane == 0) grp_cnt[g] = (uint32_t)__popcll(b);
  }
}
__global__ void __launch_bounds__(256) k_strip_scatter(const uint32_t* __restrict__ pos, uint32_t n_pos, int rb, uint32_t ovl, const uint32_t* __restrict__ grp_rank,
                                                       uint32_t n_groups, uint32_t* __restrict__ spos) {
  const uint32_t lane = threadIdx.x & 63u;
  uint64_t g = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const uint64_t stride = ((uint64_t)gridDim.x * blockDim.x) >> 6;
  for (; g < n_groups; g += stride) {
    const uint64_t i = g * 64 + lane;
    const uint32_t p = i < n_pos ? pos[i] : 0u;
    const bool f = i < n_pos && k5_strip(p, rb, ovl);
    const unsigned long long b = __ballot(f);
    if (f) spos[grp_rank[g] + (uint32_t)__popcll(b & ((1ull << lane) - 1ull))] = p;
  }
}
__global__ void __launch_bounds__(256) k_strip_dir(const uint32_t* __restrict__ dir, const uint32_t* __restrict__ pos, uint32_t n_pos, uint64_t K, int S, int rb, uint32_t ovl,
                                                   const uint32_t* __restrict__ grp_rank, uint32_t* __restrict__ sdir) {
  uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (; k <= K; k += stride) {
    const uint32_t i = dir[k * (uint64_t)S];                  // first entry of list k (dir[K * S] = n_pos)
    uint32_t r = grp_rank[i >> 6];
    for (uint32_t q = i & ~63u; q < i; q++) r += k5_strip(pos[q], rb, ovl) ? 1u : 0u;
    sdir[k] = r;
  }
}

static std::mutex g_strip_mutex;
int gm_index_derive_strips(GmIndexHost* ix, hipStream_t stream) {
  std::lock_guard<std::mutex> lock(g_strip_mutex);
  if (ix->strips_ready) return GM_OK;
  const int rb = ix->params.region_bits; const uint32_t ovl = (uint32_t)ix->params.region_overlap;
  for (int sn = 0; sn < ix->n_seeds; sn++) {
    GmSeedHost& sd = ix->seeds[sn];
    const uint64_t K = 1ull << sd.kbits;
    const uint32_t n_groups = (uint32_t)(((uint64_t)sd.n_pos + 63) / 64) + 1u;      // + one empty group: rank of n_pos itself
    uint32_t *d_cnt = nullptr, *d_rank = nullptr; void* tmp = nullptr; size_t tmp_bytes = 0;
    GM_HIP(hipMalloc(&d_cnt, (size_t)n_groups * 4)); GM_HIP(hipMalloc(&d_rank, (size_t)n_groups * 4));
    hipLaunchKernelGGL(k_strip_count, dim3(256 * 16), dim3(256), 0, stream, sd.d_pos, sd.n_pos, rb, ovl, d_cnt, n_groups);
    hipError_t e = rocprim::exclusive_scan(nullptr, tmp_bytes, d_cnt, d_rank, 0u, (size_t)n_groups, rocprim::plus<uint32_t>(), stream);
    if (e != hipSuccess) { gm_set_error("exclusive_scan size query: %s", hipGetErrorString(e)); return GM_E_NODEVICE; }
    GM_HIP(hipMalloc(&tmp, tmp_bytes ? tmp_bytes : 16));
    e = rocprim::exclusive_scan(tmp, tmp_bytes, d_cnt, d_rank, 0u, (size_t)n_groups, rocprim::plus<uint32_t>(), stream);
    if (e != hipSuccess) { gm_set_error("exclusive_scan: %s", hipGetErrorString(e)); return GM_E_NODEVICE; }
    uint32_t total = 0;
    GM_HIP(hipMemcpyAsync(&total, d_rank + (n_groups - 1), 4, hipMemcpyDeviceToHost, stream));
    GM_HIP(hipStreamSynchronize(stream));
    sd.n_spos = total;
    GM_HIP(hipMalloc(&sd.d_spos, ((size_t)total + 64) * 4));
    GM_HIP(hipMemsetAsync(sd.d_spos, 0xff, ((size_t)total + 64) * 4, stream));
    GM_HIP(hipMalloc(&sd.d_sdir, (size_t)(K + 1 + 16) * 4));
    hipLaunchKernelGGL(k_strip_scatter, dim3(256 * 16), dim3(256), 0, stream, sd.d_pos, sd.n_pos, rb, ovl, d_rank, n_groups, sd.d_spos);
    hipLaunchKernelGGL(k_strip_dir, dim3(256 * 16), dim3(256), 0, stream, sd.d_dir, sd.d_pos, sd.n_pos, K, ix->n_slabs, rb, ovl, d_rank, sd.d_sdir);
    GM_HIP(hipGetLastError());
    GM_HIP(hipStreamSynchronize(stream));
    (void)hipFree(d_cnt); (void)hipFree(d_rank); (void)hipFree(tmp);
  }
  ix->strips_ready = true;
  return GM_OK;
}

// ---------------------------------------------------------------------------------------------
// launch: returns 1 when v5 ran (0: geometry does not fit, the caller takes another kernel; < 0: error)
// ---------------------------------------------------------------------------------------------
struct K5Scratch { uint32_t* fb = nullptr; uint32_t* pl = nullptr; int fb_cap = 0; int cus = 0; };   // fb: read-strands for the lane-per-list kernel + K1b; pl: for K1b only
static K5Scratch g_k5[16];
static uint32_t* g_k5_flags = nullptr; static uint32_t g_k5_epoch = 0; static int g_k5_flag_cap = 0, g_k5_flag_grid = 0;
void gm_lookup5_set_start_flags(uint32_t* flags, int cap, uint32_t epoch) { g_k5_flags = flags; g_k5_flag_cap = cap; g_k5_epoch = epoch; g_k5_flag_grid = 0; }
int gm_lookup5_start_flag_grid(void) { return g_k5_flag_grid; }

int gm_lookup5_launch(const GmIndexDev& ix, const uint32_t* d_reads, int n_reads, int read_len, int read_words, int max_n_kmers, int NL,
                      uint64_t* d_out, uint32_t* d_out_cnt, int out_cap, uint32_t* d_surv_cnt, int prune, uint32_t D, int e_max,
                      uint32_t* d_heavy_list, uint32_t* d_heavy_cnt, int heavy_cap, unsigned long long* d_stats, hipStream_t stream,
                      uint32_t** fb_list, uint32_t** fb_cnt, int* fb_cap_out,
                      uint64_t* d_raw, int raw_cap, uint32_t* d_surv_seg, uint32_t** pl_list, uint32_t** pl_cnt) {
  int dev = 0; if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return 0;
  if (!ix.seed[0].sdir || ix.region_bits < 9 || ix.region_bits > 16 || NL <= 0) return 0;
  K5Scratch& K = g_k5[dev];
  const bool forced = gm_tune("GM_K1_V5") != nullptr;
  double entries = 0;                                          // expected list entries per read-strand
  for (int sn = 0; sn < ix.n_seeds; sn++) {
    const double lists = std::max(0, read_len - ix.seed[sn].span + 1 - ix.colour);
    entries += lists * (double)ix.seed[sn].n_pos / (double)(1ull << (ix.hflag ? 2 * GM_HASH_TABLE_POWER : 2 * ix.seed[sn].weight));
  }
  // the fixed cost per read-strand (144 KB of table clears, the barriers) pays off from several thousand list entries per read-strand: 50-colour reads on
  // 3 Gbp (17 k entries) take 74 ms per 500 k reads here against 124 ms in the slab-sweep kernel
  if (!forced && entries < 8000.0) return 0;
  // LDS: twice (1/8 of seen) | seen | 32 B per list | codes | control words
  const size_t fixed = (size_t)32 * (NL + K5_XREC) + 2 * (size_t)((read_len + 15) & ~15) + C_WORDS * 4 + GM_MAX_SEEDS * 32;   // (chunk records, + the seed table)
  const size_t budget = 160 * 1024 - 512;
  int lsw = 15;
  if (const char* e = gm_tune("GM_K5_LSW")) lsw = std::max(8, std::min(15, atoi(e)));
  while (lsw >= 8 && fixed + (size_t)(4u << lsw) + (size_t)(4u << (lsw - 3)) > budget) lsw--;
  if (lsw < 8) return 0;
  int threads = 1024;                                          // waves per workgroup: a power of two
  if (const char* e = gm_tune("GM_K1_THREADS")) { const int v = std::max(64, std::min(1024, atoi(e))); threads = 64; while (threads * 2 <= v) threads *= 2; }
  const int ltw = lsw - 3;                                     // twice[] = an eighth of seen[]
  const int hbits = lsw - 2;                                   // the region table (12 B per slot) takes three quarters of the seen[] area,
  const int cand_cap = (int)(((4u << lsw) / 4u / 6u) & ~15u);   // the candidates (6 B each) the rest
  if (cand_cap < 16) return 0;
  int cand_limit = cand_cap - 8;
  if (!forced) {
    // Expected candidates per read-strand: the reference's survivors (entries sharing a region; ~2x the independence estimate with the strip and the
    // echoes of a real hit) + later arrivals on a shared seen[] bit + first arrivals that meet a set twice[] bit.  Read-strands beyond the candidate
    // array fall back to the slab-sweep kernel: when that would be the rule (2 x 150 bp reads on 3 Gbp: ~9 k), k_lookup_v4 is the better kernel.
    const double lam = entries * (double)((1u << ix.region_bits) + ix.region_overlap) / std::max(1.0, (double)ix.total_len);
    const double memb = 2.0 * entries * std::min(1.0, lam), late = 0.5 * entries * std::min(1.0, entries / (double)(32ull << lsw));
    const double first = entries * std::min(1.0, (late + 0.5 * memb) / (double)(32ull << (lsw - 3)));
    if (memb + late + first > 0.85 * cand_cap) return 0;
  }
  if (const char* e = gm_tune("GM_K5_CANDLIMIT")) cand_limit = std::max(1, std::min(cand_limit, atoi(e)));
  if (prune && (D + (uint32_t)std::max(0, e_max) > (1u << ix.region_bits) || D > 0xFFFFu)) return 0;   // the prune rules need bins (= regions) of at least D + e_max positions
  const size_t lds = fixed + (size_t)(4u << lsw) + (size_t)(4u << ltw);
  const int fb_cap = std::max(4096, 2 * n_reads);              // every read-strand may fall back (tiny tables in the tests, repeats)
  if (!K.cus) { if (hipDeviceGetAttribute(&K.cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || K.cus < 1) K.cus = 256; }
  if (fb_cap > K.fb_cap) {
    if (K.fb) { (void)hipDeviceSynchronize(); (void)hipFree(K.fb); (void)hipFree(K.pl); K.fb = nullptr; K.pl = nullptr; K.fb_cap = 0; }
    if (hipMalloc(&K.fb, (size_t)(fb_cap + 4) * 4) != hipSuccess) return 0;
    if (hipMalloc(&K.pl, (size_t)(fb_cap + 4) * 4) != hipSuccess) return 0;
    K.fb_cap = fb_cap;
  }
  uint32_t* const fb_cnt_p = K.fb + K.fb_cap; uint32_t* const pl_cnt_p = K.pl + K.fb_cap;
  if (hipMemsetAsync(fb_cnt_p, 0, 4, stream) != hipSuccess || hipMemsetAsync(pl_cnt_p, 0, 4, stream) != hipSuccess) return GM_E_NODEVICE;
  static GmLdsLimit lim_configured; size_t& configured = lim_configured.cur();
  if (lds > 48 * 1024 && lds > configured) {
    if (hipFuncSetAttribute((const void*)k_lookup_v5<15>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess ||
        hipFuncSetAttribute((const void*)k_lookup_v5<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return 0;
    configured = lds;
  }
  int grid = std::min(2 * n_reads, K.cus);
  if (const char* e = gm_tune("GM_K5_GRID")) grid = std::max(1, std::min(2 * n_reads, atoi(e)));
  K5Args a;
  a.reads = d_reads; a.n_reads = n_reads; a.read_len = read_len; a.read_words = read_words; a.max_n_kmers = max_n_kmers; a.NL = NL;
  a.lsw = lsw; a.ltw = ltw; a.cand_cap = cand_cap; a.hbits = hbits; a.cand_limit = cand_limit;
  a.out = d_out; a.out_cnt = d_out_cnt; a.out_cap = out_cap; a.surv_cnt = d_surv_cnt; a.prune = prune; a.D = D; a.e_max = e_max;
  a.heavy_list = d_heavy_list; a.heavy_cnt = d_heavy_cnt; a.heavy_cap = heavy_cap; a.stats = d_stats;
  a.fb_list = K.fb; a.fb_cnt = fb_cnt_p; a.fb_cap = fb_cap;
  a.raw_out = prune ? d_raw : nullptr; a.raw_cap = raw_cap; a.surv_seg = d_surv_seg; a.n_slabs = ix.n_slabs; a.pl_list = K.pl; a.pl_cnt = pl_cnt_p; a.pl_cap = fb_cap;
  const bool use_flags = g_k5_flags && grid <= g_k5_flag_cap;
  g_k5_flag_grid = use_flags ? grid : 0;
  a.start_flags = use_flags ? g_k5_flags : nullptr; a.start_epoch = g_k5_epoch;
  if (lsw == 15) hipLaunchKernelGGL(k_lookup_v5<15>, dim3(grid), dim3(threads), lds, stream, ix, a);
  else hipLaunchKernelGGL(k_lookup_v5<0>, dim3(grid), dim3(threads), lds, stream, ix, a);
  if (hipGetLastError() != hipSuccess) return GM_E_NODEVICE;
  *fb_list = K.fb; *fb_cnt = fb_cnt_p; *fb_cap_out = fb_cap; if (pl_list) *pl_list = K.pl; if (pl_cnt) *pl_cnt = pl_cnt_p;
  return 1;
}
