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
#include <cmath>
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
#define K5_Q 4            // list chunks in flight per wave (r03 sweep, ms per 262 144 reads: 1: 87.3, 2: 71.0, 3: 68.6, 4: 65.2, 5: see DESIGN, 6: 69.9, 10: 74.3)
#endif
enum { C_NCAND = 0, C_OVERFLOW, C_NMEMB, C_NKEEP, C_NSTRIP, C_NEDGE, C_NLISTS /* two words: read-strands alternate */, C_SINK = 8, C_NRAW = 9, C_NDEFER = 10, C_NCAND0 = 11 /* rounds: the candidates of part 0 */, C_SPILL = 12 /* K5_MAX_ROUNDS - 1 words */, C_WORDS = 16 };
#define K5_MAX_ROUNDS 4

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
  int xrec;                 // chunk records beyond one per list
  int rounds; uint32_t round_regs;      // the exact stages run `rounds` times, each over `round_regs` regions of the genome (+ one region either side)
  uint2* spill; int spill_cap;          // rounds > 1: (rounds - 1) rows of spill_cap candidates (position, y / seed) per workgroup
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
// MULTI: long reads (2 x 150 bp on 3 Gbp: 72 k list entries, ~13 k candidates per read-strand) have more candidates than the LDS holds.  The marks of pass A are complete
// whatever comes next, so pass B and the exact stages run a.rounds times, each time over the candidates of one part of the genome: a.round_regs regions plus ONE region
// either side (a region's count takes the strip marks of the region behind it, its members look at the region before it, the prune rules at both neighbours -- with the
// halo all of that is complete for every region of the part itself), and only candidates of the part itself are emitted.
#define K5_KERNEL_HEAD template <int LSWC> __global__ void __launch_bounds__(1024) k_lookup_v5(GmIndexDev ix, K5Args a)
#define K5_KERNEL_MULTI false
#include "gm_lookup5_kernel.inc"
#undef K5_KERNEL_HEAD
#undef K5_KERNEL_MULTI
// (the variant with rounds carries a few more live values; five waves per SIMD as the allocator's target = 96 registers, so that -- like k_lookup_v5<15> -- its four waves per
// SIMD leave 128 registers to the other stream's kernels; what the allocator then keeps in scratch is stored and re-read per read-strand or per round, outside the streaming loops)
#define K5_KERNEL_HEAD __global__ void __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(5, 5))) k_lookup_v5_rounds(GmIndexDev ix, K5Args a)
#define K5_KERNEL_MULTI true
#define K5_KERNEL_LSWC 15
#include "gm_lookup5_kernel.inc"
#undef K5_KERNEL_HEAD
#undef K5_KERNEL_MULTI
#undef K5_KERNEL_LSWC
// The HALF-SIZE shape with rounds (round 4): 512 threads, tables of 2^19 + 2^16 bits -- 80 KB of LDS with the records, so that TWO workgroups share a CU and the phases of two
// read-strands (pass A: memory; pass B: L2 / vector issue; exact stages: LDS round trips) overlap instead of running one after the other.  The smaller seen[] table is a blocked
// Bloom filter (two bits per region, see the kernel body); the candidates (2 720 slots, region table of 4 096) go through the rounds of k_lookup_v5_rounds.
#define K5_KERNEL_HEAD __global__ void __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(5, 5))) k_lookup_v5_half(GmIndexDev ix, K5Args a)
#define K5_KERNEL_MULTI true
#define K5_KERNEL_LSWC 14
#define K5_KERNEL_BLOOM true
#include "gm_lookup5_kernel.inc"
#undef K5_KERNEL_HEAD
#undef K5_KERNEL_BLOOM
#ifdef GM_TUNING
// (the same without the Bloom bits: the measurement beside it, tuning builds only)
#define K5_KERNEL_HEAD __global__ void __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(5, 5))) k_lookup_v5_half_plain(GmIndexDev ix, K5Args a)
#define K5_KERNEL_BLOOM false
#include "gm_lookup5_kernel.inc"
#undef K5_KERNEL_HEAD
#undef K5_KERNEL_BLOOM
#endif
#undef K5_KERNEL_MULTI
#undef K5_KERNEL_LSWC

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
struct K5Scratch { uint32_t* fb = nullptr; uint32_t* pl = nullptr; int fb_cap = 0; int cus = 0; uint2* spill = nullptr; size_t spill_n = 0; };   // fb: read-strands for the lane-per-list kernel + K1b; pl: for K1b only
static K5Scratch g_k5[16][2];                                   // per device, two sets: two mapping calls may be in flight on a device, each with the set its thread was given
static thread_local int g_k5_slot = 0;
void gm_lookup5_set_scratch_slot(int slot) { g_k5_slot = slot & 1; }
// (launch state of the calling thread, like gm_lookup.hip's: set, used and read back by one host thread per launch)
static thread_local uint32_t* g_k5_flags = nullptr; static thread_local uint32_t g_k5_epoch = 0; static thread_local int g_k5_flag_cap = 0, g_k5_flag_grid = 0;
void gm_lookup5_set_start_flags(uint32_t* flags, int cap, uint32_t epoch) { g_k5_flags = flags; g_k5_flag_cap = cap; g_k5_epoch = epoch; g_k5_flag_grid = 0; }
int gm_lookup5_start_flag_grid(void) { return g_k5_flag_grid; }
static thread_local int g_k5_last_rounds = 1, g_k5_last_half = 0;
int gm_lookup5_last_rounds(void) { return g_k5_last_rounds; }      // rounds of the last launch (> 1: k_lookup_v5_rounds)
int gm_lookup5_last_half(void) { return g_k5_last_half; }            // the last launch was k_lookup_v5_half

int gm_lookup5_launch(const GmIndexDev& ix, const uint32_t* d_reads, int n_reads, int read_len, int read_words, int max_n_kmers, int NL,
                      uint64_t* d_out, uint32_t* d_out_cnt, int out_cap, uint32_t* d_surv_cnt, int prune, uint32_t D, int e_max,
                      uint32_t* d_heavy_list, uint32_t* d_heavy_cnt, int heavy_cap, unsigned long long* d_stats, hipStream_t stream,
                      uint32_t** fb_list, uint32_t** fb_cnt, int* fb_cap_out,
                      uint64_t* d_raw, int raw_cap, uint32_t* d_surv_seg, uint32_t** pl_list, uint32_t** pl_cnt) {
  int dev = 0; if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return 0;
  if (!ix.seed[0].sdir || ix.region_bits < 9 || ix.region_bits > 16 || NL <= 0) return 0;
  K5Scratch& K = g_k5[dev][g_k5_slot];
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
  // chunk records beyond one per list: 64, or what is left beside full-size tables when the lists alone nearly fill the rest (150 bp reads: 417 lists)
  const size_t budget = 160 * 1024 - 512;
  const size_t fixed0 = (size_t)24 * NL + 2 * (size_t)((read_len + 15) & ~15) + C_WORDS * 4 + GM_MAX_SEEDS * 32;              // (one record per list: 16 + 8 bytes, + the seed table)
  const size_t full_tables = (size_t)(4u << 15) + (size_t)(4u << 12);
  // (long reads -- two stripes in the vector filter, 450 B of LDS a wave -- keep the extra records few: what the tables leave of a CU's LDS is what pass 1 runs in beside this kernel)
  int xrec = read_len > 128 ? 16 : 64;
  if (fixed0 + 24 * (size_t)xrec + full_tables > budget && fixed0 + 24 * 16 + full_tables <= budget) xrec = (int)((budget - full_tables - fixed0) / 24);
  const size_t fixed = fixed0 + 24 * (size_t)xrec;
  int lsw = 15;
  int threads = 1024;                                          // waves per workgroup: a power of two
  // Colour space with few list entries per read-strand (50-colour reads on 3 Gbp: 17 k): the half-size shape -- tables of 2^19 + 2^16 bits (72 KB) and 512 threads.  The kernel
  // itself is slower that way (89 against 70 ms per 500 k reads), but it leaves half of a CU's LDS and 320 of a SIMD's 512 registers free, and k_pass2_cs_g4 (256 registers,
  // 15 KB of LDS a wave), which cannot run beside the full shape at all, then runs beside it: the step 132 -> 125 ms.  (Letter space: k_pass2_g4 fits beside the full shape.)
  bool small_shape = false;
  if (!forced && ix.colour && NL <= 512 && !gm_tune("GM_K5_LSW") && !gm_tune("GM_K1_THREADS") && !(gm_tune("GM_K5_SMALL") && atoi(gm_tune("GM_K5_SMALL")) == 0)) {
    const double lam = entries * (double)((1u << ix.region_bits) + ix.region_overlap) / std::max(1.0, (double)ix.total_len);
    const double memb = 2.0 * entries * std::min(1.0, lam), late = 0.5 * entries * std::min(1.0, entries / (double)(32ull << 14));
    const double first = entries * std::min(1.0, (late + 0.5 * memb) / (double)(32ull << 11));
    const double cap14 = (double)(((4u << 14) / 4u / 6u) & ~15u);
    if (memb + late + first <= 0.5 * cap14 && fixed + (size_t)(4u << 14) + (size_t)(4u << 11) <= 96 * 1024) { small_shape = true; lsw = 14; threads = 512; }
  }
  // The half-size shape with rounds (k_lookup_v5_half): two 512-thread workgroups per CU, 80 KB of LDS each.  GM_K5_HALF=1 forces it, =2 takes the variant without the
  // Bloom bits (tuning builds), =0 switches it off.
  int half = 0;
  if (const char* e = gm_tune("GM_K5_HALF")) half = atoi(e);
  const size_t half_budget = 80 * 1024 - 512, half_tables = (size_t)(4u << 14) + (size_t)(4u << 11);
  if (half && (ix.region_bits < 11 || NL > 512 || fixed0 + 24 * 8 + half_tables > half_budget)) half = 0;
  if (half) { lsw = 14; threads = 512; xrec = (int)std::min<size_t>(64, (half_budget - half_tables - fixed0) / 24); }
  if (half) if (const char* e = gm_tune("GM_K5_HALF_THREADS")) threads = std::max(64, std::min(1024, atoi(e) & ~63));      // (probe: 640 = ten waves a workgroup, five a SIMD with two workgroups)
  const size_t fixed_h = fixed0 + 24 * (size_t)xrec;
  if (!half)
  if (const char* e = gm_tune("GM_K5_LSW")) lsw = std::max(8, std::min(15, atoi(e)));
  while (!half && lsw >= 8 && fixed + (size_t)(4u << lsw) + (size_t)(4u << (lsw - 3)) > budget) lsw--;
  if (lsw < 8) return 0;
  if (!half)
  if (const char* e = gm_tune("GM_K1_THREADS")) { const int v = std::max(64, std::min(1024, atoi(e))); threads = 64; while (threads * 2 <= v) threads *= 2; }
  const int ltw = lsw - 3;                                     // twice[] = an eighth of seen[]
  const int hbits = lsw - 2;                                   // the region table (12 B per slot) takes three quarters of the seen[] area,
  const int cand_cap = (int)(((4u << lsw) / 4u / 6u) & ~15u);   // the candidates (6 B each) the rest
  if (cand_cap < 16) return 0;
  int cand_limit = cand_cap - 8, rounds = 1;
  if (!forced) {
    // Expected candidates per read-strand: the reference's survivors (entries sharing a region; ~2x the independence estimate with the strip and the
    // echoes of a real hit) + later arrivals on a shared seen[] bit + first arrivals that meet a set twice[] bit.  Read-strands beyond the candidate
    // array fall back to the slab-sweep kernel: when that would be the rule (2 x 150 bp reads on 3 Gbp: ~9 k), k_lookup_v4 is the better kernel.
    const double lam = entries * (double)((1u << ix.region_bits) + ix.region_overlap) / std::max(1.0, (double)ix.total_len);
    // (the Bloom bits of the half-size shape: the rate of a table of twice the size)
    const double memb = 2.0 * entries * std::min(1.0, lam), late = 0.5 * entries * std::min(1.0, entries / (double)(32ull << (lsw + (half == 1 ? 1 : 0))));
    const double first = entries * std::min(1.0, (late + 0.5 * memb) / (double)(32ull << (lsw - 3)));
    // More than one round (pass B and the exact stages per part of the genome) for long reads, as long as the passes over the lists stay cheaper than the separate
    // kernels: up to four.
    // (The estimate is generous for long reads -- 12.8 k against 9.2 k counted for 150-base reads on 3 Gbp -- and a round costs 10 % of the kernel: two parts there, 4.6 k
    // candidates each against 5 456 slots, no read-strand of 262 144 fell back; one that does is redone exactly by the fall-back kernels.)
    if (memb + late + first > 0.85 * cand_cap) {
      rounds = (int)std::ceil((memb + late + first) / ((half ? 0.95 : 1.25) * cand_cap));
      rounds = std::max(rounds, 2);
      if (rounds > K5_MAX_ROUNDS || NL > threads || lsw != (half ? 14 : 15)) return 0;
    }
  }
  if (const char* e = gm_tune("GM_K5_ROUNDS")) { rounds = std::max(1, std::min(K5_MAX_ROUNDS, atoi(e))); if (rounds > 1 && (NL > threads || lsw != (half ? 14 : 15))) return 0; }
  // the genome's regions dealt evenly to the rounds (positions are 32 bits; the last round takes what is left)
  const uint32_t n_regs = (uint32_t)((ix.total_len + (1ull << ix.region_bits) - 1) >> ix.region_bits);
  const uint32_t round_regs = std::max(1u, (n_regs + (uint32_t)rounds - 1u) / (uint32_t)rounds);
  if (const char* e = gm_tune("GM_K5_CANDLIMIT")) cand_limit = std::max(1, std::min(cand_limit, atoi(e)));
  if (prune && (D + (uint32_t)std::max(0, e_max) > (1u << ix.region_bits) || D > 0xFFFFu)) return 0;   // the prune rules need bins (= regions) of at least D + e_max positions
  const size_t lds = (half ? fixed_h : fixed) + (size_t)(4u << lsw) + (size_t)(4u << ltw);
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
        hipFuncSetAttribute((const void*)k_lookup_v5<14>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess ||
        hipFuncSetAttribute((const void*)k_lookup_v5_rounds, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess ||
        hipFuncSetAttribute((const void*)k_lookup_v5_half, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess ||
#ifdef GM_TUNING
        hipFuncSetAttribute((const void*)k_lookup_v5_half_plain, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess ||
#endif
        hipFuncSetAttribute((const void*)k_lookup_v5<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return 0;
    configured = lds;
  }
  // (the half-size shapes: two workgroups a CU.  For the colour-space shape that leaves 16 KB of a CU's LDS, so the colour-space pass 2 runs beside it on the CUs the lookup
  // has not filled yet or has left already rather than on all of them -- but the lookup itself takes 7.1 instead of 11.5 ms per launch (66 k reads on average), and since pass 2 takes eight
  // windows a wave the step is shorter that way: 190 -> 179 ms per 1 M 50-colour reads, tools/k5_shape_cfg4.sh; with the four-window pass 2 of round 3 it made no difference)
  int grid = std::min(2 * n_reads, (half || small_shape) ? 2 * K.cus : K.cus);
  if (const char* e = gm_tune("GM_K5_GRID")) grid = std::max(1, std::min(2 * n_reads, atoi(e)));
  K5Args a;
  a.reads = d_reads; a.n_reads = n_reads; a.read_len = read_len; a.read_words = read_words; a.max_n_kmers = max_n_kmers; a.NL = NL;
  a.lsw = lsw; a.ltw = ltw; a.cand_cap = cand_cap; a.hbits = hbits; a.cand_limit = cand_limit; a.xrec = xrec; a.rounds = rounds; a.round_regs = round_regs;
  a.out = d_out; a.out_cnt = d_out_cnt; a.out_cap = out_cap; a.surv_cnt = d_surv_cnt; a.prune = prune; a.D = D; a.e_max = e_max;
  a.heavy_list = d_heavy_list; a.heavy_cnt = d_heavy_cnt; a.heavy_cap = heavy_cap; a.stats = d_stats;
  a.fb_list = K.fb; a.fb_cnt = fb_cnt_p; a.fb_cap = fb_cap;
  a.raw_out = prune ? d_raw : nullptr; a.raw_cap = raw_cap; a.surv_seg = d_surv_seg; a.n_slabs = ix.n_slabs; a.pl_list = K.pl; a.pl_cnt = pl_cnt_p; a.pl_cap = fb_cap;
  const bool use_flags = g_k5_flags && grid <= g_k5_flag_cap;
  g_k5_flag_grid = use_flags ? grid : 0;
  a.start_flags = use_flags ? g_k5_flags : nullptr; a.start_epoch = g_k5_epoch;
  g_k5_last_rounds = rounds; g_k5_last_half = half;
  a.spill = nullptr; a.spill_cap = cand_cap;
  if (rounds > 1) {
    const size_t need = (size_t)grid * (size_t)(rounds - 1) * (size_t)cand_cap;
    if (need > K.spill_n) {
      if (K.spill) { (void)hipDeviceSynchronize(); (void)hipFree(K.spill); K.spill = nullptr; K.spill_n = 0; }
      if (hipMalloc(&K.spill, need * sizeof(uint2)) != hipSuccess) return 0;
      K.spill_n = need;
    }
    a.spill = K.spill;
  }
  if (half) {
#ifdef GM_TUNING
    if (half == 2) hipLaunchKernelGGL(k_lookup_v5_half_plain, dim3(grid), dim3(threads), lds, stream, ix, a); else
#endif
    hipLaunchKernelGGL(k_lookup_v5_half, dim3(grid), dim3(threads), lds, stream, ix, a);
  } else if (rounds > 1) hipLaunchKernelGGL(k_lookup_v5_rounds, dim3(grid), dim3(threads), lds, stream, ix, a);
  else if (lsw == 15) hipLaunchKernelGGL((k_lookup_v5<15>), dim3(grid), dim3(threads), lds, stream, ix, a);
  else if (lsw == 14) hipLaunchKernelGGL((k_lookup_v5<14>), dim3(grid), dim3(threads), lds, stream, ix, a);
  else hipLaunchKernelGGL((k_lookup_v5<0>), dim3(grid), dim3(threads), lds, stream, ix, a);
  if (hipGetLastError() != hipSuccess) return GM_E_NODEVICE;
  *fb_list = K.fb; *fb_cnt = fb_cnt_p; *fb_cap_out = fb_cap; if (pl_list) *pl_list = K.pl; if (pl_cnt) *pl_cnt = pl_cnt_p;
  return 1;
}
