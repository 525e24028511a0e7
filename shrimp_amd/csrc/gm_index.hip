// gm_index.hip -- on-device construction of the resident seed index (S5).
// Replaces load_genome() (ref: gmapper/genome.c:1012-1182): for every genome position and every
// seed whose span fits since the last N/X or contig start, the k-mer's start position is appended
// to the list of its map index, so lists are ascending.  Here: emit (mapidx, position) keys for
// every position, stable LSD radix sort by mapidx (rocPRIM), then cut the sorted array into
// (k-mer, slab) slices -- see gm_common.h for the layout.
#include <cstring>
#include <algorithm>
#include "gm_common.h"
#include "gm_internal.h"
#include <mutex>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/iterator/counting_iterator.hpp>

// kmer_to_mapidx_orig (ref: gmapper/gmapper.h:349-368): walk the mask from the LSB (= most recent
// base = highest position); every 1-bit appends the base's low 2 bits => most recent base on top.
__device__ __forceinline__ uint32_t gm_nib(const uint32_t* g, uint64_t p) { return (g[p >> 3] >> ((p & 7) * 4)) & 0xf; }

__global__ void __launch_bounds__(256) k_emit_keys(const uint32_t* __restrict__ genome, uint64_t total_len,
                                                   const uint32_t* __restrict__ contig_off, int n_contigs,
                                                   uint64_t mask, int span, int kbits, int hflag, int max_seed_span, uint32_t* __restrict__ keys) {
  uint64_t q = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  const uint32_t invalid = 1u << kbits;
  for (; q < total_len; q += stride) {
    // contig of q: largest cn with contig_off[cn] <= q
    int lo = 0, hi = n_contigs;
    while (hi - lo > 1) { int m = (lo + hi) >> 1; if (contig_off[m] <= q) lo = m; else hi = m; }
    uint64_t cend = contig_off[lo + 1];
    uint32_t key = invalid;
    if (q + span <= cend) {
      // bases q .. q+span-1; mask bit i <-> base at q+span-1-i
      uint32_t mapidx = 0; bool has_n = false;
      // N/X anywhere inside the span (also under a 0 of the mask) resets the reference's `load`
      // counter (ref: genome.c:1147-1150), so the whole span must be free of code 15.
      if (!hflag) {
        for (int t = 0; t < span; t++) {
          uint32_t b = gm_nib(genome, q + span - 1 - t);
          has_n |= (b == 15u);
          if ((mask >> t) & 1) { mapidx = (mapidx << 2) | (b & 3u); }
        }
      } else {                                       // kmer_to_mapidx_hash, ref: gmapper.h:323-336 (see gm_mapidx in gm_common.h)
        const int nw = (max_seed_span + 7) >> 3;
        for (int w = 0; w < nw; w++) {
          uint32_t word = 0;
          for (int t = 0; t < 8; t++) {
            const int age = 8 * w + t;
            if (age < span) { const uint32_t b = gm_nib(genome, q + span - 1 - age); has_n |= (b == 15u); if ((mask >> age) & 1) word |= b << (4 * t); }
          }
          mapidx = gm_hash32(word ^ mapidx);
        }
        mapidx &= (1u << (2 * GM_HASH_TABLE_POWER)) - 1u;
      }
      if (!has_n) key = mapidx;
    }
    keys[q] = key;
  }
}

// first index e with keys[e] >= invalid  (keys sorted)
__global__ void k_count_valid(const uint32_t* __restrict__ keys, uint64_t n, uint32_t invalid, uint32_t* out) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    uint64_t lo = 0, hi = n;
    while (lo < hi) { uint64_t m = (lo + hi) >> 1; if (keys[m] < invalid) lo = m + 1; else hi = m; }
    *out = (uint32_t)lo;
  }
}

// dir[c] = first e with composite(e) >= c, composite = key*S + (pos >> slab_bits), c in [0, K*S]
__global__ void __launch_bounds__(256) k_build_dir(const uint32_t* __restrict__ keys, const uint32_t* __restrict__ pos, uint32_t n_valid,
                                                   int S, int slab_bits, uint64_t KS, uint32_t* __restrict__ dir) {
  uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (; e <= n_valid; e += stride) {
    long long prev = (e == 0) ? -1 : (long long)keys[e - 1] * S + (pos[e - 1] >> slab_bits);
    long long cur = (e == n_valid) ? (long long)KS : (long long)keys[e] * S + (pos[e] >> slab_bits);
    for (long long c = prev + 1; c <= cur; c++) dir[c] = (uint32_t)e;
  }
}

// bucket k = { list length, first min(len, 15) positions }: one 64-byte line per k-mer (S == 1 only)
__global__ void __launch_bounds__(256) k_build_buckets(const uint32_t* __restrict__ dir, const uint32_t* __restrict__ pos, uint64_t K, uint32_t* __restrict__ bkt) {
  uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (; t < K * 16; t += stride) {
    const uint64_t k = t >> 4; const uint32_t w = (uint32_t)(t & 15);
    const uint32_t b = dir[k], e = dir[k + 1];
    bkt[t] = (w == 0) ? (e - b) : ((w - 1 < e - b) ? pos[b + w - 1] : 0xFFFFFFFFu);
  }
}

// directory from CSR lists (index files): dir[k*S + s] = first entry of list k with position >= s << slab_bits
__global__ void __launch_bounds__(256) k_dir_from_lists(const uint32_t* __restrict__ start, const uint32_t* __restrict__ pos, uint64_t K, int S, int slab_bits,
                                                        uint32_t* __restrict__ dir) {
  uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (; k < K; k += stride) {
    const uint32_t b = start[k], e = start[k + 1];
    dir[k * S] = b;
    for (int s = 1; s < S; s++) {
      const uint64_t lim = (uint64_t)s << slab_bits;
      uint32_t lo = b, hi = e;
      while (lo < hi) { const uint32_t m = lo + ((hi - lo) >> 1); if ((uint64_t)pos[m] < lim) lo = m + 1; else hi = m; }
      dir[k * S + s] = lo;
    }
    if (k == K - 1) dir[K * S] = e;
  }
}

// uploads one seed's lists as read from a reference index file (lens[4^w], positions back to back, each list ascending)
int gm_index_from_lists_device(GmIndexHost* ix, int sn, const uint32_t* lens, const uint32_t* pos, uint32_t total) {
  GmSeedHost& sd = ix->seeds[sn];
  const uint64_t K = 1ull << sd.kbits;
  const uint64_t KS = K * (uint64_t)ix->n_slabs;
  std::vector<uint32_t> start(K + 1);
  uint64_t acc = 0;
  for (uint64_t k = 0; k < K; k++) { start[k] = (uint32_t)acc; acc += lens[k]; }
  start[K] = (uint32_t)acc;
  if (acc != total) { gm_set_error("seed %d: list lengths sum to %llu, file says %u", sn, (unsigned long long)acc, total); return GM_E_ARG; }
  uint32_t* d_start = nullptr;
  GM_HIP(hipMalloc(&d_start, (K + 1) * 4));
  GM_HIP(hipMemcpy(d_start, start.data(), (K + 1) * 4, hipMemcpyHostToDevice));
  sd.n_pos = total;
  GM_HIP(hipMalloc(&sd.d_pos, (size_t)(total + 64) * 4));
  GM_HIP(hipMemset(sd.d_pos, 0xff, (size_t)(total + 64) * 4));
  if (total) GM_HIP(hipMemcpy(sd.d_pos, pos, (size_t)total * 4, hipMemcpyHostToDevice));
  GM_HIP(hipMalloc(&sd.d_dir, (size_t)(KS + 1 + 16) * 4));
  hipLaunchKernelGGL(k_dir_from_lists, dim3(256 * 16), dim3(256), 0, 0, d_start, sd.d_pos, K, ix->n_slabs, ix->slab_bits, sd.d_dir);
  if (ix->n_slabs == 1 && (double)total / (double)K <= 12.0 && !gm_tune("GM_NO_BUCKETS")) {
    GM_HIP(hipMalloc(&sd.d_bkt, (size_t)K * 16 * 4));
    hipLaunchKernelGGL(k_build_buckets, dim3(256 * 16), dim3(256), 0, 0, sd.d_dir, sd.d_pos, K, sd.d_bkt);
  }
  GM_HIP(hipDeviceSynchronize());
  (void)hipFree(d_start);
  sd.dir_words = KS + 1;
  return GM_OK;
}

// Which contigs are RNA: uracil and no thymine (ref: common/fasta.c:528-542).  flags[c] collects bit 0 = a U, bit 1 = a T among contig c's letters.
__global__ void __launch_bounds__(256) k_contig_letters(const uint32_t* __restrict__ genome, uint64_t total_len, const uint32_t* __restrict__ contig_off, int n_contigs,
                                                        uint32_t* __restrict__ flags, uint64_t n_words) {
  // every thread takes one contiguous run of words and carries the flags of the contig it is in: an atomic only where the contig changes and at the end of the run --
  // and that last one once per wave when the whole wave sits in one contig (one atomic per word met a T in every word: 3.1 s on a 3 Gbp genome, all on 24 addresses)
  const uint64_t T = (uint64_t)gridDim.x * blockDim.x, tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t per = (n_words + T - 1) / T, w0 = tid * per, w1 = min(n_words, w0 + per);
  int lo = 0; uint32_t f = 0;
  if (w0 < w1 && w0 * 8 < total_len) {
    int hi = n_contigs;
    while (hi - lo > 1) { int m = (lo + hi) >> 1; if (contig_off[m] <= w0 * 8) lo = m; else hi = m; }
    for (uint64_t w = w0; w < w1; w++) {
      const uint64_t p0 = w * 8;
      if (p0 >= total_len) break;
      const uint32_t x = genome[w];
      const uint64_t next = lo + 1 < n_contigs ? (uint64_t)contig_off[lo + 1] : ~0ull;
      if (p0 + 8 <= next && p0 + 8 <= total_len) {                 // the whole word inside the contig: any nibble == 4 / == 3
        const uint32_t xu = x ^ 0x44444444u, xt = x ^ 0x33333333u;
        f |= ((((xu - 0x11111111u) & ~xu) & 0x88888888u) ? 1u : 0u) | ((((xt - 0x11111111u) & ~xt) & 0x88888888u) ? 2u : 0u);
      } else {
        for (int n = 0; n < 8; n++) {
          const uint64_t p = p0 + n;
          if (p >= total_len) break;
          if (lo + 1 < n_contigs && p >= contig_off[lo + 1]) { if (f) atomicOr(&flags[lo], f); f = 0; while (lo + 1 < n_contigs && p >= contig_off[lo + 1]) lo++; }
          const uint32_t b = (x >> (4 * n)) & 0xf;
          f |= (b == 4u ? 1u : 0u) | (b == 3u ? 2u : 0u);
        }
      }
    }
  }
  const int lo0 = __shfl(lo, 0);
  if (__all(lo == lo0)) {
    for (int d = 32; d > 0; d >>= 1) f |= (uint32_t)__shfl_xor((int)f, d);
    if ((threadIdx.x & 63) == 0 && f) atomicOr(&flags[lo0], f);
  } else if (f) atomicOr(&flags[lo], f);
}
static std::mutex g_rna_mutex;
int gm_index_derive_rna(GmIndexHost* ix, hipStream_t stream) {
  std::lock_guard<std::mutex> lock(g_rna_mutex);
  if (ix->rna_ready) return GM_OK;
  uint32_t* d_f = nullptr;
  GM_HIP(hipMalloc(&d_f, (size_t)ix->n_contigs * 4));
  GM_HIP(hipMemsetAsync(d_f, 0, (size_t)ix->n_contigs * 4, stream));
  hipLaunchKernelGGL(k_contig_letters, dim3(256 * 16), dim3(256), 0, stream, ix->d_genome, ix->total_len, ix->d_contig_off, ix->n_contigs, d_f, ix->genome_words);
  GM_HIP(hipGetLastError());
  std::vector<uint32_t> f(ix->n_contigs);
  GM_HIP(hipMemcpyAsync(f.data(), d_f, (size_t)ix->n_contigs * 4, hipMemcpyDeviceToHost, stream));
  GM_HIP(hipStreamSynchronize(stream));
  (void)hipFree(d_f);
  ix->contig_rna.assign(ix->n_contigs, 0);
  bool any = false;
  for (int c = 0; c < ix->n_contigs; c++) { ix->contig_rna[c] = (f[c] & 3u) == 1u ? 1 : 0; any |= ix->contig_rna[c] != 0; }
  ix->genome_is_rna = ix->n_contigs ? ix->contig_rna[ix->n_contigs - 1] : 0;
  if (any) {
    GM_HIP(hipMalloc(&ix->d_contig_rna, (size_t)ix->n_contigs));
    GM_HIP(hipMemcpy(ix->d_contig_rna, ix->contig_rna.data(), (size_t)ix->n_contigs, hipMemcpyHostToDevice));
  }
  ix->rna_ready = true;
  return GM_OK;
}

// colour-space translation of the resident genome (ref: common/fasta.c:586-606, genome.c:1108-1136): colour p =
// lstocs(letter p-1, letter p) with a 'T' before the first letter of every contig; anything but A/C/G/T gives 15 -- in an RNA contig (contig_rna, or null) a U reads as T.
__global__ void __launch_bounds__(256) k_colour_genome(const uint32_t* __restrict__ genome, uint64_t total_len, const uint32_t* __restrict__ contig_off,
                                                       int n_contigs, const uint8_t* __restrict__ contig_rna, uint32_t* __restrict__ genome_cs, uint64_t n_words) {
  uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (; w < n_words; w += stride) {
    uint32_t out = 0;
    for (int n = 0; n < 8; n++) {
      const uint64_t p = w * 8 + n;
      if (p >= total_len) break;
      int lo = 0, hi = n_contigs;
      while (hi - lo > 1) { int m = (lo + hi) >> 1; if (contig_off[m] <= p) lo = m; else hi = m; }
      uint32_t a = (p == contig_off[lo]) ? 3u : gm_nib(genome, p - 1), b = gm_nib(genome, p);
      if (contig_rna && contig_rna[lo]) { a = a == 4u ? 3u : a; b = b == 4u ? 3u : b; }
      out |= ((a > 3u || b > 3u) ? 15u : (a ^ b)) << (4 * n);
    }
    genome_cs[w] = out;
  }
}
int gm_index_colour_genome_device(GmIndexHost* ix, hipStream_t stream) {
  { const int rc = gm_index_derive_rna(ix, stream); if (rc) return rc; }
  if (!ix->d_genome_cs) GM_HIP(hipMalloc(&ix->d_genome_cs, ix->genome_words * 4));
  hipLaunchKernelGGL(k_colour_genome, dim3(256 * 16), dim3(256), 0, stream, ix->d_genome, ix->total_len, ix->d_contig_off, ix->n_contigs, ix->d_contig_rna,
                     ix->d_genome_cs, ix->genome_words);
  GM_HIP(hipGetLastError());
  GM_HIP(hipStreamSynchronize(stream));
  return GM_OK;
}

int gm_index_build_device(GmIndexHost* ix, hipStream_t stream) {
  const uint64_t n = ix->total_len;
  if (ix->params.colour_space) { int rc = gm_index_colour_genome_device(ix, stream); if (rc) return rc; }
  const uint32_t* d_seq = ix->params.colour_space ? ix->d_genome_cs : ix->d_genome;     // the sequence the seeds are cut from (ref: genome.c:1126-1136)
  uint32_t *keys_a = nullptr, *keys_b = nullptr, *vals_b = nullptr, *d_cnt = nullptr;
  void* tmp = nullptr;
  GM_HIP(hipMalloc(&keys_a, n * 4));
  GM_HIP(hipMalloc(&keys_b, n * 4));
  GM_HIP(hipMalloc(&vals_b, n * 4));
  GM_HIP(hipMalloc(&d_cnt, 4));
  size_t tmp_bytes = 0;
  rocprim::counting_iterator<uint32_t> iota(0);
  int maxbits = 0;
  for (int sn = 0; sn < ix->n_seeds; sn++) maxbits = std::max(maxbits, ix->seeds[sn].kbits + 1);
  hipError_t e = rocprim::radix_sort_pairs(nullptr, tmp_bytes, keys_a, keys_b, iota, vals_b, (size_t)n, 0, maxbits, stream);
  if (e != hipSuccess) { gm_set_error("radix_sort_pairs size query: %s", hipGetErrorString(e)); return GM_E_NODEVICE; }
  GM_HIP(hipMalloc(&tmp, tmp_bytes ? tmp_bytes : 16));
  const int grid = 256 * 16;
  for (int sn = 0; sn < ix->n_seeds; sn++) {
    GmSeedHost& sd = ix->seeds[sn];
    const uint64_t K = 1ull << sd.kbits;
    const uint64_t KS = K * (uint64_t)ix->n_slabs;
    hipLaunchKernelGGL(k_emit_keys, dim3(grid), dim3(256), 0, stream, d_seq, n, ix->d_contig_off, ix->n_contigs,
                       sd.mask, sd.span, sd.kbits, ix->params.hash_seeds ? 1 : 0, ix->max_seed_span, keys_a);
    e = rocprim::radix_sort_pairs(tmp, tmp_bytes, keys_a, keys_b, iota, vals_b, (size_t)n, 0, sd.kbits + 1, stream);
    if (e != hipSuccess) { gm_set_error("radix_sort_pairs: %s", hipGetErrorString(e)); return GM_E_NODEVICE; }
    hipLaunchKernelGGL(k_count_valid, dim3(1), dim3(64), 0, stream, keys_b, n, (uint32_t)K, d_cnt);
    uint32_t n_valid = 0;
    GM_HIP(hipMemcpyAsync(&n_valid, d_cnt, 4, hipMemcpyDeviceToHost, stream));
    GM_HIP(hipStreamSynchronize(stream));
    sd.n_pos = n_valid;
    GM_HIP(hipMalloc(&sd.d_pos, (size_t)(n_valid + 64) * 4));
    GM_HIP(hipMemsetAsync(sd.d_pos, 0xff, (size_t)(n_valid + 64) * 4, stream));   // tail pad: 0xffffffff sentinels
    GM_HIP(hipMemcpyAsync(sd.d_pos, vals_b, (size_t)n_valid * 4, hipMemcpyDeviceToDevice, stream));
    GM_HIP(hipMalloc(&sd.d_dir, (size_t)(KS + 1 + 16) * 4));
    hipLaunchKernelGGL(k_build_dir, dim3(grid), dim3(256), 0, stream, keys_b, vals_b, n_valid, ix->n_slabs, ix->slab_bits, KS, sd.d_dir);
    // small genomes (one slab, short lists): add the 64-byte buckets so that a lookup is a single HBM sector
    if (ix->n_slabs == 1 && (double)n_valid / (double)K <= 12.0 && !gm_tune("GM_NO_BUCKETS")) {
      GM_HIP(hipMalloc(&sd.d_bkt, (size_t)K * 16 * 4));
      hipLaunchKernelGGL(k_build_buckets, dim3(grid), dim3(256), 0, stream, sd.d_dir, sd.d_pos, K, sd.d_bkt);
    }
    GM_HIP(hipStreamSynchronize(stream));
    sd.dir_words = KS + 1;
  }
  (void)hipFree(keys_a); (void)hipFree(keys_b); (void)hipFree(vals_b); (void)hipFree(d_cnt); (void)hipFree(tmp);
  return GM_OK;
}
