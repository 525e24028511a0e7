// gm_region_table.h -- open-addressing table keyed by genome region, with single-shot LDS atomics only (gfx950 only).
// Shared by k_lookup_v5's exact stage (gm_lookup5.hip) and k_prune_v2 (gm_prune.hip).
#pragma once
#include <stdint.h>
#include <hip/hip_runtime.h>

// Three parallel arrays of 2^hbits words:
//   htag[h] = (region + 1) << 8 | flags     A: marked once, B: marked twice or more (the reference's count >= 2), C / D / E: 1 / 2 / >= 3 candidates inside
//   hmin[h] = 0x10000 - smallest offset of a candidate inside the region (0: none), hmax[h] = largest offset + 1 (0: none)
// Only single-shot atomics (one CAS to claim a slot, then OR / MAX): a read that really maps puts ~250 candidates into one region, and a
// compare-and-swap retry loop on that slot serialises them (measured: 19 k cycles per read-strand for the insert phase alone).
#define K5_FA 1u
#define K5_FB 2u
#define K5_FC 4u
#define K5_FD 8u
#define K5_FE 16u
// (region + 1 < 2^24 -- positions are 32 bits, regions hold at least 2^9 of them -- so both products are 24 x 24 bits: v_mul_u32_u24 is a full-rate instruction, the 32-bit
// v_mul_lo_u32 takes four times as long, and an insert or a find needs two of them)
__device__ __forceinline__ uint32_t k5_mul24(uint32_t a, uint32_t c) { uint32_t r; asm("v_mul_u32_u24 %0, %1, %2" : "=v"(r) : "v"(a), "v"(c)); return r; }
__device__ __forceinline__ uint32_t k5_hash(uint32_t r1, int hshift) { return k5_mul24(r1, 0x9E3779u) >> hshift; }
__device__ __forceinline__ uint32_t k5_step(uint32_t r1) { return (k5_mul24(r1, 0x7FEB35u) >> 7) | 1u; }      // odd: the probe sequence h, h + step, ... visits every slot (double hashing: no primary clustering)

// one more mark on a slot that holds region r already (`prev`: the tag word a compare-and-swap or a load returned): the flags this mark adds
__device__ __forceinline__ void k5_again(uint32_t* htag, uint32_t h, uint32_t prev, bool own) {
  const uint32_t first = own ? (K5_FA | K5_FC) : K5_FA;
  uint32_t old = prev;
  if ((old & first) != first) old = atomicOr(&htag[h], first);     // (the claimer's flags are there already)
  uint32_t need = ((old & K5_FA) ? K5_FB : 0u) | ((own && (old & K5_FC)) ? K5_FD : 0u) | ((own && (old & K5_FD)) ? K5_FE : 0u);
  need &= ~old;
  if (need) {
    const uint32_t old2 = atomicOr(&htag[h], need);
    if (own && (need & K5_FD) && (old2 & K5_FD) && !(old2 & K5_FE)) atomicOr(&htag[h], K5_FE);
  }
}
// returns the slot of region r (claiming one if needed) after OR-ing `first` into a fresh slot / `again` bookkeeping into an existing one; 0xFFFFFFFF: table full.
// `second`: the caller has tried the first slot of the probe sequence itself (k_lookup_v5's stage 1) and found another region there.
__device__ __forceinline__ uint32_t k5_insert(uint32_t* htag, uint32_t hmask, int hshift, uint32_t r, bool own, bool second = false) {
  const uint32_t r1 = r + 1u, t = r1 << 8, first = own ? (K5_FA | K5_FC) : K5_FA;
  uint32_t h = k5_hash(r1, hshift); const uint32_t step = k5_step(r1);
  if (second) h = (h + step) & hmask;
  for (uint32_t n = 0; n <= hmask; n++) {
    const uint32_t prev = atomicCAS(&htag[h], 0u, t | first);
    if (prev == 0u) return h;
    if ((prev >> 8) == r1) { k5_again(htag, h, prev, own); return h; }
    h = (h + step) & hmask;
  }
  return 0xFFFFFFFFu;
}
// slot of region r, or 0xFFFFFFFF
__device__ __forceinline__ uint32_t k5_find(const uint32_t* htag, uint32_t hmask, int hshift, uint32_t r, uint32_t& tagword) {
  const uint32_t r1 = r + 1u;
  uint32_t h = k5_hash(r1, hshift); const uint32_t step = k5_step(r1);
  for (uint32_t n = 0; n <= hmask; n++) {
    const uint32_t cur = htag[h];
    if (cur == 0u) break;
    if ((cur >> 8) == r1) { tagword = cur; return h; }
    h = (h + step) & hmask;
  }
  tagword = 0u;
  return 0xFFFFFFFFu;
}

